/* Oracle (TEST INFRASTRUCTURE, never linked into the product): plain-C restatement of
 * the BLAS call sequence the reference issues for one `trainNetwork` step of the
 * ffLayer stack through `BTensor` -> `HMat` (hmatrix -> system BLAS), in double like
 * the apps (`HMatD`, src/TensorOps/BLAS/HMat.hs:35).  PARITY UNPINNED: the reference
 * ships no golden vectors and cannot be built here; this file is validated against
 * the independent numpy restatement (oracle/neuralnet.py) in tests/test_oracle_c.py.
 *
 * Which primitive is called how often per sample is dictated by the composition rule
 * `g3 xs ds = g1 xs (g2 (f1 xs) ds)` (src/TensorOps/Types.hs:155: every node recomputes
 * its left operand's forward pass), by Haskell's laziness (a thunk nobody demands is
 * never run: `add`'s backward ignores its input, TOp.hs:218; `trainNetwork` drops the
 * input's cotangent, FeedForward.hs:142) and by the fixity of `>>>` (infixr 1).  The
 * numpy oracle reproduces those rules; `hmat_call_counts` publishes the counts this
 * file hard-codes and the test asserts they equal the oracle's trace.
 *
 * Primitive forms follow src/TensorOps/BLAS/HMat.hs:135-163 (`gemv alpha a x` first
 * materialises `scale alpha x`, `axpy` is `scale` then `add`) and the dispatch of
 * src/TensorOps/Backend/BTensor.hs:149-174.
 */
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
/* The fp32 build (liboracle_hmat_f32.so: -DHMAT_F32 -fsingle-precision-constant): the SAME text with every `double`
 * a `float` and exp / log resolved to expf / logf -- what `HMat Float` would compute (the class is generic
 * in the element, HMat.hs:103; the apps instantiate Double).  It is the like-for-like CPU number beside the fp32 GPU
 * step; parity claims stay on the fp64 build.  (After the system headers: only this file's own text is re-typed.) */
#ifdef HMAT_F32
#define double float
#define exp(x) expf(x)
#define log(x) logf(x)
#endif

/* the repeated forward passes write the same outputs; keep the compiler from folding them */
#define NOINLINE __attribute__((noinline))
#define BARRIER() __asm__ volatile("" ::: "memory")

/* ---- HMat-level primitives (single thread) ---------------------------------------- */
/* gemv 1 A x  (HMat.hs:147-152): tmp = scale 1 x ; y = A #> tmp */
static NOINLINE void gemv(int n, int m, const double* A, const double* x, double* y, double* tmp) {
  for (int j = 0; j < m; ++j) tmp[j] = 1.0 * x[j];
  for (int i = 0; i < n; ++i) {
    const double* a = A + (size_t)i * m;
    double s = 0.0;
    for (int j = 0; j < m; ++j) s += a[j] * tmp[j];
    y[i] = s;
  }
}
/* gemv 1 (transpB A) x  (BTensor.hs:162,171 on `transp`): y[j] = sum_i A[i][j] x[i] */
static NOINLINE void gemv_t(int n, int m, const double* A, const double* x, double* y, double* tmp) {
  for (int i = 0; i < n; ++i) tmp[i] = 1.0 * x[i];
  for (int j = 0; j < m; ++j) y[j] = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* a = A + (size_t)i * m;
    const double xi = tmp[i];
    for (int j = 0; j < m; ++j) y[j] += a[j] * xi;
  }
}
/* axpy alpha x (Just y)  (HMat.hs:135-139) */
static NOINLINE void axpy(int n, double alpha, const double* x, const double* y, double* out) {
  for (int i = 0; i < n; ++i) out[i] = alpha * x[i] + y[i];
}
/* ger x y  (HMat.hs:144-145), accumulated into a gradient sum: G += x (x) y */
static NOINLINE void ger_acc(int n, int m, const double* x, const double* y, double* G) {
  for (int i = 0; i < n; ++i) {
    double* g = G + (size_t)i * m;
    const double xi = x[i];
    for (int j = 0; j < m; ++j) g[j] += xi * y[j];
  }
}
static NOINLINE void ger(int n, int m, const double* x, const double* y, double* G) {
  for (int i = 0; i < n; ++i) {
    double* g = G + (size_t)i * m;
    const double xi = x[i];
    for (int j = 0; j < m; ++j) g[j] = xi * y[j];
  }
}
static NOINLINE double dot(int n, const double* x, const double* y) { /* HMat.hs:141-142 */
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i] * y[i];
  return s;
}
static NOINLINE double sum_b(int n, const double* x) { /* sumB, HMat.hs:228-231 */
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += x[i];
  return s;
}
static double logistic(double x) { return 1.0 / (1.0 + exp(-x)); } /* NeuralNet.hs:42-44 */
/* `diff logistic` by forward-mode dual numbers (TOp.hs:212): (1/(1+e))' with e = exp(-x) */
static double dlogistic_ad(double x) {
  const double e = exp(-x), u = 1.0 + e, q = 1.0 / u;
  return -(q / u) * (e * -1.0);
}

/* per-sample call counts of the MNIST-style stack (hidden `actMap logistic`, output
 * `actSoftmax`, loss `crossEntropy`), as traced by the numpy oracle under lazy evaluation
 * with only the parameter cotangents demanded (FeedForward.hs:142) */
enum {
  N_GEMV_L1 = 3,   /* W1 x : forward of layer 1 runs three times               */
  N_ADD_B1 = 3,    /* + b1                                                      */
  N_LOGISTIC = 2,  /* map logistic on z1                                        */
  N_GEMV_L2 = 2,   /* W2 h                                                      */
  N_ADD_B2 = 2,    /* + b2                                                      */
  N_EXP = 2,       /* softmax: map exp                                          */
  N_SUMROWS = 3,   /* softmax: sumRows (once more inside `sumRows >>> map recip`) */
  N_RECIP = 2,     /* softmax: map recip                                        */
  N_SCALE_SV = 1,  /* softmax: outer LZ (LS LZ) = axpy r e (BTensor.hs:155)     */
  N_LOG = 1        /* crossEntropy: map log                                     */
};

void hmat_call_counts(int* out) {
  out[0] = N_GEMV_L1; out[1] = N_ADD_B1; out[2] = N_LOGISTIC; out[3] = N_GEMV_L2; out[4] = N_ADD_B2;
  out[5] = N_EXP; out[6] = N_SUMROWS; out[7] = N_RECIP; out[8] = N_SCALE_SV; out[9] = N_LOG;
}

typedef struct {
  int i, h, o;
  double *t1, *z1, *hh, *t2, *z2, *e, *yh, *lg, *dyh, *de, *dz2, *dh, *dz1, *tmp, *bc;
} Work;

static Work work_new(int i, int h, int o) {
  Work w;
  w.i = i; w.h = h; w.o = o;
  const int mx = i > h ? (i > o ? i : o) : (h > o ? h : o);
  w.t1 = malloc(sizeof(double) * h); w.z1 = malloc(sizeof(double) * h); w.hh = malloc(sizeof(double) * h);
  w.t2 = malloc(sizeof(double) * o); w.z2 = malloc(sizeof(double) * o); w.e = malloc(sizeof(double) * o);
  w.yh = malloc(sizeof(double) * o); w.lg = malloc(sizeof(double) * o); w.dyh = malloc(sizeof(double) * o);
  w.de = malloc(sizeof(double) * o); w.dz2 = malloc(sizeof(double) * o); w.dh = malloc(sizeof(double) * h);
  w.dz1 = malloc(sizeof(double) * h); w.tmp = malloc(sizeof(double) * mx); w.bc = malloc(sizeof(double) * o);
  return w;
}
static void work_free(Work* w) {
  free(w->t1); free(w->z1); free(w->hh); free(w->t2); free(w->z2); free(w->e); free(w->yh);
  free(w->lg); free(w->dyh); free(w->de); free(w->dz2); free(w->dh); free(w->dz1); free(w->tmp); free(w->bc);
}

/* parameter cotangents of ONE sample of the MNIST-style stack.
 * recompute != 0: run every forward primitive as often as the reference does.
 * acc != 0: add into gW1.. (batched sum); else overwrite. Returns the loss. */
static double netgrad_mnist(Work* w, const double* x, const double* y, const double* W1,
                            const double* b1, const double* W2, const double* b2, double* gW1,
                            double* gb1, double* gW2, double* gb2, int recompute, int acc) {
  const int I = w->i, H = w->h, O = w->o;
  double s = 0.0, r = 0.0;
  /* forward (the extra passes reproduce Types.hs:155) */
  for (int k = 0; k < (recompute ? N_GEMV_L1 : 1); ++k) { gemv(H, I, W1, x, w->t1, w->tmp); BARRIER(); }
  for (int k = 0; k < (recompute ? N_ADD_B1 : 1); ++k) { axpy(H, 1.0, w->t1, b1, w->z1); BARRIER(); } /* sumT [t1,b1], BTensor.hs:112 */
  for (int k = 0; k < (recompute ? N_LOGISTIC : 1); ++k) {
    BARRIER();
    for (int j = 0; j < H; ++j) w->hh[j] = logistic(w->z1[j]);                            /* liftB cmap, HMat.hs:120-122 */
  }
  for (int k = 0; k < (recompute ? N_GEMV_L2 : 1); ++k) { gemv(O, H, W2, w->hh, w->t2, w->tmp); BARRIER(); }
  for (int k = 0; k < (recompute ? N_ADD_B2 : 1); ++k) axpy(O, 1.0, w->t2, b2, w->z2);
  for (int k = 0; k < (recompute ? N_EXP : 1); ++k)
    for (int j = 0; j < O; ++j) w->e[j] = exp(w->z2[j]);
  for (int k = 0; k < (recompute ? N_SUMROWS : 1); ++k) s = sum_b(O, w->e);               /* BTensor.hs:768 */
  for (int k = 0; k < (recompute ? N_RECIP : 1); ++k) r = 1.0 / s;
  for (int k = 0; k < (recompute ? N_SCALE_SV : 1); ++k)
    for (int j = 0; j < O; ++j) w->yh[j] = r * w->e[j];                                   /* axpy r e Nothing */
  for (int k = 0; k < (recompute ? N_LOG : 1); ++k)
    for (int j = 0; j < O; ++j) w->lg[j] = log(w->yh[j]);
  const double loss = -dot(O, w->lg, y);
  /* backward, seed 1 (Types.hs:127-132) */
  const double dneg = -1.0 * 1.0;                                    /* negate: scaleT (-1) */
  for (int j = 0; j < O; ++j) w->dyh[j] = (dneg * y[j]) * (1.0 / w->yh[j]); /* dot grad (axpy) then d * (diff log) */
  /* softmax backward: outer LZ (LS LZ) r e */
  const double dr = dot(O, w->dyh, w->e);                            /* gmul dyh (transp e) -> dot */
  for (int j = 0; j < O; ++j) w->de[j] = r * w->dyh[j];              /* gmul (transp r) dyh -> axpy */
  const double ds = dr * (-(r * r));                                 /* d * diff recip (s) = -(1/s)^2 */
  for (int j = 0; j < O; ++j) w->bc[j] = ds;                         /* sumRows grad: mapRows (\_ -> ds) */
  for (int j = 0; j < O; ++j) w->de[j] = w->de[j] + w->bc[j];        /* duplicate grad: sumT [d1,d2] */
  for (int j = 0; j < O; ++j) w->dz2[j] = w->de[j] * w->e[j];        /* d * diff exp (z2) = d * exp z2 */
  /* layer 2: add grad passes dz2 to both; matVec grad: dW2 = ger dz2 h, dh = W2^T dz2 */
  if (acc) {
    for (int j = 0; j < O; ++j) gb2[j] += w->dz2[j];
    ger_acc(O, H, w->dz2, w->hh, gW2);
  } else {
    memcpy(gb2, w->dz2, sizeof(double) * O);
    ger(O, H, w->dz2, w->hh, gW2);
  }
  gemv_t(O, H, W2, w->dz2, w->dh, w->tmp);
  for (int j = 0; j < H; ++j) w->dz1[j] = w->dh[j] * dlogistic_ad(w->z1[j]);
  if (acc) {
    for (int j = 0; j < H; ++j) gb1[j] += w->dz1[j];
    ger_acc(H, I, w->dz1, x, gW1);
  } else {
    memcpy(gb1, w->dz1, sizeof(double) * H);
    ger(H, I, w->dz1, x, gW1);
  }
  /* dx = W1^T dz1 is never demanded (FeedForward.hs:142, laziness) */
  return loss;
}

/* G = sum_b networkGradient(x_b, y_b) at fixed parameters; returns sum of losses */
double hmat_batched_grads(int B, int i, int h, int o, const double* X, const double* Y,
                          const double* W1, const double* b1, const double* W2, const double* b2,
                          double* gW1, double* gb1, double* gW2, double* gb2, int recompute) {
  Work w = work_new(i, h, o);
  memset(gW1, 0, sizeof(double) * (size_t)h * i);
  memset(gb1, 0, sizeof(double) * h);
  memset(gW2, 0, sizeof(double) * (size_t)o * h);
  memset(gb2, 0, sizeof(double) * o);
  double loss = 0.0;
  for (int b = 0; b < B; ++b)
    loss += netgrad_mnist(&w, X + (size_t)b * i, Y + (size_t)b * o, W1, b1, W2, b2, gW1, gb1, gW2, gb2,
                          recompute, 1);
  work_free(&w);
  return loss;
}

/* the reference's own loop: per-sample online SGD, `trainAll = foldl' trainNetwork`
 * (app/MNIST.hs:390-396): p' = p - r * g after EVERY sample (FeedForward.hs:141-147) */
double hmat_train_online(int B, int i, int h, int o, const double* X, const double* Y, double* W1,
                         double* b1, double* W2, double* b2, double rate, int recompute) {
  Work w = work_new(i, h, o);
  double* gW1 = malloc(sizeof(double) * (size_t)h * i);
  double* gb1 = malloc(sizeof(double) * h);
  double* gW2 = malloc(sizeof(double) * (size_t)o * h);
  double* gb2 = malloc(sizeof(double) * o);
  double loss = 0.0;
  for (int b = 0; b < B; ++b) {
    loss += netgrad_mnist(&w, X + (size_t)b * i, Y + (size_t)b * o, W1, b1, W2, b2, gW1, gb1, gW2, gb2,
                          recompute, 0);
    for (size_t k = 0; k < (size_t)h * i; ++k) W1[k] = W1[k] - rate * gW1[k]; /* liftB zipWith */
    for (int k = 0; k < h; ++k) b1[k] = b1[k] - rate * gb1[k];
    for (size_t k = 0; k < (size_t)o * h; ++k) W2[k] = W2[k] - rate * gW2[k];
    for (int k = 0; k < o; ++k) b2[k] = b2[k] - rate * gb2[k];
  }
  free(gW1); free(gb1); free(gW2); free(gb2);
  work_free(&w);
  return loss;
}

/* `gemm 1 A B Nothing` = `a <> scale 1 b` (HMat.hs:154-159), row-major, i-k-j order */
void hmat_gemm(int n, int o, int m, const double* A, const double* B, double* C) {
  memset(C, 0, sizeof(double) * (size_t)n * m);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < o; ++k) {
      const double a = A[(size_t)i * o + k];
      const double* b = B + (size_t)k * m;
      double* c = C + (size_t)i * m;
      for (int j = 0; j < m; ++j) c[j] += a * b[j];
    }
}

/* `liftB` 1-ary with logistic = `cmap` (HMat.hs:120-122) */
void hmat_map_logistic(long n, const double* x, double* y) {
  for (long k = 0; k < n; ++k) y[k] = logistic(x[k]);
}

/* ---- the app's stack: any number of ffLayers (app/MNIST.hs:89-133 defaults: 784 -> 300 -> 100 -> 10) ----------------
 * `genNet (layers `zip` repeat (actMap logistic)) actSoftmax` (FeedForward.hs:216-235, app/MNIST.hs:262-263), loss
 * crossEntropy, per-sample `trainNetwork` (FeedForward.hs:131-148).  The per-sample primitive counts, traced with the
 * numpy oracle under lazy evaluation on 2-, 3- and 4-layer stacks (tests/test_oracle_c.py asserts them): every HIDDEN
 * layer's `W a`, `+ b` run three times and its `map logistic` twice, the LAST layer's `W a`, `+ b` twice, the softmax /
 * crossEntropy head as in netgrad_mnist above -- whatever the depth (the composition is right-nested, `infixr`, so a
 * layer's forward pass is recomputed by the node above it and by the gradient's own `f1 xs`, Types.hs:155). */
enum { N_HID_GEMV = 3, N_HID_ADD = 3, N_HID_LOGISTIC = 2, N_LAST_GEMV = 2, N_LAST_ADD = 2 };
void hmat_stack_call_counts(int* out) {
  out[0] = N_HID_GEMV; out[1] = N_HID_ADD; out[2] = N_HID_LOGISTIC; out[3] = N_LAST_GEMV; out[4] = N_LAST_ADD;
}

typedef struct {
  int L;
  const int* d;                 /* L + 1 widths */
  double **t, **z, **a, **dz;   /* per layer l = 1..L (index l) */
  double *e, *yh, *lg, *dyh, *de, *bc, *dh, *tmp;
} StackWork;

static StackWork stack_new(int L, const int* d) {
  StackWork w;
  w.L = L; w.d = d;
  int mx = 1;
  for (int l = 0; l <= L; ++l) if (d[l] > mx) mx = d[l];
  w.t = malloc(sizeof(double*) * (L + 1)); w.z = malloc(sizeof(double*) * (L + 1));
  w.a = malloc(sizeof(double*) * (L + 1)); w.dz = malloc(sizeof(double*) * (L + 1));
  w.t[0] = w.z[0] = w.a[0] = w.dz[0] = NULL;
  for (int l = 1; l <= L; ++l) {
    w.t[l] = malloc(sizeof(double) * d[l]); w.z[l] = malloc(sizeof(double) * d[l]);
    w.a[l] = malloc(sizeof(double) * d[l]); w.dz[l] = malloc(sizeof(double) * d[l]);
  }
  const int O = d[L];
  w.e = malloc(sizeof(double) * O); w.yh = malloc(sizeof(double) * O); w.lg = malloc(sizeof(double) * O);
  w.dyh = malloc(sizeof(double) * O); w.de = malloc(sizeof(double) * O); w.bc = malloc(sizeof(double) * O);
  w.dh = malloc(sizeof(double) * mx); w.tmp = malloc(sizeof(double) * mx);
  return w;
}
static void stack_free(StackWork* w) {
  for (int l = 1; l <= w->L; ++l) { free(w->t[l]); free(w->z[l]); free(w->a[l]); free(w->dz[l]); }
  free(w->t); free(w->z); free(w->a); free(w->dz);
  free(w->e); free(w->yh); free(w->lg); free(w->dyh); free(w->de); free(w->bc); free(w->dh); free(w->tmp);
}
/* the flat parameter buffer: W_1 (d1 x d0), b_1 (d1), W_2, b_2, ... */
static size_t stack_offsets(int L, const int* d, size_t* offW, size_t* offb) {
  size_t o = 0;
  for (int l = 1; l <= L; ++l) {
    offW[l] = o; o += (size_t)d[l] * d[l - 1];
    offb[l] = o; o += (size_t)d[l];
  }
  return o;
}

/* the head on z_L (softmax >>> crossEntropy against y): loss, and dz_L -- the statements of netgrad_mnist */
static double stack_head(StackWork* w, const double* y, int recompute) {
  const int O = w->d[w->L];
  const double* z2 = w->z[w->L];
  double s = 0.0, r = 0.0;
  for (int k = 0; k < (recompute ? N_EXP : 1); ++k)
    for (int j = 0; j < O; ++j) w->e[j] = exp(z2[j]);
  for (int k = 0; k < (recompute ? N_SUMROWS : 1); ++k) s = sum_b(O, w->e);
  for (int k = 0; k < (recompute ? N_RECIP : 1); ++k) r = 1.0 / s;
  for (int k = 0; k < (recompute ? N_SCALE_SV : 1); ++k)
    for (int j = 0; j < O; ++j) w->yh[j] = r * w->e[j];
  for (int k = 0; k < (recompute ? N_LOG : 1); ++k)
    for (int j = 0; j < O; ++j) w->lg[j] = log(w->yh[j]);
  const double loss = -dot(O, w->lg, y);
  const double dneg = -1.0 * 1.0;
  for (int j = 0; j < O; ++j) w->dyh[j] = (dneg * y[j]) * (1.0 / w->yh[j]);
  const double dr = dot(O, w->dyh, w->e);
  for (int j = 0; j < O; ++j) w->de[j] = r * w->dyh[j];
  const double ds = dr * (-(r * r));
  for (int j = 0; j < O; ++j) w->bc[j] = ds;
  for (int j = 0; j < O; ++j) w->de[j] = w->de[j] + w->bc[j];
  for (int j = 0; j < O; ++j) w->dz[w->L][j] = w->de[j] * w->e[j];
  return loss;
}

/* parameter cotangents of ONE sample into g (same layout as the parameters), overwriting; returns the loss */
static double netgrad_stack(StackWork* w, const double* x, const double* y, const double* p, double* g,
                            const size_t* offW, const size_t* offb, int recompute) {
  const int L = w->L;
  const int* d = w->d;
  const double* in = x;
  for (int l = 1; l <= L; ++l) {
    const int last = l == L;
    for (int k = 0; k < (recompute ? (last ? N_LAST_GEMV : N_HID_GEMV) : 1); ++k) { gemv(d[l], d[l - 1], p + offW[l], in, w->t[l], w->tmp); BARRIER(); }
    for (int k = 0; k < (recompute ? (last ? N_LAST_ADD : N_HID_ADD) : 1); ++k) { axpy(d[l], 1.0, w->t[l], p + offb[l], w->z[l]); BARRIER(); }
    if (!last) {
      for (int k = 0; k < (recompute ? N_HID_LOGISTIC : 1); ++k) {
        BARRIER();
        for (int j = 0; j < d[l]; ++j) w->a[l][j] = logistic(w->z[l][j]);
      }
      in = w->a[l];
    }
  }
  const double loss = stack_head(w, y, recompute);
  for (int l = L; l >= 1; --l) {
    const double* a_in = l == 1 ? x : w->a[l - 1];
    memcpy(g + offb[l], w->dz[l], sizeof(double) * d[l]);             /* add grad: dz to the bias */
    ger(d[l], d[l - 1], w->dz[l], a_in, g + offW[l]);                  /* matVec grad: dW = ger dz a */
    if (l > 1) {                                                       /* da = W^T dz ; dz_{l-1} = da * diff logistic (z) */
      gemv_t(d[l], d[l - 1], p + offW[l], w->dz[l], w->dh, w->tmp);
      for (int j = 0; j < d[l - 1]; ++j) w->dz[l - 1][j] = w->dh[j] * dlogistic_ad(w->z[l - 1][j]);
    }                                                                  /* (dx of layer 1 is never demanded) */
  }
  return loss;
}

/* `trainAll = foldl' trainNetwork` (app/MNIST.hs:390-396) over B samples on an L-layer stack; params updated in place */
double hmat_train_online_stack(int B, int L, const int* dims, const double* X, const double* Y, double* params,
                               double rate, int recompute) {
  size_t offW[16], offb[16];
  if (L < 1 || L > 15) return -1.0;
  const size_t n = stack_offsets(L, dims, offW, offb);
  StackWork w = stack_new(L, dims);
  double* g = malloc(sizeof(double) * n);
  double loss = 0.0;
  for (int s = 0; s < B; ++s) {
    loss += netgrad_stack(&w, X + (size_t)s * dims[0], Y + (size_t)s * dims[L], params, g, offW, offb, recompute);
    for (size_t k = 0; k < n; ++k) params[k] = params[k] - rate * g[k];   /* TT.zip (\p g -> p - r*g), one liftB per tensor */
  }
  free(g);
  stack_free(&w);
  return loss;
}

/* validation as the app runs it (app/MNIST.hs:366-389): `runNetwork` once per sample (runTOp: every primitive once),
 * then `argMax` (Tensor.hs:295-302: the first maximal element) */
void hmat_classify_stack(int B, int L, const int* dims, const double* X, const double* params, int* out) {
  size_t offW[16], offb[16];
  if (L < 1 || L > 15) return;
  stack_offsets(L, dims, offW, offb);
  StackWork w = stack_new(L, dims);
  const int O = dims[L];
  for (int s = 0; s < B; ++s) {
    const double* in = X + (size_t)s * dims[0];
    for (int l = 1; l <= L; ++l) {
      gemv(dims[l], dims[l - 1], params + offW[l], in, w.t[l], w.tmp);
      axpy(dims[l], 1.0, w.t[l], params + offb[l], w.z[l]);
      if (l < L) {
        for (int j = 0; j < dims[l]; ++j) w.a[l][j] = logistic(w.z[l][j]);
        in = w.a[l];
      }
    }
    for (int j = 0; j < O; ++j) w.e[j] = exp(w.z[L][j]);
    const double r = 1.0 / sum_b(O, w.e);
    for (int j = 0; j < O; ++j) w.yh[j] = r * w.e[j];
    int best = 0;
    for (int j = 1; j < O; ++j)
      if (w.yh[j] > w.yh[best]) best = j;
    out[s] = best;
  }
  stack_free(&w);
}

/* ---- BASELINE.md section 3, legs CPU-B and CPU-D: the same restatement over all host cores ---------------------------
 * (reported baselines only; nothing here is a reference for parity) */
#include <pthread.h>

typedef struct GradShared GradShared;
typedef struct {
  int t, b0, b1;
  GradShared* sh;
  double *gW1, *gb1, *gW2, *gb2;   /* this thread's own sums */
  double loss;
} GradJob;
struct GradShared {
  int threads, i, h, o, recompute;
  const double *X, *Y, *W1, *b1, *W2, *b2;
  double *oW1, *ob1, *oW2, *ob2;   /* the caller's outputs */
  GradJob* jobs;
  pthread_barrier_t bar;
};

static void reduce_slice(const GradShared* sh, int t, size_t n, double* out, size_t which) {
  const size_t k0 = n * (size_t)t / (size_t)sh->threads, k1 = n * (size_t)(t + 1) / (size_t)sh->threads;
  for (size_t k = k0; k < k1; ++k) {
    double s = 0.0;
    for (int u = 0; u < sh->threads; ++u) {   /* thread order: the same bits whatever the timing */
      const GradJob* j = &sh->jobs[u];
      const double* g = which == 0 ? j->gW1 : which == 1 ? j->gb1 : which == 2 ? j->gW2 : j->gb2;
      s += g[k];
    }
    out[k] = s;
  }
}

static void* grad_worker(void* arg) {
  GradJob* j = (GradJob*)arg;
  GradShared* sh = j->sh;
  Work w = work_new(sh->i, sh->h, sh->o);
  j->loss = 0.0;
  memset(j->gW1, 0, sizeof(double) * (size_t)sh->h * sh->i); memset(j->gb1, 0, sizeof(double) * sh->h);
  memset(j->gW2, 0, sizeof(double) * (size_t)sh->o * sh->h); memset(j->gb2, 0, sizeof(double) * sh->o);
  for (int b = j->b0; b < j->b1; ++b)
    j->loss += netgrad_mnist(&w, sh->X + (size_t)b * sh->i, sh->Y + (size_t)b * sh->o, sh->W1, sh->b1, sh->W2, sh->b2,
                             j->gW1, j->gb1, j->gW2, j->gb2, sh->recompute, 1);
  work_free(&w);
  pthread_barrier_wait(&sh->bar);
  /* every thread adds up ITS slice of the parameters over all threads' sums */
  reduce_slice(sh, j->t, (size_t)sh->h * sh->i, sh->oW1, 0);
  reduce_slice(sh, j->t, (size_t)sh->h, sh->ob1, 1);
  reduce_slice(sh, j->t, (size_t)sh->o * sh->h, sh->oW2, 2);
  reduce_slice(sh, j->t, (size_t)sh->o, sh->ob2, 3);
  return NULL;
}

/* CPU-B: hmat_batched_grads with the samples split over `threads` pthreads -- every thread the per-sample gemv / ger /
 * axpy / liftB sequence on its own contiguous run of samples into its own gradient buffers; behind a barrier every
 * thread adds up its slice of the parameters over all the buffers, in thread order.  (The reference has no such loop:
 * `foldl' trainNetwork` is sequential, app/MNIST.hs:390-396; this is what "the CPU path over all host cores" can mean
 * for a summed gradient at fixed parameters.) */
/* CPU-B with a persistent pool: `reps` batches in ONE call -- the threads are created once, every batch is
 * {zero own sums, own samples} | barrier | {add up own slice over all threads, in thread order} | barrier.  What the
 * all-cores number should measure is the cores, not pthread_create (256 creations cost more than a thread's four
 * samples) -- and not 256 private 1.6 MB sums either: the caller picks a thread count with >= 8 samples per thread. */
typedef struct { GradJob* j; int reps; } PoolArg;
static void* pool_worker(void* arg) {
  PoolArg* pa = (PoolArg*)arg;
  GradJob* j = pa->j;
  GradShared* sh = j->sh;
  Work w = work_new(sh->i, sh->h, sh->o);
  for (int rep = 0; rep < pa->reps; ++rep) {
    j->loss = 0.0;
    memset(j->gW1, 0, sizeof(double) * (size_t)sh->h * sh->i); memset(j->gb1, 0, sizeof(double) * sh->h);
    memset(j->gW2, 0, sizeof(double) * (size_t)sh->o * sh->h); memset(j->gb2, 0, sizeof(double) * sh->o);
    for (int b = j->b0; b < j->b1; ++b)
      j->loss += netgrad_mnist(&w, sh->X + (size_t)b * sh->i, sh->Y + (size_t)b * sh->o, sh->W1, sh->b1, sh->W2, sh->b2,
                               j->gW1, j->gb1, j->gW2, j->gb2, sh->recompute, 1);
    pthread_barrier_wait(&sh->bar);
    reduce_slice(sh, j->t, (size_t)sh->h * sh->i, sh->oW1, 0);
    reduce_slice(sh, j->t, (size_t)sh->h, sh->ob1, 1);
    reduce_slice(sh, j->t, (size_t)sh->o * sh->h, sh->oW2, 2);
    reduce_slice(sh, j->t, (size_t)sh->o, sh->ob2, 3);
    pthread_barrier_wait(&sh->bar);   /* nobody zeroes its sums while a peer still reads them */
  }
  work_free(&w);
  return NULL;
}
double hmat_batched_grads_pool(int reps, int B, int i, int h, int o, const double* X, const double* Y, const double* W1,
                               const double* b1, const double* W2, const double* b2, double* gW1, double* gb1,
                               double* gW2, double* gb2, int recompute, int threads) {
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  const size_t n1 = (size_t)h * i, n2 = (size_t)o * h, per = n1 + (size_t)h + n2 + (size_t)o;
  GradShared sh;
  sh.threads = threads; sh.i = i; sh.h = h; sh.o = o; sh.recompute = recompute;
  sh.X = X; sh.Y = Y; sh.W1 = W1; sh.b1 = b1; sh.W2 = W2; sh.b2 = b2;
  sh.oW1 = gW1; sh.ob1 = gb1; sh.oW2 = gW2; sh.ob2 = gb2;
  sh.jobs = calloc((size_t)threads, sizeof(GradJob));
  double* arena = malloc(per * (size_t)threads * sizeof(double));
  PoolArg* pa = calloc((size_t)threads, sizeof(PoolArg));
  pthread_t* tid = calloc((size_t)threads, sizeof(pthread_t));
  pthread_barrier_init(&sh.bar, NULL, (unsigned)threads);
  for (int t = 0; t < threads; ++t) {
    GradJob* j = &sh.jobs[t];
    j->t = t; j->sh = &sh;
    j->b0 = (int)((long)B * t / threads);
    j->b1 = (int)((long)B * (t + 1) / threads);
    double* base = arena + (size_t)t * per;
    j->gW1 = base; j->gb1 = base + n1; j->gW2 = base + n1 + h; j->gb2 = base + n1 + h + n2;
    pa[t].j = j; pa[t].reps = reps;
  }
  for (int t = 0; t < threads; ++t) pthread_create(&tid[t], NULL, pool_worker, &pa[t]);
  double loss = 0.0;
  for (int t = 0; t < threads; ++t) {
    pthread_join(tid[t], NULL);
    loss += sh.jobs[t].loss;
  }
  pthread_barrier_destroy(&sh.bar);
  free(sh.jobs); free(tid); free(pa); free(arena);
  return loss;
}

double hmat_batched_grads_mt(int B, int i, int h, int o, const double* X, const double* Y, const double* W1,
                             const double* b1, const double* W2, const double* b2, double* gW1, double* gb1,
                             double* gW2, double* gb2, int recompute, int threads) {
  if (threads < 1) threads = 1;
  if (threads > B) threads = B;
  const size_t n1 = (size_t)h * i, n2 = (size_t)o * h;
  GradShared sh;
  sh.threads = threads; sh.i = i; sh.h = h; sh.o = o; sh.recompute = recompute;
  sh.X = X; sh.Y = Y; sh.W1 = W1; sh.b1 = b1; sh.W2 = W2; sh.b2 = b2;
  sh.oW1 = gW1; sh.ob1 = gb1; sh.oW2 = gW2; sh.ob2 = gb2;
  sh.jobs = calloc((size_t)threads, sizeof(GradJob));
  static double* arena = NULL;
  static size_t arena_n = 0;
  const size_t per = n1 + (size_t)h + n2 + (size_t)o;
  if (arena_n < per * (size_t)threads) {
    free(arena);
    arena_n = per * (size_t)threads;
    arena = malloc(arena_n * sizeof(double));
  }
  pthread_barrier_init(&sh.bar, NULL, (unsigned)threads);
  pthread_t* tid = calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    GradJob* j = &sh.jobs[t];
    j->t = t; j->sh = &sh;
    j->b0 = (int)((long)B * t / threads);
    j->b1 = (int)((long)B * (t + 1) / threads);
    /* the per-thread sums live in one arena that is kept between calls (first-touch page faults of 1.6 MB per
     * thread would otherwise be most of a call); every thread zeroes its own part */
    double* base = arena + (size_t)t * per;
    j->gW1 = base; j->gb1 = base + n1; j->gW2 = base + n1 + h; j->gb2 = base + n1 + h + n2;
  }
  for (int t = 0; t < threads; ++t) pthread_create(&tid[t], NULL, grad_worker, &sh.jobs[t]);
  double loss = 0.0;
  for (int t = 0; t < threads; ++t) {
    pthread_join(tid[t], NULL);
    loss += sh.jobs[t].loss;
  }
  pthread_barrier_destroy(&sh.bar);
  free(sh.jobs); free(tid);
  return loss;
}

/* CPU-D: `cmap logistic` (HMat.hs:120-122) over fp32 -- the scalar loop, one libm call per element, on `threads`
 * contiguous slices */
typedef struct { long k0, k1; const float* x; float* y; } MapJob;
static void* map_worker(void* arg) {
  MapJob* j = (MapJob*)arg;
  for (long k = j->k0; k < j->k1; ++k) j->y[k] = 1.0f / (1.0f + expf(-j->x[k]));
  return NULL;
}
void hmat_map_logistic_f32_mt(long n, const float* x, float* y, int threads) {
  if (threads < 1) threads = 1;
  MapJob* jobs = calloc((size_t)threads, sizeof(MapJob));
  pthread_t* tid = calloc((size_t)threads, sizeof(pthread_t));
  for (int t = 0; t < threads; ++t) {
    jobs[t].k0 = n * t / threads; jobs[t].k1 = n * (t + 1) / threads; jobs[t].x = x; jobs[t].y = y;
    pthread_create(&tid[t], NULL, map_worker, &jobs[t]);
  }
  for (int t = 0; t < threads; ++t) pthread_join(tid[t], NULL);
  free(jobs); free(tid);
}
