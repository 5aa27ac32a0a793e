"""A SECOND, independently written checker for the config-3 gradient: textbook closed-form backprop in numpy
fp64, sharing no code with `oracle/` (which restates the reference's op-by-op composition).  Two restatements that
were written separately have to agree before either is trusted as the yardstick for the HIP path; the reference
itself holds no vectors (`test/Spec.hs:1-2`), so this is the strongest pin available (parity stays "partial").

  a1 = sigma(X W1^T + b1) ; z2 = a1 W2^T + b2 ; p = softmax(z2) ; L = -sum_b <y_b, log p_b>
  dz2 = p * sum(y) - y ; dW2 = dz2^T a1 ; db2 = sum_b dz2 ; dz1 = (dz2 W2) * a1 (1 - a1) ; dW1 = dz1^T X ; db1 = sum_b dz1
(`sum(y)` because the reference differentiates `-<y, log softmax z>` for ANY y, not just one-hot rows.)"""
import numpy as np


def softmax_ce_grads(X, Y, W1, b1, W2, b2):
    X, Y, W1, b1, W2, b2 = (np.asarray(a, dtype=np.float64) for a in (X, Y, W1, b1, W2, b2))
    a1 = 1.0 / (1.0 + np.exp(-(X @ W1.T + b1)))
    z2 = a1 @ W2.T + b2
    z2 = z2 - z2.max(axis=1, keepdims=True)
    e = np.exp(z2)
    p = e / e.sum(axis=1, keepdims=True)
    loss = -(Y * np.log(p)).sum()
    dz2 = p * Y.sum(axis=1, keepdims=True) - Y
    dz1 = (dz2 @ W2) * a1 * (1.0 - a1)
    return [dz1.T @ X, dz1.sum(axis=0), dz2.T @ a1, dz2.sum(axis=0)], loss


def logistic_se_grads(X, Y, W1, b1, W2, b2):
    X, Y, W1, b1, W2, b2 = (np.asarray(a, dtype=np.float64) for a in (X, Y, W1, b1, W2, b2))
    a1 = 1.0 / (1.0 + np.exp(-(X @ W1.T + b1)))
    s = 1.0 / (1.0 + np.exp(-(a1 @ W2.T + b2)))
    loss = ((Y - s) ** 2).sum()
    dz2 = -2.0 * (Y - s) * s * (1.0 - s)
    dz1 = (dz2 @ W2) * a1 * (1.0 - a1)
    return [dz1.T @ X, dz1.sum(axis=0), dz2.T @ a1, dz2.sum(axis=0)], loss
