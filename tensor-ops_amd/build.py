"""Build libtensorops_hip.so (gfx950) in-tree with hipcc.  No CPU fallback is built."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtensorops_hip.so")
HOST_LIB = os.path.join(HERE, "libtensorops_host.so")
HOST_DIR = os.path.join(HERE, "host")
DOTS_BIN = os.path.join(HERE, "tensor-ops-dots-hip")
MNIST_BIN = os.path.join(HERE, "tensor-ops-mnist-hip")
SOURCES = ["runtime.cpp", "expr.cpp", "expr_jit.cpp", "rowprog.cpp", "api.cpp", "lazy.cpp", "comm.cpp", "gemm_f32_mfma.hip", "gemm_small.hip", "gemm_t32.hip", "gemm_skinnyk.hip", "gemm_kwave.hip", "gemm_kw16.hip", "gemm_kwave_f64.hip", "gemm_skinnyk_f64.hip", "gemm_f64.hip", "ewise.hip", "reduce_layout.hip", "gemv.hip", "fused_fflayer.hip", "p2p.hip", "online_sgd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]
AB = bool(os.environ.get("TOPS_BUILD_AB"))
if AB:   # development build: the A/B knobs and the extra tile-shape variants are compiled in.  Its objects and its library
    # live in build_ab/ (as tools/build_ab_lib.py's do): toggling the variable can never leave product and development
    # objects mixed in one directory or a development library under the product's name.  Load it with
    # TOPS_HIP_LIB=.../build_ab/libtensorops_hip.so LD_LIBRARY_PATH=.../build_ab
    FLAGS += ["-DTOPS_AB_KNOBS", "-DTOPS_GEMM_AB_VARIANTS"]
    LIB = os.path.join(HERE, "build_ab", "libtensorops_hip.so")
# The kernel files whose hand-written waits, barriers and wait states tools/asm_inflight_check.py proves on the GENERATED
# code (tests/test_pinned_asm.py): their device assembly -- the very text the object was assembled from -- is kept
# beside the object (build/<stem>-hip-amdgcn-amd-amdhsa-gfx950.s; -save-temps, everything else it leaves is deleted).
ASM_CHECKED = ["gemm_f32_mfma.hip", "gemm_f64.hip", "gemm_kwave.hip", "gemm_kw16.hip", "gemm_kwave_f64.hip", "gemm_skinnyk.hip",
               "gemm_skinnyk_f64.hip", "gemm_small.hip", "online_sgd.hip"]


def device_asm(src):
    """where the device assembly of csrc/<src> is kept by the build"""
    return os.path.join(HERE, "build", src.rsplit(".", 1)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _walk(d):
    return [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs]


def needs_build():
    arts = (LIB,) if AB else (LIB, HOST_LIB, DOTS_BIN, MNIST_BIN)
    if not all(os.path.exists(f) for f in arts):
        return True
    t = min(os.path.getmtime(f) for f in arts)
    deps = _walk(CSRC) + _walk(HOST_DIR) + [os.path.join(HERE, "..", "include", "tensorops_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Several processes may arrive here at once (parallel test loops on one box, pytest-xdist workers): one of them
    builds, under a lock; the others wait and find the libraries up to date.  Every artefact is written beside its final
    name and renamed into place, so a process that has the library mapped never sees it change underneath."""
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(HERE, "build_ab" if AB else "build")
    os.makedirs(objdir, exist_ok=True)
    import fcntl
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(force, verbose, objdir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _link(cmd, out):
    tmp = out + ".tmp.%d" % os.getpid()
    subprocess.check_call([c if c != out else tmp for c in cmd])
    os.replace(tmp, out)


def _build_locked(force, verbose, objdir):
    hipcc = _hipcc()
    procs = []
    objs = []
    # an object is rebuilt when its source, any header of csrc/ or the public header is newer (or on --force)
    hdrs = [f for f in _walk(CSRC) if f.endswith((".hpp", ".h"))] + [os.path.join(HERE, "..", "include", "tensorops_hip.h"), __file__]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        spath = os.path.join(CSRC, src)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(spath)):
            continue
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", spath, "-o", obj] + (["-save-temps=obj"] if src in ASM_CHECKED else [])
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    # the C++ host mirror compiles in parallel with the kernels
    host_obj = os.path.join(objdir, "toh_api.cpp.o")
    # plain C++: the host mirror talks to the GPU only through the C ABI
    host_cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-c",
                os.path.join(HOST_DIR, "toh_api.cpp"), "-o", host_obj]
    procs.append((host_cmd, subprocess.Popen(host_cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if verbose and out.strip():
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + out.decode())
    for f in os.listdir(objdir):   # -save-temps: keep the device assembly only
        stem = f.split("-hip-amdgcn")[0].split("-host-x86_64")[0]
        stem = stem[:-4] if stem.endswith(".hip") else stem
        if stem + ".hip" in ASM_CHECKED and f != stem + ".hip.o" and f != os.path.basename(device_asm(stem + ".hip")):
            os.remove(os.path.join(objdir, f))
    _link([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs +
          ["-L/opt/rocm/lib", "-lhiprtc", "-ldl", "-Wl,-rpath,/opt/rocm/lib"], LIB)
    if AB:   # (the host mirror and the apps belong to the product build: they resolve libtensorops_hip.so by LD_LIBRARY_PATH)
        return LIB
    _link(["g++", "-shared", "-fPIC", "-o", HOST_LIB, host_obj, "-L" + HERE, "-ltensorops_hip", "-Wl,-rpath,$ORIGIN"], HOST_LIB)
    # the Dots app on the HIP backend (host/apps/dots.cpp)
    _link(["g++", "-O2", "-std=c++17", "-Wall", "-o", DOTS_BIN, os.path.join(HOST_DIR, "apps", "dots.cpp"), "-L" + HERE,
           "-ltensorops_hip", "-Wl,-rpath,$ORIGIN"], DOTS_BIN)
    # tensor-ops-mnist on the HIP backend (host/apps/mnist.cpp)
    _link(["g++", "-O2", "-std=c++17", "-Wall", "-o", MNIST_BIN, os.path.join(HOST_DIR, "apps", "mnist.cpp"), "-L" + HERE,
           "-ltensorops_hip", "-Wl,-rpath,$ORIGIN"], MNIST_BIN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
