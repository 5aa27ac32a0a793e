"""A development library beside the product one, for A/B runs on ONE box: the named kernel files compiled with
-DTOPS_AB_KNOBS (their A/B knobs are read, common.hpp ab_getenv), every other object taken from the product build.

  python tools/build_ab_lib.py gemm_kwave.hip gemm_skinnyk.hip      # -> tensor-ops_amd/build_ab/libtensorops_hip.so
  D=$PWD/tensor-ops_amd/build_ab; TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_GEMM_KW_PAIR=1 python tools/gemm_ab.py 768 768 768

(TOPS_HIP_LIB is what the ctypes layer loads; LD_LIBRARY_PATH makes the host mirror, which names libtensorops_hip.so as a
dependency, resolve to the same file -- its RUNPATH is searched after LD_LIBRARY_PATH.)

(api.cpp is always one of the recompiled files, so that to_build_info reports a development build.)  The product library is
not touched.  Measurement tooling."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_build", os.path.join(ROOT, "tensor-ops_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def main():
    files = sorted(set(sys.argv[1:]) | {"api.cpp"})
    for f in files:
        assert f in B.SOURCES, f
    B.build()   # the product objects
    objdir = os.path.join(B.HERE, "build_ab")
    os.makedirs(objdir, exist_ok=True)
    hipcc = B._hipcc()
    flags = [f for f in B.FLAGS if not f.startswith("-DTOPS_")] + ["-DTOPS_AB_KNOBS"]
    hdr_t = max(os.path.getmtime(os.path.join(B.CSRC, h)) for h in os.listdir(B.CSRC) if h.endswith((".hpp", ".h")))
    procs = []
    for f in files:
        src, obj = os.path.join(B.CSRC, f), os.path.join(objdir, f + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(src)):
            continue
        cmd = [hipcc] + flags + ["-x", "hip", "-c", src, "-o", obj]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(" ".join(cmd) + "\n" + out.decode())
    objs = [os.path.join(objdir if s in files else os.path.join(B.HERE, "build"), s + ".o") for s in B.SOURCES]
    lib = os.path.join(objdir, "libtensorops_hip.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs +
                          ["-L/opt/rocm/lib", "-lhiprtc", "-ldl", "-Wl,-rpath,/opt/rocm/lib"])
    print(lib)


if __name__ == "__main__":
    main()
