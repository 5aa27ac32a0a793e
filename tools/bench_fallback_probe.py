"""bench.py with the C-ABI RCCL set-up made to fail: the run must fall back to torch.distributed's group (GPU box)."""
import sys, os
sys.path.insert(0, os.getcwd())
import tensor_ops_amd.dist as d
def boom(rank, world): raise RuntimeError("simulated RCCL set-up failure")
d.init_direct_comm = boom
sys.argv = ["bench.py", "--steps", "30", "--warmup", "5", "--force-dist"]
import runpy
runpy.run_path("bench.py", run_name="__main__")
