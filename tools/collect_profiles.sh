#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box through gpurun):
#   tools/collect_profiles.sh rNN      -> gpurun_out/prof_rNN/*.csv|json
# Kernel-trace/stats passes and the PMC passes are separate runs; PMC passes use --kernel-trace only.
set -u
TAG=${1:-r02}
ONLY=${2:-all}   # optional: run a single leg (bench|gemm4096|step_fused|step_generic|pmc)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONPATH=$REPO
cd /tmp
stats() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/rp_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -o $name -- "$@" > $OUT/${name}.log 2>&1
  find /tmp/rp_$name -name "*kernel_stats.csv" -exec cp {} $OUT/${TAG}_${name}_kernel_stats.csv \;
}
pmc() {  # name, counters, command...
  local name=$1 ctr=$2; shift 2
  rm -rf /tmp/rp_$name
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/rp_$name -o $name -- "$@" > $OUT/${name}.log 2>&1
  find /tmp/rp_$name -name "*counter_collection.csv" -exec cp {} $OUT/${TAG}_${name}_counter_collection.csv \;
}
python $REPO/bench.py --steps 500 --warmup 50 > $OUT/${TAG}_bench_unprofiled.json 2> $OUT/bench_unprofiled.err
stats bench python $REPO/bench.py --steps 500 --warmup 50
grep '^{' $OUT/bench.log > $OUT/${TAG}_bench_under_rocprof.json
WARM=30 stats gemm4096 python $REPO/tools/gemm_bench.py 4096 4096 4096 50
WARM=50 stats gemm1024 python $REPO/tools/gemm_bench.py 1024 1024 1024 300   # the wave-split kernel (gemm_kwave.hip)
WARM=50 stats gemm1536 python $REPO/tools/gemm_bench.py 1536 1536 1536 200   # ... its 96x96 tiles
WARM=50 stats gemm1000 python $REPO/tools/gemm_bench.py 1000 1000 1000 300   # ... a K tail inside the kernel
pmc pmc_gemm1024_fetch FETCH_SIZE python $REPO/tools/gemm_bench.py 1024 1024 1024 5
pmc pmc_gemm1024_write WRITE_SIZE python $REPO/tools/gemm_bench.py 1024 1024 1024 5
pmc pmc_gemm1024_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" python $REPO/tools/gemm_bench.py 1024 1024 1024 20
stats step python $REPO/tools/step_bench.py 500                       # tag-less Network, gradTOp stream fused by the library, replayed
stats step_two_call python $REPO/tools/step_bench.py 500 --two-call   # grad() + apply(): what a data-parallel rank runs
stats step_fusion_off python $REPO/tools/step_bench.py 500 --generic  # the same stream, one launch per class-method call
ITERS=50 WARM=20 stats c5 python $REPO/tools/c5_bench.py
stats c5_f64 python $REPO/tools/c5_f64_probe.py                      # config 5 in Double (gemm_skinnyk_f64.hip), plain and with the map fused
pmc pmc_gemm_fetch FETCH_SIZE python $REPO/tools/gemm_bench.py 4096 4096 4096 5
pmc pmc_gemm_write WRITE_SIZE python $REPO/tools/gemm_bench.py 4096 4096 4096 5
pmc pmc_map_fetch FETCH_SIZE python $REPO/tools/map_bench.py 5
pmc pmc_map_write WRITE_SIZE python $REPO/tools/map_bench.py 5
ITERS=5 WARM=2 pmc pmc_c5_fetch FETCH_SIZE python $REPO/tools/c5_bench.py
ITERS=5 WARM=2 pmc pmc_c5_write WRITE_SIZE python $REPO/tools/c5_bench.py
ITERS=20 WARM=10 pmc pmc_c5_sq "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" python $REPO/tools/c5_bench.py
# (round 4: the version-1 kernel is an A/B knob of development builds; its r02/r03 passes are kept in profiles/)
pmc pmc_step_fetch FETCH_SIZE python $REPO/tools/step_bench.py 20
pmc pmc_step_write WRITE_SIZE python $REPO/tools/step_bench.py 20
pmc pmc_gemm_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" python $REPO/tools/gemm_bench.py 4096 4096 4096 5
WARM=50 stats gemm640 python $REPO/tools/gemm_bench.py 640 640 640 300        # (round 6: 48x48 tiles of gemm_kw16.hip; before: two workgroups per tile)
WARM=50 stats gemm768 python $REPO/tools/gemm_bench.py 768 768 768 300        # round 6: 256 tiles of 48x48 (gemm_kw16_kernel<.,.,3,3>)
WARM=50 stats gemm1280 python $REPO/tools/gemm_bench.py 1280 1280 1280 200    # round 6: 256 tiles of 80x80 (gemm_kw16_kernel<.,.,5,5>)
WARM=50 stats gemm1088 python $REPO/tools/gemm_bench.py 1088 1088 1088 200    # round 6: stream-K over 512 workgroups (gemm_kw_kernel<...,0>)
WARM=50 stats gemm512x2048 python $REPO/tools/gemm_bench.py 512 2048 512 300   # few tiles, long K: four workgroups per tile
WARM=50 stats gemm60000x784x300 python $REPO/tools/gemm_bench.py 60000 784 300 100   # round 6 (last): a tall batch through a narrow layer, a tile per wave (gemm_kw_kernel<...,false,1>)
WARM=50 stats gemm100x60000x300 python $REPO/tools/gemm_bench.py 100 60000 300 200   # ... a narrow layer's weight gradient: stream-K over 256 workgroups
WARM=20 stats matvec16384 python $REPO/tools/gemv_bench.py matVec 16384 16384 100    # gemv.hip: a row per output
WARM=20 stats vecmat16384 python $REPO/tools/gemv_bench.py vecMat 16384 16384 100    # ... outputs contiguous (+ the finishing pass)
WARM=20 stats outer16384 python $REPO/tools/gemv_bench.py outerV 16384 16384 100
WARM=2 pmc pmc_matvec_fetch FETCH_SIZE python $REPO/tools/gemv_bench.py matVec 16384 16384 5
WARM=2 pmc pmc_vecmat_fetch FETCH_SIZE python $REPO/tools/gemv_bench.py vecMat 16384 16384 5
WARM=2 pmc pmc_outer_write WRITE_SIZE python $REPO/tools/gemv_bench.py outerV 16384 16384 5
python $REPO/tools/gemm_sweep.py 2>/dev/null | grep " x " > $OUT/${TAG}_gemm_sweep.txt
# few tiles / long K and the fp64 mid sizes (ours only; steady state)
python $REPO/tools/gemm_ab.py 640 640 640 704 704 704 768 768 768 832 832 832 1024 1024 512 512 2048 512 384 4096 384 256 4096 1024 768 4096 768 1088 1088 1088 1152 1152 1152 1280 1280 1280 1472 1472 1472 1792 1792 1792 768 1024 1024 1152 2048 1152 1280 4096 1280 2>/dev/null | grep " x " > $OUT/${TAG}_gemm_few_tiles.txt
GEMM_DTYPE=f64 python $REPO/tools/gemm_ab.py 768 768 768 1000 1000 1000 1024 1024 1024 1280 1280 1280 1536 1536 1536 2048 2048 2048 4096 784 256 1024 4096 1024 1100 528 900 4096 4096 4096 2>/dev/null | grep " x " > $OUT/${TAG}_gemm_f64_mid.txt
ls -la $OUT
