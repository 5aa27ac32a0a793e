export PYTHONPATH=$PWD
O=gpurun_out/r05_c7
mkdir -p $O
timeout 600 python tools/build_ab_lib.py gemm_t32.hip > $O/build_ab.log 2>&1
D=$PWD/tensor-ops_amd/build_ab
for pd in 4 3 2; do
  echo "== PD $pd"
  TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_T32_PD=$pd TOPS_T32_STAMPS=1 timeout 200 python tools/step_bench.py 400 2>&1 | tail -3
done
echo "== t32 off"
TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_GEMM_T32=0 timeout 200 python tools/step_bench.py 400 2>&1 | tail -1
echo "== product"
timeout 200 python tools/step_bench.py 400 2>&1 | tail -1
timeout 200 python bench.py --no-aux --steps 20 --warmup 5 > $O/bench_noaux.json 2> $O/bench_noaux.err; python -c "
import json;d=json.load(open('$O/bench_noaux.json'));print(d['value'],d['ms_per_step'])"
timeout 300 python tools/t32_check.py --time > $O/t32_check.txt 2>&1; tail -4 $O/t32_check.txt
