export PYTHONPATH=$PWD
O=gpurun_out/r05_c8
mkdir -p $O
timeout 600 python tools/build_ab_lib.py gemm_t32.hip > $O/build_ab.log 2>&1
D=$PWD/tensor-ops_amd/build_ab
SH="512 512 512 640 640 640 704 704 704 768 768 768 832 832 832 896 896 896 1024 1024 1024 1024 512 1024 1280 256 1280 2048 256 1024"
echo "== t32 window 4096 tiles"; TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_T32_MAXTILES=4096 timeout 300 python tools/gemm_ab.py $SH 2>&1 | grep " x "
echo "== default"; timeout 300 python tools/gemm_ab.py $SH 2>&1 | grep " x "
timeout 200 python tools/step_bench.py 400 2>&1 | tail -1
timeout 300 python tools/t32_check.py > $O/t32_check.txt 2>&1; tail -1 $O/t32_check.txt
