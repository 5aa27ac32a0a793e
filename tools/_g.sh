for s in 1024 1536 2048 1000 1280 1792 2000; do WARM=20 timeout 100 python tools/gemm_bench.py $s $s $s 50 2>&1 | grep "gemm "; done
timeout 600 python -m pytest tests/test_gpu_full_size.py -q -x 2>&1 | tail -2
