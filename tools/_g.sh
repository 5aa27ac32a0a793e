for w in 1 512 768 1024; do echo "W4_128=$w"; for s in 1024 1280 1536 1792 2048 2304; do TOPS_GEMM_W4_128=$w WARM=20 timeout 100 python tools/gemm_bench.py $s $s $s 50 2>&1 | grep "gemm "; done; done
