cd /tmp; export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
for n in 1024 1536 2048; do
rm -rf /tmp/rp_$n; WARM=20 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$n -o g -- python $GRAFT_REPO_ROOT/tools/gemm_bench.py $n $n $n 50 > /dev/null 2>&1
echo "== $n"; find /tmp/rp_$n -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-160 | head -5
done
