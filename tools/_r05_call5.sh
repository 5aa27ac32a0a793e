export PYTHONPATH=$PWD
O=gpurun_out/r05_c5
mkdir -p $O
timeout 300 python tools/t32_check.py --time > $O/t32_check.txt 2>&1; tail -4 $O/t32_check.txt
timeout 200 python tools/step_bench.py 400 > $O/step_bench.txt 2>&1; tail -1 $O/step_bench.txt
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_step -o step -- python $GRAFT_REPO_ROOT/tools/step_bench.py 400 > $GRAFT_REPO_ROOT/$O/prof_step.log 2>&1; cp $(find /tmp/rp_step -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/$O/step_kernel_stats.csv)
head -4 $O/step_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do n=$(echo $ctr | cut -d' ' -f1); (cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/rp_pmc && timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/rp_pmc -o pmc -- python $GRAFT_REPO_ROOT/tools/step_bench.py 20 > $GRAFT_REPO_ROOT/$O/pmc_$n.log 2>&1; cp $(find /tmp/rp_pmc -name "*counter_collection.csv" | head -1) $GRAFT_REPO_ROOT/$O/pmc_step_$n.csv); done
python - <<'PY'
import csv, collections
for n in ("FETCH_SIZE","WRITE_SIZE","TCC_HIT_sum"):
    try:
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open("gpurun_out/r05_c5/pmc_step_%s.csv"%n)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            if "t32" in k or "gemm_small" in k:
                print(k, {c: round(sum(x)/len(x)/ (1 if 'TCC' in c else 1),1) for c,x in v.items()}, "launches", len(list(v.values())[0]))
    except Exception as e: print(n, e)
PY
timeout 600 python tools/build_ab_lib.py gemm_t32.hip > $O/build_ab.log 2>&1
D=$PWD/tensor-ops_amd/build_ab; TOPS_HIP_LIB=$D/libtensorops_hip.so LD_LIBRARY_PATH=$D TOPS_T32_STAMPS=1 timeout 200 python tools/step_bench.py 400 > $O/step_stamps.txt 2>&1; tail -3 $O/step_stamps.txt
