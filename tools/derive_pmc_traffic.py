"""profiles/pmc_traffic.json from the committed rocprofv3 PMC passes: HBM bytes per launch of a kernel =
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 averaged over its launches (both counters are reported in KiB; the factor 2 on FETCH_SIZE
is the gfx950 correction MI355X_MICROARCH.md prescribes).  usage: derive_pmc_traffic.py rNN"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"


def per_launch(name, counter, kernel_substr):
    path = os.path.join(ROOT, "profiles", f"{tag}_pmc_{name}_counter_collection.csv")
    vals = {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter or kernel_substr not in row["Kernel_Name"]:
            continue
        vals.setdefault(row["Dispatch_Id"], 0.0)
        vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
    assert vals, (path, counter, kernel_substr)
    return sum(vals.values()) / len(vals)


def traffic(fetch_pass, write_pass, kernel_substr):
    return int(round((2 * per_launch(fetch_pass, "FETCH_SIZE", kernel_substr) + per_launch(write_pass, "WRITE_SIZE", kernel_substr)) * 1024))


out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); bytes = (2*FETCH_SIZE + "
               "WRITE_SIZE) * 1024 averaged over the launches of the kernel; tools/derive_pmc_traffic.py; see profiles/README.md",
       "_round": tag,
       "gmul_4096": traffic("gemm_fetch", "gemm_write", "gemm_mfma_kernel"),
       "map_logistic_512cubed": traffic("map_fetch", "map_write", "ew_stream_kernel"),
       "gmul_c5a": traffic("c5_fetch", "c5_write", "64, true"),
       "gmul_map_c5_fused": traffic("c5_fetch", "c5_write", "64, false"),
       "gmul_1024_wave_split": traffic("gemm1024_fetch", "gemm1024_write", "gemm_kw_kernel")}
# the HBM-bound forms of csrc/gemv.hip at 16384 x 16384 (1 GiB of matrix): the matrix is read (matVec, vecMat) or written (outerV) once; the
# other direction is a vector (64 KiB) and has no pass of its own
try:
    out["hbm_bound_forms"] = {
        "matVec_16384x16384": int(round(2 * per_launch("matvec_fetch", "FETCH_SIZE", "gemv_rows_kernel") * 1024)),
        "vecMat_16384x16384": int(round(2 * (per_launch("vecmat_fetch", "FETCH_SIZE", "gemv_cols_kernel") + per_launch("vecmat_fetch", "FETCH_SIZE", "gemv_finish_kernel")) * 1024)),
        "outerV_16384x16384": int(round(per_launch("outer_write", "WRITE_SIZE", "outer_kernel") * 1024))}
except (AssertionError, OSError) as e:
    out["hbm_bound_forms"] = "unavailable: %r" % (e,)
# the three launches of the config-3 step (round 5: the forward layer on gemm_t32_kernel); algorithmic bytes of a step: 4.9 MB
try:
    step = {"forward": traffic("step_fetch", "step_write", "gemm_t32_kernel"),
            "loss_head": traffic("step_fetch", "step_write", "gemm_small_kernel"),
            "weight_gradients": traffic("step_fetch", "step_write", "gemm_small_pair_kernel")}
    step["total"] = sum(step.values())
    out["step_c3"] = step
except (AssertionError, OSError) as e:   # (an older round's passes name other kernels)
    out["step_c3"] = "unavailable: %r" % (e,)
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
