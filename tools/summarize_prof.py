"""Summarise rocprofv3 outputs of tools/profile_gpu.sh: per-kernel time (kernel trace) and per-kernel PMC sums."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for key in ("gemm_mfma_f32_v3", "gemm_mfma_f32_kernel", "gemm_valu", "tp_chain_fwd_last", "tp_chain_bwd_last",
                "tp_chain_bwd_first", "tp_spec_fwd", "tp_spec_bwd", "tp_layer_fwd", "tp_layer_bwd", "edge_prologue",
                "edge_backward", "readout_reduce", "readout_backward"):
        if key in name:
            return key
    return name[:60]


# kernel trace
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    agg = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            agg[k][0] += 1
            agg[k][1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    tot = sum(v[1] for v in agg.values())
    print(f"== kernel trace ({os.path.basename(f)}), all launches of the profiled run (warmup + steps)")
    print(f"{'kernel':28s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:28s} {n:7d} {t:12.1f} {t / n:10.1f} {100 * t / tot:6.1f}")

# PMC passes
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(int)
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[(k, row["Counter_Name"])] += 1
        print(f"== PMC {os.path.basename(d)} (sum over launches / per launch)")
        for k in sorted(agg):
            for c, v in sorted(agg[k].items()):
                n = calls[(k, c)]
                print(f"{k:28s} {c:28s} sum={v:16.1f} launches={n:5d} per_launch={v / n:14.1f}")
