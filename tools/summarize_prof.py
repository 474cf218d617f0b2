"""Summarise rocprofv3 (rocpd sqlite) outputs of tools/profile_gpu.sh: per-kernel time from the kernel
trace, per-kernel PMC means from the counter passes."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]
KEYS = ("gemm_chain_bf16x3", "gemm_bf16x3_kernel<0", "gemm_bf16x3_kernel<2", "gemm_bf16x3_kernel<4", "gemm_mfma_f32_v3_kernel<0>", "gemm_mfma_f32_v3_kernel<2>", "gemm_mfma_f32_v3_kernel<4>", "gemm_mfma_f32_kernel",
        "gemm_valu", "tp_mom_fwd_first", "tp_mom_fwd_last", "tp_mom_bwd_last", "tp_mom_bwd_first", "tp_chain_fwd_last", "tp_chain_bwd_last", "tp_chain_bwd_first", "tp_spec_fwd", "tp_spec_bwd",
        "tp_layer_fwd", "tp_layer_bwd", "edge_prologue", "edge_backward", "readout_reduce", "readout_backward",
        "tp_op_moments", "tp_op_edge_fwd", "tp_op_edge_bwd", "tp_op_edge_env", "tp_op_bvecs", "tp_op_bwd_mid", "tp_op_fwd",
        "tp_op_bwd", "gemm_mfma_f64_pipe", "gemm_mfma_f64", "force_gather", "virial_partial", "virial_final", "nl_pairs",
        "nl_bin", "nl_scan", "nl_cell", "fused_")


def short(name):
    for key in KEYS:
        if key in name:
            return key
    return name[:48]


for f in sorted(glob.glob(os.path.join(out, "trace", "*.db"))):
    db = sqlite3.connect(f)
    agg = defaultdict(lambda: [0, 0.0])
    for name, start, end in db.execute("select name, start, end from kernels"):
        k = short(name)
        agg[k][0] += 1
        agg[k][1] += (end - start) * 1e-3
    tot = sum(v[1] for v in agg.values())
    print("== rocprofv3 --kernel-trace: all launches of the profiled run (2 warm-up + 5 timed steps + setup)")
    print(f"{'kernel':30s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'%':>6s}")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:30s} {n:7d} {t:12.1f} {t / n:10.1f} {100 * t / tot:6.1f}")

for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "*.db")):
        db = sqlite3.connect(f)
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for name, cname, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
            a = agg[short(name)][cname]
            a[0] += 1
            a[1] += value
        print(f"== rocprofv3 --pmc pass {os.path.basename(d)[4:]}: per kernel, mean per launch")
        for k in sorted(agg):
            line = "  ".join(f"{c}={v[1] / v[0]:.1f}" for c, v in sorted(agg[k].items()))
            n = max(v[0] for v in agg[k].values())
            print(f"{k:30s} launches={n:4d}  {line}")
