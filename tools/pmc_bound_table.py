"""Per-kernel "what bounds it" table from a tools/profile_gpu.sh summary (rocprofv3 kernel trace + the four PMC passes):

    python tools/pmc_bound_table.py profiles/r05_vNN_rocprofv3_c5_summary.txt [clock_GHz] > profiles/r05_vNN_bound_table_c5.md

Columns: average launch time; HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE, the guide's gfx950 correction) and the rate it
implies against the ~6.3 TB/s the guide calls achievable; vector-instruction ISSUE share = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x
kernel cycles) (a wave64 vector instruction occupies its SIMD for 4 cycles); WAIT share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (fraction
of resident-wave cycles spent waiting on anything); matrix-pipe share = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x kernel cycles); the last
column names the largest of them.  Kernel cycles are taken from the trace time at the given shader clock (default 2.1 GHz: what
rocm-smi shows under this load, profiles/r04_v29_power_clock_samples_c4.txt)."""
import re
import sys

path = sys.argv[1]
clock = float(sys.argv[2]) * 1e9 if len(sys.argv) > 2 else 2.1e9
sec, t_us, cnt = None, {}, {}
for line in open(path):
    if line.startswith("== rocprofv3 --kernel-trace"):
        sec = "trace"
        continue
    m = re.match(r"== rocprofv3 --pmc pass (\w+)", line)
    if m:
        sec = "pmc"
        continue
    if line.startswith("=="):
        sec = None
        continue
    if sec == "trace":
        f = line.split()
        if len(f) >= 5 and f[1].isdigit():
            t_us[f[0]] = float(f[3])
    elif sec == "pmc":
        m = re.match(r"(\S.*?)\s+launches=\s*(\d+)\s+(.*)", line)
        if m:
            d = cnt.setdefault(m.group(1).strip(), {})
            for kv in m.group(3).split():
                k, v = kv.split("=")
                d[k] = float(v)
print("| kernel | avg launch ms | HBM GB / launch | TB/s (of ~6.3) | VALU issue | waiting | matrix pipe | largest |")
print("|---|---|---|---|---|---|---|---|")
for k, us in sorted(t_us.items(), key=lambda kv: -kv[1]):
    c = cnt.get(k)
    if not c or "SQ_WAVE_CYCLES" not in c or us < 20:
        continue
    cyc = us * 1e-6 * clock
    gb = (c.get("FETCH_SIZE", 0.0) * 2 + c.get("WRITE_SIZE", 0.0)) * 1e3 / 1e9
    tbs = gb / (us * 1e-6) / 1e3
    valu = c.get("SQ_INSTS_VALU", 0.0) * 4 / (1024 * cyc)
    wait = c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1.0)
    mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc)
    shares = {"HBM": tbs / 6.3, "vector issue": valu, "matrix pipe": mfma}
    top = max(shares, key=shares.get)
    print(f"| {k} | {us / 1e3:.3f} | {gb:.2f} | {tbs:.2f} ({tbs / 6.3:.2f}) | {valu:.2f} | {wait:.2f} | {mfma:.2f} | {top} |")
