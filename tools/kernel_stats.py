"""Top kernels of a `rocprofv3 --kernel-trace` output directory (rocpd sqlite) as a small text table: per-kernel total ms, share,
calls, average -- for runs whose kernels are not the library's own (the training step: torch / rocBLAS kernels next to ours).

    python tools/kernel_stats.py <rocprofv3 -d directory> [top N] > profiles/rNN_<what>_kernel_stats.txt
"""
import glob
import re
import sqlite3
import sys
from collections import defaultdict

d, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 45
files = glob.glob(d + "/**/*.db", recursive=True)
if not files:
    sys.exit(f"no rocpd database under {d}")
agg = defaultdict(lambda: [0, 0.0])
for f in files:
    db = sqlite3.connect(f)
    for name, start, end in db.execute("select name, start, end from kernels"):
        # (template arguments of torch's elementwise kernels are long: keep the functor name)
        k = re.sub(r"\s+", " ", name)
        m = re.search(r"(\w+_kernel\w*|\w+Functor\w*|\w+_impl\w*)", k)
        k = k[:170]
        agg[k][0] += 1
        agg[k][1] += (end - start) * 1e-6
tot = sum(v[1] for v in agg.values())
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t:9.2f} ms {100 * t / tot:5.1f}% calls={n:>6} avg={t / n * 1e3:9.1f} us  {k}")
print(f"total {tot:.1f} ms over {len(agg)} kernels, {sum(v[0] for v in agg.values())} launches")
