"""Extract per-kernel HBM traffic from a tools/summarize_prof.py summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
passes) into the small JSON that bench.py attaches to its roofline as `traffic`.

    python tools/pmc_to_json.py profiles/archive/r01_v13_rocprofv3_c4_summary.txt c4 > profiles/pmc_traffic_c4.json

Units and corrections (MI355X_MICROARCH.md, HBM section): rocprofv3 prints both counters in KB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read and is doubled; WRITE_SIZE is taken as is (the
guide calls it uncalibrated -- for these kernels it matches the algorithmic written bytes within 3 %).
"""
import json
import re
import sys

path, workload = sys.argv[1], sys.argv[2]
sec = None
out = {}
for line in open(path):
    m = re.match(r"== rocprofv3 --pmc pass (\w+)", line)
    if m:
        sec = m.group(1)
        continue
    if line.startswith("=="):
        sec = None
        continue
    if sec in ("FETCH_SIZE", "WRITE_SIZE"):
        m = re.match(r"(\S.*?)\s+launches=\s*(\d+)\s+" + sec + r"=([\d.]+)", line)
        if m:
            k = out.setdefault(m.group(1).strip(), {})
            k["launches"] = int(m.group(2))
            k["fetch_bytes" if sec == "FETCH_SIZE" else "write_bytes"] = float(m.group(3)) * 1e3 * (2.0 if sec == "FETCH_SIZE" else 1.0)
import os  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from allegro_amd.build import source_hash  # noqa: E402

# source_hash: the kernel sources the measurement was taken on (run this right after profiling, before editing csrc/);
# bench.py attaches `traffic` only when it matches the sources it runs
# `source` names the TRACKED copy of the summary (tools/gpu_round.sh writes under gpurun_out/, the builder copies it to profiles/)
print(json.dumps({"workload": workload, "source": "profiles/" + os.path.basename(path), "source_hash": sys.argv[3] if len(sys.argv) > 3 else source_hash(),
                  "per_launch": out}, indent=1))
