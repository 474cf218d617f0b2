#!/bin/bash
# Run ON THE GPU BOX (via gpurun): SQ counter passes of one kernel of the C4 / C3 / C5 step, per host environment arm.
#   tools/pmc_kernel.sh <tag> <workload> <kernel substring> [ENV=val ...]   (one arm per extra argument; "default" arm first)
# Output: gpurun_out/<tag>_pmc_<kernel>.txt -- per arm and counter the mean per launch.
set -u
TAG=$1; WL=$2; KER=$3; shift 3
export TMPDIR=/tmp
ROOT=$(pwd)
OUTF=$ROOT/gpurun_out/${TAG}_pmc_${KER}.txt
: > $OUTF
PASSES=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM"
        "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES SQ_INST_CYCLES_SALU"
        "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES")
if [ -n "${PMC_ONLY_LAST:-}" ]; then PASSES=("${PASSES[3]}"); fi
for arm in default "$@"; do
  if [ "$arm" = default ]; then pre=""; else pre="$arm"; fi
  echo "== arm [$arm]" >> $OUTF
  i=0
  for C in "${PASSES[@]}"; do
    i=$((i+1))
    D=/tmp/pmck_${TAG}_$i
    rm -rf $D
    (cd /tmp && env $pre timeout 600 rocprofv3 --pmc $C -d $D -o pmc -- python $ROOT/bench.py --workload $WL --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-gpu-reference --no-secondary --no-md --sustain 0 < /dev/null > /tmp/pmck.log 2>&1)
    python - "$D" "$KER" >> $OUTF <<'PY'
import glob, os, sqlite3, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True):
    db = sqlite3.connect(f)
    for name, cname, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if sys.argv[2] in name:
            agg[cname][0] += 1
            agg[cname][1] += value
for c, (n, v) in sorted(agg.items()):
    print(f"  {c:32s} {v / n:18.1f}   (launches {n})")
PY
  done
done
cat $OUTF
