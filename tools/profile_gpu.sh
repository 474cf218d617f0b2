#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile_gpu.sh <workload> <tag>
set -u
WL=${1:-c3}; TAG=${2:-r01}
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
CMD="python $ROOT/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-gpu-reference --sustain 0"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD < /dev/null > $OUT/trace.log 2>&1
echo "trace rc=$?" >> $OUT/trace.log
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C -d $OUT/pmc_$N -o pmc -- $CMD < /dev/null > $OUT/pmc_$N.log 2>&1
  echo "pmc $N rc=$?" >> $OUT/pmc_$N.log
done
cd $ROOT
timeout 120 python tools/summarize_prof.py $OUT < /dev/null > $OUT/summary.txt 2>&1
python -c "from allegro_amd.build import source_hash; print('kernel source hash:', source_hash())" >> $OUT/summary.txt
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2000k -delete   # keep gpurun_out small (raw traces are not needed)
