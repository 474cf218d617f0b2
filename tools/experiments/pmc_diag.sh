#!/bin/bash
# Run ON THE GPU BOX: extra PMC passes for diagnosing one workload.  Usage: tools/pmc_diag.sh <workload> <tag>
WL=${1:-c3}; TAG=${2:-diag}
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_${TAG}_${WL}; mkdir -p $OUT
CMD="python $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-profile"
cd /tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" \
         "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C -d $OUT/pmc_$i -o pmc -- $CMD > $OUT/pmc_$i.log 2>&1
  echo "pass $i rc=$? : $C" >> $OUT/pmc_$i.log
done
cd $ROOT
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
grep -v "amd_rocclr\|at::\|rocprim\|elementwise\|Cijk\|vectorized\|reduce_kernel" $OUT/summary.txt
