# PMC counters of the fused forward kernel (AA_FUSED mode $1, l_max override $2)
cd /root/repo
MODE=${1:-2}; LM=${2:-1}; TAG=${3:-pmcf}
export TMPDIR=/tmp AA_FUSED=$MODE AA_BENCH_LMAX=$LM
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p $OUT
CMD="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-gpu-reference --sustain 0"
cd /tmp
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $C -d $OUT/pmc_$N -o pmc -- $CMD < /dev/null > $OUT/pmc_$N.log 2>&1
done
cd $ROOT
timeout 120 python tools/summarize_prof.py $OUT < /dev/null 2>&1 | grep -E "pass|fused" > $OUT/summary_fused.txt
cat $OUT/summary_fused.txt
rm -rf $OUT/pmc_*
