cd /root/repo
export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_c5b
mkdir -p $OUT
( cd /tmp && timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python /root/repo/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-profile < /dev/null > $OUT/trace.log 2>&1 )
timeout 30 python tools/summarize_prof.py $OUT < /dev/null > gpurun_out/r01_v31_rocprofv3_c5_summary.txt 2>&1
rm -rf $OUT
head -32 gpurun_out/r01_v31_rocprofv3_c5_summary.txt
