# A/B of the forward with in-chain tensor-track scalars (AA_CHAIN_TP=1): GPU tests, C4 / C3 stage tables with and without
cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_chain_tp.py -m gpu -q 2>&1 | tail -8 > gpurun_out/ctp_pytest.log
for v in on off; do
  if [ $v = on ]; then export AA_CHAIN_TP=1; else export AA_CHAIN_TP=0; fi
  timeout 300 python bench.py --steps 10 --warmup 3 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/ctp_bench_$v.log 2> gpurun_out/ctp_stages_$v.log
  timeout 300 python bench.py --workload c3 --steps 50 --warmup 5 --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/ctp_bench_c3_$v.log 2>&1
done
tail -4 gpurun_out/ctp_pytest.log
for v in on off; do echo $v; grep "stage" gpurun_out/ctp_stages_$v.log | head -9; grep -o '"ms_per_step": [0-9.]*' gpurun_out/ctp_bench_$v.log gpurun_out/ctp_bench_c3_$v.log; done
