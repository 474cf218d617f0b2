# channel padding (u = 32 -> 64): BASELINE config 0 model on 64 atoms and on the 23^3 box, padded vs narrow kernels
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r02_v18_channel_padding.log
for n in 2 11 23; do
  for v in "pad+auto" "pad,AA_FUSED=0" "AA_NO_PAD=1"; do
    unset AA_FUSED AA_NO_PAD
    if [ "$v" = "pad,AA_FUSED=0" ]; then export AA_FUSED=0; fi
    if [ "$v" = "AA_NO_PAD=1" ]; then export AA_NO_PAD=1; fi
    ms=$(AA_BENCH_CELLS=$n timeout 600 python bench.py --workload c1 --steps 20 --warmup 5 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "c1 model (u=32, l_max=1) cells=$n atoms=$((8*n*n*n)) [$v] $ms" >> gpurun_out/r02_v18_channel_padding.log
  done
done
unset AA_FUSED AA_NO_PAD
timeout 600 python -m pytest tests/test_hip_model.py -m gpu -q -k "c1" 2>&1 | tail -2
cat gpurun_out/r02_v18_channel_padding.log
