# channel padding for 3-layer stacks (u = 32 -> 64 on the per-atom operator kernels): padded vs narrow kernels
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r02_v19_channel_padding_L3.log
for lm in 1 2; do
for n in 2 11 23; do
  for v in "pad" "AA_NO_PAD=1"; do
    unset AA_NO_PAD
    if [ "$v" = "AA_NO_PAD=1" ]; then export AA_NO_PAD=1; fi
    ms=$(AA_BENCH_LAYERS=3 AA_BENCH_LMAX=$lm AA_BENCH_CELLS=$n timeout 600 python bench.py --workload c1 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "u=32 L=3 l_max=$lm cells=$n atoms=$((8*n*n*n)) [$v] $ms" >> gpurun_out/r02_v19_channel_padding_L3.log
  done
done
done
cat gpurun_out/r02_v19_channel_padding_L3.log
