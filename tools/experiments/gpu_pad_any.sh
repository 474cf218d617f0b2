# channel padding for channel counts between the multiples of 64 (48 -> 64, 96 -> 128): padded vs narrow kernels, 10 648 atoms
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r02_v20_channel_padding_any.log
for u in 48 96; do
  for v in "pad" "AA_NO_PAD=1"; do
    unset AA_NO_PAD
    if [ "$v" = "AA_NO_PAD=1" ]; then export AA_NO_PAD=1; fi
    ms=$(AA_BENCH_U=$u AA_BENCH_CELLS=11 timeout 600 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "u=$u L=2 l_max=2 atoms=10648 [$v] $ms" >> gpurun_out/r02_v20_channel_padding_any.log
  done
done
cat gpurun_out/r02_v20_channel_padding_any.log
