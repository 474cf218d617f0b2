# the reference's constructor defaults (num_tensor_features 16, readout width 32; allegro_models.py:126-137) at l_max = 2:
# zero-padded onto the 64-wide kernels vs the narrow kernels
cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r02_v21_padding_defaults.log
export AA_BENCH_CFG='{"num_tensor_features": 16, "readout_mlp_hidden_layers_width": 32}'
for n in 2 11 23; do
  for v in "pad" "AA_NO_PAD=1"; do
    unset AA_NO_PAD
    if [ "$v" = "AA_NO_PAD=1" ]; then export AA_NO_PAD=1; fi
    ms=$(AA_BENCH_CELLS=$n timeout 600 python bench.py --workload c2 --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "constructor defaults (u=16, readout 32) l_max=2 cells=$n atoms=$((8*n*n*n)) [$v] $ms" >> gpurun_out/r02_v21_padding_defaults.log
  done
done
cat gpurun_out/r02_v21_padding_defaults.log
