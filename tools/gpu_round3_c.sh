# round 3, GPU call C: phase timing of the fused reverse tail variants + stage table
cd /root/repo
TAG=${1:-r03_c}
mkdir -p gpurun_out
for v in tail4; do
ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_${v}timing.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 > /dev/null 2> gpurun_out/${TAG}_${v}_timing.log
grep "tail timing" gpurun_out/${TAG}_${v}_timing.log
ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_${v}.so timeout 600 python bench.py --steps 20 --warmup 5 --stages --no-cpu-baseline --no-gpu-reference --sustain 0 > gpurun_out/${TAG}_bench_c4_$v.log 2> gpurun_out/${TAG}_stages_c4_$v.log
echo "== $v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/${TAG}_bench_c4_$v.log)"
grep stage gpurun_out/${TAG}_stages_c4_$v.log
done
