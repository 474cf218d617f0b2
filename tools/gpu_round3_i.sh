cd /root/repo
ALLEGRO_AMD_LIBRARY=/root/repo/allegro_amd/liballegro_amd_timing.so timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gpu-reference --no-profile --sustain 0 2>&1 >/dev/null | grep "fused timing"
