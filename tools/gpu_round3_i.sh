cd /root/repo
mkdir -p gpurun_out
: > gpurun_out/r03_v11_pytest_gpu_x5.log
for i in 1 2 3 4 5; do
  echo "== full GPU suite, run $i" >> gpurun_out/r03_v11_pytest_gpu_x5.log
  timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3 >> gpurun_out/r03_v11_pytest_gpu_x5.log
done
python -c "from allegro_amd import build; print('kernel source hash', build.source_hash())" >> gpurun_out/r03_v11_pytest_gpu_x5.log
cat gpurun_out/r03_v11_pytest_gpu_x5.log
