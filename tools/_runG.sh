set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_bnact.py tests/test_hip_network.py tests/test_hip_kernels.py tests/test_hip_determinism.py -x -q > gpurun_out/t_k.log 2>&1 || { tail -40 gpurun_out/t_k.log; exit 1; }
tail -2 gpurun_out/t_k.log
for r in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 20 --no-roofline | tail -1 | cut -c60-110
done
