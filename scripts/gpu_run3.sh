cd /root/repo
mkdir -p gpurun_out/r3c
python scripts/c5_diverge.py > gpurun_out/r3c/c5_diverge.txt 2>&1
python -m pytest tests/test_gpu_config4.py -q -s 2>&1 | tail -30 > gpurun_out/r3c/c4.txt
