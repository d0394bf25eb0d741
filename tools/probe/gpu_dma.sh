python -m pytest tests/test_qnet_gpu.py tests/test_shipped_weights.py -m gpu -x -q 2>&1 | tail -2
tools/ab_run.sh dma base 2>&1 | grep rep
