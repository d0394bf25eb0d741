mkdir -p gpurun_out/r06e; o=gpurun_out/r06e
tools/probe/bin/trapsts_probe > $o/trapsts.txt 2>&1; cat $o/trapsts.txt
python -m pytest tests/test_qnet_gpu.py -m gpu -q -s -k "baseline_batch" 2>&1 | grep -E "ReLU pre-activation|passed|failed" > $o/fragile.txt; cat $o/fragile.txt
