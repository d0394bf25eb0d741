cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 400 bash tools/ab_run.sh xl base xearly
