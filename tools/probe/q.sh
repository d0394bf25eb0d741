cd $GRAFT_REPO_ROOT
python -m pytest tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_run.sh a1ab base a1late
