cd $GRAFT_REPO_ROOT
DQ_LIB_PATH=tools/probe/ab/wtst.so timeout 200 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu -k "training_forward or baseline_batch" 2>&1 | tail -2
timeout 400 bash tools/ab_run.sh wt base wtst
