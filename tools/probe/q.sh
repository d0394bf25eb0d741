cd $GRAFT_REPO_ROOT
DQ_LIB_PATH=tools/probe/ab/gxwarm.so timeout 120 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu -k "td_backward or one_full_update" 2>&1 | tail -2
timeout 300 bash tools/ab_run.sh gxw base gxwarm
