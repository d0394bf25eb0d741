cd $GRAFT_REPO_ROOT
for v in s_maxilp s_maxmem s_iter; do DQ_LIB_PATH=tools/probe/ab/$v.so timeout 120 python -m pytest tests/test_qnet_gpu.py -x -q -m gpu -k "training_forward_backward and c3" 2>&1 | tail -1; done
timeout 600 bash tools/ab_run.sh sched base s_maxilp s_maxmem s_iter
