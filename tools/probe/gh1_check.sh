cd $GRAFT_REPO_ROOT
python -m pytest tests/test_qnet_gpu.py tests/test_shipped_weights.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_agent_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -2
echo "== dense bwd in loop"; DQ_LIB_PATH=tools/probe/stamps/s3.so python tools/stamp_loop.py 3 2>&1 | tail -16 | cut -c1-200
echo "== dense bwd wg timeline"; DQ_LIB_PATH=tools/probe/stamps/s23.so python tools/stamp_loop.py 23 2>&1 | tail -4
bash tools/ab_run.sh gh1ab base
