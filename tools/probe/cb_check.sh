cd $GRAFT_REPO_ROOT
echo "== conv bwd"; DQ_LIB_PATH=tools/probe/stamps/s4.so python tools/stamp_run.py 4 2>&1 | tail -10
python -m pytest tests/test_qnet_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/ab_run.sh cbab base prevcb
