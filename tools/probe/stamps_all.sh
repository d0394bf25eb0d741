cd $GRAFT_REPO_ROOT
echo "== conv fwd persistent, loop shape"; DQ_LIB_PATH=tools/probe/stamps/s1.so DQ_STAMP_PERSIST=1 DQ_STAMP_LOOP=1 python tools/stamp_run.py 1 2>&1 | tail -20
echo "== dense fwd loop shape"; DQ_LIB_PATH=tools/probe/stamps/s2.so DQ_STAMP_LOOP=1 python tools/stamp_run.py 2 2>&1 | tail -10
echo "== dense bwd in loop"; DQ_LIB_PATH=tools/probe/stamps/s3.so python tools/stamp_loop.py 3 2>&1 | tail -18
echo "== dense bwd wg timeline"; DQ_LIB_PATH=tools/probe/stamps/s23.so python tools/stamp_loop.py 23 2>&1 | tail -6
echo "== conv bwd"; DQ_LIB_PATH=tools/probe/stamps/s4.so python tools/stamp_run.py 4 2>&1 | tail -10
