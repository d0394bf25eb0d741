cd $GRAFT_REPO_ROOT
echo "== dense fwd WG timeline (loop shape)"; DQ_LIB_PATH=tools/probe/stamps/s12.so python tools/probe/dense_fwd_timeline.py 2>&1 | tail -6
echo "== dense fwd training WG phases"; DQ_LIB_PATH=tools/probe/stamps/s2t.so DQ_STAMP_LOOP=1 python tools/stamp_run.py 2 2>&1 | tail -8
