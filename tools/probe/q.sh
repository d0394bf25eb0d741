cd $GRAFT_REPO_ROOT
echo "== dense fwd loop shape"; DQ_LIB_PATH=tools/probe/stamps/s2.so DQ_STAMP_LOOP=1 timeout 200 python tools/stamp_run.py 2 2>&1 | tail -10
echo "== dense bwd in loop"; DQ_LIB_PATH=tools/probe/stamps/s3.so timeout 200 python tools/stamp_loop.py 3 2>&1 | tail -18
echo "== dense bwd wg timeline"; DQ_LIB_PATH=tools/probe/stamps/s23.so timeout 200 python tools/stamp_loop.py 23 2>&1 | tail -6
echo "== rider"; DQ_LIB_PATH=tools/probe/stamps/s6.so timeout 200 python tools/stamp_loop.py 6 2>&1 | tail -6
echo "== learn-mode SQ counters"
timeout 300 tools/pmc_any.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" sqL bench.py --mode learn --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -8
rm -rf gpurun_out/sqL
