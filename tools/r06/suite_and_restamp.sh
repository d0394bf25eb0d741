mkdir -p gpurun_out/r06t; o=gpurun_out/r06t
python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; tail -4 $o/gputests.log | cut -c1-300
tools/restamp_pmc.sh > $o/restamp.log 2>&1; tail -2 $o/restamp.log
