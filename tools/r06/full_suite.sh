mkdir -p gpurun_out/r06m; o=gpurun_out/r06m
python -m pytest tests -m gpu -x -q > $o/gputests.log 2>&1; tail -6 $o/gputests.log | cut -c1-300
tools/restamp_pmc.sh > $o/restamp.log 2>&1; tail -2 $o/restamp.log
