mkdir -p gpurun_out/r06ac; python -m pytest tests -m gpu -x -q > gpurun_out/r06ac/gputests.log 2>&1; tail -4 gpurun_out/r06ac/gputests.log | cut -c1-300
