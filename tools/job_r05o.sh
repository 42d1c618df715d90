# round-5 checkpoint: whole GPU suite
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05o_gputests.log 2>&1; tail -4 gpurun_out/r05o_gputests.log
