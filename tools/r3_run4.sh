cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r3f_tests.log 2>&1; tail -30 gpurun_out/r3f_tests.log
timeout -k 10 200 python tools/soak.py 30 2>&1 | tail -3
