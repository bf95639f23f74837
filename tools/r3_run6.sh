cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -q -m gpu -x > gpurun_out/r3i_tests.log 2>&1; grep -v "^$" gpurun_out/r3i_tests.log | tail -30
