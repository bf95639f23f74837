set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_golden.py -x -q -k "multi_workgroup_block_kernel or M_metric or C4_prefix" > gpurun_out/r3c_tests.log 2>&1 || { tail -40 gpurun_out/r3c_tests.log; exit 1; }
tail -3 gpurun_out/r3c_tests.log
timeout -k 10 120 python tools/solve_n.py M 4 > gpurun_out/r3c_M_loop.log 2>&1; cat gpurun_out/r3c_M_loop.log
timeout -k 10 120 python tools/solve_n.py C4 3 > gpurun_out/r3c_C4_loop.log 2>&1; cat gpurun_out/r3c_C4_loop.log
timeout -k 10 120 python tools/solve_n.py C2 3 > gpurun_out/r3c_C2.log 2>&1; cat gpurun_out/r3c_C2.log
