cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_golden.py -x -q -k "multi_workgroup or M_metric or C4_prefix or concurrent_solves or independent_large" > gpurun_out/r3e_tests.log 2>&1 || { tail -40 gpurun_out/r3e_tests.log; exit 1; }
tail -3 gpurun_out/r3e_tests.log
timeout -k 10 300 python tools/soak.py 60
for kn in "" "loop_grid=128" "loop_grid=512" "bt_groups=4"; do timeout -k 10 120 python tools/solve_n.py M 3 $kn 2>&1 | tail -1; done
for kn in "" "loop_grid=512" "loop_grid=128"; do timeout -k 10 120 python tools/solve_n.py C4 3 $kn 2>&1 | tail -1; done
timeout -k 10 120 python tools/first_diff.py C4 2>&1 | tail -3
