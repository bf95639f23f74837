cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_gpu_golden.py -q -m gpu -k "C1_plumbing or degenerate or exact_degenerate or extra_root or wave_of_72 or live_oracle or gives_up or zero_level or hard_paths or fuzz" > gpurun_out/r3g_tests.log 2>&1; grep -v "^$" gpurun_out/r3g_tests.log | tail -60
