set -x
OUT=gpurun_out/r4i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { "$@" > $OUT/last.log 2>&1; rc=$?; cat $OUT/last.log >> $OUT/all.log; if grep -q "Memory access fault" $OUT/last.log; then echo FAULT; tail -5 $OUT/last.log; exit 3; fi; return $rc; }
: > $OUT/all.log
run timeout -k 10 120 python tools/heavy_child.py 3 && \
run timeout -k 10 120 python tools/wave_split.py && \
run timeout -k 10 600 python -m pytest tests/test_gpu_golden.py -m gpu -x -q -k "C5_256 or C3_tree or C1_plumbing or hard_paths or artificial_exchange or degenerate_integer or equality_constrained or frontier"
echo "rc $?"
grep -v "^  File\|^Extension" $OUT/all.log | tail -40
