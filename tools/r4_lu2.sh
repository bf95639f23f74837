set -x
OUT=gpurun_out/r4e
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { "$@" > $OUT/last.log 2>&1; rc=$?; cat $OUT/last.log >> $OUT/all.log; if grep -q "Memory access fault" $OUT/last.log; then echo FAULT; tail -5 $OUT/last.log; exit 3; fi; return $rc; }
: > $OUT/all.log
run timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked_lu and 200-2" && \
run timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blocked_lu or dense_lp_matches or frontier_pool or bnb_children" && \
GOMILP_DEBUG_BUILD=1 run python tools/lu_stamps.py M && \
GOMILP_DEBUG_BUILD=1 run python tools/lu_stamps.py C3 && \
GOMILP_DEBUG_BUILD=1 run python tools/lu_stamps.py C2 && \
run python tools/final_n.py M 3 && run python tools/final_n.py C2 3 && run python tools/final_n.py C3 3 && \
run python tools/heavy_child.py 3
echo "rc $?"
grep -v "^  File\|^Extension" $OUT/all.log | tail -40
