set -x
OUT=gpurun_out/r4g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
run() { "$@" > $OUT/last.log 2>&1; rc=$?; cat $OUT/last.log >> $OUT/all.log; if grep -q "Memory access fault" $OUT/last.log; then echo FAULT; tail -5 $OUT/last.log; exit 3; fi; return $rc; }
: > $OUT/all.log
run python tools/heavy_child.py 3 && \
run python tools/wave_prof.py 8 4 && \
run timeout -k 10 900 python -m pytest tests -m gpu -x -q
echo "rc $?"
grep -v "^  File\|^Extension" $OUT/all.log | tail -40
