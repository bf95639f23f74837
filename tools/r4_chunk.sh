# pivots per launch of the persistent loop kernel (knob loop_chunk) at the metric size
set -x
OUT=gpurun_out/r4h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/chunk.log
for ch in 512 1024 2048 4096; do
  echo "== loop_chunk $ch" >> $OUT/chunk.log
  timeout -k 10 120 python tools/final_n.py M 4 loop_chunk=$ch >> $OUT/chunk.log 2>&1 || { echo FAILED; break; }
  grep -q "Memory access fault" $OUT/chunk.log && { echo FAULT; exit 3; }
done
grep -v "^Ext" $OUT/chunk.log
