# developer tool: final-solve time by panel shape (diagnostic flavour of the library: GOMILP_LUC_SLOTS)
set -x
OUT=gpurun_out/r4n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export GOMILP_DEBUG_BUILD=1
: > $OUT/sweep.log
for cfg in 12 11 10 13; do GOMILP_LUC_SLOTS=$cfg python tools/final_cfg.py M >> $OUT/sweep.log 2>&1 || exit 1; grep -q "Memory access fault" $OUT/sweep.log && exit 3; done
for cfg in 20 21 22 23; do GOMILP_LUC_SLOTS=$cfg python tools/final_cfg.py C2 >> $OUT/sweep.log 2>&1 || exit 1; grep -q "Memory access fault" $OUT/sweep.log && exit 3; done
for cfg in 20 21 22 23; do echo "heavy cfg $cfg" >> $OUT/sweep.log; GOMILP_LUC_SLOTS=$cfg python tools/heavy_child.py 3 2>&1 | grep "sample 0" | tail -1 >> $OUT/sweep.log || exit 1; grep -q "Memory access fault" $OUT/sweep.log && exit 3; done
for cfg in 99 30 31 32; do GOMILP_LUC_SLOTS=$cfg python tools/final_cfg.py C3 >> $OUT/sweep.log 2>&1 || exit 1; grep -q "Memory access fault" $OUT/sweep.log && exit 3; done
cat $OUT/sweep.log
