# kernel trace of the degenerate-tree leg (tools/degen_rate.py 1): where a 0.37 ms relaxation of a 5-row LP goes
set -x
OUT=gpurun_out/r4j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dg -- python3 tools/degen_rate.py 1 > $OUT/dg.log 2>&1
echo rc $?
tail -2 $OUT/dg.log
f=$(ls $OUT/dg/*/*kernel_stats.csv | head -1); head -22 $f | cut -c1-150
