# kernel trace of the C3 tree's waves (tools/c3_waves.py): where a 2.6 ms wave of 4 nodes goes
set -x
OUT=gpurun_out/r4i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 tools/c3_waves.py 47 > $OUT/c3.log 2>&1
echo rc $?
grep "^wave\|^nodes\|^sum" $OUT/c3.log | tail -8
f=$(ls $OUT/c3/*/*kernel_stats.csv | head -1); head -25 $f | cut -c1-170
