# kernel stats of the 2048-wide C5 wave (tools/wave_prof.py 6 4 0 11)
set -x
OUT=gpurun_out/r4k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wide -- python3 tools/wave_prof.py 6 4 0 11 > $OUT/wide.log 2>&1
echo rc $?
tail -4 $OUT/wide.log
f=$(ls $OUT/wide/*/*kernel_stats.csv | head -1); head -16 $f | cut -c1-200
