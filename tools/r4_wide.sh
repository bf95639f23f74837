# the 2048-wide C5 wave: wall against the C side's own clock, then kernel stats (tools/wave_prof.py 6 4 0 11)
set -x
OUT=gpurun_out/r4k
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/wave_prof.py 6 4 0 11 2>&1 | tail -4
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/wide -- python3 tools/wave_prof.py 6 4 0 11 > $OUT/wide.log 2>&1
echo rc $?
f=$(ls -t $OUT/wide/*/*kernel_stats.csv | head -1); head -14 $f | cut -c1-170
