set -x
OUT=gpurun_out/r4l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/wave_prof.py 4 4 > $OUT/wave.log 2>&1
cat $OUT/wave.log | tail -5
ls -la $OUT/trace/*/
