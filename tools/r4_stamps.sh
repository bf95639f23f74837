set -x
OUT=gpurun_out/r4f
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/stamps.py C3 > $OUT/stamps_C3.out 2> $OUT/stamps_C3.err && python tools/stamps_summary.py $OUT/stamps_C3.err
