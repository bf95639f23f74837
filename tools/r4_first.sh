set -x
OUT=gpurun_out/r4a
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
python tools/final_n.py M 4 > $OUT/final_M.log 2>&1
python tools/final_n.py C2 4 > $OUT/final_C2.log 2>&1
python tools/heavy_child.py 3 > $OUT/heavy.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_heavy -- python3 tools/heavy_child.py 3 > $OUT/heavy_prof.log 2>&1
cat $OUT/final_M.log $OUT/heavy.log
