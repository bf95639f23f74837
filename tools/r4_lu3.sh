# kernel trace of the final solve at the metric size (look-ahead schedule vs the plain one)
set -x
OUT=gpurun_out/r4f
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/look -- python3 tools/final_n.py M 4 > $OUT/look.log 2>&1 && \
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/plain -- python3 tools/final_n.py M 4 lu_blocked=2 > $OUT/plain.log 2>&1
echo rc $?
for d in look plain; do f=$(ls $OUT/$d/*/*kernel_stats.csv | head -1); echo $d; grep -i "luc\|Name" $f | cut -c1-170; done
tail -n 3 $OUT/look.log $OUT/plain.log
