set -x
OUT=gpurun_out/var
mkdir -p $OUT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
HEAD="--no-cpu-baseline --concurrent 0 --milp-nodes 0 --c4 0 --frontier-vars 0"
for i in 1 2; do
python bench.py $HEAD > $OUT/un_$i.json 2> $OUT/un_$i.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$i -- python3 bench.py $HEAD > $OUT/pr_$i.json 2> $OUT/pr_$i.err
done
