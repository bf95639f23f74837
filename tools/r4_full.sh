set -x
OUT=gpurun_out/r4k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?
tail -8 $OUT/pytest.log
grep -q "Memory access fault" $OUT/pytest.log && exit 3
[ $rc -eq 0 ] || exit $rc
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4k/bench.json').read().strip().splitlines()[-1])
print(json.dumps(d["summary"], indent=1))
P
