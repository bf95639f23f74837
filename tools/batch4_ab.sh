# developer tool: the 4-LP batched leg of bench.py alone, three times (box-to-box spread against code changes)
for i in 1 2 3; do python bench.py --steps 1 --warmup 1 --no-cpu-baseline --milp-nodes 0 --c4 0 --frontier-vars 0 --frontier-wide-vars 0 --frontier-xwide-vars 0 --general 0 --concurrent 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
b = d.get('batched', {})
print('batched: %.2f ms, %.0f pivots/s, vs_single %.3f | single %.2f ms' % (1e3 * b.get('seconds', 0), b.get('pivots_per_s', 0), b.get('vs_single', 0), d['ms_per_step']))
"; done
