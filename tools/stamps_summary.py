"""Summarise the bt_stamps JSON lines the engine prints (tools/stamps.py ... 2> file)."""
import json, sys
import numpy as np
SEG = ["r+wavemin A", "barrier A", "blockmin A", "col load", "col fma", "ratio+wmin B", "barrier B", "blockmin B", "row load wait", "row fma/r/V", "rowissue+u/xb/U"]
for line in open(sys.argv[1]):
    line = line.strip()
    if not line.startswith('{'):
        continue
    d = json.loads(line)['bt_stamps']
    a = np.array([x for x in d['cycles_per_pivot_by_wave'] if sum(x) > 0])
    tot = a.sum(1)
    print('m', d['m'], 'nn', d['nn'], 'phase', d['phase'], 'pivots', d['pivots'], 'waves', len(a), 'cycles/pivot %.0f..%.0f' % (tot.min(), tot.max()))
    print('   ' + '  '.join('%s %.0f' % (s, a[:, i].mean()) for i, s in enumerate(SEG)))
