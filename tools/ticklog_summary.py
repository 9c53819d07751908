"""Summarise a PSD_TICKLOG file (tick, longest workgroup of the chase launch in 100 MHz ticks, mask of the phases it
ran): time and count per phase combination.  usage: ticklog_summary.py ticklog.txt"""
import collections
import sys

import numpy as np

a = np.loadtxt(sys.argv[1], dtype=np.int64)
dur = a[:, 1] / 100.0
mask = a[:, 2]
names = {0: 'DEC', 1: 'RQ', 2: 'SHIFT', 3: 'QR', 4: 'DEFL', 5: 'NEXT', 6: 'FINAL', 8: 'TWAIT'}


def nm(m):
    return '+'.join(v for k, v in names.items() if m >> k & 1) or '(no leader)'


agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for d, m in zip(dur, mask):
    k = nm(m)
    agg[k][0] += 1
    agg[k][1] += d
    agg[k][2] = max(agg[k][2], d)
print('ticks', len(dur), 'sum of the longest leader per tick: %.1f ms' % (dur.sum() / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:40s} n={v[0]:5d} total={v[1] / 1e3:7.1f} ms avg={v[1] / v[0]:7.1f} us max={v[2]:7.1f}")
