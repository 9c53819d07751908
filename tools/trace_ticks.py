"""Summarise the iteration kernels of a rocprofv3 kernel trace (csv): per-queue totals, a few ticks in order, idle gaps
of the main queue.  usage: trace_ticks.py kernel_trace.csv"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'psd_rq' in r['Kernel_Name'] and 'init' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(len(rows), "dispatches")
for frac in (0.3, 0.6):
    i0 = int(len(rows) * frac)
    while 'step_mb' not in rows[i0]['Kernel_Name']:
        i0 += 1
    t0 = int(rows[i0]['Start_Timestamp'])
    for r in rows[i0:i0 + 9]:
        s = int(r['Start_Timestamp']) - t0
        e = int(r['End_Timestamp']) - t0
        print(r['Queue_Id'], r['Kernel_Name'][:16], r['Grid_Size_X'], round(s / 1e3, 1), round(e / 1e3, 1), round((e - s) / 1e3, 1))
    print()
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = (r['Queue_Id'], r['Kernel_Name'][:16], r['Grid_Size_X'])
    agg[k][0] += 1
    agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in agg.items():
    print(k, v[0], round(v[1] / 1e6, 1), 'ms', round(v[1] / v[0] / 1e3, 1), 'us')
print('span ms', (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6)
mainq = collections.Counter(r['Queue_Id'] for r in rows if 'step_mb' in r['Kernel_Name']).most_common(1)[0][0]
q1 = [r for r in rows if r['Queue_Id'] == mainq]
gap = collections.defaultdict(lambda: [0, 0])
for a, b in zip(q1, q1[1:]):
    g = int(b['Start_Timestamp']) - int(a['End_Timestamp'])
    k = (a['Kernel_Name'][:14], b['Kernel_Name'][:14])
    gap[k][0] += 1
    gap[k][1] += g
for k, v in gap.items():
    print('gap', k, v[0], round(v[1] / 1e6, 1), 'ms')
