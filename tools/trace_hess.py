"""Summarise the Hessenberg kernels of a rocprofv3 kernel trace (csv): per kernel/queue totals, chain-queue idle gaps,
a short excerpt.  usage: trace_hess.py kernel_trace.csv"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'psd_hess2' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
print(len(rows), "dispatches")
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = (r['Queue_Id'], r['Kernel_Name'][5:24], r['Grid_Size_X'], r['Grid_Size_Y'])
    agg[k][0] += 1
    agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for k, v in agg.items():
    print(k, v[0], round(v[1] / 1e6, 1), 'ms', round(v[1] / v[0] / 1e3, 2), 'us')
print('span ms', (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6)
chain = [r for r in rows if 'link' in r['Kernel_Name']]
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(chain, chain[1:])]
gaps.sort()
print('chain gaps: total ms', sum(gaps) / 1e6, 'median us', gaps[len(gaps) // 2] / 1e3, 'p90', gaps[int(len(gaps) * .9)] / 1e3,
      'p99', gaps[int(len(gaps) * .99)] / 1e3, 'max', gaps[-1] / 1e3)
# chain durations by quartile of the run
n = len(chain)
for q in range(4):
    part = chain[q * n // 4:(q + 1) * n // 4]
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in part]
    print('quartile', q, 'avg chain us', sum(d) / len(d) / 1e3)
i0 = len(rows) // 3
t0 = int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0 + 40]:
    s = int(r['Start_Timestamp']) - t0
    e = int(r['End_Timestamp']) - t0
    print(r['Queue_Id'], r['Kernel_Name'][5:20], r['Grid_Size_Y'], round(s / 1e3, 1), round(e / 1e3, 1), round((e - s) / 1e3, 1))
