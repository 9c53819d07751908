"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: dispatch count and summed counter values."""
import csv
import glob
import json
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"].split("(")[0]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k].add(row["Dispatch_Id"])
out = {k: {"dispatches": len(cnt[k]), **dict(v)} for k, v in agg.items()}
print(json.dumps(out, indent=1, sort_keys=True))
