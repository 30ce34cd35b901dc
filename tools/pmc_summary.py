#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, average counter value per dispatch over the last N dispatches."""
import collections, csv, glob, sys
root = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for f in sorted(glob.glob(root + '/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name'].split('(')[0][:20]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in sorted(agg):
        if 'mobi' not in k:
            continue
        print(f'[{f.split("/")[-2]}] {k}: ' + '  '.join(f'{c}={sum(v[-N:]) / len(v[-N:]):.4g}' for c, v in sorted(agg[k].items())))
