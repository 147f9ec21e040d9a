#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (written by profiles/collect.sh) -> the committed summaries profiles/<tag>_*.csv|json.

    python profiles/summarize.py r01
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, 'gpurun_out', f'prof_{tag}')
dst = os.path.join(root, 'profiles')


def short(name):
    return name.split('(')[0][:90]


# 1. bench line
line = open(os.path.join(src, 'bench_n1.json')).read().strip().splitlines()[-1]
json.loads(line)
open(os.path.join(dst, f'{tag}_bench_n1.json'), 'w').write(line + '\n')

# 2. kernel stats (rocprofv3 --kernel-trace --stats)
stats = max(glob.glob(os.path.join(src, 'stats', '*', '*_kernel_stats.csv')), key=os.path.getmtime)   # the latest collection
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, f'{tag}_bench_kernel_stats.csv'), 'w') as f:
    f.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline  (MI355X, hipGraph mode;\n')
    f.write('# the run holds 1 eager setup step + 2 instrumented eager fwd+bwd passes + 3 capture/warm-up + 3 timed steps)\n')
    f.write('Name,Calls,TotalDurationNs,AverageNs,Percentage\n')
    for r in rows[:40]:
        f.write(f'"{r["Name"][:110]}",{r["Calls"]},{r["TotalDurationNs"]},{int(float(r["AverageNs"]))},{r["Percentage"]}\n')


# 3. HBM traffic per kernel from the two --pmc passes
def pmc(dirname, counter):
    files = sorted(glob.glob(os.path.join(src, dirname, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[-1])):
        if r.get('Counter_Name') != counter:
            continue
        a = acc[short(r['Kernel_Name'])]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return acc


try:
    fetch, write = pmc('pmc_fetch', 'FETCH_SIZE'), pmc('pmc_write', 'WRITE_SIZE')
    names = sorted(fetch, key=lambda n: -(2 * fetch[n][1] + write.get(n, [0, 0.0])[1]))
    with open(os.path.join(dst, f'{tag}_hbm_traffic_per_kernel.csv'), 'w') as f:
        f.write('# HBM traffic per launch from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, eager bench step).\n')
        f.write('# Counter unit: KB.  FETCH_SIZE is doubled below per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read).\n')
        f.write('kernel,launches,avg_fetch_MB_corrected,avg_write_MB,avg_total_MB\n')
        for n in names[:30]:
            c, fs = fetch[n]
            ws = write.get(n, [c, 0.0])[1]
            fm, wm = 2 * fs / c / 1024, ws / max(write.get(n, [c, 0.0])[0], 1) / 1024
            f.write(f'"{n}",{c},{fm:.1f},{wm:.1f},{fm + wm:.1f}\n')
except (IndexError, KeyError, FileNotFoundError) as e:
    print('no PMC data:', e)
print('wrote', sorted(os.listdir(dst)))
