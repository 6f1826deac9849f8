#!/usr/bin/env python3
"""Launch-by-launch timeline of one training step from a rocprofv3 --kernel-trace CSV directory:
    python3 tools/step_timeline.py <trace dir> [out.txt]
Steps are delimited by k_in_pack (the step's first engine launch after the packing prologue); every position of the step takes the
MEDIAN duration over the steady-state steps (launch sequences of equal length), so one disturbed step does not show."""
import csv, glob, re, sys

d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))


def short(n):
    n = n.split('(')[0].replace('void stcd::', '').replace('void ', '').replace('stcd::', '')
    m = re.match(r'_ZN4stcd\d+(k_[a-z_0-9]+)', n)
    return (m.group(1) if m else n)[:64]


idx = [i for i, r in enumerate(rows) if 'k_dropout_gen' in r['Kernel_Name']] or [i for i, r in enumerate(rows) if 'k_pack_jobs' in r['Kernel_Name']]
steps = [rows[idx[k]:idx[k + 1]] for k in range(len(idx) - 1)]
L = len(steps[-1])
steps = [s for s in steps if len(s) == L][-5:]
out = []
t0 = int(steps[-1][0]['Start_Timestamp'])
tot = 0.0
for j in range(L):
    durs = sorted((int(s[j]['End_Timestamp']) - int(s[j]['Start_Timestamp'])) / 1e3 for s in steps)
    med = durs[len(durs) // 2]
    tot += med
    out.append(f"{(int(steps[-1][j]['Start_Timestamp']) - t0) / 1e3:9.1f} {med:8.1f}  {short(steps[-1][j]['Kernel_Name'])}  grid {steps[-1][j].get('Grid_Size_X', steps[-1][j].get('Grid_Size', '?'))}")
wall = (int(steps[-1][-1]['End_Timestamp']) - t0) / 1e3
out.append(f"# {L} launches, {tot:.1f} us of kernel time (median per position over {len(steps)} steps), {wall:.1f} us wall from first start to last end")
txt = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(txt + "\n")
print(txt)
