#!/usr/bin/env python3
"""Summarise gpurun_out/steptrace kernel trace by kernel name (per step)."""
import csv, glob, collections, re, sys
f = glob.glob('gpurun_out/steptrace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_in_pack' in r['Kernel_Name']]
a, b = idx[-6], idx[-1]          # 5 steady-state steps
steps = 5.0
agg = collections.OrderedDict()
for r in rows[a:b]:
    name = r['Kernel_Name'].split('(')[0].replace('void stcd::', '').replace('void ', '').replace('stcd::', '')
    m = re.match(r'_ZN4stcd\d+(k_[a-z_0-9]+)', name)
    if m: name = m.group(1)
    agg.setdefault(name[:44], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
wall = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 / steps
out = sorted(((sum(v) / steps, k, len(v) / steps) for k, v in agg.items()), reverse=True)
tot = sum(s for s, _, _ in out)
for s, k, n in out[:int(sys.argv[1]) if len(sys.argv) > 1 else 30]:
    print(f"{k:46s} n/step {n:6.1f}  sum/step {s:8.1f} us   avg {s/n:7.1f}")
print(f"kernel sum {tot:.1f} us/step   wall {wall:.1f} us/step")
