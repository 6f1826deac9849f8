#!/usr/bin/env python3
"""Copy the judged artefacts of tools/profile_round.sh from gpurun_out/<tag>/ into profiles/ (tracked):
  <tag>_bench.json          the bench.py line (roofline + cpu_baseline)
  <tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats summary of `bench.py --no-cpu-baseline`
  <tag>_pmc_traffic.txt     per-kernel HBM-side bytes per launch from the separate FETCH_SIZE / WRITE_SIZE passes
                            (FETCH_SIZE, WRITE_SIZE are KiB; FETCH_SIZE doubled per the gfx950 correction of
                            /opt/skills/guides/MI355X_MICROARCH.md, section HBM)
usage: python tools/collect_profiles.py r01
"""
import collections, csv, glob, json, os, re, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
STEPS = 9            # bench.py --steps 3 --warmup 1 in the --pmc passes of tools/profile_round.sh, + the 5 idle-queue steps every run appends
src, dst = f"gpurun_out/{tag}", "profiles"
os.makedirs(dst, exist_ok=True)
line = [l for l in open(f"{src}/bench.json") if l.startswith("{")][-1]
json.loads(line)
open(f"{dst}/{tag}_bench.json", "w").write(line)
newest = lambda pattern: max(glob.glob(pattern, recursive=True), key=os.path.getmtime)   # re-runs leave older files behind
stats = newest(f"{src}/stats/**/*kernel_stats.csv")
shutil.copy(stats, f"{dst}/{tag}_kernel_stats.csv")


def short(n):
    n = n.split("(")[0].replace("void ", "")
    m = re.match(r"_ZN4stcd\d+(k_[a-z_0-9]+)", n)
    return "stcd::" + m.group(1) if m else n


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for f in [newest(f"{path}/**/*counter_collection.csv")]:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return agg

fetch, write = per_kernel(f"{src}/pmc_fetch", "FETCH_SIZE"), per_kernel(f"{src}/pmc_write", "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    if "stcd" not in k:
        continue
    f = fetch.get(k, [0.0]); w = write.get(k, [0.0])
    fb = 2.0 * 1024.0 * sum(f) / len(f)          # KiB -> bytes, x2 (gfx950: 128-B requests tallied at 64 B)
    wb = 1024.0 * sum(w) / len(w)
    rows.append((fb + wb, k, len(f), fb, wb))
rows.sort(reverse=True)
with open(f"{dst}/{tag}_pmc_traffic.txt", "w") as o:
    o.write("# HBM-side traffic per launch (bytes), averaged over the launches of 4 bench steps; separate --pmc passes\n")
    o.write("# read = 2 * 1024 * FETCH_SIZE (gfx950 correction), write = 1024 * WRITE_SIZE\n")
    o.write(f"# step_total_B (sum over kernels of launches x total_B / {STEPS} steps): {sum(t * n for t, _, n, _, _ in rows) / STEPS:.0f}\n")
    o.write(f"{'kernel':48s} {'launches':>8s} {'read_B':>14s} {'write_B':>14s} {'total_B':>14s}\n")
    for tot, k, n, fb, wb in rows:
        o.write(f"{k[:48]:48s} {n:8d} {fb:14.0f} {wb:14.0f} {tot:14.0f}\n")
print(open(f"{dst}/{tag}_pmc_traffic.txt").read()[:3000])

# the bench line was printed before the counter passes of the same command ran: fill its traffic fields from THIS summary
d = json.loads(line)
r = d.get("roofline", {})
dom = r.get("kernel")
step_total = sum(t * n for t, _, n, _, _ in rows) / STEPS
for tot, k, n, fb, wb in rows:
    if dom and (k == dom or (len(dom) > 48 and k[:48] == dom[:48]) or k.startswith(dom.split("<")[0] + "<") and k.replace(" ", "") == dom.replace(" ", "")):
        r["traffic"] = int(tot); r["traffic_source"] = f"profiles/{tag}_pmc_traffic.txt"
        break
if dom and r.get("traffic") is None:      # a kernel CLASS of the engine's timers: launch-weighted mean over its instantiations
    base = dom.split("<")[0]
    fam = [(tot, n) for tot, k, n, fb, wb in rows if k.startswith(base + "<") or k == base]
    if fam:
        r["traffic"] = int(sum(t * n for t, n in fam) / sum(n for _, n in fam))
        r["traffic_source"] = f"profiles/{tag}_pmc_traffic.txt (launch-weighted mean over {len(fam)} instantiations)"
r["traffic_step"] = int(step_total); r["traffic_step_source"] = f"profiles/{tag}_pmc_traffic.txt"
if "step" in r and r["step"].get("alg_bytes"):
    r["traffic_step_over_algorithmic"] = round(step_total / r["step"]["alg_bytes"], 3)
# the dominant kernel's fraction must follow from the committed rocprof summary: merged AverageNs over its instantiations
rk = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(l for l in open(f"{dst}/{tag}_kernel_stats.csv") if not l.startswith("#")):
    nm = short(row["Name"])
    rk[nm.split("<")[0]][0] += float(row["TotalDurationNs"]); rk[nm.split("<")[0]][1] += int(row["Calls"])
if dom and dom.split("<")[0] in rk and r.get("alg_bytes_per_launch"):
    tot_ns, calls = rk[dom.split("<")[0]]
    avg_us = tot_ns / calls / 1e3
    r["rocprof_avg_launch_us"] = round(avg_us, 3)
    if r.get("bound") == "hbm":
        r["rocprof_frac"] = round(r["alg_bytes_per_launch"] / (avg_us * 1e-6) / 1e9 / r["peak"], 4)
    else:
        r["rocprof_frac"] = round(r["alg_flops_per_launch"] / (avg_us * 1e-6) / 1e12 / r["peak"], 4)
    r["rocprof_note"] = ("merged AverageNs of the dominant kernel's instantiations in profiles/%s_kernel_stats.csv (the same command; ALL its launches: "
                         "warm-up and timed steps with the weight gradients on the side stream -- concurrent kernels stretch each other's durations -- "
                         "plus the instrumented steps, which run on one stream)" % tag)
open(f"{dst}/{tag}_bench.json", "w").write(json.dumps(d) + "\n")
print("bench line:", {k: r.get(k) for k in ("kernel", "frac", "avg_launch_us", "rocprof_avg_launch_us", "rocprof_frac", "traffic", "traffic_step", "traffic_step_over_algorithmic")})
