"""Print the kernel table of a bench.py JSON line (tools aid)."""
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(r["metric"], "|", r["value"], r["unit"], "|", r["ms_per_step"], "ms/step | host", r["host_enqueue_ms_per_step"])
rf = r["roofline"]
print({k: rf.get(k) for k in ("kernel", "frac", "bound", "achieved", "avg_launch_us", "launches_per_step", "launches_per_step_all_kernels", "instrumented_ms_per_step")})
print(rf["step"])
for k, v in rf.get("kernel_ms_per_step", {}).items():
    print(f"{v:9.3f}  {k}")
print(rf.get("class_ms_per_step"))
