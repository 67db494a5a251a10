#!/usr/bin/env python3
"""gpurun_out/<tag>_cfg<N>_pmc{A,B,C} (tools/profile_counters.sh) -> gpurun_out/<tag>_cfg<N>_sq_counters.csv: per kernel,
per pass of the hot path (sum over the kernel's launches of a pass), one column per counter."""
import collections, csv, glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cfg = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
agg = collections.defaultdict(lambda: collections.defaultdict(float))
order = []
for grp in "ABC":
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_cfg{cfg}_pmc{grp}", "*", "*_counter_collection.csv"))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"mzd::(k_\w+)", r["Kernel_Name"])
        if not m or m.group(1) in ("k_fse_build", "k_huf_build", "k_copy_ceiling", "k_init"):
            continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in order:
            order.append(r["Counter_Name"])
out = os.path.join(ROOT, "gpurun_out", f"{tag}_cfg{cfg}_sq_counters.csv")
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --config {cfg} --steps 3 --warmup 1 --cpu-seconds 0 (three passes, tools/profile_counters.sh);\n")
    f.write("# values per pass of the hot path; SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over wavefronts\n")
    w = csv.writer(f)
    w.writerow(["kernel"] + order)
    for k in sorted(agg):
        w.writerow([k] + [f"{agg[k].get(c, 0) / steps:.4g}" for c in order])
print(open(out).read())
