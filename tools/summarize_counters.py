#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc{A,B,C} (tools/profile_counters.sh) -> profiles/<tag>_sq_counters.csv: per kernel,
per bench step (sum over the kernel's launches of a step), one column per counter."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1_v5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
agg = collections.defaultdict(lambda: collections.defaultdict(float))
order = []
for grp in "ABC":
    fs = glob.glob(os.path.join(ROOT, "gpurun_out", f"{tag}_pmc{grp}", "*", "*_counter_collection.csv"))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        name = next((n for n in ("k_init", "k_huf", "k_seq_pipe", "k_exec", "k_fse_build", "k_huf_build") if "mzd::" + n + "(" in r["Kernel_Name"]), None)
        if name is None:
            continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] not in order:
            order.append(r["Counter_Name"])
out = os.path.join(ROOT, "profiles", tag + "_sq_counters.csv")
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --pmc <group> -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 (three passes, tools/profile_counters.sh);\n")
    f.write("# values per bench step (65536 frames, 13.7 M 64-sequence tiles); SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over wavefronts\n")
    w = csv.writer(f)
    w.writerow(["kernel"] + order)
    for k in ("k_huf", "k_seq_pipe", "k_exec"):
        if k in agg:
            w.writerow([k] + [f"{agg[k].get(c, 0) / steps:.4g}" for c in order])
print(open(out).read())
