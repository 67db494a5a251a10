#!/usr/bin/env python3
"""Turn the scratch outputs of tools/profile_round.sh (gpurun_out/<tag>_stats|fetch|write, <tag>_bench.json)
into the committed summaries under profiles/: <tag>_kernel_stats.csv, <tag>_bench.json, r1_traffic.json."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1_v5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7  # bench steps + warmup of the profiled command
go, po = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")

st = glob.glob(os.path.join(go, tag + "_stats", "*", "*_kernel_stats.csv"))[0]
rows = list(csv.reader(open(st)))
with open(os.path.join(po, tag + "_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0"
            "   (default workload: 65536 frames, config 4; k_seq_pipe and k_exec are launched twice per step)\n")
    f.write(f"# bench line of the same build: profiles/{tag}_bench.json\n")
    w = csv.writer(f)
    for r in rows:
        w.writerow([c[:100] for c in r])

def per_kernel(kind):
    f = glob.glob(os.path.join(go, f"{tag}_{kind}", "*", "*_counter_collection.csv"))[0]
    agg = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = next((n for n in ("k_init", "k_huf", "k_seq", "k_exec") if "mzd::" + n in k), None)
        if name:
            agg[name] += float(r["Counter_Value"])
    return agg

fetch, write = per_kernel("fetch"), per_kernel("write")
old = json.load(open(os.path.join(po, "r1_traffic.json")))
kern = {}
for k in ("k_init", "k_seq", "k_huf", "k_exec"):
    fb = int(fetch[k] / steps * 1024)
    kern[k] = {"fetch_bytes": fb, "fetch_bytes_x2": 2 * fb, "write_bytes": int(write[k] / steps * 1024)}
old["kernels"] = kern
old["command"] = "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0"
old["build"] = tag
json.dump(old, open(os.path.join(po, "r1_traffic.json"), "w"), indent=1)
line = open(os.path.join(go, tag + "_bench.json")).read().strip().splitlines()[-1]
open(os.path.join(po, tag + "_bench.json"), "w").write(line + "\n")
print(json.dumps(kern, indent=1))
