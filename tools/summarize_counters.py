#!/usr/bin/env python3
"""gpurun_out/<tag>_cfg<N>_pmc{A,B,C,D} (tools/profile_counters.sh) -> gpurun_out/<tag>_cfg<N>_sq_counters.csv: per kernel,
per pass of the hot path (sum over the kernel's launches of a pass), one column per counter; and
gpurun_out/<tag>_issue_cfg<N>.json: what each kernel keeps busy INSIDE the CU (bench.py's roofline.issue), stamped like the
traffic file with the hash of the device sources and the workload."""
import collections, csv, datetime, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_src_sha16  # noqa: E402
tag, cfg = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cmd = sys.argv[4] if len(sys.argv) > 4 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
order = []
for grp in "ABCD":
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
    f.write(f"# rocprofv3 --kernel-trace --pmc <group> -- {cmd or 'python3 bench.py --config ' + cfg + ' --steps 3 --warmup 1 --cpu-seconds 0'} (tools/profile_counters.sh); device sources {kernel_src_sha16()}\n")
    f.write("# values per pass of the hot path; SQ_*CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over wavefronts\n")
    w = csv.writer(f)
    w.writerow(["kernel"] + order)
    for k in sorted(agg):
        w.writerow([k] + [f"{agg[k].get(c, 0) / steps:.4g}" for c in order])
print(open(out).read())

# ---- the binding resource inside the CU, per kernel (MI355X: 256 CUs x 4 SIMD-32; a wave64 VALU instruction holds its SIMD 2
# cycles -- /opt/skills/guides/MI355X_MICROARCH.md; GRBM_GUI_ACTIVE is summed over the 8 XCDs)
N_SIMD, N_XCD = 1024, 8
issue = {}
for k in sorted(agg):
    c = {n: v / steps for n, v in agg[k].items()}
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / N_XCD
    busy = c.get("SQ_BUSY_CU_CYCLES", 0)
    if cyc <= 0:
        continue
    issue[k] = {"kernel_cycles": int(cyc), "valu_wave_insts": int(c.get("SQ_INSTS_VALU", 0)), "salu_insts": int(c.get("SQ_INSTS_SALU", 0)),
                "lds_insts": int(c.get("SQ_INSTS_LDS", 0)), "branch_insts": int(c.get("SQ_INSTS_BRANCH", 0)),
                "valu_frac": round(c.get("SQ_INSTS_VALU", 0) * 2 / (N_SIMD * cyc), 4),
                "lds_pipe_frac": round(c.get("SQ_LDS_IDX_ACTIVE", 0) / busy, 4) if busy else None,
                "ta_busy_frac": round(c.get("TA_TA_BUSY_sum", 0) / (256 * cyc), 4),
                "l2_hit_frac": round(c.get("TCC_HIT_sum", 0) / (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)), 4) if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) else None}
m = re.search(r"--frames(?:-per-gpu)? (\d+)", cmd)
frames = int(m.group(1)) if m else {"2": 4096, "3": 4096, "4": 65536}.get(cfg)
m = re.search(r"--frame-bytes (\d+)", cmd)
m2 = re.search(r"--workload (\w+)", cmd)
m3 = re.search(r"--corpus-gib ([\d.]+)", cmd)
json.dump({"note": "per pass of the hot path, from separate --pmc passes (tools/profile_counters.sh). valu_frac = SQ_INSTS_VALU x 2 cycles / "
                   "(1024 SIMDs x kernel cycles); lds_pipe_frac = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES; ta_busy_frac = TA_TA_BUSY_sum / (256 CUs x "
                   "kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs",
           "measured": {"by": "the builder (tools/profile_counters.sh inside a gpurun call)", "date": datetime.date.today().isoformat()},
           "command": cmd, "build": tag, "kernel_src_sha16": kernel_src_sha16(), "config": int(cfg), "workload": m2.group(1) if m2 else "synthetic", "corpus_gib": float(m3.group(1)) if m3 else 4.0,
           "frames_per_gpu": frames, "frame_bytes": int(m.group(1)) if m else 131072, "kernels": issue},
          open(os.path.join(ROOT, "gpurun_out", f"{tag}_issue_cfg{cfg}.json"), "w"), indent=1)
print(json.dumps(issue))
