#!/usr/bin/env python3
"""Runs ON THE GPU BOX right after the rocprofv3 passes of tools/profile_round.sh: turns
gpurun_out/<tag>_cfg<N>_{stats,fetch,write} into
  gpurun_out/<tag>_cfg<N>_kernel_stats.csv   the rocprofv3 --stats summary (per-kernel calls / average ns)
  gpurun_out/<tag>_traffic_cfg<N>.json       FETCH_SIZE / WRITE_SIZE per kernel and per bench step, stamped with
                                             the hash of the device sources and the workload they were measured on
                                             (bench.py quotes the file only when both match its own run).
usage: summarize_profile.py <tag> <config> <launched steps incl. warmup> "<profiled command>" """
import collections, csv, datetime, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_src_sha16  # noqa: E402

tag, cfg, steps, cmd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
go = os.path.join(ROOT, "gpurun_out")
base = f"{tag}_cfg{cfg}"

st = glob.glob(os.path.join(go, base + "_stats", "*", "*_kernel_stats.csv"))
if st:
    rows = list(csv.reader(open(st[0])))
    with open(os.path.join(go, base + "_kernel_stats.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}\n")
        f.write(f"# {steps} passes of the hot path per run (warm-up included); device sources {kernel_src_sha16()}\n")
        w = csv.writer(f)
        for r in rows:
            w.writerow([c[:110] for c in r])


def per_kernel(kind):
    fs = glob.glob(os.path.join(go, f"{base}_{kind}", "*", "*_counter_collection.csv"))
    agg, calls = collections.Counter(), collections.Counter()
    if not fs:
        return None, None
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"mzd::(k_\w+)", r["Kernel_Name"])
        if m:
            agg[m.group(1)] += float(r["Counter_Value"])
            calls[m.group(1)] += 1
    return agg, calls


fetch, calls = per_kernel("fetch")
write, _ = per_kernel("write")
if fetch is not None and write is not None:
    frames = {2: 4096, 3: 4096, 4: 65536}.get(cfg)
    m = re.search(r"--frames(?:-per-gpu)? (\d+)", cmd)
    if m:
        frames = int(m.group(1))
    m = re.search(r"--frame-bytes (\d+)", cmd)
    frame_bytes = int(m.group(1)) if m else 131072
    m = re.search(r"--workload (\w+)", cmd)
    workload = m.group(1) if m else "synthetic"
    m = re.search(r"--corpus-gib ([\d.]+)", cmd)
    corpus_gib = float(m.group(1)) if m else 4.0
    kern = {}
    for k in sorted(set(fetch) | set(write)):
        if k in ("k_fse_build", "k_huf_build", "k_parse", "k_copy_ceiling"):
            continue  # once per upload / measurement utility: not part of a pass
        fb = int(fetch[k] / steps * 1024)
        kern[k] = {"fetch_bytes": fb, "fetch_bytes_x2": 2 * fb, "write_bytes": int(write[k] / steps * 1024),
                   "launches_per_pass": round(calls[k] / steps, 2)}
    out = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (kernel-trace only); totals per pass of the hot "
                "path (a kernel launched twice per pass -- head + tail of a split batch -- is summed). Counter unit is KB (TCC_EA0 "
                "requests x 64 B / 1024). Per /opt/skills/guides/MI355X_MICROARCH.md the gfx950 FETCH_SIZE under-counts wide coalesced "
                "streaming reads by 2x and is uncalibrated for other access widths; Infinity-Cache hits are included. bytes = KB * "
                "1024; fetch_bytes_x2 applies the guide's streaming correction as an upper bound.",
        "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- {cmd}",
        "measured": {"by": "the builder (tools/profile_round.sh inside a gpurun call)", "date": datetime.date.today().isoformat()},
        "build": tag, "kernel_src_sha16": kernel_src_sha16(), "config": cfg, "workload": workload, "corpus_gib": corpus_gib, "frames_per_gpu": frames,
        "frame_bytes": frame_bytes, "kernels": kern,
    }
    json.dump(out, open(os.path.join(go, f"{tag}_traffic_cfg{cfg}.json"), "w"), indent=1)
    print(json.dumps(kern))
