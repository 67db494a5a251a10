#!/usr/bin/env python3
"""Block mode in K slices (the walk of slice s beside the passes of slice s + 1): ONE generated batch per shape, every K on it.
Needs a library built with -DMZD_EXPERIMENTS (MZD_LIB=...): K and the fused / per-pass launches come from the environment.
usage: blk_slices.py "frames:frame_bytes[:window_log]" ... [--ks 1,2,4,8] [--fused default,0,1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import sparkzstd_amd as z
from tools import synth_binding as sb
import bench

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opt = dict(a[2:].split("=", 1) for a in sys.argv[1:] if a.startswith("--") and "=" in a)
ks = [int(k) for k in opt.get("ks", "1,2,4,8").split(",")]
fus = opt.get("fused", "default").split(",")
gws = opt.get("g", "default").split(",")  # fix-up workgroups per frame
threads = max(1, bench.usable_cores()[0])
for shape in args:
    parts = shape.split(":")
    n, fb = int(parts[0]), int(parts[1])
    sb.set_max_offset((1 << int(parts[2])) if len(parts) > 2 else 0)
    t0 = time.perf_counter()
    distinct = n
    blob, off, ln, cks, nseq = sb.make_batch(4, 0, distinct, fb, threads=threads)
    plan = z.Plan(device_tables=True)
    assert plan.add_frames(blob, off, ln, threads=threads) == 0
    batch = plan.finalize()
    d_in = torch.zeros(blob.size + 128, dtype=torch.uint8, device="cuda")
    d_in[64:64 + blob.size].copy_(torch.from_numpy(blob))
    d_out = torch.zeros(batch.out_size, dtype=torch.uint8, device="cuda")
    ctx = z.Context(0)
    rb = ctx.upload(batch, device_in_ptr=d_in.data_ptr() + 64, device_out_ptr=d_out.data_ptr())
    torch.cuda.synchronize()
    print(f"# {n} x {fb >> 20} MiB{' window log ' + parts[2] if len(parts) > 2 else ''}: generated + planned + uploaded in {time.perf_counter() - t0:.0f} s", flush=True)
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        for f, gw in [(f, gw) for f in fus for gw in gws]:
            for k in ks:
                os.environ["MZD_EXP_BLK_SLICES"] = str(k)
                if gw == "default":
                    os.environ.pop("MZD_EXP_BLK_G", None)
                else:
                    os.environ["MZD_EXP_BLK_G"] = gw
                if f == "default":
                    os.environ.pop("MZD_EXP_BLK_FUSED", None)
                else:
                    os.environ["MZD_EXP_BLK_FUSED"] = f
                d_out.fill_(0xA5)
                rb.run(stream)
                torch.cuda.synchronize()
                ok = bench.verify_synth(torch, d_out, n, fb, cks)
                ctx.timing_reset(True)
                t = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    rb.run(stream)
                    torch.cuda.synchronize()
                    t.append((time.perf_counter() - t1) * 1e3)
                km = ctx.kernel_ms()
                print(f"{n} x {fb >> 20} MiB  slices {k} fused {f} G {gw}: {min(t):.2f} ms  (wall, best of 3; kernels {km})  bit-exact {ok}  flags {rb.last_pass()}", flush=True)
    rb.free()
    ctx.close()
    del d_in, d_out
    torch.cuda.empty_cache()
