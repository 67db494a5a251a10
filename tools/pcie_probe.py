#!/usr/bin/env python3
"""PCIe probe: D2H / H2D alone and concurrently on two streams (pinned memory), to read the streaming
numbers of tools/stream_bench.py against."""
import time, torch
n_out, n_in = 1 << 30, 375 << 20
h_out = torch.empty(n_out, dtype=torch.uint8).pin_memory()
h_in = torch.empty(n_in, dtype=torch.uint8).pin_memory()
d_out = torch.empty(n_out, dtype=torch.uint8, device="cuda")
d_in = torch.empty(n_in, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(do_out, do_in, reps=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if do_out:
            with torch.cuda.stream(s1):
                h_out.copy_(d_out, non_blocking=True)
        if do_in:
            with torch.cuda.stream(s2):
                d_in.copy_(h_in, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
run(True, True, 1)
a, b, c = run(True, False), run(False, True), run(True, True)
print(f"D2H 1 GiB alone {a:.2f} ms ({n_out / a / 1e6:.1f} GB/s)  H2D 375 MiB alone {b:.2f} ms ({n_in / b / 1e6:.1f} GB/s)  both {c:.2f} ms")
