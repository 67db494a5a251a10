#!/usr/bin/env python3
"""GPU box: lower bound of a POINTER-JUMPING sequence executor on real match graphs -- tools/ubench/pj_resolve.hip
resolves 16 KiB output tiles built here from the oracle's sequence trace of BASELINE config-4 frames and is checked byte for byte
against the frames' content.  Prints the time per tile, what that makes for the 65 536-frame batch, and the depth statistics.
usage: python tools/pj_resolve.py [frames=64] [reps=40]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tools import synth_binding as sb
from tests.oracle_binding import load_oracle

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
TILE = 16384
so = os.path.join(ROOT, "tools", "ubench", "libpj_resolve.so")
if not os.path.exists(so):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "ubench", "pj_resolve.hip")])
L = ctypes.CDLL(so)
L.pj_resolve_run.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_uint32] * 3 + [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
orc = load_oracle()
blob, off, ln, cks, _ = sb.make_batch(4, 0, n_frames, 131072, threads=os.cpu_count() or 8)
src_tiles, val_tiles, want_tiles, far_frac = [], [], [], []
for f in range(n_frames):
    comp = bytes(blob[int(off[f]):int(off[f] + ln[f])])
    rc, out, _, tr = orc.decode_frame(comp, cap=131072 + 64, want_trace=True)
    assert rc == 0 and len(out) == 131072
    outb = np.frombuffer(out, dtype=np.uint8)
    # absolute source of every output byte (-1: a literal), from the trace's (LL, ML, offset value, resolved offset)
    src_abs = np.full(131072, -1, dtype=np.int64)
    pos = 0
    for ll, ml, _, roff in tr["seqs"]:
        pos += ll
        if ml:
            src_abs[pos:pos + ml] = np.arange(pos - roff, pos - roff + ml)
            pos += ml
    p = np.arange(131072)
    for t0 in range(0, 131072, TILE):
        loc = p[t0:t0 + TILE] - t0
        s = src_abs[t0:t0 + TILE] - t0
        root = (src_abs[t0:t0 + TILE] < t0)  # literals (-1) and sources before the tile
        s16 = np.where(root, loc, s).astype(np.uint16)
        src_tiles.append(s16)
        val_tiles.append(np.where(root, outb[t0:t0 + TILE], 0).astype(np.uint8))
        want_tiles.append(outb[t0:t0 + TILE])
        far_frac.append(float(((src_abs[t0:t0 + TILE] >= 0) & (src_abs[t0:t0 + TILE] < t0)).mean()))
src = torch.from_numpy(np.concatenate(src_tiles).view(np.int16)).cuda()
val = torch.from_numpy(np.concatenate(val_tiles)).cuda()
out = torch.zeros_like(val)
n_tiles = len(src_tiles)
rounds = torch.zeros(n_tiles, dtype=torch.int32, device="cuda")
ms = ctypes.c_float()


def run(r, max_rounds):
    rc = L.pj_resolve_run(src.data_ptr(), val.data_ptr(), out.data_ptr(), n_tiles, r, max_rounds, rounds.data_ptr(), ctypes.byref(ms))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return ms.value

t_full = run(reps, 32)
ok = bool((out.cpu().numpy() == np.concatenate(want_tiles)).all())
rr = rounds.cpu().numpy()
t_copy = run(reps, 0)  # the LDS-to-LDS copy of the pristine tile + the tile's store alone
cus = torch.cuda.get_device_properties(0).multi_processor_count
per_tile_us = (t_full - t_copy) / reps * 1e3 / max(1.0, n_tiles / cus)  # a workgroup (96 KiB of LDS) per CU at a time
tiles_batch = 65536 * 8
print(f"{n_tiles} tiles of {TILE} bytes from {n_frames} config-4 frames, {reps} repetitions: bit-exact {ok}")
print(f"rounds of pointer doubling per tile: mean {rr.mean():.1f}, max {rr.max()}; far-match bytes handed in as roots: {np.mean(far_frac) * 100:.0f} % of a tile")
print(f"kernel {t_full:.3f} ms, its LDS copy + store alone {t_copy:.3f} ms -> resolve + gather {per_tile_us:.2f} us per tile on one CU")
print(f"the 65 536-frame batch = {tiles_batch} tiles over {cus} CUs: {tiles_batch / cus * per_tile_us / 1e3:.2f} ms for the resolve step alone "
      f"(k_exec_c, the whole executor with its far-match traffic: 9.44 ms)")
