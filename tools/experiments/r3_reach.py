#!/usr/bin/env python3
"""How far back do a block's matches reach over its start?  (Block mode spells a copied position with 3 or 4 passes; a block
whose matches stay within 8 MiB of its start would need only three.)  usage: python tools/experiments/r3_reach.py [MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 128
data = sb.generate(sb.TEXT, 5, mib << 20)
comp = sb.compress(data, sb.MODE_FULL)[0]
plan = z.Plan(device_tables=True)
plan.add_frame(comp)
ctx = z.Context(0, exec_variant=2)
rb = ctx.upload(plan.finalize())
rb.run(); ctx.sync()
nb = rb.stats().n_blocks[0] + rb.stats().n_blocks[1] + rb.stats().n_blocks[2]
blocks = rb.debug_blocks(int(nb))
pos = 0
reach = []
far_frac = []
for b in blocks:
    if b.type != 2 or b.n_seq == 0:
        pos += b.size if b.type != 2 else b.lit_regen
        reach.append(0); continue
    rec = rb.debug_read(_lib.MZD_DEBUG_RECORDS, np.uint64, int(b.rec_off) * 8, int(b.n_seq))
    ll = (rec & np.uint64(0x1FFFF)).astype(np.int64)
    ml = ((rec >> np.uint64(17)) & np.uint64(0x3FFFF)).astype(np.int64)
    off = ((rec >> np.uint64(35)) & np.uint64(0x1FFFFFFF)).astype(np.int64)
    sym = (off & (1 << 28)) != 0
    off = np.where(sym, 1, off)  # (symbolic repeat offsets at a block's start: short reach; ignored here)
    mstart = pos + np.cumsum(ll + ml) - ml
    src = mstart - off
    d = np.where(src < pos, pos - src, 0)
    reach.append(int(d.max()))
    far_frac.append(float((d > (1 << 23) - (1 << 17)).mean()))
    pos += int((ll + ml).sum()) + (b.lit_regen - int(ll.sum()))
reach = np.array(reach)
lim = (1 << 23) - (1 << 17)
print(f"{mib} MiB frame, {len(reach)} blocks: blocks whose matches reach back more than {lim} bytes over their start: {(reach > lim).sum()} "
      f"({100.0 * (reach > lim).mean():.1f} %); median reach {int(np.median(reach))}, max {int(reach.max())}; sequences beyond the limit per block: {100 * np.mean(far_frac):.3f} %")
