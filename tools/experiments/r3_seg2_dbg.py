import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb
blob, off, ln, ck, ns = sb.make_batch(3, 5, 4, frame_bytes=131072, threads=2)
frames = [blob[int(o):int(o + l)].tobytes() for o, l in zip(off, ln)]
c1 = z.Context(0, huf_variant=1)
c2 = z.Context(0, huf_variant=2)
o1, s1 = z.decode_frames(frames, c1)
o2, s2 = z.decode_frames(frames, c2)
print(s1, s2)
for a, b in zip(o1, o2):
    a = np.frombuffer(a, np.uint8); b = np.frombuffer(b, np.uint8) if b is not None else None
    if b is None: print("none"); continue
    bad = np.nonzero(a != b)[0]
    print(len(a), len(bad), bad[:40].tolist())
    if len(bad):
        i = int(bad[0]); print(a[i-8:i+24].tolist()); print(b[i-8:i+24].tolist())
a = np.frombuffer(o1[0], np.uint8); b = np.frombuffer(o2[0], np.uint8)
neq = (a != b).astype(np.int8)
d = np.diff(np.concatenate([[0], neq, [0]]))
starts = np.nonzero(d == 1)[0]; ends = np.nonzero(d == -1)[0]
print([(int(s), int(e - s)) for s, e in zip(starts[:60], ends[:60])])
i = 0
print(a[:80].tolist()); print(b[:80].tolist())
