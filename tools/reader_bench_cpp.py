#!/usr/bin/env python3
"""Host-to-host throughput of the C++ mirror's BatchFrameReader (include/sparkzstd_frame.hpp, over mzd_stream_*): writes synthetic
frames to a scratch file and runs `tools/verify/sparkzstd_verify --bench` on it (framereader.go:35-109 consumer shape).
usage: python tools/reader_bench_cpp.py [distinct 128 KiB frames=2048] [frames served=32768]"""
import json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tools import synth_binding as sb

distinct = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
served = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
exe = os.path.join(ROOT, "tools", "verify", "sparkzstd_verify")
tmp = tempfile.mkdtemp(prefix="mzd_reader_")
blob, off, ln, cks, _ = sb.make_batch(4, 0, distinct, 131072, threads=os.cpu_count() or 8)
small = os.path.join(tmp, "frames_128k.zst")
np.ascontiguousarray(blob[:int(off[-1] + ln[-1])]).tofile(small)
big = os.path.join(tmp, "frame_256m.zst")
with open(big, "wb") as f:
    f.write(sb.compress(sb.generate(sb.TEXT, 9, 256 << 20), sb.MODE_FULL)[0])
rows = []
for path, n, look, mode in [(small, served, 256, "read"), (small, served, 1024, "read"), (small, served, 4096, "read"),
                            (small, served, 256, "view"), (small, served, 1024, "view"), (small, served, 4096, "view"), (small, 2 * served, 8192, "view"),
                            (big, 8, 1, "read"), (big, 8, 1, "view")]:
    r = subprocess.run([exe, "--bench", path, str(n), str(look), mode], capture_output=True, text=True, timeout=900)
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    row = json.loads(line[-1]) if line else {"error": (r.stdout + r.stderr)[-400:]}
    row["input"] = os.path.basename(path)
    rows.append(row)
    print(json.dumps(row), flush=True)
