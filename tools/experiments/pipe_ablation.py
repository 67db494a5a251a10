import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import sparkzstd_amd as z
from sparkzstd_amd import _lib
from tools import synth_binding as sb
n = 27648
blob, off, ln, ck, ns = sb.make_batch(4, 0, n, threads=16)
ctx = z.Context(0, no_split=True)
rb = ctx.upload_frames(blob[:int(off[-1] + ln[-1])], off, ln)
L = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
rb.run(); ctx.sync()
L.mzd_debug_pipe_stats(buf, 1)
ctx.timing_reset(True)
for _ in range(3): rb.run()
ctx.sync()
L.mzd_debug_pipe_stats(buf, 0)
wg, steps, cyc, qp, rp = (buf[i] / 3 for i in range(5))
print(f"{sys.argv[1]}: cycles/step {cyc / steps:.1f}, queue-full polls/batch {qp / (steps / 4):.3f}, ring polls/batch {rp / (steps / 4):.3f}, k_seq ms {ctx.kernel_ms()['k_seq']:.3f}")
