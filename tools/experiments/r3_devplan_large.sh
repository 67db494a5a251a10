# round 3: what the two planning routes cost for ONE large frame (not in the timed pass): host planner (one thread per frame) and
# device planning (k_parse: one lane per frame)
cd $GRAFT_REPO_ROOT
timeout 900 python - <<'PY'
import sys, time
sys.path.insert(0, '.')
import numpy as np
import sparkzstd_amd as z
from tools import synth_binding as sb
for mib in (64, 256):
    data = sb.generate(sb.TEXT, 5, mib << 20)
    comp = sb.compress(data, sb.MODE_FULL)[0]
    ctx = z.Context(0)
    for dp in (False, True):
        z.decode_frames([comp], ctx, device_plan=dp)  # warm
        t0 = time.time()
        outs, sts = z.decode_frames([comp], ctx, device_plan=dp)
        t1 = time.time()
        assert sts == [0] and outs[0] == data
        print(f"{mib} MiB frame, device_plan={dp}: {1e3 * (t1 - t0):.1f} ms host to host ({(mib << 20) / (t1 - t0) / 1e6:.0f} MB/s)", flush=True)
    ctx.close()
PY
