# round 3: stability of the protocols that depend on timing (the fix-up walk's hand-offs): the block-mode tests over and over
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do timeout 900 python -m pytest tests/test_gpu_corpus.py -m gpu -x -q -k "blocks or segments or block_mode or large_frames or 0-0-3 or 0-0-4 or 0-3 or 0-4" 2>&1 | tail -1; done
timeout 1200 python - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import sparkzstd_amd as z
from tools import synth_binding as sb
frames, want = [], []
for kind, n in [(sb.TEXT, 48 << 20), (sb.TEXT, (33 << 20) + 777), (sb.EXP, 20 << 20)]:
    d = sb.generate(kind, n ^ 0x77, n)
    frames.append(sb.compress(d, sb.MODE_FULL)[0]); want.append(d)
for v in (0, 3, 4):
    c = z.Context(0, exec_variant=v)
    for rep in range(8):
        outs, sts = z.decode_frames(frames, c)
        assert sts == [0] * len(frames) and outs == want, (v, rep)
    c.close()
print("large frames x 3 variants x 8 repetitions: ok")
PY
