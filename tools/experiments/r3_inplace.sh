cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in 3 4; do timeout 300 python bench.py --config $c --cpu-seconds 0 --no-ceiling 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$c', d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])"; done
