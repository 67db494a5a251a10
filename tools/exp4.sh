cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stages.py -x -q -k "huf_seg or oracle_trace or config3" 2>&1 | tail -5
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for l in 0 53000 80000; do MZD_HUF_SEG_LDS=$l python bench.py --config 3 --cpu-seconds 0 --no-ceiling 2>/dev/null | pick "cfg3 seg lds=$l"; done
python bench.py --cpu-seconds 0 --no-ceiling --huf-variant 2 2>/dev/null | pick "cfg4 seg"
