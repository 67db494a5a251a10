# k_huf_seg: how long must the approach run be?  config-3 pass with -DMZD_SEG_APPROACH=n builds (tmp_ab/libmzd_seg<n>.so)
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 200 python bench.py --config 3 --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "approach 256 (default)"
for a in 128 96; do
  MZD_LIB=$PWD/tmp_ab/libmzd_seg$a.so timeout 200 python bench.py --config 3 --cpu-seconds 0 --no-ceiling --steps 10 2>/dev/null | pick "approach $a"
done
