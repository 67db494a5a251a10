cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'], d['roofline'].get('frac_of_copy_ceiling'))" "$1"; }
python bench.py --config 3 --cpu-seconds 0 2>/dev/null | pick "cfg3 approach256"
MZD_LIB=$PWD/tmp_ab/libmzd_hufstats128.so python bench.py --config 3 --cpu-seconds 0 2>/dev/null | pick "cfg3 approach128(stats build)"
MZD_LIB=$PWD/tmp_ab/libmzd_hufstats.so python tools/huf_seg_stats.py 3 512
MZD_LIB=$PWD/tmp_ab/libmzd_hufstats128.so python tools/huf_seg_stats.py 3 512
MZD_LIB=$PWD/tmp_ab/libmzd_hufstats.so python tools/huf_seg_stats.py 4 2048
MZD_LIB=$PWD/tmp_ab/libmzd_hufstats128.so python tools/huf_seg_stats.py 4 2048
