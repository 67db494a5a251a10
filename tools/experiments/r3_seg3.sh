# round 3: k_huf_seg with the round's bytes transposed through LDS before they are stored (against -DMZD_SEG_NO_TRANSPOSE)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "huf_seg or config3 or literals_and_sequences" 2>&1 | tail -3
for lib in "" tmp_ab/libmzd_notr.so; do for i in 1 2; do MZD_LIB=$lib timeout 300 python bench.py --config 3 --cpu-seconds 0 --no-ceiling 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 $lib', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])"; done; done
MZD_LIB=tmp_ab/libmzd_v2s.so timeout 300 python tools/huf_seg_stats.py 3 4096 2>&1 | tail -2
