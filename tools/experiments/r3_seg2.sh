# round 3: k_huf_seg keeps the count pass's symbols (no second walk): config 3 over approach run / segment length / strip stride
# v2 128/384/35  v5 128/256/35  v6 96/384/35  v7 128/320/35  v8 96/320/35  v9 160/384/35  v10 128/384/33
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in v2 v5 v6 v7 v8 v9 v10; do
for i in 1 2; do MZD_LIB=tmp_ab/libmzd_$lib.so timeout 300 python bench.py --config 3 --cpu-seconds 0 --no-ceiling 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 $lib', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])"; done; done
