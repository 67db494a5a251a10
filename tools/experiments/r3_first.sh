# round 3: k_exec_b on the GPU: corpus parity, then the config-4 pass with either execution kernel
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_corpus.py -x -q -k "decodecorpus or k_exec_b" 2>&1 | tail -15
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
for v in ${VARIANTS:-2 1}; do
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --exec-variant $v 2>gpurun_out/err_$v.log | pick "cfg4 exec_variant=$v"
  timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 5 --exec-variant $v --no-split 2>>gpurun_out/err_$v.log | pick "cfg4 nosplit exec_variant=$v"
done
tail -5 gpurun_out/err_2.log
[ -f tmp_ab/libmzd_xbstats.so ] && MZD_LIB=$PWD/tmp_ab/libmzd_xbstats.so timeout 300 python tools/xb_stats.py 16384 2>&1 | grep -v amdgpu.ids
