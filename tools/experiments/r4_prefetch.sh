# k_exec_c with the next block's descriptor and summary loaded a block ahead (shipped) against tmp_ab/libmzd_prev.so, same box
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
timeout 900 python -m pytest tests/test_gpu_corpus.py -x -q 2>&1 | tail -1
for rep in 1 2; do
  for lib in "" $PWD/tmp_ab/libmzd_prev.so; do
    if [ -z "$lib" ]; then unset MZD_LIB; tag=ahead; else export MZD_LIB=$lib; tag=before; fi
    timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --workload corpus 2>/dev/null | pick "corpus $tag"
    timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 2>/dev/null | pick "config 4 $tag"
  done
done
unset MZD_LIB
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --frames 8192 --frame-bytes 1048576 2>/dev/null | pick "8192 x 1 MiB ahead"
MZD_LIB=$PWD/tmp_ab/libmzd_prev.so timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 4 --frames 8192 --frame-bytes 1048576 2>/dev/null | pick "8192 x 1 MiB before"
timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --frames 131072 --frame-bytes 4096 2>/dev/null | pick "131072 x 4 KiB ahead"
MZD_LIB=$PWD/tmp_ab/libmzd_prev.so timeout 300 python bench.py --cpu-seconds 0 --no-ceiling --steps 6 --frames 131072 --frame-bytes 4096 2>/dev/null | pick "131072 x 4 KiB before"
