# VERDICT r3 #1a: one workgroup owning a CU-sized share of LDS with the WHOLE 128 KiB block in it, 4-16 wavefronts cooperating
# on one frame (k_exec's dataflow executor, tiles dealt to the wavefronts; libmzd_ex512/ex1024 = -DMZD_EXEC_MAX_THREADS=512/1024)
pick() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', 'pass', d['ms_per_step'], d['roofline']['kernel_ms'], 'bit_exact', d.get('bit_exact'))"; }
run() { # lib threads chunk
  MZD_LIB=$PWD/tmp_ab/$1 timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --exec-variant 1 --no-split --exec-threads $2 --exec-chunk $3 2>/dev/null | pick "k_exec threads=$2 chunk=$3"
}
run libmzd_ex512.so 128 8192
run libmzd_ex512.so 256 32768
run libmzd_ex512.so 512 32768
run libmzd_ex512.so 512 65536
run libmzd_ex1024.so 512 131072
run libmzd_ex1024.so 1024 131072
run libmzd_ex1024.so 1024 65536
run libmzd_ex512.so 256 16384
