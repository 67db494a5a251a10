# round 3: block mode on frames whose blocks DO reach back (the generator's matcher keeps its table over a frame): fix-up workgroups
# per frame and blocks per job (experiments build: MZD_EXP_BLK_G, MZD_EXP_BLK_GS)
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['kernel_ms']['k_exec'], d['bit_exact'])" "$1"; }
export MZD_LIB=tmp_ab/libmzd_exp.so
for g in 8 16 32; do MZD_EXP_BLK_G=$g timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB, $g fix-up workgroups per frame"; done
for s in "32 1" "32 2" "64 1" "64 2" "128 1"; do set -- $s; MZD_EXP_BLK_G=$1 MZD_EXP_BLK_GS=$2 timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 1 --frame-bytes 1073741824 --gen-seconds 200 2>/dev/null | pick "1 x 1 GiB, $1 workgroups, jobs of $2"; done
