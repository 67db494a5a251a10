# round 3: fix-up workgroups per frame (= pollers of the frame's counter) for 64 frames of 128 MiB (experiments build, MZD_EXP_BLK_G)
cd $GRAFT_REPO_ROOT
pick() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d['ms_per_step'], d['value'], d['roofline']['kernel_ms'], d['bit_exact'])" "$1"; }
export MZD_LIB=tmp_ab/libmzd_exp.so
for g in 2 4 8 16; do MZD_EXP_BLK_G=$g timeout 900 python bench.py --cpu-seconds 0 --no-ceiling --steps 2 --warmup 1 --frames 64 --frame-bytes 134217728 --gen-seconds 200 2>/dev/null | pick "64 x 128 MiB, $g fix-up workgroups per frame"; done
