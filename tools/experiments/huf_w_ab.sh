#!/bin/bash
# GPU box: same-box A/B of k_huf_w builds on config 3 (4 096 and, with "full", 65 536 frames) and config 4 with the kernel forced.
# usage: tools/experiments/huf_w_ab.sh [full] <lib tag> ...   (tmp_ab/libmzd_<tag>.so; "shipped" = sparkzstd_amd/libmzd.so)
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
FULL=0; [ "$1" = full ] && { FULL=1; shift; }
B="python3 bench.py --cpu-seconds 0 --no-ceiling --no-secondary --steps 10 --warmup 2"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"].get("k_huf"), d["bit_exact"])'
for rep in 1 2; do
for tag in "$@"; do
  lib=$R/tmp_ab/libmzd_$tag.so; [ "$tag" = shipped ] && lib=$R/sparkzstd_amd/libmzd.so
  echo "== $tag: config 3 x 4096 | config 4 forced"
  MZD_LIB=$lib $B --config 3 2>/dev/null | python3 -c "$P"
  [ $FULL = 1 ] && MZD_LIB=$lib $B --config 3 --frames 65536 --steps 5 2>/dev/null | python3 -c "$P"
  MZD_LIB=$lib $B --huf-variant 4 --frames 16384 2>/dev/null | python3 -c "$P"
done
done
