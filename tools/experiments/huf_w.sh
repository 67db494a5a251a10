#!/bin/bash
# GPU box: k_huf_w (round 6) -- parity of the Huffman suites, its phase cycles (tmp_ab/libmzd_hwstats.so: -DMZD_HUF_W_STATS), then
# config 3 at both sizes and config 4 with the Huffman kernel forced (--huf-variant 4) against the library's choice.
# usage: tools/experiments/huf_w.sh [quick]
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -m pytest tests/test_gpu_stages.py -m gpu -x -q -k "huf or literals_and_sequences or config3" 2>&1 | tail -4
if [ -f tmp_ab/libmzd_hwstats.so ]; then
  MZD_LIB=$R/tmp_ab/libmzd_hwstats.so python3 tools/huf_w_stats.py 3 2048 4 2>&1 | grep -v amdgpu.ids
  MZD_LIB=$R/tmp_ab/libmzd_hwstats.so python3 tools/huf_w_stats.py 4 8192 4 2>&1 | grep -v amdgpu.ids
fi
B="python3 bench.py --cpu-seconds 0 --no-ceiling --no-secondary --steps 10 --warmup 2"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["roofline"]["kernel_ms"], d["bit_exact"])'
for hv in 2 0; do
  echo "== config 3, 4096 frames, huf_variant $hv"; $B --config 3 --huf-variant $hv 2>/dev/null | python3 -c "$P"
  [ "$1" = quick ] || { echo "== config 3, 65536 frames, huf_variant $hv"; $B --config 3 --frames 65536 --steps 5 --huf-variant $hv 2>/dev/null | python3 -c "$P"; }
done
for hv in 0 4; do
  echo "== config 4, huf_variant $hv"; $B --huf-variant $hv 2>/dev/null | python3 -c "$P"
done
# ablations (timing only): the kernel without its stores / without the line touches
for lib in hw_nostore hw_notouch; do
  [ -f tmp_ab/libmzd_$lib.so ] && { echo "== config 3, 4096 frames, $lib"; MZD_LIB=$R/tmp_ab/libmzd_$lib.so $B --config 3 --no-verify 2>/dev/null | python3 -c "$P"; }
done
