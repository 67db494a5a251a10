cd $GRAFT_REPO_ROOT
for t in segst2w segst33 segst45; do echo $t; MZD_LIB=tmp_ab/libmzd_$t.so timeout 300 python tools/huf_seg_stats.py 3 4096 2>&1 | tail -2; done
