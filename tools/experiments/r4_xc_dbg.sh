echo "== default (OOR 0x00FF0000)"; timeout 600 python tools/xc_debug.py 8 2>&1 | tail -30
echo "== exec-masked stores"; MZD_LIB=$PWD/tmp_ab/libmzd_nooor.so timeout 600 python tools/xc_debug.py 8 2>&1 | tail -30
