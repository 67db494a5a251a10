cd $GRAFT_REPO_ROOT
export MZD_DEBUG_SEQ_ONLY=1 MZD_PROF_IGNORE_STATUS=1
for n in fastbc abl_NOWAIT abl_NORING abl_NOQW abl_NORINGDMZD_ABL_NOQW abl_NORINGDMZD_ABL_NOQWDMZD_ABL_NOWAIT; do MZD_LIB=$PWD/tmp_ab/libmzd_$n.so timeout 200 python tools/abl.py $n 2>&1 | tail -1; done
