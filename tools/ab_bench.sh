#!/bin/bash
# same-box A/B of two builds of libmzd.so: put them at tmp_ab/libmzd_old.so and tmp_ab/libmzd_new.so (tmp_ab/ is git-ignored but travels to the GPU box), then gpurun -- bash tools/ab_bench.sh.  Run-to-run noise on one box is +-0.3 ms.
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in old new; do
    cp tmp_ab/libmzd_$v.so sparkzstd_amd/libmzd.so
    echo -n "$v "; timeout 600 python bench.py --cpu-seconds 0 --steps 10 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms']['k_seq'], d['roofline']['kernel_ms']['k_exec'], d['bit_exact'])"
  done
done
