# round 3 evidence: the GPU suite, then rocprofv3 kernel stats + FETCH / WRITE traffic + SQ counters + the bench line (quoting
# them) for BASELINE config 4, the real-data workload and the 8192-frame shard of configs[4]; configs 2 and 3 bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r3_pytest_gpu.log
bash tools/profile_counters.sh r3 4 2>&1 | tail -2
bash tools/profile_round.sh r3 4 --issue-from gpurun_out/r3_issue_cfg4.json 2>&1 | tail -1 | cut -c1-400
bash tools/profile_counters.sh r3corpus 4 --workload corpus 2>&1 | tail -2
bash tools/profile_round.sh r3corpus 4 --workload corpus --issue-from gpurun_out/r3corpus_issue_cfg4.json 2>&1 | tail -1 | cut -c1-400
bash tools/profile_round.sh r3s8192 4 --frames 8192 2>&1 | tail -1 | cut -c1-300
for c in 2 3; do timeout 300 python bench.py --config $c 2>/dev/null | tee gpurun_out/r3_cfg${c}_bench.json | cut -c1-200; done
ls gpurun_out | head -50
