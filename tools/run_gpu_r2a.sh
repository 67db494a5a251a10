cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r2a_pytest.log 2>&1; tail -5 gpurun_out/r2a_pytest.log
for c in 2 3 4; do timeout 600 python bench.py --config $c > gpurun_out/r2a_cfg$c.json 2> gpurun_out/r2a_cfg$c.err; tail -c 3000 gpurun_out/r2a_cfg$c.json; echo; done
