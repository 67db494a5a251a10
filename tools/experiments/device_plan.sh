# device planning block by block: its tests, then one 256 MiB frame through mzd_batch_upload_frames with the blob resident on the device
cd ${GRAFT_REPO_ROOT:-$PWD}
timeout 1200 python3 -m pytest tests -m gpu -x -q -k "device_planner or stream or upload_frames or device_plan" 2>&1 | tail -6
timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --warmup 1 --frames 1 --frame-bytes 268435456 --device-plan --gen-seconds 200 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('1 x 256 MiB device-planned', d['ms_per_step'], d['setup_s'], d['bit_exact'])"
timeout 600 python3 bench.py --cpu-seconds 0 --no-ceiling --steps 3 --device-plan 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('config 4 device-planned', d['ms_per_step'], d['setup_s'], d['bit_exact'])"
