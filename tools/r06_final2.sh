#!/bin/bash
# Round 6, the very last GPU call: the other configurations with the final library (the float build changed after tools/r06_session5.sh),
# the driver's bench command once more, the whole GPU suite
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
python tools/bench_configs.py c3f32 c3host rccl1 > $O/other_configs_final.jsonl 2> $O/other_configs_final.err; echo "rc $?"; grep '^{' $O/other_configs_final.jsonl | cut -c1-200
time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/final4_bench_driver_like.json 2> $O/final4_bench_driver_like.err; echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/final4_bench_driver_like.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["legs_skipped"], d["budget"]["used_s"])
PY
time python -m pytest tests -x -q -m gpu > $O/final4_gpu_suite.log 2>&1; echo "rc $?"; tail -3 $O/final4_gpu_suite.log
