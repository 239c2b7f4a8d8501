#!/bin/bash
# Round 6, third GPU call: tune16 (C2 as one launch), the default bench again (CPU thread placement, OpenBLAS leg), the n = 1e8
# adaQN free-running distances, the bench control-flow tests (budget, watchdog, every leg)
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
echo "== tune16: the C2 step as one cooperative launch"
(cd profiles/src && timeout -k 10 300 ./tune16 10000000 > ../../$O/tune16_c2.log 2>&1; echo "rc $?"; timeout -k 10 120 ./tune16 1000000 >> ../../$O/tune16_c2.log 2>&1; echo "rc $?")
cat $O/tune16_c2.log
echo "== default bench run"
time python bench.py > $O/s3_bench_default.json 2> $O/s3_bench_default.err; echo "rc $?"; tail -2 $O/s3_bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/s3_bench_default.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["legs_skipped"], json.dumps(d["budget"]["leg_seconds"]), d["budget"]["used_s"])
c = d["cpu_baseline"]
print("cpu", c["value"], c["allcores_cycles_s"], c["omp"], "blas", json.dumps(c.get("blas")))
PY
echo "== adaQN at n = 1e8: free-running distances (printed by the tests)"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "adaqn_trajectory_at_full_size or (full_size_steps_agree and adaQN)" > $O/s3_adaqn_full_size.log 2>&1; echo "rc $?"; grep -E "free-running|passed|failed|agree" $O/s3_adaqn_full_size.log | tail -5
echo "== bench control flow tests"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inside_its_budget or auxiliary_leg_hangs or default_multi_gpu or multi_process_control_flow" > $O/s3_bench_tests.log 2>&1; echo "rc $?"; tail -5 $O/s3_bench_tests.log
