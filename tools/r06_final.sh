#!/bin/bash
# Round 6, last GPU call: smoke(), the bench as the driver starts it (N = 1), the 3-rank rehearsals (default and small budget), the whole GPU suite
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== python3 bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command)"
time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench_driver_like.json 2> $O/final_bench_driver_like.err; echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/final_bench_driver_like.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["legs_skipped"], d["budget"]["used_s"], json.dumps(d["budget"]["leg_seconds"]))
c = d["cpu_baseline"]; print("cpu", c["value"], c["omp"]["placement"], "blas", (c["blas"] or {}).get("value"))
PY
echo "== python bench.py (default K)"
time python bench.py > $O/final_bench_default.json 2> $O/final_bench_default.err; echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/final_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["legs_skipped"], d["budget"]["used_s"])
PY
echo "== 3-rank rehearsal, default budget"
time python bench.py --gpus 3 --rehearse > $O/final_rehearse.json 2> $O/final_rehearse.err; echo "rc $?"
echo "== 3-rank rehearsal, BENCH_BUDGET_S=45"
time BENCH_BUDGET_S=45 python bench.py --gpus 3 --rehearse > $O/final_rehearse_small_budget.json 2> $O/final_rehearse_small_budget.err; echo "rc $?"
python - <<'PY'
import json
for f in ("final_rehearse", "final_rehearse_small_budget"):
    d = json.load(open("gpurun_out/r06/%s.json" % f))
    print(f, d["n_gpus"], d["value"], d["legs_skipped"], list(d.get("legs", {}).keys()), d["legs_failed"], d["budget"]["used_s"], json.dumps(d["budget"]["leg_seconds"]))
PY
echo "== the whole GPU suite"
time python -m pytest tests -x -q -m gpu > $O/final_gpu_suite.log 2>&1; echo "rc $?"; tail -3 $O/final_gpu_suite.log
