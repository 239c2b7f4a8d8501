#!/bin/bash
# Round 6, second GPU call: default bench (budget, leg_seconds, OpenBLAS baseline), the rehearsals, the Fisher sweep, the free-running report
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
echo "== default bench run (budget, leg_seconds)"
time python bench.py > $O/s2_bench_default.json 2> $O/s2_bench_default.err; echo "rc $?"; tail -2 $O/s2_bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/s2_bench_default.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["legs_skipped"], json.dumps(d["budget"]))
c = d["cpu_baseline"]
print("cpu", c["value"] if c else None, "blas", json.dumps(c.get("blas")) if c else None, "host", list((d["host_caller"] or {}).keys()))
PY
echo "== 3-rank rehearsal under a 75 s budget"
time BENCH_BUDGET_S=75 python bench.py --gpus 3 --rehearse > $O/s2_rehearse_small_budget.json 2> $O/s2_rehearse_small_budget.err; echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/s2_rehearse_small_budget.json"))
print(d["n_gpus"], d["value"], d["legs_skipped"], list(d.get("legs", {}).keys()), d["legs_failed"], json.dumps(d["budget"]))
PY
echo "== 3-rank rehearsal, default budget"
time python bench.py --gpus 3 --rehearse > $O/s2_rehearse.json 2> $O/s2_rehearse.err; echo "rc $?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/s2_rehearse.json"))
print(d["n_gpus"], d["value"], d["legs_skipped"], list(d.get("legs", {}).keys()), d["legs_failed"], json.dumps(d["budget"]))
PY
echo "== Fisher pass 1 sweep on C4 (fu = 128): workgroups per CU x lag"
for opts in fisher_split=0 fisher_split=1,fisher_split_per_cu=1 fisher_split=1,fisher_split_per_cu=2,fisher_lag=0 fisher_split=1,fisher_split_per_cu=2,fisher_lag=2 fisher_split=1,fisher_split_per_cu=2,fisher_lag=8 fisher_split=1,fisher_split_per_cu=2,fisher_lag=64 fisher_split=1,fisher_split_per_cu=3; do
	SQN_OPTS=$opts C4_QUICK=1 python tools/bench_configs.py c4 > $O/s2_c4_one.json 2> $O/s2_c4.err || tail -3 $O/s2_c4.err
	python - $opts $O/s2_c4_one.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().splitlines()[0])
k = d["kernels"]
print(sys.argv[1], d["steps_per_s"], {n: (k[n]["avg_ms"], k[n].get("frac_of_8TBps")) for n in ("fisher_t", "fisher_y") if n in k}, flush=True)
PY
	cat $O/s2_c4_one.json >> $O/s2_c4_sweep.jsonl
done
echo "== free-running: device against oracle"
python tools/free_run_report.py > $O/free_run_report.json 2> $O/free_run_report.err; echo "rc $?"; tail -4 $O/free_run_report.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/free_run_report.json"))
print(json.dumps(d["worst_free_running_rel_err_by_config"], indent=1))
PY
