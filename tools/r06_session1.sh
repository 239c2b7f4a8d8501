#!/bin/bash
# Round 6, first GPU call: (1) the Fisher tests with the row-split pass 1, (2) interleaved A/B of fisher_split on C4 (fu = 128)
# and on the headline's Hessian-vector product (fu = 32), (3) the default bench run with the wall-clock budget (leg_seconds),
# (4) the 3-rank rehearsal under a small budget: one line, legs_skipped.
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
echo "== fisher / adaQN parity tests"; python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fisher or adaqn or adaQN" > $O/s1_pytest_fisher.log 2>&1; echo "rc $?"; tail -3 $O/s1_pytest_fisher.log
echo "== A/B fisher_split on C4 (fu = 128)"
for rep in 1 2; do for v in 0 1; do
	SQN_OPTS=fisher_split=$v C4_QUICK=1 python tools/bench_configs.py c4 > $O/s1_c4_split$v.json 2> $O/s1_c4.err || tail -3 $O/s1_c4.err
	python - $v $O/s1_c4_split$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().splitlines()[0])
k = d["kernels"]
print("fisher_split=%s" % sys.argv[1], d["steps_per_s"], {n: (k[n]["avg_ms"], k[n].get("frac_of_8TBps")) for n in ("fisher_t", "fisher_y", "qdot", "sadd") if n in k}, flush=True)
PY
	cat $O/s1_c4_split$v.json >> $O/s1_c4_ab.jsonl
done; done
echo "== A/B fisher_split on the headline's Hessian-vector product (fu = 32)"
for v in 0 1; do
	python bench.py --steps 40 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --opt fisher_split=$v > $O/s1_c3_split$v.json 2> $O/s1_c3.err || tail -3 $O/s1_c3.err
	python - $v $O/s1_c3_split$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
k = d["kernels"]
print("fisher_split=%s" % sys.argv[1], d["value"], {n: k[n] for n in ("fisher_t", "fisher_y") if n in k}, flush=True)
PY
done
echo "== default bench run (budget, leg_seconds)"
/usr/bin/time -v python bench.py > $O/s1_bench_default.json 2> $O/s1_bench_default.err; echo "rc $?"; grep -E "Elapsed|Maximum resident" $O/s1_bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/s1_bench_default.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["legs_skipped"], json.dumps(d["budget"]))
print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, "host", list((d["host_caller"] or {}).keys()))
PY
echo "== 3-rank rehearsal under a 75 s budget"
BENCH_BUDGET_S=75 /usr/bin/time -v python bench.py --gpus 3 --rehearse > $O/s1_rehearse_small_budget.json 2> $O/s1_rehearse_small_budget.err; echo "rc $?"; grep -E "Elapsed" $O/s1_rehearse_small_budget.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/s1_rehearse_small_budget.json"))
print(d["n_gpus"], d["value"], d["legs_skipped"], list(d.get("legs", {}).keys()), d["legs_failed"], json.dumps(d["budget"]))
PY
echo "== 3-rank rehearsal, default budget"
/usr/bin/time -v python bench.py --gpus 3 --rehearse > $O/s1_rehearse.json 2> $O/s1_rehearse.err; echo "rc $?"; grep -E "Elapsed" $O/s1_rehearse.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06/s1_rehearse.json"))
print(d["n_gpus"], d["value"], d["legs_skipped"], list(d.get("legs", {}).keys()), d["legs_failed"], json.dumps(d["budget"]))
PY
