#!/bin/bash
# Round 6: batched totals in the prologues of passes 2 and 3 (fold_batch) -- bit identity, then interleaved A/B at C2, n = 1e6, C3
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "batched_prologue or two_loop_matches or steps_match_the_oracle_at_full_size" > $O/s11_tests.log 2>&1; echo "rc $?"; tail -3 $O/s11_tests.log
echo "== C2 interleaved"
for rep in 1 2 3; do for v in 0 1; do
	SQN_OPTS=fold_batch=$v python tools/bench_configs.py c2 > $O/s11_c2_one.json 2> $O/s11_c2.err || tail -3 $O/s11_c2.err
	python - "fold_batch=$v" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06/s11_c2_one.json").read().splitlines()[0])
k = d["kernels"]
print(sys.argv[1], d["steps_per_s"], d["step"]["frac_of_8TBps"], {n: k[n]["avg_ms"] for n in ("pair_y_diff", "apply", "sdot2", "qdot", "sadd") if n in k}, flush=True)
PY
	cat $O/s11_c2_one.json >> $O/s11_c2_fold_batch.jsonl
done; done
echo "== small n: tools/latency (C caller), oLBFGS and SQN at n = 1e5, 1e6"
if [ -x tools/latency ]; then for v in 0 1; do for n in 100000 1000000; do echo "fold_batch=$v n=$n"; STOCHQN_HIP_OPTS=fold_batch=$v tools/latency olbfgs $n 10 300 2>&1 | tail -2; done; done; fi
echo "== C3 interleaved"
for rep in 1 2; do for v in 0 1; do
	python bench.py --steps 100 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --opt fold_batch=$v > $O/s11_one.json 2> $O/s11.err || tail -3 $O/s11.err
	python - "fold_batch=$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06/s11_one.json"))
k = d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["two_loop"]["ms"], {n: round(k[n]["avg_ms"], 4) for n in ("sdot", "qdot", "sadd", "apply") if n in k}, flush=True)
PY
done; done
