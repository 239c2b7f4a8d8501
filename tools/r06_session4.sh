#!/bin/bash
# Round 6, fourth GPU call: pair kernels' grid on C2, host callers' x_upload A/B, oracle vs oracle at n = 1e8, the tightened bars
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
echo "== C2: workgroups per CU of the pair kernels (interleaved)"
for rep in 1 2; do for v in 1 2 3 4; do
	SQN_OPTS=pair_per_cu=$v python tools/bench_configs.py c2 > $O/s4_c2_one.json 2> $O/s4_c2.err || tail -3 $O/s4_c2.err
	python - $v $O/s4_c2_one.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().splitlines()[0])
k = d["kernels"]
print("pair_per_cu=%s" % sys.argv[1], d["steps_per_s"], d["step"]["frac_of_8TBps"], {n: k[n]["avg_ms"] for n in ("pair_y_diff", "apply", "sdot2", "qdot", "sadd") if n in k}, flush=True)
PY
	cat $O/s4_c2_one.json >> $O/s4_c2_pair_grid.jsonl
done; done
echo "== host callers: x_upload 1 vs 2"
python tools/r06_host_upload_ab.py > $O/host_upload_ab.jsonl 2> $O/host_upload_ab.err; echo "rc $?"; tail -1 $O/host_upload_ab.jsonl; tail -3 $O/host_upload_ab.err
echo "== tests: arrays from stochqn_hip_alloc_host; every test on the configurations whose free-running bar was tightened"
python -m pytest tests/test_gpu_host_path.py -x -q -m gpu -k "handed_out" > $O/s4_tests_a.log 2>&1; echo "rc $?"; tail -2 $O/s4_tests_a.log
python -m pytest tests -x -q -m gpu -k "adaqn_ring25 or adaqn_ring20 or adaqn_trajectory_at_full_size" > $O/s4_tests_b.log 2>&1; echo "rc $?"; tail -3 $O/s4_tests_b.log
echo "== the oracle against itself at n = 1e8 (CPU)"
time python tools/oracle_sensitivity.py full_size 100000000 > $O/oracle_sensitivity_full_size.json 2> $O/oracle_sensitivity_full_size.err; echo "rc $?"; cat $O/oracle_sensitivity_full_size.json
