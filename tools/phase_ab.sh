#!/bin/bash
# Interleaved A/B of the clock-phased stores on one box: the headline workload with phase_ticks = 8000 / 0 / 8000 / 0 (one bench line
# each, appended to $1), then config 4 (adaQN) the same way (appended to $2).  Run on the GPU box from the repository root.
OUT=${1:-gpurun_out/r04b_ab_phase_ticks.jsonl}
OUT4=${2:-gpurun_out/r04b_ab_c4.jsonl}
: > $OUT; : > $OUT4
for t in 8000 0 8000 0; do
	python bench.py --steps 60 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --opt phase_ticks=$t >> $OUT 2> gpurun_out/phase_ab.err
	tail -1 $OUT | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("c3 phase_ticks", d["config"].get("options"), d["value"], {n: round(k[n]["avg_ms"], 3) for n in ("sdot", "qdot", "sadd", "apply", "fisher_y") if n in k}, flush=True)'
done
for t in 8000 0 8000 0; do
	SQN_OPTS=phase_ticks=$t python tools/bench_configs.py c4 2>> gpurun_out/phase_ab.err >> $OUT4
done
python - $OUT4 <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(d["config"], d["workload"][-34:], d["steps_per_s"], {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items() if v["avg_ms"] > 0.1})
PY
