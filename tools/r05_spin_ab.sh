#!/bin/bash
# tools/r05_spin_ab.sh -- does a call that polls its stream (option spin_wait_us) come back sooner than one that waits in
# hipStreamSynchronize?  (The option existed for this measurement only -- in wait_stream, for contexts without RCCL: poll hipStreamQuery
# for up to spin_wait_us, then hipStreamSynchronize -- and was removed again: no difference, profiles/r05_c2_ab_polling_wait.jsonl.)  C2 (oLBFGS n = 1e7: two synchronous calls per 0.72 ms step) interleaved, three times each; C3 once each.
O=gpurun_out/r05_spin_ab.jsonl; : > $O
for rep in 1 2 3; do
	for v in 0 2000; do
		SQN_OPTS=spin_wait_us=$v PROFILE_STEPS=0 python tools/bench_configs.py c2 2>/dev/null | tail -1 >> $O || exit 1
	done
done
for v in 0 20000; do SQN_OPTS=spin_wait_us=$v python bench.py --no-cpu-baseline --no-host-caller --no-live-pmc --opt spin_wait_us=$v 2>/dev/null | tail -1 >> $O || exit 1; done
python - <<'PY'
import json
for l in open('gpurun_out/r05_spin_ab.jsonl'):
    d=json.loads(l)
    print(d.get('config') if isinstance(d.get('config'),str) else 'C3-bench', d.get('workload',d.get('config'))if isinstance(d.get('config'),str) else '', d.get('steps_per_s', d.get('value')), d.get('ms_per_step'))
PY
