#!/bin/bash
# Round 5: C2 (oLBFGS n = 1e7, m = 10) from a C caller, the kernel-shape knobs that have not been tried at this size, a soak of the
# host path with the allocator provoked under the NEW pinning rule, then the whole suite without the mask.
set -o pipefail
O=gpurun_out/r05; mkdir -p $O
{ for mode in 0 2; do tools/latency olbfgs 1e7 10 300 0 1 $mode; done; tools/latency olbfgs 1e7 10 300 1 1 0; } > $O/c2_c_caller.log 2>&1; cat $O/c2_c_caller.log
: > $O/c2_ab.jsonl
for o in "" grid_cap=512 phase_ticks=2000 sdot2_per_cu=2 qdot_per_cu=2,sadd_per_cu=2 phase_ticks=0; do
	SQN_OPTS=$o python tools/bench_configs.py c2 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(json.dumps({'opts':'$o','steps_per_s':d['steps_per_s'],'ms_per_step':d['ms_per_step'],'kernel_ms':d['step']['kernel_ms'],'kernels':{k:v['avg_ms'] for k,v in d['kernels'].items()}}))" | tee -a $O/c2_ab.jsonl
done
python tools/host_fault_soak.py --tag new_rule_brk --cycles 120 --heap brk --max-seconds 330 > $O/soak_new_rule.txt 2>&1; tail -2 $O/soak_new_rule.txt | cut -c1-600
bash tools/suite_soak.sh default 300 tests
