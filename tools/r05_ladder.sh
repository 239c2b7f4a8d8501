#!/bin/bash
# tools/r05_ladder.sh [n] -- the objective after K steps with ONE optimiser state on the DEVICE, K = 250 .. 3500, at an n the CPU
# oracle gets through as well (tools/oracle_long_run.py prints the same ladder): rounds 1 - 4's batch ("rank32": stalls) and
# round 5's ("averaged": falls geometrically).  profiles/r05_long_run_device_vs_oracle.log.  GPU box, repository root.
N=${1:-1000000}
one() {  # K batch
	SQN_BENCH_BATCH=$2 python bench.py --vars-per-gpu $N --steps $1 --warmup 0 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --no-profile > gpurun_out/ladder_one.json 2> gpurun_out/ladder_one.err
	if [ $? -eq 0 ]; then python -c "
import json; d=json.load(open('gpurun_out/ladder_one.json')); c=d['config']; print('device n=$N batch=$2 K=%5d  f_start %.17g  f_end %.17g  rejected steps %d pairs %d' % ($1, c['f_start'], c['f_end'], c['rejected_steps'], c['rejected_pairs']))"
	else echo "device n=$N batch=$2 K=$1 FAILED: $(tail -1 gpurun_out/ladder_one.err | cut -c1-200)"; fi
}
for K in 250 500 1000 1500 2000 2500 3000 3500; do one $K rank32; done
for K in 500 1000 2000 3000 4000; do one $K averaged; done
