#!/bin/bash
# How long does the headline workload stay in its stable regime?  The objective after K steps with ONE optimiser state, for a ladder
# of K (bench.py refuses to print a line when f_end >= f_start), then K = 1500 and 3000 in the three ways a step can be computed.
# Run on the GPU box from the repository root; profiles/r04b_long_run_ladder.log, r04b_long_run_forms.log.
one() {  # K option
	python bench.py --steps $1 --warmup 5 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --no-profile --opt $2 > gpurun_out/long_one.json 2> gpurun_out/long_one.err
	if [ $? -eq 0 ]; then python -c "
import json; d=json.load(open('gpurun_out/long_one.json')); c=d['config']; print($1, '$2', 'f_start %.17g f_end %.17g rejected steps %d pairs %d' % (c['f_start'], c['f_end'], c['rejected_steps'], c['rejected_pairs']), d['value'])"
	else echo "$1 $2 FAILED: $(tail -1 gpurun_out/long_one.err)"; fi
}
for K in 1500 1750 2000 2250 2500 2750 3000; do one $K phase_ticks=8000; done
for K in 1500 3000; do for o in phase_ticks=8000 phase_ticks=0 threepass=0; do one $K $o; done; done
