#!/bin/bash
# interleaved sweep of phase_ticks and keep_tail on the headline workload (run on the GPU box from the repository root; profiles/r04b_phase_sweep.log)
for rep in 1 2; do
for o in "phase_ticks=8000" "phase_ticks=4000" "phase_ticks=6000" "phase_ticks=12000" "phase_ticks=8000 --opt keep_tail=0" "phase_ticks=8000 --opt keep_tail=1" "phase_ticks=8000 --opt keep_tail=0.2"; do
python bench.py --steps 60 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --opt $o > gpurun_out/sweep_one.json 2>gpurun_out/sweep.err
python - "$o" <<PY
import json,sys
d=json.load(open("gpurun_out/sweep_one.json"))
k=d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n:round(k[n]["avg_ms"],3) for n in ("sdot","qdot","sadd","apply","fisher_y") if n in k}, flush=True)
PY
done
done
