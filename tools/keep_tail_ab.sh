#!/bin/bash
# interleaved A/B of keep_tail (the share of r0 / r stored with the default policy) with the clock-phased stores on; profiles/r04b_keep_tail_ab.log
for rep in 1 2 3 4; do
for o in "keep_tail=0.35" "keep_tail=0"; do
python bench.py --steps 100 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --opt $o > gpurun_out/sweep_one.json 2>gpurun_out/sweep.err
python - "$o" <<PY
import json,sys
d=json.load(open("gpurun_out/sweep_one.json"))
k=d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n:round(k[n]["avg_ms"],3) for n in ("sdot","qdot","sadd","apply") if n in k}, flush=True)
PY
done
done
