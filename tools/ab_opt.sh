#!/bin/bash
# interleaved A/B of one library option on the headline workload: tools/ab_opt.sh <name> <value A> <value B> [repetitions]
# (run on the GPU box from the repository root; one line per run)
N=${4:-4}
for rep in $(seq $N); do
for v in $2 $3; do
python bench.py --steps 100 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --opt $1=$v > gpurun_out/ab_one.json 2>gpurun_out/ab.err
python - "$1=$v" <<PY
import json,sys
d=json.load(open("gpurun_out/ab_one.json"))
k=d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n:round(k[n]["avg_ms"],3) for n in ("sdot","sdot2","qdot","sadd","apply") if n in k}, flush=True)
PY
done
done
