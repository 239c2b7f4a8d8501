#!/bin/bash
# Round 6, fifth GPU call: the per-configuration measurement with the round's library (kernel stats, PMC traffic, lines), the other configurations
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
COMMIT=${COMMIT:-unknown} bash tools/r06_measure.sh c3 c2 c4 2>&1 | cut -c1-400
echo "== other configurations"
python tools/bench_configs.py c3f32 c3host rccl1 > $O/other_configs.jsonl 2> $O/other_configs.err; echo "rc $?"; cut -c1-300 $O/other_configs.jsonl; tail -2 $O/other_configs.err
echo "== C5 shard on one GPU"
python bench.py --config c5 --no-cpu-baseline --no-host-caller --no-live-pmc > $O/c5_shard_1gpu.json 2> $O/c5_shard_1gpu.err; echo "rc $?"; cut -c1-300 $O/c5_shard_1gpu.json
ls -la $O | head -60
du -sh $O
