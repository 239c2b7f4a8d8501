#!/bin/bash
# Round 5, last GPU call: the default bench run as the driver starts it, then the whole suite once more (no mask).
set -o pipefail
python bench.py > gpurun_out/r05_bench_default_run.json 2> gpurun_out/r05_bench_default_run.err; echo "bench rc $?"; cut -c1-700 gpurun_out/r05_bench_default_run.json
bash tools/suite_soak.sh default 300 tests
