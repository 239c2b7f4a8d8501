#!/bin/bash
# Round 4, second session (clock-phased stores): the same measurement session as tools/r04_measure.sh, files named r04b_*.
# Run on the GPU box from the repository root; profiles/README.md quotes these commands.
set -o pipefail
R=$PWD
O=$R/gpurun_out
mkdir -p $O/r04b
python tools/bench_configs.py c2 c4 c3f32 rccl1 > $O/r04b/other_configs.jsonl 2> $O/r04b/other_configs.err; cut -c1-200 $O/r04b/other_configs.jsonl
python bench.py --config c5 --no-cpu-baseline --no-host-caller --no-live-pmc > $O/r04b/c5_shard_1gpu.json 2> $O/r04b/c5.err; tail -c 300 $O/r04b/c5_shard_1gpu.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 > $O/prof_stats.json 2> $O/r04b/prof_stats.err; tail -c 300 $O/prof_stats.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile --value-runs 1 > /dev/null 2> $O/r04b/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile --value-runs 1 > /dev/null 2> $O/r04b/prof_write.err
cd $R
ls $O/prof_stats/*/ $O/prof_fetch/*/ | head
