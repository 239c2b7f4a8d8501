#!/bin/bash
# Round-4 measurement session on the GPU box (run from the repository root): profiles/README.md quotes these commands.
# Part 1 ("profile"): rocprofv3 kernel stats + the two PMC passes of the default workload, the other configurations, the plain C
# config-5 caller.  Part 2 ("arena"): placement_arena_ab.sh, removed in round 5 with the arena code.
set -o pipefail
R=$PWD
O=$R/gpurun_out
mkdir -p $O/r04
if [ "$1" != "arena" ]; then
	./tools/c5_host_caller 200000000 20 64 2 > $O/r04/c5_host_one_device_full_ring.log 2>&1; tail -5 $O/r04/c5_host_one_device_full_ring.log
	python tools/bench_configs.py c2 c4 c3f32 rccl1 > $O/r04/other_configs.jsonl 2> $O/r04/other_configs.err; cut -c1-200 $O/r04/other_configs.jsonl
	cd /tmp && export TMPDIR=/tmp
	rm -rf $O/prof_stats $O/prof_fetch $O/prof_write
	rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 > $O/prof_stats.json 2> $O/r04/prof_stats.err; tail -c 300 $O/prof_stats.json
	rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile --value-runs 1 > /dev/null 2> $O/r04/prof_fetch.err
	rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile --value-runs 1 > /dev/null 2> $O/r04/prof_write.err
	cd $R
	ls $O/prof_stats/*/ $O/prof_fetch/*/ | head
else
	echo "the arena experiment is closed (round 5): the script and the code are gone"
fi
