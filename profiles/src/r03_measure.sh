#!/bin/bash
# Round-3 measurement session on the GPU box (run from the repository root): profiles/README.md quotes these commands.
set -o pipefail
R=$PWD
O=$R/gpurun_out
mkdir -p $O/r03
python -m pytest tests/test_gpu_devices.py -q -k "host_path_of_the_multi_device_mode" > $O/r03/pytest_grouphost.log 2>&1; tail -3 $O/r03/pytest_grouphost.log
python -m pytest tests/test_gpu_parity.py -q -k "float_lockstep_parity_full_grids" > $O/r03/pytest_f32b.log 2>&1; tail -3 $O/r03/pytest_f32b.log
gcc -O2 -fopenmp -std=c99 -I include tools/c5_host_caller.c -L stochqn_amd/lib -lstochqn -lm -Wl,-rpath,$R/stochqn_amd/lib -o tools/c5_host_caller
./tools/c5_host_caller 200000000 20 24 > $O/r03/c5_host_one_device.log 2>&1; tail -4 $O/r03/c5_host_one_device.log
STOCHQN_HIP_DEVICES=2 STOCHQN_HIP_VIRTUAL_DEVICES=1 ./tools/c5_host_caller 200000000 20 24 > $O/r03/c5_host_two_virtual.log 2>&1; tail -4 $O/r03/c5_host_two_virtual.log
python tools/bench_configs.py c2 c4 c3f32 rccl1 > $O/r03/other_configs.jsonl 2> $O/r03/other_configs.err; cut -c1-220 $O/r03/other_configs.jsonl
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-caller --no-live-pmc > $O/prof_stats.json 2> $O/r03/prof_stats.err; tail -c 300 $O/prof_stats.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile > /dev/null 2> $O/r03/prof_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_write -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-caller --no-live-pmc --no-profile > /dev/null 2> $O/r03/prof_write.err
cd $R
python bench.py > $O/r03/bench_default_run.json 2> $O/r03/bench_default_run.err; python -c "
import json; d=json.load(open('$O/r03/bench_default_run.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'][:5], d['roofline']['frac'], d['two_loop']['ms'], d['host_caller']['strict_grad_0']['ms_per_step'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
ls $O/prof_stats/*/ $O/prof_fetch/*/ | head
