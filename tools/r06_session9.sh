#!/bin/bash
# Round 6: adjacent tiles in the sweep kernels (sweep_tile) and in the two-probe pass 1 (sdot2_tile) -- parity of the variants, then interleaved A/B
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
echo "== parity with sweep_tile = 2, sdot2_tile = 2 (lock-step, traces, two-loop in the sweeps form, host path)"
STOCHQN_TEST_OPTS=sweep_tile=2,sdot2_tile=2 python - <<'PY' > gpurun_out/r06/s9_parity.log 2>&1
import ctypes as C, os, sys, pytest
import stochqn_amd
lib = stochqn_amd.cdll(); lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
f32 = stochqn_amd.cdll(use_float=True); f32.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
for kv in os.environ["STOCHQN_TEST_OPTS"].split(","):
    k, v = kv.split("=")
    assert lib.stochqn_hip_set_option(k.encode(), float(v)) == 0 and f32.stochqn_hip_set_option(k.encode(), float(v)) == 0
sys.exit(pytest.main(["tests/test_gpu_parity.py", "tests/test_gpu_host_path.py", "-x", "-q", "-m", "gpu", "-k",
                      "two_loop or trace_parity or lockstep_parity or bit_for_bit or sliced or full_size or take_step or x_sent_ahead"]))
PY
echo "rc $?"; tail -3 $O/s9_parity.log
echo "== headline, interleaved: sweep_tile"
for rep in 1 2 3; do for v in 1 2; do
	python bench.py --steps 100 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --opt sweep_tile=$v > $O/s9_one.json 2> $O/s9.err || tail -3 $O/s9.err
	python - "sweep_tile=$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06/s9_one.json"))
k = d["kernels"]; rf = d["reference_form"]
print(sys.argv[1], d["value"], d["ms_per_step"], {n: round(k[n]["avg_ms"], 4) for n in ("apply", "pair_s", "pair_y_hv", "sdot", "qdot", "sadd") if n in k}, "reference form:", rf["two_loop_ms"], rf["two_loop_frac_of_8TBps"], rf["roofline"]["avg_launch_ms"], flush=True)
PY
done; done
echo "== C2, interleaved: sweep_tile x sdot2_tile"
for rep in 1 2; do for opts in sweep_tile=1,sdot2_tile=1 sweep_tile=2,sdot2_tile=1 sweep_tile=1,sdot2_tile=2 sweep_tile=2,sdot2_tile=2; do
	SQN_OPTS=$opts python tools/bench_configs.py c2 > $O/s9_c2_one.json 2> $O/s9_c2.err || tail -3 $O/s9_c2.err
	python - $opts <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r06/s9_c2_one.json").read().splitlines()[0])
k = d["kernels"]
print(sys.argv[1], d["steps_per_s"], d["step"]["frac_of_8TBps"], {n: k[n]["avg_ms"] for n in ("pair_y_diff", "apply", "sdot2", "qdot", "sadd") if n in k}, flush=True)
PY
	cat $O/s9_c2_one.json >> $O/s9_c2_tiles.jsonl
done; done
