#!/bin/bash
# Round 6: pass 1 with two adjacent column tiles per iteration (sdot_tile = 2) -- parity, then interleaved A/B on the headline, C2, C4
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
echo "== parity (two-loop, traces, host path bit-identity, sharding, float)"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_path.py -x -q -m gpu -k "two_loop or trace_parity or lockstep_parity or bit_for_bit or sliced or full_size" > $O/s8_parity.log 2>&1; echo "rc $?"; tail -3 $O/s8_parity.log
echo "== headline, interleaved"
for rep in 1 2 3; do for v in 1 2; do
	python bench.py --steps 100 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --opt sdot_tile=$v > $O/s8_one.json 2> $O/s8.err || tail -3 $O/s8.err
	python - "sdot_tile=$v" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r06/s8_one.json"))
k = d["kernels"]
print(sys.argv[1], d["value"], d["ms_per_step"], d["two_loop"]["ms"], d["two_loop"]["frac_of_8TBps_on_bytes_moved"], {n: round(k[n]["avg_ms"], 4) for n in ("sdot", "sdot2", "qdot", "sadd", "apply") if n in k}, flush=True)
PY
done; done
echo "== C2 and C4"
for v in 1 2; do
	SQN_OPTS=sdot_tile=$v python tools/bench_configs.py c2 > $O/s8_c2_$v.json 2> $O/s8_c2.err || tail -3 $O/s8_c2.err
	SQN_OPTS=sdot_tile=$v C4_QUICK=1 python tools/bench_configs.py c4 > $O/s8_c4_$v.json 2> $O/s8_c4.err || tail -3 $O/s8_c4.err
	python - $v <<'PY'
import json, sys
for c in ("c2", "c4"):
    d = json.loads(open("gpurun_out/r06/s8_%s_%s.json" % (c, sys.argv[1])).read().splitlines()[0])
    k = d["kernels"]
    print(c, "sdot_tile=%s" % sys.argv[1], d["steps_per_s"], {n: (k[n]["avg_ms"], k[n].get("frac_of_8TBps")) for n in ("sdot", "sdot2", "qdot", "sadd", "fisher_t") if n in k}, flush=True)
PY
done
