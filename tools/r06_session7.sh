#!/bin/bash
# Round 6: Fisher pass 1 with two column tiles per trip (fisher_tile = 2) against one, interleaved; parity of the variant
set -o pipefail
O=gpurun_out/r06; mkdir -p $O
python - <<'PY'
import ctypes as C, numpy as np, torch, stochqn_amd, sys
sys.path.insert(0, "tests")
from oracle import oracle
lib = stochqn_amd.cdll()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_fisher_product.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
for n, fu in ((1, 3), (7, 5), (4097, 33), (70001, 128), (1000003, 130)):
    rng = np.random.default_rng(n)
    F = rng.random((fu, n)) - 0.5; s = rng.random(n) - 0.5
    tw, yw = oracle.fisher_product(F.reshape(-1), fu, s)
    for tile in (1, 2):
        assert lib.stochqn_hip_set_option(b"fisher_tile", float(tile)) == 0
        Fd, sd = torch.as_tensor(F.reshape(-1), device="cuda"), torch.as_tensor(s, device="cuda")
        t = np.zeros(fu); y = torch.zeros(n, dtype=torch.float64, device="cuda")
        assert lib.stochqn_hip_fisher_product(Fd.data_ptr(), fu, n, sd.data_ptr(), t.ctypes.data, y.data_ptr()) == 0
        et = np.linalg.norm(t - tw) / np.linalg.norm(tw); ey = np.linalg.norm(y.cpu().numpy() - yw) / np.linalg.norm(yw)
        print("n=%d fu=%d tile=%d: t %.2e y %.2e" % (n, fu, tile, et, ey)); assert et <= 1e-10 and ey <= 1e-10
lib.stochqn_hip_set_option(b"fisher_tile", 1.0)
PY
echo "== C4 (fu = 128), interleaved"
for rep in 1 2; do for opts in fisher_tile=1 fisher_tile=2 fisher_tile=2,fisher_split_per_cu=1; do
	SQN_OPTS=$opts C4_QUICK=1 python tools/bench_configs.py c4 > $O/s7_c4_one.json 2> $O/s7_c4.err || tail -3 $O/s7_c4.err
	python - $opts $O/s7_c4_one.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().splitlines()[0])
k = d["kernels"]
print(sys.argv[1], d["steps_per_s"], {n: (k[n]["avg_ms"], k[n].get("frac_of_8TBps")) for n in ("fisher_t", "fisher_y") if n in k}, flush=True)
PY
	cat $O/s7_c4_one.json >> $O/s7_c4_tile.jsonl
done; done
echo "== C3 Hv (fu = 32)"
for v in 1 2; do
	python bench.py --steps 40 --no-cpu-baseline --no-host-caller --no-live-pmc --value-runs 1 --sustain-seconds 0 --no-reference-form --opt fisher_tile=$v > $O/s7_c3_tile$v.json 2> $O/s7_c3.err || tail -3 $O/s7_c3.err
	python - $v $O/s7_c3_tile$v.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print("fisher_tile=%s" % sys.argv[1], d["value"], {n: d["kernels"][n] for n in ("fisher_t", "fisher_y")}, flush=True)
PY
done
