#!/usr/bin/env python3
"""bench.py -- optimiser steps/s of the stochastic quasi-Newton step path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): SQN with
Hessian-vector correction pairs, n = 1e8 variables per GPU, m = 20 stored pairs, L = 10,
"bsize" = 32 (the Hessian-vector product is A'(A v)/32 over a synthetic dense 32 x n mini-batch), fp64,
check_nan = 1, synthetic noisy-quadratic gradients.  One *step* = everything the library does
from one `niter` to the next: one `run_SQN` call with the two-loop recursion, the guard and the
position update, plus -- every L-th step -- the calls that build the new correction pair.
`--config c5` is BASELINE.json configs[4]'s per-GPU shard (n = 1.25e8 per GPU: n_total = 1e9 on 8 GPUs).

All inputs live in HBM before the timed region starts (torch tensors handed to the C ABI as
device pointers); the library is driven through run_SQN exactly like the reference's callers do.
Inputs come from a counter-based generator (stochqn_hip_synth_*: element i depends only on i, the
seed and a stream id), so rank p of P holds exactly its slice of the one-rank problem.

N > 1: one process per GPU, the n dimension is sharded, every dot product inside the library is a
local partial + one RCCL all-reduce (stochqn_hip_comm_init).  `python bench.py --gpus N` WITHOUT a
launcher starts the N ranks itself (a fresh `python -m torch.distributed.run` child, decided before
this process touches torch or HIP) and relays rank 0's JSON line; under a launcher (WORLD_SIZE set)
it is one of the ranks.  It never falls through to a 1-GPU run labelled `--gpus N`.
`--in-process`: the same N shards driven by ONE host process through the library's single-process
multi-device mode (option "devices"; what a C / R caller of the reference ABI gets).
Weak scaling: n per GPU is fixed, so `value` is normalised to the n = 1e8 problem
(value = steps/s * n_total / 1e8) to stay an aggregate that grows with N.

`--gpus N` with N > 1 and no further flags measures, after the primary leg (config 3, n = 1e8 per GPU: `value`), in the
SAME JSON line under "legs": `c5` -- BASELINE config 5's weak-scaling point, n = 1.25e8 per GPU (n_total = 1e9 on 8
GPUs), next to the committed 1-GPU yardstick and its 85 % line; `strong` -- n = 1e8 split over the GPUs; `allreduce_us`
-- latency of one reduction on the library's own communicator, and how many a step issues; `in_process` -- the same
per-GPU problem driven by ONE host process (a fresh child of rank 0, started when the ranks have finished).  A leg that
fails is reported as {"error": ...} and named in "legs_failed"; the primary result is printed regardless.
N = 1 adds `host_caller` (every array in pageable host memory, as R and numpy callers have them: PCIe-inclusive) and
`cpu_baseline` (the oracle on the host cores).

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20240611
PEAK = 8000.0  # GB/s, MI355X HBM3E (MI355X_MICROARCH.md)

# ------------------------------------------------------------------------------------------------
# ONE wall-clock budget for the whole command: the clock starts when this process is entered, every
# auxiliary leg is admitted only if its measured cost still fits, every child process gets no more than what is left, and
# the one JSON line is printed no later than the budget -- with what there is, and `legs_skipped` naming the rest.
# ------------------------------------------------------------------------------------------------
ENTRY_EPOCH = time.time()
DEFAULT_BUDGET_S = 420.0
RESERVE_S = 6.0                # printing the line, tearing the communicators down, leaving
# Seconds each auxiliary leg took on one MI355X box (profiles/r06_bench_leg_seconds.json: the `budget.leg_seconds` of
# default runs at N = 1 and of the 3-rank rehearsal), rounded up by a third.  A leg is admitted when this much is left.
LEG_COST_S = {"profile": 2.0, "value_runs": 6.0, "sustained": 10.0, "reference_form": 2.0, "two_loop_micro": 2.0,
              "host_copies": 6.0, "host_caller": 45.0, "cpu_baseline": 60.0, "live_pmc": 10.0,
              "c5": 6.0, "strong": 4.0, "allreduce_us": 2.0, "in_process": 40.0, "c5_yardstick": 15.0}
# (measured, round 6, gpurun_out/r06/s2_*: N = 1 default run: profile 0.4, value_runs 4.3, sustained 6.0, reference_form 0.3,
#  two_loop_micro 0.7, host_copies 4.0, host_caller 32.5, live_pmc 5.2 s; cpu_baseline 92 s with the thread team packed on two
#  core complexes -- it now probes the placement and sizes itself to the budget; 3-rank rehearsal at n / 50: c5 0.4, strong 0.3,
#  all-reduce 0.1, in_process 5.6, yardstick 1.5 s, which scale to what is written above at full size.)


class Budget:
    """What is left of the command's wall-clock budget.  The deadline is an epoch time that travels to every child process
    through BENCH_DEADLINE_EPOCH, so that the ranks a launcher-less `--gpus N` starts, the in-process child, the yardstick
    child and the PMC children all count down to the same moment.  `clock` is injectable (tests)."""

    def __init__(self, total_s=None, entry=None, clock=time.time, costs=None, env=os.environ):
        self.clock = clock
        entry = ENTRY_EPOCH if entry is None else entry
        if total_s is None:
            total_s = float(env.get("BENCH_BUDGET_S", DEFAULT_BUDGET_S))
        self.total_s = total_s
        inherited = env.get("BENCH_DEADLINE_EPOCH")
        self.deadline = float(inherited) if inherited else entry + total_s
        self.inherited = bool(inherited)
        self.costs = dict(LEG_COST_S if costs is None else costs)
        self.skipped, self.spent = [], {}

    def export(self, env):
        env["BENCH_DEADLINE_EPOCH"] = repr(self.deadline)
        return env

    def left(self):
        """Seconds that may still be spent on measuring (the reserve for printing and leaving taken off)."""
        return self.deadline - self.clock() - RESERVE_S

    def fits(self, leg, cost=None):
        return self.left() >= (self.costs.get(leg, 0.0) if cost is None else cost)

    def skip(self, leg, why=None):
        self.skipped.append({"leg": leg, "needs_s": self.costs.get(leg), "left_s": round(self.left(), 1),
                             "why": why or "does not fit into what is left of the budget"})

    def admit(self, leg, cost=None):
        """One rank's decision (run() makes it collective): True, or False with the leg named in `skipped`."""
        if self.fits(leg, cost):
            return True
        self.skip(leg)
        return False

    def child_timeout(self, cap):
        """A child process gets `cap` seconds or what is left, whichever is less (never less than a second: it then fails at once)."""
        return max(1.0, min(float(cap), self.left()))

    def note(self, leg, seconds):
        self.spent[leg] = round(self.spent.get(leg, 0.0) + seconds, 2)

    def report(self):
        return {"budget_s": round(self.deadline - ENTRY_EPOCH, 1) if self.inherited else self.total_s,
                "deadline_inherited": self.inherited,
                "used_s": round(self.clock() - ENTRY_EPOCH, 1), "left_s": round(self.left() + RESERVE_S, 1),
                "leg_seconds": dict(self.spent), "leg_cost_s": self.costs}


LIVE_CHILDREN = []             # Popen objects of child processes in flight (the watchdog takes them along when it leaves)


def run_child(cmd, env, timeout, cwd=None, quiet=False):
    """subprocess.run with a process group of its own: at the timeout (or when the watchdog leaves) the whole group goes.
    Returns (returncode, stdout, stderr); returncode None = timed out."""
    import signal
    out = subprocess.DEVNULL if quiet else subprocess.PIPE
    p = subprocess.Popen(cmd, stdout=out, stderr=out, text=True, env=env, cwd=cwd, start_new_session=True)
    LIVE_CHILDREN.append(p)
    try:
        so, se = p.communicate(timeout=timeout)
        return p.returncode, so or "", se or ""
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        so, se = p.communicate()
        return None, so or "", se or ""
    finally:
        LIVE_CHILDREN.remove(p)


def kill_children():
    import signal
    for p in list(LIVE_CHILDREN):
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass


RANK_ENV = ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE", "MASTER_ADDR",
            "MASTER_PORT", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS")


def child_env(budget):
    """Environment of a 1-process child of rank 0: nothing of the launcher's, the same deadline."""
    env = {k: v for k, v in os.environ.items() if k not in RANK_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return budget.export(env)
CONFIGS = {"c3": 100_000_000, "c5": 125_000_000}
# counter-based generator streams (stochqn_hip.h): which vector a draw belongs to
ST_D, ST_S, ST_X0, ST_NOISE = 0, 1, 3, 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="timed steps (default: about 2.2 s of device time)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3",
                    help="c3: n = 1e8 per GPU (the headline configuration); c5: n = 1.25e8 per GPU (n_total = 1e9 on 8 GPUs)")
    ap.add_argument("--n", "--vars-per-gpu", dest="n", type=int, default=0,
                    help="variables per GPU, overrides --config (--vars-per-gpu under torch.distributed.run, whose parser takes --n for itself)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: the configuration's n is the TOTAL problem, split evenly over the GPUs (default: weak, n per GPU)")
    ap.add_argument("--mem", type=int, default=20)
    ap.add_argument("--upd-freq", type=int, default=10)
    ap.add_argument("--bsize", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-caller", action="store_true",
                    help="skip the host-caller leg (N = 1): the same step with every array in pageable host memory, PCIe included")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="N = 1: take roofline.traffic from the committed profiles/ instead of counting it now (two rocprofv3 --pmc "
                         "passes over a 5-step child run of this workload, about a minute)")
    ap.add_argument("--strict-legs", action="store_true",
                    help="N > 1: a failing secondary leg makes the whole run fail (non-zero exit, no JSON line) instead of being "
                         "reported in `legs_failed` next to the primary result")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="N > 1: only the primary weak-scaling leg (default: also config 5's n = 1.25e8 per GPU, the strong-scaling "
                         "split of n = 1e8, the all-reduce latency and the one-process / N-devices mode, all in the same JSON line)")
    ap.add_argument("--host-n", type=int, default=0, help="--in-process: total problem size of the host-caller leg (0 = 1e8; 4e6 per shard with --virtual-devices)")
    ap.add_argument("--cpu-n", type=int, default=0, help="CPU baseline: problem size (0 = n when host memory allows, else the largest that fits)")
    ap.add_argument("--no-profile", action="store_true", help="skip the second, HIP-event-profiled pass (no roofline object)")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="after the K timed steps: the same workload for about this long (whole steps, profiler off), reported as "
                         "`sustained` -- the rate over seconds instead of a fraction of one; 0 = skip")
    ap.add_argument("--value-runs", type=int, default=3,
                    help="the K-step region this many times in all (the first one is `value`; before each further one the gradient array is "
                         "freed and allocated again): `value_runs` = every value, minimum / median / maximum -- the run-to-run spread of "
                         "5-8 %% that comes with where the arrays land (DESIGN.md 3.3), next to the one draw")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (process group, RCCL communicator) even with one rank")
    ap.add_argument("--rehearse", action="store_true",
                    help="rehearsal of the N > 1 control flow on a box with ONE GPU: every rank uses cuda:0, "
                         "torch.distributed runs over gloo and the library's all-reduce goes through a gloo callback "
                         "(stochqn_hip_comm_init_custom).  Exercises exactly the code the N-GPU run takes, minus RCCL; "
                         "the number it prints is not a measurement.")
    ap.add_argument("--in-process", action="store_true",
                    help="N shards inside ONE process: the library's single-process multi-device mode (option 'devices')")
    ap.add_argument("--virtual-devices", action="store_true",
                    help="with --in-process: allow more shards than GPUs (all on the visible devices, host-side reducer); rehearsal only")
    ap.add_argument("--no-reference-form", action="store_true",
                    help="skip the extra untimed steps in the reference's sweep form and the two-loop micro-benchmark")
    ap.add_argument("--dump-x", default="", help="write the final x of this rank to <path>.<rank>.npy (sharded-vs-unsharded checks)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="stochqn_hip_set_option before the run (grid_cap, reverse, nontemporal, threepass ...)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------
# N > 1 without a launcher: start the ranks ourselves, from a process that has not touched the GPU
# ------------------------------------------------------------------------------------------------
def self_launch(args):
    """Start `--gpus N` ranks as a fresh torch.distributed.run child, relay rank 0's JSON line.
    Returns the exit code: non-zero if any rank failed, fewer than N devices are visible, or no
    result line came back."""
    if not args.rehearse:
        import torch                                   # device_count() does not initialise HIP
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d asked for but only %d device(s) are visible; refusing to run a "
                             "smaller job under that label\n" % (args.gpus, have))
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    cmd += [a if a != "--n" else "--vars-per-gpu" for a in sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    Budget().export(env)                               # the ranks count down to THIS process's deadline
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in child.stdout:
        if out.startswith('{"metric"'):
            line = out
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank job failed (exit code %d)\n" % (args.gpus, rc))
        return rc
    if line is None:
        sys.stderr.write("bench.py: the %d-rank job printed no result line\n" % args.gpus)
        return 1
    got = json.loads(line)
    if got.get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: asked for %d GPUs, the job reports %r\n" % (args.gpus, got.get("n_gpus")))
        return 1
    sys.stdout.write(line)
    sys.stdout.flush()
    return 0


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.in_process:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("--in-process is a single-process mode: do not start it under a launcher")
        return run_in_process(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    run(args)


def prototypes(lib):
    """ctypes prototypes of the measurement helpers of include/stochqn_hip.h that Workload calls."""
    u64 = C.c_ulonglong
    lib.stochqn_hip_synth_uniform.argtypes = [C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double, C.c_double]
    lib.stochqn_hip_synth_noisy_grad.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, u64, u64, u64, u64, C.c_double]
    lib.stochqn_hip_synth_batch_row.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, u64, C.c_uint, C.c_uint]
    lib.stochqn_hip_fisher_product.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]


REHEARSAL_SCALE = 50           # --rehearse without --vars-per-gpu: every configuration's n divided by this (ranks share one GPU)


class Workload:
    """One instance of the benchmark problem on this rank: f(x) = 1/2 sum d_i x_i^2 with noisy gradients, a dense
    `bs` x n Hessian mini-batch, caller-owned optimiser state with the ring already full, driven through
    run_SQN exactly as the reference's callers do.  Everything lives in HBM (torch tensors handed over as device pointers)."""

    def __init__(self, ctx, n, first, m, L, bs):
        import numpy as np
        import torch
        from stochqn_amd import _abi
        self.ctx, self.n, self.first, self.m, self.L, self.bs = ctx, n, first, m, L, bs
        lib, dev = ctx["lib"], ctx["dev"]
        f64 = torch.float64
        self.d = self.uniform(torch.empty(n, dtype=f64, device=dev), ST_D, 0, 0.5, 1.0)          # d_i = 0.5 + u(i,0,0)
        self.x = self.uniform(torch.empty(n, dtype=f64, device=dev), ST_X0, 0, 1.0, 1.0)         # x0_i = 1 + u(i,3,0)
        # Hessian mini-batch of `bs` sample vectors, stored dense [bs][n].  The samples have disjoint supports -- sample k lives
        # on the variables i = k mod bs -- and are scaled by the size of their class: a_k,i = bs sqrt(d_i / n_total).  Then
        # A'A/bs = D^(1/2) P D^(1/2), P the projector onto the bs class indicators: a rank-bs matrix with eigenvalues ~ mean(d)
        # that lies BELOW the objective's Hessian D = diag(d) in the Loewner order -- what bs aggregated probes see of D.  (Rounds
        # 1 - 4 left the 1/n_total out: bs eigenvalues of ~ n/bs, gamma = s'y/y'y ~ bs/n, a step that collapses -- the objective
        # stalled at 0.37 f0 and, at n = 1e8, blew up near step 2800; the CPU oracle stalls the same way,
        # profiles/r05_oracle_long_run.log.)  The product streams the full dense batch (2*bs*n words, like any real mini-batch).
        self.A = torch.empty(bs * n, dtype=f64, device=dev)
        d_batch = self.d * (float(bs) / float(n * ctx["world"]))
        if os.environ.get("SQN_BENCH_BATCH") == "rank32":          # rounds 1 - 4's scaling, for tools/r05_ladder.sh only (device against oracle in the stall)
            d_batch = self.d.clone()
        for k in range(bs):
            assert lib.stochqn_hip_synth_batch_row(self.A.data_ptr() + 8 * k * n, d_batch.data_ptr(), n, first, k, bs) == 0
        del d_batch
        # optimiser state, owned by the caller (profile B), ring already full
        self.S = torch.empty(m * n, dtype=f64, device=dev)
        self.Y = torch.empty(m * n, dtype=f64, device=dev)
        self.fill_ring()
        self.grad = torch.empty(n, dtype=f64, device=dev)
        self.hv = torch.empty(n, dtype=f64, device=dev)
        self.x_sum = torch.zeros(n, dtype=f64, device=dev)
        self.x_avg_prev = self.x.clone()
        self.t_buf = torch.zeros(bs, dtype=f64, device=dev)
        self.rho_h, self.alpha_h = np.zeros(m), np.zeros(m)
        self.dummy = torch.zeros(1, dtype=f64, device=dev)
        self.b = _abi.bfgs_mem(self.S.data_ptr(), self.Y.data_ptr(), self.rho_h.ctypes.data, self.alpha_h.ctypes.data,
                               self.dummy.data_ptr(), self.dummy.data_ptr(), m, m, 3 % m, L, 0.0, 0.0)
        self.w = _abi.workspace_SQN(C.pointer(self.b), self.dummy.data_ptr(), self.x_sum.data_ptr(), self.x_avg_prev.data_ptr(), 0,
                                    L, 1, 1, 1, n)    # niter = L: the "first average" special case is behind us
        self.req, self.req_vec = C.c_void_p(self.x.data_ptr()), C.c_void_p()
        self.task, self.info = C.c_int(101), C.c_int(200)
        self.step_size = 0.05
        self.ptr2t = {self.x.data_ptr(): self.x, self.x_sum.data_ptr(): self.x_sum, self.x_avg_prev.data_ptr(): self.x_avg_prev}
        self.counters = {"calls": 0, "hv": 0, "bad": 0, "rejected": 0}
        self.t_idx = 0
        self.steps_done, self.resets, self.warm = 0, 0, 5
        # the caller's Hessian-vector routine A'(A v)/bs, checked once (on v = x0) against plain torch products over the same batch.
        # This is also where its one-time set-up happens (the product's scratch and stream: 5.8 ms on the first call, which would
        # otherwise land in whichever timed step builds the first pair -- 2.7 % of a K = 20 window)
        assert lib.stochqn_hip_fisher_product(self.A.data_ptr(), bs, n, self.x.data_ptr(), self.t_buf.data_ptr(), self.hv.data_ptr()) == 0
        rows = [self.A[k * n:(k + 1) * n] for k in range(bs)]       # row by row: one [32 x 1e8] gemv is beyond rocBLAS's 32-bit indexing
        t = torch.stack([torch.dot(r, self.x) for r in rows])
        if ctx["dist"] is not None:
            th = t.to(ctx["cpu_or_dev"])
            ctx["dist"].all_reduce(th)
            t = th.to(dev)
        want = torch.zeros(n, dtype=f64, device=dev)
        for k in range(bs):
            want.add_(rows[k], alpha=float(t[k]) / bs)
        err = float(torch.linalg.vector_norm(self.hv - want) / torch.linalg.vector_norm(want))
        assert err <= 1e-12, "Hessian-vector product A'(Av)/bs: %r from torch's" % err
        del rows, t, want

    def uniform(self, out, stream, t, a, b):
        assert self.ctx["lib"].stochqn_hip_synth_uniform(out.data_ptr(), out.numel(), self.first, SEED, stream, t, a, b) == 0
        return out

    def fill_ring(self):
        import torch
        n = self.n
        for k in range(self.m):                                   # s_k,i = 1e-3 (u(i,1,k) - 0.5), y_k = d .* s_k
            sk = self.uniform(self.S[k * n:(k + 1) * n], ST_S, k, -0.5e-3, 1e-3)
            torch.mul(self.d, sk, out=self.Y[k * n:(k + 1) * n])

    def one_step(self):
        """Advance the optimiser by exactly one iteration (niter + 1)."""
        lib, be, n, t = self.ctx["lib"], self.ctx["be"], self.n, self.t_idx
        w, task, req, req_vec, info, counters = self.w, self.task, self.req, self.req_vec, self.info, self.counters
        target = w.niter + 1
        while w.niter < target:
            if task.value == 101:                                  # calc_grad at *req: g = d x (1 + 0.01 (2 u(i,4,t) - 1))
                at = self.ptr2t[req.value]
                assert lib.stochqn_hip_synth_noisy_grad(self.grad.data_ptr(), self.d.data_ptr(), at.data_ptr(), n, self.first, SEED, ST_NOISE, t, 0.01) == 0
            elif task.value == 104:                                # calc_hess_vec: A'(A v)/bs at x_avg
                counters["hv"] += 1
                rc = lib.stochqn_hip_fisher_product(self.A.data_ptr(), self.bs, n, req_vec.value, self.t_buf.data_ptr(), self.hv.data_ptr())
                assert rc == 0
            rc = be.run_SQN(self.step_size, self.x.data_ptr(), self.grad.data_ptr(), self.hv.data_ptr(), C.byref(req), C.byref(req_vec),
                            C.byref(task), C.byref(w), C.byref(info))
            assert rc in (0, 1), rc
            counters["calls"] += 1
            counters["bad"] += info.value == 203
            counters["rejected"] += info.value == 202
        self.t_idx += 1

    def steps(self, k):
        if os.environ.get("BENCH_STEP_TIMES"):                     # diagnostic: wall time of every step (each run_SQN is synchronous), on stderr
            ts = []
            for _ in range(k):
                t0 = time.perf_counter()
                self.one_step()
                ts.append(round(1e3 * (time.perf_counter() - t0), 3))
            sys.stderr.write("bench.py: %d steps, ms each: %s\n" % (k, ts))
            self.steps_done += k
            return
        for _ in range(k):
            self.one_step()
        self.steps_done += k

    # The gradient noise of this workload is multiplicative (g = d x (1 + 0.01 (2u - 1))), so there is no noise floor: the
    # objective falls geometrically (1e-17 per 500 steps, tools/oracle_long_run.py, profiles/r05_oracle_long_run.log) and leaves
    # the range of a double after ~9000 steps with one optimiser state -- s = 0, rho = 1/0, the guard rejects the steps and a
    # rejected step skips the update (less work).  No leg may walk into that.
    STABLE_STEPS = 5000

    def keep_stable(self, planned):
        """Before a leg of `planned` steps: back to the initial state if this state would leave the stable regime during it."""
        if self.steps_done + planned > self.STABLE_STEPS:
            self.reset()

    def reset(self):
        """x0, the synthetic ring and empty averages again -- what the K-step region started from (then `warm` warm-up steps)."""
        self.uniform(self.x, ST_X0, 0, 1.0, 1.0)
        self.fill_ring()
        self.x_sum.zero_()
        self.x_avg_prev.copy_(self.x)
        self.b.mem_used, self.b.mem_st_ix = self.m, 3 % self.m
        self.w.niter, self.w.section = self.L, 1
        self.req.value, self.req_vec.value = self.x.data_ptr(), None
        self.task.value, self.info.value = 101, 200
        self.ctx["lib"].stochqn_hip_invalidate(C.c_void_p(self.S.data_ptr()))      # the ring was rewritten from outside
        self.steps_done = 0
        self.resets += 1
        self.steps(self.warm)

    def objective(self):
        import torch
        v = float(0.5 * torch.sum(self.d * self.x * self.x))
        dist = self.ctx["dist"]
        if dist is not None:
            tv = torch.tensor([v], dtype=torch.float64, device=self.ctx["cpu_or_dev"])
            dist.all_reduce(tv)
            v = float(tv.item())
        return v

    def free(self):
        """Give the device memory back before the next leg allocates its own (C5: 40 GB of S and Y + 32 GB of batch per GPU)."""
        import torch
        self.ctx["lib"].stochqn_hip_release_all()
        for name in ("d", "x", "A", "S", "Y", "grad", "hv", "x_sum", "x_avg_prev", "t_buf", "dummy"):
            setattr(self, name, None)
        self.ptr2t = {}
        torch.cuda.empty_cache()


def barrier(ctx):
    import torch
    torch.cuda.synchronize()
    if ctx["dist"] is not None:
        ctx["dist"].barrier()
    torch.cuda.synchronize()


def max_over_ranks(ctx, v):
    import torch
    if ctx["dist"] is None:
        return v
    t = torch.tensor([v], dtype=torch.float64, device=ctx["cpu_or_dev"])
    ctx["dist"].all_reduce(t, op=ctx["dist"].ReduceOp.MAX)
    return float(t.item())


def all_ok(ctx, ok):
    """Every rank learns whether the leg worked on ALL ranks (a leg is only reported when it did)."""
    import torch
    if ctx["dist"] is None:
        return ok
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=ctx["cpu_or_dev"])
    ctx["dist"].all_reduce(t, op=ctx["dist"].ReduceOp.MIN)
    return float(t.item()) > 0.5


STAT_NAMES = ("steps_three_pass", "steps_sweeps", "steps_plain", "steps_kappa_fallback",
              "allreduces", "allreduce_doubles")


def stats(lib):
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    return {k: int(lib.stochqn_hip_stat(k.encode())) for k in STAT_NAMES}


def stat_diff(a, b):
    return {k: b[k] - a[k] for k in a}


def timed_leg(ctx, n_gpu, m, L, bs, steps, warmup, what):
    """A secondary leg of a multi-GPU run: its own problem at `n_gpu` variables per GPU, W warm-up + K timed steps with the
    library's event profiler off (barrier + synchronize on both sides, max over ranks), freed afterwards."""
    world, rank = ctx["world"], ctx["rank"]
    wl = Workload(ctx, n_gpu, rank * n_gpu, m, L, bs)
    try:
        f0 = wl.objective()
        wl.steps(warmup)
        ctx["lib"].stochqn_hip_profile_enable(0)
        barrier(ctx)
        s0 = stats(ctx["lib"])
        t0 = time.perf_counter()
        wl.steps(steps)
        barrier(ctx)
        el = max_over_ranks(ctx, time.perf_counter() - t0)
        s1 = stat_diff(s0, stats(ctx["lib"]))
        f1 = wl.objective()
        n_total = n_gpu * world
        sps = steps / el
        return {"what": what, "n_per_gpu": n_gpu, "n_total": n_total, "steps": steps, "warmup": warmup,
                "ms_per_step": round(1e3 * el / steps, 3), "steps_per_s": round(sps, 3),
                "value_normalised_to_1e8": round(sps * n_total / 1e8, 3),
                "hess_vec_requests": wl.counters["hv"], "rejected_steps": wl.counters["bad"], "rejected_pairs": wl.counters["rejected"],
                "allreduces_per_step": round(s1["allreduces"] / steps, 2),
                "objective_fell": bool(f1 < f0)}
    finally:
        wl.free()


def allreduce_latency(ctx):
    """Median of 200 all-reduces of 20 doubles on the library's own communicator, each followed by a stream synchronisation:
    what one reduction of the kernel chain costs on this fabric (three of them per step in the three-pass form)."""
    lib = ctx["lib"]
    med, mn = C.c_double(), C.c_double()
    lib.stochqn_hip_comm_allreduce_probe.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    rc = lib.stochqn_hip_comm_allreduce_probe(20, 200, C.byref(med), C.byref(mn))
    if rc != 0:
        raise RuntimeError("stochqn_hip_comm_allreduce_probe returned %d" % rc)
    return {"median_us": round(max_over_ranks(ctx, med.value), 2), "min_us": round(max_over_ranks(ctx, mn.value), 2),
            "doubles": 20, "reps": 200, "measured": "ncclAllReduce (or the rehearsal's reducer) + hipStreamSynchronize, wall clock, max over ranks"}


def run(args):
    # Everything that libraries print on fd 1 (RCCL's version banner, for one) goes to stderr; the one
    # JSON line is written to the real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import stochqn_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus (%d) must equal WORLD_SIZE (%d)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the library has no CPU path)")
    if args.rehearse:
        local_rank = 0                                   # all ranks share the one GPU of the box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    lib = stochqn_amd.cdll()
    be = stochqn_amd.lib()
    assert lib.stochqn_hip_available() == 1
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    for kv in args.opt:
        name, val = kv.split("=")
        assert lib.stochqn_hip_set_option(name.encode(), float(val)) == 0, kv
    dist, reducer = None, None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # The harness's own collectives (barriers, max over ranks, the communicator id) travel over gloo on the CPU: a few bytes
        # per call, and one RCCL communicator fewer in the process -- the library's.  Its reductions go over RCCL; in a
        # rehearsal (ranks sharing one GPU, which RCCL refuses) or when RCCL cannot be brought up they go over gloo too.
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        reducer = "gloo (rehearsal: the ranks share one GPU)" if args.rehearse else "rccl"
        if not args.rehearse:
            uid = torch.zeros(129, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_ubyte * 128)()
                if lib.stochqn_hip_comm_unique_id(buf) == 0:
                    uid = torch.tensor(list(buf) + [1], dtype=torch.uint8)
            dist.broadcast(uid, 0)
            # First contact with RCCL on N > 1 ranks happens on the driver's node.  ncclCommInitRank is a rendezvous: if a peer
            # or the fabric never answers it does not return -- and there is no result yet that a watchdog could print.  So it
            # runs on a thread of its own and gets BENCH_RCCL_INIT_S seconds (90); a rank that has not come back by then says so,
            # the ranks agree (MIN below) and the whole run goes over the gloo reducer instead: degraded, but a measurement.
            box = {"rc": None}

            def comm_init():
                if os.environ.get("BENCH_TEST_RCCL_INIT_HANGS"):     # tests: an init that never returns
                    time.sleep(1e6)
                box["rc"] = lib.stochqn_hip_comm_init(rank, world, bytes(uid[:128].tolist())) if int(uid[128]) == 1 else -1
            th = threading.Thread(target=comm_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("BENCH_RCCL_INIT_S", "90")))
            rc = box["rc"] if box["rc"] is not None else -2
            if rc == -2:
                sys.stderr.write("bench.py: rank %d: stochqn_hip_comm_init (ncclCommInitRank) had not returned after %s s\n"
                                 % (rank, os.environ.get("BENCH_RCCL_INIT_S", "90")))
            if os.environ.get("BENCH_TEST_RCCL_FAILS"):              # tests: the fall-back below
                rc = -1
            up = torch.tensor([1.0 if rc == 0 else 0.0], dtype=torch.float64)
            dist.all_reduce(up, op=dist.ReduceOp.MIN)
            if float(up.item()) < 0.5:
                sys.stderr.write("bench.py: rank %d: the library's RCCL communicator could not be set up (rc %d here): the reductions of this "
                                 "run go over gloo on the host instead -- three sums of a few dozen doubles per step\n" % (rank, rc))
                if rc != -2:                                         # (a rank whose init never returned has nothing to finalise -- and must not wait for it)
                    fin = threading.Thread(target=lib.stochqn_hip_comm_finalize, daemon=True)
                    fin.start()
                    fin.join(20.0)
                reducer = "gloo (the library's RCCL communicator could not be set up)"
        if reducer != "rccl":
            hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))   # the runtime already loaded
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            REDUCER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

            def gloo_allreduce(user, buf, count, stream):
                # hipMemcpy on the null stream orders itself after the library's (blocking) stream and
                # before whatever the library enqueues next; the sum itself travels over gloo on the CPU
                host = torch.empty(count, dtype=torch.float64)
                if hip.hipMemcpy(host.data_ptr(), buf, 8 * count, 2) != 0:
                    return 1
                dist.all_reduce(host)
                return 0 if hip.hipMemcpy(buf, host.data_ptr(), 8 * count, 1) == 0 else 1

            run.keep_alive = REDUCER(gloo_allreduce)
            lib.stochqn_hip_comm_init_custom.argtypes = [C.c_int, C.c_int, REDUCER, C.c_void_p]
            assert lib.stochqn_hip_comm_init_custom(rank, world, run.keep_alive, None) == 0
    cpu_or_dev = "cpu"
    ctx = {"lib": lib, "be": be, "dev": dev, "dist": dist, "cpu_or_dev": cpu_or_dev, "rank": rank, "world": world}

    def config_n(name):
        n = CONFIGS[name]
        return n // REHEARSAL_SCALE if args.rehearse else n       # a rehearsal shares one GPU between the ranks

    n_gpu = args.n if args.n > 0 else config_n(args.config)    # variables per GPU
    if args.strong:
        n_gpu = n_gpu // world                                 # the same total problem over 1 / 2 / 4 / 8 GPUs
    n = n_gpu                                                  # variables this process holds
    n_total = n * world
    m, L, bs = args.mem, args.upd_freq, args.bsize
    f64 = torch.float64
    prototypes(lib)

    wl = Workload(ctx, n, rank * n, m, L, bs)
    x, S, Y, A, d = wl.x, wl.S, wl.Y, wl.A, wl.d

    # ---- the measurement: W warm-up steps, then EXACTLY K steps, profiler off -----------------------
    f0 = wl.objective()
    wl.warm = args.warmup
    wl.steps(args.warmup)
    lib.stochqn_hip_profile_enable(0)
    barrier(ctx)
    st0 = stats(lib)
    t0 = time.perf_counter()
    wl.steps(args.steps)
    barrier(ctx)
    elapsed_local = time.perf_counter() - t0
    forms = stat_diff(st0, stats(lib))
    elapsed = elapsed_local
    per_rank_ms = [round(1e3 * elapsed_local / args.steps, 3)]
    rccl_nranks = lib.stochqn_hip_comm_nranks()
    if dist is not None:
        elapsed = max_over_ranks(ctx, elapsed)
        gathered = [torch.zeros(1, dtype=f64, device=cpu_or_dev) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([elapsed_local], dtype=f64, device=cpu_or_dev))
        per_rank_ms = [round(1e3 * float(g.item()) / args.steps, 3) for g in gathered]
        tn = torch.tensor([rccl_nranks], dtype=torch.int64, device=cpu_or_dev)
        dist.all_reduce(tn, op=dist.ReduceOp.MIN)
        rccl_nranks = int(tn.item())
    f1 = wl.objective()
    assert np.isfinite(f1) and f1 < f0, "optimiser diverged on the synthetic quadratic: %r -> %r" % (f0, f1)
    timed_counters = dict(wl.counters)
    if args.dump_x:
        np.save("%s.%d.npy" % (args.dump_x, rank), x.cpu().numpy())

    # ---- from here on the primary result is in hand: everything below is auxiliary, admitted leg by leg against the ONE
    # wall-clock budget of the command, and the line is printed no later than that budget whatever a leg does ----------------
    steps_per_s = args.steps / elapsed
    value = steps_per_s * n_total / 1e8
    budget = Budget()
    value_runs = sustained = ref_form = micro = host_leg = cpu = roof = two_loop = prof_elapsed = None
    kern, detail, prof_steps, what = {}, {}, 0, {}
    legs, legs_failed = {}, []
    want_legs = world > 1 and not args.no_extra_legs and args.config == "c3" and args.n <= 0 and not args.strong
    emitted, emit_lock, all_done, stage = threading.Event(), threading.Lock(), threading.Event(), {"name": "after the timed region", "since": time.time()}
    par = ("n sharded over %d GPU(s), one process per GPU; one RCCL all-reduce per dot product" % world)
    if args.rehearse:
        par = "REHEARSAL: %d ranks sharing one GPU, all-reduce over gloo -- not a measurement" % world
    elif reducer is not None and reducer != "rccl":
        par = ("n sharded over %d GPU(s), one process per GPU; RCCL COULD NOT BE BROUGHT UP: every reduction (a few dozen doubles, "
               "three per step) goes device -> host -> gloo -> device instead" % world)

    def build_out():
        return {
            "metric": "optimizer steps/sec + achieved HBM GB/s, two-loop at n=10^8 m=20 fp64",
            "value": round(value, 3),
            "unit": "steps/s" if n_total == 100_000_000 else "steps/s normalised to n=1e8 (steps/s * n_total/1e8)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SQN n=%g per GPU (n_total=%g), m=%d, L=%d, Hessian-vector pairs via A'(Av)/%d, "
                                   "check_nan=1, ring full, fp64" % (n_gpu, n_total, m, L, bs),
                       "name": args.config if args.n <= 0 else "custom",
                       "parallelism": par,
                       "inputs": "counter-based generator (stochqn_hip_synth_*, seed %d): shard-invariant" % SEED,
                       "calls": timed_counters["calls"], "hess_vec_requests": timed_counters["hv"],
                       "rejected_steps": timed_counters["bad"], "rejected_pairs": timed_counters["rejected"],
                       "options": args.opt,
                       "deviation_from_survey_8d": "Hessian mini-batch: the 32 sample vectors have disjoint supports and are scaled by the size of "
                                                   "their class (a_k,i = 32 sqrt(d_i / n_total) for i = k mod 32, else 0; stored dense, streamed in full): "
                                                   "A'A/32 = D^(1/2) P D^(1/2) with P the projector onto the 32 class indicators -- rank 32, eigenvalues ~ mean(d), "
                                                   "below the objective's Hessian D = diag(d).  SURVEY 8d writes a_k,i = sqrt(d_i)(1 + 0.1(2u - 1)) for every i, "
                                                   "whose A'A/32 is one rank-32 matrix of norm ~ n (gamma = s'y/y'y ~ 1/n: the step collapses; rounds 1 - 4 "
                                                   "had the same defect by a factor n/32 and stalled at 0.37 f0).  Same storage, same traffic (PMC: 27.2 GB per "
                                                   "pass).  With this batch the objective falls geometrically for as long as doubles can hold it (~9000 steps; "
                                                   "the CPU oracle does the same: profiles/r05_oracle_long_run.log); legs after the timed region start again from "
                                                   "the initial state before that (`workload_resets`).",
                       "workload_resets": wl.resets,
                       "f_start": f0, "f_end": f1},
            "rccl_nranks": 0 if (reducer or "").startswith("gloo (the library") else rccl_nranks,      # ranks of the communicator the reductions used
            "reducer": reducer,
            "per_rank_ms_per_step": per_rank_ms,
            "steps_per_s_unnormalised": round(steps_per_s, 3),
            "profiled_pass": None if prof_elapsed is None else {"steps": prof_steps, "ms_per_step": round(1e3 * prof_elapsed / prof_steps, 3)},
            "sustained": sustained,
            "value_runs": value_runs,
            "roofline": roof,
            "two_loop": two_loop,
            "two_loop_micro": micro,
            "reference_form": ref_form,
            "kernels": detail,
            "forms": {"three_pass": forms["steps_three_pass"], "sweeps": forms["steps_sweeps"],
                      "sweeps_because_of_kappa": forms["steps_kappa_fallback"], "no_pairs_yet": forms["steps_plain"],
                      "note": "steps of the timed region by the form of the two-loop recursion that ran (stochqn_hip_stat)"},
            "allreduces_per_step": round(forms["allreduces"] / args.steps, 2),
            "host_caller": host_leg,
            "cpu_baseline": cpu,
        }

    def emit(extra_failed=(), watchdog_fired=None):
        """THE line (once).  Normally at the very end; from the watchdog when a leg hangs or the budget runs out, with what there is."""
        with emit_lock:
            if emitted.is_set():
                return
            out = build_out()
            # a reader who looks at `value` alone must see at a glance when RCCL did not produce it: the reductions of a multi-rank
            # run went over gloo (RCCL could not be brought up, or a rehearsal), or an auxiliary leg hung and was cut off
            why = []
            if reducer is not None and reducer != "rccl":
                why.append("the reductions of this run did not go over RCCL: %s" % reducer)
                out["rccl_nranks"] = 0
            if watchdog_fired:
                why.append("the watchdog cut the run off (%s): what follows the primary result is incomplete" % watchdog_fired)
            out["degraded"] = bool(why)
            if why:
                out["degraded_because"] = why
            if want_legs:
                out["legs"] = dict(legs)
                out["legs_failed"] = legs_failed + list(extra_failed)
            out["legs_skipped"] = [s["leg"] for s in budget.skipped]
            out["budget"] = dict(budget.report(), skipped=list(budget.skipped))
            if args.config == "c5" or n_gpu == CONFIGS["c5"]:
                out["shard_reference_1gpu"] = shard_reference(world, steps_per_s, None)
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
            emitted.set()

    def watchdog(leg_limit):
        # The primary result is in hand.  Two things must not take it along: a collective of an auxiliary leg that never completes
        # (first contact with RCCL on N > 1 ranks happens on the driver's node) -- no stage of the multi-rank part may last longer
        # than `leg_limit` -- and the command's budget running out under a leg that cannot be interrupted.  Every rank leaves at
        # that moment, rank 0 prints first: the line says "degraded": true.  With --strict-legs a hung leg means no line, code 3.
        while not all_done.wait(0.25):
            now = time.time()
            hung = want_legs and stage.get("collective") and now - stage["since"] > leg_limit
            spent = now >= budget.deadline - 2.0
            if not (hung or spent):
                continue
            why = ("leg '%s' had not finished after %g s" % (stage["name"], leg_limit)) if hung else \
                  ("the budget of %g s ran out during '%s'" % (budget.deadline - ENTRY_EPOCH, stage["name"]))
            if hung and args.strict_legs:
                sys.stderr.write("bench.py: rank %d: %s (--strict-legs: no result line)\n" % (rank, why))
                kill_children()
                os._exit(3)
            if rank == 0:
                if not hung:
                    budget.skip(stage["name"], "was running when the budget ran out")
                emit(["watchdog: %s; nothing after it was run" % why] if hung else [], watchdog_fired=why)
            else:
                time.sleep(3.0)
            kill_children()
            os._exit(0)

    threading.Thread(target=watchdog, args=(float(os.environ.get("BENCH_WATCHDOG_S", "300")),), daemon=True).start()

    def admitted(leg, cost=None, collective=True):
        """Is there room for `leg`?  Every rank must take the same turn: the leg runs only when it fits on ALL ranks."""
        ok = budget.fits(leg, cost)
        if collective:
            ok = all_ok(ctx, ok)
        if not ok:
            budget.skip(leg)
        stage.update(name=leg, since=time.time())
        return ok

    class on_the_clock:
        def __init__(self, leg):
            self.leg = leg

        def __enter__(self):
            self.t = time.time()

        def __exit__(self, *exc):
            budget.note(self.leg, time.time() - self.t)
            stage.update(name="after " + self.leg, since=time.time())
            return False

    # ---- second pass, same workload, every launch bracketed by HIP events on the library's stream:
    # per-kernel durations -> roofline of the dominant kernel (first of the auxiliary legs: the line's contract needs it) ----
    if not args.no_profile and admitted("profile"):
        with on_the_clock("profile"):
            prof_steps = max(L, min(args.steps, 4 * L)) // L * L                # whole L-cycles: the pair-building calls in proportion
            wl.keep_stable(prof_steps + 12)                                     # + the reference-form steps below
            lib.stochqn_hip_profile_enable(1)
            lib.stochqn_hip_profile_reset()
            barrier(ctx)
            t1 = time.perf_counter()
            wl.steps(prof_steps)
            barrier(ctx)
            prof_elapsed = time.perf_counter() - t1
            lib.stochqn_hip_profile_enable(0)
            kern = kernel_table(lib)
    detail, roof, two_loop, what = analyse_kernels(kern, n_gpu, m, bs, prof_steps, 1, config=args.config)

    # ---- the K-step region again, on a gradient array that was freed and allocated anew (another placement): the spread ----
    if args.value_runs > 1 and admitted("value_runs"):
        with on_the_clock("value_runs"):
            vals = [args.steps / elapsed * n_total / 1e8]
            for _ in range(args.value_runs - 1):
                wl.grad = None
                torch.cuda.empty_cache()
                wl.pad = torch.empty(int(7 + 64 * len(vals)) << 18, dtype=f64, device=dev)     # shifts where the new array lands
                wl.grad = torch.empty(n, dtype=f64, device=dev)
                wl.keep_stable(L + args.steps)
                wl.steps(L)
                barrier(ctx)
                tv = time.perf_counter()
                wl.steps(args.steps)
                barrier(ctx)
                vals.append(args.steps / max_over_ranks(ctx, time.perf_counter() - tv) * n_total / 1e8)
            wl.pad = None
            sv = sorted(vals)
            value_runs = {"values": [round(v, 3) for v in vals], "min": round(sv[0], 3), "median": round(sv[len(sv) // 2], 3), "max": round(sv[-1], 3),
                          "note": "`value` is values[0]; before each later repetition the caller's gradient array was re-allocated"}

    # ---- the same workload over seconds (K is the driver's choice and may last a quarter of a second): every rank
    # derives the same number of steps from the max-over-ranks time of the K steps above ---------------------------------
    if args.sustain_seconds > 0 and admitted("sustained", args.sustain_seconds + 4.0):
        with on_the_clock("sustained"):
            extra = max(L, int(args.sustain_seconds / (elapsed / args.steps)) // L * L)
            extra = min(extra, wl.STABLE_STEPS - wl.warm) // L * L
            wl.keep_stable(extra)
            barrier(ctx)
            ts = time.perf_counter()
            wl.steps(extra)
            barrier(ctx)
            dt = max_over_ranks(ctx, time.perf_counter() - ts)
            sustained = {"steps": extra, "seconds": round(dt, 3), "ms_per_step": round(1e3 * dt / extra, 3),
                         "value": round(extra / dt * n_total / 1e8, 3)}

    # ---- outside the timed region: the same workload in the reference's own dependency structure
    # (2m+1 dependent fused sweeps, 64*m*n algorithmic bytes) for the roofline the north star names --
    if "sadd" in kern and not args.no_reference_form and admitted("reference_form"):
        with on_the_clock("reference_form"):
            wl.keep_stable(12)
            lib.stochqn_hip_set_option(b"threepass", 0.0)
            wl.steps(2)
            lib.stochqn_hip_profile_enable(1)
            lib.stochqn_hip_profile_reset()
            barrier(ctx)
            t1 = time.perf_counter()
            extra = 10
            wl.steps(extra)
            barrier(ctx)
            el2 = time.perf_counter() - t1
            lib.stochqn_hip_profile_enable(0)
            lib.stochqn_hip_set_option(b"threepass", 1.0)
            k2 = kernel_table(lib)
            if "bwd" in k2:
                cnt, ms = k2["bwd"]
                ach = 4 * n_gpu * 8 / (ms / cnt * 1e-3) / 1e9
                tl = sum(k2[k][1] for k in ("first", "bwd", "mid", "fwd", "fwd_last") if k in k2) / extra
                tr, src = pmc_traffic("bwd", n_gpu, m, 4 * n_gpu * 8, config=args.config)
                ref_form = {"note": "same workload with --opt threepass=0, %d steps after the timed region" % extra,
                            "steps_per_s": round(extra / el2 * n_total / 1e8, 3),
                            "two_loop_ms": round(tl, 3), "two_loop_alg_bytes": 64 * m * n_gpu,
                            "two_loop_alg_GBps": round(64.0 * m * n_gpu / (tl * 1e-3) / 1e9, 1),
                            "two_loop_frac_of_8TBps": round(64.0 * m * n_gpu / (tl * 1e-3) / 1e9 / PEAK, 4),
                            "roofline": {"bound": "hbm", "kernel": "bwd (%s)" % what["bwd"], "achieved": round(ach, 1),
                                         "peak": PEAK, "unit": "GB/s", "frac": round(ach / PEAK, 4), "traffic": tr,
                                         "traffic_source": src, "alg_bytes_per_launch": 4 * n_gpu * 8,
                                         "avg_launch_ms": round(ms / cnt, 4)}}

    # ---- the two-loop recursion on its own (SURVEY.md 8d "two-loop micro-benchmark"): the ring as the
    # run left it (m pairs, oldest in row mem_st_ix), H0 = NULL, h0 = 0; 3 warm-up + 20 timed calls of
    # stochqn_hip_two_loop per form, median wall clock of the synchronous call --------------------------
    if not args.no_reference_form and admitted("two_loop_micro"):
        with on_the_clock("two_loop_micro"):
            lib.stochqn_hip_two_loop.restype = C.c_int
            lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                                 C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
            g0 = wl.uniform(torch.empty(n, dtype=f64, device=dev), 2, 0, -0.5, 1.0)          # g_i = u(i,2,t) - 0.5
            gq = torch.empty_like(g0)
            micro = {"note": "stochqn_hip_two_loop alone: mem_used=%d, oldest pair in row %d, H0=NULL, 3 warm-up + 20 calls, median; "
                             "reference-form bytes = SURVEY 8d's 64*m*n (the sweeps form moves exactly those; the three-pass form moves "
                             "(3m+5)*n*8)" % (m, wl.b.mem_st_ix)}
            lib.stochqn_hip_set_option(b"raw_reuse_cache", 1.0)      # S and Y do not change between these calls
            for form, three in (("three_pass", 1.0), ("sweeps", 0.0)):
                lib.stochqn_hip_set_option(b"threepass", three)
                ts = []
                for rep in range(23):
                    gq.copy_(g0)
                    barrier(ctx)
                    tq = time.perf_counter()
                    rc = lib.stochqn_hip_two_loop(gq.data_ptr(), n, None, 0.0, Y.data_ptr(), S.data_ptr(), m, m, wl.b.mem_st_ix,
                                                  wl.rho_h.ctypes.data, wl.alpha_h.ctypes.data)
                    assert rc == 0
                    ts.append(time.perf_counter() - tq)
                med = max_over_ranks(ctx, sorted(ts[3:])[10])
                moved = {"three_pass": 3 * m + 5, "sweeps": 8 * m}[form] * n * 8        # bytes this form has to stream
                micro[form] = {"median_ms": round(1e3 * med, 3), "bytes_moved": moved,
                               "GBps_on_bytes_moved": round(moved / med / 1e9, 1),
                               "frac_of_8TBps_on_bytes_moved": round(moved / med / 1e9 / PEAK, 4),
                               "GBps_on_reference_form_bytes": round(64.0 * m * n / med / 1e9, 1)}
            lib.stochqn_hip_set_option(b"threepass", 1.0)
            lib.stochqn_hip_set_option(b"raw_reuse_cache", 0.0)
            lib.stochqn_hip_release(C.c_void_p(S.data_ptr()))      # the raw context keyed by S; the optimiser is finished
            del g0, gq

    # ---- N = 1 only: the reference's real callers own their arrays in HOST memory (R / numpy), so the same step is also
    # timed PCIe-inclusive, and the CPU baseline runs the oracle on the host cores; both work on one host copy of the inputs.
    # The CPU baseline goes first: the line's contract asks for it, the host-caller variants are extra --
    if rank == 0 and world == 1 and not (args.no_cpu_baseline and args.no_host_caller):
        want_cpu = not args.no_cpu_baseline and budget.fits("cpu_baseline", budget.costs["cpu_baseline"] + budget.costs["host_copies"])
        want_host = not args.no_host_caller and budget.fits("host_caller", budget.costs["host_caller"] + budget.costs["host_copies"]
                                                            + (budget.costs["cpu_baseline"] if want_cpu else 0.0))
        for leg, asked, got in (("cpu_baseline", not args.no_cpu_baseline, want_cpu), ("host_caller", not args.no_host_caller, want_host)):
            if asked and not got:
                budget.skip(leg)
        if want_cpu or want_host:
            stage.update(name="host_copies", since=time.time())
            with on_the_clock("host_copies"):
                lib.stochqn_hip_release_all()
                wl.fill_ring()                                         # the state the GPU leg started from
                wl.uniform(x, ST_X0, 0, 1.0, 1.0)
                torch.cuda.synchronize()
                gpu = {"S": S, "Y": Y, "A": A, "d": d, "x": x, "noise": lambda t, out: wl.uniform(out, ST_NOISE, t, 0.99, 0.02)}
                hostc = HostCopies(args, gpu, n, m, bs, need_batch=want_cpu)
            if want_cpu:
                stage.update(name="cpu_baseline", since=time.time())
                with on_the_clock("cpu_baseline"):
                    cpu = cpu_baseline(args, gpu, hostc, n, m, L, bs, wl.step_size, budget)
            if want_host and admitted("host_caller", collective=False):
                with on_the_clock("host_caller"):
                    host_leg = host_caller_leg(args, lib, be, hostc, gpu, n, m, L, wl.step_size, two_loop, budget)
            del hostc
            gpu = None

    # no leg after the timed region may have had a step rejected: a rejected step skips the update, and its time would be that of less work
    assert wl.counters["bad"] == timed_counters["bad"], "a leg after the timed region had %d step(s) rejected" % (wl.counters["bad"] - timed_counters["bad"])

    # ---- N > 1 with no further flags: what BASELINE config 5 and SURVEY 8e ask for, in the same line ----
    wl.free()
    del x, S, Y, A, d
    if want_legs:
        leg_steps = max(20, args.steps)

        def attempt(name, fn):
            if not admitted(name):
                return
            stage["collective"] = True
            res, ok = None, True
            t_leg = time.time()
            try:
                if os.environ.get("BENCH_TEST_HANG_LEG") == name:      # tests: a leg that never comes back
                    time.sleep(1e6)
                res = fn()
            except Exception as e:                              # a failing leg is reported, it does not take the headline down with it
                res, ok = {"error": "%s: %s" % (type(e).__name__, e)}, False
            if all_ok(ctx, ok):
                legs[name] = res
            else:
                legs[name] = res if not ok else {"error": "failed on another rank"}
                legs_failed.append(name)
            budget.note(name, time.time() - t_leg)
            stage.update(name="after " + name, since=time.time())

        attempt("c5", lambda: timed_leg(ctx, config_n("c5"), m, L, bs, leg_steps, args.warmup,
                                        "BASELINE config 5's weak-scaling point: n = 1.25e8 per GPU (n_total = 1e9 on 8 GPUs)"))
        attempt("strong", lambda: timed_leg(ctx, config_n("c3") // world, m, L, bs, leg_steps, args.warmup,
                                            "SURVEY 8e strong scaling: the n = 1e8 problem split over the GPUs"))
        attempt("allreduce_us", lambda: allreduce_latency(ctx))
        if "allreduce_us" in legs and "allreduce_us" not in legs_failed and forms:
            legs["allreduce_us"]["allreduces_per_step"] = round(forms["allreduces"] / args.steps, 2)

    if dist is not None:
        stage.update(name="teardown of the communicators", since=time.time(), collective=True)
        barrier(ctx)
        lib.stochqn_hip_comm_finalize()
        dist.destroy_process_group()
    stage.update(name="after the ranks' part", since=time.time(), collective=False)
    lib.stochqn_hip_release_all()
    if rank != 0:
        all_done.set()
        return                                                 # rank 0 alone starts the children and prints

    if want_legs:
        # SURVEY 8e's own process model: ONE host process driving all N devices (ncclCommInitAll, one thread per device).
        # A fresh child, started when the other ranks are on their way out; this process keeps only an idle HIP context.
        if admitted("in_process", collective=False):
            with on_the_clock("in_process"):
                try:
                    torch.cuda.empty_cache()
                    time.sleep(3.0)
                    legs["in_process"] = in_process_leg(args, world, n_gpu, max(20, args.steps), budget)
                except Exception as e:
                    legs["in_process"] = {"error": "%s: %s" % (type(e).__name__, e)}
                    legs_failed.append("in_process")
        # C5's yardstick -- ONE GPU at the same per-GPU shard -- measured on THIS node, now that the ranks are gone: the
        # "within 15 % of linear" verdict compares two numbers of the same box and the same minute (boxes differ by 5-8 %)
        if "c5" in legs and "c5" not in legs_failed:
            here = None
            if admitted("c5_yardstick", collective=False):
                with on_the_clock("c5_yardstick"):
                    try:
                        here = c5_yardstick_here(args, config_n("c5"), max(20, args.steps), budget)
                    except Exception as e:
                        here = {"error": "%s: %s" % (type(e).__name__, e)}
            else:
                here = {"error": "skipped: did not fit into what was left of the budget"}
            ref = shard_reference(world, legs["c5"]["steps_per_s"], here)
            legs["c5"]["shard_reference_1gpu"] = ref
            if ref and "steps_per_s" in ref:
                legs["c5"]["this_run_over_reference"] = ref.get("this_run_over_reference")
                legs["c5"]["within_15pct_of_linear"] = bool(legs["c5"]["steps_per_s"] >= ref["within_15pct_means_at_least"])

    # ---- the dominant kernel's HBM traffic counted NOW (PMC), not looked up: at N = 1 in this process's own time, at N > 1 on
    # rank 0's device once the ranks have gone (a per-GPU shard of the weak-scaling run IS the one-GPU problem) --------------
    if roof and not args.no_live_pmc and not args.rehearse:
        dom = roof["kernel"].split()[0]
        live = None
        if admitted("live_pmc", collective=False):
            with on_the_clock("live_pmc"):
                live = live_pmc(args, dom, n_gpu, budget)
        if live:
            roof["traffic"] = live["bytes"]
            roof["traffic_source"] = live["source"]
            roof["traffic_read_bytes"], roof["traffic_write_bytes"], roof["traffic_dispatches"] = live["read_bytes"], live["write_bytes"], live["dispatches"]
            roof["traffic_over_algorithmic"] = round(live["bytes"] / roof["alg_bytes_per_launch"], 4)
        else:
            roof["traffic_note"] = "live PMC passes unavailable here (or no room left for them): " + \
                                   ("committed profile quoted" if roof.get("traffic") else "no committed profile of this kernel instantiation passes the cross-check either")

    if want_legs and legs_failed and args.strict_legs:
        sys.stderr.write("bench.py: leg(s) %s failed: %s\n" % (", ".join(legs_failed), json.dumps({k: legs[k] for k in legs_failed})))
        raise SystemExit(3)
    emit()
    all_done.set()


def c5_yardstick_here(args, n_gpu, steps, budget):
    """`bench.py --gpus 1` at config 5's per-GPU shard as a fresh child on this node's GPU 0 (timed region only: no profiler
    pass, no PMC, no host legs): the 1-GPU point of the weak-scaling curve, from the same box as the N-GPU point."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--vars-per-gpu", str(n_gpu), "--steps", str(steps), "--warmup", str(args.warmup),
           "--mem", str(args.mem), "--upd-freq", str(args.upd_freq), "--bsize", str(args.bsize), "--no-cpu-baseline", "--no-host-caller",
           "--no-live-pmc", "--no-reference-form", "--no-profile", "--sustain-seconds", "0", "--value-runs", "1"]
    for kv in args.opt:
        cmd += ["--opt", kv]
    rc, so, se = run_child(cmd, child_env(budget), budget.child_timeout(300))
    lines = [l for l in so.splitlines() if l.startswith('{"metric"')]
    if rc != 0 or not lines:
        raise RuntimeError("%s: %s" % ("timed out" if rc is None else "exit code %d" % rc, se[-600:].replace("\n", " | ")))
    d = json.loads(lines[0])
    return {"steps_per_s": d["steps_per_s_unnormalised"], "ms_per_step": d["ms_per_step"], "n_per_gpu": n_gpu, "steps": d["steps"]}


def in_process_leg(args, world, n_gpu, steps, budget):
    """`bench.py --gpus N --in-process` as a fresh child process: the same per-GPU problem driven by ONE host process through
    the library's single-process multi-device mode (group.cpp; SURVEY 8e's process model)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--in-process", "--vars-per-gpu", str(n_gpu),
           "--steps", str(steps), "--warmup", str(args.warmup), "--mem", str(args.mem), "--upd-freq", str(args.upd_freq),
           "--bsize", str(args.bsize), "--no-cpu-baseline"]
    if args.rehearse:
        cmd.append("--virtual-devices")
    if args.no_host_caller:
        cmd.append("--no-host-caller")
    for kv in args.opt:
        cmd += ["--opt", kv]
    # a child that hangs must not cost the line its primary result: it gets what is left of the budget, at most 420 s.  The
    # child's host-caller part comes last and is the first thing to go when time is short
    if not budget.fits("in_process", budget.costs["in_process"] + 30.0) and "--no-host-caller" not in cmd:
        cmd.append("--no-host-caller")
    rc, so, se = run_child(cmd, child_env(budget), budget.child_timeout(420))
    lines = [l for l in so.splitlines() if l.startswith('{"metric"')]
    if rc != 0 or not lines:
        raise RuntimeError("%s: %s" % ("timed out" if rc is None else "exit code %d" % rc, se[-600:].replace("\n", " | ")))
    d = json.loads(lines[0])
    return {"what": "one host process, %d device shards behind the plain ABI (option devices): %s" % (world, d["config"]["parallelism"]),
            "n_per_gpu": n_gpu, "n_total": n_gpu * world, "steps": d["steps"], "ms_per_step": d["ms_per_step"],
            "steps_per_s": d["steps_per_s_unnormalised"], "value_normalised_to_1e8": d["value"], "rccl_nranks": d["rccl_nranks"],
            "device_shards": d["device_shards"], "hess_vec_requests": d["config"]["hess_vec_requests"],
            "rejected_steps": d["config"]["rejected_steps"], "allreduces_per_step": d.get("allreduces_per_step"),
            "allreduce_us": d.get("allreduce_us"), "host_caller": d.get("host_caller")}


def run_in_process(args):
    """`--gpus N --in-process`: ONE host process, the library's single-process multi-device mode (group.cpp), a
    device-resident caller: initialize_SQN(n_total) hands out a workspace sharded over the N devices, the
    caller keeps x / grad / hess_vec as per-device slices (stochqn_hip_devices_layout / _bind), calls run_SQN
    once per step and does its own per-shard work -- the synthetic gradient, the Hessian-vector product
    A'(Av)/bs through stochqn_hip_fisher_product -- on the shards' threads (stochqn_hip_devices_foreach), so
    that the product's reduction spans the shards.  Same problem as the one-process-per-GPU run (counter-
    based inputs); the ring is filled by running the optimiser itself (m*L untimed steps) because the arrays
    of a library-owned sharded workspace are not the caller's to pre-fill."""
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import stochqn_amd
    P = args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the library has no CPU path)")
    if torch.cuda.device_count() < P and not args.virtual_devices:
        raise SystemExit("bench.py: --gpus %d --in-process asked for but only %d device(s) are visible "
                         "(--virtual-devices rehearses the mode on fewer)" % (P, torch.cuda.device_count()))
    lib = stochqn_amd.cdll()
    be = stochqn_amd.lib()
    assert lib.stochqn_hip_available() == 1
    lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    for kv in args.opt:
        name, val = kv.split("=")
        assert lib.stochqn_hip_set_option(name.encode(), float(val)) == 0, kv
    if args.virtual_devices:
        assert lib.stochqn_hip_set_option(b"virtual_devices", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices_min_n", 1.0) == 0
    assert lib.stochqn_hip_set_option(b"devices", float(P)) == 0

    n_gpu = args.n if args.n > 0 else CONFIGS[args.config]
    n_total = n_gpu * P
    m, L, bs = args.mem, args.upd_freq, args.bsize
    f64 = torch.float64
    u64 = C.c_ulonglong
    vp = C.c_void_p
    lib.stochqn_hip_synth_uniform.argtypes = [vp, C.c_size_t, u64, u64, u64, u64, C.c_double, C.c_double]
    lib.stochqn_hip_synth_noisy_grad.argtypes = [vp, vp, vp, C.c_size_t, u64, u64, u64, u64, C.c_double]
    lib.stochqn_hip_synth_batch_row.argtypes = [vp, vp, C.c_size_t, u64, C.c_uint, C.c_uint]
    lib.stochqn_hip_fisher_product.argtypes = [vp, C.c_size_t, C.c_int, vp, vp, vp]
    lib.stochqn_hip_devices_active.argtypes = [vp]
    lib.stochqn_hip_devices_reducer.argtypes = [vp]
    lib.stochqn_hip_devices_layout.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.stochqn_hip_devices_bind.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.stochqn_hip_devices_request.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]
    SHARD_FN = C.CFUNCTYPE(None, vp, C.c_int, C.c_int, C.c_size_t, C.c_size_t)
    lib.stochqn_hip_devices_foreach.argtypes = [vp, SHARD_FN, vp]

    w = be.initialize_SQN(n_total, m, L, 0.0, 0, 0.0, 1, 1)
    assert bool(w), "initialize_SQN(n = %d) on %d device shards failed" % (n_total, P)
    key = vp(w.contents.bfgs_memory.contents.s_mem)
    assert lib.stochqn_hip_devices_active(key) == P
    reducer = {1: "RCCL (ncclCommInitAll)", 3: "host-side rendezvous (virtual devices)"}.get(lib.stochqn_hip_devices_reducer(key), "?")

    shards = []
    for p in range(P):
        dv, off, cnt = C.c_int(), C.c_size_t(), C.c_size_t()
        assert lib.stochqn_hip_devices_layout(key, p, C.byref(dv), C.byref(off), C.byref(cnt)) == 0
        dev = torch.device("cuda", dv.value)
        torch.cuda.set_device(dev)                       # the synth kernels go to the current device's null stream
        k = cnt.value
        sh = {"dev": dev, "off": off.value, "cnt": k}
        for name in ("d", "x", "grad", "hv"):
            sh[name] = torch.empty(k, dtype=f64, device=dev)
        sh["t"] = np.zeros(bs)
        assert lib.stochqn_hip_synth_uniform(sh["d"].data_ptr(), k, off.value, SEED, ST_D, 0, 0.5, 1.0) == 0
        assert lib.stochqn_hip_synth_uniform(sh["x"].data_ptr(), k, off.value, SEED, ST_X0, 0, 1.0, 1.0) == 0
        sh["A"] = torch.empty(bs * k, dtype=f64, device=dev)
        d_batch = sh["d"] * (float(bs) / float(n_total))              # rows scaled by the size of their class: Workload.__init__
        for r in range(bs):
            assert lib.stochqn_hip_synth_batch_row(sh["A"].data_ptr() + 8 * r * k, d_batch.data_ptr(), k, off.value, r, bs) == 0
        del d_batch
        assert lib.stochqn_hip_devices_bind(key, p, sh["x"].data_ptr(), sh["grad"].data_ptr(), sh["hv"].data_ptr()) == 0
        shards.append(sh)
    for sh in shards:
        torch.cuda.synchronize(sh["dev"])
    torch.cuda.set_device(shards[0]["dev"])

    req, req_vec, task, info = vp(), vp(), C.c_int(0), C.c_int(200)
    dummy = np.zeros(1)                                  # run_SQN wants non-NULL x / grad / hess_vec; bound shards never read them
    cur = {"task": 0, "t": 0}
    counters = {"calls": 0, "hv": 0, "bad": 0, "rejected": 0}
    errors = []

    def shard_work(user, p, device, off, cnt):           # on shard p's own thread, its device current, its reducer bound
        try:
            sh = shards[p]
            rq, rqv = vp(), vp()
            assert lib.stochqn_hip_devices_request(key, p, C.byref(rq), C.byref(rqv)) == 0
            if cur["task"] == 101:
                assert lib.stochqn_hip_synth_noisy_grad(sh["grad"].data_ptr(), sh["d"].data_ptr(), rq.value, cnt, off, SEED, ST_NOISE, cur["t"], 0.01) == 0
            elif cur["task"] == 104:
                assert lib.stochqn_hip_fisher_product(sh["A"].data_ptr(), bs, cnt, rqv.value, sh["t"].ctypes.data, sh["hv"].data_ptr()) == 0
        except Exception as e:                           # an exception must not escape into the C thread
            errors.append(repr(e))

    work_cb = SHARD_FN(shard_work)

    def call():
        rc = be.run_SQN(0.05, dummy.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, C.byref(req), C.byref(req_vec),
                        C.byref(task), w, C.byref(info))
        assert rc in (0, 1), rc
        counters["calls"] += 1
        counters["bad"] += info.value == 203
        counters["rejected"] += info.value == 202

    def one_step(t):
        target = w.contents.niter + 1
        while w.contents.niter < target:
            cur["task"], cur["t"] = task.value, t
            if task.value == 104:
                counters["hv"] += 1
            assert lib.stochqn_hip_devices_foreach(key, work_cb, None) == 0 and not errors, errors
            call()

    def sync_all():
        for sh in shards:
            torch.cuda.synchronize(sh["dev"])

    def objective():
        return sum(float(0.5 * torch.sum(sh["d"] * sh["x"] * sh["x"])) for sh in shards)

    call()                                               # section 0: "give me a gradient at x"
    f0 = objective()
    t_idx = 0
    fill = m * L + L                                     # the optimiser fills its own ring: m pairs, one every L steps
    for _ in range(fill + args.warmup):
        one_step(t_idx)
        t_idx += 1
    assert w.contents.bfgs_memory.contents.mem_used == m, "the ring did not fill up"
    for k in counters:
        counters[k] = 0                                  # report the timed region's calls, not the ring filling
    lib.stochqn_hip_profile_enable(0)
    sync_all()
    st0 = stats(lib)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step(t_idx)
        t_idx += 1
    sync_all()
    elapsed = time.perf_counter() - t0
    forms = stat_diff(st0, stats(lib))
    f1 = objective()
    assert np.isfinite(f1) and f1 < f0, "optimiser diverged on the synthetic quadratic: %r -> %r" % (f0, f1)
    timed_counters = dict(counters)

    kern, prof_elapsed, prof_steps = {}, None, 0
    if not args.no_profile:
        prof_steps = max(L, min(args.steps, 4 * L)) // L * L
        lib.stochqn_hip_profile_enable(1)
        lib.stochqn_hip_profile_reset()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(prof_steps):
            one_step(t_idx)
            t_idx += 1
        sync_all()
        prof_elapsed = time.perf_counter() - t1
        lib.stochqn_hip_profile_enable(0)
        kern = kernel_table(lib)
    detail, roof, two_loop, _ = analyse_kernels(kern, n_gpu, m, bs, prof_steps, P)
    steps_per_s = args.steps / elapsed

    # latency of one reduction between the shards, on the shards' own threads and communicators
    probe = {}
    lib.stochqn_hip_comm_allreduce_probe.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]

    def shard_probe(user, p, device, off, cnt):
        try:
            med, mn = C.c_double(), C.c_double()
            rc = lib.stochqn_hip_comm_allreduce_probe(20, 200, C.byref(med), C.byref(mn))
            probe[p] = (rc, med.value, mn.value)
        except Exception as e:
            errors.append(repr(e))
    probe_cb = SHARD_FN(shard_probe)
    allreduce_us = None
    if lib.stochqn_hip_devices_foreach(key, probe_cb, None) == 0 and len(probe) == P and all(v[0] == 0 for v in probe.values()):
        allreduce_us = {"median_us": round(max(v[1] for v in probe.values()), 2), "min_us": round(max(v[2] for v in probe.values()), 2),
                        "doubles": 20, "reps": 200}
    # ---- the same process model with a HOST caller (R / numpy arrays, the reference's real callers): n cut over the shards, every
    # shard moves ITS slice of grad / x over ITS link -- what one link cannot give (36 ms per step at n = 1e8) P links may ----
    host_leg = None
    if args.dump_x:
        np.save("%s.0.npy" % args.dump_x, np.concatenate([sh["x"].cpu().numpy() for sh in shards]))
    if not args.no_host_caller:
        be.dealloc_SQN(w)
        w = None
        lib.stochqn_hip_release_all()
        for sh in shards:
            for name in ("A", "d", "x", "grad", "hv"):
                sh[name] = None
        torch.cuda.empty_cache()
        try:
            host_leg = in_process_host_leg(args, lib, be, P, [sh["dev"] for sh in shards], m, L)
        except Exception as e:                           # a leg of its own: it must not cost the line its primary result
            host_leg = {"error": "%s: %s" % (type(e).__name__, e)}
    out = {
        "metric": "optimizer steps/sec + achieved HBM GB/s, two-loop at n=10^8 m=20 fp64",
        "value": round(steps_per_s * n_total / 1e8, 3),
        "unit": "steps/s" if n_total == 100_000_000 else "steps/s normalised to n=1e8 (steps/s * n_total/1e8)",
        "n_gpus": P, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "SQN n=%g per GPU (n_total=%g), m=%d, L=%d, Hessian-vector pairs via A'(Av)/%d, "
                               "check_nan=1, ring full (filled by %d untimed steps of the run itself), fp64" % (n_gpu, n_total, m, L, bs, fill),
                   "name": args.config if args.n <= 0 else "custom",
                   "parallelism": "n sharded over %d device shard(s) inside ONE process behind the plain ABI (option devices=%d), "
                                  "device-resident caller (stochqn_hip_devices_bind); reducer: %s%s" % (
                                      P, P, reducer, " -- shards share devices: a rehearsal, not a measurement" if args.virtual_devices else ""),
                   "inputs": "counter-based generator (stochqn_hip_synth_*, seed %d): shard-invariant" % SEED,
                   "calls": timed_counters["calls"], "hess_vec_requests": timed_counters["hv"],
                   "rejected_steps": timed_counters["bad"], "rejected_pairs": timed_counters["rejected"],
                   "options": args.opt, "f_start": f0, "f_end": f1},
        "rccl_nranks": P if reducer.startswith("RCCL") else 1,
        "device_shards": P,
        "allreduces_per_step": round(forms["allreduces"] / args.steps / P, 2),
        "allreduce_us": allreduce_us,
        "forms": {"three_pass": forms["steps_three_pass"] // P, "sweeps": forms["steps_sweeps"] // P,
                  "sweeps_because_of_kappa": forms["steps_kappa_fallback"] // P},
        "steps_per_s_unnormalised": round(steps_per_s, 3),
        "profiled_pass": None if prof_elapsed is None else {"steps": prof_steps, "ms_per_step": round(1e3 * prof_elapsed / prof_steps, 3)},
        "roofline": roof, "two_loop": two_loop, "two_loop_micro": None, "reference_form": None,
        "kernels": detail, "cpu_baseline": None,
        "host_caller": host_leg,
    }
    os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if w is not None:
        be.dealloc_SQN(w)
    lib.stochqn_hip_release_all()


def in_process_host_leg(args, lib, be, P, devices, m, L):
    """ONE host process, every array in host memory (numpy: the R / Cython protocol, reference src/Rwrapper.c:98-125,
    stochqn/pywrapper.pxi:161-207), the library's single-process multi-device mode: shard p moves its slice of grad and x over
    device p's own link.  Reported: what a pinned transfer gives per device (one at a time, and all devices at once), and per
    ordinary step the bytes each shard moved and the rate that makes per link.  n_total = 1e8 (the headline problem), the five
    per-call arrays page-locked by their owner (stochqn_hip_pin_host, as stochqn_amd/free.py does); a short run: 3 L-cycles
    from an empty ring -- the link, not the two-loop, is what is being measured."""
    import numpy as np
    import torch
    from stochqn_amd import _abi
    n = args.host_n if args.host_n > 0 else (100_000_000 if not args.virtual_devices else 4_000_000 * P)
    per_dev = {}
    for p, dev in enumerate(devices if not args.virtual_devices else devices[:1]):
        torch.cuda.set_device(dev)
        per_dev[str(dev)] = pcie_probe(n // P * 8, dev)
    # all links at once: one thread per device
    together = None
    if not args.virtual_devices and P > 1:
        res = [None] * P

        def probe(p):
            torch.cuda.set_device(devices[p])
            res[p] = pcie_probe(n // P * 8, devices[p])
        ths = [threading.Thread(target=probe, args=(p,)) for p in range(P)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if all(res):
            together = {"h2d_GBps_sum": round(sum(r["h2d_GBps"] for r in res), 1), "d2h_GBps_sum": round(sum(r["d2h_GBps"] for r in res), 1),
                        "per_device": res}
    torch.cuda.set_device(devices[0])
    lib.stochqn_hip_pin_host.argtypes = [C.c_void_p, C.c_size_t]
    lib.stochqn_hip_unpin_host.argtypes = [C.c_void_p]
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    d = 0.5 + np.random.default_rng(SEED).random(n)
    S, Y = np.zeros(m * n), np.zeros(m * n)              # an empty ring: nothing is imported, the pages are touched by nobody
    x, grad, hv = 1.0 + d, np.empty(n), np.empty(n)
    x_sum, x_avg_prev, rho_h, alpha_h, dummy = np.zeros(n), np.zeros(n), np.zeros(m), np.zeros(m), np.zeros(1)
    pinned = [a for a in (x, grad, hv, x_sum, x_avg_prev) if lib.stochqn_hip_pin_host(a.ctypes.data, a.nbytes) == 0]
    b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho_h.ctypes.data, alpha_h.ctypes.data, dummy.ctypes.data, dummy.ctypes.data, m, 0, 0, L, 0.0, 0.0)
    w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, 0, 0, 1, 1, n)
    req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)
    view = lambda ptr: np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), (n,))
    clock = {"lib": 0.0}

    def one_step():
        target = w.niter + 1
        while w.niter < target or task.value != 101:
            if task.value == 104:
                np.multiply(d, view(req_vec.value), out=hv)
            else:
                np.multiply(d, view(req.value), out=grad)
            t0 = time.perf_counter()
            rc = be.run_SQN(0.05, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
            clock["lib"] += time.perf_counter() - t0
            assert rc in (0, 1), rc
    try:
        rc = be.run_SQN(0.05, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec), C.byref(task), C.byref(w), C.byref(info))
        assert rc == 0
        lib.stochqn_hip_stats_reset()
        for _ in range(2 * L):
            one_step()
        assert lib.stochqn_hip_devices_active(C.c_void_p(S.ctypes.data)) == P
        per = []
        for _ in range(L):
            clock["lib"] = 0.0
            one_step()
            per.append(1e3 * clock["lib"])
        ordinary = sorted(per[1:-1])
        ord_ms = ordinary[len(ordinary) // 2]
        up = down = 8 * n                                   # grad and x up, x down; x rides under the update (full duplex)
        out = {"what": "ONE host process, numpy arrays, %d device shards behind the plain ABI; n_total=%g, m=%d (ring of %d pairs so far), L=%d" % (P, n, m, b.mem_used, L),
               "pcie_probe_per_device_alone": per_dev, "pcie_probe_all_devices_at_once": together,
               "ms_per_step": round(sum(per) / L, 2), "ordinary_step_ms": round(ord_ms, 2), "per_step_ms": [round(v, 2) for v in per],
               "bytes_up_per_shard": 2 * up // P, "bytes_down_per_shard": down // P,
               "link_GBps_per_shard_up": round(2 * up / P / (ord_ms * 1e-3) / 1e9, 1), "link_GBps_per_shard_down": round(down / P / (ord_ms * 1e-3) / 1e9, 1),
               "note": "rates are bytes / the WHOLE ordinary step (kernels included): a lower bound of what each link carried",
               "arrays_pinned_by_the_caller": len(pinned), "x_sent_ahead_of_the_guard": int(lib.stochqn_hip_stat(b"x_sent_ahead"))}
    finally:
        lib.stochqn_hip_release_all()
        for a in pinned:
            lib.stochqn_hip_unpin_host(a.ctypes.data)
    return out


def kernel_table(lib):
    """Per-kernel (launches, total ms) of the library's HIP-event profiler."""
    kern = {}
    lib.stochqn_hip_profile_name.restype = C.c_char_p
    for i in range(lib.stochqn_hip_profile_kernels()):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value:
            kern[lib.stochqn_hip_profile_name(i).decode()] = (cnt.value, ms.value)
    return kern


def analyse_kernels(kern, n_gpu, m, bs, prof_steps, shards, config="c3"):
    """Kernel table -> (per-kernel detail, roofline of the dominant kernel, two-loop summary, kernel descriptions).
    `shards` = device shards whose launches the table aggregates (1 per process except --in-process)."""
    # algorithmic n-words per launch
    words = {"first": 2, "bwd": 4, "mid": 3, "fwd": 4, "fwd_last": 3, "apply": 5,
             "sdot": m + 1, "sdot2": m + 2, "qdot": m + 2, "sadd": m + 2,
             "fisher_t": bs + 1, "fisher_y": bs + 2, "pair_s": 4, "pair_y_hv": 6}
    what = {"bwd": "fused backward sweep: read y_i, q, s_{i-1}; write q",
            "fwd": "fused forward sweep: read s_i, r, y_{i+1}; write r",
            "sdot": "three-pass form, pass 1: read g and the %d rows of S" % m,
            "qdot": "three-pass form, pass 2: read g and the %d rows of Y; write r0" % m,
            "sadd": "three-pass form, pass 3: read r0 and the %d rows of S; write r" % m}
    detail = {}
    for name, (cnt, ms) in kern.items():
        avg = ms / cnt
        e = {"launches": cnt, "avg_ms": round(avg, 4)}
        if name in words:
            e["alg_GBps"] = round(words[name] * n_gpu * 8 / (avg * 1e-3) / 1e9, 1)
        detail[name] = e
    roof = None
    cands = [k for k in what if k in kern]
    if cands:
        dom = max(cands, key=lambda k: kern[k][1])               # largest share of device time
        cnt, ms = kern[dom]
        alg = words[dom] * n_gpu * 8
        ach = alg / (ms / cnt * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom, n_gpu, m, alg, config=config)
        roof = {"bound": "hbm", "kernel": "%s (%s)" % (dom, what[dom]),
                "achieved": round(ach, 1), "peak": PEAK, "unit": "GB/s", "frac": round(ach / PEAK, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": alg, "avg_launch_ms": round(ms / cnt, 4),
                "measured": "HIP events on the library's stream, %d steps of the same workload right after the timed "
                            "region (the timed region itself runs with the event profiler off)" % prof_steps}
    chain = ("first", "bwd", "mid", "fwd", "fwd_last", "coef", "sdot", "sdot2", "qdot", "sadd")
    two_loop_ms = sum(kern[k][1] for k in chain if k in kern) / max(prof_steps * shards, 1)
    two_loop = None
    if two_loop_ms > 0:
        form = "three-pass" if "sadd" in kern else "sweeps"
        own = {"three-pass": 3 * m + 5, "sweeps": 8 * m}[form]      # n-words this form has to move
        two_loop = {"form": form, "ms": round(two_loop_ms, 3),
                    "bytes_moved": own * n_gpu * 8, "GBps_on_bytes_moved": round(own * n_gpu * 8 / (two_loop_ms * 1e-3) / 1e9, 1),
                    "frac_of_8TBps_on_bytes_moved": round(own * n_gpu * 8 / (two_loop_ms * 1e-3) / 1e9 / PEAK, 4),
                    "reference_form_bytes": 64 * m * n_gpu,
                    "effective_GBps_vs_reference_form": round(64.0 * m * n_gpu / (two_loop_ms * 1e-3) / 1e9, 1)}

    return detail, roof, two_loop, what


def shard_reference(n_gpus, steps_per_s, here):
    """C5's yardstick: one GPU at the same per-GPU shard, n = 1.25e8, m = 20.  With one GPU this run IS that
    measurement; with more it is `here` -- a 1-GPU child run on this very node (c5_yardstick_here) -- and only when that
    could not be had the newest committed 1-GPU profile (another box: 5-8 % either way), labelled as the fallback it is."""
    import glob
    if n_gpus == 1:
        return {"steps_per_s": round(steps_per_s, 3), "source": "this run", "within_15pct_means_at_least": round(0.85 * steps_per_s, 3)}
    if here and "steps_per_s" in here:
        sps = here["steps_per_s"]
        return {"steps_per_s": round(sps, 3), "source": "this node", "measured": here, "within_15pct_means_at_least": round(0.85 * sps, 3),
                "this_run_over_reference": round(steps_per_s / sps, 4)}
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c5_shard_1gpu.json")), reverse=True):
        try:
            ref = json.load(open(f))
            sps = ref.get("steps_per_s_unnormalised") or ref["value"] / 1.25
            return {"steps_per_s": round(sps, 3), "source": "FALLBACK, another box: " + os.path.relpath(f, ROOT),
                    "same_node_attempt": here,
                    "within_15pct_means_at_least": round(0.85 * sps, 3),
                    "this_run_over_reference": round(steps_per_s / sps, 4)}
        except (OSError, ValueError, KeyError):
            continue
    return None


# Which optimiser a configuration of this file runs -- and with it which instantiation of the three-pass kernels: pass 2 is
# k_qdot<W, NG, NT, MODE, PH> with MODE 0 (scalar H0: oLBFGS, SQN) or 2 (adaQN: four more streams, 20.0 GB instead of 17.6 GB at
# n = 1e8, m = 20).  A committed PMC file serves a line only if it holds THAT instantiation (the adaQN
# file's pass 2 was quoted for SQN lines -- a "wasted traffic" of 1.136 that does not exist).
CONFIG_PROFILE = {"c3": ("c3", 0), "c5": ("c3", 0), "c2": ("c2", 0), "c4": ("c4", 2)}      # profiles/rNN_<tag>_pmc_traffic.json, k_qdot MODE
PROFILE_N = {"c2": 10_000_000}  # problem size a configuration's committed counters were taken at (every other one: 1e8)
PMC_RATIO_OK = (0.97, 1.06)    # bytes counted / algorithmic bytes outside this band: the file is not about this launch -- refuse it


def pmc_key(kernel, m, mode=0, older=False):
    """The exact kernel instantiation (as rocprofv3 prints it, namespace stripped) a bench leg's `kernel` launches at ring size m:
    W = 2 doubles per pack, NG = ceil(m / 8) row groups, non-temporal row loads, clock-phased stores; pass 1 since round 6 with
    U = 2 adjacent column tiles per iteration for rings of up to 24 pairs (`older`: its name in the profiles of rounds 2 - 5)."""
    ng = (m + 7) // 8
    sdot = "k_rows_dot_all<2, %d, true, 1, false>" % ng if older else "k_rows_dot_all<2, %d, true, 1, false, %d>" % (ng, 2 if ng <= 3 else 1)
    return {"bwd": "BwdOp", "fwd": "FwdOp<true, false, false>", "sdot": sdot,
            "qdot": "k_qdot<2, %d, true, %d, true>" % (ng, mode),
            "sadd": "k_sadd<2, %d, true, true, false>" % ng}.get(kernel)


def live_pmc(args, kernel, n_gpu, budget):
    """HBM bytes per launch of `kernel` counted NOW: two rocprofv3 passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE, separate runs
    with --kernel-trace only, as MI355X_MICROARCH.md prescribes) over a short ONE-GPU child run of this very workload at this
    per-GPU size (at N > 1: on rank 0's device, after the ranks have gone); FETCH_SIZE is doubled (gfx950 counts wide streaming
    reads at half their size).  None when rocprofv3 is missing, a pass fails or the budget has no room for both passes."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    key = pmc_key(kernel, args.mem, CONFIG_PROFILE.get(args.config, ("", 0))[1])
    if not os.path.exists(exe) or key is None:
        return None
    tmp = tempfile.mkdtemp(prefix="sqn_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", "3", "--warmup", "2", "--vars-per-gpu", str(n_gpu), "--mem", str(args.mem),
             "--upd-freq", str(args.upd_freq), "--bsize", str(args.bsize), "--no-profile", "--no-cpu-baseline", "--no-host-caller",
             "--no-reference-form", "--sustain-seconds", "0", "--no-live-pmc", "--value-runs", "1"]
    for kv in args.opt:
        child += ["--opt", kv]
    got = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            if not budget.fits("live_pmc", budget.costs["live_pmc"] / 2):
                return None
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", out, "--"] + child
            rc, _, _ = run_child(cmd, dict(child_env(budget), TMPDIR="/tmp"), budget.child_timeout(300), cwd="/tmp", quiet=True)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                return None
            vals = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                          if r.get("Counter_Name") == ctr and key in r.get("Kernel_Name", "").replace("sqn::(anonymous namespace)::", ""))
            if not vals:
                return None
            got[ctr] = (vals[-1] if kernel == "sdot" else vals[len(vals) // 2], len(vals))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    b = got["FETCH_SIZE"][0] * 1024 * 2 + got["WRITE_SIZE"][0] * 1024
    return {"bytes": int(round(b)), "read_bytes": int(round(got["FETCH_SIZE"][0] * 2048)), "write_bytes": int(round(got["WRITE_SIZE"][0] * 1024)),
            "dispatches": got["FETCH_SIZE"][1],
            "source": "counted in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over a 5-step "
                      "one-GPU child run of this workload (instantiation %s); FETCH_SIZE x 2 (gfx950 half-count of wide streaming reads)" % key}


def pmc_traffic(kernel, n, m, alg_bytes, config="c3", profiles_dir=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate
    passes; profiles/summarise.py), or (None, why).  The fall-back of every line whose traffic is not counted live.  Selection:
    newest round first, and within a round the file of THIS configuration (`_c3_` for the SQN lines, `_c4_` adaQN, `_c2_` oLBFGS)
    before an untagged one; a file serves only if it holds the exact instantiation this leg launches (pmc_key: ring size and
    k_qdot's MODE) and its bytes, scaled from the n = 1e8 they were counted at, lie within PMC_RATIO_OK of `alg_bytes` -- a file
    that holds some other launch of the same kernel name (another optimiser, a ring still filling) is refused, not quoted."""
    import glob
    import re
    tag, mode = CONFIG_PROFILE.get(config, ("c3", 0))
    key = pmc_key(kernel, m, mode)
    if key is None:
        return None, None
    keys = [key] + ([pmc_key(kernel, m, mode, older=True)] if kernel == "sdot" else [])
    pdir = profiles_dir or os.path.join(ROOT, "profiles")

    def order(f):
        base = os.path.basename(f)
        rnd = re.match(r"(r\d+[a-z]?)_", base)
        cfg = re.match(r"r\d+[a-z]?_(c\d)_", base)
        return (rnd.group(1) if rnd else "", 1 if (cfg and cfg.group(1) == tag) else 0, base)
    refused = []
    for f in sorted(glob.glob(os.path.join(pdir, "r*_pmc_traffic*.json")), key=order, reverse=True):
        cfg = re.match(r"r\d+[a-z]?_(c\d)_", os.path.basename(f))
        if cfg and cfg.group(1) != tag:
            continue                                            # another configuration's run: never (its pass 2 is another kernel)
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        for k, v in d.items():
            if k.replace("sqn::(anonymous namespace)::", "").replace("void ", "").startswith(tuple(keys)) or (kernel in ("bwd", "fwd") and key in k):
                raw = v.get("raw", {})
                # sdot: the same kernel also rebuilds columns of the cached block (a y row as the probe); pass 1 is the largest dispatch
                stat = "max_KiB" if kernel == "sdot" else "median_KiB"
                at_n = PROFILE_N.get(cfg.group(1), 100_000_000) if cfg else 100_000_000
                b = (raw.get("FETCH_SIZE", {}).get(stat, 0.0) * 1024 * 2 + raw.get("WRITE_SIZE", {}).get(stat, 0.0) * 1024) * n / at_n
                ratio = b / alg_bytes if alg_bytes else 0.0
                if PMC_RATIO_OK[0] <= ratio <= PMC_RATIO_OK[1]:
                    return int(round(b)), os.path.relpath(f, ROOT) + " (%s, measured at n=%g)" % (key, at_n)
                refused.append("%s: %s at %.3f of the algorithmic bytes" % (os.path.basename(f), k, ratio))
    return None, ("no committed PMC file holds %s within %.2f-%.2f of the algorithmic bytes" % (key, PMC_RATIO_OK[0], PMC_RATIO_OK[1])
                  + ("; refused: " + "; ".join(refused[:3]) if refused else ""))


# ------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------
def host_facts():
    model = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail, int(lim) - cur) if avail is not None else int(lim) - cur
    except (OSError, ValueError):
        pass
    quota = "?"
    try:
        quota = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    return model, avail, quota


def build_native_oracle():
    """the timed CPU path is built for THIS host (-O3 -march=native -fopenmp); the prebuilt
    liboracle.so (-O2, generic x86-64: it has to run wherever the tests run) is the fallback when no compiler is around."""
    flags = "gcc -O2 -fopenmp (prebuilt, generic x86-64)"
    native = os.path.join(os.environ.get("TMPDIR", "/tmp"), "liboracle_native_%d.so" % os.getpid())
    try:
        subprocess.check_call(["gcc", "-O3", "-march=native", "-std=c99", "-fPIC", "-fopenmp", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                               "-shared", "-o", native, os.path.join(ROOT, "oracle", "stochqn_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120)
        os.environ["ORACLE_SO"] = native
        flags = "gcc -O3 -march=native -fopenmp, built on this host"
    except (OSError, subprocess.SubprocessError):
        native = None
    return flags, native


class HostCopies:
    """The GPU leg's initial state in (pageable) host memory, shared by the host-caller leg and the CPU baseline: S, Y,
    d, x -- and the Hessian batch A for the CPU baseline -- cut to the first `nc` columns when the host is short of memory."""

    def __init__(self, args, gpu, n, m, bs, need_batch):
        import numpy as np
        _, avail, _ = host_facts()
        need = lambda k: (2 * m + (bs if need_batch else 0) + 12) * k * 8
        nc = args.cpu_n if args.cpu_n > 0 else n
        if args.cpu_n <= 0 and avail is not None:
            while nc > 1_000_000 and need(nc) > 0.7 * avail:
                nc //= 2
        self.nc = nc = min(nc, n)

        def rows_to_host(t, rows):                # the first nc columns of every row of a [rows][n] device array
            return t.view(rows, n)[:, :nc].cpu().numpy().reshape(-1) if nc < n else t.cpu().numpy()
        t0 = time.perf_counter()
        self.S, self.Y = rows_to_host(gpu["S"], m), rows_to_host(gpu["Y"], m)
        self.A = rows_to_host(gpu["A"], bs) if need_batch else None
        self.d, self.x0 = gpu["d"][:nc].cpu().numpy(), gpu["x"][:nc].cpu().numpy()
        self.seconds = time.perf_counter() - t0


def pcie_probe(nbytes, device="cuda"):
    """What the link gives a plain pinned transfer of `nbytes` in each direction (torch, hipHostMalloc'ed memory): the
    yardstick for the host-caller leg."""
    import torch
    count = nbytes // 8
    try:
        h = torch.empty(count, dtype=torch.float64).pin_memory()
    except RuntimeError:
        return None
    dv = torch.empty(count, dtype=torch.float64, device=device)
    out = {}
    for name, (dst, src) in (("h2d", (dv, h)), ("d2h", (h, dv))):
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(dv.device)
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize(dv.device)
            ts.append(time.perf_counter() - t0)
        out[name + "_GBps"] = round(nbytes / min(ts) / 1e9, 1)
    return out


def host_caller_leg(args, lib, be, hostc, gpu, n, m, L, step_size, two_loop, budget=None):
    """The step as the reference's own callers take it (R .Call: reference src/Rwrapper.c:98-125; Cython:
    stochqn/pywrapper.pxi:161-207): EVERY array in pageable host memory, structs rebuilt per call, x / grad / *req read and
    written on the host.  S and Y are mirrored in HBM at the first call; per step the gradient goes up and x comes down
    (the direction too with strict_grad = 1).  Timed: seconds inside run_SQN, PCIe included; the caller's own gradient
    (numpy) is not.  Hessian-vector product: d .* v on the host (plumbing variant; not timed either)."""
    import numpy as np
    from stochqn_amd import _abi
    import torch
    nc = hostc.nc
    d, S, Y = hostc.d, hostc.S, hostc.Y
    noise_dev = torch.empty(n, dtype=torch.float64, device=gpu["d"].device)
    res = {"what": "every array in pageable host memory (numpy), run_SQN timed PCIe-inclusive; n=%g, m=%d, L=%d" % (nc, m, L),
           "pcie_probe_pinned_0.8GB" if nc == 100_000_000 else "pcie_probe_pinned": pcie_probe(nc * 8)}
    lib.stochqn_hip_stat.argtypes = [C.c_char_p]
    lib.stochqn_hip_stat.restype = C.c_longlong
    lib.stochqn_hip_pin_host.argtypes = [C.c_void_p, C.c_size_t]
    lib.stochqn_hip_unpin_host.argtypes = [C.c_void_p]
    # (name, strict_grad, the caller pins its arrays, options): the first two are what a binding that owns its arrays gets with
    # the library's defaults (stochqn_amd/free.py pins through stochqn_hip_pin_host); `pageable` is a raw C caller that pins
    # nothing; `vouched` a caller that also promises not to touch x while *req designates it (round 3's default); `checksum`:
    # nobody promises anything, the library compares a checksum of all of x (host threads, under the gradient's upload)
    variants = (("strict_grad_0", 0, True, {}), ("strict_grad_1", 1, True, {}), ("pageable", 0, False, {}),
                ("vouched", 0, True, {"x_upload": 0.0, "x_prefetch": 1.0}), ("checksum", 0, True, {"x_upload": 2.0}))
    per_variant = None
    for vname, strict, pin, opts in variants:
        # the first variant is the library's default and always runs; a further one only while the budget has room for it
        if budget is not None and per_variant is not None and not budget.fits("host_caller:" + vname, 1.3 * per_variant):
            budget.skip("host_caller:" + vname)
            continue
        t_variant = time.time()
        lib.stochqn_hip_release_all()
        lib.stochqn_hip_stats_reset()
        assert lib.stochqn_hip_set_option(b"strict_grad", float(strict)) == 0
        for k, v in opts.items():
            assert lib.stochqn_hip_set_option(k.encode(), v) == 0
        x = hostc.x0.copy()
        grad, hv = np.empty(nc), np.empty(nc)
        x_sum, x_avg_prev = np.zeros(nc), x.copy()
        pinned = [a for a in (x, grad, hv, x_sum, x_avg_prev) if pin and lib.stochqn_hip_pin_host(a.ctypes.data, a.nbytes) == 0]
        rho_h, alpha_h, dummy = np.zeros(m), np.zeros(m), np.zeros(1)
        b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho_h.ctypes.data, alpha_h.ctypes.data,
                          dummy.ctypes.data, dummy.ctypes.data, m, m, 3 % m, L, 0.0, 0.0)
        w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, L, 1, 1, 1, nc)
        req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)
        view = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (nc,))
        clock = {"lib": 0.0, "calls": 0}
        t = [0]

        def one_step():
            target = w.niter + 1
            while w.niter < target or task.value != 101:
                if task.value == 101:
                    noise = gpu["noise"](t[0], noise_dev)[:nc].cpu().numpy()
                    np.multiply(d, view(req.value), out=grad)
                    np.multiply(grad, noise, out=grad)
                elif task.value == 104:
                    np.multiply(d, view(req_vec.value), out=hv)
                t0 = time.perf_counter()
                rc = be.run_SQN(step_size, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec),
                                C.byref(task), C.byref(w), C.byref(info))
                clock["lib"] += time.perf_counter() - t0
                clock["calls"] += 1
                assert rc in (0, 1), rc
            t[0] += 1

        t_first = time.perf_counter()
        one_step()                                  # first call: S and Y (2 m n words) are mirrored, the caller's arrays are pinned
        t_first = time.perf_counter() - t_first
        for _ in range(L - 1):                      # up to niter = 2L: the rest of the first cycle, with its pair
            one_step()
        per = []
        for _ in range(L):                          # niter 2L+1 .. 3L, one whole cycle, every step on the clock
            clock["lib"] = 0.0
            one_step()
            per.append(1e3 * clock["lib"])
        after_pair, pair = per[0], per[-1]          # the step after a request at x_avg uploads x again; the last one builds the pair
        ordinary = sorted(per[1:-1])
        ord_ms = ordinary[len(ordinary) // 2]
        cycle_ms = sum(per)
        x_every_step = lib.stochqn_hip_stat(b"x_uploads") > 4
        up = 8 * nc * (1 + x_every_step)            # grad (+ x when it is uploaded every step)
        down = 8 * nc * (1 + strict)
        dev_ms = two_loop["ms"] + 0.75 if two_loop else None
        res[vname] = {
            "arrays_pinned_by_the_caller": len(pinned), "options": opts,
            "ms_per_step": round(cycle_ms / L, 2), "steps_per_s": round(1e3 * L / cycle_ms * nc / 1e8, 3),
            "ordinary_step_ms": round(ord_ms, 2), "step_after_a_pair_ms": round(after_pair, 2), "pair_step_ms": round(pair, 2),
            "per_step_ms": [round(v, 2) for v in per],
            "bytes_up_per_ordinary_step": up, "bytes_down_per_ordinary_step": down,
            "link_GBps_per_ordinary_step": None if not dev_ms or ord_ms <= dev_ms else round((up + down) / ((ord_ms - dev_ms) * 1e-3) / 1e9, 1),
            "first_call_s": round(t_first, 2),
            "x_uploads": int(lib.stochqn_hip_stat(b"x_uploads")), "x_uploads_skipped": int(lib.stochqn_hip_stat(b"x_uploads_skipped")),
            "host_ranges_pinned": int(lib.stochqn_hip_stat(b"host_ranges_registered")),
            "x_sent_ahead_of_the_guard": int(lib.stochqn_hip_stat(b"x_sent_ahead")), "x_sent_again": int(lib.stochqn_hip_stat(b"x_sent_again")),
            "x_prefetched": int(lib.stochqn_hip_stat(b"x_prefetched"))}
        lib.stochqn_hip_release_all()
        for a in pinned:
            lib.stochqn_hip_unpin_host(a.ctypes.data)
        for k in opts:
            lib.stochqn_hip_set_option(k.encode(), {"x_upload": 1.0, "x_prefetch": 0.0}[k])
        per_variant = max(per_variant or 0.0, time.time() - t_variant)
    lib.stochqn_hip_set_option(b"strict_grad", 0.0)
    lib.stochqn_hip_release_all()
    res["note"] = ("strict_grad_0 / strict_grad_1: the library's defaults (x goes up on every step like the reference's *req = x; the library "
                   "pins nothing by itself) with the caller's five per-call arrays page-locked by their owner through stochqn_hip_pin_host, "
                   "which is what stochqn_amd/free.py does for numpy arrays; strict_grad = 0 is the default (the reference documents `grad` "
                   "as an input that is clobbered, no shipped caller reads it back). pageable: the same with nothing pinned (a raw C caller). "
                   "vouched: the caller also promises not to touch x while *req designates it (x_upload = 0, x_prefetch = 1: round 3's "
                   "default, 36.6 ms there). checksum: x_upload = 2 -- nobody vouches, the library takes a checksum of all of the caller's "
                   "x on host threads of its own while the gradient travels and compares it with the device copy's. ms_per_step = one whole L-cycle on the clock (the pair-building step and the step after it "
                   "included); link_GBps = (bytes up + down) / (ordinary step - the device-resident step's kernels). Round 2 measured "
                   "91.6 ms per step on this path (pageable copies, x and the direction moved on every call in one piece).")
    return res


def cpu_baseline(args, gpu, hostc, n, m, L, bs, step_size, budget=None):
    """The same SQN workload (same inputs, copied from the GPU; Hessian-vector product A'(Av)/bs through
    oracle_fisher_product) on the CPU oracle (kind 'port'), at n itself when host memory allows.  Timed:
    the seconds spent inside the oracle (run_SQN + the Hessian-vector product), not the caller's gradient.
    All usable cores, every thread of the team pinned to a CPU of its own, state first touched by the same team:
    three whole L-cycles, minimum and median.  One thread: one ordinary step and one L-th step (the one that builds a
    pair), composed into a cycle -- a 1-thread step at n = 1e8 takes seconds."""
    import numpy as np
    import torch
    flags, native = build_native_oracle()
    from oracle import oracle
    from stochqn_amd import _abi
    model, avail, quota = host_facts()
    nc = hostc.nc
    threads = oracle.usable_cpus()            # affinity capped by the cgroup quota (16 on the GPU boxes)
    be = oracle.bound()
    olib = oracle.cdll()
    S, Y, A, d = hostc.S, hostc.Y, hostc.A, hostc.d
    t_copy = hostc.seconds
    oracle.set_threads(threads)
    # the team stays where it is put (0.56 vs 1.42 steps/s on the same CPU model with floating threads) -- and
    # WHERE decides as much: 16 CPUs' worth of quota over 256 allowed CPUs, packed on the first 16 (two core complexes) streamed
    # 0.59 steps/s, 2.33 on another box of the same model (round 5 / round 6 default runs).  Both placements are probed with one
    # dot product over two rows of S (1.6 GB at n = 1e8) and the faster one is kept; the probe's rates are in the line.
    bound, placement = 0, None
    if hasattr(olib, "oracle_bind_threads_how") and hasattr(olib, "oracle_probe_dot_GBps"):
        olib.oracle_bind_threads_how.argtypes = [C.c_int]
        olib.oracle_probe_dot_GBps.restype = C.c_double
        olib.oracle_probe_dot_GBps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rates = {}
        for how, name in ((0, "packed"), (1, "spread")):
            olib.oracle_bind_threads_how(how)
            rates[name] = round(olib.oracle_probe_dot_GBps(S.ctypes.data, S.ctypes.data + 8 * nc, nc, 3), 1)
        best = max(rates, key=rates.get)
        bound = olib.oracle_bind_threads_how(1 if best == "spread" else 0)
        placement = {"chosen": best, "probe_dot_GBps": rates,
                     "how": "packed: thread t on the t-th allowed CPU; spread: on allowed CPU t * (allowed / threads)"}
    elif hasattr(olib, "oracle_bind_threads"):
        bound = olib.oracle_bind_threads()

    def team_array(src=None):
        """An n-vector whose pages are first touched by the thread team that will stream it (static schedule, like every
        loop of the oracle), then filled."""
        a = np.empty(nc)
        if hasattr(olib, "oracle_first_touch"):
            olib.oracle_first_touch(a.ctypes.data, nc)
        else:
            a[:] = 0.0
        if src is not None:
            a[:] = src
        return a
    x = team_array(hostc.x0)
    noise_dev = torch.empty(n, dtype=torch.float64, device=gpu["d"].device)
    grad, hv, tb = team_array(), team_array(), np.zeros(bs)
    x_sum, x_avg_prev = team_array(), team_array(x)
    x_sum[:] = 0.0
    rho_h, alpha_h, dummy = np.zeros(m), np.zeros(m), np.zeros(1)
    b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho_h.ctypes.data, alpha_h.ctypes.data,
                      dummy.ctypes.data, dummy.ctypes.data, m, m, 3 % m, L, 0.0, 0.0)
    w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, L, 1, 1, 1, nc)
    req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)

    def view(p):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (nc,))

    clock = {"lib": 0.0}
    tstep = [0]
    room = lambda seconds: budget is None or budget.left() >= seconds      # is there this much of the command's budget left?

    def one_step():
        # one iteration INCLUDING the pair-building calls it triggers (they follow the niter increment), so
        # that a single timed step is self-contained: on until the next request is a plain gradient again
        target = w.niter + 1
        while w.niter < target or task.value != 101:
            if task.value == 101:             # the caller's gradient (not timed): d x (1 + 0.01 (2u - 1)), noise from the GPU generator
                noise = gpu["noise"](tstep[0], noise_dev)[:nc].cpu().numpy()
                np.multiply(d, view(req.value), out=grad)
                np.multiply(grad, noise, out=grad)
            t0 = time.perf_counter()
            if task.value == 104:
                olib.oracle_fisher_product(A.ctypes.data, bs, nc, req_vec.value, tb.ctypes.data, hv.ctypes.data)
            be.run_SQN(step_size, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec),
                       C.byref(task), C.byref(w), C.byref(info))
            clock["lib"] += time.perf_counter() - t0
        tstep[0] += 1

    def timed(k):
        clock["lib"] = 0.0
        for _ in range(k):
            one_step()
        return clock["lib"]

    # niter starts at L.  All cores: three whole cycles (the first also warms the caches and the page tables up);
    # one thread: the first ordinary step and the pair-building step of a fourth cycle, the steps between them on all cores.
    # The budget of the command decides how much of that is done: one cycle always; the second and third, the one-thread steps
    # (a pair-building step on one thread takes as long as a whole cycle on all) and the BLAS cycles only while there is room.
    cycles = [timed(L)]
    while len(cycles) < 3 and room(1.5 * cycles[-1] + 5.0):
        cycles.append(timed(L))
    t1_ord = t1_pair = cycle_1t = None
    # (a pair-building step on ONE thread takes 1.5 all-core cycles: where a cycle takes more than 6 s -- the thread team on a slow
    # share of the memory fabric -- the one-thread figure would cost half a minute and is left out)
    if cycles[-1] <= 6.0 and room(4.0 * cycles[-1] + 5.0):
        oracle.set_threads(1)
        t1_ord = timed(1)
        oracle.set_threads(threads)
        timed(L - 2)
        oracle.set_threads(1)
        t1_pair = timed(1)
        oracle.set_threads(threads)
        cycle_1t = (L - 1) * t1_ord + t1_pair
    scale = nc / 1e8
    c_min, c_med = min(cycles), sorted(cycles)[(len(cycles) - 1) // 2]
    out = {"value": round(L / c_med * scale, 4), "unit": "steps/s at n=1e8" + ("" if nc == 100_000_000 else " (measured at n=%g, scaled by n/1e8)" % nc),
           "cores": threads, "kind": "port",
           "value_allcores": round(L / c_med * scale, 4), "value_allcores_best": round(L / c_min * scale, 4),
           "value_1thread": None if cycle_1t is None else round(L / cycle_1t * scale, 4),
           "n_measured": nc, "cpu_model": model, "nproc": os.cpu_count(), "cgroup_cpu_max": quota,
           "host_mem_available_GB": None if avail is None else round(avail / 1e9, 1),
           "allcores_cycles_s": [round(c, 3) for c in cycles], "allcores_cycle_s": round(c_med, 3),
           "one_thread_ordinary_step_s": None if t1_ord is None else round(t1_ord, 3), "one_thread_pair_step_s": None if t1_pair is None else round(t1_pair, 3),
           "omp": {"threads_pinned": bound, "how": "every thread of the team on a CPU of its own (sched_setaffinity, oracle_bind_threads_how)", "placement": placement,
                   "first_touch": "thread team" if hasattr(olib, "oracle_first_touch") else "one thread"},
           "build": flags,
           "sample": "oracle/stochqn_oracle.c (CPU restatement of the reference; " + flags + "; own BLAS-1 loops, no BLAS library), "
                     "SQN m=%d L=%d bsize=%d at n=%g, the GPU leg's own inputs copied to the host (%.1f s); seconds inside "
                     "run_SQN + the Hessian-vector product A'(Av)/%d (oracle_fisher_product), caller's gradient excluded. "
                     "All usable cores (%d threads, pinned): %d whole L-cycle(s) of %d steps incl. one pair each (%s s; value = median, "
                     "value_allcores_best = minimum; three when the command's budget has room). One thread: %s"
                     % (m, L, bs, nc, t_copy, bs, threads, len(cycles), L, " / ".join("%.2f" % c for c in cycles),
                        "skipped (a cycle on all cores took more than 6 s, or no room in the budget)" if t1_ord is None else
                        "one ordinary step (%.2f s) and one pair-building step (%.2f s), composed into a cycle." % (t1_ord, t1_pair))}
    # ---- the same cycles with the reference's kind of BLAS behind the same restatement (north_star: "src/stochqn.c + BLAS";
    # reference src/stochqn.c:676-706, 946-949 call cblas_ddot / daxpy / dscal / dnrm2 / dgemv of whatever CBLAS they were linked
    # with): the OpenBLAS that scipy / numpy bundle on this box, all usable cores.  Both numbers are reported; `value` stays the
    # own-loops port (kind "port"), whose threads are pinned and whose pages were first touched by the team that streams them.
    out["blas"] = None
    try:
        blas_info = oracle.find_openblas()
        if blas_info is None:
            out["blas"] = {"error": "no OpenBLAS with a CBLAS interface found on this box (scipy.libs / numpy.libs / libopenblas.so)"}
        elif not room(4.0 * cycles[-1] + 5.0):
            out["blas"] = {"skipped": "no room left in the command's budget (a cycle of the port took %.1f s)" % cycles[-1]}
        else:
            if hasattr(olib, "oracle_unbind_threads"):
                olib.oracle_unbind_threads()          # the BLAS runs its own thread pool: the calling thread gets its full mask back first
            got = oracle.use_cblas(blas_info, threads=threads)
            if got is None:
                out["blas"] = {"error": "the CBLAS entry points of %s could not be resolved" % blas_info["path"]}
            else:
                bc = [timed(L)]
                while len(bc) < 2 and room(1.5 * bc[-1] + 5.0):
                    bc.append(timed(L))
                b_min, b_med = min(bc), sorted(bc)[(len(bc) - 1) // 2]
                out["blas"] = {"value": round(L / b_med * scale, 4), "value_best": round(L / b_min * scale, 4), "unit": out["unit"], "kind": "openblas",
                               "cores": got["threads"] or threads, "library": got["library"], "config": got["config"], "ilp64": got["ilp64"],
                               "cycles_s": [round(c, 3) for c in bc],
                               "what": "the same oracle with v_dot / v_axpy / v_scal / v_nrm2 and the two gemv of the Hessian-vector product routed "
                                       "through this library's cblas_ddot / daxpy / dscal / dnrm2 / dgemv (oracle_use_cblas): up to two whole "
                                       "L-cycles (the first also warms the library's thread pool up): `value` = the better one; the library's own thread pool, not pinned"}
    except Exception as e:                                   # a baseline beside the baseline: it never costs the line
        out["blas"] = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        try:
            oracle.use_cblas(None)
        except Exception:
            pass
    if hasattr(olib, "oracle_unbind_threads"):
        olib.oracle_unbind_threads()
    if native:
        try:
            os.unlink(native)                                  # the mapping stays valid; nothing is left behind in TMPDIR
        except OSError:
            pass
    return out


if __name__ == "__main__":
    main()
