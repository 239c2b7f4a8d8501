#!/usr/bin/env python3
"""bench.py -- optimiser steps/s of the stochastic quasi-Newton step path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): SQN with
Hessian-vector correction pairs, n = 1e8 variables per GPU, m = 20 stored pairs, L = 10,
"bsize" = 32 (the Hessian-vector product is A'(A v)/32 over a synthetic dense 32 x n mini-batch), fp64,
check_nan = 1, synthetic noisy-quadratic gradients.  One *step* = everything the library does
from one `niter` to the next: one `run_SQN` call with the two-loop recursion, the guard and the
position update, plus -- every L-th step -- the calls that build the new correction pair.

All inputs live in HBM before the timed region starts (torch tensors handed to the C ABI as
device pointers); the library is driven through run_SQN exactly like the reference's callers do.

N > 1: one process per GPU (torch.distributed.run), the n dimension is sharded, every dot
product inside the library is a local partial + one RCCL all-reduce (stochqn_hip_comm_init).
Weak scaling: n per GPU is fixed, so `value` is normalised to the n = 1e8 problem
(value = steps/s * n_total / 1e8) to stay an aggregate that grows with N.

Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 20240611


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", "--vars-per-gpu", dest="n", type=int, default=100_000_000,
                    help="variables per GPU (--vars-per-gpu under torch.distributed.run, whose parser takes --n for itself)")
    ap.add_argument("--mem", type=int, default=20)
    ap.add_argument("--upd-freq", type=int, default=10)
    ap.add_argument("--bsize", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=10_000_000)
    ap.add_argument("--cpu-steps", type=int, default=0, help="CPU baseline: number of steps (0 = as many as fit in --cpu-seconds)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline: size of the timed sample")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (process group, RCCL communicator) even with one rank")
    ap.add_argument("--rehearse", action="store_true",
                    help="rehearsal of the N > 1 control flow on a box with ONE GPU: every rank uses cuda:0, "
                         "torch.distributed runs over gloo and the library's all-reduce goes through a gloo callback "
                         "(stochqn_hip_comm_init_custom).  Exercises exactly the code the N-GPU run takes, minus RCCL; "
                         "the number it prints is not a measurement.")
    ap.add_argument("--no-reference-form", action="store_true",
                    help="skip the extra untimed steps in the reference's sweep form")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="stochqn_hip_set_option before the run (grid_cap, reverse, nontemporal)")
    return ap.parse_args()


def main():
    args = parse()
    # Everything that libraries print on fd 1 (RCCL's version banner, for one) goes to stderr; the one
    # JSON line is written to the real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    import stochqn_amd
    from stochqn_amd import _abi

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
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

    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
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

            main.keep_alive = REDUCER(gloo_allreduce)
            lib.stochqn_hip_comm_init_custom.argtypes = [C.c_int, C.c_int, REDUCER, C.c_void_p]
            assert lib.stochqn_hip_comm_init_custom(rank, world, main.keep_alive, None) == 0
        else:
            dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                buf = (C.c_ubyte * 128)()
                assert lib.stochqn_hip_comm_unique_id(buf) == 0
                uid = torch.tensor(list(buf), dtype=torch.uint8)
            uid = uid.to(dev)
            dist.broadcast(uid, 0)
            raw = bytes(uid.cpu().tolist())
            assert lib.stochqn_hip_comm_init(rank, world, raw) == 0, "RCCL communicator init failed"

    n, m, L, bs = args.n, args.mem, args.upd_freq, args.bsize
    f64 = torch.float64
    gen = torch.Generator(device=dev).manual_seed(SEED + rank)
    rnd = lambda k: torch.rand(k, dtype=f64, device=dev, generator=gen)

    # ---- synthetic problem: f(x) = 1/2 sum d_i x_i^2, noisy gradients, Hessian batch A ------------
    d = 0.5 + rnd(n)
    x = 1.0 + rnd(n)
    NOISE = 4
    dn = [d * (1.0 + 0.01 * (2.0 * rnd(n) - 1.0)) for _ in range(NOISE)]      # g_t = dn[t % 4] * x
    # Hessian mini-batch of `bs` sample vectors, stored dense [bs][n].  The samples have disjoint
    # supports (a_k,i = sqrt(bs*d_i) for i = k mod bs, else 0) so that A'A/bs = diag(d) exactly:
    # the product streams the full dense batch (2*bs*n words, like any real mini-batch) while the
    # optimiser sees the true Hessian of the quadratic and stays in a sane regime for any n >> bs.
    A = torch.zeros(bs * n, dtype=f64, device=dev)
    sq = torch.sqrt(bs * d)
    for k in range(bs):
        A[k * n + k:(k + 1) * n:bs] = sq[k::bs]
    del sq

    # ---- optimiser state, owned by the caller (profile B), ring already full ------------------------
    S = torch.empty(m * n, dtype=f64, device=dev)
    Y = torch.empty(m * n, dtype=f64, device=dev)
    for k in range(m):
        s = 1e-3 * (rnd(n) - 0.5)
        S[k * n:(k + 1) * n] = s
        Y[k * n:(k + 1) * n] = d * s
    del s
    grad = torch.empty(n, dtype=f64, device=dev)
    hv = torch.empty(n, dtype=f64, device=dev)
    x_sum = torch.zeros(n, dtype=f64, device=dev)
    x_avg_prev = x.clone()
    t_buf = torch.zeros(bs, dtype=f64, device=dev)
    import numpy as np
    rho_h, alpha_h = np.zeros(m), np.zeros(m)
    dummy = torch.zeros(1, dtype=f64, device=dev)

    b = _abi.bfgs_mem(S.data_ptr(), Y.data_ptr(), rho_h.ctypes.data, alpha_h.ctypes.data,
                      dummy.data_ptr(), dummy.data_ptr(), m, m, 3 % m, L, 0.0, 0.0)
    w = _abi.workspace_SQN(C.pointer(b), dummy.data_ptr(), x_sum.data_ptr(), x_avg_prev.data_ptr(), 0,
                           L, 1, 1, 1, n)    # niter = L: the "first average" special case is behind us
    req, req_vec, task, info = C.c_void_p(x.data_ptr()), C.c_void_p(), C.c_int(101), C.c_int(200)
    lib.stochqn_hip_fisher_product.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    step_size = 0.05
    ptr2t = {x.data_ptr(): x, x_sum.data_ptr(): x_sum, x_avg_prev.data_ptr(): x_avg_prev}
    counters = {"calls": 0, "hv": 0, "bad": 0, "rejected": 0}

    def one_step(t):
        """Advance the optimiser by exactly one iteration (niter + 1)."""
        target = w.niter + 1
        while w.niter < target:
            if task.value == 101:                                  # calc_grad at *req
                torch.mul(dn[t % NOISE], ptr2t[req.value], out=grad)
            elif task.value == 104:                                # calc_hess_vec: A'(A v)/bs at x_avg
                counters["hv"] += 1
                rc = lib.stochqn_hip_fisher_product(A.data_ptr(), bs, n, req_vec.value, t_buf.data_ptr(), hv.data_ptr())
                assert rc == 0
            rc = be.run_SQN(step_size, x.data_ptr(), grad.data_ptr(), hv.data_ptr(), C.byref(req), C.byref(req_vec),
                            C.byref(task), C.byref(w), C.byref(info))
            assert rc in (0, 1), rc
            counters["calls"] += 1
            counters["bad"] += info.value == 203
            counters["rejected"] += info.value == 202

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    f0 = float(0.5 * torch.sum(d * x * x))
    for t in range(args.warmup):
        one_step(t)
    if not args.no_profile:
        lib.stochqn_hip_profile_enable(1)
        lib.stochqn_hip_profile_reset()
    barrier()
    t0 = time.perf_counter()
    for t in range(args.warmup, args.warmup + args.steps):
        one_step(t)
    barrier()
    elapsed = time.perf_counter() - t0
    lib.stochqn_hip_profile_enable(0)

    if dist is not None:
        te = torch.tensor([elapsed], dtype=f64, device="cpu" if args.rehearse else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    f1 = float(0.5 * torch.sum(d * x * x))
    if dist is not None:
        tf = torch.tensor([f0, f1], dtype=f64, device="cpu" if args.rehearse else dev)
        dist.all_reduce(tf)
        f0, f1 = tf.tolist()
    assert np.isfinite(f1) and f1 < f0, "optimiser diverged on the synthetic quadratic: %r -> %r" % (f0, f1)

    # ---- per-kernel HIP-event timings -> roofline of the dominant kernel ----------------------------
    kern = {}
    nk = lib.stochqn_hip_profile_kernels()
    lib.stochqn_hip_profile_name.restype = C.c_char_p
    for i in range(nk):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value:
            kern[lib.stochqn_hip_profile_name(i).decode()] = (cnt.value, ms.value)
    PEAK = 8000.0  # GB/s, MI355X HBM3E (MI355X_MICROARCH.md)
    # algorithmic n-words per launch
    words = {"first": 2, "bwd": 4, "mid": 3, "fwd": 4, "fwd_last": 3, "apply": 5,
             "rows_dot": 2 * m + 1, "rows_dot3": 2 * m + 3, "combine": 2 * m + 2}
    what = {"bwd": "fused backward sweep: read y_i, q, s_{i-1}; write q",
            "fwd": "fused forward sweep: read s_i, r, y_{i+1}; write r",
            "combine": "two-pass form, pass B: read g and the %d rows of S and Y; write r" % (2 * m),
            "rows_dot": "two-pass form, pass A: read g and the %d rows of S and Y" % (2 * m)}
    detail = {}
    for name, (cnt, ms) in kern.items():
        avg = ms / cnt
        e = {"launches": cnt, "avg_ms": round(avg, 4)}
        if name in words:
            e["alg_GBps"] = round(words[name] * n * 8 / (avg * 1e-3) / 1e9, 1)
        detail[name] = e
    roof = None
    cands = [k for k in what if k in kern]
    if cands:
        dom = max(cands, key=lambda k: kern[k][1])               # largest share of device time
        cnt, ms = kern[dom]
        alg = words[dom] * n * 8
        ach = alg / (ms / cnt * 1e-3) / 1e9
        traffic, traffic_src = pmc_traffic(dom, n, m)
        roof = {"bound": "hbm", "kernel": "%s (%s)" % (dom, what[dom]),
                "achieved": round(ach, 1), "peak": PEAK, "unit": "GB/s", "frac": round(ach / PEAK, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_launch": alg, "avg_launch_ms": round(ms / cnt, 4)}
    chain = ("first", "bwd", "mid", "fwd", "fwd_last", "rows_dot", "rows_dot3", "coef", "combine")
    two_loop_ms = sum(kern[k][1] for k in chain if k in kern) / max(args.steps, 1)
    two_loop = None
    if two_loop_ms > 0:
        form = "two-pass" if "combine" in kern else "sweeps"
        own = (4 * m + 3) if form == "two-pass" else 8 * m           # n-words this form has to move
        two_loop = {"form": form, "ms": round(two_loop_ms, 3),
                    "bytes_moved": own * n * 8, "GBps_on_bytes_moved": round(own * n * 8 / (two_loop_ms * 1e-3) / 1e9, 1),
                    "frac_of_8TBps_on_bytes_moved": round(own * n * 8 / (two_loop_ms * 1e-3) / 1e9 / PEAK, 4),
                    "reference_form_bytes": 64 * m * n,
                    "effective_GBps_vs_reference_form": round(64.0 * m * n / (two_loop_ms * 1e-3) / 1e9, 1)}

    steps_per_s = args.steps / elapsed
    n_total = n * world
    value = steps_per_s * n_total / 1e8

    # ---- outside the timed region: the same workload in the reference's own dependency structure
    # (2m+1 dependent fused sweeps, 64*m*n algorithmic bytes) for the roofline the north star names --
    ref_form = None
    if "combine" in kern and not args.no_profile and not args.no_reference_form:
        lib.stochqn_hip_set_option(b"twopass", 0.0)
        for t in range(2):
            one_step(args.warmup + args.steps + t)
        lib.stochqn_hip_profile_enable(1)
        lib.stochqn_hip_profile_reset()
        barrier()
        t1 = time.perf_counter()
        extra = 10
        for t in range(extra):
            one_step(args.warmup + args.steps + 2 + t)
        barrier()
        el2 = time.perf_counter() - t1
        lib.stochqn_hip_profile_enable(0)
        lib.stochqn_hip_set_option(b"twopass", 1.0)
        k2 = {}
        for i in range(nk):
            cnt, ms = C.c_longlong(), C.c_double()
            lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
            if cnt.value:
                k2[lib.stochqn_hip_profile_name(i).decode()] = (cnt.value, ms.value)
        if "bwd" in k2:
            cnt, ms = k2["bwd"]
            ach = 4 * n * 8 / (ms / cnt * 1e-3) / 1e9
            tl = sum(k2[k][1] for k in ("first", "bwd", "mid", "fwd", "fwd_last") if k in k2) / extra
            tr, src = pmc_traffic("bwd", n, m)
            ref_form = {"note": "same workload with --opt twopass=0, %d steps after the timed region" % extra,
                        "steps_per_s": round(extra / el2 * n_total / 1e8, 3),
                        "two_loop_ms": round(tl, 3), "two_loop_alg_bytes": 64 * m * n,
                        "two_loop_alg_GBps": round(64.0 * m * n / (tl * 1e-3) / 1e9, 1),
                        "two_loop_frac_of_8TBps": round(64.0 * m * n / (tl * 1e-3) / 1e9 / PEAK, 4),
                        "roofline": {"bound": "hbm", "kernel": "bwd (%s)" % what["bwd"], "achieved": round(ach, 1),
                                     "peak": PEAK, "unit": "GB/s", "frac": round(ach / PEAK, 4), "traffic": tr,
                                     "traffic_source": src, "alg_bytes_per_launch": 4 * n * 8,
                                     "avg_launch_ms": round(ms / cnt, 4)}}

    # ---- the two-loop recursion on its own (SURVEY.md 8d "two-loop micro-benchmark"): the ring as the
    # run left it (m pairs, oldest in row mem_st_ix), H0 = NULL, h0 = 0; 3 warm-up + 20 timed calls of
    # stochqn_hip_two_loop per form, median wall clock of the synchronous call --------------------------
    micro = None
    if not args.no_reference_form:
        lib.stochqn_hip_two_loop.restype = C.c_int
        lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                             C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        g0 = torch.mul(dn[0], x)
        gq = torch.empty_like(g0)
        micro = {"note": "stochqn_hip_two_loop alone: mem_used=%d, oldest pair in row %d, H0=NULL, 3 warm-up + 20 calls, median; "
                         "reference-form bytes = SURVEY 8d's 64*m*n (the sweeps form moves exactly those; the two-pass form moves (4m+3)*n*8)" % (m, b.mem_st_ix)}
        for form, flag in (("two_pass", 1.0), ("sweeps", 0.0)):
            lib.stochqn_hip_set_option(b"twopass", flag)
            ts = []
            for rep in range(23):
                gq.copy_(g0)
                barrier()
                tq = time.perf_counter()
                rc = lib.stochqn_hip_two_loop(gq.data_ptr(), n, None, 0.0, Y.data_ptr(), S.data_ptr(), m, m, b.mem_st_ix,
                                              rho_h.ctypes.data, alpha_h.ctypes.data)
                assert rc == 0
                ts.append(time.perf_counter() - tq)
            med = sorted(ts[3:])[10]
            if dist is not None:
                tm = torch.tensor([med], dtype=f64, device="cpu" if args.rehearse else dev)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                med = float(tm.item())
            moved = ((4 * m + 3) if form == "two_pass" else 8 * m) * n * 8        # bytes this form has to stream
            micro[form] = {"median_ms": round(1e3 * med, 3), "bytes_moved": moved,
                           "GBps_on_bytes_moved": round(moved / med / 1e9, 1),
                           "frac_of_8TBps_on_bytes_moved": round(moved / med / 1e9 / PEAK, 4),
                           "GBps_on_reference_form_bytes": round(64.0 * m * n / med / 1e9, 1)}
        lib.stochqn_hip_set_option(b"twopass", 1.0)
        lib.stochqn_hip_release(C.c_void_p(S.data_ptr()))      # the raw context keyed by S; the optimiser is finished
        del g0, gq

    # ---- CPU baseline: the oracle on the host cores, bounded sample ---------------------------------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:     # a reported baseline of the N = 1 line only
        cpu = cpu_baseline(args, m, L)

    if rank == 0:
        out = {
            "metric": "optimizer steps/sec + achieved HBM GB/s, two-loop at n=10^8 m=20 fp64",
            "value": round(value, 3),
            "unit": "steps/s" if world == 1 else "steps/s normalised to n=1e8 (steps/s * n_total/1e8)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SQN n=%g per GPU (n_total=%g), m=%d, L=%d, Hessian-vector pairs via A'(Av)/%d, "
                                   "check_nan=1, ring full, fp64" % (n, n_total, m, L, bs),
                       "parallelism": ("n sharded over %d GPU(s); one RCCL all-reduce per dot product" % world) if not args.rehearse else
                                      ("REHEARSAL: %d ranks sharing one GPU, all-reduce over gloo -- not a measurement" % world),
                       "calls": counters["calls"], "hess_vec_requests": counters["hv"],
                       "rejected_steps": counters["bad"], "rejected_pairs": counters["rejected"],
                       "options": args.opt,
                       "f_start": f0, "f_end": f1},
            "roofline": roof,
            "two_loop": two_loop,
            "two_loop_micro": micro,
            "reference_form": ref_form,
            "kernels": detail,
            "cpu_baseline": cpu,
        }
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist is not None:
        lib.stochqn_hip_comm_finalize()
        dist.destroy_process_group()


def pmc_traffic(kernel, n, m):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on
    gfx950 + WRITE_SIZE, separate passes; profiles/summarise.py).  The counters were taken at
    n = 1e8, m = 20; the kernels are pure streams, so bytes scale with n."""
    import glob
    key = {"bwd": "BwdOp", "fwd": "FwdOp<true, false, false>", "combine": "k_combine<2", "rows_dot": "k_rows_dot_all<2, 5"}[kernel]
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        d = json.load(open(f))
        for k, v in d.items():
            if key in k and (m == 20 or kernel in ("bwd", "fwd")):
                raw = v["raw"]
                # rows_dot: the same kernel also runs Gram maintenance with the probe among the rows
                # (one stream fewer); pass A is the largest dispatch
                stat = "max_KiB" if kernel == "rows_dot" else "median_KiB"
                b = raw.get("FETCH_SIZE", {}).get(stat, 0.0) * 1024 * 2 + raw.get("WRITE_SIZE", {}).get(stat, 0.0) * 1024
                return int(round(b * n / 1e8)), os.path.relpath(f, ROOT) + " (measured at n=1e8, m=20)"
    return None, None


def cpu_baseline(args, m, L):
    """Same SQN workload on the CPU oracle (kind 'port'), at a bounded size, scaled to n = 1e8."""
    import numpy as np
    from oracle import oracle
    from stochqn_amd import _abi
    nc = min(args.cpu_n, args.n)
    threads = oracle.usable_cpus()            # affinity capped by the cgroup quota (16 on the GPU boxes)
    oracle.set_threads(threads)
    be = oracle.bound()
    rng = np.random.default_rng(SEED)
    d = 0.5 + rng.random(nc)
    x = 1.0 + rng.random(nc)
    dn = d * (1.0 + 0.01 * (2.0 * rng.random(nc) - 1.0))
    S = np.empty(m * nc)
    Y = np.empty(m * nc)
    for k in range(m):
        s = 1e-3 * (rng.random(nc) - 0.5)
        S[k * nc:(k + 1) * nc] = s
        Y[k * nc:(k + 1) * nc] = d * s
    grad, hv = np.empty(nc), np.empty(nc)
    x_sum, x_avg_prev = np.zeros(nc), x.copy()
    rho_h, alpha_h, dummy = np.zeros(m), np.zeros(m), np.zeros(1)
    b = _abi.bfgs_mem(S.ctypes.data, Y.ctypes.data, rho_h.ctypes.data, alpha_h.ctypes.data,
                      dummy.ctypes.data, dummy.ctypes.data, m, m, 3 % m, L, 0.0, 0.0)
    w = _abi.workspace_SQN(C.pointer(b), dummy.ctypes.data, x_sum.ctypes.data, x_avg_prev.ctypes.data, 0, L, 1, 1, 1, nc)
    req, req_vec, task, info = C.c_void_p(x.ctypes.data), C.c_void_p(), C.c_int(101), C.c_int(200)

    def view(p):
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (nc,))

    def one_step():
        target = w.niter + 1
        while w.niter < target:
            if task.value == 101:
                np.multiply(dn, view(req.value), out=grad)
            elif task.value == 104:
                np.multiply(d, view(req_vec.value), out=hv)    # plumbing variant of the Hessian product
            be.run_SQN(0.05, x.ctypes.data, grad.ctypes.data, hv.ctypes.data, C.byref(req), C.byref(req_vec),
                       C.byref(task), C.byref(w), C.byref(info))

    one_step()
    t0 = time.perf_counter()
    done = 0
    while (done < args.cpu_steps) if args.cpu_steps > 0 else (done < 2 * L or time.perf_counter() - t0 < args.cpu_seconds):
        one_step()                                     # at least two full L-cycles, then up to the time bound
        done += 1
    dt = time.perf_counter() - t0
    rate = done / dt * (nc / 1e8)
    return {"value": round(rate, 4), "unit": "steps/s at n=1e8 (scaled)", "cores": threads, "kind": "port",
            "sample": "oracle/liboracle.so (CPU restatement, OpenMP, %d threads), SQN m=%d L=%d at n=%g for %d steps "
                      "(%.2f s); rate scaled by n/1e8 (cost is linear in n); Hessian product = d*v"
                      % (threads, m, L, nc, done, dt)}


if __name__ == "__main__":
    main()
