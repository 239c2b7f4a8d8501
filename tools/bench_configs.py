#!/usr/bin/env python3
"""Secondary measurements (not the headline bench.py line): the other BASELINE.json configurations
and the PCIe-inclusive host-caller rate, all through the free-mode objects of stochqn_amd.

    python tools/bench_configs.py [c2] [c3host] [c4] [rccl1]

Prints one JSON line per configuration.
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import stochqn_amd
from stochqn_amd import oLBFGS_free, SQN_free, adaQN_free

dev = torch.device("cuda", 0)
lib = stochqn_amd.cdll()
lib.stochqn_hip_profile_name.restype = C.c_char_p
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
for _name, _val in [kv.split("=") for kv in os.environ.get("SQN_OPTS", "").split(",") if kv]:     # A/B of kernel-shape knobs
    assert lib.stochqn_hip_set_option(_name.encode(), float(_val)) == 0, _name
    _f32 = stochqn_amd.cdll(use_float=True)                                                        # the float library has options of its own
    _f32.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
    assert _f32.stochqn_hip_set_option(_name.encode(), float(_val)) == 0, _name


def kernels():
    out = {}
    for i in range(lib.stochqn_hip_profile_kernels()):
        cnt, ms = C.c_longlong(), C.c_double()
        lib.stochqn_hip_profile_get(i, C.byref(cnt), C.byref(ms))
        if cnt.value:
            out[lib.stochqn_hip_profile_name(i).decode()] = {"launches": cnt.value, "avg_ms": round(ms.value / cnt.value, 4)}
    return out


class DeviceQuadratic:
    def __init__(self, n, seed=20240611):
        g = torch.Generator(device=dev).manual_seed(seed)
        self.n = n
        self.d = 0.5 + torch.rand(n, dtype=torch.float64, device=dev, generator=g)
        self.dn = [self.d * (1 + 0.01 * (2 * torch.rand(n, dtype=torch.float64, device=dev, generator=g) - 1)) for _ in range(2)]
        self.x0 = 1 + torch.rand(n, dtype=torch.float64, device=dev, generator=g)

    def f(self, x):
        return float(0.5 * torch.sum(self.d * x * x))


def drive(opt, P, x, step, steps, warmup, host=False):
    """Run until `steps` iterations after `warmup`; returns seconds and calls."""
    t = 0
    calls = 0
    last_k = 0
    lib_s = [0.0]

    def advance(k):
        nonlocal t, calls, last_k
        target = opt.niter + k if opt.initialized else k
        while (opt.niter if opt.initialized else 0) < target:
            tc = time.perf_counter()
            r = opt.run_optimizer(x, step)
            lib_s[0] += time.perf_counter() - tc
            calls += 1
            task, req = r["task"], r["requested_on"]
            if task == "calc_grad":
                last_k = t % 2
                t += 1
            if task in ("calc_grad", "calc_grad_same_batch", "calc_grad_big_batch"):
                if host:
                    opt.gradient[:] = (P.dn_h[last_k] * req)
                else:
                    torch.mul(P.dn[last_k], req, out=opt.gradient)
            elif task == "calc_hess_vec":
                rx, rv = req
                if host:
                    opt.hess_vec[:] = P.d_h * rv
                else:
                    torch.mul(P.d, rv, out=opt.hess_vec)
            elif task == "calc_fun_val_batch":
                opt.update_function(0.5 * float(np.sum(P.d_h * req * req)) if host else P.f(req))

    advance(warmup)
    torch.cuda.synchronize()
    # what the CALLER's own kernels cost per step (its gradient / Hessian-vector products run on the GPU too and sit inside the
    # wall-clock step): the same element-wise product timed alone, times the number of such products per step
    caller_ms = None
    if not host:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        scratch = torch.empty_like(x)
        for _ in range(5):
            torch.mul(P.dn[0], x, out=scratch)
        ev[0].record()
        for _ in range(20):
            torch.mul(P.dn[0], x, out=scratch)
        ev[1].record()
        torch.cuda.synchronize()
        caller_ms = ev[0].elapsed_time(ev[1]) / 20
        del scratch
    drive.caller_kernel_ms = caller_ms
    # timed pass WITHOUT the library's event profiler (an event pair costs ~10 us per launch, which is
    # 8 % of a C2 step) ...
    c0 = calls
    lib_s[0] = 0.0
    t0 = time.perf_counter()
    advance(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    drive.lib_seconds = lib_s[0]
    ncalls = calls - c0
    # ... then a shorter untimed pass with it, for the per-kernel breakdown
    lib.stochqn_hip_profile_enable(1)
    lib.stochqn_hip_profile_reset()
    advance(max(1, min(steps, int(os.environ.get("PROFILE_STEPS", "20")))))
    torch.cuda.synchronize()
    lib.stochqn_hip_profile_enable(0)
    return dt, ncalls


PEAK = 8000.0            # GB/s, MI355X HBM3E (MI355X_MICROARCH.md)
PMC_KEYS = {"sdot": "k_rows_dot_all", "sdot2": "k_rows_dot_all", "qdot": "k_qdot", "sadd": "k_sadd", "fisher_t": "k_fisher_t", "fisher_y": "k_fisher_y"}


def words_per_launch(kind, k, fisher_rows):
    """Algorithmic n-words one launch of each kernel has to move, for the optimiser `kind` with k pairs in
    the ring and `fisher_rows` rows in the Fisher product."""
    w = {"sdot": k + 1,                 # pass 1: g and the k rows of S
         "sdot2": k + 2,                # ... plus the new pair's y as a probe (its column of the cached s'y block)
         "qdot": k + 2,                 # pass 2: g and the k rows of Y; writes r0
         "sadd": k + 2,                 # pass 3: r0 and the k rows of S; writes r
         "apply": 5,                    # SQN / adaQN: r, x, x_sum -> x, x_sum;  oLBFGS: r, x -> x, s_slot, grad
         "pair_s": 4, "pair_y_hv": 6, "pair_y_diff": 4,      # g, g_prev, s -> y (+ the three dots)
         "fisher_t": fisher_rows + 1, "fisher_y": fisher_rows + 2,
         "first": 2, "bwd": 4, "mid": 3, "fwd": 4, "fwd_last": 3}
    if kind == "oLBFGS":
        w["sdot"] += 1; w["sdot2"] += 1  # pass 1 also writes g_prev
    if kind == "adaQN":
        w["qdot"] += 4                  # pass 2 also reads G and writes G, H0 and the Fisher row
        w["mid"] += 1
    return w


def pmc_of(kernel):
    """HBM bytes per launch of `kernel` from a summary of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over THIS
    command (profiles/summarise.py; the path comes in through PMC_JSON -- tools/r05_measure.sh sets it for its last pass)."""
    path = os.environ.get("PMC_JSON", "")
    if not path or not os.path.exists(path) or kernel not in PMC_KEYS:
        return None, None
    d = json.load(open(path))
    best = None
    for key, v in d.items():
        if key.startswith(PMC_KEYS[kernel]) and (best is None or v["hbm_bytes_per_launch"] > best["hbm_bytes_per_launch"]):
            best = v                    # several instantiations share a name: the one that moves the most is the pass itself
    return (int(best["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT) + " (FETCH_SIZE x 2 + WRITE_SIZE, median dispatch)") if best else (None, None)


def report(name, workload, n, m, dt, steps, calls, extra=None, kind="SQN", k_pairs=None, fisher_rows=0, elem_bytes=8):
    prof_steps = max(1, min(steps, int(os.environ.get("PROFILE_STEPS", "20"))))      # what drive() ran under the event profiler
    k = kernels()
    kp = m if k_pairs is None else k_pairs
    words = words_per_launch(kind, kp, fisher_rows)
    nb = n * elem_bytes                 # bytes per n-vector (the float build streams 4-byte elements)
    for kn, e in k.items():
        if kn in words:
            e["alg_GB"] = round(words[kn] * nb / 1e9, 3)
            e["alg_GBps"] = round(words[kn] * nb / (e["avg_ms"] * 1e-3) / 1e9, 1)
            e["frac_of_8TBps"] = round(e["alg_GBps"] / PEAK, 4)
    # A host caller's pass 1 and pass 3 run in SLICES of one traversal (kernels.hip: Slice; each slice is a launch of its own
    # as soon as its part of the gradient has arrived over PCIe): the pass moves its words ONCE however many launches it takes.
    # Every step of the three-pass form has exactly one pass 1 (sdot or sdot2), one pass 2 (qdot, never sliced) and one pass 3,
    # so traversals = launches of pass 2.  (Until round 6 the words were multiplied by the launches: 8077.8 GB/s "moved" by a
    # sliced two-loop, above the 8 TB/s peak -- VERDICT r05 weak #5.)
    traversals = {x: e["launches"] for x, e in k.items()}
    if "qdot" in k:
        passes = k["qdot"]["launches"]
        p1 = sum(k[x]["launches"] for x in ("sdot", "sdot2") if x in k)
        for x in ("sdot", "sdot2"):
            if x in k and p1 > passes:
                traversals[x] = k[x]["launches"] * passes / p1
        if "sadd" in k and k["sadd"]["launches"] > passes:
            traversals["sadd"] = passes
        # the update of a host caller runs in `apply_chunks` slices as well, and the launches that send x ahead of the guard
        # (launch_spec_x, one per slice of pass 3: r, x -> x - step r, 3 n words a traversal) are timed under the same name: one
        # update traversal per step is what the step HAS to move; no per-launch rate is quoted for that mixture
        if "apply" in k and k["apply"]["launches"] > passes:
            traversals["apply"] = passes
            for key in ("alg_GB", "alg_GBps", "frac_of_8TBps"):
                k["apply"].pop(key, None)
            k["apply"]["note"] = "sliced update + the launches that send x ahead of the guard: no per-launch rate"
    for kn, e in k.items():
        if kn in words and kn != "apply" and traversals[kn] != e["launches"]:      # per-launch figures of a sliced pass: a launch moves its share of the words
            share = traversals[kn] / e["launches"]
            e["launches_per_traversal"] = round(1 / share, 2)
            e["alg_GB"] = round(words[kn] * nb * share / 1e9, 3)
            e["alg_GBps"] = round(words[kn] * nb * share / (e["avg_ms"] * 1e-3) / 1e9, 1)
            e["frac_of_8TBps"] = round(e["alg_GBps"] / PEAK, 4)
    chain = ("first", "bwd", "mid", "fwd", "fwd_last", "sdot", "sdot2", "qdot", "sadd")
    tl = sum(k[x]["avg_ms"] * k[x]["launches"] for x in chain if x in k) / prof_steps
    form = "three-pass" if "sadd" in k else "sweeps"
    moved_tl = sum(words[x] * traversals[x] for x in chain if x in k) * nb / prof_steps
    moved_step = sum(words[x] * traversals[x] for x, e in k.items() if x in words) * nb / prof_steps
    kern_ms = sum(e["avg_ms"] * e["launches"] for e in k.values()) / prof_steps
    ms_step = 1e3 * dt / steps
    # the caller's own GPU kernels inside the step: one product per gradient request (oLBFGS: two per step), L-th steps one more
    caller_per_step = (calls / steps) * drive.caller_kernel_ms if getattr(drive, "caller_kernel_ms", None) else None
    cands = [x for x in k if x in words]
    dom = max(cands, key=lambda x: k[x]["avg_ms"] * k[x]["launches"]) if cands else None
    roof = None
    if dom:
        alg = int(round(words[dom] * nb * traversals[dom] / k[dom]["launches"]))      # a launch of a sliced pass moves its share
        ach = alg / (k[dom]["avg_ms"] * 1e-3) / 1e9
        traffic, src = pmc_of(dom)
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": PEAK, "unit": "GB/s", "frac": round(ach / PEAK, 4),
                "traffic": traffic, "traffic_source": src, "traffic_over_algorithmic": round(traffic / alg, 4) if traffic else None,
                "alg_bytes_per_launch": alg, "avg_launch_ms": k[dom]["avg_ms"],
                "measured": "HIP events on the library's stream, %d steps of the same workload right after the timed region" % prof_steps}
    out = {"config": name, "workload": workload, "steps_per_s": round(steps / dt, 2), "ms_per_step": round(ms_step, 3),
           "calls": calls, "roofline": roof,
           "two_loop": {"form": form, "ms": round(tl, 4), "bytes_moved": int(moved_tl), "GBps_on_bytes_moved": round(moved_tl / (tl * 1e-3) / 1e9, 1) if tl > 0 else None,
                        "frac_of_8TBps_on_bytes_moved": round(moved_tl / (tl * 1e-3) / 1e9 / PEAK, 4) if tl > 0 else None,
                        "reference_form_bytes": 8 * elem_bytes * kp * n, "pairs_in_ring": kp},
           "step": {"bytes_moved": int(moved_step), "kernel_ms": round(kern_ms, 4), "ms": round(ms_step, 4),
                    "caller_kernels_ms": None if caller_per_step is None else round(caller_per_step, 4),
                    "library_ms": None if caller_per_step is None else round(ms_step - caller_per_step, 4),
                    "launch_and_sync_gap_ms": round(max(0.0, ms_step - kern_ms - (caller_per_step or 0.0)), 4), "gap_frac": round(max(0.0, ms_step - kern_ms - (caller_per_step or 0.0)) / ms_step, 4),
                    "GBps_on_bytes_moved": round(moved_step / (ms_step * 1e-3) / 1e9, 1), "frac_of_8TBps": round(moved_step / (ms_step * 1e-3) / 1e9 / PEAK, 4),
                    "library_GBps_on_bytes_moved": None if caller_per_step is None else round(moved_step / ((ms_step - caller_per_step) * 1e-3) / 1e9, 1),
                    "GBps_inside_kernels": round(moved_step / (kern_ms * 1e-3) / 1e9, 1) if kern_ms > 0 else None},
           "two_loop_ms": round(tl, 4), "kernels": k}
    if extra:
        out.update(extra)
    print(json.dumps(out), flush=True)
    lib.stochqn_hip_release_all()


def c2():
    n, m = 10_000_000, 10
    P = DeviceQuadratic(n)
    x = P.x0.clone()
    opt = oLBFGS_free(mem_size=m, min_curvature=None, check_nan=True, space="device")
    dt, calls = drive(opt, P, x, 0.1, 200, 30)
    report("C2", "oLBFGS n=1e7 m=10 fp64 check_nan=1, device-resident, opts=%s" % os.environ.get("SQN_OPTS", ""), n, m, dt, 200, calls, {"f_end": P.f(x), "mem_used": opt.BFGS_mem.mem_used},
           kind="oLBFGS", k_pairs=opt.BFGS_mem.mem_used)


def c3host():
    """SQN, every array in host memory (profile B): PCIe-inclusive rate."""
    n, m = int(float(os.environ.get("HOST_N", "1e7"))), 20
    P = DeviceQuadratic(n)
    P.d_h = P.d.cpu().numpy()
    P.dn_h = [a.cpu().numpy() for a in P.dn]
    x = P.x0.cpu().numpy().copy()
    opt = SQN_free(mem_size=m, bfgs_upd_freq=2, min_curvature=None, space="host")
    dt, calls = drive(opt, P, x, 0.05, 30, 45, host=True)
    report("C3-host", "SQN n=%g m=20 L=2, ALL arrays in host memory; lib_ms_per_step = time inside run_SQN only (PCIe inclusive)" % n, n, m, dt, 30, calls,
           {"mem_used": opt.BFGS_mem.mem_used, "lib_ms_per_step": round(1e3 * drive.lib_seconds / 30, 3)})


def c4():
    """adaQN, empirical-Fisher pairs + RMSProp H0.  adaQN's H0 is the rescaled GRADIENT (reference
    src/stochqn.c:781,695), so its direction is only sane while x keeps its sign: small steps, L=20,
    and the ring is filled by running the optimiser itself (400 untimed steps).  The timed window is
    iterations 400..440, while the ring holds 19-20 pairs: shortly after that adaQN itself blows up on
    this synthetic quadratic (the CPU oracle does the same at n >= 2e5: the Fisher pairs built from
    nearly parallel gradients are degenerate), which flushes the ring and would understate the cost."""
    n, m, f = 100_000_000, 20, 128
    P = DeviceQuadratic(n)
    for max_incr in (None,) if os.environ.get("C4_QUICK") else (None, 1.01):
        x = P.x0.clone()
        opt = adaQN_free(mem_size=m, fisher_size=f, bfgs_upd_freq=20, max_incr=max_incr, min_curvature=1e-4,
                         scal_reg=1e-4, rmsprop_weight=0.9, space="device")
        dt, calls = drive(opt, P, x, 1e-3, 40, 400)
        report("C4", "adaQN n=1e8 m=20 fisher_size=128 L=20 rmsprop=0.9 max_incr=%s, device-resident, opts=%s" % (max_incr, os.environ.get("SQN_OPTS", "")), n, m, dt, 40, calls,
               {"mem_used": opt.BFGS_mem.mem_used, "fisher_used": opt.Fisher_mem.mem_used, "f_end": P.f(x), "f_start": P.f(P.x0)},
               kind="adaQN", k_pairs=opt.BFGS_mem.mem_used, fisher_rows=f)


def c3f32():
    """SQN n=1e8 m=20 L=10 through the single-precision ABI (libstochqn_f32.so), device-resident."""
    global lib
    lib64 = lib
    lib = stochqn_amd.cdll(use_float=True)
    lib.stochqn_hip_profile_name.restype = C.c_char_p
    n, m = 100_000_000, 20
    P = DeviceQuadratic(n)
    P.d, P.dn, P.x0 = P.d.float(), [a.float() for a in P.dn], P.x0.float()
    x = P.x0.clone()
    opt = SQN_free(mem_size=m, bfgs_upd_freq=1, min_curvature=None, use_float=True, space="device")
    drive(opt, P, x, 0.05, 1, 25)
    opt.BFGS_mem.upd_freq = 10
    opt.bfgs_upd_freq = 10
    opt.niter = 10 * ((opt.niter + 9) // 10)
    dt, calls = drive(opt, P, x, 0.05, 40, 2)
    report("C3-f32", "SQN n=1e8 m=20 L=10 fp32 storage (libstochqn_f32.so), Hv = d*v, device-resident", n, m, dt, 40, calls,
           {"mem_used": opt.BFGS_mem.mem_used, "f_end": float(0.5 * torch.sum(P.d.double() * x.double() ** 2))}, elem_bytes=4)
    lib = lib64


def rccl1():
    """Exercise the all-reduce code path with a 1-rank RCCL communicator."""
    buf = (C.c_ubyte * 128)()
    assert lib.stochqn_hip_comm_unique_id(buf) == 0
    assert lib.stochqn_hip_comm_init(0, 1, bytes(buf)) == 0
    n, m = 100_000_000, 20
    P = DeviceQuadratic(n)
    x = P.x0.clone()
    opt = SQN_free(mem_size=m, bfgs_upd_freq=1, min_curvature=None, space="device")
    drive(opt, P, x, 0.05, 1, 25)
    opt.BFGS_mem.upd_freq = 10
    opt.bfgs_upd_freq = 10
    opt.niter = 10 * ((opt.niter + 9) // 10)
    dt, calls = drive(opt, P, x, 0.05, 30, 2)
    report("RCCL-1rank", "SQN n=1e8 m=20 L=10 with a 1-rank RCCL communicator (k_fin + ncclAllReduce per sweep)", n, m, dt, 30, calls,
           {"nranks": lib.stochqn_hip_comm_nranks(), "mem_used": opt.BFGS_mem.mem_used})
    lib.stochqn_hip_comm_finalize()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c2", "c3host", "c4"]
    for w in which:
        globals()[w]()
