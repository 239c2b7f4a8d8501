"""The product's host logic under sanitizers, on the CPU (SURVEY.md section 5 "race detection / sanitizers").

`tests/hostsim` compiles stochqn_amd/csrc/{runtime,machines,group}.cpp UNCHANGED with g++ against the real HIP headers and
links them with a malloc-backed stand-in for the 29 HIP runtime entry points they call (fake_hip.cpp), stand-ins for the
kernel launchers (fake_launch.cpp: no numerics of the recursion, the memory footprint of every kernel touched) and a
rendezvous stand-in for RCCL (libfake_rccl_*.so, through STOCHQN_HIP_RCCL_LIB).  No kernel, no oracle, nothing that ships.

Every scenario of host_logic_test.cpp runs twice per build -- streams that execute at once and streams that execute only
when something synchronises (the two ends of what the real runtime may do) -- under -fsanitize=address,undefined and
under -fsanitize=thread; the host-path scenarios a third time with a queue PER STREAM (what a stream other than the one the
caller waited for may legally do):
  model_a_failure (per-stream model only: its collectives are operations on the stream that wait for their peers, like RCCL's):
      one process per GPU over ncclCommInitRank -- three forked ranks, rank-dependent delays, rank 1's 17th all-reduce fails at
      the call: every rank returns -1000 FROM THE SAME CALL (the waiting ones after reducer_patience_s: runtime.cpp wait_stream
      aborts the communicator), every later call fails at once, nobody hangs;
  model_a_no_abort: the same over an RCCL without ncclCommAbort (libfake_rccl_noabort_*.so): the waiting ranks still return -1000
      after the patience, every later call fails at once, release / finalize do not hang -- the contexts are abandoned, not freed;
  caller_heap: a caller whose arrays live in a garbage-collected heap (reference src/Rwrapper.c:106-123, stochqn/pywrapper.pxi:
      161-172) -- x, grad and hess_vec are replaced by new arrays between calls, the old ones unpinned, poisoned and kept out of
      circulation; no copy through host memory may be queued on ANY stream when a call returns, none may ever go through a dead
      array, the library's own hipStreamQuery check ("host_copies_in_flight") must agree;
  registry / reclaim_resume / mirror_cap / host_path / branches / owned_and_raw / threads: the context registry, the LRU
      reclaim -> spill -> resume cycle (incl. a resume that fails half-way and must not lose the state), the cap on mirrors,
      pinning bookkeeping, sliced transfers, x sent ahead of the guard and put right, every branch of the state machines;
  group_rccl / group_virtual: the single-process multi-device mode with 4 worker threads over ncclCommInitAll, and over
      the host-side reducer; one shard over a real communicator (option devices_rccl_single);
  group_alloc_failures: every allocation of that mode failing in turn (NULL / -1000, no hang, no leak);
  glibc_heaps (a third build, WITHOUT a sanitizer -- they replace malloc): which blocks of glibc's own heaps the library agrees to
      page-lock in place -- mmap'ed blocks of either thread yes; blocks of the program break and of a worker thread's arena heap no
      (first block, later block, a page-aligned piece) -- each verdict checked against the flags in glibc's own chunk header; the
      heap's header read through process_vm_readv and through /proc/self/maps; unmapped and ordinary 64 MiB boundaries;
  fault_sweep / fault_sweep_group: EVERY call site of the HIP runtime failing in turn, ~850 runs: a message and -1000 or
      a correct result, never an abort, a wrong x or a leak -- the bound on round 3's unexplained abort.
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIM = os.path.join(ROOT, "tests", "hostsim")
SCENARIOS = ["registry", "reclaim_resume", "mirror_cap", "host_path", "xhash", "branches", "owned_and_raw", "threads",
             "group_rccl", "group_virtual", "group_alloc_failures", "fault_sweep", "fault_sweep_group", "caller_heap"]
# the third stream model (round 5): a queue PER STREAM -- synchronising one stream leaves the others' work queued, hipHostUnregister
# waits for nothing -- for the scenarios in which a copy left behind on a side stream would touch memory the caller has freed
PER_STREAM = ["caller_heap", "host_path", "reclaim_resume", "group_virtual", "group_rccl"]


@pytest.fixture(scope="module")
def built():
    out = subprocess.run(["make", "-C", SIM, "-j2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:]
    return os.path.join(SIM, "build")


CASES = ([(sc, st) for sc in SCENARIOS for st in ("immediate", "lazy")] + [(sc, "per_stream") for sc in PER_STREAM]
         + [("model_a_failure", "per_stream"), ("model_a_no_abort", "per_stream")])


@pytest.mark.parametrize("scenario,streams", CASES)
@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_host_logic(san, scenario, streams, built):
    if san == "tsan" and scenario == "caller_heap" and streams != "per_stream":
        pytest.skip("caller_heap is single-threaded: under TSan (40 s a run) it runs with the stream model it was written for only")
    # model_a_no_abort: an RCCL that does not export ncclCommAbort (round 6, ADVICE r05: the bounded wait must fail the call, not
    # synchronise on a stream that can never drain)
    rccl = "libfake_rccl_noabort_%s.so" if scenario == "model_a_no_abort" else "libfake_rccl_%s.so"
    env = dict(os.environ, STOCHQN_HIP_RCCL_LIB=os.path.join(built, rccl % san),
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1")
    cmd = [os.path.join(built, "host_logic_%s" % san), scenario] + ([streams] if streams != "immediate" else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=600)
    tail = "\n".join(l for l in out.stdout.splitlines() if not l.startswith(("stochqn:", "Error: Could not", "SQN got", "oLBFGS got", "adaQN got")))[-6000:]
    assert out.returncode == 0, tail
    assert "%s (%s streams): ok" % (scenario, streams) in out.stdout, tail
    assert "ERROR: AddressSanitizer" not in out.stdout and "WARNING: ThreadSanitizer" not in out.stdout and "runtime error:" not in out.stdout, tail


@pytest.mark.parametrize("streams", ["immediate", "per_stream"])
def test_pinning_rule_on_glibcs_own_heaps(streams, built):
    """runtime.cpp: pinnable_in_place against glibc's malloc itself (scenario glibc_heaps, the build without a sanitizer)."""
    env = dict(os.environ, STOCHQN_HIP_RCCL_LIB=os.path.join(built, "libfake_rccl_plain.so"))
    cmd = [os.path.join(built, "host_logic_plain"), "glibc_heaps"] + ([streams] if streams != "immediate" else [])
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stdout[-4000:]
    assert "glibc_heaps (%s streams): ok" % streams in out.stdout, out.stdout[-4000:]
    assert "the thread-arena rule was not exercised" not in out.stdout, out.stdout[-2000:]
