"""Real compiled C and C++ programs against include/stochqn.h: built with gcc / g++ at test time and
linked (a) against the CPU oracle through a thin symbol-forwarding shim and (b) against libstochqn.so.
The GPU run must print the same numbers as the oracle run (1e-10 relative, integers exactly)."""
import os
import re
import subprocess

import numpy as np
import pytest

import stochqn_amd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
SRC = os.path.join(ROOT, "tests", "c")

# forwards the public names to the oracle's prefixed ones, so the very same program can run on the CPU
SHIM = r'''
#include "stochqn.h"
#define FWD(ret, name, params, args) ret oracle_##name params; ret name params { return oracle_##name args; }
FWD(workspace_oLBFGS*, initialize_oLBFGS, (const int n, const size_t m, const real_t h, const real_t yr, const real_t mc, const int cn, const int nt), (n, m, h, yr, mc, cn, nt))
FWD(workspace_SQN*, initialize_SQN, (const int n, const size_t m, const size_t L, const real_t mc, const int gd, const real_t yr, const int cn, const int nt), (n, m, L, mc, gd, yr, cn, nt))
FWD(workspace_adaQN*, initialize_adaQN, (const int n, const size_t m, const size_t f, const size_t L, const real_t mi, const real_t mc, const real_t sr, const real_t rw, const int gd, const real_t yr, const int cn, const int nt), (n, m, f, L, mi, mc, sr, rw, gd, yr, cn, nt))
void oracle_dealloc_oLBFGS(workspace_oLBFGS*); void dealloc_oLBFGS(workspace_oLBFGS* w) { oracle_dealloc_oLBFGS(w); }
void oracle_dealloc_SQN(workspace_SQN*); void dealloc_SQN(workspace_SQN* w) { oracle_dealloc_SQN(w); }
void oracle_dealloc_adaQN(workspace_adaQN*); void dealloc_adaQN(workspace_adaQN* w) { oracle_dealloc_adaQN(w); }
FWD(int, run_oLBFGS, (real_t s, real_t x[], real_t g[], real_t** r, task_enum* t, workspace_oLBFGS* w, info_enum* i), (s, x, g, r, t, w, i))
FWD(int, run_SQN, (real_t s, real_t x[], real_t g[], real_t h[], real_t** r, real_t** rv, task_enum* t, workspace_SQN* w, info_enum* i), (s, x, g, h, r, rv, t, w, i))
FWD(int, run_adaQN, (real_t s, real_t x[], real_t f, real_t g[], real_t** r, task_enum* t, workspace_adaQN* w, info_enum* i), (s, x, f, g, r, t, w, i))
'''


def build(tmp_path, source, cxx, against_oracle):
    exe = tmp_path / (os.path.basename(source).split(".")[0] + ("_oracle" if against_oracle else "_hip"))
    cc = ["g++", "-std=c++11"] if cxx else ["gcc", "-std=c99"]
    if against_oracle:
        from oracle import oracle
        so = oracle.build()
        shim = tmp_path / "shim.c"
        shim.write_text(SHIM)
        shim_o = tmp_path / "shim.o"
        subprocess.check_call(["gcc", "-std=c99", "-c", "-I", INC, str(shim), "-o", str(shim_o)])
        link = [str(shim_o), so, "-Wl,-rpath," + os.path.dirname(so), "-lm"]
    else:
        libdir = os.path.dirname(stochqn_amd.LIB_PATH)
        link = ["-L", libdir, "-lstochqn", "-Wl,-rpath," + libdir, "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lm"]
    subprocess.check_call(cc + ["-O1", "-I", INC, source] + link + ["-o", str(exe)])
    return str(exe)


def numbers(line):
    return [float(t) for t in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", line)]


def compare_outputs(got, want):
    gl, wl = got.strip().splitlines(), want.strip().splitlines()
    assert len(gl) == len(wl) and len(wl) > 0
    for a, b in zip(gl, wl):
        na, nb = numbers(a), numbers(b)
        assert len(na) == len(nb), (a, b)
        assert re.sub(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?", "#", a).split("#")[0] == re.sub(r"[-+]?\d+\.\d+(?:[eE][-+]?\d+)?", "#", b).split("#")[0]
        assert np.allclose(na, nb, rtol=1e-10, atol=1e-300), (a, b)


@pytest.mark.parametrize("source,cxx", [("sqn_host_caller.c", False), ("raii_callers.cpp", True)])
def test_programs_build_and_run_on_the_oracle(tmp_path, source, cxx):
    """CPU: the programs compile against the header as C99 / C++11 and behave sanely on the oracle."""
    exe = build(tmp_path, os.path.join(SRC, source), cxx, against_oracle=True)
    out = subprocess.check_output([exe]).decode()
    assert "niter" in out
    assert all(np.isfinite(numbers(l)).all() for l in out.strip().splitlines())


@pytest.mark.gpu
@pytest.mark.parametrize("source,cxx", [("sqn_host_caller.c", False), ("raii_callers.cpp", True)])
def test_compiled_callers_match_oracle_on_gpu(tmp_path, source, cxx):
    want = subprocess.check_output([build(tmp_path, os.path.join(SRC, source), cxx, against_oracle=True)]).decode()
    got = subprocess.check_output([build(tmp_path, os.path.join(SRC, source), cxx, against_oracle=False)]).decode()
    compare_outputs(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("shards", [2, 3])
@pytest.mark.parametrize("source,cxx", [("sqn_host_caller.c", False), ("raii_callers.cpp", True)])
def test_compiled_callers_on_several_device_shards(tmp_path, source, cxx, shards):
    """The same unmodified programs, one host process each, with STOCHQN_HIP_DEVICES in the environment: the
    library shards their workspaces (rehearsed on the one GPU of the box: virtual devices) and they must
    print the oracle's numbers."""
    want = subprocess.check_output([build(tmp_path, os.path.join(SRC, source), cxx, against_oracle=True)]).decode()
    env = dict(os.environ, STOCHQN_HIP_DEVICES=str(shards), STOCHQN_HIP_VIRTUAL_DEVICES="1", STOCHQN_HIP_DEVICES_MIN_N="1",
               STOCHQN_HIP_VERBOSE="1")
    run = subprocess.run([build(tmp_path, os.path.join(SRC, source), cxx, against_oracle=False)], capture_output=True, env=env, timeout=300)
    assert run.returncode == 0, run.stderr.decode()[-2000:]
    assert ("sharded over %d device shards (library-owned" % shards) in run.stderr.decode()
    compare_outputs(run.stdout.decode(), want)
