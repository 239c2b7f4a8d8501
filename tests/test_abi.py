"""CPU-side checks of the drop-in boundary: the library loads, exports every symbol that
include/*.h declares, the struct layouts are the reference's, and -- with no GPU -- every entry
point fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

import stochqn_amd
from stochqn_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = src.split("#ifdef __cplusplus\n#include <new>")[0]          # C part only
    src = re.sub(r"\btypedef\b[^;{}]*\([^;{}]*\)\s*;", "", src)          # function-pointer typedefs are not functions
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", src)
    return sorted(set(n for n in names if not n.startswith("defined")))


def test_struct_layouts_match_reference():
    # sizes/offsets of reference include/stochqn.h:86-151 on x86-64 LP64
    for st, size in list(_abi.EXPECTED_SIZES.items()) + list(_abi.EXPECTED_SIZES_F32.items()):
        assert C.sizeof(st) == size, st
    assert _abi.bfgs_mem.mem_size.offset == 48 and _abi.bfgs_mem.min_curvature.offset == 88
    assert _abi.fisher_mem.mem_st_ix.offset == 32
    assert _abi.workspace_oLBFGS.niter.offset == 24 and _abi.workspace_oLBFGS.n.offset == 44
    assert _abi.workspace_SQN.niter.offset == 40 and _abi.workspace_SQN.n.offset == 60
    assert _abi.workspace_adaQN.f_prev.offset == 56 and _abi.workspace_adaQN.niter.offset == 96
    assert _abi.workspace_adaQN.n.offset == 116


@pytest.mark.parametrize("use_float", [False, True])
def test_library_exports_every_declared_symbol(use_float):
    lib = stochqn_amd.cdll(use_float)
    for header in ("stochqn.h", "stochqn_hip.h"):
        names = declared_functions(header)
        assert len(names) >= 9
        for name in names:
            assert hasattr(lib, name), "%s declared in %s but not exported" % (name, header)
    for name in _abi.PUBLIC_SYMBOLS:
        assert hasattr(lib, name)


def test_header_compiles_as_c_and_cxx(tmp_path):
    inc = os.path.join(ROOT, "include")
    c_src = tmp_path / "t.c"
    c_src.write_text('#include "stochqn.h"\n#include "stochqn_hip.h"\n'
                     '_Static_assert(sizeof(bfgs_mem)==96 && sizeof(fisher_mem)==40 && sizeof(workspace_oLBFGS)==48 &&'
                     ' sizeof(workspace_SQN)==64 && sizeof(workspace_adaQN)==120, "layout");\n'
                     '_Static_assert(calc_grad==101 && invalid_input==100 && calc_fun_val_batch==105 &&'
                     ' func_increased==201 && search_direction_was_nan==203 && received_invalid_input==-1000, "enums");\n'
                     'int main(void){return 0;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, "-c", str(c_src), "-o", str(tmp_path / "t.o")])
    cxx = tmp_path / "t.cpp"
    cxx.write_text('#include "stochqn.h"\n'
                   'int use(oLBFGS& a, SQN& b, adaQN& c, double* x, double* g){\n'
                   '  a.run(0.1, x, g); b.run(0.1, x, g, g); c.run(0.1, x, 0.0, g);\n'
                   '  return (int) (a.get_n_iter() + b.get_n_iter() + c.get_n_iter()) + (b.get_req_vec() != 0)\n'
                   '         + (int) a.get_task() + (int) c.get_iter_info() + (a.get_req() != 0) + (int) b.workspace->niter;\n}\n')
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Werror", "-I", inc, "-c", str(cxx), "-o", str(tmp_path / "t2.o")])


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_examples_link_against_the_new_library(tmp_path):
    """Link-level drop-in proof: the reference's own C and C++ callers, unmodified, against this
    repository's header and libstochqn.so (they cannot RUN here: no GPU)."""
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(stochqn_amd.LIB_PATH)
    hip = ["-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, os.path.join(REF, "example", "c_rosen.c"),
                           "-L", libdir, "-lstochqn", *hip, "-o", str(tmp_path / "c_rosen")])
    subprocess.check_call(["g++", "-I", inc, os.path.join(REF, "example", "cpp_rosen.cpp"),
                           "-L", libdir, "-lstochqn", *hip, "-o", str(tmp_path / "cpp_rosen")])
    out = subprocess.check_output(["nm", "-u", str(tmp_path / "c_rosen")]).decode()
    assert "initialize_SQN" in out and "run_SQN" in out and "dealloc_SQN" in out


def test_no_gpu_means_loud_failure_not_fallback(capfd):
    lib = stochqn_amd.cdll()
    if lib.stochqn_hip_available() == 1:
        pytest.skip("a GPU is visible: the loud-failure path cannot be exercised")
    be = stochqn_amd.lib()
    assert not be.initialize_oLBFGS(4, 3, 0.0, 0.0, 0.0, 1, 1)
    assert not be.initialize_SQN(4, 3, 2, 0.0, 0, 0.0, 1, 1)
    assert not be.initialize_adaQN(4, 3, 5, 2, 1.01, 1e-4, 1e-4, 0.9, 0, 0.0, 1, 1)
    err = capfd.readouterr().err
    assert "no usable HIP device" in err
    # caller-owned state (profile B): section 1 needs arithmetic -> must refuse
    opt = stochqn_amd.oLBFGS_free(mem_size=3)
    x = np.array([0.0, 2.0])
    with pytest.raises(ValueError):
        opt.run_optimizer(x, 0.1)
    assert "no usable HIP device" in capfd.readouterr().err
    lib.stochqn_hip_two_loop.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                         C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
    g = np.zeros(4)
    assert lib.stochqn_hip_two_loop(g.ctypes.data, 4, None, 0.0, g.ctypes.data, g.ctypes.data, 1, 1, 0, None, None) == -1000


def test_host_arrays_from_the_library_work_without_a_device_too():
    """stochqn_hip_alloc_host / _free_host (include/stochqn_hip.h): a private anonymous mapping of its own, zero-filled, pinned when a
    device is there -- and plain pageable memory when none is (this container): the pointer is usable either way."""
    lib = stochqn_amd.cdll()
    lib.stochqn_hip_alloc_host.restype = C.c_void_p
    lib.stochqn_hip_alloc_host.argtypes = [C.c_size_t, C.POINTER(C.c_int)]
    lib.stochqn_hip_free_host.argtypes = [C.c_void_p, C.c_size_t]
    pinned = C.c_int(-7)
    n = 100_003
    p = lib.stochqn_hip_alloc_host(8 * n, C.byref(pinned))
    assert p and p % 4096 == 0 and pinned.value in (0, 1)
    if lib.stochqn_hip_available() != 1:
        assert pinned.value == 0
    a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), (n,))
    assert not a.any()
    a[:] = np.arange(n)
    assert a[-1] == n - 1
    del a
    assert lib.stochqn_hip_free_host(p, 8 * n) == 0
    assert lib.stochqn_hip_free_host(p, 8 * n) == -1                     # not (any longer) one of the library's
    assert lib.stochqn_hip_alloc_host(0, None) is None
    q = lib.stochqn_hip_alloc_host(4096, None)                            # `pinned` is optional
    assert q and lib.stochqn_hip_free_host(q, 4096) == 0


def test_product_package_never_touches_the_oracle():
    """The product path must not import, link or dlopen anything under oracle/."""
    pkg = os.path.join(ROOT, "stochqn_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile", ".map")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "liboracle" not in text and "oracle_" not in text, os.path.join(dirpath, fn)
                for line in text.splitlines():
                    if re.match(r"\s*(from|import)\s+oracle", line):
                        raise AssertionError("%s imports the oracle" % fn)
    if shutil.which("ldd"):
        out = subprocess.check_output(["ldd", stochqn_amd.LIB_PATH]).decode()
        assert "oracle" not in out


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_cython_shim_builds_against_the_new_library(tmp_path):
    """Link-level drop-in proof for the Python binding: the reference's own
    stochqn/wrapper_double.pyx + pywrapper.pxi, read where they lie and cythonized OUT of tree,
    compile against this repository's header and link against libstochqn.so; the extension imports
    and exposes the three py_run_* entry points.  (It cannot compute here: no GPU.)"""
    cython = shutil.which("cython")
    if cython is None:
        pytest.skip("cython not installed")
    (tmp_path / "stochqn").mkdir()
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(ROOT, "include", "stochqn.h"), tmp_path / "include" / "stochqn.h")   # OUR header
    c_file = tmp_path / "stochqn" / "wrapper_double.c"
    subprocess.check_call([cython, "-3", os.path.join(REF, "stochqn", "wrapper_double.pyx"), "-o", str(c_file)],
                          cwd=str(tmp_path), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    import sysconfig
    so = tmp_path / "stochqn" / ("wrapper_double" + sysconfig.get_config_var("EXT_SUFFIX"))
    libdir = os.path.dirname(stochqn_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-w", "-I", np.get_include(), "-I", sysconfig.get_paths()["include"],
                           str(c_file), "-L", libdir, "-lstochqn", "-Wl,-rpath," + libdir,
                           "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(so)])
    undefined = subprocess.check_output(["nm", "-D", "-u", str(so)]).decode()
    for sym in ("run_oLBFGS", "run_SQN", "run_adaQN"):
        assert sym in undefined                   # resolved by libstochqn.so at load time
    code = ("import sys; sys.path.insert(0, %r); import wrapper_double as w; "
            "assert all(hasattr(w, f) for f in ('py_run_oLBFGS', 'py_run_SQN', 'py_run_adaQN')); print('ok')"
            % str(tmp_path / "stochqn"))
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.check_output([os.sys.executable, "-c", code], env=env).decode()
    assert out.strip() == "ok"


# The few R API names the reference's .Call shim uses, declared (not implemented) so that the shim
# can be COMPILED where R is not installed.  This is a compile/link check of a caller, not a build
# of the reference's algorithm: src/stochqn.c is not part of it.
_R_API_DECLS = {
    "R.h": "#pragma once\n#include <stddef.h>\n#ifndef TRUE\n#define TRUE 1\n#define FALSE 0\n#endif\n",
    "Rinternals.h": ("#pragma once\ntypedef struct SEXPREC *SEXP;\nextern SEXP R_NilValue;\n"
                     "double *REAL(SEXP x);\nint *INTEGER(SEXP x);\n"),
    "R_ext/Rdynload.h": ("#pragma once\ntypedef void *(*DL_FUNC)();\ntypedef struct _DllInfo DllInfo;\n"
                         "typedef struct { const char *name; DL_FUNC fun; int numArgs; } R_CallMethodDef;\n"
                         "int R_registerRoutines(DllInfo *info, const void *c, const R_CallMethodDef *call, const void *f, const void *e);\n"
                         "int R_useDynamicSymbols(DllInfo *info, int value);\n"
                         "void R_RegisterCCallable(const char *package, const char *name, DL_FUNC fptr);\n"),
}


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the build container")
def test_reference_r_shim_compiles_and_links_against_the_new_library(tmp_path):
    """Link-level drop-in proof for the R binding (INTEGRATION.md "R package"): the
    reference's own src/Rwrapper.c, read where it lies, compiles against this repository's header
    with the package's flags (-D_FOR_R -DUSE_DOUBLE) and links against libstochqn.so with
    stochqn.o dropped from the objects; every run_* / initialize_* / dealloc_* it calls or
    re-exports is left undefined for libstochqn.so, and the only other undefined names are R's."""
    for name, text in _R_API_DECLS.items():
        path = tmp_path / "rapi" / name
        path.parent.mkdir(parents=True, exist_ok=True)
        path.write_text(text)
    obj, so = tmp_path / "Rwrapper.o", tmp_path / "stochQN.so"
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-fPIC", "-Wall", "-Werror=implicit-function-declaration",
                           "-Werror=incompatible-pointer-types", "-D_FOR_R", "-DUSE_DOUBLE",
                           "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path / "rapi"),
                           "-c", os.path.join(REF, "src", "Rwrapper.c"), "-o", str(obj)])
    libdir = os.path.dirname(stochqn_amd.LIB_PATH)
    subprocess.check_call(["gcc", "-shared", str(obj), "-L", libdir, "-lstochqn", "-Wl,-rpath," + libdir,
                           "-L/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(so)])
    undefined = {line.split()[-1].split("@")[0] for line in
                 subprocess.check_output(["nm", "-D", "-u", str(so)]).decode().splitlines() if line.strip()}
    ours = {p + k for p in ("initialize_", "dealloc_", "run_") for k in ("oLBFGS", "SQN", "adaQN")}
    assert ours <= undefined
    exported = subprocess.check_output(["nm", "-D", "--defined-only", stochqn_amd.LIB_PATH]).decode()
    assert all((" T " + s) in exported for s in ours)
    r_names = {"REAL", "INTEGER", "R_NilValue", "R_registerRoutines", "R_useDynamicSymbols", "R_RegisterCCallable"}
    leftovers = {u for u in undefined - ours - r_names if not u.startswith(("__", "_ITM", "_Jv")) and u not in ("memcpy",)}
    assert not leftovers, leftovers
    needed = subprocess.check_output(["readelf", "-d", str(so)]).decode()
    assert "libstochqn.so" in needed
    defined = subprocess.check_output(["nm", "-D", "--defined-only", str(so)]).decode()
    for entry in ("r_run_oLBFGS", "r_run_SQN", "r_run_adaQN", "R_init_stochQN"):
        assert entry in defined


_ONE_RUNTIME = r"""
import sys
sys.path.insert(0, %r)
import stochqn_amd
lib = stochqn_amd.cdll()                     # the library BEFORE torch
import torch
maps = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})
assert len(maps) == 1, maps
print('one-runtime', torch.cuda.is_available() == bool(lib.stochqn_hip_available()))
"""


def test_library_and_torch_share_one_hip_runtime():
    """torch bundles its own libamdhip64.so; loaded after libstochqn.so it used to become a second HIP
    runtime in the process (torch.cuda then sees no device).  stochqn_amd.cdll() loads the bundled
    copy first, so there is one runtime whichever import comes first, and both agree on the GPU."""
    pytest.importorskip("torch")
    out = subprocess.check_output([os.sys.executable, "-c", _ONE_RUNTIME % ROOT], stderr=subprocess.STDOUT).decode()
    assert "one-runtime True" in out, out


def test_every_option_and_counter_the_header_names_exists():
    """include/stochqn_hip.h documents options (stochqn_hip_set_option) and counters (stochqn_hip_stat) by name: every quoted
    name in those two comment blocks must be one the library knows -- and every name the library knows must be documented.
    Runs in a child process (set_option changes process-wide state)."""
    code = r'''
import ctypes as C, re, sys
sys.path.insert(0, %r)
import stochqn_amd
lib = stochqn_amd.cdll()
lib.stochqn_hip_set_option.argtypes = [C.c_char_p, C.c_double]
lib.stochqn_hip_stat.argtypes = [C.c_char_p]
lib.stochqn_hip_stat.restype = C.c_longlong
text = open(%r).read()
opts = text[text.index("/* ---- options"):text.index("int stochqn_hip_set_option(")]
stats = text[text.index("/* ---- event counters"):text.index("long long stochqn_hip_stat(")]
is_opt = lambda n: lib.stochqn_hip_set_option(n.encode(), 1.0) == 0
is_stat = lambda n: lib.stochqn_hip_stat(n.encode()) >= 0
quoted = lambda block: sorted(set(re.findall(r'"([a-z][a-z_0-9]+)"', block)))
bad = [n for n in quoted(opts) if not (is_opt(n) or is_stat(n))]
bad += [n for n in quoted(stats) if not (is_stat(n) or is_opt(n))]
assert not bad, "named in the header, unknown to the library: %%s" %% bad
assert lib.stochqn_hip_set_option(b"no_such_option", 1.0) == -1 and lib.stochqn_hip_stat(b"no_such_counter") == -1
# the other way round: what the sources accept is in the header
src = open(%r).read()
known_opts = set(re.findall(r'std::strcmp\(name, "([a-z_0-9]+)"\)', src))
known_stats = set(re.findall(r'"([a-z_0-9]+)"', src[src.index("kStatNames[ST_COUNT] = {"):src.index("};", src.index("kStatNames[ST_COUNT] = {"))]))
assert len(known_opts) > 30 and len(known_stats) > 10
missing = sorted(n for n in known_opts if '"%%s"' %% n not in opts and '"%%s"' %% n not in stats) + sorted(n for n in known_stats if '"%%s"' %% n not in stats)
assert not missing, "known to the library, missing from the header: %%s" %% missing
print("ok", len(known_opts), len(known_stats))
''' % (ROOT, os.path.join(ROOT, "include", "stochqn_hip.h"), os.path.join(ROOT, "stochqn_amd", "csrc", "runtime.cpp"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("ok")
