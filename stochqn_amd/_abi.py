"""ctypes view of the free-mode C ABI declared in include/stochqn.h.

The structure layouts mirror reference include/stochqn.h:86-151 (sizes 96/40/48/64/120 bytes on
x86-64 LP64) and the enum values of reference include/stochqn.h:268-291.  `bind()` attaches the
prototypes of the 9 public entry points (reference include/stochqn.h:227-238,381-383) to a loaded
shared library; `prefix` lets the test-suite bind a differently-prefixed checker library with the very
same declarations.
"""
import ctypes as C

real_p = C.POINTER(C.c_double)


class bfgs_mem(C.Structure):
    _fields_ = [
        ("s_mem", C.c_void_p), ("y_mem", C.c_void_p),
        ("buffer_rho", C.c_void_p), ("buffer_alpha", C.c_void_p),
        ("s_bak", C.c_void_p), ("y_bak", C.c_void_p),
        ("mem_size", C.c_size_t), ("mem_used", C.c_size_t), ("mem_st_ix", C.c_size_t),
        ("upd_freq", C.c_size_t), ("y_reg", C.c_double), ("min_curvature", C.c_double),
    ]


class fisher_mem(C.Structure):
    _fields_ = [
        ("F", C.c_void_p), ("buffer_y", C.c_void_p),
        ("mem_size", C.c_size_t), ("mem_used", C.c_size_t), ("mem_st_ix", C.c_size_t),
    ]


class workspace_oLBFGS(C.Structure):
    _fields_ = [
        ("bfgs_memory", C.POINTER(bfgs_mem)), ("grad_prev", C.c_void_p), ("hess_init", C.c_double),
        ("niter", C.c_size_t), ("section", C.c_int), ("nthreads", C.c_int),
        ("check_nan", C.c_int), ("n", C.c_int),
    ]


class workspace_SQN(C.Structure):
    _fields_ = [
        ("bfgs_memory", C.POINTER(bfgs_mem)), ("grad_prev", C.c_void_p),
        ("x_sum", C.c_void_p), ("x_avg_prev", C.c_void_p), ("use_grad_diff", C.c_int),
        ("niter", C.c_size_t), ("section", C.c_int), ("nthreads", C.c_int),
        ("check_nan", C.c_int), ("n", C.c_int),
    ]


class workspace_adaQN(C.Structure):
    _fields_ = [
        ("bfgs_memory", C.POINTER(bfgs_mem)), ("fisher_memory", C.POINTER(fisher_mem)),
        ("H0", C.c_void_p), ("grad_prev", C.c_void_p), ("x_sum", C.c_void_p),
        ("x_avg_prev", C.c_void_p), ("grad_sum_sq", C.c_void_p),
        ("f_prev", C.c_double), ("max_incr", C.c_double), ("scal_reg", C.c_double),
        ("rmsprop_weight", C.c_double), ("use_grad_diff", C.c_int),
        ("niter", C.c_size_t), ("section", C.c_int), ("nthreads", C.c_int),
        ("check_nan", C.c_int), ("n", C.c_int),
    ]


EXPECTED_SIZES = {bfgs_mem: 96, fisher_mem: 40, workspace_oLBFGS: 48, workspace_SQN: 64, workspace_adaQN: 120}

# task_enum / info_enum / iter_status (reference include/stochqn.h:268-291)
TASKS = {101: "calc_grad", 102: "calc_grad_same_batch", 103: "calc_grad_big_batch",
         104: "calc_hess_vec", 105: "calc_fun_val_batch", 100: "invalid_input"}
INFOS = {200: "no_problems_encountered", 201: "func_increased", 202: "curvature_too_small",
         203: "search_direction_was_nan"}
DID_NOT_UPDATE_X, UPDATED_X, RECEIVED_INVALID_INPUT = 0, 1, -1000

PUBLIC_SYMBOLS = [
    "initialize_oLBFGS", "dealloc_oLBFGS", "initialize_SQN", "dealloc_SQN",
    "initialize_adaQN", "dealloc_adaQN", "run_oLBFGS", "run_SQN", "run_adaQN",
    # exported but undeclared in the reference header (reference src/stochqn.c:300,331,342,355)
    "initialize_bfgs_mem", "dealloc_bfgs_mem", "initialize_fisher_mem", "dealloc_fisher_mem",
]


class Bound:
    """The three run_* / initialize_* / dealloc_* families of one shared library."""

    def __init__(self, lib, prefix=""):
        self.lib = lib
        self.prefix = prefix
        g = lambda name: getattr(lib, prefix + name)
        vp, d, i, sz = C.c_void_p, C.c_double, C.c_int, C.c_size_t

        self.initialize_oLBFGS = g("initialize_oLBFGS")
        self.initialize_oLBFGS.restype = C.POINTER(workspace_oLBFGS)
        self.initialize_oLBFGS.argtypes = [i, sz, d, d, d, i, i]
        self.dealloc_oLBFGS = g("dealloc_oLBFGS")
        self.dealloc_oLBFGS.restype = None
        self.dealloc_oLBFGS.argtypes = [C.POINTER(workspace_oLBFGS)]

        self.initialize_SQN = g("initialize_SQN")
        self.initialize_SQN.restype = C.POINTER(workspace_SQN)
        self.initialize_SQN.argtypes = [i, sz, sz, d, i, d, i, i]
        self.dealloc_SQN = g("dealloc_SQN")
        self.dealloc_SQN.restype = None
        self.dealloc_SQN.argtypes = [C.POINTER(workspace_SQN)]

        self.initialize_adaQN = g("initialize_adaQN")
        self.initialize_adaQN.restype = C.POINTER(workspace_adaQN)
        self.initialize_adaQN.argtypes = [i, sz, sz, sz, d, d, d, d, i, d, i, i]
        self.dealloc_adaQN = g("dealloc_adaQN")
        self.dealloc_adaQN.restype = None
        self.dealloc_adaQN.argtypes = [C.POINTER(workspace_adaQN)]

        self.run_oLBFGS = g("run_oLBFGS")
        self.run_oLBFGS.restype = i
        self.run_oLBFGS.argtypes = [d, vp, vp, C.POINTER(vp), C.POINTER(i), C.POINTER(workspace_oLBFGS), C.POINTER(i)]
        self.run_SQN = g("run_SQN")
        self.run_SQN.restype = i
        self.run_SQN.argtypes = [d, vp, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i),
                                 C.POINTER(workspace_SQN), C.POINTER(i)]
        self.run_adaQN = g("run_adaQN")
        self.run_adaQN.restype = i
        self.run_adaQN.argtypes = [d, vp, d, vp, C.POINTER(vp), C.POINTER(i), C.POINTER(workspace_adaQN), C.POINTER(i)]
