"""ctypes view of the free-mode C ABI declared in include/stochqn.h.

The structure layouts mirror reference include/stochqn.h:86-151 (sizes 96/40/48/64/120 bytes on
x86-64 LP64) and the enum values of reference include/stochqn.h:268-291.  `bind()` attaches the
prototypes of the 9 public entry points (reference include/stochqn.h:227-238,381-383) to a loaded
shared library; `prefix` lets the test-suite bind a differently-prefixed checker library with the very
same declarations.
"""
import ctypes as C


def _make_structs(real):
    """The five ABI structs for one precision (`real` = c_double or c_float; reference
    include/stochqn.h:62-76 selects it with USE_DOUBLE / USE_FLOAT)."""

    class bfgs_mem(C.Structure):
        _fields_ = [
            ("s_mem", C.c_void_p), ("y_mem", C.c_void_p),
            ("buffer_rho", C.c_void_p), ("buffer_alpha", C.c_void_p),
            ("s_bak", C.c_void_p), ("y_bak", C.c_void_p),
            ("mem_size", C.c_size_t), ("mem_used", C.c_size_t), ("mem_st_ix", C.c_size_t),
            ("upd_freq", C.c_size_t), ("y_reg", real), ("min_curvature", real),
        ]

    class fisher_mem(C.Structure):
        _fields_ = [
            ("F", C.c_void_p), ("buffer_y", C.c_void_p),
            ("mem_size", C.c_size_t), ("mem_used", C.c_size_t), ("mem_st_ix", C.c_size_t),
        ]

    class workspace_oLBFGS(C.Structure):
        _fields_ = [
            ("bfgs_memory", C.POINTER(bfgs_mem)), ("grad_prev", C.c_void_p), ("hess_init", real),
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
            ("f_prev", real), ("max_incr", real), ("scal_reg", real),
            ("rmsprop_weight", real), ("use_grad_diff", C.c_int),
            ("niter", C.c_size_t), ("section", C.c_int), ("nthreads", C.c_int),
            ("check_nan", C.c_int), ("n", C.c_int),
        ]

    return bfgs_mem, fisher_mem, workspace_oLBFGS, workspace_SQN, workspace_adaQN


bfgs_mem, fisher_mem, workspace_oLBFGS, workspace_SQN, workspace_adaQN = _make_structs(C.c_double)
STRUCTS_F32 = _make_structs(C.c_float)

EXPECTED_SIZES = {bfgs_mem: 96, fisher_mem: 40, workspace_oLBFGS: 48, workspace_SQN: 64, workspace_adaQN: 120}
EXPECTED_SIZES_F32 = dict(zip(STRUCTS_F32, (88, 40, 48, 64, 104)))

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
    """The three run_* / initialize_* / dealloc_* families of one shared library, for one precision."""

    def __init__(self, lib, prefix="", use_float=False):
        self.lib = lib
        self.prefix = prefix
        self.use_float = bool(use_float)
        self.real = C.c_float if use_float else C.c_double
        structs = STRUCTS_F32 if use_float else (bfgs_mem, fisher_mem, workspace_oLBFGS, workspace_SQN, workspace_adaQN)
        self.bfgs_mem, self.fisher_mem, self.workspace_oLBFGS, self.workspace_SQN, self.workspace_adaQN = structs
        workspace_oLBFGS_, workspace_SQN_, workspace_adaQN_ = structs[2:]
        g = lambda name: getattr(lib, prefix + name)
        vp, d, i, sz = C.c_void_p, self.real, C.c_int, C.c_size_t

        self.initialize_oLBFGS = g("initialize_oLBFGS")
        self.initialize_oLBFGS.restype = C.POINTER(workspace_oLBFGS_)
        self.initialize_oLBFGS.argtypes = [i, sz, d, d, d, i, i]
        self.dealloc_oLBFGS = g("dealloc_oLBFGS")
        self.dealloc_oLBFGS.restype = None
        self.dealloc_oLBFGS.argtypes = [C.POINTER(workspace_oLBFGS_)]

        self.initialize_SQN = g("initialize_SQN")
        self.initialize_SQN.restype = C.POINTER(workspace_SQN_)
        self.initialize_SQN.argtypes = [i, sz, sz, d, i, d, i, i]
        self.dealloc_SQN = g("dealloc_SQN")
        self.dealloc_SQN.restype = None
        self.dealloc_SQN.argtypes = [C.POINTER(workspace_SQN_)]

        self.initialize_adaQN = g("initialize_adaQN")
        self.initialize_adaQN.restype = C.POINTER(workspace_adaQN_)
        self.initialize_adaQN.argtypes = [i, sz, sz, sz, d, d, d, d, i, d, i, i]
        self.dealloc_adaQN = g("dealloc_adaQN")
        self.dealloc_adaQN.restype = None
        self.dealloc_adaQN.argtypes = [C.POINTER(workspace_adaQN_)]

        self.run_oLBFGS = g("run_oLBFGS")
        self.run_oLBFGS.restype = i
        self.run_oLBFGS.argtypes = [d, vp, vp, C.POINTER(vp), C.POINTER(i), C.POINTER(workspace_oLBFGS_), C.POINTER(i)]
        self.run_SQN = g("run_SQN")
        self.run_SQN.restype = i
        self.run_SQN.argtypes = [d, vp, vp, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(i),
                                 C.POINTER(workspace_SQN_), C.POINTER(i)]
        self.run_adaQN = g("run_adaQN")
        self.run_adaQN.restype = i
        self.run_adaQN.argtypes = [d, vp, d, vp, C.POINTER(vp), C.POINTER(i), C.POINTER(workspace_adaQN_), C.POINTER(i)]
