/*  stochqn_hip.h -- device-side extras of libstochqn.so (MI355X build).
 *
 *  Nothing in here is needed by a caller of the reference ABI (stochqn.h); these entry points
 *  exist for (a) callers that keep their vectors in HBM, (b) one-process-per-GPU sharding of the
 *  n dimension over RCCL, (c) measurement, (d) checkpointing host-mirrored state.  All symbols
 *  are plain C, pointers and sizes only.
 */
#ifndef STOCHQN_HIP_INCLUDE
#define STOCHQN_HIP_INCLUDE

#include <stddef.h>
#include "stochqn.h"   /* real_t */

#ifdef __cplusplus
extern "C" {
#endif

/* 1 when a HIP device is usable from this process, else 0 (never aborts). */
int stochqn_hip_available(void);

/* ---- isolated two-loop recursion --------------------------------------------------------------
 * Exposes what the reference keeps `static inline`: approx_inv_hess_grad (reference
 * src/stochqn.c:663-708), same argument meaning.  `mem_st_ix` is the row of the OLDEST pair, as
 * take_step passes it (reference src/stochqn.c:820).  grad/H0/y_mem/s_mem may be device or host
 * pointers; buffer_rho/buffer_alpha receive rho_i and alpha_i by logical index.
 * Like the function it stands for, every call recomputes every inner product from the arrays as they
 * are (a pure function of its arguments).  A caller that calls repeatedly on UNCHANGED s_mem / y_mem may
 * set option "raw_reuse_cache" = 1: s'y, y'y and the cached products s_a'y_b of the three-pass form are then kept
 * per (s_mem, row) between calls (stochqn_hip_invalidate(s_mem) after changing rows).  With a caller-supplied
 * diagonal H0 the three-pass form scales q0 by it in its second pass.
 * Without "raw_reuse_cache" an isolated call runs as the chain of sweeps: when every inner
 * product has to be rebuilt for one call that is the cheapest way to evaluate it.
 * The state behind these isolated entries is separate from any optimiser's using the same arrays and
 * is freed by stochqn_hip_release(s_mem) / stochqn_hip_release_all().
 * Returns 0 on success, -1000 on invalid input / no device. */
int stochqn_hip_two_loop(real_t grad[], int n, real_t H0[], real_t h0, real_t y_mem[], real_t s_mem[],
	size_t mem_size, size_t mem_used, size_t mem_st_ix, real_t buffer_rho[], real_t buffer_alpha[]);

/* take_step of reference src/stochqn.c:802-840 (static inline there), same argument meaning: with pairs in
 * memory [H0 <- grad / sqrt(G + scal_reg) after G <- update(G, grad) when grad_sum_sq != NULL] -> two-loop
 * recursion -> guard (check_nan: non-finite or ||dir|| > 1e3 n: memory flushed, *iter_info =
 * search_direction_was_nan, x untouched) -> x -= step_size * dir.  grad is overwritten with the direction;
 * bfgs_memory->mem_used / mem_st_ix are the caller's counters (mem_st_ix as stored in the struct, reference
 * :820), buffer_rho / buffer_alpha are filled.  grad_sum_sq == NULL: oLBFGS / SQN step (h0 as in two_loop, H0
 * ignored); else adaQN's step (H0 built and applied inside the cached forms' own passes when they are on).  Device or host pointers.  Cache policy as for stochqn_hip_two_loop.  Returns 0 or -1000. */
int stochqn_hip_take_step(real_t step_size, int n, real_t x[], real_t grad[], bfgs_mem *bfgs_memory, real_t rmsprop_weight,
	real_t H0[], real_t h0, real_t grad_sum_sq[], real_t scal_reg, int check_nan, info_enum *iter_info);

/* Empirical Fisher product y = F'(F s)/fu of reference src/stochqn.c:946-949 (update_y_fisher
 * without the curvature check).  F is [fu][n] row-major.  Device or host pointers. */
int stochqn_hip_fisher_product(real_t F[], size_t fu, int n, real_t s[], real_t buffer_y[], real_t y[]);

/* ---- contract for callers that pass DEVICE pointers ------------------------------------------------
 * Streams: each context works on its own blocking-flavour HIP stream, ordered after the NULL stream only.
 *   Inputs must be complete on (or ordered before) the NULL stream when a call is entered; outputs are
 *   complete when it returns.  Work on hipStreamNonBlocking / per-thread / side streams must be
 *   synchronised by the caller first.
 * Caches: s'y, y'y and the products s_a'y_b of the pairs in the ring are cached per (s_mem, ring row).  run_*
 *   notices a different optimiser at the same address (counters that do not continue) but NOT rows of
 *   s_mem / y_mem edited or restored in place at the same counters: call stochqn_hip_invalidate(s_mem)
 *   after such an edit.  Option "verify_cache" = 1 re-derives one pair's cached numbers per step and fails
 *   the call (-1000) on a mismatch: a debugging aid, costs a synchronisation and up to (2k+2) n words. */

/* ---- device-context management ------------------------------------------------------------------
 * State that mirrors caller-owned HOST arrays lives in a context keyed by the address of
 * `bfgs_mem.s_mem`.  R and Python never call dealloc_*, so contexts are released explicitly, or
 * all together at process exit. */
void stochqn_hip_invalidate(const void *s_mem);       /* forget cached rho / gamma             */
void stochqn_hip_release(const void *s_mem);          /* free the context of this workspace    */
void stochqn_hip_release_all(void);
/* Copy every mirrored array back into the caller's host arrays (checkpoint / pickle / saveRDS
 * support).  No-op for arrays the caller already keeps in device memory.  Returns 0 or -1000. */
int stochqn_hip_export(const void *s_mem);

/* ---- host arrays pinned by their owner --------------------------------------------------------------
 * A host caller's x / grad / hess_vec (and the x_sum / x_avg_prev it reads requests from) cross PCIe on every step.  Pinned
 * (page-locked) arrays move at link speed, in slices, under the kernels; pageable ones through the runtime's staged copies.
 * The library pins nothing behind the caller's back by default: it cannot know when an array is freed, and a range that is
 * freed and mapped again while still registered faults the GPU at the next copy.  Whoever OWNS an array for a known lifetime
 * pins it here and unpins it before freeing it (a binding's optimiser object for the arrays it allocates; a finaliser on the
 * user's x: stochqn_amd/free.py does both).  Calls nest (a count per address).  Equivalent to the caller's own
 * hipHostRegister(p, bytes, hipHostRegisterPortable) / hipHostUnregister(p) for callers that do not link HIP -- except for the
 * one rule the library adds: a range is pinned IN PLACE only if it has its pages to itself.  hipHostRegister maps the pages that
 * contain [p, p + bytes) into the device's address space at their own addresses (measured on MI355X, ROCm 7.0 and 7.2:
 * tools/pin_probe.hip); a block inside the program-break heap shares its first and last page with its neighbours, and the
 * break moves under it whenever the allocator trims the heap.  Such a range -- and one whose pages overlap a range pinned here
 * already -- is DECLINED (return 1, "host_pins_declined"): it still works, through the runtime's pageable path.  Arrays with a
 * mapping of their own (malloc / numpy / R above the mmap threshold; anything from mmap or posix_memalign(4096, a multiple of
 * 4096)) are pinned.  "register_host" = 1 and the multi-device mode's own registrations follow the same rule.  A block from one
 * of glibc's THREAD arenas is declined as well: memory malloc'ed below the mmap threshold by a thread other than the main one
 * lives in 64 MiB-aligned heaps that share pages among their blocks and are trimmed like the break heap; they are recognised
 * by the header glibc keeps at their start (runtime.cpp: inside_a_thread_arena_heap; other allocators' pools are not known to
 * the library -- a binding on jemalloc / tcmalloc should give the arrays it pins mappings of their own, as stochqn_amd/free.py does).
 * pin: 0 = pinned now (or once more), 1 = not pinned by this call (already page-locked by other means, or declined: the range
 * stays pageable), -1 = refused (not host memory, no device).  unpin: 0, or -1 when the range was not pinned here. */
int stochqn_hip_pin_host(void *p, size_t bytes);
int stochqn_hip_unpin_host(void *p);
/* The one supported way for a caller that does NOT control its allocator (an R .Call shim, a numpy caller, a C program on
 * jemalloc / tcmalloc) to get arrays that the rule above will always pin: `bytes` of zero-filled host memory in a private
 * anonymous mapping of its own (mmap: page-aligned, its pages shared with nothing, unmapped as a whole), page-locked for the
 * device through stochqn_hip_pin_host.  Use it for x, grad, hess_vec and the workspace vectors the caller reads requests from
 * (reference src/Rwrapper.c:106-123 takes them from R vectors, stochqn/pywrapper.pxi:161-172 from numpy arrays: wrap the
 * pointer -- an ALTREP / external-pointer vector in R, np.frombuffer / np.ctypeslib.as_array in Python -- INTEGRATION.md).
 * Returns NULL when the mapping cannot be had; when only the pin is declined or fails (no device) the memory is still returned,
 * pageable, and *pinned (nullable) says 0.  free: unpins and unmaps; `bytes` as given to alloc.  Returns 0, or -1 for a pointer
 * that did not come from stochqn_hip_alloc_host. */
void *stochqn_hip_alloc_host(size_t bytes, int *pinned);
int stochqn_hip_free_host(void *p, size_t bytes);

/* ---- options -------------------------------------------------------------------------------------
 * "nontemporal" (default 1)  stream pair / Fisher rows with non-temporal loads
 * "grid_cap"    (default 0 = one workgroup per compute unit) maximum workgroups per sweep
 * "reverse"     (default 1)  alternate the traversal direction of consecutive sweeps
 * "threepass"   (default 1)  1: the two-loop recursion from cached inner products between the stored pairs, in three
 *                            passes -- S'g, then q0 / r0 / Y'r0 with Y held in registers, then r0 + S'c: S is read twice,
 *                            Y once, (3m+5)n words; 0: always the reference's chain of 2m+1 dependent sweeps (8mn words; also
 *                            used for m > 48 and for ill-conditioned pairs, "kappa_max").  (Round 1's two-pass form -- [S;Y]g,
 *                            a recursion over Gram blocks, one combine pass: (4m+3)n words -- was retired in round 4 together
 *                            with its options: the three-pass form dominated it on bytes, time and accuracy.)
 * "sdot_tile" (default 2)    pass 1 with a single probe over at most 24 rows: adjacent column tiles a workgroup takes per iteration
 *                            (1 or 2: with 2 a lane holds the packs p and p + 256 of every row)
 * "rows_split", "sdot_per_cu", "sdot2_per_cu", "qdot_per_cu", "sadd_per_cu", "pair_per_cu": kernel-shape knobs (rows of pass 1 split over the
 *                            waves of a workgroup; grid sizes in workgroups per compute unit); the defaults are the measured
 *                            optima, DESIGN.md 3.0.  With "phase_ticks" on, passes 2 and 3 (and the second Fisher pass) hold
 *                            ~150 KB of LDS per workgroup, so ONE workgroup is resident per compute unit whatever these say:
 *                            "qdot_per_cu" / "sadd_per_cu" > 1 then only make the grid larger (more, shorter rounds), they do
 *                            not put two workgroups side by side on a compute unit
 * "phase_ticks" (default 8000)  three-pass form, passes 2 and 3: the lanes park their results in LDS and every wave stores what
 *                            it has parked when the chip-wide 100 MHz clock enters a new period of this many ticks (or when
 *                            its 32 slots are full), so that the whole chip writes at the same moment and reads the rest of
 *                            the time: the one store stream among ~20 read streams then costs 0.18 ms instead of 0.48 ms per
 *                            pass at n = 1e8.  Same values at the same addresses, only later.  0 (or < 2): every
 *                            pack is stored as soon as it is final (rounds 2 - 3); values from 2 to 63 are raised to 64
 * "strict_grad" (default 0)  host callers: copy the search direction back into `grad` (n words over PCIe per step).  The
 *                            reference documents `grad` as an INPUT that "will be modified in-place" (reference
 *                            include/stochqn.h:356-358), and none of its callers reads it afterwards (src/Rwrapper.c:98-196,
 *                            stochqn/pywrapper.pxi:161-207, example/c_rosen.c:103-118): off unless asked for.  Device
 *                            callers always find the direction in `grad` (it is computed there).
 * "keep_tail"   (default 0)  three-pass form: fraction of r0 / r -- the part written last, which the next pass reads
 *                            first -- stored with the default cache policy; the rest streams out (sc1 nt).  0.35 was worth 1 % while
 *                            every pack was stored at once (rounds 2 - 3); with "phase_ticks" on, a write-back store is a store out
 *                            of phase and 0 measures 1 - 2 % ahead
 * -- host callers (arrays of R / numpy / malloc crossing the ABI; INTEGRATION.md "host callers") --
 * "register_host" (default 0), "register_min_bytes" (default 4 MiB)  1: the library pins the caller's x / grad / hess_vec /
 *                            x_sum / x_avg_prev in place by itself (hipHostRegister) once an array has been seen at the same
 *                            address in two consecutive calls, and keeps it pinned until the context goes.  Only for callers
 *                            that vouch for the lifetime of their arrays (see stochqn_hip_pin_host above for why it is off);
 *                            arrays below register_min_bytes are never worth pinning
 * "x_upload"    (default 1)  1: a host caller's x is uploaded on every call that uses it (the reference's semantics: *req
 *                            aliases x, an edit between calls moves the iterate); for a large pinned x the upload rides under
 *                            the update, slice by slice.  0: only when the device copy may be out of date -- first call,
 *                            another array, after a request that was not at x, or when any of 256 spread-out probe values
 *                            differs from what the library handed back -- for callers that vouch they do not touch x while
 *                            *req designates it (reference include/stochqn.h:364-366).  2: nobody vouches and the library
 *                            finds out: a checksum of ALL of the caller's x (the buffer as 64-bit words w_i, each mixed with its
 *                            position by a non-linear bijection of 64 bits, h_i = mix(w_i xor key_i): sum h_i and
 *                            sum (2i+1) rot32(h_i) mod 2^64), taken by "hash_threads" host threads while the gradient travels,
 *                            against the same sums of the device copy, taken on the device when the last call ended.  Equal: no
 *                            upload ("x_uploads_skipped").  A change of any one coordinate changes the first sum for certain;
 *                            any other edit (two sign flips, x -> -x, a swap: what sums of the plain words cannot see) passes
 *                            only on a collision of two 64-bit sums of mixed values.  Large x only (>= "host_slice_min" elements)
 * "hash_threads" (default 0) host threads that take the checksum of x ("x_upload" = 2); 0: 8 (half the hardware threads below 16),
 *                            shared among the shards of a multi-device group
 * "upload_slices" (default 8)  three-pass form: pass 1 runs in this many slices, each as soon as its part of `grad` has landed
 *                            (bit-identical to one launch: the lanes' accumulators are carried between the launches); 0 / 1: off
 * "host_slice_min" (default 2^21)  vectors of fewer elements than this cross the link in one piece (no slices, nothing sent ahead)
 * "apply_chunks" (default 8) the update pass runs in this many slices so that the download of x overlaps it (bit-identical)
 * "x_prefetch"  (default 0)  with "x_upload" = 0: a call that returns with *req == x while the device copy of x is out of date (the request before was at
 *                            x_avg) starts the upload of x on a side stream and returns: it runs while the caller evaluates its
 *                            gradient; the next call orders itself behind it and still compares the probe values ("x_prefetched")
 * "spec_x"      (default 1)  three-pass form, n >= ~4e6: pass 3 runs in `apply_chunks` slices and x - step r of each finished
 *                            slice starts its way to the caller's x at once, BEFORE the guard (a sum over all of r) has spoken; the
 *                            guarded update then runs under the transfer, and a step it rejects (NaN / Inf / norm test: rare) is
 *                            put right by sending the untouched x again.  What the caller reads on return is unchanged, bit for bit
 * "max_mirror_bytes" (default 0 = no cap)  cap on the device memory held by mirrors of host arrays: beyond it -- and whenever
 *                            a device allocation fails -- the least recently used context that is not inside a call is moved
 *                            to host memory owned by the library and comes back on its object's next call ("contexts_reclaimed")
 * -- device callers --
 * "null_stream" (default 2)  which stream a call works on.  0: the context's own blocking-flavour stream; 1: the NULL stream;
 *                            2: the NULL stream for device-resident callers of problems up to 2^22 variables (handing work
 *                            between the caller's NULL stream and another stream costs ~15 us each way per call, a third of
 *                            a step at small n), the own stream otherwise.  Inputs must be complete on -- or ordered
 *                            before -- the NULL stream either way.
 * "async_device" (default 0) stream-ordered calls: with check_nan = 0 and min_curvature = 0 (nothing can be rejected) and every
 *                            array in device memory, run_* returns once its kernels are enqueued; the context's stream is a
 *                            blocking one, so work the caller then puts on the NULL stream is ordered after them.  Nothing is
 *                            read back: buffer_rho / buffer_alpha / buffer_y are not filled, the kappa rule is off, a device
 *                            fault surfaces at the caller's next synchronisation.
 * "kappa_max"   (default 1e6)  the three-pass form is used only while every pair in use has
 *                            |s||y| / |s'y| <= this (s almost orthogonal to y: every fp64 evaluation loses
 *                            digits); beyond, the chain of sweeps. inf = off
 * "fisher_split" (default 1) Fisher pass 1 (t = F s) with the rows divided among the waves of a workgroup: s is read once per
 *                            128 rows; 0 = every lane accumulates "fisher_rows" rows (s re-read once per group)
 * "fisher_split_per_cu" (default 0 = as many as fit, at most 4) workgroups per CU of that kernel
 * "fisher_tile" (default 2)  column tiles of 64 packs such a workgroup takes per trip (1 or 2)
 * "fisher_lag" (default 8)   its waves meet at a workgroup barrier every this many column tiles (0 = never)
 * "fisher_rows" (default 16) with "fisher_split" = 0: Fisher rows one workgroup accumulates per pass (8, 16, 32)
 * "verify_cache" (default 0) see "contract for callers that pass DEVICE pointers"
 * "raw_reuse_cache" (default 0)  stochqn_hip_two_loop / _take_step keep cached inner products between calls
 * "devices", "virtual_devices", "devices_min_n": single-process multi-device mode, see below
 * "devices_rccl_single" (default 0)  with "devices" < 2: eligible workspaces (host arrays or library-owned, n >= "devices_min_n")
 *                            run as a group of ONE shard on the current device over a real RCCL communicator (ncclCommInitAll
 *                            of one device, the shard's own thread, ncclAllReduce per reduction): the multi-device mode's code,
 *                            exercised where only one GPU exists
 * "reducer_patience_s" (default 120)  host-side rendezvous reducer (virtual devices, loop-back): seconds a shard waits for the
 *                            others in a reduction before the call fails (-1000); a shard that failed on its own never arrives
 * "fail_alloc_after" (default -1 = off)  fault injection for tests: the (value+1)-th device or
 *                             pinned allocation from now fails once
 * "inject_device_fault" (default 0)      fault injection for tests: the next stream synchronisation
 *                             behaves as if a kernel launch had failed
 * Returns 0, or -1 for an unknown name. Applies to contexts created afterwards and existing ones. */
int stochqn_hip_set_option(const char *name, double value);

/* ---- built-in HIP-event profiler -----------------------------------------------------------------
 * When enabled every kernel launch is bracketed by two events on the library's stream; durations
 * are accumulated per kernel id.  Ids 0..stochqn_hip_profile_kernels()-1. */
/* Callers that cannot reach this API (an R or Python session on top of a binding): with
 * STOCHQN_HIP_PROFILE=1 in the environment the profiler is on from the first call and a per-kernel
 * table (launches, total and average ms) is printed to stderr when the process exits. */
/* With STOCHQN_HIP_ROCTX=1 every run_* call is a named roctx range ("run_SQN section 1") for
 * `rocprofv3 --marker-trace`. */
void stochqn_hip_profile_enable(int on);
void stochqn_hip_profile_reset(void);
int stochqn_hip_profile_kernels(void);
const char* stochqn_hip_profile_name(int kernel_id);
int stochqn_hip_profile_get(int kernel_id, long long *launches, double *total_ms);

/* ---- synthetic inputs for measurement --------------------------------------------
 * Counter-based generator: element i of a vector depends only on (i, seed, stream, t), so rank p of P
 * produces exactly its slice [first_index, first_index + count) of the one-rank problem.
 *   u(i, stream, t) = (splitmix64_finalise(key + i * 0x9E3779B97F4A7C15) >> 11) / 2^53,
 *   key = seed ^ stream * 0x9E3779B97F4A7C15 ^ t * 0xD1B54A32D192ED03     (all arithmetic mod 2^64)
 * uniform    : out_j  = a + b * u(first_index + j)
 * noisy_grad : grad_j = d_j x_j (1 + amp (2 u(first_index + j) - 1)), the stochastic gradient of 1/2 sum d x^2
 * batch_row  : row k of a `bs`-sample Hessian mini-batch with disjoint supports, a_j = sqrt(bs d_j) if
 *              (first_index + j) mod bs == k else 0: A'A/bs has the diagonal d (and couples only j of one residue class)
 * Device pointers only; the kernels are enqueued on the null stream and NOT synchronised. 0 or -1000. */
int stochqn_hip_synth_uniform(real_t *out, size_t count, unsigned long long first_index, unsigned long long seed,
	unsigned long long stream, unsigned long long t, double a, double b);
int stochqn_hip_synth_noisy_grad(real_t *grad, const real_t *d, const real_t *x, size_t count,
	unsigned long long first_index, unsigned long long seed, unsigned long long stream, unsigned long long t, double amp);
int stochqn_hip_synth_batch_row(real_t *row, const real_t *d, size_t count, unsigned long long first_index,
	unsigned k, unsigned bs);

/* ---- sharding n across GPUs (one process per GPU, RCCL over xGMI) ---------------------------------
 * Every rank owns a contiguous slice of all n-vectors and of every row of S, Y, F and passes its
 * LOCAL n to initialize_* / run_*.  After comm_init every dot product inside the library becomes
 * local partial sums + one ncclAllReduce(sum, double) of 1..3 scalars (Fisher: fu scalars) on
 * the library's stream, and all ranks take identical decisions.  RCCL is dlopen()ed on first use.
 *   unique_id : NCCL_UNIQUE_ID_BYTES (128) bytes, produced on one rank and broadcast by the host
 *               program (bench.py uses torch.distributed for that hand-shake only). */
int stochqn_hip_comm_unique_id(void *out128);
int stochqn_hip_comm_init(int rank, int nranks, const void *unique_id128);
int stochqn_hip_comm_nranks(void);
void stochqn_hip_comm_finalize(void);
/* Latency of one reduction as the kernel chain pays it: `reps` in-place sums of `count` (<= 128) doubles through the
 * reducer of the calling thread (the RCCL communicator, a caller-supplied reducer, the loop-back), each followed by a
 * stream synchronisation.  Collective -- every rank (every shard thread under stochqn_hip_devices_foreach) calls it
 * alike.  Median and minimum in microseconds.  0, -1 when no reducer is installed, -1000 on failure. */
int stochqn_hip_comm_allreduce_probe(int count, int reps, double *median_us, double *min_us);

/* ---- event counters ---------------------------------------------------------------------------------
 * Process-wide counts since start (or the last reset).  Names:
 *   "steps_three_pass", "steps_sweeps"  take_step calls with pairs in memory, by the
 *                               form of the two-loop recursion that ran; "steps_plain": no pairs yet (reference :808-812);
 *   "steps_kappa_fallback"      of the sweeps: steps the three-pass form was configured for but a pair with
 *                               |s||y|/|s'y| > "kappa_max" sent to the reference's chain of sweeps;
 *   "allreduces", "allreduce_doubles"   reductions issued by this process (per shard context) and doubles summed;
 *   "contexts_created", "contexts_reclaimed"   device contexts made / exported-and-dropped under memory pressure;
 *   "x_uploads", "x_uploads_skipped", "host_ranges_registered", "x_sent_ahead", "x_sent_again", "x_prefetched"   host-caller path
 *                               (INTEGRATION.md): the last two count steps whose x started its way to the host while pass 3 was still
 *                               running (option "spec_x"), and those of them that the guard then rejected (the old x was sent again).
 *   "host_pins_live" (a gauge: ranges pinned through stochqn_hip_pin_host right now), "host_unpin_failed" (unpins the
 *                               runtime refused: such a range stays page-locked in its books and must not be freed);
 *   "host_pins_declined"        ranges that stochqn_hip_pin_host / "register_host" / the multi-device mode left pageable because
 *                               they lie in the program-break heap or share a page with another pin; "host_pins_foreign": requests
 *                               for ranges the runtime already reports as page-locked (by the caller's own hipHostRegister /
 *                               hipHostMalloc); "host_pin_errors": requests that hipHostRegister itself refused;
 *   "host_copies_in_flight"     the invariant of the host path, asked of the runtime (hipStreamQuery) at the return of every
 *                               run_* call of a host caller: streams of the context that still had work on them.  Always 0 --
 *                               the caller may free x, grad or the requested vector as soon as it has them back (the one
 *                               upload that "x_prefetch" leaves running on purpose is not counted).
 * Returns the count, or -1 for an unknown name. */
long long stochqn_hip_stat(const char *name);
void stochqn_hip_stats_reset(void);

/* ---- sharding n across the GPUs of ONE process (single-process multi-device mode) -------------------------
 * What a caller of the plain reference ABI gets -- an R session through .Call (reference src/Rwrapper.c:98-125),
 * a C program like reference example/c_rosen.c:100-125, the Cython binding: ONE host process, one run_* call
 * per step, and the library drives P devices.  Switched on with stochqn_hip_set_option("devices", P) or
 * STOCHQN_HIP_DEVICES=P in the environment.  It applies to workspaces whose arrays are HOST memory (profile
 * B) and to workspaces made by initialize_* while the option is on (profile A: the struct's array fields
 * are then opaque non-NULL tokens, the arrays exist only as per-device slices, so n is bounded by the SUM of
 * the devices' memories: initialize_SQN(n = 1e9, m = 20) needs 8 GPUs).  Workspaces whose arrays are device
 * pointers, and problems with n < "devices_min_n" (default 2^20), keep running on one device.
 * Per call every shard's host thread uploads its slice of x / grad / hess_vec and brings back its slice of
 * x, of the direction (grad) and of *req / *req_vec; reductions are RCCL all-reduces between the shards'
 * streams (ncclCommInitAll).  "virtual_devices" = 1 (STOCHQN_HIP_VIRTUAL_DEVICES=1) allows more shards than
 * devices with a host-side reducer: a rehearsal of the mode on one GPU, not a way to go faster.
 * stochqn_hip_export / _release / _invalidate take the workspace's s_mem as for one device. */
int stochqn_hip_devices_active(const void *s_mem);    /* shards this workspace runs on (0: single-device path) */
int stochqn_hip_devices_reducer(const void *s_mem);   /* 1 = RCCL, 3 = host-side rendezvous (virtual devices), 0 = none */

/* Device-resident callers of the single-process mode (a process that keeps its vectors on the GPUs, e.g. torch
 * tensors on cuda:0..P-1): no PCIe per step.  For a workspace made by initialize_* while "devices" is on:
 *   layout   which device shard `shard` lives on and which slice [offset, offset + count) of n it owns;
 *   bind     shard `shard` uses the caller's own device vectors x / grad / hess_vec (count elements each, on that
 *            shard's device) in place: nothing is uploaded or downloaded for it any more; run_* is still called once
 *            per step with non-NULL x / grad arguments (they are not dereferenced for bound shards).  All-NULL unbinds;
 *   request  the shard-local device pointers behind *req / *req_vec of the last run_* call (x -> the bound x);
 *   foreach  runs fn(user, shard, device, offset, count) on every shard's own host thread, concurrently, with that
 *            shard's device current and its reducer bound: the place for the caller's per-shard work (gradient
 *            kernels; isolated entry points such as stochqn_hip_fisher_product, whose reductions then span the shards).
 * Return 0, or -1000 when s_mem is not a sharded workspace / arguments are invalid. */
typedef void (*stochqn_hip_shard_fn)(void *user, int shard, int device, size_t offset, size_t count);
int stochqn_hip_devices_layout(const void *s_mem, int shard, int *device, size_t *offset, size_t *count);
int stochqn_hip_devices_bind(const void *s_mem, int shard, real_t *x, real_t *grad, real_t *hess_vec);
int stochqn_hip_devices_request(const void *s_mem, int shard, real_t **req, real_t **req_vec);
int stochqn_hip_devices_foreach(const void *s_mem, stochqn_hip_shard_fn fn, void *user);

/* ---- caller-supplied reducer (MPI, gloo, a fabric RCCL does not speak) ------------------------------
 * Same sharding, but every reduction is handed to `fn`: sum device_buf[0..count) over all ranks, in
 * place, identically on every rank (count <= max(384, fisher_size)).  `device_buf` is device memory; `hip_stream` is
 * the library's stream (a hipStream_t), on which the producing kernel has been enqueued and the
 * consuming kernel will be.  The simplest conforming reducer synchronises that stream, reduces any
 * way it likes (GPU-aware MPI_Allreduce on device_buf, or a copy through the host) and returns with
 * the result in place.  Returns 0 on success.  Mutually exclusive with stochqn_hip_comm_init. */
typedef int (*stochqn_hip_allreduce_fn)(void *user, double *device_buf, int count, void *hip_stream);
int stochqn_hip_comm_init_custom(int rank, int nranks, stochqn_hip_allreduce_fn fn, void *user);

/* ---- loop-back reducer (rehearsal of the sharded path on ONE GPU) -----------------------------------
 * P host threads each drive one shard (its own arrays, its own context) on the same device; the
 * all-reduce becomes a host-side rendezvous of those threads that sums in rank order.  Protocol:
 * loopback_init(P) once; every worker thread calls loopback_join(rank) before its first run_*;
 * all workers must issue the same sequence of calls; loopback_finalize() when they are done.
 * Not for production: it exists so that the sharding logic is testable without a second GPU. */
int stochqn_hip_loopback_init(int nranks);
int stochqn_hip_loopback_join(int rank);
void stochqn_hip_loopback_finalize(void);

#ifdef __cplusplus
}
#endif
#endif /* STOCHQN_HIP_INCLUDE */
