// runtime.hpp -- device contexts behind the C ABI: where each caller array lives, the mirrors of
// host arrays in HBM, scratch for the sweep chain, the RCCL hook and the global options.
#pragma once
#include "sqn_device.hpp"
#include "stochqn.h"

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <mutex>
#include <vector>

namespace sqn {

enum Kind { KIND_OLBFGS = 1, KIND_SQN = 2, KIND_ADAQN = 3, KIND_RAW = 4 };

// One caller array as the kernels see it.  `caller` is the pointer found in the caller's struct;
// when that is device memory `dev == caller`, otherwise `dev` is a mirror owned by the context.
struct View {
	const void* caller = nullptr;
	real* dev = nullptr;
	size_t count = 0;
	bool mirror = false;
};

struct Options {
	bool nontemporal = true;
	int grid_cap = 0;            // 0 = one workgroup per compute unit (measured optimum)
	// pass 1 without a second probe: every lane keeps all rows (the default in both builds since round 6), or the waves of a
	// workgroup split the rows (k_rows_dot: 1 KB of a row per workgroup and trip).  Rounds 1 - 5 used the row-split form in the float
	// build (round 1's all-rows kernel: 5.9 against 2.5 ms); with the kernel as it is now the all-rows form is 9 % ahead there too
	// (1.31 against 1.43 ms at n = 1e8, k = 20, interleaved: profiles/r06_f32_ab_rows_split.log)
	bool rows_split = false;
	bool reverse = true;
	// 1: the three-pass form (S twice, Y once: (3k+5) n words) when the ring has <= kPairsMax3 pairs and every pair in use is
	// tame ("kappa_max"); 0: always the reference's chain of dependent sweeps (8k n words)
	bool threepass = true;
	int fisher_rows = 16;        // Fisher rows per workgroup pass (8, 16, 32) of the all-rows-per-lane kernel (fisher_split = 0)
	// Fisher pass 1 (t = F s) with the rows divided among the 8 waves of a workgroup, which share the columns: s is fetched from
	// HBM once per 128 rows instead of once per 16 (kernels.hip: k_fisher_t_split; PMC 103.2 GB against 108.8 GB at fu = 128)
	bool fisher_split = true;
	int fisher_split_per_cu = 0;
	int fisher_tile = 2;         // column tiles of 64 packs per trip of such a workgroup (1 or 2): 2 KB of every row back to back, 15.67 against 16.2 ms at fu = 128
	int fisher_lag = 8;          // the waves of such a workgroup meet at a barrier every this many column tiles (0 = never): how long a line of s must survive in L2
	int qdot_per_cu = 0, sadd_per_cu = 0, sdot2_per_cu = 0, sdot_per_cu = 0;
	int sdot_tile = 2;           // pass 1 with one probe over <= 24 rows takes two adjacent column tiles per iteration (8 KB of a row per workgroup): 3 - 4 % faster
	int pair_per_cu = 0;         // workgroups per CU of the pair kernels (0 = 1)
	// three-pass form: fraction of r0 / r stored with the default (cacheable) policy -- the part the next pass, which walks
	// the other way, reads first; the rest leaves with sc1 nt.  Rounds 2 - 3 (every pack stored at once): 0.25-0.5 measured 1 % ahead
	// of 0 and of 1 (profiles/r03_ab_fold_tail.jsonl).  With the clock-phased stores a write-back store is a store out of phase:
	// 0 measures 1 - 2 % ahead of 0.35 (profiles/r04b_keep_tail_ab.log), 1 loses the phases altogether
	double keep_tail = 0;
	bool x_prefetch = false;      // with x_upload = 0: when a call hands *req == x back and the device copy of x is stale, x starts its way up on a side stream while the caller evaluates its gradient (reads the caller's x after the call has returned: opt-in)
	bool spec_x = true;           // host callers, three-pass form: slices of x start their way down while pass 3 is still running
	// pass 2 / pass 3: results parked in LDS, every wave stores them when the chip-wide 100 MHz clock enters a new period of this
	// many ticks (kernels.hip: Parked; 4000 - 10000 measured alike, 2000 and 16000 1 - 3 % behind); 0 = each pack stored at once
	int phase_ticks = 8000;
	double kappa_max = 1e6;      // three-pass form only while every pair in use has |s||y|/|s'y| <= this (else: sweeps)
	// host callers: copy the search direction back into `grad` (n words over PCIe).  The reference documents `grad` as an
	// input that "will be modified in-place" (include/stochqn.h:356-358), not as an output, and no shipped caller reads it
	// afterwards (Rwrapper.c, pywrapper.pxi, c_rosen.c): off by default, on for callers that do.
	bool strict_grad = false;
	// single-process multi-device mode (group.cpp): shard n over `devices` GPUs of this process
	int devices = 0;             // 0 / 1 = off; also STOCHQN_HIP_DEVICES in the environment
	bool virtual_devices = false;   // shards may share a physical device (host-side reducer): rehearsal on one GPU
	// with "devices" < 2: run eligible workspaces as a group of ONE shard over a real RCCL communicator (ncclCommInitAll(1)): every
	// line of the multi-device mode that a one-GPU box can execute, executed
	bool devices_rccl_single = false;
	long devices_min_n = 1 << 20;   // problems smaller than this stay on one device (SURVEY.md 8e "no-shard fallback")
	double reducer_patience_s = 120;   // host-side rendezvous reducer: how long a shard waits for the others before the reduction fails
	// host callers (R / numpy / malloc arrays crossing the ABI): PCIe is what a step costs.  Arrays that are PINNED -- by the
	// caller (hipHostMalloc / hipHostRegister) or through stochqn_hip_pin_host by a binding that owns them -- move at link speed
	// and in slices under the kernels; anything else goes through the runtime's staged copies.
	// 1: the library pins what it is handed by itself (an array seen at the same address in two consecutive calls), and keeps it
	// pinned until the context goes.  OFF by default: the library cannot know when the caller frees an array, and a range that was
	// freed and mapped again while still registered takes the process down at the next copy (GPU memory access fault, measured:
	// profiles/r04_host_link_probes.log -- round 3's one unexplained abort).  For callers that vouch for the lifetime of their arrays.
	int register_host = 0;
	long register_min_bytes = 4l << 20;   // ... for arrays of at least this many bytes (below, the runtime's staged copy is as fast)
	// 1 (default): a host caller's x goes up on every call that uses it, like the reference, where *req aliases x and an edit
	// between two calls simply moves the iterate (for large pinned x the upload rides under the update, slice by slice);
	// 0: not when the device copy is what the library handed back and 256 probe values still agree -- for callers that vouch
	// they keep their hands off x while *req designates it (reference include/stochqn.h:364-366)
	// 2: the library finds out by itself -- a checksum of ALL of the caller's x (sqn_device.hpp: XHash; `hash_threads` host threads,
	// under the upload of the gradient) against the checksum of what the device holds, taken on the device at the end of the
	// last call.  Equal: no upload.  A change of any one coordinate changes the first sum for certain (the words are mixed one-to-one
	// before they are summed); other edits pass only on a collision of two 64-bit sums of mixed values.
	int x_upload = 1;
	int hash_threads = 0;           // host threads that checksum x (x_upload = 2); 0: min(8, half the hardware threads), split among the shards of a group
	long host_slice_min = 1l << 21; // host callers: vectors of fewer elements cross the link in one piece (a slice below ~8 MB is all launch overhead)
	int upload_slices = 8;          // host callers, three-pass form: pass 1 runs in this many slices, each as soon as its part of the gradient has arrived
	int apply_chunks = 8;           // host callers: the update pass runs in this many slices so that the download of x overlaps it
	long max_mirror_bytes = 0;      // > 0: cap on the device memory held by mirrors of host arrays (least recently used contexts are exported and dropped)
	// Device-resident callers of configurations in which no decision depends on device data (check_nan = 0 and
	// min_curvature = 0: no step and no pair can be rejected): run_* returns as soon as its kernels are enqueued.  The
	// context's stream is a blocking one, so the caller's next kernel on the null stream is ordered after them; nothing is
	// read back (buffer_rho / buffer_alpha / buffer_y stay untouched, the kappa rule is off), a device fault surfaces at
	// the caller's next synchronisation.  What is left per call is launch cost: DESIGN.md 4.1.
	bool async_device = false;
	// Which stream a call works on: the context's own (blocking flavour, ordered after the NULL stream) or the NULL stream
	// itself.  A device caller's gradient kernels run on the NULL stream (torch's default), and handing work from one
	// stream to the other costs about 15 us each way per call -- a third of a step below n ~ 1e6, nothing
	// that matters at n = 1e8.  The price of the NULL stream is that it serialises with every blocking stream of the process.
	// 0: always the own stream; 1: always the NULL stream; 2 (default): the NULL stream for device-resident callers of
	// problems up to 2^22 variables, the own stream otherwise.  async_device implies the NULL stream.
	int null_stream = 2;
	bool raw_reuse_cache = false;   // isolated entry points keep their cached inner products between calls (caller vouches for S, Y)
	bool verify_cache = false;   // debugging aid for device callers: recompute cached dots every call and compare
};
int default_grid_cap();
Options& options();

// Host-side rendezvous reducer: P shards of one problem driven by P host threads of this process,
// summed in rank order.  Used where RCCL cannot serve: shards that share a physical device (tests,
// the "virtual_devices" rehearsal of the single-process multi-device mode).
struct Loopback {
	int nranks = 0;
	std::mutex mu;
	std::condition_variable cv;
	int arrived = 0;
	long generation = 0;
	bool broken = false;            // a shard failed to show up: every later reduction fails immediately
	double patience_s = 120;        // how long a shard waits for the others before giving up (option "reducer_patience_s")
	std::vector<double> slots;      // [nranks][kRedMax]
};

// What a context sums its partial dot products through (fixed when the context is created).
struct Reducer {
	enum Kind { NONE = 0, RCCL, CUSTOM, LOOP } kind = NONE;
	void* comm = nullptr;           // ncclComm_t (RCCL)
	Loopback* loop = nullptr;       // LOOP
	int rank = 0, nranks = 1;
};

struct DevCtx {
	Reducer red;
	const void* key = nullptr;
	int kind = 0;
	int n = 0;
	size_t m = 0;
	size_t fsize = 0;
	double n_global = 0;

	View S, Y, sbak, ybak, gprev, xsum, xprev, H0, G, F;
	// bytes of device memory the mirrors above hold: what the registry reads of a context that another thread may be inside of
	// (the views themselves belong to the thread that is inside the call)
	std::atomic<size_t> mirrored{0};
	Scratch sc{};
	Profiler prof;
	int buf = 0;                       // ping-pong index of the next partial buffer
	int phase = 1;                     // sweep parity inside the current API call (reset by begin_call)
	double* pool = nullptr;            // one allocation behind sc.part/red/sy/yy/alpha/rho/report
	size_t pool_bytes = 0;
	double* fisher_t = nullptr;        // [fsize] F*s on device
	hipEvent_t x_pre_ev = nullptr;                    // option "x_prefetch": the upload of x that was started when the last call returned
	bool x_pre_pending = false;
	real* spec = nullptr;                             // x as the pending update will write it (option "spec_x"), source of the early slices
	real* stage[3] = {nullptr, nullptr, nullptr};     // device staging for host x / grad / hess_vec
	// host-caller path: the caller's arrays pinned in place, the side stream the download of x runs on while the
	// update pass is still working on later slices, and what is known about the device copy of x
	struct HostRange { const void* p = nullptr; size_t bytes = 0; };
	HostRange regs[6];                 // option "register_host": ranges this context pinned by itself
	int reg_turn = 0;
	struct Seen { const void* p = nullptr; unsigned long long call = 0; };
	Seen seen[8];                      // large host arrays of the recent calls: only one that has been there before is worth pinning
	unsigned long long call_index = 1; // API calls on this context (note_state)
	hipStream_t own_stream = nullptr;  // the context's own (blocking-flavour) stream; sc.stream is this or the NULL stream (option "null_stream")
	hipStream_t copy_stream = nullptr; // host -> device: slices of the gradient and of x on their way up
	hipStream_t down_stream = nullptr; // device -> host: slices of x (and of the direction) on their way down -- a stream of its own: the link is full duplex
	std::vector<hipEvent_t> chunk_ev, up_ev, xup_ev;
	double* carry = nullptr;           // accumulators of a sliced pass 1 between its launches (kernels.hip: Slice)
	size_t carry_count = 0;
	bool copy_busy = false;            // work was enqueued on copy_stream / down_stream during this call
	const void* x_host = nullptr;      // the host array stage[0] mirrors
	bool x_valid = false;              // stage[0] holds the caller's current x (the library wrote both; *req == x went back)
	static constexpr int kProbe = 256;
	double x_probe[kProbe];            // the caller's x at kProbe spread-out positions when it was last handed back
	XHash x_hash;                      // option "x_upload" = 2: checksum of stage[0] as the last call left it ...
	bool x_hash_valid = false;         // ... when that call got as far as taking it
	size_t x_hash_count = 0;
	int device = 0;                    // the HIP device the context (its mirrors, scratch and streams) lives on
	unsigned long long last_use = 0;   // registry clock at the last call (least-recently-used reclaim)
	bool in_call = false;              // between acquire() and the end of the API call: never reclaimed
	bool async_call = false;           // this call returns without synchronising (option "async_device"; set by open_call)
	bool no_spill = false;             // host memory for a spill of this context could not be had: leave it on the device
	void* spill = nullptr;             // the state a reclaimed predecessor of this context left in host memory (runtime.cpp: Spill)
	real* host_stage[2] = {nullptr, nullptr};         // host landing zones for *req / *req_vec
	bool host_stage_pinned[2] = {false, false};       // (pageable when pinned memory ran out)
	double* pin = nullptr;             // pinned host read-back block
	size_t pin_count = 0;
	// what the caller's struct looked like when the last call on this context returned; a call that
	// does not continue from there belongs to a different optimiser object at the same address
	bool attached = false;             // the reducer was bound and n_global agreed (first call on this context)
	bool fault = false;                // a launch or the stream reported an error: the call that saw it fails loudly
	bool wedged = false;               // a collective of this context can never end (peer gone, RCCL without ncclCommAbort): every call fails, no stream of it is waited for again, its memory is abandoned (runtime.cpp: wait_stream)
	bool has_last = false;
	size_t last_niter = 0;
	int last_section = 0;
	std::vector<char> rho_ok;          // per physical row: sc.sy / sc.yy hold this row's dots
	// per physical row: |s||y| / |s'y| of the pair (1 / cosine of the angle between s and y; < 0 = not known yet).
	// From all-reduced dots, so identical on every rank of a sharded run.  Pairs that are almost orthogonal
	// make the recursion amplify rounding errors by about this factor per pair; the three-pass form is only
	// used while every pair in use stays below option "kappa_max" (machines.cpp: pairs_tame).
	std::vector<char> sy_ok;           // per physical row r: the cached s_i'y_r of every pair i in use (three-pass form) are current
	std::vector<double> kappa;
	size_t verify_turn = 0;            // option verify_cache: which pair in use is re-derived on the next call
	double* kap_dev = nullptr;         // [3 m] landing zone of (s'y, s's, y'y) for rows whose kappa has to be computed
	void touch_row(size_t r) { rho_ok[r] = 0; sy_ok[r] = 0; kappa[r] = -1; }   // row r of S or Y was rewritten
	void forget_rows() { rho_ok.assign(m, 0); sy_ok.assign(m, 0); kappa.assign(m, -1.0); }

	int next_buf() { int b = buf; buf ^= 1; return b; }
};

// Event counters behind stochqn_hip_stat(): which form of the recursion each step took, how many reductions crossed
// the shards, what the context manager did.  Process-wide, relaxed atomics; stochqn_hip_stats_reset() zeroes them.
enum StatId {
	ST_STEP_THREE_PASS = 0, ST_STEP_SWEEPS, ST_STEP_PLAIN, ST_KAPPA_FALLBACK,
	ST_ALLREDUCE, ST_ALLREDUCE_DOUBLES, ST_CTX_CREATED, ST_CTX_RECLAIMED, ST_X_UPLOAD, ST_X_UPLOAD_SKIPPED,
	ST_HOST_REGISTERED, ST_X_AHEAD, ST_X_RESENT, ST_X_PREFETCH, ST_HOST_UNPIN_FAILED, ST_WORK_IN_FLIGHT, ST_HOST_PIN_DECLINED, ST_HOST_PIN_FOREIGN, ST_HOST_PIN_ERRORS, ST_COUNT
};
void stat_add(int id, long long v = 1);
// May [p, p + bytes) of ordinary host memory be page-locked in place (hipHostRegister)?  Only a range that has its pages to
// itself for as long as it lives: not one inside the program-break heap (malloc'ed blocks there share their first and last
// page with neighbours, and the break moves under them when the heap is trimmed), not one whose pages overlap a range that is
// page-locked already.  Whatever is declined still works: the runtime's pageable path carries it (runtime.cpp).
// "Yes" RESERVES the pages in the same critical section (two askers with a page in common cannot both be told yes): who then
// fails to register gives them back with note_unpinned(p).  own: a range of the asker that the new one replaces.
bool pinnable_in_place(const void* p, size_t bytes, const void* own = nullptr);
extern bool g_pin_probe_by_maps;   // tests/hostsim only: look a heap header up through /proc/self/maps even where process_vm_readv works
void note_pinned(const void* p, size_t bytes);          // bookkeeping of the ranges this library has registered, for the overlap rule
void note_unpinned(const void* p);

bool device_ready();                                   // a HIP device exists and is usable
// Every device / pinned-host allocation of the library goes through these two: they report failure
// instead of printing and carrying on, and honour the fault-injection option "fail_alloc_after"
// (tests: the k-th allocation from now fails once).
bool device_alloc(void** p, size_t bytes);
bool pinned_alloc(void** p, size_t bytes);
bool is_device_pointer(const void* p);

// Find or create the context of a workspace.  `fresh` tells the caller whether it was created now.
DevCtx* acquire(const void* key, int kind, int n, size_t m, size_t fsize, bool* fresh);
DevCtx* lookup(const void* key);
// lookup + mark the context as inside a call (never reclaimed meanwhile), under the registry lock; pair with end_use()
DevCtx* hold(const void* key);
void forget_rows_of(const void* key);                  // stochqn_hip_invalidate: drop the cached inner products of the context at `key`
// Destroy a context whose creation could not be completed (a mirror could not be had), KEEPING what a reclaimed predecessor
// left in host memory: the object's next call finds it again.  release() drops that state too (another object / explicit release).
void abandon_context(const void* key);
// Create the context of a workspace ahead of its first call (all its allocations, no collective): the shards of
// a multi-device group do this together, so that none of them can fail on memory while the others are already
// waiting in an all-reduce.
bool prepare_context(const void* key, int kind, int n, size_t m, size_t fsize, int stages = 0);   // stages: staging vectors for host x / grad / hess_vec made now
// registry key of the context behind the isolated entry points (stochqn_hip_two_loop / _take_step) for the arrays at `s_mem`
inline const void* raw_key(const void* s_mem) { return static_cast<const char*>(s_mem) + 1; }
// remember the caller-visible state on return; true if the context saw a HIP error during the call
void end_use(DevCtx* c);                                // the API call that acquired `c` is over (isolated entry points)
void attach_spill(DevCtx* c, size_t niter, int section);   // a freshly created context picks up what its reclaimed predecessor left
void detach_spill(DevCtx* c);
bool has_spill(const void* key);
bool export_spill(const void* key);
void enforce_mirror_cap();
bool note_state(const void* key, size_t niter, int section, bool req_is_x = true, const real* x = nullptr);
void release(const void* key);
void release_all();

// (Re)bind a view to the caller's pointer.  Host pointers get a device mirror of `count` doubles;
// with `import` the host contents are uploaded into a newly created mirror.
bool bind(DevCtx* c, View& v, real* caller, size_t count, bool import);
void export_view(DevCtx* c, View& v);                  // mirror -> caller's host array

// Pin [p, p + bytes) of the caller's host memory in place (option "register_host"); false = not pinned (too small,
// refused by the runtime, switched off): copies from / to it then go through the runtime's staging path as before.
bool ensure_registered(DevCtx* c, const void* p, size_t bytes);
// x of a host caller: upload unless the device copy is known to be current (option "x_upload"); stage[0] on return
real* stage_x(DevCtx* c, real* caller, size_t count);
bool x_is_current(DevCtx* c, const real* caller, size_t count);    // option "x_upload" = 0 and the device copy is what the caller still holds
void x_handed_back(DevCtx* c, const real* caller, size_t count);     // after the download of x has completed
// Option "x_upload" = 2: the checksum of a host buffer on `threads` threads of its own, started now and collected later (the
// caller's thread goes on enqueueing meanwhile).  xhash_start returns nullptr when no thread could be started; xhash_finish
// joins, frees the job and returns the sum.
struct XHashJob;
XHashJob* xhash_start(const void* buf, size_t bytes, int threads);
XHash xhash_finish(XHashJob* job);
int xhash_threads(const DevCtx* c);
bool ensure_copy_stream(DevCtx* c, int chunks);
bool ensure_upload_slices(DevCtx* c, int slices, size_t carry_count);      // side stream, events and carry scratch of a sliced pass 1
bool ensure_stage(DevCtx* c, int which);               // device staging vector `which` exists
real* stage_in(DevCtx* c, int which, real* caller, size_t count, bool host);       // H2D if host; nullptr = out of memory
real* host_landing(DevCtx* c, int slot);               // host landing zone for *req / *req_vec (pinned if possible)
void begin_call(DevCtx* c);                            // refresh options, restart the sweep parity
void sync(DevCtx* c);
// hipStreamSynchronize, bounded when the stream carries RCCL collectives (option "reducer_patience_s": a peer that never arrives
// gets the communicator aborted and the call failed instead of a wait for ever); plain hipStreamSynchronize otherwise
hipError_t wait_stream(DevCtx* c, hipStream_t s);                                  // stream sync + profiler collection

// One named range per API call for `rocprofv3 --marker-trace` (roctx), only with STOCHQN_HIP_ROCTX=1 in the
// environment; the marker library is dlopen()ed then, never linked.
struct ApiRange {
	ApiRange(const char* name, int section);
	~ApiRange();
	bool on = false;
};

// multi-GPU
int comm_nranks();
void comm_attach(DevCtx* c);                           // install the all-reduce hook, compute n_global
Reducer current_reducer();                             // the calling thread's binding, else the process-wide communicator
void set_thread_reducer(const Reducer& r);             // contexts created by this thread from now on reduce through `r`
bool comm_init_all(int ndev, const int* devices, void** comms_out);   // ncclCommInitAll (one process, ndev devices)
void comm_destroy(void* comm);

// A HIP call whose failure nobody can repair on the spot: it is reported, and the API call the calling thread is inside of
// FAILS (-1000 / invalid_input; machines.cpp: after_call and the isolated entries) instead of carrying on with whatever the
// failed copy / synchronisation left behind.  (Reference src/stochqn.c:362-432, 1033-1035: a failure is a message and an
// error code, never an abort and never a silently wrong result.)
void note_hip_failure(const char* expr, hipError_t e, const char* file, int line);
bool take_hip_failure();               // true, once, when a SQN_HIP_OK call failed on this thread since the last take
#define SQN_HIP_OK(expr)                                                                             \
	do {                                                                                             \
		hipError_t e_ = (expr);                                                                      \
		if (e_ != hipSuccess) ::sqn::note_hip_failure(#expr, e_, __FILE__, __LINE__);               \
	} while (0)

}  // namespace sqn
