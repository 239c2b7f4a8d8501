// sqn_device.hpp -- interface between the host state machines (machines.cpp) and the HIP
// kernels (kernels.hip).  Everything here takes DEVICE pointers and enqueues work on a stream;
// nothing synchronises.  Names follow the reference's domain: correction pairs (s, y), the
// two-loop recursion, the Fisher ring, the diagonal rescale.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace sqn {

// Element type of every n-vector (x, grad, rows of S / Y / F ...).  All arithmetic, every partial sum
// and every scalar of the recursion is double in both builds; the float build (-DSQN_USE_FLOAT,
// libstochqn_f32.so) only stores and streams the vectors in single precision.
#ifdef SQN_USE_FLOAT
typedef float real;
#else
typedef double real;
#endif
constexpr int kVec = 16 / (int) sizeof(real);    // elements per 16-byte pack: 2 doubles or 4 floats

constexpr int kBlock = 256;       // threads per workgroup (4 wave64)
constexpr int kMaxGrid = 2048;    // 256 CUs x 8 resident workgroups; also the partial-sum stride
constexpr int kMaxSums = 3;       // sums one sweep can produce (s'y, s's, y'y)
constexpr int kRowsMax = 48;      // rows one rows-dot launch can take (one accumulator per row and lane)
constexpr int kPairsMax3 = 48;    // largest ring the three-pass form handles (its passes take k rows each; one lane per pair in the recursion)
constexpr int kRedMax = 128;      // doubles per all-reduce landing zone and quantities per rows-dot pass (>= 2*kPairsMax3: pass 1 with the new column)

// Kernel ids for the built-in HIP-event profiler (stochqn_hip_profile_*).
enum KernelId {
	K_FIRST = 0, K_BWD, K_MID, K_FWD, K_FWD_LAST, K_APPLY, K_PAIR_S, K_PAIR_Y_DIFF, K_PAIR_Y_HV,
	K_DOTS3, K_FISHER_T, K_FISHER_Y, K_FIN, K_SMALL, K_COPY, K_SDOT, K_SDOT2, K_QDOT, K_SADD, K_XHASH, K_COUNT
};
const char* kernel_name(int id);

// HIP-event stopwatch around every kernel launch (one event pair per launch, resolved after the
// stream has been synchronised).  Durations are accumulated per KernelId.
struct Profiler {
	struct Pending { int id; hipEvent_t a, b; };
	std::vector<Pending> pending;
	std::vector<hipEvent_t> pool;
	double total_ms[K_COUNT] = {0};
	long long launches[K_COUNT] = {0};

	hipEvent_t get()
	{
		if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
		hipEvent_t e;
		(void) hipEventCreate(&e);
		return e;
	}
	// begin returns the index of the pair it opened and end takes it: scopes NEST (a sliced pass hands each finished slice to
	// a callback that launches kernels of its own inside the pass's scope), so "the last pair opened" is not the one to close.
	// (Until round 5 end() closed pending.back(): the outer scope's second event was never recorded, hipEventElapsedTime on it
	// left "invalid resource handle" behind, and the next synchronisation of a PROFILED host caller failed the call.)
	size_t begin(int id, hipStream_t st)
	{
		Pending p{id, get(), get()};
		(void) hipEventRecord(p.a, st);
		pending.push_back(p);
		return pending.size() - 1;
	}
	void end(size_t which, hipStream_t st) { if (which < pending.size()) (void) hipEventRecord(pending[which].b, st); }
	void collect()  // call only after the stream has been synchronised
	{
		for (auto& p : pending) {
			float ms = 0;
			if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { total_ms[p.id] += ms; launches[p.id]++; }
			else (void) hipGetLastError();           // a pair that cannot be read is not counted -- and must not fail somebody's call later
			pool.push_back(p.a);
			pool.push_back(p.b);
		}
		pending.clear();
	}
	void reset()
	{
		for (int i = 0; i < K_COUNT; i++) { total_ms[i] = 0; launches[i] = 0; }
	}
};

// A reduction result as the NEXT kernel sees it: `count` partial sums per quantity, quantity j
// starting at parts[j * stride].  Single GPU: the producer's per-workgroup partials (count =
// producer grid).  Multi GPU: the all-reduced scalars (count = 1, stride = 1).
struct Partials {
	const double* parts;
	int count;
	int stride;
};

// Per-context device scratch (owned by DevCtx in machines.cpp).
struct Scratch {
	hipStream_t stream;
	double* part[2];      // two ping-pong partial buffers, each kMaxSums * kMaxGrid doubles
	double* fisher_part;  // [fisher_size][kMaxGrid] partials of pass 1 (lazily sized)
	double* red[2];       // all-reduce landing zones, kRedMax doubles each
	double* sy;           // [m] s'y of each physical row (rho = 1/sy)
	double* yy;           // [m] y'y of each physical row (gamma = sy/yy)
	double* alpha;        // [m] alpha by logical index
	double* rho;          // [m] rho by logical index (for buffer_rho write-back)
	double* report;       // [4]: bad flag, sum r^2, nonfinite count, spare
	double* rows_part[2]; // two [kRedMax][kMaxGrid] partial buffers of the rows-dot passes (pass 1, pass 2)
	double* gsy;          // [m][m] cached block  gsy[i*m+j] = s_i'y_j   (physical rows; pair i older than pair j)
	unsigned* ticket;     // arrival counter of the pair kernels' last-workgroup verdict (zero between launches)
	int grid_cap;         // max workgroups per sweep (<= kMaxGrid); default = one per CU
	bool rows_split;      // pass 1 without a second probe as the row-split rows-dot kernel (float build)
	int fisher_rows;      // Fisher rows one workgroup accumulates per pass over its columns: 8, 16 or 32 (the all-rows-per-lane kernel: fisher_split = 0)
	bool fisher_split;    // Fisher pass 1 with the rows divided among the 8 waves of a workgroup: s is fetched once per 128 rows (default)
	int fisher_split_per_cu; // workgroups per CU of that kernel (0 = what the occupancy query says, at most 4)
	int fisher_tile;      // column tiles of 64 packs such a workgroup takes per trip: 1 or 2
	int fisher_lag;       // column tiles the waves of such a workgroup may drift apart before a barrier brings them together (0 = no barrier)
	int qdot_per_cu, sadd_per_cu, sdot2_per_cu, sdot_per_cu;   // workgroups per CU of the three-pass kernels (0 = default)
	int sdot_tile;        // pass 1 (single probe, <= 24 rows): adjacent column tiles a workgroup takes per iteration, 1 or 2 (kernels.hip: k_rows_dot_all U)
	int pair_per_cu;      // workgroups per CU of the pair kernels (s, y = g - g_prev, y = Hv; 0 = default: 1)
	double keep_tail;     // three-pass form: fraction of r0 / r (the part written last) stored with the default policy instead of sc1 nt
	uint32_t phase_inv;   // pass 2 / pass 3: 2^32 / (ticks of the 100 MHz clock per store phase), 0 = every pack stored at once (kernels.hip: Parked)
	bool nontemporal;     // stream S/Y/F rows with nt loads
	bool reverse;         // alternate the traversal direction from sweep to sweep (Infinity-Cache reuse of q / r)
	int* phase;           // sweep counter of the current API call (parity = direction)
	Profiler* prof;       // NULL unless stochqn_hip_profile_enable(1)
	// multi-GPU: in-place sum of `count` doubles at `buf` across all ranks, enqueued on `stream`.
	// NULL on a single GPU.
	void (*allreduce)(void* user, double* buf, int count, hipStream_t stream);
	void* user;
};

int sweep_grid(const Scratch& sc, size_t n, int per_cu = 1);

// ---- two-loop chain ---------------------------------------------------------------------------
// first sweep: optional side effects on the raw gradient, then either the newest pair's s'q
// (s_newest != NULL) or the guard sums (sum dir^2, nonfinite) of the plain / rescaled gradient.
struct FirstArgs {
	real* q;               // gradient, n
	const real* s_newest;  // NULL when the ring is empty
	real* gprev_out;       // oLBFGS: grad_prev <- g            (nullable)
	real* frow_out;        // adaQN : Fisher row  <- g           (nullable)
	real* G;               // adaQN : grad_sum_sq in/out         (nullable)
	real* H0_out;          // adaQN : g/sqrt(G+eps) goes here; NULL -> into q itself
	double rmsprop_weight, scal_reg;
};
Partials launch_first(const Scratch& sc, int buf, size_t n, const FirstArgs& a);

// backward sweep i: alpha_i = rho_i * <in>; q -= alpha_i*y_i; out = s_prev'q
Partials launch_bwd(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int logical_i,
                    const real* y_i, real* q, const real* s_prev);

// middle sweep: alpha_0; q -= alpha_0*y_0; r = H0-scaling(q); out = y_0'r
struct MidScale {
	const double* sy_newest;  // gamma = *sy_newest / *yy_newest when both non-NULL
	const double* yy_newest;
	double h0;                // else scalar h0 when H0 == NULL
	const real* H0;         // else element-wise
};
Partials launch_mid(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row,
                    const real* y_0, real* q, const MidScale& ms);

// forward sweep i: beta = rho_i*<in>; r += (alpha_i-beta)*s_i; out = y_next'r
Partials launch_fwd(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int logical_i,
                    const real* s_i, real* r, const real* y_next);

// last forward sweep; out = (sum r^2, nonfinite).  With `fuse` non-NULL (check_nan == 0) the
// position update is applied in the same pass and nothing is produced.
struct ApplyArgs {
	real* x;            // n
	real* x_sum;        // nullable (SQN / adaQN)
	real* s_slot;       // nullable (oLBFGS: s <- -step*r, grad <- -step*r)
	double step;
};
Partials launch_fwd_last(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int logical_i,
                         const real* s_i, real* r, const ApplyArgs* fuse);

// guarded position update: bad = nonfinite>0 || sqrt(sum r^2) > 1e3*n_global; writes report[0..2]
void launch_apply(const Scratch& sc, size_t n, double n_global, Partials guard, const real* r_in, real* grad_out,
                  const ApplyArgs& a, bool guarded);

// ---- correction pairs ---------------------------------------------------------------------------
void launch_pair_s(const Scratch& sc, size_t n, real* x_sum, double inv_L, bool scale, const real* x_avg_prev,
                   real* s_out);
// The verdict on a new pair (check_min_curvature, reference src/stochqn.c:883-900) taken by the pair kernel itself: the last
// workgroup to arrive totals the partials of (s'y, s's, y'y) in index order -- the sums k_verdict would form, bit for bit --,
// decides, commits s'y and y'y of an accepted pair and fills out[0..3] = s'y, s's, y'y, rejected.  One launch less per pair
// (14 us of a 0.8 ms oLBFGS step at n = 1e7).  Single device only: with a reducer the sums have to be all-reduced first.
// A launcher that was handed VerdictArgs and used them returns Partials with parts == NULL.
struct VerdictArgs { double min_curvature; double* sy_dst; double* yy_dst; double* out; };
Partials launch_pair_y_diff(const Scratch& sc, int buf, size_t n, const real* g, const real* g_prev,
                            const real* s, double lambda, real* y_out, const VerdictArgs* verdict = nullptr);
Partials launch_pair_y_hv(const Scratch& sc, int buf, size_t n, const real* hv, const real* s, real* y_out,
                          real* x_sum, real* x_avg_prev, const VerdictArgs* verdict = nullptr);
Partials launch_dots3(const Scratch& sc, int buf, size_t n, const real* s, const real* y);
// Fisher product y = F'(F s)/fu; returns the (s'y, s's, y'y) partials; t_out[fu] receives F s.
Partials launch_fisher(const Scratch& sc, int buf, size_t n, const real* F, size_t fu, const real* s,
                       double* t_dev, real* y_out);

// ---- rows of the ring handed to one launch ----------------------------------------------------------
struct RowSet {
	const real* row[kRowsMax];
	int count;
};
struct CoefArgs {
	int k;                    // pairs in use
	int m;                    // ring size (leading dimension of the cached block)
	int rows[kPairsMax3];     // physical row of logical pair i (oldest first)
	double h0;                // > 0: scalar H0, else gamma from the newest pair
};

// ---- three-pass form: S twice, Y once -- (3k+5) n words (kernels.hip "three-pass form") -----------------------
// The recursion only needs the inner products of g with the stored vectors and the inner products among the
// stored vectors; the latter (s_a'y_b, a older than b) are cached when a pair enters the ring.  Same quantities as
// reference src/stochqn.c:663-708, other association of the floating-point sums.
// pass 1: quantities [0,k) s_i'g, and with probe_y (the pair that just entered, ring row r) [k,2k) s_i'y_r
// `feed` (host callers): the pass runs in feed->slices slices of the traversal; before the kernel of a slice is enqueued
// feed->arrive(user, lo, hi, slice) is called for the element range [lo, hi) of g that slice reads -- the caller enqueues the
// upload of that range and makes the stream wait for it.  Bit-identical to the unsliced pass (kernels.hip: Slice).
// carry: device scratch of carry_count doubles; a pass whose grid needs more than that (sdot_carry_count for pass 1, two
// per lane of the grid for pass 3) runs as one launch.
struct SliceFeed {
	int slices;
	double* carry;
	size_t carry_count;
	void (*arrive)(void* user, size_t lo, size_t hi, int slice);
	void* user;
};
bool sdot_can_slice(const Scratch& sc, const RowSet& s_rows, const real* g, real* copy_out, const real* probe_y);
size_t sdot_carry_count(const Scratch& sc, size_t n, int k);
Partials launch_sdot(const Scratch& sc, size_t n, const RowSet& s_rows /*logical order*/, const real* g, real* copy_out, const real* probe_y,
                     const SliceFeed* feed = nullptr);
struct QdotScale {
	const real* H0_in;      // caller-supplied diagonal, or NULL
	real* G;                // adaQN: grad_sum_sq (in/out), or NULL
	real* H0_out;           // adaQN: receives g/sqrt(G+eps)
	real* frow_out;         // adaQN: Fisher row <- raw gradient, nullable
	double rmsprop_weight, scal_reg;
};
// pass 2: q0, r0 (replaces g), v_i = y_i'r0 -> quantities [0,k).  The backward recursion -- totals of pass 1, (fresh_row >= 0) the
// new column of the cached s_old'y_new block, alpha_i, the scale of q0 -- runs in the pass's own prologue, in every workgroup.
Partials launch_qdot(const Scratch& sc, size_t n, const RowSet& y_rows /*logical order*/, real* g, const QdotScale& q,
                     const Partials& pass1, const CoefArgs& a, int fresh_row);
// pass 3: r = r0 + sum c_j s_j; returns the guard partials (sum r^2, nonfinite).  The forward recursion (c_j from the totals of
// pass 2) runs in the prologue.  `drain`: the pass in slices (sadd_can_slice), drain->arrive(user, lo, hi, slice) right after the
// launch that makes r[lo, hi) final.
bool sadd_can_slice(const Scratch& sc, size_t n, const RowSet& s_rows, const real* r, const SliceFeed* drain);
Partials launch_sadd(const Scratch& sc, int buf, size_t n, const RowSet& s_rows, real* r, const Partials& pass2, const CoefArgs& a,
                     const SliceFeed* drain = nullptr);
// out = x - step * r, the expression (and the bits) of the guarded update, without touching x
void launch_spec_x(const Scratch& sc, size_t n, const real* r, const real* x, double step, real* out);
void launch_store_column(const Scratch& sc, Partials in /*k: s_i'y_col*/, const CoefArgs& a, int col_row);

// Checksum of the bit pattern of a vector (host callers, option "x_upload" = 2: is the caller's x still what the device holds?).
// The buffer as 64-bit little-endian words w_0 .. w_{W-1} (a last partial word zero-extended).  Every word is first mixed with
// its position by a BIJECTION of 64 bits that is not linear over the integers or over GF(2) (splitmix64's finaliser):
// h_i = mix(w_i xor (i G + K)); then A = sum h_i and B = sum (2i+1) rot32(h_i), both mod 2^64.  A change of any ONE word changes
// A for certain (mix is one-to-one); any other edit -- sign flips of two coordinates, x -> -x, a swap of two words: the edits a
// plain sum of the words is blind to -- goes unnoticed only if two 64-bit sums of unrelated mixed values collide.
// Integer sums: the same on the device (launch_xhash, into out[0..1] viewed as two 64-bit words) and on the host (xhash_host,
// any partition, any order).
struct XHash { unsigned long long a = 0, b = 0; };
#if defined(__HIPCC__)
__host__ __device__
#endif
inline unsigned long long xhash_mix(unsigned long long w, unsigned long long i)
{
	unsigned long long z = w ^ (i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	return z ^ (z >> 31);
}
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void xhash_word(unsigned long long w, unsigned long long i, unsigned long long& a, unsigned long long& b)
{
	const unsigned long long h = xhash_mix(w, i);
	a += h;
	b += (2 * i + 1) * ((h << 32) | (h >> 32));
}
void launch_xhash(const Scratch& sc, const real* x, size_t n, double* out2);       // ADDS into out2[0..1]: zero them first, on sc.stream
void xhash_host(const void* buf, size_t bytes, size_t word_lo, size_t word_hi, XHash* out);    // words [lo, hi) of the buffer (runtime.cpp)
inline size_t xhash_words(size_t bytes) { return (bytes + 7) / 8; }

// reduce `nsums` partial arrays to scalars: out[j] = sum_b parts[j*stride+b]
void launch_fin(const Scratch& sc, Partials in, int nsums, double* out);
// sy_dst <- total of quantity 0, yy_dst <- total of quantity 2 (device-side commit of a pair's dots)
void launch_commit(const Scratch& sc, Partials in, double* sy_dst, double* yy_dst);
// totals of (s'y, s's, y'y), accept / reject against min_curvature, commit of s'y and y'y when accepted;
// out[0..3] = s'y, s's, y'y, rejected flag
void launch_verdict(const Scratch& sc, Partials in, double min_curvature, double* sy_dst, double* yy_dst, double* out);
// ---- synthetic inputs for measurement (counter-based, shard-invariant) ------------
uint64_t synth_key(uint64_t seed, uint64_t stream, uint64_t t);
void launch_synth_uniform(hipStream_t stream, real* out, size_t count, uint64_t first, uint64_t key, double a, double b);
void launch_synth_grad(hipStream_t stream, real* g, const real* d, const real* x, size_t count, uint64_t first, uint64_t key, double amp);
void launch_synth_batch_row(hipStream_t stream, real* row, const real* d, size_t count, uint64_t first, uint32_t k, uint32_t bs);
// tiny helpers
void launch_scale(const Scratch& sc, size_t n, real* x, double a);

}  // namespace sqn
