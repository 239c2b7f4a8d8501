// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the stochastic quasi-Newton step.
//
// The first half: every kernel is one *sweep*: a grid-stride pass over n doubles that fuses an element-wise
// update with the dot product the NEXT step of the recursion needs, so that each n-vector is
// read once and written at most once per sweep.  The chain for one two-loop recursion
// (reference src/stochqn.c:663-708) over k stored pairs is
//
//     first      p = s_{k-1}'q                                   (+ oLBFGS/adaQN side effects)
//     bwd  i     alpha_i = rho_i p ; q -= alpha_i y_i ; p = s_{i-1}'q          i = k-1 .. 1
//     mid        alpha_0 ; q -= alpha_0 y_0 ; r = H0 q ; p = y_0'r
//     fwd  i     beta = rho_i p ; r += (alpha_i - beta) s_i ; p = y_{i+1}'r    i = 0 .. k-2
//     fwd_last   r += (alpha_{k-1} - beta) s_{k-1} ; p = (r'r, #nonfinite)
//     apply      guard (reference src/stochqn.c:825-835) ; x -= step r ; x_sum += x
//
// = 8*k*n words of compulsory HBM traffic for the two-loop.  A sweep's
// reduction is NOT finished inside the sweep: every workgroup stores one partial per quantity
// and every workgroup of the following kernel re-adds those partials in a fixed order
// (`total_of`).  That removes the 1-block "finalise" launches from the dependency chain, needs
// no atomics and no inter-workgroup hand-off, and is bit-reproducible for a given (n, grid).
//
// Memory access: 16-byte (double2) loads/stores per lane, fully coalesced (1 KiB per wave
// instruction), `kUnroll` independent packs per lane in flight; correction-pair rows and Fisher
// rows are streamed with non-temporal loads (they are dead until the next optimiser step) so
// that q / r keep what cache residency they can.  These are BLAS-1 reductions at 0.25 flop/B:
// no MFMA; in the sweeps LDS only carries the per-workgroup reduction.
//
// The second half of the file is the default, three-pass form of the same recursion over cached inner products
// between the stored pairs -- pass 1 (rows-dot: S g and, after a new pair, its column of the cached block), pass 2
// (q0, r0, Y'r0 with Y held in registers), pass 3 (r = r0 + S c), the scalar recursions in the prologues of
// passes 2 and 3.  DESIGN.md section 3.
#include "sqn_device.hpp"

#include <cmath>
#include <type_traits>

namespace sqn {

namespace {

constexpr int kUnroll = 2;
constexpr int kWaves = kBlock / 64;

typedef double d2 __attribute__((ext_vector_type(2)));
typedef real rvec __attribute__((ext_vector_type(kVec)));     // one 16-byte pack of vector elements
// The same packs as they sit in memory: element-aligned only.  Row j of the ring starts at element
// j*n, so for odd n every other row of S, Y and F is off the 16-byte grid.  gfx950 executes
// global_load / global_store_dwordx4 at any element alignment (the compiler emits them for these
// types); measured on the rows-dot pass at n = 1e8: 5.39 ms with every other row misaligned against
// 5.15 ms aligned -- and 13.8 ms for the element-wise path that alignment gating used to fall back
// to (profiles/src/tune5.hip).
typedef rvec rvec_u __attribute__((aligned(sizeof(real))));

// W elements of a vector, widened to double for the arithmetic.  W = kVec: one 16-byte access per
// lane (any element alignment); W = 1: one element.
template <int W> struct Pack { double v[W]; };

template <int W, bool NT> __device__ __forceinline__ Pack<W> ld(const real* p, uint32_t i)
{
	Pack<W> r;
	if constexpr (W == kVec) {
		const rvec t = NT ? __builtin_nontemporal_load(reinterpret_cast<const rvec_u*>(p + i))
		                  : *reinterpret_cast<const rvec_u*>(p + i);
		#pragma unroll
		for (int k = 0; k < W; k++) r.v[k] = (double) t[k];
	} else {
		static_assert(W == 1, "packs are 16 bytes or one element");
		r.v[0] = (double) (NT ? __builtin_nontemporal_load(p + i) : p[i]);
	}
	return r;
}

// The same load without widening: kernels that keep many row packs in flight (rows-dot, Fisher,
// combine) hold them as loaded (a float pack is 4 registers, widened it would be 8) and widen at use.
template <int W> struct RPack { real v[W]; };

template <int W, bool NT> __device__ __forceinline__ RPack<W> ldr(const real* p, uint32_t i)
{
	RPack<W> r;
	if constexpr (W == kVec) {
		const rvec t = NT ? __builtin_nontemporal_load(reinterpret_cast<const rvec_u*>(p + i))
		                  : *reinterpret_cast<const rvec_u*>(p + i);
		#pragma unroll
		for (int k = 0; k < W; k++) r.v[k] = t[k];
	} else {
		r.v[0] = NT ? __builtin_nontemporal_load(p + i) : p[i];
	}
	return r;
}

template <int W> __device__ __forceinline__ void st(real* p, uint32_t i, const Pack<W>& a)
{
	if constexpr (W == kVec) {
		rvec t;
		#pragma unroll
		for (int k = 0; k < W; k++) t[k] = (real) a.v[k];
		*reinterpret_cast<rvec_u*>(p + i) = t;
	} else p[i] = (real) a.v[0];
}

// write-once outputs (new s / y rows, Fisher row): keep them out of the caches
template <int W> __device__ __forceinline__ void st_nt(real* p, uint32_t i, const Pack<W>& a)
{
	if constexpr (W == kVec) {
		rvec t;
		#pragma unroll
		for (int k = 0; k < W; k++) t[k] = (real) a.v[k];
		__builtin_nontemporal_store(t, reinterpret_cast<rvec_u*>(p + i));
	} else __builtin_nontemporal_store((real) a.v[0], p + i);
}

// The single store stream of pass B (1 of 2k+2 streams): agent-scope, non-temporal cache policy on the store
// (gfx942 / gfx950 "sc1 nt").  Measured on the pass-B micro-benchmark (profiles/src/tune7.hip,
// profiles/r02_tune_store_policy.log, n = 1e8, k = 20): 6.01 ms plain, 5.98 ms nt, 5.84 ms sc1, 5.75 ms sc1 nt
// (no store at all: 4.77 ms).  There is no builtin for the sc bits, hence the one line of assembly.
template <int W> __device__ __forceinline__ void st_stream(real* p, uint32_t i, const Pack<W>& a)
{
	if constexpr (W == kVec) {
		rvec t;
		#pragma unroll
		for (int k = 0; k < W; k++) t[k] = (real) a.v[k];
		asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p + i), "v"(t) : "memory");
	} else p[i] = (real) a.v[0];
}

// ---- clock-phased stores (round 4) ---------------------------------------------------------------------------------------
// One store stream among ~20 read streams costs several times its bytes when the stores trickle out between the loads
// (DESIGN.md 3.2).  What cures it is the WHOLE CHIP writing at the same moment and reading the rest of the time: every lane
// parks its results in LDS (kParkSlots packs per lane, 128 KB per workgroup: the passes use no LDS otherwise) and every wave
// flushes its parked packs when the chip-wide 100 MHz counter (s_memrealtime) enters a new period of `phase_ticks` ticks --
// or when its slots are full.  No communication, no barrier: the clock is the only thing the 1024 waves share, a wave never
// waits for another one, and nothing but the moment of a store changes (same values, same addresses, same bits).
// profiles/src/tune9.hip, tune10.hip: pass 2 at n = 1e8, k = 20: 2.98 ms storing at once, 2.68 ms phased (2.50 ms without
// the store); a grid barrier instead of the clock: 2.84 ms.
constexpr int kParkSlots = 32;
template <int W> struct Stored { typedef real type; };
template <> struct Stored<kVec> { typedef rvec type; };

template <int W> __device__ __forceinline__ typename Stored<W>::type to_stored(const Pack<W>& a)
{
	if constexpr (W == kVec) {
		rvec t;
		#pragma unroll
		for (int k = 0; k < W; k++) t[k] = (real) a.v[k];
		return t;
	} else return (real) a.v[0];
}

// a parked pack on its way out.  pol 0: the default store policy; 1: sc1 nt like st_stream; 2: non-temporal like st_nt
template <int W> __device__ __forceinline__ void st_stored(real* p, uint32_t i, const typename Stored<W>::type& t, int pol)
{
	if constexpr (W == kVec) {
		if (pol == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p + i), "v"(t) : "memory");
		else if (pol == 2) __builtin_nontemporal_store(t, reinterpret_cast<rvec_u*>(p + i));
		else *reinterpret_cast<rvec_u*>(p + i) = t;
	} else {
		if (pol == 2) __builtin_nontemporal_store(t, p + i); else p[i] = t;
	}
}

// period number of the chip-wide clock: floor(ticks * inv / 2^32) on the low 32 bits of the counter (it wraps every 43 s: one
// flush out of turn)
__device__ __forceinline__ uint32_t phase_epoch(uint32_t inv) { return __umulhi((uint32_t) __builtin_amdgcn_s_memrealtime(), inv); }

// The parked results of one lane, NS output vectors per pack (pass 2 of adaQN writes four: r0, G, H0 and the Fisher row): pack
// b of stream s of lane t at slots[(b * NS + s) * kBlock + t] (a lane reads back only what it wrote itself, so no barrier is
// involved); pack numbers p_first, p_first + stride, ...  The slots are shared out among the streams and the period shrinks
// with them, so that the clock -- not the full slots -- is what makes the waves flush.
template <int W, int NS = 1> struct Parked {
	static constexpr int kCap = kParkSlots / NS;
	typename Stored<W>::type* slots;
	real* out[NS];                 // where stream s goes (NULL: nowhere)
	int pol[NS];                   // 0: default store policy; 1: sc1 nt (stream 0: except the tail from keep_from on); 2: non-temporal
	uint32_t p_first, epoch, inv, stride, last, keep_from;
	bool rev;
	int b;
	__device__ __forceinline__ void open(typename Stored<W>::type* lds, uint32_t inv_, uint32_t stride_, bool rev_, uint32_t last_, uint32_t keep_from_)
	{
		slots = lds; inv = inv_ > 0xFFFFFFFFu / (uint32_t) NS ? 0xFFFFFFFFu : inv_ * (uint32_t) NS; stride = stride_; rev = rev_; last = last_; keep_from = keep_from_;
		b = 0; p_first = 0; epoch = phase_epoch(inv);
	}
	__device__ __forceinline__ void flush()
	{
		#pragma unroll 1
		for (int bb = 0; bb < b; bb++) {
			const uint32_t p = p_first + (uint32_t) bb * stride;
			const uint32_t i = (rev ? last - p : p) * W;
			#pragma unroll
			for (int s = 0; s < NS; s++)
				if (out[s]) st_stored<W>(out[s], i, slots[(bb * NS + s) * kBlock + threadIdx.x], (s == 0 && p >= keep_from) ? 0 : pol[s]);
		}
		b = 0;
	}
	// park the results of pack p; flush when the slots are full or the clock has entered a new period
	__device__ __forceinline__ void put(const Pack<W> (&v)[NS], uint32_t p)
	{
		if (b == 0) p_first = p;
		#pragma unroll
		for (int s = 0; s < NS; s++) slots[(b * NS + s) * kBlock + threadIdx.x] = to_stored<W>(v[s]);
		b++;
		const uint32_t e = phase_epoch(inv);
		if (b == kCap || e != epoch) { epoch = e; flush(); }
	}
};

__device__ __forceinline__ double wave_sum(double v)
{
	#pragma unroll
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
	return v;
}

// Sum over the workgroup, returned to every thread.  Fixed tree: 6 shuffle steps per wave, then
// the kWaves wave sums in wave order.
__device__ __forceinline__ double block_sum(double v, double* sh)
{
	v = wave_sum(v);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
	__syncthreads();
	double t = sh[0];
	#pragma unroll
	for (int w = 1; w < kWaves; w++) t += sh[w];
	return t;
}

// Total of the previous sweep's partials, identical in every workgroup of the grid.
__device__ __forceinline__ double total_of(const double* parts, int count, double* sh)
{
	double a = 0;
	for (int i = threadIdx.x; i < count; i += kBlock) a += parts[i];
	return block_sum(a, sh);
}

// Sum over the wave, returned to every lane.
__device__ __forceinline__ double wave_sum_all(double v) { return __shfl(wave_sum(v), 0, 64); }

// Total of one partial array computed by ONE wave (lane-strided adds, then the shuffle tree); lets
// the 1-workgroup scalar kernels reduce kWaves quantities at a time without workgroup barriers.
__device__ __forceinline__ double wave_total_of(const double* parts, int count)
{
	const int lane = threadIdx.x & 63;
	double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
	int i = lane;
	for (; i + 192 < count; i += 256) {       // four independent loads in flight per lane
		a0 += parts[i]; a1 += parts[i + 64]; a2 += parts[i + 128]; a3 += parts[i + 192];
	}
	for (; i < count; i += 64) a0 += parts[i];
	const double a = wave_sum((a0 + a1) + (a2 + a3));
	return __shfl(a, 0, 64);
}

// The sweep skeleton.  Op supplies `In<W> load<W>(i)` (loads only) and `apply<W>(i, in, acc)`
// (arithmetic + stores) so that all loads of an unrolled group are issued before its first store.
template <int W, int NP, class Op>
__device__ __forceinline__ void sweep(uint32_t n, bool rev, const Op& op, double (&acc)[NP > 0 ? NP : 1])
{
	const uint32_t packs = n / W;
	const uint32_t stride = gridDim.x * kBlock;
	const uint32_t last = packs - 1;
	uint32_t p = blockIdx.x * kBlock + threadIdx.x;
	// `rev` walks the packs from the far end: the previous sweep finished there, so the part of
	// q / r it wrote last is the part still sitting in the 256 MiB Infinity Cache.
	// packs <= 2^31 and stride*(kUnroll) <= 2^21, so the sums below stay inside 32 bits
	for (; p + (kUnroll - 1) * stride < packs; p += kUnroll * stride) {
		typename Op::template In<W> in[kUnroll];
		#pragma unroll
		for (int u = 0; u < kUnroll; u++) { const uint32_t i = p + u * stride; in[u] = op.template load<W>((rev ? last - i : i) * W); }
		#pragma unroll
		for (int u = 0; u < kUnroll; u++) { const uint32_t i = p + u * stride; op.template apply<W>((rev ? last - i : i) * W, in[u], acc); }
	}
	for (; p < packs; p += stride) {
		const uint32_t i = rev ? last - p : p;
		typename Op::template In<W> in = op.template load<W>(i * W);
		op.template apply<W>(i * W, in, acc);
	}
	if constexpr (W > 1) {  // odd tail (n not a multiple of W): the last workgroup's first lanes
		const uint32_t i = packs * W + threadIdx.x;
		if (blockIdx.x == gridDim.x - 1 && i < n) {
			typename Op::template In<1> in = op.template load<1>(i);
			op.template apply<1>(i, in, acc);
		}
	}
}

template <int W, int NP, class Op>
__global__ void __launch_bounds__(kBlock) k_sweep(Op op, uint32_t n, int rev, double* parts_out)
{
	__shared__ double sh[kWaves];
	op.prologue(sh);
	double acc[NP > 0 ? NP : 1] = {0};
	sweep<W, NP>(n, rev != 0, op, acc);
	#pragma unroll
	for (int j = 0; j < NP; j++) {
		double t = block_sum(acc[j], sh);
		if (threadIdx.x == 0) parts_out[j * kMaxGrid + blockIdx.x] = t;
	}
}

// A pair kernel (three sums: s'y, s's, y'y) that also takes the verdict on its pair: every workgroup stores its partials, makes
// them visible device-wide (__threadfence: write-back of this XCD's L2) and takes a ticket; the workgroup that draws the last
// ticket re-reads ALL partials (behind a fence of its own: the other XCDs' lines are not in its L2) and does exactly what
// k_verdict does -- the same total_of over the same array, so the same bits -- then puts the ticket counter back to zero.
template <int W, class Op>
__global__ void __launch_bounds__(kBlock) k_sweep_verdict(Op op, uint32_t n, int rev, double* parts_out, unsigned* ticket, VerdictArgs v)
{
	__shared__ double sh[kWaves];
	__shared__ unsigned drawn;
	op.prologue(sh);
	double acc[3] = {0, 0, 0};
	sweep<W, 3>(n, rev != 0, op, acc);
	#pragma unroll
	for (int j = 0; j < 3; j++) {
		double t = block_sum(acc[j], sh);
		if (threadIdx.x == 0) parts_out[j * kMaxGrid + blockIdx.x] = t;
	}
	if (threadIdx.x == 0) {
		__threadfence();
		drawn = atomicAdd(ticket, 1u);
	}
	__syncthreads();
	if (drawn != gridDim.x - 1) return;
	__threadfence();
	const double sy = total_of(parts_out, (int) gridDim.x, sh);
	const double ss = total_of(parts_out + kMaxGrid, (int) gridDim.x, sh);
	const double yy = total_of(parts_out + 2 * kMaxGrid, (int) gridDim.x, sh);
	if (threadIdx.x != 0) return;
	const bool rejected = v.min_curvature > 0 && sy / ss <= v.min_curvature;      // NaN curvature is accepted (k_verdict)
	if (!rejected) { *v.sy_dst = sy; *v.yy_dst = yy; }
	v.out[0] = sy; v.out[1] = ss; v.out[2] = yy; v.out[3] = rejected ? 1.0 : 0.0;
	*ticket = 0;
}

// ------------------------------------------------------------------------------------------------
// first sweep (take_step prologue; reference src/stochqn.c:808-818, 996, 1174 + first dot of :677)
// ------------------------------------------------------------------------------------------------
template <bool NT> struct FirstOp {
	FirstArgs a;
	bool rms;
	double w_old, w_new;
	template <int W> struct In { Pack<W> q, s, G; };

	__device__ void prologue(double*) {}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		In<W> in;
		in.q = ld<W, false>(a.q, i);
		if (a.s_newest) in.s = ld<W, NT>(a.s_newest, i);
		if (a.G) in.G = ld<W, false>(a.G, i);
		return in;
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& in, double (&acc)[2]) const
	{
		if (a.gprev_out) st<W>(a.gprev_out, i, in.q);
		if (a.frow_out) st_nt<W>(a.frow_out, i, in.q);
		Pack<W> dir = in.q;
		if (a.G) {
			Pack<W> Gn;
			#pragma unroll
			for (int k = 0; k < W; k++) {
				const double g = in.q.v[k];
				// reference src/stochqn.c:738 / :745, rounded as written (no contraction)
				Gn.v[k] = rms ? (w_old * in.G.v[k] + w_new * (g * g)) : (in.G.v[k] + g * g);
				dir.v[k] = g / sqrt(Gn.v[k] + a.scal_reg);               // :778 / :781
			}
			st<W>(a.G, i, Gn);
			st<W>(a.H0_out ? a.H0_out : a.q, i, dir);
		}
		#pragma unroll
		for (int k = 0; k < W; k++) {
			if (a.s_newest) {
				acc[0] = fma(in.s.v[k], in.q.v[k], acc[0]);              // s_{k-1}' q on the RAW gradient
			} else {
				acc[0] = fma(dir.v[k], dir.v[k], acc[0]);                 // guard: sum dir^2
				acc[1] += (isfinite(dir.v[k]) ? 0.0 : 1.0);               //        #nonfinite
			}
		}
	}
};

// ------------------------------------------------------------------------------------------------
// backward sweep (reference src/stochqn.c:676-678, fused with the next iteration's :677)
// ------------------------------------------------------------------------------------------------
template <bool NT> struct BwdOp {
	Partials in;
	const double* sy_row;     // s'y of pair i  (rho_i = 1 / s'y, :676)
	double* alpha_out;        // &alpha[i]
	double* rho_out;          // &rho[i]
	const real* y;
	real* q;
	const real* s_prev;
	double alpha;
	template <int W> struct In { Pack<W> y, q, s; };

	__device__ void prologue(double* sh)
	{
		const double p = total_of(in.parts, in.count, sh);
		const double rho = 1.0 / *sy_row;
		alpha = rho * p;
		if (blockIdx.x == 0 && threadIdx.x == 0) { *alpha_out = alpha; *rho_out = rho; }
	}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		return In<W>{ld<W, NT>(y, i), ld<W, false>(q, i), ld<W, NT>(s_prev, i)};
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& in_, double (&acc)[1]) const
	{
		Pack<W> o;
		#pragma unroll
		for (int k = 0; k < W; k++) {
			o.v[k] = fma(-alpha, in_.y.v[k], in_.q.v[k]);
			acc[0] = fma(in_.s.v[k], o.v[k], acc[0]);
		}
		st<W>(q, i, o);
	}
};

// ------------------------------------------------------------------------------------------------
// middle sweep: last backward update + initial scaling (:683-699) + first forward dot (:705)
// ------------------------------------------------------------------------------------------------
template <bool NT> struct MidOp {
	Partials in;
	const double* sy_row;
	double* alpha_out;
	double* rho_out;
	const real* y;
	real* q;
	MidScale ms;
	double alpha, scal;
	template <int W> struct In { Pack<W> y, q, h; };

	__device__ void prologue(double* sh)
	{
		const double p = total_of(in.parts, in.count, sh);
		const double rho = 1.0 / *sy_row;
		alpha = rho * p;
		scal = (ms.sy_newest != nullptr) ? (*ms.sy_newest / *ms.yy_newest) : ms.h0;
		if (blockIdx.x == 0 && threadIdx.x == 0) { *alpha_out = alpha; *rho_out = rho; }
	}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		In<W> r;
		r.y = ld<W, NT>(y, i);
		r.q = ld<W, false>(q, i);
		if (ms.H0) r.h = ld<W, false>(ms.H0, i);
		return r;
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& in_, double (&acc)[1]) const
	{
		Pack<W> o;
		#pragma unroll
		for (int k = 0; k < W; k++) {
			const double qn = fma(-alpha, in_.y.v[k], in_.q.v[k]);
			o.v[k] = ms.H0 ? qn * in_.h.v[k] : scal * qn;
			acc[0] = fma(in_.y.v[k], o.v[k], acc[0]);
		}
		st<W>(q, i, o);
	}
};

// ------------------------------------------------------------------------------------------------
// forward sweeps (reference src/stochqn.c:705-706, fused with the next iteration's :705)
// ------------------------------------------------------------------------------------------------
template <bool NT, bool LAST, bool FUSE> struct FwdOp {
	Partials in;
	const double* sy_row;
	const double* alpha_i;
	const real* s;
	real* r;
	const real* y_next;   // !LAST
	ApplyArgs ap;           // FUSE
	double coef;
	template <int W> struct In { Pack<W> s, r, y, x, xs; };

	__device__ void prologue(double* sh)
	{
		const double p = total_of(in.parts, in.count, sh);
		const double beta = (1.0 / *sy_row) * p;
		coef = *alpha_i - beta;
	}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		In<W> v;
		v.s = ld<W, NT>(s, i);
		v.r = ld<W, false>(r, i);
		if constexpr (!LAST) v.y = ld<W, NT>(y_next, i);
		if constexpr (FUSE) {
			v.x = ld<W, false>(ap.x, i);
			if (ap.x_sum) v.xs = ld<W, false>(ap.x_sum, i);
		}
		return v;
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& v, double (&acc)[2]) const
	{
		Pack<W> o;
		#pragma unroll
		for (int k = 0; k < W; k++) {
			o.v[k] = fma(coef, v.s.v[k], v.r.v[k]);
			if constexpr (!LAST) acc[0] = fma(v.y.v[k], o.v[k], acc[0]);
			else if constexpr (!FUSE) { acc[0] = fma(o.v[k], o.v[k], acc[0]); acc[1] += (isfinite(o.v[k]) ? 0.0 : 1.0); }
		}
		if constexpr (FUSE) {
			Pack<W> xn, sg;
			#pragma unroll
			for (int k = 0; k < W; k++) {
				xn.v[k] = fma(-ap.step, o.v[k], v.x.v[k]);                // :838
				sg.v[k] = (-ap.step) * o.v[k];                            // :1006
			}
			st<W>(ap.x, i, xn);
			if (ap.x_sum) {
				Pack<W> t;
				#pragma unroll
				for (int k = 0; k < W; k++) t.v[k] = v.xs.v[k] + (double) (real) xn.v[k];  // :283, x as stored
				st<W>(ap.x_sum, i, t);
			}
			if (ap.s_slot) { st_nt<W>(ap.s_slot, i, sg); st<W>(r, i, sg); }
			else st<W>(r, i, o);
		} else {
			st<W>(r, i, o);
		}
	}
};

// ------------------------------------------------------------------------------------------------
// guarded position update (reference src/stochqn.c:825-838, 1006-1007, 1067/1191)
// ------------------------------------------------------------------------------------------------
struct ApplyOp {
	Partials guard;
	bool guarded;
	double n_global;
	const real* r;
	real* grad_out;   // oLBFGS: -step*r is written back here (and to s_slot)
	ApplyArgs ap;
	double* report;
	bool bad;
	template <int W> struct In { Pack<W> r, x, xs; };

	__device__ void prologue(double* sh)
	{
		bad = false;
		if (guarded) {
			const double ss = total_of(guard.parts, guard.count, sh);
			const double nf = total_of(guard.parts + guard.stride, guard.count, sh);
			bad = (nf > 0.0) || !(sqrt(ss) <= 1e3 * n_global);
			if (blockIdx.x == 0 && threadIdx.x == 0) { report[0] = bad ? 1.0 : 0.0; report[1] = ss; report[2] = nf; }
		} else if (blockIdx.x == 0 && threadIdx.x == 0) {
			report[0] = 0.0; report[1] = 0.0; report[2] = 0.0;
		}
	}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		In<W> v;
		if (!bad) v.r = ld<W, false>(r, i);
		v.x = ld<W, false>(ap.x, i);
		if (ap.x_sum) v.xs = ld<W, false>(ap.x_sum, i);
		return v;
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& v, double (&)[1]) const
	{
		Pack<W> xn = v.x;
		if (!bad) {
			Pack<W> sg;
			#pragma unroll
			for (int k = 0; k < W; k++) {
				xn.v[k] = fma(-ap.step, v.r.v[k], v.x.v[k]);
				sg.v[k] = (-ap.step) * v.r.v[k];
			}
			st<W>(ap.x, i, xn);
			if (ap.s_slot) { st_nt<W>(ap.s_slot, i, sg); st<W>(grad_out, i, sg); }
		}
		if (ap.x_sum) {
			Pack<W> t;
			#pragma unroll
			for (int k = 0; k < W; k++) t.v[k] = v.xs.v[k] + (double) (real) xn.v[k];      // x as stored (:283 reads the array back)
			st<W>(ap.x_sum, i, t);
		}
	}
};

// x as the guarded update WILL write it if the guard passes (same expression as ApplyOp, so the same bits), into a vector of
// its own: the source of the slices of x that a host caller is sent while pass 3 is still running (machines.cpp: enqueue_step).
struct SpecXOp {
	const real* r;
	const real* x;
	real* out;
	double step;
	template <int W> struct In { Pack<W> r, x; };
	__device__ void prologue(double*) {}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		In<W> v;
		v.r = ld<W, false>(r, i);
		v.x = ld<W, false>(x, i);
		return v;
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& v, double (&)[1]) const
	{
		Pack<W> xn;
		#pragma unroll
		for (int k = 0; k < W; k++) xn.v[k] = fma(-step, v.r.v[k], v.x.v[k]);
		st<W>(out, i, xn);
	}
};

// ------------------------------------------------------------------------------------------------
// correction pairs
// ------------------------------------------------------------------------------------------------
// s = x_avg - x_avg_prev, with x_avg = x_sum * (1/L) written back first (:286-291, :861-870)
struct PairSOp {
	real* x_sum;
	double inv_L;
	bool scale;
	const real* x_avg_prev;
	real* s_out;
	template <int W> struct In { Pack<W> xs, xp; };
	__device__ void prologue(double*) {}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		return In<W>{ld<W, false>(x_sum, i), ld<W, false>(x_avg_prev, i)};
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& v, double (&)[1]) const
	{
		Pack<W> avg = v.xs, s;
		#pragma unroll
		for (int k = 0; k < W; k++) {
			// the average passes through the array x_sum in the reference (dscal, then the difference reads it back: :286-291,
			// :861-870): in the float build it is rounded to float on the way, and s = x_avg - x_avg_prev, which cancels to
			// ~1e-3 of its operands, follows that rounding
			if (scale) avg.v[k] = (double) (real) (v.xs.v[k] * inv_L);
			s.v[k] = avg.v[k] - v.xp.v[k];
		}
		if (scale) st<W>(x_sum, i, avg);
		st<W>(s_out, i, s);
	}
};

__device__ __forceinline__ void three_dots(double s, double y, double (&acc)[3])
{
	acc[0] = fma(s, y, acc[0]);
	acc[1] = fma(s, s, acc[1]);
	acc[2] = fma(y, y, acc[2]);
}

// y = g - g_prev (+ lambda s) ; s'y, s's, y'y   (:915-923 + the dots of :892 and of :676,:686-687)
struct PairYDiffOp {
	const real* g;
	const real* g_prev;
	const real* s;
	double lambda;
	real* y_out;
	template <int W> struct In { Pack<W> g, gp, s; };
	__device__ void prologue(double*) {}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		return In<W>{ld<W, false>(g, i), ld<W, false>(g_prev, i), ld<W, false>(s, i)};
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& v, double (&acc)[3]) const
	{
		Pack<W> y;
		#pragma unroll
		for (int k = 0; k < W; k++) {
			double d = v.g.v[k] - v.gp.v[k];
			if (lambda > 0) d = fma(lambda, v.s.v[k], d);
			y.v[k] = d;
			three_dots(v.s.v[k], d, acc);
		}
		st<W>(y_out, i, y);
	}
};

// y = hess_vec ; x_avg_prev <- x_avg ; x_sum <- 0 ; dots   (:1139-1140, :962-966)
struct PairYHvOp {
	const real* hv;
	const real* s;
	real* y_out;
	real* x_sum;        // nullable
	real* x_avg_prev;
	template <int W> struct In { Pack<W> hv, s, xs; };
	__device__ void prologue(double*) {}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		In<W> v;
		v.hv = ld<W, false>(hv, i);
		v.s = ld<W, false>(s, i);
		if (x_sum) v.xs = ld<W, false>(x_sum, i);
		return v;
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& v, double (&acc)[3]) const
	{
		#pragma unroll
		for (int k = 0; k < W; k++) three_dots(v.s.v[k], v.hv.v[k], acc);
		st<W>(y_out, i, v.hv);
		if (x_sum) {
			Pack<W> z;
			#pragma unroll
			for (int k = 0; k < W; k++) z.v[k] = 0.0;
			st<W>(x_avg_prev, i, v.xs);
			st<W>(x_sum, i, z);
		}
	}
};

struct Dots3Op {
	const real* s;
	const real* y;
	template <int W> struct In { Pack<W> s, y; };
	__device__ void prologue(double*) {}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const
	{
		return In<W>{ld<W, false>(s, i), ld<W, false>(y, i)};
	}
	template <int W> __device__ __forceinline__ void apply(uint32_t, const In<W>& v, double (&acc)[3]) const
	{
		#pragma unroll
		for (int k = 0; k < W; k++) three_dots(v.s.v[k], v.y.v[k], acc);
	}
};

struct ScaleOp {
	real* x;
	double a;
	template <int W> struct In { Pack<W> x; };
	__device__ void prologue(double*) {}
	template <int W> __device__ __forceinline__ In<W> load(uint32_t i) const { return In<W>{ld<W, false>(x, i)}; }
	template <int W> __device__ __forceinline__ void apply(uint32_t i, const In<W>& v, double (&)[1]) const
	{
		Pack<W> o;
		#pragma unroll
		for (int k = 0; k < W; k++) o.v[k] = v.x.v[k] * a;
		st<W>(x, i, o);
	}
};

// ------------------------------------------------------------------------------------------------
// empirical Fisher product (reference src/stochqn.c:946-949): t = F s ; y = F' t / fu
// ------------------------------------------------------------------------------------------------
// pass 1: blockIdx.y selects a group of kFisherRows rows; each lane keeps one accumulator per row
// of the group while it strides over its columns, so F is read exactly once and s once per group.
template <int W, bool NT, int kFisherRows>
__global__ void __launch_bounds__(kBlock) k_fisher_t(const real* F, size_t ld_, uint32_t n, uint32_t fu,
                                                     const real* s, double* parts)
{
	__shared__ double sh[kWaves];
	const uint32_t row0 = blockIdx.y * kFisherRows;
	const uint32_t nrows = (fu - row0 < (uint32_t) kFisherRows) ? fu - row0 : (uint32_t) kFisherRows;
	double acc[kFisherRows];
	#pragma unroll
	for (int k = 0; k < kFisherRows; k++) acc[k] = 0;
	const uint32_t packs = n / W;
	const uint32_t stride = gridDim.x * kBlock;
	const real* Fg = F + (size_t) row0 * ld_;
	for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < packs; p += stride) {
		const Pack<W> sv = ld<W, false>(s, p * W);
		RPack<W> f[kFisherRows];
		#pragma unroll
		for (int k = 0; k < kFisherRows; k++)
			if ((uint32_t) k < nrows) f[k] = ldr<W, NT>(Fg + (size_t) k * ld_, p * W);
		#pragma unroll
		for (int k = 0; k < kFisherRows; k++)
			if ((uint32_t) k < nrows) {
				#pragma unroll
				for (int j = 0; j < W; j++) acc[k] = fma((double) f[k].v[j], sv.v[j], acc[k]);
			}
	}
	if (W > 1) {
		const uint32_t i = packs * W + threadIdx.x;
		if (blockIdx.x == gridDim.x - 1 && i < n)
			for (uint32_t k = 0; k < nrows; k++) acc[k] = fma(Fg[(size_t) k * ld_ + i], s[i], acc[k]);
	}
	#pragma unroll
	for (int k = 0; k < kFisherRows; k++) {
		const double t = block_sum(acc[k], sh);
		if (threadIdx.x == 0 && (uint32_t) k < nrows) parts[(size_t) (row0 + k) * kMaxGrid + blockIdx.x] = t;
	}
}

// pass 1, row-split (round 6; default): the NW waves of a workgroup share the SAME columns and divide the rows among them --
// wave w of row block blockIdx.y owns rows [blockIdx.y*NW*RPW + w*RPW, +RPW) -- so the pack of s that belongs to a column tile is
// fetched from HBM once per 128 rows (NW = 8, RPW = 16): the first wave to ask misses, the other seven hit in the CU's L1 or the
// XCD's L2.  k_fisher_t above re-reads s once per group of 16 rows: at fu = 128 eight times, PMC 108.78 GB against 103.2 GB
// algorithmic.  A lane carries only RPW accumulators and RPW row packs in flight; the waves are kept within
// `lag` column tiles of each other (option "fisher_lag", default 8) by a workgroup barrier (the trip count depends on blockIdx.x only, so every wave
// reaches every barrier), which bounds how long a line of s has to survive in L2.  Same row-major gemv as reference
// src/stochqn.c:946 (t = F s), another association of the sums than k_fisher_t (lanes own other columns): parity is held
// against the oracle at north_star's tolerance, not against the other kernel's bits.
// U: column tiles of 64 packs a workgroup takes per trip (option "fisher_tile", default 2): with U = 2 a wave reads 2 KB of every row
// of its share back to back (2 x RPW row packs in flight per lane): 15.67 against 16.19 - 16.28 ms at fu = 128, n = 1e8
// (profiles/r06_c4_ab_fisher_tile.jsonl).
template <int W, int RPW, bool NT, int NW, int U>
__global__ void __launch_bounds__(64 * NW) k_fisher_t_split(const real* F, size_t ld_, uint32_t n, uint32_t fu, const real* s, double* parts, uint32_t lag)
{
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const uint32_t row0 = blockIdx.y * (uint32_t) (NW * RPW) + (uint32_t) (wave * RPW);
	const uint32_t nrows = row0 >= fu ? 0u : (fu - row0 < (uint32_t) RPW ? fu - row0 : (uint32_t) RPW);
	double acc[RPW];
	#pragma unroll
	for (int j = 0; j < RPW; j++) acc[j] = 0;
	const uint32_t packs = n / W, stride = gridDim.x * (uint32_t) (64 * U), first = blockIdx.x * (uint32_t) (64 * U);
	const uint32_t trips = first < packs ? (packs - first + stride - 1) / stride : 0u;      // the same for every wave of the workgroup
	const real* Fg = F + (size_t) row0 * ld_;
	for (uint32_t it = 0; it < trips; it++) {
		const uint32_t p0 = first + it * stride + (uint32_t) lane;
		if (nrows) {
			Pack<W> sv[U];
			RPack<W> f[U][RPW];
			#pragma unroll
			for (int u = 0; u < U; u++) {
				const uint32_t p = p0 + (uint32_t) (64 * u);
				if (p < packs) {
					sv[u] = ld<W, false>(s, p * W);
					#pragma unroll
					for (int j = 0; j < RPW; j++)
						if ((uint32_t) j < nrows) f[u][j] = ldr<W, NT>(Fg + (size_t) j * ld_, p * W);
				}
			}
			#pragma unroll
			for (int u = 0; u < U; u++) {
				const uint32_t p = p0 + (uint32_t) (64 * u);
				if (p < packs) {
					#pragma unroll
					for (int j = 0; j < RPW; j++)
						if ((uint32_t) j < nrows) {
							#pragma unroll
							for (int e = 0; e < W; e++) acc[j] = fma((double) f[u][j].v[e], sv[u].v[e], acc[j]);
						}
				}
			}
		}
		if (lag && (it % lag) == lag - 1) __syncthreads();
	}
	if (W > 1) {
		const uint32_t i = packs * W + (uint32_t) lane;                    // tail elements (n not a multiple of W)
		if (blockIdx.x == gridDim.x - 1 && i < n) {
			const double sv = (double) s[i];
			#pragma unroll
			for (int j = 0; j < RPW; j++)
				if ((uint32_t) j < nrows) acc[j] = fma((double) Fg[(size_t) j * ld_ + i], sv, acc[j]);
		}
	}
	#pragma unroll
	for (int j = 0; j < RPW; j++) {
		const double t = wave_sum(acc[j]);
		if (lane == 0 && (uint32_t) j < nrows) parts[(size_t) (row0 + j) * kMaxGrid + blockIdx.x] = t;
	}
}

// pass 2: y_j = (1/fu) * sum_k t_k F[k][j], rows accumulated in index order; fused with the
// curvature dots of the new pair.  One store stream (y) among fu + 1 read streams: PH = its stores clock-phased like those
// of passes 2 and 3 (Parked).
template <int W, bool NT, bool PH>
__global__ void __launch_bounds__(kBlock) k_fisher_y(const real* F, size_t ld_, uint32_t n, uint32_t fu,
                                                     const double* t, double inv_fu, const real* s, real* y,
                                                     double* parts_out, uint32_t phase_inv)
{
	extern __shared__ double t_sh[];   // fu doubles
	__shared__ double sh[kWaves];
	__shared__ typename Stored<W>::type park[PH ? kParkSlots * kBlock : 1];
	// t passes through a real_t buffer between the two products in the reference (buffer_y, src/stochqn.c:946-949): in the
	// float build it is rounded to float on the way, and y = F't/fu -- which can cancel to 1e-4 of its terms -- follows that rounding
	for (uint32_t k = threadIdx.x; k < fu; k += kBlock) t_sh[k] = (double) (real) t[k];
	__syncthreads();
	double acc[3] = {0, 0, 0};
	const uint32_t packs = n / W;
	const uint32_t stride = gridDim.x * kBlock;
	constexpr int R = 8;
	Parked<W> pk;
	if constexpr (PH) { pk.open(park, phase_inv, stride, false, 0u, 0xFFFFFFFFu); pk.out[0] = y; pk.pol[0] = 1; }
	for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < packs; p += stride) {
		Pack<W> a;
		#pragma unroll
		for (int j = 0; j < W; j++) a.v[j] = 0;
		const Pack<W> sv = ld<W, false>(s, p * W);
		uint32_t k = 0;
		for (; k + R <= fu; k += R) {
			RPack<W> f[R];
			#pragma unroll
			for (int u = 0; u < R; u++) f[u] = ldr<W, NT>(F + (size_t) (k + u) * ld_, p * W);
			#pragma unroll
			for (int u = 0; u < R; u++) {
				#pragma unroll
				for (int j = 0; j < W; j++) a.v[j] = fma((double) f[u].v[j], t_sh[k + u], a.v[j]);
			}
		}
		for (; k < fu; k++) {
			const Pack<W> f = ld<W, NT>(F + (size_t) k * ld_, p * W);
			#pragma unroll
			for (int j = 0; j < W; j++) a.v[j] = fma(f.v[j], t_sh[k], a.v[j]);
		}
		#pragma unroll
		for (int j = 0; j < W; j++) { a.v[j] = inv_fu * a.v[j]; three_dots(sv.v[j], a.v[j], acc); }
		if constexpr (PH) { const Pack<W> one[1] = {a}; pk.put(one, p); }
		else st<W>(y, p * W, a);
	}
	if constexpr (PH) pk.flush();
	if (W > 1) {
		const uint32_t i = packs * W + threadIdx.x;
		if (blockIdx.x == gridDim.x - 1 && i < n) {
			double a = 0;
			for (uint32_t k = 0; k < fu; k++) a = fma(F[(size_t) k * ld_ + i], t_sh[k], a);
			a = inv_fu * a;
			three_dots(s[i], a, acc);
			y[i] = a;
		}
	}
	#pragma unroll
	for (int j = 0; j < 3; j++) {
		const double v = block_sum(acc[j], sh);
		if (threadIdx.x == 0) parts_out[j * kMaxGrid + blockIdx.x] = v;
	}
}

// ------------------------------------------------------------------------------------------------
// rows-dot: pass 1 of the three-pass form (and the rebuild of a column of the cached block)
// ------------------------------------------------------------------------------------------------
// partial[j][workgroup] = sum over this workgroup's packs of rows[j] . probe.  NG groups of 8 rows,
// one accumulator per row and lane; the probe pack is loaded once and reused for every row.
// NPR probes: quantity (pr * rows + j) = rows[j] . probe[pr].  With NPR = 2 the pass that computes
// S g for the recursion also produces the new pair's column s_i'y_new of the cached block.
struct Probes { const real* p[2]; };

// One accumulator per row, probe and lane, NG groups of 8 rows; the compiler hoists the row loads of a
// pack ahead of the arithmetic (up to 512 registers per lane).
// A pass cut into slices of the traversal (host callers: slice s runs as soon as ITS part of the gradient has arrived over
// PCIe, while the rest is still on its way): every lane's accumulators are carried from launch to launch through `carry`,
// the slice boundaries are whole rounds of the grid, so each lane adds exactly the terms it adds in one launch, in the same
// order -- the partials, and with them everything downstream, are bit-identical to the unsliced pass.
struct Slice {
	uint32_t p_begin, p_end;    // packs [p_begin, p_end) of the traversal; p_begin is a multiple of gridDim.x * kBlock
	double* carry;              // [quantity][gridDim.x * kBlock] accumulators between the slices (NULL: one launch)
	int first, last;
};

// SL = false is the pass as one launch (the slice argument is ignored: the device-resident path keeps its register budget --
// 145 VGPRs and three waves per SIMD for the two-probe variant against 203 and two with the carry code in)
// U (round 6): adjacent column tiles a workgroup takes per iteration.  With U = 2 a lane holds the packs p and p + kBlock of every
// row: 8 KB of a row per workgroup and iteration instead of 4, 2 x (k + 1) loads in flight per lane -- the read-only pass 1 then
// streams 3 - 4 % faster (2.48 - 2.53 against 2.59 - 2.69 ms at n = 1e8, k = 20, interleaved; 0.87 of the HBM peak on the bare shape:
// profiles/src/tune17.hip), where the passes that also store gain nothing, nor does the two-probe variant (fewer registers left: 0.2205
// against 0.2003 ms at the C2 shape).  A lane adds its terms in the order of the traversal (tile 0, then tile 1), slices are whole
// rounds of grid x kBlock x U packs, so a sliced pass is still bit-identical to the unsliced one.
template <int W, int NG, bool NT, int NPR, bool SL, int U>
__global__ void __launch_bounds__(kBlock) k_rows_dot_all(RowSet rs, Probes pr, real* copy_out, uint32_t n, int rev,
                                                         double* parts, Slice sl)
{
	__shared__ double sh[NPR * NG * 8 * kWaves];
	double acc[NPR][NG * 8];
	const uint32_t packs = n / W, lanes = gridDim.x * kBlock, stride = lanes * (uint32_t) U, last = packs - 1;
	const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;                      // this lane among the lanes of the grid (carry slot)
	const uint32_t first = blockIdx.x * (uint32_t) (kBlock * U) + threadIdx.x;     // its first pack of a round
	#pragma unroll
	for (int q = 0; q < NPR; q++)
		#pragma unroll
		for (int j = 0; j < NG * 8; j++) {
			if constexpr (SL) acc[q][j] = (!sl.first && j < rs.count) ? sl.carry[(size_t) (q * rs.count + j) * lanes + gtid] : 0.0;
			else acc[q][j] = 0;
		}
	const uint32_t p_end = SL ? sl.p_end : packs;
	for (uint32_t p0 = (SL ? sl.p_begin : 0u) + first; p0 < p_end; p0 += stride) {
		Pack<W> pv[U][NPR];
		RPack<W> f[U][NG * 8];
		#pragma unroll
		for (int t = 0; t < U; t++) {
			const uint32_t p = p0 + (uint32_t) (t * kBlock);
			if (U == 1 || p < p_end) {
				const uint32_t i = (rev ? last - p : p) * W;
				#pragma unroll
				for (int q = 0; q < NPR; q++) pv[t][q] = ld<W, false>(pr.p[q], i);
				if (copy_out) st<W>(copy_out, i, pv[t][0]);
				#pragma unroll
				for (int j = 0; j < NG * 8; j++)
					if (j < rs.count) f[t][j] = ldr<W, NT>(rs.row[j], i);
			}
		}
		#pragma unroll
		for (int t = 0; t < U; t++) {
			const uint32_t p = p0 + (uint32_t) (t * kBlock);
			if (U == 1 || p < p_end) {
				#pragma unroll
				for (int j = 0; j < NG * 8; j++)
					if (j < rs.count) {
						#pragma unroll
						for (int q = 0; q < NPR; q++)
							#pragma unroll
							for (int k = 0; k < W; k++) acc[q][j] = fma((double) f[t][j].v[k], pv[t][q].v[k], acc[q][j]);
					}
			}
		}
	}
	if constexpr (SL) {
		if (!sl.last) {                                       // hand the accumulators to the next slice
			#pragma unroll
			for (int q = 0; q < NPR; q++)
				#pragma unroll
				for (int j = 0; j < NG * 8; j++)
					if (j < rs.count) sl.carry[(size_t) (q * rs.count + j) * lanes + gtid] = acc[q][j];
			return;
		}
	}
	if (W > 1) {
		const uint32_t i = packs * W + threadIdx.x;
		if (blockIdx.x == gridDim.x - 1 && i < n) {
			if (copy_out) copy_out[i] = pr.p[0][i];
			#pragma unroll
			for (int q = 0; q < NPR; q++) {
				const double pv = (double) pr.p[q][i];
				#pragma unroll
				for (int j = 0; j < NG * 8; j++)
					if (j < rs.count) acc[q][j] = fma((double) rs.row[j][i], pv, acc[q][j]);
			}
		}
	}
	// all quantities through the shuffle tree first, then ONE barrier (same tree and wave order as
	// block_sum, so the same bits; 2 barriers per quantity made this epilogue the whole kernel at small n)
	#pragma unroll
	for (int q = 0; q < NPR; q++)
		#pragma unroll
		for (int j = 0; j < NG * 8; j++) {
			if (j < rs.count) {           // uniform
				const double t = wave_sum(acc[q][j]);
				if ((threadIdx.x & 63) == 0) sh[(q * NG * 8 + j) * kWaves + (threadIdx.x >> 6)] = t;
			}
		}
	__syncthreads();
	for (int e = threadIdx.x; e < NPR * rs.count; e += kBlock) {
		const int q = e / rs.count, j = e % rs.count;
		double t = sh[(q * NG * 8 + j) * kWaves];
		#pragma unroll
		for (int w = 1; w < kWaves; w++) t += sh[(q * NG * 8 + j) * kWaves + w];
		parts[(size_t) (q * rs.count + j) * kMaxGrid + blockIdx.x] = t;
	}
}

// Work split (single probe; the float build's pass 1: 2.5 against 5.9 ms for the all-rows form): the NW waves of a
// workgroup share the same columns and divide the ROWS among them (wave w owns rows [w*RPW, (w+1)*RPW)), so a lane
// carries only RPW accumulators and the loads of a whole pack fit in registers (all issued before the first use).
// The probe packs are read by every wave with default-policy loads: one HBM fetch, the rest L2/L1 hits.
template <int W, int RPW, bool NT, int NW>
__global__ void __launch_bounds__(64 * NW) k_rows_dot(RowSet rs, const real* probe, real* copy_out, uint32_t n, int rev, double* parts)
{
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int row0 = wave * RPW;
	double acc[RPW];
	#pragma unroll
	for (int j = 0; j < RPW; j++) acc[j] = 0;
	const uint32_t packs = n / W, stride = gridDim.x * 64, last = packs - 1;
	for (uint32_t p = blockIdx.x * 64 + lane; p < packs; p += stride) {
		const uint32_t i = (rev ? last - p : p) * W;
		RPack<W> f[RPW];
		const Pack<W> pv = ld<W, false>(probe, i);
		#pragma unroll
		for (int j = 0; j < RPW; j++)
			if (row0 + j < rs.count) f[j] = ldr<W, NT>(rs.row[row0 + j], i);
		if (copy_out && wave == 0) st<W>(copy_out, i, pv);
		#pragma unroll
		for (int j = 0; j < RPW; j++)
			if (row0 + j < rs.count) {
				#pragma unroll
				for (int k = 0; k < W; k++) acc[j] = fma((double) f[j].v[k], pv.v[k], acc[j]);
			}
	}
	if (W > 1) {
		const uint32_t i = packs * W + lane;                           // tail elements (n not a multiple of W)
		if (blockIdx.x == gridDim.x - 1 && i < n) {
			if (copy_out && wave == 0) copy_out[i] = probe[i];
			const double pv = (double) probe[i];
			#pragma unroll
			for (int j = 0; j < RPW; j++)
				if (row0 + j < rs.count) acc[j] = fma((double) rs.row[row0 + j][i], pv, acc[j]);
		}
	}
	#pragma unroll
	for (int j = 0; j < RPW; j++) {
		const double t = wave_sum(acc[j]);
		if (lane == 0 && row0 + j < rs.count) parts[(size_t) (row0 + j) * kMaxGrid + blockIdx.x] = t;
	}
}

// three-pass form, rebuild of one column: gsy[rows[i]][col] = total of partial i (= s_i'y_col), i < k
__global__ void __launch_bounds__(kBlock) k_store_column(const double* a, int a_count, int a_stride, CoefArgs ca, int col_row, double* gsy)
{
	__shared__ double sh[kWaves];
	for (int i = 0; i < ca.k; i++) {
		const double t = total_of(a + (size_t) i * a_stride, a_count, sh);
		if (threadIdx.x == 0) gsy[(size_t) ca.rows[i] * ca.m + col_row] = t;
	}
}

// ------------------------------------------------------------------------------------------------
// three-pass form: S is streamed twice, Y ONCE -- (3k+5) n words per two-loop instead of the sweeps' 8k n.
//   pass 1  b_i = s_i'g                      (k rows of S + g; after a new pair: + the probe y_new -> s_i'y_new)
//   coef a  backward recursion:  alpha_i = rho_i (b_i - sum_{j>i} alpha_j s_i'y_j)          (cached s_old'y_new)
//   pass 2  q0 = g - sum_j alpha_j y_j (newest first, element by element exactly as the reference's sweeps);
//           r0 = gamma q0 | h0 q0 | H0 .* q0 (adaQN: with its side effects on the raw gradient);
//           v_i = y_i'r0 for every i -- the k rows of Y are held in registers for both uses; r0 replaces g
//   coef b  forward recursion:   beta_i = rho_i (v_i + sum_{j<i} c_j s_j'y_i),  c_i = alpha_i - beta_i
//   pass 3  r = r0 + sum_j c_j s_j (oldest first, as the sweeps), guard sums
// The only cached inner products are s_a'y_b for pairs a older than b (+ the diagonal): when pair b enters the
// ring as the newest, one extra probe in the next pass 1 yields its column.  No inner products with Y are cached
// (round 1's two-pass form kept y_i'g, y_i'y_j and adaQN's H0-weighted W_ij, and read Y twice; retired in round 4):
// y_i'r0 is a direct dot with the vector it belongs to.  Reference: src/stochqn.c:663-708.
// ------------------------------------------------------------------------------------------------
// The scalar recursions of the three-pass form run inside the prologues of pass 2 and pass 3: every workgroup totals the
// previous pass's partials and runs the O(k^2) recursion itself, instead of waiting for a one-workgroup kernel between the
// passes (round 2 had two: k_coef3a / k_coef3b -- two launches and two dependent kernel boundaries more per step, the same
// bits; retired in round 4).  Workgroup 0 also stores what later kernels / the host read: the new column of the cached
// block, alpha, rho.
struct Fold3 {
	const double* parts;        // previous pass's partials
	int count, stride;
	CoefArgs a;
	int fresh_row;              // pass 2 only: ring row whose column s_i'y_fresh the previous pass produced (quantities k..2k-1), or -1
	double* gsy;
	const double* sy;
	const double* yy;
	double* alpha;              // [k] logical order: written by pass 2, read by pass 3
	double* rho_out;
};

// SY[i*k+j] = s_i'y_j for i <= j in logical order (diagonal from `sy`); a column that pass 1 has only just produced
// comes from `col` (LDS) because workgroup 0's store to the cached block is not visible to the other workgroups yet
__device__ __forceinline__ void fold_load_sy(const Fold3& f, const double* col, double* SY)
{
	const int k = f.a.k;
	for (int e = threadIdx.x; e < k * k; e += kBlock) {
		const int i = e / k, j = e % k;
		double v;
		if (i == j) v = f.sy[f.a.rows[i]];
		else if (col != nullptr && f.a.rows[j] == f.fresh_row) v = col[i];
		else v = f.gsy[(size_t) f.a.rows[i] * f.a.m + f.a.rows[j]];
		SY[e] = v;
	}
}

// pass 2's prologue: totals of pass 1, the new pair's column, the backward recursion.  cf[0] = scale, cf[1 + i] = alpha_i
__device__ __forceinline__ void fold_backward(const Fold3& f, double* SY, double* bS, double* col, double* cf)
{
	const int k = f.a.k, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	if (f.fresh_row >= 0) {
		for (int i = wave; i < k; i += kWaves) {
			const double t = wave_total_of(f.parts + (size_t) (k + i) * f.stride, f.count);      // s_i'y_fresh
			if (lane == 0) {
				col[i] = t;
				if (blockIdx.x == 0) f.gsy[(size_t) f.a.rows[i] * f.a.m + f.fresh_row] = t;
			}
		}
		__syncthreads();
	}
	fold_load_sy(f, f.fresh_row >= 0 ? col : nullptr, SY);
	for (int q = wave; q < k; q += kWaves) {
		const double t = wave_total_of(f.parts + (size_t) q * f.stride, f.count);
		if (lane == 0) bS[q] = t;
	}
	__syncthreads();
	if (wave == 0) {
		const bool mine = lane < k;
		double al = 0;
		for (int i = k - 1; i >= 0; i--) {                       // alpha_i = rho_i s_i'q_{i+1}  (:676-677)
			const double t = (mine && lane > i) ? al * SY[i * k + lane] : 0.0;
			const double sq = bS[i] - wave_sum_all(t);
			const double rho_i = 1.0 / SY[i * k + i];
			if (lane == i) {
				al = rho_i * sq;
				if (blockIdx.x == 0) { f.alpha[i] = al; f.rho_out[i] = rho_i; }
			}
		}
		if (lane == 0) cf[0] = (f.a.h0 > 0) ? f.a.h0 : SY[(k - 1) * k + (k - 1)] / f.yy[f.a.rows[k - 1]];   // :683-689 / :698
		if (mine) cf[1 + lane] = al;
	}
	__syncthreads();
}

// pass 3's prologue: totals of pass 2 (v_i = y_i'r0), the forward recursion.  cf[i] = c_i = alpha_i - beta_i
__device__ __forceinline__ void fold_forward(const Fold3& f, double* SY, double* V, double* cf)
{
	const int k = f.a.k, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	fold_load_sy(f, nullptr, SY);
	for (int q = wave; q < k; q += kWaves) {
		const double t = wave_total_of(f.parts + (size_t) q * f.stride, f.count);
		if (lane == 0) V[q] = t;
	}
	__syncthreads();
	if (wave == 0) {
		const bool mine = lane < k;
		const double al = mine ? f.alpha[lane] : 0.0;
		double c = 0;
		for (int i = 0; i < k; i++) {                            // beta_i = rho_i y_i'r_i, r_i = r0 + sum_{j<i} c_j s_j  (:705-706)
			const double t = (lane < i) ? c * SY[lane * k + i] : 0.0;
			const double yr = V[i] + wave_sum_all(t);
			if (lane == i) c = al - (1.0 / SY[i * k + i]) * yr;
		}
		if (mine) cf[lane] = c;
	}
	__syncthreads();
}

struct DiagArgs {               // how pass 2 scales q0
	const real* H0_in;          // caller-supplied diagonal (isolated two-loop), or NULL
	real* G;                    // adaQN: grad_sum_sq in/out (then H0_out receives g/sqrt(G+eps)), or NULL
	real* H0_out;
	real* frow_out;             // adaQN: Fisher row <- raw gradient (nullable)
	double w_old, w_new, scal_reg;
	bool rms;
};

// PH: the stores of r0 clock-phased (above); otherwise each pack is stored as soon as it is final.
template <int W, int NG, bool NT, int MODE /*0 scalar, 1 given diagonal, 2 adaQN*/, bool PH>
__global__ void __launch_bounds__(kBlock) k_qdot(RowSet ys, Fold3 fo, real* g, DiagArgs dg, uint32_t n, int rev,
                                                 uint32_t keep_from, double* parts, uint32_t phase_inv)
{
	__shared__ double sh[NG * 8 * kWaves];
	__shared__ double cf[1 + kPairsMax3];
	__shared__ typename Stored<W>::type park[PH ? kParkSlots * kBlock : 1];
	const int k = ys.count;
	{
		__shared__ double SY[kPairsMax3 * kPairsMax3], bS[kPairsMax3], col[kPairsMax3];
		fold_backward(fo, SY, bS, col, cf);
	}
	double acc[NG * 8];
	#pragma unroll
	for (int j = 0; j < NG * 8; j++) acc[j] = 0;
	const uint32_t packs = n / W, stride = gridDim.x * kBlock, last = packs - 1;
	constexpr int NS = (PH && MODE == 2) ? 4 : 1;          // adaQN parks G, H0 and the Fisher row next to r0
	Parked<W, NS> pk;
	if constexpr (PH) {
		pk.open(park, phase_inv, stride, rev != 0, last, keep_from);
		pk.out[0] = g; pk.pol[0] = 1;
		if constexpr (MODE == 2) { pk.out[1] = dg.G; pk.pol[1] = 1; pk.out[2] = dg.H0_out; pk.pol[2] = 1; pk.out[3] = dg.frow_out; pk.pol[3] = 1; }
	}
	for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < packs; p += stride) {
		const uint32_t i = (rev ? last - p : p) * W;
		const Pack<W> gv = ld<W, false>(g, i);
		RPack<W> f[NG * 8];
		#pragma unroll
		for (int j = 0; j < NG * 8; j++)
			if (j < k) f[j] = ldr<W, NT>(ys.row[j], i);
		Pack<W> h, Gn;
		if constexpr (MODE == 1) h = ld<W, false>(dg.H0_in, i);
		if constexpr (MODE == 2) {
			const Pack<W> Gv = ld<W, false>(dg.G, i);
			#pragma unroll
			for (int e = 0; e < W; e++) {
				const double x = gv.v[e];
				Gn.v[e] = dg.rms ? (dg.w_old * Gv.v[e] + dg.w_new * (x * x)) : (Gv.v[e] + x * x);     // reference :738 / :745
				h.v[e] = x / sqrt(Gn.v[e] + dg.scal_reg);                                            // :781
			}
			if constexpr (!PH) {
				st<W>(dg.G, i, Gn);
				st<W>(dg.H0_out, i, h);
				if (dg.frow_out) st_nt<W>(dg.frow_out, i, gv);
			}
		}
		Pack<W> q = gv;
		#pragma unroll
		for (int j = NG * 8 - 1; j >= 0; j--)                     // newest pair first (:671-679)
			if (j < k) {
				#pragma unroll
				for (int e = 0; e < W; e++) q.v[e] = fma(-cf[1 + j], (double) f[j].v[e], q.v[e]);
			}
		#pragma unroll
		for (int e = 0; e < W; e++) q.v[e] = (MODE == 0) ? cf[0] * q.v[e] : q.v[e] * h.v[e];      // :688 / :698 / :695
		#pragma unroll
		for (int j = 0; j < NG * 8; j++)
			if (j < k) {
				#pragma unroll
				for (int e = 0; e < W; e++) {
					if constexpr (W > 2) {
						// float build: the row value is widened AGAIN for its second use.  Left to itself the compiler keeps all 4 x k doubles
						// of the chain above alive (256 VGPRs + 126 AGPRs of copies: pass 2 at 0.69 of peak where pass 3 runs at 0.83)
						real fv = f[j].v[e];
						asm volatile("" : "+v"(fv));
						acc[j] = fma((double) fv, q.v[e], acc[j]);
					} else acc[j] = fma((double) f[j].v[e], q.v[e], acc[j]);
				}
			}
		// the part of r0 this pass writes last is what pass 3 (opposite direction) reads first: those packs are stored with
		// the default policy so that they may still sit in the Infinity Cache, the rest streams past it (keep_from)
		if constexpr (PH) {
			if constexpr (MODE == 2) { const Pack<W> four[4] = {q, Gn, h, gv}; pk.put(four, p); }
			else { const Pack<W> one[1] = {q}; pk.put(one, p); }
		}
		else { if (p >= keep_from) st<W>(g, i, q); else st_stream<W>(g, i, q); }
	}
	if constexpr (PH) pk.flush();
	if (W > 1) {
		const uint32_t i = packs * W + threadIdx.x;
		if (blockIdx.x == gridDim.x - 1 && i < n) {
			const double x = (double) g[i];
			double h = 0;
			if (MODE == 1) h = (double) dg.H0_in[i];
			if (MODE == 2) {
				const double Gv = (double) dg.G[i];
				const double Gn = dg.rms ? (dg.w_old * Gv + dg.w_new * (x * x)) : (Gv + x * x);
				h = x / sqrt(Gn + dg.scal_reg);
				dg.G[i] = (real) Gn;
				dg.H0_out[i] = (real) h;
				if (dg.frow_out) dg.frow_out[i] = (real) x;
			}
			double q = x;
			for (int j = k - 1; j >= 0; j--) q = fma(-cf[1 + j], (double) ys.row[j][i], q);
			q = (MODE == 0) ? cf[0] * q : q * h;
			#pragma unroll
			for (int j = 0; j < NG * 8; j++)
				if (j < k) acc[j] = fma((double) ys.row[j][i], q, acc[j]);
			g[i] = (real) q;
		}
	}
	#pragma unroll
	for (int j = 0; j < NG * 8; j++)
		if (j < k) {
			const double t = wave_sum(acc[j]);
			if ((threadIdx.x & 63) == 0) sh[j * kWaves + (threadIdx.x >> 6)] = t;
		}
	__syncthreads();
	for (int j = threadIdx.x; j < k; j += kBlock) {
		double t = sh[j * kWaves];
		#pragma unroll
		for (int w = 1; w < kWaves; w++) t += sh[j * kWaves + w];
		parts[(size_t) j * kMaxGrid + blockIdx.x] = t;
	}
}

// pass 3: r = r0 + sum_j c_j s_j, oldest pair first (:702-707); guard sums (sum r^2, #non-finite).  The k rows of a pack sit
// in registers like pass 2's (all their loads issued before the first fma); PH: the stores of r clock-phased like pass 2's.
// (Folding the position update into this pass for check_nan == 0 was built in round 3 and measured slower -- 3.68 ms against
// 2.79 + 0.73: two more store streams among 21 read streams -- and removed in round 4; profiles/r03_ab_fuse_apply.jsonl.)
// SL: the pass in slices of the traversal (whole rounds of the grid), the two guard sums carried from slice to slice
// per lane exactly like the accumulators of pass 1 (k_rows_dot_all): same terms, same order, same bits as one launch.
template <int W, int NG, bool NT, bool PH, bool SL>
__global__ void __launch_bounds__(kBlock) k_sadd(RowSet ss, Fold3 fo, real* r, uint32_t n, int rev, uint32_t keep_from, double* parts, Slice sl,
                                                 uint32_t phase_inv)
{
	__shared__ double sh[kWaves];
	__shared__ double cf[kPairsMax3];
	__shared__ typename Stored<W>::type park[PH ? kParkSlots * kBlock : 1];
	const int k = ss.count;
	{
		__shared__ double SY[kPairsMax3 * kPairsMax3], V[kPairsMax3];
		fold_forward(fo, SY, V, cf);
	}
	double acc0 = 0, acc1 = 0;
	const uint32_t packs = n / W, stride = gridDim.x * kBlock, last = packs - 1;
	const uint32_t gtid = blockIdx.x * kBlock + threadIdx.x;
	if constexpr (SL) {
		if (!sl.first) { acc0 = sl.carry[gtid]; acc1 = sl.carry[(size_t) stride + gtid]; }
	}
	const uint32_t p_end = SL ? sl.p_end : packs;
	Parked<W> pk;
	if constexpr (PH) { pk.open(park, phase_inv, stride, rev != 0, last, keep_from); pk.out[0] = r; pk.pol[0] = 1; }
	for (uint32_t p = (SL ? sl.p_begin : 0u) + gtid; p < p_end; p += stride) {
		const uint32_t i = (rev ? last - p : p) * W;
		Pack<W> v = ld<W, false>(r, i);
		RPack<W> f[NG * 8];
		#pragma unroll
		for (int j = 0; j < NG * 8; j++)
			if (j < k) f[j] = ldr<W, NT>(ss.row[j], i);
		#pragma unroll
		for (int j = 0; j < NG * 8; j++)
			if (j < k) {
				#pragma unroll
				for (int e = 0; e < W; e++) v.v[e] = fma(cf[j], (double) f[j].v[e], v.v[e]);
			}
		#pragma unroll
		for (int e = 0; e < W; e++) { acc0 = fma(v.v[e], v.v[e], acc0); acc1 += (isfinite(v.v[e]) ? 0.0 : 1.0); }
		// the tail of r stays cacheable for the apply pass (see k_qdot)
		if constexpr (PH) { const Pack<W> one[1] = {v}; pk.put(one, p); }
		else { if (p >= keep_from) st<W>(r, i, v); else st_stream<W>(r, i, v); }
	}
	if constexpr (PH) pk.flush();
	if constexpr (SL) {
		if (!sl.last) { sl.carry[gtid] = acc0; sl.carry[(size_t) stride + gtid] = acc1; return; }
	}
	if (W > 1) {
		const uint32_t i = packs * W + threadIdx.x;
		if (blockIdx.x == gridDim.x - 1 && i < n) {
			double v = (double) r[i];
			for (int j = 0; j < k; j++) v = fma(cf[j], (double) ss.row[j][i], v);
			acc0 = fma(v, v, acc0); acc1 += (isfinite(v) ? 0.0 : 1.0);
			r[i] = (real) v;
		}
	}
	const double t0 = block_sum(acc0, sh), t1 = block_sum(acc1, sh);
	if (threadIdx.x == 0) { parts[blockIdx.x] = t0; parts[kMaxGrid + blockIdx.x] = t1; }
}

// out[j] = sum of partial array j (one workgroup per quantity)
__global__ void __launch_bounds__(kBlock) k_fin(const double* parts, int count, int stride, double* out)
{
	__shared__ double sh[kWaves];
	const double t = total_of(parts + (size_t) blockIdx.x * stride, count, sh);
	if (threadIdx.x == 0) out[blockIdx.x] = t;
}

// sy_dst = sum of partial array 0 (s'y), yy_dst = sum of partial array 2 (y'y)
__global__ void __launch_bounds__(kBlock) k_commit(const double* parts, int count, int stride, double* sy_dst, double* yy_dst)
{
	__shared__ double sh[kWaves];
	const double a = total_of(parts, count, sh);
	const double b = total_of(parts + 2 * (size_t) stride, count, sh);
	if (threadIdx.x == 0) { *sy_dst = a; *yy_dst = b; }
}

// check_min_curvature on the device (reference src/stochqn.c:883-900): totals of the pair's three
// dots (s'y, s's, y'y), the accept / reject decision, the commit of s'y and y'y for an accepted pair.
// out = { s'y, s's, y'y, rejected ? 1 : 0 } for the host's bookkeeping.
__global__ void __launch_bounds__(kBlock) k_verdict(const double* parts, int count, int stride, double min_curvature,
                                                    double* sy_dst, double* yy_dst, double* out)
{
	__shared__ double sh[kWaves];
	const double sy = total_of(parts, count, sh);
	const double ss = total_of(parts + (size_t) stride, count, sh);
	const double yy = total_of(parts + 2 * (size_t) stride, count, sh);
	if (threadIdx.x != 0) return;
	const bool rejected = min_curvature > 0 && sy / ss <= min_curvature;      // NaN curvature is accepted
	if (!rejected) { *sy_dst = sy; *yy_dst = yy; }
	out[0] = sy; out[1] = ss; out[2] = yy; out[3] = rejected ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------------
// synthetic inputs for measurement: a counter-based generator, so that a shard
// [first, first + count) of a vector is the same numbers whoever generates it and however n is split.
//   u(i, stream, t) = top 53 bits of splitmix64-finalise(key + i * golden) / 2^53,
//   key = seed ^ stream * 0x9E3779B97F4A7C15 ^ t * 0xD1B54A32D192ED03            (host: synth_key)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double synth_u(uint64_t key, uint64_t i)
{
	uint64_t z = key + i * 0x9E3779B97F4A7C15ull;
	z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
	z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
	z = z ^ (z >> 31);
	return (double) (z >> 11) * (1.0 / 9007199254740992.0);
}

// out_j = a + b * u(first + j)
__global__ void __launch_bounds__(kBlock) k_synth_uniform(real* out, size_t count, uint64_t first, uint64_t key, double a, double b)
{
	for (size_t j = blockIdx.x * (size_t) kBlock + threadIdx.x; j < count; j += (size_t) gridDim.x * kBlock)
		out[j] = (real) (a + b * synth_u(key, first + j));
}

// g_j = d_j x_j (1 + amp (2 u(first + j) - 1)): the noisy gradient of f = 1/2 sum d x^2 (3 n words), one 16-byte
// pack per lane and step like the sweeps
__global__ void __launch_bounds__(kBlock) k_synth_grad(real* g, const real* d, const real* x, size_t count, uint64_t first,
                                                      uint64_t key, double amp)
{
	const size_t packs = count / kVec;
	for (size_t p = blockIdx.x * (size_t) kBlock + threadIdx.x; p < packs; p += (size_t) gridDim.x * kBlock) {
		const size_t j = p * kVec;
		const rvec dv = *reinterpret_cast<const rvec_u*>(d + j), xv = *reinterpret_cast<const rvec_u*>(x + j);
		rvec gv;
		#pragma unroll
		for (int k = 0; k < kVec; k++) {
			const double noise = 1.0 + amp * (2.0 * synth_u(key, first + j + k) - 1.0);
			gv[k] = (real) (((double) dv[k] * (double) xv[k]) * noise);
		}
		*reinterpret_cast<rvec_u*>(g + j) = gv;
	}
	const size_t j = packs * kVec + threadIdx.x;                  // tail (count not a multiple of the pack)
	if (blockIdx.x == 0 && j < count) {
		const double noise = 1.0 + amp * (2.0 * synth_u(key, first + j) - 1.0);
		g[j] = (real) (((double) d[j] * (double) x[j]) * noise);
	}
}

// One sample row of the Hessian mini-batch with disjoint supports: a_j = sqrt(bs d_j) where
// (first + j) mod bs == k, else 0 -- A'A/bs = diag(d) exactly (bench.py)
__global__ void __launch_bounds__(kBlock) k_synth_batch_row(real* row, const real* d, size_t count, uint64_t first, uint32_t k, uint32_t bs)
{
	for (size_t j = blockIdx.x * (size_t) kBlock + threadIdx.x; j < count; j += (size_t) gridDim.x * kBlock)
		row[j] = ((first + j) % bs == k) ? (real) sqrt((double) bs * (double) d[j]) : (real) 0;
}

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// The vector path needs element alignment only (see rvec_u); a pointer that is not even that goes
// through the element-wise instantiation.
inline bool elem_aligned(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % sizeof(real)) == 0; }

template <class... P> inline bool all_aligned(P... ps) { return (elem_aligned(ps) && ...); }

struct ProfScope {
	const Scratch& sc;
	size_t which = 0;
	ProfScope(const Scratch& s, int id) : sc(s) { if (sc.prof) which = sc.prof->begin(id, sc.stream); }
	~ProfScope() { if (sc.prof) sc.prof->end(which, sc.stream); }
};

template <int NP, class Op>
void run_sweep(const Scratch& sc, int id, size_t n, bool vec, const Op& op, double* parts_out, int grid)
{
	ProfScope ps(sc, id);
	const int rev = (sc.reverse && sc.phase) ? ((*sc.phase)++ & 1) : 0;
	if (vec) hipLaunchKernelGGL((k_sweep<kVec, NP, Op>), dim3(grid), dim3(kBlock), 0, sc.stream, op, (uint32_t) n, rev, parts_out);
	else     hipLaunchKernelGGL((k_sweep<1, NP, Op>), dim3(grid), dim3(kBlock), 0, sc.stream, op, (uint32_t) n, rev, parts_out);
}

template <class Op>
void run_sweep_verdict(const Scratch& sc, int id, size_t n, bool vec, const Op& op, double* parts_out, int grid, const VerdictArgs& v)
{
	ProfScope ps(sc, id);
	const int rev = (sc.reverse && sc.phase) ? ((*sc.phase)++ & 1) : 0;
	if (vec) hipLaunchKernelGGL((k_sweep_verdict<kVec, Op>), dim3(grid), dim3(kBlock), 0, sc.stream, op, (uint32_t) n, rev, parts_out, sc.ticket, v);
	else     hipLaunchKernelGGL((k_sweep_verdict<1, Op>), dim3(grid), dim3(kBlock), 0, sc.stream, op, (uint32_t) n, rev, parts_out, sc.ticket, v);
}

// What the consumer of buffer `buf` has to read: the raw partials, or (multi-GPU) the summed scalars.
Partials finish(const Scratch& sc, int buf, int nsums, int grid)
{
	Partials raw{sc.part[buf], grid, kMaxGrid};
	if (!sc.allreduce) return raw;
	launch_fin(sc, raw, nsums, sc.red[buf]);
	sc.allreduce(sc.user, sc.red[buf], nsums, sc.stream);
	return Partials{sc.red[buf], 1, 1};
}

}  // namespace

const char* kernel_name(int id)
{
	static const char* names[K_COUNT] = {
		"first", "bwd", "mid", "fwd", "fwd_last", "apply", "pair_s", "pair_y_diff", "pair_y_hv",
		"dots3", "fisher_t", "fisher_y", "fin", "small", "copy", "sdot", "sdot2", "qdot", "sadd", "xhash"};
	return (id >= 0 && id < K_COUNT) ? names[id] : "?";
}

int sweep_grid(const Scratch& sc, size_t n, int per_cu)
{
	// one pack pair per lane and unroll step; never more workgroups than the partial stride.
	// `per_cu`: kernels with a large share of stores (apply) or a long per-pack dependency chain
	// (combine) measured best with two workgroups per CU, the 3-read/1-write sweeps with one.
	size_t per_block = (size_t) kBlock * 2 * kUnroll;
	size_t g = (n + per_block - 1) / per_block;
	size_t cap = (size_t) sc.grid_cap * per_cu;
	if (cap > (size_t) kMaxGrid) cap = kMaxGrid;
	if (g < 1) g = 1;
	if (g > cap) g = cap;
	return (int) g;
}

Partials launch_first(const Scratch& sc, int buf, size_t n, const FirstArgs& a)
{
	const int grid = sweep_grid(sc, n);
	const bool vec = all_aligned(a.q, a.s_newest, a.gprev_out, a.frow_out, a.G, a.H0_out);
	const bool rms = a.rmsprop_weight > 0 && a.rmsprop_weight < 1;
	if (sc.nontemporal) run_sweep<2>(sc, K_FIRST, n, vec, FirstOp<true>{a, rms, a.rmsprop_weight, 1 - a.rmsprop_weight}, sc.part[buf], grid);
	else                run_sweep<2>(sc, K_FIRST, n, vec, FirstOp<false>{a, rms, a.rmsprop_weight, 1 - a.rmsprop_weight}, sc.part[buf], grid);
	return finish(sc, buf, a.s_newest ? 1 : 2, grid);
}

Partials launch_bwd(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int i,
                    const real* y_i, real* q, const real* s_prev)
{
	const int grid = sweep_grid(sc, n);
	const bool vec = all_aligned(y_i, q, s_prev);
	if (sc.nontemporal) run_sweep<1>(sc, K_BWD, n, vec, BwdOp<true>{in, sy_row, sc.alpha + i, sc.rho + i, y_i, q, s_prev, 0}, sc.part[buf], grid);
	else                run_sweep<1>(sc, K_BWD, n, vec, BwdOp<false>{in, sy_row, sc.alpha + i, sc.rho + i, y_i, q, s_prev, 0}, sc.part[buf], grid);
	return finish(sc, buf, 1, grid);
}

Partials launch_mid(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, const real* y_0,
                    real* q, const MidScale& ms)
{
	const int grid = sweep_grid(sc, n);
	const bool vec = all_aligned(y_0, q, ms.H0);
	if (sc.nontemporal) run_sweep<1>(sc, K_MID, n, vec, MidOp<true>{in, sy_row, sc.alpha, sc.rho, y_0, q, ms, 0, 0}, sc.part[buf], grid);
	else                run_sweep<1>(sc, K_MID, n, vec, MidOp<false>{in, sy_row, sc.alpha, sc.rho, y_0, q, ms, 0, 0}, sc.part[buf], grid);
	return finish(sc, buf, 1, grid);
}

Partials launch_fwd(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int i,
                    const real* s_i, real* r, const real* y_next)
{
	const int grid = sweep_grid(sc, n);
	const bool vec = all_aligned(s_i, r, y_next);
	ApplyArgs none{};
	if (sc.nontemporal) run_sweep<2>(sc, K_FWD, n, vec, FwdOp<true, false, false>{in, sy_row, sc.alpha + i, s_i, r, y_next, none, 0}, sc.part[buf], grid);
	else                run_sweep<2>(sc, K_FWD, n, vec, FwdOp<false, false, false>{in, sy_row, sc.alpha + i, s_i, r, y_next, none, 0}, sc.part[buf], grid);
	return finish(sc, buf, 1, grid);
}

Partials launch_fwd_last(const Scratch& sc, int buf, size_t n, Partials in, const double* sy_row, int i,
                         const real* s_i, real* r, const ApplyArgs* fuse)
{
	const int grid = sweep_grid(sc, n);
	if (fuse) {
		const bool vec = all_aligned(s_i, r, fuse->x, fuse->x_sum, fuse->s_slot);
		if (sc.nontemporal) run_sweep<2>(sc, K_FWD_LAST, n, vec, FwdOp<true, true, true>{in, sy_row, sc.alpha + i, s_i, r, nullptr, *fuse, 0}, sc.part[buf], grid);
		else                run_sweep<2>(sc, K_FWD_LAST, n, vec, FwdOp<false, true, true>{in, sy_row, sc.alpha + i, s_i, r, nullptr, *fuse, 0}, sc.part[buf], grid);
		return Partials{nullptr, 0, 0};
	}
	const bool vec = all_aligned(s_i, r);
	ApplyArgs none{};
	if (sc.nontemporal) run_sweep<2>(sc, K_FWD_LAST, n, vec, FwdOp<true, true, false>{in, sy_row, sc.alpha + i, s_i, r, nullptr, none, 0}, sc.part[buf], grid);
	else                run_sweep<2>(sc, K_FWD_LAST, n, vec, FwdOp<false, true, false>{in, sy_row, sc.alpha + i, s_i, r, nullptr, none, 0}, sc.part[buf], grid);
	return finish(sc, buf, 2, grid);
}

void launch_apply(const Scratch& sc, size_t n, double n_global, Partials guard, const real* r_in, real* grad_out,
                  const ApplyArgs& a, bool guarded)
{
	const int grid = sweep_grid(sc, n, 2);
	const bool vec = all_aligned(r_in, grad_out, a.x, a.x_sum, a.s_slot);
	run_sweep<0>(sc, K_APPLY, n, vec, ApplyOp{guard, guarded, n_global, r_in, grad_out, a, sc.report, false}, nullptr, grid);
}

void launch_pair_s(const Scratch& sc, size_t n, real* x_sum, double inv_L, bool scale, const real* x_avg_prev,
                   real* s_out)
{
	const int grid = sweep_grid(sc, n, sc.pair_per_cu > 0 ? sc.pair_per_cu : 1);
	const bool vec = all_aligned(x_sum, x_avg_prev, s_out);
	run_sweep<0>(sc, K_PAIR_S, n, vec, PairSOp{x_sum, inv_L, scale, x_avg_prev, s_out}, nullptr, grid);
}

Partials launch_pair_y_diff(const Scratch& sc, int buf, size_t n, const real* g, const real* g_prev,
                            const real* s, double lambda, real* y_out, const VerdictArgs* verdict)
{
	const int grid = sweep_grid(sc, n, sc.pair_per_cu > 0 ? sc.pair_per_cu : 1);
	const bool vec = all_aligned(g, g_prev, s, y_out);
	if (verdict && !sc.allreduce && sc.ticket) {               // the last workgroup takes the verdict: no second launch
		run_sweep_verdict(sc, K_PAIR_Y_DIFF, n, vec, PairYDiffOp{g, g_prev, s, lambda, y_out}, sc.part[buf], grid, *verdict);
		return Partials{nullptr, 0, 0};
	}
	run_sweep<3>(sc, K_PAIR_Y_DIFF, n, vec, PairYDiffOp{g, g_prev, s, lambda, y_out}, sc.part[buf], grid);
	return finish(sc, buf, 3, grid);
}

Partials launch_pair_y_hv(const Scratch& sc, int buf, size_t n, const real* hv, const real* s, real* y_out,
                          real* x_sum, real* x_avg_prev, const VerdictArgs* verdict)
{
	const int grid = sweep_grid(sc, n, sc.pair_per_cu > 0 ? sc.pair_per_cu : 1);
	const bool vec = all_aligned(hv, s, y_out, x_sum, x_avg_prev);
	if (verdict && !sc.allreduce && sc.ticket) {
		run_sweep_verdict(sc, K_PAIR_Y_HV, n, vec, PairYHvOp{hv, s, y_out, x_sum, x_avg_prev}, sc.part[buf], grid, *verdict);
		return Partials{nullptr, 0, 0};
	}
	run_sweep<3>(sc, K_PAIR_Y_HV, n, vec, PairYHvOp{hv, s, y_out, x_sum, x_avg_prev}, sc.part[buf], grid);
	return finish(sc, buf, 3, grid);
}

Partials launch_dots3(const Scratch& sc, int buf, size_t n, const real* s, const real* y)
{
	const int grid = sweep_grid(sc, n);
	const bool vec = all_aligned(s, y);
	run_sweep<3>(sc, K_DOTS3, n, vec, Dots3Op{s, y}, sc.part[buf], grid);
	return finish(sc, buf, 3, grid);
}

// grid of the row-split Fisher pass: a whole number of resident rounds (register-limited, like the row-split rows-dot kernel)
template <int W, int RPW, int U>
static int fisher_split_launch(const Scratch& sc, size_t n, const real* F, size_t fu, const real* s)
{
	constexpr int NW = 8;
	static const int per_cu = [] {
		int blocks = 0;
		if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_fisher_t_split<W, RPW, true, NW, U>, 64 * NW, 0) != hipSuccess || blocks < 1) { (void) hipGetLastError(); blocks = 1; }
		return blocks > 4 ? 4 : blocks;
	}();
	const int want = sc.fisher_split_per_cu > 0 ? sc.fisher_split_per_cu : per_cu;
	size_t g = (size_t) sc.grid_cap * (size_t) want;
	const size_t max_grid = (n / W + (size_t) (64 * U) - 1) / (size_t) (64 * U);      // one workgroup covers 64 * U packs per trip
	if (g > max_grid) g = max_grid;
	if (g > (size_t) kMaxGrid) g = kMaxGrid;
	if (g < 1) g = 1;
	const dim3 grid((unsigned) g, (unsigned) ((fu + (size_t) (NW * RPW) - 1) / (size_t) (NW * RPW)));
	hipLaunchKernelGGL((k_fisher_t_split<W, RPW, true, NW, U>), grid, dim3(64 * NW), 0, sc.stream, F, n, (uint32_t) n, (uint32_t) fu, s, sc.fisher_part, (uint32_t) (sc.fisher_lag < 0 ? 0 : sc.fisher_lag));
	return (int) g;
}

Partials launch_fisher(const Scratch& sc, int buf, size_t n, const real* F, size_t fu, const real* s,
                       double* t_dev, real* y_out)
{
	const int grid = sweep_grid(sc, n);
	const bool vec = all_aligned(F, s, y_out);
	int grid_t = grid;
	if (sc.fisher_split) {
		// pass 1 with the rows divided among the waves of a workgroup: s once per 128 rows (k_fisher_t_split)
		ProfScope ps(sc, K_FISHER_T);
		#define SQN_FS(WW, UU) (fu <= 16 ? fisher_split_launch<WW, 2, UU>(sc, n, F, fu, s) : fu <= 32 ? fisher_split_launch<WW, 4, UU>(sc, n, F, fu, s) : \
		                        fu <= 64 ? fisher_split_launch<WW, 8, UU>(sc, n, F, fu, s) : fisher_split_launch<WW, 16, UU>(sc, n, F, fu, s))
		if (sc.fisher_tile >= 2) grid_t = vec ? SQN_FS(kVec, 2) : SQN_FS(1, 2);
		else                     grid_t = vec ? SQN_FS(kVec, 1) : SQN_FS(1, 1);
		#undef SQN_FS
	} else {
		// rows one workgroup accumulates per pass over its columns: s is re-read once per group of rows (fu = 128:
		// 16 groups of 8 re-read 12.5 % on top of F, 8 groups of 16 6 %).  Measured at n = 1e8 (r02_ab_rows3_fisher.jsonl):
		// fu = 128: 17.54 / 17.09 / 23.79 ms for 8 / 16 / 32 rows (32 rows: 64 row packs in flight per lane, the
		// compiler serialises them); fu = 32: 4.52 / 4.45 / 6.50 ms.  Default 16.
		const int fr = sc.fisher_rows >= 32 ? 32 : (sc.fisher_rows >= 16 ? 16 : 8);
		const dim3 g1(grid, (unsigned) ((fu + fr - 1) / fr));
		ProfScope ps(sc, K_FISHER_T);
		#define SQN_FT(WW, FR) hipLaunchKernelGGL((k_fisher_t<WW, true, FR>), g1, dim3(kBlock), 0, sc.stream, F, n, (uint32_t) n, (uint32_t) fu, s, sc.fisher_part)
		if (vec) { if (fr == 32) SQN_FT(kVec, 32); else if (fr == 16) SQN_FT(kVec, 16); else SQN_FT(kVec, 8); }
		else     { if (fr == 32) SQN_FT(1, 32); else if (fr == 16) SQN_FT(1, 16); else SQN_FT(1, 8); }
		#undef SQN_FT
	}
	launch_fin(sc, Partials{sc.fisher_part, grid_t, kMaxGrid}, (int) fu, t_dev);
	if (sc.allreduce) sc.allreduce(sc.user, t_dev, (int) fu, sc.stream);
	{
		ProfScope ps(sc, K_FISHER_Y);
		const size_t shmem = fu * sizeof(double);
		const double inv = 1.0 / (double) fu;
		// the parked results take 128 KB of a workgroup's 160 KB of LDS: t has to fit beside them
		const bool ph = sc.phase_inv != 0 && shmem <= 16384;
		#define SQN_FY(WW, PH) hipLaunchKernelGGL((k_fisher_y<WW, true, PH>), dim3(grid), dim3(kBlock), shmem, sc.stream, F, n, (uint32_t) n, (uint32_t) fu, t_dev, inv, s, y_out, sc.part[buf], sc.phase_inv)
		if (vec) { if (ph) SQN_FY(kVec, true); else SQN_FY(kVec, false); }
		else     { if (ph) SQN_FY(1, true); else SQN_FY(1, false); }
		#undef SQN_FY
	}
	return finish(sc, buf, 3, grid);
}

static bool rows_aligned(const RowSet& r)
{
	for (int j = 0; j < r.count; j++) if (!elem_aligned(r.row[j])) return false;
	return true;
}

// Workgroups of the row-split kernel that are resident per CU (register-limited).  The grid is a whole
// number of such rounds: 1024 workgroups at 3 per CU ran 40 % slower than 768 (a second, quarter-full round).
template <class K> static int resident_per_cu(K kernel, int threads)
{
	int blocks = 0;
	if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, threads, 0) != hipSuccess || blocks < 1) blocks = 1;
	return blocks > 8 ? 8 : blocks;
}

constexpr int kSplitWaves = 8;    // waves of a workgroup that split the rows: 4.94 ms against 5.10 for the all-rows form and for 4 waves (n = 1e8, 40 rows)

template <int W>
static int rows_dot_dispatch(const Scratch& sc, size_t max_grid, int rpw, const RowSet& rows, const real* probe, real* copy_out, uint32_t n, int rev)
{
	int grid = 1;
	#define SQN_RD(RPW)                                                                                                  \
		{                                                                                                                \
			static const int per_cu = resident_per_cu(k_rows_dot<W, RPW, true, kSplitWaves>, 64 * kSplitWaves);             \
			size_t g = (size_t) sc.grid_cap * per_cu;                                                                    \
			if (g > max_grid) g = max_grid;                                                                              \
			if (g > (size_t) kMaxGrid) g = kMaxGrid;                                                                     \
			grid = (int) g;                                                                                              \
			hipLaunchKernelGGL((k_rows_dot<W, RPW, true, kSplitWaves>), dim3(grid), dim3(64 * kSplitWaves), 0, sc.stream, rows, probe, copy_out, n, rev, sc.rows_part[0]); \
		}
	if (rpw <= 2) SQN_RD(2)
	else if (rpw <= 4) SQN_RD(4)
	else SQN_RD(6)                                   // k <= 48 rows over 8 waves
	#undef SQN_RD
	return grid;
}

// column tiles per iteration of the all-rows pass: two for the single-probe pass over at most 24 rows (option "sdot_tile", default 2:
// the registers of 2 x 24 row packs fit the one wave per SIMD the pass runs with), one otherwise
static int sdot_tiles(const Scratch& sc, int ng, bool two_probes)
{
	return (!two_probes && ng <= 3 && sc.sdot_tile >= 2) ? 2 : 1;
}

template <int W, int NPR>
static void rows_dot_all_dispatch(const Scratch& sc, int slot, int grid, int ng, const RowSet& rows, const Probes& probe,
                                  real* copy_out, uint32_t n, int rev, const Slice* slice = nullptr)
{
	const Slice sl = slice ? *slice : Slice{0u, n / (uint32_t) W, nullptr, 1, 1};
	#define SQN_RA_U(NG, UU) { if (slice) hipLaunchKernelGGL((k_rows_dot_all<W, NG, true, NPR, true, UU>), dim3(grid), dim3(kBlock), 0, sc.stream, rows, probe, copy_out, n, rev, sc.rows_part[slot], sl); \
	                           else hipLaunchKernelGGL((k_rows_dot_all<W, NG, true, NPR, false, UU>), dim3(grid), dim3(kBlock), 0, sc.stream, rows, probe, copy_out, n, rev, sc.rows_part[slot], sl); }
	#define SQN_RA(NG) SQN_RA_U(NG, 1)
	if constexpr (NPR == 1) {
		if (sdot_tiles(sc, ng, false) == 2) {
			switch (ng) {
			case 1: SQN_RA_U(1, 2); return;
			case 2: SQN_RA_U(2, 2); return;
			default: SQN_RA_U(3, 2); return;
			}
		}
	}
	switch (ng) {
	case 1: SQN_RA(1); break;
	case 2: SQN_RA(2); break;
	case 3: SQN_RA(3); break;
	case 4: SQN_RA(4); break;
	case 5: SQN_RA(5); break;
	default: SQN_RA(6); break;
	}
	#undef SQN_RA
	#undef SQN_RA_U
}

// pass 1 without a second probe as the row-split kernel (option "rows_split": the float build's default); one launch, never sliced.
// A workgroup covers 64 packs per step here (its waves split the rows), so it takes more workgroups than a sweep to keep
// the same number of lanes on the columns.
static int sdot_row_split(const Scratch& sc, size_t n, const RowSet& rows, const real* probe, real* copy_out, int rev)
{
	const bool vec = rows_aligned(rows) && all_aligned(probe, copy_out);
	size_t max_grid = (n / (vec ? kVec : 1) + 63) / 64;
	if (max_grid < 1) max_grid = 1;
	const int rpw = (rows.count + kSplitWaves - 1) / kSplitWaves;      // rows per wave
	ProfScope ps(sc, K_SDOT);
	return vec ? rows_dot_dispatch<kVec>(sc, max_grid, rpw, rows, probe, copy_out, (uint32_t) n, rev)
	           : rows_dot_dispatch<1>(sc, max_grid, rpw, rows, probe, copy_out, (uint32_t) n, rev);
}

// ---- three-pass form ---------------------------------------------------------------------------------
// pass 1: the single-probe / two-probe all-rows rows-dot over the k rows of S
static int sdot_grid(const Scratch& sc, size_t n, bool two_probes)
{
	// the two-probe variant (2 x k accumulators, 145 VGPRs: three waves per SIMD) wants three workgroups per CU: 2.78 ms
	// against 5.12 ms with one and 3.01 ms for the row-split shape (n = 1e8, k = 20; profiles/r02_ab_threepass_shapes.jsonl)
	return sweep_grid(sc, n, two_probes ? (sc.sdot2_per_cu > 0 ? sc.sdot2_per_cu : 3) : (sc.sdot_per_cu > 0 ? sc.sdot_per_cu : 1));
}

// doubles of carry scratch a sliced pass 1 over k rows needs (either probe variant): one accumulator per quantity and lane of the grid
size_t sdot_carry_count(const Scratch& sc, size_t n, int k)
{
	const size_t one = (size_t) k * (size_t) sdot_grid(sc, n, false), two = 2 * (size_t) k * (size_t) sdot_grid(sc, n, true);
	return (one > two ? one : two) * kBlock;
}

bool sdot_can_slice(const Scratch& sc, const RowSet& s_rows, const real* g, real* copy_out, const real* probe_y)
{
	return (probe_y || !sc.rows_split) && rows_aligned(s_rows) && all_aligned(g, copy_out, probe_y);
}

Partials launch_sdot(const Scratch& sc, size_t n, const RowSet& s_rows, const real* g, real* copy_out, const real* probe_y, const SliceFeed* feed)
{
	int grid = sdot_grid(sc, n, probe_y != nullptr);
	const bool vec = rows_aligned(s_rows) && all_aligned(g, copy_out, probe_y);
	const int rev = (sc.reverse && sc.phase) ? ((*sc.phase)++ & 1) : 0;
	const Probes pr{{g, probe_y}};
	const int ng = (s_rows.count + 7) / 8;
	const size_t packs = n / kVec, round = (size_t) grid * kBlock * (size_t) sdot_tiles(sc, ng, probe_y != nullptr);
	if (!probe_y && sc.rows_split) {
		if (feed) feed->arrive(feed->user, 0, n, 0);
		grid = sdot_row_split(sc, n, s_rows, g, copy_out, rev);
	} else if (feed && vec && feed->slices >= 2 && packs >= 2 * round && feed->carry &&
	           feed->carry_count >= (size_t) (probe_y ? 2 : 1) * (size_t) s_rows.count * (size_t) grid * kBlock) {      // one accumulator per quantity and LANE
		// the pass in slices of whole grid rounds; before slice s its part of g is sent for (feed->arrive), and the kernel of
		// that slice is what the stream runs once it has landed
		size_t per = (packs + (size_t) feed->slices - 1) / (size_t) feed->slices;
		per = (per + round - 1) / round * round;
		const size_t last = packs - 1;
		int s = 0;
		for (size_t pb = 0; pb < packs; pb += per, s++) {
			const size_t pe = pb + per < packs ? pb + per : packs;
			size_t lo = rev ? (last - (pe - 1)) * kVec : pb * kVec, hi = rev ? (last - pb + 1) * kVec : pe * kVec;
			if (hi == packs * kVec) hi = n;                      // the odd elements beyond the last pack travel with the slice next to them
			feed->arrive(feed->user, lo, hi, s);
			const Slice sl{(uint32_t) pb, (uint32_t) pe, feed->carry, pb == 0, pe == packs};
			ProfScope ps(sc, probe_y ? K_SDOT2 : K_SDOT);
			if (probe_y) rows_dot_all_dispatch<kVec, 2>(sc, 0, grid, ng, s_rows, pr, copy_out, (uint32_t) n, rev, &sl);
			else         rows_dot_all_dispatch<kVec, 1>(sc, 0, grid, ng, s_rows, pr, copy_out, (uint32_t) n, rev, &sl);
		}
	} else {
		if (feed) feed->arrive(feed->user, 0, n, 0);             // no slicing here: the whole vector at once
		ProfScope ps(sc, probe_y ? K_SDOT2 : K_SDOT);
		if (probe_y) {
			if (vec) rows_dot_all_dispatch<kVec, 2>(sc, 0, grid, ng, s_rows, pr, copy_out, (uint32_t) n, rev);
			else     rows_dot_all_dispatch<1, 2>(sc, 0, grid, ng, s_rows, pr, copy_out, (uint32_t) n, rev);
		} else {
			if (vec) rows_dot_all_dispatch<kVec, 1>(sc, 0, grid, ng, s_rows, pr, copy_out, (uint32_t) n, rev);
			else     rows_dot_all_dispatch<1, 1>(sc, 0, grid, ng, s_rows, pr, copy_out, (uint32_t) n, rev);
		}
	}
	const int nq = (probe_y ? 2 : 1) * s_rows.count;
	Partials raw{sc.rows_part[0], grid, kMaxGrid};
	if (!sc.allreduce) return raw;
	launch_fin(sc, raw, nq, sc.red[0]);
	sc.allreduce(sc.user, sc.red[0], nq, sc.stream);
	return Partials{sc.red[0], 1, 1};
}

static uint32_t keep_from_pack(const Scratch& sc, size_t n, bool vec)
{
	// first pack (in traversal order) that is stored with the default cache policy; the packs before it stream past the caches
	const double f = sc.keep_tail;
	if (!(f > 0)) return 0xFFFFFFFFu;
	const size_t packs = n / (vec ? kVec : 1);
	if (f >= 1) return 0;
	return (uint32_t) (packs - (size_t) ((double) packs * f));
}

static Fold3 fold_args(const Scratch& sc, const Partials& in, const CoefArgs& a, int fresh_row)
{
	Fold3 f{};
	f.parts = in.parts; f.count = in.count; f.stride = in.stride;
	f.a = a;
	f.fresh_row = fresh_row;
	f.gsy = sc.gsy; f.sy = sc.sy; f.yy = sc.yy; f.alpha = sc.alpha; f.rho_out = sc.rho;
	return f;
}

Partials launch_qdot(const Scratch& sc, size_t n, const RowSet& y_rows, real* g, const QdotScale& q, const Partials& pass1,
                     const CoefArgs& a, int fresh_row)
{
	const int grid = sweep_grid(sc, n, sc.qdot_per_cu > 0 ? sc.qdot_per_cu : 1);
	const bool vec = rows_aligned(y_rows) && all_aligned(g, q.H0_in, q.G, q.H0_out, q.frow_out);
	const int rev = (sc.reverse && sc.phase) ? ((*sc.phase)++ & 1) : 0;
	const int ng = (y_rows.count + 7) / 8;
	const int mode = q.G ? 2 : (q.H0_in ? 1 : 0);
	DiagArgs dg{q.H0_in, q.G, q.H0_out, q.frow_out, q.rmsprop_weight, 1 - q.rmsprop_weight, q.scal_reg, q.rmsprop_weight > 0 && q.rmsprop_weight < 1};
	const Fold3 fo = fold_args(sc, pass1, a, fresh_row);
	const uint32_t keep = keep_from_pack(sc, n, vec);
	const bool ph = sc.phase_inv != 0;
	{
		ProfScope ps(sc, K_QDOT);
		#define SQN_QD3(WW, NG, MODE) { if (ph) hipLaunchKernelGGL((k_qdot<WW, NG, true, MODE, true>), dim3(grid), dim3(kBlock), 0, sc.stream, y_rows, fo, g, dg, (uint32_t) n, rev, keep, sc.rows_part[1], sc.phase_inv); \
		                                else hipLaunchKernelGGL((k_qdot<WW, NG, true, MODE, false>), dim3(grid), dim3(kBlock), 0, sc.stream, y_rows, fo, g, dg, (uint32_t) n, rev, keep, sc.rows_part[1], 0u); }
		#define SQN_QD2(WW, NG) { if (mode == 2) SQN_QD3(WW, NG, 2) else if (mode == 1) SQN_QD3(WW, NG, 1) else SQN_QD3(WW, NG, 0) }
		#define SQN_QD1(WW) { if (ng <= 1) SQN_QD2(WW, 1) else if (ng == 2) SQN_QD2(WW, 2) else if (ng == 3) SQN_QD2(WW, 3) else if (ng == 4) SQN_QD2(WW, 4) else if (ng == 5) SQN_QD2(WW, 5) else SQN_QD2(WW, 6) }
		if (vec) SQN_QD1(kVec) else SQN_QD1(1)
		#undef SQN_QD1
		#undef SQN_QD2
		#undef SQN_QD3
	}
	Partials raw{sc.rows_part[1], grid, kMaxGrid};
	if (!sc.allreduce) return raw;
	launch_fin(sc, raw, y_rows.count, sc.red[1]);
	sc.allreduce(sc.user, sc.red[1], y_rows.count, sc.stream);
	return Partials{sc.red[1], 1, 1};
}

// pass 3: one workgroup per CU (its lanes hold all k rows of a pack, and the parked results take most of a CU's LDS)
static int sadd_grid(const Scratch& sc, size_t n) { return sweep_grid(sc, n, sc.sadd_per_cu > 0 ? sc.sadd_per_cu : 1); }

// pass 3 can run in slices when the lanes work on packs (the elements beyond the last pack are written by the last launch,
// wherever the traversal ends: they are reported on their own when that is not where the last slice lies)
bool sadd_can_slice(const Scratch& sc, size_t n, const RowSet& s_rows, const real* r, const SliceFeed* drain)
{
	if (!drain || drain->slices < 2 || !rows_aligned(s_rows) || !all_aligned(r)) return false;
	const int grid = sadd_grid(sc, n);
	if (!drain->carry || drain->carry_count < 2 * (size_t) grid * kBlock) return false;
	return n / kVec >= 2 * (size_t) grid * kBlock;
}

// ---- checksum of a vector's bit pattern (sqn_device.hpp: XHash) ---------------------------------------------------------
// Read-only, one pass, 16-byte loads; every word is mixed with its position (xhash_word), the sums of the mixed words are integers
// mod 2^64, so any order of accumulation gives the same two words.
__global__ void __launch_bounds__(kBlock) k_xhash(const ulonglong2* __restrict__ w2, size_t pairs, const unsigned char* __restrict__ rest,
                                                  int rest_bytes, unsigned long long* __restrict__ out)
{
	unsigned long long a = 0, b = 0;
	const size_t stride = (size_t) gridDim.x * kBlock;
	for (size_t p = (size_t) blockIdx.x * kBlock + threadIdx.x; p < pairs; p += stride) {
		const ulonglong2 v = w2[p];
		xhash_word(v.x, 2 * p, a, b);                         // words 2p and 2p + 1
		xhash_word(v.y, 2 * p + 1, a, b);
	}
	if (blockIdx.x == 0 && threadIdx.x == 0 && rest_bytes > 0) {      // what is left of the buffer after the last pair: up to 15 bytes
		unsigned long long w[2] = {0, 0};
		for (int i = 0; i < rest_bytes; i++) w[i / 8] |= (unsigned long long) rest[i] << (8 * (i % 8));
		xhash_word(w[0], 2 * pairs, a, b);
		if (rest_bytes > 8) xhash_word(w[1], 2 * pairs + 1, a, b);
	}
	for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
	__shared__ unsigned long long la[kBlock / 64], lb[kBlock / 64];
	const int lane = threadIdx.x % 64, wave = threadIdx.x / 64;
	if (lane == 0) { la[wave] = a; lb[wave] = b; }
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int i = 1; i < kBlock / 64; i++) { a += la[i]; b += lb[i]; }
		atomicAdd(out, a);
		atomicAdd(out + 1, b);
	}
}

void launch_xhash(const Scratch& sc, const real* x, size_t n, double* out2)
{
	const size_t bytes = n * sizeof(real), pairs = bytes / 16;
	unsigned long long* out = reinterpret_cast<unsigned long long*>(out2);      // zeroed by the caller, on this stream
	size_t grid = (pairs + (size_t) kBlock * 8 - 1) / ((size_t) kBlock * 8);
	if (grid < 1) grid = 1;
	if (grid > (size_t) kMaxGrid) grid = (size_t) kMaxGrid;
	ProfScope ps(sc, K_XHASH);
	hipLaunchKernelGGL(k_xhash, dim3((unsigned) grid), dim3(kBlock), 0, sc.stream, reinterpret_cast<const ulonglong2*>(x), pairs,
	                   reinterpret_cast<const unsigned char*>(x) + pairs * 16, (int) (bytes - pairs * 16), out);
}

void launch_spec_x(const Scratch& sc, size_t n, const real* r, const real* x, double step, real* out)
{
	Scratch local = sc;
	local.phase = nullptr;                                     // element-wise: the direction of the traversal is of no consequence
	const int grid = sweep_grid(local, n, 2);
	run_sweep<0>(local, K_APPLY, n, all_aligned(r, x, out), SpecXOp{r, x, out, step}, nullptr, grid);
}

Partials launch_sadd(const Scratch& sc, int buf, size_t n, const RowSet& s_rows, real* r, const Partials& pass2, const CoefArgs& a,
                     const SliceFeed* drain)
{
	const int grid = sadd_grid(sc, n);
	const bool vec = rows_aligned(s_rows) && all_aligned(r);
	const int rev = (sc.reverse && sc.phase) ? ((*sc.phase)++ & 1) : 0;
	const Fold3 fo = fold_args(sc, pass2, a, -1);
	const uint32_t keep = keep_from_pack(sc, n, vec);
	const bool ph = sc.phase_inv != 0;
	const int ng = (s_rows.count + 7) / 8;
	{
		ProfScope ps(sc, K_SADD);
		#define SQN_SA3(WW, NG, SL, SLICE) { if (ph) hipLaunchKernelGGL((k_sadd<WW, NG, true, true, SL>), dim3(grid), dim3(kBlock), 0, sc.stream, s_rows, fo, r, (uint32_t) n, rev, keep, sc.part[buf], SLICE, sc.phase_inv); \
		                                     else hipLaunchKernelGGL((k_sadd<WW, NG, true, false, SL>), dim3(grid), dim3(kBlock), 0, sc.stream, s_rows, fo, r, (uint32_t) n, rev, keep, sc.part[buf], SLICE, 0u); }
		#define SQN_SA(WW, SL, SLICE) { if (ng <= 1) SQN_SA3(WW, 1, SL, SLICE) else if (ng == 2) SQN_SA3(WW, 2, SL, SLICE) else if (ng == 3) SQN_SA3(WW, 3, SL, SLICE) \
		                                else if (ng == 4) SQN_SA3(WW, 4, SL, SLICE) else if (ng == 5) SQN_SA3(WW, 5, SL, SLICE) else SQN_SA3(WW, 6, SL, SLICE) }
		if (sadd_can_slice(sc, n, s_rows, r, drain)) {
			// the pass in slices of whole rounds of the grid; after the launch of slice s its part of r is final (a phased kernel
			// flushes what it has parked before it ends) and drain->arrive is told: a host caller's x is sent on its way from
			// there (machines.cpp: enqueue_step)
			const size_t packs = n / kVec, round = (size_t) grid * kBlock;
			size_t per = (packs + (size_t) drain->slices - 1) / (size_t) drain->slices;
			per = (per + round - 1) / round * round;
			const size_t last = packs - 1;
			int s = 0;
			for (size_t pb = 0; pb < packs; pb += per, s++) {
				const size_t pe = pb + per < packs ? pb + per : packs;
				const size_t lo = rev ? (last - (pe - 1)) * kVec : pb * kVec;
				size_t hi = rev ? (last - pb + 1) * kVec : pe * kVec;
				if (!rev && pe == packs) hi = n;                  // forward: the odd elements lie next to the last slice
				const Slice sl{(uint32_t) pb, (uint32_t) pe, drain->carry, pb == 0, pe == packs};
				SQN_SA(kVec, true, sl)
				drain->arrive(drain->user, lo, hi, s);
			}
			if (rev && n > packs * kVec) drain->arrive(drain->user, packs * kVec, n, s);   // reversed: they lie at the other end
		}
		else if (vec) SQN_SA(kVec, false, Slice{})
		else SQN_SA(1, false, Slice{})
		#undef SQN_SA
		#undef SQN_SA3
	}
	return finish(sc, buf, 2, grid);
}

// one column of the cached block outside the normal path (imported state, isolated entries): out[j] = total of partial j
void launch_store_column(const Scratch& sc, Partials in, const CoefArgs& a, int col_row)
{
	ProfScope ps(sc, K_SMALL);
	hipLaunchKernelGGL(k_store_column, dim3(1), dim3(kBlock), 0, sc.stream, in.parts, in.count, in.stride, a, col_row, sc.gsy);
}

void launch_fin(const Scratch& sc, Partials in, int nsums, double* out)
{
	ProfScope ps(sc, K_FIN);
	hipLaunchKernelGGL(k_fin, dim3(nsums), dim3(kBlock), 0, sc.stream, in.parts, in.count, in.stride, out);
}

void launch_commit(const Scratch& sc, Partials in, double* sy_dst, double* yy_dst)
{
	ProfScope ps(sc, K_SMALL);
	hipLaunchKernelGGL(k_commit, dim3(1), dim3(kBlock), 0, sc.stream, in.parts, in.count, in.stride, sy_dst, yy_dst);
}

void launch_verdict(const Scratch& sc, Partials in, double min_curvature, double* sy_dst, double* yy_dst, double* out)
{
	ProfScope ps(sc, K_SMALL);
	hipLaunchKernelGGL(k_verdict, dim3(1), dim3(kBlock), 0, sc.stream, in.parts, in.count, in.stride, min_curvature, sy_dst, yy_dst, out);
}

uint64_t synth_key(uint64_t seed, uint64_t stream, uint64_t t)
{
	return seed ^ (stream * 0x9E3779B97F4A7C15ull) ^ (t * 0xD1B54A32D192ED03ull);
}

static int synth_grid(size_t count)
{
	size_t g = (count + kBlock - 1) / kBlock;
	return (int) (g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

void launch_synth_uniform(hipStream_t stream, real* out, size_t count, uint64_t first, uint64_t key, double a, double b)
{
	hipLaunchKernelGGL(k_synth_uniform, dim3(synth_grid(count)), dim3(kBlock), 0, stream, out, count, first, key, a, b);
}

void launch_synth_grad(hipStream_t stream, real* g, const real* d, const real* x, size_t count, uint64_t first, uint64_t key, double amp)
{
	size_t blocks = (count / kVec + kBlock - 1) / kBlock;
	if (blocks < 1) blocks = 1;
	if (blocks > 1024) blocks = 1024;
	hipLaunchKernelGGL(k_synth_grad, dim3((unsigned) blocks), dim3(kBlock), 0, stream, g, d, x, count, first, key, amp);
}

void launch_synth_batch_row(hipStream_t stream, real* row, const real* d, size_t count, uint64_t first, uint32_t k, uint32_t bs)
{
	hipLaunchKernelGGL(k_synth_batch_row, dim3(synth_grid(count)), dim3(kBlock), 0, stream, row, d, count, first, k, bs);
}

void launch_scale(const Scratch& sc, size_t n, real* x, double a)
{
	const int grid = sweep_grid(sc, n);
	run_sweep<0>(sc, K_SMALL, n, elem_aligned(x), ScaleOp{x, a}, nullptr, grid);
}

}  // namespace sqn
