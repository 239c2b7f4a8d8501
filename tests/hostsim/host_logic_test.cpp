// host_logic_test.cpp -- scenarios that drive the product's host logic on the CPU (TEST INFRASTRUCTURE ONLY).
//
//   host_logic_{asan,tsan} <scenario> [lazy | per_stream]
//
// Linked against the unchanged runtime.cpp / machines.cpp / group.cpp, the malloc-backed HIP stand-in (fake_hip.cpp), the
// launcher stand-ins (fake_launch.cpp: x -= step * grad, the scalars of the recursion scripted) and, through
// STOCHQN_HIP_RCCL_LIB, the rendezvous stand-in for RCCL.  Every scenario goes through the C ABI of include/stochqn.h the
// way a profile-B caller does (R / Cython: every array the caller's, structs rebuilt per call, counters copied back) or a
// profile-A caller (initialize_* / dealloc_*), checks the iterate against x0 - sum step * grad after every call, and ends
// with a leak check of the fake runtime (no device allocation, stream, event or pinned range left behind).
// Exit code 0 = the scenario passed; the sanitizers add their own verdict.
#include "fake_hip.hpp"
#include "fake_launch.hpp"

#include "stochqn.h"
#include "stochqn_hip.h"
#include "runtime.hpp"          // the host checksum of x (xhash_*): tested against its definition directly

#include <dlfcn.h>
#include <malloc.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/asan_interface.h>
#endif
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

namespace {

int g_failures = 0;
std::string g_tag;                                 // what the fault sweeps are injecting right now
#define CHECK(cond, ...)                                                                                    \
	do {                                                                                                    \
		if (!(cond)) {                                                                                      \
			std::fprintf(stderr, "CHECK FAILED %s:%d: [%s] %s -- ", __FILE__, __LINE__, g_tag.c_str(), #cond);  \
			std::fprintf(stderr, __VA_ARGS__);                                                              \
			std::fprintf(stderr, "\n");                                                                     \
			g_failures++;                                                                                   \
		}                                                                                                   \
	} while (0)

void opt(const char* name, double v)
{
	if (stochqn_hip_set_option(name, v) != 0) { std::fprintf(stderr, "unknown option %s\n", name); g_failures++; }
}
long long stat(const char* name) { return stochqn_hip_stat(name); }

enum Kind { OLBFGS, SQN, ADAQN };

// A profile-B optimiser object: the caller owns every array (host memory), rebuilds the structs on its stack for every call
// and copies the counters back (reference src/Rwrapper.c:98-196, stochqn/pywrapper.pxi:161-207).
struct Opt {
	Kind kind;
	int n;
	size_t m, L, fsize;
	bool grad_diff;
	double min_curv, max_incr;
	int check_nan = 1;
	std::vector<double> S, Y, rho, alpha, sbak, ybak, gprev, xsum, xprev, H0, G, F, fy;
	std::vector<double> x, grad, hv, x_ref;
	size_t mem_used = 0, st = 0, niter = 0, f_used = 0, f_st = 0;
	int section = 0;
	double f_prev = 0, f = 1.0;
	double *req = nullptr, *req_vec = nullptr;
	task_enum task = calc_grad;
	info_enum info = no_problems_encountered;
	int calls = 0, accepted = 0, failed = 0;
	bool check_x = true;

	Opt(Kind k, int n_, size_t m_, size_t L_, bool gd = false, double mc = 0.0, size_t fs = 0, double mi = 0.0)
		: kind(k), n(n_), m(m_), L(L_), fsize(fs), grad_diff(gd), min_curv(mc), max_incr(mi)
	{
		const size_t N = (size_t) n;
		S.assign(m * N, 0.0); Y.assign(m * N, 0.0); rho.assign(m, 0.0); alpha.assign(m, 0.0);
		sbak.assign(mc > 0 ? N : 1, 0.0); ybak.assign(mc > 0 ? N : 1, 0.0);
		gprev.assign((k == OLBFGS || gd) ? N : 1, 0.0);
		xsum.assign(N, 0.0); xprev.assign(N, 0.0);
		if (k == ADAQN) { H0.assign(N, 0.0); G.assign(N, 0.0); F.assign(gd ? 1 : (fs ? fs : 1) * N, 0.0); fy.assign(fs ? fs : 1, 0.0); }
		x.resize(N); grad.assign(N, 0.0); hv.assign(gd ? 1 : N, 0.0);
		for (size_t i = 0; i < N; i++) x[i] = 1.0 + 0.001 * (double) (i % 97);
		x_ref = x;
		req = x.data();
	}

	const void* key() const { return S.data(); }

	// the caller's side of the protocol: what the library asked for at *req, on the host
	void answer()
	{
		const size_t N = (size_t) n;
		if (task == calc_grad || task == calc_grad_same_batch || task == calc_grad_big_batch)
			for (size_t i = 0; i < N; i++) grad[i] = 0.25 * req[i] + 0.01;
		else if (task == calc_hess_vec)
			for (size_t i = 0; i < N; i++) hv[i] = 2.0 * req_vec[i];
		else if (task == calc_fun_val_batch) { f = 0; for (size_t i = 0; i < N; i++) f += req[i] * req[i]; }
	}

	int call(double step)
	{
		bfgs_mem b{S.data(), Y.data(), rho.data(), alpha.data(), sbak.data(), ybak.data(), m, mem_used, st, kind == OLBFGS ? 1 : L, 0.0, min_curv};
		int rc = -1000;
		const bool takes_step = section == 1;
		std::vector<double> g_in = takes_step ? grad : std::vector<double>();
		if (kind == OLBFGS) {
			workspace_oLBFGS w{&b, gprev.data(), 0.0, niter, section, 1, check_nan, n};
			rc = run_oLBFGS(step, x.data(), grad.data(), &req, &task, &w, &info);
			niter = w.niter; section = w.section;
		} else if (kind == SQN) {
			workspace_SQN w{&b, gprev.data(), xsum.data(), xprev.data(), grad_diff, niter, section, 1, check_nan, n};
			rc = run_SQN(step, x.data(), grad.data(), hv.data(), &req, &req_vec, &task, &w, &info);
			niter = w.niter; section = w.section;
		} else {
			fisher_mem fm{F.data(), fy.data(), fsize, f_used, f_st};
			workspace_adaQN w{&b, grad_diff ? nullptr : &fm, H0.data(), gprev.data(), xsum.data(), xprev.data(), G.data(), f_prev, max_incr, 1e-4, 0.9,
			                  grad_diff, niter, section, 1, check_nan, n};
			rc = run_adaQN(step, x.data(), f, grad.data(), &req, &task, &w, &info);
			niter = w.niter; section = w.section; f_prev = w.f_prev;
			f_used = fm.mem_used; f_st = fm.mem_st_ix;
		}
		mem_used = b.mem_used; st = b.mem_st_ix;
		calls++;
		if (rc == -1000) { failed++; check_x = false; return rc; }      // after a failed call nothing is promised about x any more
		if (takes_step && rc == 1 && info != func_increased) {
			accepted++;
			for (size_t i = 0; i < (size_t) n; i++) x_ref[i] -= step * g_in[i];      // the stand-in's direction is the gradient
		}
		if (info == func_increased) x_ref = x;                                       // x <- x_avg_prev (not followed here)
		if (check_x) {
			double worst = 0;
			for (size_t i = 0; i < (size_t) n; i++) worst = std::fmax(worst, std::fabs(x[i] - x_ref[i]));
			CHECK(worst <= 1e-12, "call %d (kind %d, section -> %d): the caller's x is %.3g away from x0 - sum step * grad", calls, (int) kind, section, worst);
		}
		return rc;
	}

	// `count` calls of the reverse-communication loop; returns the last return code
	int drive(int count, double step = 0.01)
	{
		int rc = 0;
		for (int k = 0; k < count; k++) {
			answer();
			rc = call(step);
			if (rc == -1000) break;
		}
		return rc;
	}
};

void leak_check(const char* where, long pinned_allowed = 1)
{
	stochqn_hip_release_all();
	const fakehip::Live l = fakehip::live();
	CHECK(l.device_allocs == 0, "%s: %ld device allocations (%ld bytes) left behind", where, l.device_allocs, l.device_bytes);
	CHECK(l.streams == 0 && l.events == 0, "%s: %ld streams, %ld events left behind", where, l.streams, l.events);
	CHECK(l.registered_ranges == 0, "%s: %ld host ranges still pinned", where, l.registered_ranges);
	CHECK(l.pinned_allocs <= pinned_allowed, "%s: %ld pinned allocations left behind (the bounce buffer of the reclaim path may stay)", where, l.pinned_allocs);
	CHECK(fakehip::violations() == 0, "%s: %ld violations of the runtime's rules, last: %s", where, fakehip::violations(), fakehip::last_violation());
}

void defaults()
{
	fakehip::reset();
	fakehip::set_devices(1);
	fakehip::set_capacity(0);
	fakelaunch::script() = fakelaunch::Script{};
	opt("devices", 0); opt("virtual_devices", 0); opt("devices_min_n", 1 << 20); opt("max_mirror_bytes", 0);
	opt("register_host", 0); opt("register_min_bytes", 4 << 20); opt("x_upload", 1); opt("x_prefetch", 0);
	opt("upload_slices", 8); opt("apply_chunks", 8); opt("spec_x", 1); opt("strict_grad", 0); opt("threepass", 1);
	opt("fail_alloc_after", -1); opt("reducer_patience_s", 120); opt("host_slice_min", 1 << 21);
	stochqn_hip_stats_reset();
}

// ---- scenarios ----------------------------------------------------------------------------------------------------------
// context registry: create / look up / release, another object at the same address, section 0 after a restart, export
void sc_registry()
{
	Opt a(SQN, 3000, 4, 3), b(OLBFGS, 2000, 3, 1), c(ADAQN, 1500, 3, 4, false, 1e-4, 6, 1.01);
	a.drive(25); b.drive(30); c.drive(40);
	CHECK(stat("contexts_created") == 3, "contexts_created = %lld", stat("contexts_created"));
	CHECK(a.accepted >= 10 && b.accepted >= 10 && c.accepted >= 10, "steps taken: %d %d %d", a.accepted, b.accepted, c.accepted);
	CHECK(a.mem_used == 4 && b.mem_used == 3, "rings: %zu %zu", a.mem_used, b.mem_used);
	CHECK(stochqn_hip_export(a.key()) == 0 && stochqn_hip_export(b.key()) == 0 && stochqn_hip_export(c.key()) == 0, "export");
	stochqn_hip_invalidate(a.key());
	a.drive(8);
	// a brand-new object in the same arrays (R / Python never call dealloc_*): section 0 drops the old context
	a.section = 0; a.niter = 0; a.mem_used = 0; a.st = 0; a.x_ref = a.x;
	a.drive(12);
	CHECK(stat("contexts_created") == 4, "a restart makes a new context: %lld", stat("contexts_created"));
	// the same address with another shape
	Opt d(SQN, 1000, 4, 3);
	d.S.swap(a.S);                                   // d now lives at a's old s_mem address, with another n
	d.S.resize(4 * 1000);
	d.drive(10);
	stochqn_hip_release(b.key());
	b.drive(6);                                      // released in mid-flight: re-imported from the caller's arrays, goes on
	CHECK(stochqn_hip_export((const void*) 0x10) == -1000, "export of nothing");
	leak_check("registry");
}

// least-recently-used reclaim under memory pressure -> spill -> the object comes back -> resume from the library's own copy
void sc_reclaim_resume()
{
	const int n = 4000;
	const size_t m = 5;
	Opt a(SQN, n, m, 1000), b(SQN, n, m, 1000);       // L = 1000: no pair is built, S and Y stay what was imported
	for (size_t i = 0; i < a.S.size(); i++) { a.S[i] = 1.0 + (double) i; a.Y[i] = -2.0 - (double) i; }
	a.mem_used = 3; a.st = 3; a.niter = 7; a.section = 1;                 // a resumed object: its arrays are imported
	// room for one object's mirrors (2 m n + 2 n doubles) and the scratch of two contexts, not for two objects' mirrors
	const size_t mirrors = (2 * m + 2) * (size_t) n * 8;
	a.drive(4);
	const fakehip::Live one = fakehip::live();
	fakehip::set_capacity((size_t) one.device_bytes + mirrors / 2 + (one.device_bytes - mirrors));
	const std::vector<double> S0 = a.S, Y0 = a.Y;
	b.drive(4);                                       // needs a's memory: a is spilled to library-owned host memory and destroyed
	CHECK(stat("contexts_reclaimed") == 1, "contexts_reclaimed = %lld", stat("contexts_reclaimed"));
	CHECK(b.failed == 0, "the newcomer must not fail");
	// the caller's arrays of a are stale by now (an R vector may even be gone): scribble over them
	std::vector<double> xs = a.xsum;
	for (double& v : a.S) v = 777.0;
	for (double& v : a.Y) v = 777.0;
	for (double& v : a.xsum) v = 777.0;
	// resume under an injected failure first: the half-made context is dropped, the library's copy must survive it
	for (long k = 0; k < 6; k++) {
		opt("fail_alloc_after", (double) k);
		fakehip::set_capacity(0);                    // plenty of memory now: only the injected failure is in the way
		a.answer();
		const int rc = a.call(0.01);
		opt("fail_alloc_after", -1);
		if (rc != -1000) break;
		a.failed = 0;
	}
	CHECK(a.failed == 0, "a resumed in the end");
	a.drive(3);
	CHECK(stochqn_hip_export(a.key()) == 0, "export");
	size_t bad = 0;
	for (size_t i = 0; i < S0.size(); i++) bad += (a.S[i] != S0[i]) + (a.Y[i] != Y0[i]);
	CHECK(bad == 0, "%zu elements of S / Y differ from what the reclaimed context held (the stale arrays were imported?)", bad);
	// x_sum went on from the spilled value, not from the scribble: 777 * n would show
	double sum = 0;
	for (double v : a.xsum) sum += v;
	CHECK(sum < 700.0 * n, "x_sum continued from the stale caller array");
	leak_check("reclaim_resume");
}

// option "max_mirror_bytes": idle contexts beyond the cap are spilled, everybody keeps working
void sc_mirror_cap()
{
	const int n = 2000;
	std::vector<Opt> objs;
	for (int k = 0; k < 4; k++) objs.emplace_back(SQN, n, 4, 3);
	opt("max_mirror_bytes", (double) ((2 * 4 + 2) * n * 8 * 2 + 100));        // two objects' worth
	for (int round = 0; round < 5; round++)
		for (auto& o : objs) o.drive(5);
	CHECK(stat("contexts_reclaimed") >= 4, "contexts_reclaimed = %lld", stat("contexts_reclaimed"));
	for (auto& o : objs) CHECK(o.failed == 0 && o.accepted >= 10, "object: failed %d accepted %d", o.failed, o.accepted);
	opt("max_mirror_bytes", 0);
	leak_check("mirror_cap");
}

// host callers of a large-enough problem: pinning in place, the gradient in slices, x ahead of the guard, a rejected step
void sc_host_path()
{
	const int n = (1 << 16) + 3;                       // the sliced paths, at a size the sanitizers get through quickly
	opt("host_slice_min", 1 << 12);
	opt("register_min_bytes", 1 << 12);
	for (int strict = 0; strict < 2; strict++) {
		opt("strict_grad", strict);
		Opt a(SQN, n, 3, 3);
		a.drive(22);
		while (a.section != 1) a.drive(1);           // the next call takes a step
		CHECK(a.failed == 0 && a.mem_used >= 2, "failed %d ring %zu", a.failed, a.mem_used);
		fakelaunch::script().reject_step = true;     // the guard says no: x untouched, what went ahead is put right
		const size_t before = a.mem_used;
		a.drive(1);
		fakelaunch::script().reject_step = false;
		CHECK(a.info == search_direction_was_nan && a.mem_used == 0 && before > 0, "a rejected step flushes the ring (info %d)", (int) a.info);
		a.drive(10);
		stochqn_hip_release(a.key());
	}
	CHECK(stat("x_sent_ahead") > 0 && stat("x_sent_again") > 0, "x ahead of the guard: %lld, sent again: %lld", stat("x_sent_ahead"), stat("x_sent_again"));
	// who pins what.  By default the library registers nothing by itself; the owner of an array pins it (and unpins it)
	{
		CHECK(fakehip::live().registered_ranges == 0, "the library pinned %ld ranges on its own", fakehip::live().registered_ranges);
		Opt p(SQN, n, 3, 3);
		const size_t bytes = (size_t) n * sizeof(double);
		CHECK(stochqn_hip_pin_host(p.x.data(), bytes) == 0 && stochqn_hip_pin_host(p.grad.data(), bytes) == 0, "pin");
		CHECK(stochqn_hip_pin_host(p.x.data(), bytes) == 0, "pins nest");
		p.drive(12);
		CHECK(fakehip::is_registered(p.x.data()) && fakehip::is_registered(p.grad.data()) && !fakehip::is_registered(p.hv.data()), "pinned: x, grad; not hess_vec");
		stochqn_hip_release(p.key());
		CHECK(stochqn_hip_unpin_host(p.x.data()) == 0 && fakehip::is_registered(p.x.data()), "one of two pins of x released");
		CHECK(stochqn_hip_unpin_host(p.x.data()) == 0 && stochqn_hip_unpin_host(p.grad.data()) == 0 && stochqn_hip_unpin_host(p.grad.data()) == -1, "unpin");
		CHECK(fakehip::live().registered_ranges == 0, "ranges left pinned: %ld", fakehip::live().registered_ranges);
		// ... but only a range that has its pages to itself (runtime.cpp: pinnable_in_place): not a block of the program-break heap
		// (the sanitizers' allocators never use the break, so a piece of it is taken directly), not one that shares a page with a live pin
		const long long declined = stat("host_pins_declined");
		void* blk = sbrk(1 << 16);
		CHECK(blk != (void*) -1 && stochqn_hip_pin_host(blk, 1 << 16) == 1 && stat("host_pins_declined") == declined + 1, "a block of the break heap must be declined");
		CHECK(stochqn_hip_unpin_host(blk) == -1 && fakehip::live().registered_ranges == 0, "a declined range is not pinned");
		std::vector<double> two(2 * (size_t) n + 2048);
		double* first = two.data();
		double* second = (double*) (((uintptr_t) (first + n) & ~(uintptr_t) 4095) + 64);      // begins in the page the first range ends in
		CHECK(stochqn_hip_pin_host(first, bytes) == 0 && stochqn_hip_pin_host(second, bytes / 2) == 1, "a range that shares a page with a live pin must be declined");
		CHECK(stochqn_hip_pin_host((char*) second + 8192, bytes / 2) == 0, "two pages on: pages of its own");
		CHECK(stochqn_hip_unpin_host(first) == 0 && stochqn_hip_unpin_host((char*) second + 8192) == 0 && fakehip::live().registered_ranges == 0, "unpin");
	}
	// register_host = 1 (the caller vouches for the lifetime of its arrays): the library pins what it saw at the same address twice
	opt("register_host", 1);
	{
		Opt q(OLBFGS, n, 3, 1);
		q.drive(2);                                  // section 0, then the first step: x and grad seen once
		CHECK(!fakehip::is_registered(q.x.data()), "an array seen once is not pinned");
		q.drive(4);
		CHECK(fakehip::is_registered(q.x.data()) && fakehip::is_registered(q.grad.data()), "x and grad pinned after the second sighting");
	}
	CHECK(fakehip::live().registered_ranges >= 2, "the context keeps its registrations until it goes");
	stochqn_hip_release_all();
	CHECK(fakehip::live().registered_ranges == 0, "ranges left pinned: %ld", fakehip::live().registered_ranges);
	// x_upload = 0: the caller vouches for x between calls; an edit the probes see makes the library upload it again
	opt("x_upload", 0);
	Opt c(OLBFGS, n, 3, 1);
	c.drive(9);
	for (size_t i = 0; i < (size_t) n; i++) c.x[i] *= 0.5;
	c.x_ref = c.x;
	c.drive(9);
	CHECK(c.failed == 0 && stat("x_uploads_skipped") > 0, "skipped uploads: %lld", stat("x_uploads_skipped"));
	opt("x_upload", 1); opt("register_host", 0);
	// x_upload = 2: nobody vouches for anything; the library compares a checksum of ALL of the caller's x (threads of its own,
	// while the caller's thread goes on enqueueing) with the one it took of the device copy when the last call ended
	opt("x_upload", 2); opt("hash_threads", 3);
	for (int kind = 0; kind < 3; kind++) {
		stochqn_hip_stats_reset();
		Opt h(kind == 0 ? OLBFGS : (kind == 1 ? SQN : ADAQN), n + kind, 3, kind == 0 ? 1 : 3, false, 0.0, kind == 2 ? 4 : 0);
		h.drive(14);
		const long long skipped = stat("x_uploads_skipped"), sent = stat("x_uploads");
		CHECK(h.failed == 0 && skipped >= 3, "kind %d: uploads skipped on the strength of the checksum: %lld (sent %lld)", kind, skipped, sent);
		while (h.section != 1) h.drive(1);
		h.x[(size_t) n / 3] += 1e-9;                      // ONE coordinate, nowhere near a probe: the caller's own move
		h.x_ref = h.x;
		h.answer();
		for (size_t i = 0; i < (size_t) h.n; i++) h.grad[i] = 0.25 * h.x[i] + 0.01;      // ... and its gradient there
		h.call(0.01);
		CHECK(stat("x_uploads") == sent + 1, "kind %d: an edit of one coordinate must send x up again (%lld -> %lld)", kind, sent, stat("x_uploads"));
		h.drive(6);
		// a rejected step: what went ahead is put right, and the sum that was taken is the untouched device x's
		while (h.section != 1) h.drive(1);
		fakelaunch::script().reject_step = true;
		h.drive(1);
		fakelaunch::script().reject_step = false;
		h.drive(8);
		CHECK(h.failed == 0, "kind %d: failed calls %d", kind, h.failed);
		stochqn_hip_release(h.key());
	}
	opt("x_upload", 1); opt("hash_threads", 0);
	leak_check("host_path");
}

// the checksum of a host buffer (option "x_upload" = 2): any partition and any number of threads give the sums of the
// definition (sqn_device.hpp: XHash), ragged ends included; one changed byte anywhere changes the first sum
void sc_xhash()
{
	std::vector<unsigned char> buf(70000);
	unsigned long long seed = 12345;
	for (auto& b : buf) { seed = seed * 6364136223846793005ull + 1442695040888963407ull; b = (unsigned char) (seed >> 56); }
	for (size_t bytes : {(size_t) 0, (size_t) 1, (size_t) 4, (size_t) 7, (size_t) 8, (size_t) 9, (size_t) 12, (size_t) 15, (size_t) 16, (size_t) 20,
	                     (size_t) 4096, (size_t) 4100, (size_t) 65536, (size_t) 65540, (size_t) 69996}) {
		// the definition, word by word
		unsigned long long a = 0, b = 0;
		const size_t words = (bytes + 7) / 8;
		for (size_t i = 0; i < words; i++) {
			unsigned long long w = 0;
			const size_t len = bytes - 8 * i < 8 ? bytes - 8 * i : 8;
			std::memcpy(&w, buf.data() + 8 * i, len);
			// the definition written out (sqn_device.hpp: XHash): splitmix64's finaliser over the word xor a position key
			unsigned long long z = w ^ ((unsigned long long) i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull);
			z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
			z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
			z ^= z >> 31;
			a += z;
			b += (2 * i + 1) * ((z << 32) | (z >> 32));
		}
		sqn::XHash whole;
		sqn::xhash_host(buf.data(), bytes, 0, sqn::xhash_words(bytes), &whole);
		CHECK(whole.a == a && whole.b == b, "%zu bytes in one piece: (%llx, %llx) against the definition's (%llx, %llx)", bytes, whole.a, whole.b, a, b);
		for (size_t cut : {(size_t) 0, (size_t) 1, words / 3, words / 2 + 1, words}) {
			if (cut > words) continue;
			sqn::XHash lo, hi;
			sqn::xhash_host(buf.data(), bytes, 0, cut, &lo);
			sqn::xhash_host(buf.data(), bytes, cut, words, &hi);
			CHECK(lo.a + hi.a == a && lo.b + hi.b == b, "%zu bytes cut at word %zu", bytes, cut);
		}
		for (int threads : {1, 2, 3, 7, 64}) {
			sqn::XHashJob* job = sqn::xhash_start(buf.data(), bytes, threads);
			CHECK(job != nullptr, "no job for %zu bytes on %d threads", bytes, threads);
			if (!job) continue;
			const sqn::XHash h = sqn::xhash_finish(job);
			CHECK(h.a == a && h.b == b, "%zu bytes on %d threads: (%llx, %llx) against (%llx, %llx)", bytes, threads, h.a, h.b, a, b);
		}
		for (size_t at : {(size_t) 0, bytes / 2, bytes ? bytes - 1 : 0}) {
			if (at >= bytes) continue;
			buf[at] ^= 0x10;
			sqn::XHash edited;
			sqn::xhash_host(buf.data(), bytes, 0, sqn::xhash_words(bytes), &edited);
			CHECK(edited.a != a, "%zu bytes: byte %zu changed, the first sum did not", bytes, at);
			buf[at] ^= 0x10;
		}
	}
	// the edits a plain sum of the words cannot see: the sign bits of an EVEN number of doubles flipped -- two
	// coordinates, every coordinate (x -> -x, even n) -- and two words exchanged.  Bit 63 of a word adds 2^63 to a linear first
	// sum and an odd multiple of 2^63 to a linear weighted one: two flips cancel in both.  Not here.
	for (size_t n : {(size_t) 2, (size_t) 64, (size_t) 1000, (size_t) 8750}) {
		std::vector<double> x(n);
		for (size_t i = 0; i < n; i++) x[i] = 0.25 + 1e-3 * (double) ((i * 2654435761u) % 1000);
		sqn::XHash h0;
		sqn::xhash_host(x.data(), 8 * n, 0, n, &h0);
		auto differs = [&](const char* what) {
			sqn::XHash h;
			sqn::xhash_host(x.data(), 8 * n, 0, n, &h);
			CHECK(h.a != h0.a && h.b != h0.b, "n = %zu: %s leaves a sum unchanged (%llx %llx)", n, what, h.a, h.b);
		};
		x[0] = -x[0]; x[n - 1] = -x[n - 1];
		differs("the signs of two coordinates flipped");
		x[0] = -x[0]; x[n - 1] = -x[n - 1];
		for (auto& v : x) v = -v;
		differs("x -> -x");
		for (auto& v : x) v = -v;
		std::swap(x[0], x[n / 2]);
		if (x[0] != x[n / 2]) differs("two coordinates exchanged");
		std::swap(x[0], x[n / 2]);
		for (auto& v : x) v *= 0.5;
		differs("x scaled by 1/2");
		for (auto& v : x) v *= 2.0;
		sqn::XHash back;
		sqn::xhash_host(x.data(), 8 * n, 0, n, &back);
		CHECK(back.a == h0.a && back.b == h0.b, "n = %zu: the sums of the restored vector differ", n);
	}
}

// rejected pairs (rollback = bak -> slot), NaN guards off, gradient differences, Fisher pairs with the function-value check
void sc_branches()
{
	Opt a(SQN, 3000, 3, 2, true, 1e-4);               // gradient differences + min_curvature
	a.drive(15);
	fakelaunch::script().reject_pair = true;
	a.drive(12);
	fakelaunch::script().reject_pair = false;
	a.drive(10);
	Opt b(ADAQN, 2500, 3, 3, false, 1e-4, 5, 1.01);
	b.drive(60);
	Opt c(ADAQN, 2500, 3, 3, true, 0.0, 0, 0.0);
	c.drive(40);
	Opt d(OLBFGS, 2000, 4, 1, false, 1e-3);
	d.check_nan = 0;
	d.drive(30);
	opt("threepass", 0);
	Opt e(SQN, 3000, 3, 2);
	e.drive(20);
	opt("threepass", 1);
	fakelaunch::script().sy = 1e-9;                    // nearly orthogonal pairs: the kappa rule sends the step to the sweeps
	Opt f(SQN, 3000, 3, 2);
	f.drive(20);
	fakelaunch::script().sy = 1.0;
	CHECK(stat("steps_kappa_fallback") > 0, "kappa fallbacks: %lld", stat("steps_kappa_fallback"));
	for (Opt* o : {&a, &b, &c, &d, &e, &f}) CHECK(o->failed == 0 && o->accepted > 5, "failed %d accepted %d", o->failed, o->accepted);
	opt("verify_cache", 1);
	Opt g(SQN, 3000, 3, 2);
	g.drive(20);
	opt("verify_cache", 0);
	leak_check("branches");
}

// library-owned workspaces (profile A) and the isolated entries
void sc_owned_and_raw()
{
	workspace_SQN* w = initialize_SQN(3000, 4, 3, 1e-4, 0, 0.0, 1, 1);
	CHECK(w != nullptr, "initialize_SQN");
	if (w) {
		std::vector<double> x(3000, 1.0), g(3000), hv(3000);
		double *req = x.data(), *req_vec = nullptr;
		task_enum task = calc_grad;
		info_enum info;
		for (int k = 0; k < 30; k++) {
			if (task == calc_hess_vec) for (int i = 0; i < 3000; i++) hv[(size_t) i] = req_vec[i];
			else for (int i = 0; i < 3000; i++) g[(size_t) i] = 0.1 * req[i];
			const int rc = run_SQN(0.01, x.data(), g.data(), hv.data(), &req, &req_vec, &task, w, &info);
			CHECK(rc == 0 || rc == 1, "run_SQN on an owned workspace: %d", rc);
		}
		dealloc_SQN(w);
	}
	workspace_adaQN* wa = initialize_adaQN(2000, 3, 5, 3, 1.01, 1e-4, 1e-4, 0.9, 0, 0.0, 1, 1);
	CHECK(wa != nullptr, "initialize_adaQN");
	dealloc_adaQN(wa);
	workspace_oLBFGS* wo = initialize_oLBFGS(2000, 3, 0.0, 0.0, 0.0, 1, 1);
	CHECK(wo != nullptr, "initialize_oLBFGS");
	dealloc_oLBFGS(wo);
	// isolated two-loop / take_step / Fisher product on host arrays
	const int n = 2000;
	std::vector<double> S(4 * n, 1.0), Y(4 * n, 2.0), g(n, 0.5), rho(4), alpha(4), x(n, 1.0), G(n, 0.1), H0(n), F(6 * n, 0.2), t(6), y(n);
	CHECK(stochqn_hip_two_loop(g.data(), n, nullptr, 0.0, Y.data(), S.data(), 4, 4, 1, rho.data(), alpha.data()) == 0, "two_loop");
	opt("raw_reuse_cache", 1);
	CHECK(stochqn_hip_two_loop(g.data(), n, nullptr, 0.0, Y.data(), S.data(), 4, 4, 1, rho.data(), alpha.data()) == 0, "two_loop, cached");
	opt("raw_reuse_cache", 0);
	bfgs_mem b{S.data(), Y.data(), rho.data(), alpha.data(), nullptr, nullptr, 4, 4, 2, 1, 0.0, 0.0};
	info_enum info;
	CHECK(stochqn_hip_take_step(0.01, n, x.data(), g.data(), &b, 0.9, H0.data(), 0.0, G.data(), 1e-4, 1, &info) == 0, "take_step");
	CHECK(stochqn_hip_fisher_product(F.data(), 6, n, g.data(), t.data(), y.data()) == 0, "fisher_product");
	CHECK(stochqn_hip_two_loop(nullptr, n, nullptr, 0.0, Y.data(), S.data(), 4, 4, 1, nullptr, nullptr) == -1000, "invalid input");
	leak_check("owned_and_raw");
}

void use_fake_rccl(void** handle)
{
	const char* lib = std::getenv("STOCHQN_HIP_RCCL_LIB");
	CHECK(lib && *lib, "STOCHQN_HIP_RCCL_LIB must name libfake_rccl_*.so");
	*handle = lib ? dlopen(lib, RTLD_NOW | RTLD_GLOBAL) : nullptr;
	CHECK(*handle != nullptr, "dlopen(%s): %s", lib ? lib : "?", dlerror());
}
template <class F> F sym(void* h, const char* name) { return h ? (F) dlsym(h, name) : nullptr; }

// single-process multi-device mode: P shard threads, each with its own context, stream and RCCL communicator (ncclCommInitAll)
void group_body(int P, bool virt)
{
	opt("devices", P); opt("devices_min_n", 1); opt("virtual_devices", virt ? 1 : 0);
	const int n = 4001;                               // not divisible by P: ragged shards
	{
		Opt a(SQN, n, 3, 4), b(ADAQN, n, 3, 3, false, 1e-4, 5, 1.01), c(OLBFGS, n, 3, 1), d(SQN, n, 3, 3, true, 1e-4);
		a.drive(30); b.drive(45); c.drive(30); d.drive(30);
		CHECK(stochqn_hip_devices_active(a.key()) == P, "shards of a: %d", stochqn_hip_devices_active(a.key()));
		CHECK(stochqn_hip_devices_reducer(a.key()) == (virt ? 3 : 1), "reducer kind %d", stochqn_hip_devices_reducer(a.key()));
		for (Opt* o : {&a, &b, &c, &d}) CHECK(o->failed == 0 && o->accepted >= 10, "kind %d: failed %d accepted %d", (int) o->kind, o->failed, o->accepted);
		CHECK(stochqn_hip_export(a.key()) == 0 && stochqn_hip_export(b.key()) == 0, "export of sharded state");
		stochqn_hip_invalidate(a.key());
		a.drive(10);
		while (a.section != 1) a.drive(1);
		fakelaunch::script().reject_step = true;
		a.drive(1);
		fakelaunch::script().reject_step = false;
		CHECK(a.info == search_direction_was_nan, "every shard rejects alike");
		a.drive(8);
		stochqn_hip_release(a.key());
		a.drive(6);                                  // re-imported from the caller's arrays
	}
	// the sliced host path per shard, each shard taking the checksum of its slice of the caller's x on threads of its own
	{
		opt("host_slice_min", 1 << 9); opt("x_upload", 2); opt("hash_threads", 2);
		stochqn_hip_stats_reset();
		Opt a(SQN, 40013, 3, 3), b(ADAQN, 40013, 3, 3, false, 0.0, 4);
		a.drive(20); b.drive(20);
		CHECK(a.failed == 0 && b.failed == 0 && a.accepted >= 8 && b.accepted >= 8, "failed %d %d accepted %d %d", a.failed, b.failed, a.accepted, b.accepted);
		CHECK(stat("x_uploads_skipped") >= 4 * P, "uploads skipped per shard on the strength of the checksum: %lld", stat("x_uploads_skipped"));
		while (a.section != 1) a.drive(1);
		a.x[17] -= 1e-7;                                  // one coordinate, in the first shard's slice
		a.x_ref = a.x;
		for (size_t i = 0; i < a.x.size(); i++) a.grad[i] = 0.25 * a.x[i] + 0.01;
		a.call(0.01);
		a.drive(6);
		CHECK(a.failed == 0, "after an edit: failed %d", a.failed);
		stochqn_hip_release(a.key()); stochqn_hip_release(b.key());
		opt("host_slice_min", 1 << 21); opt("x_upload", 1); opt("hash_threads", 0);
	}
	// library-owned sharded workspace (n beyond one device in real life)
	workspace_SQN* w = initialize_SQN(n, 3, 3, 0.0, 0, 0.0, 1, 1);
	CHECK(w != nullptr, "initialize_SQN in group mode");
	if (w) {
		std::vector<double> x((size_t) n, 1.0), g((size_t) n), hv((size_t) n);
		double *req = x.data(), *req_vec = nullptr;
		task_enum task = calc_grad;
		info_enum info;
		for (int k = 0; k < 24; k++) {
			if (task == calc_hess_vec) for (int i = 0; i < n; i++) hv[(size_t) i] = req_vec[i];
			else for (int i = 0; i < n; i++) g[(size_t) i] = 0.1 * req[i];
			const int rc = run_SQN(0.01, x.data(), g.data(), hv.data(), &req, &req_vec, &task, w, &info);
			CHECK(rc == 0 || rc == 1, "run_SQN on an owned sharded workspace: %d", rc);
		}
		dealloc_SQN(w);
	}
	opt("devices", 0);
}

void sc_group_rccl()
{
	void* h = nullptr;
	use_fake_rccl(&h);
	fakehip::set_devices(4);
	group_body(4, false);
	auto init_all = sym<long (*)()>(h, "fake_rccl_init_all_calls");
	auto live = sym<long (*)()>(h, "fake_rccl_live_comms");
	auto reds = sym<long (*)()>(h, "fake_rccl_allreduces");
	CHECK(init_all && init_all() >= 5, "ncclCommInitAll ran %ld times", init_all ? init_all() : -1);
	CHECK(reds && reds() > 100, "all-reduces through the communicators: %ld", reds ? reds() : -1);
	leak_check("group_rccl");
	CHECK(live && live() == 0, "%ld communicators left behind", live ? live() : -1);
	// P = 1 over a real communicator: option "devices_rccl_single" (the path an 8-GPU node takes, on one device)
	fakehip::set_devices(1);
	opt("devices_rccl_single", 1);
	{
		Opt a(SQN, 3000, 3, 3);
		a.drive(25);
		CHECK(stochqn_hip_devices_active(a.key()) == 1 && stochqn_hip_devices_reducer(a.key()) == 1, "P = 1 group over RCCL: %d shards, reducer %d",
		      stochqn_hip_devices_active(a.key()), stochqn_hip_devices_reducer(a.key()));
		CHECK(a.failed == 0 && a.accepted >= 10, "failed %d accepted %d", a.failed, a.accepted);
	}
	opt("devices_rccl_single", 0);
	leak_check("group_rccl_single");
	CHECK(live && live() == 0, "%ld communicators left behind", live ? live() : -1);
}

void sc_group_virtual()
{
	fakehip::set_devices(1);
	group_body(3, true);
	leak_check("group_virtual");
}

// group mode with every allocation failing in turn: NULL / -1000, never a crash, a hang or a leak; then a clean run works
void sc_group_alloc_failures()
{
	void* h = nullptr;
	use_fake_rccl(&h);
	auto patience = sym<void (*)(int)>(h, "fake_rccl_set_patience_ms");
	if (patience) patience(300);
	opt("reducer_patience_s", 0.3);
	for (int virt = 0; virt < 2; virt++) {
		fakehip::set_devices(virt ? 1 : 4);
		const int P = virt ? 3 : 4;
		opt("devices", P); opt("devices_min_n", 1); opt("virtual_devices", virt);
		long k = 0, failed_runs = 0;
		int clean_in_a_row = 0;                       // some allocations may fail without harm (a pinned landing zone falls back to malloc)
		for (; clean_in_a_row < 10; k++) {
			opt("fail_alloc_after", (double) k);
			int worst = 1;
			{
				Opt a(SQN, 2003, 3, 3);
				a.check_x = false;                    // a failed call leaves x as it was handed in, the reference's contract for -1000
				worst = a.drive(12);
				stochqn_hip_release(a.key());
			}
			workspace_SQN* w = initialize_SQN(2003, 3, 3, 0.0, 0, 0.0, 1, 1);
			if (w) dealloc_SQN(w);
			opt("fail_alloc_after", -1);
			leak_check("group_alloc_failures");
			const bool clean = worst != -1000 && w != nullptr;
			clean_in_a_row = clean ? clean_in_a_row + 1 : 0;
			failed_runs += !clean;
			if (k > 600) { CHECK(false, "the injection never stopped firing"); break; }
		}
		CHECK(failed_runs >= 5, "only %ld of the injected failures were felt", failed_runs);
		std::fprintf(stderr, "group_alloc_failures: %s reducer, P = %d: %ld injection points, %ld failed cleanly\n", virt ? "host-side" : "RCCL", P, k, failed_runs);
	}
	opt("devices", 0);
}

// EVERY call site of the HIP runtime failing in turn, single device: a message and -1000 (or a result), never a crash / leak
struct SweepStats { long runs = 0, failed_calls = 0, survived = 0; };

void sweep_one(const std::function<int()>& body, const char* where, SweepStats& st, long stride_after, long max_per_fn)
{
	// a clean run tells how often each entry point is called
	fakehip::reset();
	(void) body();
	long counts[fakehip::F_COUNT];
	for (int f = 0; f < fakehip::F_COUNT; f++) counts[f] = fakehip::calls(f);
	leak_check(where);
	for (int f = 0; f < fakehip::F_COUNT; f++) {
		long tried = 0;
		for (long nth = 1; nth <= counts[f] && tried < max_per_fn; nth += (nth < stride_after ? 1 : 1 + nth / 8), tried++) {
			fakehip::reset();
			fakehip::fail_nth(f, nth);
			g_tag = std::string(fakehip::fn_name(f)) + " #" + std::to_string(nth);
			const int rc = body();
			st.runs++;
			if (rc == -1000) st.failed_calls++; else st.survived++;
			CHECK(rc == 0 || rc == 1 || rc == -1000, "%s: %s call %ld failing: return code %d", where, fakehip::fn_name(f), nth, rc);
			char tag[160];
			std::snprintf(tag, sizeof tag, "%s, %s #%ld failing", where, fakehip::fn_name(f), nth);
			fakehip::fail_nth(f, 0);
			leak_check(tag, 2);
			g_tag.clear();
		}
	}
}

void sc_fault_sweep()
{
	SweepStats st;
	opt("register_min_bytes", 1 << 12);
	// (1) a host caller through two pair cycles, (2) the same under memory pressure: reclaim -> spill -> resume
	sweep_one([] {
		Opt a(SQN, 1 << 13, 3, 3);
		// (x is checked after every call that did not fail: a HIP failure that goes unnoticed would show here)
		int rc = a.drive(14);
		if (rc != -1000) rc = stochqn_hip_export(a.key()) == 0 ? rc : -1000;
		stochqn_hip_release(a.key());
		return rc;
	}, "fault_sweep/host_caller", st, 40, 120);
	sweep_one([] {
		const int n = 3000;
		Opt a(SQN, n, 4, 1000), b(SQN, n, 4, 1000);

		a.mem_used = 2; a.st = 2; a.niter = 5; a.section = 1;
		int rc = a.drive(3);
		const fakehip::Live one = fakehip::live();
		fakehip::set_capacity((size_t) one.device_bytes + (2 * 4 + 2) * (size_t) n * 8 / 2 + (size_t) (one.device_bytes - (2 * 4 + 2) * (long) n * 8));
		const int rb = b.drive(3);
		fakehip::set_capacity(0);
		const int ra = a.drive(3);
		if (rb == -1000 || ra == -1000) rc = -1000;
		stochqn_hip_release(a.key()); stochqn_hip_release(b.key());
		return rc;
	}, "fault_sweep/reclaim", st, 40, 120);
	sweep_one([] {
		Opt a(ADAQN, 2048, 3, 3, false, 1e-4, 4, 1.01);
		// (x is checked after every call that did not fail: a HIP failure that goes unnoticed would show here)
		const int rc = a.drive(30);
		stochqn_hip_release(a.key());
		return rc;
	}, "fault_sweep/adaqn", st, 30, 80);
	// (4) the SLICED host path (gradient and x in slices on the side streams, x ahead of the guard) with the checksum of x deciding
	// about its upload: the events, waits and copies of copy_stream / down_stream, the checksum kernel's read-back
	opt("host_slice_min", 1 << 10); opt("x_upload", 2); opt("hash_threads", 2);
	sweep_one([] {
		Opt a(SQN, (1 << 13) + 1, 3, 3);
		const int rc = a.drive(11);
		stochqn_hip_release(a.key());
		return rc;
	}, "fault_sweep/host_sliced_checksum", st, 24, 48);
	opt("host_slice_min", 1 << 21); opt("x_upload", 1); opt("hash_threads", 0);
	std::fprintf(stderr, "fault_sweep: %ld runs with one failing HIP call each: %ld ended in -1000, %ld completed\n", st.runs, st.failed_calls, st.survived);
	CHECK(st.runs > 300 && st.failed_calls > 50, "the sweep did not reach the call sites (%ld runs, %ld failed calls)", st.runs, st.failed_calls);
}

// the same with shard threads in the way: a shard that fails on its own must not leave the others waiting for ever
void sc_fault_sweep_group()
{
	void* h = nullptr;
	use_fake_rccl(&h);
	auto patience = sym<void (*)(int)>(h, "fake_rccl_set_patience_ms");
	if (patience) patience(250);
	opt("reducer_patience_s", 0.25);
	opt("register_min_bytes", 1 << 12);
	SweepStats st;
	for (int virt = 0; virt < 2; virt++) {
		fakehip::set_devices(virt ? 1 : 2);
		opt("devices", 2); opt("devices_min_n", 1); opt("virtual_devices", virt);
		sweep_one([] {
			Opt a(SQN, 4099, 3, 3);
			a.check_x = false;
			const int rc = a.drive(10);
			stochqn_hip_release(a.key());
			return rc;
		}, virt ? "fault_sweep_group/virtual" : "fault_sweep_group/rccl", st, 6, 14);
	}
	opt("devices", 0);
	std::fprintf(stderr, "fault_sweep_group: %ld runs: %ld ended in -1000, %ld completed\n", st.runs, st.failed_calls, st.survived);
	CHECK(st.runs > 100, "the sweep did not reach the call sites (%ld runs)", st.runs);
}

// two caller threads, each with its own objects, while a third exports and releases: the registry under contention
void sc_threads()
{
	opt("max_mirror_bytes", (double) ((2 * 3 + 2) * 2000 * 8 * 3));
	std::vector<std::thread> ts;
	for (int t = 0; t < 3; t++)
		ts.emplace_back([t] {
			Opt a(t == 1 ? OLBFGS : SQN, 2000, 3, 3), b(SQN, 2000, 3, 2);
			for (int r = 0; r < 6; r++) {
				a.drive(7); b.drive(5);
				if (stochqn_hip_export(a.key()) != 0) { /* reclaimed meanwhile: its spill was exported instead */ }
				if (r == 3) stochqn_hip_release(b.key());
			}
			CHECK(a.failed == 0 && b.failed == 0, "thread %d: failed %d %d", t, a.failed, b.failed);
			stochqn_hip_release(a.key()); stochqn_hip_release(b.key());
		});
	for (auto& t : ts) t.join();
	opt("max_mirror_bytes", 0);
	leak_check("threads");
}

// A caller whose arrays live in a garbage-collected heap (reference src/Rwrapper.c:106-123: SEXPs the R collector may move or
// free between two .Call()s; stochqn/pywrapper.pxi:161-172: numpy arrays): x, the gradient and the Hessian-vector array are
// pinned by their owner (the binding), used for a few calls, then REPLACED -- unpinned, freed, a new array elsewhere -- between
// two calls, as `x = x.copy()` or `optimizer$gradient <- g` do.  The invariant of the host path: when run_* returns, no
// operation that touches the caller's memory is queued on any stream (fakehip::pending_host_ops, the per-stream model) and the
// library's own check agrees ("host_copies_in_flight", hipStreamQuery); an array may therefore die the moment the call is back.
// fake_hip reports a violation if a range dies under a queued copy, is freed while registered, or is copied through afterwards.
struct Heap {
	std::vector<std::pair<double*, size_t>> pinned, graveyard;       // dead arrays keep their addresses to themselves until the object is gone:
	                                                                 // whoever goes through one afterwards is caught, nobody is caught by a reused address
	double* fresh(size_t count, const double* from, bool pin)
	{
		double* p = static_cast<double*>(std::malloc(count * sizeof(double)));
		fakehip::host_range_lives(p, count * sizeof(double));
		if (from) std::memcpy(p, from, count * sizeof(double)); else std::memset(p, 0, count * sizeof(double));
		if (pin && stochqn_hip_pin_host(p, count * sizeof(double)) == 0) pinned.emplace_back(p, count);
		return p;
	}
	void drop(double* p, size_t count)
	{
		for (size_t i = 0; i < pinned.size(); i++)
			if (pinned[i].first == p) { CHECK(stochqn_hip_unpin_host(p) == 0, "unpin"); pinned.erase(pinned.begin() + (long) i); break; }
		fakehip::host_range_dies(p, count * sizeof(double));
		std::memset(p, 0x5a, count * sizeof(double));            // what the allocator's next customer writes there
#if defined(__SANITIZE_ADDRESS__)
		__asan_poison_memory_region(p, count * sizeof(double));  // a read by the library's own threads (probe values, the checksum of x) shows too
#endif
		graveyard.emplace_back(p, count);
	}
	~Heap()
	{
		for (auto& d : graveyard) {
#if defined(__SANITIZE_ADDRESS__)
			__asan_unpoison_memory_region(d.first, d.second * sizeof(double));
#endif
			fakehip::host_range_lives(d.first, d.second * sizeof(double));
			std::free(d.first);
		}
	}
};

// ---- what the pinning rule says about glibc's own heaps (the PLAIN build only: the sanitizers bring allocators of their own) ------------------
// runtime.cpp: pinnable_in_place.  Every verdict is checked against what glibc itself says about the block in its chunk header
// (the word before the block: bit 1 = IS_MMAPPED, a mapping of its own; bit 2 = NON_MAIN_ARENA, a heap of a thread's arena).
void sc_glibc_heaps()
{
#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
	std::printf("note: glibc_heaps needs glibc's malloc; nothing checked in a sanitizer build\n");
#else
	const size_t bytes = 6u << 20;
	auto head = [](const void* p) { return ((const size_t*) p)[-1]; };
	auto declined = [&](void* p, const char* what) {
		const long long before = stat("host_pins_declined");
		CHECK(stochqn_hip_pin_host(p, bytes) == 1 && stat("host_pins_declined") == before + 1, "%s must be declined", what);
		CHECK(stochqn_hip_unpin_host(p) == -1 && fakehip::live().registered_ranges == 0, "%s: a declined range is not pinned", what);
	};
	auto pinned = [&](void* p, const char* what) {
		CHECK(stochqn_hip_pin_host(p, bytes) == 0 && fakehip::is_registered(p), "%s must be pinned", what);
		CHECK(stochqn_hip_unpin_host(p) == 0 && fakehip::live().registered_ranges == 0, "%s: unpin", what);
	};
	// above the mmap threshold: a mapping of its own, whichever thread asks
	mallopt(M_MMAP_THRESHOLD, 128 << 10);
	void* own = std::malloc(bytes);
	void* own_t = nullptr;
	std::thread([&] { own_t = std::malloc(bytes); }).join();
	CHECK(own && own_t && (head(own) & 2) && (head(own_t) & 2), "glibc should have mmap'ed these (%#zx, %#zx)", head(own), head(own_t));
	pinned(own, "the main thread's block above the mmap threshold");
	pinned(own_t, "a worker thread's block above the mmap threshold");
	// below it: the main thread's comes from the program break, a worker's from a 64 MiB heap of its arena
	mallopt(M_MMAP_THRESHOLD, 32 << 20);
	void* brk_blk = std::malloc(bytes);
	void* arena_blk = nullptr;
	void* arena_blk2 = nullptr;
	std::thread([&] { arena_blk = std::malloc(bytes); arena_blk2 = std::malloc(bytes); }).join();      // the second one is not at the start of the heap
	CHECK(brk_blk && !(head(brk_blk) & 6), "glibc should have taken this one from the break (%#zx)", head(brk_blk));
	declined(brk_blk, "a block of the break heap");
	if (arena_blk && arena_blk2 && (head(arena_blk) & 4) && (head(arena_blk2) & 4)) {
		for (int by_maps = 0; by_maps < 2; by_maps++) {          // the heap's header read through process_vm_readv, and looked up in /proc/self/maps
			sqn::g_pin_probe_by_maps = by_maps != 0;
			declined(arena_blk, "the first block of a thread arena's heap");
			declined(arena_blk2, "a later block of a thread arena's heap");
			pinned(own_t, "a worker thread's block above the mmap threshold, again");
		}
		sqn::g_pin_probe_by_maps = false;
		// a page-aligned piece of such a block is no better (posix_memalign inside a heap looks like this)
		void* inner = (void*) (((uintptr_t) arena_blk2 + 4095) & ~(uintptr_t) 4095);
		const long long before = stat("host_pins_declined");
		CHECK(stochqn_hip_pin_host(inner, bytes / 2) == 1 && stat("host_pins_declined") == before + 1, "a page-aligned range inside a thread arena's heap must be declined");
	} else std::printf("note: this malloc gave the worker thread no arena of its own (%#zx); the thread-arena rule was not exercised\n", arena_blk ? head(arena_blk) : (size_t) 0);
	// a caller's own anonymous mapping that happens to begin on a 64 MiB boundary and holds ordinary data is not mistaken for one
	{
		const size_t span = (size_t) 128 << 20;
		char* raw = (char*) mmap(nullptr, span, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
		CHECK(raw != MAP_FAILED, "mmap");
		double* at = (double*) (((uintptr_t) raw + ((size_t) 64 << 20) - 1) & ~(((uintptr_t) 64 << 20) - 1));
		for (int i = 0; i < 64; i++) at[i] = 1.0 + i;                    // an array of doubles from the boundary on
		pinned(at + 8, "an array in the caller's own mapping, 64 bytes past a 64 MiB boundary");
		// ... and one whose 64 MiB boundary is not mapped at all (the probe must come back empty-handed, not fault)
		munmap(raw, (size_t) ((char*) at - raw) + 4096);
		for (int by_maps = 0; by_maps < 2; by_maps++) {
			sqn::g_pin_probe_by_maps = by_maps != 0;
			pinned((char*) at + 8192, "an array whose 64 MiB boundary is unmapped");
		}
		sqn::g_pin_probe_by_maps = false;
		munmap((char*) at + 4096, span - (size_t) ((char*) at - raw) - 4096);
	}
	std::free(own); std::free(own_t); std::free(brk_blk); std::free(arena_blk); std::free(arena_blk2);
	mallopt(M_MMAP_THRESHOLD, 128 << 10);
#endif
}

void sc_caller_heap()
{
	const int n = (1 << 16) + 3;
	const size_t N = (size_t) n;
	opt("host_slice_min", 1 << 12);
	opt("register_min_bytes", 1 << 12);
	struct Policy { const char* name; int x_upload, register_host, x_prefetch, pin; };
	const Policy policies[] = {{"defaults, arrays pinned by their owner", 1, 0, 0, 1}, {"defaults, nothing pinned", 1, 0, 0, 0},
	                           {"checksum of x", 2, 0, 0, 1}, {"vouched + the library pins + x prefetched", 0, 1, 1, 1}};
	for (const Policy& pol : policies)
		for (int strict = 0; strict < 2; strict++)
			for (int kind = 0; kind < 3; kind++) {
				g_tag = std::string("caller_heap / ") + pol.name + (strict ? " / strict_grad" : "") + " / kind " + std::to_string(kind);
				opt("x_upload", pol.x_upload); opt("register_host", pol.register_host); opt("x_prefetch", pol.x_prefetch); opt("strict_grad", strict);
				opt("hash_threads", 2);
				Heap heap;
				Opt a(kind == 0 ? OLBFGS : (kind == 1 ? SQN : ADAQN), n, 3, kind == 0 ? 1 : 3, false, 0.0, kind == 2 ? 4 : 0, kind == 2 ? 1.01 : 0.0);
				// the arrays that cross the link live in the caller's heap, not in the Opt object
				double* x = heap.fresh(N, a.x.data(), pol.pin);
				double* g = heap.fresh(N, nullptr, pol.pin);
				double* hv = heap.fresh(N, nullptr, pol.pin);
				double *req = x, *req_vec = nullptr;
				std::vector<double> x_ref(a.x);
				task_enum task = calc_grad;
				info_enum info = no_problems_encountered;
				size_t mem_used = 0, st = 0, niter = 0, f_used = 0, f_st = 0;
				int section = 0, failed = 0, steps = 0, replaced = 0;
				double f_prev = 0, f = 1.0;
				for (int call = 0; call < 44; call++) {
					// the caller answers the request where it was made ...
					if (task == calc_grad || task == calc_grad_same_batch || task == calc_grad_big_batch) for (size_t i = 0; i < N; i++) g[i] = 0.25 * req[i] + 0.01;
					else if (task == calc_hess_vec) for (size_t i = 0; i < N; i++) hv[i] = 2.0 * req_vec[i];
					else if (task == calc_fun_val_batch) { f = 0; for (size_t i = 0; i < N; i++) f += req[i] * req[i]; }
					// ... and its collector moves the arrays it owns: every few calls x, the gradient or hess_vec is a NEW array
					if (call % 5 == 3) { double* nx = heap.fresh(N, x, pol.pin); if (req == x) req = nx; heap.drop(x, N); x = nx; replaced++; }
					if (call % 7 == 2) { double* ng = heap.fresh(N, g, pol.pin); heap.drop(g, N); g = ng; replaced++; }
					if (call % 11 == 6) { double* nh = heap.fresh(N, hv, pol.pin); heap.drop(hv, N); hv = nh; replaced++; }
					if (call == 30) fakelaunch::script().reject_step = true;         // one step that the guard rejects: x that went ahead is put right
					const bool takes_step = section == 1;
					std::vector<double> g_in(g, g + (takes_step ? N : 0));
					bfgs_mem b{a.S.data(), a.Y.data(), a.rho.data(), a.alpha.data(), a.sbak.data(), a.ybak.data(), a.m, mem_used, st, a.kind == OLBFGS ? 1 : a.L, 0.0, 0.0};
					int rc;
					if (a.kind == OLBFGS) {
						workspace_oLBFGS w{&b, a.gprev.data(), 0.0, niter, section, 1, 1, n};
						rc = run_oLBFGS(0.01, x, g, &req, &task, &w, &info);
						niter = w.niter; section = w.section;
					} else if (a.kind == SQN) {
						workspace_SQN w{&b, a.gprev.data(), a.xsum.data(), a.xprev.data(), 0, niter, section, 1, 1, n};
						rc = run_SQN(0.01, x, g, hv, &req, &req_vec, &task, &w, &info);
						niter = w.niter; section = w.section;
					} else {
						fisher_mem fm{a.F.data(), a.fy.data(), a.fsize, f_used, f_st};
						workspace_adaQN w{&b, &fm, a.H0.data(), a.gprev.data(), a.xsum.data(), a.xprev.data(), a.G.data(), f_prev, a.max_incr, 1e-4, 0.9, 0, niter, section, 1, 1, n};
						rc = run_adaQN(0.01, x, f, g, &req, &task, &w, &info);
						niter = w.niter; section = w.section; f_prev = w.f_prev; f_used = fm.mem_used; f_st = fm.mem_st_ix;
					}
					mem_used = b.mem_used; st = b.mem_st_ix;
					fakelaunch::script().reject_step = false;
					if (rc == -1000) { failed++; break; }
					// THE INVARIANT: the call is back, nothing that touches the caller's memory is still queued anywhere
					const long queued = fakehip::pending_host_ops();
					const bool prefetching = pol.x_prefetch && stat("x_prefetched") > 0;
					CHECK(queued == 0 || (prefetching && queued == 1), "call %d: %ld copies through host memory still queued when the call returned", call, queued);
					if (takes_step && rc == 1 && info != func_increased) { steps++; for (size_t i = 0; i < N; i++) x_ref[i] -= 0.01 * g_in[i]; }
					if (info == func_increased) x_ref.assign(x, x + N);
					double worst = 0;
					for (size_t i = 0; i < N; i++) worst = std::fmax(worst, std::fabs(x[i] - x_ref[i]));
					CHECK(worst <= 1e-12, "call %d: the caller's x is %.3g away from x0 - sum step * grad", call, worst);
				}
				CHECK(failed == 0 && steps >= 8 && replaced >= 12, "failed %d, steps %d, arrays replaced %d", failed, steps, replaced);
				CHECK(stat("host_copies_in_flight") == 0, "the library's own check found %lld busy streams at the return of a call", stat("host_copies_in_flight"));
				// the object dies: the collector frees everything, in its own order, WITHOUT telling the library (R / Python never call dealloc_*)
				heap.drop(g, N); heap.drop(hv, N); heap.drop(x, N);
				CHECK(fakehip::violations() == 0, "%ld violations, last: %s", fakehip::violations(), fakehip::last_violation());
				// a new object finds a context under the same s_mem address: section 0 drops it
			}
	g_tag.clear();
	CHECK(stat("x_sent_ahead") > 0 && stat("x_sent_again") > 0 && stat("x_prefetched") > 0 && stat("x_uploads_skipped") > 0,
	      "the paths this scenario is for: x ahead of the guard %lld, sent again %lld, prefetched %lld, uploads skipped %lld",
	      stat("x_sent_ahead"), stat("x_sent_again"), stat("x_prefetched"), stat("x_uploads_skipped"));
	opt("x_upload", 1); opt("register_host", 0); opt("x_prefetch", 0); opt("strict_grad", 0); opt("hash_threads", 0);
	leak_check("caller_heap");
}

// One process per GPU (stochqn_hip_comm_init -> ncclCommInitRank, the mode bench.py and every torch.distributed caller uses): three
// ranks = three forked processes over the stand-in's shared-memory clique, whose all-reduces are operations ON THE STREAM that
// complete only when every rank has posted -- like the real thing, a collective whose peer never comes does not return an error,
// it waits.  The ranks' contributions arrive with rank-dependent delays, and rank 1's 17th all-reduce FAILS at the call.  Required:
// rank 1 returns -1000 from that call; ranks 0 and 2, which are waiting on their streams for a contribution that will never be
// posted, give up after reducer_patience_s (runtime.cpp: wait_stream aborts the communicator) and return -1000 FROM THE SAME
// CALL; every later call of every rank fails at once (the communicator is dead); nobody hangs.
// can_abort = false (scenario model_a_no_abort, with libfake_rccl_noabort_*.so): the same story over an RCCL that does not export
// ncclCommAbort -- nothing can end the collectives that wait.  The waiting ranks must STILL return -1000 from that call after
// the patience (not synchronise on a stream that can never drain: ADVICE r05), fail every later call at once, and get through
// stochqn_hip_release_all / stochqn_hip_comm_finalize without hanging: their contexts are abandoned (wedged), nothing of them is
// freed under the kernels that still wait, the dead communicator is not destroyed.
void model_a(bool can_abort)
{
	void* handle = nullptr;
	use_fake_rccl(&handle);
	CHECK((dlsym(handle, "ncclCommAbort") != nullptr) == can_abort, "this scenario wants an RCCL stand-in %s ncclCommAbort", can_abort ? "with" : "WITHOUT");
	auto shared_bytes = (size_t (*)(void)) dlsym(handle, "fake_rccl_shared_bytes");
	auto shared_init = (void (*)(void*, int)) dlsym(handle, "fake_rccl_shared_init");
	auto shared_script = (void (*)(int, int, long)) dlsym(handle, "fake_rccl_shared_script");
	CHECK(shared_bytes && shared_init && shared_script, "the RCCL stand-in lacks its multi-process controls");
	if (!shared_bytes || !shared_init || !shared_script) return;
	constexpr int P = 3, kCalls = 26;
	struct Result { int first_failure, failures_after, calls_done, check_failures; double seconds_of_failing_call, seconds_after; };
	const size_t bytes = shared_bytes() + P * sizeof(Result);
	char* mem = (char*) mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
	CHECK(mem != MAP_FAILED, "mmap");
	shared_init(mem, P);
	for (int r = 0; r < P; r++) shared_script(r, 1500 * r, r == 1 ? 17 : 0);       // rank r's contributions arrive 1.5 r ms late
	Result* res = reinterpret_cast<Result*>(mem + shared_bytes());
	for (int r = 0; r < P; r++) res[r] = Result{-1, 0, 0, 0, 0, 0};
	opt("reducer_patience_s", 1.5);
	std::fflush(nullptr);
	pid_t kids[P];
	for (int r = 0; r < P; r++) {
		kids[r] = fork();
		if (kids[r] == 0) {
			unsigned char id[128] = {0};
			if (stochqn_hip_comm_init(r, P, id) != 0 || stochqn_hip_comm_nranks() != P) _exit(90);
			Opt a(SQN, 1500 + 7 * r, 3, 3);               // device-resident or host caller alike: here a host caller, its shard of the problem
			a.check_x = true;
			for (int call = 0; call < kCalls; call++) {
				a.answer();
				const auto t0 = std::chrono::steady_clock::now();
				const int rc = a.call(0.01);
				const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
				res[r].calls_done = call + 1;
				if (rc == -1000 && res[r].first_failure < 0) { res[r].first_failure = call; res[r].seconds_of_failing_call = dt; }
				else if (res[r].first_failure >= 0) { res[r].failures_after += rc == -1000; res[r].seconds_after += dt; }
				if (res[r].first_failure >= 0 && call >= res[r].first_failure + 3) break;
			}
			res[r].check_failures = g_failures;
			stochqn_hip_release_all();
			stochqn_hip_comm_finalize();
			_exit(fakehip::violations() ? 91 : 0);
		}
		CHECK(kids[r] > 0, "fork");
	}
	// nobody may hang: the ranks are given ten times the patience, then killed
	const auto t0 = std::chrono::steady_clock::now();
	int alive = P, status[P];
	bool done[P] = {false, false, false};
	while (alive > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 60) {
		for (int r = 0; r < P; r++)
			if (!done[r] && waitpid(kids[r], &status[r], WNOHANG) == kids[r]) { done[r] = true; alive--; }
		std::this_thread::sleep_for(std::chrono::milliseconds(20));
	}
	for (int r = 0; r < P; r++)
		if (!done[r]) { kill(kids[r], SIGKILL); waitpid(kids[r], &status[r], 0); CHECK(false, "rank %d HUNG (call %d): killed after 60 s", r, res[r].calls_done); }
		else CHECK(WIFEXITED(status[r]) && WEXITSTATUS(status[r]) == 0, "rank %d ended with status %#x", r, status[r]);
	for (int r = 0; r < P; r++) {
		std::fprintf(stderr, "model_a_failure: rank %d: first -1000 at call %d (that call took %.2f s), %d of the %d calls after it failed too (%.3f s in all), x checks failed: %d\n",
		             r, res[r].first_failure, res[r].seconds_of_failing_call, res[r].failures_after, res[r].calls_done - res[r].first_failure - 1, res[r].seconds_after, res[r].check_failures);
		CHECK(res[r].first_failure >= 4, "rank %d: the run did not get going before the failure (first -1000 at call %d)", r, res[r].first_failure);
		CHECK(res[r].first_failure == res[1].first_failure, "rank %d failed in call %d, rank 1 in call %d: not the same call", r, res[r].first_failure, res[1].first_failure);
		CHECK(res[r].failures_after == 3 && res[r].seconds_after < 0.5, "rank %d: every later call must fail at once (%d of 3 failed, %.3f s)", r, res[r].failures_after, res[r].seconds_after);
		CHECK(res[r].check_failures == 0, "rank %d: %d checks failed before the failure", r, res[r].check_failures);
	}
	CHECK(res[1].seconds_of_failing_call < 0.5, "the rank whose own all-reduce failed knows at once (%.2f s)", res[1].seconds_of_failing_call);
	CHECK(res[0].seconds_of_failing_call > 1.0 && res[0].seconds_of_failing_call < 6.0 && res[2].seconds_of_failing_call > 1.0 && res[2].seconds_of_failing_call < 6.0,
	      "the waiting ranks give up after the patience (1.5 s): %.2f s, %.2f s", res[0].seconds_of_failing_call, res[2].seconds_of_failing_call);
	munmap(mem, bytes);
	opt("reducer_patience_s", 120);
	leak_check(can_abort ? "model_a_failure" : "model_a_no_abort");
}

void sc_model_a_failure() { model_a(true); }
void sc_model_a_no_abort() { model_a(false); }

struct Scenario { const char* name; void (*fn)(); };
const Scenario kScenarios[] = {
	{"registry", sc_registry}, {"reclaim_resume", sc_reclaim_resume}, {"mirror_cap", sc_mirror_cap}, {"host_path", sc_host_path}, {"xhash", sc_xhash},
	{"branches", sc_branches}, {"owned_and_raw", sc_owned_and_raw}, {"group_rccl", sc_group_rccl}, {"group_virtual", sc_group_virtual},
	{"group_alloc_failures", sc_group_alloc_failures}, {"fault_sweep", sc_fault_sweep}, {"fault_sweep_group", sc_fault_sweep_group},
	{"threads", sc_threads}, {"caller_heap", sc_caller_heap}, {"glibc_heaps", sc_glibc_heaps}, {"model_a_failure", sc_model_a_failure},
	{"model_a_no_abort", sc_model_a_no_abort}};

}  // namespace

int main(int argc, char** argv)
{
	if (argc < 2) {
		for (const auto& s : kScenarios) std::printf("%s\n", s.name);
		return 0;
	}
	const int model = argc > 2 && !std::strcmp(argv[2], "lazy") ? 1 : (argc > 2 && !std::strcmp(argv[2], "per_stream") ? 2 : 0);
	const char* const names[] = {"immediate", "lazy", "per_stream"};
	for (const auto& s : kScenarios)
		if (!std::strcmp(argv[1], s.name)) {
			defaults();
			fakehip::set_stream_model(model);
			s.fn();
			std::fprintf(stderr, "%s (%s streams): %s\n", s.name, names[model], g_failures ? "FAILED" : "ok");
			return g_failures ? 1 : 0;
		}
	std::fprintf(stderr, "unknown scenario %s\n", argv[1]);
	return 2;
}
