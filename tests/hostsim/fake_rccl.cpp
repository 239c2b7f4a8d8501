// fake_rccl.cpp -- the six RCCL entry points the library dlopen()s (runtime.cpp: load_rccl), as a host-side rendezvous
// of the calling threads (TEST INFRASTRUCTURE ONLY; built into libfake_rccl.so, named to the library through
// STOCHQN_HIP_RCCL_LIB).  With it the single-process multi-device mode takes its real path on the CPU -- ncclCommInitAll,
// one communicator per shard thread, ncclAllReduce on the shard's stream, ncclCommDestroy -- under the sanitizers.
#include "fake_hip.hpp"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <pthread.h>
#include <atomic>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct Clique {
	int n = 0;
	std::mutex mu;
	std::condition_variable cv;
	int arrived = 0;
	long generation = 0;
	bool broken = false;
	std::vector<const double*> send;
};

// ---- ranks in DIFFERENT processes (one process per GPU: stochqn_hip_comm_init / ncclCommInitRank with nranks > 1) -------------
// The clique lives in memory the test maps MAP_SHARED before it forks the ranks.  Here an all-reduce behaves like the real thing:
// the call returns at once, the collective is an operation ON THE STREAM that completes only when every rank has posted its
// contribution (fakehip::enqueue_waitable) -- a rank whose peer never posts waits on its stream for ever, unless its
// communicator is aborted.  Per-rank arrival delays and one rank's n-th all-reduce failing at call time are scripted.
constexpr int kMaxRanks = 8, kMaxCount = 128;
struct SharedClique {
	pthread_mutex_t mu;
	int n;
	long posted[kMaxRanks];                       // sequence number of the last collective rank r has posted
	long long ready_ns[kMaxRanks];                // ... and when its contribution "arrives" (post time + the rank's delay)
	double data[2][kMaxRanks][kMaxCount];
	int delay_us[kMaxRanks];
	long fail_seq[kMaxRanks];                     // rank r's fail_seq[r]-th all-reduce returns an error at call time (0 = never)
};
SharedClique* g_shared = nullptr;

long long now_ns()
{
	return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

std::mutex g_mu;
long g_calls = 0, g_fail_at = 0;                  // fault injection: the g_fail_at-th all-reduce (counted over all ranks) fails
int g_patience_ms = 20000;
long g_init_all = 0, g_live = 0;

bool barrier(Clique& q)
{
	std::unique_lock<std::mutex> lk(q.mu);
	if (q.broken) return false;
	const long gen = q.generation;
	if (++q.arrived == q.n) {
		q.arrived = 0;
		q.generation++;
		q.cv.notify_all();
		return true;
	}
	// (system clock: pthread_cond_timedwait, which gcc 11's libtsan intercepts -- it does not know pthread_cond_clockwait)
	if (!q.cv.wait_until(lk, std::chrono::system_clock::now() + std::chrono::milliseconds(g_patience_ms), [&] { return q.generation != gen || q.broken; }) || q.broken) {
		q.broken = true;
		q.cv.notify_all();
		return false;
	}
	return true;
}

}  // namespace

struct ncclComm {
	std::shared_ptr<Clique> clique;
	int rank = 0;
	SharedClique* shared = nullptr;               // ranks in other processes
	long seq = 0;
	std::atomic<bool> aborted{false};
};

extern "C" {

// test control (looked up with dlsym by tests/hostsim/host_logic_test.cpp)
void fake_rccl_fail_nth(long nth) { std::lock_guard<std::mutex> lk(g_mu); g_fail_at = nth > 0 ? g_calls + nth : 0; }
void fake_rccl_set_patience_ms(int ms) { std::lock_guard<std::mutex> lk(g_mu); g_patience_ms = ms; }
long fake_rccl_allreduces(void) { std::lock_guard<std::mutex> lk(g_mu); return g_calls; }
long fake_rccl_init_all_calls(void) { std::lock_guard<std::mutex> lk(g_mu); return g_init_all; }
long fake_rccl_live_comms(void) { std::lock_guard<std::mutex> lk(g_mu); return g_live; }
// the clique of a multi-process run: `mem` is MAP_SHARED memory of at least fake_rccl_shared_bytes() that the ranks' common
// ancestor mapped and zeroed before forking them
size_t fake_rccl_shared_bytes(void) { return sizeof(SharedClique); }
void fake_rccl_shared_init(void* mem, int nranks)
{
	SharedClique* q = static_cast<SharedClique*>(mem);
	std::memset(q, 0, sizeof *q);
	pthread_mutexattr_t a;
	pthread_mutexattr_init(&a);
	pthread_mutexattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
	pthread_mutex_init(&q->mu, &a);
	pthread_mutexattr_destroy(&a);
	q->n = nranks;
	g_shared = q;
}
void fake_rccl_shared_script(int rank, int delay_us, long fail_seq) { if (g_shared) { g_shared->delay_us[rank] = delay_us; g_shared->fail_seq[rank] = fail_seq; } }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
	std::memset(id, 0, sizeof *id);
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank)
{
	if (nranks > 1 && g_shared && nranks == g_shared->n && rank >= 0 && rank < nranks) {      // a rank of a clique of processes
		ncclComm* c = new ncclComm();
		c->rank = rank;
		c->shared = g_shared;
		*comm = c;
		std::lock_guard<std::mutex> lk(g_mu);
		g_live++;
		return ncclSuccess;
	}
	if (nranks != 1 || rank != 0) return ncclInvalidArgument;       // one process and no shared clique: a clique of several processes cannot form
	ncclComm* c = new ncclComm();
	c->clique = std::make_shared<Clique>();
	c->clique->n = 1;
	c->clique->send.assign(1, nullptr);
	*comm = c;
	std::lock_guard<std::mutex> lk(g_mu);
	g_live++;
	return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
	int have = 0;
	if (hipGetDeviceCount(&have) != hipSuccess || ndev < 1) return ncclInvalidArgument;
	for (int i = 0; i < ndev; i++) {
		const int d = devlist ? devlist[i] : i;
		if (d < 0 || d >= have) return ncclInvalidArgument;
		for (int j = 0; j < i; j++)
			if ((devlist ? devlist[j] : j) == d) return ncclInvalidUsage;    // RCCL refuses two ranks on one device
	}
	auto q = std::make_shared<Clique>();
	q->n = ndev;
	q->send.assign((size_t) ndev, nullptr);
	for (int i = 0; i < ndev; i++) {
		ncclComm* c = new ncclComm();
		c->clique = q;
		c->rank = i;
		comms[i] = c;
	}
	std::lock_guard<std::mutex> lk(g_mu);
	g_init_all++;
	g_live += ndev;
	return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
	if (!comm || dt != ncclDouble || op != ncclSum) return ncclInvalidArgument;
	bool fail = false;
	{
		std::lock_guard<std::mutex> lk(g_mu);
		g_calls++;
		if (g_fail_at > 0 && g_calls == g_fail_at) { g_fail_at = 0; fail = true; }
	}
	if (fail) return ncclInternalError;
	if (comm->shared) {
		SharedClique* q = comm->shared;
		const long seq = ++comm->seq;
		const int me = comm->rank;
		if (q->fail_seq[me] == seq) return ncclInternalError;       // this rank's collective fails at the call: it never posts, its peers wait
		const double* in = static_cast<const double*>(sendbuff);
		double* out = static_cast<double*>(recvbuff);
		if (count > (size_t) kMaxCount) return ncclInvalidArgument;
		std::shared_ptr<bool> posted_flag(new bool(false));
		fakehip::enqueue_waitable(stream, [=] {
			if (comm->aborted.load()) return true;                  // ncclCommAbort: the kernel that waited ends, its result is garbage
			pthread_mutex_lock(&q->mu);
			if (!*posted_flag) {                                    // in stream order: what the stream computed before is in `in`
				for (size_t i = 0; i < count; i++) q->data[seq & 1][me][i] = in[i];
				q->posted[me] = seq;
				q->ready_ns[me] = now_ns() + 1000ll * q->delay_us[me];
				*posted_flag = true;
			}
			bool all = true;
			const long long t = now_ns();
			for (int r = 0; r < q->n; r++) all = all && q->posted[r] >= seq && (q->posted[r] > seq || t >= q->ready_ns[r]);
			if (all)
				for (size_t i = 0; i < count; i++) {
					double sum = 0;
					for (int r = 0; r < q->n; r++) sum += q->data[seq & 1][r][i];      // rank order: every rank gets the same bits
					out[i] = sum;
				}
			pthread_mutex_unlock(&q->mu);
			return all;
		});
		return ncclSuccess;
	}
	if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;     // what the stream computed so far is final
	Clique& q = *comm->clique;
	{
		std::lock_guard<std::mutex> lk(q.mu);
		q.send[(size_t) comm->rank] = static_cast<const double*>(sendbuff);
	}
	if (!barrier(q)) return ncclInternalError;
	std::vector<double> sum(count, 0.0);
	for (int r = 0; r < q.n; r++)                     // rank order: every rank gets the same bits
		for (size_t i = 0; i < count; i++) sum[i] += q.send[(size_t) r][i];
	if (!barrier(q)) return ncclInternalError;       // nobody writes an in-place result before everybody has read
	std::memcpy(recvbuff, sum.data(), count * sizeof(double));
	return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
	if (!comm) return ncclInvalidArgument;
	delete comm;
	std::lock_guard<std::mutex> lk(g_mu);
	g_live--;
	return ncclSuccess;
}

// ends the collectives of this communicator that wait on their streams and releases it (the caller does not destroy it afterwards).
// -DFAKE_RCCL_NO_ABORT (libfake_rccl_noabort_*.so): an RCCL that does not export the symbol -- nothing can end a collective then.
#ifdef FAKE_RCCL_NO_ABORT
__attribute__((visibility("hidden")))
#endif
ncclResult_t ncclCommAbort(ncclComm_t comm)
{
	if (!comm) return ncclInvalidArgument;
	comm->aborted.store(true);
	if (!comm->shared) delete comm;                   // in-process clique: its collectives are synchronous, nothing queued refers to it
	                                                  // (a rank of a multi-process clique stays: operations still queued on a stream refer to it)
	std::lock_guard<std::mutex> lk(g_mu);
	g_live--;
	return ncclSuccess;
}

ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* state)
{
	if (!comm || !state) return ncclInvalidArgument;
	// reads the communicator, like the real one: on a communicator that ncclCommAbort has released the address sanitizer speaks up
	*state = comm->aborted.load() ? ncclInternalError : ncclSuccess;
	return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }

}  // extern "C"
