// fake_rccl.cpp -- the six RCCL entry points the library dlopen()s (runtime.cpp: load_rccl), as a host-side rendezvous
// of the calling threads (TEST INFRASTRUCTURE ONLY; built into libfake_rccl.so, named to the library through
// STOCHQN_HIP_RCCL_LIB).  With it the single-process multi-device mode takes its real path on the CPU -- ncclCommInitAll,
// one communicator per shard thread, ncclAllReduce on the shard's stream, ncclCommDestroy -- under the sanitizers.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

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
};

extern "C" {

// test control (looked up with dlsym by tests/hostsim/host_logic_test.cpp)
void fake_rccl_fail_nth(long nth) { std::lock_guard<std::mutex> lk(g_mu); g_fail_at = nth > 0 ? g_calls + nth : 0; }
void fake_rccl_set_patience_ms(int ms) { std::lock_guard<std::mutex> lk(g_mu); g_patience_ms = ms; }
long fake_rccl_allreduces(void) { std::lock_guard<std::mutex> lk(g_mu); return g_calls; }
long fake_rccl_init_all_calls(void) { std::lock_guard<std::mutex> lk(g_mu); return g_init_all; }
long fake_rccl_live_comms(void) { std::lock_guard<std::mutex> lk(g_mu); return g_live; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
	std::memset(id, 0, sizeof *id);
	return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank)
{
	if (nranks != 1 || rank != 0) return ncclInvalidArgument;       // one process here: a clique of several processes cannot form
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

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL error"; }

}  // extern "C"
