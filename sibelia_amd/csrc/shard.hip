// shard.hip -- hash-prefix sharded bifurcation enumeration over the GPUs of one node.
//
// The reference enumerates bifurcations on one CPU thread (IndexedSequence::EnumerateBifurcationsSArrayInRAM,
// reference src/vertexenumeration.cpp:263-364).  Here, with a communicator attached to the context, the k-mer
// table is sharded by HASH PREFIX over the GPUs (north_star): one process (or host thread) per GPU, SPMD.
//
//   A  scan      every GPU slides over its contiguous slice of tiles (halo from the replicated packed sequence)
//                into a LOCAL pre-aggregating table: one 16-B record per distinct canonical k-mer of the slice
//   B  exchange  records are bucketed by owner = hash prefix and shipped in ONE all-to-all (RCCL send/recv group:
//                7 peer messages per GPU, one per xGMI link)
//   C  classify  the owner ORs the masks of its k-mers and emits the strand-specific codes of the bifurcations
//   D  rank      the codes (8 B each, ~1.5 % of the k-mers) are all-gathered; every GPU radix-sorts the same
//                list => identical id tables, and builds a bifurcation-only lookup table (L2 resident)
//   E  resolve   every GPU resolves its slice against that table, compacts its marks and all-gathers them
//                (8 B per instance); the dense mark arrays end up complete and identical on every GPU.
// Simplification (globally ordered) then runs replicated.  k > 32 (exact rank doubling) is not sharded.
//
// Transports: RCCL (dlopen'ed librccl: grouped ncclSend/ncclRecv + ncclAllGather on the context's stream) and a
// local one (contexts of one process, one host thread each, device-to-device copies + a pthread barrier) that lets
// the tests run several virtual ranks on the single GPU of the test box.
#include <cstring>
#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <link.h>
#include <pthread.h>
#include <time.h>
#include <rocprim/rocprim.hpp>
#include <rccl/rccl.h>

#include "sbl_ctx.h"
#include "kmer_kernels.h"

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------- transports
struct SblComm {
	uint32_t rank = 0, n = 1;
	virtual ~SblComm() {}
	// small host payloads (counts): out = n x bytes
	virtual void allgather_host(sbl_ctx *c, const void *in, size_t bytes, void *out) = 0;
	// device buffers; byte counts / offsets per peer
	virtual void alltoallv(sbl_ctx *c, const char *send, const size_t *sbytes, const size_t *soff,
	                       char *recv, const size_t *rbytes, const size_t *roff) = 0;
	// this rank is leaving a collective call with an error: release peers that would wait for it (local transport)
	virtual void abort_peers() {}
};

struct RcclApi {
	void *h = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
	ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
	ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi load_rccl()
{
	RcclApi a;
	{
		// A librccl that is already in the process (the one PyTorch links against lives in torch/lib and is NOT what the
		// loader finds under the bare soname) must be the one used: two RCCL copies in one process corrupt the heap at exit.
		std::string loaded;
		dl_iterate_phdr([](struct dl_phdr_info *info, size_t, void *out) -> int {
			if (info->dlpi_name && strstr(info->dlpi_name, "librccl.so")) { *static_cast<std::string *>(out) = info->dlpi_name; return 1; }
			return 0;
		}, &loaded);
		if (!loaded.empty()) a.h = dlopen(loaded.c_str(), RTLD_NOW | RTLD_GLOBAL);
		if (!a.h)
			for (const char *nm : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) { a.h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (a.h) break; }
		if (a.h) {
#define SBL_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.h, name))
			SBL_SYM(GetUniqueId, "ncclGetUniqueId"); SBL_SYM(CommInitRank, "ncclCommInitRank"); SBL_SYM(CommDestroy, "ncclCommDestroy");
			SBL_SYM(AllGather, "ncclAllGather"); SBL_SYM(Send, "ncclSend"); SBL_SYM(Recv, "ncclRecv");
			SBL_SYM(GroupStart, "ncclGroupStart"); SBL_SYM(GroupEnd, "ncclGroupEnd"); SBL_SYM(GetErrorString, "ncclGetErrorString");
			SBL_SYM(CommAbort, "ncclCommAbort"); SBL_SYM(CommGetAsyncError, "ncclCommGetAsyncError");
#undef SBL_SYM
			if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllGather || !a.Send || !a.Recv || !a.GroupStart || !a.GroupEnd) a.h = nullptr;
		}
	}
	return a;
}
static RcclApi &rccl()
{
	static RcclApi a = load_rccl();       // thread-safe one-time initialisation
	return a;
}
#define RCCL_TRY(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { \
	char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, rccl().GetErrorString ? rccl().GetErrorString(r_) : "?", __FILE__, __LINE__); \
	throw SblError{SBL_ERR_HIP, b_}; } } while (0)

struct RcclComm : SblComm {
	ncclComm_t comm = nullptr;
	~RcclComm() override { if (comm) (void)rccl().CommDestroy(comm); }
	// A peer that fails inside a collective call (out of memory, a HIP error, ...) never posts its sends: a plain
	// hipStreamSynchronize would then block for ever.  The wait polls the stream, watches the communicator's asynchronous
	// error state and gives up after SBL_COMM_TIMEOUT_S seconds (default 600): the communicator is aborted and the call
	// fails with an error on THIS rank too, instead of hanging the job.
	void wait(sbl_ctx *c)
	{
		static const double limit = [] { const char *e = getenv("SBL_COMM_TIMEOUT_S"); double v = e ? atof(e) : 0; return v > 0 ? v : 600.0; }();
		const auto t0 = std::chrono::steady_clock::now();
		for (unsigned spin = 0;; spin++) {
			hipError_t e = hipStreamQuery(c->stream);
			if (e == hipSuccess) return;
			if (e != hipErrorNotReady) { abort_peers(); HIP_TRY(e); }
			if (spin > 2000) {
				if (rccl().CommGetAsyncError && comm) {
					ncclResult_t ar = ncclSuccess;
					if (rccl().CommGetAsyncError(comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress) {
						abort_peers();
						throw SblError{SBL_ERR_HIP, std::string("RCCL asynchronous error: ") + (rccl().GetErrorString ? rccl().GetErrorString(ar) : "?")};
					}
				}
				if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
					abort_peers();
					throw SblError{SBL_ERR_HIP, "a collective did not complete within SBL_COMM_TIMEOUT_S: a peer rank failed or left the call"};
				}
				struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr);
			}
		}
	}
	void allgather_host(sbl_ctx *c, const void *in, size_t bytes, void *out) override
	{
		SBL_CHECK(comm, SBL_ERR_HIP, "the RCCL communicator was aborted by an earlier failure");
		c->d_stage.ensure(bytes * (n + 1));
		char *d = c->d_stage.as<char>();
		HIP_TRY(hipMemcpyAsync(d, in, bytes, hipMemcpyHostToDevice, c->stream));
		RCCL_TRY(rccl().AllGather(d, d + bytes, bytes, ncclChar, comm, c->stream));
		HIP_TRY(hipMemcpyAsync(out, d + bytes, bytes * n, hipMemcpyDeviceToHost, c->stream));
		wait(c);
	}
	void alltoallv(sbl_ctx *c, const char *send, const size_t *sbytes, const size_t *soff, char *recv, const size_t *rbytes, const size_t *roff) override
	{
		SBL_CHECK(comm, SBL_ERR_HIP, "the RCCL communicator was aborted by an earlier failure");
		// one message per peer, all in flight together: xGMI is point to point, every link carries its own pair
		RCCL_TRY(rccl().GroupStart());
		for (uint32_t i = 0; i < n; i++) {
			uint32_t p = (rank + i) % n;
			if (sbytes[p]) RCCL_TRY(rccl().Send(send + soff[p], sbytes[p], ncclChar, (int)p, comm, c->stream));
			uint32_t q = (rank + n - i) % n;
			if (rbytes[q]) RCCL_TRY(rccl().Recv(recv + roff[q], rbytes[q], ncclChar, (int)q, comm, c->stream));
		}
		RCCL_TRY(rccl().GroupEnd());
		wait(c);
	}
	// this rank leaves a collective call with an error: tear the communicator down so that nothing of it stays queued here;
	// the peers notice through their own wait() (asynchronous error or timeout)
	void abort_peers() override
	{
		if (!comm) return;
		if (rccl().CommAbort) (void)rccl().CommAbort(comm); else (void)rccl().CommDestroy(comm);
		comm = nullptr;
	}
};

struct sbl_group {
	uint32_t n = 0;
	pthread_mutex_t mu;
	pthread_cond_t cv;
	uint32_t waiting = 0, generation = 0;
	bool failed = false;                      // a rank gave up inside a collective call: everybody else must not wait for it
	std::vector<const char *> send;
	std::vector<const size_t *> soff;
	std::vector<std::vector<uint8_t>> host;
	// barrier that can be broken: a rank that fails (out of memory, ...) wakes the others up instead of leaving them blocked
	void wait()
	{
		pthread_mutex_lock(&mu);
		if (!failed) {
			uint32_t gen = generation;
			if (++waiting == n) { waiting = 0; generation++; pthread_cond_broadcast(&cv); }
			else while (gen == generation && !failed) pthread_cond_wait(&cv, &mu);
		}
		bool f = failed;
		pthread_mutex_unlock(&mu);
		if (f) throw SblError{SBL_ERR_INTERNAL, "a peer rank of the local group failed inside a collective call"};
	}
	void fail()
	{
		pthread_mutex_lock(&mu);
		failed = true;
		pthread_cond_broadcast(&cv);
		pthread_mutex_unlock(&mu);
	}
};
struct LocalComm : SblComm {
	sbl_group *g = nullptr;
	void allgather_host(sbl_ctx *, const void *in, size_t bytes, void *out) override
	{
		g->host[rank].assign((const uint8_t *)in, (const uint8_t *)in + bytes);
		g->wait();
		for (uint32_t p = 0; p < n; p++) memcpy((char *)out + (size_t)p * bytes, g->host[p].data(), bytes);
		g->wait();
	}
	void alltoallv(sbl_ctx *c, const char *send, const size_t *, const size_t *soff, char *recv, const size_t *rbytes, const size_t *roff) override
	{
		HIP_TRY(hipStreamSynchronize(c->stream));             // my send buffer is complete
		g->send[rank] = send; g->soff[rank] = soff;
		g->wait();
		for (uint32_t p = 0; p < n; p++)
			if (rbytes[p]) HIP_TRY(hipMemcpyAsync(recv + roff[p], g->send[p] + g->soff[p][rank], rbytes[p], hipMemcpyDefault, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		g->wait();                                             // peers may reuse their send buffers
	}
	void abort_peers() override { g->fail(); }
};

void sbl_comm_release(sbl_ctx *c)
{
	if (!c->comm) return;
	delete c->comm;
	c->comm = nullptr;
}

extern "C" sbl_status sbl_comm_unique_id(void *id)
{
	if (!id) return SBL_ERR_BAD_ARG;
	static_assert(sizeof(ncclUniqueId) == SBL_COMM_ID_BYTES, "unique id size");
	if (!rccl().h) return SBL_ERR_UNSUPPORTED;
	ncclUniqueId u;
	if (rccl().GetUniqueId(&u) != ncclSuccess) return SBL_ERR_HIP;
	memcpy(id, &u, sizeof u);
	return SBL_OK;
}
extern "C" sbl_status sbl_comm_attach_rccl(sbl_ctx *c, uint32_t rank, uint32_t nranks, const void *id)
{
	return guarded(c, [&] {
		SBL_CHECK(id && nranks >= 1 && rank < nranks && nranks <= 64, SBL_ERR_BAD_ARG, "bad communicator arguments");
		SBL_CHECK(rccl().h, SBL_ERR_UNSUPPORTED, "librccl could not be loaded");
		sbl_comm_release(c);
		RcclComm *rc = new RcclComm;
		rc->rank = rank; rc->n = nranks;
		c->comm = rc;
		ncclUniqueId u;
		memcpy(&u, id, sizeof u);
		ncclResult_t r = rccl().CommInitRank(&rc->comm, (int)nranks, u, (int)rank);
		if (r != ncclSuccess) {                                   // no half-attached context: later stages run single-GPU
			rc->comm = nullptr;
			sbl_comm_release(c);
			RCCL_TRY(r);
		}
	});
}
extern "C" sbl_group *sbl_group_create_local(uint32_t nranks)
{
	if (nranks < 1 || nranks > 64) return nullptr;
	sbl_group *g = new sbl_group;
	g->n = nranks;
	pthread_mutex_init(&g->mu, nullptr);
	pthread_cond_init(&g->cv, nullptr);
	g->send.assign(nranks, nullptr); g->soff.assign(nranks, nullptr); g->host.resize(nranks);
	return g;
}
extern "C" void sbl_group_destroy(sbl_group *g)
{
	if (!g) return;
	pthread_cond_destroy(&g->cv);
	pthread_mutex_destroy(&g->mu);
	delete g;
}
extern "C" sbl_status sbl_comm_attach_local(sbl_ctx *c, sbl_group *g, uint32_t rank)
{
	return guarded(c, [&] {
		SBL_CHECK(g && rank < g->n, SBL_ERR_BAD_ARG, "bad group arguments");
		sbl_comm_release(c);
		LocalComm *lc = new LocalComm;
		lc->rank = rank; lc->n = g->n; lc->g = g;
		c->comm = lc;
	});
}
extern "C" sbl_status sbl_comm_detach(sbl_ctx *c)
{
	return guarded(c, [&] { sbl_comm_release(c); });
}

// ------------------------------------------------------------------------------------------- the SPMD pipeline
namespace {
struct Clock {
	double ms = 0;
	template <class F> void time(F f)
	{
		auto t0 = std::chrono::steady_clock::now();
		f();
		ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	}
};

// every rank contributes `sbytes` bytes; returns the total and fills recv in rank order
size_t allgatherv(sbl_ctx *c, Clock &clk, const char *send, size_t sbytes, DevBuf &recv, std::vector<size_t> &rbytes, std::vector<size_t> &roff)
{
	SblComm *cm = c->comm;
	uint32_t n = cm->n;
	std::vector<unsigned long long> mine(1, sbytes), all(n);
	clk.time([&] { cm->allgather_host(c, mine.data(), 8, all.data()); });
	rbytes.assign(n, 0); roff.assign(n, 0);
	size_t tot = 0;
	for (uint32_t p = 0; p < n; p++) { rbytes[p] = all[p]; roff[p] = tot; tot += all[p]; }
	recv.ensure(tot + 16);
	std::vector<size_t> sb(n, sbytes), so(n, 0);
	clk.time([&] { cm->alltoallv(c, send, sb.data(), so.data(), recv.as<char>(), rbytes.data(), roff.data()); });
	c->stats.exchange_bytes += sbytes * (n - 1);
	return tot;
}
}

static void run_enumeration_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity);
void sbl_run_enumeration_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	try { run_enumeration_sharded(c, k, elem_capacity); }
	catch (...) { c->comm->abort_peers(); throw; }
}
static void run_enumeration_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	SblComm *cm = c->comm;
	const uint32_t R = cm->n, r = cm->rank;
	hipStream_t s = c->stream;
	Clock clk;
	c->stats.exchange_bytes = 0;
	size_t E = c->nelem, nwords = (E + 31) / 32;
	size_t ntiles = (nwords + KM_TILE_WORDS - 1) / KM_TILE_WORDS;
	size_t t0 = ntiles * r / R, t1 = ntiles * (r + 1) / R;
	c->cur_k = k;
	sbl_pack(c);

	// ---- A: local pre-aggregation of this GPU's slice
	size_t slice = (t1 - t0) * (size_t)(KM_TILE_WORDS * 32);
	size_t cap = 1024;
	while (cap < slice + slice / 2) cap <<= 1;
	SBL_CHECK(cap <= 0xFFFFFFFFull, SBL_ERR_TOO_LARGE, "k-mer table too large for 32-bit slot indices");
	c->d_usedslots.ensure(slice * 4 + 64);
	c->d_table.ensure(cap * sizeof(KmerSlot));
	c->table_cap = cap;
	c->d_counters.ensure(256 * 4);
	unsigned *ctr = c->d_counters.as<unsigned>();      // [0..1] classify, [8] used (scan), [9] used (owner), [64..127] counts, [128..191] cursors
	k_table_init<<<(unsigned)std::min<size_t>((cap + 255) / 256, 256 * 16), 256, 0, s>>>(c->d_table.as<KmerSlot>(), cap);
	HIP_TRY(hipMemsetAsync(ctr, 0, 256 * 4, s));
	unsigned grid = (unsigned)std::max<size_t>(1, std::min<size_t>(t1 - t0, 256 * 8));
	HIP_TRY(hipEventRecord(c->ev[0], s));
	if (t1 > t0)
		k_kmer_table_build<<<grid, KM_THREADS, 0, s>>>(c->d_pk.as<unsigned long long>(), c->d_sp.as<unsigned>(), nwords, E, k,
		                                               c->d_table.as<KmerSlot>(), (unsigned long long)cap - 1, t0, t1, ctr + 8, c->d_usedslots.as<unsigned>());
	HIP_TRY(hipEventRecord(c->ev[1], s));
	HIP_TRY(hipGetLastError());
	unsigned nused = 0;
	HIP_TRY(hipMemcpyAsync(&nused, ctr + 8, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));

	// ---- B: bucket by owner (hash prefix) and exchange
	unsigned pgrid = (unsigned)std::max<size_t>(1, std::min<size_t>(((size_t)nused + 255) / 256, 256 * 16));
	std::vector<unsigned> cnt(R, 0), off(R, 0);
	if (nused) k_shard_count<<<pgrid, 256, 0, s>>>(c->d_table.as<KmerSlot>(), c->d_usedslots.as<unsigned>(), nused, R, ctr + 64);
	HIP_TRY(hipMemcpyAsync(cnt.data(), ctr + 64, R * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	for (uint32_t p = 1; p < R; p++) off[p] = off[p - 1] + cnt[p - 1];
	HIP_TRY(hipMemcpyAsync(ctr + 192, off.data(), R * 4, hipMemcpyHostToDevice, s));
	c->d_send.ensure((size_t)nused * sizeof(KmerRecord) + 16);
	if (nused) k_shard_scatter<<<pgrid, 256, 0, s>>>(c->d_table.as<KmerSlot>(), c->d_usedslots.as<unsigned>(), nused, R, ctr + 192, ctr + 128,
	                                                  c->d_send.as<KmerRecord>());
	HIP_TRY(hipGetLastError());
	std::vector<unsigned long long> scount(R), allcount((size_t)R * R);
	for (uint32_t p = 0; p < R; p++) scount[p] = cnt[p];
	clk.time([&] { cm->allgather_host(c, scount.data(), R * 8, allcount.data()); });
	std::vector<size_t> sb(R), so(R), rb(R), ro(R);
	size_t nrecv = 0;
	for (uint32_t p = 0; p < R; p++) {
		sb[p] = (size_t)cnt[p] * sizeof(KmerRecord); so[p] = (size_t)off[p] * sizeof(KmerRecord);
		size_t m = allcount[(size_t)p * R + r];
		rb[p] = m * sizeof(KmerRecord); ro[p] = nrecv * sizeof(KmerRecord);
		nrecv += m;
		if (p != r) c->stats.exchange_bytes += sb[p];
	}
	c->d_recv.ensure(nrecv * sizeof(KmerRecord) + 16);
	clk.time([&] { cm->alltoallv(c, c->d_send.as<char>(), sb.data(), so.data(), c->d_recv.as<char>(), rb.data(), ro.data()); });

	// ---- C: owner merge + classification
	size_t ocap = 1024;
	while (ocap < nrecv + nrecv / 2) ocap <<= 1;
	SBL_CHECK(ocap <= 0xFFFFFFFFull && nrecv < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "owner table too large for 32-bit slot indices");
	c->d_otable.ensure(ocap * sizeof(KmerSlot)); c->d_oused.ensure(nrecv * 4 + 64);
	k_table_init<<<(unsigned)std::min<size_t>((ocap + 255) / 256, 256 * 16), 256, 0, s>>>(c->d_otable.as<KmerSlot>(), ocap);
	if (nrecv) k_shard_merge<<<(unsigned)std::min<size_t>((nrecv + 255) / 256, 256 * 16), 256, 0, s>>>(c->d_recv.as<KmerRecord>(), nrecv, c->d_otable.as<KmerSlot>(),
	                                                                                                  (unsigned long long)ocap - 1, ctr + 9, c->d_oused.as<unsigned>());
	unsigned oused = 0;
	HIP_TRY(hipMemcpyAsync(&oused, ctr + 9, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	unsigned cgrid = (unsigned)std::max<size_t>(1, std::min<size_t>(((size_t)oused + 255) / 256, 256 * 16));
	k_classify_slots<<<cgrid, 256, 0, s>>>(c->d_otable.as<KmerSlot>(), c->d_oused.as<unsigned>(), oused, k, ctr, nullptr, nullptr, 0);
	unsigned pc[2];
	HIP_TRY(hipMemcpyAsync(pc, ctr, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	unsigned npairs = pc[0], mykeys = pc[1];
	c->d_keys.ensure((size_t)mykeys * 8 + 16); c->d_payload.ensure((size_t)mykeys * 4 + 16);
	HIP_TRY(hipMemsetAsync(ctr, 0, 8, s));
	k_classify_slots<<<cgrid, 256, 0, s>>>(c->d_otable.as<KmerSlot>(), c->d_oused.as<unsigned>(), oused, k, ctr,
	                                       c->d_keys.as<unsigned long long>(), c->d_payload.as<unsigned>(), npairs);
	HIP_TRY(hipGetLastError());

	// ---- D: gather the bifurcation codes, rank them identically everywhere
	std::vector<size_t> gb, go;
	size_t nkeys = allgatherv(c, clk, c->d_keys.as<char>(), (size_t)mykeys * 8, c->d_allkeys, gb, go) / 8;
	SBL_CHECK(nkeys < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many bifurcations");
	c->d_allkeys2.ensure(nkeys * 8 + 16);
	size_t bcap = 1024;
	while (bcap < 2 * nkeys) bcap <<= 1;
	c->d_otable.ensure(bcap * sizeof(KmerSlot));                 // the owner table is done: reuse it as the bifurcation-only table
	k_table_init<<<(unsigned)std::min<size_t>((bcap + 255) / 256, 256 * 16), 256, 0, s>>>(c->d_otable.as<KmerSlot>(), bcap);
	if (nkeys) {
		size_t tmp = 0;
		HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp, c->d_allkeys.as<unsigned long long>(), c->d_allkeys2.as<unsigned long long>(), nkeys, 0, 2 * k, s));
		c->d_sorttmp.ensure(tmp);
		HIP_TRY(rocprim::radix_sort_keys(c->d_sorttmp.p, tmp, c->d_allkeys.as<unsigned long long>(), c->d_allkeys2.as<unsigned long long>(), nkeys, 0, 2 * k, s));
		k_bif_table_build<<<nblocks(nkeys, 256), 256, 0, s>>>(c->d_allkeys2.as<unsigned long long>(), (unsigned)nkeys, k, c->d_otable.as<KmerSlot>(), (unsigned long long)bcap - 1);
	}
	c->bif_count = (uint32_t)nkeys;

	// ---- E: resolve my slice, compact, gather the marks, scatter them into the dense arrays
	for (int st = 0; st < 2; st++) {
		c->d_bif[st].ensure(elem_capacity * 4);
		HIP_TRY(hipMemsetAsync(c->d_bif[st].p, 0xFF, elem_capacity * 4, s));
	}
	if (t1 > t0 && nkeys)
		k_resolve_marks_bif<<<grid, KM_THREADS, 0, s>>>(c->d_pk.as<unsigned long long>(), c->d_sp.as<unsigned>(), nwords, E, k,
		                                                c->d_otable.as<KmerSlot>(), (unsigned long long)bcap - 1,
		                                                c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>(), t0, t1);
	HIP_TRY(hipGetLastError());
	for (int st = 0; st < 2; st++) {
		sbl_compact_marks(c, st);
		size_t mine = c->nmarks[st];
		size_t tot = allgatherv(c, clk, c->d_melem[st].as<char>(), mine * 4, c->d_gelem[st], gb, go) / 4;
		allgatherv(c, clk, c->d_mid[st].as<char>(), mine * 4, c->d_gid[st], gb, go);
		if (tot) k_scatter_marks<<<nblocks(tot, 256), 256, 0, s>>>(c->d_gelem[st].as<unsigned>(), c->d_gid[st].as<unsigned>(), tot, c->d_bif[st].as<unsigned>());
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));

	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
	c->stats.kmer_table_ms = ms;
	c->stats.exchange_ms = clk.ms;
	size_t positions = 0;
	for (uint32_t ch = 0; ch < c->nchr; ch++) {
		size_t len = c->sepidx[ch + 1] - c->sepidx[ch] - 1;
		if (len >= k) positions += len - k + 1;
	}
	c->stats.kmer_table_bytes = positions / R * 32 + E / 4;      // this GPU's slice: one 16-B slot read + written per base position
	c->stats.strand_kmers = 2 * positions;
	c->stats.bif_count = nkeys;
}
