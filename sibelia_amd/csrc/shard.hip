// shard.hip -- hash-prefix sharded bifurcation enumeration over the GPUs of one node.
//
// The reference enumerates bifurcations on one CPU thread (IndexedSequence::EnumerateBifurcationsSArrayInRAM,
// reference src/vertexenumeration.cpp:263-364).  Here, with a communicator attached to the context, the k-mer
// table is sharded by HASH PREFIX over the GPUs (north_star): one process (or host thread) per GPU, SPMD.
//
//   A  records   every GPU runs k_kmer_records over its contiguous slice of tiles (halo from the replicated packed sequence):
//                one 16-B record {mix64(canonical code), element | prev/next masks | orientation} per base POSITION of the
//                slice -- no local pre-aggregation (round 1 had one; it cost more than it saved) -- and partitions them by
//                the low `bits` bits of the key, the same hash prefix that buckets the single-GPU table
//   B  exchange  bucket b belongs to rank (b * R) >> bits, so what goes to one owner is ONE contiguous range of the
//                partitioned arrays: a single all-to-all of keys, then values (RCCL send/recv group: one message per peer
//                and array, every xGMI link carries its own pair); 16 B x positions of the slice x (R - 1) / R per GPU
//   C  classify  the owner partitions what it received again (R runs, each already bucket-sorted) and k_bucket_classify
//                builds the per-bucket LDS tables: bifurcation codes + member positions of its buckets; an 8-B flag
//                all-gather makes everybody re-bucket with a longer prefix if a bucket overflowed anywhere
//   D  rank      the codes (8 B each, ~1.5 % of the k-mers) are all-gathered; every GPU radix-sorts the same list =>
//                identical id tables; k_rank_own_keys: binary search of the owner's own codes = their ids
//   E  marks     k_member_marks turns the owner's member positions into (element, id) marks, which are all-gathered
//                (16 B per member position) and scattered: the dense mark arrays end up complete and identical everywhere.
// Simplification: commits replicated, read-only phases shared out (simplify.hip).  k > 32: sharded rank doubling (longk.hip).
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
#include "sbl_comm.h"
#include "kmer_kernels.h"
#include "kmer_bucket_kernels.h"

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------- transports
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

// ---- the layout of the sharded table and of its one exchange, as plain arithmetic (no device, no communicator): what the
// pipeline below uses, exported so that the tables can be checked for any number of ranks without GPUs (tests/test_shard_plan.py)
//   tiles     rank r scans tiles [ntiles * r / R, ntiles * (r + 1) / R)
//   buckets   owner p holds buckets [ceil(p * 2^bits / R), ceil((p + 1) * 2^bits / R)), i.e. owner(b) = (b * R) >> bits
extern "C" sbl_status sbl_shard_layout(uint32_t nranks, uint32_t rank, uint32_t bits, uint64_t ntiles, uint32_t *first_bucket /* nranks + 1 */, uint64_t *tile_range /* 2 */)
{
	if (!nranks || rank >= nranks || bits > 30 || !first_bucket || !tile_range) return SBL_ERR_BAD_ARG;
	const unsigned long long nb = 1ull << bits;
	for (uint32_t p = 0; p <= nranks; p++) first_bucket[p] = (uint32_t)(((unsigned long long)p * nb + nranks - 1) / nranks);
	tile_range[0] = ntiles * rank / nranks; tile_range[1] = ntiles * (rank + 1) / nranks;
	return SBL_OK;
}
//   exchange  count[p * R + q] = records rank p holds for owner q (all-gathered).  Rank `rank` sends owner q the records
//             [send_at[q], send_at[q + 1]) of its partitioned arrays and stores what p sends it behind what the ranks before p sent
extern "C" sbl_status sbl_shard_exchange_plan(uint32_t nranks, uint32_t rank, const uint64_t *count /* nranks x nranks */, const uint32_t *send_at /* nranks + 1 */,
                                              uint64_t record_bytes, uint64_t *sbytes, uint64_t *soff, uint64_t *rbytes, uint64_t *roff, uint64_t *nrecv)
{
	if (!nranks || rank >= nranks || !count || !send_at || !sbytes || !soff || !rbytes || !roff || !nrecv) return SBL_ERR_BAD_ARG;
	uint64_t got = 0;
	for (uint32_t p = 0; p < nranks; p++) {
		if (count[(size_t)rank * nranks + p] != (uint64_t)send_at[p + 1] - send_at[p]) return SBL_ERR_BAD_ARG;      // my own row must be what I partitioned
		sbytes[p] = count[(size_t)rank * nranks + p] * record_bytes; soff[p] = (uint64_t)send_at[p] * record_bytes;
		const uint64_t m = count[(size_t)p * nranks + rank];
		rbytes[p] = m * record_bytes; roff[p] = got * record_bytes;
		got += m;
	}
	*nrecv = got;
	return SBL_OK;
}

static void run_enumeration_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity);
void sbl_run_enumeration_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	try { run_enumeration_sharded(c, k, elem_capacity); }
	catch (...) { c->comm->abort_peers(); throw; }
}
static void run_enumeration_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	// The radix-bucketed table of the single-GPU path (kmer_bucket_kernels.h), cut along its hash prefix:
	//   A  every GPU turns ITS slice of tiles into k-mer records and partitions them by the low `bits` bits of the mixed key;
	//      bucket b belongs to rank (b * R) >> bits, so what goes to one owner is one contiguous range of the partitioned array
	//   B  ONE all-to-all of the 16-B records (keys, then values): grouped ncclSend / ncclRecv, one message per peer and array
	//   C  owners partition what they received again, build the per-bucket LDS tables, classify: bifurcation codes + member positions
	//   D  all-gather of the bifurcation codes, sorted identically everywhere: rank = id
	//   E  owners turn their member positions into (element, id) marks, all-gather, everybody scatters them into the dense arrays
	// Slices hold the same number of positions and buckets the same number of k-mers whatever the input: balanced by construction.
	SblComm *cm = c->comm;
	const uint32_t R = cm->n, r = cm->rank;
	hipStream_t s = c->stream;
	Clock clk;
	c->stats.exchange_bytes = 0;
	const size_t E = c->nelem, nwords = (E + 31) / 32;
	const size_t ntiles = (nwords + KM_TILE_WORDS - 1) / KM_TILE_WORDS;
	std::vector<unsigned> fb(R + 1);
	uint64_t trange[2];
	SBL_CHECK(sbl_shard_layout(R, r, 4, ntiles, fb.data(), trange) == SBL_OK, SBL_ERR_INTERNAL, "shard layout");
	const size_t t0 = trange[0], t1 = trange[1];
	const size_t nall = ntiles * (size_t)(KM_TILE_WORDS * 32), nmine = (t1 - t0) * (size_t)(KM_TILE_WORDS * 32);
	SBL_CHECK(nall < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many positions for 32-bit record indices");
	c->cur_k = k;
	sbl_pack(c);
	c->d_counters.ensure(256 * 4);
	unsigned *ctr = c->d_counters.as<unsigned>();
	unsigned bits = 4;
	while (bits < 30 && (nall >> bits) > KB_SLOTS * 9 / 16) bits++;              // the same on every rank: buckets are sized by the WHOLE input
	size_t maxpairs = nall / R / 8 + 4096;
	if (const char *e = getenv("SBL_TEST_BUCKET_BITS")) bits = std::min(bits, (unsigned)std::max(1, atoi(e)));      // test hooks, as in sbl_run_enumeration
	if (const char *e = getenv("SBL_TEST_MAXPAIRS")) maxpairs = (size_t)std::max(1, atoi(e));
	unsigned cnt[4] = {0, 0, 0, 0};
	size_t nrecv = 0;
	HIP_TRY(hipEventRecord(c->ev[0], s));
	for (int attempt = 0;; attempt++) {
		SBL_CHECK(attempt < 8, SBL_ERR_INTERNAL, "k-mer bucket classification did not converge");
		const size_t nb = (size_t)1 << bits;
		// ---- A: records of my slice, partitioned by hash prefix
		for (int i = 0; i < 2; i++) { c->d_rec_keys[i].ensure(nmine * 8 + 16); c->d_rec_vals[i].ensure(nmine * 8 + 16); }
		c->d_boff.ensure((nb + 1) * 4 + 64);
		std::vector<unsigned> send_at(R + 1, 0);
		SBL_CHECK(sbl_shard_layout(R, r, bits, ntiles, fb.data(), trange) == SBL_OK, SBL_ERR_INTERNAL, "shard layout");      // first bucket of every owner
		if (nmine) {
			k_kmer_records<<<(unsigned)std::min<size_t>(t1 - t0, 256 * 16), KM_THREADS, 0, s>>>(c->d_pk.as<unsigned long long>(), c->d_sp.as<unsigned>(), nwords, E, k, t0, t1,
			                                                                                  c->d_rec_keys[0].as<unsigned long long>(), c->d_rec_vals[0].as<unsigned long long>());
			size_t tmp = 0;
			HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, c->d_rec_keys[0].as<unsigned long long>(), c->d_rec_keys[1].as<unsigned long long>(),
			                                  c->d_rec_vals[0].as<unsigned long long>(), c->d_rec_vals[1].as<unsigned long long>(), nmine, 0, bits, s));
			c->d_sorttmp.ensure(tmp);
			HIP_TRY(rocprim::radix_sort_pairs(c->d_sorttmp.p, tmp, c->d_rec_keys[0].as<unsigned long long>(), c->d_rec_keys[1].as<unsigned long long>(),
			                                  c->d_rec_vals[0].as<unsigned long long>(), c->d_rec_vals[1].as<unsigned long long>(), nmine, 0, bits, s));
			k_bucket_bounds<<<nblocks(nb + 1, 256), 256, 0, s>>>(c->d_rec_keys[1].as<unsigned long long>(), nmine, bits, c->d_boff.as<unsigned>());
			HIP_TRY(hipGetLastError());
			for (uint32_t p = 0; p <= R; p++) HIP_TRY(hipMemcpyAsync(&send_at[p], c->d_boff.as<unsigned>() + fb[p], 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
		}
		// ---- B: the exchange
		std::vector<unsigned long long> scount(R), allcount((size_t)R * R);
		for (uint32_t p = 0; p < R; p++) scount[p] = send_at[p + 1] - send_at[p];
		clk.time([&] { cm->allgather_host(c, scount.data(), R * 8, allcount.data()); });
		std::vector<size_t> sb(R), so(R), rb(R), ro(R);
		{
			static_assert(sizeof(size_t) == sizeof(uint64_t), "64-bit host");
			uint64_t got = 0;
			SBL_CHECK(sbl_shard_exchange_plan(R, r, (const uint64_t *)allcount.data(), send_at.data(), 8, (uint64_t *)sb.data(), (uint64_t *)so.data(),
			                                  (uint64_t *)rb.data(), (uint64_t *)ro.data(), &got) == SBL_OK, SBL_ERR_INTERNAL, "exchange plan: the gathered counts contradict my own");
			nrecv = (size_t)got;
			for (uint32_t p = 0; p < R; p++) if (p != r) c->stats.exchange_bytes += 2 * sb[p];
		}
		SBL_CHECK(nrecv < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many records for one owner");
		c->d_recv.ensure(nrecv * 8 + 16); c->d_send.ensure(nrecv * 8 + 16);      // received keys / values
		clk.time([&] { cm->alltoallv(c, c->d_rec_keys[1].as<char>(), sb.data(), so.data(), c->d_recv.as<char>(), rb.data(), ro.data()); });
		clk.time([&] { cm->alltoallv(c, c->d_rec_vals[1].as<char>(), sb.data(), so.data(), c->d_send.as<char>(), rb.data(), ro.data()); });
		if (const char *e = getenv("SBL_TEST_FAIL_RANK")) if ((uint32_t)atoi(e) == r) throw SblError{SBL_ERR_OOM, "out of memory (SBL_TEST_FAIL_RANK: this rank leaves the collective enumeration)"};
		// ---- C: owner side: partition again (R runs, each sorted by bucket), per-bucket tables
		for (int i = 0; i < 2; i++) { c->d_otable.ensure(nrecv * 8 + 16); c->d_oused.ensure(nrecv * 8 + 16); }
		unsigned long long *ok = c->d_otable.as<unsigned long long>(), *ov = c->d_oused.as<unsigned long long>();
		if (nrecv) {
			size_t tmp = 0;
			HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, c->d_recv.as<unsigned long long>(), ok, c->d_send.as<unsigned long long>(), ov, nrecv, 0, bits, s));
			c->d_sorttmp.ensure(tmp);
			HIP_TRY(rocprim::radix_sort_pairs(c->d_sorttmp.p, tmp, c->d_recv.as<unsigned long long>(), ok, c->d_send.as<unsigned long long>(), ov, nrecv, 0, bits, s));
		}
		k_bucket_bounds<<<nblocks(nb + 1, 256), 256, 0, s>>>(ok, nrecv, bits, c->d_boff.as<unsigned>());
		unsigned long long *members = c->d_recv.as<unsigned long long>();          // the unsorted received keys are dead: their space holds the member list
		bool rebucket = false;
		for (;;) {
			c->d_keys.ensure(maxpairs * 16 + 16); c->d_payload.ensure(maxpairs * 8 + 16);
			HIP_TRY(hipMemsetAsync(ctr, 0, KB_CTR_WORDS * 4, s));
			k_bucket_classify<<<(unsigned)((nb + KB_GROUP - 1) / KB_GROUP), KB_THREADS, 0, s>>>(ok, ov, c->d_boff.as<unsigned>(), (unsigned)nb, k, ctr, c->d_keys.as<unsigned long long>(), c->d_payload.as<unsigned>(),
			                                                     (unsigned)maxpairs, members, (unsigned)nrecv);
			HIP_TRY(hipGetLastError());
			unsigned all[KB_CTR_WORDS];
			HIP_TRY(hipMemcpyAsync(all, ctr, sizeof all, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			cnt[0] = all[KB_CTR_PAIRS]; cnt[1] = all[KB_CTR_KEYS]; cnt[2] = all[KB_CTR_MEM]; cnt[3] = all[KB_CTR_FLAG];
			if (cnt[3] & 1u) { rebucket = true; break; }
			if (cnt[0] > maxpairs || (size_t)cnt[1] > 2 * maxpairs) { maxpairs = std::max<size_t>(cnt[0], ((size_t)cnt[1] + 1) / 2) + 1024; continue; }
			break;
		}
		// a bucket that overflowed anywhere makes everybody re-bucket with a longer prefix
		std::vector<unsigned long long> flag(1, rebucket ? 1 : 0), flags(R);
		clk.time([&] { cm->allgather_host(c, flag.data(), 8, flags.data()); });
		if (std::find(flags.begin(), flags.end(), 1ull) == flags.end()) break;
		// (every rank sees the same flags and the same `bits`, so all of them stop here together)
		SBL_CHECK(bits < 28, SBL_ERR_TOO_LARGE, "k-mer buckets keep overflowing at 2^28 buckets (adversarial key distribution)");
		bits = std::min(bits + 2, 28u);
	}
	HIP_TRY(hipEventRecord(c->ev[1], s));
	const unsigned npairs = cnt[0], mykeys = cnt[1], nmem = cnt[2];
	unsigned long long *members = c->d_recv.as<unsigned long long>();

	// ---- D: gather the bifurcation codes, rank them identically everywhere
	std::vector<size_t> gb, go;
	size_t nkeys = allgatherv(c, clk, c->d_keys.as<char>(), (size_t)mykeys * 8, c->d_allkeys, gb, go) / 8;
	SBL_CHECK(nkeys < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many bifurcations");
	c->d_allkeys2.ensure(nkeys * 8 + 16);
	c->d_pairids.ensure((size_t)npairs * 8 + 16);
	if (nkeys) {
		size_t tmp = 0;
		HIP_TRY(rocprim::radix_sort_keys(nullptr, tmp, c->d_allkeys.as<unsigned long long>(), c->d_allkeys2.as<unsigned long long>(), nkeys, 0, 2 * k, s));
		c->d_sorttmp.ensure(tmp);
		HIP_TRY(rocprim::radix_sort_keys(c->d_sorttmp.p, tmp, c->d_allkeys.as<unsigned long long>(), c->d_allkeys2.as<unsigned long long>(), nkeys, 0, 2 * k, s));
		if (mykeys)
			k_rank_own_keys<<<nblocks(mykeys, 256), 256, 0, s>>>(c->d_keys.as<unsigned long long>(), c->d_payload.as<unsigned>(), mykeys,
			                                                    c->d_allkeys2.as<unsigned long long>(), (unsigned)nkeys, k, c->d_pairids.as<unsigned>());
	}
	c->bif_count = (uint32_t)nkeys;
	c->dict_keys = c->d_allkeys2.as<unsigned long long>();

	// ---- E: marks of my buckets' member positions, gathered and scattered into the dense arrays everywhere
	for (int st = 0; st < 2; st++) {
		c->d_bif[st].ensure(elem_capacity * 4);
		HIP_TRY(hipMemsetAsync(c->d_bif[st].p, 0xFF, elem_capacity * 4, s));
		c->d_melem[st].ensure((size_t)nmem * 4 + 16); c->d_mid[st].ensure((size_t)nmem * 4 + 16);
	}
	if (nmem)
		k_member_marks<<<nblocks(nmem, 256), 256, 0, s>>>(members, nmem, k, c->d_pairids.as<unsigned>(), c->d_melem[0].as<unsigned>(), c->d_mid[0].as<unsigned>(),
		                                                 c->d_melem[1].as<unsigned>(), c->d_mid[1].as<unsigned>());
	HIP_TRY(hipGetLastError());
	for (int st = 0; st < 2; st++) {
		size_t tot = allgatherv(c, clk, c->d_melem[st].as<char>(), (size_t)nmem * 4, c->d_gelem[st], gb, go) / 4;
		allgatherv(c, clk, c->d_mid[st].as<char>(), (size_t)nmem * 4, c->d_gid[st], gb, go);
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
