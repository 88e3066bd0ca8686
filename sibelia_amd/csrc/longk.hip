// longk.hip -- bifurcation enumeration for vertex sizes k > 32 (stages (100,500), (500,1500), (1000,5000), (5000,15000)
// of the reference's parameter sets, reference src/util.cpp:52-87).
//
// Replaces EnumerateBifurcationsSArrayInRAM (reference src/vertexenumeration.cpp:263-364) without a suffix array and
// without fingerprints: EXACT rank doubling (Karp-Miller-Rosenberg).  rank_h[i] is the order-preserving dense rank of
// S[i..i+h) in the reference's superGenome S = "#c0#c1#..#rc(c0)#rc(c1)#..#" (:269-286); rank_2h comes from a radix
// sort of the pairs (rank_h[i], rank_h[i+h]); the k-windows are grouped and ordered by the pair
// (rank_h[i], rank_h[i+k-h]), h = largest power of two <= k (overlapping halves compare like the whole window).
// Groups in sorted order ARE the reference's lcp >= k runs in suffix-array order, so ids come out identical.
// Every step is a data-parallel kernel, a radix sort, a scan or a segmented OR -- HBM-streaming integer work.
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

#include "sbl_ctx.h"

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// S position -> (strand, element) and window validity.  Forward half: S[i] = element i.  Reverse half of chromosome c
// starts at E + sepidx[c] and holds the complement of elements sepidx[c+1]-1 downto sepidx[c]+1, then '#'.
__device__ __forceinline__ unsigned lk_chr_of(const unsigned *__restrict__ sepidx, unsigned nchr, unsigned e)
{
	unsigned lo = 0, hi = nchr;
	while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (sepidx[mid] < e) lo = mid; else hi = mid; }
	return lo;
}

__global__ void __launch_bounds__(256) k_lk_super(const uint8_t *__restrict__ ch, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                  unsigned E, unsigned n, unsigned np, unsigned *__restrict__ rank)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= np) return;
	unsigned v = 0;
	if (i < E) {
		uint8_t c = ch[i];
		v = c == 'A' ? 1 : c == 'C' ? 2 : c == 'G' ? 3 : c == 'T' ? 4 : 0;
	} else if (i < n) {
		unsigned r = i - E;                                   // offset inside the reverse half: block c = [sepidx[c], sepidx[c+1])
		unsigned c = lk_chr_of(sepidx, nchr, r + 1);          // r in [sepidx[c], sepidx[c+1]-1]  <=>  sepidx[c] < r+1 <= sepidx[c+1]
		unsigned j = r - sepidx[c];                           // 0 .. len_c ; j == len_c is the '#'
		unsigned len = sepidx[c + 1] - sepidx[c] - 1;
		if (j < len) {
			uint8_t x = ch[sepidx[c + 1] - 1 - j];
			v = x == 'A' ? 4 : x == 'C' ? 3 : x == 'G' ? 2 : x == 'T' ? 1 : 0;
		}
	}
	rank[i] = v;
}

// rb: bits of the largest rank of this round (ranks are dense: 5, 25, 625, ... distinct values in the first rounds), so that the sort
// only runs over the 2 rb bits that can differ
__global__ void __launch_bounds__(256) k_lk_pair_keys(const unsigned *__restrict__ rank, unsigned np, unsigned h, unsigned rb,
                                                      unsigned long long *__restrict__ keys, unsigned *__restrict__ idx)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= np) return;
	unsigned hi = rank[i], lo = (unsigned long long)i + h < np ? rank[i + h] : 0u;
	keys[i] = ((unsigned long long)hi << rb) | lo;
	idx[i] = i;
}
// rank[] back in position order WITHOUT a random scatter: (position, new rank) pairs are radix-sorted by position instead
// (k_lk_scatter_rank's 4-byte writes all over a 7-GB array were 61 ms per round at 1.8 G suffixes, a quarter of config 5)
__global__ void __launch_bounds__(256) k_lk_rank_values(const unsigned *__restrict__ scan, unsigned n, unsigned *__restrict__ val)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) val[j] = scan[j] - 1;
}
// The first three doubling rounds without a sort: the rank of the 8 symbols from i is their base-5 number (order-preserving, equal iff
// the strings are equal; positions past the end read '#' = 0, like the padding) -- 0 .. 390 624.
__global__ void __launch_bounds__(256) k_lk_rank8(const unsigned *__restrict__ sym, unsigned np, unsigned *__restrict__ rank)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= np) return;
	unsigned v = 0;
#pragma unroll
	for (unsigned j = 0; j < 8; j++) v = v * 5u + ((unsigned long long)i + j < np ? sym[i + j] : 0u);
	rank[i] = v;
}
// Round 5: the FIRST sorted round starts from 27 symbols instead of 2 x 8: their base-5 number fits 63 bits (5^27 < 2^63), is
// order-preserving and equal iff the 27 symbols are (positions past the end read '#' = 0, like the padding), so ONE 64-bit sort gives
// rank_27 -- where the 8-symbol start needed the rounds 8 -> 16 -> 32 (two sorts of all suffixes and two sorts back into position
// order) to get that far.  Unrelated / random sequence is unique after 27 symbols almost everywhere (config 5: the active set of the
// second round is the planted repeats), related genomes save one of their rounds.  sym[] holds one symbol per 32-bit word.
#define LK_FIRST_H 27u
__global__ void __launch_bounds__(256) k_lk_key27(const unsigned *__restrict__ sym, unsigned np, unsigned long long *__restrict__ keys, unsigned *__restrict__ idx)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= np) return;
	unsigned long long v = 0;
#pragma unroll
	for (unsigned j = 0; j < LK_FIRST_H; j++) v = v * 5ull + ((unsigned long long)i + j < np ? sym[i + j] : 0u);
	keys[i] = v;
	idx[i] = i;
}
__global__ void __launch_bounds__(256) k_lk_heads(const unsigned long long *__restrict__ skeys, unsigned n, unsigned *__restrict__ flag)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) flag[j] = (j == 0 || skeys[j] != skeys[j - 1]) ? 1u : 0u;
}
// rank[idx[j]] = (inclusive scan of head flags)[j] - 1
__global__ void __launch_bounds__(256) k_lk_scatter_rank(const unsigned *__restrict__ sidx, const unsigned *__restrict__ scan, unsigned n,
                                                         unsigned *__restrict__ rank)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) rank[sidx[j]] = scan[j] - 1;
}

// final keys of the k-windows on 2 rb bits (rb = bits of the largest rank); windows that contain a separator get the all-ones key of that
// width and sort to the end.  The number of valid windows is the metric's N = 2 sum(len - k + 1), known on the host: counting them here
// -- one atomic per wave on one address, 1.15 M of them on the bench workload -- was 13 ms of the 43 ms of kernels of a k = 100 enumeration.
__global__ void __launch_bounds__(256) k_lk_window_keys(const unsigned *__restrict__ rank, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                        unsigned E, unsigned n, unsigned k, unsigned h, unsigned rb,
                                                        unsigned long long *__restrict__ keys, unsigned *__restrict__ idx)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	bool valid = false;
	if (i < E) {
		unsigned c = lk_chr_of(sepidx, nchr, i);
		valid = i > sepidx[c] && (unsigned long long)i + k <= sepidx[c + 1];
	} else {
		unsigned r = i - E, c = lk_chr_of(sepidx, nchr, r + 1);
		unsigned j = r - sepidx[c], len = sepidx[c + 1] - sepidx[c] - 1;
		valid = (unsigned long long)j + k <= len;
	}
	const unsigned long long none = 2u * rb >= 64u ? ~0ull : (1ull << (2u * rb)) - 1ull;
	keys[i] = valid ? (((unsigned long long)rank[i] << rb) | rank[i + k - h]) : none;
	idx[i] = i;
}

// per sorted window: prev / next character masks (bit 0-3 = A C G T, bit 4 = '#'; next in bits 8-12) and the group-head flag
__global__ void __launch_bounds__(256) k_lk_masks(const unsigned long long *__restrict__ skeys, const unsigned *__restrict__ sidx, unsigned nv,
                                                  const unsigned *__restrict__ sym /* rank_1 = S */, unsigned k,
                                                  unsigned *__restrict__ mask, unsigned *__restrict__ flag)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nv) return;
	unsigned i = sidx[j];
	unsigned p = sym[i - 1], q = sym[i + k];                  // i >= 1 and i + k < n for every valid window
	mask[j] = (1u << (p ? p - 1 : 4)) | (1u << (8 + (q ? q - 1 : 4)));
	flag[j] = (j == 0 || skeys[j] != skeys[j - 1]) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_lk_group_bif(const unsigned *__restrict__ gmask, unsigned ngroups, unsigned *__restrict__ gbif)
{
	unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= ngroups) return;
	unsigned m = gmask[g], p = m & 0x1F, q = (m >> 8) & 0x1F;
	gbif[g] = ((p & 0x10) || (q & 0x10) || __popc(p & 0xF) > 1 || __popc(q & 0xF) > 1) ? 1u : 0u;
}
// marks: window at S position i -> bif[strand][element]
__global__ void __launch_bounds__(256) k_lk_marks(const unsigned *__restrict__ sidx, const unsigned *__restrict__ gscan /* inclusive scan of head flags */,
                                                  const unsigned *__restrict__ gbif, const unsigned *__restrict__ gid /* exclusive scan of gbif */, unsigned nv,
                                                  const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E,
                                                  unsigned *__restrict__ bif0, unsigned *__restrict__ bif1)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nv) return;
	unsigned g = gscan[j] - 1;
	if (!gbif[g]) return;
	unsigned i = sidx[j], id = gid[g];
	if (i < E) bif0[i] = id;
	else {
		unsigned r = i - E, c = lk_chr_of(sepidx, nchr, r + 1), jj = r - sepidx[c];
		bif1[sepidx[c + 1] - 1 - jj] = id;
	}
}

// ---- rank doubling over the ACTIVE suffixes only -------------------------------------------------------------------------------
// A suffix whose h-prefix is unique keeps its place in the order for good.  With ranks defined as "index of the group's first member
// in the sorted order" (not dense), refining one group never moves another, so a round only has to sort the suffixes that still
// share their prefix with somebody: new rank = old rank + (first index of the member's new subgroup - first index of its old
// group), both taken in the sorted array of the active suffixes, where a group is contiguous.  Random DNA is down to a third after
// h = 16 and to the planted repeats after h = 32; a set of related genomes stays mostly active (their shared stretches are the point).
__global__ void __launch_bounds__(256) k_lk_active_keys(const unsigned *__restrict__ rank, const unsigned *__restrict__ act, unsigned na, unsigned np, unsigned h, unsigned rb,
                                                        unsigned long long *__restrict__ keys, unsigned *__restrict__ idx)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= na) return;
	const unsigned i = act[j];
	const unsigned hi = rank[i], lo = (unsigned long long)i + h < np ? rank[i + h] : 0u;
	keys[j] = ((unsigned long long)hi << rb) | lo;
	idx[j] = i;
}
// gflag / sflag: own index where an old group / a new subgroup starts, else 0 (inputs of two running-maximum scans)
__global__ void __launch_bounds__(256) k_lk_heads2(const unsigned long long *__restrict__ skeys, unsigned na, unsigned rb, unsigned *__restrict__ gflag, unsigned *__restrict__ sflag)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= na) return;
	const unsigned long long a = skeys[j], b = j ? skeys[j - 1] : 0ull;
	gflag[j] = (j == 0 || (a >> rb) != (b >> rb)) ? j : 0u;
	sflag[j] = (j == 0 || a != b) ? j : 0u;
}
__global__ void __launch_bounds__(256) k_lk_newrank(const unsigned long long *__restrict__ skeys, const unsigned *__restrict__ sidx, const unsigned *__restrict__ gstart,
                                                    const unsigned *__restrict__ sstart, unsigned na, unsigned rb, int first, unsigned *__restrict__ rank,
                                                    unsigned *__restrict__ keep, unsigned *__restrict__ newrank_out /* non-null: (position, rank) pairs instead of the scatter */)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= na) return;
	const unsigned ss = sstart[j];
	const unsigned nr = first ? ss : (unsigned)(skeys[j] >> rb) + (ss - gstart[j]);      // a round over every suffix: the index IS the place in the order
	if (newrank_out) newrank_out[j] = nr; else rank[sidx[j]] = nr;
	const bool single = ss == j && (j + 1 == na || sstart[j + 1] == j + 1);
	keep[j] = single ? 0u : 1u;
}
// keys of candidate windows: the still-active suffixes and the windows next to a chromosome end (the only unique windows that can be
// bifurcations: Bifurcation() of a one-member group needs a '#' neighbour, vertexenumeration.cpp:67-70)
__global__ void __launch_bounds__(256) k_lk_cand_keys(const unsigned *__restrict__ rank, const unsigned *__restrict__ act, unsigned na, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                      unsigned E, unsigned k, unsigned h, unsigned long long *__restrict__ keys, unsigned *__restrict__ idx, unsigned *__restrict__ nvalid)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned total = na + 4u * nchr;
	if (j >= total) return;
	unsigned i; bool valid = false;
	if (j < na) {
		i = act[j];
		if (i < E) {
			unsigned c = lk_chr_of(sepidx, nchr, i);
			valid = i < 2u * E - 1u && i > sepidx[c] && (unsigned long long)i + k <= sepidx[c + 1];
		} else if (i < 2u * E - 1u) {
			unsigned r = i - E, c = lk_chr_of(sepidx, nchr, r + 1);
			unsigned jj = r - sepidx[c], len = sepidx[c + 1] - sepidx[c] - 1;
			valid = (unsigned long long)jj + k <= len;
		}
	} else {
		const unsigned t = j - na, c = t >> 2, which = t & 3u;
		const unsigned len = sepidx[c + 1] - sepidx[c] - 1;
		valid = len >= k;
		const unsigned off = (which & 1u) ? len - k : 0u;                 // first / last window of the strand
		i = (which & 2u) ? E + sepidx[c] + off : sepidx[c] + 1u + off;
		if (!valid) i = 0;
	}
	keys[j] = valid ? (((unsigned long long)rank[i] << 32) | rank[i + k - h]) : ~0ull;
	idx[j] = i;
	const unsigned long long m = __ballot(valid);
	if (m && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m)) atomicAdd(nvalid, (unsigned)__popcll(m));
}

struct LongKShardHolder;              // workspaces of the sharded variant (below)
struct LongKScratch {
	DevBuf rank[2], keys, skeys, idx, sidx, flag, scan, mask, gkeys, gmask, gcount, gbif, gid, tmp, sym, act, aux;
	LongKShardHolder *shard = nullptr;
};
static void lk_shard_free(LongKShardHolder *h);
static LongKScratch &lk_of(sbl_ctx *c)      // grow-only scratch owned by the context (contexts may live on different devices / host threads)
{
	if (!c->lk) c->lk = new LongKScratch;
	return *c->lk;
}
void sbl_longk_free(sbl_ctx *c)
{
	if (!c->lk) return;
	LongKScratch &L = *c->lk;
	for (DevBuf *b : { &L.rank[0], &L.rank[1], &L.keys, &L.skeys, &L.idx, &L.sidx, &L.flag, &L.scan, &L.mask, &L.gkeys, &L.gmask, &L.gcount, &L.gbif, &L.gid, &L.tmp, &L.sym, &L.act, &L.aux })
		b->release();
	lk_shard_free(L.shard);
	delete c->lk;
	c->lk = nullptr;
}

static void lk_sort(sbl_ctx *c, unsigned long long *kin, unsigned long long *kout, unsigned *vin, unsigned *vout, size_t n, unsigned bits = 64)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
	lk_of(c).tmp.ensure(tmp);
	HIP_TRY(rocprim::radix_sort_pairs(lk_of(c).tmp.p, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
}
static void lk_sort32(sbl_ctx *c, unsigned *kin, unsigned *kout, unsigned *vin, unsigned *vout, size_t n, unsigned bits)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
	lk_of(c).tmp.ensure(tmp);
	HIP_TRY(rocprim::radix_sort_pairs(lk_of(c).tmp.p, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
}
static unsigned lk_bits(unsigned long long v) { unsigned b = 1; while (b < 64 && (v >> b)) b++; return b; }
static void lk_inclusive_scan(sbl_ctx *c, unsigned *in, unsigned *out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::inclusive_scan(nullptr, tmp, in, out, n, rocprim::plus<unsigned>(), c->stream));
	lk_of(c).tmp.ensure(tmp);
	HIP_TRY(rocprim::inclusive_scan(lk_of(c).tmp.p, tmp, in, out, n, rocprim::plus<unsigned>(), c->stream));
}
static void lk_exclusive_scan(sbl_ctx *c, unsigned *in, unsigned *out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
	lk_of(c).tmp.ensure(tmp);
	HIP_TRY(rocprim::exclusive_scan(lk_of(c).tmp.p, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
}

static void lk_max_scan(sbl_ctx *c, unsigned *in, unsigned *out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::inclusive_scan(nullptr, tmp, in, out, n, rocprim::maximum<unsigned>(), c->stream));
	lk_of(c).tmp.ensure(tmp);
	HIP_TRY(rocprim::inclusive_scan(lk_of(c).tmp.p, tmp, in, out, n, rocprim::maximum<unsigned>(), c->stream));
}
static void lk_select(sbl_ctx *c, unsigned *in, unsigned *flags, unsigned *out, unsigned *count_out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::select(nullptr, tmp, in, flags, out, count_out, n, c->stream));
	lk_of(c).tmp.ensure(tmp);
	HIP_TRY(rocprim::select(lk_of(c).tmp.p, tmp, in, flags, out, count_out, n, c->stream));
}

struct BitOr { __host__ __device__ unsigned operator()(unsigned a, unsigned b) const { return a | b; } };

// Pre-flight: the workspaces of a long-k enumeration are 56 B per suffix plus the sort's double buffers (~100 GB for config 5's 1.8 G
// suffixes), allocated on the first call.  What is still missing is compared with what the device has free BEFORE the first
// allocation: a clear SBL_ERR_OOM instead of a failure half way through.  (SBL_TEST_FREE_MEM_MB: test switch, pretends less is free.)
static void lk_preflight(sbl_ctx *c, LongKScratch &L, size_t np)
{
	auto miss = [](const DevBuf &b, size_t want) { return want > b.cap ? want + want / 16 + 256 : (size_t)0; };
	size_t need = miss(L.rank[0], (np + 1) * 4) + miss(L.rank[1], (np + 1) * 4) + miss(L.sym, (np + 1) * 4) + miss(L.keys, np * 8) + miss(L.skeys, np * 8)
	            + miss(L.idx, np * 4) + miss(L.sidx, np * 4) + miss(L.flag, np * 4) + miss(L.scan, np * 4) + miss(L.mask, np * 4)
	            + miss(L.act, np * 4 + 64) + miss(L.aux, np * 4 + 64) + miss(L.tmp, np / 64 + (1u << 20));
	size_t fr = 0, tot = 0;
	if (hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); return; }
	if (const char *e = getenv("SBL_TEST_FREE_MEM_MB")) fr = (size_t)atoll(e) << 20;
	if (need > fr) {
		char b[200];
		snprintf(b, sizeof b, "long-k enumeration of %zu suffixes needs %zu MB of workspace, %zu MB free on the device", np, need >> 20, fr >> 20);
		throw SblError{SBL_ERR_OOM, b};
	}
	(void)c;
}

void sbl_run_enumeration_longk(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	hipStream_t s = c->stream;
	const size_t E = c->nelem, n = 2 * E - 1, np = n + k;             // 2L + 2 nchr + 1 = 2E - 1; padded with k '#'
	SBL_CHECK(np < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "input too large for 32-bit suffix ranks");
	c->cur_k = k;
	c->stats.exchange_bytes = 0; c->stats.exchange_ms = 0;             // (replicated: nothing leaves this GPU)
	LongKScratch &L = lk_of(c);
	lk_preflight(c, L, np);
	for (int t = 0; t < 2; t++) L.rank[t].ensure((np + 1) * 4);
	L.sym.ensure((np + 1) * 4);
	L.keys.ensure(np * 8); L.skeys.ensure(np * 8); L.idx.ensure(np * 4); L.sidx.ensure(np * 4);
	L.flag.ensure(np * 4); L.scan.ensure(np * 4); L.mask.ensure(np * 4);
	c->d_counters.ensure(64 * 4);
	HIP_TRY(hipMemsetAsync(c->d_counters.p, 0, 64 * 4, s));

	unsigned *rank = L.rank[0].as<unsigned>();
	k_lk_super<<<nblocks(np, 256), 256, 0, s>>>(c->d_ch.as<uint8_t>(), c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, (unsigned)n, (unsigned)np, rank);
	HIP_TRY(hipMemcpyAsync(L.sym.p, rank, np * 4, hipMemcpyDeviceToDevice, s));
	size_t h = 1;
	unsigned maxrank = 4;                                            // symbols 0 .. 4
	const bool first27 = k >= LK_FIRST_H && getenv("SBL_LONGK_FROM_1") == nullptr && getenv("SBL_LONGK_NO_DISCARD") == nullptr && getenv("SBL_LONGK_FROM_8") == nullptr;      // (SBL_LONGK_FROM_8: A/B switch, the round-4 start)
	if (k >= 16 && getenv("SBL_LONGK_FROM_1") == nullptr) {          // (k > 32 here: always; the switch is for A/B tests)
		rank = L.rank[1].as<unsigned>();
		if (!first27) {
			k_lk_rank8<<<nblocks(np, 256), 256, 0, s>>>(L.sym.as<unsigned>(), (unsigned)np, L.rank[1].as<unsigned>());
			h = 8; maxrank = 390624;
		}
	}
	const bool by_sort = np >= (1u << 22) && getenv("SBL_LONGK_SCATTER") == nullptr;      // small inputs: the scatter stays in cache
	const bool discard = k >= 16 && getenv("SBL_LONGK_FROM_1") == nullptr && getenv("SBL_LONGK_NO_DISCARD") == nullptr;
	unsigned nv = 0;                                                 // valid windows at the front of the sorted (skeys, sidx)
	bool plain = !discard;
	unsigned na_final = 0;
	if (discard) {
		L.act.ensure(np * 4 + 64); L.aux.ensure(np * 4 + 64);
		unsigned &na = na_final;
		na = (unsigned)np;
		bool first = true;
		const unsigned rbp = lk_bits(np);                            // ranks are indices into the sorted order from the first round on
		while (2 * h <= k && na) {
			// a round over EVERY suffix (sequential key construction, ranks back by a sort) while most of them are still active -- sets of
			// related genomes stay that way --, over the active ones only (two gathers and a scatter per suffix) once they are the minority
			const bool full = first || na > np / 2;
			const unsigned rb = first ? lk_bits(maxrank) : rbp;
			const unsigned m = full ? (unsigned)np : na;
			const bool from27 = first && first27;                   // the first round: rank_27 straight from the symbols (k_lk_key27), one 63-bit sort
			if (from27) k_lk_key27<<<nblocks(np, 256), 256, 0, s>>>(L.sym.as<unsigned>(), (unsigned)np, L.keys.as<unsigned long long>(), L.idx.as<unsigned>());
			else if (full) k_lk_pair_keys<<<nblocks(np, 256), 256, 0, s>>>(rank, (unsigned)np, (unsigned)h, rb, L.keys.as<unsigned long long>(), L.idx.as<unsigned>());
			else k_lk_active_keys<<<nblocks(na, 256), 256, 0, s>>>(rank, L.act.as<unsigned>(), na, (unsigned)np, (unsigned)h, rb, L.keys.as<unsigned long long>(), L.idx.as<unsigned>());
			lk_sort(c, L.keys.as<unsigned long long>(), L.skeys.as<unsigned long long>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), m, from27 ? 63u : std::min(64u, 2 * rb));
			k_lk_heads2<<<nblocks(m, 256), 256, 0, s>>>(L.skeys.as<unsigned long long>(), m, rb, L.flag.as<unsigned>(), L.scan.as<unsigned>());
			lk_max_scan(c, L.flag.as<unsigned>(), L.mask.as<unsigned>(), m);       // gstart
			lk_max_scan(c, L.scan.as<unsigned>(), L.aux.as<unsigned>(), m);        // sstart
			const bool pairs = full && by_sort;                      // everybody in the sort: the new ranks go back into position order by a sort
			k_lk_newrank<<<nblocks(m, 256), 256, 0, s>>>(L.skeys.as<unsigned long long>(), L.sidx.as<unsigned>(), L.mask.as<unsigned>(), L.aux.as<unsigned>(), m, rb, full ? 1 : 0,
			                                            rank, L.flag.as<unsigned>(), pairs ? L.scan.as<unsigned>() : nullptr);
			if (pairs) lk_sort32(c, L.sidx.as<unsigned>(), L.idx.as<unsigned>(), L.scan.as<unsigned>(), rank, m, lk_bits(np - 1));
			lk_select(c, L.sidx.as<unsigned>(), L.flag.as<unsigned>(), L.act.as<unsigned>(), c->d_counters.as<unsigned>() + 8, m);
			HIP_TRY(hipMemcpyAsync(&na, c->d_counters.as<unsigned>() + 8, 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			const bool was_first = first;
			first = false;
			h = from27 ? LK_FIRST_H : 2 * h;
			// decided once, after the first round: an input whose suffixes mostly still share their 16-prefix with somebody is a set of related
			// genomes -- it stays that way, and the plain doubling below (dense ranks: fewer key bits, no max-scans, no compaction) is 10 - 20 %
			// faster on it (8 x 4.6 Mbp, k = 100 / 500: 56 ms against 62 - 67 ms); the ranks so far are valid ranks for it, just not dense
			// (round 5: ... but only when ONE more doubling round is left, 4 h > k.  With more to come the active set of related genomes does
			// collapse on the way -- two strains with 1 % SNPs each share a 216-window 1.3 % of the time -- and the later rounds cost next to
			// nothing in this loop: k = 500 on 8 x 4.6 Mbp 50 -> 33 ms, k = 1000 56 -> 31 ms, where k = 100 is 47 ms plain against 50 ms.)
			// From the second round on the input has shown what it is: where more than 85 % of the suffixes are still active at h = 54 (the state a
			// cascade's earlier stages leave behind: the strains have been made alike over long stretches) the set is not going to collapse, and
			// the plain rounds are the cheaper ones again (config 3's k = 500 stage: 46 ms plain, 69 ms in this loop).
			if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] long k: h = %zu, %u of %zu suffixes still active\n", h, na, np);
			const bool sticky = (double)na > (was_first ? 0.98 : 0.85) * (double)np;      // (after ONE round only the unmistakable case: every suffix shares its 27-prefix)
			if (na > np / 2 && (4 * h > k || sticky) && getenv("SBL_LONGK_FORCE_ACTIVE") == nullptr) { plain = true; maxrank = (unsigned)(np - 1); break; }
		}
	}
	if (discard && !plain) {
		unsigned na = na_final;
		while (2 * h <= k) h *= 2;                                   // (everything unique before the last level: the offset of the second half only has to be valid)
		const unsigned ncand = na + 4u * c->nchr;
		k_lk_cand_keys<<<nblocks(ncand, 256), 256, 0, s>>>(rank, L.act.as<unsigned>(), na, c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, k, (unsigned)h,
		                                                  L.keys.as<unsigned long long>(), L.idx.as<unsigned>(), c->d_counters.as<unsigned>());
		lk_sort(c, L.keys.as<unsigned long long>(), L.skeys.as<unsigned long long>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), ncand);
		HIP_TRY(hipMemcpyAsync(&nv, c->d_counters.p, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
	}
	if (plain) {
		while (2 * h <= k) {
			const unsigned rb = lk_bits(maxrank);
			k_lk_pair_keys<<<nblocks(np, 256), 256, 0, s>>>(rank, (unsigned)np, (unsigned)h, rb, L.keys.as<unsigned long long>(), L.idx.as<unsigned>());
			lk_sort(c, L.keys.as<unsigned long long>(), L.skeys.as<unsigned long long>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), np, std::min(64u, 2 * rb));
			k_lk_heads<<<nblocks(np, 256), 256, 0, s>>>(L.skeys.as<unsigned long long>(), (unsigned)np, L.flag.as<unsigned>());
			lk_inclusive_scan(c, L.flag.as<unsigned>(), L.scan.as<unsigned>(), np);
			if (by_sort) {
				// (sidx, scan - 1) sorted by sidx = rank[] in position order; L.flag / L.idx are free at this point
				k_lk_rank_values<<<nblocks(np, 256), 256, 0, s>>>(L.scan.as<unsigned>(), (unsigned)np, L.flag.as<unsigned>());
				lk_sort32(c, L.sidx.as<unsigned>(), L.idx.as<unsigned>(), L.flag.as<unsigned>(), rank, np, lk_bits(np - 1));
			} else
				k_lk_scatter_rank<<<nblocks(np, 256), 256, 0, s>>>(L.sidx.as<unsigned>(), L.scan.as<unsigned>(), (unsigned)np, rank);
			HIP_TRY(hipMemcpyAsync(&maxrank, L.scan.as<unsigned>() + (np - 1), 4, hipMemcpyDeviceToHost, s));      // number of distinct 2h-prefixes
			HIP_TRY(hipStreamSynchronize(s));
			h *= 2;
		}
		const unsigned rbw = lk_bits((unsigned long long)maxrank + 1);   // (+ 1: the all-ones key of that width stays above every valid key)
		k_lk_window_keys<<<nblocks(n, 256), 256, 0, s>>>(rank, c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, (unsigned)n, k, (unsigned)h, rbw,
		                                                L.keys.as<unsigned long long>(), L.idx.as<unsigned>());
		lk_sort(c, L.keys.as<unsigned long long>(), L.skeys.as<unsigned long long>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), n, std::min(64u, 2 * rbw));
		{	// valid windows = the metric's N, at the front of the sorted order (the invalid ones carry the largest key)
			unsigned long long N = 0;
			for (uint32_t ch = 0; ch < c->nchr; ch++) { const size_t len = c->sepidx[ch + 1] - c->sepidx[ch] - 1; if (len >= k) N += 2 * (len - k + 1); }
			nv = (unsigned)N;
		}
	}
	for (int st = 0; st < 2; st++) {
		c->d_bif[st].ensure(elem_capacity * 4);
		HIP_TRY(hipMemsetAsync(c->d_bif[st].p, 0xFF, elem_capacity * 4, s));
	}
	c->bif_count = 0;
	{	// N = 2 x sum(max(0, len - k + 1)) (the candidate list of the active-set variant holds only part of the windows)
		unsigned long long N = 0;
		for (uint32_t ch = 0; ch < c->nchr; ch++) { const size_t len = c->sepidx[ch + 1] - c->sepidx[ch] - 1; if (len >= k) N += 2 * (len - k + 1); }
		c->stats.strand_kmers = N;
	}
	c->stats.kmer_table_ms = 0; c->stats.kmer_table_bytes = 0;
	if (nv) {
		k_lk_masks<<<nblocks(nv, 256), 256, 0, s>>>(L.skeys.as<unsigned long long>(), L.sidx.as<unsigned>(), nv, L.sym.as<unsigned>(), k,
		                                           L.mask.as<unsigned>(), L.flag.as<unsigned>());
		lk_inclusive_scan(c, L.flag.as<unsigned>(), L.scan.as<unsigned>(), nv);
		unsigned ngroups = 0;
		HIP_TRY(hipMemcpyAsync(&ngroups, L.scan.as<unsigned>() + (nv - 1), 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		L.gkeys.ensure((size_t)ngroups * 4 + 16); L.gmask.ensure((size_t)ngroups * 4 + 16); L.gcount.ensure(16);
		L.gbif.ensure((size_t)ngroups * 4 + 16); L.gid.ensure((size_t)ngroups * 4 + 16);
		{	// segmented OR of the masks: key = group number (the scan), one output per group, in order
			size_t tmp = 0;
			HIP_TRY(rocprim::reduce_by_key(nullptr, tmp, L.scan.as<unsigned>(), L.mask.as<unsigned>(), nv, L.gkeys.as<unsigned>(), L.gmask.as<unsigned>(),
			                               L.gcount.as<unsigned>(), BitOr(), rocprim::equal_to<unsigned>(), s));
			L.tmp.ensure(tmp);
			HIP_TRY(rocprim::reduce_by_key(L.tmp.p, tmp, L.scan.as<unsigned>(), L.mask.as<unsigned>(), nv, L.gkeys.as<unsigned>(), L.gmask.as<unsigned>(),
			                               L.gcount.as<unsigned>(), BitOr(), rocprim::equal_to<unsigned>(), s));
		}
		k_lk_group_bif<<<nblocks(ngroups, 256), 256, 0, s>>>(L.gmask.as<unsigned>(), ngroups, L.gbif.as<unsigned>());
		lk_exclusive_scan(c, L.gbif.as<unsigned>(), L.gid.as<unsigned>(), ngroups);
		unsigned last_id = 0, last_bif = 0;
		HIP_TRY(hipMemcpyAsync(&last_id, L.gid.as<unsigned>() + (ngroups - 1), 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(&last_bif, L.gbif.as<unsigned>() + (ngroups - 1), 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		c->bif_count = last_id + last_bif;
		k_lk_marks<<<nblocks(nv, 256), 256, 0, s>>>(L.sidx.as<unsigned>(), L.scan.as<unsigned>(), L.gbif.as<unsigned>(), L.gid.as<unsigned>(), nv,
		                                           c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>());
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	c->stats.bif_count = c->bif_count;
}

// =====================================================================================================================================
// Sharded rank doubling: the k > 32 enumeration of ONE job split over the GPUs attached to the context (BASELINE.json config 5: "k = 5000
// long-k path, 8 GPUs"; the function whose replacement is split is EnumerateBifurcationsSArrayInRAM, reference
// src/vertexenumeration.cpp:263-364 -- a single-threaded suffix array there).
//
// Two partitions of the same suffixes, one exchange between them per doubling round:
//   POSITION side  rank r owns the S positions [P_r, P_r+1), P_r = np * r / R: rank_h[] of its slice, the list of its positions that are
//                  still ACTIVE (share their h-prefix with somebody), and a halo of rank_h[P_r+1 .. P_r+1 + h) fetched from the ranks behind it;
//   SORTED side    rank q owns an interval [G_q, G_q+1) of the global sorted order.  Ranks are "index of the group's first member in the
//                  sorted order" (as in the single-GPU active-set variant above), so refining a group never moves another one: a suffix
//                  whose rank lies in [G_q, G_q+1) stays there for good, and owner(rank) is a binary search in G.  Round 1 (ranks = base-5
//                  value of the first 8 symbols) cuts the value range into R equal parts and fixes G.
// A round h -> 2h:  position side: key = (rank_h[i], rank_h[i + h]) for the active i, routed to owner(rank_h[i]) -- 12 B per active suffix
//                   over xGMI; sorted side: local radix sort (1 / R of the single-GPU sort), head flags, new ranks = old rank + offset of the
//                   subgroup inside its group, singletons flagged inactive; (position, new rank | active) routed back to the position
//                   owner -- 8 B per active suffix; position side: scatter, compact the active list.
// Final: candidate windows (active suffixes + the windows at chromosome ends, as above) are routed by rank to the sorted side, sorted
// by (rank_h[i], rank_h[i + k - h]); groups never straddle two ranks (same first rank => same owner), ids are local group numbers +
// the number of bifurcation groups on the ranks before (all-gather of one count); (element, id) marks are all-gathered and scattered
// into the dense arrays everywhere.  Bit-identical to the single-GPU result for any R (tests/test_gpu_shard.py: 2 / 3 / 5 virtual ranks).
//
// The layout arithmetic is exported device-free (include/sibelia_amd.h: sbl_longk_slices, sbl_longk_halo_plan, sbl_longk_owner,
// sbl_longk_value_bounds) and is what the pipeline itself calls; tests/test_shard_plan.py drives it with real gloo process groups.
// =====================================================================================================================================
#include "sbl_comm.h"
#include <chrono>

extern "C" sbl_status sbl_longk_slices(uint32_t nranks, uint64_t np, uint64_t *first /* nranks + 1 */)
{
	if (!nranks || !first) return SBL_ERR_BAD_ARG;
	for (uint32_t r = 0; r <= nranks; r++) first[r] = (uint64_t)((unsigned __int128)np * r / nranks);
	return SBL_OK;
}
// equal parts of the value range [0, maxvalue]: bounds[q] = ceil(q * (maxvalue + 1) / nranks)
extern "C" sbl_status sbl_longk_value_bounds(uint32_t nranks, uint64_t maxvalue, uint64_t *bounds /* nranks + 1 */)
{
	if (!nranks || !bounds) return SBL_ERR_BAD_ARG;
	for (uint32_t q = 0; q <= nranks; q++) bounds[q] = (uint64_t)(((unsigned __int128)q * (maxvalue + 1) + nranks - 1) / nranks);
	return SBL_OK;
}
// owner of x: the largest q < nranks with bounds[q] <= x (bounds ascending, bounds[0] = 0; empty intervals own nothing)
extern "C" sbl_status sbl_longk_owner(uint32_t nranks, const uint64_t *bounds, uint64_t x, uint32_t *owner)
{
	if (!nranks || !bounds || !owner) return SBL_ERR_BAD_ARG;
	uint32_t lo = 0, hi = nranks;
	while (hi - lo > 1) { uint32_t mid = (lo + hi) >> 1; if (bounds[mid] <= x) lo = mid; else hi = mid; }
	*owner = lo;
	return SBL_OK;
}
// halo of H positions behind every slice: rank r needs [P_r+1, min(np, P_r+1 + H)); sbytes / soff: what `rank` sends to peer p and where
// it starts in rank's own slice, rbytes / roff: what it receives from p and where that goes in its halo (bytes, 4-B records)
extern "C" sbl_status sbl_longk_halo_plan(uint32_t nranks, uint32_t rank, uint64_t np, uint64_t H, uint64_t *sbytes, uint64_t *soff, uint64_t *rbytes, uint64_t *roff)
{
	if (!nranks || rank >= nranks || !sbytes || !soff || !rbytes || !roff) return SBL_ERR_BAD_ARG;
	std::vector<uint64_t> P(nranks + 1);
	sbl_longk_slices(nranks, np, P.data());
	auto need = [&](uint32_t r, uint64_t &a, uint64_t &b) { a = P[r + 1]; b = std::min<uint64_t>(np, P[r + 1] + H); };
	for (uint32_t p = 0; p < nranks; p++) {
		uint64_t a, b;
		need(p, a, b);                                                        // what peer p needs: my part of it
		uint64_t lo = std::max(a, P[rank]), hi = std::min(b, P[rank + 1]);
		sbytes[p] = hi > lo ? (hi - lo) * 4 : 0; soff[p] = hi > lo ? (lo - P[rank]) * 4 : 0;
		need(rank, a, b);                                                     // what I need: peer p's part of it
		lo = std::max(a, P[p]); hi = std::min(b, P[p + 1]);
		rbytes[p] = hi > lo ? (hi - lo) * 4 : 0; roff[p] = hi > lo ? (lo - a) * 4 : 0;
	}
	return SBL_OK;
}

// symbol of the superGenome at position i, straight from the element array (what k_lk_super materialises)
__device__ __forceinline__ unsigned lk_sym_at(const uint8_t *__restrict__ ch, const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E, unsigned n, unsigned long long i)
{
	if (i >= n) return 0u;
	if (i < E) { const uint8_t c = ch[i]; return c == 'A' ? 1u : c == 'C' ? 2u : c == 'G' ? 3u : c == 'T' ? 4u : 0u; }
	const unsigned r = (unsigned)i - E, c = lk_chr_of(sepidx, nchr, r + 1), j = r - sepidx[c], len = sepidx[c + 1] - sepidx[c] - 1;
	if (j >= len) return 0u;
	const uint8_t x = ch[sepidx[c + 1] - 1 - j];
	return x == 'A' ? 4u : x == 'C' ? 3u : x == 'G' ? 2u : x == 'T' ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_lks_rank8(const uint8_t *__restrict__ ch, const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E, unsigned n,
                                                   unsigned lo, unsigned cnt, unsigned *__restrict__ rk, unsigned *__restrict__ act)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= cnt) return;
	const unsigned long long i = (unsigned long long)lo + j;
	unsigned v = 0;
#pragma unroll
	for (unsigned t = 0; t < 8; t++) v = v * 5u + lk_sym_at(ch, sepidx, nchr, E, n, i + t);
	rk[j] = v;
	act[j] = lo + j;
}
// key of an active position: (rank_h[i], rank_h[i + h]); the second half comes from the slice or from the halo behind it
__global__ void __launch_bounds__(256) k_lks_keys(const unsigned *__restrict__ act, unsigned na, const unsigned *__restrict__ rk, const unsigned *__restrict__ halo,
                                                  unsigned lo, unsigned hi_, unsigned np, unsigned h, unsigned rb, unsigned long long *__restrict__ keys, unsigned *__restrict__ idx)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= na) return;
	const unsigned i = act[j];
	const unsigned long long t = (unsigned long long)i + h;
	const unsigned second = t >= np ? 0u : t < hi_ ? rk[t - lo] : halo[t - hi_];
	keys[j] = ((unsigned long long)rk[i - lo] << rb) | second;
	idx[j] = i;
}

// ---- routing: records (a[i], b[i]) go to owner(bounds, route(a[i])); a block works on a fixed chunk, so the counting pass and the
// scattering pass see the same records
#define LKS_CHUNK 4096u
struct RouteHi { unsigned rb; __device__ unsigned long long operator()(unsigned long long a) const { return a >> rb; } };
struct RouteSelf { __device__ unsigned long long operator()(unsigned a) const { return a; } };
__device__ __forceinline__ unsigned lks_owner(const unsigned long long *sb, unsigned R, unsigned long long x)
{
	unsigned lo = 0, hi = R;
	while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (sb[mid] <= x) lo = mid; else hi = mid; }
	return lo;
}
template <class TA, class Route>
__global__ void __launch_bounds__(256) k_lks_route_count(const TA *__restrict__ a, unsigned n, Route route, const unsigned long long *__restrict__ bounds, unsigned R, unsigned *__restrict__ counts)
{
	__shared__ unsigned long long sb[65];
	__shared__ unsigned sc[64];
	if (threadIdx.x <= R) sb[threadIdx.x] = bounds[threadIdx.x];
	if (threadIdx.x < 64) sc[threadIdx.x] = 0;
	__syncthreads();
	const unsigned from = blockIdx.x * LKS_CHUNK, to = from + LKS_CHUNK < n ? from + LKS_CHUNK : n;
	for (unsigned i = from + threadIdx.x; i < to; i += 256) atomicAdd(&sc[lks_owner(sb, R, route(a[i]))], 1u);
	__syncthreads();
	if (threadIdx.x < R && sc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sc[threadIdx.x]);
}
template <class TA, class Route>
__global__ void __launch_bounds__(256) k_lks_route_scatter(const TA *__restrict__ a, const unsigned *__restrict__ b, unsigned n, Route route, const unsigned long long *__restrict__ bounds, unsigned R,
                                                           unsigned *__restrict__ cursors, TA *__restrict__ ao, unsigned *__restrict__ bo)
{
	__shared__ unsigned long long sb[65];
	__shared__ unsigned sc[64], base[64];
	if (threadIdx.x <= R) sb[threadIdx.x] = bounds[threadIdx.x];
	if (threadIdx.x < 64) sc[threadIdx.x] = 0;
	__syncthreads();
	const unsigned from = blockIdx.x * LKS_CHUNK, to = from + LKS_CHUNK < n ? from + LKS_CHUNK : n;
	for (unsigned i = from + threadIdx.x; i < to; i += 256) atomicAdd(&sc[lks_owner(sb, R, route(a[i]))], 1u);
	__syncthreads();
	if (threadIdx.x < R) { base[threadIdx.x] = sc[threadIdx.x] ? atomicAdd(&cursors[threadIdx.x], sc[threadIdx.x]) : 0u; sc[threadIdx.x] = 0; }
	__syncthreads();
	for (unsigned i = from + threadIdx.x; i < to; i += 256) {
		const TA av = a[i];
		const unsigned d = lks_owner(sb, R, route(av));
		const unsigned at = base[d] + atomicAdd(&sc[d], 1u);
		ao[at] = av; bo[at] = b[i];
	}
}
// new ranks of a sorted run + "still active" in bit 31 (ranks are < 2^31: np < 0x7FFFFFF0)
__global__ void __launch_bounds__(256) k_lks_newrank(const unsigned long long *__restrict__ skeys, const unsigned *__restrict__ gstart, const unsigned *__restrict__ sstart,
                                                     unsigned m, unsigned rb, int first, unsigned goff, unsigned *__restrict__ out)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const unsigned ss = sstart[j];
	const unsigned nr = first ? goff + ss : (unsigned)(skeys[j] >> rb) + (ss - gstart[j]);
	const bool single = ss == j && (j + 1 == m || sstart[j + 1] == j + 1);
	out[j] = nr | (single ? 0u : 0x80000000u);
}
__global__ void __launch_bounds__(256) k_lks_apply(const unsigned *__restrict__ bi, const unsigned *__restrict__ bv, unsigned m, unsigned lo, unsigned *__restrict__ rk, unsigned *__restrict__ aflag)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= m) return;
	const unsigned i = bi[j] - lo, v = bv[j];
	rk[i] = v & 0x7FFFFFFFu;
	aflag[i] = v >> 31;
}
// candidate windows of a slice: its active positions + the first / last window of either strand of every chromosome that START in the
// slice and are not active (an active one is in the list already; a chromosome of exactly k characters has ONE window per strand)
__global__ void __launch_bounds__(256) k_lks_cand(const unsigned *__restrict__ act, unsigned na, const unsigned *__restrict__ rk, const unsigned *__restrict__ halo, const unsigned *__restrict__ aflag,
                                                  unsigned lo, unsigned hi_, unsigned np, const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E, unsigned k, unsigned h, unsigned rb,
                                                  unsigned long long *__restrict__ keys, unsigned *__restrict__ idx, unsigned *__restrict__ valid)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned total = na + 4u * nchr;
	if (j >= total) return;
	unsigned i = 0; bool ok = false;
	if (j < na) {
		i = act[j];
		if (i < E) {
			const unsigned c = lk_chr_of(sepidx, nchr, i);
			ok = i < 2u * E - 1u && i > sepidx[c] && (unsigned long long)i + k <= sepidx[c + 1];
		} else if (i < 2u * E - 1u) {
			const unsigned r = i - E, c = lk_chr_of(sepidx, nchr, r + 1), jj = r - sepidx[c], len = sepidx[c + 1] - sepidx[c] - 1;
			ok = (unsigned long long)jj + k <= len;
		}
	} else {
		const unsigned t = j - na, c = t >> 2, which = t & 3u, len = sepidx[c + 1] - sepidx[c] - 1;
		ok = len >= k && !((which & 1u) && len == k);
		const unsigned off = (which & 1u) ? len - k : 0u;
		i = (which & 2u) ? E + sepidx[c] + off : sepidx[c] + 1u + off;
		ok = ok && i >= lo && i < hi_ && !aflag[i - lo];
		if (!ok) i = lo;
	}
	unsigned long long key = ~0ull;
	if (ok) {
		const unsigned long long t = (unsigned long long)i + (k - h);
		const unsigned second = t >= np ? 0u : t < hi_ ? rk[t - lo] : halo[t - hi_];
		key = ((unsigned long long)rk[i - lo] << rb) | second;
	}
	keys[j] = key; idx[j] = i; valid[j] = ok ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_lks_masks(const unsigned long long *__restrict__ skeys, const unsigned *__restrict__ sidx, unsigned nv, const uint8_t *__restrict__ ch,
                                                   const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E, unsigned n, unsigned k, unsigned *__restrict__ mask, unsigned *__restrict__ flag)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nv) return;
	const unsigned i = sidx[j];
	const unsigned p = lk_sym_at(ch, sepidx, nchr, E, n, (unsigned long long)i - 1), q = lk_sym_at(ch, sepidx, nchr, E, n, (unsigned long long)i + k);
	mask[j] = (1u << (p ? p - 1 : 4)) | (1u << (8 + (q ? q - 1 : 4)));
	flag[j] = (j == 0 || skeys[j] != skeys[j - 1]) ? 1u : 0u;
}
// (element | strand << 31, id) of every member of a bifurcation group, compacted with one atomic per wave
__global__ void __launch_bounds__(256) k_lks_marks(const unsigned *__restrict__ sidx, const unsigned *__restrict__ gscan, const unsigned *__restrict__ gbif, const unsigned *__restrict__ gid,
                                                   unsigned nv, unsigned idoff, const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E,
                                                   unsigned *__restrict__ cursor, unsigned *__restrict__ mcode, unsigned *__restrict__ mid)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	bool has = false;
	unsigned code = 0, id = 0;
	if (j < nv) {
		const unsigned g = gscan[j] - 1;
		if (gbif[g]) {
			has = true; id = idoff + gid[g];
			const unsigned i = sidx[j];
			if (i < E) code = i;
			else { const unsigned r = i - E, c = lk_chr_of(sepidx, nchr, r + 1), jj = r - sepidx[c]; code = (sepidx[c + 1] - 1 - jj) | 0x80000000u; }
		}
	}
	const unsigned long long m = __ballot(has);
	if (!m) return;
	const unsigned lane = threadIdx.x & 63u, leader = (unsigned)__builtin_ctzll(m);
	unsigned base = 0;
	if (lane == leader) base = atomicAdd(cursor, (unsigned)__popcll(m));
	base = __shfl(base, leader);
	if (has) { const unsigned at = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull)); mcode[at] = code; mid[at] = id; }
}
__global__ void __launch_bounds__(256) k_lks_scatter_marks(const unsigned *__restrict__ mcode, const unsigned *__restrict__ mid, unsigned n, unsigned *__restrict__ bif0, unsigned *__restrict__ bif1)
{
	unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= n) return;
	const unsigned c = mcode[j];
	if (c >> 31) bif1[c & 0x7FFFFFFFu] = mid[j]; else bif0[c] = mid[j];
}

struct LkShardScratch {
	DevBuf rk, halo, aflag, act, k0, i0, k1, i1, rkeys, ridx, skeys, sidx, f0, f1, s0, s1, ov, oi1, ov1, bi, bv, cnt, bounds, mcode, mid, gcode, gid2;
};
struct LongKShardHolder { LkShardScratch s; };

namespace {
struct LksClock {
	double ms = 0;
	template <class F> void time(F f) { auto t0 = std::chrono::steady_clock::now(); f(); ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
// routes n records (a, b) by route(a) against `bounds` and ships them: returns the number of records received (recv_a / recv_b filled)
template <class TA, class Route>
size_t lks_route_exchange(sbl_ctx *c, LkShardScratch &S, LksClock &clk, const TA *a, const unsigned *b, size_t n, Route route, const std::vector<uint64_t> &bounds,
                          DevBuf &tmp_a, DevBuf &tmp_b, DevBuf &recv_a, DevBuf &recv_b)
{
	SblComm *cm = c->comm;
	const uint32_t R = cm->n, r = cm->rank;
	hipStream_t s = c->stream;
	S.bounds.ensure((R + 1) * 8); S.cnt.ensure(2 * 64 * 4);
	HIP_TRY(hipMemcpyAsync(S.bounds.p, bounds.data(), (R + 1) * 8, hipMemcpyHostToDevice, s));
	HIP_TRY(hipMemsetAsync(S.cnt.p, 0, 2 * 64 * 4, s));
	unsigned *counts = S.cnt.as<unsigned>(), *cursors = counts + 64;
	const unsigned nb = (unsigned)((n + LKS_CHUNK - 1) / LKS_CHUNK);
	std::vector<unsigned> hc(R, 0);
	if (n) {
		k_lks_route_count<TA, Route><<<nb, 256, 0, s>>>(a, (unsigned)n, route, S.bounds.as<unsigned long long>(), R, counts);
		HIP_TRY(hipMemcpyAsync(hc.data(), counts, R * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
	} else HIP_TRY(hipStreamSynchronize(s));
	std::vector<unsigned> send_at(R + 1, 0);
	for (uint32_t p = 0; p < R; p++) send_at[p + 1] = send_at[p] + hc[p];
	tmp_a.ensure(n * sizeof(TA) + 16); tmp_b.ensure(n * 4 + 16);
	if (n) {
		HIP_TRY(hipMemcpyAsync(cursors, send_at.data(), R * 4, hipMemcpyHostToDevice, s));
		k_lks_route_scatter<TA, Route><<<nb, 256, 0, s>>>(a, b, (unsigned)n, route, S.bounds.as<unsigned long long>(), R, cursors, tmp_a.as<TA>(), tmp_b.as<unsigned>());
		HIP_TRY(hipGetLastError());
	}
	std::vector<unsigned long long> scount(R), allcount((size_t)R * R);
	for (uint32_t p = 0; p < R; p++) scount[p] = hc[p];
	clk.time([&] { cm->allgather_host(c, scount.data(), R * 8, allcount.data()); });
	std::vector<size_t> sb(R), so(R), rb(R), ro(R);
	uint64_t got = 0;
	SBL_CHECK(sbl_shard_exchange_plan(R, r, (const uint64_t *)allcount.data(), send_at.data(), sizeof(TA), (uint64_t *)sb.data(), (uint64_t *)so.data(),
	                                  (uint64_t *)rb.data(), (uint64_t *)ro.data(), &got) == SBL_OK, SBL_ERR_INTERNAL, "long-k exchange plan: the gathered counts contradict my own");
	SBL_CHECK(got < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many records for one rank");
	recv_a.ensure(got * sizeof(TA) + 16); recv_b.ensure(got * 4 + 16);
	clk.time([&] { cm->alltoallv(c, tmp_a.as<char>(), sb.data(), so.data(), recv_a.as<char>(), rb.data(), ro.data()); });
	for (uint32_t p = 0; p < R; p++) { if (p != r) c->stats.exchange_bytes += sb[p] + sb[p] / sizeof(TA) * 4; sb[p] = sb[p] / sizeof(TA) * 4; so[p] = so[p] / sizeof(TA) * 4; rb[p] = rb[p] / sizeof(TA) * 4; ro[p] = ro[p] / sizeof(TA) * 4; }
	clk.time([&] { cm->alltoallv(c, tmp_b.as<char>(), sb.data(), so.data(), recv_b.as<char>(), rb.data(), ro.data()); });
	return (size_t)got;
}
// halo[0 .. H) = rank[P_r+1 .. P_r+1 + H) from the ranks behind (zeros past the end of S)
void lks_fetch_halo(sbl_ctx *c, LkShardScratch &S, LksClock &clk, size_t np, size_t H)
{
	SblComm *cm = c->comm;
	const uint32_t R = cm->n, r = cm->rank;
	S.halo.ensure(H * 4 + 16);
	HIP_TRY(hipMemsetAsync(S.halo.p, 0, H * 4 + 16, c->stream));
	std::vector<size_t> sb(R), so(R), rb(R), ro(R);
	SBL_CHECK(sbl_longk_halo_plan(R, r, np, H, (uint64_t *)sb.data(), (uint64_t *)so.data(), (uint64_t *)rb.data(), (uint64_t *)ro.data()) == SBL_OK, SBL_ERR_INTERNAL, "halo plan");
	for (uint32_t p = 0; p < R; p++) if (p != r) c->stats.exchange_bytes += sb[p];
	clk.time([&] { cm->alltoallv(c, S.rk.as<char>(), sb.data(), so.data(), S.halo.as<char>(), rb.data(), ro.data()); });
}
unsigned long long lks_sum(sbl_ctx *c, LksClock &clk, unsigned long long mine, std::vector<unsigned long long> *all_out = nullptr)
{
	SblComm *cm = c->comm;
	std::vector<unsigned long long> in(1, mine), all(cm->n);
	clk.time([&] { cm->allgather_host(c, in.data(), 8, all.data()); });
	unsigned long long t = 0;
	for (auto v : all) t += v;
	if (all_out) *all_out = all;
	return t;
}
}

static void run_enumeration_longk_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity);
void sbl_run_enumeration_longk_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	try { run_enumeration_longk_sharded(c, k, elem_capacity); }
	catch (...) { c->comm->abort_peers(); throw; }
}
static void run_enumeration_longk_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	hipStream_t s = c->stream;
	SblComm *cm = c->comm;
	const uint32_t R = cm->n, r = cm->rank;
	const size_t E = c->nelem, n = 2 * E - 1, np = n + k;
	SBL_CHECK(np < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "input too large for 32-bit suffix ranks");
	const bool trace = getenv("SBL_TRACE") != nullptr;
	c->cur_k = k;
	c->stats.exchange_bytes = 0;
	LksClock clk;
	LongKScratch &L = lk_of(c);
	if (!L.shard) L.shard = new LongKShardHolder;
	LkShardScratch &S = L.shard->s;
	std::vector<uint64_t> P(R + 1), G(R + 1), V(R + 1);
	SBL_CHECK(sbl_longk_slices(R, np, P.data()) == SBL_OK, SBL_ERR_INTERNAL, "slices");
	const unsigned lo = (unsigned)P[r], hi = (unsigned)P[r + 1], len = hi - lo;
	c->d_counters.ensure(64 * 4);
	HIP_TRY(hipMemsetAsync(c->d_counters.p, 0, 64 * 4, s));
	S.rk.ensure((size_t)len * 4 + 16); S.aflag.ensure((size_t)len * 4 + 16); S.act.ensure((size_t)len * 4 + 64);
	HIP_TRY(hipMemsetAsync(S.aflag.p, 0, (size_t)len * 4 + 16, s));
	// ---- the first three doubling rounds are arithmetic (base-5 number of 8 symbols), on the slice
	if (len) k_lks_rank8<<<nblocks(len, 256), 256, 0, s>>>(c->d_ch.as<uint8_t>(), c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, (unsigned)n, lo, len, S.rk.as<unsigned>(), S.act.as<unsigned>());
	size_t h = 8;
	const unsigned maxrank0 = 390624, rbp = lk_bits(np);
	SBL_CHECK(sbl_longk_value_bounds(R, maxrank0, V.data()) == SBL_OK, SBL_ERR_INTERNAL, "value bounds");
	unsigned na = len;                                                    // my active positions
	unsigned long long na_all = np;
	bool first = true;
	unsigned round = 0;
	while (2 * h <= k && na_all) {
		const unsigned rb = first ? lk_bits(maxrank0) : rbp;
		const unsigned long long bytes0 = c->stats.exchange_bytes;
		lks_fetch_halo(c, S, clk, np, h);
		S.k0.ensure((size_t)na * 8 + 16); S.i0.ensure((size_t)na * 4 + 16);
		if (na) k_lks_keys<<<nblocks(na, 256), 256, 0, s>>>(S.act.as<unsigned>(), na, S.rk.as<unsigned>(), S.halo.as<unsigned>(), lo, hi, (unsigned)np, (unsigned)h, rb,
		                                                   S.k0.as<unsigned long long>(), S.i0.as<unsigned>());
		// to the sorted side: by value range in the first round (which fixes G), by G afterwards
		const size_t m = lks_route_exchange<unsigned long long, RouteHi>(c, S, clk, S.k0.as<unsigned long long>(), S.i0.as<unsigned>(), na, RouteHi{rb}, first ? V : G,
		                                                                S.k1, S.i1, S.rkeys, S.ridx);
		if (first) {
			std::vector<unsigned long long> all;
			lks_sum(c, clk, m, &all);
			G[0] = 0;
			for (uint32_t q = 0; q < R; q++) G[q + 1] = G[q] + all[q];
		}
		S.skeys.ensure(m * 8 + 16); S.sidx.ensure(m * 4 + 16); S.f0.ensure(m * 4 + 16); S.f1.ensure(m * 4 + 16); S.s0.ensure(m * 4 + 16); S.s1.ensure(m * 4 + 16); S.ov.ensure(m * 4 + 16);
		if (m) {
			lk_sort(c, S.rkeys.as<unsigned long long>(), S.skeys.as<unsigned long long>(), S.ridx.as<unsigned>(), S.sidx.as<unsigned>(), m, std::min(64u, 2 * rb));
			k_lk_heads2<<<nblocks(m, 256), 256, 0, s>>>(S.skeys.as<unsigned long long>(), (unsigned)m, rb, S.f0.as<unsigned>(), S.f1.as<unsigned>());
			lk_max_scan(c, S.f0.as<unsigned>(), S.s0.as<unsigned>(), m);      // gstart
			lk_max_scan(c, S.f1.as<unsigned>(), S.s1.as<unsigned>(), m);      // sstart
			k_lks_newrank<<<nblocks(m, 256), 256, 0, s>>>(S.skeys.as<unsigned long long>(), S.s0.as<unsigned>(), S.s1.as<unsigned>(), (unsigned)m, rb, first ? 1 : 0, (unsigned)G[r], S.ov.as<unsigned>());
		}
		// back to the position side
		const size_t mb = lks_route_exchange<unsigned, RouteSelf>(c, S, clk, S.sidx.as<unsigned>(), S.ov.as<unsigned>(), m, RouteSelf{}, P, S.oi1, S.ov1, S.bi, S.bv);
		SBL_CHECK(mb == na, SBL_ERR_INTERNAL, "sharded rank doubling: a position did not get its rank back");
		if (mb) k_lks_apply<<<nblocks(mb, 256), 256, 0, s>>>(S.bi.as<unsigned>(), S.bv.as<unsigned>(), (unsigned)mb, lo, S.rk.as<unsigned>(), S.aflag.as<unsigned>());
		if (len) {
			size_t tmp = 0;
			rocprim::counting_iterator<unsigned> it(lo);
			HIP_TRY(rocprim::select(nullptr, tmp, it, S.aflag.as<unsigned>(), S.act.as<unsigned>(), c->d_counters.as<unsigned>() + 8, (size_t)len, s));
			L.tmp.ensure(tmp);
			HIP_TRY(rocprim::select(L.tmp.p, tmp, it, S.aflag.as<unsigned>(), S.act.as<unsigned>(), c->d_counters.as<unsigned>() + 8, (size_t)len, s));
			HIP_TRY(hipMemcpyAsync(&na, c->d_counters.as<unsigned>() + 8, 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
		} else na = 0;
		na_all = lks_sum(c, clk, na);
		first = false;
		h *= 2;
		round++;
		if (trace) fprintf(stderr, "[sbl] long-k rank %u round %u (h = %zu): sorted %zu suffixes here, %llu still active in all, %llu bytes sent this round\n", r, round, h, m, na_all,
		                   (unsigned long long)(c->stats.exchange_bytes - bytes0));
	}
	while (2 * h <= k) h *= 2;                                            // (everything unique before the last level: the offset of the second half only has to be valid)
	SBL_CHECK(!first, SBL_ERR_INTERNAL, "sharded rank doubling needs k >= 16");
	// ---- candidate windows: active suffixes + chromosome ends, keyed by (rank_h[i], rank_h[i + k - h]), to the sorted side
	lks_fetch_halo(c, S, clk, np, k - h);
	const unsigned ncand = na + 4u * c->nchr;
	S.k0.ensure((size_t)ncand * 8 + 16); S.i0.ensure((size_t)ncand * 4 + 16); S.f0.ensure((size_t)ncand * 4 + 16);
	S.k1.ensure((size_t)ncand * 8 + 16); S.i1.ensure((size_t)ncand * 4 + 16);
	k_lks_cand<<<nblocks(ncand, 256), 256, 0, s>>>(S.act.as<unsigned>(), na, S.rk.as<unsigned>(), S.halo.as<unsigned>(), S.aflag.as<unsigned>(), lo, hi, (unsigned)np,
	                                              c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, k, (unsigned)h, rbp, S.k0.as<unsigned long long>(), S.i0.as<unsigned>(), S.f0.as<unsigned>());
	unsigned nvl = 0;
	{
		size_t tmp = 0;
		HIP_TRY(rocprim::select(nullptr, tmp, S.k0.as<unsigned long long>(), S.f0.as<unsigned>(), S.k1.as<unsigned long long>(), c->d_counters.as<unsigned>() + 8, (size_t)ncand, s));
		L.tmp.ensure(tmp);
		HIP_TRY(rocprim::select(L.tmp.p, tmp, S.k0.as<unsigned long long>(), S.f0.as<unsigned>(), S.k1.as<unsigned long long>(), c->d_counters.as<unsigned>() + 8, (size_t)ncand, s));
		HIP_TRY(rocprim::select(nullptr, tmp, S.i0.as<unsigned>(), S.f0.as<unsigned>(), S.i1.as<unsigned>(), c->d_counters.as<unsigned>() + 8, (size_t)ncand, s));
		L.tmp.ensure(tmp);
		HIP_TRY(rocprim::select(L.tmp.p, tmp, S.i0.as<unsigned>(), S.f0.as<unsigned>(), S.i1.as<unsigned>(), c->d_counters.as<unsigned>() + 8, (size_t)ncand, s));
		HIP_TRY(hipMemcpyAsync(&nvl, c->d_counters.as<unsigned>() + 8, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
	}
	const size_t nv = lks_route_exchange<unsigned long long, RouteHi>(c, S, clk, S.k1.as<unsigned long long>(), S.i1.as<unsigned>(), nvl, RouteHi{rbp}, G, S.k0, S.i0, S.rkeys, S.ridx);
	for (int st = 0; st < 2; st++) {
		c->d_bif[st].ensure(elem_capacity * 4);
		HIP_TRY(hipMemsetAsync(c->d_bif[st].p, 0xFF, elem_capacity * 4, s));
	}
	{
		unsigned long long N = 0;
		for (uint32_t ch = 0; ch < c->nchr; ch++) { const size_t l = c->sepidx[ch + 1] - c->sepidx[ch] - 1; if (l >= k) N += 2 * (l - k + 1); }
		c->stats.strand_kmers = N;
	}
	c->stats.kmer_table_ms = 0; c->stats.kmer_table_bytes = 0;
	unsigned nbif_local = 0, nmarks = 0;
	S.skeys.ensure(nv * 8 + 16); S.sidx.ensure(nv * 4 + 16); S.f0.ensure(nv * 4 + 16); S.f1.ensure(nv * 4 + 16); S.s0.ensure(nv * 4 + 16);
	S.mcode.ensure(nv * 4 + 16); S.mid.ensure(nv * 4 + 16);
	unsigned ngroups = 0;
	if (nv) {
		lk_sort(c, S.rkeys.as<unsigned long long>(), S.skeys.as<unsigned long long>(), S.ridx.as<unsigned>(), S.sidx.as<unsigned>(), nv, std::min(64u, 2 * rbp));
		k_lks_masks<<<nblocks(nv, 256), 256, 0, s>>>(S.skeys.as<unsigned long long>(), S.sidx.as<unsigned>(), (unsigned)nv, c->d_ch.as<uint8_t>(), c->d_sepidx.as<unsigned>(), c->nchr,
		                                            (unsigned)E, (unsigned)n, k, S.f1.as<unsigned>(), S.f0.as<unsigned>());
		lk_inclusive_scan(c, S.f0.as<unsigned>(), S.s0.as<unsigned>(), nv);
		HIP_TRY(hipMemcpyAsync(&ngroups, S.s0.as<unsigned>() + (nv - 1), 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		L.gkeys.ensure((size_t)ngroups * 4 + 16); L.gmask.ensure((size_t)ngroups * 4 + 16); L.gcount.ensure(16);
		L.gbif.ensure((size_t)ngroups * 4 + 16); L.gid.ensure((size_t)ngroups * 4 + 16);
		{
			size_t tmp = 0;
			HIP_TRY(rocprim::reduce_by_key(nullptr, tmp, S.s0.as<unsigned>(), S.f1.as<unsigned>(), nv, L.gkeys.as<unsigned>(), L.gmask.as<unsigned>(),
			                               L.gcount.as<unsigned>(), BitOr(), rocprim::equal_to<unsigned>(), s));
			L.tmp.ensure(tmp);
			HIP_TRY(rocprim::reduce_by_key(L.tmp.p, tmp, S.s0.as<unsigned>(), S.f1.as<unsigned>(), nv, L.gkeys.as<unsigned>(), L.gmask.as<unsigned>(),
			                               L.gcount.as<unsigned>(), BitOr(), rocprim::equal_to<unsigned>(), s));
		}
		k_lk_group_bif<<<nblocks(ngroups, 256), 256, 0, s>>>(L.gmask.as<unsigned>(), ngroups, L.gbif.as<unsigned>());
		lk_exclusive_scan(c, L.gbif.as<unsigned>(), L.gid.as<unsigned>(), ngroups);
		unsigned last_id = 0, last_bif = 0;
		HIP_TRY(hipMemcpyAsync(&last_id, L.gid.as<unsigned>() + (ngroups - 1), 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(&last_bif, L.gbif.as<unsigned>() + (ngroups - 1), 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		nbif_local = last_id + last_bif;
	}
	// ids: my groups come after those of the ranks before me in the sorted order
	std::vector<unsigned long long> allb;
	const unsigned long long nbif = lks_sum(c, clk, nbif_local, &allb);
	unsigned idoff = 0;
	for (uint32_t q = 0; q < r; q++) idoff += (unsigned)allb[q];
	c->bif_count = (uint32_t)nbif;
	if (nv) {
		HIP_TRY(hipMemsetAsync(c->d_counters.as<unsigned>() + 9, 0, 4, s));
		k_lks_marks<<<nblocks(nv, 256), 256, 0, s>>>(S.sidx.as<unsigned>(), S.s0.as<unsigned>(), L.gbif.as<unsigned>(), L.gid.as<unsigned>(), (unsigned)nv, idoff,
		                                            c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, c->d_counters.as<unsigned>() + 9, S.mcode.as<unsigned>(), S.mid.as<unsigned>());
		HIP_TRY(hipMemcpyAsync(&nmarks, c->d_counters.as<unsigned>() + 9, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
	}
	// marks of everybody's groups into the dense arrays everywhere (8 B per instance)
	{
		std::vector<unsigned long long> allm;
		const unsigned long long tot = lks_sum(c, clk, nmarks, &allm);
		std::vector<size_t> sb(R, (size_t)nmarks * 4), so(R, 0), rb(R), ro(R);
		size_t off = 0;
		for (uint32_t p = 0; p < R; p++) { rb[p] = (size_t)allm[p] * 4; ro[p] = off; off += rb[p]; }
		S.gcode.ensure(tot * 4 + 16); S.gid2.ensure(tot * 4 + 16);
		clk.time([&] { cm->alltoallv(c, S.mcode.as<char>(), sb.data(), so.data(), S.gcode.as<char>(), rb.data(), ro.data()); });
		clk.time([&] { cm->alltoallv(c, S.mid.as<char>(), sb.data(), so.data(), S.gid2.as<char>(), rb.data(), ro.data()); });
		c->stats.exchange_bytes += (unsigned long long)nmarks * 8 * (R - 1);
		if (tot) k_lks_scatter_marks<<<nblocks(tot, 256), 256, 0, s>>>(S.gcode.as<unsigned>(), S.gid2.as<unsigned>(), (unsigned)tot, c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>());
	}
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	c->stats.bif_count = c->bif_count;
	c->stats.exchange_ms = clk.ms;
	if (trace) fprintf(stderr, "[sbl] long-k rank %u: %u rounds, %zu candidate windows sorted here, %u bifurcation groups here of %llu, %llu bytes sent in all\n", r, round, nv, nbif_local, nbif,
	                   (unsigned long long)c->stats.exchange_bytes);
}

static void lk_shard_free(LongKShardHolder *h)
{
	if (!h) return;
	LkShardScratch &S = h->s;
	for (DevBuf *b : { &S.rk, &S.halo, &S.aflag, &S.act, &S.k0, &S.i0, &S.k1, &S.i1, &S.rkeys, &S.ridx, &S.skeys, &S.sidx, &S.f0, &S.f1, &S.s0, &S.s1, &S.ov, &S.oi1, &S.ov1, &S.bi, &S.bv,
	                   &S.cnt, &S.bounds, &S.mcode, &S.mid, &S.gcode, &S.gid2 })
		b->release();
	delete h;
}
