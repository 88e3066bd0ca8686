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

// final keys of the k-windows; windows that contain a separator get the all-ones key and sort to the end
__global__ void __launch_bounds__(256) k_lk_window_keys(const unsigned *__restrict__ rank, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                        unsigned E, unsigned n, unsigned k, unsigned h,
                                                        unsigned long long *__restrict__ keys, unsigned *__restrict__ idx, unsigned *__restrict__ nvalid)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;                                          // (lanes that leave here take no part in the ballot below)
	bool valid = false;
	if (i < E) {
		unsigned c = lk_chr_of(sepidx, nchr, i);
		valid = i > sepidx[c] && (unsigned long long)i + k <= sepidx[c + 1];
	} else {
		unsigned r = i - E, c = lk_chr_of(sepidx, nchr, r + 1);
		unsigned j = r - sepidx[c], len = sepidx[c + 1] - sepidx[c] - 1;
		valid = (unsigned long long)j + k <= len;
	}
	keys[i] = valid ? (((unsigned long long)rank[i] << 32) | rank[i + k - h]) : ~0ull;
	idx[i] = i;
	// one atomic per wave, not per window: 1.8 G atomics on one address were 319 ms of config 5's 2.87 s (rocprofv3, round 3)
	const unsigned long long m = __ballot(valid);
	if (m && (threadIdx.x & 63u) == (unsigned)__builtin_ctzll(m)) atomicAdd(nvalid, (unsigned)__popcll(m));
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

struct LongKScratch {
	DevBuf rank[2], keys, skeys, idx, sidx, flag, scan, mask, gkeys, gmask, gcount, gbif, gid, tmp, sym, act, aux;
};
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

void sbl_run_enumeration_longk(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	hipStream_t s = c->stream;
	const size_t E = c->nelem, n = 2 * E - 1, np = n + k;             // 2L + 2 nchr + 1 = 2E - 1; padded with k '#'
	SBL_CHECK(np < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "input too large for 32-bit suffix ranks");
	c->cur_k = k;
	LongKScratch &L = lk_of(c);
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
	if (k >= 16 && getenv("SBL_LONGK_FROM_1") == nullptr) {          // (k > 32 here: always; the switch is for A/B tests)
		k_lk_rank8<<<nblocks(np, 256), 256, 0, s>>>(L.sym.as<unsigned>(), (unsigned)np, L.rank[1].as<unsigned>());
		rank = L.rank[1].as<unsigned>();
		h = 8; maxrank = 390624;
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
			if (full) k_lk_pair_keys<<<nblocks(np, 256), 256, 0, s>>>(rank, (unsigned)np, (unsigned)h, rb, L.keys.as<unsigned long long>(), L.idx.as<unsigned>());
			else k_lk_active_keys<<<nblocks(na, 256), 256, 0, s>>>(rank, L.act.as<unsigned>(), na, (unsigned)np, (unsigned)h, rb, L.keys.as<unsigned long long>(), L.idx.as<unsigned>());
			lk_sort(c, L.keys.as<unsigned long long>(), L.skeys.as<unsigned long long>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), m, std::min(64u, 2 * rb));
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
			h *= 2;
			// decided once, after the first round: an input whose suffixes mostly still share their 16-prefix with somebody is a set of related
			// genomes -- it stays that way, and the plain doubling below (dense ranks: fewer key bits, no max-scans, no compaction) is 10 - 20 %
			// faster on it (8 x 4.6 Mbp, k = 100 / 500: 56 ms against 62 - 67 ms); the ranks so far are valid ranks for it, just not dense
			if (was_first && na > np / 2 && getenv("SBL_LONGK_FORCE_ACTIVE") == nullptr) { plain = true; maxrank = (unsigned)(np - 1); break; }
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
		k_lk_window_keys<<<nblocks(n, 256), 256, 0, s>>>(rank, c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, (unsigned)n, k, (unsigned)h,
		                                                L.keys.as<unsigned long long>(), L.idx.as<unsigned>(), c->d_counters.as<unsigned>());
		lk_sort(c, L.keys.as<unsigned long long>(), L.skeys.as<unsigned long long>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), n);
		HIP_TRY(hipMemcpyAsync(&nv, c->d_counters.p, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
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
