// kmer_kernels.h -- gfx950 kernels for bifurcation enumeration at vertex size k <= 32.
//
// Replaces the suffix-array + LCP group scan of the reference
// (IndexedSequence::EnumerateBifurcationsSArrayInRAM, reference src/vertexenumeration.cpp:263-364)
// with the order-free formulation of SURVEY.md §0.3 / §8a-E1:
//   * a strand-specific k-mer is a bifurcation iff, over all its occurrences on both strands,
//     the set of preceding characters or the set of following characters has more than one
//     element or contains the chromosome boundary '#'  (vertexenumeration.cpp:67-70,:330);
//   * its id is its rank among all bifurcation k-mers in lexicographic order A<C<G<T (:348-355);
//   * a k-mer and its reverse complement are bifurcations together (prev set of w = complement of
//     the next set of rc(w)), so ONE table entry per canonical k-mer carries both.
//
// Data layout in HBM ("element array", identical for every stage):
//   ch[E]   1 B/element: '$' c0 '$' c1 '$' ...  (E = L + nchr + 1, like DNASequence, dnasequence.cpp:75-103)
//   pk[E/32] u64: 32 bases per word, 2 bit each, first base in the top bits (A=0 C=1 G=2 T=3)
//   sp[E/32] u32: separator bit per element, first element in the top bit
// The k-mer table itself is radix-bucketed (kmer_bucket_kernels.h); this file holds what every path shares: packing, the
// LDS-staged tile walker, mask / reverse-complement helpers, ranking and mark compaction kernels.
//
// Integer / indexing work only: no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SBL_NONE 0xFFFFFFFFu
__device__ __host__ __forceinline__ unsigned long long kmer_hash(unsigned long long x)
{
	x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
	x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
	x ^= x >> 33;
	return x;
}

// reverse complement of a k-mer code (k <= 32): complement = ~, then reverse the 2-bit groups
__device__ __host__ __forceinline__ unsigned long long rc_code(unsigned long long c, unsigned k)
{
	c = ~c;
	c = ((c >> 2) & 0x3333333333333333ull) | ((c & 0x3333333333333333ull) << 2);
	c = ((c >> 4) & 0x0F0F0F0F0F0F0F0Full) | ((c & 0x0F0F0F0F0F0F0F0Full) << 4);
	c = ((c >> 8) & 0x00FF00FF00FF00FFull) | ((c & 0x00FF00FF00FF00FFull) << 8);
	c = ((c >> 16) & 0x0000FFFF0000FFFFull) | ((c & 0x0000FFFF0000FFFFull) << 16);
	c = (c >> 32) | (c << 32);
	return c >> (64 - 2 * k);
}

__device__ __forceinline__ bool mask_is_bifurcation(unsigned m)
{
	unsigned p = m & 0x1F, n = (m >> 8) & 0x1F;
	return (p & 0x10) || (n & 0x10) || __popc(p & 0xF) > 1 || __popc(n & 0xF) > 1;
}

// ---------------------------------------------------------------------------------------------
// K1: ASCII -> 2-bit words + separator bits.  One thread per 32 elements (two 16-B loads).
// ch must be padded with '$' up to a multiple of 32 elements.
static __global__ void __launch_bounds__(256) k_pack2bit(const uint8_t *__restrict__ ch, unsigned long long *__restrict__ pk,
                                                  unsigned *__restrict__ sp, size_t nwords)
{
	size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= nwords) return;
	const uint4 *src = reinterpret_cast<const uint4 *>(ch + w * 32);
	uint4 q[2] = { src[0], src[1] };
	const unsigned *u = reinterpret_cast<const unsigned *>(q);
	unsigned long long code = 0;
	unsigned sep = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) {
		unsigned v = u[i];
#pragma unroll
		for (int b = 0; b < 4; b++) {
			unsigned c = (v >> (8 * b)) & 0xFF;
			unsigned x = (c >> 1) & 3;             // A:0 C:1 T:2 G:3
			x ^= x >> 1;                           // A:0 C:1 G:2 T:3
			bool s = (c == '$');
			code = (code << 2) | (s ? 0u : x);
			sep = (sep << 1) | (s ? 1u : 0u);
		}
	}
	pk[w] = code;
	sp[w] = sep;
}

// ---------------------------------------------------------------------------------------------
// Sliding-window walker shared by the table-build (K2) and resolve (K5) kernels.
// A workgroup stages TILE positions (+ one word of halo on each side) of the packed sequence
// through LDS with coalesced 8-B loads; every thread then slides over PER_THREAD consecutive
// positions updating the forward and reverse-complement codes incrementally.
#define KM_TILE_WORDS 128                     // 128 x 32 = 4096 positions per tile
#define KM_THREADS 256
#define KM_PER_THREAD (KM_TILE_WORDS * 32 / KM_THREADS)   // 16

struct KmerTile {
	unsigned long long w[KM_TILE_WORDS + 3];
	unsigned s[KM_TILE_WORDS + 3];
};

__device__ __forceinline__ void tile_load(KmerTile &t, const unsigned long long *__restrict__ pk,
                                          const unsigned *__restrict__ sp, size_t tile, size_t nwords)
{
	// LDS word j holds global word tile*KM_TILE_WORDS - 1 + j
	for (unsigned j = threadIdx.x; j < KM_TILE_WORDS + 3; j += KM_THREADS) {
		long long gw = (long long)(tile * KM_TILE_WORDS) - 1 + j;
		bool in = gw >= 0 && (size_t)gw < nwords;
		t.w[j] = in ? pk[gw] : 0ull;
		t.s[j] = in ? sp[gw] : 0xFFFFFFFFu;   // outside the array counts as separator
	}
}
// base (0..3) and separator flag of element `e` given relative to the tile start (e may be -1 .. TILE+32)
__device__ __forceinline__ unsigned tile_base(const KmerTile &t, int e)
{
	int j = (e + 32) >> 5, o = (e + 32) & 31;
	return (unsigned)(t.w[j] >> (62 - 2 * o)) & 3u;
}
__device__ __forceinline__ bool tile_sep(const KmerTile &t, int e)
{
	int j = (e + 32) >> 5, o = (e + 32) & 31;
	return (t.s[j] >> (31 - o)) & 1u;
}

// K4b: after the radix sort of the keys, rank = bifurcation id.  pairids[2p+o] = id.
static __global__ void __launch_bounds__(256) k_scatter_ids(const unsigned long long *__restrict__ skeys, const unsigned *__restrict__ spayload,
                                                     unsigned nkeys, unsigned k, unsigned *__restrict__ pairids)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nkeys) return;
	unsigned p = spayload[i];
	pairids[p] = i;
	if (!(p & 1) && rc_code(skeys[i], k) == skeys[i]) pairids[p + 1] = i;   // palindrome: one vertex for both orientations
}

// marks gathered from the owners of the sharded enumeration (shard.hip) into the dense arrays
static __global__ void __launch_bounds__(256) k_scatter_marks(const unsigned *__restrict__ elem, const unsigned *__restrict__ id, size_t n, unsigned *__restrict__ bif)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) bif[elem[i]] = id[i];
}

// ---------------------------------------------------------------------------------------------
// Ordered compaction of the marks of one strand: (element index, id) pairs in ascending element order.
// Pass 1 counts per 1024-element chunk, pass 2 (after an exclusive scan of the counts) writes.
static __global__ void __launch_bounds__(256) k_count_marks(const unsigned *__restrict__ bif, size_t nelem, unsigned *__restrict__ chunkcnt)
{
	__shared__ unsigned cnt;
	if (threadIdx.x == 0) cnt = 0;
	__syncthreads();
	size_t base = (size_t)blockIdx.x * 1024;
	unsigned c = 0;
	for (unsigned i = threadIdx.x; i < 1024; i += 256) { size_t e = base + i; c += (e < nelem && bif[e] != SBL_NONE); }
	atomicAdd(&cnt, c);
	__syncthreads();
	if (threadIdx.x == 0) chunkcnt[blockIdx.x] = cnt;
}
static __global__ void __launch_bounds__(256) k_write_marks(const unsigned *__restrict__ bif, size_t nelem, const unsigned *__restrict__ chunkoff,
                                                     unsigned *__restrict__ out_elem, unsigned *__restrict__ out_id)
{
	// 256 threads x 4 consecutive elements; wave ballot prefix + per-wave offsets through LDS
	__shared__ unsigned wsum[4];
	size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
	unsigned ids[4], n = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) { size_t e = base + i; ids[i] = e < nelem ? bif[e] : SBL_NONE; n += ids[i] != SBL_NONE; }
	// inclusive scan of n across the wave
	unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, incl = n;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { unsigned v = __shfl_up(incl, d); if (lane >= (unsigned)d) incl += v; }
	if (lane == 63) wsum[wv] = incl;
	__syncthreads();
	unsigned off = chunkoff[blockIdx.x] + incl - n;
	for (unsigned w = 0; w < wv; w++) off += wsum[w];
#pragma unroll
	for (int i = 0; i < 4; i++) if (ids[i] != SBL_NONE) { out_elem[off] = (unsigned)(base + i); out_id[off] = ids[i]; off++; }
}

// chromosome lookup: sepidx[c] = element index of the '$' before chromosome c (nchr+1 entries, ascending)
__device__ __forceinline__ unsigned chr_of(const unsigned *__restrict__ sepidx, unsigned nchr, unsigned e)
{
	unsigned lo = 0, hi = nchr;               // find c with sepidx[c] < e < sepidx[c+1]
	while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (sepidx[mid] < e) lo = mid; else hi = mid; }
	return lo;
}

// (element, id) -> sbl_inst {id, chr, pos}; strand 1 reports reverse-complement coordinates
// (vertexenumeration.cpp:334-346): element e on chromosome c <-> rc position len_c - 1 - local(e).
static __global__ void __launch_bounds__(256) k_make_instances(const unsigned *__restrict__ elem, const unsigned *__restrict__ id, unsigned n,
                                                        const unsigned *__restrict__ sepidx, unsigned nchr, unsigned strand,
                                                        unsigned *__restrict__ out /* n x 3 */)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned e = elem[i], c = chr_of(sepidx, nchr, e);
	unsigned pos = strand == 0 ? e - sepidx[c] - 1 : sepidx[c + 1] - 1 - e;
	out[3 * i] = id[i]; out[3 * i + 1] = c; out[3 * i + 2] = pos;
}
