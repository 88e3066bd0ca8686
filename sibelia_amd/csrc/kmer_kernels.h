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
//   table[cap] 16 B slots {u64 canonical code, u32 prev|next masks, u32 aux}
//
// Integer / indexing work only: no MFMA.  The bound is HBM (random 16-B slot traffic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SBL_NONE 0xFFFFFFFFu
#define SBL_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull   // never a canonical code: min(code, rc(code)) <= rc(TT..T) = 0

struct alignas(16) KmerSlot {
	unsigned long long key;   // canonical 2-bit code
	unsigned int mask;        // bits 0-4: prev {A,C,G,T,#} ; bits 8-12: next {A,C,G,T,#}  (canonical orientation)
	unsigned int aux;         // after classification: index of the bifurcation pair, or SBL_NONE
};

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

// Visit every position g of the tile whose k-window [g, g+k) holds no separator.
// f(g, fwd, rev, prevSym, nextSym) with syms 0..3 = A C G T, 4 = chromosome boundary.
template <class F>
__device__ __forceinline__ void tile_walk(const KmerTile &t, size_t tile, unsigned k, size_t nelem, F f)
{
	const int p0 = threadIdx.x * KM_PER_THREAD;
	const unsigned long long kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
	unsigned long long fwd = 0, rev = 0;
	unsigned have = 0;                          // consecutive non-separator elements ending at the window end
	// prime the window with elements p0 .. p0+k-2
	for (unsigned i = 0; i + 1 < k; i++) {
		int e = p0 + (int)i;
		if (tile_sep(t, e)) { have = 0; fwd = 0; rev = 0; }
		else {
			unsigned b = tile_base(t, e);
			fwd = ((fwd << 2) | b) & kmask;
			rev = (rev >> 2) | ((unsigned long long)(3u - b) << (2 * (k - 1)));
			have++;
		}
	}
	for (int i = 0; i < KM_PER_THREAD; i++) {
		int g = p0 + i, e = g + (int)k - 1;
		if (tile_sep(t, e)) { have = 0; fwd = 0; rev = 0; }
		else {
			unsigned b = tile_base(t, e);
			fwd = ((fwd << 2) | b) & kmask;
			rev = (rev >> 2) | ((unsigned long long)(3u - b) << (2 * (k - 1)));
			have++;
		}
		size_t gg = tile * (size_t)(KM_TILE_WORDS * 32) + (size_t)g;
		if (have >= k && gg < nelem) {
			unsigned ps = tile_sep(t, g - 1) ? 4u : tile_base(t, g - 1);
			unsigned ns = tile_sep(t, e + 1) ? 4u : tile_base(t, e + 1);
			f(gg, fwd, rev, ps, ns);
		}
	}
}

// ---------------------------------------------------------------------------------------------
// K2: k-mer table build.  One atomicCAS (key claim) + one atomicOr (mask merge) per base position;
// a position covers the + occurrence of fwd and the - occurrence of rev, which contribute the same
// bits in canonical orientation.
static __global__ void __launch_bounds__(KM_THREADS) k_kmer_table_build(const unsigned long long *__restrict__ pk,
                                                                 const unsigned *__restrict__ sp, size_t nwords, size_t nelem,
                                                                 unsigned k, KmerSlot *__restrict__ table, unsigned long long capmask,
                                                                 size_t tile_begin, size_t tile_end /* this GPU's slice of tiles */,
                                                                 unsigned *__restrict__ used_count, unsigned *__restrict__ used_slots)
{
	__shared__ KmerTile t;
	for (size_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
		__syncthreads();
		tile_load(t, pk, sp, tile, nwords);
		__syncthreads();
		tile_walk(t, tile, k, nelem, [&](size_t, unsigned long long fwd, unsigned long long rev, unsigned ps, unsigned ns) {
			unsigned long long canon = fwd < rev ? fwd : rev;
			unsigned m = 0;
			// syms: complement of base b is 3-b; '#' (4) stays '#'
			if (fwd <= rev) m |= (1u << ps) | (1u << (8 + ns));
			if (rev <= fwd) m |= (1u << (ns == 4 ? 4 : 3 - ns)) | (1u << (8 + (ps == 4 ? 4 : 3 - ps)));
			unsigned long long h = kmer_hash(canon) & capmask;
			for (;;) {
				unsigned long long old = atomicCAS(&table[h].key, SBL_EMPTY_KEY, canon);
				if (old == SBL_EMPTY_KEY) used_slots[atomicAdd(used_count, 1u)] = (unsigned)h;      // list of claimed slots: classification never scans the sparse table
				if (old == SBL_EMPTY_KEY || old == canon) { atomicOr(&table[h].mask, m); break; }
				h = (h + 1) & capmask;
			}
		});
	}
}

static __global__ void __launch_bounds__(256) k_table_init(KmerSlot *__restrict__ table, size_t cap)
{
	for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += (size_t)gridDim.x * blockDim.x) {
		KmerSlot e; e.key = SBL_EMPTY_KEY; e.mask = 0; e.aux = SBL_NONE;
		table[s] = e;
	}
}

// K3: classify table slots, compact the bifurcation slots and emit their sort keys
// (the canonical code and, unless palindromic, its reverse complement).
// keyinfo payload = 2 * pairIndex + orientation (0 = canonical code, 1 = reverse complement).
static __global__ void __launch_bounds__(256) k_classify_slots(KmerSlot *__restrict__ table, const unsigned *__restrict__ used_slots, unsigned nused, unsigned k,
                                                        unsigned *__restrict__ counters /* [0]=pairs [1]=keys */,
                                                        unsigned long long *__restrict__ keys, unsigned *__restrict__ payload,
                                                        unsigned maxpairs)
{
	for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < nused; u += (size_t)gridDim.x * blockDim.x) {
		size_t s = used_slots[u];
		KmerSlot sl = table[s];
		unsigned aux = SBL_NONE;
		if (mask_is_bifurcation(sl.mask)) {
			unsigned long long r = rc_code(sl.key, k);
			unsigned nk = r == sl.key ? 1u : 2u;
			unsigned pi = atomicAdd(&counters[0], 1u);
			unsigned ki = atomicAdd(&counters[1], nk);
			if (pi < maxpairs && ki + nk <= 2 * maxpairs) {
				aux = pi;
				keys[ki] = sl.key; payload[ki] = 2 * pi;
				if (nk == 2) { keys[ki + 1] = r; payload[ki + 1] = 2 * pi + 1; }
			}
		}
		table[s].aux = aux;
	}
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

// K5: second window pass: look every position's canonical code up and write the dense mark arrays
//   bif[0][g]       = id of the + strand k-mer starting at element g
//   bif[1][g+k-1]   = id of the - strand k-mer starting at element g+k-1 (= reverse complement of the same window)
// (what the marking loop of IndexedSequence::Init builds with AddPoint, reference src/indexedsequence.cpp:49-67).
// Arrays must be pre-filled with SBL_NONE.
static __global__ void __launch_bounds__(KM_THREADS) k_resolve_marks(const unsigned long long *__restrict__ pk, const unsigned *__restrict__ sp,
                                                              size_t nwords, size_t nelem, unsigned k,
                                                              const KmerSlot *__restrict__ table, unsigned long long capmask,
                                                              const unsigned *__restrict__ pairids,
                                                              unsigned *__restrict__ bif0, unsigned *__restrict__ bif1, size_t ntiles)
{
	__shared__ KmerTile t;
	for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
		__syncthreads();
		tile_load(t, pk, sp, tile, nwords);
		__syncthreads();
		tile_walk(t, tile, k, nelem, [&](size_t g, unsigned long long fwd, unsigned long long rev, unsigned, unsigned) {
			unsigned long long canon = fwd < rev ? fwd : rev;
			unsigned long long h = kmer_hash(canon) & capmask;
			for (;;) {
				unsigned long long key = table[h].key;
				if (key == canon) break;
				h = (h + 1) & capmask;            // the key is present: inserted by k_kmer_table_build
			}
			unsigned aux = table[h].aux;
			if (aux != SBL_NONE) {
				unsigned o = fwd <= rev ? 0u : 1u;
				bif0[g] = pairids[2 * aux + o];
				bif1[g + k - 1] = pairids[2 * aux + (o ^ 1u)];
			}
		});
	}
}

// ---------------------------------------------------------------------------------------------
// Hash-prefix sharded enumeration (shard.hip): every GPU scans a contiguous slice of tiles into a
// local pre-aggregating table, ships each distinct canonical k-mer (code + masks) to the GPU that owns its hash
// prefix, owners merge and classify, the bifurcation codes are gathered everywhere and ranked, and every GPU
// resolves its own slice against the (small) bifurcation-only table.
struct alignas(16) KmerRecord { unsigned long long key; unsigned int mask, pad; };

__device__ __forceinline__ unsigned kmer_owner(unsigned long long canon, unsigned nranks)
{
	return (unsigned)(((kmer_hash(canon) >> 32) * (unsigned long long)nranks) >> 32);   // hash PREFIX: independent of the slot index (low bits)
}

// one pass per wave over distinct owners: leader reserves, lanes take consecutive places
template <class F>
__device__ __forceinline__ void wave_group_by(bool act, unsigned key, F f)
{
	unsigned long long todo = __ballot(act);
	unsigned lane = threadIdx.x & 63;
	while (todo) {
		unsigned src = (unsigned)__builtin_ctzll(todo);
		unsigned kk = __shfl(key, src);
		unsigned long long m = __ballot(act && key == kk);
		f(kk, m, act && key == kk, (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1)), lane == src);
		todo &= ~m;
	}
}

static __global__ void __launch_bounds__(256) k_shard_count(const KmerSlot *__restrict__ table, const unsigned *__restrict__ used_slots, unsigned nused,
                                                     unsigned nranks, unsigned *__restrict__ counts)
{
	for (size_t u0 = (size_t)blockIdx.x * blockDim.x; u0 < nused; u0 += (size_t)gridDim.x * blockDim.x) {
		size_t u = u0 + threadIdx.x;
		bool act = u < nused;
		unsigned o = act ? kmer_owner(table[used_slots[u]].key, nranks) : 0u;
		wave_group_by(act, o, [&](unsigned kk, unsigned long long m, bool, unsigned, bool lead) {
			if (lead) atomicAdd(&counts[kk], (unsigned)__builtin_popcountll(m));
		});
	}
}

static __global__ void __launch_bounds__(256) k_shard_scatter(const KmerSlot *__restrict__ table, const unsigned *__restrict__ used_slots, unsigned nused,
                                                       unsigned nranks, const unsigned *__restrict__ offs, unsigned *__restrict__ cursor,
                                                       KmerRecord *__restrict__ send)
{
	for (size_t u0 = (size_t)blockIdx.x * blockDim.x; u0 < nused; u0 += (size_t)gridDim.x * blockDim.x) {
		size_t u = u0 + threadIdx.x;
		bool act = u < nused;
		KmerSlot sl; sl.key = 0; sl.mask = 0;
		if (act) sl = table[used_slots[u]];
		unsigned o = act ? kmer_owner(sl.key, nranks) : 0u;
		wave_group_by(act, o, [&](unsigned kk, unsigned long long m, bool mine, unsigned place, bool lead) {
			unsigned base = 0;
			if (lead) base = atomicAdd(&cursor[kk], (unsigned)__builtin_popcountll(m));
			base = __shfl(base, (unsigned)__builtin_ctzll(m));
			if (mine) { KmerRecord r; r.key = sl.key; r.mask = sl.mask; r.pad = 0; send[(size_t)offs[kk] + base + place] = r; }
		});
	}
}

// owner side: OR the received masks into the owner's table
static __global__ void __launch_bounds__(256) k_shard_merge(const KmerRecord *__restrict__ recv, size_t nrecv, KmerSlot *__restrict__ table,
                                                     unsigned long long capmask, unsigned *__restrict__ used_count, unsigned *__restrict__ used_slots)
{
	for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrecv; i += (size_t)gridDim.x * blockDim.x) {
		KmerRecord r = recv[i];
		unsigned long long h = kmer_hash(r.key) & capmask;
		for (;;) {
			unsigned long long old = atomicCAS(&table[h].key, SBL_EMPTY_KEY, r.key);
			if (old == SBL_EMPTY_KEY) used_slots[atomicAdd(used_count, 1u)] = (unsigned)h;
			if (old == SBL_EMPTY_KEY || old == r.key) { atomicOr(&table[h].mask, r.mask); break; }
			h = (h + 1) & capmask;
		}
	}
}

// bifurcation-only table from the globally sorted strand-specific codes: slot.mask = id of the canonical code,
// slot.aux = id of its reverse complement (the same id for a palindrome)
static __global__ void __launch_bounds__(256) k_bif_table_build(const unsigned long long *__restrict__ skeys, unsigned nkeys, unsigned k,
                                                         KmerSlot *__restrict__ table, unsigned long long capmask)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nkeys) return;
	unsigned long long key = skeys[i], r = rc_code(key, k), canon = key < r ? key : r;
	unsigned long long h = kmer_hash(canon) & capmask;
	for (;;) {
		unsigned long long old = atomicCAS(&table[h].key, SBL_EMPTY_KEY, canon);
		if (old == SBL_EMPTY_KEY || old == canon) break;
		h = (h + 1) & capmask;
	}
	if (key == canon) table[h].mask = i;
	if (r == canon) table[h].aux = i;
}

// K5 against the bifurcation-only table, over this GPU's slice of tiles
static __global__ void __launch_bounds__(KM_THREADS) k_resolve_marks_bif(const unsigned long long *__restrict__ pk, const unsigned *__restrict__ sp,
                                                                  size_t nwords, size_t nelem, unsigned k,
                                                                  const KmerSlot *__restrict__ table, unsigned long long capmask,
                                                                  unsigned *__restrict__ bif0, unsigned *__restrict__ bif1,
                                                                  size_t tile_begin, size_t tile_end)
{
	__shared__ KmerTile t;
	for (size_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
		__syncthreads();
		tile_load(t, pk, sp, tile, nwords);
		__syncthreads();
		tile_walk(t, tile, k, nelem, [&](size_t g, unsigned long long fwd, unsigned long long rev, unsigned, unsigned) {
			unsigned long long canon = fwd < rev ? fwd : rev;
			unsigned long long h = kmer_hash(canon) & capmask;
			for (;;) {
				KmerSlot sl = table[h];
				if (sl.key == canon) {
					bool o = fwd <= rev;
					bif0[g] = o ? sl.mask : sl.aux;
					bif1[g + k - 1] = o ? sl.aux : sl.mask;
					break;
				}
				if (sl.key == SBL_EMPTY_KEY) break;     // not a bifurcation
				h = (h + 1) & capmask;
			}
		});
	}
}

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
