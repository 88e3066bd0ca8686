// kmer_bucket_kernels.h -- radix-bucketed k-mer table for bifurcation enumeration at k <= 32 (single-GPU path).
//
// Replaces the suffix-array + LCP group scan of IndexedSequence::EnumerateBifurcationsSArrayInRAM
// (reference src/vertexenumeration.cpp:288-362) -- same order-free formulation as kmer_kernels.h -- with a table that never
// makes a random access to HBM:
//   B1 k_kmer_records   every base position -> one 16-B record {key = mix64(canonical code), value = element | masks | orientation},
//                       written in position order (coalesced); mix64 is a bijection, so equal keys <=> equal k-mers
//   B2 radix partition  by `bits` bits of the key = hash PREFIX (rocPRIM onesweep, 8 bits per pass): buckets of ~512 records,
//                       contiguous in HBM, each the size of an LDS table  (north_star: "radix-bucketed open-address hash")
//   B3 k_bucket_classify one workgroup per bucket: open-addressing table in LDS (ds atomics), masks OR-ed per distinct k-mer,
//                       Bifurcation() test (vertexenumeration.cpp:67-70,:330), sort keys of the bifurcation k-mers for the
//                       global ranking (id = lexicographic rank, :348-355) and the list of member positions (element, pair)
//   B4 rocPRIM sort of the bifurcation codes + k_scatter_ids (kmer_kernels.h)
//   B5 k_scatter_members bif[0][g] / bif[1][g+k-1] for the member positions only (marking, indexedsequence.cpp:49-67)
// HBM traffic: 16 B written + ~(8 + 2 x 32) B sort + 16 B read per position, all streaming; nothing else scales with N.
// The same hash prefix is the shard key of the multi-GPU path (shard.hip: owner = prefix x ranks >> 32).
#pragma once
#include "kmer_kernels.h"

// inverse of kmer_hash (murmur3 fmix64 is a bijection on 64-bit words)
__device__ __host__ __forceinline__ unsigned long long kmer_unhash(unsigned long long x)
{
	x ^= x >> 33; x *= 0x9cb4b2f8129337dbull;
	x ^= x >> 33; x *= 0x4f74430c22a54005ull;
	x ^= x >> 33;
	return x;
}

// Empty-slot marker of the LDS tables: mix64(~0).  ~0 is never a CANONICAL code (for k = 32 it is TT..T, whose reverse complement
// AA..A = 0 is smaller; for k < 32 it is not a code at all) and mix64 is a bijection, so no valid record carries this key.
#define KB_EMPTY_KEY 0x64b5720b4b825f21ull
#define KB_INVALID 0xFFFFFFFFFFFFFFFFull       // value of a position whose k-window holds a separator (or lies beyond the input)
// value layout: bits 0-31 element index g | bits 32-44 mask in canonical orientation (bits 0-4: prev {A,C,G,T,#}, bits 8-12: next) | bit 48 fwd <= rev | bit 49 rev <= fwd

// B1: one record per element index of the tile (invalid positions get KB_INVALID), coalesced 8-B stores.
static __global__ void __launch_bounds__(KM_THREADS) k_kmer_records(const unsigned long long *__restrict__ pk, const unsigned *__restrict__ sp,
                                                             size_t nwords, size_t nelem, unsigned k, size_t tile_begin, size_t tile_end /* this GPU's slice of tiles */,
                                                             unsigned long long *__restrict__ keys, unsigned long long *__restrict__ vals)
{
	__shared__ KmerTile t;
	const unsigned long long kshift = 64 - 2 * k;
	const size_t out0 = tile_begin * (size_t)(KM_TILE_WORDS * 32);      // records are stored relative to the slice
	for (size_t tile = tile_begin + blockIdx.x; tile < tile_end; tile += gridDim.x) {
		__syncthreads();
		tile_load(t, pk, sp, tile, nwords);
		__syncthreads();
		const size_t base = tile * (size_t)(KM_TILE_WORDS * 32);
#pragma unroll 4
		for (int i = 0; i < KM_PER_THREAD; i++) {
			const int e = i * KM_THREADS + (int)threadIdx.x;          // element relative to the tile start
			const size_t g = base + (size_t)e;
			const int j = (e + 32) >> 5, o = (e + 32) & 31;
			// k consecutive elements starting at e: at most two packed words
			const unsigned long long w0 = t.w[j], w1 = t.w[j + 1];
			const unsigned long long x = o ? (w0 << (2 * o)) | (w1 >> (64 - 2 * o)) : w0;
			const unsigned long long sx = (((unsigned long long)t.s[j] << 32) | t.s[j + 1]) << o;
			const bool valid = g < nelem && (sx >> (64 - k)) == 0;
			unsigned long long key = kmer_hash((unsigned long long)g), val = KB_INVALID;      // invalid records: spread over the buckets, skipped by value
			if (valid) {
				const unsigned long long fwd = x >> kshift, rev = rc_code(fwd, k);
				const unsigned ps = tile_sep(t, e - 1) ? 4u : tile_base(t, e - 1);
				const unsigned ns = tile_sep(t, e + (int)k) ? 4u : tile_base(t, e + (int)k);
				unsigned m = 0, fl = 0;
				// syms: complement of base b is 3-b; '#' (4) stays '#'
				if (fwd <= rev) { m |= (1u << ps) | (1u << (8 + ns)); fl |= 1u; }
				if (rev <= fwd) { m |= (1u << (ns == 4 ? 4 : 3 - ns)) | (1u << (8 + (ps == 4 ? 4 : 3 - ps))); fl |= 2u; }
				key = kmer_hash(fwd < rev ? fwd : rev);
				val = (unsigned long long)(unsigned)g | ((unsigned long long)m << 32) | ((unsigned long long)fl << 48);
			}
			keys[g - out0] = key;
			vals[g - out0] = val;
		}
	}
}

// bucket b = records whose key's LOW `bits` bits equal b: boff[b] = first record (lower bound in the partitioned array).
// (The low end of the mixed key is as good a hash prefix as the high end; rocPRIM 4.2's radix_sort_pairs returns unsorted,
// mismatched pairs for begin_bit > 0 below ~1 M items -- tools/dbg/sort_dbg.hip -- so the partition sorts bits [0, bits).)
static __global__ void __launch_bounds__(256) k_bucket_bounds(const unsigned long long *__restrict__ skeys, size_t n, unsigned bits, unsigned *__restrict__ boff)
{
	const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nb = (size_t)1 << bits;
	if (b > nb) return;
	if (b == nb) { boff[b] = (unsigned)n; return; }
	size_t lo = 0, hi = n;
	while (lo < hi) { size_t mid = (lo + hi) >> 1; if ((skeys[mid] & (nb - 1)) < b) lo = mid + 1; else hi = mid; }
	boff[b] = (unsigned)lo;
}

#ifndef KB_SLOTS
#define KB_SLOTS 1024u                          // LDS table of a bucket: 1024 x (8 + 4 + 4) B = 16 KB
#endif
#define KB_THREADS 256
// Threads re-read the count before every record, so at most KB_MAX_DISTINCT + KB_THREADS slots are ever claimed: that sum must
// leave the table room, or a probe for an absent key would spin in a full table.
#define KB_MAX_DISTINCT (KB_SLOTS * 23u / 32u)
static_assert(KB_MAX_DISTINCT + KB_THREADS < KB_SLOTS, "k_bucket_classify: the LDS table could fill up");
// counters: pairs, keys, members, overflow flag -- one 128-B line each: an address (or a line) takes only ~88 returning atomics per
// microsecond, whoever issues them
enum { KB_CTR_PAIRS = 0, KB_CTR_KEYS = 32, KB_CTR_MEM = 64, KB_CTR_FLAG = 96, KB_CTR_WORDS = 128 };
// A workgroup takes KB_GROUP consecutive buckets and STAGES what they emit in LDS: one reservation of output ranges per flush instead
// of three per bucket.  (Round 2 reserved per bucket: 65 536 workgroups x 3 returning atomics = 0.74 ms of serialised atomics on each
// of the three counter lines -- the whole 0.84 ms of the kernel, which streams 589 MB.)
#define KB_GROUP 16u
#define KB_REGS 3                               // records per thread held in registers (buckets of up to 768 records are read once)
#define KB_STAGE_MEM 1536u                      // staged member positions (8 B); flushed before a bucket when more than half full
#define KB_STAGE_KEYS 384u                      // staged sort keys (8 + 4 B); a bucket with more keys than fit writes them directly
// B3: see the header comment.  Capacities (maxpairs, maxmembers) guard the writes; the host re-runs with larger buffers / more
// bucket bits when a counter exceeds them or the overflow flag is set.
static __global__ void __launch_bounds__(KB_THREADS) k_bucket_classify(const unsigned long long *__restrict__ skeys, const unsigned long long *__restrict__ svals,
                                                                const unsigned *__restrict__ boff, unsigned nbuckets, unsigned k,
                                                                unsigned *__restrict__ counters,
                                                                unsigned long long *__restrict__ rank_keys, unsigned *__restrict__ rank_payload, unsigned maxpairs,
                                                                unsigned long long *__restrict__ members, unsigned maxmembers)
{
	__shared__ unsigned long long tkey[KB_SLOTS];
	__shared__ unsigned tmask[KB_SLOTS];
	__shared__ unsigned taux[KB_SLOTS];
	__shared__ unsigned long long st_mem[KB_STAGE_MEM];      // element | (2 * staged pair + orientation) << 32
	__shared__ unsigned long long st_key[KB_STAGE_KEYS];
	__shared__ unsigned st_pay[KB_STAGE_KEYS];               // 2 * staged pair + (0: canonical, 1: reverse complement)
	__shared__ unsigned s_used, s_pairs, s_keys, s_np, s_nk, s_nm, s_bpairs, s_bkeys, s_bmem, s_dpairs, s_dkeys;
	const unsigned b0 = blockIdx.x * KB_GROUP, b1 = b0 + KB_GROUP < nbuckets ? b0 + KB_GROUP : nbuckets;
	if (b0 >= nbuckets) return;
	if (threadIdx.x == 0) { s_np = 0; s_nk = 0; s_nm = 0; }
	__syncthreads();
	// everything staged goes out: one reservation per output array, then coalesced stores with the bases added
	auto flush = [&]() {
		__syncthreads();
		const unsigned np = s_np, nk = s_nk, nm = s_nm < KB_STAGE_MEM ? s_nm : KB_STAGE_MEM;
		if (threadIdx.x == 0) {
			s_bpairs = np ? atomicAdd(&counters[KB_CTR_PAIRS], np) : 0u;
			s_bkeys = nk ? atomicAdd(&counters[KB_CTR_KEYS], nk) : 0u;
			s_bmem = nm ? atomicAdd(&counters[KB_CTR_MEM], nm) : 0u;
		}
		__syncthreads();
		const unsigned bp = s_bpairs, bk = s_bkeys, bm = s_bmem;
		if (bp + np <= maxpairs && (size_t)bk + nk <= 2 * (size_t)maxpairs)
			for (unsigned i = threadIdx.x; i < nk; i += KB_THREADS) { rank_keys[bk + i] = st_key[i]; rank_payload[bk + i] = st_pay[i] + 2 * bp; }
		for (unsigned i = threadIdx.x; i < nm; i += KB_THREADS)
			if (bm + i < maxmembers) members[bm + i] = st_mem[i] + ((unsigned long long)(2 * bp) << 32);
		__syncthreads();
		if (threadIdx.x == 0) { s_np = 0; s_nk = 0; s_nm = 0; }
		__syncthreads();
	};
	// the first bucket's records: in flight while the table is cleared
	unsigned long long rk[KB_REGS], rv[KB_REGS];
	unsigned lo = boff[b0], hi = boff[b0 + 1];
#pragma unroll
	for (int r = 0; r < KB_REGS; r++) {
		const unsigned i = lo + threadIdx.x + r * KB_THREADS;
		rv[r] = i < hi ? svals[i] : KB_INVALID; rk[r] = i < hi ? skeys[i] : 0ull;
	}
	for (unsigned b = b0; b < b1; b++) {
		// the next bucket's records are requested before this one is worked on
		unsigned long long nk_[KB_REGS], nv_[KB_REGS];
		unsigned nlo = 0, nhi = 0;
		if (b + 1 < b1) {
			nlo = hi; nhi = boff[b + 2];
#pragma unroll
			for (int r = 0; r < KB_REGS; r++) {
				const unsigned i = nlo + threadIdx.x + r * KB_THREADS;
				nv_[r] = i < nhi ? svals[i] : KB_INVALID; nk_[r] = i < nhi ? skeys[i] : 0ull;
			}
		}
		if (lo < hi) {
			// room for what the register-held part of this bucket can emit (a larger bucket writes directly)
			__syncthreads();
			if (s_nm > KB_STAGE_MEM / 2) flush();
			for (unsigned i = threadIdx.x; i < KB_SLOTS; i += KB_THREADS) { tkey[i] = KB_EMPTY_KEY; tmask[i] = 0; taux[i] = SBL_NONE; }
			if (threadIdx.x == 0) { s_used = 0; s_pairs = 0; s_keys = 0; }
			__syncthreads();
			// ---- insert: one ds cmpswap (key claim) + one ds or (mask merge) per record.  At most KB_MAX_DISTINCT + KB_THREADS slots
			// are ever claimed (every thread re-reads the count before each record), so a probe always finds a free slot or its key.
			auto insert = [&](unsigned long long key, unsigned long long v) {
				unsigned h = (unsigned)(key >> 44) & (KB_SLOTS - 1);            // bits above the bucket prefix (<= 40 bits)
				for (unsigned step = 0; step < KB_SLOTS; step++) {               // bounded: a full table ends in the overflow flag, never in a spin
					unsigned long long old = atomicCAS(&tkey[h], KB_EMPTY_KEY, key);
					if (old == KB_EMPTY_KEY) atomicAdd(&s_used, 1u);
					if (old == KB_EMPTY_KEY || old == key) { atomicOr(&tmask[h], (unsigned)(v >> 32) & 0x1FFFu); break; }
					h = (h + 1) & (KB_SLOTS - 1);
					if (step + 1 == KB_SLOTS) atomicAdd(&s_used, KB_SLOTS);
				}
			};
#pragma unroll
			for (int r = 0; r < KB_REGS; r++) {
				if (*(volatile unsigned *)&s_used > KB_MAX_DISTINCT) break;
				if (rv[r] != KB_INVALID) insert(rk[r], rv[r]);
			}
			for (unsigned i = lo + threadIdx.x + KB_REGS * KB_THREADS; i < hi; i += KB_THREADS) {
				if (*(volatile unsigned *)&s_used > KB_MAX_DISTINCT) break;    // too many distinct k-mers for this table: the host re-buckets
				const unsigned long long v = svals[i];
				if (v != KB_INVALID) insert(skeys[i], v);
			}
			__syncthreads();
			if (s_used > KB_MAX_DISTINCT) { if (threadIdx.x == 0) atomicOr(&counters[KB_CTR_FLAG], 1u); return; }   // (uniform; the host discards everything)
			// ---- classify the distinct k-mers of the bucket
			for (unsigned sidx = threadIdx.x; sidx < KB_SLOTS; sidx += KB_THREADS) {
				if (tkey[sidx] == KB_EMPTY_KEY || !mask_is_bifurcation(tmask[sidx])) continue;
				const unsigned long long canon = kmer_unhash(tkey[sidx]);
				const unsigned nk = rc_code(canon, k) == canon ? 1u : 2u;
				const unsigned lp = atomicAdd(&s_pairs, 1u), lk = atomicAdd(&s_keys, nk);
				taux[sidx] = lp | (lk << 12);                       // lp < 2^11, lk < 2^12
			}
			__syncthreads();
			const unsigned npairs = s_pairs, nkeys = s_keys;
			if (npairs) {
				// sort keys and members are staged when they fit (flushing first if need be); a bucket with more keys than the stage holds,
				// or with more records than the registers hold (low-complexity input), reserves its output ranges itself
				const bool direct = nkeys > KB_STAGE_KEYS || hi - lo > KB_REGS * KB_THREADS;
				if (!direct && s_nk + nkeys > KB_STAGE_KEYS) flush();
				if (direct) {
					flush();                                        // pair indices below are final, nothing staged refers to them
					if (threadIdx.x == 0) { s_dpairs = atomicAdd(&counters[KB_CTR_PAIRS], npairs); s_dkeys = atomicAdd(&counters[KB_CTR_KEYS], nkeys); }
					__syncthreads();
				}
				const unsigned pbase = direct ? s_dpairs : s_np, kbase = direct ? s_dkeys : s_nk;
				for (unsigned sidx = threadIdx.x; sidx < KB_SLOTS; sidx += KB_THREADS) {
					const unsigned a = taux[sidx];
					if (a == SBL_NONE) continue;
					const unsigned pi = pbase + (a & 0xFFFu), ki = kbase + (a >> 12);
					const unsigned long long canon = kmer_unhash(tkey[sidx]), r = rc_code(canon, k);
					if (direct) {
						if (pi < maxpairs && ki + 2 <= 2 * maxpairs) {
							rank_keys[ki] = canon; rank_payload[ki] = 2 * pi;
							if (r != canon) { rank_keys[ki + 1] = r; rank_payload[ki + 1] = 2 * pi + 1; }
						}
					} else {
						st_key[ki] = canon; st_pay[ki] = 2 * pi;
						if (r != canon) { st_key[ki + 1] = r; st_pay[ki + 1] = 2 * pi + 1; }
					}
					taux[sidx] = pi;                                // staged pair index, or the final one (direct)
				}
				__syncthreads();
				if (threadIdx.x == 0 && !direct) { s_np += npairs; s_nk += nkeys; }
				// ---- member positions of the bifurcation k-mers.  Staged entries carry the STAGED pair index (the flush adds the base);
				// after a direct reservation the index is final and the flush must add nothing: those members go out directly too.
				auto probe = [&](unsigned long long key) -> unsigned {
					unsigned h = (unsigned)(key >> 44) & (KB_SLOTS - 1);
					while (tkey[h] != key) h = (h + 1) & (KB_SLOTS - 1);
					return taux[h];
				};
				auto emit = [&](unsigned long long key, unsigned long long v, bool have) {
					const unsigned pi = have ? probe(key) : SBL_NONE;
					const bool mem = pi != SBL_NONE;
					// payload of the code this position spells on the + strand: 2 * pair + (0: canonical, 1: reverse complement)
					const unsigned long long rec = (v & 0xFFFFFFFFull) | ((unsigned long long)(2 * pi + (((v >> 48) & 1ull) ? 0u : 1u)) << 32);
					const unsigned long long bal = __ballot(mem);
					if (!bal) return;
					const unsigned lane = threadIdx.x & 63u;
					unsigned base = 0;
					if (lane == 0) base = direct ? atomicAdd(&counters[KB_CTR_MEM], (unsigned)__popcll(bal)) : atomicAdd(&s_nm, (unsigned)__popcll(bal));
					base = __shfl(base, 0);
					const unsigned at = base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
					if (mem) { if (direct) { if (at < maxmembers) members[at] = rec; } else if (at < KB_STAGE_MEM) st_mem[at] = rec; }
				};
#pragma unroll
				for (int r = 0; r < KB_REGS; r++) emit(rk[r], rv[r], rv[r] != KB_INVALID);
				// the rest of a bucket larger than the registers hold (always direct)
				for (unsigned i0 = lo + KB_REGS * KB_THREADS; i0 < hi; i0 += KB_THREADS) {
					const unsigned i = i0 + threadIdx.x;
					const bool in = i < hi;
					const unsigned long long v = in ? svals[i] : KB_INVALID;
					emit(in ? skeys[i] : 0ull, v, v != KB_INVALID);
				}
			}
		}
		lo = nlo; hi = nhi;
#pragma unroll
		for (int r = 0; r < KB_REGS; r++) { rk[r] = nk_[r]; rv[r] = nv_[r]; }
	}
	flush();
}

// B5: marks of the member positions.  bif[0][g] = id of the + strand k-mer starting at g, bif[1][g+k-1] = id of its reverse
// complement (the - strand k-mer starting at g+k-1); arrays pre-filled with SBL_NONE.
static __global__ void __launch_bounds__(256) k_scatter_members(const unsigned long long *__restrict__ members, unsigned n, unsigned k,
                                                         const unsigned *__restrict__ pairids, unsigned *__restrict__ bif0, unsigned *__restrict__ bif1)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned long long m = members[i];
	const unsigned g = (unsigned)m, p = (unsigned)(m >> 32);
	bif0[g] = pairids[p];
	bif1[g + k - 1] = pairids[p ^ 1u];
}

// ---- multi-GPU pieces (shard.hip): owner of bucket b among R ranks = (b * R) >> bits (contiguous bucket ranges)
// rank of the owner's own bifurcation codes in the globally sorted list = id; pairids[payload] as k_scatter_ids leaves it
static __global__ void __launch_bounds__(256) k_rank_own_keys(const unsigned long long *__restrict__ mykeys, const unsigned *__restrict__ mypayload, unsigned nmine,
                                                       const unsigned long long *__restrict__ allsorted, unsigned nall, unsigned k, unsigned *__restrict__ pairids)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nmine) return;
	const unsigned long long key = mykeys[i];
	unsigned lo = 0, hi = nall;
	while (lo < hi) { unsigned mid = (lo + hi) >> 1; if (allsorted[mid] < key) lo = mid + 1; else hi = mid; }
	const unsigned p = mypayload[i];
	pairids[p] = lo;
	if (!(p & 1) && rc_code(key, k) == key) pairids[p + 1] = lo;      // palindrome: one vertex for both orientations
}
// member positions of the owner's buckets -> the two marks each of them sets (what k_scatter_members writes locally)
static __global__ void __launch_bounds__(256) k_member_marks(const unsigned long long *__restrict__ members, unsigned n, unsigned k, const unsigned *__restrict__ pairids,
                                                      unsigned *__restrict__ elem0, unsigned *__restrict__ id0, unsigned *__restrict__ elem1, unsigned *__restrict__ id1)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned long long m = members[i];
	const unsigned g = (unsigned)m, p = (unsigned)(m >> 32);
	elem0[i] = g; id0[i] = pairids[p];
	elem1[i] = g + k - 1; id1[i] = pairids[p ^ 1u];
}
