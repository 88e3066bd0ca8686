// longk_fp.hip -- bifurcation enumeration for vertex sizes k > 32 through WINDOW FINGERPRINTS and the radix-bucketed table of the
// k <= 32 path, made exact by verification ("fingerprint slots 32 B ... + k/4 B per verified occurrence").
//
// Replaces IndexedSequence::EnumerateBifurcationsSArrayInRAM (reference src/vertexenumeration.cpp:263-364) for k > 32, like the exact
// rank doubling of longk.hip (kept as the fall-back and as the A/B: SBL_LONGK_DOUBLING=1), in a handful of streaming passes instead
// of ~40 radix sorts of every suffix:
//   F1 k_fp_tiles     per tile of 1024 elements: the polynomial hash of the tile (two hash functions x two directions) and of its
//                     first r / last T - r symbols (r = k mod T) -- 0.25 B read per element
//   F2 k_fp_chunks / k_fp_carries / k_fp_apply  prefix (left to right) and suffix (right to left) hashes at the tile boundaries
//   F3 k_fp_records   per window start g: F(w) = P[g+k] - P[g] B^k and F(rc(w)) = 3 G_k - (S[g] - B^k S[g+k]) from block scans of the
//                     tile at g and of the tile range at g + k (nothing per element is kept in HBM); the record of the CANONICAL
//                     orientation (smaller fingerprint pair): {mix64(h1 | 3 bits of h2), element | prev/next masks | orientation | 20 bits of h2} -- 16 B written
//   F4 partition      by the hash prefix of the first key, as at k <= 32
//   F5 k_fp_classify  one LDS open-addressing table per bucket keyed by the 84-bit fingerprint (slot claimed on the 64-bit key by ds cmpswap,
//                     identity settled on the 20 further bits), masks OR-ed, Bifurcation() test (vertexenumeration.cpp:67-70,:330) per
//                     distinct fingerprint, representative window + member positions of the bifurcation k-mers
//   F6 k_fp_verify    EVERY member of a bifurcation group is compared with the group's representative on the 2-bit sequence
//                     (k / 4 B per occurrence).  Two different k-mers with one fingerprint can only MERGE groups (masks are OR-ed:
//                     bits are added, never lost), so an unverified table has false positives only, and all positives are verified:
//                     a mismatch (never observed: 2^-84 per pair of windows) abandons the path and the exact rank doubling runs.
//   F7 ranking        ids = lexicographic rank among the bifurcation k-mers (:348-355) -- a few per cent of the windows at most:
//                     MSD refinement over chunks of <= 27 symbols of the representatives (library sorts on this SUBSET only)
//   F8 k_fp_marks     bif[0][g] / bif[1][g+k-1] of the member positions (marking, indexedsequence.cpp:49-67)
// Hash functions: h1 = sum x_i B1^(k-1-i) mod 2^61 - 1, h2 = sum x_i B2^(k-1-i) mod 2^64 (odd B2): the second one costs three
// 32-bit multiplies, and its known weakness (Thue-Morse strings) is not shared by the first.
// Integer work only: no MFMA.
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

#include "sbl_ctx.h"
#include "sbl_comm.h"
#include "kmer_bucket_kernels.h"
#include <chrono>

typedef unsigned long long u64;
static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------- fingerprint arithmetic
#define FP_M61 0x1FFFFFFFFFFFFFFFull
struct Fp { u64 a, b; };                          // a: mod 2^61 - 1, b: mod 2^64
__host__ __device__ __forceinline__ u64 mul61(u64 x, u64 y)
{
#ifdef __HIP_DEVICE_COMPILE__
	const u64 lo = x * y, hi = __umul64hi(x, y);
#else
	const unsigned __int128 p = (unsigned __int128)x * y;
	const u64 lo = (u64)p, hi = (u64)(p >> 64);
#endif
	u64 r = (lo & FP_M61) + ((lo >> 61) | (hi << 3));      // 2^61 = 1 (mod M): x y = (hi 2^3 + lo >> 61) 2^61 + (lo & M)
	r = (r & FP_M61) + (r >> 61);
	return r >= FP_M61 ? r - FP_M61 : r;
}
__host__ __device__ __forceinline__ u64 add61(u64 x, u64 y) { const u64 r = x + y; return r >= FP_M61 ? r - FP_M61 : r; }
__host__ __device__ __forceinline__ u64 sub61(u64 x, u64 y) { return x >= y ? x - y : x + FP_M61 - y; }
__host__ __device__ __forceinline__ Fp fp_mul(Fp x, Fp y) { return Fp{mul61(x.a, y.a), x.b * y.b}; }
__host__ __device__ __forceinline__ Fp fp_add(Fp x, Fp y) { return Fp{add61(x.a, y.a), x.b + y.b}; }
__host__ __device__ __forceinline__ Fp fp_sub(Fp x, Fp y) { return Fp{sub61(x.a, y.a), x.b - y.b}; }
__host__ __device__ __forceinline__ Fp fp_horner(Fp h, Fp base, unsigned s) { return Fp{add61(mul61(h.a, base.a), (u64)s), h.b * base.b + s}; }   // h B + s

#define FP_TILE 512u                              // elements per tile = one wave x FP_RUN (the scans inside a tile are wave scans: no LDS, no barrier)
#define FP_RUN 8u
#define FP_THREADS 256u                           // four tiles per workgroup
// constants of one enumeration (host-computed, passed by value)
struct FpConst {
	Fp B;                                         // the bases
	Fp c2[6];                                     // (B^FP_RUN)^(2^i), i = 0 .. 5: steps of the wave scans
	Fp c64;                                       // (B^FP_RUN)^64: from wave to wave
	Fp Bk, G3;                                    // B^k; 3 (B^k - 1) / (B - 1) = 3 sum_{j<k} B^j
	Fp Br, BTr;                                   // B^r, B^(T - r) with r = k mod T
	Fp t2[6], t64, tT;                            // the same steps for the scan over tiles: base B^T, and (B^T)^256 for its carry
	Fp tseg;                                      // ((B^T)^256)^seg: weight of one lane's segment of chunks in k_fp_carries
	unsigned k, q, r;                             // k = q T + r
};

__device__ __forceinline__ Fp fp_shfl(Fp x, int src) { return Fp{(u64)__shfl((long long)x.a, src), (u64)__shfl((long long)x.b, src)}; }
__device__ __forceinline__ Fp fp_shfl_up(Fp x, unsigned d) { return Fp{(u64)__shfl_up((long long)x.a, d), (u64)__shfl_up((long long)x.b, d)}; }
__device__ __forceinline__ Fp fp_shfl_down(Fp x, unsigned d) { return Fp{(u64)__shfl_down((long long)x.a, d), (u64)__shfl_down((long long)x.b, d)}; }

// Scan of 256 chunk hashes, one per thread, all chunks of the same length (weight C per chunk, its powers c2[] / c64):
//   DIR = +1   excl[t] = sum_{u < t} x_u C^(t-1-u)      total = sum_u x_u C^(255-u)        (prefix: left to right)
//   DIR = -1   excl[t] = sum_{u > t} x_u C^(u-t-1)      total = sum_u x_u C^u              (suffix: right to left)
// lds: 8 Fp of scratch.  All 256 threads call it.
template <int DIR>
__device__ __forceinline__ Fp scan256(Fp x, const Fp *c2, Fp c64, Fp *lds, Fp &total)
{
	const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	Fp inc = x;
#pragma unroll
	for (int i = 0; i < 6; i++) {
		const unsigned d = 1u << i;
		const Fp y = DIR > 0 ? fp_shfl_up(inc, d) : fp_shfl_down(inc, d);
		const bool in = DIR > 0 ? lane >= d : lane + d < 64u;
		if (in) inc = fp_add(inc, fp_mul(y, c2[i]));
	}
	// inc: inclusive within the wave.  Exclusive within the wave: the neighbour's inclusive value
	Fp exc = DIR > 0 ? fp_shfl_up(inc, 1) : fp_shfl_down(inc, 1);
	if (DIR > 0 ? lane == 0 : lane == 63u) exc = Fp{0, 0};
	if (DIR > 0 ? lane == 63u : lane == 0) lds[wv] = inc;                 // the wave's total
	__syncthreads();
	// what the waves before (after) this one contribute: sum over them, each a further C^64 away; then C^(distance inside the wave)
	Fp carry{0, 0};
	if (DIR > 0) { for (unsigned v = 0; v < wv; v++) carry = fp_add(fp_mul(carry, c64), lds[v]); }
	else { for (unsigned v = 3; v > wv; v--) carry = fp_add(fp_mul(carry, c64), lds[v]); }
	Fp tot{0, 0};
	if (DIR > 0) { for (unsigned v = 0; v < 4; v++) tot = fp_add(fp_mul(tot, c64), lds[v]); }
	else { for (unsigned v = 4; v-- > 0;) tot = fp_add(fp_mul(tot, c64), lds[v]); }
	total = tot;
	// C^(number of chunks between the wave boundary and this thread): a product scan of C over the lanes
	Fp pw = c2[0];                                                     // C^(l+1) after the scan (l = distance from the wave's first chunk)
#pragma unroll
	for (int i = 0; i < 6; i++) {
		const unsigned d = 1u << i;
		const Fp y = DIR > 0 ? fp_shfl_up(pw, d) : fp_shfl_down(pw, d);
		const bool in = DIR > 0 ? lane >= d : lane + d < 64u;
		if (in) pw = fp_mul(pw, y);
	}
	// exclusive: the carry is C^dist away from this thread's exclusive value, dist = chunks of this wave before (after) the thread
	Fp pwe = DIR > 0 ? fp_shfl_up(pw, 1) : fp_shfl_down(pw, 1);         // C^dist for dist >= 1
	if (DIR > 0 ? lane == 0 : lane == 63u) pwe = Fp{1, 1};
	__syncthreads();                                                   // (lds is reused by the caller)
	return fp_add(exc, fp_mul(carry, pwe));
}

// The same over the 64 chunks of ONE wave (a tile): shuffles only.  pwl = C^lane (DIR = +1) / C^(63 - lane) (DIR = -1) is not needed here --
// there is no carry from outside the wave; the caller adds its seed.
template <int DIR>
__device__ __forceinline__ Fp scan64(Fp x, const Fp *c2, Fp &total)
{
	const unsigned lane = threadIdx.x & 63u;
	Fp inc = x;
#pragma unroll
	for (int i = 0; i < 6; i++) {
		const unsigned d = 1u << i;
		const Fp y = DIR > 0 ? fp_shfl_up(inc, d) : fp_shfl_down(inc, d);
		const bool in = DIR > 0 ? lane >= d : lane + d < 64u;
		if (in) inc = fp_add(inc, fp_mul(y, c2[i]));
	}
	Fp exc = DIR > 0 ? fp_shfl_up(inc, 1) : fp_shfl_down(inc, 1);
	if (DIR > 0 ? lane == 0 : lane == 63u) exc = Fp{0, 0};
	total = fp_shfl(inc, DIR > 0 ? 63 : 0);
	return exc;
}

// FP_RUN symbols from element e (2 bit each, first in the high bits of the result); elements beyond the packed array read 0
__device__ __forceinline__ unsigned fp_syms(const u64 *__restrict__ pk, size_t nwords, size_t e)
{
	const size_t w = e >> 5; const unsigned o = (unsigned)(e & 31u);
	const u64 w0 = w < nwords ? pk[w] : 0ull;
	u64 x = w0 << (2 * o);
	if (o > 32u - FP_RUN) { const u64 w1 = w + 1 < nwords ? pk[w + 1] : 0ull; x |= w1 >> (64 - 2 * o); }
	return (unsigned)(x >> (64 - 2 * FP_RUN));
}
__device__ __forceinline__ unsigned fp_sym_at(unsigned four, unsigned j) { return (four >> (2 * (FP_RUN - 1 - j))) & 3u; }

// F1: per tile t (FP_TILE elements from t T), one wave: out[4 t + 0] forward hash of the tile, + 1 backward hash, + 2 forward hash of its
// first r symbols, + 3 backward hash of its symbols r .. T-1 (weights B^(j - r))
__global__ void __launch_bounds__(FP_THREADS) k_fp_tiles(const u64 *__restrict__ pk, size_t nwords, unsigned ntiles_ext, FpConst C, Fp *__restrict__ out)
{
	const unsigned t = blockIdx.x * (FP_THREADS / 64u) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (t >= ntiles_ext) return;
	const unsigned syms = fp_syms(pk, nwords, (size_t)t * FP_TILE + (size_t)lane * FP_RUN);
	Fp hf{0, 0}, hb{0, 0};
#pragma unroll
	for (unsigned j = 0; j < FP_RUN; j++) hf = fp_horner(hf, C.B, fp_sym_at(syms, j));
#pragma unroll
	for (unsigned j = FP_RUN; j-- > 0;) hb = fp_horner(hb, C.B, fp_sym_at(syms, j));
	Fp totF, totB;
	// hash(first r symbols) = the exclusive prefix of the lane that holds element r, continued over its first r mod RUN symbols;
	// likewise the suffix hash from element r
	const unsigned r = C.r, rt = r / FP_RUN, rj = r % FP_RUN;
	const Fp exF = scan64<+1>(hf, C.c2, totF);
	const Fp exB = scan64<-1>(hb, C.c2, totB);
	if (lane == rt) {
		Fp pf = exF;                                                   // hash of the elements before this lane's chunk
		for (unsigned j = 0; j < rj; j++) pf = fp_horner(pf, C.B, fp_sym_at(syms, j));
		Fp pb = exB;                                                   // suffix hash from the element after this lane's chunk
		for (unsigned j = FP_RUN; j-- > rj;) pb = fp_horner(pb, C.B, fp_sym_at(syms, j));
		out[4 * (size_t)t + 2] = r ? pf : Fp{0, 0};
		out[4 * (size_t)t + 3] = pb;                                  // (r = 0: lane 0, all FP_RUN symbols: the whole tile)
	}
	if (lane == 0) { out[4 * (size_t)t + 0] = totF; out[4 * (size_t)t + 1] = totB; }
}

// F2: PT[t] = forward prefix hash at element t T (PT[0] = 0), ST[t] = suffix hash from element t T, t = 0 .. 256 nchunks.  Base of these
// scans = B^T.  Three small launches: chunks of 256 tiles scanned side by side (k_fp_chunks), the chunks' carries (k_fp_carries: two
// sequential loops over <= a few thousand chunks, one thread each), carries applied (k_fp_apply).  (One workgroup looping over all
// tiles was 0.88 ms at 37 k tiles and 21 ms at 880 k.)
__global__ void __launch_bounds__(FP_THREADS) k_fp_chunks(const Fp *__restrict__ tiles, unsigned n, FpConst C, Fp *__restrict__ PT, Fp *__restrict__ ST, Fp *__restrict__ tot /* 2 per chunk */)
{
	__shared__ Fp lds[8];
	const unsigned c = blockIdx.x, tid = threadIdx.x, i = c * FP_THREADS + tid;
	const Fp xf = i < n ? tiles[4 * (size_t)i + 0] : Fp{0, 0}, xb = i < n ? tiles[4 * (size_t)i + 1] : Fp{0, 0};
	Fp tf, tb;
	const Fp ef = scan256<+1>(xf, C.t2, C.t64, lds, tf);
	const Fp eb = scan256<-1>(xb, C.t2, C.t64, lds, tb);
	PT[i] = ef;                                                        // prefix inside the chunk, before tile i
	ST[i] = fp_add(xb, fp_mul(eb, C.t2[0]));                           // suffix inside the chunk, from the start of tile i
	if (tid == 0) { tot[2 * c] = tf; tot[2 * c + 1] = tb; }
}
// cP[c] = prefix hash at the start of chunk c, cS[c] = suffix hash from the start of chunk c (cS[nchunks] = 0).  Two waves, one per
// direction; a lane takes `seg` consecutive chunks: their hash first, a wave scan of the 64 segment hashes (weight (B^T)^(256 seg) per
// segment, passed as C.tseg), then the segment again from its carry.  (One thread per direction looping over all chunks was 1.5 ms at
// 900 Mbp: 6 867 dependent modular multiplies.)
__global__ void __launch_bounds__(128) k_fp_carries(const Fp *__restrict__ tot, unsigned nchunks, unsigned seg, FpConst C, Fp *__restrict__ cP, Fp *__restrict__ cS)
{
	const unsigned lane = threadIdx.x & 63u;
	const bool fwd = threadIdx.x < 64u;
	Fp c2[6];                                                          // C.tseg^(2^i)
	{ Fp x = C.tseg; for (int i = 0; i < 6; i++) { c2[i] = x; x = fp_mul(x, x); } }
	const unsigned lo = lane * seg, hi = lo + seg < nchunks ? lo + seg : nchunks;      // my chunks [lo, hi) (absent chunks count as zero)
	Fp h{0, 0};
	if (fwd) { for (unsigned c = lo; c < lo + seg; c++) h = fp_add(fp_mul(h, C.tT), c < hi ? tot[2 * c] : Fp{0, 0}); }
	else { for (unsigned c = lo + seg; c-- > lo;) h = fp_add(c < hi ? tot[2 * c + 1] : Fp{0, 0}, fp_mul(h, C.tT)); }
	Fp total;
	Fp carry = fwd ? scan64<+1>(h, c2, total) : scan64<-1>(h, c2, total);
	if (fwd) {
		if (lane == 0) cP[0] = Fp{0, 0};
		for (unsigned c = lo; c < hi; c++) { carry = fp_add(fp_mul(carry, C.tT), tot[2 * c]); cP[c + 1] = carry; }
	} else {
		if (lane == 0) cS[nchunks] = Fp{0, 0};
		for (unsigned c = hi; c-- > lo;) { carry = fp_add(tot[2 * c + 1], fp_mul(carry, C.tT)); cS[c] = carry; }
	}
}
__global__ void __launch_bounds__(FP_THREADS) k_fp_apply(unsigned nchunks, const Fp *__restrict__ cP, const Fp *__restrict__ cS, const Fp *__restrict__ pwT /* (B^T)^i, i = 0 .. 256 */,
                                                         Fp *__restrict__ PT, Fp *__restrict__ ST)
{
	const unsigned c = blockIdx.x, tid = threadIdx.x, i = c * FP_THREADS + tid;
	if (c == nchunks) { if (tid == 0) { PT[i] = cP[nchunks]; ST[i] = Fp{0, 0}; } return; }
	PT[i] = fp_add(PT[i], fp_mul(cP[c], pwT[tid]));
	ST[i] = fp_add(ST[i], fp_mul(cS[c + 1], pwT[FP_THREADS - tid]));
}

#define FP_INVALID KB_INVALID
// A record is 16 B, like at k <= 32: key = mix64(h1 | (h2 & 7) << 61) (a bijection of the 64 bits: bucket, table slot and 64 bits of identity),
// value = element (32) | prev mask (5) | next mask (5) | orientation flags (2) | 20 more bits of h2: an 84-bit fingerprint.  (The first
// version carried all 125 bits in 24-B records: + 50 % on every pass of the table build for a collision rate that is 1e-6 per run at
// 1.8 G windows either way far below anything observable -- and the result is exact either way: every positive is verified.)
#define FP_V_PREV 32
#define FP_V_NEXT 37
#define FP_V_FL 42
#define FP_V_H2 44
__device__ __forceinline__ bool fp_mask_bif(unsigned m10) { const unsigned p = m10 & 31u, n = (m10 >> 5) & 31u; return (p & 16u) || (n & 16u) || __popc(p & 15u) > 1 || __popc(n & 15u) > 1; }
// canonical-orientation key pair of a window and its value (element | masks << 32 | orientation flags << 48), see kmer_bucket_kernels.h
// F3: one WAVE per tile of window starts, FP_RUN consecutive windows per lane.  The prefix / suffix hashes of the tile's own range and of
// the range k further on come from two wave scans each (seeds from F2); only the two fingerprints per window are kept in registers.
__global__ void __launch_bounds__(FP_THREADS) k_fp_records(const u64 *__restrict__ pk, size_t nwords, const uint8_t *__restrict__ ch, size_t nelem,
                                                           const unsigned *__restrict__ sepidx, unsigned nchr, FpConst C, unsigned tile0, unsigned ntiles /* this GPU's slice of tiles: [tile0, ntiles) */,
                                                           const Fp *__restrict__ tiles, const Fp *__restrict__ PT, const Fp *__restrict__ ST, const Fp *__restrict__ pwrun /* (B^RUN)^i, i = 0 .. 64 */,
                                                           unsigned test_weak /* SBL_TEST_WEAK_FP: fingerprints reduced to this many bits (0 = off) */,
                                                           u64 *__restrict__ key1, u64 *__restrict__ rec)
{
	const unsigned t = tile0 + blockIdx.x * (FP_THREADS / 64u) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (t >= ntiles) return;
	const size_t a0 = (size_t)t * FP_TILE, a1 = a0 + C.k, out0 = (size_t)tile0 * FP_TILE;      // records are stored relative to the slice
	const size_t t1 = (size_t)t + C.q;
	const Fp pwl = pwrun[lane], pwr = pwrun[63u - lane];                // B^(RUN lane), B^(T - RUN (lane + 1))
	Fp F[FP_RUN], R[FP_RUN];                                           // first the prefix / suffix hashes at g, then F(w) / F(rc(w))
	{	// the tile's own range: P[g] and S[g]
		const unsigned syms = fp_syms(pk, nwords, a0 + (size_t)lane * FP_RUN);
		Fp hf{0, 0}, hb{0, 0};
#pragma unroll
		for (unsigned j = 0; j < FP_RUN; j++) hf = fp_horner(hf, C.B, fp_sym_at(syms, j));
#pragma unroll
		for (unsigned j = FP_RUN; j-- > 0;) hb = fp_horner(hb, C.B, fp_sym_at(syms, j));
		Fp tot;
		const Fp exF = scan64<+1>(hf, C.c2, tot), exB = scan64<-1>(hb, C.c2, tot);
		Fp hh = fp_add(fp_mul(PT[t], pwl), exF);                          // prefix hash at the lane's first element
#pragma unroll
		for (unsigned j = 0; j < FP_RUN; j++) { F[j] = hh; hh = fp_horner(hh, C.B, fp_sym_at(syms, j)); }
		hh = fp_add(fp_mul(ST[t + 1], pwr), exB);                         // suffix hash from the element behind the lane's last one
#pragma unroll
		for (unsigned j = FP_RUN; j-- > 0;) { hh = fp_horner(hh, C.B, fp_sym_at(syms, j)); R[j] = hh; }
	}
	{	// the range k further on: P[g + k] and S[g + k], combined at once
		const Fp seedP = fp_add(fp_mul(PT[t1], C.Br), tiles[4 * t1 + 2]);
		const Fp seedS = fp_add(tiles[4 * (t1 + 1) + 3], fp_mul(ST[t1 + 2], C.BTr));
		const unsigned syms = fp_syms(pk, nwords, a1 + (size_t)lane * FP_RUN);
		Fp hf{0, 0}, hb{0, 0};
#pragma unroll
		for (unsigned j = 0; j < FP_RUN; j++) hf = fp_horner(hf, C.B, fp_sym_at(syms, j));
#pragma unroll
		for (unsigned j = FP_RUN; j-- > 0;) hb = fp_horner(hb, C.B, fp_sym_at(syms, j));
		Fp tot;
		const Fp exF = scan64<+1>(hf, C.c2, tot), exB = scan64<-1>(hb, C.c2, tot);
		Fp hh = fp_add(fp_mul(seedP, pwl), exF);
#pragma unroll
		for (unsigned j = 0; j < FP_RUN; j++) { F[j] = fp_sub(hh, fp_mul(F[j], C.Bk)); hh = fp_horner(hh, C.B, fp_sym_at(syms, j)); }      // F(w) = P[g+k] - P[g] B^k
		hh = fp_add(fp_mul(seedS, pwr), exB);
#pragma unroll
		for (unsigned j = FP_RUN; j-- > 0;) { hh = fp_horner(hh, C.B, fp_sym_at(syms, j)); R[j] = fp_sub(C.G3, fp_sub(R[j], fp_mul(hh, C.Bk))); }   // F(rc(w)) = 3 G_k - (S[g] - B^k S[g+k])
	}
	// The lane holds FP_RUN CONSECUTIVE windows; stored like that, one store instruction would touch 64 lines (64 B apart per lane).  The
	// fingerprints are transposed through LDS (rows of FP_RUN + 1 entries: no bank conflicts) so that lane l finishes windows l, l + 64, ...:
	// every load of the neighbouring characters and every store of the wave is one contiguous run.
	// (the canonical orientation is decided first: two words per window travel, its two "<=" flags in the free top bits of the 61-bit one)
	__shared__ u64 s_t[FP_THREADS / 64u][2][64u * (FP_RUN + 1u)];
	u64 (*tr)[64u * (FP_RUN + 1u)] = s_t[threadIdx.x >> 6];
#pragma unroll
	for (unsigned j = 0; j < FP_RUN; j++) {
		Fp hf = F[j], hr = R[j];
		if (test_weak) { const u64 m = (1ull << test_weak) - 1; hf.a &= m; hf.b &= m; hr.a &= m; hr.b &= m; }
		const bool f_le = hf.a < hr.a || (hf.a == hr.a && hf.b <= hr.b), r_le = hr.a < hf.a || (hr.a == hf.a && hr.b <= hf.b);
		const Fp cn = f_le ? hf : hr;
		const unsigned at = lane * (FP_RUN + 1u) + j;
		tr[0][at] = cn.a | ((u64)f_le << 62) | ((u64)r_le << 63); tr[1][at] = cn.b;
	}
	// (one wave writes and reads its own rows: no barrier needed beyond the wave's own program order)
	__builtin_amdgcn_wave_barrier();
	unsigned c = 0;
	{	// chromosome of the wave's first window: sepidx[c] < g < sepidx[c+1] (a separator itself belongs to nobody)
		unsigned lo = 0, hi = nchr; const unsigned e = (unsigned)(a0 < nelem ? a0 : nelem - 1);
		while (hi - lo > 1) { const unsigned mid = (lo + hi) >> 1; if (sepidx[mid] < e) lo = mid; else hi = mid; }
		c = lo;
	}
#pragma unroll
	for (unsigned j = 0; j < FP_RUN; j++) {
		const unsigned w = j * 64u + lane;                              // window of the tile this lane finishes in step j
		const size_t g = a0 + w;
		while (c + 1 < nchr && g >= sepidx[c + 1]) c++;                 // (ascending within a lane: g grows by 64 per step)
		const bool valid = g < nelem && g > sepidx[c] && g + C.k <= sepidx[c + 1];
		u64 k1 = kmer_hash((u64)g), v = FP_INVALID;                       // invalid records: spread over the buckets, skipped by value
		if (valid) {
			const unsigned at = (w / FP_RUN) * (FP_RUN + 1u) + (w % FP_RUN);
			const u64 ca = tr[0][at];
			const bool f_le = (ca >> 62) & 1ull, r_le = (ca >> 63) & 1ull;
			const Fp cn{ca & FP_M61, tr[1][at]};
			const uint8_t pc = ch[g - 1], nc = ch[g + C.k];
			unsigned ps = (pc >> 1) & 3u; ps ^= ps >> 1; if (pc == '$') ps = 4u;
			unsigned ns = (nc >> 1) & 3u; ns ^= ns >> 1; if (nc == '$') ns = 4u;
			unsigned m = 0, fl = 0;                                       // masks in the canonical orientation: prev in bits 0-4, next in bits 5-9 ({A,C,G,T,#})
			if (f_le) { m |= (1u << ps) | (1u << (5 + ns)); fl |= 1u; }
			if (r_le) { m |= (1u << (ns == 4 ? 4 : 3 - ns)) | (1u << (5 + (ps == 4 ? 4 : 3 - ps))); fl |= 2u; }
			k1 = kmer_hash(cn.a | ((cn.b & 7ull) << 61));
			if (k1 == KB_EMPTY_KEY) k1 ^= 1ull;                           // (the table's empty marker; the verification keeps this exact)
			v = (u64)(unsigned)g | ((u64)m << FP_V_PREV) | ((u64)fl << FP_V_FL) | (((cn.b >> 3) & 0xFFFFFull) << FP_V_H2);
		}
		key1[g - out0] = k1; rec[g - out0] = v;
	}
}

// ------------------------------------------------------------------------------------------- F5: per-bucket tables on the 84-bit fingerprints
#define FPB_SLOTS 1024u
#define FPB_THREADS 256
#define FPB_MAX_DISTINCT (FPB_SLOTS * 23u / 32u)
static_assert(FPB_MAX_DISTINCT + FPB_THREADS < FPB_SLOTS, "k_fp_classify: the LDS table could fill up");
#define FPB_K2_EMPTY 0xFFFFFFFFu
enum { FPB_CTR_PAIRS = 0, FPB_CTR_MEM = 32, FPB_CTR_FLAG = 64, FPB_CTR_WORDS = 96 };
#define FPB_PAL 0x400u                              // tmask bit: some record of the slot had both orientation flags (fingerprint palindrome)
// pairs[p] = representative: element | orientation << 32 | palindrome << 33 (orientation 0: the + strand k-mer at the element is the canonical one)
// members[i] = element | (2 pair + orientation) << 32 | (both flags) << 63
// A workgroup takes FPB_GROUP consecutive buckets and STAGES what they emit in LDS -- one reservation of output ranges per flush instead
// of one per wave and step (as k_bucket_classify does; the first version reserved per wave: 1.5 M returning atomics on one address
// on raw strains at k = 100, where a tenth of the windows are members).  The records of the NEXT bucket are requested before this one
// is worked on (FPB_REGS per thread in registers: buckets of up to 768 records are read once).
#define FPB_GROUP 16u
#define FPB_REGS 3
#define FPB_STAGE_MEM 1536u
#define FPB_STAGE_PAIRS 384u
__global__ void __launch_bounds__(FPB_THREADS) k_fp_classify(const u64 *__restrict__ skey, const u64 *__restrict__ sval, const unsigned *__restrict__ boff, unsigned nbuckets,
                                                            unsigned *__restrict__ counters, u64 *__restrict__ pairs, unsigned maxpairs, u64 *__restrict__ members, unsigned maxmembers)
{
	__shared__ u64 tkey[FPB_SLOTS], trep[FPB_SLOTS];
	__shared__ unsigned tkey2[FPB_SLOTS], tmask[FPB_SLOTS], taux[FPB_SLOTS];
	__shared__ u64 st_mem[FPB_STAGE_MEM], st_pair[FPB_STAGE_PAIRS];
	__shared__ unsigned s_used, s_pairs, s_np, s_nm, s_bp, s_bm, s_dp;
	const unsigned b0 = blockIdx.x * FPB_GROUP, b1 = b0 + FPB_GROUP < nbuckets ? b0 + FPB_GROUP : nbuckets;
	if (b0 >= nbuckets) return;
	if (threadIdx.x == 0) { s_np = 0; s_nm = 0; }
	__syncthreads();
	// everything staged goes out: one reservation per output array, then coalesced stores with the pair base added
	auto flush = [&]() {
		__syncthreads();
		const unsigned np = s_np, nm = s_nm < FPB_STAGE_MEM ? s_nm : FPB_STAGE_MEM;
		if (threadIdx.x == 0) { s_bp = np ? atomicAdd(&counters[FPB_CTR_PAIRS], np) : 0u; s_bm = nm ? atomicAdd(&counters[FPB_CTR_MEM], nm) : 0u; }
		__syncthreads();
		const unsigned bp = s_bp, bm = s_bm;
		for (unsigned i = threadIdx.x; i < np; i += FPB_THREADS) if (bp + i < maxpairs) pairs[bp + i] = st_pair[i];
		for (unsigned i = threadIdx.x; i < nm; i += FPB_THREADS) if (bm + i < maxmembers) members[bm + i] = st_mem[i] + ((u64)(2u * bp) << 32);
		__syncthreads();
		if (threadIdx.x == 0) { s_np = 0; s_nm = 0; }
		__syncthreads();
	};
	// slot of a record: claimed on the key by compare-and-swap; its identity is whichever 20 further fingerprint bits arrive first (a second
	// compare-and-swap); a record with the same key and other bits moves on.  claim = false: look-up only (everything is inserted by then).
	auto slot_of = [&](u64 k1, unsigned k2, bool claim) -> unsigned {
		unsigned h = (unsigned)(k1 >> 44) & (FPB_SLOTS - 1);
		for (unsigned step = 0; step < FPB_SLOTS; step++, h = (h + 1) & (FPB_SLOTS - 1)) {
			u64 old = claim ? atomicCAS(&tkey[h], KB_EMPTY_KEY, k1) : tkey[h];
			if (claim && old == KB_EMPTY_KEY) { atomicAdd(&s_used, 1u); old = k1; }
			if (old != k1) { if (!claim && old == KB_EMPTY_KEY) return SBL_NONE; continue; }
			const unsigned o2 = claim ? atomicCAS(&tkey2[h], FPB_K2_EMPTY, k2) : tkey2[h];
			if (o2 == k2 || (claim && o2 == FPB_K2_EMPTY)) return h;
		}
		if (claim) atomicAdd(&s_used, FPB_SLOTS);
		return SBL_NONE;
	};
	// the first bucket's records: in flight while the table is cleared
	u64 rk[FPB_REGS], rv[FPB_REGS];
	unsigned lo = boff[b0], hi = boff[b0 + 1];
#pragma unroll
	for (int r = 0; r < FPB_REGS; r++) { const unsigned i = lo + threadIdx.x + r * FPB_THREADS; rv[r] = i < hi ? sval[i] : FP_INVALID; rk[r] = i < hi ? skey[i] : 0ull; }
	for (unsigned b = b0; b < b1; b++) {
		u64 nk_[FPB_REGS], nv_[FPB_REGS];
		unsigned nlo = 0, nhi = 0;
		if (b + 1 < b1) {
			nlo = hi; nhi = boff[b + 2];
#pragma unroll
			for (int r = 0; r < FPB_REGS; r++) { const unsigned i = nlo + threadIdx.x + r * FPB_THREADS; nv_[r] = i < nhi ? sval[i] : FP_INVALID; nk_[r] = i < nhi ? skey[i] : 0ull; }
		}
		if (lo < hi) {
			__syncthreads();
			if (s_nm > FPB_STAGE_MEM / 2 || s_np > FPB_STAGE_PAIRS / 2) flush();
			for (unsigned i = threadIdx.x; i < FPB_SLOTS; i += FPB_THREADS) { tkey[i] = KB_EMPTY_KEY; tkey2[i] = FPB_K2_EMPTY; tmask[i] = 0; taux[i] = SBL_NONE; }
			if (threadIdx.x == 0) { s_used = 0; s_pairs = 0; }
			__syncthreads();
			auto insert = [&](u64 k1, u64 v) {
				const unsigned h = slot_of(k1, (unsigned)(v >> FP_V_H2), true);
				if (h == SBL_NONE) return;
				const unsigned fl = (unsigned)(v >> FP_V_FL) & 3u;
				atomicOr(&tmask[h], ((unsigned)(v >> FP_V_PREV) & 0x3FFu) | (fl == 3u ? FPB_PAL : 0u));
			};
#pragma unroll
			for (int r = 0; r < FPB_REGS; r++) {
				if (*(volatile unsigned *)&s_used > FPB_MAX_DISTINCT) break;     // too many distinct k-mers for this table: the host re-buckets
				if (rv[r] != FP_INVALID) insert(rk[r], rv[r]);
			}
			for (unsigned i = lo + threadIdx.x + FPB_REGS * FPB_THREADS; i < hi; i += FPB_THREADS) {
				if (*(volatile unsigned *)&s_used > FPB_MAX_DISTINCT) break;
				const u64 v = sval[i];
				if (v != FP_INVALID) insert(skey[i], v);
			}
			__syncthreads();
			if (s_used > FPB_MAX_DISTINCT) { if (threadIdx.x == 0) atomicOr(&counters[FPB_CTR_FLAG], 1u); return; }   // (uniform; the host discards everything)
			for (unsigned sidx = threadIdx.x; sidx < FPB_SLOTS; sidx += FPB_THREADS) {
				if (tkey[sidx] == KB_EMPTY_KEY || !fp_mask_bif(tmask[sidx])) continue;
				taux[sidx] = atomicAdd(&s_pairs, 1u);
				trep[sidx] = ~0ull;                                     // (representatives are only taken for the bifurcation k-mers, in the member pass)
			}
			__syncthreads();
			const unsigned npairs = s_pairs;
			if (npairs) {
				// staged when it fits (flushing first if need be); a bucket with more pairs than the stage holds or more records than the member
				// stage could take (low-complexity input: one k-mer, thousands of occurrences) reserves its ranges itself and writes directly
				const bool direct = npairs > FPB_STAGE_PAIRS / 2 || hi - lo > FPB_STAGE_MEM / 2;
				if (direct) {
					flush();                                            // pair indices below are final, nothing staged refers to them
					if (threadIdx.x == 0) s_dp = atomicAdd(&counters[FPB_CTR_PAIRS], npairs);
					__syncthreads();
				}
				const unsigned pbase = direct ? s_dp : s_np;
				// member positions of the bifurcation k-mers: staged entries carry the STAGED pair index (the flush adds the base); the
				// representative of a group = its lowest element (deterministic), with its orientation
				auto emit = [&](u64 k1, u64 v, bool have) {
					unsigned pi = SBL_NONE;
					const unsigned fl = (unsigned)(v >> FP_V_FL) & 3u;
					if (have) {
						const unsigned h = slot_of(k1, (unsigned)(v >> FP_V_H2), false);
						if (h != SBL_NONE && taux[h] != SBL_NONE) { pi = pbase + taux[h]; atomicMin(&trep[h], ((v & 0xFFFFFFFFull) << 1) | ((fl & 1u) ? 0ull : 1ull)); }
					}
					const bool mem = pi != SBL_NONE;
					const u64 bal = __ballot(mem);
					if (!bal) return;
					const unsigned lane = threadIdx.x & 63u, first = (unsigned)__builtin_ctzll(bal);
					unsigned base = 0;
					if (lane == first) base = direct ? atomicAdd(&counters[FPB_CTR_MEM], (unsigned)__popcll(bal)) : atomicAdd(&s_nm, (unsigned)__popcll(bal));
					base = __shfl(base, first);
					const unsigned at = base + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
					const u64 rec = (v & 0xFFFFFFFFull) | ((u64)(2u * pi + ((fl & 1u) ? 0u : 1u)) << 32) | (fl == 3u ? 1ull << 63 : 0ull);
					if (mem) { if (direct) { if (at < maxmembers) members[at] = rec; } else if (at < FPB_STAGE_MEM) st_mem[at] = rec; }
				};
#pragma unroll
				for (int r = 0; r < FPB_REGS; r++) emit(rk[r], rv[r], rv[r] != FP_INVALID);
				for (unsigned i0 = lo + FPB_REGS * FPB_THREADS; i0 < hi; i0 += FPB_THREADS) {      // the rest of a bucket larger than the registers hold (always direct)
					const unsigned i = i0 + threadIdx.x;
					const u64 v = i < hi ? sval[i] : FP_INVALID;
					emit(i < hi ? skey[i] : 0ull, v, v != FP_INVALID);
				}
				__syncthreads();
				for (unsigned sidx = threadIdx.x; sidx < FPB_SLOTS; sidx += FPB_THREADS) {
					const unsigned a = taux[sidx];
					if (a == SBL_NONE) continue;
					const unsigned pi = pbase + a;
					const u64 rep = (trep[sidx] >> 1) | ((trep[sidx] & 1ull) << 32) | ((tmask[sidx] & FPB_PAL) ? 1ull << 33 : 0ull);
					if (direct) { if (pi < maxpairs) pairs[pi] = rep; } else st_pair[pi] = rep;
				}
				__syncthreads();
				if (threadIdx.x == 0 && !direct) s_np += npairs;
			}
		}
		lo = nlo; hi = nhi;
#pragma unroll
		for (int r = 0; r < FPB_REGS; r++) { rk[r] = nk_[r]; rv[r] = nv_[r]; }
	}
	flush();
}

// ------------------------------------------------------------------------------------------- sequence access for verification and ranking
// up to 32 symbols from element e on the + strand, left-aligned in a 64-bit word (first symbol in the top bits), zero-filled
__device__ __forceinline__ u64 fp_chunk_fwd(const u64 *__restrict__ pk, size_t e, unsigned len)
{
	const size_t w = e >> 5; const unsigned o = (unsigned)(e & 31u);
	u64 x = pk[w] << (2 * o);
	if (o && o + len > 32u) x |= pk[w + 1] >> (64 - 2 * o);
	return len >= 32u ? x : x & ~(~0ull >> (2 * len));
}
// symbols off .. off + len - 1 of the k-mer string of (window g, orientation o): o = 0 the + strand k-mer, o = 1 its reverse complement
__device__ __forceinline__ u64 fp_chunk(const u64 *__restrict__ pk, unsigned g, unsigned o, unsigned k, unsigned off, unsigned len)
{
	if (!o) return fp_chunk_fwd(pk, (size_t)g + off, len);
	// rc string symbol i = complement of the window's symbol k-1-i: the chunk is the window's symbols k-off-len .. k-off-1, reversed and complemented
	const u64 f = fp_chunk_fwd(pk, (size_t)g + (k - off - len), len);    // left-aligned
	return rc_code(f >> (64 - 2 * len), len) << (64 - 2 * len);
}

// F6: work item = (member, chunk of 32 symbols): the member's canonical string against the representative's
__global__ void __launch_bounds__(256) k_fp_verify(const u64 *__restrict__ pk, const u64 *__restrict__ members, unsigned nmem, const u64 *__restrict__ pairs, unsigned k, unsigned nchunks,
                                                   unsigned *__restrict__ bad)
{
	const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (w >= (size_t)nmem * nchunks) return;
	const unsigned mi = (unsigned)(w / nchunks), cj = (unsigned)(w % nchunks);
	const u64 m = members[mi];
	const unsigned g = (unsigned)m, po = (unsigned)(m >> 32) & 0x7FFFFFFFu, o = po & 1u;
	const u64 rep = pairs[po >> 1];
	const unsigned gr = (unsigned)rep, orr = (unsigned)(rep >> 32) & 1u;
	const unsigned off = cj * 32u, len = k - off < 32u ? k - off : 32u;
	bool ok = fp_chunk(pk, g, o, k, off, len) == fp_chunk(pk, gr, orr, k, off, len);
	if (m >> 63) ok = ok && fp_chunk(pk, g, 0, k, off, len) == fp_chunk(pk, g, 1, k, off, len);      // claims to be its own reverse complement
	if (!ok) atomicAdd(bad, 1u);
}

// F7: the strings to rank: per pair its canonical string and (unless it is its own reverse complement) the reverse complement.
// ref[i] = element | orientation << 32; payload[i] = 2 pair + (0: canonical, 1: its reverse complement)
__global__ void __launch_bounds__(256) k_fp_rank_init(const u64 *__restrict__ pairs, unsigned npairs, u64 *__restrict__ ref, unsigned *__restrict__ payload, unsigned *__restrict__ nkeys)
{
	const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
	const bool in = p < npairs;
	const u64 rep = in ? pairs[p] : 0ull;
	const unsigned n = in ? (((rep >> 33) & 1ull) ? 1u : 2u) : 0u;
	// ordered compaction is not needed: any order of the keys will do (ranks come from the sort)
	unsigned at = 0;
	{
		const unsigned lane = threadIdx.x & 63u;
		unsigned incl = n;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(incl, d); if (lane >= (unsigned)d) incl += v; }
		unsigned base = 0;
		if (lane == 63u) base = atomicAdd(nkeys, incl);
		base = __shfl(base, 63);
		at = base + incl - n;
	}
	if (!in) return;
	const unsigned g = (unsigned)rep, o = (unsigned)(rep >> 32) & 1u;
	ref[at] = (u64)g | ((u64)o << 32); payload[at] = 2u * p;
	if (n == 2u) { ref[at + 1] = (u64)g | ((u64)(o ^ 1u) << 32); payload[at + 1] = 2u * p + 1u; }
}
// sort key of a round: (rank so far << 2 csym) | the next csym symbols of the string
__global__ void __launch_bounds__(256) k_fp_rank_keys(const u64 *__restrict__ pk, const u64 *__restrict__ ref, const unsigned *__restrict__ rank, unsigned n, unsigned k, unsigned off, unsigned csym,
                                                      u64 *__restrict__ keys, unsigned *__restrict__ idx)
{
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u64 r = ref[i];
	const unsigned len = k - off < csym ? k - off : csym;
	const u64 chunk = fp_chunk(pk, (unsigned)r, (unsigned)(r >> 32) & 1u, k, off, len) >> (64 - 2 * csym);
	keys[i] = ((u64)rank[i] << (2 * csym)) | chunk;
	idx[i] = i;
}
// head[j] = j where a new group of equal keys starts in the sorted order, else 0 (input of a running maximum)
__global__ void __launch_bounds__(256) k_fp_rank_heads(const u64 *__restrict__ skeys, unsigned n, unsigned *__restrict__ head, unsigned *__restrict__ nheads)
{
	const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	const bool h = j < n && (j == 0 || skeys[j] != skeys[j - 1]);
	if (j < n) head[j] = h ? j : 0u;
	const u64 bal = __ballot(h);
	if (bal && (threadIdx.x & 63u) == 0) atomicAdd(nheads, (unsigned)__popcll(bal));
}
__global__ void __launch_bounds__(256) k_fp_rank_apply(const unsigned *__restrict__ sidx, const unsigned *__restrict__ gstart, unsigned n, unsigned *__restrict__ rank)
{
	const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) rank[sidx[j]] = gstart[j];
}
// A small set (the cascade states leave a few hundred bifurcation k-mers): rank by counting in ONE launch -- a wave per string, a lane per
// other string, compared chunk by chunk (first difference decides) -- instead of k / 27 rounds of {keys, sort, heads, scan, apply, host read}
#define FP_RANK_SMALL 1024u
__global__ void __launch_bounds__(64) k_fp_rank_small(const u64 *__restrict__ pk, const u64 *__restrict__ ref, unsigned n, unsigned k, unsigned *__restrict__ rank)
{
	const unsigned i = blockIdx.x, lane = threadIdx.x;
	if (i >= n) return;
	const u64 ri = ref[i];
	const unsigned gi = (unsigned)ri, oi = (unsigned)(ri >> 32) & 1u;
	unsigned below = 0;
	for (unsigned j = lane; j < n; j += 64) {
		if (j == i) continue;
		const u64 rj = ref[j];
		const unsigned gj = (unsigned)rj, oj = (unsigned)(rj >> 32) & 1u;
		for (unsigned off = 0; off < k; off += 32) {
			const unsigned len = k - off < 32u ? k - off : 32u;
			const u64 a = fp_chunk(pk, gi, oi, k, off, len), b = fp_chunk(pk, gj, oj, k, off, len);
			if (a != b) { below += b < a; break; }
		}
	}
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) below += __shfl_down(below, d);
	if (lane == 0) rank[i] = below;
}
// final: every string has its own rank = its id
__global__ void __launch_bounds__(256) k_fp_rank_ids(const unsigned *__restrict__ rank, const unsigned *__restrict__ payload, const u64 *__restrict__ pairs, unsigned n, unsigned *__restrict__ pairids)
{
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned p = payload[i];
	pairids[p] = rank[i];
	if (!(p & 1u) && ((pairs[p >> 1] >> 33) & 1ull)) pairids[p + 1] = rank[i];      // its own reverse complement: one vertex for both orientations
}
// F8
__global__ void __launch_bounds__(256) k_fp_marks(const u64 *__restrict__ members, unsigned n, unsigned k, const unsigned *__restrict__ pairids, unsigned *__restrict__ bif0, unsigned *__restrict__ bif1)
{
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u64 m = members[i];
	const unsigned g = (unsigned)m, p = (unsigned)(m >> 32) & 0x7FFFFFFFu;
	bif0[g] = pairids[p];
	bif1[g + k - 1] = pairids[p ^ 1u];
}
// the two marks of a member position as (element, id) for the compact lists (sbl_compact_marks' output), strand by strand
__global__ void __launch_bounds__(256) k_fp_mark_pairs(const u64 *__restrict__ members, unsigned n, unsigned k, const unsigned *__restrict__ pairids, unsigned strand,
                                                       unsigned *__restrict__ elem, unsigned *__restrict__ id)
{
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const u64 m = members[i];
	const unsigned g = (unsigned)m, p = (unsigned)(m >> 32) & 0x7FFFFFFFu;
	elem[i] = strand ? g + k - 1 : g;
	id[i] = pairids[strand ? p ^ 1u : p];
}
// bucket b = records whose first key's LOW `bits` bits equal b (see k_bucket_bounds)
__global__ void __launch_bounds__(256) k_fp_bucket_bounds(const u64 *__restrict__ skeys, size_t n, unsigned bits, unsigned *__restrict__ boff)
{
	const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nb = (size_t)1 << bits;
	if (b > nb) return;
	if (b == nb) { boff[b] = (unsigned)n; return; }
	size_t lo = 0, hi = n;
	while (lo < hi) { const size_t mid = (lo + hi) >> 1; if ((skeys[mid] & (nb - 1)) < b) lo = mid + 1; else hi = mid; }
	boff[b] = (unsigned)lo;
}
// ------------------------------------------------------------------------------------------- host
struct LongKFpScratch {
	DevBuf tiles, PT, ST, pwrun, pwT, ctot, cP, cS, key1, rec, skey1, srec, rkey, rval, gpairs, gel, gid, tmp, boff, ctr, pairs, members, ref, payload, rank, keys, skeys, idx, sidx, head, gstart, pairids;
};
struct LongKFpHolder { LongKFpScratch s; };
static LongKFpScratch &fp_of(sbl_ctx *c)
{
	if (!c->lkfp) c->lkfp = new LongKFpHolder;
	return c->lkfp->s;
}
void sbl_longk_fp_free(sbl_ctx *c)
{
	if (!c->lkfp) return;
	LongKFpScratch &L = c->lkfp->s;
	for (DevBuf *b : { &L.tiles, &L.PT, &L.ST, &L.pwrun, &L.pwT, &L.ctot, &L.cP, &L.cS, &L.key1, &L.rec, &L.skey1, &L.srec, &L.rkey, &L.rval, &L.gpairs, &L.gel, &L.gid, &L.tmp, &L.boff, &L.ctr, &L.pairs, &L.members, &L.ref, &L.payload, &L.rank,
	                   &L.keys, &L.skeys, &L.idx, &L.sidx, &L.head, &L.gstart, &L.pairids })
		b->release();
	delete c->lkfp;
	c->lkfp = nullptr;
}

static Fp fp_pow(Fp b, unsigned long long e) { Fp r{1, 1}; while (e) { if (e & 1) r = fp_mul(r, b); b = fp_mul(b, b); e >>= 1; } return r; }
static unsigned fp_bits(unsigned long long v) { unsigned b = 1; while (b < 64 && (v >> b)) b++; return b; }
struct MaxU32 { __host__ __device__ unsigned operator()(unsigned a, unsigned b) const { return a > b ? a : b; } };

// returns false when the verification found two different k-mers under one fingerprint (the caller then runs the exact rank doubling)
bool sbl_run_enumeration_longk_fp(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	hipStream_t s = c->stream;
	const size_t E = c->nelem, nwords = (E + 31) / 32;
	// (the reference's limit, kept as the drop-in's error behaviour: its suffix array of the superGenome "#c0#..#rc(c0)#..#" is int32,
	// vertexenumeration.cpp:288-300 -- this path itself would go to 2^32 elements)
	SBL_CHECK(2 * E - 1 + k < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "input too large for 32-bit suffix ranks");
	c->cur_k = k;
	c->stats.exchange_bytes = 0; c->stats.exchange_ms = 0;
	sbl_pack(c);
	LongKFpScratch &L = fp_of(c);
	// ---- constants
	FpConst C{};
	C.B = Fp{0x1D3F5A7C9B2E4F61ull % FP_M61, 0x9E3779B97F4A7C15ull | 1ull};      // fixed bases: results do not depend on them (ids are ranks, groups are verified)
	C.k = k; C.q = k / FP_TILE; C.r = k % FP_TILE;
	const Fp Crun = fp_pow(C.B, FP_RUN);
	{ Fp x = Crun; for (int i = 0; i < 6; i++) { C.c2[i] = x; x = fp_mul(x, x); } C.c64 = x; }
	C.Bk = fp_pow(C.B, k);
	{	// G3 = 3 sum_{j<k} B^j, by doubling: S(2n) = S(n) (1 + B^n), S(n+1) = S(n) B + 1
		Fp sum{0, 0}; Fp one{1, 1};
		for (int bit = 31; bit >= 0; bit--) {
			const unsigned long long n = (unsigned long long)k >> (bit + 1);      // length so far
			if (n) sum = fp_mul(sum, fp_add(one, fp_pow(C.B, n)));                 // (a few dozen modular powers on the host: microseconds)
			if ((k >> bit) & 1u) sum = fp_add(fp_mul(sum, C.B), one);
		}
		C.G3 = fp_add(fp_add(sum, sum), sum);
	}
	C.Br = fp_pow(C.B, C.r); C.BTr = fp_pow(C.B, FP_TILE - C.r);
	const Fp BT = fp_pow(C.B, FP_TILE);
	{ Fp x = BT; for (int i = 0; i < 6; i++) { C.t2[i] = x; x = fp_mul(x, x); } C.t64 = x; C.tT = fp_pow(BT, FP_THREADS); }
	const unsigned ntiles = (unsigned)((E + FP_TILE - 1) / FP_TILE), nx = ntiles + C.q + 3;
	const size_t n = (size_t)ntiles * FP_TILE;                                  // records (one per element slot of the tiles)
	// Several GPUs on one job (a communicator with more than one rank): the table is sharded by HASH PREFIX exactly as at k <= 32
	// (shard.hip) -- the records are the same 16 B.  Every rank hashes the tiles of the whole sequence (0.25 B per element), makes
	// the records of ITS slice of tiles, partitions them, ONE all-to-all takes every bucket to its owner (bucket b belongs to rank
	// (b R) >> bits), owners classify and VERIFY their buckets, the representatives of the bifurcation k-mers (8 B each) are
	// all-gathered and ranked identically everywhere, the owners' member marks are all-gathered and scattered.  Balanced by
	// construction -- the sharded rank doubling cut the VALUE range of the first symbols (rank 0 nearly idle).
	SblComm *cm = c->comm;
	const bool sh = cm && cm->n > 1;
	const uint32_t R = sh ? cm->n : 1u, r = sh ? cm->rank : 0u;
	double xms = 0;
	auto timed = [&](auto f) { const auto t0 = std::chrono::steady_clock::now(); f(); xms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
	// every rank contributes sbytes bytes; recv holds all of them in rank order; returns the total and where mine starts
	auto allgatherv = [&](const char *send, size_t sbytes, DevBuf &recv, size_t *mine_at) -> size_t {
		std::vector<unsigned long long> mine(1, sbytes), all(R);
		timed([&] { cm->allgather_host(c, mine.data(), 8, all.data()); });
		std::vector<size_t> rb(R), ro(R), sb(R, sbytes), so(R, 0);
		size_t tot = 0;
		for (uint32_t p = 0; p < R; p++) { rb[p] = all[p]; ro[p] = tot; tot += all[p]; }
		recv.ensure(tot + 16);
		timed([&] { cm->alltoallv(c, send, sb.data(), so.data(), recv.as<char>(), rb.data(), ro.data()); });
		c->stats.exchange_bytes += sbytes * (R - 1);
		if (mine_at) *mine_at = ro[r];
		return tot;
	};
	const unsigned t0 = (unsigned)((uint64_t)ntiles * r / R), t1 = (unsigned)((uint64_t)ntiles * (r + 1) / R);
	const size_t nmine = (size_t)(t1 - t0) * FP_TILE;
	L.pwrun.ensure((FP_THREADS + 1) * sizeof(Fp)); L.pwT.ensure((FP_THREADS + 1) * sizeof(Fp));
	{
		std::vector<Fp> pw(2 * (FP_THREADS + 1));
		pw[0] = pw[FP_THREADS + 1] = Fp{1, 1};
		for (unsigned i = 1; i <= FP_THREADS; i++) { pw[i] = fp_mul(pw[i - 1], Crun); pw[FP_THREADS + 1 + i] = fp_mul(pw[FP_THREADS + i], BT); }
		HIP_TRY(hipMemcpyAsync(L.pwrun.p, pw.data(), (FP_THREADS + 1) * sizeof(Fp), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(L.pwT.p, pw.data() + FP_THREADS + 1, (FP_THREADS + 1) * sizeof(Fp), hipMemcpyHostToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));                                       // (pw is a local)
	}
	const unsigned ntchunks = (nx + FP_THREADS - 1) / FP_THREADS;
	L.tiles.ensure((size_t)nx * 4 * sizeof(Fp)); L.PT.ensure(((size_t)ntchunks * FP_THREADS + 1) * sizeof(Fp)); L.ST.ensure(((size_t)ntchunks * FP_THREADS + 1) * sizeof(Fp));
	L.ctot.ensure((size_t)ntchunks * 2 * sizeof(Fp)); L.cP.ensure(((size_t)ntchunks + 1) * sizeof(Fp)); L.cS.ensure(((size_t)ntchunks + 1) * sizeof(Fp));
	{	// pre-flight (as lk_preflight of the rank doubling): 32 B per element of records + ~8 B of outputs, against what the device has free,
		// BEFORE the first large allocation -- a clear SBL_ERR_OOM instead of a failure half way through (SBL_TEST_FREE_MEM_MB: test switch)
		auto miss = [](const DevBuf &b, size_t want) { return want > b.cap ? want + want / 16 + 256 : (size_t)0; };
		const size_t nn = nmine + 16;
		const size_t need = miss(L.key1, nn * 8) + miss(L.rec, nn * 8) + miss(L.skey1, nn * 8) + miss(L.srec, nn * 8) + miss(L.members, nn * 8 + 16) + miss(L.pairs, (nn / 8 + 4096) * 8 + 16) + miss(L.tmp, nn / 64 + (1u << 20));
		size_t fr = 0, tot = 0;
		if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
			if (const char *e = getenv("SBL_TEST_FREE_MEM_MB")) fr = (size_t)atoll(e) << 20;
			if (need > fr) {
				char b[200];
				snprintf(b, sizeof b, "long-k enumeration of %zu windows needs %zu MB of workspace, %zu MB free on the device", nmine, need >> 20, fr >> 20);
				throw SblError{SBL_ERR_OOM, b};
			}
		} else (void)hipGetLastError();
	}
	L.key1.ensure(nmine * 8 + 16); L.rec.ensure(nmine * 8 + 16); L.skey1.ensure(nmine * 8 + 16); L.srec.ensure(nmine * 8 + 16);
	L.ctr.ensure(256 * 4);
	HIP_TRY(hipEventRecord(c->ev[0], s));
	k_fp_tiles<<<nblocks(nx, FP_THREADS / 64), FP_THREADS, 0, s>>>(c->d_pk.as<u64>(), nwords, nx, C, L.tiles.as<Fp>());
	k_fp_chunks<<<ntchunks, FP_THREADS, 0, s>>>(L.tiles.as<Fp>(), nx, C, L.PT.as<Fp>(), L.ST.as<Fp>(), L.ctot.as<Fp>());
	const unsigned cseg = (ntchunks + 63u) / 64u;
	C.tseg = fp_pow(C.tT, cseg);
	k_fp_carries<<<1, 128, 0, s>>>(L.ctot.as<Fp>(), ntchunks, cseg, C, L.cP.as<Fp>(), L.cS.as<Fp>());
	k_fp_apply<<<ntchunks + 1, FP_THREADS, 0, s>>>(ntchunks, L.cP.as<Fp>(), L.cS.as<Fp>(), L.pwT.as<Fp>(), L.PT.as<Fp>(), L.ST.as<Fp>());
	unsigned weak = 0;
	if (const char *e = getenv("SBL_TEST_WEAK_FP")) weak = (unsigned)std::min(60, std::max(0, atoi(e)));      // test hook: collisions on purpose (the verification must notice)
	if (t1 > t0) k_fp_records<<<nblocks(t1 - t0, FP_THREADS / 64), FP_THREADS, 0, s>>>(c->d_pk.as<u64>(), nwords, c->d_ch.as<uint8_t>(), E, c->d_sepidx.as<unsigned>(), c->nchr, C, t0, t1,
	                                          L.tiles.as<Fp>(), L.PT.as<Fp>(), L.ST.as<Fp>(), L.pwrun.as<Fp>(), weak, L.key1.as<u64>(), L.rec.as<u64>());
	HIP_TRY(hipGetLastError());

	unsigned bits = 4;
	while (bits < 30 && (n >> bits) > FPB_SLOTS * 9 / 16) bits++;                // (the same on every rank: buckets are sized by the WHOLE input)
	if (const char *e = getenv("SBL_TEST_BUCKET_BITS")) bits = std::min(bits, (unsigned)std::max(1, atoi(e)));
	size_t maxpairs = n / R / 8 + 4096;
	if (const char *e = getenv("SBL_TEST_MAXPAIRS")) maxpairs = (size_t)std::max(1, atoi(e));
	unsigned cnt[3] = {0, 0, 0};
	const u64 *ckey = nullptr, *cval = nullptr;                                   // what the classification reads: my partitioned records, or what I own of everybody's
	size_t nown = nmine;
	auto partition = [&](u64 *kin, u64 *kout, u64 *vin, u64 *vout, size_t m) {
		if (!m) return;
		size_t tmp = 0;
		HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, m, 0, bits, s));
		L.tmp.ensure(tmp);
		HIP_TRY(rocprim::radix_sort_pairs(L.tmp.p, tmp, kin, kout, vin, vout, m, 0, bits, s));
	};
	for (int attempt = 0;; attempt++) {
		SBL_CHECK(attempt < 8, SBL_ERR_INTERNAL, "k-mer bucket classification did not converge");
		const size_t nb = (size_t)1 << bits;
		L.boff.ensure((nb + 1) * 4 + 64);
		partition(L.key1.as<u64>(), L.skey1.as<u64>(), L.rec.as<u64>(), L.srec.as<u64>(), nmine);
		k_fp_bucket_bounds<<<nblocks(nb + 1, 256), 256, 0, s>>>(L.skey1.as<u64>(), nmine, bits, L.boff.as<unsigned>());
		ckey = L.skey1.as<u64>(); cval = L.srec.as<u64>(); nown = nmine;
		if (sh) {
			// ---- the exchange: what goes to one owner is ONE contiguous range of the partitioned arrays
			std::vector<unsigned> fb(R + 1), send_at(R + 1, 0);
			uint64_t trange[2];
			SBL_CHECK(sbl_shard_layout(R, r, bits, ntiles, fb.data(), trange) == SBL_OK, SBL_ERR_INTERNAL, "shard layout");
			for (uint32_t p = 0; p <= R; p++) HIP_TRY(hipMemcpyAsync(&send_at[p], L.boff.as<unsigned>() + fb[p], 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			std::vector<unsigned long long> scount(R), allcount((size_t)R * R);
			for (uint32_t p = 0; p < R; p++) scount[p] = send_at[p + 1] - send_at[p];
			timed([&] { cm->allgather_host(c, scount.data(), R * 8, allcount.data()); });
			std::vector<size_t> sb(R), so(R), rb(R), ro(R);
			uint64_t got = 0;
			static_assert(sizeof(size_t) == sizeof(uint64_t), "64-bit host");
			SBL_CHECK(sbl_shard_exchange_plan(R, r, (const uint64_t *)allcount.data(), send_at.data(), 8, (uint64_t *)sb.data(), (uint64_t *)so.data(),
			                                  (uint64_t *)rb.data(), (uint64_t *)ro.data(), &got) == SBL_OK, SBL_ERR_INTERNAL, "exchange plan: the gathered counts contradict my own");
			nown = (size_t)got;
			SBL_CHECK(nown < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many records for one owner");
			for (uint32_t p = 0; p < R; p++) if (p != r) c->stats.exchange_bytes += 2 * sb[p];
			L.rkey.ensure(2 * (nown * 8 + 16)); L.rval.ensure(2 * (nown * 8 + 16));      // received / partitioned again
			u64 *rk = L.rkey.as<u64>(), *rv = L.rval.as<u64>(), *ok = rk + nown + 2, *ov = rv + nown + 2;
			timed([&] { cm->alltoallv(c, (const char *)L.skey1.as<u64>(), sb.data(), so.data(), (char *)rk, rb.data(), ro.data()); });
			timed([&] { cm->alltoallv(c, (const char *)L.srec.as<u64>(), sb.data(), so.data(), (char *)rv, rb.data(), ro.data()); });
			partition(rk, ok, rv, ov, nown);                                      // R runs, each sorted by bucket
			k_fp_bucket_bounds<<<nblocks(nb + 1, 256), 256, 0, s>>>(ok, nown, bits, L.boff.as<unsigned>());
			ckey = ok; cval = ov;
		}
		const size_t maxmembers = nown;
		for (;;) {
			L.pairs.ensure(maxpairs * 8 + 16); L.members.ensure(maxmembers * 8 + 16);
			HIP_TRY(hipMemsetAsync(L.ctr.p, 0, 256 * 4, s));
			k_fp_classify<<<nblocks(nb, FPB_GROUP), FPB_THREADS, 0, s>>>(ckey, cval, L.boff.as<unsigned>(), (unsigned)nb, L.ctr.as<unsigned>(),
			                                                            L.pairs.as<u64>(), (unsigned)maxpairs, L.members.as<u64>(), (unsigned)maxmembers);
			HIP_TRY(hipGetLastError());
			unsigned all[FPB_CTR_WORDS];
			HIP_TRY(hipMemcpyAsync(all, L.ctr.p, sizeof all, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			cnt[0] = all[FPB_CTR_PAIRS]; cnt[1] = all[FPB_CTR_MEM]; cnt[2] = all[FPB_CTR_FLAG];
			if (!(cnt[2] & 1u) && cnt[0] > maxpairs) { maxpairs = (size_t)cnt[0] + 1024; continue; }
			break;
		}
		bool rebucket = (cnt[2] & 1u) != 0;
		if (sh) {                                                                // a bucket that overflowed anywhere makes everybody re-bucket with a longer prefix
			std::vector<unsigned long long> flag(1, rebucket ? 1 : 0), flags(R);
			timed([&] { cm->allgather_host(c, flag.data(), 8, flags.data()); });
			rebucket = std::find(flags.begin(), flags.end(), 1ull) != flags.end();
		}
		if (!rebucket) break;
		SBL_CHECK(bits < 28, SBL_ERR_TOO_LARGE, "k-mer buckets keep overflowing at 2^28 buckets (adversarial key distribution)");
		bits = std::min(bits + 2, 28u);
	}
	HIP_TRY(hipEventRecord(c->ev[1], s));
	const unsigned mypairs = cnt[0], nmem = cnt[1];
	SBL_CHECK(nmem <= nown, SBL_ERR_INTERNAL, "more member positions than windows");

	// ---- F6: verification of every member of every bifurcation group (owners verify their buckets: the packed sequence is everywhere)
	const unsigned nchunks = (k + 31) / 32;
	unsigned *d_bad = L.ctr.as<unsigned>() + 128, *d_nkeys = L.ctr.as<unsigned>() + 160, *d_nheads = L.ctr.as<unsigned>() + 192;
	{
		unsigned bad = 0;
		const size_t work = (size_t)nmem * nchunks;
		if (nmem) {
			SBL_CHECK(work / 256 < 0x7FFFFFFFull, SBL_ERR_TOO_LARGE, "verification grid too large");
			k_fp_verify<<<nblocks(work, 256), 256, 0, s>>>(c->d_pk.as<u64>(), L.members.as<u64>(), nmem, L.pairs.as<u64>(), k, nchunks, d_bad);
			HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
		}
		unsigned long long anybad = bad;
		if (sh) {                                                                // one rank's failed comparison sends EVERY rank to the rank doubling
			std::vector<unsigned long long> mine(1, bad), all(R);
			timed([&] { cm->allgather_host(c, mine.data(), 8, all.data()); });
			anybad = 0; for (auto v : all) anybad += v;
		}
		if (anybad) {
			if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] long k: %llu of the chunk comparisons of the fingerprint groups failed -- falling back to exact rank doubling\n", anybad);
			return false;
		}
	}
	c->stats.fp_verified = (uint64_t)nmem;

	// ---- F7: ids = lexicographic rank of the bifurcation k-mers; with several GPUs the representatives of all owners, in rank order
	const u64 *gp = L.pairs.as<u64>();
	unsigned npairs = mypairs;
	size_t pair0 = 0;                                                            // index of my first pair among everybody's
	if (sh) {
		size_t at = 0;
		const size_t tot = allgatherv((const char *)L.pairs.as<u64>(), (size_t)mypairs * 8, L.gpairs, &at);
		SBL_CHECK(tot / 8 < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "too many bifurcations");
		gp = L.gpairs.as<u64>(); npairs = (unsigned)(tot / 8); pair0 = at / 8;
	}
	unsigned nkeys = 0;
	L.pairids.ensure((size_t)npairs * 8 + 16);
	if (npairs) {
		const size_t cap = 2 * (size_t)npairs;
		L.ref.ensure(cap * 8); L.payload.ensure(cap * 4); L.rank.ensure(cap * 4); L.keys.ensure(cap * 8); L.skeys.ensure(cap * 8); L.idx.ensure(cap * 4); L.sidx.ensure(cap * 4);
		L.head.ensure(cap * 4); L.gstart.ensure(cap * 4);
		HIP_TRY(hipMemsetAsync(d_nkeys, 0, 4, s));
		k_fp_rank_init<<<nblocks(npairs, 256), 256, 0, s>>>(gp, npairs, L.ref.as<u64>(), L.payload.as<unsigned>(), d_nkeys);
		HIP_TRY(hipMemcpyAsync(&nkeys, d_nkeys, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemsetAsync(L.rank.p, 0, cap * 4, s));
		HIP_TRY(hipStreamSynchronize(s));
		const unsigned rbits = fp_bits(nkeys ? nkeys - 1 : 0), csym = std::min(27u, (64u - rbits) / 2u);
		const bool small = nkeys <= FP_RANK_SMALL && getenv("SBL_FP_RANK_ROUNDS") == nullptr;      // (SBL_FP_RANK_ROUNDS=1: test switch, the refinement rounds for every size)
		if (small) k_fp_rank_small<<<nkeys, 64, 0, s>>>(c->d_pk.as<u64>(), L.ref.as<u64>(), nkeys, k, L.rank.as<unsigned>());
		size_t tsort = 0, tscan = 0;
		HIP_TRY(rocprim::radix_sort_pairs(nullptr, tsort, L.keys.as<u64>(), L.skeys.as<u64>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), nkeys, 0, 64, s));
		HIP_TRY(rocprim::inclusive_scan(nullptr, tscan, L.head.as<unsigned>(), L.gstart.as<unsigned>(), nkeys, MaxU32(), s));
		L.tmp.ensure(std::max(tsort, tscan));
		for (unsigned off = 0; off < k && !small; off += csym) {
			HIP_TRY(hipMemsetAsync(d_nheads, 0, 4, s));
			k_fp_rank_keys<<<nblocks(nkeys, 256), 256, 0, s>>>(c->d_pk.as<u64>(), L.ref.as<u64>(), L.rank.as<unsigned>(), nkeys, k, off, csym, L.keys.as<u64>(), L.idx.as<unsigned>());
			HIP_TRY(rocprim::radix_sort_pairs(L.tmp.p, tsort, L.keys.as<u64>(), L.skeys.as<u64>(), L.idx.as<unsigned>(), L.sidx.as<unsigned>(), nkeys, 0, std::min(64u, rbits + 2 * csym), s));
			k_fp_rank_heads<<<nblocks(nkeys, 256), 256, 0, s>>>(L.skeys.as<u64>(), nkeys, L.head.as<unsigned>(), d_nheads);
			HIP_TRY(rocprim::inclusive_scan(L.tmp.p, tscan, L.head.as<unsigned>(), L.gstart.as<unsigned>(), nkeys, MaxU32(), s));
			k_fp_rank_apply<<<nblocks(nkeys, 256), 256, 0, s>>>(L.sidx.as<unsigned>(), L.gstart.as<unsigned>(), nkeys, L.rank.as<unsigned>());
			unsigned heads = 0;
			HIP_TRY(hipMemcpyAsync(&heads, d_nheads, 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			if (heads == nkeys) break;                                      // every string stands alone: ranks are final
			SBL_CHECK(off + csym < k || heads == nkeys, SBL_ERR_INTERNAL, "two bifurcation k-mers compare equal over their whole length");
		}
		k_fp_rank_ids<<<nblocks(nkeys, 256), 256, 0, s>>>(L.rank.as<unsigned>(), L.payload.as<unsigned>(), gp, nkeys, L.pairids.as<unsigned>());
	}
	c->bif_count = nkeys;
	const unsigned *myids = L.pairids.as<unsigned>() + 2 * pair0;                 // ids of MY pairs (the members carry my pair indices)
	for (int st = 0; st < 2; st++) {
		c->d_bif[st].ensure(elem_capacity * 4);
		HIP_TRY(hipMemsetAsync(c->d_bif[st].p, 0xFF, elem_capacity * 4, s));
	}
	if (sh) {
		// marks of my buckets' member positions, gathered and scattered into the dense arrays everywhere (16 B per member position)
		for (int st = 0; st < 2; st++) {
			L.keys.ensure((size_t)nmem * 4 + 16); L.skeys.ensure((size_t)nmem * 4 + 16);
			if (nmem) k_fp_mark_pairs<<<nblocks(nmem, 256), 256, 0, s>>>(L.members.as<u64>(), nmem, k, myids, (unsigned)st, L.keys.as<unsigned>(), L.skeys.as<unsigned>());
			HIP_TRY(hipGetLastError());
			const size_t tot = allgatherv((const char *)L.keys.as<unsigned>(), (size_t)nmem * 4, L.gel, nullptr) / 4;
			allgatherv((const char *)L.skeys.as<unsigned>(), (size_t)nmem * 4, L.gid, nullptr);
			if (tot) k_scatter_marks<<<nblocks(tot, 256), 256, 0, s>>>(L.gel.as<unsigned>(), L.gid.as<unsigned>(), tot, c->d_bif[st].as<unsigned>());
		}
	} else {
	if (nmem) k_fp_marks<<<nblocks(nmem, 256), 256, 0, s>>>(L.members.as<u64>(), nmem, k, myids, c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>());
	// The ordered (element, id) lists of the marks (what sbl_compact_marks makes by scanning every element of both mark arrays twice):
	// with few members -- a cascade state has hundreds, 900 Mbp of random sequence 40 -- sorting them is next to nothing (6.5 ms of
	// scans at 900 Mbp).  A window is a member once, so no element appears twice in a strand's list.  (SBL_FP_DENSE_MARKS=1: test switch.)
	if ((size_t)nmem * 16 <= E && getenv("SBL_FP_DENSE_MARKS") == nullptr) {
		for (int st = 0; st < 2; st++) {
			c->d_melem[st].ensure((size_t)nmem * 4 + 16); c->d_mid[st].ensure((size_t)nmem * 4 + 16);
			c->nmarks[st] = nmem;
			if (!nmem) continue;
			L.keys.ensure((size_t)nmem * 4 + 16); L.skeys.ensure((size_t)nmem * 4 + 16);
			k_fp_mark_pairs<<<nblocks(nmem, 256), 256, 0, s>>>(L.members.as<u64>(), nmem, k, myids, (unsigned)st, L.keys.as<unsigned>(), L.skeys.as<unsigned>());
			size_t tmp = 0;
			HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, L.keys.as<unsigned>(), c->d_melem[st].as<unsigned>(), L.skeys.as<unsigned>(), c->d_mid[st].as<unsigned>(), nmem, 0, 32, s));
			L.tmp.ensure(tmp);
			HIP_TRY(rocprim::radix_sort_pairs(L.tmp.p, tmp, L.keys.as<unsigned>(), c->d_melem[st].as<unsigned>(), L.skeys.as<unsigned>(), c->d_mid[st].as<unsigned>(), nmem, 0, 32, s));
		}
		c->marks_compact_ready = true;
	}
	}
	c->stats.exchange_ms = xms;
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
	c->stats.kmer_table_ms = ms;
	size_t positions = 0;
	for (uint32_t ch = 0; ch < c->nchr; ch++) { const size_t len = c->sepidx[ch + 1] - c->sepidx[ch] - 1; if (len >= k) positions += len - k + 1; }
	// algorithmic bytes (k > 32): 2-bit sequence once + one 32-B slot read and written per base position + k/4 B per verified occurrence
	c->stats.kmer_table_bytes = positions * 64 + E / 4 + (size_t)nmem * (k / 4);
	c->stats.strand_kmers = 2 * positions;
	c->stats.bif_count = nkeys;
	return true;
}
