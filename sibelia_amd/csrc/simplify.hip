// simplify.hip -- BlockFinder::PerformGraphSimplifications (reference src/blockfinder.cpp:78-98) on the GPU:
//   enumeration (sbl_api.hip) -> instance lists (E2) -> SimplifyGraph rounds (simplify_steps.h) -> copy-back (T3).
// Everything that touches sequence or graph data is a kernel in this file; the host only sequences
// launches (simplify_driver.h) and reads a 64-byte counter block back per round.
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

#include <chrono>

#include "sbl_ctx.h"
#include "sbl_comm.h"
#include "kmer_kernels.h"
// cycle counters of the decision loops (bulge_txn.h: BT_PROF_ADD), device only
__device__ unsigned long long g_phase_cycles[24];   // SBL_PHASES=1 debug: summed s_memtime deltas of k_commit's phases
#if defined(__HIP_DEVICE_COMPILE__)
#define BT_PROF_T0(t) do { if ((t).prof) (t).prof_t = __builtin_readcyclecounter(); } while (0)
#define BT_PROF_ADD(t, i) do { if ((t).prof) { unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(&g_phase_cycles[i], n_ - (t).prof_t); (t).prof_t = n_; } } while (0)
#endif
#include "simplify_driver.h"

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// ---- explicit address spaces for the transaction scratch -----------------------------------------------------------------
// The scratch arrays of a transaction (bulge_txn.h: BulgeWork) are reached through pointers kept in LDS, some into the fast scratch
// (LDS), some into the arena (HBM): to the compiler they are generic pointers, i.e. FLAT loads and stores.  On gfx9 a pending FLAT
// operation forces every later wait to s_waitcnt vmcnt(0) lgkmcnt(0) (it may complete out of order), so ONE flat store in a scan
// loop drains the bursts prefetched for the next windows as well: 73 of 86 waits in k_probe and 539 of 580 in k_commit were full
// drains.  These accessors pick the address space explicitly: global_* / ds_* instructions, partial vmcnt waits, real prefetch.
#if defined(__HIP_DEVICE_COMPILE__)
#define SBL_AS1 __attribute__((address_space(1)))
#define SBL_AS3 __attribute__((address_space(3)))
template <class T> __device__ __forceinline__ T ldg(const T *p) { return *(const SBL_AS1 T *)p; }                  // arena (t.alloc)
template <class T> __device__ __forceinline__ void stg(T *p, T v) { *(SBL_AS1 T *)p = v; }
template <class T> __device__ __forceinline__ T ldx(const T *p)                                                     // fast scratch or arena (t.alloc2 / falloc)
{ return __builtin_amdgcn_is_shared((const void *)p) ? *(const SBL_AS3 T *)p : *(const SBL_AS1 T *)p; }
template <class T> __device__ __forceinline__ void stx(T *p, T v) { if (__builtin_amdgcn_is_shared((const void *)p)) *(SBL_AS3 T *)p = v; else *(SBL_AS1 T *)p = v; }
#else       // (the host pass of hipcc only parses the kernels)
template <class T> __device__ __forceinline__ T ldg(const T *p) { return *p; }
template <class T> __device__ __forceinline__ void stg(T *p, T v) { *p = v; }
template <class T> __device__ __forceinline__ T ldx(const T *p) { return *p; }
template <class T> __device__ __forceinline__ void stx(T *p, T v) { *p = v; }
#endif

// ------------------------------------------------------------------------------------------- graph construction kernels
__global__ void __launch_bounds__(256) k_init_links(unsigned *__restrict__ nx, unsigned *__restrict__ pv, unsigned *__restrict__ nodeof0,
                                                    unsigned *__restrict__ nodeof1, uint8_t *__restrict__ ch, size_t E, size_t cap)
{
	size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= cap) return;
	if (e < E) { nx[e] = e + 1 < E ? (unsigned)(e + 1) : SBL_NONE; pv[e] = e ? (unsigned)(e - 1) : SBL_NONE; }
	else { nx[e] = pv[e] = SBL_NONE; ch[e] = BT_DEAD_CHAR; }
	nodeof0[e] = nodeof1[e] = SBL_NONE;
}

// sort key of an instance: (id << 32) | order, where ascending order reproduces the initial slist order of
// BifurcationStorage (front insertion while scanning (chr,pos) ascending, reference src/indexedsequence.cpp:51-67
// + src/bifurcationstorage.cpp:122): + list = elements descending; - list = chromosomes descending, elements ascending.
__global__ void __launch_bounds__(256) k_instance_keys(const unsigned *__restrict__ elem, const unsigned *__restrict__ id, unsigned n, unsigned strand,
                                                       const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E, unsigned ordbits,
                                                       unsigned long long *__restrict__ keys, unsigned *__restrict__ midx)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	midx[i] = i;                                    // payload of the sort: index into the positional (compact) mark arrays
	unsigned e = elem[i], ord;
	if (strand == 0) ord = E - 1u - e;              // (< E: the order field takes ordbits = bits of 2 E, the key id_bits + ordbits -- fewer radix passes than 64)
	else { unsigned c = chr_of(sepidx, nchr, e); ord = (E - sepidx[c + 1]) + (e - sepidx[c]); }
	keys[i] = ((unsigned long long)id[i] << ordbits) | ord;
}

__global__ void __launch_bounds__(256) k_build_lists(const unsigned long long *__restrict__ skeys, const unsigned *__restrict__ smidx, const unsigned *__restrict__ melem, unsigned n,
                                                     unsigned node_base, unsigned strand, unsigned ordbits, unsigned *__restrict__ nslot, unsigned *__restrict__ nnext, unsigned *__restrict__ nidst,
                                                     uint8_t *__restrict__ ndead, unsigned *__restrict__ head, unsigned *__restrict__ lsize,
                                                     unsigned *__restrict__ nodeof, unsigned *__restrict__ nmark)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned j = smidx[i];
	unsigned id = (unsigned)(skeys[i] >> ordbits), nd = node_base + i, e = melem[j];
	nmark[nd] = j;                                  // where the instance sits in the positional mark arrays (k_snapshot_first)
	bool last = i + 1 >= n || (unsigned)(skeys[i + 1] >> ordbits) != id;
	bool first = i == 0 || (unsigned)(skeys[i - 1] >> ordbits) != id;
	nslot[nd] = e; ndead[nd] = 0; nidst[nd] = (id << 1) | strand;
	nnext[nd] = last ? SBL_NONE : nd + 1;
	nodeof[e] = nd;
	if (first) {
		// the list's size = the length of its run in the sorted array (an atomic per instance kept this kernel in issue stalls for two
		// thirds of its time: SQ_WAIT_INST_ANY 66 %, profiles/r03_sq_counters.json)
		head[id] = nd;
		unsigned len = 1;
		while (i + len < n && (unsigned)(skeys[i + len] >> ordbits) == id) len++;
		lsize[id] = len;
	}
}

// largest number of instances of any id (sizes the per-transaction scratch arena)
// Snapshot order: ids sorted by where (one of) their instances lies, so that the workgroups resident at the same time scan
// overlapping windows (an element is covered by ~17 windows at 8 strains) and meet in L2 instead of re-reading HBM.
__global__ void __launch_bounds__(256) k_id_position_keys(const unsigned *__restrict__ head0, const unsigned *__restrict__ head1, const unsigned *__restrict__ nslot,
                                                          unsigned nid, unsigned long long *__restrict__ keys, unsigned *__restrict__ ids)
{
	unsigned id = blockIdx.x * blockDim.x + threadIdx.x;
	if (id >= nid) return;
	unsigned nd = head0[id] != BT_NONE ? head0[id] : head1[id];
	keys[id] = nd != BT_NONE ? nslot[nd] : 0xFFFFFFFFull;
	ids[id] = id;
}

__global__ void __launch_bounds__(256) k_max_instances(const unsigned *__restrict__ l0, const unsigned *__restrict__ l1, unsigned nid, unsigned *__restrict__ out)
{
	unsigned m = 0;
	for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nid; i += gridDim.x * blockDim.x) { unsigned v = l0[i] + l1[i]; m = v > m ? v : m; }
	for (int d = 32; d > 0; d >>= 1) { unsigned v = __shfl_down(m, d); m = v > m ? v : m; }
	if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// WSYNC(): synchronising the lanes of a ONE-wave workgroup (every kernel of the ordered rounds except k_reserve runs one wave per
// transaction).  Round 4 suspected the fence of __syncthreads() -- s_waitcnt vmcnt(0) lgkmcnt(0), a full memory round trip for stores nobody
// else waits for -- behind the 583 vmcnt(0) waits of k_commit and replaced it by a wavefront-scope fence + lgkmcnt(0).  The ISA did not
// change (593 -> 592): with __launch_bounds__(64) the compiler already knows that workgroup scope IS wavefront scope and emits neither a
// barrier nor a wait for it.  The vmcnt(0) waits are data dependencies and FLAT accesses; the macro stays as a marker of intent.
#define WSYNC() __syncthreads()

// ------------------------------------------------------------------------------------------- SimplifyGraph kernels
// Separators by SLOT.  A walk stops before a separator; it used to recognise one by its character -- a load of its own per element
// (one in four or five of a neighbourhood walk's loads, and what a round kernel costs is the number of memory instructions it issues).
// Separators never move during a stage and a walk never leaves its chromosome, so the only separators it can meet are the two that
// bound the chromosome of its first element: two compares.  Valid for walks that start at an ORIGINAL slot (the chromosome of a freshly
// inserted element is not known without looking) with at most 64 separators (one lane each); otherwise by == false and the character
// is loaded as before.
struct SepBounds { unsigned lo, hi; bool by; };
__device__ __forceinline__ SepBounds sep_bounds(const GraphView &g, const unsigned *s_sep /* LDS copy of g.sep, 64 entries, padded with BT_NONE */, unsigned e0, unsigned lane)
{
	SepBounds r; r.lo = r.hi = BT_NONE; r.by = false;
	if (!s_sep || e0 >= g.norig) return r;
	const unsigned sv = s_sep[lane];
	const unsigned long long le = __ballot(sv <= e0), ge = __ballot(sv != BT_NONE && sv >= e0);
	if (!le || !ge) return r;
	r.lo = __shfl(sv, 63 - (unsigned)__builtin_clzll(le));
	r.hi = __shfl(sv, (unsigned)__builtin_ctzll(ge));
	r.by = true;
	return r;
}

// Start stamp of a round kernel: the first workgroup writes the device wall clock (constant rate, hipDeviceAttributeWallClockRate) into
// the round's slot.  The kernels of a stream run back to back, so the difference of two consecutive start stamps is what a kernel cost,
// launch gap included -- per-kernel times of every round for one 8-byte store each, where an event pair around a kernel costs ~8 us of
// barrier packets (1.7 - 2.2 ms of a 100 ms stage for probe + reserve + commit).
__device__ __forceinline__ void round_stamp(const GraphView &g, unsigned which)
{
	if (g.tstamp && blockIdx.x == 0 && threadIdx.x == 0) g.tstamp[g.tslot + which] = wall_clock64();
}
// ---- wave-cooperative window scan ---------------------------------------------------------------------------
// Fills instance i's window cache (bulge_txn.h: BulgeWork) with 64 lanes: the same values bt_scan_instance
// writes, but 64 consecutive slots are tested per step and only real link breaks re-anchor the walk.
__device__ __forceinline__ void wave_stamp(const GraphView &g, unsigned stampv, unsigned tid, unsigned mode, unsigned id, unsigned r, unsigned wm /* wmax[r], loaded with the data */)
{
	// Exclusivity inside a round needs no per-element lock here: an owner holds every id marked in the range it reserved
	// (2(D+k+2)+k elements ahead of each instance), its scans reach D+k+2 elements, and k_commit checks after every
	// collapse that the elements it has deleted inside a window cannot carry a later scan / push beyond the reserved range.
	bool bad = false;
	unsigned other = BT_NONE;
	(void)stampv;
	if (mode == 2) atomicMax(&g.rmax[r], tid);
	if (wm > tid) bad = true;
	if (bad) {
		atomicMin(&g.ctr[CTR_VIOL], other < id ? other : id);
		if (atomicCAS(&g.ctr[CTR_DETAIL], 0u, other != BT_NONE ? 1u : 2u) == 0u) { g.ctr[CTR_DETAIL + 1] = r; g.ctr[CTR_DETAIL + 2] = other != BT_NONE ? other : wm - 1; g.ctr[CTR_DETAIL + 3] = id; g.ctr[CTR_DETAIL + 4] = mode; }
	}
}

// ListPositions (bifurcationstorage.h:59-72) with 64 lanes: + list then - list, chain order, dead nodes skipped.  Lists
// start out as runs of consecutive node indices (k_build_lists), so 64 nodes are read per step, speculatively, and the
// lanes whose predecessors all link consecutively are on the chain; front insertions and the end of a run re-anchor.
// The first step of BOTH lists is issued together (heads h0 / h1 given by the caller, who loads them while something else is
// going on): head -> nodes -> head -> nodes used to be four dependent memory round trips at the start of every probe,
// reservation and transaction.  emit(offset, node, strand, element, aux[node]) is called for every live node, in list order.
struct NodeChunk { unsigned nxt, dead, el, aux; bool inr; };
__device__ __forceinline__ NodeChunk node_chunk_load(const GraphView &g, unsigned cur, unsigned lane, const unsigned *__restrict__ aux)
{
	NodeChunk c;
	c.inr = cur != BT_NONE && (unsigned long long)cur + lane < g.cap_n;
	const unsigned nd = cur + lane;
	c.nxt = c.inr ? g.nnext[nd] : BT_NONE;
	c.dead = c.inr ? g.ndead[nd] : 1u;
	c.el = c.inr ? g.nslot[nd] : 0u;
	c.aux = c.inr && aux ? aux[nd] : 0u;
	return c;
}
template <class Emit>
__device__ __forceinline__ unsigned wave_list_nodes(const GraphView &g, unsigned h0, unsigned h1, unsigned lane, const unsigned *__restrict__ aux, Emit emit)
{
	const NodeChunk first[2] = { node_chunk_load(g, h0, lane, aux), node_chunk_load(g, h1, lane, aux) };      // both in flight
	unsigned m = 0;
	for (unsigned s = 0; s < 2; s++) {
		unsigned cur = s ? h1 : h0;
		bool prefetched = true;
		while (cur != BT_NONE) {
			const NodeChunk c = prefetched ? first[s] : node_chunk_load(g, cur, lane, aux);
			prefetched = false;
			const unsigned nd = cur + lane;
			const unsigned long long cont = __ballot(c.inr && c.nxt == nd + 1);
			const unsigned pre = cont == ~0ull ? 64u : (unsigned)__builtin_ctzll(~cont) + 1u;   // lanes 0 .. pre-1 are on the chain
			const bool on = lane < pre && c.inr;
			const unsigned long long lv = __ballot(on && !c.dead);
			const unsigned off = m + __popcll(lv & ((1ull << lane) - 1ull));
			if (on && !c.dead) emit(off, nd, s, c.el, c.aux);
			m += (unsigned)__popcll(lv);
			cur = __shfl(c.nxt, pre - 1);
		}
	}
	return m;
}
__device__ __forceinline__ unsigned wave_list_positions(const GraphView &g, unsigned h0, unsigned h1, const BulgeWork &w, unsigned lane)
{
	return wave_list_nodes(g, h0, h1, lane, nullptr, [&](unsigned off, unsigned nd, unsigned s, unsigned el, unsigned) {
		if (off < w.n) { stx(&w.start[off], (nd << 1) | s); stx(&w.sel[off], el); }
	});
}
// bt_setup with the positions listed by all lanes; `ok` lives in LDS
__device__ __forceinline__ bool wave_setup(const GraphView &g, Txn &t, BulgeWork &w, bool lite, unsigned lane, int &ok)
{
	const unsigned h0 = g.head[0][t.id], h1 = g.head[1][t.id];              // in flight while lane 0 lays the scratch out
	if (lane == 0) ok = bt_setup(t, w, lite, false) && !t.err ? 1 : 0;
	WSYNC();
	if (!ok) return false;
	unsigned m = wave_list_positions(g, h0, h1, w, lane);
	if (m != w.n && lane == 0) { t.err |= BT_ERR_SCRATCH; ok = 0; }          // cannot happen on a consistent graph
	WSYNC();
	return ok != 0;
}

// Burst: the loads of up to SCAN_BURST x 64 consecutive slots are issued together, assuming the list is laid out
// consecutively there (it almost always is); blocks are then consumed in order and the burst is abandoned at the
// first link break or separator.  One memory round trip per window instead of one per 64 elements.
enum { SCAN_BURST = 3 };
// wmv: the write stamp of every element of the burst, loaded WITH the burst (stamped scans only): the order check "nothing I read was
// written by a higher id" used to load it per 64-element block after the block had been consumed -- one exposed memory round trip
// per block, three per window, in every probe and every writer pass.
// What a burst HOLDS while it is in flight: four loaded values per element (character, own-strand mark, link, write stamp) -- the
// element indices, the in-range flags and the link of the PREVIOUS element (= the link the lane before loaded) are recomputed when the
// burst is consumed (burst_view).  Every lane loads unconditionally (lanes beyond the window read the window's first element): a
// predicated load becomes a branch around the instruction, and a load that may not have been issued makes the compiler wait for ALL
// outstanding loads wherever a later burst is consumed -- with unconditional loads it emits vmcnt(n) for exactly the younger ones, so
// the bursts of a whole batch of windows are in flight together (one memory round trip per SCAN_BATCH windows).
struct ScanBurst { unsigned chv[SCAN_BURST], bvl[SCAN_BURST], lnk[SCAN_BURST], wmv[SCAN_BURST]; unsigned cur, done; };
struct ScanView { unsigned cc[SCAN_BURST], plink[SCAN_BURST], chv[SCAN_BURST], bvl[SCAN_BURST], lnk[SCAN_BURST], wmv[SCAN_BURST]; bool inr[SCAN_BURST]; };
__device__ __forceinline__ void scan_burst_load(const GraphView &g, unsigned cur, unsigned dir, unsigned done, unsigned ws, unsigned lane, ScanBurst &b, unsigned mode = 0)
{
	(void)mode;
	b.cur = cur; b.done = done;
	const unsigned *__restrict__ link = dir ? g.pv : g.nx, *__restrict__ mark = g.bif[dir];
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) {
		const unsigned off = lane + 64u * u;
		const bool inr = done + off < ws && (dir ? off <= cur : (unsigned long long)cur + off < g.cap_e);
		const unsigned x = inr ? (dir ? cur - off : cur + off) : cur;
		b.chv[u] = g.ch[x];
		b.bvl[u] = mark[x];
		b.lnk[u] = link[x];
		b.wmv[u] = g.wmax[x >> BT_BLOCK_SHIFT];
	}
}
// the burst as its consumers see it (the values scan_burst_load used to produce directly)
__device__ __forceinline__ void burst_view(const GraphView &g, const ScanBurst &b, unsigned dir, unsigned ws, unsigned lane, unsigned mode, ScanView &v)
{
	const unsigned cur = b.cur, done = b.done;
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) {
		const unsigned off = lane + 64u * u;
		v.inr[u] = done + off < ws && (dir ? off <= cur : (unsigned long long)cur + off < g.cap_e);
		v.cc[u] = dir ? cur - off : cur + off;
		unsigned prev = __shfl_up(b.lnk[u], 1);                           // the link loaded by the lane before: the previous element's link
		if (u > 0) { const unsigned last = __shfl(b.lnk[u > 0 ? u - 1 : 0], 63); if (lane == 0) prev = last; }
		v.plink[u] = v.inr[u] && off ? prev : v.cc[u];
		v.chv[u] = v.inr[u] ? b.chv[u] : 0u;
		v.bvl[u] = v.inr[u] ? b.bvl[u] : BT_NONE;
		v.lnk[u] = v.inr[u] ? b.lnk[u] : BT_NONE;
		v.wmv[u] = mode && v.inr[u] ? b.wmv[u] : 0u;
	}
}

// one burst of a window scan: up to SCAN_BURST blocks of 64 consecutive slots, consumed in order, abandoned at the first link break or
// separator.  (A function of its own, always inlined: as a lambda inside wave_scan_instance it stayed out of line in the largest
// kernels, and a burst handed to it by reference was parked in scratch memory, every load waited for one by one.)
struct ScanState { unsigned cur, done, wl, nm, lastc; bool finished; };
__device__ __forceinline__ void scan_consume(const GraphView &g, const BulgeWork &w, unsigned i, unsigned lane, unsigned stampv, unsigned tid, unsigned mode, unsigned id,
                                             const ScanBurst &raw, ScanState &s, unsigned dir, unsigned ws, bool lite, unsigned mks,
                                             unsigned *wel, unsigned *wbf, uint8_t *wch, unsigned long long *wmk)
{
	ScanView bst;
	burst_view(g, raw, dir, ws, lane, mode, bst);
	const size_t base = (size_t)i * ws;
	const unsigned kk = g.k;
	const unsigned burst_done = s.done;
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) {
		if (burst_done + 64u * u >= ws) break;
		const unsigned c = bst.cc[u], done = s.done;
		unsigned long long ml = __ballot(bst.inr[u] && bst.plink[u] == c);
		unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);
		if (pre == 0) { s.cur = BT_NONE; s.finished = true; break; }     // cannot happen for u = 0; for u > 0 handled by the re-anchor below
		bool mine = lane < pre;
		unsigned long long ms = __ballot(mine && bst.chv[u] == BT_SEP);
		unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
		bool st = mine && lane <= stop;                               // the separator step itself is cached too
		unsigned bv = st ? bst.bvl[u] : BT_NONE;
		if (st) {
			if (!lite) { stg(&wel[base + done + lane], c); stg(&wch[base + done + lane], (uint8_t)bst.chv[u]); stg(&wbf[base + done + lane], bv); }
			if (done + lane == 0) stx(&w.wst[i], bv);
			if (done + lane == kk) stx(&w.wck[i], dir ? bt_comp((char)bst.chv[u]) : (char)bst.chv[u]);
		}
		{	// compact list of the marked steps (>= 1, before the separator), in step order
			bool marked = mine && lane < stop && bv != BT_NONE && done + lane > 0;
			unsigned long long mm = __ballot(marked);
			unsigned mo = s.nm + __popcll(mm & ((1ull << lane) - 1ull));
			if (marked && mo < mks) stx(&wmk[mo], ((unsigned long long)(done + lane) << 32) | bv);
			s.nm += __popcll(mm);
		}
		if (mode) {
			unsigned blk = c >> BT_BLOCK_SHIFT, pb = __shfl_up(blk, 1);
			if (st && bst.chv[u] != BT_SEP && (lane == 0 || pb != blk)) wave_stamp(g, stampv, tid, mode, id, blk, bst.wmv[u]);
		}
		if (stop < pre) { s.wl = done + stop; s.finished = true; break; }
		s.cur = __shfl(bst.lnk[u], pre - 1);
		s.lastc = __shfl(c, pre - 1);
		s.done = done + pre;
		if (pre < 64 || s.cur != (dir ? s.lastc - 1 : s.lastc + 1)) break;      // link break: re-anchor with a fresh burst
	}
}

// pre_burst: the first burst of this window when it was issued ahead of time (while the previous windows were being consumed); it is
// consumed from the registers it was loaded into
template <bool HAVE_PRE>
__device__ __forceinline__ void wave_scan_instance_t(const GraphView &g, const BulgeWork &w, unsigned i, unsigned lane,
                                                     unsigned stampv, unsigned tid, unsigned mode, unsigned id, const ScanBurst pre_burst)
{
	const unsigned packed = ldx(&w.start[i]), dir = packed & 1u, ws = w.ws;
	ScanState s;
	s.cur = ldx(&w.sel[i]); s.done = 0; s.wl = ws; s.nm = 0; s.lastc = 0; s.finished = false;
	unsigned nb = 0;
	const bool lite = w.lite;
	const unsigned mks = w.mks;
	unsigned *const wel = w.wel, *const wbf = w.wbf, *const wbk = w.wbk; uint8_t *const wch = w.wch;
	unsigned long long *const wmk = reinterpret_cast<unsigned long long *>(w.wmk) + (size_t)i * mks;
	if (s.cur != BT_NONE) {
		if (HAVE_PRE) scan_consume(g, w, i, lane, stampv, tid, mode, id, pre_burst, s, dir, ws, lite, mks, wel, wbf, wch, wmk);
		else { ScanBurst bst; scan_burst_load(g, s.cur, dir, 0, ws, lane, bst, mode); scan_consume(g, w, i, lane, stampv, tid, mode, id, bst, s, dir, ws, lite, mks, wel, wbf, wch, wmk); }
	}
	while (s.done < ws && s.cur != BT_NONE && !s.finished) {
		if (!lite && s.cur != (dir ? s.lastc - 1 : s.lastc + 1)) {        // the walk leaves consecutive slots here
			if (lane == 0 && nb < BT_MAX_BREAKS) stg(&wbk[i * BT_MAX_BREAKS + nb], s.done);
			nb++;
		}
		ScanBurst bst;
		scan_burst_load(g, s.cur, dir, s.done, ws, lane, bst, mode);
		scan_consume(g, w, i, lane, stampv, tid, mode, id, bst, s, dir, ws, lite, mks, wel, wbf, wch, wmk);
	}
	if (lane == 0) { stx(&w.wlen[i], s.wl < ws ? s.wl : ws); stx(&w.wmn[i], s.nm); if (!lite) stx(&w.wnb[i], nb); if (s.nm > mks) *const_cast<bool *>(&w.mk_overflow) = true; }
}

__device__ __forceinline__ void wave_scan_instance(const GraphView &g, const BulgeWork &w, unsigned i, unsigned lane,
                                                   unsigned stampv, unsigned tid, unsigned mode, unsigned id, const ScanBurst *pre_burst = nullptr)
{
	if (pre_burst) wave_scan_instance_t<true>(g, w, i, lane, stampv, tid, mode, id, *pre_burst);
	else { ScanBurst none; wave_scan_instance_t<false>(g, w, i, lane, stampv, tid, mode, id, none); }
}

// windows first, first + stride, ... of the cache, SCAN_BATCH at a time: the first bursts of a whole batch are issued together and
// then consumed in order (see ScanBurst: one memory round trip per batch instead of one per window)
#ifndef SCAN_BATCH
#define SCAN_BATCH 4
#endif
__device__ __forceinline__ void wave_scan_all(const GraphView &g, const BulgeWork &w, unsigned lane, unsigned stampv, unsigned tid, unsigned mode, unsigned id,
                                              unsigned first = 0, unsigned stride = 1)
{
	const unsigned n = w.n, ws = w.ws;
	// (Round 5 tried "light" bursts here for windows in pristine blocks without a write stamp above the runner -- no link and no stamp loads,
	// two of the four per element, decided from the block records of GraphView::bidx: k_commit + 0.6 ms.  The record look-up is a dependent
	// round trip in front of every batch of bursts; as in round 3, a load only pays when it disappears WITHOUT bookkeeping in its place.)
	for (unsigned i = first; i < n; i += SCAN_BATCH * stride) {
		unsigned sel[SCAN_BATCH], dir[SCAN_BATCH];
		ScanBurst b[SCAN_BATCH];
#pragma unroll
		for (int j = 0; j < SCAN_BATCH; j++) {                            // (all look-ups first: they may be loads from the arena themselves)
			const unsigned x = i + j * stride < n ? i + j * stride : i;   // a short last batch loads its first window again
			sel[j] = ldx(&w.sel[x]); dir[j] = ldx(&w.start[x]) & 1u;
		}
#pragma unroll
		for (int j = 0; j < SCAN_BATCH; j++) scan_burst_load(g, sel[j], dir[j], 0, ws, lane, b[j], mode);
#pragma unroll
		for (int j = 0; j < SCAN_BATCH; j++) {
			if (i + j * stride >= n) break;
			wave_scan_instance_t<true>(g, w, i + j * stride, lane, stampv, tid, mode, id, b[j]);
		}
	}
}

// AnyBulges VERDICT with 64 lanes.  "Some bulge group gets a second member" is an order-free predicate: there is an
// id b that two instances with different endChars both reach (steps 1 .. min(D, window) - 1, before their own id
// recurs) -- whichever iteration order boost::unordered_map has.  Marks are hashed into a small LDS table that
// collects the set of endChars per reached id.  Returns -1 when the marks do not fit (caller falls back to lane 0).
#define VT_SLOTS 512u
struct VerdictTable { unsigned key[VT_SLOTS]; unsigned mask[VT_SLOTS]; };

// one instance's marks into the verdict table: 1 = some id is now reached by two instances with different endChars, -1 = the table
// could fill up, 0 = nothing yet.  `distinct` counts the occupied slots.
__device__ __forceinline__ int wave_verdict_instance(const GraphView &g, const BulgeWork &w, VerdictTable &vt, unsigned lane, unsigned i, unsigned &distinct)
{
	const unsigned D = g.D, k = g.k;
	bool found = false;
	{
		const unsigned len = ldx(&w.wlen[i]);
		if (len < k + 1) return 0;                                     // endChar == ' '
		const char ec = ldx(&w.wck[i]);
		const unsigned bit = ec == 'A' ? 1u : ec == 'C' ? 2u : ec == 'G' ? 4u : 8u;
		const unsigned lim = len < D ? len : D, nm = ldx(&w.wmn[i]), start = ldx(&w.wst[i]);
		const unsigned long long *mk = reinterpret_cast<const unsigned long long *>(w.wmk) + (size_t)i * w.mks;
		for (unsigned j0 = 0; j0 < nm; j0 += 64) {
			unsigned j = j0 + lane;
			unsigned long long v = j < nm ? ldx(&mk[j]) : ~0ull;
			unsigned b = (unsigned)v, step = (unsigned)(v >> 32);
			bool stop = j >= nm || step >= lim || b == start;
			unsigned long long ms = __ballot(stop);
			unsigned upto = ms ? (unsigned)__builtin_ctzll(ms) : 64u;    // marks before the first stop condition
			if (distinct + upto > (VT_SLOTS * 3) / 4) return -1;         // the table could fill up
			bool fresh = false;
			if (lane < upto) {
				unsigned h = (b * 2654435761u) >> 23;                    // 9 bits
				for (;;) {
					unsigned old = atomicCAS(&vt.key[h], BT_NONE, b);
					if (old == BT_NONE || old == b) {
						fresh = old == BT_NONE;
						unsigned m = atomicOr(&vt.mask[h], bit) | bit;
						if (m & (m - 1)) found = true;
						break;
					}
					h = (h + 1) & (VT_SLOTS - 1);
				}
			}
			distinct += (unsigned)__popcll(__ballot(fresh));
			if (__any(found)) return 1;                                  // a second member for some group: verdict reached
			if (upto < 64) break;
		}
	}
	return __any(found) ? 1 : 0;
}

__device__ __forceinline__ int wave_verdict(const GraphView &g, const BulgeWork &w, VerdictTable &vt, unsigned lane, bool table_ready = false)
{
	if (!table_ready) {                                                // (multi-wave callers clear the table before their own barrier)
		for (unsigned i = lane; i < VT_SLOTS; i += 64) { vt.key[i] = BT_NONE; vt.mask[i] = 0; }
		WSYNC();
	}
	unsigned distinct = 0;                                             // occupied slots (homologous instances repeat the same ids)
	for (unsigned i = 0; i < w.n; i++) {
		const int r = wave_verdict_instance(g, w, vt, lane, i, distinct);
		if (r) return r;
	}
	return 0;
}

// ---- first snapshot of a stage: a stream over the position-ordered marks ------------------------------------------------
// At the start of iteration 1 the list is still position-linear (element index = position) and the compacted marks of the
// enumeration (melem / mid per strand, ascending element) ARE every window: instance j of strand 0 sees the marks j+1, j+2, ...
// while melem - pos < min(D, distance to the chromosome end), strand 1 the marks j-1, j-2, ... -- a dozen consecutive 8-byte
// records instead of 150 x (link + character + mark) per instance.  k_mark_aux adds, per mark, the endChar of the instance
// (bulgeremoval.cpp:340-347) and its distance to the end of the chromosome in walk direction.
__global__ void __launch_bounds__(256) k_mark_aux(const unsigned *__restrict__ melem, unsigned n, unsigned strand, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                  const uint8_t *__restrict__ ch, unsigned k, unsigned *__restrict__ aux)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned e = melem[i], c = chr_of(sepidx, nchr, e);
	const unsigned dist = strand == 0 ? sepidx[c + 1] - e : e - sepidx[c];      // valid steps from the instance (inclusive) to the separator
	unsigned bit = 0;
	if (dist >= k + 1) {                                                         // ProperKMer(k + 1): endChar = character at step k, oriented
		const uint8_t x = strand == 0 ? ch[e + k] : ch[e - k];
		const unsigned code = x == 'A' ? 0u : x == 'C' ? 1u : x == 'G' ? 2u : 3u;
		bit = 1u << (strand == 0 ? code : 3u - code);
	}
	aux[i] = (bit << 24) | (dist < 0xFFFFFFu ? dist : 0xFFFFFFu);
}

struct MarkStream { const unsigned *elem[2], *id[2], *aux[2]; unsigned n[2]; };

// AnyBulges verdict (see wave_verdict) of every id on the pristine graph; need[id] = 2 (known live) / 0 (clean) / 1 (the LDS
// table could not decide: the probe of its round does).  Four instances per step, 16 lanes each.
// plo / phi: the slice of the positional order this GPU looks at (everything, or its share when the read-only phases are split over
// the attached GPUs: DeviceBackend::snapshot_all)
__global__ void __launch_bounds__(64) k_snapshot_first(GraphView g, MarkStream ms, const unsigned *__restrict__ nmark, const unsigned *__restrict__ perm, unsigned plo, unsigned phi)
{
	__shared__ VerdictTable vt;
	const unsigned lane = threadIdx.x, sub = lane >> 4, sl = lane & 15u;
	const unsigned per = gridDim.x >> 3, slot = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);      // XCD-aware positional order, as k_snapshot
	for (unsigned base = plo; base < phi; base += gridDim.x) {
		if (base + slot >= phi) continue;
		const unsigned id = perm[base + slot];
		const unsigned n0 = g.lsize[0][id], n1 = g.lsize[1][id], n = n0 + n1;
		if (n < 2) { if (lane == 0) g.need[id] = 0; continue; }
		const unsigned h0 = g.head[0][id], h1 = g.head[1][id];                     // initial lists: runs of consecutive nodes (k_build_lists)
		{	// a group only gets a second member from an instance with a DIFFERENT endChar (see probe_endchars): one character per instance first
			unsigned bits = 0;
			for (unsigned i = lane; i < n; i += 64) { const unsigned s = i >= n0 ? 1u : 0u, nd = s ? h1 + (i - n0) : h0 + i; bits |= ms.aux[s][nmark[nd]] >> 24; }
#pragma unroll
			for (int d = 32; d > 0; d >>= 1) bits |= __shfl_xor(bits, d);
			if (__popc(bits) <= 1) { if (lane == 0) g.need[id] = 0; continue; }
		}
		WSYNC();
		for (unsigned i = lane; i < VT_SLOTS; i += 64) { vt.key[i] = BT_NONE; vt.mask[i] = 0; }
		WSYNC();
		bool found = false, undecided = false;
		unsigned distinct = 0;
		for (unsigned ib = 0; ib < n && !found && !undecided; ib += 4) {
			const unsigned i = ib + sub;
			const bool act = i < n;
			const unsigned s = act && i >= n0 ? 1u : 0u;
			const unsigned nd = s ? h1 + (i - n0) : h0 + i;
			const unsigned j = act ? nmark[nd] : 0u;
			const unsigned ax = act ? ms.aux[s][j] : 0u, pos = act ? ms.elem[s][j] : 0u;
			const unsigned bit = ax >> 24, dist = ax & 0xFFFFFFu, lim = dist < g.D ? dist : g.D;
			bool go = act && bit != 0;                                              // endChar == ' ': the instance takes no part
			for (unsigned t = 0; __any(go); t += 16) {
				const unsigned off = t + sl;
				const bool inr = go && (s == 0 ? (unsigned long long)j + 1 + off < ms.n[0] : off < j);
				const unsigned jj = s == 0 ? j + 1 + off : j - 1 - off;
				const unsigned p = inr ? ms.elem[s][jj] : 0u, b = inr ? ms.id[s][jj] : BT_NONE;
				const unsigned step = s == 0 ? p - pos : pos - p;
				const bool stop = !inr || step >= lim || b == id;                   // window end, or the instance's own id recurs
				const unsigned long long bal = __ballot(stop);
				const unsigned grp = (unsigned)(bal >> (sub * 16)) & 0xFFFFu;
				const unsigned upto = grp ? (unsigned)__builtin_ctz(grp) : 16u;      // marks of this instance before its first stop
				const unsigned total = (unsigned)__popcll(__ballot(go && sl < upto));
				if (distinct + total > (VT_SLOTS * 3) / 4) { undecided = true; break; }      // (uniform: the table could fill up)
				bool fresh = false;
				if (go && sl < upto) {
					unsigned h = (b * 2654435761u) >> 23;
					for (;;) {
						unsigned old = atomicCAS(&vt.key[h], BT_NONE, b);
						if (old == BT_NONE || old == b) {
							fresh = old == BT_NONE;
							unsigned m = atomicOr(&vt.mask[h], bit) | bit;
							if (m & (m - 1)) found = true;
							break;
						}
						h = (h + 1) & (VT_SLOTS - 1);
					}
				}
				distinct += (unsigned)__popcll(__ballot(fresh));
				if (__any(found)) { found = true; break; }
				if (upto < 16) go = false;
			}
		}
		if (lane == 0) g.need[id] = found ? 2 : undecided ? 1 : 0;
	}
}

// ---- later snapshots of a stage: the same stream over a LINEARISED copy of the marks -------------------------------------------
// After an iteration the list is no longer position-linear (collapses inserted and erased elements).  The segment ranking of
// the copy-back (k_seg_*) gives every live element its position in the list; elin[] is the inverse map.  The marks are then
// compacted in list order (mpos = list position, mid = id; nmark[node] = index of the instance's mark) and the verdict
// kernel is the stream again -- with the instance lists followed through their links, since they are no longer runs of nodes.
__global__ void __launch_bounds__(256) k_lin_positions(const uint8_t *__restrict__ ch, unsigned ne, const unsigned *__restrict__ flag, const unsigned *__restrict__ segidx,
                                                       const unsigned *__restrict__ seg_head, const unsigned long long *__restrict__ dist, unsigned long long total,
                                                       unsigned *__restrict__ lin, unsigned *__restrict__ elin)
{
	unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= ne) return;
	if (ch[e] == BT_DEAD_CHAR) { lin[e] = SBL_NONE; return; }
	unsigned seg = segidx[e] + flag[e] - 1;
	unsigned pos = (unsigned)(total - dist[seg] + (e - seg_head[seg]));
	lin[e] = pos; elin[pos] = e;
}
__global__ void __launch_bounds__(256) k_count_marks_lin(const unsigned *__restrict__ bif, const unsigned *__restrict__ elin, size_t n, unsigned *__restrict__ chunkcnt)
{
	__shared__ unsigned cnt;
	if (threadIdx.x == 0) cnt = 0;
	__syncthreads();
	size_t base = (size_t)blockIdx.x * 1024;
	unsigned c = 0;
	for (unsigned i = threadIdx.x; i < 1024; i += 256) { size_t p = base + i; c += (p < n && bif[elin[p]] != SBL_NONE); }
	atomicAdd(&cnt, c);
	__syncthreads();
	if (threadIdx.x == 0) chunkcnt[blockIdx.x] = cnt;
}
__global__ void __launch_bounds__(256) k_write_marks_lin(const unsigned *__restrict__ bif, const unsigned *__restrict__ elin, const unsigned *__restrict__ nodeof, size_t n,
                                                         const unsigned *__restrict__ chunkoff, unsigned *__restrict__ out_pos, unsigned *__restrict__ out_id, unsigned *__restrict__ nmark)
{
	__shared__ unsigned wsum[4];
	size_t base = (size_t)blockIdx.x * 1024 + (size_t)threadIdx.x * 4;
	unsigned ids[4], el[4], cn = 0;
#pragma unroll
	for (int i = 0; i < 4; i++) { size_t p = base + i; el[i] = p < n ? elin[p] : 0u; ids[i] = p < n ? bif[el[i]] : SBL_NONE; cn += ids[i] != SBL_NONE; }
	unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, incl = cn;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { unsigned v = __shfl_up(incl, d); if (lane >= (unsigned)d) incl += v; }
	if (lane == 63) wsum[wv] = incl;
	__syncthreads();
	unsigned off = chunkoff[blockIdx.x] + incl - cn;
	for (unsigned w = 0; w < wv; w++) off += wsum[w];
#pragma unroll
	for (int i = 0; i < 4; i++) if (ids[i] != SBL_NONE) { out_pos[off] = (unsigned)(base + i); out_id[off] = ids[i]; nmark[nodeof[el[i]]] = off; off++; }
}
// k_mark_aux in list coordinates: chromosome ends = list positions of the separators, characters through elin[]
__global__ void __launch_bounds__(256) k_mark_aux_lin(const unsigned *__restrict__ mpos, unsigned n, unsigned strand, const unsigned *__restrict__ sepelem, unsigned nchr,
                                                      const unsigned *__restrict__ lin, const unsigned *__restrict__ elin, const uint8_t *__restrict__ ch, unsigned k, unsigned *__restrict__ aux)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const unsigned p = mpos[i];
	unsigned lo = 0, hi = nchr;                               // chromosome c with lin[sep c] < p < lin[sep c + 1]
	while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (lin[sepelem[mid]] < p) lo = mid; else hi = mid; }
	const unsigned dist = strand == 0 ? lin[sepelem[lo + 1]] - p : p - lin[sepelem[lo]];
	unsigned bit = 0;
	if (dist >= k + 1) {
		const uint8_t x = ch[elin[strand == 0 ? p + k : p - k]];
		const unsigned code = x == 'A' ? 0u : x == 'C' ? 1u : x == 'G' ? 2u : 3u;
		bit = 1u << (strand == 0 ? code : 3u - code);
	}
	aux[i] = (bit << 24) | (dist < 0xFFFFFFu ? dist : 0xFFFFFFu);
}

#define SNAP_MAX_INST 256u
// AnyBulges verdict of the touched ids (incremental) on the linearised marks; ids with more than SNAP_MAX_INST instances or too
// many distinct marks for the LDS table get need = 1 (the probe of their round decides).
__global__ void __launch_bounds__(64) k_snapshot_stream(GraphView g, MarkStream ms, const unsigned *__restrict__ nmark, const unsigned *__restrict__ perm, int incremental, unsigned plo, unsigned phi)
{
	__shared__ VerdictTable vt;
	__shared__ unsigned s_inst[SNAP_MAX_INST];                    // (mark index << 1) | strand of every live instance, list order
	const unsigned lane = threadIdx.x, sub = lane >> 4, sl = lane & 15u;
	const unsigned per = gridDim.x >> 3, slot = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	for (unsigned base = plo; base < phi; base += gridDim.x) {
		if (base + slot >= phi) continue;
		const unsigned id = perm[base + slot];
		if (incremental && !g.touch[id]) { if (lane == 0) g.need[id] = 0; continue; }      // nobody touched it since its verdict was taken: still clean
		WSYNC();
		if (lane == 0) g.touch[id] = 0;
		// ---- ListPositions: + list then - list, live nodes only (64 nodes per step where the list is a run of consecutive nodes)
		const unsigned n = wave_list_nodes(g, g.head[0][id], g.head[1][id], lane, nmark, [&](unsigned off, unsigned, unsigned s, unsigned, unsigned mj) {
			if (off < SNAP_MAX_INST) s_inst[off] = (mj << 1) | s;
		});
		if (n < 2) { if (lane == 0) g.need[id] = 0; continue; }
		if (n > SNAP_MAX_INST) { if (lane == 0) g.need[id] = 1; continue; }
		WSYNC();
		{	// endChars first (see probe_endchars): all the same, or none at all => clean
			unsigned bits = 0;
			for (unsigned i = lane; i < n; i += 64) { const unsigned packed = s_inst[i]; bits |= ms.aux[packed & 1u][packed >> 1] >> 24; }
#pragma unroll
			for (int d = 32; d > 0; d >>= 1) bits |= __shfl_xor(bits, d);
			if (__popc(bits) <= 1) { if (lane == 0) g.need[id] = 0; continue; }
		}
		for (unsigned i = lane; i < VT_SLOTS; i += 64) { vt.key[i] = BT_NONE; vt.mask[i] = 0; }
		WSYNC();
		bool found = false, undecided = false;
		unsigned distinct = 0;
		for (unsigned ib = 0; ib < n && !found && !undecided; ib += 4) {
			const unsigned i = ib + sub;
			const bool act = i < n;
			const unsigned packed = act ? s_inst[i] : 0u, s = packed & 1u, j = packed >> 1;
			const unsigned ax = act ? ms.aux[s][j] : 0u, pos = act ? ms.elem[s][j] : 0u;
			const unsigned bit = ax >> 24, dist = ax & 0xFFFFFFu, lim = dist < g.D ? dist : g.D;
			bool go = act && bit != 0;
			for (unsigned t = 0; __any(go); t += 16) {
				const unsigned off = t + sl;
				const bool inr = go && (s == 0 ? (unsigned long long)j + 1 + off < ms.n[0] : off < j);
				const unsigned jj = s == 0 ? j + 1 + off : j - 1 - off;
				const unsigned p = inr ? ms.elem[s][jj] : 0u, b = inr ? ms.id[s][jj] : BT_NONE;
				const unsigned step = s == 0 ? p - pos : pos - p;
				const bool stop = !inr || step >= lim || b == id;
				const unsigned long long bal = __ballot(stop);
				const unsigned grp = (unsigned)(bal >> (sub * 16)) & 0xFFFFu;
				const unsigned upto = grp ? (unsigned)__builtin_ctz(grp) : 16u;
				const unsigned total = (unsigned)__popcll(__ballot(go && sl < upto));
				if (distinct + total > (VT_SLOTS * 3) / 4) { undecided = true; break; }
				bool fresh = false;
				if (go && sl < upto) {
					unsigned h = (b * 2654435761u) >> 23;
					for (;;) {
						unsigned old = atomicCAS(&vt.key[h], BT_NONE, b);
						if (old == BT_NONE || old == b) {
							fresh = old == BT_NONE;
							unsigned m = atomicOr(&vt.mask[h], bit) | bit;
							if (m & (m - 1)) found = true;
							break;
						}
						h = (h + 1) & (VT_SLOTS - 1);
					}
				}
				distinct += (unsigned)__popcll(__ballot(fresh));
				if (__any(found)) { found = true; break; }
				if (upto < 16) go = false;
			}
		}
		if (lane == 0) g.need[id] = found ? 2 : undecided ? 1 : 0;
	}
}

// AnyBulges verdict of every id against the graph at iteration start: one wave per id (64 lanes scan the windows,
// lane 0 evaluates the Boost-ordered map on the cached marks).
__global__ void __launch_bounds__(64) k_snapshot(GraphView g, uint8_t *arena, unsigned arena_bytes, int incremental, const unsigned *__restrict__ perm, unsigned plo, unsigned phi)
{
	__shared__ Txn t;
	__shared__ BulgeWork w;
	__shared__ VerdictTable vt;
	__shared__ int ok;
	__shared__ __attribute__((aligned(16))) uint8_t fast[2048];       // per-instance window summaries of typical ids
	const unsigned lane = threadIdx.x;
	uint8_t *mine = arena + (size_t)blockIdx.x * arena_bytes;
	// Workgroups are dealt to the 8 XCDs round robin (blockIdx & 7): every XCD takes a contiguous eighth of each chunk of
	// gridDim.x positions of the positional order, so the overlapping windows of neighbouring ids share that XCD's L2.
	const unsigned per = gridDim.x >> 3, slot = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
	for (unsigned base = plo; base < phi; base += gridDim.x) {
		if (base + slot >= phi) continue;
		const unsigned id = perm[base + slot];
		// incremental: an id nobody touched since its verdict was last taken is still clean
		if (incremental && !g.touch[id]) { if (lane == 0) g.need[id] = 0; continue; }
		WSYNC();
		if (lane == 0) { g.touch[id] = 0; t.init(g, id, 0, 0, mine, arena_bytes); t.fscr = fast; t.fscr_cap = sizeof fast; }
		WSYNC();
		wave_setup(g, t, w, true, lane, ok);
		if (ok) {
			wave_scan_all(g, w, lane, 0, 0, 0, id);
			WSYNC();
		}
		int verdict = ok ? wave_verdict(g, w, vt, lane) : 0;
		if (lane == 0) {
			bool v = verdict > 0;
			if (verdict < 0) { bt_end_chars(t, w); v = bt_any_bulges(t, w, true); }      // too many marks for the LDS table
			if (t.err & BT_ERR_SCRATCH) v = true;
			g.need[id] = v ? (verdict > 0 ? 2 : 1) : 0;                 // 2: known live, the first probe of the entry is skipped (a push resets it to 1)
		}
	}
}

// ---- one window of a probe, scan and verdict in one go ---------------------------------------------------------------------------
// The probe is issue-bound, not memory-bound (rocprofv3 SQ counters, round 3: its waves are actively issuing 26 % of their lifetime
// at ~3 waves per SIMD): wave_scan_instance writes per-window summaries and a compacted mark list, a barrier later
// wave_verdict_instance reads them back and ballots again.  For a window that lies in consecutive slots over its whole length --
// almost all do -- everything the verdict needs is in the registers of the burst: window length (first separator), endChar (step
// k), the marked steps before the window's end and before the instance's own id recurs.  Returns 1 (some id is now reached by two
// instances with different endChars), 0, -1 (the table could fill up), -2 (a link break inside the window, k or D beyond the burst:
// the generic pair of functions takes this window).
__device__ __forceinline__ int wave_probe_window(const GraphView &g, const ScanBurst &raw, unsigned dir, unsigned ws, VerdictTable &vt, unsigned lane,
                                                 unsigned id, unsigned tid, unsigned &distinct)
{
	ScanView b;
	burst_view(g, raw, dir, ws, lane, 3u, b);
	const unsigned k = g.k, D = g.D;
	if (k >= 64u * SCAN_BURST) return -2;
	unsigned firstbad = ~0u, firstsep = ~0u;
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) {
		const unsigned long long in = __ballot(b.inr[u]), good = __ballot(b.inr[u] && b.plink[u] == b.cc[u]), sep = __ballot(b.inr[u] && b.chv[u] == BT_SEP);
		const unsigned long long bad = in & ~good;
		if (bad && firstbad == ~0u) firstbad = 64u * u + (unsigned)__builtin_ctzll(bad);
		if (sep && firstsep == ~0u) firstsep = 64u * u + (unsigned)__builtin_ctzll(sep);
	}
	const unsigned covered = ws < 64u * SCAN_BURST ? ws : 64u * SCAN_BURST;      // steps the burst holds
	const unsigned len = firstsep < covered ? firstsep : covered;                // steps before the separator (wlen), as far as the burst shows
	if (firstbad < len || firstbad <= firstsep && firstbad < covered) return -2; // the walk leaves consecutive slots inside the window
	if (firstsep >= covered && covered < ws && D > covered) return -2;           // the window goes on beyond the burst
	// order check of everything read (mode 3): elements before the separator
	bool viol = false;
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) viol |= b.inr[u] && 64u * u + lane < len && b.wmv[u] > tid;
	if (__any(viol)) wave_stamp(g, 0, tid, 3, id, 0, tid + 1);
	if (len < k + 1) return 0;                                                   // endChar == ' '
	const unsigned kc = __shfl(b.chv[0], k & 63u), kc1 = SCAN_BURST > 1 ? __shfl(b.chv[1], k & 63u) : 0u, kc2 = SCAN_BURST > 2 ? __shfl(b.chv[2], k & 63u) : 0u;
	const unsigned craw = k < 64 ? kc : k < 128 ? kc1 : kc2;
	const char ec = dir ? bt_comp((char)craw) : (char)craw;
	const unsigned bit = ec == 'A' ? 1u : ec == 'C' ? 2u : ec == 'G' ? 4u : 8u;
	const unsigned start = __shfl(b.bvl[0], 0);
	const unsigned lim = len < D ? len : D;
	// marked steps 1 .. lim - 1, up to the first recurrence of the instance's own id
	unsigned firstown = ~0u;
	unsigned long long cand[SCAN_BURST];
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) {
		const unsigned step = 64u * u + lane;
		const bool c = b.inr[u] && step >= 1 && step < lim && b.bvl[u] != BT_NONE;
		cand[u] = __ballot(c);
		const unsigned long long own = __ballot(c && b.bvl[u] == start);
		if (own && firstown == ~0u) firstown = 64u * u + (unsigned)__builtin_ctzll(own);
	}
	unsigned total = 0;
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) {
		if (firstown != ~0u) {                                                   // keep the steps below firstown only
			const unsigned lo = 64u * u;
			cand[u] = firstown <= lo ? 0ull : firstown >= lo + 64u ? cand[u] : cand[u] & ((1ull << (firstown - lo)) - 1ull);
		}
		total += (unsigned)__popcll(cand[u]);
	}
	if (!total) return 0;
	if (distinct + total > (VT_SLOTS * 3) / 4) return -1;
	bool found = false, fresh_any = false;
	unsigned nfresh = 0;
#pragma unroll
	for (int u = 0; u < SCAN_BURST; u++) {
		bool fresh = false;
		if ((cand[u] >> lane) & 1ull) {
			const unsigned bb = b.bvl[u];
			unsigned h = (bb * 2654435761u) >> 23;
			for (;;) {
				unsigned old = atomicCAS(&vt.key[h], BT_NONE, bb);
				if (old == BT_NONE || old == bb) {
					fresh = old == BT_NONE;
					unsigned m = atomicOr(&vt.mask[h], bit) | bit;
					if (m & (m - 1)) found = true;
					break;
				}
				h = (h + 1) & (VT_SLOTS - 1);
			}
		}
		nfresh += (unsigned)__popcll(__ballot(fresh));
	}
	(void)fresh_any;
	distinct += nfresh;
	return __any(found) ? 1 : 0;
}

// Probe of the window entries between rounds (no writer runs): entries whose AnyBulges verdict is false NOW are retired
// without reservation (ss_probe); the others are flagged live and go through reserve / commit.
// The windows of a probed id, PROBE_BATCH at a time: the first bursts of a batch are in flight together (see ScanBurst / wave_scan_all)
// and their marks go straight into the verdict table -- an entry that IS live stops at the first id two instances with different endChars
// reach, without scanning the rest.  Returns the verdict (1 / 0; -1: undecided by the table).
#ifndef PROBE_BATCH
#define PROBE_BATCH 4
#endif
// (Recognising separators by their slot here as well -- one character per window instead of three loads -- was measured 0.9 ms SLOWER per
// stage: the probe is issue-bound, and the bounds of every window cost more instructions than the two 64-byte loads they save.)
__device__ __forceinline__ int probe_windows(const GraphView &g, BulgeWork &w, VerdictTable &vt, unsigned lane, unsigned id, unsigned tid)
{
	int verdict = 0;
	unsigned distinct = 0;
	const unsigned n = w.n, ws = w.ws;
	for (unsigned i = 0; i < n && verdict == 0; i += PROBE_BATCH) {
		unsigned sel[PROBE_BATCH], dir[PROBE_BATCH];
		ScanBurst b[PROBE_BATCH];
#pragma unroll
		for (int j = 0; j < PROBE_BATCH; j++) {
			const unsigned x = i + j < n ? i + j : i;
			sel[j] = ldx(&w.sel[x]); dir[j] = ldx(&w.start[x]) & 1u;
		}
#pragma unroll
		for (int j = 0; j < PROBE_BATCH; j++) scan_burst_load(g, sel[j], dir[j], 0, ws, lane, b[j], 3u);
#pragma unroll
		for (int j = 0; j < PROBE_BATCH; j++) {
			if (i + j >= n) break;
			int v = wave_probe_window(g, b[j], dir[j], ws, vt, lane, id, tid, distinct);
			if (v == -2) {                                                  // a link break inside the window (an earlier collapse): the generic pair
				wave_scan_instance(g, w, i + j, lane, 0, tid, 3, id, &b[j]);
				WSYNC();
				if (w.mk_overflow) return -1;                               // more marks than the LDS list holds: the generic path decides
				v = wave_verdict_instance(g, w, vt, lane, i + j, distinct);
			}
			if ((verdict = v) != 0) break;
		}
	}
	return verdict;
}

#define PROBE_WAVES 1u                       // waves per probed id (windows dealt out to them, wave 0 takes the verdict); more than one did not pay: most entries are cheap
// w0: first window entry of this launch (0, or the start of this GPU's share when the read-only phases are split over the attached GPUs)
// ---- endChar pre-pass of a probe.  AnyBulges can only give a group its second member when two instances of the id have DIFFERENT
// endChars (bulgeremoval.cpp:192-199: a branch is appended where visit[b].endChar != endChar[i]); an id whose instances all continue
// with the same character -- most ids next to a collapse do: the column at offset k carries no SNP in any strain 92 % of the time --
// is clean whatever its windows hold.  endChar needs the first k + 1 steps of a window only: one block of 64 slots and two arrays
// (+ the write stamps for the order check of what was read) instead of three blocks and four arrays per window, eight windows in flight.
// Returns 1: provably clean; 0: the full probe decides (different endChars, a link break inside the first k + 1 steps, k >= 63).
__device__ __forceinline__ int probe_endchars(const GraphView &g, const BulgeWork &w, unsigned lane, unsigned id, unsigned tid)
{
	const unsigned n = w.n, k = g.k;
	if (k >= 63u) return 0;
	if (k <= 31u) {
		// k + 1 <= 32 steps: TWO windows per wave instruction (lanes 0 - 31 / 32 - 63), sixteen windows in flight -- the probe is issue-bound
		const unsigned half = lane >> 5, hl = lane & 31u;
		const unsigned wantm = k == 31u ? 0xFFFFFFFFu : (1u << (k + 1)) - 1u;
		unsigned hmask = 0;
		bool hviol = false;
		for (unsigned i0 = 0; i0 < n; i0 += 16) {
			unsigned sel[8], dir[8], chv[8], lnk[8], wmv[8];
#pragma unroll
			for (int j = 0; j < 8; j++) { const unsigned x = i0 + 2 * j + half, xx = x < n ? x : i0; sel[j] = ldx(&w.sel[xx]); dir[j] = ldx(&w.start[xx]) & 1u; }
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const bool inr = hl <= k && (dir[j] ? hl <= sel[j] : (unsigned long long)sel[j] + hl < g.cap_e);
				const unsigned c = inr ? (dir[j] ? sel[j] - hl : sel[j] + hl) : sel[j];
				chv[j] = g.ch[c]; lnk[j] = (dir[j] ? g.pv : g.nx)[c]; wmv[j] = g.wmax[c >> BT_BLOCK_SHIFT];
			}
#pragma unroll
			for (int j = 0; j < 8; j++) {
				if (i0 + 2 * j >= n) break;
				const bool mine = i0 + 2 * j + half < n;
				const bool inr = hl <= k && (dir[j] ? hl <= sel[j] : (unsigned long long)sel[j] + hl < g.cap_e);
				const unsigned c = dir[j] ? sel[j] - hl : sel[j] + hl;
				const unsigned prev = __shfl_up(lnk[j], 1);
				const unsigned good = (unsigned)(__ballot(inr && (hl == 0 || prev == c)) >> (32u * half)) & wantm;
				const unsigned sep = (unsigned)(__ballot(inr && chv[j] == BT_SEP) >> (32u * half)) & wantm;
				const unsigned firstsep = sep ? (unsigned)__builtin_ctz(sep) : 64u, firstbad = good != wantm ? (unsigned)__builtin_ctz(~good) : 64u;
				const unsigned upto = firstsep < k + 1 ? firstsep : k + 1;
				if (__any(mine && (firstbad < upto || (firstbad == firstsep && firstsep < 64u)))) return 0;
				hviol |= mine && hl < upto && wmv[j] > tid;
				const unsigned craw = __shfl(chv[j], 32u * half + k);
				const char ec = dir[j] ? bt_comp((char)craw) : (char)craw;
				if (mine && firstsep > k) hmask |= ec == 'A' ? 1u : ec == 'C' ? 2u : ec == 'G' ? 4u : 8u;
			}
		}
		hmask |= __shfl_xor(hmask, 32);
		if (__popc(hmask) > 1) return 0;
		if (__any(hviol)) wave_stamp(g, 0, tid, 3, id, 0, tid + 1);
		return 1;
	}
	const unsigned long long want = (1ull << (k + 1)) - 1ull;
	unsigned mask = 0;
	bool viol = false;
	for (unsigned i0 = 0; i0 < n; i0 += 8) {
		unsigned sel[8], dir[8], chv[8], lnk[8], wmv[8];
#pragma unroll
		for (int j = 0; j < 8; j++) { const unsigned x = i0 + j < n ? i0 + j : i0; sel[j] = ldx(&w.sel[x]); dir[j] = ldx(&w.start[x]) & 1u; }
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const bool inr = lane <= k && (dir[j] ? lane <= sel[j] : (unsigned long long)sel[j] + lane < g.cap_e);
			const unsigned c = inr ? (dir[j] ? sel[j] - lane : sel[j] + lane) : sel[j];
			chv[j] = g.ch[c]; lnk[j] = (dir[j] ? g.pv : g.nx)[c]; wmv[j] = g.wmax[c >> BT_BLOCK_SHIFT];
		}
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (i0 + j >= n) break;
			const bool inr = lane <= k && (dir[j] ? lane <= sel[j] : (unsigned long long)sel[j] + lane < g.cap_e);
			const unsigned c = dir[j] ? sel[j] - lane : sel[j] + lane;
			const unsigned prev = __shfl_up(lnk[j], 1);
			const unsigned long long good = __ballot(inr && (lane == 0 || prev == c)) & want;
			const unsigned long long sep = __ballot(inr && chv[j] == BT_SEP) & want;
			const unsigned firstsep = sep ? (unsigned)__builtin_ctzll(sep) : 64u, firstbad = good != want ? (unsigned)__builtin_ctzll(~good) : 64u;
			const unsigned upto = firstsep < k + 1 ? firstsep : k + 1;      // steps of the walk that were read (the separator itself is never stamped)
			if (firstbad < upto || (firstbad == firstsep && firstsep < 64u)) return 0;      // the walk leaves consecutive slots before its endChar is known
			viol |= lane < upto && wmv[j] > tid;
			if (firstsep <= k) continue;                                    // fewer than k + 1 characters: endChar ' ', the instance takes no part
			const unsigned craw = __shfl(chv[j], k);
			const char ec = dir[j] ? bt_comp((char)craw) : (char)craw;
			mask |= ec == 'A' ? 1u : ec == 'C' ? 2u : ec == 'G' ? 4u : 8u;
		}
	}
	if (__popc(mask) > 1) return 0;
	if (__any(viol)) wave_stamp(g, 0, tid, 3, id, 0, tid + 1);
	return 1;
}

// ---- probe of a pending id from the BLOCK INDEX (round 5) --------------------------------------------------------------------------
// A window that lies in pristine 64-slot blocks (GraphView::bidx) is the slots a, a +- 1, ... themselves, so everything the verdict needs
// -- first separator, the character at step k, the marked steps before the window's end -- is in the three or four 32-byte records the
// window touches: one lane per (instance, block) loads its record, and only the MARKED slots' ids are gathered (a dozen per window
// instead of 175 x {character, mark, link, stamp}).  The recurrence of the instance's own id is known without a look at the marks:
// it is another instance of the same list, and the instances are in LDS.  No transaction state, no arena: the kernel runs at twice
// the occupancy of the walking probe (k_probe), which only sees the entries this one cannot serve -- an instance on an inserted
// element, a window that touches a block which is no longer pristine or carries a write stamp above the prober (the exact order check
// needs the elements), more instances than the LDS list holds -- flagged PROBE_UNSERVED in live[].
// Pass 1 takes the endChars alone (see probe_endchars); pass 2 the marks.
#define PROBE_UNSERVED 3u
__device__ unsigned long long g_rsv_ticks[8];      // SBL_TEST_FLAGS=32: summed wall-clock ticks of the reservation's phases (set-up, exclusive claims, ordering claims), entries, claims, instances
__device__ unsigned g_idx_stats[8];          // SBL_TRACE: probes by outcome of k_probe_idx (known live, < 2 instances, clean, live, table full, not served); reservations: instances served / walked
__device__ __forceinline__ unsigned long long idx_bits(int lo, int hi)      // bits lo .. hi-1 of a 64-bit word (clamped)
{
	lo = lo < 0 ? 0 : lo; hi = hi > 64 ? 64 : hi;
	if (hi <= lo) return 0ull;
	const unsigned long long up = hi >= 64 ? ~0ull : (1ull << hi) - 1ull;
	return up & ~((1ull << lo) - 1ull);
}
// verdict-table insert of one mark per lane (b == BT_NONE: none); true when some id is now reached by two different endChars
struct VtRef { unsigned *key, *mask; unsigned bits; };      // a verdict table of 1 << bits slots in (dynamic) LDS
__device__ __forceinline__ bool vt_insert(const VtRef &vt, unsigned b, unsigned bit, unsigned &distinct)
{
	bool fresh = false, found = false;
	if (b != BT_NONE) {
		unsigned h = (b * 2654435761u) >> (32u - vt.bits);
		for (;;) {
			const unsigned old = atomicCAS(&vt.key[h], BT_NONE, b);
			if (old == BT_NONE || old == b) {
				fresh = old == BT_NONE;
				const unsigned m = atomicOr(&vt.mask[h], bit) | bit;
				if (m & (m - 1u)) found = true;
				break;
			}
			h = (h + 1u) & ((1u << vt.bits) - 1u);
		}
	}
	distinct += (unsigned)__popcll(__ballot(fresh));
	return __any(found);
}
// One window WALKED by the wave, for the windows the index cannot serve (an inserted or erased element inside, an instance on an inserted
// element): up to `maxsteps` steps from element `a` on strand `dir`, 64 consecutive slots per memory round trip while the links allow it.
// Gives the window's length (steps before the separator), the raw character at step k, the order check of every element read, and -- when
// mks != nullptr -- the marked steps >= 1 before the own id recurs as (step, id) pairs in LDS (at most PIDX_WALK_MARKS; more: overflow).
struct WalkedWindow { unsigned len, craw, nm; bool viol, overflow; };
__device__ __forceinline__ WalkedWindow idx_walk_window(const GraphView &g, unsigned a, unsigned dir, unsigned maxsteps, unsigned lane, unsigned id, unsigned tid,
                                                        unsigned *mk_step, unsigned *mk_id, unsigned PIDX_WALK_MARKS /* entries of the two lists */)
{
	WalkedWindow r; r.len = maxsteps; r.craw = 0; r.nm = 0; r.viol = false; r.overflow = false;
	const unsigned k = g.k;
	const unsigned *__restrict__ link = dir ? g.pv : g.nx, *__restrict__ mark = g.bif[dir];
	unsigned cur = a, done = 0;
	bool open = true;                                                    // the own id has not recurred yet
	while (done < maxsteps && cur != BT_NONE) {
		const bool inr = done + lane < maxsteps && (dir ? lane <= cur : (unsigned long long)cur + lane < g.cap_e);
		const unsigned c = inr ? (dir ? cur - lane : cur + lane) : cur;
		const unsigned chv = g.ch[c], lnk = link[c], bv = mark[c], wm = g.wmax[c];
		const unsigned prev = __shfl_up(lnk, 1);
		const unsigned long long ml = __ballot(inr && (lane == 0 || prev == c));
		const unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);          // intact prefix, >= 1
		const unsigned long long ms = __ballot(lane < pre && chv == BT_SEP);
		const unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
		const bool proc = lane < pre && lane < stop;
		if (__any(proc && wm > tid)) r.viol = true;
		const unsigned upto = pre < stop ? pre : stop;
		if (k >= done && k < done + upto) r.craw = __shfl(chv, k - done);
		if (mk_step && open) {
			const unsigned long long own = __ballot(proc && done + lane >= 1u && bv == id);
			const unsigned ownat = own ? (unsigned)__builtin_ctzll(own) : 64u;
			const bool take = proc && lane < ownat && done + lane >= 1u && bv != BT_NONE;
			const unsigned long long tm = __ballot(take);
			const unsigned o = r.nm + (unsigned)__popcll(tm & ((1ull << lane) - 1ull));
			if (take && o < PIDX_WALK_MARKS) { mk_step[o] = done + lane; mk_id[o] = bv; }
			r.nm += (unsigned)__popcll(tm);
			if (own) open = false;
		}
		if (stop < pre) { r.len = done + stop; return r; }
		cur = __shfl(lnk, pre - 1u);
		done += pre;
	}
	if (r.nm > PIDX_WALK_MARKS) r.overflow = true;
	return r;
}
// returns 1 live, 0 clean, -1 the verdict table could fill up / a walked window has too many marks (k_probe decides)
__device__ __forceinline__ int probe_idx(const GraphView &g, const VtRef &vt, const unsigned *s_sel, const uint8_t *s_dir, unsigned *s_own, unsigned *mk_step, unsigned *mk_id, unsigned walk_marks,
                                         unsigned n, unsigned lane, unsigned id, unsigned tid)
{
	const unsigned VT_FILL = 3u << (vt.bits - 2u);                       // three quarters of the slots
	const unsigned k = g.k, D = g.D, ws = D + k + 2u, norig = g.norig;
	const unsigned nbw = (ws + 126u) >> 6;                                 // blocks a window can touch
	const unsigned lsh = nbw <= 4u ? 2u : nbw <= 8u ? 3u : nbw <= 16u ? 4u : 99u;
	if (lsh == 99u) return -1;
	const unsigned lpi = 1u << lsh, ipc = 64u >> lsh;                      // lanes per instance, instances per chunk
	const unsigned il = lane >> lsh, j = lane & (lpi - 1u);
	const unsigned nblk = (norig + 63u) >> 6;
	unsigned distinct = 0;
	bool viol = false;
	for (int pass = 1; pass <= 2; pass++) {
		unsigned ecmask = 0;
		if (pass == 2) {
			// first recurrence of the own id in every window = the nearest instance of the same list ahead (consecutive slots)
			for (unsigned i = lane; i < n; i += 64) {
				const unsigned a = s_sel[i], d = s_dir[i];
				unsigned own = ~0u;
				for (unsigned x = 0; x < n; x++) {
					const unsigned ax = s_sel[x], dx = s_dir[x];
					const unsigned delta = d ? a - ax : ax - a;
					if (dx == d && x != i && (d ? ax < a : ax > a) && delta < own) own = delta;
				}
				s_own[i] = own;
			}
			for (unsigned i = lane; i < (1u << vt.bits); i += 64) { vt.key[i] = BT_NONE; vt.mask[i] = 0; }
			WSYNC();
		}
		for (unsigned i0 = 0; i0 < n; i0 += ipc) {
			const unsigned i = i0 + il;
			const bool act = i < n;
			const unsigned a = act ? s_sel[i] : 0u, dir = act ? s_dir[i] : 0u;
			const bool fresh = act && a >= norig;                          // an instance on an inserted element: not indexed, walked below
			const unsigned ablk = a >> 6;
			const bool inr = act && !fresh && j < nbw && (dir ? j <= ablk : ablk + j < nblk);
			const unsigned bi = inr ? (dir ? ablk - j : ablk + j) : 0u;
			const ulonglong2 *rp = reinterpret_cast<const ulonglong2 *>(g.bidx + (size_t)bi * BT_IDX_WORDS);
			const ulonglong2 r0 = rp[0], r1 = rp[1];                       // marks of both strands; separators, (not pristine, write stamp)
			const unsigned cslot = fresh ? 0u : dir ? (a >= k ? a - k : a) : (a + k < norig ? a + k : a);
			const unsigned craw = g.ch[cslot];
			const unsigned long long mk = dir ? __brevll(r0.y) : r0.x, sp = dir ? __brevll(r1.x) : r1.x;      // step order: bit r = step t0 + r
			const int t0 = dir ? (int)a - (int)(bi * 64u + 63u) : (int)(bi * 64u) - (int)a;
			const unsigned long long vm = inr ? idx_bits(-t0, (int)ws - t0) : 0ull;
			const unsigned long long sepm = sp & vm;
			unsigned fs = sepm ? (unsigned)(t0 + (int)__builtin_ctzll(sepm)) : ~0u;
			for (unsigned d = 1; d < lpi; d <<= 1) { const unsigned v = __shfl_xor(fs, d); fs = v < fs ? v : fs; }
			unsigned len = fs < ws ? fs : ws;
			const unsigned reach = pass == 1 ? (fs < k ? fs : k) : (fs < ws - 1u ? fs : ws - 1u);      // last step whose block matters
			const bool touched = inr && t0 <= (int)reach && t0 + 63 >= 0;
			bool slow = fresh || (touched && (unsigned)(r1.y >> 32) != 0u);                            // a block that is no longer pristine
			for (unsigned d = 1; d < lpi; d <<= 1) slow |= __shfl_xor((int)slow, d) != 0;
			// a block written by a higher id: the exact order check of the elements READ in it (steps before the separator, up to k in pass 1)
			unsigned long long hot = __ballot(touched && !slow && (unsigned)r1.y > tid);
			for (; hot; hot &= hot - 1ull) {
				const unsigned src = (unsigned)__builtin_ctzll(hot);
				const unsigned hb = __shfl(bi, src), ha = __shfl(a, src), hd = __shfl(dir, src), hf = __shfl(fs, src);
				const unsigned slot = hb * 64u + lane;
				const int step = hd ? (int)ha - (int)slot : (int)slot - (int)ha;
				const unsigned lastread = pass == 1 ? (hf < k + 1u ? hf : k + 1u) : (hf < ws ? hf : ws);
				const unsigned wm = g.wmax[slot < norig ? slot : ha];
				if (__any(step >= 0 && (unsigned)step < lastread && slot < norig && wm > tid)) viol = true;
			}
			char ec = dir ? bt_comp((char)craw) : (char)craw;
			unsigned bit = len >= k + 1u ? (ec == 'A' ? 1u : ec == 'C' ? 2u : ec == 'G' ? 4u : 8u) : 0u;
			const unsigned long long todo = __ballot(act && slow && j == 0u);      // windows to walk
			if (pass == 1) {
				if (act && !slow) ecmask |= bit;
				for (unsigned long long td = todo; td; td &= td - 1ull) {
					const unsigned src = (unsigned)__builtin_ctzll(td);
					const unsigned wa = __shfl(a, src), wd = __shfl(dir, src);
					const WalkedWindow ww = idx_walk_window(g, wa, wd, k + 1u, lane, id, tid, nullptr, nullptr, 0u);
					if (ww.viol) viol = true;
					if (ww.len >= k + 1u) { const char e2 = wd ? bt_comp((char)ww.craw) : (char)ww.craw; ecmask |= e2 == 'A' ? 1u : e2 == 'C' ? 2u : e2 == 'G' ? 4u : 8u; }
				}
				continue;
			}
			// ---- pass 2: the marked steps 1 .. min(D, len, own recurrence) - 1 of the windows the index serves
			const unsigned own = act ? s_own[i] : 0u;
			const unsigned lim = len < D ? len : D, upper = own < lim ? own : lim;
			unsigned long long cm = act && !slow && bit ? mk & vm & idx_bits(1 - t0, (int)upper - t0) : 0ull;
			unsigned total = (unsigned)__popcll(cm);
			for (int d = 32; d > 0; d >>= 1) total += __shfl_xor(total, d);
			if (distinct + total > VT_FILL) return -1;                       // the table could fill up
			const unsigned *__restrict__ marks = g.bif[dir];
			while (__any(cm != 0ull)) {
				unsigned sl[4]; bool has[4]; unsigned bb[4];
#pragma unroll
				for (int q = 0; q < 4; q++) {
					has[q] = cm != 0ull;
					const unsigned r = has[q] ? (unsigned)__builtin_ctzll(cm) : 0u;
					if (has[q]) cm &= cm - 1ull;
					const unsigned step = (unsigned)(t0 + (int)r);
					sl[q] = has[q] ? (dir ? a - step : a + step) : a;
				}
#pragma unroll
				for (int q = 0; q < 4; q++) bb[q] = marks[fresh ? 0u : sl[q]];
				bool found = false;
#pragma unroll
				for (int q = 0; q < 4; q++) found |= vt_insert(vt, has[q] ? bb[q] : BT_NONE, bit, distinct);
				if (found) { if (viol) wave_stamp(g, 0, tid, 3, id, 0, tid + 1); return 1; }
			}
			// ---- ... and the walked ones
			for (unsigned long long td = todo; td; td &= td - 1ull) {
				const unsigned src = (unsigned)__builtin_ctzll(td);
				const unsigned wa = __shfl(a, src), wd = __shfl(dir, src);
				WSYNC();
				const WalkedWindow ww = idx_walk_window(g, wa, wd, ws, lane, id, tid, mk_step, mk_id, walk_marks);
				if (ww.viol) viol = true;
				if (ww.overflow) return -1;
				WSYNC();
				if (ww.len < k + 1u) continue;                              // endChar ' ': the instance takes no part
				const char e2 = wd ? bt_comp((char)ww.craw) : (char)ww.craw;
				const unsigned b2 = e2 == 'A' ? 1u : e2 == 'C' ? 2u : e2 == 'G' ? 4u : 8u;
				const unsigned lim2 = ww.len < D ? ww.len : D;
				if (distinct + ww.nm > VT_FILL) return -1;
				bool found = false;
				for (unsigned m0 = 0; m0 < ww.nm; m0 += 64) {
					const unsigned m = m0 + lane;
					const bool ok = m < ww.nm && mk_step[m] < lim2;
					found |= vt_insert(vt, ok ? mk_id[m] : BT_NONE, b2, distinct);
				}
				if (found) { if (viol) wave_stamp(g, 0, tid, 3, id, 0, tid + 1); return 1; }
			}
		}
		if (pass == 1) {
			for (int d = 32; d > 0; d >>= 1) ecmask |= __shfl_xor(ecmask, d);
			if (__popc(ecmask) <= 1) { if (viol) wave_stamp(g, 0, tid, 3, id, 0, tid + 1); return 0; }      // every instance continues with the same character: clean
		}
	}
	if (viol) wave_stamp(g, 0, tid, 3, id, 0, tid + 1);
	return 0;
}
// Dynamic LDS (what a probing workgroup holds decides how many are resident, and the kernel is sensitive to that: + 4 KB = + 18 %): the
// verdict table (2 x (1 << vbits) words), the instances (2 x max_inst words + max_inst bytes), the marks of a walked window (2 x walk_marks words).
// instbuf: per window entry `istride` words -- the number of instances and (element << 1) | strand of each, for the entries found live
// (BT_NONE in the first word otherwise): the reservation of the round starts from it instead of following the lists again.
// snapshot != 0: the entries are the touched ids of an incremental snapshot (DeviceBackend::snapshot_idx) -- same verdict, but the write
// stamps on the device are the PREVIOUS iteration's (they are reset after the snapshot): no order check, nothing counts as "above".
__global__ void __launch_bounds__(64) k_probe_idx(GraphView g, unsigned nwin, uint8_t *live, unsigned w0, unsigned vbits, unsigned max_inst, unsigned walk_marks,
                                                  unsigned *__restrict__ instbuf, unsigned istride, int snapshot)
{
	extern __shared__ unsigned pidx_dyn[];
	VtRef vt; vt.key = pidx_dyn; vt.mask = pidx_dyn + (1u << vbits); vt.bits = vbits;
	unsigned *const s_sel = vt.mask + (1u << vbits), *const s_own = s_sel + max_inst, *const s_mkstep = s_own + max_inst, *const s_mkid = s_mkstep + walk_marks;
	uint8_t *const s_dir = reinterpret_cast<uint8_t *>(s_mkid + walk_marks);
	const unsigned wi = blockIdx.x + w0, lane = threadIdx.x;
	if (!snapshot) round_stamp(g, 0);
	if (wi >= nwin) return;
	const unsigned id = g.win[wi], tid = snapshot ? 0xFFFFFFFEu : id + 1;
	if (!snapshot && g.need[id] == 2) { if (lane == 0) { live[wi] = 1; if (instbuf) instbuf[(size_t)wi * istride] = BT_NONE; if (g.test_flags & 32u) atomicAdd(&g_idx_stats[0], 1u); } return; }          // found live by an earlier probe and not touched since (a push resets it to 1)
	const unsigned n = wave_list_nodes(g, g.head[0][id], g.head[1][id], lane, nullptr, [&](unsigned off, unsigned, unsigned s, unsigned el, unsigned) {
		if (off < max_inst) { s_sel[off] = el; s_dir[off] = (uint8_t)s; }
	});
	int r = 0;
	if (n >= 2) {
		if (n > max_inst || n != g.lsize[0][id] + g.lsize[1][id]) r = -1;      // (lists are clean between rounds: live nodes = list sizes; anything else is the walking path's to report)
		else { WSYNC(); r = probe_idx(g, vt, s_sel, s_dir, s_own, s_mkstep, s_mkid, walk_marks, n, lane, id, tid); }
	}
	if (lane == 0) {
		if (r < 0) live[wi] = PROBE_UNSERVED;
		else if (r == 0) { g.need[id] = 0; g.touch[id] = 0; live[wi] = 0; }      // verdict taken now: clean until somebody touches it again
		else { g.need[id] = 2; live[wi] = 1; }
		if (g.test_flags & 32u) atomicAdd(&g_idx_stats[n < 2 ? 1 : r == 0 ? 2 : r == 1 ? 3 : r == -1 ? 4 : 5], 1u);
	}
	if (instbuf) {                                                       // the instances of a live entry for the reservation
		unsigned *ib = instbuf + (size_t)wi * istride;
		const bool give = r == 1 && n + 1u <= istride;
		if (give) { WSYNC(); for (unsigned i = lane; i < n; i += 64) ib[1 + i] = (s_sel[i] << 1) | s_dir[i]; }
		if (lane == 0) ib[0] = give ? n : BT_NONE;
	}
}

static_assert(PROBE_WAVES == 1u, "k_probe synchronises its lanes with WSYNC(): one wave per workgroup");
__global__ void __launch_bounds__(64 * PROBE_WAVES) k_probe(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, uint8_t *live, unsigned w0, int snapshot = 0)
{
	__shared__ Txn t;
	__shared__ BulgeWork w;
	__shared__ VerdictTable vt;
	__shared__ int ok;
	__shared__ __attribute__((aligned(16))) uint8_t fast[2048];
	const unsigned wi = blockIdx.x + w0, lane = threadIdx.x & 63u;
	if (!g.idx_probe && !snapshot) round_stamp(g, 0);                  // (behind k_probe_idx the probe phase started with that kernel)
	if (wi >= nwin) return;
	if ((g.idx_probe || snapshot) && live[wi] != PROBE_UNSERVED) return;      // decided by k_probe_idx
	const unsigned id = g.win[wi], tid = snapshot ? 0xFFFFFFFEu : id + 1;
	if (!snapshot && g.need[id] == 2) { if (threadIdx.x == 0) live[wi] = 1; return; }     // found live by an earlier probe and not touched since (a push resets it to 1)
	if (threadIdx.x == 0) { t.init(g, id, wi, snapshot ? 0u : 3u, arena + (size_t)wi * arena_bytes, arena_bytes); t.ext_stamps = true; t.fscr = fast; t.fscr_cap = sizeof fast; }      // (snapshot: the stamps are the previous iteration's -- no order check)
	WSYNC();
	wave_setup(g, t, w, true, lane, ok);
	for (unsigned i = threadIdx.x; i < VT_SLOTS; i += 64 * PROBE_WAVES) { vt.key[i] = BT_NONE; vt.mask[i] = 0; }
	WSYNC();
	// the windows go straight into the verdict table, a batch at a time (probe_windows)
	int verdict = 0;
	if (ok && g.probe_pre && probe_endchars(g, w, lane, id, tid)) ok = 0;      // every instance continues with the same character: clean (verdict stays 0)
	if (ok) {
		verdict = probe_windows(g, w, vt, lane, id, tid);
		if (verdict < 0) {                                                // undecided by the table: every window is needed
			for (unsigned i = 0; i < w.n; i++) wave_scan_instance(g, w, i, lane, 0, tid, 3, id);
			WSYNC();
		}
	}
	if (lane == 0) {
		bool has = verdict > 0;
		if (verdict < 0) { bt_end_chars(t, w); has = bt_any_bulges(t, w, true); }
		if (t.err) has = true;                                        // undecidable here: the commit path sorts it out
		if (!has) { g.need[id] = 0; g.touch[id] = 0; }             // verdict taken now: clean until somebody touches it again (counted by the next selection, k_select_count)
		else if (!t.err) g.need[id] = 2;
		else if (snapshot) g.need[id] = 1;                        // (a snapshot starts from need = 0: an undecidable id must be pending)
		live[wi] = has ? (t.err ? 2 : 1) : 0;                      // (2: live because undecidable here -- need stays 1; k_apply_probe on the other GPUs)
	}
}
// ---- read-only phases split over the attached GPUs (SURVEY.md 8e "Simplification": all GPUs work on disjoint id ranges against the
// same snapshot; the commits stay replicated, so the state is identical everywhere and only VERDICTS travel).
// Snapshot: need[] of a slice of the positional order, packed / unpacked around the all-gather (1 B per id).
__global__ void __launch_bounds__(256) k_pack_need(const unsigned *__restrict__ perm, const uint8_t *__restrict__ need, unsigned lo, unsigned hi, uint8_t *__restrict__ buf)
{
	const unsigned j = lo + blockIdx.x * blockDim.x + threadIdx.x;
	if (j < hi) buf[j] = need[perm[j]];
}
__global__ void __launch_bounds__(256) k_unpack_need(const unsigned *__restrict__ perm, const uint8_t *__restrict__ buf, unsigned n, unsigned mylo, unsigned myhi, uint8_t *__restrict__ need)
{
	const unsigned j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n && (j < mylo || j >= myhi)) need[perm[j]] = buf[j];
}
// Probe: what k_probe did to need / touch for the entries the OTHER GPUs probed (live[] all-gathered, 1 B per window entry), and the
// lowest order violation any of them saw (trail: one word per rank)
__global__ void __launch_bounds__(256) k_apply_probe(GraphView g, unsigned nwin, const uint8_t *__restrict__ live, unsigned w0, unsigned w1, const unsigned *__restrict__ trail, unsigned nranks)
{
	const unsigned wi = blockIdx.x * blockDim.x + threadIdx.x;
	if (wi == 0) { unsigned v = BT_NONE; for (unsigned p = 0; p < nranks; p++) v = trail[p] < v ? trail[p] : v; if (v != BT_NONE) atomicMin(&g.ctr[CTR_VIOL], v); }
	if (wi >= nwin || (wi >= w0 && wi < w1)) return;
	const unsigned id = g.win[wi];
	const uint8_t l = live[wi];
	if (l == 0) { g.need[id] = 0; g.touch[id] = 0; }
	else if (l == 1 && g.need[id] != 2) g.need[id] = 2;
}
__global__ void k_probe_trail(const unsigned *__restrict__ ctr, unsigned *__restrict__ trail, unsigned rank) { trail[rank] = ctr[CTR_VIOL]; }

// The lowest pending ids in [lo, limit], ascending; a pending "big" id ends the window (and runs alone if it is the lowest).
// out: win[], ctr[CTR_NWIN], ctr[CTR_LO] (lowest pending id), ctr[CTR_PUSHED] (solo flag).
// Two launches over chunks of `chunk` ids (256 threads x chunk/256 flags, 8-byte loads of the need / big bytes):
//   k_select_count  pending ids per chunk, first pending id, first pending big id (atomicMin)
//   k_select_write  every chunk below the window limit places its ids after the chunks ahead of it; the last one finalises
// sel: [0] first pending big id  [1] first pending id  [2] pending ids below the big id  [3] ticket  [8 ...] per-chunk counts
#define SEL_THREADS 256
template <class F>
__device__ __forceinline__ void select_scan_flags(const GraphView &g, unsigned long long id0, unsigned per_thread, unsigned lo, unsigned limit, F f)
{
	// f(first id of the word, pending bytes (0x01 per pending id), big-and-pending bytes)
	for (unsigned q = 0; q < per_thread; q += 8) {
		const unsigned long long idq = id0 + q;
		if (idq > limit) break;
		unsigned long long nb = *reinterpret_cast<const unsigned long long *>(g.need + idq);
		unsigned long long bb = *reinterpret_cast<const unsigned long long *>(g.big + idq);
#pragma unroll
		for (int j = 0; j < 8; j++) if (idq + j < lo || idq + j > limit) nb &= ~(0xFFull << (8 * j));
		nb = (nb | (nb >> 1) | (nb >> 2) | (nb >> 3) | (nb >> 4) | (nb >> 5) | (nb >> 6) | (nb >> 7)) & 0x0101010101010101ull;
		bb = (bb | (bb >> 1) | (bb >> 2) | (bb >> 3) | (bb >> 4) | (bb >> 5) | (bb >> 6) | (bb >> 7)) & nb;
		f(idq, nb, bb);
	}
}
// (It also counts the entries the probe of the round before retired -- live == 0 -- for the host's bookkeeping: a slice of the window per
// workgroup, one atomic each; that used to be a launch of its own behind every probe.)
__global__ void __launch_bounds__(SEL_THREADS) k_select_count(GraphView g, unsigned *__restrict__ sel, unsigned lo, unsigned limit, unsigned chunk0, unsigned chunk,
                                                              const uint8_t *__restrict__ live, unsigned probed)
{
	__shared__ unsigned s_cnt, s_ret;
	round_stamp(g, 3);                                               // the selection behind a round: its start is the end of the round's last kernel
	if (threadIdx.x == 0) { s_cnt = 0; s_ret = 0; }
	__syncthreads();
	if (probed) {
		const unsigned per = (probed + gridDim.x - 1) / gridDim.x, from = blockIdx.x * per, to = from + per < probed ? from + per : probed;
		unsigned r = 0;
		for (unsigned i = from + threadIdx.x; i < to; i += SEL_THREADS) r += live[i] == 0;
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) r += __shfl_down(r, d);
		if ((threadIdx.x & 63) == 0 && r) atomicAdd(&s_ret, r);
	}
	const unsigned per = chunk / SEL_THREADS;
	const unsigned long long id0 = (unsigned long long)(chunk0 + blockIdx.x) * chunk + (unsigned long long)threadIdx.x * per;
	unsigned cnt = 0, firstp = SBL_NONE, firstb = SBL_NONE;
	select_scan_flags(g, id0, per, lo, limit, [&](unsigned long long idq, unsigned long long nb, unsigned long long bb) {
		cnt += __popcll(nb);
		if (nb && firstp == SBL_NONE) firstp = (unsigned)(idq + (__builtin_ctzll(nb) >> 3));
		if (bb && firstb == SBL_NONE) firstb = (unsigned)(idq + (__builtin_ctzll(bb) >> 3));
	});
	if (cnt) atomicAdd(&s_cnt, cnt);
	if (firstp != SBL_NONE) atomicMin(&sel[1], firstp);
	if (firstb != SBL_NONE) atomicMin(&sel[0], firstb);
	__syncthreads();
	if (threadIdx.x == 0) { sel[8 + blockIdx.x] = s_cnt; if (s_ret) atomicAdd(&g.ctr[CTR_COMMITTED], s_ret); }
}
// post / post_seq: the last chunk also POSTS the counter block to the host (mapped pinned memory, fine-grained: plain stores cross
// PCIe) followed by a sequence number the host polls -- the round's counters and the next window arrive without a device-to-host copy
// kernel and without a stream synchronisation (the copy kernel was ~6 us and the wake-up after it ~23 us of idle GPU per round).
__global__ void __launch_bounds__(SEL_THREADS) k_select_write(GraphView g, unsigned *__restrict__ sel, unsigned *__restrict__ win, unsigned lo, unsigned limit, unsigned W,
                                                              unsigned chunk0, unsigned chunk, unsigned nchunks, volatile unsigned *post, unsigned post_seq)
{
	__shared__ unsigned s_wave[SEL_THREADS / 64], s_prefix, s_last;
	const unsigned bigid = sel[0], per = chunk / SEL_THREADS, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
	const unsigned long long cstart = (unsigned long long)(chunk0 + blockIdx.x) * chunk;
	unsigned total = 0;
	if (cstart < bigid) {                                            // (ids at or above the first pending big id are not selected)
		// ids selected by the chunks ahead of this one: all of them lie below the big id, their counts are exact
		unsigned pre = 0;
		for (unsigned j = threadIdx.x; j < blockIdx.x; j += SEL_THREADS) pre += sel[8 + j];
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) pre += __shfl_down(pre, d);
		if (lane == 0) s_wave[wv] = pre;
		__syncthreads();
		if (threadIdx.x == 0) { unsigned t = 0; for (unsigned w = 0; w < SEL_THREADS / 64; w++) t += s_wave[w]; s_prefix = t; }
		__syncthreads();
		const unsigned prefix = s_prefix;
		__syncthreads();
		const unsigned long long id0 = cstart + (unsigned long long)threadIdx.x * per;
		unsigned cnt = 0;
		select_scan_flags(g, id0, per, lo, limit, [&](unsigned long long idq, unsigned long long nb, unsigned long long) {
#pragma unroll
			for (int j = 0; j < 8; j++) if (idq + j >= bigid) nb &= ~(0xFFull << (8 * j));
			cnt += __popcll(nb);
		});
		unsigned incl = cnt;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) { unsigned v = __shfl_up(incl, d); if (lane >= (unsigned)d) incl += v; }
		if (lane == 63) s_wave[wv] = incl;
		__syncthreads();
		unsigned woff = 0;
		for (unsigned w = 0; w < SEL_THREADS / 64; w++) { if (w < wv) woff += s_wave[w]; total += s_wave[w]; }
		unsigned pos = prefix + woff + incl - cnt;
		if (prefix < W && cnt)
			select_scan_flags(g, id0, per, lo, limit, [&](unsigned long long idq, unsigned long long nb, unsigned long long) {
				while (nb) {
					unsigned j = __builtin_ctzll(nb) >> 3;
					if (idq + j < bigid) { if (pos < W) win[pos] = (unsigned)(idq + j); pos++; }
					nb &= nb - 1;
				}
			});
	}
	// the last chunk to finish publishes the result
	__syncthreads();
	if (threadIdx.x == 0) {
		if (total) atomicAdd(&sel[2], total);
		__threadfence();
		s_last = atomicAdd(&sel[3], 1u) == nchunks - 1;
	}
	__syncthreads();
	if (s_last && threadIdx.x == 0) {
		__threadfence();
		unsigned n = *(volatile unsigned *)&sel[2], solo = 0;
		if (n > W) n = W;
		if (n == 0 && bigid != SBL_NONE) { win[0] = bigid; n = 1; solo = 1; }
		const unsigned first = *(volatile unsigned *)&sel[1];
		g.ctr[CTR_NWIN] = n;
		g.ctr[CTR_LO] = first == SBL_NONE ? lo : first;
		g.ctr[CTR_PUSHED] = solo;
		sel[0] = SBL_NONE; sel[1] = SBL_NONE; sel[2] = 0; sel[3] = 0;      // ready for the next selection (stream order)
	}
	if (s_last && post) {
		__syncthreads();
		__threadfence();
		for (unsigned i = threadIdx.x; i < CTR_COUNT; i += SEL_THREADS) post[i] = __hip_atomic_load(&g.ctr[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		__threadfence_system();
		__syncthreads();
		if (threadIdx.x == 0) { post[CTR_COUNT] = post_seq; __threadfence_system(); }
	}
}

// ---- wave-cooperative neighbourhood scan -----------------------------------------------------------------
// The list is almost everywhere laid out consecutively (nx[e] == e + 1), so 64 lanes test 64 consecutive
// slots at once, keep the prefix whose links are intact, and only re-anchor at a real link break
// (an insertion or deletion made by an earlier collapse).  Visits exactly the elements bt_footprint visits.
#define CLAIM_CAP 4096u                      // ids a window entry can list; beyond that commit re-walks serially

#define RESUME_SLOTS 128u                   // instances whose core walk end is remembered for the ordering pass
// (the LDS set of the ids a workgroup has already claimed -- homologous instances repeat them -- has 1 << seen_bits slots: 1024 where ids
// have a handful of instances, 2048 where they have dozens; dynamic LDS, DeviceBackend::reserve)
struct ClaimList { unsigned *buf; unsigned *n; unsigned *seen; unsigned sbits; };      // n: LDS counter shared by the waves of the workgroup; seen: 1 << sbits slots

// Visits the elements first, next(first), ... (at most maxcount, stopping before a separator) with 64 lanes and
// calls f(b0, b1) on EVERY lane for each step of 64 (marks of both strands, BT_NONE for idle lanes) so that f may ballot.
template <class F>
__device__ __forceinline__ unsigned wave_walk_marks(const GraphView &g, unsigned first, unsigned dir, unsigned maxcount, unsigned lane,
                                                   unsigned strands /* bit s: report marks of strand s */, F f, const SepBounds sb = SepBounds{BT_NONE, BT_NONE, false})
{
	unsigned cur = first, done = 0;
	while (done < maxcount && cur != BT_NONE) {
		bool inr = done + lane < maxcount && (dir ? lane <= cur : (unsigned long long)cur + lane < g.cap_e);
		unsigned c = dir ? cur - lane : cur + lane;
		// all loads of the step are issued together (speculatively for lanes past a link break): one memory round trip per 64 elements
		unsigned chv = inr && !sb.by ? g.ch[c] : 0u;
		unsigned b0 = inr && (strands & 1u) ? g.bif[0][c] : BT_NONE, b1 = inr && (strands & 2u) ? g.bif[1][c] : BT_NONE;
		unsigned lnk = inr ? (dir ? g.pv[c] : g.nx[c]) : BT_NONE;
		const unsigned lprev = __shfl_up(lnk, 1);                        // the previous element's link is what the lane before loaded
		unsigned plink = inr && lane ? lprev : c;
		unsigned long long ml = __ballot(inr && plink == c);
		unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);      // intact prefix, >= 1
		unsigned long long ms = __ballot(lane < pre && (sb.by ? (c == sb.lo || c == sb.hi) : chv == BT_SEP));
		unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;                // first separator inside the prefix
		bool proc = lane < pre && lane < stop;
		f(proc ? b0 : BT_NONE, proc ? b1 : BT_NONE);
		if (stop < pre) return BT_NONE;
		cur = __shfl(lnk, pre - 1);
		done += pre;
	}
	return cur;                                                                  // the element after the last one visited (BT_NONE: end of chromosome)
}

// Two independent walks advancing together (one memory round trip serves both): same visiting rules as wave_walk_marks.
template <class F>
__device__ __forceinline__ void wave_walk_marks2(const GraphView &g, unsigned ca, unsigned dira, unsigned na, unsigned sa,
                                                 unsigned cb, unsigned dirb, unsigned nb, unsigned sb, unsigned lane, F f,
                                                 const SepBounds sp = SepBounds{BT_NONE, BT_NONE, false})
{
	unsigned da = 0, db = 0;
	while ((da < na && ca != BT_NONE) || (db < nb && cb != BT_NONE)) {
		const bool aa = da < na && ca != BT_NONE, ab = db < nb && cb != BT_NONE;
		const bool ina = aa && da + lane < na && (dira ? lane <= ca : (unsigned long long)ca + lane < g.cap_e);
		const bool inb = ab && db + lane < nb && (dirb ? lane <= cb : (unsigned long long)cb + lane < g.cap_e);
		const unsigned xa = dira ? ca - lane : ca + lane, xb = dirb ? cb - lane : cb + lane;
		const unsigned cha = ina && !sp.by ? g.ch[xa] : 0u, chb = inb && !sp.by ? g.ch[xb] : 0u;
		const unsigned a0 = ina && (sa & 1u) ? g.bif[0][xa] : BT_NONE, a1 = ina && (sa & 2u) ? g.bif[1][xa] : BT_NONE;
		const unsigned b0 = inb && (sb & 1u) ? g.bif[0][xb] : BT_NONE, b1 = inb && (sb & 2u) ? g.bif[1][xb] : BT_NONE;
		const unsigned lka = ina ? (dira ? g.pv[xa] : g.nx[xa]) : BT_NONE, lkb = inb ? (dirb ? g.pv[xb] : g.nx[xb]) : BT_NONE;
		const unsigned lpa = __shfl_up(lka, 1), lpb = __shfl_up(lkb, 1);      // previous links: what the lanes before loaded
		const unsigned pla = ina && lane ? lpa : xa, plb = inb && lane ? lpb : xb;
		const bool sepa = sp.by ? (xa == sp.lo || xa == sp.hi) : cha == BT_SEP, sepb = sp.by ? (xb == sp.lo || xb == sp.hi) : chb == BT_SEP;
		if (aa) {
			unsigned long long ml = __ballot(ina && pla == xa);
			unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);
			unsigned long long ms = __ballot(lane < pre && sepa);
			unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
			bool proc = lane < pre && lane < stop;
			f(proc ? a0 : BT_NONE, proc ? a1 : BT_NONE);
			if (stop < pre || pre == 0) ca = BT_NONE; else { ca = __shfl(lka, pre - 1); da += pre; }
		}
		if (ab) {
			unsigned long long ml = __ballot(inb && plb == xb);
			unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);
			unsigned long long ms = __ballot(lane < pre && sepb);
			unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
			bool proc = lane < pre && lane < stop;
			f(proc ? b0 : BT_NONE, proc ? b1 : BT_NONE);
			if (stop < pre || pre == 0) cb = BT_NONE; else { cb = __shfl(lkb, pre - 1); db += pre; }
		}
	}
}

// The seen-set remembers the KIND of a claim (bit 31: exclusive): an exclusive claim that finds the id claimed for ordering only upgrades the
// entry and is made all the same (atomicMin + list entry; the ordering entry stays in the list beside it, harmless: an id the runner owns is
// never "order-blocked"), so exclusive and ordering claims may come in any order -- the waves of a workgroup need no barrier between them.
#define SEEN_EXCL 0x80000000u
__device__ __forceinline__ void wave_claim(const GraphView &g, ClaimList &cl, unsigned st, unsigned b, unsigned lane)
{
	bool has = b != BT_NONE;
	if (has) {                                                   // claim every id once per workgroup
		unsigned h = (b * 2654435761u) >> (32u - cl.sbits);
		has = false;
		for (int probe = 0; probe < 8; probe++) {
			unsigned old = atomicCAS(&cl.seen[h], BT_NONE, b | SEEN_EXCL);
			if (old == BT_NONE) { has = true; break; }
			if (old == (b | SEEN_EXCL)) break;
			if (old == b) { has = atomicCAS(&cl.seen[h], b, b | SEEN_EXCL) == b; break; }      // claimed for ordering so far: upgrade (once)
			h = (h + 1) & ((1u << cl.sbits) - 1u);
			if (probe == 7) has = true;                          // crowded table: claim again, harmless
		}
	}
	if (has) atomicMin(&g.own[b], st);
	unsigned long long m = __ballot(has);
	if (!m) return;
	unsigned base = 0;
	if (lane == (unsigned)__builtin_ctzll(m)) base = atomicAdd(cl.n, (unsigned)__popcll(m));
	unsigned off = __shfl(base, (unsigned)__builtin_ctzll(m)) + __popcll(m & ((1ull << lane) - 1ull));
	if (has && off < CLAIM_CAP) cl.buf[1 + off] = b;
}

// ordering claim (bt_footprint kind 1): ids above the runner are stamped without being listed, ids below it are listed
// (flag bit 31) so that the commit check can see whether anything at or below them is about to run
__device__ __forceinline__ void wave_claim_order(const GraphView &g, ClaimList &cl, unsigned st, unsigned id, unsigned b, unsigned lane)
{
	bool has = b != BT_NONE && b != id;
	if (has) {
		unsigned h = (b * 2654435761u) >> (32u - cl.sbits);
		has = false;
		for (int probe = 0; probe < 8; probe++) {
			unsigned old = atomicCAS(&cl.seen[h], BT_NONE, b);
			if (old == BT_NONE) { has = true; break; }
			if ((old & ~SEEN_EXCL) == b) break;                  // claimed already, either way
			h = (h + 1) & ((1u << cl.sbits) - 1u);
			if (probe == 7) has = true;
		}
	}
	if (has && b > id) { atomicMin(&g.own[b], st); has = false; }
	unsigned long long m = __ballot(has);
	if (!m) return;
	unsigned base = 0;
	if (lane == (unsigned)__builtin_ctzll(m)) base = atomicAdd(cl.n, (unsigned)__popcll(m));
	unsigned off = __shfl(base, (unsigned)__builtin_ctzll(m)) + __popcll(m & ((1ull << lane) - 1ull));
	if (has && off < CLAIM_CAP) cl.buf[1 + off] = b | 0x80000000u;
}

__device__ __forceinline__ unsigned wave_walk_claim(const GraphView &g, unsigned first, unsigned dir, unsigned maxcount, unsigned lane,
                                                    unsigned strands, ClaimList &cl, unsigned st, const SepBounds sb = SepBounds{BT_NONE, BT_NONE, false})
{
	return wave_walk_marks(g, first, dir, maxcount, lane, strands, [&](unsigned b0, unsigned b1) { wave_claim(g, cl, st, b0, lane); wave_claim(g, cl, st, b1, lane); }, sb);
}

// ---- the reservation walks of an instance, all at once ---------------------------------------------------------------------------
// wave_walk_marks advances 64 elements per memory round trip because the next 64 are only known once the links of these have arrived:
// the neighbourhood of an instance (core D + 2k + 3, then 2(D + k + 2) + k + 1 ahead and D + k + 2 behind) was 7 dependent round trips,
// 28 for the four instances a reservation wave handles, and k_reserve is exactly that chain.  The list is laid out consecutively
// almost everywhere, so the loads of ALL chunks of a walk are issued together for the slots the walk would visit if it is, the links
// that came back are checked against that assumption, and only a walk that meets a link break (or could meet a separator: its span is
// compared with the two separators of its chromosome beforehand, SepBounds) is done again by the step-wise walk.  Same elements
// visited, same claims made.
// all of e, e +- 1, ... (n elements in direction dir) lie strictly between the separators of e's chromosome
__device__ __forceinline__ bool span_inside(const SepBounds &sp, unsigned e, unsigned dir, unsigned n)
{
	if (!sp.by || e == BT_NONE || n == 0) return false;
	if (!dir) return e > sp.lo && (unsigned long long)e + n - 1 < sp.hi;
	return e < sp.hi && (unsigned long long)e > (unsigned long long)sp.lo + n - 1;
}
#define RSV_CORE_CHUNKS 4
#define RSV_FLANK_CHUNKS 3
// core walk with exclusive claims on the marks of both strands; returns the element after the last one visited.  false: not done (step-wise walk needed)
__device__ __forceinline__ bool wave_core_claim_burst(const GraphView &g, unsigned e0, unsigned dir, unsigned core, unsigned lane, ClaimList &cl, unsigned st,
                                                     const SepBounds &sp, unsigned &nxt)
{
	if (core > 64u * RSV_CORE_CHUNKS || !span_inside(sp, e0, dir, core)) return false;
	const unsigned *__restrict__ link = dir ? g.pv : g.nx;
	unsigned b0[RSV_CORE_CHUNKS], b1[RSV_CORE_CHUNKS], lk[RSV_CORE_CHUNKS];
#pragma unroll
	for (int u = 0; u < RSV_CORE_CHUNKS; u++) {
		const unsigned off = lane + 64u * u, x = off < core ? (dir ? e0 - off : e0 + off) : e0;
		b0[u] = g.bif[0][x]; b1[u] = g.bif[1][x]; lk[u] = link[x];
	}
	bool good = true;
#pragma unroll
	for (int u = 0; u < RSV_CORE_CHUNKS; u++) {
		const unsigned off = lane + 64u * u, x = dir ? e0 - off : e0 + off;
		if (off + 1 < core) good = good && lk[u] == (dir ? x - 1 : x + 1);
	}
	if (__ballot(!good)) return false;
#pragma unroll
	for (int u = 0; u < RSV_CORE_CHUNKS; u++) {
		if (64u * u >= core) break;
		const bool in = lane + 64u * u < core;
		wave_claim(g, cl, st, in ? b0[u] : BT_NONE, lane);
		wave_claim(g, cl, st, in ? b1[u] : BT_NONE, lane);
	}
	unsigned last = 0;
#pragma unroll
	for (int u = 0; u < RSV_CORE_CHUNKS; u++) if ((core - 1) >> 6 == (unsigned)u) last = __shfl(lk[u], (core - 1) & 63u);
	nxt = last;
	return true;
}
// the two ordering walks of an instance (wave_walk_marks2 of k_reserve): ahead from nxt on the opposite strand's marks, behind from the
// element before e0 on the own strand's marks.  false: not done
template <class Order>
__device__ __forceinline__ bool wave_flank_order_burst(const GraphView &g, unsigned e0, unsigned s, unsigned nxt, unsigned na, unsigned nb, unsigned lane,
                                                      const SepBounds &sp, Order order)
{
	const unsigned da = s, db = s ^ 1u;
	const unsigned bfirst = db ? e0 - 1 : e0 + 1;                          // the element before e0 in its walking direction, if the layout is consecutive there
	if (na > 64u * RSV_FLANK_CHUNKS || nb > 64u * RSV_FLANK_CHUNKS || e0 == 0) return false;
	if ((na && !span_inside(sp, nxt, da, na)) || !span_inside(sp, bfirst, db, nb)) return false;
	const unsigned *__restrict__ linka = da ? g.pv : g.nx, *__restrict__ linkb = db ? g.pv : g.nx;
	unsigned ma[RSV_FLANK_CHUNKS], la[RSV_FLANK_CHUNKS], mb[RSV_FLANK_CHUNKS], lb[RSV_FLANK_CHUNKS];
	const unsigned l0 = linkb[e0];
#pragma unroll
	for (int u = 0; u < RSV_FLANK_CHUNKS; u++) {
		const unsigned off = lane + 64u * u;
		const unsigned xa = na && off < na ? (da ? nxt - off : nxt + off) : e0, xb = off < nb ? (db ? bfirst - off : bfirst + off) : e0;
		ma[u] = g.bif[s ^ 1u][xa]; la[u] = linka[xa];
		mb[u] = g.bif[s][xb]; lb[u] = linkb[xb];
	}
	bool good = l0 == bfirst;
#pragma unroll
	for (int u = 0; u < RSV_FLANK_CHUNKS; u++) {
		const unsigned off = lane + 64u * u;
		const unsigned xa = da ? nxt - off : nxt + off, xb = db ? bfirst - off : bfirst + off;
		if (off + 1 < na) good = good && la[u] == (da ? xa - 1 : xa + 1);
		if (off + 1 < nb) good = good && lb[u] == (db ? xb - 1 : xb + 1);
	}
	if (__ballot(!good)) return false;
#pragma unroll
	for (int u = 0; u < RSV_FLANK_CHUNKS; u++) {
		const unsigned off = lane + 64u * u;
		if (64u * u < na) order(off < na ? ma[u] : BT_NONE);
		if (64u * u < nb) order(off < nb ? mb[u] : BT_NONE);
	}
	return true;
}

// After a collapse: publish the writes of the transaction (everything from the target instance to the end of its
// look-forward flank had marks, characters, positions or links rewritten), check that no higher id read or wrote them, and
// make every id whose window can see the region and that is still ahead in the order pending (bt_push_neighbourhood with
// 64 lanes).  Only instances walking TOWARDS the region can see it: upstream that is the target's own strand, beyond the
// end of the region the opposite strand, inside it both.  The region is walked once (write stamps on its first
// newlen + 2k elements, pushes on newlen + 2k + 1), then the upstream and the downstream walk advance together.
__device__ __forceinline__ void wave_publish_collapse(const GraphView &g, unsigned id, unsigned e, unsigned d, unsigned newlen, unsigned lane, const unsigned *sepl = nullptr)
{
	const SepBounds sp = sep_bounds(g, sepl, e, lane);                      // the region and both walks stay in the chromosome of e
	const unsigned reach = g.D + g.k + 2, tid = id + 1, nstamp = newlen + 2 * g.k, nreg = nstamp + 1;
	auto push1 = [&](unsigned b) { if (b != BT_NONE && b < g.nid) { g.touch[b] = 1; if (b > id) g.need[b] = 1; } };
	// ---- everything at once where the list is laid out consecutively (a collapse that replaced a branch by one of the same length: the
	// usual SNP bulge): the region, the upstream and the downstream walk were up to seven dependent memory round trips of every collapse;
	// the loads of all their chunks are issued together for the slots the walks would visit, and the links that come back say whether
	// they did (see wave_core_claim_burst).  Anything else -- inserted elements, a separator in reach -- takes the step-wise walks below.
	{
		enum { RC = 4, FC = 3 };
		const unsigned ub = d ? e + 1 : e - 1;                              // the element before e in its walking direction, if consecutive
		if (!(g.test_flags & 8u) && nreg <= 64u * RC && reach <= 64u * FC && e != 0 && span_inside(sp, e, d, nreg + reach) && span_inside(sp, ub, d ^ 1u, reach)) {
			const unsigned *__restrict__ lf = d ? g.pv : g.nx, *__restrict__ lb = d ? g.nx : g.pv;
			unsigned r0[RC], r1[RC], rl[RC], um[FC], ul[FC], dm[FC], dl[FC];
			const unsigned l0 = lb[e], dfirst = d ? e - nreg : e + nreg;
#pragma unroll
			for (int u = 0; u < RC; u++) {
				const unsigned off = lane + 64u * u, x = off < nreg ? (d ? e - off : e + off) : e;
				r0[u] = g.bif[0][x]; r1[u] = g.bif[1][x]; rl[u] = lf[x];
			}
#pragma unroll
			for (int u = 0; u < FC; u++) {
				const unsigned off = lane + 64u * u;
				const unsigned xu = off < reach ? (d ? ub + off : ub - off) : e, xd = off < reach ? (d ? dfirst - off : dfirst + off) : e;
				um[u] = g.bif[d][xu]; ul[u] = lb[xu];
				dm[u] = g.bif[d ^ 1u][xd]; dl[u] = lf[xd];
			}
			bool good = l0 == ub;
#pragma unroll
			for (int u = 0; u < RC; u++) {
				const unsigned off = lane + 64u * u, x = d ? e - off : e + off;
				if (off < nreg) good = good && rl[u] == (d ? x - 1 : x + 1);          // (the last one leads to the first element downstream)
			}
#pragma unroll
			for (int u = 0; u < FC; u++) {
				const unsigned off = lane + 64u * u;
				const unsigned xu = d ? ub + off : ub - off, xd = d ? dfirst - off : dfirst + off;
				if (off + 1 < reach) good = good && ul[u] == (d ? xu + 1 : xu - 1) && dl[u] == (d ? xd - 1 : xd + 1);
			}
			if (!__ballot(!good)) {
#pragma unroll
				for (int u = 0; u < RC; u++) {
					const unsigned off = lane + 64u * u, c = d ? e - off : e + off;
					if (off < nreg) {
						push1(r0[u]); push1(r1[u]);
						if (off < nstamp) {
							unsigned a = atomicMax(&g.wmax[c], tid);
							if (off == 0 || (c & 63u) == (d ? 63u : 0u)) bt_idx_wstamp(g, c, tid);      // (consecutive slots: one lane per 64-slot block)
							unsigned rm = g.rmax[c];
							if (a > tid || rm > tid) {
								atomicMin(&g.ctr[CTR_VIOL], id);
								if (atomicCAS(&g.ctr[CTR_DETAIL], 0u, 4u) == 0u) { g.ctr[CTR_DETAIL + 1] = c; g.ctr[CTR_DETAIL + 2] = (a > rm ? a : rm) - 1; g.ctr[CTR_DETAIL + 3] = id; g.ctr[CTR_DETAIL + 4] = (a > tid ? 1u : 0u) | (rm > tid ? 2u : 0u); }
							}
						}
					}
				}
#pragma unroll
				for (int u = 0; u < FC; u++) {
					const unsigned off = lane + 64u * u;
					if (off < reach) { push1(um[u]); push1(dm[u]); }
				}
				return;
			}
		}
	}
	// ---- the region
	unsigned cur = e, done = 0;
	bool open = true;
	while (done < nreg && cur != BT_NONE) {
		bool inr = done + lane < nreg && (d ? lane <= cur : (unsigned long long)cur + lane < g.cap_e);
		unsigned c = d ? cur - lane : cur + lane;
		unsigned chv = inr && !sp.by ? g.ch[c] : 0u;
		unsigned b0 = inr ? g.bif[0][c] : BT_NONE, b1 = inr ? g.bif[1][c] : BT_NONE;
		unsigned lnk = inr ? (d ? g.pv[c] : g.nx[c]) : BT_NONE;
		const unsigned lprev = __shfl_up(lnk, 1);
		unsigned plink = inr && lane ? lprev : c;
		unsigned long long ml = __ballot(inr && plink == c);
		unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);
		unsigned long long ms = __ballot(lane < pre && (sp.by ? (c == sp.lo || c == sp.hi) : chv == BT_SEP));
		unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
		if (lane < pre && lane < stop) {
			push1(b0); push1(b1);
			if (done + lane < nstamp) {
				unsigned a = atomicMax(&g.wmax[c], tid);
				bt_idx_wstamp(g, c, tid);
				unsigned rm = g.rmax[c];
				if (a > tid || rm > tid) {
					atomicMin(&g.ctr[CTR_VIOL], id);
					if (atomicCAS(&g.ctr[CTR_DETAIL], 0u, 4u) == 0u) { g.ctr[CTR_DETAIL + 1] = c; g.ctr[CTR_DETAIL + 2] = (a > rm ? a : rm) - 1; g.ctr[CTR_DETAIL + 3] = id; g.ctr[CTR_DETAIL + 4] = (a > tid ? 1u : 0u) | (rm > tid ? 2u : 0u); }
				}
			}
		}
		if (stop < pre) { open = false; break; }
		cur = __shfl(lnk, pre - 1);
		done += pre;
	}
	// ---- upstream (direction d ^ 1, marks of strand d) and downstream (direction d, marks of strand d ^ 1) together
	unsigned cu = d ? g.nx[e] : g.pv[e], du = 0;
	unsigned cd = open ? cur : BT_NONE, dd = 0;
	while ((du < reach && cu != BT_NONE) || (dd < reach && cd != BT_NONE)) {
		const bool au = du < reach && cu != BT_NONE, ad = dd < reach && cd != BT_NONE;
		const unsigned diru = d ^ 1u, dird = d;
		bool inu = au && du + lane < reach && (diru ? lane <= cu : (unsigned long long)cu + lane < g.cap_e);
		bool ind = ad && dd + lane < reach && (dird ? lane <= cd : (unsigned long long)cd + lane < g.cap_e);
		unsigned xu = diru ? cu - lane : cu + lane, xd = dird ? cd - lane : cd + lane;
		unsigned chu = inu && !sp.by ? g.ch[xu] : 0u, chd = ind && !sp.by ? g.ch[xd] : 0u;
		const bool sepu = sp.by ? (xu == sp.lo || xu == sp.hi) : chu == BT_SEP, sepd = sp.by ? (xd == sp.lo || xd == sp.hi) : chd == BT_SEP;
		unsigned bu = inu ? g.bif[d][xu] : BT_NONE, bd = ind ? g.bif[d ^ 1u][xd] : BT_NONE;
		unsigned lku = inu ? (diru ? g.pv[xu] : g.nx[xu]) : BT_NONE, lkd = ind ? (dird ? g.pv[xd] : g.nx[xd]) : BT_NONE;
		const unsigned lpu = __shfl_up(lku, 1), lpd = __shfl_up(lkd, 1);
		unsigned plu = inu && lane ? lpu : xu, pld = ind && lane ? lpd : xd;
		if (au) {
			unsigned long long ml = __ballot(inu && plu == xu);
			unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);
			unsigned long long ms = __ballot(lane < pre && sepu);
			unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
			if (lane < pre && lane < stop) push1(bu);
			if (stop < pre || pre == 0) cu = BT_NONE; else { cu = __shfl(lku, pre - 1); du += pre; }
		}
		if (ad) {
			unsigned long long ml = __ballot(ind && pld == xd);
			unsigned pre = ml == ~0ull ? 64u : (unsigned)__builtin_ctzll(~ml);
			unsigned long long ms = __ballot(lane < pre && sepd);
			unsigned stop = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
			if (lane < pre && lane < stop) push1(bd);
			if (stop < pre || pre == 0) cd = BT_NONE; else { cd = __shfl(lkd, pre - 1); dd += pre; }
		}
	}
}

// ---- the reservation walks of an instance from the BLOCK INDEX (round 5) ------------------------------------------------------------
// The neighbourhood of an instance -- core D + 2k + 3 ahead (both strands, exclusive), then up to 2(D + k + 2) + k ahead on the opposite
// strand and D + k + 2 behind on the own strand (ordering) -- is ten or eleven 64-slot blocks: sixteen lanes per instance load their
// block's record (marks of both strands, separators, "not pristine"), cut the ranges at the first separator, and only the MARKED slots'
// ids are gathered and claimed.  Same (id, kind) pairs as the walks (bt_footprint_idx is the one-thread form, checked against
// bt_footprint by tests/hostsim on every reservation).  Four instances per wave and pass; an instance whose neighbourhood touches a
// block that is no longer pristine, or that sits on an inserted element, takes the walks (returned as a bit per instance of the group).
struct RsvIdxLane { unsigned long long ex0, ex1, ord; int t0; unsigned a, s; bool ahead; };
// masks of the lane's block for the group of four instances i0 .. i0 + 3; returns a bit per instance of the group that must take the walks
__device__ __forceinline__ unsigned reserve_idx_masks(const GraphView &g, const unsigned *inst, unsigned ninst, unsigned i0, unsigned lane, unsigned NA, unsigned NB,
                                                      unsigned core, unsigned fwd, unsigned back, RsvIdxLane &L)
{
	const unsigned il = lane >> 4, j = lane & 15u, i = i0 + il, norig = g.norig, nblk = (norig + 63u) >> 6;
	const bool act = i < ninst;
	const unsigned packed = act ? inst[i] : 0u, a = packed >> 1, s = packed & 1u, ablk = a >> 6;
	const bool fresh = act && a >= norig;                                  // an instance on an inserted element: not indexed
	const bool ahead = j < NA, behind = !ahead && j < NA + NB;
	const unsigned jj = ahead ? j : j - NA;
	const bool rev = ahead ? s != 0u : s == 0u;                            // the walk of this lane goes towards lower slots
	const bool inr = act && !fresh && (ahead || behind) && (rev ? jj <= ablk : ablk + jj < nblk);
	const unsigned bi = inr ? (rev ? ablk - jj : ablk + jj) : 0u;
	const ulonglong2 *rp = reinterpret_cast<const ulonglong2 *>(g.bidx + (size_t)bi * BT_IDX_WORDS);
	const ulonglong2 r0 = rp[0], r1 = rp[1];
	const unsigned long long m0 = rev ? __brevll(r0.x) : r0.x, m1 = rev ? __brevll(r0.y) : r0.y, sp = rev ? __brevll(r1.x) : r1.x;
	const int t0 = rev ? (int)a - (int)(bi * 64u + 63u) : (int)(bi * 64u) - (int)a;      // step (from the instance, in this lane's direction) of bit 0
	const unsigned lo = ahead ? 0u : 1u, hi = ahead ? fwd + 1u : back + 1u;               // the steps lo .. hi - 1 belong to this walk
	const unsigned long long vm = inr ? idx_bits((int)lo - t0, (int)hi - t0) : 0ull;
	const unsigned long long sepm = sp & vm & idx_bits(1 - t0, 64);       // (step 0 is the instance itself, never a separator)
	const unsigned fsl = sepm ? (unsigned)(t0 + (int)__builtin_ctzll(sepm)) : ~0u;
	unsigned fa = ahead ? fsl : ~0u, fb = behind ? fsl : ~0u;
#pragma unroll
	for (int d = 1; d < 16; d <<= 1) { const unsigned va = __shfl_xor(fa, d), vb = __shfl_xor(fb, d); fa = va < fa ? va : fa; fb = vb < fb ? vb : fb; }
	const unsigned stop = ahead ? fa : fb;                                  // the walk ends BEFORE this step (first separator)
	const unsigned last = stop < hi ? stop : hi - 1u;                      // last step whose block matters (the separator's own block included)
	bool slow = fresh || (inr && (unsigned)(r1.y >> 32) != 0u && t0 <= (int)last && t0 + 63 >= (int)lo);
#pragma unroll
	for (int d = 1; d < 16; d <<= 1) slow |= __shfl_xor((int)slow, d) != 0;
	const unsigned end = stop < hi ? stop : hi;
	L.t0 = t0; L.a = a; L.s = s; L.ahead = ahead;
	L.ex0 = L.ex1 = L.ord = 0ull;
	if (inr && !slow) {
		if (ahead) {
			const unsigned cend = end < core ? end : core;
			const unsigned long long cm = idx_bits(-t0, (int)cend - t0), om = idx_bits((int)core - t0, (int)end - t0);
			L.ex0 = m0 & cm; L.ex1 = m1 & cm; L.ord = (s ? m0 : m1) & om;      // the core: both strands; beyond it: the opposite strand
		} else L.ord = (s ? m1 : m0) & idx_bits(1 - t0, (int)end - t0);      // behind: the own strand
	}
	const unsigned long long sb = __ballot(act && slow && j == 0u);
	return (unsigned)((sb & 1ull) | ((sb >> 15) & 2ull) | ((sb >> 30) & 4ull) | ((sb >> 45) & 8ull));
}
// The marked slots under the set bits of the lanes' masks, COMPACTED through a per-wave LDS list and gathered 64 at a time: the sixteen
// lanes of an instance hold their marks very unevenly (the core is four of eleven blocks), and a claim step (LDS set, atomicMin, list
// append: ~40 instructions) is the same price for one id as for 64 -- the reservation is issue-bound.  bits0 / bits1: marks of strand
// st0 / st1 in this lane's block; f(id) is called on EVERY lane, once per 64 ids.  false: more marks than the list holds (nothing done).
template <class F>
__device__ __forceinline__ bool reserve_idx_emit(const GraphView &g, const RsvIdxLane &L, unsigned long long bits0, unsigned st0, unsigned long long bits1, unsigned st1,
                                                 bool backward, unsigned *list, unsigned RSV_LIST /* entries of the list */, unsigned lane, F f)
{
	const unsigned cnt = (unsigned)__popcll(bits0) + (unsigned)__popcll(bits1);
	unsigned incl = cnt;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { const unsigned v = __shfl_up(incl, d); if (lane >= (unsigned)d) incl += v; }
	const unsigned total = __shfl(incl, 63);
	if (total > RSV_LIST) return false;
	if (!total) return true;
	const bool down = backward ? L.s == 0u : L.s != 0u;                    // slots decrease with the step
	unsigned o = incl - cnt;
	for (; bits0; bits0 &= bits0 - 1ull) { const unsigned step = (unsigned)(L.t0 + (int)__builtin_ctzll(bits0)); list[o++] = (down ? L.a - step : L.a + step) | (st0 << 31); }
	for (; bits1; bits1 &= bits1 - 1ull) { const unsigned step = (unsigned)(L.t0 + (int)__builtin_ctzll(bits1)); list[o++] = (down ? L.a - step : L.a + step) | (st1 << 31); }
	__builtin_amdgcn_wave_barrier();
	for (unsigned p0 = 0; p0 < total; p0 += 256u) {
		unsigned bb[4];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const unsigned p = p0 + 64u * q + lane;
			const unsigned e = list[p < total ? p : 0u];
			bb[q] = g.bif[e >> 31][e & 0x7FFFFFFFu];
			if (p >= total) bb[q] = BT_NONE;
		}
#pragma unroll
		for (int q = 0; q < 4; q++) if (p0 + 64u * q < total) f(bb[q]);
	}
	__builtin_amdgcn_wave_barrier();
	return true;
}
// (the uncompacted form, for a group with more marks than the list holds: four gathers in flight per lane)
template <class F>
__device__ __forceinline__ void reserve_idx_gather(const GraphView &g, const RsvIdxLane &L, unsigned long long bits, unsigned strand, bool backward, F f)
{
	const unsigned *__restrict__ marks = g.bif[strand];
	const bool down = backward ? L.s == 0u : L.s != 0u;                    // slots decrease with the step
	while (__any(bits != 0ull)) {
		unsigned sl[4]; bool has[4]; unsigned bb[4];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			has[q] = bits != 0ull;
			const unsigned r = has[q] ? (unsigned)__builtin_ctzll(bits) : 0u;
			if (has[q]) bits &= bits - 1ull;
			const unsigned step = (unsigned)(L.t0 + (int)r);
			sl[q] = has[q] ? (down ? L.a - step : L.a + step) : L.a;
		}
#pragma unroll
		for (int q = 0; q < 4; q++) bb[q] = marks[sl[q]];
#pragma unroll
		for (int q = 0; q < 4; q++) f(has[q] ? bb[q] : BT_NONE);
	}
}

// one wave per window entry: claim every id of the neighbourhood and remember the list for the commit check
// The instances of the id are dealt out to the waves of the workgroup (blockDim.x / 64 of them: two where ids have a handful of
// instances -- 8 strains: 85.0 ms per stage against 86.1 with four and 88.8 with eight -- four where they have dozens, DeviceBackend::rsv_waves).
#define RSV_WAVES_MAX 4u
__global__ void __launch_bounds__(64 * RSV_WAVES_MAX) k_reserve(GraphView g, unsigned nwin, unsigned *claims, const uint8_t *live, unsigned seen_bits, unsigned list_cap,
                                                                 const unsigned *__restrict__ instbuf, unsigned istride)
{
	const unsigned RSV_WAVES = blockDim.x >> 6;
	const unsigned w = blockIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
	round_stamp(g, 1);
	if (w >= nwin || !live[w]) return;
	const bool rprof = (g.test_flags & 32u) != 0u;
	unsigned long long rt = rprof ? wall_clock64() : 0ull;
#define RSV_T(i) do { if (rprof && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_rsv_ticks[i], n_ - rt); rt = n_; } } while (0)
	extern __shared__ unsigned rsv_dyn[];                             // the seen-set (1 << seen_bits words), then one list of list_cap words per wave
	unsigned *const seen = rsv_dyn;
	__shared__ unsigned resume[RESUME_SLOTS], inst[RESUME_SLOTS];     // per instance: end of the core walk; (element << 1) | strand
	__shared__ uint8_t served[RESUME_SLOTS];                          // ... its neighbourhood came from the block index (debugging aid)
	__shared__ unsigned nclaims, ninst_s;
	__shared__ unsigned s_sep[64];                                     // the separators' slots (see SepBounds), when there are at most 64
	const unsigned *sepl = g.sep && g.nsep <= 64 ? s_sep : nullptr;
	if (sepl && threadIdx.x < 64) s_sep[threadIdx.x] = threadIdx.x < g.nsep ? g.sep[threadIdx.x] : BT_NONE;
	for (unsigned i = threadIdx.x; i < (1u << seen_bits); i += 64 * RSV_WAVES) seen[i] = BT_NONE;
	unsigned *const my_list = rsv_dyn + (1u << seen_bits) + (size_t)(threadIdx.x >> 6) * list_cap;      // the marked slots of a group of instances, compacted (reserve_idx_emit)
	if (threadIdx.x == 0) nclaims = 0;
	unsigned id = g.win[w], st = g.round_bits | w;
	// the instances: handed over by the probe of this round (k_probe_idx: one coalesced read), or ListPositions by 64 lanes (wave_list_nodes)
	const unsigned given = instbuf ? instbuf[(size_t)w * istride] : BT_NONE;
	if (given != BT_NONE && given <= RESUME_SLOTS) {
		for (unsigned i = threadIdx.x; i < given; i += 64 * RSV_WAVES) inst[i] = instbuf[(size_t)w * istride + 1 + i];
		if (threadIdx.x == 0) ninst_s = given;
	} else if (wv == 0) {
		const unsigned m = wave_list_nodes(g, g.head[0][id], g.head[1][id], lane, nullptr, [&](unsigned off, unsigned, unsigned s, unsigned el, unsigned) {
			if (off < RESUME_SLOTS) inst[off] = (el << 1) | s;
		});
		if (lane == 0) ninst_s = m;
	}
	__syncthreads();
	RSV_T(0);
	ClaimList cl; cl.buf = claims + (size_t)w * (CLAIM_CAP + 1); cl.n = &nclaims; cl.seen = seen; cl.sbits = seen_bits;
	if (wv == 0) wave_claim(g, cl, st, lane == 0 ? id : BT_NONE, lane);
	unsigned back = g.D + g.k + 2, fwd = 2 * (g.D + g.k + 2) + g.k, core = g.D + 2 * g.k + 3;
	// Who can interact with an instance: anything marked where the transaction itself reads or writes (core, both
	// strands) -- claimed exclusively, the instance lists of those ids may be rewritten; instances upstream on the same
	// strand and further downstream on the opposite strand walk towards the core -- the transaction can only make them
	// stale, which orders it against them (bt_footprint, bulge_txn.h); instances walking away cannot see or touch it.
	auto order = [&](unsigned b0, unsigned b1) { wave_claim_order(g, cl, st, id, b0, lane); wave_claim_order(g, cl, st, id, b1, lane); };
	const unsigned ninst = ninst_s;
	const bool burst = !(g.test_flags & 4u);                          // (SBL_TEST_FLAGS=4: the step-wise walks everywhere, for A/B runs)
	// block index (round 5): groups of four instances, sixteen lanes each (reserve_idx_masks); an instance it cannot serve takes the walks
	const unsigned NA = (fwd + 1u + 126u) >> 6, NB = (back + 1u + 126u) >> 6;
	const bool indexed = g.idx_reserve && NA + NB <= 16u && ninst <= RESUME_SLOTS;
	if (indexed) {
		// one pass per group: the records once, exclusive then ordering claims from the same masks (the seen-set keeps the kinds apart, see
		// wave_claim), the instances the index cannot serve through the walks -- no barrier between the waves of the workgroup
		for (unsigned i0 = 4u * wv; i0 < ninst; i0 += 4u * RSV_WAVES) {
			RsvIdxLane L;
			const unsigned slowm = reserve_idx_masks(g, inst, ninst, i0, lane, NA, NB, core, fwd, back, L);
			if (rprof && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_rsv_ticks[6], n_ - rt); rt = n_; }      // (the records have arrived)
			if (!reserve_idx_emit(g, L, L.ex0, 0u, L.ex1, 1u, false, my_list, list_cap, lane, [&](unsigned b) { wave_claim(g, cl, st, b, lane); })) {
				reserve_idx_gather(g, L, L.ex0, 0u, false, [&](unsigned b) { wave_claim(g, cl, st, b, lane); });
				reserve_idx_gather(g, L, L.ex1, 1u, false, [&](unsigned b) { wave_claim(g, cl, st, b, lane); });
			}
			if (rprof && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_rsv_ticks[1], n_ - rt); rt = n_; }
			// (ahead lanes: opposite strand, slots in the instance's direction; behind lanes: own strand, the other way)
			if (!reserve_idx_emit(g, L, L.ord, L.ahead ? L.s ^ 1u : L.s, 0ull, 0u, !L.ahead, my_list, list_cap, lane, [&](unsigned b) { wave_claim_order(g, cl, st, id, b, lane); })) {
				reserve_idx_gather(g, L, L.ahead ? L.ord : 0ull, L.s ^ 1u, false, [&](unsigned b) { wave_claim_order(g, cl, st, id, b, lane); });
				reserve_idx_gather(g, L, L.ahead ? 0ull : L.ord, L.s, true, [&](unsigned b) { wave_claim_order(g, cl, st, id, b, lane); });
			}
			if (lane < 4u && i0 + lane < ninst) { served[i0 + lane] = (uint8_t)(((slowm >> lane) & 1u) ^ 1u); if (g.test_flags & 32u) atomicAdd(&g_idx_stats[6 + ((slowm >> lane) & 1u)], 1u); }
			if (rprof && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_rsv_ticks[2], n_ - rt); rt = n_; }
			for (unsigned q = 0; q < 4u && i0 + q < ninst; q++) {
				if (!((slowm >> q) & 1u)) continue;
				const unsigned i = i0 + q;
				const unsigned e0 = inst[i] >> 1, s = inst[i] & 1u;
				const SepBounds sp = sep_bounds(g, sepl, e0, lane);
				unsigned nxt = BT_NONE;
				if (!burst || !wave_core_claim_burst(g, e0, s, core, lane, cl, st, sp, nxt))
					nxt = wave_walk_claim(g, e0, s, core, lane, 3u, cl, st, sp);
				if (burst && wave_flank_order_burst(g, e0, s, fwd + 1 > core ? nxt : BT_NONE, fwd + 1 > core ? fwd + 1 - core : 0u, back, lane, sp,
				                                    [&](unsigned b) { wave_claim_order(g, cl, st, id, b, lane); })) continue;
				wave_walk_marks2(g, fwd + 1 > core ? nxt : BT_NONE, s, fwd + 1 - core, 1u << (s ^ 1u),
				                 s ? g.nx[e0] : g.pv[e0], s ^ 1u, back, 1u << s, lane, order, sp);
			}
			if (rprof && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_rsv_ticks[7], n_ - rt); rt = n_; }
		}
	} else if (ninst <= RESUME_SLOTS) {
		for (unsigned i = wv; i < ninst; i += RSV_WAVES) {            // all exclusive claims first: the seen-set keeps the first kind
			const SepBounds sp = sep_bounds(g, sepl, inst[i] >> 1, lane);
			unsigned nxt = BT_NONE;
			if (!burst || !wave_core_claim_burst(g, inst[i] >> 1, inst[i] & 1u, core, lane, cl, st, sp, nxt))
				nxt = wave_walk_claim(g, inst[i] >> 1, inst[i] & 1u, core, lane, 3u, cl, st, sp);
			if (lane == 0) resume[i] = nxt;
		}
		__syncthreads();
		for (unsigned i = wv; i < ninst; i += RSV_WAVES) {
			const unsigned e0 = inst[i] >> 1, s = inst[i] & 1u, nxt = resume[i];
			const SepBounds sp = sep_bounds(g, sepl, e0, lane);
			// further downstream (opposite strand) and upstream (same strand) together; all three walks of an instance stay in its chromosome
			if (burst && wave_flank_order_burst(g, e0, s, fwd + 1 > core ? nxt : BT_NONE, fwd + 1 > core ? fwd + 1 - core : 0u, back, lane, sp,
			                                    [&](unsigned b) { wave_claim_order(g, cl, st, id, b, lane); })) continue;
			wave_walk_marks2(g, fwd + 1 > core ? nxt : BT_NONE, s, fwd + 1 - core, 1u << (s ^ 1u),
			                 s ? g.nx[e0] : g.pv[e0], s ^ 1u, back, 1u << s, lane, order, sp);
		}
	} else {                                                          // more instances than the LDS list holds: walk the node lists
		unsigned k1 = 0;
		for (unsigned s = 0; s < 2; s++)
			for (unsigned nd = g.head[s][id]; nd != BT_NONE; nd = g.nnext[nd]) {
				if (g.ndead[nd]) continue;
				if (k1 % RSV_WAVES == wv) wave_walk_claim(g, g.nslot[nd], s, core, lane, 3u, cl, st);
				k1++;
			}
		__syncthreads();
		unsigned k2 = 0;
		for (unsigned s = 0; s < 2; s++)
			for (unsigned nd = g.head[s][id]; nd != BT_NONE; nd = g.nnext[nd]) {
				if (g.ndead[nd]) continue;
				if (k2 % RSV_WAVES == wv) {
					unsigned e0 = g.nslot[nd];
					unsigned nxt = wave_walk_marks(g, e0, s, core, lane, 0u, [](unsigned, unsigned) {});
					if (nxt != BT_NONE && fwd + 1 > core) wave_walk_marks(g, nxt, s, fwd + 1 - core, lane, 1u << (s ^ 1u), order);
					wave_walk_marks(g, s ? g.nx[e0] : g.pv[e0], s ^ 1u, back, lane, 1u << s, order);
				}
				k2++;
			}
	}
	__syncthreads();
	RSV_T(2);
	if (threadIdx.x == 0) { cl.buf[0] = nclaims; if (rprof) { atomicAdd(&g_rsv_ticks[3], 1ull); atomicAdd(&g_rsv_ticks[4], (unsigned long long)nclaims); atomicAdd(&g_rsv_ticks[5], (unsigned long long)ninst); } }
#undef RSV_T
}
// ---- wave-wide CollapseBulgeGreedily ------------------------------------------------------------------------
// Same effect as bt_collapse (bulge_txn.h) = EraseBifurcations + DNASequence::Replace + UpdateBifurcations
// (reference src/bulgeremoval.cpp:55-95, 238-327, src/dnasequence.cpp:189-252), but every element the reference reaches
// by walking iterators is taken from the cached windows of the target (T) and source (S) instances, so the k + dT and
// dS + 1 step loops run 64 steps at a time; only the order-dependent parts stay on lane 0: the ~10 AddPoint calls
// (front insertion order matters) and the position interpolation (sequential double accumulation).
__device__ __forceinline__ void wave_stamp_id_write(const GraphView &g, unsigned stampv, unsigned tid, unsigned id, unsigned b)
{
	unsigned r = g.nblk + b;
	unsigned ow = g.own[b], wm = g.wmax[r], rm = g.rmax[r];
	bool bad = (stampv != BT_NONE && ow != stampv) || wm > tid || rm > tid;   // not in its claims (escaped the reservation; none in the serial chain), or a higher id was here first
	atomicMax(&g.wmax[r], tid);
	if (bad) {
		atomicMin(&g.ctr[CTR_VIOL], id);
		if (atomicCAS(&g.ctr[CTR_DETAIL], 0u, 3u) == 0u) { g.ctr[CTR_DETAIL + 1] = r; g.ctr[CTR_DETAIL + 2] = (wm > rm ? wm : rm) - 1; g.ctr[CTR_DETAIL + 3] = id; g.ctr[CTR_DETAIL + 4] = (wm > tid ? 1u : 0u) | (rm > tid ? 2u : 0u) | (ow != stampv ? 4u : 0u); }
	}
}
// ErasePoint (bifurcationstorage.cpp:144-155) for one (strand, element) per lane; the lazy-erase chain head lives in LDS
// b / nd: the mark and its node as the caller loaded them (all loads of a step are issued together: the erase loops used to be a chain
// of five dependent look-ups per step -- element, mark, node, mark of the other strand, its node)
__device__ __forceinline__ void wave_erase(const GraphView &g, Txn &t, unsigned strand, unsigned e, unsigned stampv, unsigned b, unsigned nd)
{
	if (b == BT_NONE) return;
	g.bif[strand][e] = BT_NONE;
	bt_idx_mark(g, strand, e, false);
	g.ndead[nd] = 1;
	g.nclr[nd] = atomicExch(&t.tc_head, nd);
	{ unsigned ix = atomicAdd(&t.tc_n, 1u); if (ix < t.tc_cap) t.tc_list[ix] = nd; }
	if (t.mode) wave_stamp_id_write(g, stampv, t.tid, t.id, b);
	if (b < g.nid) { g.touch[b] = 1; if (b > t.id) g.need[b] = 1; }
}

__device__ unsigned long long g_txn_hist[4][16];   // SBL_PHASES=1: transactions by number of collapses (0, 1, 2, 3+) x log2(duration / 8192 cycles)
__device__ unsigned long long g_txn_max[2];        // longest transaction: cycles, (instances << 32) | collapses
__device__ unsigned long long g_round_max[4096];   // SBL_PHASES=1: per launch of k_commit (slot = round stamp slot / 4), the slowest transaction: (cycles << 24) | min(instances, 255) << 16 | old-form collapses << 8 | collapses
__device__ unsigned g_old_collapses;
__device__ unsigned long long g_round_span[4096 * 3];   // SBL_PHASES=1, per launch (wall clock, 10 ns ticks): earliest start of an owner, latest end, (duration << 32) | start of the slowest
__device__ unsigned long long g_round_few[4096 * 2];    // ... and the slowest transaction with at most one / at most two collapses
#define PH_T0() unsigned long long ph_t = prof ? __builtin_readcyclecounter() : 0ull; const unsigned long long ph_start = ph_t; const unsigned long long ph_wall = prof ? wall_clock64() : 0ull
#define PH_ADD(i) do { if (prof && lane == 0) { unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(&g_phase_cycles[i], n_ - ph_t); ph_t = n_; } } while (0)
#ifndef AP_CHUNKS
#define AP_CHUNKS 4                          // AddPoints of a collapse handled one per lane: up to AP_CHUNKS x 64 (more: one lane, one after the other)
#endif
#define PC_T0() unsigned long long pc_t = prof ? __builtin_readcyclecounter() : 0ull
#define PC_ADD(i) do { if (prof && lane == 0) { unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(&g_phase_cycles[i], n_ - pc_t); pc_t = n_; } } while (0)
// The AddPoints of a collapse, one per lane and chunk of 64 lanes (NC chunks: instantiated for 1 -- the usual few dozen -- and for AP_CHUNKS).
template <int NC, class NewT>
__device__ __forceinline__ void wave_add_points(const GraphView &g, Txn &t, BulgeWork &w, unsigned lane, const unsigned *T, NewT newT, unsigned k, unsigned d, unsigned opp,
                                                unsigned dS, unsigned nlb, unsigned nlf, unsigned total, unsigned s_nodebase, const unsigned *actp)
{
	const unsigned t0 = T[0];
	// place of a restored flank mark in the reference's order = its place in its own list + the entries of the OTHER list it comes after:
	// with both lists in one wave's registers (k <= 64) that count is a loop of shuffles, not a walk over the list in memory per lane
	const bool inreg = nlb <= 64u && nlf <= 64u && (g.test_flags & 2u);
	const unsigned my_lb = inreg && lane < nlb ? w.lb[2 * lane] : ~0u, my_lf = inreg && lane < nlf ? w.lf[2 * lane] : ~0u;
	unsigned cnt_lb = 0;                                                   // lookForward entries with a smaller index than my lookBack entry
	if (inreg) for (unsigned y = 0; y < nlf; y++) cnt_lb += __shfl(my_lf, y) < my_lb ? 1u : 0u;
	// One AddPoint per lane and chunk of 64 (up to NC x 64 of them: with dozens of strains half of all positions are
	// bifurcations and a collapse copies 60 - 150 marks -- one lane doing them one after the other was 12 % of k_commit at 62 strains).
	// seq = its place in the reference's order (flanks merged by index, look-back first at equal index, then the copied source
	// marks); an element that already carries a mark ignores later AddPoints, and the insertions into one list chain up in seq
	// order (front insertion: the last one becomes the head).
	const unsigned nch = (total + 63u) >> 6;
	unsigned seq[NC], ekey[NC], lkey[NC], cur[NC];
	bool valid[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) {
		const unsigned x = lane + 64u * c;
		unsigned sq = BT_NONE, ad = 0, ae = 0, ab = BT_NONE;
		// (uniform part: the index of my lookForward entry and how many lookBack entries come before it)
		const unsigned bi_u = x >= nlb && x < nlb + nlf ? x - nlb : 0u;
		const unsigned idx_lf = inreg ? __shfl(my_lf, bi_u & 63u) : 0u;
		unsigned cnt_lf = 0;
		if (inreg) for (unsigned y = 0; y < nlb; y++) cnt_lf += __shfl(my_lb, y) <= idx_lf ? 1u : 0u;
		if (x < nlb) {
			unsigned idx = inreg ? my_lb : w.lb[2 * x], cc = cnt_lb;
			if (!inreg) for (unsigned y = 0; y < nlf; y++) cc += w.lf[2 * y] < idx;
			sq = x + cc; ad = opp; ae = T[k - 1 - idx]; ab = w.lb[2 * x + 1];
		} else if (x < nlb + nlf) {
			unsigned bi = x - nlb, idx = inreg ? idx_lf : w.lf[2 * bi], cc = cnt_lf;
			if (!inreg) for (unsigned y = 0; y < nlb; y++) cc += w.lb[2 * y] <= idx;
			sq = bi + cc; ad = d; ae = newT(dS + idx); ab = w.lf[2 * bi + 1];
		} else if (x < total) {
			unsigned xa = x - nlb - nlf;
			sq = x; ad = actp[3 * xa]; ae = actp[3 * xa + 1]; ab = actp[3 * xa + 2];
		}
		seq[c] = sq; ekey[c] = (ae << 1) | ad; lkey[c] = (ab << 1) | ad;
		// what the element carries NOW is known without a look: EraseBifurcations has just cleared both strands over the whole range the
		// AddPoints fall into (flanks and replaced span; new elements start unmarked) -- except the own-strand mark of the target instance
		// itself (step 0 is never erased, bulgeremoval.cpp:87-93), which the copy of the source's own mark at step 0 runs into
		cur[c] = sq != BT_NONE && ab != BT_NONE ? (ad == d && ae == t0 ? 0u : BT_NONE) : 0u;
	}
#pragma unroll
	for (int c = 0; c < NC; c++) valid[c] = seq[c] != BT_NONE && (lkey[c] >> 1) != BT_NONE && cur[c] == BT_NONE;
	// an earlier AddPoint on the same (strand, element) wins
#pragma unroll
	for (int c = 0; c < NC; c++) {
		if ((unsigned)c >= nch) break;
#pragma unroll
		for (int c2 = 0; c2 < NC; c2++) {
			if ((unsigned)c2 >= nch) break;
			const unsigned upto = total - 64u * c2 < 64u ? total - 64u * c2 : 64u;
			for (unsigned y = 0; y < upto; y++) {
				const unsigned ky = __shfl(ekey[c2], y), sy = __shfl(seq[c2], y);
				if (valid[c] && !(c2 == c && y == lane) && ky == ekey[c] && sy < seq[c]) valid[c] = false;
			}
		}
	}
	unsigned pred[NC], cnt[NC], hd[NC], ls[NC];
	bool last[NC];
#pragma unroll
	for (int c = 0; c < NC; c++) {
		pred[c] = BT_NONE; cnt[c] = 0; last[c] = true; hd[c] = 0; ls[c] = 0;
		if ((unsigned)c >= nch) continue;
		if (valid[c]) { hd[c] = g.head[lkey[c] & 1u][lkey[c] >> 1]; ls[c] = g.lsize[lkey[c] & 1u][lkey[c] >> 1]; }      // (every look at a head before any of them is rewritten)
#pragma unroll
		for (int c2 = 0; c2 < NC; c2++) {
			if ((unsigned)c2 >= nch) break;
			const unsigned upto = total - 64u * c2 < 64u ? total - 64u * c2 : 64u;
			for (unsigned y = 0; y < upto; y++) {
				const unsigned ky = __shfl(lkey[c2], y), sy = __shfl(seq[c2], y);
				const bool vy = __shfl((int)valid[c2], y) != 0;
				if (vy && ky == lkey[c]) {
					cnt[c]++;
					if (sy < seq[c] && (pred[c] == BT_NONE || sy > pred[c])) pred[c] = sy;
					if (sy > seq[c]) last[c] = false;
				}
			}
		}
	}
#pragma unroll
	for (int c = 0; c < NC; c++) {
		if ((unsigned)c >= nch || !valid[c]) continue;
		const unsigned ad = lkey[c] & 1u, ab = lkey[c] >> 1, ae = ekey[c] >> 1;
		const unsigned nd = s_nodebase + seq[c];
		g.nslot[nd] = ae; g.ndead[nd] = 0; g.nidst[nd] = (ab << 1) | ad;
		g.nnext[nd] = pred[c] != BT_NONE ? s_nodebase + pred[c] : hd[c];
		if (last[c]) { g.head[ad][ab] = nd; g.lsize[ad][ab] = ls[c] + cnt[c]; }
		g.bif[ad][ae] = ab; g.nodeof[ad][ae] = nd;
		bt_idx_mark(g, ad, ae, true);
		if (ab < g.nid) { g.touch[ab] = 1; if (ab > t.id) g.need[ab] = 1; }
	}
}

__device__ __forceinline__ void wave_collapse(const GraphView &g, Txn &t, BulgeWork &w, unsigned lane, unsigned stampv, const int prof = 0)
{
	PC_T0();
	const unsigned k = g.k, ws = w.ws;
	const unsigned src = w.c_src, dS = w.c_dS, tgt = w.c_tgt, dT = w.c_dT;
	const unsigned d = w.start[tgt] & 1u, opp = d ^ 1u, ds = w.start[src] & 1u;
	const unsigned *T = w.wel + (size_t)tgt * ws, *S = w.wel + (size_t)src * ws;
	const unsigned long long lt = (1ull << lane) - 1ull;
	// ---- EraseBifurcations, first loop: remember and erase the k-flanks (lookBack on the opposite strand, lookForward ahead)
	unsigned nlb = 0, nlf = 0;
	for (unsigned i0 = 0; i0 < k; i0 += 64) {
		unsigned i = i0 + lane;
		bool in = i < k;
		unsigned ea = in ? T[k - 1 - i] : 0u, eb = in ? T[dT + i] : 0u;
		unsigned ba = in ? g.bif[opp][ea] : BT_NONE, bb = in ? g.bif[d][eb] : BT_NONE;
		unsigned na = in ? g.nodeof[opp][ea] : 0u, nb2 = in ? g.nodeof[d][eb] : 0u;
		// (a lane's two flank positions are on different strands, and no two lanes share a (strand, element) pair: the preloaded marks are current)
		unsigned long long ma = __ballot(ba != BT_NONE), mb = __ballot(bb != BT_NONE);
		if (ba != BT_NONE) { unsigned o = nlb + __popcll(ma & lt); w.lb[2 * o] = i; w.lb[2 * o + 1] = ba; wave_erase(g, t, opp, ea, stampv, ba, na); }
		if (bb != BT_NONE) { unsigned o = nlf + __popcll(mb & lt); w.lf[2 * o] = i; w.lf[2 * o + 1] = bb; wave_erase(g, t, d, eb, stampv, bb, nb2); }
		nlb += __popcll(ma); nlf += __popcll(mb);
	}
	WSYNC();
	PC_ADD(9);
	// ---- second loop: every own-strand mark after the target start and every opposite-strand mark over k + dT elements
	for (unsigned i0 = 0; i0 < k + dT; i0 += 64) {
		unsigned i = i0 + lane;
		if (i < k + dT) {
			const unsigned e = T[i];
			const unsigned b0 = g.bif[d][e], b1 = g.bif[opp][e], n0 = g.nodeof[d][e], n1 = g.nodeof[opp][e];
			if (i > 0) wave_erase(g, t, d, e, stampv, b0, n0);
			wave_erase(g, t, opp, e, stampv, b1, n1);
		}
	}
	WSYNC();
	PC_ADD(10);
	// ---- DNASequence::Replace in + coordinates: P(j) = j-th element of the old span, C(j) = j-th new character.
	// All lanes: character writes, the new elements of an insertion and the position interpolation (the sequence
	// acc += ssize of dnasequence.cpp:221-227 is replayed in registers, every lane keeps the value of its own step).
	__shared__ unsigned s_newbase;
	const unsigned common = dS < dT ? dS : dT;
	auto P = [&](unsigned jx) { return d == 0 ? T[k + jx] : T[k + dT - 1 - jx]; };
	auto OC = [&](unsigned x) { char c = (char)w.wch[(size_t)src * ws + x]; return ds ? bt_comp(c) : c; };
	auto C = [&](unsigned jx) { return d == 0 ? OC(k + jx) : bt_comp(OC(k + dS - 1 - jx)); };
	const unsigned Eafter = d == 0 ? T[k + dT] : T[k - 1];
	const unsigned firstPos = g.op[P(0)] & BT_POS_MASK, lastPos = g.op[Eafter] & BT_POS_MASK;
	if (lane == 0) {
		t.wrote = true;
		unsigned newbase = BT_NONE;
		if (dS > dT) {
			unsigned span = bt_insert_span(dS - dT);
			unsigned base = atomicAdd(&g.ctr[CTR_NE], span);
			if (base + span > g.cap_e) t.err |= BT_ERR_ELEM_CAP; else newbase = base;
		}
		s_newbase = newbase;
	}
	WSYNC();
	PC_ADD(11);
	if (t.err) return;
	if (dS != dT)                                                      // links change between T[k - 1] and T[k + dT]: those blocks are no longer pristine (GraphView::bidx)
		for (unsigned i = k - 1u + lane; i <= k + dT; i += 64) bt_idx_dirty(g, T[i]);
	{
		const unsigned nb = s_newbase;
		for (unsigned j0 = 0; j0 < (dS < dT ? dT : common); j0 += 64) {
			unsigned jx = j0 + lane;
			if (jx < common) g.ch[P(jx)] = (uint8_t)C(jx);
			else if (jx < dT) g.ch[P(jx)] = BT_DEAD_CHAR;                 // deletion: the tail of the old span dies
		}
		if (dS < dT) {
			if (lane == 0) { unsigned before = P(dS - 1); g.nx[before] = Eafter; g.pv[Eafter] = before; }
		} else if (dS > dT) {
			const unsigned m = dS - dT, span = bt_insert_span(m), before0 = P(dT - 1);
			for (unsigned i0 = 0; i0 < span; i0 += 64) {
				unsigned i = i0 + lane;
				if (i >= span) break;
				unsigned ne = nb + i;
				g.bif[0][ne] = BT_NONE; g.bif[1][ne] = BT_NONE;
				if (i < m) {
					g.ch[ne] = (uint8_t)C(dT + i); g.op[ne] = 0;
					g.pv[ne] = i ? ne - 1 : before0;
					g.nx[ne] = i + 1 < m ? ne + 1 : Eafter;
				} else g.ch[ne] = BT_DEAD_CHAR;
			}
			if (lane == 0) { g.nx[before0] = nb; g.pv[Eafter] = nb + m - 1; }
		}
		double acc = (double)firstPos;
		const double ssize = (double)dT / (double)dS;
		for (unsigned j0 = 0; j0 < dS; j0 += 64) {
			const unsigned cnt = dS - j0 < 64u ? dS - j0 : 64u;
			double mine = 0.0;
			for (unsigned jj = 0; jj < cnt; jj++) { if (jj == lane) mine = acc; acc += ssize; }
			unsigned jx = j0 + lane;
			if (jx < dS) {
				unsigned long long pp = (unsigned long long)mine;
				if (pp > lastPos) pp = lastPos;
				unsigned e = jx < common ? P(jx) : nb + (jx - dT);
				g.op[e] = (unsigned)pp & BT_POS_MASK;
			}
		}
	}
	WSYNC();
	PC_ADD(12);
	if (t.err) return;
	const unsigned newbase = s_newbase;
	// element at step s of the target walk AFTER the replacement
	auto newT = [&](unsigned s) -> unsigned {
		if (s < k) return T[s];
		if (s >= k + dS) return T[s - dS + dT];
		unsigned idx = s - k, fj = d == 0 ? idx : dS - 1 - idx;
		return fj < common ? (d == 0 ? T[k + fj] : T[k + dT - 1 - fj]) : newbase + (fj - dT);
	};
	// ---- UpdateBifurcations, second loop first as DATA: source marks to copy, in the reference's order (own strand, then opposite)
	unsigned nact = 0;
	for (unsigned i0 = 0; i0 <= dS; i0 += 64) {
		unsigned i = i0 + lane;
		bool in = i <= dS;
		unsigned b1 = in ? w.wbf[(size_t)src * ws + i] : BT_NONE;
		unsigned b2 = in ? g.bif[ds ^ 1u][S[dS + k - 1 - i]] : BT_NONE;
		unsigned long long m1 = __ballot(b1 != BT_NONE), m2 = __ballot(b2 != BT_NONE);
		unsigned o = nact + __popcll(m1 & lt) + __popcll(m2 & lt);
		if (b1 != BT_NONE) { w.act[3 * o] = d; w.act[3 * o + 1] = newT(i); w.act[3 * o + 2] = b1; o++; }
		if (b2 != BT_NONE) { w.act[3 * o] = opp; w.act[3 * o + 1] = newT(dS + k - 1 - i); w.act[3 * o + 2] = b2; }
		nact += __popcll(m1) + __popcll(m2);
	}
	WSYNC();
	PC_ADD(13);
	// nodes for every AddPoint below in one allocation; the ids they touch are stamped by all lanes at once
	__shared__ unsigned s_nodebase;
	const unsigned total = nlb + nlf + nact;
	if (lane == 0) {
		unsigned base = total ? atomicAdd(&g.ctr[CTR_NN], total) : 0u;
		if (total && base + total > g.cap_n) t.err |= BT_ERR_NODE_CAP;
		s_nodebase = base;
	}
	if (t.mode)
		for (unsigned x = lane; x < total; x += 64) {
			unsigned b = x < nlb ? w.lb[2 * x + 1] : x < nlb + nlf ? w.lf[2 * (x - nlb) + 1] : w.act[3 * (x - nlb - nlf) + 2];
			wave_stamp_id_write(g, stampv, t.tid, t.id, b);
		}
	WSYNC();
	PC_ADD(14);
	if (t.err) return;
	if (total > 64u * AP_CHUNKS) {
		if (lane == 0) {
			unsigned nd = s_nodebase;
			// first loop: restore the flanks (merge of the two index-sorted lists, look-back before look-forward at equal index)
			unsigned a = 0, b = 0;
			while (a < nlb || b < nlf) {
				bool takeA = b >= nlf || (a < nlb && w.lb[2 * a] <= w.lf[2 * b]);
				SIt p;
				if (takeA) { p.e = T[k - 1 - w.lb[2 * a]]; p.d = opp; t.add_point_prepared(p, w.lb[2 * a + 1], nd++); a++; }
				else { p.e = newT(dS + w.lf[2 * b]); p.d = d; t.add_point_prepared(p, w.lf[2 * b + 1], nd++); b++; }
			}
			for (unsigned x = 0; x < nact; x++) { SIt p; p.d = w.act[3 * x]; p.e = w.act[3 * x + 1]; t.add_point_prepared(p, w.act[3 * x + 2], nd++); }
		}
	} else {
		if (total <= 64u) wave_add_points<1>(g, t, w, lane, T, newT, k, d, opp, dS, nlb, nlf, total, s_nodebase, w.act);
		else wave_add_points<AP_CHUNKS>(g, t, w, lane, T, newT, k, d, opp, dS, nlb, nlf, total, s_nodebase, w.act);
	}
	if (lane == 0) { t.push_e = T[0]; t.push_d = d; t.push_len = dS; }
	WSYNC();
	PC_ADD(15);
}

// ---- gather-first CollapseBulgeGreedily (round 4) ------------------------------------------------------------------------------------
// wave_collapse above is a chain of ~14 dependent memory round trips: window elements from the arena, then their marks, then the stamps
// of those marks -- twice, for the two loops of EraseBifurcations --, positions, allocation, the marks to copy, their stamps, the heads of
// the lists.  A collapse is 38 % of a transaction and a round lasts as long as its slowest transaction, so the order is turned round:
//   1  the steps of the target and of the source window the collapse looks at, one per lane and 64-step chunk, into REGISTERS;
//   2  every graph value it needs about them in one batch (marks and nodes of both strands of the target range, opposite-strand marks
//      of the source range, original positions);
//   3  the stamp words of every id it will erase or copy, and the two pool allocations, in one batch;
//   4  checks, then nothing but stores (erase, characters, links, new elements, positions), the AddPoint list, and the AddPoints.
// Same effect as wave_collapse (the two erase loops fuse: the flank marks of the first are a subset of the range of the second, and
// the order of erasure is unobservable -- lazy-erase chain and list sizes are order-free).  NC = 64-step chunks per window (1 or 3).
#ifndef GATHER_CHUNKS_MAX
#define GATHER_CHUNKS_MAX 1
#endif
template <int NC>
__device__ __forceinline__ unsigned gsel(const unsigned (&r)[NC], unsigned x)      // r "at step x": every lane must take part
{
	unsigned v = __shfl(r[0], x & 63u);
	if (NC > 1) { const unsigned v1 = __shfl(r[NC > 1 ? 1 : 0], x & 63u); v = (x >> 6) == 1u ? v1 : v; }
	if (NC > 2) { const unsigned v2 = __shfl(r[NC > 2 ? 2 : 0], x & 63u); v = (x >> 6) >= 2u ? v2 : v; }
	return v;
}
template <int NC>
__device__ __forceinline__ void wave_collapse_g(const GraphView &g, Txn &t, BulgeWork &w, unsigned lane, unsigned stampv, const int prof = 0)
{
	PC_T0();
	const unsigned k = g.k, ws = w.ws, tid = t.tid, id = t.id;
	const unsigned src = w.c_src, dS = w.c_dS, tgt = w.c_tgt, dT = w.c_dT;
	const unsigned d = w.start[tgt] & 1u, opp = d ^ 1u, ds = w.start[src] & 1u;
	const unsigned *T = w.wel + (size_t)tgt * ws, *S = w.wel + (size_t)src * ws, *SB = w.wbf + (size_t)src * ws;
	const uint8_t *SCH = w.wch + (size_t)src * ws;
	const unsigned nT = k + dT + 1, nS = dS + k, nE = k + dT;             // target steps looked at (incl. the element after the span) / source steps / erase range
	const unsigned long long lt = (1ull << lane) - 1ull, gt = lane == 63u ? 0ull : (~0ull << (lane + 1u));
	const bool stamped = t.mode != 0;
	// ---- 1: the two walks into registers
	unsigned Tv[NC], Sv[NC], Sb[NC], Sc[NC];
#pragma unroll
	for (int u = 0; u < NC; u++) {
		const unsigned x = lane + 64u * u;
		Tv[u] = ldg(&T[x < nT ? x : 0u]); Sv[u] = ldg(&S[x < nS ? x : 0u]);
		Sb[u] = ldg(&SB[x <= dS ? x : 0u]); Sc[u] = ldg(&SCH[x < nS ? x : 0u]);
	}
#pragma unroll
	for (int u = 0; u < NC; u++) if (lane + 64u * u > dS) Sb[u] = BT_NONE;
	// ---- 2: everything the graph knows about them
	unsigned bd[NC], bo[NC], nd[NC], no[NC], bs2[NC], opv[NC];
#pragma unroll
	for (int u = 0; u < NC; u++) {
		const unsigned e = Tv[u], es = Sv[u];                              // (lanes beyond the ranges hold step 0: loads are unconditional, results masked)
		bd[u] = g.bif[d][e]; bo[u] = g.bif[opp][e]; nd[u] = g.nodeof[d][e]; no[u] = g.nodeof[opp][e]; opv[u] = g.op[e];
		bs2[u] = g.bif[ds ^ 1u][es];
	}
	unsigned long long mA[NC], mB[NC], m1[NC], m2[NC];
	bool ed[NC], eo[NC];
	unsigned b2[NC];
	unsigned nlb = 0, nlf = 0, nact = 0;
#pragma unroll
	for (int u = 0; u < NC; u++) {
		const unsigned x = lane + 64u * u;
		if (x >= nE) { bd[u] = BT_NONE; bo[u] = BT_NONE; }
		if (x < k - 1u || x >= nS) bs2[u] = BT_NONE;
		ed[u] = x >= 1u && bd[u] != BT_NONE;                               // own-strand marks after the target start, opposite-strand marks from it on
		eo[u] = bo[u] != BT_NONE;
		mA[u] = __ballot(x < k && bo[u] != BT_NONE);                      // lookBack: opposite strand over the first k steps, index k - 1 - x
		mB[u] = __ballot(x >= dT && x < dT + k && bd[u] != BT_NONE);      // lookForward: own strand from step dT on, index x - dT
		nlb += (unsigned)__popcll(mA[u]); nlf += (unsigned)__popcll(mB[u]);
	}
#pragma unroll
	for (int u = 0; u < NC; u++) {                                         // source marks to copy at index i = x: own strand at step i, opposite strand at step dS + k - 1 - i
		const unsigned x = lane + 64u * u;
		const unsigned v = gsel<NC>(bs2, x <= dS ? dS + k - 1u - x : 0u);
		b2[u] = x <= dS ? v : BT_NONE;
		m1[u] = __ballot(Sb[u] != BT_NONE); m2[u] = __ballot(b2[u] != BT_NONE);
		nact += (unsigned)__popcll(m1[u]) + (unsigned)__popcll(m2[u]);
	}
	// the two flank lists in index order (LDS)
#pragma unroll
	for (int u = 0; u < NC; u++) {
		const unsigned x = lane + 64u * u;
		if (x < k && bo[u] != BT_NONE) {
			unsigned o = (unsigned)__popcll(mA[u] & gt);
			for (int v = u + 1; v < NC; v++) o += (unsigned)__popcll(mA[v]);
			w.lb[2 * o] = k - 1u - x; w.lb[2 * o + 1] = bo[u];
		}
		if (x >= dT && x < dT + k && bd[u] != BT_NONE) {
			unsigned o = (unsigned)__popcll(mB[u] & lt);
			for (int v = 0; v < u; v++) o += (unsigned)__popcll(mB[v]);
			w.lf[2 * o] = x - dT; w.lf[2 * o + 1] = bd[u];
		}
	}
	PC_ADD(9);
	// ---- 3: allocations and the stamp words of every id touched, in one batch
	__shared__ unsigned s_newbase_g, s_nodebase_g;
	const unsigned total = nlb + nlf + nact;
	if (lane == 0) {
		t.wrote = true;
		unsigned newbase = BT_NONE;
		if (dS > dT) {
			const unsigned span = bt_insert_span(dS - dT);
			const unsigned base = atomicAdd(&g.ctr[CTR_NE], span);
			if (base + span > g.cap_e) t.err |= BT_ERR_ELEM_CAP; else newbase = base;
		}
		s_newbase_g = newbase;
		const unsigned nbase = total ? atomicAdd(&g.ctr[CTR_NN], total) : 0u;
		if (total && nbase + total > g.cap_n) t.err |= BT_ERR_NODE_CAP;
		s_nodebase_g = nbase;
	}
	unsigned sw[NC][4][3];                                                 // own / wmax / rmax of: erased own-strand id, erased opposite-strand id, copied own-strand id, copied opposite-strand id
	if (stamped) {
#pragma unroll
		for (int u = 0; u < NC; u++) {
			const unsigned ids[4] = { ed[u] ? bd[u] : 0u, eo[u] ? bo[u] : 0u, Sb[u] != BT_NONE ? Sb[u] : 0u, b2[u] != BT_NONE ? b2[u] : 0u };
#pragma unroll
			for (int q = 0; q < 4; q++) { sw[u][q][0] = g.own[ids[q]]; sw[u][q][1] = g.wmax[g.nblk + ids[q]]; sw[u][q][2] = g.rmax[g.nblk + ids[q]]; }
		}
#pragma unroll
		for (int u = 0; u < NC; u++) {
			const bool has[4] = { ed[u], eo[u], Sb[u] != BT_NONE, b2[u] != BT_NONE };
			const unsigned ids[4] = { bd[u], bo[u], Sb[u], b2[u] };
#pragma unroll
			for (int q = 0; q < 4; q++) {
				if (!has[q]) continue;
				const unsigned r = g.nblk + ids[q], ow = sw[u][q][0], wm = sw[u][q][1], rm = sw[u][q][2];
				const bool bad = (stampv != BT_NONE && ow != stampv) || wm > tid || rm > tid;      // not in its claims, or a higher id was here first
				atomicMax(&g.wmax[r], tid);
				if (bad) {
					atomicMin(&g.ctr[CTR_VIOL], id);
					if (atomicCAS(&g.ctr[CTR_DETAIL], 0u, 3u) == 0u) { g.ctr[CTR_DETAIL + 1] = r; g.ctr[CTR_DETAIL + 2] = (wm > rm ? wm : rm) - 1; g.ctr[CTR_DETAIL + 3] = id; g.ctr[CTR_DETAIL + 4] = (wm > tid ? 1u : 0u) | (rm > tid ? 2u : 0u) | (ow != stampv ? 4u : 0u); }
				}
			}
		}
	}
	WSYNC();
	PC_ADD(10);
	if (t.err) return;
	const unsigned newbase = s_newbase_g;
	// scalars of the replacement (every lane takes part in the shuffles)
	const unsigned common = dS < dT ? dS : dT;
	const unsigned Eafter = gsel<NC>(Tv, d == 0 ? k + dT : k - 1u);
	const unsigned firstPos = gsel<NC>(opv, d == 0 ? k : k + dT - 1u) & BT_POS_MASK, lastPos = gsel<NC>(opv, d == 0 ? k + dT : k - 1u) & BT_POS_MASK;
	const unsigned jb = dS ? dS - 1u : 0u, jb0 = dT ? dT - 1u : 0u;
	const unsigned before = gsel<NC>(Tv, d == 0 ? k + jb : k + dT - 1u - jb), before0 = gsel<NC>(Tv, d == 0 ? k + jb0 : k + dT - 1u - jb0);      // P(dS - 1), P(dT - 1)
	// ---- 4a: erase (ErasePoint for one (strand, element) per lane and chunk; the marks were stamped above)
#pragma unroll
	for (int u = 0; u < NC; u++) {
		const unsigned e = Tv[u];
#pragma unroll
		for (int q = 0; q < 2; q++) {
			const bool has = q ? eo[u] : ed[u];
			if (!has) continue;
			const unsigned strand = q ? opp : d, b = q ? bo[u] : bd[u], node = q ? no[u] : nd[u];
			g.bif[strand][e] = BT_NONE;
			bt_idx_mark(g, strand, e, false);
			g.ndead[node] = 1;
			g.nclr[node] = atomicExch(&t.tc_head, node);
			{ const unsigned ix = atomicAdd(&t.tc_n, 1u); if (ix < t.tc_cap) t.tc_list[ix] = node; }
			if (b < g.nid) { g.touch[b] = 1; if (b > id) g.need[b] = 1; }
		}
	}
	// ---- 4b: DNASequence::Replace in + coordinates (wave_collapse explains P / C and the replayed double accumulation)
	if (dS != dT) {                                                    // links change between T[k - 1] and T[k + dT]: those blocks are no longer pristine (GraphView::bidx)
#pragma unroll
		for (int u = 0; u < NC; u++) { const unsigned x = lane + 64u * u; if (x + 1u >= k && x < nT) bt_idx_dirty(g, Tv[u]); }
	}
	{
		const unsigned nb = newbase;
		for (unsigned j0 = 0; j0 < (dS < dT ? dT : common); j0 += 64) {
			const unsigned jx = j0 + lane, jc = jx < dT ? jx : 0u;
			const unsigned Pj = gsel<NC>(Tv, d == 0 ? k + jc : k + dT - 1u - jc);
			const unsigned sx = jx < common ? (d == 0 ? k + jx : k + dS - 1u - jx) : 0u;
			char c = (char)gsel<NC>(Sc, sx);
			c = ds ? bt_comp(c) : c;
			c = d == 0 ? c : bt_comp(c);
			if (jx < common) g.ch[Pj] = (uint8_t)c;
			else if (jx < dT) g.ch[Pj] = BT_DEAD_CHAR;                   // deletion: the tail of the old span dies
		}
		if (dS < dT) {
			if (lane == 0) { g.nx[before] = Eafter; g.pv[Eafter] = before; }
		} else if (dS > dT) {
			const unsigned m = dS - dT, span = bt_insert_span(m);
			for (unsigned i0 = 0; i0 < span; i0 += 64) {
				const unsigned i = i0 + lane;
				const unsigned jx = dT + (i < m ? i : 0u), sx = d == 0 ? k + jx : k + dS - 1u - jx;
				char c = (char)gsel<NC>(Sc, sx);
				c = ds ? bt_comp(c) : c;
				c = d == 0 ? c : bt_comp(c);
				if (i >= span) continue;
				const unsigned ne = nb + i;
				g.bif[0][ne] = BT_NONE; g.bif[1][ne] = BT_NONE;
				if (i < m) {
					g.ch[ne] = (uint8_t)c; g.op[ne] = 0;
					g.pv[ne] = i ? ne - 1 : before0;
					g.nx[ne] = i + 1 < m ? ne + 1 : Eafter;
				} else g.ch[ne] = BT_DEAD_CHAR;
			}
			if (lane == 0) { g.nx[before0] = nb; g.pv[Eafter] = nb + m - 1; }
		}
		double acc = (double)firstPos;
		const double ssize = (double)dT / (double)dS;
		for (unsigned j0 = 0; j0 < dS; j0 += 64) {
			const unsigned cnt = dS - j0 < 64u ? dS - j0 : 64u;
			double mine = 0.0;
			for (unsigned jj = 0; jj < cnt; jj++) { if (jj == lane) mine = acc; acc += ssize; }
			const unsigned jx = j0 + lane, jc = jx < common ? jx : 0u;
			const unsigned Pj = gsel<NC>(Tv, d == 0 ? k + jc : k + dT - 1u - jc);
			if (jx < dS) {
				unsigned long long pp = (unsigned long long)mine;
				if (pp > lastPos) pp = lastPos;
				const unsigned e = jx < common ? Pj : nb + (jx - dT);
				g.op[e] = (unsigned)pp & BT_POS_MASK;
			}
		}
	}
	PC_ADD(12);
	// element at step s of the target walk AFTER the replacement (every lane takes part)
	auto newTg = [&](unsigned s) -> unsigned {
		const unsigned idx = s >= k ? s - k : 0u, fj = d == 0 ? idx : dS - 1u - (idx < dS ? idx : 0u);
		const unsigned inside = fj < common ? (d == 0 ? k + fj : k + dT - 1u - fj) : 0u;
		const unsigned at = s < k ? s : s >= k + dS ? s - dS + dT : inside;
		const unsigned v = gsel<NC>(Tv, at < nT ? at : 0u);
		return (s >= k && s < k + dS && fj >= common) ? newbase + (fj - dT) : v;
	};
	// ---- 4c: the AddPoint actions of the copied source marks, in the reference's order (own strand, then opposite, per index)
	unsigned *const act = w.act_fast && nact <= BT_ACT_FAST ? w.act_fast : w.act;      // (the usual few dozen: through LDS, not through the arena)
#pragma unroll
	for (int u = 0; u < NC; u++) {
		const unsigned x = lane + 64u * u, i = x <= dS ? x : 0u;
		const unsigned e1 = newTg(i), e2 = newTg(dS + k - 1u - i);
		unsigned o = (unsigned)__popcll(m1[u] & lt) + (unsigned)__popcll(m2[u] & lt);
		for (int v = 0; v < u; v++) o += (unsigned)__popcll(m1[v]) + (unsigned)__popcll(m2[v]);
		if (Sb[u] != BT_NONE) { act[3 * o] = d; act[3 * o + 1] = e1; act[3 * o + 2] = Sb[u]; o++; }
		if (b2[u] != BT_NONE) { act[3 * o] = opp; act[3 * o + 1] = e2; act[3 * o + 2] = b2[u]; }
	}
	WSYNC();
	PC_ADD(13);
	// ---- 4d: the AddPoints (restored flanks merged by index, then the copied marks)
	const unsigned nodebase = s_nodebase_g;
	auto newT = [&](unsigned s) -> unsigned {                              // (pointer form, for the divergent code of wave_add_points)
		if (s < k) return T[s];
		if (s >= k + dS) return T[s - dS + dT];
		unsigned idx = s - k, fj = d == 0 ? idx : dS - 1 - idx;
		return fj < common ? (d == 0 ? T[k + fj] : T[k + dT - 1 - fj]) : newbase + (fj - dT);
	};
	if (total > 64u * AP_CHUNKS) {
		if (lane == 0) {
			unsigned node = nodebase;
			unsigned a = 0, b = 0;
			while (a < nlb || b < nlf) {
				bool takeA = b >= nlf || (a < nlb && w.lb[2 * a] <= w.lf[2 * b]);
				SIt p;
				if (takeA) { p.e = T[k - 1 - w.lb[2 * a]]; p.d = opp; t.add_point_prepared(p, w.lb[2 * a + 1], node++); a++; }
				else { p.e = newT(dS + w.lf[2 * b]); p.d = d; t.add_point_prepared(p, w.lf[2 * b + 1], node++); b++; }
			}
			for (unsigned x = 0; x < nact; x++) { SIt p; p.d = act[3 * x]; p.e = act[3 * x + 1]; t.add_point_prepared(p, act[3 * x + 2], node++); }
		}
	} else {
		if (total <= 64u) wave_add_points<1>(g, t, w, lane, T, newT, k, d, opp, dS, nlb, nlf, total, nodebase, act);
		else wave_add_points<AP_CHUNKS>(g, t, w, lane, T, newT, k, d, opp, dS, nlb, nlf, total, nodebase, act);
	}
	if (lane == 0) { t.push_e = T[0]; t.push_d = d; t.push_len = dS; }
	WSYNC();
	PC_ADD(15);
}
// the collapse of an ordered round / chain transaction: gather-first where the walks fit the register chunks
__device__ __forceinline__ void wave_collapse_any(const GraphView &g, Txn &t, BulgeWork &w, unsigned lane, unsigned stampv, const int prof)
{
	const unsigned span = (w.c_dT > w.c_dS ? w.c_dT : w.c_dS) + g.k + 1u;
	// (one chunk only: the three-chunk instantiation needs ~60 more registers, and inlined into k_commit it made EVERY transaction spill --
	// 504 B of scratch, commit 40 -> 52 ms; longer branches keep the round-3 form)
	if (prof && lane == 0 && (!g.collapse_g || span > 64u * GATHER_CHUNKS_MAX)) w.nold++;
	if (!g.collapse_g || span > 64u * GATHER_CHUNKS_MAX) wave_collapse(g, t, w, lane, stampv, prof);
	else if (span <= 64u) wave_collapse_g<1>(g, t, w, lane, stampv, prof);
	else wave_collapse_g<GATHER_CHUNKS_MAX>(g, t, w, lane, stampv, prof);
}

// ---- the caller side of BulgeWork::jscan: next member of [idJ, group end) that is still valid and whose endChar differs from I's
// (bt_rb_next_j with 64 lanes x 4 members per step: member -> instance -> node -> dead flag is three dependent look-ups)
__device__ __forceinline__ void wave_next_j(const GraphView &g, BulgeWork &w, unsigned lane)
{
	const unsigned ge = w.ab.grp_off[w.gi + 1];
	const char ecI = w.endc[w.ab.grp_mem[w.idI]];
	unsigned j0 = w.idJ, found = ge;
	WSYNC();                                                       // (everybody has read idJ before lane 0 moves it)
	while (j0 < ge && found == ge) {
		unsigned m[4], st[4]; char ec[4]; bool in[4];
#pragma unroll
		for (int u = 0; u < 4; u++) { const unsigned idx = j0 + 64u * u + lane; in[u] = idx < ge; m[u] = in[u] ? w.ab.grp_mem[idx] : 0u; }
#pragma unroll
		for (int u = 0; u < 4; u++) { st[u] = in[u] ? w.start[m[u]] : 0u; ec[u] = in[u] ? w.endc[m[u]] : ecI; }
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const bool cand = in[u] && ec[u] != ecI && !g.ndead[st[u] >> 1];
			const unsigned long long b = __ballot(cand);
			if (b && found == ge) found = j0 + 64u * u + (unsigned)__builtin_ctzll(b);
		}
		j0 += 256;
	}
	if (lane == 0) { w.idJ = found; w.jready = true; }
	WSYNC();
}

// ---- the caller side of BulgeWork::mscan: MaxBifurcationMultiplicity of the two branches, one CountBifurcations per lane (bt_rb_mults
// with 64 lanes; Txn::count_bif stamps the id exactly as the one-thread form does)
__device__ __attribute__((noinline)) void wave_mults(const GraphView &g, Txn &t, BulgeWork &w, unsigned lane)      // (out of line: it runs once per dense branch and must not cost the common path its registers)
{
	(void)g;
	unsigned res[2];
#pragma unroll
	for (int q = 0; q < 2; q++) {
		const unsigned i = q ? w.mq_j : w.mq_i, dist = q ? w.mq_dj : w.mq_di, nm = w.wmn[i];
		const unsigned long long *mk = reinterpret_cast<const unsigned long long *>(w.wmk) + (size_t)i * w.mks;
		unsigned r = 0;
		for (unsigned j0 = 0; j0 < nm; j0 += 64) {
			const unsigned j = j0 + lane;
			const unsigned long long v = j < nm ? ldx(&mk[j]) : ~0ull;
			const bool in = j < nm && (unsigned)(v >> 32) < dist;
			const unsigned c = in ? t.count_bif((unsigned)v) : 0u;
			r = c > r ? c : r;
			if (!__all(in)) break;                                      // marks are in step order
		}
#pragma unroll
		for (int dd = 32; dd > 0; dd >>= 1) { const unsigned v = __shfl_xor(r, dd); r = v > r ? v : r; }
		res[q] = r;
	}
	WSYNC();
	if (lane == 0) { w.mres[0] = res[0]; w.mres[1] = res[1]; w.mready = true; }
	WSYNC();
}

// ---- marks-only window scan, one LANE per instance (64 instances in flight): what AnyBulges needs of a window -- mark at step 0,
// character at step k, length, the marked steps -- and nothing else (bt_scan_instance with lite set, minus the element cache).
// For ids with many instances: a lane walks its window with dependent loads, but 64 windows advance together, where the
// wave-cooperative scan spends a memory round trip or more on every single window.  Also writes the endChar.
__device__ __forceinline__ void lane_scan_marks(const GraphView &g, const BulgeWork &w, unsigned i)
{
	const unsigned packed = w.start[i], dir = packed & 1u, k = g.k, ws = w.ws;
	unsigned e = w.sel[i], nm = 0, s = 0;
	char ck = ' ';
	unsigned long long *mk = reinterpret_cast<unsigned long long *>(w.wmk) + (size_t)i * w.mks;
	for (; s < ws; s++) {
		const uint8_t c = g.ch[e];
		const unsigned b = g.bif[dir][e];
		if (s == 0) w.wst[i] = b;
		if (s == k) ck = dir ? bt_comp((char)c) : (char)c;
		if (c == BT_SEP) break;
		if (s && b != BT_NONE) { if (nm < w.mks) mk[nm] = ((unsigned long long)s << 32) | b; nm++; }
		e = dir ? g.pv[e] : g.nx[e];
	}
	w.wlen[i] = s; w.wmn[i] = nm; w.wck[i] = ck;
	w.endc[i] = s >= k + 1 ? ck : ' ';                                 // bt_end_chars
}

// ---- AnyBulges with 64 lanes (writer pass) --------------------------------------------------------------------
// bt_any_bulges looks every mark of every window up in the Boost-ordered map; for homologous instances nearly all of
// those look-ups change nothing (the id has an entry with the same endChar).  Here the lanes classify 64 marks at a time
// against a small shadow table (id -> entry, endChar) and only the marks that DO something -- a new entry, or the first
// entry with a different endChar, which also ends the instance -- reach lane 0, in the same order as in the serial loop.
// A first pass counts the distinct ids so that the map is sized by them (it then usually fits the LDS scratch) instead of
// by the total number of marks.  Falls back to bt_any_bulges when the tables do not fit.
struct ABShared { unsigned *skey, *sval; unsigned bits, distinct; int mode; unsigned batch[64]; };   // mode 0: serial fallback, 1: wave path

// The lane-0 part of the map-building pass (logged insertions of ABuild::lazy, bulge_txn.h: bt_ab_insert / bt_ab_append), with the
// fields of the build hoisted out of the loop and, where every array is in LDS (<true>), DS instead of FLAT accesses: as calls of
// bt_ab_insert each of the ~18 insertions of a typical id re-loaded a dozen pointers and counters of the structure through generic
// pointers, 2 - 3 k cycles apiece -- most of the "rb_begin" phase of a transaction.
// what: 1 = `run` new ids of instance i (sh.batch), 2 = instance i joins entry kt.  Returns the new sh.mode (> 0: fine).
template <bool L>
__device__ __forceinline__ int ab_lazy_lane0(Txn &t, BulgeWork &w, ABShared &sh, unsigned what, unsigned i, char ec, unsigned run, unsigned kt_join,
                                             unsigned slots, unsigned shift, bool estimate)
{
	ABuild &a = w.abb;
	unsigned *key = a.m.key, *mhead = a.mhead, *mtail = a.mtail, *mcnt = a.mcnt, *log_inst = a.log_inst, *log_next = a.log_next, *skey = sh.skey, *sval = sh.sval;
	char *echar = a.echar;
	BT_ASSUME_LDS(L, key); BT_ASSUME_LDS(L, mhead); BT_ASSUME_LDS(L, mtail); BT_ASSUME_LDS(L, mcnt); BT_ASSUME_LDS(L, log_inst); BT_ASSUME_LDS(L, log_next);
	BT_ASSUME_LDS(L, skey); BT_ASSUME_LDS(L, sval); BT_ASSUME_LDS(L, echar);
	unsigned size = a.m.size, nlog = a.nlog;
	const unsigned cap = a.m.cap, logcap = a.logcap, distinct = sh.distinct;
	int mode = sh.mode;
	if (what == 2u) {
		if (nlog >= logcap) { t.err |= BT_ERR_SCRATCH; return -1; }
		log_inst[nlog] = i; log_next[nlog] = BT_NONE;
		log_next[mtail[kt_join]] = nlog; mtail[kt_join] = nlog++; mcnt[kt_join]++;
		a.any = true; a.nlog = nlog;
		return mode;
	}
	for (unsigned x = 0; x < run; x++) {
		const unsigned bb = sh.batch[x];
		unsigned hh = (bb * 2654435761u) >> shift;
		unsigned kk = skey[hh];
		while (kk != BT_NONE && kk != bb) { hh = (hh + 1) & (slots - 1); kk = skey[hh]; }
		if (kk == bb) continue;                                          // the id occurs twice in this window: second look-up finds the entry just made
		if (estimate && size >= distinct) { mode = -2; break; }            // more distinct ids than estimated: again, with the counting pass
		if (size >= cap || nlog >= logcap) { t.err |= BT_ERR_SCRATCH; mode = -1; break; }
		const unsigned kt = size++;
		key[kt] = bb; echar[kt] = ec;
		log_inst[nlog] = i; log_next[nlog] = BT_NONE;
		mhead[kt] = nlog; mtail[kt] = nlog; mcnt[kt] = 1; nlog++;
		skey[hh] = bb; sval[hh] = (kt << 8) | (unsigned char)ec;
	}
	a.m.size = size; a.nlog = nlog;
	return mode;
}


#define AB_COUNT_SLOTS 512u
// count_slots: size of the distinct-id counting set (a power of two >= AB_COUNT_SLOTS; the dense kernel has room for more)
__device__ __forceinline__ int wave_any_bulges(const GraphView &g, Txn &t, BulgeWork &w, ABShared &sh, unsigned lane, bool endc_ready = false,
                                               const unsigned count_slots = AB_COUNT_SLOTS, unsigned *count_tab = nullptr /* caller's own table of count_slots words */,
                                               int prof = 0)
{
	PC_T0();
	const unsigned D = g.D, n = w.n;
	// (the window summaries through explicit address spaces -- ldx: DS or global instead of FLAT, see the top of this file)
	char *const endc = w.endc; const char *const wck = w.wck;
	const unsigned *const wst = w.wst, *const wlen = w.wlen, *const wmn = w.wmn;
	const unsigned long long *const wmk = reinterpret_cast<const unsigned long long *>(w.wmk);
	const unsigned mks = w.mks;
	const unsigned cshift = 32u - (unsigned)__builtin_ctz(count_slots);
	unsigned mark = 0, amark = 0;
	if (!endc_ready) {                                                     // bt_end_chars, one instance per lane
		for (unsigned i = lane; i < n; i += 64) stx(&endc[i], ldx(&wlen[i]) >= g.k + 1 ? ldx(&wck[i]) : ' ');
		WSYNC();
	}
	if (lane == 0) { mark = t.fscr_used; amark = t.scr_used; }
	for (int attempt = 0;; attempt++) {                                    // (a second attempt only after an estimate that was too low, see below)
	if (lane == 0) {
		t.fscr_used = mark; t.scr_used = amark;
		sh.skey = count_tab ? count_tab : (unsigned *)t.falloc(count_slots * 4);
		sh.mode = sh.skey ? 1 : 0;
	}
	WSYNC();
	// Ids with dozens of instances (many strains): the counting pass is a walk over all their marks of its own.  Homologous instances
	// reach the same ids, so the number of distinct ids is estimated from the longest mark list instead (x 2 + 32: a second endChar class
	// and strain-specific marks); the map-building pass counts what it really inserts and starts over WITH the counting pass if
	// the estimate was too low (ids of low-complexity sequence, whose instances are not homologous; never on the 62-strain workload).
	// (Round 4's first version let the overflow surface as a scratch error: the id was sent to the big arena, overflowed there again,
	// was sent again ... -- the `-s far` hierarchy case of the drop-in tests never came back.)
	const bool estimate = attempt == 0 && n > 32u && g.ab_estimate;
	if (sh.mode && estimate) {
		unsigned mx = 0;
		for (unsigned i = lane; i < n; i += 64) { const unsigned v = ldx(&endc[i]) == ' ' ? 0u : ldx(&wmn[i]); mx = v > mx ? v : mx; }
#pragma unroll
		for (int dd = 32; dd > 0; dd >>= 1) { const unsigned v = __shfl_xor(mx, dd); mx = v > mx ? v : mx; }
		WSYNC();
		if (lane == 0) {
			t.fscr_used = mark;
			const unsigned distinct = 2 * mx + 32;
			unsigned bits = 6;
			while ((1u << bits) < 2 * distinct + 2) bits++;
			sh.bits = bits; sh.distinct = distinct;
			sh.skey = (unsigned *)t.alloc2((2u << bits) * 4);
			sh.sval = sh.skey ? sh.skey + (1u << bits) : nullptr;
			if (!sh.skey || !bt_ab_prepare(t, w, distinct, g.lazy_map != 0)) sh.mode = -1;
		}
		WSYNC();
	} else if (sh.mode) {
		// ---- pass 1: number of distinct ids that can get an entry
		for (unsigned i = lane; i < count_slots; i += 64) sh.skey[i] = BT_NONE;
		WSYNC();
		unsigned distinct = 0;
		bool full = false;
		for (unsigned i = 0; i < n && !full; i++) {
			if (ldx(&endc[i]) == ' ') continue;
			const unsigned long long *mk = wmk + (size_t)i * mks;
			const unsigned wl = ldx(&wlen[i]);
			const unsigned start = ldx(&wst[i]), lim = wl < D ? wl : D, nm = ldx(&wmn[i]);
			for (unsigned j0 = 0; j0 < nm; j0 += 64) {
				unsigned j = j0 + lane;
				unsigned long long v = j < nm ? ldx(&mk[j]) : ~0ull;
				unsigned b = (unsigned)v;
				bool stop = j >= nm || (unsigned)(v >> 32) >= lim || b == start;
				unsigned long long ms = __ballot(stop);
				unsigned upto = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
				if (distinct + upto > (count_slots * 3) / 4) { full = true; break; }
				bool fresh = false;
				if (lane < upto) {
					unsigned h = (b * 2654435761u) >> cshift;
					for (;;) {
						unsigned old = atomicCAS(&sh.skey[h], BT_NONE, b);
						if (old == BT_NONE || old == b) { fresh = old == BT_NONE; break; }
						h = (h + 1) & (count_slots - 1);
					}
				}
				distinct += (unsigned)__popcll(__ballot(fresh));
				if (upto < 64) break;
			}
		}
		WSYNC();
		if (lane == 0) {
			t.fscr_used = mark;                                        // the counting set is done
			if (full) sh.mode = 0;
			else {
				unsigned bits = 6;
				while ((1u << bits) < 2 * distinct + 2) bits++;
				sh.bits = bits; sh.distinct = distinct;
				sh.skey = (unsigned *)t.alloc2((2u << bits) * 4);
				sh.sval = sh.skey ? sh.skey + (1u << bits) : nullptr;
				if (!sh.skey || !bt_ab_prepare(t, w, distinct, g.lazy_map != 0)) sh.mode = -1;      // log the insertions, build the Boost map only if the call has >= 2 groups (bulge_txn.h: ABuild::lazy)
			}
		}
		WSYNC();
	}
	PC_ADD(16);
	if (sh.mode < 0) return 0;                                             // t.err is set
	if (sh.mode == 0) {                                                    // tables do not fit: one thread, map sized by the total number of marks
		if (lane == 0) sh.mode = bt_any_bulges(t, w, false) ? 3 : 2;
		WSYNC();
		return sh.mode == 3;
	}
	// ---- pass 2: build the map; lanes skip what changes nothing
	const unsigned slots = 1u << sh.bits, shift = 32 - sh.bits;
	for (unsigned i = lane; i < slots; i += 64) { stx(&sh.skey[i], BT_NONE); stx(&sh.sval[i], BT_NONE); }
	WSYNC();
	bool bad = false;
	// (the first 64 marks of the NEXT instance are requested while this one is worked on: with dozens of instances the lists live in the
	// arena, and every instance used to begin with a memory round trip of its own)
	unsigned long long vpre = ~0ull;
	unsigned pre_i = n;
	auto first_chunk = [&](unsigned ii) { const unsigned long long *m0 = wmk + (size_t)ii * mks; return lane < ldx(&wmn[ii]) ? ldx(&m0[lane]) : ~0ull; };
	for (unsigned i = 0; i < n && !bad; i++) {
		const char ec = ldx(&endc[i]);
		if (ec == ' ') continue;
		const unsigned long long *mk = wmk + (size_t)i * mks;
		const unsigned wl = ldx(&wlen[i]);
		const unsigned start = ldx(&wst[i]), lim = wl < D ? wl : D, nm = ldx(&wmn[i]);
		const unsigned long long v0 = pre_i == i ? vpre : first_chunk(i);
		if (i + 1 < n) { vpre = first_chunk(i + 1); pre_i = i + 1; }
		unsigned pos = 0;
		while (pos < nm) {
			unsigned j = pos + lane;
			unsigned long long v = pos == 0 ? v0 : j < nm ? ldx(&mk[j]) : ~0ull;
			unsigned b = (unsigned)v;
			bool stop = j >= nm || (unsigned)(v >> 32) >= lim || b == start;
			unsigned long long ms = __ballot(stop);
			unsigned upto = ms ? (unsigned)__builtin_ctzll(ms) : 64u;
			unsigned ev = 0, val = BT_NONE, h = 0;
			if (lane < upto) {
				h = (b * 2654435761u) >> shift;
				const unsigned *const skey = sh.skey, *const sval = sh.sval;
				for (;;) {
					unsigned kk = ldx(&skey[h]);
					if (kk == b) { val = ldx(&sval[h]); ev = (char)(val & 0xFFu) != ec ? 2u : 0u; break; }
					if (kk == BT_NONE) { ev = 1u; break; }                 // no entry yet (h = where the shadow entry goes)
					h = (h + 1) & (slots - 1);
				}
			}
			unsigned long long em = __ballot(ev != 0);
			if (!em) { if (upto < 64) break; pos += 64; continue; }
			unsigned f = (unsigned)__builtin_ctzll(em);
			unsigned eev = __shfl(ev, f), evl = __shfl(val, f);
			// a run of consecutive new ids (the first instance of every endChar brings ~all its marks) is handed to lane 0 at once
			unsigned long long ins = __ballot(ev == 1u) >> f;
			unsigned run = eev == 1u ? (ins == ~0ull ? 64u - f : (unsigned)__builtin_ctzll(~ins)) : 0u;
			if (lane >= f && lane < f + run) sh.batch[lane - f] = b;
			WSYNC();
			if (lane == 0 && w.abb.lazy) {
				const bool lds = BT_IS_LDS(w.abb.m.key) && BT_IS_LDS(sh.skey);      // (one allocation decision for all arrays of the build, bt_ab_prepare)
				const unsigned what = eev == 1u ? 1u : 2u;
				sh.mode = lds ? ab_lazy_lane0<true>(t, w, sh, what, i, ec, run, evl >> 8, slots, shift, estimate)
				              : ab_lazy_lane0<false>(t, w, sh, what, i, ec, run, evl >> 8, slots, shift, estimate);
			} else if (lane == 0) {
				if (eev == 1u) {
					for (unsigned x = 0; x < run && sh.mode > 0; x++) {
						unsigned bb = sh.batch[x], hh = (bb * 2654435761u) >> shift;
						while (sh.skey[hh] != BT_NONE && sh.skey[hh] != bb) hh = (hh + 1) & (slots - 1);
						if (sh.skey[hh] == bb) continue;                       // the id occurs twice in this window: second look-up finds the entry just made
						int kt = estimate && w.abb.m.size >= sh.distinct ? -2 : bt_ab_insert(t, w, i, bb);
						if (kt == -2) sh.mode = -2;                               // more distinct ids than estimated: again, with the counting pass
						else if (kt < 0) sh.mode = -1;
						else { sh.skey[hh] = bb; sh.sval[hh] = ((unsigned)kt << 8) | (unsigned char)ec; }
					}
				} else if (!bt_ab_append(t, w, i, (int)(evl >> 8))) sh.mode = -1;
			}
			WSYNC();
			if (sh.mode < 0) { bad = true; break; }
			if (eev == 2u) break;                                          // the instance joined a group: next instance
			pos += f + run;
		}
	}
	if (bad && sh.mode == -2) { WSYNC(); continue; }
	if (bad) return 0;
	PC_ADD(17);
	if (lane == 0) sh.mode = bt_ab_finish(t, w) ? 3 : 2;
	WSYNC();
	PC_ADD(18);
	return sh.mode == 3;
	}
}


// One wave per window entry: ownership check on the claim list (64 lanes), then RemoveBulges with lane 0 taking
// the decisions on the cached windows and all lanes rescanning them after every collapse.
#ifndef COMMIT_FAST_BYTES
#define COMMIT_FAST_BYTES 8192               // LDS scratch of a transaction; with Txn / BulgeWork ~9 KB per workgroup = 17 workgroups per CU (12 KB: 12, and 4 % slower)
#endif
// The transaction proper (RemoveBulges for one id) on one wave; t, w, flag, absh and fast live in LDS.
// solo: 0 = ordered round (the probe found bulges, the entry owns its claims), 1 = the id runs with nothing else in flight
// (big-arena solo round, or the serial chain: stampv == BT_NONE, no reservation exists and none is checked).
__device__ __forceinline__ void commit_body(const GraphView &g, Txn &t, BulgeWork &w, int &flag, ABShared &absh, uint8_t *fast, unsigned fast_bytes,
                                            unsigned wi, unsigned id, unsigned stampv, int solo, bool prepass, uint8_t *mine, unsigned arena_bytes, int prof,
                                            const unsigned *sepl = nullptr /* LDS copy of the separators' slots (SepBounds), or none */)
{
	const unsigned lane = threadIdx.x, tid = id + 1;
	PH_T0();
	// ---- the probe of this round found bulges (solo entries were not probed: verdict pass first)
	if (lane == 0) { g.need[id] = 0; g.touch[id] = 1; flag = 1; }
	if (prepass) {
		if (lane == 0) { t.init(g, id, wi, 1, mine, arena_bytes); t.ext_stamps = true; t.chain = stampv == BT_NONE; }
		WSYNC();
		wave_setup(g, t, w, true, lane, flag);
		if (flag) {
			wave_scan_all(g, w, lane, stampv, tid, 1, id);
			WSYNC();
		}
		int verdict = flag ? wave_verdict(g, w, *reinterpret_cast<VerdictTable *>(fast), lane) : 0;   // the fast scratch is idle in this pass
		if (lane == 0) {
			bool has = verdict > 0;
			if (verdict < 0) { bt_end_chars(t, w); has = bt_any_bulges(t, w, true); }
			if (t.err & BT_ERR_SCRATCH) { atomicOr(&g.ctr[CTR_ERR], BT_ERR_SCRATCH); has = false; }   // does not even fit the big arena
			flag = has ? 1 : 0;
		}
	}
	WSYNC();
	if (lane == 0) { atomicAdd(&g.ctr[CTR_COMMITTED], 1u); atomicAdd(&g.ctr[CTR_TXN], 1u); }
	if (!flag) return;
	// ---- writer pass: reads and writes are published for order validation
	if (lane == 0) { t.init(g, id, wi, 2, mine, arena_bytes); t.chain = stampv == BT_NONE; t.defer_push = true; t.ext_stamps = true; t.fscr = fast; t.fscr_cap = fast_bytes; w.ret = 0;
	                 t.tc_cap = 1024; t.tc_list = (uint32_t *)t.alloc(t.tc_cap * 4); if (!t.tc_list) t.tc_cap = 0; t.err = 0; t.defer_cleanup = true; t.prof = prof != 0; }
	WSYNC();
	wave_setup(g, t, w, false, lane, flag);
	PH_ADD(0);
	if (flag) {
		wave_scan_all(g, w, lane, stampv, tid, 2, id);
		WSYNC();
		if (w.mk_overflow) {                                          // more marks in a window than the LDS lists hold: use the arena
			WSYNC();
			if (lane == 0) bt_marks_to_arena(t, w);
			WSYNC();
			if (!t.err) for (unsigned i = 0; i < w.n; i++) wave_scan_instance(g, w, i, lane, stampv, tid, 2, id);
			WSYNC();
		}
		PH_ADD(1);
		int any = wave_any_bulges(g, t, w, absh, lane, false, AB_COUNT_SLOTS, nullptr, prof);
		// lazy windows (bulge_txn.h: BulgeWork::lazy) when the id is large and has the graph to itself: the set of windows a collapse
		// dirties -- O(instances) to compute, and nearly all of them in the dense regime -- is only needed by the reservation check of
		// an ordered round
		if (lane == 0) { flag = bt_rb_begin(t, w, any) && !t.err ? 1 : 0; w.lazy = solo && w.wep != nullptr; w.jscan = w.lazy || (w.n > 24u && g.jscan_rounds); w.mscan = (w.n > 24u || (g.test_flags & 16u)) && g.jscan_rounds; if (g.test_flags & 16u) w.mscan_min = (g.test_flags >> 8) & 15u;
			                 w.use_stale = !w.lazy && w.n <= 256u && g.lazy_rescan && w.wdel != nullptr; }      // (many strains: groups of dozens of members, the J search with 64 lanes -- wave_next_j)
		WSYNC();
		PH_ADD(2);
		while (flag) {
			if (lane == 0) { const int r = bt_scratch_in_lds(w) ? bt_rb_run<true>(t, w) : bt_rb_run<false>(t, w); flag = t.err ? 0 : r; }      // (<true>: DS instead of FLAT accesses, bulge_txn.h: BT_ASSUME_LDS)
			WSYNC();
			PH_ADD(3);
			if (!flag) break;
			if (flag == 3) { wave_next_j(g, w, lane); continue; }       // large group: the search for the next J, 256 members per step
			if (flag == 4) { wave_mults(g, t, w, lane); continue; }     // branches with many bifurcations inside: their multiplicities, one look-up per lane
			if (flag == 2) {                                             // the loops need these windows as of now
				const unsigned nr = w.nreq;
				for (unsigned x = 0; x < nr; x++) wave_scan_instance(g, w, w.req[x], lane, stampv, tid, 2, id);
				WSYNC();
				if (lane == 0) for (unsigned x = 0; x < nr; x++) { if (w.lazy) w.wep[w.req[x]] = w.epoch; else w.stale[w.req[x] >> 6] &= ~(1ull << (w.req[x] & 63u)); }
				WSYNC();
				if (!w.lazy && w.mk_overflow) {                              // (stale-marking rounds: more marks than the LDS lists hold)
					if (lane == 0) bt_marks_to_arena(t, w);
					WSYNC();
					if (t.err) break;
					for (unsigned i = 0; i < w.n; i++) wave_scan_instance(g, w, i, lane, stampv, tid, 2, id);
					if (lane == 0) w.stale[0] = w.stale[1] = w.stale[2] = w.stale[3] = 0;
					WSYNC();
				}
				PH_ADD(8);
				continue;
			}
			if (w.lazy) {
				wave_collapse_any(g, t, w, lane, stampv, prof);
				PH_ADD(5);
				if (t.err) break;
				wave_publish_collapse(g, id, t.push_e, t.push_d, t.push_len, lane, sepl);
				if (lane == 0) w.epoch++;                                // every cached window is stale until the loops ask for it
				WSYNC();
				PH_ADD(6);
				continue;
			}
			// which cached windows see the region about to be rewritten (target start .. end of its look-forward flank)?
			// only those are rescanned afterwards -- normally just the target's own window
			unsigned long long dirty[4] = {0, 0, 0, 0};                   // up to 256 windows in registers, more in the arena (w.dirty_big)
			const bool big = w.n > 256, selective = !big || w.dirty_big != nullptr;
			if (selective) {
				const unsigned tg = w.c_tgt, span = 2 * g.k + w.c_dT + 1;
				if (w.use_stale && (w.stale[0] | w.stale[1] | w.stale[2] | w.stale[3])) {
					// stale windows the collapse might reach (their old reach + what was deleted inside it since, or a walk with link breaks)
					// are brought up to date FIRST: the test below then only ever sees fresh summaries, exactly as with eager rescans
					bool any = false;
					for (unsigned i0 = 0; i0 < w.n; i0 += 64) {
						const unsigned i = i0 + lane;
						bool f = false;
						if (i < w.n && ((w.stale[i >> 6] >> (i & 63u)) & 1ull)) {
							const unsigned len = (w.wlen[i] + 1 < w.ws ? w.wlen[i] + 1 : w.ws) + w.wdel[i] + (w.c_dT > w.c_dS ? w.c_dT - w.c_dS : 0u);
							const unsigned tl = w.wlen[tg] + 1 < w.ws ? w.wlen[tg] + 1 : w.ws;
							f = w.wnb[i] != 0 || bt_windows_intersect(w, i, len, tg, span < tl ? span : tl) != 0;
						}
						unsigned long long fresh = __ballot(f);
						if (!fresh) continue;
						any = true;
						WSYNC();
						if (lane == 0) w.stale[i0 >> 6] &= ~fresh;
						for (; fresh; fresh &= fresh - 1) wave_scan_instance(g, w, i0 + (unsigned)__builtin_ctzll(fresh), lane, stampv, tid, 2, id);
						WSYNC();
					}
					if (any && w.mk_overflow) {
						if (lane == 0) bt_marks_to_arena(t, w);
						WSYNC();
						if (t.err) break;
						for (unsigned i = 0; i < w.n; i++) wave_scan_instance(g, w, i, lane, stampv, tid, 2, id);
						if (lane == 0) w.stale[0] = w.stale[1] = w.stale[2] = w.stale[3] = 0;
						WSYNC();
					}
				}
				for (unsigned i0 = 0; i0 < w.n; i0 += 64) {
					unsigned i = i0 + lane;
					bool d = false;
					if (i < w.n) {
						unsigned len = w.wlen[i] + 1 < w.ws ? w.wlen[i] + 1 : w.ws;       // cached steps incl. the separator step
						unsigned tl = w.wlen[tg] + 1 < w.ws ? w.wlen[tg] + 1 : w.ws;
						const bool st = w.use_stale && ((w.stale[i >> 6] >> (i & 63u)) & 1ull);      // (still stale = provably out of reach, see above)
						d = i == tg || (!st && bt_windows_intersect(w, i, len, tg, span < tl ? span : tl) != 0);
					}
					const unsigned long long bits = __ballot(d);
					if (!big) dirty[i0 >> 6] = bits; else if (lane == 0) w.dirty_big[i0 >> 6] = bits;
				}
			}
			PH_ADD(4);
			wave_collapse_any(g, t, w, lane, stampv, prof);
			PH_ADD(5);
			if (t.err) break;
			if (lane == 0 && w.c_dT > w.c_dS) {
				// deletions shift what a window of fixed step count reaches: stay inside the reserved range or run alone
				const unsigned F = 2 * (g.D + g.k + 2) + g.k, del = w.c_dT - w.c_dS;
				bool escape = !selective;
				for (unsigned i = 0; i < w.n && selective; i++)
					if (((big ? w.dirty_big[i >> 6] : dirty[i >> 6]) >> (i & 63)) & 1ull) {
						w.wdel[i] += del;
						if (g.D + g.k + 2 + w.wdel[i] > F || (g.D - 1) + 3 * g.k + g.D + 2 + w.wdel[i] > F + g.D - 1 - w.c_dS) escape = true;
					}
				if (escape && !solo) { g.big[id] = 1; atomicMin(&g.ctr[CTR_VIOL], id); }     // replay with this id running alone
			}
			wave_publish_collapse(g, id, t.push_e, t.push_d, t.push_len, lane, sepl);
			PH_ADD(6);
			PH_ADD(7);
			if (w.use_stale) {                                              // marked, not rescanned: whoever reads one of them next asks for it (bt_rb_run returns 2)
				if (lane == 0) for (unsigned q = 0; q < 4; q++) w.stale[q] |= dirty[q];
				WSYNC();
				PH_ADD(8);
				continue;
			}
			for (unsigned i = 0; i < w.n; i++)
				if (!selective || (((big ? w.dirty_big[i >> 6] : dirty[i >> 6]) >> (i & 63)) & 1ull)) wave_scan_instance(g, w, i, lane, stampv, tid, 2, id);
			WSYNC();
			if (w.mk_overflow) {
				WSYNC();
				if (lane == 0) bt_marks_to_arena(t, w);
				WSYNC();
				if (t.err) break;
				for (unsigned i = 0; i < w.n; i++) wave_scan_instance(g, w, i, lane, stampv, tid, 2, id);
				WSYNC();
			}
			PH_ADD(8);
		}
	}
	// ---- Cleanup (bifurcationstorage.cpp:33-41) once the loops are over: the erased nodes leave their lists' sizes, all lanes
	WSYNC();
	if (!t.err && t.tc_n) {
		if (t.tc_n <= t.tc_cap) {
			for (unsigned x = lane; x < t.tc_n; x += 64) { unsigned v = g.nidst[t.tc_list[x]]; atomicSub(&g.lsize[v & 1u][v >> 1], 1u); }
		} else if (lane == 0) t.cleanup();                              // more erased nodes than the flat list holds: walk the chain
	}
	if (lane == 0) {
		if (prof) {
			unsigned long long dur = __builtin_readcyclecounter() - ph_start;
			unsigned bin = 0;
			while (bin < 15 && (dur >> (13 + bin))) bin++;
			atomicAdd(&g_txn_hist[w.ret < 3 ? w.ret : 3][bin], 1ull);
			if (atomicMax(&g_txn_max[0], dur) < dur) g_txn_max[1] = ((unsigned long long)w.n << 32) | w.ret;
			{
				const unsigned long long now = wall_clock64(), sl = (g.tslot >> 2) & 4095u;
				atomicMin(&g_round_span[3 * sl], ph_wall); atomicMax(&g_round_span[3 * sl + 1], now);
				atomicMax(&g_round_span[3 * sl + 2], ((now - ph_wall) << 32) | (ph_wall & 0xFFFFFFFFull));
				if (w.ret <= 1) atomicMax(&g_round_few[2 * sl], now - ph_wall);
				if (w.ret <= 2) atomicMax(&g_round_few[2 * sl + 1], now - ph_wall);
			}
			atomicMax(&g_round_max[(g.tslot >> 2) & 4095u], (dur << 24) | ((unsigned long long)(w.n < 255u ? w.n : 255u) << 16) | ((unsigned long long)(w.nold < 255u ? w.nold : 255u) << 8) | (w.ret < 255u ? w.ret : 255u));
		}
		if (t.err) {
			if (!t.wrote && t.err == BT_ERR_SCRATCH) { ss_mark_big(g, id); return; }
			atomicOr(&g.ctr[CTR_ERR], t.err);
		}
		atomicAdd(&g.ctr[CTR_BULGES], w.ret);
	}
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_commit(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, int solo, const unsigned *claims, const uint8_t *live, int prof)
{
	__shared__ Txn t;
	__shared__ BulgeWork w;
	__shared__ int flag;
	__shared__ ABShared absh;
	__shared__ __attribute__((aligned(16))) uint8_t fast[COMMIT_FAST_BYTES];     // window summaries, mark lists, FillVisit list and AnyBulges map of typical ids
	const unsigned wi = blockIdx.x, lane = threadIdx.x;
	round_stamp(g, 2);
	if (wi >= nwin) return;
	if (!solo && !live[wi]) return;                                   // retired by the probe
	const unsigned id = g.win[wi], stampv = g.round_bits | wi;
	if (!solo) {
		const unsigned *cb = claims + (size_t)wi * (CLAIM_CAP + 1);
		unsigned n = cb[0];
		bool owner = true;
		if (n <= CLAIM_CAP) {
			for (unsigned i = lane; i < n; i += 64) {
				unsigned b = cb[1 + i];
				if (b & 0x80000000u) { if (bt_order_blocked(g, b & 0x7FFFFFFFu)) owner = false; }     // something at or below a lower id of the surroundings is about to run
				else if (g.own[b] != stampv) owner = false;
			}
			owner = !__any(!owner);
		} else {
			if (lane == 0) owner = ss_owns_footprint(g, wi);      // list overflowed: serial re-walk
			owner = __shfl((int)owner, 0) != 0;
		}
		if (!owner) return;                                       // stays pending
	}
	__shared__ unsigned s_sep[64];                                    // the separators' slots (SepBounds), when there are at most 64
	const unsigned *sepl = g.sep && g.nsep <= 64 ? s_sep : nullptr;
	if (sepl) s_sep[lane] = lane < g.nsep ? g.sep[lane] : BT_NONE;
	commit_body(g, t, w, flag, absh, fast, (unsigned)sizeof fast, wi, id, stampv, solo, solo != 0, arena + (size_t)wi * arena_bytes, arena_bytes, prof, sepl);
}

// Serial chain: one wave runs what is pending in the id range of the window strictly in ascending order, one transaction
// after the other, with nothing else in flight -- the sequential order itself, so no reservation.  The driver switches to
// it when the ordered rounds stop being parallel (dense conflict neighbourhoods: small k, low-complexity sequence), where a
// round costs four launches and commits one or two transactions.  The probe has already retired the clean entries (need
// = 0) and marked the live ones (need = 2: no verdict pass needed); the first id made pending by the chain itself (need = 1)
// ends the stretch -- the next round's probe takes those verdicts in parallel.  Stops at the first error or order violation.
__global__ void __launch_bounds__(64) k_chain(GraphView g, uint8_t *arena, unsigned arena_bytes, unsigned nwin, int prof)
{
	__shared__ Txn t;
	__shared__ BulgeWork w;
	__shared__ int flag;
	__shared__ ABShared absh;
	__shared__ __attribute__((aligned(16))) uint8_t fast[12288];
	const unsigned lane = threadIdx.x;
	round_stamp(g, 2);
	if (!nwin) return;
	const unsigned long long limit = g.win[nwin - 1];
	unsigned long long cur = g.win[0];
	unsigned done = 0;
	while (cur <= limit) {
		// next pending id at or after cur: 64 lanes x 8 flags
		const unsigned long long base = cur & ~7ull, idq = base + 8ull * lane;
		unsigned long long nb = 0;
		if (idq <= limit) {
			nb = *reinterpret_cast<const unsigned long long *>(g.need + idq);
#pragma unroll
			for (int j = 0; j < 8; j++) if (idq + j < cur || idq + j > limit) nb &= ~(0xFFull << (8 * j));
		}
		unsigned long long hit = __ballot(nb != 0);
		if (!hit) { cur = base + 512; continue; }
		unsigned src = (unsigned)__builtin_ctzll(hit);
		unsigned long long nbs = __shfl(nb, src);
		const unsigned byte = (unsigned)__builtin_ctzll(nbs) >> 3;
		const unsigned id = (unsigned)(base + 8ull * src + byte);
		const bool known_live = ((nbs >> (8 * byte)) & 0xFFull) == 2ull;
		// an id made pending by the chain itself ends the stretch: its verdict is taken by the next (parallel) probe
		if (!known_live && done) break;
		done++;
		WSYNC();
		if (lane == 0) g.big[id] = 0;                                   // the chain always runs in the big arena
		commit_body(g, t, w, flag, absh, fast, (unsigned)sizeof fast, 0u, id, BT_NONE, 1, !known_live, arena, arena_bytes, prof);
		WSYNC();
		cur = (unsigned long long)id + 1;
		__threadfence();
		unsigned stop = lane == 0 ? (g.ctr[CTR_ERR] != 0 || g.ctr[CTR_VIOL] != BT_NONE || g.big[id] != 0) : 0u;   // big: did not even fit the big arena
		if (__shfl((int)stop, 0)) break;
	}
}

// ---- tiny / dense inputs: the whole SimplifyGraph in ONE launch -----------------------------------------------------------------
// for iteration: for id ascending: RemoveBulges(id) (reference src/blockfinder.cpp:29-43), literally: one wave walks the ids in order
// with nothing else in flight -- no snapshot, no probe, no reservation, no stamps, no checkpoint.  This is for inputs whose whole
// graph is a few thousand elements (the host chooses it by size, sbl_simplify_run): there the ordered rounds have nothing to run in
// parallel -- low-complexity sequence at k = 3 .. 10 makes every element a bifurcation, ids have thousands of instances, and every
// transaction conflicts with every other -- and what counts is the cost of ONE RemoveBulges call:
//   * one ListPositions and one pass over the windows per call (the round machinery examines a pending id three times: probe,
//     verdict pass, writer pass), marks only, one LANE per instance (lane_scan_marks);
//   * lazy windows (BulgeWork::lazy): FillVisit / Overlap / MaxBifurcationMultiplicity read the windows of I and J only, scanned
//     with 64 lanes when the loops ask for them, as the reference walks them when it needs them; a collapse costs two window scans,
//     not a pass over thousands of cached windows;
//   * the J search of large bulge groups with 64 lanes (wave_next_j).
// Capacity errors (element / node pool, arena) stop the kernel; the host then reruns the stage through the ordered rounds, which
// can grow their pools and replay.
#define DENSE_FAST_BYTES 40960u
#define DENSE_COUNT_SLOTS 4096u             // distinct ids of one AnyBulges map counted in LDS (16 KB of the scratch)
#define DENSE_LANE_SCAN_MIN 24u             // instances from which the marks-only scan runs one lane per instance
__device__ __forceinline__ void dense_remove_bulges(const GraphView &g, Txn &t, BulgeWork &w, int &flag, ABShared &absh, uint8_t *fast, unsigned id,
                                                    uint8_t *arena, unsigned arena_bytes)
{
	const unsigned lane = threadIdx.x;
	unsigned *count_tab = reinterpret_cast<unsigned *>(fast);              // the first 16 KB of the scratch: AnyBulges' counting set
	if (lane == 0) {
		t.init(g, id, 0, 0, arena, arena_bytes);                           // mode 0: nothing to validate against
		t.chain = true; t.defer_push = true; t.ext_stamps = true; w.ret = 0;
		t.fscr = fast + DENSE_COUNT_SLOTS * 4; t.fscr_cap = DENSE_FAST_BYTES - DENSE_COUNT_SLOTS * 4;
		t.tc_cap = 4096; t.tc_list = (uint32_t *)t.alloc(t.tc_cap * 4); if (!t.tc_list) t.tc_cap = 0; t.err = 0; t.defer_cleanup = true;
	}
	WSYNC();
	wave_setup(g, t, w, false, lane, flag);
	if (flag) {
		if (lane == 0) w.epoch = 1;                                        // wep[] = 0: no window has been scanned in full yet
		WSYNC();
		if (w.n >= DENSE_LANE_SCAN_MIN) {
			for (unsigned i0 = 0; i0 < w.n; i0 += 64) if (i0 + lane < w.n) lane_scan_marks(g, w, i0 + lane);
		} else {
			wave_scan_all(g, w, lane, BT_NONE, 0, 0, id);
			WSYNC();
			for (unsigned i = lane; i < w.n; i += 64) w.wep[i] = 1;
			if (lane == 0) bt_end_chars(t, w);
		}
		WSYNC();
		int any = wave_any_bulges(g, t, w, absh, lane, true, DENSE_COUNT_SLOTS, count_tab);
		if (lane == 0) { flag = bt_rb_begin(t, w, any) && !t.err ? 1 : 0; w.lazy = true; w.jscan = true; }
		WSYNC();
		while (flag) {
			if (lane == 0) { const int r = bt_rb_run(t, w); flag = t.err ? 0 : r; }
			WSYNC();
			if (!flag) break;
			if (flag == 3) { wave_next_j(g, w, lane); continue; }
			if (flag == 2) {
				const unsigned nr = w.nreq;
				for (unsigned x = 0; x < nr; x++) wave_scan_instance(g, w, w.req[x], lane, BT_NONE, 0, 0, id);
				WSYNC();
				if (lane == 0) for (unsigned x = 0; x < nr; x++) w.wep[w.req[x]] = w.epoch;
				WSYNC();
				continue;
			}
			wave_collapse(g, t, w, lane, BT_NONE);
			if (t.err) break;
			if (lane == 0) w.epoch++;
			WSYNC();
		}
	}
	// ---- Cleanup (bifurcationstorage.cpp:33-41)
	WSYNC();
	if (!t.err && t.tc_n) {
		if (t.tc_n <= t.tc_cap) {
			for (unsigned x = lane; x < t.tc_n; x += 64) { unsigned v = g.nidst[t.tc_list[x]]; atomicSub(&g.lsize[v & 1u][v >> 1], 1u); }
		} else if (lane == 0) t.cleanup();
	}
	if (lane == 0) {
		if (t.err) atomicOr(&g.ctr[CTR_ERR], t.err);
		if (w.n >= 2) { atomicAdd(&g.ctr[CTR_BULGES], w.ret); atomicAdd(&g.ctr[CTR_TXN], 1u); }
	}
}

// out: [0] iterations run, [1] ids examined in the last iteration (progress)
__global__ void __launch_bounds__(64) k_dense_stage(GraphView g, uint8_t *arena, unsigned arena_bytes, unsigned max_iter, unsigned *out)
{
	__shared__ Txn t;
	__shared__ BulgeWork w;
	__shared__ int flag;
	__shared__ ABShared absh;
	__shared__ __attribute__((aligned(16))) uint8_t fast[DENSE_FAST_BYTES];
	const unsigned lane = threadIdx.x;
	unsigned iter = 0, total = 0;
	bool stop = false;
	do {
		iter++;
		for (unsigned id = 0; id < g.nid && !stop; id++) {
			if (g.lsize[0][id] + g.lsize[1][id] < 2) continue;              // ListPositions < 2: nothing to do (bulgeremoval.cpp:336-337)
			WSYNC();
			dense_remove_bulges(g, t, w, flag, absh, fast, id, arena, arena_bytes);
			WSYNC();
			__threadfence();                                               // list sizes / marks changed through atomics: later plain loads must see them
			stop = __shfl((int)(lane == 0 ? *(volatile unsigned *)&g.ctr[CTR_ERR] : 0u), 0) != 0;
		}
		__threadfence();
		total = (unsigned)__shfl((int)(lane == 0 ? *(volatile unsigned *)&g.ctr[CTR_BULGES] : 0u), 0);
	} while (!stop && total > 0 && iter < max_iter);                        // `total` is cumulative (blockfinder.cpp:43)
	if (lane == 0) out[0] = iter;
}

// ids whose windows or lists changed since their verdict was taken (what an incremental snapshot has to look at)
__global__ void __launch_bounds__(256) k_count_touched(const uint8_t *__restrict__ touch, unsigned nid, unsigned *__restrict__ out)
{
	unsigned c = 0;
	for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nid; i += gridDim.x * blockDim.x) c += touch[i] != 0;
#pragma unroll
	for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d);
	if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

// ... and their list (any order: the snapshot takes the same verdict of each), appended a wave at a time
__global__ void __launch_bounds__(256) k_touched_list(const uint8_t *__restrict__ touch, unsigned nid, unsigned *__restrict__ list, unsigned *__restrict__ count)
{
	const unsigned id = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
	const bool t = id < nid && touch[id] != 0;
	const unsigned long long m = __ballot(t);
	if (!m) return;
	unsigned base = 0;
	if (lane == (unsigned)__builtin_ctzll(m)) base = atomicAdd(count, (unsigned)__popcll(m));
	base = __shfl(base, (unsigned)__builtin_ctzll(m));
	if (t) list[base + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = id;
}

// ------------------------------------------------------------------------------------------- copy-back (T3) kernels
// The list is a chain of "segments" = maximal runs of consecutive slots linked consecutively.  Heads are
// found with a flag pass, segments are ranked by pointer jumping, elements scatter to rank + offset.
__global__ void __launch_bounds__(256) k_seg_flags(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, unsigned ne, unsigned *__restrict__ flag)
{
	unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= ne) return;
	bool alive = ch[e] != BT_DEAD_CHAR;
	bool cont = e > 0 && ch[e - 1] != BT_DEAD_CHAR && nx[e - 1] == e;
	flag[e] = alive && !cont ? 1u : 0u;
}
// segidx = inclusive scan of flag.  For every alive tail element: record its segment's tail and successor.
__global__ void __launch_bounds__(256) k_seg_tails(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, unsigned ne,
                                                   const unsigned *__restrict__ flag, const unsigned *__restrict__ segidx /* exclusive scan */,
                                                   unsigned *__restrict__ seg_head, unsigned *__restrict__ seg_len, unsigned *__restrict__ seg_succ_elem)
{
	unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= ne || ch[e] == BT_DEAD_CHAR) return;
	unsigned seg = segidx[e] + flag[e] - 1;            // exclusive scan + own flag - 1 = index of the segment containing e
	if (flag[e]) seg_head[seg] = e;
	bool tail = !(e + 1 < ne && ch[e + 1] != BT_DEAD_CHAR && nx[e] == e + 1);
	if (tail) { seg_len[seg] = e; seg_succ_elem[seg] = nx[e]; }   // seg_len temporarily holds the tail element
}
__global__ void __launch_bounds__(256) k_seg_finish(unsigned nseg, const unsigned *__restrict__ seg_head, unsigned *__restrict__ seg_len,
                                                    const unsigned *__restrict__ seg_succ_elem, const unsigned *__restrict__ flag,
                                                    const unsigned *__restrict__ segidx, unsigned *__restrict__ succ, unsigned long long *__restrict__ dist)
{
	unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	unsigned len = seg_len[s] - seg_head[s] + 1;
	seg_len[s] = len;
	unsigned se = seg_succ_elem[s];
	succ[s] = se == SBL_NONE ? SBL_NONE : segidx[se] + flag[se] - 1;
	dist[s] = len;
}
// Wyllie pointer jumping: dist[s] = total length from s to the end of the chain
__global__ void __launch_bounds__(256) k_seg_jump(unsigned nseg, const unsigned *__restrict__ succ_in, const unsigned long long *__restrict__ dist_in,
                                                  unsigned *__restrict__ succ_out, unsigned long long *__restrict__ dist_out)
{
	unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
	if (s >= nseg) return;
	unsigned n = succ_in[s];
	if (n == SBL_NONE) { succ_out[s] = SBL_NONE; dist_out[s] = dist_in[s]; }
	else { succ_out[s] = succ_in[n]; dist_out[s] = dist_in[s] + dist_in[n]; }
}
__global__ void __launch_bounds__(256) k_scatter_linear(const uint8_t *__restrict__ ch, const unsigned *__restrict__ op, unsigned ne,
                                                        const unsigned *__restrict__ flag, const unsigned *__restrict__ segidx,
                                                        const unsigned *__restrict__ seg_head, const unsigned long long *__restrict__ dist,
                                                        unsigned long long total, uint8_t *__restrict__ ch_out, unsigned *__restrict__ op_out,
                                                        unsigned *__restrict__ newidx)
{
	unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= ne) return;
	if (ch[e] == BT_DEAD_CHAR) { newidx[e] = SBL_NONE; return; }
	unsigned seg = segidx[e] + flag[e] - 1;
	unsigned long long pos = total - dist[seg] + (e - seg_head[seg]);
	ch_out[pos] = ch[e];
	op_out[pos] = op[e] & BT_POS_MASK;
	newidx[e] = (unsigned)pos;
}
__global__ void k_remap_seps(const unsigned *__restrict__ newidx, unsigned *__restrict__ sepidx, unsigned n)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) sepidx[i] = newidx[sepidx[i]];
}
// The separator that ends chromosome c carries the CURRENT length of c as its position: the next stage's DNASequence is built from the
// simplified records and stamps it with record[chr].size() (dnasequence.cpp:96), and Replace clamps interpolated positions to the
// position of the element after the rewritten span -- at a chromosome's end that is this separator.
__global__ void k_sep_positions(const unsigned *__restrict__ sepidx, unsigned nchr, unsigned *__restrict__ op)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nchr) op[sepidx[i + 1]] = (sepidx[i + 1] - sepidx[i] - 1u) & BT_POS_MASK;
}
// IndexedSequence::Test() (reference src/indexedsequence.cpp:74-103, compiled under _DEBUG only): after any number of collapses, at every
// window position the stored mark equals what the dictionary of the INITIAL marking (k-mer string -> id, FormDictionary) says about the
// k characters spelled there NOW -- "same k-mer => same id everywhere" -- and a position whose k-mer is not in the dictionary (or that
// has no full window) carries no mark.  k <= 32: the dictionary is the sorted list of strand-specific bifurcation codes of the stage's
// enumeration (id = rank).  Checked on the stage's final graph: marks by old slot, characters of the copy-back's linear order.
// out: [0] windows checked, [1] mismatches, [2..5] first mismatch (slot, strand, stored, expected).   SBL_CHECK_DICTIONARY=1.
__device__ __forceinline__ unsigned dict_lookup(const unsigned long long *__restrict__ dict, unsigned nd, unsigned long long code)
{
	unsigned lo = 0, hi = nd;
	while (lo < hi) { unsigned mid = (lo + hi) >> 1; if (dict[mid] < code) lo = mid + 1; else hi = mid; }
	return lo < nd && dict[lo] == code ? lo : BT_NONE;
}
__global__ void __launch_bounds__(256) k_dict_check(const uint8_t *__restrict__ ch, unsigned ne, const unsigned *__restrict__ newidx, const uint8_t *__restrict__ ch_out, unsigned long long total,
                                                    const unsigned *__restrict__ bif0, const unsigned *__restrict__ bif1, const unsigned long long *__restrict__ dict, unsigned nd, unsigned k,
                                                    unsigned long long *__restrict__ out)
{
	const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	unsigned checked = 0, bad = 0;
	if (e < ne && ch[e] != BT_DEAD_CHAR && ch[e] != BT_SEP) {
		const unsigned long long p = newidx[e];
		auto base = [](uint8_t c) { unsigned x = (c >> 1) & 3u; return x ^ (x >> 1); };      // A0 C1 G2 T3 (k_pack2bit)
		unsigned exp0 = BT_NONE, exp1 = BT_NONE;
		if (p + k <= total) {
			unsigned long long code = 0; bool full = true;
			for (unsigned i = 0; i < k; i++) { const uint8_t c = ch_out[p + i]; if (c == BT_SEP) { full = false; break; } code = (code << 2) | base(c); }
			if (full) { exp0 = dict_lookup(dict, nd, code); checked++; }
		}
		if (p + 1 >= k) {
			unsigned long long code = 0; bool full = true;
			for (unsigned i = 0; i < k; i++) { const uint8_t c = ch_out[p - i]; if (c == BT_SEP) { full = false; break; } code = (code << 2) | (3u - base(c)); }
			if (full) { exp1 = dict_lookup(dict, nd, code); checked++; }
		}
		const unsigned s0 = bif0[e], s1 = bif1[e];
		if (s0 != exp0) { bad++; if (atomicCAS(&out[2], ~0ull, (unsigned long long)e) == ~0ull) { out[3] = 0; out[4] = s0; out[5] = exp0; } }
		if (s1 != exp1) { bad++; if (atomicCAS(&out[2], ~0ull, (unsigned long long)e) == ~0ull) { out[3] = 1; out[4] = s1; out[5] = exp1; } }
	}
	for (int d = 32; d > 0; d >>= 1) { checked += __shfl_down(checked, d); bad += __shfl_down(bad, d); }
	if ((threadIdx.x & 63) == 0) { if (checked) atomicAdd(&out[0], (unsigned long long)checked); if (bad) atomicAdd(&out[1], (unsigned long long)bad); }
}
__global__ void __launch_bounds__(256) k_fill_bytes(uint8_t *p, uint8_t v, size_t from, size_t to)
{
	size_t i = from + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < to) p[i] = v;
}

// ------------------------------------------------------------------------------------------- device backend
// ---- block index over the original slots (GraphView::bidx) -----------------------------------------------------------------------
// Built once per stage from the arrays (and again after a roll-back); from then on the transactions keep it up to date
// (bt_idx_mark / bt_idx_dirty / bt_idx_wstamp).  One wave per block of 64 slots.
__global__ void __launch_bounds__(256) k_build_blkidx(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, const unsigned *__restrict__ pv,
                                                      const unsigned *__restrict__ bif0, const unsigned *__restrict__ bif1, unsigned norig, unsigned nblk,
                                                      unsigned long long *__restrict__ bidx)
{
	const unsigned blk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (blk >= nblk) return;
	const unsigned e = blk * 64u + lane;
	const bool in = e < norig;
	const uint8_t c = in ? ch[e] : (uint8_t)BT_DEAD_CHAR;
	const unsigned long long m0 = __ballot(in && bif0[e] != BT_NONE), m1 = __ballot(in && bif1[e] != BT_NONE), sp = __ballot(in && c == BT_SEP);
	const bool bad = in && (c == BT_DEAD_CHAR || (e + 1u < norig && nx[e] != e + 1u) || (e > 0u && pv[e] != e - 1u));
	const unsigned long long dirty = __ballot(bad) ? 1ull << 32 : 0ull;
	if (lane == 0) {
		unsigned long long *w = bidx + (size_t)blk * BT_IDX_WORDS;
		w[0] = m0; w[1] = m1; w[2] = sp; w[3] = dirty;
	}
}
// SBL_CHECK_INDEX=1 (tests): the maintained index against a rebuild -- marks and separators exactly, "not pristine" and the write
// stamps at least what the arrays show.  out[0] = blocks that differ, out[1] = first of them.
__global__ void __launch_bounds__(256) k_check_blkidx(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, const unsigned *__restrict__ pv,
                                                      const unsigned *__restrict__ bif0, const unsigned *__restrict__ bif1, const unsigned *__restrict__ wmax, unsigned norig, unsigned nblk,
                                                      const unsigned long long *__restrict__ bidx, unsigned *__restrict__ out)
{
	const unsigned blk = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
	if (blk >= nblk) return;
	const unsigned e = blk * 64u + lane;
	const bool in = e < norig;
	const uint8_t c = in ? ch[e] : (uint8_t)BT_DEAD_CHAR;
	const unsigned long long m0 = __ballot(in && bif0[e] != BT_NONE), m1 = __ballot(in && bif1[e] != BT_NONE), sp = __ballot(in && c == BT_SEP);
	const bool bad = in && (c == BT_DEAD_CHAR || (e + 1u < norig && nx[e] != e + 1u) || (e > 0u && pv[e] != e - 1u));
	const bool dirty = __ballot(bad) != 0ull;
	unsigned wm = in ? wmax[e] : 0u;
	for (int d = 32; d > 0; d >>= 1) { const unsigned v = __shfl_xor(wm, d); wm = v > wm ? v : wm; }
	if (lane == 0) {
		const unsigned long long *w = bidx + (size_t)blk * BT_IDX_WORDS;
		const bool ok = w[0] == m0 && w[1] == m1 && w[2] == sp && (!dirty || (w[3] >> 32)) && (unsigned)w[3] >= wm;
		if (!ok) { atomicAdd(&out[0], 1u); atomicMin(&out[1], blk); }
	}
}
// everything but the pool cursors (CTR_NE, CTR_NN) back to its start value (DeviceBackend::clear_counters)
__global__ void __launch_bounds__(256) k_clear_counters(unsigned *__restrict__ ctr)
{
	for (unsigned i = CTR_ERR + threadIdx.x; i < CTR_COUNT; i += 256) ctr[i] = i == CTR_VIOL ? BT_NONE : 0u;
}
// the write stamps are reset with rmax / wmax at the start of every iteration attempt (DeviceBackend::reset_round_state)
__global__ void __launch_bounds__(256) k_idx_clear_stamps(unsigned long long *__restrict__ bidx, unsigned nblk)
{
	const unsigned blk = blockIdx.x * blockDim.x + threadIdx.x;
	if (blk < nblk) reinterpret_cast<unsigned *>(bidx + (size_t)blk * BT_IDX_WORDS + 3)[0] = 0u;
}

struct SimplifyState {
	DevBuf ch, op, nx, pv, nodeof[2];
	DevBuf nslot, nnext, nidst, nclr, ndead, head[2], lsize[2];
	DevBuf ctr, need, big, touch, ck_touch, own, lock, rmax, wmax, win;
	DevBuf arena, snap_arena, big_arena, claims, live, robuf, bidx, instbuf, snap_list, snap_live;
	DevBuf ck_ch, ck_op, ck_nx, ck_pv, ck_bif[2], ck_nodeof[2], ck_nslot, ck_nnext, ck_ndead, ck_head[2], ck_lsize[2];
	DevBuf keys, skeys, selem, sorttmp, scantmp, perm, permin;
	DevBuf nmark, maux[2], iota, sel, tstamp;
	DevBuf lin, elin, lmpos[2], lmid[2], cnt1k, off1k;      // linearised marks of the later snapshots
	DevBuf flag, segidx, seg_head, seg_len, seg_succ_elem, succ[2], dist[2], newidx, ch_out, op_out;
	unsigned *h_ctr = nullptr;            // pinned, mapped: CTR_COUNT counters + the sequence number of the last post (k_select_write)
	unsigned *d_hctr = nullptr;           // its device address
	unsigned post_seq = 0;
	hipEvent_t ev[8] = {};                // created once per context (DeviceBackend borrows them): commit sampling pair [2, 3]
	std::vector<hipEvent_t> snap_ev;      // start / stop pairs around the snapshots of a stage, read at its end (no host synchronisation per snapshot)
	unsigned char *h_init = nullptr;      // pinned staging for the small host-to-device initialisations of a stage (no synchronisation needed before the host moves on)
};

struct RestartStage {};      // thrown out of an optimistic attempt that would need a roll-back (DeviceBackend::restore)

struct DeviceBackend {
	sbl_ctx *c;
	SimplifyState *st;
	GraphView g{};
	uint32_t cap_e = 0, cap_n = 0, nid_ = 0;
	size_t ne0_ = 0;                                                  // elements of the stage's input (padded): cap_e - ne0_ is the insertion slack
	uint32_t ck_ne = 0, ck_nn = 0;
	uint32_t window = 0, arena_bytes = 1u << 17, snap_arena_bytes = 1u << 17, snap_threads = 256 * 32;   // snap_threads = resident waves
	uint32_t big_arena_bytes = 1u << 28;
	size_t nres = 0;
	hipEvent_t *ev = nullptr;                                          // SimplifyState::ev
	unsigned snap_used = 0;                                           // snapshot event pairs recorded in this stage
	bool timed_commit = false;
	bool later_stream = getenv("SBL_NO_STREAM_SNAPSHOT") == nullptr && getenv("SBL_NO_LATER_STREAM") == nullptr;
	bool first_stream = getenv("SBL_NO_STREAM_SNAPSHOT") == nullptr;      // measurement switch: the generic window-walking snapshot for iteration 1 too
	int prof = 0;
	// Optimistic attempt: no iteration checkpoints (1.45 GB of device-to-device copies per iteration, 3.6 ms of a 105 ms stage, for a
	// roll-back the benchmark workloads never take).  An order violation or a pool overflow then abandons the attempt (RestartStage) and
	// the stage is run again from its input -- intact until the copy-back -- with checkpoints and iteration replays (sbl_simplify_run).
	bool optimistic = false;
	unsigned rsv_waves = 4;                                           // waves of a reservation workgroup (k_reserve)
	// block index of the original slots (GraphView::bidx): read by the probe and the reservation, maintained by the transactions
	bool use_index = getenv("SBL_NO_BLOCK_INDEX") == nullptr;        // measurement / test switch: every window is walked (round 4)
	uint32_t idx_nblk = 0;
	unsigned probe_lds_pad = getenv("SBL_PROBE_LDS_PAD") ? (unsigned)atoi(getenv("SBL_PROBE_LDS_PAD")) : 0u;      // measurement switch: occupancy sensitivity of k_probe_idx
	// k_probe_idx's LDS by the instances an id has (set with rsv_waves): a handful -- 256-slot verdict table, 64 instances, 64 walked marks = 3.1 KB;
	// dozens (many strains) -- 512 slots, 256 instances, 192 marks = 7.9 KB.  An entry that does not fit goes to the walking probe.
	unsigned pidx_vbits = 9, pidx_inst = 256, pidx_marks = 192;
	unsigned istride() const { return std::min(pidx_inst, 128u) + 1u; }      // words per window entry of the instance hand-over (k_probe_idx -> k_reserve)
	unsigned pidx_lds() const { return ((2u << pidx_vbits) + 2u * pidx_inst + 2u * pidx_marks) * 4u + pidx_inst + probe_lds_pad; }
	void index_build()
	{
		if (!use_index || !idx_nblk) return;
		k_build_blkidx<<<(idx_nblk + 3) / 4, 256, 0, c->stream>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), st->pv.as<unsigned>(), c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>(),
		                                                         g.norig, idx_nblk, st->bidx.as<unsigned long long>());
		HIP_TRY(hipGetLastError());
	}
	double snapshot_ms = 0, reserve_ms = 0, commit_ms = 0, probe_ms = 0;
	double commit_event_ms = 0; uint64_t commit_event_launches = 0;      // the event pairs around every 4th launch of the commit kernel
	unsigned ev_phase = 0;

	DeviceBackend() = default;
	DeviceBackend(const DeviceBackend &) = delete;
	uint32_t nid() { return nid_; }
	void bind()
	{
		g.ch = st->ch.as<uint8_t>(); g.op = st->op.as<uint32_t>(); g.nx = st->nx.as<uint32_t>(); g.pv = st->pv.as<uint32_t>();
		for (int s = 0; s < 2; s++) {
			g.bif[s] = c->d_bif[s].as<uint32_t>(); g.nodeof[s] = st->nodeof[s].as<uint32_t>();
			g.head[s] = st->head[s].as<uint32_t>(); g.lsize[s] = st->lsize[s].as<uint32_t>();
		}
		g.nslot = st->nslot.as<uint32_t>(); g.nnext = st->nnext.as<uint32_t>(); g.nidst = st->nidst.as<uint32_t>(); g.nclr = st->nclr.as<uint32_t>(); g.ndead = st->ndead.as<uint8_t>();
		g.ctr = st->ctr.as<uint32_t>(); g.need = st->need.as<uint8_t>(); g.big = st->big.as<uint8_t>(); g.touch = st->touch.as<uint8_t>();
		g.own = st->own.as<uint32_t>(); g.lock = nullptr; g.rmax = st->rmax.as<uint32_t>(); g.wmax = st->wmax.as<uint32_t>();
		g.cap_e = cap_e; g.cap_n = cap_n; g.nid = nid_;
		g.nblk = (cap_e >> BT_BLOCK_SHIFT) + 1;
		g.win = st->win.as<uint32_t>();
		nres = (size_t)g.nblk + nid_ + 1;
		// (lock[] belongs to the element-wise stamping of the host-side test driver; the kernels check exclusivity against own[])
		st->rmax.ensure(nres * 4); st->wmax.ensure(nres * 4);
		g.lock = nullptr; g.rmax = st->rmax.as<uint32_t>(); g.wmax = st->wmax.as<uint32_t>();
		g.bidx = use_index && idx_nblk ? st->bidx.as<unsigned long long>() : nullptr;
		g.idx_probe = g.bidx && getenv("SBL_NO_IDX_PROBE") == nullptr ? 1u : 0u;        // measurement switches: the walking probe / reservation for every entry
		g.idx_reserve = g.bidx && getenv("SBL_NO_IDX_RESERVE") == nullptr ? 1u : 0u;
	}
	bool posted = false;                                              // the selection in flight posts the counters itself (k_select_write)
	void read_ctr()
	{
		if (posted && sel_pending && !sel_ready) {
			// the counters of the round and the next window come with the selection's post: poll its sequence number
			volatile unsigned *seq = st->h_ctr + CTR_COUNT;
			for (unsigned long long spin = 0;; spin++) {
				if (*seq == st->post_seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); posted = false; return; }
				if ((spin & 0xFFFF) == 0xFFFF) {                        // every ~65 k polls: is the stream still running?
					hipError_t e = hipStreamQuery(c->stream);
					if (e == hipSuccess && *seq != st->post_seq) break;      // finished without posting: fall back to the copy
					if (e != hipSuccess && e != hipErrorNotReady) HIP_TRY(e);
				}
			}
			posted = false;
		}
		HIP_TRY(hipMemcpyAsync(st->h_ctr, st->ctr.p, CTR_COUNT * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	void copy(DevBuf &dst, const DevBuf &src, size_t bytes)
	{
		dst.ensure(bytes);
		if (bytes) HIP_TRY(hipMemcpyAsync(dst.p, src.p, bytes, hipMemcpyDeviceToDevice, c->stream));
	}
	void checkpoint()
	{
		read_ctr();
		ck_ne = st->h_ctr[CTR_NE]; ck_nn = st->h_ctr[CTR_NN];
		if (optimistic) return;
		copy(st->ck_ch, st->ch, ck_ne); copy(st->ck_op, st->op, (size_t)ck_ne * 4); copy(st->ck_nx, st->nx, (size_t)ck_ne * 4); copy(st->ck_pv, st->pv, (size_t)ck_ne * 4);
		for (int s = 0; s < 2; s++) {
			copy(st->ck_bif[s], c->d_bif[s], (size_t)ck_ne * 4); copy(st->ck_nodeof[s], st->nodeof[s], (size_t)ck_ne * 4);
			copy(st->ck_head[s], st->head[s], ((size_t)nid_ + 1) * 4); copy(st->ck_lsize[s], st->lsize[s], ((size_t)nid_ + 1) * 4);
		}
		copy(st->ck_touch, st->touch, (size_t)nid_ + 1);
		copy(st->ck_nslot, st->nslot, (size_t)ck_nn * 4); copy(st->ck_nnext, st->nnext, (size_t)ck_nn * 4); copy(st->ck_ndead, st->ndead, ck_nn);
	}
	void restore()
	{
		if (optimistic) throw RestartStage{};
		auto back = [&](DevBuf &dst, const DevBuf &src, size_t bytes) { if (bytes) HIP_TRY(hipMemcpyAsync(dst.p, src.p, bytes, hipMemcpyDeviceToDevice, c->stream)); };
		back(st->ch, st->ck_ch, ck_ne); back(st->op, st->ck_op, (size_t)ck_ne * 4); back(st->nx, st->ck_nx, (size_t)ck_ne * 4); back(st->pv, st->ck_pv, (size_t)ck_ne * 4);
		for (int s = 0; s < 2; s++) {
			back(c->d_bif[s], st->ck_bif[s], (size_t)ck_ne * 4); back(st->nodeof[s], st->ck_nodeof[s], (size_t)ck_ne * 4);
			back(st->head[s], st->ck_head[s], ((size_t)nid_ + 1) * 4); back(st->lsize[s], st->ck_lsize[s], ((size_t)nid_ + 1) * 4);
		}
		back(st->touch, st->ck_touch, (size_t)nid_ + 1);
		back(st->nslot, st->ck_nslot, (size_t)ck_nn * 4); back(st->nnext, st->ck_nnext, (size_t)ck_nn * 4); back(st->ndead, st->ck_ndead, ck_nn);
		unsigned v[2] = { ck_ne, ck_nn };
		HIP_TRY(hipMemcpyAsync(st->ctr.as<unsigned>() + CTR_NE, &v[0], 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(st->ctr.as<unsigned>() + CTR_NN, &v[1], 4, hipMemcpyHostToDevice, c->stream));
		index_build();                                              // (the block index is derived state: rebuilt from the restored arrays)
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	// segment ranking of the current list (shared with the copy-back): returns the list length, leaves flag / segidx / seg_head / dist[cur] filled
	unsigned long long rank_segments(unsigned ne, int *cur_out)
	{
		hipStream_t s = c->stream;
		st->flag.ensure((size_t)ne * 4 + 16); st->segidx.ensure((size_t)ne * 4 + 16);
		k_seg_flags<<<(ne + 255) / 256, 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>());
		{
			size_t tmp = 0;
			HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), 0u, (size_t)ne + 1, rocprim::plus<unsigned>(), s));
			st->scantmp.ensure(tmp);
			HIP_TRY(rocprim::exclusive_scan(st->scantmp.p, tmp, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), 0u, (size_t)ne + 1, rocprim::plus<unsigned>(), s));
		}
		unsigned nseg = 0;
		HIP_TRY(hipMemcpyAsync(&nseg, st->segidx.as<unsigned>() + ne, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		SBL_CHECK(nseg >= 1, SBL_ERR_INTERNAL, "segment ranking: empty list");
		st->seg_head.ensure((size_t)nseg * 4); st->seg_len.ensure((size_t)nseg * 4); st->seg_succ_elem.ensure((size_t)nseg * 4);
		for (int t = 0; t < 2; t++) { st->succ[t].ensure((size_t)nseg * 4); st->dist[t].ensure((size_t)nseg * 8); }
		k_seg_tails<<<(ne + 255) / 256, 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(),
		                                            st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>());
		k_seg_finish<<<(nseg + 255) / 256, 256, 0, s>>>(nseg, st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>(),
		                                               st->flag.as<unsigned>(), st->segidx.as<unsigned>(), st->succ[0].as<unsigned>(), st->dist[0].as<unsigned long long>());
		int cur = 0;
		for (unsigned span = 1; span < nseg; span <<= 1, cur ^= 1)
			k_seg_jump<<<(nseg + 255) / 256, 256, 0, s>>>(nseg, st->succ[cur].as<unsigned>(), st->dist[cur].as<unsigned long long>(),
			                                             st->succ[cur ^ 1].as<unsigned>(), st->dist[cur ^ 1].as<unsigned long long>());
		unsigned long long total = 0;                               // segment 0 starts with element 0, the head of the whole list
		HIP_TRY(hipMemcpyAsync(&total, st->dist[cur].p, 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		*cur_out = cur;
		return total;
	}
	// marks of the current graph in list order (see k_lin_positions): fills ms
	void linearise_marks(MarkStream &ms)
	{
		hipStream_t s = c->stream;
		const unsigned ne = ck_ne;                                  // (checkpoint() has just read the counters)
		int cur = 0;
		const unsigned long long total = rank_segments(ne, &cur);
		st->lin.ensure((size_t)ne * 4 + 16); st->elin.ensure((size_t)total * 4 + 16);
		k_lin_positions<<<(ne + 255) / 256, 256, 0, s>>>(st->ch.as<uint8_t>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), st->seg_head.as<unsigned>(),
		                                                st->dist[cur].as<unsigned long long>(), total, st->lin.as<unsigned>(), st->elin.as<unsigned>());
		const unsigned nchunks = (unsigned)((total + 1023) / 1024);
		st->cnt1k.ensure((size_t)(nchunks + 1) * 4); st->off1k.ensure((size_t)(nchunks + 1) * 4);
		for (int t = 0; t < 2; t++) {
			HIP_TRY(hipMemsetAsync(st->cnt1k.p, 0, (size_t)(nchunks + 1) * 4, s));
			k_count_marks_lin<<<nchunks, 256, 0, s>>>(c->d_bif[t].as<unsigned>(), st->elin.as<unsigned>(), (size_t)total, st->cnt1k.as<unsigned>());
			size_t tmp = 0;
			HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, st->cnt1k.as<unsigned>(), st->off1k.as<unsigned>(), 0u, (size_t)nchunks + 1, rocprim::plus<unsigned>(), s));
			st->scantmp.ensure(tmp);
			HIP_TRY(rocprim::exclusive_scan(st->scantmp.p, tmp, st->cnt1k.as<unsigned>(), st->off1k.as<unsigned>(), 0u, (size_t)nchunks + 1, rocprim::plus<unsigned>(), s));
			unsigned nm = 0;
			HIP_TRY(hipMemcpyAsync(&nm, st->off1k.as<unsigned>() + nchunks, 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			st->lmpos[t].ensure((size_t)nm * 4 + 16); st->lmid[t].ensure((size_t)nm * 4 + 16); st->maux[t].ensure((size_t)nm * 4 + 16);
			if (nm) {
				k_write_marks_lin<<<nchunks, 256, 0, s>>>(c->d_bif[t].as<unsigned>(), st->elin.as<unsigned>(), st->nodeof[t].as<unsigned>(), (size_t)total, st->off1k.as<unsigned>(),
				                                         st->lmpos[t].as<unsigned>(), st->lmid[t].as<unsigned>(), st->nmark.as<unsigned>());
				k_mark_aux_lin<<<(nm + 255) / 256, 256, 0, s>>>(st->lmpos[t].as<unsigned>(), nm, (unsigned)t, c->d_sepidx.as<unsigned>(), c->nchr, st->lin.as<unsigned>(),
				                                               st->elin.as<unsigned>(), st->ch.as<uint8_t>(), g.k, st->maux[t].as<unsigned>());
			}
			ms.elem[t] = st->lmpos[t].as<unsigned>(); ms.id[t] = st->lmid[t].as<unsigned>(); ms.aux[t] = st->maux[t].as<unsigned>(); ms.n[t] = nm;
		}
		HIP_TRY(hipGetLastError());
	}
	// Linearising the marks for the stream costs ~1.5 ms whatever the number of ids to look at; when only a few per cent of the ids were
	// touched (iterations 3, 4: what the few collapses of the previous iteration can see) the window-walking snapshot of just those is cheaper.
	bool few_touched()
	{
		unsigned *cnt = st->ctr.as<unsigned>() + CTR_DETAIL + 8, n = 0;
		HIP_TRY(hipMemsetAsync(cnt, 0, 4, c->stream));
		k_count_touched<<<256, 256, 0, c->stream>>>(st->touch.as<uint8_t>(), nid_, cnt);
		HIP_TRY(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		return (unsigned long long)n * 16 < nid_;
	}
	// ---- read-only phases split over the attached GPUs (SURVEY.md 8e, row "Simplification"): the commits are replicated, so the graph is
	// identical on every GPU before a snapshot and before a probe; each GPU takes the verdicts of ITS share (a slice of the positional
	// order of the ids / of the window) and the verdict bytes are all-gathered: 1 B per id per snapshot, 1 B per window entry per round.
	// SBL_REPLICATED_PHASES=1: measurement / test switch, every GPU computes everything (the round-3 behaviour).
	bool split_ro() const { return c->comm && c->comm->n > 1 && getenv("SBL_REPLICATED_PHASES") == nullptr; }
	double ro_ms = 0;                                                 // host time inside the verdict collectives
	void share(uint32_t n, uint32_t *lo, uint32_t *hi) const
	{
		const uint32_t R = c->comm->n, r = c->comm->rank;
		*lo = (uint32_t)((uint64_t)n * r / R); *hi = (uint32_t)((uint64_t)n * (r + 1) / R);
	}
	// buf[lo_r, hi_r) of every rank r (element size `es` bytes, n elements in all) to everybody, in place
	void allgather_shares(char *buf, uint32_t n, size_t es)
	{
		SblComm *cm = c->comm;
		const uint32_t R = cm->n, r = cm->rank;
		std::vector<size_t> sb(R), so(R), rb(R), ro(R);
		for (uint32_t p = 0; p < R; p++) {
			const size_t plo = (size_t)((uint64_t)n * p / R) * es, phi = (size_t)((uint64_t)n * (p + 1) / R) * es;
			const size_t mlo = (size_t)((uint64_t)n * r / R) * es, mhi = (size_t)((uint64_t)n * (r + 1) / R) * es;
			sb[p] = p == r ? 0 : mhi - mlo; so[p] = mlo;
			rb[p] = p == r ? 0 : phi - plo; ro[p] = plo;
			if (p != r) c->stats.verdict_bytes += mhi - mlo;
		}
		const auto t0 = std::chrono::steady_clock::now();
		try { cm->alltoallv(c, buf, sb.data(), so.data(), buf, rb.data(), ro.data()); }
		catch (...) { cm->abort_peers(); throw; }
		ro_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	}
	uint32_t snap_slice = 0;                                          // entries per launch of the walking probe's arena (= the round buffers' window_max)
	void snapshot_idx()
	{
		unsigned *cnt = st->ctr.as<unsigned>() + CTR_DETAIL + 8, n = 0;
		st->snap_list.ensure((size_t)nid_ * 4 + 64); st->snap_live.ensure((size_t)nid_ + 64);
		HIP_TRY(hipMemsetAsync(cnt, 0, 4, c->stream));
		k_touched_list<<<(nid_ + 255) / 256, 256, 0, c->stream>>>(st->touch.as<uint8_t>(), nid_, st->snap_list.as<unsigned>(), cnt);
		HIP_TRY(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemsetAsync(st->need.p, 0, (size_t)nid_ + 1, c->stream));      // untouched ids: still clean
		HIP_TRY(hipMemsetAsync(st->touch.p, 0, (size_t)nid_ + 1, c->stream));     // every verdict is taken now
		HIP_TRY(hipStreamSynchronize(c->stream));
		const uint32_t slice = std::max<uint32_t>(1, snap_slice);
		for (uint32_t off = 0; off < n; off += slice) {
			const uint32_t m = std::min<uint32_t>(slice, n - off);
			GraphView gs = g;
			gs.win = st->snap_list.as<unsigned>() + off;
			gs.tstamp = nullptr;
			uint8_t *lv = st->snap_live.as<uint8_t>() + off;
			k_probe_idx<<<m, 64, pidx_lds(), c->stream>>>(gs, m, lv, 0u, pidx_vbits, pidx_inst, pidx_marks, nullptr, 0u, 1);
			k_probe<<<m, 64 * PROBE_WAVES, 0, c->stream>>>(gs, m, st->arena.as<uint8_t>(), arena_bytes, lv, 0u, 1);
		}
		HIP_TRY(hipGetLastError());
	}
	void snapshot_all(bool incremental)
	{
		st->snap_arena.ensure((size_t)snap_threads * snap_arena_bytes);
		while (st->snap_ev.size() < 2 * (size_t)(snap_used + 1)) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); st->snap_ev.push_back(e); }
		HIP_TRY(hipEventRecord(st->snap_ev[2 * snap_used], c->stream));
		uint32_t plo = 0, phi = nid_;
		const bool split = split_ro();
		if (split) share(nid_, &plo, &phi);
		if (!incremental && first_stream) {
			// iteration 1 (also after a replay: the checkpoint restored is the pristine graph): stream over the position-ordered marks
			MarkStream ms;
			for (int t = 0; t < 2; t++) { ms.elem[t] = c->d_melem[t].as<unsigned>(); ms.id[t] = c->d_mid[t].as<unsigned>(); ms.aux[t] = st->maux[t].as<unsigned>(); ms.n[t] = c->nmarks[t]; }
			HIP_TRY(hipMemsetAsync(st->touch.p, 0, (size_t)nid_ + 1, c->stream));
			k_snapshot_first<<<256 * 32, 64, 0, c->stream>>>(g, ms, st->nmark.as<unsigned>(), st->perm.as<unsigned>(), plo, phi);
		} else if (incremental && g.idx_probe && !split && getenv("SBL_NO_IDX_SNAPSHOT") == nullptr) {
			// iterations 2 ..: the touched ids through the probe of the rounds (k_probe_idx over the block index, the walking probe for what it
			// cannot serve) -- the same verdict the stream takes, without linearising the marks first (round 5)
			snapshot_idx();
		} else if (incremental && later_stream && !few_touched()) {
			// iterations 2 ..: the same stream over the marks of the current graph in list order
			st->nmark.ensure((size_t)cap_n * 4);
			MarkStream ms;
			linearise_marks(ms);
			k_snapshot_stream<<<256 * 32, 64, 0, c->stream>>>(g, ms, st->nmark.as<unsigned>(), st->perm.as<unsigned>(), 1, plo, phi);
		} else
		k_snapshot<<<snap_threads, 64, 0, c->stream>>>(g, st->snap_arena.as<uint8_t>(), snap_arena_bytes, incremental ? 1 : 0, st->perm.as<unsigned>(), plo, phi);
		HIP_TRY(hipGetLastError());
		if (split) {
			// the verdict bytes of my share of the positional order to everybody; after a snapshot no id is "touched" any more, anywhere
			st->robuf.ensure((size_t)nid_ + 64);
			if (phi > plo) k_pack_need<<<(phi - plo + 255) / 256, 256, 0, c->stream>>>(st->perm.as<unsigned>(), st->need.as<uint8_t>(), plo, phi, st->robuf.as<uint8_t>());
			allgather_shares(st->robuf.as<char>(), nid_, 1);
			if (nid_) k_unpack_need<<<(nid_ + 255) / 256, 256, 0, c->stream>>>(st->perm.as<unsigned>(), st->robuf.as<uint8_t>(), nid_, plo, phi, st->need.as<uint8_t>());
			HIP_TRY(hipMemsetAsync(st->touch.p, 0, (size_t)nid_ + 1, c->stream));
			HIP_TRY(hipGetLastError());
		}
		HIP_TRY(hipEventRecord(st->snap_ev[2 * snap_used + 1], c->stream));
		snap_used++;                                                // (read by snapshots_collect at the end of the stage: the host does not wait here)
	}
	void snapshots_collect()
	{
		for (unsigned i = 0; i < snap_used; i++) {
			float ms = 0;
			HIP_TRY(hipEventSynchronize(st->snap_ev[2 * i + 1]));
			HIP_TRY(hipEventElapsedTime(&ms, st->snap_ev[2 * i], st->snap_ev[2 * i + 1]));
			snapshot_ms += ms;
		}
		snap_used = 0;
	}
	void reset_round_state(bool stamps_too)
	{
		HIP_TRY(hipMemsetAsync(st->own.p, 0xFF, ((size_t)nid_ + 1) * 4, c->stream));
		if (stamps_too) {
			HIP_TRY(hipMemsetAsync(st->rmax.p, 0, nres * 4, c->stream));
			HIP_TRY(hipMemsetAsync(st->wmax.p, 0, nres * 4, c->stream));
			if (g.bidx) k_idx_clear_stamps<<<(idx_nblk + 255) / 256, 256, 0, c->stream>>>(st->bidx.as<unsigned long long>(), idx_nblk);
		}
	}
	void clear_counters()
	{
		// everything but the pool cursors (CTR_NE, CTR_NN); one small kernel, no host synchronisation
		k_clear_counters<<<1, 256, 0, c->stream>>>(st->ctr.as<unsigned>());
		HIP_TRY(hipGetLastError());
	}
	// The selection is launched behind a round's last kernel and read with the round's counters: one host round trip per round.
	bool sel_pending = false, sel_ready = false;
	uint32_t probed_nwin = 0;                                         // entries of the last probe whose retirements have not been counted yet
	void select_launch(uint32_t lo, uint32_t limit, uint32_t W)
	{
		// chunks of >= 8192 ids, at most ~1024 of them
		unsigned chunk = 8192;
		while ((unsigned long long)chunk * 1024 < (unsigned long long)nid_ + 1) chunk <<= 1;
		const unsigned chunk0 = lo / chunk, nchunks = limit / chunk - chunk0 + 1;
		// (one launch with a look-back over per-chunk slots was tried: 37.7 us against 10.7 + 10.6 us for the two -- the agent-scope
		// release / acquire of the slots writes back and invalidates the L2 of the XCD, which a kernel boundary does once)
		GraphView gs = g;
		if (sel_stamped) gs.tslot = TS_CAP * 4;                     // only the selection right behind a round marks that round's end
		sel_stamped = true;
		k_select_count<<<nchunks, SEL_THREADS, 0, c->stream>>>(gs, st->sel.as<unsigned>(), lo, limit, chunk0, chunk, st->live.as<uint8_t>(), probed_nwin);
		probed_nwin = 0;
		posted = st->d_hctr != nullptr;
		if (posted) st->post_seq++;
		k_select_write<<<nchunks, SEL_THREADS, 0, c->stream>>>(g, st->sel.as<unsigned>(), st->win.as<unsigned>(), lo, limit, W, chunk0, chunk, nchunks, st->d_hctr, st->post_seq);
		HIP_TRY(hipGetLastError());
		sel_pending = true; sel_ready = false;
	}
	void select_read(uint32_t *nwin, uint32_t *newlo, uint32_t *solo)
	{
		if (!sel_ready) read_ctr();                                 // (the first selection of an iteration, or one re-issued behind a fence)
		sel_pending = sel_ready = false;
		*nwin = st->h_ctr[CTR_NWIN]; *newlo = st->h_ctr[CTR_LO]; *solo = st->h_ctr[CTR_PUSHED];
	}
	// Per-kernel times of the rounds: start stamps written by the kernels themselves (round_stamp) for probe and reservation, and a
	// HIP event pair around the dominant kernel, k_commit (what bench.py's roofline is computed from).
	bool phase_events = getenv("SBL_NO_PHASE_EVENTS") == nullptr;    // measurement switch: what the per-round events themselves cost
	enum { TS_CAP = 16384 };                                         // rounds with stamps per stage (later ones go untimed)
	uint32_t ts_round = 0;
	bool sel_stamped = true;
	std::vector<uint8_t> ts_kind;                                    // per round: bit 0 probe, bit 1 reservation, bit 2 commit / chain launched
	void stamps_init()
	{
		st->tstamp.ensure((size_t)(TS_CAP + 1) * 4 * 8);
		HIP_TRY(hipMemsetAsync(st->tstamp.p, 0, (size_t)(TS_CAP + 1) * 4 * 8, c->stream));
		ts_round = 0; ts_kind.clear(); sel_stamped = true;
		g.tstamp = st->tstamp.as<unsigned long long>(); g.tslot = TS_CAP * 4;      // (the slot behind the last round: writes nobody reads)
	}
	void begin_round()
	{
		if (ts_round < TS_CAP) { g.tslot = 4 * ts_round++; ts_kind.push_back(0); sel_stamped = false; }
		else g.tslot = TS_CAP * 4;
	}
	// probe_ms / reserve_ms of the stage from the stamps (the event pair gives commit_ms)
	void stamps_collect()
	{
		if (!ts_round) return;
		std::vector<unsigned long long> h((size_t)ts_round * 4);
		HIP_TRY(hipMemcpyAsync(h.data(), st->tstamp.p, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		int khz = 0;
		if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
		const double ms_per_tick = 1.0 / (double)khz;
		for (uint32_t r = 0; r < ts_round; r++) {
			const unsigned long long tp = h[4 * r], tr = h[4 * r + 1], tc = h[4 * r + 2];
			const bool probed = ts_kind[r] & 1, reserved = ts_kind[r] & 2;
			if (probed) { const unsigned long long next = reserved ? tr : tc; if (next > tp && tp) probe_ms += (double)(next - tp) * ms_per_tick; }
			if (reserved && tc > tr && tr) reserve_ms += (double)(tc - tr) * ms_per_tick;
			const unsigned long long te = h[4 * r + 3];                 // start of the selection behind the round
			if ((ts_kind[r] & 4) && te > tc && tc) commit_ms += (double)(te - tc) * ms_per_tick;
		}
	}
	void probe(uint32_t nwin, uint32_t round)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		begin_round();
		solo_round = false;
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 1;
		if (split_ro()) {
			// my share of the window; live[] of the other shares and the lowest order violation anybody saw come back in two small all-gathers
			uint32_t w0 = 0, w1 = nwin;
			share(nwin, &w0, &w1);
			const uint32_t R = c->comm->n;
			st->robuf.ensure((size_t)R * 4 + 64);
			if (w1 > w0 && g.idx_probe) k_probe_idx<<<w1 - w0, 64, pidx_lds(), c->stream>>>(g, nwin, st->live.as<uint8_t>(), w0, pidx_vbits, pidx_inst, pidx_marks, nullptr, 0u, 0);      // (shares of a split probe: the other ranks' lists would have to travel too)
			if (w1 > w0) k_probe<<<w1 - w0, 64 * PROBE_WAVES, 0, c->stream>>>(g, nwin, st->arena.as<uint8_t>(), arena_bytes, st->live.as<uint8_t>(), w0);
			k_probe_trail<<<1, 1, 0, c->stream>>>(st->ctr.as<unsigned>(), st->robuf.as<unsigned>(), c->comm->rank);
			HIP_TRY(hipGetLastError());
			allgather_shares(st->live.as<char>(), nwin, 1);
			allgather_shares(st->robuf.as<char>(), R, 4);
			k_apply_probe<<<(nwin + 255) / 256, 256, 0, c->stream>>>(g, nwin, st->live.as<uint8_t>(), w0, w1, st->robuf.as<unsigned>(), R);
		} else {
			if (g.idx_probe) k_probe_idx<<<nwin, 64, pidx_lds(), c->stream>>>(g, nwin, st->live.as<uint8_t>(), 0u, pidx_vbits, pidx_inst, pidx_marks, st->instbuf.as<unsigned>(), istride(), 0);      // the block index first; k_probe walks what it could not serve
			k_probe<<<nwin, 64 * PROBE_WAVES, 0, c->stream>>>(g, nwin, st->arena.as<uint8_t>(), arena_bytes, st->live.as<uint8_t>(), 0u);
		}
		probed_nwin = nwin;                                          // (the next selection counts what this probe retired)
		HIP_TRY(hipGetLastError());
	}
	bool solo_round = false;                                          // the current round skipped the probe (mark_live): no instance hand-over
	void mark_live(uint32_t nwin) { begin_round(); solo_round = true; HIP_TRY(hipMemsetAsync(st->live.p, 1, nwin, c->stream)); }
	void reserve(uint32_t nwin, uint32_t round)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 2;
		// dynamic LDS: the seen-set + one compaction list per wave, sized by the instances an id has (a handful: 1024 + 2 x 256 words = 6 KB;
		// dozens: 2048 + 4 x 1024 words) -- what a reservation workgroup holds in LDS decides how many are resident
		const unsigned seen_bits = rsv_waves <= 2 ? 10u : 11u, list_cap = rsv_waves <= 2 ? 256u : 1024u;
		const bool handed = g.idx_probe && !split_ro() && !solo_round;
		k_reserve<<<nwin, 64 * rsv_waves, ((1u << seen_bits) + rsv_waves * list_cap) * 4, c->stream>>>(g, nwin, st->claims.as<unsigned>(), st->live.as<uint8_t>(), seen_bits, list_cap,
		                                                                                               handed ? st->instbuf.as<unsigned>() : nullptr, istride());
		HIP_TRY(hipGetLastError());
	}
	void commit(uint32_t nwin, uint32_t round, bool solo)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		// an event pair around every 4th launch (which ones rotates from stage to stage); the start stamps time all of them
		const bool sampled = phase_events && ((round + ev_phase) & 3u) == 0;
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 4;
		if (sampled) HIP_TRY(hipEventRecord(ev[2], c->stream));
		if (solo) {
			st->big_arena.ensure(big_arena_bytes);
			k_commit<<<1, 64, 0, c->stream>>>(g, 1, st->big_arena.as<uint8_t>(), big_arena_bytes, 1, nullptr, nullptr, prof);
		} else
			k_commit<<<nwin, 64, 0, c->stream>>>(g, nwin, st->arena.as<uint8_t>(), arena_bytes, 0, st->claims.as<unsigned>(), st->live.as<uint8_t>(), prof);
		if (sampled) HIP_TRY(hipEventRecord(ev[3], c->stream));
		timed_commit = sampled;
		HIP_TRY(hipGetLastError());
	}
	// serial chain over what is pending in the id range of the window (k_chain); timed with the commit phase
	bool chain(uint32_t nwin, uint32_t round)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		st->big_arena.ensure(big_arena_bytes);
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 4;
		k_chain<<<1, 64, 0, c->stream>>>(g, st->big_arena.as<uint8_t>(), big_arena_bytes, nwin, prof);
		timed_commit = false;
		HIP_TRY(hipGetLastError());
		return true;
	}
	SimplifyCounters counters()
	{
		read_ctr();
		if (sel_pending) sel_ready = true;                          // the snapshot holds the selection launched before it as well
		float ms = 0;
		if (timed_commit) {
			hipError_t e = hipEventElapsedTime(&ms, ev[2], ev[3]);      // (the post of the selection behind them has arrived: normally complete)
			if (e == hipErrorNotReady) { HIP_TRY(hipEventSynchronize(ev[3])); e = hipEventElapsedTime(&ms, ev[2], ev[3]); }
			HIP_TRY(e);
			commit_event_ms += ms; commit_event_launches++; timed_commit = false;
		}
		SimplifyCounters r;
		memcpy(r.v, st->h_ctr, sizeof r.v);
		return r;
	}
	bool grow(uint32_t err)
	{
		if (err & ~(uint32_t)(BT_ERR_ELEM_CAP | BT_ERR_NODE_CAP)) return false;
		if (optimistic) {
			// no checkpoint to replay from: the attempt is abandoned BEFORE any buffer is enlarged, and the capacities the rerun (and the
			// later stages of this context) should start with are left on the context -- the rerun used to start with the original
			// capacities again, overflow again at the same place, and only then grow + replay
			if (err & BT_ERR_ELEM_CAP) c->hint_elem_slack = std::max<size_t>(c->hint_elem_slack, 2 * ((size_t)cap_e - ne0_));
			if (err & BT_ERR_NODE_CAP) c->hint_cap_n = std::max<size_t>(c->hint_cap_n, 2 * (size_t)cap_n);
			throw RestartStage{};
		}
		hipStream_t s = c->stream;
		if (err & BT_ERR_ELEM_CAP) {
			size_t n = (size_t)cap_e * 2;
			SBL_CHECK(n < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "element capacity overflow");
			st->ch.grow_keep(n, cap_e, s); st->op.grow_keep(n * 4, (size_t)cap_e * 4, s); st->nx.grow_keep(n * 4, (size_t)cap_e * 4, s); st->pv.grow_keep(n * 4, (size_t)cap_e * 4, s);
			for (int k = 0; k < 2; k++) { c->d_bif[k].grow_keep(n * 4, (size_t)cap_e * 4, s); st->nodeof[k].grow_keep(n * 4, (size_t)cap_e * 4, s); }
			HIP_TRY(hipMemsetAsync(st->ch.as<uint8_t>() + cap_e, 0, n - cap_e, s));
			HIP_TRY(hipMemsetAsync(st->nx.as<unsigned>() + cap_e, 0xFF, (n - cap_e) * 4, s));
			HIP_TRY(hipMemsetAsync(st->pv.as<unsigned>() + cap_e, 0xFF, (n - cap_e) * 4, s));
			for (int k = 0; k < 2; k++) HIP_TRY(hipMemsetAsync(c->d_bif[k].as<unsigned>() + cap_e, 0xFF, (n - cap_e) * 4, s));
			cap_e = (uint32_t)n;
		}
		if (err & BT_ERR_NODE_CAP) {
			size_t n = (size_t)cap_n * 2;
			SBL_CHECK(n < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "node capacity overflow");
			st->nslot.grow_keep(n * 4, (size_t)cap_n * 4, s); st->nnext.grow_keep(n * 4, (size_t)cap_n * 4, s); st->nclr.grow_keep(n * 4, (size_t)cap_n * 4, s); st->nidst.grow_keep(n * 4, (size_t)cap_n * 4, s);
			st->ndead.grow_keep(n, cap_n, s);
			cap_n = (uint32_t)n;
		}
		bind();
		return true;
	}
};

void sbl_simplify_free(sbl_ctx *c)
{
	SimplifyState *st = c->simp;
	if (!st) return;
	DevBuf *bufs[] = { &st->ch, &st->op, &st->nx, &st->pv, &st->nodeof[0], &st->nodeof[1], &st->nslot, &st->nnext, &st->nidst, &st->nclr, &st->ndead,
	                   &st->head[0], &st->head[1], &st->lsize[0], &st->lsize[1], &st->ctr, &st->need, &st->big, &st->touch, &st->ck_touch, &st->own, &st->lock, &st->rmax, &st->wmax, &st->win,
	                   &st->arena, &st->snap_arena, &st->big_arena, &st->claims, &st->live, &st->robuf, &st->bidx, &st->instbuf, &st->snap_list, &st->snap_live, &st->ck_ch, &st->ck_op, &st->ck_nx, &st->ck_pv, &st->ck_bif[0], &st->ck_bif[1],
	                   &st->ck_nodeof[0], &st->ck_nodeof[1], &st->ck_nslot, &st->ck_nnext, &st->ck_ndead, &st->ck_head[0], &st->ck_head[1], &st->ck_lsize[0], &st->ck_lsize[1],
	                   &st->lin, &st->elin, &st->lmpos[0], &st->lmpos[1], &st->lmid[0], &st->lmid[1], &st->cnt1k, &st->off1k, &st->sel, &st->tstamp, &st->nmark, &st->maux[0], &st->maux[1], &st->iota, &st->keys, &st->skeys, &st->selem, &st->sorttmp, &st->scantmp, &st->perm, &st->permin, &st->flag, &st->segidx, &st->seg_head, &st->seg_len, &st->seg_succ_elem,
	                   &st->succ[0], &st->succ[1], &st->dist[0], &st->dist[1], &st->newidx, &st->ch_out, &st->op_out };
	for (DevBuf *b : bufs) b->release();
	if (st->h_ctr) (void)hipHostFree(st->h_ctr);
	if (st->h_init) (void)hipHostFree(st->h_init);
	for (auto &e : st->ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : st->snap_ev) (void)hipEventDestroy(e);
	delete st;
	c->simp = nullptr;
}

static void sort_pairs64(sbl_ctx *c, SimplifyState *st, unsigned long long *kin, unsigned long long *kout, unsigned *vin, unsigned *vout, size_t n, unsigned bits = 64)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
	st->sorttmp.ensure(tmp);
	HIP_TRY(rocprim::radix_sort_pairs(st->sorttmp.p, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
}
static unsigned bits_of(unsigned long long v) { unsigned b = 1; while (b < 64 && (v >> b)) b++; return b; }
static void scan_u32(sbl_ctx *c, SimplifyState *st, unsigned *in, unsigned *out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
	st->scantmp.ensure(tmp);
	HIP_TRY(rocprim::exclusive_scan(st->scantmp.p, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
}

// Inputs up to this many elements take the one-launch path (k_dense_stage) first; SBL_NO_DENSE_PATH=1 / SBL_DENSE_MAX_ELEMS=n: test switches
#define DENSE_MAX_ELEMS (1u << 16)
// The reference's callback sequence (blockfinder.cpp:23-48) is a function of the call index alone -- start, run(min(i, 50)) for i = 1, 2, ...,
// end -- so an attempt that is abandoned and run again delivers only the calls the caller has not seen yet.
struct ProgressFilter {
	sbl_progress_fn fn; void *user;
	bool started = false, ended = false;
	uint64_t delivered = 0, seen = 0;          // run calls forwarded so far / made by the current attempt
	static void relay(size_t p, int state, void *self_)
	{
		ProgressFilter *f = (ProgressFilter *)self_;
		if (state == SBL_PROGRESS_START) { f->seen = 0; if (!f->started) { f->started = true; f->fn(p, state, f->user); } }
		else if (state == SBL_PROGRESS_RUN) { if (++f->seen > f->delivered) { f->delivered = f->seen; f->fn(p, state, f->user); } }
		else if (!f->ended) { f->ended = true; f->fn(p, state, f->user); }
	}
};
enum { RUN_DONE = 0, RUN_DENSE_FAILED = 1, RUN_RESTART = 2 };
static int simplify_run_impl(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges, bool allow_dense, bool optimistic);
static void simplify_run_guarded(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges);
void sbl_simplify_run(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges)
{
	// With several GPUs on one job every snapshot and every probe of the stage is a collective (allgather_shares): a rank that leaves
	// the stage with an error anywhere -- an allocation that fails, a HIP error, a check that throws -- must release the peers that
	// would wait for it there (the deterministic RestartStage is thrown on all ranks alike and never gets here).
	if (!c->comm) { simplify_run_guarded(c, k, D, max_iter, progress, user, bulges); return; }
	try { simplify_run_guarded(c, k, D, max_iter, progress, user, bulges); }
	catch (...) { c->comm->abort_peers(); throw; }
}
static void simplify_run_guarded(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges)
{
	// the stage's input (d_ch / d_op) is only replaced by the copy-back at the very end, so an attempt that cannot finish -- a one-launch
	// run out of pool or arena space, an optimistic run that would need a roll-back -- is simply followed by the next one from the same input
	ProgressFilter pf{progress, user};
	sbl_progress_fn pfn = progress ? &ProgressFilter::relay : nullptr;
	// (SBL_CHECKPOINTS: measurement / test switch, checkpoints from the first attempt on; hint_checkpoints: the previous stage of this
	// context had to be abandoned for an order violation -- inputs that do that once tend to do it again, and an abandoned attempt
	// costs a whole stage, a checkpoint 2 - 4 %)
	const bool optimistic = getenv("SBL_CHECKPOINTS") == nullptr && !c->hint_checkpoints;
	int r = simplify_run_impl(c, k, D, max_iter, pfn, &pf, bulges, true, optimistic);
	if (r == RUN_DENSE_FAILED) r = simplify_run_impl(c, k, D, max_iter, pfn, &pf, bulges, false, optimistic);
	if (r == RUN_RESTART) {
		if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] optimistic attempt abandoned: the stage runs again with iteration checkpoints (element slack hint %zu, node capacity hint %zu)\n", c->hint_elem_slack, c->hint_cap_n);
		(void)simplify_run_impl(c, k, D, max_iter, pfn, &pf, bulges, false, false);
		c->stats.replays++;                                               // the abandoned attempt
		c->hint_checkpoints = c->stats.replays > c->stats.grow_replays + 1;      // order violations (not just a pool that was too small): the next stage starts with checkpoints
	} else if (!optimistic && r == RUN_DONE && c->stats.replays == c->stats.grow_replays) c->hint_checkpoints = false;      // a checkpointed stage that never rolled back
}
static int simplify_run_impl(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges, bool allow_dense, bool optimistic)
{
	hipStream_t s = c->stream;
	if (!c->simp) {
		c->simp = new SimplifyState();
		HIP_TRY(hipHostMalloc((void **)&c->simp->h_ctr, (CTR_COUNT + 16) * 4, hipHostMallocMapped | hipHostMallocCoherent));
		memset(c->simp->h_ctr, 0, (CTR_COUNT + 16) * 4);
		if (hipHostGetDevicePointer((void **)&c->simp->d_hctr, c->simp->h_ctr, 0) != hipSuccess) { (void)hipGetLastError(); c->simp->d_hctr = nullptr; }
		if (getenv("SBL_NO_POST")) c->simp->d_hctr = nullptr;             // measurement switch: copy + synchronise as before
	}
	SimplifyState *st = c->simp;
	DeviceBackend be;
	be.c = c; be.st = st;
	c->stats = sbl_stage_stats{};
	HIP_TRY(hipEventRecord(c->ev[2], s));

	// ---- E1: enumeration into mark arrays with room for inserted elements
	size_t E = c->nelem, ne0 = (E + 31) / 32 * 32;
	size_t cap_e = ne0 + E / 8 + (1u << 20);
	if (const char *e = getenv("SBL_TEST_ELEM_SLACK")) cap_e = ne0 + (size_t)atoll(e);      // test hook: provoke the grow / restart paths
	cap_e = std::max(cap_e, ne0 + std::min<size_t>(c->hint_elem_slack, 4 * E + (1u << 20)));      // what an abandoned attempt of this context asked for (DeviceBackend::grow), bounded by the current input
	be.ne0_ = ne0;
	SBL_CHECK(cap_e < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "element capacity overflow");
	sbl_run_enumeration(c, k, cap_e);
	if (c->comm) if (const char *e = getenv("SBL_TEST_FAIL_SIMPLIFY_RANK")) if ((uint32_t)atoi(e) == c->comm->rank)      // test hook: this rank leaves the stage between two collectives
		throw SblError{SBL_ERR_OOM, "out of memory (SBL_TEST_FAIL_SIMPLIFY_RANK: this rank leaves the simplification stage)"};
	be.nid_ = c->bif_count;
	be.cap_e = (uint32_t)cap_e;

	// ---- graph arrays
	st->ch.ensure(cap_e); st->op.ensure(cap_e * 4); st->nx.ensure(cap_e * 4); st->pv.ensure(cap_e * 4);
	st->nodeof[0].ensure(cap_e * 4); st->nodeof[1].ensure(cap_e * 4);
	HIP_TRY(hipMemcpyAsync(st->ch.p, c->d_ch.p, E, hipMemcpyDeviceToDevice, s));
	HIP_TRY(hipMemcpyAsync(st->op.p, c->d_op.p, E * 4, hipMemcpyDeviceToDevice, s));
	k_init_links<<<nblocks(cap_e, 256), 256, 0, s>>>(st->nx.as<unsigned>(), st->pv.as<unsigned>(), st->nodeof[0].as<unsigned>(), st->nodeof[1].as<unsigned>(),
	                                                st->ch.as<uint8_t>(), E, cap_e);

	// ---- E2: instance lists in the reference's initial order
	sbl_compact_marks(c, 0);
	sbl_compact_marks(c, 1);
	size_t n0 = c->nmarks[0], n1 = c->nmarks[1], ninst = n0 + n1;
	c->stats.instances = ninst;
	size_t dense_max = DENSE_MAX_ELEMS;
	if (const char *e = getenv("SBL_DENSE_MAX_ELEMS")) dense_max = (size_t)atoll(e);
	const bool dense = allow_dense && E <= dense_max && be.nid_ > 0 && getenv("SBL_NO_DENSE_PATH") == nullptr;
	// (the one-launch path cannot grow a pool and replay: low-complexity input makes hundreds of nodes per collapse, and 16 M nodes are 270 MB)
	size_t cap_n = dense ? std::max<size_t>(4 * ninst + (1u << 20), 16u << 20) : 4 * ninst + (1u << 20);
	if (dense) if (const char *e = getenv("SBL_TEST_DENSE_NODE_SLACK")) cap_n = ninst + (size_t)atoll(e);      // test hook: provoke the fall-back
	cap_n = std::max(cap_n, c->hint_cap_n);
	SBL_CHECK(cap_n < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "node capacity overflow");
	be.cap_n = (uint32_t)cap_n;
	st->nslot.ensure(cap_n * 4); st->nnext.ensure(cap_n * 4); st->nidst.ensure(cap_n * 4); st->nclr.ensure(cap_n * 4); st->ndead.ensure(cap_n);
	size_t nidp = (size_t)be.nid_ + 1;
	for (int t = 0; t < 2; t++) {
		st->head[t].ensure(nidp * 4); st->lsize[t].ensure(nidp * 4);
		HIP_TRY(hipMemsetAsync(st->head[t].p, 0xFF, nidp * 4, s));
		HIP_TRY(hipMemsetAsync(st->lsize[t].p, 0, nidp * 4, s));
	}
	size_t nmax = std::max(n0, n1);
	st->keys.ensure(nmax * 8 + 16); st->skeys.ensure(nmax * 8 + 16); st->selem.ensure(nmax * 4 + 16); st->iota.ensure(nmax * 4 + 16);
	st->nmark.ensure(cap_n * 4);
	for (int t = 0; t < 2; t++) st->maux[t].ensure(16);
	const unsigned ordbits = bits_of(2ull * E), idbits = bits_of(be.nid_);      // sort keys of id_bits + ordbits bits (48 on the benchmark workload: 6 radix passes, not 8)
	for (int t = 0; t < 2; t++) {
		unsigned n = c->nmarks[t];
		if (!n) continue;
		k_instance_keys<<<nblocks(n, 256), 256, 0, s>>>(c->d_melem[t].as<unsigned>(), c->d_mid[t].as<unsigned>(), n, (unsigned)t,
		                                               c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, ordbits, st->keys.as<unsigned long long>(), st->iota.as<unsigned>());
		sort_pairs64(c, st, st->keys.as<unsigned long long>(), st->skeys.as<unsigned long long>(), st->iota.as<unsigned>(), st->selem.as<unsigned>(), n, std::min(64u, idbits + ordbits));
		k_build_lists<<<nblocks(n, 256), 256, 0, s>>>(st->skeys.as<unsigned long long>(), st->selem.as<unsigned>(), c->d_melem[t].as<unsigned>(), n, t ? (unsigned)n0 : 0u, (unsigned)t, ordbits,
		                                             st->nslot.as<unsigned>(), st->nnext.as<unsigned>(), st->nidst.as<unsigned>(), st->ndead.as<uint8_t>(),
		                                             st->head[t].as<unsigned>(), st->lsize[t].as<unsigned>(), st->nodeof[t].as<unsigned>(), st->nmark.as<unsigned>());
		st->maux[t].ensure((size_t)n * 4 + 16);
		k_mark_aux<<<nblocks(n, 256), 256, 0, s>>>(c->d_melem[t].as<unsigned>(), n, (unsigned)t, c->d_sepidx.as<unsigned>(), c->nchr, c->d_ch.as<uint8_t>(), k, st->maux[t].as<unsigned>());
	}
	HIP_TRY(hipGetLastError());
	// positional order of the ids for the snapshot kernel
	st->perm.ensure(nidp * 4 + 16); st->permin.ensure(nidp * 4 + 16);
	st->keys.ensure(nidp * 8 + 16); st->skeys.ensure(nidp * 8 + 16);
	if (be.nid_) {
		k_id_position_keys<<<nblocks(be.nid_, 256), 256, 0, s>>>(st->head[0].as<unsigned>(), st->head[1].as<unsigned>(), st->nslot.as<unsigned>(), be.nid_,
		                                                       st->keys.as<unsigned long long>(), st->permin.as<unsigned>());
		sort_pairs64(c, st, st->keys.as<unsigned long long>(), st->skeys.as<unsigned long long>(), st->permin.as<unsigned>(), st->perm.as<unsigned>(), be.nid_, 32);      // keys are element slots, or 2^32 - 1
	}
	HIP_TRY(hipGetLastError());

	size_t be_maxn = 0;                                                // largest number of instances of an id (below)
	// ---- control state
	st->ctr.ensure(CTR_COUNT * 4);
	st->sel.ensure((1024 + 16) * 4);
	{
		// counter block and the header of the selection scratch (k_select_*: header + one count per chunk of ids; the kernels leave the header
		// reset) from a pinned staging buffer the context keeps: the copies are asynchronous and the host does not wait for them
		if (!st->h_init) HIP_TRY(hipHostMalloc((void **)&st->h_init, (CTR_COUNT + 4) * 4, hipHostMallocDefault));
		HIP_TRY(hipStreamSynchronize(s));                                  // (a previous stage's copies from the buffer have long completed; cheap when idle)
		unsigned *v = reinterpret_cast<unsigned *>(st->h_init);
		memset(v, 0, CTR_COUNT * 4);
		v[CTR_NE] = (unsigned)ne0; v[CTR_NN] = (unsigned)ninst; v[CTR_VIOL] = BT_NONE;
		v[CTR_COUNT] = SBL_NONE; v[CTR_COUNT + 1] = SBL_NONE; v[CTR_COUNT + 2] = 0u; v[CTR_COUNT + 3] = 0u;
		HIP_TRY(hipMemcpyAsync(st->ctr.p, v, CTR_COUNT * 4, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(st->sel.p, v + CTR_COUNT, 16, hipMemcpyHostToDevice, s));
	}
	st->need.ensure(nidp); st->big.ensure(nidp); st->touch.ensure(nidp); st->own.ensure(nidp * 4);
	HIP_TRY(hipMemsetAsync(st->need.p, 0, nidp, s));
	HIP_TRY(hipMemsetAsync(st->big.p, 0, nidp, s));
	HIP_TRY(hipMemsetAsync(st->touch.p, 0, nidp, s));
	// scratch arena per window entry: window caches of ~16 instances (17 B per step, D + k + 2 steps) + FillVisit / Overlap
	// buffers + the AnyBulges map; ids that need more run alone in the big arena
	{
		// instances per id: the arena holds the window caches of an id with up to 1.5x the typical maximum (ids with
		// more -- repeat families -- run alone in the big arena)
		unsigned maxn = 0;
		HIP_TRY(hipMemsetAsync(st->ctr.as<unsigned>() + CTR_BIG, 0, 4, s));
		if (be.nid_) k_max_instances<<<256, 256, 0, s>>>(st->lsize[0].as<unsigned>(), st->lsize[1].as<unsigned>(), be.nid_, st->ctr.as<unsigned>() + CTR_BIG);
		HIP_TRY(hipMemcpyAsync(&maxn, st->ctr.as<unsigned>() + CTR_BIG, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		HIP_TRY(hipMemsetAsync(st->ctr.as<unsigned>() + CTR_BIG, 0, 4, s));
		be_maxn = maxn;
		size_t slots = std::min<size_t>(std::max<size_t>(16, maxn + maxn / 2), 2048);
		size_t ws = (size_t)D + k + 2;
		size_t need = slots * 17 * ws + 12 * (size_t)D + 8 * (size_t)k + (64u << 10) + slots * 64;
		be.arena_bytes = (uint32_t)std::min<size_t>(std::max<size_t>(need, 128u << 10), 1u << 30);
		be.snap_arena_bytes = be.arena_bytes;
		be.snap_threads = (uint32_t)std::max<size_t>(256, std::min<size_t>(256 * 32, (16ull << 30) / be.arena_bytes)) & ~7u;   // a multiple of the 8 XCDs
		be.big_arena_bytes = (uint32_t)std::min<size_t>(std::max<size_t>(256u << 20, 64 * be.arena_bytes), 0xFFFFFF00u);
	}
	uint32_t base_window = 14336;                                        // (swept again at the end of round 3: 86.1 ms against 86.9 ms at 16 384, 62 strains 4.12 against 4.19 s)
	if (const char *e = getenv("SBL_BASE_WINDOW")) base_window = (uint32_t)std::max(64, atoi(e));      // measurement switch (tools/sweep_window.sh)
	uint32_t window = c->window ? c->window : std::min<uint32_t>(base_window, std::max<uint32_t>(2048, be.nid_ / 64));
	window = std::min<uint32_t>(window, (1u << 20) - 1);
	window = (uint32_t)std::min<size_t>(window, std::max<size_t>(64, (24ull << 30) / be.arena_bytes));
	window = std::max<uint32_t>(1, std::min<uint32_t>(window, be.nid_ ? be.nid_ : 1));
	be.window = window;
	// the driver widens the window up to 4x while rounds are capacity-bound (simplify_driver.h); an explicit sbl_set_window pins it
	uint32_t window_max = c->window ? window : (uint32_t)std::min<size_t>((size_t)window * 4, std::max<size_t>(window, (48ull << 30) / be.arena_bytes));
	window_max = std::max<uint32_t>(window, std::min<uint32_t>(window_max, be.nid_ ? be.nid_ : 1));
	auto round_buffers = [&](uint32_t w) {
		st->win.ensure((size_t)w * 4 + 16);
		st->arena.ensure((size_t)w * be.arena_bytes);
		st->claims.ensure((size_t)w * (CLAIM_CAP + 1) * 4);
		st->live.ensure((size_t)w + 64);
		st->instbuf.ensure((size_t)w * 129 * 4);                          // (DeviceBackend::istride() <= 129)
	};
	be.snap_slice = window_max;
	try { if (!dense) round_buffers(window_max); }
	catch (const SblError &) {                                        // a smaller or partly occupied GPU: the pinned window always was enough
		if (window_max == window) throw;
		(void)hipGetLastError();
		window_max = window;
		be.snap_slice = window;
		round_buffers(window);
	}
	for (auto &e : st->ev) if (!e) HIP_TRY(hipEventCreate(&e));
	be.ev = st->ev;
	be.optimistic = optimistic;
	be.rsv_waves = ninst > 12 * (size_t)std::max<uint32_t>(1, be.nid_) ? 4u : 2u;      // instances per id: a handful, or dozens (many strains)
	if (const char *e = getenv("SBL_RSV_WAVES")) be.rsv_waves = (unsigned)std::min(4, std::max(1, atoi(e)));      // measurement switch
	if (getenv("SBL_PROBE_BIG_LDS") == nullptr) {
		// (the 512-slot table stays: a probe it cannot hold goes to the walking kernel, whose launch then lasts as long as a full probe -- 15 such
		// entries per round cost more than the 2 KB save; the instance and mark lists follow the input: largest instance count, mark density)
		be.pidx_inst = (unsigned)std::min<size_t>(256, std::max<size_t>(64, (be_maxn + 15) / 16 * 16));
		be.pidx_marks = ninst * 8 > E ? 192u : 64u;
	}
	be.ev_phase = c->stage_seq++;
	be.prof = getenv("SBL_PHASES") ? 1 : 0;
	if (be.prof) {
		unsigned long long z[64] = {0};
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, 24 * 8));
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_txn_hist), z, 64 * 8));
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_txn_max), z, 16));
		{ std::vector<unsigned long long> zz(4096, 0); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_round_max), zz.data(), 4096 * 8)); }
		{ std::vector<unsigned long long> zz(4096 * 2, 0); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_round_few), zz.data(), 4096 * 2 * 8)); }
		{ std::vector<unsigned long long> zz(4096 * 3, 0); for (unsigned r = 0; r < 4096; r++) zz[3 * r] = ~0ull; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_round_span), zz.data(), 4096 * 3 * 8)); }
	}
	if (dense) be.use_index = false;                                   // (the one-launch path reads no index: nothing to maintain)
	be.idx_nblk = be.use_index ? (uint32_t)((E + 63) / 64) : 0u;
	if (be.idx_nblk) st->bidx.ensure((size_t)be.idx_nblk * BT_IDX_WORDS * 8);
	be.bind();
	be.g.k = k; be.g.D = D;
	be.g.test_flags = getenv("SBL_TEST_FLAGS") ? (unsigned)atoi(getenv("SBL_TEST_FLAGS")) : 0u;
	be.g.lazy_rescan = getenv("SBL_EAGER_RESCAN") ? 0u : 1u;            // measurement switch: dirty windows rescanned right after every collapse (round 3)
	be.g.collapse_g = getenv("SBL_OLD_COLLAPSE") ? 0u : 1u;            // measurement switch: the round-3 collapse (a chain of dependent round trips) instead of the gather-first one
	be.g.ab_estimate = getenv("SBL_NO_AB_ESTIMATE") ? 0u : 1u;         // measurement switch: AnyBulges of big ids without its counting pass
	be.g.jscan_rounds = getenv("SBL_NO_JSCAN_ROUNDS") ? 0u : 1u;       // measurement switch
	be.g.probe_pre = getenv("SBL_NO_PROBE_PRE") ? 0u : 1u;              // measurement switch: the endChar pre-pass of the probe (probe_endchars)
	be.g.lazy_map = getenv("SBL_EAGER_MAP") ? 0u : 1u;                  // measurement switch: the Boost-ordered map of AnyBulges built eagerly (round 3)
	be.g.tstamp = nullptr; be.g.tslot = 0;
	be.g.sep = c->d_sepidx.as<unsigned>(); be.g.nsep = c->nchr + 1; be.g.norig = (uint32_t)E;
	if (getenv("SBL_SEP_BY_CHAR")) be.g.sep = nullptr;                   // measurement switch: separators recognised by their character everywhere
	be.index_build();
	if (!dense && be.phase_events) be.stamps_init();
	HIP_TRY(hipEventRecord(c->ev[3], s));

	// ---- SimplifyGraph
	SimplifyReport rep;
	if (dense) {
		// tiny input: for iteration, for id, RemoveBulges(id) in one launch (k_dense_stage)
		be.g.lazy_min = 1;                                            // every id keeps full-size mark lists and takes lazy windows
		st->big_arena.ensure(be.big_arena_bytes);
		unsigned *d_out = st->ctr.as<unsigned>() + CTR_DETAIL;        // (the violation-detail words are unused here)
		HIP_TRY(hipEventRecord(be.ev[2], s));
		k_dense_stage<<<1, 64, 0, s>>>(be.g, st->big_arena.as<uint8_t>(), be.big_arena_bytes, max_iter, d_out);
		HIP_TRY(hipEventRecord(be.ev[3], s));
		HIP_TRY(hipGetLastError());
		be.read_ctr();
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, be.ev[2], be.ev[3]));
		be.commit_ms = ms;
		if (st->h_ctr[CTR_ERR]) {
			if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] one-launch path: capacity error %u, falling back to the ordered rounds\n", st->h_ctr[CTR_ERR]);
			return RUN_DENSE_FAILED;
		}
		rep.iterations = st->h_ctr[CTR_DETAIL]; rep.bulges = st->h_ctr[CTR_BULGES]; rep.transactions = rep.executed = st->h_ctr[CTR_TXN];
		rep.chain_transactions = rep.transactions;
		if (progress) {                                               // the reference's callback sequence (blockfinder.cpp:23-48), delivered after the launch
			progress(0, SBL_PROGRESS_START, user);
			const uint64_t threshold = ((uint64_t)be.nid_ * max_iter) / 50, processed = (uint64_t)rep.iterations * ((uint64_t)be.nid_ + 1);
			const uint64_t due = threshold ? processed / threshold : processed;
			uint64_t tp = 0;
			for (uint64_t i = 0; i < due; i++) { tp = std::min<uint64_t>(tp + 1, 50); progress((size_t)tp, SBL_PROGRESS_RUN, user); }
			progress(50, SBL_PROGRESS_END, user);
		}
	} else {
		try { rep = simplify_graph(be, max_iter, window, progress, user, window_max); }
		catch (const RestartStage &) { HIP_TRY(hipStreamSynchronize(s)); return RUN_RESTART; }
	}
	HIP_TRY(hipEventRecord(c->ev[4], s));
	if (!dense && be.phase_events) be.stamps_collect();
	be.snapshots_collect();
	if (be.g.bidx && (be.g.test_flags & 32u)) {                          // SBL_TEST_FLAGS=32: what the block index served
		unsigned z[8];
		HIP_TRY(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_idx_stats), sizeof z));
		fprintf(stderr, "[sbl] block index: probes known-live %u, < 2 instances %u, clean %u, live %u, table full %u, not served %u; reservation instances served %u, walked %u\n", z[0], z[1], z[2], z[3], z[4], z[5], z[6], z[7]);
		memset(z, 0, sizeof z);
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_idx_stats), z, sizeof z));
		unsigned long long rz[8];
		HIP_TRY(hipMemcpyFromSymbol(rz, HIP_SYMBOL(g_rsv_ticks), sizeof rz));
		if (rz[3]) fprintf(stderr, "[sbl] reservations: %llu entries, %.1f claims and %.1f instances each; per entry (10 ns ticks of the device wall clock): set-up %.0f, records %.0f, exclusive claims %.0f, ordering claims + wait for the other waves %.0f, walked instances %.0f\n",
		                   rz[3], (double)rz[4] / rz[3], (double)rz[5] / rz[3], (double)rz[0] / rz[3], (double)rz[6] / rz[3], (double)rz[1] / rz[3], (double)rz[2] / rz[3], (double)rz[7] / rz[3]);
		memset(rz, 0, sizeof rz);
		HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_rsv_ticks), rz, sizeof rz));
	}
	if (be.g.bidx && getenv("SBL_CHECK_INDEX")) {                        // test switch: the maintained block index against a rebuild
		unsigned init[2] = {0u, BT_NONE}, res[2];
		unsigned *d_out = st->ctr.as<unsigned>() + CTR_DETAIL + 16;
		HIP_TRY(hipMemcpyAsync(d_out, init, sizeof init, hipMemcpyHostToDevice, s));
		k_check_blkidx<<<(be.idx_nblk + 3) / 4, 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), st->pv.as<unsigned>(), c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>(),
		                                                    st->wmax.as<unsigned>(), be.g.norig, be.idx_nblk, st->bidx.as<unsigned long long>(), d_out);
		HIP_TRY(hipMemcpyAsync(res, d_out, sizeof res, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] block index check: %u of %u blocks differ from a rebuild\n", res[0], be.idx_nblk);
		if (res[0]) { char b[160]; snprintf(b, sizeof b, "block index out of date: %u of %u blocks differ from a rebuild (first: block %u)", res[0], be.idx_nblk, res[1]); throw SblError{SBL_ERR_INTERNAL, b}; }
	}

	// ---- T3: copy-back (reference src/blockfinder.cpp:85-95): linearise the list into the dense state arrays
	be.read_ctr();
	unsigned ne = st->h_ctr[CTR_NE];
	st->flag.ensure((size_t)ne * 4 + 16); st->segidx.ensure((size_t)ne * 4 + 16); st->newidx.ensure((size_t)ne * 4 + 16);
	k_seg_flags<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>());
	scan_u32(c, st, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), (size_t)ne + 1);
	unsigned nseg = 0;
	HIP_TRY(hipMemcpyAsync(&nseg, st->segidx.as<unsigned>() + ne, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	SBL_CHECK(nseg >= 1, SBL_ERR_INTERNAL, "copy-back: empty list");
	st->seg_head.ensure((size_t)nseg * 4); st->seg_len.ensure((size_t)nseg * 4); st->seg_succ_elem.ensure((size_t)nseg * 4);
	for (int t = 0; t < 2; t++) { st->succ[t].ensure((size_t)nseg * 4); st->dist[t].ensure((size_t)nseg * 8); }
	k_seg_tails<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(),
	                                            st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>());
	k_seg_finish<<<nblocks(nseg, 256), 256, 0, s>>>(nseg, st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>(),
	                                               st->flag.as<unsigned>(), st->segidx.as<unsigned>(), st->succ[0].as<unsigned>(), st->dist[0].as<unsigned long long>());
	int cur = 0;
	for (unsigned span = 1; span < nseg; span <<= 1, cur ^= 1)
		k_seg_jump<<<nblocks(nseg, 256), 256, 0, s>>>(nseg, st->succ[cur].as<unsigned>(), st->dist[cur].as<unsigned long long>(),
		                                             st->succ[cur ^ 1].as<unsigned>(), st->dist[cur ^ 1].as<unsigned long long>());
	// segment 0 starts with element 0 (the first '$'), the head of the whole list: dist[0] = new total length
	unsigned long long total = 0;
	HIP_TRY(hipMemcpyAsync(&total, st->dist[cur].p, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	size_t Enew = (size_t)total, Epad = (Enew + 31) / 32 * 32 + 64;
	st->ch_out.ensure(Epad); st->op_out.ensure(Enew * 4 + 16);
	k_scatter_linear<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), st->op.as<unsigned>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(),
	                                                 st->seg_head.as<unsigned>(), st->dist[cur].as<unsigned long long>(), total,
	                                                 st->ch_out.as<uint8_t>(), st->op_out.as<unsigned>(), st->newidx.as<unsigned>());
	k_fill_bytes<<<nblocks(Epad - Enew, 256), 256, 0, s>>>(st->ch_out.as<uint8_t>(), (uint8_t)'$', Enew, Epad);
	k_remap_seps<<<nblocks(c->nchr + 1, 64), 64, 0, s>>>(st->newidx.as<unsigned>(), c->d_sepidx.as<unsigned>(), c->nchr + 1);
	k_sep_positions<<<nblocks(c->nchr, 64), 64, 0, s>>>(c->d_sepidx.as<unsigned>(), c->nchr, st->op_out.as<unsigned>());
	HIP_TRY(hipGetLastError());
	c->stats.dict_checked = 0; c->stats.dict_mismatches = 0;
	if (getenv("SBL_CHECK_DICTIONARY") && c->dict_keys && k <= 32) {
		// the reference's own invariant (IndexedSequence::Test) on the stage's final graph, see k_dict_check
		st->scantmp.ensure(64);
		unsigned long long init[6] = {0, 0, ~0ull, 0, 0, 0}, res[6];
		HIP_TRY(hipMemcpyAsync(st->scantmp.p, init, sizeof init, hipMemcpyHostToDevice, s));
		if (getenv("SBL_TEST_CORRUPT_MARK")) {                              // test hook: the check must notice ONE wrong mark among hundreds of millions
			const unsigned wrong = 0;
			HIP_TRY(hipMemcpyAsync(c->d_bif[0].as<unsigned>() + (c->sepidx[0] + 1 + (size_t)atoll(getenv("SBL_TEST_CORRUPT_MARK"))), &wrong, 4, hipMemcpyHostToDevice, s));
		}
		k_dict_check<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), ne, st->newidx.as<unsigned>(), st->ch_out.as<uint8_t>(), total, c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>(),
		                                             c->dict_keys, be.nid_, k, st->scantmp.as<unsigned long long>());
		HIP_TRY(hipMemcpyAsync(res, st->scantmp.p, sizeof res, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		c->stats.dict_checked = res[0]; c->stats.dict_mismatches = res[1];
		if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] dictionary invariant (IndexedSequence::Test): %llu windows checked, %llu mismatches\n", res[0], res[1]);
		if (res[1] && !getenv("SBL_TEST_CORRUPT_MARK")) {
			char b[256];
			snprintf(b, sizeof b, "dictionary invariant violated (IndexedSequence::Test): %llu of %llu windows; first at slot %llu strand %llu: stored id %llu, the dictionary says %llu",
			         res[1], res[0], res[2], res[3], res[4], res[5]);
			throw SblError{SBL_ERR_INTERNAL, b};
		}
	}
	std::swap(c->d_ch, st->ch_out);
	std::swap(c->d_op, st->op_out);
	HIP_TRY(hipMemcpyAsync(c->sepidx.data(), c->d_sepidx.p, (size_t)(c->nchr + 1) * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipEventRecord(c->ev[5], s));
	HIP_TRY(hipStreamSynchronize(s));
	c->nelem = Enew;

	float ms_enum = 0, ms_simp = 0, ms_copy = 0;
	HIP_TRY(hipEventElapsedTime(&ms_enum, c->ev[2], c->ev[3]));
	HIP_TRY(hipEventElapsedTime(&ms_simp, c->ev[3], c->ev[4]));
	HIP_TRY(hipEventElapsedTime(&ms_copy, c->ev[4], c->ev[5]));
	c->stats.enumerate_ms = ms_enum; c->stats.simplify_ms = ms_simp; c->stats.copyback_ms = ms_copy;
	c->stats.total_ms = ms_enum + ms_simp + ms_copy;
	c->stats.bulges = rep.bulges; c->stats.iterations = rep.iterations; c->stats.rounds = rep.rounds; c->stats.replays = rep.replays; c->stats.grow_replays = rep.grow_replays;
	c->stats.snapshot_ms = be.snapshot_ms; c->stats.reserve_ms = be.reserve_ms; c->stats.commit_ms = be.commit_ms; c->stats.probe_ms = be.probe_ms;
	c->stats.commit_event_ms = be.commit_event_ms; c->stats.commit_event_launches = be.commit_event_launches;
	c->stats.verdict_ms = be.ro_ms; c->stats.ro_ranks = be.split_ro() ? c->comm->n : 1;
	c->stats.executed = rep.executed; c->stats.transactions = rep.transactions; c->stats.chain_transactions = rep.chain_transactions;
	if (be.prof) {
		unsigned long long z[24];
		HIP_TRY(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_phase_cycles), sizeof z));
		const char *nm[24] = {"setup", "scan", "rb_begin", "rb_run", "dirty-calc", "collapse", "publish", "(unused)", "rescan",
		                      " c:erase-flanks", " c:erase-span", " c:positions+NE-alloc", " c:replace", " c:copy-marks-data", " c:NN-alloc+stamps", " c:addpoints",
		                      " b:endchars+sizing", " b:map-build", " b:finish", " b:loop-setup", " r:FillVisit", " r:Overlap", " r:multiplicities", " r:J walk + search"};
		for (int i = 0; i < 24; i++) fprintf(stderr, "[sbl] commit phase %-12s %10.3f Mcycles\n", nm[i], z[i] / 1e6);
		unsigned long long hh[4][16], mx[2];
		HIP_TRY(hipMemcpyFromSymbol(hh, HIP_SYMBOL(g_txn_hist), sizeof hh));
		HIP_TRY(hipMemcpyFromSymbol(mx, HIP_SYMBOL(g_txn_max), sizeof mx));
		for (int r = 0; r < 4; r++) {
			fprintf(stderr, "[sbl] transactions with %d%s collapses by duration (bins of 2^k x 8192 cycles):", r, r == 3 ? "+" : "");
			for (int b = 0; b < 16; b++) fprintf(stderr, " %llu", hh[r][b]);
			fprintf(stderr, "\n");
		}
		fprintf(stderr, "[sbl] longest transaction: %llu cycles, %llu instances, %llu collapses\n", mx[0], mx[1] >> 32, mx[1] & 0xFFFFFFFFull);
		{
			std::vector<unsigned long long> rm(4096);
			HIP_TRY(hipMemcpyFromSymbol(rm.data(), HIP_SYMBOL(g_round_max), 4096 * 8));
			{
				// does a launch of k_commit wait for work that started late, or for one long transaction?  (round 4: the slowest transaction of a
				// launch starts ~5 us after the first and IS the launch -- nothing to gain from dispatching long ones first)
				std::vector<unsigned long long> sp(4096 * 3);
				HIP_TRY(hipMemcpyFromSymbol(sp.data(), HIP_SYMBOL(g_round_span), 4096 * 3 * 8));
				std::vector<unsigned long long> fw(4096 * 2);
				HIP_TRY(hipMemcpyFromSymbol(fw.data(), HIP_SYMBOL(g_round_few), 4096 * 2 * 8));
				double span = 0, slow = 0, off = 0, few1 = 0, few2 = 0; unsigned nl = 0;
				for (unsigned r = 0; r < 4096; r++) {
					if (sp[3 * r] == ~0ull || !sp[3 * r + 1]) continue;
					nl++; span += (double)(sp[3 * r + 1] - sp[3 * r]) * 0.01; slow += (double)(sp[3 * r + 2] >> 32) * 0.01;
					off += (double)(unsigned)((unsigned)sp[3 * r + 2] - (unsigned)sp[3 * r]) * 0.01;
					few1 += (double)fw[2 * r] * 0.01; few2 += (double)fw[2 * r + 1] * 0.01;
				}
				fprintf(stderr, "[sbl] slowest transactions with at most one collapse %.1f us in total, with at most two %.1f us\n", few1, few2);
				fprintf(stderr, "[sbl] %u commit launches: owners' span %.1f us in total, slowest transactions %.1f us, their start offsets %.1f us\n", nl, span, slow, off);
			}
			fprintf(stderr, "[sbl] slowest transaction of every launch (kcycles/instances/old-form collapses/collapses):");
			for (unsigned r = 0; r < 4096 && r < be.ts_round; r++) if (rm[r]) fprintf(stderr, " %llu/%llu/%llu/%llu", (rm[r] >> 24) / 1000, (rm[r] >> 16) & 255, (rm[r] >> 8) & 255, rm[r] & 255);
			fprintf(stderr, "\n");
		}
	}
	*bulges = rep.bulges;
	return RUN_DONE;
}
