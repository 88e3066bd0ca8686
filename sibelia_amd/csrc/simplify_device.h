// simplify_device.h -- device-side helpers shared by the kernels of the simplification (snapshot.hip, rounds.hip, commit.hip):
// address-space accessors for the transaction scratch, separators by slot, start stamps, wave-cooperative list and window scans,
// the AnyBulges verdict table.  Header-only (__device__ __forceinline__): every translation unit gets its own copies, nothing is linked
// across units on the device side.
#pragma once
#include "sbl_ctx.h"
#include "kmer_kernels.h"
#include "simplify_steps.h"
#include "simplify_kernels.h"

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
// compare-and-swap on a word of the fast scratch or the arena; returns what was there
__device__ __forceinline__ unsigned casx(unsigned *p, unsigned expect, unsigned v)
{
	unsigned e = expect;
	if (__builtin_amdgcn_is_shared((const void *)p)) __atomic_compare_exchange_n((SBL_AS3 unsigned *)p, &e, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
	else __atomic_compare_exchange_n((SBL_AS1 unsigned *)p, &e, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
	return e;
}
#else       // (the host pass of hipcc only parses the kernels)
__device__ __forceinline__ unsigned casx(unsigned *p, unsigned expect, unsigned v) { unsigned o = *p; if (o == expect) *p = v; return o; }
template <class T> __device__ __forceinline__ T ldg(const T *p) { return *p; }
template <class T> __device__ __forceinline__ void stg(T *p, T v) { *p = v; }
template <class T> __device__ __forceinline__ T ldx(const T *p) { return *p; }
template <class T> __device__ __forceinline__ void stx(T *p, T v) { *p = v; }
#endif

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
__device__ __forceinline__ void wave_stamp(const GraphView &g, unsigned stampv, unsigned tid, unsigned mode, unsigned id, unsigned r, unsigned wm /* wmax[r], loaded with the data */,
                                           bool rstamp = true /* this lane publishes the read stamp of its 64-slot block (one lane per block does) */)
{
	// Exclusivity inside a round needs no per-element lock here: an owner holds every id marked in the range it reserved
	// (2(D+k+2)+k elements ahead of each instance), its scans reach D+k+2 elements, and k_commit checks after every
	// collapse that the elements it has deleted inside a window cannot carry a later scan / push beyond the reserved range.
	bool bad = false;
	unsigned other = BT_NONE;
	(void)stampv;
	if (mode == 2 && rstamp) atomicMax(&g.rmax[r >> BT_RSHIFT], tid);      // (r: an element's block)
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
template <bool PARK_AWARE = false>
__device__ __forceinline__ bool wave_setup(const GraphView &g, Txn &t, BulgeWork &w, bool lite, unsigned lane, int &ok)
{
	const unsigned h0 = g.head[0][t.id], h1 = g.head[1][t.id];              // in flight while lane 0 lays the scratch out
	if (lane == 0) ok = bt_setup(t, w, lite, false) && !t.err ? 1 : 0;
	WSYNC();
	if (!ok) return false;
	unsigned m = wave_list_positions(g, h0, h1, w, lane);
	if (m != w.n && lane == 0) {
		// (list sizes = live nodes between transactions -- except around a PARKED transaction, GraphView::park_of: the nodes it erased stay
		// in their lists, counted, until it is through (Cleanup, bifurcationstorage.cpp:33-41, belongs to the end of RemoveBulges))
		if (PARK_AWARE && g.any_parked && m < w.n) { w.n = m; if (m < 2) ok = 0; }      // (the probes of the ordered rounds: nothing is parked when a snapshot runs)
		else { t.err |= BT_ERR_SCRATCH; ok = 0; }                         // cannot happen on a consistent graph
	}
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
			// (the write stamp of every element is checked; the READ stamp is published once per 64-slot block: by the first lane and wherever the block changes)
			if (st && bst.chv[u] != BT_SEP && (lane == 0 || pb != blk)) wave_stamp(g, stampv, tid, mode, id, blk, bst.wmv[u], lane == 0 || (pb >> BT_RSHIFT) != (blk >> BT_RSHIFT));
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

