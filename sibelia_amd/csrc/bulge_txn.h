// bulge_txn.h -- one bifurcation id's bulge removal as a device "transaction".
//
// Replaces BlockFinder::RemoveBulges and everything it calls
//   (reference src/bulgeremoval.cpp:39-430, src/bifurcationstorage.cpp:113-162, src/dnasequence.cpp:189-252)
// on a GPU-resident graph:
//   * the editable sequence is an index-linked list over flat HBM arrays (stable element identity,
//     O(1) insert/erase) instead of the reference's unrolled list;
//   * bifurcation marks are two dense arrays bif[strand][element] instead of an address-keyed hash set;
//   * per-(strand,id) instance lists are index-linked nodes with front insertion and lazy erase,
//     which is the order the reference's slist produces (bifurcationstorage.cpp:122,144-155);
//   * the iteration order of the reference's boost::unordered_map in AnyBulges is reproduced by a
//     fixed-capacity restatement of Boost 1.54's bucket list (bulgeremoval.cpp:168,203-215).
//
// One GPU thread executes one transaction; thousands run per launch (simplify.hip).  The functions
// are __host__ __device__ so that tests/hostsim can unit-test this exact code without a GPU; the
// shipped library only ever calls them from kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define BT_NONE 0xFFFFFFFFu
#define BT_SEP '$'
#define BT_DEAD_CHAR 0            // ch[] of an element that was erased from the list
#define BT_POS_MASK 0x1FFFFFFFu   // 29-bit original positions (reference src/stranditerator.cpp:19-27)
#define BT_MAX_BREAKS 16u
#ifndef BT_LDS_MARKS
#define BT_LDS_MARKS 48u
#endif
#define BT_BLOCK_SHIFT 0          // validation granularity: single elements (coarser blocks flag neighbours across a chromosome boundary)
// Fresh element slots per insertion are a multiple of this.  Validation works on single elements (BT_BLOCK_SHIFT 0), so an
// insertion takes exactly the slots it needs: low-complexity inputs make tens of thousands of 1-3 element insertions per
// stage and would otherwise burn the element pool 32 slots at a time (grow + replay of the iteration, again and again).
#define BT_INSERT_ALIGN 1u
// Granularity of the READ stamps of elements (rmax: highest id that read an element in its writer pass): rmax[e >> BT_RSHIFT].
// They only serve the check "a lower id must not write what a higher id has read", and a coarser stamp can only report MORE such
// violations -- so round 6 tried one stamp per 64 slots (one atomic per block of a window instead of one per element: 3.7 M atomics and
// 110 MB of counter traffic per launch of k_commit).  NEGATIVE: two transactions of one round own disjoint ids but abut inside a
// 64-slot block all the time; every such pair is a false violation, the attempt replays with a fence, and the next pair is already
// there -- 409 013 rounds on a stage that takes 100.  Read stamps stay per element (0); the shift is kept for the record.
#ifndef BT_RSHIFT
#define BT_RSHIFT 0u
#endif
#define BT_LAZY_MIN 64u           // see BulgeWork::lazy
#define BT_ACT_FAST 64u           // see BulgeWork::act_fast
#define BT_MSCAN_MIN 6u           // see BulgeWork::mscan
__host__ __device__ __forceinline__ uint32_t bt_insert_span(uint32_t m) { return (m + BT_INSERT_ALIGN - 1u) & ~(BT_INSERT_ALIGN - 1u); }

// Counter block.  Every counter has a 128-byte line of its own: thousands of workgroups per launch bump them (retired entries,
// transactions, bulges) or reserve pool ranges through them with RETURNING atomics (element / node slots of a collapse), and all
// atomics on one line queue up behind each other in the L2 (~11 ns each) -- on one shared line a collapse waited for everybody's
// bookkeeping.  CTR_DETAIL .. +4: first violation (kind, resource, other, id, info), debugging only.
enum { CTR_STRIDE = 32,
       CTR_NE = 0 * CTR_STRIDE, CTR_NN = 1 * CTR_STRIDE, CTR_ERR = 2 * CTR_STRIDE, CTR_BULGES = 3 * CTR_STRIDE, CTR_VIOL = 4 * CTR_STRIDE,
       CTR_NWIN = 5 * CTR_STRIDE, CTR_LO = 6 * CTR_STRIDE, CTR_COMMITTED = 7 * CTR_STRIDE, CTR_BIG = 8 * CTR_STRIDE, CTR_PUSHED = 9 * CTR_STRIDE,
       CTR_TXN = 10 * CTR_STRIDE, CTR_DETAIL = 11 * CTR_STRIDE, CTR_COUNT = 12 * CTR_STRIDE };
enum { BT_ERR_SCRATCH = 1, BT_ERR_ELEM_CAP = 2, BT_ERR_NODE_CAP = 4, BT_ERR_LAYOUT = 8 /* a parked image from another LDS layout (commit.hip: park_load) */ };

// optional cycle counters of the decision loops (simplify.hip defines them for SBL_PHASES=1; nothing elsewhere)
#ifndef BT_PROF_ADD
#define BT_PROF_T0(t) ((void)0)
#define BT_PROF_ADD(t, i) ((void)0)
#endif

struct GraphView {
	uint8_t *ch; uint32_t *op, *nx, *pv;
	uint32_t *bif[2], *nodeof[2];
	uint32_t *nslot, *nnext, *nidst, *nclr; uint8_t *ndead;   // nidst: (id << 1) | strand of the node's list; nclr: chain of nodes erased by the running transaction
	uint32_t *head[2], *lsize[2];
	uint32_t *ctr;
	uint32_t cap_e, cap_n;
	uint32_t k, D, nid;                 // ids 0 .. nid-1
	uint8_t *need;                      // need[id] != 0: RemoveBulges(id) must run at its turn
	uint8_t *big;                       // big[id] != 0: needs the large scratch arena (runs alone)
	uint8_t *touch;                     // touch[id] != 0: windows or lists of the id changed since its verdict was last taken
	// reservation (ordered-commit rounds) and order validation, see simplify.hip
	uint32_t *own;                      // per id: round-stamped owner (atomicMin)
	uint32_t *lock, *rmax, *wmax;       // per resource: blocks [0,nblk) then ids [nblk, nblk+nid]
	uint32_t nblk;
	uint32_t round_bits;                // (ROUND_MAX - round) << 20
	const uint32_t *win;                // ids of the current window
	uint32_t lazy_min;                  // a run with more instances than this that has the graph to itself rescans windows on demand (0: default, BT_LAZY_MIN)
	uint32_t test_flags;                // A/B switches of experiments (SBL_TEST_FLAGS)
	uint32_t lazy_rescan;               // ordered rounds: dirty windows are marked stale instead of rescanned at once (BulgeWork::use_stale)
	uint32_t collapse_g;                // ordered rounds / chain: the gather-first collapse (simplify.hip: wave_collapse_g)
	uint32_t ab_estimate;               // AnyBulges of ids with more than 32 instances sizes its tables by an estimate instead of a counting pass (simplify.hip)
	uint32_t jscan_rounds;              // ordered rounds: ids with more than 24 instances hand the search for the next J to 64 lanes (BulgeWork::jscan)
	uint32_t probe_pre;                 // the probe of a round first looks at the endChars alone (simplify.hip: probe_endchars)
	uint32_t lazy_map;                  // the kernels' AnyBulges logs its map insertions and builds the Boost-ordered map only for calls with >= 2 groups (ABuild::lazy)
	uint32_t test_lazy_map;             // tests/hostsim: bt_any_bulges (one thread) builds its map lazily too (ABuild::lazy), look-ups by linear search
	// start stamps of the round kernels (device wall clock; simplify.hip: DeviceBackend::stamp_*): 4 slots per round, nullptr = off
	unsigned long long *tstamp; uint32_t tslot;
	// the separators' slots (ascending; they never move during a stage) and the number of original slots: a walk that starts at an
	// original slot recognises the separators of its chromosome by their slot instead of loading characters (simplify.hip: SepBounds)
	const uint32_t *sep; uint32_t nsep, norig;
	// Block index over the ORIGINAL slots (round 5): four 64-bit words per block of 64 consecutive slots --
	//   [0] / [1]  bit i: slot 64 b + i carries a mark on strand 0 / 1 (bif[s][e] != BT_NONE)
	//   [2]        bit i: the slot is a separator (never changes during a stage)
	//   [3]        low half: highest write stamp (wmax) of any element of the block; high half: != 0 once the block is no longer
	//              PRISTINE (an element died, or a link of / into the block stopped pointing at the neighbouring slot)
	// A window that lies in pristine blocks is the slots a, a +- 1, ... themselves: the probe and the reservation of a round read three
	// or four 32-byte records per window and gather only the marked ids instead of walking 175 x (character + mark + link + stamp).
	// Maintained at the mark write sites (AddPoint / ErasePoint: bt_idx_mark), where a collapse changes links (bt_idx_dirty) and
	// where write stamps are published (bt_idx_wstamp); nullptr = no index (every reader then takes the walking path).
	unsigned long long *bidx;
	uint32_t idx_probe, idx_reserve;    // the probe / the reservation of a round read the index (simplify.hip: k_probe_idx, reserve_idx)
	// Parked transactions (round 5, commit.hip): a launch of k_commit lasts as long as its slowest transaction, and that is one with
	// several collapses.  A transaction that has made park_cap collapses in a launch and has decided another one PARKS: its LDS state goes
	// to the end of its arena slice, the id stays pending (the probes take a parked id for live without a verdict), and the next round
	// in which it owns its claims resumes it where it stopped (k_resume, beside k_commit on a second stream).  park_of[id]: 0 = never
	// parked; bits 0-19 = its arena slice + 1, bits 20-30 = the round it parked in (it is resumed in a LATER round: both kernels have a
	// workgroup for every window entry, and an entry must be one kernel's for the whole launch), bit 31 = finished in the round of bits
	// 20-30 (k_commit leaves it alone in that round, afterwards it is an ordinary id again).  slice_busy[w] != 0: the slice of window
	// position w holds a parked transaction; the entry at that position works in the shadow slice shadow_base + w (and does not park: it
	// may be a LOWER id that the parked one has to wait for).  park_cap = 0: off.
	uint32_t *park_of; uint8_t *slice_busy; uint32_t park_cap, shadow_base;      // (shadow_base + w: the spare slice of window position w)
	// Round 6: park_hold != 0 -- nothing NEW parks (the driver has asked for the serial chain, which starts once what is parked has drained;
	// parked transactions still resume, and run to their end).  any_parked != 0 -- something was parked when this launch started (the host
	// knows from the counters of the round before): only then may a list hold fewer live nodes than its size says (the deferred Cleanup
	// of a parked transaction); otherwise that is list corruption and stays a hard error.  park_list: window positions of the parked
	// entries of this round (appended by k_reserve, ctr[CTR_PLIST] of them): k_resume's grid is the parked transactions, not the window.
	uint32_t park_hold, any_parked; uint32_t *park_list;
};
// The LDS state of a parked transaction (Txn first) sits in the last PARK_IMG bytes of its arena slice (commit.hip: park_store / park_load).
// Cleanup is part of a transaction's END (bifurcationstorage.cpp:33-41; CountBifurcations, :71-75, counts erased instances until then --
// the transaction itself reads such sizes, bt_max_mult_pair), so while a transaction is parked the ids it has erased instances of carry
// list sizes that are one transaction's private view.  The marks are gone from the graph, so no neighbourhood walk finds those ids any
// more: the reservation of a parked entry claims them from the image's erase chain (Txn::tc_head through nclr, k_reserve) -- exclusively,
// every round until the transaction is through -- and whoever would read such a size waits, exactly as for an id inside the core.
// (Found by tools/stress.py seed 93194: a neighbour read a size one too high, collapsed in the other direction and counted one bulge less.)
#define PARK_IMG 12288u
#define CTR_PARKED (CTR_DETAIL + 8)      // parked transactions at the moment
#define CTR_PLIST (CTR_DETAIL + 10)      // entries of GraphView::park_list in this round (reset behind every round by k_select_write)
// (the tag is 11 bits of a 12-bit round: finished markers are swept every 1024 rounds, DeviceBackend::commit, so that none survives to the round with the same tag)
__host__ __device__ __forceinline__ uint32_t bt_round_tag(const GraphView &g) { return (g.round_bits >> 20) & 0x7FFu; }
// index of a resource's read stamp: element resources (r < nblk) by 64-slot block, id resources as they are
__host__ __device__ __forceinline__ uint32_t bt_ridx(const GraphView &g, uint32_t r) { return r < g.nblk ? r >> BT_RSHIFT : r; }
__host__ __device__ __forceinline__ uint32_t bt_ridx_elem(uint32_t e) { return (e >> BT_BLOCK_SHIFT) >> BT_RSHIFT; }
__host__ __device__ __forceinline__ bool bt_parked(const GraphView &g, uint32_t id) { if (!g.park_of) return false; const uint32_t pk = g.park_of[id]; return pk != 0 && !(pk >> 31); }

// ------------------------------------------------------------------------------------------- atomics (host + device)
__host__ __device__ __forceinline__ uint32_t bt_atomic_add(uint32_t *p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
__host__ __device__ __forceinline__ uint32_t bt_atomic_min(uint32_t *p, uint32_t v) { return __atomic_fetch_min(p, v, __ATOMIC_RELAXED); }
__host__ __device__ __forceinline__ uint32_t bt_atomic_max(uint32_t *p, uint32_t v) { return __atomic_fetch_max(p, v, __ATOMIC_RELAXED); }
__host__ __device__ __forceinline__ uint32_t bt_atomic_or(uint32_t *p, uint32_t v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }

// ------------------------------------------------------------------------------------------- block index maintenance
#define BT_IDX_WORDS 4u
__host__ __device__ __forceinline__ void bt_idx_mark(const GraphView &g, uint32_t s, uint32_t e, bool set)
{
	if (!g.bidx || e >= g.norig) return;
	unsigned long long *w = &g.bidx[(size_t)(e >> 6) * BT_IDX_WORDS + s];
	const unsigned long long bit = 1ull << (e & 63u);
	if (set) (void)__atomic_fetch_or(w, bit, __ATOMIC_RELAXED); else (void)__atomic_fetch_and(w, ~bit, __ATOMIC_RELAXED);
}
__host__ __device__ __forceinline__ void bt_idx_dirty(const GraphView &g, uint32_t e)
{
	if (!g.bidx || e >= g.norig) return;
	reinterpret_cast<uint32_t *>(&g.bidx[(size_t)(e >> 6) * BT_IDX_WORDS + 3])[1] = 1u;
}
__host__ __device__ __forceinline__ void bt_idx_wstamp(const GraphView &g, uint32_t e, uint32_t tid)
{
	if (!g.bidx || e >= g.norig) return;
	(void)__atomic_fetch_max(reinterpret_cast<uint32_t *>(&g.bidx[(size_t)(e >> 6) * BT_IDX_WORDS + 3]), tid, __ATOMIC_RELAXED);
}

// ------------------------------------------------------------------------------------------- strand iterators
struct SIt { uint32_t e; uint32_t d; };   // element + direction (0 positive, 1 negative); reference src/stranditerator.cpp

__host__ __device__ __forceinline__ char bt_comp(char c) { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c; }

#if defined(BT_HOST_TRACE) && !defined(__HIP_DEVICE_COMPILE__)
#include <stdio.h>
#define BT_TRACE_VIOL(what, r, a, b) fprintf(stderr, "[viol] id %u mode %u %s res %u (%s %u) other %u %u\n", id, mode, what, r, \
	(r) < g.nblk ? "block" : "id", (r) < g.nblk ? (r) : (r) - g.nblk, (unsigned)(a), (unsigned)(b))
#else
#define BT_TRACE_VIOL(what, r, a, b) ((void)0)
#endif

struct Txn {
	GraphView g;
	uint32_t id, stamp;                 // stamp = round_bits | window index
	uint32_t tid;                       // id + 1, the value written to rmax / wmax
	// 0: no validation (snapshot verdicts, solo runs)
	// 1: read-only pass of a round: exclusive block locks + "nothing I read was written by a higher id"
	// 2: writer pass: additionally publishes its reads / writes in rmax / wmax
	// 3: probe (no writer runs concurrently): only "nothing I read was written by a higher id"
	uint32_t mode;
	uint32_t last_r, last_w;            // one-entry caches of the last stamped blocks
	uint8_t *scr; uint32_t scr_cap, scr_used;
	uint8_t *fscr; uint32_t fscr_cap, fscr_used;   // optional small fast scratch (LDS in the kernels); falloc never fails loudly
	uint32_t err;
	uint32_t tc_head;                   // lazy erase chain (BifurcationStorage::toClear_), linked through g.nclr
	uint32_t *tc_list; uint32_t tc_cap, tc_n;   // the same nodes as a flat list when the caller runs Cleanup with 64 lanes (defer_cleanup)
	bool defer_cleanup;
	bool wrote;                         // the graph has been modified by this transaction
	bool defer_push;                    // the caller performs bt_push_neighbourhood's work itself (64 lanes, simplify.hip)
	bool ext_stamps;                    // element stamps are done by the caller's wave-wide scans (reads) and post-collapse pass (writes)
	bool chain;                         // serial chain (simplify.hip: k_chain): nothing else is in flight and no reservation exists to be checked
	uint32_t push_e, push_d, push_len;  // ... for this target instance / new branch length
	bool prof; unsigned long long prof_t;      // SBL_PHASES=1 (BT_PROF_ADD)

	__host__ __device__ void init(const GraphView &gv, uint32_t id_, uint32_t widx, uint32_t mode_, uint8_t *arena, uint32_t arena_bytes)
	{
		g = gv; id = id_; tid = id_ + 1; stamp = gv.round_bits | widx; mode = mode_;
		last_r = last_w = BT_NONE; scr = arena; scr_cap = arena_bytes; scr_used = 0; fscr = nullptr; fscr_cap = 0; fscr_used = 0; err = 0; tc_head = BT_NONE; tc_list = nullptr; tc_cap = 0; tc_n = 0; defer_cleanup = false; wrote = false; defer_push = false; ext_stamps = false; chain = false; push_e = BT_NONE; push_d = 0; push_len = 0; prof = false; prof_t = 0;
	}
	// ---- scratch
	__host__ __device__ __forceinline__ void *alloc(uint32_t bytes)
	{
		uint32_t a = (scr_used + 7u) & ~7u;
		if (a + bytes > scr_cap || a + bytes < a) { err |= BT_ERR_SCRATCH; return nullptr; }
		scr_used = a + bytes;
		return scr + a;
	}
	__host__ __device__ __forceinline__ void *falloc(uint32_t bytes)
	{
		uint32_t a = (fscr_used + 7u) & ~7u;
		if (!fscr || a + bytes > fscr_cap || a + bytes < a) return nullptr;
		fscr_used = a + bytes;
		return fscr + a;
	}
	// fast scratch if it fits, the arena otherwise
	__host__ __device__ void *alloc2(uint32_t bytes) { void *p = falloc(bytes); return p ? p : alloc(bytes); }
	// ---- order validation (simplify.hip explains the protocol)
	__host__ __device__ void violation(uint32_t other_prio)
	{
		uint32_t x = id;
		if (other_prio != BT_NONE && g.win) { uint32_t o = g.win[other_prio]; if (o < x) x = o; }
		bt_atomic_min(&g.ctr[CTR_VIOL], x);
	}
	__host__ __device__ void stamp_res(uint32_t r, bool write)
	{
		if (mode == 3) { if (g.wmax[r] > tid) { BT_TRACE_VIOL("probe-read-after-higher-write", r, g.wmax[r], 0); violation(BT_NONE); } return; }
		uint32_t old = bt_atomic_min(&g.lock[r], stamp);
		if (old != stamp && (old >> 20) == (stamp >> 20)) { BT_TRACE_VIOL("lock", r, old, 0); violation(old & 0xFFFFFu); }   // two transactions of one round share r
		if (mode == 1) { if (g.wmax[r] > tid) { BT_TRACE_VIOL("read-after-higher-write", r, g.wmax[r], 0); violation(BT_NONE); } return; }
		if (write) {
			uint32_t a = bt_atomic_max(&g.wmax[r], tid);
			if (r < g.nblk) bt_idx_wstamp(g, r << BT_BLOCK_SHIFT, tid);
			if (a > tid || g.rmax[bt_ridx(g, r)] > tid) { BT_TRACE_VIOL("write-after-higher-access", r, a, g.rmax[bt_ridx(g, r)]); violation(BT_NONE); }
		} else {
			bt_atomic_max(&g.rmax[bt_ridx(g, r)], tid);
			if (g.wmax[r] > tid) { BT_TRACE_VIOL("wread-after-higher-write", r, g.wmax[r], 0); violation(BT_NONE); }
		}
	}
	// separators are immutable and shared by the two chromosomes they delimit: never stamped
	__host__ __device__ __forceinline__ void tr(uint32_t e)     // element read
	{
		if (!mode || ext_stamps || g.ch[e] == BT_SEP) return;
		uint32_t b = e >> BT_BLOCK_SHIFT;
		if (b == last_r || b == last_w) return;
		last_r = b; stamp_res(b, false);
	}
	__host__ __device__ __forceinline__ void tw(uint32_t e)     // element write
	{
		wrote = true;
		if (!mode || ext_stamps || g.ch[e] == BT_SEP) return;
		uint32_t b = e >> BT_BLOCK_SHIFT;
		if (b == last_w) return;
		last_w = b; stamp_res(b, true);
	}
	// Id resources in the wave kernels (ext_stamps): the transaction OWNS every id it may touch (its claims), so exclusivity
	// inside the round is checked against own[] with a plain load instead of a returning atomic on lock[].
	__host__ __device__ void stamp_id_light(uint32_t b, bool write)
	{
		uint32_t r = g.nblk + b;
		uint32_t ow = g.own[b], wm = g.wmax[r], rm = write ? g.rmax[r] : 0u;
		if (mode != 3 && !chain && ow != stamp) violation(BT_NONE);            // escaped its reservation
		if (wm > tid || rm > tid) violation(BT_NONE);
		if (mode == 2) { if (write) bt_atomic_max(&g.wmax[r], tid); else bt_atomic_max(&g.rmax[r], tid); }
	}
	__host__ __device__ __forceinline__ void ir(uint32_t b) { if (!mode) return; if (ext_stamps) stamp_id_light(b, false); else stamp_res(g.nblk + b, false); }   // id (list / count) read
	__host__ __device__ __forceinline__ void iw(uint32_t b) { if (!mode) return; if (ext_stamps) stamp_id_light(b, true); else stamp_res(g.nblk + b, true); }

	// ---- iterator primitives
	__host__ __device__ __forceinline__ SIt next(SIt a) { tr(a.e); a.e = a.d ? g.pv[a.e] : g.nx[a.e]; return a; }     // operator++ :117-130
	__host__ __device__ __forceinline__ SIt adv(SIt a, uint32_t n) { while (n--) a = next(a); return a; }
	__host__ __device__ __forceinline__ SIt inv(SIt a)                                                                // Invert :192-200
	{ tr(a.e); SIt r; if (a.d == 0) { r.e = g.pv[a.e]; r.d = 1; } else { r.e = g.nx[a.e]; r.d = 0; } return r; }
	__host__ __device__ __forceinline__ char chr(SIt a) { tr(a.e); char c = (char)g.ch[a.e]; return a.d ? bt_comp(c) : c; }   // operator* :202-210
	__host__ __device__ __forceinline__ bool valid(SIt a) { tr(a.e); return g.ch[a.e] != BT_SEP; }                     // AtValidPosition :97-100
	__host__ __device__ __forceinline__ uint32_t getbif(SIt a) { tr(a.e); return g.bif[a.d][a.e]; }                    // GetBifurcation, bifurcationstorage.cpp:157

	// a changed instance list makes that id's verdict stale: it must (re)run at its turn in this iteration
	__host__ __device__ __forceinline__ void push_dirty(uint32_t b)
	{ if (b < g.nid) { g.touch[b] = 1; if (b > id) g.need[b] = 1; } }

	// AddPoint, bifurcationstorage.cpp:113-126
	__host__ __device__ void add_point(SIt a, uint32_t b)
	{
		tr(a.e);
		if (g.bif[a.d][a.e] != BT_NONE || b == BT_NONE) return;
		uint32_t nd = bt_atomic_add(&g.ctr[CTR_NN], 1u);
		if (nd >= g.cap_n) { err |= BT_ERR_NODE_CAP; return; }
		tw(a.e); iw(b);
		g.nslot[nd] = a.e; g.ndead[nd] = 0; g.nidst[nd] = (b << 1) | a.d;
		g.nnext[nd] = g.head[a.d][b]; g.head[a.d][b] = nd;
		g.lsize[a.d][b]++;
		g.bif[a.d][a.e] = b; g.nodeof[a.d][a.e] = nd;
		bt_idx_mark(g, a.d, a.e, true);
		push_dirty(b);
	}
	// AddPoint with the node already allocated and the id already stamped by the caller (wave-wide collapse)
	__host__ __device__ bool add_point_prepared(SIt a, uint32_t b, uint32_t nd)
	{
		if (g.bif[a.d][a.e] != BT_NONE || b == BT_NONE) return false;
		g.nslot[nd] = a.e; g.ndead[nd] = 0; g.nidst[nd] = (b << 1) | a.d;
		g.nnext[nd] = g.head[a.d][b]; g.head[a.d][b] = nd;
		g.lsize[a.d][b]++;
		g.bif[a.d][a.e] = b; g.nodeof[a.d][a.e] = nd;
		bt_idx_mark(g, a.d, a.e, true);
		push_dirty(b);
		return true;
	}
	// ErasePoint, bifurcationstorage.cpp:144-155 (physical removal deferred to cleanup())
	__host__ __device__ void erase_point(SIt a)
	{
		tr(a.e);
		uint32_t b = g.bif[a.d][a.e];
		if (b == BT_NONE) return;
		tw(a.e); iw(b);
		uint32_t nd = g.nodeof[a.d][a.e];
		g.bif[a.d][a.e] = BT_NONE;
		bt_idx_mark(g, a.d, a.e, false);
		g.ndead[nd] = 1;                     // the node keeps its element: a proxy that is already in use stays dereferenceable
		g.nclr[nd] = tc_head; tc_head = nd;
		push_dirty(b);
	}
	// Cleanup, bifurcationstorage.cpp:33-41
	__host__ __device__ void cleanup()
	{
		for (uint32_t nd = tc_head; nd != BT_NONE; nd = g.nclr[nd]) g.lsize[g.nidst[nd] & 1][g.nidst[nd] >> 1]--;
		tc_head = BT_NONE;
	}
	__host__ __device__ __forceinline__ uint32_t count_bif(uint32_t b) { ir(b); return g.lsize[0][b] + g.lsize[1][b]; }   // :71-75
};

// ------------------------------------------------------------------------------------------- Boost 1.54 unordered_map order
// boost/unordered/detail/{buckets.hpp:603-654, unique.hpp:302-354,591-619, table.hpp:321-338,808-824} as vendored by the
// reference: identity hash + mix64, power-of-two bucket counts from 16, max load factor 1, one singly linked node list.
struct BoostMap {
	uint32_t *key; int32_t *nxt; int32_t *bprev;     // node arrays [cap]; buckets [bcap]
	uint32_t size, cap, bc, bcap;
	int32_t first;
	bool started;
};
__host__ __device__ __forceinline__ uint64_t bt_mix64(uint64_t key)
{
	key = (~key) + (key << 21);
	key = key ^ (key >> 24);
	key = (key + (key << 3)) + (key << 8);
	key = key ^ (key >> 14);
	key = (key + (key << 2)) + (key << 4);
	key = key ^ (key >> 28);
	key = key + (key << 31);
	return key;
}
__host__ __device__ __forceinline__ uint32_t bt_new_bucket_count(uint32_t m)
{
	if (m <= 4) return 4;
	--m; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16;
	return m + 1;
}
__host__ __device__ __forceinline__ int32_t *bm_link(BoostMap &m, int32_t l) { return l == -2 ? &m.first : &m.nxt[l]; }
__host__ __device__ inline void bm_create_buckets(BoostMap &m, uint32_t bc)
{
	for (uint32_t i = 0; i < bc; i++) m.bprev[i] = -1;
	m.bc = bc;
}
__host__ __device__ inline int32_t bm_find(BoostMap &m, uint32_t key)
{
	if (!m.started) return -1;
	uint32_t b = (uint32_t)(bt_mix64(key) & (m.bc - 1));
	if (m.bprev[b] == -1) return -1;
	for (int32_t p = *bm_link(m, m.bprev[b]); p != -1 && (uint32_t)(bt_mix64(m.key[p]) & (m.bc - 1)) == b; p = m.nxt[p])
		if (m.key[p] == key) return p;
	return -1;
}
// operator[] for an absent key; returns the node index or -1 if the fixed capacity is exhausted
__host__ __device__ inline int32_t bm_insert(BoostMap &m, uint32_t key)
{
	uint32_t need = m.size + 1;
	if (need > m.cap) return -1;
	if (!m.started) { bm_create_buckets(m, 16); m.started = true; }                 // max(16, min_buckets_for_size(1))
	else if (need > m.bc) {                                                          // reserve_for_insert
		uint32_t want = need > m.size + (m.size >> 1) ? need : m.size + (m.size >> 1);
		uint32_t nb = bt_new_bucket_count(want + 1);
		if (nb > m.bcap) return -1;
		if (nb != m.bc) {                                                            // rehash_impl / place_in_bucket
			bm_create_buckets(m, nb);
			int32_t prev = -2;
			while (*bm_link(m, prev) != -1) {
				int32_t n = *bm_link(m, prev);
				uint32_t b = (uint32_t)(bt_mix64(m.key[n]) & (nb - 1));
				if (m.bprev[b] == -1) { m.bprev[b] = prev; prev = n; }
				else {
					*bm_link(m, prev) = m.nxt[n];
					m.nxt[n] = *bm_link(m, m.bprev[b]);
					*bm_link(m, m.bprev[b]) = n;
				}
			}
		}
	}
	int32_t n = (int32_t)m.size++;
	m.key[n] = key;
	uint32_t b = (uint32_t)(bt_mix64(key) & (m.bc - 1));
	if (m.bprev[b] == -1) {                                                          // add_node
		if (m.first != -1) m.bprev[(uint32_t)(bt_mix64(m.key[m.first]) & (m.bc - 1))] = n;
		m.bprev[b] = -2;
		m.nxt[n] = m.first;
		m.first = n;
	} else {
		m.nxt[n] = *bm_link(m, m.bprev[b]);
		*bm_link(m, m.bprev[b]) = n;
	}
	return n;
}

// The per-transaction scratch (BulgeWork's arrays) lives in LDS when it fits and in the arena otherwise, so the pointers kept in
// BulgeWork are GENERIC and every access through them is a FLAT instruction -- several hundred cycles even when it resolves to LDS, and
// the decision loops of RemoveBulges are one lane chasing such accesses.  The loops therefore exist twice (template <bool L>): with L
// the caller has checked (bt_scratch_in_lds) that every array they touch IS in LDS and says so to the compiler, which then emits DS
// instructions; without it the generic form.  Host builds (tests/hostsim) see no difference.
#if defined(__HIP_DEVICE_COMPILE__)
#define BT_ASSUME_LDS(L, p) do { if (L) __builtin_assume(__builtin_amdgcn_is_shared((const void *)(p))); } while (0)
#define BT_IS_LDS(p) __builtin_amdgcn_is_shared((const void *)(p))
#else
#define BT_ASSUME_LDS(L, p) ((void)0)
#define BT_IS_LDS(p) false
#endif

// ------------------------------------------------------------------------------------------- small sorts
template <bool L = false>
__host__ __device__ inline void bt_sort_u64(uint64_t *a, uint32_t n)
{
	BT_ASSUME_LDS(L, a);
	for (uint32_t gap = n / 2; gap > 0; gap /= 2)                 // shell sort
		for (uint32_t i = gap; i < n; i++) {
			uint64_t v = a[i]; uint32_t j = i;
			for (; j >= gap && a[j - gap] > v; j -= gap) a[j] = a[j - gap];
			a[j] = v;
		}
}
__host__ __device__ inline void bt_sort_u32(uint32_t *a, uint32_t n)
{
	for (uint32_t gap = n / 2; gap > 0; gap /= 2)
		for (uint32_t i = gap; i < n; i++) {
			uint32_t v = a[i]; uint32_t j = i;
			for (; j >= gap && a[j - gap] > v; j -= gap) a[j] = a[j - gap];
			a[j] = v;
		}
}

// ------------------------------------------------------------------------------------------- the transaction
// AnyBulges result: group g's members are grp_mem[grp_off[g] .. grp_off[g+1]), in unordered_map iteration order.
struct AnyBulgesOut { uint32_t ngroups; uint32_t *grp_off; uint32_t *grp_mem; };

// AnyBulges under construction: the Boost-ordered map + per-entry member lists (a log of instances chained per entry)
// lazy: the iteration order of the reference's unordered_map only matters when a call has TWO OR MORE bulge groups (8.6 % of the
// bulge-bearing calls on 8 strains, 0.2 % on two) -- a single group is a single group in any order.  A lazy build only logs the
// operator[] insertions (entry index = position in the log, which is the node index bm_insert would hand out) and bt_ab_finish
// replays them through the Boost restatement when there is more than one group: the same sequence of insertions, hence the same
// bucket list, hence the same order.  (The map was 16 % of a transaction: ~25 insertions of ~40 dependent LDS operations each, on one lane.)
struct ABuild {
	BoostMap m;
	char *echar;
	uint32_t *mhead, *mtail, *mcnt, *log_inst, *log_next;
	uint32_t logcap, nlog;
	bool any;
	bool lazy;
};

struct BulgeWork {
	uint32_t n;                  // instances of the id
	uint32_t *start;             // (node << 1) | strand, list order: + list then - list (ListPositions, bifurcationstorage.h:59-72)
	uint32_t *sel;               // element of each instance (a live node never changes its element)
	char *endc;                  // endChar
	// Window cache: what instance i sees walking its own strand, steps 0 .. ws-1 (step 0 = the instance).
	// Filled by bt_scan_instance (one thread) or by the wave-cooperative scan of simplify.hip (64 lanes);
	// all decision logic below reads the cache, never the graph, so the serial part has no pointer chasing.
	uint32_t ws;                 // stride = D + k + 2
	uint32_t *wel, *wbf;         // element, own-strand mark per step
	uint8_t *wch;                // raw character per step
	uint32_t *wlen;              // number of leading steps before the first separator (<= ws)
	uint64_t *wmk; uint32_t *wmn; // compact list of the marked steps >= 1 of each window: (step << 32) | id, and their number
	uint32_t mks;                // stride of wmk per window: ws in the arena, BT_LDS_MARKS when the lists live in the fast scratch
	bool mk_overflow;            // a window had more marks than mks: bt_marks_to_arena + full rescan required
	uint32_t *wst; char *wck;     // mark at step 0 and (oriented) character at step k of each window
	uint32_t *wbk, *wnb;          // steps at which the walk leaves consecutive slots (BT_MAX_BREAKS per window) and their number
	uint32_t *wdel;              // elements this transaction has deleted inside each window (reach beyond the reserved range, simplify.hip)
	unsigned long long *dirty_big; // ids with more than 256 instances: bit per window "saw the region of the last collapse" (simplify.hip)
	bool lite;                   // verdict-only use: wel / wbf / wch are not materialised
	// Lazy windows.  The reference walks an instance's window at the moment it needs it (FillVisit, the J walk, Overlap,
	// MaxBifurcationMultiplicity: bulgeremoval.cpp:371-407).  The cache reproduces that either EAGERLY -- after a collapse every
	// cached window that sees the rewritten region is rescanned at once (cheap for the usual dozen instances, and the set of
	// those windows is what the reservation check of an ordered round needs) -- or LAZILY: a collapse only bumps `epoch`, and
	// the loops ask for a window whose last scan is older (wep) right before they read it.  Ids with thousands of instances on a
	// few hundred bases (dense regime) make nearly every cached window see every collapse: eager is O(instances) per collapse,
	// lazy two window scans.  bt_rb_run returns 2 with the windows it needs in req[].
	bool lazy;
	// ... or, in an ordered round (where the set of windows a collapse dirties is computed anyway): those windows are only MARKED stale and
	// rescanned when the loops next read them (bt_rb_run returns 2).  The usual dirty window is the target's own, which nobody reads
	// again once its endChar equals the source's -- a transaction that collapses seven instances onto one rescanned seven windows for nothing,
	// and such transactions are what a round waits for.  Up to 256 instances (bit per window); visit(I) is NOT refreshed, as in the reference.
	bool use_stale;
	unsigned long long stale[4];
	uint32_t epoch, *wep;
	uint32_t req[2], nreq;
	// The J loop of a group skips members that are no longer valid or share I's endChar (bulgeremoval.cpp:383-386): one look per
	// member and per I, i.e. quadratic in the group size -- tens of millions of looks on the dense vectors.  With jscan set
	// bt_rb_run hands that search to the caller (returns 3: move idJ to the next candidate of [idJ, group end), or to the end;
	// then set jready), which the kernels do with 64 lanes x 4 members per step.
	bool jscan, jready;
	// MaxBifurcationMultiplicity of the two branches of a bulge (bulgeremoval.cpp:405-407) is one CountBifurcations per bifurcation
	// INSIDE a branch: a handful of dependent look-ups where bifurcations are sparse, dozens where every other position is one (many
	// strains).  With mscan set bt_rb_run hands branches with more than BT_MSCAN_MIN marks inside to the caller (returns 4: evaluate
	// mq_* with all lanes -- one look-up per lane, the same stamps -- into mres[], set mready, call again).
	uint32_t nold;               // SBL_PHASES=1: collapses of this transaction that took the round-3 form
	bool mscan, mready;
	// FillVisit by the caller's 64 lanes (commit.hip: wave_fill_visit) instead of one thread's shell sort: with wfill set bt_rb_run returns 5
	// (fill_i = the instance) where it would call bt_fill_visit; the caller clears need_fill
	bool wfill; uint32_t fill_i;
	// the I loop of a large group (pscan): bt_rb_run returns 6 and the caller moves idI to the next member that is valid AND has a valid later
	// member with another endChar (pj = the first such J) -- see bt_rb_next_pair; pready = idI / pj are such a pair (or idI is the group's end)
	bool pscan, pready, pjknown; uint32_t pj;
	uint32_t ret0;               // parked transactions (GraphView::park_of): value of ret when this launch took the transaction up
	uint32_t mscan_min;          // ... with more than this many marks inside the two branches together
	uint32_t mq_i, mq_di, mq_j, mq_dj, mres[2];
	uint64_t *visit; uint32_t nvisit, visit_cap;      // FillVisit result sorted by (bif, distance)
	uint32_t *occ; uint32_t occ_cap;
	uint32_t *lb, *lf;           // lookBack / lookForward (index, id) pairs
	AnyBulgesOut ab;
	ABuild abb;
	// resumable loop state of RemoveBulges (a collapse interrupts the loops for a window rescan)
	uint32_t gi, idI, idJ, ret;
	bool inI, need_fill;
	// the collapse bt_rb_run has decided on (performed by the caller: bt_collapse on one thread, or 64 lanes in simplify.hip)
	uint32_t c_src, c_dS, c_tgt, c_dT;
	uint32_t *act;               // scratch for the wave-wide collapse: (strand, element, id) AddPoint actions
	uint32_t *act_fast;          // ... up to BT_ACT_FAST of them in the fast scratch
};

__host__ __device__ __forceinline__ SIt bt_deref(Txn &t, uint32_t packed) { SIt a; a.e = t.g.nslot[packed >> 1]; a.d = packed & 1; return a; }
__host__ __device__ __forceinline__ bool bt_pvalid(Txn &t, uint32_t packed) { return !t.g.ndead[packed >> 1]; }      // IteratorProxy::Valid :23-26
__host__ __device__ __forceinline__ char bt_wchar(const BulgeWork &w, uint32_t i, uint32_t step)
{ char c = (char)w.wch[(size_t)i * w.ws + step]; return (w.start[i] & 1) ? bt_comp(c) : c; }

// number of live instances of an id (skipping nodes erased by an earlier Cleanup)
__host__ __device__ inline uint32_t bt_count_instances(const GraphView &g, uint32_t id)
{
	uint32_t n = 0;
	for (int s = 0; s < 2; s++)
		for (uint32_t nd = g.head[s][id]; nd != BT_NONE; nd = g.nnext[nd]) n += !g.ndead[nd];
	return n;
}

// ListPositions (bulgeremoval.cpp:335) + all scratch of the transaction.  False when fewer than two instances.
__host__ __device__ inline bool bt_setup(Txn &t, BulgeWork &w, bool lite = false, bool fill_list = true)
{
	GraphView &g = t.g;
	uint32_t k = g.k, D = g.D;
	t.ir(t.id);
	uint32_t n = g.lsize[0][t.id] + g.lsize[1][t.id];      // lists are clean on entry (Cleanup ran): live nodes = list sizes
	w.n = n;
	if (n < 2) return false;
	w.ws = D + k + 2;
	w.start = (uint32_t *)t.alloc2(n * 4);          // small per-instance arrays: fast scratch (LDS) when there is one
	w.sel = (uint32_t *)t.alloc2(n * 4);
	w.endc = (char *)t.alloc2(n);
	w.wlen = (uint32_t *)t.alloc2(n * 4);
	w.wmn = (uint32_t *)t.alloc2(n * 4);
	w.wst = (uint32_t *)t.alloc2(n * 4);
	w.wck = (char *)t.alloc2(n);
	// mark lists: in the fast scratch (LDS) for the writer pass of typical ids, lane 0 walks them many times
	w.mk_overflow = false;
	w.use_stale = false; w.stale[0] = w.stale[1] = w.stale[2] = w.stale[3] = 0;
	w.lazy = false; w.epoch = 0; w.wep = nullptr; w.nreq = 0; w.jscan = false; w.jready = false; w.mscan = false; w.mscan_min = BT_MSCAN_MIN; w.mready = false; w.nold = 0; w.wfill = false; w.fill_i = 0; w.pscan = false; w.pready = false; w.pjknown = false; w.pj = 0; w.ret0 = 0;
	const uint32_t lazy_min = g.lazy_min ? g.lazy_min : BT_LAZY_MIN;
	w.wmk = lite || n > lazy_min ? nullptr : (uint64_t *)t.falloc(n * BT_LDS_MARKS * 8);      // (a lazy run never moves its mark lists: full-size lists from the start)
	w.mks = BT_LDS_MARKS;
	if (!w.wmk) { w.wmk = (uint64_t *)t.alloc(n * w.ws * 8); w.mks = w.ws; }
	w.lite = lite;
	w.wel = w.wbf = nullptr; w.wch = nullptr; w.wbk = w.wnb = w.wdel = nullptr; w.visit = nullptr; w.occ = nullptr; w.lb = w.lf = nullptr; w.act = nullptr; w.act_fast = nullptr; w.dirty_big = nullptr;
	w.visit_cap = D; w.occ_cap = D + k;
	if (!lite) {
		w.wel = (uint32_t *)t.alloc(n * w.ws * 4);
		w.wbf = (uint32_t *)t.alloc(n * w.ws * 4);
		w.wch = (uint8_t *)t.alloc(n * w.ws);
		w.wbk = (uint32_t *)t.alloc(n * BT_MAX_BREAKS * 4);
		w.wnb = (uint32_t *)t.alloc2(n * 4);          // (fast scratch: the interval tests below start with it)
		w.wdel = (uint32_t *)t.alloc2(n * 4);
		if (w.wdel) for (uint32_t i = 0; i < n; i++) w.wdel[i] = 0;
		w.visit = (uint64_t *)t.alloc2(w.visit_cap * 8);
		w.occ = (uint32_t *)t.alloc(w.occ_cap * 4);
		w.lb = (uint32_t *)t.alloc2(k * 8); w.lf = (uint32_t *)t.alloc2(k * 8);   // flank lists of a collapse: read back by other lanes, LDS when it fits
		w.act = (uint32_t *)t.alloc((2 * D + 4) * 12);
		w.act_fast = g.test_flags & 1u ? (uint32_t *)t.falloc(BT_ACT_FAST * 12) : nullptr;      // the usual few dozen AddPoint actions of a collapse stay in the fast scratch (nullptr: no room)
		if (n > 256) w.dirty_big = (unsigned long long *)t.alloc(((n + 63) / 64) * 8);
		if (n > lazy_min) { w.wep = (uint32_t *)t.alloc(n * 4); if (w.wep) for (uint32_t i = 0; i < n; i++) w.wep[i] = 0; }
	}
	if (t.err) return false;
	if (!fill_list) return true;                    // the caller lists the positions itself (64 lanes, simplify.hip: wave_list_positions)
	uint32_t m = 0;
	for (uint32_t s = 0; s < 2; s++)
		for (uint32_t nd = g.head[s][t.id]; nd != BT_NONE; nd = g.nnext[nd])
			if (!g.ndead[nd] && m < n) { w.start[m] = (nd << 1) | s; w.sel[m] = g.nslot[nd]; m++; }
	if (m != n) { t.err |= BT_ERR_SCRATCH; return false; }     // cannot happen on a consistent graph
	return true;
}

// one thread fills instance i's window from the live graph (the wave version in simplify.hip writes the same values)
__host__ __device__ inline void bt_scan_instance(Txn &t, BulgeWork &w, uint32_t i)
{
	size_t base = (size_t)i * w.ws;
	SIt a = bt_deref(t, w.start[i]);
	uint32_t s = 0, nm = 0, nb = 0, prev = 0;
	const uint32_t k = t.g.k;
	for (; s < w.ws; s++) {
		if (!w.lite && s && a.e != (a.d ? prev - 1 : prev + 1)) { if (nb < BT_MAX_BREAKS) w.wbk[i * BT_MAX_BREAKS + nb] = s; nb++; }
		prev = a.e;
		t.tr(a.e);
		uint8_t c = t.g.ch[a.e];
		uint32_t b = t.g.bif[a.d][a.e];
		if (!w.lite) { w.wel[base + s] = a.e; w.wch[base + s] = c; w.wbf[base + s] = b; }
		if (s == 0) w.wst[i] = b;
		if (s == k) w.wck[i] = a.d ? bt_comp((char)c) : (char)c;
		if (c == BT_SEP) break;
		if (s && b != BT_NONE) { if (nm < w.mks) w.wmk[(size_t)i * w.mks + nm] = ((uint64_t)s << 32) | b; else w.mk_overflow = true; nm++; }
		a.e = a.d ? t.g.pv[a.e] : t.g.nx[a.e];
	}
	w.wlen[i] = s; w.wmn[i] = nm;
	if (!w.lite) w.wnb[i] = nb;
}
// a window had more marks than the fast scratch holds: move the lists to the arena (the caller rescans every window)
__host__ __device__ inline void bt_marks_to_arena(Txn &t, BulgeWork &w)
{
	w.wmk = (uint64_t *)t.alloc(w.n * w.ws * 8);
	w.mks = w.ws;
	w.mk_overflow = false;
}
__host__ __device__ inline void bt_scan_all(Txn &t, BulgeWork &w)
{
	for (uint32_t i = 0; i < w.n; i++) bt_scan_instance(t, w, i);
	if (w.mk_overflow) { bt_marks_to_arena(t, w); if (!t.err) for (uint32_t i = 0; i < w.n; i++) bt_scan_instance(t, w, i); }
}

// endChar (bulgeremoval.cpp:340-347, ProperKMer(k + 1) dnasequence.h:154-165)
__host__ __device__ inline void bt_end_chars(Txn &t, BulgeWork &w)
{
	uint32_t k = t.g.k;
	for (uint32_t i = 0; i < w.n; i++) w.endc[i] = w.wlen[i] >= k + 1 ? w.wck[i] : ' ';
}

// FillVisit, bulgeremoval.cpp:122-146
template <bool L = false>
__host__ __device__ inline void bt_fill_visit(Txn &t, BulgeWork &w, uint32_t i)
{
	uint32_t D = t.g.D, n = 0;
	const uint64_t *mk = w.wmk + (size_t)i * w.mks;
	uint64_t *visit = w.visit;
	const uint32_t *wst = w.wst, *wlen = w.wlen, *wmn = w.wmn;
	BT_ASSUME_LDS(L, mk); BT_ASSUME_LDS(L, visit); BT_ASSUME_LDS(L, wst); BT_ASSUME_LDS(L, wlen); BT_ASSUME_LDS(L, wmn);
	uint32_t start = wst[i], lim = wlen[i] < D ? wlen[i] : D, nm = wmn[i];
	for (uint32_t j = 0; j < nm; j++) {
		uint32_t step = (uint32_t)(mk[j] >> 32), b = (uint32_t)mk[j];
		if (step >= lim || b == start) break;
		if (n >= w.visit_cap) { t.err |= BT_ERR_SCRATCH; break; }
		visit[n++] = ((uint64_t)b << 32) | step;
	}
	bt_sort_u64<L>(visit, n);
	w.nvisit = n;
}

// Overlap, bulgeremoval.cpp:97-120: do the element sets [I, I + dI + k) and [J, J + dJ + k) intersect?
// A window is a few runs of consecutive slots (breaks only where earlier collapses inserted or erased elements),
// so the sets are compared as slot intervals; the sorted-set form is kept for windows with many breaks.
__host__ __device__ inline bool bt_overlap_sets(Txn &t, BulgeWork &w, uint32_t i, uint32_t di, uint32_t j, uint32_t dj)
{
	uint32_t k = t.g.k, n = di + k;
	if (n > w.occ_cap) { t.err |= BT_ERR_SCRATCH; return true; }
	const uint32_t *ei = w.wel + (size_t)i * w.ws, *ej = w.wel + (size_t)j * w.ws;
	for (uint32_t x = 0; x < n; x++) w.occ[x] = ei[x];
	bt_sort_u32(w.occ, n);
	for (uint32_t x = 0; x < dj + k; x++) {
		uint32_t e = ej[x], lo = 0, hi = n;
		while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (w.occ[mid] < e) lo = mid + 1; else hi = mid; }
		if (lo < n && w.occ[lo] == e) return true;
	}
	return false;
}
// do the first ni steps of window i and the first nj steps of window j share an element?  (-1: too many breaks to tell)
// two walks that never leave consecutive slots are two intervals that start at the instances' own elements: no look at the cache at all
template <bool L = false>
__host__ __device__ __forceinline__ bool bt_plain_intervals_meet(const BulgeWork &w, uint32_t i, uint32_t ni, uint32_t j, uint32_t nj)
{
	const uint32_t *sel = w.sel, *start = w.start;
	BT_ASSUME_LDS(L, sel); BT_ASSUME_LDS(L, start);
	const uint32_t a0 = sel[i], b0 = sel[j];
	const uint32_t alo = (start[i] & 1u) ? a0 - (ni - 1) : a0, ahi = (start[i] & 1u) ? a0 : a0 + (ni - 1);
	const uint32_t blo = (start[j] & 1u) ? b0 - (nj - 1) : b0, bhi = (start[j] & 1u) ? b0 : b0 + (nj - 1);
	return alo <= bhi && blo <= ahi;
}
__host__ __device__ inline int bt_windows_intersect(const BulgeWork &w, uint32_t i, uint32_t ni, uint32_t j, uint32_t nj)
{
	if (w.wnb[i] > BT_MAX_BREAKS || w.wnb[j] > BT_MAX_BREAKS) return -1;
	if (!ni || !nj) return 0;
	if (w.wnb[i] == 0 && w.wnb[j] == 0) return bt_plain_intervals_meet(w, i, ni, j, nj) ? 1 : 0;
	const uint32_t *ei = w.wel + (size_t)i * w.ws, *ej = w.wel + (size_t)j * w.ws;
	const uint32_t *bi = w.wbk + i * BT_MAX_BREAKS, *bj = w.wbk + j * BT_MAX_BREAKS;
	uint32_t xi = 0;
	for (uint32_t s0 = 0; s0 < ni;) {
		while (xi < w.wnb[i] && bi[xi] <= s0) xi++;
		uint32_t s1 = xi < w.wnb[i] && bi[xi] < ni ? bi[xi] : ni;
		uint32_t a0 = ei[s0], a1 = ei[s1 - 1], alo = a0 < a1 ? a0 : a1, ahi = a0 < a1 ? a1 : a0;
		uint32_t xj = 0;
		for (uint32_t r0 = 0; r0 < nj;) {
			while (xj < w.wnb[j] && bj[xj] <= r0) xj++;
			uint32_t r1 = xj < w.wnb[j] && bj[xj] < nj ? bj[xj] : nj;
			uint32_t b0 = ej[r0], b1 = ej[r1 - 1], blo = b0 < b1 ? b0 : b1, bhi = b0 < b1 ? b1 : b0;
			if (alo <= bhi && blo <= ahi) return 1;
			r0 = r1;
		}
		s0 = s1;
	}
	return 0;
}
template <bool L = false>
__host__ __device__ inline bool bt_overlap(Txn &t, BulgeWork &w, uint32_t i, uint32_t di, uint32_t j, uint32_t dj)
{
	const uint32_t k = t.g.k, ni = di + k, nj = dj + k;
	{
		const uint32_t *wnb = w.wnb;
		BT_ASSUME_LDS(L, wnb);
		if (wnb[i] == 0 && wnb[j] == 0) return bt_plain_intervals_meet<L>(w, i, ni, j, nj);
	}
	if (w.wnb[i] > BT_MAX_BREAKS || w.wnb[j] > BT_MAX_BREAKS) return bt_overlap_sets(t, w, i, di, j, dj);
	const uint32_t *ei = w.wel + (size_t)i * w.ws, *ej = w.wel + (size_t)j * w.ws;
	const uint32_t *bi = w.wbk + i * BT_MAX_BREAKS, *bj = w.wbk + j * BT_MAX_BREAKS;
	// runs of window i: [s0, s1) with s1 = next break (or ni)
	uint32_t xi = 0;
	for (uint32_t s0 = 0; s0 < ni;) {
		while (xi < w.wnb[i] && bi[xi] <= s0) xi++;
		uint32_t s1 = xi < w.wnb[i] && bi[xi] < ni ? bi[xi] : ni;
		uint32_t a0 = ei[s0], a1 = ei[s1 - 1], alo = a0 < a1 ? a0 : a1, ahi = a0 < a1 ? a1 : a0;
		uint32_t xj = 0;
		for (uint32_t r0 = 0; r0 < nj;) {
			while (xj < w.wnb[j] && bj[xj] <= r0) xj++;
			uint32_t r1 = xj < w.wnb[j] && bj[xj] < nj ? bj[xj] : nj;
			uint32_t b0 = ej[r0], b1 = ej[r1 - 1], blo = b0 < b1 ? b0 : b1, bhi = b0 < b1 ? b1 : b0;
			if (alo <= bhi && blo <= ahi) return true;
			r0 = r1;
		}
		s0 = s1;
	}
	return false;
}

// MaxBifurcationMultiplicity, bulgeremoval.cpp:39-53
template <bool L = false>
__host__ __device__ inline uint32_t bt_max_mult(Txn &t, BulgeWork &w, uint32_t i, uint32_t distance)
{
	const uint32_t *wmn = w.wmn;
	const uint64_t *mk = w.wmk + (size_t)i * w.mks;
	BT_ASSUME_LDS(L, wmn); BT_ASSUME_LDS(L, mk);
	uint32_t r = 0, nm = wmn[i];
	for (uint32_t j = 0; j < nm; j++) {
		if ((uint32_t)(mk[j] >> 32) >= distance) break;
		uint32_t c = t.count_bif((uint32_t)mk[j]);
		if (c > r) r = c;
	}
	return r;
}

// MaxBifurcationMultiplicity of BOTH branches of a bulge, four bifurcations per memory round trip.  count_bif is a stamp check (owner,
// write stamp of the id) followed by the two list sizes -- two dependent round trips per bifurcation inside a branch when one thread
// walks them one after the other, ~14 for the usual three or four bifurcations per branch: the largest single item of the decision
// loops (SBL_PHASES: "r:multiplicities", 40 % of rb_run).  Here the loads of four bifurcations are issued together (the stamp words and
// the sizes do not depend on each other: a violation is reported either way) and only then looked at.  Same values, same stamps.
template <bool L = false>
__host__ __device__ __forceinline__ void bt_max_mult_pair(Txn &t, BulgeWork &w, uint32_t i, uint32_t di, uint32_t j, uint32_t dj, uint32_t &ri, uint32_t &rj)
{
	if (!t.mode || !t.ext_stamps) { ri = bt_max_mult<L>(t, w, i, di); rj = bt_max_mult<L>(t, w, j, dj); return; }
	const uint32_t *wmn = w.wmn;
	const uint64_t *mki = w.wmk + (size_t)i * w.mks, *mkj = w.wmk + (size_t)j * w.mks;
	BT_ASSUME_LDS(L, wmn); BT_ASSUME_LDS(L, mki); BT_ASSUME_LDS(L, mkj);
	uint32_t ni = 0, nj = 0;
	{ const uint32_t nm = wmn[i]; while (ni < nm && (uint32_t)(mki[ni] >> 32) < di) ni++; }
	{ const uint32_t nm = wmn[j]; while (nj < nm && (uint32_t)(mkj[nj] >> 32) < dj) nj++; }
	const GraphView &g = t.g;
	uint32_t a = 0, b = 0;
	const uint32_t total = ni + nj;
	for (uint32_t p0 = 0; p0 < total; p0 += 4) {
		uint32_t id[4], ow[4], wm[4], l0[4], l1[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint32_t p = p0 + u < total ? p0 + u : p0;            // (a short last batch loads its first bifurcation again)
			id[u] = p < ni ? (uint32_t)mki[p] : (uint32_t)mkj[p - ni];
		}
#pragma unroll
		for (int u = 0; u < 4; u++) { ow[u] = g.own[id[u]]; wm[u] = g.wmax[g.nblk + id[u]]; l0[u] = g.lsize[0][id[u]]; l1[u] = g.lsize[1][id[u]]; }
#pragma unroll
		for (int u = 0; u < 4; u++) {
			const uint32_t p = p0 + u;
			if (p >= total) break;
			if (t.mode != 3 && !t.chain && ow[u] != t.stamp) t.violation(BT_NONE);      // stamp_id_light(id, false)
			if (wm[u] > t.tid) t.violation(BT_NONE);
			if (t.mode == 2) bt_atomic_max(&g.rmax[g.nblk + id[u]], t.tid);
			const uint32_t c = l0[u] + l1[u];
			if (p < ni) { if (c > a) a = c; } else if (c > b) b = c;
		}
	}
	ri = a; rj = b;
}

// number of marked steps strictly inside a branch of `distance` steps (what bt_max_mult would look up)
template <bool L = false>
__host__ __device__ inline uint32_t bt_marks_inside(const BulgeWork &w, uint32_t i, uint32_t distance)
{
	const uint32_t *wmn = w.wmn;
	const uint64_t *mk = w.wmk + (size_t)i * w.mks;
	BT_ASSUME_LDS(L, wmn); BT_ASSUME_LDS(L, mk);
	uint32_t c = 0, nm = wmn[i];
	while (c < nm && (uint32_t)(mk[c] >> 32) < distance) c++;
	return c;
}
// the evaluation bt_rb_run asks for with return code 4, one thread
__host__ __device__ inline void bt_rb_mults(Txn &t, BulgeWork &w)
{
	w.mres[0] = bt_max_mult(t, w, w.mq_i, w.mq_di);
	w.mres[1] = bt_max_mult(t, w, w.mq_j, w.mq_dj);
	w.mready = true;
}

// DNASequence::ReplaceDirect, dnasequence.cpp:189-230 (target = first element of the old span in + direction)
__host__ __device__ inline void bt_replace_direct(Txn &t, SIt source, uint32_t dS, uint32_t target, uint32_t dT)
{
	GraphView &g = t.g;
	uint32_t save = target, e = target;
	uint32_t common = dS < dT ? dS : dT;
	t.tr(save);
	uint32_t firstPos = g.op[save] & BT_POS_MASK;
	for (uint32_t i = 0; i < dT; i++) { t.tr(e); e = g.nx[e]; }
	t.tr(e);
	uint32_t lastPos = g.op[e] & BT_POS_MASK;
	for (uint32_t i = 0; i < common; i++) { char c = t.chr(source); t.tw(target); g.ch[target] = (uint8_t)c; target = g.nx[target]; source = t.next(source); }
	if (dS < dT) {                                   // erase the surplus
		t.tr(target);
		uint32_t before = g.pv[target], cur = target;
		for (uint32_t i = 0; i < dT - dS; i++) { t.tw(cur); g.ch[cur] = BT_DEAD_CHAR; bt_idx_dirty(g, cur); cur = g.nx[cur]; }
		t.tw(before); t.tw(cur);
		bt_idx_dirty(g, before); bt_idx_dirty(g, cur);
		g.nx[before] = cur; g.pv[cur] = before;
	} else if (dS > dT) {                            // insert the deficit before `target`
		uint32_t m = dS - dT;
		uint32_t span = bt_insert_span(m);
		uint32_t base = bt_atomic_add(&g.ctr[CTR_NE], span);
		if (base + span > g.cap_e) { t.err |= BT_ERR_ELEM_CAP; return; }
		t.tr(target);
		uint32_t before = g.pv[target];
		t.tw(before); t.tw(target);
		bt_idx_dirty(g, before); bt_idx_dirty(g, target);
		for (uint32_t i = 0; i < span; i++) {        // the characters are read from the source one by one (no element is both source and target)
			uint32_t ne = base + i;
			if (i < m) {
				g.ch[ne] = (uint8_t)t.chr(source); source = t.next(source);
				g.op[ne] = 0;
				g.bif[0][ne] = g.bif[1][ne] = BT_NONE;
				t.tw(ne);
				g.nx[before] = ne; g.pv[ne] = before;
				before = ne;
			} else { g.ch[ne] = BT_DEAD_CHAR; g.bif[0][ne] = g.bif[1][ne] = BT_NONE; }
		}
		g.nx[before] = target; g.pv[target] = before;
	}
	double acc = (double)firstPos;                   // :221-227, same operation order: acc += ssize
	double ssize = (double)dT / (double)dS;
	for (uint32_t i = 0; i < dS; i++, acc += ssize) {
		uint64_t p = (uint64_t)acc;
		if (p > lastPos) p = lastPos;
		t.tw(save);
		g.op[save] = (uint32_t)p & BT_POS_MASK;
		save = g.nx[save];
	}
}

// every id whose forward window can see the rewritten region must be re-examined at its turn
__host__ __device__ inline void bt_push_neighbourhood(Txn &t, SIt tstart, uint32_t newlen)
{
	// Only instances walking TOWARDS the rewritten region can see it: upstream of the target that is the target's own
	// strand, beyond the end of the region the opposite strand; inside the region both.
	uint32_t reach = t.g.D + t.g.k + 2, k = t.g.k, d = tstart.d;   // windows are scanned (and stamped) over D + k + 2 steps
	uint32_t e = d ? t.g.nx[tstart.e] : t.g.pv[tstart.e];
	for (uint32_t i = 0; i < reach && e != BT_NONE; i++) {
		if (t.g.ch[e] == BT_SEP) break;
		uint32_t b = t.g.bif[d][e];
		if (b != BT_NONE) t.push_dirty(b);
		e = d ? t.g.nx[e] : t.g.pv[e];
	}
	e = tstart.e;
	bool open = true;
	for (uint32_t i = 0; i <= newlen + 2 * k; i++) {
		if (t.g.ch[e] == BT_SEP) { open = false; break; }
		uint32_t b0 = t.g.bif[0][e], b1 = t.g.bif[1][e];
		if (b0 != BT_NONE) t.push_dirty(b0);
		if (b1 != BT_NONE) t.push_dirty(b1);
		e = d ? t.g.pv[e] : t.g.nx[e];
		if (e == BT_NONE) { open = false; break; }
	}
	for (uint32_t i = 0; open && i < reach; i++) {
		if (t.g.ch[e] == BT_SEP) break;
		uint32_t b = t.g.bif[d ^ 1][e];
		if (b != BT_NONE) t.push_dirty(b);
		e = d ? t.g.pv[e] : t.g.nx[e];
		if (e == BT_NONE) break;
	}
}

// CollapseBulgeGreedily = EraseBifurcations + DNASequence::Replace + UpdateBifurcations
// (bulgeremoval.cpp:284-327, :55-95, dnasequence.cpp:232-252, bulgeremoval.cpp:238-282)
__host__ __device__ inline void bt_collapse(Txn &t, BulgeWork &w, uint32_t srcK, uint32_t dS, uint32_t tgtK, uint32_t dT)
{
	uint32_t k = t.g.k;
	SIt tt = bt_deref(t, w.start[tgtK]), ss = bt_deref(t, w.start[srcK]);
	uint32_t nlb = 0, nlf = 0, anear = 0, bnear = 0;
	SIt amer = t.inv(t.adv(tt, k));
	SIt bmer = t.adv(tt, dT);
	for (uint32_t i = 0; i < k; i++, amer = t.next(amer), bmer = t.next(bmer)) {
		uint32_t b = t.getbif(amer);
		if (b != BT_NONE) { t.erase_point(amer); w.lb[2 * nlb] = i; w.lb[2 * nlb + 1] = b; nlb++; }
		b = t.getbif(bmer);
		if (b != BT_NONE) { t.erase_point(bmer); w.lf[2 * nlf] = i; w.lf[2 * nlf + 1] = b; nlf++; }
	}
	amer = tt;
	bmer = t.inv(t.adv(tt, k + dT));
	for (uint32_t i = 0; i < k + dT; i++, amer = t.next(amer), bmer = t.next(bmer)) {
		if (i > 0) t.erase_point(amer);
		t.erase_point(bmer);
	}
	{
		SIt source = t.adv(ss, k), target = t.adv(tt, k);
		if (target.d == 0) bt_replace_direct(t, source, dS, target.e, dT);
		else {
			source = t.inv(t.adv(source, dS));
			bt_replace_direct(t, source, dS, t.inv(t.adv(target, dT)).e, dT);
		}
	}
	if (t.err) return;
	amer = t.inv(t.adv(tt, k));
	bmer = t.adv(tt, dS);
	for (uint32_t i = 0; i < k; i++, amer = t.next(amer), bmer = t.next(bmer)) {
		if (anear < nlb && i == w.lb[2 * anear]) { t.add_point(amer, w.lb[2 * anear + 1]); anear++; }
		if (bnear < nlf && i == w.lf[2 * bnear]) { t.add_point(bmer, w.lf[2 * bnear + 1]); bnear++; }
	}
	amer = tt;
	bmer = t.inv(t.adv(tt, dS + k));
	SIt sa = ss, sb = t.inv(t.adv(ss, dS + k));
	for (uint32_t i = 0; i < dS + 1; i++, amer = t.next(amer), bmer = t.next(bmer), sa = t.next(sa), sb = t.next(sb)) {
		uint32_t b = t.getbif(sa);
		if (b != BT_NONE) t.add_point(amer, b);
		b = t.getbif(sb);
		if (b != BT_NONE) t.add_point(bmer, b);
	}
	if (t.defer_push) { t.push_e = tt.e; t.push_d = tt.d; t.push_len = dS; }
	else bt_push_neighbourhood(t, tt, dS);
}

// AnyBulges (bulgeremoval.cpp:158-218) into a BoostMap + per-entry member lists, reading the window cache.
// The map-building loop exists twice: bt_any_bulges below (one thread) and wave_any_bulges in simplify.hip (64 lanes skip
// the look-ups that change nothing); both go through bt_ab_prepare / bt_ab_insert / bt_ab_append / bt_ab_finish, so the
// sequence of operator[] insertions -- and with it the iteration order -- is the same.
// cap: upper bound of the number of distinct ids that get an entry (the total number of marks is always one).
// lazy: the insertions are only logged (ABuild::lazy); the node links and buckets of the Boost restatement are then allocated by
// bt_ab_finish, and only for a call that ends with two or more groups.  Without them an entry takes 29 bytes instead of 45, and the map
// of a typical id (50 - 60 reached ids) fits what a transaction has left of its LDS scratch: in the arena every append and every step of
// bt_ab_finish's loops was a memory round trip of its own on lane 0 -- the larger part of the "rb_begin" phase of k_commit.
__host__ __device__ inline bool bt_ab_prepare(Txn &t, BulgeWork &w, uint32_t cap, bool lazy = false)
{
	ABuild &a = w.abb;
	uint32_t n = w.n;
	if (cap < 16) cap = 16;
	// the map goes to the fast scratch (LDS) when it fits there, otherwise it takes whatever arena is left
	// (lazy: key 4 + echar 1 + head / tail / cnt 12 per entry, the log 8 per entry and instance, 8 bytes of alignment for each of the 7 arrays)
	const uint32_t ffree = t.fscr ? t.fscr_cap - ((t.fscr_used + 7u) & ~7u) : 0u;
	bool fast = lazy ? ffree >= 17u * cap + 8u * (cap + n) + 64u : ffree / 48 >= cap + n / 2 + 2;
	if (!fast) {
		uint32_t left = t.scr_cap - ((t.scr_used + 7u) & ~7u);
		uint32_t fit = left / 48;                    // key 4 + nxt 4 + echar 1 + head/tail/cnt 12 + buckets 2x4 + log 2x8 < 48
		if (fit < 16) { t.err |= BT_ERR_SCRATCH; return false; }
		if (cap > fit) cap = fit;
	}
	uint32_t bcap = bt_new_bucket_count(cap + 1);
	if (bcap > 2 * cap) bcap >>= 1;
	if (bcap < 16) bcap = 16;
	BoostMap &m = a.m;
	auto A = [&](uint32_t bytes) { return fast ? t.alloc2(bytes) : t.alloc(bytes); };
	m.key = (uint32_t *)A(cap * 4);
	m.nxt = nullptr; m.bprev = nullptr;
	if (!lazy) { m.nxt = (int32_t *)A(cap * 4); m.bprev = (int32_t *)A(bcap * 4); }
	a.echar = (char *)A(cap);
	a.mhead = (uint32_t *)A(cap * 4); a.mtail = (uint32_t *)A(cap * 4); a.mcnt = (uint32_t *)A(cap * 4);
	a.logcap = cap + n;
	a.log_inst = (uint32_t *)A(a.logcap * 4); a.log_next = (uint32_t *)A(a.logcap * 4);
	if (t.err) return false;
	m.size = 0; m.cap = cap; m.bc = 0; m.bcap = bcap; m.first = -1; m.started = false;
	a.nlog = 0; a.any = false; a.lazy = lazy;
	return true;
}
// id b is reached by instance i and has no entry yet: operator[] creates it.  Returns the entry or -1 (scratch exhausted).
__host__ __device__ inline int32_t bt_ab_insert(Txn &t, BulgeWork &w, uint32_t i, uint32_t b)
{
	ABuild &a = w.abb;
	int32_t kt;
	if (a.lazy) { if (a.m.size >= a.m.cap) kt = -1; else { kt = (int32_t)a.m.size++; a.m.key[kt] = b; } }
	else kt = bm_insert(a.m, b);
	if (kt < 0 || a.nlog >= a.logcap) { t.err |= BT_ERR_SCRATCH; return -1; }
	a.echar[kt] = w.endc[i];
	a.log_inst[a.nlog] = i; a.log_next[a.nlog] = BT_NONE;
	a.mhead[kt] = a.mtail[kt] = a.nlog++; a.mcnt[kt] = 1;
	return kt;
}
// instance i reaches entry kt, whose endChar differs: second (or later) member of the bulge group
__host__ __device__ inline bool bt_ab_append(Txn &t, BulgeWork &w, uint32_t i, int32_t kt)
{
	ABuild &a = w.abb;
	if (a.nlog >= a.logcap) { t.err |= BT_ERR_SCRATCH; return false; }
	a.log_inst[a.nlog] = i; a.log_next[a.nlog] = BT_NONE;
	a.log_next[a.mtail[kt]] = a.nlog; a.mtail[kt] = a.nlog++; a.mcnt[kt]++;
	a.any = true;
	return true;
}
// groups with more than one member, in unordered_map iteration order
__host__ __device__ inline bool bt_ab_finish(Txn &t, BulgeWork &w)
{
	ABuild &a = w.abb;
	BoostMap &m = a.m;
	if (!a.any) return false;
	uint32_t ng = 0, total = 0;
	if (a.lazy) {
		int32_t only = -1;
		for (uint32_t p = 0; p < m.size; p++) if (a.mcnt[p] > 1) { ng++; total += a.mcnt[p]; only = (int32_t)p; }
		a.lazy = false;
		if (ng < 2) {                                     // one group (the usual case): no iteration order to reproduce
			w.ab.grp_off = (uint32_t *)t.alloc2(2 * 4);
			w.ab.grp_mem = (uint32_t *)t.alloc2(total * 4);
			if (t.err || only < 0) return false;
			uint32_t o = 0;
			w.ab.grp_off[0] = 0;
			for (uint32_t l = a.mhead[only]; l != BT_NONE; l = a.log_next[l]) w.ab.grp_mem[o++] = a.log_inst[l];
			w.ab.grp_off[1] = o;
			w.ab.ngroups = 1;
			return true;
		}
		// the order matters: the logged insertions through the Boost restatement, in order (its links and buckets are allocated now)
		m.nxt = (int32_t *)t.alloc2(m.cap * 4); m.bprev = (int32_t *)t.alloc2(m.bcap * 4);
		if (t.err) return false;
		const uint32_t n = m.size;
		m.size = 0;
		for (uint32_t i = 0; i < n; i++) if (bm_insert(m, m.key[i]) != (int32_t)i) { t.err |= BT_ERR_SCRATCH; return false; }
		ng = 0; total = 0;
	}
	for (int32_t p = m.first; p != -1; p = m.nxt[p]) if (a.mcnt[p] > 1) { ng++; total += a.mcnt[p]; }
	w.ab.grp_off = (uint32_t *)t.alloc2((ng + 1) * 4);
	w.ab.grp_mem = (uint32_t *)t.alloc2(total * 4);
	if (t.err) return false;
	uint32_t gi = 0, o = 0;
	for (int32_t p = m.first; p != -1; p = m.nxt[p]) {
		if (a.mcnt[p] <= 1) continue;
		w.ab.grp_off[gi++] = o;
		for (uint32_t l = a.mhead[p]; l != BT_NONE; l = a.log_next[l]) w.ab.grp_mem[o++] = a.log_inst[l];
	}
	w.ab.grp_off[gi] = o;
	w.ab.ngroups = ng;
	return true;
}

// verdict_only: stop at the first group that gets a second member.
__host__ __device__ inline bool bt_any_bulges(Txn &t, BulgeWork &w, bool verdict_only)
{
	uint32_t D = t.g.D, n = w.n;
	uint32_t marks = 0;
	for (uint32_t i = 0; i < n; i++) marks += w.wmn[i];
	if (!bt_ab_prepare(t, w, marks, t.g.test_lazy_map != 0 && !verdict_only)) return false;      // (tests/hostsim: the lazy build of the kernels' wave_any_bulges, with a linear search as its shadow table)
	ABuild &a = w.abb;
	for (uint32_t i = 0; i < n; i++) {
		if (w.endc[i] == ' ') continue;
		const uint64_t *mk = w.wmk + (size_t)i * w.mks;
		uint32_t start = w.wst[i], lim = w.wlen[i] < D ? w.wlen[i] : D, nm = w.wmn[i];
		for (uint32_t j = 0; j < nm; j++) {
			uint32_t b = (uint32_t)mk[j];
			if ((uint32_t)(mk[j] >> 32) >= lim || b == start) break;
			int32_t kt = -1;
			if (a.lazy) { for (uint32_t x = 0; x < a.m.size; x++) if (a.m.key[x] == b) { kt = (int32_t)x; break; } }
			else kt = bm_find(a.m, b);
			if (kt < 0) { if (bt_ab_insert(t, w, i, b) < 0) return false; }
			else if (a.echar[kt] != w.endc[i]) {
				if (!bt_ab_append(t, w, i, kt)) return false;
				if (verdict_only) return true;
				break;
			}
		}
	}
	if (!a.any || verdict_only) return a.any;
	return bt_ab_finish(t, w);
}

// RemoveBulges, bulgeremoval.cpp:330-430, as a resumable routine over the window cache.
//   bt_rb_begin: endChar + AnyBulges on freshly scanned windows; false = nothing to do.
//   bt_rb_run:   runs the group / I / J loops until a collapse has been decided (returns true: the caller performs
//                CollapseBulgeGreedily(c_src -> c_tgt), rescans the windows and calls again) or everything is done
//                (returns false, Cleanup performed).
__host__ __device__ inline bool bt_rb_begin(Txn &t, BulgeWork &w, int any_bulges = -1 /* >= 0: AnyBulges already evaluated by the caller */)
{
	if (any_bulges < 0) { bt_end_chars(t, w); any_bulges = bt_any_bulges(t, w, false) ? 1 : 0; }
	if (!any_bulges) return false;
	t.iw(t.id);
	w.gi = 0; w.idI = w.ab.grp_off[0]; w.idJ = 0; w.ret = 0; w.inI = false; w.need_fill = false; w.pready = false;
	return true;
}

// the search bt_rb_run asks for with return code 3, one thread
__host__ __device__ inline void bt_rb_next_j(Txn &t, BulgeWork &w)
{
	const uint32_t ge = w.ab.grp_off[w.gi + 1], kmerI = w.ab.grp_mem[w.idI];
	while (w.idJ < ge) {
		const uint32_t kmerJ = w.ab.grp_mem[w.idJ];
		if (w.endc[kmerI] != w.endc[kmerJ] && bt_pvalid(t, w.start[kmerJ])) break;
		w.idJ++;
	}
	w.jready = true;
}

// the search bt_rb_run asks for with return code 6, one thread: the next I at or after idI (current group) that is valid and has a valid
// later member with a different endChar.  Every I in between would enter its J loop and leave it without having found anything (no
// side effects: FillVisit is only evaluated when a J needs it) -- in a group of 62 members with one deviating instance that is 60 trips
// through the caller-side J search.
__host__ __device__ inline void bt_rb_next_pair(Txn &t, BulgeWork &w)
{
	const uint32_t ge = w.ab.grp_off[w.gi + 1];
	uint32_t i = w.idI;
	w.pj = 0;
	for (; i < ge; i++) {
		const uint32_t mi = w.ab.grp_mem[i];
		if (!bt_pvalid(t, w.start[mi])) continue;
		uint32_t j = i + 1;
		for (; j < ge; j++) { const uint32_t mj = w.ab.grp_mem[j]; if (w.endc[mj] != w.endc[mi] && bt_pvalid(t, w.start[mj])) break; }
		if (j < ge) { w.pj = j; break; }
	}
	w.idI = i; w.pready = true; w.pjknown = i < ge;
}

// 4 (mscan only): see BulgeWork::mscan (bt_rb_mults is the one-thread form).
// returns 0: all loops done (Cleanup performed unless deferred), 1: a collapse has been decided (c_src -> c_tgt), 2 (lazy runs only):
// the windows req[0 .. nreq) must be rescanned (and their wep set to epoch) before the loops can go on -- call again afterwards,
// 3 (jscan only): see BulgeWork::jscan (bt_rb_next_j is the one-thread form of that search), 5 (wfill only): see BulgeWork::wfill.
// 6 (pscan only): see BulgeWork::pready (bt_rb_next_pair is the one-thread form).
// every array the decision loops (bt_rb_run) touch is in LDS: the caller may then use bt_rb_run<true>
__host__ __device__ __forceinline__ bool bt_scratch_in_lds(const BulgeWork &w)
{
	return BT_IS_LDS(w.ab.grp_off) && BT_IS_LDS(w.ab.grp_mem) && BT_IS_LDS(w.start) && BT_IS_LDS(w.sel) && BT_IS_LDS(w.endc) && BT_IS_LDS(w.wmk) && BT_IS_LDS(w.wlen)
	       && BT_IS_LDS(w.wmn) && BT_IS_LDS(w.wst) && BT_IS_LDS(w.visit) && BT_IS_LDS(w.wnb);
}
template <bool L = false>
__host__ __device__ __forceinline__ int bt_rb_run(Txn &t, BulgeWork &w)      // (always inline: as a call it takes t / w as generic pointers and spills around itself -- +5 ms per stage when the inliner gave up on it)
{
	const uint32_t D = t.g.D;
	BT_PROF_T0(t);
	const uint32_t *grp_off = w.ab.grp_off, *grp_mem = w.ab.grp_mem, *start = w.start, *wlen = w.wlen, *wmn = w.wmn;
	char *endc = w.endc;
	const uint64_t *wmk = w.wmk, *visit = w.visit;
	BT_ASSUME_LDS(L, grp_off); BT_ASSUME_LDS(L, grp_mem); BT_ASSUME_LDS(L, start); BT_ASSUME_LDS(L, wlen); BT_ASSUME_LDS(L, wmn); BT_ASSUME_LDS(L, endc);
	BT_ASSUME_LDS(L, wmk); BT_ASSUME_LDS(L, visit);
	while (w.gi < w.ab.ngroups) {
		const uint32_t ge = grp_off[w.gi + 1];
		while (w.idI < ge) {
			if (!w.inI && w.pscan && !w.pready && ge - w.idI > 8) return 6;      // the caller finds the next I that has a J at all (see pready); short tails are walked here
			const uint32_t kmerI = grp_mem[w.idI];
			if (!w.inI) {
				if (w.pready) { w.inI = true; w.idJ = w.pjknown ? w.pj : w.idI + 1; w.need_fill = true; w.jready = w.pjknown; w.pready = false; }      // (valid, and pj is its first candidate J -- when known)
				else {
					if (!bt_pvalid(t, start[kmerI])) { w.idI++; continue; }
					w.inI = true; w.idJ = w.idI + 1; w.need_fill = true; w.jready = false;
				}
			}
			while (w.idJ < ge) {
				if (w.jscan && !w.jready && ge - w.idJ > 8) return 3;  // the caller finds the next candidate J (see jscan); short tails are walked here
				const uint32_t kmerJ = grp_mem[w.idJ];
				if (endc[kmerI] == endc[kmerJ] || !bt_pvalid(t, start[kmerJ])) { w.idJ++; w.jready = false; continue; }      // (the endChars first: an LDS byte against a look at the node in memory; neither has a side effect)
				if (w.lazy) {                                        // everything below reads the windows of I and J: as of NOW, like the reference's walks
					uint32_t nr = 0;
					if (w.wep[kmerI] != w.epoch) w.req[nr++] = kmerI;
					if (w.wep[kmerJ] != w.epoch) w.req[nr++] = kmerJ;
					if (nr) { w.nreq = nr; return 2; }
				} else if (w.use_stale) {
					uint32_t nr = 0;
					if ((w.stale[kmerI >> 6] >> (kmerI & 63u)) & 1ull) w.req[nr++] = kmerI;
					if ((w.stale[kmerJ >> 6] >> (kmerJ & 63u)) & 1ull) w.req[nr++] = kmerJ;
					if (nr) { w.nreq = nr; return 2; }
				}
				// FillVisit(I) (bulgeremoval.cpp:352) has no side effects: it is evaluated when the first J needs it, and again
				// after a collapse that rewrote I's own window
				if (w.need_fill && w.wfill) { w.fill_i = kmerI; w.jready = true; return 5; }      // (this J again once the caller has filled `visit`)
				w.idJ++; w.jready = false;
				BT_PROF_ADD(t, 23);
				if (w.need_fill) { bt_fill_visit<L>(t, w, kmerI); w.need_fill = false; if (t.err) return 0; }
				BT_PROF_ADD(t, 20);
				const uint64_t *mkJ = wmk + (size_t)kmerJ * w.mks;
				const uint32_t limJ = wlen[kmerJ] < D ? wlen[kmerJ] : D, nmJ = wmn[kmerJ], nvisit = w.nvisit;
				for (uint32_t j = 0; j < nmJ; j++) {
					uint32_t step = (uint32_t)(mkJ[j] >> 32), nowBif = (uint32_t)mkJ[j];
					if (step >= limJ) break;
					if (nowBif == t.id) break;
					uint32_t lo = 0, hi = nvisit;                    // lower_bound(BifurcationMark(nowBif, 0))
					uint64_t probe = (uint64_t)nowBif << 32;
					while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (visit[mid] < probe) lo = mid + 1; else hi = mid; }
					if (lo < nvisit && (uint32_t)(visit[lo] >> 32) == nowBif) {
						uint32_t dJ = step, dI = (uint32_t)visit[lo];
						BT_PROF_ADD(t, 23);
						const bool ov_ = bt_overlap<L>(t, w, kmerI, dI, kmerJ, dJ);
						BT_PROF_ADD(t, 21);
						if (ov_) break;
						if (t.err) return 0;
						if (w.mscan && !w.mready && bt_marks_inside<L>(w, kmerI, dI) + bt_marks_inside<L>(w, kmerJ, dJ) > w.mscan_min) {
							w.mq_i = kmerI; w.mq_di = dI; w.mq_j = kmerJ; w.mq_dj = dJ;
							w.idJ--; w.jready = true;                    // this J again once the caller has the multiplicities (nothing has been decided or counted yet)
							return 4;
						}
						++w.ret;
						uint32_t imlp, jmlp;
						if (w.mready) { imlp = w.mres[0]; jmlp = w.mres[1]; }
						else bt_max_mult_pair<L>(t, w, kmerI, dI, kmerJ, dJ, imlp, jmlp);
						w.mready = false;
						BT_PROF_ADD(t, 22);
						if (imlp > jmlp || (imlp == jmlp && kmerI < kmerJ)) {
							endc[kmerJ] = endc[kmerI];
							w.c_src = kmerI; w.c_dS = dI; w.c_tgt = kmerJ; w.c_dT = dJ;
						} else {
							endc[kmerI] = endc[kmerJ];
							w.c_src = kmerJ; w.c_dS = dJ; w.c_tgt = kmerI; w.c_dT = dI;
							w.need_fill = true;                      // FillVisit(I) again, on the rescanned window
						}
						return 1;                                    // caller: collapse(c_src -> c_tgt), rescan (or bump epoch), then call again (next J)
					}
				}
			}
			w.inI = false; w.idI++;
		}
		w.gi++; w.pready = false;
		if (w.gi < w.ab.ngroups) w.idI = grp_off[w.gi];
	}
	if (!t.defer_cleanup) t.cleanup();           // (simplify.hip: Cleanup by all lanes once the loops are over)
	return 0;
}

// ------------------------------------------------------------------------------------------- reservation footprint
// Reservation footprint of a transaction.  f(id, kind):
//   kind 0  EXCLUSIVE: the id itself and every id marked in the core of an instance (what the transaction itself reads or
//           writes, both strands): their instance lists may be rewritten, nobody else may claim them this round;
//   kind 1  ORDERING: ids marked upstream on the same strand / further downstream on the opposite strand (instances walking
//           TOWARDS the core; those walking away cannot see or touch it).  The transaction can only make them stale:
//             x > id  it claims x (atomicMin) without needing to own it -- whoever is above x and can see x must wait;
//             x < id  it must find x unclaimed by anything at or below x (x itself live, or a lower id that may wake x up),
//                     see bt_order_blocked; x is not claimed (ids below the runner are never woken up by it).
// k_reserve (simplify.hip) walks exactly the same elements with 64 lanes.
__host__ __device__ inline bool bt_order_blocked(const GraphView &g, uint32_t x)
{
	uint32_t o = g.own[x];
	return (o & ~0xFFFFFu) == g.round_bits && g.win[o & 0xFFFFFu] <= x;
}

template <class F>
__host__ __device__ inline void bt_footprint(const GraphView &g, uint32_t id, F f)
{
	uint32_t back = g.D + g.k + 2, fwd = 2 * (g.D + g.k + 2) + g.k, core = g.D + 2 * g.k + 3;
	f(id, 0u);
	for (uint32_t pass = 0; pass < 2; pass++)                                   // all exclusive claims first (a claim list keeps the first kind it sees)
		for (uint32_t s = 0; s < 2; s++)
			for (uint32_t nd = g.head[s][id]; nd != BT_NONE; nd = g.nnext[nd]) {
				if (g.ndead[nd]) continue;
				uint32_t e0 = g.nslot[nd], e = e0;
				for (uint32_t i = 0; i <= fwd; i++) {
					if (i && g.ch[e] == BT_SEP) break;
					uint32_t b0 = g.bif[0][e], b1 = g.bif[1][e];
					if (i < core) { if (pass == 0) { if (b0 != BT_NONE) f(b0, 0u); if (b1 != BT_NONE) f(b1, 0u); } }
					else if (pass == 1) { uint32_t b = s ? b0 : b1; if (b != BT_NONE) f(b, 1u); }   // opposite strand only
					else break;
					e = s ? g.pv[e] : g.nx[e];
					if (e == BT_NONE) break;
				}
				if (pass == 0) continue;
				e = s ? g.nx[e0] : g.pv[e0];
				for (uint32_t i = 1; i <= back && e != BT_NONE; i++) {
					if (g.ch[e] == BT_SEP) break;
					uint32_t b = g.bif[s][e];
					if (b != BT_NONE) f(b, 1u);
					e = s ? g.nx[e] : g.pv[e];
				}
			}
}
