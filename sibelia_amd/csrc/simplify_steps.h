// simplify_steps.h -- per-thread bodies of the simplification kernels (simplify.hip).
//
// SimplifyGraph (reference src/blockfinder.cpp:16-51) is strictly ordered: for iter, for id ascending,
// RemoveBulges(id).  On the GPU one iteration becomes
//   1. snapshot:  AnyBulges verdict of EVERY id against the graph at iteration start, one wave per id (one thread here)
//                 (replaces ~instances x (D-1) x iterations cache-cold window steps of the reference);
//   2. ordered rounds over the ids that need to run (verdict true, or made stale by an earlier commit):
//        select   the W lowest pending ids,
//        reserve  each claims the ids marked in its neighbourhood with atomicMin(round | rank): exclusively where it
//                 reads or writes itself, as an ordering hint where it can only make a later id stale (bt_footprint),
//        commit   a transaction that owns its core and finds nothing below it about to run in its surroundings runs
//                 RemoveBulges for real; ids whose windows or lists it changed are marked pending if they are still
//                 ahead in the order.
//      A committed transaction is isolated inside its round (nobody else owns anything it can reach) and
//      no lower pending id can reach what it touches, so the result equals the sequential order.
//   3. validation: commits publish per-element and per-id read/write stamps; touching something
//      a HIGHER id already wrote (or, for writers, read) means a lower id became pending too late -- the
//      host then restores the iteration checkpoint and replays with that id as a fence.
// Results are therefore exact by construction + detection, never by assumption.
#pragma once
#include "bulge_txn.h"

#define SS_ROUND_MAX 4095u

// ---- one-thread forms (tests/hostsim); simplify.hip runs the same steps with 64 lanes scanning the windows
// `incremental`: an id nobody touched since its verdict was last taken (previous snapshot, a probe, or its own
// RemoveBulges) still has verdict false -- only touched ids are examined again.
__host__ __device__ inline void ss_snapshot(const GraphView &g, uint32_t id, uint8_t *arena, uint32_t arena_bytes, bool incremental)
{
	if (incremental && !g.touch[id]) { g.need[id] = 0; return; }
	g.touch[id] = 0;
	Txn t;
	BulgeWork w;
	t.init(g, id, 0, 0, arena, arena_bytes);
	bool v = false;
	if (bt_setup(t, w, true)) { bt_scan_all(t, w); bt_end_chars(t, w); v = bt_any_bulges(t, w, true); }
	if (t.err & BT_ERR_SCRATCH) v = true;      // undecidable in the small arena: let the ordered phase run it
	g.need[id] = v ? 1 : 0;
}

__host__ __device__ inline void ss_reserve(const GraphView &g, uint32_t widx)
{
	uint32_t id = g.win[widx], st = g.round_bits | widx;
	bt_footprint(g, id, [&](uint32_t b, uint32_t kind) { if (kind == 0 || b > id) bt_atomic_min(&g.own[b], st); });
}

__host__ __device__ inline void ss_mark_big(const GraphView &g, uint32_t id)
{
	g.big[id] = 1; g.need[id] = 1;
	bt_atomic_add(&g.ctr[CTR_BIG], 1u);
}

// Probe of a pending id between rounds (no writer is running): ids whose AnyBulges verdict is false NOW are retired
// without any reservation -- exactly like ids the snapshot found clean, they become pending again if a lower id later
// rewrites something they can see.  Returns true when the id really has bulges (it then goes through reserve / commit).
__host__ __device__ inline bool ss_probe(const GraphView &g, uint32_t widx, uint8_t *arena, uint32_t arena_bytes, uint32_t *err_out = nullptr)
{
	uint32_t id = g.win[widx];
	Txn t;
	BulgeWork w;
	t.init(g, id, widx, 3, arena, arena_bytes);
	bool has = false;
	if (bt_setup(t, w, true)) { bt_scan_all(t, w); bt_end_chars(t, w); has = bt_any_bulges(t, w, true); }
	if (err_out) *err_out = t.err;
	if (t.err) return true;                    // undecidable here: let the commit path sort it out
	if (!has) { g.need[id] = 0; g.touch[id] = 0; bt_atomic_add(&g.ctr[CTR_COMMITTED], 1u); }   // verdict taken now: clean until touched again
	return has;
}

// the transaction proper, for a window entry that owns its whole neighbourhood
// alone: nothing else is in flight (solo round / serial chain) -- the precondition of lazy windows in the kernels, where an ordered
// round needs the set of windows a collapse dirtied for its reservation check; this one-thread form has no such check and takes
// lazy windows whenever the id is large enough (g.lazy_min lets the tests force them everywhere)
__host__ __device__ inline void ss_commit_run(const GraphView &g, uint32_t widx, uint8_t *arena, uint32_t arena_bytes,
                                              uint8_t *fast = nullptr, uint32_t fast_bytes = 0)
{
	uint32_t id = g.win[widx];
	g.need[id] = 0;                            // cleared BEFORE running: a later push must survive
	g.touch[id] = 1;                           // whatever it leaves behind is examined again by the next snapshot
	Txn t;
	BulgeWork w;
	t.init(g, id, widx, 1, arena, arena_bytes);
	bool has = false;
	if (bt_setup(t, w)) { bt_scan_all(t, w); bt_end_chars(t, w); has = bt_any_bulges(t, w, true); }
	if (t.err & BT_ERR_SCRATCH) { ss_mark_big(g, id); return; }   // nothing written yet: big-arena path
	bt_atomic_add(&g.ctr[CTR_COMMITTED], 1u);
	bt_atomic_add(&g.ctr[CTR_TXN], 1u);
	if (!has) return;
	t.init(g, id, widx, 2, arena, arena_bytes);                   // writer pass: publish reads and writes
	t.fscr = fast; t.fscr_cap = fast_bytes;                       // (the kernels put summaries, mark lists and the AnyBulges map in LDS)
	w.ret = 0;
	bt_setup(t, w);
	bt_scan_all(t, w);
	w.lazy = !t.err && w.wep != nullptr;
	w.jscan = w.lazy;                                              // (exercises the caller-side J search of the kernels)
	w.wfill = w.lazy;                                              // (... and the caller-side FillVisit)
	w.pscan = w.lazy;                                              // (... and the caller-side search for the next I that has a J)
	w.mscan = g.test_lazy_map != 0;                                // (tests/hostsim, HOSTSIM_LAZY_MAP: ... and the caller-side multiplicities)
	int more = !t.err && bt_rb_begin(t, w) ? 1 : 0;
	// windows that see the region a collapse rewrites (target start .. end of its look-forward flank) are the only ones whose
	// cache changes: like k_commit (simplify.hip), only those are rescanned -- normally just the target's own window
	uint64_t *dirty = more && !w.lazy ? (uint64_t *)t.alloc(((w.n + 63) / 64) * 8) : nullptr;
	if (more && !w.lazy && !dirty) more = 0;
	w.use_stale = more && !w.lazy && g.test_lazy_map != 0 && w.n <= 256;      // (tests/hostsim, HOSTSIM_LAZY_MAP: ... and stale-marking instead of eager rescans)
	while (more) {
		more = bt_rb_run(t, w);
		if (t.err) break;
		if (more == 3) bt_rb_next_j(t, w);
		else if (more == 4) bt_rb_mults(t, w);
		else if (more == 5) { bt_fill_visit(t, w, w.fill_i); w.need_fill = false; if (t.err) break; }
		else if (more == 6) bt_rb_next_pair(t, w);
		else if (more == 2) {                                          // lazy run: the loops need these windows as of now
			for (uint32_t x = 0; x < w.nreq; x++) {
				bt_scan_instance(t, w, w.req[x]);
				if (w.lazy) w.wep[w.req[x]] = w.epoch; else w.stale[w.req[x] >> 6] &= ~(1ull << (w.req[x] & 63));
			}
			if (!w.lazy && w.mk_overflow) {
				bt_marks_to_arena(t, w);
				if (!t.err) for (uint32_t i = 0; i < w.n; i++) bt_scan_instance(t, w, i);
				w.stale[0] = w.stale[1] = w.stale[2] = w.stale[3] = 0;
			}
			if (t.err) break;
		} else if (more && w.lazy) {
			bt_collapse(t, w, w.c_src, w.c_dS, w.c_tgt, w.c_dT);
			if (t.err) break;
			w.epoch++;                                                 // every cached window is stale until somebody asks for it
		} else if (more) {
			const uint32_t tg = w.c_tgt, span = 2 * g.k + w.c_dT + 1;
			const uint32_t tl = w.wlen[tg] + 1 < w.ws ? w.wlen[tg] + 1 : w.ws;
			if (w.use_stale)                                                // stale windows the collapse might reach are brought up to date first (as k_commit does)
				for (uint32_t i = 0; i < w.n; i++)
					if ((w.stale[i >> 6] >> (i & 63)) & 1ull) {
						const uint32_t len = (w.wlen[i] + 1 < w.ws ? w.wlen[i] + 1 : w.ws) + w.wdel[i] + (w.c_dT > w.c_dS ? w.c_dT - w.c_dS : 0u);
						if (w.wnb[i] != 0 || bt_windows_intersect(w, i, len, tg, span < tl ? span : tl) != 0) { bt_scan_instance(t, w, i); w.stale[i >> 6] &= ~(1ull << (i & 63)); }
					}
			if (w.mk_overflow) { bt_marks_to_arena(t, w); if (!t.err) for (uint32_t i = 0; i < w.n; i++) bt_scan_instance(t, w, i); w.stale[0] = w.stale[1] = w.stale[2] = w.stale[3] = 0; }
			for (uint32_t i = 0; i < w.n; i++) {
				if ((i & 63) == 0) dirty[i >> 6] = 0;
				const uint32_t len = w.wlen[i] + 1 < w.ws ? w.wlen[i] + 1 : w.ws;      // cached steps incl. the separator step
				const bool st = w.use_stale && ((w.stale[i >> 6] >> (i & 63)) & 1ull);
				if (i == tg || (!st && bt_windows_intersect(w, i, len, tg, span < tl ? span : tl) != 0)) dirty[i >> 6] |= 1ull << (i & 63);
			}
			bt_collapse(t, w, w.c_src, w.c_dS, w.c_tgt, w.c_dT);
			if (t.err) break;
			if (w.c_dT > w.c_dS) for (uint32_t i = 0; i < w.n; i++) if ((dirty[i >> 6] >> (i & 63)) & 1ull) w.wdel[i] += w.c_dT - w.c_dS;
			if (w.use_stale) { for (uint32_t q = 0; q < 4 && q < (w.n + 63) / 64; q++) w.stale[q] |= dirty[q]; continue; }
			for (uint32_t i = 0; i < w.n; i++) if ((dirty[i >> 6] >> (i & 63)) & 1ull) bt_scan_instance(t, w, i);
			if (w.mk_overflow) { bt_marks_to_arena(t, w); if (!t.err) for (uint32_t i = 0; i < w.n; i++) bt_scan_instance(t, w, i); }
		}
	}
	if (t.err) {
		if (!t.wrote && t.err == BT_ERR_SCRATCH) { ss_mark_big(g, id); return; }
		bt_atomic_or(&g.ctr[CTR_ERR], t.err);
	}
	bt_atomic_add(&g.ctr[CTR_BULGES], w.ret);
}

// ---- the block index (GraphView::bidx) read by ONE thread: reference forms of what k_probe / k_reserve evaluate with 64 lanes
// (simplify.hip: probe_idx, reserve_idx).  tests/hostsim runs them beside ss_probe / bt_footprint on every probed / reserved entry and
// compares, and rebuilds the index from the arrays at the end of a stage: the maintenance sites (bt_idx_mark / bt_idx_dirty /
// bt_idx_wstamp) are complete iff the two agree on every vector.
struct IdxRec { unsigned long long m[2], sep; uint32_t wmaxb, dirty; };
__host__ __device__ inline IdxRec bt_idx_load(const GraphView &g, uint32_t blk)
{
	IdxRec r;
	const unsigned long long *p = g.bidx + (size_t)blk * BT_IDX_WORDS;
	r.m[0] = p[0]; r.m[1] = p[1]; r.sep = p[2]; r.wmaxb = (uint32_t)p[3]; r.dirty = (uint32_t)(p[3] >> 32);
	return r;
}
// AnyBulges verdict of id from the index: 1 live, 0 clean, -2 the index cannot serve one of its windows (a fresh slot, a block that is
// no longer pristine, or -- with tid != 0 -- a block that carries a write stamp above tid: the walking path makes the exact order check)
__host__ __device__ inline int ss_verdict_idx(const GraphView &g, uint32_t id, uint32_t tid)
{
	if (!g.bidx) return -2;
	const uint32_t k = g.k, D = g.D, ws = D + k + 2;
	enum { CAP = 2048 };
	uint32_t keys[CAP]; uint8_t masks[CAP];
	for (uint32_t i = 0; i < CAP; i++) { keys[i] = BT_NONE; masks[i] = 0; }
	uint32_t used = 0;
	bool found = false;
	for (uint32_t s = 0; s < 2; s++)
		for (uint32_t nd = g.head[s][id]; nd != BT_NONE; nd = g.nnext[nd]) {
			if (g.ndead[nd]) continue;
			const uint32_t a = g.nslot[nd];
			if (a >= g.norig) return -2;
			uint32_t len = ws;
			for (uint32_t t = 0; t < ws; t++) {
				if (s && t > a) { len = t; break; }
				const uint32_t slot = s ? a - t : a + t;
				if (slot >= g.norig) { len = t; break; }
				const IdxRec r = bt_idx_load(g, slot >> 6);
				if (r.dirty) return -2;
				if ((r.sep >> (slot & 63u)) & 1ull) { len = t; break; }
				if (tid && r.wmaxb > tid) return -2;
			}
			if (len < k + 1) continue;                                // endChar ' ': the instance takes no part
			const uint8_t raw = g.ch[s ? a - k : a + k];
			const char ec = s ? bt_comp((char)raw) : (char)raw;
			const uint8_t bit = ec == 'A' ? 1 : ec == 'C' ? 2 : ec == 'G' ? 4 : 8;
			const uint32_t lim = len < D ? len : D;
			for (uint32_t t = 1; t < lim; t++) {
				const uint32_t slot = s ? a - t : a + t;
				const IdxRec r = bt_idx_load(g, slot >> 6);
				if (!((r.m[s] >> (slot & 63u)) & 1ull)) continue;
				const uint32_t b = g.bif[s][slot];
				if (b == id) break;
				uint32_t h = (b * 2654435761u) >> 21;
				for (;;) {
					if (keys[h] == BT_NONE) { if (++used > CAP / 2) return -2; keys[h] = b; }
					if (keys[h] == b) { masks[h] |= bit; if (masks[h] & (masks[h] - 1)) found = true; break; }
					h = (h + 1) & (CAP - 1);
				}
			}
		}
	return found ? 1 : 0;
}
// bt_footprint from the index: calls f exactly for the (id, kind) pairs bt_footprint reports (possibly in another order / multiplicity);
// false: the index cannot serve the neighbourhood of one of the instances (nothing may have been reported then -- callers collect first)
template <class F>
__host__ __device__ inline bool bt_footprint_idx(const GraphView &g, uint32_t id, F f)
{
	if (!g.bidx) return false;
	const uint32_t back = g.D + g.k + 2, fwd = 2 * (g.D + g.k + 2) + g.k, core = g.D + 2 * g.k + 3;
	f(id, 0u);
	for (uint32_t s = 0; s < 2; s++)
		for (uint32_t nd = g.head[s][id]; nd != BT_NONE; nd = g.nnext[nd]) {
			if (g.ndead[nd]) continue;
			const uint32_t a = g.nslot[nd];
			if (a >= g.norig) return false;
			for (uint32_t t = 0; t <= fwd; t++) {                     // ahead: the core (both strands, exclusive), then the opposite strand (ordering)
				if (s && t > a) break;
				const uint32_t slot = s ? a - t : a + t;
				if (slot >= g.norig) break;
				const IdxRec r = bt_idx_load(g, slot >> 6);
				if (r.dirty) return false;
				const unsigned long long bit = 1ull << (slot & 63u);
				if (t && (r.sep & bit)) break;
				if (t < core) { if (r.m[0] & bit) f(g.bif[0][slot], 0u); if (r.m[1] & bit) f(g.bif[1][slot], 0u); }
				else if (r.m[s ^ 1u] & bit) f(g.bif[s ^ 1u][slot], 1u);
			}
			for (uint32_t t = 1; t <= back; t++) {                    // behind: the own strand (ordering)
				if (!s && t > a) break;
				const uint32_t slot = s ? a + t : a - t;
				if (slot >= g.norig) break;
				const IdxRec r = bt_idx_load(g, slot >> 6);
				if (r.dirty) return false;
				const unsigned long long bit = 1ull << (slot & 63u);
				if (r.sep & bit) break;
				if (r.m[s] & bit) f(g.bif[s][slot], 1u);
			}
		}
	return true;
}

__host__ __device__ inline bool ss_owns_footprint(const GraphView &g, uint32_t widx)
{
	uint32_t id = g.win[widx], st = g.round_bits | widx;
	bool owner = true;
	bt_footprint(g, id, [&](uint32_t b, uint32_t kind) {
		if (kind == 0) { if (g.own[b] != st) owner = false; }
		else if (b < id && g.own[b] != st && bt_order_blocked(g, b)) owner = false;      // (own == st: also in a core, claimed exclusively)
	});
	return owner;
}

__host__ __device__ inline void ss_commit(const GraphView &g, uint32_t widx, uint8_t *arena, uint32_t arena_bytes, bool solo)
{
	if (!solo && !ss_owns_footprint(g, widx)) return;      // stays pending
	ss_commit_run(g, widx, arena, arena_bytes);
}
