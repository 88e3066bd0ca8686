// commit.hip -- the transactions: RemoveBulges of one id as a wave (reference src/bulgeremoval.cpp:330-430 and everything it calls),
// for the owners of an ordered round (k_commit), one after the other where conflict neighbourhoods are dense (k_chain), and the whole
// SimplifyGraph of a tiny input in one launch (k_dense_stage).  bulge_txn.h holds the decision logic shared with tests/hostsim.
#include <cstring>
#include <algorithm>
#include <vector>
#include <hip/hip_runtime.h>
// cycle counters of the decision loops (bulge_txn.h: BT_PROF_ADD), device only
__device__ unsigned long long g_phase_cycles[32];   // SBL_PHASES=1 debug: summed s_memtime deltas of k_commit's phases
#if defined(__HIP_DEVICE_COMPILE__)
#define BT_PROF_T0(t) do { if ((t).prof) (t).prof_t = __builtin_readcyclecounter(); } while (0)
#define BT_PROF_ADD(t, i) do { if ((t).prof) { unsigned long long n_ = __builtin_readcyclecounter(); atomicAdd(&g_phase_cycles[i], n_ - (t).prof_t); (t).prof_t = n_; } } while (0)
#endif
#include "simplify_walks.h"

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
#ifdef SBL_DBG_IDRET
// debug build only (tools/diag_case.py): collapses reported and parks per id, summed over the iterations of a stage
__device__ unsigned g_dbg_ret[1u << 20], g_dbg_park[1u << 20], g_dbg_fin[1u << 20];
__device__ unsigned g_dbg_log[8u << 20], g_dbg_nlog;      // every collapse: id, round, ret, source element, target element, dS, dT, resumed
extern "C" unsigned sbl_dbg_log(unsigned *out, unsigned cap)
{
	unsigned n = 0;
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_dbg_nlog), 4);
	if (n > cap) n = cap;
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg_log), (size_t)n * 32u);
	return n;
}
extern "C" void sbl_dbg_idret(unsigned *ret, unsigned *park, unsigned *fin, unsigned n, int reset)
{
	(void)hipDeviceSynchronize();
	(void)hipMemcpyFromSymbol(ret, HIP_SYMBOL(g_dbg_ret), n * 4u); (void)hipMemcpyFromSymbol(park, HIP_SYMBOL(g_dbg_park), n * 4u); (void)hipMemcpyFromSymbol(fin, HIP_SYMBOL(g_dbg_fin), n * 4u);
	if (reset) { static unsigned z[1u << 20]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_ret), z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_park), z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_fin), z, sizeof z); }
}
#endif
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

// ---- the caller side of BulgeWork::pready (bt_rb_next_pair with 64 lanes): the next member I of the group, at or after idI, that is valid
// and has a valid later member with another endChar, and the first such J.  Pass A: per endChar, the LAST valid member that carries it;
// pass B: the first valid member before the last one of some other endChar.  Three dependent look-ups per 64 members and pass
// (member -> instance -> node's dead flag); the I loop of bt_rb_run spent a trip through wave_next_j on every member.
__device__ __forceinline__ void wave_next_pair(const GraphView &g, Txn &t, BulgeWork &w, unsigned lane)
{
	const unsigned gs = w.idI, ge = w.ab.grp_off[w.gi + 1];
	WSYNC();                                                       // (everybody has read idI before lane 0 moves it)
	enum { NC = 8 };
	unsigned cls_c[NC] = {0, 0, 0, 0, 0, 0, 0, 0}, cls_last[NC] = {0, 0, 0, 0, 0, 0, 0, 0}, ncls = 0;                    // (uniform: endChars seen among the valid members, index of the last one + 1)
	bool too_many = false;
	auto load = [&](unsigned j0, unsigned &ec, bool &valid) {
		const unsigned idx = j0 + lane;
		const bool in = idx < ge;
		const unsigned m = in ? ldx(&w.ab.grp_mem[idx]) : 0u;
		const unsigned st = in ? ldx(&w.start[m]) : 0u;
		ec = in ? (unsigned)(unsigned char)ldx(&w.endc[m]) : 0u;
		valid = in && !g.ndead[st >> 1];
	};
	const bool one = ge - gs <= 64u;
	unsigned ec0 = 0; bool v0 = false;
	for (unsigned j0 = gs; j0 < ge && !too_many; j0 += 64) {
		unsigned ec; bool valid;
		load(j0, ec, valid);
		if (one) { ec0 = ec; v0 = valid; }
		unsigned long long rem = __ballot(valid);
		while (rem) {
			const unsigned c = (unsigned)__shfl((int)ec, (int)__builtin_ctzll(rem));
			const unsigned long long mask = __ballot(valid && ec == c);
			const unsigned last = j0 + 64u - (unsigned)__builtin_clzll(mask);      // index of the last one + 1
			bool known = false;                                        // (fixed-trip loops: the table stays in registers)
#pragma unroll
			for (int q = 0; q < NC; q++) if ((unsigned)q < ncls && cls_c[q] == c) { known = true; cls_last[q] = last > cls_last[q] ? last : cls_last[q]; }
			if (!known) {
				if (ncls == NC) { too_many = true; break; }
#pragma unroll
				for (int q = 0; q < NC; q++) if ((unsigned)q == ncls) { cls_c[q] = c; cls_last[q] = last; }
				ncls++;
			}
			rem &= ~mask;
		}
	}
	unsigned found = ge, pj = 0;
	if (!too_many) {
		for (unsigned j0 = gs; j0 < ge && found == ge; j0 += 64) {
			unsigned ec; bool valid;
			if (one) { ec = ec0; valid = v0; } else load(j0, ec, valid);
			unsigned other = 0;                                        // the last valid member with another endChar, + 1
#pragma unroll
			for (int q = 0; q < NC; q++) if ((unsigned)q < ncls && cls_c[q] != ec && cls_last[q] > other) other = cls_last[q];
			const unsigned idx = j0 + lane;
			const unsigned long long b = __ballot(valid && other > idx + 1u);
			if (b) {
				const unsigned li = (unsigned)__builtin_ctzll(b);
				found = j0 + li;
				const unsigned eci = (unsigned)__shfl((int)ec, (int)li);
				const unsigned long long later = __ballot(valid && ec != eci) & (li == 63u ? 0ull : (~0ull << (li + 1u)));
				pj = later ? j0 + (unsigned)__builtin_ctzll(later) : 0u;    // (0: in a later block of 64 -- wave_next_j finds it)
			}
		}
	}
	if (lane == 0) {
		if (too_many) bt_rb_next_pair(t, w);                            // (more endChars than the table holds -- no alphabet of this program has: the one-thread form)
		else { w.idI = found; w.pready = true; w.pj = pj; w.pjknown = pj != 0u; }
	}
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

// ---- the caller side of BulgeWork::wfill: FillVisit (bulgeremoval.cpp:122-146, bt_fill_visit) with 64 lanes.  One thread's shell sort of
// the ~15 (id, distance) pairs of a window is ~100 dependent LDS round trips (26 k cycles per call, once per I and again after every
// collapse that rewrote I's window); here every lane holds one pair and counts the smaller ones.
template <int NCH>
__device__ __forceinline__ void wave_fill_visit_t(Txn &t, BulgeWork &w, unsigned D, unsigned lane)
{
	const unsigned i = w.fill_i, nm = ldx(&w.wmn[i]);
	const unsigned long long *mk = reinterpret_cast<const unsigned long long *>(w.wmk) + (size_t)i * w.mks;
	unsigned long long *visit = reinterpret_cast<unsigned long long *>(w.visit);
	const unsigned wl = ldx(&w.wlen[i]), start = ldx(&w.wst[i]), lim = wl < D ? wl : D;
	unsigned long long key[NCH];
	unsigned n = 64u * NCH;
#pragma unroll
	for (int u = 0; u < NCH; u++) {
		const unsigned j = lane + 64u * u;
		const unsigned long long v = j < nm ? ldx(&mk[j]) : ~0ull;
		const unsigned step = (unsigned)(v >> 32), b = (unsigned)v;
		const unsigned long long ms = __ballot(j >= nm || step >= lim || b == start);
		if (ms && n == 64u * NCH) n = 64u * u + (unsigned)__builtin_ctzll(ms);      // the first mark the walk stops at
		key[u] = ((unsigned long long)b << 32) | step;
	}
	bool over = false;
	if (n > w.visit_cap) { n = w.visit_cap; over = true; }
	unsigned rank[NCH];
#pragma unroll
	for (int u = 0; u < NCH; u++) { rank[u] = 0; if (lane + 64u * u >= n) key[u] = ~0ull; }
#pragma unroll
	for (int c = 0; c < NCH; c++) {                                        // (the pairs are distinct: every step occurs once)
		const unsigned upto = n > 64u * c ? (n - 64u * c < 64u ? n - 64u * c : 64u) : 0u;
		for (unsigned y = 0; y < upto; y++) {
			const unsigned long long ky = __shfl(key[c], y);
#pragma unroll
			for (int u = 0; u < NCH; u++) rank[u] += ky < key[u] ? 1u : 0u;
		}
	}
#pragma unroll
	for (int u = 0; u < NCH; u++) if (lane + 64u * u < n) stx(&visit[rank[u]], key[u]);
	if (lane == 0) { w.nvisit = n; w.need_fill = false; if (over) t.err |= BT_ERR_SCRATCH; }
	WSYNC();
}
__device__ __forceinline__ void wave_fill_visit(Txn &t, BulgeWork &w, unsigned D, unsigned lane)
{
	const unsigned nm = ldx(&w.wmn[w.fill_i]);
	if (nm <= 64u) wave_fill_visit_t<1>(t, w, D, lane);
#ifndef SBL_VAR_NOFV3
	else if (nm <= 192u) wave_fill_visit_t<3>(t, w, D, lane);                // (62 strains: ~90 marks per window, lists in the arena -- one thread's sort there was 330 k cycles per transaction)
#endif
	else {                                                                 // (longer than any window of D <= 150 steps: the one-thread form)
		if (lane == 0) { if (bt_scratch_in_lds(w)) bt_fill_visit<true>(t, w, w.fill_i); else bt_fill_visit<false>(t, w, w.fill_i); w.need_fill = false; }
		WSYNC();
	}
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
	// LS_AHEAD steps at a time, on the guess that the list is laid out in consecutive slots there (it nearly always is): the loads of a
	// batch are issued together and consumed as far as the links confirm the guess.  (One step per dependent round trip -- character,
	// mark, link -- made a window of D + k + 2 = 400 steps 400 round trips: 0.3 ms per 64 instances, and a stage of 62 000 calls on
	// 57 strains 32 s; round 6.)
	enum { LS_AHEAD = 8 };
	bool end = false;
	while (s < ws && !end) {
		uint8_t cc[LS_AHEAD]; unsigned bb[LS_AHEAD], ll[LS_AHEAD], ee[LS_AHEAD];
#pragma unroll
		for (int u = 0; u < LS_AHEAD; u++) {
			const bool in = dir ? (unsigned)u <= e : (unsigned long long)e + (unsigned)u < g.cap_e;
			ee[u] = in ? (dir ? e - (unsigned)u : e + (unsigned)u) : e;
			cc[u] = g.ch[ee[u]]; bb[u] = g.bif[dir][ee[u]]; ll[u] = dir ? g.pv[ee[u]] : g.nx[ee[u]];
		}
		unsigned nxt = e;
#pragma unroll
		for (int u = 0; u < LS_AHEAD; u++) {
			if (u && (ll[u - 1] != ee[u] || ee[u] == ee[u - 1])) break;      // the guess ends here: go on from where the link leads
			if (s >= ws) { end = true; break; }
			const uint8_t c = cc[u];
			const unsigned b = bb[u];
			if (s == 0) w.wst[i] = b;
			if (s == k) ck = dir ? bt_comp((char)c) : (char)c;
			if (c == BT_SEP) { end = true; break; }
			if (s && b != BT_NONE) { if (nm < w.mks) mk[nm] = ((unsigned long long)s << 32) | b; nm++; }
			s++;
			nxt = ll[u];
		}
		e = nxt;
		if (e == BT_NONE) break;
	}
	w.wlen[i] = s; w.wmn[i] = nm; w.wck[i] = ck;
	w.endc[i] = s >= k + 1 ? ck : ' ';                                 // bt_end_chars
}

// endChar of instance i alone (bulgeremoval.cpp:340-347): the character k steps on if the k + 1 characters from the instance are valid
__device__ __forceinline__ char lane_end_char(const GraphView &g, const BulgeWork &w, unsigned i)
{
	const unsigned dir = w.start[i] & 1u, k = g.k;
	unsigned e = w.sel[i], s = 0;
	enum { LS_AHEAD = 8 };
	for (;;) {
		uint8_t cc[LS_AHEAD]; unsigned ll[LS_AHEAD], ee[LS_AHEAD];
#pragma unroll
		for (int u = 0; u < LS_AHEAD; u++) {
			const bool in = dir ? (unsigned)u <= e : (unsigned long long)e + (unsigned)u < g.cap_e;
			ee[u] = in ? (dir ? e - (unsigned)u : e + (unsigned)u) : e;
			cc[u] = g.ch[ee[u]]; ll[u] = dir ? g.pv[ee[u]] : g.nx[ee[u]];
		}
		unsigned nxt = e;
#pragma unroll
		for (int u = 0; u < LS_AHEAD; u++) {
			if (u && (ll[u - 1] != ee[u] || ee[u] == ee[u - 1])) break;
			if (cc[u] == BT_SEP) return ' ';
			if (s == k) return dir ? bt_comp((char)cc[u]) : (char)cc[u];
			s++;
			nxt = ll[u];
		}
		e = nxt;
		if (e == BT_NONE) return ' ';
	}
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


// The same `run` insertions by the lanes that hold the ids (lane f + x holds the x-th id of the run): the first occurrence of an id in
// the run takes entry size + (number of first occurrences before it) -- the entry indices, and with them the log the Boost restatement
// replays, are those of the one-thread loop -- and claims a slot of the shadow table with a compare-and-swap (where an id ends up in an
// open-addressing table does not matter to a look-up).  The first instance of each endChar brings ~15 ids at 8 strains, ~90 at 62: one
// thread spent 1.5 - 3 k cycles on each.
__device__ __forceinline__ int ab_lazy_wave(Txn &t, BulgeWork &w, ABShared &sh, unsigned lane, unsigned i, char ec, unsigned f, unsigned run, unsigned b,
                                            unsigned slots, unsigned shift, bool estimate)
{
	ABuild &a = w.abb;
	unsigned *const key = a.m.key, *const mhead = a.mhead, *const mtail = a.mtail, *const mcnt = a.mcnt, *const log_inst = a.log_inst, *const log_next = a.log_next;
	unsigned *const skey = sh.skey, *const sval = sh.sval;
	char *const echar = a.echar;
	const unsigned size = a.m.size, nlog = a.nlog, cap = a.m.cap, logcap = a.logcap, distinct = sh.distinct;
	const bool in = lane >= f && lane < f + run;
	bool first = in;                                                       // no earlier lane of the run holds the same id
	for (unsigned y = 0; y + 1 < run; y++) { const unsigned by = (unsigned)__shfl((int)b, (int)(f + y)); if (in && lane > f + y && by == b) first = false; }
	const unsigned long long fm = __ballot(first);
	const unsigned total = (unsigned)__popcll(fm), mine = (unsigned)__popcll(fm & ((1ull << lane) - 1ull));
	if (estimate && size + total > distinct) return -2;                    // more distinct ids than estimated: again, with the counting pass (nothing of this attempt is kept)
	if (size + total > cap || nlog + total > logcap) { if (lane == 0) t.err |= BT_ERR_SCRATCH; return -1; }
	if (first) {
		const unsigned kt = size + mine, nl = nlog + mine;
		stx(&key[kt], b); stx(&echar[kt], ec);
		stx(&log_inst[nl], i); stx(&log_next[nl], (unsigned)BT_NONE);
		stx(&mhead[kt], nl); stx(&mtail[kt], nl); stx(&mcnt[kt], 1u);
	}
	// slots of the shadow table.  In LDS: compare-and-swap.  In the arena: plain loads and stores only (an atomic is performed in the L2 and
	// the lanes' later look-ups are ordinary loads through the L1) -- lanes that want the same slot settle it among themselves: the lowest
	// lane takes it, the others move on.
	if (BT_IS_LDS(skey)) {
		if (first) {
			unsigned hh = (b * 2654435761u) >> shift;
			for (;;) {
				if (casx(&skey[hh], (unsigned)BT_NONE, b) == BT_NONE) break;
				hh = (hh + 1) & (slots - 1);
			}
			stx(&sval[hh], ((size + mine) << 8) | (unsigned char)ec);
		}
	} else {
		bool pending = first;
		unsigned hh = (b * 2654435761u) >> shift;
		while (__ballot(pending)) {
			if (pending) { while (ldg(&skey[hh]) != BT_NONE) hh = (hh + 1) & (slots - 1); }      // (slots taken before this run)
			bool lose = false;
			for (unsigned y = 0; y < run; y++) {
				const unsigned hy = (unsigned)__shfl((int)hh, (int)(f + y));
				const bool py = __shfl((int)pending, (int)(f + y)) != 0;
				if (pending && py && f + y < lane && hy == hh) lose = true;
			}
			if (pending && !lose) { stg(&skey[hh], b); stg(&sval[hh], ((size + mine) << 8) | (unsigned)(unsigned char)ec); pending = false; }
			else if (pending) hh = (hh + 1) & (slots - 1);
		}
	}
	if (lane == 0) { a.m.size = size + total; a.nlog = nlog + total; }
	return sh.mode;
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
	const bool estimate = attempt == 0 && g.ab_estimate && (n > 32u || !(g.test_flags & 2048u));      // (round 5: small ids too -- the counting pass was 4 % of a transaction; 2048: measurement switch)
	if (sh.mode && estimate) {
		unsigned mx = 0;
		for (unsigned i = lane; i < n; i += 64) { const unsigned v = ldx(&endc[i]) == ' ' ? 0u : ldx(&wmn[i]); mx = v > mx ? v : mx; }
#pragma unroll
		for (int dd = 32; dd > 0; dd >>= 1) { const unsigned v = __shfl_xor(mx, dd); mx = v > mx ? v : mx; }
		WSYNC();
		if (lane == 0) {
			t.fscr_used = mark;
			const unsigned distinct = n > 32u ? 2 * mx + 32 : mx + (mx >> 2) + 8;      // (a handful of homologous instances: the longest list plus what the minority branches add)
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
	// (the marks of the NEXT instance -- up to ABPF x 64 of them: a whole window -- are requested while this one is worked on: with dozens
	// of instances the lists live in the arena, and every instance used to begin with a memory round trip of its own, and with another
	// one per further chunk of 64 marks (62 strains: ~90 marks per window); a chunk that starts in the middle of a block of 64 -- after
	// an event -- is put together from two of the registers)
#ifndef ABPF_N
#define ABPF_N 1             // (3 = a whole window of the next instance in flight: -0.7 % at 62 strains, + 0.3 ms of k_commit at 8 -- registers)
#endif
	enum { ABPF = ABPF_N };
	unsigned long long vpre[ABPF];
#pragma unroll
	for (int u = 0; u < ABPF; u++) vpre[u] = ~0ull;
	unsigned pre_i = n;
	auto chunk_at = [&](unsigned ii, unsigned u) { const unsigned long long *m0 = wmk + (size_t)ii * mks; const unsigned jj = lane + 64u * u; return jj < ldx(&wmn[ii]) ? ldx(&m0[jj]) : ~0ull; };
	for (unsigned i = 0; i < n && !bad; i++) {
		const char ec = ldx(&endc[i]);
		if (ec == ' ') continue;
		const unsigned long long *mk = wmk + (size_t)i * mks;
		const unsigned wl = ldx(&wlen[i]);
		const unsigned start = ldx(&wst[i]), lim = wl < D ? wl : D, nm = ldx(&wmn[i]);
		unsigned long long vc[ABPF];
#pragma unroll
		for (int u = 0; u < ABPF; u++) vc[u] = pre_i == i ? vpre[u] : chunk_at(i, (unsigned)u);
		if (i + 1 < n) {
#pragma unroll
			for (int u = 0; u < ABPF; u++) vpre[u] = chunk_at(i + 1, (unsigned)u);
			pre_i = i + 1;
		}
		unsigned pos = 0;
		while (pos < nm) {
			unsigned j = pos + lane;
			unsigned long long v;
			const unsigned c0 = pos >> 6, l0 = pos & 63u;
			if (c0 + 1u < (unsigned)ABPF || (c0 < (unsigned)ABPF && l0 == 0u)) {
				const unsigned lj = (l0 + lane) & 63u;
				const unsigned long long lo = c0 == 0u ? vc[0] : c0 == 1u ? vc[ABPF > 1 ? 1 : 0] : vc[ABPF > 2 ? 2 : 0], hi = c0 == 0u ? vc[ABPF > 1 ? 1 : 0] : vc[ABPF > 2 ? 2 : 0];
				const unsigned long long va = __shfl(lo, lj), vb = __shfl(hi, lj);
				v = l0 + lane < 64u ? va : c0 + 1u < (unsigned)ABPF ? vb : ~0ull;
			} else v = j < nm ? ldx(&mk[j]) : ~0ull;
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
			if (w.abb.lazy && eev == 1u && !(g.test_flags & 512u)) {              // the run by the lanes that hold its ids
				const int m = ab_lazy_wave(t, w, sh, lane, i, ec, f, run, b, slots, shift, estimate);
				WSYNC();
				if (lane == 0) sh.mode = m;
				WSYNC();
				if (m < 0) { bad = true; break; }
				pos += f + run;
				continue;
			}
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
// ---- parked transactions (GraphView::park_of): the LDS state of a transaction at the end of its arena slice (PARK_IMG bytes, bulge_txn.h)
// The image holds raw LDS addresses (t.fscr, the w.* arrays laid out in `fast`, absh.skey / sval): it is only valid in a kernel that
// places t, w, absh and fast where the parking kernel had them.  k_commit and k_resume instantiate the same declarations
// (commit_kernel), so they do; the image records the four addresses and park_load refuses (BT_ERR_LAYOUT) an image from another layout.
#define PARK_OFF_LAYOUT (PARK_IMG - 16u)
#define PARK_OFF_W ((unsigned)((sizeof(Txn) + 15u) & ~15u))
#define PARK_OFF_AB (PARK_OFF_W + (unsigned)((sizeof(BulgeWork) + 15u) & ~15u))
#define PARK_OFF_FAST (PARK_OFF_AB + (unsigned)((sizeof(ABShared) + 15u) & ~15u))
static_assert(sizeof(Txn) % 4 == 0 && sizeof(BulgeWork) % 4 == 0 && sizeof(ABShared) % 4 == 0, "park_move copies words");
static_assert(PARK_OFF_FAST + COMMIT_FAST_BYTES <= PARK_OFF_LAYOUT, "the LDS image of a transaction must fit PARK_IMG");
__device__ __forceinline__ unsigned lds_addr(const void *p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p; }
__device__ __forceinline__ void park_move(unsigned *dst, const unsigned *src, unsigned words, unsigned lane) { for (unsigned i = lane; i < words; i += 64) dst[i] = src[i]; }
__device__ __forceinline__ void park_store(const GraphView &g, Txn &t, BulgeWork &w, ABShared &absh, uint8_t *fast, unsigned fast_bytes, uint8_t *image, unsigned id, unsigned slice)
{
	const unsigned lane = threadIdx.x;
	WSYNC();
	park_move(reinterpret_cast<unsigned *>(image), reinterpret_cast<const unsigned *>(&t), sizeof(Txn) / 4, lane);
	park_move(reinterpret_cast<unsigned *>(image + PARK_OFF_W), reinterpret_cast<const unsigned *>(&w), sizeof(BulgeWork) / 4, lane);
	park_move(reinterpret_cast<unsigned *>(image + PARK_OFF_AB), reinterpret_cast<const unsigned *>(&absh), sizeof(ABShared) / 4, lane);
	park_move(reinterpret_cast<unsigned *>(image + PARK_OFF_FAST), reinterpret_cast<const unsigned *>(fast), fast_bytes / 4, lane);
	if (lane == 0) {
		unsigned *lay = reinterpret_cast<unsigned *>(image + PARK_OFF_LAYOUT);
		lay[0] = lds_addr(&t); lay[1] = lds_addr(&w); lay[2] = lds_addr(&absh); lay[3] = lds_addr(fast);
		g.park_of[id] = (slice + 1u) | (bt_round_tag(g) << 20); g.slice_busy[slice] = 1; g.need[id] = 2; atomicAdd(&g.ctr[CTR_PARKED], 1u);
#ifdef SBL_DBG_IDRET
		if (id < (1u << 20)) atomicCAS(&g_dbg_park[id], 0u, ((SS_ROUND_MAX - (g.round_bits >> 20)) << 8) | (w.ret & 255u));      // first park: round, ret
#endif
	}
}
// ... and back; what belongs to the round (the graph view with its round stamp, the claim stamp) is renewed
__device__ __forceinline__ bool park_load(const GraphView &g, Txn &t, BulgeWork &w, ABShared &absh, uint8_t *fast, unsigned fast_bytes, const uint8_t *image, unsigned id, unsigned wi, int prof)
{
	const unsigned lane = threadIdx.x;
	{
		const unsigned *lay = reinterpret_cast<const unsigned *>(image + PARK_OFF_LAYOUT);
		const bool same = lay[0] == lds_addr(&t) && lay[1] == lds_addr(&w) && lay[2] == lds_addr(&absh) && lay[3] == lds_addr(fast);
		if (!same) { if (lane == 0) atomicOr(&g.ctr[CTR_ERR], BT_ERR_LAYOUT); return false; }      // (uniform: one wave, the same four words)
	}
	park_move(reinterpret_cast<unsigned *>(&t), reinterpret_cast<const unsigned *>(image), sizeof(Txn) / 4, lane);
	park_move(reinterpret_cast<unsigned *>(&w), reinterpret_cast<const unsigned *>(image + PARK_OFF_W), sizeof(BulgeWork) / 4, lane);
	park_move(reinterpret_cast<unsigned *>(&absh), reinterpret_cast<const unsigned *>(image + PARK_OFF_AB), sizeof(ABShared) / 4, lane);
	park_move(reinterpret_cast<unsigned *>(fast), reinterpret_cast<const unsigned *>(image + PARK_OFF_FAST), fast_bytes / 4, lane);
	WSYNC();
	if (lane == 0) { t.g = g; t.stamp = g.round_bits | wi; t.prof = prof != 0; w.ret0 = w.ret - 1; atomicSub(&g.ctr[CTR_PARKED], 1u); }      // (ret0: the collapse it parked with belongs to this launch)
	WSYNC();
	return true;
}
// The transaction proper (RemoveBulges for one id) on one wave; t, w, flag, absh and fast live in LDS.
// solo: 0 = ordered round (the probe found bulges, the entry owns its claims), 1 = the id runs with nothing else in flight
// (big-arena solo round, or the serial chain: stampv == BT_NONE, no reservation exists and none is checked).
template <bool RESUME = false>
__device__ __forceinline__ void commit_body(const GraphView &g, Txn &t, BulgeWork &w, int &flag, ABShared &absh, uint8_t *fast, unsigned fast_bytes,
                                            unsigned wi, unsigned id, unsigned stampv, int solo, bool prepass, uint8_t *mine, unsigned arena_bytes, int prof,
                                            const unsigned *sepl = nullptr /* LDS copy of the separators' slots (SepBounds), or none */,
                                            unsigned park_slice = BT_NONE /* may park (GraphView::park_of): its arena slice; RESUME: it is parked there */)
{
	const unsigned lane = threadIdx.x, tid = id + 1;
	PH_T0();
	// parking: the last PARK_IMG bytes of the slice are kept for the LDS image (t, w, absh, fast)
	const bool can_park = park_slice != BT_NONE && g.park_cap != 0 && arena_bytes >= 4u * PARK_IMG;
	if (can_park) arena_bytes -= PARK_IMG;
	// ---- the probe of this round found bulges (solo entries were not probed: verdict pass first)
	if (lane == 0) { g.need[id] = 0; g.touch[id] = 1; flag = 1; }
	if (RESUME) {                                                     // the state it parked with; what belongs to the round is renewed
		if (!park_load(g, t, w, absh, fast, fast_bytes, mine + arena_bytes, id, wi, prof)) return;      // (an image of another LDS layout: CTR_ERR, the stage ends with an internal error)
		if (lane == 0) flag = 1;
		WSYNC();
	} else {
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
	// (counted at the END of the transaction: every owner of a launch starts at the same moment, and two atomics on two addresses by ~3 000
	// waves at once stood in front of the first barrier of each of them -- the wait for the acknowledgement, not the atomic, is what costs)
	if (!flag) { if (lane == 0) { atomicAdd(&g.ctr[CTR_COMMITTED], 1u); atomicAdd(&g.ctr[CTR_TXN], 1u); } return; }
	// ---- writer pass: reads and writes are published for order validation
	if (lane == 0) { t.init(g, id, wi, 2, mine, arena_bytes); t.chain = stampv == BT_NONE; t.defer_push = true; t.ext_stamps = true; t.fscr = fast; t.fscr_cap = fast_bytes; w.ret = 0;
	                 t.tc_cap = 1024; t.tc_list = (uint32_t *)t.alloc(t.tc_cap * 4); if (!t.tc_list) t.tc_cap = 0; t.err = 0; t.defer_cleanup = true; t.prof = prof != 0; }
	WSYNC();
	PH_ADD(7);
	{	// (wave_setup, with a timer between its two halves)
		const unsigned h0 = g.head[0][t.id], h1 = g.head[1][t.id];          // in flight while lane 0 lays the scratch out
		if (lane == 0) flag = bt_setup(t, w, false, false) && !t.err ? 1 : 0;
		WSYNC();
		PH_ADD(19);
		if (flag) {
			const unsigned m = wave_list_positions(g, h0, h1, w, lane);
			if (m != w.n && lane == 0) {
				if (t.g.any_parked && m < w.n) { w.n = m; if (m < 2) flag = 0; }      // (nodes erased by a parked transaction: see wave_setup)
				else { t.err |= BT_ERR_SCRATCH; flag = 0; }                   // cannot happen on a consistent graph
			}
			WSYNC();
		}
	}
	}
	PH_ADD(0);
	if (flag) {
		if (!RESUME) {
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
			                 w.use_stale = !w.lazy && w.n <= 256u && g.lazy_rescan && w.wdel != nullptr; w.wfill = !(g.test_flags & 64u); w.pscan = false; }      // (pscan: measured -1.2 % at 62 strains, +1.4 % of k_commit at 8 -- the handler's registers; the one-launch kernel uses it)      // (many strains: groups of dozens of members, the J search with 64 lanes -- wave_next_j)
		WSYNC();
		PH_ADD(2);
		}
		bool decided = RESUME;                                        // (a parked transaction stopped with its next collapse decided)
		while (flag) {
			if (!RESUME || !decided) {
			if (lane == 0) { const int r = bt_scratch_in_lds(w) ? bt_rb_run<true>(t, w) : bt_rb_run<false>(t, w); flag = t.err ? 0 : r; }      // (<true>: DS instead of FLAT accesses, bulge_txn.h: BT_ASSUME_LDS)
			WSYNC();
			PH_ADD(3);
			if (!flag) break;
			if (flag == 3) { wave_next_j(g, w, lane); PH_ADD(24); continue; }       // large group: the search for the next J, 256 members per step
			if (flag == 4) { wave_mults(g, t, w, lane); PH_ADD(25); continue; }     // branches with many bifurcations inside: their multiplicities, one look-up per lane
			if (flag == 5) { wave_fill_visit(t, w, g.D, lane); PH_ADD(26); if (t.err) break; continue; }      // FillVisit(I), one (id, distance) pair per lane
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
			}
			decided = false;
			// ---- a collapse has been decided (c_src -> c_tgt).  Enough of them for one launch: park (GraphView::park_of)
			if (can_park && !t.g.park_hold && !w.lazy && w.ret - w.ret0 > g.park_cap) {      // (ret counts the collapse just decided; park_hold: the driver waits for what is parked to drain -- read from the LDS copy of the view: one more live kernel argument tipped k_commit's scratch from 232 to 584 B)
				park_store(g, t, w, absh, fast, fast_bytes, mine + arena_bytes, id, park_slice);      // (arena_bytes: already without the image)
				return;                                                    // (no Cleanup, no counters: the transaction is not over)
			}
#ifdef SBL_DBG_IDRET
			if (lane == 0) {
				const unsigned o = atomicAdd(&g_dbg_nlog, 1u);
				if (o < (1u << 20)) { unsigned *r = g_dbg_log + (size_t)o * 8u; r[0] = id; r[1] = SS_ROUND_MAX - (g.round_bits >> 20); r[2] = w.ret; r[3] = (g.nslot[w.start[w.c_src] >> 1] << 1) | (w.start[w.c_src] & 1u); r[4] = (g.nslot[w.start[w.c_tgt] >> 1] << 1) | (w.start[w.c_tgt] & 1u); r[5] = w.c_dS; r[6] = w.c_dT; r[7] = RESUME ? 1u : 0u; }
			}
#endif
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
		atomicAdd(&g.ctr[CTR_COMMITTED], 1u); atomicAdd(&g.ctr[CTR_TXN], 1u);
		if (RESUME) { __threadfence(); g.slice_busy[park_slice] = 0; g.park_of[id] = 0x80000000u | (bt_round_tag(g) << 20); }      // (everything this transaction wrote into the slice is out before somebody else takes it)
		if (t.err) {
			if (!t.wrote && t.err == BT_ERR_SCRATCH) { ss_mark_big(g, id); return; }
			atomicOr(&g.ctr[CTR_ERR], t.err);
		}
		atomicAdd(&g.ctr[CTR_BULGES], w.ret);
#ifdef SBL_DBG_IDRET
		if (id < (1u << 20)) { atomicAdd(&g_dbg_ret[id], w.ret); atomicCAS(&g_dbg_fin[id], 0u, ((SS_ROUND_MAX - (g.round_bits >> 20)) << 8) | (RESUME ? 128u : 0u) | (w.ret & 127u)); }      // first finish: round, resumed, collapses
#endif
	}
}

template <bool RESUME_KERNEL>
__device__ __forceinline__ void commit_kernel(const GraphView &g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, int solo, const unsigned *claims, const uint8_t *live, int prof_every)
{
	const int prof = prof_every && blockIdx.x % (unsigned)prof_every == 0u;      // SBL_PHASES=N: every Nth entry is timed (all of them distort what they measure)
	__shared__ Txn t;
	__shared__ BulgeWork w;
	__shared__ int flag;
	__shared__ ABShared absh;
	__shared__ __attribute__((aligned(16))) uint8_t fast[COMMIT_FAST_BYTES];     // window summaries, mark lists, FillVisit list and AnyBulges map of typical ids
	unsigned wi = blockIdx.x;
	const unsigned lane = threadIdx.x;
	if (!RESUME_KERNEL) round_stamp(g, 2);
	if (RESUME_KERNEL) {                                              // one workgroup per parked entry of the window (GraphView::park_list, filled by k_reserve)
		if (wi >= __builtin_amdgcn_readfirstlane((int)g.ctr[CTR_PLIST])) return;
		wi = (unsigned)__builtin_amdgcn_readfirstlane((int)g.park_list[wi]);
	}
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
	// parked here in an earlier round (GraphView::park_of): resume in the slice it parked in; a fresh entry whose own slice holds somebody's
	// parked state waits a round
	// parked here in an earlier round: k_resume's (same launch configuration, behind this kernel); a fresh entry whose own slice holds
	// somebody's parked state waits a round
	if (RESUME_KERNEL) {
		const unsigned pk = (unsigned)__builtin_amdgcn_readfirstlane((int)g.park_of[id]);
		if (!pk || (pk >> 31) || ((pk >> 20) & 0x7FFu) == bt_round_tag(g)) return;      // not parked / parked in this very launch
		const unsigned slice = (pk & 0xFFFFFu) - 1u;
		commit_body<true>(g, t, w, flag, absh, fast, (unsigned)sizeof fast, wi, id, stampv, 0, false, arena + (size_t)slice * arena_bytes, arena_bytes, prof, sepl, slice);
	} else {
		if (!solo && g.park_cap) {
			const unsigned pk = (unsigned)__builtin_amdgcn_readfirstlane((int)g.park_of[id]);
			if (pk && (!(pk >> 31) || ((pk >> 20) & 0x7FFu) == bt_round_tag(g))) return;      // parked (k_resume's), or finished in this round
		}
		const bool shadow = !solo && g.park_cap && __builtin_amdgcn_readfirstlane((int)g.slice_busy[g.shadow_base + 64u + wi]) != 0;      // my slice held a parked transaction when the round's commits started (k_reserve's copy: k_resume may be releasing it right now): the spare one
		commit_body<false>(g, t, w, flag, absh, fast, (unsigned)sizeof fast, wi, id, stampv, solo, solo != 0, arena + (size_t)(shadow ? g.shadow_base + wi : wi) * arena_bytes, arena_bytes, prof, sepl,
		                   !solo && g.park_cap && !shadow ? wi : (unsigned)BT_NONE);
	}
}
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_commit(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, int solo, const unsigned *claims, const uint8_t *live, int prof_every)
{
	commit_kernel<false>(g, nwin, arena, arena_bytes, solo, claims, live, prof_every);
}

// the parked transactions of the window that own their claims this round (GraphView::park_of): the rest of their loops.  A kernel of its
// own, beside k_commit on a second stream: with both bodies in one kernel every transaction spilled (scratch 232 -> 616 B), and a call
// made the kernel's scratch 880 B (commit 32 -> 52 ms either way)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_resume(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, const unsigned *claims, const uint8_t *live, int prof_every)
{
	commit_kernel<true>(g, nwin, arena, arena_bytes, 0, claims, live, prof_every);
}

// finished markers of park_of (bit 31 | round tag) are only read in the round they were written in; the tag is 11 bits of a 12-bit round,
// so they are swept before a round with the same tag comes round again (DeviceBackend::commit, every 1024 rounds)
__global__ void __launch_bounds__(256) k_park_sweep(unsigned *__restrict__ park_of, unsigned n)
{
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n && (park_of[i] >> 31)) park_of[i] = 0u;
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
		// AnyBulges cannot find anything unless two instances continue with DIFFERENT characters (an entry only gets a second member from
		// an instance whose endChar differs from the entry's, bulgeremoval.cpp:158-218): the endChars alone, k + 1 steps per instance,
		// before the windows (D + k + 2 steps) and the map (every mark of every instance) -- this path examines EVERY id in every
		// iteration, and nearly all of them end here (round 6: a call on 57 strains was 0.5 ms of map building that found nothing).
		unsigned cls = 0;
		for (unsigned i0 = 0; i0 < w.n; i0 += 64) {
			const char ec = i0 + lane < w.n ? lane_end_char(g, w, i0 + lane) : ' ';
			cls |= ec == 'A' ? 1u : ec == 'C' ? 2u : ec == 'G' ? 4u : ec == 'T' ? 8u : 0u;
		}
#pragma unroll
		for (int d = 32; d > 0; d >>= 1) cls |= __shfl_xor(cls, d);
		if (__popc(cls) <= 1) { WSYNC(); if (lane == 0) flag = 0; WSYNC(); }
	}
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
		if (lane == 0) { flag = bt_rb_begin(t, w, any) && !t.err ? 1 : 0; w.lazy = true; w.jscan = true; w.pscan = true; }
		WSYNC();
		while (flag) {
			if (lane == 0) { const int r = bt_rb_run(t, w); flag = t.err ? 0 : r; }
			WSYNC();
			if (!flag) break;
			if (flag == 3) { wave_next_j(g, w, lane); continue; }
			if (flag == 6) { wave_next_pair(g, t, w, lane); continue; }
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


void sbl_commit_prof_reset()
{
	unsigned long long z[64] = {0};
	HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, 32 * 8));
	HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_txn_hist), z, 64 * 8));
	HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_txn_max), z, 16));
	{ std::vector<unsigned long long> zz(4096, 0); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_round_max), zz.data(), 4096 * 8)); }
	{ std::vector<unsigned long long> zz(4096 * 2, 0); HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_round_few), zz.data(), 4096 * 2 * 8)); }
	{ std::vector<unsigned long long> zz(4096 * 3, 0); for (unsigned r = 0; r < 4096; r++) zz[3 * r] = ~0ull; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_round_span), zz.data(), 4096 * 3 * 8)); }
}
void sbl_commit_prof_report(unsigned ts_round)
{
	unsigned long long z[32];
	HIP_TRY(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_phase_cycles), sizeof z));
	const char *nm[32] = {"list-positions", "scan", "rb_begin", "rb_run", "dirty-calc", "collapse", "publish", "init", "rescan",
	                      " c:erase-flanks", " c:erase-span", " c:positions+NE-alloc", " c:replace", " c:copy-marks-data", " c:NN-alloc+stamps", " c:addpoints",
	                      " b:endchars+sizing", " b:map-build", " b:finish", "bt_setup", " r:FillVisit", " r:Overlap", " r:multiplicities", " r:J walk + search",
	                      " w:next J", " w:multiplicities", " w:FillVisit", "(unused)", "(unused)", "(unused)", "(unused)", "(unused)"};
	for (int i = 0; i < 32; i++) fprintf(stderr, "[sbl] commit phase %-12s %10.3f Mcycles\n", nm[i], z[i] / 1e6);
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
		for (unsigned r = 0; r < 4096 && r < ts_round; r++) if (rm[r]) fprintf(stderr, " %llu/%llu/%llu/%llu", (rm[r] >> 24) / 1000, (rm[r] >> 16) & 255, (rm[r] >> 8) & 255, rm[r] & 255);
		fprintf(stderr, "\n");
	}
}
