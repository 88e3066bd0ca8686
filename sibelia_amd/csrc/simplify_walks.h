// simplify_walks.h -- the wave-cooperative neighbourhood walks of the ordered rounds: claims of a reservation (k_reserve, rounds.hip) and
// the publish step of a collapse (k_commit, commit.hip) visit the same elements bt_footprint / bt_push_neighbourhood (bulge_txn.h) visit.
#pragma once
#include "simplify_device.h"
#include "simplify_kernels.h"

// ---- wave-cooperative neighbourhood scan -----------------------------------------------------------------
// The list is almost everywhere laid out consecutively (nx[e] == e + 1), so 64 lanes test 64 consecutive
// slots at once, keep the prefix whose links are intact, and only re-anchor at a real link break
// (an insertion or deletion made by an earlier collapse).  Visits exactly the elements bt_footprint visits.

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
							unsigned rm = g.rmax[bt_ridx_elem(c)];
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
				unsigned rm = g.rmax[bt_ridx_elem(c)];
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

