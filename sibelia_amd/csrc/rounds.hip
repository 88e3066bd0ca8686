// rounds.hip -- the read-only kernels of an ordered round: probe (from the block index: k_probe_idx; walking: k_probe), selection of the
// next window, reservation (k_reserve).  simplify_steps.h explains the round protocol, DESIGN.md 4.1 why it is exact.
#include <cstring>
#include <algorithm>
#include <vector>
#include "simplify_walks.h"

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
	// A chunk whose marks COULD pass VT_FILL (dozens of instances x dozens of marks: 16 windows of a 62-strain input carry > 384 marks between
	// them, nearly all of them the same few ids) is inserted under the exact bound instead: the distinct ids so far + the marks of the next
	// batch of inserts may not pass VT_HARD, 7/8 of the slots (a full table would never end vt_insert's probing).
	const unsigned VT_HARD = 7u << (vt.bits - 3u);
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
			const bool tight = distinct + total > VT_FILL;                   // the table could fill up: count batch by batch
			const unsigned *__restrict__ marks = g.bif[dir];
			while (__any(cm != 0ull)) {
				if (tight) {
					unsigned nins = (unsigned)__popcll(cm);
					nins = nins < 4u ? nins : 4u;
					for (int d = 32; d > 0; d >>= 1) nins += __shfl_xor(nins, d);
					if (distinct + nins > VT_HARD) return -1;
				}
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
				bool found = false;
				for (unsigned m0 = 0; m0 < ww.nm; m0 += 64) {
					if (distinct + (ww.nm - m0 < 64u ? ww.nm - m0 : 64u) > VT_HARD) return -1;
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
	// found live by an earlier probe and not touched since (a push resets it to 1) -- or parked (GraphView::park_of: need = 2 as well; a
	// push to a parked id comes from a lower transaction that rewrote what it had read, which is an order violation and ends the attempt).
	// A parked id still hands its instances to the reservation: it reserves every round until it is through.
	const bool known_live = !snapshot && g.need[id] == 2;
	if (known_live && !(instbuf && (bt_parked(g, id) || (g.test_flags & 16384u)))) { if (lane == 0) { live[wi] = 1; if (instbuf) instbuf[(size_t)wi * istride] = BT_NONE; if (g.test_flags & 32u) atomicAdd(&g_idx_stats[0], 1u); } return; }          // found live by an earlier probe and not touched since (a push resets it to 1)
	const unsigned n = wave_list_nodes(g, g.head[0][id], g.head[1][id], lane, nullptr, [&](unsigned off, unsigned, unsigned s, unsigned el, unsigned) {
		if (off < max_inst) { s_sel[off] = el; s_dir[off] = (uint8_t)s; }
	});
	int r = 0;
	if (known_live) {                                                    // no verdict: the instances only
		if (lane == 0) live[wi] = 1;
		unsigned *ib = instbuf + (size_t)wi * istride;
		const bool give = n <= max_inst && n + 1u <= istride;
		if (give) { WSYNC(); for (unsigned i = lane; i < n; i += 64) ib[1 + i] = (s_sel[i] << 1) | s_dir[i]; }
		if (lane == 0) ib[0] = give ? n : BT_NONE;
		return;
	}
	if (n >= 2) {
		if (n > max_inst || (!g.any_parked && n != g.lsize[0][id] + g.lsize[1][id])) r = -1;      // (with parked transactions about, the ids they erased marks of keep their dead nodes until those are through)      // (lists are clean between rounds: live nodes = list sizes; anything else is the walking path's to report)
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
__global__ void __launch_bounds__(64 * PROBE_WAVES) k_probe(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, uint8_t *live, unsigned w0, int snapshot)
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
	if (threadIdx.x == 0) { t.init(g, id, wi, snapshot ? 0u : 3u, arena + (size_t)(!snapshot && g.slice_busy && g.slice_busy[wi] ? g.shadow_base + wi : wi) * arena_bytes, arena_bytes); t.ext_stamps = true; t.fscr = fast; t.fscr_cap = sizeof fast; }      // (a slice that holds a parked transaction: the spare one, GraphView::park_of)      // (snapshot: the stamps are the previous iteration's -- no order check)
	WSYNC();
	wave_setup<true>(g, t, w, true, lane, ok);
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
// Probe: ONE collective per round (round 6; two before: the verdict bytes, then the violation words).  Every rank packs a record of
// `stride` bytes -- the lowest order violation it has seen (4 B) + the verdict bytes of its share of the window -- into slot `rank` of
// robuf (k_pack_probe); the slots are all-gathered; k_apply_probe then does for the entries the OTHER GPUs probed what k_probe did to
// need / touch / live for its own, and takes the lowest violation anybody saw.  Share of rank p: [nwin p / R, nwin (p + 1) / R).
__global__ void __launch_bounds__(256) k_pack_probe(const unsigned *__restrict__ ctr, const uint8_t *__restrict__ live, unsigned w0, unsigned w1, unsigned rank, unsigned stride, uint8_t *__restrict__ robuf)
{
	uint8_t *slot = robuf + (size_t)rank * stride;
	const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i == 0) *reinterpret_cast<unsigned *>(slot) = ctr[CTR_VIOL];
	if (w0 + i < w1) slot[4 + i] = live[w0 + i];
}
__global__ void __launch_bounds__(256) k_apply_probe(GraphView g, unsigned nwin, uint8_t *__restrict__ live, unsigned w0, unsigned w1, const uint8_t *__restrict__ robuf, unsigned stride, unsigned nranks)
{
	const unsigned wi = blockIdx.x * blockDim.x + threadIdx.x;
	if (wi == 0) { unsigned v = BT_NONE; for (unsigned p = 0; p < nranks; p++) { const unsigned t = *reinterpret_cast<const unsigned *>(robuf + (size_t)p * stride); v = t < v ? t : v; } if (v != BT_NONE) atomicMin(&g.ctr[CTR_VIOL], v); }
	if (wi >= nwin || (wi >= w0 && wi < w1)) return;
	unsigned p = (unsigned)(((unsigned long long)wi * nranks) / nwin);       // owner of the entry: the estimate is at most one off
	while ((unsigned)(((unsigned long long)nwin * p) / nranks) > wi) p--;
	while ((unsigned)(((unsigned long long)nwin * (p + 1)) / nranks) <= wi) p++;
	const uint8_t l = robuf[(size_t)p * stride + 4 + (wi - (unsigned)(((unsigned long long)nwin * p) / nranks))];
	live[wi] = l;
	const unsigned id = g.win[wi];
	if (l == 0) { g.need[id] = 0; g.touch[id] = 0; }
	else if (l == 1 && g.need[id] != 2) g.need[id] = 2;
}

// The lowest pending ids in [lo, limit], ascending; a pending "big" id ends the window (and runs alone if it is the lowest).
// out: win[], ctr[CTR_NWIN], ctr[CTR_LO] (lowest pending id), ctr[CTR_PUSHED] (solo flag).
// Two launches over chunks of `chunk` ids (256 threads x chunk/256 flags, 8-byte loads of the need / big bytes):
//   k_select_count  pending ids per chunk, first pending id, first pending big id (atomicMin)
//   k_select_write  every chunk below the window limit places its ids after the chunks ahead of it; the last one finalises
// sel: [0] first pending big id  [1] first pending id  [2] pending ids below the big id  [3] ticket  [8 ...] per-chunk counts
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
	if (s_last && threadIdx.x == 0) g.ctr[CTR_PLIST] = 0;              // (the list of parked window entries is per round: k_reserve appends, k_resume reads)
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
	// the walk ahead and the walk behind are served separately: a block that is no longer pristine on one side sends only that side to the walks
	const bool dirty = inr && (unsigned)(r1.y >> 32) != 0u && t0 <= (int)last && t0 + 63 >= (int)lo;
	bool sa = fresh || (dirty && ahead), sb2 = fresh || (dirty && behind);
#pragma unroll
	for (int d = 1; d < 16; d <<= 1) { sa |= __shfl_xor((int)sa, d) != 0; sb2 |= __shfl_xor((int)sb2, d) != 0; }
	const bool slow = ahead ? sa : sb2;
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
	// bits 0 - 3: the walk AHEAD (core + opposite-strand flank) of instance i0 + x must be walked; bits 4 - 7: the walk BEHIND
	const unsigned long long ba = __ballot(act && sa && j == 0u), bb = __ballot(act && sb2 && j == 0u);
	const unsigned ma = (unsigned)((ba & 1ull) | ((ba >> 15) & 2ull) | ((ba >> 30) & 4ull) | ((ba >> 45) & 8ull));
	const unsigned mb = (unsigned)((bb & 1ull) | ((bb >> 15) & 2ull) | ((bb >> 30) & 4ull) | ((bb >> 45) & 8ull));
	return ma | (mb << 4);
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
__global__ void __launch_bounds__(64 * RSV_WAVES_MAX) k_reserve(GraphView g, unsigned nwin, unsigned *claims, const uint8_t *live, unsigned seen_bits, unsigned list_cap,
                                                                 const unsigned *__restrict__ instbuf, unsigned istride, const uint8_t *arena, unsigned arena_bytes)
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
	if (g.park_list && threadIdx.x == 0) {
		if (bt_parked(g, id)) g.park_list[atomicAdd(&g.ctr[CTR_PLIST], 1u)] = w;      // k_resume's work list: the parked entries of this window
		// slice_busy as of the START of the round's commit launches (second half of the array): whether the entry at position w works in
		// its own slice (and may park) or in the shadow slice must not depend on whether k_resume, running beside k_commit, has already
		// released the slice -- either outcome is exact, but the rounds differ, and the ranks of a job on several GPUs (replicated
		// commits, collectives sized by the window) must stay in step: that race was round 5's crash with a communicator attached
		g.slice_busy[g.shadow_base + 64u + w] = g.slice_busy[w];
	}
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
	// a parked transaction: the ids it has erased instances of keep their list sizes until its Cleanup -- nobody else may read them
	// meanwhile, and no walk finds them (the marks are gone): claimed from the erase chain of the parked image (bulge_txn.h: PARK_IMG)
	if (wv == RSV_WAVES - 1u && arena && bt_parked(g, id)) {
		const unsigned slice = (g.park_of[id] & 0xFFFFFu) - 1u;
		const Txn *pt = reinterpret_cast<const Txn *>(arena + (size_t)slice * arena_bytes + (arena_bytes - PARK_IMG));
		unsigned nd = pt->tc_head;                                   // (uniform: every lane follows the chain, lane x keeps the x-th id)
		while (nd != BT_NONE) {
			unsigned mine = BT_NONE;
			for (unsigned x = 0; x < 64u && nd != BT_NONE; x++) { const unsigned v = g.nidst[nd] >> 1; if (lane == x) mine = v; nd = g.nclr[nd]; }
			wave_claim(g, cl, st, mine, lane);
		}
	}
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
			if (lane < 4u && i0 + lane < ninst) { served[i0 + lane] = (uint8_t)((((slowm | (slowm >> 4)) >> lane) & 1u) ^ 1u); if (g.test_flags & 32u) atomicAdd(&g_idx_stats[6 + (((slowm | (slowm >> 4)) >> lane) & 1u)], 1u); }
			if (rprof && threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(&g_rsv_ticks[2], n_ - rt); rt = n_; }
			for (unsigned q = 0; q < 4u && i0 + q < ninst; q++) {
				const bool wa = (slowm >> q) & 1u, wb = (slowm >> (4u + q)) & 1u;      // which of the instance's two walks the index could not serve
				if (!wa && !wb) continue;
				const unsigned i = i0 + q;
				const unsigned e0 = inst[i] >> 1, s = inst[i] & 1u;
				const SepBounds sp = sep_bounds(g, sepl, e0, lane);
				unsigned nxt = BT_NONE;
				if (wa) {
					if (!burst || !wave_core_claim_burst(g, e0, s, core, lane, cl, st, sp, nxt))
						nxt = wave_walk_claim(g, e0, s, core, lane, 3u, cl, st, sp);
				}
				if (wa && wb && burst && wave_flank_order_burst(g, e0, s, fwd + 1 > core ? nxt : BT_NONE, fwd + 1 > core ? fwd + 1 - core : 0u, back, lane, sp,
				                                                [&](unsigned b) { wave_claim_order(g, cl, st, id, b, lane); })) continue;
				wave_walk_marks2(g, wa && fwd + 1 > core ? nxt : BT_NONE, s, fwd + 1 - core, 1u << (s ^ 1u),
				                 wb ? (s ? g.nx[e0] : g.pv[e0]) : BT_NONE, s ^ 1u, back, 1u << s, lane, order, sp);
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

void sbl_rounds_stats_report()
{
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
