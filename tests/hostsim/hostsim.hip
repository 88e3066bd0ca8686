// hostsim.hip -- TEST INFRASTRUCTURE: runs the product's bulge-removal transaction code
// (sibelia_amd/csrc/bulge_txn.h, simplify_steps.h, simplify_driver.h) on the HOST, one "thread" at a
// time, so that the ordered-commit scheme can be checked against the oracle without a GPU
// (tests/test_hostsim.py, CPU suite).  It is never linked into libsibelia_amd.so and is not a
// fallback: the shipped library calls the same __host__ __device__ functions from kernels only.
//
// The window entries of a round are executed in ascending, descending or shuffled order to show
// that the committed result does not depend on the execution order inside a round.
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>

#include "../../sibelia_amd/csrc/sbl_common.h"
#include "../../sibelia_amd/csrc/simplify_driver.h"

namespace {

struct HostBackend {
	GraphView g{};
	std::vector<uint8_t> ch, ndead, need, big, touch;
	std::vector<uint32_t> op, nx, pv, bif[2], nodeof[2], nslot, nnext, nidst, nclr, head[2], lsize[2], ctr, own, lock, rmax, wmax, win;
	uint32_t nid_ = 0;
	int order_mode = 0;
	uint32_t arena_bytes = 1u << 16, big_arena_bytes = 1u << 26;
	std::vector<uint8_t> arena, big_arena;
	// block index (GraphView::bidx): maintained by the transaction code, rebuilt here from the arrays for comparison
	std::vector<unsigned long long> bidx;
	uint64_t idx_probe_checked = 0, idx_probe_served = 0, idx_fp_checked = 0, idx_fp_served = 0;
	void idx_build(std::vector<unsigned long long> &out, bool stamps) const
	{
		const uint32_t norig = g.norig, nb = (norig + 63) / 64;
		out.assign((size_t)nb * BT_IDX_WORDS, 0ull);
		for (uint32_t e = 0; e < norig; e++) {
			unsigned long long *w = &out[(size_t)(e >> 6) * BT_IDX_WORDS];
			const unsigned long long bit = 1ull << (e & 63u);
			if (bif[0][e] != BT_NONE) w[0] |= bit;
			if (bif[1][e] != BT_NONE) w[1] |= bit;
			if (ch[e] == BT_SEP) w[2] |= bit;
			const bool bad = ch[e] == BT_DEAD_CHAR || (e + 1 < norig && nx[e] != e + 1) || (e > 0 && pv[e] != e - 1);
			if (bad) w[3] |= 1ull << 32;
			if (stamps && wmax[e] > (uint32_t)w[3]) w[3] = (w[3] & ~0xFFFFFFFFull) | wmax[e];
		}
	}
	// the maintained index against a rebuild: marks and separators exactly, "not pristine" and the stamps at least what a rebuild finds
	void idx_check(const char *where) const
	{
		if (!g.bidx) return;
		std::vector<unsigned long long> want;
		idx_build(want, true);
		for (size_t b = 0; b * BT_IDX_WORDS < want.size(); b++) {
			const unsigned long long *h = &bidx[b * BT_IDX_WORDS], *w = &want[b * BT_IDX_WORDS];
			const bool ok = h[0] == w[0] && h[1] == w[1] && h[2] == w[2] && (!(w[3] >> 32) || (h[3] >> 32)) && (uint32_t)h[3] >= (uint32_t)w[3];
			if (!ok) {
				char msg[256];
				snprintf(msg, sizeof msg, "block index out of date (%s): block %zu has %llx %llx %llx %llx, a rebuild gives %llx %llx %llx %llx", where, b, h[0], h[1], h[2], h[3], w[0], w[1], w[2], w[3]);
				throw SblError{SBL_ERR_INTERNAL, msg};
			}
		}
	}
	uint64_t rng = 88172645463325252ull;
	// checkpoint
	struct Ck { std::vector<uint8_t> ch, ndead, touch; std::vector<uint32_t> op, nx, pv, bif[2], nodeof[2], nslot, nnext, head[2], lsize[2]; uint32_t ne, nn; } ck;

	void bind()
	{
		g.ch = ch.data(); g.op = op.data(); g.nx = nx.data(); g.pv = pv.data();
		for (int s = 0; s < 2; s++) { g.bif[s] = bif[s].data(); g.nodeof[s] = nodeof[s].data(); g.head[s] = head[s].data(); g.lsize[s] = lsize[s].data(); }
		g.nslot = nslot.data(); g.nnext = nnext.data(); g.nidst = nidst.data(); g.nclr = nclr.data(); g.ndead = ndead.data();
		g.ctr = ctr.data(); g.need = need.data(); g.big = big.data(); g.touch = touch.data();
		g.own = own.data(); g.lock = lock.data(); g.rmax = rmax.data(); g.wmax = wmax.data();
		g.cap_e = (uint32_t)ch.size(); g.cap_n = (uint32_t)nslot.size();
		g.nblk = (g.cap_e >> BT_BLOCK_SHIFT) + 1;
		g.win = win.data();
		g.bidx = bidx.empty() ? nullptr : bidx.data();
	}
	uint32_t nid() { return nid_; }
	void checkpoint()
	{
		ck.ne = ctr[CTR_NE]; ck.nn = ctr[CTR_NN];
		ck.ch = ch; ck.ndead = ndead; ck.touch = touch; ck.op = op; ck.nx = nx; ck.pv = pv; ck.nslot = nslot; ck.nnext = nnext;
		for (int s = 0; s < 2; s++) { ck.bif[s] = bif[s]; ck.nodeof[s] = nodeof[s]; ck.head[s] = head[s]; ck.lsize[s] = lsize[s]; }
	}
	void restore()
	{
		auto cp = [](auto &dst, const auto &src) { std::copy(src.begin(), src.end(), dst.begin()); };
		cp(ch, ck.ch); cp(ndead, ck.ndead); cp(touch, ck.touch); cp(op, ck.op); cp(nx, ck.nx); cp(pv, ck.pv); cp(nslot, ck.nslot); cp(nnext, ck.nnext);
		for (int s = 0; s < 2; s++) { cp(bif[s], ck.bif[s]); cp(nodeof[s], ck.nodeof[s]); cp(head[s], ck.head[s]); cp(lsize[s], ck.lsize[s]); }
		ctr[CTR_NE] = ck.ne; ctr[CTR_NN] = ck.nn;
		if (g.bidx) idx_build(bidx, false);
	}
	void snapshot_all(bool incremental)
	{
		for (uint32_t id = 0; id < nid_; id++) ss_snapshot(g, id, arena.data(), 1u << 14, incremental);
	}
	void reset_round_state(bool stamps_too)
	{
		std::fill(own.begin(), own.end(), 0xFFFFFFFFu);
		std::fill(lock.begin(), lock.end(), 0xFFFFFFFFu);
		if (stamps_too) {
			std::fill(rmax.begin(), rmax.end(), 0u); std::fill(wmax.begin(), wmax.end(), 0u);
			for (size_t b = 3; b < bidx.size(); b += BT_IDX_WORDS) bidx[b] &= ~0xFFFFFFFFull;
		}
	}
	void clear_counters()
	{
		ctr[CTR_ERR] = 0; ctr[CTR_BULGES] = 0; ctr[CTR_VIOL] = BT_NONE; ctr[CTR_BIG] = 0; ctr[CTR_COMMITTED] = 0; ctr[CTR_TXN] = 0;
	}
	// (the device backend launches the selection behind a round and reads it with the round's counters; here it is evaluated when read)
	uint32_t sel_lo = 0, sel_limit = 0, sel_W = 0;
	void select_launch(uint32_t lo, uint32_t limit, uint32_t W) { sel_lo = lo; sel_limit = limit; sel_W = W; }
	void select_read(uint32_t *nwin, uint32_t *newlo, uint32_t *solo) { select(sel_lo, sel_limit, sel_W, nwin, newlo, solo); }
	void select(uint32_t lo, uint32_t limit, uint32_t W, uint32_t *nwin, uint32_t *newlo, uint32_t *solo)
	{
		uint32_t n = 0;
		*solo = 0; *newlo = lo;
		bool first = true;
		for (uint32_t id = lo; id <= limit && n < W; id++) {
			if (!need[id]) continue;
			if (first) { *newlo = id; first = false; }
			if (big[id]) { if (n == 0) { win[n++] = id; *solo = 1; } break; }
			win[n++] = id;
		}
		*nwin = n;
	}
	std::vector<uint32_t> order(uint32_t n)
	{
		std::vector<uint32_t> o(n);
		for (uint32_t i = 0; i < n; i++) o[i] = order_mode == 1 ? n - 1 - i : i;
		if (order_mode == 2)
			for (uint32_t i = n; i > 1; i--) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; std::swap(o[i - 1], o[rng % i]); }
		return o;
	}
	// like k_reserve / k_commit: the ids claimed at RESERVATION time are remembered and checked at commit time
	std::vector<std::vector<uint32_t>> claims;
	std::vector<uint32_t> dbg_bif0, dbg_bif1; uint32_t dbg_nn = 0;
	alignas(16) uint8_t fastbuf[24576];      // stands in for the kernels' LDS scratch
	std::vector<uint8_t> live;
	bool chain(uint32_t, uint32_t) { return false; }      // the host driver always runs ordered rounds
	void probe(uint32_t nwin, uint32_t round)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		live.assign(nwin, 0);
		for (uint32_t w : order(nwin)) {
			// the verdict the index gives (ss_verdict_idx: what k_probe's probe_idx evaluates) must be the walking probe's wherever it gives one
			const uint32_t id = win[w];
			const uint32_t viol_before = ctr[CTR_VIOL];
			const int vi = ss_verdict_idx(g, id, id + 1);
			uint8_t need_before = need[id];
			uint32_t perr = 0;
			const bool has = ss_probe(g, w, arena.data(), arena_bytes, &perr);
			live[w] = has ? 1 : 0;
			idx_probe_checked++;
			if (vi >= 0 && ctr[CTR_VIOL] == viol_before && !perr) {
				idx_probe_served++;
				const bool few = g.lsize[0][id] + g.lsize[1][id] < 2;      // (bt_setup: fewer than two instances, nothing to do)
				if (!few && (vi == 1) != has) {
					char msg[160];
					snprintf(msg, sizeof msg, "index verdict %d of id %u differs from the walking probe's %d (need was %u)", vi, id, (int)has, need_before);
					throw SblError{SBL_ERR_INTERNAL, msg};
				}
			}
		}
	}
	void mark_live(uint32_t nwin) { live.assign(nwin, 1); }
	void reserve(uint32_t nwin, uint32_t round)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		claims.assign(nwin, {});
		if (getenv("HOSTSIM_DEBUG")) { dbg_bif0 = bif[0]; dbg_bif1 = bif[1]; dbg_nn = ctr[CTR_NN]; }
		for (uint32_t w : order(nwin)) {
			if (!live[w]) continue;
			uint32_t st = g.round_bits | w;
			uint32_t me = win[w];
			std::set<std::pair<uint32_t, uint32_t>> walked, indexed;
			bt_footprint(g, me, [&](uint32_t b, uint32_t kind) {
				walked.insert({b, kind});
				if (kind == 0) { bt_atomic_min(&g.own[b], st); claims[w].push_back(b); }
				else if (b > me) bt_atomic_min(&g.own[b], st);
				else if (b < me) claims[w].push_back(b | 0x80000000u);
			});
			// the footprint the index gives (bt_footprint_idx: what k_reserve's reserve_idx walks) must be the same set of (id, kind) pairs
			idx_fp_checked++;
			if (bt_footprint_idx(g, me, [&](uint32_t b, uint32_t kind) { indexed.insert({b, kind}); })) {
				idx_fp_served++;
				if (indexed != walked) {
					char msg[200];
					snprintf(msg, sizeof msg, "index footprint of id %u differs from the walked one (%zu against %zu pairs)", me, indexed.size(), walked.size());
					throw SblError{SBL_ERR_INTERNAL, msg};
				}
			}
		}
	}
	void commit(uint32_t nwin, uint32_t round, bool solo)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		if (solo) { ss_commit_run(g, 0, big_arena.data(), big_arena_bytes, fastbuf, sizeof fastbuf); return; }
		for (uint32_t w : order(nwin)) {
			if (!live[w]) continue;
			uint32_t st = g.round_bits | w;
			bool owner = true;
			for (uint32_t b : claims[w]) {
				if (b & 0x80000000u) { uint32_t x = b & 0x7FFFFFFFu; if (own[x] != st && bt_order_blocked(g, x)) { owner = false; break; } }
				else if (own[b] != st) { owner = false; break; }
			}
			uint32_t before = ctr[CTR_VIOL];
			if (owner) ss_commit_run(g, w, arena.data(), arena_bytes, fastbuf, sizeof fastbuf);
			if (getenv("HOSTSIM_DEBUG") && ctr[CTR_VIOL] != before) {
				uint32_t a = win[w];
				fprintf(stderr, "[dbg] violation while committing id %u (widx %u)\n", a, w);
				for (uint32_t x = 0; x < nwin; x++) {
					if (x == w) continue;
					bool xo = true; for (uint32_t b : claims[x]) if (!(b & 0x80000000u) && own[b] != (g.round_bits | x)) { xo = false; break; }
					if (!xo) continue;
					uint32_t b = win[x];
					bool near = false;
					for (int s1 = 0; s1 < 2; s1++) for (uint32_t n1 = g.head[s1][a]; n1 != BT_NONE; n1 = g.nnext[n1])
						for (int s2 = 0; s2 < 2; s2++) for (uint32_t n2 = g.head[s2][b]; n2 != BT_NONE; n2 = g.nnext[n2]) {
							long d = (long)g.nslot[n2] - (long)g.nslot[n1];
							if (d > -400 && d < 400) near = true;
						}
					if (!near) continue;
					bool ainx = false, xina = false;
					for (uint32_t q : claims[x]) if (q == a) ainx = true;
					for (uint32_t q : claims[w]) if (q == b) xina = true;
					fprintf(stderr, "  co-winner id %u (widx %u) a-in-its-claims %d it-in-a-claims %d nclaims %zu/%zu\n", b, x, ainx, xina, claims[x].size(), claims[w].size());
					for (uint32_t n1 = g.head[1][a]; n1 != BT_NONE; n1 = g.nnext[n1]) {
						if (g.ndead[n1]) continue;
						uint32_t e = g.nslot[n1];
						for (uint32_t i = 0; i <= 2 * (g.D + g.k) + g.k; i++) {
							if (i && g.ch[e] == BT_SEP) { fprintf(stderr, "     walk from %u hit SEP at step %u slot %u\n", g.nslot[n1], i, e); break; }
							if (g.bif[0][e] == b || g.bif[1][e] == b) fprintf(stderr, "     walk from %u sees id %u at step %u slot %u\n", g.nslot[n1], b, i, e);
							e = g.pv[e];
							if (e == BT_NONE) break;
						}
					}
					for (int which = 0; which < 2; which++) {
						uint32_t id = which ? b : a;
						for (int s1 = 0; s1 < 2; s1++) for (uint32_t n1 = g.head[s1][id]; n1 != BT_NONE; n1 = g.nnext[n1])
							fprintf(stderr, "     id %u node %u%s strand %d slot %u dead %d  bif-at-slot now %d/%d at-reserve %d/%d\n", id, n1, n1 >= dbg_nn ? "(NEW this round)" : "", s1, g.nslot[n1], g.ndead[n1],
							        (int)g.bif[0][g.nslot[n1]], (int)g.bif[1][g.nslot[n1]], (int)dbg_bif0[g.nslot[n1]], (int)dbg_bif1[g.nslot[n1]]);
					}
				}
			}
		}
	}
	SimplifyCounters counters() { SimplifyCounters c; memcpy(c.v, ctr.data(), sizeof c.v); return c; }
	bool grow(uint32_t err)
	{
		if (err & BT_ERR_ELEM_CAP) {
			size_t n = ch.size() * 2;
			ch.resize(n); op.resize(n); nx.resize(n); pv.resize(n);
			for (int s = 0; s < 2; s++) { bif[s].resize(n, BT_NONE); nodeof[s].resize(n); }
			lock.assign((n >> BT_BLOCK_SHIFT) + 1 + nid_ + 1, 0xFFFFFFFFu); rmax.assign(lock.size(), 0); wmax.assign(lock.size(), 0);
		}
		if (err & BT_ERR_NODE_CAP) { size_t n = nslot.size() * 2; nslot.resize(n); nnext.resize(n); nidst.resize(n); nclr.resize(n); ndead.resize(n); }
		if (err & ~(uint32_t)(BT_ERR_ELEM_CAP | BT_ERR_NODE_CAP)) return false;
		bind();
		return true;
	}
};

}  // namespace

// Input: sanitised sequences + original positions, the enumeration (instances sorted by (chr,pos) per strand,
// negative strand in reverse-complement coordinates).  Output: post-stage sequences / positions.
// stats: [0]=iterations [1]=rounds [2]=replays [3]=solo rounds [4]=executed transactions
extern "C" int hostsim_stage(uint32_t nchr, const uint8_t *const *seq, const uint32_t *const *opos, const uint64_t *len,
                             uint32_t k, uint32_t D, uint32_t max_iter, uint32_t bif_count,
                             const uint32_t *pos_inst, uint64_t n0, const uint32_t *neg_inst, uint64_t n1,
                             uint32_t window, int order_mode, uint32_t arena_bytes, uint32_t slack_elems,
                             uint8_t **out_seq, uint32_t **out_op, uint64_t *out_len, uint64_t *bulges, uint64_t *stats)
{
	try {
		HostBackend be;
		size_t L = 0;
		for (uint32_t c = 0; c < nchr; c++) L += len[c];
		size_t E = L + nchr + 1, ne0 = (E + 31) / 32 * 32, cap = ne0 + slack_elems;
		be.nid_ = bif_count; be.order_mode = order_mode; be.arena_bytes = arena_bytes;
		be.ch.assign(cap, BT_DEAD_CHAR); be.op.assign(cap, 0); be.nx.assign(cap, BT_NONE); be.pv.assign(cap, BT_NONE);
		for (int s = 0; s < 2; s++) {
			be.bif[s].assign(cap, BT_NONE); be.nodeof[s].assign(cap, BT_NONE);
			be.head[s].assign((size_t)bif_count + 1, BT_NONE); be.lsize[s].assign((size_t)bif_count + 1, 0);
		}
		size_t ncap = n0 + n1 + 1024 + (size_t)slack_elems * 4;   // (the product sizes its node pool 4 x instances + 1 M)
		be.nslot.assign(ncap, 0); be.nidst.assign(ncap, 0); be.nnext.assign(ncap, BT_NONE); be.nclr.assign(ncap, BT_NONE); be.ndead.assign(ncap, 0);
		be.ctr.assign(CTR_COUNT, 0); be.need.assign((size_t)bif_count + 1, 0); be.big.assign((size_t)bif_count + 1, 0); be.touch.assign((size_t)bif_count + 1, 0);
		be.own.assign((size_t)bif_count + 1, 0xFFFFFFFFu);
		be.lock.assign((cap >> BT_BLOCK_SHIFT) + 1 + bif_count + 1, 0xFFFFFFFFu);
		be.rmax.assign(be.lock.size(), 0); be.wmax.assign(be.lock.size(), 0);
		be.win.assign(window ? window : 1, 0);
		be.arena.assign(std::max<uint32_t>(arena_bytes, 1u << 14), 0); be.big_arena.assign(be.big_arena_bytes, 0);
		std::vector<uint32_t> sep(nchr + 1);
		size_t e = 0;
		be.ch[e] = BT_SEP; sep[0] = 0; e++;
		for (uint32_t c = 0; c < nchr; c++) {
			for (uint64_t j = 0; j < len[c]; j++, e++) { be.ch[e] = seq[c][j]; be.op[e] = opos[c][j] & BT_POS_MASK; }
			be.ch[e] = BT_SEP; be.op[e] = (uint32_t)len[c] & BT_POS_MASK; sep[c + 1] = (uint32_t)e; e++;
		}
		for (size_t i = 0; i < E; i++) { be.nx[i] = i + 1 < E ? (uint32_t)(i + 1) : BT_NONE; be.pv[i] = i ? (uint32_t)(i - 1) : BT_NONE; }
		be.ctr[CTR_NE] = (uint32_t)ne0;
		be.bind();
		be.g.k = k; be.g.D = D; be.g.nid = bif_count;
		if (const char *e = getenv("HOSTSIM_LAZY_MIN")) be.g.lazy_min = (uint32_t)atoi(e);      // 1: lazy windows for every transaction
		if (getenv("HOSTSIM_LAZY_MAP")) be.g.test_lazy_map = 1;                                 // AnyBulges logs its insertions, Boost map only for >= 2 groups
		// marking loop (reference src/indexedsequence.cpp:49-67): (chr,pos) ascending, front insertion
		uint32_t nn = 0;
		for (int s = 0; s < 2; s++) {
			const uint32_t *inst = s ? neg_inst : pos_inst;
			uint64_t n = s ? n1 : n0;
			for (uint64_t i = 0; i < n; i++) {
				uint32_t id = inst[3 * i], c = inst[3 * i + 1], p = inst[3 * i + 2];
				uint32_t el = s == 0 ? sep[c] + 1 + p : sep[c + 1] - 1 - p;
				uint32_t nd = nn++;
				be.nslot[nd] = el; be.nidst[nd] = (id << 1) | (uint32_t)s; be.nnext[nd] = be.head[s][id]; be.head[s][id] = nd; be.lsize[s][id]++;
				be.bif[s][el] = id; be.nodeof[s][el] = nd;
			}
		}
		be.ctr[CTR_NN] = nn;
		if (!getenv("HOSTSIM_NO_INDEX")) {
			be.g.norig = (uint32_t)E;
			be.idx_build(be.bidx, false);
			be.bind();
		}
		SimplifyReport rep = simplify_graph(be, max_iter, window, nullptr, nullptr);
		be.idx_check("end of the stage");
		if (getenv("HOSTSIM_INDEX_STATS")) fprintf(stderr, "[hostsim] index served %llu of %llu probes, %llu of %llu footprints\n", (unsigned long long)be.idx_probe_served,
		                                           (unsigned long long)be.idx_probe_checked, (unsigned long long)be.idx_fp_served, (unsigned long long)be.idx_fp_checked);
		for (uint32_t c = 0; c < nchr; c++) {
			std::vector<uint8_t> s; std::vector<uint32_t> p;
			for (uint32_t x = be.nx[sep[c]]; x != sep[c + 1]; x = be.nx[x]) { s.push_back(be.ch[x]); p.push_back(be.op[x] & BT_POS_MASK); }
			out_len[c] = s.size();
			out_seq[c] = (uint8_t *)malloc(s.size() + 1); out_op[c] = (uint32_t *)malloc(s.size() * 4 + 4);
			memcpy(out_seq[c], s.data(), s.size()); memcpy(out_op[c], p.data(), p.size() * 4);
		}
		*bulges = rep.bulges;
		stats[0] = rep.iterations; stats[1] = rep.rounds; stats[2] = rep.replays; stats[3] = rep.solo; stats[4] = rep.executed; stats[5] = rep.grow_replays;
		return 0;
	} catch (const SblError &err) {
		fprintf(stderr, "hostsim: %s\n", err.msg.c_str());
		return (int)err.st;
	}
}
extern "C" void hostsim_free(void *p) { free(p); }
