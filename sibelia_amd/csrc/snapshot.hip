// snapshot.hip -- AnyBulges verdicts of an iteration's start (reference src/bulgeremoval.cpp:158-218 for every id): the stream over the
// position-ordered marks (iteration 1), its linearised form and the generic window-walking form (later iterations without the block
// index), and the list of touched ids the index-based incremental snapshot probes (rounds.hip: k_probe_idx).
#include <cstring>
#include <algorithm>
#include <vector>
#include "simplify_device.h"

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

