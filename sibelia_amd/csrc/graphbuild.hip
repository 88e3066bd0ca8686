// graphbuild.hip -- the kernels around the ordered rounds that only stream: graph construction of a stage (links, instance lists in the
// reference's initial order, E2), the block index of the original slots, and the copy-back (T3, reference src/blockfinder.cpp:85-95).
#include <cstring>
#include <algorithm>
#include <vector>
#include "simplify_device.h"

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

