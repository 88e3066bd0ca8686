// synteny.hip -- downstream of the hot path: BlockFinder::GenerateSyntenyBlocks behind the C ABI.
//
// Replaces GenerateSyntenyBlocks / ResolveOverlap / TrimBlocks (reference src/synteny.cpp:229-286, :124-166, :31-122).  The stage
// re-enters the hot path twice: the edge list of a fresh index at k (sbl_list_edges: enumeration + ListEdges kernels) and, for
// every candidate block, a fresh index at trimK over the block's ORIGINAL sequences (a child context on the same GPU: the
// block ranges are gathered device-to-device from the original records kept at load time, sanitised through the parent's
// rand() stream and enumerated by the same kernels).  What stays on the host is the reference's sequential bookkeeping:
// the three std::sort calls whose handling of equal elements the result depends on (common.h:153, synteny.cpp:249,:254 --
// the same libstdc++ calls on the same element order), the occupancy indicators in original coordinates, and the per-block
// search for the outermost shared bifurcations, here on flat arrays (CSR instance lists, per-chromosome occupancy bytes
// with an undo list instead of a std::set of positions).
#include <cstring>
#include <algorithm>
#include <numeric>

#include "sbl_ctx.h"

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

struct GatherDesc { unsigned long long src, dst; unsigned long long len; };
// child element array: '$' b0 '$' b1 '$' ...  (pre-filled with '$'); one workgroup-stride loop per block range
__global__ void __launch_bounds__(256) k_gather_blocks(const uint8_t *__restrict__ orig, const GatherDesc *__restrict__ desc, unsigned nrec, uint8_t *__restrict__ out)
{
	for (unsigned r = blockIdx.y; r < nrec; r += gridDim.y) {
		const GatherDesc d = desc[r];
		for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < d.len; i += (unsigned long long)gridDim.x * blockDim.x)
			out[d.dst + i] = orig[d.src + i];
	}
}

namespace {

struct BEdge {                                   // BlockFinder::Edge (src/blockfinder.h:58-90) with the fields N2 uses
	uint32_t chr, dir, startVertex, endVertex;
	uint64_t origPos, origLen;
	char firstChar;
};

inline bool natural_less(const BEdge &a, const BEdge &b)      // CompareEdgesNaturally: (startVertex, endVertex, firstChar as size_t), src/edge.cpp:26-40
{
	if (a.startVertex != b.startVertex) return a.startVertex < b.startVertex;
	if (a.endVertex != b.endVertex) return a.endVertex < b.endVertex;
	return static_cast<size_t>(a.firstChar) < static_cast<size_t>(b.firstChar);
}

struct Synteny {
	sbl_ctx *c;
	sbl_ctx *child;
	uint32_t trimK, minSize;
	std::vector<std::vector<uint8_t>> occupied;   // overlap[chr][original position] (POS_FREE / POS_OCCUPIED)
	std::vector<std::vector<uint8_t>> local;      // localOverlap of the group being resolved, cleared through `undo`
	std::vector<std::pair<uint32_t, std::pair<uint64_t, uint64_t>>> undo;

	// ResolveOverlap (src/synteny.cpp:124-166): per edge the longest run of original positions that is free both globally and
	// within the group (the first one among equals), kept when it is at least minSize long
	void resolve_overlap(const BEdge *first, const BEdge *last, std::vector<BEdge> &now)
	{
		now.clear();
		for (; first != last; ++first) {
			const uint8_t *occ = occupied[first->chr].data(), *loc = local[first->chr].data();
			const uint64_t begin = first->origPos, end = first->origPos + first->origLen;
			uint64_t bestStart = 0, bestEnd = 0;
			for (uint64_t s = begin; s < end;) {
				uint64_t e = s;
				while (e < end && occ[e] == 0 && loc[e] == 0) e++;
				if (e - s > bestEnd - bestStart) { bestStart = s; bestEnd = e; }
				s = e == s ? e + 1 : e;
			}
			if (bestEnd - bestStart >= minSize) {
				BEdge x = *first;
				x.origPos = bestStart; x.origLen = bestEnd - bestStart;
				now.push_back(x);
				memset(local[first->chr].data() + bestStart, 1, bestEnd - bestStart);
				undo.push_back({first->chr, {bestStart, bestEnd}});
			}
		}
		for (auto &u : undo) memset(local[u.first].data() + u.second.first, 0, u.second.second - u.second.first);
		undo.clear();
	}

	// a fresh index at trimK over the block sequences (IndexedSequence iseq(blockSeq, trimK, ""), synteny.cpp:44): on the GPU
	void child_index(const std::vector<BEdge> &block, uint32_t *bif_count, const sbl_inst **inst, uint64_t *ninst)
	{
		const uint32_t nrec = (uint32_t)block.size();
		uint64_t L = 0;
		for (auto &b : block) L += b.origLen;
		const size_t E = (size_t)L + nrec + 1, Epad = (E + 31) / 32 * 32 + 64;
		hipStream_t s = child->stream;
		child->d_ch.ensure(Epad);
		HIP_TRY(hipMemsetAsync(child->d_ch.p, '$', Epad, s));
		std::vector<GatherDesc> desc(nrec);
		child->sepidx.assign(nrec + 1, 0);
		size_t e = 1;
		uint64_t maxlen = 1;
		for (uint32_t i = 0; i < nrec; i++) {
			child->sepidx[i] = (uint32_t)(e - 1);
			desc[i].src = (unsigned long long)c->orig_sepidx[block[i].chr] + 1 + block[i].origPos;
			desc[i].dst = e; desc[i].len = block[i].origLen;
			maxlen = std::max<uint64_t>(maxlen, block[i].origLen);
			e += block[i].origLen + 1;
		}
		child->sepidx[nrec] = (uint32_t)(e - 1);
		child->nchr = nrec; child->nelem = E;
		child->d_stage.ensure((size_t)nrec * sizeof(GatherDesc) + 64); child->d_sepidx.ensure((size_t)(nrec + 1) * 4);
		HIP_TRY(hipMemcpyAsync(child->d_stage.p, desc.data(), (size_t)nrec * sizeof(GatherDesc), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(child->d_sepidx.p, child->sepidx.data(), (size_t)(nrec + 1) * 4, hipMemcpyHostToDevice, s));
		dim3 grid((unsigned)std::min<uint64_t>((maxlen + 255) / 256, 1024), std::min<unsigned>(nrec, 1024));
		k_gather_blocks<<<grid, 256, 0, s>>>(c->d_orig_ch.as<uint8_t>(), child->d_stage.as<GatherDesc>(), nrec, child->d_ch.as<uint8_t>());
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(s));
		sbl_finish_load(child);                                  // identity positions + list of non-ACGT positions, on the device
		child->rng = c->rng;                                     // the process-global rand() of the reference: one stream for parent and child
		const sbl_inst *pos = nullptr, *neg = nullptr; uint64_t np = 0, nn = 0;
		sbl_status st = sbl_enumerate(child, trimK, bif_count, &pos, &np, &neg, &nn);
		c->rng = child->rng;
		if (st != SBL_OK) throw SblError{st, std::string("TrimBlocks index: ") + child->err};
		inst[0] = pos; inst[1] = neg; ninst[0] = np; ninst[1] = nn;
	}

	// TrimBlocks (src/synteny.cpp:31-122): for every sequence of the block the bifurcations shared with another sequence that lie
	// closest to its two ends; a sequence sharing none makes the caller try again without it
	bool trim_blocks(std::vector<BEdge> &block)
	{
		if (block.empty()) return false;                         // (an index over no sequences: nothing to trim, nothing dropped, no rand() drawn)
		uint32_t bifCount = 0; const sbl_inst *inst[2]; uint64_t ninst[2];
		child_index(block, &bifCount, inst, ninst);
		const size_t nrec = block.size();
		std::vector<uint64_t> len(nrec), base(nrec + 1, 0);
		for (size_t i = 0; i < nrec; i++) { len[i] = block[i].origLen; base[i + 1] = base[i] + len[i]; }
		// mark[strand][base[chr] + element position] = id of the k-mer starting there on that strand (GetBifurcation)
		const uint32_t NONE = 0xFFFFFFFFu;
		std::vector<uint32_t> mark[2];
		for (int s = 0; s < 2; s++) {
			mark[s].assign(base[nrec], NONE);
			for (uint64_t i = 0; i < ninst[s]; i++) {
				const sbl_inst &x = inst[s][i];
				mark[s][base[x.chr] + (s == 0 ? x.pos : len[x.chr] - 1 - x.pos)] = x.id;
			}
		}
		// ListPositions order per id (bifurcationstorage.h:59-72 after the front insertions of indexedsequence.cpp:49-67): + instances in
		// descending (chr, position), then - instances in descending (chr, reverse-complement position); CSR over ids
		std::vector<uint64_t> off((size_t)bifCount + 2, 0);
		for (int s = 0; s < 2; s++) for (uint64_t i = 0; i < ninst[s]; i++) off[inst[s][i].id + 1]++;
		for (size_t i = 1; i < off.size(); i++) off[i] += off[i - 1];
		std::vector<uint64_t> cursor(off.begin(), off.end() - 1);
		std::vector<uint32_t> ichr(off.back()); std::vector<uint64_t> ipos(off.back());
		for (int s = 0; s < 2; s++)
			for (uint64_t i = ninst[s]; i-- > 0;) {
				const sbl_inst &x = inst[s][i];
				const uint64_t at = cursor[x.id]++;
				ichr[at] = x.chr; ipos[at] = s == 0 ? x.pos : len[x.chr] - 1 - x.pos;
			}
		bool drop = false;
		std::vector<BEdge> ret;
		const uint64_t oo = 0xFFFFFFFFull;                       // UINT_MAX, synteny.cpp:41
		auto dist = [](uint64_t a, uint64_t b) { return a > b ? a - b : b - a; };      // StrandIteratorDistance, indexedsequence.cpp:162-167
		for (size_t chr = 0; chr < nrec; chr++) {
			const uint32_t dir = block[chr].dir;
			const uint64_t n = len[chr], beginPos = dir == 0 ? 0 : n - 1, lastPos = dir == 0 ? n - 1 : 0;
			uint64_t trimStart = 0, trimEnd = 0, minBifStart = oo, minBifEnd = oo, minStartSum = oo, minEndSum = oo;
			for (uint64_t t = 0; t < n; t++) {
				const uint64_t itPos = dir == 0 ? t : n - 1 - t;
				const uint32_t bifId = mark[dir][base[chr] + itPos];
				if (bifId == NONE) continue;
				const uint64_t itStart = dist(itPos, beginPos), itEnd = dist(itPos, lastPos);
				for (uint64_t a = off[bifId]; a < off[bifId + 1]; a++) {
					const uint32_t kc = ichr[a];
					if (kc == chr) continue;
					const uint64_t kn = len[kc], kStart = block[kc].dir == 0 ? 0 : kn - 1, kLast = block[kc].dir == 0 ? kn - 1 : 0;
					const uint64_t startSum = dist(ipos[a], kStart) + itStart, endSum = dist(ipos[a], kLast) + itEnd;
					if (startSum < minStartSum || (startSum == minStartSum && bifId < minBifStart)) { minBifStart = bifId; minStartSum = startSum; trimStart = itPos; }
					if (endSum < minEndSum || (endSum == minEndSum && bifId < minBifEnd)) { minBifEnd = bifId; minEndSum = endSum; trimEnd = itPos; }
				}
			}
			if (minStartSum < oo && minEndSum < oo) {
				if (dist(trimStart, trimEnd) + trimK >= minSize) {
					const uint64_t endElem = dir == 0 ? trimEnd + (trimK - 1) : trimEnd - (trimK - 1);      // std::advance(trimEnd, trimK - 1)
					BEdge x = block[chr];
					x.origPos = block[chr].origPos + std::min(trimStart, endElem);
					x.origLen = std::max(trimStart, endElem) + 1 - std::min(trimStart, endElem);
					ret.push_back(x);
				}
			} else drop = true;
		}
		block.swap(ret);
		return drop;
	}
};

}  // namespace

extern "C" sbl_status sbl_generate_blocks(sbl_ctx *c, uint32_t k, uint32_t trim_k, uint32_t min_size, int shared_only, const sbl_block **blocks, uint64_t *n)
{
	return guarded(c, [&] {
		SBL_CHECK(k >= 2 && trim_k >= 2, SBL_ERR_BAD_ARG, "vertex sizes must be at least 2");
		SBL_CHECK(c->d_orig_ch.p && c->orig_sepidx.size() == (size_t)c->nchr + 1, SBL_ERR_BAD_ARG, "no original records (load first)");
		// ---- ListEdges of a fresh index at k (serialization.cpp:56-86) on the GPU
		const sbl_edge *ge = nullptr; uint64_t ne = 0;
		sbl_status st = sbl_list_edges(c, k, &ge, &ne);
		if (st != SBL_OK) throw SblError{st, c->err};
		std::vector<BEdge> edge;
		edge.reserve(ne);
		for (uint64_t i = 0; i < ne; i++) {
			if (ge[i].orig_len < min_size) continue;                  // EdgeEmpty, edge.cpp:32-35
			BEdge e; e.chr = ge[i].chr; e.dir = ge[i].strand; e.startVertex = ge[i].start_vertex; e.endVertex = ge[i].end_vertex;
			e.origPos = ge[i].orig_pos; e.origLen = ge[i].orig_len; e.firstChar = ge[i].first_char;
			edge.push_back(e);
		}
		Synteny sy;
		sy.c = c; sy.trimK = trim_k; sy.minSize = min_size;
		if (!c->child) {
			sbl_status cs = sbl_create(&c->child, c->device);
			if (cs != SBL_OK) throw SblError{cs, "cannot create the trim context"};
		}
		sy.child = c->child;
		sy.occupied.resize(c->nchr); sy.local.resize(c->nchr);
		for (uint32_t i = 0; i < c->nchr; i++) {
			const size_t len = c->orig_sepidx[i + 1] - c->orig_sepidx[i] - 1;          // originalSize_
			sy.occupied[i].assign(len, 0); sy.local[i].assign(len, 0);
		}
		// ---- groups of edges that spell the same path (GroupBy over CompareEdgesNaturally, common.h:150-160), largest first
		std::sort(edge.begin(), edge.end(), natural_less);
		std::vector<std::pair<size_t, size_t>> group;
		for (size_t now = 0; now < edge.size();) {
			size_t prev = now;
			while (now < edge.size() && !natural_less(edge[prev], edge[now])) now++;
			group.push_back({prev, now});
		}
		std::sort(group.begin(), group.end(), [](const std::pair<size_t, size_t> &a, const std::pair<size_t, size_t> &b) { return a.second - a.first > b.second - b.first; });
		c->blocks.clear();
		int blockCount = 1;
		std::vector<BEdge> now;
		std::vector<uint32_t> occur(c->nchr);
		for (const auto &g : group) {
			BEdge *first = edge.data() + g.first, *last = edge.data() + g.second;
			std::sort(first, last, [](const BEdge &a, const BEdge &b) { return a.dir < b.dir; });      // CompareEdgesByDirection
			if (last - first < 2 || first->dir != 0) continue;          // fewer than two edges, or none on the positive strand (sorted: it would be first)
			sy.resolve_overlap(first, last, now);
			while (sy.trim_blocks(now)) { }
			std::fill(occur.begin(), occur.end(), 0u);
			for (const BEdge &e : now) occur[e.chr]++;
			if (now.size() > 1 && (!shared_only || (size_t)std::count(occur.begin(), occur.end(), 1u) == c->nchr)) {
				for (const BEdge &e : now) {
					memset(sy.occupied[e.chr].data() + e.origPos, 1, e.origLen);
					sbl_block b; b.id = e.dir == 0 ? blockCount : -blockCount; b.chr = e.chr; b.start = e.origPos; b.end = e.origPos + e.origLen;
					c->blocks.push_back(b);
				}
				blockCount++;
			}
		}
		std::sort(c->blocks.begin(), c->blocks.end(), [](const sbl_block &a, const sbl_block &b) { return std::make_pair(a.chr, a.start) < std::make_pair(b.chr, b.start); });
		if (blocks) *blocks = c->blocks.data();
		if (n) *n = c->blocks.size();
	});
}
