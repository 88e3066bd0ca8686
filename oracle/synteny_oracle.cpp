/*
 * synteny_oracle.cpp -- CPU ORACLE for SURVEY.md 8f N2.  TEST INFRASTRUCTURE ONLY (see sibelia_oracle.h).
 *
 * A restatement of BlockFinder::GenerateSyntenyBlocks, TrimBlocks and ResolveOverlap
 * (reference src/synteny.cpp:229-286, :31-122, :124-166) with their helpers (src/edge.cpp:16-42,
 * src/common.h:150-160 GroupBy, src/blockfinder.h:93-108 EdgeGroupComparer, src/blockinstance.cpp:131-134)
 * on top of the C oracle's enumeration (orc_list_edges / orc_enumerate).
 *
 * C++ because the reference's result depends on what libstdc++'s std::sort does with equal elements (three unstable sorts:
 * common.h:153, synteny.cpp:249 and :254); the same calls on the same element order are made here.
 * Parity status: PINNED -- tests/golden/vectors.json holds `blocks:` outputs of the unmodified reference (oracle/_ref).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <set>
#include <string>
#include <utility>
#include <vector>

extern "C" {
#include "sibelia_oracle.h"
}

namespace {

struct Edge {                                   /* BlockFinder::Edge, src/blockfinder.h:58-90 */
	size_t chr; int direction;                  /* 0 = positive, 1 = negative (DNASequence::Direction) */
	size_t startVertex, endVertex, actualPosition, actualLength, originalPosition, originalLength;
	char firstChar;
};

/* EdgeToVector / CompareEdgesNaturally, src/edge.cpp:26-40 */
bool CompareEdgesNaturally(const Edge &a, const Edge &b)
{
	size_t fa[3] = { a.startVertex, a.endVertex, static_cast<size_t>(a.firstChar) };
	size_t fb[3] = { b.startVertex, b.endVertex, static_cast<size_t>(b.firstChar) };
	return std::lexicographical_compare(fa, fa + 3, fb, fb + 3);
}
bool CompareEdgesByDirection(const Edge &a, const Edge &b) { return a.direction < b.direction; }     /* src/edge.cpp:16-19 */

typedef std::pair<size_t, size_t> ChrPos;
typedef std::vector<char> Indicator;
const char POS_FREE = 0, POS_OCCUPIED = 1;

/* ResolveOverlap, src/synteny.cpp:124-166 */
void ResolveOverlap(std::vector<Edge>::iterator start, std::vector<Edge>::iterator end, size_t minSize, std::vector<Indicator> &overlap, std::vector<Edge> &nowBlock)
{
	nowBlock.clear();
	std::set<ChrPos> localOverlap;
	for (; start != end; ++start) {
		size_t segEnd = 0, bestStart = 0, bestEnd = 0;
		size_t chrNumber = start->chr;
		size_t stop = start->originalPosition + start->originalLength;
		for (size_t segStart = start->originalPosition; segStart < stop; segStart = segEnd) {
			for (segEnd = segStart; segEnd < stop && overlap[chrNumber][segEnd] == POS_FREE; segEnd++)
				if (localOverlap.count(ChrPos(chrNumber, segEnd)) > 0) break;
			if (segEnd - segStart > bestEnd - bestStart) { bestStart = segStart; bestEnd = segEnd; }
			segEnd += segEnd == segStart ? 1 : 0;
		}
		if (bestEnd - bestStart >= minSize) {
			Edge e = *start;
			e.originalPosition = bestStart; e.originalLength = bestEnd - bestStart;
			nowBlock.push_back(e);
			for (size_t pos = bestStart; pos < bestEnd; pos++) localOverlap.insert(ChrPos(chrNumber, pos));
		}
	}
}

struct Ctx {
	orc_ctx *main;
	std::vector<std::string> original;          /* originalChrList_ sequences */
};

/* TrimBlocks, src/synteny.cpp:31-122.  The IndexedSequence over the block sequences is a child oracle context that shares
 * the parent's rand() stream (sanitising the copy consumes it, src/indexedsequence.cpp:31-37). */
bool TrimBlocks(Ctx &cx, std::vector<Edge> &block, size_t trimK, size_t minSize)
{
	bool drop = false;
	std::vector<std::string> blockSeq(block.size());
	for (size_t i = 0; i < block.size(); i++)
		blockSeq[i] = cx.original[block[i].chr].substr(block[i].originalPosition, block[i].originalLength);
	const size_t oo = UINT32_MAX;
	/* IndexedSequence iseq(blockSeq, trimK, ""): enumeration of a fresh index */
	orc_ctx *child = orc_create();
	std::vector<const uint8_t *> ptr(block.size()); std::vector<uint64_t> len(block.size());
	for (size_t i = 0; i < block.size(); i++) { ptr[i] = reinterpret_cast<const uint8_t *>(blockSeq[i].data()); len[i] = blockSeq[i].size(); }
	orc_load(child, (uint32_t)block.size(), ptr.data(), len.data());
	orc_rng_copy(child, cx.main);
	uint32_t bifCount = 0; const orc_inst *inst[2] = { 0, 0 }; uint64_t ninst[2] = { 0, 0 };
	orc_enumerate(child, (uint32_t)trimK, &bifCount, &inst[0], &ninst[0], &inst[1], &ninst[1]);
	orc_rng_copy(cx.main, child);
	/* GetBifurcation(it) on the strand of the walk: mark[strand][chr][original position of the element the k-mer starts at] */
	std::vector<std::vector<uint32_t> > mark[2];
	for (int s = 0; s < 2; s++) {
		mark[s].resize(block.size());
		for (size_t c = 0; c < block.size(); c++) mark[s][c].assign(blockSeq[c].size(), UINT32_MAX);
		for (uint64_t i = 0; i < ninst[s]; i++) {
			size_t c = inst[s][i].chr, p = inst[s][i].pos;          /* negative strand: reverse-complement coordinate -> element len - 1 - p */
			mark[s][c][s == 0 ? p : blockSeq[c].size() - 1 - p] = inst[s][i].id;
		}
	}
	/* ListPositions(bifId) (src/bifurcationstorage.h:59-72): + list then - list, each in slist order = front insertion while the
	 * marking loop scans (chr, walk position) ascending (src/indexedsequence.cpp:49-67) => descending (chr, walk position) */
	std::vector<std::vector<std::pair<size_t, size_t> > > positions(bifCount + 1);   /* (chr, element position) */
	for (int s = 0; s < 2; s++)
		for (uint64_t i = ninst[s]; i-- > 0; ) {
			size_t c = inst[s][i].chr, p = inst[s][i].pos;
			positions[inst[s][i].id].push_back(std::make_pair(c, s == 0 ? p : blockSeq[c].size() - 1 - p));
		}
	std::vector<Edge> ret;
	for (size_t chr = 0; chr < block.size(); chr++) {
		const int dir = block[chr].direction;
		const size_t n = blockSeq[chr].size();
		/* begin .. end of the walk on strand dir; walk step t visits element (dir == 0 ? t : n - 1 - t) */
		size_t trimStart = SIZE_MAX, trimEnd = SIZE_MAX;           /* element positions; SIZE_MAX = `end` */
		size_t minBifStart = oo, minBifEnd = oo, minStartSum = oo, minEndSum = oo;
		const size_t beginPos = dir == 0 ? 0 : n - 1, lastPos = dir == 0 ? n - 1 : 0;      /* begin, AdvanceBackward(end, 1) */
		for (size_t t = 0; t < n; t++) {
			const size_t itPos = dir == 0 ? t : n - 1 - t;
			const size_t bifId = mark[dir][chr][itPos];
			if (bifId == UINT32_MAX) continue;
			const std::vector<std::pair<size_t, size_t> > &startKMer = positions[bifId];
			for (size_t pos = 0; pos < startKMer.size(); pos++) {
				const size_t kmerChr = startKMer[pos].first, kmerPos = startKMer[pos].second;
				if (chr == kmerChr) continue;
				const size_t kn = blockSeq[kmerChr].size();
				const size_t kmerChrStart = block[kmerChr].direction == 0 ? 0 : kn - 1, kmerChrLast = block[kmerChr].direction == 0 ? kn - 1 : 0;
				/* StrandIteratorDistance = |difference of original positions| (src/indexedsequence.cpp:162-167) */
				const size_t kmerStartDist = kmerPos > kmerChrStart ? kmerPos - kmerChrStart : kmerChrStart - kmerPos;
				const size_t kmerEndDist = kmerPos > kmerChrLast ? kmerPos - kmerChrLast : kmerChrLast - kmerPos;
				const size_t itStartDist = itPos > beginPos ? itPos - beginPos : beginPos - itPos;
				const size_t itEndDist = itPos > lastPos ? itPos - lastPos : lastPos - itPos;
				const size_t nowStartSum = kmerStartDist + itStartDist, nowEndSum = kmerEndDist + itEndDist;
				if (nowStartSum < minStartSum || (nowStartSum == minStartSum && bifId < minBifStart)) { minBifStart = bifId; minStartSum = nowStartSum; trimStart = itPos; }
				if (nowEndSum < minEndSum || (nowEndSum == minEndSum && bifId < minBifEnd)) { minBifEnd = bifId; minEndSum = nowEndSum; trimEnd = itPos; }
			}
		}
		if (minStartSum < oo && minEndSum < oo) {
			size_t size = (trimStart > trimEnd ? trimStart - trimEnd : trimEnd - trimStart) + trimK;
			if (size >= minSize) {
				/* std::advance(trimEnd, trimK - 1) along the walk */
				const size_t endElem = dir == 0 ? trimEnd + (trimK - 1) : trimEnd - (trimK - 1);
				size_t start = block[chr].originalPosition + std::min(trimStart, endElem);
				size_t end = block[chr].originalPosition + std::max(trimStart, endElem) + 1;
				Edge e = block[chr];
				e.originalPosition = start; e.originalLength = end - start;
				ret.push_back(e);
			}
		} else drop = true;
	}
	orc_destroy(child);
	block.swap(ret);
	return drop;
}

}  // namespace

/* GenerateSyntenyBlocks, src/synteny.cpp:229-286.  orig_seq / orig_len: the FASTA records the BlockFinder was built from
 * (originalChrList_).  out (malloc'd, orc_free): per BlockInstance i32 signed id, u32 chr, u64 start, u64 end = 24 bytes. */
extern "C" int orc_generate_blocks(orc_ctx *c, const uint8_t *const *orig_seq, const uint64_t *orig_len, uint32_t k, uint32_t trimK, uint32_t minSize,
                                   int sharedOnly, orc_block **out, uint64_t *nout)
{
	const uint32_t nchr = orc_nchr(c);
	Ctx cx; cx.main = c;
	for (uint32_t i = 0; i < nchr; i++) cx.original.push_back(std::string(reinterpret_cast<const char *>(orig_seq[i]), orig_len[i]));
	std::vector<Indicator> overlap(nchr);
	for (uint32_t i = 0; i < nchr; i++) overlap[i].assign(orig_len[i], POS_FREE);           /* originalSize_ */
	const orc_edge *oe = 0; uint64_t ne = 0;
	if (orc_list_edges(c, k, &oe, &ne)) return 1;
	std::vector<Edge> edge(ne);
	for (uint64_t i = 0; i < ne; i++) {
		Edge e; e.chr = oe[i].chr; e.direction = (int)oe[i].strand; e.startVertex = oe[i].start_vertex; e.endVertex = oe[i].end_vertex;
		e.actualPosition = oe[i].pos; e.actualLength = oe[i].len; e.originalPosition = oe[i].orig_pos; e.originalLength = oe[i].orig_len; e.firstChar = oe[i].first_char;
		edge[i] = e;
	}
	std::vector<orc_block> block;
	int blockCount = 1;
	edge.erase(std::remove_if(edge.begin(), edge.end(), [&](const Edge &a) { return a.originalLength < minSize; }), edge.end());      /* EdgeEmpty */
	std::vector<std::pair<size_t, size_t> > group;
	std::sort(edge.begin(), edge.end(), CompareEdgesNaturally);                                                                      /* GroupBy, src/common.h:150-160 */
	for (size_t now = 0; now < edge.size(); ) {
		size_t prev = now;
		for (; now < edge.size() && !CompareEdgesNaturally(edge[prev], edge[now]); now++);
		group.push_back(std::make_pair(prev, now));
	}
	std::sort(group.begin(), group.end(), [](const std::pair<size_t, size_t> &a, const std::pair<size_t, size_t> &b) { return a.second - a.first > b.second - b.first; });
	for (size_t g = 0; g < group.size(); g++) {
		std::vector<Edge>::iterator firstEdge = edge.begin() + group[g].first, lastEdge = edge.begin() + group[g].second;
		std::sort(firstEdge, lastEdge, CompareEdgesByDirection);
		if (lastEdge - firstEdge < 2 || std::find_if(firstEdge, lastEdge, [](const Edge &e) { return e.direction == 0; }) == lastEdge) continue;
		std::vector<Edge> nowBlock;
		std::vector<size_t> occur(nchr, 0);
		ResolveOverlap(firstEdge, lastEdge, minSize, overlap, nowBlock);
		while (TrimBlocks(cx, nowBlock, trimK, minSize));
		for (size_t i = 0; i < nowBlock.size(); i++) occur[nowBlock[i].chr]++;
		if (nowBlock.size() > 1 && (!sharedOnly || (size_t)std::count(occur.begin(), occur.end(), (size_t)1) == nchr)) {
			for (size_t i = 0; i < nowBlock.size(); i++) {
				int strand = nowBlock[i].direction == 0 ? +1 : -1;
				size_t start = nowBlock[i].originalPosition, end = start + nowBlock[i].originalLength;
				std::fill(overlap[nowBlock[i].chr].begin() + start, overlap[nowBlock[i].chr].begin() + end, POS_OCCUPIED);
				orc_block b; b.id = blockCount * strand; b.chr = (uint32_t)nowBlock[i].chr; b.start = start; b.end = end;
				block.push_back(b);
			}
			blockCount++;
		}
	}
	std::sort(block.begin(), block.end(), [](const orc_block &a, const orc_block &b) { return std::make_pair(a.chr, a.start) < std::make_pair(b.chr, b.start); });
	orc_block *r = (orc_block *)malloc((block.size() ? block.size() : 1) * sizeof *r);
	if (!r) return 2;
	if (!block.empty()) memcpy(r, block.data(), block.size() * sizeof *r);
	*out = r; *nout = block.size();
	return 0;
}
