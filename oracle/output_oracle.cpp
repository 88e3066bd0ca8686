/*
 * output_oracle.cpp -- CPU ORACLE for SURVEY.md 8f N4.  TEST INFRASTRUCTURE ONLY (see sibelia_oracle.h).
 *
 * A restatement of Postprocessor::GlueStripes (reference src/postprocessor.cpp:37-154) and of the writers
 * OutputGenerator::ListBlocksIndices, ListChromosomesAsPermutations and GenerateReport
 * (src/outputgenerator.cpp:227-233 with :52-67, :203-219, :162-201 with :116-145 and :150-160).
 * C++ for the same reason as synteny_oracle.cpp: the order of the rows depends on libstdc++'s std::sort on equal keys.
 * Parity status: PINNED -- `write:` outputs of the unmodified reference in tests/golden/vectors.json.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include <algorithm>
#include <iomanip>
#include <iterator>
#include <sstream>
#include <string>
#include <vector>

extern "C" {
#include "sibelia_oracle.h"
}

namespace {

struct Block { int id; size_t chr, start, end; };
int Abs(int x) { return x > 0 ? x : -x; }
bool compareById(const Block &a, const Block &b) { return (size_t)Abs(a.id) < (size_t)Abs(b.id); }      /* src/blockinstance.cpp:11-16 */
bool compareByChrId(const Block &a, const Block &b) { return a.chr < b.chr; }
bool compareByStart(const Block &a, const Block &b) { return a.start < b.start; }
bool naturally(const Block &a, const Block &b) { return std::make_pair(a.chr, a.start) < std::make_pair(b.chr, b.start); }
const std::string DELIMITER(80, '-');                                                                       /* src/util.cpp:9 */

struct Stripe { int firstBlock, secondBlock; bool operator<(const Stripe &o) const { return firstBlock < o.firstBlock; } };

/* GlueStripes, src/postprocessor.cpp:37-154 */
void GlueStripes(std::vector<Block> &block, size_t nchr)
{
	std::vector<std::vector<Block> > perm(nchr);
	for (size_t i = 0; i < block.size(); i++) perm[block[i].chr].push_back(block[i]);
	for (size_t i = 0; i < perm.size(); i++) std::sort(perm[i].begin(), perm[i].end(), compareByStart);
	int sentinel = INT_MAX >> 1;
	bool glue = false;
	do {
		std::vector<Stripe> stripe;
		for (size_t chr = 0; chr < perm.size(); chr++)
			for (size_t i = 0; i < perm[chr].size(); i++) {
				int bid = perm[chr][i].id;
				Stripe s;
				if (bid > 0) { s.firstBlock = bid; s.secondBlock = i < perm[chr].size() - 1 ? perm[chr][i + 1].id : sentinel; }
				else { int prevBid = i > 0 ? perm[chr][i - 1].id : -sentinel; s.firstBlock = -bid; s.secondBlock = -prevBid; }
				stripe.push_back(s);
			}
		size_t now = 0, next = 0;
		std::sort(stripe.begin(), stripe.end());
		for (; now < stripe.size(); now = next) {
			glue = true;
			for (; next < stripe.size() && stripe[next].firstBlock == stripe[now].firstBlock; next++)
				if (stripe[next].secondBlock != stripe[now].secondBlock || stripe[next].secondBlock == sentinel || Abs(stripe[next].secondBlock) == stripe[next].firstBlock) glue = false;
			if (glue) {
				Stripe probe; probe.firstBlock = Abs(stripe[now].secondBlock); probe.secondBlock = 0;
				std::pair<std::vector<Stripe>::iterator, std::vector<Stripe>::iterator> range = std::equal_range(stripe.begin(), stripe.end(), probe);
				if ((size_t)(range.second - range.first) != next - now) glue = false; else break;
			}
		}
		if (glue) {
			int glueBid = stripe[now].firstBlock;
			for (size_t chr = 0; chr < perm.size(); chr++)
				for (size_t i = 0; i < perm[chr].size(); i++) {
					if (Abs(perm[chr][i].id) != glueBid) continue;
					if (perm[chr][i].id > 0) {
						perm[chr][i].end = perm[chr][i + 1].end;
						perm[chr].erase(perm[chr].begin() + i + 1);
					} else {
						Block &a = perm[chr][--i]; Block &b = perm[chr][i + 1];
						a.id = b.id; a.end = b.end;
						perm[chr].erase(perm[chr].begin() + i + 1);
					}
				}
		}
	} while (glue);
	block.clear();
	std::vector<int> oldId;
	for (size_t chr = 0; chr < perm.size(); chr++)
		for (size_t i = 0; i < perm[chr].size(); i++) { block.push_back(perm[chr][i]); oldId.push_back(Abs(perm[chr][i].id)); }
	std::sort(oldId.begin(), oldId.end());
	oldId.erase(std::unique(oldId.begin(), oldId.end()), oldId.end());
	for (size_t i = 0; i < block.size(); i++) {
		int sign = block[i].id > 0 ? +1 : -1;
		size_t newId = std::lower_bound(oldId.begin(), oldId.end(), Abs(block[i].id)) - oldId.begin() + 1;
		block[i].id = static_cast<int>(newId) * sign;
	}
}

/* GroupBy, src/common.h:150-160 */
template <class T, class F> void GroupBy(std::vector<T> &store, F pred, std::vector<std::pair<size_t, size_t> > &out)
{
	std::sort(store.begin(), store.end(), pred);
	for (size_t now = 0; now < store.size();) {
		size_t prev = now;
		for (; now < store.size() && !pred(store[prev], store[now]); now++);
		out.push_back(std::make_pair(prev, now));
	}
}

struct Chr { std::string name; size_t size; };

void ListChrs(const std::vector<Chr> &chr, std::ostream &out)          /* src/outputgenerator.cpp:150-160 */
{
	out << "Seq_id\tSize\tDescription" << std::endl;
	for (size_t i = 0; i < chr.size(); i++) out << i + 1 << '\t' << chr[i].size << '\t' << chr[i].name << std::endl;
	out << DELIMITER << std::endl;
}

std::string ListBlocksIndices(const std::vector<Block> &block, const std::vector<Chr> &chr)      /* :227-233, OutputBlocks :52-67, OutputIndex :44-50 */
{
	std::ostringstream out;
	ListChrs(chr, out);
	std::vector<std::pair<size_t, size_t> > group;
	std::vector<Block> blockList = block;
	GroupBy(blockList, compareById, group);
	for (size_t g = 0; g < group.size(); g++) {
		std::sort(blockList.begin() + group[g].first, blockList.begin() + group[g].second, compareByChrId);
		out << "Block #" << Abs(blockList[group[g].first].id) << std::endl;
		out << "Seq_id\tStrand\tStart\tEnd\tLength" << std::endl;
		for (size_t i = group[g].first; i < group[g].second; i++) {
			const Block &b = blockList[i];
			size_t cs = b.id > 0 ? b.start + 1 : b.end, ce = b.id > 0 ? b.end : b.start + 1;      /* GetConventionalStart / End, src/blockinstance.cpp:57-75 */
			out << b.chr + 1 << '\t' << (b.id < 0 ? '-' : '+') << '\t' << cs << '\t' << ce << '\t' << b.end - b.start << "\n";
		}
		out << DELIMITER << std::endl;
	}
	return out.str();
}

std::string ListChromosomesAsPermutations(const std::vector<Block> &block, const std::vector<Chr> &chr)      /* :203-219 */
{
	std::ostringstream out;
	std::vector<std::pair<size_t, size_t> > group;
	std::vector<Block> blockList = block;
	GroupBy(blockList, compareByChrId, group);
	for (size_t g = 0; g < group.size(); g++) {
		out.setf(std::ios_base::showpos);
		out << '>' << chr[blockList[group[g].first].chr].name << std::endl;
		std::sort(blockList.begin() + group[g].first, blockList.begin() + group[g].second, naturally);
		for (size_t i = group[g].first; i < group[g].second; i++) out << blockList[i].id << " ";
		out << "$" << std::endl;
	}
	return out.str();
}

typedef std::pair<size_t, std::vector<Block> > GroupedBlock;
bool ByFirstElement(const GroupedBlock &a, const GroupedBlock &b) { return a.first < b.first; }

std::string GenerateReport(const std::vector<Block> &block, const std::vector<Chr> &chr)      /* :162-201, CalculateCoverage :116-145 */
{
	std::ostringstream out;
	std::vector<GroupedBlock> sepBlock;
	std::vector<std::pair<size_t, size_t> > group;
	std::vector<Block> blockList = block;
	GroupBy(blockList, compareById, group);
	for (size_t g = 0; g < group.size(); g++)
		sepBlock.push_back(std::make_pair(group[g].second - group[g].first, std::vector<Block>(blockList.begin() + group[g].first, blockList.begin() + group[g].second)));
	ListChrs(chr, out);
	out << "Degree\tCount\tTotal";
	for (size_t i = 0; i < chr.size(); i++) out << "\tSeq " << i + 1;
	out << std::endl;
	group.clear();
	GroupBy(sepBlock, ByFirstElement, group);
	group.push_back(std::make_pair((size_t)0, sepBlock.size()));
	for (size_t g = 0; g < group.size(); g++) {
		if (g + 1 != group.size()) out << sepBlock[group[g].first].first << '\t' << group[g].second - group[g].first << '\t';
		else out << "All\t" << group[g].second - group[g].first << "\t";
		out.precision(2);
		out.setf(std::ostream::fixed);
		std::vector<double> ret;
		std::vector<char> cover;
		double totalBp = 0, totalCoveredBp = 0;
		for (size_t c = 0; c < chr.size(); c++) {
			totalBp += chr[c].size;
			cover.assign(chr[c].size, 0);
			for (size_t it = group[g].first; it < group[g].second; it++)
				for (size_t i = 0; i < sepBlock[it].second.size(); i++)
					if (sepBlock[it].second[i].chr == c) std::fill(cover.begin() + sepBlock[it].second[i].start, cover.begin() + sepBlock[it].second[i].end, 1);
			double nowCoveredBp = static_cast<double>(std::count(cover.begin(), cover.end(), 1));
			ret.push_back(nowCoveredBp / cover.size() * 100);
			totalCoveredBp += nowCoveredBp;
		}
		ret.insert(ret.begin(), totalCoveredBp / totalBp * 100);
		std::copy(ret.begin(), ret.end(), std::ostream_iterator<double>(out, "%\t"));
		out << std::endl;
	}
	out << DELIMITER << std::endl;
	return out.str();
}

}  // namespace

/* in: blocks as GenerateSyntenyBlocks returns them; names / sizes: the FASTA records.  glue != 0: GlueStripes first.
 * out_blocks + texts[0..2] (blocks_coords.txt, genomes_permutations.txt, coverage_report.txt) are malloc'd (orc_free). */
extern "C" int orc_postprocess(const orc_block *in, uint64_t n, uint32_t nchr, const char *const *names, const uint64_t *sizes, int glue,
                               orc_block **out_blocks, uint64_t *nout, char **texts, uint64_t *text_len)
{
	std::vector<Block> block(n);
	for (uint64_t i = 0; i < n; i++) { block[i].id = in[i].id; block[i].chr = in[i].chr; block[i].start = in[i].start; block[i].end = in[i].end; }
	std::vector<Chr> chr(nchr);
	for (uint32_t i = 0; i < nchr; i++) { chr[i].name = names[i]; chr[i].size = sizes[i]; }
	if (glue) GlueStripes(block, nchr);
	orc_block *ob = (orc_block *)malloc((block.size() ? block.size() : 1) * sizeof *ob);
	for (size_t i = 0; i < block.size(); i++) { ob[i].id = block[i].id; ob[i].chr = (uint32_t)block[i].chr; ob[i].start = block[i].start; ob[i].end = block[i].end; }
	*out_blocks = ob; *nout = block.size();
	std::string t[3] = { ListBlocksIndices(block, chr), ListChromosomesAsPermutations(block, chr), GenerateReport(block, chr) };
	for (int w = 0; w < 3; w++) {
		texts[w] = (char *)malloc(t[w].size() + 1);
		memcpy(texts[w], t[w].data(), t[w].size()); texts[w][t[w].size()] = 0;
		text_len[w] = t[w].size();
	}
	return 0;
}
