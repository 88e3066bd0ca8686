// postprocess.hip -- downstream of the synteny stage: Postprocessor::GlueStripes and the three text reports.
//
// Replaces Postprocessor::GlueStripes (reference src/postprocessor.cpp:37-154) and OutputGenerator::ListBlocksIndices,
// ListChromosomesAsPermutations and GenerateReport (src/outputgenerator.cpp:227-233 + :44-67, :203-219, :162-201 + :116-160): what
// main runs between GenerateSyntenyBlocks and the end (src/sibelia.cpp:287-315) to produce blocks_coords.txt,
// genomes_permutations.txt and coverage_report.txt.  A few hundred to a few thousand block instances of host bookkeeping and
// text formatting -- there is nothing here for the GPU; it lives behind the C ABI so that the reference's pipeline is complete
// on top of it.  Row order follows the reference's unstable sorts (the same libstdc++ calls on the same element order).
#include <climits>
#include <cstring>
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <set>

#include "sbl_ctx.h"

namespace {

inline int iabs(int x) { return x > 0 ? x : -x; }
struct ById { bool operator()(const sbl_block &a, const sbl_block &b) const { return (size_t)iabs(a.id) < (size_t)iabs(b.id); } };
struct ByChr { bool operator()(const sbl_block &a, const sbl_block &b) const { return a.chr < b.chr; } };
struct ByChrStart { bool operator()(const sbl_block &a, const sbl_block &b) const { return a.chr != b.chr ? a.chr < b.chr : a.start < b.start; } };

// runs of equal elements after a sort by `less` (GroupBy, src/common.h:150-160)
template <class T, class Less> std::vector<std::pair<size_t, size_t>> group_by(std::vector<T> &v, Less less)
{
	std::vector<std::pair<size_t, size_t>> g;
	std::sort(v.begin(), v.end(), less);
	for (size_t i = 0; i < v.size();) {
		size_t j = i;
		while (j < v.size() && !less(v[i], v[j])) j++;
		g.push_back({i, j});
		i = j;
	}
	return g;
}

// GlueStripes: two blocks that always occur next to each other, in the same relative orientation and as often as each other, are
// merged into one; repeated until nothing glues; ids renumbered densely at the end.
//
// The reference rebuilds and sorts the whole (block, follower) table for every single merge and takes the LOWEST block id that
// glues (postprocessor.cpp:60-131): O(n log n) per merge, a minute for the 88 k instances GenerateSyntenyBlocks(15, 15, 15) leaves on
// a raw bacterial graph.  The same sequence of merges falls out of a worklist, because merging X with its follower Y changes the
// status of X only: a block that saw a Y instance as its follower enters it from the far end and now sees the merged block
// reversed (-X) where it saw -Y, an injective relabelling that keeps "all followers equal" true or false, and count(X) = count(Y)
// keeps the count test; nothing but Y itself ever saw X from its Y side.  So: chromosomes as linked lists, instance lists per id,
// an ordered set of the ids that glue, and after each merge only X is examined again.  (The A/B test compares with the oracle's
// restatement of the reference's procedure, oracle/output_oracle.cpp -- test infrastructure, not part of this library.)
void glue_stripes(std::vector<sbl_block> &block, uint32_t nchr)
{
	const int sentinel = INT_MAX >> 1, NIL = -1;
	std::vector<std::vector<sbl_block>> perm(nchr);
	for (const sbl_block &b : block) perm[b.chr].push_back(b);
	std::vector<sbl_block> node;
	std::vector<int> prev, next, head(nchr, NIL);
	int maxId = 0;
	for (auto &p : perm) {
		std::sort(p.begin(), p.end(), [](const sbl_block &a, const sbl_block &b) { return a.start < b.start; });
		for (size_t i = 0; i < p.size(); i++) {
			const int me = (int)node.size();
			node.push_back(p[i]);
			prev.push_back(i ? me - 1 : NIL); next.push_back(i + 1 < p.size() ? me + 1 : NIL);
			if (!i) head[p[i].chr] = me;
			maxId = std::max(maxId, iabs(p[i].id));
		}
	}
	std::vector<std::vector<int>> inst((size_t)maxId + 1);           // live nodes of every block id
	for (int i = 0; i < (int)node.size(); i++) inst[iabs(node[i].id)].push_back(i);
	// what follows node i when its block is read in its own orientation (postprocessor.cpp:66-77)
	auto follower = [&](int i) -> int {
		if (node[i].id > 0) return next[i] != NIL ? node[next[i]].id : sentinel;
		return -(prev[i] != NIL ? node[prev[i]].id : -sentinel);
	};
	auto glues = [&](int b) -> bool {
		const std::vector<int> &v = inst[b];
		if (v.empty()) return false;
		const int f = follower(v[0]);
		if (f == sentinel || iabs(f) == b) return false;
		for (size_t i = 1; i < v.size(); i++) if (follower(v[i]) != f) return false;
		return inst[iabs(f)].size() == v.size();
	};
	std::set<int> work;
	for (int b = 1; b <= maxId; b++) if (glues(b)) work.insert(b);
	while (!work.empty()) {
		const int x = *work.begin();
		const int y = iabs(follower(inst[x][0]));
		for (int i : inst[x]) {
			const int j = node[i].id > 0 ? next[i] : prev[i];            // the follower's instance: swallowed by this one
			if (node[i].id > 0) { node[i].end = node[j].end; next[i] = next[j]; if (next[j] != NIL) prev[next[j]] = i; }
			else { node[i].start = node[j].start; prev[i] = prev[j]; if (prev[j] != NIL) next[prev[j]] = i; else head[node[i].chr] = i; }
		}
		inst[y].clear();
		work.erase(y);
		if (!glues(x)) work.erase(x);
	}
	block.clear();
	std::vector<int> ids;
	for (uint32_t c = 0; c < nchr; c++)
		for (int i = head[c]; i != NIL; i = next[i]) { block.push_back(node[i]); ids.push_back(iabs(node[i].id)); }
	std::sort(ids.begin(), ids.end());
	ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
	for (sbl_block &b : block) {
		const int rank = (int)(std::lower_bound(ids.begin(), ids.end(), iabs(b.id)) - ids.begin()) + 1;
		b.id = b.id > 0 ? rank : -rank;
	}
}

struct Records { std::vector<std::string> name; std::vector<uint64_t> size; };

// ---- the three reports.  The texts are fixed by the files the reference's users parse (blocks_coords.txt, genomes_permutations.txt,
// coverage_report.txt); they are assembled here as rows of a string with printf conversions -- "%+d" is what the reference's showpos
// stream prints for a signed block id, "%.2f%%" its fixed two-digit percentage (libstdc++ formats doubles through the same C
// conversion, NaN included) -- and the coverage of a group of blocks is the length of a union of intervals, not a byte mask per
// chromosome and row.  Row ORDER is the one part that cannot be chosen: ties of the reference's unstable sorts show in the output, so
// the same std::sort calls run on the same element order (group_by, then the sort inside each group).
struct Text {
	std::string s;
	void row(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
	{
		char buf[256];
		va_list ap;
		va_start(ap, fmt);
		const int n = vsnprintf(buf, sizeof buf, fmt, ap);
		va_end(ap);
		if (n < (int)sizeof buf) { s.append(buf, (size_t)(n > 0 ? n : 0)); return; }
		std::string big((size_t)n + 1, '\0');
		va_start(ap, fmt);
		vsnprintf(&big[0], big.size(), fmt, ap);
		va_end(ap);
		s.append(big.data(), (size_t)n);
	}
	void rule() { s.append(80, '-'); s += '\n'; }                       // DELIMITER, src/util.cpp:9
	// the table of records every report but the permutations starts with
	void records(const Records &r)
	{
		s += "Seq_id\tSize\tDescription\n";
		for (size_t i = 0; i < r.size.size(); i++) { row("%zu\t%llu\t", i + 1, (unsigned long long)r.size[i]); s += r.name[i]; s += '\n'; }
		rule();
	}
};

std::string blocks_coords(const std::vector<sbl_block> &block, const Records &r)
{
	Text t;
	t.records(r);
	std::vector<sbl_block> v = block;
	for (const auto &g : group_by(v, ById())) {
		std::sort(v.begin() + g.first, v.begin() + g.second, ByChr());
		t.row("Block #%d\nSeq_id\tStrand\tStart\tEnd\tLength\n", iabs(v[g.first].id));
		for (size_t i = g.first; i < g.second; i++) {
			// 1-based, and a reverse-strand instance is reported from its far end (src/blockinstance.cpp:57-75)
			const sbl_block &b = v[i];
			const bool fwd = b.id > 0;
			const unsigned long long from = fwd ? b.start + 1 : b.end, to = fwd ? b.end : b.start + 1;
			t.row("%u\t%c\t%llu\t%llu\t%llu\n", b.chr + 1, fwd ? '+' : '-', from, to, (unsigned long long)(b.end - b.start));
		}
		t.rule();
	}
	return t.s;
}

std::string permutations(const std::vector<sbl_block> &block, const Records &r)
{
	Text t;
	std::vector<sbl_block> v = block;
	for (const auto &g : group_by(v, ByChr())) {
		t.s += '>'; t.s += r.name[v[g.first].chr]; t.s += '\n';
		std::sort(v.begin() + g.first, v.begin() + g.second, ByChrStart());
		for (size_t i = g.first; i < g.second; i++) t.row("%+d ", v[i].id);
		t.s += "$\n";
	}
	return t.s;
}

// bases covered by a set of half-open intervals (sorted here)
uint64_t union_length(std::vector<std::pair<uint64_t, uint64_t>> &iv)
{
	std::sort(iv.begin(), iv.end());
	uint64_t total = 0, reach = 0;
	for (const auto &x : iv) {
		const uint64_t lo = x.first > reach ? x.first : reach;
		if (x.second > lo) { total += x.second - lo; reach = x.second; }
	}
	return total;
}

std::string coverage_report(const std::vector<sbl_block> &block, const Records &r)
{
	Text t;
	const size_t nrec = r.size.size();
	std::vector<sbl_block> v = block;
	// the blocks by degree (number of instances): (degree, first instance in v, one past the last)
	struct Deg { size_t degree, lo, hi; };
	std::vector<Deg> blocks;
	for (const auto &g : group_by(v, ById())) blocks.push_back({g.second - g.first, g.first, g.second});
	auto rows = group_by(blocks, [](const Deg &a, const Deg &b) { return a.degree < b.degree; });
	rows.push_back({0, blocks.size()});                                 // the last row: every block
	t.records(r);
	t.s += "Degree\tCount\tTotal";
	for (size_t c = 0; c < nrec; c++) t.row("\tSeq %zu", c + 1);
	t.s += '\n';
	std::vector<std::vector<std::pair<uint64_t, uint64_t>>> iv(nrec);
	for (size_t ri = 0; ri < rows.size(); ri++) {
		const size_t lo = rows[ri].first, hi = rows[ri].second;
		if (ri + 1 < rows.size()) t.row("%zu\t%zu\t", blocks[lo].degree, hi - lo);
		else t.row("All\t%zu\t", hi - lo);
		for (auto &x : iv) x.clear();
		for (size_t x = lo; x < hi; x++)
			for (size_t i = blocks[x].lo; i < blocks[x].hi; i++) iv[v[i].chr].push_back({v[i].start, v[i].end});
		double all_bases = 0, all_covered = 0;
		std::vector<double> share(nrec);
		for (size_t c = 0; c < nrec; c++) {
			const double covered = (double)union_length(iv[c]);
			all_bases += (double)r.size[c];
			all_covered += covered;
			share[c] = covered / (double)r.size[c] * 100;
		}
		t.row("%.2f%%\t", all_covered / all_bases * 100);
		for (size_t c = 0; c < nrec; c++) t.row("%.2f%%\t", share[c]);
		t.s += '\n';
	}
	t.rule();
	return t.s;
}

}  // namespace

extern "C" sbl_status sbl_postprocess(sbl_ctx *c, int glue, const char *const *names, const sbl_block **blocks, uint64_t *n,
                                      const char **coords, const char **perms, const char **coverage)
{
	return guarded(c, [&] {
		SBL_CHECK(c->orig_sepidx.size() == (size_t)c->nchr + 1, SBL_ERR_BAD_ARG, "no records loaded");
		Records r;
		for (uint32_t i = 0; i < c->nchr; i++) {
			r.size.push_back(c->orig_sepidx[i + 1] - c->orig_sepidx[i] - 1);
			r.name.push_back(names ? std::string(names[i]) : i < c->fa_names.size() ? c->fa_names[i] : std::string());
		}
		if (glue) glue_stripes(c->blocks, c->nchr);
		c->report[0] = blocks_coords(c->blocks, r);
		c->report[1] = permutations(c->blocks, r);
		c->report[2] = coverage_report(c->blocks, r);
		if (blocks) *blocks = c->blocks.data();
		if (n) *n = c->blocks.size();
		if (coords) *coords = c->report[0].c_str();
		if (perms) *perms = c->report[1].c_str();
		if (coverage) *coverage = c->report[2].c_str();
	});
}

// Postprocessor::GlueStripes on a caller's block list (the reference's main applies it to the blocks of EVERY stage with -v / --allstages,
// not only to the last GenerateSyntenyBlocks): in place, *n updated.  Host bookkeeping only -- no context, no device.
extern "C" sbl_status sbl_glue_stripes(sbl_block *blocks, uint64_t *n, uint32_t nchr)
{
	if (!n || (*n && !blocks)) return SBL_ERR_BAD_ARG;
	try {
		std::vector<sbl_block> v(blocks, blocks + *n);
		for (const sbl_block &b : v) if (b.chr >= nchr || b.id == 0 || b.end < b.start) return SBL_ERR_BAD_ARG;
		glue_stripes(v, nchr);
		if (v.size() > *n) return SBL_ERR_INTERNAL;
		std::copy(v.begin(), v.end(), blocks);
		*n = v.size();
		return SBL_OK;
	} catch (const std::bad_alloc &) { return SBL_ERR_OOM; }
	catch (...) { return SBL_ERR_INTERNAL; }
}
