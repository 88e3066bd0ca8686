// Golden-vector generator (TEST INFRASTRUCTURE, runs only in the build container).
//
// This is NOT reference code: it is a small driver that links against the object
// files of the *unmodified* reference build (out-of-tree under /tmp, see
// make_golden.sh) and dumps the observable results of the hot path
//   BlockFinder::PerformGraphSimplifications  (reference src/blockfinder.cpp:78-98)
//   IndexedSequence enumeration + marking     (reference src/indexedsequence.cpp:28-72)
//   BlockFinder::SerializeCondensedGraph      (reference src/serialization.cpp:88-110)
// into flat binary files that make_golden.py turns into committed fixtures.
//
// usage: ref_dump <in.fasta> <out-prefix> <cmd>...
//   cmd = state | hash:K | blocks:K:TRIMK:MINSIZE:SHARED | write:K:TRIMK:MINSIZE:SHARED:GLUE | graph:K | enum:K | stage:K:D:ITER | dot:K      (output i goes to <out-prefix>.<i>.out)
#include "common.h"
#include "fasta.h"
#include "dnasequence.h"
#define private public
#include "blockfinder.h"
#undef private
#include "hashing.h"
#include "postprocessor.h"
#include "outputgenerator.h"
#include <stdint.h>
#include <string.h>
#include <time.h>

const std::string VERSION("golden-dump");
using namespace SyntenyFinder;

static void put32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static void put64(FILE *f, uint64_t v) { fwrite(&v, 8, 1, f); }

static void dump_enum(BlockFinder &bf, size_t k, const std::string &path)
{
	IndexedSequence iseq(bf.rawSeq_, k, "");
	DNASequence &seq = iseq.Sequence();
	BifurcationStorage &st = iseq.BifStorage();
	FILE *f = fopen(path.c_str(), "wb");
	put32(f, (uint32_t)st.GetMaxId());
	for(size_t strand = 0; strand < 2; strand++)
	{
		std::vector<uint32_t> rec;
		for(size_t chr = 0; chr < seq.ChrNumber(); chr++)
		{
			size_t pos = 0;
			StrandIterator end = seq.End((DNASequence::Direction)strand, chr);
			for(StrandIterator it = seq.Begin((DNASequence::Direction)strand, chr); it != end; ++it, ++pos)
			{
				size_t id = st.GetBifurcation(it);
				if(id != BifurcationStorage::NO_BIFURCATION)
				{
					rec.push_back((uint32_t)id); rec.push_back((uint32_t)chr); rec.push_back((uint32_t)pos);
				}
			}
		}
		put64(f, rec.size() / 3);
		if(!rec.empty()) fwrite(&rec[0], 4, rec.size(), f);
	}
	fclose(f);
}

// H0 (reference src/hashing.h:14-100): SlidingWindow<StrandIterator> hash of every k-mer, strand 0 then 1, chromosomes ascending,
// walk order; per (strand, chr): u64 count, count x u64.  Every value is what Move() maintains and (assert) CalcKMerHash gives.
static void dump_hashes(BlockFinder &bf, size_t k, const std::string &path)
{
	std::vector<std::vector<Pos> > opos(bf.originalPos_);
	DNASequence seq(bf.rawSeq_, opos);
	FILE *f = fopen(path.c_str(), "wb");
	for(size_t strand = 0; strand < 2; strand++)
	{
		for(size_t chr = 0; chr < seq.ChrNumber(); chr++)
		{
			std::vector<uint64_t> val;
			StrandIterator begin = seq.Begin((DNASequence::Direction)strand, chr), end = seq.End((DNASequence::Direction)strand, chr);
			if(bf.rawSeq_[chr].size() >= k)
			{
				SlidingWindow<StrandIterator> window(begin, end, k);
				for(bool ok = window.Valid(); ok; ok = window.Move())
				{
					val.push_back(window.GetValue());
					if(window.GetValue() != SlidingWindow<StrandIterator>::CalcKMerHash(window.GetBegin(), k)) { fprintf(stderr, "rolling hash differs from CalcKMerHash\n"); exit(5); }
				}
			}
			put64(f, val.size());
			if(!val.empty()) fwrite(&val[0], 8, val.size(), f);
		}
	}
	fclose(f);
}

static void dump_state(BlockFinder &bf, uint64_t bulges, const std::string &path)
{
	FILE *f = fopen(path.c_str(), "wb");
	put64(f, bulges);
	put32(f, (uint32_t)bf.rawSeq_.size());
	for(size_t chr = 0; chr < bf.rawSeq_.size(); chr++)
	{
		put64(f, bf.rawSeq_[chr].size());
		fwrite(bf.rawSeq_[chr].data(), 1, bf.rawSeq_[chr].size(), f);
		if(!bf.originalPos_[chr].empty()) fwrite(&bf.originalPos_[chr][0], 4, bf.originalPos_[chr].size(), f);
	}
	fclose(f);
}

int main(int argc, char **argv)
{
	if(argc < 4) { fprintf(stderr, "usage\n"); return 2; }
	std::vector<FASTARecord> chrList;
	FASTAReader reader(argv[1]);
	if(!reader.IsOk()) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
	std::string prefix(argv[2]);
	try { reader.GetSequences(chrList); }
	catch(const std::exception & e)                    // parse errors of the reference's FASTA reader (src/fasta.cpp:66-71): <prefix>.err
	{
		FILE *f = fopen((prefix + ".err").c_str(), "wb");
		fputs(e.what(), f);
		fclose(f);
		return 4;
	}
	BlockFinder finder(chrList);
	for(int a = 3; a < argc; a++)
	{
		unsigned k = 0, d = 0, it = 0, sh = 0, gl = 0;
		char buf[64]; sprintf(buf, ".%d.out", a - 3);      // one output file per command, in order
		if(sscanf(argv[a], "enum:%u", &k) == 1)
		{
			dump_enum(finder, k, prefix + buf);
		}
		else if(sscanf(argv[a], "stage:%u:%u:%u", &k, &d, &it) == 3)
		{
			struct timespec t0, t1;
			clock_gettime(CLOCK_MONOTONIC, &t0);
			size_t bulges = finder.PerformGraphSimplifications(k, d, it);
			clock_gettime(CLOCK_MONOTONIC, &t1);
			dump_state(finder, bulges, prefix + buf);
			// seconds = wall time of the reference's own BlockFinder::PerformGraphSimplifications (bench.py: cpu_baseline kind "reference")
			fprintf(stderr, "stage k=%u D=%u iter=%u -> bulges=%zu seconds=%.6f\n", k, d, it, bulges,
			        (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
		}
		else if(sscanf(argv[a], "hash:%u", &k) == 1)
		{
			dump_hashes(finder, k, prefix + buf);
		}
		else if(strcmp(argv[a], "state") == 0)             // rawSeq_ / originalPos_ as BlockFinder::Init leaves them + the record descriptions
		{
			dump_state(finder, 0, prefix + buf);
			FILE *f = fopen((prefix + buf + ".names").c_str(), "wb");
			for(size_t i = 0; i < chrList.size(); i++) fprintf(f, "%s\n", chrList[i].GetDescription().c_str());
			fclose(f);
		}
		else if(sscanf(argv[a], "blocks:%u:%u:%u:%u", &k, &d, &it, &sh) == 4)
		{
			// N2: BlockFinder::GenerateSyntenyBlocks(k, trimK, minSize, block, sharedOnly) (src/synteny.cpp:229-286);
			// per BlockInstance, in output order: i32 signed block id | u32 chr | u64 start | u64 end
			std::vector<BlockInstance> block;
			struct timespec t0, t1;
			clock_gettime(CLOCK_MONOTONIC, &t0);
			finder.GenerateSyntenyBlocks(k, d, it, block, sh != 0);
			clock_gettime(CLOCK_MONOTONIC, &t1);
			FILE *f = fopen((prefix + buf).c_str(), "wb");
			put64(f, block.size());
			for(size_t i = 0; i < block.size(); i++)
			{
				put32(f, (uint32_t)block[i].GetSignedBlockId()); put32(f, (uint32_t)block[i].GetChrId());
				put64(f, block[i].GetStart()); put64(f, block[i].GetEnd());
			}
			fclose(f);
			fprintf(stderr, "blocks k=%u trimK=%u minSize=%u shared=%u -> %zu instances seconds=%.6f\n", k, d, it, sh, block.size(),
			        (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec));
		}
		else if(sscanf(argv[a], "write:%u:%u:%u:%u:%u", &k, &d, &it, &sh, &gl) == 5)
		{
			// N4: what main does after the stages (src/sibelia.cpp:287-315): GenerateSyntenyBlocks, Postprocessor::GlueStripes (unless
			// GL = 0), then the writers ListBlocksIndices (blocks_coords.txt), ListChromosomesAsPermutations, GenerateReport.
			// Output: u64 n, n x (i32 id, u32 chr, u64 start, u64 end) after glueing | 3 x (u64 len, text)
			std::vector<BlockInstance> block;
			finder.GenerateSyntenyBlocks(k, d, it, block, sh != 0);
			Postprocessor processor(chrList, it);
			if(gl) processor.GlueStripes(block);
			OutputGenerator generator(chrList);
			std::string tmp = prefix + buf + ".tmp";
			FILE *f = fopen((prefix + buf).c_str(), "wb");
			put64(f, block.size());
			for(size_t i = 0; i < block.size(); i++)
			{
				put32(f, (uint32_t)block[i].GetSignedBlockId()); put32(f, (uint32_t)block[i].GetChrId());
				put64(f, block[i].GetStart()); put64(f, block[i].GetEnd());
			}
			for(int w = 0; w < 3; w++)
			{
				if(w == 0) generator.ListBlocksIndices(block, tmp);
				if(w == 1) generator.ListChromosomesAsPermutations(block, tmp);
				if(w == 2) generator.GenerateReport(block, tmp);
				std::ifstream in(tmp.c_str(), std::ios::binary);
				std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
				put64(f, text.size());
				fwrite(text.data(), 1, text.size(), f);
			}
			fclose(f);
			remove(tmp.c_str());
		}
		else if(sscanf(argv[a], "graph:%u", &k) == 1)           // BlockFinder::SerializeGraph (src/serialization.cpp:112-138): the uncondensed graph
		{
			std::ofstream out((prefix + buf).c_str());
			finder.SerializeGraph(k, out);
		}
		else if(sscanf(argv[a], "dot:%u", &k) == 1)
		{
			std::ofstream out((prefix + buf).c_str());
			finder.SerializeCondensedGraph(k, out);
		}
		else { fprintf(stderr, "bad cmd %s\n", argv[a]); return 2; }
	}
	return 0;
}
