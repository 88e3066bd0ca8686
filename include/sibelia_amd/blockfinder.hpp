// blockfinder.hpp -- the reference's BlockFinder class surface on top of the C ABI (include/sibelia_amd.h).
//
// Mirrors SyntenyFinder::BlockFinder's public section (reference src/blockfinder.h:28-45) for the hot path:
// same constructors (a FASTARecord only needs GetSequence()), same method names, argument order and meaning:
// PerformGraphSimplifications, GenerateSyntenyBlocks, SerializeCondensedGraph, SerializeGraph.
// Header-only; link with -lsibelia_amd.  Errors that the reference cannot produce (no device, OOM, input
// beyond the 29-bit limits) are thrown as std::runtime_error, the only exception type the reference itself throws
// (src/platform.cpp:40,79,118,126).
#ifndef SIBELIA_AMD_BLOCKFINDER_HPP
#define SIBELIA_AMD_BLOCKFINDER_HPP
#include <cstdio>
#include <exception>
#include <functional>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sibelia_amd.h"

namespace SyntenyFinderAMD
{
	// BlockInstance (reference src/blockinstance.h:21-47): signed block id (sign = strand), chromosome, [start, end) in original coordinates
	struct BlockInstance
	{
		int id; size_t chr, start, end;
		int GetSignedBlockId() const { return id; }
		int GetBlockId() const { return id < 0 ? -id : id; }
		int GetSign() const { return id > 0 ? +1 : -1; }
		size_t GetChrId() const { return chr; }
		size_t GetStart() const { return start; }
		size_t GetEnd() const { return end; }
		size_t GetLength() const { return end - start; }
		size_t GetConventionalStart() const { return id > 0 ? start + 1 : end; }      // blockinstance.cpp:57-75
		size_t GetConventionalEnd() const { return id > 0 ? end : start + 1; }
		bool operator<(const BlockInstance &o) const { return chr != o.chr ? chr < o.chr : start < o.start; }
	};

	struct FromFasta { std::string path; };     // BlockFinder(FromFasta{"genomes.fasta"}): FASTAReader + Init on the device (sbl_load_fasta)

	class BlockFinder
	{
	public:
		enum State { start, run, end };                                   // blockfinder.h:31-36
		typedef std::function<void(size_t, State)> ProgressCallBack;      // boost::function in the reference

		template <class FASTARecordVector>
		explicit BlockFinder(const FASTARecordVector &chrList, int device = -1) { Init(chrList, device); }
		template <class FASTARecordVector>
		// tempDir: nothing is spilled; a non-empty one (the reference's mode without -r) keeps the rand() stream in step with the names
		// of the temporary files the reference would create there (sbl_set_tempfile_mode)
		BlockFinder(const FASTARecordVector &chrList, const std::string &tempDir, int device = -1)
		{
			Init(chrList, device);
			if (!tempDir.empty()) Check(sbl_set_tempfile_mode(ctx_, 1), "BlockFinder");
		}
		explicit BlockFinder(const FromFasta &f, int device = -1)
		{
			sbl_status st = sbl_create(&ctx_, device);
			if (st != SBL_OK) throw std::runtime_error(std::string("sibelia_amd: ") + sbl_strerror(st));
			try { Check(sbl_load_fasta(ctx_, f.path.c_str()), "FASTAReader"); }
			catch (...) { sbl_destroy(ctx_); ctx_ = nullptr; throw; }
		}
		~BlockFinder() { sbl_destroy(ctx_); }
		size_t ChrNumber() const { return sbl_nchr(ctx_); }
		std::string Description(size_t chr) const { return sbl_record_name(ctx_, (uint32_t)chr); }      // FASTARecord::GetDescription
		BlockFinder(const BlockFinder &) = delete;
		BlockFinder &operator=(const BlockFinder &) = delete;

		// blockfinder.cpp:78-98
		size_t PerformGraphSimplifications(size_t k, size_t minBranchSize, size_t maxIterations, ProgressCallBack f = ProgressCallBack())
		{
			uint64_t bulges = 0;
			CallBackBox box{f, nullptr};
			sbl_status st = sbl_simplify_stage(ctx_, (uint32_t)k, (uint32_t)minBranchSize, (uint32_t)maxIterations,
			                                   f ? &Trampoline : nullptr, f ? &box : nullptr, &bulges);
			if (box.thrown) std::rethrow_exception(box.thrown);      // a throwing callback never unwinds through the C ABI: the stage completes first
			Check(st, "PerformGraphSimplifications");
			return (size_t)bulges;
		}

		// synteny.cpp:229-286 (blockfinder.h:43)
		void GenerateSyntenyBlocks(size_t k, size_t trimK, size_t minSize, std::vector<BlockInstance> &block, bool sharedOnly = false, ProgressCallBack = ProgressCallBack())
		{
			const sbl_block *b = nullptr;
			uint64_t n = 0;
			Check(sbl_generate_blocks(ctx_, (uint32_t)k, (uint32_t)trimK, (uint32_t)minSize, sharedOnly ? 1 : 0, &b, &n), "GenerateSyntenyBlocks");
			block.clear();
			for (uint64_t i = 0; i < n; i++) block.push_back(BlockInstance{b[i].id, b[i].chr, (size_t)b[i].start, (size_t)b[i].end});
		}

		// What main runs after GenerateSyntenyBlocks (sibelia.cpp:287-315) on the blocks of the last call: Postprocessor::GlueStripes
		// (postprocessor.cpp:37-154, unless glue is false) and the texts of blocks_coords.txt / genomes_permutations.txt /
		// coverage_report.txt (outputgenerator.cpp:162-233).  descriptions: FASTARecord::GetDescription per record (empty: those of the FASTA file).
		void PostProcess(bool glue, std::vector<BlockInstance> &block, std::string &blocksCoords, std::string &permutations, std::string &coverageReport,
		                 const std::vector<std::string> &descriptions = std::vector<std::string>())
		{
			std::vector<const char *> nm;
			if (!descriptions.empty() && descriptions.size() != sbl_nchr(ctx_)) throw std::runtime_error("sibelia_amd: PostProcess: one description per record expected");
			for (const std::string &d : descriptions) nm.push_back(d.c_str());
			const sbl_block *b = nullptr; uint64_t n = 0;
			const char *t0 = nullptr, *t1 = nullptr, *t2 = nullptr;
			Check(sbl_postprocess(ctx_, glue ? 1 : 0, nm.empty() ? nullptr : nm.data(), &b, &n, &t0, &t1, &t2), "PostProcess");
			block.clear();
			for (uint64_t i = 0; i < n; i++) block.push_back(BlockInstance{b[i].id, b[i].chr, (size_t)b[i].start, (size_t)b[i].end});
			blocksCoords = t0; permutations = t1; coverageReport = t2;
		}

		// Postprocessor::GlueStripes (postprocessor.cpp:37-154) on any block list, e.g. the blocks of an earlier stage (sibelia.cpp:247-253)
		static void GlueStripes(std::vector<BlockInstance> &block, size_t chrCount)
		{
			std::vector<sbl_block> flat(block.size());
			for (size_t i = 0; i < block.size(); i++) { flat[i].id = block[i].id; flat[i].chr = (uint32_t)block[i].chr; flat[i].start = block[i].start; flat[i].end = block[i].end; }
			uint64_t n = flat.size();
			if (sbl_glue_stripes(flat.empty() ? nullptr : flat.data(), &n, (uint32_t)chrCount) != SBL_OK) throw std::runtime_error("sibelia_amd: GlueStripes: bad block list");
			block.clear();
			for (uint64_t i = 0; i < n; i++) block.push_back(BlockInstance{flat[i].id, flat[i].chr, (size_t)flat[i].start, (size_t)flat[i].end});
		}

		// serialization.cpp:88-110 (same text, byte for byte)
		void SerializeCondensedGraph(size_t k, std::ostream &out, ProgressCallBack = ProgressCallBack())
		{
			const sbl_edge *e = nullptr;
			uint64_t n = 0;
			Check(sbl_list_edges(ctx_, (uint32_t)k, &e, &n), "SerializeCondensedGraph");
			out << "digraph G" << std::endl << "{" << std::endl << "rankdir=LR" << std::endl;
			for (uint64_t i = 0; i < n; i++) {
				char buf[256];
				std::snprintf(buf, sizeof buf, "[color=\"%s\", label=\"chr=%i pos=%i len=%i orpos=%i orlen=%i  ch='%c'\"];",
				              e[i].strand == 0 ? "blue" : "red", (int)e[i].chr, (int)e[i].pos, (int)e[i].len,
				              (int)e[i].orig_pos, (int)e[i].orig_len, e[i].first_char);
				out << e[i].start_vertex << " -> " << e[i].end_vertex << " " << buf << std::endl;
			}
			out << "}" << std::endl;
		}

		// serialization.cpp:112-138
		void SerializeGraph(size_t k, std::ostream &out)
		{
			const char *t = nullptr; uint64_t n = 0;
			Check(sbl_serialize_graph(ctx_, (uint32_t)k, &t, &n), "SerializeGraph");
			out.write(t, (std::streamsize)n);
		}

		// rawSeq_ / originalPos_ (blockfinder.h:52-54) after the last stage
		std::string Sequence(size_t chr) const
		{
			const uint8_t *s = nullptr; const uint32_t *p = nullptr; uint64_t n = 0;
			Check(sbl_get_state(ctx_, (uint32_t)chr, &s, &p, &n), "Sequence");
			return std::string(reinterpret_cast<const char *>(s), (size_t)n);
		}
		std::vector<uint32_t> OriginalPositions(size_t chr) const
		{
			const uint8_t *s = nullptr; const uint32_t *p = nullptr; uint64_t n = 0;
			Check(sbl_get_state(ctx_, (uint32_t)chr, &s, &p, &n), "OriginalPositions");
			return std::vector<uint32_t>(p, p + n);
		}
		sbl_ctx *Context() const { return ctx_; }

		// Several GPUs (no counterpart in the reference): one BlockFinder per GPU on the same input; with a communicator
		// attached the enumeration of every stage is sharded by k-mer hash prefix and the calls become collective.
		static std::string CommUniqueId()                                 // rank 0; distribute the bytes to every rank
		{
			std::string id(SBL_COMM_ID_BYTES, '\0');
			if (sbl_comm_unique_id(&id[0]) != SBL_OK) throw std::runtime_error("sibelia_amd: RCCL is not available");
			return id;
		}
		void AttachRccl(unsigned rank, unsigned nranks, const std::string &id) { Check(sbl_comm_attach_rccl(ctx_, rank, nranks, id.data()), "AttachRccl"); }
		void AttachLocal(sbl_group *group, unsigned rank) { Check(sbl_comm_attach_local(ctx_, group, rank), "AttachLocal"); }
		void Detach() { Check(sbl_comm_detach(ctx_), "Detach"); }

	private:
		template <class FASTARecordVector>
		void Init(const FASTARecordVector &chrList, int device)       // blockfinder.cpp:65-76
		{
			sbl_status st = sbl_create(&ctx_, device);
			if (st != SBL_OK) throw std::runtime_error(std::string("sibelia_amd: ") + sbl_strerror(st));
			std::vector<const uint8_t *> ptr;
			std::vector<uint64_t> len;
			for (const auto &rec : chrList) {
				const std::string &s = rec.GetSequence();
				ptr.push_back(reinterpret_cast<const uint8_t *>(s.data()));
				len.push_back(s.size());
			}
			try { Check(sbl_load(ctx_, (uint32_t)ptr.size(), ptr.data(), len.data()), "Init"); }
			catch (...) { sbl_destroy(ctx_); ctx_ = nullptr; throw; }        // the destructor of a half-built object never runs
		}
		struct CallBackBox { ProgressCallBack &f; std::exception_ptr thrown; };
		static void Trampoline(size_t progress, int state, void *user)
		{
			CallBackBox *box = static_cast<CallBackBox *>(user);
			if (box->thrown) return;
			try { box->f(progress, static_cast<State>(state)); } catch (...) { box->thrown = std::current_exception(); }
		}
		void Check(sbl_status st, const char *what) const
		{
			if (st != SBL_OK) throw std::runtime_error(std::string("sibelia_amd: ") + what + ": " + sbl_strerror(st) + " (" + sbl_last_error(ctx_) + ")");
		}
		sbl_ctx *ctx_ = nullptr;
	};
}
#endif
