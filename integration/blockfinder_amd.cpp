// blockfinder_amd.cpp -- the reference-side binding of INTEGRATION.md as a real translation unit.
//
// Defines the PUBLIC members of the reference's own class SyntenyFinder::BlockFinder (declared in the reference's unmodified
// src/blockfinder.h:28-45) on top of libsibelia_amd.so.  Linked INSTEAD of the reference's blockfinder.cpp, bulgeremoval.cpp,
// edge.cpp, serialization.cpp and synteny.cpp, it turns the reference's program (src/sibelia.cpp: command line, FASTA reader,
// post-processor, output writers -- all unchanged) into one whose BlockFinder runs on the MI355X.  oracle/build_dropin.sh
// builds that program into oracle/_ref/ (it contains reference objects, so it is never committed);
// tests/test_gpu_dropin.py runs it on the GPU box and compares every output file with what the unmodified reference wrote.
//
// The header is not touched: the device context of a BlockFinder lives in a side table keyed by `this`.  The reference's class has no
// destructor (src/blockfinder.h:28-45), so nothing tells this file when an object dies: an entry is released when another BlockFinder
// is constructed at the same address, by SyntenyFinderAMD_ReleaseDevice(this) -- the one line a maintainer's `~BlockFinder()` holds
// ("Releasing the device context") -- or at process exit; the reference's own program keeps its single object in
// main's auto_ptr until it returns.
#include <map>
#include <memory>
#include <mutex>

#include "blockfinder.h"                      // the reference's (-I <reference>/src)
#include "platform.h"                         // the reference's CreateOutDirectory
#include "sibelia_amd/blockfinder.hpp"        // the C ABI + a thin C++ wrapper (this repository's include/)

namespace
{
	typedef SyntenyFinderAMD::BlockFinder Device;
	std::mutex tableLock;
	std::map<const SyntenyFinder::BlockFinder *, std::unique_ptr<Device> > table;

	Device & Of(const SyntenyFinder::BlockFinder * self)
	{
		std::lock_guard<std::mutex> hold(tableLock);
		return *table.at(self);
	}

	// boost::function<void(size_t, State)> of the reference -> std::function<void(size_t, State)> of the wrapper
	Device::ProgressCallBack Adapt(SyntenyFinder::BlockFinder::ProgressCallBack f)
	{
		if (f.empty()) return Device::ProgressCallBack();
		return [f](size_t progress, Device::State state) { f(progress, static_cast<SyntenyFinder::BlockFinder::State>(state)); };
	}
}

// releases the device context (sbl_destroy) of a BlockFinder that is about to die; harmless for an object that has none
extern "C" void SyntenyFinderAMD_ReleaseDevice(const void * blockFinder)
{
	std::lock_guard<std::mutex> hold(tableLock);
	table.erase(static_cast<const SyntenyFinder::BlockFinder *>(blockFinder));
}

namespace SyntenyFinder
{
	const char BlockFinder::SEPARATION_CHAR = '#';
	const char BlockFinder::POS_FREE = 0;
	const char BlockFinder::POS_OCCUPIED = 1;

	BlockFinder::BlockFinder(const std::vector<FASTARecord> & chrList): iseq_(0), originalChrList_(&chrList)
	{
		Init(chrList);
	}

	// tempDir: the reference spills its suffix array there unless -r is given; nothing is spilled here, but see Init
	BlockFinder::BlockFinder(const std::vector<FASTARecord> & chrList, const std::string & tempDir): tempDir_(tempDir), iseq_(0), originalChrList_(&chrList)
	{
		Init(chrList);
	}

	void BlockFinder::Init(const std::vector<FASTARecord> & chrList)
	{
		// sbl_create + sbl_load: throws std::runtime_error without an MI355X.  tempDir_ (set by the two-argument constructor, i.e. without
		// -r) switches the context to temp-file mode: the names of the reference's temporary files come out of the same rand() stream as
		// the replacements of ambiguous bases (src/platform.cpp:57), so the stream has to advance as if they had been created
		std::unique_ptr<Device> device(new Device(chrList, tempDir_));
		std::lock_guard<std::mutex> hold(tableLock);
		table[this] = std::move(device);
	}

	// Without -r every index the reference builds first makes sure its temp directory exists (src/vertexenumeration.cpp:187) -- by default
	// that is the output directory, and main relies on the side effect: the per-stage graph files of --allstages -g are only written
	// because the directory is already there (src/sibelia.cpp:256-262 open them without creating it).  Kept, so that the program behaves alike.
	namespace { void TempDirSideEffect(const std::string & tempDir) { if (!tempDir.empty()) CreateOutDirectory(tempDir); } }

	size_t BlockFinder::PerformGraphSimplifications(size_t k, size_t minBranchSize, size_t maxIterations, ProgressCallBack f)
	{
		TempDirSideEffect(tempDir_);
		return Of(this).PerformGraphSimplifications(k, minBranchSize, maxIterations, Adapt(f));
	}

	void BlockFinder::GenerateSyntenyBlocks(size_t k, size_t trimK, size_t minSize, std::vector<BlockInstance> & block, bool sharedOnly, ProgressCallBack f)
	{
		std::vector<SyntenyFinderAMD::BlockInstance> found;
		TempDirSideEffect(tempDir_);
		Of(this).GenerateSyntenyBlocks(k, trimK, minSize, found, sharedOnly, Adapt(f));
		block.clear();
		for (size_t i = 0; i < found.size(); i++)
		{
			block.push_back(BlockInstance(found[i].id, &(*originalChrList_)[found[i].chr], found[i].start, found[i].end));
		}
	}

	void BlockFinder::SerializeCondensedGraph(size_t k, std::ostream & out, ProgressCallBack f)
	{
		TempDirSideEffect(tempDir_);
		Of(this).SerializeCondensedGraph(k, out, Adapt(f));
	}

	void BlockFinder::SerializeGraph(size_t k, std::ostream & out)
	{
		Of(this).SerializeGraph(k, out);
	}
}
