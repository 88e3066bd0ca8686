// postprocessor_amd.cpp -- optional second binding: Postprocessor::GlueStripes (the reference's unmodified src/postprocessor.h:22)
// over sbl_glue_stripes.  The reference rebuilds and sorts its whole (block, follower) table for every merge; on the 88 k block
// instances its own -v / --allstages switches produce at k = 15 that is a minute of CPU time, after the device has found the
// blocks in a second.  oracle/build_dropin.sh links this definition in front of the reference's (whose symbol it weakens with
// objcopy; the rest of postprocessor.cpp -- constructor, ImproveBlockBoundaries, the alignment helpers -- stays the reference's).
#include <map>
#include <stdexcept>
#include <vector>

#include "postprocessor.h"                    // the reference's
#include "sibelia_amd.h"

namespace SyntenyFinder
{
	void Postprocessor::GlueStripes(std::vector<BlockInstance> & block)
	{
		std::vector<sbl_block> flat(block.size());
		std::map<size_t, const FASTARecord *> record;
		uint32_t chrCount = 0;
		for (size_t i = 0; i < block.size(); i++)
		{
			flat[i].id = block[i].GetSignedBlockId();
			flat[i].chr = static_cast<uint32_t>(block[i].GetChrId());
			flat[i].start = block[i].GetStart();
			flat[i].end = block[i].GetEnd();
			record[block[i].GetChrId()] = &block[i].GetChrInstance();
			chrCount = std::max(chrCount, flat[i].chr + 1);
		}
		uint64_t n = flat.size();
		if (sbl_glue_stripes(flat.empty() ? 0 : &flat[0], &n, chrCount) != SBL_OK)
		{
			throw std::runtime_error("sibelia_amd: GlueStripes failed");
		}
		block.clear();
		for (uint64_t i = 0; i < n; i++)
		{
			block.push_back(BlockInstance(flat[i].id, record[flat[i].chr], flat[i].start, flat[i].end));
		}
	}
}
