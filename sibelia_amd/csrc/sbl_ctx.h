// sbl_ctx.h -- the context behind the C ABI (one per host thread, one GPU each).
#pragma once
#include "sbl_common.h"

struct sbl_ctx {
	int device = 0;
	unsigned stage_seq = 0;          // stages run on this context (rotates which launches carry an event pair, simplify.hip)
	hipStream_t stream = nullptr;
	std::string err;
	GlibcRand rng;
	bool tempfile_mode = false;          // sbl_set_tempfile_mode: draw the reference's TempFile names from the stream (see sbl_api.hip)

	// ---- state carried between stages (rawSeq_ / originalPos_, reference src/blockfinder.h:52-54),
	//      resident in HBM as the element array '$' c0 '$' c1 '$' ... '$'
	uint32_t nchr = 0;
	std::vector<uint32_t> sepidx;        // host copy: element index of the '$' before chromosome c (nchr+1)
	size_t nelem = 0;                    // E = L + nchr + 1
	DevBuf d_ch;                         // uint8  [E padded to 32 with '$']
	DevBuf d_op;                         // uint32 [E] original positions (29 bit)
	DevBuf d_sepidx;                     // uint32 [nchr+1]
	// characters the reference would replace through rand() at the next IndexedSequence::Init
	std::vector<uint32_t> amb_elem;      // element indices, chromosome-major order
	std::vector<uint8_t> amb_orig;       // their original characters
	DevBuf d_amb_elem, d_amb_char;
	DevBuf d_fa_text, d_fa_lines, d_fa_recs;   // sbl_load_fasta: file text, per-line and per-record tables
	std::vector<std::string> fa_names;   // record descriptions of the last sbl_load_fasta

	// the records as loaded (originalChrList_): GenerateSyntenyBlocks trims blocks on the ORIGINAL sequences (src/synteny.cpp:36-40)
	DevBuf d_orig_ch;
	std::vector<uint32_t> orig_sepidx;
	sbl_ctx *child = nullptr;            // index over block sequences (TrimBlocks), same device
	void *tiny_out = nullptr;            // mapped host buffer of the small-block index (synteny.hip), hipHostFree'd with the context

	// stage-boundary checkpoint (sbl_save_state / sbl_restore_state)
	DevBuf d_save_ch, d_save_op;
	size_t save_nelem = 0;
	std::vector<uint32_t> save_sepidx, save_amb_elem;
	std::vector<uint8_t> save_amb_orig;
	GlibcRand save_rng;
	bool saved = false;

	// host mirrors for sbl_get_state
	// (copy-back contract of the reference: blockfinder.cpp:85-95).  ONE pinned staging buffer for the whole element array, filled
	// by two bulk device-to-host copies (1 + 4 B per element); the per-chromosome pointers handed out are slices of it.
	bool host_state_valid = false;
	uint8_t *h_ch = nullptr;             // pinned [h_cap] (pageable when the pinned allocation fails: h_pinned)
	uint32_t *h_opos = nullptr;          // pinned [h_cap]
	size_t h_cap = 0;
	bool h_pinned = true;
	// the staging buffer it replaced: pointers handed out by sbl_get_state stay readable (stale, not dangling) for one more
	// generation -- a caller that still holds a slice across a stage + sbl_get_state reads old data instead of freed memory
	uint8_t *h_old_ch = nullptr; uint32_t *h_old_opos = nullptr; bool h_old_pinned = true;
	void host_free(void *p, bool pinned) { if (!p) return; if (pinned) (void)hipHostFree(p); else free(p); }
	void release_host_state()
	{
		host_free(h_ch, h_pinned); host_free(h_opos, h_pinned); host_free(h_old_ch, h_old_pinned); host_free(h_old_opos, h_old_pinned);
		h_ch = h_old_ch = nullptr; h_opos = h_old_opos = nullptr; h_cap = 0;
	}

	// ---- enumeration workspace
	DevBuf d_pk, d_sp;                   // packed bases / separator bits
	DevBuf d_rec_keys[2], d_rec_vals[2]; // k-mer records {mix64(canonical code), element | masks}: position order / partitioned by hash prefix
	DevBuf d_boff;                       // bucket offsets (2^bits + 1)
	DevBuf d_counters;                   // small uint32 scratch block
	DevBuf d_keys, d_payload, d_skeys, d_spayload, d_pairids, d_sorttmp;
	DevBuf d_bif[2];                     // dense marks, uint32 [element capacity]
	DevBuf d_chunkcnt, d_chunkoff, d_scantmp;
	DevBuf d_melem[2], d_mid[2];         // compacted marks per strand (element, id), ascending element
	uint32_t nmarks[2] = {0, 0};
	bool marks_compact_ready = false;    // the enumeration that just ran left d_melem / d_mid / nmarks itself (longk_fp.hip: few members): sbl_compact_marks has nothing to do
	uint32_t bif_count = 0;
	uint32_t cur_k = 0;
	const unsigned long long *dict_keys = nullptr;   // k <= 32: the sorted strand-specific bifurcation codes of the last enumeration (id = rank); nullptr for long k
	DevBuf d_inst;                       // marshalling buffer
	DevBuf d_edges, d_valid;             // sbl_list_edges staging

	// results handed out through the ABI
	std::vector<sbl_inst> inst[2];
	std::vector<sbl_edge> edges;
	std::vector<uint64_t> h_hashes;
	std::vector<sbl_block> blocks;
	std::string graph_text;              // sbl_serialize_graph
	std::string report[3];               // sbl_postprocess: blocks_coords.txt, genomes_permutations.txt, coverage_report.txt

	// ---- multi-GPU enumeration (shard.hip): attached communicator + exchange buffers
	struct SblComm *comm = nullptr;
	DevBuf d_send, d_recv, d_otable, d_oused, d_allkeys, d_allkeys2, d_gelem[2], d_gid[2], d_stage;

	struct LongKScratch *lk = nullptr;   // k > 32 workspace (longk.hip)
	struct LongKFpHolder *lkfp = nullptr;   // k > 32 through window fingerprints (longk_fp.hip)

	// ---- simplification workspace lives in simplify.hip (opaque here)
	struct SimplifyState *simp = nullptr;
	uint32_t window = 0;
	// what an abandoned attempt learnt, carried into the rerun and into later stages (simplify.hip: sbl_simplify_run): the element slack
	// and node capacity a pool overflow asked for, and "the previous stage needed a roll-back: take iteration checkpoints from the start"
	size_t hint_elem_slack = 0, hint_cap_n = 0;
	bool hint_checkpoints = false;

	sbl_stage_stats stats{};
	hipEvent_t ev[8] = {};
};

// every ABI entry point runs under this: no exception crosses the boundary
template <class F>
static sbl_status guarded(sbl_ctx *c, F f)
{
	if (!c) return SBL_ERR_BAD_ARG;
	try {
		(void)hipSetDevice(c->device);
		f();
		return SBL_OK;
	} catch (const SblError &e) {
		c->err = e.msg;
		return e.st;
	} catch (const std::bad_alloc &) {
		c->err = "host allocation failed";
		return SBL_ERR_OOM;
	} catch (...) {
		c->err = "unexpected exception";
		return SBL_ERR_INTERNAL;
	}
}

// implemented in sbl_api.hip
void sbl_pack(sbl_ctx *c);
void sbl_run_enumeration(sbl_ctx *c, uint32_t k, size_t elem_capacity);   // fills d_bif[0..1], bif_count
void sbl_compact_marks(sbl_ctx *c, int strand);
// implemented in fasta_load.hip
void sbl_finish_load(sbl_ctx *c);   // d_ch / sepidx in place: original positions + list of non-ACGT elements, on the device
// implemented in shard.hip
void sbl_run_enumeration_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity);   // k <= 32, c->comm attached: hash-prefix sharded table
void sbl_comm_release(sbl_ctx *c);
// implemented in longk.hip
void sbl_run_enumeration_longk(sbl_ctx *c, uint32_t k, size_t elem_capacity);   // k > 32: exact rank doubling
void sbl_run_enumeration_longk_sharded(sbl_ctx *c, uint32_t k, size_t elem_capacity);   // ... split over the GPUs of c->comm
void sbl_longk_free(sbl_ctx *c);
// implemented in longk_fp.hip
bool sbl_run_enumeration_longk_fp(sbl_ctx *c, uint32_t k, size_t elem_capacity);   // k > 32: window fingerprints + bucketed table + exact verification; false = a verification failed, run the doubling
void sbl_longk_fp_free(sbl_ctx *c);
// implemented in simplify.hip
void sbl_simplify_run(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges);
void sbl_simplify_free(sbl_ctx *c);
