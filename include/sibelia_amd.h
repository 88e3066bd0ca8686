/*
 * sibelia_amd.h -- C ABI of the MI355X-native BlockFinder hot path (libsibelia_amd.so).
 *
 * Drop-in boundary for the reference's SyntenyFinder::BlockFinder (bioinf/Sibelia 3.0.7,
 * src/blockfinder.h:28-45).  The reference has no FFI layer; these entry points are what a
 * C++ maintainer binds BlockFinder's methods to (see INTEGRATION.md and
 * include/sibelia_amd/blockfinder.hpp, which restores the reference's class surface on top).
 *
 * Plain pointers and sizes only; no exceptions cross the boundary; every function returns an
 * sbl_status.  One context per host thread; a context owns one GPU (HIP device) and its own
 * glibc-compatible rand() stream (the reference consumes the process-global rand(),
 * src/indexedsequence.cpp:35).
 *
 * All compute runs in HIP kernels on the context's device.  There is no CPU fallback:
 * without a usable gfx950 device sbl_create fails with SBL_ERR_NO_DEVICE.
 */
#ifndef SIBELIA_AMD_H
#define SIBELIA_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbl_ctx sbl_ctx;

typedef enum {
	SBL_OK = 0,
	SBL_ERR_BAD_ARG = 1,       /* k < 2 (stage files enforce k >= 2, src/util.cpp:38-41), null pointers, ... */
	SBL_ERR_NO_DEVICE = 2,     /* no usable HIP device: the product never computes on the host */
	SBL_ERR_OOM = 3,
	SBL_ERR_HIP = 4,           /* HIP runtime error, see sbl_last_error */
	SBL_ERR_TOO_LARGE = 5,     /* a chromosome >= 2^29 bp or total input > 2^30 (src/stranditerator.cpp:19-27, src/common.h:52) */
	SBL_ERR_UNSUPPORTED = 6,   /* vertex size not supported by this build */
	SBL_ERR_INTERNAL = 7
} sbl_status;

/* BlockFinder::State + ProgressCallBack (src/blockfinder.h:31-39): called on the calling thread. */
typedef enum { SBL_PROGRESS_START = 0, SBL_PROGRESS_RUN = 1, SBL_PROGRESS_END = 2 } sbl_progress_state;
typedef void (*sbl_progress_fn)(size_t progress, int state, void *user);

/* BifurcationInstance (src/indexedsequence.h:57-68).  Negative-strand `pos` is in reverse-complement
 * coordinates, exactly as EnumerateBifurcationsSArrayInRAM reports it (src/vertexenumeration.cpp:334-346). */
typedef struct { uint32_t id, chr, pos; } sbl_inst;

/* BlockFinder::Edge (src/blockfinder.h:58-90) as produced by ListEdges (src/serialization.cpp:56-86). */
typedef struct {
	uint32_t chr, strand;            /* strand 0 = positive, 1 = negative */
	uint32_t start_vertex, end_vertex;
	uint32_t pos, len;               /* actual position (+ coordinates) and length */
	uint32_t orig_pos, orig_len;     /* DNASequence::SpellOriginal (src/dnasequence.cpp:254-260) */
	char first_char;
	char pad_[3];
} sbl_edge;

/* Per-stage counters / timings of the last sbl_simplify_stage (times measured on the device). */
typedef struct {
	uint64_t strand_kmers;           /* N = 2 * sum(max(0, len - k + 1)), the metric's unit */
	uint64_t bif_count, instances;   /* after enumeration */
	uint64_t bulges;                 /* return value of PerformGraphSimplifications */
	uint32_t iterations, rounds;     /* SimplifyGraph iterations run; ordered-commit rounds launched */
	uint32_t replays;                /* iterations re-run (order validation fired, or a pool had to grow), + 1 for an abandoned checkpoint-free first attempt */
	uint32_t grow_replays;           /* ... of which because the element / node pool had to grow */
	double enumerate_ms, simplify_ms, copyback_ms, total_ms;
	double kmer_table_ms;            /* duration of the dominant kernel (k-mer table build) */
	uint64_t kmer_table_bytes;       /* its algorithmic HBM bytes */
	double snapshot_ms, reserve_ms, commit_ms;   /* SimplifyGraph kernel time by phase */
	double probe_ms;
	uint64_t executed;               /* pending ids examined in ordered rounds (retired by the probe or committed) */
	uint64_t transactions;           /* ... of which RemoveBulges transactions that owned their neighbourhood and ran */
	double exchange_ms;              /* sharded enumeration: host time inside the collectives (all-to-all + gathers) */
	uint64_t exchange_bytes;         /* ... and the bytes this GPU sent to its peers */
	uint64_t chain_transactions;     /* transactions run by the serial chain (dense conflict neighbourhoods: small k, low complexity) */
	/* probe_ms / reserve_ms / commit_ms come from start stamps the round kernels write themselves (device wall clock, every launch);
	 * HIP event pairs are recorded around every 4th launch of the commit kernel only (an event pair costs ~8 us of barrier packets): */
	double commit_event_ms;          /* sum of those event-pair times */
	uint64_t commit_event_launches;  /* ... and how many launches they cover */
	/* SBL_CHECK_DICTIONARY=1 (k <= 32): the reference's _DEBUG invariant IndexedSequence::Test (src/indexedsequence.cpp:74-103) on the
	 * stage's final graph -- windows whose stored mark was compared with the dictionary of the initial marking, and how many differed
	 * (a stage with mismatches fails with SBL_ERR_INTERNAL) */
	uint64_t dict_checked, dict_mismatches;
	/* read-only simplification phases split over the attached GPUs (snapshots by id range, probes by window share; commits replicated):
	 * GPUs sharing them (1 = not split), host time inside the verdict all-gathers, bytes this GPU sent (1 B per id per snapshot,
	 * 1 B per window entry per round) */
	uint64_t ro_ranks;
	double verdict_ms;
	uint64_t verdict_bytes;
	/* k > 32 (round 6): which enumeration ran -- 0 none (k <= 32), 1 window fingerprints + bucketed table + exact verification
	 * (longk_fp.hip), 2 exact rank doubling on request / on several GPUs (longk.hip), 3 rank doubling after a failed verification --
	 * and the occurrences of bifurcation k-mers that were compared with their group's representative on the sequence (k/4 B each) */
	uint64_t longk_path;
	uint64_t fp_verified;
	/* device memory held by the library's buffers of this PROCESS when the stage ended (grow-only workspaces: the peak so far) */
	uint64_t device_bytes;
} sbl_stage_stats;

/* Replaces: BlockFinder::BlockFinder(chrList[, tempDir]) + Init (src/blockfinder.cpp:53-76).
 * device < 0 selects the current HIP device. */
sbl_status sbl_create(sbl_ctx **out, int device);
void sbl_destroy(sbl_ctx *ctx);

/* Replaces BlockFinder::Init (src/blockfinder.cpp:65-76): sequences are upper-case ASCII as
 * delivered by the reference FASTA reader (src/fasta.cpp:92-104); originalPos = identity.
 * Uploads the state to HBM; inputs are borrowed for the call only. */
sbl_status sbl_load(sbl_ctx *ctx, uint32_t nchr, const uint8_t *const *seq, const uint64_t *len);

/* Upstream of the hot path: replaces FASTAReader::GetSequences (src/fasta.cpp:23-104) + BlockFinder::Init.
 * The file is mapped and copied to the device as text; line splitting, trimming, header / sequence classification,
 * upper-casing, validation ("ACGTURYKMSWBDHWNX-"), concatenation, identity original positions and the scan for non-ACGT
 * characters run in kernels.  Parse errors come back as SBL_ERR_BAD_ARG with the reference's message
 * ("parse error in <file> on line <n>: empty sequence | empty header | illegal character: <c>") in sbl_last_error.
 * sbl_record_name: FASTARecord::GetDescription (text between '>' and the first blank), valid until the next load. */
sbl_status sbl_load_fasta(sbl_ctx *ctx, const char *path);
const char *sbl_record_name(const sbl_ctx *ctx, uint32_t chr);

/* Replaces IndexedSequence::Init's enumeration (src/indexedsequence.cpp:28-47 ->
 * src/vertexenumeration.cpp:263-364) on the current state at vertex size k.
 * Arrays are owned by the ctx, sorted by (chr,pos), valid until the next call on the ctx. */
sbl_status sbl_enumerate(sbl_ctx *ctx, uint32_t k, uint32_t *bif_count,
                         const sbl_inst **pos, uint64_t *npos, const sbl_inst **neg, uint64_t *nneg);

/* Replaces BlockFinder::PerformGraphSimplifications (src/blockfinder.cpp:78-98): enumeration,
 * marking, SimplifyGraph (bulge removal, src/blockfinder.cpp:16-51, src/bulgeremoval.cpp) and
 * copy-back, all on the device; the state stays resident in HBM. */
sbl_status sbl_simplify_stage(sbl_ctx *ctx, uint32_t k, uint32_t min_branch_size, uint32_t max_iterations,
                              sbl_progress_fn progress, void *user, uint64_t *bulges);

/* rawSeq_[chr] / originalPos_[chr] (src/blockfinder.h:52-54), downloaded on demand.
 * Borrowed pointers, valid until the next mutating call. */
sbl_status sbl_get_state(sbl_ctx *ctx, uint32_t chr, const uint8_t **seq, const uint32_t **orig_pos, uint64_t *len);
uint32_t sbl_nchr(const sbl_ctx *ctx);

/* Replaces BlockFinder::ListEdges on a fresh index at k (src/serialization.cpp:56-86), the
 * observation channel behind SerializeCondensedGraph (src/serialization.cpp:88-110). */
sbl_status sbl_list_edges(sbl_ctx *ctx, uint32_t k, const sbl_edge **edges, uint64_t *n);

/* Downstream of the hot path: replaces BlockFinder::GenerateSyntenyBlocks (src/synteny.cpp:229-286, with
 * ResolveOverlap :124-166 and TrimBlocks :31-122; src/blockfinder.h:43) on the current state.  Both indices it needs -- the edge
 * list at k and, per candidate block, a fresh index at trim_k over the block's ORIGINAL sequences (kept on the device since
 * sbl_load / sbl_load_fasta) -- are built by the enumeration kernels; rand() is consumed exactly as the reference does.
 * sbl_block = BlockInstance (src/blockinstance.h:21-47): signed block id (sign = strand), chromosome, [start, end) in original
 * coordinates; sorted by (chr, start) like the reference's result.  Owned by the ctx, valid until the next call. */
typedef struct { int32_t id; uint32_t chr; uint64_t start, end; } sbl_block;
sbl_status sbl_generate_blocks(sbl_ctx *ctx, uint32_t k, uint32_t trim_k, uint32_t min_size, int shared_only,
                               const sbl_block **blocks, uint64_t *n);

/* what the reference's main runs after GenerateSyntenyBlocks (src/sibelia.cpp:287-315) on the blocks of the last
 * sbl_generate_blocks: Postprocessor::GlueStripes (src/postprocessor.cpp:37-154; skipped when glue == 0) and the texts of
 * blocks_coords.txt (OutputGenerator::ListBlocksIndices, src/outputgenerator.cpp:227-233), genomes_permutations.txt
 * (ListChromosomesAsPermutations, :203-219) and coverage_report.txt (GenerateReport, :162-201), byte for byte.
 * names: record descriptions, EXACTLY sbl_nchr(ctx) pointers (NULL: those of the last sbl_load_fasta).  Host-side bookkeeping and formatting only.
 * Everything returned is owned by the ctx and valid until the next call. */
sbl_status sbl_postprocess(sbl_ctx *ctx, int glue, const char *const *names, const sbl_block **blocks, uint64_t *n,
                           const char **blocks_coords, const char **genomes_permutations, const char **coverage_report);

/* Postprocessor::GlueStripes (src/postprocessor.cpp:37-154) on a caller's block list, in place (*n updated; never grows): the
 * reference's main applies it to the blocks of every stage under -v / --allstages (src/sibelia.cpp:247-253).  Same merges in the same
 * order as the reference, found by a worklist instead of one rescan per merge (88 k instances: 0.04 s instead of a minute).
 * Host bookkeeping only: needs neither a context nor a device. */
sbl_status sbl_glue_stripes(sbl_block *blocks, uint64_t *n, uint32_t nchr);

/* Replaces BlockFinder::SerializeGraph (src/serialization.cpp:112-138; defined for records of at least k + 1 characters -- the
 * reference walks off the end of a shorter one): DOT text of the UNcondensed de Bruijn graph of the
 * current state, one line per (k+1)-window, generated on the device (a debugging dump: main only reaches it with -q and never
 * with production options).  Owned by the ctx, valid until the next call. */
sbl_status sbl_serialize_graph(sbl_ctx *ctx, uint32_t k, const char **text, uint64_t *len);

/* H0: the k-mer hash of the reference's hashing.h (SlidingWindow / KMerHashFunction, src/hashing.h:14-112; HASH_BASE 57,
 * arithmetic mod 2^64) for every k-mer of the current state: strand 0 then strand 1 (complemented characters, walk order),
 * chromosomes ascending.  The reference's production path never executes it; provided with a known-answer test.
 * Array owned by the ctx, valid until the next call. */
sbl_status sbl_kmer_hashes(sbl_ctx *ctx, uint32_t k, const uint64_t **values, uint64_t *n);

/* The reference's rand() is process-global: besides sanitising ambiguous bases (src/indexedsequence.cpp:31-37) it names the two
 * temporary files every index built WITHOUT -r spills its suffix array to (src/platform.cpp:52-58 via src/vertexenumeration.cpp:101,125:
 * 24 draws per index).  A context owns its stream; sbl_set_tempfile_mode(ctx, 1) makes every full-state index (sbl_simplify_stage,
 * sbl_list_edges, sbl_generate_blocks' main index, sbl_enumerate) draw those 24 values after its sanitising draws, as
 * BlockFinder(chrList, tempDir) does -- nothing is spilled, only the stream stays in step.  sbl_rand_advance skips n values, for a
 * host whose surrounding code consumes the same stream in other places. */
sbl_status sbl_set_tempfile_mode(sbl_ctx *ctx, int on);
sbl_status sbl_rand_advance(sbl_ctx *ctx, uint64_t n);

/* Stage-boundary checkpoint of the resident state (sequences + original positions), device to device.
 * The reference keeps no resumable state; the stage boundary is the natural one. */
sbl_status sbl_save_state(sbl_ctx *ctx);
sbl_status sbl_restore_state(sbl_ctx *ctx);

sbl_status sbl_last_stats(const sbl_ctx *ctx, sbl_stage_stats *out);
const char *sbl_last_error(const sbl_ctx *ctx);
const char *sbl_strerror(sbl_status s);

/* ---- Multi-GPU (one context per GPU).  The reference is a single-threaded CPU program with no
 * counterpart; these entry points attach a communicator to a context, after which the enumeration inside
 * sbl_enumerate / sbl_simplify_stage / sbl_list_edges shards the k-mer table by hash prefix (k <= 32; k > 32: see sbl_longk_* below):
 * every GPU turns its contiguous slice of base positions into 16-B k-mer records (one per position, no local
 * pre-aggregation), partitions them by hash prefix and sends every owner GPU its contiguous bucket range in ONE
 * all-to-all; owners classify their buckets (LDS tables), the bifurcation codes and the member marks are all-gathered.  Simplification is globally ordered and runs replicated (bit-identical) on
 * every attached GPU.  The calls are collective: every attached context must make them with the same arguments.
 *   RCCL transport (one process or thread per GPU, xGMI): rank 0 calls sbl_comm_unique_id, the host distributes the
 *   128 bytes (MPI, torch.distributed, a file), every rank calls sbl_comm_attach_rccl.
 *   Local transport: contexts of ONE process driven by one host thread each (device-to-device copies);
 *   used by the tests to run several virtual ranks on a single GPU. */
#define SBL_COMM_ID_BYTES 128
typedef struct sbl_group sbl_group;
sbl_status sbl_comm_unique_id(void *id /* SBL_COMM_ID_BYTES */);
sbl_status sbl_comm_attach_rccl(sbl_ctx *ctx, uint32_t rank, uint32_t nranks, const void *id);
sbl_group *sbl_group_create_local(uint32_t nranks);
void sbl_group_destroy(sbl_group *group);
sbl_status sbl_comm_attach_local(sbl_ctx *ctx, sbl_group *group, uint32_t rank);
sbl_status sbl_comm_detach(sbl_ctx *ctx);

/* The layout arithmetic of the sharded table, device-free (what the pipeline itself uses; for tests and for a host that wants to
 * size its buffers): tiles scanned by `rank`, first bucket of every owner (owner(b) = (b * nranks) >> bits), and -- given the
 * all-gathered count matrix count[p * nranks + q] = records p holds for owner q, and where the owners' ranges start in this
 * rank's partitioned arrays -- the byte counts / offsets of the one all-to-all. */
sbl_status sbl_shard_layout(uint32_t nranks, uint32_t rank, uint32_t bits, uint64_t ntiles, uint32_t *first_bucket /* nranks + 1 */, uint64_t *tile_range /* 2 */);
sbl_status sbl_shard_exchange_plan(uint32_t nranks, uint32_t rank, const uint64_t *count, const uint32_t *send_at /* nranks + 1 */, uint64_t record_bytes,
                                   uint64_t *sbytes, uint64_t *soff, uint64_t *rbytes, uint64_t *roff, uint64_t *nrecv);

/* k > 32 with a communicator attached: the exact rank doubling of ONE job is split over the GPUs (csrc/longk.hip, "sharded rank
 * doubling"; replaces the single-threaded suffix array of EnumerateBifurcationsSArrayInRAM, src/vertexenumeration.cpp:263-364, for
 * BASELINE.json's config 5).  Two partitions of the suffixes of the superGenome S (np = 2E - 1 + k positions) and one exchange
 * between them per doubling round: the POSITION side (rank r owns S[first[r], first[r + 1]) and a halo of H ranks behind it) and the
 * SORTED side (rank q owns an interval of the global sorted order; owner = binary search in its bounds).  The layout arithmetic,
 * device-free, as the pipeline itself calls it:
 *   sbl_longk_slices        first[r] = np * r / nranks
 *   sbl_longk_value_bounds  equal parts of the value range [0, maxvalue] (round 1 routes by the base-5 value of 8 symbols)
 *   sbl_longk_owner         largest q < nranks with bounds[q] <= x (position owner: bounds = first; sorted-side owner: bounds = G)
 *   sbl_longk_halo_plan     byte counts / offsets (4-B ranks) of the halo fetch: what `rank` sends to every peer out of its slice,
 *                           what it receives from every peer into its halo of H positions. */
sbl_status sbl_longk_slices(uint32_t nranks, uint64_t np, uint64_t *first /* nranks + 1 */);
sbl_status sbl_longk_value_bounds(uint32_t nranks, uint64_t maxvalue, uint64_t *bounds /* nranks + 1 */);
sbl_status sbl_longk_owner(uint32_t nranks, const uint64_t *bounds /* nranks + 1 */, uint64_t x, uint32_t *owner);
sbl_status sbl_longk_halo_plan(uint32_t nranks, uint32_t rank, uint64_t np, uint64_t H, uint64_t *sbytes, uint64_t *soff, uint64_t *rbytes, uint64_t *roff);

/* Tuning knob (0 = default): number of bifurcation ids speculatively committed per ordered round. */
sbl_status sbl_set_window(sbl_ctx *ctx, uint32_t window);

#ifdef __cplusplus
}
#endif
#endif
