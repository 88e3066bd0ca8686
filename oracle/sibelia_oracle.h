/*
 * sibelia_oracle.h -- CPU ORACLE for the BlockFinder hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference algorithm (bioinf/Sibelia 3.0.7) for
 *   IndexedSequence::Init / EnumerateBifurcationsSArrayInRAM   (src/indexedsequence.cpp:28-72,
 *                                                               src/vertexenumeration.cpp:263-364)
 *   BlockFinder::SimplifyGraph / RemoveBulges / CollapseBulgeGreedily
 *                                                              (src/blockfinder.cpp:16-51,
 *                                                               src/bulgeremoval.cpp:39-430)
 *   DNASequence::Replace position interpolation                (src/dnasequence.cpp:189-252)
 *   BlockFinder::ListEdges / SerializeCondensedGraph           (src/serialization.cpp:56-110)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * library, and only as the checker.  The product (sibelia_amd/csrc) never links,
 * loads or calls it.
 *
 * Parity status: PINNED -- checked bit-for-bit against outputs of the unmodified
 * reference binary on the vectors in tests/golden/vectors.json
 * (tests/test_oracle_golden.py; generator tests/golden/gen/).
 */
#ifndef SIBELIA_ORACLE_H
#define SIBELIA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

typedef struct { uint32_t id, chr, pos; } orc_inst;   /* reference BifurcationInstance, src/indexedsequence.h:57-68 */

typedef struct {
	uint32_t chr, strand;        /* strand 0 = positive ("blue"), 1 = negative ("red") */
	uint32_t start_vertex, end_vertex;
	uint32_t pos, len;           /* actual position / length (conventional + coordinates) */
	uint32_t orig_pos, orig_len; /* SpellOriginal, src/dnasequence.cpp:254-260 */
	char first_char;
} orc_edge;

orc_ctx *orc_create(void);
void orc_destroy(orc_ctx *c);

/* BlockFinder::Init (src/blockfinder.cpp:65-76): copies sequences, originalPos = identity.
 * Sequences must already be upper-case (what the reference FASTA reader delivers). */
int orc_load(orc_ctx *c, uint32_t nchr, const uint8_t *const *seq, const uint64_t *len);

/* Enumeration of the CURRENT state at vertex size k, as a fresh IndexedSequence would see it
 * (sanitises a copy through the ctx's glibc-compatible rand()).  Instance arrays are owned
 * by the ctx and valid until the next call. */
int orc_enumerate(orc_ctx *c, uint32_t k, uint32_t *bif_count,
                  const orc_inst **pos, uint64_t *npos, const orc_inst **neg, uint64_t *nneg);

/* BlockFinder::PerformGraphSimplifications (src/blockfinder.cpp:78-98). */
int orc_simplify_stage(orc_ctx *c, uint32_t k, uint32_t min_branch, uint32_t max_iter, uint64_t *bulges);

/* rawSeq_[chr], originalPos_[chr] (borrowed pointers, valid until the next mutating call). */
int orc_get_state(orc_ctx *c, uint32_t chr, const uint8_t **seq, const uint32_t **orig_pos, uint64_t *len);
uint32_t orc_nchr(const orc_ctx *c);

/* BlockFinder::ListEdges on a fresh IndexedSequence at k (src/serialization.cpp:56-86). */
int orc_list_edges(orc_ctx *c, uint32_t k, const orc_edge **edges, uint64_t *n);

/* H0: k-mer hashes of the reference's hashing.h (src/hashing.h:14-112) over the current state, both strands; malloc'd, orc_free. */
int orc_kmer_hashes(orc_ctx *c, uint32_t k, uint64_t **out, uint64_t *n);
void orc_free(void *p);

/* N2: BlockFinder::GenerateSyntenyBlocks (src/synteny.cpp:229-286; synteny_oracle.cpp).  orig_seq / orig_len: the records the
 * BlockFinder was built from (originalChrList_).  out: malloc'd (orc_free), sorted like the reference's result. */
typedef struct { int32_t id; uint32_t chr; uint64_t start, end; } orc_block;      /* BlockInstance: signed block id, chr, [start, end) */
int orc_generate_blocks(orc_ctx *c, const uint8_t *const *orig_seq, const uint64_t *orig_len, uint32_t k, uint32_t trimK, uint32_t minSize,
                        int sharedOnly, orc_block **out, uint64_t *n);
void orc_rng_copy(orc_ctx *dst, const orc_ctx *src);
/* N4: Postprocessor::GlueStripes (src/postprocessor.cpp:37-154) + the writers of blocks_coords.txt, genomes_permutations.txt and
 * coverage_report.txt (src/outputgenerator.cpp:162-233; output_oracle.cpp).  Everything returned is malloc'd (orc_free). */
int orc_postprocess(const orc_block *in, uint64_t n, uint32_t nchr, const char *const *names, const uint64_t *sizes, int glue,
                    orc_block **out_blocks, uint64_t *nout, char **texts /* 3 */, uint64_t *text_len /* 3 */);   /* a child index shares the parent's rand() stream */

/* BlockFinder::SerializeGraph (src/serialization.cpp:112-138): DOT text of the uncondensed graph; malloc'd, orc_free. */
int orc_serialize_graph(orc_ctx *c, uint32_t k, char **out, uint64_t *n);

/* test hooks */
void orc_force_long_k_path(orc_ctx *c, int on);      /* use the rank-doubling grouping even for k <= 32 */
uint32_t orc_rand(orc_ctx *c);                       /* next value of the ctx's glibc rand() stream   */
/* Boost 1.54 unordered_map<size_t,...> iteration order for a key insertion sequence (test hook). */
size_t orc_boost_order(const uint64_t *keys, size_t n, uint64_t *out);

/* timing of the last orc_simplify_stage: seconds spent in enumerate+mark, simplify, copy-back */
void orc_last_timing(const orc_ctx *c, double *enumerate_s, double *simplify_s, double *copyback_s);

#ifdef __cplusplus
}
#endif
#endif
