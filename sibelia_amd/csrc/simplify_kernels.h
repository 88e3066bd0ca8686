// simplify_kernels.h -- the kernels of the simplification as the host driver (simplify.hip: DeviceBackend) launches them, and the
// constants both sides share.  Definitions: graphbuild.hip (graph construction, copy-back, block index), snapshot.hip, rounds.hip
// (probe, selection, reservation), commit.hip (transactions, serial chain, one-launch stage).
#pragma once
#include "sbl_ctx.h"
#include "simplify_steps.h"

#define CLAIM_CAP 4096u                      // ids a window entry can list; beyond that commit re-walks serially
#define PROBE_WAVES 1u                       // waves per probed id (windows dealt out to them, wave 0 takes the verdict); more than one did not pay: most entries are cheap
#define PROBE_UNSERVED 3u                    // live[] value: k_probe_idx could not serve the entry, the walking probe decides
#define SEL_THREADS 256
#define RSV_WAVES_MAX 4u
struct MarkStream { const unsigned *elem[2], *id[2], *aux[2]; unsigned n[2]; };

// debugging / measurement counters that live beside the kernels that bump them (device globals are per translation unit)
void sbl_commit_prof_reset();
void sbl_commit_prof_report(unsigned ts_round);                       // SBL_PHASES=1: prints the phase cycle counters of k_commit
void sbl_rounds_stats_report();                                      // SBL_TEST_FLAGS=32: what the block index served, reservation phase ticks

__global__ void __launch_bounds__(256) k_init_links(unsigned *__restrict__ nx, unsigned *__restrict__ pv, unsigned *__restrict__ nodeof0,
                                                    unsigned *__restrict__ nodeof1, uint8_t *__restrict__ ch, size_t E, size_t cap);
__global__ void __launch_bounds__(256) k_instance_keys(const unsigned *__restrict__ elem, const unsigned *__restrict__ id, unsigned n, unsigned strand,
                                                       const unsigned *__restrict__ sepidx, unsigned nchr, unsigned E, unsigned ordbits,
                                                       unsigned long long *__restrict__ keys, unsigned *__restrict__ midx);
__global__ void __launch_bounds__(256) k_build_lists(const unsigned long long *__restrict__ skeys, const unsigned *__restrict__ smidx, const unsigned *__restrict__ melem, unsigned n,
                                                     unsigned node_base, unsigned strand, unsigned ordbits, unsigned *__restrict__ nslot, unsigned *__restrict__ nnext, unsigned *__restrict__ nidst,
                                                     uint8_t *__restrict__ ndead, unsigned *__restrict__ head, unsigned *__restrict__ lsize,
                                                     unsigned *__restrict__ nodeof, unsigned *__restrict__ nmark);
__global__ void __launch_bounds__(256) k_id_position_keys(const unsigned *__restrict__ head0, const unsigned *__restrict__ head1, const unsigned *__restrict__ nslot,
                                                          unsigned nid, unsigned long long *__restrict__ keys, unsigned *__restrict__ ids);
__global__ void __launch_bounds__(256) k_max_instances(const unsigned *__restrict__ l0, const unsigned *__restrict__ l1, unsigned nid, unsigned *__restrict__ out);
__global__ void __launch_bounds__(256) k_mark_aux(const unsigned *__restrict__ melem, unsigned n, unsigned strand, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                  const uint8_t *__restrict__ ch, unsigned k, unsigned *__restrict__ aux);
__global__ void __launch_bounds__(64) k_snapshot_first(GraphView g, MarkStream ms, const unsigned *__restrict__ nmark, const unsigned *__restrict__ perm, unsigned plo, unsigned phi);
__global__ void __launch_bounds__(256) k_lin_positions(const uint8_t *__restrict__ ch, unsigned ne, const unsigned *__restrict__ flag, const unsigned *__restrict__ segidx,
                                                       const unsigned *__restrict__ seg_head, const unsigned long long *__restrict__ dist, unsigned long long total,
                                                       unsigned *__restrict__ lin, unsigned *__restrict__ elin);
__global__ void __launch_bounds__(256) k_count_marks_lin(const unsigned *__restrict__ bif, const unsigned *__restrict__ elin, size_t n, unsigned *__restrict__ chunkcnt);
__global__ void __launch_bounds__(256) k_write_marks_lin(const unsigned *__restrict__ bif, const unsigned *__restrict__ elin, const unsigned *__restrict__ nodeof, size_t n,
                                                         const unsigned *__restrict__ chunkoff, unsigned *__restrict__ out_pos, unsigned *__restrict__ out_id, unsigned *__restrict__ nmark);
__global__ void __launch_bounds__(256) k_mark_aux_lin(const unsigned *__restrict__ mpos, unsigned n, unsigned strand, const unsigned *__restrict__ sepelem, unsigned nchr,
                                                      const unsigned *__restrict__ lin, const unsigned *__restrict__ elin, const uint8_t *__restrict__ ch, unsigned k, unsigned *__restrict__ aux);
__global__ void __launch_bounds__(64) k_snapshot_stream(GraphView g, MarkStream ms, const unsigned *__restrict__ nmark, const unsigned *__restrict__ perm, int incremental, unsigned plo, unsigned phi);
__global__ void __launch_bounds__(64) k_snapshot(GraphView g, uint8_t *arena, unsigned arena_bytes, int incremental, const unsigned *__restrict__ perm, unsigned plo, unsigned phi);
__global__ void __launch_bounds__(64) k_probe_idx(GraphView g, unsigned nwin, uint8_t *live, unsigned w0, unsigned vbits, unsigned max_inst, unsigned walk_marks,
                                                  unsigned *__restrict__ instbuf, unsigned istride, int snapshot);
__global__ void __launch_bounds__(64 * PROBE_WAVES) k_probe(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, uint8_t *live, unsigned w0, int snapshot);
__global__ void __launch_bounds__(256) k_pack_need(const unsigned *__restrict__ perm, const uint8_t *__restrict__ need, unsigned lo, unsigned hi, uint8_t *__restrict__ buf);
__global__ void __launch_bounds__(256) k_unpack_need(const unsigned *__restrict__ perm, const uint8_t *__restrict__ buf, unsigned n, unsigned mylo, unsigned myhi, uint8_t *__restrict__ need);
__global__ void __launch_bounds__(256) k_apply_probe(GraphView g, unsigned nwin, uint8_t *__restrict__ live, unsigned w0, unsigned w1, const uint8_t *__restrict__ robuf, unsigned stride, unsigned nranks);
__global__ void __launch_bounds__(SEL_THREADS) k_select_count(GraphView g, unsigned *__restrict__ sel, unsigned lo, unsigned limit, unsigned chunk0, unsigned chunk,
                                                              const uint8_t *__restrict__ live, unsigned probed);
__global__ void __launch_bounds__(SEL_THREADS) k_select_write(GraphView g, unsigned *__restrict__ sel, unsigned *__restrict__ win, unsigned lo, unsigned limit, unsigned W,
                                                              unsigned chunk0, unsigned chunk, unsigned nchunks, volatile unsigned *post, unsigned post_seq);
__global__ void __launch_bounds__(256) k_pack_probe(const unsigned *__restrict__ ctr, const uint8_t *__restrict__ live, unsigned w0, unsigned w1, unsigned rank, unsigned stride, uint8_t *__restrict__ robuf);
__global__ void __launch_bounds__(64 * RSV_WAVES_MAX) k_reserve(GraphView g, unsigned nwin, unsigned *claims, const uint8_t *live, unsigned seen_bits, unsigned list_cap,
                                                                 const unsigned *__restrict__ instbuf, unsigned istride, const uint8_t *arena, unsigned arena_bytes);
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_commit(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, int solo, const unsigned *claims, const uint8_t *live, int prof);
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 4))) k_resume(GraphView g, unsigned nwin, uint8_t *arena, unsigned arena_bytes, const unsigned *claims, const uint8_t *live, int prof);
__global__ void __launch_bounds__(256) k_park_sweep(unsigned *__restrict__ park_of, unsigned n);
__global__ void __launch_bounds__(64) k_chain(GraphView g, uint8_t *arena, unsigned arena_bytes, unsigned nwin, int prof);
__global__ void __launch_bounds__(64) k_dense_stage(GraphView g, uint8_t *arena, unsigned arena_bytes, unsigned max_iter, unsigned *out);
__global__ void __launch_bounds__(256) k_count_touched(const uint8_t *__restrict__ touch, unsigned nid, unsigned *__restrict__ out);
__global__ void __launch_bounds__(256) k_touched_list(const uint8_t *__restrict__ touch, unsigned nid, unsigned *__restrict__ list, unsigned *__restrict__ count);
__global__ void __launch_bounds__(256) k_seg_flags(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, unsigned ne, unsigned *__restrict__ flag);
__global__ void __launch_bounds__(256) k_seg_tails(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, unsigned ne,
                                                   const unsigned *__restrict__ flag, const unsigned *__restrict__ segidx /* exclusive scan */,
                                                   unsigned *__restrict__ seg_head, unsigned *__restrict__ seg_len, unsigned *__restrict__ seg_succ_elem);
__global__ void __launch_bounds__(256) k_seg_finish(unsigned nseg, const unsigned *__restrict__ seg_head, unsigned *__restrict__ seg_len,
                                                    const unsigned *__restrict__ seg_succ_elem, const unsigned *__restrict__ flag,
                                                    const unsigned *__restrict__ segidx, unsigned *__restrict__ succ, unsigned long long *__restrict__ dist);
__global__ void __launch_bounds__(256) k_seg_jump(unsigned nseg, const unsigned *__restrict__ succ_in, const unsigned long long *__restrict__ dist_in,
                                                  unsigned *__restrict__ succ_out, unsigned long long *__restrict__ dist_out);
__global__ void __launch_bounds__(256) k_scatter_linear(const uint8_t *__restrict__ ch, const unsigned *__restrict__ op, unsigned ne,
                                                        const unsigned *__restrict__ flag, const unsigned *__restrict__ segidx,
                                                        const unsigned *__restrict__ seg_head, const unsigned long long *__restrict__ dist,
                                                        unsigned long long total, uint8_t *__restrict__ ch_out, unsigned *__restrict__ op_out,
                                                        unsigned *__restrict__ newidx);
__global__ void k_remap_seps(const unsigned *__restrict__ newidx, unsigned *__restrict__ sepidx, unsigned n);
__global__ void k_sep_positions(const unsigned *__restrict__ sepidx, unsigned nchr, unsigned *__restrict__ op);
__global__ void __launch_bounds__(256) k_dict_check(const uint8_t *__restrict__ ch, unsigned ne, const unsigned *__restrict__ newidx, const uint8_t *__restrict__ ch_out, unsigned long long total,
                                                    const unsigned *__restrict__ bif0, const unsigned *__restrict__ bif1, const unsigned long long *__restrict__ dict, unsigned nd, unsigned k,
                                                    unsigned long long *__restrict__ out);
__global__ void __launch_bounds__(256) k_fill_bytes(uint8_t *p, uint8_t v, size_t from, size_t to);
__global__ void __launch_bounds__(256) k_build_blkidx(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, const unsigned *__restrict__ pv,
                                                      const unsigned *__restrict__ bif0, const unsigned *__restrict__ bif1, unsigned norig, unsigned nblk,
                                                      unsigned long long *__restrict__ bidx);
__global__ void __launch_bounds__(256) k_check_blkidx(const uint8_t *__restrict__ ch, const unsigned *__restrict__ nx, const unsigned *__restrict__ pv,
                                                      const unsigned *__restrict__ bif0, const unsigned *__restrict__ bif1, const unsigned *__restrict__ wmax, unsigned norig, unsigned nblk,
                                                      const unsigned long long *__restrict__ bidx, unsigned *__restrict__ out);
__global__ void __launch_bounds__(256) k_clear_counters(unsigned *__restrict__ ctr);
__global__ void __launch_bounds__(256) k_idx_clear_stamps(unsigned long long *__restrict__ bidx, unsigned nblk);
