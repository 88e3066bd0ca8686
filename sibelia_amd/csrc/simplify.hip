// simplify.hip -- BlockFinder::PerformGraphSimplifications (reference src/blockfinder.cpp:78-98) on the GPU:
//   enumeration (sbl_api.hip) -> instance lists (E2) -> SimplifyGraph rounds (simplify_steps.h) -> copy-back (T3).
// Everything that touches sequence or graph data is a kernel (graphbuild.hip, snapshot.hip, rounds.hip, commit.hip; prototypes in
// simplify_kernels.h); the host only sequences launches (simplify_driver.h) and reads a counter block back per round.
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

#include <chrono>

#include "sbl_ctx.h"
#include "sbl_comm.h"
#include "kmer_kernels.h"
#include "simplify_driver.h"
#include "simplify_kernels.h"
// cycle counters of the decision loops (bulge_txn.h: BT_PROF_ADD), device only

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------- device backend
struct SimplifyState {
	DevBuf ch, op, nx, pv, nodeof[2];
	DevBuf nslot, nnext, nidst, nclr, ndead, head[2], lsize[2];
	DevBuf ctr, need, big, touch, ck_touch, own, lock, rmax, wmax, win;
	DevBuf arena, snap_arena, big_arena, claims, live, robuf, bidx, instbuf, snap_list, snap_live, park_of, slice_busy, park_list;
	DevBuf ck_ch, ck_op, ck_nx, ck_pv, ck_bif[2], ck_nodeof[2], ck_nslot, ck_nnext, ck_ndead, ck_head[2], ck_lsize[2];
	DevBuf keys, skeys, selem, sorttmp, scantmp, perm, permin;
	DevBuf nmark, maux[2], iota, sel, tstamp;
	DevBuf lin, elin, lmpos[2], lmid[2], cnt1k, off1k;      // linearised marks of the later snapshots
	DevBuf flag, segidx, seg_head, seg_len, seg_succ_elem, succ[2], dist[2], newidx, ch_out, op_out;
	hipStream_t park_stream = nullptr; hipEvent_t park_ev[2] = {nullptr, nullptr};      // k_resume beside k_commit (GraphView::park_of)
	unsigned *h_ctr = nullptr;            // pinned, mapped: CTR_COUNT counters + the sequence number of the last post (k_select_write)
	unsigned *d_hctr = nullptr;           // its device address
	unsigned post_seq = 0;
	hipEvent_t ev[8] = {};                // created once per context (DeviceBackend borrows them): commit sampling pair [2, 3]
	std::vector<hipEvent_t> snap_ev;      // start / stop pairs around the snapshots of a stage, read at its end (no host synchronisation per snapshot)
	unsigned char *h_init = nullptr;      // pinned staging for the small host-to-device initialisations of a stage (no synchronisation needed before the host moves on)
};

struct RestartStage {};      // thrown out of an optimistic attempt that would need a roll-back (DeviceBackend::restore)
// thrown out of the ordered rounds when they have turned into a slow serial chain and the one-launch path is the better bet (round 6):
// many strains x genomes of a few kbp x a minBranchSize of a tenth of a genome -- a transaction's neighbourhood IS the genome, the
// rounds commit one or two transactions each, and a round costs 10 - 30 ms there (tools/stress.py MANY=1: seed 67000 121 s, 67008 17.8 s,
// 67012 19.8 s against 8.8 / 2.2 / 2.5 s through k_dense_stage, exact either way; the same path is 3 - 100 x SLOWER on every input the
// rounds parallelise -- measured case by case, tools/gpu_r06_k.sh -- so it is only taken after the rounds have shown what they are)
struct TryDense {};
#define DENSE_SWITCH_MAX_ELEMS (4u << 20)

struct DeviceBackend {
	sbl_ctx *c;
	SimplifyState *st;
	GraphView g{};
	uint32_t cap_e = 0, cap_n = 0, nid_ = 0;
	size_t ne0_ = 0;                                                  // elements of the stage's input (padded): cap_e - ne0_ is the insertion slack
	uint32_t ck_ne = 0, ck_nn = 0;
	uint32_t window = 0, arena_bytes = 1u << 17, snap_arena_bytes = 1u << 17, snap_threads = 256 * 32;   // snap_threads = resident waves
	uint32_t big_arena_bytes = 1u << 28;
	size_t nres = 0;
	hipEvent_t *ev = nullptr;                                          // SimplifyState::ev
	unsigned snap_used = 0;                                           // snapshot event pairs recorded in this stage
	bool timed_commit = false;
	bool later_stream = getenv("SBL_NO_STREAM_SNAPSHOT") == nullptr && getenv("SBL_NO_LATER_STREAM") == nullptr;
	bool first_stream = getenv("SBL_NO_STREAM_SNAPSHOT") == nullptr;      // measurement switch: the generic window-walking snapshot for iteration 1 too
	int prof = 0;
	// Optimistic attempt: no iteration checkpoints (1.45 GB of device-to-device copies per iteration, 3.6 ms of a 105 ms stage, for a
	// roll-back the benchmark workloads never take).  An order violation or a pool overflow then abandons the attempt (RestartStage) and
	// the stage is run again from its input -- intact until the copy-back -- with checkpoints and iteration replays (sbl_simplify_run).
	bool optimistic = false;
	unsigned rsv_waves = 4;                                           // waves of a reservation workgroup (k_reserve)
	// block index of the original slots (GraphView::bidx): read by the probe and the reservation, maintained by the transactions
	bool use_index = getenv("SBL_NO_BLOCK_INDEX") == nullptr;        // measurement / test switch: every window is walked (round 4)
	uint32_t idx_nblk = 0;
	unsigned probe_lds_pad = getenv("SBL_PROBE_LDS_PAD") ? (unsigned)atoi(getenv("SBL_PROBE_LDS_PAD")) : 0u;      // measurement switch: occupancy sensitivity of k_probe_idx
	// k_probe_idx's LDS by the instances an id has (set with rsv_waves): a handful -- 256-slot verdict table, 64 instances, 64 walked marks = 3.1 KB;
	// dozens (many strains) -- 1024 slots, 256 instances, 192 marks = 12 KB.  An entry that does not fit goes to the walking probe.
	unsigned pidx_vbits = 9, pidx_inst = 256, pidx_marks = 192;
	unsigned istride() const { return std::min(pidx_inst, 128u) + 1u; }      // words per window entry of the instance hand-over (k_probe_idx -> k_reserve)
	unsigned pidx_lds() const { return ((2u << pidx_vbits) + 2u * pidx_inst + 2u * pidx_marks) * 4u + pidx_inst + probe_lds_pad; }
	void index_build()
	{
		if (!use_index || !idx_nblk) return;
		k_build_blkidx<<<(idx_nblk + 3) / 4, 256, 0, c->stream>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), st->pv.as<unsigned>(), c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>(),
		                                                         g.norig, idx_nblk, st->bidx.as<unsigned long long>());
		HIP_TRY(hipGetLastError());
	}
	double snapshot_ms = 0, reserve_ms = 0, commit_ms = 0, probe_ms = 0;
	double commit_event_ms = 0; uint64_t commit_event_launches = 0;      // the event pairs around every 4th launch of the commit kernel
	unsigned ev_phase = 0;

	DeviceBackend() = default;
	DeviceBackend(const DeviceBackend &) = delete;
	uint32_t nid() { return nid_; }
	void bind()
	{
		g.ch = st->ch.as<uint8_t>(); g.op = st->op.as<uint32_t>(); g.nx = st->nx.as<uint32_t>(); g.pv = st->pv.as<uint32_t>();
		for (int s = 0; s < 2; s++) {
			g.bif[s] = c->d_bif[s].as<uint32_t>(); g.nodeof[s] = st->nodeof[s].as<uint32_t>();
			g.head[s] = st->head[s].as<uint32_t>(); g.lsize[s] = st->lsize[s].as<uint32_t>();
		}
		g.nslot = st->nslot.as<uint32_t>(); g.nnext = st->nnext.as<uint32_t>(); g.nidst = st->nidst.as<uint32_t>(); g.nclr = st->nclr.as<uint32_t>(); g.ndead = st->ndead.as<uint8_t>();
		g.ctr = st->ctr.as<uint32_t>(); g.need = st->need.as<uint8_t>(); g.big = st->big.as<uint8_t>(); g.touch = st->touch.as<uint8_t>();
		g.own = st->own.as<uint32_t>(); g.lock = nullptr; g.rmax = st->rmax.as<uint32_t>(); g.wmax = st->wmax.as<uint32_t>();
		g.cap_e = cap_e; g.cap_n = cap_n; g.nid = nid_;
		g.nblk = (cap_e >> BT_BLOCK_SHIFT) + 1;
		g.win = st->win.as<uint32_t>();
		nres = (size_t)g.nblk + nid_ + 1;
		// (lock[] belongs to the element-wise stamping of the host-side test driver; the kernels check exclusivity against own[])
		st->rmax.ensure(nres * 4); st->wmax.ensure(nres * 4);
		g.lock = nullptr; g.rmax = st->rmax.as<uint32_t>(); g.wmax = st->wmax.as<uint32_t>();
		g.bidx = use_index && idx_nblk ? st->bidx.as<unsigned long long>() : nullptr;
		g.idx_probe = g.bidx && getenv("SBL_NO_IDX_PROBE") == nullptr ? 1u : 0u;        // measurement switches: the walking probe / reservation for every entry
		g.idx_reserve = g.bidx && getenv("SBL_NO_IDX_RESERVE") == nullptr ? 1u : 0u;
	}
	// ---- the switch to the one-launch path (TryDense): allowed for this attempt, chain-mode rounds so far, progress of the stage
	bool may_try_dense = false;
	uint32_t chain_rounds = 0, iter_count = 0, last_lo = 0, max_iter_ = 1;
	std::chrono::steady_clock::time_point t_stage = std::chrono::steady_clock::now();
	bool posted = false;                                              // the selection in flight posts the counters itself (k_select_write)
	void read_ctr()
	{
		if (posted && sel_pending && !sel_ready) {
			// the counters of the round and the next window come with the selection's post: poll its sequence number
			volatile unsigned *seq = st->h_ctr + CTR_COUNT;
			for (unsigned long long spin = 0;; spin++) {
				if (*seq == st->post_seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); posted = false; return; }
				if ((spin & 0xFFFF) == 0xFFFF) {                        // every ~65 k polls: is the stream still running?
					hipError_t e = hipStreamQuery(c->stream);
					if (e == hipSuccess && *seq != st->post_seq) break;      // finished without posting: fall back to the copy
					if (e != hipSuccess && e != hipErrorNotReady) HIP_TRY(e);
				}
			}
			posted = false;
		}
		HIP_TRY(hipMemcpyAsync(st->h_ctr, st->ctr.p, CTR_COUNT * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	void copy(DevBuf &dst, const DevBuf &src, size_t bytes)
	{
		dst.ensure(bytes);
		if (bytes) HIP_TRY(hipMemcpyAsync(dst.p, src.p, bytes, hipMemcpyDeviceToDevice, c->stream));
	}
	void checkpoint()
	{
		iter_count++;                                               // (called once per iteration, before its first snapshot)
		read_ctr();
		ck_ne = st->h_ctr[CTR_NE]; ck_nn = st->h_ctr[CTR_NN];
		if (optimistic) return;
		copy(st->ck_ch, st->ch, ck_ne); copy(st->ck_op, st->op, (size_t)ck_ne * 4); copy(st->ck_nx, st->nx, (size_t)ck_ne * 4); copy(st->ck_pv, st->pv, (size_t)ck_ne * 4);
		for (int s = 0; s < 2; s++) {
			copy(st->ck_bif[s], c->d_bif[s], (size_t)ck_ne * 4); copy(st->ck_nodeof[s], st->nodeof[s], (size_t)ck_ne * 4);
			copy(st->ck_head[s], st->head[s], ((size_t)nid_ + 1) * 4); copy(st->ck_lsize[s], st->lsize[s], ((size_t)nid_ + 1) * 4);
		}
		copy(st->ck_touch, st->touch, (size_t)nid_ + 1);
		copy(st->ck_nslot, st->nslot, (size_t)ck_nn * 4); copy(st->ck_nnext, st->nnext, (size_t)ck_nn * 4); copy(st->ck_ndead, st->ndead, ck_nn);
	}
	void restore()
	{
		if (optimistic) throw RestartStage{};
		auto back = [&](DevBuf &dst, const DevBuf &src, size_t bytes) { if (bytes) HIP_TRY(hipMemcpyAsync(dst.p, src.p, bytes, hipMemcpyDeviceToDevice, c->stream)); };
		back(st->ch, st->ck_ch, ck_ne); back(st->op, st->ck_op, (size_t)ck_ne * 4); back(st->nx, st->ck_nx, (size_t)ck_ne * 4); back(st->pv, st->ck_pv, (size_t)ck_ne * 4);
		for (int s = 0; s < 2; s++) {
			back(c->d_bif[s], st->ck_bif[s], (size_t)ck_ne * 4); back(st->nodeof[s], st->ck_nodeof[s], (size_t)ck_ne * 4);
			back(st->head[s], st->ck_head[s], ((size_t)nid_ + 1) * 4); back(st->lsize[s], st->ck_lsize[s], ((size_t)nid_ + 1) * 4);
		}
		back(st->touch, st->ck_touch, (size_t)nid_ + 1);
		back(st->nslot, st->ck_nslot, (size_t)ck_nn * 4); back(st->nnext, st->ck_nnext, (size_t)ck_nn * 4); back(st->ndead, st->ck_ndead, ck_nn);
		unsigned v[2] = { ck_ne, ck_nn };
		HIP_TRY(hipMemcpyAsync(st->ctr.as<unsigned>() + CTR_NE, &v[0], 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(st->ctr.as<unsigned>() + CTR_NN, &v[1], 4, hipMemcpyHostToDevice, c->stream));
		index_build();                                              // (the block index is derived state: rebuilt from the restored arrays)
		HIP_TRY(hipStreamSynchronize(c->stream));
		parked_known = 0;                                           // (reset_round_state clears the device side before the first round)
	}
	// segment ranking of the current list (shared with the copy-back): returns the list length, leaves flag / segidx / seg_head / dist[cur] filled
	unsigned long long rank_segments(unsigned ne, int *cur_out)
	{
		hipStream_t s = c->stream;
		st->flag.ensure((size_t)ne * 4 + 16); st->segidx.ensure((size_t)ne * 4 + 16);
		k_seg_flags<<<(ne + 255) / 256, 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>());
		{
			size_t tmp = 0;
			HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), 0u, (size_t)ne + 1, rocprim::plus<unsigned>(), s));
			st->scantmp.ensure(tmp);
			HIP_TRY(rocprim::exclusive_scan(st->scantmp.p, tmp, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), 0u, (size_t)ne + 1, rocprim::plus<unsigned>(), s));
		}
		unsigned nseg = 0;
		HIP_TRY(hipMemcpyAsync(&nseg, st->segidx.as<unsigned>() + ne, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		SBL_CHECK(nseg >= 1, SBL_ERR_INTERNAL, "segment ranking: empty list");
		st->seg_head.ensure((size_t)nseg * 4); st->seg_len.ensure((size_t)nseg * 4); st->seg_succ_elem.ensure((size_t)nseg * 4);
		for (int t = 0; t < 2; t++) { st->succ[t].ensure((size_t)nseg * 4); st->dist[t].ensure((size_t)nseg * 8); }
		k_seg_tails<<<(ne + 255) / 256, 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(),
		                                            st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>());
		k_seg_finish<<<(nseg + 255) / 256, 256, 0, s>>>(nseg, st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>(),
		                                               st->flag.as<unsigned>(), st->segidx.as<unsigned>(), st->succ[0].as<unsigned>(), st->dist[0].as<unsigned long long>());
		int cur = 0;
		for (unsigned span = 1; span < nseg; span <<= 1, cur ^= 1)
			k_seg_jump<<<(nseg + 255) / 256, 256, 0, s>>>(nseg, st->succ[cur].as<unsigned>(), st->dist[cur].as<unsigned long long>(),
			                                             st->succ[cur ^ 1].as<unsigned>(), st->dist[cur ^ 1].as<unsigned long long>());
		unsigned long long total = 0;                               // segment 0 starts with element 0, the head of the whole list
		HIP_TRY(hipMemcpyAsync(&total, st->dist[cur].p, 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		*cur_out = cur;
		return total;
	}
	// marks of the current graph in list order (see k_lin_positions): fills ms
	void linearise_marks(MarkStream &ms)
	{
		hipStream_t s = c->stream;
		const unsigned ne = ck_ne;                                  // (checkpoint() has just read the counters)
		int cur = 0;
		const unsigned long long total = rank_segments(ne, &cur);
		st->lin.ensure((size_t)ne * 4 + 16); st->elin.ensure((size_t)total * 4 + 16);
		k_lin_positions<<<(ne + 255) / 256, 256, 0, s>>>(st->ch.as<uint8_t>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), st->seg_head.as<unsigned>(),
		                                                st->dist[cur].as<unsigned long long>(), total, st->lin.as<unsigned>(), st->elin.as<unsigned>());
		const unsigned nchunks = (unsigned)((total + 1023) / 1024);
		st->cnt1k.ensure((size_t)(nchunks + 1) * 4); st->off1k.ensure((size_t)(nchunks + 1) * 4);
		for (int t = 0; t < 2; t++) {
			HIP_TRY(hipMemsetAsync(st->cnt1k.p, 0, (size_t)(nchunks + 1) * 4, s));
			k_count_marks_lin<<<nchunks, 256, 0, s>>>(c->d_bif[t].as<unsigned>(), st->elin.as<unsigned>(), (size_t)total, st->cnt1k.as<unsigned>());
			size_t tmp = 0;
			HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, st->cnt1k.as<unsigned>(), st->off1k.as<unsigned>(), 0u, (size_t)nchunks + 1, rocprim::plus<unsigned>(), s));
			st->scantmp.ensure(tmp);
			HIP_TRY(rocprim::exclusive_scan(st->scantmp.p, tmp, st->cnt1k.as<unsigned>(), st->off1k.as<unsigned>(), 0u, (size_t)nchunks + 1, rocprim::plus<unsigned>(), s));
			unsigned nm = 0;
			HIP_TRY(hipMemcpyAsync(&nm, st->off1k.as<unsigned>() + nchunks, 4, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			st->lmpos[t].ensure((size_t)nm * 4 + 16); st->lmid[t].ensure((size_t)nm * 4 + 16); st->maux[t].ensure((size_t)nm * 4 + 16);
			if (nm) {
				k_write_marks_lin<<<nchunks, 256, 0, s>>>(c->d_bif[t].as<unsigned>(), st->elin.as<unsigned>(), st->nodeof[t].as<unsigned>(), (size_t)total, st->off1k.as<unsigned>(),
				                                         st->lmpos[t].as<unsigned>(), st->lmid[t].as<unsigned>(), st->nmark.as<unsigned>());
				k_mark_aux_lin<<<(nm + 255) / 256, 256, 0, s>>>(st->lmpos[t].as<unsigned>(), nm, (unsigned)t, c->d_sepidx.as<unsigned>(), c->nchr, st->lin.as<unsigned>(),
				                                               st->elin.as<unsigned>(), st->ch.as<uint8_t>(), g.k, st->maux[t].as<unsigned>());
			}
			ms.elem[t] = st->lmpos[t].as<unsigned>(); ms.id[t] = st->lmid[t].as<unsigned>(); ms.aux[t] = st->maux[t].as<unsigned>(); ms.n[t] = nm;
		}
		HIP_TRY(hipGetLastError());
	}
	// Linearising the marks for the stream costs ~1.5 ms whatever the number of ids to look at; when only a few per cent of the ids were
	// touched (iterations 3, 4: what the few collapses of the previous iteration can see) the window-walking snapshot of just those is cheaper.
	bool few_touched()
	{
		unsigned *cnt = st->ctr.as<unsigned>() + CTR_DETAIL + 8, n = 0;
		HIP_TRY(hipMemsetAsync(cnt, 0, 4, c->stream));
		k_count_touched<<<256, 256, 0, c->stream>>>(st->touch.as<uint8_t>(), nid_, cnt);
		HIP_TRY(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		// (... or few in absolute terms: linearising costs ~25 ns per thousand elements whatever the number of ids -- 23 ms of a 138 ms stage at
		// 900 Mbp, k = 5000, for 28 ids)
		return (unsigned long long)n * 16 < nid_ || n <= std::max<size_t>(1024, ne0_ >> 12);
	}
	// ---- read-only phases split over the attached GPUs (SURVEY.md 8e, row "Simplification"): the commits are replicated, so the graph is
	// identical on every GPU before a snapshot and before a probe; each GPU takes the verdicts of ITS share (a slice of the positional
	// order of the ids / of the window) and the verdict bytes are all-gathered: 1 B per id per snapshot, 1 B per window entry per round.
	// SBL_REPLICATED_PHASES=1: measurement / test switch, every GPU computes everything (the round-3 behaviour).
	bool split_ro() const { return c->comm && c->comm->n > 1 && getenv("SBL_REPLICATED_PHASES") == nullptr; }
	double ro_ms = 0;                                                 // host time inside the verdict collectives
	void share(uint32_t n, uint32_t *lo, uint32_t *hi) const
	{
		const uint32_t R = c->comm->n, r = c->comm->rank;
		*lo = (uint32_t)((uint64_t)n * r / R); *hi = (uint32_t)((uint64_t)n * (r + 1) / R);
	}
	// buf[lo_r, hi_r) of every rank r (element size `es` bytes, n elements in all) to everybody, in place
	void allgather_shares(char *buf, uint32_t n, size_t es)
	{
		SblComm *cm = c->comm;
		const uint32_t R = cm->n, r = cm->rank;
		std::vector<size_t> sb(R), so(R), rb(R), ro(R);
		for (uint32_t p = 0; p < R; p++) {
			const size_t plo = (size_t)((uint64_t)n * p / R) * es, phi = (size_t)((uint64_t)n * (p + 1) / R) * es;
			const size_t mlo = (size_t)((uint64_t)n * r / R) * es, mhi = (size_t)((uint64_t)n * (r + 1) / R) * es;
			sb[p] = p == r ? 0 : mhi - mlo; so[p] = mlo;
			rb[p] = p == r ? 0 : phi - plo; ro[p] = plo;
			if (p != r) c->stats.verdict_bytes += mhi - mlo;
		}
		const auto t0 = std::chrono::steady_clock::now();
		try { cm->alltoallv(c, buf, sb.data(), so.data(), buf, rb.data(), ro.data()); }
		catch (...) { cm->abort_peers(); throw; }
		ro_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	}
	// slot r (es bytes) of every rank r to everybody, in place
	void allgather_slots(char *buf, uint32_t R, size_t es) { allgather_shares(buf, R, es); }
	uint32_t snap_slice = 0;                                          // entries per launch of the walking probe's arena (= the round buffers' window_max)
	void snapshot_idx()
	{
		unsigned *cnt = st->ctr.as<unsigned>() + CTR_DETAIL + 8, n = 0;
		st->snap_list.ensure((size_t)nid_ * 4 + 64); st->snap_live.ensure((size_t)nid_ + 64);
		HIP_TRY(hipMemsetAsync(cnt, 0, 4, c->stream));
		k_touched_list<<<(nid_ + 255) / 256, 256, 0, c->stream>>>(st->touch.as<uint8_t>(), nid_, st->snap_list.as<unsigned>(), cnt);
		HIP_TRY(hipMemcpyAsync(&n, cnt, 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemsetAsync(st->need.p, 0, (size_t)nid_ + 1, c->stream));      // untouched ids: still clean
		HIP_TRY(hipMemsetAsync(st->touch.p, 0, (size_t)nid_ + 1, c->stream));     // every verdict is taken now
		HIP_TRY(hipStreamSynchronize(c->stream));
		const uint32_t slice = std::max<uint32_t>(1, snap_slice);
		for (uint32_t off = 0; off < n; off += slice) {
			const uint32_t m = std::min<uint32_t>(slice, n - off);
			GraphView gs = g;
			gs.win = st->snap_list.as<unsigned>() + off;
			gs.tstamp = nullptr;
			uint8_t *lv = st->snap_live.as<uint8_t>() + off;
			k_probe_idx<<<m, 64, pidx_lds(), c->stream>>>(gs, m, lv, 0u, pidx_vbits, pidx_inst, pidx_marks, nullptr, 0u, 1);
			k_probe<<<m, 64 * PROBE_WAVES, 0, c->stream>>>(gs, m, st->arena.as<uint8_t>(), arena_bytes, lv, 0u, 1);
		}
		HIP_TRY(hipGetLastError());
	}
	void snapshot_all(bool incremental)
	{
		while (st->snap_ev.size() < 2 * (size_t)(snap_used + 1)) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); st->snap_ev.push_back(e); }
		HIP_TRY(hipEventRecord(st->snap_ev[2 * snap_used], c->stream));
		uint32_t plo = 0, phi = nid_;
		const bool split = split_ro();
		if (split) share(nid_, &plo, &phi);
		if (!incremental && first_stream) {
			// iteration 1 (also after a replay: the checkpoint restored is the pristine graph): stream over the position-ordered marks
			MarkStream ms;
			for (int t = 0; t < 2; t++) { ms.elem[t] = c->d_melem[t].as<unsigned>(); ms.id[t] = c->d_mid[t].as<unsigned>(); ms.aux[t] = st->maux[t].as<unsigned>(); ms.n[t] = c->nmarks[t]; }
			HIP_TRY(hipMemsetAsync(st->touch.p, 0, (size_t)nid_ + 1, c->stream));
			k_snapshot_first<<<256 * 32, 64, 0, c->stream>>>(g, ms, st->nmark.as<unsigned>(), st->perm.as<unsigned>(), plo, phi);
		} else if (incremental && g.idx_probe && !split && getenv("SBL_NO_IDX_SNAPSHOT") == nullptr) {
			// iterations 2 ..: the touched ids through the probe of the rounds (k_probe_idx over the block index, the walking probe for what it
			// cannot serve) -- the same verdict the stream takes, without linearising the marks first (round 5)
			snapshot_idx();
		} else if (incremental && later_stream && !few_touched()) {
			// iterations 2 ..: the same stream over the marks of the current graph in list order
			st->nmark.ensure((size_t)cap_n * 4);
			MarkStream ms;
			linearise_marks(ms);
			k_snapshot_stream<<<256 * 32, 64, 0, c->stream>>>(g, ms, st->nmark.as<unsigned>(), st->perm.as<unsigned>(), 1, plo, phi);
		} else {
			// (the walking snapshot's scratch -- up to 16 GB where D is in the thousands -- is only allocated when that kernel runs: a fresh
			// context that never needs it used to spend seconds mapping it, tools/stress.py LONGK=1)
			st->snap_arena.ensure((size_t)snap_threads * snap_arena_bytes);
			k_snapshot<<<snap_threads, 64, 0, c->stream>>>(g, st->snap_arena.as<uint8_t>(), snap_arena_bytes, incremental ? 1 : 0, st->perm.as<unsigned>(), plo, phi);
		}
		HIP_TRY(hipGetLastError());
		if (split) {
			// the verdict bytes of my share of the positional order to everybody; after a snapshot no id is "touched" any more, anywhere
			st->robuf.ensure((size_t)nid_ + 64);
			if (phi > plo) k_pack_need<<<(phi - plo + 255) / 256, 256, 0, c->stream>>>(st->perm.as<unsigned>(), st->need.as<uint8_t>(), plo, phi, st->robuf.as<uint8_t>());
			allgather_shares(st->robuf.as<char>(), nid_, 1);
			if (nid_) k_unpack_need<<<(nid_ + 255) / 256, 256, 0, c->stream>>>(st->perm.as<unsigned>(), st->robuf.as<uint8_t>(), nid_, plo, phi, st->need.as<uint8_t>());
			HIP_TRY(hipMemsetAsync(st->touch.p, 0, (size_t)nid_ + 1, c->stream));
			HIP_TRY(hipGetLastError());
		}
		HIP_TRY(hipEventRecord(st->snap_ev[2 * snap_used + 1], c->stream));
		snap_used++;                                                // (read by snapshots_collect at the end of the stage: the host does not wait here)
	}
	void snapshots_collect()
	{
		for (unsigned i = 0; i < snap_used; i++) {
			float ms = 0;
			HIP_TRY(hipEventSynchronize(st->snap_ev[2 * i + 1]));
			HIP_TRY(hipEventElapsedTime(&ms, st->snap_ev[2 * i], st->snap_ev[2 * i + 1]));
			snapshot_ms += ms;
		}
		snap_used = 0;
	}
	void reset_round_state(bool stamps_too)
	{
		if (stamps_too && g.park_cap) {                                // (start of an iteration / of a replay: nothing is parked)
			HIP_TRY(hipMemsetAsync(st->park_of.p, 0, ((size_t)nid_ + 1) * 4, c->stream));
			HIP_TRY(hipMemsetAsync(st->slice_busy.p, 0, park_slices, c->stream));
			parked_known = 0; g.park_hold = 0;
		}
		HIP_TRY(hipMemsetAsync(st->own.p, 0xFF, ((size_t)nid_ + 1) * 4, c->stream));
		if (stamps_too) {
			HIP_TRY(hipMemsetAsync(st->rmax.p, 0, nres * 4, c->stream));
			HIP_TRY(hipMemsetAsync(st->wmax.p, 0, nres * 4, c->stream));
			if (g.bidx) k_idx_clear_stamps<<<(idx_nblk + 255) / 256, 256, 0, c->stream>>>(st->bidx.as<unsigned long long>(), idx_nblk);
		}
	}
	void clear_counters()
	{
		// everything but the pool cursors (CTR_NE, CTR_NN); one small kernel, no host synchronisation
		k_clear_counters<<<1, 256, 0, c->stream>>>(st->ctr.as<unsigned>());
		HIP_TRY(hipGetLastError());
	}
	// The selection is launched behind a round's last kernel and read with the round's counters: one host round trip per round.
	bool sel_pending = false, sel_ready = false;
	uint32_t probed_nwin = 0;                                         // entries of the last probe whose retirements have not been counted yet
	void select_launch(uint32_t lo, uint32_t limit, uint32_t W)
	{
		// chunks of >= 8192 ids, at most ~1024 of them
		unsigned chunk = 8192;
		while ((unsigned long long)chunk * 1024 < (unsigned long long)nid_ + 1) chunk <<= 1;
		const unsigned chunk0 = lo / chunk, nchunks = limit / chunk - chunk0 + 1;
		// (one launch with a look-back over per-chunk slots was tried: 37.7 us against 10.7 + 10.6 us for the two -- the agent-scope
		// release / acquire of the slots writes back and invalidates the L2 of the XCD, which a kernel boundary does once)
		GraphView gs = g;
		if (sel_stamped) gs.tslot = TS_CAP * 4;                     // only the selection right behind a round marks that round's end
		sel_stamped = true;
		k_select_count<<<nchunks, SEL_THREADS, 0, c->stream>>>(gs, st->sel.as<unsigned>(), lo, limit, chunk0, chunk, st->live.as<uint8_t>(), probed_nwin);
		probed_nwin = 0;
		posted = st->d_hctr != nullptr;
		if (posted) st->post_seq++;
		k_select_write<<<nchunks, SEL_THREADS, 0, c->stream>>>(g, st->sel.as<unsigned>(), st->win.as<unsigned>(), lo, limit, W, chunk0, chunk, nchunks, st->d_hctr, st->post_seq);
		HIP_TRY(hipGetLastError());
		sel_pending = true; sel_ready = false;
	}
	void select_read(uint32_t *nwin, uint32_t *newlo, uint32_t *solo)
	{
		if (!sel_ready) read_ctr();                                 // (the first selection of an iteration, or one re-issued behind a fence)
		sel_pending = sel_ready = false;
		*nwin = st->h_ctr[CTR_NWIN]; *newlo = st->h_ctr[CTR_LO]; *solo = st->h_ctr[CTR_PUSHED];
		last_lo = *newlo;
	}
	// Per-kernel times of the rounds: start stamps written by the kernels themselves (round_stamp) for probe and reservation, and a
	// HIP event pair around the dominant kernel, k_commit (what bench.py's roofline is computed from).
	bool phase_events = getenv("SBL_NO_PHASE_EVENTS") == nullptr;    // measurement switch: what the per-round events themselves cost
	enum { TS_CAP = 16384 };                                         // rounds with stamps per stage (later ones go untimed)
	uint32_t ts_round = 0;
	bool sel_stamped = true;
	std::vector<uint8_t> ts_kind;                                    // per round: bit 0 probe, bit 1 reservation, bit 2 commit / chain launched
	void stamps_init()
	{
		st->tstamp.ensure((size_t)(TS_CAP + 1) * 4 * 8);
		HIP_TRY(hipMemsetAsync(st->tstamp.p, 0, (size_t)(TS_CAP + 1) * 4 * 8, c->stream));
		ts_round = 0; ts_kind.clear(); sel_stamped = true;
		g.tstamp = st->tstamp.as<unsigned long long>(); g.tslot = TS_CAP * 4;      // (the slot behind the last round: writes nobody reads)
	}
	void begin_round()
	{
		if (ts_round < TS_CAP) { g.tslot = 4 * ts_round++; ts_kind.push_back(0); sel_stamped = false; }
		else g.tslot = TS_CAP * 4;
	}
	// probe_ms / reserve_ms of the stage from the stamps (the event pair gives commit_ms)
	void stamps_collect()
	{
		if (!ts_round) return;
		std::vector<unsigned long long> h((size_t)ts_round * 4);
		HIP_TRY(hipMemcpyAsync(h.data(), st->tstamp.p, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		int khz = 0;
		if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) != hipSuccess || khz <= 0) khz = 100000;
		const double ms_per_tick = 1.0 / (double)khz;
		for (uint32_t r = 0; r < ts_round; r++) {
			const unsigned long long tp = h[4 * r], tr = h[4 * r + 1], tc = h[4 * r + 2];
			const bool probed = ts_kind[r] & 1, reserved = ts_kind[r] & 2;
			if (probed) { const unsigned long long next = reserved ? tr : tc; if (next > tp && tp) probe_ms += (double)(next - tp) * ms_per_tick; }
			if (reserved && tc > tr && tr) reserve_ms += (double)(tc - tr) * ms_per_tick;
			const unsigned long long te = h[4 * r + 3];                 // start of the selection behind the round
			if ((ts_kind[r] & 4) && te > tc && tc) commit_ms += (double)(te - tc) * ms_per_tick;
		}
	}
	void probe(uint32_t nwin, uint32_t round)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		g.any_parked = parked_known != 0;                            // (exact: nothing parks or resumes between the counters of the round before and this launch)
		begin_round();
		solo_round = false;
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 1;
		if (split_ro()) {
			// my share of the window; live[] of the other shares and the lowest order violation anybody saw come back in ONE small all-gather
			// (k_pack_probe / k_apply_probe: a record of 4 + ceil(nwin / R) bytes per rank)
			uint32_t w0 = 0, w1 = nwin;
			share(nwin, &w0, &w1);
			const uint32_t R = c->comm->n;
			const uint32_t stride = (4u + (nwin + R - 1) / R + 1u + 3u) & ~3u;
			st->robuf.ensure((size_t)R * stride + 64);
			if (w1 > w0 && g.idx_probe) k_probe_idx<<<w1 - w0, 64, pidx_lds(), c->stream>>>(g, nwin, st->live.as<uint8_t>(), w0, pidx_vbits, pidx_inst, pidx_marks, nullptr, 0u, 0);      // (shares of a split probe: the other ranks' lists would have to travel too)
			if (w1 > w0) k_probe<<<w1 - w0, 64 * PROBE_WAVES, 0, c->stream>>>(g, nwin, st->arena.as<uint8_t>(), arena_bytes, st->live.as<uint8_t>(), w0, 0);
			k_pack_probe<<<std::max<uint32_t>(1u, (w1 - w0 + 255) / 256), 256, 0, c->stream>>>(st->ctr.as<unsigned>(), st->live.as<uint8_t>(), w0, w1, c->comm->rank, stride, st->robuf.as<uint8_t>());
			HIP_TRY(hipGetLastError());
			allgather_slots(st->robuf.as<char>(), R, stride);
			k_apply_probe<<<(nwin + 255) / 256, 256, 0, c->stream>>>(g, nwin, st->live.as<uint8_t>(), w0, w1, st->robuf.as<uint8_t>(), stride, R);
		} else {
			if (g.idx_probe) k_probe_idx<<<nwin, 64, pidx_lds(), c->stream>>>(g, nwin, st->live.as<uint8_t>(), 0u, pidx_vbits, pidx_inst, pidx_marks, st->instbuf.as<unsigned>(), istride(), 0);      // the block index first; k_probe walks what it could not serve
			k_probe<<<nwin, 64 * PROBE_WAVES, 0, c->stream>>>(g, nwin, st->arena.as<uint8_t>(), arena_bytes, st->live.as<uint8_t>(), 0u, 0);
		}
		probed_nwin = nwin;                                          // (the next selection counts what this probe retired)
		HIP_TRY(hipGetLastError());
	}
	bool solo_round = false;                                          // the current round skipped the probe (mark_live): no instance hand-over
	void mark_live(uint32_t nwin) { g.any_parked = parked_known != 0; begin_round(); solo_round = true; HIP_TRY(hipMemsetAsync(st->live.p, 1, nwin, c->stream)); }
	void reserve(uint32_t nwin, uint32_t round)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 2;
		// dynamic LDS: the seen-set + one compaction list per wave, sized by the instances an id has (a handful: 1024 + 2 x 256 words = 6 KB;
		// dozens: 2048 + 4 x 1024 words) -- what a reservation workgroup holds in LDS decides how many are resident
		const unsigned seen_bits = rsv_waves <= 2 ? 10u : 11u, list_cap = rsv_waves <= 2 ? 256u : 1024u;
		const bool handed = g.idx_probe && !split_ro() && !solo_round;
		k_reserve<<<nwin, 64 * rsv_waves, ((1u << seen_bits) + rsv_waves * list_cap) * 4, c->stream>>>(g, nwin, st->claims.as<unsigned>(), st->live.as<uint8_t>(), seen_bits, list_cap,
		                                                                                               handed ? st->instbuf.as<unsigned>() : nullptr, istride(),
		                                                                                               g.park_cap ? st->arena.as<uint8_t>() : nullptr, arena_bytes);
		HIP_TRY(hipGetLastError());
	}
	void commit(uint32_t nwin, uint32_t round, bool solo)
	{
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		// an event pair around every 4th launch (which ones rotates from stage to stage); the start stamps time all of them
		const bool sampled = phase_events && ((round + ev_phase) & 3u) == 0;
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 4;
		if (sampled) HIP_TRY(hipEventRecord(ev[2], c->stream));
		if (solo) {
			st->big_arena.ensure(big_arena_bytes);
			k_commit<<<1, 64, 0, c->stream>>>(g, 1, st->big_arena.as<uint8_t>(), big_arena_bytes, 1, nullptr, nullptr, prof);
		} else
		{
			// parked transactions resume beside the fresh ones: a second stream between two events (only in rounds that have any -- the
			// counters of the previous round say so: what is parked now was parked then)
			const bool resume = g.park_cap && parked_known != 0;
			if (g.park_cap && (round & 1023u) == 0u) k_park_sweep<<<nblocks((size_t)nid_ + 1, 256), 256, 0, c->stream>>>(st->park_of.as<unsigned>(), nid_ + 1);      // (bt_round_tag)
			if (resume) {
				// one workgroup per parked transaction at most (k_reserve has listed those of this window: GraphView::park_list)
				HIP_TRY(hipEventRecord(st->park_ev[0], c->stream));
				HIP_TRY(hipStreamWaitEvent(st->park_stream, st->park_ev[0], 0));
				k_resume<<<std::min<uint32_t>(parked_known, nwin), 64, 0, st->park_stream>>>(g, nwin, st->arena.as<uint8_t>(), arena_bytes, st->claims.as<unsigned>(), st->live.as<uint8_t>(), prof);
				HIP_TRY(hipEventRecord(st->park_ev[1], st->park_stream));
			}
			k_commit<<<nwin, 64, 0, c->stream>>>(g, nwin, st->arena.as<uint8_t>(), arena_bytes, 0, st->claims.as<unsigned>(), st->live.as<uint8_t>(), prof);
			if (resume) HIP_TRY(hipStreamWaitEvent(c->stream, st->park_ev[1], 0));
		}
		if (sampled) HIP_TRY(hipEventRecord(ev[3], c->stream));
		timed_commit = sampled;
		HIP_TRY(hipGetLastError());
	}
	// serial chain over what is pending in the id range of the window (k_chain); timed with the commit phase
	size_t park_slices = 0;
	uint32_t parked_known = 0;                                        // ctr[CTR_PARKED] as of the last counters(): transactions parked when the next launches start
	bool chain(uint32_t nwin, uint32_t round)
	{
		// the driver is in chain mode: the rounds have stopped being parallel.  After a second, with less than 40 % of the stage behind it
		// (the first iteration counted as 80 % of a stage: the later ones see what the few collapses of the one before left), the attempt
		// is given up for the one-launch path -- which is within 2 x of the rounds on every many-strains case measured and up to 8 x
		// faster on the slow ones (tools/gpu_r06_k.sh).
		if (may_try_dense && ++chain_rounds >= 32u) {
			const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage).count();
			const double within = (double)last_lo / (double)(nid_ ? nid_ : 1u);
			const double progress = iter_count <= 1u ? 0.8 * within : 0.8 + 0.2 * ((double)(iter_count - 2u) + within) / (double)(max_iter_ > 1u ? max_iter_ - 1u : 1u);
			if (el > 1.0 && progress < 0.4) throw TryDense{};
		}
		// parked transactions resume in an ordered round (the chain kernel has another LDS layout): the driver has asked for the chain, so
		// nothing NEW parks from here on (GraphView::park_hold) -- what is parked runs to its end in the next rounds and the chain starts then
		if (g.park_cap && parked_known) { g.park_hold = 1; return false; }
		g.round_bits = (SS_ROUND_MAX - round) << 20;
		st->big_arena.ensure(big_arena_bytes);
		if (g.tslot < TS_CAP * 4) ts_kind.back() |= 4;
		k_chain<<<1, 64, 0, c->stream>>>(g, st->big_arena.as<uint8_t>(), big_arena_bytes, nwin, prof);
		timed_commit = false;
		HIP_TRY(hipGetLastError());
		return true;
	}
	SimplifyCounters counters()
	{
		read_ctr();
		if (sel_pending) sel_ready = true;                          // the snapshot holds the selection launched before it as well
		float ms = 0;
		if (timed_commit) {
			hipError_t e = hipEventElapsedTime(&ms, ev[2], ev[3]);      // (the post of the selection behind them has arrived: normally complete)
			if (e == hipErrorNotReady) { HIP_TRY(hipEventSynchronize(ev[3])); e = hipEventElapsedTime(&ms, ev[2], ev[3]); }
			HIP_TRY(e);
			commit_event_ms += ms; commit_event_launches++; timed_commit = false;
		}
		SimplifyCounters r;
		memcpy(r.v, st->h_ctr, sizeof r.v);
		parked_known = r.v[CTR_PARKED];
		return r;
	}
	bool grow(uint32_t err)
	{
		if (err & ~(uint32_t)(BT_ERR_ELEM_CAP | BT_ERR_NODE_CAP)) return false;
		if (optimistic) {
			// no checkpoint to replay from: the attempt is abandoned BEFORE any buffer is enlarged, and the capacities the rerun (and the
			// later stages of this context) should start with are left on the context -- the rerun used to start with the original
			// capacities again, overflow again at the same place, and only then grow + replay
			if (err & BT_ERR_ELEM_CAP) c->hint_elem_slack = std::max<size_t>(c->hint_elem_slack, 2 * ((size_t)cap_e - ne0_));
			if (err & BT_ERR_NODE_CAP) c->hint_cap_n = std::max<size_t>(c->hint_cap_n, 2 * (size_t)cap_n);
			throw RestartStage{};
		}
		hipStream_t s = c->stream;
		if (err & BT_ERR_ELEM_CAP) {
			size_t n = (size_t)cap_e * 2;
			SBL_CHECK(n < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "element capacity overflow");
			st->ch.grow_keep(n, cap_e, s); st->op.grow_keep(n * 4, (size_t)cap_e * 4, s); st->nx.grow_keep(n * 4, (size_t)cap_e * 4, s); st->pv.grow_keep(n * 4, (size_t)cap_e * 4, s);
			for (int k = 0; k < 2; k++) { c->d_bif[k].grow_keep(n * 4, (size_t)cap_e * 4, s); st->nodeof[k].grow_keep(n * 4, (size_t)cap_e * 4, s); }
			HIP_TRY(hipMemsetAsync(st->ch.as<uint8_t>() + cap_e, 0, n - cap_e, s));
			HIP_TRY(hipMemsetAsync(st->nx.as<unsigned>() + cap_e, 0xFF, (n - cap_e) * 4, s));
			HIP_TRY(hipMemsetAsync(st->pv.as<unsigned>() + cap_e, 0xFF, (n - cap_e) * 4, s));
			for (int k = 0; k < 2; k++) HIP_TRY(hipMemsetAsync(c->d_bif[k].as<unsigned>() + cap_e, 0xFF, (n - cap_e) * 4, s));
			cap_e = (uint32_t)n;
		}
		if (err & BT_ERR_NODE_CAP) {
			size_t n = (size_t)cap_n * 2;
			SBL_CHECK(n < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "node capacity overflow");
			st->nslot.grow_keep(n * 4, (size_t)cap_n * 4, s); st->nnext.grow_keep(n * 4, (size_t)cap_n * 4, s); st->nclr.grow_keep(n * 4, (size_t)cap_n * 4, s); st->nidst.grow_keep(n * 4, (size_t)cap_n * 4, s);
			st->ndead.grow_keep(n, cap_n, s);
			cap_n = (uint32_t)n;
		}
		bind();
		return true;
	}
};

void sbl_simplify_free(sbl_ctx *c)
{
	SimplifyState *st = c->simp;
	if (!st) return;
	DevBuf *bufs[] = { &st->ch, &st->op, &st->nx, &st->pv, &st->nodeof[0], &st->nodeof[1], &st->nslot, &st->nnext, &st->nidst, &st->nclr, &st->ndead,
	                   &st->head[0], &st->head[1], &st->lsize[0], &st->lsize[1], &st->ctr, &st->need, &st->big, &st->touch, &st->ck_touch, &st->own, &st->lock, &st->rmax, &st->wmax, &st->win,
	                   &st->arena, &st->snap_arena, &st->big_arena, &st->claims, &st->live, &st->robuf, &st->bidx, &st->instbuf, &st->snap_list, &st->snap_live, &st->park_of, &st->slice_busy, &st->park_list, &st->ck_ch, &st->ck_op, &st->ck_nx, &st->ck_pv, &st->ck_bif[0], &st->ck_bif[1],
	                   &st->ck_nodeof[0], &st->ck_nodeof[1], &st->ck_nslot, &st->ck_nnext, &st->ck_ndead, &st->ck_head[0], &st->ck_head[1], &st->ck_lsize[0], &st->ck_lsize[1],
	                   &st->lin, &st->elin, &st->lmpos[0], &st->lmpos[1], &st->lmid[0], &st->lmid[1], &st->cnt1k, &st->off1k, &st->sel, &st->tstamp, &st->nmark, &st->maux[0], &st->maux[1], &st->iota, &st->keys, &st->skeys, &st->selem, &st->sorttmp, &st->scantmp, &st->perm, &st->permin, &st->flag, &st->segidx, &st->seg_head, &st->seg_len, &st->seg_succ_elem,
	                   &st->succ[0], &st->succ[1], &st->dist[0], &st->dist[1], &st->newidx, &st->ch_out, &st->op_out };
	for (DevBuf *b : bufs) b->release();
	if (st->park_stream) { (void)hipStreamDestroy(st->park_stream); for (auto &e : st->park_ev) if (e) (void)hipEventDestroy(e); }
	if (st->h_ctr) (void)hipHostFree(st->h_ctr);
	if (st->h_init) (void)hipHostFree(st->h_init);
	for (auto &e : st->ev) if (e) (void)hipEventDestroy(e);
	for (auto &e : st->snap_ev) (void)hipEventDestroy(e);
	delete st;
	c->simp = nullptr;
}

static void sort_pairs64(sbl_ctx *c, SimplifyState *st, unsigned long long *kin, unsigned long long *kout, unsigned *vin, unsigned *vout, size_t n, unsigned bits = 64)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
	st->sorttmp.ensure(tmp);
	HIP_TRY(rocprim::radix_sort_pairs(st->sorttmp.p, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
}
static unsigned bits_of(unsigned long long v) { unsigned b = 1; while (b < 64 && (v >> b)) b++; return b; }
static void scan_u32(sbl_ctx *c, SimplifyState *st, unsigned *in, unsigned *out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
	st->scantmp.ensure(tmp);
	HIP_TRY(rocprim::exclusive_scan(st->scantmp.p, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
}

// Inputs up to this many elements take the one-launch path (k_dense_stage) first; SBL_NO_DENSE_PATH=1 / SBL_DENSE_MAX_ELEMS=n: test switches
#define DENSE_MAX_ELEMS (1u << 16)
// The reference's callback sequence (blockfinder.cpp:23-48) is a function of the call index alone -- start, run(min(i, 50)) for i = 1, 2, ...,
// end -- so an attempt that is abandoned and run again delivers only the calls the caller has not seen yet.
struct ProgressFilter {
	sbl_progress_fn fn; void *user;
	bool started = false, ended = false;
	uint64_t delivered = 0, seen = 0;          // run calls forwarded so far / made by the current attempt
	static void relay(size_t p, int state, void *self_)
	{
		ProgressFilter *f = (ProgressFilter *)self_;
		if (state == SBL_PROGRESS_START) { f->seen = 0; if (!f->started) { f->started = true; f->fn(p, state, f->user); } }
		else if (state == SBL_PROGRESS_RUN) { if (++f->seen > f->delivered) { f->delivered = f->seen; f->fn(p, state, f->user); } }
		else if (!f->ended) { f->ended = true; f->fn(p, state, f->user); }
	}
};
enum { RUN_DONE = 0, RUN_DENSE_FAILED = 1, RUN_RESTART = 2, RUN_TRY_DENSE = 3 };
static int simplify_run_impl(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges, bool allow_dense, bool optimistic,
                             bool force_dense = false /* the one-launch path for up to DENSE_SWITCH_MAX_ELEMS elements: the rounds gave up (TryDense) */, bool may_switch = false /* this attempt may give up for it */);
static void simplify_run_guarded(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges);
void sbl_simplify_run(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges)
{
	// With several GPUs on one job every snapshot and every probe of the stage is a collective (allgather_shares): a rank that leaves
	// the stage with an error anywhere -- an allocation that fails, a HIP error, a check that throws -- must release the peers that
	// would wait for it there (the deterministic RestartStage is thrown on all ranks alike and never gets here).
	if (!c->comm) { simplify_run_guarded(c, k, D, max_iter, progress, user, bulges); return; }
	try { simplify_run_guarded(c, k, D, max_iter, progress, user, bulges); }
	catch (...) { c->comm->abort_peers(); throw; }
}
static void simplify_run_guarded(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges)
{
	// the stage's input (d_ch / d_op) is only replaced by the copy-back at the very end, so an attempt that cannot finish -- a one-launch
	// run out of pool or arena space, an optimistic run that would need a roll-back -- is simply followed by the next one from the same input
	ProgressFilter pf{progress, user};
	sbl_progress_fn pfn = progress ? &ProgressFilter::relay : nullptr;
	// (SBL_CHECKPOINTS: measurement / test switch, checkpoints from the first attempt on; hint_checkpoints: the previous stage of this
	// context had to be abandoned for an order violation -- inputs that do that once tend to do it again, and an abandoned attempt
	// costs a whole stage, a checkpoint 2 - 4 %)
	const bool optimistic = getenv("SBL_CHECKPOINTS") == nullptr && !c->hint_checkpoints;
	int r = simplify_run_impl(c, k, D, max_iter, pfn, &pf, bulges, true, optimistic, false, true);
	if (r == RUN_TRY_DENSE) {
		if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] the ordered rounds have turned into a slow serial chain: the stage runs again through the one-launch path\n");
		r = simplify_run_impl(c, k, D, max_iter, pfn, &pf, bulges, true, optimistic, true, false);
		c->stats.replays++;                                               // the abandoned attempt
	}
	if (r == RUN_DENSE_FAILED) r = simplify_run_impl(c, k, D, max_iter, pfn, &pf, bulges, false, optimistic);
	if (r == RUN_RESTART) {
		if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] optimistic attempt abandoned: the stage runs again with iteration checkpoints (element slack hint %zu, node capacity hint %zu)\n", c->hint_elem_slack, c->hint_cap_n);
		(void)simplify_run_impl(c, k, D, max_iter, pfn, &pf, bulges, false, false);
		c->stats.replays++;                                               // the abandoned attempt
		c->hint_checkpoints = c->stats.replays > c->stats.grow_replays + 1;      // order violations (not just a pool that was too small): the next stage starts with checkpoints
	} else if (!optimistic && r == RUN_DONE && c->stats.replays == c->stats.grow_replays) c->hint_checkpoints = false;      // a checkpointed stage that never rolled back
}
static int simplify_run_impl(sbl_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter, sbl_progress_fn progress, void *user, uint64_t *bulges, bool allow_dense, bool optimistic,
                             bool force_dense, bool may_switch)
{
	hipStream_t s = c->stream;
	if (!c->simp) {
		c->simp = new SimplifyState();
		HIP_TRY(hipHostMalloc((void **)&c->simp->h_ctr, (CTR_COUNT + 16) * 4, hipHostMallocMapped | hipHostMallocCoherent));
		memset(c->simp->h_ctr, 0, (CTR_COUNT + 16) * 4);
		if (hipHostGetDevicePointer((void **)&c->simp->d_hctr, c->simp->h_ctr, 0) != hipSuccess) { (void)hipGetLastError(); c->simp->d_hctr = nullptr; }
		if (getenv("SBL_NO_POST")) c->simp->d_hctr = nullptr;             // measurement switch: copy + synchronise as before
	}
	SimplifyState *st = c->simp;
	DeviceBackend be;
	be.c = c; be.st = st;
	c->stats = sbl_stage_stats{};
	HIP_TRY(hipEventRecord(c->ev[2], s));

	// ---- E1: enumeration into mark arrays with room for inserted elements
	size_t E = c->nelem, ne0 = (E + 31) / 32 * 32;
	size_t cap_e = ne0 + E / 8 + (1u << 20);
	if (const char *e = getenv("SBL_TEST_ELEM_SLACK")) cap_e = ne0 + (size_t)atoll(e);      // test hook: provoke the grow / restart paths
	cap_e = std::max(cap_e, ne0 + std::min<size_t>(c->hint_elem_slack, 4 * E + (1u << 20)));      // what an abandoned attempt of this context asked for (DeviceBackend::grow), bounded by the current input
	be.ne0_ = ne0;
	SBL_CHECK(cap_e < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "element capacity overflow");
	sbl_run_enumeration(c, k, cap_e);
	if (c->comm) if (const char *e = getenv("SBL_TEST_FAIL_SIMPLIFY_RANK")) if ((uint32_t)atoi(e) == c->comm->rank)      // test hook: this rank leaves the stage between two collectives
		throw SblError{SBL_ERR_OOM, "out of memory (SBL_TEST_FAIL_SIMPLIFY_RANK: this rank leaves the simplification stage)"};
	be.nid_ = c->bif_count;
	be.cap_e = (uint32_t)cap_e;

	// ---- graph arrays
	st->ch.ensure(cap_e); st->op.ensure(cap_e * 4); st->nx.ensure(cap_e * 4); st->pv.ensure(cap_e * 4);
	st->nodeof[0].ensure(cap_e * 4); st->nodeof[1].ensure(cap_e * 4);
	HIP_TRY(hipMemcpyAsync(st->ch.p, c->d_ch.p, E, hipMemcpyDeviceToDevice, s));
	HIP_TRY(hipMemcpyAsync(st->op.p, c->d_op.p, E * 4, hipMemcpyDeviceToDevice, s));
	k_init_links<<<nblocks(cap_e, 256), 256, 0, s>>>(st->nx.as<unsigned>(), st->pv.as<unsigned>(), st->nodeof[0].as<unsigned>(), st->nodeof[1].as<unsigned>(),
	                                                st->ch.as<uint8_t>(), E, cap_e);

	// ---- E2: instance lists in the reference's initial order
	sbl_compact_marks(c, 0);
	sbl_compact_marks(c, 1);
	size_t n0 = c->nmarks[0], n1 = c->nmarks[1], ninst = n0 + n1;
	c->stats.instances = ninst;
	size_t dense_max = DENSE_MAX_ELEMS;
	if (const char *e = getenv("SBL_DENSE_MAX_ELEMS")) dense_max = (size_t)atoll(e);
	if (force_dense) dense_max = std::max<size_t>(dense_max, DENSE_SWITCH_MAX_ELEMS);
	const bool dense = allow_dense && E <= dense_max && be.nid_ > 0 && getenv("SBL_NO_DENSE_PATH") == nullptr;
	// (a job on several GPUs never switches: the decision is timed, and the ranks must stay in step)
	be.may_try_dense = may_switch && !dense && !c->comm && E <= DENSE_SWITCH_MAX_ELEMS && getenv("SBL_NO_DENSE_PATH") == nullptr && getenv("SBL_NO_DENSE_SWITCH") == nullptr;
	be.max_iter_ = max_iter ? max_iter : 1u;
	// (the one-launch path cannot grow a pool and replay: low-complexity input makes hundreds of nodes per collapse, and 16 M nodes are 270 MB)
	size_t cap_n = dense ? std::max<size_t>(4 * ninst + (1u << 20), 16u << 20) : 4 * ninst + (1u << 20);
	if (dense) if (const char *e = getenv("SBL_TEST_DENSE_NODE_SLACK")) cap_n = ninst + (size_t)atoll(e);      // test hook: provoke the fall-back
	cap_n = std::max(cap_n, c->hint_cap_n);
	SBL_CHECK(cap_n < 0x7FFFFFF0ull, SBL_ERR_TOO_LARGE, "node capacity overflow");
	be.cap_n = (uint32_t)cap_n;
	st->nslot.ensure(cap_n * 4); st->nnext.ensure(cap_n * 4); st->nidst.ensure(cap_n * 4); st->nclr.ensure(cap_n * 4); st->ndead.ensure(cap_n);
	size_t nidp = (size_t)be.nid_ + 1;
	for (int t = 0; t < 2; t++) {
		st->head[t].ensure(nidp * 4); st->lsize[t].ensure(nidp * 4);
		HIP_TRY(hipMemsetAsync(st->head[t].p, 0xFF, nidp * 4, s));
		HIP_TRY(hipMemsetAsync(st->lsize[t].p, 0, nidp * 4, s));
	}
	size_t nmax = std::max(n0, n1);
	st->keys.ensure(nmax * 8 + 16); st->skeys.ensure(nmax * 8 + 16); st->selem.ensure(nmax * 4 + 16); st->iota.ensure(nmax * 4 + 16);
	st->nmark.ensure(cap_n * 4);
	for (int t = 0; t < 2; t++) st->maux[t].ensure(16);
	const unsigned ordbits = bits_of(2ull * E), idbits = bits_of(be.nid_);      // sort keys of id_bits + ordbits bits (48 on the benchmark workload: 6 radix passes, not 8)
	for (int t = 0; t < 2; t++) {
		unsigned n = c->nmarks[t];
		if (!n) continue;
		k_instance_keys<<<nblocks(n, 256), 256, 0, s>>>(c->d_melem[t].as<unsigned>(), c->d_mid[t].as<unsigned>(), n, (unsigned)t,
		                                               c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)E, ordbits, st->keys.as<unsigned long long>(), st->iota.as<unsigned>());
		sort_pairs64(c, st, st->keys.as<unsigned long long>(), st->skeys.as<unsigned long long>(), st->iota.as<unsigned>(), st->selem.as<unsigned>(), n, std::min(64u, idbits + ordbits));
		k_build_lists<<<nblocks(n, 256), 256, 0, s>>>(st->skeys.as<unsigned long long>(), st->selem.as<unsigned>(), c->d_melem[t].as<unsigned>(), n, t ? (unsigned)n0 : 0u, (unsigned)t, ordbits,
		                                             st->nslot.as<unsigned>(), st->nnext.as<unsigned>(), st->nidst.as<unsigned>(), st->ndead.as<uint8_t>(),
		                                             st->head[t].as<unsigned>(), st->lsize[t].as<unsigned>(), st->nodeof[t].as<unsigned>(), st->nmark.as<unsigned>());
		st->maux[t].ensure((size_t)n * 4 + 16);
		k_mark_aux<<<nblocks(n, 256), 256, 0, s>>>(c->d_melem[t].as<unsigned>(), n, (unsigned)t, c->d_sepidx.as<unsigned>(), c->nchr, c->d_ch.as<uint8_t>(), k, st->maux[t].as<unsigned>());
	}
	HIP_TRY(hipGetLastError());
	// positional order of the ids for the snapshot kernel
	st->perm.ensure(nidp * 4 + 16); st->permin.ensure(nidp * 4 + 16);
	st->keys.ensure(nidp * 8 + 16); st->skeys.ensure(nidp * 8 + 16);
	if (be.nid_) {
		k_id_position_keys<<<nblocks(be.nid_, 256), 256, 0, s>>>(st->head[0].as<unsigned>(), st->head[1].as<unsigned>(), st->nslot.as<unsigned>(), be.nid_,
		                                                       st->keys.as<unsigned long long>(), st->permin.as<unsigned>());
		sort_pairs64(c, st, st->keys.as<unsigned long long>(), st->skeys.as<unsigned long long>(), st->permin.as<unsigned>(), st->perm.as<unsigned>(), be.nid_, 32);      // keys are element slots, or 2^32 - 1
	}
	HIP_TRY(hipGetLastError());

	size_t be_maxn = 0;                                                // largest number of instances of an id (below)
	// ---- control state
	st->ctr.ensure(CTR_COUNT * 4);
	st->sel.ensure((1024 + 16) * 4);
	{
		// counter block and the header of the selection scratch (k_select_*: header + one count per chunk of ids; the kernels leave the header
		// reset) from a pinned staging buffer the context keeps: the copies are asynchronous and the host does not wait for them
		if (!st->h_init) HIP_TRY(hipHostMalloc((void **)&st->h_init, (CTR_COUNT + 4) * 4, hipHostMallocDefault));
		HIP_TRY(hipStreamSynchronize(s));                                  // (a previous stage's copies from the buffer have long completed; cheap when idle)
		unsigned *v = reinterpret_cast<unsigned *>(st->h_init);
		memset(v, 0, CTR_COUNT * 4);
		v[CTR_NE] = (unsigned)ne0; v[CTR_NN] = (unsigned)ninst; v[CTR_VIOL] = BT_NONE;
		v[CTR_COUNT] = SBL_NONE; v[CTR_COUNT + 1] = SBL_NONE; v[CTR_COUNT + 2] = 0u; v[CTR_COUNT + 3] = 0u;
		HIP_TRY(hipMemcpyAsync(st->ctr.p, v, CTR_COUNT * 4, hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(st->sel.p, v + CTR_COUNT, 16, hipMemcpyHostToDevice, s));
	}
	st->need.ensure(nidp); st->big.ensure(nidp); st->touch.ensure(nidp); st->own.ensure(nidp * 4);
	HIP_TRY(hipMemsetAsync(st->need.p, 0, nidp, s));
	HIP_TRY(hipMemsetAsync(st->big.p, 0, nidp, s));
	HIP_TRY(hipMemsetAsync(st->touch.p, 0, nidp, s));
	// scratch arena per window entry: window caches of ~16 instances (17 B per step, D + k + 2 steps) + FillVisit / Overlap
	// buffers + the AnyBulges map; ids that need more run alone in the big arena
	{
		// instances per id: the arena holds the window caches of an id with up to 1.5x the typical maximum (ids with
		// more -- repeat families -- run alone in the big arena)
		unsigned maxn = 0;
		HIP_TRY(hipMemsetAsync(st->ctr.as<unsigned>() + CTR_BIG, 0, 4, s));
		if (be.nid_) k_max_instances<<<256, 256, 0, s>>>(st->lsize[0].as<unsigned>(), st->lsize[1].as<unsigned>(), be.nid_, st->ctr.as<unsigned>() + CTR_BIG);
		HIP_TRY(hipMemcpyAsync(&maxn, st->ctr.as<unsigned>() + CTR_BIG, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		HIP_TRY(hipMemsetAsync(st->ctr.as<unsigned>() + CTR_BIG, 0, 4, s));
		be_maxn = maxn;
		size_t slots = std::min<size_t>(std::max<size_t>(16, maxn + maxn / 2), 2048);
		size_t ws = (size_t)D + k + 2;
		size_t need = slots * 17 * ws + 12 * (size_t)D + 8 * (size_t)k + (64u << 10) + slots * 64;
		be.arena_bytes = (uint32_t)std::min<size_t>(std::max<size_t>(need, 128u << 10), 1u << 30);
		be.snap_arena_bytes = be.arena_bytes;
		be.snap_threads = (uint32_t)std::max<size_t>(256, std::min<size_t>(256 * 32, (16ull << 30) / be.arena_bytes)) & ~7u;   // a multiple of the 8 XCDs
		be.big_arena_bytes = (uint32_t)std::min<size_t>(std::max<size_t>(256u << 20, 64 * be.arena_bytes), 0xFFFFFF00u);
	}
	uint32_t base_window = 14336;                                        // (swept again at the end of round 3: 86.1 ms against 86.9 ms at 16 384, 62 strains 4.12 against 4.19 s)
	if (const char *e = getenv("SBL_BASE_WINDOW")) base_window = (uint32_t)std::max(64, atoi(e));      // measurement switch (tools/sweep_window.sh)
	uint32_t window = c->window ? c->window : std::min<uint32_t>(base_window, std::max<uint32_t>(2048, be.nid_ / 64));
	window = std::min<uint32_t>(window, (1u << 20) - 1);
	window = (uint32_t)std::min<size_t>(window, std::max<size_t>(64, (24ull << 30) / be.arena_bytes));
	window = std::max<uint32_t>(1, std::min<uint32_t>(window, be.nid_ ? be.nid_ : 1));
	be.window = window;
	// the driver widens the window up to 4x while rounds are capacity-bound (simplify_driver.h); an explicit sbl_set_window pins it
	uint32_t window_max = c->window ? window : (uint32_t)std::min<size_t>((size_t)window * 4, std::max<size_t>(window, (48ull << 30) / be.arena_bytes));
	window_max = std::max<uint32_t>(window, std::min<uint32_t>(window_max, be.nid_ ? be.nid_ : 1));
	// parked transactions (GraphView::park_of): SBL_PARK=n collapses per launch and transaction (0: off)
	// (8 x 4.6 Mbp: 59.5 ms without, 56.9 with 2, 57.5 with 3, 61 with 1 -- every parked transaction costs its dependants a round; 62 strains,
	// where an id has dozens of instances and a transaction up to 28 collapses: 2.44 s without, 2.36 with 2, 2.18 with 3, 2.135 with 4, 2.15 with 5.
	// But ids with dozens of instances are also the regime of the serial chain -- one or two transactions per round, which chain() does not
	// enter while anything is parked, and something always is when every transaction parks seven times: 57 strains x 8 kbp at D = 369 took
	// 17 734 rounds, 238 s instead of seconds (tools/stress.py MANY=1, seed 67000; still exact).  Until chain() makes parking stop, it is OFF
	// by default there; SBL_PARK=4 is what the 2.15 s of profiles/r05_bench_config4.json were measured with.)
	// Round 6: the serial chain and parking compose (DeviceBackend::chain stops NEW parking, what is parked drains), so the many-instances
	// regime parks too -- with the cap that was measured best there (4: 2.135 s at 62 strains against 2.36 with 2) -- and so do the
	// replicated commits of a job on several GPUs (SBL_PARK_COMM=0: debugging switch, the round-5 behaviour).
	unsigned park_cap = ninst > 12 * (size_t)std::max<uint32_t>(1, be.nid_) ? 4u : 2u;
	if (const char *e = getenv("SBL_PARK")) park_cap = (unsigned)std::max(0, atoi(e));
	if (dense || (c->comm && getenv("SBL_PARK_COMM") && atoi(getenv("SBL_PARK_COMM")) == 0)) park_cap = 0;
	auto round_buffers = [&](uint32_t w) {
		st->win.ensure((size_t)w * 4 + 16);
		st->arena.ensure((size_t)w * be.arena_bytes * (park_cap ? 2u : 1u));      // (second half: the SHADOW slices -- where the entry of a window position works while its own slice holds a parked transaction)
		st->claims.ensure((size_t)w * (CLAIM_CAP + 1) * 4);
		st->live.ensure((size_t)w + 64);
		st->instbuf.ensure((size_t)w * 129 * 4);                          // (DeviceBackend::istride() <= 129)
		st->slice_busy.ensure(2 * ((size_t)w + 64));                      // (second half: the copy k_reserve takes for k_commit)
		st->park_list.ensure((size_t)w * 4 + 64);
	};
	be.snap_slice = window_max;
	if (!dense) {
		// a smaller or partly occupied GPU: first without the shadow slices (parking is a ~5 % optimisation, a 4 x smaller window is not),
		// then with the pinned window, which always was enough
		try { round_buffers(window_max); }
		catch (const SblError &) {
			(void)hipGetLastError();
			bool ok = false;
			if (park_cap) { park_cap = 0; try { round_buffers(window_max); ok = true; } catch (const SblError &) { (void)hipGetLastError(); } }
			if (!ok) {
				if (window_max == window) throw;
				window_max = window;
				be.snap_slice = window;
				round_buffers(window);
			}
		}
	}
	for (auto &e : st->ev) if (!e) HIP_TRY(hipEventCreate(&e));
	be.ev = st->ev;
	be.optimistic = optimistic;
	be.rsv_waves = ninst > 12 * (size_t)std::max<uint32_t>(1, be.nid_) ? 4u : 2u;      // instances per id: a handful, or dozens (many strains)
	if (const char *e = getenv("SBL_RSV_WAVES")) be.rsv_waves = (unsigned)std::min(4, std::max(1, atoi(e)));      // measurement switch
	if (getenv("SBL_PROBE_BIG_LDS") == nullptr) {
		// (the 512-slot table stays: a probe it cannot hold goes to the walking kernel, whose launch then lasts as long as a full probe -- 15 such
		// entries per round cost more than the 2 KB save; the instance and mark lists follow the input: largest instance count, mark density)
		be.pidx_inst = (unsigned)std::min<size_t>(256, std::max<size_t>(64, (be_maxn + 15) / 16 * 16));
		be.pidx_marks = ninst * 8 > E ? 192u : 64u;
	}
	// dozens of instances per id: a 1024-slot table.  With 512 slots 1.7 M of the 17 M probes of the 62-strain stage (2.35 M before the exact
	// bound in probe_idx) went on to the walking kernel for the table alone; with 1024 none does and the stage takes 2.04 s instead of 2.17 s.
	if (ninst > 12 * (size_t)std::max<uint32_t>(1, be.nid_)) be.pidx_vbits = 10;
	if (const char *e = getenv("SBL_PIDX_VBITS")) be.pidx_vbits = (unsigned)std::min(11, std::max(8, atoi(e)));      // measurement switch
	be.ev_phase = c->stage_seq++;
	be.prof = getenv("SBL_PHASES") ? (atoi(getenv("SBL_PHASES")) > 0 ? atoi(getenv("SBL_PHASES")) : 1) : 0;
	if (be.prof) sbl_commit_prof_reset();
	if (dense) be.use_index = false;                                   // (the one-launch path reads no index: nothing to maintain)
	be.idx_nblk = be.use_index ? (uint32_t)((E + 63) / 64) : 0u;
	if (be.idx_nblk) st->bidx.ensure((size_t)be.idx_nblk * BT_IDX_WORDS * 8);
	be.bind();
	be.g.k = k; be.g.D = D;
	{
		const unsigned cap = park_cap;
		be.g.park_cap = cap; be.g.park_of = nullptr; be.g.slice_busy = nullptr; be.g.shadow_base = 0; be.park_slices = 0;
		be.g.park_hold = 0; be.g.any_parked = 0; be.g.park_list = nullptr;
		if (cap) {
			if (!st->park_stream) { HIP_TRY(hipStreamCreateWithFlags(&st->park_stream, hipStreamNonBlocking)); for (auto &e : st->park_ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
			st->park_of.ensure(((size_t)be.nid_ + 1) * 4);
			be.park_slices = (size_t)window_max + 64;
			be.g.park_of = st->park_of.as<unsigned>(); be.g.slice_busy = st->slice_busy.as<uint8_t>(); be.g.shadow_base = window_max;
			be.g.park_list = st->park_list.as<unsigned>();
		}
	}
	if (((D + k + 2u + 126u) >> 6) > 16u) be.g.idx_probe = 0;           // windows of more than 16 blocks: k_probe_idx could serve nobody (every entry walks, as before round 5)
	be.g.test_flags = getenv("SBL_TEST_FLAGS") ? (unsigned)atoi(getenv("SBL_TEST_FLAGS")) : 0u;
	be.g.lazy_rescan = getenv("SBL_EAGER_RESCAN") ? 0u : 1u;            // measurement switch: dirty windows rescanned right after every collapse (round 3)
	be.g.collapse_g = getenv("SBL_OLD_COLLAPSE") ? 0u : 1u;            // measurement switch: the round-3 collapse (a chain of dependent round trips) instead of the gather-first one
	be.g.ab_estimate = getenv("SBL_NO_AB_ESTIMATE") ? 0u : 1u;         // measurement switch: AnyBulges of big ids without its counting pass
	be.g.jscan_rounds = getenv("SBL_NO_JSCAN_ROUNDS") ? 0u : 1u;       // measurement switch
	be.g.probe_pre = getenv("SBL_NO_PROBE_PRE") ? 0u : 1u;              // measurement switch: the endChar pre-pass of the probe (probe_endchars)
	be.g.lazy_map = getenv("SBL_EAGER_MAP") ? 0u : 1u;                  // measurement switch: the Boost-ordered map of AnyBulges built eagerly (round 3)
	be.g.tstamp = nullptr; be.g.tslot = 0;
	be.g.sep = c->d_sepidx.as<unsigned>(); be.g.nsep = c->nchr + 1; be.g.norig = (uint32_t)E;
	if (getenv("SBL_SEP_BY_CHAR")) be.g.sep = nullptr;                   // measurement switch: separators recognised by their character everywhere
	be.index_build();
	if (!dense && be.phase_events) be.stamps_init();
	HIP_TRY(hipEventRecord(c->ev[3], s));

	// ---- SimplifyGraph
	SimplifyReport rep;
	if (dense) {
		// tiny input: for iteration, for id, RemoveBulges(id) in one launch (k_dense_stage)
		be.g.lazy_min = 1;                                            // every id keeps full-size mark lists and takes lazy windows
		st->big_arena.ensure(be.big_arena_bytes);
		unsigned *d_out = st->ctr.as<unsigned>() + CTR_DETAIL;        // (the violation-detail words are unused here)
		HIP_TRY(hipEventRecord(be.ev[2], s));
		k_dense_stage<<<1, 64, 0, s>>>(be.g, st->big_arena.as<uint8_t>(), be.big_arena_bytes, max_iter, d_out);
		HIP_TRY(hipEventRecord(be.ev[3], s));
		HIP_TRY(hipGetLastError());
		be.read_ctr();
		float ms = 0;
		HIP_TRY(hipEventElapsedTime(&ms, be.ev[2], be.ev[3]));
		be.commit_ms = ms;
		if (st->h_ctr[CTR_ERR]) {
			if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] one-launch path: capacity error %u, falling back to the ordered rounds\n", st->h_ctr[CTR_ERR]);
			return RUN_DENSE_FAILED;
		}
		rep.iterations = st->h_ctr[CTR_DETAIL]; rep.bulges = st->h_ctr[CTR_BULGES]; rep.transactions = rep.executed = st->h_ctr[CTR_TXN];
		rep.chain_transactions = rep.transactions;
		if (progress) {                                               // the reference's callback sequence (blockfinder.cpp:23-48), delivered after the launch
			progress(0, SBL_PROGRESS_START, user);
			const uint64_t threshold = ((uint64_t)be.nid_ * max_iter) / 50, processed = (uint64_t)rep.iterations * ((uint64_t)be.nid_ + 1);
			const uint64_t due = threshold ? processed / threshold : processed;
			uint64_t tp = 0;
			for (uint64_t i = 0; i < due; i++) { tp = std::min<uint64_t>(tp + 1, 50); progress((size_t)tp, SBL_PROGRESS_RUN, user); }
			progress(50, SBL_PROGRESS_END, user);
		}
	} else {
		try { rep = simplify_graph(be, max_iter, window, progress, user, window_max); }
		catch (const RestartStage &) { HIP_TRY(hipStreamSynchronize(s)); return RUN_RESTART; }
		catch (const TryDense &) { HIP_TRY(hipStreamSynchronize(s)); return RUN_TRY_DENSE; }
	}
	HIP_TRY(hipEventRecord(c->ev[4], s));
	if (!dense && be.phase_events) be.stamps_collect();
	be.snapshots_collect();
	if (be.g.bidx && (be.g.test_flags & 32u)) sbl_rounds_stats_report();      // SBL_TEST_FLAGS=32: what the block index served
	if (be.g.bidx && getenv("SBL_CHECK_INDEX")) {                        // test switch: the maintained block index against a rebuild
		unsigned init[2] = {0u, BT_NONE}, res[2];
		unsigned *d_out = st->ctr.as<unsigned>() + CTR_DETAIL + 16;
		HIP_TRY(hipMemcpyAsync(d_out, init, sizeof init, hipMemcpyHostToDevice, s));
		k_check_blkidx<<<(be.idx_nblk + 3) / 4, 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), st->pv.as<unsigned>(), c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>(),
		                                                    st->wmax.as<unsigned>(), be.g.norig, be.idx_nblk, st->bidx.as<unsigned long long>(), d_out);
		HIP_TRY(hipMemcpyAsync(res, d_out, sizeof res, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] block index check: %u of %u blocks differ from a rebuild\n", res[0], be.idx_nblk);
		if (res[0]) { char b[160]; snprintf(b, sizeof b, "block index out of date: %u of %u blocks differ from a rebuild (first: block %u)", res[0], be.idx_nblk, res[1]); throw SblError{SBL_ERR_INTERNAL, b}; }
	}

	// ---- T3: copy-back (reference src/blockfinder.cpp:85-95): linearise the list into the dense state arrays
	be.read_ctr();
	unsigned ne = st->h_ctr[CTR_NE];
	st->flag.ensure((size_t)ne * 4 + 16); st->segidx.ensure((size_t)ne * 4 + 16); st->newidx.ensure((size_t)ne * 4 + 16);
	k_seg_flags<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>());
	scan_u32(c, st, st->flag.as<unsigned>(), st->segidx.as<unsigned>(), (size_t)ne + 1);
	unsigned nseg = 0;
	HIP_TRY(hipMemcpyAsync(&nseg, st->segidx.as<unsigned>() + ne, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	SBL_CHECK(nseg >= 1, SBL_ERR_INTERNAL, "copy-back: empty list");
	st->seg_head.ensure((size_t)nseg * 4); st->seg_len.ensure((size_t)nseg * 4); st->seg_succ_elem.ensure((size_t)nseg * 4);
	for (int t = 0; t < 2; t++) { st->succ[t].ensure((size_t)nseg * 4); st->dist[t].ensure((size_t)nseg * 8); }
	k_seg_tails<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), st->nx.as<unsigned>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(),
	                                            st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>());
	k_seg_finish<<<nblocks(nseg, 256), 256, 0, s>>>(nseg, st->seg_head.as<unsigned>(), st->seg_len.as<unsigned>(), st->seg_succ_elem.as<unsigned>(),
	                                               st->flag.as<unsigned>(), st->segidx.as<unsigned>(), st->succ[0].as<unsigned>(), st->dist[0].as<unsigned long long>());
	int cur = 0;
	for (unsigned span = 1; span < nseg; span <<= 1, cur ^= 1)
		k_seg_jump<<<nblocks(nseg, 256), 256, 0, s>>>(nseg, st->succ[cur].as<unsigned>(), st->dist[cur].as<unsigned long long>(),
		                                             st->succ[cur ^ 1].as<unsigned>(), st->dist[cur ^ 1].as<unsigned long long>());
	// segment 0 starts with element 0 (the first '$'), the head of the whole list: dist[0] = new total length
	unsigned long long total = 0;
	HIP_TRY(hipMemcpyAsync(&total, st->dist[cur].p, 8, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	size_t Enew = (size_t)total, Epad = (Enew + 31) / 32 * 32 + 64;
	st->ch_out.ensure(Epad); st->op_out.ensure(Enew * 4 + 16);
	k_scatter_linear<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), st->op.as<unsigned>(), ne, st->flag.as<unsigned>(), st->segidx.as<unsigned>(),
	                                                 st->seg_head.as<unsigned>(), st->dist[cur].as<unsigned long long>(), total,
	                                                 st->ch_out.as<uint8_t>(), st->op_out.as<unsigned>(), st->newidx.as<unsigned>());
	k_fill_bytes<<<nblocks(Epad - Enew, 256), 256, 0, s>>>(st->ch_out.as<uint8_t>(), (uint8_t)'$', Enew, Epad);
	k_remap_seps<<<nblocks(c->nchr + 1, 64), 64, 0, s>>>(st->newidx.as<unsigned>(), c->d_sepidx.as<unsigned>(), c->nchr + 1);
	k_sep_positions<<<nblocks(c->nchr, 64), 64, 0, s>>>(c->d_sepidx.as<unsigned>(), c->nchr, st->op_out.as<unsigned>());
	HIP_TRY(hipGetLastError());
	c->stats.dict_checked = 0; c->stats.dict_mismatches = 0;
	if (getenv("SBL_CHECK_DICTIONARY") && c->dict_keys && k <= 32) {
		// the reference's own invariant (IndexedSequence::Test) on the stage's final graph, see k_dict_check
		st->scantmp.ensure(64);
		unsigned long long init[6] = {0, 0, ~0ull, 0, 0, 0}, res[6];
		HIP_TRY(hipMemcpyAsync(st->scantmp.p, init, sizeof init, hipMemcpyHostToDevice, s));
		if (getenv("SBL_TEST_CORRUPT_MARK")) {                              // test hook: the check must notice ONE wrong mark among hundreds of millions
			const unsigned wrong = 0;
			HIP_TRY(hipMemcpyAsync(c->d_bif[0].as<unsigned>() + (c->sepidx[0] + 1 + (size_t)atoll(getenv("SBL_TEST_CORRUPT_MARK"))), &wrong, 4, hipMemcpyHostToDevice, s));
		}
		k_dict_check<<<nblocks(ne, 256), 256, 0, s>>>(st->ch.as<uint8_t>(), ne, st->newidx.as<unsigned>(), st->ch_out.as<uint8_t>(), total, c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>(),
		                                             c->dict_keys, be.nid_, k, st->scantmp.as<unsigned long long>());
		HIP_TRY(hipMemcpyAsync(res, st->scantmp.p, sizeof res, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		c->stats.dict_checked = res[0]; c->stats.dict_mismatches = res[1];
		if (getenv("SBL_TRACE")) fprintf(stderr, "[sbl] dictionary invariant (IndexedSequence::Test): %llu windows checked, %llu mismatches\n", res[0], res[1]);
		if (res[1] && !getenv("SBL_TEST_CORRUPT_MARK")) {
			char b[256];
			snprintf(b, sizeof b, "dictionary invariant violated (IndexedSequence::Test): %llu of %llu windows; first at slot %llu strand %llu: stored id %llu, the dictionary says %llu",
			         res[1], res[0], res[2], res[3], res[4], res[5]);
			throw SblError{SBL_ERR_INTERNAL, b};
		}
	}
	std::swap(c->d_ch, st->ch_out);
	std::swap(c->d_op, st->op_out);
	HIP_TRY(hipMemcpyAsync(c->sepidx.data(), c->d_sepidx.p, (size_t)(c->nchr + 1) * 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipEventRecord(c->ev[5], s));
	HIP_TRY(hipStreamSynchronize(s));
	c->nelem = Enew;

	float ms_enum = 0, ms_simp = 0, ms_copy = 0;
	HIP_TRY(hipEventElapsedTime(&ms_enum, c->ev[2], c->ev[3]));
	HIP_TRY(hipEventElapsedTime(&ms_simp, c->ev[3], c->ev[4]));
	HIP_TRY(hipEventElapsedTime(&ms_copy, c->ev[4], c->ev[5]));
	c->stats.enumerate_ms = ms_enum; c->stats.simplify_ms = ms_simp; c->stats.copyback_ms = ms_copy;
	c->stats.total_ms = ms_enum + ms_simp + ms_copy;
	c->stats.bulges = rep.bulges; c->stats.iterations = rep.iterations; c->stats.rounds = rep.rounds; c->stats.replays = rep.replays; c->stats.grow_replays = rep.grow_replays;
	c->stats.snapshot_ms = be.snapshot_ms; c->stats.reserve_ms = be.reserve_ms; c->stats.commit_ms = be.commit_ms; c->stats.probe_ms = be.probe_ms;
	c->stats.commit_event_ms = be.commit_event_ms; c->stats.commit_event_launches = be.commit_event_launches;
	c->stats.verdict_ms = be.ro_ms; c->stats.ro_ranks = be.split_ro() ? c->comm->n : 1;
	c->stats.executed = rep.executed; c->stats.transactions = rep.transactions; c->stats.chain_transactions = rep.chain_transactions;
	c->stats.device_bytes = sbl_devbuf_total().load();
	if (be.prof) sbl_commit_prof_report(be.ts_round);
	*bulges = rep.bulges;
	return RUN_DONE;
}
