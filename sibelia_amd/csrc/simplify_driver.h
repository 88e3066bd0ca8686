// simplify_driver.h -- host orchestration of SimplifyGraph (reference src/blockfinder.cpp:16-51).
//
// The driver only sequences launches and reads a handful of counters back per round; all graph
// work is done by the backend's kernels (simplify.hip).  It is a template so that tests/hostsim can
// drive the very same control flow over a host-memory backend.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "simplify_steps.h"

struct SimplifyCounters {
	uint32_t v[CTR_COUNT];
};

struct SimplifyReport {
	uint64_t bulges = 0;
	uint32_t iterations = 0, rounds = 0, replays = 0, solo = 0, grow_replays = 0;
	uint64_t executed = 0, transactions = 0;
	uint64_t chain_transactions = 0;     // ... of which run by the serial chain (dense conflict neighbourhoods)
};

// Backend concept:
//   uint32_t nid();                                  number of bifurcation ids
//   void checkpoint(); void restore();               iteration-level copy of every mutable array
//   void snapshot_all(bool incremental);             need[id] = AnyBulges verdict, for every id (incremental: touched ids only)
//   void reset_round_state(bool stamps_too);         own/lock = 0xFFFFFFFF (and rmax/wmax = 0)
//   void clear_counters();                           ctr[ERR, BULGES, VIOL, BIG, COMMITTED] reset (VIOL = NONE)
//   void select_launch(lo, limit, W);                lowest pending ids in [lo, limit]: issued right behind a round's last launch, so that
//   void select_read(&nwin, &newlo, &solo);          ONE host round trip (counters()) brings back the round's counters and the next window
//   void probe(nwin, round);                         retire window entries whose verdict is false now, flag the others
//   void mark_live(nwin);                            flag every window entry live (solo rounds skip the probe)
//   void reserve(nwin, round); void commit(nwin, round, solo);   (flagged entries only)
//   bool chain(nwin, round);                         run what is pending in the id range of the window one after the other, in order
//                                                    (instead of reserve + commit; false: not supported)
//   SimplifyCounters counters();                     device -> host
//   bool grow(uint32_t err);                         enlarge element / node capacity after BT_ERR_*_CAP
// window: ids examined per ordered round; window_max > window lets the driver widen it while a round is CAPACITY-bound
// (the backend's buffers must hold window_max entries).
template <class Backend>
SimplifyReport simplify_graph(Backend &be, uint32_t max_iter, uint32_t window, sbl_progress_fn progress, void *user, uint32_t window_max = 0)
{
	SimplifyReport rep;
	const bool trace = getenv("SBL_TRACE") != nullptr;
	const bool use_chain = getenv("SBL_NO_CHAIN") == nullptr;          // debugging / measurement switch
	const uint32_t nid = be.nid();
	const uint64_t per_iter = (uint64_t)nid + 1;                      // ids 0 .. GetMaxId() inclusive
	const uint64_t threshold = ((uint64_t)nid * max_iter) / 50;       // PROGRESS_STRIDE, blockfinder.cpp:28
	uint64_t progress_calls = 0, total_progress = 0;
	auto report_progress = [&](uint64_t processed) {
		if (!progress) return;
		uint64_t due = threshold ? processed / threshold : processed;
		for (; progress_calls < due; progress_calls++) {
			total_progress = std::min<uint64_t>(total_progress + 1, 50);
			progress((size_t)total_progress, SBL_PROGRESS_RUN, user);
		}
	};
	if (progress) progress(0, SBL_PROGRESS_START, user);
	if (window == 0) window = 1;
	if (window > (1u << 20) - 1) window = (1u << 20) - 1;
	if (window_max < window) window_max = window;
	if (window_max > (1u << 20) - 1) window_max = (1u << 20) - 1;
	if (getenv("SBL_FIXED_WINDOW")) window_max = window;              // measurement switch
	double widen = 0.15;                                              // share of the base window that may stay blocked before the window widens
	if (const char *e = getenv("SBL_WIDEN")) widen = atof(e);         // measurement switch
	do {
		rep.iterations++;
		if (nid) {
			be.checkpoint();
			std::vector<uint32_t> fences;                                 // ids that must not be overtaken
			uint64_t iter_bulges = 0;
			for (;;) {                                                    // replay loop
				bool replay = false;
				be.snapshot_all(rep.iterations > 1);
				be.reset_round_state(true);
				be.clear_counters();
				std::sort(fences.begin(), fences.end());
				uint32_t lo = 0, round = 0;
				size_t fi = 0;
				uint32_t prev_txn = 0, prev_done = 0, starved = 0;
				bool chain_mode = false;
				uint32_t wcur = window;
				auto next_limit = [&]() { while (fi < fences.size() && fences[fi] < lo) fi++; return fi < fences.size() ? fences[fi] : nid - 1; };
				uint32_t limit = next_limit();
				be.select_launch(lo, limit, wcur);
				for (;;) {
					uint32_t nwin = 0, newlo = lo, solo = 0;
					be.select_read(&nwin, &newlo, &solo);
					if (nwin == 0) {
						if (limit >= nid - 1) break;
						lo = limit + 1;                                       // the fence's turn has passed
						limit = next_limit();
						be.select_launch(lo, limit, wcur);
						continue;
					}
					lo = newlo;
					if (++round > SS_ROUND_MAX) { be.reset_round_state(false); round = 1; }
					rep.rounds++;
					if (solo) { rep.solo++; be.mark_live(nwin); }       // a big id runs alone but still claims its neighbourhood
					else be.probe(nwin, round);
					const bool chained = !solo && chain_mode && be.chain(nwin, round);
					if (!chained) {
						be.reserve(nwin, round);
						be.commit(nwin, round, solo != 0);
					}
					// the next window is selected behind this round's last launch, BEFORE the host looks at the round: one host round trip per
					// round instead of two (its width is the one the previous round's outcome suggested)
					const uint32_t this_limit = limit;
					limit = next_limit();
					be.select_launch(lo, limit, wcur);
					SimplifyCounters c = be.counters();
					if (trace) fprintf(stderr, "[sbl] iter %u round %u lo %u limit %u nwin %u solo %u committed %u bulges %u big %u viol %d err %u\n",
					                   rep.iterations, round, lo, this_limit, nwin, solo, c.v[CTR_COMMITTED], c.v[CTR_BULGES], c.v[CTR_BIG], (int)c.v[CTR_VIOL], c.v[CTR_ERR]);
					if (trace && c.v[CTR_VIOL] != BT_NONE) fprintf(stderr, "[sbl] violation detail: kind %u resource %u other %u id %u info %u\n", c.v[CTR_DETAIL], c.v[CTR_DETAIL + 1], c.v[CTR_DETAIL + 2], c.v[CTR_DETAIL + 3], c.v[CTR_DETAIL + 4]);
					if (c.v[CTR_ERR]) {
						if (!be.grow(c.v[CTR_ERR])) throw SblError{SBL_ERR_INTERNAL, "bulge removal: unrecoverable capacity error"};
						replay = true; rep.grow_replays++;
					} else if (c.v[CTR_VIOL] != BT_NONE) {
						fences.push_back(c.v[CTR_VIOL]);
						replay = true;
					}
					if (replay) break;
					report_progress((uint64_t)(rep.iterations - 1) * per_iter + lo);
					// Dense conflict neighbourhoods (small k, low-complexity sequence): nearly every live entry shares a claim
					// with a lower one, a round commits one or two transactions and costs four launches.  Once most of the
					// window stays blocked behind one or two transactions twice in a row, the rest of the iteration runs in chain mode: the probe still retires
					// the clean entries in parallel, the live ones are then run one after the other, in order (k_chain).
					const uint32_t txn = c.v[CTR_TXN] - prev_txn, retired = c.v[CTR_COMMITTED] - prev_done;
					const uint32_t blocked = nwin > retired ? nwin - retired : 0;
					prev_txn = c.v[CTR_TXN]; prev_done = c.v[CTR_COMMITTED];
					// Early in an iteration most of a window stays blocked behind lower neighbours: the number of rounds is the dependency
					// depth and a larger window only adds probes and reservations (every blocked live entry reserves again next round).
					// Later nearly every entry retires (the probe finds it clean, or it owns its neighbourhood): the round is capacity-bound
					// and its fixed costs -- the latency of the slowest transaction, five launches, two host round trips -- buy `window`
					// retirements.  Widen while fewer than ~15 % of the base window stay blocked (measured optimum: 0.08 .. 0.25 within 1 %).
					if (window_max > window && !solo && nwin) {
						const double f = (double)blocked / (double)nwin;
						const double want = f > 0 ? (widen * window) / f : (double)window_max;
						wcur = want >= (double)window_max ? window_max : want <= (double)window ? window : (uint32_t)want;
					}
					if (chained) rep.chain_transactions += txn;
					else if (!solo && blocked >= 8 && txn <= 2) {       // absolute: a handful of slow transactions per round still beat one wave
						if (++starved >= 2 && use_chain && !chain_mode) {
							chain_mode = true;
							if (trace) fprintf(stderr, "[sbl] iter %u: chain mode from id %u\n", rep.iterations, lo);
						}
					}
					else starved = 0;
				}
				if (!replay) {
					SimplifyCounters c = be.counters();
					iter_bulges = c.v[CTR_BULGES];
					rep.executed += c.v[CTR_COMMITTED];
					rep.transactions += c.v[CTR_TXN];
					break;
				}
				rep.replays++;
				be.restore();
			}
			rep.bulges += iter_bulges;
		}
		report_progress((uint64_t)rep.iterations * per_iter);
	} while (rep.bulges > 0 && rep.iterations < max_iter);
	if (progress) progress(50, SBL_PROGRESS_END, user);
	return rep;
}
