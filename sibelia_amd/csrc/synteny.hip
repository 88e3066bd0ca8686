// synteny.hip -- downstream of the hot path: BlockFinder::GenerateSyntenyBlocks behind the C ABI.
//
// Replaces GenerateSyntenyBlocks / ResolveOverlap / TrimBlocks (reference src/synteny.cpp:229-286, :124-166, :31-122).  The stage
// re-enters the hot path twice: the edge list of a fresh index at k (sbl_list_edges: enumeration + ListEdges kernels) and, for
// every candidate block, a fresh index at trimK over the block's ORIGINAL sequences (a child context on the same GPU: the
// block ranges are gathered device-to-device from the original records kept at load time, sanitised through the parent's
// rand() stream and enumerated by the same kernels).  What stays on the host is the reference's sequential bookkeeping:
// the three std::sort calls whose handling of equal elements the result depends on (common.h:153, synteny.cpp:249,:254 --
// the same libstdc++ calls on the same element order), the occupancy indicators in original coordinates, and the per-block
// search for the outermost shared bifurcations, here on flat arrays (CSR instance lists, per-chromosome occupancy bytes
// with an undo list instead of a std::set of positions).
#include <cstring>
#include <algorithm>
#include <numeric>

#include "sbl_ctx.h"
#include "kmer_kernels.h"       // rc_code, mask_is_bifurcation

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

struct GatherDesc { unsigned long long src, dst; unsigned long long len; };
// child element array: '$' b0 '$' b1 '$' ...  (pre-filled with '$'); one workgroup-stride loop per block range
__global__ void __launch_bounds__(256) k_gather_blocks(const uint8_t *__restrict__ orig, const GatherDesc *__restrict__ desc, unsigned nrec, uint8_t *__restrict__ out)
{
	for (unsigned r = blockIdx.y; r < nrec; r += gridDim.y) {
		const GatherDesc d = desc[r];
		for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < d.len; i += (unsigned long long)gridDim.x * blockDim.x)
			out[d.dst + i] = orig[d.src + i];
	}
}

// ---- index over a SMALL block (TrimBlocks at small k: tens of thousands of candidate blocks of a few dozen bases) -------------
// The general child index costs ~38 launches and ~10 host synchronisations whatever its size.  Blocks of at most TE_MAX_ELEM
// elements in at most TE_MAX_REC sequences, trimK <= 32, are indexed by ONE workgroup in ONE launch instead: gather from the
// original records, canonical codes, LDS table of the distinct k-mers with their merged neighbour masks, Bifurcation() test, id =
// rank of the code among the bifurcation codes of both orientations, instance lists in element order -- the definitions of
// kmer_bucket_kernels.h (k_kmer_records / k_bucket_classify / k_scatter_ids / k_scatter_members / k_make_instances) applied in LDS.
// The result goes straight into mapped host memory.  A character other than A, C, G, T (the general path draws rand() for
// those) makes the kernel decline (status 1) and the caller takes the general path.
#define TE_MAX_ELEM 1024u
#define TE_MAX_REC 16u
#define TE_SLOTS 2048u
#define TE_THREADS 256u
struct TinyDesc { unsigned nrec, k; unsigned long long src[TE_MAX_REC]; unsigned len[TE_MAX_REC]; };
struct TinyOut { unsigned status, bif_count, n[2], pad[4]; unsigned inst[2][TE_MAX_ELEM * 3]; };
// head: status, bif_count, instances per strand; inst0 / inst1: (id, sequence, position) triples of the + / - strand
__device__ __forceinline__ void tiny_enumerate_body(const uint8_t *__restrict__ orig, const TinyDesc &d, unsigned *__restrict__ head,
                                                    unsigned *__restrict__ inst0, unsigned *__restrict__ inst1)
{
	__shared__ uint8_t ch[TE_MAX_ELEM + 8];
	__shared__ unsigned short pslot[TE_MAX_ELEM];             // table slot of the k-mer starting at g | orientation flags << 12 ... see below
	__shared__ uint8_t pfl[TE_MAX_ELEM];
	__shared__ unsigned sep[TE_MAX_REC + 1];
	__shared__ unsigned long long tkey[TE_SLOTS];
	__shared__ unsigned tmask[TE_SLOTS];
	__shared__ unsigned short taux[TE_SLOTS];
	__shared__ unsigned long long keys[TE_SLOTS];
	__shared__ unsigned short payload[TE_SLOTS], pairids[TE_SLOTS];
	__shared__ unsigned s_bad, s_pairs, s_keys, s_wsum[2][TE_THREADS / 64];
	const unsigned tid = threadIdx.x, k = d.k, lane = tid & 63u, wv = tid >> 6;
	if (tid == 0) {
		unsigned e = 0;
		for (unsigned r = 0; r < d.nrec; r++) { sep[r] = e; e += d.len[r] + 1; }
		sep[d.nrec] = e;
		s_bad = 0; s_pairs = 0; s_keys = 0;
	}
	for (unsigned i = tid; i < TE_SLOTS; i += TE_THREADS) { tkey[i] = ~0ull; tmask[i] = 0; taux[i] = 0xFFFFu; }
	__syncthreads();
	const unsigned E = sep[d.nrec] + 1;
	// ---- gather: '$' b0 '$' b1 '$' ...
	for (unsigned e = tid; e < E; e += TE_THREADS) {
		unsigned r = 0;
		while (r + 1 <= d.nrec && sep[r + 1] <= e) r++;                                  // sep[r] <= e < sep[r + 1] (r == nrec: the last separator)
		uint8_t c = '$';
		if (r < d.nrec && e != sep[r]) {
			c = orig[d.src[r] + (e - sep[r] - 1)];
			if (c != 'A' && c != 'C' && c != 'G' && c != 'T') s_bad = 1;
		}
		ch[e] = c;
	}
	__syncthreads();
	if (s_bad) { if (tid == 0) head[0] = 1; return; }
	auto base = [&](unsigned e) -> unsigned { unsigned x = (ch[e] >> 1) & 3u; return x ^ (x >> 1); };      // A:0 C:1 G:2 T:3 (k_pack2bit)
	// ---- records: canonical code, neighbour masks in canonical orientation (k_kmer_records), merged per distinct k-mer in the LDS table
	for (unsigned g = tid; g < E; g += TE_THREADS) {
		unsigned short slot = 0xFFFFu;
		uint8_t fl = 0;
		bool valid = g + k < E;                                                          // the element after the window exists (the last one is '$')
		for (unsigned i = 0; valid && i < k; i++) if (ch[g + i] == '$') valid = false;
		if (valid) {
			unsigned long long fwd = 0, rev = 0;
			for (unsigned i = 0; i < k; i++) {
				const unsigned long long b = base(g + i);
				fwd = (fwd << 2) | b;
				rev |= (3ull - b) << (2 * i);
			}
			const unsigned ps = ch[g - 1] == '$' ? 4u : base(g - 1), ns = ch[g + k] == '$' ? 4u : base(g + k);
			unsigned m = 0;
			if (fwd <= rev) { m |= (1u << ps) | (1u << (8 + ns)); fl |= 1u; }
			if (rev <= fwd) { m |= (1u << (ns == 4 ? 4 : 3 - ns)) | (1u << (8 + (ps == 4 ? 4 : 3 - ps))); fl |= 2u; }
			const unsigned long long canon = fwd < rev ? fwd : rev;
			unsigned h = (unsigned)((canon * 0x9E3779B97F4A7C15ull) >> 53) & (TE_SLOTS - 1);
			for (;;) {
				unsigned long long old = atomicCAS(&tkey[h], ~0ull, canon);
				if (old == ~0ull || old == canon) { atomicOr(&tmask[h], m); break; }
				h = (h + 1) & (TE_SLOTS - 1);
			}
			slot = (unsigned short)h;
		}
		pslot[g] = slot; pfl[g] = fl;
	}
	__syncthreads();
	// ---- Bifurcation() per distinct k-mer; the codes of both orientations (a palindrome once) are the keys of the ranking
	for (unsigned h = tid; h < TE_SLOTS; h += TE_THREADS) {
		if (tkey[h] == ~0ull || !mask_is_bifurcation(tmask[h])) continue;
		const unsigned long long canon = tkey[h], r = rc_code(canon, k);
		const unsigned nk = r == canon ? 1u : 2u;
		const unsigned lp = atomicAdd(&s_pairs, 1u), lk = atomicAdd(&s_keys, nk);
		keys[lk] = canon; payload[lk] = (unsigned short)(2 * lp);
		if (nk == 2) { keys[lk + 1] = r; payload[lk + 1] = (unsigned short)(2 * lp + 1); }
		taux[h] = (unsigned short)lp;
	}
	__syncthreads();
	const unsigned nkeys = s_keys;
	for (unsigned i = tid; i < nkeys; i += TE_THREADS) {                                 // id = rank of the code (the keys are distinct)
		const unsigned long long mine = keys[i];
		unsigned rank = 0;
		for (unsigned j = 0; j < nkeys; j++) rank += keys[j] < mine;
		const unsigned p = payload[i];
		pairids[p] = (unsigned short)rank;
		if (!(p & 1u) && rc_code(mine, k) == mine) pairids[p + 1] = (unsigned short)rank;
	}
	__syncthreads();
	// ---- instance lists in element order: + strand element g, - strand element g + k - 1 (k_scatter_members / k_make_instances)
	const unsigned per = (E + TE_THREADS - 1) / TE_THREADS, g0 = tid * per, g1 = g0 + per < E ? g0 + per : E;
	unsigned cnt = 0;
	for (unsigned g = g0; g < g1; g++) cnt += pslot[g] != 0xFFFFu && taux[pslot[g]] != 0xFFFFu;
	unsigned incl = cnt;
#pragma unroll
	for (int dd = 1; dd < 64; dd <<= 1) { unsigned x = __shfl_up(incl, dd); if (lane >= (unsigned)dd) incl += x; }
	if (lane == 63) s_wsum[0][wv] = incl;
	__syncthreads();
	unsigned off = incl - cnt, total = 0;
	for (unsigned w = 0; w < TE_THREADS / 64; w++) { if (w < wv) off += s_wsum[0][w]; total += s_wsum[0][w]; }
	for (unsigned g = g0; g < g1; g++) {
		if (pslot[g] == 0xFFFFu) continue;
		const unsigned lp = taux[pslot[g]];
		if (lp == 0xFFFFu) continue;
		const unsigned p = 2 * lp + ((pfl[g] & 1u) ? 0u : 1u);
		unsigned r = 0;
		while (sep[r + 1] <= g) r++;
		inst0[3 * off] = pairids[p]; inst0[3 * off + 1] = r; inst0[3 * off + 2] = g - sep[r] - 1;
		const unsigned e1 = g + k - 1;
		inst1[3 * off] = pairids[p ^ 1u]; inst1[3 * off + 1] = r; inst1[3 * off + 2] = sep[r + 1] - 1 - e1;
		off++;
	}
	if (tid == 0) { head[1] = nkeys; head[2] = total; head[0] = 0; }
}
__global__ void __launch_bounds__(TE_THREADS) k_tiny_enumerate(const uint8_t *__restrict__ orig, TinyDesc d, TinyOut *__restrict__ out)
{
	__shared__ unsigned head[4];
	tiny_enumerate_body(orig, d, head, out->inst[0], out->inst[1]);
	__syncthreads();
	if (threadIdx.x == 0) { if (head[0] == 0) { out->bif_count = head[1]; out->n[0] = head[2]; out->n[1] = head[2]; } out->status = head[0]; }
}
// The same for MANY candidate blocks in one launch, one workgroup each: TrimBlocks is called once per candidate block, and on a raw
// graph at small k (-v / --allstages call GenerateSyntenyBlocks(k, k, k) before every stage) that is tens of thousands of blocks of a
// few dozen bases -- one launch and one host synchronisation each cost more than the index itself.  The caller speculates the
// blocks of a whole chunk of groups (sbl_generate_blocks) and reads all results back with one copy.
struct TinyBatchDesc { TinyDesc d; unsigned long long out_off; unsigned cap, pad; };      // instances at pool[out_off ..): + strand, then (3 * cap words on) - strand
__global__ void __launch_bounds__(TE_THREADS) k_tiny_enumerate_batch(const uint8_t *__restrict__ orig, const TinyBatchDesc *__restrict__ desc, unsigned *__restrict__ heads,
                                                                     unsigned *__restrict__ pool)
{
	__shared__ TinyDesc d;
	__shared__ unsigned long long s_off;
	__shared__ unsigned s_cap;
	if (threadIdx.x == 0) { d = desc[blockIdx.x].d; s_off = desc[blockIdx.x].out_off; s_cap = desc[blockIdx.x].cap; heads[4 * blockIdx.x] = 2u; }
	__syncthreads();
	tiny_enumerate_body(orig, d, heads + 4 * (size_t)blockIdx.x, pool + s_off, pool + s_off + 3ull * s_cap);
}

namespace {

struct BEdge {                                   // BlockFinder::Edge (src/blockfinder.h:58-90) with the fields N2 uses
	uint32_t chr, dir, startVertex, endVertex;
	uint64_t origPos, origLen;
	char firstChar;
};

inline bool natural_less(const BEdge &a, const BEdge &b)      // CompareEdgesNaturally: (startVertex, endVertex, firstChar as size_t), src/edge.cpp:26-40
{
	if (a.startVertex != b.startVertex) return a.startVertex < b.startVertex;
	if (a.endVertex != b.endVertex) return a.endVertex < b.endVertex;
	return static_cast<size_t>(a.firstChar) < static_cast<size_t>(b.firstChar);
}

struct Synteny {
	sbl_ctx *c;
	sbl_ctx *child;
	uint32_t trimK, minSize;
	std::vector<std::vector<uint8_t>> occupied;   // overlap[chr][original position] (POS_FREE / POS_OCCUPIED)
	std::vector<std::vector<uint8_t>> local;      // localOverlap of the group being resolved, cleared through `undo`
	std::vector<std::pair<uint32_t, std::pair<uint64_t, uint64_t>>> undo;

	// ResolveOverlap (src/synteny.cpp:124-166): per edge the longest run of original positions that is free both globally and
	// within the group (the first one among equals), kept when it is at least minSize long
	void resolve_overlap(const BEdge *first, const BEdge *last, std::vector<BEdge> &now)
	{
		now.clear();
		for (; first != last; ++first) {
			const uint8_t *occ = occupied[first->chr].data(), *loc = local[first->chr].data();
			const uint64_t begin = first->origPos, end = first->origPos + first->origLen;
			uint64_t bestStart = 0, bestEnd = 0;
			for (uint64_t s = begin; s < end;) {
				uint64_t e = s;
				while (e < end && occ[e] == 0 && loc[e] == 0) e++;
				if (e - s > bestEnd - bestStart) { bestStart = s; bestEnd = e; }
				s = e == s ? e + 1 : e;
			}
			if (bestEnd - bestStart >= minSize) {
				BEdge x = *first;
				x.origPos = bestStart; x.origLen = bestEnd - bestStart;
				now.push_back(x);
				memset(local[first->chr].data() + bestStart, 1, bestEnd - bestStart);
				undo.push_back({first->chr, {bestStart, bestEnd}});
			}
		}
		for (auto &u : undo) memset(local[u.first].data() + u.second.first, 0, u.second.second - u.second.first);
		undo.clear();
	}

	// small blocks: one workgroup, one launch, result in mapped host memory (k_tiny_enumerate); false = not applicable, take the general path
	TinyOut *tiny_out = nullptr, *tiny_out_dev = nullptr;
	std::vector<sbl_inst> tiny_neg;
	bool tiny_index(const std::vector<BEdge> &block, uint64_t L, uint32_t *bif_count, const sbl_inst **inst, uint64_t *ninst)
	{
		const uint32_t nrec = (uint32_t)block.size();
		if (!tiny_out || trimK > 32 || nrec > TE_MAX_REC || L + nrec + 1 > TE_MAX_ELEM) return false;
		TinyDesc d;
		d.nrec = nrec; d.k = trimK;
		for (uint32_t i = 0; i < nrec; i++) {
			d.src[i] = (unsigned long long)c->orig_sepidx[block[i].chr] + 1 + block[i].origPos;
			d.len[i] = (unsigned)block[i].origLen;
		}
		tiny_out->status = 2;
		k_tiny_enumerate<<<1, TE_THREADS, 0, c->stream>>>(c->d_orig_ch.as<uint8_t>(), d, tiny_out_dev);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(c->stream));
		if (tiny_out->status == 1) return false;                   // a non-ACGT character: the general path draws rand() for it
		SBL_CHECK(tiny_out->status == 0, SBL_ERR_INTERNAL, "small-block index did not report");
		const unsigned head[4] = { 0u, tiny_out->bif_count, tiny_out->n[0], 0u };
		tiny_result(head, tiny_out->inst[0], tiny_out->inst[1], bif_count, inst, ninst);
		return true;
	}

	// ---- speculated small-block indices of a chunk of groups (k_tiny_enumerate_batch), see sbl_generate_blocks
	DevBuf d_bdesc, d_bheads, d_bpool;
	std::vector<TinyBatchDesc> bdesc;
	std::vector<unsigned> bheads, bpool;
	const unsigned *pre_head = nullptr, *pre_inst = nullptr; unsigned pre_cap = 0;      // index speculated for the NEXT child_index call (one use)
	bool tiny_eligible(const std::vector<BEdge> &block) const
	{
		if (!tiny_out || trimK > 32 || block.empty() || block.size() > TE_MAX_REC) return false;
		uint64_t L = 0;
		for (auto &b : block) L += b.origLen;
		return L + block.size() + 1 <= TE_MAX_ELEM;
	}
	// queue the index of `block`; returns its slot in the batch
	unsigned batch_add(const std::vector<BEdge> &block)
	{
		TinyBatchDesc bd;
		memset(&bd, 0, sizeof bd);
		bd.d.nrec = (unsigned)block.size(); bd.d.k = trimK;
		unsigned E = 1;
		for (size_t i = 0; i < block.size(); i++) {
			bd.d.src[i] = (unsigned long long)c->orig_sepidx[block[i].chr] + 1 + block[i].origPos;
			bd.d.len[i] = (unsigned)block[i].origLen;
			E += (unsigned)block[i].origLen + 1;
		}
		bd.cap = E;
		bd.out_off = bdesc.empty() ? 0 : bdesc.back().out_off + 6ull * bdesc.back().cap;
		bdesc.push_back(bd);
		return (unsigned)bdesc.size() - 1;
	}
	void batch_run()
	{
		if (bdesc.empty()) return;
		const size_t nb = bdesc.size(), words = (size_t)(bdesc.back().out_off + 6ull * bdesc.back().cap);
		d_bdesc.ensure(nb * sizeof(TinyBatchDesc)); d_bheads.ensure(nb * 16); d_bpool.ensure(words * 4 + 16);
		HIP_TRY(hipMemcpyAsync(d_bdesc.p, bdesc.data(), nb * sizeof(TinyBatchDesc), hipMemcpyHostToDevice, c->stream));
		k_tiny_enumerate_batch<<<(unsigned)nb, TE_THREADS, 0, c->stream>>>(c->d_orig_ch.as<uint8_t>(), d_bdesc.as<TinyBatchDesc>(), d_bheads.as<unsigned>(), d_bpool.as<unsigned>());
		HIP_TRY(hipGetLastError());
		bheads.resize(nb * 4); bpool.resize(words);
		HIP_TRY(hipMemcpyAsync(bheads.data(), d_bheads.p, nb * 16, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipMemcpyAsync(bpool.data(), d_bpool.p, words * 4, hipMemcpyDeviceToHost, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
	}
	void batch_release() { d_bdesc.release(); d_bheads.release(); d_bpool.release(); }
	~Synteny() { batch_release(); }
	// the negative list is reported per sequence in descending element order (sbl_enumerate: each chromosome's run reversed)
	void tiny_result(const unsigned *head, const unsigned *inst0, const unsigned *inst1, uint32_t *bif_count, const sbl_inst **inst, uint64_t *ninst)
	{
		const unsigned n = head[2];
		const sbl_inst *neg = reinterpret_cast<const sbl_inst *>(inst1);
		tiny_neg.assign(neg, neg + n);
		for (size_t a = 0; a < tiny_neg.size();) {
			size_t b = a;
			while (b < tiny_neg.size() && tiny_neg[b].chr == tiny_neg[a].chr) b++;
			std::reverse(tiny_neg.begin() + a, tiny_neg.begin() + b);
			a = b;
		}
		*bif_count = head[1];
		inst[0] = reinterpret_cast<const sbl_inst *>(inst0); inst[1] = tiny_neg.data();
		ninst[0] = n; ninst[1] = n;
	}

	// a fresh index at trimK over the block sequences (IndexedSequence iseq(blockSeq, trimK, ""), synteny.cpp:44): on the GPU
	void child_index(const std::vector<BEdge> &block, uint32_t *bif_count, const sbl_inst **inst, uint64_t *ninst)
	{
		const uint32_t nrec = (uint32_t)block.size();
		uint64_t L = 0;
		for (auto &b : block) L += b.origLen;
		if (pre_head) {                                          // speculated with the rest of its chunk, and the speculation held
			const unsigned *h = pre_head, *i0 = pre_inst; const unsigned cap = pre_cap;
			pre_head = nullptr;
			if (h[0] == 0) { tiny_result(h, i0, i0 + 3ull * cap, bif_count, inst, ninst); return; }
			// (status 1: a non-ACGT character -- the general path below draws rand() for it, now, in the reference's order)
		}
		if (tiny_index(block, L, bif_count, inst, ninst)) return;
		const size_t E = (size_t)L + nrec + 1, Epad = (E + 31) / 32 * 32 + 64;
		hipStream_t s = child->stream;
		child->d_ch.ensure(Epad);
		HIP_TRY(hipMemsetAsync(child->d_ch.p, '$', Epad, s));
		std::vector<GatherDesc> desc(nrec);
		child->sepidx.assign(nrec + 1, 0);
		size_t e = 1;
		uint64_t maxlen = 1;
		for (uint32_t i = 0; i < nrec; i++) {
			child->sepidx[i] = (uint32_t)(e - 1);
			desc[i].src = (unsigned long long)c->orig_sepidx[block[i].chr] + 1 + block[i].origPos;
			desc[i].dst = e; desc[i].len = block[i].origLen;
			maxlen = std::max<uint64_t>(maxlen, block[i].origLen);
			e += block[i].origLen + 1;
		}
		child->sepidx[nrec] = (uint32_t)(e - 1);
		child->nchr = nrec; child->nelem = E;
		child->d_stage.ensure((size_t)nrec * sizeof(GatherDesc) + 64); child->d_sepidx.ensure((size_t)(nrec + 1) * 4);
		HIP_TRY(hipMemcpyAsync(child->d_stage.p, desc.data(), (size_t)nrec * sizeof(GatherDesc), hipMemcpyHostToDevice, s));
		HIP_TRY(hipMemcpyAsync(child->d_sepidx.p, child->sepidx.data(), (size_t)(nrec + 1) * 4, hipMemcpyHostToDevice, s));
		dim3 grid((unsigned)std::min<uint64_t>((maxlen + 255) / 256, 1024), std::min<unsigned>(nrec, 1024));
		k_gather_blocks<<<grid, 256, 0, s>>>(c->d_orig_ch.as<uint8_t>(), child->d_stage.as<GatherDesc>(), nrec, child->d_ch.as<uint8_t>());
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipStreamSynchronize(s));
		sbl_finish_load(child);                                  // identity positions + list of non-ACGT positions, on the device
		child->rng = c->rng;                                     // the process-global rand() of the reference: one stream for parent and child
		const sbl_inst *pos = nullptr, *neg = nullptr; uint64_t np = 0, nn = 0;
		sbl_status st = sbl_enumerate(child, trimK, bif_count, &pos, &np, &neg, &nn);
		c->rng = child->rng;
		if (st != SBL_OK) throw SblError{st, std::string("TrimBlocks index: ") + child->err};
		inst[0] = pos; inst[1] = neg; ninst[0] = np; ninst[1] = nn;
	}

	// TrimBlocks (src/synteny.cpp:31-122): for every sequence of the block the bifurcations shared with another sequence that lie
	// closest to its two ends; a sequence sharing none makes the caller try again without it
	bool trim_blocks(std::vector<BEdge> &block)
	{
		if (block.empty()) return false;                         // (an index over no sequences: nothing to trim, nothing dropped, no rand() drawn)
		uint32_t bifCount = 0; const sbl_inst *inst[2]; uint64_t ninst[2];
		child_index(block, &bifCount, inst, ninst);
		const size_t nrec = block.size();
		std::vector<uint64_t> len(nrec), base(nrec + 1, 0);
		for (size_t i = 0; i < nrec; i++) { len[i] = block[i].origLen; base[i + 1] = base[i] + len[i]; }
		// mark[strand][base[chr] + element position] = id of the k-mer starting there on that strand (GetBifurcation)
		const uint32_t NONE = 0xFFFFFFFFu;
		std::vector<uint32_t> mark[2];
		for (int s = 0; s < 2; s++) {
			mark[s].assign(base[nrec], NONE);
			for (uint64_t i = 0; i < ninst[s]; i++) {
				const sbl_inst &x = inst[s][i];
				mark[s][base[x.chr] + (s == 0 ? x.pos : len[x.chr] - 1 - x.pos)] = x.id;
			}
		}
		// ListPositions order per id (bifurcationstorage.h:59-72 after the front insertions of indexedsequence.cpp:49-67): + instances in
		// descending (chr, position), then - instances in descending (chr, reverse-complement position); CSR over ids
		std::vector<uint64_t> off((size_t)bifCount + 2, 0);
		for (int s = 0; s < 2; s++) for (uint64_t i = 0; i < ninst[s]; i++) off[inst[s][i].id + 1]++;
		for (size_t i = 1; i < off.size(); i++) off[i] += off[i - 1];
		std::vector<uint64_t> cursor(off.begin(), off.end() - 1);
		std::vector<uint32_t> ichr(off.back()); std::vector<uint64_t> ipos(off.back());
		for (int s = 0; s < 2; s++)
			for (uint64_t i = ninst[s]; i-- > 0;) {
				const sbl_inst &x = inst[s][i];
				const uint64_t at = cursor[x.id]++;
				ichr[at] = x.chr; ipos[at] = s == 0 ? x.pos : len[x.chr] - 1 - x.pos;
			}
		bool drop = false;
		std::vector<BEdge> ret;
		const uint64_t oo = 0xFFFFFFFFull;                       // UINT_MAX, synteny.cpp:41
		auto dist = [](uint64_t a, uint64_t b) { return a > b ? a - b : b - a; };      // StrandIteratorDistance, indexedsequence.cpp:162-167
		for (size_t chr = 0; chr < nrec; chr++) {
			const uint32_t dir = block[chr].dir;
			const uint64_t n = len[chr], beginPos = dir == 0 ? 0 : n - 1, lastPos = dir == 0 ? n - 1 : 0;
			uint64_t trimStart = 0, trimEnd = 0, minBifStart = oo, minBifEnd = oo, minStartSum = oo, minEndSum = oo;
			for (uint64_t t = 0; t < n; t++) {
				const uint64_t itPos = dir == 0 ? t : n - 1 - t;
				const uint32_t bifId = mark[dir][base[chr] + itPos];
				if (bifId == NONE) continue;
				const uint64_t itStart = dist(itPos, beginPos), itEnd = dist(itPos, lastPos);
				for (uint64_t a = off[bifId]; a < off[bifId + 1]; a++) {
					const uint32_t kc = ichr[a];
					if (kc == chr) continue;
					const uint64_t kn = len[kc], kStart = block[kc].dir == 0 ? 0 : kn - 1, kLast = block[kc].dir == 0 ? kn - 1 : 0;
					const uint64_t startSum = dist(ipos[a], kStart) + itStart, endSum = dist(ipos[a], kLast) + itEnd;
					if (startSum < minStartSum || (startSum == minStartSum && bifId < minBifStart)) { minBifStart = bifId; minStartSum = startSum; trimStart = itPos; }
					if (endSum < minEndSum || (endSum == minEndSum && bifId < minBifEnd)) { minBifEnd = bifId; minEndSum = endSum; trimEnd = itPos; }
				}
			}
			if (minStartSum < oo && minEndSum < oo) {
				if (dist(trimStart, trimEnd) + trimK >= minSize) {
					const uint64_t endElem = dir == 0 ? trimEnd + (trimK - 1) : trimEnd - (trimK - 1);      // std::advance(trimEnd, trimK - 1)
					BEdge x = block[chr];
					x.origPos = block[chr].origPos + std::min(trimStart, endElem);
					x.origLen = std::max(trimStart, endElem) + 1 - std::min(trimStart, endElem);
					ret.push_back(x);
				}
			} else drop = true;
		}
		block.swap(ret);
		return drop;
	}
};

}  // namespace

extern "C" sbl_status sbl_generate_blocks(sbl_ctx *c, uint32_t k, uint32_t trim_k, uint32_t min_size, int shared_only, const sbl_block **blocks, uint64_t *n)
{
	return guarded(c, [&] {
		SBL_CHECK(k >= 2 && trim_k >= 2, SBL_ERR_BAD_ARG, "vertex sizes must be at least 2");
		SBL_CHECK(c->d_orig_ch.p && c->orig_sepidx.size() == (size_t)c->nchr + 1, SBL_ERR_BAD_ARG, "no original records (load first)");
		// ---- ListEdges of a fresh index at k (serialization.cpp:56-86) on the GPU
		const sbl_edge *ge = nullptr; uint64_t ne = 0;
		sbl_status st = sbl_list_edges(c, k, &ge, &ne);
		if (st != SBL_OK) throw SblError{st, c->err};
		std::vector<BEdge> edge;
		edge.reserve(ne);
		for (uint64_t i = 0; i < ne; i++) {
			if (ge[i].orig_len < min_size) continue;                  // EdgeEmpty, edge.cpp:32-35
			BEdge e; e.chr = ge[i].chr; e.dir = ge[i].strand; e.startVertex = ge[i].start_vertex; e.endVertex = ge[i].end_vertex;
			e.origPos = ge[i].orig_pos; e.origLen = ge[i].orig_len; e.firstChar = ge[i].first_char;
			edge.push_back(e);
		}
		Synteny sy;
		sy.c = c; sy.trimK = trim_k; sy.minSize = min_size;
		if (!c->child) {
			sbl_status cs = sbl_create(&c->child, c->device);
			if (cs != SBL_OK) throw SblError{cs, "cannot create the trim context"};
		}
		sy.child = c->child;
		if (!c->tiny_out && !getenv("SBL_NO_TINY_INDEX")) {          // mapped host buffer of the small-block index, owned by the context
			void *h = nullptr;
			HIP_TRY(hipHostMalloc(&h, sizeof(TinyOut), hipHostMallocMapped));
			c->tiny_out = h;
		}
		if (c->tiny_out && !getenv("SBL_NO_TINY_INDEX")) {
			void *dv = nullptr;
			HIP_TRY(hipHostGetDevicePointer(&dv, c->tiny_out, 0));
			sy.tiny_out = static_cast<TinyOut *>(c->tiny_out); sy.tiny_out_dev = static_cast<TinyOut *>(dv);
		}
		sy.occupied.resize(c->nchr); sy.local.resize(c->nchr);
		for (uint32_t i = 0; i < c->nchr; i++) {
			const size_t len = c->orig_sepidx[i + 1] - c->orig_sepidx[i] - 1;          // originalSize_
			sy.occupied[i].assign(len, 0); sy.local[i].assign(len, 0);
		}
		// ---- groups of edges that spell the same path (GroupBy over CompareEdgesNaturally, common.h:150-160), largest first
		std::sort(edge.begin(), edge.end(), natural_less);
		std::vector<std::pair<size_t, size_t>> group;
		for (size_t now = 0; now < edge.size();) {
			size_t prev = now;
			while (now < edge.size() && !natural_less(edge[prev], edge[now])) now++;
			group.push_back({prev, now});
		}
		std::sort(group.begin(), group.end(), [](const std::pair<size_t, size_t> &a, const std::pair<size_t, size_t> &b) { return a.second - a.first > b.second - b.first; });
		c->blocks.clear();
		int blockCount = 1;
		std::vector<BEdge> now;
		std::vector<uint32_t> occur(c->nchr);
		// Groups are resolved strictly in order (a block occupies its positions for every later group), but a group's candidate block
		// only depends on the earlier ones where they overlap it -- rarely.  So the candidate blocks of a whole CHUNK of groups are
		// computed against the occupancy at the start of the chunk and their small-block indices built in ONE launch; a group whose
		// candidate block comes out the same when its turn comes (it is recomputed: a byte scan) uses that index, any other group
		// and every further trim iteration takes the one-block path as before.
		const bool speculate = sy.tiny_out != nullptr && getenv("SBL_NO_TINY_BATCH") == nullptr;
		const size_t CHUNK = 8192;
		struct Spec { std::vector<BEdge> now; int slot; };
		std::vector<Spec> spec;
		for (size_t gi = 0; gi < group.size(); gi++) {
			const auto &g = group[gi];
			BEdge *first = edge.data() + g.first, *last = edge.data() + g.second;
			if (gi % CHUNK == 0) {                                      // speculation pass over the next chunk
				const size_t gend = std::min(group.size(), gi + CHUNK);
				spec.assign(gend - gi, Spec{{}, -1});
				sy.bdesc.clear();
				for (size_t gj = gi; gj < gend; gj++) {
					BEdge *f = edge.data() + group[gj].first, *l = edge.data() + group[gj].second;
					std::sort(f, l, [](const BEdge &a, const BEdge &b) { return a.dir < b.dir; });      // CompareEdgesByDirection (once per group, here)
					if (!speculate || l - f < 2 || f->dir != 0) continue;
					Spec &sp = spec[gj - gi];
					sy.resolve_overlap(f, l, sp.now);
					if (sy.tiny_eligible(sp.now)) sp.slot = (int)sy.batch_add(sp.now);
				}
				sy.batch_run();
			}
			if (last - first < 2 || first->dir != 0) continue;          // fewer than two edges, or none on the positive strand (sorted: it would be first)
			sy.resolve_overlap(first, last, now);
			{
				const Spec &sp = spec[gi % CHUNK];
				bool same = sp.slot >= 0 && sp.now.size() == now.size();
				for (size_t i = 0; same && i < now.size(); i++)
					same = sp.now[i].chr == now[i].chr && sp.now[i].origPos == now[i].origPos && sp.now[i].origLen == now[i].origLen && sp.now[i].dir == now[i].dir;
				if (same) {
					const TinyBatchDesc &bd = sy.bdesc[sp.slot];
					sy.pre_head = sy.bheads.data() + 4 * (size_t)sp.slot; sy.pre_inst = sy.bpool.data() + bd.out_off; sy.pre_cap = bd.cap;
				}
			}
			while (sy.trim_blocks(now)) { }
			sy.pre_head = nullptr;                                      // (an empty candidate block makes no index)
			std::fill(occur.begin(), occur.end(), 0u);
			for (const BEdge &e : now) occur[e.chr]++;
			if (now.size() > 1 && (!shared_only || (size_t)std::count(occur.begin(), occur.end(), 1u) == c->nchr)) {
				for (const BEdge &e : now) {
					memset(sy.occupied[e.chr].data() + e.origPos, 1, e.origLen);
					sbl_block b; b.id = e.dir == 0 ? blockCount : -blockCount; b.chr = e.chr; b.start = e.origPos; b.end = e.origPos + e.origLen;
					c->blocks.push_back(b);
				}
				blockCount++;
			}
		}
		sy.batch_release();
		std::sort(c->blocks.begin(), c->blocks.end(), [](const sbl_block &a, const sbl_block &b) { return std::make_pair(a.chr, a.start) < std::make_pair(b.chr, b.start); });
		if (blocks) *blocks = c->blocks.data();
		if (n) *n = c->blocks.size();
	});
}
