// sbl_api.hip -- C ABI (include/sibelia_amd.h) + enumeration pipeline of the MI355X-native BlockFinder hot path.
//
// Host code here only orchestrates: every pass over sequence data, every table access and every
// sort runs in a HIP kernel on the context's device.  There is no host compute path.
#include <cstring>
#include <algorithm>
#include <rocprim/rocprim.hpp>

#include "sbl_ctx.h"
#include "sbl_comm.h"
#include "kmer_kernels.h"
#include "kmer_bucket_kernels.h"

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------------------------------- small kernels
__global__ void __launch_bounds__(256) k_scatter_chars(uint8_t *ch, const unsigned *elem, const uint8_t *val, unsigned n)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) ch[elem[i]] = val[i];
}

// ListEdges (reference src/serialization.cpp:56-86) over the dense mark arrays of a fresh index.
// Edge i of a strand connects the consecutive marks (i, i+1) of the compacted list when both lie
// on the same chromosome; `valid[i]` says whether it exists.
__global__ void __launch_bounds__(256) k_list_edges(const unsigned *__restrict__ melem, const unsigned *__restrict__ mid, unsigned n,
                                                    unsigned strand, unsigned k, const uint8_t *__restrict__ ch,
                                                    const unsigned *__restrict__ op, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                    sbl_edge *__restrict__ out, uint8_t *__restrict__ valid)
{
	unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i + 1 >= n) return;
	unsigned lo = melem[i], hi = melem[i + 1];
	unsigned c = chr_of(sepidx, nchr, lo);
	if (hi > sepidx[c + 1]) { valid[i] = 0; return; }
	unsigned length = sepidx[c + 1] - sepidx[c] - 1, step = hi - lo;
	sbl_edge e;
	unsigned p1, p2;
	if (strand == 0) {
		unsigned pos = lo - sepidx[c] - 1;
		e.start_vertex = mid[i]; e.end_vertex = mid[i + 1];
		e.pos = pos;
		e.first_char = (char)ch[lo + k];
		p1 = op[lo] & 0x1FFFFFFFu; p2 = op[hi + k - 1] & 0x1FFFFFFFu;
	} else {
		unsigned pos = sepidx[c + 1] - 1 - hi;            // walk position of the origin (the higher element)
		e.start_vertex = mid[i + 1]; e.end_vertex = mid[i];
		e.pos = length - (pos + step + k);
		unsigned char x = ch[hi - k];
		e.first_char = (char)(x == 'A' ? 'T' : x == 'C' ? 'G' : x == 'G' ? 'C' : x == 'T' ? 'A' : x);
		p1 = op[hi] & 0x1FFFFFFFu; p2 = op[lo - k + 1] & 0x1FFFFFFFu;
	}
	e.chr = c; e.strand = strand; e.len = step + k;
	e.orig_pos = p1 < p2 ? p1 : p2;
	e.orig_len = (p1 < p2 ? p2 : p1) + 1 - e.orig_pos;
	e.pad_[0] = e.pad_[1] = e.pad_[2] = 0;
	out[i] = e;
	valid[i] = 1;
}

// H0: the k-mer hash of the reference's hashing.h (SlidingWindow::CalcKMerHash, src/hashing.h:73-88):
// H(w) = sum w[i] * 57^(k-1-i) mod 2^64 over the characters a StrandIterator yields (negative strand: complemented,
// DNASequence::Translate).  One thread per k-mer; out index = [strand 0: chromosomes ascending, walk order][strand 1: the same].
__global__ void __launch_bounds__(256) k_kmer_hashes(const uint8_t *__restrict__ ch, const unsigned *__restrict__ sepidx, unsigned nchr, unsigned k,
                                                     const unsigned long long *__restrict__ chroff /* nchr + 1: k-mers before chromosome c on one strand */,
                                                     unsigned long long per_strand, unsigned long long *__restrict__ out)
{
	unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= 2 * per_strand) return;
	const unsigned strand = i >= per_strand;
	const unsigned long long j = strand ? i - per_strand : i;
	unsigned lo = 0, hi = nchr;                               // chromosome c with chroff[c] <= j < chroff[c + 1]
	while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (chroff[mid] <= j) lo = mid; else hi = mid; }
	const unsigned c = lo, p = (unsigned)(j - chroff[c]);
	const unsigned first = sepidx[c] + 1, last = sepidx[c + 1] - 1;
	unsigned long long h = 0;
	for (unsigned t = 0; t < k; t++) {
		uint8_t x = strand ? ch[last - p - t] : ch[first + p + t];
		if (strand) x = x == 'A' ? 'T' : x == 'T' ? 'A' : x == 'C' ? 'G' : x == 'G' ? 'C' : x == 'a' ? 't' : x == 't' ? 'a' : x == 'c' ? 'g' : x == 'g' ? 'c' : x;
		h = h * 57ull + (unsigned long long)(long long)(signed char)x;
	}
	out[i] = h;
}

// BlockFinder::SerializeGraph (reference src/serialization.cpp:112-138, OutputEdge :15-24): DOT text of the UNcondensed graph, one
// line per (k+1)-window, generated on the device.  Line i of [strand-major within a chromosome] has a fixed part of 2k + 35 (+1 for
// "blue") characters and the decimal digits of (chr, pos): lengths -> exclusive scan -> every thread writes its own line.
__device__ __forceinline__ unsigned dec_digits(unsigned v) { unsigned d = 1; while (v >= 10) { v /= 10; d++; } return d; }
struct GraphLine { unsigned chr, strand, pos; };
__device__ __forceinline__ GraphLine graph_line(unsigned long long i, const unsigned long long *__restrict__ lineoff, unsigned nchr)
{
	// lineoff[c] = lines before chromosome c (two strands each); order: chr, strand, pos
	unsigned lo = 0, hi = nchr;
	while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (lineoff[mid] <= i) lo = mid; else hi = mid; }
	const unsigned long long per = (lineoff[lo + 1] - lineoff[lo]) / 2, j = i - lineoff[lo];
	GraphLine g; g.chr = lo; g.strand = j >= per; g.pos = (unsigned)(g.strand ? j - per : j);
	return g;
}
__global__ void __launch_bounds__(256) k_graph_line_len(const unsigned long long *__restrict__ lineoff, unsigned nchr, unsigned k, unsigned long long nlines, unsigned *__restrict__ len)
{
	unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nlines) return;
	const GraphLine g = graph_line(i, lineoff, nchr);
	// "<k> -> <k> [color=\"red\", label=\"(c, p)\"];\n"
	len[i] = 2 * k + 4 + 9 + (g.strand ? 3 : 4) + 11 + dec_digits(g.chr) + 2 + dec_digits(g.pos) + 5;
}
__global__ void __launch_bounds__(256) k_graph_lines(const uint8_t *__restrict__ ch, const unsigned *__restrict__ sepidx, const unsigned long long *__restrict__ lineoff, unsigned nchr,
                                                     unsigned k, unsigned long long nlines, const unsigned long long *__restrict__ textoff, char *__restrict__ text)
{
	unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nlines) return;
	const GraphLine g = graph_line(i, lineoff, nchr);
	char *o = text + textoff[i];
	const unsigned first = sepidx[g.chr] + 1, last = sepidx[g.chr + 1] - 1;
	auto at = [&](unsigned p) -> char {
		uint8_t x = g.strand ? ch[last - p] : ch[first + p];
		if (g.strand) x = x == 'A' ? 'T' : x == 'T' ? 'A' : x == 'C' ? 'G' : x == 'G' ? 'C' : x == 'a' ? 't' : x == 't' ? 'a' : x == 'c' ? 'g' : x == 'g' ? 'c' : x;
		return (char)x;
	};
	auto put = [&](const char *s) { while (*s) *o++ = *s++; };
	auto num = [&](unsigned v) { unsigned d = dec_digits(v); for (unsigned x = d; x-- > 0;) { o[x] = (char)('0' + v % 10); v /= 10; } o += d; };
	for (unsigned t = 0; t < k; t++) *o++ = at(g.pos + t);
	put(" -> ");
	for (unsigned t = 0; t < k; t++) *o++ = at(g.pos + 1 + t);
	put(" [color=\""); put(g.strand ? "red" : "blue"); put("\", label=\"("); num(g.chr); put(", "); num(g.pos); put(")\"];\n");
}

// ------------------------------------------------------------------------------------------- helpers
static void drop_host_state(sbl_ctx *c) { c->host_state_valid = false; }

// IndexedSequence::Init sanitise (reference src/indexedsequence.cpp:31-37): every non-ACGT character,
// chromosome-major, becomes "ACGT"[rand() % 4].  Returns true if something was replaced.
static bool sanitise_apply(sbl_ctx *c)
{
	size_t n = c->amb_elem.size();
	if (!n) return false;
	std::vector<uint8_t> rep(n);
	for (size_t i = 0; i < n; i++) rep[i] = (uint8_t)"ACGT"[c->rng.next() % 4];
	c->d_amb_elem.ensure(n * 4); c->d_amb_char.ensure(n);
	HIP_TRY(hipMemcpyAsync(c->d_amb_elem.p, c->amb_elem.data(), n * 4, hipMemcpyHostToDevice, c->stream));
	HIP_TRY(hipMemcpyAsync(c->d_amb_char.p, rep.data(), n, hipMemcpyHostToDevice, c->stream));
	k_scatter_chars<<<nblocks(n, 256), 256, 0, c->stream>>>(c->d_ch.as<uint8_t>(), c->d_amb_elem.as<unsigned>(), c->d_amb_char.as<uint8_t>(), (unsigned)n);
	HIP_TRY(hipStreamSynchronize(c->stream));
	return true;
}
static void sanitise_revert(sbl_ctx *c)
{
	size_t n = c->amb_elem.size();
	if (!n) return;
	HIP_TRY(hipMemcpyAsync(c->d_amb_char.p, c->amb_orig.data(), n, hipMemcpyHostToDevice, c->stream));
	k_scatter_chars<<<nblocks(n, 256), 256, 0, c->stream>>>(c->d_ch.as<uint8_t>(), c->d_amb_elem.as<unsigned>(), c->d_amb_char.as<uint8_t>(), (unsigned)n);
	HIP_TRY(hipStreamSynchronize(c->stream));
}
void sbl_sanitise_commit(sbl_ctx *c) { c->amb_elem.clear(); c->amb_orig.clear(); }
// Without -r the reference builds its suffix array through two temporary files per IndexedSequence (CalculateLCP's lcpFile, then
// CreateFileWithSA's posFile: src/vertexenumeration.cpp:125,101), and TempFile draws each name from the SAME process-global rand()
// that sanitises ambiguous bases: "Sib_" + 12 x ('a' + rand() % 26) (src/platform.cpp:52-58) -- 24 draws right after the sanitising
// draws of every index built with a temp directory (the stage, SerializeCondensedGraph, GenerateSyntenyBlocks' main index; TrimBlocks'
// indices are built in RAM, src/synteny.cpp:44).  Nothing is spilled here; in temp-file mode the context only keeps its stream in
// step, so that inputs with several ambiguous bases come out as the reference's temp-file mode produces them.
static void tempfile_draws(sbl_ctx *c) { if (c->tempfile_mode) for (int i = 0; i < 24; i++) (void)c->rng.next(); }
// sbl_enumerate / sbl_list_edges look at a sanitised COPY (a fresh IndexedSequence, reference src/indexedsequence.cpp:28-37):
// the replacement is applied to the resident state for the duration of the call and taken back on every way out --
// also when the enumeration throws (OOM, TOO_LARGE, a failed collective), together with the rand() stream.
struct SanitiseScope {
	sbl_ctx *c; bool applied; GlibcRand rng0; bool ok = false;
	explicit SanitiseScope(sbl_ctx *cx) : c(cx), rng0(cx->rng) { applied = sanitise_apply(cx); tempfile_draws(cx); }
	void done() { ok = true; }
	~SanitiseScope()
	{
		if (applied) { try { sanitise_revert(c); } catch (...) { } }
		if (!ok) c->rng = rng0;
	}
};

// K1
void sbl_pack(sbl_ctx *c)
{
	size_t nwords = (c->nelem + 31) / 32;
	c->d_pk.ensure(nwords * 8); c->d_sp.ensure(nwords * 4);
	k_pack2bit<<<nblocks(nwords, 256), 256, 0, c->stream>>>(c->d_ch.as<uint8_t>(), c->d_pk.as<unsigned long long>(), c->d_sp.as<unsigned>(), nwords);
	HIP_TRY(hipGetLastError());
}

static void device_sort_pairs(sbl_ctx *c, unsigned long long *kin, unsigned long long *kout, unsigned *vin, unsigned *vout, size_t n, unsigned bits)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
	c->d_sorttmp.ensure(tmp);
	HIP_TRY(rocprim::radix_sort_pairs(c->d_sorttmp.p, tmp, kin, kout, vin, vout, n, 0, bits, c->stream));
}
static void device_exclusive_scan(sbl_ctx *c, unsigned *in, unsigned *out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
	c->d_scantmp.ensure(tmp);
	HIP_TRY(rocprim::exclusive_scan(c->d_scantmp.p, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
}

static void device_sort_records(sbl_ctx *c, unsigned long long *kin, unsigned long long *kout, unsigned long long *vin, unsigned long long *vout,
                                size_t n, unsigned begin_bit, unsigned end_bit)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
	c->d_sorttmp.ensure(tmp);
	HIP_TRY(rocprim::radix_sort_pairs(c->d_sorttmp.p, tmp, kin, kout, vin, vout, n, begin_bit, end_bit, c->stream));
}

// E1 (+ the dense form of E2): pack -> k-mer records -> radix partition by hash prefix -> per-bucket LDS tables (classify)
// -> rank of the bifurcation codes -> marks of the member positions (kmer_bucket_kernels.h).
// elem_capacity >= nelem is the allocated length of the mark arrays (simplification appends elements).
void sbl_run_enumeration(sbl_ctx *c, uint32_t k, size_t elem_capacity)
{
	SBL_CHECK(k >= 2, SBL_ERR_BAD_ARG, "vertex size k must be at least 2");
	c->dict_keys = nullptr;
	c->marks_compact_ready = false;
	if (k > 32) {
		// long k: window fingerprints through the bucketed table, every bifurcation group verified on the sequence (longk_fp.hip, round 6);
		// exact rank doubling (longk.hip) if a verification ever fails, on request (SBL_LONGK_DOUBLING=1: the A/B), and -- split over the
		// attached GPUs -- for a job on several GPUs
		const bool doubling = getenv("SBL_LONGK_DOUBLING") != nullptr && atoi(getenv("SBL_LONGK_DOUBLING")) != 0;
		c->stats.longk_path = 0;
		const bool sharded = c->comm && getenv("SBL_LONGK_REPLICATED") == nullptr;      // one job on several GPUs: the table (or the doubling) is split over them
		bool done = false;
		if (!doubling) {
			// (with a communicator the fingerprint table is sharded by hash prefix, all-to-all like the k <= 32 table; its verdict "a
			// verification failed" is agreed on by all ranks, so they all take the fall-back together.  A rank that leaves with an error
			// releases its peers.)
			if (sharded) { try { done = sbl_run_enumeration_longk_fp(c, k, elem_capacity); } catch (...) { c->comm->abort_peers(); throw; } }
			else { struct SblComm *cm = c->comm; c->comm = nullptr; try { done = sbl_run_enumeration_longk_fp(c, k, elem_capacity); } catch (...) { c->comm = cm; throw; } c->comm = cm; }      // (SBL_LONGK_REPLICATED: every rank builds the whole table)
		}
		if (done) c->stats.longk_path = 1;
		else {
			if (sharded) sbl_run_enumeration_longk_sharded(c, k, elem_capacity); else sbl_run_enumeration_longk(c, k, elem_capacity);
			c->stats.longk_path = doubling ? 2 : 3;                                      // (3: fell back after a failed verification)
		}
		return;
	}
	if (c->comm) { sbl_run_enumeration_sharded(c, k, elem_capacity); return; }    // k-mer table sharded by hash prefix over the attached GPUs
	SBL_CHECK(kmer_hash(~0ull) == KB_EMPTY_KEY && kmer_unhash(kmer_hash(0x123456789ABCDEFull)) == 0x123456789ABCDEFull, SBL_ERR_INTERNAL, "k-mer hash constants");
	hipStream_t s = c->stream;
	const size_t E = c->nelem, nwords = (E + 31) / 32;
	const size_t ntiles = (nwords + KM_TILE_WORDS - 1) / KM_TILE_WORDS;
	const size_t n = ntiles * (size_t)(KM_TILE_WORDS * 32);                          // records (one per element slot of the tiles)
	SBL_CHECK(n < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many positions for 32-bit record indices");
	c->cur_k = k;
	sbl_pack(c);
	for (int i = 0; i < 2; i++) { c->d_rec_keys[i].ensure(n * 8); c->d_rec_vals[i].ensure(n * 8); }
	c->d_counters.ensure(256 * 4);
	unsigned long long *k0 = c->d_rec_keys[0].as<unsigned long long>(), *k1 = c->d_rec_keys[1].as<unsigned long long>();
	unsigned long long *v0 = c->d_rec_vals[0].as<unsigned long long>(), *v1 = c->d_rec_vals[1].as<unsigned long long>();

	HIP_TRY(hipEventRecord(c->ev[0], s));
	const unsigned grid = (unsigned)std::min<size_t>(ntiles, 256 * 16);
	k_kmer_records<<<grid, KM_THREADS, 0, s>>>(c->d_pk.as<unsigned long long>(), c->d_sp.as<unsigned>(), nwords, E, k, (size_t)0, ntiles, k0, v0);
	HIP_TRY(hipGetLastError());

	// buckets of ~350-700 records (an LDS table holds KB_MAX_DISTINCT distinct k-mers); more bits if a bucket overflows
	unsigned bits = 4;
	while (bits < 30 && (n >> bits) > KB_SLOTS * 9 / 16) bits++;
	size_t maxpairs = n / 8 + 4096, maxmembers = n;              // capacities of the classification outputs; grown on demand
	// test hooks: start with too few bucket bits / too small an output buffer so that the re-bucket and grow paths run
	if (const char *e = getenv("SBL_TEST_BUCKET_BITS")) bits = std::min(bits, (unsigned)std::max(1, atoi(e)));
	if (const char *e = getenv("SBL_TEST_MAXPAIRS")) maxpairs = (size_t)std::max(1, atoi(e));
	unsigned cnt[4] = {0, 0, 0, 0};
	unsigned long long *members = k0;                            // the unsorted records are dead after the partition: their space holds the member list
	for (int attempt = 0;; attempt++) {
		SBL_CHECK(attempt < 8, SBL_ERR_INTERNAL, "k-mer bucket classification did not converge");
		if (attempt == 0 || (cnt[3] & 1u)) {
			if (attempt) {                                       // re-bucket with a longer prefix: the records have to be generated again (k0 was reused)
				SBL_CHECK(bits < 28, SBL_ERR_TOO_LARGE, "k-mer buckets keep overflowing at 2^28 buckets (adversarial key distribution)");
				bits = std::min(bits + 2, 28u);                  // grid = 2^bits workgroups, offsets (2^bits + 1) x 4 B: bounded
				k_kmer_records<<<grid, KM_THREADS, 0, s>>>(c->d_pk.as<unsigned long long>(), c->d_sp.as<unsigned>(), nwords, E, k, (size_t)0, ntiles, k0, v0);
			}
			device_sort_records(c, k0, k1, v0, v1, n, 0, bits);
			c->d_boff.ensure((((size_t)1 << bits) + 1) * 4 + 64);
			k_bucket_bounds<<<nblocks(((size_t)1 << bits) + 1, 256), 256, 0, s>>>(k1, n, bits, c->d_boff.as<unsigned>());
		}
		c->d_keys.ensure(maxpairs * 16 + 16); c->d_payload.ensure(maxpairs * 8 + 16);
		HIP_TRY(hipMemsetAsync(c->d_counters.p, 0, KB_CTR_WORDS * 4, s));
		k_bucket_classify<<<nblocks((size_t)1 << bits, KB_GROUP), KB_THREADS, 0, s>>>(k1, v1, c->d_boff.as<unsigned>(), (unsigned)((size_t)1 << bits), k, c->d_counters.as<unsigned>(),
		                                                                      c->d_keys.as<unsigned long long>(), c->d_payload.as<unsigned>(), (unsigned)maxpairs,
		                                                                      members, (unsigned)maxmembers);
		HIP_TRY(hipGetLastError());
		{
			unsigned all[KB_CTR_WORDS];
			HIP_TRY(hipMemcpyAsync(all, c->d_counters.p, sizeof all, hipMemcpyDeviceToHost, s));
			HIP_TRY(hipStreamSynchronize(s));
			cnt[0] = all[KB_CTR_PAIRS]; cnt[1] = all[KB_CTR_KEYS]; cnt[2] = all[KB_CTR_MEM]; cnt[3] = all[KB_CTR_FLAG];
		}
		if (cnt[3] & 1u) continue;
		if (cnt[0] > maxpairs || (size_t)cnt[1] > 2 * maxpairs) { maxpairs = std::max<size_t>(cnt[0], ((size_t)cnt[1] + 1) / 2) + 1024; continue; }
		break;
	}
	HIP_TRY(hipEventRecord(c->ev[1], s));
	const unsigned npairs = cnt[0], nkeys = cnt[1], nmem = cnt[2];
	c->d_skeys.ensure((size_t)nkeys * 8 + 16); c->d_spayload.ensure((size_t)nkeys * 4 + 16);
	c->d_pairids.ensure((size_t)npairs * 8 + 16);
	if (nkeys) {
		device_sort_pairs(c, c->d_keys.as<unsigned long long>(), c->d_skeys.as<unsigned long long>(),
		                  c->d_payload.as<unsigned>(), c->d_spayload.as<unsigned>(), nkeys, 2 * k);
		k_scatter_ids<<<nblocks(nkeys, 256), 256, 0, s>>>(c->d_skeys.as<unsigned long long>(), c->d_spayload.as<unsigned>(), nkeys, k,
		                                                 c->d_pairids.as<unsigned>());
	}
	c->bif_count = nkeys;

	for (int st = 0; st < 2; st++) {
		c->d_bif[st].ensure(elem_capacity * 4);
		HIP_TRY(hipMemsetAsync(c->d_bif[st].p, 0xFF, elem_capacity * 4, s));
	}
	if (nmem)
		k_scatter_members<<<nblocks(nmem, 256), 256, 0, s>>>(members, nmem, k, c->d_pairids.as<unsigned>(), c->d_bif[0].as<unsigned>(), c->d_bif[1].as<unsigned>());
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(s));
	float ms = 0;
	HIP_TRY(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
	c->stats.kmer_table_ms = ms;
	// algorithmic bytes of the table build: 2-bit sequence read once + one 16-B slot read and written per base position
	size_t positions = 0;
	for (uint32_t ch = 0; ch < c->nchr; ch++) {
		size_t len = c->sepidx[ch + 1] - c->sepidx[ch] - 1;
		if (len >= k) positions += len - k + 1;
	}
	c->stats.kmer_table_bytes = positions * 32 + E / 4;
	c->stats.strand_kmers = 2 * positions;
	c->stats.bif_count = nkeys;
	c->dict_keys = c->d_skeys.as<unsigned long long>();
}

// ordered compaction of one strand's marks into (element, id) arrays
void sbl_compact_marks(sbl_ctx *c, int strand)
{
	if (c->marks_compact_ready) return;                          // (the enumeration sorted its few member positions instead of scanning every element)
	hipStream_t s = c->stream;
	size_t E = c->nelem;
	unsigned nchunks = nblocks(E, 1024);
	c->d_chunkcnt.ensure((size_t)(nchunks + 1) * 4); c->d_chunkoff.ensure((size_t)(nchunks + 1) * 4);
	HIP_TRY(hipMemsetAsync(c->d_chunkcnt.p, 0, (size_t)(nchunks + 1) * 4, s));
	k_count_marks<<<nchunks, 256, 0, s>>>(c->d_bif[strand].as<unsigned>(), E, c->d_chunkcnt.as<unsigned>());
	device_exclusive_scan(c, c->d_chunkcnt.as<unsigned>(), c->d_chunkoff.as<unsigned>(), nchunks + 1);
	unsigned total = 0;
	HIP_TRY(hipMemcpyAsync(&total, c->d_chunkoff.as<unsigned>() + nchunks, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	c->nmarks[strand] = total;
	c->d_melem[strand].ensure((size_t)total * 4 + 16); c->d_mid[strand].ensure((size_t)total * 4 + 16);
	if (total)
		k_write_marks<<<nchunks, 256, 0, s>>>(c->d_bif[strand].as<unsigned>(), E, c->d_chunkoff.as<unsigned>(),
		                                      c->d_melem[strand].as<unsigned>(), c->d_mid[strand].as<unsigned>());
	HIP_TRY(hipGetLastError());
}

// ------------------------------------------------------------------------------------------- C ABI
extern "C" sbl_status sbl_create(sbl_ctx **out, int device)
{
	if (!out) return SBL_ERR_BAD_ARG;
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SBL_ERR_NO_DEVICE;
	if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
	if (device >= ndev) return SBL_ERR_BAD_ARG;
	if (hipSetDevice(device) != hipSuccess) return SBL_ERR_NO_DEVICE;
	sbl_ctx *c = new (std::nothrow) sbl_ctx();
	if (!c) return SBL_ERR_OOM;
	c->device = device;
	if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return SBL_ERR_HIP; }
	for (auto &e : c->ev) if (hipEventCreate(&e) != hipSuccess) { delete c; return SBL_ERR_HIP; }
	*out = c;
	return SBL_OK;
}

extern "C" void sbl_destroy(sbl_ctx *c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	if (c->child) { sbl_destroy(c->child); c->child = nullptr; }
	if (c->tiny_out) { (void)hipHostFree(c->tiny_out); c->tiny_out = nullptr; }
	c->release_host_state();
	sbl_simplify_free(c);
	sbl_comm_release(c);
	sbl_longk_free(c);
	sbl_longk_fp_free(c);
	DevBuf *bufs[] = { &c->d_send, &c->d_recv, &c->d_otable, &c->d_oused, &c->d_allkeys, &c->d_allkeys2, &c->d_gelem[0], &c->d_gelem[1], &c->d_gid[0], &c->d_gid[1], &c->d_stage, &c->d_ch, &c->d_op, &c->d_sepidx, &c->d_amb_elem, &c->d_amb_char, &c->d_pk, &c->d_sp, &c->d_counters,
	                   &c->d_keys, &c->d_payload, &c->d_skeys, &c->d_spayload, &c->d_pairids, &c->d_sorttmp, &c->d_bif[0], &c->d_bif[1],
	                   &c->d_chunkcnt, &c->d_chunkoff, &c->d_scantmp, &c->d_save_ch, &c->d_save_op, &c->d_melem[0], &c->d_melem[1], &c->d_mid[0], &c->d_mid[1], &c->d_inst, &c->d_edges, &c->d_valid, &c->d_rec_keys[0], &c->d_rec_keys[1], &c->d_rec_vals[0], &c->d_rec_vals[1], &c->d_boff, &c->d_fa_text, &c->d_fa_lines, &c->d_fa_recs, &c->d_orig_ch };
	for (DevBuf *b : bufs) b->release();
	for (auto &e : c->ev) if (e) (void)hipEventDestroy(e);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

extern "C" sbl_status sbl_load(sbl_ctx *c, uint32_t nchr, const uint8_t *const *seq, const uint64_t *len)
{
	return guarded(c, [&] {
		SBL_CHECK(nchr == 0 || (seq && len), SBL_ERR_BAD_ARG, "null input");
		uint64_t L = 0;
		for (uint32_t i = 0; i < nchr; i++) {
			SBL_CHECK(len[i] < (1ull << 29), SBL_ERR_TOO_LARGE, "a chromosome must be shorter than 2^29 bp (29-bit original positions)");
			L += len[i];
		}
		SBL_CHECK(L <= (1ull << 30), SBL_ERR_TOO_LARGE, "total input larger than 2^30 bp");
		size_t E = (size_t)L + nchr + 1, Epad = (E + 31) / 32 * 32 + 64;
		// the element array '$' c0 '$' c1 '$' ...: one memcpy per record on the host (1 B/base); original positions and the
		// list of non-ACGT positions are derived on the device (sbl_finish_load, fasta_load.hip)
		std::vector<uint8_t> ch(Epad, (uint8_t)'$');
		c->sepidx.assign(nchr + 1, 0);
		size_t e = 1;
		for (uint32_t i = 0; i < nchr; i++) {
			c->sepidx[i] = (uint32_t)(e - 1);
			if (len[i]) memcpy(&ch[e], seq[i], len[i]);
			e += len[i] + 1;
		}
		c->sepidx[nchr] = (uint32_t)(e - 1);
		c->nchr = nchr; c->nelem = E;
		c->hint_elem_slack = 0; c->hint_cap_n = 0; c->hint_checkpoints = false;      // capacities an earlier, unrelated input asked for do not carry over
		c->fa_names.clear();
		c->d_ch.ensure(Epad); c->d_sepidx.ensure((size_t)(nchr + 1) * 4);
		HIP_TRY(hipMemcpyAsync(c->d_ch.p, ch.data(), Epad, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->d_sepidx.p, c->sepidx.data(), (size_t)(nchr + 1) * 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		sbl_finish_load(c);
	});
}

extern "C" uint32_t sbl_nchr(const sbl_ctx *c) { return c ? c->nchr : 0; }

extern "C" sbl_status sbl_enumerate(sbl_ctx *c, uint32_t k, uint32_t *bif_count,
                                    const sbl_inst **pos, uint64_t *npos, const sbl_inst **neg, uint64_t *nneg)
{
	return guarded(c, [&] {
		SBL_CHECK(k >= 2, SBL_ERR_BAD_ARG, "vertex size k must be at least 2");
		SanitiseScope scope(c);
		sbl_run_enumeration(c, k, c->nelem);
		for (int st = 0; st < 2; st++) {
			sbl_compact_marks(c, st);
			unsigned n = c->nmarks[st];
			c->inst[st].resize(n);
			if (n) {
				c->d_inst.ensure((size_t)n * 12);
				k_make_instances<<<nblocks(n, 256), 256, 0, c->stream>>>(c->d_melem[st].as<unsigned>(), c->d_mid[st].as<unsigned>(), n,
				                                                        c->d_sepidx.as<unsigned>(), c->nchr, (unsigned)st, c->d_inst.as<unsigned>());
				HIP_TRY(hipMemcpyAsync(c->inst[st].data(), c->d_inst.p, (size_t)n * 12, hipMemcpyDeviceToHost, c->stream));
				HIP_TRY(hipStreamSynchronize(c->stream));
			}
		}
		// the negative list is sorted by (chr, reverse-complement pos): reverse each chromosome's run
		{
			auto &v = c->inst[1];
			size_t a = 0;
			while (a < v.size()) {
				size_t b = a;
				while (b < v.size() && v[b].chr == v[a].chr) b++;
				std::reverse(v.begin() + a, v.begin() + b);
				a = b;
			}
		}
		scope.done();
		c->stats.instances = c->inst[0].size() + c->inst[1].size();
		if (bif_count) *bif_count = c->bif_count;
		if (pos) *pos = c->inst[0].data();
		if (npos) *npos = c->inst[0].size();
		if (neg) *neg = c->inst[1].data();
		if (nneg) *nneg = c->inst[1].size();
	});
}

extern "C" sbl_status sbl_list_edges(sbl_ctx *c, uint32_t k, const sbl_edge **edges, uint64_t *n)
{
	return guarded(c, [&] {
		SBL_CHECK(k >= 2, SBL_ERR_BAD_ARG, "vertex size k must be at least 2");
		SanitiseScope scope(c);
		sbl_run_enumeration(c, k, c->nelem);
		c->edges.clear();
		DevBuf &d_edges = c->d_edges, &d_valid = c->d_valid;      // ctx-owned: nothing leaks when a later step throws
		for (int st = 0; st < 2; st++) {
			sbl_compact_marks(c, st);
			unsigned m = c->nmarks[st];
			if (m < 2) continue;
			d_edges.ensure((size_t)m * sizeof(sbl_edge)); d_valid.ensure(m);
			HIP_TRY(hipMemsetAsync(d_valid.p, 0, m, c->stream));
			k_list_edges<<<nblocks(m, 256), 256, 0, c->stream>>>(c->d_melem[st].as<unsigned>(), c->d_mid[st].as<unsigned>(), m, (unsigned)st, k,
			                                                    c->d_ch.as<uint8_t>(), c->d_op.as<unsigned>(), c->d_sepidx.as<unsigned>(), c->nchr,
			                                                    d_edges.as<sbl_edge>(), d_valid.as<uint8_t>());
			std::vector<sbl_edge> he(m);
			std::vector<uint8_t> hv(m);
			HIP_TRY(hipMemcpyAsync(he.data(), d_edges.p, (size_t)m * sizeof(sbl_edge), hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipMemcpyAsync(hv.data(), d_valid.p, m, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			size_t first = c->edges.size();
			for (unsigned i = 0; i + 1 < m; i++) if (hv[i]) c->edges.push_back(he[i]);
			if (st == 1) {                          // walk order on the negative strand is descending element order
				size_t a = first;
				while (a < c->edges.size()) {
					size_t b = a;
					while (b < c->edges.size() && c->edges[b].chr == c->edges[a].chr) b++;
					std::reverse(c->edges.begin() + a, c->edges.begin() + b);
					a = b;
				}
			}
		}
		scope.done();
		if (edges) *edges = c->edges.data();
		if (n) *n = c->edges.size();
	});
}

extern "C" sbl_status sbl_simplify_stage(sbl_ctx *c, uint32_t k, uint32_t min_branch_size, uint32_t max_iterations,
                                         sbl_progress_fn progress, void *user, uint64_t *bulges)
{
	return guarded(c, [&] {
		SBL_CHECK(k >= 2, SBL_ERR_BAD_ARG, "vertex size k must be at least 2");
		if (sanitise_apply(c)) sbl_sanitise_commit(c);       // the sanitised copy flows back through the copy-back (src/blockfinder.cpp:85-95)
		tempfile_draws(c);
		uint64_t b = 0;
		sbl_simplify_run(c, k, min_branch_size, max_iterations, progress, user, &b);
		drop_host_state(c);
		if (bulges) *bulges = b;
	});
}

extern "C" sbl_status sbl_set_tempfile_mode(sbl_ctx *c, int on)
{
	return guarded(c, [&] { c->tempfile_mode = on != 0; });
}
extern "C" sbl_status sbl_rand_advance(sbl_ctx *c, uint64_t n)
{
	return guarded(c, [&] { for (uint64_t i = 0; i < n; i++) (void)c->rng.next(); });
}

extern "C" sbl_status sbl_get_state(sbl_ctx *c, uint32_t chr, const uint8_t **seq, const uint32_t **orig_pos, uint64_t *len)
{
	return guarded(c, [&] {
		SBL_CHECK(chr < c->nchr, SBL_ERR_BAD_ARG, "chromosome index out of range");
		if (!c->host_state_valid) {
			if (c->nelem > c->h_cap) {                               // pinned staging, grown with slack (stages change the length by a few per cent)
				// the buffer being replaced is kept for one generation (see sbl_ctx::h_old_ch), the one before it goes
				c->host_free(c->h_old_ch, c->h_old_pinned); c->host_free(c->h_old_opos, c->h_old_pinned);
				c->h_old_ch = c->h_ch; c->h_old_opos = c->h_opos; c->h_old_pinned = c->h_pinned;
				c->h_ch = nullptr; c->h_opos = nullptr; c->h_cap = 0;
				const size_t cap = c->nelem + c->nelem / 8 + 4096;
				// 5 B per element pinned (1.4 GB for 62 strains x 4.6 Mbp); a host that cannot pin that much still gets its state, through
				// pageable memory (SBL_TEST_NO_PINNED_STATE=1: test switch)
				c->h_pinned = getenv("SBL_TEST_NO_PINNED_STATE") == nullptr
				              && hipHostMalloc((void **)&c->h_ch, cap) == hipSuccess && hipHostMalloc((void **)&c->h_opos, cap * 4) == hipSuccess;
				if (!c->h_pinned) {
					(void)hipGetLastError();
					if (c->h_ch) { (void)hipHostFree(c->h_ch); c->h_ch = nullptr; }
					c->h_opos = nullptr;
					c->h_ch = (uint8_t *)malloc(cap); c->h_opos = (uint32_t *)malloc(cap * 4);
					if (!c->h_ch || !c->h_opos) { free(c->h_ch); free(c->h_opos); c->h_ch = nullptr; c->h_opos = nullptr; throw SblError{SBL_ERR_OOM, "host staging buffer for the state"}; }
				}
				c->h_cap = cap;
			}
			// d_op never carries anything above the 29 position bits (the marks live in arrays of their own; the copy-back kernel masks)
			HIP_TRY(hipMemcpyAsync(c->h_ch, c->d_ch.p, c->nelem, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipMemcpyAsync(c->h_opos, c->d_op.p, c->nelem * 4, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
			c->host_state_valid = true;
		}
		const size_t a = (size_t)c->sepidx[chr] + 1, b = c->sepidx[chr + 1];
		if (seq) *seq = c->h_ch + a;
		if (orig_pos) *orig_pos = c->h_opos + a;
		if (len) *len = b - a;
	});
}

extern "C" sbl_status sbl_serialize_graph(sbl_ctx *c, uint32_t k, const char **text, uint64_t *len)
{
	return guarded(c, [&] {
		SBL_CHECK(k >= 1, SBL_ERR_BAD_ARG, "k must be at least 1");
		hipStream_t s = c->stream;
		std::vector<unsigned long long> off(c->nchr + 1, 0);
		for (uint32_t ch = 0; ch < c->nchr; ch++) {
			size_t n = c->sepidx[ch + 1] - c->sepidx[ch] - 1;
			off[ch + 1] = off[ch] + 2 * (n >= (size_t)k + 1 ? n - k : 0);       // (k+1)-windows on both strands
		}
		const unsigned long long nlines = off[c->nchr];
		const std::string head = "digraph G\n{\nrankdir=LR\n", tail = "}\n";
		c->graph_text = head;
		if (nlines) {
			SBL_CHECK(nlines < 0xFFFFFFF0ull, SBL_ERR_TOO_LARGE, "too many lines");
			DevBuf d_off, d_len, d_toff, d_text;                         // freed below (a debugging dump: no workspace kept)
			try {
				d_off.ensure((size_t)(c->nchr + 1) * 8); d_len.ensure((size_t)(nlines + 1) * 4); d_toff.ensure((size_t)(nlines + 1) * 8);
				HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), (size_t)(c->nchr + 1) * 8, hipMemcpyHostToDevice, s));
				HIP_TRY(hipMemsetAsync(d_len.p, 0, (size_t)(nlines + 1) * 4, s));
				k_graph_line_len<<<nblocks(nlines, 256), 256, 0, s>>>(d_off.as<unsigned long long>(), c->nchr, k, nlines, d_len.as<unsigned>());
				size_t tmp = 0;
				HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, d_len.as<unsigned>(), d_toff.as<unsigned long long>(), 0ull, (size_t)nlines + 1, rocprim::plus<unsigned long long>(), s));
				c->d_scantmp.ensure(tmp);
				HIP_TRY(rocprim::exclusive_scan(c->d_scantmp.p, tmp, d_len.as<unsigned>(), d_toff.as<unsigned long long>(), 0ull, (size_t)nlines + 1, rocprim::plus<unsigned long long>(), s));
				unsigned long long total = 0;
				HIP_TRY(hipMemcpyAsync(&total, d_toff.as<unsigned long long>() + nlines, 8, hipMemcpyDeviceToHost, s));
				HIP_TRY(hipStreamSynchronize(s));
				d_text.ensure(total + 16);
				k_graph_lines<<<nblocks(nlines, 256), 256, 0, s>>>(c->d_ch.as<uint8_t>(), c->d_sepidx.as<unsigned>(), d_off.as<unsigned long long>(), c->nchr, k, nlines,
				                                                  d_toff.as<unsigned long long>(), d_text.as<char>());
				HIP_TRY(hipGetLastError());
				c->graph_text.resize(head.size() + total);
				HIP_TRY(hipMemcpyAsync(&c->graph_text[head.size()], d_text.p, total, hipMemcpyDeviceToHost, s));
				HIP_TRY(hipStreamSynchronize(s));
			} catch (...) { d_off.release(); d_len.release(); d_toff.release(); d_text.release(); throw; }
			d_off.release(); d_len.release(); d_toff.release(); d_text.release();
		}
		c->graph_text += tail;
		if (text) *text = c->graph_text.data();
		if (len) *len = c->graph_text.size();
	});
}

extern "C" sbl_status sbl_kmer_hashes(sbl_ctx *c, uint32_t k, const uint64_t **values, uint64_t *n)
{
	return guarded(c, [&] {
		SBL_CHECK(k >= 1, SBL_ERR_BAD_ARG, "k must be at least 1");
		std::vector<unsigned long long> off(c->nchr + 1, 0);
		for (uint32_t ch = 0; ch < c->nchr; ch++) {
			size_t len = c->sepidx[ch + 1] - c->sepidx[ch] - 1;
			off[ch + 1] = off[ch] + (len >= k ? len - k + 1 : 0);
		}
		const unsigned long long per = off[c->nchr];
		c->h_hashes.resize(2 * per);
		if (per) {
			c->d_inst.ensure(2 * per * 8 + (size_t)(c->nchr + 1) * 8 + 64);
			unsigned long long *d_out = c->d_inst.as<unsigned long long>(), *d_off = d_out + 2 * per;
			HIP_TRY(hipMemcpyAsync(d_off, off.data(), (size_t)(c->nchr + 1) * 8, hipMemcpyHostToDevice, c->stream));
			k_kmer_hashes<<<nblocks(2 * per, 256), 256, 0, c->stream>>>(c->d_ch.as<uint8_t>(), c->d_sepidx.as<unsigned>(), c->nchr, k, d_off, per, d_out);
			HIP_TRY(hipGetLastError());
			HIP_TRY(hipMemcpyAsync(c->h_hashes.data(), d_out, 2 * per * 8, hipMemcpyDeviceToHost, c->stream));
			HIP_TRY(hipStreamSynchronize(c->stream));
		}
		if (values) *values = c->h_hashes.data();
		if (n) *n = 2 * per;
	});
}

extern "C" sbl_status sbl_save_state(sbl_ctx *c)
{
	return guarded(c, [&] {
		size_t Epad = (c->nelem + 31) / 32 * 32 + 64;
		c->d_save_ch.ensure(Epad); c->d_save_op.ensure(c->nelem * 4 + 16);
		HIP_TRY(hipMemcpyAsync(c->d_save_ch.p, c->d_ch.p, Epad, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->d_save_op.p, c->d_op.p, c->nelem * 4, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		c->save_nelem = c->nelem; c->save_sepidx = c->sepidx; c->save_amb_elem = c->amb_elem; c->save_amb_orig = c->amb_orig;
		c->save_rng = c->rng; c->saved = true;
	});
}
extern "C" sbl_status sbl_restore_state(sbl_ctx *c)
{
	return guarded(c, [&] {
		SBL_CHECK(c->saved, SBL_ERR_BAD_ARG, "no saved state");
		size_t Epad = (c->save_nelem + 31) / 32 * 32 + 64;
		c->d_ch.ensure(Epad); c->d_op.ensure(c->save_nelem * 4 + 16);
		HIP_TRY(hipMemcpyAsync(c->d_ch.p, c->d_save_ch.p, Epad, hipMemcpyDeviceToDevice, c->stream));
		HIP_TRY(hipMemcpyAsync(c->d_op.p, c->d_save_op.p, c->save_nelem * 4, hipMemcpyDeviceToDevice, c->stream));
		c->nelem = c->save_nelem; c->sepidx = c->save_sepidx; c->amb_elem = c->save_amb_elem; c->amb_orig = c->save_amb_orig;
		c->rng = c->save_rng;
		HIP_TRY(hipMemcpyAsync(c->d_sepidx.p, c->sepidx.data(), c->sepidx.size() * 4, hipMemcpyHostToDevice, c->stream));
		HIP_TRY(hipStreamSynchronize(c->stream));
		drop_host_state(c);
	});
}

extern "C" sbl_status sbl_last_stats(const sbl_ctx *c, sbl_stage_stats *out)
{
	if (!c || !out) return SBL_ERR_BAD_ARG;
	*out = c->stats;
	return SBL_OK;
}
extern "C" const char *sbl_last_error(const sbl_ctx *c) { return c ? c->err.c_str() : "null context"; }
extern "C" const char *sbl_strerror(sbl_status s)
{
	switch (s) {
	case SBL_OK: return "ok";
	case SBL_ERR_BAD_ARG: return "bad argument";
	case SBL_ERR_NO_DEVICE: return "no usable HIP device (this library has no host compute path)";
	case SBL_ERR_OOM: return "out of memory";
	case SBL_ERR_HIP: return "HIP runtime error";
	case SBL_ERR_TOO_LARGE: return "input exceeds the 29-bit position / 2^30 total limits";
	case SBL_ERR_UNSUPPORTED: return "unsupported vertex size";
	default: return "internal error";
	}
}
extern "C" sbl_status sbl_set_window(sbl_ctx *c, uint32_t w) { if (!c) return SBL_ERR_BAD_ARG; c->window = w; return SBL_OK; }
