// fasta_load.hip -- upstream of the hot path: FASTA text -> HBM-resident BlockFinder state, on the device.
//
// Replaces FASTAReader::GetSequences (reference src/fasta.cpp:23-104) + BlockFinder::Init (src/blockfinder.cpp:65-76) +
// the scan for indefinite bases of IndexedSequence::Init (src/indexedsequence.cpp:31-37).  The host only maps the file and
// copies its bytes to the device; line splitting, trimming, header / sequence classification, upper-casing, validation,
// concatenation into the element array '$' c0 '$' c1 '$' ..., the identity original positions and the list of non-ACGT
// positions are all computed by kernels -- no per-base host loop, no 4 B/base host staging.
//   F1/F2  newline compaction           -> line_start[]                               (stream, 1 B/byte)
//   F3     one thread per line           -> trimmed span [a, b), kind, header validity (reads only the ends of a line)
//   scans  (rocPRIM)                     -> non-empty line numbers, record index, sequence offsets
//   F4/F5  one thread per BYTE           -> upper-case, validate, scatter into ch[]     (line found by binary search)
//   L1/L2  state kernels (shared with sbl_load): op[] = position in the record, compaction of the non-ACGT elements
// Semantics kept from the reference: boost::trim of every line, empty lines skipped and not counted, header name = text
// between '>' and the first blank, sequence characters upper-cased and checked against "ACGTURYKMSWBDHWNX-", sequence lines
// ahead of the first header join the first record, "empty sequence" / "empty header" / "illegal character" errors with the
// reference's line numbering (1 + number of non-empty lines before).
#include <cstring>
#include <algorithm>
#include <string>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rocprim/rocprim.hpp>

#include "sbl_ctx.h"
#include "kmer_kernels.h"

static inline unsigned nblocks(size_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

__device__ __forceinline__ bool fa_space(uint8_t c) { return c == ' ' || (c >= 9 && c <= 13); }      // std::isspace in the "C" locale (boost::trim)

#define FA_CHUNK 4096u
__global__ void __launch_bounds__(256) k_fa_count_nl(const uint8_t *__restrict__ txt, size_t n, unsigned *__restrict__ cnt)
{
	__shared__ unsigned s;
	if (threadIdx.x == 0) s = 0;
	__syncthreads();
	size_t base = (size_t)blockIdx.x * FA_CHUNK + (size_t)threadIdx.x * 16;
	unsigned c = 0;
	for (int i = 0; i < 16; i++) if (base + i < n && txt[base + i] == '\n') c++;
	if (c) atomicAdd(&s, c);
	__syncthreads();
	if (threadIdx.x == 0) cnt[blockIdx.x] = s;
}
// line 0 starts at 0, line j + 1 right after the j-th newline
__global__ void __launch_bounds__(256) k_fa_line_starts(const uint8_t *__restrict__ txt, size_t n, const unsigned *__restrict__ off, unsigned *__restrict__ line_start)
{
	__shared__ unsigned wsum[4];
	size_t base = (size_t)blockIdx.x * FA_CHUNK + (size_t)threadIdx.x * 16;
	unsigned c = 0;
	for (int i = 0; i < 16; i++) if (base + i < n && txt[base + i] == '\n') c++;
	unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6, incl = c;
#pragma unroll
	for (int d = 1; d < 64; d <<= 1) { unsigned v = __shfl_up(incl, d); if (lane >= (unsigned)d) incl += v; }
	if (lane == 63) wsum[wv] = incl;
	__syncthreads();
	unsigned at = off[blockIdx.x] + incl - c;
	for (unsigned w = 0; w < wv; w++) at += wsum[w];
	if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
	for (int i = 0; i < 16; i++) if (base + i < n && txt[base + i] == '\n') line_start[1 + at++] = (unsigned)(base + i + 1);
}

enum { FA_EMPTY = 0, FA_HEADER = 1, FA_SEQ = 2 };
// err: [0] = (line index << 8 | code) of the first offending line (atomicMin); codes: 1 empty header
__global__ void __launch_bounds__(256) k_fa_lines(const uint8_t *__restrict__ txt, size_t n, const unsigned *__restrict__ line_start, unsigned nlines,
                                                  unsigned *__restrict__ la, unsigned *__restrict__ lb, unsigned *__restrict__ nonempty, unsigned *__restrict__ isheader,
                                                  unsigned *__restrict__ seqlen, unsigned long long *__restrict__ err)
{
	unsigned l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= nlines) return;
	size_t a = line_start[l], b = l + 1 < nlines ? (size_t)line_start[l + 1] - 1 : n;      // [a, b) without the newline
	while (a < b && fa_space(txt[a])) a++;
	while (b > a && fa_space(txt[b - 1])) b--;
	unsigned kind = a == b ? FA_EMPTY : txt[a] == '>' ? FA_HEADER : FA_SEQ;
	if (kind == FA_HEADER && (b - a == 1 || txt[a + 1] == ' ')) atomicMin(err, ((unsigned long long)l << 40) | 1ull);   // ValidateHeader: empty name
	la[l] = (unsigned)a; lb[l] = (unsigned)b;
	nonempty[l] = kind != FA_EMPTY; isheader[l] = kind == FA_HEADER; seqlen[l] = kind == FA_SEQ ? (unsigned)(b - a) : 0u;
}
// per line: element index of its first character; per header line: record bookkeeping
__global__ void __launch_bounds__(256) k_fa_place(unsigned nlines, const unsigned *__restrict__ isheader, const unsigned *__restrict__ hdr_before,
                                                  const unsigned *__restrict__ seq_before, const unsigned *__restrict__ nonempty_before,
                                                  unsigned *__restrict__ elem0, unsigned *__restrict__ rec_start, unsigned *__restrict__ rec_line, unsigned *__restrict__ rec_lineno)
{
	unsigned l = blockIdx.x * blockDim.x + threadIdx.x;
	if (l >= nlines) return;
	unsigned h = hdr_before[l];                                  // headers strictly before this line
	if (isheader[l]) {
		// sequence lines ahead of the first header stay in the first record (fasta.cpp:41-50: nothing is pushed for an empty header)
		rec_start[h] = h == 0 ? 0u : seq_before[l];
		rec_line[h] = l; rec_lineno[h] = 1 + nonempty_before[l];
		elem0[l] = 0;
	} else {
		unsigned rec = h ? h - 1 : 0;
		elem0[l] = 1 + seq_before[l] + rec;                      // one '$' ahead of every record
	}
}
// codes: 2 illegal character (low byte of the payload = the character as written)
__global__ void __launch_bounds__(256) k_fa_scatter(const uint8_t *__restrict__ txt, size_t n, const unsigned *__restrict__ line_start, unsigned nlines,
                                                    const unsigned *__restrict__ la, const unsigned *__restrict__ lb, const unsigned *__restrict__ isheader,
                                                    const unsigned *__restrict__ elem0, uint8_t *__restrict__ ch, unsigned long long *__restrict__ err)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned lo = 0, hi = nlines;                               // the line that holds byte i: last line_start <= i
	while (hi - lo > 1) { unsigned mid = (lo + hi) >> 1; if (line_start[mid] <= i) lo = mid; else hi = mid; }
	const unsigned l = lo, a = la[l], b = lb[l];
	if (isheader[l] || i < a || i >= b) return;
	const uint8_t orig = txt[i];
	const uint8_t c = orig >= 'a' && orig <= 'z' ? orig - 32 : orig;
	bool ok = false;
	switch (c) { case 'A': case 'C': case 'G': case 'T': case 'U': case 'R': case 'Y': case 'K': case 'M': case 'S': case 'W': case 'B': case 'D': case 'H': case 'N': case 'X': case '-': ok = true; }
	if (!ok) { atomicMin(err, ((unsigned long long)l << 40) | ((unsigned long long)((i - a) < 0xFFFFFFull ? (i - a) : 0xFFFFFFull) << 16) | ((unsigned long long)orig << 8) | 2ull); return; }   // column clamped to its 24-bit field
	ch[(size_t)elem0[l] + (i - a)] = c;
}

// ---- state kernels shared with sbl_load (sbl_api.hip)
// originalPos_ = identity (Counter<Pos>, blockfinder.cpp:74); every trailing '$' stores the length (dnasequence.cpp:96)
__global__ void __launch_bounds__(256) k_identity_positions(const uint8_t *__restrict__ ch, size_t nelem, const unsigned *__restrict__ sepidx, unsigned nchr,
                                                            unsigned *__restrict__ op, unsigned *__restrict__ amb_count)
{
	size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	bool amb = false;
	if (e < nelem) {
		// chr_of(e) = the chromosome c with sepidx[c] < e <= sepidx[c + 1]: a letter of c, or the '$' that ends c (which stores len_c)
		op[e] = e ? (unsigned)e - sepidx[chr_of(sepidx, nchr, (unsigned)e)] - 1u : 0u;
		const uint8_t x = ch[e];
		amb = x != '$' && x != 'A' && x != 'C' && x != 'G' && x != 'T';
	}
	unsigned long long m = __ballot(amb);
	if (m && (threadIdx.x & 63) == 0) atomicAdd(amb_count, (unsigned)__popcll(m));
}
__global__ void __launch_bounds__(256) k_amb_flags(const uint8_t *__restrict__ ch, size_t nelem, unsigned *__restrict__ chunkcnt)
{
	__shared__ unsigned s;
	if (threadIdx.x == 0) s = 0;
	__syncthreads();
	size_t base = (size_t)blockIdx.x * 1024;
	unsigned c = 0;
	for (unsigned i = threadIdx.x; i < 1024; i += 256) { size_t e = base + i; if (e < nelem) { uint8_t x = ch[e]; c += x != 'A' && x != 'C' && x != 'G' && x != 'T' && x != '$'; } }
	if (c) atomicAdd(&s, c);
	__syncthreads();
	if (threadIdx.x == 0) chunkcnt[blockIdx.x] = s;
}
// ordered compaction (element order = chromosome-major order of the reference's scan); the few chunks that hold any are walked by one thread
__global__ void __launch_bounds__(256) k_amb_write(const uint8_t *__restrict__ ch, size_t nelem, const unsigned *__restrict__ chunkcnt, const unsigned *__restrict__ chunkoff,
                                                   unsigned nchunks, unsigned *__restrict__ out_elem, uint8_t *__restrict__ out_char)
{
	unsigned cidx = blockIdx.x * blockDim.x + threadIdx.x;
	if (cidx >= nchunks || chunkcnt[cidx] == 0) return;
	unsigned at = chunkoff[cidx];
	size_t base = (size_t)cidx * 1024;
	for (unsigned i = 0; i < 1024; i++) {
		size_t e = base + i;
		if (e >= nelem) break;
		uint8_t x = ch[e];
		if (x != 'A' && x != 'C' && x != 'G' && x != 'T' && x != '$') { out_elem[at] = (unsigned)e; out_char[at] = x; at++; }
	}
}

static void scan_u32(sbl_ctx *c, const unsigned *in, unsigned *out, size_t n)
{
	size_t tmp = 0;
	HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
	c->d_scantmp.ensure(tmp);
	HIP_TRY(rocprim::exclusive_scan(c->d_scantmp.p, tmp, in, out, 0u, n, rocprim::plus<unsigned>(), c->stream));
}

// d_ch (padded with '$'), d_sepidx and c->sepidx / nchr / nelem are in place: derive op[] and the ambiguity list on the device
void sbl_finish_load(sbl_ctx *c)
{
	hipStream_t s = c->stream;
	const size_t E = c->nelem;
	c->d_op.ensure(E * 4 + 16);
	c->d_counters.ensure(64 * 4);
	HIP_TRY(hipMemsetAsync(c->d_counters.p, 0, 64 * 4, s));
	k_identity_positions<<<nblocks(E, 256), 256, 0, s>>>(c->d_ch.as<uint8_t>(), E, c->d_sepidx.as<unsigned>(), c->nchr, c->d_op.as<unsigned>(), c->d_counters.as<unsigned>());
	HIP_TRY(hipGetLastError());
	unsigned namb = 0;
	HIP_TRY(hipMemcpyAsync(&namb, c->d_counters.p, 4, hipMemcpyDeviceToHost, s));
	HIP_TRY(hipStreamSynchronize(s));
	c->amb_elem.assign(namb, 0); c->amb_orig.assign(namb, 0);
	if (namb) {
		unsigned nchunks = nblocks(E, 1024);
		c->d_chunkcnt.ensure((size_t)(nchunks + 1) * 4); c->d_chunkoff.ensure((size_t)(nchunks + 1) * 4);
		HIP_TRY(hipMemsetAsync(c->d_chunkcnt.p, 0, (size_t)(nchunks + 1) * 4, s));
		k_amb_flags<<<nchunks, 256, 0, s>>>(c->d_ch.as<uint8_t>(), E, c->d_chunkcnt.as<unsigned>());
		scan_u32(c, c->d_chunkcnt.as<unsigned>(), c->d_chunkoff.as<unsigned>(), nchunks + 1);
		c->d_amb_elem.ensure((size_t)namb * 4); c->d_amb_char.ensure(namb);
		k_amb_write<<<nblocks(nchunks, 256), 256, 0, s>>>(c->d_ch.as<uint8_t>(), E, c->d_chunkcnt.as<unsigned>(), c->d_chunkoff.as<unsigned>(), nchunks,
		                                                 c->d_amb_elem.as<unsigned>(), c->d_amb_char.as<uint8_t>());
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipMemcpyAsync(c->amb_elem.data(), c->d_amb_elem.p, (size_t)namb * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(c->amb_orig.data(), c->d_amb_char.p, namb, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
	}
	c->host_state_valid = false;
	// keep the records as loaded: the synteny stage trims on them
	c->d_orig_ch.ensure(E + 64);
	HIP_TRY(hipMemcpyAsync(c->d_orig_ch.p, c->d_ch.p, E, hipMemcpyDeviceToDevice, s));
	HIP_TRY(hipStreamSynchronize(s));
	c->orig_sepidx = c->sepidx;
}

namespace {
struct Mapped {
	int fd = -1; void *p = MAP_FAILED; size_t n = 0;
	~Mapped() { if (p != MAP_FAILED) munmap(p, n); if (fd >= 0) close(fd); }
};
}

extern "C" sbl_status sbl_load_fasta(sbl_ctx *c, const char *path)
{
	return guarded(c, [&] {
		SBL_CHECK(path, SBL_ERR_BAD_ARG, "null path");
		Mapped m;
		m.fd = open(path, O_RDONLY);
		SBL_CHECK(m.fd >= 0, SBL_ERR_BAD_ARG, std::string("cannot open ") + path);
		struct stat st;
		SBL_CHECK(fstat(m.fd, &st) == 0, SBL_ERR_BAD_ARG, std::string("cannot stat ") + path);
		const size_t n = (size_t)st.st_size;
		SBL_CHECK(n > 0, SBL_ERR_BAD_ARG, std::string("parse error in ") + path + " on line 1: empty sequence");
		SBL_CHECK(n < 0xFFFFFF00ull, SBL_ERR_TOO_LARGE, "FASTA file larger than 4 GB (the input cap is 2^30 bp)");
		m.n = n;
		m.p = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, m.fd, 0);
		SBL_CHECK(m.p != MAP_FAILED, SBL_ERR_OOM, "mmap failed");
		hipStream_t s = c->stream;
		auto fail = [&](unsigned lineno, const std::string &what) {
			throw SblError{SBL_ERR_BAD_ARG, std::string("parse error in ") + path + " on line " + std::to_string(lineno) + ": " + what};
		};

		// ---- the text goes to the device as it is (the only host-side touch of the data)
		DevBuf &txt = c->d_fa_text;
		txt.ensure(n + 64);
		HIP_TRY(hipMemcpyAsync(txt.p, m.p, n, hipMemcpyHostToDevice, s));
		const unsigned nchk = nblocks(n, FA_CHUNK);
		c->d_chunkcnt.ensure((size_t)(nchk + 1) * 4); c->d_chunkoff.ensure((size_t)(nchk + 1) * 4);
		HIP_TRY(hipMemsetAsync(c->d_chunkcnt.p, 0, (size_t)(nchk + 1) * 4, s));
		k_fa_count_nl<<<nchk, 256, 0, s>>>(txt.as<uint8_t>(), n, c->d_chunkcnt.as<unsigned>());
		scan_u32(c, c->d_chunkcnt.as<unsigned>(), c->d_chunkoff.as<unsigned>(), nchk + 1);
		unsigned nnl = 0;
		HIP_TRY(hipMemcpyAsync(&nnl, c->d_chunkoff.as<unsigned>() + nchk, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		const unsigned nlines = nnl + 1;
		// per-line tables: start, [a, b), flags + their exclusive scans (one extra slot = totals)
		DevBuf &L = c->d_fa_lines;
		const size_t stride = (size_t)nlines + 1;
		L.ensure(stride * 4 * 10 + 64);
		HIP_TRY(hipMemsetAsync(L.p, 0, stride * 4 * 10, s));
		unsigned *line_start = L.as<unsigned>(), *la = line_start + stride, *lb = la + stride, *nonempty = lb + stride, *isheader = nonempty + stride,
		         *seqlen = isheader + stride, *ne_before = seqlen + stride, *hdr_before = ne_before + stride, *seq_before = hdr_before + stride, *elem0 = seq_before + stride;
		c->d_counters.ensure(64 * 4);
		unsigned long long *err = c->d_counters.as<unsigned long long>() + 8;
		HIP_TRY(hipMemsetAsync(err, 0xFF, 8, s));
		k_fa_line_starts<<<nchk, 256, 0, s>>>(txt.as<uint8_t>(), n, c->d_chunkoff.as<unsigned>(), line_start);
		k_fa_lines<<<nblocks(nlines, 256), 256, 0, s>>>(txt.as<uint8_t>(), n, line_start, nlines, la, lb, nonempty, isheader, seqlen, err);
		scan_u32(c, nonempty, ne_before, stride);
		scan_u32(c, isheader, hdr_before, stride);
		scan_u32(c, seqlen, seq_before, stride);
		unsigned tot[3];
		HIP_TRY(hipMemcpyAsync(&tot[0], ne_before + nlines, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(&tot[1], hdr_before + nlines, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(&tot[2], seq_before + nlines, 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		const unsigned nheaders = tot[1], nrec = nheaders ? nheaders : 1u;      // no header at all: one record with an empty description (fasta.cpp:62-63)
		const size_t Ltot = tot[2];
		SBL_CHECK(Ltot <= (1ull << 30), SBL_ERR_TOO_LARGE, "total input larger than 2^30 bp");
		const size_t E = Ltot + nrec + 1, Epad = (E + 31) / 32 * 32 + 64;
		c->d_ch.ensure(Epad);
		HIP_TRY(hipMemsetAsync(c->d_ch.p, '$', Epad, s));
		DevBuf &R = c->d_fa_recs;
		R.ensure(((size_t)nrec + 1) * 4 * 3 + 64);
		HIP_TRY(hipMemsetAsync(R.p, 0, ((size_t)nrec + 1) * 4 * 3, s));
		unsigned *rec_start = R.as<unsigned>(), *rec_line = rec_start + nrec + 1, *rec_lineno = rec_line + nrec + 1;
		k_fa_place<<<nblocks(nlines, 256), 256, 0, s>>>(nlines, isheader, hdr_before, seq_before, ne_before, elem0, rec_start, rec_line, rec_lineno);
		k_fa_scatter<<<nblocks(n, 256), 256, 0, s>>>(txt.as<uint8_t>(), n, line_start, nlines, la, lb, isheader, elem0, c->d_ch.as<uint8_t>(), err);
		HIP_TRY(hipGetLastError());
		std::vector<unsigned> hstart(nrec + 1, 0), hline(nrec + 1, 0), hlineno(nrec + 1, 0);
		unsigned long long herr = ~0ull;
		HIP_TRY(hipMemcpyAsync(hstart.data(), rec_start, (size_t)nrec * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(hline.data(), rec_line, (size_t)nrec * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(hlineno.data(), rec_lineno, (size_t)nrec * 4, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipMemcpyAsync(&herr, err, 8, hipMemcpyDeviceToHost, s));
		HIP_TRY(hipStreamSynchronize(s));
		hstart[nrec] = (unsigned)Ltot; hlineno[nrec] = 1 + tot[0];
		// ---- errors, in the order the reference would meet them: the first offending line wins
		unsigned long long first = ~0ull; std::string what;
		if (herr != ~0ull) {
			unsigned l = (unsigned)(herr >> 40);
			SBL_CHECK(l < nlines, SBL_ERR_INTERNAL, "FASTA loader: corrupt error record");
			// line number of line l = 1 + non-empty lines before it
			unsigned before = 0;
			HIP_TRY(hipMemcpy(&before, ne_before + l, 4, hipMemcpyDeviceToHost));
			first = ((unsigned long long)l << 1) | 1ull;
			what = (herr & 0xFF) == 1 ? "empty header" : std::string("illegal character: ") + (char)((herr >> 8) & 0xFF);
			hlineno.push_back(1 + before);
		}
		for (unsigned r = 0; r < nrec; r++) {
			if (hstart[r + 1] > hstart[r]) continue;                           // record r has a sequence
			// the reference notices at the NEXT header (or at the end of the file)
			unsigned long long at = r + 1 < nrec ? ((unsigned long long)hline[r + 1] << 1) : ((unsigned long long)nlines << 1);
			if (at < first) { first = at; what = "empty sequence"; hlineno.push_back(r + 1 < nrec ? hlineno[r + 1] : 1 + tot[0]); }
			break;
		}
		if (first != ~0ull) fail(hlineno.back(), what);
		// ---- record table
		c->nchr = nrec; c->nelem = E;
		c->hint_elem_slack = 0; c->hint_cap_n = 0; c->hint_checkpoints = false;      // (as sbl_load)
		c->sepidx.assign(nrec + 1, 0);
		for (unsigned r = 0; r <= nrec; r++) {
			c->sepidx[r] = hstart[r] + r;
			if (r < nrec) SBL_CHECK(hstart[r + 1] - hstart[r] < (1u << 29), SBL_ERR_TOO_LARGE, "a chromosome must be shorter than 2^29 bp (29-bit original positions)");
		}
		c->d_sepidx.ensure((size_t)(nrec + 1) * 4);
		HIP_TRY(hipMemcpyAsync(c->d_sepidx.p, c->sepidx.data(), (size_t)(nrec + 1) * 4, hipMemcpyHostToDevice, s));
		// descriptions: text between '>' and the first blank of each header line (host: nrec short strings out of the mapped file)
		c->fa_names.assign(nrec, std::string());
		if (nheaders) {
			std::vector<unsigned> ha(nheaders), hb(nheaders);
			for (unsigned r = 0; r < nheaders; r++) {
				HIP_TRY(hipMemcpyAsync(&ha[r], la + hline[r], 4, hipMemcpyDeviceToHost, s));
				HIP_TRY(hipMemcpyAsync(&hb[r], lb + hline[r], 4, hipMemcpyDeviceToHost, s));
			}
			HIP_TRY(hipStreamSynchronize(s));
			const char *t = static_cast<const char *>(m.p);
			for (unsigned r = 0; r < nheaders; r++) {
				size_t a = ha[r] + 1, b = hb[r], sp = a;
				while (sp < b && t[sp] != ' ') sp++;
				c->fa_names[r].assign(t + a, t + sp);
			}
		}
		sbl_finish_load(c);
	});
}

extern "C" const char *sbl_record_name(const sbl_ctx *c, uint32_t chr)
{
	return c && chr < c->fa_names.size() ? c->fa_names[chr].c_str() : "";
}
