/*
 * sibelia_oracle.c -- CPU ORACLE (test infrastructure, see sibelia_oracle.h).
 *
 * Every function names the reference file:line it restates.  Storage is NOT the
 * reference's (no unrolled list, no Boost, no suffix array): the sequence is an
 * index-linked list over flat arrays with stable element identity, bifurcation
 * marks are two dense arrays, per-id instance lists are index-linked nodes, and
 * enumeration groups k-mers by sorting packed codes (k <= 32) or by rank doubling
 * (k > 32).  Only the observable semantics are the reference's.
 */
#define _POSIX_C_SOURCE 200809L
#include "sibelia_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define NONE 0xFFFFFFFFu           /* BifurcationStorage::NO_BIFURCATION, src/bifurcationstorage.cpp:12 */
#define SEP '$'                    /* DNASequence::SEPARATION_CHAR, src/dnasequence.cpp:33 */
#define POS_MASK 0x1FFFFFFFu       /* StrandIterator::PositionMask(), src/stranditerator.cpp:19-27 */
#define EMPTY_CHAR ' '             /* bulgeremoval.cpp:13 */

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void *xrealloc(void *p, size_t n)
{
	void *q = realloc(p, n ? n : 1);
	if (!q) { fprintf(stderr, "oracle: out of memory (%zu bytes)\n", n); abort(); }
	return q;
}
#define GROW(ptr, cap, need, type) do { \
	if ((need) > (cap)) { \
		size_t nc_ = (cap) ? (cap) : 16; \
		while (nc_ < (need)) { nc_ *= 2; } \
		(ptr) = (type *)xrealloc((ptr), nc_ * sizeof(type)); \
		(cap) = nc_; \
	} } while (0)

/* ------------------------------------------------------------------ glibc rand() (TYPE_3) */
/* The reference calls unseeded rand() (src/indexedsequence.cpp:35).  glibc's default
 * generator is the additive feedback r[i] = r[i-3] + r[i-31] seeded by srandom(1). */
typedef struct { int32_t r[34]; int f, b; } grand;
static void grand_seed(grand *g, uint32_t seed)
{
	int i;
	int32_t word = seed ? (int32_t)seed : 1;
	g->r[0] = word;
	for (i = 1; i < 31; i++) {
		long hi = word / 127773, lo = word % 127773;
		word = (int32_t)(16807 * lo - 2836 * hi);
		if (word < 0) word += 2147483647;
		g->r[i] = word;
	}
	g->f = 3; g->b = 0;
	for (i = 0; i < 310; i++) {
		g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]);
		g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31;
	}
}
static uint32_t grand_next(grand *g)
{
	uint32_t v;
	g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]);
	v = ((uint32_t)g->r[g->f] >> 1) & 0x7fffffffu;
	g->f = (g->f + 1) % 31; g->b = (g->b + 1) % 31;
	return v;
}

/* ------------------------------------------------------------------ context */
typedef struct { uint32_t key; uint32_t strand, node; } clear_rec;

struct orc_ctx {
	/* state carried between stages: rawSeq_, originalPos_ (src/blockfinder.h:52-54) */
	uint32_t nchr;
	uint8_t **seq; uint32_t **opos; uint64_t *len;
	grand rng;
	int force_long;
	/* enumeration output */
	orc_inst *inst[2]; size_t ninst[2], cinst[2];
	uint32_t bif_count;
	/* editable sequence (DNASequence, src/dnasequence.cpp:75-103): index-linked elements */
	size_t ne, ce;                 /* used / capacity */
	char *ch; uint32_t *op; uint32_t *nx, *pv;
	uint32_t *bif[2];              /* id of the k-mer STARTING at this element on strand s */
	uint32_t *nodeof[2];           /* instance node of that mark */
	uint32_t *sep;                 /* sep[c] = element index of the '$' before chromosome c (nchr+1 entries) */
	/* BifurcationStorage (src/bifurcationstorage.cpp): per (strand,id) front-inserted lists */
	uint32_t *head[2]; uint32_t *lsize[2];
	uint32_t *nslot, *nnext; uint8_t *ndead; size_t nn, cn;
	clear_rec *toclear; size_t ntoclear, ctoclear;
	uint32_t k;
	/* edges */
	orc_edge *edge; size_t nedge, cedge;
	double t_enum, t_simp, t_copy;
};

orc_ctx *orc_create(void)
{
	orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
	grand_seed(&c->rng, 1);
	return c;
}

static void free_index(orc_ctx *c)
{
	int s;
	free(c->ch); free(c->op); free(c->nx); free(c->pv); free(c->sep);
	c->ch = 0; c->op = 0; c->nx = c->pv = 0; c->sep = 0;
	for (s = 0; s < 2; s++) {
		free(c->bif[s]); free(c->nodeof[s]); free(c->head[s]); free(c->lsize[s]);
		c->bif[s] = c->nodeof[s] = c->head[s] = c->lsize[s] = 0;
	}
	free(c->nslot); free(c->nnext); free(c->ndead); c->nslot = c->nnext = 0; c->ndead = 0;
	c->nn = c->cn = 0; c->ne = c->ce = 0;
}

void orc_destroy(orc_ctx *c)
{
	uint32_t i;
	if (!c) return;
	for (i = 0; i < c->nchr; i++) { free(c->seq[i]); free(c->opos[i]); }
	free(c->seq); free(c->opos); free(c->len);
	free(c->inst[0]); free(c->inst[1]);
	free_index(c);
	free(c->toclear); free(c->edge);
	free(c);
}

uint32_t orc_nchr(const orc_ctx *c) { return c->nchr; }
void orc_force_long_k_path(orc_ctx *c, int on) { c->force_long = on; }
uint32_t orc_rand(orc_ctx *c) { return grand_next(&c->rng); }
void orc_rng_copy(orc_ctx *dst, const orc_ctx *src) { dst->rng = src->rng; }
void orc_last_timing(const orc_ctx *c, double *e, double *s, double *b)
{ if (e) *e = c->t_enum; if (s) *s = c->t_simp; if (b) *b = c->t_copy; }

int orc_load(orc_ctx *c, uint32_t nchr, const uint8_t *const *seq, const uint64_t *len)
{
	uint32_t i; uint64_t j;
	for (i = 0; i < c->nchr; i++) { free(c->seq[i]); free(c->opos[i]); }
	free(c->seq); free(c->opos); free(c->len);
	c->nchr = nchr;
	c->seq = (uint8_t **)calloc(nchr ? nchr : 1, sizeof(*c->seq));
	c->opos = (uint32_t **)calloc(nchr ? nchr : 1, sizeof(*c->opos));
	c->len = (uint64_t *)calloc(nchr ? nchr : 1, sizeof(*c->len));
	for (i = 0; i < nchr; i++) {
		c->len[i] = len[i];
		c->seq[i] = (uint8_t *)xrealloc(0, len[i]);
		c->opos[i] = (uint32_t *)xrealloc(0, len[i] * 4);
		memcpy(c->seq[i], seq[i], len[i]);
		for (j = 0; j < len[i]; j++) c->opos[i][j] = (uint32_t)j;   /* Counter<Pos>, blockfinder.cpp:74 */
	}
	return 0;
}

int orc_get_state(orc_ctx *c, uint32_t chr, const uint8_t **seq, const uint32_t **orig_pos, uint64_t *len)
{
	if (chr >= c->nchr) return 1;
	if (seq) *seq = c->seq[chr];
	if (orig_pos) *orig_pos = c->opos[chr];
	if (len) *len = c->len[chr];
	return 0;
}

/* ------------------------------------------------------------------ T2: sanitise
 * IndexedSequence::Init, src/indexedsequence.cpp:31-37: every char not in "ACGT" becomes
 * "ACGT"[rand() % 4], chromosome-major. */
static uint8_t **sanitised_copy(orc_ctx *c)
{
	uint32_t i; uint64_t j;
	uint8_t **r = (uint8_t **)calloc(c->nchr ? c->nchr : 1, sizeof(*r));
	for (i = 0; i < c->nchr; i++) {
		r[i] = (uint8_t *)xrealloc(0, c->len[i]);
		for (j = 0; j < c->len[i]; j++) {
			uint8_t x = c->seq[i][j];
			if (x != 'A' && x != 'C' && x != 'G' && x != 'T') x = (uint8_t)"ACGT"[grand_next(&c->rng) % 4];
			r[i][j] = x;
		}
	}
	return r;
}
static void free_copy(orc_ctx *c, uint8_t **r)
{
	uint32_t i;
	for (i = 0; i < c->nchr; i++) free(r[i]);
	free(r);
}

/* ------------------------------------------------------------------ radix sort of (key,val) */
static void radix_sort_kv(uint64_t *key, uint32_t *val, size_t n, int key_bits)
{
	uint64_t *k2 = (uint64_t *)xrealloc(0, n * 8);
	uint32_t *v2 = (uint32_t *)xrealloc(0, n * 4);
	size_t *cnt = (size_t *)xrealloc(0, 65536 * sizeof(size_t));
	int shift;
	for (shift = 0; shift < key_bits; shift += 16) {
		size_t i, sum = 0;
		memset(cnt, 0, 65536 * sizeof(size_t));
		for (i = 0; i < n; i++) cnt[(key[i] >> shift) & 0xFFFF]++;
		for (i = 0; i < 65536; i++) { size_t t = cnt[i]; cnt[i] = sum; sum += t; }
		for (i = 0; i < n; i++) { size_t d = cnt[(key[i] >> shift) & 0xFFFF]++; k2[d] = key[i]; v2[d] = val[i]; }
		{ uint64_t *tk = key; key = k2; k2 = tk; }
		{ uint32_t *tv = val; val = v2; v2 = tv; }
	}
	if (((key_bits + 15) / 16) & 1) {        /* odd number of passes: result sits in the scratch pair */
		memcpy(k2, key, n * 8); memcpy(v2, val, n * 4);
		{ uint64_t *tk = key; key = k2; k2 = tk; }
		{ uint32_t *tv = val; val = v2; v2 = tv; }
	}
	free(k2); free(v2); free(cnt);
}

/* ------------------------------------------------------------------ E1: enumeration
 * Restates EnumerateBifurcationsSArrayInRAM (src/vertexenumeration.cpp:263-364): same
 * superGenome "#c0#c1#..#rc(c0)#rc(c1)#..#" (:269-286); a group = all suffixes sharing
 * their first k characters (:316-328, lcp >= k); prev/next character sets (:318-326);
 * Bifurcation() = more than one char or a '#' (:67-70); candidates / terminal (:334-347);
 * ids handed out in suffix-array order of the groups = lexicographic order of the k-mers
 * (:348-355); instance lists sorted by (chr,pos) (:361-362).
 * Grouping is done by sorting, not by a suffix array. */
static void enumerate(orc_ctx *c, uint8_t **data, uint32_t k)
{
	size_t L = 0, n, i, nv = 0, pad = k;
	uint32_t chr, nchr = c->nchr, s;
	uint8_t *S;                       /* 0 = '#', 1..4 = A C G T */
	size_t *cum;                      /* start offset in S of each of the 2*nchr strings */
	uint32_t *run;                    /* number of consecutive non-# symbols starting at i (capped) */
	uint64_t *key; uint32_t *occ; uint32_t *idat;
	static const uint8_t sym[256] = { ['A'] = 1, ['C'] = 2, ['G'] = 3, ['T'] = 4 };

	for (chr = 0; chr < nchr; chr++) L += c->len[chr];
	n = 2 * L + 2 * (size_t)nchr + 1;
	S = (uint8_t *)calloc(n + pad + 1, 1);
	cum = (size_t *)xrealloc(0, (2 * (size_t)nchr + 1) * sizeof(size_t));
	i = 1;
	for (chr = 0; chr < nchr; chr++) {
		size_t j;
		cum[chr] = i;
		for (j = 0; j < c->len[chr]; j++) S[i++] = sym[data[chr][j]];
		i++;                                           /* '#' */
	}
	for (chr = 0; chr < nchr; chr++) {                     /* reverse complements, DNASequence::Translate */
		size_t j, ln = c->len[chr];
		cum[nchr + chr] = i;
		for (j = 0; j < ln; j++) S[i++] = (uint8_t)(5 - sym[data[chr][ln - 1 - j]]);
		i++;
	}
	cum[2 * nchr] = n;

	run = (uint32_t *)xrealloc(0, (n + 1) * 4);
	run[n] = 0;
	for (i = n; i-- > 0;) run[i] = S[i] ? (run[i + 1] < k ? run[i + 1] + 1 : k) : 0;
	for (i = 0; i < n; i++) nv += run[i] >= k;

	key = (uint64_t *)xrealloc(0, nv * 8);
	occ = (uint32_t *)xrealloc(0, nv * 4);
	if (k <= 32 && !c->force_long) {
		size_t m = 0;
		uint64_t code = 0, mask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
		uint32_t have = 0;
		for (i = 0; i < n; i++) {
			if (!S[i]) { have = 0; code = 0; continue; }
			code = ((code << 2) | (uint64_t)(S[i] - 1)) & mask;
			if (++have >= k) { key[m] = code; occ[m++] = (uint32_t)(i + 1 - k); }
		}
		radix_sort_kv(key, occ, nv, (int)(2 * k));
	} else {
		/* rank doubling (Karp-Miller-Rosenberg): rank_h[i] = order-preserving dense rank of
		 * S[i..i+h); rank_2h from pairs (rank_h[i], rank_h[i+h]); k-windows from the pair
		 * (rank_h[i], rank_h[i+k-h]) with h the largest power of two <= k. */
		size_t np = n + pad, h = 1, m = 0;
		uint32_t *rk = (uint32_t *)xrealloc(0, (np + 1) * 4), *rk2 = (uint32_t *)xrealloc(0, (np + 1) * 4);
		uint64_t *pk = (uint64_t *)xrealloc(0, np * 8);
		uint32_t *pi = (uint32_t *)xrealloc(0, np * 4);
		for (i = 0; i < np; i++) rk[i] = i < n ? S[i] : 0;
		while (2 * h <= k) {
			uint32_t r = 0;
			for (i = 0; i < np; i++) {
				pk[i] = ((uint64_t)rk[i] << 32) | (i + h < np ? rk[i + h] : 0);
				pi[i] = (uint32_t)i;
			}
			radix_sort_kv(pk, pi, np, 64);
			for (i = 0; i < np; i++) {
				if (i && pk[i] != pk[i - 1]) r++;
				rk2[pi[i]] = r;
			}
			{ uint32_t *t = rk; rk = rk2; rk2 = t; }
			h *= 2;
		}
		for (i = 0; i < n; i++)
			if (run[i] >= k) { key[m] = ((uint64_t)rk[i] << 32) | rk[i + k - h]; occ[m++] = (uint32_t)i; }
		radix_sort_kv(key, occ, nv, 64);
		free(rk); free(rk2); free(pk); free(pi);
	}

	idat = (uint32_t *)xrealloc(0, n * 4);
	memset(idat, 0xFF, n * 4);
	c->bif_count = 0;
	for (i = 0; i < nv;) {
		size_t e = i, j;
		unsigned prev = 0, next = 0;               /* bit 0 = '#', bits 1..4 = A C G T */
		int terminal = 0;
		while (e < nv && key[e] == key[i]) {
			size_t p = occ[e];
			prev |= 1u << S[p - 1];
			next |= 1u << S[p + k];
			terminal |= S[p - 1] == 0 || S[p + k] == 0;
			e++;
		}
		if ((prev & (prev - 1)) || (prev & 1) || (next & (next - 1)) || (next & 1)) {
			if (e - i > 1 || terminal) {
				for (j = i; j < e; j++) idat[occ[j]] = c->bif_count;
				c->bif_count++;
			}
		}
		i = e;
	}
	for (s = 0; s < 2; s++) {
		c->ninst[s] = 0;
		for (chr = 0; chr < nchr; chr++) {
			size_t base = cum[s * nchr + chr], p;
			for (p = 0; p + k <= c->len[chr]; p++)
				if (idat[base + p] != NONE) {
					orc_inst *d;
					GROW(c->inst[s], c->cinst[s], c->ninst[s] + 1, orc_inst);
					d = &c->inst[s][c->ninst[s]++];
					d->id = idat[base + p]; d->chr = chr; d->pos = (uint32_t)p;
				}
		}
	}
	free(S); free(cum); free(run); free(key); free(occ); free(idat);
}

int orc_enumerate(orc_ctx *c, uint32_t k, uint32_t *bif_count,
                  const orc_inst **pos, uint64_t *npos, const orc_inst **neg, uint64_t *nneg)
{
	uint8_t **d;
	if (k < 2) return 1;
	d = sanitised_copy(c);
	enumerate(c, d, k);
	free_copy(c, d);
	if (bif_count) *bif_count = c->bif_count;
	if (pos) *pos = c->inst[0];
	if (npos) *npos = c->ninst[0];
	if (neg) *neg = c->inst[1];
	if (nneg) *nneg = c->ninst[1];
	return 0;
}

/* ------------------------------------------------------------------ strand iterators
 * StrandIterator (src/stranditerator.cpp): an element index plus a direction. */
typedef struct { uint32_t e; int d; } sit;     /* d: 0 positive, 1 negative */

static const char COMP[256] = { ['A'] = 'T', ['C'] = 'G', ['G'] = 'C', ['T'] = 'A', ['$'] = '$' };

static inline sit it_next(const orc_ctx *c, sit a) { a.e = a.d ? c->pv[a.e] : c->nx[a.e]; return a; }   /* operator++ :117-130 */
static inline sit it_adv(const orc_ctx *c, sit a, size_t n) { while (n--) a = it_next(c, a); return a; }
static inline sit it_inv(const orc_ctx *c, sit a)                                                          /* Invert :192-200 */
{ sit r; if (a.d == 0) { r.e = c->pv[a.e]; r.d = 1; } else { r.e = c->nx[a.e]; r.d = 0; } return r; }
static inline char it_chr(const orc_ctx *c, sit a) { return a.d ? COMP[(uint8_t)c->ch[a.e]] : c->ch[a.e]; } /* operator* :202-210 */
static inline int it_valid(const orc_ctx *c, sit a) { return c->ch[a.e] != SEP; }                          /* AtValidPosition :97-100 */
static inline uint32_t get_bif(const orc_ctx *c, sit a) { return c->bif[a.d][a.e]; }                       /* GetBifurcation, bifurcationstorage.cpp:157 */

/* AddPoint, src/bifurcationstorage.cpp:113-126: no-op if marked; else front-insert into list[strand][id] */
static void add_point(orc_ctx *c, sit a, uint32_t id)
{
	uint32_t nd;
	if (c->bif[a.d][a.e] != NONE || id == NONE) return;
	if (c->nn + 1 > c->cn) {
		size_t nc = c->cn ? c->cn * 2 : 1024;
		c->nslot = (uint32_t *)xrealloc(c->nslot, nc * 4);
		c->nnext = (uint32_t *)xrealloc(c->nnext, nc * 4);
		c->ndead = (uint8_t *)xrealloc(c->ndead, nc);
		c->cn = nc;
	}
	nd = (uint32_t)c->nn++;
	c->nslot[nd] = a.e; c->ndead[nd] = 0;
	c->nnext[nd] = c->head[a.d][id]; c->head[a.d][id] = nd;
	c->lsize[a.d][id]++;
	c->bif[a.d][a.e] = id; c->nodeof[a.d][a.e] = nd;
}

/* ErasePoint, src/bifurcationstorage.cpp:144-155: tag the list node, defer physical removal to Cleanup */
static void erase_point(orc_ctx *c, sit a)
{
	uint32_t id = c->bif[a.d][a.e], nd;
	clear_rec *r;
	if (id == NONE) return;
	nd = c->nodeof[a.d][a.e];
	c->bif[a.d][a.e] = NONE;
	c->ndead[nd] = 1;
	GROW(c->toclear, c->ctoclear, c->ntoclear + 1, clear_rec);
	r = &c->toclear[c->ntoclear++];
	r->key = id; r->strand = (uint32_t)a.d; r->node = nd;
}

/* Cleanup, src/bifurcationstorage.cpp:33-41 (dead nodes stay linked but are skipped and no longer counted) */
static void cleanup(orc_ctx *c)
{
	size_t i;
	for (i = 0; i < c->ntoclear; i++) c->lsize[c->toclear[i].strand][c->toclear[i].key]--;
	c->ntoclear = 0;
}

static inline uint32_t count_bif(const orc_ctx *c, uint32_t id) { return c->lsize[0][id] + c->lsize[1][id]; }  /* :71-75 */

/* ------------------------------------------------------------------ index construction
 * DNASequence ctor (src/dnasequence.cpp:75-103) + marking loop (src/indexedsequence.cpp:49-67). */
static void build_index(orc_ctx *c, uint8_t **data, uint32_t k)
{
	size_t L = 0, e;
	uint32_t chr, s;
	free_index(c);
	c->k = k;
	for (chr = 0; chr < c->nchr; chr++) L += c->len[chr];
	c->ne = L + c->nchr + 1;
	c->ce = c->ne + c->ne / 8 + 1024;
	c->ch = (char *)xrealloc(0, c->ce);
	c->op = (uint32_t *)xrealloc(0, c->ce * 4);
	c->nx = (uint32_t *)xrealloc(0, c->ce * 4);
	c->pv = (uint32_t *)xrealloc(0, c->ce * 4);
	c->sep = (uint32_t *)xrealloc(0, ((size_t)c->nchr + 1) * 4);
	for (s = 0; s < 2; s++) {
		c->bif[s] = (uint32_t *)xrealloc(0, c->ce * 4);
		c->nodeof[s] = (uint32_t *)xrealloc(0, c->ce * 4);
		memset(c->bif[s], 0xFF, c->ce * 4);
		c->head[s] = (uint32_t *)xrealloc(0, ((size_t)c->bif_count + 1) * 4);
		c->lsize[s] = (uint32_t *)calloc((size_t)c->bif_count + 1, 4);
		memset(c->head[s], 0xFF, ((size_t)c->bif_count + 1) * 4);
	}
	e = 0;
	c->ch[e] = SEP; c->op[e] = 0; c->sep[0] = 0; e++;
	for (chr = 0; chr < c->nchr; chr++) {
		size_t j;
		for (j = 0; j < c->len[chr]; j++, e++) { c->ch[e] = (char)data[chr][j]; c->op[e] = c->opos[chr][j] & POS_MASK; }
		c->ch[e] = SEP; c->op[e] = (uint32_t)c->len[chr] & POS_MASK;      /* dnasequence.cpp:96 */
		c->sep[chr + 1] = (uint32_t)e; e++;
	}
	for (e = 0; e < c->ne; e++) { c->nx[e] = (uint32_t)(e + 1); c->pv[e] = (uint32_t)(e - 1); }
	c->nx[c->ne - 1] = NONE; c->pv[0] = NONE;
	/* marking: strand 0 then 1, (chr,pos) ascending, AddPoint => front insertion */
	for (s = 0; s < 2; s++) {
		size_t i;
		for (i = 0; i < c->ninst[s]; i++) {
			const orc_inst *b = &c->inst[s][i];
			sit a;
			a.d = (int)s;
			a.e = s == 0 ? c->sep[b->chr] + 1 + b->pos : c->sep[b->chr + 1] - 1 - b->pos;
			add_point(c, a, b->id);
		}
	}
}

/* ------------------------------------------------------------------ Boost 1.54 unordered_map order
 * boost/unordered/detail/{buckets.hpp:603-654, unique.hpp:302-331,591-619, table.hpp:321-338,808-824}
 * (vendored under the reference's src/include/boost).  Restated: identity hash + mix64,
 * power-of-two bucket counts starting at 16, max load factor 1.0, all nodes on one singly
 * linked list; a node entering an empty bucket goes to the list head, otherwise right after
 * its bucket's predecessor node; rehash re-places nodes walking the list. */
typedef struct {
	uint64_t *key, *hash; int32_t *next;     /* node arrays; node index -1 = none */
	int32_t *bprev;                          /* per bucket: node before the bucket's first node; -2 = list head sentinel; -1 = empty */
	size_t size, cap, bc, max_load;
	int32_t first;                           /* sentinel.next */
} bmap;

static uint64_t mix64(uint64_t key)
{
	key = (~key) + (key << 21);
	key = key ^ (key >> 24);
	key = (key + (key << 3)) + (key << 8);
	key = key ^ (key >> 14);
	key = (key + (key << 2)) + (key << 4);
	key = key ^ (key >> 28);
	key = key + (key << 31);
	return key;
}
static size_t new_bucket_count(size_t m)
{
	if (m <= 4) return 4;
	--m; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16; m |= m >> 32;
	return m + 1;
}
static void bmap_init(bmap *m) { memset(m, 0, sizeof(*m)); m->first = -1; }
static void bmap_free(bmap *m) { free(m->key); free(m->hash); free(m->next); free(m->bprev); }
static int32_t *bm_nextp(bmap *m, int32_t link) { return link == -2 ? &m->first : &m->next[link]; }
static void bmap_create_buckets(bmap *m, size_t bc)
{
	size_t i;
	m->bprev = (int32_t *)xrealloc(m->bprev, bc * sizeof(int32_t));
	for (i = 0; i < bc; i++) m->bprev[i] = -1;
	m->bc = bc; m->max_load = bc;            /* ceil(1.0 * bucket_count) */
}
static void bmap_rehash(bmap *m, size_t bc)
{
	int32_t prev = -2;
	bmap_create_buckets(m, bc);
	while (*bm_nextp(m, prev) != -1) {       /* place_in_bucket */
		int32_t n = *bm_nextp(m, prev);
		size_t b = m->hash[n] & (bc - 1);
		if (m->bprev[b] == -1) { m->bprev[b] = prev; prev = n; }
		else {
			*bm_nextp(m, prev) = m->next[n];
			m->next[n] = *bm_nextp(m, m->bprev[b]);
			*bm_nextp(m, m->bprev[b]) = n;
		}
	}
}
static int32_t bmap_find(const bmap *m, uint64_t key)
{
	size_t b;
	int32_t p;
	if (!m->bprev) return -1;
	b = mix64(key) & (m->bc - 1);
	if (m->bprev[b] == -1) return -1;
	for (p = m->bprev[b] == -2 ? m->first : m->next[m->bprev[b]]; p != -1 && (m->hash[p] & (m->bc - 1)) == b; p = m->next[p])
		if (m->key[p] == key) return p;
	return -1;
}
static int32_t bmap_insert(bmap *m, uint64_t key)   /* operator[] for an absent key, unique.hpp:335-354 */
{
	int32_t n;
	size_t b, need = m->size + 1;
	uint64_t h = mix64(key);
	if (!m->bprev) {                                  /* reserve_for_insert, table.hpp:808-824 */
		size_t mb = new_bucket_count(need + 1);
		bmap_create_buckets(m, mb > 16 ? mb : 16);
	} else if (need > m->max_load) {
		size_t want = need > m->size + (m->size >> 1) ? need : m->size + (m->size >> 1);
		size_t nb = new_bucket_count(want + 1);
		if (nb != m->bc) bmap_rehash(m, nb);
	}
	if (m->size + 1 > m->cap) {
		size_t nc = m->cap ? m->cap * 2 : 64;
		m->key = (uint64_t *)xrealloc(m->key, nc * 8);
		m->hash = (uint64_t *)xrealloc(m->hash, nc * 8);
		m->next = (int32_t *)xrealloc(m->next, nc * 4);
		m->cap = nc;
	}
	n = (int32_t)m->size++;
	m->key[n] = key; m->hash[n] = h;
	b = h & (m->bc - 1);
	if (m->bprev[b] == -1) {                          /* add_node, unique.hpp:302-331 */
		if (m->first != -1) m->bprev[m->hash[m->first] & (m->bc - 1)] = n;
		m->bprev[b] = -2;
		m->next[n] = m->first;
		m->first = n;
	} else {
		m->next[n] = *bm_nextp(m, m->bprev[b]);
		*bm_nextp(m, m->bprev[b]) = n;
	}
	return n;
}

size_t orc_boost_order(const uint64_t *keys, size_t n, uint64_t *out)
{
	bmap m; size_t i, o = 0; int32_t p;
	bmap_init(&m);
	for (i = 0; i < n; i++) if (bmap_find(&m, keys[i]) < 0) bmap_insert(&m, keys[i]);
	for (p = m.first; p != -1; p = m.next[p]) out[o++] = m.key[p];
	bmap_free(&m);
	return o;
}

/* ------------------------------------------------------------------ bulge removal */
typedef struct { uint32_t node; int d; } proxy;          /* IteratorProxy, bifurcationstorage.h:45-58 */
typedef struct { uint32_t bif, dist; } bmark;            /* BifurcationMark, bulgeremoval.cpp:20-37 */
typedef struct { size_t kmer, dist; } vdata;             /* VisitData, blockfinder.h:17-23 */
typedef struct { uint32_t *v; size_t n, c; } u32vec;

static int cmp_mark(const void *a, const void *b)
{
	const bmark *x = (const bmark *)a, *y = (const bmark *)b;
	if (x->bif != y->bif) return x->bif < y->bif ? -1 : 1;
	return x->dist < y->dist ? -1 : x->dist > y->dist;
}
static int cmp_u32(const void *a, const void *b)
{ uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b; return x < y ? -1 : x > y; }

static inline sit deref(const orc_ctx *c, proxy p) { sit a; a.e = c->nslot[p.node]; a.d = p.d; return a; }
static inline int pvalid(const orc_ctx *c, proxy p) { return !c->ndead[p.node]; }   /* IteratorProxy::Valid :23-26 */

/* FillVisit, src/bulgeremoval.cpp:122-146 */
static size_t fill_visit(const orc_ctx *c, sit kmer, size_t D, bmark **visit, size_t *cap)
{
	size_t n = 0, step;
	uint32_t start = get_bif(c, kmer);
	kmer = it_next(c, kmer);
	for (step = 1; step < D && it_valid(c, kmer); kmer = it_next(c, kmer), step++) {
		uint32_t b = get_bif(c, kmer);
		if (b == start) break;
		if (b != NONE) { GROW(*visit, *cap, n + 1, bmark); (*visit)[n].bif = b; (*visit)[n].dist = (uint32_t)step; n++; }
	}
	qsort(*visit, n, sizeof(bmark), cmp_mark);
	return n;
}

/* Overlap, src/bulgeremoval.cpp:97-120 */
static int overlap(const orc_ctx *c, uint32_t k, sit si, size_t di, sit sj, size_t dj)
{
	size_t n = di + k, i;
	uint32_t *occ = (uint32_t *)xrealloc(0, n * 4);
	int r = 0;
	for (i = 0; i < n; i++, si = it_next(c, si)) occ[i] = si.e;
	qsort(occ, n, 4, cmp_u32);
	for (i = 0; i < dj + k; i++, sj = it_next(c, sj))
		if (bsearch(&sj.e, occ, n, 4, cmp_u32)) { r = 1; break; }
	free(occ);
	return r;
}

/* MaxBifurcationMultiplicity, src/bulgeremoval.cpp:39-53 */
static size_t max_mult(const orc_ctx *c, sit a, size_t distance)
{
	size_t r = 0, i;
	for (i = 0; i + 1 < distance; i++) {
		uint32_t b;
		a = it_next(c, a);
		b = get_bif(c, a);
		if (b != NONE && count_bif(c, b) > r) r = count_bif(c, b);
	}
	return r;
}

static uint32_t new_elem(orc_ctx *c)
{
	if (c->ne + 1 > c->ce) {
		size_t nc = c->ce * 2; int s;
		c->ch = (char *)xrealloc(c->ch, nc);
		c->op = (uint32_t *)xrealloc(c->op, nc * 4);
		c->nx = (uint32_t *)xrealloc(c->nx, nc * 4);
		c->pv = (uint32_t *)xrealloc(c->pv, nc * 4);
		for (s = 0; s < 2; s++) {
			c->bif[s] = (uint32_t *)xrealloc(c->bif[s], nc * 4);
			c->nodeof[s] = (uint32_t *)xrealloc(c->nodeof[s], nc * 4);
			memset(c->bif[s] + c->ce, 0xFF, (nc - c->ce) * 4);
		}
		c->ce = nc;
	}
	return (uint32_t)c->ne++;
}

/* DNASequence::ReplaceDirect, src/dnasequence.cpp:189-230.  `target` is the first element of the
 * old span in + direction; `source` yields the new characters already oriented for + storage. */
static void replace_direct(orc_ctx *c, sit source, size_t dS, uint32_t target, size_t dT)
{
	uint32_t save = target, e;
	size_t i, common = dS < dT ? dS : dT;
	uint32_t firstPos = c->op[save] & POS_MASK, lastPos;
	double acc, ssize;
	e = target;
	for (i = 0; i < dT; i++) e = c->nx[e];
	lastPos = c->op[e] & POS_MASK;
	for (i = 0; i < common; i++) { c->ch[target] = it_chr(c, source); target = c->nx[target]; source = it_next(c, source); }
	if (dS < dT) {                                   /* erase the surplus [target, target + dT - dS) */
		uint32_t before = c->pv[target], cur = target;
		for (i = 0; i < dT - dS; i++) cur = c->nx[cur];
		c->nx[before] = cur; c->pv[cur] = before;
	} else if (dS > dT) {                            /* insert the deficit before `target` */
		size_t m = dS - dT;
		char *buf = (char *)xrealloc(0, m);
		uint32_t before;
		for (i = 0; i < m; i++, source = it_next(c, source)) buf[i] = it_chr(c, source);
		before = c->pv[target];
		for (i = 0; i < m; i++) {
			uint32_t ne = new_elem(c);
			c->ch[ne] = buf[i]; c->op[ne] = 0;
			c->bif[0][ne] = c->bif[1][ne] = NONE;
			c->nx[before] = ne; c->pv[ne] = before;
			before = ne;
		}
		c->nx[before] = target; c->pv[target] = before;
		free(buf);
	}
	acc = (double)firstPos;                          /* :221-227, evaluated in the same operation order */
	ssize = (double)dT / (double)dS;
	for (i = 0; i < dS; i++, save = c->nx[save], acc += ssize) {
		size_t p = (size_t)acc;
		if (p > lastPos) p = lastPos;
		c->op[save] = (uint32_t)p & POS_MASK;
	}
}

/* CollapseBulgeGreedily = EraseBifurcations + Replace + UpdateBifurcations
 * (src/bulgeremoval.cpp:284-327, :55-95, src/dnasequence.cpp:232-252, src/bulgeremoval.cpp:238-282) */
/* analysis hook (tools/dependency_depth.py): ORC_TRACE=<file> logs, per RemoveBulges call that has bulge groups, the instances it
 * starts from ("T id n" + n x "I slot strand"), the members of its bulge groups ("M instance-index") and every collapse
 * ("C target-slot strand dT dS"); slots = element indices */
static FILE *orc_trace;
static void collapse(orc_ctx *c, uint32_t k, const proxy *startKMer, vdata src, vdata tgt)
{
	sit t = deref(c, startKMer[tgt.kmer]), s = deref(c, startKMer[src.kmer]);
	if (orc_trace) fprintf(orc_trace, "C %u %d %zu %zu\n", t.e, t.d, tgt.dist, src.dist);
	sit amer, bmer, sa, sb;
	size_t i, nlb = 0, nlf = 0, anear = 0, bnear = 0;
	uint32_t *lb = (uint32_t *)xrealloc(0, (size_t)k * 8), *lf = (uint32_t *)xrealloc(0, (size_t)k * 8);
	/* EraseBifurcations */
	amer = it_inv(c, it_adv(c, t, k));
	bmer = it_adv(c, t, tgt.dist);
	for (i = 0; i < k; i++, amer = it_next(c, amer), bmer = it_next(c, bmer)) {
		uint32_t b = get_bif(c, amer);
		if (b != NONE) { erase_point(c, amer); lb[2 * nlb] = (uint32_t)i; lb[2 * nlb + 1] = b; nlb++; }
		b = get_bif(c, bmer);
		if (b != NONE) { erase_point(c, bmer); lf[2 * nlf] = (uint32_t)i; lf[2 * nlf + 1] = b; nlf++; }
	}
	amer = t;
	bmer = it_inv(c, it_adv(c, t, k + tgt.dist));
	for (i = 0; i < k + tgt.dist; i++, amer = it_next(c, amer), bmer = it_next(c, bmer)) {
		if (i > 0) erase_point(c, amer);
		erase_point(c, bmer);
	}
	/* DNASequence::Replace */
	{
		sit source = it_adv(c, s, k), target = it_adv(c, t, k);
		if (target.d == 0) replace_direct(c, source, src.dist, target.e, tgt.dist);
		else {
			source = it_inv(c, it_adv(c, source, src.dist));
			replace_direct(c, source, src.dist, it_inv(c, it_adv(c, target, tgt.dist)).e, tgt.dist);
		}
	}
	/* UpdateBifurcations */
	amer = it_inv(c, it_adv(c, t, k));
	bmer = it_adv(c, t, src.dist);
	for (i = 0; i < k; i++, amer = it_next(c, amer), bmer = it_next(c, bmer)) {
		if (anear < nlb && i == lb[2 * anear]) { add_point(c, amer, lb[2 * anear + 1]); anear++; }
		if (bnear < nlf && i == lf[2 * bnear]) { add_point(c, bmer, lf[2 * bnear + 1]); bnear++; }
	}
	amer = t;
	bmer = it_inv(c, it_adv(c, t, src.dist + k));
	sa = s;
	sb = it_inv(c, it_adv(c, s, src.dist + k));
	for (i = 0; i < src.dist + 1; i++, amer = it_next(c, amer), bmer = it_next(c, bmer), sa = it_next(c, sa), sb = it_next(c, sb)) {
		uint32_t b = get_bif(c, sa);
		if (b != NONE) add_point(c, amer, b);
		b = get_bif(c, sb);
		if (b != NONE) add_point(c, bmer, b);
	}
	free(lb); free(lf);
}

/* RemoveBulges, src/bulgeremoval.cpp:330-430 (with AnyBulges :158-218 inlined) */
static size_t remove_bulges(orc_ctx *c, uint32_t k, size_t D, uint32_t bifId)
{
	size_t ret = 0, n = 0, cap = 0, i, s, ngroups = 0;
	proxy *startKMer = 0;
	char *endChar;
	bmap visitmap;
	char *vchar = 0; u32vec *vids = 0; size_t vcap = 0;
	bmark *visit = 0; size_t visitcap = 0, nvisit;
	u32vec *groups = 0;
	int32_t p;

	for (s = 0; s < 2; s++) {                       /* ListPositions, bifurcationstorage.h:59-72 */
		uint32_t nd;
		for (nd = c->head[s][bifId]; nd != NONE; nd = c->nnext[nd]) {
			if (c->ndead[nd]) continue;             /* physically removed by an earlier Cleanup */
			GROW(startKMer, cap, n + 1, proxy);
			startKMer[n].node = nd; startKMer[n].d = (int)s; n++;
		}
	}
	if (n < 2) { free(startKMer); return 0; }
	endChar = (char *)xrealloc(0, n);
	for (i = 0; i < n; i++) {                       /* :340-347, ProperKMer dnasequence.h:154-165 */
		sit a = deref(c, startKMer[i]);
		size_t j; int ok = 1;
		for (j = 0; j < (size_t)k + 1; j++, a = it_next(c, a)) if (!it_valid(c, a)) { ok = 0; break; }
		endChar[i] = ok ? it_chr(c, it_adv(c, deref(c, startKMer[i]), k)) : EMPTY_CHAR;
	}
	/* AnyBulges */
	bmap_init(&visitmap);
	for (i = 0; i < n; i++) {
		sit kmer; uint32_t start; size_t step;
		if (endChar[i] == EMPTY_CHAR) continue;
		kmer = deref(c, startKMer[i]);
		start = get_bif(c, kmer);
		kmer = it_next(c, kmer);
		for (step = 1; step < D && it_valid(c, kmer); kmer = it_next(c, kmer), step++) {
			uint32_t b = get_bif(c, kmer);
			int32_t kt;
			if (b == start) break;
			if (b == NONE) continue;
			kt = bmap_find(&visitmap, b);
			if (kt < 0) {
				kt = bmap_insert(&visitmap, b);
				if ((size_t)kt + 1 > vcap) {
					size_t nc = vcap ? vcap * 2 : 64;
					vchar = (char *)xrealloc(vchar, nc);
					vids = (u32vec *)xrealloc(vids, nc * sizeof(u32vec));
					memset(vids + vcap, 0, (nc - vcap) * sizeof(u32vec));
					vcap = nc;
				}
				vchar[kt] = endChar[i];
				vids[kt].n = 0;
				GROW(vids[kt].v, vids[kt].c, 1, uint32_t);
				vids[kt].v[vids[kt].n++] = (uint32_t)i;
			} else if (vchar[kt] != endChar[i]) {
				GROW(vids[kt].v, vids[kt].c, vids[kt].n + 1, uint32_t);
				vids[kt].v[vids[kt].n++] = (uint32_t)i;
				break;
			}
		}
	}
	for (p = visitmap.first; p != -1; p = visitmap.next[p]) ngroups += vids[p].n > 1;
	if (ngroups) {
		size_t g = 0;
		groups = (u32vec *)calloc(ngroups, sizeof(u32vec));
		for (p = visitmap.first; p != -1; p = visitmap.next[p])
			if (vids[p].n > 1) groups[g++] = vids[p];        /* unordered_map iteration order, :203-215 */
	}
	if (!ngroups) goto done;
	if (orc_trace) {
		fprintf(orc_trace, "T %u %zu\n", bifId, n);
		for (i = 0; i < n; i++) { sit a = deref(c, startKMer[i]); fprintf(orc_trace, "I %u %d\n", a.e, a.d); }
		for (s = 0; s < ngroups; s++)                       /* "M index": instance `index` is a member of a bulge group (only members are ever source or target) */
			for (i = 0; i < groups[s].n; i++) fprintf(orc_trace, "M %u\n", groups[s].v[i]);
	}

	for (s = 0; s < ngroups; s++) {
		size_t idI, idJ;
		for (idI = 0; idI < groups[s].n; idI++) {
			size_t kmerI = groups[s].v[idI];
			if (!pvalid(c, startKMer[kmerI])) continue;
			nvisit = fill_visit(c, deref(c, startKMer[kmerI]), D, &visit, &visitcap);
			for (idJ = idI + 1; idJ < groups[s].n; idJ++) {
				size_t kmerJ = groups[s].v[idJ], step;
				sit kmer;
				if (!pvalid(c, startKMer[kmerJ]) || endChar[kmerI] == endChar[kmerJ]) continue;
				kmer = it_next(c, deref(c, startKMer[kmerJ]));
				for (step = 1; it_valid(c, kmer) && step < D; kmer = it_next(c, kmer), step++) {
					uint32_t nowBif = get_bif(c, kmer);
					size_t lo, hi;
					if (nowBif == NONE) continue;
					if (nowBif == bifId) break;
					lo = 0; hi = nvisit;                             /* lower_bound(BifurcationMark(nowBif, 0)) */
					while (lo < hi) { size_t mid = (lo + hi) / 2; if (visit[mid].bif < nowBif) lo = mid + 1; else hi = mid; }
					if (lo < nvisit && visit[lo].bif == nowBif) {
						vdata jdata, idata;
						size_t imlp, jmlp;
						jdata.kmer = kmerJ; jdata.dist = step;
						idata.kmer = kmerI; idata.dist = visit[lo].dist;
						if (overlap(c, k, deref(c, startKMer[kmerI]), idata.dist, deref(c, startKMer[kmerJ]), jdata.dist)) break;
						++ret;
						imlp = max_mult(c, deref(c, startKMer[kmerI]), idata.dist);
						jmlp = max_mult(c, deref(c, startKMer[kmerJ]), jdata.dist);
						if (imlp > jmlp || (imlp == jmlp && idata.kmer < jdata.kmer)) {
							endChar[jdata.kmer] = endChar[idata.kmer];
							collapse(c, k, startKMer, idata, jdata);
						} else {
							endChar[idata.kmer] = endChar[jdata.kmer];
							collapse(c, k, startKMer, jdata, idata);
							nvisit = fill_visit(c, deref(c, startKMer[kmerI]), D, &visit, &visitcap);
						}
						break;
					}
				}
			}
		}
	}
	cleanup(c);
done:
	for (i = 0; i < vcap; i++) free(vids[i].v);
	free(vids); free(vchar); free(groups); free(visit);
	bmap_free(&visitmap);
	free(startKMer); free(endChar);
	return ret;
}

/* BlockFinder::SimplifyGraph, src/blockfinder.cpp:16-51 */
static uint64_t simplify_graph(orc_ctx *c, uint32_t k, uint32_t D, uint32_t max_iter)
{
	uint64_t total = 0;
	uint32_t iterations = 0, id;
	const char *tr = getenv("ORC_TRACE");
	orc_trace = tr ? fopen(tr, "w") : 0;
	do {
		iterations++;
		if (orc_trace) fprintf(orc_trace, "ITER %u\n", iterations);
		for (id = 0; id <= c->bif_count; id++) {
			total += remove_bulges(c, k, D, id);
			if (id == 0xFFFFFFFFu) break;
		}
	} while (total > 0 && iterations < max_iter);
	if (orc_trace) { fclose(orc_trace); orc_trace = 0; }
	return total;
}

int orc_simplify_stage(orc_ctx *c, uint32_t k, uint32_t min_branch, uint32_t max_iter, uint64_t *bulges)
{
	uint8_t **data;
	uint64_t total;
	uint32_t chr;
	double t0, t1, t2, t3;
	if (k < 2) return 1;
	t0 = now_s();
	data = sanitised_copy(c);
	enumerate(c, data, k);
	build_index(c, data, k);
	free_copy(c, data);
	t1 = now_s();
	total = simplify_graph(c, k, min_branch, max_iter);
	t2 = now_s();
	for (chr = 0; chr < c->nchr; chr++) {           /* copy-back, src/blockfinder.cpp:85-95 */
		size_t n = 0, cs = 0, cp = 0;
		uint32_t e;
		free(c->seq[chr]); free(c->opos[chr]);
		c->seq[chr] = 0; c->opos[chr] = 0;
		for (e = c->nx[c->sep[chr]]; e != c->sep[chr + 1]; e = c->nx[e]) {
			GROW(c->seq[chr], cs, n + 1, uint8_t);
			GROW(c->opos[chr], cp, n + 1, uint32_t);
			c->seq[chr][n] = (uint8_t)c->ch[e];
			c->opos[chr][n] = c->op[e] & POS_MASK;
			n++;
		}
		c->len[chr] = n;
	}
	free_index(c);
	t3 = now_s();
	c->t_enum = t1 - t0; c->t_simp = t2 - t1; c->t_copy = t3 - t2;
	if (bulges) *bulges = total;
	return 0;
}

/* ------------------------------------------------------------------ ListEdges, src/serialization.cpp:56-86 */
int orc_list_edges(orc_ctx *c, uint32_t k, const orc_edge **edges, uint64_t *nout)
{
	uint8_t **data;
	uint32_t strand, chr;
	if (k < 2) return 1;
	data = sanitised_copy(c);
	enumerate(c, data, k);
	build_index(c, data, k);
	free_copy(c, data);
	c->nedge = 0;
	for (strand = 0; strand < 2; strand++) {
		for (chr = 0; chr < c->nchr; chr++) {
			size_t pos = 0, length = c->len[chr];
			sit start, end;
			uint32_t prevVertex;
			if (strand == 0) { start.e = c->nx[c->sep[chr]]; start.d = 0; end.e = c->sep[chr + 1]; end.d = 0; }
			else { start.e = c->pv[c->sep[chr + 1]]; start.d = 1; end.e = c->sep[chr]; end.d = 1; }
			if (start.e == end.e) continue;            /* empty chromosome */
			prevVertex = get_bif(c, start);
			while (start.e != end.e) {
				size_t step = 1;
				sit origin = start;
				for (start = it_next(c, start); start.e != end.e && get_bif(c, start) == NONE; start = it_next(c, start)) ++step;
				if (start.e != end.e) {
					orc_edge *ed;
					sit it2 = it_adv(c, start, k);
					uint32_t p1 = c->op[origin.e] & POS_MASK, p2;
					uint32_t lo, hi;
					it2.e = it2.d ? c->nx[it2.e] : c->pv[it2.e];      /* --it2 (SpellOriginal, dnasequence.cpp:254-260) */
					p2 = c->op[it2.e] & POS_MASK;
					lo = p1 < p2 ? p1 : p2; hi = p1 < p2 ? p2 : p1;
					GROW(c->edge, c->cedge, c->nedge + 1, orc_edge);
					ed = &c->edge[c->nedge++];
					ed->chr = chr; ed->strand = strand;
					ed->start_vertex = prevVertex; ed->end_vertex = get_bif(c, start);
					ed->pos = (uint32_t)(strand == 0 ? pos : length - (pos + step + k));
					ed->len = (uint32_t)(step + k);
					ed->orig_pos = lo; ed->orig_len = hi + 1 - lo;
					ed->first_char = it_chr(c, it_adv(c, origin, k));
					prevVertex = ed->end_vertex;
					pos += step;
				}
			}
		}
	}
	free_index(c);
	if (edges) *edges = c->edge;
	if (nout) *nout = c->nedge;
	return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * H0: the k-mer hash of the reference's hashing.h (SlidingWindow / KMerHashFunction, src/hashing.h:14-112), which
 * the production path never executes: H(w) = sum w[i] * 57^(k-1-i) mod 2^64 over the characters a
 * StrandIterator yields (negative strand: DNASequence::Translate of the character, src/dnasequence.cpp:10-28,41-44).
 * Computed here the way Move() does (hashing.h:52-66): value = (value - first * 57^(k-1)) * 57 + next.
 * out: for strand 0 then 1, chromosomes ascending: len - k + 1 values in walk order (none if len < k); malloc'd. */
static unsigned char orc_translate(unsigned char c)
{
	switch (c) { case 'a': return 't'; case 't': return 'a'; case 'g': return 'c'; case 'c': return 'g';
	             case 'A': return 'T'; case 'T': return 'A'; case 'G': return 'C'; case 'C': return 'G'; default: return c; }
}
int orc_kmer_hashes(orc_ctx *c, uint32_t k, uint64_t **out, uint64_t *nout)
{
	uint64_t total = 0, at = 0, high = 1;
	if (k < 1) return 1;
	for (uint32_t ch = 0; ch < c->nchr; ch++) if (c->len[ch] >= k) total += 2 * (c->len[ch] - k + 1);
	for (uint32_t i = 1; i < k; i++) high *= 57;
	uint64_t *v = (uint64_t *)malloc((total ? total : 1) * sizeof *v);
	if (!v) return 2;
	for (int strand = 0; strand < 2; strand++)
		for (uint32_t ch = 0; ch < c->nchr; ch++) {
			uint64_t n = c->len[ch];
			if (n < k) continue;
#define ORC_AT(p) ((uint64_t)(signed char)(strand ? orc_translate(c->seq[ch][n - 1 - (p)]) : c->seq[ch][(p)]))
			uint64_t h = 0;
			for (uint32_t i = 0; i < k; i++) h = h * 57 + ORC_AT(i);          /* CalcKMerHash, hashing.h:73-88 */
			v[at++] = h;
			for (uint64_t p = 1; p + k <= n; p++) {
				h = (h - ORC_AT(p - 1) * high) * 57 + ORC_AT(p + k - 1);  /* Move() */
				v[at++] = h;
			}
#undef ORC_AT
		}
	*out = v; *nout = total;
	return 0;
}
void orc_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------------------------------
 * BlockFinder::SerializeGraph (src/serialization.cpp:112-138 with OutputEdge :15-24): the UNcondensed de Bruijn graph of the
 * current rawSeq_ as DOT text -- per chromosome, strand 0 then 1, one line per (k+1)-window in walk order:
 *   <k-mer> -> <next k-mer> [color="blue|red", label="(chr, pos)"];
 * characters as a StrandIterator yields them (no sanitising; negative strand complemented).  malloc'd, orc_free. */
int orc_serialize_graph(orc_ctx *c, uint32_t k, char **out, uint64_t *nout)
{
	size_t cap = 1 << 16, n = 0;
	char *t = (char *)malloc(cap);
	char buf[256];
#define ORC_PUT(ptr, len) do { size_t l_ = (len); while (n + l_ + 1 > cap) { cap *= 2; t = (char *)realloc(t, cap); } memcpy(t + n, (ptr), l_); n += l_; } while (0)
	ORC_PUT("digraph G\n{\nrankdir=LR\n", strlen("digraph G\n{\nrankdir=LR\n"));
	for (uint32_t ch = 0; ch < c->nchr; ch++)
		for (int strand = 0; strand < 2; strand++) {
			uint64_t len = c->len[ch];
			for (uint64_t pos = 0; pos + k + 1 <= len; pos++) {
				for (int part = 0; part < 2; part++) {
					for (uint32_t i = 0; i < k; i++) {
						uint64_t p = pos + part + i;
						char x = (char)(strand ? orc_translate(c->seq[ch][len - 1 - p]) : c->seq[ch][p]);
						ORC_PUT(&x, 1);
					}
					if (part == 0) ORC_PUT(" -> ", 4);
				}
				int l = snprintf(buf, sizeof buf, " [color=\"%s\", label=\"(%i, %i)\"];\n", strand ? "red" : "blue", (int)ch, (int)pos);
				ORC_PUT(buf, (size_t)l);
			}
		}
	ORC_PUT("}\n", 2);
#undef ORC_PUT
	t[n] = 0;
	*out = t; *nout = n;
	return 0;
}
