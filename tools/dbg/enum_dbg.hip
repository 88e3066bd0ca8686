// debugging harness (not part of the product): runs the bucketed enumeration kernels on a tiny input and checks each step on the host
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <vector>
#include <map>
#include <string>
#include "../../sibelia_amd/csrc/kmer_bucket_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
	std::vector<std::string> seqs = {"ACGTTGCAAGGCTTACGGATCCATGACCTGAATCGTTAGC", "ACGTTGCAAGGCTAACGGATCCATGACCTGAATCGTTAGC"};
	unsigned k = 5;
	std::string ch = "$";
	for (auto &s : seqs) ch += s + "$";
	size_t E = ch.size(), Epad = (E + 31) / 32 * 32 + 64;
	ch.resize(Epad, '$');
	size_t nwords = (E + 31) / 32, ntiles = (nwords + KM_TILE_WORDS - 1) / KM_TILE_WORDS, n = ntiles * 4096;
	uint8_t *dch; unsigned long long *pk, *k0, *k1, *v0, *v1; unsigned *sp;
	CK(hipMalloc(&dch, Epad)); CK(hipMalloc(&pk, nwords * 8)); CK(hipMalloc(&sp, nwords * 4));
	CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 8)); CK(hipMalloc(&v1, n * 8));
	CK(hipMemcpy(dch, ch.data(), Epad, hipMemcpyHostToDevice));
	k_pack2bit<<<1, 256>>>(dch, pk, sp, nwords);
	k_kmer_records<<<1, KM_THREADS>>>(pk, sp, nwords, E, k, ntiles, k0, v0);
	CK(hipDeviceSynchronize());
	std::vector<unsigned long long> hk(n), hv(n);
	CK(hipMemcpy(hk.data(), k0, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hv.data(), v0, n * 8, hipMemcpyDeviceToHost));
	auto code = [&](size_t g, bool &ok) { unsigned long long c = 0; ok = true; for (unsigned i = 0; i < k; i++) { char x = ch[g + i]; if (x == '$') { ok = false; return 0ull; } c = c * 4 + (x == 'A' ? 0 : x == 'C' ? 1 : x == 'G' ? 2 : 3); } return c; };
	int bad = 0, nvalid = 0;
	for (size_t g = 0; g < n; g++) {
		bool ok = false; unsigned long long f = g + k <= E ? code(g, ok) : 0; if (g >= E) ok = false;
		if (!ok) { if (hv[g] != KB_INVALID) { if (bad++ < 5) printf("g %zu should be invalid\n", g); } continue; }
		nvalid++;
		unsigned long long r = rc_code(f, k), canon = f < r ? f : r;
		if (hv[g] == KB_INVALID || hk[g] != kmer_hash(canon) || (unsigned)hv[g] != g) { if (bad++ < 5) printf("g %zu record wrong: key %llx want %llx val %llx\n", g, hk[g], kmer_hash(canon), hv[g]); }
	}
	printf("records: %d valid, %d bad\n", nvalid, bad);
	unsigned bits = 4;
	size_t tmp = 0; void *dt = nullptr;
	CK(rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, n, 64 - bits, 64));
	CK(hipMalloc(&dt, tmp));
	CK(rocprim::radix_sort_pairs(dt, tmp, k0, k1, v0, v1, n, 64 - bits, 64));
	CK(hipDeviceSynchronize());
	std::vector<unsigned long long> sk(n), sv(n);
	CK(hipMemcpy(sk.data(), k1, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(sv.data(), v1, n * 8, hipMemcpyDeviceToHost));
	int unsorted = 0; std::map<unsigned long long, int> in, outm;
	for (size_t i = 0; i < n; i++) { in[hk[i] ^ hv[i]]++; outm[sk[i] ^ sv[i]]++; if (i && (sk[i] >> 60) < (sk[i - 1] >> 60)) unsorted++; }
	printf("sort: %d order violations, multiset %s\n", unsorted, in == outm ? "equal" : "DIFFERENT");
	unsigned *boff, *ctr, *pay; unsigned long long *rk, *mem;
	CK(hipMalloc(&boff, 18 * 4)); CK(hipMalloc(&ctr, 64 * 4)); CK(hipMemset(ctr, 0, 64 * 4));
	CK(hipMalloc(&rk, n * 16)); CK(hipMalloc(&pay, n * 8)); CK(hipMalloc(&mem, n * 8));
	k_bucket_bounds<<<1, 256>>>(k1, n, bits, boff);
	k_bucket_classify<<<16, KB_THREADS>>>(k1, v1, boff, k, ctr, rk, pay, (unsigned)n, mem, (unsigned)n);
	CK(hipDeviceSynchronize());
	unsigned hb[17], hc[4];
	CK(hipMemcpy(hb, boff, 17 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hc, ctr, 16, hipMemcpyDeviceToHost));
	printf("boff:"); for (int i = 0; i <= 16; i++) printf(" %u", hb[i]); printf("\ncounters pairs %u keys %u members %u flag %u\n", hc[0], hc[1], hc[2], hc[3]);
	std::vector<unsigned long long> hr(hc[1]);
	CK(hipMemcpy(hr.data(), rk, hc[1] * 8, hipMemcpyDeviceToHost));
	std::map<unsigned long long, int> seen; for (auto x : hr) seen[x]++;
	for (auto &kv : seen) if (kv.second > 1) printf("rank key %llx appears %d times\n", kv.first, kv.second);
	// host truth
	std::map<unsigned long long, unsigned> masks;
	for (size_t g = 0; g < E; g++) if (hv[g] != KB_INVALID) masks[hk[g]] |= (unsigned)(hv[g] >> 32) & 0x1FFF;
	unsigned tp = 0, tk = 0; for (auto &kv : masks) { unsigned m = kv.second, p = m & 0x1F, q = (m >> 8) & 0x1F; bool b = (p & 0x10) || (q & 0x10) || __builtin_popcount(p & 0xF) > 1 || __builtin_popcount(q & 0xF) > 1; if (b) { tp++; unsigned long long c = kmer_unhash(kv.first); tk += rc_code(c, k) == c ? 1 : 2; } }
	printf("host truth: distinct %zu pairs %u keys %u\n", masks.size(), tp, tk);
	return 0;
}
