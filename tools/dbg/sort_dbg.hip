#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <algorithm>
int main()
{
	for (size_t n : {4096ul, 100000ul, 5000000ul}) for (unsigned bits : {4u, 8u, 16u, 17u, 64u}) {
		std::vector<unsigned long long> hk(n), hv(n);
		unsigned long long x = 88172645463325252ull;
		for (size_t i = 0; i < n; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hk[i] = x; hv[i] = i; }
		unsigned long long *k0, *k1, *v0, *v1; void *dt; size_t tmp = 0;
		hipMalloc(&k0, n * 8); hipMalloc(&k1, n * 8); hipMalloc(&v0, n * 8); hipMalloc(&v1, n * 8);
		hipMemcpy(k0, hk.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(v0, hv.data(), n * 8, hipMemcpyHostToDevice);
		rocprim::radix_sort_pairs(nullptr, tmp, k0, k1, v0, v1, n, 0, bits);
		hipMalloc(&dt, tmp);
		hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
		rocprim::radix_sort_pairs(dt, tmp, k0, k1, v0, v1, n, 0, bits);
		hipEventRecord(a);
		hipError_t e = rocprim::radix_sort_pairs(dt, tmp, k0, k1, v0, v1, n, 0, bits);
		hipEventRecord(b);
		hipDeviceSynchronize();
		float ms; hipEventElapsedTime(&ms, a, b);
		std::vector<unsigned long long> sk(n), sv(n);
		hipMemcpy(sk.data(), k1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(sv.data(), v1, n * 8, hipMemcpyDeviceToHost);
		size_t viol = 0, mism = 0;
		for (size_t i = 0; i < n; i++) { if (i && (bits == 64 ? sk[i] < sk[i - 1] : (sk[i] & ((1ull << bits) - 1)) < (sk[i - 1] & ((1ull << bits) - 1)))) viol++; if (sv[i] >= n || hk[sv[i]] != sk[i]) mism++; }
		printf("n %zu bits %u: err %d tmp %zu order violations %zu pair mismatches %zu  %.3f ms\n", n, bits, (int)e, tmp, viol, mism, ms);
		hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(dt);
	}
}
