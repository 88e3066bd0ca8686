// micro-benchmark (not part of the product): a probe-shaped window scan -- 3 x 64 consecutive elements at a random position, next
// position dependent on what was read -- with (a) the four arrays the scans read today (ch 1 B, bif 4 B, nx 4 B, wmax 4 B per
// element) and (b) ONE summary byte per element (character code, "links are consecutive" bits, "has a mark" bits, "was written"
// bit) plus a sparse gather of bif for the ~10 % of the elements that carry a mark.  Reports cycles per dependent step and the
// throughput of the whole launch (the probe and the reservation walks are bound by the number of lines they touch).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned mixu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(64) k_scan(const unsigned char *ch, const unsigned *bif, const unsigned *nx, const unsigned *wmax, const unsigned char *meta,
                                             unsigned n, int iters, int mode, unsigned long long *out)
{
	unsigned lane = threadIdx.x, pos = mixu(blockIdx.x * 977u + 13u) % (n - 1024);
	unsigned acc = 0;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
		unsigned s = 0;
		for (int b = 0; b < 3; b++) {
			unsigned e = pos + 64 * b + lane;
			if (mode == 0) s += ch[e] + bif[e] + nx[e] + wmax[e];
			else {
				unsigned m = meta[e];
				s += m;
				if (m & 0x20u) s += bif[e];          // ~10 % of the elements carry a mark
				if (m & 0x80u) s += wmax[e];         // written this iteration (mode 2: ~25 %)
			}
		}
		for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
		acc += s;
		pos = mixu(pos + s + it) % (n - 1024);
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}
__global__ void k_fill(unsigned char *meta, unsigned n, int wfrac)
{
	unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
	if (e >= n) return;
	unsigned h = mixu(e * 2654435761u);
	meta[e] = (unsigned char)((h % 10 == 0 ? 0x20u : 0u) | ((int)((h >> 8) % 100) < wfrac ? 0x80u : 0u) | 0x18u);
}
int main()
{
	const unsigned n = 42u << 20;
	unsigned char *ch, *meta; unsigned *b0, *nx, *wm; unsigned long long *out;
	hipMalloc(&ch, n); hipMalloc(&meta, n); hipMalloc(&b0, n * 4ull); hipMalloc(&nx, n * 4ull); hipMalloc(&wm, n * 4ull); hipMalloc(&out, 1 << 20);
	hipMemset(ch, 0, n); hipMemset(b0, 0, n * 4ull); hipMemset(nx, 0, n * 4ull); hipMemset(wm, 0, n * 4ull);
	const int iters = 100;
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	for (int waves : {4096, 16384, 65536}) for (int mode = 0; mode < 3; mode++) {
		k_fill<<<(n + 255) / 256, 256>>>(meta, n, mode == 2 ? 25 : 0);
		float ms = 0;
		for (int rep = 0; rep < 2; rep++) {
			hipEventRecord(e0);
			k_scan<<<waves, 64>>>(ch, b0, nx, wm, meta, n, iters, mode, out);
			hipEventRecord(e1); hipDeviceSynchronize();
			hipEventElapsedTime(&ms, e0, e1);
		}
		std::vector<unsigned long long> h(waves * 2);
		hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
		double sum = 0; for (int i = 0; i < waves; i++) sum += h[2 * i];
		printf("waves %6d %-46s %7.0f cycles per dependent window, %8.1f M windows/s\n", waves,
		       mode == 0 ? "ch + bif + nx + wmax (13 B / element)" : mode == 1 ? "summary byte + marks (10 %)" : "summary byte + marks (10 %) + stamps (25 %)",
		       sum / waves / iters, (double)waves * iters / (ms * 1e-3) / 1e6);
	}
}
