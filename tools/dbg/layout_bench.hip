// micro-benchmark (not part of the product): latency of a "window scan" step -- 64 consecutive elements at a random position,
// then a dependent next position -- with the element fields in separate arrays (SoA, as the product has them) or in one 32-B record (AoS)
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct alignas(32) Rec { unsigned nx, pv, bif0, bif1, rmax, wmax, op, ch; };
__device__ __forceinline__ unsigned mixu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(64) k_soa(const unsigned char *ch, const unsigned *bif0, const unsigned *bif1, const unsigned *nx, unsigned *rmax, const unsigned *wmax,
                                            unsigned n, int iters, int bursts, unsigned long long *out, int atom)
{
	unsigned lane = threadIdx.x, pos = mixu(blockIdx.x * 977u + 13u) % (n - 1024);
	unsigned acc = 0;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
		unsigned s = 0;
		for (int b = 0; b < bursts; b++) {
			unsigned e = pos + 64 * b + lane;
			s += ch[e] + bif0[e] + nx[e] + (atom == 3 ? 0u : wmax[e]);
			if (atom == 1) atomicMax(&rmax[e], 1u); else if (atom == 2 && (lane & 15) == 0) atomicMax(&rmax[e >> 4], 1u);
		}
		for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
		acc += s;
		pos = mixu(pos + s + it) % (n - 1024);
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}
__global__ void __launch_bounds__(64) k_aos(Rec *rec, unsigned n, int iters, int bursts, unsigned long long *out)
{
	unsigned lane = threadIdx.x, pos = mixu(blockIdx.x * 977u + 13u) % (n - 1024);
	unsigned acc = 0;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
		unsigned s = 0;
		for (int b = 0; b < bursts; b++) {
			unsigned e = pos + 64 * b + lane;
			const uint4 *p = reinterpret_cast<const uint4 *>(&rec[e]);
			uint4 a = p[0], c = p[1];
			s += c.w + a.z + a.x + c.y;
			atomicMax(&rec[e].rmax, 1u);
		}
		for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
		acc += s;
		pos = mixu(pos + s + it) % (n - 1024);
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}
int main()
{
	const unsigned n = 42u << 20;
	unsigned char *ch; unsigned *b0, *b1, *nx, *rm, *wm; Rec *rec; unsigned long long *out;
	hipMalloc(&ch, n); hipMalloc(&b0, n * 4ull); hipMalloc(&b1, n * 4ull); hipMalloc(&nx, n * 4ull); hipMalloc(&rm, n * 4ull); hipMalloc(&wm, n * 4ull);
	hipMalloc(&rec, n * 32ull); hipMalloc(&out, 1 << 20);
	hipMemset(ch, 0, n); hipMemset(b0, 0, n * 4ull); hipMemset(b1, 0, n * 4ull); hipMemset(nx, 0, n * 4ull); hipMemset(rm, 0, n * 4ull); hipMemset(wm, 0, n * 4ull); hipMemset(rec, 0, n * 32ull);
	const int iters = 200;
	for (int waves : {256, 1024, 4096, 16384}) for (int bursts : {1, 3}) {
		for (int mode = 0; mode < 4; mode++) {
			for (int rep = 0; rep < 2; rep++) {
				k_soa<<<waves, 64>>>(ch, b0, b1, nx, rm, wm, n, iters, bursts, out, mode);
				hipDeviceSynchronize();
			}
			std::vector<unsigned long long> h(waves * 2);
			hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
			double sum = 0; for (int i = 0; i < waves; i++) sum += h[2 * i];
			printf("waves %5d bursts %d %s: %.0f cycles per dependent step\n", waves, bursts, mode == 0 ? "loads only (4 arrays)" : mode == 1 ? "loads + atomicMax per element" : mode == 2 ? "loads + atomicMax per 16 elements" : "3 arrays, no stamps", sum / waves / iters);
		}
	}
}
