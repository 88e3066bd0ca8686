// micro-benchmark (not part of the product): one dependent "window scan" step (3 x 64 consecutive elements at a random position, next
// position dependent on what was read) with the element fields in separate arrays (SoA, the product's layout) against a blocked layout
// (the fields of 64 consecutive elements in one 1344-byte block: one page, one DRAM row), at two working-set sizes.
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ unsigned mixu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
#define BLK_WORDS 336u      // 64 B ch + 5 x 256 B (nx, bif0, bif1, rmax, wmax) = 1344 B
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
			s += ch[e] + bif0[e] + bif1[e] + nx[e] + wmax[e];
			if (atom) atomicMax(&rmax[e], 1u);
		}
		for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
		acc += s;
		pos = mixu(pos + s + it) % (n - 1024);
	}
	unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}
__global__ void __launch_bounds__(64) k_blk(unsigned *buf, unsigned n, int iters, int bursts, unsigned long long *out, int atom)
{
	unsigned lane = threadIdx.x, pos = mixu(blockIdx.x * 977u + 13u) % (n - 1024);
	unsigned acc = 0;
	unsigned long long t0 = __builtin_readcyclecounter();
	for (int it = 0; it < iters; it++) {
		unsigned s = 0;
		for (int b = 0; b < bursts; b++) {
			unsigned e = pos + 64 * b + lane;
			unsigned *blk = buf + (size_t)(e >> 6) * BLK_WORDS;
			unsigned l = e & 63u;
			s += reinterpret_cast<const unsigned char *>(blk)[l] + blk[16 + l] + blk[80 + l] + blk[144 + l] + blk[272 + l];
			if (atom) atomicMax(&blk[208 + l], 1u);
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
	const unsigned nmax = 42u << 20;
	unsigned char *ch; unsigned *b0, *b1, *nx, *rm, *wm, *buf; unsigned long long *out;
	hipMalloc(&ch, nmax); hipMalloc(&b0, nmax * 4ull); hipMalloc(&b1, nmax * 4ull); hipMalloc(&nx, nmax * 4ull); hipMalloc(&rm, nmax * 4ull); hipMalloc(&wm, nmax * 4ull);
	hipMalloc(&buf, (size_t)(nmax / 64 + 1) * BLK_WORDS * 4); hipMalloc(&out, 1 << 20);
	hipMemset(ch, 0, nmax); hipMemset(b0, 0, nmax * 4ull); hipMemset(b1, 0, nmax * 4ull); hipMemset(nx, 0, nmax * 4ull); hipMemset(rm, 0, nmax * 4ull); hipMemset(wm, 0, nmax * 4ull);
	hipMemset(buf, 0, (size_t)(nmax / 64 + 1) * BLK_WORDS * 4);
	const int iters = 200;
	for (unsigned n : {42u << 20, 4u << 20, 400u << 10}) for (int waves : {4096, 16384}) for (int atom = 0; atom < 2; atom++) for (int layout = 0; layout < 2; layout++) {
		for (int rep = 0; rep < 2; rep++) {
			if (layout == 0) k_soa<<<waves, 64>>>(ch, b0, b1, nx, rm, wm, n, iters, 3, out, atom);
			else k_blk<<<waves, 64>>>(buf, n, iters, 3, out, atom);
			hipDeviceSynchronize();
		}
		std::vector<unsigned long long> h(waves * 2);
		hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost);
		double sum = 0; for (int i = 0; i < waves; i++) sum += h[2 * i];
		printf("n %9u waves %5d %s %s: %.0f cycles per dependent step\n", n, waves, layout ? "blocked" : "SoA    ", atom ? "5 loads + atomicMax" : "5 loads            ", sum / waves / iters);
	}
}
