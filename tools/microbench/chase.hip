// Dependent-gather latency under a k_commit-like load: W waves (one per workgroup), each chasing `steps` dependent steps; a step is
// `fan` coalesced 256-byte reads (one per "array" = region of the buffer) at a pseudo-random offset derived from the previous step's data.
// usage: chase <span_MB> <waves> <steps> <fan>     prints average core cycles and wall ns per step
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void __launch_bounds__(64) k_fill(unsigned *p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x; for (; i < n; i += st) p[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7); }
__global__ void __launch_bounds__(64) k_chase(const unsigned *p, size_t words, unsigned steps, unsigned fan, unsigned long long *out)
{
	const unsigned lane = threadIdx.x;
	const size_t region = words / fan;
	unsigned long long x = blockIdx.x * 0x9E3779B97F4A7C15ull + 12345u;
	unsigned acc = 0;
	const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
	for (unsigned s = 0; s < steps; s++) {
		x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
		const size_t off = ((x % (region / 64u - 1u)) * 64u);
		unsigned v = 0;
		for (unsigned f = 0; f < fan; f++) v += p[f * region + off + lane];
		v = __shfl(v, 0) + __shfl(v, 63);
		acc += v; x += v;                                             // the next offset depends on the data
	}
	const unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
	if (lane == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
	if (acc == 0x12345u) out[0] = 0;
}
int main(int argc, char **argv)
{
	const size_t mb = argc > 1 ? atol(argv[1]) : 1024; const unsigned waves = argc > 2 ? atoi(argv[2]) : 3000, steps = argc > 3 ? atoi(argv[3]) : 200, fan = argc > 4 ? atoi(argv[4]) : 1;
	const size_t words = mb * 1024 * 1024 / 4;
	unsigned *p; unsigned long long *out;
	CK(hipMalloc(&p, words * 4)); CK(hipMalloc(&out, waves * 16));
	k_fill<<<4096, 64>>>(p, words);
	CK(hipDeviceSynchronize());
	for (int rep = 0; rep < 2; rep++) {
		hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
		CK(hipEventRecord(a));
		k_chase<<<waves, 64>>>(p, words, steps, fan, out);
		CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
		float ms; CK(hipEventElapsedTime(&ms, a, b));
		std::vector<unsigned long long> h(2 * waves);
		CK(hipMemcpy(h.data(), out, waves * 16, hipMemcpyDeviceToHost));
		double c = 0, w = 0; for (unsigned i = 0; i < waves; i++) { c += h[2 * i]; w += h[2 * i + 1]; }
		if (rep) printf("span %zu MB waves %u steps %u fan %u: %.0f cycles/step, %.0f ns/step, kernel %.3f ms\n", mb, waves, steps, fan, c / waves / steps, w / waves / steps * 10.0, ms);
	}
	return 0;
}
