// sbl_common.h -- host-side plumbing shared by the C-ABI implementation files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/sibelia_amd.h"

#define SBL_NONE 0xFFFFFFFFu

struct SblError {
	sbl_status st;
	std::string msg;
};

#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
	char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
	throw SblError{e_ == hipErrorOutOfMemory ? SBL_ERR_OOM : SBL_ERR_HIP, b_}; } } while (0)

#define SBL_CHECK(cond, status, text) do { if (!(cond)) throw SblError{(status), (text)}; } while (0)

// Bytes held by all DevBufs of the process, and the test cap on them (SBL_TEST_ALLOC_LIMIT_MB: an allocation that would take the total
// beyond it fails exactly like a hipMalloc that finds no memory -- the out-of-memory paths can be exercised on a 288-GB device).
inline std::atomic<size_t> &sbl_devbuf_total() { static std::atomic<size_t> t{0}; return t; }
inline void sbl_devbuf_cap_check(size_t old_cap, size_t want)
{
	const char *e = getenv("SBL_TEST_ALLOC_LIMIT_MB");
	if (!e) return;
	const size_t limit = (size_t)atoll(e) << 20, now = sbl_devbuf_total().load();
	if (now - old_cap + want > limit) {
		char b[160]; snprintf(b, sizeof b, "out of memory: device allocation of %zu bytes refused (%zu held, test cap %zu)", want, now, limit);
		throw SblError{SBL_ERR_OOM, b};
	}
}

// Grow-only device buffer.
struct DevBuf {
	void *p = nullptr;
	size_t cap = 0;
	void ensure(size_t bytes)
	{
		if (bytes <= cap) return;
		size_t want = bytes + bytes / 16 + 256;
		sbl_devbuf_cap_check(cap, want);
		void *q = nullptr;
		// the new allocation first: a failure then leaves the old buffer (which views may still point into) intact ...
		if (hipMalloc(&q, want) != hipSuccess) {
			(void)hipGetLastError();
			// ... unless only releasing the old one makes room (contents are not preserved by ensure() anyway)
			if (p) { (void)hipFree(p); p = nullptr; sbl_devbuf_total() -= cap; cap = 0; }
			HIP_TRY(hipMalloc(&q, want));
		}
		if (p) (void)hipFree(p);
		sbl_devbuf_total() += want - cap;
		p = q; cap = want;
	}
	// grow preserving the first `keep` bytes
	void grow_keep(size_t bytes, size_t keep, hipStream_t s)
	{
		if (bytes <= cap) return;
		void *q = nullptr;
		size_t want = bytes + bytes / 16 + 256;
		sbl_devbuf_cap_check(0, want);                                // (old and new buffer coexist during the copy)
		HIP_TRY(hipMalloc(&q, want));
		if (p && keep) HIP_TRY(hipMemcpyAsync(q, p, keep, hipMemcpyDeviceToDevice, s));
		HIP_TRY(hipStreamSynchronize(s));
		if (p) (void)hipFree(p);
		sbl_devbuf_total() += want - cap;
		p = q; cap = want;
	}
	void release() { if (p) (void)hipFree(p); sbl_devbuf_total() -= cap; p = nullptr; cap = 0; }
	template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// glibc rand() (TYPE_3 additive feedback, r[i] = r[i-3] + r[i-31], srandom(1)): the reference calls the
// unseeded process-global rand() when it replaces non-ACGT characters (reference src/indexedsequence.cpp:31-37).
struct GlibcRand {
	int32_t r[31];
	int f, b;
	GlibcRand() { seed(1); }
	void seed(uint32_t s)
	{
		int32_t word = s ? (int32_t)s : 1;
		r[0] = word;
		for (int i = 1; i < 31; i++) {
			long hi = word / 127773, lo = word % 127773;
			word = (int32_t)(16807 * lo - 2836 * hi);
			if (word < 0) word += 2147483647;
			r[i] = word;
		}
		f = 3; b = 0;
		for (int i = 0; i < 310; i++) next();
	}
	uint32_t next()
	{
		r[f] = (int32_t)((uint32_t)r[f] + (uint32_t)r[b]);
		uint32_t v = ((uint32_t)r[f] >> 1) & 0x7fffffffu;
		f = (f + 1) % 31; b = (b + 1) % 31;
		return v;
	}
};
