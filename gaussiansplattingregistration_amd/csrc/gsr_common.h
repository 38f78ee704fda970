// gsr_common.h -- host-side plumbing shared by the HEM and ICP halves of libgsr_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/gsr_hip.h"

namespace gsr {

// thread-local message behind gsr_last_error()
std::string& last_error();
int32_t fail(int32_t code, const char* fmt, ...);

#define GSR_HIP(expr)                                                                            \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return ::gsr::fail(GSR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                              \
    } while (0)

#define GSR_TRY(expr)                  \
    do {                               \
        int32_t _r = (expr);           \
        if (_r != GSR_OK) return _r;   \
    } while (0)

// Grow-only device buffer: the library owns its workspace and performs no allocation in steady
// state (a second level or ICP call of the same size reuses everything).
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int32_t reserve(size_t bytes) {
        if (bytes <= cap) return GSR_OK;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return fail(GSR_E_HIP, "hipFree: %s", hipGetErrorString(e)); }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return fail(GSR_E_HIP, "hipMalloc(%zu bytes): %s", want, hipGetErrorString(e)); }
        cap = want;
        return GSR_OK;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
    void swap(DevBuf& o) { void* tp = p; p = o.p; o.p = tp; size_t tc = cap; cap = o.cap; o.cap = tc; }
};

// one step of a host spin loop on device-written pinned memory: a pause instruction on x86, a yield elsewhere; after ~20 us of
// spinning (the read-backs normally arrive in ~5 us) the core is handed back between looks
inline void cpu_relax(unsigned spins) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    asm volatile("" ::: "memory");
#endif
    if (spins > 4096u && (spins & 63u) == 0u) sched_yield();
}

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// grid for a grid-stride elementwise kernel: enough blocks to fill 256 CUs x 8, no more
inline int stride_grid(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

// glibc TYPE_3 rand() model (the reference draws parent flags from the never-seeded libc rand(),
// src/cpp_ext/include/base.hpp:44-56).  Product-side implementation; the oracle has its own.
struct GlibcRng {
    uint32_t st[31];
    int f = 3, r = 0;
    void seed(uint32_t s);
    inline uint32_t next() {
        uint32_t v = (st[f] += st[r]);
        if (++f == 31) f = 0;
        if (++r == 31) r = 0;
        return v >> 1;
    }
    inline uint32_t hem_rand() {
        uint32_t x = 0;
        for (int i = 0; i < 8; ++i) x |= (next() & 15u) << (4 * i);
        return x;
    }
};

}  // namespace gsr
