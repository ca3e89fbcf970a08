// Internal helpers shared by the kernels of libdcap_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include "../../include/dcap.h"

namespace dcap {

void set_error(const char* fmt, ...);

// Kernels that use more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per (kernel, device):
// the static below is per expansion site (per template instantiation), one bit per device ordinal, so a host process that drives
// several GPUs sets the attribute on each of them.
#define DC_ENSURE_DYN_LDS(fn, bytes)                                                                                                  \
    do {                                                                                                                              \
        static std::atomic<unsigned long long> dc_done_{0};                                                                           \
        int dc_dev_ = 0;                                                                                                              \
        (void)hipGetDevice(&dc_dev_);                                                                                                 \
        const unsigned long long dc_bit_ = 1ull << (dc_dev_ & 63);                                                                    \
        if (!(dc_done_.load(std::memory_order_acquire) & dc_bit_)) {                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes));        \
            dc_done_.fetch_or(dc_bit_, std::memory_order_release);                                                                    \
        }                                                                                                                             \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DC_ELAUNCH;
    }
    return DC_OK;
}

// An integer tuning knob from the environment.  Call as  static const int v = env_int("NAME", default);  -- a function-local
// static with a dynamic initialiser is initialised exactly once, thread-safely (host threads may call the C-ABI concurrently).
inline int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define DC_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            ::dcap::set_error(__VA_ARGS__);    \
            return (code);                     \
        }                                      \
    } while (0)

constexpr int kNumCU = 256;   // MI355X: 8 XCDs x 32 CUs
constexpr int kNumXCD = 8;

// Bijective XCD-aware remap (blocks b and b+8 share an XCD): consecutive logical ids land on the
// same XCD so tiles that share an operand panel hit that XCD's private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk / kNumXCD, r = nblk % kNumXCD, xcd = bid % kNumXCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / kNumXCD;
}

// Zero `bytes` bytes at `p` (both multiples of 4) with a KERNEL.  Used instead of hipMemsetAsync wherever a launch sequence may be
// captured into a hipGraph: a captured memset becomes a memset NODE, and on this runtime the joint train step replayed as a graph
// faulted on its second launch inside the top-k radix select (whose histograms a memset node was supposed to clear) while the same
// sequence issued eagerly, or with this kernel in the graph, is fine (round 4; tools/diag_joint_graph4.py).
static __global__ __launch_bounds__(256) void zero_fill_kernel(unsigned* __restrict__ p, size_t words) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
inline int zero_fill_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return DC_OK;
    if ((reinterpret_cast<uintptr_t>(p) & 3u) || (bytes & 3u)) {
        set_error("zero_fill: pointer and size must be multiples of 4");
        return DC_EALIGN;
    }
    const size_t words = bytes / 4;
    const int blocks = (int)((words + 1023) / 1024 < 2048 ? (words + 1023) / 1024 : 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, static_cast<unsigned*>(p), words);
    return check_launch("zero_fill_kernel");
}

// Philox-2x32-10 (counter (c0, c1), key): element i of stream (seed, offset) depends on nothing else -- dropout masks (loss.hip) and
// the detection-target shuffle (proposal.hip) draw from it.
__device__ __forceinline__ unsigned philox2x32(unsigned c0, unsigned c1, unsigned key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p = (unsigned long long)0xD256D193u * c0;
        const unsigned hi = (unsigned)(p >> 32), lo = (unsigned)p;
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += 0x9E3779B9u;
    }
    return c0;
}

}  // namespace dcap
