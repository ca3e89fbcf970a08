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

}  // namespace dcap
