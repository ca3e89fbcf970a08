// Internal helpers shared by the kernels of libdcap_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dcap.h"

namespace dcap {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return DC_ELAUNCH;
    }
    return DC_OK;
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
