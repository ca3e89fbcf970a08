// How fast can every CU stream the SAME L2-resident buffer (the chained pointwise kernel's pre-split weights: 1.5 MB read by all 256
// blocks in the same order at about the same time)?  Variants: every block walks the buffer from the start (what the kernel does) /
// from a block-dependent offset (blocks of an XCD are never on the same KiB at the same time) / a private copy per XCD / per block.
// 512 threads = 8 waves per block, one block per CU, 16-byte loads (1 KiB per wave-instruction), R loads in flight per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(512, 1) void stream_k(const f4* __restrict__ w, float* out, int kib, int passes, int rotate, long copy_stride_f4, int copy_mode) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // copy_mode 0: one buffer; 1: one copy per XCD (blockIdx & 7); 2: one copy per block
    const f4* base = w + (copy_mode == 0 ? 0 : copy_mode == 1 ? (long)(blockIdx.x & 7) * copy_stride_f4 : (long)blockIdx.x * copy_stride_f4);
    const int per_wave = kib / 8;                       // KiB (= wave-loads) per wave and pass
    const int start = rotate ? (int)(((blockIdx.x >> 3) * 37u) % (unsigned)per_wave) : 0;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < passes; ++p) {
        for (int i = 0; i < per_wave; i += R) {
            f4 v[R];
#pragma unroll
            for (int u = 0; u < R; ++u) {
                int j = i + u + start;
                if (j >= per_wave) j -= per_wave;
                v[u] = base[((long)(wave * per_wave + j)) * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < R; ++u) acc += v[u];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
    const int kib = 1536;                               // 1.5 MB
    const long n_f4 = (long)kib * 64;
    f4* w; float* out;
    hipMalloc(&w, sizeof(f4) * n_f4 * 256);
    hipMalloc(&out, 256 * 512 * 4);
    hipMemset(w, 0, sizeof(f4) * n_f4 * 256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int passes = 20;
    for (int mode = 0; mode < 3; ++mode)
        for (int rot = 0; rot < 2; ++rot) {
            stream_k<8><<<256, 512>>>(w, out, kib, 2, rot, n_f4, mode);
            hipEventRecord(e0);
            stream_k<8><<<256, 512>>>(w, out, kib, passes, rot, n_f4, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double bytes = (double)kib * 1024 * passes;
            printf("%-22s %-26s %6.1f us per 1.5 MB pass   %6.1f GB/s per CU   %5.1f TB/s chip\n", mode == 0 ? "one shared buffer" : mode == 1 ? "one copy per XCD" : "one copy per block",
                   rot ? "block-dependent start" : "every block from the start", ms * 1e3 / passes, bytes / (ms * 1e-3) / 1e9, bytes * 256 / (ms * 1e-3) / 1e12);
        }
    // depth of the request ring
    {
        hipEventRecord(e0);
        stream_k<4><<<256, 512>>>(w, out, kib, passes, 0, n_f4, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("one shared buffer, 4 loads in flight per wave: %6.1f us per pass\n", ms * 1e3 / passes);
        hipEventRecord(e0);
        stream_k<16><<<256, 512>>>(w, out, kib, passes, 0, n_f4, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("one shared buffer, 16 loads in flight per wave: %6.1f us per pass\n", ms * 1e3 / passes);
    }
    return 0;
}
