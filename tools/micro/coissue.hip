// Does v_mfma_f32_32x32x2_f32 co-issue with VALU / LDS / VMEM work of the same wave, or of a partner wave on
// the same SIMD?  Times 4096 x 16 MFMAs per wave with K extra instructions of one kind per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND, int NX>   // KIND 0 none, 1 VALU fma, 2 ds_read_b128, 3 global_load_dwordx4
__global__ __launch_bounds__(256) void k(float* out, const f4* g, int iters, float a0) {
    __shared__ f4 lds[1024];
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    lds[threadIdx.x] = (f4)(a0); lds[threadIdx.x + 256] = (f4)(a0);
    __syncthreads();
    float a = a0 + threadIdx.x * 1e-6f, b = 1.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = a0 + i;
    f4 t = (f4)(0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int x = 0; x < NX; ++x) {
                if (KIND == 1) v[x & 7] = v[x & 7] * 1.0001f + 0.5f;
                if (KIND == 2) { f4 q = lds[(threadIdx.x + 17 * x + u) & 1023]; t += q; }
                if (KIND == 3) { f4 q = g[((threadIdx.x + 64 * x + 1024 * u) & 65535)]; t += q; }
            }
        }
    }
    float s = t.x + t.y + t.z + t.w;
    for (int r = 0; r < 16; ++r) s += acc[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int NX>
void run(const char* name, int bpc, float* out, f4* g) {
    const int iters = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, NX><<<256 * bpc, 256>>>(out, g, 8, 1.f);
    hipEventRecord(e0);
    k<KIND, NX><<<256 * bpc, 256>>>(out, g, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s x%d blocks/CU=%d : %.1f cycles per MFMA per SIMD (@2.4GHz)\n", name, NX, bpc, ms * 1e-3 * 2.4e9 / (iters * 16.0 * bpc));
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    f4* g; hipMalloc(&g, 65536 * 16); hipMemset(g, 0, 65536 * 16);
    for (int b = 1; b <= 2; ++b) {
        run<0, 0>("none", b, out, g);
        run<1, 4>("valu", b, out, g); run<1, 8>("valu", b, out, g); run<1, 16>("valu", b, out, g);
        run<2, 1>("ds_read", b, out, g); run<2, 2>("ds_read", b, out, g); run<2, 4>("ds_read", b, out, g);
        run<3, 1>("gload", b, out, g); run<3, 2>("gload", b, out, g);
    }
    return 0;
}
