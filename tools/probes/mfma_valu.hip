// Does VALU work issued between fp32 MFMAs cost matrix-pipe time?  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate:
// if it shares the vector ALUs, every VALU instruction of any wave on the SIMD takes its cycles away from the MFMAs.
// Bare MFMA loop (16 MFMAs per iteration) + NV independent VALU instructions per iteration, 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu tools/probes/mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>
__global__ __launch_bounds__(256) void probe(float* __restrict__ out, int iters) {
    extern __shared__ float lds[];
    const float a0 = 1.0f + threadIdx.x * 1e-3f, b0 = 0.5f - threadIdx.x * 1e-4f;
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.25f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, b0, acc[3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV / 4; ++v) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(a0));
                if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[v & 7]));
                if (KIND == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[v & 7]) : "v"(a0));
            }
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NV, int KIND>
void run(int wg_per_cu, float* out) {
    const int iters = 60000 / wg_per_cu, grid = 256 * wg_per_cu;
    const size_t lds = wg_per_cu == 1 ? 80 * 1024 : 73728;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<NV, KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<NV, KIND>), dim3(grid), dim3(256), lds, 0, out, iters);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flops = (double)grid * 4 * iters * 16.0 * 4096.0;
    const char* kind = KIND == 0 ? "v_add_f32" : KIND == 1 ? "v_exp_f32" : "v_cndmask";
    printf("%2d x %-10s per 16 MFMAs, %d wave/SIMD: %8.3f ms %7.1f TFLOP/s  (%.1f cycles per 16 MFMAs at 2.4 GHz)\n", NV, kind,
           wg_per_cu, best, flops / best / 1e9, best * 1e-3 * 2.4e9 / iters / wg_per_cu);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4096);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(w, out);
        run<16, 0>(w, out);
        run<32, 0>(w, out);
        run<64, 0>(w, out);
        run<16, 1>(w, out);
        run<32, 1>(w, out);
        run<32, 2>(w, out);
    }
    return 0;
}
