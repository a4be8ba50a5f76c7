// What fp32 MFMA rate does the chip SUSTAIN under a GEMM mainloop's instruction mix?  (development probe)
//   mode 0: bare MFMA loop, operands in registers (pseudo-random values, different per lane and per MFMA)
//   mode 1: + the 128x128x32 tile's LDS fragment reads, software-pipelined one group ahead (4 ds_read_b128 / 16 MFMAs)
//   mode 2: + its global loads (2 x 16 B per lane per 16 MFMAs, streaming, never waited for inside the loop)
//   mode 3: + its LDS staging stores (2 ds_write_b128 per 16 MFMAs) and one workgroup barrier per 64 MFMAs
// DATA 0: the same operand values in every lane and MFMA (low switching activity); 1: pseudo-random.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_ceiling tools/probes/mfma_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

struct Frag { float4 a0, a1, b0, b1; };

template <int MODE>
__device__ __forceinline__ void step(const Frag& cur, Frag& nxt, f32x16 (&acc)[4], const float* arow, int it, float* lds,
                                     const float* gp, int64_t& goff, int64_t gstride, int64_t gn, float4& gv0, float4& gv1) {
    if (MODE >= 1) {
        const int g8 = 8 * ((it + 1) & 3);
        nxt.a0 = *reinterpret_cast<const float4*>(arow + g8);
        nxt.a1 = *reinterpret_cast<const float4*>(arow + 32 * 36 + g8);
        nxt.b0 = *reinterpret_cast<const float4*>(arow + 4608 + g8);
        nxt.b1 = *reinterpret_cast<const float4*>(arow + 4608 + 32 * 36 + g8);
    }
    if (MODE >= 2) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv0) : "v"(gp + goff));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(gv1) : "v"(gp + goff + gstride));
        goff += 2 * gstride;
        if (goff + 2 * gstride + 1024 >= gn) goff = 0;
    }
    __builtin_amdgcn_sched_barrier(0);
    const float a0[4] = {cur.a0.x, cur.a0.y, cur.a0.z, cur.a0.w}, a1[4] = {cur.a1.x, cur.a1.y, cur.a1.z, cur.a1.w};
    const float b0[4] = {cur.b0.x, cur.b0.y, cur.b0.z, cur.b0.w}, b1[4] = {cur.b1.x, cur.b1.y, cur.b1.z, cur.b1.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b0[j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], b1[j], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b0[j], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], b1[j], acc[3], 0, 0, 0);
    }
    if (MODE >= 3) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the loads of three steps ago have landed
        float* w = lds + 9216 + ((threadIdx.x >> 3) + 32 * (it & 3)) * 36 + 4 * (threadIdx.x & 7);
        *reinterpret_cast<float4*>(w) = gv0;
        *reinterpret_cast<float4*>(w + 4608) = gv1;
        if ((it & 3) == 3) __syncthreads();
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int MODE, int DATA>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ g, float* __restrict__ out, int iters, int64_t gn) {
    extern __shared__ float lds[];   // [0, 9216): fragment images; [9216, 18432): staging target
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 18432; i += 256) lds[i] = DATA ? rnd(i * 7919u + blockIdx.x) : 0.001f;
    __syncthreads();
    Frag f0, f1;
    const unsigned t = threadIdx.x * 64;
    f0.a0 = DATA ? make_float4(rnd(t), rnd(t + 1), rnd(t + 2), rnd(t + 3)) : make_float4(1.f, 1.f, 1.f, 1.f);
    f0.a1 = DATA ? make_float4(rnd(t + 4), rnd(t + 5), rnd(t + 6), rnd(t + 7)) : f0.a0;
    f0.b0 = DATA ? make_float4(rnd(t + 8), rnd(t + 9), rnd(t + 10), rnd(t + 11)) : f0.a0;
    f0.b1 = DATA ? make_float4(rnd(t + 12), rnd(t + 13), rnd(t + 14), rnd(t + 15)) : f0.a0;
    f1.a0 = DATA ? make_float4(rnd(t + 16), rnd(t + 17), rnd(t + 18), rnd(t + 19)) : f0.a0;
    f1.a1 = DATA ? make_float4(rnd(t + 20), rnd(t + 21), rnd(t + 22), rnd(t + 23)) : f0.a0;
    f1.b0 = DATA ? make_float4(rnd(t + 24), rnd(t + 25), rnd(t + 26), rnd(t + 27)) : f0.a0;
    f1.b1 = DATA ? make_float4(rnd(t + 28), rnd(t + 29), rnd(t + 30), rnd(t + 31)) : f0.a0;
    const float* arow = lds + (lane & 31) * 36 + 4 * (lane >> 5);
    const float* gp = g + ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t gstride = (int64_t)gridDim.x * 256 * 4;
    int64_t goff = 0;
    float4 gv0 = make_float4(0.f, 0.f, 0.f, 0.f), gv1 = gv0;
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    for (int it = 0; it < iters; it += 2) {
        step<MODE>(f0, f1, acc, arow, it, lds, gp, goff, gstride, gn, gv0, gv1);
        step<MODE>(f1, f0, acc, arow, it + 1, lds, gp, goff, gstride, gn, gv0, gv1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = gv0.x + gv1.y;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE, int DATA>
void run(int wg_per_cu, const float* g, float* out, int64_t gn, int iters) {
    const int cus = 256;
    const size_t lds = wg_per_cu == 1 ? 80 * 1024 : 73728;   // 72 KB: two workgroups per CU
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE, DATA>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = cus * wg_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<MODE, DATA>), dim3(grid), dim3(256), lds, 0, g, out, iters / wg_per_cu, gn);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 /*waves*/ * (iters / wg_per_cu) * 16.0 * (32.0 * 32 * 2 * 2);
        printf("32x32x2 mode %d data %s %d wave/SIMD rep %d: %8.3f ms  %7.1f TFLOP/s\n", MODE, DATA ? "random  " : "constant",
               wg_per_cu, rep, ms, flops / ms / 1e9);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 120000;
    const int64_t gn = (int64_t)1 << 28;  // 1 GiB of floats to stream from
    float *g, *out;
    (void)hipMalloc(&g, gn * 4); (void)hipMalloc(&out, 4096);
    (void)hipMemset(g, 0x3c, gn * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>(w, g, out, gn, iters);
        run<0, 1>(w, g, out, gn, iters);
        run<1, 1>(w, g, out, gn, iters);
        run<2, 1>(w, g, out, gn, iters);
        run<3, 1>(w, g, out, gn, iters);
        run<3, 0>(w, g, out, gn, iters);
    }
    return 0;
}
