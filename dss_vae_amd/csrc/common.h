// Shared helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dssvae_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DSSVAE_CHECK_LAUNCH()                         \
    do {                                              \
        hipError_t e__ = hipGetLastError();           \
        if (e__ != hipSuccess) return (int)e__;       \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// counter-based RNG (splitmix64 finaliser over (seed, index)); uniform in [0,1)
__device__ __forceinline__ float u01(uint64_t seed, uint64_t idx) {
    uint64_t x = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    x = x ^ (x >> 31);
    return (float)(x >> 40) * (1.0f / 16777216.0f);
}

// Bijective XCD-aware remap (blocks b and b+8 share an XCD under round-robin dispatch): each
// XCD gets a contiguous chunk of the logical tile order, so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// Zero fill by a plain kernel launch: hipMemset*Async goes through the runtime's blit path, which costs
// ~8 us of idle stream time per call in front of its fill kernel (rocprofv3 trace, ~95 calls per step).
// rows x cols floats, row stride ld floats.
static __global__ void dssvae_zero2d_kernel(float* __restrict__ p, int64_t ld, int rows, int cols, int vec) {
    const int n4 = (cols + 3) >> 2;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)rows * n4) return;
    const int r = (int)(g / n4), c = 4 * (int)(g % n4);
    float* q = p + (int64_t)r * ld + c;
    if (vec) {
        *reinterpret_cast<float4*>(q) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < cols) q[j] = 0.f;
    }
}

static inline int dssvae_zero2d(float* p, int64_t ld, int rows, int cols, hipStream_t s) {
    if (rows <= 0 || cols <= 0) return DSSVAE_OK;
    const int vec = aligned16(p) && (ld % 4) == 0 && (cols % 4) == 0;
    const int64_t n = (int64_t)rows * ((cols + 3) >> 2);
    hipLaunchKernelGGL(dssvae_zero2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, ld, rows, cols, vec);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

// The same for up to 8 buffers in ONE launch (the split-K outputs of a grouped GEMM: a launch each cost ~4.5 us).
struct DssvaeZeroMany {
    float* p[8]; int64_t ld[8]; int rows[8], cols[8], vec[8], blk_begin[9];
    int n;
};
static __global__ void dssvae_zero_many_kernel(DssvaeZeroMany z) {
    int i = 0;
    while (i + 1 < z.n && (int)blockIdx.x >= z.blk_begin[i + 1]) ++i;
    const int n4 = (z.cols[i] + 3) >> 2;
    const int64_t g = (int64_t)((int)blockIdx.x - z.blk_begin[i]) * blockDim.x + threadIdx.x;
    if (g >= (int64_t)z.rows[i] * n4) return;
    const int r = (int)(g / n4), c = 4 * (int)(g % n4);
    float* q = z.p[i] + (int64_t)r * z.ld[i] + c;
    if (z.vec[i]) {
        *reinterpret_cast<float4*>(q) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c + j < z.cols[i]) q[j] = 0.f;
    }
}
static inline void dssvae_zero_many_add(DssvaeZeroMany& z, float* p, int64_t ld, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return;
    const int i = z.n++;
    z.p[i] = p; z.ld[i] = ld; z.rows[i] = rows; z.cols[i] = cols;
    z.vec[i] = aligned16(p) && (ld % 4) == 0 && (cols % 4) == 0;
    const int64_t n = (int64_t)rows * ((cols + 3) >> 2);
    z.blk_begin[i + 1] = z.blk_begin[i] + (int)((n + 255) / 256);
}
static inline int dssvae_zero_many_launch(const DssvaeZeroMany& z, hipStream_t s) {
    if (z.n == 0) return DSSVAE_OK;
    hipLaunchKernelGGL(dssvae_zero_many_kernel, dim3((unsigned)z.blk_begin[z.n]), dim3(256), 0, s, z);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}
