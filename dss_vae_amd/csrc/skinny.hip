// Skinny projections for the generation loop (M = a decode batch of a few dozen rows; the arg-max over a large
// vocabulary at 33+ rows has a kernel of its own further down, skinny_argmax64_kernel):
//   C[M,N] = [X1 | X2] . W^T + bias            (GRU gate pre-activations of one step)
//   argmax_n ([X1 | X2] . W^T + bias)[m, n]    (greedy token: decoder/rnn_decoder.py:70-73 + topk(1))
// At M <= 128 the 128x128 LDS-tiled GEMM wastes most of its tile and runs a handful of workgroups;
// here the problem is what it really is, a stream over W (108 MB for the 30k-vocabulary projection,
// resident in the Infinity Cache between steps): the 32 activation rows of a workgroup sit in LDS for
// the whole kernel, every wave streams 16 rows of W at a time straight from memory into the A operand
// of v_mfma_f32_16x16x4_f32 (exact fp32) with register double-buffered prefetch, and applies them to
// both 16-row activation tiles.  Bound: HBM / Infinity-Cache bandwidth (2 flop per byte of W per tile).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int SK_ROWS = 32;       // activation rows per workgroup (2 MFMA tiles of 16)
constexpr int SK_THREADS = 512;   // 8 waves
constexpr int SK_WAVES = SK_THREADS / 64;
constexpr int SK_GC = 4;          // k-groups (16 k each) per register chunk: gate products (latency-bound chains)
constexpr int SK_GC_STREAM = 16;  // ... and for the vocabulary stream: 16 KB per wave in flight to cover the fabric latency
constexpr int SK_LDS_BUDGET = 150 * 1024;
constexpr float SK_NEG = -3.0e38f;

// row stride of the LDS image: multiple of 16 k plus 4 floats => stride = 4 (mod 8) floats, so the 16 rows
// read by one 16-lane group of a ds_read_b128 fall into 16 distinct 4-bank slots (conflict-free)
inline int padded_k(int K) { return ((K + 15) / 16) * 16 + 4; }

// the 32-row kernels' image + its 16 spare bytes + their static arrays (partial sums of the K split 16 KB, arg-max
// reduction 2 KB) within the 160 KB of a CU
inline bool sk_image_fits(int K) { return SK_ROWS * padded_k(K) * 4 + 16 + 19 * 1024 <= 160 * 1024; }

// The LDS image [ROWS][Kp] of a workgroup's activation rows [X1 | X2], columns k_lo .. k_lo + kvalid of them (zeros
// beyond, and in rows past M).  A thread owns ONE row and every TPR-th 16-byte column of it (TPR = threads / ROWS):
// no division by a run-time value, and no branch around a load -- the addresses are clamped into the row, the values
// masked (with AND: a select on a loaded value came out as a branch, with the load inside it) when they are stored,
// and a column past the row goes to 16 spare bytes behind the image so that the store is unconditional and its load
// cannot sink into a branch either.  IU loads are in flight per thread before the first store; the caller may put
// other work between image_load and image_store.  (The fill this replaces computed row and column of every element
// with a division and guarded every load: ~1500 instructions per thread and pass in the 64-row arg-max kernel.)
template <int ROWS, int IU>
struct RowImage {
    static constexpr int TPR = SK_THREADS / ROWS;
    const float *x1row, *x2row;
    int K1, K, Kp, kp4, ir, ic;
    bool row_ok;
    __device__ __forceinline__ RowImage(const float* X1, int64_t ld1, int K1_, const float* X2, int64_t ld2, int K2,
                                        int m0, int M, int Kp_)
        : K1(K1_), K(K1_ + K2), Kp(Kp_), kp4(Kp_ >> 2), ir(threadIdx.x / TPR), ic(threadIdx.x % TPR) {
        const int m = min(m0 + ir, M - 1);
        row_ok = m0 + ir < M;
        x1row = X1 + (int64_t)m * ld1;
        x2row = K2 > 0 ? X2 + (int64_t)m * ld2 : x1row;
    }
    __device__ __forceinline__ void load(f32x4 (&v)[IU], int k_lo, int c0) const {
#pragma unroll
        for (int u = 0; u < IU; ++u) {
            const int kc = min(k_lo + 4 * (c0 + ic + u * TPR), K - 4);
            v[u] = *reinterpret_cast<const f32x4*>(kc < K1 ? x1row + kc : x2row + (kc - K1));
        }
    }
    __device__ __forceinline__ void store(float* Xs, const f32x4 (&v)[IU], int kvalid, int c0) const {
#pragma unroll
        for (int u = 0; u < IU; ++u) {
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const int c = c0 + ic + u * TPR, kk = 4 * c;
            const unsigned keep = row_ok && kk < kvalid ? 0xffffffffu : 0u;
            *reinterpret_cast<u32x4*>(__builtin_assume_aligned(Xs + (c < kp4 ? ir * Kp + kk : ROWS * Kp), 16)) =
                __builtin_bit_cast(u32x4, v[u]) & (u32x4){keep, keep, keep, keep};
        }
    }
    // the rest of a row wider than TPR * IU columns
    __device__ __forceinline__ void rest(float* Xs, f32x4 (&v)[IU], int k_lo, int kvalid) const {
        for (int c0 = TPR * IU; c0 < kp4; c0 += TPR * IU) {
            load(v, k_lo, c0);
            store(Xs, v, kvalid, c0);
        }
    }
};

// KS = waves that share one 16-column block, each taking a contiguous range of k (partials summed
// through LDS).  The arg-max streams hundreds of blocks per workgroup (throughput-bound: KS = 1); the
// gate pre-activation products have ~100 blocks in all and are latency-bound: KS = 4 shortens the
// dependent chain of memory round trips per wave.
template <bool ARGMAX, int KS>
__global__ __launch_bounds__(SK_THREADS) void skinny_kernel(
    const float* __restrict__ X1, int64_t ld1, int K1, const float* __restrict__ X2, int64_t ld2, int K2,
    const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias, int M, int N, int Kp,
    int blocks_per_wg, float* __restrict__ C, int64_t ldc, int cvec, float* __restrict__ part_max,
    float* __restrict__ part_sum, int64_t* __restrict__ part_arg, int n_split, int m_chunks) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];  // [SK_ROWS][Kp] + 4 spare
    __shared__ float red_v[SK_WAVES][SK_ROWS];
    __shared__ int red_i[SK_WAVES][SK_ROWS];
    __shared__ __attribute__((aligned(16))) float part[KS > 1 ? SK_WAVES : 1][2][64][4];
    constexpr int GC = ARGMAX ? SK_GC_STREAM : SK_GC;
    // workgroup id -> (column range bx, row chunk by): ids go round-robin over the 8 XCDs, so the row chunks of
    // one column range get consecutive slots of ONE XCD: they stream the same rows of W at about the same
    // time and the second read is served by that XCD's L2 instead of crossing the fabric again
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int bx = (seq / m_chunks) * 8 + xcd, by = seq % m_chunks;
    if (bx >= n_split) return;
    const int K = K1 + K2;
    const int m0 = by * SK_ROWS;
    {
        const RowImage<SK_ROWS, 16> img(X1, ld1, K1, X2, ld2, K2, m0, M, Kp);
        f32x4 iv[16];
        img.load(iv, 0, 0);
        img.store(Xs, iv, K, 0);
        img.rest(Xs, iv, 0, K);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int nG = (K + 15) >> 4;
    const int NB = (N + 15) >> 4;
    const int b_begin = bx * blocks_per_wg, b_end = min(NB, b_begin + blocks_per_wg);
    float best_v[2] = {SK_NEG, SK_NEG};
    int best_i[2] = {0, 0};
    const float* xs0 = Xs + r16 * Kp + 4 * kq;
    const float* xs1 = xs0 + 16 * Kp;

    constexpr int SLOTS = SK_WAVES / KS;  // blocks in flight per workgroup
    const int slot = wave / KS, kpart = wave % KS;
    const int gper = (nG + KS - 1) / KS;
    const int g_lo = kpart * gper, g_hi = min(nG, g_lo + gper);
    const int iters = (blocks_per_wg + SLOTS - 1) / SLOTS;  // uniform over the workgroup (barriers inside)
    for (int it = 0; it < iters; ++it) {
        const int blk = b_begin + it * SLOTS + slot;
        const bool active = blk < b_end;
        const int n = min(blk * 16 + r16, N - 1);  // clamped rows are computed but never used
        const float* wrow = W + (int64_t)n * ldw + 4 * kq;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            f32x4 cur[GC], nxt[GC];
#pragma unroll
            for (int i = 0; i < GC; ++i) {
                const int g = g_lo + i;
                const bool ok = g < g_hi && 16 * g + 4 * kq < K;
                cur[i] = ok ? *reinterpret_cast<const f32x4*>(wrow + 16 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int g0 = g_lo; g0 < g_hi; g0 += GC) {
#pragma unroll
                for (int i = 0; i < GC; ++i) {
                    const int g = g0 + GC + i;
                    const bool ok = g < g_hi && 16 * g + 4 * kq < K;
                    nxt[i] = ok ? *reinterpret_cast<const f32x4*>(wrow + 16 * g) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int i = 0; i < GC; ++i) {
                    if (g0 + i < g_hi) {
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(xs0 + 16 * (g0 + i));
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(xs1 + 16 * (g0 + i));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[i][j], x0[j], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[i][j], x1[j], acc1, 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < GC; ++i) cur[i] = nxt[i];
            }
        }
        if (KS > 1) {
            // sum the k ranges: every wave leaves its partial, the first wave of the slot adds them up
            *reinterpret_cast<f32x4*>(&part[wave][0][lane][0]) = acc0;
            *reinterpret_cast<f32x4*>(&part[wave][1][lane][0]) = acc1;
            __syncthreads();
            if (kpart == 0) {
#pragma unroll
                for (int q = 1; q < KS; ++q) {
                    acc0 += *reinterpret_cast<const f32x4*>(&part[wave + q][0][lane][0]);
                    acc1 += *reinterpret_cast<const f32x4*>(&part[wave + q][1][lane][0]);
                }
            }
            __syncthreads();
            if (kpart != 0) continue;
        }
        if (!active) continue;
        // lane holds columns nb..nb+3 of activation rows (m0 + r16) and (m0 + 16 + r16)
        const int nb = blk * 16 + 4 * kq;
        float bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[r] = (bias && nb + r < N) ? bias[nb + r] : 0.f;
        if (ARGMAX) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (nb + r < N) {
                    const float v0 = acc0[r] + bv[r], v1 = acc1[r] + bv[r];
                    if (v0 > best_v[0]) { best_v[0] = v0; best_i[0] = nb + r; }
                    if (v1 > best_v[1]) { best_v[1] = v1; best_i[1] = nb + r; }
                }
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = m0 + mt * 16 + r16;
                if (m >= M) continue;
                const f32x4 a = mt == 0 ? acc0 : acc1;
                float* c = C + (int64_t)m * ldc + nb;
                if (cvec && nb + 3 < N) {
                    *reinterpret_cast<float4*>(c) = make_float4(a[0] + bv[0], a[1] + bv[1], a[2] + bv[2], a[3] + bv[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < N) c[r] = a[r] + bv[r];
                }
            }
        }
    }
    if (!ARGMAX) return;
    // best over the 4 lane groups of the wave (ties: the smaller column), then over the 8 waves
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        float v = best_v[mt];
        int ix = best_i[mt];
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float v2 = __shfl_xor(v, o, 64);
            const int i2 = __shfl_xor(ix, o, 64);
            if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
        }
        if (kq == 0) { red_v[wave][mt * 16 + r16] = v; red_i[wave][mt * 16 + r16] = ix; }
    }
    __syncthreads();
    if (threadIdx.x < SK_ROWS) {
        const int m = m0 + threadIdx.x;
        if (m < M) {
            float v = red_v[0][threadIdx.x];
            int ix = red_i[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < SK_WAVES; ++w) {
                const float v2 = red_v[w][threadIdx.x];
                const int i2 = red_i[w][threadIdx.x];
                if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
            }
            part_max[(int64_t)m * n_split + bx] = v;
            part_sum[(int64_t)m * n_split + bx] = 1.f;
            part_arg[(int64_t)m * n_split + bx] = ix;
        }
    }
}


// The arg-max for decode batches of 33 .. 64 rows per row chunk (generation at eval_bs 50): with 32-row workgroups
// every element of W is streamed twice and feeds 8 MFMAs per KB -- at the full MFMA rate that is 8 TB/s into the
// CUs, so the kernel sat at half the matrix rate.  Here a workgroup holds 64 activation rows (4 MFMA tiles) and
// every wave owns ONE 16-column block of W for the whole kernel: 16 MFMAs per KB of W, W streamed once.  64 rows
// x K = 900 do not fit LDS (231 KB), so K is cut into passes (2 at K = 900): the image of a pass's k range is
// built in LDS, the waves stream that k range of their block, and the block's 4 accumulators stay in registers
// across the passes.  The first weight chunk of a pass is in flight while the image is built.
constexpr int SK64_ROWS = 64, SK64_TILES = 4;
#ifdef SK_PROFILE
// development build: shader-clock and 100 MHz wall-clock ticks of workgroup 0's wave 0 over the kernel -> the clock
// the chip actually runs the generation loop at
__device__ unsigned long long g_skclk[2];
__device__ unsigned long long g_skg[8];   // skinny_gru_kernel, workgroup 0 wave 0: 100 MHz stamps of its phases
#define SKG_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_skg[i] = wall_clock64(); } while (0)
#else
#define SKG_STAMP(i) do {} while (0)
#endif

#ifdef SK_NO_LOAD   // development: the matrix work alone (no weight stream)
#define SK_LOADW(p) ((f32x4){(float)(uintptr_t)(p), 1.f, 2.f, 3.f})
#else
#define SK_LOADW(p) (*reinterpret_cast<const f32x4*>(p))
#endif
constexpr int SK64_GC = 8;  // k-groups per register chunk: 8 KB per wave, 128 KB per CU in flight

// One register chunk of the stream: GC k-groups x (4 activation reads from LDS + 16 MFMAs).  FULL = every group of
// the chunk is inside the pass: no branches at all, so the compiler counts the outstanding weight loads (the prefetch
// of the next chunk stays in flight: s_waitcnt vmcnt(n > 0)) and moves the LDS reads ahead of the MFMAs that hide them.
// TILES 16-row tiles go through the matrix pipe; VR more rows (the one or two left behind the last full tile of a decode
// batch: rows 48, 49 of the reference's eval_bs = 50) are dot products on the vector ALUs -- the lane's four weights
// times the row's four activations of the same k, summed over the lane groups at the end -- instead of a fourth MFMA
// tile that would multiply 14 rows of padding (a quarter of the launch's matrix work).
template <bool FULL, int TILES, int VR>
__device__ __forceinline__ void sk64_chunk(f32x4 (&acc)[TILES], float (&pv)[VR > 0 ? VR : 1], const f32x4 (&cur)[SK64_GC],
                                           const float* xs, const float* xv, int Kp, int goff, int n_valid) {
#pragma unroll
    for (int i = 0; i < SK64_GC; ++i) {
        if (FULL || i < n_valid) {
            f32x4 x[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t)
                x[t] = *reinterpret_cast<const f32x4*>(xs + t * 16 * Kp + 16 * (goff + i));
            f32x4 xr[VR > 0 ? VR : 1];
#pragma unroll
            for (int v = 0; v < VR; ++v) xr[v] = *reinterpret_cast<const f32x4*>(xv + v * Kp + 16 * (goff + i));
#ifdef SK_NO_MFMA   // development: the stream alone (the loaded values are consumed by cheap adds)
#pragma unroll
            for (int t = 0; t < TILES; ++t) acc[t] += cur[i] * x[t];
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[i][j], x[t][j], acc[t], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VR; ++v) pv[v] = fmaf(cur[i][j], xr[v][j], pv[v]);
            }
#endif
        }
    }
}

template <int TILES, int VR>
__global__ __launch_bounds__(SK_THREADS) void skinny_argmax64_kernel(
    const float* __restrict__ X1, int64_t ld1, int K1, const float* __restrict__ X2, int64_t ld2, int K2,
    const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias, int M, int N, int Kp, int g_per_pass,
    float* __restrict__ part_max, float* __restrict__ part_sum, int64_t* __restrict__ part_arg, int n_split,
    int m_chunks) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];  // [64][Kp]: the k range of the current pass; + 4 spare
    __shared__ float red_v[SK_WAVES][SK64_ROWS];
    __shared__ int red_i[SK_WAVES][SK64_ROWS];
    constexpr int GC = SK64_GC;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int bx = (seq / m_chunks) * 8 + xcd, by = seq % m_chunks;
    if (bx >= n_split) return;
    const int K = K1 + K2;
    const int m0 = by * SK64_ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#ifdef SK_PROFILE
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
#endif
    const int r16 = lane & 15, kq = lane >> 4;
    const int nG = (K + 15) >> 4;
    const int NB = (N + 15) >> 4;
    const int blk = bx * SK_WAVES + wave;
    const bool active = blk < NB;
    const int n = min(blk * 16 + r16, N - 1);  // clamped rows are computed but never used
    const float* wrow = W + (int64_t)n * ldw;
    // Unconditional loads, addresses clamped into the row (a group past the pass, or a lane past K, re-reads valid
    // weights of its row: such a group is never multiplied -- the tail chunk is guarded -- and such a lane meets the
    // image's zero padding).  A load inside a branch, or one whose value goes through a select, hides the count of
    // outstanding loads from the compiler or drags the load down to its use: either way the prefetch of the next
    // chunk was waited for before the first MFMA of every chunk.
    auto loadw = [&](int g, int g_hi) -> f32x4 {
        const int k = 16 * min(g, g_hi - 1) + 4 * kq;
        return SK_LOADW(wrow + min(k, K - 4));
    };
    f32x4 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float pv[VR > 0 ? VR : 1] = {};
    const float* xs = Xs + r16 * Kp + 4 * kq;
    const float* xv = Xs + 16 * TILES * Kp + 4 * kq;   // rows 16 TILES .. of the image, the same for the 16 lanes of a group
    const RowImage<SK64_ROWS, 16> img(X1, ld1, K1, X2, ld2, K2, m0, M, Kp);
    for (int g_lo = 0; g_lo < nG; g_lo += g_per_pass) {
        const int g_hi = min(nG, g_lo + g_per_pass);
        f32x4 cur[GC], nxt[GC];
        // The pass's image (RowImage).  Order: image loads, then a barrier that waits for no memory (every wave has
        // issued its image loads and is done with the previous pass's image), then the first weight chunk, in flight
        // while the image is stored -- loads return in order, so the image is not queued behind other waves' weights.
        const int k_lo = 16 * g_lo, kvalid = min(16 * (g_hi - g_lo), K - k_lo);
        {
            f32x4 iv[16];
            img.load(iv, k_lo, 0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GC; ++i) cur[i] = loadw(g_lo + i, g_hi);
            __builtin_amdgcn_sched_barrier(0);
            img.store(Xs, iv, kvalid, 0);
            img.rest(Xs, iv, k_lo, kvalid);   // (passes wider than 512 columns)
        }
        __syncthreads();
        if (!active) continue;
        int g0 = g_lo;
        for (; g0 + GC <= g_hi; g0 += GC) {
#pragma unroll
            for (int i = 0; i < GC; ++i) nxt[i] = loadw(g0 + GC + i, g_hi);
            __builtin_amdgcn_sched_barrier(0);  // the prefetch is issued before the chunk's MFMAs, not after
            sk64_chunk<true, TILES, VR>(acc, pv, cur, xs, xv, Kp, g0 - g_lo, GC);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GC; ++i) cur[i] = nxt[i];
        }
        if (g0 < g_hi) sk64_chunk<false, TILES, VR>(acc, pv, cur, xs, xv, Kp, g0 - g_lo, g_hi - g0);
    }
#ifdef SK_PROFILE
    asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[TILES - 1]), "v"(pv[0]));
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        g_skclk[0] = clock64() - c0;
        g_skclk[1] = wall_clock64() - w0;
    }
#endif
    // lane holds columns nb..nb+3 of activation rows m0 + 16 t + r16
    float best_v[TILES];
    int best_i[TILES];
    const int nb = blk * 16 + 4 * kq;
#pragma unroll
    for (int t = 0; t < TILES; ++t) { best_v[t] = SK_NEG; best_i[t] = 0; }
    if (active) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (nb + r < N) {
                const float bv = bias ? bias[nb + r] : 0.f;
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const float v = acc[t][r] + bv;
                    if (v > best_v[t]) { best_v[t] = v; best_i[t] = nb + r; }
                }
            }
        }
    }
    // best over the 4 lane groups of the wave (ties: the smaller column), then over the 8 waves
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        float v = best_v[t];
        int ix = best_i[t];
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float v2 = __shfl_xor(v, o, 64);
            const int i2 = __shfl_xor(ix, o, 64);
            if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
        }
        if (kq == 0) { red_v[wave][t * 16 + r16] = v; red_i[wave][t * 16 + r16] = ix; }
    }
    // the vector-ALU rows: a lane group holds a quarter of every k group's products of column blk * 16 + r16; summed
    // over the four groups every lane has the column's logit, then the best of the wave's 16 columns
#pragma unroll
    for (int q = 0; q < VR; ++q) {
        float p = pv[q];
        p += __shfl_xor(p, 16, 64);
        p += __shfl_xor(p, 32, 64);
        const int col = blk * 16 + r16;
        const bool ok = active && col < N;
        float v = ok ? p + (bias ? bias[min(col, N - 1)] : 0.f) : SK_NEG;
        int ix = ok ? col : 0;
#pragma unroll
        for (int o = 1; o <= 8; o <<= 1) {
            const float v2 = __shfl_xor(v, o, 64);
            const int i2 = __shfl_xor(ix, o, 64);
            if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
        }
        if (lane == 0) { red_v[wave][16 * TILES + q] = v; red_i[wave][16 * TILES + q] = ix; }
    }
    __syncthreads();
    if (threadIdx.x < SK64_ROWS) {
        const int m = m0 + threadIdx.x;
        if (m < M) {
            float v = red_v[0][threadIdx.x];
            int ix = red_i[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < SK_WAVES; ++w) {
                const float v2 = red_v[w][threadIdx.x];
                const int i2 = red_i[w][threadIdx.x];
                if (v2 > v || (v2 == v && i2 < ix)) { v = v2; ix = i2; }
            }
            part_max[(int64_t)m * n_split + bx] = v;
            part_sum[(int64_t)m * n_split + bx] = 1.f;
            part_arg[(int64_t)m * n_split + bx] = ix;
        }
    }
}

// One GRU cell step for a decode batch, fused: gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh and the gate math
// (decoder/rnn_decoder.py:61-66 from the greedy loop :160-168) in ONE launch instead of three.  A workgroup owns
// 16 hidden units for 16 MT batch rows; [x | h] sits in LDS; its twelve waves each stream half the K range of
// one of the six (gate, operand) weight blocks -- 16 rows of W_ih or W_hh -- into the MFMA A operand (the
// products are chains of memory round trips: twelve short chains instead of six long ones); the partial
// pre-activations meet in LDS and one wave per 16-row tile applies the cell and writes h'.
// h' goes to a buffer of its own (other workgroups still read h).
constexpr int SKG_WAVES = 12;
constexpr int SKG_THREADS = SKG_WAVES * 64;
constexpr int SKG_GC = 12;  // 16-column groups of W per chunk, double-buffered (a 20-group single buffer -- the whole k range of
                            // H, In <= 640 requested at once -- measured 0.8 % slower end to end)

__device__ __forceinline__ float sk_sigm(float x) { return 1.f / (1.f + __expf(-x)); }

// PICK: the step's input rows are not in X yet -- the previous step's arg-max left its per-split partials, and this
// launch first picks every sentence's token from them (what combine_gather_kernel does: best of the S splits, ties to
// the smaller split), reads the token's embedding row straight into the image, and the workgroups of unit block 0
// write tokens and rows out for the projection that follows.  One launch (4 us + the gap in front of it) less per step.
struct SkgPick {
    const float* pm;       // [M][S] split maxima
    const int64_t* pa;     // [M][S] their columns
    int S;
    int64_t* tok_out;      // [M]
    const float* table;    // embedding rows
    int64_t ldt;
    float* xout;           // [M][In] the picked rows, for the projection's second panel
    int64_t ldxo;
};

// LSTM: the same launch for nn.LSTM's cell (gate order i, f, g, o; decoder/rnn_decoder.py:43,65 with rnn_cell = "lstm").
// Its pre-activations are plain sums gi + gh, so [x | h] is ONE k range per gate: the twelve waves are 4 gates x 3
// thirds of the concatenated range (a wave's third may straddle the panels: the weight row is picked per group), the
// cell adds the three partials, reads c (Cp) and writes c' (Cn; Cn == Cp is fine: an element is read and written by
// one lane) beside h'.
struct SkgCell {
    const float* cp;   // [M][H] c of the previous step (LSTM)
    float* cn;         // [M][H] c' (may alias cp)
    int64_t ldc;
};

template <int MT, bool PICK, bool LSTM>
__global__ __launch_bounds__(SKG_THREADS) void skinny_gru_kernel(
    const float* __restrict__ X, int64_t ldx, int In, const float* __restrict__ Hp, int64_t ldh, int H,
    const float* __restrict__ Wih, int64_t ldi, const float* __restrict__ Whh, int64_t ldhh,
    const float* __restrict__ bih, const float* __restrict__ bhh, int M, int Kp, float* __restrict__ Hn, int64_t ldo,
    int n_ub, int m_chunks, SkgPick pk, SkgCell cell) {
    extern __shared__ __attribute__((aligned(16))) float Xs[];  // [16 MT][Kp] = [x .. 0 | h .. 0], each padded to 16; + 4 spare
    __shared__ __attribute__((aligned(16))) float part[SKG_WAVES][MT][64][4];
    __shared__ int tok_s[16 * MT];
    constexpr int ROWS = 16 * MT;
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;  // row chunks of one unit block share an XCD (see above)
    const int bx = (seq / m_chunks) * 8 + xcd, by = seq % m_chunks;
    if (bx >= n_ub) return;
    SKG_STAMP(0);
    const int Inp = (In + 15) & ~15;
    const int m0 = by * ROWS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int u0 = bx * 16;
    // wave -> (weight block, k half): waves land on SIMD wave % 4, and the x blocks are longer than the h blocks when
    // In > H; swapping the pairs of the second quad gives every SIMD both kinds instead of three of one
    // (LSTM: wave -> (gate = wave % 4, third = wave / 4): every SIMD holds one gate's three thirds)
    const int slot = LSTM ? wave : wave ^ ((wave >> 2) & 1);
    const int prod = slot % 6, khalf = slot / 6;
    const int gate = LSTM ? (wave & 3) : (prod >> 1), panel = LSTM ? 0 : (prod & 1);
    const int Kw = panel ? H : In, xoff = panel ? Inp : 0;
    const float* wrow = (panel ? Whh : Wih) + (int64_t)(gate * H + min(u0 + r16, H - 1)) * (panel ? ldhh : ldi) + 4 * kq;
    const float* wrow_h = Whh + (int64_t)(gate * H + min(u0 + r16, H - 1)) * ldhh + 4 * kq;   // (LSTM: the h panel's row)
    const int nGx = Inp >> 4;
    const int nG = LSTM ? nGx + ((H + 15) >> 4) : (Kw + 15) >> 4, gper = LSTM ? (nG + 2) / 3 : (nG + 1) >> 1;
    const int g_lo = (LSTM ? wave >> 2 : khalf) * gper, g_hi = min(nG, g_lo + gper);
    // the first weight chunk is on its way while the workgroup builds the [x | h] image
    // Unconditional loads at clamped addresses (a group past the wave's range is never multiplied; a lane past Kw
    // meets the image's zero padding): guarded loads came out as a branch each, with a full wait after the second
    // -- a memory round trip more in a kernel that is three of them long.
    auto loadw = [&](int g) -> f32x4 {
        const int gc = min(g, max(g_hi, g_lo + 1) - 1);
        if constexpr (LSTM) {
            const bool hp = gc >= nGx;   // which panel the group lies in (the image is [x .. 0 | h .. 0]: same index)
            const int k = 16 * (hp ? gc - nGx : gc);
            return *reinterpret_cast<const f32x4*>((hp ? wrow_h : wrow) + max(min(k, (hp ? H : In) - 4 - 4 * kq), -4 * kq));
        } else {
            const int k = 16 * gc;   // (wrow already carries the lane's 4 * kq)
            return *reinterpret_cast<const f32x4*>(wrow + max(min(k, Kw - 4 - 4 * kq), -4 * kq));
        }
    };
    // The [x | h] image: a thread owns one row and every TPR-th 16-byte column of it (no division by a run-time
    // value, no branch around a load: clamped addresses, masked when stored).  The image's loads are issued BEFORE
    // the first weight chunk -- loads return in order, so the image is complete a chunk's transfer time earlier
    // and the workgroup meets at the barrier while the weights are still arriving.
    constexpr int TPR = SKG_THREADS / ROWS, IU = 8;
    const int kp4 = Kp >> 2;
    const int ir = threadIdx.x / TPR, ic = threadIdx.x % TPR;
    const bool row_ok = m0 + ir < M;
    const float* hrow = Hp + (int64_t)min(m0 + ir, M - 1) * ldh;
    f32x4 cur[SKG_GC], nxt[SKG_GC];
    if (PICK) {
        // wave w picks the tokens of rows w, w + 12, ...: every lane takes every 64th split (value and column in one
        // batch of loads, clamped and masked), then the wave's shuffle tree.  The first weight chunk is requested
        // behind these loads and arrives while the tokens are picked.
        constexpr int PJ = 4;
        for (int r = wave; r < ROWS; r += SKG_WAVES) {
            const int m = min(m0 + r, M - 1);
            const float* rm = pk.pm + (int64_t)m * pk.S;
            const int64_t* ra = pk.pa + (int64_t)m * pk.S;
            float mx = -3.0e38f;
            int best = 0x7fffffff, arg = 0;
            for (int s0 = 0; s0 < pk.S; s0 += 64 * PJ) {
                float v[PJ];
                int64_t a[PJ];
#pragma unroll
                for (int j = 0; j < PJ; ++j) {
                    const int sc = min(s0 + lane + 64 * j, pk.S - 1);
                    v[j] = rm[sc];
                    a[j] = ra[sc];
                }
                if (r == wave && s0 == 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < SKG_GC; ++i) cur[i] = loadw(g_lo + i);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int j = 0; j < PJ; ++j) {
                    const int sj = s0 + lane + 64 * j;
                    if (sj < pk.S && v[j] > mx) { mx = v[j]; best = sj; arg = (int)a[j]; }
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float m2 = __shfl_xor(mx, o, 64);
                const int b2 = __shfl_xor(best, o, 64), a2 = __shfl_xor(arg, o, 64);
                if (m2 > mx || (m2 == mx && b2 < best)) { mx = m2; best = b2; arg = a2; }
            }
            if (best == 0x7fffffff) arg = (int)ra[0];   // (no split above the floor: the first, as combine_gather_kernel)
            if (lane == 0) {
                tok_s[r] = arg;
                if (bx == 0 && m0 + r < M) pk.tok_out[m0 + r] = arg;
            }
        }
        __syncthreads();
    }
    const float* xrow = PICK ? pk.table + (int64_t)tok_s[PICK ? ir : 0] * pk.ldt : X + (int64_t)min(m0 + ir, M - 1) * ldx;
    auto img_load = [&](int c0, f32x4* v) {
#pragma unroll
        for (int u = 0; u < IU; ++u) {
            const int k = 4 * (c0 + ic + u * TPR);
            v[u] = *reinterpret_cast<const f32x4*>(k < Inp ? xrow + min(k, In - 4) : hrow + min(k - Inp, H - 4));
        }
    };
    auto img_store = [&](int c0, const f32x4* v) {
#pragma unroll
        for (int u = 0; u < IU; ++u) {
            const int c = c0 + ic + u * TPR, k = 4 * c;
            const bool live = row_ok && (k < Inp ? k < In : k - Inp < H);
            // a column past the row goes to the spare 16 bytes behind the image: an unconditional store keeps the
            // compiler from sinking its load into a branch (where it would be waited for with everything else)
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const unsigned keep = live ? 0xffffffffu : 0u;   // (masked with AND: a select on the loaded value came out as a branch)
            *reinterpret_cast<u32x4*>(__builtin_assume_aligned(Xs + (c < kp4 ? ir * Kp + k : ROWS * Kp), 16)) =
                __builtin_bit_cast(u32x4, v[u]) & (u32x4){keep, keep, keep, keep};
            if (PICK && bx == 0 && row_ok && k < In)
                *reinterpret_cast<f32x4*>(pk.xout + (int64_t)(m0 + ir) * pk.ldxo + k) = v[u];
        }
    };
    {
        f32x4 iv[IU];
        img_load(0, iv);
        if (!PICK) {
            __builtin_amdgcn_sched_barrier(0);
            // (no memory is waited for here: the twelve waves only agree that every image load has been issued, so
            // the image -- a third of what the CU pulls in before the first product -- is not queued behind other
            // waves' weights)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < SKG_GC; ++i) cur[i] = loadw(g_lo + i);
            __builtin_amdgcn_sched_barrier(0);
        }
        img_store(0, iv);
        for (int c0 = TPR * IU; c0 < kp4; c0 += TPR * IU) {   // (32-row images, K beyond 1536)
            img_load(c0, iv);
            img_store(c0, iv);
        }
    }
    SKG_STAMP(1);
    __syncthreads();
    SKG_STAMP(2);
    {
        const float* xs = Xs + r16 * Kp + xoff + 4 * kq;
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int g0 = g_lo; g0 < g_hi; g0 += SKG_GC) {
            if (g0 + SKG_GC < g_hi) {   // (wave-uniform; only K ranges beyond one chunk: H > 768)
#pragma unroll
                for (int i = 0; i < SKG_GC; ++i) nxt[i] = loadw(g0 + SKG_GC + i);
            }
#pragma unroll
            for (int i = 0; i < SKG_GC; ++i) {
                if (g0 + i < g_hi) {
                    // past Kw both the weight registers and the image's padding are zero
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const f32x4 xv = *reinterpret_cast<const f32x4*>(xs + mt * 16 * Kp + 16 * (g0 + i));
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[i][j], xv[j], acc[mt], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < SKG_GC; ++i) cur[i] = nxt[i];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<f32x4*>(&part[slot][mt][lane][0]) = acc[mt];
    }
    SKG_STAMP(3);
    // the cell's biases for the waves that apply it: all 24 loads back to back at clamped indices BEFORE the barrier
    // (one memory round trip under the wait for the other waves; guarded loads inside the per-unit branches below
    // were a round trip per unit)
    constexpr int NGATE = LSTM ? 4 : 3;
    float bi[NGATE][4], bh[NGATE][4], cprev[4];
    if (wave < MT) {
#pragma unroll
        for (int g = 0; g < NGATE; ++g)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int u = min(u0 + 4 * kq + c, H - 1);
                bi[g][c] = bih[g * H + u];
                bh[g][c] = bhh[g * H + u];
            }
        if constexpr (LSTM) {
            const float* crow = cell.cp + (int64_t)min(m0 + 16 * wave + r16, M - 1) * cell.ldc;
#pragma unroll
            for (int c = 0; c < 4; ++c) cprev[c] = crow[min(u0 + 4 * kq + c, H - 1)];
        }
    }
    __syncthreads();
    SKG_STAMP(4);
    if (wave < MT) {
        // lane holds units u0 + 4 kq .. +3 of batch row m0 + 16 wave + r16; k halves added low + high
        const int m = m0 + 16 * wave + r16;
        if constexpr (LSTM) {
            f32x4 pre[4];   // gate q: the three thirds of its k range (waves q, 4 + q, 8 + q)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pre[q] = (*reinterpret_cast<const f32x4*>(&part[q][wave][lane][0]) +
                          *reinterpret_cast<const f32x4*>(&part[4 + q][wave][lane][0])) +
                         *reinterpret_cast<const f32x4*>(&part[8 + q][wave][lane][0]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int u = u0 + 4 * kq + c;
                if (m < M && u < H) {
                    const float gi_ = sk_sigm(pre[0][c] + (bi[0][c] + bh[0][c]));
                    const float gf = sk_sigm(pre[1][c] + (bi[1][c] + bh[1][c]));
                    const float gg = tanhf(pre[2][c] + (bi[2][c] + bh[2][c]));
                    const float go = sk_sigm(pre[3][c] + (bi[3][c] + bh[3][c]));
                    const float cn = gf * cprev[c] + gi_ * gg;
                    cell.cn[(int64_t)m * cell.ldc + u] = cn;
                    Hn[(int64_t)m * ldo + u] = go * tanhf(cn);
                }
            }
        } else {
            f32x4 pre[6];
#pragma unroll
            for (int q = 0; q < 6; ++q)
                pre[q] = *reinterpret_cast<const f32x4*>(&part[q][wave][lane][0]) +
                         *reinterpret_cast<const f32x4*>(&part[6 + q][wave][lane][0]);
            const float* hold = Xs + (16 * wave + r16) * Kp + Inp;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int u = u0 + 4 * kq + c;
                if (m < M && u < H) {
                    // same association as the unfused path: (x W^T + b_ih) + (h W^T + b_hh)
                    const float r = sk_sigm((pre[0][c] + bi[0][c]) + (pre[1][c] + bh[0][c]));
                    const float z = sk_sigm((pre[2][c] + bi[1][c]) + (pre[3][c] + bh[1][c]));
                    const float n = tanhf((pre[4][c] + bi[2][c]) + r * (pre[5][c] + bh[2][c]));
                    Hn[(int64_t)m * ldo + u] = (1.f - z) * n + z * hold[u];
                }
            }
        }
    }
    SKG_STAMP(5);
}

int sk_cus() {
    static int cus = 0;
    if (cus == 0) {
        int n = 0;
        if (dssvae_device_cus(&n) != 0 || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

bool sk_ok(const float* X1, int64_t ld1, int K1, const float* X2, int64_t ld2, int K2, const float* W, int64_t ldw,
           int M) {
    if (M > DSSVAE_SKINNY_MAX_ROWS) return false;
    if ((K1 % 4) || (K2 % 4) || (ld1 % 4) || (ldw % 4) || !aligned16(X1) || !aligned16(W)) return false;
    if (K2 > 0 && ((ld2 % 4) || !aligned16(X2))) return false;
    return sk_image_fits(K1 + K2);
}

template <bool ARGMAX, int KS>
int sk_launch(const float* X1, int64_t ld1, int K1, const float* X2, int64_t ld2, int K2, const float* W,
              int64_t ldw, const float* bias, int M, int N, int bpw, int n_wg, float* C, int64_t ldc, float* pm,
              float* ps, int64_t* pa, hipStream_t s) {
    const int Kp = padded_k(K1 + K2);
    const int lds = SK_ROWS * Kp * 4 + 16;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_kernel<ARGMAX, KS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int cvec = C && aligned16(C) && (ldc % 4) == 0;
    const int m_chunks = (M + SK_ROWS - 1) / SK_ROWS;
    hipLaunchKernelGGL((skinny_kernel<ARGMAX, KS>), dim3(((n_wg + 7) / 8) * 8 * m_chunks), dim3(SK_THREADS), lds, s, X1,
                       ld1, K1, X2, ld2, K2, W, ldw, bias, M, N, Kp, bpw, C, ldc, cvec, pm, ps, pa, n_wg, m_chunks);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

// 16-column blocks per workgroup so that the launch is about one workgroup per CU
int argmax_bpw(int M, int N) {
    const int NB = (N + 15) / 16, m_chunks = (M + SK_ROWS - 1) / SK_ROWS;
    const int target = max(1, sk_cus() / m_chunks);
    int bpw = (NB + target - 1) / target;
    return (bpw + SK_WAVES - 1) / SK_WAVES * SK_WAVES;
}

// Small vocabularies (the 300-symbol tree vocabulary: 19 blocks) leave a wave ONE block with the whole K range behind it:
// a chain of K / 256 memory round trips (3 at K = 600; 20 us per launch for 0.7 MB of weights).  While the launch
// stays within a workgroup per CU, four waves share a block instead (KS = 4: a quarter of the chain each, partials
// summed through LDS, as in the gate products) and a workgroup takes 2 blocks.
constexpr int SK_ARGMAX_KS = 4;
bool argmax_ksplit(int M, int N) {
    const int NB = (N + 15) / 16, m_chunks = (M + SK_ROWS - 1) / SK_ROWS, slots = SK_WAVES / SK_ARGMAX_KS;
    return (NB + slots - 1) / slots * m_chunks <= sk_cus();
}

// 64-row arg-max kernel: k-groups (of 16) per pass so that the [64][16 g + 4] image fits the LDS budget; used for
// decode batches of more than one 32-row chunk.  DSSVAE_SKINNY_ROWS=32 keeps the 32-row kernel (A/B, tests).
struct Plan64 { int g_per_pass, Kp, n_split; bool use; };
Plan64 plan64(int M, int N, int K) {
    Plan64 p;
    const int nG = (K + 15) / 16;
    const int g_max = (SK_LDS_BUDGET / (SK64_ROWS * 4) - 4) / 16;  // groups whose image fits
    const int n_pass = (nG + g_max - 1) / g_max;
    p.g_per_pass = (nG + n_pass - 1) / n_pass;
    p.Kp = 16 * p.g_per_pass + 4;
    p.n_split = ((N + 15) / 16 + SK_WAVES - 1) / SK_WAVES;
    const char* f = getenv("DSSVAE_SKINNY_ROWS");
    // a wave owns one 16-column block: only vocabularies that give every CU a workgroup or so (the 30k sentence
    // vocabulary: 235) -- the 300-symbol tree vocabulary is 3 workgroups per row chunk and stays with the 32-row kernel
    p.use = M > SK_ROWS && !(f && atoi(f) == 32) &&
            p.n_split * ((M + SK64_ROWS - 1) / SK64_ROWS) >= sk_cus() / 2;
    return p;
}

}  // namespace

#ifdef SK_PROFILE
extern "C" int dssvae_skinny_profile_read(unsigned long long* dst_host) {
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_skclk), sizeof(g_skclk), 0, hipMemcpyDeviceToHost);
}
extern "C" int dssvae_skinny_gru_profile_read(unsigned long long* dst_host) {
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_skg), sizeof(g_skg), 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int dssvae_skinny_supported(int M, int K1, int K2, int64_t ld1, int64_t ld2, int64_t ldw) {
    // alignment of the pointers is checked at the call; this answers for the shape
    if (M <= 0 || M > DSSVAE_SKINNY_MAX_ROWS || K1 <= 0 || K2 < 0) return 0;
    if ((K1 % 4) || (K2 % 4) || (ld1 % 4) || (ldw % 4) || (K2 > 0 && (ld2 % 4))) return 0;
    return sk_image_fits(K1 + K2);
}

extern "C" int dssvae_skinny_argmax_splits(int M, int N) {
    if (M <= 0 || N <= 0) return DSSVAE_EINVAL;
    if (M > SK_ROWS) {  // the 64-row kernel's split count does not depend on K
        const Plan64 p = plan64(M, N, 16);
        if (p.use) return p.n_split;
    }
    const int NB = (N + 15) / 16;
    const int bpw = argmax_ksplit(M, N) ? SK_WAVES / SK_ARGMAX_KS : argmax_bpw(M, N);
    return (NB + bpw - 1) / bpw;
}

extern "C" int dssvae_skinny_linear(const float* X1, int64_t ld1, int K1, const float* X2, int64_t ld2, int K2,
                                    const float* W, int64_t ldw, const float* bias, int M, int N, float* C,
                                    int64_t ldc, void* stream) {
    if (!X1 || !W || !C || M <= 0 || N <= 0 || K1 <= 0 || K2 < 0 || (K2 > 0 && !X2) || ld1 < K1 || ldw < K1 + K2 ||
        ldc < N || (K2 > 0 && ld2 < K2))
        return DSSVAE_EINVAL;
    if (!sk_ok(X1, ld1, K1, X2, ld2, K2, W, ldw, M)) return DSSVAE_ETOOBIG;
    const int NB = (N + 15) / 16;
    constexpr int KS = 4, BPW = SK_WAVES / KS;
    return sk_launch<false, KS>(X1, ld1, K1, X2, ld2, K2, W, ldw, bias, M, N, BPW, (NB + BPW - 1) / BPW, C, ldc,
                                nullptr, nullptr, nullptr, (hipStream_t)stream);
}

// Which rows of a single 64-row chunk go where: full 16-row tiles on the matrix pipe and, when one or two rows are left
// behind them, those on the vector ALUs (M = 50: 3 tiles + 2 rows); three or more left-over rows take a tile of their
// own.  DSSVAE_SKINNY_VALU_ROWS=0: tiles only (A/B).  Launches of several chunks keep four tiles per chunk.
struct Shape64 { int tiles, vr; };
Shape64 shape64(int M) {
    if (M > SK64_ROWS) return {SK64_TILES, 0};
    static const bool valu = !(getenv("DSSVAE_SKINNY_VALU_ROWS") && atoi(getenv("DSSVAE_SKINNY_VALU_ROWS")) == 0);
    const int full = M / 16, rem = M % 16;
    if (rem == 0) return {full, 0};
    if (valu && rem <= 2 && full >= 2) return {full, rem};
    return {full + 1, 0};
}

extern "C" int dssvae_skinny_argmax_rows(int M, int N) {
    if (M <= 0 || N <= 0) return DSSVAE_EINVAL;
    return plan64(M, N, 16).use ? SK64_ROWS : SK_ROWS;
}

extern "C" int dssvae_skinny_argmax_padded_rows(int M, int N) {
    if (M <= 0 || N <= 0) return DSSVAE_EINVAL;
    if (!plan64(M, N, 16).use) return (M + SK_ROWS - 1) / SK_ROWS * SK_ROWS;
    if (M > SK64_ROWS) return (M + SK64_ROWS - 1) / SK64_ROWS * SK64_ROWS;
    const Shape64 sh = shape64(M);
    return 16 * sh.tiles + sh.vr;
}

extern "C" int dssvae_skinny_argmax(const float* X1, int64_t ld1, int K1, const float* X2, int64_t ld2, int K2,
                                    const float* W, int64_t ldw, const float* bias, int M, int N, float* part_max,
                                    float* part_sum, int64_t* part_arg, void* stream) {
    if (!X1 || !W || !part_max || !part_sum || !part_arg || M <= 0 || N <= 0 || K1 <= 0 || K2 < 0 ||
        (K2 > 0 && !X2) || ld1 < K1 || ldw < K1 + K2 || (K2 > 0 && ld2 < K2))
        return DSSVAE_EINVAL;
    if (!sk_ok(X1, ld1, K1, X2, ld2, K2, W, ldw, M)) return DSSVAE_ETOOBIG;
    const Plan64 p64 = plan64(M, N, K1 + K2);
    if (p64.use) {
        const int lds = SK64_ROWS * p64.Kp * 4 + 16, m_chunks = (M + SK64_ROWS - 1) / SK64_ROWS;
        const Shape64 sh = shape64(M);
#define SK64_LAUNCH(T, V)                                                                                             \
    do {                                                                                                              \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_argmax64_kernel<T, V>),                        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                   \
        hipLaunchKernelGGL((skinny_argmax64_kernel<T, V>), dim3(((p64.n_split + 7) / 8) * 8 * m_chunks),              \
                           dim3(SK_THREADS), lds, (hipStream_t)stream, X1, ld1, K1, X2, ld2, K2, W, ldw, bias, M, N,  \
                           p64.Kp, p64.g_per_pass, part_max, part_sum, part_arg, p64.n_split, m_chunks);              \
    } while (0)
        switch (sh.tiles * 4 + sh.vr) {
            case 2 * 4 + 1: SK64_LAUNCH(2, 1); break;
            case 2 * 4 + 2: SK64_LAUNCH(2, 2); break;
            case 3 * 4 + 0: SK64_LAUNCH(3, 0); break;
            case 3 * 4 + 1: SK64_LAUNCH(3, 1); break;
            case 3 * 4 + 2: SK64_LAUNCH(3, 2); break;
            default: SK64_LAUNCH(4, 0); break;
        }
#undef SK64_LAUNCH
        DSSVAE_CHECK_LAUNCH();
        return DSSVAE_OK;
    }
    const int NB = (N + 15) / 16;
    if (argmax_ksplit(M, N)) {
        constexpr int bpw = SK_WAVES / SK_ARGMAX_KS;
        return sk_launch<true, SK_ARGMAX_KS>(X1, ld1, K1, X2, ld2, K2, W, ldw, bias, M, N, bpw, (NB + bpw - 1) / bpw,
                                             nullptr, 0, part_max, part_sum, part_arg, (hipStream_t)stream);
    }
    const int bpw = argmax_bpw(M, N);
    return sk_launch<true, 1>(X1, ld1, K1, X2, ld2, K2, W, ldw, bias, M, N, bpw, (NB + bpw - 1) / bpw, nullptr, 0,
                           part_max, part_sum, part_arg, (hipStream_t)stream);
}

namespace {
int skg_launch(const float* X, int64_t ldx, int In, const float* Hprev, int64_t ldh, int H, const float* Wih,
               int64_t ldi, const float* Whh, int64_t ldhh, const float* bih, const float* bhh, int M, float* Hnew,
               int64_t ldo, const SkgPick* pick, hipStream_t s, const SkgCell* lstm = nullptr) {
    const int Kp = ((In + 15) & ~15) + padded_k(H);
    const SkgCell cell = lstm ? *lstm : SkgCell{};
    const int n_ub = (H + 15) / 16, grid_x = ((n_ub + 7) / 8) * 8;
    const SkgPick pk = pick ? *pick : SkgPick{};
#define SKG_LAUNCH_C(MT, PICK, LSTM)                                                                                \
    do {                                                                                                            \
        const int lds = 16 * MT * Kp * 4 + 16, m_chunks = (M + 16 * MT - 1) / (16 * MT);                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(skinny_gru_kernel<MT, PICK, LSTM>),                 \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                 \
        hipLaunchKernelGGL((skinny_gru_kernel<MT, PICK, LSTM>), dim3(grid_x * m_chunks), dim3(SKG_THREADS), lds, s, \
                           X, ldx, In, Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, Kp, Hnew, ldo, n_ub,        \
                           m_chunks, pk, cell);                                                                     \
    } while (0)
#define SKG_LAUNCH(MT, PICK)                                                                                        \
    do {                                                                                                            \
        if (lstm) SKG_LAUNCH_C(MT, PICK, true); else SKG_LAUNCH_C(MT, PICK, false);                                 \
    } while (0)
    // The step is bound by the fp32 MFMA rate of the CUs it occupies, not by the weight stream (a few MB from the
    // Infinity Cache): one 16-row tile per workgroup spreads it over twice the CUs.  Two tiles per workgroup
    // (weights fetched once per 32 rows) only when that still fills the chip and the [x | h] image fits in LDS.
    const bool wide = (int64_t)grid_x * ((M + 31) / 32) >= sk_cus();
    if (wide && 32 * Kp * 4 + 16 + SKG_WAVES * 2 * 64 * 4 * 4 + 256 <= SK_LDS_BUDGET) {
        if (pick) SKG_LAUNCH(2, true); else SKG_LAUNCH(2, false);
    } else if (16 * Kp * 4 + 16 + SKG_WAVES * 64 * 4 * 4 + 256 <= SK_LDS_BUDGET) {
        if (pick) SKG_LAUNCH(1, true); else SKG_LAUNCH(1, false);
    } else {
        return DSSVAE_ETOOBIG;
    }
#undef SKG_LAUNCH
#undef SKG_LAUNCH_C
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

int skg_check(const float* Hprev, int64_t ldh, int H, const float* Wih, int64_t ldi, const float* Whh, int64_t ldhh,
              const float* bih, const float* bhh, int M, int In, float* Hnew, int64_t ldo) {
    if (!Hprev || !Wih || !Whh || !bih || !bhh || !Hnew || M <= 0 || In <= 0 || H <= 0 || ldh < H || ldi < In ||
        ldhh < H || ldo < H || Hnew == Hprev)
        return DSSVAE_EINVAL;
    if (M > DSSVAE_SKINNY_MAX_ROWS || (In % 4) || (H % 4) || (ldh % 4) || (ldi % 4) || (ldhh % 4) ||
        !aligned16(Hprev) || !aligned16(Wih) || !aligned16(Whh))
        return DSSVAE_ETOOBIG;
    return DSSVAE_OK;
}
}  // namespace

extern "C" int dssvae_skinny_gru_step(const float* X, int64_t ldx, int In, const float* Hprev, int64_t ldh, int H,
                                      const float* Wih, int64_t ldi, const float* Whh, int64_t ldhh,
                                      const float* bih, const float* bhh, int M, float* Hnew, int64_t ldo,
                                      void* stream) {
    if (!X || ldx < In) return DSSVAE_EINVAL;
    const int e = skg_check(Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, In, Hnew, ldo);
    if (e != DSSVAE_OK) return e;
    if ((ldx % 4) || !aligned16(X)) return DSSVAE_ETOOBIG;
    return skg_launch(X, ldx, In, Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, Hnew, ldo, nullptr,
                      (hipStream_t)stream);
}

extern "C" int dssvae_skinny_gru_step_pick(const float* part_max, const int64_t* part_arg, int n_split,
                                           int64_t* tokens_out, const float* table, int64_t ldt, int In,
                                           float* x_out, int64_t ldxo, const float* Hprev, int64_t ldh, int H,
                                           const float* Wih, int64_t ldi, const float* Whh, int64_t ldhh,
                                           const float* bih, const float* bhh, int M, float* Hnew, int64_t ldo,
                                           void* stream) {
    if (!part_max || !part_arg || n_split <= 0 || !tokens_out || !table || !x_out || ldt < In || ldxo < In)
        return DSSVAE_EINVAL;
    const int e = skg_check(Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, In, Hnew, ldo);
    if (e != DSSVAE_OK) return e;
    if ((ldt % 4) || (ldxo % 4) || !aligned16(table) || !aligned16(x_out)) return DSSVAE_ETOOBIG;
    SkgPick pk;
    pk.pm = part_max; pk.pa = part_arg; pk.S = n_split; pk.tok_out = tokens_out; pk.table = table; pk.ldt = ldt;
    pk.xout = x_out; pk.ldxo = ldxo;
    return skg_launch(nullptr, 0, In, Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, Hnew, ldo, &pk,
                      (hipStream_t)stream);
}

// ---- the same two entry points for the LSTM cell (weights [4H, .], gate order i, f, g, o) ------------------------
namespace {
int skl_check(const float* Cprev, float* Cnew, int64_t ldc, int H) {
    if (!Cprev || !Cnew || ldc < H) return DSSVAE_EINVAL;
    return DSSVAE_OK;
}
}  // namespace

extern "C" int dssvae_skinny_lstm_step(const float* X, int64_t ldx, int In, const float* Hprev, int64_t ldh, int H,
                                       const float* Cprev, float* Cnew, int64_t ldc, const float* Wih, int64_t ldi,
                                       const float* Whh, int64_t ldhh, const float* bih, const float* bhh, int M,
                                       float* Hnew, int64_t ldo, void* stream) {
    if (!X || ldx < In) return DSSVAE_EINVAL;
    int e = skg_check(Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, In, Hnew, ldo);
    if (e == DSSVAE_OK || e == DSSVAE_ETOOBIG) {
        const int e2 = skl_check(Cprev, Cnew, ldc, H);
        if (e2 != DSSVAE_OK) e = e2;
    }
    if (e != DSSVAE_OK) return e;
    if ((ldx % 4) || !aligned16(X)) return DSSVAE_ETOOBIG;
    const SkgCell cell{Cprev, Cnew, ldc};
    return skg_launch(X, ldx, In, Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, Hnew, ldo, nullptr,
                      (hipStream_t)stream, &cell);
}

extern "C" int dssvae_skinny_lstm_step_pick(const float* part_max, const int64_t* part_arg, int n_split,
                                            int64_t* tokens_out, const float* table, int64_t ldt, int In,
                                            float* x_out, int64_t ldxo, const float* Hprev, int64_t ldh, int H,
                                            const float* Cprev, float* Cnew, int64_t ldc, const float* Wih,
                                            int64_t ldi, const float* Whh, int64_t ldhh, const float* bih,
                                            const float* bhh, int M, float* Hnew, int64_t ldo, void* stream) {
    if (!part_max || !part_arg || n_split <= 0 || !tokens_out || !table || !x_out || ldt < In || ldxo < In)
        return DSSVAE_EINVAL;
    int e = skg_check(Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, In, Hnew, ldo);
    if (e == DSSVAE_OK || e == DSSVAE_ETOOBIG) {
        const int e2 = skl_check(Cprev, Cnew, ldc, H);
        if (e2 != DSSVAE_OK) e = e2;
    }
    if (e != DSSVAE_OK) return e;
    if ((ldt % 4) || (ldxo % 4) || !aligned16(table) || !aligned16(x_out)) return DSSVAE_ETOOBIG;
    SkgPick pk;
    pk.pm = part_max; pk.pa = part_arg; pk.S = n_split; pk.tok_out = tokens_out; pk.table = table; pk.ldt = ldt;
    pk.xout = x_out; pk.ldxo = ldxo;
    const SkgCell cell{Cprev, Cnew, ldc};
    return skg_launch(nullptr, 0, In, Hprev, ldh, H, Wih, ldi, Whh, ldhh, bih, bhh, M, Hnew, ldo, &pk,
                      (hipStream_t)stream, &cell);
}
