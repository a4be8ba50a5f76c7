// fp32 MFMA GEMM family for gfx950: one LDS-tiled mainloop (128x128x32 block, 4 waves, each
// wave a 2x2 grid of v_mfma_f32_32x32x2_f32 tiles), four epilogues:
//   STORE    C = alpha*A.B^T + beta*C + bias                      (x-projections, dgrad, wgrad)
//   LSE      online log-sum-exp over column splits, no logits out  (fused vocab projection + CE)
//   DLOGITS  coef[m]*exp(logit - lse[m]) into a row-chunk slab     (CE backward recompute)
// Roofline: fp32 MFMA (157 TFLOP/s dense).  LDS image is k-major ([k][m]) because the f32 MFMA
// operand map is lane -> (row = lane&31, k = lane>>5): a fragment read is 32 consecutive floats.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, NTHREADS = 256;
constexpr int LD_KC = 129;  // operand stored k-contiguous in memory: transposing ds_write_b32, conflict-free
constexpr int LD_RC = 132;  // operand stored row-contiguous in memory: ds_write_b128, 16-B aligned rows
constexpr float NEG_BIG = -3.0e38f;

struct OpA {
    const float* p1; int64_t ld1; int K1;
    const float* p2; int64_t ld2;
    const int64_t* rows;
    int vec;
};
struct OpB {
    const float* p; int64_t ld; int vec;
};

template <bool T> struct LdsLd { static constexpr int v = T ? LD_RC : LD_KC; };

// ---- global -> register staging --------------------------------------------------------------
// k-contiguous operand: thread t covers rows (t>>3)+32*i, i=0..3, at k = k0 + 4*(t&7)
struct RowBase {
    const float* b1[4];
    const float* b2[4];
    bool ok[4];
};

__device__ __forceinline__ void stage_kc(float4 (&reg)[4], const RowBase& rb, int K1, int K, int k0, int vec) {
    const int k = k0 + 4 * (threadIdx.x & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rb.ok[i]) {
            if (vec) {
                if (k < K) v = *reinterpret_cast<const float4*>(k < K1 ? rb.b1[i] + k : rb.b2[i] + (k - K1));
            } else {
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kk = k + j;
                    e[j] = kk < K ? (kk < K1 ? rb.b1[i][kk] : rb.b2[i][kk - K1]) : 0.f;
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
        }
        reg[i] = v;
    }
}

__device__ __forceinline__ void store_kc(float* lds, const float4 (&reg)[4]) {
    const int kq = 4 * (threadIdx.x & 7), r = threadIdx.x >> 3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* d = lds + kq * LD_KC + r + 32 * i;
        d[0] = reg[i].x; d[LD_KC] = reg[i].y; d[2 * LD_KC] = reg[i].z; d[3 * LD_KC] = reg[i].w;
    }
}

// row-contiguous operand (memory is [k][rows]): thread t covers k = k0 + (t>>5) + 8*i at rows 4*(t&31)..+3
__device__ __forceinline__ void stage_rc(float4 (&reg)[4], const float* p, int64_t ld, int R, int row0, int K,
                                         int k0, int vec) {
    const int r4 = row0 + 4 * (threadIdx.x & 31);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + (threadIdx.x >> 5) + 8 * i;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < K) {
            const float* s = p + (int64_t)k * ld + r4;
            if (vec && r4 + 3 < R) {
                v = *reinterpret_cast<const float4*>(s);
            } else {
                if (r4 < R) v.x = s[0];
                if (r4 + 1 < R) v.y = s[1];
                if (r4 + 2 < R) v.z = s[2];
                if (r4 + 3 < R) v.w = s[3];
            }
        }
        reg[i] = v;
    }
}

__device__ __forceinline__ void store_rc(float* lds, const float4 (&reg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(lds + ((threadIdx.x >> 5) + 8 * i) * LD_RC + 4 * (threadIdx.x & 31)) = reg[i];
}

// ---- one 128x128 output tile, full K --------------------------------------------------------
template <bool TA, bool TB>
__device__ __forceinline__ void tile_mainloop(const OpA& a, const OpB& b, int M, int N, int K, int m0, int n0,
                                              float* lds, f32x16 (&acc)[2][2]) {
    constexpr int LDA = LdsLd<TA>::v, LDB = LdsLd<TB>::v;
    float* As = lds;                       // [2][BK*LDA]
    float* Bs = lds + 2 * BK * LDA;        // [2][BK*LDB]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    RowBase ra, rbB;
    if (!TA) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gr = m0 + (threadIdx.x >> 3) + 32 * i;
            ra.ok[i] = gr < M;
            const int64_t ri = ra.ok[i] ? (a.rows ? a.rows[gr] : (int64_t)gr) : 0;
            ra.b1[i] = a.p1 + ri * a.ld1;
            ra.b2[i] = a.p2 ? a.p2 + ri * a.ld2 : a.p1;
        }
    }
    if (!TB) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gr = n0 + (threadIdx.x >> 3) + 32 * i;
            rbB.ok[i] = gr < N;
            rbB.b1[i] = b.p + (int64_t)(rbB.ok[i] ? gr : 0) * b.ld;
            rbB.b2[i] = rbB.b1[i];
        }
    }
    float4 rA[4], rB[4];
    const int nk = (K + BK - 1) / BK;
    auto stage = [&](int kt) {
        if (TA) stage_rc(rA, a.p1, a.ld1, M, m0, K, kt * BK, a.vec);
        else stage_kc(rA, ra, a.K1, K, kt * BK, a.vec);
        if (TB) stage_rc(rB, b.p, b.ld, N, n0, K, kt * BK, b.vec);
        else stage_kc(rB, rbB, K, K, kt * BK, b.vec);
    };
    auto commit = [&](int buf) {
        if (TA) store_rc(As + buf * BK * LDA, rA); else store_kc(As + buf * BK * LDA, rA);
        if (TB) store_rc(Bs + buf * BK * LDB, rB); else store_kc(Bs + buf * BK * LDB, rB);
    };
    stage(0);
    commit(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(kt + 1);
        const float* Ab = As + cur * BK * LDA + wr * 64 + (lane & 31);
        const float* Bb = Bs + cur * BK * LDB + wc * 64 + (lane & 31);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const int kidx = 2 * ks + (lane >> 5);
            const float a0 = Ab[kidx * LDA], a1 = Ab[kidx * LDA + 32];
            const float b0 = Bb[kidx * LDB], b1 = Bb[kidx * LDB + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) commit(cur ^ 1);
        __syncthreads();
    }
}

constexpr int lds_floats(bool ta, bool tb) { return 2 * BK * ((ta ? LD_RC : LD_KC) + (tb ? LD_RC : LD_KC)); }

// accumulator element (mt, nt, r) of this lane -> tile-local (row, col)
__device__ __forceinline__ int acc_row(int mt, int r) {
    const int lane = threadIdx.x & 63, wr = threadIdx.x >> 7;
    return wr * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int nt) {
    const int lane = threadIdx.x & 63, wc = (threadIdx.x >> 6) & 1;
    return wc * 64 + nt * 32 + (lane & 31);
}

// ---- STORE ------------------------------------------------------------------------------------
template <bool TA, bool TB>
__global__ __launch_bounds__(NTHREADS) void gemm_store_kernel(OpA a, OpB b, float* __restrict__ C, int64_t ldc,
                                                               const float* __restrict__ bias, int M, int N, int K,
                                                               float alpha, float beta, int tiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    f32x16 acc[2][2];
    tile_mainloop<TA, TB>(a, b, M, N, K, m0, n0, lds, acc);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = n0 + acc_col(nt);
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + acc_row(mt, r);
                if (m < M) {
                    float v = alpha * acc[mt][nt][r] + bv;
                    float* c = C + (int64_t)m * ldc + n;
                    if (beta != 0.f) v += beta * *c;
                    *c = v;
                }
            }
    }
}

// ---- LSE (optionally with argmax) ------------------------------------------------------------
template <bool ARG>
__global__ __launch_bounds__(NTHREADS) void linear_lse_kernel(OpA a, OpB b, const float* __restrict__ bias, int M,
                                                               int N, int K, int n_split, int tiles_per_split,
                                                               int tiles_n, float* __restrict__ part_max,
                                                               float* __restrict__ part_sum,
                                                               int64_t* __restrict__ part_arg) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int split = blockIdx.x % n_split, mb = blockIdx.x / n_split;
    const int m0 = mb * BM;
    const int t_begin = split * tiles_per_split;
    const int t_end = min(t_begin + tiles_per_split, tiles_n);
    float rm[2][16], rs[2][16];
    int ra[2][16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { rm[mt][r] = NEG_BIG; rs[mt][r] = 0.f; ra[mt][r] = 0; }

    for (int tn = t_begin; tn < t_end; ++tn) {
        const int n0 = tn * BN;
        f32x16 acc[2][2];
        tile_mainloop<false, false>(a, b, M, N, K, m0, n0, lds, acc);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + acc_col(nt);
            if (n < N) {
                const float bv = bias ? bias[n] : 0.f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float x = acc[mt][nt][r] + bv;
                        const float mo = rm[mt][r];
                        const float mn = fmaxf(mo, x);
                        rs[mt][r] = rs[mt][r] * __expf(mo - mn) + __expf(x - mn);
                        if (ARG) { if (x > mo) ra[mt][r] = n; }
                        rm[mt][r] = mn;
                    }
            }
        }
    }
    // reduce over the 32 lanes that hold different columns of the same rows
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float m = rm[mt][r], s = rs[mt][r];
            int ai = ra[mt][r];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {
                const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
                const int a2 = __shfl_xor(ai, o, 64);
                const float mn = fmaxf(m, m2);
                s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
                if (ARG) { if (m2 > m || (m2 == m && a2 < ai)) ai = a2; }
                m = mn;
            }
            rm[mt][r] = m; rs[mt][r] = s; ra[mt][r] = ai;
        }
    // combine the two waves that share rows (wc = 0/1) through LDS, then write the split partials
    float* red = lds;  // [2][BM][3]
    const int lane = threadIdx.x & 63, wc = (threadIdx.x >> 6) & 1;
    if ((lane & 31) == 0) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* d = red + (wc * BM + acc_row(mt, r)) * 3;
                d[0] = rm[mt][r]; d[1] = rs[mt][r]; d[2] = __int_as_float(ra[mt][r]);
            }
    }
    __syncthreads();
    if (threadIdx.x < BM) {
        const int m = m0 + threadIdx.x;
        if (m < M) {
            const float* x0 = red + threadIdx.x * 3;
            const float* x1 = red + (BM + threadIdx.x) * 3;
            const float mn = fmaxf(x0[0], x1[0]);
            const float s = x0[1] * __expf(x0[0] - mn) + x1[1] * __expf(x1[0] - mn);
            part_max[(int64_t)m * n_split + split] = mn;
            part_sum[(int64_t)m * n_split + split] = s;
            if (ARG) {
                const int a0 = __float_as_int(x0[2]), a1 = __float_as_int(x1[2]);
                part_arg[(int64_t)m * n_split + split] = (x1[0] > x0[0] || (x1[0] == x0[0] && a1 < a0)) ? a1 : a0;
            }
        }
    }
}

__global__ void lse_combine_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                   const int64_t* __restrict__ pa, int M, int S, float* __restrict__ lse,
                                   int64_t* __restrict__ amax) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float mx = NEG_BIG;
    int best = 0;
    for (int s = 0; s < S; ++s) {
        const float v = pm[(int64_t)m * S + s];
        if (v > mx) { mx = v; best = s; }
    }
    float acc = 0.f;
    for (int s = 0; s < S; ++s) acc += ps[(int64_t)m * S + s] * __expf(pm[(int64_t)m * S + s] - mx);
    if (lse) lse[m] = mx + __logf(acc);
    if (amax && pa) amax[m] = pa[(int64_t)m * S + best];
}

// ---- DLOGITS ---------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void linear_dlogits_kernel(OpA a, OpB b, const float* __restrict__ bias,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ coef, int M, int N, int K,
                                                                   float* __restrict__ dL, int64_t ldd, int tiles_n,
                                                                   int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    f32x16 acc[2][2];
    tile_mainloop<false, false>(a, b, M, N, K, m0, n0, lds, acc);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + acc_row(mt, r);
            if (m >= M) continue;
            const float l = lse[m], c = coef[m];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int n = n0 + acc_col(nt);
                if (n < N) dL[(int64_t)m * ldd + n] = c * __expf(acc[mt][nt][r] + (bias ? bias[n] : 0.f) - l);
            }
        }
}

// picked[m,t] = X[m,:] . W[target,:] + bias[target]; one wave per (m,t)
__global__ void picked_logit_kernel(const float* __restrict__ Xh, int64_t ldh, int Kh, const float* __restrict__ Xe,
                                    int64_t lde, int Ke, const float* __restrict__ W, int64_t ldw,
                                    const float* __restrict__ bias, const int64_t* __restrict__ target, int T,
                                    int64_t ignore, int64_t total, float* __restrict__ picked) {
    const int64_t idx = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= total) return;
    const int lane = threadIdx.x & 63;
    const int64_t m = idx / T;
    const int64_t tg = target[idx];
    if (tg == ignore) {
        if (lane == 0) picked[idx] = 0.f;
        return;
    }
    const float* w = W + tg * ldw;
    float s = 0.f;
    for (int k = lane; k < Kh; k += 64) s += Xh[m * ldh + k] * w[k];
    for (int k = lane; k < Ke; k += 64) s += Xe[m * lde + k] * w[Kh + k];
    s = wave_sum(s);
    if (lane == 0) picked[idx] = s + (bias ? bias[tg] : 0.f);
}

__global__ void onehot_sub_kernel(float* __restrict__ dL, int64_t ldd, const int64_t* __restrict__ target,
                                  int64_t ignore, const float* __restrict__ g, int64_t total, int T, int N) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int64_t tg = target[idx];
    if (tg == ignore || tg < 0 || tg >= N) return;
    atomicAdd(dL + (idx / T) * ldd + tg, -g[idx]);
}

// out[n] (+)= sum_m X[m,n]; block = 64 columns x 4 row slices
__global__ void colsum_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, float* __restrict__ out,
                              int accumulate) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rs = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float s = 0.f;
    if (n < N)
        for (int m = rs; m < M; m += 4) s += X[(int64_t)m * ldx + n];
    red[rs][c] = s;
    __syncthreads();
    if (rs == 0 && n < N) {
        const float v = red[0][c] + red[1][c] + red[2][c] + red[3][c];
        out[n] = accumulate ? out[n] + v : v;
    }
}

template <typename F>
void allow_lds(F* func, int bytes) {
    // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel (gfx950 has 160 KiB per CU)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

bool vec_ok(const float* p, int64_t ld) { return p == nullptr || (aligned16(p) && (ld % 4) == 0); }

OpA make_panels(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke) {
    OpA a;
    a.p1 = Xh; a.ld1 = ldh; a.K1 = Kh; a.p2 = Ke > 0 ? Xe : nullptr; a.ld2 = lde; a.rows = nullptr;
    a.vec = vec_ok(Xh, ldh) && (Ke == 0 || vec_ok(Xe, lde)) && (Kh % 4 == 0) && (Ke % 4 == 0);
    return a;
}

}  // namespace

extern "C" int dssvae_gemm_f32(const float* A, int64_t lda, int trans_a, const int64_t* a_rows, const float* B,
                               int64_t ldb, int trans_b, float* C, int64_t ldc, const float* bias, int M, int N,
                               int K, float alpha, float beta, void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K <= 0) return DSSVAE_EINVAL;
    if (trans_a && a_rows) return DSSVAE_EINVAL;
    if (M == 0 || N == 0) return DSSVAE_OK;
    OpA a;
    a.p1 = A; a.ld1 = lda; a.K1 = K; a.p2 = nullptr; a.ld2 = 0; a.rows = a_rows;
    a.vec = vec_ok(A, lda) && (trans_a ? true : (K % 4 == 0));
    OpB b;
    b.p = B; b.ld = ldb;
    b.vec = vec_ok(B, ldb) && (trans_b ? true : (K % 4 == 0));
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(TA, TB)                                                                                            \
    do {                                                                                                          \
        allow_lds(gemm_store_kernel<TA, TB>, lds_floats(TA, TB) * 4);                                             \
        hipLaunchKernelGGL((gemm_store_kernel<TA, TB>), dim3(ntiles), dim3(NTHREADS), lds_floats(TA, TB) * 4, s, \
                           a, b, C, ldc, bias, M, N, K, alpha, beta, tiles_n, ntiles);                            \
    } while (0)
    if (trans_a) { if (trans_b) LAUNCH(true, true); else LAUNCH(true, false); }
    else { if (trans_b) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_linear_lse_fwd(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                                     const float* W, int64_t ldw, const float* bias, int M, int N, int n_split,
                                     float* part_max, float* part_sum, int64_t* part_arg, int want_argmax,
                                     void* stream) {
    if (!Xh || !W || !part_max || !part_sum || M <= 0 || N <= 0 || Kh <= 0 || Ke < 0 || n_split <= 0)
        return DSSVAE_EINVAL;
    if (Ke > 0 && !Xe) return DSSVAE_EINVAL;
    if (want_argmax && !part_arg) return DSSVAE_EINVAL;
    OpA a = make_panels(Xh, ldh, Kh, Xe, lde, Ke);
    OpB b;
    b.p = W; b.ld = ldw; b.vec = vec_ok(W, ldw) && ((Kh + Ke) % 4 == 0);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int tps = (tiles_n + n_split - 1) / n_split;
    hipStream_t s = (hipStream_t)stream;
    allow_lds(linear_lse_kernel<true>, lds_floats(false, false) * 4);
    allow_lds(linear_lse_kernel<false>, lds_floats(false, false) * 4);
    if (want_argmax)
        hipLaunchKernelGGL((linear_lse_kernel<true>), dim3(tiles_m * n_split), dim3(NTHREADS),
                           lds_floats(false, false) * 4, s, a, b, bias, M, N, Kh + Ke, n_split, tps, tiles_n,
                           part_max, part_sum, part_arg);
    else
        hipLaunchKernelGGL((linear_lse_kernel<false>), dim3(tiles_m * n_split), dim3(NTHREADS),
                           lds_floats(false, false) * 4, s, a, b, bias, M, N, Kh + Ke, n_split, tps, tiles_n,
                           part_max, part_sum, part_arg);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_lse_combine(const float* part_max, const float* part_sum, const int64_t* part_arg, int M,
                                  int n_split, float* lse, int64_t* argmax, void* stream) {
    if (!part_max || !part_sum || M <= 0 || n_split <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(lse_combine_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, part_max,
                       part_sum, part_arg, M, n_split, lse, argmax);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_picked_logit(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                                   const float* W, int64_t ldw, const float* bias, const int64_t* target, int T,
                                   int64_t ignore_index, int M, float* picked, void* stream) {
    if (!Xh || !W || !target || !picked || M <= 0 || T <= 0 || Kh <= 0 || (Ke > 0 && !Xe)) return DSSVAE_EINVAL;
    const int64_t total = (int64_t)M * T;
    hipLaunchKernelGGL(picked_logit_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, Xh,
                       ldh, Kh, Xe, lde, Ke, W, ldw, bias, target, T, ignore_index, total, picked);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_linear_ce_dlogits(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                                        const float* W, int64_t ldw, const float* bias, const float* lse,
                                        const float* coef, int M, int N, float* dL, int64_t ldd, void* stream) {
    if (!Xh || !W || !lse || !coef || !dL || M <= 0 || N <= 0 || Kh <= 0 || ldd < N || (Ke > 0 && !Xe))
        return DSSVAE_EINVAL;
    OpA a = make_panels(Xh, ldh, Kh, Xe, lde, Ke);
    OpB b;
    b.p = W; b.ld = ldw; b.vec = vec_ok(W, ldw) && ((Kh + Ke) % 4 == 0);
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    allow_lds(linear_dlogits_kernel, lds_floats(false, false) * 4);
    hipLaunchKernelGGL(linear_dlogits_kernel, dim3(ntiles), dim3(NTHREADS), lds_floats(false, false) * 4,
                       (hipStream_t)stream, a, b, bias, lse, coef, M, N, Kh + Ke, dL, ldd, tiles_n, ntiles);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_onehot_sub(float* dL, int64_t ldd, const int64_t* target, int T, int64_t ignore_index,
                                 const float* g, int M, int N, void* stream) {
    if (!dL || !target || !g || M <= 0 || T <= 0) return DSSVAE_EINVAL;
    const int64_t total = (int64_t)M * T;
    hipLaunchKernelGGL(onehot_sub_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dL, ldd, target, ignore_index, g, total, T, N);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, void* stream) {
    if (!X || !out || M <= 0 || N <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, (hipStream_t)stream, X, ldx, M, N, out,
                       accumulate);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}
