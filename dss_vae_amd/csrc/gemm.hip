// fp32 MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 157 TFLOP/s dense peak).
//
// One LDS-tiled mainloop, (64*WM) x (64*WN) x 32 block, 4 waves as 2x2, each wave a WM x WN grid of
// 32x32 MFMA tiles, double-buffered LDS with register staging (loads of k-tile t+1 are in flight
// while tile t is multiplied; one barrier per k-tile).  Three epilogues:
//   STORE    C = alpha*A.B^T + beta*C + bias, optional split-K (atomic accumulate)
//   LSE      online log-sum-exp (+argmax) over column splits, logits never written
//   DLOGITS  coef[m]*exp(logit - lse[m]) into a slab, + column sums for the bias gradient
//
// Operand images in LDS.  The f32 MFMA takes ONE float per lane per operand: lane l supplies
// A[row l&31][k-slot l>>5].  Which k a (step, lane-half) pair multiplies is free as long as A and
// B agree; here k(g, j, h) = 8g + 4h + j  (g: group of 8 k, j: MFMA within the group, h = l>>5).
//   * k-contiguous operand (row-major [rows, K]: activations, nn.Linear weights): image
//     [rows][36]; global float4 -> ds_write_b128 as is (no transposing stores), and a lane's four
//     j-values of a group are ONE ds_read_b128 (row stride 36 floats: conflict-free for every
//     16-lane ds_read_b128 group).
//   * row-contiguous operand (memory [K, rows]: the transposed operands of dgrad / wgrad): image
//     [32][rows+4]; float4 along rows in, ds_read_b32 out (32 consecutive floats per lane-half).
// The fast path (VEC) has no branches in the mainloop: out-of-range rows read a clamped valid
// row (their results are never stored), the k tail is zeroed by select.  Shapes whose leading
// dimensions / K are not multiples of 4 floats take the scalar-load instantiation (same math).
#include "common.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <stdlib.h>

namespace {

#ifndef GEMM_BK
#define GEMM_BK 32
#endif
constexpr int BK = GEMM_BK, NTHREADS = 256;
constexpr int LDK = BK + 4;
constexpr float NEG_BIG = -3.0e38f;

struct OpA {
    const float* p1; int64_t ld1; int K1;
    const float* p2; int64_t ld2;
    const int64_t* rows;
    // XF stagers only: the operand in memory is a LOGITS slab [slab rows, vocabulary columns]; what enters the
    // LDS image is the softmax part of dlogits, coef * exp(logit - lse) = exp2(logit * log2(e) - l2), computed
    // while the tile is staged from ONE float per slab row: l2 = (lse - ln coef) * log2(e)  (+inf where coef = 0).
    // The one-hot part of dlogits is a rank-one-per-row correction of the products (ce_onehot_fixup_kernel).
    const float* xf;   // XF: per (128-column tile, row) scale of the normalised slab, [tile][xf_ld]
    int64_t xf_ld;
};
struct OpB {
    const float* p; int64_t ld;
};

constexpr int img_floats(bool trans, int R) { return trans ? BK * (R + 4) : R * LDK; }
template <bool TA, bool TB, int WM, int WN>
constexpr int lds_floats() { return 2 * (img_floats(TA, 64 * WM) + img_floats(TB, 64 * WN)); }

// ---- global -> register -> LDS staging of one operand tile (R rows x 32 k) -----------------------
// VW = vector width of the global loads / LDS stores: 4 (16-byte granules), 2 (8-byte granules: leading
// dimensions and extents that are even but not multiples of 4 -- the 3H = 450 gate rows of the default
// encoder) or 1 (scalar, any shape).  A granule is entirely in or out of range by the host's choice of VW.
template <int VW> struct VecT;
template <> struct VecT<4> { typedef float4 T; static __device__ __forceinline__ T zero() { return make_float4(0.f, 0.f, 0.f, 0.f); } };
template <> struct VecT<2> { typedef float2 T; static __device__ __forceinline__ T zero() { return make_float2(0.f, 0.f); } };
template <> struct VecT<1> { typedef float T; static __device__ __forceinline__ T zero() { return 0.f; } };

template <bool TRANS, int VW, int R, bool XF = false>
struct Stager {
    static_assert(!XF || VW == 4, "the dlogits transform is written for 16-byte granules");
    static constexpr bool MASK_FREE = false;
    static constexpr bool VEC = VW > 1;
    typedef typename VecT<VW>::T V;
    static constexpr int NV = R * BK / (NTHREADS * (VEC ? VW : 4));  // granules per thread (VEC)
    static constexpr int NS = R * BK / 256;   // floats per thread (scalar)
    static constexpr int TPR = BK / (VEC ? VW : 4);   // k-contiguous image: threads per row (VEC)
    static constexpr int RPP = NTHREADS / TPR;   // rows per pass (VEC)
    static constexpr int RPS = NTHREADS / BK;    // rows per pass (scalar)
    V v[VEC ? NV : 1];
    float s[VEC ? 1 : NS];
    float ri[XF ? NV : 1];    // l2 of the slab rows of the staged granules (XF)
    const float* xf;
    bool xf_vec;              // xf is 16-byte aligned
    const float* b1[(!TRANS && VEC) ? NV : 1];
    const float* b2[(!TRANS && VEC) ? NV : 1];
    const float* p1; const float* p2; const int64_t* rows;
    int64_t ld1, ld2;
    int K1, K, Rtot, r0;
    unsigned okmask;  // which staged registers hold in-range k (the k tail is zeroed when written to LDS)

    __device__ __forceinline__ void init(const float* p1_, int64_t ld1_, int K1_, const float* p2_, int64_t ld2_,
                                         const int64_t* rows_, int Rtot_, int r0_, int K_, const float* xf_ = nullptr,
                                         int64_t = 0) {
        xf = xf_;
        xf_vec = (reinterpret_cast<uintptr_t>(xf_) & 15u) == 0;
        p1 = p1_; ld1 = ld1_; K1 = K1_; p2 = p2_ ? p2_ : p1_; ld2 = p2_ ? ld2_ : ld1_; rows = rows_;
        Rtot = Rtot_; r0 = r0_; K = K_;
        if (!TRANS && VEC) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = min(r0 + (int)(threadIdx.x / TPR) + RPP * i, Rtot - 1);
                const int64_t ri_ = rows ? rows[r] : (int64_t)r;
                b1[i] = p1 + ri_ * ld1;
                b2[i] = p2 + ri_ * ld2;
                if (XF) ri[i] = xf[r];   // the slab rows of this thread's granules never change
            }
        }
    }

    // TRANS, VEC: k row (inside the tile) of this thread's granule i.  XF: 4 consecutive rows per thread (their row
    // records are one 16-byte load); else rows KSTEP apart (neighbouring threads of a pass share a row either way)
    static __device__ __forceinline__ int krow(int i) {
        constexpr int TPK = R / (VEC ? VW : 1), KSTEP = NTHREADS / TPK;
        return XF ? NV * (int)(threadIdx.x / TPK) + i : (int)(threadIdx.x / TPK) + KSTEP * i;
    }

    // softmax part of dlogits for one granule of a slab row.  The slab holds exp(x - max of the row in its 128-column
    // tile) (the LSE kernel wrote it so) and `s` = coef * exp(that max - lse): one multiply per element, where raw
    // logits cost a v_fma and a v_exp each -- and every VALU instruction between MFMAs costs matrix time
    static __device__ __forceinline__ float4 dlogits(float4 x, float s) {
        float4 o;
        o.x = x.x * s; o.y = x.y * s; o.z = x.z * s; o.w = x.w * s;
        return o;
    }

    // The tile's loads and LDS stores are cut into 4 parts so that the mainloop can spread them between
    // the MFMAs of a k-group (one part per MFMA sub-step): item i belongs to part i*4/N.
    static constexpr int NITEM = VEC ? NV : NS;
    static constexpr __host__ __device__ bool in_part(int i, int part) { return (i * 4) / NITEM == part; }
    int loff;             // !TRANS: column offset inside the chosen panel
    bool lfirst;
    int lk0;

    // per-tile scalars of the load (k range check, panel choice); resets okmask
    __device__ __forceinline__ void begin_load(int k0) {
#ifdef GEMM_DEV_SAMEK    // development build: every k-tile loads the first one again (cache hits; results are garbage)
        k0 = 0;
#endif
        if (!TRANS) {
            const int k = k0 + (VEC ? VW * (int)(threadIdx.x % TPR) : (int)(threadIdx.x % BK));
            const bool ok = k < K;
            const int kc = ok ? k : 0;
            lfirst = kc < K1;
            loff = lfirst ? kc : kc - K1;
            okmask = ok ? ~0u : 0u;
        } else {
            okmask = 0u;
        }
        lk0 = k0;
    }

    __device__ __forceinline__ void load_part(int part) {
#ifdef GEMM_DEV_NOLOAD   // development build: the mainloop without its global loads (results are garbage)
        return;
#endif
        if (!TRANS) {
            if (VEC) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (in_part(i, part)) v[i] = *reinterpret_cast<const V*>((lfirst ? b1[i] : b2[i]) + loff);
            } else {
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    if (in_part(e, part)) {
                        const int r = min(r0 + (int)(threadIdx.x / BK) + RPS * e, Rtot - 1);
                        const int64_t ri = rows ? rows[r] : (int64_t)r;
                        s[e] = lfirst ? p1[ri * ld1 + loff] : p2[ri * ld2 + loff];
                    }
            }
        } else {
            if (VEC) {
                constexpr int TPK = R / VW;          // threads per k row
                constexpr int KSTEP = NTHREADS / TPK;
                const int rv = min(r0 + VW * (int)(threadIdx.x % TPK), Rtot - VW);
                if constexpr (XF) {
                    // the slab rows' l2 values: this thread's NV granules sit in CONSECUTIVE slab rows (krow), so
                    // they are one 16-byte load per k-tile instead of one dword load per granule
                    static_assert(NV == 4 && KSTEP * NV == BK, "row record load assumes 4 consecutive rows per thread");
                    if (part == 0) {
                        const int kb = lk0 + NV * (int)(threadIdx.x / TPK);
                        if (kb + NV <= K && xf_vec) {
                            const float4 q = *reinterpret_cast<const float4*>(xf + kb);
                            ri[0] = q.x; ri[1] = q.y; ri[2] = q.z; ri[3] = q.w;
                        } else {
#pragma unroll
                            for (int i = 0; i < NV; ++i) ri[i] = xf[min(kb + i, K - 1)];
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (in_part(i, part)) {
                        const int k = lk0 + krow(i);
                        const bool ok = k < K;
                        okmask |= ok ? (1u << i) : 0u;
                        v[i] = *reinterpret_cast<const V*>(p1 + (int64_t)(ok ? k : 0) * ld1 + rv);
                    }
            } else {
                constexpr int KSTEP = NTHREADS / R;  // R = 64 -> 4, R = 128 -> 2
                const int r = min(r0 + (int)(threadIdx.x % R), Rtot - 1);
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    if (in_part(e, part)) {
                        const int k = lk0 + (int)(threadIdx.x / R) + KSTEP * e;
                        const bool ok = k < K;
                        okmask |= ok ? (1u << e) : 0u;
                        s[e] = p1[(int64_t)(ok ? k : 0) * ld1 + r];
                    }
            }
        }
    }

    __device__ __forceinline__ void load(int k0) {
        begin_load(k0);
#pragma unroll
        for (int part = 0; part < 4; ++part) load_part(part);
    }
    // (the interface of FastStager: this stager masks every tile the same way)
    template <bool TAIL> __device__ __forceinline__ void load_part(int part) { load_part(part); }
    template <bool TAIL> __device__ __forceinline__ void store_part(float* img, int part) const { store_part(img, part); }

    // XF: the staged granules of one part become dlogits in place (before store_part writes them to LDS)
    __device__ __forceinline__ void xform_part(int part) {
        if constexpr (XF) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_part(i, part)) v[i] = dlogits(v[i], ri[i]);
        }
    }

    __device__ __forceinline__ void store_part(float* img, int part) const {
#ifdef GEMM_DEV_NOSTORE  // development build: the mainloop without its LDS staging stores (results are garbage)
#ifdef GEMM_DEV_KEEPLOAD  // ... but with the global loads kept alive
        if (VEC) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_part(i, part)) asm volatile("" ::"v"(v[i]));
        }
#endif
        return;
#endif
        if (!TRANS) {
            if (VEC) {
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (in_part(i, part)) {
                        const V val = v[i];
                        *reinterpret_cast<V*>(img + ((threadIdx.x / TPR) + RPP * i) * LDK + VW * (threadIdx.x % TPR)) =
                            okmask ? val : VecT<VW>::zero();
                    }
            } else {
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    if (in_part(e, part))
                        img[((threadIdx.x / BK) + RPS * e) * LDK + (threadIdx.x % BK)] = okmask ? s[e] : 0.f;
            }
        } else {
            if (VEC) {
                constexpr int TPK = R / VW;
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (in_part(i, part)) {
                        const V val = v[i];
                        *reinterpret_cast<V*>(img + krow(i) * (R + 4) + VW * (threadIdx.x % TPK)) =
                            ((okmask >> i) & 1u) ? val : VecT<VW>::zero();
                    }
            } else {
                constexpr int KSTEP = NTHREADS / R;
#pragma unroll
                for (int e = 0; e < NS; ++e)
                    if (in_part(e, part))
                        img[((threadIdx.x / R) + KSTEP * e) * (R + 4) + (threadIdx.x % R)] = ((okmask >> e) & 1u) ? s[e] : 0.f;
            }
        }
    }

    __device__ __forceinline__ void store(float* img) const {
#pragma unroll
        for (int part = 0; part < 4; ++part) store_part(img, part);
    }
};

// ---- the same for 16-byte granules (VW = 4), with NO vector-ALU work per k-tile ---------------------------------
// v_mfma_f32_32x32x2_f32 runs on the fp32 vector ALUs (its rate IS the fp32 vector rate), so every VALU instruction
// any wave of the SIMD issues takes its cycles away from the MFMAs: measured with tools/probes/mfma_valu.hip, a
// v_add_f32 / v_cndmask beside a bare MFMA loop costs ~5 cycles of matrix time at one wave per SIMD (~3 at two), a
// v_exp_f32 ~9 -- against 64 cycles per MFMA.  The generic stager above spends ~60 VALU instructions per k-tile on
// addresses (64-bit pointer arithmetic per granule), panel selection and the k-tail select: the same mainloop without
// its loads runs at 150 TFLOP/s, with them at 124 (4096^3; tools/gemm_variants.py).  Here the per-tile part of every
// address is UNIFORM -- the tile's first k, times the leading dimension for a transposed operand -- and lives in
// scalar registers; a thread keeps one 32-bit byte offset per granule for the whole tile loop (global_load with an
// SGPR base), the LDS addresses are a per-thread constant plus immediates, and the k-tail costs a scalar branch
// that only the last k-tile of a product takes.  Host-checked (fast_ok): one panel, no row gather, offsets < 4 GiB.
template <bool TRANS, int R, bool XF = false>
struct FastStager {
    static constexpr int NV = R * BK / (NTHREADS * 4);   // 16-byte granules per thread (R = 128: 4, R = 64: 2)
    static constexpr int TPR = BK / 4, RPP = NTHREADS / TPR;      // k-contiguous image: 8 threads per row, 32 rows per pass
    static constexpr int TPK = R / 4, KSTEP = NTHREADS / TPK;     // transposed image: threads per k row, k rows per pass
    static constexpr bool MASK_FREE = true;   // load_part<false> / store_part<false> keep no per-tile state
    static_assert(!XF || (NV == 4 && (!TRANS || KSTEP * NV == BK)), "the dlogits stagers are written for 128-row tiles");
    float4 v[NV];
    float ri[XF ? NV : 1];    // scale of the staged granules' (slab row, column tile) (XF)
    uint32_t rowi[(XF && !TRANS) ? NV : 1];   // slab row of granule i (XF, not transposed: fixed for the tile loop)
    int64_t xf_ld;
    uint32_t off[NV];         // byte offset of granule i from the tile's base address
    uint32_t st;              // float offset of granule 0 in the LDS image
    const float* p; const float* xf;
    int64_t ld;
    int K, lk0;
    bool xf_vec;              // uniform
    unsigned okmask;          // set by load_part<true>: which granules hold in-range k

    static __device__ __forceinline__ int krow(int i) {
        return XF ? NV * (int)(threadIdx.x / TPK) + i : (int)(threadIdx.x / TPK) + KSTEP * i;
    }
    static constexpr __host__ __device__ bool in_part(int i, int part) { return (i * 4) / NV == part; }

    __device__ __forceinline__ void init(const float* p1_, int64_t ld1_, int, const float*, int64_t, const int64_t*,
                                         int Rtot, int r0, int K_, const float* xf_ = nullptr, int64_t xf_ld_ = 0) {
        p = p1_; ld = ld1_; K = K_; xf = xf_; okmask = ~0u; lk0 = 0; xf_ld = xf_ld_;
        // transposed (wgrad): the tile's rows ARE one 128-column tile of the slab; its scales are a vector over k
        if (XF && TRANS) xf = xf_ + (int64_t)(r0 >> 7) * xf_ld_;
        xf_vec = (reinterpret_cast<uintptr_t>(xf) & 15u) == 0;
        if (!TRANS) {
            st = (threadIdx.x / TPR) * LDK + 4 * (threadIdx.x % TPR);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int r = min(r0 + (int)(threadIdx.x / TPR) + RPP * i, Rtot - 1);
                off[i] = (uint32_t)(((int64_t)r * ld + 4 * (int)(threadIdx.x % TPR)) * 4);
                if constexpr (XF) rowi[i] = (uint32_t)r;   // the slab rows of this thread's granules never change
            }
        } else {
            const int rv = min(r0 + 4 * (int)(threadIdx.x % TPK), Rtot - 4);
            st = krow(0) * (R + 4) + 4 * (threadIdx.x % TPK);
#pragma unroll
            for (int i = 0; i < NV; ++i) off[i] = (uint32_t)(((int64_t)krow(i) * ld + rv) * 4);
        }
    }

    __device__ __forceinline__ void begin_load(int k0) { lk0 = k0; }

    // TAIL = false: the tile is entirely in range (k0 + 32 <= K): scalar base + per-thread 32-bit offsets, nothing else.
    // TAIL = true: any tile; out-of-range granules read an address that is certainly valid (k = 0 of their row / k row
    // 0 of their column) and are zeroed when they are written to LDS.  Only the last three k-tiles of a product whose
    // K is not a multiple of 32 (and the prologue) run the TAIL code.
    template <bool TAIL>
    __device__ __forceinline__ void load_part(int part) {
#ifdef GEMM_DEV_NOLOAD
        return;
#endif
        const char* base = reinterpret_cast<const char*>(TRANS ? p + (int64_t)lk0 * ld : p + lk0);
        if constexpr (!TAIL) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_part(i, part)) v[i] = *reinterpret_cast<const float4*>(base + off[i]);
            if constexpr (XF && !TRANS) {
                if (part == 3) {   // the scales of this k-tile's column tile (k = slab column; 32 | 128)
                    const float* t = xf + (int64_t)(lk0 >> 7) * xf_ld;
#pragma unroll
                    for (int i = 0; i < NV; ++i) ri[i] = t[rowi[i]];
                }
            }
            if constexpr (XF && TRANS) {
                if (part == 3) {   // (behind the last transform of the tile whose records these registers still hold)
                    if (xf_vec) {
                        const float4 q = *reinterpret_cast<const float4*>(
                            reinterpret_cast<const char*>(xf + lk0) + 16u * (uint32_t)(threadIdx.x / TPK));
                        ri[0] = q.x; ri[1] = q.y; ri[2] = q.z; ri[3] = q.w;
                    } else {
#pragma unroll
                        for (int i = 0; i < NV; ++i) ri[i] = xf[lk0 + krow(i)];
                    }
                }
            }
        } else if constexpr (!TRANS) {
            const uint32_t kc = 4 * (threadIdx.x % TPR);
            const bool ok = lk0 + (int)kc < K;
            okmask = ok ? ~0u : 0u;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_part(i, part))
                    v[i] = *reinterpret_cast<const float4*>(ok ? base + off[i] : reinterpret_cast<const char*>(p) + (off[i] - 4 * kc));
            if constexpr (XF) {
                if (part == 0) {
                    const float* t = xf + (int64_t)(min(lk0, K - 1) >> 7) * xf_ld;
#pragma unroll
                    for (int i = 0; i < NV; ++i) ri[i] = t[rowi[i]];
                }
            }
        } else {
            if (part == 0) okmask = 0u;
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_part(i, part)) {
                    const bool ok = lk0 + krow(i) < K;
                    okmask |= ok ? (1u << i) : 0u;
                    v[i] = *reinterpret_cast<const float4*>(
                        ok ? base + off[i] : reinterpret_cast<const char*>(p) + (off[i] - (uint32_t)((int64_t)krow(i) * ld * 4)));
                }
            if constexpr (XF) {
                if (part == 0) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) ri[i] = xf[min(lk0 + krow(i), K - 1)];
                }
            }
        }
    }

    __device__ __forceinline__ void load(int k0) {   // (prologue: any tile)
        begin_load(k0);
#pragma unroll
        for (int part = 0; part < 4; ++part) load_part<true>(part);
    }

    __device__ __forceinline__ void xform_part(int part) {
        if constexpr (XF) {
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (in_part(i, part)) v[i] = Stager<TRANS, 4, R, true>::dlogits(v[i], ri[i]);
        }
    }

    // TAIL: the registers were filled by load_part<true> (okmask is valid)
    template <bool TAIL>
    __device__ __forceinline__ void store_part(float* img, int part) const {
#ifdef GEMM_DEV_NOSTORE
        return;
#endif
        constexpr int ISTRIDE = !TRANS ? RPP * LDK : (XF ? (R + 4) : KSTEP * (R + 4));
#pragma unroll
        for (int i = 0; i < NV; ++i)
            if (in_part(i, part)) {
                if constexpr (!TAIL) {
                    *reinterpret_cast<float4*>(img + st + i * ISTRIDE) = v[i];
                } else {
                    *reinterpret_cast<float4*>(img + st + i * ISTRIDE) =
                        ((okmask >> (TRANS ? i : 0)) & 1u) ? v[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
    }

    __device__ __forceinline__ void store(float* img) const {   // (prologue)
#pragma unroll
        for (int part = 0; part < 4; ++part) store_part<true>(img, part);
    }
};

template <bool TRANS, int VW, int R, bool XF> struct StagerOf { typedef Stager<TRANS, VW, R, XF> T; };
template <bool TRANS, int R, bool XF> struct StagerOf<TRANS, 4, R, XF> { typedef FastStager<TRANS, R, XF> T; };

// fragment of one 32-row MFMA tile for group g: f[j] multiplies k = 8g + 4h + j
template <bool TRANS, int R>
__device__ __forceinline__ void read_frag(float (&f)[4], const float* img, int row, int g, int h) {
    if (!TRANS) {
        const float4 t = *reinterpret_cast<const float4*>(img + row * LDK + 8 * g + 4 * h);
        f[0] = t.x; f[1] = t.y; f[2] = t.z; f[3] = t.w;
    } else {
        const float* p = img + (8 * g + 4 * h) * (R + 4) + row;
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = p[j * (R + 4)];
    }
}

// ---- one k-tile of the rotated software pipeline; CUR (which LDS buffer holds the tile) is a template parameter so
// that every LDS address of the tile is a per-thread constant plus an immediate ------------------------------------
template <bool TA, bool TB, int WM, int WN, bool XF, int CUR, bool TAIL, typename SA, typename SB>
__device__ __forceinline__ void k_tile(SA& sa, SB& sb, float* As, float* Bs, int arow, int brow, int h, bool more, int k2,
                                       float (&fa)[2][WM][4], float (&fb)[2][WN][4], f32x16 (&acc)[WM][WN]) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int AF = img_floats(TA, BM), BF = img_floats(TB, BN);
    constexpr int NG = BK / 8;
    // The loads of tile kt + 2 overwrite the registers the stores of tile kt + 1 read.  The mask-free stagers load part jj
    // right behind its store, in the same sub-step of group 0: a group of MFMAs more between a load and its use than
    // from group 1, where the masking code has to wait (its per-tile mask is shared by the four parts).
    constexpr int G_STORE = 0, G_LOAD = (!TAIL && SA::MASK_FREE && SB::MASK_FREE) ? 0 : 1;
    const float* Ab = As + CUR * AF;
    const float* Bb = Bs + CUR * BF;
    float* An = As + (CUR ^ 1) * AF;
    float* Bn = Bs + (CUR ^ 1) * BF;
#pragma unroll
    for (int g = 0; g < NG - 1; ++g) {
        const int c = g & 1, n = c ^ 1;
#pragma unroll
        for (int i = 0; i < WM; ++i) read_frag<TA, BM>(fa[n][i], Ab, arow + 32 * i, g + 1, h);
#pragma unroll
        for (int j = 0; j < WN; ++j) read_frag<TB, BN>(fb[n][j], Bb, brow + 32 * j, g + 1, h);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][jj], fb[c][j][jj], acc[i][j], 0, 0, 0);
            if (g == G_STORE) {
                sa.xform_part(jj);
                sa.template store_part<TAIL>(An, jj);
                sb.template store_part<TAIL>(Bn, jj);
                if constexpr (XF) {
                    // one granule's transform (4 multiplies) rides between this sub-step's MFMAs instead of
                    // queueing behind the last one
#pragma unroll
                    for (int q = 0; q < WM * WN; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, (4 + WM * WN - 1) / (WM * WN), 0);  // VALU
                    }
                }
            }
            if (g == G_LOAD) {
                if (jj == 0) { sa.begin_load(k2); sb.begin_load(k2); }
                sa.template load_part<TAIL>(jj);
                sb.template load_part<TAIL>(jj);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the sub-steps apart: MFMAs, then their share of staging
        }
    }
#ifndef GEMM_DEV_NOBARRIER  // development build: no workgroup barrier per k-tile (racy; results are garbage)
    __syncthreads();
#endif
    constexpr int cl = (NG - 1) & 1, nl = cl ^ 1;  // NG is even: the next tile's group 0 lands in slot 0
    static_assert(nl == 0, "fragment slot rotation assumes an even number of groups per k-tile");
    if (more) {
#pragma unroll
        for (int i = 0; i < WM; ++i) read_frag<TA, BM>(fa[nl][i], An, arow + 32 * i, 0, h);
#pragma unroll
        for (int j = 0; j < WN; ++j) read_frag<TB, BN>(fb[nl][j], Bn, brow + 32 * j, 0, h);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cl][i][jj], fb[cl][j][jj], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- one (64*WM) x (64*WN) output tile over k-tiles [kt0, kt1) -----------------------------------
template <bool TA, bool TB, int VW, int WM, int WN, bool XF = false>
__device__ __forceinline__ void tile_mainloop(const OpA& a, const OpB& b, int M, int N, int K, int m0, int n0,
                                              int kt0, int kt1, float* lds, f32x16 (&acc)[WM][WN]) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int AF = img_floats(TA, BM);
    float* As = lds;            // [2][AF]
    float* Bs = lds + 2 * AF;   // [2][BF]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, h = lane >> 5;
    const int arow = wr * 32 * WM + (lane & 31), brow = wc * 32 * WN + (lane & 31);
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename StagerOf<TA, VW, BM, XF>::T sa;
    typename StagerOf<TB, VW, BN, false>::T sb;
    sa.init(a.p1, a.ld1, a.K1, a.p2, a.ld2, a.rows, M, m0, K, XF ? a.xf : nullptr, XF ? a.xf_ld : 0);
    sb.init(b.p, b.ld, K, nullptr, 0, nullptr, N, n0, K);

    // Rotated software pipeline.  Per k-tile: NG groups of 8 k (4 MFMAs per 32x32 tile each); the
    // fragments of group g+1 are read from LDS while group g multiplies.  The staging of the following
    // tiles rides BETWEEN the MFMAs: group 0 writes the staged registers of tile kt+1 to the other LDS
    // buffer, one quarter per MFMA sub-step, group 1 issues the global loads of tile kt+2 the same way
    // (global loads run ~a k-tile ahead of their LDS stores).  No branches on the tile count: the last iterations
    // re-read the last k-tile (cache hits) and write it to a buffer nobody reads.  The barrier is passed BEFORE
    // the last group of tile kt multiplies, and the first fragments of tile kt+1 are fetched right behind
    // it: the barrier skew and the LDS latency at the tile boundary hide behind that group's MFMAs.
    // The loop is unrolled by two k-tiles so that "which buffer" is a compile-time constant.
    constexpr int NG = BK / 8;
    static_assert(NG >= 4 && (NG % 2) == 0, "the rotation needs an even number (>= 4) of k-groups per tile");
    sa.load(kt0 * BK);
    sb.load(kt0 * BK);
#pragma unroll
    for (int part = 0; part < 4; ++part) sa.xform_part(part);
    sa.store(As);
    sb.store(Bs);
    {
        const int k1 = min(kt0 + 1, kt1 - 1) * BK;
        sa.load(k1);
        sb.load(k1);
    }
    __syncthreads();
    float fa[2][WM][4], fb[2][WN][4];
#pragma unroll
    for (int i = 0; i < WM; ++i) read_frag<TA, BM>(fa[0][i], As, arow + 32 * i, 0, h);
#pragma unroll
    for (int j = 0; j < WN; ++j) read_frag<TB, BN>(fb[0][j], Bs, brow + 32 * j, 0, h);
    // Tiles [kt0, fast_end) neither store nor load the product's partial last k-tile (a tile stores tile kt + 1 and
    // loads tile kt + 2): they run the code without masks, in pairs; what is left (at most one such tile, and the last
    // three when there is a partial tile) runs the masking code.
    const bool partial = kt1 * BK > K;
    const int fast_end = partial ? max(kt0, kt1 - 3) : kt1;
    int kt = kt0;
    for (; kt + 1 < fast_end; kt += 2) {
        k_tile<TA, TB, WM, WN, XF, 0, false>(sa, sb, As, Bs, arow, brow, h, true, min(kt + 2, kt1 - 1) * BK, fa, fb, acc);
        k_tile<TA, TB, WM, WN, XF, 1, false>(sa, sb, As, Bs, arow, brow, h, kt + 2 < kt1, min(kt + 3, kt1 - 1) * BK, fa, fb, acc);
    }
    for (; kt < kt1; kt += 2) {
        k_tile<TA, TB, WM, WN, XF, 0, true>(sa, sb, As, Bs, arow, brow, h, kt + 1 < kt1, min(kt + 2, kt1 - 1) * BK, fa, fb, acc);
        if (kt + 1 >= kt1) break;
        k_tile<TA, TB, WM, WN, XF, 1, true>(sa, sb, As, Bs, arow, brow, h, kt + 2 < kt1, min(kt + 3, kt1 - 1) * BK, fa, fb, acc);
    }
    // every wave's LDS reads completed before the last barrier: the buffers are free for the epilogue
}

// accumulator element (mt, nt, r) of this lane -> tile-local (row, col)
template <int WM>
__device__ __forceinline__ int acc_row(int mt, int r) {
    const int lane = threadIdx.x & 63, wr = threadIdx.x >> 7;
    return wr * 32 * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}
template <int WN>
__device__ __forceinline__ int acc_col(int nt) {
    const int lane = threadIdx.x & 63, wc = (threadIdx.x >> 6) & 1;
    return wc * 32 * WN + nt * 32 + (lane & 31);
}

// Accumulator tile -> global through LDS: lanes write their elements (one row segment of 32 consecutive
// floats per lane-half: conflict-free ds_write_b32), the tile is read back row-major and leaves as
// 16-byte stores of full 128-/256-/512-B row segments (4x fewer store instructions than the
// accumulator's natural one-dword-per-lane pattern, which is store-issue bound).  Needs ldc % 4 == 0,
// C 16-B aligned, N % 4 == 0 (cvec); otherwise dword stores from the same image.  beta != 0: C is read.
// Callers must have passed a barrier since the last LDS use; ends with a barrier.
template <int WM, int WN>
__device__ __forceinline__ void tile_to_global(const f32x16 (&acc)[WM][WN], float* lds, float* __restrict__ C,
                                               int64_t ldc, int m0, int n0, int M, int N, float beta, bool cvec) {
    constexpr int BM = 64 * WM, BN = 64 * WN, LDT = BN + 4;
    static_assert(BM * LDT <= lds_floats<false, false, WM, WN>() && BM * LDT <= lds_floats<true, true, WM, WN>(),
                  "output tile image must fit in the operand buffers");
#pragma unroll
    for (int mt = 0; mt < WM; ++mt)
#pragma unroll
        for (int nt = 0; nt < WN; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) lds[acc_row<WM>(mt, r) * LDT + acc_col<WN>(nt)] = acc[mt][nt][r];
    __syncthreads();
    constexpr int TPR = BN / 4, RPP = NTHREADS / TPR;
    const int col = 4 * (threadIdx.x % TPR), n = n0 + col;
#pragma unroll 4
    for (int i = 0; i < BM / RPP; ++i) {
        const int row = threadIdx.x / TPR + RPP * i, m = m0 + row;
        if (m >= M || n >= N) continue;
        float4 v = *reinterpret_cast<const float4*>(lds + row * LDT + col);
        float* c = C + (int64_t)m * ldc + n;
        if (cvec) {
            if (beta != 0.f) {
                const float4 o = *reinterpret_cast<const float4*>(c);
                v.x += beta * o.x; v.y += beta * o.y; v.z += beta * o.z; v.w += beta * o.w;
            }
            // streaming store: a product's output is read next by another kernel (same-box A/B: 29.30 -> 29.19 ms/step)
            __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(c));
        } else {
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (n + j < N) c[j] = beta != 0.f ? e[j] + beta * c[j] : e[j];
        }
    }
    __syncthreads();
}

// ---- STORE ---------------------------------------------------------------------------------------
// k-tiles [kt0, kt1) of the output tile at (m0, n0): `atomic` adds the partial product to C (zeroed, or holding the
// addend, by the host) and only the piece with `with_bias` adds the bias; otherwise alpha * product + bias leaves
// through LDS as 16-byte stores (beta applied)
template <bool TA, bool TB, int VW, int WM, int WN, bool XF = false>
__device__ __forceinline__ void store_tile_seg(const OpA& a, const OpB& b, float* __restrict__ C, int64_t ldc,
                                               const float* __restrict__ bias, int M, int N, int K, float alpha,
                                               float beta, int m0, int n0, int kt0, int kt1, bool atomic,
                                               bool with_bias, int cvec, float* lds) {
    f32x16 acc[WM][WN];
    tile_mainloop<TA, TB, VW, WM, WN, XF>(a, b, M, N, K, m0, n0, kt0, kt1, lds, acc);
    if (atomic) {
#pragma unroll
        for (int nt = 0; nt < WN; ++nt) {
            const int n = n0 + acc_col<WN>(nt);
            if (n >= N) continue;
            const float bv = (bias && with_bias) ? bias[n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < WM; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + acc_row<WM>(mt, r);
                    // C was zeroed (beta = 0) or holds the addend (beta = 1)
                    if (m < M) unsafeAtomicAdd(C + (int64_t)m * ldc + n, alpha * acc[mt][nt][r] + bv);
                }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < WN; ++nt) {
        const int n = n0 + acc_col<WN>(nt);
        const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < WM; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = alpha * acc[mt][nt][r] + bv;
    }
    tile_to_global<WM, WN>(acc, lds, C, ldc, m0, n0, M, N, beta, cvec != 0);
}

template <bool TA, bool TB, int VW, int WM, int WN, bool XF = false>
__device__ __forceinline__ void store_tile_body(const OpA& a, const OpB& b, float* __restrict__ C, int64_t ldc,
                                                const float* __restrict__ bias, int M, int N, int K, float alpha,
                                                float beta, int tiles_n, int ksplit, int kt_per_split, int cvec,
                                                int wid, float* lds) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int tile = wid / ksplit, ks = wid - tile * ksplit;
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int nk = (K + BK - 1) / BK;
    const int kt0 = ks * kt_per_split, kt1 = min(nk, kt0 + kt_per_split);
    if (kt0 >= kt1) return;
    store_tile_seg<TA, TB, VW, WM, WN, XF>(a, b, C, ldc, bias, M, N, K, alpha, beta, m0, n0, kt0, kt1, ksplit > 1,
                                            ks == 0, cvec, lds);
}

template <bool TA, bool TB, int VW, int WM, int WN, bool XF = false>
__global__ __launch_bounds__(NTHREADS) void gemm_store_kernel(OpA a, OpB b, float* __restrict__ C, int64_t ldc,
                                                               const float* __restrict__ bias, int M, int N, int K,
                                                               float alpha, float beta, int tiles_n, int ntiles,
                                                               int ksplit, int kt_per_split, int cvec) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int wid = xcd_remap(blockIdx.x, ntiles * ksplit);
    store_tile_body<TA, TB, VW, WM, WN, XF>(a, b, C, ldc, bias, M, N, K, alpha, beta, tiles_n, ksplit, kt_per_split,
                                             cvec, wid, lds);
}

// scale of (128-column tile t, slab row m) for the XF stagers: the slab holds exp(x - tile_max[t][m]), the softmax part of
// dlogits is coef * exp(x - lse) = slab * coef * exp(tile_max - lse); in place over the table the LSE kernel filled
__global__ void ce_rowscale_kernel(const float* __restrict__ lse, const float* __restrict__ coef, int M, int tiles,
                                   float* __restrict__ tab, int64_t ld) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float c = coef[m], l = lse[m];
    for (int t = blockIdx.y; t < tiles; t += gridDim.y) {
        float* q = tab + (int64_t)t * ld + m;
        *q = c > 0.f ? c * __expf(*q - l) : 0.f;
    }
}

// One-hot part of dlogits, dL[m, target[m]] -= w, as corrections of the two products (one wave per slab row):
//   dX[m, :]        -= w * Wp[target[m], :]      (Wp: the weight panel of the dgrad)
//   dW[target[m], :] -= w * Xc[m, :]             (Xc: the wgrad's right operand, ones column included; atomic)
__global__ void ce_onehot_fixup_kernel(const int64_t* __restrict__ target, int64_t ignore, const float* __restrict__ g,
                                       int rows, const float* __restrict__ Wp, int64_t ldw, int Kx,
                                       float* __restrict__ dX, int64_t lddx, const float* __restrict__ Xc, int64_t ldxc,
                                       int Kw, float* __restrict__ dW, int64_t lddw) {
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= rows) return;
    const int64_t t = target[m];
    if (t == ignore) return;
    const float w = g ? g[m] : 1.f;
    const int lane = threadIdx.x & 63;
    if (dX) {
        const float* src = Wp + t * ldw;
        float* dst = dX + (int64_t)m * lddx;
        for (int k = lane; k < Kx; k += 64) dst[k] -= w * src[k];
    }
    if (dW) {
        const float* src = Xc + (int64_t)m * ldxc;
        float* dst = dW + t * lddw;
        for (int k = lane; k < Kw; k += 64) unsafeAtomicAdd(dst + k, -w * src[k]);
    }
}

// The wgrad's last few output columns (904 = 7 x 128 + 8: what is left of [h | e | 1 0 0 0] behind the 128-wide tile
// columns) without the matrix pipe: C[v, j] (+)= sum_m slab[m, v] * scale[v >> 7][m] * X[m, j], j < nj <= 8.  A 128x64
// MFMA tile for 8 live columns multiplies 56 columns of padding (0.25 ms per step at the headline shape); this is a
// stream over the slab -- HBM-bound, 9 flop per element on the vector ALUs -- that needs no LDS while it runs, so it
// shares the CUs with the wide tiles' launch on a second stream.  A workgroup owns 256 slab columns (a float4 per
// lane) and a range of rows, its four waves take every fourth row; the partial sums meet in LDS and leave as atomic
// adds in (v, j) order, 8 consecutive floats per column.
// (X and the scales staged in LDS per row range, so that the loop issues only the slab loads, measured slower: 0.222 ms
// against 0.203 at 7936 x 30000 -- 4.7 TB/s of slab; tools/wcols_bench.py)
constexpr int WCOLS_NJ = 8;
__global__ __launch_bounds__(256) void ce_wgrad_cols_kernel(const float* __restrict__ slab, int64_t ldl, int rows,
                                                            int cols, const float* __restrict__ scale, int64_t ld_scale,
                                                            const float* __restrict__ X, int64_t ldx, int nj,
                                                            float* __restrict__ C, int64_t ldc, int rows_per_split) {
    __shared__ float red[4][256][WCOLS_NJ + 1];   // (+1: the (v, j) read-out below strides by a row)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int vb = blockIdx.x * 256, v0 = vb + 4 * lane;
    const int vc = min(v0, cols - 4);   // (cols is a multiple of 4: a lane's float4 is all inside or all outside)
    const int m_lo = blockIdx.y * rows_per_split, m_hi = min(rows, m_lo + rows_per_split);
    const float* sp = slab + vc;
    const float* sc = scale + (int64_t)(vc >> 7) * ld_scale;
    float acc[4][WCOLS_NJ];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < WCOLS_NJ; ++j) acc[c][j] = 0.f;
#pragma unroll 4
    for (int m = m_lo + wave; m < m_hi; m += 4) {
        const float4 s4 = *reinterpret_cast<const float4*>(sp + (int64_t)m * ldl);
        const float w = sc[m];
        const float* xr = X + (int64_t)m * ldx;   // wave-uniform: scalar loads
        float x[WCOLS_NJ];
#pragma unroll
        for (int j = 0; j < WCOLS_NJ; ++j) x[j] = j < nj ? xr[min(j, nj - 1)] : 0.f;
        const float a[4] = {s4.x * w, s4.y * w, s4.z * w, s4.w * w};
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < WCOLS_NJ; ++j) acc[c][j] = fmaf(a[c], x[j], acc[c][j]);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < WCOLS_NJ; ++j) red[wave][4 * lane + c][j] = acc[c][j];
    __syncthreads();
    const int j = threadIdx.x & 7;
    if (j < nj) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int vl = (threadIdx.x >> 3) + 32 * i, v = vb + vl;
            if (v < cols) {
                const float t = (red[0][vl][j] + red[1][vl][j]) + (red[2][vl][j] + red[3][vl][j]);
                unsafeAtomicAdd(C + (int64_t)v * ldc + j, t);
            }
        }
    }
}

// ---- grouped STORE: up to DSSVAE_GEMM_MAX_GROUP independent products of one layout in ONE launch ----
// (the per-direction / per-head GRU projections and weight gradients are 1-3 GFLOP each: alone they
// cannot fill the chip and pay a launch + a tail each).  C = A.B^T + bias.
// A launch holds up to GROUP_ENTRIES tile problems; a caller's product becomes one or two of them: its full 128-wide
// tile columns on the 128x128 tile body and, when the last tile column would be at most half full (N = 300 = 2 x 128 + 44,
// N = 150 = 128 + 22), the remainder on the 128x64 body -- the tile shape is per problem, inside ONE launch.
constexpr int GROUP_ENTRIES = 2 * DSSVAE_GEMM_MAX_GROUP;
struct GroupProblem {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc; const float* bias;
    int M, N, K, tiles_n, ksplit, ktps, cvec, wg_begin, narrow;
    int tiles, unit_begin;  // stream-K launches: output tiles, and where the problem starts in the launch's unit list
};
struct GroupArgs {
    GroupProblem p[GROUP_ENTRIES];
    int n;
    int units, per;  // stream-K launches: length of the unit list, units per workgroup
};
// stream-K: a k-tile of a 128x128 tile is SK_WIDE units of work, one of a 128x64 tile SK_NARROW (0.535 of it, measured)
constexpr int SK_WIDE = 15, SK_NARROW = 8;

template <bool TA, bool TB, int VW>
__global__ __launch_bounds__(NTHREADS) void gemm_group_kernel(GroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // each XCD (blockIdx % 8) works on a contiguous stretch of the launch's tile list: neighbouring tiles share an
    // operand panel, and an XCD's L2 is its own
    // (in windows of 64 workgroups, 8 neighbouring tiles per XCD: a contiguous eighth of the WHOLE launch per XCD
    // would give the XCDs problems of different cost -- measured: 0.49 -> 0.74 ms on a five-problem launch)
    const int full = (int)gridDim.x & ~63, b0 = (int)blockIdx.x;
    const int bid = b0 < full ? (b0 & ~63) + ((b0 & 7) << 3) + ((b0 >> 3) & 7) : b0;
    int i = 0;
    while (i + 1 < g.n && bid >= g.p[i + 1].wg_begin) ++i;
    const GroupProblem& q = g.p[i];
    OpA a;
    a.p1 = q.A; a.ld1 = q.lda; a.K1 = q.K; a.p2 = nullptr; a.ld2 = 0; a.rows = nullptr; a.xf = nullptr; a.xf_ld = 0;
    OpB b;
    b.p = q.B; b.ld = q.ldb;
    const int wid = bid - q.wg_begin;
    // one workgroup per tile: plain store; split K: fp32 atomics into the C the host has zero-filled
    if (q.narrow)
        store_tile_body<TA, TB, VW, 2, 1>(a, b, q.C, q.ldc, q.bias, q.M, q.N, q.K, 1.f, 0.f, q.tiles_n, q.ksplit, q.ktps,
                                           q.cvec, wid, lds);
    else
        store_tile_body<TA, TB, VW, 2, 2>(a, b, q.C, q.ldc, q.bias, q.M, q.N, q.K, 1.f, 0.f, q.tiles_n, q.ksplit, q.ktps,
                                           q.cvec, wid, lds);
}

// Stream-K form of the grouped launch, for groups with few output tiles and long K (the weight gradients: 24 .. 99
// tiles, 240 .. 728 k-tiles each).  All k-tiles of all tiles of all problems form ONE list, weighted by tile shape;
// every workgroup takes an equal contiguous share of it -- whole tiles leave as plain stores, pieces as atomics into the
// zero-filled output -- so the launch is exactly one full round of the chip, whatever the tile counts are (a split
// factor per tile fills 441 of 512 slots at best on the decoder's weight-gradient pair).
template <bool TA, bool TB, int VW>
__global__ __launch_bounds__(NTHREADS) void gemm_group_sk_kernel(GroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lo = (int)blockIdx.x * g.per, hi = min(g.units, lo + g.per);
    for (int i = 0; i < g.n; ++i) {
        const GroupProblem& q = g.p[i];
        const int w = q.narrow ? SK_NARROW : SK_WIDE;
        const int nk = (q.K + BK - 1) / BK, total = q.tiles * nk;
        int f = min((max(lo - q.unit_begin, 0) + w - 1) / w, total);
        const int f_hi = min((max(hi - q.unit_begin, 0) + w - 1) / w, total);
        if (f >= f_hi) continue;
        OpA a;
        a.p1 = q.A; a.ld1 = q.lda; a.K1 = q.K; a.p2 = nullptr; a.ld2 = 0; a.rows = nullptr; a.xf = nullptr; a.xf_ld = 0;
        OpB b;
        b.p = q.B; b.ld = q.ldb;
        while (f < f_hi) {
            const int tile = f / nk, kt0 = f - tile * nk, kt1 = min(nk, kt0 + (f_hi - f));
            const bool whole = kt0 == 0 && kt1 == nk;
            const int m0 = (tile / q.tiles_n) * 128, n0 = (tile % q.tiles_n) * (q.narrow ? 64 : 128);
            __syncthreads();  // the previous piece's LDS reads (and a plain store's tile image) are done with
            if (q.narrow)
                store_tile_seg<TA, TB, VW, 2, 1>(a, b, q.C, q.ldc, q.bias, q.M, q.N, q.K, 1.f, 0.f, m0, n0, kt0, kt1,
                                                  !whole, kt0 == 0, q.cvec, lds);
            else
                store_tile_seg<TA, TB, VW, 2, 2>(a, b, q.C, q.ldc, q.bias, q.M, q.N, q.K, 1.f, 0.f, m0, n0, kt0, kt1,
                                                  !whole, kt0 == 0, q.cvec, lds);
            f += kt1 - kt0;
        }
    }
}

// ---- LSE (optionally with argmax) ----------------------------------------------------------------
// Row statistics are taken from the tile's LDS image, not from the accumulators: the image is read
// back row-major (the same 16-byte pieces that leave for the logits slab), so a thread meets the SAME
// 16 rows and the same 4 column offsets in every column tile and carries 16 (max, sum[, arg]) triples
// instead of one per accumulator row (32 per lane): the kernel fits 2 waves per SIMD (two
// workgroups per CU), which is what hides the epilogue and the barrier skew behind the other
// workgroup's MFMAs.  One 32-lane shuffle reduction per row at the very end.
// STORE: 0 = statistics only; 1 = + the logits of the row chunk into the slab (the normalised slab of the training
// path has a kernel of its own below)
template <bool ARG, int VW, int STORE>
__global__ __launch_bounds__(NTHREADS, 2) void linear_lse_kernel(OpA a, OpB b, const float* __restrict__ bias, int M,
                                                                  int N, int K, int n_split, int tiles_per_split,
                                                                  int tiles_n, float* __restrict__ part_max,
                                                                  float* __restrict__ part_sum,
                                                                  int64_t* __restrict__ part_arg,
                                                                  float* __restrict__ logits, int64_t ldl, int lvec) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = 128, BN = 128, LDT = BN + 4;
    // 8 threads per row, each with every 8th 16-byte piece of it (4 pieces); 32 rows per pass, 4 passes: 4 running
    // (max, sum[, arg]) triples per thread live across the mainloop.  (32 threads per row with one piece each carried
    // 16 triples: with them the kernel needed more than the 256 registers two workgroups per CU leave a wave, and the
    // 36 spilled ones were reloaded inside the mainloop -- its marginal rate was 116 TFLOP/s against the plain
    // product's 125 on the same shape.)
    constexpr int TPR = 8, QPT = BN / 4 / TPR, RPP = NTHREADS / TPR, NR = BM / RPP;
    static_assert(BM * LDT <= lds_floats<false, false, 2, 2>(), "tile image must fit in the operand buffers");
    const int split = blockIdx.x % n_split, mb = blockIdx.x / n_split;
    const int m0 = mb * BM;
    // column tiles dealt evenly: the first tiles_n % n_split splits take one tile more than the others
    // (tiles_per_split = tiles_n / n_split, rounded down): no split is more than one tile longer than another
    const int rem = tiles_n - tiles_per_split * n_split;
    const int t_begin = split * tiles_per_split + min(split, rem);
    const int t_end = t_begin + tiles_per_split + (split < rem ? 1 : 0);
    const int nk = (K + BK - 1) / BK;
    const int col = 4 * (threadIdx.x % TPR), rbase = threadIdx.x / TPR;
    float rm[NR], rs[NR];
    int ra[ARG ? NR : 1];
#pragma unroll
    for (int i = 0; i < NR; ++i) { rm[i] = NEG_BIG; rs[i] = 0.f; if (ARG) ra[i] = 0; }

    for (int tn = t_begin; tn < t_end; ++tn) {
        const int n0 = tn * BN;
        f32x16 acc[2][2];
        tile_mainloop<false, false, VW, 2, 2>(a, b, M, N, K, m0, n0, 0, nk, lds, acc);
        // logits (+bias; columns past N masked) -> LDS image [128][132]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n0 + acc_col<2>(nt);
            const bool ok = c < N;
            const float bv = (bias && ok) ? bias[c] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    lds[acc_row<2>(mt, r) * LDT + acc_col<2>(nt)] = ok ? acc[mt][nt][r] + bv : NEG_BIG;
        }
        __syncthreads();
#ifdef LSE_DEV_NOEPI  // development build: the tiles without their row statistics and slab stores (results are garbage)
        if (n0 < 0)
#endif
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = rbase + RPP * i;
            float4 v[QPT];
#pragma unroll
            for (int q = 0; q < QPT; ++q) v[q] = *reinterpret_cast<const float4*>(lds + row * LDT + col + 4 * TPR * q);
            const float mo = rm[i];
            float m4[QPT];
#pragma unroll
            for (int q = 0; q < QPT; ++q) m4[q] = fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w));
            float mt_ = m4[0];
#pragma unroll
            for (int q = 1; q < QPT; ++q) mt_ = fmaxf(mt_, m4[q]);
            const float mn = fmaxf(mo, mt_);
            float add = 0.f;
#pragma unroll
            for (int q = 0; q < QPT; ++q)
                add += (__expf(v[q].x - mn) + __expf(v[q].y - mn)) + (__expf(v[q].z - mn) + __expf(v[q].w - mn));
            rs[i] = rs[i] * __expf(mo - mn) + add;
            if (ARG) {
                if (mt_ > mo) {
                    // the first (smallest) column that holds the maximum: pieces in ascending column order
                    int at = 0;
#pragma unroll
                    for (int q = QPT - 1; q >= 0; --q)
                        if (m4[q] == mt_)
                            at = n0 + col + 4 * TPR * q +
                                 (v[q].x == mt_ ? 0 : (v[q].y == mt_ ? 1 : (v[q].z == mt_ ? 2 : 3)));
                    ra[i] = at;
                }
            }
            rm[i] = mn;
#ifdef LSE_DEV_NOSLAB  // development build: statistics only, no slab stores
            if (STORE == 1 && n0 < 0) {
#else
            if (STORE == 1) {
#endif
                // side output: the logits of this row chunk into the slab
                const int m = m0 + row;
#pragma unroll
                for (int q = 0; q < QPT; ++q) {
                    const int n = n0 + col + 4 * TPR * q;
                    if (m < M && n < N) {
                        float* c = logits + (int64_t)m * ldl + n;
                        if (lvec) {
                            // streaming store: the slab (1 GB) is written once and read by other kernels; kept out
                            // of the L2 it would otherwise sweep the operand tiles from (+1.3 % at K = 900)
                            __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v[q]), reinterpret_cast<f32x4*>(c));
                        } else {
                            const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (n + j < N) c[j] = e[j];
                        }
                    }
                }
            }
        }
        __syncthreads();  // the image aliases the operand buffers of the next tile
    }
    // reduce over the TPR lanes that hold the other columns of the same rows, then write the split partials
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        float m = rm[i], s = rs[i];
        int ai = ARG ? ra[i] : 0;
#pragma unroll
        for (int o = TPR / 2; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
            const float mn = fmaxf(m, m2);
            s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
            if (ARG) {
                const int a2 = __shfl_xor(ai, o, 64);
                if (m2 > m || (m2 == m && a2 < ai)) ai = a2;
            }
            m = mn;
        }
        const int gm = m0 + rbase + RPP * i;
        if ((threadIdx.x % TPR) == 0 && gm < M) {
            part_max[(int64_t)gm * n_split + split] = m;
            part_sum[(int64_t)gm * n_split + split] = s;
            if (ARG) part_arg[(int64_t)gm * n_split + split] = ai;
        }
    }
}

// The training path's form: the slab NORMALISED per 128-column tile -- slab[m, n] = exp(x - max of row m within the
// tile), exponentials the statistics need anyway -- with (max, sum) of every (tile, row) in two [tiles_n][ldt] tables:
// the gradient GEMMs multiply a staged granule by ONE scale per (row, tile) (dssvae_ce_rowscale makes it from the maxima).
// No state crosses a tile, so the tiles are dealt as ONE list -- column tile major, row block minor -- in equal shares to
// a grid of exactly two workgroups per CU: 14 570 tiles over 512 workgroups are 28 or 29 each (column splits per row
// block gave 496 workgroups 29 or 30); a workgroup keeps its W tile through ~28 row blocks, neighbours on its XCD (a
// contiguous share of the list per XCD) the same or the next one.
template <int VW>
__global__ __launch_bounds__(NTHREADS, 2) void linear_lse_norm_kernel(OpA a, OpB b, const float* __restrict__ bias, int M,
                                                                       int N, int K, int tiles_m, int tiles_n,
                                                                       float* __restrict__ slab, int64_t ldl,
                                                                       float* __restrict__ tile_max,
                                                                       float* __restrict__ tile_sum, int64_t ldt) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = 128, BN = 128, LDT = BN + 4;
    constexpr int TPR = 8, QPT = BN / 4 / TPR, RPP = NTHREADS / TPR, NR = BM / RPP;
    static_assert(BM * LDT <= lds_floats<false, false, 2, 2>(), "tile image must fit in the operand buffers");
    const int T = tiles_m * tiles_n, G = (int)gridDim.x, w = xcd_remap(blockIdx.x, G);
    const int q = T / G, rem = T - q * G;
    const int f0 = w * q + min(w, rem), f1 = f0 + q + (w < rem ? 1 : 0);
    const int nk = (K + BK - 1) / BK;
    const int col = 4 * (threadIdx.x % TPR), rbase = threadIdx.x / TPR;
    for (int f = f0; f < f1; ++f) {
        const int tn = f / tiles_m, mb = f - tn * tiles_m;
        const int m0 = mb * BM, n0 = tn * BN;
        f32x16 acc[2][2];
        tile_mainloop<false, false, VW, 2, 2>(a, b, M, N, K, m0, n0, 0, nk, lds, acc);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = n0 + acc_col<2>(nt);
            const bool ok = c < N;
            const float bv = (bias && ok) ? bias[c] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    lds[acc_row<2>(mt, r) * LDT + acc_col<2>(nt)] = ok ? acc[mt][nt][r] + bv : NEG_BIG;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = rbase + RPP * i, m = m0 + row;
            float4 v[QPT];
#pragma unroll
            for (int qq = 0; qq < QPT; ++qq) v[qq] = *reinterpret_cast<const float4*>(lds + row * LDT + col + 4 * TPR * qq);
            float mr = fmaxf(fmaxf(v[0].x, v[0].y), fmaxf(v[0].z, v[0].w));
#pragma unroll
            for (int qq = 1; qq < QPT; ++qq) mr = fmaxf(mr, fmaxf(fmaxf(v[qq].x, v[qq].y), fmaxf(v[qq].z, v[qq].w)));
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1) mr = fmaxf(mr, __shfl_xor(mr, o, 64));   // the row's 8 lanes
            float add = 0.f;
#pragma unroll
            for (int qq = 0; qq < QPT; ++qq) {
                v[qq].x = __expf(v[qq].x - mr); v[qq].y = __expf(v[qq].y - mr);
                v[qq].z = __expf(v[qq].z - mr); v[qq].w = __expf(v[qq].w - mr);
                add += (v[qq].x + v[qq].y) + (v[qq].z + v[qq].w);
            }
#pragma unroll
            for (int o = TPR / 2; o > 0; o >>= 1) add += __shfl_xor(add, o, 64);
            if (m < M) {
                if ((threadIdx.x % TPR) == 0) {
                    tile_max[(int64_t)tn * ldt + m] = mr;
                    tile_sum[(int64_t)tn * ldt + m] = add;
                }
#pragma unroll
                for (int qq = 0; qq < QPT; ++qq) {
                    const int n = n0 + col + 4 * TPR * qq;
                    if (n < N)   // streaming store (see linear_lse_kernel); ldl and N are multiples of 4 (host check)
                        __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v[qq]),
                                                    reinterpret_cast<f32x4*>(slab + (int64_t)m * ldl + n));
                }
            }
        }
        __syncthreads();  // the image aliases the operand buffers of the next tile
    }
}

// lse[m] from the per-tile (max, sum) tables of linear_lse_norm_kernel: a workgroup takes 32 rows, 8 threads per row
// walk every 8th tile with a running (max, sum) and meet in LDS (a thread per row looping over all 235 tiles twice
// was 100 us of dependent latency per call)
__global__ __launch_bounds__(256) void lse_combine_tiles_kernel(const float* __restrict__ tmax,
                                                                const float* __restrict__ tsum, int M, int tiles,
                                                                int64_t ld, float* __restrict__ lse) {
    __shared__ float sm[8][32], ss[8][32];
    const int r = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int m = blockIdx.x * 32 + r;
    float mx = NEG_BIG, acc = 0.f;
    if (m < M) {
        for (int t = part; t < tiles; t += 8) {
            const float x = tmax[(int64_t)t * ld + m], ts = tsum[(int64_t)t * ld + m];
            const float mn = fmaxf(mx, x);
            acc = acc * __expf(mx - mn) + ts * __expf(x - mn);
            mx = mn;
        }
    }
    sm[part][r] = mx;
    ss[part][r] = acc;
    __syncthreads();
    if (part == 0 && m < M) {
        float m2 = sm[0][r];
#pragma unroll
        for (int p = 1; p < 8; ++p) m2 = fmaxf(m2, sm[p][r]);
        float a2 = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) a2 += ss[p][r] * __expf(sm[p][r] - m2);
        lse[m] = m2 + __logf(a2);
    }
}

// one wave per row: lanes stride over the splits, shuffle-reduce (max, first split holding it), then the
// rescaled sum
__global__ void lse_combine_kernel(const float* __restrict__ pm, const float* __restrict__ ps,
                                   const int64_t* __restrict__ pa, int M, int S, float* __restrict__ lse,
                                   int64_t* __restrict__ amax) {
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    const float* rm = pm + (int64_t)m * S;
    const float* rs = ps + (int64_t)m * S;
    float mx = NEG_BIG;
    int best = 0x7fffffff;
    for (int s = lane; s < S; s += 64) {
        const float v = rm[s];
        if (v > mx) { mx = v; best = s; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(mx, o, 64);
        const int b2 = __shfl_xor(best, o, 64);
        if (m2 > mx || (m2 == mx && b2 < best)) { mx = m2; best = b2; }
    }
    if (lse) {
        float acc = 0.f;
        for (int s = lane; s < S; s += 64) acc += rs[s] * __expf(rm[s] - mx);
        acc = wave_sum(acc);
        if (lane == 0) lse[m] = mx + __logf(acc);
    }
    if (lane == 0 && amax && pa) amax[m] = pa[(int64_t)m * S + (best == 0x7fffffff ? 0 : best)];
}

// ---- DLOGITS -------------------------------------------------------------------------------------
template <int VW>
__global__ __launch_bounds__(NTHREADS) void linear_dlogits_kernel(OpA a, OpB b, const float* __restrict__ bias,
                                                                   const float* __restrict__ lse,
                                                                   const float* __restrict__ coef, int M, int N, int K,
                                                                   float* __restrict__ dL, int64_t ldd,
                                                                   float* __restrict__ colsum, int tiles_n, int ntiles,
                                                                   int dvec) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BM = 128, BN = 128;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    f32x16 acc[2][2];
    tile_mainloop<false, false, VW, 2, 2>(a, b, M, N, K, m0, n0, 0, (K + BK - 1) / BK, lds, acc);
    const int c0 = n0 + acc_col<2>(0), c1 = n0 + acc_col<2>(1);
    const bool ok0 = c0 < N, ok1 = c1 < N;
    const float b0 = (bias && ok0) ? bias[c0] : 0.f, b1 = (bias && ok1) ? bias[c1] : 0.f;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + acc_row<2>(mt, r);
            const bool okm = m < M;
            const float l = okm ? lse[m] : 0.f, c = okm ? coef[m] : 0.f;
            const float v0 = (okm && ok0) ? c * __expf(acc[mt][0][r] + b0 - l) : 0.f;
            const float v1 = (okm && ok1) ? c * __expf(acc[mt][1][r] + b1 - l) : 0.f;
            acc[mt][0][r] = v0; acc[mt][1][r] = v1;
            s0 += v0; s1 += v1;
        }
    tile_to_global<2, 2>(acc, lds, dL, ldd, m0, n0, M, N, 0.f, dvec != 0);
    if (colsum) {
        s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 32, 64);
        if ((threadIdx.x & 63) < 32) {
            if (ok0) unsafeAtomicAdd(colsum + c0, s0);
            if (ok1) unsafeAtomicAdd(colsum + c1, s1);
        }
    }
}

// in place on the logits slab: x[m,n] <- coef[m] * exp(x[m,n] - lse[m]); colsum[n] += column sums.
// block = 256 threads x 4 columns, ROWS rows per block
__global__ void softmax_grad_kernel(float* __restrict__ X, int64_t ldx, const float* __restrict__ lse,
                                    const float* __restrict__ coef, int M, int N, float* __restrict__ colsum,
                                    int rows_per_block) {
    const int n = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (n >= N) return;
    const int m_begin = blockIdx.y * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (n + 3 < N) {
        for (int m = m_begin; m < m_end; ++m) {
            const float l = lse[m], c = coef[m];
            float4* p = reinterpret_cast<float4*>(X + (int64_t)m * ldx + n);
            float4 v = *p;
            v.x = c * __expf(v.x - l); v.y = c * __expf(v.y - l); v.z = c * __expf(v.z - l); v.w = c * __expf(v.w - l);
            *p = v;
            s0 += v.x; s1 += v.y; s2 += v.z; s3 += v.w;
        }
        if (colsum) {
            unsafeAtomicAdd(colsum + n, s0); unsafeAtomicAdd(colsum + n + 1, s1);
            unsafeAtomicAdd(colsum + n + 2, s2); unsafeAtomicAdd(colsum + n + 3, s3);
        }
    } else {
        for (int j = 0; n + j < N; ++j) {
            float sj = 0.f;
            for (int m = m_begin; m < m_end; ++m) {
                float* p = X + (int64_t)m * ldx + n + j;
                const float v = coef[m] * __expf(*p - lse[m]);
                *p = v;
                sj += v;
            }
            if (colsum) unsafeAtomicAdd(colsum + n + j, sj);
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

// cnt[m] = number of targets of row m that are not ignore_index (the row's weight in the loss), as float
__global__ void ce_count_kernel(const int64_t* __restrict__ target, int T, int64_t ignore, int M,
                                float* __restrict__ cnt) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    int c = 0;
    for (int t = 0; t < T; ++t) c += target[(int64_t)m * T + t] != ignore;
    cnt[m] = (float)c;
}

// nll[m] = sum over the row's targets t of (lse[m] - logit[m, target[m,t]]), ignored targets skipped.  A wave per
// (row, target) dot product: WPR waves share a row (WPR = 1 for the sequence decoders' one target per row, 4 for the
// bag-of-words heads' T targets per row: wave w takes targets w, w + 4, ...); partials meet in LDS, fixed order.
template <int WPR>
__global__ __launch_bounds__(256) void ce_row_nll_kernel(const float* __restrict__ Xh, int64_t ldh, int Kh,
                                                         const float* __restrict__ Xe, int64_t lde, int Ke,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias,
                                                         const int64_t* __restrict__ target, int T, int64_t ignore,
                                                         const float* __restrict__ lse, int M,
                                                         float* __restrict__ nll) {
    __shared__ float red[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = WPR == 1 ? blockIdx.x * 4 + wave : blockIdx.x;
    float acc = 0.f;
    if (m < M) {
        const float l = lse[m];
        for (int t = WPR == 1 ? 0 : wave; t < T; t += WPR) {
            const int64_t tg = target[(int64_t)m * T + t];
            if (tg == ignore) continue;
            const float* w = W + tg * ldw;
            float d = 0.f;
            for (int k = lane; k < Kh; k += 64) d += Xh[(int64_t)m * ldh + k] * w[k];
            for (int k = lane; k < Ke; k += 64) d += Xe[(int64_t)m * lde + k] * w[Kh + k];
            d = wave_sum(d);
            acc += l - (d + (bias ? bias[tg] : 0.f));
        }
    }
    if (WPR == 1) {
        if (lane == 0 && m < M) nll[m] = acc;
    } else {
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0 && m < M) nll[m] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// out[s] = sum of nll[s*group .. s*group+group-1]: one wave per sentence, fixed order
__global__ void ce_group_sum_kernel(const float* __restrict__ nll, int M, int group, int S, float* __restrict__ out) {
    const int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= S) return;
    const int lane = threadIdx.x & 63;
    const int m0 = s * group, m1 = min(M, m0 + group);
    float acc = 0.f;
    for (int m = m0 + lane; m < m1; m += 64) acc += nll[m];
    acc = wave_sum(acc);
    if (lane == 0) out[s] = acc;
}

__global__ void onehot_sub_kernel(float* __restrict__ dL, int64_t ldd, const int64_t* __restrict__ target,
                                  int64_t ignore, const float* __restrict__ g, int64_t total, int T, int64_t n0,
                                  int64_t n1, float* __restrict__ colsum) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int64_t tg = target[idx];
    if (tg == ignore || tg < n0 || tg >= n1) return;
    const float gv = g[idx];
    unsafeAtomicAdd(dL + (idx / T) * ldd + (tg - n0), -gv);
    if (colsum) unsafeAtomicAdd(colsum + (tg - n0), -gv);
}

// out[n] (+)= sum_m X[m,n]; block = 64 columns x 4 row slices, grid.y row chunks (atomic combine)
__global__ void colsum_kernel(const float* __restrict__ X, int64_t ldx, int M, int N, float* __restrict__ out,
                              int rows_per_block) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rs = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int m_begin = blockIdx.y * rows_per_block, m_end = min(M, m_begin + rows_per_block);
    float s = 0.f;
    if (n < N)
        for (int m = m_begin + rs; m < m_end; m += 4) s += X[(int64_t)m * ldx + n];
    red[rs][c] = s;
    __syncthreads();
    if (rs == 0 && n < N) unsafeAtomicAdd(out + n, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}

template <typename F>
void allow_lds(F* func, int bytes) {
    // > 64 KiB of dynamic LDS needs an explicit opt-in per kernel (gfx950 has 160 KiB per CU)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(func), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

bool vec_ok(const float* p, int64_t ld) { return p == nullptr || (aligned16(p) && (ld % 4) == 0); }

// The 16-byte stagers (FastStager) keep one 32-bit byte offset per granule relative to a per-tile scalar base: the
// span those offsets cover -- all rows of a k-contiguous operand, 32 k rows of a transposed one -- must stay below 4 GiB
bool fast_span_ok(int64_t ld, bool trans, int rows) {
    return (trans ? 32 * ld + rows : (int64_t)rows * ld + 32) < ((int64_t)1 << 30);
}

// widest granule (4, 2 or 1 floats) that every load of an operand can use: base alignment, leading dimension
// and the extent the granules run along (K for a k-contiguous operand, the row count for a transposed one)
int operand_vw(const float* p, int64_t ld, bool trans, int rows, int K) {
    const int ext = trans ? rows : K;
    for (int vw = 4; vw > 1; vw >>= 1)
        if ((reinterpret_cast<uintptr_t>(p) % (4 * vw)) == 0 && (ld % vw) == 0 && (ext % vw) == 0 && ext >= vw) return vw;
    return 1;
}

OpA make_panels(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke, bool* vec) {
    OpA a;
    a.p1 = Xh; a.ld1 = ldh; a.K1 = Kh; a.p2 = Ke > 0 ? Xe : nullptr; a.ld2 = lde; a.rows = nullptr; a.xf = nullptr; a.xf_ld = 0;
    *vec = vec_ok(Xh, ldh) && (Ke == 0 || vec_ok(Xe, lde)) && (Kh % 4 == 0) && (Ke % 4 == 0);
    return a;
}

int n_cus_cached() {
    static int cus = 0;
    if (cus == 0) {
        int n = 0;
        if (dssvae_device_cus(&n) != 0 || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

// development knobs read from the environment at every call (cheap next to a launch)
int tune_env(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// Launch plan of a STORE GEMM: tile width and split-K factor from a small cost model, in units of one
// 128x128 k-tile (64 MFMAs per wave).  The grid runs in rounds of `slots` co-resident workgroups
// (2 per CU); a workgroup costs a prologue + its k-tiles + an epilogue (dearer when it leaves through
// atomics).  A 128x64 k-tile takes 0.535 of a 128x128 one (measured: tools/gemm_sweep.py).
struct StorePlan { int ksplit; float cost; };
StorePlan plan_store(int M, int N, int K, int bn, bool may_split) {
    const int tiles = ((M + 127) / 128) * ((N + bn - 1) / bn);
    const int nk = (K + BK - 1) / BK;
    const int slots = tune_env("DSSVAE_GEMM_SLOTS", 2) * n_cus_cached();
    const float c_tile = bn == 128 ? 1.f : 0.535f, c_pro = 1.f, c_epi = bn == 128 ? 2.f : 1.2f;
    StorePlan best = {1, 3.0e38f};
    const int ks_max = may_split ? max(1, min(nk / 4, 64)) : 1;
    for (int ks = 1; ks <= ks_max; ++ks) {
        const int ktps = (nk + ks - 1) / ks;
        if ((nk + ktps - 1) / ktps != ks) continue;  // same launch as a smaller ks
        const int64_t wgs = (int64_t)tiles * ks;
        const float rounds = (float)((wgs + slots - 1) / slots);
        // a partly filled last round runs faster per workgroup (the CU's matrix pipe is shared by fewer)
        const float frac_last = (float)(wgs - (int64_t)(rounds - 1.f) * slots) / (float)slots;
        const float eff_rounds = (rounds - 1.f) + (frac_last > 0.5f ? 1.f : 0.6f);
        const float epi = ks > 1 ? 1.5f * c_epi : c_epi;
        const float cost = eff_rounds * (c_pro + ktps * c_tile + epi) + (ks > 1 ? 1.f : 0.f);
        if (cost < best.cost) { best.ksplit = ks; best.cost = cost; }
    }
    return best;
}

template <bool TA, bool TB, int VW, int WM, int WN, bool XF = false>
int launch_store(const OpA& a, const OpB& b, float* C, int64_t ldc, const float* bias, int M, int N, int K,
                 float alpha, float beta, int ksplit_req, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n;
    const int nk = (K + BK - 1) / BK;
    int ksplit = ksplit_req;
    if (ksplit <= 0) ksplit = plan_store(M, N, K, BN, beta == 0.f || beta == 1.f).ksplit;
    if (beta != 0.f && beta != 1.f) ksplit = 1;
    ksplit = max(1, min(ksplit, nk));
    int ktps = (nk + ksplit - 1) / ksplit;
    ksplit = (nk + ktps - 1) / ktps;  // no empty splits
    if (ksplit > 1 && beta == 0.f) {
        const int e = dssvae_zero2d(C, ldc, M, N, s);
        if (e != DSSVAE_OK) return e;
    }
    constexpr int lds_bytes = lds_floats<TA, TB, WM, WN>() * 4;
    auto kern = gemm_store_kernel<TA, TB, VW, WM, WN, XF>;
    allow_lds(kern, lds_bytes);
    const int cvec = aligned16(C) && (ldc % 4) == 0 && (N % 4) == 0;
    hipLaunchKernelGGL(kern, dim3(ntiles * ksplit), dim3(NTHREADS), lds_bytes, s, a, b, C, ldc, bias, M, N, K, alpha,
                       beta, tiles_n, ntiles, ksplit, ktps, cvec);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

thread_local int g_last_store_launches = 1;  // gemm_store_kernel launches of this thread's last store-GEMM call

template <bool TA, bool TB, int VW, bool XF = false>
int launch_store_tile(const OpA& a, const OpB& b, float* C, int64_t ldc, const float* bias, int M, int N, int K,
                      float alpha, float beta, int ksplit, hipStream_t s) {
    // tile width (128x128 or 128x64) by the cost model above
    g_last_store_launches = 1;
    const bool may_split = (beta == 0.f || beta == 1.f) && ksplit <= 0;
    const float c64 = plan_store(M, N, K, 64, may_split).cost, c128 = plan_store(M, N, K, 128, may_split).cost;
    bool narrow = c64 < c128;
    const int force = tune_env("DSSVAE_GEMM_TILE", 0);  // development knob: 64 / 128 force the tile width, 1 = no mixing
    if (force == 64) narrow = true;
    if (force == 128) narrow = false;
    // Mixed widths: when the last 128-wide tile column would be at most half full (N = 904 = 7 x 128 + 8: the
    // CE wgrad; N = 300 = 2 x 128 + 44: its dgrad), the full tile columns go to the 128x128 kernel and the
    // remainder to the 128x64 kernel as a GEMM of its own on the same A (disjoint columns of C): 7.5 tile columns'
    // worth of work instead of 8, 2.5 instead of 3 (or 5 narrow ones).
    const int rem = N % 128, n_main = N - rem;
    if (force == 0 && n_main > 0 && rem > 0 && rem <= 64 &&
        plan_store(M, n_main, K, 128, may_split).cost + plan_store(M, rem, K, 64, may_split).cost < fminf(c64, c128)) {
        const int e = launch_store<TA, TB, VW, 2, 2, XF>(a, b, C, ldc, bias, M, n_main, K, alpha, beta, ksplit, s);
        if (e != DSSVAE_OK) return e;
        g_last_store_launches = 2;
        OpB b2 = b;
        b2.p += TB ? (int64_t)n_main : (int64_t)n_main * b.ld;
        return launch_store<TA, TB, VW, 2, 1, XF>(a, b2, C + n_main, ldc, bias ? bias + n_main : nullptr, M, rem, K,
                                                  alpha, beta, ksplit, s);
    }
    if (narrow) return launch_store<TA, TB, VW, 2, 1, XF>(a, b, C, ldc, bias, M, N, K, alpha, beta, ksplit, s);
    return launch_store<TA, TB, VW, 2, 2, XF>(a, b, C, ldc, bias, M, N, K, alpha, beta, ksplit, s);
}

}  // namespace

extern "C" int dssvae_gemm_last_launches(void) { return g_last_store_launches; }

extern "C" int dssvae_gemm_f32(const float* A, int64_t lda, int trans_a, const int64_t* a_rows, const float* B,
                               int64_t ldb, int trans_b, float* C, int64_t ldc, const float* bias, int M, int N,
                               int K, float alpha, float beta, void* stream) {
    return dssvae_gemm_f32_splitk(A, lda, trans_a, a_rows, B, ldb, trans_b, C, ldc, bias, M, N, K, alpha, beta, 0,
                                  stream);
}

extern "C" int dssvae_gemm_f32_splitk(const float* A, int64_t lda, int trans_a, const int64_t* a_rows,
                                      const float* B, int64_t ldb, int trans_b, float* C, int64_t ldc,
                                      const float* bias, int M, int N, int K, float alpha, float beta, int ksplit,
                                      void* stream) {
    if (!A || !B || !C || M < 0 || N < 0 || K <= 0 || ldc < N) return DSSVAE_EINVAL;
    if (trans_a && a_rows) return DSSVAE_EINVAL;
    if ((trans_a ? lda < M : lda < K) || (trans_b ? ldb < N : ldb < K)) return DSSVAE_EINVAL;
    if (M == 0 || N == 0) return DSSVAE_OK;
    OpA a;
    a.p1 = A; a.ld1 = lda; a.K1 = K; a.p2 = nullptr; a.ld2 = 0; a.rows = a_rows; a.xf = nullptr; a.xf_ld = 0;
    OpB b;
    b.p = B; b.ld = ldb;
    // widest granule both operands allow: every granule is aligned and entirely in or out of range
    int vw = min(operand_vw(A, lda, trans_a != 0, M, K), operand_vw(B, ldb, trans_b != 0, N, K));
    // the 16-byte stagers take one panel, no row gather, 32-bit offsets: anything else runs on 8-byte granules
    if (vw == 4 && (a_rows || !fast_span_ok(lda, trans_a != 0, M) || !fast_span_ok(ldb, trans_b != 0, N))) vw = 2;
    hipStream_t s = (hipStream_t)stream;
#define GO(TA, TB)                                                                                                  \
    do {                                                                                                            \
        if (vw == 4) return launch_store_tile<TA, TB, 4>(a, b, C, ldc, bias, M, N, K, alpha, beta, ksplit, s);      \
        if (vw == 2) return launch_store_tile<TA, TB, 2>(a, b, C, ldc, bias, M, N, K, alpha, beta, ksplit, s);      \
        return launch_store_tile<TA, TB, 1>(a, b, C, ldc, bias, M, N, K, alpha, beta, ksplit, s);                   \
    } while (0)
    if (trans_a) { if (trans_b) GO(true, true); else GO(true, false); }
    else { if (trans_b) GO(false, true); else GO(false, false); }
#undef GO
}

namespace {

// k-tiles per workgroup of a narrow (128x64) entry for `ktps` of a wide one: equal time (a 128x64 k-tile takes 0.535
// of a 128x128 one), so that the workgroups of a launch finish together
inline int narrow_ktps(int ktps) { return (ktps * 15 + 7) / 8; }

template <bool TA, bool TB, int VW>
int launch_group(GroupArgs& g, const float* const* c_full, const int64_t* c_ld, const int* c_m, const int* c_n,
                 const int* owner, int n_orig, hipStream_t s) {
    // one k-tile count per workgroup for the whole group (they finish together): the smallest that keeps
    // the launch within one round of 2 workgroups per CU (and >= 4 k-tiles per workgroup)
    const int slots = 2 * n_cus_cached();
    int nk_max = 1;
    for (int i = 0; i < g.n; ++i) nk_max = max(nk_max, (g.p[i].K + BK - 1) / BK);
    auto wgs_of = [&](const GroupProblem& q, int ktps) {
        const int nk = (q.K + BK - 1) / BK, bn = q.narrow ? 64 : 128;
        const int kt = min(nk, q.narrow ? narrow_ktps(ktps) : ktps);
        return (int64_t)((q.M + 127) / 128) * ((q.N + bn - 1) / bn) * ((nk + kt - 1) / kt);
    };
    constexpr int lds_bytes = lds_floats<TA, TB, 2, 2>() * 4;
    static_assert(lds_floats<TA, TB, 2, 1>() <= lds_floats<TA, TB, 2, 2>(), "the wide body sizes the launch's LDS");
    int ktps = tune_env("DSSVAE_GROUP_KTPS", 0);  // development knob
    // stream-K when the group has too few tiles to fill the chip by itself and enough k-tiles to share out
    int64_t tiles_all = 0, units = 0;
    for (int i = 0; i < g.n; ++i) {
        GroupProblem& q = g.p[i];
        const int nk = (q.K + BK - 1) / BK, bn = q.narrow ? 64 : 128;
        q.tiles_n = (q.N + bn - 1) / bn;
        q.tiles = ((q.M + 127) / 128) * q.tiles_n;
        q.unit_begin = (int)units;
        tiles_all += q.tiles;
        units += (int64_t)q.tiles * nk * (q.narrow ? SK_NARROW : SK_WIDE);
    }
    const int sk_mode = tune_env("DSSVAE_GROUP_SK", 1);
    if (ktps <= 0 && sk_mode && units < (1 << 30) && 2 * tiles_all <= 3 * slots && units >= (int64_t)slots * 6 * SK_WIDE) {
        DssvaeZeroMany zm;
        zm.n = 0; zm.blk_begin[0] = 0;
        for (int i = 0; i < n_orig; ++i) dssvae_zero_many_add(zm, const_cast<float*>(c_full[i]), c_ld[i], c_m[i], c_n[i]);
        const int e = dssvae_zero_many_launch(zm, s);
        if (e != DSSVAE_OK) return e;
        g.units = (int)units;
        g.per = (int)((units + slots - 1) / slots);
        const int wgs = (int)((units + g.per - 1) / g.per);
        auto kern = gemm_group_sk_kernel<TA, TB, VW>;
        allow_lds(kern, lds_bytes);
        hipLaunchKernelGGL(kern, dim3(wgs), dim3(NTHREADS), lds_bytes, s, g);
        DSSVAE_CHECK_LAUNCH();
        return DSSVAE_OK;
    }
    if (ktps <= 0) {
        ktps = min(4, nk_max);
        for (;; ++ktps) {
            int64_t wgs = 0;
            for (int i = 0; i < g.n; ++i) wgs += wgs_of(g.p[i], ktps);
            if (wgs <= slots || ktps >= nk_max) break;
        }
    }
    int wg = 0;
    bool split[DSSVAE_GEMM_MAX_GROUP] = {};
    for (int i = 0; i < g.n; ++i) {
        GroupProblem& q = g.p[i];
        const int nk = (q.K + BK - 1) / BK, bn = q.narrow ? 64 : 128;
        q.tiles_n = (q.N + bn - 1) / bn;
        q.ktps = min(nk, q.narrow ? narrow_ktps(ktps) : ktps);
        q.ksplit = (nk + q.ktps - 1) / q.ktps;
        q.wg_begin = wg;
        wg += ((q.M + 127) / 128) * q.tiles_n * q.ksplit;
        if (q.ksplit > 1) split[owner[i]] = true;
    }
    DssvaeZeroMany zm;
    zm.n = 0; zm.blk_begin[0] = 0;
    for (int i = 0; i < n_orig; ++i)  // all split outputs in one fill launch
        if (split[i]) dssvae_zero_many_add(zm, const_cast<float*>(c_full[i]), c_ld[i], c_m[i], c_n[i]);
    {
        const int e = dssvae_zero_many_launch(zm, s);
        if (e != DSSVAE_OK) return e;
    }
    auto kern = gemm_group_kernel<TA, TB, VW>;
    allow_lds(kern, lds_bytes);
    hipLaunchKernelGGL(kern, dim3(wg), dim3(NTHREADS), lds_bytes, s, g);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

}  // namespace

extern "C" int dssvae_gemm_f32_grouped(const dssvae_gemm_problem* pr, int n, int trans_a, int trans_b, void* stream) {
    if (!pr || n <= 0 || n > DSSVAE_GEMM_MAX_GROUP) return DSSVAE_EINVAL;
    GroupArgs g;
    g.n = 0;
    int vw = 4;
    const float* c_full[DSSVAE_GEMM_MAX_GROUP];
    int64_t c_ld[DSSVAE_GEMM_MAX_GROUP];
    int c_m[DSSVAE_GEMM_MAX_GROUP], c_n[DSSVAE_GEMM_MAX_GROUP], owner[GROUP_ENTRIES];
    // development knobs: 64 / 128 force the tile width, 1 = no mixing (DSSVAE_GROUP_TILE: grouped launches only)
    const int force = tune_env("DSSVAE_GROUP_TILE", tune_env("DSSVAE_GEMM_TILE", 0));
    for (int i = 0; i < n; ++i) {
        const dssvae_gemm_problem& x = pr[i];
        if (!x.A || !x.B || !x.C || x.M <= 0 || x.N <= 0 || x.K <= 0 || x.ldc < x.N) return DSSVAE_EINVAL;
        if ((trans_a ? x.lda < x.M : x.lda < x.K) || (trans_b ? x.ldb < x.N : x.ldb < x.K)) return DSSVAE_EINVAL;
        vw = min(vw, min(operand_vw(x.A, x.lda, trans_a != 0, x.M, x.K), operand_vw(x.B, x.ldb, trans_b != 0, x.N, x.K)));
        if (vw == 4 && (!fast_span_ok(x.lda, trans_a != 0, x.M) || !fast_span_ok(x.ldb, trans_b != 0, x.N))) vw = 2;
        c_full[i] = x.C; c_ld[i] = x.ldc; c_m[i] = x.M; c_n[i] = x.N;
        // tile shape of THIS product: full 128-wide tile columns + a 64-wide remainder column, or all of it in
        // one width (a 128x64 k-tile takes 0.535 of a 128x128 one)
        // Measured on the step's groups (tools/group_sweep.py): below ~6 tile columns the 128x64 body wins whatever the
        // remainder is (more and shorter workgroups, three instead of two per CU: nn 4 x 7680x300x450 0.134 -> 0.104 ms,
        // nn 23296x300x900 0.151 -> 0.131, tn 1800x600x7936 + 900x300x23296 0.357 -> 0.333); from N = 900 up the wide /
        // mixed split does (nt 5 x ..x900x300: 0.482 against 0.500)
        const int rem = x.N % 128, n_main = x.N - rem;
        const bool small_n = x.N < 768;
        const bool mixed = force == 0 && !small_n && n_main > 0 && rem > 0 && rem <= 64;
        const bool narrow = force == 64 || (force != 128 && !mixed &&
                                            (small_n || ((x.N + 63) / 64) * 535 < ((x.N + 127) / 128) * 1000));
        for (int part = 0; part < (mixed ? 2 : 1); ++part) {
            const int c0 = part ? n_main : 0, cols = mixed ? (part ? rem : n_main) : x.N;
            GroupProblem& q = g.p[g.n];
            owner[g.n++] = i;
            q.A = x.A; q.lda = x.lda;
            q.B = x.B + (trans_b ? (int64_t)c0 : (int64_t)c0 * x.ldb); q.ldb = x.ldb;
            q.C = x.C + c0; q.ldc = x.ldc; q.bias = x.bias ? x.bias + c0 : nullptr;
            q.M = x.M; q.N = cols; q.K = x.K;
            q.narrow = mixed ? part : (narrow ? 1 : 0);
            q.cvec = aligned16(q.C) && (x.ldc % 4) == 0 && (cols % 4) == 0;
        }
    }
    hipStream_t s = (hipStream_t)stream;
#define GO(TA, TB)                                                                                   \
    do {                                                                                             \
        if (vw == 4) return launch_group<TA, TB, 4>(g, c_full, c_ld, c_m, c_n, owner, n, s);         \
        if (vw == 2) return launch_group<TA, TB, 2>(g, c_full, c_ld, c_m, c_n, owner, n, s);         \
        return launch_group<TA, TB, 1>(g, c_full, c_ld, c_m, c_n, owner, n, s);                      \
    } while (0)
    if (trans_a) { if (trans_b) GO(true, true); else GO(true, false); }
    else { if (trans_b) GO(false, true); else GO(false, false); }
#undef GO
}

static int lse_fwd(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke, const float* W,
                   int64_t ldw, const float* bias, int M, int N, int n_split, float* part_max, float* part_sum,
                   int64_t* part_arg, int want_argmax, float* logits, int64_t ldl, void* stream) {
    if (!Xh || !W || !part_max || !part_sum || M <= 0 || N <= 0 || Kh <= 0 || Ke < 0 || n_split <= 0)
        return DSSVAE_EINVAL;
    if (Ke > 0 && !Xe) return DSSVAE_EINVAL;
    if (want_argmax && !part_arg) return DSSVAE_EINVAL;
    if (logits && (ldl < N || want_argmax)) return DSSVAE_EINVAL;
    bool vec;
    OpA a = make_panels(Xh, ldh, Kh, Xe, lde, Ke, &vec);
    OpB b;
    b.p = W; b.ld = ldw;
    vec = vec && vec_ok(W, ldw);
    // 16-byte granules (the fast stagers) want ONE activation panel; two panels ([h | e] as separate tensors) run
    // on 8-byte granules -- callers that can should concatenate (dss_vae_amd/ops.py does)
    const int vw = !vec ? 1 : ((Ke == 0 && fast_span_ok(ldh, false, M) && fast_span_ok(ldw, false, N)) ? 4 : 2);
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    if (n_split > tiles_n) return DSSVAE_EINVAL;
    const int tps = tiles_n / n_split;  // rounded down: the kernel deals the remainder one tile each
    hipStream_t s = (hipStream_t)stream;
    constexpr int lds_bytes = lds_floats<false, false, 2, 2>() * 4;
    const int lvec = logits && aligned16(logits) && (ldl % 4) == 0 && (N % 4) == 0;
#define GO1(ARG, VW, ST)                                                                                         \
    do {                                                                                                          \
        allow_lds(linear_lse_kernel<ARG, VW, ST>, lds_bytes);                                                     \
        hipLaunchKernelGGL((linear_lse_kernel<ARG, VW, ST>), dim3(tiles_m * n_split), dim3(NTHREADS), lds_bytes,  \
                           s, a, b, bias, M, N, Kh + Ke, n_split, tps, tiles_n, part_max, part_sum, part_arg,     \
                           logits, ldl, lvec);                                                                    \
    } while (0)
#define GO(ARG, ST)                                                                                               \
    do {                                                                                                          \
        if (vw == 4) GO1(ARG, 4, ST); else if (vw == 2) GO1(ARG, 2, ST); else GO1(ARG, 1, ST);                    \
    } while (0)
    if (logits) GO(false, 1);
    else if (want_argmax) GO(true, 0);
    else GO(false, 0);
#undef GO
#undef GO1
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_linear_lse_fwd(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                                     const float* W, int64_t ldw, const float* bias, int M, int N, int n_split,
                                     float* part_max, float* part_sum, int64_t* part_arg, int want_argmax,
                                     float* logits, int64_t ldl, void* stream) {
    return lse_fwd(Xh, ldh, Kh, Xe, lde, Ke, W, ldw, bias, M, N, n_split, part_max, part_sum, part_arg, want_argmax,
                   logits, ldl, stream);
}

extern "C" int dssvae_linear_lse_fwd_norm(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                                          const float* W, int64_t ldw, const float* bias, int M, int N, float* slab,
                                          int64_t ldl, float* tile_max, float* tile_sum, int64_t ldt, void* stream) {
    if (!Xh || !W || !slab || !tile_max || !tile_sum || M <= 0 || N <= 0 || Kh <= 0 || Ke < 0 || (Ke > 0 && !Xe))
        return DSSVAE_EINVAL;
    // the normalised slab leaves as 16-byte pieces only
    if (ldl < N || ldt < M || !aligned16(slab) || (ldl % 4) != 0 || (N % 4) != 0) return DSSVAE_EINVAL;
    bool vec;
    OpA a = make_panels(Xh, ldh, Kh, Xe, lde, Ke, &vec);
    OpB b;
    b.p = W; b.ld = ldw;
    vec = vec && vec_ok(W, ldw);
    const int vw = !vec ? 1 : ((Ke == 0 && fast_span_ok(ldh, false, M) && fast_span_ok(ldw, false, N)) ? 4 : 2);
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    const int64_t T = (int64_t)tiles_m * tiles_n;
    if (T >= ((int64_t)1 << 30)) return DSSVAE_ETOOBIG;
    const int slots = tune_env("DSSVAE_GEMM_SLOTS", 2) * n_cus_cached();
    // fewer than two rounds of tiles: one tile per workgroup, the dispatcher backfills (546 tiles as 512 workgroups of
    // one or two tiles took 0.333 ms, as 546 workgroups 0.257); else equal shares of the list to one round
    const int grid = (int)(T < 2 * (int64_t)slots ? T : slots);
    hipStream_t s = (hipStream_t)stream;
    constexpr int lds_bytes = lds_floats<false, false, 2, 2>() * 4;
#define GO1(VW)                                                                                                   \
    do {                                                                                                          \
        allow_lds(linear_lse_norm_kernel<VW>, lds_bytes);                                                         \
        hipLaunchKernelGGL((linear_lse_norm_kernel<VW>), dim3(grid), dim3(NTHREADS), lds_bytes, s, a, b, bias, M, \
                           N, Kh + Ke, tiles_m, tiles_n, slab, ldl, tile_max, tile_sum, ldt);                      \
    } while (0)
    if (vw == 4) GO1(4); else if (vw == 2) GO1(2); else GO1(1);
#undef GO1
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_lse_combine_tiles(const float* tile_max, const float* tile_sum, int M, int tiles, int64_t ldt,
                                        float* lse, void* stream) {
    if (!tile_max || !tile_sum || !lse || M <= 0 || tiles <= 0 || ldt < M) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(lse_combine_tiles_kernel, dim3((M + 31) / 32), dim3(256), 0, (hipStream_t)stream, tile_max,
                       tile_sum, M, tiles, ldt, lse);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_lse_combine(const float* part_max, const float* part_sum, const int64_t* part_arg, int M,
                                  int n_split, float* lse, int64_t* argmax, void* stream) {
    if (!part_max || !part_sum || M <= 0 || n_split <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(lse_combine_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, part_max,
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
                                        const float* coef, int M, int N, float* dL, int64_t ldd, float* colsum,
                                        void* stream) {
    if (!Xh || !W || !lse || !coef || !dL || M <= 0 || N <= 0 || Kh <= 0 || ldd < N || (Ke > 0 && !Xe))
        return DSSVAE_EINVAL;
    bool vec;
    OpA a = make_panels(Xh, ldh, Kh, Xe, lde, Ke, &vec);
    OpB b;
    b.p = W; b.ld = ldw;
    vec = vec && vec_ok(W, ldw);
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128, ntiles = tiles_m * tiles_n;
    constexpr int lds_bytes = lds_floats<false, false, 2, 2>() * 4;
    const int dvec = aligned16(dL) && (ldd % 4) == 0 && (N % 4) == 0;
    const int vw = !vec ? 1 : ((Ke == 0 && fast_span_ok(ldh, false, M) && fast_span_ok(ldw, false, N)) ? 4 : 2);
#define GO(VW)                                                                                                       \
    do {                                                                                                             \
        allow_lds(linear_dlogits_kernel<VW>, lds_bytes);                                                             \
        hipLaunchKernelGGL(linear_dlogits_kernel<VW>, dim3(ntiles), dim3(NTHREADS), lds_bytes, (hipStream_t)stream,  \
                           a, b, bias, lse, coef, M, N, Kh + Ke, dL, ldd, colsum, tiles_n, ntiles, dvec);            \
    } while (0)
    if (vw == 4) GO(4); else if (vw == 2) GO(2); else GO(1);
#undef GO
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_softmax_grad_inplace(float* X, int64_t ldx, const float* lse, const float* coef, int M, int N,
                                           float* colsum, void* stream) {
    if (!X || !lse || !coef || M <= 0 || N <= 0 || ldx < N || (ldx % 4) != 0 || !aligned16(X)) return DSSVAE_EINVAL;
    const int col_blocks = (N + 1023) / 1024;
    int row_blocks = max(1, min((M + 15) / 16, (8 * n_cus_cached() + col_blocks - 1) / col_blocks));
    const int rpb = (M + row_blocks - 1) / row_blocks;
    row_blocks = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(softmax_grad_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, (hipStream_t)stream, X, ldx,
                       lse, coef, M, N, colsum, rpb);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_onehot_sub(float* dL, int64_t ldd, const int64_t* target, int T, int64_t ignore_index,
                                 const float* g, int M, int64_t n0, int64_t n1, float* colsum, void* stream) {
    if (!dL || !target || !g || M <= 0 || T <= 0 || n1 <= n0) return DSSVAE_EINVAL;
    const int64_t total = (int64_t)M * T;
    hipLaunchKernelGGL(onehot_sub_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       dL, ldd, target, ignore_index, g, total, T, n0, n1, colsum);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, void* stream) {
    if (!X || !out || M <= 0 || N <= 0) return DSSVAE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        const int e = dssvae_zero2d(out, N, 1, N, s);
        if (e != DSSVAE_OK) return e;
    }
    const int col_blocks = (N + 63) / 64;
    int row_blocks = max(1, min((M + 63) / 64, (4 * n_cus_cached() + col_blocks - 1) / col_blocks));
    const int rpb = ((M + row_blocks - 1) / row_blocks + 3) / 4 * 4;
    row_blocks = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, s, X, ldx, M, N, out, rpb);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_ce_rowscale(const float* lse, const float* coef, int M, int tiles, float* table, int64_t ld,
                                  void* stream) {
    if (!lse || !coef || !table || M <= 0 || tiles <= 0 || ld < M) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(ce_rowscale_kernel, dim3((M + 255) / 256, tiles < 64 ? tiles : 64), dim3(256), 0,
                       (hipStream_t)stream, lse, coef, M, tiles, table, ld);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_ce_onehot_fixup(const int64_t* target, int64_t ignore_index, const float* g, int rows,
                                      const float* W_panel, int64_t ldw, int Kx, float* dX, int64_t lddx,
                                      const float* Xc, int64_t ldxc, int Kw, float* dW, int64_t lddw, void* stream) {
    if (!target || rows <= 0 || (dX && (!W_panel || Kx <= 0 || lddx < Kx)) || (dW && (!Xc || Kw <= 0 || lddw < Kw)))
        return DSSVAE_EINVAL;
    if (!dX && !dW) return DSSVAE_OK;
    hipLaunchKernelGGL(ce_onehot_fixup_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, target,
                       ignore_index, g, rows, W_panel, ldw, Kx, dX, lddx, Xc, ldxc, Kw, dW, lddw);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_ce_wgrad_cols(const float* slab, int64_t ldl, int rows, int cols, const float* scale,
                                    int64_t ld_scale, const float* X, int64_t ldx, int nj, float* C, int64_t ldc,
                                    float beta, void* stream) {
    if (!slab || !scale || !X || !C || rows <= 0 || cols <= 0 || nj <= 0 || ldl < cols || ldx < nj || ldc < nj ||
        ld_scale < rows)
        return DSSVAE_EINVAL;
    if (beta != 0.f && beta != 1.f) return DSSVAE_EINVAL;
    if (nj > WCOLS_NJ || !aligned16(slab) || (ldl % 4) != 0 || (cols % 4) != 0) return DSSVAE_ETOOBIG;
    hipStream_t s = (hipStream_t)stream;
    if (beta == 0.f) {   // the row ranges add their sums atomically
        const int e = dssvae_zero2d(C, ldc, cols, nj, s);
        if (e != DSSVAE_OK) return e;
    }
    // enough workgroups for a few rounds of the chip, at least 64 rows per wave
    const int vblocks = (cols + 255) / 256;
    int splits = (8 * n_cus_cached() + vblocks - 1) / vblocks;
    splits = max(1, min(splits, (rows + 255) / 256));
    const int rps = ((rows + splits - 1) / splits + 3) & ~3;
    hipLaunchKernelGGL(ce_wgrad_cols_kernel, dim3(vblocks, (rows + rps - 1) / rps), dim3(256), 0, s, slab, ldl, rows,
                       cols, scale, ld_scale, X, ldx, nj, C, ldc, rps);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gemm_f32_dlogits(const float* slab, int64_t ldl, int rows, int cols, const float* scale,
                                       int64_t ld_scale, int trans_a, const float* B, int64_t ldb, float* C,
                                       int64_t ldc, int N, float beta, void* stream) {
    const float* logits = slab;
    if (!logits || !scale || !B || !C || rows <= 0 || cols <= 0 || N <= 0 || ldl < cols || ldb < N || ldc < N ||
        ld_scale < rows)
        return DSSVAE_EINVAL;
    if (beta != 0.f && beta != 1.f) return DSSVAE_EINVAL;
    // 16-byte granules everywhere: the caller falls back to dssvae_softmax_grad_inplace + dssvae_gemm_f32 otherwise
    if (!aligned16(logits) || (ldl % 4) != 0 || (cols % 4) != 0 || !aligned16(B) ||
        (ldb % 4) != 0 || (N % 4) != 0 || !fast_span_ok(ldl, trans_a != 0, trans_a ? cols : rows) ||
        !fast_span_ok(ldb, true, N))
        return DSSVAE_ETOOBIG;
    OpA a;
    a.p1 = logits; a.ld1 = ldl; a.p2 = nullptr; a.ld2 = 0; a.rows = nullptr;
    a.xf = scale; a.xf_ld = ld_scale;
    OpB b;
    b.p = B; b.ld = ldb;
    hipStream_t s = (hipStream_t)stream;
    if (!trans_a) {  // C[rows, N] = dlogits . B       (contraction over the vocabulary)
        a.K1 = cols;
        return launch_store_tile<false, true, 4, true>(a, b, C, ldc, nullptr, rows, N, cols, 1.f, beta, 0, s);
    }
    // C[cols, N] = dlogits^T . B                   (contraction over the slab rows)
    a.K1 = rows;
    return launch_store_tile<true, true, 4, true>(a, b, C, ldc, nullptr, cols, N, rows, 1.f, beta, 0, s);
}

extern "C" int dssvae_ce_count(const int64_t* target, int T, int64_t ignore_index, int M, float* cnt, void* stream) {
    if (!target || !cnt || M <= 0 || T <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(ce_count_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, target, T,
                       ignore_index, M, cnt);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_ce_finish(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                                const float* W, int64_t ldw, const float* bias, const int64_t* target, int T,
                                int64_t ignore_index, const float* lse, int M, int group, float* row_nll, float* out,
                                void* stream) {
    if (!Xh || !W || !target || !lse || !out || M <= 0 || T <= 0 || Kh <= 0 || group <= 0 || (Ke > 0 && !Xe) ||
        (group > 1 && !row_nll))
        return DSSVAE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (T == 1)
        hipLaunchKernelGGL(ce_row_nll_kernel<1>, dim3((M + 3) / 4), dim3(256), 0, s, Xh, ldh, Kh, Xe, lde, Ke, W, ldw,
                           bias, target, T, ignore_index, lse, M, group > 1 ? row_nll : out);
    else
        hipLaunchKernelGGL(ce_row_nll_kernel<4>, dim3(M), dim3(256), 0, s, Xh, ldh, Kh, Xe, lde, Ke, W, ldw, bias,
                           target, T, ignore_index, lse, M, group > 1 ? row_nll : out);
    DSSVAE_CHECK_LAUNCH();
    if (group > 1) {
        const int S = (M + group - 1) / group;
        hipLaunchKernelGGL(ce_group_sum_kernel, dim3((S + 3) / 4), dim3(256), 0, s, row_nll, M, group, S, out);
        DSSVAE_CHECK_LAUNCH();
    }
    return DSSVAE_OK;
}
