// DEVELOPMENT PROBE -- NOT PART OF THE LIBRARY (not in csrc/Makefile, no header declaration, nothing loads it).
// Round 3's attempt at SURVEY K11 as ONE persistent launch for the whole greedy decode, kept for the measurement
// DESIGN.md section 5 ("What bounds generation") quotes.  Status when it was shelved:
//   * a first version (16 waves = 4 batch tiles x 4 parts of K, register chunks of five slices) passed every parity case
//     (tokens == the fp64 loop == the launch loop, five decoder shapes) and ran the tree-like decoder (E 300, H 600,
//     2 layers, V 300, batch 50) at 51 us per step against 39 us for the launch loop of decode.hip;
//   * this version (8 waves, a part's whole operand range loaded in one round trip) measures 50 us per step and
//     still MISMATCHES the reference on one of the five parity cases (50 x E300 H600 V300, 2 layers) -- it was not
//     debugged further because the design cannot win: every one of the 240 workgroups pulls the whole state
//     (154 - 231 KB per phase through its CU's 64 B/clk path; 55 MB per step over the chip), three dependent phases
//     per step at >= 10 us each (loads 3.8 - 6.9, cell math 1.3 - 2.0, store drain 1.5 - 2.2, publish 1.0 - 1.5).
// To build it again: add it to SRCS, declare dssvae_greedy_decode_persistent in the header, give gru.hip the
// dssvae_sticky_word_of_current_device() accessor it expects.
// Persistent greedy decoder for gfx950 (decoder/rnn_decoder.py:160-168 with :61-73 per step):
// ONE launch runs all max_len steps of embedding row -> n_layers GRU cells -> [h ; e] W_out^T + b -> arg-max -> next
// token, for decoders whose weights fit the chip's LDS (the 300-symbol tree vocabulary of DSS-VAE: 16 MB of weights
// over 240 workgroups; the 30 000-word sentence decoder does not -- its W_out is 108 MB -- and stays on the launch
// loop of decode.hip, which this entry point refers the caller to with DSSVAE_ETOOBIG).
//
// Decomposition (decode batch B <= 64 = 4 batch tiles of 16 rows):
//   * layer l is cut into n_cell = ceil(H / 5) workgroups; one owns 5 hidden units = 15 gate rows (r, z, n) of
//     [W_ih | W_hh], resident in LDS for the whole decode as the A operand of v_mfma_f32_16x16x4_f32 (row 16 is pad);
//   * the vocabulary is cut into n_proj = ceil(V / 16) tiles of 16 rows of W_out, held by the first n_proj workgroups
//     of layer 0 beside their cell slice (a projection tile has the same K = H + E as a layer-0 cell slice);
//   * a workgroup is 8 waves = 4 batch tiles x 2 parts of K: the input range and the recurrent range, so that the n
//     gate's two halves (W_in x + b_in and W_hn h + b_hn, which the cell keeps apart) are simply the sums of the two
//     parts; a part issues ALL loads of its range before its first MFMA (one memory round trip per phase: 8 waves
//     leave a wave 256 registers for that) and polls the counter of what IT reads, so the recurrent part of step
//     t + 1 runs ahead, under the other layers' and the projection's phases of step t;
//   * states travel through ping-pong buffers in global memory laid out [16-unit slice][64 rows][16 units] (a wave's
//     B operand of a slice is one contiguous 1 KB block), write-through stores and L1-bypassing loads (sc1), one
//     monotonic arrival counter per (producer stage, batch tile): the hand-off of gru.hip;
//   * the projection's workgroups leave (max, arg) per row and tile; the layer-0 waves that read the embedding rows
//     pick the token themselves from the n_proj partials (ties to the smaller index, as topk(1) of the reference's
//     sorted order is not defined on ties either: the launch loop does the same).
// Every spin is bounded and raises the launch's error word and the device's sticky word (ops.check_persistent_errors).
#include "common.h"
#include <stdlib.h>

extern unsigned* dssvae_sticky_word_of_current_device();  // gru.hip

namespace {

constexpr int PW = 8;     // waves per workgroup: 256 registers per wave, a part's whole operand range in flight
constexpr int KS = 2;     // parts of K per batch tile: the input range and the recurrent range
constexpr int NBT = 4;    // batch tiles
constexpr int UPW = 5;    // hidden units per cell workgroup
constexpr int CNT_STRIDE = 32;
constexpr unsigned SPIN_LIMIT = 1u << 24;
constexpr int LDS_MAX = 160 * 1024;
#ifndef AWAIT_SLEEP
#define AWAIT_SLEEP 0
#endif
constexpr int SLOT_FLOATS = 2 * PW * 64 * 4;  // two buffers of one f32x4 per lane and wave
constexpr int SYNC_WORDS = 64;
constexpr int NPK = 8;    // arg-max records a lane picks from: at most 4 * NPK vocabulary tiles

typedef unsigned int u32x4v __attribute__((__vector_size__(16)));
typedef unsigned int u32x2v __attribute__((__vector_size__(8)));
typedef __attribute__((address_space(3))) volatile unsigned lds_word;

struct PdArgs {
    dssvae_decode_params p;
    const float* h0;
    int B, T, L;
    long long sos;
    int n_cell, n_proj;
    int nsx[DSSVAE_DECODE_MAX_LAYERS], kx[DSSVAE_DECODE_MAX_LAYERS];  // input slices / input parts of layer l
    int nsh, nse;      // slices of H, of E
    int w_floats;      // LDS floats reserved for the weight images
    unsigned* cnt;     // [L + 1][NBT] counters, CNT_STRIDE words apart: layer l's states, then the arg-max partials
    float* hx;         // [L][2][64 rows][nsh slices][16 units]
    float* prec;       // [2][64][NPK * 4] records (max, arg) of the vocabulary tiles: tile w at [w % 4][w / 4]
    long long* tokens;
    float* h_final;
    unsigned* err;
    unsigned* sticky;
};

struct Err {
    unsigned* err;
    unsigned* sticky;
};
__device__ __forceinline__ void raise_timeout(Err e) {
    __hip_atomic_store(e.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e.sticky) __hip_atomic_store(e.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int voff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 16));
}
__device__ __forceinline__ void publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void await(unsigned* cnt, unsigned target, Err err) {
    if ((threadIdx.x & 63) == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(AWAIT_SLEEP);
            if ((++spins & 1023u) == 0) {
                if (spins > SPIN_LIMIT || __hip_atomic_load(err.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    raise_timeout(err);
                    break;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void lds_signal(lds_word* w, unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *w = v;
}
__device__ __forceinline__ void lds_await(lds_word* w, unsigned target, Err err) {
    unsigned spins = 0;
    while (*w < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 4095u) == 0 &&
            (spins > SPIN_LIMIT || __hip_atomic_load(err.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            raise_timeout(err);
            break;
        }
    }
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ float sigm(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// acc += W[16 rows][16 k] . X[16 k][16 rows] over the n <= N slices of a part.  Wl: the LDS image per slice
// [4 c][4 kq][16 i], already offset by the lane's kq * 16 + i; the lane's values of slice i (k = 16 i + 4 kq + 0 .. 3)
// are the 16 bytes at voff + 64 i of resource r.  EVERY load of the part is issued before the first MFMA -- N
// unconditional loads off one address register and immediates (slices past n are fetched and not multiplied; past
// the resource they read as zero): one memory round trip per phase.  With register chunks of five slices and one
// chunk in flight a part of 19 slices took 5 us -- four dependent round trips around 0.3 us of MFMAs each.
constexpr int MAXS = 40;
template <int N, int AUX>
__device__ __forceinline__ void products_n(f32x4 (&acc)[2], const float* Wl, int n, __amdgpu_buffer_rsrc_t r, int voff) {
    f32x4 x[N];
#pragma unroll
    for (int i = 0; i < N; ++i)
        x[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 64 * i, 0, AUX));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (i < n) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                acc[c & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[i * 256 + c * 64], x[i][c], acc[c & 1], 0, 0, 0);
        }
    }
}
template <int AUX>
__device__ __forceinline__ void products(f32x4 (&acc)[2], const float* Wl, int n, __amdgpu_buffer_rsrc_t r, int voff) {
    if (n <= 0) return;
    if (n <= MAXS / 2) products_n<MAXS / 2, AUX>(acc, Wl, n, r, voff);
    else products_n<MAXS, AUX>(acc, Wl, n, r, voff);
}

// Development build only (-DPD_PROFILE, tools/decode_phases.py): 100 MHz stamps of the lead wave of batch tile 0 in
// three sampled workgroups (layer 0 with a projection tile, layer 0 without, the last layer)
#ifdef PD_PROFILE
constexpr int PD_STEPS = 64, PD_PH = 12;
__device__ unsigned long long g_pd_prof[3][PD_STEPS][PD_PH];
#define PDP(ph)                                                                                           \
    do {                                                                                                  \
        if (prof_slot >= 0 && t < PD_STEPS && lane == 0) g_pd_prof[prof_slot][t][ph] = wall_clock64();    \
    } while (0)
#else
#define PDP(ph) do {} while (0)
#endif

__global__ __launch_bounds__(64 * PW) void decode_persistent_kernel(PdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int layer = (int)blockIdx.x / a.n_cell, idx = (int)blockIdx.x % a.n_cell;
    const bool has_proj = layer == 0 && idx < a.n_proj;
    const int H = a.p.H, E = a.p.E, V = a.p.V, B = a.B, L = a.L, T = a.T;
    const int In = layer == 0 ? E : H;
    const int nsx = a.nsx[layer], nsh = a.nsh, nse = a.nse, nsc = nsx + nsh, nsp = nsh + nse;
    const int kx = a.kx[layer];
    const Err err = {a.err, a.sticky};
    float* Wc = lds;               // cell slice: [nsc][256]
    float* Wp = Wc + nsc * 256;    // projection tile: [nsp][256] (layer-0 workgroups below n_proj)
    f32x4* slots = reinterpret_cast<f32x4*>(lds + a.w_floats);  // [2][PW][64]
    lds_word* sync = (lds_word*)(slots + 2 * PW * 64);          // [PW] done words
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // batch tile and part of K: waves 0 .. 3 are the leads (part 0) of the four tiles, one per SIMD, each beside the
    // other part of its tile
    const int bt = wave % NBT, kp = wave / NBT;
    const bool lead = kp == 0;
    const int j = lane & 15, kq = lane >> 4;
    const int b = bt * 16 + j;

    // ---- weight images ----
    const int u0 = idx * UPW;
    for (int e = threadIdx.x; e < nsc * 256; e += 64 * PW) {
        const int s = e >> 8, c = (e >> 6) & 3, q = (e >> 4) & 3, i = e & 15;
        const int g = i / UPW, u = u0 + i % UPW;
        float w = 0.f;
        if (i < 3 * UPW && u < H) {
            if (s < nsx) {
                const int k = 16 * s + 4 * q + c;
                if (k < In) w = a.p.w_ih[layer][(size_t)(g * H + u) * In + k];
            } else {
                const int k = 16 * (s - nsx) + 4 * q + c;
                if (k < H) w = a.p.w_hh[layer][(size_t)(g * H + u) * H + k];
            }
        }
        Wc[e] = w;
    }
    if (has_proj) {
        for (int e = threadIdx.x; e < nsp * 256; e += 64 * PW) {
            const int s = e >> 8, c = (e >> 6) & 3, q = (e >> 4) & 3, i = e & 15;
            const int v = idx * 16 + i;
            float w = 0.f;
            if (v < V) {
                const int k = 16 * s + 4 * q + c;  // [h | e] of W_out's columns, each range padded to whole slices
                if (s < nsh) { if (k < H) w = a.p.w_out[(size_t)v * a.p.ld_out + k]; }
                else { const int ke = k - 16 * nsh; if (ke < E) w = a.p.w_out[(size_t)v * a.p.ld_out + H + ke]; }
            }
            Wp[e] = w;
        }
    }
    if (threadIdx.x < SYNC_WORDS) sync[threadIdx.x] = 0u;

    // ---- this wave's ranges ----
    // cell: parts below kx share the input slices, the others the recurrent slices
    int c_lo, c_n;  // first slice (in the image's order [x | h]) and count
    if (kp < kx) {
        const int per = (nsx + kx - 1) / kx;
        c_lo = min(nsx, kp * per); c_n = min(nsx - c_lo, per);
    } else {
        const int per = (nsh + (KS - kx) - 1) / (KS - kx), q = kp - kx;
        const int lo = min(nsh, q * per);
        c_lo = nsx + lo; c_n = min(nsh - lo, per);
    }
    // projection: part 0 the embedding slices (it holds the tokens), parts 1 .. KS-1 the state slices
    int p_lo, p_n;
    if (kp == 0) { p_lo = nsh; p_n = nse; }
    else {
        const int per = (nsh + KS - 2) / (KS - 1);
        p_lo = min(nsh, (kp - 1) * per); p_n = min(nsh - p_lo, per);
    }
    static_assert(PW == KS * NBT, "a wave per (batch tile, part)");
    const unsigned half_bytes = (unsigned)nsh * 64 * 16 * 4;
    auto hx_half = [&](int l, int half) { return a.hx + ((size_t)(l * 2 + half) * nsh) * 64 * 16; };
    const int voff = ((bt * 16 + j) * nsh * 16 + 4 * kq) * 4;  // the lane's 16 bytes of slice 0 of its row
    unsigned* cnt_own = a.cnt + (size_t)(layer * NBT + bt) * CNT_STRIDE;
    unsigned* cnt_below = a.cnt + (size_t)((layer > 0 ? layer - 1 : 0) * NBT + bt) * CNT_STRIDE;
    unsigned* cnt_top = a.cnt + (size_t)((L - 1) * NBT + bt) * CNT_STRIDE;
    unsigned* cnt_tok = a.cnt + (size_t)(L * NBT + bt) * CNT_STRIDE;
    const float* Wl_c = Wc + c_lo * 256 + kq * 16 + j;
    const float* Wl_p = Wp + p_lo * 256 + kq * 16 + j;

    // ---- the lead's cell state: lane (j, kq) owns unit u0 + kq of row b, and lanes with kq == 0 also unit u0 + 4 ----
    float hprev[2] = {0.f, 0.f}, bias[2][6];
    int un[2];
    un[0] = u0 + kq; un[1] = u0 + 4;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const bool ok = lead && un[ps] < H && (ps == 0 || kq == 0);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            bias[ps][g] = ok ? a.p.b_ih[layer][g * H + un[ps]] : 0.f;
            bias[ps][3 + g] = ok ? a.p.b_hh[layer][g * H + un[ps]] : 0.f;
        }
        if (ok && b < B) hprev[ps] = a.h0[((size_t)layer * B + b) * H + un[ps]];
    }
    auto store_state = [&](int half) {
        float* dst = hx_half(layer, half);
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            if (un[ps] < H && (ps == 0 || kq == 0)) {
                const __amdgpu_buffer_rsrc_t r = make_rsrc(dst, half_bytes);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hprev[ps]), r,
                                                      ((b * nsh + (un[ps] >> 4)) * 16 + (un[ps] & 15)) * 4, 0, 16);
            }
    };
    // the projection lead's bias values: vocabulary rows idx * 16 + 4 kq + c
    float pb[4] = {0.f, 0.f, 0.f, 0.f};
    if (has_proj && lead && a.p.b_out) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int v = idx * 16 + 4 * kq + c;
            if (v < V) pb[c] = a.p.b_out[v];
        }
    }
    __syncthreads();  // images and sync words are there; from here on the waves run free
    if (lead) {
        store_state(0);
        publish(cnt_own);
    }

    long long tok = a.sos;  // of row b, in the lanes of the waves that read embedding rows (layer 0, part 0)
    // best of the n_proj partials of row b left by step `st` (ties: the smaller index)
    auto pick = [&](int st) {
        // lane (j, kq) takes the tiles kq, kq + 4, ...: its NPK records lie side by side, all loads leave together
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.prec + (size_t)(st & 1) * 64 * NPK * 8, 64u * NPK * 8u * 4u);
        f32x4 rec[NPK / 2];
#pragma unroll
        for (int i = 0; i < NPK / 2; ++i) rec[i] = ld_sc1(rr, ((b * 4 + kq) * NPK * 2 + 4 * i) * 4);
        float m = -3.0e38f;
        int at = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NPK; ++i) {
            const float m2 = rec[i / 2][2 * (i & 1)];
            const int a2 = __builtin_bit_cast(int, rec[i / 2][2 * (i & 1) + 1]);
            if (kq + 4 * i < a.n_proj && (m2 > m || (m2 == m && a2 < at))) { m = m2; at = a2; }
        }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            const float m2 = __shfl_xor(m, o, 64);
            const int a2 = __shfl_xor(at, o, 64);
            if (m2 > m || (m2 == m && a2 < at)) { m = m2; at = a2; }
        }
        return (long long)(at == 0x7fffffff ? 0 : at);
    };
    // embedding rows through a resource over the whole table: a lane past its row's end reads its neighbour's values
    // (multiplied by the image's zero columns), past the table zeros
    const __amdgpu_buffer_rsrc_t r_tab =
        make_rsrc(a.p.embedding, (unsigned)((size_t)a.p.n_embeddings * a.p.ld_embedding * 4));
    const int ld_tab = (int)a.p.ld_embedding;
#ifdef PD_PROFILE
    const int prof_slot = wave != 0 ? -1 : ((int)blockIdx.x == 0 ? 0 : (int)blockIdx.x == a.n_cell - 1 ? 1
                                            : (int)blockIdx.x == (L - 1) * a.n_cell && L > 1 ? 2 : -1);
#endif
    unsigned seq = 0;  // phases this workgroup has been through (cell, projection, cell, ...)
    for (int t = 0; t < T; ++t) {
        PDP(0);
        const int rh = t & 1, wh = rh ^ 1;  // the state before step t lives in half rh, the state after it in wh
        // ================= cell =================
        {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (kp < kx) {
                if (layer == 0) {
                    if (t > 0) {
                        await(cnt_tok, (unsigned)t * a.n_proj, err);
                        PDP(1);
                        tok = pick(t - 1);
                        if (idx == 0 && kq == 0 && b < B) a.tokens[(size_t)(t - 1) * B + b] = tok;
                    }
#ifdef PD_PROFILE
                    asm volatile("s_nop 0" ::"v"(tok));
#endif
                    PDP(2);
                    products<0>(acc, Wl_c, c_n, r_tab, ((int)tok * ld_tab + 16 * c_lo + 4 * kq) * 4);
                } else {
                    await(cnt_below, (unsigned)(t + 2) * a.n_cell, err);
                    PDP(1);
                    PDP(2);
                    products<16>(acc, Wl_c, c_n, make_rsrc(hx_half(layer - 1, wh), half_bytes), voff + c_lo * 64);
                }
            } else {
                await(cnt_own, (unsigned)(t + 1) * a.n_cell, err);
                products<16>(acc, Wl_c, c_n, make_rsrc(hx_half(layer, rh), half_bytes), voff + (c_lo - nsx) * 64);
            }
            f32x4* buf = slots + (seq & 1) * PW * 64;
            buf[(bt * KS + kp) * 64 + lane] = acc[0] + acc[1];
            lds_signal(sync + wave, seq + 1u);
            PDP(3);
            if (lead) {
                for (int p = 1; p < KS; ++p) lds_await(sync + p * NBT + bt, seq + 1u, err);
                PDP(4);
                const float* sf = reinterpret_cast<const float*>(buf + bt * KS * 64);  // [part][64 lanes][4]
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) {
                    const int uu = ps == 0 ? kq : 4;
                    float sx[3] = {0.f, 0.f, 0.f}, sh[3] = {0.f, 0.f, 0.f};
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        const int rho = g * UPW + uu, at = ((rho >> 2) * 16 + j) * 4 + (rho & 3);
#pragma unroll
                        for (int p = 0; p < KS; ++p) {
                            const float v = sf[p * 256 + at];
                            if (p < kx) sx[g] += v; else sh[g] += v;
                        }
                    }
                    const float r = sigm(sx[0] + sh[0] + bias[ps][0] + bias[ps][3]);
                    const float z = sigm(sx[1] + sh[1] + bias[ps][1] + bias[ps][4]);
                    const float n = tanh_fast(sx[2] + bias[ps][2] + r * (sh[2] + bias[ps][5]));
                    hprev[ps] = (1.f - z) * n + z * hprev[ps];
                }
#ifdef PD_PROFILE
                asm volatile("s_nop 0" ::"v"(hprev[0]), "v"(hprev[1]));
#endif
                PDP(5);
                store_state(wh);
#ifdef PD_PROFILE
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                PDP(6);
#endif
                publish(cnt_own);
                PDP(7);
            }
            ++seq;
        }
        // ================= projection + arg-max partials =================
        if (has_proj) {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (kp == 0) {
                products<0>(acc, Wl_p, p_n, r_tab, ((int)tok * ld_tab + 4 * kq) * 4);
            } else {
                await(cnt_top, (unsigned)(t + 2) * a.n_cell, err);
                products<16>(acc, Wl_p, p_n, make_rsrc(hx_half(L - 1, wh), half_bytes), voff + p_lo * 64);
            }
            f32x4* buf = slots + (seq & 1) * PW * 64;
            buf[(bt * KS + kp) * 64 + lane] = acc[0] + acc[1];
            lds_signal(sync + wave, seq + 1u);
            PDP(8);
            if (lead) {
                f32x4 sum = acc[0] + acc[1];
                for (int p = 1; p < KS; ++p) {
                    lds_await(sync + p * NBT + bt, seq + 1u, err);
                    sum += buf[(bt * KS + p) * 64 + lane];
                }
#ifdef PD_PROFILE
                asm volatile("s_nop 0" ::"v"(sum));
#endif
                PDP(9);
                // the lane holds vocabulary rows idx * 16 + 4 kq + c of sentence b
                float m = -3.0e38f;
                int at = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int v = idx * 16 + 4 * kq + c;
                    if (v < V) {
                        const float x = sum[c] + pb[c];
                        if (x > m) { m = x; at = v; }
                    }
                }
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float m2 = __shfl_xor(m, o, 64);
                    const int a2 = __shfl_xor(at, o, 64);
                    if (m2 > m || (m2 == m && a2 < at)) { m = m2; at = a2; }
                }
                if (kq == 0) {
                    const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.prec + (size_t)(t & 1) * 64 * NPK * 8, 64u * NPK * 8u * 4u);
                    __builtin_amdgcn_raw_buffer_store_b64(
                        (u32x2v){__builtin_bit_cast(unsigned, m), (unsigned)at}, rr,
                        (((b * 4 + (idx & 3)) * NPK + (idx >> 2)) * 2) * 4, 0, 16);
                }
                publish(cnt_tok);
                PDP(10);
            }
            ++seq;
        }
    }
    // the last step's tokens, the final states
    if (layer == 0 && idx == 0 && lead) {
        await(cnt_tok, (unsigned)T * a.n_proj, err);
        tok = pick(T - 1);
        if (kq == 0 && b < B) a.tokens[(size_t)(T - 1) * B + b] = tok;
    }
    if (lead && a.h_final && b < B) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps)
            if (un[ps] < H && (ps == 0 || kq == 0)) a.h_final[((size_t)layer * B + b) * H + un[ps]] = hprev[ps];
    }
}

struct Plan {
    int n_cell, n_proj, nsh, nse, nsx[DSSVAE_DECODE_MAX_LAYERS], kx[DSSVAE_DECODE_MAX_LAYERS];
    int w_floats, lds_bytes, grid;
    size_t cnt_off, hx_off, pr_off, bytes;
};

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// DSSVAE_OK and the plan, or DSSVAE_ETOOBIG when the shape is outside this kernel's range
int make_plan(int B, int E, int H, int V, int L, Plan* pl) {
    if (B > 16 * NBT || L > DSSVAE_DECODE_MAX_LAYERS || (E % 4) != 0 || (H % 4) != 0 || E < 4 || H < 4)
        return DSSVAE_ETOOBIG;
    Plan p;
    p.n_cell = (H + UPW - 1) / UPW;
    p.n_proj = (V + 15) / 16;
    p.nsh = (H + 15) / 16;
    p.nse = (E + 15) / 16;
    if (p.n_proj > p.n_cell || p.n_proj > 4 * NPK || p.nse > MAXS || (p.nsh + KS - 2) / (KS - 1) > MAXS)
        return DSSVAE_ETOOBIG;
    int slices = 0;
    for (int l = 0; l < L; ++l) {
        p.nsx[l] = l == 0 ? p.nse : p.nsh;
        const int nsc = p.nsx[l] + p.nsh;
        p.kx[l] = 1;  // part 0 the input range, part 1 the recurrent range
        if ((p.nsx[l] + p.kx[l] - 1) / p.kx[l] > MAXS || (p.nsh + KS - p.kx[l] - 1) / (KS - p.kx[l]) > MAXS)
            return DSSVAE_ETOOBIG;  // a part's slices are loaded in one go
        const int need = nsc + (l == 0 ? p.nsh + p.nse : 0);
        if (need > slices) slices = need;
    }
    p.w_floats = slices * 256;
    p.lds_bytes = (p.w_floats + SLOT_FLOATS + SYNC_WORDS) * 4;
    p.grid = L * p.n_cell;
    if (p.lds_bytes > LDS_MAX) return DSSVAE_ETOOBIG;
    size_t off = 256;  // error word
    p.cnt_off = off; off += up256((size_t)(L + 1) * NBT * CNT_STRIDE * 4);
    p.hx_off = off; off += up256((size_t)L * 2 * p.nsh * 64 * 16 * 4);
    p.pr_off = off; off += up256((size_t)2 * 64 * NPK * 4 * 8);
    p.bytes = off;
    *pl = p;
    return DSSVAE_OK;
}

// every workgroup of the launch must be resident at once: the runtime's occupancy answer x the CU count
int resident_limit(int lds_bytes) {
    static int cached_lds = -1, cached = 0;
    if (cached_lds == lds_bytes) return cached;
    int per_cu = 0, cus = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(decode_persistent_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, decode_persistent_kernel, 64 * PW, (size_t)lds_bytes) !=
            hipSuccess ||
        dssvae_device_cus(&cus) != 0)
        return 0;
    cached_lds = lds_bytes;
    cached = per_cu * cus;
    return cached;
}

}  // namespace

extern "C" int64_t dssvae_greedy_decode_persistent_workspace_bytes(int B, int E, int H, int n_layers, int V) {
    if (B <= 0 || E <= 0 || H <= 0 || V <= 0 || n_layers <= 0) return DSSVAE_EINVAL;
    Plan pl;
    const int rc = make_plan(B, E, H, V, n_layers, &pl);
    if (rc != DSSVAE_OK) return rc;
    return (int64_t)pl.bytes;
}

extern "C" int dssvae_greedy_decode_persistent(const dssvae_decode_params* p, const float* h0, int B, int max_len,
                                               int64_t sos_id, int64_t* tokens, float* h_final, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
    if (!p || !h0 || !tokens || !workspace || B <= 0 || max_len <= 0) return DSSVAE_EINVAL;
    const int E = p->E, H = p->H, L = p->n_layers, V = p->V;
    if (E <= 0 || H <= 0 || V <= 0 || L <= 0 || L > DSSVAE_DECODE_MAX_LAYERS || !p->embedding || !p->w_out ||
        p->ld_embedding < E || p->ld_out < H + E || sos_id < 0 || sos_id >= p->n_embeddings || p->n_embeddings < V)
        return DSSVAE_EINVAL;
    for (int l = 0; l < L; ++l)
        if (!p->w_ih[l] || !p->w_hh[l] || !p->b_ih[l] || !p->b_hh[l]) return DSSVAE_EINVAL;
    if (!aligned16(workspace)) return DSSVAE_EINVAL;
    Plan pl;
    int rc = make_plan(B, E, H, V, L, &pl);
    if (rc != DSSVAE_OK) return rc;
    // 16-byte loads of embedding rows
    if (!aligned16(p->embedding) || (p->ld_embedding % 4) != 0 ||
        (int64_t)p->n_embeddings * p->ld_embedding * 4 >= ((int64_t)1 << 31))
        return DSSVAE_ETOOBIG;
    if ((int64_t)pl.bytes > workspace_bytes) return DSSVAE_EINVAL;
    if (pl.grid > resident_limit(pl.lds_bytes)) return DSSVAE_ETOOBIG;
    hipStream_t s = (hipStream_t)stream;
    // counters, partials and the pad units of the state buffers start at zero
    rc = dssvae_zero2d(reinterpret_cast<float*>(workspace), (int64_t)(pl.bytes / 4), 1, (int)(pl.bytes / 4), s);
    if (rc != DSSVAE_OK) return rc;
    PdArgs a;
    a.p = *p;
    a.h0 = h0;
    a.B = B; a.T = max_len; a.L = L;
    a.sos = sos_id;
    a.n_cell = pl.n_cell; a.n_proj = pl.n_proj;
    for (int l = 0; l < DSSVAE_DECODE_MAX_LAYERS; ++l) { a.nsx[l] = l < L ? pl.nsx[l] : 0; a.kx[l] = l < L ? pl.kx[l] : 1; }
    a.nsh = pl.nsh; a.nse = pl.nse;
    a.w_floats = pl.w_floats;
    char* base = static_cast<char*>(workspace);
    a.err = reinterpret_cast<unsigned*>(base);
    a.sticky = dssvae_sticky_word_of_current_device();
    a.cnt = reinterpret_cast<unsigned*>(base + pl.cnt_off);
    a.hx = reinterpret_cast<float*>(base + pl.hx_off);
    a.prec = reinterpret_cast<float*>(base + pl.pr_off);
    a.tokens = reinterpret_cast<long long*>(tokens);
    a.h_final = h_final;
    hipLaunchKernelGGL(decode_persistent_kernel, dim3(pl.grid), dim3(64 * PW), (size_t)pl.lds_bytes, s, a);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

#ifdef PD_PROFILE
extern "C" int dssvae_decode_persistent_profile_read(void* dst_host, int64_t bytes) {
    if (bytes > (int64_t)sizeof(g_pd_prof)) bytes = sizeof(g_pd_prof);
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_pd_prof), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
#endif
