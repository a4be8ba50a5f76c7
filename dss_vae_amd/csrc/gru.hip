// Persistent GRU recurrence for gfx950 (forward and backward through time).
//
// Decomposition: an "instance" is one independent recurrence (a direction of an encoder, a decoder
// layer, a task head).  Each instance is cut into (batch block of 64 rows) x (slice of 16 hidden
// units); one 256-thread workgroup owns one cell of that grid for ALL timesteps:
//   * its W_hh slice (3 gates x 16 units x H, fp32) is loaded into LDS once and stays resident;
//   * per step each of its 4 waves multiplies 16 batch rows of h_{t-1} (B operand) with the
//     resident slice (A operand) on v_mfma_f32_16x16x4_f32, then applies the fused gate math in
//     registers and stores its 16x16 piece of h_t;
//   * the state is exchanged through a ping-pong buffer laid out [slice][row][16 units], so a
//     workgroup publishes ONE contiguous 4 KB block and a reading wave takes 1 KB contiguous per
//     slice; the waves that own the same 16 batch rows in the workgroups of one (instance, batch
//     block) meet on one monotonic arrival counter per step -- there is no grid-wide barrier and no
//     workgroup barrier in the step loop, and independent instances / batch blocks / row groups never
//     wait for each other.
// Cross-workgroup visibility (MI355X_MICROARCH.md, "Valid forms", first table row): EVERY byte of
// the exchange buffer is stored write-through (buffer_store_dwordx4 sc1) and EVERY load of it is a
// buffer_load_dwordx4 sc1 (bypasses the CU's L1), so no release / acquire cache maintenance is
// needed: a storing wave drains its stores (s_waitcnt vmcnt(0)), then its lane 0 adds to the row
// group's counter (agent-scope atomic); a consuming wave's lane 0 polls that counter with sc1 loads,
// then the wave issues its sc1 loads (every byte it reads was stored by a wave that signalled after
// its own drain).
// The exchange loads go through a buffer resource that ends with the batch block's half of the ping-pong, with the
// slice offset in the VGPR offset: a load past the last slice is out of range and returns zeros without a fetch
// (the prefetch chunks need no clamping -- clamped duplicates cost 6 % of a backward launch).
// All workgroups of a launch must be co-resident: the host caps the grid at the CU count (1
// workgroup per CU); every spin is bounded and reports through an error word instead of hanging.
//
// Roofline: per step 2*B*H*3H flop on fp32 MFMA; in practice latency-bound by the step hand-off.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int UU = 16;   // hidden units per workgroup (M of the 16x16x4 MFMA)
constexpr int CH = 8;    // exchange slices fetched per register chunk (forward)
constexpr int CHB = 6;   // ... backward (measured: 8 -> 6 takes 6 % off the backward launches, forward loses 3 %)
constexpr unsigned SPIN_LIMIT = 1u << 24;
constexpr int CNT_STRIDE = 32;  // one 128-B line per counter
constexpr int LDS_BUDGET = 150 * 1024;  // W_hh slice of a workgroup
constexpr int MAX_WG_PER_CU = 4;

typedef unsigned int u32x4v __attribute__((__vector_size__(16)));

struct FwdDev {
    dssvae_gru_fwd_inst in;
    int wg_begin, n_slices, n_bblocks, rows;  // rows: batch rows per cell of this instance (16, 32, 64 or 128)
    unsigned* counters;
    float* hx;  // [2][n_bblocks][n_slices][64][16]
};
struct FwdArgs {
    FwdDev inst[DSSVAE_GRU_MAX_INST];
    int n_inst;
    unsigned* err;
};
struct BwdDev {
    dssvae_gru_bwd_inst in;
    int wg_begin, n_slices, n_bblocks, rows;
    unsigned* counters;
    float* gx;  // [2][n_bblocks][n_slices][3][64][16]
};
struct BwdArgs {
    BwdDev inst[DSSVAE_GRU_MAX_INST];
    int n_inst;
    unsigned* err;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16));
}
__device__ __forceinline__ void st_sc1(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, voff, soff, 16);
}

// Hand-off at WAVE granularity: wave w of a workgroup owns batch rows 16w..16w+15 of its block and needs,
// from every peer slice, exactly the rows that the peers' wave w produced.  So each (batch block, wave)
// row group has its own arrival counter: a wave drains its own write-through stores and signals; a wave
// polls its own counter and then loads.  No workgroup barrier in the step loop: the waves of a workgroup
// (and the two waves that share a SIMD in the 8-wave configuration) drift apart, one's MFMAs run under
// the other's hand-off.
__device__ __forceinline__ void publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane 0 polls (sc1 loads) while the rest of the wave is masked off; the payload loads that follow are all sc1
__device__ __forceinline__ void await(unsigned* cnt, unsigned target, unsigned* err) {
    if ((threadIdx.x & 63) == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(0);
            if ((++spins & 1023u) == 0) {
                if (spins > SPIN_LIMIT ||
                    __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + __expf(-x)); }

__device__ __forceinline__ void load4(float (&v)[4], const float* p, bool vec, int n_ok) {
    if (vec && n_ok >= 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = c < n_ok ? p[c] : 0.f;
    }
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4], bool vec, int n_ok) {
    if (vec && n_ok >= 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < n_ok) p[c] = v[c];
    }
}

// acc[g] += W_g[16 units][16 k of chunk] . X[16 k][16 rows]; Wc: this chunk's LDS image [4 c][4 kq][16 j]
// xv: the lane's 4 k-values (k = 4*kq + c) of its batch row
template <int NG>
__device__ __forceinline__ void mfma_chunk(f32x4 (&acc)[NG], const float* Wl, int gate_stride, f32x4 xv) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int g = 0; g < NG; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[g * gate_stride + c * 64], xv[c], acc[g], 0, 0, 0);
}

// ------------------------------------------------------------------------------------------
// WV waves per workgroup: 4 (64 batch rows) or 8 (128 rows: half the workgroups per recurrence, two waves per
// SIMD share the resident slice -- what lets every decoder of a layer share ONE co-resident launch)
template <int WV>
__global__ __launch_bounds__(64 * WV) void gru_fwd_kernel(FwdArgs args) {
    constexpr int NTHR = 64 * WV;
    // Ws[g][slice s][c][kq][j] = whh[(g*H + u0 + j)*H + 16 s + 4 kq + c]
    extern __shared__ __attribute__((aligned(16))) float Ws[];
    int ii = 0;
    while (ii + 1 < args.n_inst && (int)blockIdx.x >= args.inst[ii + 1].wg_begin) ++ii;
    const FwdDev& d = args.inst[ii];
    const dssvae_gru_fwd_inst& in = d.in;
    const int local = blockIdx.x - d.wg_begin;
    const int bblock = local / d.n_slices, slice = local % d.n_slices;
    const int H = in.H, T = in.T, B = in.B, NS = d.n_slices;
    // cells of this instance hold BB batch rows (<= 16 WV): waves beyond BB / 16 only help to fill LDS
    const int BB = d.rows, WVI = BB >> 4, BLK = BB * UU;
    const int Tt = in.T_total > 0 ? in.T_total : T;   // row stride of the [B, T_total, .] histories
    const int u0 = slice * UU;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int b = bblock * BB + wave * 16 + j;
    const bool row_ok = b < B;
    const int ub = u0 + 4 * kq;                       // first of this lane's 4 units
    const int n_ok = max(0, min(4, H - ub));          // how many of them exist
    unsigned* cnt = d.counters + ((size_t)bblock * WVI + wave) * CNT_STRIDE;  // this wave's row group
    const bool vec_h = (H % 4) == 0;
    const int gate_stride = NS * 256;

    for (int idx = threadIdx.x; idx < 3 * NS * 256; idx += NTHR) {
        const int g = idx / (NS * 256), rem = idx % (NS * 256);
        const int jj = rem / (NS * 16), k = rem % (NS * 16);  // k fastest: coalesced global reads
        const int u = u0 + jj;
        const float w = (u < H && k < H) ? in.whh[(size_t)(g * H + u) * H + k] : 0.f;
        const int s = k >> 4, kk = k & 15;
        Ws[g * gate_stride + s * 256 + (kk & 3) * 64 + (kk >> 2) * 16 + jj] = w;
    }
    float bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) bh[g][c] = (c < n_ok) ? in.bhh[g * H + ub + c] : 0.f;
    const int len = (in.lengths && row_ok) ? in.lengths[b] : Tt;

    // exchange buffers of this batch block: [2][NS][64][16]; this lane's piece of its own block
    const unsigned half_bytes = (unsigned)NS * BLK * 4u;
    float* hx0 = d.hx + (size_t)bblock * NS * BLK;
    float* hx1 = hx0 + (size_t)d.n_bblocks * NS * BLK;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(hx0, half_bytes), rs1 = make_rsrc(hx1, half_bytes);
    const int voff = ((wave * 16 + j) * UU + 4 * kq) * 4;  // same lane offset inside every slice block
    const int own = slice * BLK * 4;

    __syncthreads();  // the slice is in LDS for every wave; from here on the waves run free
    if (wave >= WVI) return;
    // state 0 into the exchange buffer (pad units and pad rows are zero and stay zero)
    float hcur[4] = {0.f, 0.f, 0.f, 0.f};
    if (row_ok && in.h0) load4(hcur, in.h0 + (size_t)b * H + ub, vec_h, n_ok);
    st_sc1((f32x4){hcur[0], hcur[1], hcur[2], hcur[3]}, rs0, voff, own);
    publish(cnt);

    const float* Wl = Ws + kq * 16 + j;
    for (int step = 0; step < T; ++step) {
        const int gs = in.t0 + step;  // step of the whole recurrence (this launch runs steps t0 .. t0 + T - 1)
        const int t = in.reverse ? Tt - 1 - gs : gs;
        const __amdgpu_buffer_rsrc_t rin = (step & 1) ? rs1 : rs0, rout = (step & 1) ? rs0 : rs1;
        // the x-projection gates of this step do not depend on the hand-off: fetch them first
        const size_t bt = (size_t)(row_ok ? b : 0) * Tt + t;
        const float* xp = in.xproj + bt * in.ldx;
        float xr[4], xz[4], xn[4];
        const bool vec_x = vec_h && (in.ldx % 4) == 0;
        load4(xr, xp + ub, vec_x, n_ok);
        load4(xz, xp + H + ub, vec_x, n_ok);
        load4(xn, xp + 2 * H + ub, vec_x, n_ok);

        await(cnt, (unsigned)(step + 1) * NS, args.err);

        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 cur[CH], nxt[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) cur[i] = ld_sc1(rin, voff + i * BLK * 4, 0);  // past NS: out of range, no fetch
        for (int s0 = 0; s0 < NS; s0 += CH) {
            if (s0 + CH < NS) {
#pragma unroll
                for (int i = 0; i < CH; ++i) nxt[i] = ld_sc1(rin, voff + (s0 + CH + i) * BLK * 4, 0);
            }
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (s0 + i < NS) mfma_chunk<3>(acc, Wl + (s0 + i) * 256, gate_stride, cur[i]);
#pragma unroll
            for (int i = 0; i < CH; ++i) cur[i] = nxt[i];
        }
        // fused gate math: lane holds units ub..ub+3 of batch row b (pad units keep writing zeros)
        const bool live = row_ok && t < len;
        float r[4], z[4], n[4], hn[4], hnew[4], outv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            r[c] = sigm(xr[c] + acc[0][c] + bh[0][c]);
            z[c] = sigm(xz[c] + acc[1][c] + bh[1][c]);
            hn[c] = acc[2][c] + bh[2][c];
            n[c] = tanhf(xn[c] + r[c] * hn[c]);
            const float hv = (1.f - z[c]) * n[c] + z[c] * hcur[c];
            hnew[c] = (c < n_ok && row_ok) ? (live ? hv : hcur[c]) : 0.f;
            outv[c] = live ? hv : 0.f;
        }
        // hand the new state over first; the history stores below then overlap the peers' next step
        if (step + 1 < T) {
            st_sc1((f32x4){hnew[0], hnew[1], hnew[2], hnew[3]}, rout, voff, own);
            publish(cnt);
        }
        if (row_ok) {
            store4(in.out + bt * in.ldo + ub, outv, vec_h && (in.ldo % 4) == 0, n_ok);
            if (in.gates) {
                float* gs = in.gates + bt * 5 * H + ub;
                store4(gs, r, vec_h, n_ok);
                store4(gs + H, z, vec_h, n_ok);
                store4(gs + 2 * H, n, vec_h, n_ok);
                store4(gs + 3 * H, hn, vec_h, n_ok);
                store4(gs + 4 * H, hcur, vec_h, n_ok);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) hcur[c] = hnew[c];
    }
    if (row_ok && n_ok > 0 && in.h_last) store4(in.h_last + (size_t)b * H + ub, hcur, vec_h, n_ok);
}

// ------------------------------------------------------------------------------------------
// WV waves per workgroup: 4 (64 batch rows) or 8 (128 rows: half the workgroups per recurrence, two waves per
// SIMD share the resident slice -- what lets every decoder of a layer share ONE co-resident launch)
template <int WV>
__global__ __launch_bounds__(64 * WV) void gru_bwd_kernel(BwdArgs args) {
    constexpr int NTHR = 64 * WV;
    // chunk q = 3*p + g covers W_hh rows g*H + 16p .. +15; Ws[q][c][kq][i] = whh[g*H + 16p + 4kq + c][u0 + i]
    extern __shared__ __attribute__((aligned(16))) float Ws[];
    int ii = 0;
    while (ii + 1 < args.n_inst && (int)blockIdx.x >= args.inst[ii + 1].wg_begin) ++ii;
    const BwdDev& d = args.inst[ii];
    const dssvae_gru_bwd_inst& in = d.in;
    const int local = blockIdx.x - d.wg_begin;
    const int bblock = local / d.n_slices, slice = local % d.n_slices;
    const int H = in.H, T = in.T, B = in.B, K = 3 * H, NS = d.n_slices, NQ = 3 * NS;
    const int BB = d.rows, WVI = BB >> 4, BLK = BB * UU;
    const int Tt = in.T_total > 0 ? in.T_total : T;
    const int u0 = slice * UU;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int b = bblock * BB + wave * 16 + j;
    const bool row_ok = b < B;
    const int ub = u0 + 4 * kq;
    const int n_ok = max(0, min(4, H - ub));
    unsigned* cnt = d.counters + ((size_t)bblock * WVI + wave) * CNT_STRIDE;  // this wave's row group
    const bool vec_h = (H % 4) == 0;

    for (int idx = threadIdx.x; idx < NQ * 256; idx += NTHR) {
        const int q = idx >> 8, rem = idx & 255;
        const int kk = rem >> 4, i = rem & 15;  // i fastest: 16 consecutive floats of a W_hh row
        const int p = q / 3, g = q - 3 * p;
        const int cu = 16 * p + kk, u = u0 + i;
        const float w = (cu < H && u < H) ? in.whh[(size_t)(g * H + cu) * H + u] : 0.f;
        Ws[q * 256 + (kk & 3) * 64 + (kk >> 2) * 16 + i] = w;
    }
    const int len = (in.lengths && row_ok) ? in.lengths[b] : Tt;
    float dh[4] = {0.f, 0.f, 0.f, 0.f};
    if (row_ok && in.dh_last) load4(dh, in.dh_last + (size_t)b * H + ub, vec_h, n_ok);

    const unsigned half_bytes = (unsigned)NQ * BLK * 4u;
    float* gx0 = d.gx + (size_t)bblock * NQ * BLK;
    float* gx1 = gx0 + (size_t)d.n_bblocks * NQ * BLK;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(gx0, half_bytes), rs1 = make_rsrc(gx1, half_bytes);
    const int voff = ((wave * 16 + j) * UU + 4 * kq) * 4;
    const int own = slice * 3 * BLK * 4;
    __syncthreads();
    if (wave >= WVI) return;

    const float* Wl = Ws + kq * 16 + j;
    // bias gradients (column sums of dgx / dgh over batch and time) ride along in registers
    float sb_r[4] = {0.f, 0.f, 0.f, 0.f}, sb_z[4] = {0.f, 0.f, 0.f, 0.f}, sb_n[4] = {0.f, 0.f, 0.f, 0.f},
          sb_m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int step = 0; step < T; ++step) {
        // backward walks the forward order in reverse: forward steps t0 + T - 1 down to t0 of the recurrence
        const int fstep = in.t0 + T - 1 - step;
        const int t = in.reverse ? Tt - 1 - fstep : fstep;
        const __amdgpu_buffer_rsrc_t rcur = (step & 1) ? rs1 : rs0;
        float dhz[4] = {0.f, 0.f, 0.f, 0.f};
        float dgr[4] = {0.f, 0.f, 0.f, 0.f}, dgz[4] = {0.f, 0.f, 0.f, 0.f}, dgn[4] = {0.f, 0.f, 0.f, 0.f},
              dgm[4] = {0.f, 0.f, 0.f, 0.f};
        const size_t bt = (size_t)(row_ok ? b : 0) * Tt + t;
        if (row_ok && n_ok > 0) {
            const bool live = t < len;
            if (live) {
                float go[4] = {0.f, 0.f, 0.f, 0.f};
                if (in.dout) load4(go, in.dout + bt * in.ldd + ub, vec_h && (in.ldd % 4) == 0, n_ok);
                const float* gs = in.gates + bt * 5 * H + ub;
                float r[4], z[4], n[4], hn[4], hp[4];
                load4(r, gs, vec_h, n_ok);
                load4(z, gs + H, vec_h, n_ok);
                load4(n, gs + 2 * H, vec_h, n_ok);
                load4(hn, gs + 3 * H, vec_h, n_ok);
                load4(hp, gs + 4 * H, vec_h, n_ok);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float g = dh[c] + go[c];
                    const float dn = g * (1.f - z[c]) * (1.f - n[c] * n[c]);
                    const bool ok = c < n_ok;
                    dgn[c] = ok ? dn : 0.f;
                    dgz[c] = ok ? g * (hp[c] - n[c]) * z[c] * (1.f - z[c]) : 0.f;
                    dgr[c] = ok ? dn * hn[c] * r[c] * (1.f - r[c]) : 0.f;
                    dgm[c] = ok ? dn * r[c] : 0.f;
                    dhz[c] = g * z[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) dhz[c] = dh[c];
            }
        }
        // this slice's three gate blocks (pad units / pad rows publish zeros); hand over first, the
        // dgx / dgh history stores then overlap the wait
#pragma unroll
        for (int c = 0; c < 4; ++c) { sb_r[c] += dgr[c]; sb_z[c] += dgz[c]; sb_n[c] += dgn[c]; sb_m[c] += dgm[c]; }
        st_sc1((f32x4){dgr[0], dgr[1], dgr[2], dgr[3]}, rcur, voff, own);
        st_sc1((f32x4){dgz[0], dgz[1], dgz[2], dgz[3]}, rcur, voff, own + BLK * 4);
        st_sc1((f32x4){dgm[0], dgm[1], dgm[2], dgm[3]}, rcur, voff, own + 2 * BLK * 4);
        publish(cnt);
        if (row_ok && n_ok > 0) {
            float* ox = in.dgx + bt * K + ub;
            float* oh = in.dgh + bt * K + ub;
            store4(ox, dgr, vec_h, n_ok); store4(ox + H, dgz, vec_h, n_ok); store4(ox + 2 * H, dgn, vec_h, n_ok);
            store4(oh, dgr, vec_h, n_ok); store4(oh + H, dgz, vec_h, n_ok); store4(oh + 2 * H, dgm, vec_h, n_ok);
        }
        await(cnt, (unsigned)(step + 1) * NS, args.err);

        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 cur[CHB], nxt[CHB];
#pragma unroll
        for (int i = 0; i < CHB; ++i) cur[i] = ld_sc1(rcur, voff + i * BLK * 4, 0);  // past NQ: out of range, no fetch
        for (int q0 = 0; q0 < NQ; q0 += CHB) {
            if (q0 + CHB < NQ) {
#pragma unroll
                for (int i = 0; i < CHB; ++i) nxt[i] = ld_sc1(rcur, voff + (q0 + CHB + i) * BLK * 4, 0);
            }
#pragma unroll
            for (int i = 0; i < CHB; ++i)
                if (q0 + i < NQ) {
                    // three independent accumulators in rotation hide the 40-cycle dependent latency
                    const float* w = Wl + (q0 + i) * 256;
                    f32x4& a = acc[i % 3];
#pragma unroll
                    for (int c = 0; c < 4; ++c) a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c * 64], cur[i][c], a, 0, 0, 0);
                }
#pragma unroll
            for (int i = 0; i < CHB; ++i) cur[i] = nxt[i];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) dh[c] = dhz[c] + acc[0][c] + acc[1][c] + acc[2][c];
    }
    if (row_ok && n_ok > 0 && in.dh0) store4(in.dh0 + (size_t)b * H + ub, dh, vec_h, n_ok);
    if (in.dbias_x || in.dbias_h) {
        // sum over the 16 batch rows of the wave (lanes with the same unit quad), then one atomic per unit
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                sb_r[c] += __shfl_xor(sb_r[c], o, 64); sb_z[c] += __shfl_xor(sb_z[c], o, 64);
                sb_n[c] += __shfl_xor(sb_n[c], o, 64); sb_m[c] += __shfl_xor(sb_m[c], o, 64);
            }
        }
        if (j == 0) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < n_ok) {
                    const int u = ub + c;
                    if (in.dbias_x) {
                        atomicAdd(in.dbias_x + u, sb_r[c]); atomicAdd(in.dbias_x + H + u, sb_z[c]);
                        atomicAdd(in.dbias_x + 2 * H + u, sb_n[c]);
                    }
                    if (in.dbias_h) {
                        atomicAdd(in.dbias_h + u, sb_r[c]); atomicAdd(in.dbias_h + H + u, sb_z[c]);
                        atomicAdd(in.dbias_h + 2 * H + u, sb_m[c]);
                    }
                }
        }
    }
}

struct Layout {
    size_t header_bytes, total_bytes;
    size_t cnt_off[DSSVAE_GRU_MAX_INST], buf_off[DSSVAE_GRU_MAX_INST];
};

inline int round16(int x) { return (x + 15) & ~15; }

// exchange buffers are sized for the backward pass (3 gate blocks per slice); forward uses a third.  The sizes
// hold for every cell height (16 .. 128 rows): counters are per 16-row group, buffers per 128-row-padded batch.
Layout plan(int n_inst, const int32_t* B, const int32_t* H) {
    Layout L;
    size_t off = 256;  // error word + padding
    for (int i = 0; i < n_inst; ++i) {
        L.cnt_off[i] = off;
        off += (size_t)((B[i] + 127) / 128) * 8 * CNT_STRIDE * sizeof(unsigned);  // one per 16-row group
    }
    off = (off + 255) & ~(size_t)255;
    L.header_bytes = off;
    for (int i = 0; i < n_inst; ++i) {
        L.buf_off[i] = off;
        off += (size_t)2 * ((B[i] + 127) / 128) * 128 * 3 * round16(H[i]) * sizeof(float);  // either block size
        off = (off + 255) & ~(size_t)255;
    }
    L.total_bytes = off;
    return L;
}

int cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int n = 0;
        if (dssvae_device_cus(&n) != 0 || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}

const void* fwd_kernel(int wv) {
    switch (wv) {
        case 1: return reinterpret_cast<const void*>(gru_fwd_kernel<1>);
        case 2: return reinterpret_cast<const void*>(gru_fwd_kernel<2>);
        case 4: return reinterpret_cast<const void*>(gru_fwd_kernel<4>);
        default: return reinterpret_cast<const void*>(gru_fwd_kernel<8>);
    }
}
const void* bwd_kernel(int wv) {
    switch (wv) {
        case 1: return reinterpret_cast<const void*>(gru_bwd_kernel<1>);
        case 2: return reinterpret_cast<const void*>(gru_bwd_kernel<2>);
        case 4: return reinterpret_cast<const void*>(gru_bwd_kernel<4>);
        default: return reinterpret_cast<const void*>(gru_bwd_kernel<8>);
    }
}

inline int lds_bytes_for(int H_max) { return 3 * (round16(H_max) / UU) * 256 * (int)sizeof(float); }

// How many workgroups of a persistent launch the chip holds AT ONCE: the runtime's occupancy answer for this
// kernel, block size and LDS slice, times the CU count.  Every workgroup of a launch must be resident (the step
// hand-off spins on peers), so this is the bound on the grid -- not the CU count: an H = 300 slice is 57 KB of
// the CU's 160 KB of LDS, two such workgroups share a CU.  0 = the slice does not fit at all.
int resident_workgroups(bool backward, int wv, int lds_bytes) {
    if (lds_bytes > LDS_BUDGET || lds_bytes <= 0) return 0;
    // the answer depends on (kernel, block size, LDS slice) only: asked once (lds is a multiple of 3 KB)
    static int cache[2][4][LDS_BUDGET / 3072 + 1];  // 0 = not asked yet, -1 = asked: none
    const int wi = wv == 1 ? 0 : wv == 2 ? 1 : wv == 4 ? 2 : 3;
    int& c = cache[backward ? 1 : 0][wi][lds_bytes / 3072];
    if (c == 0) {
        const void* k = backward ? bwd_kernel(wv) : fwd_kernel(wv);
        int per_cu = 0;
        // the attribute is per kernel, not per launch: raise it to the budget once, whatever this launch needs
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BUDGET) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * wv, (size_t)lds_bytes) != hipSuccess)
            per_cu = 0;
        if (per_cu > MAX_WG_PER_CU) per_cu = MAX_WG_PER_CU;
        c = per_cu > 0 ? per_cu * cu_count() : -1;
    }
    return c > 0 ? c : 0;
}

// cell height (batch rows per workgroup) of every instance and the launch's waves per workgroup.  An instance
// may ask for its own height (inst.rows in {16, 32, 64, 128}); 0 = the library's choice for the whole launch:
// 64 rows when the grid is resident that way, else 128.  DSSVAE_GRU_WAVES=8 forces 128 (tests).
// -> waves per workgroup (1, 2, 4, 8), 0 = not resident either way, -1 = bad rows value
int pick_rows(bool backward, int n_inst, const int32_t* B, const int32_t* H, const int32_t* want, int* rows_out) {
    int max_h = 0;
    bool any_auto = false;
    for (int i = 0; i < n_inst; ++i) {
        if (H[i] > max_h) max_h = H[i];
        if (want[i] == 0) any_auto = true;
        else if (want[i] != 16 && want[i] != 32 && want[i] != 64 && want[i] != 128) return -1;
    }
    const int lds = lds_bytes_for(max_h);
    const char* f = getenv("DSSVAE_GRU_WAVES");
    const bool force8 = f && atoi(f) == 8;
    const int cand[2] = {force8 ? 128 : 64, 128};
    for (int c = 0; c < (any_auto ? 2 : 1); ++c) {
        int wg = 0, max_rows = 16;
        for (int i = 0; i < n_inst; ++i) {
            rows_out[i] = want[i] ? want[i] : cand[c];
            wg += ((B[i] + rows_out[i] - 1) / rows_out[i]) * (round16(H[i]) / UU);
            if (rows_out[i] > max_rows) max_rows = rows_out[i];
        }
        const int wv = max_rows / 16;
        if (wg <= resident_workgroups(backward, wv, lds)) return wv;
    }
    return 0;
}

}  // namespace

extern "C" int64_t dssvae_gru_workspace_bytes(int n_inst, const int32_t* B_host, const int32_t* H_host) {
    if (n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !B_host || !H_host) return DSSVAE_EINVAL;
    for (int i = 0; i < n_inst; ++i)
        if (B_host[i] <= 0 || H_host[i] <= 0) return DSSVAE_EINVAL;
    return (int64_t)plan(n_inst, B_host, H_host).total_bytes;
}

extern "C" int dssvae_gru_resident_workgroups(int H_max, int rows_max, int backward, int* n_host) {
    if (!n_host || H_max <= 0 || (rows_max != 16 && rows_max != 32 && rows_max != 64 && rows_max != 128))
        return DSSVAE_EINVAL;
    *n_host = resident_workgroups(backward != 0, rows_max / 16, lds_bytes_for(H_max));
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_fwd(const dssvae_gru_fwd_inst* inst, int n_inst, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (!inst || n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !workspace) return DSSVAE_EINVAL;
    int32_t Bs[DSSVAE_GRU_MAX_INST], Hs[DSSVAE_GRU_MAX_INST], want[DSSVAE_GRU_MAX_INST];
    for (int i = 0; i < n_inst; ++i) {
        const dssvae_gru_fwd_inst& x = inst[i];
        if (!x.xproj || !x.whh || !x.bhh || !x.out || x.B <= 0 || x.T <= 0 || x.H <= 0 || x.ldx < 3 * x.H ||
            x.ldo < x.H || x.t0 < 0 || (x.T_total > 0 && x.t0 + x.T > x.T_total) || (x.T_total <= 0 && x.t0 != 0))
            return DSSVAE_EINVAL;
        Bs[i] = x.B; Hs[i] = x.H; want[i] = x.rows;
    }
    const Layout L = plan(n_inst, Bs, Hs);
    if ((int64_t)L.total_bytes > workspace_bytes || !aligned16(workspace)) return DSSVAE_EINVAL;
    FwdArgs args;
    args.n_inst = n_inst;
    args.err = reinterpret_cast<unsigned*>(workspace);
    int rows[DSSVAE_GRU_MAX_INST];
    const int wv = pick_rows(false, n_inst, Bs, Hs, want, rows);
    if (wv < 0) return DSSVAE_EINVAL;
    if (wv == 0) return DSSVAE_ETOOBIG;
    int wg = 0, max_h = 0;
    for (int i = 0; i < n_inst; ++i) {
        FwdDev& d = args.inst[i];
        d.in = inst[i];
        d.rows = rows[i];
        d.n_slices = round16(inst[i].H) / UU;
        d.n_bblocks = (inst[i].B + rows[i] - 1) / rows[i];
        d.wg_begin = wg;
        wg += d.n_slices * d.n_bblocks;
        d.counters = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + L.cnt_off[i]);
        d.hx = reinterpret_cast<float*>(static_cast<char*>(workspace) + L.buf_off[i]);
        if (inst[i].H > max_h) max_h = inst[i].H;
    }
    const int lds_bytes = lds_bytes_for(max_h);
    hipStream_t s = (hipStream_t)stream;
    const int e = dssvae_zero2d(reinterpret_cast<float*>(workspace), (int64_t)(L.header_bytes / 4), 1,
                                (int)(L.header_bytes / 4), s);
    if (e != DSSVAE_OK) return e;
    void* kargs[] = {&args};
    const hipError_t le = hipLaunchKernel(fwd_kernel(wv), dim3(wg), dim3(64 * wv), kargs, lds_bytes, s);
    if (le != hipSuccess) return (int)le;
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_bwd(const dssvae_gru_bwd_inst* inst, int n_inst, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (!inst || n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !workspace) return DSSVAE_EINVAL;
    int32_t Bs[DSSVAE_GRU_MAX_INST], Hs[DSSVAE_GRU_MAX_INST], want[DSSVAE_GRU_MAX_INST];
    for (int i = 0; i < n_inst; ++i) {
        const dssvae_gru_bwd_inst& x = inst[i];
        if (!x.whh || !x.gates || !x.dgx || !x.dgh || x.B <= 0 || x.T <= 0 || x.H <= 0 || (x.dout && x.ldd < x.H) ||
            x.t0 < 0 || (x.T_total > 0 && x.t0 + x.T > x.T_total) || (x.T_total <= 0 && x.t0 != 0))
            return DSSVAE_EINVAL;
        Bs[i] = x.B; Hs[i] = x.H; want[i] = x.rows;
    }
    const Layout L = plan(n_inst, Bs, Hs);
    if ((int64_t)L.total_bytes > workspace_bytes || !aligned16(workspace)) return DSSVAE_EINVAL;
    BwdArgs args;
    args.n_inst = n_inst;
    args.err = reinterpret_cast<unsigned*>(workspace);
    int rows[DSSVAE_GRU_MAX_INST];
    const int wv = pick_rows(true, n_inst, Bs, Hs, want, rows);
    if (wv < 0) return DSSVAE_EINVAL;
    if (wv == 0) return DSSVAE_ETOOBIG;
    int wg = 0, max_h = 0;
    for (int i = 0; i < n_inst; ++i) {
        BwdDev& d = args.inst[i];
        d.in = inst[i];
        d.rows = rows[i];
        d.n_slices = round16(inst[i].H) / UU;
        d.n_bblocks = (inst[i].B + rows[i] - 1) / rows[i];
        d.wg_begin = wg;
        wg += d.n_slices * d.n_bblocks;
        d.counters = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + L.cnt_off[i]);
        d.gx = reinterpret_cast<float*>(static_cast<char*>(workspace) + L.buf_off[i]);
        if (inst[i].H > max_h) max_h = inst[i].H;
    }
    const int lds_bytes = lds_bytes_for(max_h);
    hipStream_t s = (hipStream_t)stream;
    const int e = dssvae_zero2d(reinterpret_cast<float*>(workspace), (int64_t)(L.header_bytes / 4), 1,
                                (int)(L.header_bytes / 4), s);
    if (e != DSSVAE_OK) return e;
    void* kargs[] = {&args};
    const hipError_t le = hipLaunchKernel(bwd_kernel(wv), dim3(wg), dim3(64 * wv), kargs, lds_bytes, s);
    if (le != hipSuccess) return (int)le;
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_check(void* workspace, void* stream) {
    if (!workspace) return DSSVAE_EINVAL;
    unsigned err = 0;
    hipError_t e = hipMemcpyAsync(&err, workspace, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return err ? DSSVAE_ETIMEOUT : DSSVAE_OK;
}
