// Persistent GRU recurrence for gfx950 (forward and backward through time).
//
// Decomposition: an "instance" is one independent recurrence (a direction of an encoder, a decoder
// layer, a task head).  Each instance is cut into cells of (batch block of 16 .. 128 rows) x (slice of 16 hidden
// units); one workgroup owns one cell for ALL timesteps of the launch:
//   * its W_hh slice (3 gates x 16 units x H, fp32) is loaded into LDS once and stays resident;
//   * a cell is 1 .. 8 row groups of 16 batch rows; per step a row group multiplies its 16 rows of h_{t-1}
//     (B operand) with the resident slice (A operand) on v_mfma_f32_16x16x4_f32, applies the fused gate math in
//     registers and stores its 16x16 piece of h_t.  In the 16-wave layouts a row group is KS = 2 or 4 waves on
//     different SIMDs that split the K range of the product and meet through LDS (see lds_signal below); in the
//     4- / 8-wave layouts it is one wave;
//   * the state is exchanged through a ping-pong buffer laid out [slice][row][16 units], so a row group
//     publishes ONE contiguous 1 KB block per step and reads 1 KB contiguous per slice; the row groups that own
//     the same 16 batch rows in the workgroups of one (instance, batch block) meet on one monotonic arrival
//     counter per step -- there is no grid-wide barrier and no workgroup barrier in the step loop, and
//     independent instances / batch blocks / row groups never wait for each other.
// A launch may cover steps [t0, t0 + T) of a longer recurrence (inst.t0 / T_total): the caller cuts a launch in
// time where its short recurrences end, so that the long ones continue in finer cells (ops._segments).
// Measured where a step goes (tools/gru_phases.py, -DGRU_PROFILE build; 2 x B256 H300, 64-row cells, KS = 4;
// profiles/r02_z_gru_phases.txt).  Forward 7.7 us = counter match 2.3 | exchange loads + own products 1.85 | partial
// sums of the other parts 1.7 | gate math 0.04 | state store drained 1.0 | signal 0.55 | history stores 0.3.
// Backward 9.8 us = saved values + gate gradients 2.2 | exchange stores + publish 0.9 | history stores 0.2 | counter
// match 1.9 | exchange loads + own products 3.0 | partial sums 1.4.  Four dependent memory round trips per step; the
// arithmetic is a quarter of it.  Two rules paid for themselves in these kernels: (1) loads and stores the step waits
// on are issued back to back without branches -- unconditional loads at clamped addresses masked afterwards, all
// stores under one predicate -- because a load under a branch is waited for at the join (the backward's seven
// guarded loads cost a round trip each: 4.4 us of every step, 1.5 now); (2) a 128-bit buffer store takes its scalar
// offset in the VGPR (see st_sc1).
// Cross-workgroup visibility (MI355X_MICROARCH.md, "Valid forms", first table row): EVERY byte of
// the exchange buffer is stored write-through (buffer_store_dwordx4 sc1) and EVERY load of it is a
// buffer_load_dwordx4 sc1 (bypasses the CU's L1), so no release / acquire cache maintenance is
// needed: a storing wave drains its stores (s_waitcnt vmcnt(0)), then its lane 0 adds to the row
// group's counter (agent-scope atomic); a consuming wave's lane 0 polls that counter with sc1 loads,
// then the wave issues its sc1 loads (every byte it reads was stored by a wave that signalled after
// its own drain).
// The exchange loads go through a buffer resource that ends with the batch block's half of the ping-pong, with the
// slice offset in the VGPR offset.  Register-chunk slots past a wave's part are clamped to the part's last slice
// (loaded again, never multiplied) rather than left to the range check: every load of the stream then returns through
// the same path, so the in-order return that s_waitcnt vmcnt(n > 0) counts on does not hang on how the hardware
// retires an out-of-range load (a precaution -- no failure was traced to it).
// All workgroups of a launch must be co-resident: the host admits a grid only up to the runtime's occupancy for
// the kernel at the launch's LDS slice and block size x the CU count (dssvae_gru_resident_workgroups: 1 to 2
// workgroups per CU at these slices), else DSSVAE_ETOOBIG; every spin is bounded and reports through an error
// word instead of hanging.  The bound is for a chip the launch has to itself or shares with kernels whose
// workgroups retire (GEMMs, collectives): two persistent launches must never overlap each other (ops._PersistentTurn
// serialises them across streams), and host threads launching them concurrently must do the same.
//
// Roofline: per step 2*B*H*3H flop on fp32 MFMA; in practice latency-bound by the step hand-off.
#include "common.h"
typedef float f32x2 __attribute__((ext_vector_type(2)));
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int UU = 16;   // hidden units per workgroup (M of the 16x16x4 MFMA)
constexpr int CH = 8;    // exchange slices fetched per register chunk (forward)
constexpr int CHB = 6;   // ... backward (measured: 8 -> 6 takes 6 % off the backward launches, forward loses 3 %)
constexpr unsigned SPIN_LIMIT = 1u << 24;
constexpr int CNT_STRIDE = 32;  // one 128-B line per counter
constexpr int PART_BYTES = 64 * 16;      // LDS per non-lead wave and partial sum (forward: 3 gates; backward: 1)
constexpr int SYNC_WORDS = 64;           // LDS words behind them: [0,16) release per row group, [16,32) saved values
                                         // of the next step staged (backward), [32,64) done per wave
constexpr int SV_BYTES = 64 * 16;        // backward: one staged saved-value vector of a row group (16 B per lane)
constexpr int LDS_MAX = 160 * 1024;
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
    unsigned* err;     // error word of this launch (first word of its workspace)
    unsigned* sticky;  // the device's registered error word (dssvae_gru_error_word), or NULL
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
    int sv_stage;  // the launch's LDS has room for the staged saved values of every instance (see gru_bwd_kernel)
    unsigned* err;
    unsigned* sticky;
};
struct ErrWords {
    unsigned* err;
    unsigned* sticky;
};
__device__ __forceinline__ void raise_timeout(ErrWords e) {
    __hip_atomic_store(e.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e.sticky) __hip_atomic_store(e.sticky, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ld_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16));
}
// The store's scalar offset goes into the VGPR offset, soffset stays the literal 0: with an SGPR there the compiler
// takes a 128-bit buffer store to be free of the "store data overwritten right after issue" hazard and schedules no
// wait states -- on gfx950 the data registers of four back-to-back stores rewritten in between (what the register
// allocator chose for the LSTM backward) reached memory partly overwritten: a run-to-run varying error of the
// exchanged gate gradients.  With a literal soffset it inserts the wait states itself.
__device__ __forceinline__ void st_sc1(f32x4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4v, v), r, voff + soff, 0, 16);
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
__device__ __forceinline__ void await(unsigned* cnt, unsigned target, ErrWords err) {
    if ((threadIdx.x & 63) == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(0);
            if ((++spins & 1023u) == 0) {
                if (spins > SPIN_LIMIT ||
                    __hip_atomic_load(err.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    raise_timeout(err);
                    break;
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// K split INSIDE a workgroup: the KS waves of a row group (on KS different SIMDs) each take 1/KS of the exchange
// slices -- a KS-th of the loads in flight per wave and of the dependent MFMA chain -- and meet through LDS: the
// lead wave (part 0) polls the row group's counter and releases the others through an LDS word (the relayed form
// of the hand-off table's first row: "the other waves load ... after an LDS word it then sets"); the others leave
// their partial sums in LDS and raise a word each; the lead adds them up, applies the gate math and publishes.
#ifndef LDS_POLL_SLEEP
#define LDS_POLL_SLEEP 1
#endif
typedef __attribute__((address_space(3))) volatile unsigned lds_word;  // LDS address space: ds_ instructions, not flat
__device__ __forceinline__ void lds_signal(lds_word* w, unsigned v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // what this wave wrote to LDS before is there
    *w = v;
}
__device__ __forceinline__ void lds_await(lds_word* w, unsigned target, ErrWords err) {
    unsigned spins = 0;
    while (*w < target) {
        __builtin_amdgcn_s_sleep(LDS_POLL_SLEEP);
        if ((++spins & 4095u) == 0 &&
            (spins > SPIN_LIMIT || __hip_atomic_load(err.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
            raise_timeout(err);
            break;
        }
    }
    asm volatile("" ::: "memory");
}

// The gate math sits on the step's critical path in ONE wave (4 values per lane, 12 transcendentals): hardware
// exp2 / rcp (1 ulp each) instead of the library's division and tanhf -- ~0.6 us of VALU time per step less.
// |error| <= ~2e-7 absolute, inside the fp32 noise of the 300..600-term dot products that feed the gates.
__device__ __forceinline__ float sigm(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// Development build only (-DGRU_PROFILE, tools/gru_phases.py): 100 MHz timestamps of the step's phases, taken by the
// lead wave of row group 0 in a few sampled workgroups.  Not compiled into the shipped library.
#ifdef GRU_PROFILE
constexpr int PROF_WG = 4, PROF_STEPS = 128, PROF_PH = 8;
__device__ unsigned long long g_prof[PROF_WG][PROF_STEPS][PROF_PH];
__device__ unsigned long long g_prof_part[PROF_WG][PROF_STEPS][4];  // wave 1 (a non-lead part): top, released, products, signalled
__device__ unsigned g_simd[PROF_WG][16];  // SIMD id (HW_REG_HW_ID[5:4]) of every wave of the sampled workgroups
#define PROF_SLOT(nwg) ((int)blockIdx.x == 0 ? 0 : (int)blockIdx.x == (nwg) / 3 ? 1 : (int)blockIdx.x == 2 * (nwg) / 3 ? 2 \
                        : (int)blockIdx.x == (nwg) - 1 ? 3 : -1)
#define PROFP(ph)                                                                                              \
    do {                                                                                                       \
        if (wave >= 1 && wave <= 3 && blockIdx.x == 0 && step < PROF_STEPS && lane == 0)                      \
            g_prof_part[wave][step][ph] = wall_clock64();                                                     \
    } while (0)
#define PROF(ph)                                                                                     \
    do {                                                                                             \
        if (prof_slot >= 0 && step < PROF_STEPS && lane == 0) g_prof[prof_slot][step][ph] = wall_clock64(); \
    } while (0)
#else
#define PROF(ph) do {} while (0)
#define PROFP(ph) do {} while (0)
#endif

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

// acc[g] += W_g[16 units][16 k of slice] . X[16 k][16 rows] over N consecutive slices; Wl: the LDS image, per slice
// [4 c][4 kq][16 j]; xv[i]: the lane's 4 k-values (k = 4*kq + c) of its batch row in slice i.  Branch-free, the W
// fragments read from LDS ahead of the MFMAs that use them: the LDS latency hides under the matrix pipe instead of
// stalling it once per slice.
template <int NG, int N, int M>
__device__ __forceinline__ void mfma_slices(f32x4 (&acc)[NG], const float* Wl, int gate_stride, const f32x4 (&xv)[M]) {
    // k-steps of N slices in one stream; the W fragments of step s + 2 are read from LDS while step s runs (a ring of
    // three steps: 3 * NG registers, where whole-slice double buffering spilled the 16-wave layouts)
    constexpr int S = N * 4;
    float w[3][NG];
#pragma unroll
    for (int s = 0; s < 2 && s < S; ++s)
#pragma unroll
        for (int g = 0; g < NG; ++g) w[s][g] = Wl[(s / 4) * 256 + g * gate_stride + (s % 4) * 64];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + 2 < S) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
                w[(s + 2) % 3][g] = Wl[((s + 2) / 4) * 256 + g * gate_stride + ((s + 2) % 4) * 64];
        }
        // the order is pinned step by step (the scheduler otherwise chains the 4 MFMAs of an accumulator back to
        // back, each waiting for the one before, and pairs every LDS read with a wait): NG independent MFMAs per k
#pragma unroll
        for (int g = 0; g < NG; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s % 3][g], xv[s / 4][s % 4], acc[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// the first n (1 .. M, wave-uniform) slices of xv: the branch-free stream of exactly that length
template <int NG, int M, int N = M>
__device__ __forceinline__ void mfma_tail(f32x4 (&acc)[NG], const float* Wl, int gate_stride, const f32x4 (&xv)[M], int n) {
    if constexpr (N >= 1) {
        if (n == N) mfma_slices<NG, N, M>(acc, Wl, gate_stride, xv);
        else mfma_tail<NG, M, N - 1>(acc, Wl, gate_stride, xv, n);
    }
}

// backward: N blocks (16 gate columns each) into three accumulators in rotation; the W fragments of block i + 1 are read
// from LDS before the MFMAs of block i, blocks stay in order (consecutive blocks use different accumulators)
template <int N, int M>
__device__ __forceinline__ void mfma_blocks(f32x4 (&acc)[3], const float* wb, const f32x4 (&xv)[M]) {
    // one stream of 4 N k-steps; the W fragment of step s + 3 is read while step s runs (a ring of four registers)
    constexpr int S = N * 4;
    float w[4];
#pragma unroll
    for (int s = 0; s < 3 && s < S; ++s) w[s] = wb[(s / 4) * 256 + (s % 4) * 64];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        if (s + 3 < S) w[(s + 3) % 4] = wb[((s + 3) / 4) * 256 + ((s + 3) % 4) * 64];
        f32x4& a = acc[(s / 4) % 3];
        a = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s % 4], xv[s / 4][s % 4], a, 0, 0, 0);
        if ((s % 4) == 3) __builtin_amdgcn_sched_barrier(0);
    }
}
template <int M, int N = M>
__device__ __forceinline__ void mfma_blocks_tail(f32x4 (&acc)[3], const float* wb, const f32x4 (&xv)[M], int n) {
    if constexpr (N >= 1) {
        if (n == N) mfma_blocks<N, M>(acc, wb, xv);
        else mfma_blocks_tail<M, N - 1>(acc, wb, xv, n);
    }
}

// ------------------------------------------------------------------------------------------
// WV waves per workgroup: 4 (64 batch rows) or 8 (128 rows: half the workgroups per recurrence, two waves per
// SIMD share the resident slice -- what lets every decoder of a layer share ONE co-resident launch)
// LSTM = true: the same kernel for an LSTM cell (gate order i, f, g, o; 4 gate rows per unit in the slice, the cell
// state beside the hidden state in registers; saved row [i | f | g | o | c_{t-1} | h_{t-1} | tanh c_t], 7H wide)
template <int WV, bool LSTM>
__global__ __launch_bounds__(64 * WV) void gru_fwd_kernel(FwdArgs args) {
    constexpr int NTHR = 64 * WV;
    constexpr int NG = LSTM ? 4 : 3;
    constexpr int CHF = WV > 8 ? (LSTM ? 3 : 5) : CH;
    // Ws[g][slice s][c][kq][j] = whh[(g*H + u0 + j)*H + 16 s + 4 kq + c]
    extern __shared__ __attribute__((aligned(16))) float Ws[];
    int ii = 0;
    while (ii + 1 < args.n_inst && (int)blockIdx.x >= args.inst[ii + 1].wg_begin) ++ii;
    const FwdDev& d = args.inst[ii];
    const dssvae_gru_fwd_inst& in = d.in;
    const int local = blockIdx.x - d.wg_begin;
    const int bblock = local / d.n_slices, slice = local % d.n_slices;
    const int H = in.H, T = in.T, B = in.B, NS = d.n_slices;
    // cells of this instance hold BB batch rows = WVI row groups of 16; the WV waves are KS per row group
    const int BB = d.rows, WVI = BB >> 4, KS = WV / WVI, BLK = BB * UU;
    const int Tt = in.T_total > 0 ? in.T_total : T;   // row stride of the [B, T_total, .] histories
    const int u0 = slice * UU;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // row group and part of the K range; the lead (part 0) of row group r is its wave r % KS, so that the leads of a
    // workgroup -- the waves that also poll, reduce, apply the gates and publish -- sit on different SIMDs
    const int rg = wave / KS, kp = (wave - rg * KS + KS - rg % KS) % KS;
    const bool lead = kp == 0;
    const int j = lane & 15, kq = lane >> 4;
    const int b = bblock * BB + rg * 16 + j;
    const bool row_ok = b < B;
    const int ub = u0 + 4 * kq;                       // first of this lane's 4 units
    const int n_ok = max(0, min(4, H - ub));          // how many of them exist
    unsigned* cnt = d.counters + ((size_t)bblock * WVI + rg) * CNT_STRIDE;  // this wave's row group
    const bool vec_h = (H % 4) == 0;
    const int gate_stride = NS * 256;
    // behind the slice, only there (and touched) when KS > 1: a slot of partial sums per non-lead wave, sync words
    f32x4* parts = reinterpret_cast<f32x4*>(Ws + NG * NS * 256);
    lds_word* sync = (lds_word*)(parts + (WV - WVI) * NG * 64);
    const int slot0 = rg * (KS - 1);  // first slot of this row group's non-lead waves

    for (int idx = threadIdx.x; idx < NG * NS * 256; idx += NTHR) {
        const int g = idx / (NS * 256), rem = idx % (NS * 256);
        const int jj = rem / (NS * 16), k = rem % (NS * 16);  // k fastest: coalesced global reads
        const int u = u0 + jj;
        const float w = (u < H && k < H) ? in.whh[(size_t)(g * H + u) * H + k] : 0.f;
        const int s = k >> 4, kk = k & 15;
        Ws[g * gate_stride + s * 256 + (kk & 3) * 64 + (kk >> 2) * 16 + jj] = w;
    }
    if (KS > 1 && threadIdx.x < SYNC_WORDS) sync[threadIdx.x] = 0u;
    float bh[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) bh[g][c] = (c < n_ok) ? in.bhh[g * H + ub + c] : 0.f;
    const int len = (in.lengths && row_ok) ? in.lengths[b] : Tt;

    // exchange buffers of this batch block: [2][NS][BB][16]; this lane's piece of its own block
    const unsigned half_bytes = (unsigned)NS * BLK * 4u;
    float* hx0 = d.hx + (size_t)bblock * NS * BLK;
    float* hx1 = hx0 + (size_t)d.n_bblocks * NS * BLK;
    const __amdgpu_buffer_rsrc_t ws0 = make_rsrc(hx0, half_bytes), ws1 = make_rsrc(hx1, half_bytes);
    // this wave's part of the slices: a resource of its own (an empty part fetches nothing)
    // (an uneven split -- the lead, released first, taking 1 .. 3 slices more than the parts -- was measured: +2 .. +6 %)
    const int per = (NS + KS - 1) / KS, s_lo = min(NS, kp * per), s_n = min(NS - s_lo, per), s_last = max(s_n - 1, 0);
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(hx0 + (size_t)s_lo * BLK, (unsigned)s_n * BLK * 4u),
                                 rs1 = make_rsrc(hx1 + (size_t)s_lo * BLK, (unsigned)s_n * BLK * 4u);
    const int voff = ((rg * 16 + j) * UU + 4 * kq) * 4;  // same lane offset inside every slice block
    const int own = slice * BLK * 4;
    lds_word* go = sync + rg;
    lds_word* done = sync + 32 + wave;

    __syncthreads();  // the slice is in LDS for every wave; from here on the waves run free
    // state 0 into the exchange buffer (pad units and pad rows are zero and stay zero)
    float hcur[4] = {0.f, 0.f, 0.f, 0.f}, ccur[4] = {0.f, 0.f, 0.f, 0.f};
    if (lead) {
        if (row_ok && in.h0) load4(hcur, in.h0 + (size_t)b * H + ub, vec_h, n_ok);
        if (LSTM && row_ok && in.c0) load4(ccur, in.c0 + (size_t)b * H + ub, vec_h, n_ok);
        st_sc1((f32x4){hcur[0], hcur[1], hcur[2], hcur[3]}, ws0, voff, own);
        publish(cnt);
    }

    const float* Wl = Ws + s_lo * 256 + kq * 16 + j;
#ifdef GRU_PROFILE
    const int prof_slot = (wave == 0) ? PROF_SLOT((int)gridDim.x) : -1;
    if (PROF_SLOT((int)gridDim.x) >= 0 && lane == 0 && wave < 16)
        g_simd[PROF_SLOT((int)gridDim.x)][wave] = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
#endif
    const float* const xproj_p = in.xproj;
    const int64_t ldx = in.ldx;
    const bool fast4 = (H % 4) == 0 && H >= 4 && (ldx % 4) == 0 && (reinterpret_cast<uintptr_t>(xproj_p) % 16) == 0;
    const bool fast2 = !fast4 && (H % 2) == 0 && H >= 2 && (ldx % 2) == 0 && (reinterpret_cast<uintptr_t>(xproj_p) % 8) == 0;
    float* const out_p = in.out;
    float* const hist_p = in.gates;
    const int64_t ldo = in.ldo;
    const uintptr_t st_bits = reinterpret_cast<uintptr_t>(out_p) | reinterpret_cast<uintptr_t>(hist_p);
    const bool fast4s = (H % 4) == 0 && (ldo % 4) == 0 && (st_bits % 16) == 0;
    const bool fast2s = !fast4s && (H % 2) == 0 && (ldo % 2) == 0 && (st_bits % 8) == 0;
    for (int step = 0; step < T; ++step) {
        PROF(0);
        const int gs = in.t0 + step;  // step of the whole recurrence (this launch runs steps t0 .. t0 + T - 1)
        const int t = in.reverse ? Tt - 1 - gs : gs;
        const __amdgpu_buffer_rsrc_t rin = (step & 1) ? rs1 : rs0, rout = (step & 1) ? ws0 : ws1;
        // the x-projection gates of this step do not depend on the hand-off: fetch them first
        const size_t bt = (size_t)(row_ok ? b : 0) * Tt + t;
        float xg[NG][4];
        if (lead) {
            const float* xp = xproj_p + bt * ldx;
            if (fast4) {
                // all NG loads back to back, clamped into the row (a lane's pad units are masked by the gate math's
                // selects): one round trip, where the guarded loads below are waited for one after the other
                const int uc = min(ub, H - 4);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(xp + g * H + uc);
#pragma unroll
                    for (int c = 0; c < 4; ++c) xg[g][c] = v[c];
                }
            } else if (fast2) {
                const int ua = min(ub, H - 2), uh = min(ub + 2, H - 2);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const f32x2 lo = *reinterpret_cast<const f32x2*>(xp + g * H + ua),
                                hi = *reinterpret_cast<const f32x2*>(xp + g * H + uh);
                    xg[g][0] = lo[0]; xg[g][1] = lo[1]; xg[g][2] = hi[0]; xg[g][3] = hi[1];
                }
            } else {
                const bool vec_x = vec_h && (ldx % 4) == 0;
#pragma unroll
                for (int g = 0; g < NG; ++g) load4(xg[g], xp + g * H + ub, vec_x, n_ok);
            }
            await(cnt, (unsigned)(step + 1) * NS, (ErrWords){args.err, args.sticky});
            PROF(1);
            if (KS > 1) lds_signal(go, (unsigned)step + 1u);
        } else {
            PROFP(0);
            // (measured: every part polling the row group's counter itself instead of this relay -- four times the
            // polls on the counter's line -- makes the forward launches 29 - 34 % slower: 0.955 -> 1.231 ms)
            lds_await(go, (unsigned)step + 1u, (ErrWords){args.err, args.sticky});
            PROFP(1);
        }

        f32x4 acc[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 cur[CHF], nxt[CHF];
#pragma unroll
        for (int i = 0; i < CHF; ++i) cur[i] = ld_sc1(rin, voff + min(i, s_last) * BLK * 4, 0);
        // Full chunks with another chunk behind them: prefetch UNCONDITIONALLY (a load inside a branch hides the
        // number of outstanding loads from the compiler, which then waits for all of them -- the prefetch included
        // -- before the chunk's first MFMA) and multiply without per-slice branches (the LDS reads of the slice
        // move ahead of the MFMAs that hide them).  Slices past the part are clamped to its last slice (loaded
        // again, never multiplied): see the note on the exchange loads at the top of the file.
        int s0 = 0;
        for (; s0 + CHF < s_n; s0 += CHF) {
#pragma unroll
            for (int i = 0; i < CHF; ++i) nxt[i] = ld_sc1(rin, voff + min(s0 + CHF + i, s_last) * BLK * 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_slices<NG, CHF, CHF>(acc, Wl + s0 * 256, gate_stride, cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CHF; ++i) cur[i] = nxt[i];
        }
        mfma_tail<NG, CHF>(acc, Wl + s0 * 256, gate_stride, cur, s_n - s0);  // the last chunk: 1 .. CHF slices (0: none)
        if (!lead) {
#ifdef GRU_PROFILE
            asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]));
#endif
            PROFP(2);
#pragma unroll
            for (int g = 0; g < NG; ++g) parts[((slot0 + kp - 1) * NG + g) * 64 + lane] = acc[g];
            lds_signal(done, (unsigned)step + 1u);
            PROFP(3);
            continue;
        }
#ifdef GRU_PROFILE
        asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]));  // the products are there
#endif
        PROF(2);
        for (int p = 1; p < KS; ++p) {
            lds_await(sync + 32 + rg * KS + (p + rg % KS) % KS, (unsigned)step + 1u, (ErrWords){args.err, args.sticky});
#ifdef GRU_PROFILE
            if (p == 1) PROF(7);
#endif
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] += parts[((slot0 + p - 1) * NG + g) * 64 + lane];
        }
        PROF(3);
        // fused gate math: lane holds units ub..ub+3 of batch row b (pad units keep writing zeros)
        const bool live = row_ok && t < len;
        // GRU: r, z, n, W_hn h + b_hn.  LSTM: i, f, g, o; v4 = tanh c_t
        float v0[4], v1[4], v2[4], v3[4], v4[4], hnew[4], cnew[4], outv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float hv;
            if constexpr (LSTM) {
                v0[c] = sigm(xg[0][c] + acc[0][c] + bh[0][c]);
                v1[c] = sigm(xg[1][c] + acc[1][c] + bh[1][c]);
                v2[c] = tanh_fast(xg[2][c] + acc[2][c] + bh[2][c]);
                v3[c] = sigm(xg[3][c] + acc[3][c] + bh[3][c]);
                const float cv = v1[c] * ccur[c] + v0[c] * v2[c];
                v4[c] = tanh_fast(cv);
                hv = v3[c] * v4[c];
                cnew[c] = (c < n_ok && row_ok) ? (live ? cv : ccur[c]) : 0.f;
            } else {
                v0[c] = sigm(xg[0][c] + acc[0][c] + bh[0][c]);
                v1[c] = sigm(xg[1][c] + acc[1][c] + bh[1][c]);
                v3[c] = acc[2][c] + bh[2][c];
                v2[c] = tanh_fast(xg[2][c] + v0[c] * v3[c]);
                hv = (1.f - v1[c]) * v2[c] + v1[c] * hcur[c];
                v4[c] = 0.f;
                cnew[c] = 0.f;
            }
            hnew[c] = (c < n_ok && row_ok) ? (live ? hv : hcur[c]) : 0.f;
            outv[c] = live ? hv : 0.f;
        }
        // hand the new state over first; the history stores below then overlap the peers' next step
        PROF(4);
        if (step + 1 < T) {
            st_sc1((f32x4){hnew[0], hnew[1], hnew[2], hnew[3]}, rout, voff, own);
#ifdef GRU_PROFILE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PROF(5);
#endif
            publish(cnt);
        }
        PROF(6);
        // history: every store of the step under ONE predicate per vector width (a lane's 4 units are all inside H
        // or all pad when H % 4 == 0; two halves when H % 2 == 0), not a branch diamond per store
        auto store_all = [&](auto st) {
            st(out_p + bt * ldo + ub, outv);
            if (hist_p) {
                float* gs = hist_p + bt * (LSTM ? 7 : 5) * H + ub;
                st(gs, v0); st(gs + H, v1); st(gs + 2 * H, v2); st(gs + 3 * H, v3);
                if constexpr (LSTM) {
                    st(gs + 4 * H, ccur); st(gs + 5 * H, hcur); st(gs + 6 * H, v4);
                } else {
                    st(gs + 4 * H, hcur);
                }
            }
        };
        if (fast4s) {
            if (row_ok && ub + 4 <= H)
                store_all([](float* q, const float (&v)[4]) { *reinterpret_cast<f32x4*>(q) = (f32x4){v[0], v[1], v[2], v[3]}; });
        } else if (fast2s) {
            if (row_ok && ub + 2 <= H)
                store_all([](float* q, const float (&v)[4]) { *reinterpret_cast<f32x2*>(q) = (f32x2){v[0], v[1]}; });
            if (row_ok && ub + 4 <= H)
                store_all([](float* q, const float (&v)[4]) { *reinterpret_cast<f32x2*>(q + 2) = (f32x2){v[2], v[3]}; });
        } else if (row_ok) {
            store_all([&](float* q, const float (&v)[4]) { store4(q, v, false, n_ok); });
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) { hcur[c] = hnew[c]; ccur[c] = cnew[c]; }
    }
    if (lead && row_ok && n_ok > 0 && in.h_last) store4(in.h_last + (size_t)b * H + ub, hcur, vec_h, n_ok);
    if (LSTM && lead && row_ok && n_ok > 0 && in.c_last) store4(in.c_last + (size_t)b * H + ub, ccur, vec_h, n_ok);
}

// ------------------------------------------------------------------------------------------
// WV waves per workgroup: 4 (64 batch rows) or 8 (128 rows: half the workgroups per recurrence, two waves per
// SIMD share the resident slice -- what lets every decoder of a layer share ONE co-resident launch)
// LSTM = true: four gate-gradient blocks per slice (d i, d f, d g, d o: the same for the input and the recurrent
// product), d c carried in registers beside d h
template <int WV, bool LSTM>
__global__ __launch_bounds__(64 * WV) void gru_bwd_kernel(BwdArgs args) {
    constexpr int NTHR = 64 * WV;
    constexpr int CB = WV > 8 ? (LSTM ? 4 : 5) : CHB;
    constexpr int NG = LSTM ? 4 : 3;
    // chunk q = NG*p + g covers W_hh rows g*H + 16p .. +15; Ws[q][c][kq][i] = whh[g*H + 16p + 4kq + c][u0 + i]
    extern __shared__ __attribute__((aligned(16))) float Ws[];
    int ii = 0;
    while (ii + 1 < args.n_inst && (int)blockIdx.x >= args.inst[ii + 1].wg_begin) ++ii;
    const BwdDev& d = args.inst[ii];
    const dssvae_gru_bwd_inst& in = d.in;
    const int local = blockIdx.x - d.wg_begin;
    const int bblock = local / d.n_slices, slice = local % d.n_slices;
    const int H = in.H, T = in.T, B = in.B, K = NG * H, NS = d.n_slices, NQ = NG * NS;
    const int BB = d.rows, WVI = BB >> 4, KS = WV / WVI, BLK = BB * UU;
    const int Tt = in.T_total > 0 ? in.T_total : T;
    const int u0 = slice * UU;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave / KS, kp = (wave - rg * KS + KS - rg % KS) % KS;  // row group, part of the K range (see gru_fwd_kernel)
    const bool lead = kp == 0;
    const int j = lane & 15, kq = lane >> 4;
    const int b = bblock * BB + rg * 16 + j;
    const bool row_ok = b < B;
    const int ub = u0 + 4 * kq;
    const int n_ok = max(0, min(4, H - ub));
    unsigned* cnt = d.counters + ((size_t)bblock * WVI + rg) * CNT_STRIDE;  // this wave's row group
    const bool vec_h = (H % 4) == 0;
    // behind the slice, only there (and touched) when KS > 1: a partial sum per non-lead wave, sync words
    f32x4* parts = reinterpret_cast<f32x4*>(Ws + NQ * 256);
    lds_word* sync = (lds_word*)(parts + (WV - WVI) * 64);
    const int slot0 = rg * (KS - 1);
    // behind the sync words: the NV saved-value vectors of the NEXT step per row group, staged by the row group's
    // part 1 while the lead is busy with this step (see the step loop)
    constexpr int NV = LSTM ? 7 : 6;
    f32x4* svbuf = parts + (WV - WVI) * 64 + SYNC_WORDS / 4 + rg * NV * 64 + lane;
    lds_word* sv_ready = sync + 16 + rg;

    for (int idx = threadIdx.x; idx < NQ * 256; idx += NTHR) {
        const int q = idx >> 8, rem = idx & 255;
        const int kk = rem >> 4, i = rem & 15;  // i fastest: 16 consecutive floats of a W_hh row
        const int p = q / NG, g = q - NG * p;
        const int cu = 16 * p + kk, u = u0 + i;
        const float w = (cu < H && u < H) ? in.whh[(size_t)(g * H + cu) * H + u] : 0.f;
        Ws[q * 256 + (kk & 3) * 64 + (kk >> 2) * 16 + i] = w;
    }
    if (KS > 1 && threadIdx.x < SYNC_WORDS) sync[threadIdx.x] = 0u;
    const int len = (in.lengths && row_ok) ? in.lengths[b] : Tt;
    float dh[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    if (lead && row_ok && in.dh_last) load4(dh, in.dh_last + (size_t)b * H + ub, vec_h, n_ok);
    if (LSTM && lead && row_ok && in.dc_last) load4(dc, in.dc_last + (size_t)b * H + ub, vec_h, n_ok);

    const unsigned half_bytes = (unsigned)NQ * BLK * 4u;
    float* gx0 = d.gx + (size_t)bblock * NQ * BLK;
    float* gx1 = gx0 + (size_t)d.n_bblocks * NQ * BLK;
    const __amdgpu_buffer_rsrc_t ws0 = make_rsrc(gx0, half_bytes), ws1 = make_rsrc(gx1, half_bytes);
    // this wave's part of the gate blocks: a resource of its own (an empty part fetches nothing)
    const int per = (NQ + KS - 1) / KS, q_lo = min(NQ, kp * per), q_n = min(NQ - q_lo, per), q_last = max(q_n - 1, 0);
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(gx0 + (size_t)q_lo * BLK, (unsigned)q_n * BLK * 4u),
                                 rs1 = make_rsrc(gx1 + (size_t)q_lo * BLK, (unsigned)q_n * BLK * 4u);
    const int voff = ((rg * 16 + j) * UU + 4 * kq) * 4;
    const int own = slice * NG * BLK * 4;
    lds_word* go = sync + rg;
    lds_word* done = sync + 32 + wave;
    __syncthreads();

    const float* Wl = Ws + q_lo * 256 + kq * 16 + j;
    // bias gradients (column sums of dgx / dgh over batch and time) ride along in registers
    float sb_r[4] = {0.f, 0.f, 0.f, 0.f}, sb_z[4] = {0.f, 0.f, 0.f, 0.f}, sb_n[4] = {0.f, 0.f, 0.f, 0.f},
          sb_m[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef GRU_PROFILE
    const int prof_slot = (wave == 0) ? PROF_SLOT((int)gridDim.x) : -1;
#endif
    const float* const gates_p = in.gates;
    const float* const dout_p = in.dout;
    const int64_t ldd = in.ldd;
    const uintptr_t ptr_bits = reinterpret_cast<uintptr_t>(gates_p) | reinterpret_cast<uintptr_t>(dout_p);
    const bool fast4 = (H % 4) == 0 && H >= 4 && (dout_p == nullptr || (ldd % 4) == 0) && (ptr_bits % 16) == 0;
    const bool fast2 = !fast4 && (H % 2) == 0 && H >= 2 && (dout_p == nullptr || (ldd % 2) == 0) && (ptr_bits % 8) == 0;
    float* const dgx_p = in.dgx;
    float* const dgh_p = LSTM ? in.dgx : in.dgh;
    const uintptr_t st_bits = reinterpret_cast<uintptr_t>(dgx_p) | reinterpret_cast<uintptr_t>(dgh_p);
    const bool fast4s = (H % 4) == 0 && (st_bits % 16) == 0;
    const bool fast2s = !fast4s && (H % 2) == 0 && (st_bits % 8) == 0;
    // The saved values of a step (upstream gradient, gates, previous state: NV 16-byte vectors per lane), every one in
    // ONE memory round trip: unconditional 16-byte (H % 4 == 0) or 8-byte (H % 2 == 0) loads back to back at clamped
    // addresses, masked afterwards (vd).
    auto sv_fetch = [&](int step_n, f32x4 (&v)[NV]) {
        const int fs = in.t0 + T - 1 - step_n;
        const int tt = in.reverse ? Tt - 1 - fs : fs;
        const size_t bt = (size_t)(row_ok ? b : 0) * Tt + tt;
        const float* grow = gates_p + bt * (LSTM ? 7 : 5) * H;
        const float* go_row = dout_p ? dout_p + bt * ldd : grow;
        if (fast4) {
            const int uc = min(ub, H - 4);
            auto ld = [&](const float* row) { return *reinterpret_cast<const f32x4*>(row + uc); };
            v[0] = ld(go_row);
#pragma unroll
            for (int i = 0; i < 5; ++i) v[1 + i] = ld(grow + i * H);
            if constexpr (LSTM) v[6] = ld(grow + 6 * H);
        } else {
            const int ua = min(ub, H - 2), uh = min(ub + 2, H - 2);
            auto ld = [&](const float* row) {
                const f32x2 lo = *reinterpret_cast<const f32x2*>(row + ua), hi = *reinterpret_cast<const f32x2*>(row + uh);
                return (f32x4){lo[0], lo[1], hi[0], hi[1]};
            };
            v[0] = ld(go_row);
#pragma unroll
            for (int i = 0; i < 5; ++i) v[1 + i] = ld(grow + i * H);
            if constexpr (LSTM) v[6] = ld(grow + 6 * H);
        }
    };
    // With a K split the lead never waits for them: part 1 of the row group -- idle until the lead has seen the
    // counter -- fetches the NEXT step's vectors at the top of its step, keeps them in registers until the lead's
    // release (the lead has read this step's by then), and leaves them in LDS.  That took the first 1.5 - 2.2 us
    // (a memory round trip) off every step of the lead, which is the step's critical path.
    const bool sv_stage = args.sv_stage != 0 && KS > 1 && (fast4 || fast2);
    const bool sv_stager = sv_stage && kp == 1;
    if (sv_stager) {
        f32x4 v[NV];
        sv_fetch(0, v);
#pragma unroll
        for (int i = 0; i < NV; ++i) svbuf[i * 64] = v[i];
        lds_signal(sv_ready, 1u);
    }
    for (int step = 0; step < T; ++step) {
        PROF(0);
        // backward walks the forward order in reverse: forward steps t0 + T - 1 down to t0 of the recurrence
        const int fstep = in.t0 + T - 1 - step;
        const int t = in.reverse ? Tt - 1 - fstep : fstep;
        const __amdgpu_buffer_rsrc_t rcur = (step & 1) ? rs1 : rs0;
        float dhz[4] = {0.f, 0.f, 0.f, 0.f};
        if (lead) {
            const __amdgpu_buffer_rsrc_t wcur = (step & 1) ? ws1 : ws0;
            // GRU: d r, d z, d n (input side), d n * r (recurrent side).  LSTM: d i, d f, d g, d o.
            float g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[4] = {0.f, 0.f, 0.f, 0.f}, g2[4] = {0.f, 0.f, 0.f, 0.f},
                  g3[4] = {0.f, 0.f, 0.f, 0.f};
            const size_t bt = (size_t)(row_ok ? b : 0) * Tt + t;
            if (fast4 || fast2) {
                // Every saved value of the step in ONE memory round trip: unconditional 16-byte (H % 4 == 0) or
                // 8-byte (H % 2 == 0) loads back to back; a lane's units past H (pad units of the last slice) and
                // pad rows read a valid, clamped address and are masked afterwards.  The guarded per-value loads of
                // the general path below cost a round trip EACH -- a load under a branch is waited for at the join
                // -- which made this phase 3-7 us of every step.
                f32x4 sv[NV];
                bool vd[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) vd[c] = row_ok && ub + (fast4 || c >= 2 ? 4 : 2) <= H;
                if (sv_stage) {
                    // staged by part 1 of this row group while the previous step ran: LDS reads, no memory round trip
                    lds_await(sv_ready, (unsigned)step + 1u, (ErrWords){args.err, args.sticky});
#pragma unroll
                    for (int i = 0; i < NV; ++i) sv[i] = svbuf[i * 64];
                } else {
                    sv_fetch(step, sv);
                }
                const f32x4 r_go = sv[0], r0 = sv[1], r1 = sv[2], r2 = sv[3], r3 = sv[4], r4 = sv[5], r6 = sv[NV - 1];
                __builtin_amdgcn_sched_barrier(0);
                const bool in_len = t < len;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const bool valid = vd[c], live = valid && in_len;
                    const float g = dh[c] + (dout_p ? r_go[c] : 0.f);
                    if constexpr (LSTM) {
                        const float dct = dc[c] + g * r3[c] * (1.f - r6[c] * r6[c]);
                        g0[c] = live ? dct * r2[c] * r0[c] * (1.f - r0[c]) : 0.f;
                        g1[c] = live ? dct * r4[c] * r1[c] * (1.f - r1[c]) : 0.f;
                        g2[c] = live ? dct * r0[c] * (1.f - r2[c] * r2[c]) : 0.f;
                        g3[c] = live ? g * r6[c] * r3[c] * (1.f - r3[c]) : 0.f;
                        dc[c] = live ? dct * r1[c] : dc[c];
                        dhz[c] = live ? 0.f : (valid ? dh[c] : 0.f);
                    } else {
                        const float dn = g * (1.f - r1[c]) * (1.f - r2[c] * r2[c]);
                        g2[c] = live ? dn : 0.f;
                        g1[c] = live ? g * (r4[c] - r2[c]) * r1[c] * (1.f - r1[c]) : 0.f;
                        g0[c] = live ? dn * r3[c] * r0[c] * (1.f - r0[c]) : 0.f;
                        g3[c] = live ? dn * r0[c] : 0.f;
                        dhz[c] = live ? g * r1[c] : (valid ? dh[c] : 0.f);
                    }
                }
            } else
            if (row_ok && n_ok > 0) {
                const bool live = t < len;
                if (live) {
                    float go_[4] = {0.f, 0.f, 0.f, 0.f};
                    if (in.dout) load4(go_, in.dout + bt * in.ldd + ub, vec_h && (in.ldd % 4) == 0, n_ok);
                    const float* gs = in.gates + bt * (LSTM ? 7 : 5) * H + ub;
                    float s0[4], s1[4], s2[4], s3[4], s4[4];
                    load4(s0, gs, vec_h, n_ok);
                    load4(s1, gs + H, vec_h, n_ok);
                    load4(s2, gs + 2 * H, vec_h, n_ok);
                    load4(s3, gs + 3 * H, vec_h, n_ok);
                    if constexpr (LSTM) {
                        float tc[4];
                        load4(s4, gs + 4 * H, vec_h, n_ok);   // c_{t-1}
                        load4(tc, gs + 6 * H, vec_h, n_ok);   // tanh c_t
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const bool ok = c < n_ok;
                            const float g = dh[c] + go_[c];
                            const float dct = dc[c] + g * s3[c] * (1.f - tc[c] * tc[c]);
                            g0[c] = ok ? dct * s2[c] * s0[c] * (1.f - s0[c]) : 0.f;
                            g1[c] = ok ? dct * s4[c] * s1[c] * (1.f - s1[c]) : 0.f;
                            g2[c] = ok ? dct * s0[c] * (1.f - s2[c] * s2[c]) : 0.f;
                            g3[c] = ok ? g * tc[c] * s3[c] * (1.f - s3[c]) : 0.f;
                            dc[c] = ok ? dct * s1[c] : 0.f;
                            dhz[c] = 0.f;
                        }
                    } else {
                        load4(s4, gs + 4 * H, vec_h, n_ok);   // h_{t-1}
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float g = dh[c] + go_[c];
                            const float dn = g * (1.f - s1[c]) * (1.f - s2[c] * s2[c]);
                            const bool ok = c < n_ok;
                            g2[c] = ok ? dn : 0.f;
                            g1[c] = ok ? g * (s4[c] - s2[c]) * s1[c] * (1.f - s1[c]) : 0.f;
                            g0[c] = ok ? dn * s3[c] * s0[c] * (1.f - s0[c]) : 0.f;
                            g3[c] = ok ? dn * s0[c] : 0.f;
                            dhz[c] = g * s1[c];
                        }
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) dhz[c] = dh[c];   // (LSTM: d c passes through unchanged)
                }
            }
            // this slice's gate blocks (pad units / pad rows publish zeros); hand over first, the history stores
            // then overlap the wait
#pragma unroll
            for (int c = 0; c < 4; ++c) { sb_r[c] += g0[c]; sb_z[c] += g1[c]; sb_n[c] += g2[c]; sb_m[c] += g3[c]; }
#ifdef GRU_PROFILE
            asm volatile("s_nop 0" ::"v"(g0[0]), "v"(g1[0]), "v"(g2[0]), "v"(g3[0]));  // the saved gates are there
#endif
            PROF(1);
            st_sc1((f32x4){g0[0], g0[1], g0[2], g0[3]}, wcur, voff, own);
            st_sc1((f32x4){g1[0], g1[1], g1[2], g1[3]}, wcur, voff, own + BLK * 4);
            if constexpr (LSTM) {
                st_sc1((f32x4){g2[0], g2[1], g2[2], g2[3]}, wcur, voff, own + 2 * BLK * 4);
                st_sc1((f32x4){g3[0], g3[1], g3[2], g3[3]}, wcur, voff, own + 3 * BLK * 4);
            } else {
                st_sc1((f32x4){g3[0], g3[1], g3[2], g3[3]}, wcur, voff, own + 2 * BLK * 4);
            }
            publish(cnt);
            PROF(2);
            auto store_all = [&](auto st) {   // one predicate for all stores of the step (see gru_fwd_kernel)
                float* ox = dgx_p + bt * K + ub;
                st(ox, g0); st(ox + H, g1); st(ox + 2 * H, g2);
                if constexpr (LSTM) {
                    st(ox + 3 * H, g3);
                } else {
                    float* oh = dgh_p + bt * K + ub;
                    st(oh, g0); st(oh + H, g1); st(oh + 2 * H, g3);
                }
            };
            if (fast4s) {
                if (row_ok && ub + 4 <= H)
                    store_all([](float* q, const float (&v)[4]) {   // streaming: read next by a GEMM, not by this launch
                        __builtin_nontemporal_store((f32x4){v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(q));
                    });
            } else if (fast2s) {
                if (row_ok && ub + 2 <= H)
                    store_all([](float* q, const float (&v)[4]) { *reinterpret_cast<f32x2*>(q) = (f32x2){v[0], v[1]}; });
                if (row_ok && ub + 4 <= H)
                    store_all([](float* q, const float (&v)[4]) { *reinterpret_cast<f32x2*>(q + 2) = (f32x2){v[2], v[3]}; });
            } else if (row_ok && n_ok > 0) {
                store_all([&](float* q, const float (&v)[4]) { store4(q, v, false, n_ok); });
            }
            PROF(3);
            await(cnt, (unsigned)(step + 1) * NS, (ErrWords){args.err, args.sticky});
            PROF(4);
            if (KS > 1) lds_signal(go, (unsigned)step + 1u);
        } else if (sv_stager && step + 1 < T) {
            f32x4 v[NV];
            sv_fetch(step + 1, v);
            lds_await(go, (unsigned)step + 1u, (ErrWords){args.err, args.sticky});
#pragma unroll
            for (int i = 0; i < NV; ++i) svbuf[i * 64] = v[i];
            lds_signal(sv_ready, (unsigned)step + 2u);
        } else {
            lds_await(go, (unsigned)step + 1u, (ErrWords){args.err, args.sticky});
        }

        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 cur[CB], nxt[CB];
#pragma unroll
        for (int i = 0; i < CB; ++i) cur[i] = ld_sc1(rcur, voff + min(i, q_last) * BLK * 4, 0);
        // as in gru_fwd_kernel: unconditional prefetch, clamped into the part; branch-free chunks
        int q0 = 0;
        for (; q0 + CB < q_n; q0 += CB) {
#pragma unroll
            for (int i = 0; i < CB; ++i) nxt[i] = ld_sc1(rcur, voff + min(q0 + CB + i, q_last) * BLK * 4, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_blocks<CB, CB>(acc, Wl + q0 * 256, cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < CB; ++i) cur[i] = nxt[i];
        }
        mfma_blocks_tail<CB>(acc, Wl + q0 * 256, cur, q_n - q0);  // the last chunk: 1 .. CB blocks (0: none)
        f32x4 sum = acc[0] + acc[1] + acc[2];
#ifdef GRU_PROFILE
        asm volatile("s_nop 0" ::"v"(sum));  // the products are there
#endif
        if (lead) PROF(5);
        if (!lead) {
            parts[(slot0 + kp - 1) * 64 + lane] = sum;
            lds_signal(done, (unsigned)step + 1u);
            continue;
        }
        for (int p = 1; p < KS; ++p) {
            lds_await(sync + 32 + rg * KS + (p + rg % KS) % KS, (unsigned)step + 1u, (ErrWords){args.err, args.sticky});
            sum += parts[(slot0 + p - 1) * 64 + lane];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) dh[c] = dhz[c] + sum[c];
        PROF(6);
    }
    if (!lead) return;
    if (row_ok && n_ok > 0 && in.dh0) store4(in.dh0 + (size_t)b * H + ub, dh, vec_h, n_ok);
    if (LSTM && row_ok && n_ok > 0 && in.dc0) store4(in.dc0 + (size_t)b * H + ub, dc, vec_h, n_ok);
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
                        if (LSTM) atomicAdd(in.dbias_x + 3 * H + u, sb_m[c]);
                    }
                    if (in.dbias_h) {
                        atomicAdd(in.dbias_h + u, sb_r[c]); atomicAdd(in.dbias_h + H + u, sb_z[c]);
                        if (LSTM) {  // the gate pre-activations are sums of both products: the same gradient
                            atomicAdd(in.dbias_h + 2 * H + u, sb_n[c]); atomicAdd(in.dbias_h + 3 * H + u, sb_m[c]);
                        } else {
                            atomicAdd(in.dbias_h + 2 * H + u, sb_m[c]);
                        }
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

// exchange buffers are sized for the backward pass of either cell (4 gate blocks per slice: the LSTM's; the GRU's
// backward uses 3, forward 1).  The sizes
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
        off += (size_t)2 * ((B[i] + 127) / 128) * 128 * 4 * round16(H[i]) * sizeof(float);  // either block size
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

const void* fwd_kernel(int wv, bool lstm) {
#define PICK(L)                                                                              \
    switch (wv) {                                                                            \
        case 1: return reinterpret_cast<const void*>(gru_fwd_kernel<1, L>);                  \
        case 2: return reinterpret_cast<const void*>(gru_fwd_kernel<2, L>);                  \
        case 4: return reinterpret_cast<const void*>(gru_fwd_kernel<4, L>);                  \
        case 8: return reinterpret_cast<const void*>(gru_fwd_kernel<8, L>);                  \
        default: return reinterpret_cast<const void*>(gru_fwd_kernel<16, L>);                \
    }
    if (lstm) { PICK(true) }
    PICK(false)
#undef PICK
}
const void* bwd_kernel(int wv, bool lstm) {
#define PICK(L)                                                                              \
    switch (wv) {                                                                            \
        case 1: return reinterpret_cast<const void*>(gru_bwd_kernel<1, L>);                  \
        case 2: return reinterpret_cast<const void*>(gru_bwd_kernel<2, L>);                  \
        case 4: return reinterpret_cast<const void*>(gru_bwd_kernel<4, L>);                  \
        case 8: return reinterpret_cast<const void*>(gru_bwd_kernel<8, L>);                  \
        default: return reinterpret_cast<const void*>(gru_bwd_kernel<16, L>);                \
    }
    if (lstm) { PICK(true) }
    PICK(false)
#undef PICK
}

inline int slice_bytes_for(int H_max, bool lstm) {
    return (lstm ? 4 : 3) * (round16(H_max) / UU) * 256 * (int)sizeof(float);
}
// LDS of a launch: the resident W_hh slice; with the K split (some instance has more waves than row groups) the
// waves' partial sums and the sync words behind it
inline int lds_bytes_for(bool backward, bool lstm, int H_max, int wv, int min_rows) {
    const int parts = wv - min_rows / 16;  // non-lead waves of the instance with the fewest row groups
    return slice_bytes_for(H_max, lstm) +
           (parts > 0 ? parts * (backward ? 1 : lstm ? 4 : 3) * PART_BYTES + SYNC_WORDS * 4 : 0);
}
// backward with a K split: what ONE instance needs when its row groups stage the next step's saved values in LDS
// (gru_bwd_kernel: NV vectors of SV_BYTES per row group behind the instance's own slice, partial sums and sync words)
inline int lds_bytes_staged(bool lstm, int H, int wv, int rows) {
    const int groups = rows / 16, parts = wv - groups;
    if (parts <= 0) return slice_bytes_for(H, lstm);
    return slice_bytes_for(H, lstm) + parts * PART_BYTES + SYNC_WORDS * 4 + groups * (lstm ? 7 : 6) * SV_BYTES;
}

// How many workgroups of a persistent launch the chip holds AT ONCE: the runtime's occupancy answer for this
// kernel, block size and LDS size, times the CU count.  Every workgroup of a launch must be resident (the step
// hand-off spins on peers), so this is the bound on the grid -- not the CU count: an H = 300 slice is 57 KB of
// the CU's 160 KB of LDS, two such 4-wave workgroups share a CU.  0 = it does not fit at all.
int resident_workgroups(bool backward, bool lstm, int wv, int lds_bytes) {
    if (lds_bytes > LDS_MAX || lds_bytes <= 0) return 0;
    // the answer depends on (kernel, block size, LDS size) only: asked once (sizes are multiples of 256 B)
    static int cache[4][5][LDS_MAX / 256 + 1];  // 0 = not asked yet, -1 = asked: none
    const int wi = wv == 1 ? 0 : wv == 2 ? 1 : wv == 4 ? 2 : wv == 8 ? 3 : 4;
    int& c = cache[(backward ? 1 : 0) + (lstm ? 2 : 0)][wi][lds_bytes / 256];
    if (c == 0) {
        const void* k = backward ? bwd_kernel(wv, lstm) : fwd_kernel(wv, lstm);
        int per_cu = 0;
        // the attribute is per kernel, not per launch: raise it to the CU's LDS once, whatever this launch needs
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, 64 * wv, (size_t)lds_bytes) != hipSuccess)
            per_cu = 0;
        if (per_cu > MAX_WG_PER_CU) per_cu = MAX_WG_PER_CU;
        c = per_cu > 0 ? per_cu * cu_count() : -1;
    }
    return c > 0 ? c : 0;
}

// Cell height (batch rows per workgroup) of every instance, the launch's waves per workgroup and its LDS size.  An
// instance may ask for its own height (inst.rows in {16, 32, 64, 128}); 0 = the library's choice for the whole
// launch.  Tried in this order, the first whose grid is resident wins:
//   64-row cells in 16-wave workgroups (4 row groups x K split 4), 128-row cells in 16-wave workgroups (8 x 2),
//   then the layouts without K split: 64 rows / 4 waves (several workgroups per CU), 128 rows / 8 waves.
// DSSVAE_GRU_WAVES=8 forces 128 rows / 8 waves, DSSVAE_GRU_KSPLIT=0 skips the K-split layouts (tests, A/B).
// -> waves per workgroup (1 .. 16), 0 = not resident any way, -1 = bad rows value
int pick_rows(bool backward, bool lstm, int n_inst, const int32_t* B, const int32_t* H, const int32_t* want,
              int* rows_out, int* lds_out, int* sv_stage_out = nullptr) {
    int max_h = 0;
    bool any_auto = false;
    for (int i = 0; i < n_inst; ++i) {
        if (H[i] > max_h) max_h = H[i];
        if (want[i] == 0) any_auto = true;
        else if (want[i] != 16 && want[i] != 32 && want[i] != 64 && want[i] != 128) return -1;
    }
    const char* f = getenv("DSSVAE_GRU_WAVES");
    const bool force8 = f && atoi(f) == 8;
    const char* ks = getenv("DSSVAE_GRU_KSPLIT");
    const bool ksplit_ok = !force8 && !(ks && atoi(ks) == 0);
    struct Cand { int rows, wv16; };
    const Cand cand[4] = {{64, 1}, {128, 1}, {force8 ? 128 : 64, 0}, {128, 0}};
    for (int c = ksplit_ok ? 0 : 2; c < 4; ++c) {
        if (!any_auto && (c == 1 || c == 3)) continue;  // every height is given: the row candidates do not differ
        int wg = 0, max_rows = 16, min_rows = 128;
        for (int i = 0; i < n_inst; ++i) {
            rows_out[i] = want[i] ? want[i] : cand[c].rows;
            wg += ((B[i] + rows_out[i] - 1) / rows_out[i]) * (round16(H[i]) / UU);
            if (rows_out[i] > max_rows) max_rows = rows_out[i];
            if (rows_out[i] < min_rows) min_rows = rows_out[i];
        }
        const int wv = cand[c].wv16 ? 16 : max_rows / 16;
        int lds = lds_bytes_for(backward, lstm, max_h, wv, min_rows);
        if (slice_bytes_for(max_h, lstm) > LDS_BUDGET) return 0;
        // backward, K split: the staged saved values ride along when every instance's own need still fits the CU
        // (H = 600 in 64-row cells: 117 KB slice + 12 + 24 = 154 KB); else the leads load them themselves
        int sv_stage = 0;
        if (backward && cand[c].wv16) {
            int need = lds;
            for (int i = 0; i < n_inst; ++i) need = max(need, lds_bytes_staged(lstm, H[i], wv, rows_out[i]));
            if (need <= LDS_MAX) { lds = need; sv_stage = 1; }
        }
        const int res = resident_workgroups(backward, lstm, wv, lds);
        static const bool verbose = getenv("DSSVAE_GRU_VERBOSE") != nullptr;
        if (verbose)
            fprintf(stderr, "[dssvae %s %s] layout %d: rows %d..%d, %d waves, LDS %d B%s, grid %d, resident %d\n",
                    lstm ? "lstm" : "gru", backward ? "bwd" : "fwd", c, min_rows, max_rows, wv, lds,
                    sv_stage ? " (saved values staged)" : "", wg, res);
        if (wg <= res) {
            *lds_out = lds;
            if (sv_stage_out) *sv_stage_out = sv_stage;
            return wv;
        }
    }
    return 0;
}

// the error word registered for the current device (dssvae_gru_error_word), or NULL
unsigned* g_sticky[64];
unsigned* sticky_word() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    return g_sticky[dev];
}

}  // namespace

extern "C" int dssvae_gru_error_word(void* word) {
    int dev = -1;
    const hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 64) return DSSVAE_EINVAL;
    g_sticky[dev] = static_cast<unsigned*>(word);
    return DSSVAE_OK;
}

extern "C" int64_t dssvae_gru_workspace_bytes(int n_inst, const int32_t* B_host, const int32_t* H_host) {
    if (n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !B_host || !H_host) return DSSVAE_EINVAL;
    for (int i = 0; i < n_inst; ++i)
        if (B_host[i] <= 0 || H_host[i] <= 0) return DSSVAE_EINVAL;
    return (int64_t)plan(n_inst, B_host, H_host).total_bytes;
}

extern "C" int dssvae_gru_resident_workgroups(int H_max, int rows_max, int backward, int* n_host) {
    if (!n_host || H_max <= 0 || (rows_max != 16 && rows_max != 32 && rows_max != 64 && rows_max != 128))
        return DSSVAE_EINVAL;
    *n_host = resident_workgroups(backward != 0, false, rows_max / 16, slice_bytes_for(H_max, false));
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_layout(int n_inst, const int32_t* B_host, const int32_t* H_host, const int32_t* rows_host,
                                 int backward, int cell, int* waves_host) {
    if (n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !B_host || !H_host || !rows_host || !waves_host ||
        (cell != DSSVAE_CELL_GRU && cell != DSSVAE_CELL_LSTM))
        return DSSVAE_EINVAL;
    for (int i = 0; i < n_inst; ++i)
        if (B_host[i] <= 0 || H_host[i] <= 0) return DSSVAE_EINVAL;
    int rows[DSSVAE_GRU_MAX_INST], lds = 0;
    const int wv = pick_rows(backward != 0, cell == DSSVAE_CELL_LSTM, n_inst, B_host, H_host, rows_host, rows, &lds);
    if (wv < 0) return DSSVAE_EINVAL;
    *waves_host = wv;
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_fwd(const dssvae_gru_fwd_inst* inst, int n_inst, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (!inst || n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !workspace) return DSSVAE_EINVAL;
    int32_t Bs[DSSVAE_GRU_MAX_INST], Hs[DSSVAE_GRU_MAX_INST], want[DSSVAE_GRU_MAX_INST];
    const bool lstm = inst[0].cell == DSSVAE_CELL_LSTM;
    for (int i = 0; i < n_inst; ++i) {
        const dssvae_gru_fwd_inst& x = inst[i];
        if ((x.cell != DSSVAE_CELL_GRU && x.cell != DSSVAE_CELL_LSTM) || (x.cell == DSSVAE_CELL_LSTM) != lstm)
            return DSSVAE_EINVAL;  // one cell type per launch
        if (!x.xproj || !x.whh || !x.bhh || !x.out || x.B <= 0 || x.T <= 0 || x.H <= 0 ||
            x.ldx < (lstm ? 4 : 3) * x.H ||
            x.ldo < x.H || x.t0 < 0 || (x.T_total > 0 && x.t0 + x.T > x.T_total) || (x.T_total <= 0 && x.t0 != 0))
            return DSSVAE_EINVAL;
        Bs[i] = x.B; Hs[i] = x.H; want[i] = x.rows;
    }
    const Layout L = plan(n_inst, Bs, Hs);
    if ((int64_t)L.total_bytes > workspace_bytes || !aligned16(workspace)) return DSSVAE_EINVAL;
    FwdArgs args;
    args.n_inst = n_inst;
    args.err = reinterpret_cast<unsigned*>(workspace);
    args.sticky = sticky_word();
    int rows[DSSVAE_GRU_MAX_INST];
    int lds_bytes = 0;
    const int wv = pick_rows(false, lstm, n_inst, Bs, Hs, want, rows, &lds_bytes);
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
    hipStream_t s = (hipStream_t)stream;
    const int e = dssvae_zero2d(reinterpret_cast<float*>(workspace), (int64_t)(L.header_bytes / 4), 1,
                                (int)(L.header_bytes / 4), s);
    if (e != DSSVAE_OK) return e;
    void* kargs[] = {&args};
    const hipError_t le = hipLaunchKernel(fwd_kernel(wv, lstm), dim3(wg), dim3(64 * wv), kargs, lds_bytes, s);
    if (le != hipSuccess) return (int)le;
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_bwd(const dssvae_gru_bwd_inst* inst, int n_inst, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (!inst || n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !workspace) return DSSVAE_EINVAL;
    int32_t Bs[DSSVAE_GRU_MAX_INST], Hs[DSSVAE_GRU_MAX_INST], want[DSSVAE_GRU_MAX_INST];
    const bool lstm = inst[0].cell == DSSVAE_CELL_LSTM;
    for (int i = 0; i < n_inst; ++i) {
        const dssvae_gru_bwd_inst& x = inst[i];
        if ((x.cell != DSSVAE_CELL_GRU && x.cell != DSSVAE_CELL_LSTM) || (x.cell == DSSVAE_CELL_LSTM) != lstm)
            return DSSVAE_EINVAL;
        if (!x.whh || !x.gates || !x.dgx || (!lstm && !x.dgh) || x.B <= 0 || x.T <= 0 || x.H <= 0 ||
            (x.dout && x.ldd < x.H) ||
            x.t0 < 0 || (x.T_total > 0 && x.t0 + x.T > x.T_total) || (x.T_total <= 0 && x.t0 != 0))
            return DSSVAE_EINVAL;
        Bs[i] = x.B; Hs[i] = x.H; want[i] = x.rows;
    }
    const Layout L = plan(n_inst, Bs, Hs);
    if ((int64_t)L.total_bytes > workspace_bytes || !aligned16(workspace)) return DSSVAE_EINVAL;
    BwdArgs args;
    args.n_inst = n_inst;
    args.err = reinterpret_cast<unsigned*>(workspace);
    args.sticky = sticky_word();
    int rows[DSSVAE_GRU_MAX_INST];
    int lds_bytes = 0;
    int sv_stage = 0;
    const int wv = pick_rows(true, lstm, n_inst, Bs, Hs, want, rows, &lds_bytes, &sv_stage);
    if (wv < 0) return DSSVAE_EINVAL;
    if (wv == 0) return DSSVAE_ETOOBIG;
    args.sv_stage = sv_stage;
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
    hipStream_t s = (hipStream_t)stream;
    const int e = dssvae_zero2d(reinterpret_cast<float*>(workspace), (int64_t)(L.header_bytes / 4), 1,
                                (int)(L.header_bytes / 4), s);
    if (e != DSSVAE_OK) return e;
    void* kargs[] = {&args};
    const hipError_t le = hipLaunchKernel(bwd_kernel(wv, lstm), dim3(wg), dim3(64 * wv), kargs, lds_bytes, s);
    if (le != hipSuccess) return (int)le;
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

#ifdef GRU_PROFILE
extern "C" int dssvae_gru_profile_read(void* dst_host, int64_t bytes) {
    if (bytes > (int64_t)sizeof(g_prof)) bytes = sizeof(g_prof);
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_prof), (size_t)bytes, 0, hipMemcpyDeviceToHost);
}
extern "C" int dssvae_gru_profile_part(void* dst_host) {
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_prof_part), sizeof(g_prof_part), 0, hipMemcpyDeviceToHost);
}
extern "C" int dssvae_gru_profile_simd(void* dst_host) {
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(g_simd), sizeof(g_simd), 0, hipMemcpyDeviceToHost);
}
#endif

extern "C" int dssvae_gru_check(void* workspace, void* stream) {
    if (!workspace) return DSSVAE_EINVAL;
    unsigned err = 0;
    hipError_t e = hipMemcpyAsync(&err, workspace, sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    return err ? DSSVAE_ETIMEOUT : DSSVAE_OK;
}

// ---- residency rehearsal --------------------------------------------------------------------------------------
// A neighbour that STAYS on the chip: `workgroups` workgroups of 256 threads, each holding `lds_bytes` of LDS, wait
// until the host writes a non-zero word to `release` (coherent host memory, read with a system-scope load by one
// lane per workgroup) or `max_ms` have passed on the 100 MHz wall clock -- the exit every wave reaches.  What a
// communication kernel looks like to a persistent recurrence launched beside it (tests/test_overlap_gpu.py).
namespace {
__global__ __launch_bounds__(256) void hold_cus_kernel(const unsigned* release, unsigned long long max_ticks,
                                                       unsigned long long* held_ticks) {
    extern __shared__ unsigned hold_lds[];  // the launch's dynamic LDS: held, not used
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned long long now = t0;
        while (now - t0 < max_ticks) {
            if (__hip_atomic_load(release, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
            __builtin_amdgcn_s_sleep(64);
            now = wall_clock64();
        }
        if (held_ticks) held_ticks[blockIdx.x] = now - t0;
        hold_lds[0] = (unsigned)(now - t0);
    }
    __syncthreads();  // the other three waves wait here, resident
}
}  // namespace

extern "C" int dssvae_hold_cus(const uint32_t* release, int workgroups, int lds_bytes, int max_ms,
                               unsigned long long* held_ticks, void* stream) {
    if (!release || workgroups < 1 || workgroups > 256 || lds_bytes < 4 || lds_bytes > 64 * 1024 || max_ms < 1 ||
        max_ms > 5000)
        return DSSVAE_EINVAL;
    hipLaunchKernelGGL(hold_cus_kernel, dim3(workgroups), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, release,
                       (unsigned long long)max_ms * 100000ull, held_ticks);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}
