// Persistent GRU recurrence for gfx950 (forward and backward through time).
//
// Decomposition: an "instance" is one independent recurrence (a direction of an encoder, a decoder
// layer, a task head).  Each instance is cut into (batch block of 64 rows) x (slice of 16 hidden
// units); one 256-thread workgroup owns one cell of that grid for ALL timesteps:
//   * its W_hh slice (3 gates x 16 units x H, fp32) is loaded into LDS once and stays resident;
//   * per step each of its 4 waves multiplies 16 batch rows of h_{t-1} (B operand, streamed from
//     the L2-resident exchange buffer as float4) with the resident slice (A operand, ds_read_b32)
//     on v_mfma_f32_16x16x4_f32, then applies the fused gate math in registers;
//   * the new 64x16 slice of h_t goes to a ping-pong exchange buffer; workgroups of the same
//     (instance, batch block) meet on one monotonic arrival counter per step -- there is no
//     grid-wide barrier, and independent instances / batch blocks never wait for each other.
// Cross-workgroup visibility follows the agent-scope release / acquire recipe (producer: every
// wave drains its stores, workgroup barrier, one lane release-fences and adds to the counter;
// consumer: one lane polls relaxed, acquire-fences once, workgroup barrier, plain loads).
// All workgroups must be co-resident: the host caps the grid at the CU count (1 workgroup per
// CU); every spin is bounded and reports through an error word instead of hanging.
//
// Roofline: per step 2*B*H*3H flop on fp32 MFMA; in practice latency-bound by the step hand-off.
#include "common.h"

namespace {

constexpr int BB = 64;   // batch rows per workgroup (4 waves x 16 rows)
constexpr int UU = 16;   // hidden units per workgroup (M of the 16x16x4 MFMA)
constexpr int NTHR = 256;
constexpr unsigned SPIN_LIMIT = 1u << 26;
constexpr int CNT_STRIDE = 32;  // one 128-B line per counter

struct FwdDev {
    dssvae_gru_fwd_inst in;
    int wg_begin, n_slices, n_bblocks, H16;
    unsigned* counters;
    float* hx;  // [2][B][H16]
};
struct FwdArgs {
    FwdDev inst[DSSVAE_GRU_MAX_INST];
    int n_inst;
    unsigned* err;
};
struct BwdDev {
    dssvae_gru_bwd_inst in;
    int wg_begin, n_slices, n_bblocks, K16;  // K16 = roundup(3H,16)
    unsigned* counters;
    float* gx;  // [2][B][K16]
};
struct BwdArgs {
    BwdDev inst[DSSVAE_GRU_MAX_INST];
    int n_inst;
    unsigned* err;
};

__device__ __forceinline__ int lds_row(int k) { return (k & ~15) | ((k & 3) << 2) | ((k >> 2) & 3); }

__device__ __forceinline__ void publish(unsigned* cnt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__device__ __forceinline__ void await(unsigned* cnt, unsigned target, unsigned* err) {
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 1023u) == 0) {
                if (spins > SPIN_LIMIT ||
                    __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
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

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void gru_fwd_kernel(FwdArgs args) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [3][H16][16]
    int ii = 0;
    while (ii + 1 < args.n_inst && (int)blockIdx.x >= args.inst[ii + 1].wg_begin) ++ii;
    const FwdDev& d = args.inst[ii];
    const dssvae_gru_fwd_inst& in = d.in;
    const int local = blockIdx.x - d.wg_begin;
    const int bblock = local / d.n_slices, slice = local % d.n_slices;
    const int H = in.H, H16 = d.H16, T = in.T, B = in.B;
    const int u0 = slice * UU;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int b = bblock * BB + wave * 16 + j;
    const bool row_ok = b < B;
    const int ub = u0 + 4 * kq;                       // first of this lane's 4 units
    const int n_ok = max(0, min(4, H - ub));          // how many of them exist
    unsigned* cnt = d.counters + (size_t)bblock * CNT_STRIDE;
    const bool vec_h = (H % 4) == 0;

    // resident weights: Ws[g][lds_row(k)][i] = whh[(g*H + u0 + i)*H + k]
    for (int idx = threadIdx.x; idx < 3 * H16 * UU; idx += NTHR) {
        const int g = idx / (H16 * UU), rem = idx % (H16 * UU);
        const int i = rem / H16, k = rem % H16;  // k fastest: coalesced global reads
        const int u = u0 + i;
        const float w = (u < H && k < H) ? in.whh[(size_t)(g * H + u) * H + k] : 0.f;
        Ws[(g * H16 + lds_row(k)) * UU + i] = w;
    }
    float bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) bh[g][c] = (c < n_ok) ? in.bhh[g * H + ub + c] : 0.f;
    const int len = (in.lengths && row_ok) ? in.lengths[b] : T;

    // state 0 into the exchange buffer (pad columns are zero and stay zero)
    float hcur[4] = {0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
        if (in.h0) load4(hcur, in.h0 + (size_t)b * H + ub, vec_h, n_ok);
        store4(d.hx + (size_t)b * H16 + ub, hcur, true, 4);
    }
    publish(cnt);

    for (int step = 0; step < T; ++step) {
        const int t = in.reverse ? T - 1 - step : step;
        const float* hin = d.hx + (size_t)(step & 1) * B * H16;
        float* hout = d.hx + (size_t)((step + 1) & 1) * B * H16;
        await(cnt, (unsigned)(step + 1) * d.n_slices, args.err);

        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* hrow = hin + (size_t)(row_ok ? b : 0) * H16 + 4 * kq;
        const float* wl = Ws + kq * UU + j;
#pragma unroll 2
        for (int s = 0; s < H16; s += 16) {
            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row_ok) hv = *reinterpret_cast<const float4*>(hrow + s);
            const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float* w = wl + (s + 4 * c) * UU;
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[g * H16 * UU], hvv[c], acc[g], 0, 0, 0);
            }
        }
        // fused gate math: lane holds units ub..ub+3 of batch row b (pad units keep writing zeros)
        if (row_ok) {
            const size_t bt = (size_t)b * T + t;
            const float* xp = in.xproj + bt * in.ldx;
            float xr[4], xz[4], xn[4];
            const bool vec_x = vec_h && (in.ldx % 4) == 0;
            load4(xr, xp + ub, vec_x, n_ok);
            load4(xz, xp + H + ub, vec_x, n_ok);
            load4(xn, xp + 2 * H + ub, vec_x, n_ok);
            const bool live = t < len;
            float r[4], z[4], n[4], hn[4], hnew[4], outv[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                r[c] = sigm(xr[c] + acc[0][c] + bh[0][c]);
                z[c] = sigm(xz[c] + acc[1][c] + bh[1][c]);
                hn[c] = acc[2][c] + bh[2][c];
                n[c] = tanhf(xn[c] + r[c] * hn[c]);
                const float hv = (1.f - z[c]) * n[c] + z[c] * hcur[c];
                hnew[c] = (c < n_ok) ? (live ? hv : hcur[c]) : 0.f;
                outv[c] = live ? hv : 0.f;
            }
            store4(hout + (size_t)b * H16 + ub, hnew, true, 4);
            store4(in.out + bt * in.ldo + ub, outv, vec_h && (in.ldo % 4) == 0, n_ok);
            if (in.gates) {
                float* gs = in.gates + bt * 5 * H + ub;
                store4(gs, r, vec_h, n_ok);
                store4(gs + H, z, vec_h, n_ok);
                store4(gs + 2 * H, n, vec_h, n_ok);
                store4(gs + 3 * H, hn, vec_h, n_ok);
                store4(gs + 4 * H, hcur, vec_h, n_ok);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) hcur[c] = hnew[c];
        }
        if (step + 1 < T) publish(cnt);
    }
    if (row_ok && n_ok > 0 && in.h_last) store4(in.h_last + (size_t)b * H + ub, hcur, vec_h, n_ok);
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHR) void gru_bwd_kernel(BwdArgs args) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [K16][16]: Ws[lds_row(c)][i] = whh[c][u0+i]
    int ii = 0;
    while (ii + 1 < args.n_inst && (int)blockIdx.x >= args.inst[ii + 1].wg_begin) ++ii;
    const BwdDev& d = args.inst[ii];
    const dssvae_gru_bwd_inst& in = d.in;
    const int local = blockIdx.x - d.wg_begin;
    const int bblock = local / d.n_slices, slice = local % d.n_slices;
    const int H = in.H, K16 = d.K16, T = in.T, B = in.B, K = 3 * H;
    const int u0 = slice * UU;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int b = bblock * BB + wave * 16 + j;
    const bool row_ok = b < B;
    const int ub = u0 + 4 * kq;
    const int n_ok = max(0, min(4, H - ub));
    unsigned* cnt = d.counters + (size_t)bblock * CNT_STRIDE;
    const bool vec_h = (H % 4) == 0;

    for (int idx = threadIdx.x; idx < K16 * UU; idx += NTHR) {
        const int c = idx / UU, i = idx % UU;  // i fastest: 16 consecutive floats of a W_hh row
        const int u = u0 + i;
        Ws[lds_row(c) * UU + i] = (c < K && u < H) ? in.whh[(size_t)c * H + u] : 0.f;
    }
    const int len = (in.lengths && row_ok) ? in.lengths[b] : T;
    float dh[4] = {0.f, 0.f, 0.f, 0.f};
    if (row_ok && in.dh_last) load4(dh, in.dh_last + (size_t)b * H + ub, vec_h, n_ok);
    // pad columns [3H, K16) of both exchange buffers must read as zero
    if (slice == 0 && row_ok && K16 > K) {
        for (int c = K + kq; c < K16; c += 4) {
            d.gx[(size_t)b * K16 + c] = 0.f;
            d.gx[(size_t)(B + b) * K16 + c] = 0.f;
        }
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        // backward walks the forward order in reverse
        const int fstep = T - 1 - step;
        const int t = in.reverse ? T - 1 - fstep : fstep;
        float* gcur = d.gx + (size_t)(step & 1) * B * K16;
        float dhz[4] = {0.f, 0.f, 0.f, 0.f};
        if (row_ok && n_ok > 0) {
            const size_t bt = (size_t)b * T + t;
            const bool live = t < len;
            float dgr[4], dgz[4], dgn[4], dgm[4];
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
                    dgn[c] = dn;
                    dgz[c] = g * (hp[c] - n[c]) * z[c] * (1.f - z[c]);
                    dgr[c] = dn * hn[c] * r[c] * (1.f - r[c]);
                    dgm[c] = dn * r[c];
                    dhz[c] = g * z[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) { dgr[c] = dgz[c] = dgn[c] = dgm[c] = 0.f; dhz[c] = dh[c]; }
            }
            float* ox = in.dgx + bt * K + ub;
            float* oh = in.dgh + bt * K + ub;
            store4(ox, dgr, vec_h, n_ok); store4(ox + H, dgz, vec_h, n_ok); store4(ox + 2 * H, dgn, vec_h, n_ok);
            store4(oh, dgr, vec_h, n_ok); store4(oh + H, dgz, vec_h, n_ok); store4(oh + 2 * H, dgm, vec_h, n_ok);
            float* ge = gcur + (size_t)b * K16 + ub;
            store4(ge, dgr, false, n_ok); store4(ge + H, dgz, false, n_ok); store4(ge + 2 * H, dgm, false, n_ok);
        }
        publish(cnt);
        await(cnt, (unsigned)(step + 1) * d.n_slices, args.err);

        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        const float* grow = gcur + (size_t)(row_ok ? b : 0) * K16 + 4 * kq;
        const float* wl = Ws + kq * UU + j;
        for (int s = 0; s < K16; s += 32) {
            float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
            if (row_ok) {
                g0 = *reinterpret_cast<const float4*>(grow + s);
                if (s + 16 < K16) g1 = *reinterpret_cast<const float4*>(grow + s + 16);
            }
            const float a0[4] = {g0.x, g0.y, g0.z, g0.w}, a1[4] = {g1.x, g1.y, g1.z, g1.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(s + 4 * c) * UU], a0[c], acc0, 0, 0, 0);
                if (s + 16 < K16)
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wl[(s + 16 + 4 * c) * UU], a1[c], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) dh[c] = dhz[c] + acc0[c] + acc1[c];
    }
    if (row_ok && n_ok > 0 && in.dh0) store4(in.dh0 + (size_t)b * H + ub, dh, vec_h, n_ok);
}

struct Layout {
    size_t header_bytes, total_bytes;
    size_t cnt_off[DSSVAE_GRU_MAX_INST], buf_off[DSSVAE_GRU_MAX_INST];
};

inline int round16(int x) { return (x + 15) & ~15; }

// mult = row width multiplier of the exchange buffer: 1 (forward: H) or 3 (backward: 3H)
Layout plan(int n_inst, const int32_t* B, const int32_t* H, int mult_max) {
    Layout L;
    size_t off = 256;  // error word + padding
    for (int i = 0; i < n_inst; ++i) {
        L.cnt_off[i] = off;
        off += (size_t)((B[i] + BB - 1) / BB) * CNT_STRIDE * sizeof(unsigned);
    }
    off = (off + 255) & ~(size_t)255;
    L.header_bytes = off;
    for (int i = 0; i < n_inst; ++i) {
        L.buf_off[i] = off;
        off += (size_t)2 * B[i] * round16(mult_max * H[i]) * sizeof(float);
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

}  // namespace

extern "C" int64_t dssvae_gru_workspace_bytes(int n_inst, const int32_t* B_host, const int32_t* H_host) {
    if (n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !B_host || !H_host) return DSSVAE_EINVAL;
    return (int64_t)plan(n_inst, B_host, H_host, 3).total_bytes;
}

extern "C" int dssvae_gru_fwd(const dssvae_gru_fwd_inst* inst, int n_inst, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (!inst || n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !workspace) return DSSVAE_EINVAL;
    int32_t Bs[DSSVAE_GRU_MAX_INST], Hs[DSSVAE_GRU_MAX_INST];
    for (int i = 0; i < n_inst; ++i) {
        const dssvae_gru_fwd_inst& x = inst[i];
        if (!x.xproj || !x.whh || !x.bhh || !x.out || x.B <= 0 || x.T <= 0 || x.H <= 0 || x.ldx < 3 * x.H ||
            x.ldo < x.H)
            return DSSVAE_EINVAL;
        Bs[i] = x.B; Hs[i] = x.H;
    }
    const Layout L = plan(n_inst, Bs, Hs, 3);
    if ((int64_t)L.total_bytes > workspace_bytes || !aligned16(workspace)) return DSSVAE_EINVAL;
    FwdArgs args;
    args.n_inst = n_inst;
    args.err = reinterpret_cast<unsigned*>(workspace);
    int wg = 0, max_h16 = 0;
    for (int i = 0; i < n_inst; ++i) {
        FwdDev& d = args.inst[i];
        d.in = inst[i];
        d.H16 = round16(inst[i].H);
        d.n_slices = d.H16 / UU;
        d.n_bblocks = (inst[i].B + BB - 1) / BB;
        d.wg_begin = wg;
        wg += d.n_slices * d.n_bblocks;
        d.counters = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + L.cnt_off[i]);
        d.hx = reinterpret_cast<float*>(static_cast<char*>(workspace) + L.buf_off[i]);
        if (d.H16 > max_h16) max_h16 = d.H16;
    }
    const int lds_bytes = 3 * max_h16 * UU * (int)sizeof(float);
    if (wg > cu_count() || lds_bytes > 150 * 1024) return DSSVAE_ETOOBIG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, L.header_bytes, s);
    if (e != hipSuccess) return (int)e;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_fwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(gru_fwd_kernel, dim3(wg), dim3(NTHR), lds_bytes, s, args);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_bwd(const dssvae_gru_bwd_inst* inst, int n_inst, void* workspace, int64_t workspace_bytes,
                              void* stream) {
    if (!inst || n_inst <= 0 || n_inst > DSSVAE_GRU_MAX_INST || !workspace) return DSSVAE_EINVAL;
    int32_t Bs[DSSVAE_GRU_MAX_INST], Hs[DSSVAE_GRU_MAX_INST];
    for (int i = 0; i < n_inst; ++i) {
        const dssvae_gru_bwd_inst& x = inst[i];
        if (!x.whh || !x.gates || !x.dgx || !x.dgh || x.B <= 0 || x.T <= 0 || x.H <= 0 || (x.dout && x.ldd < x.H))
            return DSSVAE_EINVAL;
        Bs[i] = x.B; Hs[i] = x.H;
    }
    const Layout L = plan(n_inst, Bs, Hs, 3);
    if ((int64_t)L.total_bytes > workspace_bytes || !aligned16(workspace)) return DSSVAE_EINVAL;
    BwdArgs args;
    args.n_inst = n_inst;
    args.err = reinterpret_cast<unsigned*>(workspace);
    int wg = 0, max_k16 = 0;
    for (int i = 0; i < n_inst; ++i) {
        BwdDev& d = args.inst[i];
        d.in = inst[i];
        d.K16 = round16(3 * inst[i].H);
        d.n_slices = round16(inst[i].H) / UU;
        d.n_bblocks = (inst[i].B + BB - 1) / BB;
        d.wg_begin = wg;
        wg += d.n_slices * d.n_bblocks;
        d.counters = reinterpret_cast<unsigned*>(static_cast<char*>(workspace) + L.cnt_off[i]);
        d.gx = reinterpret_cast<float*>(static_cast<char*>(workspace) + L.buf_off[i]);
        if (d.K16 > max_k16) max_k16 = d.K16;
    }
    const int lds_bytes = max_k16 * UU * (int)sizeof(float);
    if (wg > cu_count() || lds_bytes > 150 * 1024) return DSSVAE_ETOOBIG;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(workspace, 0, L.header_bytes, s);
    if (e != hipSuccess) return (int)e;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gru_bwd_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(gru_bwd_kernel, dim3(wg), dim3(NTHR), lds_bytes, s, args);
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
