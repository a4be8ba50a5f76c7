// HBM-bound pointwise / reduction kernels (roofline: HBM3E ~8 TB/s spec, ~6.3 TB/s achievable):
// embedding rows, dropout masks, word dropout, reparameterisation + KL (wavefront reduction),
// GRU cell for the greedy loop, grad-norm and Adam over flat buffers.
#include "common.h"

namespace {

__global__ void embed_gather_kernel(const float* __restrict__ table, int64_t ldt, const int64_t* __restrict__ ids,
                                    int M, int E, const float* __restrict__ mask, float* __restrict__ out,
                                    int64_t ldo) {
    // one wave per row: 64 lanes x float4 cover 256 floats per trip, coalesced
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    const float* src = table + ids[row] * ldt;
    float* dst = out + (int64_t)row * ldo;
    const float* mk = mask ? mask + (int64_t)row * E : nullptr;
    for (int e = lane; e < E; e += 64) dst[e] = mk ? src[e] * mk[e] : src[e];
}

__global__ void embed_scatter_kernel(float* __restrict__ dtable, int64_t ldt, const int64_t* __restrict__ ids, int M,
                                     int E, const float* __restrict__ mask, const float* __restrict__ g,
                                     int64_t ldg) {
    // one wave per row, 256 contiguous bytes per atomic wave-instruction (full atomic rate shape)
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    float* dst = dtable + ids[row] * ldt;
    const float* src = g + (int64_t)row * ldg;
    const float* mk = mask ? mask + (int64_t)row * E : nullptr;
    for (int e = lane; e < E; e += 64) atomicAdd(dst + e, mk ? src[e] * mk[e] : src[e]);
}

__global__ void dropout_mask_kernel(float* __restrict__ mask, int64_t n, float p, float scale, uint64_t seed,
                                    const uint64_t* __restrict__ seed_base) {
    if (seed_base) seed += seed_base[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mask[i] = u01(seed, (uint64_t)i) < p ? 0.f : scale;
}

// y = x * keep-mask(seed) / (1 - p): the mask is a function of (seed, element index), so backward regenerates
// it from the seed (dx = dy * mask) and no mask tensor ever exists
__global__ void dropout_apply_kernel(const float* __restrict__ x, int64_t n, float p, float scale, uint64_t seed,
                                     const uint64_t* __restrict__ seed_base, float* __restrict__ y) {
    if (seed_base) seed += seed_base[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = u01(seed, (uint64_t)i) < p ? 0.f : x[i] * scale;
}

__global__ void unk_replace_kernel(const int64_t* __restrict__ ids, int64_t n, const float* __restrict__ rnd,
                                   float rate, const float* __restrict__ rate_dev, uint64_t seed,
                                   const uint64_t* __restrict__ seed_base, int64_t sos, int64_t pad, int64_t eos,
                                   int64_t unk, int64_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (seed_base) seed += seed_base[0];
    if (rate_dev) rate = rate_dev[0];
    const int64_t v = ids[i];
    const float r = rnd ? rnd[i] : u01(seed, (uint64_t)i);
    const bool special = v == sos || v == pad || v == eos;
    out[i] = (!special && r < rate) ? unk : v;
}

__global__ void reparam_kl_fwd_kernel(const float* __restrict__ mean, const float* __restrict__ logv,
                                      const float* __restrict__ eps, int64_t n, float* __restrict__ z,
                                      float* __restrict__ kl) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mean[i], lv = logv[i];
        if (z) z[i] = eps ? eps[i] * __expf(0.5f * lv) + m : m;
        acc += 1.f + lv - m * m - __expf(lv);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0 && kl) atomicAdd(kl, -0.5f * acc);
}

__global__ void reparam_kl_bwd_kernel(const float* __restrict__ mean, const float* __restrict__ logv,
                                      const float* __restrict__ eps, const float* __restrict__ dz,
                                      const float* __restrict__ dkl, int64_t n, float* __restrict__ dmean,
                                      float* __restrict__ dlogv) {
    const float gk = dkl ? dkl[0] : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mean[i], lv = logv[i];
        const float g = dz ? dz[i] : 0.f;
        dmean[i] = g + gk * m;
        dlogv[i] = (eps ? g * eps[i] * 0.5f * __expf(0.5f * lv) : 0.f) + gk * 0.5f * (__expf(lv) - 1.f);
    }
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ void gru_cell_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                const float* __restrict__ h, int B, int H, float* __restrict__ hn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H) return;
    const int64_t b = i / H, u = i % H;
    const float* a = gi + b * 3 * H;
    const float* c = gh + b * 3 * H;
    const float r = sigmoidf_(a[u] + c[u]);
    const float z = sigmoidf_(a[H + u] + c[H + u]);
    const float n = tanhf(a[2 * H + u] + r * c[2 * H + u]);
    hn[i] = (1.f - z) * n + z * h[i];
}

// One timestep of a GRU whose recurrent product gh = h_{t-1} W_hh^T was made by the GEMM (the stepwise
// path for instances too large for the persistent kernel).  Same saved history layout as gru_fwd_kernel:
// gates row = [r | z | n | W_hn h + b_hn | h_{t-1}].  Rows with t >= len keep their state, output 0.
__global__ void gru_step_fwd_kernel(const float* __restrict__ x, int64_t x_rs, const float* __restrict__ gh,
                                    const float* __restrict__ bhh, const float* __restrict__ h,
                                    const int32_t* __restrict__ lengths, int t, int B, int H,
                                    float* __restrict__ hn, float* __restrict__ out, int64_t out_rs,
                                    float* __restrict__ gates, int64_t g_rs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H) return;
    const int64_t b = i / H, u = i % H;
    const float* a = x + b * x_rs;
    const float* c = gh + b * 3 * H;
    const float hp = h ? h[i] : 0.f;
    const float r = sigmoidf_(a[u] + c[u] + bhh[u]);
    const float z = sigmoidf_(a[H + u] + c[H + u] + bhh[H + u]);
    const float m = c[2 * H + u] + bhh[2 * H + u];
    const float n = tanhf(a[2 * H + u] + r * m);
    const float hv = (1.f - z) * n + z * hp;
    const bool live = !lengths || t < lengths[b];
    hn[i] = live ? hv : hp;
    if (out) out[b * out_rs + u] = live ? hv : 0.f;
    if (gates) {
        float* g = gates + b * g_rs;
        g[u] = r; g[H + u] = z; g[2 * H + u] = n; g[3 * H + u] = m; g[4 * H + u] = hp;
    }
}

// Backward of one timestep: dh (gradient wrt h_t from the future) + dout_t -> dgx_t, dgh_t [3H] and
// dhz = the part of dh_{t-1} that does not go through W_hh (the caller adds dgh_t . W_hh by GEMM).
__global__ void gru_step_bwd_kernel(const float* __restrict__ gates, int64_t g_rs, const float* __restrict__ dout,
                                    int64_t d_rs, const float* __restrict__ dh, const int32_t* __restrict__ lengths,
                                    int t, int B, int H, float* __restrict__ dgx, float* __restrict__ dgh,
                                    int64_t o_rs, float* __restrict__ dhz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * H) return;
    const int64_t b = i / H, u = i % H;
    const float gin = dh ? dh[i] : 0.f;
    float* ox = dgx + b * o_rs;
    float* oh = dgh + b * o_rs;
    const bool live = !lengths || t < lengths[b];
    if (!live) {
        ox[u] = ox[H + u] = ox[2 * H + u] = 0.f;
        oh[u] = oh[H + u] = oh[2 * H + u] = 0.f;
        dhz[i] = gin;
        return;
    }
    const float* gs = gates + b * g_rs;
    const float r = gs[u], z = gs[H + u], n = gs[2 * H + u], m = gs[3 * H + u], hp = gs[4 * H + u];
    const float g = gin + (dout ? dout[b * d_rs + u] : 0.f);
    const float dn = g * (1.f - z) * (1.f - n * n);
    const float dz = g * (hp - n) * z * (1.f - z);
    const float dr = dn * m * r * (1.f - r);
    ox[u] = dr; ox[H + u] = dz; ox[2 * H + u] = dn;
    oh[u] = dr; oh[H + u] = dz; oh[2 * H + u] = dn * r;
    dhz[i] = g * z;
}

// ---- LSTM (nn.LSTM gate order i, f, g, o), stepwise: the recurrent product gh = h_{t-1} W_hh^T comes from the GEMM,
// these fuse the cell of ONE timestep.  Saved history row (for backward): [i | f | g | o | c_{t-1} | h_{t-1} | tanh c_t].
__global__ void lstm_step_fwd_kernel(const float* __restrict__ x, int64_t x_rs, const float* __restrict__ gh,
                                     const float* __restrict__ bhh, const float* __restrict__ h,
                                     const float* __restrict__ c, const int32_t* __restrict__ lengths, int t, int B,
                                     int H, float* __restrict__ hn, float* __restrict__ cn, float* __restrict__ out,
                                     int64_t out_rs, float* __restrict__ gates, int64_t g_rs) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * H) return;
    const int64_t b = idx / H, u = idx % H;
    const float* a = x + b * x_rs;
    const float* r = gh + b * 4 * H;
    const float hp = h ? h[idx] : 0.f, cp = c ? c[idx] : 0.f;
    const float gi = sigmoidf_(a[u] + r[u] + bhh[u]);
    const float gf = sigmoidf_(a[H + u] + r[H + u] + bhh[H + u]);
    const float gg = tanhf(a[2 * H + u] + r[2 * H + u] + bhh[2 * H + u]);
    const float go = sigmoidf_(a[3 * H + u] + r[3 * H + u] + bhh[3 * H + u]);
    const float cv = gf * cp + gi * gg;
    const float tc = tanhf(cv);
    const float hv = go * tc;
    const bool live = !lengths || t < lengths[b];
    hn[idx] = live ? hv : hp;
    cn[idx] = live ? cv : cp;
    if (out) out[b * out_rs + u] = live ? hv : 0.f;
    if (gates) {
        float* g = gates + b * g_rs;
        g[u] = gi; g[H + u] = gf; g[2 * H + u] = gg; g[3 * H + u] = go; g[4 * H + u] = cp; g[5 * H + u] = hp;
        g[6 * H + u] = tc;
    }
}

// Backward of one timestep: dh, dc (gradients wrt h_t, c_t from the future) + dout_t -> dg_t [4H] (the gradient wrt the
// gate pre-activations: the same for the input and the recurrent product), dhz = 0-part of dh_{t-1} that bypasses W_hh
// (only for rows past their length), dc_{t-1}.  The caller adds dg_t W_hh to dhz.
__global__ void lstm_step_bwd_kernel(const float* __restrict__ gates, int64_t g_rs, const float* __restrict__ dout,
                                     int64_t d_rs, const float* __restrict__ dh, const float* __restrict__ dc,
                                     const int32_t* __restrict__ lengths, int t, int B, int H,
                                     float* __restrict__ dg, int64_t o_rs, float* __restrict__ dhz,
                                     float* __restrict__ dcp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)B * H) return;
    const int64_t b = idx / H, u = idx % H;
    const float gin = dh ? dh[idx] : 0.f, cin = dc ? dc[idx] : 0.f;
    float* o = dg + b * o_rs;
    const bool live = !lengths || t < lengths[b];
    if (!live) {
        o[u] = o[H + u] = o[2 * H + u] = o[3 * H + u] = 0.f;
        dhz[idx] = gin;
        dcp[idx] = cin;
        return;
    }
    const float* gs = gates + b * g_rs;
    const float gi = gs[u], gf = gs[H + u], gg = gs[2 * H + u], go = gs[3 * H + u], cp = gs[4 * H + u], tc = gs[6 * H + u];
    const float g = gin + (dout ? dout[b * d_rs + u] : 0.f);
    const float dct = cin + g * go * (1.f - tc * tc);
    o[u] = dct * gg * gi * (1.f - gi);
    o[H + u] = dct * cp * gf * (1.f - gf);
    o[2 * H + u] = dct * gi * (1.f - gg * gg);
    o[3 * H + u] = g * tc * go * (1.f - go);
    dhz[idx] = 0.f;
    dcp[idx] = dct * gf;
}

// Sum of squares in a FIXED order (block partials -> one block adds them up): data-parallel replicas must
// compute bit-identical clip coefficients from bit-identical averaged gradients, which float atomics in
// arrival order do not give.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                    float* __restrict__ partials) {
    __shared__ float red[4];
    float acc = 0.f;
    const int64_t n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = g[(n4 << 2) + threadIdx.x]; acc += v * v; }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int n_partials,
                                                          float* __restrict__ out) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 256) acc += partials[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] += (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, const float* __restrict__ sumsq, float max_norm,
                            const float* __restrict__ live, const int32_t* __restrict__ skip,
                            const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }  // per-step values of a replayed graph
    // a persistent launch of this step timed out waiting for a peer (its results are garbage): the whole
    // update is dropped on device, before the host gets to read the error word
    if (skip && skip[0] != 0) return;
    float clip = 1.f;
    if (sumsq && max_norm > 0.f) clip = fminf(1.f, max_norm / (sqrtf(sumsq[0]) + 1e-6f));
    const float step = lr / bc1;
    int64_t done = 0;
    // 16-byte granules when the four arrays allow it (the flat buffers' slots are 16-byte aligned): same arithmetic per
    // element, a quarter of the memory instructions
    if (!live && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15u) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 g4 = reinterpret_cast<const float4*>(g)[i];
            float4 m4 = reinterpret_cast<float4*>(m)[i], v4 = reinterpret_cast<float4*>(v)[i], p4 = reinterpret_cast<float4*>(p)[i];
            float* gp = (float*)&g4; float* mp = (float*)&m4; float* vp = (float*)&v4; float* pp = (float*)&p4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float gi = gp[c] * clip;
                const float mi = b1 * mp[c] + (1.f - b1) * gi;
                const float vi = b2 * vp[c] + (1.f - b2) * gi * gi;
                mp[c] = mi;
                vp[c] = vi;
                pp[c] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
            }
            reinterpret_cast<float4*>(m)[i] = m4;
            reinterpret_cast<float4*>(v)[i] = v4;
            reinterpret_cast<float4*>(p)[i] = p4;
        }
        done = n4 << 2;
    }
    for (int64_t i = done + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (live && live[i] == 0.f) continue;  // parameter had no gradient this step: Adam skips it
        const float gi = g[i] * clip;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

inline unsigned grid_for(int64_t n, int block = 256, int cap = 2048) {
    int64_t g = (n + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

// Scalar loss assembly (vae/enhance_syntax_vae.py:212-340 in the reference: ~40 one-element torch ops forward and
// ~25 backward): out[j] = sum_k coef[j][k] * sum(in[k]).  One workgroup: wave w sums input w, w + 16, ... in a fixed
// order (lane-strided partials, then the wave's shuffle tree), the sums meet in LDS, thread j forms out[j].
struct WSums {
    const float* in[DSSVAE_WSUMS_MAX];
    float* din[DSSVAE_WSUMS_MAX];
    const float* dout[DSSVAE_WSUMS_MAX];
    int len[DSSVAE_WSUMS_MAX];
    float coef[DSSVAE_WSUMS_MAX][DSSVAE_WSUMS_MAX];   // [out][in]
    float coefd[DSSVAE_WSUMS_MAX][DSSVAE_WSUMS_MAX];  // [out][in]: the part multiplied by *dyn (the KL weight of the step)
    const float* dyn;
    int n_in, n_out;
};

__global__ __launch_bounds__(1024) void weighted_sums_fwd_kernel(WSums a, float* __restrict__ out) {
    __shared__ float comp[DSSVAE_WSUMS_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = wave; k < a.n_in; k += 16) {
        float s = 0.f;
        for (int i = lane; i < a.len[k]; i += 64) s += a.in[k][i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) comp[k] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < a.n_out) {
        float v = 0.f;
        const float d = a.dyn ? a.dyn[0] : 0.f;
        for (int k = 0; k < a.n_in; ++k) v += (a.coef[threadIdx.x][k] + a.coefd[threadIdx.x][k] * d) * comp[k];
        out[threadIdx.x] = v;
    }
}

// backward: din[k][i] = sum_j *dout[j] * coef[j][k] for every element of every input that wants a gradient (an output
// without a gradient has dout[j] = NULL)
__global__ __launch_bounds__(256) void weighted_sums_bwd_kernel(WSums a) {
    const int k = blockIdx.y;
    if (!a.din[k]) return;
    float g = 0.f;
    const float d = a.dyn ? a.dyn[0] : 0.f;
    for (int j = 0; j < a.n_out; ++j)
        if (a.dout[j]) g += a.dout[j][0] * (a.coef[j][k] + a.coefd[j][k] * d);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < a.len[k]; i += gridDim.x * 256) a.din[k][i] = g;
}

extern "C" int dssvae_abi_version(void) { return DSSVAE_ABI_VERSION; }

extern "C" int dssvae_device_cus(int* n) {
    if (!n) return DSSVAE_EINVAL;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return (int)e;
    *n = prop.multiProcessorCount;
    return DSSVAE_OK;
}

extern "C" int dssvae_embed_gather(const float* table, int64_t ldt, const int64_t* ids, int M, int E,
                                   const float* mask, float* out, int64_t ldo, void* stream) {
    if (!table || !ids || !out || M <= 0 || E <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, table, ldt, ids, M,
                       E, mask, out, ldo);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_embed_scatter_add(float* dtable, int64_t ldt, const int64_t* ids, int M, int E,
                                        const float* mask, const float* g, int64_t ldg, void* stream) {
    if (!dtable || !ids || !g || M <= 0 || E <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(embed_scatter_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, dtable, ldt, ids,
                       M, E, mask, g, ldg);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_dropout_mask_dyn(float* mask, int64_t n, float p, uint64_t seed, const uint64_t* seed_base,
                                       void* stream) {
    if (!mask || n <= 0 || p < 0.f || p >= 1.f) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, p,
                       1.f / (1.f - p), seed, seed_base);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, void* stream) {
    return dssvae_dropout_mask_dyn(mask, n, p, seed, nullptr, stream);
}

extern "C" int dssvae_dropout_apply_dyn(const float* x, int64_t n, float p, uint64_t seed, const uint64_t* seed_base,
                                        float* y, void* stream) {
    if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3(grid_for(n, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, n, p,
                       1.f / (1.f - p), seed, seed_base, y);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_dropout_apply(const float* x, int64_t n, float p, uint64_t seed, float* y, void* stream) {
    return dssvae_dropout_apply_dyn(x, n, p, seed, nullptr, y, stream);
}

extern "C" int dssvae_unk_replace_dyn(const int64_t* ids, int64_t n, const float* rnd, float rate,
                                      const float* rate_dev, uint64_t seed, const uint64_t* seed_base, int64_t sos,
                                      int64_t pad, int64_t eos, int64_t unk, int64_t* out, void* stream) {
    if (!ids || !out || n <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(unk_replace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids,
                       n, rnd, rate, rate_dev, seed, seed_base, sos, pad, eos, unk, out);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_unk_replace(const int64_t* ids, int64_t n, const float* rnd, float rate, uint64_t seed,
                                  int64_t sos, int64_t pad, int64_t eos, int64_t unk, int64_t* out, void* stream) {
    return dssvae_unk_replace_dyn(ids, n, rnd, rate, nullptr, seed, nullptr, sos, pad, eos, unk, out, stream);
}

extern "C" int dssvae_reparam_kl_fwd(const float* mean, const float* logv, const float* eps, int64_t n, float* z,
                                     float* kl_out, void* stream) {
    if (!mean || !logv || n <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(grid_for(n, 256, 256)), dim3(256), 0, (hipStream_t)stream, mean,
                       logv, eps, n, z, kl_out);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_reparam_kl_bwd(const float* mean, const float* logv, const float* eps, const float* dz,
                                     const float* dkl, int64_t n, float* dmean, float* dlogv, void* stream) {
    if (!mean || !logv || !dmean || !dlogv || n <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(grid_for(n, 256, 256)), dim3(256), 0, (hipStream_t)stream, mean,
                       logv, eps, dz, dkl, n, dmean, dlogv);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_cell(const float* gi, const float* gh, const float* h, int B, int H, float* h_new,
                               void* stream) {
    if (!gi || !gh || !h || !h_new || B <= 0 || H <= 0) return DSSVAE_EINVAL;
    const int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(gru_cell_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gi, gh,
                       h, B, H, h_new);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_step_fwd(const float* x_t, int64_t x_row_stride, const float* gh, const float* bhh,
                                   const float* h_prev, const int32_t* lengths, int t, int B, int H, float* h_new,
                                   float* out_t, int64_t out_row_stride, float* gates_t, int64_t gates_row_stride,
                                   void* stream) {
    if (!x_t || !gh || !bhh || !h_new || B <= 0 || H <= 0 || x_row_stride < 3 * H) return DSSVAE_EINVAL;
    if ((out_t && out_row_stride < H) || (gates_t && gates_row_stride < 5 * H)) return DSSVAE_EINVAL;
    const int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_t,
                       x_row_stride, gh, bhh, h_prev, lengths, t, B, H, h_new, out_t, out_row_stride, gates_t,
                       gates_row_stride);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_gru_step_bwd(const float* gates_t, int64_t gates_row_stride, const float* dout_t,
                                   int64_t dout_row_stride, const float* dh, const int32_t* lengths, int t, int B,
                                   int H, float* dgx_t, float* dgh_t, int64_t dg_row_stride, float* dhz, void* stream) {
    if (!gates_t || !dgx_t || !dgh_t || !dhz || B <= 0 || H <= 0 || gates_row_stride < 5 * H ||
        dg_row_stride < 3 * H || (dout_t && dout_row_stride < H))
        return DSSVAE_EINVAL;
    const int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(gru_step_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gates_t, gates_row_stride, dout_t, dout_row_stride, dh, lengths, t, B, H, dgx_t, dgh_t,
                       dg_row_stride, dhz);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_lstm_step_fwd(const float* x_t, int64_t x_row_stride, const float* gh, const float* bhh,
                                    const float* h_prev, const float* c_prev, const int32_t* lengths, int t, int B,
                                    int H, float* h_new, float* c_new, float* out_t, int64_t out_row_stride,
                                    float* gates_t, int64_t gates_row_stride, void* stream) {
    if (!x_t || !gh || !bhh || !h_new || !c_new || B <= 0 || H <= 0 || x_row_stride < 4 * H) return DSSVAE_EINVAL;
    if ((out_t && out_row_stride < H) || (gates_t && gates_row_stride < 7 * H)) return DSSVAE_EINVAL;
    const int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_t,
                       x_row_stride, gh, bhh, h_prev, c_prev, lengths, t, B, H, h_new, c_new, out_t, out_row_stride,
                       gates_t, gates_row_stride);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_lstm_step_bwd(const float* gates_t, int64_t gates_row_stride, const float* dout_t,
                                    int64_t dout_row_stride, const float* dh, const float* dc, const int32_t* lengths,
                                    int t, int B, int H, float* dg_t, int64_t dg_row_stride, float* dhz, float* dc_prev,
                                    void* stream) {
    if (!gates_t || !dg_t || !dhz || !dc_prev || B <= 0 || H <= 0 || gates_row_stride < 7 * H ||
        dg_row_stride < 4 * H || (dout_t && dout_row_stride < H))
        return DSSVAE_EINVAL;
    const int64_t n = (int64_t)B * H;
    hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gates_t, gates_row_stride, dout_t, dout_row_stride, dh, dc, lengths, t, B, H, dg_t,
                       dg_row_stride, dhz, dc_prev);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_sumsq(const float* g, int64_t n, float* partials, float* sumsq_out, void* stream) {
    if (!g || !partials || !sumsq_out || n <= 0 || !aligned16(g)) return DSSVAE_EINVAL;
    const unsigned nb = grid_for(n / 4 + 1, 256, DSSVAE_SUMSQ_PARTIALS);
    hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, g, n, partials);
    DSSVAE_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)nb, sumsq_out);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, int step, const float* sumsq, float max_norm,
                                const float* grad_scale_mask, const int32_t* skip_flag, void* stream) {
    if (!p || !g || !m || !v || n <= 0 || step < 1) return DSSVAE_EINVAL;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1,
                       beta2, eps, bc1, bc2_sqrt, sumsq, max_norm, grad_scale_mask, skip_flag, (const float*)nullptr);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_adam_step_dyn(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper,
                                    float beta1, float beta2, float eps, const float* sumsq, float max_norm,
                                    const int32_t* skip_flag, void* stream) {
    if (!p || !g || !m || !v || !hyper || n <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, 0.f, beta1,
                       beta2, eps, 1.f, 1.f, sumsq, max_norm, (const float*)nullptr, skip_flag, hyper);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

static int wsums_pack(WSums* a, const float* const* in, float* const* din, const int32_t* len, int n_in,
                      const float* coef, int n_out, const float* coef_dyn = nullptr, const float* dyn = nullptr) {
    if (!len || !coef || n_in <= 0 || n_out <= 0) return DSSVAE_EINVAL;
    if (n_in > DSSVAE_WSUMS_MAX || n_out > DSSVAE_WSUMS_MAX) return DSSVAE_ETOOBIG;
    a->n_in = n_in;
    a->n_out = n_out;
    for (int k = 0; k < n_in; ++k) {
        if (len[k] < 0 || (in && len[k] > 0 && !in[k])) return DSSVAE_EINVAL;
        a->in[k] = in ? in[k] : nullptr;
        a->din[k] = din ? din[k] : nullptr;
        a->len[k] = len[k];
    }
    for (int j = 0; j < n_out; ++j)
        for (int k = 0; k < n_in; ++k) {
            a->coef[j][k] = coef[j * n_in + k];
            a->coefd[j][k] = (coef_dyn && dyn) ? coef_dyn[j * n_in + k] : 0.f;
        }
    a->dyn = (coef_dyn && dyn) ? dyn : nullptr;
    return DSSVAE_OK;
}

extern "C" int dssvae_weighted_sums_fwd(const float* const* in, const int32_t* len, int n_in, const float* coef,
                                        int n_out, float* out, void* stream) {
    return dssvae_weighted_sums_fwd_dyn(in, len, n_in, coef, nullptr, nullptr, n_out, out, stream);
}

extern "C" int dssvae_weighted_sums_fwd_dyn(const float* const* in, const int32_t* len, int n_in, const float* coef,
                                            const float* coef_dyn, const float* dyn, int n_out, float* out,
                                            void* stream) {
    if (!in || !out) return DSSVAE_EINVAL;
    WSums a = {};
    const int e = wsums_pack(&a, in, nullptr, len, n_in, coef, n_out, coef_dyn, dyn);
    if (e != DSSVAE_OK) return e;
    hipLaunchKernelGGL(weighted_sums_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, out);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_weighted_sums_bwd(float* const* din, const int32_t* len, int n_in, const float* coef, int n_out,
                                        const float* const* dout, void* stream) {
    return dssvae_weighted_sums_bwd_dyn(din, len, n_in, coef, nullptr, nullptr, n_out, dout, stream);
}

extern "C" int dssvae_weighted_sums_bwd_dyn(float* const* din, const int32_t* len, int n_in, const float* coef,
                                            const float* coef_dyn, const float* dyn, int n_out,
                                            const float* const* dout, void* stream) {
    if (!din || !dout) return DSSVAE_EINVAL;
    WSums a = {};
    const int e = wsums_pack(&a, nullptr, din, len, n_in, coef, n_out, coef_dyn, dyn);
    if (e != DSSVAE_OK) return e;
    for (int j = 0; j < n_out; ++j) a.dout[j] = dout[j];
    int longest = 0;
    for (int k = 0; k < n_in; ++k)
        if (din[k]) longest = len[k] > longest ? len[k] : longest;
    if (longest == 0) return DSSVAE_OK;
    const int bx = (longest + 255) / 256;
    hipLaunchKernelGGL(weighted_sums_bwd_kernel, dim3(bx > 64 ? 64 : bx, n_in), dim3(256), 0, (hipStream_t)stream, a);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Per-step scalars of a replayed HIP graph, and minibatches cut from the device-resident id cache
// ---------------------------------------------------------------------------------------------------------------
namespace {

struct StepWords { uint32_t w[DSSVAE_STEP_WORDS]; };

__global__ void set_words_kernel(uint32_t* __restrict__ dst, StepWords v, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = v.w[threadIdx.x];
}

// one wave per output row: row b = [sos] ids[off[e] .. off[e+1]) [eos] pad.., cut at W
__global__ void batch_rows_kernel(const int64_t* __restrict__ flat, const int64_t* __restrict__ off,
                                  const int64_t* __restrict__ idx, int B, int W, int boundary, int64_t sos,
                                  int64_t eos, int64_t pad, int64_t* __restrict__ out, int32_t* __restrict__ lens) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    const int64_t e = idx[b], lo = off[e];
    const int64_t L = off[e + 1] - lo;
    if (lane == 0 && lens) lens[b] = (int32_t)L;
    int64_t* row = out + (int64_t)b * W;
    for (int t = lane; t < W; t += 64) {
        const int64_t k = boundary ? t - 1 : t;   // position within the example
        int64_t v = pad;
        if (boundary && t == 0) v = sos;
        else if (k < L) v = flat[lo + k];
        else if (boundary && k == L) v = eos;
        row[t] = v;
    }
}

// keep[] = the positions b*T + t, t < min(len[b] + 1, T), in order; one workgroup scans the B lengths in LDS
__global__ __launch_bounds__(1024) void keep_rows_kernel(const int32_t* __restrict__ lens, int B, int T,
                                                         int64_t* __restrict__ keep, int n_keep) {
    __shared__ int start[8192 + 1];
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int b = base + threadIdx.x;
        const int c = b < B ? min(lens[b] + 1, T) : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int before = carry_s;
        for (int w = 0; w < wave; ++w) before += wave_tot[w];
        if (b < B) start[b] = before + incl - c;
        __syncthreads();
        if (threadIdx.x == 0) {
            int tot = carry_s;
            for (int w = 0; w < 16; ++w) tot += wave_tot[w];
            carry_s = tot;
        }
        __syncthreads();
    }
    const int total = carry_s;
    if (threadIdx.x == 0) start[B] = total;
    __syncthreads();
    // one wave per sentence writes its run of positions
    for (int b = wave; b < B; b += 16) {
        const int s0 = start[b], c = start[b + 1] - s0;
        for (int t = lane; t < c; t += 64)
            if (s0 + t < n_keep) keep[s0 + t] = (int64_t)b * T + t;
    }
    // the caller may have asked for more positions than there are (a padded list): repeat the last real one
    if (total > 0) {
        int lb = B - 1;
        while (lb > 0 && start[lb + 1] - start[lb] == 0) --lb;
        const int64_t last = (int64_t)lb * T + (start[lb + 1] - start[lb] - 1);
        for (int i = total + threadIdx.x; i < n_keep; i += 1024) keep[i] = last;
    }
}

}  // namespace

extern "C" int dssvae_set_words(uint32_t* dst, const uint32_t* words_host, int n, void* stream) {
    if (!dst || !words_host || n <= 0 || n > DSSVAE_STEP_WORDS) return DSSVAE_EINVAL;
    StepWords v = {};
    for (int i = 0; i < n; ++i) v.w[i] = words_host[i];
    hipLaunchKernelGGL(set_words_kernel, dim3(1), dim3(DSSVAE_STEP_WORDS), 0, (hipStream_t)stream, dst, v, n);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_batch_rows(const int64_t* flat_ids, const int64_t* offsets, const int64_t* idx, int B, int W,
                                 int boundary, int64_t sos, int64_t eos, int64_t pad, int64_t* out,
                                 int32_t* lengths_out, void* stream) {
    if (!flat_ids || !offsets || !idx || !out || B <= 0 || W <= 0) return DSSVAE_EINVAL;
    hipLaunchKernelGGL(batch_rows_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, flat_ids, offsets, idx,
                       B, W, boundary, sos, eos, pad, out, lengths_out);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}

extern "C" int dssvae_keep_rows(const int32_t* lengths, int B, int T, int64_t* keep, int n_keep, void* stream) {
    if (!lengths || !keep || B <= 0 || T <= 0 || n_keep <= 0) return DSSVAE_EINVAL;
    if (B > 8192) return DSSVAE_ETOOBIG;
    hipLaunchKernelGGL(keep_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, lengths, B, T, keep, n_keep);
    DSSVAE_CHECK_LAUNCH();
    return DSSVAE_OK;
}
