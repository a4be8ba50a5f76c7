// Greedy generation loop (decoder/rnn_decoder.py:160-168,193-199) as ONE native call: for max_len steps,
// embedding rows of the previous tokens -> per layer two small MFMA GEMMs (x W_ih^T + b_ih, h W_hh^T + b_hh)
// + the fused GRU cell -> output projection fused with a running arg-max over vocabulary splits (no
// softmax is needed for an arg-max, no [B,V] row is written) -> tokens[t].  The host loop only enqueues
// kernels on the caller's stream: no allocation, no synchronisation, no device->host traffic (the
// reference does a topk + .cpu() per step), so the whole decode is also capturable in a HIP graph.
#include "common.h"

namespace {
struct Ws {
    float* x;        // [B,E]   embedding rows of the current tokens
    float* gi;       // [B,3H]
    float* gh;       // [B,3H]
    float* pmax;     // [B,S]
    float* psum;     // [B,S]
    int64_t* parg;   // [B,S]
    int64_t* tok;    // [B]
    float* h;        // [2][L][B,H]  (ping-pong: the fused cell step writes h' beside h)
    size_t bytes;
};

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

Ws carve(void* base, int B, int E, int H, int L, int S) {
    Ws w;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += up256(n); return p ? p + o : nullptr; };
    w.x = reinterpret_cast<float*>(take((size_t)B * E * 4));
    w.gi = reinterpret_cast<float*>(take((size_t)B * 3 * H * 4));
    w.gh = reinterpret_cast<float*>(take((size_t)B * 3 * H * 4));
    w.pmax = reinterpret_cast<float*>(take((size_t)B * S * 4));
    w.psum = reinterpret_cast<float*>(take((size_t)B * S * 4));
    w.parg = reinterpret_cast<int64_t*>(take((size_t)B * S * 8));
    w.tok = reinterpret_cast<int64_t*>(take((size_t)B * 8));
    w.h = reinterpret_cast<float*>(take((size_t)2 * L * B * H * 4));
    w.bytes = off;
    return w;
}

__global__ void fill_i64_kernel(int64_t* p, int64_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// partial columns: enough for the general kernel's n_split and for the skinny kernel's own split count
// (at most one 16-column block per workgroup)
inline int part_cols(int n_split, int V) { return max(n_split, (V + 15) / 16); }

// [X | X2] W^T + b: the skinny kernel when the shape is in its range, else the tiled GEMM (X2 == NULL there)
int small_linear(const float* X, int K, const float* W, const float* b, int B, int N, float* C, void* stream) {
    int rc = dssvae_skinny_linear(X, K, K, nullptr, 0, 0, W, K, b, B, N, C, N, stream);
    if (rc == DSSVAE_ETOOBIG) rc = dssvae_gemm_f32(X, K, 0, nullptr, W, K, 0, C, N, b, B, N, K, 1.f, 0.f, stream);
    return rc;
}
}  // namespace

extern "C" int64_t dssvae_greedy_decode_workspace_bytes(int B, int E, int H, int n_layers, int n_split, int V) {
    if (B <= 0 || E <= 0 || H <= 0 || V <= 0 || n_layers <= 0 || n_layers > DSSVAE_DECODE_MAX_LAYERS || n_split <= 0)
        return DSSVAE_EINVAL;
    return (int64_t)carve(nullptr, B, E, H, n_layers, part_cols(n_split, V)).bytes;
}

extern "C" int dssvae_greedy_decode(const dssvae_decode_params* p, const float* h0, int B, int max_len,
                                    int64_t sos_id, int n_split, int64_t* tokens, float* h_final, void* workspace,
                                    int64_t workspace_bytes, void* stream) {
    if (!p || !h0 || !tokens || !workspace || B <= 0 || max_len <= 0 || n_split <= 0) return DSSVAE_EINVAL;
    const int E = p->E, H = p->H, L = p->n_layers, V = p->V;
    if (E <= 0 || H <= 0 || V <= 0 || L <= 0 || L > DSSVAE_DECODE_MAX_LAYERS || !p->embedding || !p->w_out ||
        p->ld_embedding < E || p->ld_out < H + E || sos_id < 0 || sos_id >= p->n_embeddings)
        return DSSVAE_EINVAL;
    for (int l = 0; l < L; ++l)
        if (!p->w_ih[l] || !p->w_hh[l] || !p->b_ih[l] || !p->b_hh[l]) return DSSVAE_EINVAL;
    if (!aligned16(workspace)) return DSSVAE_EINVAL;
    const Ws w = carve(workspace, B, E, H, L, part_cols(n_split, V));
    if ((int64_t)w.bytes > workspace_bytes) return DSSVAE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(w.h, h0, (size_t)L * B * H * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fill_i64_kernel, dim3((B + 255) / 256), dim3(256), 0, s, w.tok, sos_id, B);
    DSSVAE_CHECK_LAUNCH();
    int rc;
    int cur[DSSVAE_DECODE_MAX_LAYERS] = {0, 0, 0, 0};  // which half of the ping-pong holds layer l's state
#define TRY(call) do { rc = (call); if (rc != DSSVAE_OK) return rc; } while (0)
    for (int t = 0; t < max_len; ++t) {
        // the arg-max of step t-1 went straight into tokens[t-1]: it is also this step's input (no per-step
        // device-to-device copy: the runtime's blit path costs more than the step's kernels)
        const int64_t* prev = t == 0 ? w.tok : tokens + (size_t)(t - 1) * B;
        TRY(dssvae_embed_gather(p->embedding, p->ld_embedding, prev, B, E, nullptr, w.x, E, stream));
        const float* in = w.x;
        int in_dim = E;
        for (int l = 0; l < L; ++l) {
            float* hl = w.h + ((size_t)cur[l] * L + l) * B * H;
            float* hn = w.h + ((size_t)(cur[l] ^ 1) * L + l) * B * H;
            rc = dssvae_skinny_gru_step(in, in_dim, in_dim, hl, H, H, p->w_ih[l], in_dim, p->w_hh[l], H, p->b_ih[l],
                                        p->b_hh[l], B, hn, H, stream);
            if (rc == DSSVAE_OK) {
                cur[l] ^= 1;
                hl = hn;
            } else if (rc == DSSVAE_ETOOBIG) {  // outside the fused kernel's range: two products + the cell
                TRY(small_linear(in, in_dim, p->w_ih[l], p->b_ih[l], B, 3 * H, w.gi, stream));
                TRY(small_linear(hl, H, p->w_hh[l], p->b_hh[l], B, 3 * H, w.gh, stream));
                TRY(dssvae_gru_cell(w.gi, w.gh, hl, B, H, hl, stream));  // element-wise: in place is safe
            } else {
                return rc;
            }
            in = hl;
            in_dim = H;
        }
        int S = dssvae_skinny_argmax_splits(B, V);
        rc = dssvae_skinny_argmax(in, H, H, w.x, E, E, p->w_out, p->ld_out, p->b_out, B, V, w.pmax, w.psum, w.parg,
                                  stream);
        if (rc == DSSVAE_ETOOBIG) {
            S = n_split;
            rc = dssvae_linear_lse_fwd(in, H, H, w.x, E, E, p->w_out, p->ld_out, p->b_out, B, V, n_split, w.pmax,
                                       w.psum, w.parg, 1, nullptr, 0, stream);
        }
        if (rc != DSSVAE_OK) return rc;
        TRY(dssvae_lse_combine(w.pmax, w.psum, w.parg, B, S, nullptr, tokens + (size_t)t * B, stream));
    }
#undef TRY
    if (h_final) {
        for (int l = 0; l < L; ++l) {
            e = hipMemcpyAsync(h_final + (size_t)l * B * H, w.h + ((size_t)cur[l] * L + l) * B * H, (size_t)B * H * 4,
                               hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return (int)e;
        }
    }
    return DSSVAE_OK;
}
