// Greedy generation loop (decoder/rnn_decoder.py:160-168,193-199) as ONE native call: for max_len steps,
// embedding rows of the previous tokens -> per layer two small MFMA GEMMs (x W_ih^T + b_ih, h W_hh^T + b_hh)
// + the GRU cell (one fused launch per layer when the shape allows) -> output projection fused with a running
// arg-max over vocabulary splits (no softmax is needed for an arg-max, no [B,V] row is written) -> tokens[t]
// and the next step's embedding rows.  The host loop only enqueues
// kernels on the caller's stream: no allocation, no synchronisation, no device->host traffic (the
// reference does a topk + .cpu() per step), so the whole decode is also capturable in a HIP graph.
#include "common.h"
#include <mutex>
#include <vector>

// Scratch of the no-allocation entry point (dssvae_greedy_decode_noalloc): the loop's whole workspace lives in
// this file-scope device array, one copy per GPU, held until the process ends.  In host code the array's NAME is
// a host-side stand-in, not a device address: the address is taken once per device with hipGetSymbolAddress and
// only that pointer is ever handed to kernels / copies.
static __device__ __align__(256) unsigned char scratch[DSSVAE_DECODE_NOALLOC_BYTES];

namespace {
constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];
std::mutex g_scratch_mu;

// device address of `scratch` on the current device (resolved on the first call per device, with that device
// current); returns a hipError_t / DSSVAE_EINVAL on failure
int scratch_address(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return DSSVAE_EINVAL;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!ptr) return DSSVAE_EINVAL;
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return DSSVAE_OK;
}

// Optional live timing of the loop's dominant launch (projection + arg-max over the vocabulary): while armed,
// every such launch is bracketed by two HIP events on the launch stream (bench.py reads the sum afterwards).
struct ArgmaxProfile {
    std::vector<hipEvent_t> ev;  // pairs
    size_t used = 0;
    bool on = false;
} g_prof;

inline void prof_mark(hipStream_t s) {
    if (!g_prof.on) return;
    if (g_prof.used == g_prof.ev.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { g_prof.on = false; return; }
        g_prof.ev.push_back(e);
    }
    (void)hipEventRecord(g_prof.ev[g_prof.used++], s);
}

struct Ws {
    float* x;        // [B,E]   embedding rows of the current tokens
    float* gi;       // [B,3H]
    float* gh;       // [B,3H]
    float* pmax;     // [B,S]
    float* psum;     // [B,S]
    int64_t* parg;   // [B,S]
    int64_t* tok;    // [B]
    float* h;        // [2][L][B,H]  (ping-pong: the fused cell step writes h' beside h)
    float* c;        // [L][B,H]     (LSTM: the cell state, updated in place)
    size_t bytes;
};

inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

Ws carve(void* base, int B, int E, int H, int L, int S, bool lstm = false) {
    const int G = lstm ? 4 : 3;
    Ws w;
    char* p = static_cast<char*>(base);
    size_t off = 0;
    auto take = [&](size_t n) { size_t o = off; off += up256(n); return p ? p + o : nullptr; };
    w.x = reinterpret_cast<float*>(take((size_t)B * E * 4));
    w.gi = reinterpret_cast<float*>(take((size_t)B * G * H * 4));
    w.gh = reinterpret_cast<float*>(take((size_t)B * G * H * 4));
    w.pmax = reinterpret_cast<float*>(take((size_t)B * S * 4));
    w.psum = reinterpret_cast<float*>(take((size_t)B * S * 4));
    w.parg = reinterpret_cast<int64_t*>(take((size_t)B * S * 8));
    w.tok = reinterpret_cast<int64_t*>(take((size_t)B * 8));
    w.h = reinterpret_cast<float*>(take((size_t)2 * L * B * H * 4));
    w.c = lstm ? reinterpret_cast<float*>(take((size_t)L * B * H * 4)) : nullptr;
    w.bytes = off;
    return w;
}

__global__ void fill_i64_kernel(int64_t* p, int64_t v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// Best of the S vocabulary splits of a step (ties: the smaller split, i.e. the smaller column, as
// lse_combine_kernel) -> tokens[t], and straight on to the embedding rows of those tokens -> x, the input of
// step t+1: one wave per sentence, one launch instead of two.
__global__ void combine_gather_kernel(const float* __restrict__ pm, const int64_t* __restrict__ pa, int M, int S,
                                      int64_t* __restrict__ tok_out, const float* __restrict__ table, int64_t ldt,
                                      int E, float* __restrict__ x) {
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const int lane = threadIdx.x & 63;
    const float* rm = pm + (int64_t)m * S;
    float mx = -3.0e38f;
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
    const int64_t tok = pa[(int64_t)m * S + (best == 0x7fffffff ? 0 : best)];  // same in every lane
    if (lane == 0) tok_out[m] = tok;
    const float* src = table + tok * ldt;
    float* dst = x + (int64_t)m * E;
    for (int e = lane; e < E; e += 64) dst[e] = src[e];
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

int check_shape(int B, int E, int H, int n_layers, int n_split, int V) {
    return (B <= 0 || E <= 0 || H <= 0 || V <= 0 || n_layers <= 0 || n_layers > DSSVAE_DECODE_MAX_LAYERS || n_split <= 0)
               ? DSSVAE_EINVAL : DSSVAE_OK;
}

// the loop for either cell: c0 == NULL -> GRU, else LSTM (weights [4H, .], the state is the pair (h, c))
int greedy_loop(const dssvae_decode_params* p, const float* h0, const float* c0, int B, int max_len, int64_t sos_id,
                int n_split, int64_t* tokens, float* h_final, float* c_final, void* workspace, int64_t workspace_bytes,
                void* stream) {
    const bool lstm = c0 != nullptr;
    if (!p || !h0 || !tokens || !workspace || B <= 0 || max_len <= 0 || n_split <= 0) return DSSVAE_EINVAL;
    const int E = p->E, H = p->H, L = p->n_layers, V = p->V;
    if (E <= 0 || H <= 0 || V <= 0 || L <= 0 || L > DSSVAE_DECODE_MAX_LAYERS || !p->embedding || !p->w_out ||
        p->ld_embedding < E || p->ld_out < H + E || sos_id < 0 || sos_id >= p->n_embeddings || p->n_embeddings < V)
        return DSSVAE_EINVAL;
    for (int l = 0; l < L; ++l)
        if (!p->w_ih[l] || !p->w_hh[l] || !p->b_ih[l] || !p->b_hh[l]) return DSSVAE_EINVAL;
    if (!aligned16(workspace)) return DSSVAE_EINVAL;
    const Ws w = carve(workspace, B, E, H, L, part_cols(n_split, V), lstm);
    if ((int64_t)w.bytes > workspace_bytes) return DSSVAE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(w.h, h0, (size_t)L * B * H * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return (int)e;
    if (lstm) {
        e = hipMemcpyAsync(w.c, c0, (size_t)L * B * H * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fill_i64_kernel, dim3((B + 255) / 256), dim3(256), 0, s, w.tok, sos_id, B);
    DSSVAE_CHECK_LAUNCH();
    int rc;
    int cur[DSSVAE_DECODE_MAX_LAYERS] = {0, 0, 0, 0};  // which half of the ping-pong holds layer l's state
#define TRY(call) do { rc = (call); if (rc != DSSVAE_OK) return rc; } while (0)
    TRY(dssvae_embed_gather(p->embedding, p->ld_embedding, w.tok, B, E, nullptr, w.x, E, stream));
    // The tokens of step t are picked from the arg-max partials by the FIRST cell launch of step t + 1 (which reads
    // their embedding rows straight into its image and leaves tokens and rows behind for the projection), so a step
    // is n_layers + 1 launches; only the last step -- or a cell outside the fused kernel's range -- takes the
    // separate pick-and-gather launch.
    bool pending = false;  // the previous step's partials have not been turned into tokens + rows yet
    int S_prev = 0;
    auto pick_alone = [&](int t_of) {
        hipLaunchKernelGGL(combine_gather_kernel, dim3((B + 3) / 4), dim3(256), 0, s, w.pmax, w.parg, B, S_prev,
                           tokens + (size_t)t_of * B, p->embedding, p->ld_embedding, E, w.x);
    };
    for (int t = 0; t < max_len; ++t) {
        const float* in = w.x;
        int in_dim = E;
        for (int l = 0; l < L; ++l) {
            float* hl = w.h + ((size_t)cur[l] * L + l) * B * H;
            float* hn = w.h + ((size_t)(cur[l] ^ 1) * L + l) * B * H;
            float* cl = lstm ? w.c + (size_t)l * B * H : nullptr;
            rc = DSSVAE_ETOOBIG;
            if (l == 0 && pending) {
                rc = lstm ? dssvae_skinny_lstm_step_pick(w.pmax, w.parg, S_prev, tokens + (size_t)(t - 1) * B,
                                                         p->embedding, p->ld_embedding, E, w.x, E, hl, H, H, cl, cl, H,
                                                         p->w_ih[l], in_dim, p->w_hh[l], H, p->b_ih[l], p->b_hh[l], B,
                                                         hn, H, stream)
                          : dssvae_skinny_gru_step_pick(w.pmax, w.parg, S_prev, tokens + (size_t)(t - 1) * B,
                                                        p->embedding, p->ld_embedding, E, w.x, E, hl, H, H, p->w_ih[l],
                                                        in_dim, p->w_hh[l], H, p->b_ih[l], p->b_hh[l], B, hn, H, stream);
                if (rc != DSSVAE_OK && rc != DSSVAE_ETOOBIG) return rc;
                if (rc == DSSVAE_ETOOBIG) {
                    pick_alone(t - 1);
                    DSSVAE_CHECK_LAUNCH();
                }
                pending = false;
            }
            if (rc == DSSVAE_ETOOBIG)
                rc = lstm ? dssvae_skinny_lstm_step(in, in_dim, in_dim, hl, H, H, cl, cl, H, p->w_ih[l], in_dim,
                                                    p->w_hh[l], H, p->b_ih[l], p->b_hh[l], B, hn, H, stream)
                          : dssvae_skinny_gru_step(in, in_dim, in_dim, hl, H, H, p->w_ih[l], in_dim, p->w_hh[l], H,
                                                   p->b_ih[l], p->b_hh[l], B, hn, H, stream);
            if (rc == DSSVAE_OK) {
                cur[l] ^= 1;
                hl = hn;
            } else if (rc == DSSVAE_ETOOBIG) {  // outside the fused kernel's range: two products + the cell
                const int G = lstm ? 4 : 3;
                TRY(small_linear(in, in_dim, p->w_ih[l], p->b_ih[l], B, G * H, w.gi, stream));
                if (lstm) {
                    // (dssvae_lstm_step_fwd adds its bias argument to gh itself: the product goes in without one)
                    TRY(small_linear(hl, H, p->w_hh[l], nullptr, B, G * H, w.gh, stream));
                    TRY(dssvae_lstm_step_fwd(w.gi, (int64_t)G * H, w.gh, p->b_hh[l], hl, cl, nullptr, 0, B, H, hl, cl,
                                             nullptr, 0, nullptr, 0, stream));  // element-wise: in place is safe
                } else {
                    TRY(small_linear(hl, H, p->w_hh[l], p->b_hh[l], B, G * H, w.gh, stream));
                    TRY(dssvae_gru_cell(w.gi, w.gh, hl, B, H, hl, stream));  // element-wise: in place is safe
                }
            } else {
                return rc;
            }
            in = hl;
            in_dim = H;
        }
        int S = dssvae_skinny_argmax_splits(B, V);
        prof_mark(s);
        rc = dssvae_skinny_argmax(in, H, H, w.x, E, E, p->w_out, p->ld_out, p->b_out, B, V, w.pmax, w.psum, w.parg,
                                  stream);
        if (rc == DSSVAE_ETOOBIG) {
            S = n_split;
            rc = dssvae_linear_lse_fwd(in, H, H, w.x, E, E, p->w_out, p->ld_out, p->b_out, B, V, n_split, w.pmax,
                                       w.psum, w.parg, 1, nullptr, 0, stream);
        }
        prof_mark(s);
        if (rc != DSSVAE_OK) return rc;
        pending = true;
        S_prev = S;
    }
    if (pending) {
        pick_alone(max_len - 1);
        DSSVAE_CHECK_LAUNCH();
    }
#undef TRY
    if (h_final) {
        for (int l = 0; l < L; ++l) {
            e = hipMemcpyAsync(h_final + (size_t)l * B * H, w.h + ((size_t)cur[l] * L + l) * B * H, (size_t)B * H * 4,
                               hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return (int)e;
        }
    }
    if (lstm && c_final) {
        e = hipMemcpyAsync(c_final, w.c, (size_t)L * B * H * 4, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return (int)e;
    }
    return DSSVAE_OK;
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
    return greedy_loop(p, h0, nullptr, B, max_len, sos_id, n_split, tokens, h_final, nullptr, workspace,
                       workspace_bytes, stream);
}

// The loop for nn.LSTM decoders (rnn_cell = "lstm": decoder/rnn_decoder.py:43,65,160-168): the state is (h, c)
extern "C" int64_t dssvae_greedy_decode_lstm_workspace_bytes(int B, int E, int H, int n_layers, int n_split, int V) {
    if (check_shape(B, E, H, n_layers, n_split, V) != DSSVAE_OK) return DSSVAE_EINVAL;
    return (int64_t)carve(nullptr, B, E, H, n_layers, part_cols(n_split, V), true).bytes;
}

extern "C" int dssvae_greedy_decode_lstm(const dssvae_decode_params* p, const float* h0, const float* c0, int B,
                                         int max_len, int64_t sos_id, int n_split, int64_t* tokens, float* h_final,
                                         float* c_final, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!c0) return DSSVAE_EINVAL;
    return greedy_loop(p, h0, c0, B, max_len, sos_id, n_split, tokens, h_final, c_final, workspace, workspace_bytes,
                       stream);
}

// The same loop with its workspace in the library's own per-GPU `scratch` array: no hipMalloc / hipFree, no
// allocating routine; shapes whose workspace exceeds DSSVAE_DECODE_NOALLOC_BYTES are rejected (DSSVAE_ETOOBIG).
// The loop overwrites every scratch region before reading it (h from h0, tok from sos, x from the gather, the
// partials by the arg-max kernel of each step), so nothing has to be cleared between calls; calls that overlap
// on one GPU (two streams) share the array and race.
extern "C" int dssvae_greedy_decode_noalloc(const dssvae_decode_params* p, const float* h0, int B, int max_len,
                                            int64_t sos_id, int n_split, int64_t* tokens, float* h_final,
                                            void* stream) {
    if (!p || B <= 0 || n_split <= 0) return DSSVAE_EINVAL;
    const int64_t need = dssvae_greedy_decode_workspace_bytes(B, p->E, p->H, p->n_layers, n_split, p->V);
    if (need < 0) return (int)need;
    if (need > (int64_t)DSSVAE_DECODE_NOALLOC_BYTES) return DSSVAE_ETOOBIG;
    void* ptr = nullptr;
    const int rc = scratch_address(&ptr);
    if (rc != DSSVAE_OK) return rc;
    return dssvae_greedy_decode(p, h0, B, max_len, sos_id, n_split, tokens, h_final, ptr, need, stream);
}

// Measurement hooks (no reference counterpart): arm = 1 starts bracketing every projection + arg-max launch of the
// greedy loops with HIP events on their stream (and forgets earlier spans), arm = 0 stops.  dssvae_decode_profile_read
// waits for the recorded events and returns the summed duration and the number of launches.
extern "C" int dssvae_decode_profile(int arm) {
    g_prof.used = 0;
    g_prof.on = arm != 0;
    return DSSVAE_OK;
}

extern "C" int dssvae_decode_profile_read(double* total_ms_host, int* launches_host) {
    if (!total_ms_host || !launches_host) return DSSVAE_EINVAL;
    double tot = 0.0;
    int n = 0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        hipError_t e = hipEventSynchronize(g_prof.ev[i + 1]);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]);
        if (e != hipSuccess) return (int)e;
        tot += ms;
        ++n;
    }
    *total_ms_host = tot;
    *launches_host = n;
    return DSSVAE_OK;
}
