/*
 * dssvae_hip.h -- C-ABI of the MI355X (gfx950) hot path for DSS-VAE training / generation.
 *
 * Plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in
 * `_host`; `stream` is a hipStream_t passed as void*.  All matrices are fp32, row-major, with an
 * explicit leading dimension (elements).  Token ids / targets are int64 (the reference's
 * LongTensor).  Every entry point returns 0 on success, a negative DSSVAE_E* code on a
 * rejected argument (nothing launched), or a positive hipError_t.  Nothing here allocates: the
 * caller owns all buffers, workspaces included (sizes: dssvae_*_workspace_bytes).
 *
 * The reference (baoy-nlp/DSS-VAE) is pure Python on top of PyTorch ops; it has no FFI of its
 * own for this path.  Each entry point therefore cites the reference call site (file:line under
 * /root/reference) whose PyTorch op(s) it replaces; INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add at that call site.
 */
#ifndef DSSVAE_HIP_H
#define DSSVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSSVAE_OK 0
#define DSSVAE_EINVAL (-1)   /* bad shape / null pointer / misaligned leading dimension        */
#define DSSVAE_ETOOBIG (-2)  /* persistent grid would not be co-resident / LDS budget exceeded */
#define DSSVAE_ETIMEOUT (-3) /* a persistent kernel gave up waiting on a peer (error word set) */

/* library / device probing (no reference counterpart) */
#define DSSVAE_ABI_VERSION 9 /* 3: + dssvae_greedy_decode_noalloc; skip_flag in dssvae_adam_step; segments and cell heights in dssvae_gru_*_inst; dssvae_gemm_f32_dlogits; LSTM step.  4: + dssvae_gemm_last_launches, dssvae_gru_error_word; cell / c0 / c_last / dc_last / dc0 in dssvae_gru_*_inst.  5: + dssvae_gru_layout.  6: + dssvae_weighted_sums_fwd / _bwd, dssvae_skinny_gru_step_pick.  7: + dssvae_set_words and the *_dyn entry points (per-step scalars read on the device: HIP-graph replay), dssvae_batch_rows / dssvae_keep_rows (minibatches cut from a device-resident id cache).  8: the training path's logits slab is normalised per 128-column tile: + dssvae_linear_lse_fwd_norm / dssvae_lse_combine_tiles, dssvae_ce_rowscale replaces dssvae_ce_rowinfo, dssvae_gemm_f32_dlogits takes the scale table.  9: + dssvae_skinny_lstm_step / _pick, dssvae_greedy_decode_lstm (+ _workspace_bytes), dssvae_ce_wgrad_cols, dssvae_skinny_argmax_padded_rows */
int dssvae_abi_version(void);
int dssvae_device_cus(int* n_cus_host);

/* ---------------------------------------------------------------------------------------------
 * GEMM: C[M,N] = alpha * sum_k A(m,k) * B(n,k) + beta * C + bias[n]
 *   trans_a = 0: A(m,k) = A[row(m)*lda + k]   (row(m) = a_rows ? a_rows[m] : m  -> fused gather)
 *   trans_a = 1: A(m,k) = A[k*lda + m]
 *   trans_b = 0: B(n,k) = B[n*ldb + k]        (nn.Linear weight [out,in])
 *   trans_b = 1: B(n,k) = B[k*ldb + n]
 * Replaces: the input half of nn.GRU for all timesteps (encoder/rnn_encoder.py:52,
 * decoder/rnn_decoder.py:65), nn.Linear in hidden_to_latent / latent_for_init
 * (vae/enhance_syntax_vae.py:433-436,451-452), and their autograd backward (dgrad / wgrad).
 * ------------------------------------------------------------------------------------------- */
int dssvae_gemm_f32(const float* A, int64_t lda, int trans_a, const int64_t* a_rows,
                    const float* B, int64_t ldb, int trans_b,
                    float* C, int64_t ldc, const float* bias,
                    int M, int N, int K, float alpha, float beta, void* stream);
/* Same, with an explicit split of the K loop over `ksplit` workgroups per output tile (partials
 * combined with fp32 atomic adds; needs beta in {0,1}; ksplit <= 0: chosen from the tile count so
 * that small-M/N, large-K products -- the dgrad/wgrad shapes of the vocabulary projection and the
 * GRU weight gradients -- still fill the chip).  dssvae_gemm_f32 is ksplit = 0. */
/* How many gemm_store_kernel launches the calling thread's last dssvae_gemm_f32* / dssvae_gemm_f32_dlogits call made:
 * 1, or 2 when the output's last 128-wide tile column would have been at most half full and the call ran the full
 * tile columns and the remainder as two launches (profilers divide an entry point's time by this; no reference
 * counterpart). */
int dssvae_gemm_last_launches(void);
int dssvae_gemm_f32_splitk(const float* A, int64_t lda, int trans_a, const int64_t* a_rows,
                           const float* B, int64_t ldb, int trans_b,
                           float* C, int64_t ldc, const float* bias,
                           int M, int N, int K, float alpha, float beta, int ksplit, void* stream);

/* Up to DSSVAE_GEMM_MAX_GROUP independent products of ONE layout (trans_a, trans_b as above) in one
 * launch: C_i = A_i . B_i^T + bias_i (alpha = 1, beta = 0; outputs whose K loop is split are zero-filled
 * by the library first).
 * Same reference call sites as dssvae_gemm_f32: the per-direction / per-head input projections of
 * nn.GRU and their weight gradients (encoder/rnn_encoder.py:52, decoder/rnn_decoder.py:65), which
 * are too small to fill the chip one at a time. */
#define DSSVAE_GEMM_MAX_GROUP 8
typedef struct {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    const float* bias;
    int32_t M, N, K;
} dssvae_gemm_problem;
int dssvae_gemm_f32_grouped(const dssvae_gemm_problem* problems_host, int n, int trans_a, int trans_b,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused vocabulary projection + online log-sum-exp (generation and the bag-of-words heads never write logits; the
 * training path asks for them ONCE as a side output, see `logits`).
 *   logit[m,n] = Xh[m,:Kh] . W[n,:Kh] + Xe[m,:Ke] . W[n,Kh:Kh+Ke] + bias[n]
 * dssvae_linear_lse_fwd writes, per row m and column split s (n_split of them), the running
 * max and sum(exp(logit - max)) of that split into part_max/part_sum [M, n_split]; with
 * want_argmax it also writes the split's best column into part_arg [M, n_split].
 * logits (may be NULL): side output, the logits of these M rows into a caller slab [M, ldl]
 * (the training path passes a row chunk of a slab of up to 1024 MiB per head and backward reads it back through
 * dssvae_gemm_f32_dlogits, which applies the softmax gradient while staging, so the projection is never recomputed
 * and the slab never rewritten; dssvae_softmax_grad_inplace is the unfused form of that).
 * Replaces: RNNDecoder.forward_step's `self.out(cat(output, embedded))` + F.log_softmax
 * (decoder/rnn_decoder.py:70-73,94) and the scorer Linear+LogSoftmax of the bag-of-words heads
 * (vae/enhance_syntax_vae.py:607-615).  Xe may be NULL (Ke = 0).
 * ------------------------------------------------------------------------------------------- */
int dssvae_linear_lse_fwd(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                          const float* W, int64_t ldw, const float* bias, int M, int N, int n_split,
                          float* part_max, float* part_sum, int64_t* part_arg, int want_argmax,
                          float* logits, int64_t ldl, void* stream);
/* The training path's form: the slab receives the row chunk NORMALISED per 128-column tile,
 * slab[m,n] = exp(logit[m,n] - tile_max[n / 128][m]) (in (0, 1]), and tile_max / tile_sum [ceil(N / 128), ldt >= M] the
 * maximum and the sum of those exponentials of every (tile, row) (the exponentials the statistics need anyway).  No state
 * crosses a tile, so the tiles are dealt as one list in equal shares to two workgroups per CU (no n_split).
 * dssvae_lse_combine_tiles gives lse[m]; dssvae_ce_rowscale then turns tile_max into one scale per (tile, row) in place,
 * and dssvae_gemm_f32_dlogits applies it while staging: ONE multiply per slab element in the gradient GEMMs, where raw
 * logits cost an fma and an exp each (every VALU instruction between fp32 MFMAs costs matrix time: DESIGN.md section 5).
 * slab 16-byte aligned, ldl % 4 == 0, N % 4 == 0, else DSSVAE_EINVAL.  Same reference ops as dssvae_linear_lse_fwd. */
int dssvae_linear_lse_fwd_norm(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                               const float* W, int64_t ldw, const float* bias, int M, int N, float* slab,
                               int64_t ldl, float* tile_max, float* tile_sum, int64_t ldt, void* stream);
int dssvae_lse_combine_tiles(const float* tile_max, const float* tile_sum, int M, int tiles, int64_t ldt, float* lse,
                             void* stream);

/* X[m,n] <- coef[m] * exp(X[m,n] - lse[m]) in place on a logits slab [M, ldx] (ldx % 4 == 0);
 * colsum (may be NULL) [N] += column sums of the result (atomics: zero it first).  The softmax part
 * of autograd(log_softmax + gather) (decoder/rnn_decoder.py:94-102). */
int dssvae_softmax_grad_inplace(float* X, int64_t ldx, const float* lse, const float* coef, int M, int N,
                                float* colsum, void* stream);

/* The last few output columns of that weight gradient (`trans_a` form) without the matrix pipe:
 * C[v, j] (+)= sum_m slab[m, v] * scale[(v >> 7) * ld_scale + m] * X[m, j], v < cols, j < nj <= 8 -- the remainder
 * of [h | e | 1] behind the 128-wide tile columns (904 = 7 x 128 + 8; decoder/rnn_decoder.py:70-73's `out` Linear in
 * backward).  A stream over the slab on the vector ALUs (HBM-bound), meant to run on a second stream beside the
 * MFMA launch of the other columns.  beta 0 or 1.  DSSVAE_ETOOBIG: nj > 8 or the slab's rows are not 16-byte
 * granules (the caller keeps those columns in dssvae_gemm_f32_dlogits). */
int dssvae_ce_wgrad_cols(const float* slab, int64_t ldl, int rows, int cols, const float* scale, int64_t ld_scale,
                         const float* X, int64_t ldx, int nj, float* C, int64_t ldc, float beta, void* stream);
/* The two consumers of a logits slab with the softmax part of the gradient applied WHILE THE SLAB TILE IS STAGED
 * (no separate pass over the slab, nothing written back): with P[m,n] = coef[m] * exp(logit[m,n] - lse[m])
 *                                                              = slab[m,n] * scale[n / 128][m]   (slab: dssvae_linear_lse_fwd_norm),
 *   trans_a = 0:  C[rows, N] (+)= P . B          B [cols, N] row-major   (dgrad: B = W_out panel)
 *   trans_a = 1:  C[cols, N] (+)= P^T . B        B [rows, N] row-major   (wgrad: B = [h | e | 1 0 0 0]: the column of
 *                                                ones makes the bias gradient come out as one more output column)
 * beta in {0, 1}.  scale [ceil(cols / 128), ld_scale >= rows] = coef[m] * exp(tile_max - lse[m]) (0 where coef = 0), made
 * in place from the tile_max table by dssvae_ce_rowscale(lse, coef, M, tiles, table, ld).  The one-hot part of dlogits (dL[m, target[m]] -= w[m]) is applied to the PRODUCTS by
 * dssvae_ce_onehot_fixup: dX[m,:] -= w W_panel[target[m],:] and dW[target[m],:] -= w Xc[m,:] (w = g[m], or 1 when g
 * is NULL; rows whose target is ignore_index are skipped; dX or dW may be NULL).
 * The slab and B must be 16-byte aligned with leading dimensions / cols / N multiples of 4, else DSSVAE_ETOOBIG
 * (the caller then uses dssvae_softmax_grad_inplace + dssvae_onehot_sub + dssvae_gemm_f32).
 * Replaces autograd of log_softmax + gather + the Linear's backward (decoder/rnn_decoder.py:70-73,94-102). */
int dssvae_ce_rowscale(const float* lse, const float* coef, int M, int tiles, float* table, int64_t ld, void* stream);
int dssvae_gemm_f32_dlogits(const float* slab, int64_t ldl, int rows, int cols, const float* scale, int64_t ld_scale,
                            int trans_a, const float* B, int64_t ldb, float* C, int64_t ldc, int N, float beta,
                            void* stream);
int dssvae_ce_onehot_fixup(const int64_t* target, int64_t ignore_index, const float* g, int rows,
                           const float* W_panel, int64_t ldw, int Kx, float* dX, int64_t lddx,
                           const float* Xc, int64_t ldxc, int Kw, float* dW, int64_t lddw, void* stream);

/* lse[m] = log sum_n exp(logit[m,n]) from the split partials; argmax[m] (may be NULL). */
int dssvae_lse_combine(const float* part_max, const float* part_sum, const int64_t* part_arg,
                       int M, int n_split, float* lse, int64_t* argmax, void* stream);

/* picked[m,t] = logit[m, target[m,t]] (0 where target == ignore_index), T targets per row.
 * Replaces torch.gather(log_scores, 1, tgt) (decoder/rnn_decoder.py:98) and the NLLLoss gather
 * of bag_of_word_loss (utils/loss_funcs.py:117-129). */
int dssvae_picked_logit(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                        const float* W, int64_t ldw, const float* bias, const int64_t* target, int T,
                        int64_t ignore_index, int M, float* picked, void* stream);

/* Row weights and the loss itself, fused (each replaces three elementwise / reduction launches):
 *   dssvae_ce_count : cnt[m] = number of targets of row m that are not ignore_index, as float
 *   dssvae_ce_finish: out[s] = sum over rows m of sentence s (rows s*group .. s*group+group-1) and their targets t
 *                     of (lse[m] - logit[m, target[m,t]]), ignored targets skipped -- the gather + mask + sum over
 *                     time of RNNDecoder.score_decoding_results (decoder/rnn_decoder.py:96-103) and the NLLLoss sum
 *                     of bag_of_word_loss (utils/loss_funcs.py:117-129); group = 1: per-row losses.  row_nll: [M]
 *                     floats of scratch for the per-row losses (may be NULL when group = 1). */
int dssvae_ce_count(const int64_t* target, int T, int64_t ignore_index, int M, float* cnt, void* stream);
int dssvae_ce_finish(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                     const float* W, int64_t ldw, const float* bias, const int64_t* target, int T,
                     int64_t ignore_index, const float* lse, int M, int group, float* row_nll, float* out,
                     void* stream);

/* Backward, one chunk of the vocabulary (the caller passes W, bias offset to the chunk's first
 * column and N = columns in the chunk; or all of it): dL[m,n] = coef[m] * exp(logit[m,n] - lse[m])
 * written to the caller's slab dL [M, ldd] (ldd >= N); colsum (may be NULL) [N] += sum_m dL[m,n]
 * (the softmax part of the bias gradient, accumulated with atomics: zero it first).  The caller
 * then subtracts the one-hot part (dssvae_onehot_sub) and runs dssvae_gemm_f32 for dX / dW.
 * Replaces autograd of log_softmax+gather (decoder/rnn_decoder.py:94-102). */
int dssvae_linear_ce_dlogits(const float* Xh, int64_t ldh, int Kh, const float* Xe, int64_t lde, int Ke,
                             const float* W, int64_t ldw, const float* bias, const float* lse,
                             const float* coef, int M, int N, float* dL, int64_t ldd, float* colsum,
                             void* stream);

/* For every non-ignored target in the chunk's column range [n0, n1):
 * dL[m, target[m,t] - n0] -= g[m,t] and (colsum != NULL) colsum[target - n0] -= g[m,t]
 * (atomic: targets may repeat). */
int dssvae_onehot_sub(float* dL, int64_t ldd, const int64_t* target, int T, int64_t ignore_index,
                      const float* g, int M, int64_t n0, int64_t n1, float* colsum, void* stream);

/* out[n] (+)= sum_m X[m,n]  (bias gradients). */
int dssvae_colsum(const float* X, int64_t ldx, int M, int N, float* out, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Embedding rows: out[m,:] = table[ids[m],:] * (mask ? mask[m,:] : 1)   (networks/embeddings.py:77-87)
 * and its backward scatter-add dtable[ids[m],:] += g[m,:] * mask.
 * ------------------------------------------------------------------------------------------- */
int dssvae_embed_gather(const float* table, int64_t ldt, const int64_t* ids, int M, int E,
                        const float* mask, float* out, int64_t ldo, void* stream);
int dssvae_embed_scatter_add(float* dtable, int64_t ldt, const int64_t* ids, int M, int E,
                             const float* mask, const float* g, int64_t ldg, void* stream);

/* Bernoulli keep-mask scaled by 1/(1-p) (nn.Dropout semantics), counter-based RNG. */
int dssvae_dropout_mask(float* mask, int64_t n, float p, uint64_t seed, void* stream);

/* y = x * mask(seed) with the mask of dssvae_dropout_mask for the same (p, seed), without materialising it: the
 * caller applies the same call to dy in backward (nn.Dropout between GRU layers, decoder/rnn_decoder.py:43). */
int dssvae_dropout_apply(const float* x, int64_t n, float p, uint64_t seed, float* y, void* stream);

/* Word dropout: ids -> unk where rand < rate, never on sos/pad/eos (utils/nn_funcs.py:13-23).
 * `rand` NULL -> drawn on device from (seed). */
int dssvae_unk_replace(const int64_t* ids, int64_t n, const float* rand, float rate, uint64_t seed,
                       int64_t sos, int64_t pad, int64_t eos, int64_t unk, int64_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Latent heads: z = eps * exp(0.5*logv) + mean ; kl_out[0] += -0.5*sum(1 + logv - mean^2 - e^logv)
 * (vae/enhance_syntax_vae.py:417-424, vae/base_vae.py:198-201).  kl_out must be zeroed by the
 * caller.  eps NULL -> z = mean (eval).  Backward: given dz and the scalar dkl,
 * dmean = dz + dkl*mean ; dlogv = dz*eps*0.5*exp(0.5*logv) + dkl*0.5*(exp(logv)-1).
 * ------------------------------------------------------------------------------------------- */
int dssvae_reparam_kl_fwd(const float* mean, const float* logv, const float* eps, int64_t n,
                          float* z, float* kl_out, void* stream);
int dssvae_reparam_kl_bwd(const float* mean, const float* logv, const float* eps, const float* dz,
                          const float* dkl, int64_t n, float* dmean, float* dlogv, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Scalar loss assembly: out[j] = sum_k coef[j*n_in + k] * sum(in[k][0 .. len[k])), j < n_out.
 * Replaces the one-element torch ops between the heads' per-sentence log-likelihoods / KL terms and
 * the loss dictionary (vae/enhance_syntax_vae.py:212-288 forward: the adv / mul / nll sums and their
 * weights; :290-340 get_loss: division by the batch size, KL weight, 'Loss', 'ELBO', 'Model Score').
 * `in`, `len`, `coef` are HOST arrays (n_in device pointers, their element counts, the row-major
 * n_out x n_in coefficients); n_in, n_out <= DSSVAE_WSUMS_MAX.  Sums run in a fixed order.
 * Backward: din[k][i] = sum_j *dout[j] * coef[j*n_in + k] for every k with din[k] != NULL; dout is a
 * HOST array of n_out device pointers to one float each, NULL for an output without a gradient.
 * ------------------------------------------------------------------------------------------- */
#define DSSVAE_WSUMS_MAX 16
int dssvae_weighted_sums_fwd(const float* const* in, const int32_t* len, int n_in, const float* coef, int n_out,
                             float* out, void* stream);
int dssvae_weighted_sums_bwd(float* const* din, const int32_t* len, int n_in, const float* coef, int n_out,
                             const float* const* dout, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Persistent GRU recurrence (recurrent half of nn.GRU; encoder/rnn_encoder.py:52,
 * decoder/rnn_decoder.py:65).  One launch runs up to DSSVAE_GRU_MAX_INST independent
 * recurrences ("instances": directions / encoders / heads) concurrently on disjoint workgroups;
 * W_hh slices stay resident in LDS for all T steps; gate math is fused; timesteps advance
 * with per-(instance, batch-block) arrival counters (no grid-wide barrier).
 *   xproj [B,T,ldx]  : x_t W_ih^T + b_ih for all t (gate order r,z,n), row stride ldx per (b,t)
 *   whh [3H,H], bhh [3H], h0 [B,H] or NULL (zeros)
 *   lengths [B] int32 or NULL: packed-sequence semantics (row b advances only while t < len[b];
 *            `reverse` walks t = T-1..0; padded outputs are 0)
 *   out [B,T,ldo]    : h_t (row stride ldo per (b,t), so two directions can interleave)
 *   h_last [B,H]     : final state;  gates [B,T,5H]: r,z,n,(W_hn h + b_hn),h_{t-1} saved for
 *                      backward (NULL in inference)
 * sync: caller-provided workspace of dssvae_gru_workspace_bytes(...) bytes, need not be zeroed.
 *
 * cell = DSSVAE_CELL_LSTM runs an LSTM recurrence (nn.LSTM behind RNNBase's rnn_cell = "lstm", networks/rnn_base.py:
 * 32-36) through the same kernels and entry points (one cell type per launch): gate order i, f, g, o; xproj
 * [B,T,4H], whh [4H,H], bhh [4H]; c0 / c_last [B,H] beside h0 / h_last; `gates` rows are the stepwise kernels' saved
 * row [i | f | g | o | c_{t-1} | h_{t-1} | tanh c_t] (7H).  Backward: dgx [B,T,4H] is the gradient wrt the gate
 * pre-activations (the same for the input and the recurrent product: dgh is not written and may be NULL; dbias_x and
 * dbias_h, [4H] each, receive the same sums), dc_last / dc0 beside dh_last / dh0.
 * ------------------------------------------------------------------------------------------- */
#define DSSVAE_GRU_MAX_INST 8
#define DSSVAE_CELL_GRU 0
#define DSSVAE_CELL_LSTM 1
typedef struct {
    const float* xproj; int64_t ldx;
    const float* whh; const float* bhh; const float* h0;
    const int32_t* lengths;
    float* out; int64_t ldo;
    float* h_last; float* gates;
    int32_t B, T, H, reverse;
    /* A launch may run a SEGMENT of the recurrence: steps t0 .. t0 + T - 1 (in the order the recurrence walks
     * them) of T_total steps; xproj / out / gates are then [B, T_total, .] and h0 / h_last the states at the
     * segment's ends.  T_total = 0: the whole recurrence (t0 must be 0).  The caller chains segments on one
     * stream (different instances of a layer end at different steps: the long ones get the chip to themselves,
     * re-cut into smaller cells, once the short ones are done). */
    int32_t t0, T_total;
    /* batch rows per workgroup ("cell" = rows x 16 hidden units): 16, 32, 64 or 128; 0 = the library chooses for
     * the launch (64 when the grid is resident that way, else 128) */
    int32_t rows;
    int32_t cell;                     /* DSSVAE_CELL_GRU (0) or DSSVAE_CELL_LSTM */
    const float* c0; float* c_last;   /* LSTM only: cell state at the segment's ends ([B,H]; c0 NULL = zeros, c_last may be NULL) */
} dssvae_gru_fwd_inst;

typedef struct {
    /* saved from forward */
    const float* whh; const float* h0; const int32_t* lengths;
    const float* out; int64_t ldo; const float* gates;
    /* incoming gradients: dout [B,T,ldd] (or NULL), dh_last [B,H] (or NULL) */
    const float* dout; int64_t ldd; const float* dh_last;
    /* produced: dgx, dgh [B,T,3H] (grad wrt xproj / wrt W_hh h + b_hh), dh0 [B,H] */
    float* dgx; float* dgh; float* dh0;
    int32_t B, T, H, reverse;
    /* optional, ACCUMULATED into (caller zero-fills): bias gradients [3H] = column sums of dgx / dgh over
     * batch and time (grad of b_ih / b_hh), so that no separate reduction pass reads dgx / dgh again */
    float* dbias_x; float* dbias_h;
    /* segment of the recurrence and cell height, as in dssvae_gru_fwd_inst: this launch walks forward steps
     * t0 + T - 1 down to t0; dh_last / dh0 are the gradients wrt the states at the segment's ends */
    int32_t t0, T_total;
    int32_t rows;
    int32_t cell;                          /* as in dssvae_gru_fwd_inst */
    const float* dc_last; float* dc0;      /* LSTM only: gradients wrt the cell state at the segment's ends (may be NULL) */
} dssvae_gru_bwd_inst;

int64_t dssvae_gru_workspace_bytes(int n_inst, const int32_t* B_host, const int32_t* H_host);
/* Workgroups of a persistent launch the device holds at once (every one must be resident: the hand-off spins on
 * peers): the runtime's occupancy for the kernel at this LDS slice (largest H of the launch) and block size
 * (largest cell height: 16 rows per wave) x the CU count.  A launch whose grid exceeds it returns DSSVAE_ETOOBIG. */
int dssvae_gru_resident_workgroups(int H_max, int rows_max, int backward, int* n_host);
/* The bound has no margin: it assumes the launch has the chip to itself or shares it with workgroups that retire
 * (GEMMs, collectives; rehearsed beside <= 64 resident foreign workgroups of 64 KiB LDS: tests/test_overlap_gpu.py).
 * Two persistent launches must never overlap on one GPU, whatever their streams: the Python host serialises them with
 * events (ops._PersistentTurn); a C caller with several host threads has to do the same.  The library's launch
 * bookkeeping (dssvae_gemm_last_launches, dssvae_decode_profile, the error word registration) is per process, not
 * per thread. */
/* The layout the library would run a launch of these instances in (no launch): *waves_host = waves per workgroup
 * (16: the K-split layouts; 4 / 8: one wave per row group), 0 = the grid is not resident with these cell heights.
 * rows_host[i] as dssvae_gru_*_inst.rows (0 = the library's choice).  For callers that pick cell heights per
 * instance (dss_vae_amd/ops.py::_schedule) and want to know BEFORE the first launch of a chain whether every launch
 * of their plan will be admitted.  Host logic of this library: no counterpart in the reference. */
int dssvae_gru_layout(int n_inst, const int32_t* B_host, const int32_t* H_host, const int32_t* rows_host, int backward,
                      int cell, int* waves_host);
int dssvae_gru_fwd(const dssvae_gru_fwd_inst* inst_host, int n_inst, void* workspace,
                   int64_t workspace_bytes, void* stream);
int dssvae_gru_bwd(const dssvae_gru_bwd_inst* inst_host, int n_inst, void* workspace,
                   int64_t workspace_bytes, void* stream);
/* Registers (word != NULL) or clears (NULL) a device int32 for the CURRENT device that every later persistent launch on
 * it sets to 1 when one of its bounded spins gives up, besides the launch's own error word: one sticky word for many
 * launches, readable on the device (dssvae_adam_step's skip_flag) or by the host, instead of a fold-in kernel behind
 * every launch.  The caller owns the memory and keeps it alive while registered.  No reference counterpart. */
int dssvae_gru_error_word(void* word);
/* Reads the error word a persistent launch leaves in its workspace (synchronises the stream). */
int dssvae_gru_check(void* workspace, void* stream);
/* Residency rehearsal (diagnostic; no counterpart in the reference): `workgroups` (<= 256) workgroups of 256 threads,
 * each holding lds_bytes (<= 64 KiB) of LDS, stay on the chip until the host writes a non-zero word to `release`
 * (coherent pinned host memory) or max_ms (<= 5000) have passed -- what a communication kernel on another stream looks
 * like to a persistent recurrence.  held_ticks (may be NULL) [workgroups]: 100 MHz ticks each workgroup waited. */
int dssvae_hold_cus(const uint32_t* release, int workgroups, int lds_bytes, int max_ms, unsigned long long* held_ticks,
                    void* stream);

/* One GRU cell step from precomputed gi = x W_ih^T + b_ih and gh = h W_hh^T + b_hh (greedy loop,
 * decoder/rnn_decoder.py:160-168): h_new = (1-z)*n + z*h. */
int dssvae_gru_cell(const float* gi, const float* gh, const float* h, int B, int H, float* h_new,
                    void* stream);

/* Stepwise GRU (same reference call sites as the persistent kernel): for recurrences too large for
 * the persistent kernel's residency budget (W_hh slice in LDS, one workgroup per CU: e.g. the stress
 * configuration's H = 2048 decoder) the caller computes gh = h_{t-1} W_hh^T with dssvae_gemm_f32 --
 * at those sizes the step is MFMA-bound, not latency-bound -- and these fuse the gate math of ONE
 * timestep.  Pointers are pre-offset to timestep t; *_row_stride = elements between batch rows.
 * fwd: h_new [B,H]; out_t (may be NULL); gates_t (may be NULL) in the layout of dssvae_gru_fwd.
 * bwd: dh [B,H] = gradient wrt h_t arriving from t+1 (NULL = 0), dout_t (may be NULL);
 *      writes dgx_t, dgh_t [B,3H] and dhz [B,H]; the caller forms dh_{t-1} = dhz + dgh_t W_hh. */
int dssvae_gru_step_fwd(const float* x_t, int64_t x_row_stride, const float* gh, const float* bhh,
                        const float* h_prev, const int32_t* lengths, int t, int B, int H, float* h_new,
                        float* out_t, int64_t out_row_stride, float* gates_t, int64_t gates_row_stride,
                        void* stream);
int dssvae_gru_step_bwd(const float* gates_t, int64_t gates_row_stride, const float* dout_t,
                        int64_t dout_row_stride, const float* dh, const int32_t* lengths, int t, int B, int H,
                        float* dgx_t, float* dgh_t, int64_t dg_row_stride, float* dhz, void* stream);

/* LSTM cell (nn.LSTM behind RNNBase's rnn_cell = "lstm", networks/rnn_base.py:32-36; encoder/rnn_encoder.py:28,52;
 * decoder/rnn_decoder.py:43,65), stepwise like dssvae_gru_step_*: the caller computes gh = h_{t-1} W_hh^T [B,4H] with
 * dssvae_gemm_f32, these fuse the cell of ONE timestep (gate order i, f, g, o; pointers pre-offset to timestep t).
 * fwd: h_new, c_new [B,H]; out_t (may be NULL); gates_t (may be NULL): the saved row [i|f|g|o|c_{t-1}|h_{t-1}|tanh c_t]
 *      (row stride >= 7H).  Rows with t >= lengths[b] keep (h, c) and output 0 (packed-sequence semantics).
 * bwd: dh, dc = gradients wrt (h_t, c_t) from t+1 (NULL = 0), dout_t (may be NULL) -> dg_t [B,4H] (gradient wrt the gate
 *      pre-activations, the same for the input and the recurrent product), dhz [B,H] (the part of dh_{t-1} that does
 *      not pass through W_hh: the caller forms dh_{t-1} = dhz + dg_t W_hh) and dc_prev [B,H]. */
int dssvae_lstm_step_fwd(const float* x_t, int64_t x_row_stride, const float* gh, const float* bhh,
                         const float* h_prev, const float* c_prev, const int32_t* lengths, int t, int B, int H,
                         float* h_new, float* c_new, float* out_t, int64_t out_row_stride, float* gates_t,
                         int64_t gates_row_stride, void* stream);
int dssvae_lstm_step_bwd(const float* gates_t, int64_t gates_row_stride, const float* dout_t, int64_t dout_row_stride,
                         const float* dh, const float* dc, const int32_t* lengths, int t, int B, int H, float* dg_t,
                         int64_t dg_row_stride, float* dhz, float* dc_prev, void* stream);

/* Skinny projections (M <= DSSVAE_SKINNY_MAX_ROWS activation rows: one decode batch), the per-step
 * products of the greedy loop (decoder/rnn_decoder.py:61-73 called from :160-168):
 *   dssvae_skinny_linear : C[M,N] = [X1 | X2] . W^T + bias
 *   dssvae_skinny_argmax : per row, max and arg-max over n of the same product, as partials
 *                          [M, S] with S = dssvae_skinny_argmax_splits(M, N) (part_sum is set to 1
 *                          so that dssvae_lse_combine picks the winner)
 * The activation rows stay in LDS, W is streamed once per 32 rows straight into the MFMA operand.
 * X2 may be NULL (K2 = 0).  Returns DSSVAE_ETOOBIG when the shape is outside the kernel's range (rows,
 * alignment to 4 floats, LDS budget): the caller then uses dssvae_gemm_f32 / dssvae_linear_lse_fwd. */
#define DSSVAE_SKINNY_MAX_ROWS 128
int dssvae_skinny_supported(int M, int K1, int K2, int64_t ld1, int64_t ld2, int64_t ldw);
/* Activation rows per workgroup of the kernel dssvae_skinny_argmax runs for this shape: 32 (skinny_kernel<ARGMAX>) or
 * 64 (skinny_argmax64_kernel: decode batches above 32 rows over a vocabulary large enough to give every CU a workgroup).
 * The MFMA work of a launch is the decode batch padded to a multiple of it.  No reference counterpart. */
int dssvae_skinny_argmax_rows(int M, int N);
/* Rows a launch of dssvae_skinny_argmax computes for a decode batch of M: with the 64-row kernel and one row chunk, the
 * full 16-row MFMA tiles plus one or two left-over rows on the vector ALUs (M = 50: 48 + 2 = 50), else M rounded up to
 * 16 (three or more left-over rows) / to the kernel's row group.  No reference counterpart (bench.py's padded-work figure). */
int dssvae_skinny_argmax_padded_rows(int M, int N);
int dssvae_skinny_argmax_splits(int M, int N);
int dssvae_skinny_linear(const float* X1, int64_t ld1, int K1, const float* X2, int64_t ld2, int K2,
                         const float* W, int64_t ldw, const float* bias, int M, int N, float* C, int64_t ldc,
                         void* stream);
/* One GRU cell step of a decode batch in one launch: h_new = GRUCell(x, h) with gi = x W_ih^T + b_ih,
 * gh = h W_hh^T + b_hh (decoder/rnn_decoder.py:61-66 in the greedy loop :160-168).  h_new must not alias
 * h_prev.  DSSVAE_ETOOBIG when the shape is outside the kernel's range (M > DSSVAE_SKINNY_MAX_ROWS, extents
 * not multiples of 4, [x | h] rows beyond the LDS budget): the caller then uses dssvae_skinny_linear /
 * dssvae_gemm_f32 + dssvae_gru_cell. */
int dssvae_skinny_gru_step(const float* X, int64_t ldx, int In, const float* h_prev, int64_t ldh, int H,
                           const float* w_ih, int64_t ld_ih, const float* w_hh, int64_t ld_hh,
                           const float* b_ih, const float* b_hh, int M, float* h_new, int64_t ldo, void* stream);

/* The same cell step for the greedy loop's first layer when the step's input tokens have not been picked yet:
 * part_max / part_arg [M, n_split] are the previous step's arg-max partials (dssvae_skinny_argmax /
 * dssvae_linear_lse_fwd with want_arg); the launch picks every sentence's token (best split, ties to the smaller
 * split: decoder/rnn_decoder.py:73 `topk(1)` of :160-168), takes the token's row of `table` as the cell's input x
 * (decoder/rnn_decoder.py:61 `self.embedding(input_var)`), and leaves tokens_out [M] and the rows x_out [M, In]
 * behind for the projection that follows.  Same ranges and error codes as dssvae_skinny_gru_step. */
int dssvae_skinny_gru_step_pick(const float* part_max, const int64_t* part_arg, int n_split, int64_t* tokens_out,
                                const float* table, int64_t ldt, int In, float* x_out, int64_t ldxo,
                                const float* Hprev, int64_t ldh, int H, const float* Wih, int64_t ldi,
                                const float* Whh, int64_t ldhh, const float* bih, const float* bhh, int M,
                                float* Hnew, int64_t ldo, void* stream);
/* The two cell-step launches for nn.LSTM decoders (RNNBase's rnn_cell = "lstm": networks/rnn_base.py:32-36,
 * decoder/rnn_decoder.py:43,65 in the greedy loop :160-168): (h_new, c_new) = LSTMCell(x, (h, c)) with weights
 * [4H, In] / [4H, H], gate order i, f, g, o.  c_new may alias c_prev (row stride ldc); h_new must not alias h_prev.
 * Same ranges and error codes as dssvae_skinny_gru_step (the caller's fallback: two dssvae_skinny_linear /
 * dssvae_gemm_f32 products + dssvae_lstm_step_fwd). */
int dssvae_skinny_lstm_step(const float* X, int64_t ldx, int In, const float* h_prev, int64_t ldh, int H,
                            const float* c_prev, float* c_new, int64_t ldc, const float* w_ih, int64_t ld_ih,
                            const float* w_hh, int64_t ld_hh, const float* b_ih, const float* b_hh, int M,
                            float* h_new, int64_t ldo, void* stream);
int dssvae_skinny_lstm_step_pick(const float* part_max, const int64_t* part_arg, int n_split, int64_t* tokens_out,
                                 const float* table, int64_t ldt, int In, float* x_out, int64_t ldxo,
                                 const float* Hprev, int64_t ldh, int H, const float* c_prev, float* c_new,
                                 int64_t ldc, const float* Wih, int64_t ldi, const float* Whh, int64_t ldhh,
                                 const float* bih, const float* bhh, int M, float* Hnew, int64_t ldo, void* stream);
int dssvae_skinny_argmax(const float* X1, int64_t ld1, int K1, const float* X2, int64_t ld2, int K2,
                         const float* W, int64_t ldw, const float* bias, int M, int N, float* part_max,
                         float* part_sum, int64_t* part_arg, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Greedy generation (RNNDecoder.forward with inputs=None: decoder/rnn_decoder.py:160-168,193-199;
 * BaseVAE.decode_to_sentence, vae/base_vae.py:169-190; RNNPredictor.predict,
 * vae/enhance_syntax_vae.py:581-592): always max_len steps from <s>; per step embedding rows ->
 * n_layers GRU cells -> [h;e] W_out^T + b fused with the arg-max over the vocabulary.  One call
 * enqueues the whole loop on `stream`; nothing is allocated, synchronised or copied to the host.
 *   h0 [n_layers,B,H]; tokens [max_len,B] int64 (out); h_final [n_layers,B,H] (out, may be NULL)
 *   n_split: vocabulary splits of the fused projection/arg-max (<= ceil(V/128))
 *   workspace: dssvae_greedy_decode_workspace_bytes(B, E, H, n_layers, n_split, V) bytes, 16-B aligned
 * ------------------------------------------------------------------------------------------- */
#define DSSVAE_DECODE_MAX_LAYERS 4
typedef struct {
    const float* embedding; int64_t ld_embedding; int64_t n_embeddings;   /* [n_embeddings, E] */
    const float* w_ih[DSSVAE_DECODE_MAX_LAYERS];   /* [3H, E] for layer 0, [3H, H] above ([4H, .] for the LSTM loop) */
    const float* w_hh[DSSVAE_DECODE_MAX_LAYERS];   /* [3H, H] ([4H, H]) */
    const float* b_ih[DSSVAE_DECODE_MAX_LAYERS];
    const float* b_hh[DSSVAE_DECODE_MAX_LAYERS];
    const float* w_out; int64_t ld_out; const float* b_out;                /* [V, H+E], [V] or NULL */
    int32_t E, H, V, n_layers;
} dssvae_decode_params;
int64_t dssvae_greedy_decode_workspace_bytes(int B, int E, int H, int n_layers, int n_split, int V);
int dssvae_greedy_decode(const dssvae_decode_params* params, const float* h0, int B, int max_len,
                         int64_t sos_id, int n_split, int64_t* tokens, float* h_final, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* The same loop for an nn.LSTM decoder (rnn_cell = "lstm": decoder/rnn_decoder.py:43,65,160-168): params' w_ih / w_hh
 * are [4H, .] (gate order i, f, g, o), the state is the pair (h0, c0) [n_layers,B,H] each; h_final / c_final may be
 * NULL.  workspace: dssvae_greedy_decode_lstm_workspace_bytes(...) bytes, 16-B aligned. */
int64_t dssvae_greedy_decode_lstm_workspace_bytes(int B, int E, int H, int n_layers, int n_split, int V);
int dssvae_greedy_decode_lstm(const dssvae_decode_params* params, const float* h0, const float* c0, int B,
                              int max_len, int64_t sos_id, int n_split, int64_t* tokens, float* h_final,
                              float* c_final, void* workspace, int64_t workspace_bytes, void* stream);
/* Live timing of the loop's dominant launch (projection + arg-max over the vocabulary) for the roofline report: arm
 * = 1 brackets every such launch with HIP events on the launch stream, arm = 0 stops; _read waits for the events and
 * returns their summed duration and count.  No reference counterpart (measurement only; not thread-safe). */
int dssvae_decode_profile(int arm);
int dssvae_decode_profile_read(double* total_ms_host, int* launches_host);

/* dssvae_greedy_decode without a workspace argument and without any allocation.  NO COUNTERPART IN THE
 * REFERENCE: it exists for callers that cannot call hipMalloc (code under HIP graph capture, tight serving
 * loops, harnesses that forbid allocation).  The loop's workspace is a file-scope device array inside the
 * library (DSSVAE_DECODE_NOALLOC_BYTES per GPU the library runs on, held until the process ends); a shape whose
 * dssvae_greedy_decode_workspace_bytes(...) exceeds it returns DSSVAE_ETOOBIG with nothing launched (16 MiB
 * covers a decode batch of 128 at E=300, H=600, 2 layers, V=30000).  Same results as dssvae_greedy_decode.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams) race.
 * Under HIP graph capture: make ONE eager call on the device first (any shape) -- the array's address is taken with
 * hipGetSymbolAddress and the kernels' code objects are loaded on first use, and neither is legal while capturing. */
#define DSSVAE_DECODE_NOALLOC_BYTES (16u << 20)
int dssvae_greedy_decode_noalloc(const dssvae_decode_params* params, const float* h0, int B, int max_len,
                                 int64_t sos_id, int n_split, int64_t* tokens, float* h_final, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser over flat fp32 buffers (bin/train.py:124-126, utils/model_utils.py:195):
 * sumsq_out[0] += sum(g^2), added up in a fixed order (block partials in `partials`, then one block): the same
 * bits on every data-parallel replica;  Adam step with bias correction and the clip coefficient
 * min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)) read on device (no host sync).  skip_flag (may be NULL): a device
 * int32; when it is non-zero at execution time the whole update is dropped (the caller points it at the sticky
 * error word of its persistent GRU launches, so a step whose recurrence timed out never reaches the parameters).
 * ------------------------------------------------------------------------------------------- */
#define DSSVAE_SUMSQ_PARTIALS 2048 /* floats of caller-owned scratch for dssvae_sumsq */
int dssvae_sumsq(const float* g, int64_t n, float* partials, float* sumsq_out, void* stream);
int dssvae_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                     float beta2, float eps, int step, const float* sumsq, float max_norm,
                     const float* grad_scale_mask, const int32_t* skip_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-step scalars read ON THE DEVICE, so that a captured HIP graph of the whole training step
 * (bin/train.py:84-88,119-126) can be replayed step after step: the values that change from one
 * step to the next -- the RNG stream of the dropout masks, the annealed word-dropout rate
 * (vae/enhance_syntax_vae.py:292 `step_unk_rate`), the KL weight (vae/base_vae.py:192-196),
 * Adam's learning rate and bias corrections (utils/model_utils.py:195) -- live in a small device
 * buffer that ONE launch per step rewrites (dssvae_set_words: the words travel BY VALUE in the
 * kernel arguments, so the host array may be reused as soon as the call returns), and the *_dyn
 * entry points below take device pointers into it beside the host values of their plain
 * counterparts (NULL pointer = the host value alone, i.e. exactly the plain entry point).
 * No reference counterpart for the mechanism; each *_dyn cites its plain twin's call site.
 * ------------------------------------------------------------------------------------------- */
#define DSSVAE_STEP_WORDS 128
int dssvae_set_words(uint32_t* dst, const uint32_t* words_host, int n, void* stream);
/* seed used = seed + (seed_base ? *seed_base : 0)  (64-bit wrap-around) */
int dssvae_dropout_mask_dyn(float* mask, int64_t n, float p, uint64_t seed, const uint64_t* seed_base, void* stream);
int dssvae_dropout_apply_dyn(const float* x, int64_t n, float p, uint64_t seed, const uint64_t* seed_base, float* y,
                             void* stream);
/* rate used = rate_dev ? *rate_dev : rate */
int dssvae_unk_replace_dyn(const int64_t* ids, int64_t n, const float* rand, float rate, const float* rate_dev,
                           uint64_t seed, const uint64_t* seed_base, int64_t sos, int64_t pad, int64_t eos,
                           int64_t unk, int64_t* out, void* stream);
/* coefficient used = coef[j*n_in + k] + (dyn ? coef_dyn[j*n_in + k] * *dyn : 0): the rows of the loss dictionary that
 * carry the KL weight ('KL Item', 'ELBO', 'Loss') */
int dssvae_weighted_sums_fwd_dyn(const float* const* in, const int32_t* len, int n_in, const float* coef,
                                 const float* coef_dyn, const float* dyn, int n_out, float* out, void* stream);
int dssvae_weighted_sums_bwd_dyn(float* const* din, const int32_t* len, int n_in, const float* coef,
                                 const float* coef_dyn, const float* dyn, int n_out, const float* const* dout,
                                 void* stream);
/* hyper (device, 3 floats): lr, 1 - beta1^step, sqrt(1 - beta2^step) */
int dssvae_adam_step_dyn(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                         float beta2, float eps, const float* sumsq, float max_norm, const int32_t* skip_flag,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Minibatches cut on the device from a dataset that was tensorised ONCE (train.bin as flat ids +
 * offsets resident in HBM; replaces the per-step Python of utils/nn_funcs.py:143-174 `to_input_variable` /
 * :122-129 `padding_input` / :43-47 `word2id` and structs/dataset.py:64-75):
 *   flat_ids [sum of lengths] int64, offsets [n_examples + 1] int64, idx [B] int64 = the examples of the batch;
 *   out [B, W] int64: row b = [sos] ids of example idx[b] [eos] pad..  (boundary = 1) or ids pad.. (boundary = 0),
 *   cut at W; lengths_out [B] int32 (may be NULL) = the examples' own lengths (without boundary symbols).
 * dssvae_keep_rows: the flattened [B, T] positions (b*T + t, t < min(lengths[b] + 1, T)) whose teacher-forced target
 * is not <pad>, in order, into keep [n_keep] (n_keep known to the caller from the lengths; positions beyond the real
 * count are filled with the last real position).  B <= 8192.
 * ------------------------------------------------------------------------------------------- */
int dssvae_batch_rows(const int64_t* flat_ids, const int64_t* offsets, const int64_t* idx, int B, int W, int boundary,
                      int64_t sos, int64_t eos, int64_t pad, int64_t* out, int32_t* lengths_out, void* stream);
int dssvae_keep_rows(const int32_t* lengths, int B, int T, int64_t* keep, int n_keep, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DSSVAE_HIP_H */
