/*
 * acai_omr_hip.h - C ABI of the MI355X (gfx950) backend for the acai-omr model hot path.
 *
 * The reference (jsnchon/acai-omr) is pure Python on stock PyTorch: it has no FFI of its own.
 * Each entry point below therefore replaces the stock ATen op sequence behind one reference
 * call site (file:line cited per function; acai_omr/models/models.py = M, kv_caching.py = K).
 * Callers own every buffer (PyTorch-ROCm tensors in practice); nothing here allocates, frees,
 * synchronises or reads device memory on the host, so every call is hipGraph-capturable.
 * All launches go to the hipStream_t passed as `stream` (void* in this header so that plain C /
 * ctypes callers need no HIP headers).  Return value: 0 = ok, negative = argument error,
 * positive = hipError_t; acai_last_error() gives a message.  No C++ exception crosses the ABI.
 *
 * dtypes: ACAI_F32 = fp32 storage + exact-fp32 MFMA (v_mfma_f32_32x32x2_f32);
 *         ACAI_BF16 = bf16 storage + bf16 MFMA with fp32 accumulate.
 * The residual stream, LayerNorm, softmax statistics, biases and logits are always fp32.
 */
#ifndef ACAI_OMR_HIP_H
#define ACAI_OMR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ACAI_ABI_VERSION 1
#define ACAI_F32 0
#define ACAI_BF16 1

/* gemm epilogue flags */
#define ACAI_GEMM_GELU 1       /* exact-erf GELU after bias (M:31 activation="gelu", M:657 nn.GELU) */
#define ACAI_GEMM_ROUND_BF16 2 /* round (acc + bias) to bf16 first: restates autocast's bf16 linear output */

int acai_version(void);
const char *acai_last_error(void);

/* nn.LayerNorm (M:33, eps 1e-6 final norms; torch TransformerEncoderLayer norm1/2/3 eps 1e-5).
 * y = LN(x) * w + b over the last dim; out_f32 and/or out_bf16 may be NULL. */
int acai_layernorm_fwd(const float *x, const float *w, const float *b, float eps, float *out_f32, void *out_bf16,
                       int rows, int dim, void *stream);

/* nn.Linear / F.linear (M:29,57,204,205,428,655-660; K:193,215,244):
 * C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]) (+ residual[M,N]); A, W have dtype `in_dtype`,
 * C has `out_dtype`; bias and residual are fp32 (NULL = absent). */
int acai_gemm_nt(const void *A, int lda, const void *W, int ldw, const float *bias, const float *residual, int ldr,
                 void *C, int ldc, int M, int N, int K, int in_dtype, int out_dtype, int flags, void *stream);
/* acai_gemm_nt with an auxiliary [M][N] operand of C's dtype (the MLP of nn.TransformerEncoderLayer / DecoderLayer in training:
 * linear1 -> GELU -> linear2, acai_omr/models/models.py:30-34,186-190,422-426 and their autograd):
 *   aux_mode 1 (with ACAI_GEMM_GELU): aux receives the pre-activation (bias added, bf16-rounded if asked), C its GELU - the forward keeps both;
 *   aux_mode 2: C = round(A.W^T) * gelu'(aux) - the dX GEMM of linear2 multiplies by the GELU derivative of the saved pre-activation;
 *   aux_mode 3 (with ACAI_GEMM_GELU): as 1, but aux receives gelu'(pre-activation) - the forward epilogue holds Phi(-|a|) for the GELU anyway;
 *   aux_mode 4: C = round(A.W^T) * aux - with 3, the backward epilogue is one multiply per element (what the training steps use since round 4;
 *               the derivative is rounded to C's dtype once more than in the 1 / 2 pair: 2^-9 relative in bf16, nothing in fp32).
 * scale_cols > 0: columns [0, scale_cols) of (A.W^T + bias) are multiplied by col_scale before rounding - the in-projection of
 * nn.MultiheadAttention hands q to the attention kernels as q * log2(e) / sqrt(dh) (acai_attn_varlen_fwd, q_prescaled). */
int acai_gemm_nt_ex(const void *A, int lda, const void *W, int ldw, const float *bias, const float *residual, int ldr,
                    void *C, int ldc, void *aux, int ldaux, int aux_mode, int M, int N, int K, int in_dtype, int out_dtype, int flags,
                    int scale_cols, float col_scale, void *stream);
/* Testing / tuning aid: pin the row-major GEMM kernel (0 auto; 1 128x128 two-stage; 2 256x128 two-stage; 3 256x128 three-stage; 4 256x128
 * persistent three-stage ring; 5 256x256 two-stage; 6 persistent 256x256 ring of half-stages; 7 ping-pong wave groups with the register
 * epilogue; 8 = 7 with the GELU forms' deferred epilogue).  A pinned variant still falls back when the shape cannot use it.  No reference counterpart. */
int acai_gemm_set_variant(int variant);

/* The same contraction with either operand stored reduction-major, for the backward of nn.Linear (autograd of M:29,57,...):
 *   dX = dY . W   -> acai_gemm(dY, ldy, 0,  W, ldw, 1, ...)  (M = rows, N = in_features, K = out_features)
 *   dW = dY^T . X -> acai_gemm(dY, ldy, 1,  X, ldx, 1, ...)  (M = out_features, N = in_features, K = rows)
 * trans_a: A stored [K][M]; trans_w: W stored [K][N].  residual may alias C.
 * trans_a && trans_w (weight gradient, K = number of rows): C must be fp32 and is ACCUMULATED into with fp32 atomics
 * (split-K over workgroups); zero it for a fresh gradient.  No bias / residual / flags in that form. */
int acai_gemm(const void *A, int lda, int trans_a, const void *W, int ldw, int trans_w, const float *bias, const float *residual, int ldr,
              void *C, int ldc, int M, int N, int K, int in_dtype, int out_dtype, int flags, void *stream);
/* Both parameter gradients of an nn.Linear from one pass over the output gradient (autograd of F.linear: torch's mm + sum):
 * dW[M][N] += dY[K][M]^T X[K][N] and, if db != NULL, db[M] += sum over the K token rows of dY.  fp32 accumulators (zero them for fresh
 * gradients), bf16 or fp32 operands.  M = out_features, N = in_features, K = token rows. */
int acai_gemm_dw(const void *dY, int ldy, const void *X, int ldx, float *dW, int lddw, float *db, int M, int N, int K, int dtype, void *stream);

/* MemoryCache.cache_memory_keys_and_vals (K:235-253): KV = mem . W_kv^T + b_kv with W_kv = rows E..3E of the
 * cross-attention in_proj; written head-major and ragged for the decode kernels:
 * k_out[seq_off[b] + (h*len[b] + s)*dhp + d], same for v_out; row_seq/row_pos give (b, s) of each memory row. */
int acai_cross_kv_prefill(const void *mem, int ldm, const void *Wkv, int ldw, const float *bkv, const int32_t *row_seq,
                          const int32_t *row_pos, const int64_t *seq_off, const int32_t *seq_len, void *k_out, void *v_out,
                          int M, int E, int H, int dh, int dhp, int dtype, int flags, void *stream);

/* nn.Unfold(P, stride P) on one (1,H,W) fp32 image, transposed to rows of P*P pixels (M:23,48-52);
 * rows are written starting at out + row0*ld (dtype `out_dtype`). */
int acai_patchify(const float *img, int H, int W, int P, void *out, int ld, int row0, int out_dtype, void *stream);

/* DynamicResize / PatchDivisibleResize resize step (acai_omr/utils/utils.py:325-330 `v2.Resize(...)`, :351-356 `F.resize(img, size,
 * BICUBIC, antialias=True)` on a float32 C x H x W tensor = aten `_upsample_bicubic2d_aa`, align_corners = False), optionally followed by
 * DynamicResize's `.clamp(0.0, 1.0)` (:367).  img [C][H][W] -> out [C][OH][OW], all fp32 contiguous; tmp holds C*H*OW floats (the
 * width pass).  C*H, OH and C <= 65535. */
int acai_resize_bicubic_aa(const float *img, int C, int H, int W, float *tmp, float *out, int OH, int OW, int clamp01, void *stream);

/* The same resize of ONE grayscale image written straight into the packed patch stream the encoder's projection GEMM reads (SURVEY 8f-2:
 * `DynamicResize` -> `Encoder.batchify`'s Unfold, acai_omr/utils/utils.py:334-367 + acai_omr/models/models.py:48-52, without the image tensor
 * in between): img is fp32 [H][W] in [0, 1], or uint8 (in_u8 = 1) scaled by 1/255 on load (`v2.ToDtype(torch.float32, scale=True)`,
 * acai_omr/train/pre_train.py:56); the crop window (top, left, ch, cw) of the OH x OW result (DynamicResize's centre crop, :360-364; the whole
 * image: 0, 0, OH, OW; ch, cw multiples of P) becomes rows row0 .. row0 + (ch/P)(cw/P) - 1 of `patches` [rows][ld >= P*P] in `out_dtype`
 * (fp32 / bf16), row (y/P)(cw/P) + x/P, column (y%P) P + x%P as nn.Unfold(P, stride P) orders them.  tmp holds H*OW floats. */
int acai_resize_to_patches(const void *img, int in_u8, int H, int W, float *tmp, void *patches, int ld, int row0, int OH, int OW, int top,
                           int left, int ch, int cw, int P, int out_dtype, int clamp01, void *stream);

/* out[i,:] = table[idx[i],:] (+ add[i,:]) : pos_embedding slices (M:50), nn.Embedding (M:460),
 * MAE shuffle / restore index_select (M:114,123,229). table/out fp32. */
int acai_gather_rows(const float *table, const int32_t *idx, const float *add, float *out, int rows, int dim, void *stream);

/* OMREncoder.interpolate_pe (M:291-302): F.interpolate(pos_embedding (Hin,Win,E) as NCHW, size=(Hout,Wout), mode="bilinear",
 * align_corners=False), result (Hout*Wout, E) row-major - aten's upsample_bilinear2d arithmetic (source index (dst + 0.5) * in/out - 0.5
 * clamped at 0, neighbour clamped at in-1).  bwd: dtable[Hin*Win, E] += the transposed stencil applied to dout (float atomics; dtable is
 * NOT zeroed here). */
int acai_pe_interp_fwd(const float *table, int Hin, int Win, int E, float *out, int Hout, int Wout, void *stream);
int acai_pe_interp_bwd(const float *dout, int Hout, int Wout, int E, float *dtable, int Hin, int Win, void *stream);

/* Diagnostic aid (tools/stamp_decode.py): s_memrealtime stamps of the decode GEMV kernel's stages, buf[launch][1024][8] uint64. */
int acai_debug_stamps(void *buf, int cap_launches);

/* Hardware-assumption probe (tests/test_gpu_kernels.py::test_lds_dma_out_of_range_lanes_write_zeros): one wave issues the weight-gradient
 * GEMMs' `buffer_load_dwordx4 ... offen lds` (gemm_tn_glds / gemm_tn_pp kernels, ragged last token tile) over a 1 KiB LDS image preset to
 * 0xFFFFFFFF with a resource of `valid_bytes` bytes at `src`; out[256] receives the image.  The kernels rely on lanes past num_records
 * writing ZEROS (observed on gfx950 / ROCm 7.2, not documented): a ROCm change shows up as a red test instead of wrong gradients. */
int acai_debug_lds_dma_oob(const void *src, int valid_bytes, void *out, void *stream);

/* autocast's fp32 -> bf16 input cast (round to nearest even) for an activation that feeds a bf16 GEMM. */
int acai_cast_f32_bf16(const float *x, void *y, int64_t n, void *stream);

/* F.scaled_dot_product_attention on packed ragged streams (torch nn.MultiheadAttention inside
 * nn.TransformerEncoderLayer M:30-34,186-190 and nn.TransformerDecoderLayer M:422-426).
 * q row i of sequence b is q + (cu_q[b]+i)*ldq + h*dh; same for k, v (cu_k) and out.
 * causal != 0 applies the triu(diagonal=1) mask of M:468.  dh <= 64.
 * lse (optional, [H][total_q] fp32): log2-domain log-sum-exp of the scaled scores, saved for acai_attn_varlen_bwd.
 * dropout_p > 0: attention-probability dropout (nn.MultiheadAttention(dropout=p) in train mode); the keep mask is a counter-based
 * hash of (dropout_seed, head, query, key) that the backward regenerates.
 * q_prescaled != 0: q already carries the softmax scale in the log2 domain, q' = q * log2(e) / sqrt(dh) - applied by the in-projection's
 * epilogue (acai_gemm_nt_ex scale_cols / col_scale) before its one rounding, as torch's math SDPA applies the scale to q before the
 * product - so K . q' is the exponent itself and the kernel spends no multiply per score.  Needs 16-byte aligned operands. */
int acai_attn_varlen_fwd(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, void *out, int ldo,
                         const int32_t *cu_q, const int32_t *cu_k, int B, int H, int dh, int max_q, int causal,
                         int dtype, float *lse, int total_q, float dropout_p, uint32_t dropout_seed, int q_prescaled, void *stream);

/* Backward of acai_attn_varlen_fwd (autograd of the same SDPA; training loops pre_train.py:59, omr_teacher_force_train.py:118).
 * o / lse are the forward's outputs, dout the incoming gradient; dq/dk/dv take the layout of q/k/v (own row strides).
 * delta: workspace [H][total_q] floats.  Deterministic (no atomics): S and P are recomputed per kernel.
 * q_prescaled != 0: q is the forward's q' (see there); dq is still the gradient with respect to the UNSCALED in-projection output
 * (what the in-projection's backward GEMMs consume), dk and dv are unchanged in meaning.
 * `causal`: bit 0 = the causal mask; bit 1 (round 4) = dk, dv += instead of = (bf16, 16-byte aligned operands): when two passes attend to ONE stored
 * K / V (ScheduledSamplingViTOMR.forward_train's two decoder passes over the same memory, models.py:822-834) the second pass's backward adds
 * its gradient in the kernel's epilogue - fp32 add, one rounding - instead of autograd summing two [keys, 2E] tensors afterwards. */
int acai_attn_varlen_bwd(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, const void *o, int ldo,
                         const void *dout, int lddo, void *dq, int lddq, void *dk, int lddk, void *dv, int lddv, const float *lse,
                         float *delta, const int32_t *cu_q, const int32_t *cu_k, int B, int H, int dh, int max_q, int max_k,
                         int total_q, int causal, int dtype, float dropout_p, uint32_t dropout_seed, int q_prescaled, void *stream);
/* The same backward with a caller-lent device workspace (round 4).  With bf16, d_h = 32, q_prescaled, no dropout / mask / accumulation and
 * max_q, max_k >= 512 - the MAE decoder's self-attention (models.py:186-190), ragged batches included - dQ, dK and dV come from ONE pass over
 * the scores (attn_bwd1p.hip): the key blocks add their part of a query's gradient to the fp32 workspace with float atomics, so dQ is
 * reproducible to fp32 rounding, not bit for bit (ACAI_ATTN_BWD_1P=0 in the environment, or no workspace, keeps the two-kernel form, which
 * is).  total_k = the rows of k / v (0 = unknown: when it equals B * max_k with max_k % 512 == 0 the launch for partial key blocks is left out).  acai_attn_varlen_bwd_workspace_bytes: bytes that form needs for a call with these
 * arguments, 0 when it does not apply (then any workspace is ignored).  The workspace is used only inside the call (stream order). */
size_t acai_attn_varlen_bwd_workspace_bytes(int B, int H, int dh, int max_q, int max_k, int total_q, int total_k, int causal, int dtype,
                                            float dropout_p, int q_prescaled);
int acai_attn_varlen_bwd_ws(const void *q, int ldq, const void *k, int ldk, const void *v, int ldv, const void *o, int ldo,
                            const void *dout, int lddo, void *dq, int lddq, void *dk, int lddk, void *dv, int lddv, const float *lse,
                            float *delta, const int32_t *cu_q, const int32_t *cu_k, int B, int H, int dh, int max_q, int max_k,
                            int total_q, int total_k, int causal, int dtype, float dropout_p, uint32_t dropout_seed, int q_prescaled,
                            void *workspace, size_t workspace_bytes, void *stream);

/* Backward of nn.LayerNorm: dx (fp32) from x, w, dy; dw/db (both or neither NULL) are ACCUMULATED with fp32 atomics (zero or seed
 * them); dx_bf16 (may be NULL; needs dim % 256 == 0, dim <= 1024): bf16 copy of dx for the GEMM that consumes it; dxsum (may be NULL, same
 * condition; ACCUMULATED): column sums of dx as that GEMM sees it = the consuming nn.Linear's bias gradient; stats: workspace [rows][2]. */
int acai_layernorm_bwd(const float *x, const float *w, const float *dy, float eps, float *dx, void *dx_bf16, float *dw, float *db, float *dxsum,
                       float *stats, int rows, int dim, void *stream);
/* exact-erf GELU as a stand-alone pass (training keeps the pre-activation) and its derivative: da = dh * gelu'(a). */
int acai_gelu_fwd(const void *a, void *h, int64_t n, int dtype, void *stream);
int acai_gelu_bwd(const void *a, const void *dh, void *da, int64_t n, int dtype, void *stream);
/* out[c] += sum_r x[r,c] (bias gradients); out fp32, accumulated with atomics. */
int acai_colsum(const void *x, int ld, float *out, int rows, int cols, int dtype, void *stream);
/* dst[idx[r],:] += src[r,:] (gradients of nn.Embedding M:460, pos_embedding slices M:50, MAE index_select M:114,229).
 * shared_row >= 0: the caller guarantees that only that index occurs more than once (the MAE mask token) - the other rows are then
 * updated without atomics; shared_row < 0: every row through fp32 atomics. */
int acai_scatter_add_rows(const float *src, const int32_t *idx, float *dst, int rows, int dim, int shared_row, void *stream);
/* nn.Dropout on a projection output followed by the residual add (torch TransformerEncoderLayer dropout1/dropout2, decoder dropout1-3,
 * transition head M:658): out = residual + keep * x / (1 - p); residual may be NULL (plain dropout, and its own backward on dy).
 * keep mask = counter-based hash of (seed, row, col). */
int acai_dropout_add(const void *x, const float *residual, void *out, int rows, int cols, float p, uint32_t seed, int x_dtype, int out_dtype,
                     void *stream);
/* MAELoss (M:271-288) forward + backward on packed rows: *loss += sum_r mask_r * mean_d((pred - that)^2) * inv_count,
 * dpred (may be NULL) = d loss / d pred; that = (target - mean) / sqrt(var_unbiased + 1e-6). */
int acai_mae_loss(const float *pred, const float *target, const unsigned char *mask, float inv_count, float *loss, float *dpred,
                  int rows, int dim, void *stream);
/* OMRCELoss (M:784-796) forward + backward: mean cross entropy over rows whose target != ignore_index, with nn.CrossEntropyLoss's
 * label_smoothing (M:786-788; 0 = plain NLL). */
int acai_ce_loss(const float *logits, int ld, const int64_t *target, int ignore_index, float inv_count, float label_smoothing, float *loss,
                 float *dlogits, int rows, int V, void *stream);

/* Fused multi-tensor AdamW: one launch steps every parameter tensor (reference: torch.optim.AdamW in acai_omr/train/pre_train.py:105 and
 * omr_teacher_force_train.py:207 over the param groups of acai_omr/models/models.py:761-781; the cosine/warm-up schedule of
 * acai_omr/utils/utils.py:204-222 only changes `lr`).  All tables live in DEVICE memory.  tensors[i]: fp32 parameter, gradient and the two
 * moment buffers (n elements) + the index of its hyper-parameter group; groups[j]: this step's lr, betas, eps, weight decay and bias
 * corrections bc1 = 1 - beta1^t, sqrt(bc2) = sqrt(1 - beta2^t).  chunk_tensor / chunk_off: one entry per workgroup = (tensor, first element)
 * of a chunk of chunk_elems (multiple of 4) elements.  grad_scale multiplies every gradient on load (loss-scale / accumulation mean; 1 = none). */
typedef struct AcaiAdamWTensor {
    float *p;
    const float *g;
    float *m, *v;
    int64_t n;
    int32_t group, pad_;
    float bias_c1, bias_c2_sqrt; /* 1 - beta1^t, sqrt(1 - beta2^t) with THIS tensor's step count t (torch keeps `step` per parameter) */
} AcaiAdamWTensor;
typedef struct AcaiAdamWGroup {
    float lr, beta1, beta2, eps, weight_decay, pad0_, pad1_, pad2_;
} AcaiAdamWGroup;
int acai_adamw_step(const AcaiAdamWTensor *tensors, const AcaiAdamWGroup *groups, const int32_t *chunk_tensor, const int64_t *chunk_off,
                    int n_chunks, int chunk_elems, float grad_scale, void *stream);

/* The operand copies autocast makes of the fp32 master weights (torch casts each nn.Linear weight / bias to bf16 on every call under
 * torch.autocast, omr_teacher_force_train.py:112-116; this path caches them per parameter version) for ALL parameters in one launch, after an
 * optimizer step: per entry any of the bf16 copy [rows][cols], the transposed bf16 copy [cols][rows] (the dX GEMM's row-major operand) and
 * the bf16-rounded fp32 copy (biases).  `table` is device memory; tile0 = running sum of ceil(rows/64) * ceil(cols/64) over the entries
 * before this one, n_tiles the total.  src and the destinations are contiguous; destinations are 8-byte aligned. */
typedef struct AcaiCastEntry {
    const float *src;
    void *dst16, *dst16t;
    float *dst32r;
    int32_t rows, cols, tile0, pad_;
} AcaiCastEntry;
int acai_cast_weights(const AcaiCastEntry *table, int n_entries, int n_tiles, void *stream);

/* ---- KV-cached greedy decode (K:190-223, K:292-302, M:518-528, M:575-583) ------------------------------- */
typedef struct {
    const void *self_in_w;   const float *self_in_b;   /* self_attn.in_proj_{weight,bias} [3E,E] */
    const void *self_out_w;  const float *self_out_b;  /* self_attn.out_proj */
    const void *cross_q_w;   const float *cross_q_b;   /* rows 0..E of multihead_attn.in_proj (K:212-213) */
    const void *cross_out_w; const float *cross_out_b; /* multihead_attn.out_proj */
    const void *lin1_w;      const float *lin1_b;      /* linear1 [F,E] */
    const void *lin2_w;      const float *lin2_b;      /* linear2 [E,F] */
    const float *n1_w, *n1_b, *n2_w, *n2_b, *n3_w, *n3_b;
    void *k_self, *v_self;              /* KVCache (K:35-41) as [Bmax][H][Tmax][dhp] */
    const void *k_cross, *v_cross;      /* acai_cross_kv_prefill output */
} AcaiDecLayer;

typedef struct {
    int32_t B, E, H, dh, dhp, F, V, L, Tmax, dtype, flags, max_len;
    int32_t self_chunk, cross_chunk;    /* keys per attention workgroup */
    int32_t self_nsplit, cross_nsplit;  /* workgroups per (b, h) */
    int32_t bos, pad, eos;
    int32_t cross_group;   /* > 1: every `cross_group` consecutive rows share one memory (GRPO rollouts of one image): its K/V is streamed once per group */
    const AcaiDecLayer *layers;         /* host array of L entries */
    const float *emb;                   /* vocab_embedding.weight [V,E] fp32 */
    const float *pos;                   /* decoder pos_embedding [Tmax,E] fp32 */
    const float *fn_w, *fn_b;           /* decoder_blocks.norm (eps 1e-6) */
    const void *unembed_w;              /* [V,E] in `dtype` */
    const float *unembed_b;
    const int64_t *cross_off;           /* [B] element offset of sequence b in k_cross / v_cross */
    const int32_t *cross_len;           /* [B] memory length S_b */
    int64_t *seqs;                      /* [B,max_len] token ids, seqs[:,0] = <bos> (M:568-569) */
    float *logprobs;                    /* [B,max_len] (M:570) */
    int32_t *step;                      /* device scalar t: next position to fill (starts at 1) */
    int32_t *finished;                  /* [B] flags + [B] = count of unfinished rows after the step */
    float *x, *xn, *qkv, *attn, *proj, *hid, *logits, *partial; /* workspaces, see DESIGN.md */
    uint32_t *tickets;                  /* B*H zeroed arrival counters for the in-launch split merge; NULL = separate combine launch */
    float *stats;                       /* 6*B floats: published LayerNorm (mean, rstd) rows; NULL disables the fused-LN path */
} AcaiDecoder;

/* Input of the FIRST step after the device-side loop state was armed: x = vocab_embedding[seqs[:, t-1]] + pos_embedding[t], t = step[0]
 * (M:521-524 with quirk Q1).  Every later step's input is written by the previous step's argmax / sampling kernel. */
int acai_decode_embed(const AcaiDecoder *dec, void *stream);
/* One greedy step t = *step for all B rows: input x = embedding of seqs[:,t-1] at pos_embedding[t] (quirk Q1, M:576), 12x cached_forward,
 * final norm, unembed, argmax + log_softmax gather, seqs[:,t] / logprobs[:,t] update, finished flags, ++*step.  Enqueues only kernels:
 * capture it in a hipGraph and replay.
 * CONTRACT (E % 4 == 0): the step does NOT embed its own input - dec->x must hold it: written by acai_decode_embed after arming, and by every
 * step's argmax / sampling kernel for the next one.  acai_decode_logits and acai_decode_hidden overwrite dec->x: after either, call
 * acai_decode_embed again before the next acai_decode_step / acai_decode_sample_step.  ENFORCED since round 4: the library records per
 * decoder state (keyed by dec->x) whether x holds a chained step's input - set by acai_decode_embed, kept by the step entry points, cleared
 * by acai_decode_logits / acai_decode_hidden - and acai_decode_step / acai_decode_sample_step return an argument error (rc < 0,
 * acai_last_error() names acai_decode_embed) when it does not.  Replays of a captured graph do not pass through the check.
 * tickets: the in-launch merge is used only while decode_attn_kernel's residency is the one it was validated at (two workgroups per CU,
 * hipOccupancyMaxActiveBlocksPerMultiprocessor); otherwise the step issues the separate combine launch as if tickets were NULL. */
int acai_decode_step(const AcaiDecoder *dec, void *stream);
/* One SAMPLING decode step for every sequence (GRPOViTOMR.cached_forward_rollout_policy, acai_omr/models/models.py:988-1049): as
 * acai_decode_step, but the next token is drawn from softmax(top_k(logits) / temperature) and its log-probability is taken under
 * softmax(top_k(logits)) (models.py:1006-1019).  The draw is the inverse CDF of uniforms[b * max_len + t] over the kept logits in
 * descending order (ties: lower index first), 1 <= top_k <= 64: torch.multinomial's random stream is replaced by caller-supplied uniforms. */
int acai_decode_sample_step(const AcaiDecoder *d, const float *uniforms, int top_k, float temperature, void *stream);
/* The same without the token bookkeeping: logits for caller-supplied tokens/time_step (OMRDecoder.cached_generate). */
int acai_decode_logits(const AcaiDecoder *dec, const int64_t *tokens, int time_step, void *stream);

/* CachedTransformerDecoder.cached_generate (K:292-302): x_in [B,E] fp32 is this step's embedding; the hidden state after
 * the L cached layers and the optional final norm lands in dec->xn.  emb / pos / unembed may be NULL for this call. */
int acai_decode_hidden(const AcaiDecoder *dec, const float *x_in, void *stream);

/* 1 when a decode step with `dec->tickets` set merges the split partials of its attention launches inside those launches (last-arriver
 * hand-off), 0 when it takes the separate combine launch instead (tickets ignored: the hand-off is used only at the workgroup residency it
 * was validated at, or as ACAI_DATTN_MERGE forces), negative for a bad argument.  Same arithmetic either way (kv_caching.py:131 - the SDPA
 * of a cached step); bench.py reports it so that a silent change of path shows in the headline line. */
int acai_decode_merge_in_launch(int dtype, int dhp);

/* F.linear on a (B,1,K) activation (K:193,215; nn.Linear inside cached_forward K:139,222): y[B,N] = x[B,K].W[N,K]^T
 * + bias (+GELU) (+residual); x, y, bias, residual fp32, W in `dtype` (bf16: x is rounded to bf16 first, as autocast does). */
int acai_skinny_gemm(const float *x, int ldx, const void *W, int ldw, const float *bias, const float *residual, int ldr,
                     float *y, int ldy, int B, int N, int K, int dtype, int flags, void *stream);

/* acai_skinny_gemm with the fusions the decode step uses (bf16 weights, K % 256 == 0): x may be bf16 (x_dtype), y may be bf16;
 * ln_w/ln_b: x := LayerNorm(x) on load (norm1/2/3 of the post-LN layer, K:208,220,222), its per-row (mean, rstd) optionally
 * published to stats_out[B][2]; rln_w/rln_b/rstats: residual := LayerNorm(residual) from published statistics. */
int acai_skinny_gemm_ex(const void *x, int ldx, int x_dtype, const void *W, int ldw, const float *bias, const float *residual, int ldr,
                        void *y, int ldy, int y_dtype, int B, int N, int K, int dtype, int flags, const float *ln_w, const float *ln_b,
                        float ln_eps, float *stats_out, const float *rln_w, const float *rln_b, const float *rstats, void *stream);

/* CachedMultiheadAttention.cached_forward's SDPA (K:131-136) for one query per sequence:
 * keys/values of sequence b, head h at kc/vc + seq_off[b] + (h*seq_len[b] + s)*dhp; out[b, h*dh + d] fp32.
 * partial: workspace of B*H*nsplit*(dhp+2) floats; chunk*nsplit must cover max(seq_len).
 * out == NULL stops after the split partials (m, l, o[dhp]) - the streaming kernel alone, for benchmarking.
 * tickets: NULL = a second launch merges the splits; else B*H zeroed counters: the last-arriving workgroup of each (b,h)
 * merges them inside the launch (agent-scope release / acquire hand-off) and re-zeroes its counter. */
int acai_decode_attn(const float *q, int ldq, const void *kc, const void *vc, const int64_t *seq_off, const int32_t *seq_len,
                     float *partial, float *out, int ldo, int B, int H, int dh, int dhp, int chunk, int nsplit, int dtype,
                     int round_out, uint32_t *tickets, void *stream);

/* hipGraph helpers (capture on `stream`, replay). */
int acai_graph_begin(void *stream);
int acai_graph_end(void *stream, void **graph_exec_out);
int acai_graph_launch(void *graph_exec, void *stream);
int acai_graph_destroy(void *graph_exec);

#ifdef __cplusplus
}
#endif
#endif
