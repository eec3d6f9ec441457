// Argument block of the varlen attention backward kernels (attn_bwd.hip: every dtype / head size / mask / dropout; attn_bwd64w.hip: the
// one-wave-per-SIMD bf16 d_h = 64 form of the training steps).
#pragma once
#include "common.h"

struct BwdArgs {
    const void *q, *k, *v, *o, *dout;
    void *dq, *dk, *dv;
    const float *lse, *delta;  // [H][total_q]
    const int32_t *cu_q, *cu_k;
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, H, dh, causal, total_q;
    float scale_log2e, scale;
    uint32_t drop_thr, drop_seed;  // the forward's attention-probability dropout, regenerated element-wise
    float drop_scale;
    int accum_dkv; // 1: dk, dv += instead of = (bf16, aligned operands only): a second pass over the same K / V adds its gradient in the kernel's epilogue
    int tail256;   // (attn_bwd1p.hip: != 0 = XCD-aware block order)  attn_bwd.hip one-block kernels: > 0 = cover only the rows past each sequence's last full 256-row block (the full blocks belong to attn_bwd64w.hip)
    int nblk;      // attn_bwd64w.hip / attn_bwd1p.hip: key or query blocks per (sequence, head) of their one-dimensional, XCD-swizzled grids
};

// bf16, d_h = 64 exactly, q prescaled, no dropout, no causal mask, 16-byte aligned operands: the full 256-query blocks of dQ (which also
// publishes -delta for every row of those blocks) and the full 256-key blocks of dK / dV (attn_bwd64w.hip)
void acai_attn_bwd64w_dq_launch(const BwdArgs &a, int B, int max_q, hipStream_t st);
void acai_attn_bwd64w_dkv_launch(const BwdArgs &a, int B, int max_k, int equal_len, hipStream_t st);   // equal_len: every sequence has max_k keys (XCD-aware block order)

// bf16, d_h = 32 exactly, q prescaled, no dropout, no causal mask, no accumulation, aligned operands: dQ, dK, dV in one pass over the scores
// (attn_bwd1p.hip; a sequence's full 512-key blocks in one launch, the keys past them - partial_blocks != 0 - in a second).  workspace:
// acai_attn_bwd1p_workspace(total_q, H) bytes of device memory (the fp32 query gradient the key blocks add into), used only inside the call.
size_t acai_attn_bwd1p_workspace(int total_q, int H);
void acai_attn_bwd1p_launch(const BwdArgs &a, int B, int max_k, int partial_blocks, int equal_len, void *workspace, hipStream_t st);
