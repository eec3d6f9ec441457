// Argument block shared by the varlen attention forward kernels (attn_varlen.hip: every dtype / head size / mask; attn_fwd64.hip: the
// software-pipelined bf16 d_h = 64 form of the training steps).
#pragma once
#include "common.h"

struct AttnArgs {
    const void *q, *k, *v;
    void *out;
    const int32_t *cu_q, *cu_k;
    int ldq, ldk, ldv, ldo, H, dh, causal;
    float scale_log2e;
    uint32_t drop_thr, drop_seed;  // attention-probability dropout (nn.MultiheadAttention(dropout=p) in train mode): keep iff hash >= thr
    float drop_scale;
    float *lse;   // optional [H][total_q]: log2-domain log-sum-exp of the scaled scores (saved for the backward pass)
    int total_q;
    int nqb;      // attn_fwd64w.hip: query blocks per (sequence, head) of its one-dimensional, XCD-swizzled grid
    int tail;     // attn_fwd64*.hip: > 0 = a sequence's last 256-query block goes to attn_fwd64_tail_kernel when it holds <= tail (32) rows
};

// bf16, d_h = 64 exactly, q prescaled, no dropout, no causal mask, 16-byte aligned operands (attn_fwd64.hip)
int acai_attn_fwd64_launch(const AttnArgs &a, int B, int max_q, hipStream_t st);
// the 64-queries-per-wave, one-wave-per-SIMD form (attn_fwd64w.hip); called by acai_attn_fwd64_launch
void acai_attn_fwd64w_launch(const AttnArgs &a, int B, int max_q, hipStream_t st);
