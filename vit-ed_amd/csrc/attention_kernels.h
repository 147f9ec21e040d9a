// Internal argument block + launchers behind vited_attention_fwd / vited_attention_bwd.
#pragma once
#include "common.h"

struct AttnArgs {
    const void *q, *k, *v;
    const int64_t* kv_index;   // forward only, nullable: batch item b reads k / v of batch item kv_index[b]
    int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts;
    const void* o;    // forward: output (written); backward: saved output
    const void* d_o;  // backward only
    int64_t o_bs, o_ts;
    float* lse;    // [B, H, Nq]
    float* delta;  // [B, H, Nq] backward scratch
    void *dq, *dk, *dv;
    int64_t dq_bs, dq_ts, dk_bs, dk_ts, dv_bs, dv_ts;
    int64_t batch, nq, nk;
    int heads, head_dim;
    float scale;
};

// portable fp32-VALU kernels (attention_portable.hip)
int attention_fwd_portable(const AttnArgs& a, int dtype, hipStream_t s);
int attention_bwd_portable(const AttnArgs& a, int dtype, hipStream_t s);

// bf16 MFMA kernels (attention_mfma.hip)
bool attention_mfma_supported(const AttnArgs& a, bool backward);
int attention_fwd_mfma(const AttnArgs& a, hipStream_t s);
int attention_bwd_mfma(const AttnArgs& a, hipStream_t s);

// tiled online-softmax kernels for long sequences (attention_flash.hip)
int attention_fwd_flash(const AttnArgs& a, hipStream_t s);
int attention_bwd_flash(const AttnArgs& a, hipStream_t s);
