// C-ABI attention entry points: validation + dispatch (bf16 MFMA kernels when the shape is
// covered, portable fp32-VALU kernels otherwise).
#include "attention_kernels.h"

static thread_local int g_last_attn_path = 0;
extern "C" int vited_last_attention_path(void) { return g_last_attn_path; }

static int check_common(const AttnArgs& a, int dtype) {
    if (!a.q || !a.k || !a.v || !a.o || !a.lse) return VITED_ERR_BAD_ARG;
    if (a.batch <= 0 || a.heads <= 0 || a.nq <= 0 || a.nk <= 0 || a.head_dim <= 0) return VITED_ERR_BAD_ARG;
    if (dtype != VITED_F32 && dtype != VITED_BF16) return VITED_ERR_UNSUPPORTED;
    if (a.batch > 65535 || a.heads > 65535) return VITED_ERR_UNSUPPORTED;
    return VITED_OK;
}

extern "C" int vited_attention_fwd(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts,
                                   const void* v, int64_t v_bs, int64_t v_ts, void* o, int64_t o_bs, int64_t o_ts,
                                   float* lse, int dtype, int64_t batch, int heads, int64_t nq, int64_t nk, int head_dim,
                                   float scale, void* stream) {
    return vited_attention_fwd_indexed(q, q_bs, q_ts, k, k_bs, k_ts, v, v_bs, v_ts, nullptr, o, o_bs, o_ts, lse, dtype, batch, heads, nq,
                                       nk, head_dim, scale, stream);
}

extern "C" int vited_attention_fwd_indexed(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts,
                                           const void* v, int64_t v_bs, int64_t v_ts, const int64_t* kv_index, void* o, int64_t o_bs,
                                           int64_t o_ts, float* lse, int dtype, int64_t batch, int heads, int64_t nq, int64_t nk,
                                           int head_dim, float scale, void* stream) {
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v;
    a.kv_index = kv_index;
    a.q_bs = q_bs; a.q_ts = q_ts; a.k_bs = k_bs; a.k_ts = k_ts; a.v_bs = v_bs; a.v_ts = v_ts;
    a.o = o; a.o_bs = o_bs; a.o_ts = o_ts;
    a.lse = lse;
    a.batch = batch; a.heads = heads; a.nq = nq; a.nk = nk; a.head_dim = head_dim; a.scale = scale;
    int rc = check_common(a, dtype);
    if (rc != VITED_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VITED_BF16 && attention_mfma_supported(a, false)) {
        g_last_attn_path = 2;
        return attention_fwd_mfma(a, s);
    }
    g_last_attn_path = 1;
    return attention_fwd_portable(a, dtype, s);
}

extern "C" int vited_attention_bwd(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts,
                                   const void* v, int64_t v_bs, int64_t v_ts, const void* o, const void* d_o, int64_t o_bs,
                                   int64_t o_ts, const float* lse, float* delta, void* dq, int64_t dq_bs, int64_t dq_ts,
                                   void* dk, int64_t dk_bs, int64_t dk_ts, void* dv, int64_t dv_bs, int64_t dv_ts, int dtype,
                                   int64_t batch, int heads, int64_t nq, int64_t nk, int head_dim, float scale, void* stream) {
    AttnArgs a = {};
    a.q = q; a.k = k; a.v = v;
    a.q_bs = q_bs; a.q_ts = q_ts; a.k_bs = k_bs; a.k_ts = k_ts; a.v_bs = v_bs; a.v_ts = v_ts;
    a.o = o; a.d_o = d_o; a.o_bs = o_bs; a.o_ts = o_ts;
    a.lse = (float*)lse; a.delta = delta;
    a.dq = dq; a.dk = dk; a.dv = dv;
    a.dq_bs = dq_bs; a.dq_ts = dq_ts; a.dk_bs = dk_bs; a.dk_ts = dk_ts; a.dv_bs = dv_bs; a.dv_ts = dv_ts;
    a.batch = batch; a.heads = heads; a.nq = nq; a.nk = nk; a.head_dim = head_dim; a.scale = scale;
    int rc = check_common(a, dtype);
    if (rc != VITED_OK) return rc;
    if (!d_o || !delta || !dq || !dk || !dv) return VITED_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VITED_BF16 && attention_mfma_supported(a, true)) {
        g_last_attn_path = 2;
        return attention_bwd_mfma(a, s);
    }
    g_last_attn_path = 1;
    return attention_bwd_portable(a, dtype, s);
}
