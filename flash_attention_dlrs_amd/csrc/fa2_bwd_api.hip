// fa2_bwd_api.hip -- extern "C" entry points of the backward (declared in include/fa2_bwd.h).
#include <string.h>

#include "fa2_bwd_common.h"

namespace {

bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

int validate(const Fa2BwdProblem &p) {
    if (!p.Q || !p.K || !p.V || !p.O || !p.dO || !p.L || !p.dQ || !p.dK || !p.dV || !p.D) {
        fa2_set_error("null tensor pointer");
        return FA2_ERR_BAD_ARG;
    }
    if (p.B <= 0 || p.H <= 0 || p.d <= 0) {
        fa2_set_error("B, H, d must be positive (got B=%d H=%d d=%d)", p.B, p.H, p.d);
        return FA2_ERR_BAD_ARG;
    }
    if (p.N < 1) {
        fa2_set_error("N must be >= 1 (got %d)", p.N);
        return FA2_ERR_BAD_N;
    }
    if (p.dtype != FA2_DTYPE_F32 && p.dtype != FA2_DTYPE_F16 && p.dtype != FA2_DTYPE_BF16 && p.dtype != FA2_DTYPE_F64) {
        fa2_set_error("backward: dtype enum %d is not supported (f64, f32, f16, bf16)", p.dtype);
        return FA2_ERR_UNSUPPORTED;
    }
    if (!is_pow2(p.d) || p.d < 16 || p.d > 512) {  // the glue pads d exactly as in the forward (torch.py:95-99)
        fa2_set_error("d=%d must be a power of two in [16, 512] (pad on the host as the reference does)", p.d);
        return FA2_ERR_UNSUPPORTED;
    }
    for (int k = 0; k < 4; ++k)
        if (p.qs[k] < 0 || p.ks[k] < 0 || p.vs[k] < 0 || p.os[k] < 0 || p.dos[k] < 0 || p.dqs[k] < 0 || p.dks[k] < 0 || p.dvs[k] < 0) {
            fa2_set_error("negative strides are not supported");
            return FA2_ERR_BAD_ARG;
        }
    if (p.dqs[3] == 0 || p.dks[3] == 0 || p.dvs[3] == 0 || (p.N > 1 && (p.dqs[2] == 0 || p.dks[2] == 0 || p.dvs[2] == 0))) {
        fa2_set_error("dQ, dK, dV must not alias themselves (zero stride)");
        return FA2_ERR_BAD_ARG;
    }
    if (!(p.scale == p.scale)) {
        fa2_set_error("scale is NaN");
        return FA2_ERR_BAD_ARG;
    }
    return FA2_OK;
}

int run(const Fa2BwdProblem &p, int variant) {
    const int rc = validate(p);
    if (rc != FA2_OK) return rc;
    if (variant == FA2_BWD_VARIANT_AUTO)
        variant = fa2_bwd_mfma16_supports(p)   ? FA2_BWD_VARIANT_MFMA16
                  : fa2_bwd_mfma32_supports(p) ? FA2_BWD_VARIANT_MFMA32
                                               : FA2_BWD_VARIANT_GENERIC;
    switch (variant) {
    case FA2_BWD_VARIANT_GENERIC: return fa2_bwd_launch_generic(p);
    case FA2_BWD_VARIANT_MFMA16: return fa2_bwd_launch_mfma16(p);
    case FA2_BWD_VARIANT_MFMA32: return fa2_bwd_launch_mfma32(p);
    default: fa2_set_error("unknown backward kernel variant %d", variant); return FA2_ERR_BAD_ARG;
    }
}

Fa2BwdProblem make(const void *Q, const void *K, const void *V, const void *O, const void *dO, const void *L, void *dQ,
                   void *dK, void *dV, void *D, const int64_t *qs, const int64_t *ks, const int64_t *vs,
                   const int64_t *os, const int64_t *dos, const int64_t *dqs, const int64_t *dks, const int64_t *dvs,
                   const int64_t *ls, int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype, int32_t causal,
                   float scale, void *stream) {
    Fa2BwdProblem p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.K = K; p.V = V; p.O = O; p.dO = dO; p.L = L;
    p.dQ = dQ; p.dK = dK; p.dV = dV; p.D = D;
    if (qs && ks && vs && os && dos && dqs && dks && dvs && ls) {
        for (int k = 0; k < 4; ++k) {
            p.qs[k] = qs[k]; p.ks[k] = ks[k]; p.vs[k] = vs[k]; p.os[k] = os[k]; p.dos[k] = dos[k];
            p.dqs[k] = dqs[k]; p.dks[k] = dks[k]; p.dvs[k] = dvs[k];
        }
        p.ls[0] = ls[0]; p.ls[1] = ls[1];
    } else {
        p.Q = nullptr;  // fails validate()
    }
    p.B = B; p.H = H; p.N = N; p.d = d; p.dtype = dtype; p.causal = causal ? 1 : 0; p.scale = scale;
    p.stream = (hipStream_t)stream;
    return p;
}

}  // namespace

extern "C" {

int fa2_bwd(const void *Q, const void *K, const void *V, const void *O, const void *dO, const void *L, void *dQ,
            void *dK, void *dV, void *D, const int64_t q_strides[4], const int64_t k_strides[4],
            const int64_t v_strides[4], const int64_t o_strides[4], const int64_t do_strides[4],
            const int64_t dq_strides[4], const int64_t dk_strides[4], const int64_t dv_strides[4],
            const int64_t l_strides[2], int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum, int32_t causal,
            float scale, void *hip_stream) {
    return run(make(Q, K, V, O, dO, L, dQ, dK, dV, D, q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides,
                    dk_strides, dv_strides, l_strides, B, H, N, d, dtype_enum, causal, scale, hip_stream),
               FA2_BWD_VARIANT_AUTO);
}

int fa2_bwd_variant(const void *Q, const void *K, const void *V, const void *O, const void *dO, const void *L,
                    void *dQ, void *dK, void *dV, void *D, const int64_t q_strides[4], const int64_t k_strides[4],
                    const int64_t v_strides[4], const int64_t o_strides[4], const int64_t do_strides[4],
                    const int64_t dq_strides[4], const int64_t dk_strides[4], const int64_t dv_strides[4],
                    const int64_t l_strides[2], int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum,
                    int32_t causal, float scale, void *hip_stream, int32_t variant) {
    return run(make(Q, K, V, O, dO, L, dQ, dK, dV, D, q_strides, k_strides, v_strides, o_strides, do_strides, dq_strides,
                    dk_strides, dv_strides, l_strides, B, H, N, d, dtype_enum, causal, scale, hip_stream),
               variant);
}

}  // extern "C"
