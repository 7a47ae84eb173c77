/*
 * fa2_bwd.h -- C ABI of the MI355X-native Flash-Attention-2 backward (libfa2_hip.so), SURVEY.md section 8 row f1.
 *
 * Drop-in boundary for the launch PAIR the reference's host glue makes in FlashAttention.backward
 * (src/flash_attention_torch.py:124-155; the deterministic class makes the same pair at :262-291):
 *
 *     bwd_D_kernel[grid](O, dO, D, <O strides>, <dO strides>, DB, DH, B, H, N, d, dtype)          kernels.py:115-166
 *     bwd_kernel[grid](Q, K, V, dQ, dK, dV, dO, L, D, lock_dQ, written_dQ,                        kernels.py:174-334
 *                      <Q, K, V, dQ, dK, dV, dO strides>, LB, LH, DB, DH, <lock strides>, B, H, N, d, dtype)
 *
 * fa2_bwd() takes the same tensors as plain pointers with element strides, plus the reference-preserving
 * extensions of the forward (causal, scale, stream).  What is NOT carried over: lock_dQ / written_dQ.  The
 * reference sums dQ across key-block programs through a spin lock (bwd_kernel, documented by its author as wrong on
 * first use) or an ordered hand-off that cannot complete with more than one key block (bwd_deterministic_kernel,
 * see tests/golden/gen_golden_bwd.py).  Here dK/dV and dQ come from two kernels that each own their outputs
 * (a key block per workgroup, then a query block per workgroup), so there is no cross-workgroup sum at all: the
 * result is deterministic and both reference classes (FlashAttention, FlashAttentionDeterministic) map to it.
 *
 * Ownership: the caller owns every buffer, D included: the reference's glue allocates D = empty_like(L)
 * (torch.py:105) before the launch; here D is a float32 (float64 for FA2_DTYPE_F64) scratch of 2*B*H*N elements:
 * rowsum(dO * O) kept unrounded (the reference rounds it to the I/O dtype, kernels.py:165), followed by the fp32 row
 * statistic L + log2(rowsum P) that the dQ launch measures and the dK/dV launch consumes -- the forward stores L in
 * the I/O dtype (kernels.py:108) and that rounding alone would scale a row of P by up to 2^0.125 in bf16.  Nothing is allocated, freed or
 * retained by the library; the launches are asynchronous on `hip_stream`.
 */
#ifndef FA2_BWD_H
#define FA2_BWD_H

#include <stdint.h>

#include "fa2_fwd.h" /* FA2_DTYPE_*, FA2_OK / FA2_ERR_*, fa2_last_error() */

#ifdef __cplusplus
extern "C" {
#endif

/* Backward kernel variants (fa2_bwd_variant). */
#define FA2_BWD_VARIANT_AUTO 0
#define FA2_BWD_VARIANT_GENERIC 1 /* any dtype but fp8, any strides, d = 2^k in [16,512], any N; VALU            */
#define FA2_BWD_VARIANT_MFMA16 2  /* f16 / bf16, d in {64,128}, unit d-stride, 16-byte aligned rows; MFMA        */
#define FA2_BWD_VARIANT_MFMA32 3  /* f32 via v_mfma_f32_32x32x2_f32 (exact fp32), d in {64,128}, same layout rules */

/*
 * dQ, dK, dV = gradients of  O = softmax(scale * Q K^T [+ causal mask]) V  given dO, with P recomputed from the
 * forward's log2-domain log-sum-exp:  P = exp2(scale * log2(e) * Q K^T - L)   (kernels.py:283-285).
 *
 *   Q, K, V, O, dO : device pointers, logical shape (B, H, N, d), element strides *_strides[4]
 *   L              : (B, H, N, 1) in dtype_enum as written by fa2_fwd; l_strides = {LB, LH}, unit stride over N
 *   dQ, dK, dV     : outputs, (B, H, N, d) in dtype_enum, element strides d*_strides[4]
 *   D              : scratch, 2*B*H*N contiguous float32 (float64 if dtype_enum == FA2_DTYPE_F64), see above
 *   fp8 dtypes are rejected with FA2_ERR_UNSUPPORTED (the reference maps float8_e5m2 but its backward cannot
 *   represent the gradients in it either).
 */
int fa2_bwd(const void *Q, const void *K, const void *V, const void *O, const void *dO, const void *L,
            void *dQ, void *dK, void *dV, void *D,
            const int64_t q_strides[4], const int64_t k_strides[4], const int64_t v_strides[4],
            const int64_t o_strides[4], const int64_t do_strides[4], const int64_t dq_strides[4],
            const int64_t dk_strides[4], const int64_t dv_strides[4], const int64_t l_strides[2],
            int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum, int32_t causal, float scale,
            void *hip_stream);

/* Same, forcing one kernel variant (tests and bench A/B). */
int fa2_bwd_variant(const void *Q, const void *K, const void *V, const void *O, const void *dO, const void *L,
                    void *dQ, void *dK, void *dV, void *D,
                    const int64_t q_strides[4], const int64_t k_strides[4], const int64_t v_strides[4],
                    const int64_t o_strides[4], const int64_t do_strides[4], const int64_t dq_strides[4],
                    const int64_t dk_strides[4], const int64_t dv_strides[4], const int64_t l_strides[2],
                    int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum, int32_t causal, float scale,
                    void *hip_stream, int32_t variant);

#ifdef __cplusplus
}
#endif
#endif /* FA2_BWD_H */
