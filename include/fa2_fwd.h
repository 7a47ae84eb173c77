/*
 * fa2_fwd.h -- C ABI of the MI355X-native Flash-Attention-2 forward (libfa2_hip.so).
 *
 * This is the drop-in boundary for ONE path of 17ex/flash_attention_dlrs: the Triton launch
 *
 *     fwd_kernel[grid](Q, K, V, O, L,
 *                      QB,QH,QN,Qd, KB,KH,KN,Kd, VB,VH,VN,Vd, OB,OH,ON,Od, LB,LH,
 *                      B, H, N, d, dtype)
 *
 * made at  src/flash_attention_torch.py:61-74, :201-214  and  src/flash_attention_wrappers.py:48-61
 * of the kernel defined at  src/flash_attention_kernels.py:17-109.  fa2_fwd() takes the same
 * argument list (pointers instead of torch tensors, strides in ELEMENTS exactly as the reference
 * passes them) plus the three reference-preserving extensions BASELINE.json asks for (causal,
 * scale, stream).  No torch types, no C++ types, no exceptions cross this boundary.
 *
 * Ownership: the caller owns every buffer.  O and L are allocated by the caller BEFORE the call
 * (flash_attention_torch.py:50-51); the library only writes into them, allocates nothing, and keeps
 * no pointer after returning.  The launch is asynchronous on `hip_stream`.
 * Threading: re-entrant; the only mutable state is the thread-local last-error string.
 */
#ifndef FA2_FWD_H
#define FA2_FWD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dtype_enum.  Replaces convert_triton_dtype (src/flash_attention_torch.py:7-18), which maps
 * float64 / float32 / float16 / float8_e5m2.  bf16 and OCP e4m3fn are extensions. */
#define FA2_DTYPE_F32 0
#define FA2_DTYPE_F16 1
#define FA2_DTYPE_BF16 2
#define FA2_DTYPE_F8E5M2 3
#define FA2_DTYPE_F8E4M3 4
#define FA2_DTYPE_F64 5

/* Return codes (the Python glue re-raises them as the exception classes the reference uses,
 * flash_attention_torch.py:24-32, :18). */
#define FA2_OK 0
#define FA2_ERR_BAD_ARG (-1)     /* null pointer, non-positive size, misaligned/negative stride  */
#define FA2_ERR_UNSUPPORTED (-2) /* dtype enum unknown, d outside [1, 512], or a variant that cannot run the problem */
#define FA2_ERR_BAD_N (-3)       /* N < 1                                                        */
#define FA2_ERR_LAUNCH (-4)      /* HIP reported an error at launch                              */

/* Kernel variants (fa2_fwd_variant / fa2_query_tile).  AUTO = the static gfx950 tile table that
 * replaces the reference's run-time autotuner (src/autotune_configs.py:24-201, kernels.py:11-15). */
#define FA2_VARIANT_AUTO 0
#define FA2_VARIANT_GENERIC 1 /* any dtype, any strides, any d in [1,512], any N; FMA on VALU     */
#define FA2_VARIANT_MFMA16 2  /* f16/bf16, unit d-stride; 4 waves x 32 rows.  d = 64 / 128, and every other multiple of 8  */
                              /* up to 128 with the missing columns zero-filled on load (no host padding: the        */
                              /* reference pads Q, K, V to a power of two, torch.py:38-47).  Also the fallback when  */
                              /* N * row stride does not fit 32-bit buffer offsets                                    */
#define FA2_VARIANT_MFMA16_W8 3 /* same, 8 waves x 32 rows (256-row Q tile)                        */
#define FA2_VARIANT_MFMA32 4  /* f32 via v_mfma_f32_32x32x2_f32; d = 64 / 128, other multiples of 4 up to 128 zero-filled on load */
#define FA2_VARIANT_MFMA16D 8 /* f16/bf16 software-pipelined (32-key blocks), LDS-DMA staging (buffer_load ... lds), 8 waves x 32 rows */
#define FA2_VARIANT_MFMA16D_W4 9 /* same, 4 waves x 32 rows                                          */
#define FA2_VARIANT_MFMA16H 14 /* MFMA16D with a persistent grid, next-job prefetch and a hand-ordered steady loop; 8 waves */
#define FA2_VARIANT_MFMA16H_W4 15 /* same, 4 waves x 32 rows                                          */
#define FA2_VARIANT_MFMA8X 16 /* fp8 on the double-rate v_mfma_f32_32x32x64_f8f6f4: 64-key units, 8 waves x 32 rows       */
#define FA2_VARIANT_MFMA8X_W4 17 /* same, 4 waves x 32 rows                                          */
#define FA2_VARIANT_MFMA16K 19 /* f16/bf16 small grids: 8 waves on a 128-row tile, waves w and w+4 split the KEYS and merge through LDS */
#define FA2_VARIANT_MFMA16K_R2K2 20 /* same with a 64-row tile: 2 row blocks x 2 key groups, four waves                    */
#define FA2_VARIANT_MFMA16K_R2K4 23 /* 64-row tile, 2 row blocks x 4 key groups, eight waves (d = 64)                       */
#define FA2_VARIANT_A64 24 /* f16/bf16, d = 128, N >= 256 (any): generated gfx950 assembly, 4 waves x 64 rows, one wave per SIMD */
                           /* with the whole register file (O, Q, V^T in AGPRs), persistent grid, continuous tile stream.     */
                           /* Non-finite inputs: as the reference, except that a NaN in query row q also makes row q ^ 16 of   */
                           /* the same 32-row block NaN (the row sums run on the matrix pipe, where q ^ 16's P meets a zero    */
                           /* weight); finite inputs are unaffected.                                                          */
#define FA2_VARIANT_A16 25 /* the A64 structure on the other matrix shape, v_mfma_f32_16x16x32 (the chip holds a higher clock on it):  */
                           /* same shapes, same job stream; a 64-key step is 136 MFMAs of 16 cycles instead of 72 of 32.          */
#define FA2_VARIANT_A8 26  /* OCP fp8 (e4m3fn, e5m2), d = 128, N >= 256 (causal or not): the A64 structure on the double-rate            */
                           /* v_mfma_f32_32x32x64_f8f6f4 (generated assembly, asm/fa2_a8_gen.py), P.V on its block-scaled form: the  */
                           /* running maximum is an integer and rides in P's scale operand, O is never rescaled; BASELINE configs[4]  */
#define FA2_VARIANT_A64D 27 /* f16/bf16, HEAD SIZE 64, N >= 256: the A64 kernel at d = 64 (generated assembly,                         */
                            /* asm/fa2_a64d_gen.py); causal and not                                                                  */
/* (ids 5-7, 10-13, 18 and the ablation ids belong to experimental kernels that are not part of this library:
 *  flash_attention_dlrs_amd/csrc/fa2_experiments.h, `make -C flash_attention_dlrs_amd/csrc experiments`) */

/*
 * O = softmax(scale * Q K^T [+ causal mask]) V   and   L = log2-domain log-sum-exp of the scores,
 * L = m + log2(l)  (src/flash_attention_kernels.py:105-108).  scale = 1, causal = 0 is the reference.
 *
 *   Q, K, V : device pointers, logical shape (B, H, N, d), element strides q/k/v_strides[4]
 *   O       : device pointer, (B, H, N, d), strides o_strides[4], written in dtype_enum
 *   L       : device pointer, (B, H, N, 1) in dtype_enum; l_strides = {LB, LH}; unit stride over N
 *             (kernels.py:59-65)
 *   hip_stream : hipStream_t (may be NULL = default stream)
 */
int fa2_fwd(const void *Q, const void *K, const void *V, void *O, void *L,
            const int64_t q_strides[4], const int64_t k_strides[4], const int64_t v_strides[4],
            const int64_t o_strides[4], const int64_t l_strides[2], int32_t B, int32_t H, int32_t N,
            int32_t d, int32_t dtype_enum, int32_t causal, float scale, void *hip_stream);

/* Same, forcing one kernel variant (tests and bench A/B).  FA2_ERR_UNSUPPORTED if the variant
 * cannot run the given problem. */
int fa2_fwd_variant(const void *Q, const void *K, const void *V, void *O, void *L,
                    const int64_t q_strides[4], const int64_t k_strides[4],
                    const int64_t v_strides[4], const int64_t o_strides[4],
                    const int64_t l_strides[2], int32_t B, int32_t H, int32_t N, int32_t d,
                    int32_t dtype_enum, int32_t causal, float scale, void *hip_stream,
                    int32_t variant);

/* Which tile the static table picks for a contiguous problem: out4 = {variant, B_r, B_c, waves}.
 * Counterpart of fwd_conf_prune + the autotuner's choice (src/autotune_configs.py:176-194). */
int fa2_query_tile(int32_t N, int32_t d, int32_t dtype_enum, int32_t causal, int32_t out4[4]);

/* The table's choice depends on the grid size (B * H tiles must fill 256 CUs): fa2_query_tile answers for a large grid
 * (B = 64, H = 8), fa2_query_tile_ex for the given B and H -- the variant fa2_fwd() runs for that contiguous problem at
 * scale = 1 (the reference's). */
int fa2_query_tile_ex(int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum, int32_t causal,
                      int32_t out4[4]);

/* ... and for a given softmax scale: f16 rescales its accumulators every few key tiles at the reference's scale of 1 (P must stay
 * below 65 504) and hardly ever at the usual 1 / sqrt(d), which moves two of the table's thresholds (fa2_api.hip). */
int fa2_query_tile_scaled(int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum, int32_t causal, float scale,
                          int32_t out4[4]);

/* "fa2-hip <semver> gfx950". */
const char *fa2_version(void);

/* Message of the last non-zero return on the calling thread ("" if none). */
const char *fa2_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* FA2_FWD_H */
