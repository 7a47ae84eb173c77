// Experimental kernel variants and timing-only ablations (internal).  They are NOT part of libfa2_hip.so or of include/fa2_fwd.h:
// `make experiments` builds libfa2_hip_exp.so (all shipped kernels plus these, tuning environment variables enabled), which
// benchmarks/ load through FA2_HIP_LIB for A/B runs.
#pragma once
#define FA2_VARIANT_MFMA16P 5 /* f16/bf16 software-pipelined, register staging, 4 waves x 32 rows (A/B baseline of MFMA16D)  */
#define FA2_VARIANT_MFMA16P_W8 6 /* same, 8 waves */
#define FA2_VARIANT_MFMA16X 7 /* first 4 waves x 64 rows, one wave per SIMD attempt (compiler-scheduled; superseded by A64) */
#define FA2_VARIANT_MFMA8 10  /* fp8 on v_mfma_f32_32x32x16_fp8_fp8 (the bf16 rate): A/B baseline of MFMA8X */
#define FA2_VARIANT_MFMA8_W4 11
#define FA2_VARIANT_MFMA16S 12 /* MFMA16D on v_mfma_f32_16x16x32: no lane exchange in the loop */
#define FA2_VARIANT_MFMA16S_W4 13
#define FA2_VARIANT_MFMA8U 18 /* MFMA8X_W4 unpipelined at <= 168 registers: three workgroups per CU */
