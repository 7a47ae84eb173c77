// fa2_api.hip -- extern "C" entry points of libfa2_hip.so (declared in include/fa2_fwd.h) and the
// static gfx950 tile table that stands in for the reference's run-time autotuner
// (src/autotune_configs.py:24-201, src/flash_attention_kernels.py:11-15: 114 Triton configs,
// pruned by an SRAM heuristic and benchmarked on first use of every (B, H, N, d)).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fa2_common.h"

namespace {
thread_local char g_err[512] = "";

bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }

int validate(const Fa2Problem &p) {
    if (!p.Q || !p.K || !p.V || !p.O || !p.L) {
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
    if (fa2_dtype_size(p.dtype) == 0) {
        fa2_set_error("unknown dtype enum %d", p.dtype);
        return FA2_ERR_UNSUPPORTED;
    }
    // The reference's host glue pads d to max(next_pow2(d), 16) before launching (src/flash_attention_torch.py:38); here
    // any head size goes to the kernels as it is: the MFMA kernels zero-fill the missing columns
    // on load, the generic kernel loops to d.
    if (p.d < 1 || p.d > 512) {
        fa2_set_error("d=%d must be in [1, 512]", p.d);
        return FA2_ERR_UNSUPPORTED;
    }
    for (int k = 0; k < 4; ++k)
        if (p.qs[k] < 0 || p.ks[k] < 0 || p.vs[k] < 0 || p.os[k] < 0) {
            fa2_set_error("negative strides are not supported");
            return FA2_ERR_BAD_ARG;
        }
    if (p.os[3] == 0 || (p.N > 1 && p.os[2] == 0)) {
        fa2_set_error("O must not alias itself (zero stride)");
        return FA2_ERR_BAD_ARG;
    }
    if (!(p.scale == p.scale)) {
        fa2_set_error("scale is NaN");
        return FA2_ERR_BAD_ARG;
    }
    return FA2_OK;
}

// The static tile table, keyed like the reference's autotuner on (B, H, N, d) (plus dtype, causal, strides): the grid
// size B * H * tiles decides between the key-split, 4-wave and one-workgroup-per-CU kernels.
int pick_variant(const Fa2Problem &p) {
    // The thresholds below were measured on the full chip (256 CUs) and are written in tiles / jobs per 256 CUs: T(n) scales
    // them to the CU count of the current device (a partitioned or smaller part fills at proportionally smaller grids).
    const int cus = fa2_device_cus();
    auto T = [cus](long long n) { return (n * cus + 128) / 256; };
    if (fa2_mfma16_supports(p)) {
        // Software-pipelined kernel with LDS-DMA staging.  8 waves x 32 rows halves the K/V traffic per query
        // row; it needs enough 256-row tiles to fill 256 CUs, otherwise the 128-row tile spreads the work wider.
        // (N * row stride >= 2 GiB does not fit the 32-bit buffer offsets: fall back to the first MFMA kernel.)
        const bool fits32 = (int64_t)(p.N + 512) * p.ks[2] * 2 < (1LL << 31) && (int64_t)(p.N + 512) * p.vs[2] * 2 < (1LL << 31) &&
                            (int64_t)(p.N + 512) * p.os[2] * 2 < (1LL << 31);
        const long long wg256 = (long long)((p.N + 255) / 256) * p.B * p.H;
        if (!fits32) return wg256 >= T(512) ? FA2_VARIANT_MFMA16_W8 : FA2_VARIANT_MFMA16;
        // d = 128, N a multiple of 256: the generated assembly kernel A64 (4 waves x 64 rows, one wave per SIMD, O / Q / V^T in
        // the accumulator file) wins from 64 jobs of 256 rows on -- a quarter of the CUs busy -- non-causal, and causal from
        // 192 jobs (or 96 when a job has at least eight key tiles); below that the key-split kernels keep more CUs busy.
        // benchmarks/mid_grid.py, 48 shapes, bf16, d = 128, profiles/r02/mid_grid_a64.jsonl: against the best of
        // MFMA16D_W4 / MFMA16H / MFMA16K it is 4-33 % faster on every shape from those sizes up (the persistent grid also
        // takes job counts that are not a multiple of the CU count better: 1.5 jobs per CU 180 vs 213 us); on the same
        // MI355X against MFMA16H: c3 causal +15-19 %, c3 shape non-causal +15-17 %.
        // head size 64: the generated kernel at d = 64 (A64D, asm/fa2_a64d_gen.py; N a multiple of 256 or its ragged form) under a64's grid rule.
        // Same-device A/B against the rest of this table (profiles/r03/a64d_vs_table.jsonl), bf16 / f16: B8 H16 N4096 1 013 vs 840
        // TFLOP/s (+21 %), causal 1 014 vs 827 (+23 %); N = 8192 1 092 vs 929, causal 1 060 vs 887; B16 H32 N2048 984 vs 821,
        // causal 860 vs 677 (+27 %)
        // Grid rule from the d = 64 mid-grid sweep (benchmarks/mid_grid.py, profiles/r03/mid_grid_d64.jsonl, 48 shapes): from 192 jobs
        // of 256 rows on A64D is the best of the six kernels or within 4 % of it; below, the key-split and 128-row kernels keep more
        // CUs busy (64 jobs: 13.1 us for MFMA16K_R2K4 against 17.0)
        // (f16 at the reference's scale of 1 -- rescales every few tiles, see A64 below -- causal: only from 384 jobs = 192 units on;
        // at 128 units the 8-wave kernels lead by 3 .. 13 %, profiles/r03/mid_grid_f16_d64.jsonl)
        const bool f16_hot64 = p.dtype == FA2_DTYPE_F16 && p.scale > 0.5f;
        if (fa2_a64d_supports(p) && wg256 >= T(p.causal ? (f16_hot64 ? 384 : 192) : 160)) return FA2_VARIANT_A64D;
        // f16 at the reference's scale of 1 rescales every few tiles (P must stay below 65 504); with one wave per SIMD a rescale is
        // ~2 000 cycles with three waves waiting, which the 8-wave kernels hide.  On a full chip A64 still leads; on half-filled
        // grids it loses 12 .. 30 % to them (profiles/r03/mid_grid_f16.jsonl: 64 jobs 30.9 vs 21.5 us, 128 jobs 32.0 vs 27.0, causal
        // N = 2048 128 jobs 60.1 vs 45.2), where bf16 -- which never rescales -- wins.  At softmax scales <= 0.5 f16 behaves like bf16.
        const bool f16_hot = p.dtype == FA2_DTYPE_F16 && p.scale > 0.5f;
        const bool a64_grid = f16_hot ? wg256 >= T(256)
                                      : (p.causal ? (wg256 >= T(192) || (wg256 >= T(96) && p.N >= 2048)) : wg256 >= T(64));
        if (fa2_a64_supports(p) && a64_grid) {
            // The same kernel on the other matrix shape (A16: v_mfma_f32_16x16x32, asm/fa2_a16_gen.py): 15 % more cycles per key
            // step, but the chip holds a 10-17 % higher clock under it.  Same-device A/B against A64 (benchmarks/variants.py,
            // profiles/r03/a16_vs_a64.jsonl), bf16: non-causal N = 4096 +2.4 .. +4.4 %, N = 8192 (BASELINE configs[3]'s shard)
            // +4.4 %, N = 2048 +1.3 %; causal N = 8192 +2.0 %, N = 4096 -1.0 %, N = 2048 -3.5 % (short jobs: the seam and the
            // epilogue grow with the cycles, the clock gain is smaller there); f16 (rescales every few tiles) -2.5 %.
            // f16 at the reference's scale of 1 rescales every few tiles (P must stay below 65 504), which the 16x16 form pays for
            // twice (-3 %); at the usual softmax scales the maximum rarely moves: scale 0.5 / 0.25 / 1 / sqrt(128) on the reference
            // bench's shape: +3.0 / +5.4 / +3.6 % (scripts/gpu_f16_scale.sh, profiles/r03/f16_scale_a16_vs_a64.jsonl)
            const bool a16_dtype = p.dtype == FA2_DTYPE_BF16 || (p.dtype == FA2_DTYPE_F16 && p.scale <= 0.5f);
            // ... and only where the chip is full enough to sit on its power limit: on half-filled grids (128 jobs, or 128 causal
            // units: 2.4 GHz whatever the shape) the extra cycles cost 15 % (profiles/r03/mid_grid_final.jsonl: N = 4096 B*H = 8 78.9 vs
            // 68.3 us, causal N = 8192 B*H = 8 158.2 vs 137.8); from 192 jobs on it leads
            const long long units16 = p.causal ? (wg256 + 1) / 2 : wg256;
            if (a16_dtype && (p.causal ? p.N >= 8192 : p.N >= 4096) && units16 >= T(192)) return FA2_VARIANT_A16;
            return FA2_VARIANT_A64;
        }
        // Small grids: at most 128 work units (128-row tiles, tile PAIRS when causal) leave half of the 256 CUs idle and
        // every workgroup walks its key tiles in sequence -- latency-bound.  MFMA16K splits the keys of a tile among
        // wave groups of the same workgroup (no workspace): benchmarks/tiny_grid.py, HIP-graph replay, fp16:
        // B2 H8 N1024 d64 (BASELINE.json configs[1]) 16.8 -> 11.2 us; d = 128 25.3 -> 21.2; causal d = 128 38.6 -> 21.2;
        // B1 H8 N4096 d128 causal 112.8 -> 79.3.  At 192 units and above the plain kernels are as fast or faster.
        const long long wg128 = (long long)((p.N + 127) / 128) * p.B * p.H;
        if ((p.causal ? wg128 / 2 : wg128) <= T(128)) {
            // causal, 129..256 tiles: 128-row tiles, two key groups -- except d = 64 at two 64-row workgroups per CU and long rows,
            // where the (2, 4) shape leads (mid_grid_final.jsonl: N = 4096 B*H = 8 39.6 vs 51.2 us, N = 2048 B*H = 16 25.0 vs 29.1; at
            // 1.5 per CU or N = 1024 it loses)
            if (wg128 > T(128)) return p.d == 64 && p.N >= 2048 && wg128 > T(224) ? FA2_VARIANT_MFMA16K_R2K4 : FA2_VARIANT_MFMA16K;
            if (p.d == 64 && p.N >= 512) return FA2_VARIANT_MFMA16K_R2K4;
            return FA2_VARIANT_MFMA16K_R2K2;
        }
        // Mid-size grids (benchmarks/mid_grid.py, profiles/r01/mid_grid_bf16.jsonl: 96 shapes, bf16, 64..1024 256-row
        // tiles).  MFMA16H is a persistent grid of one 8-wave workgroup per CU: its time goes in steps of whole jobs per
        // CU (256-row tiles, tile PAIRS when causal), so it wants the job count near a multiple of 256; the 4-wave
        // MFMA16D_W4 has twice the jobs at half the size.  Readings behind the rules: d = 128 non-causal, 192 / 256 tiles:
        // MFMA16H -12..16 % / -5..10 %, 384 tiles (1.5 jobs per CU): MFMA16D_W4 -5 %; d = 128 causal, 384 tiles: MFMA16H
        // -16..19 %; d = 64 causal, 768 tiles (1.5 pairs per CU): MFMA16D_W4
        // -20 %, 256 tiles: MFMA16D_W4 -25 %.
        const int nq256 = (p.N + 255) / 256;
        const long long jobs = (long long)(p.causal ? (nq256 + 1) / 2 : nq256) * p.B * p.H;  // of MFMA16H
        const double x = (double)jobs / (double)cus;
        const bool even = x <= 1.0 ? x >= 0.6 : (double)((jobs + cus - 1) / cus) / x <= 1.2;  // <= 20 % lost to whole jobs
        // (two full rounds of the 8-wave key-split workgroups: B1 H16 N4096 98 vs 105 us, B1 H8 N8192 173 vs 191; at
        // 1.5 rounds -- 384 tiles -- it loses 50 %)
        if (p.causal && p.d == 128 && wg128 > T(448) && wg128 <= T(512) && p.N >= 4096) return FA2_VARIANT_MFMA16K;
        if (p.d == 128) {
            if (p.causal ? wg256 < T(320) : (wg256 < T(160) || !even)) return FA2_VARIANT_MFMA16D_W4;
        } else {
            if (wg256 < T(160) || !even) return FA2_VARIANT_MFMA16D_W4;
        }
        // 8-wave tiles: MFMA16H (persistent grid, next-job prefetch, hand-ordered steady loop).  Against MFMA16D on
        // MI355X (benchmarks/lottery.py, alternating order): non-causal +4.4 % (d = 128, N = 4096), +3.4 % (N = 8192),
        // +9 % (d = 64); causal, once its causal kernels got a translation unit of their own: +4.1 % at the north-star
        // shape (N = 4096), +4 % (N = 2048), +2.6 % (N = 8192), +2.5 % (N = 16384), +2.4 % (d = 64).
        return FA2_VARIANT_MFMA16H;
    }
    // (The generated fp8 kernel A8 -- asm/fa2_a8_gen.py: the A64 structure on v_mfma_f32_32x32x64_f8f6f4 -- in its first form rescaled
    // O whenever the running maximum moved, like MFMA8X: with one wave per SIMD that cost ~900 cycles per rescale with three waves
    // waiting, and on N(0, 1) inputs (scores of sigma 16 log2 units against fp8's few units of deferral) it lost to the 8-wave kernel,
    // 2 080 vs 2 216 TFLOP/s on BASELINE configs[4]'s shard.  Its P.V now runs on the BLOCK-SCALED MFMA: the running maximum is an
    // integer and the power of two rides in P's E8M0 scale operand, O is never touched: 2 457 vs 2 260 on N(0, 1), 2 418 vs 2 231 on
    // N(0, 1/4) (profiles/r03/a8_scaled_vs_mfma8x.jsonl); grids: benchmarks/mid_grid_fp8.py, profiles/r03/mid_grid_fp8_a8.jsonl)
    if (fa2_mfma8x_supports(p)) {
        // fp8: the double-rate k = 64 MFMA (64-key units).  Against MFMA8 (32x32x16 fp8, the bf16 rate) on MI355X:
        // +26 % at the c5 per-GPU shape (N = 16384 non-causal: 1 880 vs 1 492 TFLOP/s), +19 % at c3 causal.
        // 8 waves (one workgroup per CU) when the job count -- 256-row tiles, tile pairs when causal -- fills the 256 CUs
        // evenly, else 4 waves: the rule of the bf16 table above, checked on 52 fp8 shapes (benchmarks/mid_grid_fp8.py,
        // profiles/r01/mid_grid_fp8.jsonl; the old single threshold lost up to 20 % at 1.5 jobs per CU).
        const int nq256 = (p.N + 255) / 256;
        const long long wg256 = (long long)nq256 * p.B * p.H;
        const long long jobs = (long long)(p.causal ? (nq256 + 1) / 2 : nq256) * p.B * p.H;
        const double x = (double)jobs / (double)cus;
        const bool even = x <= 1.0 ? x >= 0.6 : (double)((jobs + cus - 1) / cus) / x <= 1.2;
        // the generated kernel (N a multiple of 256): best or equal on every even grid from 192 jobs on, +5 .. +9 % from 512 jobs on
        // (causal: +8 .. +19 %, profiles/r03/mid_grid_fp8_a8_causal.jsonl, a8_causal_vs_mfma8x.jsonl); at 1.5 jobs per CU (384) its
        // whole-job steps lose 10 % to the 4-wave kernel like the 8-wave one's
        if (fa2_a8_supports(p) && wg256 >= T(192) && even) return FA2_VARIANT_A8;
        return wg256 >= T(160) && even ? FA2_VARIANT_MFMA8X : FA2_VARIANT_MFMA8X_W4;
    }
    if (fa2_mfma32_supports(p)) return FA2_VARIANT_MFMA32;
    // 16-bit head sizes other than 64 / 128 (multiples of 8): the first MFMA kernel with the missing columns zero-filled
    // on load; 8 waves when there are enough 256-row tiles for the 256 CUs
    if (fa2_mfma16_supports_dp(p))
        return (long long)((p.N + 255) / 256) * p.B * p.H >= T(512) ? FA2_VARIANT_MFMA16_W8 : FA2_VARIANT_MFMA16;
    return FA2_VARIANT_GENERIC;
}

int run(const Fa2Problem &p, int variant) {
    int rc = validate(p);
    if (rc != FA2_OK) return rc;
    if (variant == FA2_VARIANT_AUTO) variant = pick_variant(p);
    switch (variant) {
    case FA2_VARIANT_GENERIC: return fa2_launch_generic(p);
    case FA2_VARIANT_MFMA16: return fa2_launch_mfma16(p, 4);
    case FA2_VARIANT_MFMA16_W8: return fa2_launch_mfma16(p, 8);
    case FA2_VARIANT_MFMA32: return fa2_launch_mfma32(p);
#ifdef FA2_EXPERIMENTS
    case FA2_VARIANT_MFMA16P: return fa2_launch_mfma16p(p, 4, 0);
    case FA2_VARIANT_MFMA16P_W8: return fa2_launch_mfma16p(p, 8, 64);
    case FA2_VARIANT_MFMA16X: return fa2_launch_mfma16x(p, 0);
    case FA2_VARIANT_MFMA8: return fa2_launch_mfma8(p, 8);
    case FA2_VARIANT_MFMA8_W4: return fa2_launch_mfma8(p, 4);
#endif
    case FA2_VARIANT_MFMA8X: return fa2_launch_mfma8x(p, 8);
    case FA2_VARIANT_MFMA8X_W4: return fa2_launch_mfma8x(p, 4);
#ifdef FA2_EXPERIMENTS
    case FA2_VARIANT_MFMA8U: return fa2_launch_mfma8x(p, 12);
#endif
    case FA2_VARIANT_MFMA16K: return fa2_launch_mfma16k(p, 42);
    case FA2_VARIANT_MFMA16K_R2K2: return fa2_launch_mfma16k(p, 22);
    case FA2_VARIANT_MFMA16K_R2K4: return fa2_launch_mfma16k(p, 24);
    case FA2_VARIANT_MFMA16D: return fa2_launch_mfma16d(p, 8);
    case FA2_VARIANT_MFMA16D_W4: return fa2_launch_mfma16d(p, 4);
    case FA2_VARIANT_MFMA16H: return fa2_launch_mfma16h(p, 8);
    case FA2_VARIANT_MFMA16H_W4: return fa2_launch_mfma16h(p, 4);
    case FA2_VARIANT_A64: return fa2_launch_a64(p);
    case FA2_VARIANT_A16: return fa2_launch_a16(p);
    case FA2_VARIANT_A8: return fa2_launch_a8(p);
    case FA2_VARIANT_A64D: return fa2_launch_a64d(p);
#ifdef FA2_EXPERIMENTS
    case FA2_VARIANT_MFMA16S: return fa2_launch_mfma16s(p, 8);
    case FA2_VARIANT_MFMA16S_W4: return fa2_launch_mfma16s(p, 4);
    case FA2_VARIANT_MFMA16X + 2048 * 1: return fa2_launch_mfma16x(p, 1);   // ablations (FA2_ABLATIONS builds)
    case FA2_VARIANT_MFMA16X + 2048 * 3: return fa2_launch_mfma16x(p, 3);
    case FA2_VARIANT_MFMA16X + 2048 * 4: return fa2_launch_mfma16x(p, 4);
    case FA2_VARIANT_MFMA16X + 2048 * 7: return fa2_launch_mfma16x(p, 7);
    case FA2_VARIANT_MFMA16X + 2048 * 8: return fa2_launch_mfma16x(p, 8);
    case FA2_VARIANT_MFMA16X + 2048 * 15: return fa2_launch_mfma16x(p, 15);
    case FA2_VARIANT_MFMA16X + 2048 * 16: return fa2_launch_mfma16x(p, 16);
    case FA2_VARIANT_MFMA16X + 2048 * 32: return fa2_launch_mfma16x(p, 32);
    case FA2_VARIANT_MFMA16X + 2048 * 48: return fa2_launch_mfma16x(p, 48);
    case FA2_VARIANT_MFMA16P + 16: return fa2_launch_mfma16p(p, 4, 1);     // experimental schedules (A/B only)
    case FA2_VARIANT_MFMA16P_W8 + 16: return fa2_launch_mfma16p(p, 8, 1);
    case FA2_VARIANT_MFMA16P_W8 + 32: return fa2_launch_mfma16p(p, 8, 2);   // ablations: only in -DFA2_ABLATIONS builds
    case FA2_VARIANT_MFMA16P_W8 + 64: return fa2_launch_mfma16p(p, 8, 4);
    case FA2_VARIANT_MFMA16P_W8 + 128: return fa2_launch_mfma16p(p, 8, 8);
    case FA2_VARIANT_MFMA16P_W8 + 224: return fa2_launch_mfma16p(p, 8, 14);
    case FA2_VARIANT_MFMA16P_W8 + 256: return fa2_launch_mfma16p(p, 8, 16);
    case FA2_VARIANT_MFMA16P_W8 + 512: return fa2_launch_mfma16p(p, 8, 32);
    case FA2_VARIANT_MFMA16P_W8 + 736: return fa2_launch_mfma16p(p, 8, 46);
    case FA2_VARIANT_MFMA16P_W8 + 192 * 16: return fa2_launch_mfma16p(p, 8, 192);
    case FA2_VARIANT_MFMA16P_W8 + 1024: return fa2_launch_mfma16p(p, 8, 0);
    case FA2_VARIANT_MFMA16P + 1024: return fa2_launch_mfma16p(p, 4, 64);
#endif
    default: fa2_set_error("unknown kernel variant %d", variant); return FA2_ERR_BAD_ARG;
    }
}

Fa2Problem make_problem(const void *Q, const void *K, const void *V, void *O, void *L, const int64_t *qs,
                        const int64_t *ks, const int64_t *vs, const int64_t *os, const int64_t *ls, int32_t B,
                        int32_t H, int32_t N, int32_t d, int32_t dtype, int32_t causal, float scale, void *stream) {
    Fa2Problem p;
    memset(&p, 0, sizeof(p));
    p.Q = Q; p.K = K; p.V = V; p.O = O; p.L = L;
    if (qs && ks && vs && os && ls) {
        for (int k = 0; k < 4; ++k) { p.qs[k] = qs[k]; p.ks[k] = ks[k]; p.vs[k] = vs[k]; p.os[k] = os[k]; }
        p.ls[0] = ls[0]; p.ls[1] = ls[1];
    } else {
        p.Q = nullptr;  // fails validate()
    }
    p.B = B; p.H = H; p.N = N; p.d = d; p.dtype = dtype; p.causal = causal ? 1 : 0; p.scale = scale;
    p.stream = (hipStream_t)stream;
    return p;
}
}  // namespace

int fa2_launch_mfma16h(const Fa2Problem &p, int waves) {
    return p.causal ? fa2_launch_mfma16h_causal(p, waves) : fa2_launch_mfma16h_noncausal(p, waves);
}

void fa2_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fa2_env_int(const char *name, int dflt) {
#ifdef FA2_TUNING_ENV
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

namespace {
int current_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    return dev;
}
}  // namespace

int fa2_device_cus() {
    static int cus[64] = {0};
    const int dev = current_device();
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

bool Fa2DeviceLatch::need() const { return !((__atomic_load_n(&done, __ATOMIC_RELAXED) >> current_device()) & 1ull); }
void Fa2DeviceLatch::mark() { __atomic_fetch_or(&done, 1ull << current_device(), __ATOMIC_RELAXED); }

extern "C" {

int fa2_fwd(const void *Q, const void *K, const void *V, void *O, void *L, const int64_t q_strides[4],
            const int64_t k_strides[4], const int64_t v_strides[4], const int64_t o_strides[4],
            const int64_t l_strides[2], int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum,
            int32_t causal, float scale, void *hip_stream) {
    const Fa2Problem p = make_problem(Q, K, V, O, L, q_strides, k_strides, v_strides, o_strides, l_strides, B, H, N,
                                      d, dtype_enum, causal, scale, hip_stream);
    return run(p, FA2_VARIANT_AUTO);
}

int fa2_fwd_variant(const void *Q, const void *K, const void *V, void *O, void *L, const int64_t q_strides[4],
                    const int64_t k_strides[4], const int64_t v_strides[4], const int64_t o_strides[4],
                    const int64_t l_strides[2], int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum,
                    int32_t causal, float scale, void *hip_stream, int32_t variant) {
    const Fa2Problem p = make_problem(Q, K, V, O, L, q_strides, k_strides, v_strides, o_strides, l_strides, B, H, N,
                                      d, dtype_enum, causal, scale, hip_stream);
    return run(p, variant);
}

int fa2_query_tile(int32_t N, int32_t d, int32_t dtype_enum, int32_t causal, int32_t out4[4]) {
    return fa2_query_tile_ex(64, 8, N, d, dtype_enum, causal, out4);
}

int fa2_query_tile_ex(int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum, int32_t causal, int32_t out4[4]) {
    return fa2_query_tile_scaled(B, H, N, d, dtype_enum, causal, 1.0f, out4);
}

int fa2_query_tile_scaled(int32_t B, int32_t H, int32_t N, int32_t d, int32_t dtype_enum, int32_t causal, float scale,
                          int32_t out4[4]) {
    if (!out4) {
        fa2_set_error("out4 is null");
        return FA2_ERR_BAD_ARG;
    }
    // A contiguous (B, H, N, d) problem with aligned dummy pointers: only the table is consulted.
    Fa2Problem p;
    memset(&p, 0, sizeof(p));
    p.Q = p.K = p.V = (const void *)0x1000;
    p.O = p.L = (void *)0x1000;
    p.B = B; p.H = H; p.N = N; p.d = d; p.dtype = dtype_enum; p.causal = causal ? 1 : 0; p.scale = scale;
    const int64_t s[4] = {(int64_t)H * N * d, (int64_t)N * d, d, 1};
    for (int k = 0; k < 4; ++k) p.qs[k] = p.ks[k] = p.vs[k] = p.os[k] = s[k];
    p.ls[0] = (int64_t)H * N; p.ls[1] = N;
    const int rc = validate(p);
    if (rc != FA2_OK) return rc;
    const int v = pick_variant(p);
    out4[0] = v;
    switch (v) {
    case FA2_VARIANT_MFMA16: out4[1] = 128; out4[2] = 64; out4[3] = 4; break;
    case FA2_VARIANT_MFMA16_W8: out4[1] = 256; out4[2] = 64; out4[3] = 8; break;
#ifdef FA2_EXPERIMENTS
    case FA2_VARIANT_MFMA16P: out4[1] = 128; out4[2] = 32; out4[3] = 4; break;
    case FA2_VARIANT_MFMA16P_W8: out4[1] = 256; out4[2] = 32; out4[3] = 8; break;
    case FA2_VARIANT_MFMA16X: out4[1] = 256; out4[2] = 32; out4[3] = 4; break;
#endif
    case FA2_VARIANT_A64: out4[1] = 256; out4[2] = 64; out4[3] = 4; break;
    case FA2_VARIANT_A16: out4[1] = 256; out4[2] = 64; out4[3] = 4; break;
    case FA2_VARIANT_A8: out4[1] = 256; out4[2] = 64; out4[3] = 4; break;
    case FA2_VARIANT_A64D: out4[1] = 256; out4[2] = 64; out4[3] = 4; break;
#ifdef FA2_EXPERIMENTS
    case FA2_VARIANT_MFMA8: out4[1] = 256; out4[2] = 32; out4[3] = 8; break;
    case FA2_VARIANT_MFMA8_W4: out4[1] = 128; out4[2] = 32; out4[3] = 4; break;
#endif
    case FA2_VARIANT_MFMA8X: out4[1] = 256; out4[2] = 64; out4[3] = 8; break;
    case FA2_VARIANT_MFMA8X_W4: out4[1] = 128; out4[2] = 64; out4[3] = 4; break;
#ifdef FA2_EXPERIMENTS
    case FA2_VARIANT_MFMA8U: out4[1] = 128; out4[2] = 64; out4[3] = 4; break;
#endif
    case FA2_VARIANT_MFMA16K: out4[1] = 128; out4[2] = 64; out4[3] = 8; break;
    case FA2_VARIANT_MFMA16K_R2K2: out4[1] = 64; out4[2] = 64; out4[3] = 4; break;
    case FA2_VARIANT_MFMA16K_R2K4: out4[1] = 64; out4[2] = 64; out4[3] = 8; break;
    case FA2_VARIANT_MFMA16D: out4[1] = 256; out4[2] = 32; out4[3] = 8; break;
    case FA2_VARIANT_MFMA16D_W4: out4[1] = 128; out4[2] = 32; out4[3] = 4; break;
    case FA2_VARIANT_MFMA16H: out4[1] = 256; out4[2] = 32; out4[3] = 8; break;
    case FA2_VARIANT_MFMA16H_W4: out4[1] = 128; out4[2] = 32; out4[3] = 4; break;
#ifdef FA2_EXPERIMENTS
    case FA2_VARIANT_MFMA16S: out4[1] = 256; out4[2] = 32; out4[3] = 8; break;
    case FA2_VARIANT_MFMA16S_W4: out4[1] = 128; out4[2] = 32; out4[3] = 4; break;
#endif
    case FA2_VARIANT_MFMA32: out4[1] = 128; out4[2] = 32; out4[3] = 4; break;
    default: out4[1] = 16; out4[2] = 64; out4[3] = 4; break;
    }
    return FA2_OK;
}

const char *fa2_version(void) { return "fa2-hip 0.2.0 gfx950"; }

const char *fa2_last_error(void) { return g_err; }

}  // extern "C"
