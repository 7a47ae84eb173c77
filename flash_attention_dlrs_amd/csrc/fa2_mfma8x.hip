// fa2_mfma8x.hip -- FA-2 forward for OCP fp8 (e4m3fn and e5m2), d = 128, on the DOUBLE-RATE fp8 matrix path of gfx950:
// v_mfma_f32_32x32x64_f8f6f4 (variant "mfma8x"; BASELINE.json config c5).  fa2_mfma8.hip uses the 32x32x16 fp8 form,
// which costs the cycles of the bf16 form (2.5 PF peak); the k = 64 form does four times the k in twice the cycles
// (MI355X_MICROARCH.md, dense peaks table: ~5 PF).  Same arithmetic as every other kernel here
// (src/flash_attention_kernels.py:84-108): fp32 S, m, l, O; P rounded RTNE to fp8 before P.V (:98); O / l and L
// rounded to fp8 on store (:107-108).  The builtin is the block-scaled one with both scales = literal 0, which hipcc
// selects into the unscaled v_mfma_f32_32x32x64_f8f6f4.
//
// What k = 64 changes against fa2_mfma8.hip:
//   * S^T = K_blk . Q^T over d = 128 is TWO MFMAs per 32-key block: a lane (row i, half h) supplies 32 bytes per
//     step s -- the 16-byte chunks 4s + h and 4s + 2 + h of its row (two ds_read_b128; Q held with the same mapping;
//     k is only a summation index, so any assignment works as long as both operands use it).
//   * O^T += V^T . P^T sums over 64 KEYS per MFMA, so the pipeline runs in 64-key units: S of unit t+1 (four MFMAs,
//     two 32-key blocks) is computed under the softmax of unit t; the running max is decided per UNIT, before either
//     block of P is rounded; then four MFMAs (one per 32-column block of O^T) consume the unit's P.  A lane's 32 P
//     values are its own S results of the two blocks (register r of block b = key 32b + (r&3) + 8(r>>2) + 4h is k slot
//     16b + r): no lane exchange.  The matching V^T operand is four ds_read_b64_tr_b8 (8 keys each, the lane map of
//     fa2_mfma8.hip) concatenated in the same (b, r) order.
//   * K tiles are aligned with V tiles (keys 64t .. 64t+63), no 32-key offset.
// LDS images, DMA staging, causal tile pairs, launch order and epilogue are those of fa2_mfma8.hip.
// rocprofv3 on the c5 per-GPU shape (profiles/r01/c5_per_gpu_mfma8x_rocprof.json): 2.23 GHz, matrix pipe busy 49 % -- unlike the
// bf16 kernels this one is not at the power limit but VALU-issue bound (32 fma + 32 exp2 + 16 max3 + 16 cvt per 9 MFMAs).
// Measured and dropped: (1) staging two tiles ahead through rings of three K / V buffers with a counted vmcnt: -3 %,
// the kernel is not waiting for its DMA; (2) running the P.V of unit t one iteration late so that all nine MFMAs are
// independent of the softmax in flight (two-deep rescale queue, three V buffers; correct, parity green): -10 % with
// the compiler's schedule, which clusters the MFMAs instead of spreading them through the exp2 stream -- it needs the
// hand-ordered sub-steps of fa2_mfma16h.hip to pay.
#include "fa2_common.h"

#ifndef FA2_8X_MSUM
#define FA2_8X_MSUM 1  // row sums by a 16x16x128 MFMA against a 0/1 operand instead of v_add_f32
#endif

#ifndef FA2_8X_ABL
#define FA2_8X_ABL 0  // 1..5: timing-only ablations (WRONG results), see DESIGN.md section 5
#endif

#ifdef FA2_STAMPS
// Diagnostic build only: s_memtime sums of workgroup 0 per wave -- [0] iteration body, [1] DMA wait, [2] barrier wait,
// [15] iterations (steady loop only).  benchmarks/stamps.py <config> mfma8x.
__device__ unsigned long long fa2_stamp_buf[8][16];
extern "C" int fa2_debug_read_stamps(unsigned long long *host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fa2_stamp_buf), sizeof(fa2_stamp_buf));
}
#define STAMP(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define STAMP(k) do { } while (0)
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(8))) int i32x8;

#define LDS_PTR(T) __attribute__((address_space(3))) T *

// E4M3 = true: OCP e4m3fn (v_mfma ... fp8_fp8, v_cvt_pk_fp8_f32); false: e5m2 (bf8).
template <bool E4M3> struct F8 {
    static __device__ __forceinline__ f32x16 mfma(i32x8 a, i32x8 b, f32x16 c) {
        if constexpr (E4M3) return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
        else return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4 mfma_sum(i32x8 a, i32x8 b, f32x4 c) {  // 16x16x128
        if constexpr (E4M3) return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
        else return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 1, 1, 0, 0, 0, 0);
    }
    static constexpr int kOnes = E4M3 ? 0x38383838 : 0x3c3c3c3c;  // four fp8 1.0
    // two floats -> two fp8 (RTNE) into the low (hi = false) or high half of `old`
    template <bool HI> static __device__ __forceinline__ int cvt_pk(float x, float y, int old) {
        if constexpr (E4M3) return __builtin_amdgcn_cvt_pk_fp8_f32(x, y, old, HI);
        else return __builtin_amdgcn_cvt_pk_bf8_f32(x, y, old, HI);
    }
};

struct F8Args {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // B, H, N strides in bytes
    int64_t ls[2];
    int B, H, N;
    float c_log2e;
    int group;
    float thr;  // deferral threshold of the running maximum, log2 units (launch_t)
};

__device__ __forceinline__ void half_swap(float x, float &lo, float &hi) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned a = r[0], b = r[1];  // scalars first: bit_cast on a vector element reads element 0 (clang bug)
    lo = __builtin_bit_cast(float, a);
    hi = __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float half_swap_max(float x) {
    float lo, hi;
    half_swap(x, lo, hi);
    return fmaxf(lo, hi);
}
__device__ __forceinline__ float half_swap_sum(float x) {
    float lo, hi;
    half_swap(x, lo, hi);
    return lo + hi;
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

// One LDS-DMA piece: 64 lanes x 16 bytes from (descriptor, per-lane byte offset) to LDS at lds_base + lane*16.
// Inline asm ON PURPOSE: with the builtin, hipcc cannot tell the DMA's destination buffer from the buffer being
// read and puts `s_waitcnt vmcnt(0)` in front of the first ds_read of the V tile -- the transfer then has a
// quarter of an iteration to land instead of a whole one.  The compiler does not see these loads: the
// `s_waitcnt vmcnt(0)` in front of the publishing barrier is ours (dma_wait()).  M0 is saved and restored.
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds_base, int voffset) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(voffset), "s"(rsrc)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// K image: 16-byte chunk c of row r at r*128 + ((c ^ g(r)) << 4); g is a bijection of (r >> 1) & 7 together with
// r & 1 selecting the half of the 256-byte bank row: 16 rows distinct mod 16 -> 16 distinct slots (ds_read_b128).
__device__ __forceinline__ int swz_k(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
// V image: 32-byte unit u of row r at r*128 + ((u ^ f(r)) << 5).
__device__ __forceinline__ int swz_v(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

// WPS = waves per SIMD the register budget is cut for: 2 = the software-pipelined loop (S of unit t+1 under the softmax
// of unit t); 3 = one unit at a time (S, softmax, P.V in sequence, <= 168 registers), three workgroups of four waves per
// CU, the overlap left to the hardware's choice among three instruction streams (variant "mfma8u").
template <bool E4M3, int NW, bool CAUSAL, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void fa2_fwd_mfma8x_kernel(const F8Args a) {
    using M = F8<E4M3>;
    constexpr int D = 128, BR = NW * 32;
    constexpr int ROWB = D;                        // one byte per element
    constexpr int TILEB = 64 * ROWB;               // 8 KiB
    constexpr int RPP = 1024 / ROWB;               // 8 rows per 1-KiB DMA piece
    constexpr int PIECES = TILEB / 1024, PPW = PIECES / NW;
    constexpr int VBASE = 2 * TILEB;               // LDS: Kunit0 | Kunit1 | Vtile0 | Vtile1 (32 KiB)
    constexpr int KP = D / 32, DB = D / 32;        // 16-byte chunk pairs of a K row (k_off), 32-row blocks of O^T
    static_assert(PPW >= 1, "too many waves for this tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;

    const int nq = (N + BR - 1) / BR, nbh = a.B * a.H;
    const int nunit = CAUSAL ? (nq + 1) / 2 : nq;  // causal: one workgroup per Q-tile pair (see fa2_mfma16d.hip)
    int bh, unit;
    {
        const int bid = blockIdx.x;
        if ((nbh & 7) == 0) {
            const int slot = bid >> 3, G = a.group;
            const int batch = slot / (G * nunit), r = slot - batch * (G * nunit);
            bh = (batch * G + r % G) * 8 + (bid & 7);
            unit = r / G;
        } else {
            bh = bid / nunit;
            unit = bid % nunit;
        }
    }
    const int qi_first = CAUSAL ? nq - 1 - unit : unit, qi_second = unit;
    const int npass = (CAUSAL && qi_second != qi_first) ? 2 : 1;
    const int b = bh / a.H, hh = bh - b * a.H;
    int q0 = 0, qrow = 0;

    const char *Qp = a.Q + (int64_t)b * a.qs[0] + (int64_t)hh * a.qs[1];
    const char *Kp = a.K + (int64_t)b * a.ks[0] + (int64_t)hh * a.ks[1];
    const char *Vp = a.V + (int64_t)b * a.vs[0] + (int64_t)hh * a.vs[1];

    i32x8 qf[2];  // k-step s: Q[row][16 (4s + h) ..+15] | Q[row][16 (4s + 2 + h) ..+15]

    // ---- DMA staging: piece p = rows 8p..8p+7; lane l fills (row 8p + l/8, 16-byte slot l%8)
    const int krs = (int)a.ks[2], vrs = (int)a.vs[2];
    auto make_rsrc = [&](const char *base, int bytes) {
        const uint64_t ba = (uint64_t)base;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)ba);
        r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(ba >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane(bytes);
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 krsrc = make_rsrc(Kp, (N - 1) * krs + ROWB);
    const i32x4 vrsrc = make_rsrc(Vp, (N - 1) * vrs + ROWB);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    int kvo[PPW], vvo[PPW];
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
        const int row = RPP * (wave + pp * NW) + (lane >> 3), slot = lane & 7;
        kvo[pp] = row * krs + ((slot ^ swz_k(row)) << 4);
        vvo[pp] = row * vrs + ((((slot >> 1) ^ swz_v(row)) << 5) | ((slot & 1) << 4));
    }
    auto dma_k = [&](int u, int buf) {  // K tile u = keys 64u .. 64u+63
        const int base = u * 64 * krs;
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) dma16(krsrc, lds_base + buf * TILEB + (wave + pp * NW) * 1024, kvo[pp] + base);
    };
    auto dma_v = [&](int t, int buf) {
        const int base = t * 64 * vrs;
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
            dma16(vrsrc, lds_base + VBASE + buf * TILEB + (wave + pp * NW) * 1024, vvo[pp] + base);
    };

    int kend = 0, nt = 0, nu = 0;

    // ---- per-lane read offsets
    int k_off[KP];  // K row read: row (half*32 + i), 16-byte chunk 2s + h
#pragma unroll
    for (int sp = 0; sp < KP; ++sp) k_off[sp] = i * ROWB + (((2 * sp + h) ^ swz_k(i)) << 4);
    // V transposed read: in its 16-lane group (w = column half of the 32-column block, h = key half) lane idx
    // addresses key_e (e = idx >> 1) = 16ss + 8(e>>2) + 4h + (e&3), columns 32db + 16w + 8(idx&1) .. +7
    int v_off[DB];
    {
        const int w = (lane >> 4) & 1, idx = lane & 15, e = idx >> 1, p8 = idx & 1;
        const int key = 8 * (e >> 2) + 4 * h + (e & 3);  // + 16 ss + block base (multiples of 16: swizzle unchanged)
#pragma unroll
        for (int db = 0; db < DB; ++db) v_off[db] = VBASE + key * ROWB + ((db ^ swz_v(key)) << 5) + 16 * w + 8 * p8;
    }

    f32x16 o[DB];
    float m = -INFINITY;
#if !FA2_8X_MSUM
    float lsum = 0.0f;
#else
    // Row sums on the matrix pipe: one v_mfma_f32_16x16x128_f8f6f4 per 64-key unit with a constant 0/1 A operand
    // instead of 32 v_add_f32 (this kernel is VALU-issue bound).  The P^T fragment (lane = query lane & 31, 32 keys of
    // half lane >> 5) read as a 16x16x128 B operand is column n = lane & 15, k group g = lane >> 4: groups 0 / 2 are
    // the two key halves of query n, groups 1 / 3 those of query n + 16.  A row m sums groups {0, 2} for m = 0, 8 and
    // {1, 3} for m = 4, 12, so register 0 of the result (row 4 (lane >> 4), column lane & 15) is the complete 64-key
    // sum of the lane's OWN query in all 64 lanes.  The sum is over P as rounded to fp8 -- the values P.V consumes.
    f32x4 lacc = {0.0f, 0.0f, 0.0f, 0.0f};
    i32x8 ones;
    {
        const int mm = lane & 15, gg = lane >> 4;
        const bool on = ((mm & 7) == 0 && (gg & 1) == 0) || ((mm & 7) == 4 && (gg & 1) == 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) ones[e] = on ? M::kOnes : 0;
    }
#endif
    const float c = a.c_log2e;
    const float kThr = a.thr;  // P <= 2^thr before the running max is raised (launch: 6; e4m3 tops out at 448 = 2^8.8)

    auto qk = [&](f32x16 &s, int koff) __attribute__((always_inline)) {  // koff = K buffer base + block * 32 rows
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const u32x4 ka = *(LDS_PTR(u32x4))(lds + koff + k_off[2 * st]);
            const u32x4 kb = *(LDS_PTR(u32x4))(lds + koff + k_off[2 * st + 1]);
            i32x8 kf;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                kf[e] = (int)ka[e];
                kf[4 + e] = (int)kb[e];
            }
            s = M::mfma(kf, qf[st], s);
        }
    };
    // running max over one 64-key unit u (blocks 2u, 2u+1), decided BEFORE either block of P is rounded
    auto partial = [&](f32x16 &s0, f32x16 &s1, int u, float &coeff, bool masked) __attribute__((always_inline)) -> bool {
        if (masked) {
            int lim = N - 1;
            if (CAUSAL) lim = qrow < lim ? qrow : lim;
            const int klim = lim - (u * 64 + 4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((r & 3) + 8 * (r >> 2) > klim) s0[r] = -INFINITY;
                if ((r & 3) + 8 * (r >> 2) + 32 > klim) s1[r] = -INFINITY;
            }
        }
        float mx = fmaxf(s0[0], s1[0]);
#if FA2_8X_ABL != 3   /* 3: no running max */
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);  // one v_max3_f32 per pair (+3 %)
#endif
        mx = half_swap_max(mx) * c;
        const bool fire = !__all(mx - m <= kThr);
        coeff = 1.0f;
        if (fire) {
            const float m_new = fmaxf(m, mx);
            coeff = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
        }
        return fire;
    };
    // P = exp2(S*c - m), P -> fp8 RTNE (kernels.py:94-98), row sum (of the rounded P with FA2_8X_MSUM, else unrounded); dword 4b + 2ss + e of pf =
    // registers 8ss + 4e .. +3 of block b
    auto finish = [&](f32x16 &s0, f32x16 &s1, i32x8 &pf) __attribute__((always_inline)) {
#if !FA2_8X_MSUM
        float rs0 = 0.0f, rs1 = 0.0f;
#endif
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            f32x16 &s = b ? s1 : s0;
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#if FA2_8X_ABL == 1   /* no exp2 */
                p[r] = __builtin_fmaf(s[r], c, -m);
#elif FA2_8X_ABL == 2 /* no fma, no exp2 */
                p[r] = s[r];
#else
                p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -m));
#endif
#if !FA2_8X_MSUM
                if (r & 1) rs1 += p[r];
                else rs0 += p[r];
#endif
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int w = M::template cvt_pk<false>(p[4 * q + 0], p[4 * q + 1], 0);
                w = M::template cvt_pk<true>(p[4 * q + 2], p[4 * q + 3], w);
                pf[4 * b + q] = w;
            }
        }
#if FA2_8X_MSUM
        lacc = M::mfma_sum(ones, pf, lacc);
#else
        lsum += rs0 + rs1;
#endif
    };
    auto rescale = [&](bool fire, float coeff) __attribute__((always_inline)) {
        if (fire) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float x = o[db][r];
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(coeff));
                    o[db][r] = x;
                }
            asm volatile("s_nop 7" ::: "memory");
#if FA2_8X_MSUM
            lacc[0] *= coeff;
#else
            lsum *= coeff;
#endif
        }
    };
    auto pv = [&](const i32x8 &pf, int voff) __attribute__((always_inline)) {  // voff = V buffer base; 64 keys
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            i32x8 va;
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // q = 2b + ss: keys 32b + 16ss + ...
                const i32x2 vf = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDS_PTR(i32x2))(lds + voff + q * 16 * ROWB + v_off[db]));
                va[2 * q] = vf[0];
                va[2 * q + 1] = vf[1];
            }
            o[db] = M::mfma(va, pf, o[db]);
        }
    };
    auto unit_masked = [&](int u) __attribute__((always_inline)) { return (CAUSAL && (u * 64 + 63 > q0)) || (u * 64 + 64 > N); };

#ifdef FA2_STAMPS
    unsigned long long st_acc[16] = {0}, st_last = 0;
#endif
    for (int pass = 0; pass < npass; ++pass) {
        const int qi = pass == 0 ? qi_first : qi_second;
        q0 = qi * BR + wave * 32;
        qrow = q0 + i;
        {
            const int row = qrow < N ? qrow : N - 1;
            const char *qp = Qp + (int64_t)row * a.qs[2] + h * 16;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const u32x4 qa = *(const u32x4 *)(qp + st * 64), qb = *(const u32x4 *)(qp + st * 64 + 32);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    qf[st][e] = (int)qa[e];
                    qf[st][4 + e] = (int)qb[e];
                }
            }
        }
        kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
        nt = (kend + 63) >> 6;  // 64-key units of this tile
        nu = nt;                // ... of this wave (causal: up to the unit holding its diagonal)
        if (CAUSAL) nu = ((q0 + 31) >> 6) + 1 < nt ? ((q0 + 31) >> 6) + 1 : nt;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
        m = -INFINITY;
#if FA2_8X_MSUM
        lacc[0] = lacc[1] = lacc[2] = lacc[3] = 0.0f;
#else
        lsum = 0.0f;
#endif

        if constexpr (WPS == 3) {
            dma_k(0, 0);
            dma_v(0, 0);
            dma_wait();
            __syncthreads();
            f32x16 sA, sB;
            float coeff = 1.0f;
            i32x8 pf;
            for (int t = 0; t < nt; ++t) {  // K tile t and V tile t both live in buffer t & 1
                if (t + 1 < nt) {
                    dma_k(t + 1, (t + 1) & 1);
                    dma_v(t + 1, (t + 1) & 1);
                }
                if (t < nu) {
                    const int cur = (t & 1) * TILEB;
                    qk(sA, cur);
                    qk(sB, cur + 32 * ROWB);
                    const bool fire = partial(sA, sB, t, coeff, unit_masked(t));
                    rescale(fire, coeff);
                    finish(sA, sB, pf);
                    pv(pf, cur);
                }
                dma_wait();
                __syncthreads();
            }
        } else {
            dma_k(0, 0);
            dma_v(0, 0);
            dma_k(1, 1);
            dma_wait();
            __syncthreads();

            f32x16 sA, sB, sC, sD;
            float coeff = 1.0f;
            bool fire = false;
            i32x8 pf;
            qk(sA, 0);  // unit 0 = K tile 0
            qk(sB, 32 * ROWB);
            fire = partial(sA, sB, 0, coeff, unit_masked(0));
            __syncthreads();  // K tile 0 is overwritten by tile 2 in iteration 0

            // iteration t: DMA of K tile t+2 / V tile t+1; S of unit t+1 (K tile t+1) under the softmax of unit t; P.V of
            // unit t (V tile t); running max of unit t+1.  Steady iterations: units t and t+1 need no mask for this wave.
            int n_free = N >> 6;  // leading units without any masked element
            if (CAUSAL) n_free = ((q0 + 1) >> 6) < n_free ? ((q0 + 1) >> 6) : n_free;
            int t_steady = n_free - 1;
            t_steady = t_steady < 0 ? 0 : (t_steady > nt - 1 ? nt - 1 : t_steady);

            auto steady = [&](int t, f32x16 &c0, f32x16 &c1, f32x16 &n0, f32x16 &n1) __attribute__((always_inline)) {
                dma_k(t + 2, t & 1);
                dma_v(t + 1, (t + 1) & 1);
                const int knext = ((t + 1) & 1) * TILEB, vcur = (t & 1) * TILEB;
                rescale(fire, coeff);
                qk(n0, knext);
                qk(n1, knext + 32 * ROWB);
                finish(c0, c1, pf);
#if FA2_8X_ABL != 5   /* 5: no P.V */
                pv(pf, vcur);
#else
                asm volatile("" ::"v"(pf));
#endif
                fire = partial(n0, n1, t + 1, coeff, false);
                STAMP(0);
                dma_wait();
                STAMP(1);
#if FA2_8X_ABL != 4   /* 4: no barrier in the steady loop */
                __syncthreads();
#endif
                STAMP(2);
#ifdef FA2_STAMPS
                st_acc[15] += 1;
#endif
            };
            auto guarded = [&](int t, f32x16 &c0, f32x16 &c1, f32x16 &n0, f32x16 &n1) __attribute__((always_inline)) {
                if (t + 1 < nt) {
                    dma_k(t + 2, t & 1);
                    dma_v(t + 1, (t + 1) & 1);
                }
                const int knext = ((t + 1) & 1) * TILEB, vcur = (t & 1) * TILEB;
                const bool cur = t < nu, nxt = t + 1 < nu;
                if (cur) rescale(fire, coeff);
                if (nxt) {
                    qk(n0, knext);
                    qk(n1, knext + 32 * ROWB);
                }
                if (cur) {
                    finish(c0, c1, pf);
                    pv(pf, vcur);
                }
                if (nxt) fire = partial(n0, n1, t + 1, coeff, unit_masked(t + 1));
                dma_wait();
                __syncthreads();
            };
            int t = 0;
#ifdef FA2_STAMPS
            st_last = __builtin_amdgcn_s_memtime();
#endif
            for (; t + 1 < t_steady; t += 2) {
                steady(t, sA, sB, sC, sD);
                steady(t + 1, sC, sD, sA, sB);
            }
            for (; t < nt; t += 2) {  // t is even: unit t sits in (sA, sB)
                guarded(t, sA, sB, sC, sD);
                if (t + 1 < nt) guarded(t + 1, sC, sD, sA, sB);
            }
        }

        // ---- epilogue: O = O / l and L = m + log2 l, both rounded to fp8 (kernels.py:105-108).  Lane (i, h) owns
        // row q0+i, columns 32db + 8g + 4h .. +3 (four fp8 = one dword); the wave's 32 x 128-byte tile goes through
        // its own 4-KiB slice of the idle K/V buffers and leaves as whole rows (see fa2_mfma16d.hip).
#if FA2_8X_MSUM
        const float l = lacc[0];
#else
        const float l = half_swap_sum(lsum);
#endif
        const float inv = 1.0f / l;
        {
            const int ebase = wave * 32 * ROWB;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int w = M::template cvt_pk<false>(o[db][4 * g + 0] * inv, o[db][4 * g + 1] * inv, 0);
                    w = M::template cvt_pk<true>(o[db][4 * g + 2] * inv, o[db][4 * g + 3] * inv, w);
                    const int ch = 2 * db + (g >> 1);
                    *(LDS_PTR(int))(lds + ebase + i * ROWB + ((ch ^ swz_k(i)) << 4) + 8 * (g & 1) + 4 * h) = w;
                }
            const int er = lane >> 3, ec = lane & 7;  // 8 rows x 8 chunks per store instruction
            char *ob = a.O + (int64_t)b * a.os[0] + (int64_t)hh * a.os[1];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = k * 8 + er;
                const u32x4 val = *(LDS_PTR(u32x4))(lds + ebase + r * ROWB + ((ec ^ swz_k(r)) << 4));
                if (q0 + r < N) *(u32x4 *)(ob + (int64_t)(q0 + r) * a.os[2] + ec * 16) = val;
            }
        }
        if (qrow < N && h == 0) {
            const int w = M::template cvt_pk<false>(m + __builtin_amdgcn_logf(l), 0.0f, 0);
            a.L[b * a.ls[0] + hh * a.ls[1] + qrow] = (char)(w & 0xff);
        }
        if (pass + 1 < npass) __syncthreads();  // the next pass's DMA reuses the slices
    }  // pass
#ifdef FA2_STAMPS
    if (blockIdx.x == 0 && lane == 0)
        for (int k = 0; k < 16; ++k) fa2_stamp_buf[wave][k] = st_acc[k];
#endif
}

template <bool E4M3, int NW, int WPS = 2> int launch_t(const Fa2Problem &p, const F8Args &a) {
    constexpr int BR = NW * 32;
    const int nq = (p.N + BR - 1) / BR;
    const long long nblk = (long long)(p.causal ? (nq + 1) / 2 : nq) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma8x: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(NW * 64);
    constexpr size_t smem = 4 * 64 * 128;
    if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma8x_kernel<E4M3, NW, true, WPS>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma8x_kernel<E4M3, NW, false, WPS>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma8x kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

}  // namespace

bool fa2_mfma8x_supports(const Fa2Problem &p) {
    if (p.dtype != FA2_DTYPE_F8E4M3 && p.dtype != FA2_DTYPE_F8E5M2) return false;
    if (p.d != 128) return false;
    if (!(p.scale > 0.0f) || !(p.scale < INFINITY)) return false;
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1 || p.os[3] != 1) return false;
    for (int k = 0; k < 3; ++k)
        if ((p.qs[k] & 15) || (p.ks[k] & 15) || (p.vs[k] & 15) || (p.os[k] & 15)) return false;
    if (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.O) & 15) return false;
    if ((int64_t)(p.N + 512) * p.ks[2] >= (1LL << 31) || (int64_t)(p.N + 512) * p.vs[2] >= (1LL << 31)) return false;
    return true;
}

int fa2_launch_mfma8x(const Fa2Problem &p, int waves) {
    if (!fa2_mfma8x_supports(p)) {
        fa2_set_error("mfma8x kernel: needs fp8 (e4m3fn / e5m2), d = 128, unit d-stride, 16-byte aligned rows, scale > 0");
        return FA2_ERR_UNSUPPORTED;
    }
    F8Args a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V;
    a.O = (char *)p.O; a.L = (char *)p.L;
    for (int k = 0; k < 3; ++k) { a.qs[k] = p.qs[k]; a.ks[k] = p.ks[k]; a.vs[k] = p.vs[k]; a.os[k] = p.os[k]; }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    a.group = 1;
    // Deferral threshold of the running maximum: P = exp2(s c - m) may reach 2^thr before m is raised and O, l are rescaled.  fp8 is a
    // floating-point format: a larger P loses nothing, and entries far below the maximum keep MORE of their bits while m lags.  6
    // until round 3; 8.5 (e4m3 tops out at 448 = 2^8.8) is +1.0 % on c5's shard with N(0, 1) inputs, +-0 on N(0, 1/4)
    // (profiles/r03/fp8_threshold_ab.jsonl)
    a.thr = p.dtype == FA2_DTYPE_F8E4M3 ? 8.5f : 15.0f;
    if (p.causal && ((p.B * p.H) & 7) == 0) {
        const int per_xcd = p.B * p.H / 8;
        int g = fa2_env_int("FA2_CAUSAL_GROUP", 2);
        g = g < 1 ? 1 : (g > per_xcd ? per_xcd : g);
        while (per_xcd % g) --g;
        a.group = g;
    }
    const bool e4 = p.dtype == FA2_DTYPE_F8E4M3;
    if (waves == 8) return e4 ? launch_t<true, 8>(p, a) : launch_t<false, 8>(p, a);
    if (waves == 12) return e4 ? launch_t<true, 4, 3>(p, a) : launch_t<false, 4, 3>(p, a);  // "mfma8u"
    return e4 ? launch_t<true, 4>(p, a) : launch_t<false, 4>(p, a);
}
