// fa2_mfma16x.hip -- FA-2 forward for f16 / bf16, d = 128, one wave per SIMD with the whole register file
// (variant "mfma16x": 4 waves x 64 query rows, 256-row Q tile per workgroup, one workgroup per CU).
//
// Arithmetic: the reference's (src/flash_attention_kernels.py:84-108), as in fa2_mfma16p.hip: exp2-domain
// online softmax in fp32, P rounded RTNE to the I/O dtype before P.V, deferred running-max update.
//
// Why this shape.  Ablations of fa2_mfma16p.hip on MI355X (8 waves x 32 rows, two waves per SIMD) showed
// that with ALL softmax arithmetic removed the MFMA + LDS-operand skeleton still stops at 52 % of the bf16
// peak: every 32x32x16 MFMA pulls its A operand (a K row fragment or a transposed V fragment) from LDS,
// 1.5 LDS instructions per MFMA per wave, 8 waves per CU.  Here a wave owns TWO 32-row query blocks and
// every K / V fragment read from LDS feeds two MFMAs (one per query block), halving LDS traffic and LDS
// instruction issue per FLOP; with one wave per SIMD the kernel may use up to 512 registers, enough for
// O (128) + Q (64) + two score tiles per query block (64) + staging and fragments.
//
// Schedule per 32-key block j (skewed by one block as in fa2_mfma16p.hip), both query blocks in step:
//     phase 1:  S_next[qb] = K_blk(j+1) . Q[qb]^T   (16 MFMA)  ||  P_j[qb] = exp2(S_j[qb]*c - m[qb]), row sums, cvt
//     phase 2:  O[qb]     += V_blk(j)^T . P_j[qb]^T (16 MFMA)  ||  row max of S_next[qb], rescale decision
// K is staged in 64-row units offset by 32 rows against V (unit u = keys 64u-32 .. 64u+31), three K units and two
// V tiles in LDS, filled by LDS-DMA (buffer_load ... lds, inline asm as in fa2_mfma16d.hip: source-side XOR swizzle of
// the 16-byte chunks, conflict-free row and transposed reads), one DMA piece per even step of the first half
// iteration, one barrier per 64 keys.
#include "fa2_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(T) __attribute__((address_space(3))) T *

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
    using frag = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<_Float16> {
    using frag = f16x8;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

struct XArgs {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // B, H, N strides in bytes
    int64_t ls[2];
    int B, H, N;
    float c_log2e;
    int group;
};

// v_permlane32_swap exchange between the wave's two 32-lane halves (see fa2_mfma16p.hip for the
// __builtin_bit_cast-on-a-vector-element pitfall that the scalar copies avoid).
__device__ __forceinline__ void half_swap(float x, float &lo, float &hi) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned a = r[0], b = r[1];
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
// one 1-KiB LDS-DMA piece (see fa2_mfma16d.hip: inline asm on purpose; our own vmcnt wait before the barrier)
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds_base, int voffset) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(voffset), "s"(rsrc)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// 16-byte chunk swizzle of a 256-byte row (function of row & 15), fa2_mfma16.hip lds_off()
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int lds_off(int row, int ch) { return row * 256 + ((ch ^ swz(row)) << 4); }

// ABL (timing-only ablations, -DFA2_ABLATIONS builds): 1 = no exp/sum, 2 = no row max, 4 = no LDS operand reads in
// the steady loop, 8 = no staging (global loads, LDS writes, barrier) in the steady loop.
template <typename T, bool CAUSAL, int ABL>
__global__ __launch_bounds__(256, 1) void fa2_fwd_mfma16x_kernel(const XArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int D = 128, BR = 256, QB = 2;
    constexpr int ROWB = D * 2;
    constexpr int KROWB = ROWB, VROWB = ROWB;  // plain rows: the DMA writes lane-linearly, swizzle instead of padding
    constexpr int KUNIT = 64 * KROWB, VTILE = 64 * VROWB;
    constexpr int NKB = 3;            // K units in LDS: the one in use, the next (read ahead), the one being written
    constexpr int VBASE = NKB * KUNIT;  // LDS: Kunit x3 | Vtile0 | Vtile1
    constexpr int KS = D / 16, DB = D / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;

    const int nq = (N + BR - 1) / BR, nbh = a.B * a.H;
    int bh, qi;
    {
        const int bid = blockIdx.x;
        if ((nbh & 7) == 0) {  // whole (b, h) groups per XCD: K/V reuse in that XCD's L2 (speed only)
            const int slot = bid >> 3, G = a.group;
            const int batch = slot / (G * nq), r = slot - batch * (G * nq);
            bh = (batch * G + r % G) * 8 + (bid & 7);
            qi = r / G;
        } else {
            bh = bid / nq;
            qi = bid % nq;
        }
        if (CAUSAL) qi = nq - 1 - qi;  // heaviest tiles first
    }
    const int b = bh / a.H, hh = bh - b * a.H;
    const int q0 = qi * BR + wave * 64;  // first row of this wave; query block qb covers q0 + 32 qb .. +31

    const char *Qp = a.Q + (int64_t)b * a.qs[0] + (int64_t)hh * a.qs[1];
    const char *Kp = a.K + (int64_t)b * a.ks[0] + (int64_t)hh * a.ks[1];
    const char *Vp = a.V + (int64_t)b * a.vs[0] + (int64_t)hh * a.vs[1];

    frag qf[QB][KS];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        int row = q0 + 32 * qb + i;
        row = row < N ? row : N - 1;
        const char *qp = Qp + (int64_t)row * a.qs[2] + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[qb][ks] = __builtin_bit_cast(frag, *(const u32x4 *)(qp + ks * 32));
    }

    // ---- LDS-DMA staging: piece p of a unit/tile = rows 4p .. 4p+3 (1 KiB); wave w issues pieces w, w+4, w+8, w+12.
    // Lane l fills (row 4p + l/16, slot l%16) with global chunk slot ^ swz(row).  Rows past N and the "negative"
    // rows of K unit 0 are zero-filled by the descriptor's range check (offsets in the VGPR operand).
    constexpr int PPW = 4;
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
        const int row = 4 * (wave + 4 * pp) + (lane >> 4), slot = lane & 15;
        const int chunk = slot ^ swz(row);
        kvo[pp] = row * krs + chunk * 16;
        vvo[pp] = row * vrs + chunk * 16;
    }
    auto dma_k1 = [&](int pp, int u, int buf) {
        dma16(krsrc, lds_base + buf * KUNIT + (wave + 4 * pp) * 1024, kvo[pp] + (u * 64 - 32) * krs);
    };
    auto dma_v1 = [&](int pp, int t, int buf) {
        dma16(vrsrc, lds_base + VBASE + buf * VTILE + (wave + 4 * pp) * 1024, vvo[pp] + t * 64 * vrs);
    };
    auto dma_k = [&](int u, int buf) {
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) dma_k1(pp, u, buf);
    };
    auto dma_v = [&](int t, int buf) {
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) dma_v1(pp, t, buf);
    };

    const int kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
    const int nt = (kend + 63) >> 6;    // V tiles = loop iterations of this workgroup
    const int nblk = (kend + 31) >> 5;  // 32-key blocks of this workgroup
    int nb = nblk;                      // ... of this wave: up to the diagonal block of its second query block
    if (CAUSAL) nb = (q0 >> 5) + 2 < nblk ? (q0 >> 5) + 2 : nblk;

    int k_off[KS];  // K row read: row (half*32 + i), chunk 2ks + h
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_off[ks] = lds_off(i, 2 * ks + h);
    int v_off[2][DB];  // V transposed read (fa2_mfma16.hip): u = keys +0..3 / +8..11 of the 16-key step
    {
        const int w = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                v_off[u][db] = VBASE + lds_off(8 * u + 4 * h + qq, 4 * db + 2 * w + (pp >> 1)) + 8 * (pp & 1);
    }

    f32x16 o[QB][DB];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][db][r] = 0.0f;
    float m[QB] = {-INFINITY, -INFINITY}, lsum[QB] = {0.0f, 0.0f};
    const float c = a.c_log2e;
    // Rescale threshold in log2 units: P may reach 2^kThr before the running max is raised.  bf16 P has the
    // fp32 exponent range (24 leaves 2^24 * N far below fp32 overflow in l and O); f16 P must stay below 65504.
    // On N(0,1) inputs at scale 1 (score sigma ~ 16 log2 units) a threshold of 8 still fired ~20 times per wave
    // and 4096 keys -- each time the whole workgroup waits at the next barrier -- 24 makes it rare.
    constexpr float kThr = sizeof(T) == 2 && __is_same(T, _Float16) ? 12.0f : 24.0f;

    // Register-file steering (one wave per SIMD: 256 arch VGPRs + 256 AGPRs).  Q fragments are only ever MFMA
    // B operands, which may live in AGPRs; the score tiles are read by the VALU and must not.  Left alone
    // hipcc does the opposite (scores in AGPRs: one v_accvgpr_read per score element per block).
    auto pin_q_agpr = [&]() {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(qf[qb][ks]));
    };
    // S^T[qb] = K rows [koff ..] . Q[qb]^T : every K fragment feeds both query blocks
    auto qk = [&](f32x16 (&s)[QB], int koff) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[qb][r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const frag kf = __builtin_bit_cast(frag, *(LDS_PTR(u32x4))(lds + koff + k_off[ks]));
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) s[qb] = M::mfma(kf, qf[qb][ks], s[qb]);
        }
    };
    // row max of block j per query block, deferred running-max update (see fa2_mfma16p.hip); returns
    // whether either query block has to rescale (coeff[qb] == 1 for one that does not).
    auto partial = [&](f32x16 (&s)[QB], int j, float (&coeff)[QB], bool masked) -> bool {
        float mx[QB];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (masked) {
                int lim = N - 1;
                if (CAUSAL) {
                    const int qrow = q0 + 32 * qb + i;
                    lim = qrow < lim ? qrow : lim;
                }
                const int klim = lim - (j * 32 + 4 * h);  // key(r) = 32j + 4h + (r&3) + 8(r>>2)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) > klim) s[qb][r] = -INFINITY;
            }
            float x = fmaxf(s[qb][0], s[qb][1]);
#pragma unroll
            for (int r = 2; r < 16; ++r) x = fmaxf(x, s[qb][r]);
            mx[qb] = half_swap_max(x) * c;
        }
        // a fully masked block (query block 0 on the wave's last block) has mx = -inf: never fires
        const bool fire = !__all((mx[0] - m[0] <= kThr) && (mx[1] - m[1] <= kThr));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) coeff[qb] = 1.0f;
        if (fire) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float m_new = fmaxf(m[qb], mx[qb]);
                coeff[qb] = __builtin_amdgcn_exp2f(m[qb] - m_new);
                m[qb] = m_new;
            }
        }
        return fire;
    };
    auto finish = [&](f32x16 (&s)[QB], frag (&pf)[QB][2]) {
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            float rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qb][r], c, -m[qb]));
                rs += p;
                pf[qb][r >> 3][r & 7] = (T)p;
            }
            lsum[qb] += rs;
        }
    };
    // in-place O *= coeff, l *= coeff (rare path; asm keeps the accumulators where they are)
    auto rescale = [&](bool fire, float (&coeff)[QB]) {
        if (fire) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float x = o[qb][db][r];
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(coeff[qb]));
                        o[qb][db][r] = x;
                    }
                lsum[qb] *= coeff[qb];
            }
            asm volatile("s_nop 7" ::: "memory");
        }
    };
    // O^T[qb] += V[rows voff ..]^T . P[qb]^T : every V fragment feeds both query blocks
    auto pv = [&](frag (&pf)[QB][2], int voff) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int off = voff + ss * 16 * VROWB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off + v_off[0][db]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off + v_off[1][db]));
                const frag vf = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) o[qb][db] = M::mfma(vf, pf[qb][ss], o[qb][db]);
            }
    };
    auto block_masked = [&](int j) { return (CAUSAL && (j * 32 + 31 > q0)) || (j * 32 + 32 > N); };

    // ---- prologue: K units 0..2 and V tile 0
    dma_k(0, 0);
    dma_v(0, 0);
    dma_k(1, 1);
    dma_k(2, 2);
    dma_wait();
    __syncthreads();

    f32x16 sA[QB], sB[QB];
    float coeffA[QB] = {1.0f, 1.0f}, coeffB[QB] = {1.0f, 1.0f};
    bool fireA = false, fireB = false;
    frag pf[QB][2];
    qk(sA, 32 * KROWB);  // block 0 = rows 32..63 of K unit 0
    fireA = partial(sA, 0, coeffA, block_masked(0));
    __syncthreads();     // K unit 0 is overwritten by unit 3 in iteration 0

    int jm = nb;  // first block of this wave that needs a mask
    if (CAUSAL) jm = (q0 >> 5) < jm ? (q0 >> 5) : jm;
    if ((N >> 5) < jm) jm = N >> 5;
    int t_steady = (jm - 1) / 2;
    t_steady = t_steady < 0 ? 0 : (t_steady > nt ? nt : t_steady);

    // ---- hand-ordered steady-state half iteration: sixteen steps of two MFMAs (one K or V fragment feeding both
    // query blocks).  Every step also carries (a) the LDS read of the fragment needed TWO steps later, (b) its
    // share of the softmax arithmetic, (c) on even steps one staging operation (LDS write or global load of the
    // next K unit / V tile).  A sched_barrier after each step pins the order: left to itself hipcc issues the
    // 16 QK^T MFMAs back to back and the whole softmax after them, and with ONE wave per SIMD nothing else
    // covers for a stall.
    //   steps 0..7  (phase 1): S_nxt[qb] += K[ks] . Q[qb][ks]      exp/sum/cvt of 12 of the 16 element pairs of S_cur
    //   steps 8..15 (phase 2): O[qb][db] += V[ss][db] . P[qb][ss]  the last 4 pairs (steps 8..11), running max of S_nxt
    // Operand stream: ops 0..7 = K fragments, ops 8..15 = V fragments, ops 16, 17 = K fragments 0, 1 of the
    // FOLLOWING half; op n lives in ring[n & 3], step n consumes op n and issues the read of op n+2.
    auto lds_k = [&](int off) { return __builtin_bit_cast(frag, *(LDS_PTR(u32x4))(lds + off)); };
    auto lds_v = [&](int off, int db) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off + v_off[0][db]));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off + v_off[1][db]));
        return __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto half_iter = [&](f32x16 (&sCur)[QB], f32x16 (&sNxt)[QB], frag (&ring)[4], int koff, int voff, int knext,
                         float (&coeff)[QB], auto &&staging) -> bool {
        auto issue = [&](int n) {  // n is a compile-time constant at every call site (unrolled loops)
            if constexpr (ABL & 4) return;
            if (n < 8) ring[n & 3] = lds_k(koff + k_off[n & 7]);
            else if (n < 16) ring[n & 3] = lds_v(voff + ((n - 8) >> 2) * 16 * VROWB, (n - 8) & 3);
            else ring[n & 3] = lds_k(knext + k_off[(n - 16) & 7]);
        };
        constexpr int PD = 3;  // read-ahead distance in steps; the ring holds ops n .. n+PD
        float rs[QB] = {0.0f, 0.0f};
        // element pair q = 0..15: query block q & 1, elements 2(q>>1), 2(q>>1)+1.  Pairs 0..7 (P of keys 0..15,
        // both query blocks) are due before step 8, all of them before step 12.  The arithmetic of a pair is spread
        // over THREE consecutive steps -- F: t = s*c - m at step E-1, E: p = exp2(t) at step E, A/C: row sum and cvt at
        // step E+1 -- so that no instruction waits on the one issued just before it (in-kernel stamps on
        // fa2_mfma16h.hip: the fma, exp2, add sequence on one element made the phase dependency-latency bound, and
        // with ONE wave per SIMD nothing else covers such a stall).
        constexpr int ESTEP[16] = {0, 0, 1, 1, 2, 3, 3, 4, 5, 5, 6, 7, 7, 8, 9, 10};
        auto stageF = [&](int q) {
            const int qb = q & 1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * (q >> 1) + e;
                if constexpr (!(ABL & 1)) sCur[qb][r] = __builtin_fmaf(sCur[qb][r], c, -m[qb]);
            }
        };
        auto stageE = [&](int q) {
            const int qb = q & 1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * (q >> 1) + e;
                if constexpr (!(ABL & 1)) sCur[qb][r] = __builtin_amdgcn_exp2f(sCur[qb][r]);
            }
        };
        auto stageAC = [&](int q) {
            const int qb = q & 1;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int r = 2 * (q >> 1) + e;
                if constexpr (!(ABL & 1)) rs[qb] += sCur[qb][r];
                pf[qb][r >> 3][r & 7] = (T)sCur[qb][r];
            }
        };
        auto softmax_step = [&](int st) {  // st = 0..15, compile-time constant at every call site
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (ESTEP[q] == st + 1) stageF(q);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (ESTEP[q] == st) stageE(q);
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (ESTEP[q] == st - 1) stageAC(q);
        };
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (ESTEP[q] == 0) stageF(q);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            issue(ks + PD);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if (ks == 0) {
                    f32x16 z;
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                    sNxt[qb] = M::mfma(ring[0], qf[qb][0], z);
                } else {
                    sNxt[qb] = M::mfma(ring[ks & 3], qf[qb][ks], sNxt[qb]);
                }
            }
            softmax_step(ks);
            staging(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
        float mx[QB];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int ss = st >> 2, db = st & 3;
            issue(8 + st + PD);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) o[qb][db] = M::mfma(ring[(8 + st) & 3], pf[qb][ss], o[qb][db]);
            softmax_step(8 + st);
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                if constexpr (ABL & 2) {
                    if (st == 0) mx[qb] = sNxt[qb][0];
                } else {
                    const float x = fmaxf(sNxt[qb][2 * st], sNxt[qb][2 * st + 1]);
                    mx[qb] = st == 0 ? x : fmaxf(mx[qb], x);
                }
            }
            staging(8 + st);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) lsum[qb] += rs[qb];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) mx[qb] = half_swap_max(mx[qb]) * c;
        const bool fire = !__all((mx[0] - m[0] <= kThr) && (mx[1] - m[1] <= kThr));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) coeff[qb] = 1.0f;
        if (fire) {
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const float m_new = fmaxf(m[qb], mx[qb]);
                coeff[qb] = __builtin_amdgcn_exp2f(m[qb] - m_new);
                m[qb] = m_new;
            }
        }
        return fire;
    };

    // K unit u lives in ring buffer u % 3: kb1 -> unit t+1 (in use), kb2 -> unit t+2 (published, read ahead at the
    // end of the iteration), kb3 -> unit t+3 (written during iteration t from the staging registers).
    int t = 0, kb1 = 1, kb2 = 2, kb3 = 0;
    if (t_steady > 0) {
        frag ring[4];
        ring[0] = lds_k(KUNIT + k_off[0]);    // K fragments 0..2 of block 1 (rows 0..31 of K unit 1)
        ring[1] = lds_k(KUNIT + k_off[1]);
        ring[2] = lds_k(KUNIT + k_off[2]);
        for (; t < t_steady; ++t) {
            pin_q_agpr();
            const int kcur = kb1 * KUNIT;      // K unit t+1: rows 0..31 = block 2t+1, rows 32..63 = block 2t+2
            const int vcur = (t & 1) * VTILE;  // V tile t:   rows 0..31 = block 2t,   rows 32..63 = block 2t+1
            const int vwr = (t + 1) & 1;
            // first half: one DMA piece per even step -- K unit t+3 into the ring buffer released by the last barrier,
            // V tile t+1 into the other V buffer; they land during this iteration and are published by its barrier
            auto stage_wr = [&](int n) {
                if (n & 1) return;
                if (n < 8) dma_k1(n >> 1, t + 3, kb3);
                else dma_v1((n - 8) >> 1, t + 1, vwr);
            };
            auto stage_ld = [&](int) {};
            rescale(fireA, coeffA);
            fireB = half_iter(sA, sB, ring, kcur, vcur, kcur + 32 * KROWB, coeffB, stage_wr);
            rescale(fireB, coeffB);
            // ops 16, 17 of this half = K fragments 0, 1 of block 2t+3 = rows 0..31 of unit t+2 (published during
            // iteration t-1): the next iteration starts with its operands already in registers
            fireA = half_iter(sB, sA, ring, kcur + 32 * KROWB, vcur + 32 * VROWB, kb2 * KUNIT, coeffA, stage_ld);
            dma_wait();
            __syncthreads();
            const int k0 = kb1;
            kb1 = kb2;
            kb2 = kb3;
            kb3 = k0;
        }
    }
    for (; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if (more) {
            dma_k(t + 3, kb3);
            dma_v(t + 1, (t + 1) & 1);
        }
        const int kcur = kb1 * KUNIT;
        const int vcur = (t & 1) * VTILE;
        const int jA = 2 * t, jB = 2 * t + 1, jA2 = 2 * t + 2;
        if (jA < nb) rescale(fireA, coeffA);
        if (jB < nb) qk(sB, kcur);
        if (jA < nb) {
            finish(sA, pf);
            pv(pf, vcur);
        }
        if (jB < nb) {
            fireB = partial(sB, jB, coeffB, block_masked(jB));
            rescale(fireB, coeffB);
        }
        if (jA2 < nb) qk(sA, kcur + 32 * KROWB);
        if (jB < nb) {
            finish(sB, pf);
            pv(pf, vcur + 32 * VROWB);
        }
        if (jA2 < nb) fireA = partial(sA, jA2, coeffA, block_masked(jA2));
        dma_wait();
        __syncthreads();
        const int k0 = kb1;
        kb1 = kb2;
        kb2 = kb3;
        kb3 = k0;
    }

    // ---- epilogue (kernels.py:105-108)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const int qrow = q0 + 32 * qb + i;
        const float l = half_swap_sum(lsum[qb]);
        const float inv = 1.0f / l;
        if (qrow < N) {
            char *op = a.O + (int64_t)b * a.os[0] + (int64_t)hh * a.os[1] + (int64_t)qrow * a.os[2] + h * 8;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef __attribute__((ext_vector_type(4))) T Tx4;
                    Tx4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (T)(o[qb][db][4 * g + j] * inv);
                    *(u32x2 *)(op + db * 64 + g * 16) = __builtin_bit_cast(u32x2, v);
                }
            if (h == 0) {
                T *lp = (T *)a.L + b * a.ls[0] + hh * a.ls[1] + qrow;
                *lp = (T)(m[qb] + __builtin_amdgcn_logf(l));
            }
        }
    }
}

template <typename T, int ABL> int launch_t(const Fa2Problem &p, const XArgs &a) {
    const long long nblk = (long long)((p.N + 255) / 256) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma16x: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(256);
    constexpr size_t smem = 5 * 64 * 256;  // three K units + two V tiles
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)fa2_fwd_mfma16x_kernel<T, true, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void *)fa2_fwd_mfma16x_kernel<T, false, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma16x_kernel<T, true, ABL>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma16x_kernel<T, false, ABL>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma16x kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

}  // namespace

int fa2_launch_mfma16x(const Fa2Problem &p, int abl) {
    const bool fits32 = (int64_t)(p.N + 512) * p.ks[2] * 2 < (1LL << 31) && (int64_t)(p.N + 512) * p.vs[2] * 2 < (1LL << 31);
    if (!fa2_mfma16_supports(p) || p.d != 128 || !fits32) {
        fa2_set_error("mfma16x kernel: needs f16/bf16, d = 128, unit d-stride, 16-byte aligned rows, scale > 0, "
                      "N * row stride < 2 GiB");
        return FA2_ERR_UNSUPPORTED;
    }
    XArgs a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V;
    a.O = (char *)p.O; a.L = (char *)p.L;
    for (int k = 0; k < 3; ++k) {
        a.qs[k] = p.qs[k] * 2; a.ks[k] = p.ks[k] * 2; a.vs[k] = p.vs[k] * 2; a.os[k] = p.os[k] * 2;
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    a.group = 1;
    if (p.causal && ((p.B * p.H) & 7) == 0) {
        const int per_xcd = p.B * p.H / 8;
        int g = fa2_env_int("FA2_CAUSAL_GROUP", 2);
        g = g < 1 ? 1 : (g > per_xcd ? per_xcd : g);
        while (per_xcd % g) --g;
        a.group = g;
    }
#ifdef FA2_ABLATIONS
    if (p.dtype == FA2_DTYPE_BF16) switch (abl) {
        case 1: return launch_t<__bf16, 1>(p, a);
        case 3: return launch_t<__bf16, 3>(p, a);
        case 4: return launch_t<__bf16, 4>(p, a);
        case 7: return launch_t<__bf16, 7>(p, a);
        case 8: return launch_t<__bf16, 8>(p, a);
        case 15: return launch_t<__bf16, 15>(p, a);
        case 16: return launch_t<__bf16, 16>(p, a);
        case 32: return launch_t<__bf16, 32>(p, a);
        case 48: return launch_t<__bf16, 48>(p, a);
        default: break;
        }
#endif
    if (abl != 0) {
        fa2_set_error("mfma16x: ablation %d not built", abl);
        return FA2_ERR_UNSUPPORTED;
    }
    return p.dtype == FA2_DTYPE_BF16 ? launch_t<__bf16, 0>(p, a) : launch_t<_Float16, 0>(p, a);
}
