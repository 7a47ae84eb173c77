// fa2_mfma8.hip -- FA-2 forward for OCP fp8 (e4m3fn and e5m2), d = 128, on the gfx950 fp8 matrix path
// (variant "mfma8"; BASELINE.json config c5).  The reference maps torch.float8_e5m2 only
// (src/flash_attention_torch.py:14-15); e4m3fn is an extension.  Same arithmetic as every other kernel here
// (src/flash_attention_kernels.py:84-108): fp32 S, m, l, O; P rounded RTNE to the I/O dtype -- fp8 -- before
// P.V (:98); O / l and L rounded to fp8 on store (:107-108).
//
// Structure = fa2_mfma16d.hip (software-pipelined 32-key blocks, LDS-DMA staging, causal tile pairs) with
// one-byte elements:
//   * v_mfma_f32_32x32x16_{fp8_fp8,bf8_bf8}: 8 elements = 8 bytes per lane per operand.  The k index is only a
//     summation index, so one ds_read_b128 of a K row (bytes 32s+16h ..+15) feeds TWO k-steps (2s: low 8 bytes,
//     2s+1: high 8 bytes) and Q is held with the same mapping.
//   * V^T fragments come from ds_read_b64_tr_b8 -- ONE read per MFMA.  Its lane map was measured on the device
//     (scripts/probes/tr8_probe.hip, profiles/r01/ds_read_b64_tr_b8_lane_map.txt): in a 16-lane group, lane i < 8
//     receives byte i of the 8-byte segments addressed by lanes 0,2,..,14 and lane i >= 8 byte i-8 of those
//     addressed by lanes 1,3,..,15.  Source lane 2e+p therefore points at V[key_e][d0 + 8p ..+7]: the group
//     transposes an 8-key x 16-column block and lane i ends up with column d0+i over key_0..key_7.
//   * LDS images: 128-byte rows (two per 256-byte bank row).  K: 16-byte chunks ^ g(row) (the bf16 d=64 swizzle);
//     V: 32-byte units ^ f(row), f = bit1 | bit3<<1 of the row -- the eight keys of a transposed read (rows
//     {0..3, 8..11} + 4h of a 16-key step) then occupy eight different 32-byte spans.  Both through the DMA's
//     source address.
//   * the deferred running-max threshold is 6 (P <= 64 before the max is raised): fp8 has no headroom above 448.
#include "fa2_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(T) __attribute__((address_space(3))) T *

// E4M3 = true: OCP e4m3fn (v_mfma ... fp8_fp8, v_cvt_pk_fp8_f32); false: e5m2 (bf8).
template <bool E4M3> struct F8 {
    static __device__ __forceinline__ f32x16 mfma(long a, long b, f32x16 c) {
        if constexpr (E4M3) return __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(a, b, c, 0, 0, 0);
    }
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

template <bool E4M3, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void fa2_fwd_mfma8_kernel(const F8Args a) {
    using M = F8<E4M3>;
    constexpr int D = 128, BR = NW * 32;
    constexpr int ROWB = D;                        // one byte per element
    constexpr int TILEB = 64 * ROWB;               // 8 KiB
    constexpr int RPP = 1024 / ROWB;               // 8 rows per 1-KiB DMA piece
    constexpr int PIECES = TILEB / 1024, PPW = PIECES / NW;
    constexpr int VBASE = 2 * TILEB;               // LDS: Kunit0 | Kunit1 | Vtile0 | Vtile1 (32 KiB)
    constexpr int KP = D / 32, DB = D / 32;        // k-step PAIRS of Q.K^T (one b128 each), 32-row blocks of O^T
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

    long qf[KP][2];  // [pair s][k-step 2s / 2s+1]: Q[row][32s + 16h + 8e .. +7]

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
    auto dma_k = [&](int u, int buf) {  // K unit u = keys 64u-32 .. 64u+31
        const int base = (u * 64 - 32) * krs;
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) dma16(krsrc, lds_base + buf * TILEB + (wave + pp * NW) * 1024, kvo[pp] + base);
    };
    auto dma_v = [&](int t, int buf) {
        const int base = t * 64 * vrs;
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
            dma16(vrsrc, lds_base + VBASE + buf * TILEB + (wave + pp * NW) * 1024, vvo[pp] + base);
    };

    int kend = 0, nt = 0, nblk = 0, nb = 0;

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
    float m = -INFINITY, lsum = 0.0f;
    const float c = a.c_log2e;
    constexpr float kThr = 6.0f;  // P <= 64 before the running max is raised (e4m3 tops out at 448)

    auto qk = [&](f32x16 &s, int koff) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int sp = 0; sp < KP; ++sp) {
            const u32x4 kf = *(LDS_PTR(u32x4))(lds + koff + k_off[sp]);
            const long lo = (long)(((unsigned long)kf[1] << 32) | kf[0]), hi = (long)(((unsigned long)kf[3] << 32) | kf[2]);
            s = M::mfma(lo, qf[sp][0], s);
            s = M::mfma(hi, qf[sp][1], s);
        }
    };
    auto partial = [&](f32x16 &s, int j, float &coeff, bool masked) -> bool {
        if (masked) {
            int lim = N - 1;
            if (CAUSAL) lim = qrow < lim ? qrow : lim;
            const int klim = lim - (j * 32 + 4 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) > klim) s[r] = -INFINITY;
        }
        float mx = fmaxf(s[0], s[1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) mx = fmaxf(mx, s[r]);
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
    // P = exp2(S*c - m), row sum of the unrounded P, P -> fp8 RTNE (kernels.py:94-98); pf[ss] = keys 16ss .. 16ss+15
    auto finish = [&](f32x16 &s, long (&pf)[2]) {
        float rs = 0.0f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -m));
            rs += p[r];
        }
        lsum += rs;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            int w0 = M::template cvt_pk<false>(p[8 * ss + 0], p[8 * ss + 1], 0);
            w0 = M::template cvt_pk<true>(p[8 * ss + 2], p[8 * ss + 3], w0);
            int w1 = M::template cvt_pk<false>(p[8 * ss + 4], p[8 * ss + 5], 0);
            w1 = M::template cvt_pk<true>(p[8 * ss + 6], p[8 * ss + 7], w1);
            pf[ss] = (long)(((unsigned long)(unsigned)w1 << 32) | (unsigned)w0);
        }
    };
    auto rescale = [&](bool fire, float coeff) {
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
            lsum *= coeff;
        }
    };
    auto pv = [&](long (&pf)[2], int voff) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const i32x2 vf = __builtin_amdgcn_ds_read_tr8_b64_v2i32((LDS_PTR(i32x2))(lds + voff + ss * 16 * ROWB + v_off[db]));
                const long va = (long)(((unsigned long)(unsigned)vf[1] << 32) | (unsigned)vf[0]);
                o[db] = M::mfma(va, pf[ss], o[db]);
            }
    };
    auto block_masked = [&](int j) { return (CAUSAL && (j * 32 + 31 > q0)) || (j * 32 + 32 > N); };

    for (int pass = 0; pass < npass; ++pass) {
        const int qi = pass == 0 ? qi_first : qi_second;
        q0 = qi * BR + wave * 32;
        qrow = q0 + i;
        {
            const int row = qrow < N ? qrow : N - 1;
            const char *qp = Qp + (int64_t)row * a.qs[2] + h * 16;
#pragma unroll
            for (int sp = 0; sp < KP; ++sp) {
                const u32x4 q4 = *(const u32x4 *)(qp + sp * 32);
                qf[sp][0] = (long)(((unsigned long)q4[1] << 32) | q4[0]);
                qf[sp][1] = (long)(((unsigned long)q4[3] << 32) | q4[2]);
            }
        }
        kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
        nt = (kend + 63) >> 6;
        nblk = (kend + 31) >> 5;
        nb = nblk;
        if (CAUSAL) nb = (q0 >> 5) + 1 < nblk ? (q0 >> 5) + 1 : nblk;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
        m = -INFINITY;
        lsum = 0.0f;

        dma_k(0, 0);
        dma_v(0, 0);
        dma_k(1, 1);
        dma_wait();
        __syncthreads();

        f32x16 sA, sB;
        float coeffA = 1.0f, coeffB = 1.0f;
        bool fireA = false, fireB = false;
        long pf[2];
        qk(sA, 32 * ROWB);  // block 0 = rows 32..63 of K unit 0
        fireA = partial(sA, 0, coeffA, block_masked(0));
        __syncthreads();

        int jm = nb;
        if (CAUSAL) jm = (q0 >> 5) < jm ? (q0 >> 5) : jm;
        if ((N >> 5) < jm) jm = N >> 5;
        int t_steady = (jm - 1) / 2;
        t_steady = t_steady < 0 ? 0 : (t_steady > nt ? nt : t_steady);

        int t = 0;
        for (; t < t_steady; ++t) {
            dma_k(t + 2, t & 1);
            dma_v(t + 1, (t + 1) & 1);
            const int kcur = ((t + 1) & 1) * TILEB;
            const int vcur = (t & 1) * TILEB;
            rescale(fireA, coeffA);
            qk(sB, kcur);
            finish(sA, pf);
            pv(pf, vcur);
            fireB = partial(sB, 2 * t + 1, coeffB, false);
            rescale(fireB, coeffB);
            qk(sA, kcur + 32 * ROWB);
            finish(sB, pf);
            pv(pf, vcur + 32 * ROWB);
            fireA = partial(sA, 2 * t + 2, coeffA, false);
            dma_wait();
            __syncthreads();
        }
        for (; t < nt; ++t) {
            const bool more = t + 1 < nt;
            if (more) {
                dma_k(t + 2, t & 1);
                dma_v(t + 1, (t + 1) & 1);
            }
            const int kcur = ((t + 1) & 1) * TILEB;
            const int vcur = (t & 1) * TILEB;
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
            if (jA2 < nb) qk(sA, kcur + 32 * ROWB);
            if (jB < nb) {
                finish(sB, pf);
                pv(pf, vcur + 32 * ROWB);
            }
            if (jA2 < nb) fireA = partial(sA, jA2, coeffA, block_masked(jA2));
            dma_wait();
            __syncthreads();
        }

        // ---- epilogue: O = O / l and L = m + log2 l, both rounded to fp8 (kernels.py:105-108).  Lane (i, h) owns
        // row q0+i, columns 32db + 8g + 4h .. +3 (four fp8 = one dword); the wave's 32 x 128-byte tile goes through
        // its own 4-KiB slice of the idle K/V buffers and leaves as whole rows (see fa2_mfma16d.hip).
        const float l = half_swap_sum(lsum);
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
}

template <bool E4M3, int NW> int launch_t(const Fa2Problem &p, const F8Args &a) {
    constexpr int BR = NW * 32;
    const int nq = (p.N + BR - 1) / BR;
    const long long nblk = (long long)(p.causal ? (nq + 1) / 2 : nq) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma8: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(NW * 64);
    constexpr size_t smem = 4 * 64 * 128;
    if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma8_kernel<E4M3, NW, true>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma8_kernel<E4M3, NW, false>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma8 kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

}  // namespace

bool fa2_mfma8_supports(const Fa2Problem &p) {
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

int fa2_launch_mfma8(const Fa2Problem &p, int waves) {
    if (!fa2_mfma8_supports(p)) {
        fa2_set_error("mfma8 kernel: needs fp8 (e4m3fn / e5m2), d = 128, unit d-stride, 16-byte aligned rows, scale > 0");
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
    if (p.causal && ((p.B * p.H) & 7) == 0) {
        const int per_xcd = p.B * p.H / 8;
        int g = fa2_env_int("FA2_CAUSAL_GROUP", 2);
        g = g < 1 ? 1 : (g > per_xcd ? per_xcd : g);
        while (per_xcd % g) --g;
        a.group = g;
    }
    const bool e4 = p.dtype == FA2_DTYPE_F8E4M3;
    if (waves == 8) return e4 ? launch_t<true, 8>(p, a) : launch_t<false, 8>(p, a);
    return e4 ? launch_t<true, 4>(p, a) : launch_t<false, 4>(p, a);
}
