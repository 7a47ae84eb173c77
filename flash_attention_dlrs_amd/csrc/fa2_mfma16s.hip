// fa2_mfma16s.hip -- the LDS-DMA pipelined f16 / bf16 kernel of fa2_mfma16d.hip on the OTHER matrix shape,
// v_mfma_f32_16x16x32_{bf16,f16} (variant "mfma16s").  Same arithmetic (src/flash_attention_kernels.py:84-108),
// same 32-key block schedule, same K-unit / V-tile staging; d = 128 only.  Why a second shape: the chip holds a
// higher clock on 16x16x32 than on 32x32x16 at equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7),
// and this kernel is clock/power-limited.
//
// Layout (swapped products, as everywhere here).  A wave owns 32 query rows = two 16-query halves x = 0, 1.
//   S^T tile (T, x), T = 0, 1:  16 keys x 16 queries = K_T (A, from LDS) . Q_x^T (B, registers), k = 32 d-columns
//   per step.  MFMA row m of tile T is loaded with key 8(m >> 2) + 4T + (m & 3) of the 32-key block, so lane
//   (n = lane & 15, g = lane >> 4) ends up with keys 8g + 4T + r (r = 0..3) of query n + 16x: after exp2 and a
//   pairwise cvt its 8 values ARE the B fragment (k = 8g + j, j = 4T + r) of  O^T[d][query] += V^T . P^T,
//   one k = 32 step per block, and the V^T fragment of lane group g is two transposed reads of rows 8g..8g+3 and
//   8g+4..8g+7.  No lane exchange anywhere in the loop: the running max is raised only in the (rare) rescale
//   branch, which is taken when ANY lane's local maximum exceeds m + threshold -- the same decision as a test
//   on the true row maximum -- and row sums stay per lane until the epilogue.
// LDS images: K rows swizzled by fK(row) = (row & 3) | ((row >> 3) & 3) << 2 (= the MFMA row m: conflict-free
// ds_read_b128 of the 16x16x32 A operand), V rows by fV of fa2_mfma16d.hip (conflict-free transposed reads).
#include "fa2_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define LDS_PTR(T) __attribute__((address_space(3))) T *

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
    using frag = bf16x8;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<_Float16> {
    using frag = f16x8;
    static __device__ __forceinline__ f32x4 mfma(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

struct SArgs {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // B, H, N strides in bytes
    int64_t ls[2];
    int B, H, N;
    float c_log2e;
    int group;
    int flags;  // experiment switches (FA2_FLAGS): 1 = static priority for waves 4..7
};

// LDS-DMA piece (see fa2_mfma16d.hip: inline asm on purpose; the vmcnt wait is ours)
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds_base, int voffset) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(voffset), "s"(rsrc)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// 16-byte-chunk swizzles of a 256-byte row (functions of row & 31 / row & 15)
__device__ __forceinline__ int swzK(int row) { return (row & 3) | (((row >> 3) & 3) << 2); }
__device__ __forceinline__ int swzV(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ int offK(int row, int ch) { return row * 256 + ((ch ^ swzK(row)) << 4); }
__device__ __forceinline__ int offV(int row, int ch) { return row * 256 + ((ch ^ swzV(row)) << 4); }

template <typename T, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void fa2_fwd_mfma16s_kernel(const SArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int D = 128;
    constexpr int BR = NW * 32;
    constexpr int ROWB = D * 2, CPR = ROWB / 16;
    constexpr int TILEB = 64 * ROWB;
    constexpr int RPP = 1024 / ROWB;
    constexpr int PIECES = TILEB / 1024, PPW = PIECES / NW;
    constexpr int VBASE = 2 * TILEB;  // LDS: Kunit0 | Kunit1 | Vtile0 | Vtile1
    constexpr int KS = D / 32, DT = D / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int N = a.N;
    // T5 static form (cdna_hip_programming.md): the second-dispatched half loses VALU arbitration on every segment
    if ((a.flags & 1) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);

    // work-unit mapping: identical to fa2_mfma16d.hip (causal tile pairs, whole (b, h) groups per XCD)
    const int nq = (N + BR - 1) / BR, nbh = a.B * a.H;
    const int nunit = CAUSAL ? (nq + 1) / 2 : nq;
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
    int q0 = 0;

    const char *Qp = a.Q + (int64_t)b * a.qs[0] + (int64_t)hh * a.qs[1];
    const char *Kp = a.K + (int64_t)b * a.ks[0] + (int64_t)hh * a.ks[1];
    const char *Vp = a.V + (int64_t)b * a.vs[0] + (int64_t)hh * a.vs[1];

    frag qf[2][KS];

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
        const int row = RPP * (wave + pp * NW) + lane / CPR, slot = lane % CPR;
        kvo[pp] = row * krs + (slot ^ swzK(row)) * 16;
        vvo[pp] = row * vrs + (slot ^ swzV(row)) * 16;
    }
    auto dma_k = [&](int u, int buf) {  // K unit u = keys 64u-32 .. 64u+31 -> LDS K buffer buf
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

    // ---- per-lane swizzled read offsets
    int k_off[2][KS];  // K row read of tile T, k-step ks: row 8(n >> 2) + 4T + (n & 3), chunk 4ks + g
#pragma unroll
    for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) k_off[T2][ks] = offK(8 * (n >> 2) + 4 * T2 + (n & 3), 4 * ks + g);
    int v_off[2][DT];  // V transposed read u (rows 8g + 4u + q), d-tile dt; q = (lane >> 2) & 3, p = lane & 3
    {
        const int qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) v_off[u][dt] = VBASE + offV(8 * g + 4 * u + qq, 2 * dt + (pp >> 1)) + 8 * (pp & 1);
    }

    f32x4 o[DT][2];
    float m[2] = {-INFINITY, -INFINITY}, lsum[2] = {0.0f, 0.0f};
    const float c = a.c_log2e;
    constexpr float kThr = __is_same(T, _Float16) ? 12.0f : 24.0f;  // see fa2_mfma16d.hip

    struct Stile { f32x4 t[2][2]; };  // [T][x]

    auto qk = [&](Stile &s, int koff) {  // koff = buffer base + half * 32 rows
#pragma unroll
        for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
            for (int x = 0; x < 2; ++x) s.t[T2][x] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 kf = *(LDS_PTR(u32x4))(lds + koff + k_off[T2][ks]);
                s.t[T2][0] = M::mfma(__builtin_bit_cast(frag, kf), qf[0][ks], s.t[T2][0]);
                s.t[T2][1] = M::mfma(__builtin_bit_cast(frag, kf), qf[1][ks], s.t[T2][1]);
            }
    };
    // row maximum over the four lanes (n, n + 16, n + 32, n + 48) that share a query: rescale branch only
    auto row_max = [&](float x) {
        x = fmaxf(x, __shfl_xor(x, 16));
        return fmaxf(x, __shfl_xor(x, 32));
    };
    auto partial = [&](Stile &s, int j, float (&coeff)[2], bool masked) -> bool {
        if (masked) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                int lim = N - 1;
                const int qrow = q0 + n + 16 * x;
                if (CAUSAL) lim = qrow < lim ? qrow : lim;
                const int klim = lim - (j * 32 + 8 * g);
#pragma unroll
                for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * T2 + r > klim) s.t[T2][x][r] = -INFINITY;
            }
        }
        float mx[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float v = fmaxf(fmaxf(s.t[0][x][0], s.t[0][x][1]), fmaxf(s.t[0][x][2], s.t[0][x][3]));
            float w = fmaxf(fmaxf(s.t[1][x][0], s.t[1][x][1]), fmaxf(s.t[1][x][2], s.t[1][x][3]));
            mx[x] = fmaxf(v, w) * c;
        }
        // local maxima against the shared running max: "no lane exceeds m + thr" <=> "row max <= m + thr"
        const bool fire = !__all((mx[0] - m[0] <= kThr) && (mx[1] - m[1] <= kThr));
        coeff[0] = coeff[1] = 1.0f;
        if (fire) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const float m_new = fmaxf(m[x], row_max(mx[x]));
                coeff[x] = __builtin_amdgcn_exp2f(m[x] - m_new);
                m[x] = m_new;
            }
        }
        return fire;
    };
    auto finish = [&](Stile &s, frag (&pf)[2]) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float rs = 0.0f;
#pragma unroll
            for (int T2 = 0; T2 < 2; ++T2)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s.t[T2][x][r], c, -m[x]));
                    rs += p;
                    pf[x][4 * T2 + r] = (T)p;
                }
            lsum[x] += rs;
        }
    };
    auto rescale = [&](bool fire, const float (&coeff)[2]) {
        if (fire) {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = o[dt][x][r];
                        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(coeff[x]));
                        o[dt][x][r] = v;
                    }
            asm volatile("s_nop 7" ::: "memory");
            lsum[0] *= coeff[0];
            lsum[1] *= coeff[1];
        }
    };
    auto pv = [&](frag (&pf)[2], int voff) {  // voff = buffer base + half * 32 rows
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + voff + v_off[0][dt]));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + voff + v_off[1][dt]));
            const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            o[dt][0] = M::mfma(__builtin_bit_cast(frag, vf), pf[0], o[dt][0]);
            o[dt][1] = M::mfma(__builtin_bit_cast(frag, vf), pf[1], o[dt][1]);
        }
    };
    auto block_masked = [&](int j) { return (CAUSAL && (j * 32 + 31 > q0)) || (j * 32 + 32 > N); };

    for (int pass = 0; pass < npass; ++pass) {
        const int qi = pass == 0 ? qi_first : qi_second;
        q0 = qi * BR + wave * 32;
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int qrow = q0 + n + 16 * x;
            const int row = qrow < N ? qrow : N - 1;
            const char *qp = Qp + (int64_t)row * a.qs[2] + g * 16;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) qf[x][ks] = __builtin_bit_cast(frag, *(const u32x4 *)(qp + ks * 64));
        }
        kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
        nt = (kend + 63) >> 6;
        nblk = (kend + 31) >> 5;
        nb = nblk;
        if (CAUSAL) nb = (q0 >> 5) + 1 < nblk ? (q0 >> 5) + 1 : nblk;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int x = 0; x < 2; ++x) o[dt][x] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        m[0] = m[1] = -INFINITY;
        lsum[0] = lsum[1] = 0.0f;

        dma_k(0, 0);
        dma_v(0, 0);
        dma_k(1, 1);
        dma_wait();
        __syncthreads();

        Stile sA, sB;
        float coeffA[2] = {1.0f, 1.0f}, coeffB[2] = {1.0f, 1.0f};
        bool fireA = false, fireB = false;
        frag pf[2];
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

        // ---- epilogue (kernels.py:105-108): the wave's 32 x 128 tile leaves through its own LDS slice as whole rows
        float l[2], inv[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            float s = lsum[x];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            l[x] = s;
            inv[x] = 1.0f / s;
        }
        {
            const int ebase = wave * 32 * ROWB;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    typedef __attribute__((ext_vector_type(4))) T Tx4;
                    Tx4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (T)(o[dt][x][r] * inv[x]);
                    // row n + 16x, d = 16dt + 4g + r: chunk 2dt + (g >> 1), byte 8 (g & 1)
                    *(LDS_PTR(u32x2))(lds + ebase + offV(n + 16 * x, 2 * dt + (g >> 1)) + 8 * (g & 1)) = __builtin_bit_cast(u32x2, v);
                }
            constexpr int RPI = 64 / CPR;  // 4 rows per store instruction
            const int er = lane / CPR, ec = lane % CPR;
            char *ob = a.O + (int64_t)b * a.os[0] + (int64_t)hh * a.os[1];
#pragma unroll
            for (int k = 0; k < 32 / RPI; ++k) {
                const int r = k * RPI + er;
                const u32x4 val = *(LDS_PTR(u32x4))(lds + ebase + offV(r, ec));
                if (q0 + r < N) *(u32x4 *)(ob + (int64_t)(q0 + r) * a.os[2] + ec * 16) = val;
            }
        }
        if (g == 0) {
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int qrow = q0 + n + 16 * x;
                if (qrow < N) {
                    T *lp = (T *)a.L + b * a.ls[0] + hh * a.ls[1] + qrow;
                    *lp = (T)(m[x] + __builtin_amdgcn_logf(l[x]));
                }
            }
        }
        if (pass + 1 < npass) __syncthreads();
    }  // pass
}

template <typename T, int NW> int launch_t(const Fa2Problem &p, const SArgs &a) {
    constexpr int BR = NW * 32;
    const int nq = (p.N + BR - 1) / BR;
    const long long nblk = (long long)(p.causal ? (nq + 1) / 2 : nq) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma16s: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(NW * 64);
    constexpr size_t smem = 4 * 64 * 128 * 2;  // 64 KiB
    if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma16s_kernel<T, NW, true>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma16s_kernel<T, NW, false>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma16s kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

}  // namespace

int fa2_launch_mfma16s(const Fa2Problem &p, int waves) {
    const bool fits32 = (int64_t)(p.N + 512) * p.ks[2] * 2 < (1LL << 31) && (int64_t)(p.N + 512) * p.vs[2] * 2 < (1LL << 31);
    if (!fa2_mfma16_supports(p) || !fits32 || p.d != 128) {
        fa2_set_error("mfma16s kernel: needs f16/bf16, d = 128, unit d-stride, 16-byte aligned rows, scale > 0, "
                      "N * row stride < 2 GiB");
        return FA2_ERR_UNSUPPORTED;
    }
    SArgs a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V;
    a.O = (char *)p.O; a.L = (char *)p.L;
    for (int k = 0; k < 3; ++k) {
        a.qs[k] = p.qs[k] * 2; a.ks[k] = p.ks[k] * 2; a.vs[k] = p.vs[k] * 2; a.os[k] = p.os[k] * 2;
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    a.group = 1;
    a.flags = fa2_env_int("FA2_FLAGS", 0);
    if (p.causal && ((p.B * p.H) & 7) == 0) {
        const int per_xcd = p.B * p.H / 8;
        int g = fa2_env_int("FA2_CAUSAL_GROUP", 2);
        g = g < 1 ? 1 : (g > per_xcd ? per_xcd : g);
        while (per_xcd % g) --g;
        a.group = g;
    }
    const int nw = waves == 4 ? 4 : 8;
    if (p.dtype == FA2_DTYPE_BF16) return nw == 8 ? launch_t<__bf16, 8>(p, a) : launch_t<__bf16, 4>(p, a);
    return nw == 8 ? launch_t<_Float16, 8>(p, a) : launch_t<_Float16, 4>(p, a);
}
