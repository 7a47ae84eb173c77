// fa2_mfma16d.hip -- the software-pipelined f16 / bf16 kernel of fa2_mfma16p.hip with LDS-DMA staging
// (variant "mfma16d").  Same arithmetic (src/flash_attention_kernels.py:84-108), same 32-key block schedule
// (QK^T of block j+1 under the softmax of block j), same K-unit / V-tile scheme; what changes is how a tile
// gets from HBM/L2 into LDS: `buffer_load_dwordx4 ... lds` writes 1 KiB per wave-instruction straight into
// LDS -- no staging registers, no ds_write_b128 (ablation of fa2_mfma16p.hip: its LDS writes cost ~11 %).
//
// LDS-DMA writes lane-linearly (M0 base + lane * 16), so rows cannot be padded: the tile image is plain
// 256-byte (d = 128) or 128-byte (d = 64) rows with the 16-byte chunks XOR-swizzled THROUGH THE SOURCE
// ADDRESS (the lane that fills (row, slot) fetches global chunk slot ^ f(row)) and the same XOR on every read.
// f is the involution of fa2_mfma16.hip's lds_off(): conflict-free for the ds_read_b128 row reads of K and for
// the ds_read_b64_tr_b16 transposed reads of V.  Rows past N (and the "negative" rows of K unit 0) are
// zero-filled by the buffer descriptor's range check.
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

struct DmaArgs {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // B, H, N strides in bytes
    int64_t ls[2];
    int B, H, N;
    float c_log2e;
    int group;
    int flags;  // experiment switches (FA2_FLAGS): 1 = static priority for waves 4..7
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

// chunk swizzle of a row inside a [64][D] 16-bit tile (function of row & 15 only)
template <int D> __device__ __forceinline__ int swz(int row) {
    if constexpr (D == 128) return ((row & 3) << 2) | ((row >> 2) & 3);
    else return (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
}
template <int D> __device__ __forceinline__ int lds_off(int row, int ch) { return row * (D * 2) + ((ch ^ swz<D>(row)) << 4); }

template <typename T, int D, int NW, bool CAUSAL>
__global__ __launch_bounds__(NW * 64, 2) void fa2_fwd_mfma16d_kernel(const DmaArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int BR = NW * 32;
    constexpr int ROWB = D * 2, CPR = ROWB / 16;   // bytes per row, 16-byte chunks per row
    constexpr int TILEB = 64 * ROWB;               // K unit = V tile = 64 rows
    constexpr int RPP = 1024 / ROWB;               // rows per 1-KiB DMA piece (4 or 8)
    constexpr int PIECES = TILEB / 1024, PPW = PIECES / NW;  // pieces per tile, per wave
    constexpr int VBASE = 2 * TILEB;               // LDS: Kunit0 | Kunit1 | Vtile0 | Vtile1
    constexpr int KS = D / 16, DB = D / 32;
    static_assert(PPW >= 1, "too many waves for this tile");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (M0, scalar branches)
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;
    // T5 static form (cdna_hip_programming.md): the second-dispatched half loses VALU arbitration on every segment
    if ((a.flags & 1) && wave >= NW / 2) __builtin_amdgcn_s_setprio(1);

    // Causal: a workgroup owns the PAIR of Q tiles (nq-1-p, p) of one (b, h) and runs them back to back, heavy one
    // first -- every workgroup then does the same amount of work (nq+1 tile-steps), so the 256 CUs finish together
    // instead of trailing off through the light tiles; the pair shares its K/V through L2.  Non-causal: one tile.
    const int nq = (N + BR - 1) / BR, nbh = a.B * a.H;
    const int nunit = CAUSAL ? (nq + 1) / 2 : nq;  // work units (tile pairs / tiles) per (b, h)
    int bh, unit;
    {
        const int bid = blockIdx.x;
        if ((nbh & 7) == 0) {  // whole (b, h) groups per XCD (speed only)
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
    int q0 = 0, qrow = 0;  // set per pass

    const char *Qp = a.Q + (int64_t)b * a.qs[0] + (int64_t)hh * a.qs[1];
    const char *Kp = a.K + (int64_t)b * a.ks[0] + (int64_t)hh * a.ks[1];
    const char *Vp = a.V + (int64_t)b * a.vs[0] + (int64_t)hh * a.vs[1];

    frag qf[KS];

    // ---- DMA staging.  Piece p of a tile = rows RPP*p .. RPP*p+RPP-1 = 1 KiB of LDS; wave w issues pieces
    // w, w+NW, ...  Lane l fills LDS (row = RPP*p + l / CPR, slot = l % CPR) with global chunk slot ^ f(row).
    const int krs = (int)a.ks[2], vrs = (int)a.vs[2];
    // descriptors built from wave-uniform scalars: {base lo, base hi, bytes, flags}; raw buffer (stride 0)
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
        const int chunk = slot ^ swz<D>(row);
        kvo[pp] = row * krs + chunk * 16;
        vvo[pp] = row * vrs + chunk * 16;
    }
    auto dma_k = [&](int u, int buf) {  // K unit u = keys 64u-32 .. 64u+31 -> LDS K buffer buf
        const int base = (u * 64 - 32) * krs;  // in the VGPR offset: range-checked ("negative" rows wrap -> zero)
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) dma16(krsrc, lds_base + buf * TILEB + (wave + pp * NW) * 1024, kvo[pp] + base);
    };
    auto dma_v = [&](int t, int buf) {
        const int base = t * 64 * vrs;
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp)
            dma16(vrsrc, lds_base + VBASE + buf * TILEB + (wave + pp * NW) * 1024, vvo[pp] + base);
    };

    int kend = 0, nt = 0, nblk = 0, nb = 0;  // set per pass

    // ---- per-lane swizzled read offsets
    int k_off[KS];  // K row read: row (half*32 + i), chunk 2ks + h
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_off[ks] = lds_off<D>(i, 2 * ks + h);
    int v_off[2][DB];  // V transposed read (see fa2_mfma16.hip): u = keys +0..3 / +8..11 of the 16-key step
    {
        const int w = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                v_off[u][db] = VBASE + lds_off<D>(8 * u + 4 * h + qq, 4 * db + 2 * w + (pp >> 1)) + 8 * (pp & 1);
    }

    f32x16 o[DB];
    float m = -INFINITY, lsum = 0.0f;
    const float c = a.c_log2e;
    // Rescale threshold in log2 units: P may reach 2^kThr before the running max is raised.  bf16 P has the
    // fp32 exponent range (24 leaves 2^24 * N far below fp32 overflow in l and O); f16 P must stay below 65504.
    // On N(0,1) inputs at scale 1 (score sigma ~ 16 log2 units) a threshold of 8 still fired ~20 times per wave
    // and 4096 keys -- each time the whole workgroup waits at the next barrier -- 24 makes it rare.
    constexpr float kThr = sizeof(T) == 2 && __is_same(T, _Float16) ? 12.0f : 60.0f;  // bf16: 60 (round 2): see fa2_a64.hip

    auto qk = [&](f32x16 &s, int koff) {  // koff = buffer base + half * 32 rows
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 kf = *(LDS_PTR(u32x4))(lds + koff + k_off[ks]);
            s = M::mfma(__builtin_bit_cast(frag, kf), qf[ks], s);
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
        const bool fire = !__all(mx - m <= kThr);  // deferred running max, see fa2_mfma16p.hip
        coeff = 1.0f;
        if (fire) {
            const float m_new = fmaxf(m, mx);
            coeff = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
        }
        return fire;
    };
    auto finish = [&](f32x16 &s, frag (&pf)[2]) {
        float rs = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], c, -m));
            rs += p;
            pf[r >> 3][r & 7] = (T)p;
        }
        lsum += rs;
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
    auto pv = [&](frag (&pf)[2], int voff) {  // voff = buffer base + half * 32 rows
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int rowb = voff + ss * 16 * ROWB;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[0][db]));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[1][db]));
                const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = M::mfma(__builtin_bit_cast(frag, vf), pf[ss], o[db]);
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
            for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(frag, *(const u32x4 *)(qp + ks * 32));
        }
        kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
        nt = (kend + 63) >> 6;    // V tiles (= loop iterations)
        nblk = (kend + 31) >> 5;  // 32-key blocks of this tile
        nb = nblk;                // ... of this wave (causal: up to its diagonal block)
        if (CAUSAL) nb = (q0 >> 5) + 1 < nblk ? (q0 >> 5) + 1 : nblk;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
        m = -INFINITY;
        lsum = 0.0f;

        // ---- prologue: K units 0, 1 and V tile 0
        dma_k(0, 0);
        dma_v(0, 0);
        dma_k(1, 1);
        dma_wait();
        __syncthreads();

        f32x16 sA, sB;
        float coeffA = 1.0f, coeffB = 1.0f;
        bool fireA = false, fireB = false;
        frag pf[2];
        qk(sA, 32 * ROWB);  // block 0 = rows 32..63 of K unit 0
        fireA = partial(sA, 0, coeffA, block_masked(0));
        __syncthreads();    // K unit 0 is overwritten by unit 2 in iteration 0

        int jm = nb;
        if (CAUSAL) jm = (q0 >> 5) < jm ? (q0 >> 5) : jm;
        if ((N >> 5) < jm) jm = N >> 5;
        int t_steady = (jm - 1) / 2;
        t_steady = t_steady < 0 ? 0 : (t_steady > nt ? nt : t_steady);

        int t = 0;
        for (; t < t_steady; ++t) {
            // both target buffers were released by the barrier that ended iteration t-1; the DMA has the whole
            // iteration to land and is published by the barrier at its end
            dma_k(t + 2, t & 1);
            dma_v(t + 1, (t + 1) & 1);
            const int kcur = ((t + 1) & 1) * TILEB;  // K unit t+1: rows 0..31 = block 2t+1, rows 32..63 = block 2t+2
            const int vcur = (t & 1) * TILEB;        // V tile t:   rows 0..31 = block 2t,   rows 32..63 = block 2t+1
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
            dma_wait();  // this wave's pieces of (K unit t+2, V tile t+1) have landed; the barrier publishes them
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

        // ---- epilogue (kernels.py:105-108).  A lane owns one ROW of O (columns 32db + 8g + 4h ..+3): stored straight
        // from the accumulators that is 16 eight-byte stores per lane, each instruction touching 32 rows.  Instead the
        // wave's 32 x D tile goes through its own 32*ROWB-byte slice of the (now idle) K/V buffers and leaves as
        // whole rows: ROWB/16 lanes x 16 bytes per row, 1 KiB contiguous per store instruction.
        const float l = half_swap_sum(lsum);
        const float inv = 1.0f / l;
        {
            const int ebase = wave * 32 * ROWB;  // NW * 32 * ROWB <= 4 * TILEB
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef __attribute__((ext_vector_type(4))) T Tx4;
                    Tx4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (T)(o[db][4 * g + j] * inv);
                    *(LDS_PTR(u32x2))(lds + ebase + lds_off<D>(i, 4 * db + g) + 8 * h) = __builtin_bit_cast(u32x2, v);
                }
            // same wave wrote and reads: LDS executes a wave's accesses in order; no other wave touches this slice
            // until the barrier below
            constexpr int RPI = 64 / CPR;  // rows per store instruction (4 at d = 128, 8 at d = 64)
            const int er = lane / CPR, ec = lane % CPR;
            char *ob = a.O + (int64_t)b * a.os[0] + (int64_t)hh * a.os[1];
#pragma unroll
            for (int k = 0; k < 32 / RPI; ++k) {
                const int r = k * RPI + er;
                const u32x4 val = *(LDS_PTR(u32x4))(lds + ebase + lds_off<D>(r, ec));
                if (q0 + r < N) *(u32x4 *)(ob + (int64_t)(q0 + r) * a.os[2] + ec * 16) = val;
            }
        }
        if (qrow < N && h == 0) {
            T *lp = (T *)a.L + b * a.ls[0] + hh * a.ls[1] + qrow;
            *lp = (T)(m + __builtin_amdgcn_logf(l));
        }
        if (pass + 1 < npass) __syncthreads();  // the next pass's DMA reuses the slices

    }  // pass
}

template <typename T, int D, int NW> int launch_t(const Fa2Problem &p, const DmaArgs &a) {
    constexpr int BR = NW * 32;
    const int nq = (p.N + BR - 1) / BR;
    const long long nblk = (long long)(p.causal ? (nq + 1) / 2 : nq) * p.B * p.H;  // causal: one workgroup per tile pair
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma16d: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(NW * 64);
    constexpr size_t smem = 4 * 64 * D * 2;  // 64 KiB (d = 128) / 32 KiB (d = 64)
    if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma16d_kernel<T, D, NW, true>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma16d_kernel<T, D, NW, false>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma16d kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

template <typename T> int launch_d(const Fa2Problem &p, const DmaArgs &a, int waves) {
    if (p.d == 128) return waves == 8 ? launch_t<T, 128, 8>(p, a) : launch_t<T, 128, 4>(p, a);
    return waves == 8 ? launch_t<T, 64, 8>(p, a) : launch_t<T, 64, 4>(p, a);
}

}  // namespace

int fa2_launch_mfma16d(const Fa2Problem &p, int waves) {
    const bool fits32 = (int64_t)(p.N + 512) * p.ks[2] * 2 < (1LL << 31) && (int64_t)(p.N + 512) * p.vs[2] * 2 < (1LL << 31);
    if (!fa2_mfma16_supports(p) || !fits32) {
        fa2_set_error("mfma16d kernel: needs f16/bf16, d in {64,128}, unit d-stride, 16-byte aligned rows, scale > 0, "
                      "N * row stride < 2 GiB");
        return FA2_ERR_UNSUPPORTED;
    }
    DmaArgs a;
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
    return p.dtype == FA2_DTYPE_BF16 ? launch_d<__bf16>(p, a, waves) : launch_d<_Float16>(p, a, waves);
}
