// fa2_mfma16p.hip -- software-pipelined FA-2 forward for f16 / bf16 on gfx950 (variant "mfma16p").
//
// Same arithmetic and data layout ideas as fa2_mfma16.hip (swapped products, one query row per lane,
// P taken from the S^T accumulator without touching LDS; reference: src/flash_attention_kernels.py:84-108).
// What changes is the SCHEDULE.  Profiling fa2_mfma16.hip on MI355X showed the matrix pipe busy 31 %
// of the time: both waves of a SIMD run QK^T, then both run the softmax on the VALU, then both run
// P.V.  Here the key dimension is processed in 32-key BLOCKS and the loop is skewed by one block so
// that every phase of a wave carries independent matrix AND vector work:
//
//     phase 1:  S_next = K_blk(j+1) . Q^T      (8 MFMA)   ||  P_j = exp2(S_j*c - m), row sum, cvt   (VALU)
//     phase 2:  O^T   += V_blk(j)^T . P_j^T    (8 MFMA)   ||  row max of S_next, new m, rescale factor (VALU)
//
// K is staged through LDS in 64-row units offset by half a unit against V (K unit u = keys
// 64u-32 .. 64u+31) so that one loop iteration touches exactly one K unit and one V tile and a
// single barrier per 64 keys still suffices with two buffers each.
//
// LDS rows are PADDED instead of XOR-swizzled (K: +16 B, V: +64 B per row): conflict-free for the
// ds_read_b128 row reads of K ((17*row + chunk) mod 16 is a bijection in row) and for the
// ds_read_b64_tr_b16 transposed reads of V (four consecutive rows land in four different 64-B spans
// of the 256-B bank row), and every read offset is lane_base + immediate -- two address registers.
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

struct PipeArgs {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // B, H, N strides in bytes
    int64_t ls[2];
    int B, H, N;
    float c_log2e;
    int group;  // causal launch order: (b,h) groups interleaved per XCD batch (1 = one group at a time)
};

// Exchange between the two 32-lane halves of the wave with v_permlane32_swap (VALU, no LDS trip):
// r[0][l] = x[l & 31], r[1][l] = x[32 + (l & 31)].  NB: the elements are copied to scalars before the
// bit cast -- __builtin_bit_cast applied directly to a vector ELEMENT (r[1]) reads element 0 with this
// clang (ROCm 7.2), which silently turned the exchange into a no-op.
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

// OPT bit 0: operand prefetch -- the LDS reads of a phase's MFMA operands are issued one phase ahead
// (K fragments of the next QK^T during P.V, V fragments of P.V at the start of QK^T).
template <typename T, int D, int NW, bool CAUSAL, int OPT>
__global__ __launch_bounds__(NW * 64, 2) void fa2_fwd_mfma16p_kernel(const PipeArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int NT = NW * 64, BR = NW * 32;
    constexpr int ROWB = D * 2, CPR = ROWB / 16, CPT = 64 * CPR / NT, RPI = NT / CPR;
    constexpr int KROWB = ROWB + 16, VROWB = ROWB + 64;      // padded LDS rows
    constexpr int KUNIT = 64 * KROWB, VTILE = 64 * VROWB;
    constexpr int VBASE = 2 * KUNIT;                         // LDS: Kunit0 | Kunit1 | Vtile0 | Vtile1
    constexpr int KS = D / 16, DB = D / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;

    const int nq = (N + BR - 1) / BR, nbh = a.B * a.H;
    int bh, qi;
    {
        const int bid = blockIdx.x;
        if ((nbh & 7) == 0) {  // whole (b, h) groups per XCD: K/V reuse in that XCD's L2 (speed only)
            const int slot = bid >> 3, G = a.group;   // G divides nbh/8 (host)
            const int batch = slot / (G * nq), r = slot - batch * (G * nq);
            bh = (batch * G + r % G) * 8 + (bid & 7);
            qi = r / G;   // G groups advance together: tile qi of all of them, then qi+1, ...
        } else {
            bh = bid / nq;
            qi = bid % nq;
        }
        if (CAUSAL) qi = nq - 1 - qi;  // heaviest tiles first
    }
    const int b = bh / a.H, hh = bh - b * a.H;
    const int q0 = qi * BR + wave * 32;
    const int qrow = q0 + i;

    const char *Qp = a.Q + b * a.qs[0] + hh * a.qs[1];
    const char *Kp = a.K + (int64_t)b * a.ks[0] + (int64_t)hh * a.ks[1];
    const char *Vp = a.V + (int64_t)b * a.vs[0] + (int64_t)hh * a.vs[1];

    frag qf[KS];
    {
        const int row = qrow < N ? qrow : N - 1;
        const char *qp = Qp + (int64_t)row * a.qs[2] + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(frag, *(const u32x4 *)(qp + ks * 32));
    }

    // ---- staging: thread owns chunk (row it*RPI + st_row, 16-byte column st_ch) of every unit/tile.
    // Buffer loads with a descriptor that ends at row N: rows past the end (and the "negative" rows of
    // K unit 0) read as zero in hardware -- no exec-masked branches, no 64-bit address registers.
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const int st_k = st_row * KROWB + st_ch * 16, st_v = VBASE + st_row * VROWB + st_ch * 16;
    const int krs = (int)a.ks[2], vrs = (int)a.vs[2];  // row strides in bytes (host guarantees N*stride < 2^31)
    const __amdgpu_buffer_rsrc_t krsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Kp, 0, (N - 1) * krs + ROWB, 0x00020000);
    const __amdgpu_buffer_rsrc_t vrsrc = __builtin_amdgcn_make_buffer_rsrc((void *)Vp, 0, (N - 1) * vrs + ROWB, 0x00020000);
    int kvo[CPT], vvo[CPT];
#pragma unroll
    for (int it = 0; it < CPT; ++it) {
        kvo[it] = (it * RPI + st_row) * krs + st_ch * 16;
        vvo[it] = (it * RPI + st_row) * vrs + st_ch * 16;
    }

    const int kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
    const int nt = (kend + 63) >> 6;      // V tiles (= loop iterations) of this workgroup
    const int nblk = (kend + 31) >> 5;    // 32-key blocks of this workgroup
    int nb = nblk;                        // ... of this wave (causal: up to its diagonal block)
    if (CAUSAL) nb = (q0 >> 5) + 1 < nblk ? (q0 >> 5) + 1 : nblk;

    u32x4 kreg[CPT], vreg[CPT];
    // The tile offset goes into the VGPR offset, never into soffset: the hardware range check covers
    // inst_offset + voffset only, and soffset is unsigned (a negative one would address base + 4 GiB).
    auto load_k = [&](int u) {  // K unit u = keys 64u-32 .. 64u+31; "negative" rows wrap to huge offsets -> zero
        const int base = (u * 64 - 32) * krs;
#pragma unroll
        for (int it = 0; it < CPT; ++it)
            kreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(krsrc, kvo[it] + base, 0, 0));
    };
    auto load_v = [&](int t) {
        const int base = t * 64 * vrs;
#pragma unroll
        for (int it = 0; it < CPT; ++it)
            vreg[it] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, vvo[it] + base, 0, 0));
    };
    auto write_k = [&](int buf) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) *(LDS_PTR(u32x4))(lds + buf * KUNIT + st_k + it * RPI * KROWB) = kreg[it];
    };
    auto write_v = [&](int buf) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) *(LDS_PTR(u32x4))(lds + buf * VTILE + st_v + it * RPI * VROWB) = vreg[it];
    };

    // ---- per-lane read bases (everything else is an immediate).
    const int kbase = i * KROWB + h * 16;  // K row read: row (half*32 + i), bytes 32ks + 16h
    int vbase;                             // V transposed read, see fa2_mfma16.hip
    {
        const int w = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
        vbase = VBASE + (4 * h + qq) * VROWB + (2 * w + (pp >> 1)) * 16 + 8 * (pp & 1);
    }

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m = -INFINITY, lsum = 0.0f;
    const float c = a.c_log2e;
    // Rescale threshold in log2 units: P may reach 2^kThr before the running max is raised.  bf16 P has the
    // fp32 exponent range (24 leaves 2^24 * N far below fp32 overflow in l and O); f16 P must stay below 65504.
    // On N(0,1) inputs at scale 1 (score sigma ~ 16 log2 units) a threshold of 8 still fired ~20 times per wave
    // and 4096 keys -- each time the whole workgroup waits at the next barrier -- 24 makes it rare.
    constexpr float kThr = sizeof(T) == 2 && __is_same(T, _Float16) ? 12.0f : 24.0f;

    // S^T block = K rows [koff ..] . Q^T
    auto qk = [&](f32x16 &s, int koff) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const u32x4 kf = *(LDS_PTR(u32x4))(lds + koff + kbase + ks * 32);
            s = M::mfma(__builtin_bit_cast(frag, kf), qf[ks], s);
        }
    };
    // Row max of block j and the rescale decision.  The running max m is only raised when some row of
    // the wave grew by more than kThr (log2 units): until then P = exp2(S*c - m) may reach 2^kThr instead
    // of 1, which fp32 sums and 16-bit P carry without loss (same relative rounding), and
    // O / l and L = m + log2 l are unchanged in exact arithmetic.  When the branch fires every row moves
    // to its true max and EVERYTHING accumulated at the old max (O and l) is scaled, once (kernels.py:93-97).
    auto partial = [&](f32x16 &s, int j, float &coeff, bool masked) -> bool {
        if (masked) {
            int lim = N - 1;
            if (CAUSAL) lim = qrow < lim ? qrow : lim;
            const int klim = lim - (j * 32 + 4 * h);  // key(r) = 32j + 4h + (r&3) + 8(r>>2)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((r & 3) + 8 * (r >> 2) > klim) s[r] = -INFINITY;
        }
        float mx = fmaxf(s[0], s[1]);
        if constexpr (!(OPT & 8)) {  // OPT bit 3: timing-only ablation, max of two elements only
#pragma unroll
            for (int r = 2; r < 16; ++r) mx = fmaxf(mx, s[r]);
        }
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
    // P = exp2(S*c - m) (kernels.py:94), row sum (:96), P -> 16 bit RTNE (:98)
    auto finish = [&](f32x16 &s, frag (&pf)[2]) {
        float rs = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = __builtin_fmaf(s[r], c, -m);
            if constexpr (!(OPT & 2)) p = __builtin_amdgcn_exp2f(p);   // OPT bit 1: timing-only ablation, no exp
            if constexpr (!(OPT & 4)) rs += p;                          // OPT bit 2: timing-only ablation, no row sum
            pf[r >> 3][r & 7] = (T)p;
        }
        lsum += rs;
    };
    // O *= coeff, l *= coeff (:96-97), in place.  Inline asm keeps the 64 accumulator registers where they
    // are: written as C++ the conditional update makes hipcc keep a second copy of O (64 v_mov per block
    // plus spills).  Rare path, so the MFMA -> VALU and VALU -> MFMA wait states are padded generously.
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
    // O^T += V[rows voff ..]^T . P^T for one 32-key block
    auto pv = [&](frag (&pf)[2], int voff) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int db = 0; db < DB; ++db) {
                const int off = voff + vbase + ss * 16 * VROWB + db * 64;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off + 8 * VROWB));
                const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                o[db] = M::mfma(__builtin_bit_cast(frag, vf), pf[ss], o[db]);
            }
    };
    auto block_masked = [&](int j) { return (CAUSAL && (j * 32 + 31 > q0)) || (j * 32 + 32 > N); };
    // ---- split forms used by the prefetching schedule (OPT & 1)
    auto load_kf = [&](frag (&kf)[KS], int koff) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            kf[ks] = __builtin_bit_cast(frag, *(LDS_PTR(u32x4))(lds + koff + kbase + ks * 32));
    };
    auto qk_regs = [&](f32x16 &s, frag (&kf)[KS]) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s = M::mfma(kf[ks], qf[ks], s);
    };
    auto load_vf = [&](frag (&vf)[DB], int voff, int ss) {
#pragma unroll
        for (int db = 0; db < DB; ++db) {
            const int off = voff + vbase + ss * 16 * VROWB + db * 64;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + off + 8 * VROWB));
            vf[db] = __builtin_bit_cast(frag, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto pv_regs = [&](frag (&vf)[DB], frag pfs) {
#pragma unroll
        for (int db = 0; db < DB; ++db) o[db] = M::mfma(vf[db], pfs, o[db]);
    };

    // ---- prologue: K units 0 and 1, V tile 0; scores + statistics of block 0.
    load_k(0);
    load_v(0);
    write_k(0);
    write_v(0);
    load_k(1);
    write_k(1);
    if constexpr (OPT & 64) {  // registers carry (K unit t+2, V tile t+1) into iteration t
        load_k(2);
        load_v(1);
    }
    __syncthreads();

    f32x16 sA, sB;
    float coeffA = 1.0f, coeffB = 1.0f;
    bool fireA = false, fireB = false;
    frag pf[2];
    qk(sA, 32 * KROWB);  // block 0 = rows 32..63 of K unit 0
    fireA = partial(sA, 0, coeffA, block_masked(0));
    __syncthreads();     // K unit 0 is overwritten by unit 2 in iteration 0: every wave must have read block 0

    // Iterations whose three blocks (2t, 2t+1, 2t+2) all exist for this wave and need no mask run the
    // branch-free steady-state body; the remaining ones (diagonal / tail / nothing left) the guarded body.
    int jm = nb;  // first block of this wave that needs a mask
    if (CAUSAL) jm = (q0 >> 5) < jm ? (q0 >> 5) : jm;
    if ((N >> 5) < jm) jm = N >> 5;
    int t_steady = (jm - 1) / 2;
    t_steady = t_steady < 0 ? 0 : (t_steady > nt ? nt : t_steady);

    int t = 0;
    if constexpr (OPT & 1) {
        frag kf[KS], vf0[DB], vf1[DB];
        if (t_steady > 0) load_kf(kf, KUNIT);  // block 1 = rows 0..31 of K unit 1
        for (; t < t_steady; ++t) {
            load_k(t + 2);
            load_v(t + 1);
            const int kcur = ((t + 1) & 1) * KUNIT;
            const int vcur = (t & 1) * VTILE;
            // ---- half A: S_B = QK(2t+1) || finish(S_A);  O += PV(2t) || partial(S_B)
            rescale(fireA, coeffA);
            load_vf(vf0, vcur, 0);
            load_vf(vf1, vcur, 1);
            qk_regs(sB, kf);
            finish(sA, pf);
            load_kf(kf, kcur + 32 * KROWB);  // K of block 2t+2, needed by half B
            pv_regs(vf0, pf[0]);
            pv_regs(vf1, pf[1]);
            fireB = partial(sB, 2 * t + 1, coeffB, false);
            // ---- half B: S_A = QK(2t+2) || finish(S_B);  O += PV(2t+1) || partial(S_A)
            rescale(fireB, coeffB);
            load_vf(vf0, vcur + 32 * VROWB, 0);
            load_vf(vf1, vcur + 32 * VROWB, 1);
            qk_regs(sA, kf);
            finish(sB, pf);
            pv_regs(vf0, pf[0]);
            pv_regs(vf1, pf[1]);
            fireA = partial(sA, 2 * t + 2, coeffA, false);
            write_k(t & 1);
            write_v((t + 1) & 1);
            __syncthreads();
            // K of block 2t+3 = rows 0..31 of unit t+2, just published; the next iteration's VALU work
            // (finish of S_A) does not depend on it and runs while these reads are in flight.
            if (t + 1 < t_steady) load_kf(kf, (t & 1) * KUNIT);
        }
    } else {
        for (; t < t_steady; ++t) {
            // Default: loads of (K unit t+2, V tile t+1) at the top, LDS writes at the bottom before the barrier.
            // OPT & 64: the registers already hold them (loaded during iteration t-1): write them NOW -- both
            // target buffers were released by the barrier that ended iteration t-1 -- and re-issue the loads of
            // the following unit/tile at once, so nothing but the barrier itself sits at the loop end.
            if constexpr (OPT & 64) {
                write_k(t & 1);
                write_v((t + 1) & 1);
                load_k(t + 3);
                load_v(t + 2);
            } else if constexpr (!(OPT & 32)) {
                load_k(t + 2);  // t_steady <= nt - 1 whenever it is > 0: there is always a next tile here
                load_v(t + 1);
            }
            const int kcur = ((t + 1) & 1) * KUNIT;   // K unit t+1: rows 0..31 = block 2t+1, rows 32..63 = block 2t+2
            const int vcur = (t & 1) * VTILE;         // V tile t:   rows 0..31 = block 2t,   rows 32..63 = block 2t+1
            rescale(fireA, coeffA);
            qk(sB, kcur);
            finish(sA, pf);
            pv(pf, vcur);
            fireB = partial(sB, 2 * t + 1, coeffB, false);
            rescale(fireB, coeffB);
            qk(sA, kcur + 32 * KROWB);
            finish(sB, pf);
            pv(pf, vcur + 32 * VROWB);
            fireA = partial(sA, 2 * t + 2, coeffA, false);
            if constexpr (OPT & (16 | 32)) {  // timing-only ablation: no staging writes, no barrier
#pragma unroll
                for (int it = 0; it < CPT; ++it) asm volatile("" ::"v"(kreg[it]), "v"(vreg[it]));
            } else {
                if constexpr (!(OPT & 64)) {
                    write_k(t & 1);        // K unit t+2 replaces unit t (last read in iteration t-1)
                    write_v((t + 1) & 1);  // V tile t+1 replaces tile t-1
                }
                if constexpr (!(OPT & 128)) __syncthreads();   // OPT 128: timing-only ablation, barrier only
            }
        }
    }
    for (; t < nt; ++t) {
        const bool more = t + 1 < nt;
        if constexpr (OPT & 64) {
            if (more) {
                write_k(t & 1);
                write_v((t + 1) & 1);
                load_k(t + 3);
                load_v(t + 2);
            }
        } else {
            if (more) {
                load_k(t + 2);
                load_v(t + 1);
            }
        }
        const int kcur = ((t + 1) & 1) * KUNIT;
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
        if constexpr (!(OPT & 64)) {
            if (more) {
                write_k(t & 1);
                write_v((t + 1) & 1);
            }
        }
        __syncthreads();
    }

    // ---- epilogue (kernels.py:105-108)
    const float l = half_swap_sum(lsum);
    const float inv = 1.0f / l;
    if (qrow < N) {
        char *op = a.O + b * a.os[0] + hh * a.os[1] + (int64_t)qrow * a.os[2] + h * 8;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef __attribute__((ext_vector_type(4))) T Tx4;
                Tx4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (T)(o[db][4 * g + j] * inv);
                *(u32x2 *)(op + db * 64 + g * 16) = __builtin_bit_cast(u32x2, v);
            }
        if (h == 0) {
            T *lp = (T *)a.L + b * a.ls[0] + hh * a.ls[1] + qrow;
            *lp = (T)(m + __builtin_amdgcn_logf(l));
        }
    }
}

template <typename T, int D, int NW, int OPT> int launch_t(const Fa2Problem &p, const PipeArgs &a) {
    constexpr int BR = NW * 32;
    const long long nblk = (long long)((p.N + BR - 1) / BR) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma16p: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(NW * 64);
    constexpr size_t smem = 2 * 64 * (D * 2 + 16) + 2 * 64 * (D * 2 + 64);
    static bool attr_done = false;  // > 64 KiB of dynamic LDS needs the opt-in (idempotent, racing is harmless)
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void *)fa2_fwd_mfma16p_kernel<T, D, NW, true, OPT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void *)fa2_fwd_mfma16p_kernel<T, D, NW, false, OPT>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done = true;
    }
    if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma16p_kernel<T, D, NW, true, OPT>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma16p_kernel<T, D, NW, false, OPT>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma16p kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

template <typename T, int OPT> int launch_d(const Fa2Problem &p, const PipeArgs &a, int waves) {
    if (p.d == 128) return waves == 8 ? launch_t<T, 128, 8, OPT>(p, a) : launch_t<T, 128, 4, OPT>(p, a);
    return waves == 8 ? launch_t<T, 64, 8, OPT>(p, a) : launch_t<T, 64, 4, OPT>(p, a);
}
template <typename T> int launch_o(const Fa2Problem &p, const PipeArgs &a, int waves, int opt) {
    switch (opt) {
    case 0: return launch_d<T, 0>(p, a, waves);
    case 1: return launch_d<T, 1>(p, a, waves);
    case 64: return launch_d<T, 64>(p, a, waves);
#ifdef FA2_ABLATIONS  // timing-only builds (wrong results by construction); never compiled into the shipped library
    case 2: return launch_d<T, 2>(p, a, waves);
    case 4: return launch_d<T, 4>(p, a, waves);
    case 8: return launch_d<T, 8>(p, a, waves);
    case 14: return launch_d<T, 14>(p, a, waves);
    case 16: return launch_d<T, 16>(p, a, waves);
    case 32: return launch_d<T, 32>(p, a, waves);
    case 46: return launch_d<T, 46>(p, a, waves);
    case 192: return launch_d<T, 192>(p, a, waves);
#endif
    default: fa2_set_error("mfma16p: schedule option %d not built", opt); return FA2_ERR_UNSUPPORTED;
    }
}

}  // namespace

int fa2_launch_mfma16p(const Fa2Problem &p, int waves, int opt) {
    const bool fits32 = (int64_t)(p.N + 512) * p.ks[2] * 2 < (1LL << 31) && (int64_t)(p.N + 512) * p.vs[2] * 2 < (1LL << 31);
    if (!fa2_mfma16_supports(p) || !fits32) {
        fa2_set_error("mfma16p kernel: needs f16/bf16, d in {64,128}, unit d-stride, 16-byte aligned rows, scale > 0, "
                      "N * row stride < 2 GiB");
        return FA2_ERR_UNSUPPORTED;
    }
    PipeArgs a;
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
        int g = fa2_env_int("FA2_CAUSAL_GROUP", 2);  // measured on c3: 2 groups/XCD batch +9 % over 1, 4-16 lower
        g = g < 1 ? 1 : (g > per_xcd ? per_xcd : g);
        while (per_xcd % g) --g;
        a.group = g;
    }
    return p.dtype == FA2_DTYPE_BF16 ? launch_o<__bf16>(p, a, waves, opt) : launch_o<_Float16>(p, a, waves, opt);
}
