// fa2_mfma16k.hip -- FA-2 forward for f16 / bf16, d in {64, 128}, for SMALL GRIDS (variant "mfma16k"): when B*H*ceil(N/128)
// workgroups do not fill the 256 CUs, a 128-row tile walked by four waves is latency-bound -- every workgroup runs its
// N/64 key tiles one after the other at one wave per SIMD (BASELINE.json configs[1], B2 H8 N1024 d64: 17.8 us).  Here
// the workgroup has EIGHT waves for the same 128 rows: waves w and w + 4 own the same 32 query rows and split the KEYS,
// group g = w >> 2 taking the 64-key tiles t = g (mod 2).  Both groups run the plain tile loop of fa2_mfma16.hip (same
// arithmetic: src/flash_attention_kernels.py:84-108) on their own double-buffered K/V tiles, so the sequential depth
// halves and every SIMD holds two waves; at the end group 1 hands its (O, m, l) to group 0 through LDS and the two
// partial softmaxes are merged exactly as two key tiles are (O = O0 2^(m0-m) + O1 2^(m1-m), same for l).  No global
// workspace, no second launch.  Results differ from the four-wave kernels only by the association of the fp32 sums.
#include "fa2_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
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

struct Mfma16kArgs {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // B, H, N strides in BYTES (d stride is 1 element)
    int64_t ls[2];                       // L strides in elements
    int B, H, N;
    float c_log2e;  // scale * log2(e) > 0
};

// Byte offset of 16-byte chunk `ch` of row `row` inside one [64][D] 16-bit tile.
// D = 128 (256-B rows): chunk ^= ((row&3)<<2 | (row>>2)&3).  A ds_read_b128 lane group reads 16 rows
//   distinct mod 16 at one chunk -> 16 different 16-B slots of the 256-B bank row; a transposed read's
//   half-wave touches rows 4n..4n+3 x one 64-B span -> the (row&3)<<2 term moves each row to its own span.
// D = 64 (128-B rows, two rows per bank row): chunk ^= ((row>>1)&1)<<2 | (row>>2)&3, same argument with
//   row&1 selecting the half of the bank row.
template <int D> __device__ __forceinline__ int lds_off(int row, int ch) {
    if constexpr (D == 128) return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    else return row * 128 + ((ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4);
}

// RB = 32-row blocks of the Q tile, KG = key groups: RB * KG waves.  (4, 2): 128-row tile, eight waves; (2, 2): 64-row tile,
// four waves -- twice the workgroups; (2, 4): 64-row tile, eight waves, a quarter of the sequential depth (d = 64 only:
// KG * 4 tiles of LDS).
template <typename T, int D, int RB, int KG, bool CAUSAL>
__global__ __launch_bounds__(RB * KG * 64, 2) void fa2_fwd_mfma16k_kernel(const Mfma16kArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int NT = RB * KG * 64, BR = RB * 32, BC = 64;
    constexpr int ROWB = D * 2, TILEB = BC * ROWB, CPR = ROWB / 16, CPT = BC * CPR / NT;
    constexpr int RPI = NT / CPR;  // tile rows covered per staging pass
    constexpr int KS = D / 16, DB = D / 32;
    constexpr int GRPB = 4 * TILEB;  // LDS per key group: K0 | K1 | V0 | V1
    extern __shared__ __attribute__((aligned(16))) char smem[];  // group 0 | group 1 | ...
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rb = wave % RB, grp = wave / RB;  // 32-row block of the tile, key group
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;

    // ---- workgroup -> (b, h, Q tile).  Round-robin dispatch puts block ids equal mod 8 on one XCD
    // (speed only): give every XCD whole (b, h) groups so their K/V stay in that XCD's L2.
    const int nq = (N + BR - 1) / BR, nbh = a.B * a.H;
    int bh, qi;
    {
        const int bid = blockIdx.x;
        if ((nbh & 7) == 0) {
            const int slot = bid >> 3;
            bh = (slot / nq) * 8 + (bid & 7);
            qi = slot % nq;
        } else {
            bh = bid / nq;
            qi = bid % nq;
        }
        if (CAUSAL) qi = nq - 1 - qi;  // heaviest tiles first
    }
    const int b = bh / a.H, hh = bh - b * a.H;
    const int q0 = qi * BR + rb * 32;  // first query row of this wave (and of wave ^ 4)

    const char *Qp = a.Q + b * a.qs[0] + hh * a.qs[1];
    const char *Kp = a.K + b * a.ks[0] + hh * a.ks[1];
    const char *Vp = a.V + b * a.vs[0] + hh * a.vs[1];

    // ---- Q fragments: B operand of S^T = K Q^T.  Lane (i, h) holds Q[q0+i][16ks + 8h .. +7].
    frag qf[KS];
    {
        int row = q0 + i;
        row = row < N ? row : N - 1;
        const char *qp = Qp + (int64_t)row * a.qs[2] + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) qf[ks] = __builtin_bit_cast(frag, *(const u32x4 *)(qp + ks * 32));
    }

    const int kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
    const int nt = (kend + BC - 1) / BC;
    const int nstep = (nt + KG - 1) / KG;

    // ---- staging map: thread handles chunk (row = it*RPI + tid/CPR, ch = tid%CPR) of each tile.
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *kg = Kp + (int64_t)st_row * a.ks[2] + st_ch * 16;
    const char *vg = Vp + (int64_t)st_row * a.vs[2] + st_ch * 16;
    const int st_lds = lds_off<D>(st_row, st_ch);  // + it*RPI*ROWB (swizzle depends on row&15 only)
    static_assert(RPI % 16 == 0, "staging pass must cover a multiple of 16 rows");

    // step s stages tiles KG s + g for every group g; all threads load all of them
    u32x4 kreg[KG][CPT], vreg[KG][CPT];
    auto stage_load = [&](int s) {
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int it = 0; it < CPT; ++it) {
                const int t = KG * s + g;
                const int key = t * BC + it * RPI + st_row;
                const bool ok = key < kend;  // beyond the tile's last key (causal) or N: zeros, masked anyway
                const int64_t ro = (int64_t)(t * BC + it * RPI);
                kreg[g][it] = ok ? *(const u32x4 *)(kg + ro * a.ks[2]) : u32x4{0, 0, 0, 0};
                vreg[g][it] = ok ? *(const u32x4 *)(vg + ro * a.vs[2]) : u32x4{0, 0, 0, 0};
            }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int g = 0; g < KG; ++g)
#pragma unroll
            for (int it = 0; it < CPT; ++it) {
                *(LDS_PTR(u32x4))(lds + g * GRPB + buf * TILEB + st_lds + it * RPI * ROWB) = kreg[g][it];
                *(LDS_PTR(u32x4))(lds + g * GRPB + 2 * TILEB + buf * TILEB + st_lds + it * RPI * ROWB) = vreg[g][it];
            }
    };

    // ---- per-lane LDS read offsets.
    // K row read: row kb*32 + i, chunk 2ks + h.
    int k_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_off[ks] = grp * GRPB + lds_off<D>(i, 2 * ks + h);
    // V transposed read (ds_read_b64_tr_b16): within its 16-lane group g, lane 4q+p supplies the address of
    // row (base + 4h + q), columns 32db + 16(g&1) + 4p..+3 and receives column (lane&15) of the 4 rows.
    // u selects keys +0..3 (elements 0-3 of the fragment) or +8..11 (elements 4-7).
    int v_off[2][DB];
    {
        const int w = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                v_off[u][db] = grp * GRPB + 2 * TILEB + lds_off<D>(8 * u + 4 * h + qq, 4 * db + 2 * w + (pp >> 1)) + 8 * (pp & 1);
    }

    f32x16 o[DB];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m = -INFINITY, lsum = 0.0f;
    const float c = a.c_log2e;
    const int qrow = q0 + i;

    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int st = 0; st < nstep; ++st) {
        const int cur = st & 1;
        const bool more = st + 1 < nstep;
        if (more) stage_load(st + 1);

        const int t = KG * st + grp;  // this wave's key tile of the step
        const bool active = t < nt && (!CAUSAL || (t * BC <= q0 + 31));
        if (active) {
            // ---- S^T = K . Q^T : two 32-key blocks.
            f32x16 s[2];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[kb][r] = 0.0f;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 kf = *(LDS_PTR(u32x4))(lds + cur * TILEB + kb * 32 * ROWB + k_off[ks]);
                    s[kb] = M::mfma(__builtin_bit_cast(frag, kf), qf[ks], s[kb]);
                }
            }
            // ---- masks (diagonal tiles of the causal case, and the tail tile when N % 64 != 0).
            const bool need_mask = (CAUSAL && (t * BC + BC - 1 > q0)) || (t * BC + BC > N);
            if (need_mask) {
                int lim = N - 1;
                if (CAUSAL) lim = qrow < lim ? qrow : lim;
                const int klim = lim - (t * BC + 4 * h);  // key(kb, r) = t*64 + 4h + kb*32 + (r&3) + 8*(r>>2)
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (kb * 32 + (r & 3) + 8 * (r >> 2) > klim) s[kb][r] = -INFINITY;
            }
            // ---- online softmax, one query per lane (kernels.py:93-97).
            float mx = s[0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx * c);
            const float coeff = __builtin_amdgcn_exp2f(m - m_new);
            m = m_new;
            float rs = 0.0f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[kb][r], c, -m_new));
                    s[kb][r] = p;
                    rs += p;
                }
            lsum = lsum * coeff + rs;
            // O *= coeff only when some row's max moved (multiplying by 1.0f is exact, so skipping is too).
            if (__any(coeff != 1.0f)) {
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= coeff;
            }
            // ---- O^T += V^T . P^T.  k-step (kb, ss) = keys kb*32 + 16ss .. +15; registers 8ss..8ss+7 of s[kb]
            // are exactly the B fragment (element j <-> key 16ss + 8(j>>2) + 4h + (j&3)).
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    frag pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (T)s[kb][8 * ss + j];  // RTNE (kernels.py:98)
                    const int rowb = cur * TILEB + (kb * 32 + ss * 16) * ROWB;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[0][db]));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[1][db]));
                        const s16x8 vf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        o[db] = M::mfma(__builtin_bit_cast(frag, vf), pf, o[db]);
                    }
                }
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
    }

    // ---- merge of the key groups: waves rb + RB g (g >= 1) hand (O, m, l) to wave rb through LDS ([register][lane]
    // floats; the last barrier of the loop has retired every read of the K/V tiles), combined like key tiles.
    {
        constexpr int NREG = DB * 16 + 2;
        static_assert((KG - 1) * RB * NREG * 256 <= KG * GRPB, "exchange area exceeds the K/V buffers");
        if (grp > 0) {
            LDS_PTR(float) xch = (LDS_PTR(float))lds + ((grp - 1) * RB + rb) * NREG * 64 + lane;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) xch[(db * 16 + r) * 64] = o[db][r];
            xch[(DB * 16) * 64] = m;
            xch[(DB * 16 + 1) * 64] = lsum;
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KG; ++g) {
            LDS_PTR(float) xch = (LDS_PTR(float))lds + ((g - 1) * RB + rb) * NREG * 64 + lane;
            const float m1 = xch[(DB * 16) * 64], l1 = xch[(DB * 16 + 1) * 64];
            const float mm = fmaxf(m, m1);  // finite: group 0 always holds tile 0, whose key 0 no row masks
            const float a0 = __builtin_amdgcn_exp2f(m - mm), a1 = __builtin_amdgcn_exp2f(m1 - mm);
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] = o[db][r] * a0 + xch[(db * 16 + r) * 64] * a1;
            lsum = lsum * a0 + l1 * a1;
            m = mm;
        }
    }

    // ---- epilogue: O = O / l, L = m + log2 l (kernels.py:105-108).  Lane (i, h) owns row q0+i,
    // columns 32db + 8g + 4h .. +3 for g = 0..3.
    const float l = lsum + __shfl_xor(lsum, 32, 64);
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

template <typename T, int D, int RB, int KG> int launch_t(const Fa2Problem &p, const Mfma16kArgs &a) {
    constexpr int BR = RB * 32;
    const int nq = (p.N + BR - 1) / BR;
    const long long nblk = (long long)nq * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma16k: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(RB * KG * 64);
    constexpr size_t smem = KG * 4 * 64 * D * 2;
    static_assert(smem <= 160 * 1024, "tile buffers exceed the LDS");
    static Fa2DeviceLatch attr_done;  // > 64 KiB of dynamic LDS needs the opt-in, once per instantiation and device
    if (attr_done.need()) {
        (void)hipFuncSetAttribute((const void *)fa2_fwd_mfma16k_kernel<T, D, RB, KG, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        (void)hipFuncSetAttribute((const void *)fa2_fwd_mfma16k_kernel<T, D, RB, KG, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_done.mark();
    }
    if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma16k_kernel<T, D, RB, KG, true>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma16k_kernel<T, D, RB, KG, false>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma16k kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

template <typename T> int launch_d(const Fa2Problem &p, const Mfma16kArgs &a, int shape) {
    if (shape == 24) {  // 64-row tile, four key groups: the LDS holds it at d = 64 only
        if (p.d != 64) {
            fa2_set_error("mfma16k (2 row blocks x 4 key groups) needs d = 64");
            return FA2_ERR_UNSUPPORTED;
        }
        return launch_t<T, 64, 2, 4>(p, a);
    }
    if (shape == 22) return p.d == 128 ? launch_t<T, 128, 2, 2>(p, a) : launch_t<T, 64, 2, 2>(p, a);
    return p.d == 128 ? launch_t<T, 128, 4, 2>(p, a) : launch_t<T, 64, 4, 2>(p, a);
}

}  // namespace

int fa2_launch_mfma16k(const Fa2Problem &p, int shape) {  // shape = 10 * row blocks + key groups: 42, 22, 24
    if (!fa2_mfma16_supports(p)) {
        fa2_set_error("mfma16k kernel: needs f16/bf16, d in {64,128}, unit d-stride, 16-byte aligned rows, scale > 0");
        return FA2_ERR_UNSUPPORTED;
    }
    Mfma16kArgs a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V;
    a.O = (char *)p.O; a.L = (char *)p.L;
    for (int k = 0; k < 3; ++k) {
        a.qs[k] = p.qs[k] * 2; a.ks[k] = p.ks[k] * 2; a.vs[k] = p.vs[k] * 2; a.os[k] = p.os[k] * 2;
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    return p.dtype == FA2_DTYPE_BF16 ? launch_d<__bf16>(p, a, shape) : launch_d<_Float16>(p, a, shape);
}
