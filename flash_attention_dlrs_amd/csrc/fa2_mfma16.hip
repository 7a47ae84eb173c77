// fa2_mfma16.hip -- FA-2 forward for f16 / bf16 on gfx950 matrix cores (the north-star kernel).
//
// Replaces the reference's Triton fwd_kernel (src/flash_attention_kernels.py:17-109) for 16-bit
// inputs with d in {64, 128}.  Same arithmetic (fp32 S, m, l, O; exp2-domain online softmax; P
// rounded RTNE to the input dtype before P@V; O/l and L = m + log2 l rounded on store,
// kernels.py:84-108), organised for CDNA4:
//
//  * one workgroup = NW waves (4 or 8) x 32 query rows; a 64-key K/V tile is streamed through LDS,
//    double-buffered, loaded HBM -> registers -> LDS with the loads of tile t+1 issued before the
//    MFMAs of tile t and written after them (one barrier per tile).
//  * "swapped" products so that a query row lives on ONE lane:
//        S^T[key][query] = K_tile . Q^T      v_mfma_f32_32x32x16  A = K rows (ds_read_b128), B = Q (registers)
//        O^T[d][query]  += V_tile^T . P^T    A = V^T (ds_read_b64_tr_b16), B = P^T
//    The accumulator of S^T (query on the lane, 16 keys in registers) IS the B operand of the second
//    product after a pairwise cvt to 16 bit, so P never touches LDS and the softmax row max / row
//    sum are in-lane loops plus one exchange with lane^32.
//  * LDS image of a tile: plain rows, 16-byte chunks XOR-swizzled so that both the row reads of K
//    and the transposed reads of V are bank-conflict free (see lds_off()).
//  * causal: tiles above the diagonal are never loaded, waves whose 32 rows are entirely masked
//    skip the MFMAs of a diagonal tile, only diagonal / tail tiles pay for masking, and heavy Q tiles
//    are launched first; workgroups sharing a (b, h) are placed on one XCD (L2 reuse of K/V).
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

struct Mfma16Args {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // B, H, N strides in BYTES (d stride is 1 element)
    int64_t ls[2];                       // L strides in elements
    int B, H, N;
    float c_log2e;  // scale * log2(e) > 0
    int dbytes;     // row bytes of the head size the caller passed (<= 2 D): the DP kernels zero-fill the rest
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

// DP: the head size is not D itself (any multiple of 8 below it -- SURVEY section 8 row f2): 16-byte chunks at or past
// a.dbytes are zero-filled on their way into the fragments / LDS and never stored, instead of the host padding Q, K, V
// (reference torch.py:38-47).  Zero columns add nothing to Q.K^T and yield the zero columns of O the reference slices away.
template <typename T, int D, int NW, bool CAUSAL, bool DP>
__global__ __launch_bounds__(NW * 64, 2) void fa2_fwd_mfma16_kernel(const Mfma16Args a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    constexpr int NT = NW * 64, BR = NW * 32, BC = 64;
    constexpr int ROWB = D * 2, TILEB = BC * ROWB, CPR = ROWB / 16, CPT = BC * CPR / NT;
    constexpr int RPI = NT / CPR;  // tile rows covered per staging pass
    constexpr int KS = D / 16, DB = D / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const int q0 = qi * BR + wave * 32;  // first query row of this wave

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
        for (int ks = 0; ks < KS; ++ks)
            qf[ks] = __builtin_bit_cast(frag, (!DP || h * 16 + ks * 32 < a.dbytes) ? *(const u32x4 *)(qp + ks * 32) : u32x4{0, 0, 0, 0});
    }

    // ---- staging map: thread handles chunk (row = it*RPI + tid/CPR, ch = tid%CPR) of each tile.
    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *kg = Kp + (int64_t)st_row * a.ks[2] + st_ch * 16;
    const char *vg = Vp + (int64_t)st_row * a.vs[2] + st_ch * 16;
    const int st_lds = lds_off<D>(st_row, st_ch);  // + it*RPI*ROWB (swizzle depends on row&15 only)
    static_assert(RPI % 16 == 0, "staging pass must cover a multiple of 16 rows");

    const int kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
    const int nt = (kend + BC - 1) / BC;

    u32x4 kreg[CPT], vreg[CPT];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int key = t * BC + it * RPI + st_row;
            const bool ok = key < N && (!DP || st_ch * 16 < a.dbytes);
            const int64_t ro = (int64_t)(t * BC + it * RPI);
            kreg[it] = ok ? *(const u32x4 *)(kg + ro * a.ks[2]) : u32x4{0, 0, 0, 0};
            vreg[it] = ok ? *(const u32x4 *)(vg + ro * a.vs[2]) : u32x4{0, 0, 0, 0};
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            *(LDS_PTR(u32x4))(lds + buf * TILEB + st_lds + it * RPI * ROWB) = kreg[it];
            *(LDS_PTR(u32x4))(lds + 2 * TILEB + buf * TILEB + st_lds + it * RPI * ROWB) = vreg[it];
        }
    };

    // ---- per-lane LDS read offsets.
    // K row read: row kb*32 + i, chunk 2ks + h.
    int k_off[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_off[ks] = lds_off<D>(i, 2 * ks + h);
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
                v_off[u][db] = 2 * TILEB + lds_off<D>(8 * u + 4 * h + qq, 4 * db + 2 * w + (pp >> 1)) + 8 * (pp & 1);
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

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < nt;
        if (more) stage_load(t + 1);

        const bool active = !CAUSAL || (t * BC <= q0 + 31);
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
                if (!DP || h * 8 + db * 64 + g * 16 < a.dbytes) *(u32x2 *)(op + db * 64 + g * 16) = __builtin_bit_cast(u32x2, v);
            }
        if (h == 0) {
            T *lp = (T *)a.L + b * a.ls[0] + hh * a.ls[1] + qrow;
            *lp = (T)(m + __builtin_amdgcn_logf(l));
        }
    }
}

template <typename T, int D, int NW> int launch_t(const Fa2Problem &p, const Mfma16Args &a) {
    constexpr int BR = NW * 32;
    const int nq = (p.N + BR - 1) / BR;
    const long long nblk = (long long)nq * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma16: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(NW * 64);
    const size_t smem = 4 * 64 * D * 2;
    if (p.d != D) {
        if (p.causal)
            hipLaunchKernelGGL((fa2_fwd_mfma16_kernel<T, D, NW, true, true>), grid, block, smem, p.stream, a);
        else
            hipLaunchKernelGGL((fa2_fwd_mfma16_kernel<T, D, NW, false, true>), grid, block, smem, p.stream, a);
    } else if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma16_kernel<T, D, NW, true, false>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma16_kernel<T, D, NW, false, false>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma16 kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

template <typename T> int launch_d(const Fa2Problem &p, const Mfma16Args &a, int waves) {
    if (p.d > 64) return waves == 8 ? launch_t<T, 128, 8>(p, a) : launch_t<T, 128, 4>(p, a);
    return waves == 8 ? launch_t<T, 64, 8>(p, a) : launch_t<T, 64, 4>(p, a);
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

// what this file's kernel takes: d = 64 and 128 natively, the other multiples of 8 up to 128 predicated (DP)
bool fa2_mfma16_supports_dp(const Fa2Problem &p) {
    if (p.dtype != FA2_DTYPE_F16 && p.dtype != FA2_DTYPE_BF16) return false;
    if (p.d < 8 || p.d > 128 || (p.d & 7)) return false;
    if (!(p.scale > 0.0f) || !(p.scale < INFINITY)) return false;
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1 || p.os[3] != 1) return false;
    // 16-byte vector loads of Q/K/V rows, 8-byte stores of O: every row start must stay aligned.
    for (int k = 0; k < 3; ++k)
        if ((p.qs[k] & 7) || (p.ks[k] & 7) || (p.vs[k] & 7) || (p.os[k] & 7)) return false;
    if (!aligned16(p.Q) || !aligned16(p.K) || !aligned16(p.V) || !aligned16(p.O)) return false;
    if (p.N > (1 << 24)) return false;
    return true;
}

// the common precondition of the 16-bit MFMA kernels (mfma16d / 16h / 16k / a64 add their own)
bool fa2_mfma16_supports(const Fa2Problem &p) { return (p.d == 64 || p.d == 128) && fa2_mfma16_supports_dp(p); }

int fa2_launch_mfma16(const Fa2Problem &p, int waves) {
    if (!fa2_mfma16_supports_dp(p)) {
        fa2_set_error("mfma16 kernel: needs f16/bf16, d a multiple of 8 up to 128, unit d-stride, 16-byte aligned rows, scale > 0");
        return FA2_ERR_UNSUPPORTED;
    }
    Mfma16Args a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V;
    a.O = (char *)p.O; a.L = (char *)p.L;
    for (int k = 0; k < 3; ++k) {
        a.qs[k] = p.qs[k] * 2; a.ks[k] = p.ks[k] * 2; a.vs[k] = p.vs[k] * 2; a.os[k] = p.os[k] * 2;
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    a.dbytes = p.d * 2;
    return p.dtype == FA2_DTYPE_BF16 ? launch_d<__bf16>(p, a, waves) : launch_d<_Float16>(p, a, waves);
}
