// fa2_mfma32.hip -- FA-2 forward for float32 on the gfx950 matrix cores.
//
// This is the dtype the reference's own correctness script exercises (src/test_correctness.py:9-14,
// (32,32,256,128) fp32, allclose(atol=1e-4, rtol=1e-5) against SDPA(scale=1)).  The reference asks
// Triton for input_precision="ieee" (src/flash_attention_kernels.py:6,92,98), i.e. no TF32: here both
// contractions run on v_mfma_f32_32x32x2_f32, which is exact fp32 (a k-ordered fma chain) at the
// fp32 vector rate -- gfx950 has no reduced-precision fp32 matrix path, so there is nothing to opt
// out of.
//
// Structure = fa2_mfma16.hip with fp32 fragments: 4 waves x 32 query rows per workgroup, 32-key
// tiles double-buffered in LDS, swapped products so a query row lives on one lane:
//     S^T[key][query] = K . Q^T     A = K[key=i][d]   (ds_read_b128: 4 k-steps per read)
//     O^T[d][query]  += V^T . P^T   A = V[key][d=i]   (ds_read_b32),  B = the S^T accumulator register
// The k index of an MFMA is only a summation index, so each operand pair picks the d (resp. key)
// order that makes its loads wide and conflict-free:
//     QK^T  k-step 4c+jj, lane half h  <->  d   = 8c + 4h + jj
//     PV    k-step r,     lane half h  <->  key = (r&3) + 8(r>>2) + 4h   (= row of accumulator register r)
#include "fa2_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(T) __attribute__((address_space(3))) T *

struct Mfma32Args {
    const char *Q, *K, *V;
    char *O, *L;
    int64_t qs[3], ks[3], vs[3], os[3];  // bytes
    int64_t ls[2];                       // elements
    int B, H, N;
    float c_log2e;
    int dbytes;  // row bytes of the head size the caller passed (<= 4 D): the DP kernels zero-fill the rest
};

// 16-byte chunk `ch` of row `row` in a [32][D] fp32 tile: chunk ^= row & 15 (low four chunk bits).
template <int D> __device__ __forceinline__ int lds_off32(int row, int ch) { return row * (D * 4) + ((ch ^ (row & 15)) << 4); }

// DP: the head size is a multiple of 4 below D (e.g. the reference's d = 40 and d = 8 padding
// cases, torch.py:38-47): 16-byte chunks at or past a.dbytes are zero-filled on load and never stored -- no host padding.
template <int D, bool CAUSAL, bool DP>
// d = 128 needs Q (64) + O (64) + S (16) + staging (32) + fragments > 256 registers: one workgroup per CU there
// (with two, 61 registers spilled and Q was re-read from scratch every tile: 40 % of the fp32 MFMA peak).
__global__ __launch_bounds__(256, (D == 128 ? 1 : 2)) void fa2_fwd_mfma32_kernel(const Mfma32Args a) {
    constexpr int NW = 4, NT = 256, BR = 128, BC = 32;
    constexpr int ROWB = D * 4, TILEB = BC * ROWB, CPR = ROWB / 16, CPT = BC * CPR / NT, RPI = NT / CPR;
    constexpr int NC = D / 8;   // 16-byte Q/K chunks per lane (4 k-steps each)
    constexpr int DB = D / 32;
    (void)NW;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // K0 | K1 | V0 | V1
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;

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
        if (CAUSAL) qi = nq - 1 - qi;
    }
    const int b = bh / a.H, hh = bh - b * a.H;
    const int q0 = qi * BR + wave * 32;

    const char *Qp = a.Q + b * a.qs[0] + hh * a.qs[1];
    const char *Kp = a.K + b * a.ks[0] + hh * a.ks[1];
    const char *Vp = a.V + b * a.vs[0] + hh * a.vs[1];

    // Q: lane (i, h) holds Q[q0+i][8c + 4h + jj] in qf[c][jj].
    f32x4 qf[NC];
    {
        int row = q0 + i;
        row = row < N ? row : N - 1;
        const char *qp = Qp + (int64_t)row * a.qs[2] + h * 16;
#pragma unroll
        for (int cidx = 0; cidx < NC; ++cidx)
            qf[cidx] = (!DP || h * 16 + cidx * 32 < a.dbytes) ? *(const f32x4 *)(qp + cidx * 32) : f32x4{0, 0, 0, 0};
    }

    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *kg = Kp + (int64_t)st_row * a.ks[2] + st_ch * 16;
    const char *vg = Vp + (int64_t)st_row * a.vs[2] + st_ch * 16;
    int st_lds[CPT];  // the swizzle depends on row & 15 and a staging pass covers RPI (8 or 16) rows
#pragma unroll
    for (int it = 0; it < CPT; ++it) st_lds[it] = lds_off32<D>(it * RPI + st_row, st_ch);

    const int kend = CAUSAL ? ((qi * BR + BR) < N ? (qi * BR + BR) : N) : N;
    const int nt = (kend + BC - 1) / BC;

    f32x4 kreg[CPT], vreg[CPT];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int key = t * BC + it * RPI + st_row;
            const bool ok = key < N && (!DP || st_ch * 16 < a.dbytes);
            const int64_t ro = (int64_t)(t * BC + it * RPI);
            kreg[it] = ok ? *(const f32x4 *)(kg + ro * a.ks[2]) : f32x4{0, 0, 0, 0};
            vreg[it] = ok ? *(const f32x4 *)(vg + ro * a.vs[2]) : f32x4{0, 0, 0, 0};
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            *(LDS_PTR(f32x4))(lds + buf * TILEB + st_lds[it]) = kreg[it];
            *(LDS_PTR(f32x4))(lds + 2 * TILEB + buf * TILEB + st_lds[it]) = vreg[it];
        }
    };

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
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const f32x4 kf = *(LDS_PTR(f32x4))(lds + cur * TILEB + lds_off32<D>(i, 2 * cidx + h));
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[jj], qf[cidx][jj], s, 0, 0, 0);
            }
            // kernels.py:92 -- S = dot * log2e (fp32 product, then the subtraction below: two roundings,
            // as in the reference).
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] *= c;
            const bool need_mask = (CAUSAL && (t * BC + BC - 1 > q0)) || (t * BC + BC > N);
            if (need_mask) {
                int lim = N - 1;
                if (CAUSAL) lim = qrow < lim ? qrow : lim;
                const int klim = lim - (t * BC + 4 * h);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) > klim) s[r] = -INFINITY;
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m, mx);                 // :93
            const float coeff = __builtin_amdgcn_exp2f(m - m_new);  // :95 (v_exp_f32: 1 ulp, results below 2^-126 flush)
            m = m_new;
            float rs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(s[r] - m_new);  // :94
                s[r] = p;
                rs += p;
            }
            lsum = lsum * coeff + rs;                         // :96
            if (__any(coeff != 1.0f)) {
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[db][r] *= coeff;  // :97
            }
            // O^T += V^T P^T: k-step r uses accumulator register r as B; A = V[key_r + 4h][32db + i].
            // The V operands are read in groups of 4 k-steps, one group AHEAD of the MFMAs that consume them
            // (left to itself hipcc issues each ds_read_b32 right in front of its MFMA: every MFMA then waits
            // out an LDS round trip and this phase runs at half rate).  sched_barrier pins the order.
            float vf[2][4][DB];
            auto load_group = [&](int g) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * g + rr;
                    const int key = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const int col = 32 * db + i;
                        vf[g & 1][rr][db] = *(LDS_PTR(float))(lds + 2 * TILEB + cur * TILEB + lds_off32<D>(key, col >> 2) + (col & 3) * 4);
                    }
                }
            };
            load_group(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) load_group(g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int db = 0; db < DB; ++db)
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[g & 1][rr][db], s[4 * g + rr], o[db], 0, 0, 0);  // :98
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
    }

    const float l = lsum + __shfl_xor(lsum, 32, 64);
    if (qrow < N) {
        char *op = a.O + b * a.os[0] + hh * a.os[1] + (int64_t)qrow * a.os[2] + h * 16;
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[db][4 * g + j] / l;  // :105
                if (!DP || h * 16 + db * 128 + g * 32 < a.dbytes) *(f32x4 *)(op + db * 128 + g * 32) = v;
            }
        if (h == 0) {
            float *lp = (float *)a.L + b * a.ls[0] + hh * a.ls[1] + qrow;
            *lp = m + log2f(l);  // :106
        }
    }
}

template <int D> int launch_d(const Fa2Problem &p, const Mfma32Args &a) {
    constexpr int BR = 128;
    const long long nblk = (long long)((p.N + BR - 1) / BR) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("mfma32: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    const dim3 grid((unsigned)nblk), block(256);
    const size_t smem = 4 * 32 * D * 4;
    if (p.d != D) {
        if (p.causal)
            hipLaunchKernelGGL((fa2_fwd_mfma32_kernel<D, true, true>), grid, block, smem, p.stream, a);
        else
            hipLaunchKernelGGL((fa2_fwd_mfma32_kernel<D, false, true>), grid, block, smem, p.stream, a);
    } else if (p.causal)
        hipLaunchKernelGGL((fa2_fwd_mfma32_kernel<D, true, false>), grid, block, smem, p.stream, a);
    else
        hipLaunchKernelGGL((fa2_fwd_mfma32_kernel<D, false, false>), grid, block, smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("mfma32 kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

}  // namespace

bool fa2_mfma32_supports(const Fa2Problem &p) {
    if (p.dtype != FA2_DTYPE_F32) return false;
    if (p.d < 4 || p.d > 128 || (p.d & 3)) return false;   // 64 and 128 natively, the other multiples of 4 predicated (DP)
    if (!(p.scale > 0.0f) || !(p.scale < INFINITY)) return false;
    if (p.qs[3] != 1 || p.ks[3] != 1 || p.vs[3] != 1 || p.os[3] != 1) return false;
    for (int k = 0; k < 3; ++k)
        if ((p.qs[k] & 3) || (p.ks[k] & 3) || (p.vs[k] & 3) || (p.os[k] & 3)) return false;
    if (((uintptr_t)p.Q | (uintptr_t)p.K | (uintptr_t)p.V | (uintptr_t)p.O) & 15) return false;
    if (p.N > (1 << 24)) return false;
    return true;
}

int fa2_launch_mfma32(const Fa2Problem &p) {
    if (!fa2_mfma32_supports(p)) {
        fa2_set_error("mfma32 kernel: needs f32, d a multiple of 4 up to 128, unit d-stride, 16-byte aligned rows, scale > 0");
        return FA2_ERR_UNSUPPORTED;
    }
    Mfma32Args a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V;
    a.O = (char *)p.O; a.L = (char *)p.L;
    for (int k = 0; k < 3; ++k) {
        a.qs[k] = p.qs[k] * 4; a.ks[k] = p.ks[k] * 4; a.vs[k] = p.vs[k] * 4; a.os[k] = p.os[k] * 4;
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    a.dbytes = p.d * 4;
    return p.d > 64 ? launch_d<128>(p, a) : launch_d<64>(p, a);
}
