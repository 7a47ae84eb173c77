// fa2_bwd_mfma32.hip -- FA-2 backward for float32 on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, a
// k-ordered fma chain at the fp32 vector rate; gfx950 has no reduced-precision fp32 matrix path).
//
// fp32 is the dtype of the reference's own backward checks (src/test_correctness.py:44-76 at (32,32,256,128),
// src/test_torch.py gradcheck at (2,2,32,128)).  Arithmetic and work split are those of fa2_bwd_mfma16.hip
// (src/flash_attention_kernels.py:115-166, :276-317; two owner launches, no cross-workgroup sums, the fp32 row statistic
// handed from the dQ launch to the dK/dV launch); fragments and LDS image are those of fa2_mfma32.hip:
//     first products   X0 = T0 . f0^T, X1 = T1 . f1^T   A = row read of the swept tile (ds_read_b128 = 4 k-steps),
//                                                        B = the owned row held in registers;  k-step 4c+jj, lane
//                                                        half h  <->  d = 8c + 4h + jj
//     second products  acc^T[d][own] += T^T . X          k-step r uses ACCUMULATOR REGISTER r of X as B (its row is
//                                                        (r&3) + 8(r>>2) + 4h), A = T[that row][32db + i] (ds_read_b32)
// MODE 1 (dQ): a wave owns 32 query rows, the workgroup sweeps the keys 32 at a time.  MODE 0 (dK, dV): a wave owns 32
// keys and both gradients (128 accumulators; one wave per SIMD either way at d = 128 -- the owned rows alone are 128
// registers in fp32), the workgroup sweeps the query rows.
#include "fa2_bwd_common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define LDS_PTR(T) __attribute__((address_space(3))) T *

struct B32Args {
    const char *Q, *K, *V, *O, *dO, *L;
    char *dQ, *dK, *dV;
    float *D;
    int64_t qs[3], ks[3], vs[3], os[3], dos[3], dqs[3], dks[3], dvs[3];  // B, H, N strides in BYTES
    int64_t ls[2];                                                        // elements
    int B, H, N, causal;
    float c_log2e, scale;
};

// 16-byte chunk `ch` of row `row` in a [32][D] fp32 tile (fa2_mfma32.hip)
template <int D> __device__ __forceinline__ int lds_off32(int row, int ch) { return row * (D * 4) + ((ch ^ (row & 15)) << 4); }

// D[b, h, n] = sum_x O * dO  (kernels.py:115-166): D/4 lanes per row, 16 bytes of each operand per lane
template <int D> __global__ __launch_bounds__(256) void bwd32_D_kernel(const B32Args a) {
    constexpr int LPR = D / 4;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = gid / LPR, rows = (long long)a.B * a.H * a.N;
    const int ch = (int)(gid % LPR);
    float s = 0.0f;
    if (r < rows) {
        const int n = (int)(r % a.N);
        const long long bh = r / a.N;
        const int h = (int)(bh % a.H), b = (int)(bh / a.H);
        const f32x4 o = *(const f32x4 *)(a.O + b * a.os[0] + h * a.os[1] + (int64_t)n * a.os[2] + ch * 16);
        const f32x4 g = *(const f32x4 *)(a.dO + b * a.dos[0] + h * a.dos[1] + (int64_t)n * a.dos[2] + ch * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) s += o[j] * g[j];
    }
#pragma unroll
    for (int w = 1; w < LPR && w < 64; w <<= 1) s += __shfl_xor(s, w, 64);
    if (r < rows && ch == 0) a.D[r] = s;
}

template <int D, int MODE>
__global__ __launch_bounds__(256, (D == 128 ? 1 : 2)) void bwd_mfma32_kernel(const B32Args a) {
    constexpr int NT = 256, BO = 128, BS = 32;  // owned rows per workgroup, swept rows per tile
    constexpr int ROWB = D * 4, TILEB = BS * ROWB, CPR = ROWB / 16, CPT = BS * CPR / NT, RPI = NT / CPR;
    constexpr int NC = D / 8, DB = D / 32;
    constexpr int LOFF = 4 * TILEB;  // LDS: T0[2] | T1[2] | L[2][32] | D[2][32] (MODE 0 only)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;

    const int nblk = (N + BO - 1) / BO, nbh = a.B * a.H;
    int bh, blk;
    {
        const int bid = blockIdx.x;
        if ((nbh & 7) == 0) {
            const int slot = bid >> 3;
            bh = (slot / nblk) * 8 + (bid & 7);
            blk = slot % nblk;
        } else {
            bh = bid / nblk;
            blk = bid % nblk;
        }
    }
    if (MODE == 1 && a.causal) blk = nblk - 1 - blk;  // heaviest query blocks first
    const int b = bh / a.H, hh = bh - b * a.H;
    const int own0 = blk * BO + wave * 32, orow = own0 + i;

    const char *T0p = (MODE == 1 ? a.K + b * a.ks[0] + hh * a.ks[1] : a.Q + b * a.qs[0] + hh * a.qs[1]);
    const char *T1p = (MODE == 1 ? a.V + b * a.vs[0] + hh * a.vs[1] : a.dO + b * a.dos[0] + hh * a.dos[1]);
    const int64_t t0rs = MODE == 1 ? a.ks[2] : a.qs[2], t1rs = MODE == 1 ? a.vs[2] : a.dos[2];
    const char *F0p = (MODE == 1 ? a.Q + b * a.qs[0] + hh * a.qs[1] : a.K + b * a.ks[0] + hh * a.ks[1]);
    const char *F1p = (MODE == 1 ? a.dO + b * a.dos[0] + hh * a.dos[1] : a.V + b * a.vs[0] + hh * a.vs[1]);
    const int64_t f0rs = MODE == 1 ? a.qs[2] : a.ks[2], f1rs = MODE == 1 ? a.dos[2] : a.vs[2];
    const float *Lp = (const float *)a.L + b * a.ls[0] + hh * a.ls[1];
    const float *Dp = a.D + ((int64_t)b * a.H + hh) * N;
    float *Lc = a.D + (int64_t)a.B * a.H * N + ((int64_t)b * a.H + hh) * N;  // fp32 row statistic, see fa2_bwd_mfma16.hip

    // owned row: lane (i, h) holds F[orow][8c + 4h + jj] in f[c][jj]
    f32x4 f0[NC], f1[NC];
    {
        const int row = orow < N ? orow : N - 1;
        const char *p0 = F0p + (int64_t)row * f0rs + h * 16, *p1 = F1p + (int64_t)row * f1rs + h * 16;
#pragma unroll
        for (int cidx = 0; cidx < NC; ++cidx) {
            f0[cidx] = *(const f32x4 *)(p0 + cidx * 32);
            f1[cidx] = *(const f32x4 *)(p1 + cidx * 32);
        }
    }
    float Lown = 0.0f, Down = 0.0f;
    if (MODE == 1) {
        const int row = orow < N ? orow : N - 1;
        Lown = Lp[row];
        Down = Dp[row];
    }

    const bool is_causal = a.causal != 0;
    const int wg0 = blk * BO;
    int t_begin = 0, t_end = (N + BS - 1) / BS;
    if (is_causal) {
        if (MODE == 1) t_end = (wg0 + BO - 1 < N - 1 ? wg0 + BO - 1 : N - 1) / BS + 1;
        else t_begin = wg0 / BS;
    }

    const int st_row = tid / CPR, st_ch = tid % CPR;
    const char *g0 = T0p + (int64_t)st_row * t0rs + st_ch * 16;
    const char *g1 = T1p + (int64_t)st_row * t1rs + st_ch * 16;
    int st_lds[CPT];
#pragma unroll
    for (int it = 0; it < CPT; ++it) st_lds[it] = lds_off32<D>(it * RPI + st_row, st_ch);
    f32x4 r0[CPT], r1[CPT];
    float lreg = 0.0f, dreg = 0.0f;
    auto stage_load = [&](int t) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            const int row = t * BS + it * RPI + st_row;
            const bool ok = row < N;
            const int64_t ro = (int64_t)(t * BS + it * RPI);
            r0[it] = ok ? *(const f32x4 *)(g0 + ro * t0rs) : f32x4{0, 0, 0, 0};
            r1[it] = ok ? *(const f32x4 *)(g1 + ro * t1rs) : f32x4{0, 0, 0, 0};
        }
        if (MODE == 0 && tid < BS) {
            const int row = t * BS + tid;
            lreg = row < N ? Lc[row] : INFINITY;
            dreg = row < N ? Dp[row] : 0.0f;
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int it = 0; it < CPT; ++it) {
            *(LDS_PTR(f32x4))(lds + buf * TILEB + st_lds[it]) = r0[it];
            *(LDS_PTR(f32x4))(lds + 2 * TILEB + buf * TILEB + st_lds[it]) = r1[it];
        }
        if (MODE == 0 && tid < BS) {
            *(LDS_PTR(float))(lds + LOFF + (buf * BS + tid) * 4) = lreg;
            *(LDS_PTR(float))(lds + LOFF + (2 * BS + buf * BS + tid) * 4) = dreg;
        }
    };

    f32x16 acc0[DB], acc1[MODE == 0 ? DB : 1];
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[db][r] = 0.0f;
    if (MODE == 0) {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[db][r] = 0.0f;
    }
    const float c = a.c_log2e;
    float rsum = 0.0f;

    if (t_begin < t_end) {
        stage_load(t_begin);
        stage_write(t_begin & 1);
    }
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < t_end;
        if (more) stage_load(t + 1);
        const int srow0 = t * BS;
        const bool skip = is_causal && (MODE == 1 ? srow0 > own0 + 31 : srow0 + 31 < own0);
        if (!skip) {
            f32x16 x0, x1;
#pragma unroll
            for (int r = 0; r < 16; ++r) x0[r] = x1[r] = 0.0f;
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const f32x4 tf = *(LDS_PTR(f32x4))(lds + cur * TILEB + lds_off32<D>(i, 2 * cidx + h));
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) x0 = __builtin_amdgcn_mfma_f32_32x32x2f32(tf[jj], f0[cidx][jj], x0, 0, 0, 0);  // :283
            }
#pragma unroll
            for (int cidx = 0; cidx < NC; ++cidx) {
                const f32x4 tf = *(LDS_PTR(f32x4))(lds + 2 * TILEB + cur * TILEB + lds_off32<D>(i, 2 * cidx + h));
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) x1 = __builtin_amdgcn_mfma_f32_32x32x2f32(tf[jj], f1[cidx][jj], x1, 0, 0, 0);  // :289
            }
            const bool need_mask = (srow0 + 32 > N) || (is_causal && (MODE == 1 ? srow0 + 31 > own0 : srow0 < own0 + 31));
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 lv = {Lown, Lown, Lown, Lown}, dv = {Down, Down, Down, Down};
                if (MODE == 0) {
                    lv = *(LDS_PTR(f32x4))(lds + LOFF + (cur * BS + 8 * g + 4 * h) * 4);
                    dv = *(LDS_PTR(f32x4))(lds + LOFF + (2 * BS + cur * BS + 8 * g + 4 * h) * 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    float p = __builtin_amdgcn_exp2f(x0[r] * c - lv[j]);  // :283-285
                    if (need_mask) {
                        const int srow = srow0 + 8 * g + 4 * h + j;
                        const int key = MODE == 1 ? srow : orow, qry = MODE == 1 ? orow : srow;
                        if (srow >= N || (is_causal && key > qry)) p = 0.0f;
                    }
                    if (MODE == 1) rsum += p;
                    x0[r] = p;
                    x1[r] = p * (x1[r] - dv[j]);  // :291 (scale applied once, at the end)
                }
            }
            // second products, k-step r = accumulator register r; operands fetched one group of 4 k-steps ahead
            float tf0[2][4][DB], tf1[2][4][MODE == 0 ? DB : 1];
            auto load_group = [&](int g) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int r = 4 * g + rr;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const int col = 32 * db + i;
                        const int off = lds_off32<D>(row, col >> 2) + (col & 3) * 4;
                        tf0[g & 1][rr][db] = *(LDS_PTR(float))(lds + cur * TILEB + off);
                        if (MODE == 0) tf1[g & 1][rr][db] = *(LDS_PTR(float))(lds + 2 * TILEB + cur * TILEB + off);
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
                    for (int db = 0; db < DB; ++db) {
                        acc0[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(tf0[g & 1][rr][db], x1[4 * g + rr], acc0[db], 0, 0, 0);  // :293 / :317
                        if (MODE == 0)
                            acc1[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(tf1[g & 1][rr][db], x0[4 * g + rr], acc1[db], 0, 0, 0);  // :287
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
    }

    float tot = 1.0f;
    if (MODE == 1) tot = rsum + __shfl_xor(rsum, 32, 64);  // lanes i and i + 32 share a query
    if (orow < N) {
        auto store_rows = [&](char *base, const int64_t *st, f32x16 *acc, float mul) {
            char *op = base + b * st[0] + hh * st[1] + (int64_t)orow * st[2] + h * 16;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[db][4 * g + j] * mul;
                    *(f32x4 *)(op + db * 128 + g * 32) = v;
                }
        };
        if (MODE == 1) {
            store_rows(a.dQ, a.dqs, acc0, a.scale / tot);
            if (h == 0) Lc[orow] = Lown + __builtin_amdgcn_logf(tot);
        } else {
            store_rows(a.dK, a.dks, acc0, a.scale);
            store_rows(a.dV, a.dvs, acc1, 1.0f);
        }
    }
}

template <int D> int launch_d(const Fa2BwdProblem &p, const B32Args &a) {
    const long long rows = (long long)p.B * p.H * p.N, lanes = rows * (D / 4);
    hipLaunchKernelGGL((bwd32_D_kernel<D>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, p.stream, a);
    const long long nblk = (long long)((p.N + 127) / 128) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("backward mfma32: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    constexpr size_t smem0 = 4 * 32 * D * 4 + 4 * 32 * 4, smem1 = 4 * 32 * D * 4;
    static Fa2DeviceLatch attr_set;  // > 64 KiB of dynamic LDS (d = 128, key-owner launch) needs the attribute
    if (attr_set.need()) {
        (void)hipFuncSetAttribute((const void *)bwd_mfma32_kernel<D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem0);
        attr_set.mark();
    }
    hipLaunchKernelGGL((bwd_mfma32_kernel<D, 1>), dim3((unsigned)nblk), dim3(256), smem1, p.stream, a);  // leaves Lc
    hipLaunchKernelGGL((bwd_mfma32_kernel<D, 0>), dim3((unsigned)nblk), dim3(256), smem0, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("backward mfma32 launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool fa2_bwd_mfma32_supports(const Fa2BwdProblem &p) {
    if (p.dtype != FA2_DTYPE_F32) return false;
    if (p.d != 64 && p.d != 128) return false;
    if (!(p.scale > 0.0f) || !(p.scale < INFINITY)) return false;
    const int64_t *all[8] = {p.qs, p.ks, p.vs, p.os, p.dos, p.dqs, p.dks, p.dvs};
    for (int t = 0; t < 8; ++t) {
        if (all[t][3] != 1) return false;
        for (int k = 0; k < 3; ++k)
            if (all[t][k] & 3) return false;  // 16-byte vector accesses of whole rows
    }
    const void *ptrs[8] = {p.Q, p.K, p.V, p.O, p.dO, p.dQ, p.dK, p.dV};
    for (int t = 0; t < 8; ++t)
        if (!aligned16(ptrs[t])) return false;
    if (p.N > (1 << 24)) return false;
    return true;
}

int fa2_bwd_launch_mfma32(const Fa2BwdProblem &p) {
    if (!fa2_bwd_mfma32_supports(p)) {
        fa2_set_error("backward mfma32: needs f32, d in {64,128}, unit d-stride, 16-byte aligned rows, scale > 0");
        return FA2_ERR_UNSUPPORTED;
    }
    B32Args a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V; a.O = (const char *)p.O;
    a.dO = (const char *)p.dO; a.L = (const char *)p.L;
    a.dQ = (char *)p.dQ; a.dK = (char *)p.dK; a.dV = (char *)p.dV; a.D = (float *)p.D;
    for (int k = 0; k < 3; ++k) {
        a.qs[k] = p.qs[k] * 4; a.ks[k] = p.ks[k] * 4; a.vs[k] = p.vs[k] * 4; a.os[k] = p.os[k] * 4;
        a.dos[k] = p.dos[k] * 4; a.dqs[k] = p.dqs[k] * 4; a.dks[k] = p.dks[k] * 4; a.dvs[k] = p.dvs[k] * 4;
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N; a.causal = p.causal;
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    a.scale = p.scale;
    return p.d == 128 ? launch_d<128>(p, a) : launch_d<64>(p, a);
}
