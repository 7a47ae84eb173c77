// fa2_bwd_generic.hip -- catch-all FA-2 backward kernels for gfx950 (contractions on the VALU).
//
// Covers what the reference's backward glue can hand to its kernels (src/flash_attention_torch.py:86-158):
// fp64 / fp32 / fp16 (+ bf16), arbitrary element strides, d = 2^k in [16, 512], and any N >= 1.  It is the
// fallback behind fa2_bwd_mfma16.hip and the path of the reference's own fp32 gradcheck (src/test_torch.py).
//
// Arithmetic = the reference's (src/flash_attention_kernels.py):
//   D  = rowsum(dO * O)                               bwd_D_kernel :115-166  (kept in the accumulate type)
//   S  = Q K^T * log2e ; P = exp2(S - L)              bwd_kernel   :283-285
//   dV += cast(P)^T dO                                             :287
//   dP = dO V^T ; dS = P * (dP - D)                                :289-291
//   dK += cast(dS)^T Q ; dQ += cast(dS) K                          :293, :317
// with P and dS rounded to the I/O dtype (RTNE) before the second contractions as the reference's casts do, and
// every accumulation in fp32 (fp64 for the fp64 dtype) -- the reference accumulates dK/dV/dQ in the I/O dtype
// (out_dtype=..., rounding at every query / key block), which this does not imitate.
//
// Work split (no cross-workgroup sums, hence deterministic; include/fa2_bwd.h):
//   bwd_dkdv: one workgroup per block of TB keys, sweeping the query rows TB at a time  -> dK, dV
//   bwd_dq:   one workgroup per block of TB query rows, sweeping the keys TB at a time  -> dQ
// TB x TB (P, dS) pairs are one thread each; the TB x d outputs are spread over the 256 threads and live in LDS.
#include "fa2_bwd_common.h"
#include "fa2_elem.h"

namespace {

template <typename A> __device__ __forceinline__ A exp2_a(A x);
template <> __device__ __forceinline__ float exp2_a<float>(float x) { return exp2f(x); }
template <> __device__ __forceinline__ double exp2_a<double>(double x) { return exp2(x); }
template <typename A> __device__ __forceinline__ A log2_a(A x);
template <> __device__ __forceinline__ float log2_a<float>(float x) { return log2f(x); }
template <> __device__ __forceinline__ double log2_a<double>(double x) { return log2(x); }

struct GArgs {
    const void *Q, *K, *V, *O, *dO, *L;
    void *dQ, *dK, *dV, *D;
    int64_t qs[4], ks[4], vs[4], os[4], dos[4], dqs[4], dks[4], dvs[4], ls[2];
    int H, N, d, causal, TB;
    double c_log2e, scale;
};

// D[b, h, n] = sum_x O[b, h, n, x] * dO[b, h, n, x]   (kernels.py:115-166)
template <typename E> __global__ __launch_bounds__(256) void bwd_D_kernel(const GArgs a, long long rows) {
    using A = typename E::acc_t;
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const int n = (int)(r % a.N);
    const long long bh = r / a.N;
    const int h = (int)(bh % a.H), b = (int)(bh / a.H);
    const int64_t o0 = b * a.os[0] + h * a.os[1] + n * a.os[2], g0 = b * a.dos[0] + h * a.dos[1] + n * a.dos[2];
    A s = 0;
    for (int x = 0; x < a.d; ++x) s += E::load(a.O, o0 + x * a.os[3]) * E::load(a.dO, g0 + x * a.dos[3]);
    ((A *)a.D)[r] = s;
}

// MODE 0: key-block owner (dK, dV).  MODE 1: query-block owner (dQ).
template <typename E, int MODE> __global__ __launch_bounds__(256) void bwd_main_kernel(const GArgs a) {
    using A = typename E::acc_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int TB = a.TB, d = a.d, ld = d + 1, N = a.N, tid = threadIdx.x;
    A *own0 = (A *)smem_raw;        // owner rows, first operand  (MODE 0: K_j ; MODE 1: Q_i)
    A *own1 = own0 + TB * ld;       //             second operand (MODE 0: V_j ; MODE 1: dO_i)
    A *swp0 = own1 + TB * ld;       // swept rows                 (MODE 0: Q_i ; MODE 1: K_j)
    A *swp1 = swp0 + TB * ld;       //                            (MODE 0: dO_i; MODE 1: V_j)
    A *acc0 = swp1 + TB * ld;       // outputs: MODE 0 dK, MODE 1 dQ
    A *acc1 = acc0 + TB * ld;       //          MODE 0 dV
    A *Ps = acc1 + TB * ld;         // [query r][key c]
    A *dSs = Ps + TB * TB;
    A *Lq = dSs + TB * TB;          // per query row of the current pairing
    A *Dq = Lq + TB;
    A *Rs = Dq + TB;                // MODE 1: rowsum(P) of the owned query rows (see Lc below)

    const int b = blockIdx.y, h = blockIdx.z, blk = blockIdx.x;
    const int64_t qb = b * a.qs[0] + h * a.qs[1], kb = b * a.ks[0] + h * a.ks[1], vb = b * a.vs[0] + h * a.vs[1];
    const int64_t gb = b * a.dos[0] + h * a.dos[1];
    const int64_t lb = b * a.ls[0] + h * a.ls[1];
    const long long db = ((long long)b * a.H + h) * N;
    // fp32 / fp64 row statistic: written by the MODE 1 launch (which runs first), read by MODE 0.  The forward stores
    // L in the I/O dtype (kernels.py:108); its rounding scales a whole row of P (bf16: by up to 2^0.125).  The
    // query-owner kernel sees full rows, measures rowsum(P) = 2^(L_true - L_stored) and hands L + log2(rowsum) on.
    A *Lc = (A *)a.D + (long long)gridDim.y * a.H * N + db;
    const A c_s = (A)a.c_log2e, scale = (A)a.scale;

    auto load_rows = [&](A *dst, const void *src, int64_t base, const int64_t *st, int row0) {
        for (int e = tid; e < TB * d; e += 256) {
            const int r = e / d, x = e % d, row = row0 + r;
            dst[r * ld + x] = row < N ? E::load(src, base + row * st[2] + x * st[3]) : (A)0;
        }
    };
    const int own_row0 = blk * TB;
    if (MODE == 0) {
        load_rows(own0, a.K, kb, a.ks, own_row0);
        load_rows(own1, a.V, vb, a.vs, own_row0);
    } else {
        load_rows(own0, a.Q, qb, a.qs, own_row0);
        load_rows(own1, a.dO, gb, a.dos, own_row0);
        if (tid < TB) {
            const int row = own_row0 + tid;
            Lq[tid] = row < N ? E::load(a.L, lb + row) : (A)0;
            Dq[tid] = row < N ? ((const A *)a.D)[db + row] : (A)0;
            Rs[tid] = (A)0;
        }
    }
    for (int e = tid; e < TB * ld; e += 256) acc0[e] = acc1[e] = (A)0;
    __syncthreads();

    const int nsw = (N + TB - 1) / TB;
    // causal: a key block only meets query rows >= its first key; a query block only keys <= its last row
    const int sw_begin = (MODE == 0 && a.causal) ? own_row0 / TB : 0;
    const int sw_end = (MODE == 1 && a.causal) ? ((own_row0 + TB - 1 < N ? own_row0 + TB - 1 : N - 1) / TB + 1) : nsw;
    for (int sw = sw_begin; sw < sw_end; ++sw) {
        const int sw_row0 = sw * TB;
        if (MODE == 0) {
            load_rows(swp0, a.Q, qb, a.qs, sw_row0);
            load_rows(swp1, a.dO, gb, a.dos, sw_row0);
            if (tid < TB) {
                const int row = sw_row0 + tid;
                Lq[tid] = row < N ? Lc[row] : (A)0;
                Dq[tid] = row < N ? ((const A *)a.D)[db + row] : (A)0;
            }
        } else {
            load_rows(swp0, a.K, kb, a.ks, sw_row0);
            load_rows(swp1, a.V, vb, a.vs, sw_row0);
        }
        __syncthreads();
        if (tid < TB * TB) {  // one (query r, key c) pair per thread
            const int r = tid / TB, c = tid % TB;
            const A *qrow = (MODE == 0 ? swp0 : own0) + r * ld, *krow = (MODE == 0 ? own0 : swp0) + c * ld;
            const A *grow = (MODE == 0 ? swp1 : own1) + r * ld, *vrow = (MODE == 0 ? own1 : swp1) + c * ld;
            const int qi = (MODE == 0 ? sw_row0 : own_row0) + r, kj = (MODE == 0 ? own_row0 : sw_row0) + c;
            A s = 0, dp = 0;
            for (int x = 0; x < d; ++x) {
                s += qrow[x] * krow[x];   // kernels.py:283
                dp += grow[x] * vrow[x];  // :289
            }
            A p = exp2_a<A>(s * c_s - Lq[r]);  // :285
            if (qi >= N || kj >= N || (a.causal && kj > qi)) p = 0;
            Ps[r * TB + c] = p;  // unrounded: MODE 1 sums it; the cast of :287 happens where it is consumed
            dSs[r * TB + c] = E::round(p * (dp - Dq[r]) * scale);  // :291
        }
        __syncthreads();
        if (MODE == 1 && tid < TB) {
            A t = 0;
            for (int c = 0; c < TB; ++c) t += Ps[tid * TB + c];
            Rs[tid] += t;
        }
        for (int e = tid; e < TB * d; e += 256) {
            const int o = e / d, x = e % d;
            if (MODE == 0) {  // o = key c:  dV[c] += sum_r P[r][c] dO[r],  dK[c] += sum_r dS[r][c] Q[r]   (:287, :293)
                A av = 0, ak = 0;
                for (int r = 0; r < TB; ++r) {
                    av += E::round(Ps[r * TB + o]) * swp1[r * ld + x];
                    ak += dSs[r * TB + o] * swp0[r * ld + x];
                }
                acc1[o * ld + x] += av;
                acc0[o * ld + x] += ak;
            } else {  // o = query r:  dQ[r] += sum_c dS[r][c] K[c]   (:317)
                A aq = 0;
                for (int c = 0; c < TB; ++c) aq += dSs[o * TB + c] * swp0[c * ld + x];
                acc0[o * ld + x] += aq;
            }
        }
        __syncthreads();
    }
    for (int e = tid; e < TB * d; e += 256) {
        const int o = e / d, x = e % d, row = own_row0 + o;
        if (row >= N) continue;
        if (MODE == 0) {
            E::store(a.dK, b * a.dks[0] + h * a.dks[1] + row * a.dks[2] + x * a.dks[3], acc0[o * ld + x]);
            E::store(a.dV, b * a.dvs[0] + h * a.dvs[1] + row * a.dvs[2] + x * a.dvs[3], acc1[o * ld + x]);
        } else {
            E::store(a.dQ, b * a.dqs[0] + h * a.dqs[1] + row * a.dqs[2] + x * a.dqs[3], acc0[o * ld + x] / Rs[o]);
        }
    }
    if (MODE == 1 && tid < TB && own_row0 + tid < N) Lc[own_row0 + tid] = Lq[tid] + log2_a<A>(Rs[tid]);
}

template <typename E> int launch_e(const Fa2BwdProblem &p, GArgs &a) {
    using A = typename E::acc_t;
    int TB = 16;
    auto need = [&](int tb) { return (size_t)(6 * tb * (p.d + 1) + 2 * tb * tb + 3 * tb) * sizeof(A); };
    while (TB > 1 && need(TB) > 60 * 1024) TB >>= 1;
    a.TB = TB;
    const size_t smem = need(TB);
    const long long rows = (long long)p.B * p.H * p.N;
    hipLaunchKernelGGL((bwd_D_kernel<E>), dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, p.stream, a, rows);
    const dim3 grid((p.N + TB - 1) / TB, p.B, p.H);
    hipLaunchKernelGGL((bwd_main_kernel<E, 1>), grid, dim3(256), smem, p.stream, a);  // first: leaves Lc for MODE 0
    hipLaunchKernelGGL((bwd_main_kernel<E, 0>), grid, dim3(256), smem, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("generic backward launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

}  // namespace

int fa2_bwd_launch_generic(const Fa2BwdProblem &p) {
    if (p.B > 65535 || p.H > 65535) {
        fa2_set_error("generic backward: B and H must be <= 65535");
        return FA2_ERR_BAD_ARG;
    }
    GArgs a;
    a.Q = p.Q; a.K = p.K; a.V = p.V; a.O = p.O; a.dO = p.dO; a.L = p.L;
    a.dQ = p.dQ; a.dK = p.dK; a.dV = p.dV; a.D = p.D;
    for (int k = 0; k < 4; ++k) {
        a.qs[k] = p.qs[k]; a.ks[k] = p.ks[k]; a.vs[k] = p.vs[k]; a.os[k] = p.os[k]; a.dos[k] = p.dos[k];
        a.dqs[k] = p.dqs[k]; a.dks[k] = p.dks[k]; a.dvs[k] = p.dvs[k];
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.H = p.H; a.N = p.N; a.d = p.d; a.causal = p.causal;
    a.c_log2e = (double)p.scale * FA2_LOG2E;
    a.scale = (double)p.scale;
    switch (p.dtype) {
    case FA2_DTYPE_F32: return launch_e<ElemF32>(p, a);
    case FA2_DTYPE_F16: return launch_e<ElemF16>(p, a);
    case FA2_DTYPE_BF16: return launch_e<ElemBF16>(p, a);
    case FA2_DTYPE_F64: return launch_e<ElemF64>(p, a);
    default: fa2_set_error("backward: dtype enum %d is not supported", p.dtype); return FA2_ERR_UNSUPPORTED;
    }
}
