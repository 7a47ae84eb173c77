// fa2_generic.hip -- the catch-all FA-2 forward kernel for gfx950.
//
// Covers everything the reference's host glue can hand to its Triton kernel
// (src/flash_attention_torch.py:24-47): every dtype of convert_triton_dtype (:7-18) plus bf16 / e4m3,
// arbitrary element strides on all four axes (src/flash_attention_kernels.py:45-79), any d in
// [1, 512] (the reference pads to a power of two on the host, torch.py:38), and -- beyond the reference, whose tiles need
// N % 16 == 0 (src/autotune_configs.py:184-187) -- any N >= 1.  It is the fallback behind the MFMA
// kernels (fa2_mfma16.hip, fa2_mfma32.hip), not the fast path: contractions run on the VALU.
//
// Arithmetic = the reference's, statement for statement (kernels.py:84-108): fp32 state (m, l, O),
// S = dot * log2e, exp2-domain online softmax, P rounded to the I/O dtype (RTNE) before P@V, O / l
// and L = m + log2 l rounded to the I/O dtype on store.  (float64 is carried in double: the
// reference cannot run that dtype at all, see oracle/fa2_oracle.c.)
//
// Work split: grid (ceil(N/16), B, H) -- axis order of kernels.py:38-40.  A 256-thread workgroup
// owns 16 query rows, 4 per wave.  Per 64-key tile: lane = key for the scores (one K row per lane,
// Q rows broadcast from LDS), wave shuffle reduction for the row max, then lane = output column
// for P@V with P broadcast from LDS.
#include <math.h>

#include "fa2_common.h"
#include "fa2_elem.h"

namespace {

constexpr int kRowsPerWave = 4;
constexpr int kWaves = 4;
constexpr int kBr = kRowsPerWave * kWaves;  // 16 = the reference's smallest B_r
constexpr int kBc = 64;                     // one key per lane

template <typename A> __device__ __forceinline__ A exp2_acc(A x);
template <> __device__ __forceinline__ float exp2_acc<float>(float x) { return exp2f(x); }
template <> __device__ __forceinline__ double exp2_acc<double>(double x) { return exp2(x); }
template <typename A> __device__ __forceinline__ A log2_acc(A x);
template <> __device__ __forceinline__ float log2_acc<float>(float x) { return log2f(x); }
template <> __device__ __forceinline__ double log2_acc<double>(double x) { return log2(x); }

template <typename A> __device__ __forceinline__ A wave_max(A v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const A t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}
template <typename A> __device__ __forceinline__ A wave_sum(A v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct GenericArgs {
    const void *Q, *K, *V;
    void *O, *L;
    int64_t qs[4], ks[4], vs[4], os[4], ls[2];
    int N, d, causal;
    double c_log2e;  // scale * log2(e); rounded to float for the fp32 dtypes (kernels.py:92)
};

// DPL = output columns per lane = ceil(d / 64).
template <class E, int DPL>
__global__ __launch_bounds__(kWaves * 64) void fa2_fwd_generic_kernel(const GenericArgs a) {
    using A = typename E::acc_t;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    A *q_lds = (A *)smem_raw;                           // [kBr][d]
    A *p_lds = q_lds + (size_t)kBr * a.d;               // [kWaves][kRowsPerWave][kBc]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x, b = blockIdx.y, h = blockIdx.z;  // kernels.py:38-40
    const int N = a.N, d = a.d;
    const char *Qb = (const char *)a.Q;
    const int64_t q_off = b * a.qs[0] + h * a.qs[1];
    const int64_t k_off = b * a.ks[0] + h * a.ks[1];
    const int64_t v_off = b * a.vs[0] + h * a.vs[1];
    const int64_t o_off = b * a.os[0] + h * a.os[1];
    (void)Qb;

    // Q tile -> LDS (rows past N are clamped; their results are never stored).
    for (int idx = tid; idx < kBr * d; idx += kWaves * 64) {
        const int r = idx / d, x = idx - r * d;
        int row = i * kBr + r;
        row = row < N ? row : N - 1;
        q_lds[idx] = E::load(a.Q, q_off + (int64_t)row * a.qs[2] + (int64_t)x * a.qs[3]);
    }
    __syncthreads();

    const A c = sizeof(A) == 8 ? (A)a.c_log2e : (A)(float)a.c_log2e;
    const int row0 = i * kBr + wave * kRowsPerWave;
    A m[kRowsPerWave], lsum[kRowsPerWave], o[kRowsPerWave][DPL];
#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        m[r] = -INFINITY;
        lsum[r] = 0;
#pragma unroll
        for (int cc = 0; cc < DPL; ++cc) o[r][cc] = 0;
    }

    // Same trip count for every wave of the workgroup (barriers inside).
    int kend = N;
    if (a.causal) {
        const int lim = i * kBr + kBr;
        kend = lim < N ? lim : N;
    }
    A *p_w = p_lds + (size_t)wave * kRowsPerWave * kBc;

    for (int kt = 0; kt < kend; kt += kBc) {
        const int key = kt + lane;
        const bool valid = key < N;
        A dot[kRowsPerWave];
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) dot[r] = 0;
        if (valid) {
            const int64_t kb = k_off + (int64_t)key * a.ks[2];
            for (int x = 0; x < d; ++x) {
                const A kx = E::load(a.K, kb + (int64_t)x * a.ks[3]);
#pragma unroll
                for (int r = 0; r < kRowsPerWave; ++r)
                    dot[r] += q_lds[(wave * kRowsPerWave + r) * d + x] * kx;
            }
        }
#pragma unroll
        for (int r = 0; r < kRowsPerWave; ++r) {
            A s = dot[r] * c;  // kernels.py:92
            if (!valid || (a.causal && key > row0 + r)) s = -INFINITY;
            const A mx = wave_max(s);
            const A m_new = m[r] > mx ? m[r] : mx;           // :93
            const A p = exp2_acc<A>(s - m_new);              // :94
            const A coeff = exp2_acc<A>(m[r] - m_new);       // :95
            lsum[r] = coeff * lsum[r] + p;                   // :96 (per-lane partial of the row sum)
#pragma unroll
            for (int cc = 0; cc < DPL; ++cc) o[r][cc] *= coeff;  // :97
            m[r] = m_new;                                    // :99
            p_w[r * kBc + lane] = E::round(p);               // :98 cast(P)
        }
        __syncthreads();
        const int kmax = (N - kt) < kBc ? (N - kt) : kBc;
        for (int kk = 0; kk < kmax; ++kk) {
            const int64_t vb = v_off + (int64_t)(kt + kk) * a.vs[2];
            A pr[kRowsPerWave];
#pragma unroll
            for (int r = 0; r < kRowsPerWave; ++r) pr[r] = p_w[r * kBc + kk];
#pragma unroll
            for (int cc = 0; cc < DPL; ++cc) {
                const int x = lane + 64 * cc;
                if (x < d) {
                    const A v = E::load(a.V, vb + (int64_t)x * a.vs[3]);
#pragma unroll
                    for (int r = 0; r < kRowsPerWave; ++r) o[r][cc] += pr[r] * v;  // :98
                }
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int row = row0 + r;
        const A l = wave_sum(lsum[r]);
        if (row < N) {
#pragma unroll
            for (int cc = 0; cc < DPL; ++cc) {
                const int x = lane + 64 * cc;
                if (x < d)
                    E::store(a.O, o_off + (int64_t)row * a.os[2] + (int64_t)x * a.os[3], o[r][cc] / l);  // :105,:107
            }
            if (lane == 0)
                E::store(a.L, b * a.ls[0] + h * a.ls[1] + row, m[r] + log2_acc<A>(l));  // :106,:108
        }
    }
}

template <class E> int launch_e(const Fa2Problem &p, const GenericArgs &a) {
    const dim3 grid((p.N + kBr - 1) / kBr, p.B, p.H), block(kWaves * 64);
    const size_t smem = sizeof(typename E::acc_t) * ((size_t)kBr * p.d + (size_t)kWaves * kRowsPerWave * kBc);
    // output columns per lane: ceil(d / 64) rounded up to the next instantiation (columns >= d are skipped in the kernel)
    const int dpl = (p.d + 63) / 64;
    if (dpl <= 1) hipLaunchKernelGGL((fa2_fwd_generic_kernel<E, 1>), grid, block, smem, p.stream, a);
    else if (dpl <= 2) hipLaunchKernelGGL((fa2_fwd_generic_kernel<E, 2>), grid, block, smem, p.stream, a);
    else if (dpl <= 4) hipLaunchKernelGGL((fa2_fwd_generic_kernel<E, 4>), grid, block, smem, p.stream, a);
    else if (dpl <= 8) hipLaunchKernelGGL((fa2_fwd_generic_kernel<E, 8>), grid, block, smem, p.stream, a);
    else {
        fa2_set_error("generic kernel: d=%d not in [1, 512]", p.d);
        return FA2_ERR_UNSUPPORTED;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("generic kernel launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

}  // namespace

int fa2_launch_generic(const Fa2Problem &p) {
    if (p.B > 65535 || p.H > 65535) {
        fa2_set_error("generic kernel: B and H must be <= 65535");
        return FA2_ERR_BAD_ARG;
    }
    GenericArgs a;
    a.Q = p.Q; a.K = p.K; a.V = p.V; a.O = p.O; a.L = p.L;
    for (int k = 0; k < 4; ++k) { a.qs[k] = p.qs[k]; a.ks[k] = p.ks[k]; a.vs[k] = p.vs[k]; a.os[k] = p.os[k]; }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.N = p.N; a.d = p.d; a.causal = p.causal;
    a.c_log2e = (double)p.scale * FA2_LOG2E;
    switch (p.dtype) {
    case FA2_DTYPE_F32: return launch_e<ElemF32>(p, a);
    case FA2_DTYPE_F16: return launch_e<ElemF16>(p, a);
    case FA2_DTYPE_BF16: return launch_e<ElemBF16>(p, a);
    case FA2_DTYPE_F8E5M2: return launch_e<ElemF8E5M2>(p, a);
    case FA2_DTYPE_F8E4M3: return launch_e<ElemF8E4M3>(p, a);
    case FA2_DTYPE_F64: return launch_e<ElemF64>(p, a);
    default: fa2_set_error("unknown dtype enum %d", p.dtype); return FA2_ERR_UNSUPPORTED;
    }
}
