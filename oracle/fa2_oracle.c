/*
 * fa2_oracle.c -- CPU restatement of the reference's Flash-Attention-2 forward kernel.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (flash_attention_dlrs_amd/,
 * the C-ABI library libfa2_hip.so) may link, import or call this file.  Only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py use it, and only as a checker.
 *
 * What it restates (all line numbers are /root/reference/src/flash_attention_kernels.py):
 *   :38-40   one "program" per (i, b, h): Q tile i of B_r rows of batch b, head h
 *   :84-86   O_i = 0 (fp32), m_i = -inf, l_i = 0
 *   :88-101  for j in range(T_c):
 *   :92         S  = dot(Q_i, K_j^T) * LOG2_e            (no 1/sqrt(d): scale = 1)
 *   :93         m' = max(m, rowmax(S))
 *   :94         P  = exp2(S - m')
 *   :95         c  = exp2(m - m')
 *   :96         l  = c*l + rowsum(P)
 *   :97-98      O  = c*O + cast(P -> V dtype, RTNE) @ V_j   (fp32 accumulate)
 *   :99         m  = m'
 *   :105-108 O /= l ; L = m + log2(l) ; both cast to the I/O dtype on store
 *
 * Extensions beyond the reference (BASELINE.json configs c3..c5), reference-preserving defaults:
 *   causal != 0 : S[r][c] = -inf where key index > query index (applied before the row max)
 *   scale       : S = dot * fl32(scale * log2(e))   (scale = 1 reproduces :92 bit for bit)
 *
 * Pinning: tests/test_oracle.py checks this file against tests/golden/ (vectors produced by
 * running the reference fwd_kernel itself under TRITON_INTERPRET=1 and by torch SDPA(scale=1),
 * see tests/golden/gen_golden.py).
 *
 * Values travel as float (fp16 / bf16 / fp8 values are exactly representable); the dtype enum
 * only selects the rounding applied to P before P@V and to O, L on store.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Same numbering as include/fa2_fwd.h (FA2_DTYPE_*). */
enum { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2, DT_F8E5M2 = 3, DT_F8E4M3 = 4, DT_F64 = 5 };

/* Round a float to a narrower binary floating-point format, round-to-nearest-even.
 * mbits = explicit mantissa bits, emin = exponent of the smallest normal, maxv = largest finite. */
static float round_fmt(float x, int mbits, int emin, float maxv, int has_inf) {
    if (x == 0.0f || isnan(x) || isinf(x)) {
        if (isinf(x) && !has_inf) return NAN;
        return x;
    }
    int e;
    (void)frexpf(fabsf(x), &e); /* |x| = f * 2^e, f in [0.5, 1) -> floor(log2|x|) = e-1 */
    e -= 1;
    if (e < emin) e = emin;
    const float q = ldexpf(1.0f, e - mbits);
    float r = nearbyintf(x / q) * q; /* default rounding mode = RTNE; x/q is exact (power of 2) */
    if (fabsf(r) > maxv) r = has_inf ? copysignf(INFINITY, x) : NAN;
    return r;
}

static float round_dtype(float x, int dt) {
    switch (dt) {
    case DT_F16: return round_fmt(x, 10, -14, 65504.0f, 1);
    case DT_BF16: return round_fmt(x, 7, -126, 3.3895313892515355e38f, 1);
    case DT_F8E5M2: return round_fmt(x, 2, -14, 57344.0f, 1);
    case DT_F8E4M3: return round_fmt(x, 3, -6, 448.0f, 0);
    default: return x;
    }
}

/* Exposed so tests can pin the rounding helper against torch's casts. */
float fa2_oracle_round(float x, int dt) { return round_dtype(x, dt); }

#define LOG2_E 1.4426950408889634 /* np.log2(np.e), kernels.py:9 */

/*
 * Strides are in ELEMENTS, as in the reference launch (flash_attention_torch.py:53-57).
 * Returns 0, or -1 on a bad argument.  B_r / B_c are the reference's tile meta-parameters
 * (autotune_configs.py:24-140); the result depends on them only through fp32 rounding.
 */
int fa2_oracle_fwd(const float *Q, const float *K, const float *V, float *O, float *L,
                   const int64_t qs[4], const int64_t ks[4], const int64_t vs[4],
                   const int64_t os[4], const int64_t ls[2], int B, int H, int N, int d, int dtype,
                   int causal, float scale, int B_r, int B_c) {
    if (!Q || !K || !V || !O || !L || B <= 0 || H <= 0 || N <= 0 || d <= 0 || B_r <= 0 || B_c <= 0)
        return -1;
    if (N % B_r || N % B_c) return -1; /* autotune_configs.py:184-187 */
    const float c_log2e = (float)((double)scale * LOG2_E);
    float *S = (float *)malloc(sizeof(float) * (size_t)B_r * B_c);
    float *Oi = (float *)malloc(sizeof(float) * (size_t)B_r * d);
    float *m = (float *)malloc(sizeof(float) * B_r);
    float *l = (float *)malloc(sizeof(float) * B_r);
    if (!S || !Oi || !m || !l) return -1;
    const int T_r = N / B_r, T_c = N / B_c;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int i = 0; i < T_r; ++i) {
                const float *Qb = Q + b * qs[0] + h * qs[1];
                const float *Kb = K + b * ks[0] + h * ks[1];
                const float *Vb = V + b * vs[0] + h * vs[1];
                for (int r = 0; r < B_r; ++r) {
                    m[r] = -INFINITY;
                    l[r] = 0.0f;
                }
                memset(Oi, 0, sizeof(float) * (size_t)B_r * d);
                for (int j = 0; j < T_c; ++j) {
                    for (int r = 0; r < B_r; ++r) {
                        const int qrow = i * B_r + r;
                        const float *q = Qb + (int64_t)qrow * qs[2];
                        float mx = -INFINITY;
                        for (int c = 0; c < B_c; ++c) {
                            const int krow = j * B_c + c;
                            const float *k = Kb + (int64_t)krow * ks[2];
                            float acc = 0.0f;
                            for (int x = 0; x < d; ++x) acc += q[x * qs[3]] * k[x * ks[3]];
                            float s = acc * c_log2e; /* :92 */
                            if (causal && krow > qrow) s = -INFINITY;
                            S[r * B_c + c] = s;
                            if (s > mx) mx = s;
                        }
                        const float m_new = m[r] > mx ? m[r] : mx; /* :93 */
                        const float coeff = exp2f(m[r] - m_new);   /* :95 */
                        float rs = 0.0f;
                        for (int c = 0; c < B_c; ++c) {
                            const float p = exp2f(S[r * B_c + c] - m_new); /* :94 */
                            rs += p;
                            S[r * B_c + c] = round_dtype(p, dtype); /* :98 cast(P) */
                        }
                        l[r] = coeff * l[r] + rs; /* :96 */
                        float *o = Oi + (size_t)r * d;
                        for (int x = 0; x < d; ++x) o[x] *= coeff; /* :97 */
                        for (int c = 0; c < B_c; ++c) {
                            const float p = S[r * B_c + c];
                            if (p == 0.0f) continue; /* masked / underflowed: contributes exactly 0 */
                            const float *v = Vb + (int64_t)(j * B_c + c) * vs[2];
                            for (int x = 0; x < d; ++x) o[x] += p * v[x * vs[3]];
                        }
                        m[r] = m_new; /* :99 */
                    }
                }
                for (int r = 0; r < B_r; ++r) {
                    const int qrow = i * B_r + r;
                    float *o = O + b * os[0] + h * os[1] + (int64_t)qrow * os[2];
                    for (int x = 0; x < d; ++x)
                        o[x * os[3]] = round_dtype(Oi[(size_t)r * d + x] / l[r], dtype); /* :105,:107 */
                    L[b * ls[0] + h * ls[1] + qrow] = round_dtype(m[r] + log2f(l[r]), dtype); /* :106,:108 */
                }
            }
    free(S);
    free(Oi);
    free(m);
    free(l);
    return 0;
}

/*
 * The same algorithm with the two liberties the gfx950 MFMA kernels take, made explicit so that those kernels can be
 * compared ELEMENT-WISE (tests/test_a64_parity.py, tests/test_fwd_parity.py fp8) instead of through a tolerance:
 *
 *   deferred running maximum   :93/:95/:99 raise m for every tile.  The kernels keep m while no row of a G-row group
 *                              (one wave's query block) has rowmax(S) - m > thr, so P = exp2(S - m) may reach 2^thr
 *                              (fa2_a64.hip: thr = 60 bf16 / 15.875 f16; fa2_mfma8x.hip: kThr = 6); when one does, every
 *                              row of the group takes m' = max(m, rowmax(S)).  thr < 0 raises m for every tile (= :93).
 *   one rounding in S - m      exp2(fma(dot, c, -m)) instead of :92 + :94's two roundings;
 *   sum_rounded != 0           l accumulates cast(P) (the row sums ride on the matrix pipe with the rounded P) instead
 *                              of :96's unrounded P.
 * Everything else (fp32 state, RTNE casts, :105-108) is the function above.  N need not be a multiple of G or B_c.
 */
int fa2_oracle_fwd_deferred(const float *Q, const float *K, const float *V, float *O, float *L,
                            const int64_t qs[4], const int64_t ks[4], const int64_t vs[4],
                            const int64_t os[4], const int64_t ls[2], int B, int H, int N, int d, int dtype,
                            int causal, float scale, int G, int B_c, float thr, int sum_rounded, int ceil_m) {
    if (!Q || !K || !V || !O || !L || B <= 0 || H <= 0 || N <= 0 || d <= 0 || G <= 0 || B_c <= 0) return -1;
    const float c_log2e = (float)((double)scale * LOG2_E);
    float *S = (float *)malloc(sizeof(float) * (size_t)G * B_c);
    float *Oi = (float *)malloc(sizeof(float) * (size_t)G * d);
    float *m = (float *)malloc(sizeof(float) * G);
    float *l = (float *)malloc(sizeof(float) * G);
    float *mx = (float *)malloc(sizeof(float) * G);
    if (!S || !Oi || !m || !l || !mx) return -1;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int q0 = 0; q0 < N; q0 += G) {
                const float *Qb = Q + b * qs[0] + h * qs[1];
                const float *Kb = K + b * ks[0] + h * ks[1];
                const float *Vb = V + b * vs[0] + h * vs[1];
                const int rows = N - q0 < G ? N - q0 : G;
                for (int r = 0; r < rows; ++r) {
                    m[r] = -INFINITY;
                    l[r] = 0.0f;
                }
                memset(Oi, 0, sizeof(float) * (size_t)G * d);
                const int kend = causal ? (q0 + rows < N ? q0 + rows : N) : N;
                for (int k0 = 0; k0 < kend; k0 += B_c) {
                    const int cols = kend - k0 < B_c ? kend - k0 : B_c;
                    int fire = 0;
                    for (int r = 0; r < rows; ++r) {
                        const int qrow = q0 + r;
                        const float *q = Qb + (int64_t)qrow * qs[2];
                        mx[r] = -INFINITY;
                        for (int c = 0; c < cols; ++c) {
                            const int krow = k0 + c;
                            const float *k = Kb + (int64_t)krow * ks[2];
                            float acc = 0.0f;
                            for (int x = 0; x < d; ++x) acc += q[x * qs[3]] * k[x * ks[3]];
                            if (causal && krow > qrow) acc = -INFINITY;
                            S[r * B_c + c] = acc;
                            if (acc > mx[r]) mx[r] = acc;
                        }
                        mx[r] *= c_log2e;
                        if (!(mx[r] - m[r] <= thr)) fire = 1;
                    }
                    for (int r = 0; r < rows; ++r) {
                        float *o = Oi + (size_t)r * d;
                        if (fire || thr < 0.0f) {
                            /* ceil_m: the block-scaled fp8 kernel (fa2_a8_gen.py) keeps the running maximum an integer, so that
                             * every factor is an exact power of two and O is scaled through the MFMA's block scale instead */
                            const float mxr = ceil_m ? ceilf(mx[r]) : mx[r];
                            const float m_new = m[r] > mxr ? m[r] : mxr;
                            const float coeff = exp2f(m[r] - m_new);
                            l[r] *= coeff;
                            for (int x = 0; x < d; ++x) o[x] *= coeff;
                            m[r] = m_new;
                        }
                        for (int c = 0; c < cols; ++c) {
                            const float p = exp2f(fmaf(S[r * B_c + c], c_log2e, -m[r]));
                            const float pr = round_dtype(p, dtype);
                            l[r] += sum_rounded ? pr : p;
                            if (pr == 0.0f) continue;
                            const float *v = Vb + (int64_t)(k0 + c) * vs[2];
                            for (int x = 0; x < d; ++x) o[x] += pr * v[x * vs[3]];
                        }
                    }
                }
                for (int r = 0; r < rows; ++r) {
                    const int qrow = q0 + r;
                    float *o = O + b * os[0] + h * os[1] + (int64_t)qrow * os[2];
                    for (int x = 0; x < d; ++x) o[x * os[3]] = round_dtype(Oi[(size_t)r * d + x] / l[r], dtype);
                    L[b * ls[0] + h * ls[1] + qrow] = round_dtype(m[r] + log2f(l[r]), dtype);
                }
            }
    free(S);
    free(Oi);
    free(m);
    free(l);
    free(mx);
    return 0;
}

/*
 * fp64 entry.  The reference maps torch.float64 (flash_attention_torch.py:8-9) but its kernel
 * cannot run that dtype (tl.dot with an fp32 accumulator asserts; verified under the Triton
 * interpreter) -> "parity unpinned".  We define it as the same algorithm carried in double.
 */
int fa2_oracle_fwd_f64(const double *Q, const double *K, const double *V, double *O, double *L,
                       const int64_t qs[4], const int64_t ks[4], const int64_t vs[4],
                       const int64_t os[4], const int64_t ls[2], int B, int H, int N, int d,
                       int causal, double scale) {
    if (!Q || !K || !V || !O || !L || B <= 0 || H <= 0 || N <= 0 || d <= 0) return -1;
    const double c = scale * LOG2_E;
    double *S = (double *)malloc(sizeof(double) * N);
    if (!S) return -1;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h)
            for (int r = 0; r < N; ++r) {
                const double *q = Q + b * qs[0] + h * qs[1] + (int64_t)r * qs[2];
                double mx = -INFINITY;
                for (int k = 0; k < N; ++k) {
                    const double *kk = K + b * ks[0] + h * ks[1] + (int64_t)k * ks[2];
                    double acc = 0.0;
                    for (int x = 0; x < d; ++x) acc += q[x * qs[3]] * kk[x * ks[3]];
                    acc *= c;
                    if (causal && k > r) acc = -INFINITY;
                    S[k] = acc;
                    if (acc > mx) mx = acc;
                }
                double lsum = 0.0;
                for (int k = 0; k < N; ++k) {
                    S[k] = exp2(S[k] - mx);
                    lsum += S[k];
                }
                double *o = O + b * os[0] + h * os[1] + (int64_t)r * os[2];
                for (int x = 0; x < d; ++x) {
                    double acc = 0.0;
                    for (int k = 0; k < N; ++k)
                        acc += S[k] * V[b * vs[0] + h * vs[1] + (int64_t)k * vs[2] + x * vs[3]];
                    o[x * os[3]] = acc / lsum;
                }
                L[b * ls[0] + h * ls[1] + r] = mx + log2(lsum);
            }
    free(S);
    return 0;
}
