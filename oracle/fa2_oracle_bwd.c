/*
 * fa2_oracle_bwd.c -- CPU restatement of the reference's Flash-Attention-2 BACKWARD kernels.
 *
 * TEST INFRASTRUCTURE ONLY (same rule as fa2_oracle.c): only tests/ and bench.py's cpu_baseline leg use it.
 *
 * What it restates (line numbers: /root/reference/src/flash_attention_kernels.py):
 *   bwd_D_kernel  :115-166   D_i = rowsum(O_i * dO_i), stored in the I/O dtype (:165-166)
 *   bwd_kernel    :174-334   one "program" per (j, b, h) = key block j of B_c rows:
 *     :268-271  K_j, V_j loaded; dK_j = dV_j = 0 in the I/O dtype
 *     :276-334  for each query block i (B_r rows):
 *     :283        S  = dot(Q_i, K_j^T) * LOG2_e                      (fp32; scale = 1)
 *     :285        P  = exp2(S - L_i)                                  (L is the forward's log2-domain LSE)
 *     :287        dV_j += dot(cast(P^T), dO_i, out_dtype = I/O dtype)
 *     :289        dP = dot(dO_i, V_j^T)                               (fp32)
 *     :291        dS = P * (dP - D_i)
 *     :293        dK_j += dot(cast(dS^T), Q_i, out_dtype = I/O dtype)
 *     :308-320    dQ_i = (0 if first writer else load dQ_i) + dot(cast(dS), K_j, out_dtype = I/O dtype); store
 *   Programs are taken in grid order j = 0, 1, ... (what the Triton interpreter does; on a GPU the dQ order is
 *   whatever the lock hands out).  "out_dtype = I/O dtype" is restated as: accumulate the dot in fp32, round the
 *   result to the I/O dtype, add, round -- for fp32 every rounding is the identity and the restatement is the
 *   plain fp32 chain.
 *
 * Extensions (reference-preserving defaults): causal != 0 masks key > query before exp2 (P = 0 there);
 * scale: S = dot * fl32(scale * log2 e), dS *= scale.
 *
 * Layout: contiguous (B, H, N, d) arrays, L and D contiguous (B, H, N).  Values travel as float.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

float fa2_oracle_round(float x, int dt); /* fa2_oracle.c */

#define LOG2_E 1.4426950408889634

/* kernels.py:115-166 */
int fa2_oracle_bwd_D(const float *O, const float *dO, float *D, int B, int H, int N, int d, int dtype) {
    for (int64_t r = 0; r < (int64_t)B * H * N; ++r) {
        float s = 0.0f;
        for (int c = 0; c < d; ++c) s += O[r * d + c] * dO[r * d + c];
        D[r] = fa2_oracle_round(s, dtype);
    }
    return 0;
}

/* kernels.py:174-334 */
int fa2_oracle_bwd(const float *Q, const float *K, const float *V, const float *dO, const float *L,
                   const float *D, float *dQ, float *dK, float *dV, int B, int H, int N, int d, int dtype,
                   int causal, float scale, int B_r, int B_c) {
    if (B_r <= 0 || B_c <= 0 || N % B_r || N % B_c) return -1;
    const float c_s = (float)((double)scale * LOG2_E);
    float *P = (float *)malloc(sizeof(float) * B_r * B_c), *dS = (float *)malloc(sizeof(float) * B_r * B_c);
    float *dKj = (float *)malloc(sizeof(float) * B_c * d), *dVj = (float *)malloc(sizeof(float) * B_c * d);
    if (!P || !dS || !dKj || !dVj) return -2;
    for (int b = 0; b < B; ++b)
        for (int h = 0; h < H; ++h) {
            const int64_t base = ((int64_t)b * H + h) * N;
            const float *q = Q + base * d, *k = K + base * d, *v = V + base * d, *go = dO + base * d;
            const float *l = L + base, *dd = D + base;
            float *gq = dQ + base * d, *gk = dK + base * d, *gv = dV + base * d;
            for (int j = 0; j < N / B_c; ++j) { /* program (j, b, h) */
                memset(dKj, 0, sizeof(float) * B_c * d);
                memset(dVj, 0, sizeof(float) * B_c * d);
                for (int i = 0; i < N / B_r; ++i) {
                    for (int r = 0; r < B_r; ++r)
                        for (int c = 0; c < B_c; ++c) {
                            const int qi = i * B_r + r, kj = j * B_c + c;
                            float s = 0.0f, dp = 0.0f;
                            for (int x = 0; x < d; ++x) {
                                s += q[(int64_t)qi * d + x] * k[(int64_t)kj * d + x];    /* :283 */
                                dp += go[(int64_t)qi * d + x] * v[(int64_t)kj * d + x];  /* :289 */
                            }
                            float p = exp2f(s * c_s - l[qi]);                            /* :285 */
                            if (causal && kj > qi) p = 0.0f;
                            P[r * B_c + c] = p;
                            dS[r * B_c + c] = p * (dp - dd[qi]) * scale;                 /* :291 */
                        }
                    for (int c = 0; c < B_c; ++c) /* :287, :293 */
                        for (int x = 0; x < d; ++x) {
                            float av = 0.0f, ak = 0.0f;
                            for (int r = 0; r < B_r; ++r) {
                                const int qi = i * B_r + r;
                                av += fa2_oracle_round(P[r * B_c + c], dtype) * go[(int64_t)qi * d + x];
                                ak += fa2_oracle_round(dS[r * B_c + c], dtype) * q[(int64_t)qi * d + x];
                            }
                            dVj[c * d + x] = fa2_oracle_round(dVj[c * d + x] + fa2_oracle_round(av, dtype), dtype);
                            dKj[c * d + x] = fa2_oracle_round(dKj[c * d + x] + fa2_oracle_round(ak, dtype), dtype);
                        }
                    for (int r = 0; r < B_r; ++r) /* :308-320 */
                        for (int x = 0; x < d; ++x) {
                            const int qi = i * B_r + r;
                            float aq = 0.0f;
                            for (int c = 0; c < B_c; ++c)
                                aq += fa2_oracle_round(dS[r * B_c + c], dtype) * k[(int64_t)(j * B_c + c) * d + x];
                            const float prev = j == 0 ? 0.0f : gq[(int64_t)qi * d + x];
                            gq[(int64_t)qi * d + x] = fa2_oracle_round(prev + fa2_oracle_round(aq, dtype), dtype);
                        }
                }
                for (int c = 0; c < B_c; ++c) /* :331-332 */
                    for (int x = 0; x < d; ++x) {
                        gk[(int64_t)(j * B_c + c) * d + x] = dKj[c * d + x];
                        gv[(int64_t)(j * B_c + c) * d + x] = dVj[c * d + x];
                    }
            }
        }
    free(P); free(dS); free(dKj); free(dVj);
    return 0;
}
