// fa2_bwd_mfma16.hip -- FA-2 backward for f16 / bf16, d in {64, 128}, on v_mfma_f32_32x32x16_{bf16,f16}.
//
// Arithmetic: src/flash_attention_kernels.py:115-166 (D = rowsum(dO * O)) and :276-317 of the reference's
// bwd_kernel (S = Q K^T log2e, P = exp2(S - L), dV += cast(P)^T dO, dP = dO V^T, dS = P (dP - D),
// dK += cast(dS)^T Q, dQ += cast(dS) K), fp32 accumulation throughout, P and dS rounded RTNE to the I/O dtype
// before the second contractions.  The reference sums dQ across key-block programs through a lock; here two
// launches each OWN their outputs (include/fa2_bwd.h), so nothing is summed across workgroups:
//
//   MODE 0  "dK, dV": a wave owns 32 KEYS and ONE of the two gradients: waves 0-3 of the 8-wave workgroup hold dK^T
//           of key groups 0-3 (K and V fragments in registers), waves 4-7 hold dV^T of the same key groups (K
//           fragments only); the workgroup sweeps the query rows 64 at a time (Q and dO tiles through LDS).  Both
//           gradients in one wave need 128 accumulators + 64 fragment registers = one wave per SIMD, measured at 22 %
//           of the MFMA peak; split, every wave fits 256 registers, two waves share a SIMD, and the two are a dK wave
//           (24 MFMAs per 32 query rows) and a dV wave (16): different instruction streams that fill each other's
//           gaps.  The price is S = Q K^T computed by both (5 products per block pair instead of 4);
//   MODE 1  "dQ":     a wave owns 32 QUERY rows (Q, dO fragments in registers, dQ^T in 64 accumulators),
//           the workgroup sweeps the keys 64 at a time (K and V tiles through LDS).
//
// Both modes are one code path.  With T0 / T1 the swept tiles (Q / dO or K / V) and f0 / f1 the owned fragments
// (K / V or Q / dO), per 32 swept rows:
//     X0 = T0 . f0^T   (MODE 1: S^T = K Q^T, MODE 0: S = Q K^T)       A = row read of T0, B = f0
//     X1 = T1 . f1^T   (MODE 1: dP^T = V dO^T, MODE 0: dP = dO V^T)   A = row read of T1, B = f1
//     P = exp2(c X0 - L),  dS = P (X1 - D)      -- the owned index is the lane, the swept index the register row,
//     acc0 += T0^T . dS    (dQ^T = K^T dS^T  /  dK^T = Q^T dS)          A = TRANSPOSED read of T0, B = cvt(dS)
//     acc1 += T1^T . P     (MODE 0 only: dV^T = dO^T P)                 A = transposed read of T1, B = cvt(P)
// i.e. the accumulator tiles X feed the second products as B operands with no lane movement (the summation index
// of the second product is X's row index: cdna_hip_programming.md section 3), and one LDS image per tile serves
// the row reads and the transposed reads (the XOR-swizzled image of fa2_mfma16.hip).
#include "fa2_bwd_common.h"

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

template <int V> struct IC { static constexpr int value = V; };

#ifndef FA2_BWD_PIPE
#define FA2_BWD_PIPE 0  // measured: 41-45 spills, 27 % slower (two more score tiles do not fit 256 registers)
#endif

struct BArgs {
    const char *Q, *K, *V, *O, *dO, *L;
    char *dQ, *dK, *dV;
    float *D;
    int64_t qs[3], ks[3], vs[3], os[3], dos[3], dqs[3], dks[3], dvs[3];  // B, H, N strides in BYTES
    int64_t ls[2];                                                        // elements
    int B, H, N, causal;
    float c_log2e, scale;
};

// 16-byte-chunk swizzle of fa2_mfma16.hip: conflict-free ds_read_b128 row reads and ds_read_b64_tr_b16 reads
template <int D> __device__ __forceinline__ int lds_off(int row, int ch) {
    if constexpr (D == 128) return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
    else return row * 128 + ((ch ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3))) << 4);
}

typedef __attribute__((ext_vector_type(4))) int i32x4;
// One LDS-DMA piece (64 lanes x 16 bytes, lane-linear in LDS); inline asm and our own vmcnt wait, as in fa2_mfma16d.hip
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds_base, int voffset) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(voffset), "s"(rsrc)
                 : "memory");
}
__device__ __forceinline__ void dma4(const i32x4 rsrc, unsigned lds_base, int voffset) {  // 64 lanes x 4 bytes
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "v"(voffset), "s"(rsrc)
                 : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ---- D[b, h, n] = sum_x O * dO  (kernels.py:115-166): D/8 lanes per row, 16 bytes of each operand per lane
template <typename T, int D> __global__ __launch_bounds__(256) void bwd_D_kernel(const BArgs a) {
    constexpr int LPR = D / 8;  // lanes per row
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long r = gid / LPR, rows = (long long)a.B * a.H * a.N;
    const int ch = (int)(gid % LPR);
    float s = 0.0f;
    if (r < rows) {
        const int n = (int)(r % a.N);
        const long long bh = r / a.N;
        const int h = (int)(bh % a.H), b = (int)(bh / a.H);
        typedef __attribute__((ext_vector_type(8))) T Tx8;
        const Tx8 o = *(const Tx8 *)(a.O + b * a.os[0] + h * a.os[1] + (int64_t)n * a.os[2] + ch * 16);
        const Tx8 g = *(const Tx8 *)(a.dO + b * a.dos[0] + h * a.dos[1] + (int64_t)n * a.dos[2] + ch * 16);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)o[j] * (float)g[j];
    }
#pragma unroll
    for (int w = 1; w < LPR; w <<= 1) s += __shfl_xor(s, w, 64);
    if (r < rows && ch == 0) a.D[r] = s;
}

template <typename T, int D, int MODE, int NWQ = 4>
__global__ __launch_bounds__(MODE == 0 ? 512 : NWQ * 64, 2) void bwd_mfma16_kernel(const BArgs a) {
    using M = Mma<T>;
    using frag = typename M::frag;
    // owned rows per workgroup (MODE 1: NWQ waves x 32 query rows), swept rows per tile
    constexpr int NW = MODE == 0 ? 8 : NWQ, BO = MODE == 0 ? 128 : NWQ * 32, BS = 64;
    constexpr int ROWB = D * 2, TILEB = BS * ROWB, CPR = ROWB / 16;
    constexpr int KS = D / 16, DB = D / 32;
    constexpr int LOFF = 4 * TILEB;  // LDS: T0[2] | T1[2] | L[2][64] | D[2][64] (floats; MODE 0 only)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    LDS_PTR(char) lds = (LDS_PTR(char))smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: every block / mask decision
                                                                // below is then a scalar branch, not an EXEC mask
    const int i = lane & 31, h = lane >> 5;
    const int N = a.N;

    const int nblk = (N + BO - 1) / BO, nbh = a.B * a.H;
    int bh, blk;
    {
        const int bid = blockIdx.x;
        if ((nbh & 7) == 0) {  // whole (b, h) groups per XCD (speed only)
            const int slot = bid >> 3;
            bh = (slot / nblk) * 8 + (bid & 7);
            blk = slot % nblk;
        } else {
            bh = bid / nblk;
            blk = bid % nblk;
        }
    }
    if (MODE == 1 && (a.causal & 1)) blk = nblk - 1 - blk;  // causal: the last query blocks sweep the most keys -- start them first
    const int b = bh / a.H, hh = bh - b * a.H;
    const bool roleV = MODE == 0 && wave >= 4;   // MODE 0: waves 4..7 accumulate dV^T, waves 0..3 dK^T
    // The dV waves carry the exp2 work and are the second-dispatched half, which loses every issue arbitration to the
    // older half: in-kernel stamps showed them at 2 900 busy cycles per block step against 1 950 for the dK waves, which
    // then waited 1 100 cycles at the barrier.  FA2_BWD_PRIO (default 1): static priority for the dV waves.
    if (MODE == 0 && roleV && (a.causal & 2)) __builtin_amdgcn_s_setprio(1);
    const int own0 = blk * BO + (MODE == 0 ? wave & 3 : wave) * 32;  // first owned row of this wave
    const int orow = own0 + i;              // this lane's owned row (query in MODE 1, key in MODE 0)

    // swept tiles T0, T1 and owned fragments f0, f1
    const char *T0p = (MODE == 1 ? a.K + b * a.ks[0] + hh * a.ks[1] : a.Q + b * a.qs[0] + hh * a.qs[1]);
    const char *T1p = (MODE == 1 ? a.V + b * a.vs[0] + hh * a.vs[1] : a.dO + b * a.dos[0] + hh * a.dos[1]);
    const int64_t t0rs = MODE == 1 ? a.ks[2] : a.qs[2], t1rs = MODE == 1 ? a.vs[2] : a.dos[2];
    const char *F0p = (MODE == 1 ? a.Q + b * a.qs[0] + hh * a.qs[1] : a.K + b * a.ks[0] + hh * a.ks[1]);
    const char *F1p = (MODE == 1 ? a.dO + b * a.dos[0] + hh * a.dos[1] : a.V + b * a.vs[0] + hh * a.vs[1]);
    const int64_t f0rs = MODE == 1 ? a.qs[2] : a.ks[2], f1rs = MODE == 1 ? a.dos[2] : a.vs[2];
    const T *Lp = (const T *)a.L + b * a.ls[0] + hh * a.ls[1];
    const float *Dp = a.D + ((int64_t)b * a.H + hh) * N;
    // fp32 row statistic written by the MODE 1 launch (which runs first) and read by MODE 0: the forward stores L in
    // the I/O dtype (kernels.py:108), whose rounding (bf16: +-0.125 at |L| ~ 50) would scale a whole row of P by up to
    // 2^0.125; the query-owner kernel sees the full row, measures rowsum(P) = 2^(L_true - L_stored) and hands
    // L_stored + log2(rowsum) on, so both kernels work with P normalised to fp32 accuracy.
    float *Lc = a.D + (int64_t)a.B * a.H * N + ((int64_t)b * a.H + hh) * N;

    frag f0[KS], f1[KS];
    {
        const int row = orow < N ? orow : N - 1;
        const char *p0 = F0p + (int64_t)row * f0rs + h * 16, *p1 = F1p + (int64_t)row * f1rs + h * 16;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            f0[ks] = __builtin_bit_cast(frag, *(const u32x4 *)(p0 + ks * 32));
            if (!roleV) f1[ks] = __builtin_bit_cast(frag, *(const u32x4 *)(p1 + ks * 32));
        }
    }
    // per-lane row constants (MODE 1: the owned query's L and D)
    float Lown = 0.0f, Down = 0.0f;
    if (MODE == 1) {
        const int row = orow < N ? orow : N - 1;
        Lown = (float)Lp[row];
        Down = Dp[row];
    }
    // Wait for the owned fragments HERE.  Their first use is inside the sweep loop, and hipcc puts its counted
    // `s_waitcnt vmcnt(7..0)` in front of those MFMAs, in every iteration -- where the counter also holds the LDS-DMA
    // pieces just issued for the next tile (inline asm, invisible to the compiler), so each iteration waited for its own
    // prefetch to land.  An explicit wait the compiler can see empties its scoreboard before the loop.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt / lgkmcnt untouched

    // swept range (causal: MODE 1 keys <= last owned query; MODE 0 queries >= first owned key)
    const int wg0 = blk * BO;
    int t_begin = 0, t_end = (N + BS - 1) / BS;

    // ---- staging of the swept tiles by LDS-DMA (buffer_load ... lds): 1 KiB pieces, wave w issues pieces w, w + NW, ...;
    // lane l of piece p fills LDS (row = RPP p + l / CPR, slot = l % CPR) with global chunk slot ^ f(row) -- the swizzle
    // goes through the SOURCE address, the descriptor's range check zero-fills rows past N.
    constexpr int RPP = 1024 / ROWB, PIECES = TILEB / 1024, PPW = PIECES / NW;
    static_assert(PPW >= 1, "too many waves for this tile");
    auto make_rsrc = [&](const char *base, int bytes) {
        const uint64_t ba = (uint64_t)base;
        i32x4 r;
        r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)ba);
        r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(ba >> 32) & 0xffffu));
        r[2] = __builtin_amdgcn_readfirstlane(bytes);
        r[3] = 0x00020000;
        return r;
    };
    const i32x4 rs0 = make_rsrc(T0p, (N - 1) * (int)t0rs + ROWB), rs1 = make_rsrc(T1p, (N - 1) * (int)t1rs + ROWB);
    const unsigned lds_base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
    int so0[PPW], so1[PPW];
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
        const int row = RPP * (wave + pp * NW) + lane / CPR, slot = lane % CPR;
        const int chunk = slot ^ ((lds_off<D>(row, 0) - row * ROWB) >> 4);
        so0[pp] = row * (int)t0rs + chunk * 16;
        so1[pp] = row * (int)t1rs + chunk * 16;
    }
    // MODE 0: the row statistics L and D of the 64 swept query rows travel by LDS-DMA too (two 256-byte pieces issued by
    // wave 0; rows past N read as 0 and are masked like every other out-of-range row) -- staged through registers they
    // cost wave 0 a global-load wait of ~480 cycles per tile (in-kernel stamps)
    const i32x4 rsL = make_rsrc((const char *)Lc, N * 4), rsD = make_rsrc((const char *)Dp, N * 4);
    auto stage_load = [&](int t) {  // the target buffer t & 1 must be free when this is called
        const int buf = t & 1;
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
            dma16(rs0, lds_base + buf * TILEB + (wave + pp * NW) * 1024, so0[pp] + t * BS * (int)t0rs);
            dma16(rs1, lds_base + 2 * TILEB + buf * TILEB + (wave + pp * NW) * 1024, so1[pp] + t * BS * (int)t1rs);
        }
        if (MODE == 0 && wave == 0) {
            dma4(rsL, lds_base + LOFF + buf * BS * 4, (t * BS + lane) * 4);
            dma4(rsD, lds_base + LOFF + (2 * BS + buf * BS) * 4, (t * BS + lane) * 4);
        }
    };
    auto stage_write = [&](int) { dma_wait(); };  // this wave's pieces have landed; the barrier that follows publishes them

    int k_off[KS];  // row read: row kb*32 + i, chunk 2ks + h
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) k_off[ks] = lds_off<D>(i, 2 * ks + h);
    int v_off[2][DB];  // transposed read (fa2_mfma16.hip): u = rows +0..3 / +8..11 of the 16-row k-step
    {
        const int w = (lane >> 4) & 1, qq = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int db = 0; db < DB; ++db)
                v_off[u][db] = lds_off<D>(8 * u + 4 * h + qq, 4 * db + 2 * w + (pp >> 1)) + 8 * (pp & 1);
    }

    f32x16 acc0[DB];  // dQ^T (MODE 1), dK^T (MODE 0, waves 0..3) or dV^T (MODE 0, waves 4..7)
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[db][r] = 0.0f;
    const float c = a.c_log2e;
    float rsum = 0.0f;  // MODE 1: this lane's share of rowsum(P) of its query
    const bool is_causal = (a.causal & 1) != 0;
    if (is_causal) {
        if (MODE == 1) {
            const int last = (wg0 + BO - 1 < N - 1 ? wg0 + BO - 1 : N - 1);
            t_end = last / BS + 1;
        } else {
            t_begin = wg0 / BS;
        }
    }
    if (t_begin >= t_end) t_begin = t_end;  // nothing to sweep (cannot happen for N >= 1, kept for safety)

    if (t_begin < t_end) {
        stage_load(t_begin);
        stage_write(t_begin & 1);
    }
    __syncthreads();

    if constexpr (MODE == 0) {
        // ---- key-owner launch.  The dV wave and the dK wave of a key group (waves kg + 4 and kg: one SIMD) share P
        // through LDS instead of both computing S = Q K^T:
        //     dV wave, step b:   S_b (8 MFMA), P_b = exp2(c S_b - L) -> its fp32 accumulator image to slot[kg][b & 1],
        //                        dV^T += dO_b^T P_b (8 MFMA)
        //     dK wave, step b:   block b - 1:  dP (8 MFMA), P from slot[kg][(b - 1) & 1], dS = P (dP - D),
        //                        dK^T += Q^T dS (8 MFMA)
        // one barrier per step publishes the slot written in it (and, every second step, the next Q / dO tile): the dK
        // wave runs one 32-row block behind, whose tile is still resident -- tile t+1 is written into the buffer of tile
        // t-1 at the END of step 2t+1, after the dK wave's last use of tile t-1 in step 2t.  16 + 16 MFMAs per block.
        constexpr int POFF = LOFF + 4 * BS * 4;  // P slots: [4 key groups][2][4 KiB]
        const int kg = wave & 3;
        auto blk_skip = [&](int bidx) {  // causal: all 32 queries of the block precede all 32 keys of the wave
            return is_causal && bidx * 32 + 31 < own0;
        };
        auto blk_masked = [&](int bidx) { return (bidx * 32 + 32 > N) || (is_causal && bidx * 32 < own0 + 31); };
        for (int sidx = 2 * t_begin; sidx <= 2 * t_end; ++sidx) {
            const int t = sidx >> 1, kb = sidx & 1;
            // tile t+1 goes into the buffer of tile t-1, which the dK waves still read in the kb == 0 step
            if (kb == 1 && t + 1 < t_end) stage_load(t + 1);
            const int bidx = roleV ? sidx : sidx - 1;  // the block this wave works on in this step
            const bool work = roleV ? sidx < 2 * t_end : sidx - 1 >= 2 * t_begin;
            if (work && !blk_skip(bidx)) {
                const int cur = (bidx >> 1) & 1, kbb = bidx & 1, srow0 = bidx * 32;
                const int tb = cur * TILEB + kbb * 32 * ROWB;
                const int slot = POFF + (kg * 2 + (bidx & 1)) * 4096 + lane * 16;
                f32x16 x;
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = 0.0f;
                if (roleV) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const u32x4 qf = *(LDS_PTR(u32x4))(lds + tb + k_off[ks]);
                        x = M::mfma(__builtin_bit_cast(frag, qf), f0[ks], x);  // kernels.py:283 (without the log2e factor)
                    }
                    auto pbody = [&](auto masked_) __attribute__((always_inline)) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 lv = *(LDS_PTR(f32x4))(lds + LOFF + (cur * BS + kbb * 32 + 8 * g + 4 * h) * 4);
                            f32x4 pv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float pe = __builtin_amdgcn_exp2f(__builtin_fmaf(x[4 * g + j], c, -lv[j]));  // :285
                                if (decltype(masked_)::value) {
                                    const int qry = srow0 + 8 * g + 4 * h + j;
                                    if (qry >= N || (is_causal && orow > qry)) pe = 0.0f;
                                }
                                pv[j] = pe;
                                x[4 * g + j] = pe;
                            }
                            *(LDS_PTR(f32x4))(lds + slot + g * 1024) = pv;
                        }
                    };
                    if (blk_masked(bidx)) pbody(IC<1>{});  // one wave-uniform branch, two straight-line bodies
                    else pbody(IC<0>{});
                } else {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const u32x4 gf = *(LDS_PTR(u32x4))(lds + 2 * TILEB + tb + k_off[ks]);
                        x = M::mfma(__builtin_bit_cast(frag, gf), f1[ks], x);  // :289
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 dv = *(LDS_PTR(f32x4))(lds + LOFF + (2 * BS + cur * BS + kbb * 32 + 8 * g + 4 * h) * 4);
                        const f32x4 pv = *(LDS_PTR(f32x4))(lds + slot + g * 1024);
#pragma unroll
                        for (int j = 0; j < 4; ++j) x[4 * g + j] = pv[j] * (x[4 * g + j] - dv[j]);  // :291
                    }
                }
                const int tbase = (roleV ? 2 * TILEB : 0) + tb;  // dV^T = dO^T P reads the dO tile, dK^T = Q^T dS the Q tile
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    frag bf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bf[j] = (T)x[8 * ss + j];  // RTNE casts of :287 / :293
                    const int rowb = tbase + ss * 16 * ROWB;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[0][db]));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[1][db]));
                        const s16x8 tf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        acc0[db] = M::mfma(__builtin_bit_cast(frag, tf), bf, acc0[db]);
                    }
                }
            }
            if (kb == 1 && t + 1 < t_end) stage_write((t + 1) & 1);
            __syncthreads();
        }
    } else
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = t & 1;
        const bool more = t + 1 < t_end;
        if (more) stage_load(t + 1);
        // does this wave meet the tile at all?  (causal: MODE 1 keys of the tile <= last owned query of the wave;
        // MODE 0 queries of the tile >= first owned key of the wave)
        const bool active = !is_causal || (MODE == 1 ? t * BS <= own0 + 31 : t * BS + BS - 1 >= own0);
        if (active) {
            auto first = [&](int kb, f32x16 &x0, f32x16 &x1) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x0[r] = x1[r] = 0.0f;
                const int tb = cur * TILEB + kb * 32 * ROWB;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 t0f = *(LDS_PTR(u32x4))(lds + tb + k_off[ks]);
                    x0 = M::mfma(__builtin_bit_cast(frag, t0f), f0[ks], x0);  // kernels.py:283 (without the log2e factor)
                }
                // dP = dO V^T (8 MFMAs) with P = exp2(c S - L) of the finished S tile underneath: one fenced step per MFMA
                // carrying the fma + exp2 of 16/KS elements, one step behind so that the first exp2 does not wait for the
                // last S MFMA (this launch is the query owner: L is a per-lane constant).  Masks are applied in soft().
                constexpr int EP = 16 / KS;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const u32x4 t1f = *(LDS_PTR(u32x4))(lds + 2 * TILEB + tb + k_off[ks]);
                    x1 = M::mfma(__builtin_bit_cast(frag, t1f), f1[ks], x1);  // :289
                    if (ks >= 1) {
#pragma unroll
                        for (int e = 0; e < EP; ++e) {
                            const int r = (ks - 1) * EP + e;
                            x0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(x0[r], c, -Lown));  // :285
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int e = 0; e < EP; ++e) {
                    const int r = (KS - 1) * EP + e;
                    x0[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(x0[r], c, -Lown));
                }
            };
            auto soft = [&](int kb, f32x16 &x0, f32x16 &x1) __attribute__((always_inline)) {
                const int srow0 = t * BS + kb * 32;  // first swept row of this 32-row block
                // swept row of register r: srow0 + (r & 3) + 8 (r >> 2) + 4 h
                const bool need_mask = (srow0 + 32 > N) || (is_causal && (MODE == 1 ? srow0 + 31 > own0 : srow0 < own0 + 31));
                auto body = [&](auto masked_) __attribute__((always_inline)) {
                    constexpr bool MASKED = decltype(masked_)::value;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 dv = {Down, Down, Down, Down};  // this path is the query owner's: D is per lane
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int r = 4 * g + j;
                            float p = x0[r];  // exp2(c S - L), computed under the dP MFMAs in first()
                            if (MASKED) {
                                const int srow = srow0 + 8 * g + 4 * h + j;
                                const int key = MODE == 1 ? srow : orow, qry = MODE == 1 ? orow : srow;
                                if (srow >= N || (is_causal && key > qry)) p = 0.0f;
                            }
                            if (MODE == 1) rsum += p;
                            x1[r] = roleV ? p : p * (x1[r] - dv[j]);  // :291 (the scale factor is applied once, at the end)
                        }
                    }
                };
                if (need_mask) body(IC<1>{});
                else body(IC<0>{});
            };
            // second products: k-step ss = swept rows 16ss .. 16ss+15 of the block; registers 8ss..8ss+7 of X are the B
            // fragment (element j <-> row 16ss + 8(j>>2) + 4h + (j&3)), the A fragment is the transposed read
            auto second = [&](int kb, f32x16 &x0, f32x16 &x1) __attribute__((always_inline)) {
                const int tb = (roleV ? 2 * TILEB : 0) + cur * TILEB + kb * 32 * ROWB;  // dV^T = dO^T P reads the dO tile
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    frag bf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bf[j] = (T)x1[8 * ss + j];  // RTNE casts of :287 / :293 / :317
                    const int rowb = tb + ss * 16 * ROWB;
#pragma unroll
                    for (int db = 0; db < DB; ++db) {
                        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[0][db]));
                        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_PTR(s16x4))(lds + rowb + v_off[1][db]));
                        const s16x8 tf = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        acc0[db] = M::mfma(__builtin_bit_cast(frag, tf), bf, acc0[db]);
                    }
                }
            };
            auto skip = [&](int kb) {
                const int srow0 = t * BS + kb * 32;
                return is_causal && (MODE == 1 ? srow0 > own0 + 31 : srow0 + 31 < own0);
            };
            const bool do0 = !skip(0), do1 = !skip(1);
            // (issuing both blocks' first products before any softmax arithmetic was measured SLOWER in the key-owner
            // mode: two more score tiles push it past 256 arch registers and every use then pays v_accvgpr_read)
            f32x16 xa0, xa1;
            if (FA2_BWD_PIPE && do0 && do1) {
                // both blocks' first products go out before any softmax arithmetic: the exp2 / dS arithmetic of block 0
                // runs under the MFMAs of block 1, that of block 1 under the second products of block 0
                f32x16 xb0, xb1;
                first(0, xa0, xa1);
                first(1, xb0, xb1);
                soft(0, xa0, xa1);
                second(0, xa0, xa1);
                soft(1, xb0, xb1);
                second(1, xb0, xb1);
            } else {
#pragma nounroll
                for (int kb = 0; kb < 2; ++kb) {
                    if (kb == 0 ? !do0 : !do1) continue;
                    first(kb, xa0, xa1);
                    soft(kb, xa0, xa1);
                    second(kb, xa0, xa1);
                }
            }
        }
        if (more) stage_write(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane (i, h) owns output row `orow`, columns 32db + 8g + 4h .. +3
    float tot = 1.0f;
    if (MODE == 1) {
        const unsigned u = __builtin_bit_cast(unsigned, rsum);
        const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);  // lanes i and i + 32 share a query
        const unsigned s0 = sw[0], s1 = sw[1];
        tot = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
    }
    if (orow < N) {
        auto store_rows = [&](char *base, const int64_t *st, f32x16 (&acc)[DB], float mul) {
            char *op = base + b * st[0] + hh * st[1] + (int64_t)orow * st[2] + h * 8;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef __attribute__((ext_vector_type(4))) T Tx4;
                    Tx4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (T)(acc[db][4 * g + j] * mul);
                    *(u32x2 *)(op + db * 64 + g * 16) = __builtin_bit_cast(u32x2, v);
                }
        };
        if (MODE == 1) {
            store_rows(a.dQ, a.dqs, acc0, a.scale / tot);
            if (h == 0) Lc[orow] = Lown + __builtin_amdgcn_logf(tot);
        } else if (!roleV) {
            store_rows(a.dK, a.dks, acc0, a.scale);
        } else {
            store_rows(a.dV, a.dvs, acc0, 1.0f);
        }
    }
}

template <typename T, int D> int launch_d(const Fa2BwdProblem &p, const BArgs &a) {
    const long long rows = (long long)p.B * p.H * p.N, lanes = rows * (D / 8);
    hipLaunchKernelGGL((bwd_D_kernel<T, D>), dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, p.stream, a);
    const long long nblk = (long long)((p.N + 127) / 128) * p.B * p.H;
    if (nblk > 0x7fffffffLL) {
        fa2_set_error("backward mfma16: grid too large");
        return FA2_ERR_BAD_ARG;
    }
    constexpr size_t smem0 = 4 * 64 * D * 2 + 4 * 64 * 4 + 4 * 2 * 4096, smem1 = 4 * 64 * D * 2;
    // the query-owner launch goes first: it leaves the fp32 row statistic the key-owner launch reads
    if (fa2_env_int("FA2_BWD_DQ_WAVES", 4) == 8) {
        const long long nblk8 = (long long)((p.N + 255) / 256) * p.B * p.H;
        hipLaunchKernelGGL((bwd_mfma16_kernel<T, D, 1, 8>), dim3((unsigned)nblk8), dim3(512), smem1, p.stream, a);
    } else {
        hipLaunchKernelGGL((bwd_mfma16_kernel<T, D, 1, 4>), dim3((unsigned)nblk), dim3(256), smem1, p.stream, a);
    }
    static Fa2DeviceLatch attr_set;  // > 64 KiB of dynamic LDS needs the attribute; once per instantiation
    if (attr_set.need()) {
        (void)hipFuncSetAttribute((const void *)bwd_mfma16_kernel<T, D, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem0);
        attr_set.mark();
    }
    hipLaunchKernelGGL((bwd_mfma16_kernel<T, D, 0>), dim3((unsigned)nblk), dim3(512), smem0, p.stream, a);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fa2_set_error("backward mfma16 launch failed: %s", hipGetErrorString(e));
        return FA2_ERR_LAUNCH;
    }
    return FA2_OK;
}

bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

bool fa2_bwd_mfma16_supports(const Fa2BwdProblem &p) {
    if (p.dtype != FA2_DTYPE_F16 && p.dtype != FA2_DTYPE_BF16) return false;
    if (p.d != 64 && p.d != 128) return false;
    if (!(p.scale > 0.0f) || !(p.scale < INFINITY)) return false;
    const int64_t *all[8] = {p.qs, p.ks, p.vs, p.os, p.dos, p.dqs, p.dks, p.dvs};
    for (int t = 0; t < 8; ++t) {
        if (all[t][3] != 1) return false;
        for (int k = 0; k < 3; ++k)
            if (all[t][k] & 7) return false;  // 16-byte vector accesses of whole rows
    }
    const void *ptrs[8] = {p.Q, p.K, p.V, p.O, p.dO, p.dQ, p.dK, p.dV};
    for (int t = 0; t < 8; ++t)
        if (!aligned16(ptrs[t])) return false;
    if (p.N > (1 << 24)) return false;
    // 32-bit buffer offsets of the LDS-DMA staging: (N + 64) rows of every swept tensor below 2 GiB
    for (int t = 0; t < 5; ++t)
        if ((int64_t)(p.N + 64) * all[t][2] * 2 >= (1LL << 31)) return false;
    return true;
}

int fa2_bwd_launch_mfma16(const Fa2BwdProblem &p) {
    if (!fa2_bwd_mfma16_supports(p)) {
        fa2_set_error("backward mfma16: needs f16/bf16, d in {64,128}, unit d-stride, 16-byte aligned rows, scale > 0");
        return FA2_ERR_UNSUPPORTED;
    }
    BArgs a;
    a.Q = (const char *)p.Q; a.K = (const char *)p.K; a.V = (const char *)p.V; a.O = (const char *)p.O;
    a.dO = (const char *)p.dO; a.L = (const char *)p.L;
    a.dQ = (char *)p.dQ; a.dK = (char *)p.dK; a.dV = (char *)p.dV; a.D = (float *)p.D;
    for (int k = 0; k < 3; ++k) {
        a.qs[k] = p.qs[k] * 2; a.ks[k] = p.ks[k] * 2; a.vs[k] = p.vs[k] * 2; a.os[k] = p.os[k] * 2;
        a.dos[k] = p.dos[k] * 2; a.dqs[k] = p.dqs[k] * 2; a.dks[k] = p.dks[k] * 2; a.dvs[k] = p.dvs[k] * 2;
    }
    a.ls[0] = p.ls[0]; a.ls[1] = p.ls[1];
    a.B = p.B; a.H = p.H; a.N = p.N;
    a.causal = (p.causal ? 1 : 0) | (fa2_env_int("FA2_BWD_PRIO", 1) ? 2 : 0);  // bit 1: static priority for the dV waves
    a.c_log2e = (float)((double)p.scale * FA2_LOG2E);
    a.scale = p.scale;
    if (p.dtype == FA2_DTYPE_BF16) return p.d == 128 ? launch_d<__bf16, 128>(p, a) : launch_d<__bf16, 64>(p, a);
    return p.d == 128 ? launch_d<_Float16, 128>(p, a) : launch_d<_Float16, 64>(p, a);
}
