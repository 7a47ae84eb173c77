#!/usr/bin/env python3
"""Generator of the gfx950 assembly kernel `fa2_fwd_a64_<dtype>_<c|n>` -- FA-2 forward, d = 128, f16 / bf16.

Reference arithmetic: /root/reference/src/flash_attention_kernels.py:84-108 (exp2-domain online softmax in fp32, P rounded
RTNE to the I/O dtype before P.V, O /= l once at the end, L = m + log2 l), as in fa2_mfma16h.hip.

Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"):
  * workgroup = 4 waves = one 256-row Q block; a wave owns two 32-row query blocks (qb = 0, 1) and the WHOLE 512-entry
    register file: O^T in a[0:127], Q in a[128:191], the current K tile in a[192:255]; two score buffers, the V^T
    fragments and the softmax state in the arch VGPRs;
  * swapped products: S^T[key][query] = K.Q^T, O^T[d][query] += V^T.P^T on v_mfma_f32_32x32x16 -- a lane owns one query
    row per query block, P never leaves the registers (the S accumulator, packed in place, is the B operand of P.V);
  * 64-key K/V tiles arrive by LDS-DMA (buffer_load ... lds) into two K and two V buffers; LDS image = 8-row x 32-column
    subtiles of 512 B with the 16-byte slots XOR-swizzled (T10 image (a)): row reads (ds_read_b128) and transposed reads
    (ds_read_b64_tr_b16) are conflict-free and need two per-lane base registers each, everything else is an immediate;
  * per tile two phases of 32 MFMAs:   A(t) = QK^T(t+1) || finish-softmax(t) (exp2, row sums, cvt) || V(t) tr-reads
                                       B(t) = P.V(t)    || start-softmax(t+1) (row max, decision, s*c - m) || K(t+2) reads
                                                        || LDS-DMA of V(t+2), K(t+3)
    with ONE barrier per tile (between A and B); the loop is unrolled over the buffer parity;
  * persistent grid: a workgroup walks its jobs (non-causal: one Q block; causal: the pair (nq-1-u, u)).

The instruction stream is built as isa.Inst objects: printed to a .s file for the assembler, and executed by emu.py in the
CPU test-suite (tests/test_asm_emu.py) against the oracle.
"""
from __future__ import annotations

import argparse
import sys

from .isa import A, EXEC, I, Inst, Label, M0, Reg, S, V, VCC, comment, label, waitcnt

# ------------------------------------------------------------------------------------------------- register map
# arch VGPRs
SBUF = (0, 64)            # two score buffers of 64 registers: group g = 2*qb + kb at +16 g
VF = 128                  # V^T fragments: (kstep, db) at VF + 4 * (4 * kstep + db)
V_KRE, V_KRO = 192, 193   # K row-read lane bases (even / odd k-step)
V_VR0, V_VR1 = 194, 195   # V transposed-read lane bases (u = 0 / 1)
V_DKO, V_DVO = 196, 197   # LDS-DMA per-lane source offsets (K / V row stride)
V_MC = (198, 199)         # running row maximum in the exp2 domain (c * max), per query block
V_RS = ((200, 201), (202, 203))  # row-sum accumulators [qb][2]
V_MX = ((204, 205), (206, 207))  # row-max chains [qb][kb]
V_CO = (208, 209)         # rescale coefficient per query block
V_T = tuple(range(210, 220))     # temporaries
V_QOFF = 220              # Q load lane offset (row stride qs_n)
V_LANE = 221
V_EW = 222                # epilogue LDS write base (row i, +8h)
V_ESW = 223               # epilogue swizzle term swz(i)
V_ER = 224                # epilogue LDS read base
V_EO = 225                # epilogue global store lane offset (os_n)
V_L2 = 226                # L store lane offset
V_TRI = 232               # causal: 16 registers, 0 / -inf triangle of a diagonal 32 x 32 block (lane = query)
V_NINF = 248              # causal: one register of -inf (C operand blocks are built on the fly)

# AGPRs
def A_O(qb, db):
    return A((qb * 4 + db) * 16, 16)


def A_Q(qb, ks):
    return A(128 + (qb * 8 + ks) * 4, 4)


def A_K(kb, ks):
    return A(192 + (kb * 8 + ks) * 4, 4)


def V_F(kstep, db):
    return V(VF + 4 * (4 * kstep + db), 4)


# SGPRs
S_KARG = S(0, 2)
S_WGID = S(2)
S_Q, S_K, S_V, S_O, S_L = S(4, 2), S(6, 2), S(8, 2), S(10, 2), S(12, 2)
S_QSB, S_QSH, S_KSB, S_KSH, S_VSB, S_VSH, S_OSB, S_OSH, S_LSB, S_LSH = (S(14 + 2 * k, 2) for k in range(10))
S_QSN, S_KSN, S_VSN, S_OSN = S(34), S(35), S(36), S(37)
S_N, S_H, S_NQ, S_TOTAL = S(38), S(39), S(40), S(41)
S_C, S_THR, S_NUNIT, S_G = S(42), S(43), S(44), S(45)
S_NBH, S_NWG = S(46), S(47)
S_KRS, S_VRS, S_QRS, S_ORS, S_LRS = S(48, 4), S(52, 4), S(56, 4), S(60, 4), S(64, 4)
S_JOB, S_WAVE = S(68), S(69)
S_KDMA, S_VDMA = S(70), S(71)
S_K32, S_K32P, S_V32, S_V32P = S(72), S(73), S(74), S(75)
S_K64, S_V64 = S(76), S(77)
S_LDSW = S(78)
S_LOOP, S_FLAG = S(79), S(80)
S_QI, S_BH, S_B, S_HH, S_UNIT, S_PASS = S(81), S(82), S(83), S(84), S(85), S(86)
S_T = tuple(S(88 + k) for k in range(8))  # temporaries s88..s95 (S_T[0] even: usable as a 64-bit pair)
S_NT = S(87)       # tiles of this job
S_QROW = (S(96), S(97))  # first row of the wave's query block qb
S_DBG = S(98, 2)
S_DIAG = S(100)    # causal: first diagonal tile of this job (tile index), per wave relation computed on the fly
S_KMAX = S(101)    # K DMA offset of the job's last tile (the look-ahead is clamped to it)

# LDS map (bytes)
KB = (0, 16384)
VB = (32768, 49152)
EPI = 65536        # + 16384 * wave: the wave's 64 x 256-byte output slice
LDS_TOTAL = 131072

KARG_SIZE = 184


class Gen:
    def __init__(self, dtype="bf16", causal=False, name=None, nexp_b=8, dbg=False):
        assert dtype in ("bf16", "f16")
        self.dtype = dtype
        self.causal = causal
        self.name = name or f"fa2_fwd_a64_{dtype}_{'c' if causal else 'n'}"
        self.prog: list[Inst] = []
        self.uid = 0
        self.nexp_b = nexp_b
        self.mfma = "v_mfma_f32_32x32x16_" + dtype
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.ool: list[list[Inst]] = []  # out-of-line blocks (rare paths), appended after the main body
        self.dbg = dbg

    # ------------------------------------------------------------------ small helpers
    def e(self, *insts):
        for x in insts:
            if isinstance(x, (list, tuple)):
                self.e(*x)
            else:
                self.prog.append(x)

    def lab(self, stem):
        self.uid += 1
        return f".L{self.name}_{stem}_{self.uid}"

    def udiv(self, q: Reg, r: Reg | None, n: Reg, d: Reg):
        """q = n / d, r = n % d for wave-uniform 32-bit values < 2^22 (float reciprocal + one correction each way)"""
        t0, t1 = V(V_T[0]), V(V_T[1])
        st, sr = S_T[6], S_T[7]
        self.e(I("v_cvt_f32_u32", t0, n), I("v_cvt_f32_u32", t1, d), I("s_nop", 0), I("v_rcp_f32", t1, t1), I("s_nop", 1),
               I("v_mul_f32", t0, t0, t1), I("v_cvt_u32_f32", t0, t0), I("s_nop", 1), I("v_readfirstlane_b32", q, t0), I("s_nop", 4),
               I("s_mul_i32", st, q, d), I("s_sub_i32", sr, n, st),
               # r < 0 -> q--, r += d
               I("s_cmp_lt_i32", sr, 0), I("s_cselect_b32", st, 1, 0), I("s_sub_u32", q, q, st),
               I("s_cmp_lt_i32", sr, 0), I("s_cselect_b32", st, d, 0), I("s_add_i32", sr, sr, st),
               # r >= d -> q++, r -= d
               I("s_cmp_ge_i32", sr, d), I("s_cselect_b32", st, 1, 0), I("s_add_u32", q, q, st),
               I("s_cmp_ge_i32", sr, d), I("s_cselect_b32", st, d, 0), I("s_sub_i32", sr, sr, st))
        if r is not None:
            self.e(I("s_mov_b32", r, sr))

    def mad64(self, dst: Reg, idx: Reg, stride: Reg):
        """dst(64) += idx(32, unsigned) * stride(64)"""
        lo, hi = S_T[6], S_T[7]
        self.e(I("s_mul_i32", lo, idx, stride.sub(0)), I("s_mul_hi_u32", hi, idx, stride.sub(0)),
               I("s_add_u32", dst.sub(0), dst.sub(0), lo), I("s_addc_u32", dst.sub(1), dst.sub(1), hi),
               I("s_mul_i32", lo, idx, stride.sub(1)), I("s_add_u32", dst.sub(1), dst.sub(1), lo))

    def make_rsrc(self, rs: Reg, base: Reg):
        self.e(I("s_mov_b32", rs.sub(0), base.sub(0)), I("s_and_b32", rs.sub(1), base.sub(1), 0xFFFF),
               I("s_mov_b32", rs.sub(2), 0x7FFFFFF0), I("s_mov_b32", rs.sub(3), 0x00020000))

    # ------------------------------------------------------------------ kernel prologue: arguments, lane constants
    def k_setup(self):
        e = self.e
        e(comment("kernel arguments"),
          I("s_load_dwordx8", S(4, 8), S_KARG, 0), I("s_load_dwordx4", S(12, 4), S_KARG, 32),
          I("s_load_dwordx16", S(16, 16), S_KARG, 48), I("s_load_dwordx4", S(32, 4), S_KARG, 112),
          I("s_load_dwordx8", S(36, 8), S_KARG, 128), I("s_load_dwordx4", S(44, 4), S_KARG, 160),
          I("s_load_dwordx2", S_DBG, S_KARG, 176))
        lane, t0, t1, t2, t3 = V(V_LANE), V(V_T[0]), V(V_T[1]), V(V_T[2]), V(V_T[3])
        e(I("v_and_b32", lane, 63, V(0)), I("v_lshrrev_b32", t0, 6, V(0)), I("s_nop", 1), I("v_readfirstlane_b32", S_WAVE, t0), I("s_nop", 4),
          I("s_lshl_b32", S_LDSW, S_WAVE, 11))
        # i = lane & 31, h = lane >> 5
        # ---- K row-read bases: 2048 (i >> 3) + 64 (i & 7) + 16 ((2 e + h) ^ ((i >> 2) & 3))
        e(comment("K row-read lane bases"),
          I("v_and_b32", t0, 31, lane),                    # i
          I("v_lshrrev_b32", t1, 3, t0), I("v_lshlrev_b32", t1, 11, t1),   # 2048 (i >> 3)
          I("v_and_b32", t2, 7, t0), I("v_lshl_add_u32", t1, t2, 6, t1),   # + 64 (i & 7)
          I("v_bfe_u32", t2, t0, 2, 2),                    # g = (i >> 2) & 3
          I("v_lshrrev_b32", t3, 5, lane),                 # h
          I("v_xor_b32", t2, t2, t3),                      # h ^ g          (even k-step: chunk slot (0 + h) ^ g)
          I("v_lshl_add_u32", V(V_KRE), t2, 4, t1),
          I("v_xor_b32", t2, 2, t2),                       # (2 + h) ^ g
          I("v_lshl_add_u32", V(V_KRO), t2, 4, t1))
        # ---- V transposed-read bases: 64 (4 h + q) + 16 ((2 w + (p >> 1)) ^ ((2 u + h) & 3)) + 8 (p & 1)
        #      w = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3
        e(comment("V transposed-read lane bases"),
          I("v_bfe_u32", t0, lane, 2, 2),                  # q
          I("v_lshrrev_b32", t3, 5, lane),                 # h
          I("v_lshl_add_u32", t0, t3, 2, t0),              # 4 h + q
          I("v_lshlrev_b32", t0, 6, t0),                   # 64 (4 h + q)
          I("v_and_b32", t1, 1, lane), I("v_lshl_add_u32", t0, t1, 3, t0),  # + 8 (p & 1)
          I("v_bfe_u32", t1, lane, 4, 1), I("v_lshlrev_b32", t1, 1, t1),    # 2 w
          I("v_bfe_u32", t2, lane, 1, 1), I("v_or_b32", t1, t1, t2),        # 2 w + (p >> 1)
          I("v_xor_b32", t2, t1, t3),                      # u = 0: ^ h
          I("v_lshl_add_u32", V(V_VR0), t2, 4, t0),
          I("v_xor_b32", t2, 2, t2),                       # u = 1: ^ (2 + h)
          I("v_lshl_add_u32", V(V_VR1), t2, 4, t0))
        # ---- LDS-DMA lane source offsets: l' = lane & 31: row_in = l' >> 2, slot = l' & 3, sub = lane >> 5
        #      chunk = 4 sub + (slot ^ (2 (wave & 1) + (l' >> 4)));  offset = row_in * stride + 16 chunk
        e(comment("LDS-DMA per-lane source offsets"),
          I("v_and_b32", t0, 3, lane),                     # slot
          I("v_bfe_u32", t1, lane, 4, 1),                  # l' >> 4
          I("s_and_b32", S_T[0], S_WAVE, 1), I("s_lshl_b32", S_T[0], S_T[0], 1),
          I("v_or_b32", t1, S_T[0], t1),
          I("v_xor_b32", t0, t0, t1),
          I("v_lshrrev_b32", t1, 5, lane), I("v_lshl_or_b32", t0, t1, 2, t0),   # 4 sub + ...
          I("v_lshlrev_b32", t0, 4, t0),                   # 16 chunk
          I("v_bfe_u32", t1, lane, 2, 3))                  # row_in
        e(waitcnt(lgkmcnt=0, comment="kernel arguments are in"))
        e(I("v_mul_lo_u32", t2, t1, S_KSN), I("v_add_u32", V(V_DKO), t2, t0),
          I("v_mul_lo_u32", t2, t1, S_VSN), I("v_add_u32", V(V_DVO), t2, t0))
        # ---- Q load lane offset: i * qs_n + 16 h
        e(comment("Q load / epilogue lane constants"),
          I("v_and_b32", t0, 31, lane), I("v_lshrrev_b32", t3, 5, lane),
          I("v_mul_lo_u32", t1, t0, S_QSN), I("v_lshl_add_u32", V(V_QOFF), t3, 4, t1))
        # ---- epilogue: write base EPI + 16384 wave + 256 i + 8 h; swizzle term (((i & 3) << 2) | ((i >> 2) & 3)) << 4
        e(I("s_lshl_b32", S_T[0], S_WAVE, 14), I("s_add_u32", S_T[0], S_T[0], EPI),
          I("v_lshlrev_b32", t1, 8, t0), I("v_lshl_add_u32", t1, t3, 3, t1), I("v_add_u32", V(V_EW), S_T[0], t1),
          I("v_and_b32", t1, 3, t0), I("v_lshlrev_b32", t1, 2, t1), I("v_bfe_u32", t2, t0, 2, 2), I("v_or_b32", t1, t1, t2),
          I("v_mov_b32", V(V_ESW), t1))
        # read-back: a = lane >> 4, ec = lane & 15: base + 256 a + ((ec ^ (a << 2)) << 4); store offset a * os_n + 16 ec
        e(I("v_lshrrev_b32", t0, 4, lane), I("v_and_b32", t1, 15, lane),
          I("v_lshlrev_b32", t2, 2, t0), I("v_xor_b32", t2, t2, t1), I("v_lshlrev_b32", t2, 4, t2),
          I("v_lshl_add_u32", t2, t0, 8, t2), I("v_add_u32", V(V_ER), S_T[0], t2),
          I("v_mul_lo_u32", t2, t0, S_OSN), I("v_lshl_add_u32", V(V_EO), t1, 4, t2),
          I("v_and_b32", t0, 31, lane), I("v_lshlrev_b32", V(V_L2), 1, t0))
        # ---- scalar constants
        e(I("s_lshl_b32", S_K32, S_KSN, 5), I("s_add_u32", S_K32P, S_K32, 128),
          I("s_lshl_b32", S_V32, S_VSN, 5), I("s_add_u32", S_V32P, S_V32, 128),
          I("s_lshl_b32", S_K64, S_KSN, 6), I("s_lshl_b32", S_V64, S_VSN, 6),
          I("s_mov_b32", S_FLAG, 0), I("s_mov_b32", S_PASS, 0),
          I("s_mov_b32", S_JOB, S_WGID))
        if self.causal:
            # triangle block: register r <-> key (r & 3) + 8 (r >> 2) + 4 h of a 32-key block, lane <-> query i: -inf where key > i
            e(comment("causal: diagonal-block mask (0 / -inf) as an MFMA C operand"),
              I("v_and_b32", t0, 31, lane), I("v_lshrrev_b32", t3, 5, lane), I("v_lshlrev_b32", t3, 2, t3),
              I("v_sub_u32", t0, t0, t3),   # i - 4 h
              I("v_mov_b32", V(V_NINF), float("-inf")))
            for r in range(16):
                key = (r & 3) + 8 * (r >> 2)
                e(I("v_cmp_lt_i32", VCC, t0, key), I("v_cndmask_b32", V(V_TRI + r), 0, V(V_NINF), VCC))

    # ------------------------------------------------------------------ job decode -> S_BH, S_UNIT (then S_QI by the caller)
    def k_decode(self):
        e = self.e
        l_else, l_done = self.lab("dec_else"), self.lab("dec_done")
        t = S_T
        e(comment("job index -> (b, h) and work unit"),
          I("s_and_b32", t[0], S_NBH, 7), I("s_cmp_lg_u32", t[0], 0), I("s_cbranch_scc1", Label(l_else)))
        # slot = id >> 3; GN = G * nunit; batch = slot / GN; r = slot % GN; bh = (batch * G + r % G) * 8 + (id & 7); unit = r / G
        e(I("s_lshr_b32", t[0], S_JOB, 3), I("s_mul_i32", t[1], S_G, S_NUNIT))
        self.udiv(t[2], t[3], t[0], t[1])       # batch, r
        self.udiv(S_UNIT, t[4], t[3], S_G)      # unit = r / G, r % G
        e(I("s_mul_i32", t[2], t[2], S_G), I("s_add_u32", t[2], t[2], t[4]), I("s_lshl_b32", t[2], t[2], 3),
          I("s_and_b32", t[0], S_JOB, 7), I("s_add_u32", S_BH, t[2], t[0]), I("s_branch", Label(l_done)))
        e(label(l_else))
        self.udiv(S_BH, S_UNIT, S_JOB, S_NUNIT)
        e(label(l_done))
        self.udiv(S_B, S_HH, S_BH, S_H)

    # ------------------------------------------------------------------ per-job scalars, descriptors, first loads
    def dma_piece(self, rsrc, vlane, soff_base, piece, lds_const, strides32):
        """one 1-KiB LDS-DMA piece.  piece j in 0..3: rows R = wave, wave + 4; halves 0 / 1"""
        s32, s32p = strides32
        out = []
        if piece == 0:
            so = soff_base
        else:
            so = S_T[5]
            out.append(I("s_add_u32", so, soff_base, (128, s32, s32p)[piece - 1]))
        out.append(I("s_add_u32", M0, S_LDSW, lds_const + (0, 1024, 8192, 9216)[piece]))
        out.append(I("s_nop", 0))
        out.append(I("buffer_load_dwordx4", vlane, rsrc, so, offen=1, lds=1))
        return out

    def dma_tile(self, which, buf):
        out = []
        for j in range(4):
            if which == "k":
                out += self.dma_piece(S_KRS, V(V_DKO), S_KDMA, j, KB[buf], (S_K32, S_K32P))
            else:
                out += self.dma_piece(S_VRS, V(V_DVO), S_VDMA, j, VB[buf], (S_V32, S_V32P))
        return out

    def k_job_setup(self):
        """S_BH/S_B/S_HH/S_QI known: descriptors, query rows, tile count"""
        e = self.e
        e(comment("per-job bases and descriptors"))
        for base, sb, sh, rs in ((S_K, S_KSB, S_KSH, S_KRS), (S_V, S_VSB, S_VSH, S_VRS), (S_Q, S_QSB, S_QSH, S_QRS),
                                 (S_O, S_OSB, S_OSH, S_ORS), (S_L, S_LSB, S_LSH, S_LRS)):
            tmp = S(S_T[0].idx, 2)
            e(I("s_mov_b64", tmp, base))
            self.mad64(tmp, S_B, sb)
            self.mad64(tmp, S_HH, sh)
            self.make_rsrc(rs, tmp)
        # query rows of this wave: qrow[qb] = 256 qi + 64 wave + 32 qb
        e(I("s_lshl_b32", S_T[0], S_QI, 8), I("s_lshl_b32", S_T[1], S_WAVE, 6), I("s_add_u32", S_QROW[0], S_T[0], S_T[1]),
          I("s_add_u32", S_QROW[1], S_QROW[0], 32))
        if self.causal:
            e(I("s_add_u32", S_T[0], S_QI, 1), I("s_lshl_b32", S_NT, S_T[0], 2))     # tiles = 4 (qi + 1)
        else:
            e(I("s_lshr_b32", S_NT, S_N, 6))
        # wave's DMA row base: 8 * wave * stride
        e(I("s_lshl_b32", S_T[0], S_WAVE, 3), I("s_mul_i32", S_KDMA, S_T[0], S_KSN), I("s_mul_i32", S_VDMA, S_T[0], S_VSN),
          I("s_sub_u32", S_T[0], S_NT, 1), I("s_mul_i32", S_T[0], S_T[0], S_K64), I("s_add_u32", S_KMAX, S_T[0], S_KDMA))

    def k_job_first_loads(self):
        e = self.e
        e(comment("first loads of the job: K(0), V(0), K(1) by LDS-DMA, Q rows into a[128:191]"))
        e(self.dma_tile("k", 0), self.dma_tile("v", 0))
        e(I("s_add_u32", S_KDMA, S_KDMA, S_K64), I("s_add_u32", S_VDMA, S_VDMA, S_V64))
        e(self.dma_tile("k", 1))
        e(I("s_add_u32", S_KDMA, S_KDMA, S_K64))
        for qb in range(2):
            e(I("s_mul_i32", S_T[0], S_QROW[qb], S_QSN))
            for ks in range(8):
                e(I("buffer_load_dwordx4", A_Q(qb, ks), V(V_QOFF), S_QRS, S_T[0], offen=1, offset=32 * ks))

    # ------------------------------------------------------------------ the two phases
    def interleave(self, mfmas, fillers):
        """fillers: list of (position, [insts]); a filler at position x goes after MFMA floor(x) (x < 0: before MFMA 0)"""
        fillers = sorted(enumerate(fillers), key=lambda kv: (kv[1][0], kv[0]))
        out, fi = [], 0
        n = len(fillers)
        while fi < n and fillers[fi][1][0] < 0:
            out += fillers[fi][1][1]
            fi += 1
        for k, m in enumerate(mfmas):
            out.append(m)
            while fi < n and fillers[fi][1][0] < k + 1:
                out += fillers[fi][1][1]
                fi += 1
        while fi < n:
            out += fillers[fi][1][1]
            fi += 1
        return out

    def qk_mfmas(self, Y, cinit=None):
        """S^T(next) chains g = 2 qb + kb into score buffer Y.  cinit[g]: None -> C = 0, Reg -> C operand of the first MFMA"""
        out = []
        for g in range(4):
            qb, kb = g >> 1, g & 1
            d = V(Y + 16 * g, 16)
            for ks in range(8):
                c = d if ks else (cinit[g] if cinit and cinit[g] is not None else 0)
                out.append(I(self.mfma, d, A_K(kb, ks), A_Q(qb, ks), c, tag=f"qk g{g} ks{ks}"))
        return out

    def pv_mfmas(self, X, first):
        """O^T[qb][db] += V^T(kstep, db) . P^T(qb, kstep); P(qb, kstep = 2 kb + s) = X + 16 (2 qb + kb) + 4 s"""
        out = []
        for kstep in range(4):
            kb, s = kstep >> 1, kstep & 1
            for db in range(4):
                for qb in range(2):
                    p = V(X + 16 * (2 * qb + kb) + 4 * s, 4)
                    c = 0 if (first and kstep == 0) else A_O(qb, db)
                    out.append(I(self.mfma, A_O(qb, db), V_F(kstep, db), p, c, tag=f"pv ks{kstep} db{db} qb{qb}"))
        return out

    def v_reads(self, buf):
        """32 transposed reads of V tile in VB[buf]: fragment (kstep, db) <- u = 0, 1"""
        out = []
        for kstep in range(4):
            for db in range(4):
                f = V_F(kstep, db)
                for u in range(2):
                    imm = VB[buf] + 2048 * (2 * kstep + u) + 512 * db
                    out.append(I("ds_read_b64_tr_b16", f.sub(2 * u, 2), V(V_VR1 if u else V_VR0), offset=imm, tag=f"vread ks{kstep} db{db}"))
        return out

    def k_reads(self, buf):
        out = []
        for kb in range(2):
            for ks in range(8):
                imm = KB[buf] + 8192 * kb + 512 * (ks >> 1)
                out.append(I("ds_read_b128", A_K(kb, ks), V(V_KRO if ks & 1 else V_KRE), offset=imm, tag=f"kread kb{kb} ks{ks}"))
        return out

    def finish_ops(self, X, skip_exp=0):
        """finish-softmax of the tile in score buffer X: exp2 (elements >= skip_exp; the first skip_exp were done in the
        previous phase B), row sums, in-place pack.  Returns [(element index, [exp], [add], [cvt or None])]"""
        ops = []
        for g in range(4):
            qb = g >> 1
            for r in range(16):
                el = 16 * g + r
                x = V(X + el)
                ex = [] if el < skip_exp else [I("v_exp_f32", x, x, tag=f"exp {el}")]
                ad = [I("v_add_f32", V(V_RS[qb][r & 1]), V(V_RS[qb][r & 1]), x, tag=f"sum {el}")]
                cv = []
                if r & 1:
                    cv = [I(self.cvt, V(X + 16 * g + (r >> 1)), V(X + el - 1), x, tag=f"cvt {el}")]
                ops.append((el, ex, ad, cv))
        return ops

    def phase_a(self, p, with_qk=True, with_finish=True, cinit=None):
        """A(t), parity p = t & 1: QK^T(t+1) -> S[1-p]  ||  finish(S[p])  ||  V(t) reads from VB[p]"""
        X, Y = SBUF[p], SBUF[1 - p]
        mf = self.qk_mfmas(Y, cinit) if with_qk else []
        fill = []
        if with_finish:
            vr = self.v_reads(p)
            for k, ins in enumerate(vr):           # 2 reads per gap over the first 16 gaps
                fill.append((k * 0.5, [ins]))
            fin = self.finish_ops(X, self.nexp_b)
            n = len(fin)
            span = 31.0
            for k, (el, ex, ad, cv) in enumerate(fin):
                pos = k * span / n
                if ex:
                    fill.append((pos, ex))
                fill.append((pos + 1.2, ad))
                if cv:
                    fill.append((pos + 1.4, cv))
        if not mf:
            # no MFMAs (drain): plain sequence in position order
            return [x for _, ins in sorted(fill, key=lambda f: f[0]) for x in ins]
        return self.interleave(mf, fill)

    def max_ops(self, Y):
        out = []  # list of (group, [insts])
        for g in range(4):
            qb, kb = g >> 1, g & 1
            mx = V(V_MX[qb][kb])
            y = lambda r: V(Y + 16 * g + r)
            ops = [I("v_max3_f32", mx, y(0), y(1), y(2), tag=f"max g{g}")]
            for r in range(3, 15, 2):
                ops.append(I("v_max3_f32", mx, mx, y(r), y(r + 1), tag=f"max g{g}"))
            ops.append(I("v_max_f32", mx, mx, y(15), tag=f"max g{g}"))
            out.append((g, ops))
        return out

    def phase_b(self, p, with_pv=True, first_pv=False, with_start=True, init=False, with_kread=True, with_dma=True):
        """B(t), parity p: P.V(t) from S[p]  ||  start-softmax(t+1) on S[1-p]  ||  K(t+2) reads from KB[p]
        ||  LDS-DMA V(t+2) -> VB[p], K(t+3) -> KB[1-p].   init: first tile of a job (m := max, no decision)"""
        X, Y = SBUF[p], SBUF[1 - p]
        mf = self.pv_mfmas(X, first_pv) if with_pv else []
        fill = []
        pre = []
        post = []
        if with_kread:
            for k, ins in enumerate(self.k_reads(p)):
                fill.append((0.2 + k * 0.75, [ins]))
        if with_dma:
            pieces = []
            for j in range(4):
                pieces.append(self.dma_piece(S_VRS, V(V_DVO), S_VDMA, j, VB[p], (S_V32, S_V32P)))
            for j in range(4):
                pieces.append(self.dma_piece(S_KRS, V(V_DKO), S_KDMA, j, KB[1 - p], (S_K32, S_K32P)))
            for k, pc in enumerate(pieces):
                fill.append((13.5 + 2.2 * k, pc))
            post += [I("s_add_u32", S_VDMA, S_VDMA, S_V64), I("s_add_u32", S_KDMA, S_KDMA, S_K64), I("s_min_u32", S_KDMA, S_KDMA, S_KMAX)]
        if with_start:
            t0, t1 = V(V_T[0]), V(V_T[1])
            mxs = self.max_ops(Y)
            # groups 0..2 are long complete; group 3's chain ended with the last MFMA of phase A: keep it last
            pos = 0.0
            for g, ops in mxs:
                for ins in ops:
                    fill.append((pos, [ins]))
                    pos += 0.22
            pos = max(pos, 7.1)
            comb = []
            for qb in range(2):
                a, b = V(V_MX[qb][0]), V(V_MX[qb][1])
                comb += [I("v_max_f32", a, a, b), I("v_mov_b32", b, a)]
            comb += [I("s_nop", 1)]
            for qb in range(2):
                a, b = V(V_MX[qb][0]), V(V_MX[qb][1])
                comb += [I("v_permlane32_swap_b32", a, b)]
            for qb in range(2):
                a, b = V(V_MX[qb][0]), V(V_MX[qb][1])
                comb += [I("v_max_f32", a, a, b)]          # full-row maximum (raw score units) in every lane
            if init:
                for qb in range(2):
                    comb += [I("v_mul_f32", V(V_MC[qb]), S_C, V(V_MX[qb][0])),
                             I("v_mov_b32", V(V_RS[qb][0]), 0), I("v_mov_b32", V(V_RS[qb][1]), 0)]
                fill.append((pos, comb))
                for k in range(128):   # O^T := 0 (this phase has no MFMAs)
                    fill.append((pos + 0.01, [I("v_accvgpr_write_b32", A(k), 0)]))
            else:
                l_fire, l_back = self.lab("fire"), self.lab("fire_back")
                comb += [I("v_fma_f32", t0, V(V_MX[0][0]), S_C, -V(V_MC[0])), I("v_fma_f32", t1, V(V_MX[1][0]), S_C, -V(V_MC[1])),
                         I("v_max_f32", t0, t0, t1), I("v_cmp_gt_f32", VCC, t0, S_THR), I("s_cbranch_vccnz", Label(l_fire)),
                         label(l_back)]
                fill.append((pos, comb))
                # rare: raise the running maximum; O and the row sums are scaled at the END of this phase (after the
                # P.V MFMAs of tile t, which were exponentiated against the old maximum)
                blk = [label(l_fire)]
                for qb in range(2):
                    t2, t3 = V(V_T[2 + 2 * qb]), V(V_T[3 + 2 * qb])
                    blk += [I("v_mul_f32", t2, S_C, V(V_MX[qb][0])), I("v_max_f32", t2, t2, V(V_MC[qb])),
                            I("v_sub_f32", t3, V(V_MC[qb]), t2), I("v_mov_b32", V(V_MC[qb]), t2), I("v_exp_f32", V(V_CO[qb]), t3)]
                blk += [I("s_mov_b32", S_FLAG, 1), I("s_branch", Label(l_back))]
                self.ool.append(blk)
            # s' = s * c - m
            pos += 0.6
            nf = 64
            span = 31.0 - pos
            for k in range(nf):
                g = k >> 4
                qb = g >> 1
                y = V(Y + k)
                fill.append((pos + k * span / nf, [I("v_fma_f32", y, y, S_C, -V(V_MC[qb]), tag=f"fma {k}")]))
            # a few exp2 of the next finish phase ride here (phase A is the VALU-heavier one)
            for k in range(self.nexp_b):
                y = V(Y + k)
                fill.append((pos + k * span / nf + 3.0, [I("v_exp_f32", y, y, tag=f"exp {k}")]))
        if not mf:
            body = [x for _, ins in sorted(fill, key=lambda f: f[0]) for x in ins]
        else:
            body = self.interleave(mf, fill)
        body = pre + body + post
        if with_start and not init:
            # deferred rescale of O and the row sums (rare)
            l_rs, l_back = self.lab("rescale"), self.lab("rescale_back")
            body += [I("s_cmp_lg_u32", S_FLAG, 0), I("s_cbranch_scc1", Label(l_rs)), label(l_back)]
            blk = [label(l_rs), I("s_nop", 15)]
            tmp = [V(V_T[k]) for k in range(8)]
            for qb in range(2):
                for base in range(0, 64, 8):
                    regs = [A(qb * 64 + base + k) for k in range(8)]
                    blk += [I("v_accvgpr_read_b32", tmp[k], regs[k]) for k in range(8)]
                    blk += [I("v_mul_f32", tmp[k], tmp[k], V(V_CO[qb])) for k in range(8)]
                    blk += [I("v_accvgpr_write_b32", regs[k], tmp[k]) for k in range(8)]
                blk += [I("v_mul_f32", V(V_RS[qb][k]), V(V_RS[qb][k]), V(V_CO[qb])) for k in range(2)]
            blk += [I("s_mov_b32", S_FLAG, 0), I("s_nop", 3), I("s_branch", Label(l_back))]
            self.ool.append(blk)
        return body

    def sync_mid(self):
        return [waitcnt(vmcnt=0, lgkmcnt=0, comment="own DMA pieces landed; V fragments in"), I("s_barrier")]

    def step(self, p, **kw):
        """one tile step of parity p"""
        out = [comment(f"---- step parity {p}: phase A")]
        out += [waitcnt(lgkmcnt=0, comment="K fragments in")]
        out += self.phase_a(p, **{k: v for k, v in kw.items() if k in ("with_qk", "with_finish", "cinit")})
        out += self.sync_mid()
        out += [comment(f"---- step parity {p}: phase B")]
        out += self.phase_b(p, **{k: v for k, v in kw.items() if k in ("with_pv", "first_pv", "with_start", "init", "with_kread", "with_dma")})
        return out

    # ------------------------------------------------------------------ epilogue
    def k_epilogue(self):
        e = self.e
        t = [V(x) for x in V_T]
        e(comment("epilogue: l, 1/l, L; O^T -> rows through the wave's LDS slice -> global"))
        e(I("s_nop", 15))  # last P.V MFMAs -> accumulator reads
        for qb in range(2):
            l, m2, inv = t[0 + 3 * qb], t[1 + 3 * qb], t[2 + 3 * qb]
            e(I("v_add_f32", l, V(V_RS[qb][0]), V(V_RS[qb][1])), I("v_mov_b32", m2, l), I("s_nop", 1),
              I("v_permlane32_swap_b32", l, m2), I("v_add_f32", l, l, m2),
              I("v_rcp_f32", inv, l), I("v_log_f32", m2, l), I("s_nop", 0),
              # one Newton step: inv += inv * (1 - l * inv)
              I("v_fma_f32", l, -l, inv, 1.0), I("v_fma_f32", inv, l, inv, inv),
              I("v_add_f32", m2, m2, V(V_MC[qb])))
        # L store (lanes 0..31), in the I/O dtype
        e(I("s_lshr_b64", EXEC, EXEC, 32))
        for qb in range(2):
            m2 = t[1 + 3 * qb]
            e(I(self.cvt, m2, m2, m2), I("s_lshl_b32", S_T[0], S_QROW[qb], 1),
              I("buffer_store_short", m2, V(V_L2), S_LRS, S_T[0], offen=1))
        e(I("s_mov_b64", EXEC, -1))
        # O: 4 accumulators -> 2 packed registers -> ds_write_b64 at (row 32 qb + i, chunk 4 db + g4, +8 h)
        tmp = [V(V_T[6]), V(V_T[7]), V(V_T[8]), V(V_T[9])]
        addr = V(V_MX[0][0])
        for qb in range(2):
            inv = t[2 + 3 * qb]
            for db in range(4):
                for g4 in range(4):
                    src = A_O(qb, db)
                    e([I("v_accvgpr_read_b32", tmp[k], src.sub(4 * g4 + k)) for k in range(4)])
                    e([I("v_mul_f32", tmp[k], tmp[k], inv) for k in range(4)])
                    e(I(self.cvt, tmp[0], tmp[0], tmp[1]), I(self.cvt, tmp[1], tmp[2], tmp[3]))
                    e(I("v_xor_b32", addr, 4 * db + g4, V(V_ESW)), I("v_lshl_add_u32", addr, addr, 4, V(V_EW)))
                    e(I("ds_write_b64", addr, V(tmp[0].idx, 2), offset=8192 * qb))
        # read back whole rows and store: row 32 qb + 4 k + a
        e(waitcnt(lgkmcnt=0))
        rows = [V(SBUF[0] + 4 * k, 4) for k in range(16)]  # the score buffers are free now
        idx = 0
        for qb in range(2):
            for k in range(8):
                if k & 3:
                    e(I("v_xor_b32", addr, V(V_ER), (k & 3) << 4))
                    src_a = addr
                else:
                    src_a = V(V_ER)
                e(I("ds_read_b128", rows[idx], src_a, offset=8192 * qb + 1024 * k))
                idx += 1
        idx = 0
        for qb in range(2):
            e(I("s_mul_i32", S_T[0], S_QROW[qb], S_OSN), I("s_lshl_b32", S_T[1], S_OSN, 2))
            for k in range(8):
                e(waitcnt(lgkmcnt=15 - idx))
                e(I("buffer_store_dwordx4", rows[idx], V(V_EO), S_ORS, S_T[0], offen=1))
                if k < 7:
                    e(I("s_add_u32", S_T[0], S_T[0], S_T[1]))
                idx += 1

    # ------------------------------------------------------------------ the whole kernel
    def build(self):
        e = self.e
        name = self.name
        l_job, l_loop, l_tail, l_end, l_next = (f".L{name}_{s}" for s in ("job", "loop", "tail", "end", "next"))
        self.k_setup()
        e(I("s_cmp_ge_u32", S_JOB, S_TOTAL), I("s_cbranch_scc1", Label(l_end)))
        e(label(l_job))
        self.k_decode()
        if self.causal:
            # unit u, pass 0: qi = nq - 1 - u (heavy), pass 1: qi = u
            l_p1, l_pd = self.lab("pass1"), self.lab("passd")
            e(I("s_cmp_lg_u32", S_PASS, 0), I("s_cbranch_scc1", Label(l_p1)),
              I("s_sub_u32", S_QI, S_NQ, 1), I("s_sub_u32", S_QI, S_QI, S_UNIT), I("s_branch", Label(l_pd)),
              label(l_p1), I("s_mov_b32", S_QI, S_UNIT), label(l_pd))
        else:
            e(I("s_mov_b32", S_QI, S_UNIT))
        self.k_job_setup()
        self.k_job_first_loads()
        e(waitcnt(vmcnt=0), I("s_barrier"))
        # K(0) -> registers
        e(self.k_reads(0))
        # step -1 (parity 1): A = QK^T(0) only; B = start(0) as init, K(1) reads, DMA V(1), K(2)
        e(self.step(1, with_qk=True, with_finish=False, with_pv=False, init=True, cinit=self.cinit_for(0) if self.causal else None))
        # main loop over tile pairs (t, t+1), t = 0, 2, .. NT - 4
        e(I("s_sub_u32", S_LOOP, S_NT, 2), I("s_lshr_b32", S_LOOP, S_LOOP, 1))
        if self.causal:
            # the last two pairs (tiles NT-4 .. NT-1) are the diagonal: the steady loop covers t < NT - 4
            e(I("s_sub_u32", S_LOOP, S_LOOP, 1))
        e(I("s_cmp_eq_u32", S_LOOP, 0), I("s_cbranch_scc1", Label(l_tail)))
        e(label(l_loop))
        first = True
        # first_pv: the very first P.V of a job initialises O (C = 0): handled by peeling -- the loop body uses C = O, so O is
        # zeroed explicitly at job start instead (cheap: 128 v_accvgpr_write would cost more than the MFMA C = 0 form; we peel)
        e(self.step(0), self.step(1))
        e(I("s_sub_u32", S_LOOP, S_LOOP, 1), I("s_cmp_lg_u32", S_LOOP, 0), I("s_cbranch_scc1", Label(l_loop)))
        e(label(l_tail))
        if self.causal:
            self.k_causal_diag()
        else:
            e(self.step(0, with_kread=False, with_dma=False))
            e(self.step(1, with_qk=False, with_start=False, with_kread=False, with_dma=False))
        self.k_epilogue()
        # next job
        if self.causal:
            l_adv = self.lab("adv")
            # pass 0 -> pass 1 of the same unit unless the pair is a single tile (nq odd, middle)
            e(I("s_cmp_lg_u32", S_PASS, 0), I("s_cbranch_scc1", Label(l_adv)),
              I("s_sub_u32", S_T[0], S_NQ, 1), I("s_sub_u32", S_T[0], S_T[0], S_UNIT), I("s_cmp_eq_u32", S_T[0], S_UNIT),
              I("s_cbranch_scc1", Label(l_adv)),
              I("s_mov_b32", S_PASS, 1), I("s_branch", Label(l_next)),
              label(l_adv), I("s_mov_b32", S_PASS, 0), I("s_add_u32", S_JOB, S_JOB, S_NWG))
        else:
            e(I("s_add_u32", S_JOB, S_JOB, S_NWG))
        e(I("s_cmp_ge_u32", S_JOB, S_TOTAL), I("s_cbranch_scc1", Label(l_end)))
        e(label(l_next))
        # the epilogue's LDS slice and the K/V buffers are disjoint, but the next job's DMA must not overtake the slowest
        # wave's last K/V reads: all waves passed the last mid-step barrier after their final reads -> safe
        e(waitcnt(vmcnt=0), I("s_barrier"))
        e(I("s_branch", Label(l_job)))
        e(label(l_end), I("s_endpgm"))
        for blk in self.ool:
            e(blk)
        return self.prog

    # ------------------------------------------------------------------ causal diagonal (tiles NT-4 .. NT-1)
    def cinit_for(self, _):
        return None

    def k_causal_diag(self):
        raise NotImplementedError

    # ------------------------------------------------------------------ text
    def text(self):
        lines = [f'.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', ".amdhsa_code_object_version 6", ".text",
                 f".protected {self.name}", f".globl {self.name}", ".p2align 8", f".type {self.name},@function", f"{self.name}:"]
        lines += [x.text() for x in self.prog]
        lines += [f".L{self.name}_fend:", f".size {self.name}, .L{self.name}_fend-{self.name}", "",
                  '.section .rodata,"a",@progbits', ".p2align 6, 0x0", f".amdhsa_kernel {self.name}",
                  f"  .amdhsa_group_segment_fixed_size {LDS_TOTAL}", "  .amdhsa_private_segment_fixed_size 0",
                  f"  .amdhsa_kernarg_size {KARG_SIZE}", "  .amdhsa_user_sgpr_count 2", "  .amdhsa_user_sgpr_kernarg_segment_ptr 1",
                  "  .amdhsa_system_sgpr_workgroup_id_x 1", "  .amdhsa_system_vgpr_workitem_id 0",
                  "  .amdhsa_next_free_vgpr 512", "  .amdhsa_next_free_sgpr 102", "  .amdhsa_accum_offset 256",
                  "  .amdhsa_reserve_vcc 1", "  .amdhsa_ieee_mode 1", "  .amdhsa_dx10_clamp 1",
                  "  .amdhsa_float_round_mode_32 0", "  .amdhsa_float_round_mode_16_64 0",
                  "  .amdhsa_float_denorm_mode_32 3", "  .amdhsa_float_denorm_mode_16_64 3", ".end_amdhsa_kernel", ".text", ""]
        return "\n".join(lines)

    def metadata(self):
        return "\n".join([
            f"  - .args:", f"      - .offset: 0", f"        .size: {KARG_SIZE}", f"        .value_kind: by_value",
            f"    .group_segment_fixed_size: {LDS_TOTAL}", f"    .kernarg_segment_align: 8", f"    .kernarg_segment_size: {KARG_SIZE}",
            f"    .max_flat_workgroup_size: 256", f"    .name: {self.name}", f"    .private_segment_fixed_size: 0",
            f"    .sgpr_count: 108", f"    .symbol: {self.name}.kd", f"    .vgpr_count: 512", f"    .agpr_count: 256",
            f"    .wavefront_size: 64"])


def module_text(gens):
    body = "\n".join(g.text() for g in gens)
    md = ["", ".amdgpu_metadata", "---", "amdhsa.kernels:"] + [g.metadata() for g in gens] + [
        "amdhsa.target: amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", ".end_amdgpu_metadata", ""]
    return body + "\n".join(md)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--output", required=True)
    args = ap.parse_args(argv)
    from .check import check
    gens = []
    for dtype in ("bf16", "f16"):
        for causal in (False,):
            g = Gen(dtype, causal)
            g.build()
            errs = check(g.prog)
            if errs:
                print(f"{g.name}: {len(errs)} wait-state violations", file=sys.stderr)
                return 1
            gens.append(g)
    with open(args.output, "w") as f:
        f.write(module_text(gens))
    return 0


if __name__ == "__main__":
    sys.exit(main())
