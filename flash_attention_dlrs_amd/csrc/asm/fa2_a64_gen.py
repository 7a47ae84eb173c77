#!/usr/bin/env python3
"""Generator of the gfx950 assembly kernels `fa2_fwd_a64_<dtype>_<c|n>` -- FA-2 forward, d = 128, f16 / bf16.

Reference arithmetic: /root/reference/src/flash_attention_kernels.py:84-108 (exp2-domain online softmax in fp32, P rounded
RTNE to the I/O dtype before P.V, O /= l once at the end, L = m + log2 l), as in fa2_mfma16h.hip.

Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"):
  * workgroup = 4 waves = one 256-row Q block; a wave owns two 32-row query blocks (qb = 0, 1) and the WHOLE 512-entry
    register file: O^T in a[0:127], Q in a[128:191], the V^T fragments in a[192:255]; two score buffers, the current
    K tile and the softmax state in the arch VGPRs;
  * swapped products: S^T[key][query] = K.Q^T, O^T[d][query] += V^T.P^T on v_mfma_f32_32x32x16 -- a lane owns one query
    row per query block, P never leaves the registers (the S accumulator, packed in place, is the B operand of P.V);
  * 64-key K/V tiles arrive by LDS-DMA (buffer_load ... lds) into rings of FOUR K and four V buffers; LDS image = 8-row x
    32-column subtiles of 512 B with the 16-byte slots XOR-swizzled (T10 image (a)): row reads (ds_read_b128) and
    transposed reads (ds_read_b64_tr_b16) are conflict-free and need two per-lane base registers each, the rest is an
    immediate;
  * per tile t two phases of 32 MFMAs:  A(t) = QK^T(t+1) || finish-softmax(t) (exp2, row sums, cvt) || V(t) tr-reads
                                        B(t) = P.V(t)    || start-softmax(t+1) (row max, decision, s*c - m) || K(t+2) reads
                                                         || LDS-DMA of V(t+3), K(t+4)
    ONE barrier per tile (between A and B) behind a COUNTED `s_waitcnt vmcnt(8)`: a tile's DMA pieces have two tile steps
    to land.  The loop body is four tiles (buffer indices and score-buffer parity are immediates);
  * the tile stream is CONTINUOUS across jobs: every job has a multiple of four tiles, and its last body (the "seam")
    already streams the next job's K(0..3), V(0..2) and Q rows and computes its first QK^T; only the epilogue (O through
    the wave's LDS slice, L) sits between two jobs;
  * persistent grid: a workgroup walks its jobs (non-causal: one Q block; causal: the pair (nq-1-u, u)).

The instruction stream is built as isa.Inst objects: printed to a .s file for the assembler, checked by check.py (wait
states) and executed by emu.py in the CPU test-suite (tests/test_asm_emu.py) against an fp64 reference.
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
V_VR0, V_VR1 = 194, 195   # V transposed-read lane bases (u = 0 / 1), the V ring's LDS offset included
V_DKO, V_DVO = 196, 197   # LDS-DMA per-lane source offsets (K / V row stride)
V_MC = (198, 199)         # running row maximum in the exp2 domain (c * max), per query block
V_RS = ((200, 201), (202, 203))  # row-sum accumulators [qb][2]
V_MX = ((204, 205), (206, 207))  # row-max chains [qb][kb]
V_CO = (208, 209)         # rescale coefficient per query block
V_T = tuple(range(210, 220))     # temporaries (V_T[2] = v212 is 4-aligned: a zero MFMA operand in the epilogue)
V_QOFF = 220              # (unused)
V_LANE = 221
V_EW = 222                # epilogue LDS write base (row i, +8h)
V_ESW = 223               # epilogue swizzle term swz(i)
V_ER = 224                # epilogue LDS read base
V_EO = 225                # epilogue global store lane offset (os_n)
V_L2 = 226                # L store lane offset
V_ST_LAST, V_ST_ACC = 227, 228   # diagnostic builds: last stamp (low word), accumulators [4] (228..231)
V_DKO2, V_DVO2 = 232, 233  # V_DKO / V_DVO + 128 (second half of an 8-row piece)
V_LSV = (234, 235)        # row sums of the finished job, saved for its epilogue
V_MSV = (236, 237)        # running maximum of the finished job
V_DQE, V_DQO = 240, 241   # LDS-DMA per-lane source offsets of the Q rows (row stride qs_n; even / odd 8-row group)
V_QRE, V_QRO = 242, 243   # Q row-read lane bases in the wave's slice (even / odd k-step)
V_IMH = 238               # causal: i - 4 h (query row inside a 32-row block minus the lane half's key offset)
V_NINF = 239              # causal: -inf


# AGPRs
def A_O(qb, db):
    return A((qb * 4 + db) * 16, 16)


def A_Q(qb, ks):
    return A(128 + (qb * 8 + ks) * 4, 4)


def A_K(kb, ks):
    # the K tile lives in ARCH VGPRs v[128:191]: ds_read into accumulator registers while MFMAs write accumulators was
    # measured 470 cycles per tile slower (and skews the four waves at the barrier)
    return V(VF + (kb * 8 + ks) * 4, 4)


def V_F(kstep, db):
    # V^T fragments in a[192:255]: read in phase A, whose MFMAs (QK^T) write arch VGPRs
    return A(192 + 4 * (4 * kstep + db), 4)


# SGPRs.  s4..s47 hold the kernel arguments (loaded once).
S_KARG = S(0, 2)
S_WGID = S(2)
S_FINAL = S(3)
S_Q, S_K, S_V, S_O, S_L = S(4, 2), S(6, 2), S(8, 2), S(10, 2), S(12, 2)
S_QSB, S_QSH, S_KSB, S_KSH, S_VSB, S_VSH, S_OSB, S_OSH, S_LSB, S_LSH = (S(14 + 2 * k, 2) for k in range(10))
S_QSN, S_KSN, S_VSN, S_OSN = S(34), S(35), S(36), S(37)
S_N, S_H, S_NQ, S_TOTAL = S(38), S(39), S(40), S(41)
S_C, S_THR, S_NUNIT, S_G = S(42), S(43), S(44), S(45)
S_NBH, S_NWG = S(46), S(47)
S_KRS, S_VRS, S_NVRS, S_SQ = S(48, 4), S(52, 4), S(56, 4), S(60, 4)   # K / V descriptors, the next job's V, a scratch one
S_NB, S_NHH, S_NQI, S_NNT = S(64), S(65), S(66), S(67)                # the next job
S_JOB, S_WAVE = S(68), S(69)
S_KDMA, S_VDMA = S(70), S(71)    # source offset of the next K / V tile to stream (the wave's row base included)
S_K32, S_V32 = S(72), S(74)      # 32 rows of K / V in bytes
S_K64, S_V64 = S(76), S(77)
S_LDSW = S(78)                   # 2048 * wave: the wave's piece offset inside a ring buffer
S_LOOP, S_FLAG = S(79), S(80)
S_QI, S_B, S_HH, S_UNIT, S_PASS, S_NT = S(81), S(83), S(84), S(85), S(86), S(87)   # the current job
S_T = tuple(S(88 + k) for k in range(8))  # temporaries s88..s95 (S_T[0] even: usable as a 64-bit pair)
S_QROW = (S(96), S(97))          # first row of the wave's query block qb (current job)
S_DBG = S(98, 2)
S_KW, S_VW = S(100), S(101)      # 8 * wave * row stride: the wave's row base inside a tile
S_LG, S_X1 = S(73), S(75)        # decode shifts: lgH | lgG << 8 | lg(G * nunit) << 16 | pow2-mode << 24;  spare
S_X2 = S(82)

# LDS map (bytes)
KB = (0, 16384)
VBASE = 32768
VB = (0, 16384)                  # relative to VBASE (folded into the V read lane bases; absolute for the DMA)
EPI = 65536                      # + 16384 * wave: the wave's private 64 x 256-byte slice: the next job's Q rows land here by
                                 # LDS-DMA (K-tile image) on their way to a[128:191]; later the job's O rows leave through it
LDS_TOTAL = 131072

KARG_SIZE = 192
NSLOT = 24


class Gen:
    def __init__(self, dtype="bf16", causal=False, name=None, nexp_b=8, stamps=False, abl=(), ring=(2, 3, 2)):
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
        self.abl = set(abl)    # timing-only ablations of the steady loop (diagnostic builds; results wrong by construction)
        self.R, self.dk, self.dv = ring   # ring depth; K(t + dk) and V(t + dv) are streamed in phase B(t): dk <= R + 1, dv <= R
        assert 3 <= self.dk <= min(self.R + 1, 4) and 2 <= self.dv <= self.R and 4 % self.R == 0
        self.vm = 8 * min(self.dk - 3, self.dv - 2)  # DMA pieces that may stay in flight across the mid-step barrier
        self.stamps = stamps   # diagnostic build: s_memtime stamps of the job timeline go to the debug buffer

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

    def stamp(self, slot, real=False):
        """diagnostic builds only: dbg[(wg * 4 + wave) * NSLOT + slot] = s_memtime (or s_memrealtime)"""
        if not self.stamps:
            return []
        t = S(S_T[0].idx, 2)
        v = V(V_T[8], 2)
        return [I("s_memrealtime" if real else "s_memtime", t), waitcnt(lgkmcnt=0),
                I("v_mov_b32", v.sub(0), t.sub(0)), I("v_mov_b32", v.sub(1), t.sub(1)),
                I("v_mov_b32", V(V_T[7]), 0), I("global_store_dwordx2", V(V_T[7]), v, S_DBG, offset=8 * slot)]

    def stamp_acc(self, k):
        """diagnostic builds only: acc[k] += cycles since the previous stamp_acc"""
        if not self.stamps:
            return []
        t = S(S_T[0].idx, 2)
        tmp = V(V_T[9])
        return [I("s_memtime", t), waitcnt(lgkmcnt=0), I("v_sub_u32", tmp, t.sub(0), V(V_ST_LAST)),
                I("v_add_u32", V(V_ST_ACC + k), V(V_ST_ACC + k), tmp), I("v_mov_b32", V(V_ST_LAST), t.sub(0))]

    def stamp_flush(self):
        if not self.stamps:
            return []
        out = [I("v_mov_b32", V(V_T[7]), 0)]
        for k in range(4):
            out += [I("global_store_dword", V(V_T[7]), V(V_ST_ACC + k), S_DBG, offset=8 * (10 + k)),
                    I("v_mov_b32", V(V_ST_ACC + k), 0)]
        return out

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
        return [I("s_mul_i32", lo, idx, stride.sub(0)), I("s_mul_hi_u32", hi, idx, stride.sub(0)),
                I("s_add_u32", dst.sub(0), dst.sub(0), lo), I("s_addc_u32", dst.sub(1), dst.sub(1), hi),
                I("s_mul_i32", lo, idx, stride.sub(1)), I("s_add_u32", dst.sub(1), dst.sub(1), lo)]

    def make_desc(self, rs: Reg, base: Reg, sb: Reg, sh: Reg, b: Reg, hh: Reg):
        """raw buffer descriptor of the (b, hh) slice of a tensor: base + b * sb + hh * sh (the range check is not used:
        soffset is unchecked anyway; every address the kernel forms lies inside the tensor, N being a multiple of 256)"""
        tmp = S(S_T[0].idx, 2)
        return ([I("s_mov_b64", tmp, base)] + self.mad64(tmp, b, sb) + self.mad64(tmp, hh, sh) +
                [I("s_mov_b32", rs.sub(0), tmp.sub(0)), I("s_and_b32", rs.sub(1), tmp.sub(1), 0xFFFF),
                 I("s_mov_b32", rs.sub(2), 0x7FFFFFF0), I("s_mov_b32", rs.sub(3), 0x00020000)])

    # ------------------------------------------------------------------ kernel prologue: arguments, lane constants
    def k_setup(self):
        e = self.e
        e(comment("kernel arguments"),
          I("s_load_dwordx8", S(4, 8), S_KARG, 0), I("s_load_dwordx4", S(12, 4), S_KARG, 32),
          I("s_load_dwordx16", S(16, 16), S_KARG, 48), I("s_load_dwordx4", S(32, 4), S_KARG, 112),
          I("s_load_dwordx8", S(36, 8), S_KARG, 128), I("s_load_dwordx4", S(44, 4), S_KARG, 160),
          I("s_load_dwordx2", S_DBG, S_KARG, 176), I("s_load_dword", S_LG, S_KARG, 184))
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
        # ---- V transposed-read bases: VBASE + 64 (4 h + q) + 16 ((2 w + (p >> 1)) ^ ((2 u + h) & 3)) + 8 (p & 1)
        #      w = (lane >> 4) & 1, q = (lane >> 2) & 3, p = lane & 3
        e(comment("V transposed-read lane bases"),
          I("v_bfe_u32", t0, lane, 2, 2),                  # q
          I("v_lshrrev_b32", t3, 5, lane),                 # h
          I("v_lshl_add_u32", t0, t3, 2, t0),              # 4 h + q
          I("v_lshlrev_b32", t0, 6, t0),                   # 64 (4 h + q)
          I("v_and_b32", t1, 1, lane), I("v_lshl_add_u32", t0, t1, 3, t0),  # + 8 (p & 1)
          I("v_add_u32", t0, VBASE, t0),
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
        e(I("v_mul_lo_u32", t2, t1, S_KSN), I("v_add_u32", V(V_DKO), t2, t0), I("v_add_u32", V(V_DKO2), 128, V(V_DKO)),
          I("v_mul_lo_u32", t2, t1, S_VSN), I("v_add_u32", V(V_DVO), t2, t0), I("v_add_u32", V(V_DVO2), 128, V(V_DVO)))
        # ---- Q rows: the same piece shape for the 8-row groups R = 0..7 of the wave's 64 rows; the slot XOR is 2 (R & 1) + (l' >> 4)
        e(comment("Q staging: LDS-DMA lane offsets (even / odd row group) and row-read bases in the wave's slice"),
          I("v_and_b32", t0, 3, lane), I("v_bfe_u32", t2, lane, 4, 1),
          I("v_xor_b32", t3, t0, t2),                       # even R: slot ^ (l' >> 4)
          I("v_lshrrev_b32", t2, 5, lane), I("v_lshl_or_b32", t3, t2, 2, t3), I("v_lshlrev_b32", t3, 4, t3),
          I("v_mul_lo_u32", t2, t1, S_QSN), I("v_add_u32", V(V_DQE), t2, t3),
          I("v_xor_b32", t3, 32, t3),                       # odd R: slot ^ (2 + (l' >> 4)): bit 1 of the slot = bit 5 of 16 * chunk
          I("v_add_u32", V(V_DQO), t2, t3),
          I("s_lshl_b32", S_T[0], S_WAVE, 14), I("s_add_u32", S_T[0], S_T[0], EPI),
          I("v_add_u32", V(V_QRE), S_T[0], V(V_KRE)), I("v_add_u32", V(V_QRO), S_T[0], V(V_KRO)))
        # ---- epilogue: write base EPI + 16384 wave + 256 i + 8 h; swizzle term ((i & 3) << 2) | ((i >> 2) & 3)
        e(comment("epilogue lane constants"),
          I("v_and_b32", t0, 31, lane), I("v_lshrrev_b32", t3, 5, lane),
          I("v_lshlrev_b32", t1, 8, t0), I("v_lshl_add_u32", t1, t3, 3, t1), I("v_add_u32", V(V_EW), S_T[0], t1),
          I("v_and_b32", t1, 3, t0), I("v_lshlrev_b32", t1, 2, t1), I("v_bfe_u32", t2, t0, 2, 2), I("v_or_b32", t1, t1, t2),
          I("v_mov_b32", V(V_ESW), t1))
        # read-back: a = lane >> 4, ec = lane & 15: base + 256 a + ((ec ^ (a << 2)) << 4); store offset a * os_n + 16 ec
        e(I("v_lshrrev_b32", t0, 4, lane), I("v_and_b32", t1, 15, lane),
          I("v_lshlrev_b32", t2, 2, t0), I("v_xor_b32", t2, t2, t1), I("v_lshlrev_b32", t2, 4, t2),
          I("v_lshl_add_u32", t2, t0, 8, t2), I("v_add_u32", V(V_ER), S_T[0], t2),
          I("v_mul_lo_u32", t2, t0, S_OSN), I("v_lshl_add_u32", V(V_EO), t1, 4, t2),
          I("v_and_b32", t0, 31, lane), I("v_lshlrev_b32", V(V_L2), 1, t0))
        if self.stamps:
            e(I("s_lshl_b32", S_T[0], S_WGID, 2), I("s_add_u32", S_T[0], S_T[0], S_WAVE), I("s_mul_i32", S_T[0], S_T[0], 8 * NSLOT),
              I("s_add_u32", S_DBG.sub(0), S_DBG.sub(0), S_T[0]), I("s_addc_u32", S_DBG.sub(1), S_DBG.sub(1), 0))
            e(self.stamp(6, real=True), self.stamp(8))
            e([I("v_mov_b32", V(V_ST_ACC + k), 0) for k in range(4)], I("v_mov_b32", V(V_ST_LAST), 0))
        # ---- scalar constants
        e(I("s_lshl_b32", S_K32, S_KSN, 5), I("s_lshl_b32", S_V32, S_VSN, 5),
          I("s_lshl_b32", S_K64, S_KSN, 6), I("s_lshl_b32", S_V64, S_VSN, 6),
          I("s_lshl_b32", S_T[0], S_WAVE, 3), I("s_mul_i32", S_KW, S_T[0], S_KSN), I("s_mul_i32", S_VW, S_T[0], S_VSN),
          I("s_mov_b32", S_FLAG, 0), I("s_mov_b32", S_PASS, 0), I("s_mov_b32", S_FINAL, 0),
          I("s_mov_b32", S_JOB, S_WGID))
        if self.causal:
            e(comment("causal: lane constants of the diagonal mask"),
              I("v_and_b32", t0, 31, lane), I("v_lshrrev_b32", t3, 5, lane), I("v_lshlrev_b32", t3, 2, t3),
              I("v_sub_u32", V(V_IMH), t0, t3),   # i - 4 h
              I("v_mov_b32", V(V_NINF), float("-inf")))

    # ------------------------------------------------------------------ job decode: S_JOB (+ S_PASS) -> S_NB, S_NHH, S_NQI, S_NNT
    def k_decode_next(self):
        e = self.e
        l_else, l_done = self.lab("dec_else"), self.lab("dec_done")
        t = S_T
        bh = S_X2
        l_gen = self.lab("dec_generic")
        e(comment("job index -> (b, h), work unit, query block, tile count of the NEXT job"),
          I("s_lshr_b32", t[0], S_LG, 24), I("s_cmp_eq_u32", t[0], 0), I("s_cbranch_scc1", Label(l_gen)))
        # H, G, G * nunit powers of two and B * H a multiple of 8 (the host says so): shifts and masks only
        e(I("s_lshr_b32", t[0], S_JOB, 3),                                   # slot
          I("s_lshr_b32", t[1], S_LG, 16), I("s_and_b32", t[1], t[1], 255),   # lg(G nunit)
          I("s_lshr_b32", t[2], t[0], t[1]),                                  # batch
          I("s_lshl_b32", t[3], 1, t[1]), I("s_sub_u32", t[3], t[3], 1), I("s_and_b32", t[3], t[0], t[3]),   # r
          I("s_lshr_b32", t[1], S_LG, 8), I("s_and_b32", t[1], t[1], 255),    # lg G
          I("s_lshr_b32", S_UNIT, t[3], t[1]),                                # unit = r >> lgG
          I("s_lshl_b32", t[4], 1, t[1]), I("s_sub_u32", t[4], t[4], 1), I("s_and_b32", t[4], t[3], t[4]),   # r % G
          I("s_lshl_b32", t[2], t[2], t[1]), I("s_add_u32", t[2], t[2], t[4]), I("s_lshl_b32", t[2], t[2], 3),
          I("s_and_b32", t[0], S_JOB, 7), I("s_add_u32", bh, t[2], t[0]),
          I("s_and_b32", t[1], S_LG, 255),                                    # lg H
          I("s_lshr_b32", S_NB, bh, t[1]),
          I("s_lshl_b32", t[4], 1, t[1]), I("s_sub_u32", t[4], t[4], 1), I("s_and_b32", S_NHH, bh, t[4]))
        l_qi = self.lab("dec_qi")
        e(I("s_branch", Label(l_qi)), label(l_gen),
          I("s_and_b32", t[0], S_NBH, 7), I("s_cmp_lg_u32", t[0], 0), I("s_cbranch_scc1", Label(l_else)))
        # slot = id >> 3; GN = G * nunit; batch = slot / GN; r = slot % GN; bh = (batch * G + r % G) * 8 + (id & 7); unit = r / G
        e(I("s_lshr_b32", t[0], S_JOB, 3), I("s_mul_i32", t[1], S_G, S_NUNIT))
        self.udiv(t[2], t[3], t[0], t[1])       # batch, r
        self.udiv(S_UNIT, t[4], t[3], S_G)      # unit = r / G, r % G
        e(I("s_mul_i32", t[2], t[2], S_G), I("s_add_u32", t[2], t[2], t[4]), I("s_lshl_b32", t[2], t[2], 3),
          I("s_and_b32", t[0], S_JOB, 7), I("s_add_u32", bh, t[2], t[0]), I("s_branch", Label(l_done)))
        e(label(l_else))
        self.udiv(bh, S_UNIT, S_JOB, S_NUNIT)
        e(label(l_done))
        self.udiv(S_NB, S_NHH, bh, S_H)
        e(label(l_qi))
        if self.causal:
            # unit u, pass 0: qi = nq - 1 - u (heavy), pass 1: qi = u;  tiles = 4 (qi + 1)
            l_p1, l_pd = self.lab("pass1"), self.lab("passd")
            e(I("s_cmp_lg_u32", S_PASS, 0), I("s_cbranch_scc1", Label(l_p1)),
              I("s_sub_u32", S_NQI, S_NQ, 1), I("s_sub_u32", S_NQI, S_NQI, S_UNIT), I("s_branch", Label(l_pd)),
              label(l_p1), I("s_mov_b32", S_NQI, S_UNIT), label(l_pd),
              I("s_add_u32", t[0], S_NQI, 1), I("s_lshl_b32", S_NNT, t[0], 2))
        else:
            e(I("s_mov_b32", S_NQI, S_UNIT), I("s_lshr_b32", S_NNT, S_N, 6))

    def k_advance(self):
        """S_JOB / S_PASS -> the job after the most recently decoded one, decoded into the next-job registers;
        S_FINAL = 1 if there is none (the next-job registers then repeat the current job)"""
        e = self.e
        l_fin, l_ok = self.lab("adv_final"), self.lab("adv_ok")
        if self.causal:
            l_adv = self.lab("adv")
            # pass 0 -> pass 1 of the same unit unless the pair is a single tile (nq odd, middle)
            e(I("s_cmp_lg_u32", S_PASS, 0), I("s_cbranch_scc1", Label(l_adv)),
              I("s_sub_u32", S_T[0], S_NQ, 1), I("s_sub_u32", S_T[0], S_T[0], S_UNIT), I("s_cmp_eq_u32", S_T[0], S_UNIT),
              I("s_cbranch_scc1", Label(l_adv)),
              I("s_mov_b32", S_PASS, 1), I("s_branch", Label(l_ok)),
              label(l_adv), I("s_mov_b32", S_PASS, 0), I("s_add_u32", S_JOB, S_JOB, S_NWG))
        else:
            e(I("s_add_u32", S_JOB, S_JOB, S_NWG))
        e(I("s_cmp_ge_u32", S_JOB, S_TOTAL), I("s_cbranch_scc1", Label(l_fin)), label(l_ok))
        self.k_decode_next()
        l_done = self.lab("adv_done")
        e(I("s_branch", Label(l_done)), label(l_fin),
          comment("no further job: the seam streams the current job's first tiles again (results discarded)"),
          I("s_mov_b32", S_FINAL, 1), I("s_mov_b32", S_NB, S_B), I("s_mov_b32", S_NHH, S_HH), I("s_mov_b32", S_NQI, S_QI),
          I("s_mov_b32", S_NNT, S_NT), label(l_done))

    def k_promote(self):
        """next job -> current job"""
        self.e(I("s_mov_b32", S_B, S_NB), I("s_mov_b32", S_HH, S_NHH), I("s_mov_b32", S_QI, S_NQI), I("s_mov_b32", S_NT, S_NNT),
               # query rows of this wave: qrow[qb] = 256 qi + 64 wave + 32 qb
               I("s_lshl_b32", S_T[0], S_QI, 8), I("s_lshl_b32", S_T[1], S_WAVE, 6), I("s_add_u32", S_QROW[0], S_T[0], S_T[1]),
               I("s_add_u32", S_QROW[1], S_QROW[0], 32))

    # ------------------------------------------------------------------ LDS-DMA
    def dma_piece(self, which, piece, buf):
        """one 1-KiB LDS-DMA piece of the next K / V tile into ring buffer `buf`.  piece j: rows 8 R .. 8 R + 7 with
        R = wave (j < 2) or wave + 4, 128-byte half j & 1 (the +128 rides in the second lane-offset register)"""
        if which == "k":
            rsrc, vl, vl2, base, s32, lds0 = S_KRS, V(V_DKO), V(V_DKO2), S_KDMA, S_K32, KB[buf]
        else:
            rsrc, vl, vl2, base, s32, lds0 = S_VRS, V(V_DVO), V(V_DVO2), S_VDMA, S_V32, VBASE + VB[buf]
        out = []
        so = base
        if piece >= 2:
            so = S_T[5]
            out.append(I("s_add_u32", so, base, s32))
        out.append(I("s_add_u32", M0, S_LDSW, lds0 + (0, 1024, 8192, 9216)[piece]))
        out.append(I("s_nop", 0))
        out.append(I("buffer_load_dwordx4", vl2 if piece & 1 else vl, rsrc, so, offen=1, lds=1, tag=f"dma {which}{piece}"))
        return out

    def dma_tile(self, which, buf):
        out = []
        for j in range(4):
            out += self.dma_piece(which, j, buf)
        out.append(I("s_add_u32", S_KDMA, S_KDMA, S_K64) if which == "k" else I("s_add_u32", S_VDMA, S_VDMA, S_V64))
        return out

    def q_stage(self, b: Reg, hh: Reg, qi: Reg):
        """Q rows of job (b, hh, qi) of this wave -> the wave's LDS slice by LDS-DMA, in the K-tile image (16 pieces of 8 rows x
        128 bytes: coalesced, ~25 cycles of issue each; the same rows fetched straight into the MFMA operand layout -- 32 rows x
        32 bytes per instruction -- cost ~210 cycles per load).  Returns (descriptor / offset setup, [pieces])"""
        setup = self.make_desc(S_SQ, S_Q, S_QSB, S_QSH, b, hh)
        setup += [I("s_lshl_b32", S_T[0], qi, 8), I("s_lshl_b32", S_T[1], S_WAVE, 6), I("s_add_u32", S_T[0], S_T[0], S_T[1]),
                  I("s_mul_i32", S_T[2], S_T[0], S_QSN),                # byte offset of the wave's first row
                  I("s_lshl_b32", S_T[3], S_QSN, 3),                    # 8 rows
                  I("s_lshl_b32", S_T[4], S_WAVE, 14), I("s_add_u32", S_T[4], S_T[4], EPI)]
        pieces = []
        for R in range(8):
            for half in range(2):
                pc = []
                so = S_T[2]
                if half:
                    so = S_T[5]
                    pc.append(I("s_add_u32", so, S_T[2], 128))
                pc += [I("s_add_u32", M0, S_T[4], 2048 * R + 1024 * half), I("s_nop", 0),
                       I("buffer_load_dwordx4", V(V_DQO if R & 1 else V_DQE), S_SQ, so, offen=1, lds=1, tag=f"qdma R{R} h{half}")]
                if half and R < 7:
                    pc.append(I("s_add_u32", S_T[2], S_T[2], S_T[3]))
                pieces.append(pc)
        return setup, pieces

    def q_reads(self):
        """the staged Q rows -> a[128:191] (fragment (qb, ks) = rows 32 qb + i, 16-byte chunk 2 ks + h, as a K row read)"""
        out = []
        for qb in range(2):
            for ks in range(8):
                out.append(I("ds_read_b128", A_Q(qb, ks), V(V_QRO if ks & 1 else V_QRE), offset=8192 * qb + 512 * (ks >> 1), tag=f"qread qb{qb} ks{ks}"))
        return out

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

    def pv_mfmas(self, X):
        """O^T[qb][db] += V^T(kstep, db) . P^T(qb, kstep); P(qb, kstep = 2 kb + s) = X + 16 (2 qb + kb) + 4 s"""
        out = []
        for kstep in range(4):
            kb, s = kstep >> 1, kstep & 1
            for db in range(4):
                for qb in range(2):
                    p = V(X + 16 * (2 * qb + kb) + 4 * s, 4)
                    out.append(I(self.mfma, A_O(qb, db), V_F(kstep, db), p, A_O(qb, db), tag=f"pv ks{kstep} db{db} qb{qb}"))
        return out

    def v_reads(self, buf):
        """32 transposed reads of the V tile in VB[buf]: fragment (kstep, db) <- u = 0, 1"""
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

    def phase_a(self, t4, with_qk=True, with_finish=True, cinit=None, steady=False):
        """A(t), t4 = t & 3: QK^T(t+1) -> S[1-p]  ||  finish(S[p])  ||  V(t) reads from VB[t4]"""
        p = t4 & 1
        X, Y = SBUF[p], SBUF[1 - p]
        mf = self.qk_mfmas(Y, cinit) if with_qk else []
        fill = []
        abl = self.abl if steady else set()
        if with_finish:
            vr = self.v_reads(t4 % self.R)
            for k, ins in enumerate(vr):           # 2 reads per gap over the first 16 gaps
                if "novread" not in abl:
                    fill.append((k * 0.5, [ins]))
            fin = self.finish_ops(X, self.nexp_b) if "nofinish" not in abl else []
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
            return [x for _, ins in sorted(fill, key=lambda f: f[0]) for x in ins]
        return self.interleave(mf, fill)

    def max_ops(self, Y):
        """row-max chains of the four score groups, interleaved so that no instruction waits on its predecessor; group 3's
        chain (whose MFMAs ended phase A) starts late (MFMA result -> VALU read needs 12 wait states)"""
        chains = []
        for g in range(4):
            qb, kb = g >> 1, g & 1
            mx = V(V_MX[qb][kb])
            y = lambda r, g=g: V(Y + 16 * g + r)
            ops = [I("v_max3_f32", mx, y(0), y(1), y(2), tag=f"max g{g}")]
            for r in range(3, 15, 2):
                ops.append(I("v_max3_f32", mx, mx, y(r), y(r + 1), tag=f"max g{g}"))
            ops.append(I("v_max_f32", mx, mx, y(15), tag=f"max g{g}"))
            chains.append(ops)
        out = []
        idx = [0, 0, 0, 0]
        while any(idx[g] < len(chains[g]) for g in range(4)):
            for g in range(4):
                if g == 3 and len(out) < 14 and any(idx[k] < len(chains[k]) for k in range(3)):
                    continue
                if idx[g] < len(chains[g]):
                    out.append(chains[g][idx[g]])
                    idx[g] += 1
        return out

    def mask_block(self, Y, jd, cond=None):
        """causal: the tile whose softmax starts in this phase B is diagonal tile jd (0..3) of its job: keys 64 jd .. 64 jd + 63 of
        the 256-key diagonal span against this wave's rows 64 w .. 64 w + 63.  w > jd: nothing; w == jd: score groups
        (qb0, kb0) and (qb1, kb1) get the triangle (key > query -> -inf), (qb0, kb1) is all -inf; w < jd: all -inf.
        Register r of a group <-> key (r & 3) + 8 (r >> 2) + 4 h, lane <-> query i.  cond: (sgpr, value) extra run-time
        condition (the tile is diagonal at all).  Returns the in-line test; the masking runs out of line."""
        l_eq, l_lt, l_back = self.lab("mask_eq"), self.lab("mask_lt"), self.lab("mask_back")
        out = []
        if cond is not None:
            out += [I("s_cmp_lg_u32", cond[0], cond[1]), I("s_cbranch_scc1", Label(l_back))]
        out += [I("s_cmp_eq_u32", S_WAVE, jd), I("s_cbranch_scc1", Label(l_eq))]
        if jd > 0:
            out += [I("s_cmp_lt_u32", S_WAVE, jd), I("s_cbranch_scc1", Label(l_lt))]
        out += [label(l_back)]
        ninf = V(V_NINF)
        blk = [label(l_eq), I("s_nop", 11)]   # the chain of group 3 ended with the last MFMA of phase A
        for r in range(16):
            key = (r & 3) + 8 * (r >> 2)
            blk += [I("v_cmp_lt_i32", VCC, V(V_IMH), key),
                    I("v_cndmask_b32", V(Y + r), V(Y + r), ninf, VCC), I("v_cndmask_b32", V(Y + 48 + r), V(Y + 48 + r), ninf, VCC),
                    I("v_mov_b32", V(Y + 16 + r), ninf)]
        blk += [I("s_branch", Label(l_back))]
        self.ool.append(blk)
        if jd > 0:
            blk = [label(l_lt), I("s_nop", 11)]
            blk += [I("v_mov_b32", V(Y + r), ninf) for r in range(64)]
            blk += [I("s_branch", Label(l_back))]
            self.ool.append(blk)
        return out

    def phase_b(self, t4, with_pv=True, with_start=True, init=False, with_kread=True, with_dma=True, steady=False,
                save=False, pre=(), qload=None, mask=None, early=(), late=()):
        """B(t), t4 = t & 3: P.V(t) from S[p]  ||  start-softmax(t+1) on S[1-p]  ||  K(t+2) reads from KB[(t+2) & 3]
        ||  LDS-DMA V(t+3) -> VB[(t+3) & 3], K(t+4) -> KB[t4].
        init: the tile started here is the first of a job (m := its row maximum, sums := 0, no decision)
        save: this is the last tile of a job: its row sums and maximum are put aside for the epilogue
        pre: instructions ahead of the phase (descriptor switches);  qload: the next job's Q loads, issued first"""
        p = t4 & 1
        X, Y = SBUF[p], SBUF[1 - p]
        mf = self.pv_mfmas(X) if with_pv else []
        fill = []
        head = list(pre)
        post = []
        abl = self.abl if steady else set()
        if "nokread" in abl:
            with_kread = False
        if "nostart" in abl:
            with_start = False
        if save:
            for qb in range(2):
                head += [I("v_add_f32", V(V_LSV[qb]), V(V_RS[qb][0]), V(V_RS[qb][1])), I("v_mov_b32", V(V_MSV[qb]), V(V_MC[qb]))]
        if qload:
            head += qload
        if mask is not None:
            head += self.mask_block(Y, *mask)
        # scalar work and register loads that only have to precede this phase's DMA pieces: spread over the first gaps
        ne = len(early)
        for k, ins in enumerate(early):
            fill.append((0.3 + 11.0 * k / max(ne, 1), [ins]))
        if with_kread:
            for k, ins in enumerate(self.k_reads((t4 + 2) % self.R)):
                fill.append((0.2 + k * 0.75, [ins]))
        if with_dma:
            pieces = [self.dma_piece("v", j, (t4 + self.dv) % self.R) for j in range(4)] + \
                [self.dma_piece("k", j, (t4 + self.dk) % self.R) for j in range(4)]
            for k, pc in enumerate(pieces):
                if "nodma" in abl:
                    continue
                fill.append((12.5 + 2.3 * k, pc) if not late else (6.0 + 1.2 * k, pc))
            # further DMA pieces (the next job's Q rows) behind this step's own: the counted waits rely on that order
            for k, pc in enumerate(late):
                fill.append((16.0 + 15.5 * k / len(late), pc))
            post += [I("s_add_u32", S_VDMA, S_VDMA, S_V64), I("s_add_u32", S_KDMA, S_KDMA, S_K64)]
        if with_start:
            t0, t1 = V(V_T[0]), V(V_T[1])
            pos = 0.0
            for ins in self.max_ops(Y):
                if "nomax" not in abl:
                    fill.append((pos, [ins]))
                pos += 0.22
            pos = max(pos, 7.1)
            comb = []
            for qb in range(2):
                a, b = V(V_MX[qb][0]), V(V_MX[qb][1])
                comb += [I("v_max_f32", a, a, b)]
            for qb in range(2):
                a, b = V(V_MX[qb][0]), V(V_MX[qb][1])
                comb += [I("v_mov_b32", b, a)]
            comb += [I("s_nop", 0)]
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
                if "nofma" not in abl:
                    fill.append((pos + k * span / nf, [I("v_fma_f32", y, y, S_C, -V(V_MC[qb]), tag=f"fma {k}")]))
            # a few exp2 of the next finish phase ride here (phase A is the VALU-heavier one)
            for k in range(self.nexp_b):
                y = V(Y + k)
                fill.append((pos + k * span / nf + 3.0, [I("v_exp_f32", y, y, tag=f"exp {k}")]))
        if not mf:
            body = [x for _, ins in sorted(fill, key=lambda f: f[0]) for x in ins]
        else:
            body = self.interleave(mf, fill)
        body = head + body + post
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

    def sync_mid(self, steady=False, vm=None):
        if vm is not None:
            return [waitcnt(vmcnt=vm, lgkmcnt=0), I("s_barrier")]
        if steady and "novmwait" in self.abl:
            return [waitcnt(lgkmcnt=0), I("s_barrier")]
        if steady and "nobarrier" in self.abl:
            return [waitcnt(vmcnt=self.vm, lgkmcnt=0)]
        return [waitcnt(vmcnt=self.vm, lgkmcnt=0, comment="the DMA pieces of two steps ago have landed; V fragments in"), I("s_barrier")]

    def step(self, t4, a_pre=(), **kw):
        """one tile step, t4 = t & 3"""
        out = [comment(f"---- step {t4}: phase A")]
        out += self.stamp_acc(2)
        out += list(a_pre)
        out += [waitcnt(lgkmcnt=0, comment="K fragments in")]
        out += self.phase_a(t4, **{k: v for k, v in kw.items() if k in ("with_qk", "with_finish", "cinit", "steady")})
        out += self.stamp_acc(0)
        out += self.sync_mid(kw.get("steady", False), kw.get("vm"))
        out += self.stamp_acc(1)
        out += [comment(f"---- step {t4}: phase B")]
        out += self.phase_b(t4, **{k: v for k, v in kw.items()
                                  if k in ("with_pv", "with_start", "init", "with_kread", "with_dma", "steady", "save", "pre", "qload", "mask", "early", "late")})
        return out

    # ------------------------------------------------------------------ epilogue of the current job
    def k_epilogue(self):
        e = self.e
        t = [V(x) for x in V_T]
        e(comment("epilogue: l, 1/l, L; O^T -> rows through the wave's LDS slice -> global; O^T := 0"))
        e(self.make_desc(S_SQ, S_L, S_LSB, S_LSH, S_B, S_HH))
        e(I("s_nop", 7))  # last P.V MFMAs -> accumulator reads (the descriptor arithmetic above counts as well)
        l = [t[0], t[3]]
        m2 = [t[1], t[4]]
        inv = [t[2], t[5]]
        for qb in range(2):
            e(I("v_mov_b32", l[qb], V(V_LSV[qb])), I("v_mov_b32", m2[qb], V(V_LSV[qb])))
        e(I("s_nop", 1))
        for qb in range(2):
            e(I("v_permlane32_swap_b32", l[qb], m2[qb]))
        for qb in range(2):
            e(I("v_add_f32", l[qb], l[qb], m2[qb]))
        for qb in range(2):
            e(I("v_rcp_f32", inv[qb], l[qb]), I("v_log_f32", m2[qb], l[qb]))
        e(I("s_nop", 0))
        for qb in range(2):
            # one Newton step: inv += inv * (1 - l * inv)
            e(I("v_fma_f32", l[qb], -l[qb], inv[qb], 1.0), I("v_add_f32", m2[qb], m2[qb], V(V_MSV[qb])))
        for qb in range(2):
            e(I("v_fma_f32", inv[qb], l[qb], inv[qb], inv[qb]), I(self.cvt, m2[qb], m2[qb], m2[qb]))
        # L store (lanes 0..31), in the I/O dtype
        e(I("s_lshr_b64", EXEC, EXEC, 32))
        for qb in range(2):
            e(I("s_lshl_b32", S_T[0], S_QROW[qb], 1), I("buffer_store_short", m2[qb], V(V_L2), S_SQ, S_T[0], offen=1))
        e(I("s_mov_b64", EXEC, -1))
        e(self.make_desc(S_SQ, S_O, S_OSB, S_OSH, S_B, S_HH))
        # O: 4 accumulators -> 2 packed registers -> ds_write_b64 at (row i, chunk 4 db + g4, +8 h); one query block at a time.
        # (S[0] already holds the next job's first scores and v[128:191] its K(1): rows and temporaries are score buffer 1,
        # whose P was consumed by the job's last P.V.)  A batch = the four 8-byte groups of one 32-column block; the stages
        # of consecutive batches (accumulator reads | scale | pack, address | LDS write) are woven so that no instruction
        # waits on its predecessor.
        rows = [V(SBUF[1] + 4 * k, 4) for k in range(8)]
        tset = [[V(SBUF[1] + 32 + 16 * sidx + k) for k in range(16)] for sidx in range(2)]
        aset = [[V(V_T[6]), V(V_T[7]), V(V_T[8]), V(V_T[9])], [V(V_MX[0][0]), V(V_MX[0][1]), V(V_MX[1][0]), V(V_MX[1][1])]]
        addr2 = V(V_CO[0])

        def weave(*lists):
            out, idx = [], [0] * len(lists)
            while any(idx[k] < len(lists[k]) for k in range(len(lists))):
                for k in range(len(lists)):
                    if idx[k] < len(lists[k]):
                        out.append(lists[k][idx[k]])
                        idx[k] += 1
            return out
        for qb in range(2):
            stages = []  # per batch: [reads, muls + address, packs, writes]
            for db in range(4):
                tm, ad = tset[db & 1], aset[db & 1]
                src = A_O(qb, db)
                rd = [I("v_accvgpr_read_b32", tm[k], src.sub(k)) for k in range(16)]
                mu = [I("v_mul_f32", tm[k], tm[k], inv[qb]) for k in range(16)]
                ax = [I("v_xor_b32", ad[g4], 4 * db + g4, V(V_ESW)) for g4 in range(4)]
                al = [I("v_lshl_add_u32", ad[g4], ad[g4], 4, V(V_EW)) for g4 in range(4)]
                cv = []
                for g4 in range(4):
                    cv += [I(self.cvt, tm[4 * g4], tm[4 * g4], tm[4 * g4 + 1]), I(self.cvt, tm[4 * g4 + 1], tm[4 * g4 + 2], tm[4 * g4 + 3])]
                wr = [I("ds_write_b64", ad[g4], V(tm[4 * g4].idx, 2)) for g4 in range(4)]
                stages.append((rd, weave(mu, ax), weave(cv, al), wr))
            # software pipeline over the four batches (two register sets): batch b + 1 is read while batch b is scaled, ...
            e(stages[0][0])
            e(weave(stages[0][1], stages[1][0]))
            e(stages[0][2], stages[0][3])
            e(weave(stages[1][1], stages[2][0]))
            e(stages[1][2], stages[1][3])
            e(weave(stages[2][1], stages[3][0]))
            e(stages[2][2], stages[2][3])
            e(stages[3][1], stages[3][2], stages[3][3])
            # read back whole rows: row 4 k + a
            for k in range(8):
                if k & 3:
                    e(I("v_xor_b32", addr2, V(V_ER), (k & 3) << 4))
                    src_a = addr2
                else:
                    src_a = V(V_ER)
                e(I("ds_read_b128", rows[k], src_a, offset=1024 * k))
            e(I("s_mul_i32", S_T[0], S_QROW[qb], S_OSN), I("s_lshl_b32", S_T[1], S_OSN, 2))
            for k in range(8):
                e(waitcnt(lgkmcnt=7 - k))
                e(I("buffer_store_dwordx4", rows[k], V(V_EO), S_SQ, S_T[0], offen=1))
                if k < 7:
                    e(I("s_add_u32", S_T[0], S_T[0], S_T[1]))
        # O^T := 0 for the next job, on the matrix pipe (8 MFMAs instead of 128 v_accvgpr_write)
        z = V(V_T[2], 4)
        e([I("v_mov_b32", z.sub(k), 0) for k in range(4)], I("s_nop", 1))
        for qb in range(2):
            for db in range(4):
                e(I(self.mfma, A_O(qb, db), z, z, 0))

    # ------------------------------------------------------------------ the whole kernel
    def build(self):
        e = self.e
        name = self.name
        l_job, l_loop, l_seam, l_end = (f".L{name}_{s}" for s in ("job", "loop", "seam", "end"))
        self.k_setup()
        e(I("s_cmp_ge_u32", S_JOB, S_TOTAL), I("s_cbranch_scc1", Label(l_end)))
        # ---- first job of this workgroup: decode, descriptors, first loads, pipeline fill
        self.k_decode_next()
        self.k_promote()
        e(comment("first job: K / V descriptors, K(0..2), V(0..1) by LDS-DMA, Q rows"))
        e(self.make_desc(S_KRS, S_K, S_KSB, S_KSH, S_B, S_HH), self.make_desc(S_VRS, S_V, S_VSB, S_VSH, S_B, S_HH))
        e(I("s_mov_b32", S_KDMA, S_KW), I("s_mov_b32", S_VDMA, S_VW))
        e(self.stamp(0))
        for j in range(self.dk - 1):
            e(self.dma_tile("k", j % self.R))
            if j < self.dv - 1:
                e(self.dma_tile("v", j % self.R))
        qs_setup, qs_pieces = self.q_stage(S_B, S_HH, S_QI)
        e(qs_setup, qs_pieces)
        e([I("v_accvgpr_write_b32", A(k), 0) for k in range(128)])   # O^T := 0
        e(waitcnt(vmcnt=0), I("s_barrier"))
        e(self.stamp(1))
        e(self.q_reads(), self.k_reads(0))
        # step -1 (buffers as t4 = 3): A = QK^T(0) only; B = start(0) as init, K(1) reads, DMA V(2), K(3)
        e(self.step(3, with_qk=True, with_finish=False, with_pv=False, init=True, mask=(0, (S_NT, 4)) if self.causal else None))
        e(self.stamp(2), self.stamp_flush(), self.stamp_acc(3))
        # ---- job loop
        e(label(l_job))
        e(I("s_lshr_b32", S_LOOP, S_NT, 2), I("s_sub_u32", S_LOOP, S_LOOP, 1),
          I("s_cmp_eq_u32", S_LOOP, 0), I("s_cbranch_scc1", Label(l_seam)))
        e(label(l_loop))
        for t4 in range(4):
            # causal: the last steady body starts the job's first diagonal tile in its last phase B
            e(self.step(t4, steady=True, mask=(0, (S_LOOP, 1)) if self.causal and t4 == 3 else None))
        e(I("s_sub_u32", S_LOOP, S_LOOP, 1), I("s_cmp_lg_u32", S_LOOP, 0), I("s_cbranch_scc1", Label(l_loop)))
        e(label(l_seam))
        e(self.stamp(3), self.stamp_acc(2), self.stamp_flush())
        # ---- the job's last four tiles: the next job's K / V / Q stream in, its first QK^T and softmax start run here
        self.k_advance()
        cm = self.causal
        if True:
            kpre = self.make_desc(S_KRS, S_K, S_KSB, S_KSH, S_NB, S_NHH) + [I("s_mov_b32", S_KDMA, S_KW)] + \
                self.make_desc(S_NVRS, S_V, S_VSB, S_VSH, S_NB, S_NHH)
            vpre = [I("s_mov_b32", S_VRS.sub(k), S_NVRS.sub(k)) for k in range(4)] + [I("s_mov_b32", S_VDMA, S_VW)]
            sk, sv = 4 - self.dk, 4 - self.dv      # seam step whose phase B streams the next job's first K / V tile
            qs_setup, qs_pieces = self.q_stage(S_NB, S_NHH, S_NQI)
            for st in range(4):
                kw = dict(mask=((st + 1,) if st < 3 else (0, (S_NNT, 4))) if cm else None)
                early, pre = [], []
                if st == 0:
                    # the next job's Q rows start their way into the wave's LDS slice, BEHIND this step's K / V pieces; the
                    # next barrier wait leaves them in flight (vmcnt(16 + ...)), the one after retires them
                    early += qs_setup
                    kw.update(late=qs_pieces)
                if st == 1:
                    kw.update(vm=self.vm + 16)
                if st == sk:
                    early += kpre
                if st == sv:
                    pre += vpre
                if st == 2:
                    early += self.q_reads()      # slice -> a[128:191] (Q was last read by this step's phase A)
                if st == 3:
                    kw.update(init=True, save=True)
                e(self.stamp(16 + st))
                e(self.step(st, early=early, pre=pre, **kw))
        e(self.stamp(4))
        self.k_epilogue()
        e(self.stamp(5))
        e(I("s_cmp_lg_u32", S_FINAL, 0), I("s_cbranch_scc1", Label(l_end)))
        self.k_promote()
        e(self.stamp(0), self.stamp_acc(3))
        e(I("s_branch", Label(l_job)))
        e(label(l_end), waitcnt(vmcnt=0), self.stamp(7, real=True), self.stamp(9), I("s_endpgm"))
        for blk in self.ool:
            e(blk)
        return self.prog

    # ------------------------------------------------------------------ text
    def text(self):
        lines = [f".protected {self.name}", f".globl {self.name}", ".p2align 8", f".type {self.name},@function", f"{self.name}:"]
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


ABLATIONS = {"kreadv": ("kread_vgpr",), "novmwait": ("novmwait",), "nobarrier": ("nobarrier",), "nomax": ("nomax",), "nofma": ("nofma",),
             "nodma": ("nodma",), "nokread": ("nokread",), "nostart": ("nostart",), "nofinish": ("nofinish",),
             "novread": ("novread",), "mfmaonly": ("nodma", "nokread", "nostart", "nofinish", "novread")}


def module_text(gens):
    head = ['.amdgcn_target "amdgcn-amd-amdhsa--gfx950"', ".amdhsa_code_object_version 6", ".text", ""]
    body = "\n".join(head) + "\n".join(g.text() for g in gens)
    md = ["", ".amdgpu_metadata", "---", "amdhsa.kernels:"] + [g.metadata() for g in gens] + [
        "amdhsa.target: amdgcn-amd-amdhsa--gfx950", "amdhsa.version:", "  - 1", "  - 2", "...", ".end_amdgpu_metadata", ""]
    return body + "\n".join(md)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", "--output", required=True)
    ap.add_argument("--stamps", action="store_true", help="diagnostic build: job-timeline stamps into the debug buffer")
    args = ap.parse_args(argv)
    from .check import check
    gens = []
    for dtype in ("bf16", "f16"):
        for causal in (False, True):
            g = Gen(dtype, causal, stamps=args.stamps)
            g.build()
            errs = check(g.prog)
            if errs:
                print(f"{g.name}: {len(errs)} wait-state violations", file=sys.stderr)
                return 1
            gens.append(g)
    if args.stamps:  # timing-only ablations ride in the diagnostic code object
        for nm, abl in ABLATIONS.items():
            g = Gen("bf16", False, name=f"fa2_fwd_a64_bf16_n_{nm}", stamps=True, abl=abl)
            g.build()
            gens.append(g)
    with open(args.output, "w") as f:
        f.write(module_text(gens))
    return 0


if __name__ == "__main__":
    sys.exit(main())
