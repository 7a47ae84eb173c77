#!/usr/bin/env python3
"""Generator of the gfx950 assembly kernels `fa2_fwd_a16_<dtype>_<c|n>` -- FA-2 forward, d = 128, f16 / bf16: the a64 structure
(fa2_a64_gen.py: 4 waves x 64 query rows, one wave per SIMD with all 512 registers, persistent grid, continuous tile stream,
LDS-DMA staging, modulo-scheduled softmax in the MFMA gaps) on the OTHER matrix shape, v_mfma_f32_16x16x32.

Why: under the a64 kernel's filler load the chip holds 1.74-1.81 GHz on the 32x32x16 stream and 2.09-2.20 GHz on the 16x16x32
stream of the same FLOPs (profiles/r03/powerprobe_mfma_shapes.txt; MI355X_MICROARCH.md 'DVFS give-back' item 7) -- the 16x16 form
costs 12 % more cycles (an MFMA holds the vector issue port 8 of its 16 cycles) and still finishes 6-8 % earlier.

Reference arithmetic: /root/reference/src/flash_attention_kernels.py:84-108, as fa2_a64_gen.py.

What changes against fa2_a64_gen.py is the LAYOUT, not the schedule.  A wave's 64 query rows are four 16-row blocks qb16; a
64-key tile is four 16-key blocks kb16 = 2 kk + k' (kk: the 32-key half = one k-step of P.V); lane l = 16 g + li.
  * S^T(kb16, qb16) = K(kb16, ks) . Q(qb16, ks)^T summed over the four 32-column steps ks: A = K rows (lane: key li, columns
    32 ks + 8 g ..), B = Q rows (lane: query li), D: lane holds query li and keys 16 kb16 + 4 g + r, r = 0..3;
  * a score GROUP gi = 2 qh + kk (qh = qb16 >> 1: the 32-row "query block" of the a64 plan) is 16 registers: + 8 (qb16 & 1)
    + 4 k' + r.  Packed in place, registers + 8 (qb16 & 1) .. + 3 are the B operand P^T(qb16, kk) of the second product:
    k index 8 g + j <-> key 32 kk + 16 (j >> 2) + 4 g + (j & 3) -- no lane exchange, as in a64;
  * O^T(db16, qb16) += V^T(db16, kk) . P^T(qb16, kk): the V^T fragment of a lane (d = 16 db16 + li) is two transposing LDS reads
    of four keys each (32 kk + 16 u + 4 g + 0..3);
  * one MFMA "slot" of the a64 schedule (32 cycles) is a PAIR of 16x16x32 MFMAs that share their A operand (the two 16-row
    blocks of a query block); the slot's fillers are split between the two.  Slot counts, the softmax plan, the seam and the
    job stream are a64's;
  * row sums: v_mfma_f32_16x16x32 with an all-ones A operand -- every row of its D is the 32-key sum of the lane's own query.
LDS image of a K / V tile: two column halves of 64 rows x 128 bytes, the eight 16-byte chunks of a half row XOR-swizzled by
(row & 6): conflict-free for the ds_read_b128 row reads of the 16x16x32 A operand and for the transposing reads
(tests/test_asm_a16.py checks both against the bank model of MI355X_MICROARCH.md).
"""
from __future__ import annotations

import argparse
import sys

from .isa import A, EXEC, I, Inst, Label, M0, Reg, S, V, VCC, comment, label, waitcnt

# ------------------------------------------------------------------------------------------------- register map
# arch VGPRs
SBUF = (0, 64)            # two score buffers of 64 registers: group gi = 2 qh + kk at + 16 gi, inside + 8 (qb16 & 1) + 4 (kb16 & 1) + r
KF = 128                  # K fragments of the next tile: (kb16, ks) at KF + 4 (4 kb16 + ks)
V_KR = (192, 193)         # K row-read lane bases (even / odd 32-column step)
V_VR = (194, 195, 196, 197)   # V transposed-read lane bases (db16 & 3), the V ring's LDS offset included
V_DKO, V_DKO2 = 198, 199  # LDS-DMA per-lane source offsets of K rows (column half 0 / 1: + 128 bytes)
V_DVO, V_DVO2 = 200, 201  # ... of V rows
V_DQ = 202                # ... of Q rows (the second half rides in the scalar offset)
V_QR = (203, 204)         # Q row-read lane bases in the wave's slice (even / odd 32-column step)
V_MC = (205, 206, 207, 208)   # running row maximum in the exp2 domain (c * max), per 16-row block
V_CO = (209, 210, 211, 212)   # rescale coefficient per 16-row block
V_MX = (213, 214, 215, 216)   # row-max chain per 16-row block (one chain over both key halves)
V_MSV = (217, 218, 219, 220)  # running maximum of the finished job
V_LANE = 221
V_T = tuple(range(222, 232))  # temporaries (V_T[0], V_T[2], ... even; V_T[2] = v224 is 4-aligned: a zero MFMA operand in the epilogue)
V_ONES = 232              # 4 registers: the all-ones A operand of the row-sum MFMA
V_LACC = (236, 240, 244, 248)  # row-sum accumulators per 16-row block (4 registers each, all equal: the lane's own row sum)
V_NINF = 252              # causal / ragged: -inf
NINF = V(V_NINF)
V_IMH = 253               # causal: li - 4 g (query row inside a 16-row block minus the lane group's key offset)
V_PM = (254, 255)         # causal: AND masks of the two packed P registers of a diagonal 16 x 16 block
# during the epilogue (the row sums have been read, the accumulators are zeroed at its end) V_LACC[0..] hold its lane constants:
V_EW, V_ER, V_EO, V_L2 = 236, 237, 238, 239   # LDS write base, LDS read base, global store lane offset, L store lane offset


# AGPRs
def A_O(qb16, db16):
    return A((qb16 * 8 + db16) * 4, 4)


def A_Q(qb16, ks):
    return A(128 + (qb16 * 4 + ks) * 4, 4)


def A_K(kb16, ks):
    # the K tile lives in ARCH VGPRs v[128:191] (a64: LDS reads into accumulator registers while MFMAs write accumulators cost
    # 470 cycles per tile)
    return V(KF + (kb16 * 4 + ks) * 4, 4)


def V_F(kk, db16):
    # V^T fragments in a[192:255]: read in phase A, whose MFMAs (QK^T) write arch VGPRs
    return A(192 + 4 * (8 * kk + db16), 4)


def S_BLK(Y, kb16, qb16):
    """accumulator of S^T(kb16, qb16) in score buffer Y"""
    return V(Y + 16 * (2 * (qb16 >> 1) + (kb16 >> 1)) + 8 * (qb16 & 1) + 4 * (kb16 & 1), 4)


def P_OP(X, qb16, kk):
    """the packed P^T(qb16, kk): B operand of the second product (the first four registers of the group's 8-register half)"""
    return V(X + 16 * (2 * (qb16 >> 1) + kk) + 8 * (qb16 & 1), 4)


# SGPRs.  s4..s47 hold the kernel arguments (loaded once).
S_KARG = S(0, 2)
S_WGID = S(2)
S_FINAL = S(75)
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
S_KT0 = S(86)                    # (= S_PASS, causal only) non-causal ragged: real keys in the job's last 256 = N - 256 (nq - 1)
S_LG = S(73)                     # decode shifts: lgH | lgG << 8 | lg(G * nunit) << 16 | pow2-mode << 24
S_FIRE = (S(0, 2), S(2, 2))      # per query block: lanes whose row maximum passed the deferral threshold (s0..s3 are free after the set-up)
S_X2 = S(82)

# LDS map (bytes)
KB = (0, 16384)
VBASE = 32768
VB = (0, 16384)                  # relative to VBASE (folded into the V read lane bases; absolute for the DMA)
EPI = 65536                      # + 16384 * wave: the wave's private 64 x 256-byte slice: the next job's Q rows land here by
                                 # LDS-DMA (K-tile image) on their way to a[128:191]; later the job's O rows leave through it
EPI_ROW = 272                    # (= 256 + 16) byte stride of an O row in the slice during the epilogue: 16-byte aligned for the row reads;
                                 # the 8-byte column writes of a 16-lane group are 2-way conflicted (as in a64), the 16-byte row reads conflict-free
LDS_TOTAL = 131072
V_ST_LAST, V_ST_ACC = 252, 253   # diagnostic (stamps) builds, non-causal only: last stamp (low word), accumulators [3] (253..255)


def tile_addr(row, c16):
    """byte offset of 16-byte chunk c16 (0..15) of row `row` (0..63) inside a K / V / Q tile image: two column halves of
    64 rows x 128 bytes, the eight chunks of a half row XOR-swizzled by (row & 6)"""
    return 8192 * (c16 >> 3) + 128 * row + 16 * ((c16 & 7) ^ (row & 6))

KARG_SIZE = 192
NSLOT = 24


class Gen:
    def __init__(self, dtype="bf16", causal=False, name=None, stamps=False, abl=(), ring=(2, 3, 2), vread_double=4, ragged=False,
                 caps=(5, 24), split=True, soft=None):
        assert dtype in ("bf16", "f16")
        self.dtype = dtype
        self.causal = causal
        self.name = name or f"fa2_fwd_a16_{dtype}_{'c' if causal else 'n'}{'r' if ragged else ''}"
        self.atmp = 0          # (ragged) which of the two address temporaries the next buffer operation takes
        self.atmp_regs = (V_T[8], V_T[9])
        self.prog: list[Inst] = []
        self.uid = 0
        self.mfma = "v_mfma_f32_16x16x32_" + dtype
        self.cvt = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
        self.ool: list[list[Inst]] = []  # out-of-line blocks (rare paths), appended after the main body
        self.soft = soft       # (limit, window) of the softmax plan's soft per-gap issue limit, or None (tile_plan.place)
        self.caps = caps       # fillers / issue cycles a gap behind a 32x32x16 MFMA may carry in the softmax plan
        # causal row map "split": wave w owns the 32-row blocks w (qb 0) and w + 4 (qb 1) of the job's 256 rows instead of
        # 2 w and 2 w + 1.  Diagonal tile j (key blocks 2 j, 2 j + 1) is then hidden from query block 0 of EVERY wave for
        # j >= 2 and fully visible to query block 1 for j < 2: the job's last steps run on one query block (half the MFMAs)
        # for all four waves instead of on both for a shrinking set of waves -- see build()
        assert split, "the a16 kernels use the split row map only"
        self.split = bool(split) and causal
        assert self.split or not causal, "a16: the causal kernels use the split row map (the contiguous map's lean bodies know two maxima per wave, not four)"
        self.cls = None        # split seam bodies: "low" (waves 0, 1) / "high" (waves 2, 3) while their code is generated
        self.ragged = ragged   # N is not a multiple of 256: range-checked descriptors, every offset in the VGPR operand, masked key tail
        assert not (ragged and stamps), "the ragged kernels use the stamps' temporaries as address registers"
        # (the accumulating stamps use the causal kernels' mask registers: causal diagnostic builds carry the plain job-timeline
        # stamps only -- "noacc")
        if stamps and causal:
            abl = tuple(abl) + ("noacc",)
        self.vread_double = vread_double   # phase-A gaps that carry two V transposed reads (the last read sits in gap 31 - this)
        self.abl = set(abl)    # timing-only ablations of the steady loop (diagnostic builds; results wrong by construction)
        self.R, self.dk, self.dv = ring   # ring depth; K(t + dk) and V(t + dv) are streamed in phase B(t): dk <= R + 1, dv <= R
        assert 3 <= self.dk <= min(self.R + 1, 4) and 2 <= self.dv <= self.R and 4 % self.R == 0
        self.vm = 8 * min(self.dk - 3, self.dv - 2)  # DMA pieces that may stay in flight across the mid-step barrier
        self._cache = {}
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

    ASYNC_PAIRS = (S(90, 2), S(92, 2), S(94, 2), S(0, 2), S(2, 2))   # S_T[2..7], S_FIRE: idle in the epilogue

    def stamp_async(self, k):
        """diagnostic builds only: s_memtime into spare pair k WITHOUT a wait (the epilogue's LDS queue is not drained; its
        counted lgkmcnt waits may be satisfied early by the returning s_memtime: timing-only)"""
        return [I("s_memtime", self.ASYNC_PAIRS[k])] if self.stamps else []

    def stamp_async_flush(self, slots):
        if not self.stamps:
            return []
        out = [waitcnt(lgkmcnt=0), I("v_mov_b32", V(V_T[7]), 0)]
        v = V(V_T[8], 2)
        for k, slot in enumerate(slots):
            t = self.ASYNC_PAIRS[k]
            out += [I("v_mov_b32", v.sub(0), t.sub(0)), I("v_mov_b32", v.sub(1), t.sub(1)),
                    I("global_store_dwordx2", V(V_T[7]), v, S_DBG, offset=8 * slot), I("s_nop", 7)]
        return out

    def stamp_acc(self, k):
        """diagnostic builds only: acc[k] += cycles since the previous stamp_acc (its s_waitcnt drains the LDS queue as well: the
        per-phase shares cost cycles of their own -- the "lite" kernels carry the job-level stamps only)"""
        if not self.stamps or "lite" in self.abl or "noacc" in self.abl:
            return []
        t = S(S_T[0].idx, 2)
        tmp = V(V_T[9])
        acc = [I("v_sub_u32", tmp, t.sub(0), V(V_ST_LAST)), I("v_add_u32", V(V_ST_ACC + k), V(V_ST_ACC + k), tmp)] if k < 3 else []
        return [I("s_memtime", t), waitcnt(lgkmcnt=0)] + acc + [I("v_mov_b32", V(V_ST_LAST), t.sub(0))]

    def stamp_job(self, k):
        """lite diagnostic builds: acc[k] += cycles since the previous stamp_job, summed over ALL jobs of the workgroup
        (0 steady loops, 1 seam bodies, 2 epilogues + job bookkeeping, 3 the pipeline fill of the first job)"""
        if not self.stamps or "lite" not in self.abl or "noacc" in self.abl:
            return []
        t = S(S_T[0].idx, 2)
        tmp = V(V_T[9])
        acc = [I("v_sub_u32", tmp, t.sub(0), V(V_ST_LAST)), I("v_add_u32", V(V_ST_ACC + k), V(V_ST_ACC + k), tmp)] if k < 3 else []
        return [I("s_memtime", t), waitcnt(lgkmcnt=0)] + acc + [I("v_mov_b32", V(V_ST_LAST), t.sub(0))]

    def stamp_job_flush(self):
        if not self.stamps or "lite" not in self.abl or "noacc" in self.abl:
            return []
        out = [I("v_mov_b32", V(V_T[7]), 0)]
        for k, slot in enumerate((10, 11, 12)):
            out += [I("global_store_dword", V(V_T[7]), V(V_ST_ACC + k), S_DBG, offset=8 * slot)]
        return out

    def stamp_flush(self):
        if not self.stamps or "lite" in self.abl or "noacc" in self.abl:
            return []
        out = [I("v_mov_b32", V(V_T[7]), 0)]
        for k in range(3):
            out += [I("global_store_dword", V(V_T[7]), V(V_ST_ACC + k), S_DBG, offset=8 * (10 + k)),
                    I("v_mov_b32", V(V_ST_ACC + k), 0)]
        return out

    def udiv(self, q: Reg, r: Reg | None, n: Reg, d: Reg, vt=None):
        """q = n / d, r = n % d for wave-uniform 32-bit values < 2^22 (float reciprocal + one correction each way)"""
        t0, t1 = vt if vt is not None else (V(V_T[0]), V(V_T[1]))
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

    def make_desc(self, rs: Reg, base: Reg, sb: Reg, sh: Reg, b: Reg, hh: Reg, sn=None):
        """raw buffer descriptor of the (b, hh) slice of a tensor: base + b * sb + hh * sh.  N a multiple of 256: the range check
        is not used (soffset is unchecked anyway; every address the kernel forms lies inside the tensor).  Ragged kernels:
        num_records = (N - 1) * sn + 256 bytes of rows (sn: the row stride register; an int: bytes per row of L) -- loads of
        rows past N come back as zeros, stores to them are dropped; those kernels keep every offset in the VGPR operand"""
        tmp = S(S_T[0].idx, 2)
        out = ([I("s_mov_b64", tmp, base)] + self.mad64(tmp, b, sb) + self.mad64(tmp, hh, sh) +
               [I("s_mov_b32", rs.sub(0), tmp.sub(0)), I("s_and_b32", rs.sub(1), tmp.sub(1), 0xFFFF), I("s_mov_b32", rs.sub(3), 0x00020000)])
        if not self.ragged:
            return out + [I("s_mov_b32", rs.sub(2), 0x7FFFFFF0)]
        assert sn is not None
        if isinstance(sn, int):
            return out + [I("s_mul_i32", rs.sub(2), S_N, sn)]
        return out + [I("s_sub_u32", S_T[6], S_N, 1), I("s_mul_i32", S_T[6], S_T[6], sn), I("s_add_u32", rs.sub(2), S_T[6], 256)]

    def buf_op(self, op, data, voff: Reg, rsrc: Reg, soff, **mods):
        """a buffer operation at byte offset voff (per lane) + soff (scalar).  The scalar operand of the instruction is not
        range-checked: the ragged kernels add it into an address temporary first (two, taken alternately: a set-up may run
        ahead of the previous piece's load by one gap).  Returns (set-up instructions, the memory instruction)"""
        ops = (lambda v, so: (data, v, rsrc, so) if data is not None else (v, rsrc, so))
        if not self.ragged:
            return [], I(op, *ops(voff, soff), offen=1, **mods)
        tmp = V(self.atmp_regs[self.atmp])
        self.atmp ^= 1
        return [I("v_add_u32", tmp, soff, voff)], I(op, *ops(tmp, 0), offen=1, **mods)

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
          I("s_lshl_b32", S_LDSW, S_WAVE, 10))
        # lane = 16 g + li
        # ---- K row-read bases (tile_addr): row 16 kb16 + li, chunk 4 ks + g:  128 li + 16 ((4 e + g) ^ (li & 6)), e = ks & 1
        e(comment("K row-read lane bases"),
          I("v_and_b32", t0, 15, lane),                    # li
          I("v_lshrrev_b32", t1, 4, lane),                 # g
          I("v_and_b32", t2, 6, t0),                       # li & 6
          I("v_xor_b32", t2, t2, t1),                      # g ^ (li & 6)            (e = 0)
          I("v_lshlrev_b32", t3, 7, t0),                   # 128 li
          I("v_lshl_add_u32", V(V_KR[0]), t2, 4, t3),
          I("v_xor_b32", t2, 4, t2),                       # (4 + g) ^ (li & 6)      (e = 1)
          I("v_lshl_add_u32", V(V_KR[1]), t2, 4, t3))
        # ---- V transposed-read bases: inside a 16-lane group lane li = 4 q + p supplies key row 4 g + q, columns 4 p .. 4 p + 3 of
        #      the 16-column block db16:  VBASE + 128 (4 g + q) + 16 ((2 b + (p >> 1)) ^ (4 (g & 1) + (q & 2))) + 8 (p & 1), b = db16 & 3
        e(comment("V transposed-read lane bases"),
          I("v_bfe_u32", t0, lane, 2, 2),                  # q
          I("v_lshrrev_b32", t1, 4, lane),                 # g
          I("v_lshl_add_u32", t2, t1, 2, t0),              # 4 g + q
          I("v_lshlrev_b32", t2, 7, t2),                   # 128 (4 g + q)
          I("v_and_b32", t3, 1, lane), I("v_lshl_add_u32", t2, t3, 3, t2),   # + 8 (p & 1)
          I("v_add_u32", t2, VBASE, t2),
          I("v_and_b32", t1, 1, t1), I("v_lshlrev_b32", t1, 2, t1),          # 4 (g & 1)
          I("v_and_b32", t0, 2, t0), I("v_or_b32", t1, t1, t0),              # + (q & 2)
          I("v_bfe_u32", t0, lane, 1, 1), I("v_xor_b32", t1, t1, t0))        # ^ (p >> 1)          (b = 0)
        for b in range(4):
            e(I("v_xor_b32", t0, 2 * b, t1), I("v_lshl_add_u32", V(V_VR[b]), t0, 4, t2))
        # ---- LDS-DMA lane source offsets: a piece is 8 rows x 128 bytes; lane i lands at row i >> 3, chunk position i & 7 of the
        #      image, which holds source chunk (i & 7) ^ ((i >> 3) & 6):  offset = (i >> 3) * stride + 16 ((i & 7) ^ ((i >> 3) & 6))
        e(comment("LDS-DMA per-lane source offsets"),
          I("v_lshrrev_b32", t1, 3, lane),                 # row_in
          I("v_and_b32", t0, 6, t1), I("v_and_b32", t2, 7, lane), I("v_xor_b32", t0, t0, t2),
          I("v_lshlrev_b32", t0, 4, t0))                   # 16 chunk
        e(waitcnt(lgkmcnt=0, comment="kernel arguments are in"))
        e(I("v_mul_lo_u32", t2, t1, S_KSN), I("v_add_u32", V(V_DKO), t2, t0), I("v_add_u32", V(V_DKO2), 128, V(V_DKO)),
          I("v_mul_lo_u32", t2, t1, S_VSN), I("v_add_u32", V(V_DVO), t2, t0), I("v_add_u32", V(V_DVO2), 128, V(V_DVO)),
          I("v_mul_lo_u32", t2, t1, S_QSN), I("v_add_u32", V(V_DQ), t2, t0))
        e(comment("Q row-read bases in the wave's slice"),
          I("s_lshl_b32", S_T[0], S_WAVE, 14), I("s_add_u32", S_T[0], S_T[0], EPI),
          I("v_add_u32", V(V_QR[0]), S_T[0], V(V_KR[0])), I("v_add_u32", V(V_QR[1]), S_T[0], V(V_KR[1])))
        if self.stamps:
            e(I("s_lshl_b32", S_T[0], S_WGID, 2), I("s_add_u32", S_T[0], S_T[0], S_WAVE), I("s_mul_i32", S_T[0], S_T[0], 8 * NSLOT),
              I("s_add_u32", S_DBG.sub(0), S_DBG.sub(0), S_T[0]), I("s_addc_u32", S_DBG.sub(1), S_DBG.sub(1), 0))
            e(self.stamp(6, real=True), self.stamp(8))
            if "noacc" not in self.abl:
                e([I("v_mov_b32", V(V_ST_ACC + k), 0) for k in range(3)], I("v_mov_b32", V(V_ST_LAST), 0))
        # ---- row sums on the matrix pipe: the all-ones operand, accumulators; rescale factors
        one2 = 0x3F803F80 if self.dtype == "bf16" else 0x3C003C00
        e(comment("row-sum MFMA operand, accumulators, rescale factors"))
        e([I("v_mov_b32", V(V_ONES + k), one2) for k in range(4)])
        e([I("v_mov_b32", V(V_LACC[q] + k), 0) for q in range(4) for k in range(4)])
        e([I("v_mov_b32", V(V_CO[q]), 1.0) for q in range(4)])
        # ---- scalar constants
        e(I("s_lshl_b32", S_K32, S_KSN, 5), I("s_lshl_b32", S_V32, S_VSN, 5),
          I("s_lshl_b32", S_K64, S_KSN, 6), I("s_lshl_b32", S_V64, S_VSN, 6),
          I("s_lshl_b32", S_T[0], S_WAVE, 3), I("s_mul_i32", S_KW, S_T[0], S_KSN), I("s_mul_i32", S_VW, S_T[0], S_VSN),
          I("s_mov_b32", S_FLAG, 0), I("s_mov_b32", S_PASS, 0), I("s_mov_b32", S_FINAL, 0),
          I("s_mov_b32", S_JOB, S_WGID))
        if self.ragged and not self.causal:
            e(comment("ragged, non-causal: real keys in a job's last 256; -inf"),
              I("s_sub_u32", S_T[0], S_NQ, 1), I("s_lshl_b32", S_T[0], S_T[0], 8), I("s_sub_u32", S_KT0, S_N, S_T[0]),
              I("v_mov_b32", NINF, float("-inf")))
        if self.causal:
            e(comment("causal: lane constants of the diagonal mask"),
              I("v_and_b32", t0, 15, lane), I("v_lshrrev_b32", t3, 4, lane), I("v_lshlrev_b32", t3, 2, t3),
              I("v_sub_u32", V(V_IMH), t0, t3),   # li - 4 g
              I("v_mov_b32", NINF, float("-inf")))
            e(I("v_mov_b32", V(V_T[4]), 0xFFFF0000), I("v_mov_b32", V(V_T[5]), 0x0000FFFF))
            # packed P register jj (0, 1) of a diagonal 16 x 16 block holds keys k0 = 2 jj + 4 g and k0 + 1 of query li:
            # keep both (k0 + 1 <= li), the low one only (k0 == li) or none
            for jj in range(2):
                k0 = 2 * jj
                e(I("v_cmp_ge_i32", VCC, V(V_IMH), k0 + 1), I("v_cndmask_b32", t1, 0, V(V_T[4]), VCC),
                  I("v_cmp_ge_i32", VCC, V(V_IMH), k0), I("v_cndmask_b32", t2, 0, V(V_T[5]), VCC),
                  I("v_or_b32", V(V_PM[jj]), t1, t2))

    def k_epi_consts(self):
        """lane constants of the epilogue, formed at its start in the (read-out) row-sum accumulators: the slice holds a query
        block's 32 O rows at a stride of EPI_ROW bytes.  Write base (row li, + 8 g): slice + EPI_ROW li + 8 g;  read-back
        (a = lane >> 4, ec = lane & 15): slice + EPI_ROW a + 16 ec (+ 4 EPI_ROW k: row 4 k + a); store offset a * os_n + 16 ec"""
        lane = V(V_LANE)
        t0, t1, t2 = V(V_LACC[1]), V(V_LACC[1] + 1), V(V_LACC[1] + 2)
        return [I("s_lshl_b32", S_X2, S_WAVE, 14), I("s_add_u32", S_X2, S_X2, EPI),
                I("v_and_b32", t0, 15, lane), I("v_lshrrev_b32", t1, 4, lane),
                I("v_mul_u32_u24", t2, EPI_ROW, t0), I("v_lshl_add_u32", t2, t1, 3, t2), I("v_add_u32", V(V_EW), S_X2, t2),
                I("v_mul_u32_u24", t2, EPI_ROW, t1), I("v_lshl_add_u32", t2, t0, 4, t2), I("v_add_u32", V(V_ER), S_X2, t2),
                I("v_mul_lo_u32", t2, t1, S_OSN), I("v_lshl_add_u32", V(V_EO), t0, 4, t2),
                I("v_lshlrev_b32", V(V_L2), 1, t0)]

    # ------------------------------------------------------------------ job decode: S_JOB (+ S_PASS) -> S_NB, S_NHH, S_NQI, S_NNT
    def k_decode_next(self, vt=None):
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
        self.udiv(t[2], t[3], t[0], t[1], vt)       # batch, r
        self.udiv(S_UNIT, t[4], t[3], S_G, vt)      # unit = r / G, r % G
        e(I("s_mul_i32", t[2], t[2], S_G), I("s_add_u32", t[2], t[2], t[4]), I("s_lshl_b32", t[2], t[2], 3),
          I("s_and_b32", t[0], S_JOB, 7), I("s_add_u32", bh, t[2], t[0]), I("s_branch", Label(l_done)))
        e(label(l_else))
        self.udiv(bh, S_UNIT, S_JOB, S_NUNIT, vt)
        e(label(l_done))
        self.udiv(S_NB, S_NHH, bh, S_H, vt)
        e(label(l_qi))
        if self.causal:
            # unit u, pass 0: qi = nq - 1 - u (heavy), pass 1: qi = u;  tiles = 4 (qi + 1)
            l_p1, l_pd = self.lab("pass1"), self.lab("passd")
            e(I("s_cmp_lg_u32", S_PASS, 0), I("s_cbranch_scc1", Label(l_p1)),
              I("s_sub_u32", S_NQI, S_NQ, 1), I("s_sub_u32", S_NQI, S_NQI, S_UNIT), I("s_branch", Label(l_pd)),
              label(l_p1), I("s_mov_b32", S_NQI, S_UNIT), label(l_pd),
              I("s_add_u32", t[0], S_NQI, 1), I("s_lshl_b32", S_NNT, t[0], 2))
        else:
            e(I("s_mov_b32", S_NQI, S_UNIT), I("s_lshl_b32", S_NNT, S_NQ, 2))      # (4 tiles per 256 rows, N rounded up)

    def k_advance(self, vt=None):
        """S_JOB / S_PASS -> the job after the most recently decoded one, decoded into the next-job registers;
        S_FINAL = 1 if there is none (the next-job registers then repeat the current job).  vt: two VGPR temporaries for the
        divisions of the generic decode (default V_T[0], V_T[1])"""
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
        self.k_decode_next(vt)
        l_done = self.lab("adv_done")
        e(I("s_branch", Label(l_done)), label(l_fin),
          comment("no further job: the seam streams the current job's first tiles again (results discarded)"),
          I("s_mov_b32", S_FINAL, 1), I("s_mov_b32", S_NB, S_B), I("s_mov_b32", S_NHH, S_HH), I("s_mov_b32", S_NQI, S_QI),
          I("s_mov_b32", S_NNT, S_NT), label(l_done))

    def k_promote(self):
        """next job -> current job"""
        self.e(I("s_mov_b32", S_B, S_NB), I("s_mov_b32", S_HH, S_NHH), I("s_mov_b32", S_QI, S_NQI), I("s_mov_b32", S_NT, S_NNT),
               # query rows of this wave: qrow[qb] = 256 qi + 64 wave + 32 qb  (split row map: 256 qi + 32 wave + 128 qb)
               I("s_lshl_b32", S_T[0], S_QI, 8), I("s_lshl_b32", S_T[1], S_WAVE, 5 if self.split else 6),
               I("s_add_u32", S_QROW[0], S_T[0], S_T[1]),
               I("s_add_u32", S_QROW[1], S_QROW[0], 128 if self.split else 32))

    # ------------------------------------------------------------------ LDS-DMA
    def dma_piece(self, which, piece, buf):
        """one 1-KiB LDS-DMA piece of the next K / V tile into ring buffer `buf`.  piece j: rows 8 R .. 8 R + 7 with
        R = wave (j < 2) or wave + 4, 128-byte column half j & 1 (the +128 rides in the second lane-offset register); it lands at
        8192 (j & 1) + 1024 R of the tile image (tile_addr)"""
        if which == "k":
            rsrc, vl, vl2, base, s32, lds0 = S_KRS, V(V_DKO), V(V_DKO2), S_KDMA, S_K32, KB[buf]
        else:
            rsrc, vl, vl2, base, s32, lds0 = S_VRS, V(V_DVO), V(V_DVO2), S_VDMA, S_V32, VBASE + VB[buf]
        out = []
        so = base
        if piece >= 2:
            so = S_T[5]
            out.append(I("s_add_u32", so, base, s32))
        out.append(I("s_add_u32", M0, S_LDSW, lds0 + (0, 8192, 4096, 12288)[piece]))
        pre, ld = self.buf_op("buffer_load_dwordx4", None, vl2 if piece & 1 else vl, rsrc, so, lds=1, tag=f"dma {which}{piece}")
        return out + pre + [I("s_nop", 0), ld]

    def dma_tile(self, which, buf):
        out = []
        for j in range(4):
            out += self.dma_piece(which, j, buf)
        out.append(I("s_add_u32", S_KDMA, S_KDMA, S_K64) if which == "k" else I("s_add_u32", S_VDMA, S_VDMA, S_V64))
        return out

    def q_stage(self, b: Reg, hh: Reg, qi: Reg):
        """Q rows of job (b, hh, qi) of this wave -> the wave's LDS slice by LDS-DMA, in the tile image (16 pieces of 8 rows x
        128 bytes: coalesced, ~25 cycles of issue each).  Returns (descriptor / offset setup, [pieces])"""
        setup = self.make_desc(S_SQ, S_Q, S_QSB, S_QSH, b, hh, S_QSN)
        setup += [I("s_lshl_b32", S_T[0], qi, 8), I("s_lshl_b32", S_T[1], S_WAVE, 5 if self.split else 6), I("s_add_u32", S_T[0], S_T[0], S_T[1]),
                  I("s_mul_i32", S_T[2], S_T[0], S_QSN),                # byte offset of the wave's first row
                  I("s_lshl_b32", S_T[3], S_QSN, 3),                    # 8 rows
                  I("s_lshl_b32", S_T[4], S_WAVE, 14), I("s_add_u32", S_T[4], S_T[4], EPI)]
        pieces = []   # (scalar set-up, load): the set-up ends one MFMA gap, the load opens the next (the MFMA between them is the
        # wait state the M0 write needs), like the K / V pieces of phase B
        for R in range(8):
            for half in range(2):
                pc = []
                so = S_T[2]
                if self.split and R == 4 and not half:
                    # split row map: the second query block starts 128 rows behind the first (96 = 12 x 8 rows further on)
                    pc += [I("s_mul_i32", S_X2, S_T[3], 12), I("s_add_u32", S_T[2], S_T[2], S_X2)]
                if half:
                    so = S_T[5]
                    pc.append(I("s_add_u32", so, S_T[2], 128))
                    if R < 7:
                        pc.append(I("s_add_u32", S_T[2], S_T[2], S_T[3]))
                pc.append(I("s_add_u32", M0, S_T[4], 1024 * R + 8192 * half))
                pre, ld = self.buf_op("buffer_load_dwordx4", None, V(V_DQ), S_SQ, so, lds=1, tag=f"qdma R{R} h{half}")
                pieces.append((pc + pre, ld))
        return setup, pieces

    def q_reads(self):
        """the staged Q rows -> a[128:191] (fragment (qb16, ks) = rows 16 qb16 + li, 16-byte chunk 4 ks + g, as a K row read)"""
        out = []
        for qb16 in range(4):
            for ks in range(4):
                out.append(I("ds_read_b128", A_Q(qb16, ks), V(V_QR[ks & 1]), offset=2048 * qb16 + 8192 * (ks >> 1), tag=f"qread qb{qb16} ks{ks}"))
        return out

    # ------------------------------------------------------------------ the two phases
    def qk_mfmas(self, Y, cinit=None, qbs=(0, 1)):
        """S^T(next) into score buffer Y as 32 SLOTS of two MFMAs: slot (qh, kb16, ks) = the blocks qb16 = 2 qh, 2 qh + 1 against
        K(kb16, ks) -- consecutive MFMAs share their A operand; an accumulator is revisited every second MFMA.
        qbs: the query blocks qh whose scores are computed -- the others' slots are None in the returned list (emit_phase)"""
        assert cinit is None
        out = []
        for qh in range(2):
            for kb16 in range(4):
                for ks in range(4):
                    pair = []
                    for qb16 in (2 * qh, 2 * qh + 1):
                        d = S_BLK(Y, kb16, qb16)
                        pair.append(I(self.mfma, d, A_K(kb16, ks), A_Q(qb16, ks), d if ks else 0, tag=f"qk qb{qb16} kb{kb16} ks{ks}"))
                    out.append(pair if qh in qbs else None)
        return out

    def v_reads(self, buf):
        """32 transposed reads of the V tile in VB[buf]: fragment (kk, db16) <- u = 0, 1 (keys 32 kk + 16 u + 4 g + 0..3)"""
        out = []
        for kk in range(2):
            for db16 in range(8):
                f = V_F(kk, db16)
                for u in range(2):
                    imm = VB[buf] + 8192 * (db16 >> 2) + 4096 * kk + 2048 * u
                    out.append(I("ds_read_b64_tr_b16", f.sub(2 * u, 2), V(V_VR[db16 & 3]), offset=imm, tag=f"vread kk{kk} db{db16}"))
        return out

    def k_reads(self, buf):
        out = []
        for kb16 in range(4):
            for ks in range(4):
                imm = KB[buf] + 2048 * kb16 + 8192 * (ks >> 1)
                out.append(I("ds_read_b128", A_K(kb16, ks), V(V_KR[ks & 1]), offset=imm, tag=f"kread kb{kb16} ks{ks}"))
        return out

    # ------------------------------------------------------------------ the softmax of one tile as a list of placed operations
    # Time line of a tile, in MFMA gaps (tau): [0, 32) = the phase A that computes its scores (QK^T chains g = 0..3, eight
    # MFMAs each), [32, 32 + NB) = the following phase B (NB = 40 MFMAs: P.V of the previous tile plus its row sums),
    # [32 + NB, 64 + NB) = the next phase A, at whose end P must be packed (its own P.V follows).  In steady state the
    # physical gap (tau mod PERIOD) therefore carries operations of two tiles: a modulo reservation table keeps every gap
    # within what hides beside an MFMA (measured, scripts/probes/mb_run + asm/microbench.py: at most five fillers per gap,
    # issue costs v_exp 8 / three-operand VALU 5 / two-operand 4 summing to <= 24; LDS reads first in their gap).
    NB = 40
    PERIOD = 72
    T_END = 104
    LAZY_TAU = {"ms0": 28, "ms1": 29, "mr": 99, "pm": 100}   # tau of the lazy-masking operations of a diagonal tile (mask_lazy)

    def lazy_tau(self):
        """(the split row map has no 'ms' / 'mr'; its packed-P masking sits behind the last pack of the plan, which ends two gaps
        later there: tile_plan, gap2)"""
        return dict(self.LAZY_TAU, pm=103) if self.split else self.LAZY_TAU

    def tile_plan(self, init=False, lean=False):
        """placement of the per-tile softmax operations: returns [(tau, kind, payload)] sorted by tau.
        kinds: 'mx' (g, j)  'dec' (qb, part)  'f' e  'e' e  'cv' (g, j)"""
        key = "plan"
        if key in self._cache:
            return self._cache[key]
        P = self.PERIOD
        slots = [0.0] * P
        cost = [0.0] * P
        nexp = [0] * P
        cap_s = [float(self.caps[0])] * P
        cap_c = [float(self.caps[1])] * P
        # phase B: the gap behind a 16x16x32 row-sum MFMA is half as long
        for b in self.b_short_gaps():
            cap_s[32 + b], cap_c[32 + b] = 2.0, 8.0
        # pre-reserved: the V transposed reads of phase A; K reads, DMA pieces and their scalar set-up in phase B
        for k in range(32):
            slots[self.a_vread_gap(k)] += 1
            cost[self.a_vread_gap(k)] += 2
        for b, n in self.b_reserved().items():
            slots[32 + b] += n
            cost[32 + b] += 3 * n
        placed = []

        def place(earliest, c, kind, payload, is_exp=False):
            t = int(earliest)
            # a pair of 16x16x32 MFMAs hides ~16 issue cycles of fillers, not the 24 a 32x32x16 MFMA does: prefer, within a few
            # gaps of the earliest one, a gap that stays under the soft limit (the hard capacities still bound every gap)
            if self.soft is not None:
                lim, win = self.soft
                for tt in range(t, t + win + 1):
                    g = tt % P
                    if tt < self.T_END and slots[g] + 1 <= cap_s[g] and cost[g] + c <= min(lim, cap_c[g]) and (not is_exp or nexp[g] < 2):
                        slots[g] += 1
                        cost[g] += c
                        nexp[g] += int(is_exp)
                        placed.append((tt, kind, payload))
                        return tt
            while True:
                assert t < self.T_END + 40, (kind, payload)
                g = t % P
                if slots[g] + 1 <= cap_s[g] and cost[g] + c <= cap_c[g] and (not is_exp or nexp[g] < 2):
                    slots[g] += 1
                    cost[g] += c
                    nexp[g] += int(is_exp)
                    placed.append((t, kind, payload))
                    return t
                t += 1
        # row maxima: the scores of group gi = 2 qh + kk are complete behind slot 16 qh + 8 kk + 7 (key blocks 2 kk, 2 kk + 1 of
        # query block qh); their maxima may start 4 slots later (12 wait states behind the last MFMA), one operation per gap.
        # One chain per 16-row block runs over both key halves: group (qh, 1) continues where (qh, 0) ended
        t_mx = {}
        for g in range(4):
            t = 16 * (g >> 1) + 8 * (g & 1) + 11
            if g & 1:
                t = max(t, t_mx[g - 1])
            for j in range(8):
                t = place(t, 5, "mx", (g, j)) + 1
            t_mx[g] = t
        t_dec = {}
        for qb in range(2):
            t = max(t_mx[2 * qb], t_mx[2 * qb + 1])
            # (the compare and the branch on it sit in different gaps: back to back the branch waits ~30 cycles for the mask.
            # A job's first tile has no decision to take but keeps the slots: the loop body that finishes it is the one that
            # finishes every other tile, so both placements must agree)
            for part in range(5):
                # (the branch two gaps behind its compare: with one MFMA between them the scalar compare waits ~10 cycles for the
                # mask -- microbenchmark mb_cmps_a_scmp_brs against mb_cmps_aaa_scmp_brs -- and query block 0's pair sat in the last
                # gaps in front of the mid-step barrier; now its branch is the first thing behind the barrier)
                # (not where the lazy masking of the contiguous row map / the ragged key tail pins 'mr' in front of the last packs)
                gap2 = part == 4 and "fire_adjacent" not in self.abl and (self.split or not (self.causal or self.ragged))
                t = place(t + (1 if gap2 else 0), (2, 2, 13, 4, 2)[part], "dec", (qb, part)) + 1
            t_dec[qb] = t
        # (causal diagonal tiles handled lazily -- mask_lazy -- add four small operations at LAZY_TAU: gaps of phase A that are
        # nearly empty in every tile, so the plan itself does not reserve anything for them)
        t_d2 = {qb: next(t for t, k, p_ in placed if k == "dec" and p_ == (qb, 2)) for qb in range(2)}
        lz = self.lazy_tau()
        assert self.split or (lz["mr"] - P < lz["ms0"] < t_d2[0] and lz["mr"] - P < lz["ms1"] < t_d2[1])
        assert lz["pm"] < self.T_END
        # s' = s * c - m, exp2, pack -- element order inside a group is the packing order
        for qb in range(2):
            t_f = t_dec[qb]
            last_cv = {}
            for g in (2 * qb, 2 * qb + 1):
                t_e_prev = None
                for r in range(16):
                    e = 16 * g + r
                    tf = place(t_f, 5, "f", e)
                    t_f = tf  # keep the fma stream in order (several per gap allowed)
                    te = place(tf + 1, 8, "e", e, is_exp=True)
                    if r & 1:
                        j = r >> 1
                        tc = max(te, t_e_prev) + 1
                        if j - 1 in last_cv.get(g, {}):
                            tc = max(tc, last_cv[g][j - 1] + 0)
                        tc = place(tc, 5, "cv", (g, j))
                        last_cv.setdefault(g, {})[j] = tc
                    t_e_prev = te
        if self.causal or self.ragged:     # (the kernels that mask lazily)
            assert max(t for t, k, _ in placed if k == "cv") < (lz["pm"] if self.split else min(lz["mr"], lz["pm"])), \
                "a pack operation behind the packed-P masking"
        placed.sort(key=lambda x: x[0])
        assert max(t for t, _, _ in placed) < self.T_END, max(t for t, _, _ in placed)
        self._cache[key] = placed
        return placed

    def a_vread_gap(self, k):
        """phase-A gap of V transposed read k: two per gap at the start, none in the last four -- the wait in front of the barrier
        then finds the youngest read ~130 cycles old instead of just issued"""
        nd = self.vread_double
        return k // 2 if k < 2 * nd else k - nd

    def b_short_gaps(self):
        """indices (0..39) of the phase-B gaps that follow a 16x16x32 row-sum MFMA"""
        return [10 * k + 8 for k in range(4)] + [10 * k + 9 for k in range(4)]

    def b_reserved(self):
        """phase-B gap -> number of pre-reserved fillers (K reads, DMA loads, DMA scalar set-up)"""
        r = {}
        for k in range(16):
            r[self.b_kread_gap(k)] = r.get(self.b_kread_gap(k), 0) + 1
        for k in range(8):
            g = self.b_dma_gap(k)
            r[g] = r.get(g, 0) + 1
            r[g - 1] = r.get(g - 1, 0) + 1
        return r

    def b_kread_gap(self, k):
        if "kfront" in self.abl:     # (experiment) two K reads per gap from the start of the phase
            g = k // 2
            while g in self.b_short_gaps():
                g += 1
            return g
        g = 2 * k
        while g in self.b_short_gaps():
            g += 1
        return g

    def b_dma_gap(self, k):
        """phase-B gap whose FIRST filler is DMA piece k's load; its scalar set-up (soffset, M0) ends the gap before, so the MFMA
        between them is the wait state the M0 write needs.  Distinct, two apart, clear of the short row-sum gaps."""
        return (11, 13, 15, 17, 21, 23, 25, 27)[k]

    def tile_op(self, Sb, kind, payload, init, lazy=None):
        """the instructions of one placed operation, for the tile whose scores live in score buffer Sb.
        lazy = (jd, cond): the tile is diagonal tile jd of its job (if cond holds) and masked lazily (mask_lazy)"""
        if kind in ("ms", "mr", "pm"):
            return self.mask_lazy(Sb, kind, payload, lazy)
        if kind == "mx":
            g, j = payload
            qh, kk = g >> 1, g & 1
            q1, jj = j >> 2, j & 3
            mx = V(V_MX[2 * qh + q1])
            y = lambda r: V(Sb + 16 * g + 8 * q1 + r)     # the block's eight scores of this key half
            if kk:      # (the chain of the first key half goes on)
                return [I("v_max3_f32", mx, mx, y(2 * jj), y(2 * jj + 1), tag=f"max g{g}")]
            if jj == 0:
                return [I("v_max3_f32", mx, y(0), y(1), y(2), tag=f"max g{g}")]
            if jj == 3:
                return [I("v_max_f32", mx, mx, y(7), tag=f"max g{g}")]
            return [I("v_max3_f32", mx, mx, y(2 * jj + 1), y(2 * jj + 2), tag=f"max g{g}")]
        if kind == "dec":
            qh, part = payload
            qs = (2 * qh, 2 * qh + 1)
            a = [V(V_MX[q]) for q in qs]
            d = V(V_T[qh])
            b = [V(V_T[2 + 2 * qh]), V(V_T[3 + 2 * qh])]
            # A row's 64 scores of a tile sit in four lanes (g = 0..3).  Whether the running maximum must move is decided on the
            # lanes' PARTIAL maxima: some lane exceeds the threshold exactly when the row's maximum does -- so the exchanges with
            # lanes ^ 16 and ^ 32 happen only where the complete maximum is used: in a job's first tile (it sets m) and at the
            # head of the rare firing path.
            def row_max(x, tmp):
                return [I("v_mov_b32", tmp, x), I("v_permlane16_swap_b32", x, tmp), I("v_max_f32", x, x, tmp),
                        I("v_mov_b32", tmp, x), I("v_permlane32_swap_b32", x, tmp), I("v_max_f32", x, x, tmp)]
            if part in (0, 1):
                return []
            if part == 2:
                if init:
                    return row_max(a[0], b[0]) + row_max(a[1], b[1]) + [I("v_mul_f32", V(V_MC[q]), S_C, a[k]) for k, q in enumerate(qs)]
                return [I("v_fma_f32", d, a[0], S_C, -V(V_MC[qs[0]])), I("v_fma_f32", b[0], a[1], S_C, -V(V_MC[qs[1]])),
                        I("v_max_f32", d, d, b[0])]
            if init:
                return []
            if part == 3:
                return [I("v_cmp_gt_f32", S_FIRE[qh], d, S_THR)]
            if "fire_nobranch" in self.abl:       # (timing-only: the compare without its branch)
                return []
            l_fire, l_back = self.lab("fire"), self.lab("fire_back")
            # rare: raise this query block's running maxima now (every s' = s * c - m of the PREVIOUS tile has been formed:
            # plan order), remember the factors; O and the row sums are scaled at the end of the coming phase B
            exact = self.fire_exact(Sb, qh, lazy) if lazy is not None else []
            blk = [label(l_fire)] + exact
            for k, q in enumerate(qs):
                t2, t3 = b[1], d
                blk += row_max(a[k], b[0]) + [I("v_mul_f32", t2, S_C, a[k]), I("v_max_f32", t2, t2, V(V_MC[q])),
                                              I("v_sub_f32", t3, V(V_MC[q]), t2), I("v_mov_b32", V(V_MC[q]), t2), I("v_exp_f32", V(V_CO[q]), t3)]
            self.ool.append(blk + [I("s_or_b32", S_FLAG, S_FLAG, 1 << qh), I("s_branch", Label(l_back))])
            return [I("s_cmp_lg_u64", S_FIRE[qh], 0), I("s_cbranch_scc1", Label(l_fire)), label(l_back)]
        if kind == "f":
            e = payload
            y = V(Sb + e)
            return [I("v_fma_f32", y, y, S_C, -V(V_MC[2 * (e >> 5) + ((e >> 3) & 1)]), tag=f"fma {e}")]
        if kind == "e":
            y = V(Sb + payload)
            return [I("v_exp_f32", y, y, tag=f"exp {payload}")]
        if kind == "cv":
            g, j = payload
            # (in place: the 8 scores of a block's key half -> its first four registers, the B operand P^T(qb16, kk))
            return [I(self.cvt, V(Sb + 16 * g + 8 * (j >> 2) + (j & 3)), V(Sb + 16 * g + 2 * j), V(Sb + 16 * g + 2 * j + 1), tag=f"cvt g{g} {j}")]
        raise KeyError(kind)

    @staticmethod
    def op_qb(kind, payload):
        """query block a placed softmax operation belongs to (None: not tied to one)"""
        if kind in ("mx", "cv"):
            return payload[0] >> 1
        if kind == "dec":
            return payload[0]
        if kind in ("f", "e"):
            return payload >> 5
        if kind == "ms":
            return payload
        return None

    def tile_fill(self, Sb, lo, hi, init, masks=None, abl=(), qbs=(0, 1)):
        """[(gap - lo, [insts], is_exp)] of the tile's operations with lo <= tau < hi.  masks: causal (jd, cond): the tile is
        diagonal tile jd of its job (when cond = (sgpr, value) holds, if given).  A job's first tile (init) gets its scores
        masked up front -- the tests of score group g go in front of its first row-maximum operation; every other diagonal
        tile is masked lazily (mask_lazy)"""
        out = []
        seen_mask = set()
        tail = masks is not None and masks[0] == "tail"     # non-causal ragged: ("tail", j[, cond]) -- seam tile j may hold keys >= N
        if tail:
            masks = masks[1:]
        jd, cond = (masks + (None,))[:2] if masks is not None else (None, None)
        lazy = (jd, cond) if masks is not None and not init else None
        plan = self.tile_plan(init)
        if lazy is not None:
            lz = self.lazy_tau()
            if self.split and not tail:
                # split row map: a hidden (tile, query block) is not computed at all -- no running-maximum swap ('ms' / 'mr');
                # the packed-P masking only where this body's waves sit on the tile's diagonal and the block is computed
                pm = self.cls is not None and (jd >> 1) in qbs and self.cls == ("low", "high")[jd & 1]
                plan = sorted(plan + ([(lz["pm"], "pm", None)] if pm else []), key=lambda x: x[0])
            else:
                plan = sorted(plan + [(lz["ms0"], "ms", 0), (lz["ms1"], "ms", 1), (lz["mr"], "mr", None)] +
                              ([] if tail else [(lz["pm"], "pm", None)]), key=lambda x: x[0])
        for t, kind, payload in plan:
            if not (lo <= t < hi):
                continue
            if self.op_qb(kind, payload) is not None and self.op_qb(kind, payload) not in qbs:
                continue
            if ("no_" + kind) in abl or (kind == "dec" and payload[1] >= 3 and "no_fire" in abl):
                continue   # timing-only ablations (diagnostic build)
            if tail:
                assert not init
                ins = self.mask_tail(Sb, kind, payload, jd, cond) if kind in ("ms", "mr") else self.tile_op(Sb, kind, payload, init, None)
            else:
                ins = self.tile_op(Sb, kind, payload, init, lazy)
            if not ins:
                continue
            if masks is not None and (init or tail) and kind == "mx" and payload[1] == 0 and payload[0] not in seen_mask:
                seen_mask.add(payload[0])
                ins = (self.mask_tail_tests(Sb, payload[0], jd, cond) if tail else self.mask_tests(Sb, payload[0], jd, cond)) + ins
            out.append((t - lo, ins, kind == "e"))
        return out

    # ------------------------------------------------------------------ causal masks
    def group_mask_ops(self, Y, g, what):
        """-inf into the scores of group g = 2 qh + kk (32 rows x 32 keys: register 8 q' + 4 k' + rr <-> row 16 q' + li, key
        16 k' + 4 g_l + rr): what = "all", or "tri" (the key half is the query block's own: keys behind the query) -- sub-block
        (q', k') = (0, 0) and (1, 1) get the 16 x 16 triangle (keep rr <= li - 4 g_l = V_IMH), (0, 1) goes entirely, (1, 0) stays"""
        if what == "all":
            return [I("v_mov_b32", V(Y + 16 * g + r), NINF) for r in range(16)]
        out = []
        for q1 in range(2):
            for rr in range(4):
                y = V(Y + 16 * g + 8 * q1 + 4 * q1 + rr)       # sub-block (q', k' = q')
                out += [I("v_cmp_ge_i32", VCC, V(V_IMH), rr), I("v_cndmask_b32", y, NINF, y, VCC)]
        out += [I("v_mov_b32", V(Y + 16 * g + 4 + rr), NINF) for rr in range(4)]   # (0, 1)
        return out

    def mask_lazy(self, Sb, kind, payload, lazy):
        """Diagonal tiles other than a job's first are not masked before the softmax: the row maxima are taken over all 64 keys
        (a masked key can only RAISE a maximum: harmless unless it fires the deferred-maximum rescale, and that path --
        fire_exact -- masks the scores exactly and takes the maxima again), and
          'pm'  the wave on the diagonal clears the masked entries of the PACKED P (AND masks from the set-up for the two packed
                registers of a 16 x 16 triangle, zero moves for what is hidden).
        Every wave of this body sits on the diagonal of tile jd (tile_fill adds 'pm' only there): the even wave has pattern D0 on
        query block qa = jd >> 1 (key half 0: triangle, key half 1: hidden), the odd one D1 (key half 0 visible, key half 1:
        triangle).  D0 in line, D1 out of line (a taken branch costs what eight VALU operations do).  Packed register 8 q' + jj of
        a group: rows 16 q' + li, keys 16 (jj >> 1) + 4 g_l + 2 (jj & 1), + 1"""
        jd, cond = lazy
        assert self.split and kind == "pm"
        qa = jd >> 1
        g0, g1 = 2 * qa, 2 * qa + 1

        def tri(g):
            p = lambda q1, jj: V(Sb + 16 * g + 8 * q1 + jj)
            return [I("v_and_b32", p(q1, 2 * q1 + jj), p(q1, 2 * q1 + jj), V(V_PM[jj])) for q1 in range(2) for jj in range(2)] + \
                [I("v_mov_b32", p(0, 2 + jj), 0) for jj in range(2)]
        l_d1, l_back, l_skip = self.lab("pmask_d1"), self.lab("pmask_back"), self.lab("pmask_skip")
        self.ool.append([label(l_d1)] + tri(g1) + [I("s_branch", Label(l_back))])
        head = [I("s_cmp_eq_u32", cond[0], cond[1]), I("s_cbranch_scc0", Label(l_skip))] if cond is not None else []
        return head + [I("s_bitcmp1_b32", S_WAVE, 0), I("s_cbranch_scc1", Label(l_d1))] + tri(g0) + \
            [I("v_mov_b32", V(Sb + 16 * g1 + 8 * q1 + jj), 0) for q1 in range(2) for jj in range(4)] + [label(l_back)] + \
            ([label(l_skip)] if cond is not None else [])

    def mask_tail(self, Sb, kind, payload, j, cond):
        """non-causal ragged, seam tile j (keys 64 j .. 64 j + 63 of the job's last 256; S_KT0 of them are real): a tile wholly
        behind N gets +inf for the running maximum like a tile below the causal diagonal (mask_lazy 'ms' / 'mr')"""
        if j == 0:
            return []          # at least one key of tile 0 is real
        assert cond is None
        sel = [I("s_cmp_gt_i32", S_KT0, 64 * j), I("s_cselect_b64", VCC, -1, 0)]     # VCC: the tile holds a real key
        if kind == "ms":       # (payload: the 32-row query block of the plan = the 16-row blocks 2 qh, 2 qh + 1, each with its own maximum)
            out = list(sel)
            for q in (2 * payload, 2 * payload + 1):
                out += [I("v_mov_b32", V(V_MSV[q]), V(V_MC[q])), I("v_cndmask_b32", V(V_MC[q]), -NINF, V(V_MC[q]), VCC)]
            return out
        return sel + [I("v_cndmask_b32", V(V_MC[q]), V(V_MSV[q]), V(V_MC[q]), VCC) for q in range(4)]

    def mask_tail_tests(self, Y, g, j, cond):
        """in front of score group g's first row-maximum operation: if some of its 32 keys lie at or behind N (and some key of
        the tile is real: else mask_tail deals with it), -inf into those scores, out of line.  Register r = 8 q' + 4 k' + rr of a group
        <-> key 16 k' + 4 g_l + rr of the group's 32 (g_l = lane >> 4)"""
        kb = g & 1
        l_m, l_back = self.lab("tail"), self.lab("tail_back")
        t = V(V_T[6])
        blk = [label(l_m)]
        if cond is not None:
            blk += [I("s_cmp_eq_u32", cond[0], cond[1]), I("s_cbranch_scc0", Label(l_back))]
        if j > 0:
            blk += [I("s_cmp_gt_i32", S_KT0, 64 * j), I("s_cbranch_scc0", Label(l_back))]
        # T = real keys of this group minus the lane group's offset 4 g_l: register r is kept iff T > (r & 3) + 16 ((r >> 2) & 1)
        # (S_X2, the job decode's scratch: the S_T temporaries may be in the middle of a descriptor computation spread over gaps)
        blk += [I("s_sub_i32", S_X2, S_KT0, 64 * j + 32 * kb), I("v_lshrrev_b32", t, 4, V(V_LANE)), I("v_lshlrev_b32", t, 2, t),
                I("v_sub_u32", t, S_X2, t)]
        for r in range(16):
            blk += [I("v_cmp_gt_i32", VCC, t, (r & 3) + 16 * ((r >> 2) & 1)), I("v_cndmask_b32", V(Y + 16 * g + r), NINF, V(Y + 16 * g + r), VCC)]
        self.ool.append(blk + [I("s_branch", Label(l_back))])
        # in line: one compare and an untaken branch while all 32 keys of the group are real
        return [I("s_cmp_lt_i32", S_KT0, 64 * j + 32 * kb + 32), I("s_cbranch_scc1", Label(l_m)), label(l_back)]

    def fire_exact(self, Sb, qb, lazy):
        """head of the rare rescale path of a lazily masked diagonal tile: on the waves that sit on the diagonal the row maxima
        were taken over masked keys too -- mask this query block's scores now and take the maxima of its two 16-row blocks again
        (then the plain path decides with the exact maxima; the packed-P masking later is a no-op on the -inf entries).
        Only query block jd >> 1 has waves on the diagonal of tile jd: waves 2 p (pattern D0) and 2 p + 1 (D1), p = jd & 1"""
        jd, cond = lazy
        assert self.split
        if qb != jd >> 1:
            return []
        l_plain = self.lab("fire_plain")
        blk = []
        p2 = 2 * (jd & 1)
        l_d1, l_max = self.lab("fire_d1"), self.lab("fire_max")
        if cond is not None:
            blk += [I("s_cmp_eq_u32", cond[0], cond[1]), I("s_cbranch_scc0", Label(l_plain))]
        blk += [I("s_cmp_eq_u32", S_WAVE, p2 + 1), I("s_cbranch_scc1", Label(l_d1)),
                I("s_cmp_eq_u32", S_WAVE, p2), I("s_cbranch_scc0", Label(l_plain))]
        blk += self.group_mask_ops(Sb, 2 * qb, "tri") + self.group_mask_ops(Sb, 2 * qb + 1, "all") + [I("s_branch", Label(l_max))]
        blk += [label(l_d1)] + self.group_mask_ops(Sb, 2 * qb + 1, "tri") + [label(l_max)]
        for q1 in range(2):
            mx = V(V_MX[2 * qb + q1])
            ys = [V(Sb + 16 * (2 * qb + kk) + 8 * q1 + r) for kk in range(2) for r in range(8)]
            blk += [I("v_max3_f32", mx, ys[0], ys[1], ys[2])] + [I("v_max3_f32", mx, mx, ys[2 * j + 1], ys[2 * j + 2]) for j in range(1, 7)] + \
                [I("v_max_f32", mx, mx, ys[15])]
        return blk + [label(l_plain)]

    def mask_tests(self, Y, g, jd, cond=None):
        """causal: the tile whose softmax starts is diagonal tile jd (0..3) of its job: keys 64 jd .. 64 jd + 63 of the 256-key
        diagonal span against this wave's rows 64 w .. 64 w + 63.  w > jd: nothing; w == jd: score groups (qb0, kb0) and
        (qb1, kb1) get the triangle (key > query -> -inf), (qb0, kb1) is all -inf; w < jd: all -inf.  Register r of a
        group <-> key (r & 3) + 8 (r >> 2) + 4 h, lane <-> query i.  cond: (sgpr, value): the tile is diagonal at all.
        Returns the in-line tests for score group g (in front of its first row-maximum operation); the masking itself runs
        out of line.  The branch sits where that row-maximum operation is legal, i.e. the MFMA -> VALU wait states have passed
        (check.check_branch_targets verifies it on the built program)."""
        if self.split:    # (always)
            # waves 2 p / 2 p + 1 (p = jd & 1) carry patterns D0 / D1 on query block jd >> 1 (see mask_lazy); a block wholly
            # hidden from a wave is not computed at all by the split bodies
            l_back = self.lab("mask_back")
            qb, kb = g >> 1, g & 1
            if qb != jd >> 1:
                return []
            p2 = 2 * (jd & 1)
            tests = []
            l_d0 = self.lab("mask_d0")
            self.ool.append([label(l_d0)] + self.group_mask_ops(Y, g, "tri" if kb == 0 else "all") + [I("s_branch", Label(l_back))])
            tests += [I("s_cmp_eq_u32", S_WAVE, p2), I("s_cbranch_scc1", Label(l_d0))]
            if kb == 1:
                l_d1 = self.lab("mask_d1")
                self.ool.append([label(l_d1)] + self.group_mask_ops(Y, g, "tri") + [I("s_branch", Label(l_back))])
                tests += [I("s_cmp_eq_u32", S_WAVE, p2 + 1), I("s_cbranch_scc1", Label(l_d1))]
            if cond is None:
                return tests + [label(l_back)]
            l_tests = self.lab("mask_tests")
            self.ool.append([label(l_tests)] + tests + [I("s_branch", Label(l_back))])
            return [I("s_cmp_eq_u32", cond[0], cond[1]), I("s_cbranch_scc1", Label(l_tests)), label(l_back)]
        raise AssertionError("the a16 kernels use the split row map only")

    # ------------------------------------------------------------------ the two phases
    def emit_slot(self, m, entries, keep_order=False):
        """one MFMA slot and the fillers behind it.  entries: [(order, [insts])] with order 0 = LDS / DMA loads, 1 = exp2, 2 = the
        rest, 3 = last.  A slot of two MFMAs splits its fillers between the two half gaps (each MFMA holds the vector issue port
        for 8 of its 16 cycles): in their order, cut where the issue costs of the halves balance"""
        from .isa import issue_cost
        cost = lambda ins: sum(issue_cost(x) for x in ins)
        if keep_order:
            # (fillers of several original gaps, merged: they may depend on each other -- the order stands, the split is by cost)
            if isinstance(m, Inst):
                return [m] + [x for _, ins in entries for x in ins]
            total, acc, out, second = sum(cost(i) for _, i in entries), 0, [m[0]], False
            for _, ins in entries:
                if not second and 2 * acc >= total:
                    out.append(m[1])
                    second = True
                out += ins
                acc += cost(ins)
            return out if second else out + [m[1]]
        entries = sorted(entries, key=lambda x: x[0])      # (stable: entries of equal order keep the order they were added in --
        # the scalar units of `early` depend on each other)
        if isinstance(m, Inst):
            return [m] + [x for _, ins in entries for x in ins]
        # the order stands; the second MFMA goes where the two half gaps balance best (an LDS read costs the port about half of
        # what a VALU instruction does; 'last' entries -- a DMA piece's scalar set-up -- stay behind the second MFMA)
        w = [cost(ins) // 2 if o == 0 else cost(ins) for o, ins in entries]
        n_mov = sum(1 for o, _ in entries if o != 3)
        best, cut = None, 0
        for k in range(n_mov + 1):
            imb = abs(sum(w[:k]) - sum(w[k:]))
            if best is None or imb < best:
                best, cut = imb, k
        out = [m[0]] + [x for _, ins in entries[:cut] for x in ins] + [m[1]] + [x for _, ins in entries[cut:] for x in ins]
        return out

    def emit_phase(self, mfmas, gaps):
        """gaps[k] = fillers behind slot k: (order, [insts]); a slot is one MFMA or a pair (emit_slot)"""
        out = []
        kept = [k for k, m in enumerate(mfmas) if m is not None]
        if len(kept) == len(mfmas):
            for k, m in enumerate(mfmas):
                out += self.emit_slot(m, gaps.get(k, []))
            return out
        # some slots are left out (a query block the tile is hidden from): the fillers of the original gaps, each gap's group
        # kept whole and in the original order, are spread over the remaining slots in proportion -- a filler never moves in
        # front of an MFMA it followed (check.fix pads what lands too close behind one)
        n, nk = len(mfmas), len(kept)
        assert nk > 0, "a phase without MFMAs is emitted by its caller"
        buckets = [[] for _ in range(nk)]
        for k in range(n):
            buckets[min(k * nk // n, nk - 1)] += sorted(gaps.get(k, []), key=lambda x: x[0])
        for idx, k in enumerate(kept):
            out += self.emit_slot(mfmas[k], buckets[idx], keep_order=True)
        return out

    def phase_a(self, t4, with_qk=True, cur=True, nxt=True, nxt_init=False, masks=None, steady=False, dma=(), cur_masks=None, extra=(),
                cur_qbs=(0, 1), nxt_qbs=(0, 1)):
        """A(t), t4 = t & 3: QK^T(t+1) -> S[1-p]  ||  V(t) reads from VB[t % R]  ||  the late softmax operations of tile t (on S[p])
        ||  the early ones of tile t+1 (on S[1-p])"""
        p = t4 & 1
        X, Y = SBUF[p], SBUF[1 - p]
        abl = self.abl if steady else set()
        mf = self.qk_mfmas(Y, qbs=nxt_qbs) if with_qk else []
        gaps = {}
        add = lambda k, order, ins: gaps.setdefault(min(max(int(k), 0), 31), []).append((order, ins))
        if cur:
            if "novread" not in abl:
                for k, ins in enumerate(self.v_reads(t4 % self.R)):
                    add(self.a_vread_gap(k), 0, [ins])
            if "nofinish" not in abl:
                for k, ins, is_exp in self.tile_fill(X, 32 + self.NB, self.T_END, False, cur_masks, abl=abl, qbs=cur_qbs):
                    add(k, 1 if is_exp else 2, ins)
        if nxt and "nostart" not in abl:
            for k, ins, is_exp in self.tile_fill(Y, 0, 32, nxt_init, masks, abl=abl, qbs=nxt_qbs):
                add(k, 1 if is_exp else 2, ins)
        for g, ins in extra:
            add(g, 2, ins)
        for g, setup, load in dma:   # LDS-DMA pieces riding in this phase (the seam's Q rows): set-up ends gap g - 1, load opens gap g
            add(g - 1, 3, setup)
            add(g, 0, [load])
        if not mf:
            return [x for k in sorted(gaps) for _, ins in sorted(gaps[k], key=lambda x: x[0]) for x in ins]
        return self.emit_phase(mf, gaps)

    def pv_mfmas(self, X, qbs=(0, 1)):
        """O^T(qb16, db16) += V^T(kk, db16) . P^T(qb16, kk) as 32 slots of two MFMAs (the blocks of a query block qh share the V^T
        fragment, and so do the two slots of a (kk, db16)); behind every eight slots the row sums of two P fragments on the matrix
        pipe: against the all-ones operand V_ONES every row of D is the 32-key sum of the lane's own query (V_LACC[qb16])"""
        out = []
        for kk in range(2):
            for half in range(2):
                for db16 in range(4 * half, 4 * half + 4):
                    for qh in range(2):
                        pair = [I(self.mfma, A_O(q, db16), V_F(kk, db16), P_OP(X, q, kk), A_O(q, db16), tag=f"pv kk{kk} db{db16} qb{q}")
                                for q in (2 * qh, 2 * qh + 1)]
                        out.append(pair if qh in qbs else None)
                for q in (2 * half, 2 * half + 1):
                    out.append(I(self.mfma, V(V_LACC[q], 4), V(V_ONES, 4), P_OP(X, q, kk), V(V_LACC[q], 4), tag=f"rowsum kk{kk} qb{q}")
                               if half in qbs and "no_rowsum" not in self.abl else None)   # (no_rowsum: timing-only, l stays 0)
        return out

    def phase_b(self, t4, with_pv=True, nxt=True, nxt_init=False, with_kread=True, with_dma=True, steady=False,
                pre=(), early=(), late=(), masks=None, own_gaps=None, post=(), cur_qbs=(0, 1), nxt_qbs=(0, 1)):
        """B(t), t4 = t & 3: P.V(t) and the row sums of P(t) from S[p]  ||  the middle softmax operations of tile t+1 (on S[1-p])
        ||  K(t+2) reads from KB[(t+2) % R]  ||  LDS-DMA V(t+dv) -> VB[(t+dv) % R], K(t+dk) -> KB[(t+dk) % R].
        pre: instructions ahead of the phase;  early: scalar work / register loads spread over the first gaps;
        late: further DMA pieces (the next job's Q rows) as (gap, set-up, load);  own_gaps: the gaps of this step's own pieces"""
        p = t4 & 1
        X, Y = SBUF[p], SBUF[1 - p]
        abl = self.abl if steady else set()
        mf = self.pv_mfmas(X, qbs=cur_qbs) if with_pv else []
        NB = self.NB
        gaps = {}
        add = lambda k, order, ins: gaps.setdefault(min(max(int(k), 0), NB - 1), []).append((order, ins))
        head = list(pre)
        post, post_arg = [], list(post)
        if with_kread and "nokread" not in abl:
            for k, ins in enumerate(self.k_reads((t4 + 2) % self.R)):
                add(self.b_kread_gap(k), 0, [ins])
        if with_dma:
            pieces = [self.dma_piece("v", j, (t4 + self.dv) % self.R) for j in range(4)] + \
                [self.dma_piece("k", j, (t4 + self.dk) % self.R) for j in range(4)]
            for k, pc in enumerate(pieces):
                if "nodma" in abl:
                    continue
                # (with further pieces behind them -- the seam's Q rows -- this step's own go first: the counted waits
                # assume all eight are older than the sixteen)
                g = own_gaps[k] if own_gaps else self.b_dma_gap(k)
                # scalar set-up (soffset, M0) at the end of the previous gap, the load first in its own: the MFMA between
                # them is the wait state the M0 write needs
                setup, load = [x for x in pc if not x.op.startswith("buffer_load") and x.op != "s_nop"], [x for x in pc if x.op.startswith("buffer_load")]
                add(g - 1, 3, setup)
                add(g, 0, load if mf else [I("s_nop", 0)] + load)
            post += [I("s_add_u32", S_VDMA, S_VDMA, S_V64), I("s_add_u32", S_KDMA, S_KDMA, S_K64)]
        # scalar work rides in the first gaps one UNIT at a time: an instruction that consumes SCC (the s_addc of a 64-bit
        # add, a select or branch on a compare) stays glued to the instructions since its producer -- other fillers write
        # SCC too (the DMA set-up's s_add_u32), and a descriptor base once lost its carry that way (check.py R9)
        units = []
        for ins in early:
            d_, u_ = ins.defs_uses()
            if ("scc", 0) in u_ and units:
                units[-1].append(ins)
            else:
                units.append([ins])
        ne = len(units)
        for k, unit in enumerate(units):
            add(1 + 12 * k // max(ne, 1), 2, unit)   # done before this phase's own DMA pieces (gap 15 on) and the late ones
        for g, setup, load in late:   # (gap pairs disjoint from the own pieces': both use M0 and the scratch offset register)
            add(g - 1, 3, setup)
            add(g, 0, [load])
        if nxt and "nostart" not in abl:
            for k, ins, is_exp in self.tile_fill(Y, 32, 32 + NB, nxt_init, masks, abl=abl, qbs=nxt_qbs):
                add(k, 1 if is_exp else 2, ins)
        if not mf:
            body = [x for k in sorted(gaps) for _, ins in sorted(gaps[k], key=lambda x: x[0]) for x in ins]
        else:
            body = self.emit_phase(mf, gaps)
        body = head + body + post + post_arg
        if nxt and not nxt_init or True:
            # deferred rescale of O and the row sums by the factors the decisions of this step left (rare)
            l_rs, l_back = self.lab("rescale"), self.lab("rescale_back")
            body += [I("s_cmp_lg_u32", S_FLAG, 0), I("s_cbranch_scc1", Label(l_rs)), label(l_back)]
            # (bit qb of S_FLAG: query block qb fired in this step -- only its accumulators are touched; packed multiplies: the
            # matrix pipe is idle here.  f16 inputs fire a few times per job: P must stay below 65 504)
            blk = [label(l_rs), I("s_nop", 15)]
            tmp = [V(V_T[k]) for k in range(8)]
            co = V(V_T[8], 2)      # (an even register: the factor is read as the low word of an aligned 64-bit operand)
            for qh in range(2):
                l_skip = self.lab("rescale_skip")
                blk += [I("s_bitcmp1_b32", S_FLAG, qh), I("s_cbranch_scc0", Label(l_skip))]
                for q in (2 * qh, 2 * qh + 1):
                    blk += [I("v_mov_b32", co.sub(0), V(V_CO[q])), I("s_nop", 0)]
                    for base in range(0, 32, 8):
                        regs = [A(q * 32 + base + k) for k in range(8)]
                        blk += [I("v_accvgpr_read_b32", tmp[k], regs[k]) for k in range(8)]
                        blk += [I("v_pk_mul_f32", V(tmp[k].idx, 2), V(tmp[k].idx, 2), co, op_sel_hi=(1, 0)) for k in range(0, 8, 2)]
                        blk += [I("v_accvgpr_write_b32", regs[k], tmp[k]) for k in range(8)]
                    blk += [I("v_pk_mul_f32", V(V_LACC[q] + k, 2), V(V_LACC[q] + k, 2), co, op_sel_hi=(1, 0)) for k in (0, 2)]
                    blk += [I("v_mov_b32", V(V_CO[q]), 1.0)]
                blk += [label(l_skip)]
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
        out = [waitcnt(vmcnt=self.vm, lgkmcnt=0, comment="the DMA pieces the next reads need have landed; V fragments in"), I("s_barrier")]
        if steady and "skew" in self.abl:   # experiment: wave w leaves the barrier 8 w cycles late
            l1, l2 = self.lab("skew1"), self.lab("skew2")
            out += [I("s_bitcmp1_b32", S_WAVE, 0), I("s_cbranch_scc0", Label(l1)), I("s_nop", 7), label(l1),
                    I("s_bitcmp1_b32", S_WAVE, 1), I("s_cbranch_scc0", Label(l2)), I("s_nop", 15), label(l2)]
        return out

    def step(self, t4, a_pre=(), **kw):
        """one tile step, t4 = t & 3"""
        ka = {k: v for k, v in kw.items() if k in ("with_qk", "cur", "nxt", "nxt_init", "steady", "dma", "cur_masks", "extra",
                                                   "cur_qbs", "nxt_qbs")}
        kb = {k: v for k, v in kw.items() if k in ("with_pv", "nxt", "nxt_init", "with_kread", "with_dma", "steady", "pre", "early", "late",
                                                   "own_gaps", "post", "cur_qbs", "nxt_qbs")}
        if kw.get("masks") is not None:   # the masking tests of a score group sit in front of its first row-maximum operation
            ka["masks"] = kw["masks"]
            kb["masks"] = kw["masks"]
        out = [comment(f"---- step {t4}: phase A")]
        out += self.stamp_acc(2)
        out += list(a_pre)
        out += [waitcnt(lgkmcnt=0, comment="K fragments in")]
        out += self.phase_a(t4, **ka)
        out += self.stamp_acc(0)
        out += self.sync_mid(kw.get("steady", False), kw.get("vm"))
        out += self.stamp_acc(1)
        if kw.get("mid_stamp") is not None:      # (diagnostic builds: the seam's steps 2 and 3 split at their barrier)
            out += self.stamp(kw["mid_stamp"])
        out += [comment(f"---- step {t4}: phase B")]
        out += self.phase_b(t4, **kb)
        return out

    # ------------------------------------------------------------------ epilogue of the current job
    def epilogue_descs(self):
        """descriptors of the current job's O rows (S_SQ: the next job's Q rows are through by then) and L (S_NVRS: the next job's V
        descriptor has moved to S_VRS): scalar work that rides in the gaps of the seam's last phase B instead of standing in
        front of the epilogue"""
        return self.make_desc(S_SQ, S_O, S_OSB, S_OSH, S_B, S_HH, S_OSN) + self.make_desc(S_NVRS, S_L, S_LSB, S_LSH, S_B, S_HH, 2)

    def k_epilogue(self):
        """1 / l (one Newton step), L = m + log2 l, O^T -> rows through the wave's LDS slice -> 16-byte row stores, O^T := 0.
        A workgroup's O tile is 64 KiB and the CU's vector-memory path takes 64 bytes per cycle: the row stores of the first
        query block are sprinkled through the arithmetic of the second (back to back they cost ~85 cycles each with all four
        waves storing at once), those of the second run under the MFMAs that zero O and the L arithmetic that is left."""
        e = self.e
        t = [V(x) for x in V_T]
        e(comment("epilogue: l, 1/l, L; O^T -> rows through the wave's LDS slice -> global; O^T := 0"))
        # (the O and L descriptors of this job were formed in the seam's last phase B: epilogue_descs)
        e(I("s_nop", 15))  # last P.V MFMAs -> accumulator reads
        inv = [t[0], t[2], t[4], t[6]]    # (even registers: they are read as the low word of an aligned 64-bit operand below)
        m2 = [t[1], t[3], t[5], t[7]]
        nt = [t[8], t[9]]
        lsum = [V(V_LACC[q]) for q in range(4)]   # every register of a row-sum accumulator is the lane's own complete row sum
        for q in range(4):
            e(I("v_rcp_f32", inv[q], lsum[q]), I("v_log_f32", m2[q], lsum[q]))
        e(I("s_nop", 0))
        for q in range(4):
            # one Newton step: inv += inv * (1 - l * inv)
            e(I("v_fma_f32", nt[q & 1], -lsum[q], inv[q], 1.0), I("v_add_f32", m2[q], m2[q], V(V_MSV[q])))
            e(I("v_fma_f32", inv[q], nt[q & 1], inv[q], inv[q]), I(self.cvt, m2[q], m2[q], m2[q]))
        e(self.k_epi_consts())      # (the row sums are consumed: their registers hold the epilogue's lane constants from here on)
        e(self.stamp_async(0))
        # O: 4 accumulators (four consecutive columns of one query) -> 2 packed registers -> ds_write_b64 at (row li + 16 q', byte
        # 32 db16 + 8 g); one query block (32 rows) at a time.  (S[0] already holds the next job's first scores and v[128:191]
        # its K(1): rows and temporaries are score buffer 1, whose P was consumed by the job's last P.V.)  A batch = four 16-column
        # blocks of one 16-row block; the stages of consecutive batches (accumulator reads | scale | pack | LDS write) are woven so
        # that no instruction waits on its predecessor.
        rows = [V(SBUF[1] + 4 * k, 4) for k in range(8)]
        tset = [[V(SBUF[1] + 32 + 16 * sidx + k) for k in range(16)] for sidx in range(2)]

        def weave(*lists):
            out, idx = [], [0] * len(lists)
            while any(idx[k] < len(lists[k]) for k in range(len(lists))):
                for k in range(len(lists)):
                    if idx[k] < len(lists[k]):
                        out.append(lists[k][idx[k]])
                        idx[k] += 1
            return out

        def sprinkle(body, extras):
            """`extras` (instruction groups) spread evenly through `body`"""
            out, n = [], len(extras)
            for k, ins in enumerate(body):
                out.append(ins)
                j0, j1 = (k * n) // len(body), ((k + 1) * n) // len(body)
                for j in range(j0, j1):
                    out += extras[j]
            return out

        def row_stores(qh, off=S_T[0], stride=S_T[1], restride=False):
            """[[instructions of one row store]] of query block qh (its rows are back in `rows`).  restride: the 4-row stride is
            formed again in front of every store (scalar code that uses `stride` runs between the stores)"""
            pre, st = self.buf_op("buffer_store_dwordx4", rows[0], V(V_EO), S_SQ, off)
            out = [[I("s_mul_i32", off, S_QROW[qh], S_OSN), I("s_lshl_b32", stride, S_OSN, 2)] + pre + [st]]
            for k in range(1, 8):
                pre, st = self.buf_op("buffer_store_dwordx4", rows[k], V(V_EO), S_SQ, off)
                out.append(([I("s_lshl_b32", stride, S_OSN, 2)] if restride else []) + [I("s_add_u32", off, off, stride)] + pre + [st])
            return out

        def read_back():
            out = []
            for k in range(8):   # whole rows: row 4 k + a
                out.append(I("ds_read_b128", rows[k], V(V_ER), offset=4 * EPI_ROW * k))
            return out

        for qh in range(2):
            stages = []  # per batch: [reads, muls, packs, writes]
            for bi in range(4):
                q1, c = bi >> 1, bi & 1
                q = 2 * qh + q1
                tm = tset[bi & 1]
                src = A(32 * q + 16 * c, 16)       # O^T(q, db16 = 4 c .. 4 c + 3)
                rd = [I("v_accvgpr_read_b32", tm[k], src.sub(k)) for k in range(16)]
                # (packed fp32 multiplies: half the instructions; beside an MFMA they would cost ~50 cycles each, here the
                # matrix pipe is idle and every VALU instruction takes the same ~5.8 cycles at one wave per SIMD)
                mu = [I("v_pk_mul_f32", V(tm[k].idx, 2), V(tm[k].idx, 2), V(inv[q].idx, 2), op_sel_hi=(1, 0)) for k in range(0, 16, 2)]
                cv = []
                for k4 in range(4):
                    cv += [I(self.cvt, tm[4 * k4], tm[4 * k4], tm[4 * k4 + 1]), I(self.cvt, tm[4 * k4 + 1], tm[4 * k4 + 2], tm[4 * k4 + 3])]
                wr = [I("ds_write_b64", V(V_EW), V(tm[4 * k4].idx, 2), offset=16 * EPI_ROW * q1 + 32 * (4 * c + k4)) for k4 in range(4)]
                stages.append((rd, mu, cv, wr))
            # software pipeline over the four batches (two register sets): batch b + 1 is read while batch b is scaled, ...
            head = stages[0][0] + weave(stages[0][1], stages[1][0]) + stages[0][2] + stages[0][3]
            tail = weave(stages[1][1], stages[2][0]) + stages[1][2] + stages[1][3] + weave(stages[2][1], stages[3][0]) + \
                stages[2][2] + stages[2][3] + stages[3][1] + stages[3][2] + stages[3][3]
            e(head)
            if qh == 1:
                e(self.stamp_async(2))
                # the first block's rows have been on their way back from LDS since before this block started (four of this
                # block's writes are younger): wait for them once, then one row store every ~25 instructions
                e(waitcnt(lgkmcnt=4))
                e(sprinkle(tail, row_stores(0)))
            else:
                e(tail)
            e(read_back())
            e(self.stamp_async(1 if qh == 0 else 3))
        # O^T := 0 for the next job on the matrix pipe: eight 32x32x16 MFMAs on zero operands clear 16 accumulators each (instead
        # of 128 v_accvgpr_write)
        z = V(SBUF[1] + 32, 4)    # (the second block's temporaries: free again)
        e([I("v_mov_b32", z.sub(k), 0) for k in range(4)], I("s_nop", 1))
        # (the wave can issue one of these every 32 cycles and one row store every ~75 with all four waves storing: interleaved,
        # the stores hide the MFMAs; two MFMAs go first, under the read-back's LDS round trip)
        zero = [I("v_mfma_f32_32x32x16_" + self.dtype, A(16 * k, 16), z, z, 0) for k in range(8)]
        # (offset / stride registers the job bookkeeping below leaves alone: k_promote and k_advance use S_T[0..4], [6], [7])
        st1 = row_stores(1, off=S_T[5], stride=S_T[6], restride=True)
        e(zero[0], zero[1])
        e(self.stamp_async(4))
        e(waitcnt(lgkmcnt=0))
        e(self.stamp_async_flush((13, 14, 15, 20, 21)))   # (before the scalar code below: it uses the stamp registers)
        # L store (lanes 0..15: the lanes of group g = 0 hold the sixteen rows of a block), in the I/O dtype: in front of the job
        # bookkeeping (it needs this job's row numbers and S_T)
        e(I("s_mov_b64", EXEC, 0xFFFF))
        for q in range(4):
            pre, st = self.buf_op("buffer_store_short", m2[q], V(V_L2), S_NVRS, S_T[0])
            e(I("s_add_u32", S_T[0], S_QROW[q >> 1], 16 * (q & 1)), I("s_lshl_b32", S_T[0], S_T[0], 1), pre, st)
        e(I("s_mov_b64", EXEC, -1))
        for k in range(8):
            e(st1[k])
            if k + 2 < 8:
                e(zero[k + 2])
            if k == 1:
                # Job bookkeeping in the shadow of the row stores (the vector-memory path takes ~75 cycles per store with four
                # waves storing: the scalar code runs while the first two drain): the next job becomes the current one and the
                # job after it is decoded -- ~200 cycles that used to stand in front of the seam.  S_FINAL still says "this
                # job is the workgroup's last" for the branch behind the epilogue: it is put aside in S_FLAG (idle between steps)
                l_last = self.lab("epi_last")
                e(I("s_mov_b32", S_FLAG, S_FINAL), I("s_cmp_lg_u32", S_FINAL, 0), I("s_cbranch_scc1", Label(l_last)))
                self.k_promote()
                self.k_advance(vt=(tset[0][4], tset[0][5]))     # (not z: SBUF[1] + 32..35 is the zero MFMAs' operand)
                e(label(l_last))
        # the row sums of the next job start from zero (their registers held the epilogue's lane constants until the last store)
        e([I("v_mov_b32", V(V_LACC[q] + k), 0) for q in (1, 2, 3, 0) for k in range(4)])

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
        self.k_advance()      # (every later job is decoded in its predecessor's epilogue, under the row stores)
        e(comment("first job: K / V descriptors, K(0..2), V(0..1) by LDS-DMA, Q rows"))
        e(self.make_desc(S_KRS, S_K, S_KSB, S_KSH, S_B, S_HH, S_KSN), self.make_desc(S_VRS, S_V, S_VSB, S_VSH, S_B, S_HH, S_VSN))
        e(I("s_mov_b32", S_KDMA, S_KW), I("s_mov_b32", S_VDMA, S_VW))
        e(self.stamp(0))
        # (the first QK^T needs the Q rows and K(0) only: they go first, and the wait in front of the first barrier leaves the other
        # tiles in flight -- all 256 workgroups start at once and the burst is bandwidth-bound, ~7 us for everything)
        qs_setup, qs_pieces = self.q_stage(S_B, S_HH, S_QI)
        e(qs_setup, [pc + [I("s_nop", 0), ld] for pc, ld in qs_pieces])
        late = []
        for j in range(self.dk - 1):
            (e if j == 0 else late.append)(self.dma_tile("k", j % self.R))
            if j < self.dv - 1:
                late.append(self.dma_tile("v", j % self.R))
        e(late)
        n_late = sum(1 for t in late for x in t if x.op.startswith("buffer_load"))
        e([I("v_accvgpr_write_b32", A(k), 0) for k in range(128)])   # O^T := 0
        e(waitcnt(vmcnt=n_late), I("s_barrier"))
        e(self.stamp(1))
        e(self.q_reads(), self.k_reads(0))
        # step -1 (buffers as t4 = 3): A = QK^T(0) only; B = start(0) as init, K(1) reads, DMA V(2), K(3)
        e(self.step(3, with_qk=True, cur=False, with_pv=False, nxt_init=True, masks=(0, (S_NT, 4)) if self.causal else None))
        e(self.stamp(2), self.stamp_flush(), self.stamp_acc(3), self.stamp_job(3))
        # ---- job loop
        e(label(l_job))
        e(I("s_lshr_b32", S_LOOP, S_NT, 2), I("s_sub_u32", S_LOOP, S_LOOP, 1),
          I("s_cmp_eq_u32", S_LOOP, 0), I("s_cbranch_scc1", Label(l_seam)))
        e(label(l_loop))
        for t4 in range(4):
            # causal: the last steady body starts the job's first diagonal tile in its last phase B
            tailm = ("tail", 0, (S_LOOP, 1)) if self.ragged and not self.causal else None
            e(self.step(t4, steady=True, masks=((0, (S_LOOP, 1)) if self.causal else tailm) if t4 == 3 else None))
        e(I("s_sub_u32", S_LOOP, S_LOOP, 1), I("s_cmp_lg_u32", S_LOOP, 0), I("s_cbranch_scc1", Label(l_loop)))
        e(label(l_seam))
        e(self.stamp(3), self.stamp_acc(2), self.stamp_flush(), self.stamp_job(0))
        # ---- the job's last four tiles: the next job's K / V / Q stream in, its first QK^T and softmax start run here
        cm = self.causal
        if True:
            kpre = self.make_desc(S_KRS, S_K, S_KSB, S_KSH, S_NB, S_NHH, S_KSN) + [I("s_mov_b32", S_KDMA, S_KW)] + \
                self.make_desc(S_NVRS, S_V, S_VSB, S_VSH, S_NB, S_NHH, S_VSN)
            vpre = [I("s_mov_b32", S_VRS.sub(k), S_NVRS.sub(k)) for k in range(4)] + [I("s_mov_b32", S_VDMA, S_VW)]
            sk, sv = 4 - self.dk, 4 - self.dv      # seam step whose phase B streams the next job's first K / V tile
            assert sk <= 2 and sv <= 2             # (step 3 re-uses S_SQ and S_NVRS for the epilogue's descriptors)
            qs_setup, qs_pieces = self.q_stage(S_NB, S_NHH, S_NQI)
            for st in range(4):
                kw = dict(masks=((st + 1,) if st < 3 else (0, (S_NNT, 4))) if cm else None, cur_masks=(st,) if cm else None)
                if self.ragged and not cm:   # keys at or behind N in the job's last four tiles (a ragged N has at least eight)
                    kw = dict(masks=("tail", st + 1) if st < 3 else None, cur_masks=("tail", st))
                early, pre = [], []
                if st == 0:
                    # the next job's Q rows start their way into the wave's LDS slice: sixteen pieces, never more than one
                    # DMA piece per two gaps (that rate is free beside the MFMAs): eight behind this step's own K / V pieces,
                    # four in the quiet end of the next phase A, four in front of the next step's own.  The barrier waits
                    # in between leave them in flight (vmcnt(8 / 12 + ...)); the one of step 2 retires them
                    early += qs_setup
                    kw.update(own_gaps=(1, 3, 5, 7, 11, 13, 15, 17),
                              late=[(g, *qs_pieces[k]) for k, g in enumerate((21, 23, 25, 27, 31, 33, 35, 37))])
                if st == 1:
                    kw.update(vm=self.vm + 12,
                              dma=[(g, *qs_pieces[8 + k]) for k, g in enumerate((24, 26, 28, 30))],
                              late=[(g, *qs_pieces[12 + k]) for k, g in enumerate((1, 3, 5, 7))])
                if st == sk:
                    early += kpre
                if st == sv:
                    pre += vpre
                if st == 2 and "noqreads" not in self.abl:   # (timing-only ablation: what the AGPR-destination reads cost)
                    early += self.q_reads()      # slice -> a[128:191] (Q was last read by this step's phase A)
                if st == 3:
                    early += self.epilogue_descs()
                    # the job's last tile: its running maxima are put aside for the epilogue before the next job's first
                    # tile re-initialises them (its row sums stay in V_LACC until the epilogue has read them)
                    save = [I("v_mov_b32", V(V_MSV[q]), V(V_MC[q])) for q in range(4)]
                    if (cm and not self.split) or (self.ragged and not cm):   # (behind the tile's 'mr': until then V_MSV holds what 'ms' put aside, mask_lazy)
                        kw.update(nxt_init=True, extra=list(kw.get("extra", ())) + [(self.LAZY_TAU["mr"] - self.PERIOD + 1, save)])
                    else:
                        kw.update(nxt_init=True, a_pre=save)
                e(self.stamp(16 + st))
                if self.split:
                    # split row map (wave w: 32-row blocks w and w + 4).  Diagonal tile j -- key blocks 2 j, 2 j + 1 -- against
                    # query block 0 (row block w): hidden for w < 2 j, on the diagonal for w = 2 j (D0) / 2 j + 1 (D1), visible
                    # above; against query block 1 (row block w + 4): the same with w + 4.  So tile 0: everything runs (waves
                    # 0 / 1 mask block 0), tile 1: block 0 only on waves 2, 3 (which mask it), tile 2: block 1 only (waves 0 / 1
                    # mask), tile 3: block 1 on waves 2, 3 only (which mask).  Step st finishes tile st and starts tile st + 1:
                    # two bodies per step, waves 0-1 ("low") out of line, waves 2-3 ("high") in line, each with only the MFMAs
                    # and softmax operations of the blocks it needs -- 216 MFMA slots on the critical path instead of 288.
                    cur_q = (((0, 1), (0, 1)), ((1,), (0, 1)), ((1,), (1,)), ((), (1,)))[st]
                    nxt_q = (((1,), (0, 1)), ((1,), (1,)), ((), (1,)), ((0, 1), (0, 1)))[st]
                    l_low, l_join = self.lab("low"), self.lab("low_join")
                    e(I("s_cmp_lt_u32", S_WAVE, 2), I("s_cbranch_scc1", Label(l_low)))
                    for ci, cls in ((1, "high"), (0, "low")):
                        ckw = dict(kw)
                        cq, nq_ = cur_q[ci], nxt_q[ci]
                        ckw.update(cur_qbs=cq or (0, 1), nxt_qbs=nq_ or (0, 1))
                        if not cq:
                            ckw.update(cur=False, with_pv=False)
                        if not nq_:
                            ckw.update(with_qk=False, nxt=False)
                        self.cls = cls
                        if cls == "high":
                            e(self.step(st, early=early, pre=pre, mid_stamp=20 + st if st >= 2 else None, **ckw), label(l_join))
                        else:
                            body, self.prog = self.prog, []
                            e(label(l_low), self.step(st, early=early, pre=pre, **ckw), I("s_branch", Label(l_join)))
                            self.ool.append(self.prog)
                            self.prog = body
                        self.cls = None
                    continue
                lean = cm and st >= 1 and "nolean" not in self.abl
                half = cm and st < 3 and "nolean" not in self.abl
                if lean:
                    # waves below this step's diagonal tile (w < st): the tile whose softmax finishes and whose P.V runs here is
                    # hidden from them, and so is the one that starts (steps 1, 2; step 3 starts the next job's first tile).
                    # They take a body with the same loads, DMA pieces, waits and barriers but without those MFMAs and softmax
                    # operations, and idle at the barriers: at the package power limit what one wave does not execute, the
                    # others run faster (+1.2 % on c3 causal, A/B in one process).  Such a wave's running maximum was swapped for
                    # +inf when the hidden tile started ('ms'): the lean body puts it back.
                    l_lean, l_join = self.lab("lean"), self.lab("lean_join")
                    e(I("s_cmp_lt_u32", S_WAVE, st), I("s_cbranch_scc1", Label(l_lean)))
                if half:
                    # the wave ON this step's diagonal (w == st): the tile that starts here is hidden from it -- no QK^T, no start
                    # of its softmax (+0.2 % at c3, +1 % at N = 2048 on top of the lean bodies)
                    l_half = self.lab("half")
                    l_join2 = l_join if lean else self.lab("half_join")
                    e(I("s_cmp_eq_u32", S_WAVE, st), I("s_cbranch_scc1", Label(l_half)))
                e(self.step(st, early=early, pre=pre, **kw))
                if lean or half:
                    e(label(l_join if lean else l_join2))
                    body, self.prog = self.prog, []
                    if lean:
                        lkw = dict(kw)
                        lkw["a_pre"] = [I("v_mov_b32", V(V_MC[qb]), V(V_MSV[qb])) for qb in range(2)] + list(kw.get("a_pre", ()))
                        lkw.update(dict(with_qk=False, cur=False, nxt=False, with_pv=False) if st < 3 else dict(cur=False, with_pv=False))
                        e(label(l_lean), self.step(st, early=early, pre=pre, **lkw), I("s_branch", Label(l_join)))
                    if half:
                        hkw = dict(kw)
                        # (no 'ms' runs for the hidden tile: the lean body of the next step restores from V_MSV all the same)
                        hkw["a_pre"] = list(kw.get("a_pre", ())) + [I("v_mov_b32", V(V_MSV[qb]), V(V_MC[qb])) for qb in range(2)]
                        hkw.update(with_qk=False, nxt=False)
                        e(label(l_half), self.step(st, early=early, pre=pre, **hkw), I("s_branch", Label(l_join2)))
                    self.ool.append(self.prog)
                    self.prog = body
        e(self.stamp(4), self.stamp_job(1))
        self.k_epilogue()
        e(self.stamp(5))
        # (S_FLAG: S_FINAL as it stood in front of the epilogue's job bookkeeping; back to 0 for the next step's rescale flag)
        e(I("s_cmp_lg_u32", S_FLAG, 0), I("s_mov_b32", S_FLAG, 0), I("s_cbranch_scc1", Label(l_end)))
        e(self.stamp(0), self.stamp_acc(3), self.stamp_job(2))
        e(I("s_branch", Label(l_job)))
        e(label(l_end), self.stamp_job(2), self.stamp_job_flush(), waitcnt(vmcnt=0), self.stamp(7, real=True), self.stamp(9), I("s_endpgm"))
        for blk in self.ool:
            e(blk)
        from .check import check_branch_targets, fix
        self.prog, self.pads = fix(self.prog)
        bad = check_branch_targets(self.prog)
        assert not bad, ("a branch enters a block that touches fresh MFMA results", bad[:4])
        return self.prog

    # ------------------------------------------------------------------ text
    def lds_total(self):
        return LDS_TOTAL

    def text(self):
        lines = [f".protected {self.name}", f".globl {self.name}", ".p2align 8", f".type {self.name},@function", f"{self.name}:"]
        lines += [x.text() for x in self.prog]
        lines += [f".L{self.name}_fend:", f".size {self.name}, .L{self.name}_fend-{self.name}", "",
                  '.section .rodata,"a",@progbits', ".p2align 6, 0x0", f".amdhsa_kernel {self.name}",
                  f"  .amdhsa_group_segment_fixed_size {self.lds_total()}", "  .amdhsa_private_segment_fixed_size 0",
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
            f"    .group_segment_fixed_size: {self.lds_total()}", f"    .kernarg_segment_align: 8", f"    .kernarg_segment_size: {KARG_SIZE}",
            f"    .max_flat_workgroup_size: 256", f"    .name: {self.name}", f"    .private_segment_fixed_size: 0",
            f"    .sgpr_count: 108", f"    .symbol: {self.name}.kd", f"    .vgpr_count: 512", f"    .agpr_count: 256",
            f"    .wavefront_size: 64"])


def product_gens():
    """the kernels of this generator that ship in libfa2_hip.so's code object (built by fa2_a64_gen.main)"""
    out = []
    for dtype in ("bf16", "f16"):
        for causal, ragged in ((False, False), (True, False), (False, True), (True, True)):
            g = Gen(dtype, causal, ragged=ragged)
            g.build()
            out.append(g)
    return out
