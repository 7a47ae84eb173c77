#!/usr/bin/env python3
"""Micro-benchmarks of instruction ISSUE cost beside v_mfma_f32_32x32x16_bf16 at one wave per SIMD (four waves per CU), the
regime of the a64 kernel: each kernel runs  64 x [ 8 x ( MFMA ; n copies of instruction X ) ]  and writes the s_memtime
cycles of the loop to the output buffer.  cycles per MFMA gap = total / 512; the slope over n is X's cost, the knee where
it leaves 32 the number of free slots.   (diagnostic: scripts/probes/mb_run.hip loads the code object and prints the table)
"""
from __future__ import annotations

import sys

from .isa import A, I, Label, M0, S, V, VCC, label, waitcnt
from .fa2_a64_gen import module_text


class MB:
    def __init__(self, name, fillers, agpr_c=True, chain=False, mfma=True, f8=False):
        self.name = name
        self.prog = []
        e = self.prog.append
        # s[0:1] kernarg: {out ptr, src ptr}; s2 = workgroup id
        e(I("s_load_dwordx4", S(4, 4), S(0, 2), 0))
        e(I("v_and_b32", V(200), 63, V(0)))
        e(I("v_lshlrev_b32", V(201), 4, V(200)))            # lane * 16: an LDS / buffer offset
        # store patterns: V(206) = (lane >> 4) * 256 + (lane & 15) * 16  (4 whole rows);  V(207) = (lane & 31) * 256 + (lane >> 5) * 8
        e(I("v_lshrrev_b32", V(206), 4, V(200))), e(I("v_lshlrev_b32", V(206), 8, V(206))), e(I("v_and_b32", V(207), 15, V(200)))
        e(I("v_lshl_add_u32", V(206), V(207), 4, V(206)))
        e(I("v_and_b32", V(207), 31, V(200))), e(I("v_lshlrev_b32", V(207), 8, V(207))), e(I("v_lshrrev_b32", V(205), 5, V(200)))
        e(I("v_lshl_add_u32", V(207), V(205), 3, V(207)))
        e(I("v_lshrrev_b32", V(202), 6, V(0)))
        e(I("s_nop", 1))
        e(I("v_readfirstlane_b32", S(20), V(202)))           # wave
        e(I("s_nop", 4))
        for k in range(0, 192):
            e(I("v_mov_b32", V(k), 0.5 if k % 3 else 1.0))
        for k in range(0, 256, 1):
            e(I("v_accvgpr_write_b32", A(k), 0))
        e(I("v_mov_b32", V(203), 0.25))
        for k in (208, 209, 210):
            e(I("v_mov_b32", V(k), 0x7F7F7F7F))      # scale bytes 127 = 2^0
        e(waitcnt(lgkmcnt=0))
        # buffer descriptor of the source buffer
        e(I("s_mov_b32", S(8), S(6)))
        e(I("s_and_b32", S(9), S(7), 0xFFFF))
        e(I("s_mov_b32", S(10), 0x7FFFFFF0))
        e(I("s_mov_b32", S(11), 0x00020000))
        e(I("s_mov_b32", S(21), 1.4426950408889634))
        e(I("s_lshl_b32", S(22), S(20), 12))                 # wave * 4096: LDS-DMA destination
        e(I("s_mov_b32", S(23), 0))
        e(I("s_mov_b32", S(24), 60.0))
        e(I("s_mov_b64", S(26, 2), 0)), e(I("s_mov_b64", S(30, 2), 0))
        e(I("s_barrier"))
        e(I("s_memtime", S(12, 2)))
        e(waitcnt(lgkmcnt=0))
        e(I("s_mov_b32", S(16), 64))
        lp = f".Lmb_{name}_loop"
        e(label(lp))
        for u in range(8):
            if chain:
                acc = A(0, 16) if agpr_c else V(0, 16)
            else:
                acc = A(16 * u, 16) if agpr_c else V(16 * (u % 4), 16)
            if mfma and f8 == "scale":     # ... its block-scaled form (two more VGPR operands: the lanes' E8M0 scale bytes)
                e(I("v_mfma_scale_f32_32x32x64_f8f6f4", acc, V(128 + 8 * (u % 4), 8), V(160 + 8 * (u % 4), 8), acc, V(208), V(209 + (u & 1)),
                    op_sel_hi=(0, 0, 0)))
            elif mfma and f8:     # the 64-cycle fp8 form of the a8 kernel (operands of 8 registers)
                e(I("v_mfma_f32_32x32x64_f8f6f4", acc, V(128 + 8 * (u % 4), 8), V(160 + 8 * (u % 4), 8), acc))
            elif mfma:
                e(I("v_mfma_f32_32x32x16_bf16", acc, V(128 + 8 * (u % 4), 4), V(160 + 4 * (u % 4), 4), acc))
            for f in fillers(u):
                if f.op.startswith("s_cbranch"):      # (the never-taken branch of the fire-test cases: a label of this kernel)
                    f = I(f.op, Label(f".Lmb_never_{name}"))
                e(f)
        e(I("s_sub_u32", S(16), S(16), 1))
        e(I("s_cmp_lg_u32", S(16), 0))
        e(I("s_cbranch_scc1", Label(lp)))
        e(waitcnt(vmcnt=0, lgkmcnt=0))
        e(I("s_memtime", S(14, 2)))
        e(waitcnt(lgkmcnt=0))
        e(I("s_sub_u32", S(14), S(14), S(12)))
        # out[(wg * 4 + wave)] = cycles
        e(I("s_lshl_b32", S(17), S(2), 2))
        e(I("s_add_u32", S(17), S(17), S(20)))
        e(I("s_lshl_b32", S(17), S(17), 2))
        e(I("v_mov_b32", V(204), S(17)))
        e(I("v_mov_b32", V(205), S(14)))
        e(I("global_store_dword", V(204), V(205), S(4, 2)))
        e(waitcnt(vmcnt=0))
        e(I("s_endpgm"))
        e(label(f".Lmb_never_{name}"))
        e(I("s_endpgm"))

    # reuse the kernel text / metadata emitters of the a64 generator (same descriptor: 512 registers, static LDS)
    def text(self):
        from .fa2_a64_gen import Gen
        g = Gen.__new__(Gen)
        g.name, g.prog = self.name, self.prog
        return Gen.text(g)

    def metadata(self):
        from .fa2_a64_gen import Gen
        g = Gen.__new__(Gen)
        g.name = self.name
        return Gen.metadata(g)


def cases():
    out = []
    r = lambda u, k: 64 + 8 * u + k      # arch VGPR 64..127: never an MFMA operand here
    EXP = lambda u, k: I("v_exp_f32", V(r(u, k)), V(r(u, k)))
    ADD = lambda u, k: I("v_add_f32", V(120 + (k & 1)), V(120 + (k & 1)), V(r(u, k)))
    FMA = lambda u, k: I("v_fma_f32", V(r(u, k)), V(r(u, k)), S(21), -V(203))
    MX3 = lambda u, k: I("v_max3_f32", V(124 + (k & 3)), V(124 + (k & 3)), V(r(u, k)), V(r(u, (k + 1) % 8)))
    CVT = lambda u, k: I("v_cvt_pk_bf16_f32", V(r(u, k)), V(r(u, k)), V(r(u, (k + 1) % 8)))
    DSR = lambda u, k: I("ds_read_b128", V(64 + 4 * ((8 * u + k) % 8), 4), V(201), offset=1024 * ((8 * u + k) % 16))
    TRR = lambda u, k: I("ds_read_b64_tr_b16", V(64 + 2 * ((8 * u + k) % 16), 2), V(201), offset=512 * ((8 * u + k) % 32))
    SAL = lambda u, k: I("s_add_u32", S(23), S(23), 1)
    PKA = lambda u, k: I("v_pk_add_f32", V(120, 2), V(120, 2), V(64 + 8 * u + 2 * (k % 4), 2))
    def DMA3(u, k):
        return I("buffer_load_dwordx4", V(201), S(8, 4), 0, offen=1, lds=1)
    M0S = lambda u, k: I("s_add_u32", M0, S(22), 1024 * (u % 4))
    # the fire test of the softmax plan: a compare into a scalar mask and a branch on it that is never taken (operands are 0.5 / 1.0
    # against a threshold of 60)
    THR = lambda: I("s_mov_b32", S(24), 60.0)
    CMPS = lambda u, k: I("v_cmp_gt_f32", S(26, 2), V(r(u, k)), S(24))
    CMPV = lambda u, k: I("v_cmp_gt_f32", VCC, V(r(u, k)), S(24))
    SCMP = lambda u, k: I("s_cmp_lg_u64", S(26, 2), 0)
    BRS = lambda u, k: I("s_cbranch_scc1", Label(".Lmb_never"))
    BRV = lambda u, k: I("s_cbranch_vccnz", Label(".Lmb_never"))
    SOR = lambda u, k: I("s_or_b64", S(28, 2), S(26, 2), S(30, 2))
    pats = {
        "none": [],
        "aa": [ADD, ADD], "cmps_a": [CMPS, ADD], "cmpv_a": [CMPV, ADD],
        "cmps_a_scmp_brs": [CMPS, ADD, SCMP, BRS], "cmpv_a_brv": [CMPV, ADD, BRV], "a_scmp_brs": [ADD, SCMP, BRS],
        "cmps_aaa_scmp_brs": [CMPS, ADD, ADD, ADD, SCMP, BRS], "cmpv_aaa_brv": [CMPV, ADD, ADD, ADD, BRV],
        "t_eafc": [TRR, EXP, ADD, FMA, CVT], "e_t_afc": [EXP, TRR, ADD, FMA, CVT], "eafc_t": [EXP, ADD, FMA, CVT, TRR],
        "t_e": [TRR, EXP], "e_t": [EXP, TRR], "tt_e": [TRR, TRR, EXP], "t_ea": [TRR, EXP, ADD], "t_eaa": [TRR, EXP, ADD, ADD],
        "tt_afc": [TRR, TRR, ADD, FMA, CVT], "t_afcm": [TRR, ADD, FMA, CVT, MX3], "tt_aff": [TRR, TRR, ADD, FMA, FMA],
        "d_aaff": [DSR, ADD, ADD, FMA, FMA], "dd_aff": [DSR, DSR, ADD, FMA, FMA], "d_e": [DSR, EXP], "d_eaa": [DSR, EXP, ADD, ADD],
        "tt_aa": [TRR, TRR, ADD, ADD], "tt_a": [TRR, TRR, ADD], "tt": [TRR, TRR], "t_aaaa": [TRR, ADD, ADD, ADD, ADD],
        "pk": [PKA], "pkpk": [PKA, PKA], "e_pkpk": [EXP, PKA, PKA], "pk4": [PKA] * 4,
        "m_dma": [M0S, DMA3], "m_dma_e": [M0S, DMA3, EXP], "m_dma_eaa": [M0S, DMA3, EXP, ADD, ADD], "m_dma_aaf": [M0S, DMA3, ADD, ADD, FMA],
        "e_aa_m_dma": [EXP, ADD, ADD, M0S, DMA3],
    }
    for nm, pat in pats.items():
        out.append(MB(f"mb_{nm}", (lambda u, pat=pat: [f(u, k) for k, f in enumerate(pat)])))
    # beside the 64-cycle fp8 MFMA (a8): unit costs and orderings of a run of the step's fillers (4 exp, 4 fma, 2 cvt, 2 max3, 1 read)
    C8 = lambda u, k: I("v_cvt_pk_fp8_f32", V(r(u, k)), V(r(u, (k + 1) % 8)), V(r(u, (k + 2) % 8)), op_sel=(0, 0, k & 1))
    C8D = lambda u, k: I("v_cvt_pk_fp8_f32", V(r(u, 0)), V(r(u, (k + 1) % 8)), V(r(u, (k + 2) % 8)), op_sel=(0, 0, k & 1))   # same destination
    MULS = lambda u, k: I("v_mul_f32", V(r(u, k)), S(21), V(r(u, k)))      # x = c * s' (VOP2, scalar c): what is left of the fma if -m rides in the MFMA's C operand
    CS8 = lambda u, k: I("v_cvt_scalef32_pk_fp8_f32", V(r(u, k)), V(r(u, (k + 1) % 8)), V(r(u, (k + 2) % 8)), 1.0, op_sel=(0, 0, 0, k & 1))
    CH8 = lambda u, k: I("v_cvt_scalef32_pk_fp8_f16", V(r(u, k)), V(r(u, (k + 1) % 8)), 1.0)
    CHF = lambda u, k: I("v_cvt_pk_f16_f32", V(r(u, k)), V(r(u, (k + 1) % 8)), V(r(u, (k + 2) % 8)))
    TR8 = lambda u, k: I("ds_read_b64_tr_b8", V(64 + 2 * ((8 * u + k) % 16), 2), V(201), offset=512 * ((8 * u + k) % 32))
    f8pats = {
        "none": [], "e4": [EXP] * 4, "e8": [EXP] * 8, "e12": [EXP] * 12, "f8": [FMA] * 8, "f12": [FMA] * 12, "f16": [FMA] * 16,
        "c4": [C8] * 4, "c8": [C8] * 8, "c12": [C8] * 12, "c8d": [C8D] * 8, "cb8": [CVT] * 8, "m8": [MX3] * 8, "m12": [MX3] * 12,
        "t4": [TR8] * 4, "t8": [TR8] * 8, "cs4": [CS8] * 4, "cs8": [CS8] * 8, "cs12": [CS8] * 12, "ch8": [CH8] * 8, "chf8": [CHF, CH8] * 4,
        "mix_il_cs": [TR8, EXP, FMA, CS8, MX3, EXP, FMA, EXP, FMA, CS8, EXP, FMA, MX3],
        "mul8": [MULS] * 8, "mul12": [MULS] * 12, "mul16": [MULS] * 16, "em4": [EXP, MULS] * 4, "em6": [EXP, MULS] * 6, "em8": [EXP, MULS] * 8,
        "mix_il_mul": [TR8, EXP, MULS, C8, MX3, EXP, MULS, EXP, MULS, C8, EXP, MULS, MX3],
        "mix_il_mul_x2": [TR8, EXP, MULS, C8, MX3, EXP, MULS, EXP, MULS, C8, EXP, MULS, MX3] * 2,
        "mix_il": [TR8, EXP, FMA, C8, MX3, EXP, FMA, EXP, FMA, C8, EXP, FMA, MX3],
        "mix_grp": [TR8, EXP, EXP, EXP, EXP, FMA, FMA, FMA, FMA, C8, C8, MX3, MX3],
        "mix_il_nocv": [TR8, EXP, FMA, MX3, EXP, FMA, EXP, FMA, EXP, FMA, MX3],
        "mix_il_x2": [TR8, EXP, FMA, C8, MX3, EXP, FMA, EXP, FMA, C8, EXP, FMA, MX3] * 2,
        "ef4": [EXP, FMA] * 4, "ef6": [EXP, FMA] * 6, "ef8": [EXP, FMA] * 8, "eff4": [EXP, FMA, FMA] * 4,
    }
    for nm, pat in f8pats.items():
        out.append(MB(f"mb8_{nm}", (lambda u, pat=pat: [f(u, k) for k, f in enumerate(pat)]), f8=True))
    for nm in ("none", "e4", "e8", "f8", "mix_il", "mix_il_x2"):
        out.append(MB(f"mb8s_{nm}", (lambda u, pat=f8pats[nm]: [f(u, k) for k, f in enumerate(pat)]), f8="scale"))
    # no MFMA at all (the epilogue's regime): cycles per group of 8 instructions -> / 8 = cycles per instruction
    ACR = lambda u, k: I("v_accvgpr_read_b32", V(r(u, k)), A(16 * u + k))
    MUL = lambda u, k: I("v_mul_f32", V(r(u, k)), V(r(u, k)), V(203))
    XOR = lambda u, k: I("v_xor_b32", V(r(u, k)), 4 + k, V(200))
    LSA = lambda u, k: I("v_lshl_add_u32", V(r(u, k)), V(r(u, k)), 4, V(201))
    DSW = lambda u, k: I("ds_write_b64", V(201), V(64 + 2 * ((8 * u + k) % 16), 2), offset=1024 * ((8 * u + k) % 16))
    # stores of the epilogue (no MFMA): whole 256-byte rows, 4 rows per instruction (what the kernel does after transposing through
    # LDS) against the accumulator layout stored directly: 32 rows x 16 bytes per instruction.  V(206) / V(207): lane offsets
    STR = lambda u, k: I("buffer_store_dwordx4", V(64 + 4 * ((8 * u + k) % 8), 4), V(206), S(8, 4), 0, offen=1, offset=1024 * ((8 * u + k) % 4))
    STS = lambda u, k: I("buffer_store_dwordx2", V(64 + 2 * ((8 * u + k) % 16), 2), V(207), S(8, 4), 0, offen=1, offset=16 * ((8 * u + k) % 16))
    solo = {"st_rows8": [STR] * 8, "st_scat8": [STS] * 8, "acr8": [ACR] * 8, "mul8": [MUL] * 8, "cvt8": [CVT] * 8, "xor8": [XOR] * 8, "lsa8": [LSA] * 8, "dsw8": [DSW] * 8,
            "fma8": [FMA] * 8, "exp8": [EXP] * 8, "acr_mul": [ACR, MUL] * 4, "mul_cvt": [MUL, CVT] * 4, "pk8": [PKA] * 8,
            "dsr8": [DSR] * 8, "mix": [ACR, MUL, CVT, XOR, ACR, MUL, LSA, DSW]}
    for nm, pat in solo.items():
        out.append(MB(f"mb_solo_{nm}", (lambda u, pat=pat: [f(u, k) for k, f in enumerate(pat)]), mfma=False))
    # alternating gaps: even gaps carry the exps, odd gaps the plain VALU
    def alt(u):
        if u & 1:
            return [ADD(u, 0), ADD(u, 1), FMA(u, 2), FMA(u, 3), ADD(u, 4), FMA(u, 5)]
        return [EXP(u, 0), EXP(u, 1), ADD(u, 2), FMA(u, 3)]
    out.append(MB("mb_alt_2e2v_6v", alt))
    def alt2(u):
        if u & 1:
            return [ADD(u, 0), ADD(u, 1), FMA(u, 2), FMA(u, 3), ADD(u, 4), FMA(u, 5), CVT(u, 6)]
        return [EXP(u, 0), EXP(u, 1), EXP(u, 2)]
    out.append(MB("mb_alt_3e_7v", alt2))
    return out


def main(argv=None):
    out = argv[0] if argv else "mb.s"
    ks = cases()
    for k in ks:
        k.prog = [x for x in k.prog if x is not None]
    with open(out, "w") as f:
        f.write(module_text(ks))
    with open(out + ".names", "w") as f:
        f.write("\n".join(k.name for k in ks) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
