#!/usr/bin/env python3
"""Power / clock probe of the two bf16 MFMA shapes UNDER THE FILLER LOAD of the a64 kernel (MI355X_MICROARCH.md, 'DVFS give-back'
item 7: the chip holds a higher clock on v_mfma_f32_16x16x32 than on 32x32x16 -- but the 16x16 form has half the issue room
per MFMA).  Each kernel runs  iters x [ one "unit" = 8 MFMAs 32x32x16 or 16 MFMAs 16x16x32 (the same FLOPs) with the same
multiset of fillers spread over its gaps ] on random operands, one wave per SIMD (512 registers), and writes per wave the
s_memtime and s_memrealtime deltas of the loop: cycles per unit, clock, TFLOP/s (scripts/probes/pw_run.hip).

Filler multiset per unit = 1/9 of a 64-key tile step of the a64 kernel (72 MFMA gaps: 64 exp2, 64 fma, 36 max3, 32 cvt, 32
transposing LDS reads, 16 row reads, scalar work): 8 exp, 8 fma, 4 max3, 4 cvt, 4 ds_read_b64_tr_b16, 2 ds_read_b128, 2 s_add.
"""
from __future__ import annotations

import sys

from .isa import A, I, Label, S, V, label, waitcnt
from .fa2_a64_gen import Gen, module_text


class PW:
    def __init__(self, name, shape, fill, lds=True):
        """shape: "32" or "16";  fill: None (bare MFMAs), "full" (the multiset above), "noexp", "half" (every second unit's worth)"""
        self.name = name
        self.prog = []
        e = self.prog.append
        e(I("s_load_dwordx4", S(4, 4), S(0, 2), 0))
        e(I("s_load_dword", S(16), S(0, 2), 16))
        e(I("v_and_b32", V(200), 63, V(0)))
        e(I("v_lshlrev_b32", V(201), 4, V(200)))
        e(I("v_lshrrev_b32", V(202), 6, V(0)))
        e(I("s_nop", 1))
        e(I("v_readfirstlane_b32", S(20), V(202)))
        e(I("s_nop", 4))
        # random operands: v[128:191] <- hash(lane, k), as bf16 pairs with random sign and mantissa, exponent 126 (|x| in [0.5, 1))
        e(I("v_mov_b32", V(203), 0x9E3779B1 - (1 << 32)))
        e(I("v_mul_lo_u32", V(204), V(0), V(203)))
        for k in range(128, 192):
            e(I("v_add_u32", V(k), 0x85EBCA6B * (k - 127) & 0x7FFFFFFF, V(204)))
            e(I("v_lshrrev_b32", V(205), 15, V(k)))
            e(I("v_xor_b32", V(k), V(k), V(205)))
            e(I("v_mul_lo_u32", V(k), V(k), V(203)))
            e(I("v_lshrrev_b32", V(205), 13, V(k)))
            e(I("v_xor_b32", V(k), V(k), V(205)))
            e(I("v_and_b32", V(k), 0x807F807F - (1 << 32), V(k)))
            e(I("v_or_b32", V(k), 0x3F003F00, V(k)))
        for k in range(0, 128):
            e(I("v_mov_b32", V(k), 0.5 if k % 3 else 1.0))
        for k in range(0, 256):
            e(I("v_accvgpr_write_b32", A(k), 0))
        e(I("v_mov_b32", V(206), 0.25))
        e(waitcnt(lgkmcnt=0))
        e(I("s_mov_b32", S(21), 1.4426950408889634))
        e(I("s_mov_b32", S(23), 0))
        # LDS: every lane writes its operand registers once so the reads return random data too
        for k in range(16):
            e(I("ds_write_b128", V(201), V(128 + 4 * k, 4), offset=1024 * k))
        e(waitcnt(lgkmcnt=0))
        e(I("s_barrier"))
        e(I("s_memtime", S(12, 2)))
        e(I("s_memrealtime", S(24, 2)))
        e(waitcnt(lgkmcnt=0))
        lp = f".Lpw_{name}_loop"
        e(label(lp))
        r = lambda k: 64 + (k % 56)      # arch VGPR 64..119: filler data (never an MFMA operand)
        EXP = lambda k: I("v_exp_f32", V(r(k)), V(r(k + 1)))
        FMA = lambda k: I("v_fma_f32", V(r(k)), V(r(k + 1)), S(21), -V(206))
        MX3 = lambda k: I("v_max3_f32", V(120 + (k & 3)), V(120 + (k & 3)), V(r(k)), V(r(k + 1)))
        CVT = lambda k: I("v_cvt_pk_bf16_f32", V(r(k)), V(r(k + 1)), V(r(k + 2)))
        # LDS reads land in v[192:199] / a-registers are busy: use arch registers 192..199 (not operands)
        TRR = lambda k: I("ds_read_b64_tr_b16", V(192 + 2 * (k % 4), 2), V(201), offset=512 * (k % 32))
        DSR = lambda k: I("ds_read_b128", V(208 + 4 * (k % 4), 4), V(201), offset=1024 * (k % 16))
        SAL = lambda k: I("s_add_u32", S(23), S(23), 1)
        if not lds:
            TRR = DSR = lambda k: None
        # the multiset as 16 half-gap groups (one per 16x16x32 MFMA; two consecutive ones per 32x32x16 MFMA)
        if fill is None:
            groups = [[] for _ in range(16)]
        elif fill in ("split", "split_nolds", "nomax"):
            # exp alone in the even half-gaps; fma + (max3 | cvt) in the odd ones; LDS reads lead the odd ones
            groups = []
            for v in range(16):
                g = []
                if v % 2 == 0:
                    g.append(EXP(3 * v))
                    if v in (6, 14):
                        g.append(SAL(v))
                else:
                    if fill == "split":
                        if v % 4 == 1:
                            g.append(TRR(v))
                        if v in (3, 11):
                            g.append(DSR(v))
                    g.append(FMA(3 * v))
                    if v % 4 == 1:
                        if fill != "nomax":
                            g.append(MX3(5 * v))
                    else:
                        g.append(CVT(5 * v))
                groups.append([x for x in g if x is not None])
        else:
            groups = []
            for v in range(16):
                g = []
                if v % 4 == 0:
                    g.append(TRR(v))
                if v in (2, 10):
                    g.append(DSR(v))
                if v % 2 == 0:
                    if fill != "noexp":
                        g.append(EXP(3 * v))
                    if v % 4 == 2:
                        g.append(SAL(v)) if v in (6, 14) else None
                else:
                    g.append(FMA(3 * v))
                    g.append(MX3(5 * v) if v % 4 == 1 else CVT(5 * v))
                groups.append([x for x in g if x is not None])
        if shape == "32":
            for u in range(8):
                acc = A(16 * u, 16)
                e(I("v_mfma_f32_32x32x16_bf16", acc, V(128 + 8 * (u % 4), 4), V(160 + 4 * (u % 4), 4), acc))
                for x in sorted(groups[2 * u] + groups[2 * u + 1], key=lambda i: 0 if i.op.startswith("ds_") else 1 if i.op == "v_exp_f32" else 2):
                    e(x)
        elif shape == "mix":
            # half the FLOPs on each shape (QK^T on 32x32x16, P.V on 16x16x32 -- the hybrid kernel), plus the two lane-group
            # exchanges that turn packed P of the 32x32 layout into 16x16x32 operands
            SWP = lambda k: I("v_permlane16_swap_b32", V(r(k)), V(r(k + 7)))
            for u in range(4):
                acc = A(16 * u, 16)
                e(I("v_mfma_f32_32x32x16_bf16", acc, V(128 + 8 * (u % 4), 4), V(160 + 4 * (u % 4), 4), acc))
                for x in sorted(groups[2 * u] + groups[2 * u + 1], key=lambda i: 0 if i.op.startswith("ds_") else 1 if i.op == "v_exp_f32" else 2):
                    e(x)
            for v in range(8, 16):
                acc = A(64 + 4 * v, 4)
                e(I("v_mfma_f32_16x16x32_bf16", acc, V(128 + 4 * (v // 4), 4), V(160 + 4 * (v % 8), 4), acc))
                for x in sorted(groups[v], key=lambda i: 0 if i.op.startswith("ds_") else 1 if i.op == "v_exp_f32" else 2):
                    e(x)
                if fill is not None and v in (9, 13):
                    e(SWP(11 * v))
        else:
            for v in range(16):
                acc = A(4 * v, 4)
                # (four consecutive MFMAs share their A operand, as the 16x16 form of the attention kernel would: one K / V^T
                # fragment against the four 16-row query blocks)
                e(I("v_mfma_f32_16x16x32_bf16", acc, V(128 + 4 * (v // 4), 4), V(160 + 4 * (v % 8), 4), acc))
                for x in sorted(groups[v], key=lambda i: 0 if i.op.startswith("ds_") else 1 if i.op == "v_exp_f32" else 2):
                    e(x)
        e(I("s_sub_u32", S(16), S(16), 1))
        e(I("s_cmp_lg_u32", S(16), 0))
        e(I("s_cbranch_scc1", Label(lp)))
        e(waitcnt(vmcnt=0, lgkmcnt=0))
        e(I("s_memtime", S(14, 2)))
        e(I("s_memrealtime", S(26, 2)))
        e(waitcnt(lgkmcnt=0))
        e(I("s_sub_u32", S(14), S(14), S(12)))
        e(I("s_sub_u32", S(26), S(26), S(24)))
        # out[2 * (wg * 4 + wave)] = cycles, [+1] = realtime ticks (100 MHz)
        e(I("s_lshl_b32", S(17), S(2), 2))
        e(I("s_add_u32", S(17), S(17), S(20)))
        e(I("s_lshl_b32", S(17), S(17), 3))
        e(I("v_mov_b32", V(204), S(17)))
        e(I("v_mov_b32", V(210), S(14)))
        e(I("v_mov_b32", V(211), S(26)))
        e(I("global_store_dwordx2", V(204), V(210, 2), S(4, 2)))
        e(waitcnt(vmcnt=0))
        e(I("s_endpgm"))

    def text(self):
        g = Gen.__new__(Gen)
        g.name, g.prog = self.name, self.prog
        return Gen.text(g)

    def metadata(self):
        g = Gen.__new__(Gen)
        g.name = self.name
        return Gen.metadata(g)


def cases():
    out = []
    for shape in ("32", "16", "mix"):
        out.append(PW(f"pw_{shape}_bare", shape, None))
        out.append(PW(f"pw_{shape}_full", shape, "full"))
        out.append(PW(f"pw_{shape}_noexp", shape, "noexp"))
        out.append(PW(f"pw_{shape}_valu", shape, "full", lds=False))
        out.append(PW(f"pw_{shape}_split", shape, "split"))
        out.append(PW(f"pw_{shape}_splitv", shape, "split_nolds"))
        out.append(PW(f"pw_{shape}_nomax", shape, "nomax"))
    return out


def main(argv=None):
    out = argv[0] if argv else "pw.s"
    ks = cases()
    with open(out, "w") as f:
        f.write(module_text(ks))
    with open(out + ".names", "w") as f:
        f.write("\n".join(k.name for k in ks) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
