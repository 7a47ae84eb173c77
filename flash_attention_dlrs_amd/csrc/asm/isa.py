"""Minimal gfx950 instruction layer for the generated FA-2 forward kernel.

An `Inst` is (opcode, operands, modifiers).  It can be printed as assembler text (`text()`) and executed by the
wave64 emulator in emu.py.  Only the instructions the generator uses are modelled.

Operands:
    V(n) / V(n, cnt)   arch VGPR or range          A(n) / A(n, cnt)   accumulator VGPR or range
    S(n) / S(n, cnt)   SGPR or range               VCC, EXEC, M0, SCC  special registers
    int / float        inline constant or literal  Label('name')       branch target
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field


@dataclass(frozen=True)
class Reg:
    kind: str  # 'v', 'a', 's', 'vcc', 'exec', 'm0'
    idx: int = 0
    cnt: int = 1
    neg: bool = False  # VOP3 source modifier
    abs_: bool = False

    def __str__(self):
        if self.kind in ("vcc", "exec", "m0"):
            base = self.kind
        elif self.cnt == 1:
            base = f"{self.kind}{self.idx}"
        else:
            base = f"{self.kind}[{self.idx}:{self.idx + self.cnt - 1}]"
        if self.abs_:
            base = f"|{base}|"
        if self.neg:
            base = "-" + base
        return base

    def __neg__(self):
        return Reg(self.kind, self.idx, self.cnt, not self.neg, self.abs_)

    def sub(self, off, cnt=1):
        assert 0 <= off and off + cnt <= self.cnt, (self, off, cnt)
        return Reg(self.kind, self.idx + off, cnt)

    def regs(self):
        """flat list of (kind, index) this operand covers"""
        if self.kind in ("vcc",):
            return [("s", 106), ("s", 107)]
        if self.kind == "exec":
            return [("s", 126), ("s", 127)]
        if self.kind == "m0":
            return [("s", 124)]
        return [(self.kind, self.idx + k) for k in range(self.cnt)]


def V(n, cnt=1):
    assert 0 <= n and n + cnt <= 256, (n, cnt)
    return Reg("v", n, cnt)


def A(n, cnt=1):
    assert 0 <= n and n + cnt <= 256, (n, cnt)
    return Reg("a", n, cnt)


def S(n, cnt=1):
    assert 0 <= n and n + cnt <= 102, (n, cnt)
    return Reg("s", n, cnt)


VCC = Reg("vcc", 0, 2)
EXEC = Reg("exec", 0, 2)
M0 = Reg("m0")


@dataclass(frozen=True)
class Label:
    name: str

    def __str__(self):
        return self.name


def f2u(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", x))[0]


def u2f(u: int) -> float:
    return struct.unpack("<f", struct.pack("<I", u & 0xFFFFFFFF))[0]


def fmt_operand(o) -> str:
    if isinstance(o, (Reg, Label)):
        return str(o)
    if isinstance(o, bool):
        raise TypeError(o)
    if isinstance(o, int):
        if -16 <= o <= 64:
            return str(o)
        return hex(o & 0xFFFFFFFF)
    if isinstance(o, float):
        if o in (0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0):
            return repr(o)
        if o == 0.0:
            return "0"
        return hex(f2u(o))
    raise TypeError(o)


# opcode classes (for the hazard checker and the issue-cost model)
MFMA_OPS = {"v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x32_f16",
            "v_mfma_f32_32x32x64_f8f6f4", "v_mfma_f32_16x16x128_f8f6f4",
            # block-scaled form: D = C + (A 2^(sa - 127)) (B 2^(sb - 127)), sa / sb = a byte of the two trailing VGPR operands (E8M0,
            # one per lane: its row / column); op_sel / op_sel_hi [a, b, 0] pick the byte (bit 0 / bit 1 of its index)
            "v_mfma_scale_f32_32x32x64_f8f6f4"}
TRANS_OPS = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"}
DS_OPS = {"ds_read_b128", "ds_read_b64_tr_b16", "ds_read_b64_tr_b8", "ds_write_b64", "ds_read_b64", "ds_write_b128", "ds_read_b32", "ds_write_b32"}
VMEM_OPS = {"global_store_dwordx2", "buffer_load_dwordx4", "buffer_store_dwordx4", "buffer_store_short", "buffer_store_dword", "global_store_dword",
            "global_store_dwordx4", "buffer_load_dword", "buffer_store_byte"}
SMEM_OPS = {"s_memrealtime", "s_load_dword", "s_load_dwordx2", "s_load_dwordx4", "s_load_dwordx8", "s_load_dwordx16", "s_memtime"}


@dataclass
class Inst:
    op: str
    ops: tuple = ()
    mods: dict = field(default_factory=dict)
    comment: str = ""
    tag: str = ""  # free-form: which logical operation this instruction belongs to (debugging, checkers)

    # ---- classification
    @property
    def is_mfma(self):
        return self.op in MFMA_OPS

    @property
    def is_valu(self):
        return self.op.startswith("v_") and not self.is_mfma

    @property
    def is_salu(self):
        return self.op.startswith("s_") and self.op not in SMEM_OPS

    @property
    def is_label(self):
        return self.op == ".label"

    def text(self) -> str:
        if self.op == ".label":
            return f"{self.ops[0]}:"
        if self.op == ".comment":
            return f"  ; {self.comment}"
        if self.op == "s_waitcnt":
            parts = []
            if "vmcnt" in self.mods:
                parts.append(f"vmcnt({self.mods['vmcnt']})")
            if "lgkmcnt" in self.mods:
                parts.append(f"lgkmcnt({self.mods['lgkmcnt']})")
            s = "  s_waitcnt " + " ".join(parts)
        else:
            s = "  " + self.op
            if self.ops:
                s += " " + ", ".join(fmt_operand(o) for o in self.ops)
            for k, v in self.mods.items():
                if k == "offset":
                    if v:
                        s += f" offset:{v}"
                elif k in ("offen", "lds", "glc", "sc0", "sc1", "nt"):
                    if v:
                        s += f" {k}"
                elif k == "op_sel_hi":   # VOP3P: which half of each source feeds the HIGH result ([1,0]: src1's low word twice)
                    s += " op_sel_hi:[" + ",".join(str(int(x)) for x in v) + "]"
                elif k == "op_sel":      # VOP3 (v_cvt_pk_fp8_f32: [0,0,1] = the HIGH half of the destination is written)
                    if any(v):
                        s += " op_sel:[" + ",".join(str(int(x)) for x in v) + "]"
                elif k in ("cbsz", "blgp"):   # v_mfma_*_f8f6f4: the formats of A / B (0 = fp8 e4m3, 1 = bf8 e5m2)
                    if v:
                        s += f" {k}:{v}"
                elif k == "off":  # global_* with no saddr
                    pass
                else:
                    raise KeyError(k)
        if self.comment:
            s = f"{s:<72}; {self.comment}"
        return s

    # ---- register read/write sets (for the hazard checker)
    def defs_uses(self):
        """returns (defs, uses): lists of (kind, idx)"""
        op, o = self.op, self.ops
        d, u = [], []

        def R(x):
            return x.regs() if isinstance(x, Reg) else []

        if op in (".label", ".comment", "s_waitcnt", "s_barrier", "s_nop", "s_endpgm", "s_branch", "s_setprio", "s_sleep"):
            return d, u
        if op.startswith("s_cbranch_scc"):
            return d, [("scc", 0)]
        if op.startswith("s_cbranch_vcc"):
            return d, VCC.regs()
        if op.startswith("s_cmp_") or op.startswith("s_bitcmp"):
            return [("scc", 0)], R(o[0]) + R(o[1])
        if op.startswith("s_load_") or op in ("s_memtime", "s_memrealtime"):
            return R(o[0]), (R(o[1]) if len(o) > 1 else [])
        if self.is_mfma:
            return R(o[0]), [r for x in o[1:] for r in R(x)]
        if op.startswith("ds_read"):
            return R(o[0]), R(o[1])
        if op.startswith("ds_write"):
            return [], R(o[0]) + R(o[1])
        if op.startswith("buffer_load"):
            if self.mods.get("lds"):
                return [], R(o[0]) + R(o[1]) + R(o[2]) + M0.regs()
            return R(o[0]), R(o[1]) + R(o[2]) + R(o[3])
        if op.startswith("buffer_store"):
            return [], R(o[0]) + R(o[1]) + R(o[2]) + R(o[3])
        if op.startswith("global_store"):
            return [], R(o[0]) + R(o[1]) + (R(o[2]) if len(o) > 2 else [])
        if op in ("v_permlane32_swap_b32", "v_permlane16_swap_b32"):
            return R(o[0]) + R(o[1]), R(o[0]) + R(o[1])
        if op.startswith("v_cmp_"):
            return R(o[0]), R(o[1]) + R(o[2])
        if op == "v_cndmask_b32":
            return R(o[0]), R(o[1]) + R(o[2]) + R(o[3])
        if op == "v_readfirstlane_b32":
            return R(o[0]), R(o[1])
        if op in ("v_cvt_pk_fp8_f32", "v_cvt_pk_bf8_f32"):
            return R(o[0]), R(o[0]) + R(o[1]) + R(o[2])
        if op in ("v_cvt_scalef32_pk_fp8_f32", "v_cvt_scalef32_pk_bf8_f32"):     # (gfx950; the fourth operand: the scale, an f32)
            return R(o[0]), R(o[0]) + R(o[1]) + R(o[2]) + R(o[3])
        # generic: first operand is the destination
        if op.startswith("v_") or op.startswith("s_"):
            d = R(o[0])
            for x in o[1:]:
                u += R(x)
            if op in ("s_add_u32", "s_sub_u32", "s_addc_u32", "s_and_b32", "s_or_b32", "s_lshl_b32", "s_lshr_b32", "s_and_b64",
                      "s_or_b64", "s_lshr_b64", "s_min_u32", "s_max_u32", "s_sub_i32", "s_add_i32", "s_andn2_b64", "s_xor_b32", "s_bfe_u32"):
                d = d + [("scc", 0)]
            if op in ("s_addc_u32", "s_cselect_b32", "s_cselect_b64"):
                u = u + [("scc", 0)]
            return d, u
        raise NotImplementedError(op)


def I(op, *ops, comment="", tag="", **mods):
    return Inst(op, tuple(ops), dict(mods), comment, tag)


def label(name):
    return Inst(".label", (Label(name),))


def comment(text):
    return Inst(".comment", (), {}, text)


def waitcnt(vmcnt=None, lgkmcnt=None, comment=""):
    m = {}
    if vmcnt is not None:
        m["vmcnt"] = vmcnt
    if lgkmcnt is not None:
        m["lgkmcnt"] = lgkmcnt
    return Inst("s_waitcnt", (), m, comment)


# issue cost in cycles of one wave's stream on one SIMD (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost')
def issue_cost(inst: Inst) -> int:
    if inst.is_label or inst.op == ".comment":
        return 0
    if inst.is_mfma:
        return 8
    if inst.op in TRANS_OPS:
        return 8
    if inst.op == "s_nop":
        return 4 * (inst.ops[0] + 1)
    if inst.op.startswith("buffer_load") and inst.mods.get("lds"):
        return 16
    return 4
