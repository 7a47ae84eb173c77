"""Wave64 emulator for the instruction subset of isa.py -- TEST INFRASTRUCTURE for the kernel generator.

It executes the generated instruction list for one workgroup (several waves, released together at s_barrier) on numpy
arrays, so that indexing, register allocation, LDS layouts, masks and the software pipeline can be checked in this
container (no GPU).  It is not a CPU path of the product: nothing outside tests/ and the generator's self-check uses it.

Asynchrony is modelled pessimistically so that missing waits show up as wrong results:
  * a load's destination registers (or, for LDS-DMA, its LDS bytes) are POISONED at issue and receive the data only
    when an s_waitcnt retires the operation (vmcnt / lgkmcnt queues, in order);
  * data is sampled at issue time.
Hardware wait-state hazards (MFMA -> VALU etc.) are NOT modelled here; see check.py.
"""
from __future__ import annotations

import numpy as np

from .isa import Inst, Reg, Label, f2u

POISON32 = np.uint32(0x7FC17FC1)  # NaN as f32, NaN as bf16 pairs
LDS_BYTES = 160 * 1024

np.seterr(all="ignore")


def bf16_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_rne(f):
    u = np.asarray(f, np.float32).view(np.uint32)
    nan = np.isnan(np.asarray(f, np.float32))
    r = ((u + np.uint32(0x7FFF) + ((u >> 16) & np.uint32(1))) >> 16).astype(np.uint16)
    r = np.where(nan, ((u >> 16) | np.uint32(0x40)).astype(np.uint16), r)
    return r


class Memory:
    """flat device memory made of named buffers at fake addresses"""

    def __init__(self):
        self.bufs = []  # (base, array uint8)
        self.next = 0x7F00_0000_0000

    def alloc(self, arr_u8: np.ndarray) -> int:
        base = self.next
        self.bufs.append((base, arr_u8))
        self.next += (len(arr_u8) + 0xFFFF) & ~0xFFFF
        self.next += 0x10000  # guard gap
        return base

    def _find(self, addr, n):
        for base, arr in self.bufs:
            if base <= addr and addr + n <= base + len(arr):
                return arr, addr - base
        raise RuntimeError(f"memory access fault: {addr:#x} +{n}")

    def read(self, addr, n):
        arr, off = self._find(addr, n)
        return arr[off:off + n].copy()

    def write(self, addr, data_u8):
        arr, off = self._find(addr, len(data_u8))
        arr[off:off + len(data_u8)] = data_u8


class Wave:
    def __init__(self, wg, wid):
        self.wg = wg
        self.wid = wid
        self.s = np.zeros(128, np.uint32)
        self.scc = 0
        self.v = np.zeros((512, 64), np.uint32)
        self.v[:] = POISON32
        self.pc = 0
        self.vm_q = []
        self.lgkm_q = []
        self.state = "run"
        self.s[126] = 0xFFFFFFFF
        self.s[127] = 0xFFFFFFFF
        self.n_exec = 0

    # ---------------------------------------------------------------- operand access
    def exec_mask(self):
        e = int(self.s[126]) | (int(self.s[127]) << 32)
        return np.array([(e >> l) & 1 for l in range(64)], bool)

    def sidx(self, r: Reg):
        if r.kind == "s":
            return r.idx
        return {"vcc": 106, "exec": 126, "m0": 124}[r.kind]

    def rd_s(self, o) -> int:
        if isinstance(o, Reg):
            assert o.kind in ("s", "vcc", "exec", "m0"), o
            i = self.sidx(o)
            if o.cnt == 1 or o.kind == "m0":
                return int(self.s[i])
            return int(self.s[i]) | (int(self.s[i + 1]) << 32)
        if isinstance(o, float):
            return f2u(o)
        return int(o) & 0xFFFFFFFF

    def wr_s(self, r: Reg, val: int):
        i = self.sidx(r)
        self.s[i] = val & 0xFFFFFFFF
        if r.cnt == 2 and r.kind != "m0":
            self.s[i + 1] = (val >> 32) & 0xFFFFFFFF

    def vrow(self, r: Reg, k=0):
        return (r.idx + k) + (256 if r.kind == "a" else 0)

    def rd_v(self, o, k=0) -> np.ndarray:
        """uint32[64]"""
        if isinstance(o, Reg):
            if o.kind in ("v", "a"):
                return self.v[self.vrow(o, k)].copy()
            return np.full(64, self.rd_s(o) & 0xFFFFFFFF, np.uint32)
        if isinstance(o, float):
            return np.full(64, f2u(o), np.uint32)
        return np.full(64, int(o) & 0xFFFFFFFF, np.uint32)

    def rd_f(self, o) -> np.ndarray:
        x = self.rd_v(o).view(np.float32)
        if isinstance(o, Reg):
            if o.abs_:
                x = np.abs(x)
            if o.neg:
                x = -x
        elif isinstance(o, int) and not isinstance(o, bool):
            # integer inline constants used as float operands are the integer's bit pattern (0 stays 0.0)
            pass
        return x

    def wr_v(self, r: Reg, val: np.ndarray, k=0, masked=True):
        val = np.asarray(val)
        if val.dtype == np.float32:
            val = val.view(np.uint32)
        val = val.astype(np.uint32)
        row = self.vrow(r, k)
        if masked:
            m = self.exec_mask()
            self.v[row] = np.where(m, val, self.v[row])
        else:
            self.v[row] = val

    # ---------------------------------------------------------------- queues
    def retire(self, q, n):
        while len(q) > n:
            fn = q.pop(0)
            if fn is not None:
                fn()


class Workgroup:
    def __init__(self, prog: list, mem: Memory, nwaves: int, wg_id: int, kernarg_addr: int):
        self.prog = prog
        self.mem = mem
        self.lds = np.zeros(LDS_BYTES, np.uint8)
        self.lds.view(np.uint32)[:] = POISON32
        self.labels = {}
        for i, ins in enumerate(prog):
            if ins.op == ".label":
                self.labels[ins.ops[0].name] = i
        self.waves = []
        for w in range(nwaves):
            wv = Wave(self, w)
            wv.s[0] = kernarg_addr & 0xFFFFFFFF
            wv.s[1] = kernarg_addr >> 32
            wv.s[2] = wg_id
            wv.v[0] = np.arange(64, dtype=np.uint32) + 64 * w
            self.waves.append(wv)

    def run(self, order=None, max_steps=50_000_000):
        order = list(order or range(len(self.waves)))
        steps = 0
        while True:
            progressed = False
            for w in order:
                wv = self.waves[w]
                while wv.state == "run":
                    self.step(wv)
                    steps += 1
                    progressed = True
                    if steps > max_steps:
                        raise RuntimeError("emulation step limit")
            states = {wv.state for wv in self.waves}
            if states == {"done"}:
                return steps
            if "barrier" in states and states <= {"barrier"}:
                for wv in self.waves:
                    wv.state = "run"
                continue
            if states <= {"barrier", "done"} and "barrier" in states:
                raise RuntimeError("deadlock: some waves ended while others wait at a barrier")
            if not progressed:
                raise RuntimeError("no progress")

    # ---------------------------------------------------------------- one instruction
    def step(self, w: Wave):
        ins = self.prog[w.pc]
        w.pc += 1
        op = ins.op
        if op in (".label", ".comment"):
            return
        w.n_exec += 1
        fn = getattr(self, "x_" + op, None)
        if fn is None:
            raise NotImplementedError(op)
        fn(w, ins)

    def jump(self, w, lab: Label):
        w.pc = self.labels[lab.name]

    # ---- program control
    def x_s_endpgm(self, w, i):
        w.retire(w.vm_q, 0)
        w.retire(w.lgkm_q, 0)
        w.state = "done"

    def x_s_barrier(self, w, i):
        w.state = "barrier"

    def x_s_nop(self, w, i):
        pass

    def x_s_setprio(self, w, i):
        pass

    def x_s_sleep(self, w, i):
        pass

    def x_s_waitcnt(self, w, i):
        if "vmcnt" in i.mods:
            w.retire(w.vm_q, i.mods["vmcnt"])
        if "lgkmcnt" in i.mods:
            n = i.mods["lgkmcnt"]
            if n > 0 and any(getattr(f, "smem", False) for f in w.lgkm_q if f is not None):
                raise RuntimeError("counted lgkmcnt wait with SMEM outstanding (returns out of order)")
            w.retire(w.lgkm_q, n)

    def x_s_branch(self, w, i):
        self.jump(w, i.ops[0])

    def x_s_cbranch_scc0(self, w, i):
        if not w.scc:
            self.jump(w, i.ops[0])

    def x_s_cbranch_scc1(self, w, i):
        if w.scc:
            self.jump(w, i.ops[0])

    def x_s_cbranch_vccz(self, w, i):
        if w.rd_s(Reg("vcc", 0, 2)) == 0:
            self.jump(w, i.ops[0])

    def x_s_cbranch_vccnz(self, w, i):
        if w.rd_s(Reg("vcc", 0, 2)) != 0:
            self.jump(w, i.ops[0])

    # ---- SALU
    def x_s_mov_b32(self, w, i):
        w.wr_s(i.ops[0], w.rd_s(i.ops[1]) & 0xFFFFFFFF)

    def x_s_mov_b64(self, w, i):
        v = i.ops[1]
        val = w.rd_s(v) if isinstance(v, Reg) else (int(v) & 0xFFFFFFFFFFFFFFFF if v >= 0 else int(v) & 0xFFFFFFFFFFFFFFFF)
        w.wr_s(i.ops[0], val)

    def _sbin(self, w, i, f, setscc=None):
        a, b = w.rd_s(i.ops[1]) & 0xFFFFFFFF, w.rd_s(i.ops[2]) & 0xFFFFFFFF
        r = f(a, b)
        if setscc is not None:
            w.scc = int(setscc(a, b, r))
        w.wr_s(i.ops[0], r & 0xFFFFFFFF)

    def x_s_add_u32(self, w, i):
        self._sbin(w, i, lambda a, b: a + b, lambda a, b, r: r > 0xFFFFFFFF)

    def x_s_addc_u32(self, w, i):
        c = w.scc
        self._sbin(w, i, lambda a, b: a + b + c, lambda a, b, r: r > 0xFFFFFFFF)

    def x_s_sub_u32(self, w, i):
        self._sbin(w, i, lambda a, b: a - b, lambda a, b, r: b > a)

    def x_s_add_i32(self, w, i):
        self._sbin(w, i, lambda a, b: a + b, lambda a, b, r: False)

    def x_s_sub_i32(self, w, i):
        self._sbin(w, i, lambda a, b: a - b, lambda a, b, r: False)

    def x_s_mul_i32(self, w, i):
        self._sbin(w, i, lambda a, b: a * b)

    def x_s_mul_hi_u32(self, w, i):
        self._sbin(w, i, lambda a, b: (a * b) >> 32)

    def x_s_lshl_b32(self, w, i):
        self._sbin(w, i, lambda a, b: a << (b & 31), lambda a, b, r: (r & 0xFFFFFFFF) != 0)

    def x_s_lshr_b32(self, w, i):
        self._sbin(w, i, lambda a, b: a >> (b & 31), lambda a, b, r: r != 0)

    def x_s_and_b32(self, w, i):
        self._sbin(w, i, lambda a, b: a & b, lambda a, b, r: r != 0)

    def x_s_or_b32(self, w, i):
        self._sbin(w, i, lambda a, b: a | b, lambda a, b, r: r != 0)

    def x_s_xor_b32(self, w, i):
        self._sbin(w, i, lambda a, b: a ^ b, lambda a, b, r: r != 0)

    def x_s_min_u32(self, w, i):
        self._sbin(w, i, lambda a, b: min(a, b), lambda a, b, r: a <= b)

    def x_s_max_u32(self, w, i):
        self._sbin(w, i, lambda a, b: max(a, b), lambda a, b, r: a >= b)

    def x_s_and_b64(self, w, i):
        r = w.rd_s(i.ops[1]) & w.rd_s(i.ops[2])
        w.scc = int(r != 0)
        w.wr_s(i.ops[0], r)

    def x_s_or_b64(self, w, i):
        r = w.rd_s(i.ops[1]) | w.rd_s(i.ops[2])
        w.scc = int(r != 0)
        w.wr_s(i.ops[0], r)

    def x_s_lshr_b64(self, w, i):
        r = w.rd_s(i.ops[1]) >> (w.rd_s(i.ops[2]) & 63)
        w.scc = int(r != 0)
        w.wr_s(i.ops[0], r)

    def x_s_cselect_b32(self, w, i):
        w.wr_s(i.ops[0], w.rd_s(i.ops[1]) if w.scc else w.rd_s(i.ops[2]))

    def x_s_cselect_b64(self, w, i):
        def rd64(o):   # (an inline integer constant is sign-extended to 64 bits)
            if isinstance(o, int) and not isinstance(o, bool):
                return o & 0xFFFFFFFFFFFFFFFF
            return w.rd_s(o)
        w.wr_s(i.ops[0], rd64(i.ops[1]) if w.scc else rd64(i.ops[2]))

    def _scmp(self, w, i, f, signed=False):
        a, b = w.rd_s(i.ops[0]) & 0xFFFFFFFF, w.rd_s(i.ops[1]) & 0xFFFFFFFF
        if signed:
            a = a - (1 << 32) if a & 0x80000000 else a
            b = b - (1 << 32) if b & 0x80000000 else b
        w.scc = int(f(a, b))

    def x_s_cmp_lg_u64(self, w, i):
        w.scc = int(w.rd_s(i.ops[0]) != w.rd_s(i.ops[1]))

    def x_s_bitcmp1_b32(self, w, i):
        w.scc = int((w.rd_s(i.ops[0]) >> (w.rd_s(i.ops[1]) & 31)) & 1)

    def x_s_cmp_lt_u32(self, w, i):
        self._scmp(w, i, lambda a, b: a < b)

    def x_s_cmp_le_u32(self, w, i):
        self._scmp(w, i, lambda a, b: a <= b)

    def x_s_cmp_gt_u32(self, w, i):
        self._scmp(w, i, lambda a, b: a > b)

    def x_s_cmp_ge_u32(self, w, i):
        self._scmp(w, i, lambda a, b: a >= b)

    def x_s_cmp_eq_u32(self, w, i):
        self._scmp(w, i, lambda a, b: a == b)

    def x_s_cmp_lg_u32(self, w, i):
        self._scmp(w, i, lambda a, b: a != b)

    def x_s_cmp_lt_i32(self, w, i):
        self._scmp(w, i, lambda a, b: a < b, True)

    def x_s_cmp_gt_i32(self, w, i):
        self._scmp(w, i, lambda a, b: a > b, True)

    def x_s_cmp_ge_i32(self, w, i):
        self._scmp(w, i, lambda a, b: a >= b, True)

    def x_s_cmp_le_i32(self, w, i):
        self._scmp(w, i, lambda a, b: a <= b, True)

    # ---- SMEM
    def _sload(self, w, i, ndw):
        base = w.rd_s(i.ops[1])
        off = int(i.ops[2])
        data = self.mem.read(base + off, 4 * ndw).view(np.uint32)
        dst = i.ops[0]

        def apply():
            for k in range(ndw):
                w.s[dst.idx + k] = data[k]
        apply.smem = True
        for k in range(ndw):
            w.s[dst.idx + k] = 0xDEADBEEF
        w.lgkm_q.append(apply)

    def x_s_memtime(self, w, i):
        dst, val = i.ops[0], w.n_exec * 4

        def apply():
            w.wr_s(dst, val)
        apply.smem = True
        w.lgkm_q.append(apply)

    def x_s_memrealtime(self, w, i):
        self.x_s_memtime(w, i)

    def x_global_store_dwordx2(self, w, i):
        vaddr, src, sbase = i.ops
        base = w.rd_s(sbase)
        off = w.rd_v(vaddr).astype(np.int64) + int(i.mods.get("offset", 0))
        m = w.exec_mask()
        d0, d1 = w.rd_v(src, 0), w.rd_v(src, 1)
        for l in range(64):
            if m[l]:
                self.mem.write(base + int(off[l]), np.array([d0[l], d1[l]], np.uint32).view(np.uint8))
        w.vm_q.append(None)

    def x_s_load_dword(self, w, i):
        self._sload(w, i, 1)

    def x_s_load_dwordx2(self, w, i):
        self._sload(w, i, 2)

    def x_s_load_dwordx4(self, w, i):
        self._sload(w, i, 4)

    def x_s_load_dwordx8(self, w, i):
        self._sload(w, i, 8)

    def x_s_load_dwordx16(self, w, i):
        self._sload(w, i, 16)

    # ---- VALU integer
    def _vbin_u(self, w, i, f):
        a, b = w.rd_v(i.ops[1]).astype(np.uint64), w.rd_v(i.ops[2]).astype(np.uint64)
        w.wr_v(i.ops[0], (f(a, b) & np.uint64(0xFFFFFFFF)).astype(np.uint32))

    def x_v_mov_b32(self, w, i):
        w.wr_v(i.ops[0], w.rd_v(i.ops[1]))

    def x_v_accvgpr_read_b32(self, w, i):
        w.wr_v(i.ops[0], w.rd_v(i.ops[1]))

    def x_v_accvgpr_write_b32(self, w, i):
        w.wr_v(i.ops[0], w.rd_v(i.ops[1]))

    def x_v_add_u32(self, w, i):
        self._vbin_u(w, i, lambda a, b: a + b)

    def x_v_sub_u32(self, w, i):
        self._vbin_u(w, i, lambda a, b: a - b)

    def x_v_subrev_u32(self, w, i):
        self._vbin_u(w, i, lambda a, b: b - a)

    def x_v_mul_lo_u32(self, w, i):
        self._vbin_u(w, i, lambda a, b: a * b)

    def x_v_mul_u32_u24(self, w, i):
        self._vbin_u(w, i, lambda a, b: (a & np.uint64(0xFFFFFF)) * (b & np.uint64(0xFFFFFF)))

    def x_v_lshlrev_b32(self, w, i):
        self._vbin_u(w, i, lambda a, b: b << (a & np.uint64(31)))

    def x_v_lshrrev_b32(self, w, i):
        self._vbin_u(w, i, lambda a, b: b >> (a & np.uint64(31)))

    def x_v_and_b32(self, w, i):
        self._vbin_u(w, i, lambda a, b: a & b)

    def x_v_or_b32(self, w, i):
        self._vbin_u(w, i, lambda a, b: a | b)

    def x_v_xor_b32(self, w, i):
        self._vbin_u(w, i, lambda a, b: a ^ b)

    def _vtri_u(self, w, i, f):
        a, b, c = (w.rd_v(i.ops[k]).astype(np.uint64) for k in (1, 2, 3))
        w.wr_v(i.ops[0], (f(a, b, c) & np.uint64(0xFFFFFFFF)).astype(np.uint32))

    def x_v_lshl_add_u32(self, w, i):
        self._vtri_u(w, i, lambda a, b, c: (a << (b & np.uint64(31))) + c)

    def x_v_lshl_or_b32(self, w, i):
        self._vtri_u(w, i, lambda a, b, c: (a << (b & np.uint64(31))) | c)

    def x_v_and_or_b32(self, w, i):
        self._vtri_u(w, i, lambda a, b, c: (a & b) | c)

    def x_v_add3_u32(self, w, i):
        self._vtri_u(w, i, lambda a, b, c: a + b + c)

    def x_v_mad_u32_u24(self, w, i):
        self._vtri_u(w, i, lambda a, b, c: (a & np.uint64(0xFFFFFF)) * (b & np.uint64(0xFFFFFF)) + c)

    def x_v_bfe_u32(self, w, i):
        self._vtri_u(w, i, lambda a, b, c: (a >> (b & np.uint64(31))) & ((np.uint64(1) << (c & np.uint64(31))) - np.uint64(1)))

    def x_v_readfirstlane_b32(self, w, i):
        m = w.exec_mask()
        lane = int(np.argmax(m)) if m.any() else 0
        w.wr_s(i.ops[0], int(w.rd_v(i.ops[1])[lane]))

    def x_v_cndmask_b32(self, w, i):
        sel = w.rd_s(i.ops[3])
        bits = np.array([(sel >> l) & 1 for l in range(64)], bool)

        def src(o):   # (VOP3 form: a neg modifier flips the sign bit)
            x = w.rd_v(o)
            return x ^ np.uint32(0x80000000) if isinstance(o, Reg) and o.neg else x
        w.wr_v(i.ops[0], np.where(bits, src(i.ops[2]), src(i.ops[1])))

    def _vcmp(self, w, i, f, kind):
        if kind == "f":
            a, b = w.rd_f(i.ops[1]), w.rd_f(i.ops[2])
        elif kind == "i":
            a, b = w.rd_v(i.ops[1]).view(np.int32), w.rd_v(i.ops[2]).view(np.int32)
        else:
            a, b = w.rd_v(i.ops[1]), w.rd_v(i.ops[2])
        r = f(a, b) & w.exec_mask()
        val = 0
        for l in range(64):
            if r[l]:
                val |= 1 << l
        w.wr_s(i.ops[0], val)

    def x_v_cmp_gt_f32(self, w, i):
        self._vcmp(w, i, lambda a, b: a > b, "f")

    def x_v_cmp_lt_f32(self, w, i):
        self._vcmp(w, i, lambda a, b: a < b, "f")

    def x_v_cmp_gt_i32(self, w, i):
        self._vcmp(w, i, lambda a, b: a > b, "i")

    def x_v_cmp_lt_i32(self, w, i):
        self._vcmp(w, i, lambda a, b: a < b, "i")

    def x_v_cmp_ge_i32(self, w, i):
        self._vcmp(w, i, lambda a, b: a >= b, "i")

    def x_v_cmp_gt_u32(self, w, i):
        self._vcmp(w, i, lambda a, b: a > b, "u")

    def x_v_cmp_lt_u32(self, w, i):
        self._vcmp(w, i, lambda a, b: a < b, "u")

    def x_v_cmp_eq_u32(self, w, i):
        self._vcmp(w, i, lambda a, b: a == b, "u")

    # ---- VALU float
    def x_v_cvt_f32_u32(self, w, i):
        w.wr_v(i.ops[0], w.rd_v(i.ops[1]).astype(np.float32))

    def x_v_cvt_u32_f32(self, w, i):
        x = w.rd_f(i.ops[1])
        x = np.where(np.isnan(x), 0, np.clip(np.trunc(x.astype(np.float64)), 0, 0xFFFFFFFF))
        w.wr_v(i.ops[0], x.astype(np.uint32))

    def x_v_cvt_i32_f32(self, w, i):
        x = w.rd_f(i.ops[1])
        x = np.where(np.isnan(x), 0, np.clip(np.trunc(x.astype(np.float64)), -2 ** 31, 2 ** 31 - 1))
        w.wr_v(i.ops[0], x.astype(np.int64).astype(np.uint32))

    def x_v_ceil_f32(self, w, i):
        w.wr_v(i.ops[0], np.ceil(w.rd_f(i.ops[1])).astype(np.float32))

    def x_v_ldexp_f32(self, w, i):
        e = w.rd_v(i.ops[2]).view(np.int32) if isinstance(i.ops[2], Reg) else np.int32(i.ops[2])
        w.wr_v(i.ops[0], np.ldexp(w.rd_f(i.ops[1]).astype(np.float64), e).astype(np.float32))

    def x_v_rcp_f32(self, w, i):
        w.wr_v(i.ops[0], (np.float32(1.0) / w.rd_f(i.ops[1])).astype(np.float32))

    def x_v_exp_f32(self, w, i):
        w.wr_v(i.ops[0], np.exp2(w.rd_f(i.ops[1]).astype(np.float64)).astype(np.float32))

    def x_v_log_f32(self, w, i):
        w.wr_v(i.ops[0], np.log2(w.rd_f(i.ops[1]).astype(np.float64)).astype(np.float32))

    def x_v_pk_mul_f32(self, w, i):
        sel = i.mods.get("op_sel_hi", (1, 1))
        a = [w.rd_v(i.ops[1], k).view(np.float32) for k in range(2)]
        b = [w.rd_v(i.ops[2], k).view(np.float32) for k in range(2)]
        lo = (a[0] * b[0]).astype(np.float32)
        hi = (a[sel[0]] * b[sel[1]]).astype(np.float32)
        w.wr_v(i.ops[0], lo, 0)
        w.wr_v(i.ops[0], hi, 1)

    def x_v_mul_f32(self, w, i):
        w.wr_v(i.ops[0], (w.rd_f(i.ops[1]) * w.rd_f(i.ops[2])).astype(np.float32))

    def x_v_add_f32(self, w, i):
        w.wr_v(i.ops[0], (w.rd_f(i.ops[1]) + w.rd_f(i.ops[2])).astype(np.float32))

    def x_v_sub_f32(self, w, i):
        w.wr_v(i.ops[0], (w.rd_f(i.ops[1]) - w.rd_f(i.ops[2])).astype(np.float32))

    def x_v_fma_f32(self, w, i):
        a, b, c = (w.rd_f(i.ops[k]).astype(np.float64) for k in (1, 2, 3))
        w.wr_v(i.ops[0], (a * b + c).astype(np.float32))

    def x_v_max_f32(self, w, i):
        w.wr_v(i.ops[0], np.fmax(w.rd_f(i.ops[1]), w.rd_f(i.ops[2])))

    def x_v_max3_f32(self, w, i):
        w.wr_v(i.ops[0], np.fmax(np.fmax(w.rd_f(i.ops[1]), w.rd_f(i.ops[2])), w.rd_f(i.ops[3])))

    def x_v_cvt_pk_bf16_f32(self, w, i):
        lo = f32_to_bf16_rne(w.rd_f(i.ops[1])).astype(np.uint32)
        hi = f32_to_bf16_rne(w.rd_f(i.ops[2])).astype(np.uint32)
        w.wr_v(i.ops[0], lo | (hi << 16))

    def x_v_cvt_pk_f16_f32(self, w, i):
        lo = w.rd_f(i.ops[1]).astype(np.float16).view(np.uint16).astype(np.uint32)
        hi = w.rd_f(i.ops[2]).astype(np.float16).view(np.uint16).astype(np.uint32)
        w.wr_v(i.ops[0], lo | (hi << 16))

    # ---- fp8 (OCP e4m3fn / e5m2): conversions through torch (round to nearest even; e4m3fn has no infinities: values beyond
    #      +-448 become NaN, as the hardware conversion gives without saturation -- the kernels keep P <= 2^6)
    @staticmethod
    def _f32_to_f8(x, bf8):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.float8_e5m2 if bf8 else torch.float8_e4m3fn)
        return t.view(torch.uint8).numpy().astype(np.uint32)

    @staticmethod
    def _f8_to_f32(u8, bf8):
        import torch
        return torch.from_numpy(np.ascontiguousarray(u8, np.uint8)).view(torch.float8_e5m2 if bf8 else torch.float8_e4m3fn).float().numpy()

    def _cvt_pk_f8(self, w, i, bf8):
        lo, hi = self._f32_to_f8(w.rd_f(i.ops[1]), bf8), self._f32_to_f8(w.rd_f(i.ops[2]), bf8)
        pair = lo | (hi << 8)
        old = w.rd_v(i.ops[0])
        if tuple(i.mods.get("op_sel", (0, 0, 0)))[2]:
            w.wr_v(i.ops[0], (old & np.uint32(0x0000FFFF)) | (pair << 16))
        else:
            w.wr_v(i.ops[0], (old & np.uint32(0xFFFF0000)) | pair)

    def x_v_cvt_pk_fp8_f32(self, w, i):
        self._cvt_pk_f8(w, i, False)

    def x_v_cvt_pk_bf8_f32(self, w, i):
        self._cvt_pk_f8(w, i, True)

    def _f8_operand(self, w, o, bf8):
        """8 registers of an f8f6f4 operand -> float64 [64 lanes][32 k slots] (slot j = byte j & 3 of register j >> 2)"""
        regs = np.stack([w.rd_v(o, k) for k in range(8)], axis=1)            # [64][8] uint32
        return self._f8_to_f32(regs.view(np.uint8).reshape(64, 32), bf8).astype(np.float64)

    def x_v_mfma_f32_32x32x64_f8f6f4(self, w, i):
        """A[m = l & 31][k = 32 (l >> 5) + j], B[k = 32 (l >> 5) + j][n = l & 31], j = byte index of the lane's 32-byte operand;
        D as the 32x32 bf16 forms: row m = (r & 3) + 8 (r >> 2) + 4 (l >> 5), column n = l & 31.  cbsz / blgp: 0 = e4m3, 1 = e5m2"""
        d, a, b, c = i.ops
        fa, fb = self._f8_operand(w, a, i.mods.get("cbsz", 0) == 1), self._f8_operand(w, b, i.mods.get("blgp", 0) == 1)
        assert i.mods.get("cbsz", 0) in (0, 1) and i.mods.get("blgp", 0) in (0, 1)
        Am = np.concatenate([fa[:32], fa[32:]], axis=1)        # [32 rows][64 k]
        Bm = np.concatenate([fb[:32], fb[32:]], axis=1)        # [32 cols][64 k]
        Cm = np.zeros((32, 32), np.float64)
        rows = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
        if isinstance(c, Reg):
            for r in range(16):
                x = w.rd_v(c, r).view(np.float32).astype(np.float64)
                for h in range(2):
                    Cm[rows(r, h), :] = x[32 * h:32 * h + 32]
        Dm = (Am @ Bm.T + Cm).astype(np.float32)
        for r in range(16):
            w.wr_v(d, np.concatenate([Dm[rows(r, 0), :], Dm[rows(r, 1), :]]), r, masked=False)

    def x_v_mfma_scale_f32_32x32x64_f8f6f4(self, w, i):
        """the block-scaled form: every product term is multiplied by 2^(sa - 127) 2^(sb - 127), E8M0 bytes of the two trailing
        VGPR operands.  Measured on the device (scripts/probes/mfma_scale.hip, profiles/r03/mfma_scale_probe.txt): lane (i, h)
        supplies the scale of row / column i for the K block h -- which is registers 4 h .. 4 h + 3 of BOTH lane halves (the
        hardware numbers k = 16 h' + 0..15 in registers 0-3 and 32 + 16 h' + 0..15 in registers 4-7 of half h'); the byte is picked
        by op_sel (bit 0) and op_sel_hi (bit 1) of the operand's position [a, b, -]"""
        d, a, b, c, sa, sb = i.ops
        fa, fb = self._f8_operand(w, a, i.mods.get("cbsz", 0) == 1), self._f8_operand(w, b, i.mods.get("blgp", 0) == 1)
        lo, hi = tuple(i.mods.get("op_sel", (0, 0, 0))), tuple(i.mods.get("op_sel_hi", (0, 0, 0)))
        for f, sreg, pos in ((fa, sa, 0), (fb, sb, 1)):
            byte = (w.rd_v(sreg) >> np.uint32(8 * (lo[pos] + 2 * hi[pos]))) & np.uint32(0xFF)
            assert (byte != 255).all(), "E8M0 NaN scale"
            sc = np.exp2(byte.astype(np.float64) - 127.0)                    # [64]: lane (i, h) -> row i, K block h
            for half in range(2):                                            # the data's lane half
                f[32 * half:32 * half + 32, :16] *= sc[:32, None]            # registers 0-3: K block 0 <- lane (i, 0)
                f[32 * half:32 * half + 32, 16:] *= sc[32:, None]            # registers 4-7: K block 1 <- lane (i, 1)
        Am = np.concatenate([fa[:32], fa[32:]], axis=1)
        Bm = np.concatenate([fb[:32], fb[32:]], axis=1)
        Cm = np.zeros((32, 32), np.float64)
        rows = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
        if isinstance(c, Reg):
            for r in range(16):
                x = w.rd_v(c, r).view(np.float32).astype(np.float64)
                for h in range(2):
                    Cm[rows(r, h), :] = x[32 * h:32 * h + 32]
        Dm = (Am @ Bm.T + Cm).astype(np.float32)
        for r in range(16):
            w.wr_v(d, np.concatenate([Dm[rows(r, 0), :], Dm[rows(r, 1), :]]), r, masked=False)

    def x_v_mfma_f32_16x16x128_f8f6f4(self, w, i):
        """A[m = l & 15][k = 32 (l >> 4) + j], B[k = 32 (l >> 4) + j][n = l & 15]; D[m = 4 (l >> 4) + r][n = l & 15]"""
        d, a, b, c = i.ops
        fa, fb = self._f8_operand(w, a, i.mods.get("cbsz", 0) == 1), self._f8_operand(w, b, i.mods.get("blgp", 0) == 1)
        Am = np.concatenate([fa[16 * g:16 * g + 16] for g in range(4)], axis=1)     # [16][128]
        Bm = np.concatenate([fb[16 * g:16 * g + 16] for g in range(4)], axis=1)
        Cm = np.zeros((16, 16), np.float64)
        if isinstance(c, Reg):
            for r in range(4):
                x = w.rd_v(c, r).view(np.float32).astype(np.float64)
                for g in range(4):
                    Cm[4 * g + r, :] = x[16 * g:16 * g + 16]
        Dm = (Am @ Bm.T + Cm).astype(np.float32)
        for r in range(4):
            w.wr_v(d, np.concatenate([Dm[4 * g + r, :] for g in range(4)]), r, masked=False)

    def x_v_permlane32_swap_b32(self, w, i):
        # lanes 32..63 of vdst swap with lanes 0..31 of src
        d, s = w.rd_v(i.ops[0]), w.rd_v(i.ops[1])
        nd, ns = d.copy(), s.copy()
        nd[32:] = s[:32]
        ns[:32] = d[32:]
        w.wr_v(i.ops[0], nd, masked=False)
        w.wr_v(i.ops[1], ns, masked=False)

    def x_v_permlane16_swap_b32(self, w, i):
        # rows of 16 lanes: the odd rows of vdst (lanes 16..31, 48..63) swap with the even rows of src (lanes 0..15, 32..47)
        d, s = w.rd_v(i.ops[0]), w.rd_v(i.ops[1])
        nd, ns = d.copy(), s.copy()
        for base in (0, 32):
            nd[base + 16:base + 32] = s[base:base + 16]
            ns[base:base + 16] = d[base + 16:base + 32]
        w.wr_v(i.ops[0], nd, masked=False)
        w.wr_v(i.ops[1], ns, masked=False)

    def x_v_mbcnt_lo_u32_b32(self, w, i):
        # popcount of mask[lane-1:0] (low 32 lanes' part) + src1
        mask = int(w.rd_v(i.ops[1])[0])
        base = w.rd_v(i.ops[2])
        out = np.array([bin(mask & ((1 << min(l, 32)) - 1)).count("1") for l in range(64)], np.uint32) + base.astype(np.uint32)
        w.wr_v(i.ops[0], out)

    def x_v_mbcnt_hi_u32_b32(self, w, i):
        mask = int(w.rd_v(i.ops[1])[0])
        base = w.rd_v(i.ops[2])
        out = np.array([bin(mask & ((1 << max(l - 32, 0)) - 1)).count("1") for l in range(64)], np.uint32) + base.astype(np.uint32)
        w.wr_v(i.ops[0], out)

    # ---- MFMA 32x32x16 (bf16 / f16)
    def _mfma(self, w, i, decode):
        d, a, b, c = i.ops

        def mat(o):  # -> [32][16] row r, k = 8h + j
            m = np.zeros((32, 16), np.float64)
            for reg in range(4):
                u = w.rd_v(o, reg)
                for half in range(2):
                    vals = decode(((u >> (16 * half)) & 0xFFFF).astype(np.uint16)).astype(np.float64)
                    j = 2 * reg + half
                    m[:, j] = vals[:32]
                    m[:, 8 + j] = vals[32:]
            return m
        Am = mat(a)
        Bm = mat(b).T  # [16][32]
        Cm = np.zeros((32, 32), np.float64)
        if isinstance(c, Reg):
            for reg in range(16):
                x = w.rd_v(c, reg).view(np.float32).astype(np.float64)
                row = (reg & 3) + 8 * (reg >> 2)
                Cm[row, :] = x[:32]
                Cm[row + 4, :] = x[32:]
        else:
            assert c == 0
        Dm = (Am @ Bm + Cm).astype(np.float32)
        for reg in range(16):
            row = (reg & 3) + 8 * (reg >> 2)
            w.wr_v(d, np.concatenate([Dm[row, :], Dm[row + 4, :]]), reg, masked=False)

    def _mfma16(self, w, i, decode):
        """v_mfma_f32_16x16x32: A[m = l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][n = l & 15], D[m = 4 (l >> 4) + r][n = l & 15]"""
        d, a, b, c = i.ops

        def mat(o):  # -> [16][32]: row l & 15, k = 8 (l >> 4) + j
            m = np.zeros((16, 32), np.float64)
            for reg in range(4):
                u = w.rd_v(o, reg)
                for half in range(2):
                    vals = decode(((u >> (16 * half)) & 0xFFFF).astype(np.uint16)).astype(np.float64)
                    j = 2 * reg + half
                    for g in range(4):
                        m[:, 8 * g + j] = vals[16 * g:16 * g + 16]
            return m
        Am = mat(a)
        Bm = mat(b).T  # [32][16]
        Cm = np.zeros((16, 16), np.float64)
        if isinstance(c, Reg):
            for reg in range(4):
                x = w.rd_v(c, reg).view(np.float32).astype(np.float64)
                for g in range(4):
                    Cm[4 * g + reg, :] = x[16 * g:16 * g + 16]
        Dm = (Am @ Bm + Cm).astype(np.float32)
        for reg in range(4):
            w.wr_v(d, np.concatenate([Dm[4 * g + reg, :] for g in range(4)]), reg, masked=False)

    def x_v_mfma_f32_16x16x32_bf16(self, w, i):
        self._mfma16(w, i, bf16_to_f32)

    def x_v_mfma_f32_16x16x32_f16(self, w, i):
        self._mfma16(w, i, lambda u: u.view(np.float16).astype(np.float32))

    def x_v_mfma_f32_32x32x16_bf16(self, w, i):
        self._mfma(w, i, bf16_to_f32)

    def x_v_mfma_f32_32x32x16_f16(self, w, i):
        self._mfma(w, i, lambda u: u.view(np.float16).astype(np.float32))

    # ---- LDS
    def _lds_addr(self, w, i, areg):
        return w.rd_v(areg).astype(np.int64) + int(i.mods.get("offset", 0))

    def _lds_read_lanes(self, addr, nbytes):
        out = np.zeros((64, nbytes), np.uint8)
        for l in range(64):
            a = int(addr[l])
            if a < 0 or a + nbytes > LDS_BYTES:
                raise RuntimeError(f"LDS read out of range: {a}")
            out[l] = self.lds[a:a + nbytes]
        return out

    def _queue_vreg_write(self, w, q, dst, data_u32):  # data_u32 [nreg][64]
        m = w.exec_mask()
        for k in range(dst.cnt):
            w.v[w.vrow(dst, k)] = np.where(m, POISON32, w.v[w.vrow(dst, k)])

        def apply():
            for k in range(dst.cnt):
                row = w.vrow(dst, k)
                w.v[row] = np.where(m, data_u32[k], w.v[row])
        q.append(apply)

    # ---- LDS bank model (MI355X_MICROARCH.md, section LDS): 64 banks of 4 bytes; a wave64 access is served in fixed lane groups,
    #      one LDS cycle per group when no two DIFFERENT addresses of the group meet on a bank (identical addresses broadcast)
    B128_GROUPS = ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
                   [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59], [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63])
    HALF_GROUPS = (list(range(32)), list(range(32, 64)))

    def _bank_conflicts(self, tag, addr, nbytes, groups):
        """extra LDS cycles of one wave-instruction; summed per instruction tag in self.lds_conflicts (tests/test_asm_a16.py)"""
        extra = 0
        for grp in groups:
            banks = {}
            for l in grp:
                a = int(addr[l])
                for k in range(nbytes // 4):
                    banks.setdefault(((a >> 2) + k) & 63, set()).add(a)
            extra += max(len(v) for v in banks.values()) - 1
        st = self.__dict__.setdefault("lds_conflicts", {})
        n, x = st.get(tag, (0, 0))
        st[tag] = (n + 1, x + extra)

    def x_ds_read_b128(self, w, i):
        addr = self._lds_addr(w, i, i.ops[1])
        assert (addr % 16 == 0).all(), "ds_read_b128 misaligned"
        self._bank_conflicts((i.tag or "ds_read_b128").split()[0], addr, 16, self.B128_GROUPS)
        data = self._lds_read_lanes(addr, 16).view(np.uint32)  # [64][4]
        self._queue_vreg_write(w, w.lgkm_q, i.ops[0], data.T.copy())

    def x_ds_read_b64(self, w, i):
        addr = self._lds_addr(w, i, i.ops[1])
        assert (addr % 8 == 0).all()
        data = self._lds_read_lanes(addr, 8).view(np.uint32)
        self._queue_vreg_write(w, w.lgkm_q, i.ops[0], data.T.copy())

    def x_ds_read_b32(self, w, i):
        addr = self._lds_addr(w, i, i.ops[1])
        data = self._lds_read_lanes(addr, 4).view(np.uint32)
        self._queue_vreg_write(w, w.lgkm_q, i.ops[0], data.T.copy())

    def x_ds_read_b64_tr_b16(self, w, i):
        assert w.exec_mask().all(), "ds_read_b64_tr_b16 needs EXEC all ones"
        addr = self._lds_addr(w, i, i.ops[1])
        assert (addr % 8 == 0).all(), "ds_read_b64_tr_b16 misaligned"
        self._bank_conflicts((i.tag or "ds_read_b64_tr_b16").split()[0], addr, 8, self.HALF_GROUPS)
        raw = self._lds_read_lanes(addr, 8).view(np.uint16)  # [64][4]: lane 4q+p of a group holds row q, cols 4p..4p+3
        out = np.zeros((64, 4), np.uint16)
        for g in range(4):
            for li in range(16):
                for q in range(4):
                    out[16 * g + li, q] = raw[16 * g + 4 * q + li // 4, li % 4]
        data = out.view(np.uint32)  # [64][2]
        self._queue_vreg_write(w, w.lgkm_q, i.ops[0], data.T.copy())

    def x_ds_read_b64_tr_b8(self, w, i):
        """measured on the device (profiles/r01/ds_read_b64_tr_b8_lane_map.txt): in a 16-lane group lane i < 8 receives byte i of
        the 8-byte segments addressed by lanes 0, 2, .., 14, lane i >= 8 byte i - 8 of those addressed by lanes 1, 3, .., 15"""
        assert w.exec_mask().all(), "ds_read_b64_tr_b8 needs EXEC all ones"
        addr = self._lds_addr(w, i, i.ops[1])
        assert (addr % 8 == 0).all(), "ds_read_b64_tr_b8 misaligned"
        self._bank_conflicts((i.tag or "ds_read_b64_tr_b8").split()[0], addr, 8, self.HALF_GROUPS)
        raw = self._lds_read_lanes(addr, 8)      # [64][8]
        out = np.zeros((64, 8), np.uint8)
        for g in range(4):
            for li in range(16):
                for k in range(8):
                    out[16 * g + li, k] = raw[16 * g + 2 * k + (li >> 3), li & 7]
        self._queue_vreg_write(w, w.lgkm_q, i.ops[0], out.view(np.uint32).T.copy())

    def _lds_write(self, w, i, nbytes):
        addr = self._lds_addr(w, i, i.ops[0])
        assert (addr % min(nbytes, 8) == 0).all()
        src = i.ops[1]
        m = w.exec_mask()
        nreg = nbytes // 4
        regs = np.stack([w.rd_v(src, k) for k in range(nreg)], axis=1)  # [64][nreg]
        raw = regs.view(np.uint8).reshape(64, nbytes)
        for l in range(64):
            if m[l]:
                a = int(addr[l])
                if a < 0 or a + nbytes > LDS_BYTES:
                    raise RuntimeError(f"LDS write out of range: {a}")
                self.lds[a:a + nbytes] = raw[l]
        w.lgkm_q.append(None)

    def x_ds_write_b64(self, w, i):
        self._lds_write(w, i, 8)

    def x_ds_write_b32(self, w, i):
        self._lds_write(w, i, 4)

    def x_ds_write_b128(self, w, i):
        self._lds_write(w, i, 16)

    # ---- buffer ops
    def _buf(self, w, rsrc: Reg):
        s0, s1, s2, s3 = (int(w.s[rsrc.idx + k]) for k in range(4))
        base = s0 | ((s1 & 0xFFFF) << 32)
        assert (s1 >> 16) == 0, "buffer stride / swizzle bits set"
        return base, s2

    def _buf_addrs(self, w, i, vaddr, rsrc, soff, nbytes):
        base, nrec = self._buf(w, rsrc)
        off = np.zeros(64, np.int64)
        if i.mods.get("offen"):
            off = w.rd_v(vaddr).astype(np.int64)
        off = off + int(i.mods.get("offset", 0))
        ok = (off + nbytes) <= nrec  # raw buffer: soffset is NOT range-checked
        so = w.rd_s(soff)
        return base + off + so, ok

    def x_buffer_load_dwordx4(self, w, i):
        m = w.exec_mask()
        if i.mods.get("lds"):
            vaddr, rsrc, soff = i.ops
            addrs, ok = self._buf_addrs(w, i, vaddr, rsrc, soff, 16)
            m0 = int(w.s[124])
            assert m0 % 16 == 0, f"LDS-DMA base {m0}"
            dst0 = m0 + int(i.mods.get("offset", 0))
            data = np.zeros((64, 16), np.uint8)
            for l in range(64):
                if m[l] and ok[l]:
                    data[l] = self.mem.read(int(addrs[l]), 16)
            if dst0 + 1024 > LDS_BYTES:
                raise RuntimeError("LDS-DMA out of range")
            lds = self.lds
            pois = np.frombuffer(np.full(4, POISON32, np.uint32).tobytes(), np.uint8)
            for l in range(64):
                if m[l]:
                    lds[dst0 + 16 * l: dst0 + 16 * l + 16] = pois

            def apply():
                for l in range(64):
                    if m[l]:
                        lds[dst0 + 16 * l: dst0 + 16 * l + 16] = data[l]
            w.vm_q.append(apply)
            return
        dst, vaddr, rsrc, soff = i.ops
        addrs, ok = self._buf_addrs(w, i, vaddr, rsrc, soff, 16)
        data = np.zeros((64, 4), np.uint32)
        for l in range(64):
            if m[l] and ok[l]:
                data[l] = self.mem.read(int(addrs[l]), 16).view(np.uint32)
        self._queue_vreg_write(w, w.vm_q, dst, data.T.copy())

    def _buf_store(self, w, i, nbytes):
        src, vaddr, rsrc, soff = i.ops
        addrs, ok = self._buf_addrs(w, i, vaddr, rsrc, soff, nbytes)
        m = w.exec_mask()
        nreg = max(1, nbytes // 4)
        regs = np.stack([w.rd_v(src, k) for k in range(nreg)], axis=1)
        raw = regs.view(np.uint8).reshape(64, 4 * nreg)
        for l in range(64):
            if m[l] and ok[l]:
                self.mem.write(int(addrs[l]), raw[l, :nbytes])
        w.vm_q.append(None)

    def x_buffer_store_dwordx4(self, w, i):
        self._buf_store(w, i, 16)

    def x_buffer_store_dword(self, w, i):
        self._buf_store(w, i, 4)

    def x_buffer_store_short(self, w, i):
        self._buf_store(w, i, 2)

    def x_buffer_store_byte(self, w, i):
        self._buf_store(w, i, 1)

    def x_global_store_dword(self, w, i):
        vaddr, src, sbase = i.ops
        base = w.rd_s(sbase)
        off = w.rd_v(vaddr).astype(np.int64) + int(i.mods.get("offset", 0))
        m = w.exec_mask()
        data = w.rd_v(src)
        for l in range(64):
            if m[l]:
                self.mem.write(base + int(off[l]), np.array([data[l]], np.uint32).view(np.uint8))
        w.vm_q.append(None)
