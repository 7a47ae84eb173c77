"""Static wait-state checker for a generated instruction list (hipcc pads nothing in hand-written assembly).

Rules (ROCm 7.2 hazard recognizer as observed for gfx950, cdna_hip_programming.md section 5.7; distances are wait states:
every instruction in between counts one, `s_nop N` counts N + 1):
  R1  MFMA (8-pass) result -> read or overwritten by anything but the next MFMA of its own accumulate chain: >= 12
  R2  VALU result -> MFMA operand: >= 2
  R3  transcendental result -> other VALU: >= 1
  R4  VALU result -> v_permlane32_swap operand: >= 2
  R5  VALU result -> v_readfirstlane source: >= 1
  R6  VALU-written SGPR (v_readfirstlane, v_cmp) -> memory instruction reading it: >= 5 (SALU readers are interlocked)
  R7  SALU write of M0 -> LDS-DMA: >= 1
  R8  MFMA C operand -> overwritten by VALU: >= 12
  R10 vector-memory store of more than 64 bits of data -> VALU write of those data registers: >= 2 (the store reads its data
      after it has issued; nothing in the generated streams is that close today -- the rule guards schedule edits)
  R9  (not a wait state: a scheduling invariant) the SCC an instruction consumes was produced by the instruction meant to
      produce it -- s_addc_u32 by the s_add_u32 of the low half of its own register pair, s_cselect / s_cbranch_scc* by a
      compare -- with no label in between.  Interleaving scalar sequences into MFMA gaps one instruction at a time broke a
      64-bit carry chain this way (a descriptor base lost its carry whenever the low word overflowed).
The scan is linear over the listing (labels do not reset it): blocks reached by a taken branch must open with their own pad.
"""
from __future__ import annotations

from .isa import Inst, TRANS_OPS


def mfma_wait(ins: Inst) -> int:
    """wait states between an MFMA and a reader / overwriter of its result (rules R1, R8): 12 for the 8-pass forms (and, on the
    safe side, the 4-pass 16x16x32); the 16-pass v_mfma_f32_32x32x64_f8f6f4 holds its destination twice as long"""
    return 20 if "32x32x64" in ins.op else 12


def wait_states(ins: Inst) -> int:
    if ins.op in (".label", ".comment"):
        return 0
    if ins.op == "s_nop":
        return ins.ops[0] + 1
    return 1


REQUIRED = {"R1": 12, "R2": 2, "R3": 1, "R4": 2, "R5": 1, "R6": 5, "R7": 1, "R8": 12, "R10": 2}


def fix(prog, max_rounds=64):
    """What a compiler's hazard recognizer does: insert `s_nop` in front of every instruction that violates a wait-state rule.
    Returns (new program, number of wait states inserted).  In the densely scheduled loop nothing is inserted; the sparse
    code around it (pipeline fill, rare paths) gets its pads here instead of by hand."""
    prog = list(prog)
    added = 0
    for _ in range(max_rounds):
        errs = check(prog, verbose=False)
        if not errs:
            # taken branches: pad behind the target label by what the path into it lacks
            bt = check_branch_targets(prog)
            if not bt:
                return prog, added
            need_at = {}
            for _, tgt, _, short in bt:
                need_at[tgt] = max(need_at.get(tgt, 0), short)
            for tgt in sorted(need_at, reverse=True):
                n = need_at[tgt]
                added += n
                pads = []
                while n > 0:
                    pads.append(Inst("s_nop", (min(n, 16) - 1,), {}, "wait states behind a taken branch (check.fix)"))
                    n -= min(n, 16)
                prog[tgt + 1:tgt + 1] = pads
            continue
        if any(rule.startswith("R9") for _, rule, _, _ in errs):
            bad = [idx for idx, rule, _, _ in errs if rule.startswith("R9")]
            raise RuntimeError(f"SCC consumed from the wrong producer at instruction(s) {bad[:8]}: "
                               f"{prog[bad[0]].text().strip()} (keep scalar carry / compare sequences in one filler unit)")
        need = {}
        for idx, rule, _, dist in errs:
            key = rule.split()[0]
            req = int(key.split("@")[1]) if "@" in key else REQUIRED[key]
            need[idx] = max(need.get(idx, 0), req - dist)
        for idx in sorted(need, reverse=True):
            n = need[idx]
            added += n
            pads = []
            while n > 0:
                pads.append(Inst("s_nop", (min(n, 16) - 1,), {}, "wait states (check.fix)"))
                n -= min(n, 16)
            prog[idx:idx] = pads
    raise RuntimeError("hazard fixer did not converge")


def check_scc(prog):
    """rule R9; returns [(index, rule text, ("scc", 0), 0)]"""
    errs = []
    last = None
    for idx, ins in enumerate(prog):
        if ins.op == ".label":
            last = None
            continue
        if ins.op == ".comment":
            continue
        d, u = ins.defs_uses()
        if ("scc", 0) in u:
            if last is None:
                ok = False
            elif ins.op == "s_addc_u32":
                ok = last.op == "s_add_u32" and last.ops[0].kind == ins.ops[0].kind and last.ops[0].idx + 1 == ins.ops[0].idx
            else:
                ok = last.op.startswith("s_cmp") or last.op.startswith("s_bitcmp")
            if not ok:
                errs.append((idx, "R9 scc producer", ("scc", 0), 0))
        if ("scc", 0) in d:
            last = ins
    return errs


def check_branch_targets(prog, need=None):
    """out-of-line blocks are scanned by check() where they stand in the listing, not where they are entered from: for every
    branch, the registers the target block touches before its first branch out must be `need` wait states clear of the last
    MFMA that wrote them on the path INTO the branch (rule R1 across a taken branch).
    Returns [(branch index, target label index, reg, missing wait states)]"""
    labels = {ins.ops[0].name: i for i, ins in enumerate(prog) if ins.op == ".label"}
    errs = []
    if need is None:      # (the longest MFMA of the program decides: 20 wait states with a 16-pass MFMA in it, else 12)
        need = max([mfma_wait(ins) for ins in prog if ins.is_mfma] or [12])
    for bi, br in enumerate(prog):
        if not (br.op.startswith("s_cbranch") or br.op == "s_branch"):
            continue
        tgt = labels.get(getattr(br.ops[0], "name", None))
        if tgt is None:
            continue
        # registers touched after the label, each with the wait states the block itself puts in front of the touch
        first_touch = {}
        ahead = 0
        for ins in prog[tgt + 1:tgt + 200]:
            if ins.op in (".label", ".comment"):
                continue
            if ins.op.startswith("s_branch") or ins.op == "s_endpgm" or ahead >= need:
                break
            if ins.op != "s_nop" and not ins.is_mfma:   # (MFMA after MFMA: the linear scan's business)
                d, u = ins.defs_uses()
                for r in list(d) + list(u):
                    if r[0] in ("v", "a"):
                        first_touch.setdefault(r, ahead)
            ahead += wait_states(ins)
        if not first_touch:
            continue
        dist = 1   # the branch itself
        worst = None
        for ins in reversed(prog[max(0, bi - 400):bi]):
            if ins.op == ".label" or dist >= need:   # (a join point: paths into it are checked where they branch)
                break
            if ins.is_mfma:
                for r in ins.ops[0].regs():
                    if r in first_touch and dist + first_touch[r] < need:
                        short = need - dist - first_touch[r]
                        if worst is None or short > worst[1]:
                            worst = (r, short)
            dist += wait_states(ins)
        if worst:
            errs.append((bi, tgt, worst[0], worst[1]))
    return errs


def check(prog, verbose=True):
    last_mfma_def = {}    # reg -> (pos, inst index, dst-range tuple)
    last_mfma_csrc = {}   # reg -> pos (read as C by an MFMA)
    last_valu_def = {}    # reg -> pos
    last_trans_def = {}
    last_valu_sgpr = {}
    last_m0 = -100
    last_wide_store = {}  # data reg -> pos of a store of more than 64 bits that reads it
    pos = 0
    errs = []
    for idx, ins in enumerate(prog):
        ws = wait_states(ins)
        if ws == 0:
            continue
        d, u = ins.defs_uses()
        here = pos

        def dist(p):
            return here - p - 1  # wait states strictly between

        if ins.is_mfma:
            dst = tuple(ins.ops[0].regs())
            csrc = tuple(ins.ops[3].regs()) if hasattr(ins.ops[3], "regs") else ()
            for r in u:
                if r in last_mfma_def:
                    p, pi, rng = last_mfma_def[r]
                    same_chain = (r in csrc) and rng == dst and csrc == dst
                    if not same_chain and dist(p) < mfma_wait(prog[pi]):
                        errs.append((idx, f"R1@{mfma_wait(prog[pi])} mfma->mfma operand", r, dist(p)))
                if r in last_valu_def and dist(last_valu_def[r]) < 2:
                    errs.append((idx, "R2 valu->mfma", r, dist(last_valu_def[r])))
            for r in d:
                if r in last_mfma_def:
                    p, pi, rng = last_mfma_def[r]
                    if not (rng == dst and csrc == dst) and dist(p) < mfma_wait(prog[pi]):
                        errs.append((idx, f"R1@{mfma_wait(prog[pi])} mfma->mfma overwrite", r, dist(p)))
            for r in csrc:
                last_mfma_csrc[r] = (here, mfma_wait(ins))
            for r in dst:
                last_mfma_def[r] = (here, idx, dst)
                last_valu_def.pop(r, None)
                last_trans_def.pop(r, None)
        else:
            for r in list(u) + list(d):
                if r in last_mfma_def and dist(last_mfma_def[r][0]) < mfma_wait(prog[last_mfma_def[r][1]]):
                    errs.append((idx, f"R1@{mfma_wait(prog[last_mfma_def[r][1]])} mfma result touched", r, dist(last_mfma_def[r][0])))
            if ins.is_valu:
                for r in d:
                    if r in last_mfma_csrc and dist(last_mfma_csrc[r][0]) < last_mfma_csrc[r][1]:
                        errs.append((idx, f"R8@{last_mfma_csrc[r][1]} mfma C overwritten", r, dist(last_mfma_csrc[r][0])))
                    if r in last_wide_store and dist(last_wide_store[r]) < 2:
                        errs.append((idx, "R10 store data overwritten", r, dist(last_wide_store[r])))
                if ins.op not in TRANS_OPS:
                    for r in u:
                        if r in last_trans_def and dist(last_trans_def[r]) < 1:
                            errs.append((idx, "R3 trans->valu", r, dist(last_trans_def[r])))
                if ins.op in ("v_permlane32_swap_b32", "v_permlane16_swap_b32"):
                    for r in u:
                        if r in last_valu_def and dist(last_valu_def[r]) < 2:
                            errs.append((idx, "R4 valu->permlane", r, dist(last_valu_def[r])))
                if ins.op == "v_readfirstlane_b32":
                    for r in u:
                        if r in last_valu_def and dist(last_valu_def[r]) < 1:
                            errs.append((idx, "R5 valu->readfirstlane", r, dist(last_valu_def[r])))
            elif ins.op.startswith("buffer_") or ins.op.startswith("global_") or ins.op.startswith("ds_") or ins.op.startswith("s_load"):
                # memory instructions reading an SGPR a VALU wrote (the ISA's manual wait states list VMEM; a scalar ALU
                # instruction or branch reading it is interlocked by the hardware -- hipcc emits v_cmp / s_cbranch_vccnz and
                # v_readfirstlane / s_add back to back)
                for r in u:
                    if r[0] == "s" and r in last_valu_sgpr and dist(last_valu_sgpr[r]) < 5:
                        errs.append((idx, "R6 valu sgpr->reader", r, dist(last_valu_sgpr[r])))
            if ins.op in ("buffer_store_dwordx4", "buffer_store_dwordx3", "global_store_dwordx4", "global_store_dwordx3"):
                data = ins.ops[0] if ins.op.startswith("buffer_") else ins.ops[1]
                for r in data.regs():
                    last_wide_store[r] = here
            if ins.op.startswith("buffer_load") and ins.mods.get("lds") and dist(last_m0) < 1:
                errs.append((idx, "R7 m0->lds dma", ("s", 124), dist(last_m0)))
            for r in d:
                last_mfma_def.pop(r, None)
                if ins.is_valu:
                    if r[0] == "s":
                        last_valu_sgpr[r] = here
                    else:
                        last_valu_def[r] = here
                        if ins.op in TRANS_OPS:
                            last_trans_def[r] = here
                        else:
                            last_trans_def.pop(r, None)
                else:
                    last_valu_def.pop(r, None)
                    last_trans_def.pop(r, None)
                    last_valu_sgpr.pop(r, None)
                    if r == ("s", 124):
                        last_m0 = here
        pos += ws
    errs += check_scc(prog)
    if verbose:
        for idx, rule, r, dd in errs[:40]:
            print(f"  hazard {rule}: inst {idx} `{prog[idx].text().strip()}` reg {r} distance {dd}")
    return errs
