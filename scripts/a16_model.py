#!/usr/bin/env python3
"""Issue model of the a16 steady loop: walks the generated loop body, sums the issue cost of the fillers behind every MFMA and
estimates cycles per 64-key step as sum over MFMAs of max(T_mfma, hold + fillers) -- a tool for balancing the softmax plan
(the matrix pipe never catches up a late MFMA).  python scripts/a16_model.py [abl ...]"""
import sys
sys.path.insert(0, "/root/repo")
from flash_attention_dlrs_amd.csrc.asm.fa2_a16_gen import Gen
from flash_attention_dlrs_amd.csrc.asm.isa import TRANS_OPS

COST = dict(exp=8.0, valu=4.3, lds=2.5, dma=14.0, salu=0.6, wait=1.0, branch=2.0)


def cost(i):
    if i.op in (".label", ".comment"):
        return 0.0
    if i.op in TRANS_OPS:
        return COST["exp"]
    if i.op.startswith("ds_"):
        return COST["lds"]
    if i.op.startswith("buffer_"):
        return COST["dma"]
    if i.op == "s_nop":
        return 4.0 * (i.ops[0] + 1)
    if i.op in ("s_waitcnt", "s_barrier"):
        return COST["wait"]
    if i.op.startswith("s_cbranch") or i.op == "s_branch":
        return COST["branch"]
    if i.op.startswith("s_"):
        return COST["salu"]
    return COST["valu"]


def loop_body(g):
    p = g.prog
    a = next(k for k, i in enumerate(p) if i.op == ".label" and i.ops[0].name.endswith("_loop"))
    b = next(k for k in range(a, len(p)) if p[k].op == "s_cbranch_scc1" and p[k].ops[0].name.endswith("_loop"))
    return p[a:b + 1]


def model(body, hold=8.0):
    total, gaps, cur, tm = 0.0, [], None, None
    for i in body:
        if i.is_mfma:
            if cur is not None:
                gaps.append((tm, cur))
            cur, tm = 0.0, (16.0 if "16x16x32" in i.op else 32.0)
        elif cur is not None:
            cur += cost(i)
    gaps.append((tm, cur))
    est = sum(max(t, hold + f) for t, f in gaps)
    return est / 4.0, gaps       # the body is four steps


if __name__ == "__main__":
    import collections
    kw = {}
    for a in sys.argv[1:]:
        k, v = a.split("=")
        kw[k] = eval(v)
    g = Gen("bf16", False, **kw)
    g.build()
    est, gaps = model(loop_body(g))
    fill = [f for _, f in gaps]
    print(f"gaps per step {len(gaps) / 4:.0f}  filler cycles per step {sum(fill) / 4:.0f}  estimate {est:.0f} cycles per step "
          f"(MFMA only {sum(t for t, _ in gaps) / 4:.0f})")
    h = collections.Counter(int(f // 2) * 2 for f in fill)
    print("filler cycles per half gap:", " ".join(f"{k}:{h[k]}" for k in sorted(h)))
