import sys, torch, numpy as np
sys.path.insert(0, ".")
from oracle import fa2_oracle as oracle
import flash_attention_dlrs_amd as fa
dev = torch.device("cuda:0")
for dtype, name, thr, step in ((torch.float8_e4m3fn, "float8_e4m3fn", 8.5, 0.125), (torch.float8_e5m2, "float8_e5m2", 15.0, 0.25)):
    for shape, seed, spread in (((1, 2, 256, 128), 15, 0.5), ((2, 3, 768, 128), 16, 0.7), ((1, 24, 1024, 128), 17, 1.0), ((1, 4, 2048, 128), 18, 1.0)):
        g = torch.Generator().manual_seed(seed)
        Q, K, V = ((torch.randn(*shape, generator=g) * spread).to(dtype).to(dev) for _ in range(3))
        O, L = fa.flash_attention_forward(Q, K, V, dev, causal=True, variant="a8")
        O8, L8 = fa.flash_attention_forward(Q, K, V, dev, causal=True, variant="mfma8x")
        f = lambda t: t.float().cpu().numpy()
        line = f"{name} {shape}: vs mfma8x O== {(O.float() == O8.float()).float().mean():.4f} L== {(L.float() == L8.float()).float().mean():.4f}"
        bound = step * O8.float().abs() + 0.5 * step * V.float().abs().max()
        viol = ((O.float() - O8.float()).abs() > bound)
        line += f" viol {int(viol.sum())} nan {int(torch.isnan(O.float()).sum())}"
        if shape[0] * shape[1] <= 6:
            O_ref, L_ref = oracle.forward_deferred(f(Q), f(K), f(V), name, causal=True, G=32, B_c=64, thr=thr, sum_rounded=True, ceil_m=True)
            Of, Lf = O.float().cpu(), L.float().cpu().flatten()
            Or, Lr = torch.from_numpy(O_ref), torch.from_numpy(L_ref).flatten()
            b2 = step * Or.abs() + 0.5 * step * V.float().abs().max().cpu()
            v2 = (Of - Or).abs() > b2
            line += f" | vs oracle O== {(Of == Or).float().mean():.4f} L== {(Lf == Lr).float().mean():.4f} viol {int(v2.sum())}"
            if v2.any():
                idx = v2.nonzero()
                line += f" rows {sorted(set(idx[:, 2].tolist()))[:20]}"
        print(line, flush=True)
