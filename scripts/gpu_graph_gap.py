"""c3: 20 launches eager vs the same 20 captured in one HIP graph: the event bracket per launch (launch-to-launch gaps included)"""
import sys, torch
sys.path.insert(0, ".")
from flash_attention_dlrs_amd import flash_attention_forward
dev = torch.device("cuda:0")
torch.manual_seed(42)
Q, K, V = (torch.randn(4, 32, 4096, 128, device=dev).to(torch.bfloat16) for _ in range(3))
step = lambda: flash_attention_forward(Q, K, V, dev, causal=True)
for _ in range(300):
    step()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    step()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            step()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
res = {"eager": [], "graph": []}
for rnd in range(9):
    for mode in ("eager", "graph"):
        for _ in range(3):
            (g.replay() if mode == "graph" else [step() for _ in range(20)])
        torch.cuda.synchronize()
        a.record()
        if mode == "graph":
            g.replay()
        else:
            for _ in range(20):
                step()
        b.record()
        torch.cuda.synchronize()
        res[mode].append(a.elapsed_time(b) / 20)
for k, v in res.items():
    v.sort()
    print(k, "ms per launch: median %.5f min %.5f" % (v[len(v) // 2], v[0]))
