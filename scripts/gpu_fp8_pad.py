"""fp8 at head size 64: padded on the host to 128 (matrix kernels) against the generic kernel on the tensors as they are"""
import sys, time, torch
sys.path.insert(0, ".")
import flash_attention_dlrs_amd as fa
dev = torch.device("cuda:0")
Q, K, V = ((torch.randn(4, 32, 2048, 64, device=dev) * 0.5).to(torch.float8_e4m3fn) for _ in range(3))
for var in ("auto", "generic"):
    for _ in range(2):
        fa.flash_attention_forward(Q, K, V, dev, variant=var)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        fa.flash_attention_forward(Q, K, V, dev, variant=var)
    torch.cuda.synchronize()
    print(var, "ms per call", round((time.perf_counter() - t) / 5 * 1e3, 3), flush=True)
