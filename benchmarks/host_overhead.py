import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flash_attention_dlrs_amd import flash_attention_forward, _lib
import flash_attention_dlrs_amd.flash_attention_torch as ft
dev = torch.device('cuda:0')
Q, K, V = (torch.randn(2, 8, 64, 64, device=dev).to(torch.float16) for _ in range(3))
def host(fn, n=2000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
O = torch.empty_like(Q); L = torch.empty(2, 8, 64, 1, dtype=Q.dtype, device=dev)
dt = ft.convert_triton_dtype(Q.dtype)
print("wrapper       ", round(host(lambda: flash_attention_forward(Q, K, V, dev)), 2))
print("_lib.fa2_fwd  ", round(host(lambda: _lib.fa2_fwd(Q, K, V, O, L, dt)), 2))
print("2x torch.empty", round(host(lambda: (torch.empty(2, 8, 64, 64, dtype=Q.dtype, device=dev), torch.empty(2, 8, 64, 1, dtype=Q.dtype, device=dev))), 2))
print("current_stream", round(host(lambda: torch.cuda.current_stream(Q.device).cuda_stream), 2))
print("device guard  ", round(host(lambda: torch.cuda.device(Q.device).__enter__()), 2))
print("5x _i64       ", round(host(lambda: [_lib._i64(Q.stride()) for _ in range(5)]), 2))
print("apply (autograd)", round(host(lambda: ft.FlashAttention.apply(Q, K, V)), 2))
