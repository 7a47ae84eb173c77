import torch, time, sys
sys.path.insert(0, '/root/repo')
import flash_attention_dlrs_amd as fa
dev = torch.device('cuda:0')
for shape, dt in (((8, 8, 256, 128), torch.float32), ((32, 32, 256, 128), torch.float32), ((2, 8, 1024, 64), torch.float16), ((8, 16, 4096, 64), torch.float16)):
    Q, K, V, dO = (torch.randn(*shape, device=dev).to(dt) for _ in range(4))
    O, L = fa.flash_attention_forward(Q, K, V, dev)
    for _ in range(2): fa.flash_attention_backward(Q, K, V, O, dO, L, dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fa.flash_attention_backward(Q, K, V, O, dO, L, dev)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    B, H, N, d = shape
    print(shape, dt, f"{ms:.3f} ms", f"{10.0*B*H*N*N*d/ms/1e9:.1f} TFLOP/s")
