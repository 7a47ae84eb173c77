"""(batch, head)-sharded forward across the GPUs of one node -- an extension; the reference has no
multi-GPU code at all.

Every (b, h) pair is an independent attention problem (reference kernels.py:38-40: programs never
communicate), so the path shards with NO data-path collective: rank r owns heads
[r*H/G, (r+1)*H/G) of every batch element and runs the single-GPU kernel on its shard.  The one
optional exchange step is the gather of the output shards (RCCL all-gather over xGMI, one process
per GPU, torch.distributed backend "nccl" = RCCL on ROCm).  It is issued per batch element on a side
stream so that the copy of batch b overlaps the kernel of batch b+1, and it lands directly in
(B, H, N, d) order: for a fixed b the G shards (H/G, N, d) are contiguous and consecutive in the
gathered tensor, so no permute/copy is needed afterwards.

Global head index of local head j on rank r is r*H/G + j -- the same indexing a single GPU uses, so
concat(shards) == single-GPU output bit for bit (tests/test_sharded_gloo.py, tests/test_fwd_parity.py).
"""
import torch
import torch.distributed as dist


def head_shard_range(H, world_size, rank):
    """[h0, h1) owned by `rank`; H must divide evenly (BASELINE.json configs c4/c5 do)."""
    if H % world_size != 0:
        raise ValueError(f"H={H} is not divisible by world_size={world_size}")
    hs = H // world_size
    return rank * hs, (rank + 1) * hs


def shard_heads(t, world_size, rank):
    """View of the (B, H, N, d) tensor holding this rank's heads."""
    h0, h1 = head_shard_range(t.shape[1], world_size, rank)
    return t[:, h0:h1]


def _default_local_forward(Q, K, V, causal, scale):
    from .flash_attention_wrappers import flash_attention_forward
    return flash_attention_forward(Q, K, V, Q.device, causal=causal, scale=scale)


def flash_attention_forward_sharded(Q, K, V, *, group=None, causal=False, scale=1.0, gather=True,
                                    gather_L=False, local_forward=None):
    """Forward on this rank's head shard, optionally all-gathered.

    Q, K, V : this rank's shard, (B, H/G, N, d), resident on this rank's device.
    Returns (O, L):
      gather=False : the local shards (B, H/G, N, d), (B, H/G, N, 1) -- no collective at all.
      gather=True  : O as the full (B, H, N, d) tensor on every rank; L likewise if gather_L else local.
    `local_forward(Q, K, V, causal, scale) -> (O, L)` defaults to the HIP path; CPU tests inject one.
    """
    fwd = local_forward or _default_local_forward
    if not gather:
        return fwd(Q, K, V, causal, scale)

    G = dist.get_world_size(group)
    B, Hs, N, d = Q.shape
    O_full = torch.empty(B, G * Hs, N, d, dtype=Q.dtype, device=Q.device)
    L_full = torch.empty(B, G * Hs, N, 1, dtype=Q.dtype, device=Q.device) if gather_L else None
    on_gpu = Q.device.type == "cuda"
    u8 = Q.dtype in (torch.float8_e5m2, torch.float8_e4m3fn)  # RCCL has no fp8 datatype: move bytes

    def _ag(dst, src):
        if u8:
            dst, src = dst.view(torch.uint8), src.view(torch.uint8)
        return dist.all_gather_into_tensor(dst, src.contiguous(), group=group, async_op=True)

    works, keep = [], []
    if on_gpu:
        comm = torch.cuda.Stream(device=Q.device)
        main = torch.cuda.current_stream(Q.device)
    L_parts = []
    for b in range(B):
        O_b, L_b = fwd(Q[b:b + 1], K[b:b + 1], V[b:b + 1], causal, scale)
        L_parts.append(L_b)
        keep.append(O_b)
        if on_gpu:
            comm.wait_stream(main)  # kernel of batch b done before its shard leaves
            with torch.cuda.stream(comm):
                works.append(_ag(O_full[b], O_b[0]))
                if gather_L:
                    works.append(_ag(L_full[b], L_b[0]))
        else:
            works.append(_ag(O_full[b], O_b[0]))
            if gather_L:
                works.append(_ag(L_full[b], L_b[0]))
    for w in works:
        w.wait()
    if on_gpu:
        main.wait_stream(comm)
    L_local = torch.cat(L_parts, dim=0)
    return O_full, (L_full if gather_L else L_local)


def _default_local_backward(Q, K, V, O, dO, L, causal, scale):
    from .flash_attention_wrappers import flash_attention_backward
    return flash_attention_backward(Q, K, V, O, dO, L, Q.device, causal=causal, scale=scale)


def flash_attention_backward_sharded(Q, K, V, O, dO, L, *, causal=False, scale=1.0, local_backward=None):
    """Backward on this rank's head shard: (dQ, dK, dV) of the local heads, shapes (B, H/G, N, d).

    No collective: the gradient of head h depends only on head h's Q, K, V, O, dO, L (reference bwd_kernel,
    kernels.py:222-225: programs are indexed (j, b, h) and never communicate across (b, h)), so each rank's
    gradients ARE its shard of the full gradients -- slice of the single-GPU result bit for bit.  If dO arrives as the
    full (B, H, N, d) tensor (e.g. from a loss computed on gathered outputs), pass `shard_heads(dO, G, rank)`."""
    bwd = local_backward or _default_local_backward
    return bwd(Q, K, V, O, dO, L, causal, scale)
