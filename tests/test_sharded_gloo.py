"""world_size-2 gloo test of the (batch, head)-sharded path (flash_attention_dlrs_amd/sharded.py):
head ranges, gather landing order, bit-identical global head indexing.  Runs on CPU; the local
forward is the oracle (tests may use it as the checker's stand-in: no GPU here)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_local_forward(Q, K, V, causal, scale):
    from oracle import fa2_oracle
    O, L = fa2_oracle.forward(Q.numpy(), K.numpy(), V.numpy(), "float32", causal=causal, scale=scale)
    return torch.from_numpy(O), torch.from_numpy(L)


def _oracle_local_backward(Q, K, V, O, dO, L, causal, scale):
    from oracle import fa2_oracle
    g = fa2_oracle.backward(Q.numpy(), K.numpy(), V.numpy(), O.numpy(), dO.numpy(), L.numpy(), "float32",
                            causal=causal, scale=scale)
    return tuple(torch.from_numpy(x) for x in g[:3])


def _worker(rank, world, port, causal, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from flash_attention_dlrs_amd.sharded import flash_attention_forward_sharded, shard_heads
        torch.manual_seed(123)  # every rank draws the same full problem, then keeps its head shard
        B, H, N, d = 3, 4, 32, 16
        Q, K, V = (torch.randn(B, H, N, d) for _ in range(3))
        q, k, v = (shard_heads(t, world, rank).contiguous() for t in (Q, K, V))
        O_full, L_full = flash_attention_forward_sharded(q, k, v, causal=causal, gather=True, gather_L=True,
                                                         local_forward=_oracle_local_forward)
        O_loc, L_loc = flash_attention_forward_sharded(q, k, v, causal=causal, gather=False,
                                                       local_forward=_oracle_local_forward)
        O_one, L_one = _oracle_local_forward(Q, K, V, causal, 1.0)  # "single GPU" result
        assert O_full.shape == (B, H, N, d) and L_full.shape == (B, H, N, 1)
        assert torch.equal(O_full, O_one) and torch.equal(L_full, L_one)      # bit-identical head indexing
        assert torch.equal(O_loc, shard_heads(O_one, world, rank))
        assert torch.equal(L_loc, shard_heads(L_one, world, rank))
        # backward: no collective, the local gradients are the rank's slice of the full gradients, bit for bit
        from flash_attention_dlrs_amd.sharded import flash_attention_backward_sharded
        dO = torch.randn(B, H, N, d)
        g_loc = flash_attention_backward_sharded(q, k, v, O_loc, shard_heads(dO, world, rank).contiguous(), L_loc,
                                                 causal=causal, local_backward=_oracle_local_backward)
        g_one = _oracle_local_backward(Q, K, V, O_one, dO, L_one, causal, 1.0)
        for a, b_ in zip(g_loc, g_one):
            assert torch.equal(a, shard_heads(b_, world, rank))
        np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.array([1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("causal", [False, True])
def test_head_sharded_gather_world2(tmp_path, causal):
    from oracle import fa2_oracle
    fa2_oracle.build()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, causal, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}.npy") for r in range(2))


def test_head_shard_ranges():
    from flash_attention_dlrs_amd.sharded import head_shard_range
    assert [head_shard_range(32, 4, r) for r in range(4)] == [(0, 8), (8, 16), (16, 24), (24, 32)]
    assert head_shard_range(64, 8, 7) == (56, 64)
    with pytest.raises(ValueError):
        head_shard_range(6, 4, 0)
