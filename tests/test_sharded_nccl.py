"""The RCCL path of flash_attention_dlrs_amd/sharded.py on a real GPU: a world-size-1 `nccl` group (= RCCL on ROCm) runs the
side stream, the per-batch all_gather_into_tensor, the uint8 views RCCL needs for fp8 and the wait_stream ordering; at world
size 1 the gathered result must equal the local forward bit for bit.  (More ranks: tests/test_sharded_gloo.py on CPU.)"""
import os

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

import flash_attention_dlrs_amd as fa  # noqa: E402
from flash_attention_dlrs_amd.sharded import flash_attention_forward_sharded  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def nccl_world1():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=DEV)
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype,shape,causal", [(torch.bfloat16, (3, 4, 512, 128), True), (torch.float16, (2, 2, 300, 64), False),
                                                 (torch.float8_e4m3fn, (2, 4, 512, 128), False), (torch.float32, (2, 2, 130, 40), False)])
def test_gather_world1_equals_local_forward(nccl_world1, dtype, shape, causal):
    torch.manual_seed(7)
    Q, K, V = ((torch.randn(*shape, device=DEV) * 0.5).to(dtype) for _ in range(3))
    O_loc, L_loc = fa.flash_attention_forward(Q, K, V, DEV, causal=causal)
    O, L = flash_attention_forward_sharded(Q, K, V, causal=causal, gather=True, gather_L=True)
    torch.cuda.synchronize()
    u8 = lambda t: t.contiguous().view(torch.uint8)
    assert O.shape == O_loc.shape and torch.equal(u8(O), u8(O_loc))
    assert torch.equal(u8(L), u8(L_loc))
    O2, L2 = flash_attention_forward_sharded(Q, K, V, causal=causal, gather=False)   # no collective at all
    assert torch.equal(u8(O2), u8(O_loc))
