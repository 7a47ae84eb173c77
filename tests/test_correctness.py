"""Counterpart of the reference's src/test_correctness.py (forward half): loop seeds, draw
Q, K, V ~ N(0,1) fp32 of shape (32, 32, 256, 128) ON THE GPU, compare flash_attention_forward with
torch SDPA(scale=1) on the same tensors using the reference's tolerance
`allclose(O_torch, O_flash, atol=1e-4, rtol=1e-5)` (src/test_correctness.py:9-14, :28-40).

Differences from the reference script: it is also a pytest test (fewer seeds), it exits non-zero on
failure when run as a script, and the backward half is out of scope.

    python tests/test_correctness.py            # 200 seeds, prints the reference's summary line
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # runnable as a plain script

B = 32
H = 32
N = 256
d = 128
NUM_TESTS = 200
DTYPE = torch.float32  # FP32 is better for testing accuracy (reference comment, test_correctness.py:14)


def run(num_tests=NUM_TESTS, verbose=True):
    from flash_attention_dlrs_amd import flash_attention_forward
    gpu = torch.device("cuda")
    test_result_fwd = torch.zeros(num_tests, dtype=torch.int32)
    worst = 0.0
    for test in range(num_tests):
        torch.manual_seed(test)
        Q = torch.randn(B, H, N, d, device=gpu, dtype=DTYPE)
        K = torch.randn(B, H, N, d, device=gpu, dtype=DTYPE)
        V = torch.randn(B, H, N, d, device=gpu, dtype=DTYPE)
        O_torch = torch.nn.functional.scaled_dot_product_attention(Q, K, V, scale=1)
        O_flash, L_flash = flash_attention_forward(Q, K, V, dev=gpu)
        worst = max(worst, (O_torch - O_flash).abs().max().item())
        if torch.allclose(O_torch, O_flash, atol=1e-4, rtol=1e-5):
            test_result_fwd[test] = 1
        elif verbose:
            print(O_torch)
            print(O_flash)
    ok = int(test_result_fwd.sum().item())
    print(f"{ok} out of {num_tests} forward tests succeeded!")
    print(f"max |O_torch - O_flash| over all tests: {worst:.3e} (north_star bound 1e-3)")
    return ok, worst


@pytest.mark.gpu
def test_reference_correctness_script_forward():
    ok, worst = run(num_tests=12, verbose=False)
    assert ok == 12
    assert worst <= 1e-3


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else NUM_TESTS
    ok, _ = run(n)
    sys.exit(0 if ok == n else 1)
