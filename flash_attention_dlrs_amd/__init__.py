"""MI355X-native Flash-Attention-2 forward behind the Python surface of 17ex/flash_attention_dlrs.

Modules mirror the reference's (src/flash_attention_torch.py, src/flash_attention_wrappers.py); the
Triton launch they make is replaced by the C-ABI call fa2_fwd() of libfa2_hip.so (include/fa2_fwd.h),
a hand-written HIP/CDNA4 kernel library.  There is no CPU fallback: without the built library, or on
non-GPU tensors, calls fail loudly.
"""
from .flash_attention_torch import (MIN_TENSOR_SIZE, FlashAttention, FlashAttentionDeterministic,
                                    convert_triton_dtype)
from .flash_attention_wrappers import flash_attention_backward, flash_attention_forward

__all__ = ["FlashAttention", "FlashAttentionDeterministic", "convert_triton_dtype", "MIN_TENSOR_SIZE",
           "flash_attention_forward", "flash_attention_backward"]
