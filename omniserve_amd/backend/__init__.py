"""Host-side mirror of the reference's ``omniserve_backend`` extension package
(kernels/setup.py:156-333): same sub-module names, same function names, same positional
arguments, same ownership rules (caller allocates outputs unless the reference returns a
tensor).  Every function only marshals pointers/sizes into the C ABI of libomniserve_hip.so.
"""
from . import (activation_ops, fused_attention_ctx_pool, fused_attention_selector, prefill_attn, fused_attention_fine_grained_dense, fused_attention_fine_grained_sparse, fused_attention_per_tensor_dense,
               fused_attention_per_tensor_sparse, fused_attention_pure_dense, fused_ext,  # noqa: F401
               fused_kernels, layernorm_ops, qgemm_w4a8_per_chn, qgemm_w4a8_per_group, qgemm_w8a8)
