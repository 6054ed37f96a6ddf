"""Import shim: the reference imports flash_attn (llama_w4a8_unpad.py:36) but routes every call through
block_sparse_attn."""
from omniserve_amd.backend.prefill_attn import flash_attn_varlen_func  # noqa: F401
