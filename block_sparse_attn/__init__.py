"""Drop-in for the un-vendored `block_sparse_attn` package the reference imports
(omniserve/modeling/layers/ctx_attn/ctx_attn_func.py:3-7): MI355X implementation in omniserve_amd."""
from omniserve_amd.backend.prefill_attn import (block_streaming_attn_func, flash_attn_varlen_func,  # noqa: F401
                                                token_streaming_attn_func)
