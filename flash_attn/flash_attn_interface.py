from omniserve_amd.backend.prefill_attn import flash_attn_varlen_func  # noqa: F401
