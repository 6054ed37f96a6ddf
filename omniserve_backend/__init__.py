"""Drop-in alias: ``import omniserve_backend.qgemm_w4a8_per_chn`` etc. resolve to the MI355X
implementation in omniserve_amd.backend, so the reference's unmodified Python layers
(omniserve/modeling/layers/*) bind to the HIP kernels."""
import importlib
import sys

from omniserve_amd import backend as _backend

for _name in ("qgemm_w4a8_per_chn", "qgemm_w4a8_per_group", "qgemm_w8a8", "fused_kernels",
              "layernorm_ops", "activation_ops", "fused_attention_pure_dense",
              "fused_attention_fine_grained_dense", "fused_attention_fine_grained_sparse",
              "fused_attention_per_tensor_dense", "fused_attention_per_tensor_sparse",
              "fused_attention_ctx_pool", "fused_attention_selector"):
    _mod = importlib.import_module("omniserve_amd.backend." + _name)
    sys.modules[__name__ + "." + _name] = _mod
    globals()[_name] = _mod
