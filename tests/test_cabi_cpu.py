"""CPU-side checks of the drop-in boundary: the library builds/loads and exports exactly the
symbols include/omniserve_hip.h declares; the Python mirror has the reference's module and
function names and arities.  No compute calls (no GPU here)."""
import inspect
import os
import re

import pytest

from omniserve_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "omniserve_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(omni_[a-z0-9_]+)\s*\(", src))


def test_library_exports_every_declared_symbol():
    h = _lib.lib()
    syms = _header_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(h, s), "library does not export %s" % s
    assert syms == set(_lib.PROTOTYPES), "ctypes prototypes out of sync with the header"
    assert h.omni_abi_version() == 4


def test_invalid_arguments_are_rejected_without_a_gpu():
    h = _lib.lib()
    # null pointers / bad shapes must return -EINVAL before any launch
    assert h.omni_w4a8_per_chn_gemm(None, None, None, None, None, None, None, 16, 64, 64, 64, None, 0, None) == -22
    assert h.omni_quant(None, None, None, 4, 128, None) == -22
    # the off-path overloads (csrc/offpath.hip): null pointers, rows that are not a multiple of 8, over-long norm rows
    assert h.omni_quant_static(None, None, 0.5, 4, 128, None) == -22
    assert h.omni_quant_static(1, 1, 0.5, 4, 100, None) == -22
    assert h.omni_dequant(1, 1, 0.5, 4, 128, 64, 128, None) == -22          # row stride shorter than the row
    assert h.omni_dequant_add_residual(1, 1, None, None, 0.5, 4, 128, None) == -22
    assert h.omni_rms_norm_quant(1, 1, 1, 1e-6, 4, 16384, None) == -22      # does not fit the LDS copy of the row
    assert h.omni_rms_norm_quant(1, 1, 1, 1e-6, 4, 200, None) == -22        # partial virtual warp
    assert h.omni_rms_norm_general_static(1, 1, 1, None, 1e-6, 4, 128, None) == -22
    assert h.omni_dequant_add_residual_rms_norm_quant(1, 1, None, 1, None, 0.5, 1e-6, 4, 128, None) == -22
    assert h.omni_gelu(1, 1, 2, 4, 128, None) == -22                        # kind is 0 (new) or 1 (fast)
    assert h.omni_dequant_silu_and_mul_quant(1, 1, 1.0, 1.0, 1.0, 1, None, 4, 128, None) == -22   # scale without tmp
    assert h.omni_quant_static(1, 1, 0.5, 0, 128, None) == 0                # nothing to do: no launch
    assert h.omni_gemm_workspace_bytes(16, 4096, 4096) > 0
    assert h.omni_gemm_workspace_bytes(4096, 4096, 4096) == 0


# reference module -> {function: number of positional parameters}  (kernels/csrc/**.h)
REFERENCE_API = {
    "qgemm_w4a8_per_chn": {"gemm_forward_cuda": 7},
    "qgemm_w4a8_per_group": {"gemm_forward_cuda": 7},
    "qgemm_w8a8": {"w8a8_gemm_forward_cuda": 5},
    "fused_kernels": {"invoke_quant": 3, "invoke_quant_fuse_sum": 4, "invoke_dequant": 3,
                      "invoke_dequant_add_residual": 4},
    "layernorm_ops": {"rms_norm": 5, "rms_norm_general": 6, "rms_norm_general_fuse_sum": 7,
                      "invoke_dequant_add_residual_rms_norm_quant": 6},
    # invoke_dequant_silu_and_mul_quant: 5 (float scale_out) or 6 (tensor scale_out, tmp) positional arguments
    "activation_ops": {"silu_and_mul": 2, "gelu_new": 2, "gelu_fast": 2, "invoke_dequant_silu_and_mul_quant": 6},
    "fused_attention_pure_dense": {"single_query_attention": 15, "apply_bias_rope_update_kv_cache": 15,
                                   "compute_padding_offsets": 3},
    "fused_attention_fine_grained_dense": {"apply_bias_rope_update_kv_cache": 27, "compute_padding_offsets": 3,
                                           "single_query_attention": 27},
    "fused_attention_fine_grained_sparse": {"single_query_attention": 30},
    # per_tensor KV8 family (fused_attention_per_tensor/*/fused_attention.h, per_tensor_common/update_kv_cache.h)
    "fused_attention_per_tensor_dense": {"apply_bias_rope_update_kv_cache": 28, "single_query_attention": 29},
    "fused_attention_per_tensor_sparse": {"single_query_attention": 32},
}

THIRD_PARTY_API = {   # un-vendored packages the reference imports for prefill attention
    "block_sparse_attn": ["flash_attn_varlen_func", "token_streaming_attn_func", "block_streaming_attn_func"],
    "flash_attn.flash_attn_interface": ["flash_attn_varlen_func"],
}


def test_python_mirror_has_reference_names_and_arities():
    import importlib
    for mod, fns in REFERENCE_API.items():
        m = importlib.import_module("omniserve_backend." + mod)
        for fn, n in fns.items():
            f = getattr(m, fn)
            assert len(inspect.signature(f).parameters) == n, (mod, fn)


def test_third_party_shims_importable():
    import importlib
    for mod, fns in THIRD_PARTY_API.items():
        m = importlib.import_module(mod)
        for fn in fns:
            assert callable(getattr(m, fn))


def test_ops_fail_loudly_without_device_tensors():
    import pytest
    import torch
    import omniserve_backend.fused_kernels as fk
    x = torch.zeros(2, 128, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        fk.invoke_quant(torch.zeros(2, 128, dtype=torch.int8), x, torch.zeros(2, dtype=torch.float16))


def test_gemm_plans_of_the_baseline_decode_shapes():
    """Host-side planner (no GPU): the plans the decode drivers rely on.  The slab form of the qkv projection
    (fused_ext.decode_arm_qkv_slabs) is taken exactly where the plain qkv GEMV's plan splits K -- not at bs = 16 (one kernel,
    no epilogue launch), at bs = 64 and on a Llama-2-70B TP = 8 shard at bs = 128; and the workspace bound covers the
    deferred (slab-only) plans of every projection."""
    import ctypes
    from omniserve_amd import _lib
    lib = _lib.lib()

    def plan(M, N, K, kalign):
        mb, waves, sk = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
        lib.omni_gemm_get_plan(M, N, K, kalign, ctypes.byref(mb), ctypes.byref(waves), ctypes.byref(sk))
        return mb.value, waves.value, sk.value

    assert plan(16, 6144, 4096, 64) == (1, 1, 1)            # Llama-3-8B qkv at bs = 16: 96 single-wave tiles, no split
    assert plan(16, 28672, 4096, 64)[2] == 1                # gate_up: one kernel (the SiLU-epilogue form needs that)
    mb, _, sk = plan(64, 6144, 4096, 128)                   # configs[2]: 64-row tiles, K split -> slab epilogue / slab form
    assert mb == 4 and sk > 1
    assert plan(128, 1280, 8192, 64)[2] > 1                 # 70B TP = 8 shard's qkv at bs = 128
    assert plan(4096, 4096, 4096, 64) == (8, 4, 1)          # prefill regime: 128 x 256 tiles, no split
    for (M, N, K) in ((16, 4096, 14336), (64, 4096, 14336), (128, 8192, 3584), (1, 6144, 4096), (16, 6144, 4096)):
        assert lib.omni_gemm_partial_workspace_bytes(M, N, K) >= M * N * 4
    assert lib.omni_gemm_workspace_bytes(4096, 4096, 4096) == 0
    # ABI 4: the plain entry points pin nothing where no plan splits K (gate_up at bs = 256: 224 tiles, K whole), the slab-only
    # forms always get their slab; both sizes hold under every forced mid-M plan
    assert lib.omni_gemm_workspace_bytes(256, 28672, 4096) == 0
    assert lib.omni_gemm_partial_workspace_bytes(256, 28672, 4096) >= 256 * 28672 * 4
    assert lib.omni_gemm_partial_workspace_bytes(513, 4096, 4096) == 0
    for (M, N, K) in ((128, 28672, 4096), (64, 4096, 14336), (96, 8192, 8192)):
        plain, part = lib.omni_gemm_workspace_bytes(M, N, K), lib.omni_gemm_partial_workspace_bytes(M, N, K)
        for forced in (2, 4, 7, 8, 14, 16):
            lib.omni_gemm_set_midm_override(1, forced)
            try:
                for kalign in (64, 128):
                    sk = plan(M, N, K, kalign)[2]
                    assert plain >= (sk * M * N * 4 if sk > 1 else 0), (M, N, K, forced)
                    assert part >= sk * M * N * 4, (M, N, K, forced)
            finally:
                lib.omni_gemm_set_midm_override(-1, 0)
    # fusion level 3's gate (the runners ask before their first step; ADVICE r3: a hidden = 5120 layer passed the old size
    # test and then failed in gemm_silu because its gate_up plan splits K across workgroups)
    assert lib.omni_gemm_rowfree_ok(16, 4096, 4096, 14336, 0) == 1 and lib.omni_gemm_rowfree_ok(16, 4096, 4096, 14336, 1) == 1
    assert lib.omni_gemm_rowfree_ok(1, 4096, 4096, 14336, 2) == 1          # the LServe driver's W8A8 layer at batch 1
    assert lib.omni_gemm_rowfree_ok(16, 5120, 5120, 13824, 0) == 0         # Llama-2-13B-shaped: level 2
    assert lib.omni_gemm_rowfree_ok(17, 4096, 4096, 14336, 0) == 0         # more than the 16-row tile
    assert lib.omni_gemm_rowfree_ok(16, 512, 512, 1024, 0) == 0            # more rows than hidden / 64 rider workgroups


def test_release_library_reads_no_environment():
    """Release hygiene: the planner / debug knobs (OMNI_GEMV_DBG, OMNI_GEMV_NARROW, ...) exist only in
    -DOMNI_TUNING builds; the shipped library neither imports getenv nor carries their names."""
    import shutil
    import subprocess
    from omniserve_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    syms = subprocess.run([nm, "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    blob = open(_lib.LIB_PATH, "rb").read()
    for knob in (b"OMNI_GEMV_DBG", b"OMNI_GEMV_NARROW", b"OMNI_DEFERRED_PART", b"OMNI_W8_SMALL_SPLIT", b"OMNI_GEMM_EXACT",
                 b"OMNI_DECODE_RT", b"OMNI_PREFETCH_DELAY"):
        assert knob not in blob, knob
    assert not hasattr(_lib.lib(), "omni_gemm_set_weight_policy")      # the load policy is per call (omni_prefetch_arm_gemm)


def test_kv_formats_without_a_working_upstream_behaviour_are_named_in_the_error():
    """KV4 without zero points: upstream packs signed codes and decodes them unsigned -- the mirror refuses
    the format by name instead of guessing an arithmetic; per_tensor + int4 likewise."""
    import pytest
    from omniserve_amd.backend import _attn_common
    with pytest.raises(NotImplementedError, match="unsigned"):
        _attn_common._check_cfg(128, True, True, False, 128)
    with pytest.raises(NotImplementedError):
        _attn_common._check_cfg_kv8(128, True, True, False, 128)
    _attn_common._check_cfg(128, True, True, True, 128)          # the fine_grained KV4 format
    _attn_common._check_cfg_kv8(128, True, False, False, 128)    # the per_tensor KV8 format




def test_fast_binding_is_built_and_exports_the_routed_functions():
    """omniserve_amd/csrc_ext/omni_ext.cpp: the pybind11 bodies of the per-step mirror functions.  It imports here (it links
    the library), reports the library's ABI version, and is NOT used without a GPU (the mirror stays on ctypes)."""
    import importlib
    import os
    from omniserve_amd import _lib, build
    if not os.path.exists(build.ext_path()):
        pytest.skip("the fast path was not built on this host (python -m omniserve_amd.build builds it next to the library)")
    _lib.lib()
    e = importlib.import_module("omniserve_amd._omni_ext")
    assert e.abi_version() == _lib.lib().omni_abi_version()
    for name in ("gemm_w4a8_per_chn", "gemm_w4a8_per_group", "gemm_w8a8", "rms_norm_general_fuse_sum_f16", "rms_norm_general_f16",
                 "rms_norm_f16", "quant_fuse_sum_f16", "quant_f16", "silu_and_mul_f16", "decode_attention_kv4"):
        assert callable(getattr(e, name))
    import torch
    if not torch.cuda.is_available():
        assert _lib.fast() is None
