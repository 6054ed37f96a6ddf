"""Oracle (CPU, numpy) for the per-tensor KV8 cache of LServe's published configuration
(`--precision w8a8kv8 --kv-quant-granularity per_tensor`, scripts/lserve_benchmark/launch.sh:6-7).
Test infrastructure only.  Parity unpinned by reference tests (the reference holds no golden vectors
for this path); the arithmetic below restates, citations relative to
kernels/csrc/fused_attention/ of the upstream checkout:

* static scales          kv_scale_quant_orig fp32 [2] (K, V), kv_scale_orig_quant = 1/kv_scale_quant_orig
                         (omniserve/modeling/layers/decoding_attention.py:202-203, llama_w8a8_unpad.py:262-263);
                         per_tensor => no zero points (omniserve/engine/arg_utils.py:500-503)
* quantise / store       int8 = cvt.rni.sat.s8( kv_scale_orig_quant * f32(x) )
                         (common/decoderMaskedMultiheadAttentionUtils.h:1761-1771,2041-2048; call sites
                         fused_attention_per_tensor/per_tensor_common/applyBiasRopeUpdateKVCache.h:508-511,
                         dense_attention/decoderMaskedMultiheadAttentionTemplate.hpp:1377,2162)
* dequantise             fp16( kv_scale_quant_orig * f32(int8) )
                         (common/decoderMaskedMultiheadAttentionUtils.h:2086-2093, float_from_int8 :1485-1532)
* prefill tail           the writer still stores h(absmax/127) of every (token, head) row in the scale slot of
                         the page tail (per_tensor_common/applyBiasRopeUpdateKVCache.h:387-414); the decode
                         kernel does not (its copy of that code is commented out, dense Template.hpp:1280-1335)
* page                   int8 data [H][tokens_per_block][Dh] | fp16 scale [H][tpb] | fp16 zero slot [H][tpb]
                         (the tail always reserves 4 B per token-head, omniserve/worker/cache_engine.py:78)
                         | optional K statistics as in the KV4 pages.
Head classes, rings, page selection and the attention arithmetic are those of oracle/kv4.py (the per_tensor
kernels differ from the fine_grained ones only in the lines cited above; q is NOT reordered for int8 pages,
dense Template.hpp:1452-1470).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
F16 = np.float16


def page_bytes(num_kv_heads: int, head_dim: int, tokens_per_block: int = 64) -> int:
    return num_kv_heads * tokens_per_block * head_dim + 2 * num_kv_heads * tokens_per_block * 2


def kv8_quantize(x_h, scale_orig_quant) -> np.ndarray:
    """int8 codes [..., D] = rni_sat_s8(scale * f32(x)) (round half to even, saturate; NaN -> 0)."""
    v = (np.asarray(x_h, F16).astype(F32) * F32(scale_orig_quant)).astype(F32)
    r = np.rint(v)
    r = np.where(np.isnan(r), 0.0, r)
    return np.clip(r, -128, 127).astype(np.int8)


def kv8_dequant(codes, scale_quant_orig) -> np.ndarray:
    """-> float32 values of fp16( scale * f32(int8) )."""
    return (np.asarray(codes, np.int8).astype(F32) * F32(scale_quant_orig)).astype(F32).astype(F16).astype(F32)


class PagedKV8:
    """Page pool of int8 per-tensor pages; same interface as oracle.kv4.PagedKV4 (pool: uint8 [pages, bytes])."""

    def __init__(self, num_pages, num_kv_heads, head_dim, scale_quant_orig, tokens_per_block=64, fill=0,
                 stats_sub_chunk=0):
        self.H, self.D, self.TPB = num_kv_heads, head_dim, tokens_per_block
        self.row_bytes = head_dim
        self.bytes_per_seq = num_kv_heads * tokens_per_block * head_dim
        self.page_bytes = page_bytes(num_kv_heads, head_dim, tokens_per_block)
        if stats_sub_chunk:
            self.page_bytes += 2 * (tokens_per_block // stats_sub_chunk) * num_kv_heads * head_dim * 2
        self.pool = np.full((num_pages, self.page_bytes), fill, np.uint8)
        self.scale_quant_orig = F32(scale_quant_orig)
        self.scale_orig_quant = (F32(1.0) / F32(scale_quant_orig)).astype(F32)

    def data(self, page):
        return self.pool[page, : self.bytes_per_seq].view(np.int8).reshape(self.H, self.TPB, self.D)

    def scales(self, page):
        o = self.bytes_per_seq
        return self.pool[page, o: o + self.H * self.TPB * 2].view(F16).reshape(self.H, self.TPB)

    def write_token(self, page, slot, head, x_h, decode=False):
        self.data(page)[head, slot] = kv8_quantize(x_h, self.scale_orig_quant)
        if not decode:   # prefill writer only: per-row statistics nobody reads
            amax = np.abs(np.asarray(x_h, F16).astype(F32)).max()
            self.scales(page)[head, slot] = (amax / F32(127.0)).astype(F32).astype(F16)

    def read_row(self, page, head, slot):
        return kv8_dequant(self.data(page)[head, slot], self.scale_quant_orig)
