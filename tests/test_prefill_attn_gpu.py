"""Prefill attention (block_sparse_attn / flash_attn shims) vs the f64 oracle: 1e-3 relative
(|got - ref| <= 1e-3 |ref| + 1e-3 max|ref|), short ragged batches and the long-sequence regimes."""
import numpy as np
import pytest
import torch

from oracle import attention as oa
from tests.util import assert_attention_close, dev, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[0, 1, 2], ids=["rows16", "rows32", "pingpong"])
def kernel_form(request):
    """Every case runs on both forms of the kernel: 16 query rows per wave (16x16x32 MFMA, the default) and 32 rows per wave
    (32x32x16 MFMA, csrc/attn_prefill.hip: prefill_attn32_kernel), and the ping-pong schedule of the 16-row form (prefill_attn_pp_kernel)."""
    from omniserve_amd import _lib
    _lib.lib().omni_prefill_set_variant(request.param)
    yield request.param
    _lib.lib().omni_prefill_set_variant(0)


def _case(seq_lens, Hq, Hk, seed, streaming=None, strided=True, block=False):
    import block_sparse_attn as bsa
    rng = np.random.default_rng(seed)
    L = int(sum(seq_lens))
    D = 128
    qkv = rng.standard_normal((L, (Hq + 2 * Hk) * D)).astype(np.float16)
    q = qkv[:, : Hq * D].reshape(L, Hq, D); k = qkv[:, Hq * D:(Hq + Hk) * D].reshape(L, Hk, D)
    v = qkv[:, (Hq + Hk) * D:].reshape(L, Hk, D)
    cu = np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)
    d = to_dev(qkv)
    if strided:   # views of the fused qkv buffer, as the reference passes them
        qd = d[:, : Hq * D].view(L, Hq, D); kd = d[:, Hq * D:(Hq + Hk) * D].view(L, Hk, D); vd = d[:, (Hq + Hk) * D:].view(L, Hk, D)
    else:
        qd, kd, vd = to_dev(q), to_dev(k), to_dev(v)
    cu_d = to_dev(cu)
    if streaming is None:
        out = bsa.flash_attn_varlen_func(qd, kd, vd, cu_d, cu_d, max(seq_lens), max(seq_lens), dropout_p=0.0, causal=True)
        want = oa.varlen_attention(q, k, v, cu, cu, True)
    else:
        hmt, sink, local = streaming
        hm = np.repeat(np.asarray(hmt, np.int32), Hq // Hk)
        si = np.asarray([sink, local] * Hq, np.int32)
        fn = bsa.block_streaming_attn_func if block else bsa.token_streaming_attn_func
        out = fn(qd, kd, vd, cu_d, cu_d, to_dev(hm), to_dev(si), max(seq_lens), max(seq_lens))
        want = oa.varlen_attention(q, k, v, cu, cu, True, hm, si, block=128 if block else 1)
    torch.cuda.synchronize()
    got = out.cpu().numpy().astype(np.float32)
    ref = want.astype(np.float32)
    assert_attention_close(got, ref, "prefill attention, lengths %s" % (list(seq_lens),))


@pytest.mark.parametrize("seq_lens,Hq,Hk", [([1], 4, 1), ([5, 64, 65], 8, 2), ([200, 33], 32, 8), ([300], 8, 8), ([1000, 17], 4, 4)])
def test_dense_causal(seq_lens, Hq, Hk):
    _case(seq_lens, Hq, Hk, seed=sum(seq_lens))


def test_dense_contiguous_inputs():
    _case([70, 130], 8, 2, seed=1, strided=False)


@pytest.mark.parametrize("seq_lens,sink,local", [([300], 16, 64), ([513, 90], 128, 256), ([40], 128, 256), ([700], 4, 33)])
def test_token_streaming_heads(seq_lens, sink, local):
    # kv heads alternate dense (0) / streaming (-1), expanded to q heads like ctx_attn_init.py:28-47
    _case(seq_lens, 8, 4, seed=len(seq_lens) + sink, streaming=([0, -1, -1, 0], sink, local))


@pytest.mark.parametrize("seq_lens,sink,local", [([300], 1, 1), ([1000, 129], 1, 2), ([40], 1, 1), ([700, 128, 127], 2, 3),
                                                 ([1500], 0, 2)])
def test_block_streaming_heads(seq_lens, sink, local):
    """block_streaming_attn_func: (sink, local) in blocks of 128 tokens; sequences shorter than one block, lengths on and next to
    block boundaries, no sink at all."""
    _case(seq_lens, 8, 4, seed=len(seq_lens) + 7 * sink + local, streaming=([0, -1, -1, 0], sink, local), block=True)


# ---- long sequences: the regimes the published prefill numbers come from (XCD-ordered 1-D grid, whole-tile skipping of
# the causal / Lambda masks, LDS-DMA double buffering over hundreds of key tiles).  The f64 oracle is evaluated chunk
# by chunk over the query rows (oracle.attention.varlen_attention_rows). ------------------------------------------------
@pytest.mark.parametrize("split", [1, 2, 4, 8])
@pytest.mark.parametrize("Hq,Hk,classes", [(32, 8, [0, -1, -1, 0, -1, 0, -1, -1]),      # three dense kv heads, irregular
                                           (8, 2, [-1, 0]), (6, 3, [0, -1, -1]),        # Hk not a divisor of 8: W is raised
                                           (4, 1, [-1])])
def test_head_to_xcd_maps_give_the_same_result(split, Hq, Hk, classes):
    """prefill_map_block (csrc/attn_prefill.hip): whatever the number of XCDs a kv head's query tiles are dealt to, and
    whatever the dense / streaming pattern, every (sequence, head, tile) is computed exactly once."""
    from omniserve_amd import _lib
    _lib.lib().omni_prefill_set_xcd_split(split)
    try:
        _case([700, 129, 1], Hq, Hk, seed=7 + split, streaming=(classes, 32, 96))
    finally:
        _lib.lib().omni_prefill_set_xcd_split(0)


def _long_case(L, Hq, Hk, seed, streaming=None, rows=None, lens=None):
    import block_sparse_attn as bsa
    rng = np.random.default_rng(seed)
    D = 128
    lens = [L] if lens is None else lens
    T = int(sum(lens))
    qkv = rng.standard_normal((T, (Hq + 2 * Hk) * D)).astype(np.float16)
    q = qkv[:, : Hq * D].reshape(T, Hq, D); k = qkv[:, Hq * D:(Hq + Hk) * D].reshape(T, Hk, D)
    v = qkv[:, (Hq + Hk) * D:].reshape(T, Hk, D)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    d = to_dev(qkv)
    qd = d[:, : Hq * D].view(T, Hq, D); kd = d[:, Hq * D:(Hq + Hk) * D].view(T, Hk, D); vd = d[:, (Hq + Hk) * D:].view(T, Hk, D)
    cu_d = to_dev(cu)
    hm = si = None
    if streaming is None:
        out = bsa.flash_attn_varlen_func(qd, kd, vd, cu_d, cu_d, max(lens), max(lens), dropout_p=0.0, causal=True)
    else:
        hmt, sink, local = streaming
        hm = np.repeat(np.asarray(hmt, np.int32), Hq // Hk)
        si = np.asarray([sink, local] * Hq, np.int32)
        out = bsa.token_streaming_attn_func(qd, kd, vd, cu_d, cu_d, to_dev(hm), to_dev(si), max(lens), max(lens))
    torch.cuda.synchronize()
    rows = np.arange(T) if rows is None else np.unique(rows)
    got = out.cpu().numpy().astype(np.float32)[rows]
    assert torch.isfinite(out.float()).all().item()
    ref = oa.varlen_attention_rows(q, k, v, cu, cu, rows, True, hm, si).astype(np.float32)
    assert_attention_close(got, ref, "prefill attention, %d tokens, %d rows checked" % (T, len(rows)))


def test_long_lambda_heads_configs3_window():
    """L > sink + local at the BASELINE configs[3] parameters (sink 128 / local 8192): 2 dense + 2 streaming kv heads,
    every query row checked (rows past 8320 exercise the skipped middle tiles of the Lambda mask)."""
    _long_case(10000, 8, 4, seed=3, streaming=([0, -1, -1, 0], 128, 8192))


def test_long_dense_16k():
    _long_case(16384, 2, 1, seed=4)


def test_long_dense_two_sequences_ragged():
    _long_case(0, 4, 2, seed=6, lens=[5000, 3001])


def test_long_streaming_only_64k():
    """L = 65536, streaming heads only: 512 query tiles per head, each attending 128 sink + 8192 local keys; a sample of
    rows (the first and last tiles in full, every tile boundary region around sink + local, 4096 random rows)."""
    L = 65536
    rng = np.random.default_rng(9)
    rows = np.concatenate([np.arange(0, 384), np.arange(8192 - 64, 8192 + 384), np.arange(L - 384, L),
                           rng.choice(L, 4096, replace=False)])
    _long_case(L, 2, 2, seed=8, streaming=([-1, -1], 128, 8192), rows=rows)
