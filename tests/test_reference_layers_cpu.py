"""The reference's OWN Python layers (omniserve/modeling/layers/*.py, unmodified, from /root/reference) running on top
of this repo's `omniserve_backend` / `block_sparse_attn` mirror on the CPU -- the drop-in boundary exercised from the
reference's side (build container only: skipped where /root/reference is absent).

tests/refstack.py swaps the ctypes handle of libomniserve_hip.so for an object with the same `omni_*` entry points that
reads the raw pointers as host memory and computes with oracle/.  Every call below therefore goes
    reference module  ->  omniserve_backend.<module>.<function>(positional tensors)  ->  mirror marshalling
    (data_ptr / sizes / strides, argument checks)  ->  C-ABI entry point,
with the reference's real argument order and values, and each result is checked twice: bit-exact against the oracle
called directly with the buffers BY NAME (a swapped or mis-sized argument in the mirror shows up), and within a float
tolerance against an independent textbook evaluation (dequantised weights / un-quantised attention)."""
import types

import numpy as np
import pytest
import torch

from tests import refstack

pytestmark = pytest.mark.skipif(not refstack.reference_available(), reason="/root/reference is not on this machine")

from omniserve_amd import ckpt  # noqa: E402
from oracle import attention as oattn  # noqa: E402
from oracle import elementwise as oe  # noqa: E402
from oracle import kv4, w4a8  # noqa: E402


def _bits(t):
    return t.detach().numpy().view(np.uint16)


@pytest.mark.parametrize("group_size", [-1, 128])
def test_reference_w4a8_linear_over_the_mirror(group_size):
    with refstack.reference_over_mirror() as lib:
        from omniserve.modeling.layers.quantized_linear.w4a8_linear import W4A8OF16LinearDynamicInputScale as RefLinear
        g = torch.Generator().manual_seed(3)
        N, K, M = 128, 256, 5
        w = torch.randn((N, K), generator=g) * 0.05
        fake, s1, s2, z = ckpt.qoq_quantize_weight(w, group_size)
        lin = torch.nn.Linear(K, N, bias=False)
        lin.weight.data = fake.clone()
        layer = RefLinear.from_linear(lin, 4, group_size, s1_scale=s1.float(), s2_scale=s2, zeros=z)     # reference packer
        ours = ckpt.convert_linear(fake, s1.float(), z, group_size, s2)                                   # our converter
        assert torch.equal(layer.qweight, ours["qweight"])
        x = (torch.randn((M, K), generator=g)).half()
        q, sa, asum = oe.quant_per_token(x.numpy(), fuse_sum=True)
        out = torch.full((M + 2, N), 7.0, dtype=torch.float16)[1:M + 1]          # a row-slice view, as upstream passes
        layer(torch.from_numpy(q), torch.from_numpy(sa), torch.from_numpy(asum), out)
        if group_size == -1:
            assert lib.calls == ["omni_w4a8_per_chn_gemm"]
            want = w4a8.gemm_per_chn(q, layer.qweight.numpy(), layer.s1_scales.numpy(), sa, layer.s1_szeros.numpy(), asum)
        else:
            assert lib.calls == ["omni_w4a8_per_group_gemm"]
            want = w4a8.gemm_per_group(q, layer.qweight.numpy(), layer.s2_zeros.numpy(), layer.s2_scales.numpy(),
                                       layer.s1_scales.numpy(), sa)
        assert np.array_equal(_bits(out), want.view(np.uint16))
        # textbook check: (dequantised activations) @ (the fake-quantised weight)^T
        a_deq = q.astype(np.float64) * sa.astype(np.float64)[:, None]
        ref = a_deq @ fake.numpy().astype(np.float64).T
        assert np.allclose(out.float().numpy(), ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())


def test_reference_w8a8_linear_over_the_mirror():
    with refstack.reference_over_mirror() as lib:
        from omniserve.modeling.layers.quantized_linear.w8a8_linear import W8A8OF16LinearDynamicInputScale as RefLinear
        g = torch.Generator().manual_seed(4)
        N, K, M = 64, 128, 3
        layer = RefLinear(K, N, bias=False)
        layer.weight.data = torch.randint(-127, 128, (N, K), generator=g, dtype=torch.int8)
        layer.dequant_scale.data = (torch.rand((N,), generator=g) * 0.01 + 0.001).to(layer.dequant_scale.dtype)
        x = torch.randn((M, K), generator=g).half()
        q, sa, _ = oe.quant_per_token(x.numpy(), fuse_sum=False)
        out = torch.empty((M, N), dtype=torch.float16)
        layer(torch.from_numpy(q), torch.from_numpy(sa), out)
        assert lib.calls == ["omni_w8a8_gemm"]
        want = w4a8.gemm_w8a8(q, layer.weight.numpy(), layer.dequant_scale.half().numpy(), sa)
        assert np.array_equal(_bits(out), want.view(np.uint16))


def test_reference_static_scale_and_gelu_modules_over_the_mirror():
    """The modules no Llama model instantiates (activation.py:84-157, layernorm.py:19-43,104-155): their calls bind to
    the mirror's overloads and land on the right C-ABI entry with the right operands."""
    with refstack.reference_over_mirror() as lib:
        from omniserve.modeling.layers.activation import DequantSiluAndMulQuant, FastGELU, NewGELU
        from omniserve.modeling.layers.layernorm import DequantAddResidualI8RMSNormQuant, RMSNorm, RMSNormGeneral
        g = torch.Generator().manual_seed(9)
        T, H = 5, 256
        acc = torch.randint(-50000, 50000, (T, H), generator=g, dtype=torch.int32)
        res = torch.randn((T, H), generator=g).half()
        # dequant + residual + T5-style norm + static quant: per-tensor and per-token dequant scale
        for per_token in (False, True):
            lib.calls.clear()
            m = DequantAddResidualI8RMSNormQuant(H, dequant_scale=0.0003, use_per_token_dequant=per_token, eps=1e-6)
            m.weight.data = (20.0 * (1.0 + 0.1 * torch.randn((H,), generator=g))).half()
            tok = (0.5 + torch.rand((T,), generator=g)).half() if per_token else None
            r = res.clone()
            r_out, q = m(r, acc, tok)
            assert lib.calls == ["omni_dequant_add_residual_rms_norm_quant"] and r_out is r
            sc = (tok * m.dequant_scale.item()).numpy() if per_token else float(m.dequant_scale.item())
            want_q, want_r = oe.dequant_add_residual_rms_norm_quant(acc.numpy(), res.numpy(), m.weight.data.numpy(), sc, 1e-6)
            assert np.array_equal(q.numpy(), want_q) and np.array_equal(_bits(r), want_r.view(np.uint16))
            xf = acc.double().numpy() * (np.asarray(sc, np.float64).reshape(-1, 1) if per_token else sc) + res.double().numpy()
            y = xf / np.sqrt((xf * xf).mean(1, keepdims=True) + 1e-6) * m.weight.data.double().numpy()
            assert np.abs(q.numpy() - np.clip(np.rint(y), -128, 127)).max() <= 1          # textbook, within a code
        # int32 gate/up -> silu*mul -> int8, static and per-token output scale
        gu = torch.randint(-40000, 40000, (T, 2 * H), generator=g, dtype=torch.int32)
        for per_token in (True, False):
            lib.calls.clear()
            m = DequantSiluAndMulQuant(dequant_scale=1e-4, quant_scale=0.05, use_per_token_quant=per_token)
            outs = m(gu)
            assert lib.calls == ["omni_dequant_silu_and_mul_quant"]
            sg = float(m.dequant_scale.item())
            if per_token:
                q, s, _ = oe.dequant_silu_and_mul_quant(gu.numpy(), sg, sg)
                assert np.array_equal(outs[0].numpy(), q) and np.array_equal(outs[1].numpy(), s)
            else:
                assert np.array_equal(outs[0].numpy(), oe.dequant_silu_and_mul_quant(gu.numpy(), sg, sg, float(m.quant_scale.item())))
            x64, y64 = gu[:, :H].double().numpy() * sg, gu[:, H:].double().numpy() * sg
            t = x64 / (1 + np.exp(-x64)) * y64
            deq = outs[0].double().numpy() * (outs[1].double().numpy()[:, None] if per_token else float(m.quant_scale.item()))
            step = outs[1].max().item() if per_token else float(m.quant_scale.item())
            inside = np.abs(t) < 127 * step if not per_token else np.ones_like(t, bool)      # the static scale saturates
            assert np.abs(deq - t)[inside].max() <= 0.51 * step + 1e-6
            assert np.all(np.abs(outs[0].numpy().astype(np.int32)[~inside]) >= 127)
        # GELUs
        x = (2.0 * torch.randn((T, H), generator=g)).half()
        for mod, fn in ((NewGELU(), oe.gelu_new), (FastGELU(), oe.gelu_fast)):
            lib.calls.clear()
            out = mod(x)
            assert lib.calls == ["omni_gelu"]
            assert np.array_equal(_bits(out), fn(x.numpy()).view(np.uint16))
            assert np.abs(out.float().numpy() - torch.nn.functional.gelu(x.float(), approximate="tanh").numpy()).max() < 4e-3
        # rms_norm(use_quant=True) and the per-tensor general norm
        lib.calls.clear()
        n1 = RMSNorm(H, eps=1e-5, use_quant=True)
        n1.weight.data = (20.0 * (1.0 + 0.1 * torch.randn((H,), generator=g))).half()
        q = n1(x)
        assert lib.calls == ["omni_rms_norm_quant"] and q.dtype == torch.int8
        assert np.array_equal(q.numpy(), oe.rms_norm_quant(x.numpy(), n1.weight.data.numpy(), 1e-5))
        lib.calls.clear()
        n2 = RMSNormGeneral(H, act_sum=False, eps=1e-5, use_per_token_quant=False)
        n2.weight.data = (1.0 + 0.1 * torch.randn((H,), generator=g)).half()
        qb, scaling = torch.empty((T, H), dtype=torch.int8), torch.tensor([21.0], dtype=torch.float16)
        n2(x, qb, scaling)
        assert lib.calls == ["omni_rms_norm_general_static"]
        assert np.array_equal(qb.numpy(), oe.rms_norm_general_static(x.numpy(), n2.weight.data.numpy(), scaling.numpy(), 1e-5))


@pytest.mark.parametrize("act_sum", [True, False])
def test_reference_norm_and_activation_modules_over_the_mirror(act_sum):
    with refstack.reference_over_mirror() as lib:
        from omniserve.modeling.layers.activation import SiluAndMulQuant
        from omniserve.modeling.layers.layernorm import RMSNorm, RMSNormGeneral
        g = torch.Generator().manual_seed(5)
        T, H = 6, 512
        x = torch.randn((T, H), generator=g).half()
        norm = RMSNormGeneral(H, act_sum=act_sum, eps=1e-5, use_per_token_quant=True)
        norm.weight.data = (1.0 + 0.1 * torch.randn((H,), generator=g)).half()     # the engine runs the model in fp16
        qbuf, sbuf, mbuf = torch.empty((T, H), dtype=torch.int8), torch.empty((T,), dtype=torch.float16), torch.empty((T,), dtype=torch.float16)
        norm(x, qbuf, sbuf, mbuf)
        assert lib.calls == ["omni_rms_norm_general_fuse_sum" if act_sum else "omni_rms_norm_general"]
        q, s, m = oe.rms_norm_general(x.numpy(), norm.weight.data.half().numpy(), 1e-5, act_sum)
        assert np.array_equal(qbuf.numpy(), q) and np.array_equal(_bits(sbuf), s.view(np.uint16))
        if act_sum:
            assert np.array_equal(_bits(mbuf), m.view(np.uint16))
        # textbook: y = (x - mean) * rsqrt(mean(x^2) + eps) * gamma, dequantised codes track it
        xf = x.float().numpy().astype(np.float64)
        y = (xf - xf.mean(1, keepdims=True)) / np.sqrt((xf * xf).mean(1, keepdims=True) + 1e-5) * norm.weight.data.half().float().numpy()
        assert np.abs(qbuf.numpy() * sbuf.float().numpy()[:, None] - y).max() <= 0.51 * sbuf.float().numpy().max() + 2e-2

        lib.calls.clear()
        act = SiluAndMulQuant(act_sum=act_sum)
        gu = torch.randn((T, 2 * H), generator=g).half()
        qb2 = torch.empty((T, H), dtype=torch.int8)
        act(gu, qb2, sbuf, mbuf)
        assert lib.calls == ["omni_silu_and_mul", "omni_quant_fuse_sum" if act_sum else "omni_quant"]
        mid = oe.silu_and_mul(gu.numpy())
        q2, s2, m2 = oe.quant_per_token(mid, fuse_sum=act_sum)
        assert np.array_equal(qb2.numpy(), q2) and np.array_equal(_bits(sbuf), s2.view(np.uint16))
        silu = gu[:, :H].float() * torch.sigmoid(gu[:, :H].float()) * gu[:, H:].float()
        assert np.abs(qb2.numpy() * sbuf.float().numpy()[:, None] - silu.numpy()).max() <= 0.51 * sbuf.float().numpy().max() + 2e-2

        lib.calls.clear()
        fin = RMSNorm(H, eps=1e-5).half()
        out = fin(x)
        assert lib.calls == ["omni_rms_norm"]
        assert np.array_equal(_bits(out), oe.rms_norm(x.numpy(), fin.weight.data.half().numpy(), 1e-5).view(np.uint16))


def _paged_pools(B, pages_per_seq, Hk, D, tpb, seed):
    """Two host page pools + the raw-pointer block table [B, 2, pages] the reference builds
    (block_table_utils.py:62-93: base_ptr + block_id * bytes_per_block)."""
    pb = kv4.page_bytes(Hk, D, tpb)
    rng = np.random.default_rng(seed)
    kpool = torch.zeros((B * pages_per_seq, pb), dtype=torch.uint8)
    vpool = torch.zeros((B * pages_per_seq, pb), dtype=torch.uint8)
    kid = torch.from_numpy(rng.permutation(B * pages_per_seq).reshape(B, pages_per_seq))
    vid = torch.from_numpy(rng.permutation(B * pages_per_seq).reshape(B, pages_per_seq))
    tab = torch.stack([kpool.data_ptr() + kid * pb, vpool.data_ptr() + vid * pb], dim=1).to(torch.int64).contiguous()
    return kpool, vpool, kid.numpy(), vid.numpy(), tab


def test_reference_attention_wrappers_over_the_mirror():
    """ApplyBiasRopeUpdateKVCacheWrapper (prefill writer), attention_wrapper (varlen prefill attention) and
    DecodingAttentionWrapper.forward_pure_dense (decode) as llama_w4a8_unpad.py:282-345 drives them."""
    with refstack.reference_over_mirror() as lib:
        from omniserve.modeling.layers.ctx_attn.ctx_attn_func import attention_wrapper
        from omniserve.modeling.layers.ctx_update_kv import ApplyBiasRopeUpdateKVCacheWrapper
        from omniserve.modeling.layers.decoding_attention import DecodingAttentionWrapper
        import omniserve_backend.fused_attention_fine_grained_dense as fgd
        Hq, Hk, D, tpb, base = 4, 2, 128, 64, 500000.0
        kvcfg = {"INT4_ENABLED": True, "ZEROS_ENABLED": True}
        lens = [70, 33]
        B, T = len(lens), sum(lens)
        g = torch.Generator().manual_seed(6)
        qkv = torch.randn((T, (Hq + 2 * Hk) * D), generator=g).half()
        qkv0 = qkv.clone()
        kpool, vpool, kid, vid, tab = _paged_pools(B, 2, Hk, D, tpb, 0)
        cu = torch.tensor([0, 70, 103], dtype=torch.int32)
        pad = fgd.compute_padding_offsets(cu, max(lens), T)
        assert pad.tolist() == kv4.compute_padding_offsets(cu.numpy(), max(lens)).tolist()
        meta = types.SimpleNamespace(retrieval_context_lens=torch.tensor(lens, dtype=torch.int32),
                                     streaming_context_lens=torch.tensor(lens, dtype=torch.int32), padding_offsets=pad,
                                     retrieval_block_tables=[tab], streaming_block_tables=[None], max_seq_len=max(lens),
                                     cu_seqlens=cu)
        flags, rank = torch.ones((Hk,), dtype=torch.int32), torch.arange(Hk, dtype=torch.int32)
        writer = ApplyBiasRopeUpdateKVCacheWrapper(0, Hq, Hk, tpb, D, base, None, 8192, True, "fine_grained", kvcfg, True)
        lib.calls.clear()
        writer(qkv, meta, flags, rank, 0, 0, 0, 0, Hk, 0, None)
        assert lib.calls == ["omni_kv4_prefill_write"]
        # oracle directly: same in-place RoPE, same pages
        ok, ov = kv4.PagedKV4(2 * B, Hk, D, tpb), kv4.PagedKV4(2 * B, Hk, D, tpb)
        want_qkv = kv4.prefill_write(qkv0.numpy(), lens, ok, ov, kid, vid, Hq, Hk, D, base)
        assert np.array_equal(_bits(qkv), want_qkv.view(np.uint16))
        assert np.array_equal(kpool.numpy(), ok.pool) and np.array_equal(vpool.numpy(), ov.pool)

        # prefill attention over the post-RoPE q, k (strided views of the fused buffer)
        q = qkv[:, : Hq * D].reshape(T, Hq, D)
        k = qkv[:, Hq * D:(Hq + Hk) * D].reshape(T, Hk, D)
        v = qkv[:, (Hq + Hk) * D:].reshape(T, Hk, D)
        lib.calls.clear()
        ctx = attention_wrapper(q, k, v, cu_seqlens_q=cu, cu_seqlens_k=cu, max_seqlen_q=70, max_seqlen_k=70,
                                dropout_p=0.0, causal=True, head_mask_type=None, streaming_info=None)
        assert lib.calls == ["omni_prefill_attention"] and tuple(ctx.shape) == (T, Hq, D)
        want_ctx = oattn.varlen_attention(q.numpy(), k.numpy(), v.numpy(), cu.numpy(), cu.numpy(), True)
        assert np.array_equal(_bits(ctx), want_ctx.view(np.uint16))

        # one decode step (lengths include the new token, decoding_attention.py:156-158)
        dec = DecodingAttentionWrapper(0, False, D, None, 8192, tpb, D, base, None, True, "fine_grained", kvcfg, True, 0,
                                       16, 4096, 2048, 4)
        new = torch.randn((B, (Hq + 2 * Hk) * D), generator=g).half()
        dq = new[:, : Hq * D].reshape(B, Hq, D)
        dk = new[:, Hq * D:(Hq + Hk) * D].reshape(B, Hk, D)
        dv = new[:, (Hq + Hk) * D:].reshape(B, Hk, D)
        dlens = torch.tensor([n + 1 for n in lens], dtype=torch.int32)
        meta.retrieval_context_lens, meta.max_seq_len = dlens, int(dlens.max())
        lib.calls.clear()
        out, sel = dec(dq, dk, dv, meta, flags, rank, 0, 0, 0, 0, Hk, 0, None, torch.ones(2))
        assert lib.calls == ["omni_kv4_decode_attention"] and sel is None and tuple(out.shape) == (B, Hq, D)
        want = kv4.decode_attention(dq.numpy(), dk.numpy(), dv.numpy(), dlens.numpy(), ok, ov, kid, vid, base)
        assert np.array_equal(_bits(out), want.view(np.uint16))
        assert np.array_equal(kpool.numpy(), ok.pool) and np.array_equal(vpool.numpy(), ov.pool)     # appended rows too
        # textbook: attention over the UN-quantised post-RoPE history (4-bit cache noise bounds the difference)
        for b, L in enumerate(lens):
            o0 = int(cu[b])
            qr = kv4.rope_neox(dq[b].numpy(), np.full((Hq,), L), base).astype(np.float64)
            kr = kv4.rope_neox(dk[b].numpy(), np.full((Hk,), L), base).astype(np.float64)
            for h in range(Hq):
                kk = np.concatenate([k[o0:o0 + L, h // 2].numpy().astype(np.float64), kr[h // 2][None]], 0)
                vv = np.concatenate([v[o0:o0 + L, h // 2].numpy().astype(np.float64), dv[b, h // 2].numpy().astype(np.float64)[None]], 0)
                s = kk @ qr[h] / np.sqrt(D)
                p = np.exp(s - s.max()); p /= p.sum()
                assert np.abs(out[b, h].float().numpy() - p @ vv).max() < 0.15
