"""Known-answer and property tests of the oracles on the CPU (no GPU, no reference checkout needed).  The reference ships
no golden vectors for these stages, so the answers below are derived by hand from the formulas the
oracle docstrings cite: they pin the rounding points the HIP kernels are then held to
bit for bit on the GPU."""
import numpy as np

from oracle import attention as oa
from oracle import elementwise as oe
from oracle import kv4, kv8

F16, F32 = np.float16, np.float32


def test_rni_sat_s8_is_round_half_even_with_saturation():      # A.3, utils.cuh:79-84
    x = np.array([0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 126.5, 127.5, 300.0, -128.5, -129.0, np.nan, 0.49999997], F32)
    assert oe.rni_sat_s8(x).tolist() == [0, 2, 2, 0, -2, -2, 126, 127, 127, -128, -128, 0, 0]


def test_quant_per_token_scale_is_amax_over_127_but_multiplier_is_127_over_amax():   # A.15, fused_kernels.cu:108-131
    x = np.zeros((1, 64), F16)
    x[0, 0], x[0, 1], x[0, 2], x[0, 3] = 3.0, -1.5, 0.0118, 1.0
    q, scale, s = oe.quant_per_token(x, fuse_sum=True)
    assert scale[0] == F16(F32(3.0) / F32(127.0))
    mult = F32(127.0) / F32(3.0)
    want = oe.rni_sat_s8((x[0].astype(F32) * mult).astype(F32))
    assert np.array_equal(q[0], want) and q[0, 0] == 127 and q[0, 1] == -64        # -63.5 -> -64 (half even)
    assert s[0] == F16(x[0].astype(F32).sum(dtype=F32))                              # fp32 sum of the fp16 inputs


def test_rms_norm_general_subtracts_the_mean_in_the_output_only():      # A.1, layernorm_kernels.cu:26-34,110-149
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((2, 4096)) + 0.75).astype(F16)
    g = np.ones(4096, F16)
    q, scale, s = oe.rms_norm_general(x, g, 1e-5, fuse_sum=True)
    xf = x.astype(np.float64)
    y = (xf - xf.mean(axis=1, keepdims=True)) / np.sqrt((xf * xf).mean(axis=1, keepdims=True) + 1e-5)   # NOT the variance
    amax = np.abs(y.astype(F16).astype(np.float64)).max(axis=1)
    assert np.allclose(scale.astype(np.float64), amax / 127.0, rtol=2e-3)
    ref_q = np.rint(y * (127.0 / amax[:, None]))
    assert np.abs(q.astype(np.float64) - ref_q).max() <= 1          # fp32 vs f64 path: at most one code apart
    assert np.allclose(s.astype(np.float64), y.astype(F16).astype(np.float64).sum(axis=1), atol=0.5)


def test_kv4_params_and_codes_known_answer():                # A.7, applyBiasRopeUpdateKVCache.h:355-376 ; A.3 (& 0xF, not clamped)
    x = np.linspace(-1.5, 3.0, 128).astype(F16)
    sc, ze = kv4.kv4_quant_params(x)
    assert sc == F16(F32(4.5) / F32(15.0)) and ze == F16(F32(22.5) / F32(4.5))        # (max-min)/15, -15*min/(max-min)
    codes = kv4.kv4_quantize(x, sc, ze)
    lo, hi = codes & 0xF, codes >> 4
    assert lo[0] == 0 and hi[-1] == 15                                               # min -> 0, max -> 15
    deq = kv4.kv4_dequant(codes, sc, ze)
    assert np.abs(deq - x.astype(F32)).max() <= 0.5 * float(sc) + 2e-3
    # the nibble is masked, not clamped: a value that quantises to 16 wraps to 0 (A.3)
    z = np.array([0.0] * 127 + [16.0], F16)
    assert (kv4.kv4_quantize(z, F16(1.0), F16(0.0))[-1] >> 4) == 0


def test_kv8_static_scale_codes_known_answer():              # decoderMaskedMultiheadAttentionUtils.h:1761-1771,2041-2048,2086-2093
    oq = F32(1.0) / F32(0.03)
    x = np.array([0.0, 0.03, 0.045, -0.045, 3.81, 3.9, -3.84, -5.0], F16)
    c = kv8.kv8_quantize(x, oq)
    want = np.clip(np.rint((x.astype(F32) * oq).astype(F32)), -128, 127).astype(np.int8)
    assert np.array_equal(c, want) and c[-1] == -128 and c[5] == 127
    d = kv8.kv8_dequant(c, 0.03)
    assert np.array_equal(d, (c.astype(F32) * F32(0.03)).astype(F32).astype(F16).astype(F32))


def test_rope_neox_pairs_and_position_zero_is_identity():     # decoderMaskedMultiheadAttentionUtils.h:1152-1165
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 128)).astype(F16)
    assert np.array_equal(kv4.rope_neox(x, np.zeros(3), 500000.0), x)
    y = kv4.rope_neox(x, np.array([7, 7, 7]), 10000.0)
    i = 5
    ang = np.float64(7) / np.power(10000.0, 2.0 * i / 128.0)
    want = np.cos(ang) * float(x[0, i]) - np.sin(ang) * float(x[0, i + 64])
    assert abs(float(y[0, i]) - want) < 2e-3
    # norms of the (i, i+64) pairs are preserved
    n0 = x[:, :64].astype(np.float64) ** 2 + x[:, 64:].astype(np.float64) ** 2
    n1 = y[:, :64].astype(np.float64) ** 2 + y[:, 64:].astype(np.float64) ** 2
    assert np.allclose(n0, n1, rtol=5e-3, atol=5e-3)


def _dense_cache(rng, B, Hk, T, tpb, cls, **kw):
    pages = (T + 8) // tpb + 1
    kc, vc = cls(B * pages, Hk, 128, **kw), cls(B * pages, Hk, 128, **kw)
    kt, vt = rng.permutation(B * pages).reshape(B, pages), rng.permutation(B * pages).reshape(B, pages)
    return kc, vc, kt, vt


def test_kv4_decode_oracle_equals_float_attention_over_the_dequantised_cache():     # Template.hpp:1374-1400,1764-1845 ; A.10, A.11
    rng = np.random.default_rng(2)
    B, Hq, Hk, D, T, tpb = 2, 4, 2, 128, 45, 16
    kc, vc, kt, vt = _dense_cache(rng, B, Hk, T, tpb, kv4.PagedKV4, tokens_per_block=tpb)
    qkv = rng.standard_normal((B * T, (Hq + 2 * Hk) * D)).astype(F16)
    kv4.prefill_write(qkv, [T, T], kc, vc, kt, vt, Hq, Hk, D, 500000.0)
    q = rng.standard_normal((B, Hq, D)).astype(F16)
    k = rng.standard_normal((B, Hk, D)).astype(F16)
    v = rng.standard_normal((B, Hk, D)).astype(F16)
    lens = np.array([T + 1, T + 1], np.int32)
    # history BEFORE the call (the call appends the quantised current token)
    hist = [[(kc.read_tokens(kt[b], h, T), vc.read_tokens(vt[b], h, T)) for h in range(Hk)] for b in range(B)]
    out = kv4.decode_attention(q, k, v, lens, kc, vc, kt, vt, 500000.0)
    qr = np.stack([kv4.rope_neox(q[b], np.full(Hq, T), 500000.0) for b in range(B)]).astype(np.float64)
    kr = np.stack([kv4.rope_neox(k[b], np.full(Hk, T), 500000.0) for b in range(B)]).astype(np.float64)
    for b in range(B):
        for hq in range(Hq):
            hk = hq // (Hq // Hk)
            K = np.concatenate([hist[b][hk][0].astype(np.float64), kr[b, hk][None]], axis=0)    # current token un-quantised
            V = np.concatenate([hist[b][hk][1].astype(np.float64), v[b, hk].astype(np.float64)[None]], axis=0)
            s = K @ qr[b, hq] / np.sqrt(D)
            p = np.exp(s - s.max())
            p = p / (p.sum() + 1e-6)                                                           # A.9
            want = p @ V
            assert np.abs(out[b, hq].astype(np.float64) - want).max() <= 2e-3 * max(1.0, np.abs(want).max())
    # the appended rows are the quantised post-RoPE k and the raw v, at slot tlen (A.11)
    for b in range(B):
        back = kc.read_tokens(kt[b], 0, T + 1)[T]
        assert np.abs(back - kr[b, 0].astype(F32)).max() <= 0.5 * float(kc.scales(int(kt[b][T // tpb]))[0, T % tpb]) + 2e-3


def test_fine_grained_streaming_heads_attend_sink_plus_local_only():        # Appendix B
    rng = np.random.default_rng(3)
    B, Hq, Hk, D, tpb, sink, local = 1, 2, 2, 128, 16, 16, 32
    T = 100
    flags, rank = [1, 0], [0, 0]
    sink_blocks, local_blocks = 1, local // tpb + 1
    rpages, spages = (T + 8) // tpb + 1, sink_blocks + local_blocks
    mk = lambda n: kv8.PagedKV8(n, 1, D, 0.05, tpb)
    rk, rv, sk, sv = mk(rpages), mk(rpages), mk(spages), mk(spages)
    fg = kv4.FineGrainedKV(rk, rv, np.arange(rpages)[None], np.arange(rpages)[None], sk, sv, np.arange(spages)[None],
                           np.arange(spages)[None], flags, rank, sink, local, sink_blocks, local_blocks)
    qkv = rng.standard_normal((T, (Hq + 2 * Hk) * D)).astype(F16)
    kv4.prefill_write_fine_grained(qkv, [T], fg, Hq, Hk, D, 500000.0)
    toks = kv4.attended_tokens(fg, 0, 1, 1, T)
    assert len(toks) == sink + local - 1 and toks[:sink].tolist() == list(range(sink))
    assert toks[sink:].tolist() == list(range(T - (local - 1), T))
    # ring: the streaming pool holds exactly the tokens a streaming head may still need
    for t in toks:
        kcache, _, kp, _, slot = fg.locate(0, 1, int(t))
        assert kcache is sk and 0 <= kp < spages
    q = rng.standard_normal((B, Hq, D)).astype(F16)
    k = rng.standard_normal((B, Hk, D)).astype(F16)
    v = rng.standard_normal((B, Hk, D)).astype(F16)
    out = kv4.decode_attention_fine_grained(q, k, v, np.array([T + 1], np.int32), fg, 500000.0)
    assert np.isfinite(out.astype(F32)).all()
    # per_tensor decode does not touch the tails (decode=True): the prefill's absmax/127 entries survive
    pg, slot = int(fg.retr_k_table[0][T // tpb]), T % tpb
    assert rk.scales(pg)[0, slot] == 0          # never written for the appended token


def test_prefill_attention_oracle_lambda_mask():        # causal & (dense | k < sink | q - k < local)
    rng = np.random.default_rng(4)
    L, Hq, Hk, D = 40, 2, 1, 128
    q = rng.standard_normal((L, Hq, D)).astype(F16)
    k = rng.standard_normal((L, Hk, D)).astype(F16)
    v = rng.standard_normal((L, Hk, D)).astype(F16)
    cu = np.array([0, L], np.int32)
    out = oa.varlen_attention(q, k, v, cu, cu, True, np.array([0, -1], np.int32), np.array([0, 0, 4, 8], np.int32))
    for h, (sink, local) in enumerate([(None, None), (4, 8)]):
        for i in (0, 5, 39):
            keys = [j for j in range(i + 1) if sink is None or j < sink or i - j < local]
            s = (k[keys, 0].astype(np.float64) @ q[i, h].astype(np.float64)) / np.sqrt(D)
            p = np.exp(s - s.max()); p /= p.sum()
            want = p @ v[keys, 0].astype(np.float64)
            assert np.abs(out[i, h].astype(np.float64) - want).max() < 2e-3


def test_prefill_attention_oracle_block_streaming_mask():      # block_streaming_attn_func: the Lambda rule on 128-token block indices
    qi = np.arange(400)[:, None]
    ki = np.arange(400)[None, :]
    assert np.array_equal(oa.streaming_mask(qi, ki, 4, 8, 1), (ki < 4) | ((qi - ki) < 8))       # block = 1 is the token rule
    m = oa.streaming_mask(qi, ki, 1, 2, 128) & (ki <= qi)
    # query 300 (block 2): the sink block 0 (keys 0..127), block 1 (the one local block before its own) and its own block up to itself
    assert m[300, :301].all()
    # query 399 (block 3): sink block 0, blocks 2 and 3 -- block 1 (keys 128..255) is masked
    assert m[399, :128].all() and not m[399, 128:256].any() and m[399, 256:400].all()
    # one local block = the query's own block only
    m1 = oa.streaming_mask(qi, ki, 0, 1, 128) & (ki <= qi)
    assert not m1[130, :128].any() and m1[130, 128:131].all()


# ---- the overloads off the Llama path (oracle/elementwise.py, second half) ----------------------------------------------
def test_fma32_is_a_single_rounding():       # nvcc contracts a*b+c (fused_kernels.cu:35-36, layernorm_kernels.cu:384-391)
    from fractions import Fraction
    rng = np.random.default_rng(0)
    a = rng.integers(-2 ** 22, 2 ** 22, 4000).astype(F32)
    b = rng.standard_normal(4000).astype(F16).astype(F32) * F32(2.0 ** -10)
    c = rng.standard_normal(4000).astype(F16).astype(F32)
    r = oe._fma32(a, b, c)
    two_step = ((a * b).astype(F32) + c).astype(F32)
    assert (r != two_step).any()             # the cases below do discriminate FMA from multiply-then-add
    for i in np.flatnonzero(r != two_step)[:50]:
        exact = Fraction(float(a[i])) * Fraction(float(b[i])) + Fraction(float(c[i]))
        err = abs(Fraction(float(r[i])) - exact)
        for nb in (np.nextafter(r[i], F32(np.inf)), np.nextafter(r[i], F32(-np.inf))):
            assert err <= abs(Fraction(float(nb)) - exact)      # r is the f32 nearest to the exact value


def test_static_quantiser_divides_by_the_fp16_rounded_scale():       # fused_kernels.cu:88-93 + the at::Half caster
    x = np.array([[0.25, 0.75, 1.25, -0.75, 100.0, -100.0, 0.0, 0.125]], F16)
    assert oe.quant_static(x, 0.5).tolist() == [[0, 2, 2, -2, 127, -128, 0, 0]]      # ties to even, saturation
    # 0.1 is not an fp16 number: h(0.1) = 0.0999755859375, and 3.25 / h(0.1) = 32.5079 -> 33 while 3.25 / 0.1 = 32.5 -> 32
    assert oe.quant_static(np.array([[3.25]], F16), 0.1).tolist() == [[33]]


def test_dequant_overloads_known_answers():        # fused_kernels.cu:24-55
    acc = np.array([[1024, -3, 7, 0]], np.int32)
    assert oe.dequant(acc, 0.5).tolist() == [[512.0, -1.5, 3.5, 0.0]]
    res = np.array([[1.0, 1.0, -3.5, 2.0]], F16)
    assert oe.dequant_add_residual(acc, res, 0.5).tolist() == [[513.0, -0.5, 0.0, 2.0]]
    tok = np.array([0.25], F16)
    assert oe.dequant_add_residual(acc, res, tok).tolist() == [[257.0, 0.25, -1.75, 2.0]]


def test_t5_style_fused_norm_does_not_subtract_the_mean():          # layernorm_kernels.cu:370-409
    acc = np.full((1, 32), 1000, np.int32)
    res = np.full((1, 32), 0.0234375, F16)                  # 1000 * 2^-10 + 3 * 2^-7 = 1.0 exactly
    q, new_res = oe.dequant_add_residual_rms_norm_quant(acc, res, np.full((32,), 100.0, F16), 2.0 ** -10, 1e-6)
    assert (new_res == F16(1.0)).all()
    assert (q == 100).all()             # a mean-subtracting norm would give 0 on a constant row


def test_static_norms_known_answers():              # layernorm_kernels.cu:58-196 (per tensor), :335-365 (use_quant)
    x = np.tile(np.array([1.0, 3.0], F16), 16)[None, :]
    q = oe.rms_norm_general_static(x, np.ones((32,), F16), np.array([100.0], F16), 0.0)
    # mean 2, mean(x^2) 5: y = +-1/sqrt(5) = +-0.44721 -> fp16 0.447265625 -> * 100 = 44.73 -> 45
    assert q[0].tolist() == [-45, 45] * 16
    q = oe.rms_norm_quant(x, np.full((32,), 10.0, F16), 0.0)
    assert q[0].tolist() == [4, 13] * 16        # 10/sqrt(5) = 4.47, 30/sqrt(5) = 13.42: no mean subtraction here


def test_gelu_in_half_arithmetic_known_answers():          # activation_kernels.cu:186-198
    x = np.array([0.0, 1.0, -1.0, 8.0, -8.0], F16)
    for fn in (oe.gelu_new, oe.gelu_fast):
        y = fn(x)
        assert y[0] == 0 and y[3] == F16(8.0) and y[4] == 0
        # x = 1: inner 1.044921875, u = 0.83349609375, tanh 0.68235 -> fp16 0.68212890625; 1 + t = 1.68212890625 is a
        # tie between two fp16 numbers -> even: 1.681640625; times 0.5.  x = -1: 1 - t = 0.31787109375 is exact.
        assert float(y[1]) == 0.8408203125 and float(y[2]) == -0.158935546875
    big = oe.gelu_new(np.array([65504.0], F16))         # x*x overflows to inf in fp16, tanh(inf) = 1: still x
    assert float(big[0]) == 65504.0


def test_dequant_silu_quant_known_answers():            # activation_kernels.cu:31-82
    acc = np.array([[0, 20000, -20000, 40000, 10000, 10000, 10000, -10000]], np.int32)      # gate | up, d = 4
    q, scale, tmp = oe.dequant_silu_and_mul_quant(acc, 1e-4, 1e-4)
    g, u = acc[0, :4] * 1e-4, acc[0, 4:] * 1e-4
    want = g / (1 + np.exp(-g)) * u
    assert np.allclose(tmp[0], want, rtol=1e-6) and np.isclose(scale[0], np.abs(want).max() / 127, rtol=1e-6)
    assert q[0].tolist() == [0, int(np.rint(want[1] / scale[0])), int(np.rint(want[2] / scale[0])), -127]
    assert oe.dequant_silu_and_mul_quant(acc, 1e-4, 1e-4, 0.01)[0].tolist() == [0, 127, -24, -128]


def test_bf16_oracle_paths_match_torch_bfloat16():
    """The oracle's bf16 element type (uint16 bit patterns): its rounding is torch's float32 -> bfloat16 conversion, and
    rms_norm / silu_and_mul with dtype="bf16" equal the same formulas evaluated with torch.bfloat16 roundings."""
    import torch
    from oracle import elementwise as oe
    rng = np.random.default_rng(11)
    f = np.concatenate([(rng.standard_normal(4096) * 10.0 ** rng.integers(-6, 6, 4096)).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.01171875, 3.3895314e38, np.inf, -np.inf], np.float32)])
    want = torch.from_numpy(f).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(oe._round_bf16_bits(f), want)
    x = oe._round_bf16_bits((rng.standard_normal((3, 512)) * 2).astype(np.float32))
    w = oe._round_bf16_bits(rng.standard_normal(512).astype(np.float32))
    xt = torch.from_numpy(x.view(np.int16)).view(torch.bfloat16)
    wt = torch.from_numpy(w.view(np.int16)).view(torch.bfloat16)
    # rms_norm: T(f32(T(x * rstd)) * f32(w)); rstd from the oracle's own reduction tree (pinned elsewhere)
    xf = xt.float().numpy()
    part = oe._thread_partials((xf * xf).astype(np.float32), 512, lambda a, c: (a + c).astype(np.float32), 0.0)
    rstd = (np.float32(1.0) / np.sqrt((oe.ref_tree_sum(part) / np.float32(512)).astype(np.float32) + np.float32(1e-5))).astype(np.float32)
    t = torch.from_numpy((xf * rstd[:, None]).astype(np.float32)).to(torch.bfloat16)
    ref = (t.float() * wt.float()).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(oe.rms_norm(x, w, 1e-5, dtype="bf16"), ref)
    # general norm: int8 codes must agree with an independent evaluation that rounds y to bfloat16 through torch
    q, scale, ssum = oe.rms_norm_general(x, w, 1e-5, True, dtype="bf16")
    assert q.dtype == np.int8 and scale.dtype == np.float16 and ssum.dtype == np.float16
    qf, scf, _ = oe.rms_norm_general(oe._load(x, "bf16"), oe._load(w, "bf16"), 1e-5, False, dtype="f32")
    assert np.abs(q.astype(np.int32) - qf.astype(np.int32)).max() <= 1      # bf16 amax vs f32 amax: at most one code apart
