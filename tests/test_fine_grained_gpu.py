"""Parity of the LServe fine-grained KV4 path: prefill writer with streaming heads
(bit-exact pages), decode attention over retrieval + streaming heads (dense) and over dynamically
selected pages (sparse): fp16 output within 1e-3 relative of the f64 oracle, appended rows / ring slots /
page statistics bit-exact."""
import numpy as np
import pytest
import torch

from oracle import kv4
from tests.util import GpuPagedKV, assert_attention_close, assert_f16_equal, dev, to_dev

pytestmark = pytest.mark.gpu

D = 128
ROPE_BASE = 500000.0


class Case:
    def __init__(self, seq_lens, Hq, flags, tpb, sink, local, seed, sub_chunk=0, extra_tokens=8, scale=1.0,
                 kv8_scales=None):
        """kv8_scales = (k_scale_quant_orig, v_scale_quant_orig): the per-tensor KV8 family instead of KV4."""
        self.rng = np.random.default_rng(seed)
        self.kv8 = kv8_scales
        self.seq_lens = [int(x) for x in seq_lens]
        self.B, self.Hq, self.Hk, self.tpb = len(seq_lens), Hq, len(flags), tpb
        self.flags = np.asarray(flags, np.int32)
        rank, nr, ns = [], 0, 0
        for f in flags:
            if f:
                rank.append(nr); nr += 1
            else:
                rank.append(ns); ns += 1
        self.rank = np.asarray(rank, np.int32)
        self.nr, self.ns = nr, ns
        self.sink, self.local = sink, local
        self.sink_blocks, self.local_blocks = (sink + tpb - 1) // tpb, local // tpb + 1
        self.sub_chunk, self.scale = sub_chunk, scale
        B = self.B
        rpages = (max(self.seq_lens) + extra_tokens) // tpb + 1
        spages = self.sink_blocks + self.local_blocks
        perm = lambda n, m: self.rng.permutation(n * m).reshape(n, m)
        self.rk_idx, self.rv_idx = perm(B, rpages), perm(B, rpages)
        self.sk_idx, self.sv_idx = perm(B, spages), perm(B, spages)
        if kv8_scales is None:
            mk_k = mk_v = kv4.PagedKV4
            self.row = D // 2
        else:
            from oracle import kv8
            mk_k = lambda n, h, d, t, **kw: kv8.PagedKV8(n, h, d, kv8_scales[0], t, **kw)
            mk_v = lambda n, h, d, t, **kw: kv8.PagedKV8(n, h, d, kv8_scales[1], t, **kw)
            self.row = D
            qo = np.asarray(kv8_scales, np.float32)
            self.qo_d, self.oq_d = to_dev(qo), to_dev((np.float32(1.0) / qo).astype(np.float32))
        self.rk = mk_k(B * rpages, max(nr, 1), D, tpb, fill=0x3C, stats_sub_chunk=sub_chunk)
        self.rv = mk_v(B * rpages, max(nr, 1), D, tpb, fill=0x3C)
        self.sk = mk_k(B * spages, max(ns, 1), D, tpb, fill=0x3C)
        self.sv = mk_v(B * spages, max(ns, 1), D, tpb, fill=0x3C)
        self.fg = kv4.FineGrainedKV(self.rk, self.rv, self.rk_idx, self.rv_idx, self.sk, self.sv, self.sk_idx,
                                    self.sv_idx, self.flags, self.rank, sink, local, self.sink_blocks,
                                    self.local_blocks, sub_chunk)
        self.g_retr = GpuPagedKV(self.rk, self.rv, self.rk_idx, self.rv_idx)
        self.g_strm = GpuPagedKV(self.sk, self.sv, self.sk_idx, self.sv_idx)
        self.flags_d, self.rank_d = to_dev(self.flags), to_dev(self.rank)

    def check_pools(self, what):
        rk, rv = self.g_retr.pools()
        sk, sv = self.g_strm.pools()
        assert np.array_equal(rk, self.rk.pool), "retrieval K pages differ (%s)" % what
        assert np.array_equal(rv, self.rv.pool), "retrieval V pages differ (%s)" % what
        assert np.array_equal(sk, self.sk.pool), "streaming K pages differ (%s)" % what
        assert np.array_equal(sv, self.sv.pool), "streaming V pages differ (%s)" % what

    def prefill(self):
        import omniserve_backend.fused_attention_fine_grained_dense as fa
        Hq, Hk = self.Hq, self.Hk
        T = sum(self.seq_lens)
        max_len = max(self.seq_lens)
        qkv = self.rng.standard_normal((T, (Hq + 2 * Hk) * D)).astype(np.float16)
        want = kv4.prefill_write_fine_grained(qkv, self.seq_lens, self.fg, Hq, Hk, D, ROPE_BASE, self.scale)
        cu = np.concatenate([[0], np.cumsum(self.seq_lens)]).astype(np.int32)
        pad = fa.compute_padding_offsets(to_dev(cu), max_len, T)
        qkv_d = to_dev(qkv)
        lens_d = to_dev(np.asarray(self.seq_lens, np.int32))
        if self.kv8 is None:
            fa.apply_bias_rope_update_kv_cache(
                qkv_d, lens_d, lens_d, pad, self.g_retr.table, self.g_strm.table, self.flags_d, self.rank_d, Hq, Hk,
                max_len, self.tpb, self.nr * D // 2, self.ns * D // 2, self.sink, self.local, self.sink_blocks,
                self.local_blocks, self.nr, self.ns, D, ROPE_BASE, self.scale, 1 << 20, True, True, True)
        else:
            import omniserve_backend.fused_attention_per_tensor_dense as fp
            fp.apply_bias_rope_update_kv_cache(
                qkv_d, self.oq_d, lens_d, lens_d, pad, self.g_retr.table, self.g_strm.table, self.flags_d,
                self.rank_d, Hq, Hk, max_len, self.tpb, self.nr * D, self.ns * D, self.sink, self.local,
                self.sink_blocks, self.local_blocks, self.nr, self.ns, D, ROPE_BASE, self.scale, 1 << 20, True,
                False, False)
        torch.cuda.synchronize()
        assert_f16_equal(qkv_d, want, "qkv after in-place RoPE")
        self.check_pools("prefill")
        self.qkv_post = want

    def decode(self, steps, dyn_fn=None):
        import omniserve_backend.fused_attention_fine_grained_dense as fad
        import omniserve_backend.fused_attention_fine_grained_sparse as fas
        Hq, Hk, B = self.Hq, self.Hk, self.B
        lens = np.asarray(self.seq_lens, np.int32)
        for step in range(steps):
            lens = lens + 1
            qkv = self.rng.standard_normal((B, (Hq + 2 * Hk) * D)).astype(np.float16)
            q = qkv[:, : Hq * D].reshape(B, Hq, D)
            k = qkv[:, Hq * D:(Hq + Hk) * D].reshape(B, Hk, D)
            v = qkv[:, (Hq + Hk) * D:].reshape(B, Hk, D)
            dyn = dyn_fn(lens - 1) if dyn_fn else None
            want = kv4.decode_attention_fine_grained(q, k, v, lens, self.fg, ROPE_BASE, self.scale, dyn)
            qkv_d = to_dev(qkv)
            qd = qkv_d[:, : Hq * D].view(B, Hq, D)
            kd = qkv_d[:, Hq * D:(Hq + Hk) * D].view(B, Hk, D)
            vd = qkv_d[:, (Hq + Hk) * D:].view(B, Hk, D)
            kv4_mode = self.kv8 is None
            common = (self.tpb, self.nr * self.row, self.ns * self.row, self.sink, self.local, self.sink_blocks,
                      self.local_blocks, self.nr, self.ns, int(lens.max()), D, ROPE_BASE, self.scale, True,
                      kv4_mode, kv4_mode)
            if not kv4_mode:
                import omniserve_backend.fused_attention_per_tensor_dense as fpd
                import omniserve_backend.fused_attention_per_tensor_sparse as fps
                if dyn is None:
                    out = fpd.single_query_attention(qd, kd, vd, self.qo_d, self.oq_d, self.g_retr.table,
                                                     self.g_strm.table, self.flags_d, self.rank_d, to_dev(lens),
                                                     None, 65536, *common, 2048)
                else:
                    out = fps.single_query_attention(qd, kd, vd, self.qo_d, self.oq_d, self.g_retr.table,
                                                     self.g_strm.table, self.flags_d, self.rank_d, to_dev(dyn),
                                                     to_dev(lens), None, 65536, *common, self.sub_chunk,
                                                     self.nr * D, 2048)
            elif dyn is None:
                out = fad.single_query_attention(qd, kd, vd, self.g_retr.table, self.g_strm.table, self.flags_d,
                                                 self.rank_d, to_dev(lens), None, 65536, *common, 2048)
            else:
                out = fas.single_query_attention(qd, kd, vd, self.g_retr.table, self.g_strm.table, self.flags_d,
                                                 self.rank_d, to_dev(dyn), to_dev(lens), None, 65536, *common,
                                                 self.sub_chunk, self.nr * D, 2048)
            torch.cuda.synchronize()
            got = out.cpu().numpy().astype(np.float32)
            ref = want.astype(np.float32)
            assert_attention_close(got, ref, "fine-grained decode attention, step %d" % step)
            self.check_pools("decode step %d" % step)


FLAGS_MIXED = [1, 0, 0, 1]


@pytest.mark.parametrize("seq_lens,Hq,flags,tpb,sink,local", [
    ([5, 40, 130, 200], 8, FLAGS_MIXED, 16, 16, 48),
    ([700, 64, 383, 385], 8, FLAGS_MIXED, 64, 128, 256),
    ([100, 300], 16, [0, 0], 64, 64, 128),          # streaming heads only
    ([90, 33], 4, [1, 1, 1, 1], 16, 16, 32),        # retrieval heads only, through the fine-grained entry
])
def test_prefill_write_streaming_heads(seq_lens, Hq, flags, tpb, sink, local):
    Case(seq_lens, Hq, flags, tpb, sink, local, seed=sum(seq_lens)).prefill()


def test_prefill_write_streaming_linear_rope_scaling():
    Case([70, 3, 150], 8, FLAGS_MIXED, 16, 16, 48, seed=5, scale=4.0).prefill()


@pytest.mark.parametrize("seq_lens,Hq,flags,tpb,sink,local,steps", [
    ([5, 40, 130, 200], 8, FLAGS_MIXED, 16, 16, 48, 3),
    ([62, 63, 64, 79], 8, FLAGS_MIXED, 16, 16, 48, 4),        # valid-length boundary sink+local-1 = 63
    ([700, 64, 383, 385], 16, FLAGS_MIXED, 64, 128, 256, 2),   # LServe defaults, group 4
    ([100, 300], 16, [0, 0], 64, 64, 128, 2),
    ([1500, 1030], 32, [1, 0, 0, 0, 1, 0, 0, 1], 64, 128, 256, 1),
])
def test_decode_fine_grained_dense(seq_lens, Hq, flags, tpb, sink, local, steps):
    c = Case(seq_lens, Hq, flags, tpb, sink, local, seed=sum(seq_lens) + Hq)
    c.prefill()
    c.decode(steps)


def test_decode_fine_grained_dense_linear_rope_scaling():
    c = Case([70, 150], 8, FLAGS_MIXED, 16, 16, 48, seed=9, scale=2.0)
    c.prefill()
    c.decode(2)


@pytest.mark.parametrize("newest_page", ["of_last_cached_token", "of_current_token"])
@pytest.mark.parametrize("seq_lens,Hq,flags,tpb,sink,local,P,sub,steps", [
    ([130, 200, 97], 8, FLAGS_MIXED, 16, 16, 48, 4, 8, 3),
    ([700, 640], 16, FLAGS_MIXED, 64, 128, 256, 6, 16, 2),
    ([255, 256], 8, [1, 1], 64, 128, 256, 3, 32, 3),            # page boundary: history ends a page exactly
    ([63, 64, 127], 8, FLAGS_MIXED, 16, 16, 48, 3, 8, 4),       # several sequences cross a 16-token page
])
def test_decode_fine_grained_sparse(seq_lens, Hq, flags, tpb, sink, local, P, sub, steps, newest_page):
    """newest_page = "of_current_token" is the reference's own page choice (decoding_attention.py:132-142: the
    last entry is timestep // tokens_per_block, the page the current token is appended to): on the step where the
    history fills a page exactly that page holds no cached token yet and must contribute none."""
    c = Case(seq_lens, Hq, flags, tpb, sink, local, seed=sum(seq_lens) + P, sub_chunk=sub)
    c.prefill()
    # statistics of the prompt (oracle side), mirrored to the GPU pool before decoding
    cu = np.concatenate([[0], np.cumsum(c.seq_lens)]).astype(np.int32)
    k_post = c.qkv_post[:, c.Hq * D:(c.Hq + c.Hk) * D].reshape(-1, c.Hk, D)
    heads = [h for h in range(c.Hk) if c.flags[h]]
    kv4.paged_min_max_pool(k_post, cu, heads, c.rk.pool, c.rk_idx, tpb, sub)
    c.g_retr.kpool.copy_(to_dev(c.rk.pool))

    def dyn_fn(hist):
        dyn = np.zeros((c.B, c.Hq, P), np.int32)
        for b in range(c.B):
            last = int(hist[b]) // tpb if newest_page == "of_current_token" else (int(hist[b]) - 1) // tpb
            for h in range(c.Hq):
                pick = c.rng.choice(last, size=P - 1, replace=False) if last >= P - 1 else np.arange(P - 1) % max(last, 1)
                dyn[b, h, : P - 1] = np.sort(pick)
                dyn[b, h, P - 1] = last
        return dyn

    c.decode(steps, dyn_fn)
