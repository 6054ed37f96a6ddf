"""SURVEY 8 f-4 as adoptable pieces: a decoder layer written with the REFERENCE's call sequence (llama_w4a8_unpad.py:271-438,
`omniserve_backend.*` names, positional arguments) on a PersistentActivationBuffer view, run eagerly and as one
GraphedStep.  Both must leave the KV4 pages byte-identical to the vectors the reference's own LlamaDecoderLayer produced
(tests/golden/decoder_layer_w4a8kv4.npz), agree with each other bit for bit, and the replay must not allocate."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Layer:
    def __init__(self, v, dev):
        import omniserve_backend.fused_attention_fine_grained_dense as fgd
        from omniserve_amd.persistent import PersistentActivationBuffer
        self.hidden, self.inter, self.hq, self.hk, self.d, self.tpb, self.B, self.L, self.steps, self.pages = [int(t) for t in v["shape"]]
        self.base, self.eps = [float(t) for t in v["rope_base_eps"]]
        self.dev = dev
        self.w = {k: torch.from_numpy(v[k]).to(dev) for k in v if "." in k or k in ("ln1", "ln2")}
        pb = 2 * (self.hk * self.tpb * self.d // 2 + self.hk * self.tpb * 4) // 2
        self.page_bytes = pb
        n = self.B * self.pages
        self.kpool = torch.zeros((n, pb), dtype=torch.uint8, device=dev)
        self.vpool = torch.zeros((n, pb), dtype=torch.uint8, device=dev)
        ids = torch.arange(n, device=dev).view(self.B, self.pages)
        self.table = torch.stack([self.kpool.data_ptr() + ids * pb, self.vpool.data_ptr() + ids * pb], dim=1).contiguous()
        T = self.B * self.L
        self.pab = PersistentActivationBuffer(self.hidden, self.inter, self.hq * self.d, self.hk * self.d, T, 1 << 20, dev)
        self.x = torch.empty((T, self.hidden), dtype=torch.float16, device=dev)        # layer input / output (persistent)
        self.attn_q = torch.empty((T, self.hq * self.d), dtype=torch.int8, device=dev)
        self.lengths = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        self.flags = torch.ones((self.hk,), dtype=torch.int32, device=dev)
        self.rank = torch.arange(self.hk, dtype=torch.int32, device=dev)
        self.cu = torch.arange(0, self.B + 1, dtype=torch.int32, device=dev) * self.L
        self.pad = fgd.compute_padding_offsets(self.cu, self.L, T)

    def pages_of(self, pool):
        return pool.view(self.B, self.pages, self.page_bytes).cpu().numpy()

    def forward(self, T, is_prompt):
        """One decoder layer in the reference's order; everything in place on persistent tensors."""
        import omniserve_backend.activation_ops as act
        import omniserve_backend.fused_attention_fine_grained_dense as fgd
        import omniserve_backend.fused_attention_pure_dense as dense
        import omniserve_backend.fused_kernels as fk
        import omniserve_backend.layernorm_ops as ln
        import omniserve_backend.qgemm_w4a8_per_chn as gemm
        from flash_attn.flash_attn_interface import flash_attn_varlen_func
        w, b = self.w, self.pab.view_for(T)
        x = self.x[:T]
        hq, hk, d = self.hq, self.hk, self.d
        ln.rms_norm_general_fuse_sum(b.quantized_hidden_states_buffer, x, w["ln1"], b.quantized_sum_buffer,
                                     b.quantized_scale_buffer, self.eps, True)
        gemm.gemm_forward_cuda(b.quantized_hidden_states_buffer, w["qkv.qweight"], w["qkv.s1_scales"], b.quantized_scale_buffer,
                               w["qkv.s1_szeros"], b.quantized_sum_buffer, b.qkv_proj_act_buffer)
        qkv = b.qkv_proj_act_buffer
        q = qkv[:, : hq * d].view(T, hq, d)
        k = qkv[:, hq * d:(hq + hk) * d].view(T, hk, d)
        vv = qkv[:, (hq + hk) * d:].view(T, hk, d)
        if is_prompt:
            fgd.apply_bias_rope_update_kv_cache(qkv, self.lengths, None, self.pad, self.table, None, self.flags, self.rank,
                                                hq, hk, self.L, self.tpb, hk * d // 2, 0, 0, 0, 0, 0, hk, 0, d, self.base, 1.0,
                                                8192, True, True, True)
            attn = flash_attn_varlen_func(q, k, vv, self.cu, self.cu, self.L, self.L, causal=True)
        else:
            attn = dense.single_query_attention(q, k, vv, self.table, self.lengths, None, 8192, self.tpb, hk * d // 2,
                                                self.L + self.steps + 1, d, self.base, True, True, True)
        fk.invoke_quant_fuse_sum(self.attn_q[:T], attn.view(T, hq * d), b.quantized_sum_buffer, b.quantized_scale_buffer)
        gemm.gemm_forward_cuda(self.attn_q[:T], w["o.qweight"], w["o.s1_scales"], b.quantized_scale_buffer, w["o.s1_szeros"],
                               b.quantized_sum_buffer, b.out_down_proj_act_buffer)       # (overwrites the qkv buffer, as upstream)
        x.add_(b.out_down_proj_act_buffer)
        ln.rms_norm_general_fuse_sum(b.quantized_hidden_states_buffer, x, w["ln2"], b.quantized_sum_buffer,
                                     b.quantized_scale_buffer, self.eps, True)
        gemm.gemm_forward_cuda(b.quantized_hidden_states_buffer, w["gate_up.qweight"], w["gate_up.s1_scales"],
                               b.quantized_scale_buffer, w["gate_up.s1_szeros"], b.quantized_sum_buffer, b.gate_up_proj_act_buffer)
        mid = b.gate_up_proj_act_buffer.view(-1)[: T * self.inter].view(T, self.inter)   # silu(gate)*up, in place over gate
        act.silu_and_mul(mid, b.gate_up_proj_act_buffer)
        fk.invoke_quant_fuse_sum(b.quantized_mlp_act_buffer, mid, b.quantized_sum_buffer, b.quantized_scale_buffer)
        gemm.gemm_forward_cuda(b.quantized_mlp_act_buffer, w["down.qweight"], w["down.s1_scales"], b.quantized_scale_buffer,
                               w["down.s1_szeros"], b.quantized_sum_buffer, b.out_down_proj_act_buffer)
        x.add_(b.out_down_proj_act_buffer)


def _run(v, dev, graphed):
    from omniserve_amd.persistent import GraphedStep
    Ly = _Layer(v, dev)
    T = Ly.B * Ly.L
    Ly.x.copy_(torch.from_numpy(v["prefill_in"]).to(dev))
    Ly.lengths.fill_(Ly.L)
    Ly.forward(T, True)
    torch.cuda.synchronize()
    out = {"prefill_k": Ly.pages_of(Ly.kpool), "prefill_v": Ly.pages_of(Ly.vpool)}

    def step():
        Ly.lengths.add_(1)
        Ly.forward(Ly.B, False)

    gs = GraphedStep(step, dev) if graphed else None
    for s in range(Ly.steps):
        Ly.x[: Ly.B].copy_(torch.from_numpy(v["decode%d_in" % s]).to(dev))
        if graphed:
            if gs.graph is None:
                saved = (Ly.lengths.clone(), Ly.x[: Ly.B].clone(), Ly.kpool.clone(), Ly.vpool.clone())

                def restore():
                    Ly.lengths.copy_(saved[0]); Ly.x[: Ly.B].copy_(saved[1]); Ly.kpool.copy_(saved[2]); Ly.vpool.copy_(saved[3])
                gs.capture(restore)
                torch.cuda.synchronize()
                base_mem = torch.cuda.memory_allocated(dev)
            gs.run()
            torch.cuda.synchronize()
            assert torch.cuda.memory_allocated(dev) == base_mem, "the replay allocated device memory"
        else:
            step()
        torch.cuda.synchronize()
        out["x%d" % s] = Ly.x[: Ly.B].cpu().numpy().copy()
        out["k%d" % s], out["v%d" % s] = Ly.pages_of(Ly.kpool), Ly.pages_of(Ly.vpool)
    return out


def test_reference_call_sequence_on_persistent_buffers_eager_and_graphed(golden_dir):
    z = np.load(os.path.join(golden_dir, "decoder_layer_w4a8kv4.npz"))
    v = {k: z[k] for k in z.files}
    dev = torch.device("cuda:0")
    eager, graphed = _run(v, dev, False), _run(v, dev, True)
    steps = int(v["shape"][8])
    for name, r in (("eager", eager), ("graphed", graphed)):
        assert np.array_equal(r["prefill_k"], v["prefill_k_pages"]) and np.array_equal(r["prefill_v"], v["prefill_v_pages"]), name
        for s in range(steps):
            assert np.array_equal(r["k%d" % s], v["decode%d_k_pages" % s]), (name, s)
            assert np.array_equal(r["v%d" % s], v["decode%d_v_pages" % s]), (name, s)
    for s in range(steps):
        assert np.array_equal(eager["x%d" % s].view(np.uint16), graphed["x%d" % s].view(np.uint16)), "graph replay differs from eager"
        want = v["decode%d_out" % s].astype(np.float32)
        rel = np.linalg.norm(eager["x%d" % s].astype(np.float32) - want) / np.linalg.norm(want)
        assert rel < 0.08, (s, rel)        # see tests/test_reference_layer_golden_gpu.py for what this bound means
