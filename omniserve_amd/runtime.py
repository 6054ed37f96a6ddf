"""Minimal decode-step driver for the QServe W4A8KV4 Llama path on one MI355X.

Not an engine: it only wires the kernels in the order of the reference's decoder layer
(omniserve/modeling/models/llama_w4a8_unpad.py:406-438) on synthetic weights so
that bench.py / smoke() can time the hot path named by BASELINE.json:

    rms_norm_general_fuse_sum -> qkv GEMM -> single_query_attention (RoPE + KV4 append fused)
    -> invoke_quant_fuse_sum -> o_proj GEMM -> residual add -> rms_norm_general_fuse_sum
    -> gate_up GEMM -> silu_and_mul -> invoke_quant_fuse_sum -> down GEMM -> residual add
    ... x num_layers, then rms_norm -> fp16 lm_head -> argmax  (llama_w4a8_unpad.py:485-488,556-563)

All calls go through the mirrored `omniserve_backend.*` modules, i.e. through the C ABI.
The whole step is captured in one HIP graph (the reference launches ~322 kernels per step
eagerly; at MI355X speeds the step is launch-bound otherwise).
"""
from __future__ import annotations

import dataclasses

import torch

from .backend import (activation_ops, fused_attention_pure_dense, fused_ext, fused_kernels, layernorm_ops,
                      qgemm_w4a8_per_chn, qgemm_w4a8_per_group)
from .rope import rope_table
from . import _lib


@dataclasses.dataclass
class LlamaConfig:
    hidden: int = 4096
    inter: int = 14336
    heads: int = 32
    kv_heads: int = 8
    head_dim: int = 128
    layers: int = 32
    vocab: int = 128256
    rope_theta: float = 500000.0
    eps: float = 1e-5
    group_size: int = -1          # -1 = per-channel, 128 = g128

    @staticmethod
    def llama3_8b(group_size=-1):
        return LlamaConfig(group_size=group_size)

    @staticmethod
    def llama2_70b(group_size=-1):
        return LlamaConfig(hidden=8192, inter=28672, heads=64, kv_heads=8, layers=80, vocab=32000, rope_theta=10000.0,
                           group_size=group_size)

    @staticmethod
    def tiny():
        return LlamaConfig(hidden=512, inter=1024, heads=4, kv_heads=2, layers=2, vocab=512)


class W4A8Linear:
    """Synthetic packed weights in the reference layout (w4a8_linear.py:43-100).  Any byte pattern
    is a valid packing of uniform random 4-bit codes, so the weights are drawn on the device."""

    def __init__(self, n, k, group_size, gen, device):
        self.n, self.k, self.group = n, k, group_size
        self.qweight = torch.randint(0, 256, (n, k // 2), dtype=torch.uint8, device=device, generator=gen).view(torch.int8)
        self.s1_scales = (torch.rand((n,), device=device, generator=gen) * 0.018 + 0.002).half()
        if group_size != -1:
            # level-2 weights span +-120 instead of +-15: keep the synthetic model's activations in fp16 range
            self.s1_scales = (self.s1_scales.float() / 8.0).half()
        if group_size == -1:
            zeros = torch.randint(0, 16, (n,), device=device, generator=gen).half()
            self.s1_szeros = (zeros * self.s1_scales).half()
        else:
            ng = k // group_size
            s2 = torch.randint(1, 9, (ng, n), device=device, generator=gen)
            z = torch.randint(0, 16, (ng, n), device=device, generator=gen)
            self.s2_scales = s2.to(torch.int8)
            self.s2_zeros = (-(z * s2)).to(torch.int8)

    def forward(self, x_i8, scales, sums, out):
        if self.group == -1:
            qgemm_w4a8_per_chn.gemm_forward_cuda(x_i8, self.qweight, self.s1_scales, scales, self.s1_szeros, sums, out)
        else:
            qgemm_w4a8_per_group.gemm_forward_cuda(x_i8, self.qweight, self.s2_zeros, self.s2_scales,
                                                   self.s1_scales, scales, out)

    def weight_bytes(self):
        b = self.qweight.numel()
        if self.group != -1:
            b += self.s2_scales.numel() + self.s2_zeros.numel()
        return b

    # ---- tensor-parallel shards of a full layer (omniserve_amd/tp.py has the granularity rules) ------------
    def select_rows(self, ranges):
        """Column-parallel shard: output channels [a,b) for (a,b) in ranges (multiples of 64), concatenated."""
        sh = object.__new__(W4A8Linear)
        cat = lambda t, dim: torch.cat([t.narrow(dim, a, b - a) for a, b in ranges], dim=dim).contiguous()
        sh.n, sh.k, sh.group = sum(b - a for a, b in ranges), self.k, self.group
        sh.qweight = cat(self.qweight, 0)
        sh.s1_scales = cat(self.s1_scales, 0)
        if self.group == -1:
            sh.s1_szeros = cat(self.s1_szeros, 0)
        else:
            sh.s2_scales, sh.s2_zeros = cat(self.s2_scales, 1), cat(self.s2_zeros, 1)
        return sh

    def shard_k(self, rank, world):
        """Row-parallel shard: the rank's slice of the reduction dimension (whole 128-k blocks)."""
        from . import tp
        sh = object.__new__(W4A8Linear)
        sh.n, sh.k, sh.group = self.n, self.k // world, self.group
        sh.qweight = tp.shard_qweight_k(self.qweight, rank, world)
        sh.s1_scales = self.s1_scales
        if self.group == -1:
            sh.s1_szeros = self.s1_szeros
        else:
            sh.s2_scales = tp.shard_group_params_k(self.s2_scales, rank, world, self.group)
            sh.s2_zeros = tp.shard_group_params_k(self.s2_zeros, rank, world, self.group)
        return sh


class DecodeRunner:
    """bs sequences with `context` cached tokens each; step() decodes one token per sequence."""

    def __init__(self, cfg: LlamaConfig, batch: int, context: int, max_new: int, device, seed=0,
                 use_graph=True, fused=True, tp_rank=0, tp_size=1, tp_group=None, shard_full=False,
                 prefetch_mb=None, prefetch_blocks=160, weight_policy=1, tp_comm=None, max_fused=4, tp_l2_attn=True,
                 l3_prefetch_down=1, qkv_slabs="auto", attn_single=True, arm_qkv=True, arm_o=True, l3_last=True):
        """tp_size > 1: Megatron-style tensor parallelism (omniserve_amd/tp.py): qkv / gate_up column-parallel,
        o / down row-parallel, attention by kv head, one fp16 sum all-reduce of the [B, hidden] projection after
        o_proj and after down_proj.  shard_full=True builds the full layers from the seed and keeps this
        rank's shards (tests: every rank then holds shards of the SAME model); otherwise only the local shapes
        are drawn (bench: synthetic weights, no point in materialising 70B parameters per rank).
        The A/B switches of the decode step are constructor arguments (this module reads no environment): max_fused caps the
        fusion level the runner may pick; tp_l2_attn / qkv_slabs ("auto", True, False) / attn_single / arm_qkv / arm_o /
        l3_prefetch_down (0, 1, 2) / l3_last select the forms described where they are used below."""
        self.cfg, self.B, self.device = cfg, batch, device
        self.tp_rank, self.tp_size, self.tp_group = int(tp_rank), int(tp_size), tp_group
        # tp_comm = "peer": the decode step's collectives run on the library's own peer-mapped all-reduce (tp.PeerComm: one
        # launch per collective, folded into the add + norm kernel where one follows) instead of torch.distributed / RCCL
        self.comm = None
        tp_comm = tp_comm or ""
        # ("peer": one / two shots by payload, csrc/tp_comm.h; "peer1" / "peer2": forced; "loopback[2]": one process, all peers =
        #  this rank -- timing only)
        if self.tp_size > 1 and tp_comm in ("peer", "peer1", "peer2", "loopback", "loopback2"):
            from . import tp
            algo = {"peer1": "one_shot", "peer2": "two_shot", "loopback2": "two_shot"}.get(tp_comm, "auto")
            self.comm = tp.PeerComm(self.tp_rank, self.tp_size, batch * cfg.hidden, device, tp_group,
                                    loopback=tp_comm.startswith("loopback"), algo=algo)
        if cfg.heads % self.tp_size or cfg.kv_heads % self.tp_size or cfg.inter % (128 * self.tp_size):
            raise ValueError("heads / kv_heads / intermediate size not divisible by the TP degree")
        self.hl, self.kl, self.il = cfg.heads // self.tp_size, cfg.kv_heads // self.tp_size, cfg.inter // self.tp_size
        # fused: 0/False = the reference call sequence; 1 = opt-in fused entry points (residual add +
        # norm + quant, silu*mul + quant); 2/True = additionally defer the split-K epilogue of o_proj /
        # down_proj into the following add+norm kernel.  All bit-identical to the reference sequence
        # (SURVEY.md 8f.1).
        # 3 = additionally NO quantiser row kernels: the gate_up GEMV
        # applies silu_and_mul in its epilogue and leaves the row maxima, the attention merge is a wide kernel that
        # leaves fp16 + row maxima, and o_proj / down_proj quantise their input on the fly (fused_ext.gemm_silu_*,
        # decode_attention_f16_amax, gemm_partial_f16_*): 7 kernels per layer instead of 9, same bits.
        # 3 is the default (fused=True) where it applies: batch <= 16, one GPU.
        # (A level 4 -- the MLP half of a layer as ONE persistent launch with in-kernel hand-offs -- was built in round 4, measured
        #  slower than level 3 (2.37 vs 2.17 ms per step) and moved out of the product: tools/experiments/mlp_fused.hip, HISTORY.md.)
        # 4 (round 6) = 3 with the two 1 -> N edges of the layer as single launches: (add + norm + quant) -> qkv and
        # (add + norm + quant) -> gate_up + SiLU (fused_ext.norm_gemm_fused, csrc/norm_gemv_fused.h: the rows and the GEMV's tiles in
        # one grid, the tiles request their whole weight part and then wait for the rows): 5 launches per layer, same bits.
        # Opt-in (fused=4): measured SLOWER than level 3 on the MI355X (2.24 vs 2.11 ms per step; profiles/r06_a: the hand-off --
        # write-through publish + a poll behind the tile CUs' own weight requests + the activation round trip -- costs what the
        # kernel boundary costs, and the rows run 1.5-3 us slower beside the stream), so fused=True still means level 3.
        want_pairs = fused is not True and int(fused) >= 4
        self.fused = 3 if fused is True else min(int(fused), 3)
        # the attention-side fusions of level 2 (split merge inside the quantiser, q / k / v from the qkv projection's slabs)
        # involve no row-parallel projection, so they also apply under tensor parallelism, where the level drops to 1
        self.l2_attn = self.fused >= 2 and batch <= 128 and bool(tp_l2_attn)
        if (self.tp_size > 1 or batch > 512) and self.fused > 1:
            # tensor parallel: the all-reduce needs the fp16 projection; batch > 512: the prefill tile's slab-only form
            # (omni_*_gemm_partial) stops there -- no deferred epilogue in either case.  (batch 129 .. 512: o_proj / down_proj
            # run on the 128 x 256 tile with K slices over grid.y and leave slabs for the norm, as at smaller batches)
            self.fused = 1
        # level 3 wants the plans its entry points accept (omni_gemm_rowfree_ok: the launchers' own conditions -- gate_up
        # without a grid-level K split, i.e. hidden <= 4096; one rider workgroup per row in a grid row of hidden / 64; the
        # rider's LDS copy of a row) and the wide attention merge's 4 heads per wave; otherwise level 2, which has no such
        # limits (a hidden = 5120 layer used to pass the old size test here and fail in its first step)
        if self.fused >= 3 and not (self.hl % 4 == 0 and int(max_fused) >= 3 and
                                    _lib.lib().omni_gemm_rowfree_ok(batch, cfg.hidden, self.hl * cfg.head_dim, self.il,
                                                                    0 if cfg.group_size == -1 else 1) == 1):
            self.fused = 2
        # L2 weight prefetch riding on the row kernels (fused extension; a hint, results are unaffected): MiB of the
        # next GEMV's weights each row kernel pulls into the L2s (0 = off), with how many extra workgroups, and
        # whether the GEMV on a prefetched tensor then uses plain instead of non-temporal weight loads (per call: the arm
        # names the tensor, omni_prefetch_arm_gemm; un-armed projections always stream non-temporally).  Defaults: on with the fused
        # entry points (prefetch_mb / prefetch_blocks / weight_policy arguments for sweeps).
        # level 4 where level 3 applies and both pair forms take the shapes (M <= 16, hidden <= 4096, grids resident at once)
        self.pairs = (want_pairs and self.fused >= 3 and int(max_fused) >= 4 and
                      fused_ext.norm_gemm_fused_ok(batch, (self.hl + 2 * self.kl) * cfg.head_dim, cfg.hidden, cfg.group_size, False) and
                      fused_ext.norm_gemm_fused_ok(batch, 2 * self.il, cfg.hidden, cfg.group_size, True))
        if prefetch_mb is None:
            # measured (profiles/r02_*): +6-7 % at bs = 16 with 28-40 MiB per row kernel (the L2s hold 32 MiB; the
            # excess lands in MALL), nothing at bs = 128, -4 % at bs = 64 where the row kernels are no longer idle
            # (batch 33..64: 12 MiB per carrier measured best at configs[2] -- 3.50 -> 3.43 ms per step, 24 MiB and more lose.
            #  A tensor-parallel shard's projections are small enough to be fetched whole: one Llama-2-70B TP = 8 rank at
            #  bs = 128 7.22-7.26 -> 6.59-6.65 ms per step with 12 .. 48 MiB.  Other batches > 64: not measured, off)
            dflt = 40.0 if batch <= 32 else (12.0 if batch <= 64 else (24.0 if self.tp_size > 1 else 0.0))
            prefetch_mb = dflt if self.fused else 0.0
        self.prefetch_bytes = int(float(prefetch_mb) * (1 << 20)) if self.fused else 0
        # (fetching workgroups per carrier: 240 was round 2's optimum; with down_proj's 29.6 MB on the norm in front of gate_up
        #  128-192 measure 1.7 % faster per step at bs = 16 -- 2.28 -> 2.24 ms --, 64-96 slower; bs = 64 / TP / LServe: flat)
        self.prefetch_blocks = int(prefetch_blocks)
        self.weight_policy = int(weight_policy) if self.prefetch_bytes > 0 else 0
        c = cfg
        gen = torch.Generator(device=device)
        gen.manual_seed(seed)
        self.gen = gen
        d = c.head_dim
        hl, kl, il, r, w = self.hl, self.kl, self.il, self.tp_rank, self.tp_size
        qkv_n = (hl + 2 * kl) * d
        self.layers = []
        for _ in range(c.layers):
            ln1 = (1.0 + 0.05 * torch.randn(c.hidden, device=device, generator=gen)).half()
            ln2 = (1.0 + 0.05 * torch.randn(c.hidden, device=device, generator=gen)).half()
            if w > 1 and shard_full:
                qkv = W4A8Linear((c.heads + 2 * c.kv_heads) * d, c.hidden, c.group_size, gen, device).select_rows(
                    [(r * hl * d, (r + 1) * hl * d), (c.heads * d + r * kl * d, c.heads * d + (r + 1) * kl * d),
                     ((c.heads + c.kv_heads) * d + r * kl * d, (c.heads + c.kv_heads) * d + (r + 1) * kl * d)])
                o = W4A8Linear(c.hidden, c.heads * d, c.group_size, gen, device).shard_k(r, w)
                gate_up = W4A8Linear(2 * c.inter, c.hidden, c.group_size, gen, device).select_rows(
                    [(r * il, (r + 1) * il), (c.inter + r * il, c.inter + (r + 1) * il)])
                down = W4A8Linear(c.hidden, c.inter, c.group_size, gen, device).shard_k(r, w)
            else:
                qkv = W4A8Linear(qkv_n, c.hidden, c.group_size, gen, device)
                o = W4A8Linear(c.hidden, hl * d, c.group_size, gen, device)
                gate_up = W4A8Linear(2 * il, c.hidden, c.group_size, gen, device)
                down = W4A8Linear(c.hidden, il, c.group_size, gen, device)
            self.layers.append(dict(ln1=ln1, ln2=ln2, qkv=qkv, o=o, gate_up=gate_up, down=down))
        self.final_norm = torch.ones(c.hidden, device=device).half()
        self.embed = (0.02 * torch.randn(c.vocab, c.hidden, device=device, generator=gen)).half()
        self.lm_head = (0.02 * torch.randn(c.vocab, c.hidden, device=device, generator=gen)).half()

        # ---- paged KV4 cache: one K pool and one V pool per layer (cache_engine.py:117-136) ----
        self.tpb = 64
        self.max_context = context + max_new + 1
        pages_per_seq = (self.max_context + self.tpb - 1) // self.tpb
        self.page_bytes = kl * self.tpb * (c.head_dim // 2) + 2 * kl * self.tpb * 2
        n_pages = batch * pages_per_seq
        self.block_tables = []
        self.pools = []
        data_bytes = kl * self.tpb * (c.head_dim // 2)
        for _ in range(c.layers):
            pools = []
            tab = torch.empty((batch, 2, pages_per_seq), dtype=torch.int64, device=device)
            for kv in range(2):
                pool = torch.empty((n_pages, self.page_bytes), dtype=torch.uint8, device=device)
                pool[:, :data_bytes] = torch.randint(0, 256, (n_pages, data_bytes), dtype=torch.uint8,
                                                     device=device, generator=gen)
                tail = pool[:, data_bytes:].view(torch.float16).view(n_pages, 2, kl * self.tpb)
                tail[:, 0] = 0.25 * (0.5 + torch.rand((n_pages, kl * self.tpb), device=device, generator=gen))
                tail[:, 1] = 7.5
                perm = torch.randperm(n_pages, device=device, generator=gen).view(batch, pages_per_seq)
                tab[:, kv] = pool.data_ptr() + perm * self.page_bytes
                pools.append(pool)
            self.pools.append(pools)
            self.block_tables.append(tab)

        # ---- persistent activation buffers (the reference re-allocates them every step) ----------
        B = batch
        f16, i8 = torch.float16, torch.int8
        self.x = torch.empty((B, c.hidden), dtype=f16, device=device)
        self._q_hidden = torch.empty((B, c.hidden), dtype=i8, device=device)
        self._q_inter = torch.empty((B, il), dtype=i8, device=device)
        self._q_attn = torch.empty((B, hl * c.head_dim), dtype=i8, device=device)
        self.act_scale = torch.empty((B,), dtype=f16, device=device)   # written by norm kernels
        self.act_sum = torch.empty((B,), dtype=f16, device=device)
        self.act_scale2 = torch.empty((B,), dtype=f16, device=device)  # written by quant kernels
        self.act_sum2 = torch.empty((B,), dtype=f16, device=device)
        pws = _lib.lib().omni_gemm_partial_workspace_bytes     # (slab-only forms: at least one M x N slab up to 512 rows)
        need = max(int(pws(B, c.hidden, k)) for k in (hl * c.head_dim, il))
        need = max(need, int(pws(B, qkv_n, c.hidden)))     # (qkv slabs, see qkv_slabs)
        self.slab = torch.empty((max(need, 1 << 20),), dtype=torch.uint8, device=device)  # deferred split-K partial sums
        self.qkv_buf = torch.empty((B, qkv_n), dtype=f16, device=device)
        self.proj_buf = torch.empty((B, c.hidden), dtype=f16, device=device)
        self.gate_up_buf = torch.empty((B, 2 * il), dtype=f16, device=device)
        self.mlp_act = torch.empty((B, il), dtype=f16, device=device)
        self.attn_f16 = torch.empty((B, hl * c.head_dim), dtype=f16, device=device)
        # row-maximum candidates of the level-3 path: [layer][0 = attention output, 1 = MLP activation][AMAX_WORDS], zeroed once
        # per step (the producers raise them with atomicMax)
        # (one buffer with the level-4 hand-off words behind them: [layer][0 = norm -> qkv, 1 = norm -> gate_up][NGF_SYNC_WORDS]
        #  arrival counters + row pairs -- the step's first kernel zeroes all of it)
        n_amax = c.layers * 2 * fused_ext.AMAX_WORDS
        self._step_words = torch.zeros((n_amax + c.layers * 2 * fused_ext.NGF_SYNC_WORDS,), dtype=torch.int32, device=device)
        self.amax = self._step_words[:n_amax].view(c.layers, 2, fused_ext.AMAX_WORDS)
        self.ngf_sync = self._step_words[n_amax:].view(c.layers, 2, fused_ext.NGF_SYNC_WORDS)
        self.ngf_err = torch.zeros((4,), dtype=torch.int32, device=device)      # sticky: a hand-off wait gave up (check())
        self.tickets = fused_ext.new_tickets(device)      # ticket words of the single-launch attention (this runner's stream)
        self.ngf_clk = {}     # timeline probe (tools/pairs_ab.py): {(layer, site): int64 [grid, 8]} handed to that launch
        # level 3: the norm in front of gate_up prefetches ALL of DOWN's weights (29.6 MB fit the 32 MB of L2s) instead of
        # the head of gate_up's 58.7 MB; gate_up then streams cold with non-temporal loads (they do not displace the
        # prefetched lines) and down reads L2: 2.322-2.334 -> 2.294-2.314 ms per step on the same box (profiles/r03_g;
        # l3_prefetch_down=0: round 3's first arrangement, 2: gate_up with plain loads -- slower, 2.35)
        self.pf_down = int(l3_prefetch_down)
        # fused level >= 2: the qkv projection leaves int32 split-K slabs and the decode attention applies its epilogue in
        # its first load trip (fused_ext.decode_arm_qkv_slabs): no slab epilogue launch between the two.  "auto": on exactly
        # where the plain qkv GEMV's plan splits K (bs = 64: (96, 2) workgroups + a 4.9-us epilogue launch per layer; 3.51 ->
        # 3.47 ms per step); at bs = 16 the plan has no split and the slab form measured 1.5 % slower.  qkv_slabs=False / True
        # forces it off / on (A/B)
        if qkv_slabs == "auto":      # on exactly where the plain qkv GEMV would split K and launch a slab epilogue
            import ctypes
            sk_q = ctypes.c_int(1)
            _lib.lib().omni_gemm_get_plan(batch, qkv_n, c.hidden, 128 if c.group_size == 128 else 64, None, None,
                                          ctypes.byref(sk_q))
            want = sk_q.value > 1
        else:
            want = bool(qkv_slabs)
        self.qkv_slabs = (self.fused >= 2 or self.l2_attn) and want
        # (A/B knobs: qkv's / o_proj's weights prefetched by the norm in front of qkv / the kernel behind the attention)
        # level 3: the attention's split merge inside the attention launch (last-arriving workgroup; attn_single=False: the
        # two-launch form, A/B)
        self.attn_single = bool(attn_single)
        self.arm_qkv = bool(arm_qkv)
        self.arm_o = bool(arm_o)
        # level 3 also in the LAST layer: its down projection's slabs are consumed by the model's final norm
        # (fused_ext.splitk_add_rms_norm) instead of GEMV epilogue + residual add + rms_norm (l3_last=False: off, A/B)
        self.last_l3 = self.fused >= 3 and self.tp_size == 1 and self.comm is None and bool(l3_last)
        self.normed = torch.empty((B, c.hidden), dtype=f16, device=device)
        self.lengths = torch.full((B,), context, dtype=torch.int32, device=device)
        self.tokens = torch.randint(0, c.vocab, (B,), device=device, generator=gen)
        rope_table(self.max_context + 1, c.head_dim, c.rope_theta, 1.0, device)  # pre-build (capture safe)
        self.graph = None
        self.use_graph = use_graph
        self.graph_error = None       # set when a tensor-parallel graph capture failed and the runner fell back to eager
        self.steps_done = 0

    def step(self):
        """Decode one token for every sequence (one HIP-graph replay once captured)."""
        if not self.use_graph:
            self._eager_step()
            self.steps_done += 1
            return
        if self.graph is None:
            # warm up eagerly on a side stream (sizes the workspaces / RoPE table), roll the
            # sequence state back, then capture the identical step
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                saved = (self.lengths.clone(), self.tokens.clone())
                self._eager_step()
                self.lengths.copy_(saved[0])
                self.tokens.copy_(saved[1])
            torch.cuda.current_stream().wait_stream(s)
            try:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    self._eager_step()
                self.graph = graph
            except RuntimeError as exc:  # (e.g. a collective that cannot be captured on this stack)
                if self.tp_size == 1:
                    raise
                import warnings
                warnings.warn("DecodeRunner: HIP-graph capture of the tensor-parallel step failed (%s); falling back to "
                              "eager launches" % exc)
                # tensor-parallel first contact: an RCCL all-reduce inside a captured HIP graph has never run on
                # hardware here; fall back to eager launches instead of losing the run, and say so
                torch.cuda.synchronize()
                self.use_graph = False
                self.graph_error = "%s: %s" % (type(exc).__name__, exc)
                if self.comm is not None:
                    self.comm.resync()      # the abandoned capture may have consumed an odd number of slots
                self.lengths.copy_(saved[0])
                self.tokens.copy_(saved[1])
                self._eager_step()
                self.steps_done += 1
                return
        self.graph.replay()
        self.steps_done += 1

    def check(self):
        """Raises if a peer wait of the last step timed out (tensor parallel, tp_comm = "peer").  Synchronises."""
        if self.comm is not None:
            self.comm.check_error()
        if self.pairs and int(self.ngf_err[0].item()) != 0:
            raise RuntimeError("DecodeRunner: a (norm -> GEMV) hand-off wait gave up (rows never arrived); the step's outputs "
                               "are NaN")

    def read_tokens(self):
        """The tokens of the last step on the host (synchronises).  Under tensor parallelism with the library's own
        collective this is also where a timed-out peer wait surfaces: the affected collectives returned NaN, the epoch did
        not advance, and this raises instead of handing back tokens computed from them."""
        toks = self.tokens.cpu()
        self.check()
        return toks

    def close(self):
        """Release the peer-mapped communication buffers (tensor parallel, tp_comm = "peer")."""
        if self.comm is not None:
            self.comm.close()
            self.comm = None

    def _partial(self, x_i8, lin):
        """o_proj / down_proj without its epilogue: int32 split-K slabs in self.slab, returns the slab count."""
        if lin.group == -1:
            return fused_ext.gemm_partial_per_chn(x_i8, lin.qweight, self.slab)
        return fused_ext.gemm_partial_per_group(x_i8, lin.qweight, lin.s2_zeros, lin.s2_scales, self.slab)

    def _partial_f16(self, act, amax, lin, sum_out, scale_out):
        """Level 3: o_proj / down_proj from fp16 activations + row maxima (codes computed inside the GEMV; sums / scales by
        rider workgroups): int32 split-K slabs in self.slab, returns the slab count."""
        if lin.group == -1:
            return fused_ext.gemm_partial_f16_per_chn(act, amax, lin.qweight, self.slab, sum_out, scale_out)
        return fused_ext.gemm_partial_f16_per_group(act, amax, lin.qweight, lin.s2_zeros, lin.s2_scales, self.slab,
                                                    sum_out, scale_out)

    def _consume(self, q_out, sk, lin, a_scale, a_sum, gamma, out_sum, out_scale):
        """residual += epilogue(sum of the slabs); norm + quant of it (the GEMM epilogue lives in this row kernel)."""
        if lin.group == -1:
            fused_ext.splitk_add_rms_norm_general_fuse_sum(q_out, self.x, self.slab, sk, lin.s1_scales, a_scale,
                                                           lin.s1_szeros, a_sum, gamma, out_sum, out_scale, self.cfg.eps)
        else:
            fused_ext.splitk_w8_add_rms_norm_general_fuse_sum(q_out, self.x, self.slab, sk, lin.s1_scales, a_scale, gamma,
                                                              out_sum, out_scale, self.cfg.eps)

    def _arm(self, lin, deferred=False, silu=False):
        """The next row kernel prefetches the head of `lin`'s weight stream into the L2s (no-op when disabled).
        silu: `lin` will run as fused_ext.gemm_silu_* (gate / up tile rows paired per workgroup)."""
        if self.prefetch_bytes > 0:      # (weight_policy 0: the consuming GEMV keeps its non-temporal loads -- A/B)
            fused_ext.prefetch_arm_gemm(lin.qweight, self.B, lin.n, lin.k,
                                        (0 if lin.group == -1 else 1) | (0x10 if silu else 0) | (0 if self.weight_policy else 0x20),
                                        deferred, self.prefetch_bytes, self.prefetch_blocks)

    def _eager_step(self):
        try:
            self._eager_step_body()
        finally:
            if self.prefetch_bytes > 0:
                fused_ext.prefetch_disarm()     # a step that raised may leave a descriptor armed
            if self.qkv_slabs:
                fused_ext.decode_arm_qkv_slabs(None, 0, 0, 0, 0, 0, 0, None, None)     # ... and the q / k / v slab source

    def _eager_step_body(self):
        # one decoder layer at decode shape = llama_w4a8_unpad.py:406-438
        c = self.cfg
        if self.fused:     # one launch: embedding rows (torch's index_select takes 12.6 us for 16 rows) + lengths += 1 + the
            fused_ext.decode_step_begin(self.x, self.embed, self.tokens, self.lengths,      # step's row-maximum slots zeroed
                                        self._step_words if self.fused >= 3 else None)
        else:
            self.lengths.add_(1)
            torch.index_select(self.embed, 0, self.tokens, out=self.x)
        B = self.B
        hq, hk, d = self.hl, self.kl, c.head_dim     # this rank's heads
        sA, mA = self.act_scale2, self.act_sum2   # scales / sums produced by quant-type kernels
        sB, mB = self.act_scale, self.act_sum     # ... by norm kernels
        per_chn = c.group_size == -1
        if not per_chn:    # g128: no activation sums anywhere (llama_w4a8_unpad.py:81,212-215,395-401): the reference calls
            mA = mB = None # rms_norm_general / invoke_quant, and the fused entry points skip the second row reduction
        pending = None                             # (sk, linear) of a down_proj whose epilogue is deferred
        nl = len(self.layers)
        for li, L in enumerate(self.layers):
            qa_h, qa_i = self._q_hidden, self._q_inter
            l3 = self.fused >= 3 and (li < nl - 1 or self.last_l3)     # row-kernel-free MLP half in this layer
            if self.pairs and l3:
                # level 4: (add + norm + quant) -> qkv as one launch (rows from the previous layer's down_proj slabs; first
                # layer: the embedding rows as they are)
                if pending is not None:
                    sk, lin = pending
                    fused_ext.norm_gemm_fused(qa_h, self.x, L["ln1"], mB, sB, c.eps, L["qkv"], self.qkv_buf, self.ngf_sync[li, 0],
                                              self.ngf_err, slab=self.slab, sk=sk, producer=lin, p_ascales=sA, p_asums=mA,
                                              clk=self.ngf_clk.get((li, 0)))
                    pending = None
                else:
                    fused_ext.norm_gemm_fused(qa_h, self.x, L["ln1"], mB, sB, c.eps, L["qkv"], self.qkv_buf, self.ngf_sync[li, 0],
                                              self.ngf_err)
            elif self.arm_qkv:
                self._arm(L["qkv"], deferred=self.qkv_slabs)
            if self.pairs and l3:
                pass
            elif pending is not None:     # residual += down_proj(prev layer) [deferred epilogue], norm + quant
                sk, lin = pending
                self._consume(qa_h, sk, lin, sA, mA, L["ln1"], mB, sB)
                pending = None
            elif self.fused and li > 0 and self.comm is not None:
                self.comm.add_rms_norm(qa_h, self.x, L["ln1"], mB, sB, c.eps)      # all-reduce(down_proj) + add + norm + quant
            elif self.fused and li > 0:
                fused_ext.add_rms_norm_general_fuse_sum(qa_h, self.x, self.proj_buf, L["ln1"], mB, sB, c.eps)
            elif per_chn:
                layernorm_ops.rms_norm_general_fuse_sum(qa_h, self.x, L["ln1"], mB, sB, c.eps, True)
            else:
                layernorm_ops.rms_norm_general(qa_h, self.x, L["ln1"], sB, c.eps, True)
            if self.pairs and l3:
                pass
            elif self.qkv_slabs:      # slabs only; the attention kernel reads q / k / v from them (qkv_buf: shapes only)
                lin = L["qkv"]
                fused_ext.decode_arm_qkv_slabs(self.slab, self._partial(qa_h, lin), B, lin.n, 0, hq * d, (hq + hk) * d,
                                               lin.s1_scales, sB, lin.s1_szeros if per_chn else None,
                                               mB if per_chn else None)
            else:
                L["qkv"].forward(qa_h, sB, mB, self.qkv_buf)
            q = self.qkv_buf[:, : hq * d].view(B, hq, d)
            k = self.qkv_buf[:, hq * d:(hq + hk) * d].view(B, hk, d)
            v = self.qkv_buf[:, (hq + hk) * d:].view(B, hk, d)
            if self.arm_o:
                self._arm(L["o"], deferred=self.fused >= 2)      # rides on the quantiser after the attention
            if self.fused >= 3:     # merge as a wide kernel (fp16 + row maxima); o_proj quantises on the fly
                fused_ext.decode_attention_f16_amax(self.attn_f16, self.amax[li, 0], q, k, v, self.block_tables[li],
                                                    self.lengths, self.tpb, self.max_context, c.rope_theta,
                                                    single_launch=self.attn_single, tickets=self.tickets)
            elif self.fused >= 2 or self.l2_attn:   # attention with its split merge fused into the activation quant
                fused_ext.decode_attention_quant_fuse_sum(self._q_attn, q, k, v, self.block_tables[li], self.lengths,
                                                          self.tpb, self.max_context, c.rope_theta, mA, sA)
            else:
                attn = fused_attention_pure_dense.single_query_attention(
                    q, k, v, self.block_tables[li], self.lengths, None, 65536, self.tpb, hk * d // 2,
                    self.max_context, d, c.rope_theta, True, True, True)
                if per_chn:
                    fused_kernels.invoke_quant_fuse_sum(self._q_attn, attn.view(B, hq * d), mA, sA)
                else:
                    fused_kernels.invoke_quant(self._q_attn, attn.view(B, hq * d), sA)
            if self.fused >= 2:
                if self.fused >= 3:
                    sk = self._partial_f16(self.attn_f16, self.amax[li, 0], L["o"], mA, sA)
                else:
                    sk = self._partial(self._q_attn, L["o"])
                if self.pairs and l3:
                    # level 4: (add + norm + quant) -> gate_up + SiLU as one launch; down_proj quantises on the fly
                    G = L["gate_up"]
                    fused_ext.norm_gemm_fused(qa_h, self.x, L["ln2"], mB, sB, c.eps, G, self.mlp_act, self.ngf_sync[li, 1],
                                              self.ngf_err, slab=self.slab, sk=sk, producer=L["o"], p_ascales=sA, p_asums=mA,
                                              amax=self.amax[li, 1], clk=self.ngf_clk.get((li, 1)))
                    pending = (self._partial_f16(self.mlp_act, self.amax[li, 1], L["down"], mA, sA), L["down"])
                    continue
                if l3 and self.pf_down:
                    self._arm(L["down"], deferred=True)
                else:
                    self._arm(L["gate_up"], silu=l3)
                self._consume(qa_h, sk, L["o"], sA, mA, L["ln2"], mB, sB)
            else:
                peer = self.comm is not None and self.fused
                proj = self.comm.slot(B * c.hidden, (B, c.hidden)) if peer else self.proj_buf
                L["o"].forward(self._q_attn, sA, mA, proj)
                if not peer:
                    self._all_reduce(self.proj_buf)
                self._arm(L["gate_up"])
                if peer:
                    self.comm.add_rms_norm(qa_h, self.x, L["ln2"], mB, sB, c.eps)  # all-reduce(o_proj) + add + norm + quant
                elif self.fused:
                    fused_ext.add_rms_norm_general_fuse_sum(qa_h, self.x, self.proj_buf, L["ln2"], mB, sB, c.eps)
                else:
                    self.x.add_(self.proj_buf)
                    if per_chn:
                        layernorm_ops.rms_norm_general_fuse_sum(qa_h, self.x, L["ln2"], mB, sB, c.eps, True)
                    else:
                        layernorm_ops.rms_norm_general(qa_h, self.x, L["ln2"], sB, c.eps, True)
            if l3:
                # gate_up with silu_and_mul in its epilogue -> fp16 activation + row maxima; down_proj quantises on the fly
                G = L["gate_up"]       # (pf_down: not the armed tensor -> streams cold, non-temporal; down_proj then reads L2)
                if per_chn:
                    fused_ext.gemm_silu_per_chn(qa_h, G.qweight, G.s1_scales, sB, G.s1_szeros, mB, self.mlp_act,
                                                self.amax[li, 1])
                else:
                    fused_ext.gemm_silu_per_group(qa_h, G.qweight, G.s2_zeros, G.s2_scales, G.s1_scales, sB,
                                                  self.mlp_act, self.amax[li, 1])
                pending = (self._partial_f16(self.mlp_act, self.amax[li, 1], L["down"], mA, sA), L["down"])
                continue
            L["gate_up"].forward(qa_h, sB, mB, self.gate_up_buf)
            self._arm(L["down"], deferred=self.fused >= 2 and li < nl - 1)
            if self.fused:
                fused_ext.silu_mul_quant_fuse_sum(qa_i, self.gate_up_buf, mA, sA)
            else:
                activation_ops.silu_and_mul(self.mlp_act, self.gate_up_buf)
                if per_chn:
                    fused_kernels.invoke_quant_fuse_sum(qa_i, self.mlp_act, mA, sA)
                else:
                    fused_kernels.invoke_quant(qa_i, self.mlp_act, sA)
            if self.fused >= 2 and li < nl - 1:
                pending = (self._partial(qa_i, L["down"]), L["down"])
            else:
                peer = self.comm is not None and self.fused
                L["down"].forward(qa_i, sA, mA, self.comm.slot(B * c.hidden, (B, c.hidden)) if peer else self.proj_buf)
                if peer and li == nl - 1:
                    self.comm.all_reduce(self.proj_buf)
                elif not peer:
                    self._all_reduce(self.proj_buf)     # (peer: consumed by the next layer's add + norm)
                if not self.fused or li == nl - 1:
                    self.x.add_(self.proj_buf)
        if pending is not None:     # the last layer's down projection: epilogue + residual add inside the final norm
            sk, lin = pending
            fused_ext.splitk_add_rms_norm(self.normed, self.x, self.slab, sk, lin.s1_scales, sA,
                                          lin.s1_szeros if per_chn else None, mA if per_chn else None, self.final_norm, c.eps)
        else:
            layernorm_ops.rms_norm(self.normed, self.x, self.final_norm, c.eps, False)
        logits = torch.matmul(self.normed, self.lm_head.t())
        if self.fused:
            fused_ext.argmax(self.tokens, logits)      # same result as torch.argmax, 6 us instead of 47
        else:
            self.tokens.copy_(torch.argmax(logits, dim=-1))

    # ---- prefill (context stage) ---------------------------------------------------------------------------
    def prefill(self, prompt_len=None, tokens=None):
        """One context-stage pass over `prompt_len` tokens per sequence in the order of the reference's decoder
        layer at prefill shape (llama_w4a8_unpad.py:406-438 with M = B * prompt_len; ctx_update_kv.py:96-135;
        ctx_attn_func.py:68-73): norm+quant -> qkv GEMM -> RoPE in place + KV4 page write -> varlen causal
        attention -> quant -> o GEMM -> add -> norm+quant -> gate_up GEMM -> silu*mul -> quant -> down GEMM -> add.
        Fills the KV pools for positions [0, prompt_len) and leaves the first generated token in self.tokens
        (self.lengths = prompt_len), i.e. the qserve_benchmark.py protocol's prefill step.  Eager launches.
        `tokens`: the B * prompt_len prompt token ids, sequence after sequence (default: random ids)."""
        import omniserve_backend.fused_attention_fine_grained_dense as fgd
        from .backend.prefill_attn import flash_attn_varlen_func
        c, B, dev = self.cfg, self.B, self.device
        L = int(prompt_len if prompt_len is not None else self.lengths[0].item())
        if L < 1 or L + 1 > self.max_context:
            raise ValueError("prompt length does not fit the runner's KV pools")
        T = B * L
        hq, hk, d = self.hl, self.kl, c.head_dim
        f16, i8 = torch.float16, torch.int8
        buf = getattr(self, "_prefill_bufs", None)
        if buf is None or buf["T"] != T:
            buf = dict(T=T, x=torch.empty((T, c.hidden), dtype=f16, device=dev),
                       qh=torch.empty((T, c.hidden), dtype=i8, device=dev),
                       qi=torch.empty((T, self.il), dtype=i8, device=dev),
                       qa=torch.empty((T, hq * d), dtype=i8, device=dev),
                       qkv=torch.empty((T, (hq + 2 * hk) * d), dtype=f16, device=dev),
                       proj=torch.empty((T, c.hidden), dtype=f16, device=dev),
                       gu=torch.empty((T, 2 * self.il), dtype=f16, device=dev),
                       s1=torch.empty((T,), dtype=f16, device=dev), m1=torch.empty((T,), dtype=f16, device=dev),
                       s2=torch.empty((T,), dtype=f16, device=dev), m2=torch.empty((T,), dtype=f16, device=dev))
            self._prefill_bufs = buf
        x, qh, qi, qa, qkv, proj, gu = buf["x"], buf["qh"], buf["qi"], buf["qa"], buf["qkv"], buf["proj"], buf["gu"]
        sB, mB, sA, mA = buf["s1"], buf["m1"], buf["s2"], buf["m2"]
        if tokens is None:
            tokens = torch.randint(0, c.vocab, (T,), device=dev, generator=self.gen)
        elif tokens.numel() != T:
            raise ValueError("prefill expects %d prompt tokens, got %d" % (T, tokens.numel()))
        torch.index_select(self.embed, 0, tokens, out=x)
        lens = torch.full((B,), L, dtype=torch.int32, device=dev)
        cu = torch.arange(0, B + 1, dtype=torch.int32, device=dev) * L
        pad = fgd.compute_padding_offsets(cu, L, T)
        flags = torch.ones((hk,), dtype=torch.int32, device=dev)
        rank = torch.arange(hk, dtype=torch.int32, device=dev)
        q = qkv[:, : hq * d].view(T, hq, d)
        k = qkv[:, hq * d:(hq + hk) * d].view(T, hk, d)
        v = qkv[:, (hq + hk) * d:].view(T, hk, d)
        per_chn = c.group_size == -1
        if not per_chn:
            mA = mB = None      # g128: no activation sums (see _eager_step_body)
        for li, Ly in enumerate(self.layers):
            if li == 0 and per_chn:
                layernorm_ops.rms_norm_general_fuse_sum(qh, x, Ly["ln1"], mB, sB, c.eps, True)
            elif li == 0:
                layernorm_ops.rms_norm_general(qh, x, Ly["ln1"], sB, c.eps, True)
            else:
                fused_ext.add_rms_norm_general_fuse_sum(qh, x, proj, Ly["ln1"], mB, sB, c.eps)
            Ly["qkv"].forward(qh, sB, mB, qkv)
            fgd.apply_bias_rope_update_kv_cache(qkv, lens, None, pad, self.block_tables[li], None, flags, rank, hq, hk, L,
                                                self.tpb, hk * d // 2, 0, 0, 0, 0, 0, hk, 0, d, c.rope_theta, 1.0,
                                                1 << 20, True, True, True)
            attn = flash_attn_varlen_func(q, k, v, cu, cu, L, L, causal=True)
            if per_chn:
                fused_kernels.invoke_quant_fuse_sum(qa, attn.view(T, hq * d), mA, sA)
            else:
                fused_kernels.invoke_quant(qa, attn.view(T, hq * d), sA)
            Ly["o"].forward(qa, sA, mA, proj)
            self._all_reduce(proj)
            fused_ext.add_rms_norm_general_fuse_sum(qh, x, proj, Ly["ln2"], mB, sB, c.eps)
            Ly["gate_up"].forward(qh, sB, mB, gu)
            fused_ext.silu_mul_quant_fuse_sum(qi, gu, mA, sA)
            Ly["down"].forward(qi, sA, mA, proj)
            self._all_reduce(proj)
        x.add_(proj)
        last = x.view(B, L, c.hidden)[:, L - 1]
        self.x.copy_(last)
        layernorm_ops.rms_norm(self.normed, self.x, self.final_norm, c.eps, False)
        logits = torch.matmul(self.normed, self.lm_head.t())
        self.tokens.copy_(torch.argmax(logits, dim=-1))
        self.lengths.fill_(L)

    def _all_reduce(self, buf):
        if self.tp_size > 1:
            from . import tp
            tp.all_reduce_(buf, self.tp_group)

    # ---- accounting -----------------------------------------------------------
    def gemm_weight_bytes_per_step(self):
        return sum(L[n].weight_bytes() for L in self.layers for n in ("qkv", "o", "gate_up", "down"))

    def kv_bytes_per_step(self, context):
        c = self.cfg
        per_tok = 2 * (self.kl * c.head_dim // 2 + self.kl * 4)
        return per_tok * context * self.B * c.layers
