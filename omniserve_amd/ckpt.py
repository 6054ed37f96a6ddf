"""QServe checkpoint path for the MI355X hot path: converter, container I/O, loader.

What upstream does (all pure torch, offline) and what this file restates in its own code:

* scripts/ckpt_converter/quant_utils.py:96-140  `pseudo_quantize_tensor`   -> :func:`pseudo_quantize_tensor`
* omniserve/modeling/layers/quantized_linear/w4a8_linear.py:141-337 `W4A8OF16LinearDynamicInputScale.from_linear`
  (fake-quantised fp weight + lmquant scales -> qweight / s1_scales / s1_szeros / s2_scales / s2_zeros buffers,
  names and shapes w4a8_linear.py:43-100)                                    -> :func:`convert_linear`
* scripts/ckpt_converter/checkpoint_converter.py:88-164 (walk the decoder layers, convert every `*_proj`, copy the
  rest, save one state dict)                                                 -> :func:`convert_state_dict`
* omniserve/utils/weight_utils.py:88-161 `hf_model_weights_iterator` (safetensors or *.bin / *.pt)
                                                                             -> :func:`load_state_dict`
* omniserve/modeling/models/llama_w4a8_unpad.py:581-718 `load_weights` (q/k/v -> qkv_proj and gate/up -> gate_up_proj
  by concatenation along the output channels, `s2_*` along dim 1; keys containing "norm" are skipped, i.e. the norm
  weights stay at one: QoQ folds them into the following projection)          -> :func:`load_into_runner`

tests/test_oracle_golden_host.py and tests/test_ckpt_cpu.py pin the quantiser and the packer against fixtures produced
by the reference's own Python (tests/golden/make_golden*.py).  Everything here is host-side preparation: the tensors
it produces are consumed by the HIP kernels unchanged (nothing is re-packed at load time).

    python -m omniserve_amd.ckpt make-tiny OUT_DIR [--group-size 128]   # synthetic HF-style QServe checkpoint
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import torch

PROJ_NAMES = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


# ---------------------------------------------------------------------------------------------------------------
# quantisers
# ---------------------------------------------------------------------------------------------------------------
def pseudo_quantize_tensor(w: torch.Tensor, n_bit: int = 8, q_group_size: int = -1):
    """Asymmetric min/max fake quantisation (quant_utils.py:96-140, zero_point=True branch).
    -> (dequantised w, scales [rows, groups], zeros [rows, groups])."""
    shape = w.shape
    if q_group_size > 0:
        if shape[-1] % q_group_size != 0:
            raise ValueError("last dimension is not a multiple of the group size")
        w = w.reshape(-1, q_group_size)
    if w.dim() != 2:
        raise ValueError("expected a matrix")
    max_val = w.amax(dim=1, keepdim=True)
    min_val = w.amin(dim=1, keepdim=True)
    max_int = 2 ** n_bit - 1
    scales = (max_val - min_val).clamp(min=1e-5) / max_int
    zeros = (-torch.round(min_val / scales)).clamp_(0, max_int)
    dq = (torch.clamp(torch.round(w / scales) + zeros, 0, max_int) - zeros) * scales
    dq = dq.reshape(shape)
    return dq, scales.view(shape[0], -1), zeros.view(shape[0], -1)


def qoq_quantize_weight(w: torch.Tensor, group_size: int = -1):
    """Calibration-free stand-in for the lmquant dump the upstream converter consumes (model.pt + scale.pt): turn an
    fp weight [N, K] into (fake-quantised weight, s1_scale [N], s2_scale [N, K/G] or None, zeros), the four inputs of
    `convert_linear`, following QoQ's progressive scheme (README.md "QoQ", w4a8_linear.py:170-199):
      per-channel:  asymmetric 4-bit per output channel: w ~ (u - z) * s1, u in 0..15.
      g128:         level 1 symmetric int8 per channel with the protective range 119 (s1 = max|w| / 119); level 2
                    asymmetric 4-bit per group of the int8 values with INTEGER scale s2 and zero z, which keeps
                    |(u - z) * s2| <= 127 (the invariant the second-level byte arithmetic relies on)."""
    w = w.float()
    N, K = w.shape
    if group_size == -1:
        dq, scales, zeros = pseudo_quantize_tensor(w, n_bit=4, q_group_size=-1)
        s1 = scales.reshape(N).half()
        # re-derive the fake-quantised weight from the fp16 scale the checkpoint stores (from_linear divides by it)
        u = torch.clamp(torch.round(w / s1.float()[:, None]) + zeros, 0, 15)
        return (u - zeros) * s1.float()[:, None], s1, None, zeros.reshape(N).to(torch.int8)
    if K % group_size != 0:
        raise ValueError("in_features is not a multiple of the group size")
    s1 = (w.abs().amax(dim=1).clamp(min=1e-5) / 119.0).half()
    w8 = torch.round(w / s1.float()[:, None]).clamp_(-119, 119).reshape(N, K // group_size, group_size)
    mx, mn = w8.amax(dim=2, keepdim=True), w8.amin(dim=2, keepdim=True)
    s2 = torch.ceil((mx - mn) / 15.0).clamp_(min=1.0)
    z = torch.round(-mn / s2).clamp_(0, 15)
    u = torch.clamp(torch.round(w8 / s2) + z, 0, 15)
    w8q = (u - z) * s2                                                  # |.| <= 127 by construction
    if w8q.abs().max() > 127:
        raise AssertionError("level-2 weights left the int8 range")
    fake = w8q.reshape(N, K) * s1.float()[:, None]
    return fake, s1, s2.reshape(N, K // group_size).to(torch.int8), z.reshape(N, K // group_size).to(torch.int8)


# ---------------------------------------------------------------------------------------------------------------
# packing (w4a8_linear.py:141-337)
# ---------------------------------------------------------------------------------------------------------------
def pack_w4_codes(u: torch.Tensor) -> torch.Tensor:
    """uint4 codes [N, K] (any integer dtype, values 0..15) -> packed int8 [N, K/2] in the layout the kernels stream:
    32 x 32-code tiles of 512 bytes, tile(nb, kb)[lane = n3*4 + k6][byte = k5*8 + n2*4 + k7], nibble n1, with
    n = nb*32 + n1*16 + n2*8 + n3 and k = kb*32 + k5*16 + k6*4 + k7 (the closed form of the two permutes at
    w4a8_linear.py:296-327)."""
    N, K = u.shape
    if N % 32 or K % 32:
        raise ValueError("W4A8 packing needs N and K to be multiples of 32")
    t = u.to(torch.uint8).reshape(N // 32, 2, 2, 8, K // 32, 2, 4, 4)      # nb n1 n2 n3 kb k5 k6 k7
    t = t.permute(0, 4, 3, 6, 5, 2, 7, 1).contiguous()                     # nb kb n3 k6 k5 n2 k7 n1
    packed = (t[..., 1] << 4) | t[..., 0]
    return packed.reshape(N, K // 2).view(torch.int8)


def _permute_group_param(p: torch.Tensor) -> torch.Tensor:
    """[N, K/G] -> stored [K/G, N] with the per-32-channel order pos = i*4 + j <-> n = j*8 + i (w4a8_linear.py:236-282)."""
    N, NG = p.shape
    return p.t().reshape(NG, N // 32, 4, 8).permute(0, 1, 3, 2).contiguous().reshape(NG, N)


def convert_linear(weight: torch.Tensor, s1_scale: torch.Tensor, zeros: torch.Tensor, group_size: int = -1,
                   s2_scale: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Fake-quantised weight [N, K] + its quantisation parameters -> the W4A8 buffers of one linear layer
    (from_linear, w4a8_linear.py:141-337; per-channel: zeros [N]; per-group: s2_scale, zeros [N, K/G])."""
    w = weight.float().clone()
    N, K = w.shape
    s1f = s1_scale.reshape(N, 1).float()
    if group_size == -1:
        z = zeros.reshape(N, 1).to(torch.float16)
        q = (w / s1_scale.reshape(N, 1).to(w.dtype)).round_() + z.float()
        if q.min() < 0 or q.max() > 15:
            raise ValueError("per-channel codes out of range: the weight is not a W4 fake-quantised tensor")
        s1h = s1_scale.reshape(N).half()
        return {"qweight": pack_w4_codes(q), "s1_scales": s1h,
                "s1_szeros": (zeros.reshape(N).to(torch.float16) * s1h).half()}
    if s2_scale is None:
        raise ValueError("per-group conversion needs s2_scale")
    w8 = (w / s1f).round_()
    if w8.min() < -128 or w8.max() > 127:
        raise ValueError("stage 1: quantised weight out of the int8 range")
    G = K // group_size
    s2 = s2_scale.reshape(N, G, 1)
    z = zeros.reshape(N, G, 1)
    q = w8.reshape(N, G, group_size) / s2.to(torch.float16).float() + z.to(torch.float16).float()
    if q.min() < 0 or q.max() > 15:
        raise ValueError("stage 2: codes out of range")
    s2p = _permute_group_param(s2.reshape(N, G).to(torch.int64))
    z2p = _permute_group_param((-z.reshape(N, G).to(torch.int64)))
    return {"qweight": pack_w4_codes(q.reshape(N, K)), "s1_scales": s1_scale.reshape(N).half(),
            "s2_scales": s2p.to(torch.int8), "s2_zeros": (z2p * s2p).to(torch.int8)}


def convert_state_dict(fp_state: Dict[str, torch.Tensor], group_size: int = -1) -> Dict[str, torch.Tensor]:
    """HF-style Llama state dict with fp `*_proj.weight` tensors -> QServe state dict (every projection replaced by
    its W4A8 buffers under the same prefix, everything else copied as fp16), using :func:`qoq_quantize_weight`."""
    out = {}
    for name, t in fp_state.items():
        if name.endswith(".weight") and any(("." + p + ".") in name for p in PROJ_NAMES):
            prefix = name[: -len("weight")]
            fake, s1, s2, z = qoq_quantize_weight(t, group_size)
            for k, v in convert_linear(fake, s1, z, group_size, s2).items():
                out[prefix + k] = v
        else:
            out[name] = t.half() if t.is_floating_point() else t
    return out


# ---------------------------------------------------------------------------------------------------------------
# container I/O (weight_utils.py:88-161: *.safetensors preferred, *.bin / *.pt as fall-back)
# ---------------------------------------------------------------------------------------------------------------
def save_checkpoint(state: Dict[str, torch.Tensor], out_dir: str, config: Optional[dict] = None,
                    safetensors: bool = True) -> str:
    os.makedirs(out_dir, exist_ok=True)
    state = {k: v.detach().cpu().contiguous() for k, v in state.items()}
    if safetensors:
        from safetensors.torch import save_file
        path = os.path.join(out_dir, "model.safetensors")
        save_file(state, path)
    else:
        path = os.path.join(out_dir, "pytorch_model.bin")
        torch.save(state, path)
    if config is not None:
        with open(os.path.join(out_dir, "config.json"), "w") as f:
            json.dump(config, f, indent=1)
    return path


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """A checkpoint directory (all *.safetensors, else all *.bin / *.pt) or a single file -> one name -> tensor dict."""
    files = [path]
    if os.path.isdir(path):
        names = sorted(os.listdir(path))
        files = [os.path.join(path, n) for n in names if n.endswith(".safetensors")]
        if not files:
            files = [os.path.join(path, n) for n in names if n.endswith((".bin", ".pt"))]
        if not files:
            raise RuntimeError("no model weights under %s" % path)
    state = {}
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors import safe_open
            with safe_open(f, framework="pt") as sf:
                for k in sf.keys():
                    state[k] = sf.get_tensor(k)
        else:
            state.update(torch.load(f, map_location="cpu"))
    return state


# ---------------------------------------------------------------------------------------------------------------
# loader: checkpoint tensors -> the decode runner's layers
# ---------------------------------------------------------------------------------------------------------------
def _cat(parts, dim):
    return torch.cat(parts, dim=dim).contiguous()


def fused_linear(state: Dict[str, torch.Tensor], prefixes, group_size: int):
    """The W4A8 buffers of the projections `prefixes` concatenated along the output channels, as load_weights does
    for qkv_proj and gate_up_proj (llama_w4a8_unpad.py:648-705): rows of qweight / s1_*, columns of s2_*."""
    names = ["qweight", "s1_scales"] + (["s1_szeros"] if group_size == -1 else ["s2_scales", "s2_zeros"])
    out = {}
    for n in names:
        parts = [state[p + "." + n] for p in prefixes]
        out[n] = _cat(parts, 1 if n.startswith("s2_") else 0)
    return out


def load_into_runner(runner, state: Dict[str, torch.Tensor], load_norm_weights: bool = False) -> None:
    """Overwrite a DecodeRunner's synthetic weights with a QServe checkpoint (in place: captured graphs stay valid).
    Norm weights are skipped as upstream does (`if "norm" in name: continue`, llama_w4a8_unpad.py:635) unless asked
    for.  Tensor-parallel runners take their shard of every projection (omniserve_amd/tp.py rules)."""
    from .runtime import W4A8Linear
    c = runner.cfg
    g = c.group_size
    r, w = runner.tp_rank, runner.tp_size
    d = c.head_dim
    hl, kl, il = runner.hl, runner.kl, runner.il

    def make(buffers, n, k):
        lin = object.__new__(W4A8Linear)
        lin.n, lin.k, lin.group = n, k, g
        lin.qweight = buffers["qweight"].view(torch.int8)
        lin.s1_scales = buffers["s1_scales"].half()
        if g == -1:
            lin.s1_szeros = buffers["s1_szeros"].half()
        else:
            lin.s2_scales, lin.s2_zeros = buffers["s2_scales"].view(torch.int8), buffers["s2_zeros"].view(torch.int8)
        return lin

    def assign(dst, src):
        for n in ("qweight", "s1_scales", "s1_szeros", "s2_scales", "s2_zeros"):
            if hasattr(dst, n):
                t = getattr(src, n)
                if getattr(dst, n).shape != t.shape:
                    raise RuntimeError("checkpoint tensor %s has shape %s, the model expects %s" % (
                        n, tuple(t.shape), tuple(getattr(dst, n).shape)))
                getattr(dst, n).copy_(t.to(getattr(dst, n).device))

    for li, L in enumerate(runner.layers):
        pre = "model.layers.%d." % li
        qkv = make(fused_linear(state, [pre + "self_attn." + p for p in ("q_proj", "k_proj", "v_proj")], g),
                   (c.heads + 2 * c.kv_heads) * d, c.hidden)
        o = make(fused_linear(state, [pre + "self_attn.o_proj"], g), c.hidden, c.heads * d)
        gate_up = make(fused_linear(state, [pre + "mlp." + p for p in ("gate_proj", "up_proj")], g), 2 * c.inter, c.hidden)
        down = make(fused_linear(state, [pre + "mlp.down_proj"], g), c.hidden, c.inter)
        if w > 1:
            qkv = qkv.select_rows([(r * hl * d, (r + 1) * hl * d),
                                   (c.heads * d + r * kl * d, c.heads * d + (r + 1) * kl * d),
                                   ((c.heads + c.kv_heads) * d + r * kl * d, (c.heads + c.kv_heads) * d + (r + 1) * kl * d)])
            o = o.shard_k(r, w)
            gate_up = gate_up.select_rows([(r * il, (r + 1) * il), (c.inter + r * il, c.inter + (r + 1) * il)])
            down = down.shard_k(r, w)
        for name, lin in (("qkv", qkv), ("o", o), ("gate_up", gate_up), ("down", down)):
            assign(L[name], lin)
        if load_norm_weights:
            L["ln1"].copy_(state[pre + "input_layernorm.weight"].half().to(L["ln1"].device))
            L["ln2"].copy_(state[pre + "post_attention_layernorm.weight"].half().to(L["ln2"].device))
        else:
            L["ln1"].fill_(1.0)
            L["ln2"].fill_(1.0)
    runner.embed.copy_(state["model.embed_tokens.weight"].half().to(runner.embed.device))
    runner.lm_head.copy_(state["lm_head.weight"].half().to(runner.lm_head.device))
    if load_norm_weights and "model.norm.weight" in state:
        runner.final_norm.copy_(state["model.norm.weight"].half().to(runner.final_norm.device))
    else:
        runner.final_norm.fill_(1.0)


# ---------------------------------------------------------------------------------------------------------------
# a tiny synthetic HF-style checkpoint (no network: random fp weights of a Llama-shaped model, converted here)
# ---------------------------------------------------------------------------------------------------------------
def tiny_fp_state(cfg, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    d = cfg.head_dim

    def rnd(n, k, s=0.05):
        return torch.randn((n, k), generator=g) * s

    st = {"model.embed_tokens.weight": rnd(cfg.vocab, cfg.hidden, 0.5), "lm_head.weight": rnd(cfg.vocab, cfg.hidden),
          "model.norm.weight": torch.ones(cfg.hidden)}
    for i in range(cfg.layers):
        p = "model.layers.%d." % i
        st[p + "self_attn.q_proj.weight"] = rnd(cfg.heads * d, cfg.hidden)
        st[p + "self_attn.k_proj.weight"] = rnd(cfg.kv_heads * d, cfg.hidden)
        st[p + "self_attn.v_proj.weight"] = rnd(cfg.kv_heads * d, cfg.hidden)
        st[p + "self_attn.o_proj.weight"] = rnd(cfg.hidden, cfg.heads * d)
        st[p + "mlp.gate_proj.weight"] = rnd(cfg.inter, cfg.hidden)
        st[p + "mlp.up_proj.weight"] = rnd(cfg.inter, cfg.hidden)
        st[p + "mlp.down_proj.weight"] = rnd(cfg.hidden, cfg.inter)
        st[p + "input_layernorm.weight"] = torch.ones(cfg.hidden)
        st[p + "post_attention_layernorm.weight"] = torch.ones(cfg.hidden)
    return st


def make_tiny_checkpoint(out_dir: str, group_size: int = -1, seed: int = 0, safetensors: bool = True):
    from .runtime import LlamaConfig
    cfg = LlamaConfig.tiny()
    cfg.group_size = group_size
    state = convert_state_dict(tiny_fp_state(cfg, seed), group_size)
    config = {"architectures": ["LlamaForCausalLM"], "hidden_size": cfg.hidden, "intermediate_size": cfg.inter,
              "num_attention_heads": cfg.heads, "num_key_value_heads": cfg.kv_heads, "num_hidden_layers": cfg.layers,
              "vocab_size": cfg.vocab, "rope_theta": cfg.rope_theta, "rms_norm_eps": cfg.eps,
              "quantization": {"w_bit": 4, "a_bit": 8, "group_size": group_size}}
    return save_checkpoint(state, out_dir, config, safetensors), cfg


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    mk = sub.add_parser("make-tiny")
    mk.add_argument("out")
    mk.add_argument("--group-size", type=int, default=-1)
    mk.add_argument("--bin", action="store_true", help="write pytorch_model.bin instead of model.safetensors")
    a = ap.parse_args()
    path, _ = make_tiny_checkpoint(a.out, a.group_size, safetensors=not a.bin)
    print("wrote", path)
