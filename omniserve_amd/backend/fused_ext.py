"""Opt-in fused entry points (NOT in the reference's omniserve_backend).
Each is bit-identical to the pair of reference calls it replaces."""
import torch

from .. import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


def add_rms_norm_general_fuse_sum(out, residual, delta, weight, input_sum, scaling, epsilon):
    """residual += delta (in place, fp16), then rms_norm_general_fuse_sum(out, residual, ...).
    input_sum=None: no row sum (= rms_norm_general, what the W8A8 / per-group layers call)."""
    _lib.require_cuda(out, residual, delta, weight, input_sum, scaling)
    hidden = residual.shape[-1]
    tokens = residual.numel() // hidden
    if not residual.is_contiguous() or not delta.is_contiguous():
        raise RuntimeError("add_rms_norm_general_fuse_sum: residual and delta must be contiguous")
    rc = _lib.lib().omni_add_rms_norm_general_fuse_sum(
        out.data_ptr(), residual.data_ptr(), delta.data_ptr(), weight.data_ptr(), _ptr(input_sum),
        scaling.data_ptr(), float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "fused_ext.add_rms_norm_general_fuse_sum")


def silu_mul_quant_fuse_sum(out, input, input_sum, scale):
    """silu_and_mul(tmp, input); invoke_quant_fuse_sum(out, tmp, input_sum, scale) without tmp.
    input_sum=None: invoke_quant (no row sum)."""
    _lib.require_cuda(out, input, input_sum, scale)
    d = input.shape[-1] // 2
    tokens = input.numel() // input.shape[-1]
    rc = _lib.lib().omni_silu_mul_quant_fuse_sum(out.data_ptr(), input.data_ptr(), _ptr(input_sum),
                                                 scale.data_ptr(), tokens, d, _lib.current_stream())
    _lib.check(rc, "fused_ext.silu_mul_quant_fuse_sum")


def gemm_partial_per_chn(in_feats, kernel, slab):
    """Decode-shape W4A8 per-channel GEMM without its epilogue: writes int32 partial sums
    slab[sk][M][N] and returns sk.  Pair with splitk_add_rms_norm_general_fuse_sum."""
    import ctypes
    _lib.require_cuda(in_feats, kernel, slab)
    M, K = in_feats.shape
    N = kernel.shape[0]
    sk = ctypes.c_int(0)
    rc = _lib.lib().omni_w4a8_per_chn_gemm_partial(in_feats.data_ptr(), kernel.data_ptr(), slab.data_ptr(),
                                                   slab.numel() * slab.element_size(), M, N, K,
                                                   ctypes.byref(sk), _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_partial_per_chn")
    return sk.value


def splitk_add_rms_norm_general_fuse_sum(out, residual, slab, sk, wscales, ascales_in, w_szs, a_ssums_in,
                                         weight, input_sum, scaling, epsilon):
    """residual += fp16(per-channel GEMM epilogue(sum of sk slabs)); then norm + quant (+sum) of it."""
    _lib.require_cuda(out, residual, slab, wscales, ascales_in, w_szs, a_ssums_in, weight, input_sum, scaling)
    hidden = residual.shape[-1]
    tokens = residual.numel() // hidden
    rc = _lib.lib().omni_splitk_add_rms_norm_general_fuse_sum(
        out.data_ptr(), residual.data_ptr(), slab.data_ptr(), int(sk), wscales.data_ptr(), ascales_in.data_ptr(),
        w_szs.data_ptr(), a_ssums_in.data_ptr(), weight.data_ptr(), input_sum.data_ptr(), scaling.data_ptr(),
        float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "fused_ext.splitk_add_rms_norm_general_fuse_sum")


def gemm_partial_w8a8(in_feats, weight, slab):
    """Decode-shape W8A8 GEMM without its epilogue (int32 partial sums slab[sk][M][N], returns sk); pair with
    splitk_w8_add_rms_norm_general_fuse_sum."""
    import ctypes
    _lib.require_cuda(in_feats, weight, slab)
    M, K = in_feats.shape
    N = weight.shape[0]
    sk = ctypes.c_int(0)
    rc = _lib.lib().omni_w8a8_gemm_partial(in_feats.data_ptr(), weight.data_ptr(), slab.data_ptr(),
                                           slab.numel() * slab.element_size(), M, N, K, ctypes.byref(sk),
                                           _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_partial_w8a8")
    return sk.value


def gemm_partial_per_group(in_feats, qweight, s2_zeros, s2_scales, slab):
    """Decode-shape g128 W4A8 GEMM without its epilogue (int32 slabs, returns sk); the epilogue formula is the W8A8 one, so
    the consumer is splitk_w8_add_rms_norm_general_fuse_sum."""
    import ctypes
    _lib.require_cuda(in_feats, qweight, s2_zeros, s2_scales, slab)
    M, K = in_feats.shape
    N = qweight.shape[0]
    sk = ctypes.c_int(0)
    rc = _lib.lib().omni_w4a8_per_group_gemm_partial(in_feats.data_ptr(), qweight.data_ptr(), s2_zeros.data_ptr(),
                                                     s2_scales.data_ptr(), slab.data_ptr(),
                                                     slab.numel() * slab.element_size(), M, N, K, ctypes.byref(sk),
                                                     _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_partial_per_group")
    return sk.value


def splitk_w8_add_rms_norm_general_fuse_sum(out, residual, slab, sk, wscales, ascales_in, weight, input_sum, scaling,
                                            epsilon):
    """residual += fp16(W8A8 GEMM epilogue(sum of sk slabs)); then norm + quant (+sum) of it."""
    _lib.require_cuda(out, residual, slab, wscales, ascales_in, weight, input_sum, scaling)
    hidden = residual.shape[-1]
    tokens = residual.numel() // hidden
    rc = _lib.lib().omni_splitk_w8_add_rms_norm_general_fuse_sum(
        out.data_ptr(), residual.data_ptr(), slab.data_ptr(), int(sk), wscales.data_ptr(), ascales_in.data_ptr(),
        weight.data_ptr(), _ptr(input_sum), scaling.data_ptr(), float(epsilon), tokens, hidden,
        _lib.current_stream())
    _lib.check(rc, "fused_ext.splitk_w8_add_rms_norm_general_fuse_sum")


def splitk_add_rms_norm(out, residual, slab, sk, wscales, ascales_in, w_szs, a_ssums_in, weight, epsilon):
    """The LAST layer's deferred down projection consumed by the model's final norm: residual += fp16(GEMM epilogue(sum of
    sk slabs)); out fp16 = rms_norm(residual) (layernorm_ops.rms_norm, no quantisation).  w_szs / a_ssums_in both None: the
    W8A8 / per-group epilogue; both given: the per-channel W4A8 one.  Bit-identical to GEMM -> add -> rms_norm."""
    _lib.require_cuda(out, residual, slab, wscales, ascales_in, w_szs, a_ssums_in, weight)
    hidden = residual.shape[-1]
    tokens = residual.numel() // hidden
    rc = _lib.lib().omni_splitk_add_rms_norm(out.data_ptr(), residual.data_ptr(), slab.data_ptr(), int(sk),
                                             wscales.data_ptr(), ascales_in.data_ptr(), _ptr(w_szs), _ptr(a_ssums_in),
                                             weight.data_ptr(), float(epsilon), tokens, hidden, _lib.current_stream())
    _lib.check(rc, "fused_ext.splitk_add_rms_norm")


def decode_arm_qkv_slabs(slab, sk, rows, width, col_q, col_k, col_v, wscales, ascales, w_szs=None, a_ssums=None):
    """The next decode-attention call of this thread reads the current token's q / k / v from the qkv projection's int32
    split-K slabs (gemm_partial_*: slab [sk][rows][width]) and applies the projection's epilogue itself: no slab epilogue
    launch between the projection and the attention, same bits.  col_*: first channel of the q / k / v blocks of a row.
    w_szs / a_ssums: per-channel W4A8 zero term (both) or None (W8A8 / per-group).  slab=None disarms."""
    if slab is None:
        _lib.check(_lib.lib().omni_decode_arm_qkv_slabs(None, 0, 0, 0, 0, 0, 0, None, None, None, None),
                   "fused_ext.decode_arm_qkv_slabs")
        return
    _lib.require_cuda(slab, wscales, ascales, w_szs, a_ssums)
    rc = _lib.lib().omni_decode_arm_qkv_slabs(slab.data_ptr(), int(sk), int(rows), int(width), int(col_q), int(col_k),
                                              int(col_v), wscales.data_ptr(), ascales.data_ptr(), _ptr(w_szs),
                                              _ptr(a_ssums))
    _lib.check(rc, "fused_ext.decode_arm_qkv_slabs")


def decode_step_begin(out, table, idx, lengths=None, zero=None):
    """First launch of a decode step: out = table[idx] (embed_rows), lengths += 1 (int32), zero[...] = 0 (int32 / uint32
    words: the step's row-maximum slots) in one kernel instead of three."""
    _lib.require_cuda(out, table, idx, lengths, zero)
    if table.dtype != torch.float16 or out.dtype != torch.float16 or table.dim() != 2 or not table.is_contiguous() or \
            not out.is_contiguous() or out.shape[-1] != table.shape[1]:
        raise RuntimeError("decode_step_begin: contiguous fp16 table [V, cols] and out [rows, cols] expected")
    if idx.dtype != torch.int64 or not idx.is_contiguous() or idx.numel() != out.numel() // table.shape[1]:
        raise RuntimeError("decode_step_begin: idx must be a contiguous int64 tensor with one id per output row")
    if lengths is not None and (lengths.dtype != torch.int32 or not lengths.is_contiguous()):
        raise RuntimeError("decode_step_begin: lengths must be a contiguous int32 tensor")
    if zero is not None and (zero.element_size() != 4 or not zero.is_contiguous()):
        raise RuntimeError("decode_step_begin: zero must be a contiguous tensor of 32-bit words")
    rc = _lib.lib().omni_decode_step_begin(out.data_ptr(), table.data_ptr(), idx.data_ptr(), idx.numel(), table.shape[1],
                                           table.shape[0], _ptr(lengths), 0 if lengths is None else lengths.numel(),
                                           _ptr(zero), 0 if zero is None else zero.numel(), _lib.current_stream())
    _lib.check(rc, "fused_ext.decode_step_begin")


def decode_attention_quant_fuse_sum(out_i8, q, k, v, kv_pointers, lengths, tokens_per_block, timestep,
                                    rotary_base, input_sum, scale):
    """single_query_attention (KV4 + zeros, neox RoPE) followed by invoke_quant_fuse_sum of its
    [B, Hq*Dh] output, with the flash-decoding merge fused into the quantisation kernel."""
    import ctypes
    from ..rope import rope_table
    _lib.require_cuda(out_i8, q, k, v, kv_pointers, lengths, input_sum, scale)
    B, Hq, D = q.shape
    Hkv = k.shape[1]
    max_ctx = max(int(timestep), 1)
    table = rope_table(max_ctx + 1, D, float(rotary_base), 1.0, q.device)
    need = _lib.lib().omni_kv4_decode_workspace_bytes(B, Hq, D, max_ctx)
    ws = _lib.workspace(need, q.device, "attn")
    ns = ctypes.c_int(0)
    rc = _lib.lib().omni_kv4_decode_attention_partial(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), kv_pointers.data_ptr(),
        lengths.data_ptr(), B, kv_pointers.shape[-1], Hq, Hkv, D, int(tokens_per_block), max_ctx,
        table.data_ptr(), table.shape[0], ws.data_ptr(), ws.numel(), ctypes.byref(ns), _lib.current_stream())
    _lib.check(rc, "fused_ext.decode_attention_quant_fuse_sum (partials)")
    ml_bytes = B * Hq * ns.value * 2 * 4
    rc = _lib.lib().omni_attn_merge_quant_fuse_sum(out_i8.data_ptr(), ws.data_ptr(), ws.data_ptr() + ml_bytes, ns.value,
                                                   _ptr(input_sum), scale.data_ptr(), B, Hq, _lib.current_stream())
    _lib.check(rc, "fused_ext.decode_attention_quant_fuse_sum (merge+quant)")


# ---- row-kernel-free decode layer (round 3): the quantisers between two projections folded into the projections ----------
AMAX_WORDS = 8 * 16 * 8      # csrc/common.h: [XCD][row][sub-slot] row-maximum candidates of one activation tensor (<= 16 rows)


def new_amax_slots(rows, device):
    """Zeroed buffer of row-maximum candidates (f32 bit patterns, layout [8 XCDs][16 rows][8 sub-slots]) for an
    activation of `rows` <= 16 rows.  A producer (gemm_silu_*, decode_attention_f16_amax) RAISES the words with an integer
    max: zero the buffer before every producer call that should start a fresh maximum."""
    if rows > 16:
        raise RuntimeError("row-maximum hand-off covers at most 16 activation rows")
    return torch.zeros((AMAX_WORDS,), dtype=torch.int32, device=device)


def amax_rows(amax, rows):
    """float32 [rows]: the row maxima a buffer of candidates currently holds (tests / debugging)."""
    return amax.view(torch.float32).view(8, 16, 8).amax(dim=(0, 2))[:rows]


def gemm_silu_per_chn(in_feats, qweight, wscales, ascales, w_szs, a_ssums, act, amax):
    """gate_up projection + silu_and_mul in one kernel: act fp16 [M, N/2] = silu_and_mul(gemm_forward_cuda(...)) bit for
    bit, and the row maxima of |act| raised in `amax` (new_amax_slots).  M <= 16."""
    _lib.require_cuda(in_feats, qweight, wscales, ascales, w_szs, a_ssums, act, amax)
    M, K = in_feats.shape
    N = qweight.shape[0]
    if act.dtype != torch.float16 or tuple(act.shape) != (M, N // 2) or not act.is_contiguous():
        raise RuntimeError("gemm_silu_per_chn: act must be a contiguous fp16 [M, N/2] tensor")
    rc = _lib.lib().omni_w4a8_per_chn_gemm_silu(in_feats.data_ptr(), qweight.data_ptr(), wscales.data_ptr(),
                                                ascales.data_ptr(), w_szs.data_ptr(), a_ssums.data_ptr(),
                                                act.data_ptr(), amax.data_ptr(), M, N, K, _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_silu_per_chn")


def gemm_silu_per_group(in_feats, qweight, s2_zeros, s2_scales, wscales, ascales, act, amax):
    """The g128 form of gemm_silu_per_chn."""
    _lib.require_cuda(in_feats, qweight, s2_zeros, s2_scales, wscales, ascales, act, amax)
    M, K = in_feats.shape
    N = qweight.shape[0]
    if act.dtype != torch.float16 or tuple(act.shape) != (M, N // 2) or not act.is_contiguous():
        raise RuntimeError("gemm_silu_per_group: act must be a contiguous fp16 [M, N/2] tensor")
    rc = _lib.lib().omni_w4a8_per_group_gemm_silu(in_feats.data_ptr(), qweight.data_ptr(), s2_zeros.data_ptr(),
                                                  s2_scales.data_ptr(), wscales.data_ptr(), ascales.data_ptr(),
                                                  act.data_ptr(), amax.data_ptr(), M, N, K, _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_silu_per_group")


def gemm_partial_f16_per_chn(act, amax, qweight, slab, sum_out, scale_out):
    """o / down projection straight from fp16 activations: the int8 codes invoke_quant_fuse_sum(act) would produce are
    computed inside the GEMV from `amax` (row maxima left by the producer of `act`); writes int32 split-K slabs (returns
    sk, as gemm_partial_per_chn) and -- by rider workgroups -- the fp16 row sums / scales of invoke_quant_fuse_sum into
    sum_out / scale_out, for splitk_add_rms_norm_general_fuse_sum.  M <= 16, K <= 16384."""
    import ctypes
    _lib.require_cuda(act, amax, qweight, slab, sum_out, scale_out)
    M, K = act.shape
    N = qweight.shape[0]
    if act.dtype != torch.float16 or not act.is_contiguous():
        raise RuntimeError("gemm_partial_f16_per_chn: act must be a contiguous fp16 [M, K] tensor")
    sk = ctypes.c_int(0)
    rc = _lib.lib().omni_w4a8_per_chn_gemm_partial_f16(act.data_ptr(), amax.data_ptr(), qweight.data_ptr(),
                                                       slab.data_ptr(), slab.numel() * slab.element_size(),
                                                       _ptr(sum_out), scale_out.data_ptr(), M, N, K, ctypes.byref(sk),
                                                       _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_partial_f16_per_chn")
    return sk.value


def gemm_partial_f16_per_group(act, amax, qweight, s2_zeros, s2_scales, slab, sum_out, scale_out):
    """The g128 form of gemm_partial_f16_per_chn (sum_out may be None: the per-group layers read no row sums)."""
    import ctypes
    _lib.require_cuda(act, amax, qweight, s2_zeros, s2_scales, slab, sum_out, scale_out)
    M, K = act.shape
    N = qweight.shape[0]
    if act.dtype != torch.float16 or not act.is_contiguous():
        raise RuntimeError("gemm_partial_f16_per_group: act must be a contiguous fp16 [M, K] tensor")
    sk = ctypes.c_int(0)
    rc = _lib.lib().omni_w4a8_per_group_gemm_partial_f16(act.data_ptr(), amax.data_ptr(), qweight.data_ptr(),
                                                         s2_zeros.data_ptr(), s2_scales.data_ptr(), slab.data_ptr(),
                                                         slab.numel() * slab.element_size(), _ptr(sum_out),
                                                         scale_out.data_ptr(), M, N, K, ctypes.byref(sk),
                                                         _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_partial_f16_per_group")
    return sk.value


def gemm_silu_w8a8(in_feats, weight, wscales, ascales, act, amax):
    """The W8A8 form of gemm_silu_per_chn (LServe models: gate_up_proj -> silu*mul, llama_w8a8_unpad.py:94-105): act fp16
    [M, N/2] bit for bit what w8a8_gemm_forward_cuda -> silu_and_mul leave, row maxima of |act| raised in `amax`.  M <= 16."""
    _lib.require_cuda(in_feats, weight, wscales, ascales, act, amax)
    M, K = in_feats.shape
    N = weight.shape[0]
    if act.dtype != torch.float16 or tuple(act.shape) != (M, N // 2) or not act.is_contiguous():
        raise RuntimeError("gemm_silu_w8a8: act must be a contiguous fp16 [M, N/2] tensor")
    rc = _lib.lib().omni_w8a8_gemm_silu(in_feats.data_ptr(), weight.data_ptr(), wscales.data_ptr(), ascales.data_ptr(),
                                        act.data_ptr(), amax.data_ptr(), M, N, K, _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_silu_w8a8")


def gemm_partial_f16_w8a8(act, amax, weight, slab, scale_out):
    """The W8A8 form of gemm_partial_f16_per_chn: slabs of gemm_partial_w8a8 on the codes invoke_quant(act) would produce,
    scale_out = invoke_quant's scales (no row sums: W8A8 has no zero-point term).  M <= 16, any K."""
    import ctypes
    _lib.require_cuda(act, amax, weight, slab, scale_out)
    M, K = act.shape
    N = weight.shape[0]
    if act.dtype != torch.float16 or not act.is_contiguous():
        raise RuntimeError("gemm_partial_f16_w8a8: act must be a contiguous fp16 [M, K] tensor")
    sk = ctypes.c_int(0)
    rc = _lib.lib().omni_w8a8_gemm_partial_f16(act.data_ptr(), amax.data_ptr(), weight.data_ptr(), slab.data_ptr(),
                                               slab.numel() * slab.element_size(), scale_out.data_ptr(), M, N, K,
                                               ctypes.byref(sk), _lib.current_stream())
    _lib.check(rc, "fused_ext.gemm_partial_f16_w8a8")
    return sk.value


_TICKETS = {}


TICKET_WORDS = 4096


def new_tickets(device):
    """Ticket words of the single-launch decode attention: zero between launches (the last arriver resets its word).
    include/omniserve_hip.h asks for one buffer per stream: a runner OWNS one (allocated once, before any capture) and hands it
    to decode_attention_f16_amax -- two runners decoding concurrently on two streams then never share words, and a captured
    graph holds no allocation made during capture (the per-(device, stream) cache below missed on torch's capture
    stream and allocated + zero-filled inside the graph)."""
    return torch.zeros((TICKET_WORDS,), dtype=torch.int32, device=device)


def _tickets(device):
    """Fallback for callers that pass no buffer (tests, one-off calls): one buffer per (device, stream)."""
    key = (torch.device(device), int(_lib.current_stream()))
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = new_tickets(device)
    return t


def decode_attention_f16_amax(out_f16, amax, q, k, v, kv_pointers, lengths, tokens_per_block, timestep, rotary_base,
                              single_launch=True, tickets=None):
    """single_query_attention (KV4 + zeros, neox RoPE) writing the fp16 [B, Hq*Dh] output (the values
    single_query_attention returns) into out_f16 and raising the row maxima of |out| in `amax` -- the input pair of
    gemm_partial_f16_*.  single_launch: the last-arriving split workgroup of every (sequence, head group) merges the splits
    inside the attention launch (omni_kv4_decode_attention_f16_amax); False: split partials + the wide merge kernel (two
    launches).  Same bits either way."""
    import ctypes
    from ..rope import rope_table
    _lib.require_cuda(out_f16, amax, q, k, v, kv_pointers, lengths)
    B, Hq, D = q.shape
    Hkv = k.shape[1]
    max_ctx = max(int(timestep), 1)
    table = rope_table(max_ctx + 1, D, float(rotary_base), 1.0, q.device)
    need = _lib.lib().omni_kv4_decode_workspace_bytes(B, Hq, D, max_ctx)
    ws = _lib.workspace(need, q.device, "attn")
    if single_launch:
        tk = tickets if tickets is not None else _tickets(q.device)
        rc = _lib.lib().omni_kv4_decode_attention_f16_amax(
            out_f16.data_ptr(), amax.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0),
            kv_pointers.data_ptr(), lengths.data_ptr(), B, kv_pointers.shape[-1], Hq, Hkv, D, int(tokens_per_block), max_ctx,
            table.data_ptr(), table.shape[0], ws.data_ptr(), ws.numel(), tk.data_ptr(), tk.numel(), _lib.current_stream())
        _lib.check(rc, "fused_ext.decode_attention_f16_amax (single launch)")
        return
    ns = ctypes.c_int(0)
    rc = _lib.lib().omni_kv4_decode_attention_partial(
        q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), kv_pointers.data_ptr(),
        lengths.data_ptr(), B, kv_pointers.shape[-1], Hq, Hkv, D, int(tokens_per_block), max_ctx,
        table.data_ptr(), table.shape[0], ws.data_ptr(), ws.numel(), ctypes.byref(ns), _lib.current_stream())
    _lib.check(rc, "fused_ext.decode_attention_f16_amax (partials)")
    ml_bytes = B * Hq * ns.value * 2 * 4
    rc = _lib.lib().omni_attn_merge_f16_amax(out_f16.data_ptr(), ws.data_ptr(), ws.data_ptr() + ml_bytes, ns.value,
                                             amax.data_ptr(), B, Hq, _lib.current_stream())
    _lib.check(rc, "fused_ext.decode_attention_f16_amax (merge)")


def prefetch_arm_gemm(weight, M, N, K, mode=0, deferred=False, budget_bytes=24 << 20, blocks=240):
    """Arm the one-shot L2 weight prefetch for the NEXT decode-shape GEMM (include/omniserve_hip.h:
    omni_prefetch_arm_gemm): the next decode-size quant / norm row kernel launched through this library carries
    `blocks` extra workgroups that pull up to `budget_bytes` of `weight` into the L2s.  mode: 0 W4A8 per-channel,
    1 per-group, 2 W8A8; | 0x10: the GEMM will run as gemm_silu_* (gate / up tile rows paired per workgroup); | 0x20: that
    GEMM keeps non-temporal weight loads (default: the GEMM enqueued next on `weight` loads plain -- its lines sit in L2;
    every other decode-shape GEMM streams non-temporally: the policy is per call, there is no process-wide switch).
    budget_bytes <= 0 disarms.  A performance hint only."""
    _lib.require_cuda(weight)
    rc = _lib.lib().omni_prefetch_arm_gemm(weight.data_ptr(), int(M), int(N), int(K), int(mode), int(bool(deferred)),
                                           int(budget_bytes), int(blocks))
    _lib.check(rc, "fused_ext.prefetch_arm_gemm")


def prefetch_disarm():
    """Drop an armed (not yet consumed) prefetch descriptor of this thread."""
    rc = _lib.lib().omni_prefetch_arm_gemm(None, 0, 0, 0, 0, 0, 0, 0)
    _lib.check(rc, "fused_ext.prefetch_disarm")


def sparse_decode_attention_quant(out_i8, input_sum, scale, q, k, v, retrieval_kv_pointers, streaming_kv_pointers,
                                  retrieval_head_flags, head_rank_table, dynamic_sparse_page_idxes, lengths,
                                  tokens_per_block, size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                                  local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads,
                                  num_streaming_kv_heads, timestep, rotary_base, rotary_embedding_scale,
                                  tokens_per_sub_chunk, kv_scale_quant_orig=None, kv_scale_orig_quant=None):
    """LServe decode attention (fused_attention_fine_grained_sparse / fused_attention_per_tensor_sparse
    .single_query_attention; per-tensor KV8 when the two scale tensors are given) followed by the per-token
    quantisation of its [B, Hq*Dh] output (fused_kernels.invoke_quant[_fuse_sum]), with the flash-decoding merge done by
    the quantiser: one launch less and no fp16 round trip; codes and scales bit-identical to the two-call sequence."""
    from ._attn_common import decode_attention_fine_grained
    per_tensor = kv_scale_quant_orig is not None
    D = q.shape[-1]
    decode_attention_fine_grained(
        q, k, v, retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags, head_rank_table,
        dynamic_sparse_page_idxes, lengths, tokens_per_block, size_per_retrieval_token, size_per_streaming_token,
        sink_token_num, local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads, num_streaming_kv_heads,
        int(timestep), D, rotary_base, 1.0 / float(rotary_embedding_scale), True, not per_tensor, not per_tensor,
        tokens_per_sub_chunk, "fused_ext.sparse_decode_attention_quant", kv_scale_quant_orig=kv_scale_quant_orig,
        kv_scale_orig_quant=kv_scale_orig_quant, per_tensor=per_tensor, merge_quant=(out_i8, input_sum, scale))


def sparse_decode_attention_f16_amax(out_f16, amax, q, k, v, retrieval_kv_pointers, streaming_kv_pointers,
                                     retrieval_head_flags, head_rank_table, dynamic_sparse_page_idxes, lengths,
                                     tokens_per_block, size_per_retrieval_token, size_per_streaming_token, sink_token_num,
                                     local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads,
                                     num_streaming_kv_heads, timestep, rotary_base, rotary_embedding_scale,
                                     tokens_per_sub_chunk, kv_scale_quant_orig=None, kv_scale_orig_quant=None):
    """LServe decode attention (as sparse_decode_attention_quant) with the merge as the wide kernel of
    decode_attention_f16_amax: out_f16 [B, Hq*Dh] = the values single_query_attention returns, row maxima of |out| raised
    in `amax` -- the input pair of gemm_partial_f16_w8a8 (o_proj quantises on the fly: no quantiser row kernel)."""
    from ._attn_common import decode_attention_fine_grained
    per_tensor = kv_scale_quant_orig is not None
    D = q.shape[-1]
    decode_attention_fine_grained(
        q, k, v, retrieval_kv_pointers, streaming_kv_pointers, retrieval_head_flags, head_rank_table,
        dynamic_sparse_page_idxes, lengths, tokens_per_block, size_per_retrieval_token, size_per_streaming_token,
        sink_token_num, local_token_num, sink_block_num, local_block_num, num_retrieval_kv_heads, num_streaming_kv_heads,
        int(timestep), D, rotary_base, 1.0 / float(rotary_embedding_scale), True, not per_tensor, not per_tensor,
        tokens_per_sub_chunk, "fused_ext.sparse_decode_attention_f16_amax", kv_scale_quant_orig=kv_scale_quant_orig,
        kv_scale_orig_quant=kv_scale_orig_quant, per_tensor=per_tensor, merge_f16=(out_f16, amax))


def embed_rows(out, table, idx):
    """out fp16 [rows, cols] = table[idx] (torch.index_select(table, 0, idx, out=out)): the embedding lookup of the decode
    drivers as one short kernel -- upstream it is torch.nn.Embedding, not one of the reference's kernels."""
    _lib.require_cuda(out, table, idx)
    if table.dtype != torch.float16 or out.dtype != torch.float16 or table.dim() != 2 or not table.is_contiguous() or \
            not out.is_contiguous() or out.shape[-1] != table.shape[1]:
        raise RuntimeError("embed_rows: contiguous fp16 table [V, cols] and out [rows, cols] expected")
    if idx.dtype != torch.int64 or not idx.is_contiguous() or idx.numel() != out.numel() // table.shape[1]:
        raise RuntimeError("embed_rows: idx must be a contiguous int64 tensor with one id per output row")
    rc = _lib.lib().omni_gather_rows_f16(out.data_ptr(), table.data_ptr(), idx.data_ptr(), idx.numel(), table.shape[1],
                                         table.shape[0], _lib.current_stream())
    _lib.check(rc, "fused_ext.embed_rows")


def argmax(out, logits):
    """out int64 [rows] = torch.argmax(logits fp16 [rows, cols], dim=-1) (first maximum); greedy-sampling helper of the
    decode runner -- the reference's sampler is torch code, this is not one of its kernels."""
    _lib.require_cuda(out, logits)
    if logits.dtype != torch.float16 or logits.dim() != 2 or logits.stride(1) != 1:
        raise RuntimeError("argmax: logits must be fp16 [rows, cols] with contiguous rows")
    if out.dtype != torch.int64 or out.numel() != logits.shape[0] or not out.is_contiguous():
        raise RuntimeError("argmax: out must be a contiguous int64 [rows] tensor")
    need = _lib.lib().omni_argmax_workspace_bytes(logits.shape[0])
    ws = _lib.workspace(need, logits.device, "argmax")
    rc = _lib.lib().omni_argmax_f16(out.data_ptr(), logits.data_ptr(), logits.stride(0), logits.shape[0],
                                    logits.shape[1], ws.data_ptr(), ws.numel(), _lib.current_stream())
    _lib.check(rc, "fused_ext.argmax")


def select_topk_pages(out, scores, subs_per_page, total_pages, k):
    """out int32 [B, Hq, k+1] <- the k best pages of [0, total_pages-1) by max-over-sub-chunks selector score (descending),
    then the newest page: one kernel for the torch view / max / topk / cat / to(int32) sequence of
    decoding_attention.py:132-142.  scores fp16 [B, Hq, padded_sub_chunks] as single_query_page_selector returns them."""
    _lib.require_cuda(out, scores)
    if scores.dtype != torch.float16 or scores.dim() != 3 or not scores.is_contiguous():
        raise RuntimeError("select_topk_pages: scores must be a contiguous fp16 [B, Hq, sub_chunks] tensor")
    B, H, _ = scores.shape
    if out.dtype != torch.int32 or tuple(out.shape) != (B, H, k + 1) or not out.is_contiguous():
        raise RuntimeError("select_topk_pages: out must be a contiguous int32 [B, Hq, k+1] tensor")
    rc = _lib.lib().omni_select_topk_pages(out.data_ptr(), scores.data_ptr(), scores.stride(1), B * H,
                                           int(subs_per_page), int(total_pages), int(k), _lib.current_stream())
    _lib.check(rc, "fused_ext.select_topk_pages")


# ---- round 6: (norm -> GEMV) pairs as one launch (csrc/norm_gemv_fused.h) -------------------------------------------
NGF_SYNC_WORDS = 32     # per call site: arrival counter + the rows' {scale, sum} pairs; zeroed by the caller before every launch


def norm_gemm_fused_ok(M, N, K, group_size=-1, silu=False):
    """True when omni_w4a8_per_*_norm_gemm_fused takes the shape and its whole grid is resident on this device."""
    return _lib.lib().omni_norm_gemm_fused_ok(int(M), int(N), int(K), 0 if group_size == -1 else 1, 1 if silu else 0) == 1


def norm_gemm_fused(codes, residual, gamma, input_sum, scaling, epsilon, lin, out, sync, err, slab=None, sk=0, delta=None,
                    producer=None, p_ascales=None, p_asums=None, amax=None, clk=None):
    """rows: residual (+= epilogue(sum of `sk` split-K slabs of `producer`) | += delta | as is) -> rms_norm_general[_fuse_sum] ->
    codes / scaling / input_sum; tiles: out = lin(codes) (amax given: out = silu_and_mul(lin(codes)) [M, N/2], row maxima
    raised) -- ONE launch, bit-identical to the two (three) reference calls.  `lin` / `producer`: objects with qweight,
    s1_scales and s1_szeros (per-channel) or s2_zeros / s2_scales (per-group).  `sync`: NGF_SYNC_WORDS zeroed uint32 words."""
    _lib.require_cuda(codes, residual, gamma, input_sum, scaling, out, sync, err, slab, delta, amax, clk)
    M, K = residual.shape
    N = lin.qweight.shape[0]
    per_chn = getattr(lin, "group", -1) == -1
    st = _lib.current_stream()
    if slab is not None:
        p_ws = producer.s1_scales.data_ptr()
        p_wsz = producer.s1_szeros.data_ptr() if per_chn else None
    else:
        p_ws = p_wsz = None
    if per_chn:
        rc = _lib.lib().omni_w4a8_per_chn_norm_gemm_fused(
            codes.data_ptr(), residual.data_ptr(), _ptr(slab), int(sk), _ptr(delta), p_ws, _ptr(p_ascales), p_wsz,
            _ptr(p_asums), gamma.data_ptr(), _ptr(input_sum), scaling.data_ptr(), float(epsilon), lin.qweight.data_ptr(),
            lin.s1_scales.data_ptr(), lin.s1_szeros.data_ptr(), out.data_ptr(), out.stride(0), _ptr(amax), sync.data_ptr(),
            err.data_ptr(), M, N, K, _ptr(clk), st)
    else:
        rc = _lib.lib().omni_w4a8_per_group_norm_gemm_fused(
            codes.data_ptr(), residual.data_ptr(), _ptr(slab), int(sk), _ptr(delta), p_ws, _ptr(p_ascales), gamma.data_ptr(),
            _ptr(input_sum), scaling.data_ptr(), float(epsilon), lin.qweight.data_ptr(), lin.s2_zeros.data_ptr(),
            lin.s2_scales.data_ptr(), lin.s1_scales.data_ptr(), out.data_ptr(), out.stride(0), _ptr(amax), sync.data_ptr(),
            err.data_ptr(), M, N, K, _ptr(clk), st)
    _lib.check(rc, "fused_ext.norm_gemm_fused")
