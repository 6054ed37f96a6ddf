"""omniserve_amd.persistent.PersistentActivationBuffer hands out what the reference's layers expect to find in
`input_metadata.activation_buffer`: pinned against the reference's own class where /root/reference is present
(omniserve/utils/input_metadata.py:18-112), and self-consistent everywhere."""
import types

import pytest
import torch

from omniserve_amd.persistent import PersistentActivationBuffer
from tests import refstack

HID, INTER, QS, KVS = 256, 512, 256, 128


def _tensors(obj):
    return {k: v for k, v in vars(obj).items() if isinstance(v, torch.Tensor)}


def test_views_are_slices_of_one_allocation_and_alias_like_upstream():
    pab = PersistentActivationBuffer(HID, INTER, QS, KVS, max_tokens=300, chunk_prefill_size=128, device="cpu")
    seen = None
    for T in (1, 5, 128, 300):
        v = pab.view_for(T)
        assert v.qkv_proj_act_buffer.shape == (T, QS + 2 * KVS) and v.out_down_proj_act_buffer.shape == (T, HID)
        assert v.qkv_proj_act_buffer.data_ptr() == v.out_down_proj_act_buffer.data_ptr() == v.act_buffer.data_ptr()
        assert v.gate_up_proj_act_buffer.shape == (min(128, T), 2 * INTER)
        assert v.quantized_mlp_act_buffer.shape == (min(128, T), INTER) and v.quantized_mlp_act_buffer.dtype == torch.int8
        assert v.quantized_hidden_states_buffer.shape == (T, HID) and v.quantized_scale_buffer.shape == (T,)
        ptrs = tuple(t.data_ptr() for t in (v.act_buffer, v.gate_up_proj_act_buffer, v.quantized_act_buffer,
                                            v.quantized_mlp_act_buffer, v.quantized_scale_buffer, v.quantized_sum_buffer))
        assert seen is None or ptrs == seen, "a step view must not allocate"
        seen = ptrs
        assert all(t.is_contiguous() for t in _tensors(v).values())
    with pytest.raises(ValueError):
        pab.view_for(301)
    with pytest.raises(ValueError):
        pab.view_for(0)


@pytest.mark.skipif(not refstack.reference_available(), reason="/root/reference is not on this machine")
@pytest.mark.parametrize("T", [1, 7, 128, 300])
def test_view_matches_the_reference_activation_buffer(T):
    with refstack.reference_over_mirror():
        from omniserve.utils.input_metadata import ActivationBuffer

        class LlamaForCausalLM:       # the class name is what upstream dispatches on (input_metadata.py:31-35)
            pass

        model = LlamaForCausalLM()
        model.model = types.SimpleNamespace(embed_tokens=types.SimpleNamespace(weight=torch.zeros((4, HID), dtype=torch.float16)))
        model.model_config = types.SimpleNamespace(chunk_prefill_size=128)
        model.q_size, model.kv_size = QS, KVS
        model.config = types.SimpleNamespace(intermediate_size=INTER, hidden_size=HID)
        ref = ActivationBuffer(model, T)
        ref.allocate_activation_buffer()
    ours = PersistentActivationBuffer(HID, INTER, QS, KVS, max_tokens=300, chunk_prefill_size=128, device="cpu").view_for(T)
    rt, ot = _tensors(ref), _tensors(ours)
    assert set(rt) == set(ot), (sorted(rt), sorted(ot))
    for name, t in rt.items():
        o = ot[name]
        assert tuple(o.shape) == tuple(t.shape) and o.dtype == t.dtype and o.stride() == t.stride(), name
    # aliasing relations inside each object are the same (offset from the fp16 / int8 base allocations)
    for base, names in (("act_buffer", ["qkv_proj_act_buffer", "out_down_proj_act_buffer"]),
                        ("quantized_act_buffer", ["quantized_hidden_states_buffer"])):
        for n in names:
            assert rt[n].data_ptr() - rt[base].data_ptr() == ot[n].data_ptr() - ot[base].data_ptr() == 0, n
    for name in ("batched_seq_len", "hidden_size", "intermediate_size", "q_size", "kv_size", "chunk_prefill_size"):
        assert getattr(ref, name) == getattr(ours, name), name
