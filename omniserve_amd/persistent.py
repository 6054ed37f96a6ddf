"""Per-step host overhead as pieces the reference's worker can adopt.

Upstream allocates a fresh `ActivationBuffer` inside every `InputMetadata` (omniserve/utils/input_metadata.py:199-200:
five `torch.empty` per engine step) and launches ~11 kernels per layer eagerly; on an MI355X the decode step is then
host-launch bound (bench.py `drop_in`: 2.9 ms against 2.4 ms captured).  Two small, independent helpers:

* `PersistentActivationBuffer` -- allocated once for the largest step; `view_for(batched_seq_len)` hands out an object
  with exactly the attributes the reference's layers read from `input_metadata.activation_buffer`
  (`llama_w4a8_unpad.py:85-108,271-345,406-438`), as views of the persistent storage and with the reference's own
  aliasing (`qkv_proj_act_buffer` and `out_down_proj_act_buffer` start at the same address, input_metadata.py:63-72).
  Opt-in patch upstream: `self.activation_buffer = model.persistent_buffer.view_for(batched_seq_len)`.
* `GraphedStep` -- captures a no-argument callable that only touches persistent tensors into one HIP graph (eager warm-up
  on a side stream first, so that lazily sized scratch -- the mirror's GEMM / attention workspaces, the RoPE table -- exists
  before capture) and replays it.  Every entry point of libomniserve_hip.so only enqueues on the current stream, so a step
  written against `omniserve_backend.*` is capturable as is.

`tests/test_persistent_cpu.py` pins the view's attribute names / shapes / dtypes / aliasing to the reference's class
(imported from /root/reference where present); `tests/test_persistent_gpu.py` replays a decoder layer written with the
reference's call sequence and checks it against the eager run, with no allocation during replay.
"""
from __future__ import annotations

import types
from typing import Callable, Optional

import torch


class PersistentActivationBuffer:
    """Storage for the activation buffers of the largest engine step; sliced per step (no allocation)."""

    def __init__(self, hidden_size: int, intermediate_size: int, q_size: int, kv_size: int, max_tokens: int,
                 chunk_prefill_size: int, device, dtype=torch.float16):
        if dtype != torch.float16:
            raise ValueError("the QServe path runs the model in fp16 (input_metadata.py:36-38)")
        if max_tokens < 1:
            raise ValueError("max_tokens must be positive")
        self.hidden_size, self.intermediate_size = int(hidden_size), int(intermediate_size)
        self.q_size, self.kv_size = int(q_size), int(kv_size)
        self.max_tokens, self.chunk_prefill_size = int(max_tokens), int(chunk_prefill_size)
        self.device = torch.device(device)
        qkv = self.q_size + 2 * self.kv_size
        chunk = min(self.chunk_prefill_size, self.max_tokens)
        self._act = torch.empty((self.max_tokens * max(qkv, 2 * self.intermediate_size),), dtype=dtype, device=self.device)
        self._gate_up = torch.empty((chunk, 2 * self.intermediate_size), dtype=dtype, device=self.device)
        self._quant = torch.empty((self.max_tokens * self.hidden_size,), dtype=torch.int8, device=self.device)
        self._quant_mlp = torch.empty((chunk, self.intermediate_size), dtype=torch.int8, device=self.device)
        self._scale = torch.empty((self.max_tokens,), dtype=dtype, device=self.device)
        self._sum = torch.empty((self.max_tokens,), dtype=dtype, device=self.device)

    def view_for(self, batched_seq_len: int):
        """The attributes of the reference's ActivationBuffer after allocate_activation_buffer(), for this step."""
        T = int(batched_seq_len)
        if T < 1 or T > self.max_tokens:
            raise ValueError("batched_seq_len %d does not fit the persistent buffer (max %d)" % (T, self.max_tokens))
        qkv = self.q_size + 2 * self.kv_size
        chunk = min(self.chunk_prefill_size, T)
        v = types.SimpleNamespace()
        v.batched_seq_len, v.hidden_size, v.intermediate_size = T, self.hidden_size, self.intermediate_size
        v.q_size, v.kv_size, v.chunk_prefill_size, v.device, v.model_dtype = (self.q_size, self.kv_size,
                                                                           self.chunk_prefill_size, self.device, torch.float16)
        v.act_buffer = self._act[: T * max(qkv, 2 * self.intermediate_size)]
        v.qkv_proj_act_buffer = v.act_buffer[: T * qkv].view(T, qkv)
        v.out_down_proj_act_buffer = v.act_buffer[: T * self.hidden_size].view(T, self.hidden_size)
        v.gate_up_proj_act_buffer = self._gate_up[:chunk]
        v.quantized_act_buffer = self._quant[: T * self.hidden_size]
        v.quantized_hidden_states_buffer = v.quantized_act_buffer.view(T, self.hidden_size)
        v.quantized_mlp_act_buffer = self._quant_mlp[:chunk]
        v.quantized_scale_buffer = self._scale[:T]
        v.quantized_sum_buffer = self._sum[:T]
        return v


class GraphedStep:
    """One HIP graph around a step function whose inputs and outputs live in persistent device tensors."""

    def __init__(self, fn: Callable[[], None], device=None, warmup: int = 1):
        self.fn = fn
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.warmup = int(warmup)
        self.graph: Optional[torch.cuda.CUDAGraph] = None

    def capture(self, restore: Optional[Callable[[], None]] = None) -> None:
        """Warm up eagerly (sizes scratch), optionally roll state back with `restore`, then capture."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU (there is no CPU fallback)")
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self.fn()
            if restore is not None:
                restore()
        torch.cuda.current_stream(self.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.fn()
        if restore is not None:
            restore()
        self.graph = graph

    def run(self) -> None:
        if self.graph is None:
            # A stateful step (lengths += 1, KV append, token feedback) cannot be warmed up without a `restore`: the
            # warm-up launches would advance it before the first replay.  Capture without warm-up instead (the capture
            # pass itself executes nothing), then replay: exactly one execution.  Scratch that is sized lazily must
            # therefore already exist -- call capture(restore) explicitly when it may not.
            if self.warmup > 0:
                raise RuntimeError("GraphedStep.run() before capture(): call capture(restore) first (warm-up launches "
                                   "advance a stateful step), or construct with warmup=0")
            self.capture()
        self.graph.replay()
