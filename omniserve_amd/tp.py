"""Tensor-parallel sharding of the QServe W4A8 path (BASELINE.json configs[4]: Llama-2-70B TP=8).

The reference has no TP implementation (tp_size is hard-coded to 1,
llama_w4a8_unpad.py:158,529,607-608); this module is the MI355X-native design:

  * one process per GPU (torchrun), `torch.distributed` backend "nccl" = RCCL over xGMI;
  * Megatron split: qkv_proj / gate_up_proj column-parallel (shard the output channels N),
    o_proj / down_proj row-parallel (shard the reduction dim K), attention by KV head;
  * the only collective: an in-place fp16 sum all-reduce of the [M, hidden] projection output
    after o_proj and after down_proj (2 per layer).  The per-channel zero-point term distributes
    over K shards (each rank subtracts szero[n] * sum_{k in shard} x[m,k]) and the per-token
    activation scale may be rank-local, so row-parallel needs no other exchange.

Shard granularity: N in multiples of 64 channels (the kernels' wave tile; covers the 32-row pack
blocks and the 32-channel permutation blocks of the g128 parameters), K in whole 32-k tiles
(multiples of 128 so that g128 groups stay whole).  A K shard is NOT a column slice of the
[N, K/2] view: it is a slice of the [N/32][K/32][512 B] tile view.
"""
from __future__ import annotations

import torch


def shard_range(total: int, rank: int, world: int, align: int):
    if total % (world * align) != 0:
        raise ValueError("dimension %d cannot be split %d-ways in multiples of %d" % (total, world, align))
    per = total // world
    return rank * per, (rank + 1) * per


def shard_qweight_n(qweight: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Column-parallel shard of packed weights [N, K/2]: a row slice (N/world multiple of 64)."""
    n0, n1 = shard_range(qweight.shape[0], rank, world, 64)
    return qweight[n0:n1].contiguous()


def shard_qweight_k(qweight: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Row-parallel shard of packed weights: slice the k-tile axis of the [N/32][K/32][512] view."""
    N, K2 = qweight.shape
    K = 2 * K2
    k0, k1 = shard_range(K, rank, world, 128)
    tiles = qweight.reshape(N // 32, K // 32, 512)
    return tiles[:, k0 // 32:k1 // 32].contiguous().reshape(N, (k1 - k0) // 2)


def shard_vec_n(v: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    n0, n1 = shard_range(v.shape[-1], rank, world, 64)
    return v[..., n0:n1].contiguous()


def shard_group_params_k(p: torch.Tensor, rank: int, world: int, group: int = 128) -> torch.Tensor:
    """g128 second-level params [K/G, N] for a row-parallel (K) shard: a row slice."""
    g0, g1 = shard_range(p.shape[0] * group, rank, world, group)
    return p[g0 // group:g1 // group].contiguous()


def all_reduce_(buf: torch.Tensor, group=None) -> torch.Tensor:
    """In-place fp16 sum all-reduce of a projection output (RCCL on GPU, gloo in the CPU tests)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        if buf.is_cuda and dist.get_backend(group) == "gloo":
            # test rigs only (several ranks on one GPU): gloo reduces on the host.  fp32 staging: the sum of
            # `world` fp16 values rounded once, as a 2-rank fp16 ring all-reduce would produce.
            host = buf.float().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            buf.copy_(host.to(buf.dtype))
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


class PeerComm:
    """The library's own tensor-parallel collective over peer-mapped device memory (include/omniserve_hip.h: omni_tp_*;
    csrc/tp_comm.h).  torch.distributed is used ONCE, to exchange the 64-byte IPC handles; after that a collective is one
    kernel launch on the current stream (HIP-graph capturable): no RCCL call, no host synchronisation.

        comm = PeerComm(rank, world, max_elems, device, group)     # collective: every rank constructs it
        out_view = comm.slot(numel)              # fp16 tensor aliasing this rank's slot of the NEXT call: the projection
                                                 # writes its partial result straight into it
        comm.all_reduce(out)                     # out <- sum over ranks           (consumes the slot, flips to the other)
        comm.add_rms_norm(...)                   # or: residual += sum; norm; quant (one launch)

    Every rank must issue the same sequence of collectives.  A rank that waits for a lost peer sets an error word instead
    of hanging (check_error())."""

    FLAG_WORDS = 64

    class _Blob:      # raw device memory as a torch tensor (torch.as_tensor understands __cuda_array_interface__)
        def __init__(self, ptr, nbytes):
            self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}

    def __init__(self, rank, world, max_elems, device, group=None, loopback=False, algo="auto"):
        """loopback=True (single process, measurements only): every "peer" is this rank's own buffer, so a collective of
        `world` ranks runs with all flags raised by the caller itself -- the kernels' cost without any fabric traffic.
        algo: "auto" (two shots -- reduce-scatter into a gather region, then all-gather -- from 4 MiB of payload on more
        than two ranks; one shot otherwise), "one_shot", "two_shot" (csrc/tp_comm.h; same bits either way)."""
        import ctypes
        import torch.distributed as dist
        from . import _lib
        if world > 8:
            raise ValueError("PeerComm supports up to 8 ranks")
        self.loopback = bool(loopback)
        self.rank, self.world, self.device = int(rank), int(world), torch.device(device)
        self.algo = {"auto": 0, "one_shot": 1, "two_shot": 2}[algo]
        if loopback and self.algo == 0:
            self.algo = 1      # (loopback aliases every peer's gather region to this rank's: one shot only)
        self.slot_elems = (int(max_elems) + 7) // 8 * 8
        # [slot 0 | slot 1 | gather region (two-shot form: this rank's reduced chunk; a whole slot keeps any row split legal) | flags]
        self.gather_off = 2 * self.slot_elems
        self.data_bytes = 3 * self.slot_elems * 2
        self.nbytes = self.data_bytes + 4 * self.FLAG_WORDS
        lib = _lib.lib()
        with torch.cuda.device(self.device):
            p = ctypes.c_void_p()
            _lib.check(lib.omni_tp_alloc(self.nbytes, ctypes.byref(p)), "omni_tp_alloc")
            self._own = p.value
            h = (ctypes.c_ubyte * 64)()
            _lib.check(lib.omni_tp_ipc_handle(self._own, h), "omni_tp_ipc_handle")
            handles = [None] * world
            if world > 1 and not loopback:
                dist.all_gather_object(handles, bytes(h), group=group)
            else:
                handles[0] = bytes(h)
            self._mapped = []
            bases = []
            for r in range(world):
                if r == rank or loopback:
                    bases.append(self._own)
                    continue
                q = ctypes.c_void_p()
                buf = (ctypes.c_ubyte * 64).from_buffer_copy(handles[r])
                _lib.check(lib.omni_tp_ipc_open(buf, ctypes.byref(q)), "omni_tp_ipc_open (rank %d)" % r)
                self._mapped.append(q.value)
                bases.append(q.value)
        self._data = (ctypes.c_void_p * world)(*bases)
        if loopback:     # flag word [rank] of "peer" p = word p of the own array: the caller's publishing threads raise them all
            self._flags = (ctypes.c_void_p * world)(*[self._own + self.data_bytes + 4 * (p - rank) for p in range(world)])
        else:
            self._flags = (ctypes.c_void_p * world)(*[b + self.data_bytes for b in bases])
        self._mine = torch.as_tensor(self._Blob(self._own, self.nbytes), device=self.device)
        self._call = 0
        if world > 1 and not loopback:
            dist.barrier(group=group)      # nobody publishes before every rank has mapped every buffer

    def slot(self, numel, shape=None):
        """fp16 view of this rank's slot for the NEXT collective (the projection's output tensor)."""
        if numel > self.slot_elems:
            raise ValueError("PeerComm: %d elements exceed the slot (%d)" % (numel, self.slot_elems))
        off = (self._call & 1) * self.slot_elems * 2
        t = self._mine[off: off + 2 * numel].view(torch.float16)
        return t.view(shape) if shape is not None else t

    def _next_slot_off(self):
        off = (self._call & 1) * self.slot_elems
        self._call += 1
        return off

    def all_reduce(self, out):
        """out (fp16, contiguous, numel % 8 == 0) <- sum over ranks of their current slots."""
        from . import _lib
        n = out.numel()
        rc = _lib.lib().omni_tp_allreduce_f16(out.data_ptr(), self._data, self._flags, self.rank, self.world,
                                              self._next_slot_off(), n, self.gather_off, self.algo, _lib.current_stream())
        _lib.check(rc, "omni_tp_allreduce_f16")

    def add_rms_norm(self, out_i8, residual, weight, input_sum, scaling, epsilon):
        """residual += sum over ranks of their slots; rms_norm_general[_fuse_sum](out_i8, residual, ...) (input_sum None:
        no row sum) -- the all-reduce folded into the norm kernel."""
        from . import _lib
        hidden = residual.shape[-1]
        tokens = residual.numel() // hidden
        rc = _lib.lib().omni_tp_add_rms_norm_general_fuse_sum(
            out_i8.data_ptr(), residual.data_ptr(), self._data, self._flags, self.rank, self.world, self._next_slot_off(),
            weight.data_ptr(), None if input_sum is None else input_sum.data_ptr(), scaling.data_ptr(), float(epsilon),
            tokens, hidden, self.gather_off, self.algo, _lib.current_stream())
        _lib.check(rc, "omni_tp_add_rms_norm_general_fuse_sum")

    def check_error(self, clear=False):
        """Raises if a wait for a peer timed out since the last clear (device word TP_W_ERROR; reading it synchronises).
        The collective that timed out produced NaN and did not advance the epoch (csrc/tp_comm.h)."""
        words = self._mine[self.data_bytes:].view(torch.int32)
        if int(words[18].item()) != 0:
            if clear:
                words[18] = 0
            raise RuntimeError("PeerComm: a wait for a peer timed out (rank %d); the affected collectives returned NaN" % self.rank)

    def resync(self):
        """Call after a step was abandoned half way (e.g. a failed graph capture): the slot parity is a host-side counter
        baked into captured launches, and every rank must restart a step on the same (even) parity."""
        if self._call & 1:
            self._call += 1

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001  (interpreter shutdown: the library may already be gone)
            pass

    def close(self):
        from . import _lib
        lib = _lib.lib()
        for q in getattr(self, "_mapped", []):
            lib.omni_tp_ipc_close(q)
        self._mapped = []
        if getattr(self, "_own", None):
            lib.omni_tp_free(self._own)
            self._own = None
