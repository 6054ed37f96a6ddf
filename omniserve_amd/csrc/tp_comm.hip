// Peer-mapped communication buffers and the plain all-reduce of the tensor-parallel decode path (see tp_comm.h).
#include "tp_comm.h"

using namespace omni;

namespace omni {

__global__ __launch_bounds__(256) void tp_allreduce_f16_kernel(half_t* __restrict__ out, TpPeers tp, long long count) {
  const uint32_t e = tp_publish_and_wait(tp);
  const long long nvec = count / 8;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256)
    *reinterpret_cast<v8h*>(out + v * 8) = tp_sum8(tp, (size_t)v * 8, e);
  tp_finish(tp, e);
}

// two-shot form (tp_comm.h): shot 1 = my chunk reduced into my gather region, shot 2 = all chunks from their owners
__global__ __launch_bounds__(256) void tp_allreduce2_f16_kernel(half_t* __restrict__ out, TpPeers tp, long long count) {
  const uint32_t e = tp_publish_and_wait(tp);
  const long long c0 = (long long)tp.rank * tp.chunk;
  const long long mine_n = c0 >= count ? 0 : (count - c0 < tp.chunk ? count - c0 : tp.chunk);     // elements of my chunk
  half_t* gather = const_cast<half_t*>(tp_data_of(tp, tp.rank)) + tp.gather_off;
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < mine_n / 8; v += (long long)gridDim.x * 256)
    *reinterpret_cast<v8h*>(gather + v * 8) = tp_sum8(tp, (size_t)(c0 + v * 8), e);
  const bool ok = tp_between_shots(tp, e);
  const half_t qnan = __builtin_bit_cast(half_t, (uint16_t)0x7E00);
  for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < count / 8; v += (long long)gridDim.x * 256) {
    const long long i = v * 8;
    const int owner = (int)(i / tp.chunk);
    v8h x = *reinterpret_cast<const v8h*>(tp_data_of(tp, owner) + tp.gather_off + (i - (long long)owner * tp.chunk));
    if (!ok) x = (v8h){qnan, qnan, qnan, qnan, qnan, qnan, qnan, qnan};
    *reinterpret_cast<v8h*>(out + i) = x;
  }
  tp_finish(tp, ok ? e : 0u);
}

}  // namespace omni

// algo: 0 = by payload, 1 = one shot, 2 = two shots (needs a gather region).  By payload: two shots from TP_TWO_SHOT_BYTES on more
// than two ranks.  The second in-kernel rendezvous (drain + one system release per workgroup, ticket, publish, poll, acquire)
// measures +16 us per collective (bench.py's TP leg, loopback: 13.6 -> 29.7 us); per link the two-shot form saves
// payload x (1 - 2 / world) / 153 GB/s: the forms cross at ~3.3 MB on 8 ranks, ~4.9 MB on 4 -- the decode step's 2 MiB
// (bs = 128) stays one-shot, larger payloads go two-shot.
static bool want_two_shot(int algo, int world, long long count, long long gather_off) {
  if (gather_off < 0 || algo == 1) return false;
  if (algo == 2) return true;
  return world > 2 && count * 2 >= omni::TP_TWO_SHOT_BYTES;
}

static int fill_peers(TpPeers& tp, const void* const* peer_data, void* const* peer_flags, int rank, int world,
                      long long slot_off) {
  if (!peer_data || !peer_flags || world < 1 || world > TP_MAX_WORLD || rank < 0 || rank >= world || slot_off < 0)
    return OMNI_EINVAL;
  tp.gather_off = 0; tp.chunk = 0; tp.two_shot = 0;
  for (int p = 0; p < TP_MAX_WORLD; ++p) {
    const int q = p < world ? p : 0;
    if (!peer_data[q] || !peer_flags[q]) return OMNI_EINVAL;
    tp.data[p] = static_cast<const half_t*>(peer_data[q]);
    tp.flags[p] = static_cast<uint32_t*>(peer_flags[q]);
  }
  tp.rank = rank; tp.world = world; tp.slot_off = slot_off; tp.epoch = 1;
  return OMNI_OK;
}

// ---- buffers: fine-grained device memory (coherent with peers that map it), shared through hipIpc handles ------------
extern "C" int omni_tp_alloc(size_t bytes, void** ptr_out) {
  if (!ptr_out || bytes == 0) return OMNI_EINVAL;
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) return OMNI_ENOMEM;
  if (hipMemset(p, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return OMNI_ELAUNCH; }
  *ptr_out = p;
  return OMNI_OK;
}
extern "C" int omni_tp_free(void* ptr) { return hipFree(ptr) == hipSuccess ? OMNI_OK : OMNI_EINVAL; }
extern "C" int omni_tp_ipc_handle(void* ptr, void* handle64) {
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "handle size");
  if (!ptr || !handle64) return OMNI_EINVAL;
  return hipIpcGetMemHandle(static_cast<hipIpcMemHandle_t*>(handle64), ptr) == hipSuccess ? OMNI_OK : OMNI_EINVAL;
}
extern "C" int omni_tp_ipc_open(const void* handle64, void** ptr_out) {
  if (!handle64 || !ptr_out) return OMNI_EINVAL;
  hipIpcMemHandle_t h = *static_cast<const hipIpcMemHandle_t*>(handle64);
  return hipIpcOpenMemHandle(ptr_out, h, hipIpcMemLazyEnablePeerAccess) == hipSuccess ? OMNI_OK : OMNI_EINVAL;
}
extern "C" int omni_tp_ipc_close(void* ptr) { return hipIpcCloseMemHandle(ptr) == hipSuccess ? OMNI_OK : OMNI_EINVAL; }

// out fp16 [count] = sum over the ranks of their slot (count % 8 == 0).  peer_data / peer_flags: host arrays of `world`
// device pointers (rank p's buffers as mapped into this process; entry `rank` = the caller's own).  Only enqueues.
// gather_offset_elems: element offset of the rank's gather region inside its data buffer (>= ceil(count / world) elements rounded
// up to 8; < 0: none, one-shot only); algo: 0 by payload, 1 one shot, 2 two shots.
extern "C" int omni_tp_allreduce_f16(void* out_f16, const void* const* peer_data, void* const* peer_flags, int rank,
                                     int world, long long slot_offset_elems, long long count, long long gather_offset_elems,
                                     int algo, void* stream) {
  if (!out_f16 || count < 0 || count % 8 != 0 || algo < 0 || algo > 2 || (algo == 2 && gather_offset_elems < 0)) return OMNI_EINVAL;
  TpPeers tp;
  const int rc = fill_peers(tp, peer_data, peer_flags, rank, world, slot_offset_elems);
  if (rc != OMNI_OK) return rc;
  if (count == 0) return OMNI_OK;
  long long wgs = (count / 8 + 255) / 256;
  if (wgs > 128) wgs = 128;                 // all workgroups must be resident together (they wait on each other's ticket)
  if (want_two_shot(algo, world, count, gather_offset_elems)) {
    tp.two_shot = 1;
    tp.gather_off = gather_offset_elems;
    tp.chunk = ((count / 8 + world - 1) / world) * 8;
    hipLaunchKernelGGL(tp_allreduce2_f16_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (half_t*)out_f16, tp,
                       count);
    return omni_launch_status();
  }
  hipLaunchKernelGGL(tp_allreduce_f16_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (half_t*)out_f16, tp,
                     count);
  return omni_launch_status();
}
