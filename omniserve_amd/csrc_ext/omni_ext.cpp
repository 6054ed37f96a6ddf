// Thin pybind11 binding of the hot omniserve_backend.* functions straight onto the C ABI (include/omniserve_hip.h).
//
// The reference binds its kernels with pybind11 / torch extensions (kernels/setup.py:156-333; e.g.
// kernels/csrc/qgemm/w4a8_per_chn/gemm_cuda.h:16 `void gemm_forward_cuda(torch::Tensor ...)`); the zero-change route of this repo
// is a ctypes mirror, whose ~7 us of Python per call is what bounds an UNMODIFIED eager host (bench.py `drop_in`, ~360 calls per
// decode step).  This module gives the mirror functions that a decoder layer calls every step a C++ body: tensor -> pointer
// marshalling, the same argument checks and error type (RuntimeError), torch's current stream, the C-ABI call.  No kernel, no
// arithmetic and no second implementation lives here; omniserve_amd/backend/*.py route to it when it is importable and keep
// the ctypes path otherwise (and for every overload not listed here).  Host-only C++: built by torch.utils.cpp_extension.
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <cstdint>
#include <map>
#include <tuple>
#include <string>

#include "omniserve_hip.h"

namespace {

inline void* stream_of(const torch::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

inline void fail(int rc, const char* what) {
  if (rc == 0) return;
  const char* why = rc == -22 ? "invalid argument" : rc == -12 ? "workspace too small" : rc == -5 ? "kernel launch failed" : "error";
  throw std::runtime_error(std::string(what) + " failed: " + why + " (" + std::to_string(rc) + ")");
}

inline void need_cuda(const torch::Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), what, ": device tensor expected (this library has no CPU path)");
}

// split-K scratch: one persistent buffer per device, grown geometrically (a superseded buffer stays alive: a captured HIP graph may
// have its address baked in), its size per (M, N, K) cached -- the mirror's policy (omniserve_amd/_lib.py: workspace)
struct Scratch { torch::Tensor buf; std::vector<torch::Tensor> retired; };
std::map<int, Scratch> g_scratch;
std::map<std::tuple<int, int, int>, size_t> g_ws_bytes;

inline torch::Tensor& gemm_scratch(int M, int N, int K, const torch::Tensor& like, bool refresh = false) {
  const auto key = std::make_tuple(M, N, K);
  auto it = g_ws_bytes.find(key);
  size_t need = (refresh || it == g_ws_bytes.end()) ? (g_ws_bytes[key] = std::max<size_t>(omni_gemm_workspace_bytes(M, N, K), 1)) : it->second;
  Scratch& s = g_scratch[like.get_device()];
  if (!s.buf.defined() || (size_t)s.buf.numel() < need) {
    size_t size = std::max<size_t>(need, (size_t)1 << 20);
    if (s.buf.defined()) { size = std::max<size_t>(size, 2 * (size_t)s.buf.numel()); s.retired.push_back(s.buf); }
    s.buf = torch::empty({(int64_t)size}, torch::TensorOptions().dtype(torch::kUInt8).device(like.device()));
  }
  return s.buf;
}

struct GemmIO { int M, N, K; int64_t stride; };
inline GemmIO check_gemm_io(const torch::Tensor& in, const torch::Tensor& w, const torch::Tensor& out, bool packed) {
  need_cuda(in, "gemm_forward"); need_cuda(w, "gemm_forward"); need_cuda(out, "gemm_forward");
  TORCH_CHECK(in.scalar_type() == torch::kInt8 && w.scalar_type() == torch::kInt8 && out.scalar_type() == torch::kFloat16,
              "gemm_forward: expected int8 activations/weights and fp16 output");
  TORCH_CHECK(in.dim() == 2 && in.is_contiguous() && w.is_contiguous(), "gemm_forward: in_feats [M,K] and kernel must be contiguous");
  const int64_t M = in.size(0), K = in.size(1), N = out.size(-1);
  TORCH_CHECK(out.dim() >= 2 && out.size(-2) == M, "gemm_forward: out_feats rows != in_feats rows");
  TORCH_CHECK(w.size(0) == N && w.size(1) == (packed ? K / 2 : K), "gemm_forward: weight shape does not match N=", N, " K=", K);
  TORCH_CHECK(out.stride(-1) == 1, "gemm_forward: out_feats rows must be contiguous");
  return {(int)M, (int)N, (int)K, out.stride(-2)};
}

// the call with the cached scratch size; OMNI_ENOMEM (a plan override raised the split since the size was cached) -> re-size once
template <class F>
inline int with_scratch(const GemmIO& g, const torch::Tensor& like, F call) {
  torch::Tensor& ws = gemm_scratch(g.M, g.N, g.K, like);
  int rc = call(ws.data_ptr(), (size_t)ws.numel());
  if (rc == -12) {
    torch::Tensor& ws2 = gemm_scratch(g.M, g.N, g.K, like, true);
    rc = call(ws2.data_ptr(), (size_t)ws2.numel());
  }
  return rc;
}

void gemm_w4a8_per_chn(const torch::Tensor& in, const torch::Tensor& w, const torch::Tensor& wscales, const torch::Tensor& ascales,
                       const torch::Tensor& w_szs, const torch::Tensor& a_ssums, torch::Tensor out) {
  const GemmIO g = check_gemm_io(in, w, out, true);
  need_cuda(wscales, "gemm_forward"); need_cuda(ascales, "gemm_forward"); need_cuda(w_szs, "gemm_forward"); need_cuda(a_ssums, "gemm_forward");
  fail(with_scratch(g, in, [&](void* ws, size_t n) {
         return omni_w4a8_per_chn_gemm(in.data_ptr(), w.data_ptr(), wscales.data_ptr(), ascales.data_ptr(), w_szs.data_ptr(), a_ssums.data_ptr(),
                                       out.data_ptr(), g.M, g.N, g.K, g.stride, ws, n, stream_of(in)); }),
       "qgemm_w4a8_per_chn.gemm_forward_cuda");
}

void gemm_w4a8_per_group(const torch::Tensor& in, const torch::Tensor& w, const torch::Tensor& zeros, const torch::Tensor& scales_i8,
                         const torch::Tensor& wscales, const torch::Tensor& ascales, torch::Tensor out) {
  const GemmIO g = check_gemm_io(in, w, out, true);
  need_cuda(zeros, "gemm_forward"); need_cuda(scales_i8, "gemm_forward"); need_cuda(wscales, "gemm_forward"); need_cuda(ascales, "gemm_forward");
  TORCH_CHECK(zeros.dim() == 2 && zeros.size(0) == g.K / 128 && zeros.size(1) == g.N && scales_i8.dim() == 2 &&
              scales_i8.size(0) == g.K / 128 && scales_i8.size(1) == g.N, "per-group gemm: zeros/scales_i8 must be [K/128, N]");
  fail(with_scratch(g, in, [&](void* ws, size_t n) {
         return omni_w4a8_per_group_gemm(in.data_ptr(), w.data_ptr(), zeros.data_ptr(), scales_i8.data_ptr(), wscales.data_ptr(), ascales.data_ptr(),
                                         out.data_ptr(), g.M, g.N, g.K, g.stride, ws, n, stream_of(in)); }),
       "qgemm_w4a8_per_group.gemm_forward_cuda");
}

void gemm_w8a8(const torch::Tensor& in, const torch::Tensor& w, const torch::Tensor& wscales, const torch::Tensor& ascales, torch::Tensor out) {
  const GemmIO g = check_gemm_io(in, w, out, false);
  need_cuda(wscales, "gemm_forward"); need_cuda(ascales, "gemm_forward");
  fail(with_scratch(g, in, [&](void* ws, size_t n) {
         return omni_w8a8_gemm(in.data_ptr(), w.data_ptr(), wscales.data_ptr(), ascales.data_ptr(), out.data_ptr(), g.M, g.N, g.K, g.stride,
                               ws, n, stream_of(in)); }),
       "qgemm_w8a8.w8a8_gemm_forward_cuda");
}

// fp16 row kernels (the other element types and the static-scale overloads stay on the ctypes mirror)
inline std::pair<int, int> rows_of(const torch::Tensor& input) {
  const int64_t hidden = input.size(-1);
  return {(int)(hidden ? input.numel() / hidden : 0), (int)hidden};
}

void rms_norm_general_fuse_sum_f16(torch::Tensor out, const torch::Tensor& input, const torch::Tensor& weight, torch::Tensor input_sum,
                                   torch::Tensor scaling, double eps) {
  need_cuda(out, "rms_norm_general_fuse_sum"); need_cuda(input, "rms_norm_general_fuse_sum"); need_cuda(weight, "rms_norm_general_fuse_sum");
  need_cuda(input_sum, "rms_norm_general_fuse_sum"); need_cuda(scaling, "rms_norm_general_fuse_sum");
  const auto r = rows_of(input);
  fail(omni_rms_norm_general_fuse_sum(out.data_ptr(), input.data_ptr(), weight.data_ptr(), input_sum.data_ptr(), scaling.data_ptr(), (float)eps,
                                      r.first, r.second, stream_of(input)),
       "layernorm_ops.rms_norm_general_fuse_sum");
}

void rms_norm_general_f16(torch::Tensor out, const torch::Tensor& input, const torch::Tensor& weight, torch::Tensor scaling, double eps) {
  need_cuda(out, "rms_norm_general"); need_cuda(input, "rms_norm_general"); need_cuda(weight, "rms_norm_general"); need_cuda(scaling, "rms_norm_general");
  const auto r = rows_of(input);
  fail(omni_rms_norm_general(out.data_ptr(), input.data_ptr(), weight.data_ptr(), scaling.data_ptr(), (float)eps, r.first, r.second, stream_of(input)),
       "layernorm_ops.rms_norm_general");
}

void rms_norm_f16(torch::Tensor out, const torch::Tensor& input, const torch::Tensor& weight, double eps) {
  need_cuda(out, "rms_norm"); need_cuda(input, "rms_norm"); need_cuda(weight, "rms_norm");
  const auto r = rows_of(input);
  fail(omni_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), (float)eps, r.first, r.second, stream_of(input)), "layernorm_ops.rms_norm");
}

void quant_fuse_sum_f16(torch::Tensor out, const torch::Tensor& input, torch::Tensor input_sum, torch::Tensor scale) {
  need_cuda(out, "invoke_quant_fuse_sum"); need_cuda(input, "invoke_quant_fuse_sum"); need_cuda(input_sum, "invoke_quant_fuse_sum"); need_cuda(scale, "invoke_quant_fuse_sum");
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "invoke_quant: tensors must be contiguous");
  const auto r = rows_of(input);
  if (r.first == 0) return;
  fail(omni_quant_fuse_sum(out.data_ptr(), input.data_ptr(), input_sum.data_ptr(), scale.data_ptr(), r.first, r.second, stream_of(input)),
       "fused_kernels.invoke_quant_fuse_sum");
}

void quant_f16(torch::Tensor out, const torch::Tensor& input, torch::Tensor scale) {
  need_cuda(out, "invoke_quant"); need_cuda(input, "invoke_quant"); need_cuda(scale, "invoke_quant");
  TORCH_CHECK(input.is_contiguous() && out.is_contiguous(), "invoke_quant: tensors must be contiguous");
  const auto r = rows_of(input);
  if (r.first == 0) return;
  fail(omni_quant(out.data_ptr(), input.data_ptr(), scale.data_ptr(), r.first, r.second, stream_of(input)), "fused_kernels.invoke_quant");
}

void silu_and_mul_f16(torch::Tensor out, const torch::Tensor& input) {
  need_cuda(out, "silu_and_mul"); need_cuda(input, "silu_and_mul");
  const int64_t d2 = input.size(-1);
  const int tokens = (int)(d2 ? input.numel() / d2 : 0);
  fail(omni_silu_and_mul(out.data_ptr(), input.data_ptr(), tokens, (int)(d2 / 2), stream_of(input)), "activation_ops.silu_and_mul");
}

// fused_attention_pure_dense.single_query_attention (KV4 pages with zero points, neox RoPE): the callee allocates the fp16 [B,Hq,Dh]
// result, as the reference does.  `table` / `ws`: the host-built RoPE table and the split-KV scratch the mirror caches per device.
torch::Tensor decode_attention_kv4(const torch::Tensor& q, const torch::Tensor& k, const torch::Tensor& v, const torch::Tensor& kv_pointers,
                                   const torch::Tensor& lengths, int64_t tokens_per_block, int64_t size_per_token, int64_t max_ctx,
                                   const torch::Tensor& table, torch::Tensor ws, const std::string& what) {
  const char* w = what.c_str();
  need_cuda(q, w); need_cuda(k, w); need_cuda(v, w); need_cuda(kv_pointers, w); need_cuda(lengths, w);
  TORCH_CHECK(q.dim() == 3 && k.dim() == 3 && v.dim() == 3, w, ": q / k / v must be [B,H,D]");
  const int64_t B = q.size(0), Hq = q.size(1), D = q.size(2), Hkv = k.size(1);
  TORCH_CHECK(size_per_token == Hkv * D / 2, w, ": size_per_token ", size_per_token, " != Hkv*Dh/2");
  TORCH_CHECK(q.scalar_type() == torch::kFloat16 && q.stride(2) == 1 && q.stride(1) == D, w, ": q must be fp16 [B,H,D] with contiguous heads");
  TORCH_CHECK(k.stride(2) == 1 && k.stride(1) == D && v.stride(2) == 1 && v.stride(1) == D, w, ": k/v heads must be contiguous");
  TORCH_CHECK(k.stride(0) == v.stride(0), w, ": k and v must share the row stride");
  TORCH_CHECK(lengths.scalar_type() == torch::kInt32 && kv_pointers.is_contiguous(), w, ": lengths must be int32, kv_pointers contiguous");
  torch::Tensor out = torch::empty({B, Hq, D}, q.options());
  fail(omni_kv4_decode_attention(out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), q.stride(0), k.stride(0), kv_pointers.data_ptr(),
                                 lengths.data_ptr(), (int)B, (int)kv_pointers.size(-1), (int)Hq, (int)Hkv, (int)D, (int)tokens_per_block,
                                 (int)max_ctx, table.data_ptr(), (int)table.size(0), ws.data_ptr(), (size_t)ws.numel(), stream_of(q)),
       w);
  return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "pybind11 fast path of the omniserve_backend mirror onto libomniserve_hip.so's C ABI";
  m.def("abi_version", []() { return omni_abi_version(); });
  m.def("gemm_w4a8_per_chn", &gemm_w4a8_per_chn);
  m.def("gemm_w4a8_per_group", &gemm_w4a8_per_group);
  m.def("gemm_w8a8", &gemm_w8a8);
  m.def("rms_norm_general_fuse_sum_f16", &rms_norm_general_fuse_sum_f16);
  m.def("rms_norm_general_f16", &rms_norm_general_f16);
  m.def("rms_norm_f16", &rms_norm_f16);
  m.def("quant_fuse_sum_f16", &quant_fuse_sum_f16);
  m.def("quant_f16", &quant_f16);
  m.def("silu_and_mul_f16", &silu_and_mul_f16);
  m.def("decode_attention_kv4", &decode_attention_kv4);
}
