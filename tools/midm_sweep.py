"""Mid-M GEMM sweep (M = 33 .. 128): the single-wave-tile plans against w4a8_midm_kernel (qgemm_midm.h) at each K split,
on the Llama-3-8B, Llama-2-70B TP = 8 shard and Llama-2-70B TP = 1 projection shapes; weights rotated over > MALL copies,
HIP events on torch's current stream (the stream the mirrors launch on).  Every mid-M result is compared bit for bit with the
legacy plan's output (which tests/test_gemm_gpu.py pins to the oracle).
Usage (GPU box): python tools/midm_sweep.py [--quick] > gpurun_out/midm_sweep.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib  # noqa: E402

if os.environ.get("OMNI_TUNE_LIB", ""):     # A/B against a library variant (tools/build_variant.sh)
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.backend import _gemm_common, fused_ext, qgemm_w4a8_per_chn, qgemm_w4a8_per_group, qgemm_w8a8  # noqa: E402

_lib.USE_EXT = False      # plan overrides change scratch sizes per shape: stay on the ctypes mirror, which this tool re-sizes
dev = torch.device("cuda:0")
lib = _lib.lib()
quick = "--quick" in sys.argv or "--big" in sys.argv or "--one" in sys.argv
SHAPES = {
    "8b": [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)],
    "70b_tp8": [(1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584)],
    "70b_tp1": [(10240, 8192), (8192, 8192), (57344, 8192), (8192, 28672)],
}
HBM, INT8 = 8000.0, 3944.0       # GB/s (datasheet), TOP/s (micro-benchmark ceiling, MI355X_MICROARCH.md)


def timed(fn, copies):
    for i in range(copies):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 4 * copies
    s.record()
    for i in range(iters):
        fn(i % copies)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def run(model, M, N, K, mode):
    wbytes = N * K if mode == "w8" else N * K // 2
    copies = max(2, min(16, int(600e6 // wbytes)))
    if mode == "w8":
        ws = [torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev) for _ in range(copies)]
    else:
        ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8) for _ in range(copies)]
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    sw = (torch.rand((N,), device=dev) * 0.01 + 0.002).half(); sz = (sw.float() * 7).half()
    sa = (torch.rand((M,), device=dev) * 0.01 + 0.002).half(); asum = (torch.randn((M,), device=dev) * 3).half()
    s2s = torch.randint(1, 9, (K // 128, N), dtype=torch.int8, device=dev)
    s2z = torch.randint(-100, 1, (K // 128, N), dtype=torch.int8, device=dev)
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    slab = torch.empty((16 * M * N,), dtype=torch.int32, device=dev)
    alg = M * K + wbytes + 2 * M * N + 4 * N + 4 * M
    ops = 2.0 * M * N * K

    def full(i):
        if mode == "chn":
            qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i], sw, sa, sz, asum, out)
        elif mode == "grp":
            qgemm_w4a8_per_group.gemm_forward_cuda(a, ws[i], s2z, s2s, sw, sa, out)
        else:
            qgemm_w8a8.w8a8_gemm_forward_cuda(a, ws[i], sw, sa, out)

    def partial(i):
        if mode == "chn":
            return fused_ext.gemm_partial_per_chn(a, ws[i], slab)
        if mode == "grp":
            return fused_ext.gemm_partial_per_group(a, ws[i], s2z, s2s, slab)
        return fused_ext.gemm_partial_w8a8(a, ws[i], slab)

    lib.omni_gemm_set_midm_override(0, 0)
    full(0)
    torch.cuda.synchronize()
    ref = out.clone()
    sk0 = partial(0)
    torch.cuda.synchronize()
    ref_acc = slab[: sk0 * M * N].view(sk0, M, N).sum(dim=0)
    t_full = timed(full, copies)
    t_part = timed(partial, copies)
    print("%-8s %-3s M=%3d N=%5d K=%5d legacy      : full %7.2f us  slab-only %7.2f us (sk %d)   [stream %5.2f us, mfma %5.2f us]"
          % (model, mode, M, N, K, t_full, t_part, sk0, alg / HBM / 1e3, ops / INT8 / 1e6), flush=True)
    variants = [(1, 0)] if quick else [(1, 0), (1, 1), (1, 2), (1, 4), (1, 8)]
    for mode_bits, sk in variants:
        if sk and (K % (sk * 256) or K // sk < 512):
            continue
        lib.omni_gemm_set_midm_override(mode_bits, sk)
        _gemm_common._ws_bytes.clear()
        _lib.workspace(16 * M * N * 4, dev, "gemm")
        out.zero_()
        full(0)
        skm = partial(0)
        torch.cuda.synchronize()
        ok = torch.equal(out, ref)
        ok2 = torch.equal(slab[: skm * M * N].view(skm, M, N).sum(dim=0), ref_acc)
        t_full = timed(full, copies)
        t_part = timed(partial, copies)
        print("%-8s %-3s M=%3d N=%5d K=%5d midm m%d sk=%-4s: full %7.2f us  slab-only %7.2f us (sk %d)   %5.2f TB/s %6.0f TOPS  %s"
              % (model, mode, M, N, K, mode_bits, sk if sk else "auto", t_full, t_part, skm, alg / t_full / 1e6, ops / t_full / 1e6,
                 "bit-equal" if ok and ok2 else "MISMATCH full=%s slab=%s" % (ok, ok2)), flush=True)
    lib.omni_gemm_set_midm_override(-1, 0)
    del ws


if __name__ == "__main__":
    if "--one" in sys.argv:
        quick = True
        run("8b", 128, 28672, 4096, "chn")
        run("8b", 64, 28672, 4096, "chn")
        run("70b_tp1", 128, 8192, 28672, "chn")
        sys.exit(0)
    if "--big" in sys.argv:      # the shapes the mid-M kernel is for (one line per plan: legacy, mid-M at its own K split)
        quick = True
        for model, N, K in (("8b", 28672, 4096), ("8b", 4096, 14336), ("70b_tp8", 7168, 8192), ("70b_tp1", 57344, 8192),
                            ("70b_tp1", 8192, 28672), ("70b_tp1", 10240, 8192)):
            for M in (128, 64):
                run(model, M, N, K, "chn")
        run("8b", 64, 28672, 4096, "grp")
        run("8b", 64, 4096, 14336, "grp")
        run("8b", 128, 28672, 4096, "w8")
        sys.exit(0)
    for model, shapes in SHAPES.items():
        for (N, K) in shapes:
            for M in (128, 64):
                run(model, M, N, K, "chn")
    for (N, K) in SHAPES["8b"]:
        for M in (64, 128):
            run("8b", M, N, K, "grp")
    for (N, K) in SHAPES["8b"][2:]:
        run("8b", 128, N, K, "w8")
