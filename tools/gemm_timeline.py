"""Per-workgroup timeline of the exact prefill GEMM (debug build: tools/build_variant.sh clk -DOMNI_DEBUG_CLOCKS,
OMNI_TUNE_LIB=tune_libs/libclk.so).  Marks (100 MHz wall clock): entry, first chunk, after the K loop, after the stores."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omniserve_amd import _lib
if os.environ.get("OMNI_TUNE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["OMNI_TUNE_LIB"])
from omniserve_amd.backend import qgemm_w4a8_per_chn
dev = torch.device("cuda:0")
lib = _lib.lib()
f = lib.omni_debug_timeline_gemm_chn; f.restype = ctypes.c_int; f.argtypes = [ctypes.c_void_p, ctypes.c_int]
for (M, N, K) in [(4096, 4096, 4096), (16384, 6144, 4096), (16384, 4096, 14336)]:
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=dev).view(torch.int8)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=dev); sz = sw.clone()
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=dev); asum = sa.clone()
    out = torch.empty((M, N), dtype=torch.float16, device=dev)
    for _ in range(3):
        qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out)
    torch.cuda.synchronize()
    nwg = min(8192, ((M // 128 + 7) // 8) * ((N // 256 + 7) // 8) * 64)
    buf = (ctypes.c_ulonglong * (nwg * 5))()
    assert f(buf, nwg) == 0
    raw = np.array(list(buf), dtype=np.uint64).reshape(nwg, 5)
    hw = raw[:, 4]
    t = raw[:, :4].astype(np.float64) / 100.0   # us
    keep = t[:, 3] > 0
    t, hw, wids = t[keep], hw[keep], np.arange(nwg)[keep]
    t0 = t[:, 0].min()
    pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print("M=%d N=%d K=%d: %d workgroups, span %.1f us | entry %.1f..%.1f | prologue %.2f (%.2f..%.2f) | K loop %.2f (%.2f..%.2f) "
          "| write-back issue %.2f (%.2f..%.2f) | last exit %.1f" % (
              M, N, K, len(t), t[:, 3].max() - t0, 0.0, t[:, 0].max() - t0, pro.mean(), pro.min(), pro.max(), loop.mean(), loop.min(),
              loop.max(), epi.mean(), epi.min(), epi.max(), t[:, 3].max() - t0))
    # rounds: histogram of entry times
    h, edges = np.histogram(t[:, 0] - t0, bins=12)
    print("   entry histogram:", " ".join("%d@%.0f" % (c, e) for c, e in zip(h, edges[:-1])))

    # who shares a CU with whom: (xcc, se, sh?, cu) from HW_REG_HW_ID (cu_id bits 11:8, sh_id 12, se_id 15:13), loop time vs dispatch order
    cu = ((hw >> np.uint64(32)) << np.uint64(8)) | ((hw >> np.uint64(8)) & np.uint64(0xFF))
    h2, e2 = np.histogram(loop, bins=10)
    print("   K-loop histogram (us):", " ".join("%d@%.1f" % (c, e) for c, e in zip(h2, e2[:-1])))
    groups = {}
    for i in range(len(t)):
        groups.setdefault(int(cu[i]), []).append((int(wids[i]), float(t[i, 0] - t0), float(loop[i])))
    sizes = {}
    for g in groups.values():
        sizes[len(g)] = sizes.get(len(g), 0) + 1
    print("   CUs seen: %d, workgroups per CU: %s" % (len(groups), sorted(sizes.items())))
    if M == 4096:
        first_faster = 0; pairs = 0; d = []
        for g in groups.values():
            if len(g) == 2:
                g.sort()
                pairs += 1; first_faster += g[0][2] < g[1][2]; d.append(g[1][2] - g[0][2])
        if pairs:
            print("   pairs %d: lower workgroup id faster in %d; mean (second - first) K-loop time %.2f us, |diff| mean %.2f" % (
                pairs, first_faster, float(np.mean(d)), float(np.mean(np.abs(d)))))
        for k in list(groups)[:4]:
            print("   cu %x:" % k, groups[k])
        print("   wave-0 slots (HW_ID & 15) histogram:", np.bincount((hw & np.uint64(15)).astype(np.int64), minlength=8).tolist(),
              " SIMD:", np.bincount(((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64), minlength=4).tolist())
