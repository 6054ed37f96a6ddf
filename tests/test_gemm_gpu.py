"""Parity of the HIP W4A8 / W8A8 GEMMs against the oracle: bit-exact fp16 outputs
(int32 accumulators exact, epilogue evaluated in the same fp32 order)."""
import numpy as np
import pytest
import torch

from oracle import w4a8
from tests.util import assert_f16_equal, dev, quantize_act, to_dev

pytestmark = pytest.mark.gpu


def _acts(M, K, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((M, K)).astype(np.float16)
    return quantize_act(x)


def _run_chn(M, N, K, seed=0, out_view=False):
    import omniserve_backend.qgemm_w4a8_per_chn as mod
    u, z, s1 = w4a8.synth_per_channel(N, K, seed)
    qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
    a, sa, asum = _acts(M, K, seed + 1)
    want = w4a8.gemm_per_chn(a, qw, s1h, sa, szh, asum)
    if out_view == "cols":  # rows that start 8 bytes into a wider buffer (row stride not a multiple of 16 bytes: the 8-byte store forms)
        buf = torch.full((M, N + 4), 7.0, dtype=torch.float16, device=dev())
        out = buf[:, 4:]
    elif out_view:  # pre-allocated larger buffer, row-slice view (llama_w4a8_unpad.py:356-361)
        buf = torch.full((M + 3, N), 7.0, dtype=torch.float16, device=dev())
        out = buf[1:M + 1]
    else:
        out = torch.empty((M, N), dtype=torch.float16, device=dev())
    mod.gemm_forward_cuda(to_dev(a), to_dev(qw), to_dev(s1h), to_dev(sa), to_dev(szh), to_dev(asum), out)
    torch.cuda.synchronize()
    assert_f16_equal(out, want, "per_chn M=%d N=%d K=%d" % (M, N, K))
    if out_view == "cols":
        assert (buf[:, :4] == 7).all()
    elif out_view:
        assert (buf[0] == 7).all() and (buf[M + 1:] == 7).all()


@pytest.mark.parametrize("M", [1, 7, 16, 17, 33, 64, 100, 128])
@pytest.mark.parametrize("N,K", [(64, 64), (256, 256), (320, 448), (192, 1024)])
def test_per_chn_decode_shapes(M, N, K):
    _run_chn(M, N, K, seed=M + N + K)


@pytest.mark.parametrize("M,N,K", [(129, 256, 256), (300, 320, 192), (1000, 512, 1024), (257, 64, 64),
                                   (2100, 2304, 128)])   # last: 17 x 9 tiles = several 8 x 8 super-blocks (XCD order)
def test_per_chn_prefill_shapes(M, N, K):
    _run_chn(M, N, K, seed=M)


# Decode at batch 129 .. 512 (the reference's published A100 figure is quoted at bs = 256): few 128 x 256 tiles, so the plan
# splits K over grid.y into int32 slabs + the slab epilogue (qgemm_plan.hip: down_proj 8 ways, o_proj / qkv 4 ways at
# bs = 256; gate_up has 224 tiles and stays whole).  Ragged M (row tiles partly filled), all three flavours.
FEW_TILE_SHAPES = [(256, 4096, 14336), (256, 6144, 4096), (256, 4096, 4096), (129, 4096, 14336), (300, 512, 2048), (512, 4096, 4096)]


@pytest.mark.parametrize("M,N,K", FEW_TILE_SHAPES)
def test_per_chn_few_tiles_split_k(M, N, K):
    import ctypes
    from omniserve_amd import _lib
    sk = ctypes.c_int(0)
    _lib.lib().omni_gemm_get_plan(M, N, K, 64, None, None, ctypes.byref(sk))
    assert sk.value > 1 and _lib.lib().omni_gemm_workspace_bytes(M, N, K) >= sk.value * M * N * 4
    _run_chn(M, N, K, seed=M + N)


@pytest.mark.parametrize("M,N,K", [(256, 4096, 14336), (256, 4096, 4096), (384, 512, 2048), (300, 512, 2048)])
def test_per_chn_partial_few_tiles(M, N, K):
    """The slab-only form (fused_ext.gemm_partial_per_chn: o_proj / down_proj of the decode drivers at batch 129 .. 512): the
    sum of the int32 slabs is the exact accumulator."""
    from omniserve_amd.backend import fused_ext
    u, z, s1 = w4a8.synth_per_channel(N, K, M)
    qw, _, _ = w4a8.pack_per_channel(u, z, s1)
    a, _, _ = _acts(M, K, M + 1)
    slab = torch.zeros((8, M, N), dtype=torch.int32, device=dev())
    sk = fused_ext.gemm_partial_per_chn(to_dev(a), to_dev(qw), slab)
    torch.cuda.synchronize()
    assert sk > 1 and not slab[sk:].any()
    assert np.array_equal(slab[:sk].sum(dim=0).cpu().numpy(), w4a8.gemm_per_chn_acc(a, qw))


@pytest.mark.parametrize("M,N,K", FEW_TILE_SHAPES[:3] + FEW_TILE_SHAPES[4:5])
def test_per_group_few_tiles_split_k(M, N, K):
    _run_grp(M, N, K, False)


@pytest.mark.parametrize("M,N,K", FEW_TILE_SHAPES[1:3] + FEW_TILE_SHAPES[4:5])
def test_w8a8_few_tiles_split_k(M, N, K):
    _run_w8(M, N, K)


@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)])
def test_per_chn_llama3_8b_decode_bs16(N, K):
    _run_chn(16, N, K, seed=1)


def test_per_chn_output_row_slice_view():
    _run_chn(16, 256, 512, seed=9, out_view=True)


def test_per_chn_config1_4096_cubed():
    _run_chn(4096, 4096, 4096, seed=0)


@pytest.mark.parametrize("waves,sk", [(1, 1), (1, 4), (4, 1), (4, 2), (4, 8)])
def test_per_chn_plan_overrides(waves, sk):
    from omniserve_amd import _lib
    _lib.lib().omni_gemm_set_plan_override(waves, sk)
    try:
        _run_chn(16, 512, 2048, seed=waves * 10 + sk)
        _run_chn(32, 512, 2048, seed=waves * 10 + sk + 1)
    finally:
        _lib.lib().omni_gemm_set_plan_override(0, 0)


def _run_grp(M, N, K, wrap, seed=None):
    import omniserve_backend.qgemm_w4a8_per_group as mod
    u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=M + K if seed is None else seed, wrap=wrap)
    qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
    a, sa, _ = _acts(M, K, M + 2)
    want = w4a8.gemm_per_group(a, qw, s2z, s2s, s1h, sa)
    out = torch.empty((M, N), dtype=torch.float16, device=dev())
    mod.gemm_forward_cuda(to_dev(a), to_dev(qw), to_dev(s2z), to_dev(s2s), to_dev(s1h), to_dev(sa), out)
    torch.cuda.synchronize()
    assert_f16_equal(out, want, "per_group M=%d N=%d K=%d wrap=%s" % (M, N, K, wrap))


def _run_w8(M, N, K):
    import omniserve_backend.qgemm_w8a8 as mod
    rng = np.random.default_rng(M + N)
    w = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    sw = rng.uniform(0.001, 0.01, size=(N,)).astype(np.float16)
    a, sa, _ = _acts(M, K, M + 3)
    want = w4a8.gemm_w8a8(a, w, sw, sa)
    out = torch.empty((M, N), dtype=torch.float16, device=dev())
    mod.w8a8_gemm_forward_cuda(to_dev(a), to_dev(w), to_dev(sw), to_dev(sa), out)
    torch.cuda.synchronize()
    assert_f16_equal(out, want, "w8a8 M=%d N=%d K=%d" % (M, N, K))


# ---- every M regime of the tile dispatch has an oracle case (the reference's: w4a8_per_group/gemm_cuda.cu:659-705) ----
# Prefill kernel (M > 128), K >= 3 whole 256-k chunks: the branch-free STEADY loop of w4a8_gemm_kernel (two-step weight
# ring refills, next-chunk group parameters, in-chunk activation publishing) with full and ragged M tiles and a ragged
# N tile (N = 320: five 64-channel groups = 1.25 workgroup tiles).
@pytest.mark.parametrize("wrap", [False, True])
@pytest.mark.parametrize("M,N,K", [(300, 256, 1024), (1000, 512, 1024), (256, 320, 2048), (1000, 512, 4096)])
def test_per_group_prefill_steady_loop(M, N, K, wrap):
    _run_grp(M, N, K, wrap)


@pytest.mark.parametrize("M,N,K", [(300, 256, 1024), (1000, 512, 1024), (256, 320, 2048), (1000, 512, 4096)])
def test_w8a8_prefill_steady_loop(M, N, K):
    _run_w8(M, N, K)


@pytest.mark.parametrize("M,N,K", [(1000, 512, 1024), (300, 320, 2304)])
def test_per_chn_prefill_steady_loop_ragged(M, N, K):
    _run_chn(M, N, K, seed=M + K)


def test_per_group_4096_cubed():
    _run_grp(4096, 4096, 4096, False)


def test_w8a8_4096_cubed():
    _run_w8(4096, 4096, 4096)


# Exact-shape prefill kernel (csrc/qgemm_exact.h: M % 128 == N % 256 == K % 256 == 0; activations by LDS-DMA into the swizzled
# row image, lane transpose of the packed int4 registers, last chunk without prefetch, packed 16-B stores) against the oracle:
# one tile x one chunk (no steady loop: prologue -> last chunk), two chunks (one steady iteration), several tiles in both
# directions incl. a partly filled 8 x 8 super-block, a long K, and the per-group wrap set (byte products above 255).
EXACT_SHAPES = [(128, 256, 256), (256, 512, 512), (384, 256, 1024), (128, 768, 2048), (1152, 2304, 256), (640, 512, 4096)]


@pytest.mark.parametrize("M,N,K", EXACT_SHAPES)
def test_per_chn_exact_shapes(M, N, K):
    _run_chn(M, N, K, seed=M + N + K)


@pytest.mark.parametrize("wrap", [False, True])
@pytest.mark.parametrize("M,N,K", EXACT_SHAPES)
def test_per_group_exact_shapes(M, N, K, wrap):
    _run_grp(M, N, K, wrap)


@pytest.mark.parametrize("M,N,K", EXACT_SHAPES)
def test_w8a8_exact_shapes(M, N, K):
    _run_w8(M, N, K)


# Llama-2-70B TP=8 shard shapes (BASELINE.json configs[4], bs up to 128): qkv (64+16)/8 heads x 128 = 1280 x 8192,
# o 8192 x 1024, gate_up 2 x 28672/8 = 7168 x 8192, down 8192 x 3584 -- the M = 65..128 decode tile with split-K slabs.
LLAMA2_70B_TP8 = [(1280, 8192), (8192, 1024), (7168, 8192), (8192, 3584)]


@pytest.mark.parametrize("M", [65, 100, 128])
@pytest.mark.parametrize("N,K", LLAMA2_70B_TP8)
def test_per_chn_llama2_70b_tp8_shard(M, N, K):
    _run_chn(M, N, K, seed=M + N)


@pytest.mark.parametrize("M", [65, 128])
@pytest.mark.parametrize("N,K", LLAMA2_70B_TP8)
def test_per_group_llama2_70b_tp8_shard(M, N, K):
    _run_grp(M, N, K, False)


# BASELINE.json configs[2] (g128, bs = 64) and the LServe W8A8 decode shapes at M = 64, Llama-3-8B projections
@pytest.mark.parametrize("N,K", [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)])
def test_per_group_llama3_8b_decode_bs64(N, K):
    _run_grp(64, N, K, False)


@pytest.mark.parametrize("M,N,K", [(64, 28672, 4096), (64, 4096, 14336), (1, 6144, 4096), (1, 28672, 4096)])
def test_w8a8_llama3_8b_decode(M, N, K):
    _run_w8(M, N, K)


@pytest.mark.parametrize("M", [17, 32, 48, 64])
def test_per_chn_llama3_8b_gate_up_mid_batches(M):
    _run_chn(M, 28672, 4096, seed=M)


@pytest.mark.parametrize("wrap", [False, True])
@pytest.mark.parametrize("M,N,K", [(16, 256, 512), (64, 320, 1024), (5, 64, 128), (200, 256, 384), (16, 4096, 4096)])
def test_per_group(M, N, K, wrap):
    import omniserve_backend.qgemm_w4a8_per_group as mod
    u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=M + K, wrap=wrap)
    qw, s1h, s2s, s2z = w4a8.pack_per_group(u, z, s2, s1)
    a, sa, _ = _acts(M, K, M + 2)
    want = w4a8.gemm_per_group(a, qw, s2z, s2s, s1h, sa)
    out = torch.empty((M, N), dtype=torch.float16, device=dev())
    mod.gemm_forward_cuda(to_dev(a), to_dev(qw), to_dev(s2z), to_dev(s2s), to_dev(s1h), to_dev(sa), out)
    torch.cuda.synchronize()
    assert_f16_equal(out, want, "per_group M=%d N=%d K=%d wrap=%s" % (M, N, K, wrap))


@pytest.mark.parametrize("M,N,K", [(16, 256, 512), (64, 320, 1024), (3, 64, 64), (200, 256, 320), (16, 4096, 4096)])
def test_w8a8(M, N, K):
    import omniserve_backend.qgemm_w8a8 as mod
    rng = np.random.default_rng(M + N)
    w = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    sw = rng.uniform(0.001, 0.01, size=(N,)).astype(np.float16)
    a, sa, _ = _acts(M, K, M + 3)
    want = w4a8.gemm_w8a8(a, w, sw, sa)
    out = torch.empty((M, N), dtype=torch.float16, device=dev())
    mod.w8a8_gemm_forward_cuda(to_dev(a), to_dev(w), to_dev(sw), to_dev(sa), out)
    torch.cuda.synchronize()
    assert_f16_equal(out, want, "w8a8 M=%d N=%d K=%d" % (M, N, K))


def test_gemm_linearity_full_size():
    """Size-independent property at the BASELINE size: with ascales=1, asum=0 and wscales=2^-k the
    epilogue is exact, so acc(A1) + acc(A2) == acc(A1 + A2) can be checked through the fp16
    outputs (values kept below 2048 so fp16 holds them exactly)."""
    import omniserve_backend.qgemm_w4a8_per_chn as mod
    M, N, K = 16, 4096, 4096
    rng = np.random.default_rng(5)
    u = rng.integers(0, 2, size=(N, K), dtype=np.uint8)           # codes 0/1
    qw = to_dev(w4a8.pack_w4(u))
    a1 = rng.integers(-1, 2, size=(M, K), dtype=np.int8)
    a2 = rng.integers(-1, 2, size=(M, K), dtype=np.int8)
    ones_n = to_dev(np.ones(N, np.float16)); zeros_n = to_dev(np.zeros(N, np.float16))
    ones_m = to_dev(np.ones(M, np.float16)); zeros_m = to_dev(np.zeros(M, np.float16))
    outs = []
    for a in (a1, a2, (a1 + a2).astype(np.int8)):
        o = torch.empty((M, N), dtype=torch.float16, device=dev())
        mod.gemm_forward_cuda(to_dev(a), qw, ones_n, ones_m, zeros_n, zeros_m, o)
        outs.append(o.float())
    torch.cuda.synchronize()
    assert torch.equal(outs[0] + outs[1], outs[2])
    assert outs[2].abs().max() < 2048


# ---- mid-M kernel (csrc/qgemm_midm.h): forced on through omni_gemm_set_midm_override so that every row-tile height, ragged M,
# short slices (fewer chunks than ring slots), forced K splits and all three flavours meet the oracle -- the planner itself
# only sends it the big matrices (test_midm_planner_takes_the_big_matrices) ----
class _ForcedMidm:
    def __init__(self, sk=0):
        self.sk = sk

    def __enter__(self):
        from omniserve_amd import _lib
        from omniserve_amd.backend import _gemm_common
        _lib.lib().omni_gemm_set_midm_override(1, self.sk)
        _gemm_common._ws_bytes.clear()
        self.keep_ext, _lib.USE_EXT = _lib.USE_EXT, False      # (the fast binding caches scratch sizes per shape as well)

    def __exit__(self, *exc):
        from omniserve_amd import _lib
        from omniserve_amd.backend import _gemm_common
        _lib.lib().omni_gemm_set_midm_override(-1, 0)
        _gemm_common._ws_bytes.clear()
        _lib.USE_EXT = self.keep_ext


MIDM_SMALL = [(33, 128, 256, 0), (64, 256, 512, 0), (65, 128, 1024, 0), (100, 384, 2048, 2), (128, 256, 4096, 4), (48, 128, 768, 0),
              (127, 256, 1280, 0), (64, 128, 2048, 8), (128, 128, 256, 0), (96, 640, 1536, 2)]


@pytest.mark.parametrize("M,N,K,sk", MIDM_SMALL)
def test_midm_forced_per_chn(M, N, K, sk):
    with _ForcedMidm(sk):
        _run_chn(M, N, K, seed=M + K)
        _run_chn(M, N, K, seed=M + K + 1, out_view=True)
        _run_chn(M, N, K, seed=M + K + 2, out_view="cols")


@pytest.mark.parametrize("wrap", [False, True])
@pytest.mark.parametrize("M,N,K,sk", MIDM_SMALL)
def test_midm_forced_per_group(M, N, K, sk, wrap):
    with _ForcedMidm(sk):
        _run_grp(M, N, K, wrap)


@pytest.mark.parametrize("M,N,K,sk", MIDM_SMALL)
def test_midm_forced_w8a8(M, N, K, sk):
    with _ForcedMidm(sk):
        _run_w8(M, N, K)


@pytest.mark.parametrize("M,N,K", [(128, 28672, 4096), (64, 28672, 4096), (128, 4096, 14336), (100, 7168, 8192), (128, 8192, 3584)])
def test_midm_forced_model_shapes(M, N, K):
    with _ForcedMidm():
        _run_chn(M, N, K, seed=N)
        _run_grp(M, N, K, wrap=True)


@pytest.mark.parametrize("M,N,K,sk", [(128, 4096, 14336, 0), (64, 512, 2048, 2), (100, 256, 1024, 1), (128, 8192, 8192, 4)])
def test_midm_forced_partial_slabs(M, N, K, sk):
    """The slab-only forms (o_proj / down_proj of the decode drivers at fusion level 2): the slabs sum to the exact accumulator."""
    from omniserve_amd.backend import fused_ext
    u, z, s1 = w4a8.synth_per_channel(N, K, M)
    qw, _, _ = w4a8.pack_per_channel(u, z, s1)
    a, _, _ = _acts(M, K, M + 1)
    slab = torch.zeros((16, M, N), dtype=torch.int32, device=dev())
    with _ForcedMidm(sk):
        got = fused_ext.gemm_partial_per_chn(to_dev(a), to_dev(qw), slab)
        torch.cuda.synchronize()
    assert got >= 1 and (sk == 0 or got == sk) and not slab[got:].any()
    assert np.array_equal(slab[:got].sum(dim=0).cpu().numpy(), w4a8.gemm_per_chn_acc(a, qw))


def test_midm_planner_takes_the_big_matrices():
    """Unforced: Llama-3-8B gate_up at 64 / 128 rows and the Llama-2-70B projections go to the mid-M kernel (plan: 8 waves per
    workgroup reported as 2 channel groups), Llama-3-8B qkv / o and the TP = 8 shards stay on the single-wave tiles."""
    import ctypes
    from omniserve_amd import _lib
    lib = _lib.lib()

    def waves(M, N, K):
        w = ctypes.c_int(0)
        lib.omni_gemm_get_plan(M, N, K, 64, None, ctypes.byref(w), None)
        return w.value
    for M, N, K in [(128, 28672, 4096), (64, 28672, 4096), (128, 57344, 8192), (128, 8192, 28672), (65, 10240, 8192)]:
        assert waves(M, N, K) == 2, (M, N, K)
    for M, N, K in [(128, 6144, 4096), (128, 4096, 4096), (128, 1280, 8192), (16, 28672, 4096), (32, 28672, 4096), (128, 8192, 1024)]:
        assert waves(M, N, K) == 1, (M, N, K)
    _run_chn(128, 28672, 4096, seed=5)
    _run_chn(64, 28672, 4096, seed=6)
