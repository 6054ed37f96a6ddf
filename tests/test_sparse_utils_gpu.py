"""LServe helpers: paged min/max pooling (bit-exact) and the page selector (<= 2 fp16 ulp: fp16 products,
f32 vs f64 accumulation order) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import kv4
from tests.util import dev, f16_ulp_diff, to_dev

pytestmark = pytest.mark.gpu
D, TPB, SUB = 128, 64, 16


def _setup(seq_lens, Hin, pooled, seed):
    rng = np.random.default_rng(seed)
    B = len(seq_lens)
    H = len(pooled)
    pages = (max(seq_lens) + TPB) // TPB + 1
    n = B * pages
    table = rng.permutation(n).reshape(B, pages)
    pb = kv4.stats_page_bytes(H, D, TPB, SUB)
    pool = np.full((n, pb), 0x11, np.uint8)
    L = int(sum(seq_lens))
    k = rng.standard_normal((L, Hin, D)).astype(np.float16)
    cu = np.concatenate([[0], np.cumsum(seq_lens)]).astype(np.int32)
    return k, cu, pool, table, pb


@pytest.mark.parametrize("seq_lens,Hin,pooled", [([5], 2, [1]), ([64, 17, 130], 8, [0, 3, 4, 7]), ([200], 4, [0, 1, 2, 3]),
                                                  ([2500, 70], 2, [1])])   # > 32 pages: several selector workgroups per head
def test_paged_min_max_pool_and_selector(seq_lens, Hin, pooled):
    import omniserve_backend.fused_attention_ctx_pool as cp
    import omniserve_backend.fused_attention_selector as sel
    k, cu, pool, table, pb = _setup(seq_lens, Hin, pooled, seed=sum(seq_lens))
    H = len(pooled)
    pool_d = to_dev(pool)
    B, M = table.shape
    ptr = np.zeros((B, 2, M), np.int64)
    ptr[:, 0, :] = pool_d.data_ptr() + table.astype(np.int64) * pb
    ptr[:, 1, :] = ptr[:, 0, :]
    ptr_d = to_dev(ptr)
    kv4.paged_min_max_pool(k, cu, pooled, pool, table, TPB, SUB)
    cp.paged_min_max_pool(to_dev(k), ptr_d, to_dev(cu), to_dev(np.asarray(pooled, np.int32)), max(seq_lens), SUB, TPB,
                          H * D // 2, True)
    torch.cuda.synchronize()
    assert np.array_equal(pool_d.cpu().numpy(), pool), "page tails differ"

    # selector on the pooled statistics: q heads = 2 per kv head
    Hkv, g = Hin, 2
    Hq = Hkv * g
    flags = np.zeros(Hkv, np.int32); flags[pooled] = 1
    rank = np.zeros(Hkv, np.int32); rank[pooled] = np.arange(H); rank[flags == 0] = np.arange(Hkv - H)
    lengths = (np.asarray(seq_lens) + 1).astype(np.int32)
    q = np.random.default_rng(1).standard_normal((B, Hq, D)).astype(np.float16)
    want = kv4.page_selector(q, lengths, flags, rank, pool, table, Hkv, H, TPB, SUB, 500000.0)
    dummy = torch.empty((B, Hkv, D), dtype=torch.float16, device=dev())
    got = sel.single_query_page_selector(to_dev(q), dummy, dummy, ptr_d, None, to_dev(flags), to_dev(rank), None,
                                         to_dev(lengths), None, 65536, TPB, H * D // 2, 0, 0, 0, 0, 0, H, Hkv - H,
                                         int(max(seq_lens)), D, 500000.0, 1.0, True, True, True, SUB, H * D, 1000000)
    torch.cuda.synchronize()
    assert tuple(got.shape) == want.shape
    assert f16_ulp_diff(got, want) <= 2


@pytest.mark.parametrize("B,H,pages,subs,k", [(1, 32, 4001, 4, 63), (2, 8, 12, 4, 2), (1, 4, 300, 2, 299), (3, 2, 5000, 1, 1024)])
def test_select_topk_pages_matches_torch(B, H, pages, subs, k):
    """fused_ext.select_topk_pages == the torch sequence of decoding_attention.py:132-142 (max over sub-chunks, topk of
    all pages but the newest, newest page appended); ties may be broken differently, so the chosen SCORES are compared."""
    from omniserve_amd.backend import fused_ext
    g = torch.Generator(device="cpu").manual_seed(pages + k)
    scores = torch.randn((B, H, pages * subs), generator=g).half()
    scores[0, 0, : 40 * subs] = 1.5                      # a block of ties
    if H > 1:
        scores[0, 1].zero_()                             # a streaming head: all zero
    sd = scores.to(dev())
    out = torch.empty((B, H, k + 1), dtype=torch.int32, device=dev())
    fused_ext.select_topk_pages(out, sd, subs, pages, k)
    torch.cuda.synchronize()
    page_scores = scores.view(B, H, pages, subs).max(dim=-1).values
    want_vals, _ = page_scores[:, :, : pages - 1].float().topk(k=k, dim=-1)
    got = out.cpu().long()
    assert (got[:, :, k] == pages - 1).all()
    sel = got[:, :, :k]
    assert (sel >= 0).all() and (sel < pages - 1).all()
    assert (sel.sort(dim=-1).values.diff(dim=-1) > 0).all(), "duplicate pages"
    got_vals = torch.gather(page_scores.float(), 2, sel)
    assert torch.equal(got_vals, want_vals), "not the top-k scores in descending order"
    # ties go to the lower page index: the 1.5-block of row (0, 0) comes out in ascending page order
    tied = [int(p) for p, v in zip(sel[0, 0].tolist(), got_vals[0, 0].tolist()) if v == 1.5]
    assert tied == sorted(tied)
