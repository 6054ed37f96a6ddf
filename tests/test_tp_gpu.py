"""Tensor-parallel decode (SURVEY 8 row e): two ranks (gloo rendezvous on 127.0.0.1, both on cuda:0 -- the test
box has one GPU; the sharding, the kernels and the all-reduce placement are what is under test) hold the
column / row shards of the SAME synthetic model and must reproduce the single-GPU hidden state up to the
noise of the rank-local int8 activation scales of the row-parallel inputs."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BATCH = 4


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_rank(rank, world, port, group_size, steps, ret, tp_comm=None, graph=False, fused=1, qkv_slabs="auto"):
    import torch.distributed as dist
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = LlamaConfig.tiny()
        cfg.group_size = group_size
        r = DecodeRunner(cfg, BATCH, 0, steps + 2, torch.device("cuda:0"), seed=5, use_graph=graph, fused=fused,
                         tp_rank=rank, tp_size=world, shard_full=True, tp_comm=tp_comm, qkv_slabs=qkv_slabs)
        assert r.fused <= 1 or world == 1     # the level of the row-parallel projections drops to 1 under TP
        for _ in range(steps):
            r.step()
        torch.cuda.synchronize()
        if r.comm is not None:
            r.comm.check_error()
        assert r.graph_error is None, r.graph_error
        ret[(world, rank, tp_comm, graph)] = (r.x.float().cpu().numpy(), r.tokens.cpu().numpy())
        ret[(world, rank)] = ret[(world, rank, tp_comm, graph)]
    finally:
        if world > 1:
            dist.destroy_process_group()


# steps = 1: the softmax sees only the current token, so the only difference between TP=2 and TP=1 is the
# rank-local int8 activation scale of the row-parallel inputs (tight bound).  steps = 3 also reads the sharded
# KV cache; the random-weight model amplifies the quantisation noise through its (near one-hot) softmax, so the
# bound is loose there -- a wrong shard or a missing all-reduce gives cos ~ 0.
@pytest.mark.parametrize("group_size,steps,min_cos,max_rel", [(-1, 1, 0.995, 0.1), (128, 1, 0.995, 0.1),
                                                              (-1, 3, 0.95, 0.35)])
def test_tp2_matches_single_gpu(group_size, steps, min_cos, max_rel):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_run_rank, args=(rk, 2, port, group_size, steps, ret)) for rk in range(2)]
    for p in procs:
        p.start()
    _run_rank(0, 1, 0, group_size, steps, ret)           # single-GPU reference in this process
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, "TP rank failed"
    ref_x, _ = ret[(1, 0)]
    x0, t0 = ret[(2, 0)]
    x1, t1 = ret[(2, 1)]
    assert np.array_equal(x0, x1) and np.array_equal(t0, t1), "ranks diverged after the all-reduce"
    assert np.isfinite(x0).all()
    cos = (ref_x * x0).sum() / (np.linalg.norm(ref_x) * np.linalg.norm(x0))
    rel = np.linalg.norm(ref_x - x0) / np.linalg.norm(ref_x)
    assert cos > min_cos and rel < max_rel, "TP=2 hidden state differs from TP=1: cos %.5f rel %.4f" % (cos, rel)


def _run_row_parallel_rank(rank, world, port, group_size, M, N, K, ret):
    """One rank of a row-parallel projection through the HIP GEMM + tp.all_reduce_ (what o_proj / down_proj do)."""
    import torch.distributed as dist
    from omniserve_amd import tp
    from omniserve_amd.backend import fused_kernels, qgemm_w4a8_per_chn, qgemm_w4a8_per_group
    from oracle import w4a8
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        x = torch.from_numpy(_row_parallel_input(M, K)).to(dev)
        k0, k1 = tp.shard_range(K, rank, world, 128)
        xs = x[:, k0:k1].contiguous()
        q = torch.empty((M, k1 - k0), dtype=torch.int8, device=dev)
        sc = torch.empty((M,), dtype=torch.float16, device=dev)
        sm = torch.empty((M,), dtype=torch.float16, device=dev)
        out = torch.empty((M, N), dtype=torch.float16, device=dev)
        if group_size == -1:
            u, z, s1 = w4a8.synth_per_channel(N, K, 11)
            qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
            fused_kernels.invoke_quant_fuse_sum(q, xs, sm, sc)            # rank-local activation scale and sum
            qgemm_w4a8_per_chn.gemm_forward_cuda(q, tp.shard_qweight_k(torch.from_numpy(qw), rank, world).to(dev),
                                                 torch.from_numpy(s1h).to(dev), sc, torch.from_numpy(szh).to(dev), sm, out)
        else:
            u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=11)
            qw, s1h, s2s_p, s2z_p = w4a8.pack_per_group(u, z, s2, s1)
            fused_kernels.invoke_quant(q, xs, sc)
            qgemm_w4a8_per_group.gemm_forward_cuda(
                q, tp.shard_qweight_k(torch.from_numpy(qw), rank, world).to(dev),
                tp.shard_group_params_k(torch.from_numpy(s2z_p), rank, world).to(dev),
                tp.shard_group_params_k(torch.from_numpy(s2s_p), rank, world).to(dev), torch.from_numpy(s1h).to(dev), sc, out)
        tp.all_reduce_(out)
        torch.cuda.synchronize()
        ret[rank] = out.cpu().numpy()
    finally:
        dist.destroy_process_group()


def _row_parallel_input(M, K):
    return (np.random.default_rng(17).standard_normal((M, K)) * 1.5).astype(np.float16)


@pytest.mark.parametrize("group_size", [-1, 128])
def test_row_parallel_projection_is_bit_exact_vs_sharded_oracle(group_size):
    """o_proj / down_proj under TP=2: every rank quantises its K shard with rank-local scales, multiplies it with its
    tile-view weight shard (HIP GEMM) and the fp16 partial projections are summed by tp.all_reduce_.  The oracle does the
    same per shard (quantiser + GEMM restatements) and adds the two fp16 partials: at world 2 the sum has one rounding, so
    the comparison is BIT-EXACT (not the cosine bound of the end-to-end test above)."""
    import torch.multiprocessing as mp
    from omniserve_amd import tp
    from oracle import elementwise as oe
    from oracle import w4a8
    M, N, K = 5, 256, 1024
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_run_row_parallel_rank, args=(rk, 2, port, group_size, M, N, K, ret)) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, "TP rank failed"
    x = _row_parallel_input(M, K)
    parts = []
    for rk in range(2):
        k0, k1 = tp.shard_range(K, rk, 2, 128)
        a, sa, asum = oe.quant_per_token(x[:, k0:k1], group_size == -1)
        if group_size == -1:
            u, z, s1 = w4a8.synth_per_channel(N, K, 11)
            qw, s1h, szh = w4a8.pack_per_channel(u, z, s1)
            qws = tp.shard_qweight_k(torch.from_numpy(qw), rk, 2).numpy()
            parts.append(w4a8.gemm_per_chn(a, qws, s1h, sa, szh, asum))
        else:
            u, z, s2, s1 = w4a8.synth_per_group(N, K, seed=11)
            qw, s1h, s2s_p, s2z_p = w4a8.pack_per_group(u, z, s2, s1)
            qws = tp.shard_qweight_k(torch.from_numpy(qw), rk, 2).numpy()
            parts.append(w4a8.gemm_per_group(a, qws, tp.shard_group_params_k(torch.from_numpy(s2z_p), rk, 2).numpy(),
                                             tp.shard_group_params_k(torch.from_numpy(s2s_p), rk, 2).numpy(), s1h, sa))
    want = (parts[0].astype(np.float32) + parts[1].astype(np.float32)).astype(np.float16)
    for rk in range(2):
        got = ret[rk]
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), "rank %d differs from the sharded oracle" % rk


def _spawn2(target, args_of_rank):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(rk, 2, port) + tuple(args_of_rank) + (ret,)) for rk in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0, "rank failed (exit code %s)" % p.exitcode
    return ret


def _peer_allreduce_rank(rank, world, port, shapes, algo, ret):
    """PeerComm on its own: a sequence of all-reduces of different sizes, eagerly and from a replayed HIP graph."""
    import torch.distributed as dist
    from omniserve_amd import tp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        comm = tp.PeerComm(rank, world, max(shapes), dev, algo=algo)
        outs = []
        for rep, n in enumerate(shapes + shapes):                        # eager
            g = torch.Generator(device="cpu").manual_seed(1000 * rep + 17 * n + rank)
            comm.slot(n).copy_(torch.randn((n,), generator=g).half())
            out = torch.empty((n,), dtype=torch.float16, device=dev)
            comm.all_reduce(out)
            outs.append(out.cpu().numpy())
        # a captured pair of collectives, replayed: epochs live on the device, slot parity is baked per call
        n = shapes[0]
        src = torch.zeros((n,), dtype=torch.float16, device=dev)
        o1 = torch.empty_like(src); o2 = torch.empty_like(src)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            pass
        with torch.cuda.graph(graph):
            comm.slot(n).copy_(src)
            comm.all_reduce(o1)
            comm.slot(n).copy_(o1)
            comm.all_reduce(o2)
        for it in range(3):
            src.fill_(float(rank + 1 + it))
            graph.replay()
            torch.cuda.synchronize()
            outs.append(o2.cpu().numpy().copy())
        comm.check_error()
        ret[rank] = outs
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


def _peer_stale_rank(rank, world, port, n, iters, algo, ret):
    """Every round: fresh values into the slot the peer read LAST time with ordinary cached loads (a torch reduction over
    the peer's mapped buffer = lines of the old contents planted in this process' caches), then the collective."""
    import torch.distributed as dist
    from omniserve_amd import tp
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        comm = tp.PeerComm(rank, world, n, dev, algo=algo)
        peer = torch.as_tensor(tp.PeerComm._Blob(comm._mapped[0], comm.data_bytes), device=dev).view(torch.int16)   # both slots
        out = torch.empty((n,), dtype=torch.float16, device=dev)
        outs, planted = [], []
        for it in range(iters):
            comm.slot(n).fill_(float(2 * it + rank + 1))
            torch.cuda.synchronize()
            dist.barrier()                                   # both slots of this round are written
            comm.all_reduce(out)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy().copy())
            planted.append(int(peer.long().sum().item()))    # plain loads over the peer's two slots: their lines are cached here now
            dist.barrier()
        comm.check_error()
        ret[rank] = (outs, planted)
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


def _peer_add_norm_rank(rank, world, port, cases, algo, ret):
    """PeerComm.add_rms_norm (the all-reduce folded into add + norm + quant) against all_reduce + the reference sequence, for row
    widths that take one, two and four vectors per thread (tp_add_norm_v2_kernel<512, 1 | 2 | 4>)."""
    import torch.distributed as dist
    from omniserve_amd import tp
    import omniserve_backend.layernorm_ops as ln
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        comm = tp.PeerComm(rank, world, max(t * h for t, h in cases), dev, algo=algo)
        ref = tp.PeerComm(rank, world, max(t * h for t, h in cases), dev, algo="one_shot")      # the reference sequence's all-reduce
        outs = []
        for ci, (tokens, hidden) in enumerate(cases):
            n = tokens * hidden
            g = torch.Generator(device="cpu").manual_seed(77 * ci + rank)
            part = torch.randn((n,), generator=g).half()
            gs = torch.Generator(device="cpu").manual_seed(500 + ci)          # the same on both ranks
            resid = (2.0 * torch.randn((tokens, hidden), generator=gs)).half().to(dev)
            gamma = (1.0 + 0.1 * torch.randn((hidden,), generator=gs)).half().to(dev)
            # reference sequence: all-reduce, residual add, rms_norm_general_fuse_sum
            ref.slot(n).copy_(part)
            red = torch.empty((n,), dtype=torch.float16, device=dev)
            ref.all_reduce(red)
            x1 = resid.clone(); x1.add_(red.view(tokens, hidden))
            q1 = torch.empty((tokens, hidden), dtype=torch.int8, device=dev)
            sc1 = torch.empty((tokens,), dtype=torch.float16, device=dev); sm1 = sc1.clone()
            ln.rms_norm_general_fuse_sum(q1, x1, gamma, sm1, sc1, 1e-5, True)
            # fused: the peers' slots summed inside the norm kernel
            comm.slot(n).copy_(part)
            x2 = resid.clone()
            q2 = torch.empty_like(q1); sc2 = torch.empty_like(sc1); sm2 = torch.empty_like(sc1)
            comm.add_rms_norm(q2, x2, gamma, sm2, sc2, 1e-5)
            torch.cuda.synchronize()
            outs.append((bool(torch.equal(x1.view(torch.int16), x2.view(torch.int16))), bool(torch.equal(q1, q2)),
                         bool(torch.equal(sc1.view(torch.int16), sc2.view(torch.int16))),
                         bool(torch.equal(sm1.view(torch.int16), sm2.view(torch.int16))), x2.cpu().numpy()))
        comm.check_error()
        ref.check_error()
        ret[rank] = outs
        dist.barrier()
        comm.close()
        ref.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("algo", ["one_shot", "two_shot"])
def test_peer_add_rms_norm_row_widths_two_ranks_one_gpu(algo):
    """tp_add_norm_v2_kernel batches the peer loads of all of a thread's vectors (round 4): rows of 4096 / 8192 / 12288
    columns = one / two / four vectors per thread, bit-identical to all-reduce -> add -> rms_norm_general_fuse_sum, same
    residual on both ranks."""
    cases = [(16, 4096), (128, 8192), (5, 12288), (3, 2048), (1, 4096)]      # (two_shot: chunk = ceil(tokens / 2) whole rows;
    ret = _spawn2(_peer_add_norm_rank, (cases, algo))                          #  one token: rank 1 owns nothing)
    for ci, case in enumerate(cases):
        for rk in range(2):
            same_x, same_q, same_sc, same_sm, _ = ret[rk][ci]
            assert same_x and same_q and same_sc and same_sm, (case, rk, same_x, same_q, same_sc, same_sm)
        assert np.array_equal(ret[0][ci][4].view(np.uint16), ret[1][ci][4].view(np.uint16)), case


@pytest.mark.parametrize("algo", ["one_shot", "two_shot"])
def test_peer_allreduce_two_ranks_one_gpu(algo):
    """The library's own all-reduce over hipIpc-mapped peer buffers (tp.PeerComm), two processes on the one test GPU:
    sums are bit-exact (f32 accumulate in rank order, one rounding) and identical on both ranks, eagerly and replayed from a
    captured graph.  (On this rig the 'peer' memory is the same device: the synchronisation protocol, the slot
    alternation and the graph capture are what is under test -- xGMI traffic is not.)"""
    shapes = [4096, 8, 128 * 8192, 16 * 4096]      # (two_shot with 8 elements: rank 1's chunk is empty)
    ret = _spawn2(_peer_allreduce_rank, (shapes, algo))
    for rep, n in enumerate(shapes + shapes):
        parts = []
        for rk in range(2):
            g = torch.Generator(device="cpu").manual_seed(1000 * rep + 17 * n + rk)
            parts.append(torch.randn((n,), generator=g).half().float())
        want = (parts[0] + parts[1]).half().numpy()
        for rk in range(2):
            assert np.array_equal(ret[rk][rep].view(np.uint16), want.view(np.uint16)), (rep, n, rk)
    for it in range(3):     # graph: o1 = (1+it) + (2+it); o2 = 2 * o1
        want = np.full((shapes[0],), 2.0 * (3 + 2 * it), np.float16)
        for rk in range(2):
            assert np.array_equal(ret[rk][len(shapes) * 2 + it], want), (it, rk)


@pytest.mark.parametrize("algo", ["one_shot", "two_shot"])
def test_peer_allreduce_reads_fresh_slots_behind_planted_stale_lines(algo):
    """VERDICT r4 item 3 for the peer collective: between two uses of a slot the OTHER process reads it with ordinary cached
    loads (so any cache level that could keep a non-coherent copy of it holds the old contents), the owner then overwrites
    it and the collective must return the new sum.  24 rounds x 64 K elements, two processes on the one test GPU (different
    processes' kernels land on different XCDs: the L2s are not shared).  The buffers are fine-grained allocations and the
    kernels acquire at system scope behind the flag wait (csrc/tp_comm.h); a plain allocation or a missing acquire fails here.
    two_shot: the same for the gather regions (the planting read covers them: the peer's reduced chunk of the LAST round sits in
    this process' caches when the next round reads the fresh one)."""
    n, iters = 65536, 24
    ret = _spawn2(_peer_stale_rank, (n, iters, algo))
    for rk in range(2):
        outs, planted = ret[rk]
        for it in range(iters):
            want = np.full((n,), float(4 * it + 3), np.float16)
            assert np.array_equal(outs[it], want), "round %d rank %d: %s ... instead of %s" % (it, rk, outs[it][:4], want[:1])
        assert all(p != 0 for p in planted)      # (the planting reads really happened)


@pytest.mark.parametrize("group_size,graph,comm", [(-1, False, "peer"), (128, False, "peer"), (-1, True, "peer"),
                                                   (-1, True, "peer2"), (128, False, "peer2")])
def test_tp2_peer_comm_matches_the_collective_path_bitwise(group_size, graph, comm):
    """DecodeRunner(tp_comm="peer"): the two all-reduces per layer run on PeerComm, folded into the add + norm kernel
    (one launch for all-reduce + residual add + norm + quant).  At world 2 the fp32-accumulated sum has one rounding, as
    the host-staged gloo path of this rig and an fp16 RCCL ring have: hidden states and tokens must be bit-identical to
    the torch.distributed path, on both ranks, eagerly and with the whole step in one HIP graph.  "peer2": the two-shot form
    (each rank reduces its half of the rows into its gather region, then both read the halves) -- same bits."""
    steps = 3
    ref = _spawn2(_run_rank, (group_size, steps))
    got = _spawn2(_run_rank_peer, (group_size, steps, graph, comm))
    for rk in range(2):
        xr, tr = ref[(2, rk)]
        xg, tg = got[(2, rk)]
        assert np.array_equal(xr, xg) and np.array_equal(tr, tg), "rank %d: peer collective differs" % rk
    assert np.array_equal(got[(2, 0)][0], got[(2, 1)][0])


def _run_rank_peer(rank, world, port, group_size, steps, graph, comm, ret):
    _run_rank(rank, world, port, group_size, steps, ret, tp_comm=comm, graph=graph)


def _run_rank_l2_attn(rank, world, port, group_size, steps, graph, ret):
    # (batch 4: qkv_slabs="auto" would leave the slab form of the qkv projection off)
    _run_rank(rank, world, port, group_size, steps, ret, tp_comm="peer", graph=graph, fused=True, qkv_slabs=True)


@pytest.mark.parametrize("group_size,graph", [(-1, False), (128, True)])
def test_tp2_attention_side_fusions_match_level_1_bitwise(group_size, graph):
    """Under TP the level of the row-parallel projections is 1, but the attention-side fusions of level 2 stay (split merge
    inside the quantiser, q / k / v read from the column-parallel qkv projection's slabs): hidden states and tokens
    bit-identical to the plain level-1 TP run, on both ranks."""
    steps = 3
    ref = _spawn2(_run_rank, (group_size, steps))
    got = _spawn2(_run_rank_l2_attn, (group_size, steps, graph))
    for rk in range(2):
        xr, tr = ref[(2, rk)]
        xg, tg = got[(2, rk)]
        assert np.array_equal(xr, xg) and np.array_equal(tr, tg), "rank %d: attention-side fusions differ" % rk


def test_bench_spawns_two_ranks_on_the_one_gpu():
    """`python bench.py --gpus 2` end to end on hardware: it re-executes itself under torch.distributed.run, both ranks build their
    TP = 2 shard of Llama-3-8B on cuda:0 (OMNI_BENCH_ONE_GPU=1: gloo process group, the library's peer collective inside the
    graph-captured step -- RCCL refuses two ranks on one device), and rank 0 prints the line with the rank count of an
    all-reduce of ones.  (The RCCL-in-graph path needs N GPUs: unmeasured on hardware, see DESIGN.md.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMNI_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--model", "llama3-8b",
                        "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 4 and line["value"] > 0 and line["scaling"] == "strong"
    tp = line["tensor_parallel"]
    assert tp["ranks_in_all_reduce"] == 2 and tp["step_collective"] == "peer" and tp["process_group_backend"] == "gloo"
    assert tp["graph_capture_fell_back_to_eager"] is False, tp
    assert line["config"]["all_ranks_on_one_gpu"] is True and "tp2" in line["config"]["parallelism"]
