#!/usr/bin/env python
"""bench.py -- decode tokens/s of the QServe W4A8KV4 hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one decode step (one new token for each of the 16 sequences) of Llama-3-8B
W4A8KV4 per-channel at bs=16, 1024 cached tokens per sequence (BASELINE.json configs[1]), with
synthetic packed weights / KV pages already resident in HBM, all kernels going through the
C ABI (libomniserve_hip.so), the step captured in one HIP graph.

With N > 1 (`--gpus N`; bench.py starts the N ranks itself when no launcher set WORLD_SIZE) the workload is BASELINE.json
configs[4]: ONE Llama-2-70B W4A8KV4 model at bs=128 sharded TP=N (column / row parallel projections, attention by kv head,
one fp16 sum all-reduce over RCCL / xGMI after o_proj and after down_proj INSIDE the step): strong scaling, value = 128 x steps /
max-over-ranks time, `tensor_parallel.all_reduce_us` reported.  Its N = 1 point is `llama2_70b_tp1` of the one-GPU line (or
`bench.py --model llama2-70b`).  `--replicas` keeps the old N > 1 mode: N independent copies of configs[1], no collective.

Rank 0 prints ONE JSON line.  Extra objects:
  roofline      dominant kernel = the gate_up W4A8 GEMV (N=28672, K=4096, M=16) in the form the step runs
                it (fusion level 3: with the silu_and_mul epilogue): algorithmic bytes (packed weights +
                activations + output) / average duration measured with HIP events on the launch stream,
                weights rotated over all 32 layers (1.9 GB) so nothing is cache resident and NO L2
                prefetch runs.  peak = 8 TB/s (MI355X HBM3E).  `gemv_aggregate` = the four projections of
                a layer together, `attention` = the KV4 decode attention of a layer vs 1088*T*B bytes,
                `step` = the whole decode step vs weights + KV + lm_head bytes.
  drop_in       the same decode step through the REFERENCE call sequence only (no fused extension
                entry points, no HIP graph, no prefetch): what an unmodified reference host stack
                would get from the mirror.
  protocol      qserve_benchmark.py protocol run for real: one prefill of 1024 tokens per sequence
                and 511 decode steps (graph replays, context growing 1024 -> 1535), B*512 / wall.
  configs2_g128_bs64   BASELINE.json configs[2]: g128 weights, batch 64 (decode step + protocol).
  cpu_baseline  the oracle restatement of the per-channel GEMMs of one decoder layer at bs=16
                (torch._int_mm on the host cores) extrapolated to a full step -- a reported
                baseline, not a target; `gemm_4096` = BASELINE.json configs[0] on the host (8 threads, 1 thread).
  w4a8_gemm_4096  BASELINE.json configs[0] shape on the GPU: int8 TOPS vs the 5 PFLOP/s dense
                int8 MFMA peak.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
HBM_ACHIEVABLE_GBS = 6300.0  # same guide, "8 TB/s peak (spec); ~6.3 TB/s achievable"
INT8_PEAK_TOPS = 5000.0     # dense int8 MFMA
# Numbers that only a profiler run can produce -- the dominant kernel's HBM traffic per launch (rocprofv3 --pmc, separate
# FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950) and its duration INSIDE
# the captured decode step (rocprofv3 --kernel-trace) -- are read from the JSON the profiling script writes
# (tools/r03_profile.sh -> tools/make_bench_constants.py -> profiles/bench_constants.json), keyed by kernel form and
# shape, together with the file they came from.  Nothing is hard-coded here: a kernel change without a new profile run
# shows up as `null`, not as a stale number.
def _profile_constants():
    path = os.path.join(ROOT, "profiles", "bench_constants.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def event_time_ms(fn, iters, warm=3):
    for _ in range(warm):
        fn(0)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def _time_projection(runner, name):
    """Event-time one projection's GEMV alone (the reference entry point, weights rotated over the layers so that
    nothing is cache resident, no prefetch) -> (ms per launch, algorithmic bytes per launch)."""
    B = runner.B
    nl = len(runner.layers)
    lin0 = runner.layers[0][name]
    N, K = lin0.n, lin0.k
    x = torch.randint(-127, 128, (B, K), dtype=torch.int8, device=runner.device)
    sc = torch.full((B,), 0.01, dtype=torch.float16, device=runner.device)
    sm = torch.zeros((B,), dtype=torch.float16, device=runner.device)
    out = torch.empty((B, N), dtype=torch.float16, device=runner.device)

    def fn(i):
        runner.layers[i % nl][name].forward(x, sc, sm, out)

    # One HIP graph of one launch per layer, replayed: host launch overhead (~10 us per eager call, more than the small
    # projections take) stays out of the figure; HIP events bracket the replays on the launch stream.
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(nl):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nl):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    reps = 6
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / (reps * nl)
    # algorithmic bytes per launch: M*K + N*K/2 + 2*M*N + 4*N + 4*M (+ g128 params)
    return ms, B * K + lin0.weight_bytes() + 2 * B * N + 4 * N + 4 * B


def _graph_time_ms(fn, nl, reps=6):
    """fn(i) launched once per layer in one HIP graph, replayed `reps` times: ms per launch (HIP events on the launch stream)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(nl):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(nl):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (reps * nl)


def _time_gate_up_silu(runner):
    """The gate_up projection in the form fusion level 3 runs it: GEMV + silu_and_mul epilogue + row maxima (one kernel),
    cold weights, no prefetch -> (ms per launch, algorithmic bytes: weights + int8 input + fp16 ACTIVATION out + params)."""
    from omniserve_amd.backend import fused_ext
    B, nl = runner.B, len(runner.layers)
    lin0 = runner.layers[0]["gate_up"]
    N, K = lin0.n, lin0.k
    x = torch.randint(-127, 128, (B, K), dtype=torch.int8, device=runner.device)
    sc = torch.full((B,), 0.01, dtype=torch.float16, device=runner.device)
    sm = torch.zeros((B,), dtype=torch.float16, device=runner.device)
    act = torch.empty((B, N // 2), dtype=torch.float16, device=runner.device)
    amax = fused_ext.new_amax_slots(B, runner.device)

    def fn(i):
        G = runner.layers[i % nl]["gate_up"]
        if G.group == -1:
            fused_ext.gemm_silu_per_chn(x, G.qweight, G.s1_scales, sc, G.s1_szeros, sm, act, amax)
        else:
            fused_ext.gemm_silu_per_group(x, G.qweight, G.s2_zeros, G.s2_scales, G.s1_scales, sc, act, amax)

    return _graph_time_ms(fn, nl), B * K + lin0.weight_bytes() + 2 * B * (N // 2) + 4 * N + 4 * B


def _time_attention(runner):
    """The KV4 decode attention of one layer as the step runs it (split partials + merge), pools rotated over the layers
    (570 MB: cold) -> (ms per layer, algorithmic KV bytes per layer = 1088 * T * B for Llama-3-8B)."""
    from omniserve_amd.backend import fused_ext
    c, B, nl = runner.cfg, runner.B, len(runner.layers)
    hq, hk, d = runner.hl, runner.kl, c.head_dim
    q = runner.qkv_buf[:, : hq * d].view(B, hq, d)
    k = runner.qkv_buf[:, hq * d:(hq + hk) * d].view(B, hk, d)
    v = runner.qkv_buf[:, (hq + hk) * d:].view(B, hk, d)
    T = int(runner.lengths[0].item())
    if runner.fused >= 3:
        def fn(i):
            fused_ext.decode_attention_f16_amax(runner.attn_f16, runner.amax[i % nl, 0], q, k, v, runner.block_tables[i % nl],
                                                runner.lengths, runner.tpb, runner.max_context, c.rope_theta)
    else:
        def fn(i):
            fused_ext.decode_attention_quant_fuse_sum(runner._q_attn, q, k, v, runner.block_tables[i % nl], runner.lengths,
                                                      runner.tpb, runner.max_context, c.rope_theta, runner.act_sum2,
                                                      runner.act_scale2)
    return _graph_time_ms(fn, nl), runner.kv_bytes_per_step(T) // nl, T


def roofline_gate_up(runner, ms_per_step=None):
    """Event-time the gate_up GEMV alone, rotating over the layers' weights; plus the four projections together, the
    attention kernel pair, and the whole step against its algorithmic bytes."""
    B = runner.B
    lin0 = runner.layers[0]["gate_up"]
    N, K = lin0.n, lin0.k
    ms_plain, alg_plain = _time_projection(runner, "gate_up")
    silu = runner.fused >= 3
    ms, alg = _time_gate_up_silu(runner) if silu else (ms_plain, alg_plain)
    achieved = alg / (ms * 1e-3) / 1e9
    parts = {"gate_up": (ms_plain, alg_plain)}
    for name in ("qkv", "o", "down"):
        parts[name] = _time_projection(runner, name)
    tot_ms = sum(v[0] for v in parts.values())
    tot_b = sum(v[1] for v in parts.values())
    aggregate = {"us_per_layer": round(tot_ms * 1e3, 2), "bytes_per_layer": tot_b,
                 "achieved": round(tot_b / (tot_ms * 1e-3) / 1e9, 1), "frac": round(tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 "us": {k: round(v[0] * 1e3, 2) for k, v in parts.items()},
                 "note": "each projection's GEMV through the reference entry point (+ its split-K epilogue kernel where the "
                         "plan splits) timed alone in a HIP graph of 32 launches over the layers' (cold) weights; every launch "
                         "pays its ~1.7 us kernel boundary; inside the decode step the row kernels prefetch the head of each "
                         "weight stream into L2 (profiles/ has the in-step durations)"}
    consts = _profile_constants()
    form = "silu" if silu else "plain"
    key = "gate_up_%s M=%d N=%d K=%d g=%d" % (form, B, N, K, lin0.group)
    prof = consts.get(key, {})
    out = {"bound": "hbm",
           "kernel": ("w4a8_gemv_kernel<1,CHN,false,4,*,1,EPI=1> (gate_up GEMV + silu_and_mul epilogue, M=%d N=%d K=%d, one kernel)"
                      if silu else "w4a8_gemv_kernel<1,CHN,false,4> (gate_up GEMV M=%d N=%d K=%d, one kernel)") % (B, N, K),
           "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": prof.get("pmc_traffic_bytes"),
           "traffic_source": prof.get("pmc_source"),
           "bytes_per_launch": alg, "us_per_launch": round(ms * 1e3, 2),
           "how_timed": "isolated: HIP graph of one launch per layer (cold weights, no L2 prefetch), HIP events on the launch "
                        "stream, duration = elapsed / launches (each launch includes its ~1.7 us dependent-kernel boundary); "
                        "rocprofv3 of the whole command mixes these launches with the shorter in-step ones",
           "in_step_us_per_launch": prof.get("in_step_us"), "in_step_source": prof.get("in_step_source"),
           "measured_copy_rate_GBps": HBM_ACHIEVABLE_GBS,      # context only (MI355X_MICROARCH.md: float4 copy 6.29 TB/s); `frac` is against the 8 TB/s peak
           "gemv_aggregate": aggregate}
    sc = consts.get("stream_ceiling M=%d N=%d K=%d" % (B, N, K))
    if sc:      # what an isolated launch that only LOADS this kernel's weight bytes costs on this chip (a measured ceiling, not a peak)
        out["stream_ceiling"] = {"us_per_launch": sc.get("us_per_launch"), "frac": sc.get("frac_of_hbm_peak"),
                                 "achieved_vs_ceiling": round(achieved / sc["GBps"], 4) if sc.get("GBps") else None,
                                 "source": sc.get("source")}
    if silu:
        out["gate_up_plain_form"] = {"us_per_launch": round(ms_plain * 1e3, 2), "bytes_per_launch": alg_plain,
                                     "frac": round(alg_plain / (ms_plain * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    try:
        a_ms, a_bytes, T = _time_attention(runner)
        single = runner.fused >= 3 and getattr(runner, "attn_single", False)
        out["attention"] = {"kernel": ("kv4_decode_flash_kernel<4, ..., LASTM> (one launch: the last-arriving split workgroup merges; "
                                       "one layer, B=%d, T=%d)" if single else
                                       "kv4_decode_flash_kernel<4> + merge (one layer, B=%d, T=%d)") % (B, T),
                            "us_per_layer": round(a_ms * 1e3, 2), "bytes_per_layer": a_bytes,
                            "achieved": round(a_bytes / (a_ms * 1e-3) / 1e9, 1),
                            "frac": round(a_bytes / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "algorithmic KV bytes 1088*T*B per layer; isolated launches over the layers' (cold) pools, as the "
                                    "step issues them (%s); at T=1024 the kernel is a latency chain (page table -> K/V -> softmax "
                                    "-> combine), not a stream" % ("one launch" if single else "split partials, then the merge launch")}
    except Exception as exc:   # noqa: BLE001
        out["attention"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if ms_per_step:
        c = runner.cfg
        lm = 2 * c.vocab * c.hidden
        sb = runner.gemm_weight_bytes_per_step() + runner.kv_bytes_per_step(int(runner.lengths[0].item())) + lm
        out["step"] = {"bytes": sb, "achieved": round(sb / (ms_per_step * 1e-3) / 1e9, 1),
                       "frac": round(sb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "whole decode step: W4A8 weights + KV4 pages + fp16 lm_head (%d B) over the timed ms_per_step" % lm}
    return out


def _step_roofline(cfg, weight_bytes, kv_bytes, ms_per_step, tp=1):
    """roofline.step of a secondary leg: W4A8 weights + KV4 pages + fp16 lm_head of ONE decode step over its measured time,
    as a fraction of the HBM peak (every leg prints its own step fraction)."""
    sb = int(weight_bytes) + int(kv_bytes) + 2 * cfg.vocab * cfg.hidden
    return {"bytes": sb, "achieved_GBps": round(sb / (ms_per_step * 1e-3) / 1e9, 1),
            "step_frac": round(sb / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "peak_GBps": HBM_PEAK_GBS}


def mid_m_leg(device):
    """The decode GEMM between the GEMV and the prefill regime: Llama-3-8B gate_up [28672, 4096] per-channel at
    M = 64 / 128 (w4a8_midm_kernel, csrc/qgemm_midm.h) and 256 (the 128 x 256 prefill tile), weights rotated over 6 copies
    (352 MB > the 256 MB memory-side cache), HIP events on the launch stream; fractions of BOTH ceilings."""
    from omniserve_amd.backend import qgemm_w4a8_per_chn
    N, K = 28672, 4096
    g = torch.Generator(device=device); g.manual_seed(0)
    ws = [torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=device, generator=g).view(torch.int8) for _ in range(6)]
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=device)
    sz = torch.full((N,), 0.05, dtype=torch.float16, device=device)
    out = {}
    for M in (64, 128, 256):
        a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=device, generator=g)
        sa = torch.full((M,), 0.01, dtype=torch.float16, device=device)
        asum = torch.zeros((M,), dtype=torch.float16, device=device)
        o = torch.empty((M, N), dtype=torch.float16, device=device)
        ms = event_time_ms(lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, ws[i % 6], sw, sa, sz, asum, o), iters=36, warm=6)
        nbytes = M * K + N * K // 2 + 2 * M * N + 4 * N + 4 * M
        ops = 2.0 * M * N * K
        out["gate_up_M%d" % M] = {"us": round(ms * 1e3, 2), "bytes": nbytes,
                                  "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "int8_tops": round(ops / (ms * 1e-3) / 1e12, 1),
                                  "frac_of_int8_mfma_peak": round(ops / (ms * 1e-3) / 1e12 / INT8_PEAK_TOPS, 4)}
    return out


def gemm_4096(device):
    from omniserve_amd.backend import qgemm_w4a8_per_chn
    M = N = K = 4096
    g = torch.Generator(device=device); g.manual_seed(0)
    a = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=device, generator=g)
    w = torch.randint(0, 256, (N, K // 2), dtype=torch.uint8, device=device, generator=g).view(torch.int8)
    sw = torch.full((N,), 0.01, dtype=torch.float16, device=device)
    sz = torch.full((N,), 0.05, dtype=torch.float16, device=device)
    sa = torch.full((M,), 0.01, dtype=torch.float16, device=device)
    asum = torch.zeros((M,), dtype=torch.float16, device=device)
    out = torch.empty((M, N), dtype=torch.float16, device=device)
    ms = event_time_ms(lambda i: qgemm_w4a8_per_chn.gemm_forward_cuda(a, w, sw, sa, sz, asum, out), iters=20)
    tops = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    return {"M": M, "N": N, "K": K, "ms": round(ms, 4), "int8_tops": round(tops, 1),
            "frac_of_int8_mfma_peak": round(tops / INT8_PEAK_TOPS, 4), "peak_tops": INT8_PEAK_TOPS}


def protocol_leg(cfg, args, device, batch=None, fused=None):
    """qserve_benchmark.py protocol run for real: one prefill of `context` tokens per sequence (eager
    launches, KV4 pages written by the prefill writer), then 511 decode steps as HIP-graph replays with the context
    growing 1024 -> 1535 (the captured step increments `lengths` itself); throughput = B * 512 / wall clock.
    One untimed round first (the reference reports the last of three rounds)."""
    from omniserve_amd.runtime import DecodeRunner
    B = batch or args.batch
    gen = 512
    r = DecodeRunner(cfg, B, args.context, gen + 8, device, seed=4321,
                     fused=(0 if args.no_fused else args.fused_level) if fused is None else fused)
    out = {}
    for rnd in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r.prefill(args.context)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(gen - 1):
            r.step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out = {"prompt_len": args.context, "gen_len": gen, "prefill_ms": round((t1 - t0) * 1e3, 2),
               "prefill_tokens_per_s": round(B * args.context / (t1 - t0), 1),
               "decode_ms_per_step_mean": round((t2 - t1) / (gen - 1) * 1e3, 4),
               "tokens_per_s": round(B * gen / (t2 - t0), 1),
               "note": "B*512 / wall clock of (1 prefill + 511 decode graph replays, context 1024 -> 1535), second of "
                       "two rounds; the reference's published A100 figure (3005 tok/s, batch 256) follows the same protocol"}
    # decode steps of the protocol: mean context (prompt + gen / 2) for the KV bytes
    out["roofline_decode_step"] = _step_roofline(cfg, r.gemm_weight_bytes_per_step(), r.kv_bytes_per_step(args.context + gen // 2),
                                                 out["decode_ms_per_step_mean"])
    if int(r.lengths[0]) != args.context + gen - 1:
        raise RuntimeError("protocol leg: unexpected final length %d" % int(r.lengths[0]))
    if not torch.isfinite(r.x.float()).all():
        raise RuntimeError("non-finite activations in the protocol leg")
    del r
    torch.cuda.empty_cache()
    return out


def drop_in_leg(cfg, args, device, steps=24, warmup=4):
    """The decode step through the reference call sequence ONLY (llama_w4a8_unpad.py:406-438: 11 mirror calls per layer
    plus the torch residual adds), launched eagerly: no fused extension entry points, no HIP graph, no L2 prefetch,
    torch.argmax -- what the unmodified reference host stack would get from the drop-in mirror (minus its own Python)."""
    from omniserve_amd import _lib
    from omniserve_amd.runtime import DecodeRunner
    r = DecodeRunner(cfg, args.batch, args.context, steps + warmup + 4, device, seed=99, use_graph=False, fused=0)

    def run(use_ext):
        keep, _lib.USE_EXT = _lib.USE_EXT, use_ext
        try:
            for _ in range(warmup):
                r.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                r.step()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps
        finally:
            _lib.USE_EXT = keep
    have_ext = _lib.fast() is not None
    dt_ctypes = run(False)
    dt = run(True) if have_ext else dt_ctypes
    del r
    torch.cuda.empty_cache()
    return {"ms_per_step": round(dt * 1e3, 4), "tokens_per_s": round(args.batch / dt, 1), "hip_graph": False,
            "fused_ext_level": 0, "prefetch": False,
            "binding": "pybind11 (omniserve_amd/csrc_ext/omni_ext.cpp) for the GEMM / norm / quant / SiLU calls, ctypes for the attention"
                       if have_ext else "ctypes",
            "ms_per_step_ctypes_mirror": round(dt_ctypes * 1e3, 4),
            "note": "eager launches of the reference's own call sequence (~450 launches per step incl. torch's adds / arg-max): bound by "
                    "per-kernel dispatch without a graph, not by the binding -- the pybind11 path halves the host time per call "
                    "(cProfile: 7 -> 3.5 us) and the step does not move (profiles/r05_d_drop_in.md)"}


def configs2_leg(args, device, steps=32, warmup=6):
    """BASELINE.json configs[2]: Llama-3-8B W4A8KV4 g128, batch 64 (benchmark_a100.sh protocol), decode step + protocol."""
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    cfg = LlamaConfig.llama3_8b(128)
    r = DecodeRunner(cfg, 64, args.context, steps + warmup + 4, device, seed=77)
    for _ in range(warmup):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if not torch.isfinite(r.x.float()).all():
        raise RuntimeError("non-finite activations in the g128 decode step")
    out = {"config": "Llama-3-8B W4A8KV4 g128, bs=64, context=%d" % args.context, "ms_per_step": round(dt * 1e3, 4),
           "decode_tokens_per_s": round(64 / dt, 1), "fused_ext_level": r.fused,
           "gemm_weight_bytes_per_step": r.gemm_weight_bytes_per_step(),
           "kv_bytes_per_step": r.kv_bytes_per_step(args.context)}
    out["roofline_step"] = _step_roofline(cfg, out["gemm_weight_bytes_per_step"], out["kv_bytes_per_step"], dt * 1e3)
    del r
    torch.cuda.empty_cache()
    out["protocol"] = protocol_leg(cfg, args, device, batch=64, fused=1)
    return out


def lserve_leg(device, context=256000, steps=32, warmup=8):
    """BASELINE.json configs[3] (parity/measurement case, not the headline): Llama-3-8B W8A8, LServe sparse decode at a
    256K-token context, batch 1 -- 4 retrieval + 4 streaming kv heads, 4096-token page budget, page selector every 4th
    step, sub-chunk 16 (scripts/lserve_benchmark/launch.sh) -- with KV4 fine_grained pages (the config as written) and
    with the per-tensor KV8 pages upstream's published numbers use; plus one layer of the block-sparse prefill attention
    at the same length (16 dense + 16 streaming q heads, sink 128 / local 8192)."""
    from omniserve_amd.lserve_runtime import LServeDecodeRunner
    from omniserve_amd.runtime import LlamaConfig
    from block_sparse_attn import token_streaming_attn_func
    out = {"config": "Llama-3-8B W8A8 bs=1 ctx=%d, 4+4 kv heads (retrieval+streaming), budget 4096, interval 4" % context}
    cfg = LlamaConfig.llama3_8b(-1)
    for fmt in ("kv4", "kv8"):
        r = LServeDecodeRunner(cfg, 1, context, steps + warmup + 4, device, seed=7, kv_format=fmt)
        for _ in range(warmup):
            r.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if not torch.isfinite(r.x.float()).all():
            raise RuntimeError("non-finite activations in the LServe decode step")
        out[fmt] = {"decode_tokens_per_s": round(steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
                    "gemm_weight_bytes_per_step": r.weight_bytes_per_step()}
        if fmt == "kv8":
            # the context stage of the same configuration (time to first token): all 32 layers over one `context`-token
            # prompt -- W8A8 GEMMs, cache write, statistics pooling, dense + Lambda-masked attention -- timed once after a
            # short warm-up prompt that sizes scratch
            r.prefill(seq_len=2048)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r.prefill(seq_len=context)
            torch.cuda.synchronize()
            out[fmt]["context_stage_s"] = round(time.perf_counter() - t0, 2)
            out[fmt]["context_stage_tokens_per_s"] = round(context / (time.perf_counter() - t0), 0)
        del r
        torch.cuda.empty_cache()
    L, Hq, Hk, D = context, cfg.heads, cfg.kv_heads, cfg.head_dim
    q = torch.randn((L, Hq, D), dtype=torch.float16, device=device)
    k = torch.randn((L, Hk, D), dtype=torch.float16, device=device)
    v = torch.randn_like(k)
    cu = torch.tensor([0, L], dtype=torch.int32, device=device)
    g = Hq // Hk      # head classes are per KV head (ctx_attn_init.py:28-50): kv heads alternate dense / streaming
    hm = torch.tensor(sum(([0] * g if kvh % 2 == 0 else [-1] * g for kvh in range(Hk)), []), dtype=torch.int32, device=device)
    si = torch.tensor([128, 8192] * Hq, dtype=torch.int32, device=device)
    ms = event_time_ms(lambda i: token_streaming_attn_func(q, k, v, cu, cu, hm, si, L, L), iters=2, warm=1)
    win = 128 + 8192
    flops = 4.0 * D * (Hq // 2) * (L * L / 2 + L * win - win * win / 2)
    out["prefill_attention_one_layer"] = {"ms": round(ms, 1), "tflops": round(flops / ms * 1e-9, 1),
                                          "frac_of_fp16_mfma_peak": round(flops / ms * 1e-9 / 2500.0, 3)}
    return out


def tp_rank_leg(device, tp=8, batch=128, context=1024, steps=16, warmup=4):
    """BASELINE.json configs[4] seen from ONE rank: the Llama-2-70B W4A8KV4 decode step of a TP=8 shard (column / row
    parallel projections, one kv head, M = 128) on this GPU with the two all-reduces per layer skipped (no process group):
    the compute the RCCL collectives would be overlapped with / added to.  The driver's multi-GPU run uses replicas of
    configs[1]; `bench.py --gpus 8 --tp` runs the real thing."""
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    cfg = LlamaConfig.llama2_70b(-1)
    def run(tp_comm):
        r = DecodeRunner(cfg, batch, context, steps + warmup + 4, device, seed=3, tp_rank=0, tp_size=tp, tp_comm=tp_comm)   # (level drops to 1 under TP; the attention-side fusions stay)
        for _ in range(warmup):
            r.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r.step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        if not torch.isfinite(r.x.float()).all():
            raise RuntimeError("non-finite activations in the TP-shard decode step")
        if r.comm is not None:
            r.comm.check_error()
        wb = (r.gemm_weight_bytes_per_step(), r.kv_bytes_per_step(context))
        del r
        torch.cuda.empty_cache()
        return dt, wb

    dt, (wbytes, kvbytes) = run(None)
    # the same step with the library's own collective kernels in place (all-reduce folded into add + norm + quant), every
    # "peer" slot aliased to this rank's own buffer: what the collectives cost on the compute side, with no fabric traffic
    dt_loop, _ = run("loopback")
    dt_loop2, _ = run("loopback2")      # ... and the two-shot form's kernels (reduce-scatter + all-gather, csrc/tp_comm.h)

    class _R:      # (keeps the return expression below unchanged)
        @staticmethod
        def gemm_weight_bytes_per_step():
            return wbytes
    r = _R
    ar_bytes = 2 * cfg.layers * batch * cfg.hidden * 2
    return {"config": "Llama-2-70B W4A8KV4 per-channel, TP=%d shard (rank 0), bs=%d, context=%d; collectives skipped" % (
                tp, batch, context),
            "ms_per_step_compute_only": round(dt * 1e3, 3), "tokens_per_s_if_collectives_were_free": round(batch / dt, 1),
            "ms_per_step_with_peer_collective_kernels_loopback": round(dt_loop * 1e3, 3),
            "ms_per_step_with_two_shot_collective_kernels_loopback": round(dt_loop2 * 1e3, 3),
            "gemm_weight_bytes_per_step": r.gemm_weight_bytes_per_step(), "kv_bytes_per_step": kvbytes,
            "roofline_step_compute_only": _step_roofline(cfg, wbytes, kvbytes, dt * 1e3),
            "all_reduce_calls_per_step": 2 * cfg.layers, "all_reduce_payload_bytes_per_step": ar_bytes}


def tp1_leg(device, batch=128, context=1024, steps=8, warmup=3):
    """BASELINE.json configs[4]'s model UNSHARDED on one GPU (Llama-2-70B W4A8KV4, bs = 128: 35 GB of packed weights + 12 GB of
    KV4 pages): the N = 1 point the TP = 2 / 4 / 8 lines of `bench.py --gpus N` (strong scaling) are to be read against."""
    from omniserve_amd.runtime import DecodeRunner, LlamaConfig
    cfg = LlamaConfig.llama2_70b(-1)
    r = DecodeRunner(cfg, batch, context, steps + warmup + 4, device, seed=5)
    for _ in range(warmup):
        r.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    if not torch.isfinite(r.x.float()).all():
        raise RuntimeError("non-finite activations in the 70B decode step")
    out = {"config": "Llama-2-70B W4A8KV4 per-channel, TP=1, bs=%d, context=%d" % (batch, context),
           "ms_per_step": round(dt * 1e3, 3), "tokens_per_s": round(batch / dt, 1), "fused_ext_level": r.fused,
           "gemm_weight_bytes_per_step": r.gemm_weight_bytes_per_step(), "kv_bytes_per_step": r.kv_bytes_per_step(context)}
    out["roofline_step"] = _step_roofline(cfg, out["gemm_weight_bytes_per_step"], out["kv_bytes_per_step"], dt * 1e3)
    del r
    torch.cuda.empty_cache()
    return out


def cpu_gemm_4096():
    """BASELINE.json configs[0] / SURVEY.md 8(d): the oracle's restatement of ONE W4A8 per-channel GEMM at M = N = K = 4096
    on the host -- codes unpacked once (untimed), timed: torch._int_mm(A int8 [M,K], U^T int8 [K,N]) -> int32 plus the
    reference's fp32 epilogue -> fp16 -- on 8 threads and on 1 thread, median of 5 after 2 warm-ups."""
    import numpy as np
    M = N = K = 4096
    rng = np.random.default_rng(0)
    a = torch.from_numpy(rng.integers(-127, 128, size=(M, K), dtype=np.int8))
    ut = torch.from_numpy(rng.integers(0, 16, size=(N, K), dtype=np.int8)).t().contiguous()
    sw = torch.full((N,), 0.01); sz = torch.full((N,), 0.05)
    sa = torch.full((M, 1), 0.01); asum = torch.zeros((M, 1))
    out = {"M": M, "N": N, "K": K, "gop": round(2.0 * M * N * K / 1e9, 1)}
    keep = torch.get_num_threads()
    for th in (8, 1):
        torch.set_num_threads(th)
        ts = []
        for i in range(7):
            t0 = time.perf_counter()
            acc = torch._int_mm(a, ut)
            _ = ((acc.float() * sw) * sa - sz * asum).half()
            ts.append(time.perf_counter() - t0)
            if sum(ts) > 25.0 and i >= 2:
                break
        ts = sorted(ts[2:] or ts)
        med = ts[len(ts) // 2]
        out["threads_%d" % th] = {"ms": round(med * 1e3, 1), "gops": round(2.0 * M * N * K / med / 1e9, 1)}
    torch.set_num_threads(keep)
    return out


def cpu_baseline(cfg, batch):
    """Oracle port of one decoder layer's four per-channel W4A8 GEMMs at M=batch on the host cores
    (unpack once, untimed; timed: torch._int_mm + the fp32 epilogue), extrapolated to a step."""
    import numpy as np
    from oracle import w4a8
    cores = min(os.cpu_count() or 1, 16)   # oneDNN int8 at M=32 does not scale past a few cores; the threads actually used
    torch.set_num_threads(cores)
    shapes = [((cfg.heads + 2 * cfg.kv_heads) * cfg.head_dim, cfg.hidden), (cfg.hidden, cfg.hidden),
              (2 * cfg.inter, cfg.hidden), (cfg.hidden, cfg.inter)]
    rng = np.random.default_rng(0)
    M = max(batch, 32)  # torch._int_mm on CPU wants M > 16; rows beyond `batch` are padding
    mats = []
    for (N, K) in shapes:
        u = torch.from_numpy(rng.integers(0, 16, size=(N, K), dtype=np.int8))
        a = torch.from_numpy(rng.integers(-127, 128, size=(M, K), dtype=np.int8))
        sw = torch.full((N,), 0.01); sz = torch.full((N,), 0.05)
        sa = torch.full((M, 1), 0.01); asum = torch.zeros((M, 1))
        mats.append((a, u.t().contiguous(), sw, sz, sa, asum))

    def layer():
        for a, ut, sw, sz, sa, asum in mats:
            acc = torch._int_mm(a, ut)
            _ = ((acc.float() * sw) * sa - sz * asum).half()

    layer()
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 8.0:
        layer()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    step_s = dt * cfg.layers
    return {"value": round(batch / step_s, 2), "unit": "tokens/s", "cores": cores, "kind": "port (GEMMs only)",
            "gemm_4096": cpu_gemm_4096(),
            "sample": "oracle port (torch._int_mm int8 + fp32 epilogue) of the 4 per-channel W4A8 GEMMs of one "
                      "Llama-3-8B decoder layer at M=%d (rows padded to %d) on %d host threads, %d reps in %.1f s, "
                      "x%d layers; attention, norms, quantisers and lm_head are NOT in the sample (GEMMs only)" % (
                          batch, M, cores, reps, dt * reps, cfg.layers)}


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _shard_shapes(cfg, world):
    """Per-rank projection shapes [N, K] of the Megatron split (omniserve_amd/tp.py; pure arithmetic, no device)."""
    d = cfg.head_dim
    hl, kl, il = cfg.heads // world, cfg.kv_heads // world, cfg.inter // world
    return {"qkv": [(hl + 2 * kl) * d, cfg.hidden], "o": [cfg.hidden, hl * d], "gate_up": [2 * il, cfg.hidden],
            "down": [cfg.hidden, il]}


def dry_run(args, cfg, world, rank):
    """No GPU: the N-rank launch path on the gloo backend (tests/test_bench_spawn_cpu.py) -- rendezvous, the configs[4]
    partitioning, and one sum all-reduce of a [B, hidden] projection over all ranks -- then the JSON line with n_gpus = N."""
    import torch.distributed as dist
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        buf = torch.full((args.batch, cfg.hidden), float(rank + 1), dtype=torch.float32)
        dist.all_reduce(buf)
        seen = int(round(float(buf[0, 0]) * 2 / (world + 1)))     # sum of 1..N = N (N + 1) / 2
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": _metric_name(args, world), "value": None, "unit": "tokens/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                          "scaling": "strong" if args.tp_mode else "weak", "vs_baseline": None, "dry_run": True,
                          "config": {"workload": _workload_name(args, cfg, world), "batch": args.batch,
                                     "parallelism": _parallelism(args, world),
                                     "rank_shard_shapes_N_K": _shard_shapes(cfg, world if args.tp_mode else 1)},
                          "tensor_parallel": {"ranks_in_all_reduce": seen, "backend": "gloo (dry run)"}}), flush=True)


def _metric_name(args, world):
    name = "Llama-2-70B" if args.model == "llama2-70b" else "Llama-3-8B"
    if args.tp_mode:
        return "decode tokens/sec (all %d GPUs, one model, TP=%d) %s W4A8KV4 bs=%d" % (world, world, name, args.batch)
    if world == 1:
        return "decode tokens/sec/GPU %s W4A8KV4 bs=%d" % (name, args.batch)
    return "decode tokens/sec (all GPUs) %s W4A8KV4 bs=%d per GPU" % (name, args.batch)


def _workload_name(args, cfg, world):
    name = "Llama-2-70B" if args.model == "llama2-70b" else "Llama-3-8B"
    gs = "per-channel" if args.group_size == -1 else "g%d" % args.group_size
    if args.tp_mode:
        return "%s W4A8KV4 %s decode, bs=%d, context=%d, TP=%d (BASELINE.json configs[4])" % (name, gs, args.batch, args.context, world)
    tag = ("BASELINE.json configs[1]" if (args.model == "llama3-8b" and args.group_size == -1 and args.batch == 16) else
           "configs[4] model on one GPU" if args.model == "llama2-70b" else "configs[2]-like")
    return "%s W4A8KV4 %s decode, bs=%d, context=%d, TP=1 (%s)" % (name, gs, args.batch, args.context, tag)


def _parallelism(args, world):
    if world == 1:
        return "single GPU"
    return ("tp%d (fp16 sum all-reduce x2 per layer, %s)" % (world, "RCCL" if args.tp_comm == "rccl" else "peer-mapped")
            if args.tp_mode else "replicas x%d (no collective)" % world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without a launcher (WORLD_SIZE unset): bench.py starts the N ranks "
                    "itself (python -m torch.distributed.run, rendezvous on 127.0.0.1)")
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=None, help="default: 16 (configs[1]) on one GPU, 128 (configs[4]) with N > 1")
    ap.add_argument("--model", choices=["llama3-8b", "llama2-70b"], default=None,
                    help="default: llama3-8b on one GPU (configs[1]); llama2-70b sharded TP=N with N > 1 (configs[4])")
    ap.add_argument("--replicas", action="store_true", help="with N > 1: N independent replicas of the one-GPU workload (weak "
                    "scaling, no collective) instead of the default tensor-parallel configs[4]")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise the N-rank launch path on gloo and print the line")
    ap.add_argument("--context", type=int, default=1024)
    ap.add_argument("--group-size", type=int, default=-1)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-fused", action="store_true", help="reference call sequence (no fused extension kernels)")
    ap.add_argument("--fused-level", type=int, default=3, help="0 reference sequence, 1 fused add+norm / silu+quant, 2 + deferred split-K epilogue, 3 + no quantiser row kernels (SiLU in the gate_up epilogue, o / down quantise on the fly))")
    ap.add_argument("--no-lserve", action="store_true", help="skip the configs[3] (LServe, 256K context) leg")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / cpu_baseline / gemm_4096 legs")
    ap.add_argument("--tp-comm", choices=["rccl", "peer"], default="rccl",
                    help="with --tp: the two all-reduces per layer on torch.distributed / RCCL (default) or on the library's own "
                         "peer-mapped collective folded into the add + norm kernel (omniserve_amd/tp.py: PeerComm; validated with "
                         "two ranks on one GPU only -- no multi-GPU box was available to its author)")
    ap.add_argument("--tp", action="store_true", help="(default with N > 1; kept for older command lines) shard ONE model over "
                    "the N GPUs: Megatron TP, fp16 sum all-reduce after o_proj / down_proj inside the step; strong scaling")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one.  One process per GPU under torch.distributed.run, same arguments.
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.tp_mode = world > 1 and not args.replicas
    if args.model is None:
        args.model = "llama2-70b" if args.tp_mode else "llama3-8b"
    if args.batch is None:
        args.batch = 128 if args.model == "llama2-70b" else 16
    from omniserve_amd.runtime import LlamaConfig
    cfg = LlamaConfig.llama2_70b(args.group_size) if args.model == "llama2-70b" else LlamaConfig.llama3_8b(args.group_size)
    if args.dry_run:
        return dry_run(args, cfg, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback: the HIP path is the product)")
    # OMNI_BENCH_ONE_GPU=1 (tests/test_tp_gpu.py): every rank on cuda:0 -- the N-rank launch path on the one GPU a test box has.
    # RCCL refuses two ranks on one device, so the process group is gloo there and the step's collective the library's own
    # peer-mapped one; the RCCL-in-graph path itself needs N GPUs and is unmeasured on hardware.
    one_gpu = world > 1 and os.environ.get("OMNI_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
        if args.tp_mode:
            args.tp_comm = "peer"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    red_dev = torch.device("cpu") if one_gpu else device       # where the line's own scalar reductions live
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from omniserve_amd.runtime import DecodeRunner
    tp = args.tp_mode
    runner = DecodeRunner(cfg, args.batch, args.context, args.steps + args.warmup + 4, device,
                          seed=1234 + (0 if tp else rank), use_graph=not args.no_graph,
                          fused=0 if args.no_fused else args.fused_level,
                          tp_rank=rank if tp else 0, tp_size=world if tp else 1,
                          tp_comm=args.tp_comm if tp else None)
    for _ in range(args.warmup):
        runner.step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        elapsed = float(t.item())
    torch.cuda.synchronize()
    if not torch.isfinite(runner.x.float()).all():
        raise SystemExit("non-finite activations in the decode step")

    ar = None
    if dist is not None:
        # did every rank take part?  A sum all-reduce of ones on the job's process group (RCCL unless OMNI_BENCH_ONE_GPU)
        ones = torch.ones((1,), dtype=torch.float32, device=red_dev)
        dist.all_reduce(ones)
        ar = {"ranks_in_all_reduce": int(ones.item()), "process_group_backend": dist.get_backend(),
              "graph_capture_fell_back_to_eager": runner.graph_error if runner.graph_error else False}
    if tp:
        ar.update({"payload_bytes": args.batch * cfg.hidden * 2, "calls_per_step": 2 * cfg.layers, "step_collective": args.tp_comm})
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        buf = torch.zeros((args.batch, cfg.hidden), dtype=torch.float16, device=device)
        nbytes = buf.numel() * 2
        if not one_gpu:
            # the collective of the TP path on its own: in-place fp16 sum all-reduce of one [B, hidden] projection
            # (2 MiB at bs = 128 x 8192), event-timed on the launch stream, max over ranks
            for _ in range(5):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / 50 * 1e3], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            us = float(t.item())
            ar.update({"all_reduce_us": round(us, 2), "bus_GBps": round(2.0 * (world - 1) / world * nbytes / us / 1e3, 1)})
        if runner.comm is not None:
            # the library's own collective on the same payload (an even number of calls keeps the slot parity of the step)
            for _ in range(6):
                runner.comm.all_reduce(buf)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                runner.comm.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            t = torch.tensor([e0.elapsed_time(e1) / 50 * 1e3], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ar["peer_all_reduce_us"] = round(float(t.item()), 2)
            runner.comm.check_error()
    total_tokens = args.batch * args.steps * (1 if tp else world)
    result = {
        "metric": _metric_name(args, world),
        "value": round(total_tokens / elapsed, 1),
        "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if tp else "weak", "vs_baseline": None,
        "dtype": "int8 (W4A8 int32-accumulate GEMM) + fp16 (KV4 attention, fp32 softmax)",
        "data": "synthetic (random packed int4 weights, random KV4 pages, random tokens)",
        "config": {"workload": _workload_name(args, cfg, world),
                   "batch_per_gpu": args.batch if not tp else None, "global_batch": args.batch * (1 if tp else world),
                   "context": args.context, "layers": cfg.layers,
                   "parallelism": _parallelism(args, world),
                   "hip_graph": bool(runner.use_graph), "fused_ext_level": runner.fused,
                   "gemm_weight_bytes_per_step": runner.gemm_weight_bytes_per_step(),
                   "kv_bytes_per_step": runner.kv_bytes_per_step(args.context)},
    }
    if ar is not None:
        result["tensor_parallel" if tp else "distributed"] = ar
        if one_gpu:
            result["config"]["all_ranks_on_one_gpu"] = True      # launch-path test mode: not a scaling measurement

    def leg(name, fn):
        """An extra leg must never cost the headline line: a failure is reported in place of its object."""
        try:
            result[name] = fn()
        except Exception as exc:   # noqa: BLE001
            result[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}

    if rank == 0 and not args.no_extras and (world == 1 or not tp):
        leg("roofline", lambda: roofline_gate_up(runner, elapsed / args.steps * 1e3))
        prefetch_mb = runner.prefetch_bytes / float(1 << 20)
        result["config"]["l2_prefetch_mib_per_row_kernel"] = prefetch_mb
        if world == 1:
            leg("w4a8_gemm_4096", lambda: gemm_4096(device))
            leg("mid_m", lambda: mid_m_leg(device))
            del runner
            torch.cuda.empty_cache()
            leg("protocol", lambda: protocol_leg(cfg, args, device))
            leg("drop_in", lambda: drop_in_leg(cfg, args, device))
            headline = args.model == "llama3-8b" and args.group_size == -1 and args.batch == 16
            if headline:
                leg("configs2_g128_bs64", lambda: configs2_leg(args, device))
                # the batch the reference's published A100 figure is quoted at (README.md:269,279; benchmark_a100.sh: bs 256)
                leg("protocol_bs128", lambda: protocol_leg(cfg, args, device, batch=128))
                leg("protocol_bs256", lambda: protocol_leg(cfg, args, device, batch=256))
            if not args.no_lserve and headline:
                leg("lserve_ctx256k", lambda: lserve_leg(device))
                torch.cuda.empty_cache()
                leg("llama2_70b_tp8_rank", lambda: tp_rank_leg(device))
                torch.cuda.empty_cache()
                # configs[4]'s model whole on this GPU (70B W4 = 35 GB): the N = 1 point of the strong-scaling curve
                leg("llama2_70b_tp1", lambda: tp1_leg(device))
                torch.cuda.empty_cache()
            leg("cpu_baseline", lambda: cpu_baseline(cfg, args.batch))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
