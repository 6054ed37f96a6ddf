"""Numerical emulation (numpy, CPU) of VERDICT r2 item 4 for the KV4 decode attention: MFMA operands = raw codes as fp16
(`(x & mask) | magic` = bias + code, no per-element dequant), the per-token affine moved to the 32 x 16 score tile
(x = s_t (S_raw - bias*sum(q)) + c_t sum(q), c_t = h(-s_t z_t) as the reference rounds it) and folded into P for the P.V
product (P' = h(p s_t); O = O_raw - bias * sum(P') + sum(p c_t)).  Compared with the f32 oracle arithmetic the GPU tests use
(tests/test_edge_cases_gpu.py: per-head relative L2 and max |d| / head max, bar 1e-3) next to the shipped arithmetic
(bit-exact fp16 dequant, P rounded to fp16).  One wave sweeping all T tokens (pessimistic for the bias cancellation: the
kernel spreads T over splits x 4 waves).  Result (profiles/r03_g_experiments.md): the folded form costs 4e-4 .. 8.5e-4 of
the 1e-3 bar at T = 1024 / 1535 with bias 64 where the shipped arithmetic costs 3e-4 -- the unrounded V / P' products and
the cancellation against bias * sum(P') eat the margin; with the 1024 bias of the low-nibble magic it fails outright at long
T.  Not built.
    python tools/attn_fold_emulation.py <seed> <T> <q scale> <K bias> <V bias> [peak]
"""
import numpy as np, sys
sys.path.insert(0, '/root/repo')
F32, F16 = np.float32, np.float16
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
qs = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
BIASK = float(sys.argv[4]) if len(sys.argv) > 4 else 64.0
BIASV = float(sys.argv[5]) if len(sys.argv) > 5 else 64.0
peak = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
D, G = 128, 4
ek = rng.integers(0, 16, (T, D)).astype(F32); ev = rng.integers(0, 16, (T, D)).astype(F32)
sk = (0.05 + 0.15 * rng.random(T)).astype(F16); zk = (6 + 3 * rng.random(T)).astype(F16)
sv = (0.05 + 0.15 * rng.random(T)).astype(F16); zv = (6 + 3 * rng.random(T)).astype(F16)
q = (qs * rng.standard_normal((G, D))).astype(F16)
if peak: q[:, :] = (q.astype(F32) + peak * (ek[T // 3] - 7.5) / 10).astype(F16)
def deq(e, s, z):
    s32, z32 = s.astype(F32), z.astype(F32)
    c = (-(s32) * z32).astype(F32).astype(F16)
    return (e.astype(np.float64) * s32.astype(np.float64)[:, None] + c.astype(np.float64)[:, None]).astype(F16).astype(F32), c
K, ck = deq(ek, sk, zk); V, cv = deq(ev, sv, zv)
inv = F32(1 / np.sqrt(128))
# reference f32
S = (K @ q.astype(F32).T).astype(F32) * inv           # [T, G]
m = S.max(0); e = np.exp(S - m); p = e / (e.sum(0) + 1e-6)
ref = (p.T.astype(np.float64) @ V.astype(np.float64))   # [G, D]
# folded emulation: one wave, all tiles
q32 = q.astype(F32)
def mfma_acc(acc, A, Bm):   # A [M,k] B [k,N] exact products, one rounding per instruction
    return (acc.astype(np.float64) + A.astype(np.float64) @ Bm.astype(np.float64)).astype(F32)
Q0 = (BIASK * q32.astype(np.float64).sum(1)).astype(F32)     # per head
Sq = q32.astype(np.float64).sum(1).astype(F32)
oacc = np.zeros((D, G), F32); sp = np.zeros(G, F32); spc = np.zeros(G, F32)
m_run = np.full(G, -1e30, F32); l_run = np.zeros(G, F32)
for t0 in range(0, T, 32):
    t1 = min(T, t0 + 32)
    hk = (ek[t0:t1] + F32(BIASK)).astype(F16)          # exact
    acc = np.zeros((t1 - t0, G), F32)
    for d0 in range(0, D, 32):
        acc = mfma_acc(acc, hk[:, d0:d0 + 32], q[:, d0:d0 + 32].T)
    x = ((acc - Q0[None, :]) * sk[t0:t1].astype(F32)[:, None] + ck[t0:t1].astype(F32)[:, None] * Sq[None, :]).astype(F32) * inv
    tmax = x.max(0); m_new = np.maximum(m_run, tmax); alpha = np.exp(m_run - m_new).astype(F32)
    pp = np.exp(x - m_new).astype(F32)
    l_run = l_run * alpha + pp.sum(0)
    spc = spc * alpha + (pp * cv[t0:t1].astype(F32)[:, None]).sum(0)
    pv = (pp * sv[t0:t1].astype(F32)[:, None]).astype(F16)        # P' fp16
    oacc = oacc * alpha[None, :]; sp = sp * alpha
    hv = (ev[t0:t1] + F32(BIASV)).astype(F16)
    oacc = mfma_acc(oacc, hv.T, pv)
    sp = mfma_acc(sp[None, :], np.ones((1, t1 - t0), F16), pv)[0]
    m_run = m_new
out = ((oacc - F32(BIASV) * sp[None, :] + spc[None, :]) / (l_run + 1e-6)).T
# current-kernel style emulation (fp16 dequant, P fp16)
def errs(got):
    num = np.sqrt(((got - ref) ** 2).sum(-1)); den = np.sqrt((ref ** 2).sum(-1))
    return (num / den).max(), (np.abs(got - ref) / np.abs(ref).max(-1, keepdims=True)).max()
print("fold  :", errs(out.astype(F16).astype(np.float64)))
pf = (e.astype(F32)).astype(F16).astype(F32)
cur = (pf.T.astype(np.float64) @ V.astype(np.float64)) / (pf.sum(0)[:, None] + 1e-6)
print("cur   :", errs(cur.astype(F16).astype(np.float64)))
print("ref16 :", errs(ref.astype(F16).astype(np.float64)), "pmax", p.max(0))
