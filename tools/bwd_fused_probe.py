"""cpfn_mlp_bwd_fused against the cpfn_mlp_wgrad + cpfn_mlp_gemm(w_trans, bwd_stats) pair it replaces: results and time.
    python tools/bwd_fused_probe.py [P reps N K] [--apply-only]
(P >= 262144: the operands no longer fit the 256 MB infinity cache, the back-to-back launches are then fed from HBM)"""
import os, sys
import torch
sys.path.insert(0, os.getcwd())
from cpfn_amd import fused_mlp, lib as _l
from cpfn_amd.ops import _ptr, _stream
_pos = [a for a in sys.argv[1:] if not a.startswith("--")][:4]
P, reps, N, K = (int(v) for v in (_pos + ["131072", "50", "128", "128"][len(_pos):]))
dev = torch.device("cuda:0")
h = _l.lib()
torch.manual_seed(0)
BF = torch.bfloat16
Gy = torch.randn(P, N, device=dev).to(BF)
A = torch.randn(P, K, device=dev).to(BF)
Wb = (torch.randn(N, K, device=dev) * 0.1).to(BF)
Yp = torch.randn(P, K, device=dev).to(BF)
asc, ash = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
bsc, bsh = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1
splits = h.cpfn_mlp_wgrad_splits(P, N, K)


def separate(stats, atr):
    ws = torch.empty(splits * N * K, dtype=torch.float32, device=dev)
    _l.check(h.cpfn_mlp_wgrad(_ptr(Gy), N, _ptr(A), K, None, P, N, K, _ptr(asc) if atr else None, _ptr(ash) if atr else None,
                              _ptr(ws), None, _stream()), "wgrad")
    if stats:
        g, part, nb = fused_mlp.gemm(Gy, Wb, w_trans=True, bwd_stats=(Yp, bsc, bsh))
        return ws, g, part[:nb].sum(0)
    g, _, _ = fused_mlp.gemm(Gy, Wb, w_trans=True)
    return ws, g, None


def fused(stats, atr):
    ws = torch.empty(splits * N * K, dtype=torch.float32, device=dev)
    g = torch.empty(P, K, dtype=BF, device=dev)
    part = torch.empty(splits, 2, K, dtype=torch.float32, device=dev) if stats else None
    _l.check(h.cpfn_mlp_bwd_fused(_ptr(Gy), N, _ptr(A), K, _ptr(Wb), P, N, K, _ptr(asc) if atr else None, _ptr(ash) if atr else None,
                                  _ptr(ws), _ptr(g), K, _ptr(Yp) if stats else None, _ptr(bsc) if stats else None,
                                  _ptr(bsh) if stats else None, _ptr(part), None, None, None, None, None, 0.0, None, None, 0,
                                  None, None, _stream()), "fused")
    return ws, g, None if part is None else part.sum(0)


Yr = torch.randn(P, N, device=dev).to(BF)
coef = torch.randn(3, N, device=dev) * 0.1
ysc, ysh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1


def fused_apply():
    """the shape the replayed step runs: BatchNorm-backward apply on the staged chunks + riding reduction of the layer below"""
    ws = torch.empty(splits * N * K, dtype=torch.float32, device=dev)
    g = torch.empty(P, K, dtype=BF, device=dev)
    part = torch.empty(splits, 2, K, dtype=torch.float32, device=dev)
    _l.check(h.cpfn_mlp_bwd_fused(_ptr(Gy), N, _ptr(A), K, _ptr(Wb), P, N, K, _ptr(asc), _ptr(ash), _ptr(ws), _ptr(g), K, _ptr(Yp),
                                  _ptr(bsc), _ptr(bsh), _ptr(part), _ptr(Yr), _ptr(coef), _ptr(ysc), _ptr(ysh), None, 0.0, None, None,
                                  0, None, None, _stream()), "fused apply")
    return ws


def timeit(f, *a):
    for _ in range(5):
        f(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f(*a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("P=%d splits=%d ok=%d" % (P, splits, h.cpfn_mlp_bwd_fused_ok(P, N, K)))
if "--apply-only" in sys.argv:
    t = timeit(fused_apply)
    mb = (4 * P * N + 4 * P * K + 4 * splits * N * K + 2 * P * K) / 1e6
    print("     fused with apply + riding reduction: %.1f us  (%.0f MB algorithmic: %.2f TB/s)" % (t, mb, mb / t))
    sys.exit(0)
for stats in (0, 1):
    for atr in (0, 1):
        w0, g0, s0 = separate(stats, atr)
        w1, g1, s1 = fused(stats, atr)
        dw0, dw1 = w0.view(splits, -1).sum(0), w1.view(splits, -1).sum(0)
        msg = "stats=%d atr=%d  dW rel %.2e (bitwise partials %s)  g mismatch %d / %d (max %.3g)" % (
            stats, atr, ((dw0 - dw1).abs().max() / dw0.abs().max()).item(), bool(torch.equal(w0, w1)),
            int((g0 != g1).sum()), g0.numel(), (g0.float() - g1.float()).abs().max().item())
        if stats:
            msg += "  stats rel %.2e" % ((s0 - s1).abs().max() / s0.abs().max()).item()
        print(msg)
        print("     separate %.1f us   fused %.1f us" % (timeit(separate, stats, atr), timeit(fused, stats, atr)))
