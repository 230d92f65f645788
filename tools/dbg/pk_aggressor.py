"""Which part of a training step disturbs the packed-fp32 sampling kernel?  (Needs a library built with the in-kernel invariant of
round 4 — the sample's own min-distance must be 0 after the distance update, counted in g_fps_dbg / cpfn_dbg_fps_read; a scratch
tree, not the product.)  The packed instantiation (ops.fps OUTSIDE background_geometry) loops on a side stream while ONE candidate
loops on the main stream; per candidate: invariant violations and launches whose indices differ from the quiet run's.
    python tools/dbg/pk_aggressor.py [seconds per candidate] [stacks | kernels | ablate]
(stacks: one MLP stack's forward + backward at a time; kernels: one backward kernel at a time; ablate: a tools/dbg/fps_ablate.py build,
the real kernel beside mlp_wgrad under one run-time ablation mask at a time)"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.getcwd())
from cpfn_amd import lib as _l, ops, synthetic, training          # noqa: E402
from cpfn_amd.PointNet2 import pn2_network                        # noqa: E402


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
    dev = torch.device("cuda:0")
    B, N = 4, 2048
    h = _l.lib()
    h.cpfn_dbg_fps_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_int * 64)()
    g = torch.Generator().manual_seed(1)
    xyz = (torch.rand(B, N, 3, generator=g) * 2 - 1).to(dev)
    start = torch.randint(0, N, (B,), generator=g).to(torch.int32).to(dev)
    ref = ops.fps(xyz, 512, start).clone()

    torch.manual_seed(0)
    m = pn2_network.PointNet2(dim_input=3, dim_pos=3, output_sizes=[2]).to(dev)
    m.set_compute_dtype(torch.bfloat16)
    tr = training.PatchSelectionTrainer(m, batch_size=B, use_graphs=True)
    c = synthetic.primitive_cloud(B, N, n_prims=6, seed=600)
    batch = {"P": c["P"].to(dev), "labels": (c["I_gt"] % 2).long().to(dev)}
    m.train()
    tr.step(batch, force_eager=True)
    flags = torch.zeros(4, dtype=torch.int32, device=dev)
    err = torch.zeros(4, dtype=torch.int32).pin_memory()
    big_a, big_b = torch.randn(1 << 22, device=dev), torch.empty(1 << 22, device=dev)
    dbl = torch.randn(1 << 20, device=dev, dtype=torch.float64)

    def forward_only():
        with torch.no_grad():
            for _ in range(4):
                m(batch["P"])

    def eager_step():
        for _ in range(2):
            tr.step(batch, force_eager=True)

    def forward_backward():
        for _ in range(2):
            tr.bucket.zero()
            tr.losses(batch)[0].backward()

    def optimizer_only():
        for _ in range(40):
            tr._checked_optimizer_step(tr._skip_counter(dev))

    def flag_spin():
        s = torch.cuda.current_stream().cuda_stream
        for _ in range(4):          # nobody sets the flag: each waiter spins (s_sleep, clock reads, atomic loads) for 1 ms
            h.cpfn_flag_wait(flags[2:].data_ptr(), 1, 100_000, err.data_ptr(), None, s)

    def copies():
        for _ in range(40):
            big_b.copy_(big_a)

    def fp64_math():
        for _ in range(40):
            (dbl * dbl + dbl).sum()

    def nothing():
        time.sleep(0.004)

    # ---- "stacks" mode: forward + backward of ONE MLP stack at a time, in the shapes this configuration's step has
    def stack_candidate(cin, widths, P, pool_k=None, xyz=False, tail=False, dropout=False, reps=2):
        sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
        from test_gpu_fused_mlp import _stack
        from cpfn_amd import mlp
        convs, bns = _stack(3 if xyz else cin + (3 if tail else 0), widths, seed=29)
        gg = torch.Generator().manual_seed(5)
        x = (torch.rand(P, 3, generator=gg) * 0.4 - 0.2).to(dev) if xyz else torch.randn(P, cin, generator=gg).to(dev).to(torch.bfloat16).requires_grad_(True)
        tl = (torch.rand(P, 3, generator=gg) * 0.4 - 0.2).to(dev) if tail else None
        gout = torch.randn(P // pool_k if pool_k else P, widths[-1], generator=gg).to(dev)
        drop = (0.5, torch.zeros(1, dtype=torch.int64, device=dev), 1234567) if dropout else None
        params = [c_.weight for c_ in convs] + [p_ for b_ in bns for p_ in (b_.weight, b_.bias)]

        def fn():
            for _ in range(reps):
                for p_ in params:
                    p_.grad = None
                y = mlp.run_stack(None if xyz else x, convs, bns, torch.bfloat16, pool_k=pool_k, xyz_rows=x if xyz else None,
                                  dropout=drop, xyz_tail=tl)
                (y.float() * gout).sum().backward()
        return fn

    if len(sys.argv) > 2 and sys.argv[2] == "stacks":
        cands = (("sa1-like: xyz rows, 64/64/128, pooled (131072 rows)", stack_candidate(3, [64, 64, 128], 131072, 64, xyz=True)),
                 ("sa2-like: 128 + xyz tail, 128/128/256, pooled (32768 rows)", stack_candidate(128, [128, 128, 256], 32768, 64, tail=True, reps=3)),
                 ("sa3-like: 256 -> 256/512/1024, pooled (512 rows)", stack_candidate(256, [256, 512, 1024], 512, 128, reps=6)),
                 ("sfp-like: 384 -> 256/256 (2048 rows)", stack_candidate(384, [256, 256], 2048, reps=6)),
                 ("sfp3-like: 128 -> 128/128/128 (8192 rows)", stack_candidate(128, [128, 128, 128], 8192, reps=4)),
                 ("fc1-like: 128 -> 128 with dropout (8192 rows)", stack_candidate(128, [128], 8192, dropout=True, reps=8)),
                 ("sfp3-like at 131072 rows (one-pass kernels)", stack_candidate(128, [128, 128, 128], 131072, reps=1)))
    elif len(sys.argv) > 2 and sys.argv[2] == "kernels":
        # ---- "kernels" mode: the launches of one generic-route layer's backward (8192 rows, 128 -> 128), one kind at a time
        from cpfn_amd import fused_mlp as fm
        P_, N_ = 8192, 128
        gk = torch.Generator().manual_seed(7)
        bf = lambda *sh: torch.randn(*sh, generator=gk).to(dev).to(torch.bfloat16)
        f32 = lambda *sh: torch.randn(*sh, generator=gk).to(dev)
        g_, Y_, A_, Gy_ = bf(P_, N_), bf(P_, N_), bf(P_, N_), torch.empty(P_, N_, dtype=torch.bfloat16, device=dev)
        sc, sh, gam, mean, rstd = f32(N_), f32(N_), f32(N_), f32(N_), f32(N_).abs() + 0.5
        nblk = h.cpfn_bn_bwd_blocks(P_)
        part, coef, dg, db = torch.empty(nblk, 2, N_, device=dev), torch.empty(3, N_, device=dev), torch.empty(N_, device=dev), torch.empty(N_, device=dev)
        splits = h.cpfn_mlp_wgrad_splits(P_, N_, N_)
        ws, dW = torch.empty(splits * N_ * N_, device=dev), torch.empty(N_, N_, device=dev)
        Wb = bf(N_, N_)
        st_ = lambda: torch.cuda.current_stream().cuda_stream
        pt = lambda t: t.data_ptr()

        def k_relu_bwd():
            for _ in range(60):
                _l.check(h.cpfn_bn_relu_bwd(pt(g_), pt(Y_), pt(sc), pt(sh), P_, N_, None, pt(part), None, 0.0, st_()), "relu_bwd")

        def k_finalize():
            for _ in range(60):
                _l.check(h.cpfn_bn_bwd_finalize(pt(part), nblk, N_, float(P_), pt(gam), pt(mean), pt(rstd), 1, pt(dg), pt(db), pt(coef), st_()), "fin")

        def k_apply():
            for _ in range(60):
                _l.check(h.cpfn_bn_bwd_apply(pt(g_), pt(Y_), pt(coef), pt(sc), pt(sh), P_, N_, pt(Gy_), None, 0.0, st_()), "apply")

        def k_wgrad():
            for _ in range(60):
                _l.check(h.cpfn_mlp_wgrad(pt(Gy_), N_, pt(A_), N_, None, P_, N_, N_, None, None, pt(ws), None, st_()), "wgrad")

        def k_reduce():
            arr = (fm._ReduceDesc * 1)(fm._ReduceDesc(ws.data_ptr(), dW.data_ptr(), N_ * N_, splits, 0, 0))
            for _ in range(60):
                _l.check(h.cpfn_multi_split_reduce(arr, 1, st_()), "reduce")

        def k_dgrad():
            for _ in range(60):
                fm.gemm(Gy_, Wb, w_trans=True)

        def k_fwd_gemm():
            for _ in range(60):
                fm.gemm(A_, Wb, stats=True)

        k_relu_bwd(); k_finalize(); k_apply()
        cands = (("bn_relu_bwd (pass 1 reduction)", k_relu_bwd), ("bn_bwd_finalize (fp64 sums)", k_finalize), ("bn_bwd_apply", k_apply),
                 ("mlp_wgrad", k_wgrad), ("multi_split_reduce", k_reduce), ("data-gradient GEMM (w_trans)", k_dgrad),
                 ("forward GEMM with statistics", k_fwd_gemm))
    elif len(sys.argv) > 2 and sys.argv[2] in ("ablate", "ablate0", "diag"):
        # ---- "ablate" mode (a tools/dbg/fps_ablate.py build): the real packed kernel beside mlp_wgrad, one ablation mask at a time
        # (the "kernels" mode's aggressor, the one round 4 found: 8192 rows, 128 -> 128 = the 64 x 64-tile weight-gradient kernel,
        #  9 KB of LDS and 116 registers, which FITS beside a sampling workgroup.  Round 5's first run used 131072 rows: the
        #  128 x 128-tile kernel, 252 registers at 8 waves — it cannot share a compute unit with anything, and nothing failed.)
        P_, N_ = int(os.environ.get("PK_ROWS", "8192")), 128
        gk = torch.Generator().manual_seed(7)
        Gy_ = torch.randn(P_, N_, generator=gk).to(dev).to(torch.bfloat16)
        splits = h.cpfn_mlp_wgrad_splits(P_, N_, N_)
        ws = torch.empty(splits * N_ * N_, device=dev)

        def k_wgrad():
            for _ in range(60):
                _l.check(h.cpfn_mlp_wgrad(Gy_.data_ptr(), N_, Gy_.data_ptr(), N_, None, P_, N_, N_, None, None, ws.data_ptr(), None,
                                          torch.cuda.current_stream().cuda_stream), "wgrad")
        h.cpfn_dbg_fps_abl.argtypes = [ctypes.c_int]
        side = torch.cuda.Stream()
        import struct
        f32 = lambda v: struct.unpack("f", struct.pack("i", v))[0]
        for mask in ((0, 1, 2, 4, 8, 16, 32, 64, 2 | 4, 8 | 16, 1 | 2 | 4 | 8 | 16, 127) if sys.argv[2] == "ablate" else (0,)):
            h.cpfn_dbg_fps_abl(mask)
            torch.cuda.synchronize()
            ref_m = ops.fps(xyz, 512, start).clone()               # quiet run under this mask (masks 1 and 32 change what is written)
            h.cpfn_dbg_fps_read(buf, 1)
            t0, launches, bad = time.time(), 0, torch.zeros((), dtype=torch.int32, device=dev)
            while time.time() - t0 < secs:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(16):
                        bad += (ops.fps(xyz, 512, start) != ref_m).any().int()
                        launches += 1
                k_wgrad()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
            h.cpfn_dbg_fps_read(buf, 0)
            print("mask %3d: %6d sampling launches, %4d with different indices%s, %4d invariant violations" %
                  (mask, launches, int(bad), " (not meaningful: nothing / a schedule is written)" if mask & 33 else "", buf[8]), flush=True)
            if sys.argv[2] == "diag":
                print("  events by 16-lane row %s; packed d == distance to the PREVIOUS sample %d; packed d wrong %d; minimum not applied to a "
                      "right d %d" % (list(buf[0:4]), buf[4], buf[5], buf[6]))
                for k in range(min(buf[8], 6)):
                    r = buf[16 + 8 * k:24 + 8 * k]
                    print("  sample %4d thread %4d (lane %2d) slot %2d: packed d %.9g (0x%08x) scalar d %.9g (0x%08x) d to previous sample %.9g "
                          "old min %.9g new min %.9g" % (r[0], r[1], r[1] % 64, r[2], f32(r[3]), r[3] & 0xFFFFFFFF, f32(r[4]), r[4] & 0xFFFFFFFF,
                                                         f32(r[5]), f32(r[6]), f32(r[7])))
        h.cpfn_dbg_fps_abl(0)
        return
    else:
        cands = None
    side = torch.cuda.Stream()
    for name, fn in cands or (("nothing", nothing), ("flag wait spinning", flag_spin), ("optimizer", optimizer_only), ("copies", copies),
                     ("fp64 elementwise + reduction", fp64_math), ("forward only", forward_only), ("forward + backward", forward_backward),
                     ("eager step", eager_step)):
        h.cpfn_dbg_fps_read(buf, 1)
        torch.cuda.synchronize()
        t0, launches, bad = time.time(), 0, torch.zeros((), dtype=torch.int32, device=dev)
        while time.time() - t0 < secs:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(16):
                    idx = ops.fps(xyz, 512, start)
                    bad += (idx != ref).any().int()
                    launches += 1
            fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
        h.cpfn_dbg_fps_read(buf, 0)
        print("%-32s %6d sampling launches, %3d with different indices, %3d invariant violations %s" %
              (name, launches, int(bad), buf[8], [(buf[16 + 8 * k], buf[17 + 8 * k] % 64) for k in range(min(buf[8], 6))]), flush=True)


if __name__ == "__main__":
    main()
