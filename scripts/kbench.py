"""Kernel micro-benchmark on the GPU box: times one encoder kernel at the DenseNet-121
layer shapes (batch 256) through the tuning hooks, on random data.

  python scripts/kbench.py [--variants 0,1] [--kernels c3,c1] [--batch 256] [--iters 20]
"""
import argparse, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tennis_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="0")
ap.add_argument("--kernels", default="c3,c1,tr")
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--pad", type=int, default=0)
ap.add_argument("--blocks", default="0,1,2,3")
ap.add_argument("--lib", default="")
ap.add_argument("--allk", action="store_true")
ap.add_argument("--stamps", action="store_true")
ap.add_argument("--ws", action="store_true", help="tr: the last transition on trans_ws.hip")
ap.add_argument("--b7stamps", type=int, default=5)
args = ap.parse_args()
if args.lib: _lib.LIB_PATH = os.path.abspath(args.lib)
ctx = _lib.default_context(0)
lib = ctx.lib
B = args.batch
rng = np.random.default_rng(0)

def timed(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us

blocks = [(56, 64, 6), (28, 128, 12), (14, 256, 24), (7, 512, 16)]
blocks = [blocks[int(i)] for i in args.blocks.split(',')]
res = []
for v in [int(s) for s in args.variants.split(",")]:
    if "c3" in args.kernels:
        w = rng.normal(0, 0.03, (32, 128, 3, 3)).astype(np.float32)
        wp = np.empty(2 * 72 * 64 * 8, np.uint16)   # both MFMA operand layouts
        lib.tn_dbg_pack_conv3x3(w.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p))
        wpd = torch.from_numpy(wp.view(np.int16)).cuda()
        sc = torch.rand(128, device="cuda") + 0.5; sh = torch.randn(128, device="cuda") * 0.3
        for (hw, cin, nl) in blocks:
            M = B * hw * hw
            x = torch.randn((M, 128), device="cuda", dtype=torch.float16)
            ctot = cin + 32 * nl
            y = torch.zeros((M, ctot), device="cuda", dtype=torch.float16)
            fn = lambda: _lib.check(lib.tn_dbg_conv3x3_dev(ctx.handle, _lib.ptr(x), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(wpd),
                                                           _lib.ptr(y), ctot, cin, B, hw, hw, v))
            us = timed(fn, args.iters)
            fl = 2.0 * M * 32 * 1152
            by = M * (128 + 32) * 2
            res.append(dict(k="c3", v=v, hw=hw, us=round(us, 1), tf=round(fl / us / 1e6, 1), tbs=round(by / us / 1e6, 2)))
            print(res[-1], flush=True)
            del x, y
    if "c1" in args.kernels:
        for (hw, cin, nl) in blocks:
            M = B * hw * hw
            ctot = cin + 32 * nl
            x = torch.randn((M, ctot), device="cuda", dtype=torch.float16)
            y = torch.zeros((M, 128), device="cuda", dtype=torch.float16)
            for K in sorted(set([cin, cin + 32 * (nl // 2), cin + 32 * (nl - 1)])):
                wd = (torch.randn((128, K), device="cuda") * (2.0 / K) ** 0.5).half()
                sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.3
                fn = lambda: _lib.check(lib.tn_dbg_conv1x1_dev(ctx.handle, _lib.ptr(x), ctot, K, _lib.ptr(sc), _lib.ptr(sh),
                                                               _lib.ptr(wd), 128, _lib.ptr(y), 128, 0, M, 0, hw, hw, v))
                us = timed(fn, args.iters)
                fl = 2.0 * M * 128 * K
                by = M * (K + 128) * 2
                res.append(dict(k="c1", v=v, hw=hw, K=K, us=round(us, 1), tf=round(fl / us / 1e6, 1), tbs=round(by / us / 1e6, 2)))
                print(res[-1], flush=True)
            del x, y
    if "tr" in args.kernels:
        for (hw, cin, nl) in blocks[:3]:
            M = B * hw * hw
            K = cin + 32 * nl
            N = K // 2
            Mo = M // 4
            x = torch.randn((M, K), device="cuda", dtype=torch.float16)
            y = torch.zeros((Mo, N), device="cuda", dtype=torch.float16)
            wd = (torch.randn((N, K), device="cuda") * (2.0 / K) ** 0.5).half()
            sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.3
            fn = lambda: _lib.check(lib.tn_dbg_conv1x1_dev(ctx.handle, _lib.ptr(x), K, K, _lib.ptr(sc), _lib.ptr(sh),
                                                           _lib.ptr(wd), N, _lib.ptr(y), N, 0, Mo, 1, hw, hw, v | ((1 << 18) if (args.ws and N in (256, 512)) else 0)))
            us = timed(fn, args.iters)
            by = M * K * 2 + Mo * N * 2
            res.append(dict(k="tr", v=v, hw=hw, K=K, us=round(us, 1), tf=round(2.0 * M * N * K / us / 1e6, 1), tbs=round(by / us / 1e6, 2)))
            print(res[-1], flush=True)
            del x, y
    if "dl" in args.kernels:
        w = rng.normal(0, 0.03, (32, 128, 3, 3)).astype(np.float32)
        wp = np.empty(2 * 72 * 64 * 8, np.uint16)   # both MFMA operand layouts
        lib.tn_dbg_pack_conv3x3(w.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p))
        wpd = torch.from_numpy(wp.view(np.int16)).cuda()
        s2 = torch.rand(128, device="cuda") + 0.5; t2 = torch.randn(128, device="cuda") * 0.3
        for (hw, cin, nl) in blocks:
            M = B * hw * hw
            ctot = cin + 32 * nl + args.pad
            buf = torch.randn((M, ctot), device="cuda", dtype=torch.float16)
            for K in (range(cin, cin + 32 * nl, 32) if args.allk else sorted(set([cin, cin + 32 * (nl // 2), cin + 32 * (nl - 1)]))):
                wd = (torch.randn((128, K), device="cuda") * (2.0 / K) ** 0.5).half()
                s1 = torch.rand(K, device="cuda") + 0.5; t1 = torch.randn(K, device="cuda") * 0.3
                fn = lambda: _lib.check(lib.tn_dbg_dense_layer_dev(ctx.handle, _lib.ptr(buf), ctot, K, _lib.ptr(s1), _lib.ptr(t1),
                                                                   _lib.ptr(wd), _lib.ptr(s2), _lib.ptr(t2), _lib.ptr(wpd), B, hw, hw, None, v))
                us = timed(fn, args.iters)
                big = ((v & 3) != 2) and hw != 7
                nwg = B * ({56: 8, 28: 2, 14: 1}[hw] if big else {56: 16, 28: 4, 14: 1, 7: 1}[hw])
                ts = torch.zeros((nwg * 12,), dtype=torch.int64, device="cuda")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.check(lib.tn_dbg_dense_layer_dev(ctx.handle, _lib.ptr(buf), ctot, K, _lib.ptr(s1), _lib.ptr(t1), _lib.ptr(wd),
                                                      _lib.ptr(s2), _lib.ptr(t2), _lib.ptr(wpd), B, hw, hw, _lib.ptr(ts), v))
                e1.record()
                torch.cuda.synchronize()
                us1 = e0.elapsed_time(e1) * 1e3
                tsn = ts.cpu().numpy().astype(np.float64)[:nwg * 8].reshape(nwg, 8)
                tsn = tsn[tsn[:, 0] > 0]          # persistent grids: only the first gridDim.x rows carry stamps
                nwg = len(tsn)
                d = np.diff(tsn[:, :7], axis=1)
                print("   phases(cycles, median): prologue %d | Kloop %d | pad %d | epiA+bar %d | phaseB %d | reduce+store %d | total %d"
                      % tuple(list(np.median(d, axis=0)) + [np.median(tsn[:, 6] - tsn[:, 0])]), flush=True)
                tot = tsn[:, 6] - tsn[:, 0]
                ok = tsn[:, 0] > 0
                xcd = np.arange(nwg) % 8
                span = np.median([tsn[xcd == x, 6].max() - tsn[xcd == x, 0].min() for x in range(8)])
                print("   total: mean %d p10 %d p90 %d max %d | span %d ticks in %.1f us -> %.2f ticks/ns | sum(total)/256 = %d"
                      % (tot[ok].mean(), np.percentile(tot[ok], 10), np.percentile(tot[ok], 90), tot[ok].max(), span, us1, span / us1 / 1e3, tot[ok].sum() / 256), flush=True)
                if not ok.all(): print("   WARNING: %d of %d WGs wrote no stamps" % ((~ok).sum(), len(ok)))
                fl = 2.0 * M * (128 * K + 32 * 1152)
                by = M * (K + 32) * 2
                res.append(dict(k="dl", v=v, hw=hw, K=K, us=round(us, 1), tf=round(fl / us / 1e6, 1), tbs=round(by / us / 1e6, 2)))
                print(res[-1], flush=True)
            del buf
    if "ds" in args.kernels:      # strip-streaming fused dense layer (dense_strip_impl.h), every layer of the 56x56 / 28x28 blocks it supports
        w3 = rng.normal(0, 0.03, (32, 128, 3, 3)).astype(np.float32)
        w3s = np.empty(36864, np.uint16)
        lib.tn_dbg_pack_strip(None, 32, None, None, None, w3.ctypes.data_as(C.c_void_p), w3s.ctypes.data_as(C.c_void_p))
        w3d = torch.from_numpy(w3s.view(np.int16)).cuda()
        s2 = (rng.random(128) + 0.5).astype(np.float32); t2 = (rng.normal(0, 0.3, 128)).astype(np.float32)
        for (hw, cin, nl) in blocks:
            if hw not in (56, 28): continue
            M = B * hw * hw
            ctot = cin + 32 * nl
            buf = torch.randn((M, ctot), device="cuda", dtype=torch.float16)
            tot = 0.0
            for K in range(cin, min(cin + 32 * nl, 321), 32):
                w1 = rng.normal(0, (2.0 / K) ** 0.5, (128, K)).astype(np.float32)
                w1s = np.empty((K + 16) * 128, np.uint16)
                lib.tn_dbg_pack_strip(w1.ctypes.data_as(C.c_void_p), K, s2.ctypes.data_as(C.c_void_p), t2.ctypes.data_as(C.c_void_p), w1s.ctypes.data_as(C.c_void_p), None, None)
                w1d = torch.from_numpy(w1s.view(np.int16)).cuda()
                s1 = torch.rand(K, device="cuda") + 0.5; t1 = torch.randn(K, device="cuda") * 0.3
                fn = lambda: _lib.check(lib.tn_dbg_dense_strip_dev(ctx.handle, _lib.ptr(buf), ctot, K, _lib.ptr(s1), _lib.ptr(t1),
                                                                   _lib.ptr(w1d), _lib.ptr(w3d), B, hw, hw, None))
                us = timed(fn, args.iters)
                tot += us
                if args.stamps:
                    ts = torch.zeros((B * 128,), dtype=torch.int64, device="cuda")
                    _lib.check(lib.tn_dbg_dense_strip_dev(ctx.handle, _lib.ptr(buf), ctot, K, _lib.ptr(s1), _lib.ptr(t1), _lib.ptr(w1d),
                                                          _lib.ptr(w3d), B, hw, hw, _lib.ptr(ts)))
                    torch.cuda.synchronize()
                    t = ts.cpu().numpy().astype(np.float64).reshape(B, 128)
                    n = int((t[0, :125] > 0).sum())
                    real_us = (t[:, 125] - t[:, 126]) / 100.0
                    print("   wave 0: %.1f us of wall clock (median; min %.1f max %.1f; launch span %.1f us), shader clock %.2f GHz"
                          % (np.median(real_us), real_us.min(), real_us.max(), (t[:, 125].max() - t[:, 126].min()) / 100.0,
                             np.median((t[:, n - 1] - t[:, 127]) / (real_us * 1e3))), flush=True)
                    d = np.diff(t[:, :n], axis=1)
                    med = np.median(d, axis=0)
                    print("   stamps (ticks, median over frames): launch->start %d | start->loop %d | rows: A %s | B %s | tail %s | total %d"
                          % (np.median(t[:, 0] - t[:, 127]), med[0] + (med[1] if n > 2 else 0), np.round(med[2:-2:2][:6]), np.round(med[3:-2:2][:6]), np.round(med[-2:]),
                             np.median(t[:, n - 1] - t[:, 127])), flush=True)
                    print("      mean A %.0f  mean B %.0f  (A slots %d, B slots 72)" % (med[2:-2:2].mean(), med[3:-2:2].mean(), K // 4), flush=True)
                fl = 2.0 * M * (128 * K + 32 * 1152)
                by = M * (K + 32) * 2
                res.append(dict(k="ds", hw=hw, K=K, us=round(us, 1), tf=round(fl / us / 1e6, 1), tbs=round(by / us / 1e6, 2)))
                print(res[-1], flush=True)
            print("   ds block %d: sum %.1f us" % (hw, tot), flush=True)
            del buf
    if "b7" in args.kernels:      # the LDS-resident 7x7 dense block (dense_block7.hip): the whole block in one launch
        K0, nl = 512, 16
        Ks = [K0 + 32 * l for l in range(nl)]
        cat = lambda xs: np.ascontiguousarray(np.concatenate([x.ravel() for x in xs]))
        w1 = cat([rng.normal(0, (2.0 / K) ** 0.5, (128, K)).astype(np.float32) for K in Ks])
        s1 = cat([(rng.random(K) + 0.5).astype(np.float32) for K in Ks]); t1 = cat([rng.normal(0, 0.3, K).astype(np.float32) for K in Ks])
        s2 = (rng.random((nl, 128)) + 0.5).astype(np.float32); t2 = rng.normal(0, 0.3, (nl, 128)).astype(np.float32)
        w3 = rng.normal(0, 0.03, (nl, 32, 128, 3, 3)).astype(np.float32)
        vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        _lib.check(lib.tn_dbg_block7_create(ctx.handle, K0, nl, vp(w1), vp(s1), vp(t1), vp(s2), vp(t2), vp(w3), C.byref(h)))
        buf = torch.randn((B * 49, 1024), device="cuda", dtype=torch.float16)
        fn = lambda: _lib.check(lib.tn_dbg_block7_run(h, _lib.ptr(buf), 1024, B))
        us = timed(fn, args.iters)
        fl = sum(2.0 * B * 49 * (128 * K + 32 * 1152) for K in Ks)
        if args.stamps:
            ts = torch.zeros((B * 128,), dtype=torch.int64, device="cuda")
            _lib.check(lib.tn_dbg_block7_run_ts(h, _lib.ptr(buf), 1024, B, _lib.ptr(ts)))
            torch.cuda.synchronize()
            t = ts.cpu().numpy().astype(np.float64).reshape(B, 128)
            real_us = (t[:, 126] - t[:, 127]) / 100.0
            ns = args.b7stamps
            d = np.diff(t[:, :1 + ns * nl], axis=1).reshape(B, nl, ns)
            med = np.median(d, axis=0)
            print("   wave 0: %.1f us wall (median; min %.1f max %.1f), shader clock %.2f GHz" % (np.median(real_us), real_us.min(), real_us.max(),
                  np.median((t[:, ns * nl] - t[:, 0]) / (real_us * 1e3))))
            print("   per layer (ticks): 1x1 | reduce | epilogue | 3x3 | append")
            for l in range(nl): print("   l%2d K=%4d  %s  sum %d" % (l, Ks[l], np.round(med[l]).astype(int), med[l].sum()))
            print("   totals %s = %d" % (np.round(med.sum(axis=0)).astype(int), med.sum()), flush=True)
        res.append(dict(k="b7", us=round(us, 1), tf=round(fl / us / 1e6, 1)))
        print(res[-1], flush=True)
        lib.tn_dbg_block7_destroy(h)
    if "b14" in args.kernels:     # the streamed 14x14 dense block (dense_block14.hip) next to the chained tile kernel (dl14)
        K0, nl = 256, 24
        Ks = [K0 + 32 * l for l in range(nl)]
        cat = lambda xs: np.ascontiguousarray(np.concatenate([x.ravel() for x in xs]))
        w1 = cat([rng.normal(0, (2.0 / K) ** 0.5, (128, K)).astype(np.float32) for K in Ks])
        s1 = cat([(rng.random(K) + 0.5).astype(np.float32) for K in Ks]); t1 = cat([rng.normal(0, 0.3, K).astype(np.float32) for K in Ks])
        s2 = (rng.random((nl, 128)) + 0.5).astype(np.float32); t2 = rng.normal(0, 0.3, (nl, 128)).astype(np.float32)
        w3 = rng.normal(0, 0.03, (nl, 32, 128, 3, 3)).astype(np.float32)
        vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
        h = C.c_void_p()
        _lib.check(lib.tn_dbg_block14_create(ctx.handle, K0, nl, vp(w1), vp(s1), vp(t1), vp(s2), vp(t2), vp(w3), C.byref(h)))
        buf = torch.randn((B * 196, 1024), device="cuda", dtype=torch.float16)
        fn = lambda: _lib.check(lib.tn_dbg_block14_run(h, _lib.ptr(buf), 1024, B))
        us = timed(fn, args.iters)
        fl = sum(2.0 * B * 196 * (128 * K + 32 * 1152) for K in Ks)
        if args.stamps:
            ts = torch.zeros((B * 160,), dtype=torch.int64, device="cuda")
            _lib.check(lib.tn_dbg_block14_run_ts(h, _lib.ptr(buf), 1024, B, _lib.ptr(ts)))
            torch.cuda.synchronize()
            tall = ts.cpu().numpy().astype(np.float64)
            t = tall[:B * 64].reshape(B, 64)
            t2 = tall[B * 64:].reshape(B, 96)
            if t2.any():      # a -DTN_B14_STAMPS build: per layer [start of tail, start of 3x3, end of 3x3]
                nsu_ = [(K - 32 + 63) // 64 for K in Ks]
                starts = np.concatenate([t[:, 63:64], t[:, :nl - 1]], axis=1)        # end of the previous layer's epilogue B
                su = np.median(t2[:, 0::3][:, :nl] - starts, axis=0); tl = np.median(t2[:, 1::3][:, :nl] - t2[:, 0::3][:, :nl], axis=0)
                bb = np.median(t2[:, 2::3][:, :nl] - t2[:, 1::3][:, :nl], axis=0); ep = np.median(t[:, :nl] - t2[:, 2::3][:, :nl], axis=0)
                print("   ticks per slot: super-steps %s" % " ".join("%.0f" % (su[l] / (32 * nsu_[l])) for l in range(nl)))
                print("                   tail (24)   %s" % " ".join("%.0f" % (tl[l] / 24) for l in range(nl)))
                print("                   3x3 (144)   %s" % " ".join("%.0f" % (bb[l] / 144) for l in range(nl)))
                print("   epilogue B (ticks)          %s" % " ".join("%.0f" % ep[l] for l in range(nl)))
            d = np.diff(np.concatenate([t[:, 63:64], t[:, :nl]], axis=1), axis=1)
            med = np.median(d, axis=0)
            nsu = [(K - 32 + 63) // 64 for K in Ks]
            print("   per layer (ticks, median over frames): prologue+l0 %d | %s" % (med[0], " ".join("%d" % x for x in med[1:])))
            print("   ticks per MFMA slot (layer l: %s slots): %s" % ("32 nsu + 168", " ".join("%.1f" % (med[l] / (32 * nsu[l] + 168)) for l in range(1, nl))))
            print("   whole kernel %d ticks" % np.median(t[:, 62] - t[:, 63]), flush=True)
        res.append(dict(k="b14", us=round(us, 1), tf=round(fl / us / 1e6, 1)))
        print(res[-1], flush=True)
        lib.tn_dbg_block14_destroy(h)
if "b28" in args.kernels:     # the streamed 28x28 dense block (dense_block28.hip)
    K0, nl = 128, 12
    Ks = [K0 + 32 * l for l in range(nl)]
    cat = lambda xs: np.ascontiguousarray(np.concatenate([x.ravel() for x in xs]))
    w1 = cat([rng.normal(0, (2.0 / K) ** 0.5, (128, K)).astype(np.float32) for K in Ks])
    s1 = cat([(rng.random(K) + 0.5).astype(np.float32) for K in Ks]); t1 = cat([rng.normal(0, 0.3, K).astype(np.float32) for K in Ks])
    s2 = (rng.random((nl, 128)) + 0.5).astype(np.float32); t2 = rng.normal(0, 0.3, (nl, 128)).astype(np.float32)
    w3 = rng.normal(0, 0.03, (nl, 32, 128, 3, 3)).astype(np.float32)
    vp = lambda a_: a_.ctypes.data_as(C.c_void_p)
    h = C.c_void_p()
    _lib.check(lib.tn_dbg_block28_create(ctx.handle, K0, nl, vp(w1), vp(s1), vp(t1), vp(s2), vp(t2), vp(w3), C.byref(h)))
    buf = torch.randn((B * 784, 512), device="cuda", dtype=torch.float16)
    fn = lambda: _lib.check(lib.tn_dbg_block28_run(h, _lib.ptr(buf), 512, B))
    us = timed(fn, args.iters)
    fl = sum(2.0 * B * 784 * (128 * K + 32 * 1152) for K in Ks)
    if args.stamps:
        ts = torch.zeros((B * 64,), dtype=torch.int64, device="cuda")
        _lib.check(lib.tn_dbg_block28_run_ts(h, _lib.ptr(buf), 512, B, _lib.ptr(ts)))
        torch.cuda.synchronize()
        t = ts.cpu().numpy().astype(np.float64).reshape(B, 64)
        d = np.diff(np.concatenate([t[:, 63:64], t[:, :nl]], axis=1), axis=1)
        med = np.median(d, axis=0)
        slots = [4 * (32 * ((K + 63) // 64) + 8 + 144) for K in Ks]
        print("   per layer (ticks, median over frames): %s" % " ".join("%d" % x for x in med))
        print("   ticks per MFMA slot: %s" % " ".join("%.1f" % (med[l] / slots[l]) for l in range(nl)))
        print("   whole kernel %d ticks" % np.median(t[:, 62] - t[:, 63]), flush=True)
    res.append(dict(k="b28", us=round(us, 1), tf=round(fl / us / 1e6, 1)))
    print(res[-1], flush=True)
    lib.tn_dbg_block28_destroy(h)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/kbench.json", "w"), indent=1)
