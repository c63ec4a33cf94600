import ctypes as C, sys, numpy as np, torch
sys.path.insert(0, '.')
from tennis_amd import _lib
ctx = _lib.default_context(0)
B,H,K,ldc=1,28,128,512
buf = np.ones((B,H,H,ldc), np.float16)
lo = np.full(K,-65504,np.float32); hi = (0.25 + np.arange(K)/256).astype(np.float16).astype(np.float32)
s2=np.ones(128,np.float32); t2=np.zeros(128,np.float32)
w1=np.eye(128,K).astype(np.float16)
for perm in (0, 32, 64, 96):
    w3=np.zeros((32,128,3,3),np.float32)
    for c in range(32): w3[c,perm+c,1,1]=1.0
    wp = np.empty(2 * 72 * 64 * 8, np.uint16)
    ctx.lib.tn_dbg_pack_conv3x3(w3.ctypes.data_as(C.c_void_p), wp.ctypes.data_as(C.c_void_p))
    d = dict(buf=torch.from_numpy(buf).cuda(), s1=torch.from_numpy(lo).cuda(), t1=torch.from_numpy(hi).cuda(),
             s2=torch.from_numpy(s2).cuda(), t2=torch.from_numpy(t2).cuda(), w1=torch.from_numpy(w1).cuda(), wp=torch.from_numpy(wp.view(np.int16)).cuda())
    _lib.check(ctx.lib.tn_dbg_dense_layer_dev(ctx.handle, _lib.ptr(d["buf"]), ldc, K, _lib.ptr(d["s1"]), _lib.ptr(d["t1"]), _lib.ptr(d["w1"]), _lib.ptr(d["s2"]),
                                              _lib.ptr(d["t2"]), _lib.ptr(d["wp"]), B, H, H, None, 1), "dense_layer")
    out = d["buf"].cpu().numpy().astype(np.float32)[0,:,:,K:K+32]
    want = hi[perm:perm+32]
    bad = np.argwhere(np.abs(out-want)>1e-3)
    print("perm",perm,"bad count",len(bad))
    if len(bad):
        ys=sorted(set(bad[:,0].tolist())); xs=sorted(set(bad[:,1].tolist())); cs=sorted(set(bad[:,2].tolist()))
        print(" rows",ys,"cols",xs,"chans",cs)
        y,x=bad[0][0],bad[0][1]
        print(" pixel",y,x,"got",np.round(out[y,x],4).tolist()); print(" want",np.round(want,4).tolist())
