"""why does the event-bracketed profile pass see 129 or 149 us for the 56x56 layers?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tennis_amd import _lib, weights as W
from tennis_amd.engine import DenseNet121Features
p = W.make_densenet121_weights(0)
enc = DenseNet121Features(p, 224, max_batch=256)
x = torch.randn(256, 224, 224, 3, device="cuda").half()
out = torch.empty(256, 1024, device="cuda")
def fam(tag):
    r = []
    for _ in range(3):
        stats, _ = enc.profile(x)
        d = {s["name"]: s for s in stats}
        r.append(round(d["dense_layer_fused_56x56"]["ms"] / d["dense_layer_fused_56x56"]["launches"] * 1e3, 1))
    print(tag, "56x56 us per launch in three profile passes:", r, flush=True)
def run(n, pipelined):
    enc.set_pipelined(pipelined)
    for i in range(n):
        enc(x, out=out)
    if pipelined: enc.join(0)
    torch.cuda.synchronize()
fam("cold")
run(100, True); fam("after 100 pipelined forwards")
run(100, False); fam("after 100 joined forwards")
enc.set_pipelined(True); fam("after set_pipelined(True), no forwards")
run(100, True); fam("after 100 pipelined forwards again")
enc.set_pipelined(False); fam("after set_pipelined(False)")
