"""-m "not gpu": the hand-counted s_waitcnt vmcnt(N) constants of dense_block28.hip, proven by replaying the kernel's issue order
(the method of tests/test_cpu_block14.py: loads complete in order, so vmcnt(N) is correct iff at least N loads were issued BEHIND
the one it needs).  The 28x28 kernel walks a frame in four passes per layer: its stages are (layer, pass) pairs, the activation
ring runs through pass and layer boundaries (two-super-step layers wrap through TWO stages), there is no tail interval (the shift
k-step opens a pass' first 1x1 unit, no loads), and the last 1x1 unit of a pass runs straight into the six 3x3 units.  A change of
the slot layout in the kernel has to be mirrored here (the comments name the lambdas)."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tennis_amd", "csrc", "dense_block28.hip")


def constants():
    text = open(SRC).read()
    c = {k: int(v) for k, v in re.findall(r"\b(kVm\w+) = (\d+)", text)}
    c["kNR"] = int(re.search(r"constexpr int kNR = (\d+)", text).group(1))
    c["kPreItems"] = int(re.search(r"constexpr int kPreItems = (\d+)", text).group(1))
    return c


class Wave:
    def __init__(self, c):
        self.c = c
        self.n = 0                      # loads issued so far
        self.done_upto = 0              # loads [0, done_upto) known complete (a vmcnt(0))
        self.ring = {}                  # (rs, kq, f) -> (issue index, (stage, su, kq))
        self.dma = {}                   # unit -> issue index of this wave's LAST piece
        self.next_unit = 0
        self.min_slack = {}

    def load(self):
        self.n += 1
        return self.n - 1

    def need(self, idx, vm, what):
        younger = self.n - 1 - idx
        assert idx < self.done_upto or younger >= vm, f"{what}: vmcnt({vm}) with only {younger} loads behind the one it waits for"
        if idx >= self.done_upto:
            k = what.split(":")[0]
            self.min_slack[k] = min(self.min_slack.get(k, 1 << 30), younger - vm)

    def dma_pair(self):
        self.load(); self.load()

    def dma_consts(self):
        self.dma[self.next_unit] = self.load()
        self.next_unit += 1

    def ring_load(self, rs, kq, f, holds):
        self.ring[(rs, kq, f)] = (self.load(), holds)

    def ring_wait(self, rs, kq, expect):
        for f in (0, 1):
            idx, holds = self.ring[(rs, kq, f)]
            assert holds == expect, f"ring[{rs}][{kq}][{f}] holds {holds}, its consumer expects {expect}"
            self.need(idx, self.c["kVmRing"], "ring: %s" % (expect,))

    def begin_interval(self, g, vm):
        assert g + 1 in self.dma, f"unit {g + 1} was never copied"
        self.need(self.dma[g + 1], vm, "dma: unit %d" % (g + 1))


def nsu_of(K):
    return (K + 63) // 64


def run_block(K0, nl, c):
    w = Wave(c)
    stages = [(l, p) for l in range(nl) for p in range(4)]
    for _ in range(4):                   # prologue: units 0 .. 3 (unit 0: the first pass' shift fragments)
        w.dma_pair(); w.dma_pair(); w.dma_consts()

    def nsu_stage(si):                   # (past the block: the kernel keeps computing targets from K0 + 32 l; never consumed)
        return nsu_of(K0 + 32 * (si // 4))

    def wrap(si, su):                    # `target` / the J = 5 refill base: super-step su of stage si, or of the stages behind it
        while su >= nsu_stage(si):
            su -= nsu_stage(si)
            si += 1
        return si, su

    for rs in (0, 1):                    # the first pass' super-steps 0 / 1
        for kq in range(4):
            for f in (0, 1):
                w.ring_load(rs, kq, f, wrap(0, rs) + (kq,))
    w.done_upto = w.n

    def pre_items(pn, si):               # b_interval J = 5 / prologue: BN of (stage si, super-step 0, k-step 0); refill <- ITS super-step 2
        w.ring_wait(pn, 0, wrap(si, 0) + (0,))
        for f in (0, 1):
            w.ring_load(pn, 0, f, wrap(si, 2) + (0,))
    pre_items(0, 0)
    w.dma_pair(); w.dma_pair(); w.dma_consts()      # "interval 0" (unit 0 consumed by the prologue) copies unit 4
    w.done_upto = w.n
    g, par = 1, 0
    for si, (l, p) in enumerate(stages):
        nsu = nsu_of(K0 + 32 * l)
        assert nsu >= 2
        for u in range(nsu):             # su_interval
            kind = 0 if u == 0 else (2 if u == nsu - 1 else 1)
            rs = (u + par) & 1
            w.begin_interval(g, c["kVmDmaSU0"] if kind == 0 else c["kVmDmaSU"])
            ta, tb = wrap(si, u + 2), wrap(si, u + 3)
            # the kernel's straight-line `target` wraps at most twice, and only with the NEXT stage's count for the second wrap
            for t_, d in ((ta, 2), (tb, 3)):
                uu, ss = u + d, 0
                n1 = nsu if p < 3 else nsu_of(K0 + 32 * (l + 1))
                if uu >= nsu:
                    uu -= nsu; ss = 1
                    if uu >= n1:
                        uu -= n1; ss = 2
                assert (si + ss, uu) == t_, ((si, u, d), (si + ss, uu), t_)
            for q in range(4):
                for e in range(8):
                    j, bf = e >> 1, e & 1
                    if q < 3:
                        if e == 0:
                            w.ring_wait(rs, q + 1, (si, u, q + 1))
                        if j == 3:
                            w.ring_load(rs, q + 1, bf, ta + (q + 1,))
                    elif kind != 2:
                        if e == 0:
                            w.ring_wait(rs ^ 1, 0, wrap(si, u + 1) + (0,))
                        if j == 3:
                            w.ring_load(rs ^ 1, 0, bf, tb + (0,))
                    if e == 7:
                        if q in (0, 1):
                            w.dma_pair()
                        elif q == 3:
                            w.dma_consts()
            g += 1
        par = (par + nsu) & 1
        for j in range(6):               # b_interval
            w.begin_interval(g, c["kVmDmaB0"] if j == 0 else c["kVmDmaB"])
            if j == 5:
                assert c["kPreItems"] <= 15      # the head of the next stage's pipeline sits in front of the interval's DMA statements
                pre_items(par, si + 1)
            w.dma_pair(); w.dma_pair(); w.dma_consts()
            g += 1
    return w, g


@pytest.mark.parametrize("K0,nl", [(128, 12), (128, 1), (160, 3), (256, 4), (352, 5)])
def test_vmcnt_constants_hold_for_every_wait(K0, nl):
    c = constants()
    w, units = run_block(K0, nl, c)
    assert units == 1 + sum(4 * (nsu_of(K0 + 32 * l) + 6) for l in range(nl))
    assert w.next_unit == units + 4              # four units of padding for the last intervals' copies (dense_block28_units)
    assert w.min_slack["dma"] <= 8, w.min_slack
    if nl > 1:
        assert w.min_slack["ring"] <= 2, w.min_slack


def test_a_wrong_constant_is_caught():
    c = constants()
    c["kVmRing"] += 8
    with pytest.raises(AssertionError, match="vmcnt"):
        run_block(128, 12, c)
    c = constants()
    c["kVmDmaB0"] += 6
    with pytest.raises(AssertionError, match="vmcnt"):
        run_block(128, 12, c)
