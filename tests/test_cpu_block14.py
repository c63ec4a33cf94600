"""-m "not gpu": the hand-counted s_waitcnt vmcnt(N) constants of dense_block14.hip, proven by replaying the kernel's issue order.

The 14x14 block kernel issues every steady-state vector-memory load from inline asm (the activation ring's refills and the
LDS-DMA pieces of the weight stream) and waits for them with literal counts: vmcnt(N) returns when at most N loads are
outstanding, and loads complete in order, so a wait is correct iff at least N loads were issued BEHIND the one it needs.  This
test re-states the issue order of one wave for whole blocks (every interval kind, both ring parities, odd and even super-step
counts), takes the constants from the source, and checks every wait - plus that each ring register holds the super-step and
k-step its consumer expects.  It is a model of the schedule, kept next to it: a change of the slot layout in the kernel has to
be mirrored here (the comments name the lambdas)."""
import os
import re

import pytest

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tennis_amd", "csrc", "dense_block14.hip")


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
        self.ring = {}                  # (rs, kq, f) -> (issue index, (layer, su, kq))
        self.dma = {}                   # unit -> issue index of this wave's LAST piece
        self.next_unit = 0              # unit the next DMA statements copy
        self.min_slack = {}

    def load(self):
        self.n += 1
        return self.n - 1

    def need(self, idx, vm, what):
        younger = self.n - 1 - idx
        ok = idx < self.done_upto or younger >= vm
        assert ok, f"{what}: vmcnt({vm}) with only {younger} loads behind the one it waits for"
        if idx >= self.done_upto:
            k = what.split(":")[0]
            self.min_slack[k] = min(self.min_slack.get(k, 1 << 30), younger - vm)

    # -- the kernel's statements
    def dma_pair(self):
        self.load(); self.load()

    def dma_consts(self):
        self.dma[self.next_unit] = self.load()
        self.next_unit += 1             # (advance_dma at the end of the interval; nothing copies in between)

    def ring_load(self, rs, kq, f, holds):
        self.ring[(rs, kq, f)] = (self.load(), holds)

    def ring_wait(self, rs, kq, expect):
        for f in (0, 1):
            idx, holds = self.ring[(rs, kq, f)]
            assert holds == expect, f"ring[{rs}][{kq}][{f}] holds {holds}, its consumer expects {expect}"
            self.need(idx, self.c["kVmRing"], "ring: %s" % (expect,))

    def begin_interval(self, g, vm):
        if g + 1 in self.dma:           # this wave's pieces of unit g + 1
            self.need(self.dma[g + 1], vm, "dma: unit %d" % (g + 1))
        else:
            raise AssertionError(f"unit {g + 1} was never copied")


def nsu_of(K):
    return (K - 32 + 63) // 64


def run_block(K0, nl, c):
    w = Wave(c)
    # prologue: units 0 .. 3, the first layer's super-steps 0 / 1, vmcnt(0), the head of the pipeline (pre_item)
    for _ in range(4):
        w.dma_pair(); w.dma_pair(); w.dma_consts()
    for rs in (0, 1):
        for kq in range(4):
            for f in (0, 1):
                w.ring_load(rs, kq, f, (0, rs, kq))
    w.done_upto = w.n

    def wrap(l, su):                     # super-step su of layer l, or of the layers behind it
        while su >= nsu_of(K0 + 32 * l):
            su -= nsu_of(K0 + 32 * l)
            l += 1
        return l, su

    def pre_items(pn, l):                # b_interval J = 5 / prologue: BN of (layer l, super-step 0, k-step 0); refill <- super-step 2, k-step 0
        w.ring_wait(pn, 0, (l, 0, 0))
        for f in (0, 1):
            w.ring_load(pn, 0, f, (l, 2, 0))
    pre_items(0, 0)
    w.done_upto = w.n                    # (the prologue waits for its two refills: no DMA statements follow them there)
    g, par = 0, 0
    for l in range(nl):
        nsu = nsu_of(K0 + 32 * l)
        assert nsu >= 4
        for u in range(nsu):             # su_interval
            kind = 0 if u == 0 else (2 if u == nsu - 1 else 1)
            rs = (u + par) & 1
            w.begin_interval(g, c["kVmDmaSU0"] if kind == 0 else c["kVmDmaSU"])
            la, ua = wrap(l, u + 2)
            lb, ub = wrap(l, u + 3)
            for q in range(4):
                for e in range(8):
                    j, bf = e >> 1, e & 1
                    if q < 3:
                        if e == 0:
                            w.ring_wait(rs, q + 1, (l, u, q + 1))
                        if j == 3:
                            w.ring_load(rs, q + 1, bf, (la, ua, q + 1))
                    elif kind != 2:
                        if e == 0:
                            w.ring_wait(rs ^ 1, 0, (l, u + 1, 0))
                        if j == 3:
                            w.ring_load(rs ^ 1, 0, bf, (lb, ub, 0))
                    if e == 7:
                        if q in (0, 1):
                            w.dma_pair()
                        elif q == 3:
                            w.dma_consts()
            g += 1
        # tail_interval
        w.begin_interval(g, c["kVmDmaTail"])
        w.dma_pair(); w.dma_pair(); w.dma_consts()
        g += 1
        par = (par + nsu) & 1
        for j in range(6):               # b_interval
            w.begin_interval(g, c["kVmDmaB0"] if j == 0 else c["kVmDmaB"])
            if j == 5:
                assert c["kPreItems"] <= 15      # the head of the next layer's pipeline sits in front of the interval's DMA statements
                pre_items(par, l + 1)
            w.dma_pair(); w.dma_pair(); w.dma_consts()
            g += 1
    return w, g


@pytest.mark.parametrize("K0,nl", [(256, 24), (512, 16), (256, 1), (288, 5), (320, 8)])
def test_vmcnt_constants_hold_for_every_wait(K0, nl):
    c = constants()
    w, units = run_block(K0, nl, c)
    assert units == sum(nsu_of(K0 + 32 * l) + 7 for l in range(nl))
    assert w.next_unit == units + 4              # the stream carries four units of padding for the last intervals' copies
    # the counts are not only safe but tight enough to keep the prefetch distance: a ring load may stay in flight for (almost) its two intervals
    assert w.min_slack["ring"] <= 2, w.min_slack
    assert w.min_slack["dma"] <= 8, w.min_slack


def test_a_wrong_constant_is_caught():
    c = constants()
    c["kVmRing"] += 8
    with pytest.raises(AssertionError, match="vmcnt"):
        run_block(256, 24, c)
    c = constants()
    c["kVmDmaB"] += 6
    with pytest.raises(AssertionError, match="vmcnt"):
        run_block(256, 24, c)
