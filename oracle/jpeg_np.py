"""ORACLE (test infrastructure only) — baseline JPEG decode restated on the CPU: the arithmetic the reference gets from
libjpeg(-turbo) [EXT] through ``mx.image.imread(path, 1)`` (reference dataset.py:204,216 -> OpenCV imdecode -> libjpeg,
default decompression parameters: ``dct_method = JDCT_ISLOW``, ``do_fancy_upsampling = TRUE``).

libjpeg is a third-party dependency that is absent from /root/reference; its published algorithm is restated here:
  * entropy decoding            ITU-T T.81 Annex F.2.2 (Huffman, sequential DCT, 8-bit), byte stuffing B.1.1.5, RSTn E.2.4
  * dequantisation + 8x8 IDCT   libjpeg ``jidctint.c::jpeg_idct_islow`` (13-bit constants, two passes, PASS1_BITS = 2)
  * chroma upsampling           libjpeg ``jdsample.c::h2v1_fancy_upsample / h2v2_fancy_upsample`` (triangle filter)
  * YCbCr -> RGB                libjpeg ``jdcolor.c::ycc_rgb_convert`` (16-bit fixed point tables)
PINNED against Pillow 12 / libjpeg-turbo (the same library family the reference's decoder uses): the fixtures under
tests/golden/jpeg_*.npz hold JPEG byte streams with Pillow's decode of them (tests/golden/make_jpeg_fixtures.py), and
tests/test_cpu_jpeg.py requires this module to reproduce them bit for bit.
Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14,
                   21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53,
                   60, 61, 54, 47, 55, 62, 63], dtype=np.int64)      # zigzag position -> natural (row-major) index


class JpegError(ValueError):
    pass


def parse(data: bytes):
    """Marker segments of a baseline (SOF0), single-scan, Huffman-coded JPEG (T.81 Annex B).
    -> dict(width, height, comps=[(id, h, v, tq, td, ta)], qt={id: int32[64] natural order},
            dc={id: (counts[16], symbols)}, ac={...}, ri, scan=bytes (entropy-coded segment, still byte-stuffed))"""
    if data[:2] != b"\xff\xd8":
        raise JpegError("not a JPEG (no SOI)")
    pos = 2
    out = dict(qt={}, dc={}, ac={}, ri=0, comps=None)
    while True:
        while pos < len(data) and data[pos] != 0xFF:
            pos += 1
        while pos < len(data) and data[pos] == 0xFF:
            pos += 1
        if pos >= len(data):
            raise JpegError("truncated before SOS")
        m = data[pos]
        pos += 1
        if m in (0x01,) or 0xD0 <= m <= 0xD7:
            continue
        ln = (data[pos] << 8) | data[pos + 1]
        seg = data[pos + 2: pos + ln]
        if len(seg) != ln - 2:
            raise JpegError("truncated marker segment")
        if m == 0xDB:                                  # DQT
            p = 0
            while p < len(seg):
                pq, tq = seg[p] >> 4, seg[p] & 15
                p += 1
                if pq:
                    vals = [(seg[p + 2 * i] << 8) | seg[p + 2 * i + 1] for i in range(64)]
                    p += 128
                else:
                    vals = list(seg[p:p + 64])
                    p += 64
                q = np.zeros(64, np.int32)
                q[ZIGZAG] = vals
                out["qt"][tq] = q
        elif m == 0xC4:                                # DHT
            p = 0
            while p < len(seg):
                tc, th = seg[p] >> 4, seg[p] & 15
                counts = list(seg[p + 1:p + 17])
                n = sum(counts)
                syms = list(seg[p + 17:p + 17 + n])
                p += 17 + n
                out["ac" if tc else "dc"][th] = (counts, syms)
        elif m == 0xC0 or m == 0xC1:                   # SOF0 / SOF1 (8-bit extended sequential Huffman decodes the same way)
            if seg[0] != 8:
                raise JpegError("only 8-bit samples are supported")
            out["height"] = (seg[1] << 8) | seg[2]
            out["width"] = (seg[3] << 8) | seg[4]
            nc = seg[5]
            out["comps"] = [[seg[6 + 3 * i], seg[7 + 3 * i] >> 4, seg[7 + 3 * i] & 15, seg[8 + 3 * i], 0, 0] for i in range(nc)]
        elif 0xC2 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            raise JpegError("unsupported JPEG process SOF%d (only baseline sequential Huffman)" % (m - 0xC0))
        elif m == 0xDD:                                # DRI
            out["ri"] = (seg[0] << 8) | seg[1]
        elif m == 0xDA:                                # SOS
            if out["comps"] is None:
                raise JpegError("SOS before SOF")
            ns = seg[0]
            if ns != len(out["comps"]):
                raise JpegError("multi-scan JPEG is not supported")
            for i in range(ns):
                cid, tt = seg[1 + 2 * i], seg[2 + 2 * i]
                c = [c for c in out["comps"] if c[0] == cid][0]
                c[4], c[5] = tt >> 4, tt & 15
            pos += ln
            end = data.rfind(b"\xff\xd9")
            out["scan"] = data[pos: end if end >= pos else len(data)]
            return out
        elif m == 0xD9:
            raise JpegError("EOI before SOS")
        pos += ln


def _huff_lut(counts, syms):
    """T.81 Annex C code assignment -> dict (length, code) -> symbol"""
    lut, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            lut[(ln, code)] = syms[k]
            code += 1
            k += 1
        code <<= 1
    return lut


class _Bits:
    def __init__(self, scan: bytes):
        # remove byte stuffing and split at RSTn markers: list of byte strings, one per restart interval
        self.segs, cur, i, n = [], bytearray(), 0, len(scan)
        while i < n:
            b = scan[i]
            if b == 0xFF and i + 1 < n:
                nx = scan[i + 1]
                if nx == 0:
                    cur.append(0xFF)
                    i += 2
                    continue
                if 0xD0 <= nx <= 0xD7:
                    self.segs.append(bytes(cur))
                    cur = bytearray()
                    i += 2
                    continue
                if nx == 0xFF:
                    i += 1
                    continue
                break                                  # any other marker ends the entropy-coded data
            cur.append(b)
            i += 1
        self.segs.append(bytes(cur))
        self.start(0)

    def start(self, k):
        self.buf = self.segs[k] if k < len(self.segs) else b""
        self.pos = 0

    def bit(self):
        byte = self.buf[self.pos >> 3] if (self.pos >> 3) < len(self.buf) else 0    # libjpeg feeds zeros past the end
        b = (byte >> (7 - (self.pos & 7))) & 1
        self.pos += 1
        return b

    def bits(self, n):
        v = 0
        for _ in range(n):
            v = (v << 1) | self.bit()
        return v


def _extend(v, s):
    return v if v >= (1 << (s - 1)) else v - (1 << s) + 1


def decode_coefficients(hdr):
    """Entropy decode -> per component int32 [blocks_h, blocks_w, 64] (natural order, NOT dequantised), block grid
    padded to whole MCUs (T.81 A.2.4)."""
    comps = hdr["comps"]
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    W, H = hdr["width"], hdr["height"]
    mcux, mcuy = -(-W // (8 * hmax)), -(-H // (8 * vmax))
    if len(comps) == 1:                                # a single-component scan is not interleaved: 1 block per MCU (A.2.2)
        mcux, mcuy = -(-W // 8), -(-H // 8)
        grid = [(mcuy, mcux)]
        per_mcu = [(0, 0, 0)]
    else:
        grid = [(mcuy * c[2], mcux * c[1]) for c in comps]
        per_mcu = [(ci, by, bx) for ci, c in enumerate(comps) for by in range(c[2]) for bx in range(c[1])]
    coef = [np.zeros((g[0], g[1], 64), np.int32) for g in grid]
    dcl = {k: _huff_lut(*v) for k, v in hdr["dc"].items()}
    acl = {k: _huff_lut(*v) for k, v in hdr["ac"].items()}
    br = _Bits(hdr["scan"])
    pred = [0] * len(comps)
    ri, seg = hdr["ri"], 0

    def sym(lut):
        code = 0
        for ln in range(1, 17):
            code = (code << 1) | br.bit()
            s = lut.get((ln, code))
            if s is not None:
                return s
        raise JpegError("bad Huffman code")

    for m in range(mcux * mcuy):
        if ri and m and m % ri == 0:
            seg += 1
            br.start(seg)
            pred = [0] * len(comps)
        my, mx = divmod(m, mcux)
        for ci, by, bx in per_mcu:
            c = comps[ci]
            blk = coef[ci][my * (c[2] if len(comps) > 1 else 1) + by, mx * (c[1] if len(comps) > 1 else 1) + bx]
            s = sym(dcl[c[4]])
            diff = _extend(br.bits(s), s) if s else 0
            pred[ci] += diff
            blk[0] = pred[ci]
            k = 1
            while k < 64:
                rs = sym(acl[c[5]])
                r, s = rs >> 4, rs & 15
                if s:
                    k += r
                    if k > 63:
                        raise JpegError("coefficient index out of range")
                    blk[ZIGZAG[k]] = _extend(br.bits(s), s)
                    k += 1
                elif r == 15:
                    k += 16
                else:
                    break
    return coef


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def idct_islow(coef, q):
    """jidctint.c::jpeg_idct_islow over an array of blocks: coef int [..., 64] (natural order), q int32[64]
    -> uint8 [..., 8, 8] samples (level shift + range limit included)."""
    F = dict(f0_298=2446, f0_390=3196, f0_541=4433, f0_765=6270, f0_899=7373, f1_175=9633, f1_501=12299, f1_847=15137,
             f1_961=16069, f2_053=16819, f2_562=20995, f3_072=25172)
    CB, P1 = 13, 2
    x = (coef.astype(np.int64) * q.astype(np.int64)).reshape(coef.shape[:-1] + (8, 8))

    def one_d(v, axis, shift):
        v = np.moveaxis(v, axis, -1)
        i0, i1, i2, i3, i4, i5, i6, i7 = [v[..., i] for i in range(8)]
        z2, z3 = i2, i6
        z1 = (z2 + z3) * F["f0_541"]
        tmp2 = z1 + z3 * (-F["f1_847"])
        tmp3 = z1 + z2 * F["f0_765"]
        tmp0 = (i0 + i4) << CB
        tmp1 = (i0 - i4) << CB
        tmp10, tmp13, tmp11, tmp12 = tmp0 + tmp3, tmp0 - tmp3, tmp1 + tmp2, tmp1 - tmp2
        tmp0, tmp1, tmp2, tmp3 = i7, i5, i3, i1
        z1, z2, z3, z4 = tmp0 + tmp3, tmp1 + tmp2, tmp0 + tmp2, tmp1 + tmp3
        z5 = (z3 + z4) * F["f1_175"]
        tmp0, tmp1, tmp2, tmp3 = tmp0 * F["f0_298"], tmp1 * F["f2_053"], tmp2 * F["f3_072"], tmp3 * F["f1_501"]
        z1, z2, z3, z4 = z1 * (-F["f0_899"]), z2 * (-F["f2_562"]), z3 * (-F["f1_961"]), z4 * (-F["f0_390"])
        z3 = z3 + z5
        z4 = z4 + z5
        tmp0, tmp1, tmp2, tmp3 = tmp0 + z1 + z3, tmp1 + z2 + z4, tmp2 + z2 + z3, tmp3 + z1 + z4
        o = [tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3]
        return np.moveaxis(np.stack([_descale(t, shift) for t in o], axis=-1), -1, axis)

    ws = one_d(x, -2, CB - P1)            # pass 1: columns (index varies along rows)
    y = one_d(ws, -1, CB + P1 + 3)        # pass 2: rows
    v = y & 1023                          # the range-limit table of jdmaster.c::prepare_range_limit_table, index masked
    out = np.where(v < 128, v + 128, np.where(v < 512, 255, np.where(v < 896, 0, v - 896)))
    return out.astype(np.uint8)


def component_planes(hdr, coef):
    """IDCT of every block -> per component uint8 plane of the padded block grid."""
    planes = []
    for c, cf in zip(hdr["comps"], coef):
        s = idct_islow(cf, hdr["qt"][c[3]])                      # (bh, bw, 8, 8)
        planes.append(s.transpose(0, 2, 1, 3).reshape(s.shape[0] * 8, s.shape[1] * 8))
    return planes


def _h2v1_fancy(p):
    """jdsample.c::h2v1_fancy_upsample over the rows of p (int32 [h, w]) -> [h, 2w]"""
    h, w = p.shape
    out = np.empty((h, 2 * w), np.int32)
    left = np.concatenate([p[:, :1], p[:, :-1]], axis=1)
    right = np.concatenate([p[:, 1:], p[:, -1:]], axis=1)
    out[:, 0::2] = (3 * p + left + 1) >> 2
    out[:, 1::2] = (3 * p + right + 2) >> 2
    out[:, 0] = p[:, 0]
    out[:, -1] = p[:, -1]
    return out


def _h2v2_fancy(p):
    """jdsample.c::h2v2_fancy_upsample: p int32 [h, w] (real rows / columns only) -> [2h, 2w]; the rows above the
    first and below the last are the edge rows themselves (jdmainct.c context rows)."""
    h, w = p.shape
    up = np.concatenate([p[:1], p[:-1]], axis=0)
    dn = np.concatenate([p[1:], p[-1:]], axis=0)
    out = np.empty((2 * h, 2 * w), np.int32)
    for v, other in ((0, up), (1, dn)):
        cs = 3 * p + other                                        # thiscolsum
        last = np.concatenate([cs[:, :1], cs[:, :-1]], axis=1)
        nxt = np.concatenate([cs[:, 1:], cs[:, -1:]], axis=1)
        o = np.empty((h, 2 * w), np.int32)
        o[:, 0::2] = (3 * cs + last + 8) >> 4
        o[:, 1::2] = (3 * cs + nxt + 7) >> 4
        o[:, 0] = (4 * cs[:, 0] + 8) >> 4
        o[:, -1] = (4 * cs[:, -1] + 7) >> 4
        out[v::2] = o
    return out


def ycc_to_rgb(y, cb, cr):
    """jdcolor.c::ycc_rgb_convert (build_ycc_rgb_table): int arrays 0..255 -> uint8 [h, w, 3]"""
    x_cb, x_cr = cb.astype(np.int64) - 128, cr.astype(np.int64) - 128
    half = 1 << 15
    r = y + ((91881 * x_cr + half) >> 16)
    b = y + ((116130 * x_cb + half) >> 16)
    g = y + ((-22554 * x_cb + half - 46802 * x_cr) >> 16)
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)


def decode(data: bytes) -> np.ndarray:
    """-> uint8 [H, W, 3] RGB (grey images replicated, as ``mx.image.imread(path, 1)`` returns them)"""
    hdr = parse(data)
    coef = decode_coefficients(hdr)
    planes = component_planes(hdr, coef)
    W, H = hdr["width"], hdr["height"]
    comps = hdr["comps"]
    if len(comps) == 1:
        y = planes[0][:H, :W]
        return np.repeat(y[:, :, None], 3, axis=2)
    if len(comps) != 3:
        raise JpegError("only 1- or 3-component images are supported")
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    if (comps[0][1], comps[0][2]) != (hmax, vmax):
        raise JpegError("luma must carry the maximum sampling factors")
    full = [planes[0][:H, :W].astype(np.int32)]
    for c, p in zip(comps[1:], planes[1:]):
        hs, vs = hmax // c[1], vmax // c[2]
        cw, ch = -(-W * c[1] // hmax), -(-H * c[2] // vmax)         # downsampled_width / height
        p = p[:ch, :cw].astype(np.int32)
        if (hs, vs) == (1, 1):
            u = p
        elif cw <= 2 and hs == 2 and vs in (1, 2):      # jdsample.c::jinit_upsampler: fancy only if downsampled_width > 2
            u = np.repeat(np.repeat(p, hs, axis=1), vs, axis=0)
        elif (hs, vs) == (2, 1):
            u = _h2v1_fancy(p)
        elif (hs, vs) == (2, 2):
            u = _h2v2_fancy(p)
        else:
            raise JpegError("unsupported chroma sampling %dx%d" % (hs, vs))
        full.append(u[:H, :W])
    return ycc_to_rgb(full[0], full[1], full[2])
