"""Baseline JPEG encoder in numpy -- the CPU restatement the device encoder (csrc/rva_jpeg.hip, K7) is held to.

TEST INFRASTRUCTURE ONLY (tests/, never the product package).  It restates, step by step, what the reference's
``cv2.imencode('.jpg', frame, [IMWRITE_JPEG_QUALITY, q, ...])`` (sinks/kafka_sink.py:260-284) does inside libjpeg for a BGR
frame -- the published algorithms of the Independent JPEG Group's library (the reference pins no version; OpenCV bundles
libjpeg-turbo, API level 6.2):

  * colour conversion RGB -> YCbCr, 16-bit fixed point (jccolor.c): Y = (19595 R + 38470 G + 7471 B + 32768) >> 16, ...
  * edge expansion to whole 16 x 16 MCUs: last column replicated before the downsampler, last DOWNSAMPLED row after it (jcprepct.c);
  * luma blocks wholly outside the image are dummy blocks: AC zero, DC copied (jccoefct.c);
  * 2 x 2 chroma downsampling with the alternating 1, 2 bias (jcsample.c h2v2_downsample);
  * forward DCT "islow" (jfdctint.c: CONST_BITS 13, PASS1_BITS 2, output scaled by 8) on samples - 128;
  * quantisation with the Annex-K tables scaled by jpeg_quality_scaling (jcparam.c), symmetric rounding (jcdctmgr.c);
  * baseline sequential Huffman coding with the Annex-K tables, byte stuffing, restart markers every MCU row.

PINNED by Pillow (libjpeg-turbo, present in the image): tests/test_oracle_golden.py decodes this encoder's stream with Pillow and
requires the pixels Pillow decodes from ITS OWN encoding of the same image at the same quality and 4:2:0 sampling -- equal
pixels mean equal quantised coefficients and tables, i.e. everything except the entropy coder (which any decoder checks by
decoding at all).  The reference asks libjpeg for a progressive, Huffman-optimised file; this is the baseline form of the same
coefficients: same decoded picture, a few per cent larger.
"""
from __future__ import annotations

import numpy as np

ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                   28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
                   47, 55, 62, 63])
STD_LUMA_Q = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51,
                       87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101,
                       72, 92, 95, 98, 112, 100, 103, 99])
STD_CHROMA_Q = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99,
                         99, 99, 99, 99] + [99] * 32)
# Annex K.3 Huffman tables: number of codes per length 1..16, then the symbols
DC_LUMA = ([0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0], list(range(12)))
DC_CHROMA = ([0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], list(range(12)))
AC_LUMA = ([0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d],
           [0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
            0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
            0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
            0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
            0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
            0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
            0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
            0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa])
AC_CHROMA = ([0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77],
             [0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81,
              0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34,
              0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
              0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
              0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92,
              0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
              0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
              0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
              0xf9, 0xfa])


def quant_tables(quality: int):
    """jcparam.c: jpeg_quality_scaling + jpeg_add_quant_table (force_baseline): natural order."""
    q = max(1, min(100, int(quality)))
    scale = 5000 // q if q < 50 else 200 - 2 * q
    f = lambda base: np.clip((base.astype(np.int64) * scale + 50) // 100, 1, 255).astype(np.int32)      # noqa: E731
    return f(STD_LUMA_Q), f(STD_CHROMA_Q)


def huff_codes(bits, vals):
    """code and length per symbol (jchuff.c jpeg_make_c_derived_tbl)."""
    code, k, ehufco, ehufsi = 0, 0, np.zeros(256, np.int64), np.zeros(256, np.int64)
    for length in range(1, 17):
        for _ in range(bits[length - 1]):
            ehufco[vals[k]] = code
            ehufsi[vals[k]] = length
            code += 1
            k += 1
        code <<= 1
    return ehufco, ehufsi


def ycbcr_planes(bgr: np.ndarray):
    """jccolor.c rgb_ycc_convert + jcprepct.c edge expansion + jcsample.c h2v2_downsample: Y [Hp, Wp], Cb / Cr [Hp/2, Wp/2]."""
    h, w = bgr.shape[:2]
    hp, wp = (h + 15) // 16 * 16, (w + 15) // 16 * 16
    b, g, r = (bgr[..., i].astype(np.int64) for i in range(3))
    y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16
    # jcprepct.c / jcsample.c: columns are replicated on the INPUT of the downsampler (expand_right_edge), rows only up to an even
    # count; the rest of the last iMCU row is filled by replicating the last DOWNSAMPLED row of each component (expand_bottom_edge
    # on the output of the downsampler) -- for chroma that is not the same as downsampling replicated pixel rows
    he = h + (h & 1)
    pad = lambda p: np.pad(p, ((0, he - h), (0, wp - w)), mode="edge")      # noqa: E731
    y, cb, cr = pad(y), pad(cb), pad(cr)
    bias = np.tile(np.array([1, 2]), wp // 4 + 1)[: wp // 2][None, :]

    def down(p):
        return (p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + bias) >> 2
    fill = lambda p, rows: np.pad(p, ((0, rows - p.shape[0]), (0, 0)), mode="edge")      # noqa: E731
    return fill(y, hp), fill(down(cb), hp // 2), fill(down(cr), hp // 2)


def fdct_islow(block: np.ndarray) -> np.ndarray:
    """jfdctint.c jpeg_fdct_islow on an int64 [..., 8, 8] array of samples - 128: output scaled by 8."""
    C, P = 13, 2
    F = dict(a=2446, b=3196, c=4433, d=6270, e=7373, f=9633, g=12299, h=15137, i=16069, j=16819, k=20995, l=25172)

    def desc(x, n):
        return (x + (1 << (n - 1))) >> n

    def one_pass(d, first):
        d0, d1, d2, d3, d4, d5, d6, d7 = (d[..., i] for i in range(8))
        t0, t7, t1, t6, t2, t5, t3, t4 = d0 + d7, d0 - d7, d1 + d6, d1 - d6, d2 + d5, d2 - d5, d3 + d4, d3 - d4
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        o = [None] * 8
        if first:
            o[0], o[4] = (t10 + t11) << P, (t10 - t11) << P
        else:
            o[0], o[4] = desc(t10 + t11, P), desc(t10 - t11, P)
        sh = C - P if first else C + P
        z1 = (t12 + t13) * F["c"]
        o[2] = desc(z1 + t13 * F["d"], sh)
        o[6] = desc(z1 - t12 * F["h"], sh)
        z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
        z5 = (z3 + z4) * F["f"]
        t4, t5, t6, t7 = t4 * F["a"], t5 * F["j"], t6 * F["l"], t7 * F["g"]
        z1, z2, z3, z4 = -z1 * F["e"], -z2 * F["k"], -z3 * F["i"] + z5, -z4 * F["b"] + z5
        o[7], o[5], o[3], o[1] = desc(t4 + z1 + z3, sh), desc(t5 + z2 + z4, sh), desc(t6 + z2 + z3, sh), desc(t7 + z1 + z4, sh)
        return np.stack(o, -1)
    rows = one_pass(block, True)                                   # pass 1: along each row
    cols = one_pass(np.swapaxes(rows, -1, -2), False)              # pass 2: along each column
    return np.swapaxes(cols, -1, -2)


def quantise(coef: np.ndarray, qtbl: np.ndarray) -> np.ndarray:
    """jcdctmgr.c: divisor = table << 3 (the DCT's scale), symmetric round-half-up of |x| / divisor."""
    div = (qtbl.reshape(8, 8).astype(np.int64) << 3)
    a = np.abs(coef)
    return (np.sign(coef) * ((a + (div >> 1)) // div)).astype(np.int64)


def blocks_of(plane: np.ndarray) -> np.ndarray:
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).swapaxes(1, 2)      # [by, bx, 8, 8]


def coefficients(bgr: np.ndarray, quality: int):
    """Quantised coefficients in zigzag order per component: Y [Hb, Wb, 64], Cb, Cr [Hb/2, Wb/2, 64]."""
    ql, qc = quant_tables(quality)
    y, cb, cr = ycbcr_planes(bgr)
    out = []
    for plane, q in ((y, ql), (cb, qc), (cr, qc)):
        c = quantise(fdct_islow(blocks_of(plane) - 128), q)
        out.append(c.reshape(*c.shape[:2], 64)[..., ZIGZAG])
    # jccoefct.c compress_data: luma blocks that lie wholly outside the image (the MCU is 2 x 2 blocks, the image may end after
    # the first) are DUMMY blocks -- all AC zero, DC = the DC of the block before them in the MCU buffer: the block to the left
    # for a dummy at the right edge, the LAST block of the row above (for both blocks of the row) for a dummy row at the bottom.
    h, w = bgr.shape[:2]
    nby, nbx = (h + 7) // 8, (w + 7) // 8
    cy = out[0]
    for my in range(cy.shape[0] // 2):
        for mx in range(cy.shape[1] // 2):
            if 2 * mx + 1 >= nbx:
                for r in (0, 1):
                    cy[2 * my + r, 2 * mx + 1, :] = 0
                    cy[2 * my + r, 2 * mx + 1, 0] = cy[2 * my + r, 2 * mx, 0]
            if 2 * my + 1 >= nby:
                cy[2 * my + 1, 2 * mx:2 * mx + 2, :] = 0
                cy[2 * my + 1, 2 * mx:2 * mx + 2, 0] = cy[2 * my, 2 * mx + 1, 0]
    return out, (ql, qc)


class _Bits:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, code: int, size: int):
        self.acc = (self.acc << size) | (code & ((1 << size) - 1))
        self.n += size
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8
        self.acc &= (1 << self.n) - 1

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)            # pad with ones


def _encode_block(bw: _Bits, zz, pred: int, dc_tbl, ac_tbl) -> int:
    diff = int(zz[0]) - pred
    t = abs(diff)
    nb = t.bit_length()
    bw.put(int(dc_tbl[0][nb]), int(dc_tbl[1][nb]))
    if nb:
        bw.put(diff if diff >= 0 else diff - 1, nb)
    run = 0
    for k in range(1, 64):
        v = int(zz[k])
        if v == 0:
            run += 1
            continue
        while run > 15:
            bw.put(int(ac_tbl[0][0xF0]), int(ac_tbl[1][0xF0]))
            run -= 16
        nb = abs(v).bit_length()
        sym = (run << 4) | nb
        bw.put(int(ac_tbl[0][sym]), int(ac_tbl[1][sym]))
        bw.put(v if v >= 0 else v - 1, nb)
        run = 0
    if run:
        bw.put(int(ac_tbl[0][0]), int(ac_tbl[1][0]))
    return int(zz[0])


def header(w: int, h: int, ql, qc, restart_interval: int) -> bytes:
    def seg(marker, payload):
        return bytes([0xFF, marker]) + (len(payload) + 2).to_bytes(2, "big") + payload
    out = bytearray(b"\xff\xd8")
    out += seg(0xE0, b"JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    out += seg(0xDB, bytes([0]) + bytes(int(v) for v in ql[ZIGZAG]))
    out += seg(0xDB, bytes([1]) + bytes(int(v) for v in qc[ZIGZAG]))
    out += seg(0xC0, bytes([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([3, 1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1]))
    for cls_id, (bits, vals) in ((0x00, DC_LUMA), (0x10, AC_LUMA), (0x01, DC_CHROMA), (0x11, AC_CHROMA)):
        out += seg(0xC4, bytes([cls_id]) + bytes(bits) + bytes(vals))
    if restart_interval:
        out += seg(0xDD, restart_interval.to_bytes(2, "big"))
    out += seg(0xDA, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0]))
    return bytes(out)


def encode(bgr: np.ndarray, quality: int) -> bytes:
    """Baseline JFIF stream, 4:2:0, one restart interval per MCU row (what the device encoder emits, byte for byte)."""
    h, w = bgr.shape[:2]
    (cy, ccb, ccr), (ql, qc) = coefficients(bgr, quality)
    mh, mw = cy.shape[0] // 2, cy.shape[1] // 2
    dcl, acl, dcc, acc = huff_codes(*DC_LUMA), huff_codes(*AC_LUMA), huff_codes(*DC_CHROMA), huff_codes(*AC_CHROMA)
    out = bytearray(header(w, h, ql, qc, mw))
    for my in range(mh):
        bw = _Bits()
        py = pcb = pcr = 0                                         # predictors reset at every restart
        for mx in range(mw):
            for by, bx in ((0, 0), (0, 1), (1, 0), (1, 1)):
                py = _encode_block(bw, cy[2 * my + by, 2 * mx + bx], py, dcl, acl)
            pcb = _encode_block(bw, ccb[my, mx], pcb, dcc, acc)
            pcr = _encode_block(bw, ccr[my, mx], pcr, dcc, acc)
        bw.flush()
        out += bw.out
        if my + 1 < mh:
            out += bytes([0xFF, 0xD0 + (my & 7)])
    out += b"\xff\xd9"
    return bytes(out)
