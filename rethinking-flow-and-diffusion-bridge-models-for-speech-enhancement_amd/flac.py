"""FLAC reader for the folder drivers (the reference reads wav AND flac through soundfile, infer_folder.py:58-65,94).

The image has no audio library, so this is a decoder of the FLAC format written from its specification (RFC 9639):
STREAMINFO, frame headers (fixed / variable block size, every block-size / sample-rate / sample-size code, CRC-8),
CONSTANT / VERBATIM / FIXED / LPC subframes with wasted bits, partitioned Rice residuals with 4- and 5-bit parameters
and escaped (raw) partitions, left-side / right-side / mid-side decorrelation, frame CRC-16, and the MD5 signature of the
decoded samples.  Pure Python over a bit reader: ~0.3 s for a 4 s mono clip - an input path, not a hot path.
No FLAC file or encoder exists in this image either: tests/test_host_api.py checks the decoder against an encoder
written from the same specification (tests/flac_encode.py), bit-exactly, including both CRCs and the MD5.
"""
import hashlib

import numpy as np


class FlacError(ValueError):
    pass


def _crc_table(poly, bits):
    top = 1 << (bits - 1)
    mask = (1 << bits) - 1
    tab = []
    for i in range(256):
        c = i << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
        tab.append(c)
    return tab


_CRC8 = _crc_table(0x07, 8)
_CRC16 = _crc_table(0x8005, 16)


def crc8(data):
    c = 0
    for b in data:
        c = _CRC8[c ^ b]
    return c


def crc16(data):
    c = 0
    for b in data:
        c = ((c << 8) & 0xFFFF) ^ _CRC16[(c >> 8) ^ b]
    return c


class _Bits:
    """MSB-first bit reader over bytes; `ones` = positions of the set bits (for unary codes)."""

    def __init__(self, data):
        self.data = data
        self.pos = 0
        self.n = 8 * len(data)
        self.ones = np.flatnonzero(np.unpackbits(np.frombuffer(data, dtype=np.uint8))).tolist()
        self.ones.append(self.n + (1 << 40))          # sentinel
        self.oi = 0

    def read(self, k):
        if k == 0:
            return 0
        p = self.pos
        if p + k > self.n:
            raise FlacError("unexpected end of stream")
        b0 = p >> 3
        nb = ((p & 7) + k + 7) >> 3
        v = int.from_bytes(self.data[b0:b0 + nb], "big")
        self.pos = p + k
        return (v >> (8 * nb - (p & 7) - k)) & ((1 << k) - 1)

    def read_signed(self, k):
        v = self.read(k)
        return v - (1 << k) if k and v >> (k - 1) else v

    def read_unary(self):
        ones, oi, p = self.ones, self.oi, self.pos
        while ones[oi] < p:
            oi += 1
        q = ones[oi] - p
        if ones[oi] >= self.n:
            raise FlacError("unexpected end of stream")
        self.pos = ones[oi] + 1
        self.oi = oi
        return q

    def align(self):
        self.pos = (self.pos + 7) & ~7


_BLOCK = {1: 192, 2: 576, 3: 1152, 4: 2304, 5: 4608, 8: 256, 9: 512, 10: 1024, 11: 2048, 12: 4096, 13: 8192, 14: 16384, 15: 32768}
_RATE = {1: 88200, 2: 176400, 3: 192000, 4: 8000, 5: 16000, 6: 22050, 7: 24000, 8: 32000, 9: 44100, 10: 48000, 11: 96000}
_BPS = {1: 8, 2: 12, 4: 16, 5: 20, 6: 24, 7: 32}
_FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}


def _residual(br, order, blocksize, out):
    method = br.read(2)
    if method > 1:
        raise FlacError("reserved residual coding method")
    pbits = 4 if method == 0 else 5
    esc = (1 << pbits) - 1
    porder = br.read(4)
    nparts = 1 << porder
    if blocksize % nparts or (blocksize >> porder) < order and porder:
        raise FlacError("invalid partition order")
    for part in range(nparts):
        n = (blocksize >> porder) - (order if part == 0 else 0)
        k = br.read(pbits)
        if k == esc:
            nb = br.read(5)
            for _ in range(n):
                out.append(br.read_signed(nb))
        else:
            ru, rd = br.read_unary, br.read
            for _ in range(n):
                u = (ru() << k) | rd(k)
                out.append((u >> 1) ^ -(u & 1))


def _subframe(br, bps, blocksize):
    if br.read(1):
        raise FlacError("subframe padding bit set")
    typ = br.read(6)
    wasted = 0
    if br.read(1):
        wasted = br.read_unary() + 1
        bps -= wasted
    if typ == 0:                                   # CONSTANT
        s = [br.read_signed(bps)] * blocksize
    elif typ == 1:                                 # VERBATIM
        s = [br.read_signed(bps) for _ in range(blocksize)]
    elif 8 <= typ <= 12:                           # FIXED, order typ - 8
        order = typ - 8
        s = [br.read_signed(bps) for _ in range(order)]
        res = []
        _residual(br, order, blocksize, res)
        c = _FIXED[order]
        for r in res:
            p = 0
            for j, cj in enumerate(c):
                p += cj * s[-1 - j]
            s.append(r + p)
    elif typ >= 32:                                # LPC, order typ - 31
        order = typ - 31
        s = [br.read_signed(bps) for _ in range(order)]
        prec = br.read(4) + 1
        if prec == 16:
            raise FlacError("invalid LPC precision")
        shift = br.read_signed(5)
        if shift < 0:
            raise FlacError("negative LPC shift")
        coef = [br.read_signed(prec) for _ in range(order)]
        res = []
        _residual(br, order, blocksize, res)
        for r in res:
            p = 0
            for j in range(order):
                p += coef[j] * s[-1 - j]
            s.append(r + (p >> shift))
    else:
        raise FlacError(f"reserved subframe type {typ}")
    if wasted:
        s = [v << wasted for v in s]
    return s


def decode(data, check_md5=True):
    """FLAC bytes -> (int samples [C, L] as int64 ndarray, sample rate, bits per sample)."""
    if data[:4] != b"fLaC":
        raise FlacError("not a FLAC stream")
    pos = 4
    info = None
    while True:
        hdr = data[pos]
        length = int.from_bytes(data[pos + 1:pos + 4], "big")
        body = data[pos + 4:pos + 4 + length]
        if hdr & 0x7F == 0:
            v = int.from_bytes(body[10:18], "big")
            info = dict(rate=v >> 44, channels=((v >> 41) & 7) + 1, bps=((v >> 36) & 31) + 1, total=v & ((1 << 36) - 1), md5=body[18:34])
        pos += 4 + length
        if hdr & 0x80:
            break
    if info is None:
        raise FlacError("no STREAMINFO block")
    C, bps0 = info["channels"], info["bps"]
    chans = [[] for _ in range(C)]
    br = _Bits(data[pos:])
    while br.pos + 16 <= br.n:
        start = br.pos >> 3
        if br.read(14) != 0x3FFE:
            raise FlacError("lost frame synchronisation")
        if br.read(1):
            raise FlacError("reserved bit set in frame header")
        br.read(1)                                   # blocking strategy (the coded number is not needed to decode)
        bs_code, sr_code = br.read(4), br.read(4)
        ch_code, ss_code = br.read(4), br.read(3)
        if br.read(1):
            raise FlacError("reserved bit set in frame header")
        first = br.read(8)                           # UTF-8-like coded frame / sample number
        extra = 0
        while first & (0x80 >> extra):
            extra += 1
        for _ in range(max(0, extra - 1)):
            if br.read(8) >> 6 != 2:
                raise FlacError("bad coded number in frame header")
        if bs_code == 0:
            raise FlacError("reserved block size code")
        blocksize = br.read(8) + 1 if bs_code == 6 else br.read(16) + 1 if bs_code == 7 else _BLOCK[bs_code]
        if sr_code == 12:
            br.read(8)
        elif sr_code in (13, 14):
            br.read(16)
        elif sr_code == 15:
            raise FlacError("invalid sample rate code")
        hdr_end = br.pos >> 3
        if br.read(8) != crc8(br.data[start:hdr_end]):
            raise FlacError("frame header CRC-8 mismatch")
        bps = bps0 if ss_code == 0 else _BPS.get(ss_code)
        if bps is None:
            raise FlacError("reserved sample size code")
        if ch_code < 8:
            if ch_code + 1 != C:
                raise FlacError("channel count changes inside the stream")
            sub = [_subframe(br, bps, blocksize) for _ in range(C)]
        elif ch_code <= 10:
            if C != 2:
                raise FlacError("stereo decorrelation in a non-stereo stream")
            a = _subframe(br, bps + (1 if ch_code == 9 else 0), blocksize)
            b = _subframe(br, bps + (0 if ch_code == 9 else 1), blocksize)
            if ch_code == 8:                         # left, side
                sub = [a, [l - s for l, s in zip(a, b)]]
            elif ch_code == 9:                       # side, right
                sub = [[s + r for s, r in zip(a, b)], b]
            else:                                    # mid, side
                left, right = [], []
                for m, s in zip(a, b):
                    m = (m << 1) | (s & 1)
                    left.append((m + s) >> 1)
                    right.append((m - s) >> 1)
                sub = [left, right]
        else:
            raise FlacError("reserved channel assignment")
        br.align()
        body_end = br.pos >> 3
        if br.read(16) != crc16(br.data[start:body_end]):
            raise FlacError("frame CRC-16 mismatch")
        for c in range(C):
            chans[c].extend(sub[c])
    x = np.asarray(chans, dtype=np.int64)
    if info["total"] and x.shape[1] != info["total"]:
        raise FlacError(f"decoded {x.shape[1]} samples, STREAMINFO announces {info['total']}")
    if check_md5 and any(info["md5"]):
        nbytes = (bps0 + 7) // 8
        inter = x.T.reshape(-1)
        raw = b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in inter) if nbytes == 3 else \
            inter.astype({1: "<i1", 2: "<i2", 4: "<i4"}[nbytes]).tobytes()
        if hashlib.md5(raw).digest() != info["md5"]:
            raise FlacError("MD5 signature of the decoded samples does not match STREAMINFO")
    return x, info["rate"], bps0


def read_flac(path):
    """-> (float32 [C, L] in [-1, 1), sample rate) like fdbm_amd.infer.read_wav."""
    with open(path, "rb") as f:
        x, rate, bps = decode(f.read())
    return np.ascontiguousarray((x / float(1 << (bps - 1))).astype(np.float32)), int(rate)
