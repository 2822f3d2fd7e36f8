"""A small FLAC ENCODER, test infrastructure only: writes streams that exercise every construct the decoder
(fdbm_amd/flac.py) implements - written from the format specification (RFC 9639), independently of the decoder's code:
the checks that tie the two together are the format's own (CRC-8 per frame header, CRC-16 per frame, MD5 of the samples).

encode(x [C, L] ints, rate, bps, blocks=[(blocksize, mode, stereo)...]) -> bytes
  mode   'constant' | 'verbatim' | ('fixed', order, partition_order, escape) | ('lpc', order, precision, shift, partition_order)
  stereo 'indep' | 'ls' | 'rs' | 'ms'
"""
import hashlib

import numpy as np

FIXED = {0: (), 1: (1,), 2: (2, -1), 3: (3, -3, 1), 4: (4, -6, 4, -1)}
BLOCK_CODE = {192: 1, 576: 2, 1152: 3, 2304: 4, 4608: 5, 256: 8, 512: 9, 1024: 10, 2048: 11, 4096: 12, 8192: 13, 16384: 14, 32768: 15}
RATE_CODE = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
BPS_CODE = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6}


class BitWriter:
    def __init__(self):
        self.acc = 0
        self.nbits = 0

    def write(self, v, k):
        if k:
            self.acc = (self.acc << k) | (v & ((1 << k) - 1))
            self.nbits += k

    def signed(self, v, k):
        assert -(1 << (k - 1)) <= v < (1 << (k - 1)), (v, k)
        self.write(v & ((1 << k) - 1), k)

    def unary(self, q):
        self.write(1, q + 1)                       # q zeros, then a one

    def align(self):
        self.write(0, (-self.nbits) % 8)

    def bytes(self):
        assert self.nbits % 8 == 0
        return self.acc.to_bytes(self.nbits // 8, "big") if self.nbits else b""


def _crc(data, poly, bits):
    c = 0
    top, mask = 1 << (bits - 1), (1 << bits) - 1
    for b in data:
        c ^= b << (bits - 8)
        for _ in range(8):
            c = ((c << 1) ^ poly) & mask if c & top else (c << 1) & mask
    return c


def _utf8(n):
    """The frame header's 'UTF-8-like' number: up to 36 bits, c continuation bytes carry 6 bits each."""
    if n < 0x80:
        return bytes([n])
    c = 1
    while n >> (5 * c + 6):
        c += 1
    out = [((0xFF << (7 - c)) & 0xFF) | (n >> (6 * c))]
    for i in range(c - 1, -1, -1):
        out.append(0x80 | ((n >> (6 * i)) & 0x3F))
    return bytes(out)


def _residual(bw, res, order, blocksize, porder, escape):
    bw.write(0, 2)                                 # 4-bit Rice parameters
    bw.write(porder, 4)
    pos = 0
    for part in range(1 << porder):
        n = (blocksize >> porder) - (order if part == 0 else 0)
        seg = res[pos:pos + n]
        pos += n
        if escape and part % 2 == 0:
            nb = max([0] + [int(abs(v)).bit_length() + 1 for v in seg]) if seg else 0
            bw.write(15, 4)
            bw.write(nb, 5)
            for v in seg:
                if nb:
                    bw.signed(v, nb)
        else:
            mean = (sum(abs(v) for v in seg) / max(1, len(seg)))
            k = min(14, max(0, int(np.log2(mean + 1))))
            bw.write(k, 4)
            for v in seg:
                u = (v << 1) if v >= 0 else ((-v) << 1) - 1
                bw.unary(u >> k)
                bw.write(u & ((1 << k) - 1), k)
    assert pos == len(res)


def _subframe(bw, s, bps, mode, rng):
    s = [int(v) for v in s]
    n = len(s)
    wasted = 0
    if any(s) and mode != "constant":
        while all((v >> wasted) & 1 == 0 for v in s) and wasted < bps - 1:
            wasted += 1
    if wasted:
        s = [v >> wasted for v in s]
        bps -= wasted
    bw.write(0, 1)
    if mode == "constant":
        assert all(v == s[0] for v in s)
        bw.write(0, 6); bw.write(0, 1)
        bw.signed(s[0], bps)
        return
    if mode == "verbatim":
        bw.write(1, 6)
    elif mode[0] == "fixed":
        bw.write(8 + mode[1], 6)
    else:
        bw.write(31 + mode[1], 6)
    if wasted:
        bw.write(1, 1); bw.unary(wasted - 1)
    else:
        bw.write(0, 1)
    if mode == "verbatim":
        for v in s:
            bw.signed(v, bps)
        return
    order = mode[1]
    for v in s[:order]:
        bw.signed(v, bps)
    if mode[0] == "fixed":
        c = FIXED[order]
        res = [s[i] - sum(cj * s[i - 1 - j] for j, cj in enumerate(c)) for i in range(order, n)]
        _residual(bw, res, order, n, mode[2], mode[3])
    else:
        _, order, prec, shift, porder = mode
        coef = [int(v) for v in rng.integers(-(1 << (prec - 3)), 1 << (prec - 3), size=order)]
        coef[0] = (1 << shift) - 3                 # a predictor that roughly follows the signal
        bw.write(prec - 1, 4)
        bw.signed(shift, 5)
        for cj in coef:
            bw.signed(cj, prec)
        res = [s[i] - (sum(coef[j] * s[i - 1 - j] for j in range(order)) >> shift) for i in range(order, n)]
        _residual(bw, res, order, n, porder, False)


def encode(x, rate, bps, blocks, rate_in_header="code", bps_in_header=True, variable=False, with_md5=True):
    x = np.asarray(x, dtype=np.int64)
    C, L = x.shape
    rng = np.random.default_rng(7)
    frames = []
    pos = 0
    fno = 0
    for (bs, mode, stereo) in blocks:
        bs = min(bs, L - pos)
        if bs <= 0:
            break
        seg = x[:, pos:pos + bs]
        bw = BitWriter()
        bw.write(0x3FFE, 14); bw.write(0, 1); bw.write(1 if variable else 0, 1)
        if bs in BLOCK_CODE:
            bs_code = BLOCK_CODE[bs]
        else:
            bs_code = 6 if bs <= 256 else 7
        bw.write(bs_code, 4)
        if rate_in_header == "code" and rate in RATE_CODE:
            sr_code = RATE_CODE[rate]
        elif rate_in_header == "khz" and rate % 1000 == 0 and rate // 1000 < 256:
            sr_code = 12
        elif rate_in_header == "hz" and rate < 65536:
            sr_code = 13
        elif rate_in_header == "tens" and rate % 10 == 0 and rate // 10 < 65536:
            sr_code = 14
        else:
            sr_code = 0
        bw.write(sr_code, 4)
        if C == 2 and stereo != "indep":
            bw.write({"ls": 8, "rs": 9, "ms": 10}[stereo], 4)
        else:
            bw.write(C - 1, 4)
        bw.write(BPS_CODE.get(bps, 0) if bps_in_header else 0, 3)
        bw.write(0, 1)
        for b in _utf8(pos if variable else fno):
            bw.write(b, 8)
        if bs_code == 6:
            bw.write(bs - 1, 8)
        elif bs_code == 7:
            bw.write(bs - 1, 16)
        if sr_code == 12:
            bw.write(rate // 1000, 8)
        elif sr_code == 13:
            bw.write(rate, 16)
        elif sr_code == 14:
            bw.write(rate // 10, 16)
        bw.write(_crc(bw.bytes(), 0x07, 8), 8)
        if C == 2 and stereo != "indep":
            l, r = seg[0], seg[1]
            side = l - r
            if stereo == "ls":
                _subframe(bw, l, bps, mode, rng); _subframe(bw, side, bps + 1, mode, rng)
            elif stereo == "rs":
                _subframe(bw, side, bps + 1, mode, rng); _subframe(bw, r, bps, mode, rng)
            else:
                _subframe(bw, (l + r) >> 1, bps, mode, rng); _subframe(bw, side, bps + 1, mode, rng)
        else:
            for c in range(C):
                _subframe(bw, seg[c], bps, mode, rng)
        bw.align()
        body = bw.bytes()
        frames.append(body + _crc(body, 0x8005, 16).to_bytes(2, "big"))
        pos += bs
        fno += 1
    assert pos == L, "blocks must cover the signal"
    nbytes = (bps + 7) // 8
    raw = b"".join(int(v).to_bytes(nbytes, "little", signed=True) for v in x.T.reshape(-1))
    md5 = hashlib.md5(raw).digest() if with_md5 else bytes(16)
    bsizes = [min(b[0], L) for b in blocks]
    info = BitWriter()
    info.write(min(bsizes), 16); info.write(max(bsizes), 16)
    info.write(0, 24); info.write(0, 24)
    info.write(rate, 20); info.write(C - 1, 3); info.write(bps - 1, 5); info.write(L, 36)
    streaminfo = info.bytes() + md5
    assert len(streaminfo) == 34
    pad = bytes([0x81]) + (8).to_bytes(3, "big") + bytes(8)             # a PADDING block, marked last
    return b"fLaC" + bytes([0x00]) + (34).to_bytes(3, "big") + streaminfo + pad + b"".join(frames)
