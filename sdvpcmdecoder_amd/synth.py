"""Synthetic STC-007 video generator (the build's own feeder; replaces the OUTPUT FORMAT of
VideoInFFMPEG::spliceFrame, vin_ffmpeg.cpp:213-364, for tests and the benchmark).

Everything here is data generation for tests/bench - it is not on the decode path.

Line layout (stc007line.h:79-101): 4 START bits "1010", 8 x 14-bit words (L0 R0 L1 R1 L2 R2 P Q),
16-bit CRCC, 5 STOP bits "01111"  = 137 bit cells.
Interleave (stc007deinterleaver.cpp:420-424): word k of block b sits in line b + 16*k, slot k.
P = XOR of the six audio words; Q = T^6 L0 + T^5 R0 + T^4 L1 + T^3 R1 + T^2 L2 + T R2 over GF(2)^14
(stc007deinterleaver.cpp:1297-1317), T = companion matrix of x^14 + x^8 + 1 (see q_matrix()).
"""
from __future__ import annotations

import numpy as np

BITS_IN_LINE = 137
WORD_MASK = 0x3FFF


def crc16_words14(words: np.ndarray) -> np.ndarray:
    """CRC-16/CCITT-FALSE over 8 x 14-bit words, MSB first (stc007line.cpp:245-251).
    words: (..., 8) integer array -> (...) uint16."""
    w = np.asarray(words).astype(np.uint32)
    crc = np.full(w.shape[:-1], 0xFFFF, dtype=np.uint32)
    for k in range(8):
        for bit in range(13, -1, -1):
            inb = (w[..., k] >> bit) & 1
            msb = (crc >> 15) & 1
            crc = (crc << 1) & 0xFFFF
            crc = np.where(msb != inb, crc ^ 0x1021, crc)
    return crc.astype(np.uint16)


def line_bits(words9: np.ndarray) -> np.ndarray:
    """(n, 9) words (8 x 14-bit + CRC) -> (n, 137) bit cells incl. START/STOP markers."""
    w = np.asarray(words9).astype(np.uint32)
    n = w.shape[0]
    bits = np.zeros((n, BITS_IN_LINE), dtype=np.uint8)
    bits[:, 0] = 1
    bits[:, 2] = 1
    pos = 4
    for k in range(8):
        for bit in range(13, -1, -1):
            bits[:, pos] = (w[:, k] >> bit) & 1
            pos += 1
    for bit in range(15, -1, -1):
        bits[:, pos] = (w[:, 8] >> bit) & 1
        pos += 1
    bits[:, pos + 1:pos + 5] = 1          # "01111"
    return bits


def render_lines(bits: np.ndarray, width: int = 720, black: int = 30, white: int = 200,
                 x0: int = 12, x1: int | None = None, shift: np.ndarray | None = None,
                 noise_sigma: float = 0.0, rng: np.random.Generator | None = None,
                 blur: int = 0) -> np.ndarray:
    """Nearest-cell rasterisation of (n, 137) bit cells into (n, width) uint8 luma.
    Data window [x0, x1); `shift` = per-line horizontal jitter in px; `blur` = box-blur radius."""
    n = bits.shape[0]
    if x1 is None:
        x1 = width - 12
    span = x1 - x0
    x = np.arange(width)
    if shift is None:
        shift = np.zeros(n, dtype=np.int64)
    xs = x[None, :] - shift[:, None]
    cell = ((xs - x0) * BITS_IN_LINE) // span
    inside = (xs >= x0) & (xs < x1)
    cell = np.clip(cell, 0, BITS_IN_LINE - 1)
    b = np.take_along_axis(bits, cell, axis=1)
    b = np.where(inside, b, 0)
    img = black + b.astype(np.float32) * (white - black)
    if blur > 0:
        k = 2 * blur + 1
        pad = np.pad(img, ((0, 0), (blur, blur)), mode="edge")
        cs = np.cumsum(pad, axis=1)
        cs = np.concatenate([np.zeros((n, 1), np.float32), cs], axis=1)
        img = (cs[:, k:] - cs[:, :-k]) / k
    if noise_sigma > 0:
        if rng is None:
            rng = np.random.default_rng(0)
        img = img + rng.normal(0.0, noise_sigma, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def random_lines(n: int, seed: int = 0, width: int = 720, **kw):
    """n independent random STC-007 lines. Returns (luma (n,width) u8, words (n,9) u16)."""
    rng = np.random.default_rng(seed)
    words = rng.integers(0, 1 << 14, size=(n, 8), dtype=np.uint32)
    crc = crc16_words14(words)
    w9 = np.concatenate([words, crc[:, None].astype(np.uint32)], axis=1).astype(np.uint16)
    luma = render_lines(line_bits(w9), width=width, rng=rng, **kw)
    return luma, w9
