"""Synthetic STC-007 video generator (the build's own feeder; replaces the OUTPUT FORMAT of
VideoInFFMPEG::spliceFrame, vin_ffmpeg.cpp:213-364, for tests and the benchmark).

Everything here is data generation for tests/bench - it is not on the decode path.

Line layout (stc007line.h:79-101): 4 START bits "1010", 8 x 14-bit words (L0 R0 L1 R1 L2 R2 P Q),
16-bit CRCC, 5 STOP bits "01111"  = 137 bit cells.
Interleave (stc007deinterleaver.cpp:420-424): word k of block b sits in line b + 16*k, slot k.
P = XOR of the six audio words; Q = T^6 L0 + T^5 R0 + T^4 L1 + T^3 R1 + T^2 L2 + T R2 over GF(2)^14
(stc007deinterleaver.cpp:1297-1317); T (row table stc007deinterleaver.cpp:8-11) is "multiply by x" in
GF(2)[x]/(x^14 + x^8 + 1): T v = (v << 1) ^ (0x0101 if v bit 13 else 0), see t_mul().
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
    """Nearest-cell rasterisation of (n, cells) bit cells (137 for STC-007, 94 for PCM-1) into (n, width) uint8 luma.
    Data window [x0, x1) - it may reach past the picture, the cells out there are cut off; `shift` = per-line horizontal
    jitter in px; `blur` = box-blur radius."""
    n = bits.shape[0]
    n_cells = bits.shape[1]
    if x1 is None:
        x1 = width - 12
    span = x1 - x0
    x = np.arange(width)
    if shift is None:
        shift = np.zeros(n, dtype=np.int64)
    xs = x[None, :] - shift[:, None]
    cell = ((xs - x0) * n_cells) // span
    inside = (xs >= x0) & (xs < x1)
    cell = np.clip(cell, 0, n_cells - 1)
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


def t_mul(v: np.ndarray) -> np.ndarray:
    """One application of the Q-code matrix T to 14-bit vectors (multiply by x mod x^14+x^8+1)."""
    v = np.asarray(v).astype(np.uint32)
    hi = (v >> 13) & 1
    return (((v << 1) & WORD_MASK) ^ (hi * 0x0101)).astype(np.uint32)


def pq_words(audio: np.ndarray):
    """audio (n, 6) 14-bit words L0 R0 L1 R1 L2 R2 -> (P, Q) per block
    (calcPcode/calcQcode, stc007deinterleaver.cpp:1297-1317)."""
    a = np.asarray(audio).astype(np.uint32)
    p = a[:, 0] ^ a[:, 1] ^ a[:, 2] ^ a[:, 3] ^ a[:, 4] ^ a[:, 5]
    q = np.zeros(a.shape[0], dtype=np.uint32)
    for k in range(6):
        q = t_mul(q ^ a[:, k])     # Horner: T^6 L0 + T^5 R0 + ... + T R2
    return p, q


def interleave_stream(audio: np.ndarray) -> np.ndarray:
    """(n_blocks, 6) audio words -> (n_blocks, 9) line words: line m, slot k = word k of block m-16k
    (stc007deinterleaver.cpp:420-424; zero before the stream starts), CRC appended."""
    n = audio.shape[0]
    p, q = pq_words(audio)
    blk = np.concatenate([np.asarray(audio).astype(np.uint32), p[:, None], q[:, None]], axis=1)   # (n, 8)
    lines = np.zeros((n, 8), dtype=np.uint32)
    for k in range(8):
        lines[16 * k:, k] = blk[:n - 16 * k, k]
    crc = crc16_words14(lines)
    return np.concatenate([lines, crc[:, None].astype(np.uint32)], axis=1).astype(np.uint16)


CTRL_BLOCK_WORDS = (0x3333, 0x0CCC, 0x3333, 0x0CCC, 0x0000)   # cue x4 + ID (stc007line.cpp:493-504)


def stc007_frames(n_frames: int, seed: int = 0, width: int = 720, height: int = 486,
                  lines_per_field: int = 245, cut_top: int | None = None, ctrl_block: bool = False,
                  audio: np.ndarray | None = None, silent: bool = False, words: np.ndarray | None = None,
                  bff: bool = False, cut_top_per_frame=None, **render_kw):
    """Synthetic STC-007 video: one continuous interleaved line stream, `lines_per_field` PCM lines per field
    (config.h:80-81), of which `height//2` are visible starting at `cut_top` (default: as many as do not fit
    are cut from the top, e.g. 2 for 486 rows).  Frame row 2r   <- odd field line cut_top + r,
                                                 frame row 2r+1 <- even field line cut_top + r
    (field/row order of VideoInFFMPEG::spliceFrame, vin_ffmpeg.cpp:281-347).
    ctrl_block=True overwrites stream line 0 of every field with a Control Block line.
    words: explicit (n_stream_lines, 9) line words (e.g. interleave_stream_f1() for 16-bit PCM-F1) instead of `audio`.
    bff: the even rows carry the field that is first in time (bottom field first).
    cut_top_per_frame: (n_frames, 2) per-field vertical offsets (tape tracking drift: the data moves inside the frame).
    Returns (luma (n_frames, height, width) u8, line words (n_stream_lines, 9) u16, audio (n_blocks, 6))."""
    rng = np.random.default_rng(seed)
    vis = height // 2
    if cut_top is None:
        cut_top = max(0, lines_per_field - vis)
    n_stream = n_frames * 2 * lines_per_field
    if audio is None:
        if silent:
            audio = np.zeros((n_stream, 6), dtype=np.uint32)
        else:
            audio = rng.integers(0, 1 << 14, size=(n_stream, 6), dtype=np.uint32)
    w9 = interleave_stream(audio) if words is None else np.array(words[:n_stream], dtype=np.uint16)
    if ctrl_block:
        cb = np.zeros(8, dtype=np.uint32)
        cb[:5] = CTRL_BLOCK_WORDS
        cb[7] = 0x0000          # control bits: P and Q present, emphasis on, copy allowed
        idx = np.arange(0, n_stream, lines_per_field)
        w9[idx, :8] = cb.astype(np.uint16)
        w9[idx, 8] = crc16_words14(cb[None, :])[0]
    # stream line index for every frame row
    f = np.arange(n_frames)[:, None]
    r = np.arange(vis)[None, :]
    if cut_top_per_frame is None:
        ct = np.full((n_frames, 2), cut_top, dtype=np.int64)
    else:
        ct = np.asarray(cut_top_per_frame, dtype=np.int64).reshape(n_frames, 2)
    first = f * 2 * lines_per_field + ct[:, 0:1] + r
    second = f * 2 * lines_per_field + lines_per_field + ct[:, 1:2] + r
    odd, even = (second, first) if bff else (first, second)
    idx = np.zeros((n_frames, height), dtype=np.int64)         # (a row behind the last full pair of an odd height stays blanking)
    idx[:, 0:2 * vis:2] = odd
    idx[:, 1:2 * vis:2] = even
    flat = idx.reshape(-1)
    # rows that fall outside their field's PCM lines (vertical offset too large) are blanking: no PCM there
    blank = np.ones((n_frames, height), dtype=bool)
    blank[:, 0:2 * vis:2] = ((ct[:, 1:2] if bff else ct[:, 0:1]) + r) >= lines_per_field
    blank[:, 1:2 * vis:2] = ((ct[:, 0:1] if bff else ct[:, 1:2]) + r) >= lines_per_field
    blank = blank.reshape(-1)
    luma = render_lines(line_bits(w9[np.minimum(flat, n_stream - 1)]), width=width, rng=rng, **render_kw)
    if blank.any():
        luma[blank] = np.uint8(render_kw.get("black", 30))
    return luma.reshape(n_frames, height, width), w9, audio


# ------------------------------------------------------------------------------------------------
# torch version of the frame generator (runs on the GPU so that the 10k-frame benchmark batch is
# created directly in HBM).  Same construction as stc007_frames(); its own RNG stream.
# ------------------------------------------------------------------------------------------------
def stc007_frames_torch(n_frames: int, seed: int = 0, device="cuda", width: int = 720, height: int = 486,
                        lines_per_field: int = 245, cut_top: int | None = None, black: int = 30, white: int = 200,
                        x0: int = 12, x1: int | None = None, noise_sigma: float = 0.0, chunk_frames: int = 256, cyclic: bool = False, frame_range=None):
    """Returns (luma (n_frames, height, width) uint8 on `device`, words (n_stream_lines, 9) int32 on `device`).
    cyclic=True interleaves the audio blocks around the end of the batch, so that playing the batch again and again is one
    seamless tape (used by the benchmark, which keeps a single batch resident in HBM).
    frame_range=(lo, hi) renders only those frames of the n_frames-frame tape (one rank's part of a sharded tape); the returned
    luma then has hi-lo frames, the words still describe the whole tape."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    vis = height // 2
    if cut_top is None:
        cut_top = max(0, lines_per_field - vis)
    if x1 is None:
        x1 = width - 12
    n_stream = n_frames * 2 * lines_per_field
    audio = torch.randint(0, 1 << 14, (n_stream, 6), generator=g, device=device, dtype=torch.int32)
    p = audio[:, 0] ^ audio[:, 1] ^ audio[:, 2] ^ audio[:, 3] ^ audio[:, 4] ^ audio[:, 5]
    q = torch.zeros(n_stream, dtype=torch.int32, device=device)
    for k in range(6):
        v = q ^ audio[:, k]
        q = ((v << 1) & WORD_MASK) ^ (((v >> 13) & 1) * 0x0101)
    blk = torch.cat([audio, p[:, None], q[:, None]], dim=1)
    lines = torch.zeros((n_stream, 8), dtype=torch.int32, device=device)
    for k in range(8):
        if cyclic:
            lines[:, k] = torch.roll(blk[:, k], 16 * k)
        else:
            lines[16 * k:, k] = blk[:n_stream - 16 * k, k]
    crc = torch.full((n_stream,), 0xFFFF, dtype=torch.int32, device=device)
    for k in range(8):
        for bit in range(13, -1, -1):
            inb = (lines[:, k] >> bit) & 1
            msb = (crc >> 15) & 1
            crc = (crc << 1) & 0xFFFF
            crc = torch.where(msb != inb, crc ^ 0x1021, crc)
    w9 = torch.cat([lines, crc[:, None]], dim=1)
    # raster
    x = torch.arange(width, device=device)
    cell = ((x - x0) * BITS_IN_LINE) // (x1 - x0)
    inside = (x >= x0) & (x < x1)
    cell = cell.clamp(0, BITS_IN_LINE - 1)
    r_lo, r_hi = (0, n_frames) if frame_range is None else frame_range
    luma = torch.empty((r_hi - r_lo, height, width), dtype=torch.uint8, device=device)
    if noise_sigma > 0 and r_lo > 0:
        g.manual_seed(seed * 1000003 + r_lo)         # noise of a part does not have to replay the parts before it
    for f0 in range(r_lo, r_hi, chunk_frames):
        f1 = min(r_hi, f0 + chunk_frames)
        f = torch.arange(f0, f1, device=device)[:, None]
        r = torch.arange(vis, device=device)[None, :]
        odd = f * 2 * lines_per_field + cut_top + r
        even = odd + lines_per_field
        idx = torch.empty((f1 - f0, height), dtype=torch.long, device=device)
        idx[:, 0:2 * vis:2] = odd
        idx[:, 1:2 * vis:2] = even
        w = w9[idx.reshape(-1)]                                          # (L, 9)
        bits = torch.zeros((w.shape[0], BITS_IN_LINE), dtype=torch.uint8, device=device)
        bits[:, 0] = 1
        bits[:, 2] = 1
        pos = 4
        for k in range(8):
            for bit in range(13, -1, -1):
                bits[:, pos] = ((w[:, k] >> bit) & 1).to(torch.uint8)
                pos += 1
        for bit in range(15, -1, -1):
            bits[:, pos] = ((w[:, 8] >> bit) & 1).to(torch.uint8)
            pos += 1
        bits[:, pos + 1:pos + 5] = 1
        b = torch.index_select(bits, 1, cell) * inside.to(torch.uint8)[None, :]
        img = b.to(torch.float32) * float(white - black) + float(black)
        if noise_sigma > 0:
            img = img + torch.randn(img.shape, generator=g, device=device) * noise_sigma
        luma[f0 - r_lo:f1 - r_lo] = img.round().clamp(0, 255).to(torch.uint8).reshape(f1 - f0, height, width)
    return luma, w9


def interleave_stream_f1(audio16: np.ndarray) -> np.ndarray:
    """PCM-F1 16-bit variant (stc007datablock.h:80-92, stc007deinterleaver.cpp:1230-1274): slots 0..6 carry the 14 MSBs
    of L0 R0 L1 R1 L2 R2 P (P = XOR of the six 16-bit words), slot 7 (the S-word) the 2 LSBs of the seven words that sit
    in the SAME line at shifts 12,10,8,6,4,2,0."""
    a = np.asarray(audio16).astype(np.uint32) & 0xFFFF
    n = a.shape[0]
    p = a[:, 0] ^ a[:, 1] ^ a[:, 2] ^ a[:, 3] ^ a[:, 4] ^ a[:, 5]
    blk = np.concatenate([a, p[:, None]], axis=1)                      # (n, 7) 16-bit
    full = np.zeros((n, 7), dtype=np.uint32)
    for k in range(7):
        full[16 * k:, k] = blk[:n - 16 * k, k]
    lines = np.zeros((n, 8), dtype=np.uint32)
    lines[:, :7] = full >> 2
    shifts = np.array([12, 10, 8, 6, 4, 2, 0], dtype=np.uint32)
    lines[:, 7] = ((full & 3) << shifts[None, :]).sum(axis=1)
    crc = crc16_words14(lines)
    return np.concatenate([lines, crc[:, None].astype(np.uint32)], axis=1).astype(np.uint16)


# ---- PCM-1 line streams (records of the PCM-1 back half, sdv_pcm1_line_rec) -----------------------------------------
PCM1_LINE_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (7,)), ("calc_crc", "<u2"),
                            ("ref_level", "u1"), ("picked_bits_left", "u1"), ("picked_bits_right", "u1"), ("service_type", "u1"),
                            ("flags", "u1"), ("_pad", "u1", (5,))])
assert PCM1_LINE_DTYPE.itemsize == 32


def pcm1_crc_words(words6):
    """PCM1Line::calcCRC over rows of six 13-bit words (vectorised restatement used only to BUILD inputs; the oracle's own
    orc_pcm1_crc_words and the reference's calcCRC are what the tests pin against each other)."""
    w = (~np.asarray(words6, dtype=np.uint32)) & 0x1FFF
    crc = np.full(w.shape[0], 0xFFFF, dtype=np.uint32)
    for i in range(6):
        for b in range(12, -1, -1):
            bit = (w[:, i] >> b) & 1
            top = ((crc >> 15) & 1) ^ bit
            crc = ((crc << 1) & 0xFFFF) ^ (top * 0x1021)
    return ((~crc) & 0xFFFF).astype(np.uint16)



def pcm1_line_bits(words7: np.ndarray) -> np.ndarray:
    """(n, 7) PCM-1 words (6 x 13 bit + CRCC) -> (n, 94) bit cells, MSB first (PCM1Line, pcm1line.h:59-96; no markers)."""
    w = np.asarray(words7).astype(np.uint32)
    bits = np.zeros((w.shape[0], 94), dtype=np.uint8)
    pos = 0
    for k in range(6):
        for bit in range(12, -1, -1):
            bits[:, pos] = (w[:, k] >> bit) & 1
            pos += 1
    for bit in range(15, -1, -1):
        bits[:, pos] = (w[:, 6] >> bit) & 1
        pos += 1
    return bits


def pcm1_random_lines(n: int, seed: int = 0, width: int = 720, x0: int = 4, x1: int | None = None, header_every: int = 0, **kw):
    """n independent random PCM-1 video lines.  Returns (luma (n, width) u8, words (n, 7) u16)."""
    rng = np.random.default_rng(seed)
    words = rng.integers(0, 1 << 13, size=(n, 6), dtype=np.uint32)
    if header_every:
        words[::header_every] = np.array([0x0666, 0x0CCC, 0x1999, 0x1333, 0x0666, 0x0CCC], dtype=np.uint32)
    crc = pcm1_crc_words(words).astype(np.uint32)
    if header_every:
        crc[::header_every] = 0xCCCC
    w7 = np.concatenate([words, crc[:, None]], axis=1).astype(np.uint16)
    luma = render_lines(pcm1_line_bits(w7), width=width, x0=x0, x1=(width - 4 if x1 is None else x1), rng=rng, **kw)
    return luma, w7


def pcm1_line_stream(n_frames, seed=0, lines=(245, 245), first=(1, 2), header=0, footer=0, p_bad=0.0, p_nobw=0.0, p_picked=0.0, p_forced=0.0,
                p_filler=0.0, noise_lines=0, new_file=False, end_file=False, empty=(), one_field=(), burst=None, first_frame=1):
    """The PCM1Line stream of a synthetic PCM-1 tape as the PCM-1 VideoToDigital branch would queue it: per frame the odd rows,
    END_FIELD, the even rows, END_FIELD, END_FRAME (videotodigital.cpp:1189-1383); header/footer rows as HEADER service lines."""
    rng = np.random.default_rng(seed)
    out = []

    def rec(frame, line, srv=0):
        r = np.zeros(1, dtype=PCM1_LINE_DTYPE)
        r["frame_number"] = frame
        r["line_number"] = line
        r["service_type"] = srv
        if srv:        # a cleared line (PCM1Line::clear): silent words, CRC_SILENT against its inverse
            r["words"][0, :6] = 1 << 12
            r["calc_crc"] = 0xECBF
            r["words"][0, 6] = 0xECBF ^ 0xFFFF
        return r

    for fi in range(n_frames):
        frame = first_frame + fi
        last_line = 0
        if new_file and fi == 0:
            out.append(rec(frame, 0, 1))
        for field in (0, 1):
            if fi in empty or (fi in one_field and field == 1):
                cnt = 0
            else:
                cnt = lines[field]
            ln = first[field]
            for _ in range(noise_lines if fi % 2 == 0 else 0):         # garbage rows above the data: B/W found, CRC bad
                r = rec(frame, ln)
                r["words"][0, :6] = rng.integers(0, 1 << 13, size=6)
                r["calc_crc"] = pcm1_crc_words(r["words"][:, :6])
                r["words"][0, 6] = r["calc_crc"][0] ^ np.uint16(rng.integers(1, 1 << 16))
                r["flags"] = 8
                r["ref_level"] = rng.integers(40, 200)
                out.append(r)
                ln += 2
            for _ in range(header):
                out.append(rec(frame, ln, 6))
                ln += 2
            block = np.zeros(cnt, dtype=PCM1_LINE_DTYPE)
            block["frame_number"] = frame
            block["line_number"] = ln + 2 * np.arange(cnt)
            block["words"][:, :6] = rng.integers(0, 1 << 13, size=(cnt, 6))
            small = rng.random((cnt, 6)) < 0.5                          # half of the words in the fine range (range bit clear)
            block["words"][:, :6] = np.where(small, block["words"][:, :6] & 0x0FFF, block["words"][:, :6])
            block["calc_crc"] = pcm1_crc_words(block["words"][:, :6]) if cnt else 0
            block["words"][:, 6] = block["calc_crc"]
            block["flags"] = 8
            block["ref_level"] = rng.integers(60, 180, size=cnt)
            bad = rng.random(cnt) < p_bad
            block["words"][bad, 6] ^= rng.integers(1, 1 << 16, size=int(bad.sum())).astype(np.uint16)
            nobw = rng.random(cnt) < p_nobw
            block["flags"][nobw] = 0
            pk = rng.random(cnt) < p_picked
            block["picked_bits_left"][pk] = rng.integers(0, 4, size=int(pk.sum()))
            block["picked_bits_right"][pk] = rng.integers(0, 3, size=int(pk.sum()))
            fb = rng.random(cnt) < p_forced
            block["flags"][fb] |= 32
            fil = rng.random(cnt) < p_filler
            for i in np.nonzero(fil)[0]:
                block[i] = rec(frame, block["line_number"][i], 3)[0]
            if burst and fi == 1 and field == 0 and cnt:
                s, length = burst
                s = min(s, 2 * cnt) // 2
                block["words"][s:s + length // 2, 6] ^= 0x5A5A
            out.append(block)
            ln += 2 * cnt
            for _ in range(footer):
                out.append(rec(frame, ln, 6))
                ln += 2
            out.append(rec(frame, ln, 4))
            last_line = max(last_line, ln)
        out.append(rec(frame, last_line + 2, 5))
    if end_file:
        frame = first_frame + n_frames
        for field in (0, 1):
            for ln in range(1 + field, 491, 2):
                out.append(rec(frame, ln, 3))
            out.append(rec(frame, 491 + field, 4))
        out.append(rec(frame, 494, 2))
        out.append(rec(frame, 496, 5))
    return np.concatenate(out)


def pcm1_tape(n_frames, seed=5, period=100, **kw):
    """`n_frames` of PCM-1 line records for throughput runs: a `period`-frame damaged tape repeated with continuing frame numbers."""
    kw.setdefault("p_bad", 0.02)
    kw.setdefault("header", 2)
    base = pcm1_line_stream(period, seed=seed, **kw)
    tiles = []
    for t in range((n_frames + period - 1) // period):
        b = base.copy()
        b["frame_number"] += period * t
        tiles.append(b)
    recs = np.concatenate(tiles)
    ends = np.nonzero(recs["service_type"] == 5)[0]
    return recs[:ends[n_frames - 1] + 1]


def pcm1_frames(n_frames: int, seed: int = 0, width: int = 720, height: int = 486, x0: int = 4, x1: int | None = None,
                header: int = 1, top_blank: int = 0, jitter: int = 0, p_dropout: float = 0.0, dup_every: int = 0,
                silent_from: int | None = None, **kw):
    """Synthetic PCM-1 video frames: every row of a field is one PCM-1 line (random 13-bit words, CRCC), the first `header`
    PCM rows of a field carry the Header pattern (pcm1line.cpp:314-323), `top_blank` rows above them are black.
    `jitter` = per-line horizontal shift of up to +-jitter px, `p_dropout` = share of rows replaced by black,
    `dup_every` = every that-many-th row repeats the row above it in its field (a dropout compensator at work),
    `silent_from` = frames from that index on carry silence.  Further keywords go to render_lines (black, white, noise_sigma,
    blur).  Returns (luma (n_frames, height, width) u8, words (n_frames * height, 7) u16 in row order)."""
    rng = np.random.default_rng(seed)
    n = n_frames * height
    words = rng.integers(0, 1 << 13, size=(n, 6), dtype=np.uint32)
    if silent_from is not None:
        words[silent_from * height:] = 0
    crc = pcm1_crc_words(words).astype(np.uint32)
    row = np.arange(n) % height
    in_field = row // 2                                      # position of the row in its field
    is_header = (in_field >= top_blank) & (in_field < top_blank + header)
    words[is_header] = np.array([0x0666, 0x0CCC, 0x1999, 0x1333, 0x0666, 0x0CCC], dtype=np.uint32)
    crc[is_header] = 0xCCCC
    if dup_every:
        src = np.arange(n)
        dup = (in_field % dup_every == dup_every - 1) & (in_field >= top_blank + header + 1)
        src[dup] -= 2
        words, crc = words[src], crc[src]
    w7 = np.concatenate([words, crc[:, None]], axis=1).astype(np.uint16)
    shift = rng.integers(-jitter, jitter + 1, size=n) if jitter else None
    luma = render_lines(pcm1_line_bits(w7), width=width, x0=x0, x1=(width - 4 if x1 is None else x1), shift=shift, rng=rng, **kw)
    black = kw.get("black", 30)
    luma[in_field < top_blank] = black
    if p_dropout > 0:
        luma[rng.random(n) < p_dropout] = black
    return luma.reshape(n_frames, height, width), w7


# ---- PCM-16x0 (Sony PCM-1610/1620/1630): 193 bit cells per line = 3 sub-lines of 3 x 16 bit + CRCC, one control bit ----------
PCM16X0_BIN_DTYPE = np.dtype([("frame_number", "<u4"), ("line_number", "<u2"), ("words", "<u2", (4,)), ("calc_crc", "<u2"),
                              ("data_start", "<i2"), ("data_stop", "<i2"), ("queue_order", "<u2"),
                              ("black_level", "u1"), ("white_level", "u1"), ("ref_low", "u1"), ("ref_level", "u1"), ("ref_high", "u1"),
                              ("hysteresis_depth", "u1"), ("shift_stage", "u1"), ("service_type", "u1"),
                              ("picked_bits_left", "u1"), ("picked_bits_right", "u1"), ("flags", "u1"), ("line_part", "u1"),
                              ("control_bit", "u1"), ("_pad", "u1")])
assert PCM16X0_BIN_DTYPE.itemsize == 36


def pcm16x0_crc_words(words3):
    """PCM16X0SubLine::calcCRC (pcm16x0subline.cpp:158-170): CRC-16/CCITT-FALSE over three 16-bit words, MSB first
    (vectorised restatement used only to BUILD inputs)."""
    w = np.asarray(words3, dtype=np.uint32)
    crc = np.full(w.shape[:-1], 0xFFFF, dtype=np.uint32)
    for i in range(3):
        for b in range(15, -1, -1):
            bit = (w[..., i] >> b) & 1
            top = ((crc >> 15) & 1) ^ bit
            crc = ((crc << 1) & 0xFFFF) ^ (top * 0x1021)
    return crc.astype(np.uint16)


def pcm16x0_line_bits(words: np.ndarray, control: np.ndarray | None = None) -> np.ndarray:
    """(n, 3, 4) sub-line words (3 x 16 bit + CRCC per sub-line) -> (n, 193) bit cells: sub-lines 0 and 1, the control bit (cell
    128), sub-line 2 (pcm16x0subline.h:90-100, binarizer.cpp:7151-7198)."""
    w = np.asarray(words).astype(np.uint32)
    n = w.shape[0]
    bits = np.zeros((n, 193), dtype=np.uint8)
    for part in range(3):
        pos = part * 64 + (1 if part == 2 else 0)
        for k in range(4):
            for bit in range(15, -1, -1):
                bits[:, pos] = (w[:, part, k] >> bit) & 1
                pos += 1
    bits[:, 128] = 1 if control is None else np.asarray(control).astype(np.uint8)
    return bits


def pcm16x0_random_lines(n: int, seed: int = 0, width: int = 720, x0: int = 4, x1: int | None = None, control=None, silent: bool = False, **kw):
    """n independent random PCM-16x0 video lines.  Returns (luma (n, width) u8, words (n, 3, 4) u16)."""
    rng = np.random.default_rng(seed)
    words = rng.integers(0, 1 << 16, size=(n, 3, 3), dtype=np.uint32)
    if silent:
        words[:] = 0
    crc = pcm16x0_crc_words(words).astype(np.uint32)
    w4 = np.concatenate([words, crc[..., None]], axis=2).astype(np.uint16)
    ctl = None if control is None else (rng.integers(0, 2, size=n) if control == "random" else np.full(n, control))
    luma = render_lines(pcm16x0_line_bits(w4, ctl), width=width, x0=x0, x1=(width - 4 if x1 is None else x1), rng=rng, **kw)
    return luma, w4


def pcm16x0_frames(n_frames: int, seed: int = 0, width: int = 720, height: int = 486, x0: int = 4, x1: int | None = None,
                   top_blank: int = 0, jitter: int = 0, p_dropout: float = 0.0, dup_every: int = 0, silent_from: int | None = None,
                   control=None, smear: tuple | None = None, **kw):
    """Synthetic PCM-16x0 video frames: every row of a field is one PCM-16x0 line (three sub-lines of random 16-bit words + CRCC),
    `top_blank` rows at the top of a field are black.  `jitter` = per-line horizontal shift of up to +-jitter px, `p_dropout` = share
    of rows replaced by black, `dup_every` = every that-many-th row repeats the row above it in its field, `silent_from` = frames from
    that index on carry silence, `control` = None (bit set) / 0 / 1 / "random", `smear` = (every, x_from, x_to): a stretch of every
    that-many-th row wiped out.  Further keywords go to render_lines.  Returns (luma (n_frames, height, width) u8, words
    (n_frames * height, 3, 4) u16 in row order)."""
    rng = np.random.default_rng(seed)
    n = n_frames * height
    words = rng.integers(0, 1 << 16, size=(n, 3, 3), dtype=np.uint32)
    if silent_from is not None:
        words[silent_from * height:] = 0
    row = np.arange(n) % height
    in_field = row // 2
    if dup_every:
        src = np.arange(n)
        dup = (in_field % dup_every == dup_every - 1) & (in_field >= top_blank + 1)
        src[dup] -= 2
        words = words[src]
    crc = pcm16x0_crc_words(words).astype(np.uint32)
    w4 = np.concatenate([words, crc[..., None]], axis=2).astype(np.uint16)
    ctl = None if control is None else (rng.integers(0, 2, size=n) if control == "random" else np.full(n, control))
    shift = rng.integers(-jitter, jitter + 1, size=n) if jitter else None
    luma = render_lines(pcm16x0_line_bits(w4, ctl), width=width, x0=x0, x1=(width - 4 if x1 is None else x1), shift=shift, rng=rng, **kw)
    black = kw.get("black", 30)
    luma[in_field < top_blank] = black
    if smear is not None:
        luma[(in_field % smear[0]) == smear[0] - 1, smear[1]:smear[2]] = 110
    if p_dropout > 0:
        luma[rng.random(n) < p_dropout] = black
    return luma.reshape(n_frames, height, width), w4


# ---- PCM-16x0 sub-line streams (records of the PCM-16x0 back half, sdv_pcm16x0_bin_rec) ---------------------------------
def pcm16x0_encode_fields(audio: np.ndarray, ei: bool = False) -> np.ndarray:
    """Audio sample pairs (n_frames * 1470, 2) i16 -> sub-line words (n_frames, 2, 735, 3) u16 of the two fields of every frame in
    playback order.  A data block holds three sub-blocks (L, R, P = L ^ R) and sits on the sub-lines s, s + 35, s + 70 of a
    105-sub-line interleave block (SI) or s, s + 490, s + 980 of the frame (EI) (pcm16x0datablock.h:38-91); which of L / R sits on
    the first line alternates with the sub-block and with the block (PCM16X0DataBlock::getWordToLine, pcm16x0datablock.cpp:1029-1155)."""
    a = np.asarray(audio).astype(np.int64) & 0xFFFF
    n_frames = a.shape[0] // 1470
    a = a[:n_frames * 1470].reshape(n_frames, 490, 3, 2)              # [frame][block][sub-block][L, R]
    out = np.zeros((n_frames, 1470, 3), dtype=np.uint16)
    blk = np.arange(490)
    if ei:
        s1 = blk; step = 490; even = (blk % 2) == 1
    else:
        s1 = (blk // 35) * 105 + blk % 35; step = 35; even = ((blk % 35) % 2) == 1
    for k in range(3):
        l_first = ((k & 1) != 0) != even                              # L on LINE_1
        lw, rw = a[:, :, k, 0], a[:, :, k, 1]
        out[:, s1, k] = np.where(l_first[None, :], lw, rw)
        out[:, s1 + 2 * step, k] = np.where(l_first[None, :], rw, lw)
        out[:, s1 + step, k] = lw ^ rw
    return out.reshape(n_frames, 2, 735, 3)


def pcm16x0_sub_stream(n_frames, seed=0, ei=False, first=(1, 2), cut=(0, 0), tail_cut=(0, 0), lead=(0, 0), trail=(0, 0), bff=False,
                       emphasis=False, rate_44100=False, code=False, ei_bit=None, silent=(), quiet=(), p_bad=0.0, p_nobw=0.0, p_picked=0.0, p_forced=0.0,
                       burst=None, wander=None, new_file=False, end_file=False, empty=(), one_field=(), first_frame=1, amplitude=1 << 15):
    """The PCM16X0SubLine stream of a synthetic PCM-1630 tape as the PCM-16x0 VideoToDigital branch queues it: per frame the odd
    rows (three sub-lines each), END_FIELD, the even rows, END_FIELD, END_FRAME.  `cut` / `tail_cut` = PCM lines the capture lost at
    the top / bottom of the (odd, even) field, `lead` / `trail` = rows of noise (levels found, CRC bad) above / below the data,
    `wander` = (period, amount): the top cut of both fields moves by up to `amount` lines every `period` frames, `silent` = frames
    of digital silence, `quiet` = frames of near-silence (+-3), p_* = share of sub-lines damaged / without levels / completed by the Bit
    Picker / forced bad, `burst` = (frame, field, first_line, lines): a dropout.  Returns (records, audio (n_frames * 1470, 2) i16)."""
    rng = np.random.default_rng(seed)
    audio = rng.integers(-amplitude, amplitude, size=(n_frames * 1470, 2)).astype(np.int16)
    for fi in silent:
        audio[fi * 1470:(fi + 1) * 1470] = 0
    for fi in quiet:
        audio[fi * 1470:(fi + 1) * 1470] = rng.integers(-3, 4, size=(1470, 2))
    words = pcm16x0_encode_fields(audio, ei=ei)                        # [frame][field in playback order][735][3]
    ctrl = np.ones((735,), dtype=np.uint8)
    for b in range(7):
        base = b * 105 + 1
        if emphasis: ctrl[base + 0] = 0
        if rate_44100: ctrl[base + 3] = 0
        if (ei if ei_bit is None else ei_bit): ctrl[base + 6] = 0
        if code: ctrl[base + 9] = 0
    out = []

    def srv(frame, line, st):
        r = np.zeros(1, dtype=PCM16X0_BIN_DTYPE)
        r["frame_number"] = frame; r["line_number"] = line; r["service_type"] = st
        r["words"][0, 3] = 0x0E10 ^ 0xFFFF; r["control_bit"] = 1
        return r

    def noise_rows(frame, ln, cnt):
        blk = np.zeros(cnt * 3, dtype=PCM16X0_BIN_DTYPE)
        if cnt == 0:
            return blk, ln
        blk["frame_number"] = frame
        blk["line_number"] = np.repeat(ln + 2 * np.arange(cnt), 3)
        blk["line_part"] = np.tile(np.arange(3), cnt)
        blk["words"][:, :3] = rng.integers(0, 1 << 16, size=(cnt * 3, 3))
        blk["calc_crc"] = pcm16x0_crc_words(blk["words"][:, :3])
        blk["words"][:, 3] = blk["calc_crc"] ^ rng.integers(1, 1 << 16, size=cnt * 3).astype(np.uint16)
        blk["control_bit"] = rng.integers(0, 2, size=cnt * 3)
        blk["flags"] = 8 | 16
        blk["data_start"] = 12; blk["data_stop"] = 700
        blk["black_level"] = 30; blk["white_level"] = 200; blk["ref_level"] = rng.integers(60, 180, size=cnt * 3)
        return blk, ln + 2 * cnt

    for fi in range(n_frames):
        frame = first_frame + fi
        if new_file and fi == 0:
            out.append(srv(frame, 0, 1))
        last_line = 0
        shift = 0
        if wander:
            shift = int(rng.integers(0, wander[1] + 1)) if (fi // wander[0]) % 2 else 0
        for parity in (0, 1):                                          # odd rows first, as the video decoder delivers them
            play = parity if not bff else 1 - parity                   # which field of the frame (in playback order) these rows carry
            ln = first[parity]
            if fi in empty or (fi in one_field and parity == 1):
                out.append(srv(frame, ln, 4)); last_line = max(last_line, ln)
                continue
            nb, ln = noise_rows(frame, ln, lead[parity]); out.append(nb)
            lo = (cut[parity] + shift) * 3
            hi = 735 - tail_cut[parity] * 3
            cnt = hi - lo
            blk = np.zeros(cnt, dtype=PCM16X0_BIN_DTYPE)
            blk["frame_number"] = frame
            blk["line_number"] = ln + 2 * (np.arange(cnt) // 3)
            blk["line_part"] = np.arange(lo, hi) % 3
            blk["queue_order"] = np.arange(cnt) // 3
            blk["words"][:, :3] = words[fi, play, lo:hi]
            blk["calc_crc"] = pcm16x0_crc_words(blk["words"][:, :3])
            blk["words"][:, 3] = blk["calc_crc"]
            blk["control_bit"] = ctrl[lo:hi]
            blk["flags"] = 8 | 16
            blk["data_start"] = 12; blk["data_stop"] = 700
            blk["black_level"] = 30; blk["white_level"] = 200; blk["ref_low"] = 100; blk["ref_high"] = 130
            blk["ref_level"] = rng.integers(60, 180, size=cnt)
            bad = rng.random(cnt) < p_bad
            if burst and burst[0] == fi and burst[1] == parity:
                bad[max(0, burst[2] * 3 - lo):max(0, (burst[2] + burst[3]) * 3 - lo)] = True
            blk["words"][bad, :3] ^= rng.integers(0, 1 << 16, size=(int(bad.sum()), 3)).astype(np.uint16) & rng.integers(0, 1 << 16, size=(int(bad.sum()), 3)).astype(np.uint16)
            blk["calc_crc"][bad] = pcm16x0_crc_words(blk["words"][bad, :3])
            same = bad & (blk["calc_crc"] == blk["words"][:, 3])
            blk["words"][same, 3] ^= 0x0101
            nobw = rng.random(cnt) < p_nobw
            blk["flags"][nobw] &= 0xFF ^ 8
            pk = rng.random(cnt) < p_picked
            blk["picked_bits_left"][pk & (blk["line_part"] == 0)] = rng.integers(1, 4, size=int((pk & (blk["line_part"] == 0)).sum()))
            blk["picked_bits_right"][pk & (blk["line_part"] == 2)] = rng.integers(1, 3, size=int((pk & (blk["line_part"] == 2)).sum()))
            # a completed line is one that read wrong and was "repaired": some of them wrongly (data still off, CRC made to fit)
            wrong = pk & (rng.random(cnt) < 0.5) & (blk["picked_bits_left"] + blk["picked_bits_right"] > 0)
            blk["words"][wrong, 0] ^= (rng.integers(1, 8, size=int(wrong.sum())) << 13).astype(np.uint16)
            blk["calc_crc"][wrong] = pcm16x0_crc_words(blk["words"][wrong, :3])
            blk["words"][wrong, 3] = blk["calc_crc"][wrong]
            fb = rng.random(cnt) < p_forced
            blk["flags"][fb] |= 32
            out.append(blk)
            ln += 2 * (cnt // 3)
            nb, ln = noise_rows(frame, ln, trail[parity]); out.append(nb)
            out.append(srv(frame, ln, 4))
            last_line = max(last_line, ln)
        out.append(srv(frame, last_line + 2, 5))
    if end_file:
        frame = first_frame + n_frames
        for field in (0, 1):
            for ln in range(1 + field, 491, 2):
                out.append(srv(frame, ln, 3))
            out.append(srv(frame, 491 + field, 4))
        out.append(srv(frame, 494, 2))
        out.append(srv(frame, 496, 5))
    return np.concatenate(out), audio


def pcm16x0_tape(n_frames, seed=5, period=50, **kw):
    """`n_frames` of PCM-16x0 sub-line records for throughput runs: a `period`-frame damaged tape repeated with continuing frame numbers."""
    kw.setdefault("p_bad", 0.02)
    kw.setdefault("cut", (6, 9))
    kw.setdefault("tail_cut", (3, 4))
    kw.setdefault("rate_44100", True)
    base, _ = pcm16x0_sub_stream(period, seed=seed, **kw)
    tiles = []
    for t in range((n_frames + period - 1) // period):
        b = base.copy()
        b["frame_number"] += period * t
        tiles.append(b)
    recs = np.concatenate(tiles)
    ends = np.nonzero(recs["service_type"] == 5)[0]
    return recs[:ends[n_frames - 1] + 1]


def pcm16x0_tape_frames(n_frames: int, seed: int = 0, ei: bool = False, width: int = 720, height: int = 486, first_pcm_line=(0, 0), bff: bool = False,
                        rate_44100: bool = True, emphasis: bool = False, code: bool = False, x0: int = 4, x1: int | None = None, p_dropout: float = 0.0,
                        amplitude: int = 1 << 15, **kw):
    """Video of a PCM-1630 tape: audio -> data blocks -> interleave (SI or EI) -> 245 PCM lines per field with their Control Bits ->
    luma.  Row r of a frame belongs to field r % 2 (odd rows of the reference = even r) and shows PCM line first_pcm_line[field] + r // 2 of that
    field: a 486-row capture sees 243 of the 245 lines of either field.  Returns (luma (n_frames, height, width) u8, audio
    (n_frames * 1470, 2) i16, seen (2, 735) bool: the sub-lines of a [odd rows, even rows] field that are in the picture)."""
    rng = np.random.default_rng(seed)
    audio = rng.integers(-amplitude, amplitude, size=(n_frames * 1470, 2)).astype(np.int16)
    words = pcm16x0_encode_fields(audio, ei=ei)                        # [frame][field in playback order][735][3]
    crc = pcm16x0_crc_words(words.reshape(-1, 3)).reshape(n_frames, 2, 735)
    w4 = np.concatenate([words, crc[..., None]], axis=3).reshape(n_frames, 2, 245, 3, 4)
    ctrl = np.ones((245,), dtype=np.uint8)                            # the Control Bit sits between the second and the third sub-line of a line
    for b in range(7):
        line = b * 35
        if emphasis: ctrl[line + 0] = 0
        if rate_44100: ctrl[line + 1] = 0
        if ei: ctrl[line + 2] = 0
        if code: ctrl[line + 3] = 0
    rows_w = np.zeros((n_frames, height, 3, 4), dtype=np.uint16)
    rows_c = np.ones((n_frames, height), dtype=np.uint8)
    blank = np.ones((height,), dtype=bool)
    seen = np.zeros((2, 735), dtype=bool)
    for r in range(height):
        parity = r % 2
        play = parity if not bff else 1 - parity
        pl = first_pcm_line[parity] + r // 2
        if 0 <= pl < 245:
            rows_w[:, r] = w4[:, play, pl]
            rows_c[:, r] = ctrl[pl]
            blank[r] = False
            seen[parity, 3 * pl:3 * pl + 3] = True
    n = n_frames * height
    luma = render_lines(pcm16x0_line_bits(rows_w.reshape(n, 3, 4), rows_c.reshape(n)), width=width, x0=x0, x1=(width - 4 if x1 is None else x1), rng=rng, **kw)
    luma = luma.reshape(n_frames, height, width)
    black = kw.get("black", 30)
    luma[:, blank] = black
    if p_dropout > 0:
        luma[rng.random((n_frames, height)) < p_dropout] = black
    return luma, audio, seen
