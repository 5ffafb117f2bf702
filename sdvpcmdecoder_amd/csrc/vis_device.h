/*
 * vis_device.h - the canvases of RenderPCM's "binarized lines" visualiser, drawn from device-resident line records.
 *
 * Reference: RenderPCM::renderNewLine(STC007Line / PCM1Line / PCM16X0SubLine) (renderpcm.cpp:939-1169, 489-624, 743-936), fed with every
 * line VideoToDigital queues that is no service line or is a filler (videotodigital.cpp:398-402, 452-456, 507-511), and
 * prepareNewFrame() per binarized frame (renderpcm.cpp:176-186; wiring mainwindow.cpp:1949-1990).  The reference draws one line at a
 * time into one QImage that is never cleared: a frame's canvas is what the frame drew over what the frames before it left there.
 *
 * Here all frames of a batch are drawn at once - a record's row is its rank among the drawn records of its frame (PCM-16x0: among the
 * right-hand sub-lines, the row advances with those) - into one canvas per frame, and a second pass carries what earlier frames (or
 * the canvas kept from the call before) left in the places a frame did not draw.  Both passes are plain streaming writes: 4 bytes per
 * pixel leave, 36-40 bytes per record come in; the bound is HBM write bandwidth.
 *
 * A place ("cell") is a row of the canvas; for PCM-16x0 a third of a row (the middle one with the control bit).  Where two records of
 * a frame land in the same cell (sub-lines whose right-hand part was lost), the later one wins like in the reference's sequential
 * drawing: the earlier one is not drawn at all.
 */
#pragma once
#include <stdint.h>
#include "../../include/sdvpcm.h"
#include "stc007_stitch_device.h"

namespace sdvvis {
using sdvs::lanemask_lt;

enum { PX_BLK = 2u,                  /* VIS_BIT0_BLK is Qt::black, the GlobalColor enumerator: the reference stores its value, 2 (renderpcm.h:52) */
       B0_GRY = 0xFF2D2D2Du, B1_GRY = 0xFF969696u, B0_YEL = 0xFF7F6E00u, B1_YEL = 0xFFFFDC00u, B0_GRN = 0xFF005F1Eu, B1_GRN = 0xFF00E146u,
       B0_RED = 0xFF8C0000u, B1_RED = 0xFFFF462Bu, B0_BLU = 0xFF005F7Fu, B1_BLU = 0xFF00BFFFu, B0_MGN = 0xFF8C008Cu, B1_MGN = 0xFFFF00FFu,
       B1_MARK = 0xFFFFFFFFu,        /* renderpcm.h:53-65 */
       LIM_OK = 0xFFFFFFFFu, LIM_MARK = 0xFFE0AAAAu,      /* VIS_LIM_OK, VIS_LIM_MARK :66-67 */
       BLANK = 0xFF000000u };        /* a canvas nothing was drawn on: QImage::fill(Qt::black) */

struct Geometry { uint32_t w, h, cells_per_row; };
__host__ __device__ inline Geometry geometry(int kind)
{
    Geometry g = { 0, 0, 0 };
    if (kind == SDV_VIS_STC007_LINES) { g.w = 5 * 137; g.h = 650; g.cells_per_row = 1; }        /* startSTC007NTSCFrame + setLineCount(VID_UNKNOWN) */
    else if (kind == SDV_VIS_PCM1_LINES) { g.w = 8 * 94; g.h = 490; g.cells_per_row = 1; }      /* startPCM1Frame */
    else if (kind == SDV_VIS_PCM16X0_LINES) { g.w = 4 * 193; g.h = 490; g.cells_per_row = 3; }  /* startPCM1600Frame */
    else if (kind == SDV_VIS_STC007_BLOCKS_NTSC) { g.w = 6 * 109; g.h = 490; g.cells_per_row = 1; }     /* startSTC007DBFrame, setLineCount(VID_NTSC) */
    else if (kind == SDV_VIS_STC007_BLOCKS_PAL) { g.w = 6 * 109; g.h = 588; g.cells_per_row = 1; }      /* ... setLineCount(VID_PAL) */
    else if (kind == SDV_VIS_STC007_ASM_NTSC) { g.w = 5 * 137; g.h = 490; g.cells_per_row = 1; }         /* startSTC007NTSCFrame, setLineCount(VID_NTSC): the assembled lines */
    else if (kind == SDV_VIS_STC007_ASM_PAL) { g.w = 5 * 137; g.h = 588; g.cells_per_row = 1; }
    else if (kind == SDV_VIS_PCM1_BLOCKS) { g.w = 6 * (8 + 3 + 8 * 16 + 4); g.h = 23 * 8 * 2; g.cells_per_row = 1; }     /* startPCM1DBFrame: 858 x 368 */
    else if (kind == SDV_VIS_PCM1_ASM) { g.w = 8 * 78; g.h = 490; g.cells_per_row = 3; }                             /* startPCM1SubFrame */
    else if (kind == SDV_VIS_PCM16X0_BLOCKS) { g.w = 6 * (9 + 16 * 6 + 8); g.h = 490; g.cells_per_row = 1; }          /* startPCM1600DBFrame: 678 x 490 */
    return g;
}
/* pixels [x0, x1) of a cell */
__host__ __device__ inline void cell_span(int kind, uint32_t part, uint32_t &x0, uint32_t &x1)
{
    if (kind == SDV_VIS_PCM1_ASM) { x0 = part * 26u * 8u; x1 = x0 + 26u * 8u; return; }        /* renderpcm.cpp:647-655 */
    if (kind != SDV_VIS_PCM16X0_LINES) { x0 = 0; x1 = geometry(kind).w; return; }
    x0 = part == 0 ? 0u : part == 1 ? 4u * 64u : 4u * 129u;
    x1 = part == 0 ? 4u * 64u : part == 1 ? 4u * 129u : 4u * 193u;
}

struct VisArgs {
    const void *recs; uint32_t n_recs; int kind;
    uint32_t *cnt;                  /* per 64 records: END_FRAME records | records that end a row << 16 */
    const uint32_t *base_end, *base_adv;    /* per 64 records: how many of either came before */
    uint32_t *frame_adv0;           /* per frame: rows ended before the frame began */
    uint32_t n_frames;
    uint32_t *out;                  /* n_frames canvases */
    uint32_t *wmask; uint32_t wmask_stride;     /* per frame: one bit per cell the frame drew */
    const uint32_t *canvas;         /* what the frames before this call left */
    int32_t *last_drawn; uint32_t n_cells;      /* per frame and cell: the last frame up to this one that drew the cell (-1: none in this call) */
};

/* what the passes need to know of record i: END_FRAME?, drawn?, the third of the row it lands in, does the row end with it */
struct Rec { bool end, drawn, adv; uint32_t part; };
template <class R> __device__ inline Rec classify(const R *recs, uint32_t i, uint32_t n, int kind)
{
    Rec c; c.end = c.drawn = c.adv = false; c.part = 0;
    if (i >= n) return c;
    const uint8_t srv = recs[i].service_type;
    c.end = srv == SDV_SRV_END_FRAME;
    c.drawn = srv == SDV_SRV_NO || srv == SDV_SRV_FILLER;
    c.adv = c.drawn;
    return c;
}
template <> __device__ inline Rec classify<sdv_pcm16x0_bin_rec>(const sdv_pcm16x0_bin_rec *recs, uint32_t i, uint32_t n, int kind)
{
    Rec c; c.end = c.drawn = c.adv = false; c.part = 0;
    if (i >= n) return c;
    const uint8_t srv = recs[i].service_type;
    c.end = srv == SDV_SRV_END_FRAME;
    c.drawn = srv == SDV_SRV_NO || srv == SDV_SRV_FILLER;
    const uint32_t part = srv == SDV_SRV_NO ? recs[i].line_part : 0u;      /* a filler is a cleared sub-line: PART_LEFT (pcm16x0subline.cpp:63-86) */
    c.adv = c.drawn && part == 2;                                           /* the row ends with a PART_RIGHT sub-line (renderpcm.cpp:929-935) */
    c.part = part > 2 ? 0u : part;                                          /* :771-773 */
    return c;
}

/* the sub-lines of the PCM-1 stitcher's queue: 1470 records are a frame (there are no END_FRAME records - the last record of a frame ends it, and
 * is drawn like the others); a record the stitcher did not hand over is not drawn and ends no row */
enum { P1_ASM_PER_FRAME = 1470 };
template <> __device__ inline Rec classify<sdv_pcm1_asm_line_rec>(const sdv_pcm1_asm_line_rec *recs, uint32_t i, uint32_t n, int kind)
{
    Rec c; c.end = c.drawn = c.adv = false; c.part = 0;
    if (i >= n) return c;
    c.end = (i + 1u) % (uint32_t)P1_ASM_PER_FRAME == 0u;
    c.drawn = (recs[i].flags & SDV_P1S_SKIP) == 0;
    const uint32_t part = recs[i].line_part;
    c.adv = c.drawn && part == 2;                       /* renderpcm.cpp:733-739 */
    c.part = part > 2 ? 0u : part;                      /* :646-648 */
    return c;
}

template <class R> __device__ inline void count_body(const VisArgs &a, uint32_t chunk, int lane)
{
    const Rec c = classify((const R *)a.recs, chunk * 64u + (uint32_t)lane, a.n_recs, a.kind);
    const uint64_t em = __ballot(c.end), am = __ballot(c.adv);
    if (lane == 0) a.cnt[chunk] = (uint32_t)__popcll(em) | ((uint32_t)__popcll(am) << 16);
}

template <class R> __device__ inline void index_body(const VisArgs &a, uint32_t chunk, int lane)
{
    const uint32_t i = chunk * 64u + (uint32_t)lane;
    const Rec c = classify((const R *)a.recs, i, a.n_recs, a.kind);
    const uint64_t em = __ballot(c.end), am = __ballot(c.adv);
    const uint32_t f = a.base_end[chunk] + (uint32_t)__popcll(em & lanemask_lt(lane)), adv = a.base_adv[chunk] + (uint32_t)__popcll(am & lanemask_lt(lane));
    if (c.end && f + 1 < a.n_frames) a.frame_adv0[f + 1] = adv + (c.adv ? 1u : 0u);       /* (a record that ends its frame and a row: PCM-1 sub-lines) */
    if (i == 0) a.frame_adv0[0] = 0;
}

/* ---- the lines ------------------------------------------------------------------------------------------------------------- */
/* What the pixels of one record depend on: the bits of the line MSB first (up to 128) and a handful of colours. */
struct Look {
    uint32_t hi_h, hi_l, lo_h, lo_l;    /* bits 0..63 in hi, 64..127 in lo, bit b at position 63 - (b % 64) */
    uint32_t c0, c1;                    /* a data bit that is 0 / 1 */
    uint32_t p0, p1;                    /* the same under the Bit Picker's mark (PCM-1, PCM-16x0) */
    uint32_t aux;                       /* STC-007: bit 0 = CRC valid, bit 1 = CRC valid or markers, bit 2 = markers; PCM-16x0: the control bit's colour */
    uint32_t pick;                      /* picked_bits_left | picked_bits_right << 8 */
};
__device__ inline uint64_t push(uint64_t acc, uint32_t word, int bits) { return (acc << bits) | (uint64_t)(word & ((1u << bits) - 1u)); }

__device__ inline Look look_of(const sdv_line_rec &r)     /* renderpcm.cpp:939-1169 */
{
    Look k; k.p0 = k.p1 = k.pick = 0;
    const bool filler = r.service_type == SDV_SRV_FILLER;   /* a cleared line (stc007line.cpp:69-98): silent words, the read CRC the inverse of CRC_SILENT, no flag set */
    uint64_t hi = 0, lo = 0;
    if (!filler) {
#pragma unroll
        for (int w = 0; w < 4; w++) hi = push(hi, r.words[w], 14);
        hi = push(hi, (uint32_t)r.words[4] >> 6, 8);                           /* 4 x 14 + 8 = 64 */
        lo = push(lo, r.words[4], 6);
#pragma unroll
        for (int w = 5; w < 8; w++) lo = push(lo, r.words[w], 14);
        lo = push(lo, r.words[8], 16);                                         /* 6 + 3 x 14 + 16 = 64 */
    } else lo = (uint16_t)~0xA96Au;
    const bool crc = !filler && (r.flags & SDV_LF_CRC_VALID), forced = !filler && (r.flags & SDV_LF_FORCED_BAD);
    const bool markers = !filler && r.mark_st_stage == 4 /* MARK_ST_BOT_2 */ && r.mark_ed_stage == 3 /* MARK_ED_LEN_OK */;
    const bool wcrc = !filler && (r.word_state & SDV_WS_WORD_CRC), wvalid = !filler && (r.word_state & SDV_WS_WORD_VALID);
    if (forced) { k.c0 = B0_MGN; k.c1 = B1_MGN; }
    else if (wcrc) { k.c0 = B0_GRY; k.c1 = B1_GRY; }
    else if (wvalid) { k.c0 = B0_GRN; k.c1 = B1_GRN; }
    else if (markers) { k.c0 = B0_YEL; k.c1 = B1_YEL; }
    else { k.c0 = B0_RED; k.c1 = B1_RED; }
    k.aux = (crc ? 1u : 0u) | ((crc || markers) ? 2u : 0u) | (markers ? 4u : 0u);
    k.hi_h = (uint32_t)(hi >> 32); k.hi_l = (uint32_t)hi; k.lo_h = (uint32_t)(lo >> 32); k.lo_l = (uint32_t)lo;
    return k;
}
/* the colours of a PCM-1 line / PCM-16x0 sub-line (renderpcm.cpp:536-607, 803-876) */
__device__ inline void colours(Look &k, bool crc, bool forced, bool bw)
{
    if (crc) { k.c0 = PX_BLK; k.c1 = B1_GRY; k.p0 = B0_BLU; k.p1 = B1_BLU; }
    else if (forced) { k.c0 = k.p0 = B0_MGN; k.c1 = k.p1 = B1_MGN; }
    else if (bw) { k.c0 = k.p0 = B0_YEL; k.c1 = k.p1 = B1_YEL; }
    else { k.c0 = k.p0 = B0_RED; k.c1 = k.p1 = B1_RED; }
}
__device__ inline Look look_of(const sdv_pcm1_bin_rec &r)   /* :489-624 */
{
    Look k; k.aux = 0;
    const bool filler = r.service_type == SDV_SRV_FILLER;   /* pcm1line.cpp:57-77 */
    uint64_t hi = 0, lo = 0;                                /* 94 bits: 64 + 30 */
    uint16_t w[7];
#pragma unroll
    for (int i = 0; i < 7; i++) w[i] = filler ? (i < 6 ? (uint16_t)(1u << 12) : (uint16_t)~0xECBFu) : r.words[i];
#pragma unroll
    for (int i = 0; i < 4; i++) hi = push(hi, w[i], 13);
    hi = push(hi, (uint32_t)w[4] >> 1, 12);                 /* 4 x 13 + 12 = 64 */
    lo = push(lo, w[4], 1); lo = push(lo, w[5], 13); lo = push(lo, w[6], 16);
    lo <<= 34;                                              /* 30 bits, left-aligned */
    colours(k, !filler && (r.flags & SDV_LF_CRC_VALID), !filler && (r.flags & SDV_LF_FORCED_BAD), !filler && (r.flags & SDV_LF_BW_SET));
    k.pick = filler ? 0u : (uint32_t)r.picked_bits_left | ((uint32_t)r.picked_bits_right << 8);
    k.hi_h = (uint32_t)(hi >> 32); k.hi_l = (uint32_t)hi; k.lo_h = (uint32_t)(lo >> 32); k.lo_l = (uint32_t)lo;
    return k;
}
__device__ inline Look look_of(const sdv_pcm16x0_bin_rec &r)    /* :743-936 */
{
    Look k;
    const bool filler = r.service_type == SDV_SRV_FILLER;   /* pcm16x0subline.cpp:63-86: silent words, control bit set */
    uint64_t hi = 0;
    if (!filler) {
#pragma unroll
        for (int i = 0; i < 4; i++) hi = push(hi, r.words[i], 16);
    } else hi = (uint16_t)~0x0E10u;
    const bool crc = !filler && (r.flags & SDV_LF_CRC_VALID), coords = !filler && (r.flags & SDV_LF_COORDS_SET), control = filler || r.control_bit != 0;
    colours(k, crc, !filler && (r.flags & SDV_LF_FORCED_BAD), !filler && (r.flags & SDV_LF_BW_SET));
    k.aux = (!coords || !crc) ? (control ? B1_RED : B0_RED) : (control ? B1_GRY : (uint32_t)PX_BLK);        /* :878-912 */
    k.pick = filler ? 0u : (uint32_t)r.picked_bits_left | ((uint32_t)r.picked_bits_right << 8);
    k.hi_h = (uint32_t)(hi >> 32); k.hi_l = (uint32_t)hi; k.lo_h = k.lo_l = 0;
    return k;
}
__device__ inline Look look_of(const sdv_pcm1_asm_line_rec &r)  /* :626-741 */
{
    Look k; k.aux = 0;
    uint64_t hi = 0;
    hi = push(hi, r.words[0], 13); hi = push(hi, r.words[1], 13);
    hi <<= 38;                                              /* 26 bits, left-aligned */
    colours(k, (r.flags & SDV_P1S_CRC_VALID) != 0, false, (r.flags & SDV_P1S_BW_SET) != 0);
    k.pick = r.line_part == 0 ? (uint32_t)r.picked_bits_left : 0u;          /* the mark belongs to the left part of a line (:686, :711) */
    k.hi_h = (uint32_t)(hi >> 32); k.hi_l = (uint32_t)hi; k.lo_h = k.lo_l = 0;
    return k;
}
__device__ inline uint32_t bit_of(const Look &k, uint32_t b)
{
    const uint32_t w = b < 32 ? k.hi_h : b < 64 ? k.hi_l : b < 96 ? k.lo_h : k.lo_l;
    return (w >> (31u - (b & 31u))) & 1u;
}
/* the pixel at x of the line / of the sub-line's part of the row (x counted from the part's first pixel) */
template <class R> __device__ inline uint32_t pixel(const Look &k, uint32_t x, uint32_t part);
template <> __device__ inline uint32_t pixel<sdv_line_rec>(const Look &k, uint32_t x, uint32_t)
{
    const uint32_t b = x / 5u;
    if (b < 4) return (k.aux & 2u) ? ((b & 1u) ? (uint32_t)B0_GRY : (uint32_t)B1_GRY) : (uint32_t)PX_BLK;       /* START: 1010 */
    if (b < 132) return bit_of(k, b - 4) ? k.c1 : k.c0;
    if (b == 132) return (k.aux & 2u) ? (uint32_t)B0_GRY : (uint32_t)PX_BLK;                                    /* STOP: 0 1111 */
    return (k.aux & 1u) ? (uint32_t)B1_MARK : (k.aux & 4u) ? (uint32_t)B1_GRY : (uint32_t)PX_BLK;
}
template <> __device__ inline uint32_t pixel<sdv_pcm1_bin_rec>(const Look &k, uint32_t x, uint32_t)
{
    const uint32_t b = x / 8u, pl = k.pick & 0xFFu, pr = k.pick >> 8;
    const bool picked = b < pl || (int)b > 94 - (int)pr - 1;
    const uint32_t v = bit_of(k, b);
    return picked ? (v ? k.p1 : k.p0) : (v ? k.c1 : k.c0);
}
template <> __device__ inline uint32_t pixel<sdv_pcm16x0_bin_rec>(const Look &k, uint32_t x, uint32_t part)
{
    const uint32_t b = x / 4u, pl = k.pick & 0xFFu, pr = k.pick >> 8;
    if (b >= 64) return k.aux;                      /* the control bit behind the middle sub-line */
    const bool picked = b < pl || (int)b > 64 - (int)pr - 1;
    const uint32_t v = bit_of(k, b);
    return picked ? (v ? k.p1 : k.p0) : (v ? k.c1 : k.c0);
}

template <> __device__ inline uint32_t pixel<sdv_pcm1_asm_line_rec>(const Look &k, uint32_t x, uint32_t)
{
    const uint32_t b = x / 8u;
    const uint32_t v = bit_of(k, b);
    return b < k.pick ? (v ? k.p1 : k.p0) : (v ? k.c1 : k.c0);
}

template <class R> __device__ inline void draw_body(const VisArgs &a, uint32_t chunk, int lane)
{
    const R *recs = (const R *)a.recs;
    const Geometry g = geometry(a.kind);
    const uint32_t i = chunk * 64u + (uint32_t)lane;
    const Rec c = classify(recs, i, a.n_recs, a.kind);
    const uint64_t em = __ballot(c.end), am = __ballot(c.adv);
    const uint32_t f = a.base_end[chunk] + (uint32_t)__popcll(em & lanemask_lt(lane)), adv = a.base_adv[chunk] + (uint32_t)__popcll(am & lanemask_lt(lane));
    bool live = c.drawn && f < a.n_frames;          /* records behind the last END_FRAME belong to a frame that has not ended */
    uint32_t row = 0;
    if (live) { row = adv - a.frame_adv0[f]; live = row < g.h; }       /* "line overflow": the canvas is full (renderpcm.cpp:955-961) */
    if (live && g.cells_per_row > 1 && !c.adv && !(c.end && c.drawn)) {
        /* a left or middle sub-line is drawn over by the next one of its kind that comes before the row ends */
        for (uint32_t j = i + 1; j < a.n_recs; j++) {
            const Rec n = classify(recs, j, a.n_recs, a.kind);
            if (n.drawn && n.part == c.part) { live = false; break; }       /* (n.adv: part 2, never c.part here) */
            if (n.end || n.adv) break;
        }
    }
    Look k = Look();
    if (live) {
        k = look_of(recs[i]);
        const uint32_t cell = row * g.cells_per_row + c.part;
        atomicOr(&a.wmask[(size_t)f * a.wmask_stride + cell / 32u], 1u << (cell % 32u));
    }
    const uint64_t lm = __ballot(live);
    for (int j = 0; j < 64; j++) {
        if (!((lm >> j) & 1ull)) continue;
        Look kj;
        kj.hi_h = (uint32_t)__shfl((int)k.hi_h, j); kj.hi_l = (uint32_t)__shfl((int)k.hi_l, j); kj.lo_h = (uint32_t)__shfl((int)k.lo_h, j); kj.lo_l = (uint32_t)__shfl((int)k.lo_l, j);
        kj.c0 = (uint32_t)__shfl((int)k.c0, j); kj.c1 = (uint32_t)__shfl((int)k.c1, j); kj.p0 = (uint32_t)__shfl((int)k.p0, j); kj.p1 = (uint32_t)__shfl((int)k.p1, j);
        kj.aux = (uint32_t)__shfl((int)k.aux, j); kj.pick = (uint32_t)__shfl((int)k.pick, j);
        const uint32_t fj = (uint32_t)__shfl((int)f, j), rowj = (uint32_t)__shfl((int)row, j), partj = (uint32_t)__shfl((int)c.part, j);
        uint32_t x0, x1;
        cell_span(a.kind, partj, x0, x1);
        uint32_t *dst = a.out + ((size_t)fj * g.h + rowj) * g.w + x0;
        for (uint32_t x = (uint32_t)lane; x < x1 - x0; x += 64u) dst[x] = pixel<R>(kj, x, partj);
    }
}

/* ---- what a frame did not draw stays as it was ---------------------------------------------------------------------------------- */
/* A cell a frame did not draw shows what the last frame before it that drew the cell put there (or the canvas kept from before the
 * call): always pixels the line pass wrote, never pixels of this pass, so every (frame, cell) is on its own once it knows that frame.
 * last_body finds it for every frame of a cell (one wave per cell walking the frames' bitmaps, 64 frames a step); fill_body copies,
 * one wave per frame and 64 cells - a frame's rows are neighbours in memory, the source rows are few and stay in L2. */
__device__ inline void last_body(const VisArgs &a, uint32_t cell, int lane)
{
    int32_t last = -1;              /* -1: the canvas kept from before */
    for (uint32_t f0 = 0; f0 < a.n_frames; f0 += 64u) {
        const uint32_t f = f0 + (uint32_t)lane;
        const uint64_t wm = __ballot(f < a.n_frames && ((a.wmask[(size_t)f * a.wmask_stride + cell / 32u] >> (cell % 32u)) & 1u));
        const uint64_t upto = wm & (lanemask_lt(lane) | (1ull << lane));
        if (f < a.n_frames) a.last_drawn[(size_t)f * a.n_cells + cell] = upto ? (int32_t)f0 + 63 - (int32_t)__clzll((long long)upto) : last;
        if (wm) last = (int32_t)f0 + 63 - (int32_t)__clzll((long long)wm);
    }
}
__device__ inline void fill_body(const VisArgs &a, uint32_t block, int lane)
{
    const Geometry g = geometry(a.kind);
    const uint32_t chunks = (a.n_cells + 63u) / 64u, f = block / chunks, cell = (block % chunks) * 64u + (uint32_t)lane;
    const int32_t from = cell < a.n_cells ? a.last_drawn[(size_t)f * a.n_cells + cell] : (int32_t)f;
    uint64_t todo = __ballot(from != (int32_t)f);
    const size_t frame_px = (size_t)g.w * g.h;
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int32_t fj = __shfl(from, j);
        const uint32_t cj = (block % chunks) * 64u + (uint32_t)j, row = cj / g.cells_per_row;
        uint32_t x0, x1;
        cell_span(a.kind, cj % g.cells_per_row, x0, x1);
        const size_t at = (size_t)row * g.w + x0;
        const uint32_t *src = fj < 0 ? a.canvas + at : a.out + (size_t)fj * frame_px + at;
        uint32_t *dst = a.out + (size_t)f * frame_px + at;
        for (uint32_t x = (uint32_t)lane; x < x1 - x0; x += 64u) dst[x] = src[x];
    }
}

/* ---- the data blocks window: renderNewBlock(STC007DataBlock), renderpcm.cpp:1770-2051 --------------------------------------------- */
/* Frame f's blocks are blocks[frame_ofs[f] .. frame_ofs[f + 1]); block i of a frame is row i (rows past the canvas are dropped, :1781-1787).
 * A wave takes 64 rows of one frame: a lane packs its block (six 16-bit samples, the per-word marks, seven block-level bits), the wave
 * then writes one row after the other, 64 consecutive pixels per store - 6 pixels per bit, 6 status bits + 96 sample bits + 7 tail bits. */
struct BlkArgs {
    const sdv_block_rec *blocks; const uint32_t *frame_ofs; uint32_t n_frames; int kind;
    uint32_t *out; uint32_t *wmask; uint32_t wmask_stride;
    int m2;                         /* the blocks hold M2 samples (SDV_VIS_M2_SAMPLES) */
};
enum { BK_FIX_P = 1u << 18, BK_FIX_Q = 1u << 19, BK_BROKEN = 1u << 20, BK_VALID = 1u << 21, BK_CWD_AUDIO = 1u << 22, BK_SILENT = 1u << 23, BK_SEAM = 1u << 24 };
__device__ inline int16_t blk_sample(const sdv_block_rec &b, int w, bool m2)      /* STC007DataBlock::getSample, stc007datablock.cpp:507-562 */
{
    const uint32_t v = b.words[w];
    if (!m2) return b.resolution == SDV_RES_16BIT ? (int16_t)v : (int16_t)(v << 2);
    if ((v & (1u << 13)) == 0) return (int16_t)(uint16_t)(v << 3);      /* M2, higher range: the value times eight */
    const uint32_t low = v & ~(1u << 13);                               /* lower range: bit 12 is the sign, extended */
    return (int16_t)(uint16_t)((v & (1u << 12)) ? (low | 0xE000u) : low);
}
__device__ inline bool blk_near_silence(const sdv_block_rec &b, int w, bool m2)      /* isNearSilence, stc007datablock.cpp:417-446 */
{
    const int v = blk_sample(b, w, m2), lim = (b.resolution == SDV_RES_16BIT || m2) ? 4 : 16;
    return v < lim && v >= -lim;
}
__device__ inline uint32_t block_pixel(uint32_t s01, uint32_t s23, uint32_t s45, uint32_t fl, uint32_t x)
{
    const uint32_t b = x / 6u;
    if (b < 6u) {                                           /* the status bar :1793-1861 */
        if (b == 0) return (fl & BK_FIX_P) ? (uint32_t)B1_GRN : (uint32_t)PX_BLK;
        if (b == 1) return (fl & BK_FIX_Q) ? (uint32_t)B1_YEL : (uint32_t)PX_BLK;
        if (b == 2) return (fl & BK_CWD_AUDIO) ? ((fl & BK_VALID) ? (uint32_t)B1_BLU : (uint32_t)B0_BLU) : (uint32_t)PX_BLK;
        if (b == 3) return (fl & BK_VALID) ? (uint32_t)PX_BLK : (uint32_t)B1_RED;
        if (b == 5) return (fl & BK_SILENT) ? (uint32_t)LIM_MARK : (uint32_t)LIM_OK;
        return PX_BLK;
    }
    if (b < 102u) {                                         /* the samples :1862-1993 */
        const uint32_t w = (b - 6u) / 16u, bit = 15u - ((b - 6u) % 16u);
        const uint32_t pair = w < 2 ? s01 : w < 4 ? s23 : s45, v = (w & 1u) ? pair >> 16 : pair & 0xFFFFu;
        const bool one = (v >> bit) & 1u, crc = (fl >> w) & 1u, cwd = (fl >> (6u + w)) & 1u, wv = (fl >> (12u + w)) & 1u;
        if (fl & BK_BROKEN) return crc ? (one ? (uint32_t)B1_GRY : (uint32_t)PX_BLK) : (one ? (uint32_t)B1_MGN : (uint32_t)B0_MGN);
        if (fl & (BK_FIX_Q | BK_FIX_P)) {
            if (cwd) return one ? (uint32_t)B1_BLU : (uint32_t)B0_BLU;
            if (!crc) return (fl & BK_FIX_Q) ? (one ? (uint32_t)B1_YEL : (uint32_t)B0_YEL) : (one ? (uint32_t)B1_GRN : (uint32_t)B0_GRN);
            return one ? (uint32_t)B1_GRY : (uint32_t)PX_BLK;
        }
        if (cwd) return one ? (uint32_t)B1_BLU : (uint32_t)B0_BLU;
        if (!wv) return one ? (uint32_t)B1_RED : (uint32_t)B0_RED;
        return one ? (uint32_t)B1_GRY : (uint32_t)PX_BLK;
    }
    const uint32_t i = b - 102u;                            /* seam, emphasis (never set for STC-007), BROKEN :1995-2045 */
    if (i == 0) return (fl & BK_SEAM) ? (uint32_t)LIM_MARK : (uint32_t)LIM_OK;
    if (i == 4 || i == 5) return (fl & BK_BROKEN) ? (uint32_t)B1_MGN : (uint32_t)PX_BLK;
    return PX_BLK;
}
__device__ inline void draw_blocks_body(const BlkArgs &a, uint32_t block, int lane)
{
    const Geometry g = geometry(a.kind);
    const uint32_t chunks = (g.h + 63u) / 64u, f = block / chunks, c = block % chunks;
    const uint32_t lo = a.frame_ofs[f], n = a.frame_ofs[f + 1] - lo, rows = n < g.h ? n : g.h;
    if (lane < 2) {                                         /* the frame's rows 64 c .. 64 c + 63 as two words of the drawn-cells bitmap */
        const uint32_t first = 64u * c + 32u * (uint32_t)lane, word = 2u * c + (uint32_t)lane;
        if (word < a.wmask_stride) a.wmask[(size_t)f * a.wmask_stride + word] = rows <= first ? 0u : rows - first >= 32u ? 0xFFFFFFFFu : (1u << (rows - first)) - 1u;
    }
    const uint32_t row = 64u * c + (uint32_t)lane;
    const bool live = row < rows;
    uint32_t s01 = 0, s23 = 0, s45 = 0, fl = 0;
    if (live) {
        const sdv_block_rec b = a.blocks[lo + row];
        const bool m2 = a.m2 != 0;
        s01 = (uint32_t)(uint16_t)blk_sample(b, 0, m2) | ((uint32_t)(uint16_t)blk_sample(b, 1, m2) << 16);
        s23 = (uint32_t)(uint16_t)blk_sample(b, 2, m2) | ((uint32_t)(uint16_t)blk_sample(b, 3, m2) << 16);
        s45 = (uint32_t)(uint16_t)blk_sample(b, 4, m2) | ((uint32_t)(uint16_t)blk_sample(b, 5, m2) << 16);
        const bool silent = (blk_near_silence(b, 0, m2) || blk_near_silence(b, 2, m2) || blk_near_silence(b, 4, m2)) &&
                            (blk_near_silence(b, 1, m2) || blk_near_silence(b, 3, m2) || blk_near_silence(b, 5, m2));
        fl = (uint32_t)(b.line_crc & 0x3F) | ((uint32_t)(b.cwd_fixed & 0x3F) << 6) | ((uint32_t)(b.word_valid & 0x3F) << 12) |
             (b.audio_state == SDV_AUD_FIX_P ? BK_FIX_P : 0u) | (b.audio_state == SDV_AUD_FIX_Q ? BK_FIX_Q : 0u) | (b.audio_state == SDV_AUD_BROKEN ? BK_BROKEN : 0u) |
             ((b.word_valid & 0x3F) == 0x3F ? BK_VALID : 0u) | ((b.cwd_fixed & 0x3F) ? BK_CWD_AUDIO : 0u) | (silent ? BK_SILENT : 0u) | (b.w_line[0] > b.w_line[7] ? BK_SEAM : 0u);
    }
    const uint64_t lm = __ballot(live);
    for (int j = 0; j < 64; j++) {
        if (!((lm >> j) & 1ull)) continue;
        const uint32_t j01 = (uint32_t)__shfl((int)s01, j), j23 = (uint32_t)__shfl((int)s23, j), j45 = (uint32_t)__shfl((int)s45, j), jfl = (uint32_t)__shfl((int)fl, j);
        uint32_t *dst = a.out + ((size_t)f * g.h + 64u * c + (uint32_t)j) * g.w;
        for (uint32_t x = (uint32_t)lane; x < g.w; x += 64u) dst[x] = block_pixel(j01, j23, j45, jfl, x);
    }
}

/* ---- the data blocks window of PCM-1: renderNewBlock(PCM1DataBlock), renderpcm.cpp:1171-1400 -------------------------------------------------
 * A block is 23 rows of eight words: row r of a frame belongs to block r / 23 and shows its words 8 (r % 23) .. + 7.  A wave takes 64 rows; a lane
 * packs what its row shows (eight samples, eight word marks, the block's marks), the wave then writes row after row. */
struct P1BlkArgs { const sdv_pcm1_block_rec *blocks; const uint32_t *frame_ofs; uint32_t n_frames; int kind; uint32_t *out; uint32_t *wmask; uint32_t wmask_stride; };
enum { P1BK_ROWS = 23, P1BK_INVALID = 1u << 24, P1BK_SILENT = 1u << 25, P1BK_ODD = 1u << 26, P1BK_EMPH = 1u << 27 };
__device__ inline int16_t p1blk_sample(uint32_t w)        /* PCM1DataBlock::getSample, pcm1datablock.cpp:309-348 */
{
    if ((w & (1u << 12)) == 0) return (int16_t)(uint16_t)(w << 4);
    const uint32_t v = (w & ~(1u << 12)) << 2;
    return (int16_t)(uint16_t)((w & (1u << 11)) ? (v | 0xC000u) : v);
}
/* fl: bits 0..7 word there (not past a short block's end), 8..15 word invalid, 16..23 word mark (2 bits per pair would do; per word: picked sample),
 * pk: bits 0..7 hasPickedWord per word */
__device__ inline uint32_t p1_block_pixel(const uint32_t (&s)[4], uint32_t fl, uint32_t pk, uint32_t x)
{
    const uint32_t b = x / 6u;
    if (b < 11u) {                                          /* the status bar :1213-1283 */
        if (b < 8u) {
            const bool there = (fl >> b) & 1u;
            if ((b & 1u) == 0) return !there ? (uint32_t)PX_BLK : ((fl >> (16u + b)) & 1u) ? (uint32_t)B1_BLU : ((pk >> b) & 1u) ? (uint32_t)B0_BLU : (uint32_t)PX_BLK;
            return (there && ((fl >> (8u + b)) & 1u)) ? (uint32_t)B1_YEL : (uint32_t)PX_BLK;
        }
        if (b == 8u) return (fl & P1BK_INVALID) ? (uint32_t)B1_RED : (uint32_t)PX_BLK;
        if (b == 10u) return (fl & P1BK_SILENT) ? (uint32_t)LIM_MARK : (uint32_t)LIM_OK;
        return PX_BLK;
    }
    if (b < 139u) {                                         /* eight samples :1286-1352 */
        const uint32_t w = (b - 11u) / 16u, bit = 15u - ((b - 11u) % 16u);
        if (!((fl >> w) & 1u)) return PX_BLK;
        const uint32_t v = (w & 1u) ? s[w >> 1] >> 16 : s[w >> 1] & 0xFFFFu;
        const bool one = (v >> bit) & 1u;
        if ((fl >> (8u + w)) & 1u) return one ? (uint32_t)B1_RED : (uint32_t)B0_RED;
        if ((fl >> (16u + w)) & 1u) return one ? (uint32_t)B1_BLU : (uint32_t)B0_BLU;
        return one ? (uint32_t)B1_GRY : (uint32_t)PX_BLK;
    }
    const uint32_t i = b - 139u;                            /* the block's parity in the field, emphasis :1355-1387 */
    if (i == 0) return (fl & P1BK_ODD) ? (uint32_t)LIM_MARK : (uint32_t)LIM_OK;
    if (i == 2) return (fl & P1BK_EMPH) ? (uint32_t)B0_GRN : (uint32_t)PX_BLK;
    return PX_BLK;
}
__device__ inline void draw_p1_blocks_body(const P1BlkArgs &a, uint32_t block, int lane)
{
    const Geometry g = geometry(a.kind);
    const uint32_t chunks = (g.h + 63u) / 64u, f = block / chunks, c = block % chunks;
    const uint32_t lo = a.frame_ofs[f], nb = a.frame_ofs[f + 1] - lo, n = nb * (uint32_t)P1BK_ROWS;
    /* a block that starts past the canvas is dropped whole (:1183-1189); one that starts on it draws all of its rows that fit */
    const uint32_t rows = n < g.h ? n : g.h;
    if (lane < 2) {
        const uint32_t first = 64u * c + 32u * (uint32_t)lane, word = 2u * c + (uint32_t)lane;
        if (word < a.wmask_stride) a.wmask[(size_t)f * a.wmask_stride + word] = rows <= first ? 0u : rows - first >= 32u ? 0xFFFFFFFFu : (1u << (rows - first)) - 1u;
    }
    const uint32_t row = 64u * c + (uint32_t)lane;
    const bool live = row < rows;
    uint32_t s[4] = { 0, 0, 0, 0 }, fl = 0, pk = 0;
    if (live) {
        const sdv_pcm1_block_rec &b = a.blocks[lo + row / (uint32_t)P1BK_ROWS];
        const uint32_t w0 = 8u * (row % (uint32_t)P1BK_ROWS), count = (b.flags & SDV_P1B_SHORT) ? 182u : 184u;
        bool invalid = false, silent = true;
        for (uint32_t w = 0; w < count; w++) {              /* isBlockValid (:190-199), isAlmostSilent (:229-244) */
            invalid = invalid || !(b.word_flags[w] & SDV_P1W_CRC_OK);
            const int v = p1blk_sample(b.words[w]);
            silent = silent && v < 16 && v >= -16;
        }
#pragma unroll
        for (uint32_t k = 0; k < 8; k++) {
            const uint32_t w = w0 + k;
            const uint32_t v = (uint32_t)(uint16_t)p1blk_sample(b.words[w]);
            s[k >> 1] |= (k & 1u) ? v << 16 : v;
            if (w < count) fl |= 1u << k;
            if (!(b.word_flags[w] & SDV_P1W_CRC_OK)) fl |= 1u << (8u + k);
            if (b.word_flags[w] & SDV_P1W_PICKED_LEFT) fl |= 1u << (16u + k);
            if (b.word_flags[w] & SDV_P1W_PICKED_WORD) pk |= 1u << k;
        }
        fl |= (invalid ? P1BK_INVALID : 0u) | (silent ? P1BK_SILENT : 0u) | ((b.interleave_num & 1u) ? P1BK_ODD : 0u) | ((b.flags & SDV_P1B_EMPHASIS) ? P1BK_EMPH : 0u);
    }
    const uint64_t lm = __ballot(live);
    for (int j = 0; j < 64; j++) {
        if (!((lm >> j) & 1ull)) continue;
        uint32_t sj[4];
#pragma unroll
        for (int q = 0; q < 4; q++) sj[q] = (uint32_t)__shfl((int)s[q], j);
        const uint32_t jfl = (uint32_t)__shfl((int)fl, j), jpk = (uint32_t)__shfl((int)pk, j);
        uint32_t *dst = a.out + ((size_t)f * g.h + 64u * c + (uint32_t)j) * g.w;
        for (uint32_t x = (uint32_t)lane; x < g.w; x += 64u) dst[x] = p1_block_pixel(sj, jfl, jpk, x);
    }
}

/* ---- the data blocks window of PCM-16x0: renderNewBlock(PCM16X0DataBlock), renderpcm.cpp:1403-1768: one row per block ------------------------------ */
struct P16BlkArgs { const sdv_pcm16x0_block_rec *blocks; const uint32_t *frame_ofs; uint32_t n_frames; int kind; uint32_t *out; uint32_t *wmask; uint32_t wmask_stride; };
/* fl: bits 0..5 crc of sample 2 blk + word, 6..11 valid, 12..17 picked sample, 18..23 state of the sample's sub-block (2 bits each: 18 + 2 blk),
 * 24 block valid, 25 some sub-block BROKEN, 26 almost silent, 27 EI, 28 emphasis; st: the nine status colours are worked out per pixel from
 * pk: bit 0 picked left (sub-block 1), bits 1..3 picked CRC by sub-block */
enum { P16K_VALID = 1u << 24, P16K_BROKEN = 1u << 25, P16K_SILENT = 1u << 26, P16K_EI = 1u << 27, P16K_EMPH = 1u << 28 };
__device__ inline uint32_t p16_block_pixel(const uint32_t (&s)[3], uint32_t fl, uint32_t pk, uint32_t x)
{
    const uint32_t b = x / 6u;
    if (b < 9u) {                                           /* the status bar :1423-1511 */
        if (b < 6u) {
            const uint32_t sb = b / 2u, st = (fl >> (18u + 2u * sb)) & 3u;
            if ((b & 1u) == 0) return (b == 0 && (pk & 1u)) ? (uint32_t)B1_BLU : ((pk >> (1u + sb)) & 1u) ? (uint32_t)B0_BLU : (uint32_t)PX_BLK;
            return st == 1u ? (uint32_t)B1_GRN : (uint32_t)PX_BLK;
        }
        if (b == 6u) return (fl & P16K_VALID) ? (uint32_t)PX_BLK : (uint32_t)B1_RED;
        if (b == 8u) return (fl & P16K_SILENT) ? (uint32_t)LIM_MARK : (uint32_t)LIM_OK;
        return PX_BLK;
    }
    if (b < 105u) {                                         /* the six samples :1513-1712 */
        const uint32_t w = (b - 9u) / 16u, bit = 15u - ((b - 9u) % 16u), blk = w >> 1;
        const uint32_t v = (w & 1u) ? s[blk] >> 16 : s[blk] & 0xFFFFu;
        const bool one = (v >> bit) & 1u, crc = (fl >> w) & 1u, wv = (fl >> (6u + w)) & 1u, picked = (fl >> (12u + w)) & 1u;
        const uint32_t st = (fl >> (18u + 2u * blk)) & 3u;
        const uint32_t plain = one ? (uint32_t)B1_GRY : (uint32_t)PX_BLK, blue = one ? (uint32_t)B1_BLU : (uint32_t)B0_BLU;
        if (!(fl & P16K_VALID)) {
            if (st != 2u) return !crc ? (one ? (uint32_t)B1_RED : (uint32_t)B0_RED) : picked ? blue : plain;
            return !wv ? (one ? (uint32_t)B1_MGN : (uint32_t)B0_MGN) : plain;
        }
        if (st == 1u) return !crc ? (one ? (uint32_t)B1_GRN : (uint32_t)B0_GRN) : picked ? blue : plain;
        if (!wv) return one ? (uint32_t)B1_RED : (uint32_t)B0_RED;
        return picked ? blue : plain;
    }
    const uint32_t i = b - 105u;                            /* format, emphasis, BROKEN :1714-1757 */
    if (i == 0) return LIM_OK;
    if (i == 2) return (fl & P16K_EI) ? (uint32_t)B1_BLU : (uint32_t)PX_BLK;
    if (i == 3) return (fl & P16K_EMPH) ? (uint32_t)B0_GRN : (uint32_t)PX_BLK;
    if (i == 5 || i == 6) return (fl & P16K_BROKEN) ? (uint32_t)B1_MGN : (uint32_t)PX_BLK;
    return PX_BLK;
}
__device__ inline void draw_p16_blocks_body(const P16BlkArgs &a, uint32_t block, int lane)
{
    const Geometry g = geometry(a.kind);
    const uint32_t chunks = (g.h + 63u) / 64u, f = block / chunks, c = block % chunks;
    const uint32_t lo = a.frame_ofs[f], n = a.frame_ofs[f + 1] - lo, rows = n < g.h ? n : g.h;
    if (lane < 2) {
        const uint32_t first = 64u * c + 32u * (uint32_t)lane, word = 2u * c + (uint32_t)lane;
        if (word < a.wmask_stride) a.wmask[(size_t)f * a.wmask_stride + word] = rows <= first ? 0u : rows - first >= 32u ? 0xFFFFFFFFu : (1u << (rows - first)) - 1u;
    }
    const uint32_t row = 64u * c + (uint32_t)lane;
    const bool live = row < rows;
    uint32_t s[3] = { 0, 0, 0 }, fl = 0, pk = 0;
    if (live) {
        const sdv_pcm16x0_block_rec b = a.blocks[lo + row];
        const bool even = (b.flags & SDV_P16B_EVEN_ORDER) != 0;
        bool valid = true, broken = false, silent = false;
#pragma unroll
        for (uint32_t i = 0; i < 3; i++) {
            const bool l_first = ((i & 1u) != 0) != even;                       /* getWordToLine, pcm16x0datablock.cpp:1029-1155 */
            const uint32_t ll = l_first ? 0u : 2u, lr = l_first ? 2u : 0u;
            const uint32_t wl = b.words[i][ll], wr = b.words[i][lr];
            s[i] = wl | (wr << 16);
            const uint32_t cl = (b.word_crc >> (3u * i + ll)) & 1u, cr = (b.word_crc >> (3u * i + lr)) & 1u;
            const uint32_t vl = (b.word_valid >> (3u * i + ll)) & 1u, vr = (b.word_valid >> (3u * i + lr)) & 1u;
            fl |= (cl << (2u * i)) | (cr << (2u * i + 1u)) | (vl << (6u + 2u * i)) | (vr << (7u + 2u * i)) | ((uint32_t)(b.audio_state[i] & 3u) << (18u + 2u * i));
            if (i == 0) fl |= (((uint32_t)b.picked_left >> ll) & 1u) << 12u | (((uint32_t)b.picked_left >> lr) & 1u) << 13u;
            valid = valid && vl && vr;
            broken = broken || b.audio_state[i] == 2;
            const int sl = (int16_t)(uint16_t)wl, sr = (int16_t)(uint16_t)wr;
            silent = silent || (sl < 4 && sl >= -4 && sr < 4 && sr >= -4);
            const bool pcrc = (b.picked_crc & 1u) || (b.picked_crc & 4u) || ((b.picked_crc & 2u) && b.audio_state[i] == 1);
            pk |= pcrc ? 1u << (1u + i) : 0u;
        }
        pk |= ((b.picked_left & 1u) || (b.picked_left & 4u)) ? 1u : 0u;
        fl |= (valid ? P16K_VALID : 0u) | (broken ? P16K_BROKEN : 0u) | (silent ? P16K_SILENT : 0u) | ((b.flags & SDV_P16B_EI_FORMAT) ? P16K_EI : 0u) | ((b.flags & SDV_P16B_EMPHASIS) ? P16K_EMPH : 0u);
    }
    const uint64_t lm = __ballot(live);
    for (int j = 0; j < 64; j++) {
        if (!((lm >> j) & 1ull)) continue;
        uint32_t sj[3];
#pragma unroll
        for (int q = 0; q < 3; q++) sj[q] = (uint32_t)__shfl((int)s[q], j);
        const uint32_t jfl = (uint32_t)__shfl((int)fl, j), jpk = (uint32_t)__shfl((int)pk, j);
        uint32_t *dst = a.out + ((size_t)f * g.h + 64u * c + (uint32_t)j) * g.w;
        for (uint32_t x = (uint32_t)lane; x < g.w; x += 64u) dst[x] = p16_block_pixel(sj, jfl, jpk, x);
    }
}

/* ---- the assembled-lines window: renderNewLine(STC007Line) on the stitcher's lines (sdv_asm_line_rec): every word in the colour of its own state ---- */
struct AsmArgs {
    const sdv_asm_line_rec *lines; const uint32_t *frame_ofs; uint32_t n_frames; int kind;
    uint32_t *out; uint32_t *wmask; uint32_t wmask_stride;
};
__device__ inline uint32_t asm_pixel(const Look &k, uint32_t st, uint32_t x)   /* st: word_crc_ok | word_valid << 9 | SDV_AL_* << 18 */
{
    const uint32_t b = x / 5u, fl = st >> 18;
    const bool crc = (fl & SDV_AL_CRC_VALID) != 0, markers = (fl & SDV_AL_MARKERS) != 0, forced = (fl & SDV_AL_FORCED_BAD) != 0;
    if (b < 4) return (crc || markers) ? ((b & 1u) ? (uint32_t)B0_GRY : (uint32_t)B1_GRY) : (uint32_t)PX_BLK;
    if (b < 132) {
        const uint32_t d = b - 4u, w = d < 112u ? d / 14u : 8u;
        const bool one = bit_of(k, d) != 0, wc = (st >> w) & 1u, wv = (st >> (9u + w)) & 1u;
        if (forced) return one ? (uint32_t)B1_MGN : (uint32_t)B0_MGN;
        if (wc) return one ? (uint32_t)B1_GRY : (uint32_t)B0_GRY;
        if (wv) return one ? (uint32_t)B1_GRN : (uint32_t)B0_GRN;
        if (markers) return one ? (uint32_t)B1_YEL : (uint32_t)B0_YEL;
        return one ? (uint32_t)B1_RED : (uint32_t)B0_RED;
    }
    if (b == 132) return (crc || markers) ? (uint32_t)B0_GRY : (uint32_t)PX_BLK;
    return crc ? (uint32_t)B1_MARK : markers ? (uint32_t)B1_GRY : (uint32_t)PX_BLK;
}
__device__ inline void draw_asm_body(const AsmArgs &a, uint32_t block, int lane)
{
    const Geometry g = geometry(a.kind);
    const uint32_t chunks = (g.h + 63u) / 64u, f = block / chunks, c = block % chunks;
    const uint32_t lo = a.frame_ofs[f], n = a.frame_ofs[f + 1] - lo, rows = n < g.h ? n : g.h;
    if (lane < 2) {
        const uint32_t first = 64u * c + 32u * (uint32_t)lane, word = 2u * c + (uint32_t)lane;
        if (word < a.wmask_stride) a.wmask[(size_t)f * a.wmask_stride + word] = rows <= first ? 0u : rows - first >= 32u ? 0xFFFFFFFFu : (1u << (rows - first)) - 1u;
    }
    const uint32_t row = 64u * c + (uint32_t)lane;
    const bool live = row < rows;
    Look k = Look(); uint32_t st = 0;
    if (live) {
        const sdv_asm_line_rec r = a.lines[lo + row];
        uint64_t hi = 0, lw = 0;
#pragma unroll
        for (int w = 0; w < 4; w++) hi = push(hi, r.words[w], 14);
        hi = push(hi, (uint32_t)r.words[4] >> 6, 8);
        lw = push(lw, r.words[4], 6);
#pragma unroll
        for (int w = 5; w < 8; w++) lw = push(lw, r.words[w], 14);
        lw = push(lw, r.words[8], 16);
        k.hi_h = (uint32_t)(hi >> 32); k.hi_l = (uint32_t)hi; k.lo_h = (uint32_t)(lw >> 32); k.lo_l = (uint32_t)lw;
        st = (uint32_t)(r.word_crc_ok & 0x1FF) | ((uint32_t)(r.word_valid & 0x1FF) << 9) | ((uint32_t)r.flags << 18);
    }
    const uint64_t lm = __ballot(live);
    for (int j = 0; j < 64; j++) {
        if (!((lm >> j) & 1ull)) continue;
        Look kj = Look();
        kj.hi_h = (uint32_t)__shfl((int)k.hi_h, j); kj.hi_l = (uint32_t)__shfl((int)k.hi_l, j); kj.lo_h = (uint32_t)__shfl((int)k.lo_h, j); kj.lo_l = (uint32_t)__shfl((int)k.lo_l, j);
        const uint32_t sj = (uint32_t)__shfl((int)st, j);
        uint32_t *dst = a.out + ((size_t)f * g.h + 64u * c + (uint32_t)j) * g.w;
        for (uint32_t x = (uint32_t)lane; x < g.w; x += 64u) dst[x] = asm_pixel(kj, sj, x);
    }
}

struct BlankArgs { uint32_t *canvas; uint32_t n_px; };
__device__ inline void blank_body(const BlankArgs &a, uint32_t i) { if (i < a.n_px) a.canvas[i] = BLANK; }
} // namespace sdvvis

#define SDV_VIS_KERNELS(R, tag) \
    __global__ void __launch_bounds__(64) sdv_k_vis_count_##tag(sdvvis::VisArgs a) { sdvvis::count_body<R>(a, blockIdx.x, (int)threadIdx.x); } \
    __global__ void __launch_bounds__(64) sdv_k_vis_index_##tag(sdvvis::VisArgs a) { sdvvis::index_body<R>(a, blockIdx.x, (int)threadIdx.x); } \
    __global__ void __launch_bounds__(64) sdv_k_vis_draw_##tag(sdvvis::VisArgs a) { sdvvis::draw_body<R>(a, blockIdx.x, (int)threadIdx.x); }
SDV_VIS_KERNELS(sdv_line_rec, stc007)
SDV_VIS_KERNELS(sdv_pcm1_bin_rec, pcm1)
SDV_VIS_KERNELS(sdv_pcm16x0_bin_rec, pcm16x0)
SDV_VIS_KERNELS(sdv_pcm1_asm_line_rec, p1asm)
__global__ void __launch_bounds__(64) sdv_k_vis_draw_blocks(sdvvis::BlkArgs a) { sdvvis::draw_blocks_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_vis_draw_p1_blocks(sdvvis::P1BlkArgs a) { sdvvis::draw_p1_blocks_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_vis_draw_p16_blocks(sdvvis::P16BlkArgs a) { sdvvis::draw_p16_blocks_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_vis_draw_asm(sdvvis::AsmArgs a) { sdvvis::draw_asm_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_vis_last(sdvvis::VisArgs a) { sdvvis::last_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_vis_fill(sdvvis::VisArgs a) { sdvvis::fill_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_vis_blank(sdvvis::BlankArgs a) { sdvvis::blank_body(a, blockIdx.x * 64u + threadIdx.x); }
