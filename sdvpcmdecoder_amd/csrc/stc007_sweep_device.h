/*
 * stc007_sweep_device.h - the reference-level sweep of the STC-007 binarizer (Binarizer::calcRefLevelBySweep / sweepRefLevel,
 * binarizer.cpp:3551-4120) as kernels of its own.  Included by stc007_device.h.
 *
 * A line that reads with no level it inherits makes the reference evaluate the line at every level between black and white (<= 234
 * levels, each a marker search at 24 hysteresis depths plus the hysteresis x shift ladder of reads), vote over the CRCs that came out
 * and pick a level.  That is 99.9 % of what such a line costs, and it depends on the chain of the decode only through four numbers:
 * the black and white levels the line measured and the data coordinates the binarizer was preset with.  So the frame kernel does not
 * sweep at all: where it needs the outcome of a sweep it looks it up (SweepMemo, a list per frame), and where there is none it leaves
 * a request, goes on as if the sweep had found nothing and marks its frame unsettled.  The engine then runs all requests of a round
 *   sdv_k_stc007_sweep_levels   one wave per (line, 64 levels): the levels on the lanes
 *   sdv_k_stc007_sweep_pick     one wave per line: the chain through the levels, the vote, the pick
 * and decodes the unsettled frames again - now with the outcomes at hand (engine.inc).
 *
 * What a level costs here (all from the scanline staged in LDS once per wave):
 *   - markers: the comparisons "pixel >= T" of the two marker windows are made once per wave for the <= 87 thresholds T the 64 levels
 *     and their 24 hysteresis depths can ask for, as bit masks (192 pixels each); the two state machines of searchSTC007Markers
 *     (binarizer.cpp:5275-5595) then jump from edge to edge with find-first-set instead of walking pixel by pixel.  The STOP marker
 *     does not depend on the hysteresis depth: searched once per level, and the START searches only where it was found.
 *   - reads: the lane gathers the 128 cell bytes of a pixel-shift stage once (packed in 32 registers), turns them into eight bit planes
 *     (planes_from_cells) and takes every hysteresis depth from those - a comparison with a threshold is a carry through the planes;
 *     automaton and CRC as bit arithmetic on the lane's own masks (solve_automaton_lane).
 *   - a level started from a zero source-CRC word (see sweep_pick_body) needs no second evaluation: its outcome follows from the
 *     first one's.
 */
#ifndef SDV_STC007_SWEEP_DEVICE_H
#define SDV_STC007_SWEEP_DEVICE_H

namespace sdv {

/* ---- requests and outcomes ------------------------------------------------------------------- */
enum { SWEEP_REQUESTED = 0, SWEEP_SETTLED = 1 };
struct SweepMemo {                  /* 32 bytes, at 32-byte strides in a pool that starts 16-byte aligned */
    int32_t frame, next;            /* frame index of the call; the next entry of the same line of that frame, -1 at the end */
    uint16_t row;                   /* row of the frame */
    uint8_t black, white;           /* the key: what the sweep of that line depends on besides its pixels and the settings ... */
    int16_t in_start, in_stop;      /* ... the levels the line measured and the binarizer's preset data coordinates */
    uint8_t state;                  /* SWEEP_* */
    uint8_t span1, span2;           /* SPAN_*: the first pick (among the levels with the most frequent valid CRC); the second one when that found nothing */
    uint8_t ref_level, t_hyst, t_shift;  /* the level picked and what the sweep found there */
    int16_t t_start, t_stop;
    uint8_t _pad[6];
};
struct SweepOutcome { uint8_t span1, span2, ref_level, t_hyst, t_shift; int16_t t_start, t_stop; };

/* what the frame kernel hands down to the line that may need a sweep */
struct SweepHook {
    SweepMemo *memo; int32_t *head; int32_t *count; int32_t cap;
    int32_t frame; uint16_t row;
    int32_t line;                   /* frame * height + row: the line's list head */
    bool pending;                   /* out: a sweep of this frame was asked for, the line went on without it */
    unsigned long long *bw_slot;    /* in: the line's slot of FrameArgs::bw_memo, or NULL */
    bool ladder_failed;             /* in: the frame loop tried the ladder of reads with the tuning the line inherits (fast_line), nothing read */
    bool stop;                      /* out: ... by a line that had no reference level preset: everything behind it hangs on what the sweep finds, the pass over the frame ends here */
    /* sdv_k_stc007_frames_fat only (fat_sweep, below): the waves beside the frame's own settle what it asks for while it waits */
    struct FatLds *fat;             /* or NULL */
    const FrameArgs *fa;            /* the launch parameters (what a sweep needs of them: pixels, geometry, settings) */
    SweepEnt *fat_levels;           /* [256] the workgroup's room for what the levels leave */
};

/* an entry as eight words (two 16-byte loads), the same in every lane */
struct SweepMemoWords { uint32_t w[8]; };
static_assert(sizeof(SweepMemo) == 32, "SweepMemo is two 16-byte loads");
__device__ inline SweepMemoWords sweep_memo_load(const SweepMemo *m)
{
    SweepMemoWords r;
    const uint4 a = ((const uint4 *)m)[0], b = ((const uint4 *)m)[1];
    r.w[0] = uniu(a.x); r.w[1] = uniu(a.y); r.w[2] = uniu(a.z); r.w[3] = uniu(a.w); r.w[4] = uniu(b.x); r.w[5] = uniu(b.y); r.w[6] = uniu(b.z); r.w[7] = uniu(b.w);
    return r;
}
/* words 2, 3 of an entry: row | black << 16 | white << 24, in_start | in_stop << 16 */
__device__ __forceinline__ uint32_t sweep_key_lo(uint16_t row, uint8_t black, uint8_t white) { return (uint32_t)row | ((uint32_t)black << 16) | ((uint32_t)white << 24); }
__device__ __forceinline__ uint32_t sweep_key_hi(const Coords &c) { return (uint32_t)(uint16_t)c.start | ((uint32_t)(uint16_t)c.stop << 16); }
/* wave-uniform; all lanes return the same */
__device__ inline bool sweep_lookup(const SweepHook &h, uint8_t black, uint8_t white, const Coords &in_coord, SweepOutcome &o)
{
    const uint32_t k0 = sweep_key_lo(h.row, black, white), k1 = sweep_key_hi(in_coord);
    int idx = uni(h.head[h.line]);
    for (int guard = 0; idx >= 0 && idx < h.cap && guard < (1 << 20); guard++) {
        const SweepMemoWords m = sweep_memo_load(&h.memo[idx]);
        if (m.w[2] == k0 && m.w[3] == k1) {
            if ((m.w[4] & 0xFFu) != SWEEP_SETTLED) return false;
            o.span1 = (uint8_t)(m.w[4] >> 8); o.span2 = (uint8_t)(m.w[4] >> 16); o.ref_level = (uint8_t)(m.w[4] >> 24);
            o.t_hyst = (uint8_t)m.w[5]; o.t_shift = (uint8_t)(m.w[5] >> 8); o.t_start = (int16_t)(m.w[5] >> 16); o.t_stop = (int16_t)(m.w[6] & 0xFFFFu);
            return true;
        }
        idx = (int)m.w[1];
    }
    return false;
}
/* (a request that is on the list already - left by an earlier pass over this frame and not settled, which does not happen, or by this
 * pass - is not entered twice) */
/* returns the entry (wave-uniform), -1 when the pool is full */
__device__ inline int sweep_request(SweepHook &h, uint8_t black, uint8_t white, const Coords &in_coord)
{
    h.pending = true;
    const uint32_t k0 = sweep_key_lo(h.row, black, white), k1 = sweep_key_hi(in_coord);
    int idx = uni(h.head[h.line]);
    for (int guard = 0; idx >= 0 && idx < h.cap && guard < (1 << 20); guard++) {
        const SweepMemoWords m = sweep_memo_load(&h.memo[idx]);
        if (m.w[2] == k0 && m.w[3] == k1) return idx;
        idx = (int)m.w[1];
    }
    int slot = 0;
    if (lane_id() == 0) {
        slot = atomicAdd(h.count, 1);
        if (slot < h.cap) {         /* a full pool: the count tells the engine, which makes room and decodes the frame again */
            SweepMemo m;
            m.frame = h.frame; m.next = h.head[h.line]; m.row = h.row; m.black = black; m.white = white; m.in_start = in_coord.start; m.in_stop = in_coord.stop;
            m.state = SWEEP_REQUESTED; m.span1 = m.span2 = SPAN_NOT_FOUND; m.ref_level = m.t_hyst = m.t_shift = 0; m.t_start = m.t_stop = 0;
            for (int i = 0; i < 6; i++) m._pad[i] = 0;
            h.memo[slot] = m;
            h.head[h.line] = slot;
        }
    }
    SDV_BLOCK_SYNC();
    slot = uni(slot);
    return slot < h.cap ? slot : -1;
}

/* ---- launch parameters of the two sweep kernels ------------------------------------------------ */
enum { SWL_HASM = 1,                /* SweepEnt::pad of a level record: the level found both markers */
       SWL_ZERO = 2 };              /* ... its evaluation left 0x0000 as the source CRC word */
struct SweepArgs {
    const uint8_t *luma; size_t frame_stride, row_stride; int width;
    uint8_t doubled, mode;
    sdv_bin_preset preset;
    SweepMemo *memo;                /* requests [first, first + count) are worked on */
    int first, count;
    SweepEnt *levels;               /* [count][256]: what sdv_k_stc007_sweep_levels found per level, read by sdv_k_stc007_sweep_pick */
};
struct SweepLds {
    alignas(16) uint8_t px[SDV_MAX_WIDTH];
    uint32_t wplanes[2][8][SWEEP_WINDOW_MAX / 32];      /* the two marker windows as bit planes: [window][bit of the pixel value][32 pixels] */
    uint64_t g_start[88][3];        /* [T - t_lo]: pixel p of the START window >= T, bit p */
    uint64_t g_stop[88][3];         /* ... pixel scan_end - i of the STOP window >= T, bit i */
};

/* the line geometry Binarizer::processLine derives from the line length (binarizer.cpp:600-641) */
__device__ inline void bin_line_geometry(Bin &b, const sdv_bin_preset &ps, int width, bool doubled)
{
    b.line_length = (uint16_t)width; b.vl_doubled = doubled;
    b.scan_start = 0; b.scan_end = (uint16_t)(b.line_length - 1);
    b.mark_start_max = (uint16_t)(b.line_length * ps.mark_max_dist);
    b.mark_start_max = b.mark_start_max / 100;
    b.mark_end_min = (uint16_t)(b.scan_end - b.mark_start_max);
    b.mark_start_max = (uint16_t)(b.scan_start + b.mark_start_max);
    uint32_t tmp_calc = (uint32_t)b.line_length * 128u;
    tmp_calc = tmp_calc / BITS_IN_LINE;
    b.estimated_ppb = (uint16_t)((tmp_calc + 64) / 128);
}

/* ---- markers from threshold masks (m192_first_set / m192_first_clear: stc007_device.h) ---------------------- */
/* START marker "1010" (binarizer.cpp:5310-5405) on L = pixel >= low threshold, H = pixel >= reference level, pixels [0, n_px):
 * returns whether the marker was found; st1e = where the data starts (end of the first "1") */
__device__ inline bool start_marker_masks(const uint64_t *L, const uint64_t *H, int n_px, int mark_start_max, int ppb, int &st1e)
{
    for (int p = 0;;) {
        const int q1 = m192_first_set(L, p);
        if (q1 >= n_px || q1 > mark_start_max) return false;
        const int q2 = m192_first_clear(L, q1 + 1);
        if (q2 >= n_px) return false;
        const int q3 = m192_first_set(H, q2 + 1);
        if (q3 >= n_px) return false;
        if ((q3 - q2) > ppb * 2 || (q3 - q2) < ppb / 2) { p = q3 + 1; continue; }
        const int q4 = m192_first_clear(H, q3 + 1);
        if (q4 >= n_px) return false;
        if ((q4 - q3) > ppb * 2 || (q4 - q3) < ppb / 2) { p = q4 + 1; continue; }
        st1e = q2;
        return true;
    }
}
/* STOP marker "01111" from the right (binarizer.cpp:5410-5455) on R = pixel >= reference level, bit i = pixel scan_end - i, bits [0, n_px) */
__device__ inline bool stop_marker_masks(const uint64_t *R, int n_px, int i_max /* scan_end - mark_end_min */, int scan_end, int ppb, int &ed_start)
{
    for (int i = 0;;) {
        const int j1 = m192_first_set(R, i);
        if (j1 >= n_px || j1 > i_max) return false;
        const int j2 = m192_first_clear(R, j1 + 1);
        if (j2 >= n_px) return false;
        if ((j2 - j1) >= ppb * 2 && (j2 - j1) <= ppb * 5) { ed_start = scan_end - j2 + 1; return true; }
        i = j2 + 1;
    }
}

/* ---- reads --------------------------------------------------------------------------------------- */
/* (SDV_OPAQUE: stc007_device.h) */
struct LaneRead { bool any_fill, valid; uint8_t hyst, shift; uint16_t crc, w8; };
/* the cell bytes of pixel-shift stage `stage` (PCMLine::getVideoPixeBylCalc, pcmline.cpp:249-311), four to a word, cell 0 in the low byte of row[0] */
__device__ __forceinline__ void gather_cells(const uint8_t *px, uint32_t (&row)[32], uint32_t psm, uint32_t hpsm, int pso, int stage, int px_lo, int px_hi)
{
    /* the cell centres in 1/128 pixel, the stage's pixel shift folded in.  (Kept opaque: the centres do not depend on the stage, and the compiler
     * would otherwise work out all 128 of them once, ahead of the loop over the stages, and hold them in 128 registers.) */
    int acc = (int)((uint32_t)(BITS_START - 1) * psm + hpsm) + 128 * (shift_of_stage(stage) + pso);
    SDV_OPAQUE(acc);
#pragma unroll
    for (int q = 0; q < 32; q++) {
        uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int vp = acc >> 7; acc += (int)psm;
            vp = vp < px_lo ? px_lo : (vp > px_hi ? px_hi : vp);
            x |= (uint32_t)px[vp] << (8 * k);
        }
        SDV_OPAQUE(x);              /* (... and would keep the 128 bytes in a register each instead of four to a word) */
        row[q] = x;
    }
}
/* ---- the cells as bit planes ------------------------------------------------------------------------------
 * A level reads its line at up to 15 (depth, stage) steps, and a step is two comparisons of the 128 cell bytes with a threshold: byte by byte that is
 * seven instructions per cell, three quarters of everything the levels kernel does.  Turned into eight bit planes once per stage (bit j of plane k of
 * group q = bit k of cell 32 q + j) a comparison is the carry out of cell + (256 - T), rippled through the planes for 32 cells at a time: 24
 * instructions per group instead of 224. */
#ifdef SDV_EMU
static inline uint32_t sdv_perm(uint32_t hi, uint32_t lo, uint32_t sel)
{
    uint32_t r = 0;
    for (int i = 0; i < 4; i++) { const uint32_t s = (sel >> (8 * i)) & 0xFF; const uint32_t b = s < 4 ? (lo >> (8 * s)) & 0xFF : (hi >> (8 * (s - 4))) & 0xFF; r |= b << (8 * i); }
    return r;
}
#else
#define sdv_perm(hi, lo, sel) __builtin_amdgcn_perm(hi, lo, sel)
#endif
/* 8 cells (cx: cells 0..3, cell 0 in the low byte; cy: cells 4..7) -> lo: byte k = bit k of the eight cells (cell i at bit i), k = 0..3; hi: k = 4..7
 * (the 8 x 8 bit transpose of Hacker's Delight 7-3, which holds row 0 in the high byte and column 0 in the high bit: both orders turned round) */
__device__ __forceinline__ void transpose8(uint32_t cx, uint32_t cy, uint32_t &lo, uint32_t &hi)
{
    uint32_t x = cy, y = cx, t;
    t = (x ^ (x >> 7)) & 0x00AA00AAu; x = x ^ t ^ (t << 7);
    t = (y ^ (y >> 7)) & 0x00AA00AAu; y = y ^ t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCCu; x = x ^ t ^ (t << 14);
    t = (y ^ (y >> 14)) & 0x0000CCCCu; y = y ^ t ^ (t << 14);
    t = (x & 0xF0F0F0F0u) | ((y >> 4) & 0x0F0F0F0Fu);
    y = ((x << 4) & 0xF0F0F0F0u) | (y & 0x0F0F0F0Fu);
    lo = y; hi = t;
}
/* a 4 x 4 byte matrix turned over: a, b, c, d become byte 0, 1, 2, 3 of the four (a's in the low byte) */
__device__ __forceinline__ void bytes4x4(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d)
{
    const uint32_t t0 = sdv_perm(b, a, 0x05010400u), t1 = sdv_perm(b, a, 0x07030602u), u0 = sdv_perm(d, c, 0x05010400u), u1 = sdv_perm(d, c, 0x07030602u);
    a = sdv_perm(u0, t0, 0x05040100u); b = sdv_perm(u0, t0, 0x07060302u); c = sdv_perm(u1, t1, 0x05040100u); d = sdv_perm(u1, t1, 0x07060302u);
}
/* in place: row[8 q + k] becomes plane k of the cells 32 q .. 32 q + 31 */
__device__ __forceinline__ void planes_from_cells(uint32_t (&row)[32])
{
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int b = 0; b < 4; b++) transpose8(row[8 * q + 2 * b], row[8 * q + 2 * b + 1], lo[b], hi[b]);
        bytes4x4(lo[0], lo[1], lo[2], lo[3]);
        bytes4x4(hi[0], hi[1], hi[2], hi[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) { row[8 * q + k] = lo[k]; row[8 * q + 4 + k] = hi[k]; }
    }
}
/* cell >= T for the 32 cells of group q (cell j at bit j); T = 0 .. 256 */
__device__ __forceinline__ uint32_t planes_ge(const uint32_t (&row)[32], int q, int T)
{
    const uint32_t nt = ~(uint32_t)T;
    uint32_t c = 0xFFFFFFFFu;             /* cell + (255 - T) + 1 */
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t p = row[8 * q + k];
        const uint32_t m = (uint32_t)0 - ((nt >> k) & 1u);
        c = (m & (p | c)) | (~m & (p & c));
    }
    return T > 255 ? 0u : c;
}
/* Binarizer::fillSTC007 (binarizer.cpp:7322-7445) from the planes of the gathered cells: the calculated CRC and the CRC word as read.  (The two-level
 * automaton's masks: cell > low, cell >= high.) */
__device__ __forceinline__ void fill_from_planes(const uint32_t (&row)[32], int low, int high, uint16_t &crc_out, uint16_t &w8_out)
{
    uint32_t am[4], bm[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { am[q] = planes_ge(row, q, low + 1); bm[q] = planes_ge(row, q, high); }
    uint64_t s_lo, s_hi;
    solve_automaton_lane((uint64_t)am[0] | ((uint64_t)am[1] << 32), (uint64_t)am[2] | ((uint64_t)am[3] << 32),
                         (uint64_t)bm[0] | ((uint64_t)bm[1] << 32), (uint64_t)bm[2] | ((uint64_t)bm[3] << 32), s_lo, s_hi);
    uint32_t crc = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) crc |= (uint32_t)((__popcll(s_lo & c_crc.klo[j]) + __popcll(s_hi & c_crc.khi[j])) & 1) << j;
    crc_out = (uint16_t)(crc ^ c_crc.init);
    w8_out = rev16((uint32_t)((s_hi >> 48) & 0xFFFF));
}
/* Binarizer::readPCMdata (binarizer.cpp:7695-8055) of one lane's trial line: the first (depth, stage) in the reference's order - depth outer,
 * stage inner, ended by the first depth whose levels leave (black, white) - that reads with a valid CRC; when there is none, what the closing read at
 * (0, 0) leaves.  The stages are taken one after the other (their cells are gathered once) and per stage only the depths that would still come
 * first; the result is the same pair. */
__device__ inline LaneRead sweep_read_lane(const uint8_t *px, int width, int16_t c_start, int16_t c_stop, int level, int black, int white, int hyst_lim, int shift_lim)
{
    LaneRead r; r.any_fill = false; r.valid = false; r.hyst = r.shift = 0; r.crc = 0; r.w8 = 0xFFFF;    /* nothing read: calc_crc 0, setInvalidCRC */
    uint32_t psm = (uint32_t)((int)c_stop - (int)c_start);
    psm = (psm * 128u + BITS_BETWEEN / 2) / BITS_BETWEEN;
    const uint32_t hpsm = (psm + 1) / 2;
    int h_ok = 0;
    for (int h = 0; h <= hyst_lim; h++) {
        if (get_low_level((uint8_t)level, (uint8_t)h) <= black || get_high_level((uint8_t)level, (uint8_t)h) >= white) break;
        h_ok++;
    }
    if (h_ok == 0) return r;
    r.any_fill = true;
    int best_h = h_ok;
    uint16_t crc00 = 0, w800 = 0;
#pragma unroll 1
    for (int s = 0; s <= shift_lim && best_h > 0; s++) {
        uint32_t row[32];
        gather_cells(px, row, psm, hpsm, c_start, s, 0, width - 2);
        planes_from_cells(row);
#pragma unroll 1
        for (int h = 0; h < best_h; h++) {
            uint16_t crc, w8;
            fill_from_planes(row, get_low_level((uint8_t)level, (uint8_t)h), get_high_level((uint8_t)level, (uint8_t)h), crc, w8);
            if (s == 0 && h == 0) { crc00 = crc; w800 = w8; }
            if (crc == w8) { best_h = h; r.valid = true; r.hyst = (uint8_t)h; r.shift = (uint8_t)s; r.crc = crc; r.w8 = w8; break; }
        }
    }
    if (!r.valid) { r.crc = crc00; r.w8 = w800; }
    return r;
}

/* ---- kernel 1: the levels ---------------------------------------------------------------------------- */
/* Which way a level goes through the body of sweepRefLevel's loop (binarizer.cpp:3626-3816) is the same for all levels of a line */
enum { SWEEP_KASE_MARKERS_OR_PRESET = 0,    /* preset coordinates and "good without markers": markers found -> read there, else read at the preset coordinates */
       SWEEP_KASE_MARKERS_ONLY = 1,         /* read only where markers are found */
       SWEEP_KASE_FORCED = 2 };             /* coordinates forced by the settings */
__device__ inline int sweep_kase(const sdv_bin_preset &ps, const Coords &forced, const Coords &in_coord)
{
    if (coords_valid(forced)) return SWEEP_KASE_FORCED;
    return (coords_valid(in_coord) && ps.en_good_no_marker) ? SWEEP_KASE_MARKERS_OR_PRESET : SWEEP_KASE_MARKERS_ONLY;
}
__device__ inline void sweep_level_range(const sdv_bin_preset &ps, uint8_t black, uint8_t white, int &low_lvl, int &high_lvl)
{
    low_lvl = (uint8_t)(black + 1); high_lvl = (uint8_t)(white - 1);
    if (ps.min_ref_lvl > low_lvl) low_lvl = ps.min_ref_lvl;
    if (ps.max_ref_lvl < high_lvl) high_lvl = ps.max_ref_lvl;
}
__device__ inline void sweep_stage_row(SweepLds &lds, const SweepArgs &a, const SweepMemo &m)
{
    const uint8_t *row = a.luma + (size_t)uni(m.frame) * a.frame_stride + (size_t)uni(m.row) * a.row_stride;
    const int lane = lane_id();
    SDV_BLOCK_SYNC();
    if (((((uintptr_t)row) | (uintptr_t)a.width) & 15) == 0) { for (int i = lane; i < (a.width >> 4); i += 64) ((uint4 *)lds.px)[i] = ((const uint4 *)row)[i]; }
    else for (int i = lane; i < a.width; i += 64) lds.px[i] = row[i];
    SDV_BLOCK_SYNC();
}

__device__ inline void sweep_levels_body(const SweepArgs &a, SweepLds &lds, int req, int group)
{
    const SweepMemo &m = a.memo[a.first + req];
    const sdv_bin_preset &ps = a.preset;
    const int lane = lane_id();
    Bin b;
    bin_set_mode(b, a.mode);
    bin_line_geometry(b, ps, a.width, a.doubled != 0);
    b.in_coord.start = (int16_t)uni(m.in_start); b.in_coord.stop = (int16_t)uni(m.in_stop); b.in_coord.doubled = a.doubled != 0;
    const int hyst_lim = b.in_max_hyst > HYST_DEPTH_MAX ? HYST_DEPTH_MAX : b.in_max_hyst, shift_lim = b.in_max_shift > SHIFT_STAGES_MAX ? SHIFT_STAGES_MAX : b.in_max_shift;
    Coords forced; calc_forced_coords(b, ps, forced);
    const int kase = sweep_kase(ps, forced, b.in_coord);
    int low_lvl, high_lvl;
    sweep_level_range(ps, (uint8_t)uni(m.black), (uint8_t)uni(m.white), low_lvl, high_lvl);
    const int base = high_lvl - 64 * group;                /* the lanes take levels base, base - 1, ... */
    if (base < low_lvl) return;
    sweep_stage_row(lds, a, m);
    const int lvl = base - lane;
    const bool active = lvl >= low_lvl;
    const int ppb = b.estimated_ppb, scan_end = b.scan_end;

    bool hasm = false; int16_t c_start = NO_COORD_LEFT, c_stop = NO_COORD_RIGHT;      /* markers found; the coordinates a trial line has after the search */
    if (kase != SWEEP_KASE_FORCED) {
        /* the windows of the two marker searches (binarizer.cpp:5300-5308, :5408-5418) */
        int n_start = b.mark_start_max + ppb * 5; n_start &= 0xFFFF; if (n_start > b.line_length) n_start = b.line_length;
        const int end_limit = b.mark_end_min > ppb * 6 ? b.mark_end_min - ppb * 6 : 0;
        const int n_stop = scan_end - end_limit, i_max = scan_end - (int)b.mark_end_min;
        if (n_start <= SWEEP_WINDOW_MAX && n_stop <= SWEEP_WINDOW_MAX) {
            /* thresholds the 64 levels can ask for: the level itself and down to 23 below it (never below 1) */
            int t_lo = base - 63 - 23; if (t_lo < 1) t_lo = 1;
            const int n_t = base - t_lo + 1;                /* <= 87 */
            /* ... from the bit planes of the two windows (see planes_from_cells): lanes 0 .. 47 turn eight pixels each, then a lane per threshold takes
             * the carry through the eight planes, 32 pixels at a time (pixel by pixel this was a sixth of the kernel) */
            if (lane < 2 * (SWEEP_WINDOW_MAX / 8)) {
                const int win = lane / (SWEEP_WINDOW_MAX / 8), blk = lane % (SWEEP_WINDOW_MAX / 8);
                uint32_t cx = 0, cy = 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int p = 8 * blk + i;
                    const uint32_t v = win == 0 ? (p < n_start ? (uint32_t)lds.px[p] : 0u) : (p < n_stop ? (uint32_t)lds.px[scan_end - p] : 0u);
                    if (i < 4) cx |= v << (8 * i); else cy |= v << (8 * (i - 4));
                }
                uint32_t lo, hi;
                transpose8(cx, cy, lo, hi);
                uint8_t *pl = (uint8_t *)lds.wplanes[win];
#pragma unroll
                for (int k = 0; k < 4; k++) { pl[k * (SWEEP_WINDOW_MAX / 8) + blk] = (uint8_t)(lo >> (8 * k)); pl[(4 + k) * (SWEEP_WINDOW_MAX / 8) + blk] = (uint8_t)(hi >> (8 * k)); }
            }
            SDV_BLOCK_SYNC();
            for (int w = 0; w < SWEEP_WINDOW_MAX / 32; w++) {
                uint32_t ps_[8], pe_[8];
#pragma unroll
                for (int k = 0; k < 8; k++) { ps_[k] = lds.wplanes[0][k][w]; pe_[k] = lds.wplanes[1][k][w]; }
                for (int tb = 0; tb < n_t; tb += 64) {
                    const int ti = tb + lane;
                    const uint32_t nt = ~(uint32_t)(t_lo + ti);         /* (t_lo + ti <= 255 + 23: a threshold above 255 is asked of no lane - ti >= n_t) */
                    uint32_t cs = 0xFFFFFFFFu, ce = 0xFFFFFFFFu;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const uint32_t m = (uint32_t)0 - ((nt >> k) & 1u);
                        cs = (m & (ps_[k] | cs)) | (~m & (ps_[k] & cs));
                        ce = (m & (pe_[k] | ce)) | (~m & (pe_[k] & ce));
                    }
                    if (ti < n_t) { ((uint32_t *)lds.g_start[ti])[w] = cs; ((uint32_t *)lds.g_stop[ti])[w] = ce; }
                }
            }
            SDV_BLOCK_SYNC();
            if (active) {
                const uint64_t *H = lds.g_start[lvl - t_lo];
                int ed_start = 0;
                if (stop_marker_masks(lds.g_stop[lvl - t_lo], n_stop, i_max, scan_end, ppb, ed_start)) {
                    int best = 0x7FFFFFFF;
#pragma unroll 1
                    for (int h = 0; h < 24; h++) {
                        int bin_low = get_low_level((uint8_t)lvl, (uint8_t)h);
                        if (bin_low < ps.min_ref_lvl) bin_low = ps.min_ref_lvl;
                        int st1e = 0;
                        if (start_marker_masks(lds.g_start[bin_low - t_lo], H, n_start, b.mark_start_max, ppb, st1e) && st1e < best) best = st1e;
                    }
                    if (best != 0x7FFFFFFF) { hasm = true; if ((int16_t)ed_start > (int16_t)best) { c_start = (int16_t)best; c_stop = (int16_t)ed_start; } }
                }
            }
        } else if (active) {
            /* marker windows wider than the masks (a very wide frame, or settings that search half the line): pixel by pixel */
            int best = 0x7FFFFFFF, ed_start = 0;
            for (int h = 0; h < 24; h++) {
                const Markers mk = search_markers_px(b, ps, lds.px, (uint8_t)lvl, (uint8_t)h);
                if (mk.has_start && mk.ed_stage == MARK_ED_LEN_OK && (int)mk.st1e < best) { best = mk.st1e; ed_start = mk.ed_start; }
            }
            if (best != 0x7FFFFFFF) { hasm = true; if ((int16_t)ed_start > (int16_t)best) { c_start = (int16_t)best; c_stop = (int16_t)ed_start; } }
        }
    }
    if (!active) return;
    /* the one read of the level, and what the level leaves */
    SweepEnt e; e.result = REF_NO_PCM; e.hyst = 0; e.shift = 0; e.pad = 0; e.crc = 0; e.start = 0; e.stop = 0; e.pad2 = 0;
    bool read = false; int16_t r_start = c_start, r_stop = c_stop; bool coords_are_set = hasm;
    if (kase == SWEEP_KASE_FORCED) { r_start = forced.start; r_stop = forced.stop; coords_are_set = true; read = true; }
    else if (hasm) read = true;
    else if (kase == SWEEP_KASE_MARKERS_OR_PRESET) { r_start = b.in_coord.start; r_stop = b.in_coord.stop; read = true; }
    if (read) {
        const LaneRead r = sweep_read_lane(lds.px, a.width, r_start, r_stop, lvl, low_lvl, high_lvl, hyst_lim, shift_lim);
        const bool cv = pod_coords_valid(r_start, r_stop);
        if (r.valid && cv) e.result = REF_CRC_OK;
        else if (coords_are_set) e.result = REF_BAD_CRC;    /* (a read at the preset coordinates that fails leaves the line without coordinates: nothing) */
        if (e.result != REF_NO_PCM) { e.start = r_start; e.stop = r_stop; e.hyst = r.hyst; e.shift = r.shift; e.crc = r.crc; }
        if (r.w8 == 0) e.pad |= SWL_ZERO;
    }
    if (hasm) e.pad |= SWL_HASM;
    a.levels[(size_t)req * 256u + (size_t)lvl] = e;
}

/* ---- kernel 2: chain, vote, pick -------------------------------------------------------------------- */
__device__ inline M256 m256_ballot(const bool (&p)[4])
{
    M256 m;
    m.w[0] = __ballot(p[0]); m.w[1] = __ballot(p[1]); m.w[2] = __ballot(p[2]); m.w[3] = __ballot(p[3]);
    return m;
}
__device__ inline bool m256_mine(const M256 &m, int g) { return ((m.w[g] >> lane_id()) & 1ull) != 0ull; }
/* the value level i's lane holds (v[g] belongs to level 64 g + lane) */
__device__ inline uint32_t level_read(const uint32_t (&v)[4], int i)
{
    uint32_t x = 0;
#pragma unroll
    for (int g = 0; g < 4; g++) x = (g == (i >> 6)) ? v[g] : x;
    return uniu((uint32_t)__shfl((int)x, i & 63));
}

__device__ inline void sweep_pick_body(const SweepArgs &a, int req)
{
    SweepMemo &m = a.memo[a.first + req];
    const sdv_bin_preset &ps = a.preset;
    const int lane = lane_id();
    Bin b;
    bin_set_mode(b, a.mode);
    bin_line_geometry(b, ps, a.width, a.doubled != 0);
    b.in_coord.start = (int16_t)uni(m.in_start); b.in_coord.stop = (int16_t)uni(m.in_stop); b.in_coord.doubled = a.doubled != 0;
    Coords forced; calc_forced_coords(b, ps, forced);
    const int kase = sweep_kase(ps, forced, b.in_coord);
    const uint8_t black = (uint8_t)uni(m.black), white = (uint8_t)uni(m.white);
    int low_lvl, high_lvl;
    sweep_level_range(ps, black, white, low_lvl, high_lvl);
    const int blk1 = (uint8_t)(black + 1), wht1 = (uint8_t)(white - 1);

    SweepEnt e[4]; bool in[4], zero[4], hasm[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int lvl = 64 * g + lane;
        in[g] = lvl >= low_lvl && lvl <= high_lvl;
        SweepEnt z; z.result = REF_NO_PCM; z.hyst = z.shift = 0x0f; z.pad = 0; z.crc = 0; z.start = z.stop = 0; z.pad2 = 0;
        e[g] = in[g] ? a.levels[(size_t)req * 256u + (size_t)lvl] : z;
        zero[g] = in[g] && (e[g].pad & SWL_ZERO); hasm[g] = in[g] && (e[g].pad & SWL_HASM);
        if (e[g].result == REF_NO_PCM) { e[g].hyst = e[g].shift = 0x0f; e[g].crc = 0; e[g].start = e[g].stop = 0; }   /* nothing stored: the table's blank */
    }
    /* The chain through the levels.  Level L of the reference starts from whatever source-CRC word level L + 1 left in the shared trial line
     * (PCMLine::clear() through a base pointer does not reset the STC007Line part, binarizer.cpp:3629).  That only matters when the word is 0x0000:
     * then "calculated CRC == source CRC" holds before anything was read.  A level that found markers then takes them without reading - a "valid"
     * line with CRC 0 at depth 0 - and leaves the zero; a level that reads at the preset coordinates reads as ever; where nothing but markers count
     * the level does nothing and leaves the zero.  So what a level yields when it is started from a zero follows from what it yields otherwise. */
    const M256 zm = m256_ballot(zero), hm = m256_ballot(hasm);
    if (m256_any(zm)) {
        M256 from_zero = m256_zero();
        bool z = false;
        for (int lvl = high_lvl; lvl >= low_lvl; lvl--) {
            if (z) m256_set(from_zero, lvl);
            const bool leaves_zero_then = kase == SWEEP_KASE_MARKERS_OR_PRESET ? (m256_test(hm, lvl) || m256_test(zm, lvl)) : true;
            z = z ? leaves_zero_then : m256_test(zm, lvl);
        }
#pragma unroll
        for (int g = 0; g < 4; g++)
            if (m256_mine(from_zero, g)) {
                if (kase == SWEEP_KASE_MARKERS_OR_PRESET) {
                    if (hasm[g]) { e[g].result = pod_coords_valid(e[g].start, e[g].stop) ? REF_CRC_OK : REF_BAD_CRC; e[g].hyst = 0; e[g].shift = 0; e[g].crc = 0; }
                } else { e[g].result = REF_NO_PCM; e[g].hyst = e[g].shift = 0x0f; e[g].crc = 0; e[g].start = e[g].stop = 0; }
            }
    }
    /* the vote (updateCRCStats / findMostFrequentCRC / invalidateNonFrequentCRCs, binarizer.cpp:1789-1982): the CRCs of the levels that read valid, from the top;
     * the most frequent one must beat every other by more than two to one (of the first 31 different ones - the reference's table holds no more) */
    bool ok[4]; uint32_t crcv[4];
#pragma unroll
    for (int g = 0; g < 4; g++) { const int lvl = 64 * g + lane; ok[g] = lvl >= blk1 && lvl <= wht1 && e[g].result == REF_CRC_OK; crcv[g] = e[g].crc; }
    M256 rem = m256_ballot(ok);
    const bool had_any = m256_any(rem);
    int top = 0, second = 0; uint32_t target = 0;
    for (int n = 0; n < MAX_COLL_CRCS - 1 && m256_any(rem); n++) {
        const uint32_t c = level_read(crcv, m256_top_le(rem, 255));
        bool eq[4];
#pragma unroll
        for (int g = 0; g < 4; g++) eq[g] = ok[g] && crcv[g] == c;
        const M256 em = m256_ballot(eq);
        const int cnt = m256_count(em);
        if (cnt > top) { second = top; top = cnt; target = c; } else if (cnt > second) second = cnt;
        rem = m256_and(rem, m256_not(em));
    }
    const bool still_valid = had_any && top > 2 * second;
    bool okf[4], bad[4]; uint32_t key[4], coords[4];
#pragma unroll
    for (int g = 0; g < 4; g++) {
        okf[g] = ok[g] && still_valid && crcv[g] == target;
        const int lvl = 64 * g + lane;
        bad[g] = lvl >= blk1 && lvl <= wht1 && e[g].result == REF_BAD_CRC;
        key[g] = ((uint32_t)e[g].hyst << 8) | e[g].shift;
        coords[g] = (uint32_t)(uint16_t)e[g].start | ((uint32_t)(uint16_t)e[g].stop << 16);
    }
    /* pickLevelByCRCStats among the lanes' levels */
    auto pick = [&](const bool (&t)[4], uint32_t max_hyst, uint32_t max_shift, uint8_t *out) -> bool {
        uint32_t mine = 0xFFFFFFFFu;
#pragma unroll
        for (int g = 0; g < 4; g++) if (t[g] && e[g].hyst <= max_hyst && e[g].shift <= max_shift && key[g] < mine) mine = key[g];
        const uint32_t best = wave_min_u32(mine);
        if (best == 0xFFFFFFFFu) return false;
        bool is[4];
#pragma unroll
        for (int g = 0; g < 4; g++) is[g] = t[g] && key[g] == best;
        const M256 E = m256_ballot(is);
        int high_ref = m256_top_le(E, wht1), low_ref;
        pick_longest_run(E, blk1, low_ref, high_ref);
        *out = (uint8_t)(low_ref + (uint8_t)(high_ref - low_ref) / 2);
        return true;
    };
    uint8_t span1 = SPAN_NOT_FOUND, span2 = SPAN_NOT_FOUND, ref = 0;
    if (had_any && still_valid) {
        if (top < (int)ps.min_valid_crcs) span1 = SPAN_TOO_NARROW;
        else span1 = pick(okf, 0x0F, SHIFT_STAGES_MAX, &ref) ? SPAN_OK : SPAN_NOT_FOUND;
    }
    if (span1 != SPAN_OK) {
        if (span1 == SPAN_TOO_NARROW) {
            /* pickLevelByCRCStatsOpt, within the limits of the mode */
            const uint32_t max_hyst = b.in_max_hyst, max_shift = b.in_max_shift;
            bool gd[4];
#pragma unroll
            for (int g = 0; g < 4; g++) gd[g] = okf[g] && e[g].hyst <= max_hyst && e[g].shift <= max_shift;
            const M256 good = m256_ballot(gd);
            int lo = blk1, hi = wht1, rl, rh;
            if (pick_widest_region(good, lo, hi, rl, rh)) { lo = rl; hi = rh; }
            span2 = pick_opt_walk(good, lo, hi, ps.max_ref_lvl, [&](int i) -> uint32_t { return level_read(key, i); }, &ref) ? SPAN_OK : SPAN_NOT_FOUND;
        } else span2 = pick(bad, 0xFF, 0xFF, &ref) ? SPAN_OK : SPAN_NOT_FOUND;
    }
    const uint32_t tk = level_read(key, ref), tc = level_read(coords, ref);
    if (lane == 0) {
        m.span1 = span1; m.span2 = span2; m.ref_level = ref; m.t_hyst = (uint8_t)(tk >> 8); m.t_shift = (uint8_t)(tk & 0xFF);
        m.t_start = (int16_t)(tc & 0xFFFF); m.t_stop = (int16_t)(tc >> 16);
        m.state = SWEEP_SETTLED;
    }
}

/* ---- the frame kernel of small rounds: a sweep settled while its frame waits ------------------------------ */
/* A round of a few frames is as long as one frame takes: the machine stands idle around them, and a frame that misses a sweep costs a round of its own to be
 * decoded again with the outcome.  sdv_k_stc007_frames_fat gives the frame's wave four more: they wait at the workgroup's barrier, and when the frame misses
 * a sweep they settle it - a wave per 64 levels, then the first of them the pick, exactly what the two sweep kernels do with a request - while the frame's wave
 * waits at the barrier in turn; it then goes on with the outcome, its pass is complete.  Barriers are the protocol (three per sweep, one at the end): the
 * frame's wave and the sweep bodies do not use the workgroup barrier for anything else (SDV_BLOCK_SYNC, SDV_WAVE_SYNC).  [The emulator runs one wave per
 * workgroup: the frame's wave does the work of the four itself.] */
enum { FAT_RUN = 1, FAT_EXIT = 2, FAT_WORKERS = 4 };
struct FatLds { SweepLds sw[FAT_WORKERS]; int32_t cmd, slot; };
__device__ inline SweepArgs fat_sweep_args(const FrameArgs &a, SweepEnt *levels, int slot)
{
    SweepArgs sa;
    sa.luma = a.luma; sa.frame_stride = a.frame_stride; sa.row_stride = a.row_stride; sa.width = a.width;
    sa.doubled = a.doubled; sa.mode = a.mode; sa.preset = a.preset;
    sa.memo = a.memo; sa.first = slot; sa.count = 1; sa.levels = levels;
    return sa;
}
/* the frame's wave: entry `slot` of the pool is settled when this returns */
__device__ inline void fat_sweep(SweepHook &h, int slot)
{
#ifdef SDV_EMU
    static SweepLds emu_lds;
    const SweepArgs sa = fat_sweep_args(*h.fa, h.fat_levels, slot);
    for (int g = 0; g < 4; g++) sweep_levels_body(sa, emu_lds, 0, g);
    SDV_BLOCK_SYNC();
    sweep_pick_body(sa, 0);
    SDV_BLOCK_SYNC();
#else
    if (lane_id() == 0) { h.fat->slot = slot; h.fat->cmd = FAT_RUN; }
    __syncthreads();            /* the four take it from here */
    __syncthreads();            /* ... the levels are done */
    __syncthreads();            /* ... the pick is done: the entry holds the outcome */
#endif
}
#ifndef SDV_EMU
__device__ inline void fat_worker(const FrameArgs &a, FatLds &fat, SweepEnt *levels, int w)
{
    for (;;) {
        __syncthreads();
        if (uni(fat.cmd) == FAT_EXIT) break;
        const SweepArgs sa = fat_sweep_args(a, levels, uni(fat.slot));
        sweep_levels_body(sa, fat.sw[w], 0, w);
        __syncthreads();
        if (w == 0) sweep_pick_body(sa, 0);
        __syncthreads();
    }
}
#endif
__device__ inline void fat_exit(FatLds &fat)
{
    if (lane_id() == 0) fat.cmd = FAT_EXIT;
    __syncthreads();
}

} // namespace sdv

#ifndef SDV_SWEEP_WAVES
#define SDV_SWEEP_WAVES 4
#endif
__global__ void __launch_bounds__(64, SDV_SWEEP_WAVES) sdv_k_stc007_sweep_levels(sdv::SweepArgs a)
{
    __shared__ sdv::SweepLds lds;
    sdv::sweep_levels_body(a, lds, (int)(blockIdx.x >> 2), (int)(blockIdx.x & 3));
}
__global__ void __launch_bounds__(64) sdv_k_stc007_sweep_pick(sdv::SweepArgs a)
{
    sdv::sweep_pick_body(a, (int)blockIdx.x);
}

#endif
