/*
 * pcm16_bin_device.h - PCM-16x0 (Sony PCM-1610/1620/1630) line binarizer: Binarizer::processLine with a PCM16X0SubLine output
 * (SURVEY.md section 8 row a9), one pass per third of a video line, the line staged once in LDS by its wavefront.
 *
 * Reference: binarizer.cpp:443-1724 (processLine, PCM-16x0 paths), :2603-2681 (findPCM16X0BW), :5819-6042 (findPCM16X0Coordinates),
 * :4514-5271 (searchPCM16X0Data), :7134-7319 (fillPCM16X0), :6599-7013 (pickCutBitsUpPCM16X0), :7560-7691 (fillDataWords),
 * :7695-8055 (readPCMdata); pcm16x0subline.cpp, pcmline.cpp for the line object.
 *
 * A sub-line is 64 bit cells (3 x 16 bit + CRCC): exactly one cell per lane, so a read is one LDS byte per lane, two ballots, the
 * two-level automaton solved on them (stc007_device.h, solve_automaton) and the CRC as 16 parities.  The 193 cells of the whole
 * line (three sub-lines and the control bit between the second and the third) share one set of data coordinates.
 *
 * The marker-less coordinate search - 21 x 21 coordinate pairs, all three parts read at each with the Bit Picker forced - is spread
 * over the lanes, one read per lane at a time, results in LDS.  The reference walks the grid in order and what it does depends on
 * what it has seen: it stops sweeping the right coordinate behind the window where all three parts read, stops sweeping the left one
 * where fewer than two still do, and a Bit Picker collision marks the one line object forced bad for every read that follows.  All of
 * that is replayed serially, over the lane-parallel results, by lane 0; the line object is then left as the last read the reference
 * would have made leaves it.
 */
#pragma once
#include "pcm1_bin_device.h"

namespace sdvp16 {
using namespace sdv;
using sdvp1b::BinCtx;
using sdvp1b::stats_reset;
using sdvp1b::stats_update;
using sdvp1b::stats_update_fresh;
using sdvp1b::sweep_blank;
using sdvp1b::rs_store;
using sdvp1b::pick_in_row;
using sdvp1b::row_read;
using sdvp1b::row_vote;
using sdvp1b::rs_unpack;

enum { P16_BITS = 193, P16_DATA = 64, P16_WORD_BITS = 16, P16_CRC_SILENT = 0x0E10, P16_SUBLINES = 3 };
enum { P16_SEARCH_STEP_DIV = 2, P16_SEARCH_MAX_OFS = 10, P16_SEARCH_STEP_CNT = (P16_SEARCH_MAX_OFS + 1) * 2 };     /* binarizer.h:262-264 */
enum { PART_FULL = 0, PART_LEFT = 1, PART_MIDDLE = 2, PART_RIGHT = 3 };        /* Binarizer::FULL_LINE, PART_PCM16X0_* (binarizer.h:217-224) */
/* where the per-part, combined and left-coordinate result rows sit in WaveLds::sweep */
enum { SW_P0 = 0, SW_P1 = 32, SW_P2 = 64, SW_RIGHT = 96, SW_LEFT = 128 };

struct P16Lds {
    WaveLds w;
    uint32_t grid[P16_SEARCH_STEP_CNT * P16_SEARCH_STEP_CNT * P16_SUBLINES];   /* per read: crc | hyst << 16 | shift << 20 | valid << 24 | collision << 25 | picked << 26 */
    CrcStat pstats[3][MAX_COLL_CRCS + 1];   /* scan_right_p0/p1/p2_crcs */
    CrcStat lstats[MAX_COLL_CRCS + 1];      /* scan_left_crcs (scan_right_crcs: w.crc_stats) */
    int32_t vote[8];
};

struct L16 {                        /* PCM16X0SubLine : PCMLine (pcmline.h:137-166, pcm16x0subline.h:113-125) */
    uint8_t black, white, ref_low, ref_level, ref_high, hyst, shift;
    Coords coords;
    bool ref_sweeped, coords_sweeped, by_ext_tune, bw_set, coords_set, forced_bad, control_bit;
    uint8_t service, line_part;
    uint16_t pixel_start, pixel_stop;
    int16_t pso; uint32_t psm, hpsm;
    uint64_t v;                     /* the 64 cells: word k = bits 48-16k .. 63-16k, CRCC = the low 16 bits */
    uint16_t calc_crc, queue_order;
    uint8_t picked_l, picked_r;
};

__device__ __forceinline__ uint16_t get_word(const L16 &l, int k) { return (uint16_t)(l.v >> (48 - 16 * k)); }

/* PCM16X0SubLine::calcCRC (pcm16x0subline.cpp:158-170): CRC-16/CCITT-FALSE over the 48 data cells, as a GF(2)-linear map of them */
struct Crc16Tables { uint64_t k[16]; uint16_t base; uint16_t col[4]; };      /* col[c]: what data cell c flips in the CRC (the Bit Picker's left bits) */
constexpr Crc16Tables make_crc16_tables()
{
    Crc16Tables t{};
    uint16_t c = 0xFFFF;
    for (int i = 0; i < 48; i++) c = crc16_step(c, 0);
    t.base = c;
    for (int b = 0; b < 48; b++) {          /* data cell b (0 = first = MSB of word 0) sits at bit 63 - b of v */
        uint16_t v = 0;
        for (int i = 0; i < 48; i++) v = crc16_step(v, i == b);
        for (int j = 0; j < 16; j++) if (v & (1u << j)) t.k[j] |= (1ull << (63 - b));
        if (b < 4) t.col[b] = v;
    }
    return t;
}
#ifdef SDV_EMU
static const Crc16Tables c_crc16 = make_crc16_tables();
#else
__device__ __constant__ const Crc16Tables c_crc16 = make_crc16_tables();
#endif
__device__ inline void calc_crc(L16 &l)             /* one lane on its own */
{
    uint32_t crc = 0;
    for (int j = 0; j < 16; j++) crc |= (uint32_t)(__popcll(l.v & c_crc16.k[j]) & 1) << j;
    l.calc_crc = (uint16_t)(crc ^ c_crc16.base);
}
__device__ inline bool crc_valid_ignore_forced(const L16 &l) { return l.calc_crc == (uint16_t)(l.v & 0xFFFF); }
__device__ inline bool crc_valid(const L16 &l) { return !l.forced_bad && crc_valid_ignore_forced(l); }
__device__ inline void set_invalid_crc(L16 &l) { l.v = (l.v & ~0xFFFFull) | (uint64_t)(uint16_t)~l.calc_crc; }
__device__ inline void base_clear(L16 &l)           /* PCMLine::clear, pcmline.cpp:96-116 */
{
    l.black = l.white = l.ref_low = l.ref_level = l.ref_high = 0;
    coords_clear(l.coords);
    l.hyst = l.shift = 0;
    l.ref_sweeped = l.coords_sweeped = l.by_ext_tune = false;
    l.calc_crc = 0;
    l.bw_set = l.coords_set = l.forced_bad = false;
    l.service = SDV_SRV_NO;
    l.pixel_start = 0; l.pixel_stop = 1; l.pso = 0; l.psm = 128; l.hpsm = 64;
}
__device__ inline void p16_clear(L16 &l)            /* PCM16X0SubLine::clear, pcm16x0subline.cpp:59-79 */
{
    base_clear(l);
    l.control_bit = true; l.line_part = 0; l.picked_l = l.picked_r = 0; l.queue_order = 0;
    l.v = 0; l.calc_crc = P16_CRC_SILENT;
    set_invalid_crc(l);
}
__device__ inline void set_service(L16 &l, uint8_t srv) { base_clear(l); l.service = srv; }     /* PCMLine::setServiceLine: base clear() only */
__device__ inline void set_ppb(L16 &l, const Coords &c)     /* pcmline.cpp:506-519 with 193 cells between the coordinates */
{
    l.psm = (uint32_t)((int)c.stop - (int)c.start);
    l.psm = (l.psm * 128u + P16_BITS / 2) / P16_BITS;
    l.pso = c.start;
    l.hpsm = (l.psm + 1) / 2;
}
__device__ inline uint8_t get_ppb(const L16 &l) { return (uint8_t)(l.psm / 128u); }
/* getVideoPixeBylCalc (pcmline.cpp:249-311): both shift tables are {0, +1, -1, +2, -2}, so the shift is uniform along the line */
__device__ inline int pixel_of(const L16 &l, int bit, int stage)
{
    int32_t vp = (int32_t)((uint32_t)bit * l.psm + l.hpsm);
    vp = vp / 128 + l.pso;
    const int sh = stage == 0 ? 0 : (stage == 1 ? 1 : (stage == 2 ? -1 : (stage == 3 ? 2 : -2)));
    vp += sh;
    if (vp < (int32_t)l.pixel_start) vp = l.pixel_start;
    else if (vp >= (int32_t)l.pixel_stop) vp = (int32_t)l.pixel_stop - 1;
    return vp;
}
__device__ __forceinline__ int part_start_bit(uint8_t part) { return part == PART_LEFT ? 0 : (part == PART_MIDDLE ? P16_DATA : 2 * P16_DATA + 1); }

/* fillPCM16X0 (binarizer.cpp:7134-7319), one lane on its own: the two comparisons of the part's 64 cell centres are collected as masks, the
 * two-level automaton is solved on them (solve_automaton_lane) */
__device__ inline void fill_pcm16(L16 &l, const uint8_t *px_row, uint8_t part, int stage)
{
    const int b0 = part_start_bit(part);
    int32_t acc = (int32_t)((uint32_t)b0 * l.psm + l.hpsm) + ((int32_t)l.pso + shift_of_stage(stage)) * 128;
    const int32_t lo = l.pixel_start, hi = (int32_t)l.pixel_stop - 1;
    uint32_t a0, a1, c0, c1;
    compare_cells32<32>(px_row, acc, (int32_t)l.psm, lo, hi, l.ref_low, l.ref_high, a0, c0);
    compare_cells32<32>(px_row, acc, (int32_t)l.psm, lo, hi, l.ref_low, l.ref_high, a1, c1);
    const uint64_t a_lo = (uint64_t)a0 | ((uint64_t)a1 << 32), b_lo = (uint64_t)c0 | ((uint64_t)c1 << 32);
    uint64_t s_lo, s_hi;
    solve_automaton_lane(a_lo, 0ull, b_lo, 0ull, s_lo, s_hi);
    l.v = __brevll(s_lo);
    calc_crc(l);
    l.control_bit = true;
    if (crc_valid(l)) if (px_row[pixel_of(l, 2 * P16_DATA, stage)] < l.ref_level) l.control_bit = false;
}
/* the same for the whole wave (wave-uniform callers only): lane i samples cell i of the part */
__device__ inline void fill_pcm16_wave(L16 &l, const uint8_t *px_row, uint8_t part, int stage)
{
    const int lane = lane_id();
    const uint8_t p0 = px_row[pixel_of(l, part_start_bit(part) + lane, stage)];
    const uint64_t a_lo = __ballot(p0 > l.ref_low), b_lo = __ballot(p0 >= l.ref_high);
    uint64_t s_lo, s_hi;
    solve_automaton(a_lo, 0ull, b_lo, 0ull, s_lo, s_hi);
    l.v = __brevll(s_lo);
    const int par = __popcll(l.v & c_crc16.k[lane & 15]) & 1;
    const uint64_t cb = __ballot(par);
    l.calc_crc = (uint16_t)((uint16_t)(cb & 0xFFFF) ^ c_crc16.base);
    l.control_bit = true;
    if (crc_valid(l)) if (px_row[pixel_of(l, 2 * P16_DATA, stage)] < l.ref_level) l.control_bit = false;
}

/* pickCutBitsUpPCM16X0 (binarizer.cpp:6599-7013): the left bits of the left part's first word, the right bits of the right part's CRCC */
__device__ inline void pick_cut_bits(const BinCtx &c, L16 &l, uint8_t part)
{
    l.picked_l = l.picked_r = 0;
    if (part != PART_LEFT && part != PART_RIGHT) return;
    const bool left = part == PART_LEFT;
    int max_cut = left ? c.ps.left_bit_pick : c.ps.right_bit_pick; if (c.mode == SDV_MODE_DRAFT) max_cut /= 2;
    int first = left ? c.scan_start : c.scan_end, bits = 0;
    const int half_ppb = ((int)get_ppb(l) + 1) / 2;
    for (int i = 0; i < max_cut; i++) {
        const int cur = pixel_of(l, left ? i : P16_BITS - 1 - i, 0);
        if ((left ? (cur - first) : (first - cur)) >= half_ppb) break;
        if (i == 0) first = cur;
        bits = i + 1;
    }
    if (c.force_bit_picker && crc_valid(l)) { if (left) l.picked_l = (uint8_t)bits; else l.picked_r = (uint8_t)bits; return; }
    if (bits == 0) return;
    if (l.forced_bad) return;           /* nothing reads valid on a line that is forced bad: the search would put everything back */
    /* Every value of the cut-off bits; exactly one may give a valid CRC (:6760-6990).  The CRC is linear in the cells: each of the left
     * part's first cells flips a fixed pattern of CRC bits; the right part's cut-off bits are the low bits of the CRCC as read, so there
     * at most one value fits. */
    const uint32_t lim = 1u << bits;
    if (left) {
        const int sh = 64 - bits;
        const uint64_t clean = l.v & ~((uint64_t)(lim - 1) << sh);
        uint32_t base = 0;
        for (int j = 0; j < 16; j++) base |= (uint32_t)(__popcll(clean & c_crc16.k[j]) & 1) << j;
        base ^= c_crc16.base;
        const uint16_t crcc = (uint16_t)(l.v & 0xFFFF);
        int found = 0; uint32_t fix = 0; uint16_t crc_fix = 0;
        for (uint32_t i = 0; i < lim; i++) {
            uint32_t crc = base;
            for (int t = 0; t < bits; t++) if ((i >> t) & 1u) crc ^= c_crc16.col[bits - 1 - t];       /* bit t of the value sits at bit sh + t of v = data cell bits - 1 - t */
            if ((uint16_t)crc == crcc) { if (found) { found = 2; break; } found = 1; fix = i; crc_fix = (uint16_t)crc; }
        }
        if (found == 2) { l.forced_bad = true; return; }
        if (found == 0) return;
        l.v = clean | ((uint64_t)fix << sh); l.calc_crc = crc_fix;
        l.picked_l = (uint8_t)bits;
    } else {
        const uint16_t mask = (uint16_t)(lim - 1), crcc_clean = (uint16_t)((uint16_t)(l.v & 0xFFFF) & ~mask);
        if ((uint16_t)(l.calc_crc & ~mask) != crcc_clean) return;       /* the data cells are not touched: calc_crc stands */
        l.v = (l.v & ~(uint64_t)0xFFFF) | (uint64_t)(uint16_t)(crcc_clean | (l.calc_crc & mask));
        l.picked_r = (uint8_t)bits;
    }
}

/* fillDataWords (binarizer.cpp:7560-7670); false = the levels clip (STG_NO_GOOD) */
template <bool kWave>
__device__ inline bool fill_data_words(const BinCtx &c, L16 &l, const uint8_t *px_row, uint8_t part, uint8_t ref_delta, uint8_t shift_stg)
{
    if (ref_delta > HYST_DEPTH_MAX || shift_stg > SHIFT_STAGES_MAX) return false;
    const uint8_t low_ref = get_low_level(l.ref_level, ref_delta), high_ref = get_high_level(l.ref_level, ref_delta);
    l.ref_low = low_ref; l.ref_high = high_ref;
    if (low_ref <= l.black) { set_invalid_crc(l); return false; }
    if (high_ref >= l.white) { set_invalid_crc(l); return false; }
    l.hyst = ref_delta; l.shift = shift_stg;
    if (part == PART_FULL) return false;                    /* fillPCM16X0 answers STG_NO_GOOD outside the three parts (:7199-7203) */
    if (kWave) fill_pcm16_wave(l, px_row, part, shift_stg); else fill_pcm16(l, px_row, part, shift_stg);
    if ((!crc_valid(l) && (l.ref_level > c.ps.min_white_lvl) && ((c.ps.left_bit_pick != 0) || (c.ps.right_bit_pick != 0))) || c.force_bit_picker)
        pick_cut_bits(c, l, part);
    return true;
}

/* readPCMdata (binarizer.cpp:7695-8055) (see pcm1_bin_device.h, read_pcm_data) */
template <bool kWave>
__device__ inline void read_pcm_data(const BinCtx &c, L16 &l, const uint8_t *px_row, uint8_t part, uint8_t hyst_lim, uint8_t shift_lim)
{
    set_ppb(l, l.coords);
    if (hyst_lim > HYST_DEPTH_MAX) hyst_lim = HYST_DEPTH_MAX;
    if (shift_lim > SHIFT_STAGES_MAX) shift_lim = SHIFT_STAGES_MAX;
    if (l.ref_sweeped) { fill_data_words<kWave>(c, l, px_row, part, hyst_lim, shift_lim); return; }     /* isDataByRefSweep(), :7741 */
    bool found = false;
    /* the first fill (depth 0, stage 0) is kept: the reference's final fill repeats it when nothing reads valid (pcm1_bin_device.h) */
    const bool entry_forced = l.forced_bad;
    bool kept = false;
    uint64_t k_v = 0; uint16_t k_crc = 0; uint8_t k_pl = 0, k_pr = 0, k_lo = 0, k_hi = 0; bool k_cb = true;
    for (uint8_t h = 0; h <= hyst_lim && !found; h++) {
        bool invalid_hyst = false;
        for (uint8_t s = 0; s <= shift_lim; s++) {
            if (!fill_data_words<kWave>(c, l, px_row, part, h, s)) { invalid_hyst = true; break; }
            if (crc_valid(l)) { found = true; break; }
            if (h == 0 && s == 0 && c.force_bit_picker) { kept = true; k_v = l.v; k_crc = l.calc_crc; k_pl = l.picked_l; k_pr = l.picked_r; k_lo = l.ref_low; k_hi = l.ref_high; k_cb = l.control_bit; }
        }
        if (invalid_hyst) break;
    }
    if (!found) {
        if (kept && l.forced_bad == entry_forced) { l.v = k_v; l.calc_crc = k_crc; l.picked_l = k_pl; l.picked_r = k_pr; l.ref_low = k_lo; l.ref_high = k_hi; l.hyst = 0; l.shift = 0; l.control_bit = k_cb; }
        else fill_data_words<kWave>(c, l, px_row, part, 0, 0);
    }
}

/* ---- searchPCM16X0Data (binarizer.cpp:4514-5271) ---------------------------------------------------------------------------- */
__device__ inline SweepEnt grid_entry(uint32_t g, bool valid, int16_t start, int16_t stop)
{
    SweepEnt e = sweep_blank();
    e.crc = (uint16_t)(g & 0xFFFF); e.hyst = (uint8_t)((g >> 16) & 0xF); e.shift = (uint8_t)((g >> 20) & 0xF);
    e.start = start; e.stop = stop;
    e.result = valid ? REF_CRC_OK : REF_BAD_CRC;
    return e;
}
/* findMostFrequentCRC without skip_equal (:1829-1928): the most frequent entry wins whatever the runner-up's count */
__device__ inline void stats_most_frequent_noskip(CrcStat *a, uint8_t &valid_cnt)
{
    a[0].result = 0; a[0].idx = 0; a[0].hyst = 0; a[0].shift = 0;
    if (valid_cnt >= MAX_COLL_CRCS) valid_cnt = MAX_COLL_CRCS - 1;
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (a[i].result > a[0].result) { a[0].result = a[i].result; a[0].crc = a[i].crc; a[0].hyst = a[i].hyst; a[0].shift = a[i].shift; a[0].idx = i; }
    if (a[0].result == 0) valid_cnt = 0;
}
__device__ __forceinline__ uint8_t sat_f(int v) { return (uint8_t)(v > 0x0F ? 0x0F : v); }

/* The walk of searchPCM16X0Data over a grid without collisions: row after row, the columns of a row on the lanes.  What the serial walk
 * keeps in tables is here a handful of masks: which columns read valid per part, which of those survive the part's vote, which columns
 * combine to a valid line.  Leaves lds.vote as the serial walk does. */
__device__ inline void walk_rows_parallel(P16Lds &lds, int nl, int nr, int l0, int r1, int scan_step, uint64_t rows_live)
{
    const int lane = lane_id();
    SweepEnt *sw = lds.w.sweep;
    uint8_t valid_left = 0;             /* lane 0's */
    bool lock_left = false;
    int last_read = -1;
    if (lane == 0) stats_reset(lds.lstats, MAX_COLL_CRCS);
    if (lane < P16_SEARCH_STEP_CNT) sw[SW_LEFT + lane] = sweep_blank();
    SDV_WAVE_SYNC();
    const uint32_t colmask = nr >= 32 ? 0xFFFFFFFFu : ((1u << nr) - 1u);
    for (int row = 0; row < nl; row++) {
        if (!((rows_live >> row) & 1ull)) { last_read = (row * nr + nr - 1) * P16_SUBLINES + 2; continue; }
        uint32_t g0 = 0, g1 = 0, g2 = 0;
        if (lane < nr) { const uint32_t *gp = &lds.grid[(row * nr + lane) * P16_SUBLINES]; g0 = gp[0]; g1 = gp[1]; g2 = gp[2]; }
        uint32_t m0 = (uint32_t)__ballot(lane < nr && ((g0 >> 24) & 1u)), m1 = (uint32_t)__ballot(lane < nr && ((g1 >> 24) & 1u)), m2 = (uint32_t)__ballot(lane < nr && ((g2 >> 24) & 1u));
        /* the columns the walk visits: it stops behind the window in which all three parts read, at the first column where none does */
        int n_vis = nr;
        {
            const uint32_t all3 = m0 & m1 & m2, none = ~(m0 | m1 | m2) & colmask;
            if (all3) { const int c0 = __ffs((int)all3) - 1; const uint32_t stop = c0 >= 31 ? 0u : (none & ~((2u << c0) - 1u)); if (stop) n_vis = __ffs((int)stop); }
        }
        const uint32_t vis = n_vis >= 32 ? 0xFFFFFFFFu : ((1u << n_vis) - 1u);
        m0 &= vis; m1 &= vis; m2 &= vis;
        last_read = (row * nr + n_vis - 1) * P16_SUBLINES + 2;
        const uint32_t any = m0 | m1 | m2;
        if (!any) continue;
        const int step_min = __ffs((int)any) - 1, step_max = 31 - __clz((int)any);
        /* per part: the most frequent CRC among the columns that read (first seen wins a tie; a rival with half its count or more voids
         * the vote, findMostFrequentCRC :1829-1928), the others are marked as collisions (invalidateNonFrequentCRCs) */
        uint32_t ok0 = 0, ok1 = 0, ok2 = 0;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const uint32_t m = p == 0 ? m0 : (p == 1 ? m1 : m2), g = p == 0 ? g0 : (p == 1 ? g1 : g2);
            if (!m) continue;
            uint32_t tcnt, tfirst;
            const uint32_t okp = row_vote(m, g & 0xFFFFu, &lds.grid[row * nr * P16_SUBLINES + p], P16_SUBLINES, step_min, step_max, tcnt, tfirst);
            if (p == 0) ok0 = okp; else if (p == 1) ok1 = okp; else ok2 = okp;
        }
        /* the line a column combines to (:4893-5050) */
        const int col = lane & 31;           /* (the masks are 32 columns wide; lanes behind them are never in range) */
        const bool o0 = lane < 32 && ((ok0 >> col) & 1u), o1 = lane < 32 && ((ok1 >> col) & 1u), o2 = lane < 32 && ((ok2 >> col) & 1u);
        const uint32_t h0 = (g0 >> 16) & 0xFu, h1 = (g1 >> 16) & 0xFu, h2 = (g2 >> 16) & 0xFu, s0 = (g0 >> 20) & 0xFu, s1 = (g1 >> 20) & 0xFu, s2 = (g2 >> 20) & 0xFu;
        const bool in_range = lane >= step_min && lane <= step_max;
        bool r_ok = false; uint32_t r_hyst = 0, r_shift = 0; int n3 = 0;
        if (o1) {
            n3 = 1; r_ok = true; r_shift = s1;
            uint32_t hy = h1;
            if (o2) { n3++; hy = (uint8_t)(hy + h2); if (s2 > r_shift) r_shift = s2; } else hy = (uint8_t)(hy + HYST_DEPTH_SAFE);
            if (o0) { n3++; hy = (uint8_t)(hy + h0); if (s0 > r_shift) r_shift = s0; } else hy = (uint8_t)(hy + HYST_DEPTH_SAFE);
            r_hyst = sat_f((int)hy);
        } else if (o0 && o2) {
            n3 = 2; r_ok = true; r_hyst = h2; r_shift = s2;
            if (h0 > r_hyst) { r_hyst = h0; r_shift = s0; }
            else if (h0 == r_hyst) { if (s0 > r_shift) r_shift = s0; }
            r_hyst = sat_f((int)(uint8_t)(r_hyst + HYST_DEPTH_SAFE));
        }
        if (__ballot(in_range && n3 == P16_SUBLINES) != 0ull) lock_left = true;
        uint8_t right_ofs = 0xFF;
        bool valid_right = __ballot(in_range && r_ok) != 0ull;
        if (valid_right) valid_right = pick_in_row(r_ok, r_hyst, r_shift, step_min, step_max, right_ofs);
        if (!valid_right) {             /* the fallback: the right part alone, else the left one (:5080-5140) */
            r_ok = false;
            if (o2) { r_ok = true; r_shift = s2; r_hyst = sat_f((int)(uint8_t)(h2 + HYST_DEPTH_MAX)); }
            else if (o0) { r_ok = true; r_shift = s0; r_hyst = sat_f((int)(uint8_t)(h0 + 2 * HYST_DEPTH_SAFE)); }
            valid_right = __ballot(in_range && r_ok) != 0ull;
            if (valid_right) valid_right = pick_in_row(r_ok, r_hyst, r_shift, step_min, step_max, right_ofs);
        }
        if (valid_right) {
            const uint32_t le_h = row_read(r_hyst, right_ofs), le_s = row_read(r_shift, right_ofs);
            if (lane == 0) {
                SweepEnt le = sweep_blank();
                le.result = REF_CRC_OK; le.crc = P16_CRC_SILENT; le.hyst = (uint8_t)le_h; le.shift = (uint8_t)le_s;
                le.start = (int16_t)(l0 + row * scan_step); le.stop = (int16_t)(r1 - (int)right_ofs * scan_step);
                sw[SW_LEFT + row] = le;
                stats_update(lds.lstats, le.crc, le.hyst, le.shift, valid_left);
            }
            if (lock_left) {
                const int nv = (int)((ok0 >> right_ofs) & 1u) + (int)((ok1 >> right_ofs) & 1u) + (int)((ok2 >> right_ofs) & 1u);
                if (nv < 2) break;
            }
        }
    }
    SDV_WAVE_SYNC();
    if (lane == 0) {
        uint8_t left_ofs = 0xFF;
        if (valid_left > 0) {
            stats_most_frequent_noskip(lds.lstats, valid_left);
            for (int i = 0; i < P16_SEARCH_STEP_CNT; i++)
                if (sw[SW_LEFT + i].result == REF_CRC_OK) { if (valid_left == 0 || sw[SW_LEFT + i].crc != lds.lstats[0].crc) sw[SW_LEFT + i].result = REF_CRC_COLL; }
        }
        if (valid_left > 0)
            if (pick_level_by_crc_stats_at(sw + SW_LEFT, &left_ofs, 0, P16_SEARCH_STEP_CNT - 1, REF_CRC_OK, 0x0F, SHIFT_STAGES_MAX) != SPAN_OK) valid_left = 0;
        lds.vote[0] = valid_left > 0 ? 1 : 0;
        if (valid_left > 0) { lds.vote[1] = sw[SW_LEFT + left_ofs].start; lds.vote[2] = sw[SW_LEFT + left_ofs].stop; }
        lds.vote[3] = last_read; lds.vote[4] = -1;
    }
}

/* Returns true when coordinates were found; l is left as the reference leaves its line object. */
__device__ inline bool search_pcm16_data(BinCtx &c, L16 &l, P16Lds &lds, Coords data_loc, uint8_t &hyst_lim, uint8_t &shift_lim)
{
    const int lane = lane_id();
    int scan_step = 1, l0 = 0, l1 = 0, r0 = 0, r1 = 0;
    for (int guard = 2; guard > 0; guard--) {
        set_ppb(l, data_loc);
        scan_step = get_ppb(l);
        scan_step = scan_step >= P16_SEARCH_STEP_DIV ? scan_step / P16_SEARCH_STEP_DIV : 1;
        const int span = (uint16_t)(scan_step * P16_SEARCH_MAX_OFS);
        l0 = (int16_t)(data_loc.start - span); l1 = (int16_t)(data_loc.start + span);
        r0 = (int16_t)(data_loc.stop - span); r1 = (int16_t)(data_loc.stop + span);
        const int ss = c.scan_start, se = c.scan_end;
        if ((l0 < ss && l1 < ss) || (l0 > ss && l1 > ss) || (r0 < se && r1 < se) || (r0 > se && r1 > se)) { data_loc.start = (int16_t)ss; data_loc.stop = (int16_t)se; }
        else break;
    }
    const bool bitpick_previous = c.force_bit_picker;
    c.force_bit_picker = true;
    hyst_lim = 0;
    shift_lim = (c.mode == SDV_MODE_DRAFT || c.mode == SDV_MODE_FAST) ? 0 : SHIFT_STAGES_SAFE;
    const int n_left = (l1 - l0) / scan_step + 1, n_right = (r1 - r0) / scan_step + 1;
    const int nl = n_left < P16_SEARCH_STEP_CNT ? n_left : P16_SEARCH_STEP_CNT, nr = n_right < P16_SEARCH_STEP_CNT ? n_right : P16_SEARCH_STEP_CNT;
    const int n_reads = nl * nr * P16_SUBLINES;
    const bool entry_forced = l.forced_bad;
    /* every read of the grid, as if the line object came to it clean: (row, col, part) in the reference's order */
    SDV_WAVE_SYNC();
    K1_T(ts0_);
    for (int q = lane; q < n_reads; q += 64) {
        const int pair = q / P16_SUBLINES, part = q - pair * P16_SUBLINES, row = pair / nr, col = pair - row * nr;
        L16 t = l;
        coords_set(t.coords, (int16_t)(l0 + row * scan_step), (int16_t)(r1 - col * scan_step));
        read_pcm_data<false>(c, t, lds.w.px, (uint8_t)(PART_LEFT + part), hyst_lim, shift_lim);
        int hy = t.hyst;
        const bool picked = (part == 0 && t.picked_l != 0) || (part == 2 && t.picked_r != 0);
        if (part == 0 && t.picked_l != 0) hy = sat_f(hy + 0x02);
        if (part == 2 && t.picked_r != 0) hy = sat_f(hy + 0x03);
        lds.grid[q] = (uint32_t)(uint16_t)(t.v & 0xFFFF) | ((uint32_t)(hy & 0xF) << 16) | ((uint32_t)(t.shift & 0xF) << 20) | ((uint32_t)(crc_valid(t) ? 1 : 0) << 24)
                      | ((uint32_t)((t.forced_bad && !entry_forced) ? 1 : 0) << 25) | ((uint32_t)(picked ? 1 : 0) << 26);
    }
    SDV_WAVE_SYNC();
    K1_T(ts1_); K1_ADD(19, ts0_, ts1_);
    /* rows of the grid in which something happens: a read that is valid, or a Bit Picker collision (the walk below leaves every other row
     * as it finds it, apart from noting its last read) */
    uint64_t rows_live;
    bool any_coll;
    {
        bool live = false, coll = false;
        if (lane < nl) for (int i = 0; i < nr * P16_SUBLINES; i++) { const uint32_t g = lds.grid[lane * nr * P16_SUBLINES + i]; live = live || ((g >> 24) & 3u) != 0; coll = coll || ((g >> 25) & 1u) != 0; }
        rows_live = __ballot(live);
        any_coll = __ballot(coll) != 0ull;
    }
    /* The walk over the grid and the votes.  Without a Bit Picker collision anywhere (and a line object that is not forced bad to begin
     * with) the rows do not influence each other's reads and a row is worked on by the whole wave (walk_rows_parallel); otherwise the
     * walk is replayed cell by cell on lane 0. */
    K1_T(ts2_); K1_ADD(18, ts1_, ts2_);
    if (!any_coll && !entry_forced && nr <= 32) { walk_rows_parallel(lds, nl, nr, l0, r1, scan_step, rows_live); K1_T(ts3_); K1_ADD(20, ts2_, ts3_); }
    else if (lane == 0) {
        SweepEnt *sw = lds.w.sweep;
        uint8_t valid_left = 0, left_ofs = 0xFF;
        bool forced = entry_forced, lock_left = false;
        int last_read = -1, coll_read = -1;
        stats_reset(lds.lstats, MAX_COLL_CRCS);
        for (int i = 0; i < P16_SEARCH_STEP_CNT; i++) sw[SW_LEFT + i] = sweep_blank();
        for (int row = 0; row < nl; row++) {
            if (forced || !((rows_live >> row) & 1ull)) { last_read = (row * nr + nr - 1) * P16_SUBLINES + 2; continue; }      /* nothing reads in this row: it changes nothing */
            /* the row's tables are not cleared: everything below reads only the entries this row has written (the columns it visited, the
             * span between its first and last valid column) and the statistics start over by count */
            uint8_t valid_right = 0, valid_p[3] = { 0, 0, 0 }, right_ofs = 0xFF;
            bool lock_right = false, lock_min = false;
            int step_min = 0, step_max = P16_SEARCH_STEP_CNT, n_vis = 0;
            const int16_t start_ofs = (int16_t)(l0 + row * scan_step);
            for (int col = 0; col < nr; col++) {
                const int16_t stop_ofs = (int16_t)(r1 - col * scan_step);
                bool ok[3];
                for (int p = 0; p < 3; p++) {
                    const int q = (row * nr + col) * P16_SUBLINES + p;
                    const uint32_t g = lds.grid[q];
                    /* a line object that is forced bad reads nothing valid; the Bit Picker then tries to patch and puts the words back */
                    ok[p] = !forced && ((g >> 24) & 1) != 0;
                    uint32_t ge = g;
                    if (forced) ge = (g & 0xFFFFu) | (g & (0xFu << 20));           /* CRCC as read stays; no picked-bits penalty; depth 0 */
                    sw[(p == 0 ? SW_P0 : (p == 1 ? SW_P1 : SW_P2)) + col] = grid_entry(ge, ok[p], start_ofs, stop_ofs);
                    if (ok[p]) {
                        const SweepEnt &e = sw[(p == 0 ? SW_P0 : (p == 1 ? SW_P1 : SW_P2)) + col];
                        stats_update_fresh(lds.pstats[p], e.crc, e.hyst, e.shift, valid_p[p]);
                        if (!lock_min) { step_min = col; lock_min = true; }
                        step_max = col;
                    }
                    if (!forced && ((g >> 25) & 1) != 0) { forced = true; if (coll_read < 0) coll_read = q; }
                    last_read = q;
                }
                n_vis = col + 1;
                if (lock_right && !ok[0] && !ok[1] && !ok[2]) break;
                if (!lock_right && ok[0] && ok[1] && ok[2]) lock_right = true;
            }
            if (!lock_min) continue;        /* nothing read valid in this row (a collision made it live): no vote, nothing to hand up */
            for (int p = 0; p < 3; p++)
                if (valid_p[p] > 0) {
                    sdvp1b::stats_most_frequent(lds.pstats[p], valid_p[p]);
                    const int base = p == 0 ? SW_P0 : (p == 1 ? SW_P1 : SW_P2);
                    /* invalidateNonFrequentCRCs (:1931-1982) on this part's row */
                    for (int i = 0; i < n_vis; i++)
                        if (sw[base + i].result == REF_CRC_OK) { if (valid_p[p] == 0 || sw[base + i].crc != lds.pstats[p][0].crc) sw[base + i].result = REF_CRC_COLL; }
                }
            if (step_max >= P16_SEARCH_STEP_CNT) step_max = P16_SEARCH_STEP_CNT - 1;
            uint8_t valid_crcs = 0;
            for (int i = step_min; i <= step_max; i++) {
                SweepEnt &r = sw[SW_RIGHT + i];
                const SweepEnt p0 = sw[SW_P0 + i], p1 = sw[SW_P1 + i], p2 = sw[SW_P2 + i];
                valid_crcs = 0;
                if (p1.result == REF_CRC_OK) {
                    valid_crcs++;
                    r.result = REF_CRC_OK; r.crc = P16_CRC_SILENT; r.shift = p1.shift; r.start = p1.start; r.stop = p1.stop;
                    int hy = p1.hyst;
                    if (p2.result == REF_CRC_OK) { valid_crcs++; hy = (uint8_t)(hy + p2.hyst); if (p2.shift > r.shift) r.shift = p2.shift; } else hy = (uint8_t)(hy + HYST_DEPTH_SAFE);
                    if (p0.result == REF_CRC_OK) { valid_crcs++; hy = (uint8_t)(hy + p0.hyst); if (p0.shift > r.shift) r.shift = p0.shift; } else hy = (uint8_t)(hy + HYST_DEPTH_SAFE);
                    r.hyst = sat_f(hy);
                    stats_update_fresh(lds.w.crc_stats, r.crc, r.hyst, r.shift, valid_right);
                } else if (p0.result == REF_CRC_OK && p2.result == REF_CRC_OK) {
                    valid_crcs = 2;
                    r.result = REF_CRC_OK; r.crc = P16_CRC_SILENT; r.hyst = p2.hyst; r.shift = p2.shift; r.start = p2.start; r.stop = p2.stop;
                    if (p0.hyst > r.hyst) { r.hyst = p0.hyst; r.shift = p0.shift; }
                    else if (p0.hyst == r.hyst) { if (p0.shift > r.shift) r.shift = p0.shift; }
                    r.hyst = sat_f((uint8_t)(r.hyst + HYST_DEPTH_SAFE));
                    stats_update_fresh(lds.w.crc_stats, r.crc, r.hyst, r.shift, valid_right);
                } else r.result = REF_BAD_CRC;
                if (valid_crcs == P16_SUBLINES) lock_left = true;
            }
            auto pick_right = [&](uint8_t &ofs) -> bool {
                return pick_level_by_crc_stats_at(sw + SW_RIGHT, &ofs, (uint8_t)step_min, (uint8_t)step_max, REF_CRC_OK, 0x0F, SHIFT_STAGES_MAX) == SPAN_OK;
            };
            if (valid_right > 0) if (!pick_right(right_ofs)) valid_right = 0;
            if (valid_right == 0) {
                for (int i = step_min; i <= step_max; i++) {
                    SweepEnt &r = sw[SW_RIGHT + i];
                    const SweepEnt p0 = sw[SW_P0 + i], p2 = sw[SW_P2 + i];
                    if (p2.result == REF_CRC_OK) {
                        r.result = REF_CRC_OK; r.crc = P16_CRC_SILENT; r.shift = p2.shift; r.start = p2.start; r.stop = p2.stop;
                        r.hyst = sat_f((uint8_t)(p2.hyst + HYST_DEPTH_MAX));
                        stats_update_fresh(lds.w.crc_stats, r.crc, r.hyst, r.shift, valid_right);
                    } else if (p0.result == REF_CRC_OK) {
                        r.result = REF_CRC_OK; r.crc = P16_CRC_SILENT; r.shift = p0.shift; r.start = p0.start; r.stop = p0.stop;
                        r.hyst = sat_f((uint8_t)(p0.hyst + 2 * HYST_DEPTH_SAFE));
                        stats_update_fresh(lds.w.crc_stats, r.crc, r.hyst, r.shift, valid_right);
                    } else r.result = REF_BAD_CRC;
                }
                if (valid_right > 0) if (!pick_right(right_ofs)) valid_right = 0;
            }
            if (valid_right > 0) {
                SweepEnt le = sw[SW_RIGHT + right_ofs];
                le.result = REF_CRC_OK;
                sw[SW_LEFT + row] = le;
                stats_update(lds.lstats, le.crc, le.hyst, le.shift, valid_left);
                if (lock_left) {
                    int nv = 0;
                    if (sw[SW_P0 + right_ofs].result == REF_CRC_OK) nv++;
                    if (sw[SW_P1 + right_ofs].result == REF_CRC_OK) nv++;
                    if (sw[SW_P2 + right_ofs].result == REF_CRC_OK) nv++;
                    if (nv < 2) break;
                }
            }
        }
        if (valid_left > 0) {
            stats_most_frequent_noskip(lds.lstats, valid_left);
            for (int i = 0; i < P16_SEARCH_STEP_CNT; i++)
                if (sw[SW_LEFT + i].result == REF_CRC_OK) { if (valid_left == 0 || sw[SW_LEFT + i].crc != lds.lstats[0].crc) sw[SW_LEFT + i].result = REF_CRC_COLL; }
        }
        if (valid_left > 0)
            if (pick_level_by_crc_stats_at(sw + SW_LEFT, &left_ofs, 0, P16_SEARCH_STEP_CNT - 1, REF_CRC_OK, 0x0F, SHIFT_STAGES_MAX) != SPAN_OK) valid_left = 0;
        lds.vote[0] = valid_left > 0 ? 1 : 0;
        if (valid_left > 0) { lds.vote[1] = sw[SW_LEFT + left_ofs].start; lds.vote[2] = sw[SW_LEFT + left_ofs].stop; }
        lds.vote[3] = last_read; lds.vote[4] = coll_read;
    }
    SDV_WAVE_SYNC();
    const bool found = lds.vote[0] != 0;
    const int f_start = lds.vote[1], f_stop = lds.vote[2], last_read = lds.vote[3], coll_read = lds.vote[4];
    SDV_WAVE_SYNC();
    /* what the last read the reference made leaves in the line object: forced bad if a collision happened before it */
    if (last_read >= 0) {
        const int pair = last_read / P16_SUBLINES, part = last_read - pair * P16_SUBLINES, row = pair / nr, col = pair - row * nr;
        if (coll_read >= 0 && coll_read < last_read) l.forced_bad = true;
        coords_set(l.coords, (int16_t)(l0 + row * scan_step), (int16_t)(r1 - col * scan_step));
        read_pcm_data<true>(c, l, lds.w.px, (uint8_t)(PART_LEFT + part), hyst_lim, shift_lim);
    }
    c.force_bit_picker = bitpick_previous;
    if (found) {
        l.coords.start = (int16_t)f_start; l.coords.stop = (int16_t)f_stop;
        l.coords_set = true; l.coords_sweeped = true;
        return true;
    }
    l.coords = data_loc;
    l.coords_sweeped = false;
    return false;
}

/* findPCM16X0Coordinates (binarizer.cpp:5819-6042); scan_done = VideoLine::scan_done of the line being read */
__device__ inline void find_pcm16_coordinates(BinCtx &c, L16 &l, P16Lds &lds, const Coords &history, bool &scan_done, uint8_t &hyst_lim, uint8_t &shift_lim)
{
    if (scan_done) return;
    Coords dc; coords_clear(dc);
    const int ss = c.scan_start, se = c.scan_end;
    const int margin = (uint16_t)(se - ss) / 40;
    if (coords_valid(history)) dc = history;
    else {
        dc.start = (int16_t)ss;
        bool state = lds.w.px[ss] > l.ref_level;
        for (int p = ss; p < ss + margin; p++) {
            if (!state) { if (lds.w.px[p] > l.ref_level) { dc.start = (int16_t)(p - 1); break; } }
            else { if (lds.w.px[p] < l.ref_level) { dc.start = (int16_t)(p - 1); break; } }
        }
        dc.stop = (int16_t)se;
        state = lds.w.px[se] > l.ref_level;
        for (int p = se; p > se - margin; p--) {
            if (!state) { if (lds.w.px[p] > l.ref_level) { dc.stop = (int16_t)(p + 1); break; } }
            else { if (lds.w.px[p] < l.ref_level) { dc.stop = (int16_t)(p + 1); break; } }
        }
    }
    const uint8_t in_hyst = hyst_lim, in_shift = shift_lim;
    search_pcm16_data(c, l, lds, dc, hyst_lim, shift_lim);
    hyst_lim = in_hyst; shift_lim = in_shift;
    scan_done = true;
}

/* ---- reference level sweep (MODE_INSANE only for this format, binarizer.cpp:1113-1120) -------------------------------------------
 * Binarizer::sweepRefLevel (:3551-3817) with a PCM16X0SubLine as the trial line: every level runs the whole coordinate search over the
 * three parts of the video line (scan_done is reset per level, :3704) and then reads the part the pass is for; the early return of
 * :3570 never fires (a fresh sub-line is PART_LEFT).  clear() through the PCMLine pointer is the base clear(): cells, picked bits and
 * Control Bit of the trial line persist from level to level.  The table of the sweep: see pcm1_bin_device.h (rs_store). */
__device__ inline void sweep_ref_level_p16(BinCtx &c, const Bin &b, uint8_t part, bool &scan_done, P16Lds &lds, const L16 &pcm_line, bool vl_doubled)
{
    Coords forced; calc_forced_coords(b, c.ps, forced);
    uint8_t low_lvl = (uint8_t)(pcm_line.black + 1), high_lvl = (uint8_t)(pcm_line.white - 1);
    if (c.ps.min_ref_lvl > low_lvl) low_lvl = c.ps.min_ref_lvl;
    if (c.ps.max_ref_lvl < high_lvl) high_lvl = c.ps.max_ref_lvl;
    L16 t; p16_clear(t);
    for (int lvl = (int)high_lvl; lvl >= (int)low_lvl; lvl--) {
        base_clear(t);
        if (c.scan_end > 0 && P16_BITS <= c.scan_end) { t.pixel_start = 0; t.pixel_stop = c.scan_end; }    /* setSourcePixels(0, size - 1) */
        t.coords.doubled = vl_doubled;
        t.black = low_lvl; t.white = high_lvl; t.ref_level = (uint8_t)lvl;
        uint8_t hyst_lim = 0, shift_lim = SHIFT_STAGES_SAFE;        /* calcRefLevelBySweep, :3847-3851 */
        if (!crc_valid(t)) {
            if (!coords_valid(forced)) { scan_done = false; find_pcm16_coordinates(c, t, lds, b.in_coord, scan_done, hyst_lim, shift_lim); }
            else { t.coords = forced; t.coords_set = true; }
            if (t.coords_set) read_pcm_data<true>(c, t, lds.w.px, part, hyst_lim, shift_lim);
        }
        if (t.picked_r != 0) t.hyst = (uint8_t)(t.hyst + HYST_DEPTH_MAX + 2);                            /* :3753-3765 */
        else if (t.picked_l != 0) t.hyst = (uint8_t)(t.hyst + HYST_DEPTH_MAX + 1);
        if (t.hyst > 0x0F) t.hyst = 0x0F;
        SweepEnt e = sweep_blank();
        e.result = REF_NO_PCM;
        if (crc_valid(t) && coords_valid(t.coords)) e.result = REF_CRC_OK;
        else if (t.coords_set) e.result = REF_BAD_CRC;
        if (e.result != REF_NO_PCM) {
            e.start = t.coords.start; e.stop = t.coords.stop; e.hyst = t.hyst; e.shift = t.shift; e.crc = t.calc_crc;
            if (lane_id() == 0) rs_store(lds.w, lvl, e);
        }
    }
}

/* Binarizer::calcRefLevelBySweep (binarizer.cpp:3821-4120), the branches a line without markers takes */
__device__ inline void calc_ref_level_by_sweep_p16(BinCtx &c, const Bin &b, uint8_t part, bool &scan_done, P16Lds &lds, L16 &l, bool vl_doubled,
                                                   uint8_t &hyst_lim, uint8_t &shift_lim)
{
    const int lane = lane_id();
    const uint8_t fast_ref = pick_center_ref_level(c.ps, l.black, l.white);
    const uint8_t blk1 = (uint8_t)(l.black + 1), wht1 = (uint8_t)(l.white - 1);
    hyst_lim = 0; shift_lim = SHIFT_STAGES_SAFE;
    SDV_WAVE_SYNC();
    { const SweepEnt z = sweep_blank(); for (int i = lane; i < 256; i += 64) rs_store(lds.w, i, z); }
    SDV_WAVE_SYNC();
    sweep_ref_level_p16(c, b, part, scan_done, lds, l, vl_doubled);
    rs_unpack(lds.w);
    uint8_t span_res = SPAN_NOT_FOUND, valid_crc_cnt = 0;
    if (lane == 0) {
        crc_stats_reset(lds.w, MAX_COLL_CRCS + 1); lds.w.crc_stats[0].hyst = 0; lds.w.crc_stats[0].shift = 0;
        for (uint8_t lv = wht1; lv > l.black; lv--)
            if (lds.w.sweep[lv].result == REF_CRC_OK) crc_stats_update(lds.w, lds.w.sweep[lv].crc, lds.w.sweep[lv].hyst, lds.w.sweep[lv].shift, valid_crc_cnt);
        const uint8_t first_cnt = valid_crc_cnt;
        if (valid_crc_cnt > 0) {
            crc_stats_most_frequent(lds.w, valid_crc_cnt);
            sweep_invalidate_non_frequent(lds.w, blk1, wht1, valid_crc_cnt, lds.w.crc_stats[0].crc);
        }
        lds.w.crc_stats[0].idx = (uint8_t)((first_cnt > 0 ? 1 : 0) | (valid_crc_cnt > 0 ? 2 : 0));
    }
    SDV_WAVE_SYNC();
    const bool had_any = (lds.w.crc_stats[0].idx & 1) != 0, still_valid = (lds.w.crc_stats[0].idx & 2) != 0;
    if (had_any && still_valid) {
        if (lds.w.crc_stats[0].result < c.ps.min_valid_crcs) span_res = SPAN_TOO_NARROW;
        else span_res = pick_level_by_crc_stats(lds.w, &l.ref_level, blk1, wht1, REF_CRC_OK, 0x0F, SHIFT_STAGES_MAX);
    }
    if (span_res == SPAN_OK) {
        const SweepEnt t = lds.w.sweep[l.ref_level];
        l.ref_sweeped = true;
        coords_set(l.coords, t.start, t.stop);
        l.coords_set = true;
        hyst_lim = t.hyst > HYST_DEPTH_MAX ? (uint8_t)HYST_DEPTH_MAX : t.hyst;
        shift_lim = t.shift;
    } else {
        if (span_res == SPAN_TOO_NARROW) {
            span_res = pick_level_by_crc_stats_opt(c.ps, lds.w, &l.ref_level, blk1, wht1, REF_CRC_OK, hyst_lim, shift_lim);
            l.forced_bad = true;
        } else span_res = pick_level_by_crc_stats(lds.w, &l.ref_level, blk1, wht1, REF_NO_PCM, 0xFF, 0xFF);      /* canUseMarkers() == false */
        if (span_res == SPAN_OK) {
            const SweepEnt t = lds.w.sweep[l.ref_level];
            coords_set(l.coords, t.start, t.stop);
            l.coords_set = true;
        } else if (is_ref_level_preset(b, c.ps)) {
            l.ref_level = b.in_ref;
            if (coords_valid(b.in_coord)) l.coords = b.in_coord;
        } else {
            l.ref_level = fast_ref;
            if (!coords_valid(b.in_coord)) coords_set(l.coords, (int16_t)c.scan_start, (int16_t)c.scan_end);
            else l.coords = b.in_coord;
        }
        hyst_lim = 0; shift_lim = SHIFT_STAGES_MIN;            /* HYST_DEPTH_MIN */
    }
    SDV_WAVE_SYNC();
}

/* findBlackWhite (binarizer.cpp:3116-3473) over the PCM-16x0 windows of the line (findPCM16X0BW, :2603-2681: one in each third) */
__device__ inline bool find_black_white_p16(const BinCtx &c, WaveLds &lds, L16 &line, bool &was_bw_scanned, bool sweep_flag)
{
    uint16_t pixel_limit = (uint16_t)(c.scan_end - c.scan_start);
    const uint16_t eighth = (uint16_t)(pixel_limit / 8);
    hist_clear(lds);
    uint16_t from = (uint16_t)(pixel_limit / 5);
    hist_add_range(lds, from, (uint16_t)(from + eighth));
    from = (uint16_t)((uint16_t)(eighth * 4) + eighth / 2);
    hist_add_range(lds, from, (uint16_t)(from + eighth));
    const uint16_t to = (uint16_t)(c.scan_end - pixel_limit / 64);
    hist_add_range(lds, (uint16_t)(to - eighth), to);

    const BwLevels bw = bw_from_spread(c.ps, spread_levels(c.ps, lds), sweep_flag);      /* Binarizer::do_ref_lvl_sweep: left by the last line that got as far as :1104 */
    was_bw_scanned = true;
    line.black = bw.black; line.white = bw.white;
    line.bw_set = bw.set;
    return line.bw_set;
}

/* Stage STG_INPUT_ALL of processLine alone (binarizer.cpp:774-931), for the lean build of the frame kernel: a part whose reference level and
 * coordinates are preset - its levels measured first when they are not (the part behind one that was marked bad) - is read with them, ladder and
 * Bit Picker included; true when that ends in STG_DATA_OK, i.e. the part is done.  false: the part needs the stages behind it (nothing is decided,
 * `out` is not to be used) - the frame goes to the full build. */
__device__ inline bool input_all_p16(const BinCtx &c, const Bin &b, uint8_t part, WaveLds &lds, L16 &out, bool vl_doubled)
{
    if (c.ps.en_force_coords) return false;
    if (!(is_ref_level_preset(b, c.ps) && coords_valid(b.in_coord))) return false;
    p16_clear(out);
    out.line_part = part == PART_MIDDLE ? 1 : (part == PART_RIGHT ? 2 : 0);
    out.coords.doubled = vl_doubled;
    if (c.scan_end > c.scan_start && P16_BITS <= (c.scan_end - c.scan_start)) { out.pixel_start = c.scan_start; out.pixel_stop = c.scan_end; }
    bool was_bw_scanned = false;
    if (are_bw_levels_preset(b, c.ps)) { out.black = b.in_black; out.white = b.in_white; out.bw_set = true; }
    if (!out.bw_set) find_black_white_p16(c, lds, out, was_bw_scanned, b.do_ref_lvl_sweep);
    out.coords = b.in_coord;
    out.ref_level = b.in_ref;
    if (!out.bw_set) return false;
    if (b.in_ref >= out.white || b.in_ref <= out.black) return false;
    read_pcm_data<true>(c, out, lds.px, part, b.in_max_hyst, b.in_max_shift);
    if (!crc_valid(out)) return false;
    out.by_ext_tune = true;
    if (out.forced_bad) return false;
    out.coords_set = true;
    return true;
}

/* Binarizer::processLine (binarizer.cpp:443-1724), PCM16X0SubLine output, one part of the video line staged in lds.w.px */
template <bool kInsane>          /* the build for MODE_INSANE: see process_line_p1 */
__device__ inline void process_line_p16(BinCtx &c, Bin &b, bool coord_search, uint8_t part, bool &scan_done, P16Lds &lds, L16 &out, bool vl_doubled)
{
    p16_clear(out);
    out.line_part = part == PART_MIDDLE ? 1 : (part == PART_RIGHT ? 2 : 0);
    out.coords.doubled = vl_doubled;
    if (c.scan_end > c.scan_start && P16_BITS <= (c.scan_end - c.scan_start)) { out.pixel_start = c.scan_start; out.pixel_stop = c.scan_end; }
    coords_set(out.coords, (int16_t)c.scan_start, (int16_t)c.scan_end);
    Coords forced; calc_forced_coords(b, c.ps, forced);
    if (c.ps.en_force_coords && coords_valid(forced)) { out.coords = forced; out.coords_set = true; }
    uint8_t state = STG_REF_FIND;
    bool was_bw_scanned = false;
    if (are_bw_levels_preset(b, c.ps)) { out.black = b.in_black; out.white = b.in_white; out.bw_set = true; }
    if (is_ref_level_preset(b, c.ps)) state = coords_valid(b.in_coord) ? STG_INPUT_ALL : STG_INPUT_LEVEL;
    uint8_t hyst_lim = b.in_max_hyst, shift_lim = b.in_max_shift;

    for (int stage_count = 1; ; stage_count++) {
        if (state == STG_INPUT_ALL) {                           /* :774-931 */
            if (!out.bw_set) find_black_white_p16(c, lds.w, out, was_bw_scanned, b.do_ref_lvl_sweep);
            if (!coords_valid(forced)) out.coords = b.in_coord;
            out.ref_level = b.in_ref;
            if (!out.bw_set) state = STG_NO_GOOD;
            else if (b.in_ref >= out.white || b.in_ref <= out.black) state = STG_REF_FIND;
            else {
                read_pcm_data<true>(c, out, lds.w.px, part, hyst_lim, shift_lim);
                if (crc_valid(out)) { out.by_ext_tune = true; state = STG_DATA_OK; } else state = STG_REF_FIND;
            }
        } else if (state == STG_INPUT_LEVEL) {                  /* :932-1072 */
            if (!was_bw_scanned) find_black_white_p16(c, lds.w, out, was_bw_scanned, b.do_ref_lvl_sweep);
            if (!coords_valid(forced)) coords_set(out.coords, (int16_t)c.scan_start, (int16_t)c.scan_end);
            out.ref_level = b.in_ref;
            state = out.bw_set ? STG_REF_FIND : STG_NO_GOOD;
        } else if (state == STG_REF_FIND) {                     /* :1073-1390 */
            if (!was_bw_scanned) find_black_white_p16(c, lds.w, out, was_bw_scanned, b.do_ref_lvl_sweep);
            if (!out.bw_set) state = STG_NO_GOOD;
            else if ((b.do_ref_lvl_sweep = kInsane)) {          /* = (mode == MODE_INSANE), :1113-1133: the member keeps this until the next line gets here; STG_REF_SWEEP_RUN, :1391-1400 */
                if (kInsane) calc_ref_level_by_sweep_p16(c, b, part, scan_done, lds, out, vl_doubled, hyst_lim, shift_lim);
                state = STG_READ_PCM;
            } else {
                hyst_lim = HYST_DEPTH_SAFE; shift_lim = SHIFT_STAGES_MIN;
                state = STG_READ_PCM;
                out.ref_level = pick_center_ref_level(c.ps, out.black, out.white);
                if (coords_valid(forced)) { out.coords = forced; out.coords_set = true; }
                else {
                    if (!coords_valid(b.in_coord)) coords_set(out.coords, (int16_t)c.scan_start, (int16_t)c.scan_end);
                    else out.coords = b.in_coord;
                    if (c.ps.en_coord_search && coord_search) {
                        /* the search reads all three parts through the one line object and leaves the part mode as it found it */
                        find_pcm16_coordinates(c, out, lds, b.in_coord, scan_done, hyst_lim, shift_lim);
                    }
                }
                if (!out.coords_set) {                          /* :1301-1320 */
                    if (c.mode == SDV_MODE_DRAFT) { hyst_lim = 2; shift_lim = SHIFT_STAGES_MIN; }
                    else { hyst_lim = HYST_DEPTH_SAFE; shift_lim = SHIFT_STAGES_SAFE; }
                } else { hyst_lim = b.in_max_hyst; shift_lim = SHIFT_STAGES_SAFE; }
            }
        } else if (state == STG_READ_PCM) {                     /* :1401-1533 */
            if (coords_valid(forced)) { hyst_lim = HYST_DEPTH_SAFE; shift_lim = SHIFT_STAGES_MIN; }
            if (out.coords_set) read_pcm_data<true>(c, out, lds.w.px, part, hyst_lim, shift_lim);
            if (crc_valid(out)) state = STG_DATA_OK;
            if (state != STG_DATA_OK) {
                if (coords_valid(b.in_coord) && !coords_valid(forced) && !b.do_ref_lvl_sweep && !out.forced_bad && !out.coords_set) {
                    if (coords_ne(out.coords, b.in_coord)) {
                        out.coords = b.in_coord;
                        read_pcm_data<true>(c, out, lds.w.px, part, hyst_lim, shift_lim);
                        if (crc_valid(out)) state = STG_DATA_OK;
                    }
                }
                if (state != STG_DATA_OK) state = STG_NO_GOOD;
            }
        } else if (state == STG_DATA_OK) {                      /* :1534-1621 */
            if (out.forced_bad) state = STG_NO_GOOD;
            else { out.coords_set = true; break; }              /* :1568-1574 */
        } else {                                                /* STG_NO_GOOD, :1622-1669 */
            if (crc_valid(out)) set_invalid_crc(out);
            break;
        }
        if (stage_count > STG_MAX) break;
    }
}

__device__ inline void emit_rec(const L16 &l, uint32_t frame, uint16_t line_no, bool from_doubled, sdv_pcm16x0_bin_rec *dst)
{
    if (lane_id() != 0) return;
    sdv_pcm16x0_bin_rec r;
    r.frame_number = frame; r.line_number = line_no;
    for (int k = 0; k < 4; k++) r.words[k] = get_word(l, k);
    r.calc_crc = l.calc_crc;
    r.data_start = l.coords.start; r.data_stop = l.coords.stop;
    r.queue_order = l.queue_order;
    r.black_level = l.black; r.white_level = l.white; r.ref_low = l.ref_low; r.ref_level = l.ref_level; r.ref_high = l.ref_high;
    r.hysteresis_depth = l.hyst; r.shift_stage = l.shift; r.service_type = l.service;
    r.picked_bits_left = l.picked_l; r.picked_bits_right = l.picked_r;
    r.flags = (uint8_t)((l.ref_sweeped ? SDV_LF_REF_SWEEPED : 0) | (l.coords_sweeped ? SDV_LF_COORDS_SWEEPED : 0) | (l.by_ext_tune ? SDV_LF_BY_EXT_TUNE : 0) | (l.bw_set ? SDV_LF_BW_SET : 0) |
                        (l.coords_set ? SDV_LF_COORDS_SET : 0) | (l.forced_bad ? SDV_LF_FORCED_BAD : 0) | (crc_valid(l) ? SDV_LF_CRC_VALID : 0) |
                        (from_doubled ? SDV_LF_FROM_DOUBLED : 0));
    r.line_part = l.line_part; r.control_bit = l.control_bit ? 1 : 0; r._pad = 0;
    *dst = r;
}

/* ---- sdv_pcm16x0_binarize_lines: the three passes over a video line, one wave per line ------------------------------------------------ */
struct LineArgs16 {
    const uint8_t *luma; size_t row_stride; int width; size_t n_lines;
    const sdv_bin_state *states;    /* [3 * n_lines]: what the caller has preset before each pass (setGoodParameters / setBWLevels ...), or NULL: nothing */
    uint32_t frame_number; uint16_t first_line, line_step;
    uint8_t doubled, mode, coord_search;
    sdv_bin_preset preset;
    sdv_pcm16x0_bin_rec *out;       /* [3 * n_lines] */
    uint8_t *scan_done;             /* [3 * n_lines] VideoLine::scan_done behind each pass, or NULL */
};
/* The reference runs its Binarizer three times over one VideoLine object (setLinePartMode, videotodigital.cpp:902-925); what the
 * passes share is the line's scan_done mark (binarizer.cpp:5819-6042: a line whose coordinate search has run is not searched again). */
template <bool kInsane>
__device__ inline void line16_body(const LineArgs16 &a, P16Lds &lds, size_t li)
{
    sdvp1b::stage_row(lds.w.px, a.luma + li * a.row_stride, a.width);
    bool scan_done = false;
    for (int sub = 0; sub < P16_SUBLINES; sub++) {
        BinCtx c; c.ps = a.preset; c.mode = a.mode; c.scan_start = 0; c.scan_end = (uint16_t)(a.width - 1);
        c.force_bit_picker = true;      /* binarizer.cpp:82 */
        Bin b;
        b.in_black = b.in_white = b.in_ref = 0; coords_clear(b.in_coord);
        b.do_ref_lvl_sweep = false;
        if (a.states) {
            const sdv_bin_state s = a.states[3 * li + (size_t)sub];
            b.in_black = s.in_def_black; b.in_white = s.in_def_white; b.in_ref = s.in_def_reference;
            b.in_coord.start = s.in_def_start; b.in_coord.stop = s.in_def_stop; b.in_coord.doubled = s.in_def_from_doubled != 0;
            b.do_ref_lvl_sweep = s.do_ref_lvl_sweep != 0;
        }
        bin_set_mode(b, a.mode);
        b.scan_start = c.scan_start; b.scan_end = c.scan_end; b.vl_doubled = a.doubled != 0;
        L16 out;
        process_line_p16<kInsane>(c, b, a.coord_search != 0, (uint8_t)(PART_LEFT + sub), scan_done, lds, out, a.doubled != 0);
        emit_rec(out, a.frame_number, (uint16_t)(a.first_line + li * a.line_step), a.doubled != 0, &a.out[3 * li + (size_t)sub]);
        if (a.scan_done && lane_id() == 0) a.scan_done[3 * li + (size_t)sub] = scan_done ? 1 : 0;
        SDV_WAVE_SYNC();
    }
}

} // namespace sdvp16

#ifndef SDV_P16B_WAVES_PER_EU
#define SDV_P16B_WAVES_PER_EU 4
#endif
__global__ void __launch_bounds__(64, SDV_P16B_WAVES_PER_EU) sdv_k_pcm16_lines(sdvp16::LineArgs16 a)
{
    __shared__ sdvp16::P16Lds lds;
    for (size_t li = blockIdx.x; li < a.n_lines; li += gridDim.x) sdvp16::line16_body<false>(a, lds, li);
}
__global__ void __launch_bounds__(64, SDV_P16B_WAVES_PER_EU) sdv_k_pcm16_lines_insane(sdvp16::LineArgs16 a)      /* MODE_INSANE */
{
    __shared__ sdvp16::P16Lds lds;
    for (size_t li = blockIdx.x; li < a.n_lines; li += gridDim.x) sdvp16::line16_body<true>(a, lds, li);
}
