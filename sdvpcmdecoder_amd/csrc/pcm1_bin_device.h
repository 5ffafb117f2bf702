/*
 * pcm1_bin_device.h - PCM-1 (Sony Standard B) line binarizer: Binarizer::processLine with a PCM1Line output
 * (front half), one wavefront per video line.
 *
 * Reference: binarizer.cpp:443-1724 (processLine, PCM-1 paths), :2560-2600 (findPCM1BW), :5601-5813 (findPCM1Coordinates),
 * :4123-4511 (searchPCM1Data), :7016-7131 (fillPCM1), :6116-6596 (pickCutBitsUpPCM1), :7560-7650 (fillDataWords),
 * :7695-8055 (readPCMdata); pcm1line.cpp, pcmline.cpp for the line object.
 *
 * What runs where.  The stage machine, the black/white search and every single read of the line are wave-uniform (all lanes
 * compute the same values; the histogram is built by all lanes).  The marker-less coordinate search - 25 x 25 candidate
 * coordinate pairs, each a full readPCMdata with the Bit Picker forced - is spread over the lanes, one candidate per lane at a
 * time, results in LDS; the vote over them (per left coordinate, then over the left coordinates) is serial on lane 0.
 * The reference evaluates the candidates one after the other on ONE line object, and two things survive from one candidate to
 * the next: the forced-bad mark a Bit Picker collision leaves (every later candidate then fails), and - after the search -
 * whatever the last candidate left in the object.  Both are reproduced: the first collision in search order is found with a wave
 * minimum, and the last candidate is evaluated once more, wave-uniformly, into the line.
 *
 * The 94 cells of a line are kept as one 94-bit number (word k = bits 81-13k .. 93-13k, CRCC = the low 16 bits).
 */
#pragma once
#include "stc007_device.h"

namespace sdvp1b {
using namespace sdv;
typedef unsigned __int128 u128;

enum { P1_BITS = 94, P1_WORD_BITS = 13, P1_CRC_BITS = 16, P1_CRC_SILENT = 0xECBF };
enum { P1_SEARCH_STEP_DIV = 4, P1_SEARCH_MAX_OFS = 12, P1_SEARCH_STEP_CNT = (P1_SEARCH_MAX_OFS + 1) * 2, P1_GRID = 2 * P1_SEARCH_MAX_OFS + 1 };   /* binarizer.h:254-256 */
#ifndef SDV_P1_SHARED_READS
#define SDV_P1_SHARED_READS 1      /* the coordinate search reads every cell pattern once (search_pcm1_data) */
#endif
enum { P1_LEFT_BASE = 32 };         /* where the left-coordinate results sit in WaveLds::sweep (the right ones use 0..25) */

struct LineArgs1 {
    const uint8_t *luma; size_t row_stride; int width; size_t n_lines;
    const sdv_bin_state *states;    /* what the caller has preset per line (setGoodParameters / setBWLevels ...), or NULL: nothing */
    uint32_t frame_number; uint16_t first_line, line_step;
    uint8_t doubled, mode, coord_search;
    sdv_bin_preset preset;
    sdv_pcm1_bin_rec *out;
    int *list;                      /* lines the lean kernel handed on (it appends; the full kernel works the list off), or NULL: all lines */
    int *counters;                  /* [0] entries in list, [1] next entry to take */
};

struct P1Lds {
    WaveLds w;
    uint32_t grid[P1_GRID * P1_GRID];       /* per candidate: crc | hyst << 16 | shift << 20 | valid << 24 */
    CrcStat lstats[MAX_COLL_CRCS + 1];      /* scan_left_crcs */
    int32_t vote[4];                        /* found, data_start, data_stop */
};

struct L1 {                         /* PCM1Line : PCMLine (pcmline.h:137-166, pcm1line.h:59-110) */
    uint8_t black, white, ref_low, ref_level, ref_high, hyst, shift;
    Coords coords;
    bool ref_sweeped, coords_sweeped, by_ext_tune, bw_set, coords_set, forced_bad;
    uint8_t service;
    uint16_t pixel_start, pixel_stop;
    int16_t pso; uint32_t psm, hpsm;
    u128 v;                         /* the 94 cells */
    uint16_t calc_crc;
    uint8_t picked_l, picked_r;
};

__device__ inline u128 word_at(uint16_t w, int k) { return (u128)(w & 0x1FFF) << (81 - 13 * k); }
__device__ inline uint16_t get_word(const L1 &l, int k) { return k == 6 ? (uint16_t)(l.v & 0xFFFF) : (uint16_t)((l.v >> (81 - 13 * k)) & 0x1FFF); }
__device__ inline void set_word0(L1 &l, uint16_t w) { l.v = (l.v & ~word_at(0x1FFF, 0)) | word_at(w, 0); }                       /* PCM1Line::setWord(WORD_L2) */
__device__ inline void set_crcc(L1 &l, uint16_t w) { l.v = (l.v & ~(u128)0xFFFF) | (u128)w; }                                   /* PCM1Line::setWord(WORD_CRCC) */

/* the same CRC as a GF(2)-linear map of the cells (cell b = bit b of lo for b < 64, bit b-64 of hi): parity masks per CRC bit */
struct Crc1Tables { uint64_t klo[16], khi[16]; uint16_t base; uint16_t col[4]; };      /* col[i]: what cell i of the line flips in the CRC (the Bit Picker's left bits) */
constexpr Crc1Tables make_crc1_tables()
{
    Crc1Tables t{};
    uint16_t c = 0xFFFF;
    for (int i = 0; i < 78; i++) c = crc16_step(c, 1);             /* all cells 0: the CRC runs over their inverse */
    t.base = (uint16_t)~c;
    for (int b = 0; b < 78; b++) {
        uint16_t v = 0;
        for (int i = 0; i < 78; i++) v = crc16_step(v, i == b);
        for (int j = 0; j < 16; j++)
            if (v & (1u << j)) { if (b < 64) t.klo[j] |= (1ull << b); else t.khi[j] |= (1ull << (b - 64)); }
        if (b < 4) t.col[b] = v;
    }
    return t;
}
#ifdef SDV_EMU
static const Crc1Tables c_crc1 = make_crc1_tables();
#else
__device__ __constant__ const Crc1Tables c_crc1 = make_crc1_tables();
#endif
/* PCM1Line::calcCRC (pcm1line.cpp:158-171): CRC-16/CCITT over the 78 inverted data cells, result inverted - computed through the
 * parity masks above (one lane on its own) */
__device__ inline void cell_masks(u128 v, uint64_t &s_lo, uint64_t &s_hi)       /* cell b = bit b of s_lo (b < 64) / bit b - 64 of s_hi */
{
    s_lo = __brevll((uint64_t)(v >> 30));
    s_hi = __brevll((uint64_t)(v & (u128)0x3FFFFFFFull) << 34);
}
__device__ inline uint16_t crc_of_masks(uint64_t s_lo, uint64_t s_hi)
{
    uint32_t crc = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) crc |= (uint32_t)((__popcll(s_lo & c_crc1.klo[j]) + __popcll(s_hi & c_crc1.khi[j])) & 1) << j;
    return (uint16_t)(crc ^ c_crc1.base);
}
__device__ inline void calc_crc(L1 &l)
{
    uint64_t s_lo, s_hi;
    cell_masks(l.v, s_lo, s_hi);
    l.calc_crc = crc_of_masks(s_lo, s_hi);
}
__device__ inline bool has_header(const L1 &l)     /* pcm1line.cpp:314-323 */
{
    const u128 hdr = word_at(0x0666, 0) | word_at(0x0CCC, 1) | word_at(0x1999, 2) | word_at(0x1333, 3) | word_at(0x0666, 4) | word_at(0x0CCC, 5) | (u128)0xCCCC;
    return l.v == hdr;
}
__device__ inline bool crc_valid_ignore_forced(const L1 &l) { return l.calc_crc == (uint16_t)(l.v & 0xFFFF) || has_header(l); }
__device__ inline bool crc_valid(const L1 &l) { return !l.forced_bad && crc_valid_ignore_forced(l); }
__device__ inline void set_invalid_crc(L1 &l) { set_crcc(l, (uint16_t)~l.calc_crc); }
__device__ inline void set_silent(L1 &l)
{
    l.v = (l.v & (u128)0xFFFF) | word_at(0x1000, 0) | word_at(0x1000, 1) | word_at(0x1000, 2) | word_at(0x1000, 3) | word_at(0x1000, 4) | word_at(0x1000, 5);
    calc_crc(l);
}
__device__ inline void base_clear(L1 &l)           /* PCMLine::clear, pcmline.cpp:96-116 */
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
__device__ inline void p1_clear(L1 &l)             /* PCM1Line::clear, pcm1line.cpp:57-77 */
{
    base_clear(l);
    l.picked_l = l.picked_r = 0;
    /* setSilent() + the CRC of a silent line (CRC_SILENT, pcm1line.h:98) + setInvalidCRC(), without running the CRC */
    l.v = word_at(0x1000, 0) | word_at(0x1000, 1) | word_at(0x1000, 2) | word_at(0x1000, 3) | word_at(0x1000, 4) | word_at(0x1000, 5);
    l.calc_crc = P1_CRC_SILENT;
    set_invalid_crc(l);
}
__device__ inline void set_ppb(L1 &l, const Coords &c)     /* pcmline.cpp:506-519 with 94 cells between the coordinates */
{
    l.psm = (uint32_t)((int)c.stop - (int)c.start);
    l.psm = (l.psm * 128u + P1_BITS / 2) / P1_BITS;
    l.pso = c.start;
    l.hpsm = (l.psm + 1) / 2;
}
__device__ inline uint8_t get_ppb(const L1 &l) { return (uint8_t)(l.psm / 128u); }
/* getVideoPixeBylCalc (pcmline.cpp:249-311): both shift tables are {0, +1, -1, +2, -2}, so the shift is uniform along the line */
__device__ inline int pixel_of(const L1 &l, int bit, int stage)
{
    int32_t vp = (int32_t)((uint32_t)bit * l.psm + l.hpsm);
    vp = vp / 128 + l.pso;
    const int sh = stage == 0 ? 0 : (stage == 1 ? 1 : (stage == 2 ? -1 : (stage == 3 ? 2 : -2)));
    vp += sh;
    if (vp < (int32_t)l.pixel_start) vp = l.pixel_start;
    else if (vp >= (int32_t)l.pixel_stop) vp = (int32_t)l.pixel_stop - 1;
    return vp;
}

/* fillPCM1 (binarizer.cpp:7016-7131): the two-level automaton over the 94 cell centres, one lane on its own: the two comparisons of every
 * cell are collected as masks and the automaton is solved on them (solve_automaton), the CRC comes from the parity masks */
__device__ inline void fill_pcm1(L1 &l, const uint8_t *px_row, int stage)
{
    int32_t acc = (int32_t)l.hpsm + ((int32_t)l.pso + shift_of_stage(stage)) * 128;
    const int32_t lo = l.pixel_start, hi = (int32_t)l.pixel_stop - 1;
    uint32_t a0, a1, a2, b0, b1, b2;
    compare_cells32<32>(px_row, acc, (int32_t)l.psm, lo, hi, l.ref_low, l.ref_high, a0, b0);
    compare_cells32<32>(px_row, acc, (int32_t)l.psm, lo, hi, l.ref_low, l.ref_high, a1, b1);
    compare_cells32<P1_BITS - 64>(px_row, acc, (int32_t)l.psm, lo, hi, l.ref_low, l.ref_high, a2, b2);
    uint64_t s_lo, s_hi;
    solve_automaton_lane((uint64_t)a0 | ((uint64_t)a1 << 32), (uint64_t)a2, (uint64_t)b0 | ((uint64_t)b1 << 32), (uint64_t)b2, s_lo, s_hi);
    s_hi &= (1ull << (P1_BITS - 64)) - 1ull;
    l.v = ((u128)__brevll(s_lo) << 30) | (u128)(__brevll(s_hi) >> 34);
    l.calc_crc = crc_of_masks(s_lo, s_hi);
}

/* the same for the whole wave (wave-uniform callers only): lane i samples cells i and i+64, the automaton is solved on the ballots
 * (stc007_device.h, solve_automaton), the CRC is 16 parities */
__device__ inline void fill_pcm1_wave(L1 &l, const uint8_t *px_row, int stage)
{
    const int lane = lane_id();
    const uint8_t p0 = px_row[pixel_of(l, lane, stage)];
    const bool second = lane + 64 < P1_BITS;
    const uint8_t p1 = px_row[pixel_of(l, second ? lane + 64 : P1_BITS - 1, stage)];
    const uint64_t a_lo = __ballot(p0 > l.ref_low), b_lo = __ballot(p0 >= l.ref_high);
    const uint64_t a_hi = __ballot(second && p1 > l.ref_low), b_hi = __ballot(second && p1 >= l.ref_high);
    uint64_t s_lo, s_hi;
    solve_automaton(a_lo, a_hi, b_lo, b_hi, s_lo, s_hi);
    s_hi &= (1ull << (P1_BITS - 64)) - 1ull;
    l.v = ((u128)__brevll(s_lo) << 30) | (u128)(__brevll(s_hi) >> 34);
    const uint64_t klo = (lane < 16) ? c_crc1.klo[lane & 15] : 0ull, khi = (lane < 16) ? c_crc1.khi[lane & 15] : 0ull;
    const int par = (__popcll(s_lo & klo) + __popcll(s_hi & khi)) & 1;
    const uint64_t cb = __ballot(par);
    l.calc_crc = (uint16_t)((uint16_t)(cb & 0xFFFF) ^ c_crc1.base);
}

struct BinCtx { sdv_bin_preset ps; uint8_t mode; uint16_t scan_start, scan_end; bool force_bit_picker; };

/* the Bit Picker's search as the reference runs it (binarizer.cpp:6418-6583): every combination of the cut-off bits in turn.  Only taken
 * when a Header line is among the combinations (its words are valid whatever the CRC says); pick_cut_bits below does the rest by linearity */
__device__ inline void pick_cut_bits_by_trial(L1 &l, int left_bits, int right_bits)
{
    bool patch_found = false, coll_lock = false;
    const uint32_t left_lim = left_bits ? (1u << left_bits) : 1u, right_lim = right_bits ? (1u << right_bits) : 1u;
    const uint16_t left_orig = get_word(l, 0), right_orig = get_word(l, 6);
    const uint16_t left_clean = left_bits ? (uint16_t)(left_orig & (uint16_t)~((left_lim - 1) << (P1_WORD_BITS - left_bits))) : left_orig;
    const uint16_t right_clean = right_bits ? (uint16_t)(right_orig & (uint16_t)~(right_lim - 1)) : right_orig;
    uint16_t left_fix = 0, right_fix = 0;
    /* every value of the cut-off bits; exactly one may give a valid CRC.  With only one side cut the other loop runs once and
     * leaves its word alone, which is the reference's single-sided loop (:6418-6583) */
    for (uint32_t li = 0; li < left_lim && !coll_lock; li++) {
        for (uint32_t ri = 0; ri < right_lim; ri++) {
            const uint16_t lp = left_bits ? (uint16_t)(li << (P1_WORD_BITS - left_bits)) : 0, rp = (uint16_t)ri;
            if (left_bits) set_word0(l, (uint16_t)(left_clean | lp));
            if (right_bits) set_crcc(l, (uint16_t)(right_clean | rp));
            calc_crc(l);
            if (crc_valid(l)) {
                if (patch_found) { coll_lock = true; break; }
                patch_found = true; left_fix = lp; right_fix = rp;
            }
        }
    }
    if (coll_lock || !patch_found) {
        if (left_bits) set_word0(l, left_orig);
        if (right_bits) set_crcc(l, right_orig);
        calc_crc(l);
        if (coll_lock) l.forced_bad = true;
        return;
    }
    if (left_bits) set_word0(l, (uint16_t)(left_clean | left_fix));
    if (right_bits) set_crcc(l, (uint16_t)(right_clean | right_fix));
    calc_crc(l);
    l.picked_l = (uint8_t)left_bits; l.picked_r = (uint8_t)right_bits;
}

/* pickCutBitsUpPCM1 (binarizer.cpp:6116-6596) */
/* (forced in line, like process_line_p1 below: as functions of their own they take the line and the Binarizer by address - and an object whose address is
 * taken lives in scratch memory, all of it, in every kernel that calls them) */
__device__ __forceinline__ void pick_cut_bits(const BinCtx &c, L1 &l)
{
    int left_bits = 0, right_bits = 0;
    l.picked_l = l.picked_r = 0;
    int max_cut = c.ps.left_bit_pick; if (c.mode == SDV_MODE_DRAFT) max_cut /= 2;
    int first = c.scan_start;
    const int half_ppb = ((int)get_ppb(l) + 1) / 2;
    for (int i = 0; i < max_cut; i++) {
        const int cur = pixel_of(l, i, 0);
        if ((cur - first) >= half_ppb) break;
        if (i == 0) first = cur;
        left_bits = i + 1;
    }
    first = c.scan_end;
    max_cut = c.ps.right_bit_pick; if (c.mode == SDV_MODE_DRAFT) max_cut /= 2;
    for (int i = 0; i < max_cut; i++) {
        const int cur = pixel_of(l, P1_BITS - 1 - i, 0);
        if ((first - cur) >= half_ppb) break;
        if (i == 0) first = cur;
        right_bits = i + 1;
    }
    if (c.force_bit_picker && crc_valid(l)) { l.picked_l = (uint8_t)left_bits; l.picked_r = (uint8_t)right_bits; return; }
    if (left_bits == 0 && right_bits == 0) return;
    if (l.forced_bad) return;           /* nothing reads valid on a line that is forced bad: the search would put everything back */
    /* Every value of the cut-off bits; exactly one may give a valid CRC (:6418-6583).  The CRC is linear in the cells: the left bits
     * (the first cells of the line) each flip a fixed pattern of CRC bits, the right bits are the low bits of the CRCC as read - so a
     * left value fits iff the CRC it gives agrees with the CRCC outside the cut-off bits, and then with exactly one right value. */
    const u128 hdr = word_at(0x0666, 0) | word_at(0x0CCC, 1) | word_at(0x1999, 2) | word_at(0x1333, 3) | word_at(0x0666, 4) | word_at(0x0CCC, 5) | (u128)0xCCCC;
    const u128 cut = (left_bits ? (((u128)((1u << left_bits) - 1)) << (P1_BITS - left_bits)) : (u128)0) | (u128)(right_bits ? ((1u << right_bits) - 1) : 0);
    if (((l.v ^ hdr) & ~cut) == 0) { pick_cut_bits_by_trial(l, left_bits, right_bits); return; }      /* a Header in reach: valid whatever its CRC */
    const uint16_t right_orig = get_word(l, 6), right_mask = right_bits ? (uint16_t)((1u << right_bits) - 1) : 0, right_clean = (uint16_t)(right_orig & ~right_mask);
    const u128 data_clean = l.v & ~(cut & ~(u128)0xFFFF);
    uint64_t s_lo, s_hi;
    cell_masks(data_clean, s_lo, s_hi);
    const uint16_t base = crc_of_masks(s_lo, s_hi);
    const uint32_t left_lim = left_bits ? (1u << left_bits) : 1u;
    int found = 0; uint32_t li_fix = 0; uint16_t crc_fix = 0;
    for (uint32_t li = 0; li < left_lim; li++) {
        uint16_t crc = base;
#pragma unroll
        for (int i = 0; i < 4; i++) if (i < left_bits && ((li >> (left_bits - 1 - i)) & 1u)) crc ^= c_crc1.col[i];       /* the top bit of the value is the first cell */
        if ((uint16_t)(crc & ~right_mask) == right_clean) { if (found) { found = 2; break; } found = 1; li_fix = li; crc_fix = crc; }
    }
    if (found == 2) { l.forced_bad = true; return; }        /* two values fit: the line is put back as it was and marked */
    if (found == 0) return;
    l.v = data_clean | (left_bits ? ((u128)li_fix << (P1_BITS - left_bits)) : (u128)0);
    set_crcc(l, (uint16_t)(right_clean | (crc_fix & right_mask)));
    l.calc_crc = crc_fix;
    l.picked_l = (uint8_t)left_bits; l.picked_r = (uint8_t)right_bits;
}

/* fillDataWords (binarizer.cpp:7560-7650); false = the levels clip (STG_NO_GOOD) */
template <bool kWave>
__device__ inline bool fill_data_words(const BinCtx &c, L1 &l, const uint8_t *px_row, uint8_t ref_delta, uint8_t shift_stg)
{
    if (ref_delta > HYST_DEPTH_MAX || shift_stg > SHIFT_STAGES_MAX) return false;
    const uint8_t low_ref = get_low_level(l.ref_level, ref_delta), high_ref = get_high_level(l.ref_level, ref_delta);
    l.ref_low = low_ref; l.ref_high = high_ref;
    if (low_ref <= l.black) { set_invalid_crc(l); return false; }
    if (high_ref >= l.white) { set_invalid_crc(l); return false; }
    l.hyst = ref_delta; l.shift = shift_stg;
    if (kWave) fill_pcm1_wave(l, px_row, shift_stg); else fill_pcm1(l, px_row, shift_stg);
    if ((!crc_valid(l) && (l.ref_level > c.ps.min_white_lvl) && ((c.ps.left_bit_pick != 0) || (c.ps.right_bit_pick != 0))) || c.force_bit_picker)
        pick_cut_bits(c, l);
    return true;
}

/* readPCMdata (binarizer.cpp:7695-8055).  A line whose reference level was not swept: hysteresis depths from 0 up, pixel shift
 * stages from 0 up, the first combination with a valid CRC wins (both loops of the reference stop at the first valid CRC, so its
 * two votes are over one entry each); none: depth 0, stage 0.  Then the final fill. */
template <bool kWave>
__device__ inline void read_pcm_data(const BinCtx &c, L1 &l, const uint8_t *px_row, uint8_t hyst_lim, uint8_t shift_lim)
{
    set_ppb(l, l.coords);
    if (hyst_lim > HYST_DEPTH_MAX) hyst_lim = HYST_DEPTH_MAX;
    if (shift_lim > SHIFT_STAGES_MAX) shift_lim = SHIFT_STAGES_MAX;
    if (l.ref_sweeped) { fill_data_words<kWave>(c, l, px_row, hyst_lim, shift_lim); return; }     /* isDataByRefSweep(): the sweep has picked depth and stage (:7741) */
    bool found = false;
    /* what the very first fill (depth 0, stage 0) leaves: when nothing reads valid the reference's final fill is that fill again, and it
     * comes out the same unless the forced-bad mark changed in between (the Bit Picker looks at it) - then it is run again */
    const bool entry_forced = l.forced_bad;
    bool kept = false;
    u128 k_v = 0; uint16_t k_crc = 0; uint8_t k_pl = 0, k_pr = 0, k_lo = 0, k_hi = 0;
    for (uint8_t h = 0; h <= hyst_lim && !found; h++) {
        bool invalid_hyst = false;
        for (uint8_t s = 0; s <= shift_lim; s++) {
            if (!fill_data_words<kWave>(c, l, px_row, h, s)) { invalid_hyst = true; break; }
            if (crc_valid(l)) { found = true; break; }
            if (h == 0 && s == 0 && c.force_bit_picker) { kept = true; k_v = l.v; k_crc = l.calc_crc; k_pl = l.picked_l; k_pr = l.picked_r; k_lo = l.ref_low; k_hi = l.ref_high; }
        }
        if (invalid_hyst) break;
    }
    /* the final fill of the reference repeats the fill that was found, on the same inputs, and leaves the line as that fill left
     * it - it only has to be reproduced when nothing was found */
    if (!found) {
        if (kept && l.forced_bad == entry_forced) { l.v = k_v; l.calc_crc = k_crc; l.picked_l = k_pl; l.picked_r = k_pr; l.ref_low = k_lo; l.ref_high = k_hi; l.hyst = 0; l.shift = 0; }
        else fill_data_words<kWave>(c, l, px_row, 0, 0);
    }
}

/* ---- the coordinate search with shared reads --------------------------------------------------------------------------------
 * What fillPCM1 reads depends on the candidate through two numbers only: where the first cell sits (coords.start + the pixel shift of the stage) and the cell
 * pitch (from coords.stop - coords.start).  On a grid with a step of one pixel the candidates of one anti-diagonal (start + 1, stop + 1 from one to the next)
 * share the pitch, and pixel-shift stage 1 (+1) / 2 (-1) of a candidate reads exactly the cells that stage 0 of its neighbour up / down the diagonal reads.
 * So the grid is walked along its diagonals, a run of consecutive candidates per lane, every read made once and used by the three candidates it belongs
 * to: 625 + 2 per run instead of 3 x 625 reads.  Everything behind the read (Bit Picker, the stop at the first valid CRC, what stays in the line when
 * nothing reads) is run per candidate and stage as before, on the shared cells. */
struct RawRead1 { u128 v; uint16_t crc; };
__device__ inline RawRead1 raw_read_pcm1(const L1 &l, const uint8_t *px_row, int32_t first_px, uint8_t ref_low, uint8_t ref_high)     /* fill_pcm1 with the first cell at first_px */
{
    int32_t acc = (int32_t)l.hpsm + first_px * 128;
    const int32_t lo = l.pixel_start, hi = (int32_t)l.pixel_stop - 1;
    uint32_t a0, a1, a2, b0, b1, b2;
    compare_cells32<32>(px_row, acc, (int32_t)l.psm, lo, hi, ref_low, ref_high, a0, b0);
    compare_cells32<32>(px_row, acc, (int32_t)l.psm, lo, hi, ref_low, ref_high, a1, b1);
    compare_cells32<P1_BITS - 64>(px_row, acc, (int32_t)l.psm, lo, hi, ref_low, ref_high, a2, b2);
    uint64_t s_lo, s_hi;
    solve_automaton_lane((uint64_t)a0 | ((uint64_t)a1 << 32), (uint64_t)a2, (uint64_t)b0 | ((uint64_t)b1 << 32), (uint64_t)b2, s_lo, s_hi);
    s_hi &= (1ull << (P1_BITS - 64)) - 1ull;
    RawRead1 r;
    r.v = ((u128)__brevll(s_lo) << 30) | (u128)(__brevll(s_hi) >> 34);
    r.crc = crc_of_masks(s_lo, s_hi);
    return r;
}
/* read_pcm_data for a candidate of the search - hysteresis depth 0 only, pixel-shift stages 0 .. 2, levels that do not clip, no level sweep behind the
 * line - with the cells of stage 0 / 1 / 2 handed in (at / up / down) */
__device__ inline void read_pcm_data_shared(const BinCtx &c, L1 &l, const RawRead1 &at, const RawRead1 &up, const RawRead1 &down)
{
    set_ppb(l, l.coords);
    const uint8_t low_ref = get_low_level(l.ref_level, 0), high_ref = get_high_level(l.ref_level, 0);
    auto fill = [&](uint8_t s) {            /* fill_data_words(c, l, px_row, 0, s) */
        l.ref_low = low_ref; l.ref_high = high_ref; l.hyst = 0; l.shift = s;
        /* (picked by masks: `s == 0 ? at : ...` over objects becomes one load through a selected address, and then all three live in scratch memory) */
        const uint64_t m0 = s == 0 ? ~0ull : 0ull, m1 = s == 1 ? ~0ull : 0ull, m2 = s >= 2 ? ~0ull : 0ull;
        const uint64_t v_lo = ((uint64_t)at.v & m0) | ((uint64_t)up.v & m1) | ((uint64_t)down.v & m2);
        const uint64_t v_hi = ((uint64_t)(at.v >> 64) & m0) | ((uint64_t)(up.v >> 64) & m1) | ((uint64_t)(down.v >> 64) & m2);
        l.v = ((u128)v_hi << 64) | (u128)v_lo;
        l.calc_crc = (uint16_t)((at.crc & (uint32_t)m0) | (up.crc & (uint32_t)m1) | (down.crc & (uint32_t)m2));
        if ((!crc_valid(l) && (l.ref_level > c.ps.min_white_lvl) && ((c.ps.left_bit_pick != 0) || (c.ps.right_bit_pick != 0))) || c.force_bit_picker)
            pick_cut_bits(c, l);
    };
    bool found = false;
    const bool entry_forced = l.forced_bad;
    bool kept = false;
    u128 k_v = 0; uint16_t k_crc = 0; uint8_t k_pl = 0, k_pr = 0;
#pragma unroll 1
    for (uint8_t s = 0; s <= SHIFT_STAGES_SAFE; s++) {
        fill(s);
        if (crc_valid(l)) { found = true; break; }
        if (s == 0 && c.force_bit_picker) { kept = true; k_v = l.v; k_crc = l.calc_crc; k_pl = l.picked_l; k_pr = l.picked_r; }
    }
    if (!found) {
        if (kept && l.forced_bad == entry_forced) { l.v = k_v; l.calc_crc = k_crc; l.picked_l = k_pl; l.picked_r = k_pr; l.hyst = 0; l.shift = 0; }
        else fill(0);
    }
}
/* the runs of the 25 x 25 grid in diagonal order (tools/gen_diag_table.py 25 2: at most 14 reads per lane) */
#ifdef SDV_EMU
static const uint16_t c_p1_runs[65] = {
#else
__device__ __constant__ const uint16_t c_p1_runs[65] = {
#endif
    0, 6, 15, 25, 35, 45, 55, 66, 78, 90, 100, 110, 120, 132, 142, 153, 165, 175, 187, 197, 209, 219, 231, 243, 253, 265, 276, 288, 300, 312, 324, 334, 346, 356, 368, 378,
    390, 400, 412, 422, 434, 444, 454, 466, 476, 488, 498, 508, 520, 532, 542, 552, 562, 572, 582, 592, 602, 610, 619, 625, 625, 625, 625, 625, 625 };

__device__ inline void stats_reset(CrcStat *a, int count) { for (int i = 0; i < count; i++) { a[i].result = 0; a[i].crc = 0; a[i].hyst = a[i].shift = 0x0f; a[i].idx = 0; } }
__device__ inline void stats_update(CrcStat *a, uint16_t crc, uint8_t hyst, uint8_t shift, uint8_t &valid_cnt)   /* :1789-1826 */
{
    bool found = false;
    if (valid_cnt >= MAX_COLL_CRCS) valid_cnt = MAX_COLL_CRCS - 1;
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (a[i].crc == crc) { a[i].result++; found = true; break; }
    if (!found) {
        valid_cnt++;
        if (valid_cnt < MAX_COLL_CRCS) { a[valid_cnt].crc = crc; a[valid_cnt].hyst = hyst; a[valid_cnt].shift = shift; a[valid_cnt].result++; }
    }
}
/* the same on a table that was not cleared: a new entry starts its count at one (never more entries than the table holds here: one per
 * candidate of a grid row) */
__device__ inline void stats_update_fresh(CrcStat *a, uint16_t crc, uint8_t hyst, uint8_t shift, uint8_t &valid_cnt)
{
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (a[i].crc == crc) { a[i].result++; return; }
    valid_cnt++;
    a[valid_cnt].crc = crc; a[valid_cnt].hyst = hyst; a[valid_cnt].shift = shift; a[valid_cnt].result = 1;
}
static_assert(P1_SEARCH_STEP_CNT < MAX_COLL_CRCS, "a row of the PCM-1 grid fits the statistics table");
__device__ inline void stats_most_frequent(CrcStat *a, uint8_t &valid_cnt)     /* :1829-1928, skip_equal */
{
    a[0].result = 0; a[0].idx = 0; a[0].hyst = 0; a[0].shift = 0;
    if (valid_cnt >= MAX_COLL_CRCS) valid_cnt = MAX_COLL_CRCS - 1;
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (a[i].result > a[0].result) { a[0].result = a[i].result; a[0].crc = a[i].crc; a[0].hyst = a[i].hyst; a[0].shift = a[i].shift; a[0].idx = i; }
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (a[0].idx != i)
            if ((int)a[0].result <= (2 * (int)a[i].result)) { a[0].result = 0; a[0].hyst = 0; a[0].shift = 0; break; }
    if (a[0].result == 0) valid_cnt = 0;
}
__device__ inline SweepEnt sweep_blank() { SweepEnt z; z.result = 0; z.hyst = z.shift = 0x0f; z.pad = 0; z.crc = 0; z.start = z.stop = 0; z.pad2 = 0; return z; }

__device__ __forceinline__ uint32_t row_read(uint32_t x, int idx) { return (uint32_t)__shfl((int)x, idx); }
/* pickLevelByCRCStats (binarizer.cpp:1985-2140) over one row of the grid held by the lanes (lane = column): ok = the entry has the
 * target result, hyst / shift = its depth and stage.  Same two passes: the lowest (depth, stage) and the highest column that has it,
 * then the longest run of that pair below it (with the reference's rule that a run still open at the low end is never compared). */
__device__ inline bool pick_in_row(bool ok, uint32_t hyst, uint32_t shift, int low, int high, uint8_t &picked)
{
    const int lane = lane_id();
    const bool cand = ok && lane >= low && lane <= high && hyst <= 0x0Fu && shift <= (uint32_t)SHIFT_STAGES_MAX;
    const uint32_t key = cand ? ((hyst << 4) | shift) : 0xFFFFu;
    const uint32_t best = wave_min_u32(key);
    if (best == 0xFFFFu) return false;
    const uint32_t M = (uint32_t)__ballot(cand && key == best);
    int high_ref = 31 - __clz((int)M), low_ref = 0, tst_low = 0, tst_high = 0;
    bool range_lock = false, second = false;
    for (int index = high_ref; ; index--) {
        if ((M >> index) & 1u) {
            if (!range_lock) low_ref = index;
            else { if (!second) { tst_high = index; second = true; } tst_low = index; }
        } else {
            range_lock = true;
            if (second) { second = false; if ((tst_high - tst_low) >= (high_ref - low_ref)) { low_ref = tst_low; high_ref = tst_high; } }
        }
        if (index == low) break;
    }
    picked = (uint8_t)(low_ref + (uint8_t)(high_ref - low_ref) / 2);
    return true;
}

/* The vote over one row of the grid held by the lanes (lane = column), findMostFrequentCRC with skip_equal (binarizer.cpp:1829-1928) +
 * invalidateNonFrequentCRCs: m = the columns that read valid (all inside [lo, hi]), crc = the lane's CRC (= grid_row[lane * stride]).  The most frequent CRC wins,
 * the one seen first on a tie; a rival with half its count or more voids the vote.  Returns the columns that carry the winner (0: void),
 * its count and the column it was first seen in. */
__device__ inline uint32_t row_vote(uint32_t m, uint32_t crc, const uint32_t *grid_row, int stride, int lo, int hi, uint32_t &tcnt, uint32_t &tfirst)
{
    const int lane = lane_id();
    const bool mine = lane < 32 && ((m >> (lane & 31)) & 1u);
    uint32_t eq = 0;        /* the other columns' CRCs are read where the candidate reads left them (grid_row[j * stride], low 16 bits) */
    /* (a tape that plays: every column that reads valid carries the one CRC - then every lane's answer is the set of those columns) */
    const uint32_t c_first = row_read(crc, lo);
    if (__ballot(mine && crc != c_first) == 0ull) eq = crc == c_first ? m : 0u;
    else
    for (int j = lo; j <= hi; j++) if ((m >> j) & 1u) { const uint32_t cj = grid_row[j * stride] & 0xFFFFu; eq |= (cj == crc ? 1u : 0u) << j; }
    const uint32_t cnt = (uint32_t)__popc(eq), first = (uint32_t)(__ffs((int)eq) - 1);
    const uint32_t top = wave_max_u32(mine ? ((cnt << 8) | (31u - first)) : 0u);
    tcnt = top >> 8; tfirst = 31u - (top & 0xFFu);
    const uint32_t tcrc = row_read(crc, (int)tfirst);
    const bool rival = __ballot(mine && crc != tcrc && 2u * cnt >= tcnt) != 0ull;
    return rival ? 0u : row_read(eq, (int)tfirst);
}

/* searchPCM1Data (binarizer.cpp:4123-4511).  Returns true when coordinates were found; l is left as the reference leaves its
 * line object (the last candidate's read, the coordinates found or the starting ones). */
__device__ inline bool search_pcm1_data(BinCtx &c, L1 &l, P1Lds &lds, Coords data_loc, uint8_t &hyst_lim, uint8_t &shift_lim)
{
    const int lane = lane_id();
    int scan_step = 1, l0 = 0, l1 = 0, r0 = 0, r1 = 0;
    for (int guard = 2; guard > 0; guard--) {
        set_ppb(l, data_loc);
        scan_step = get_ppb(l);
        scan_step = scan_step >= P1_SEARCH_STEP_DIV ? scan_step / P1_SEARCH_STEP_DIV : 1;
        const int span = (uint16_t)(scan_step * P1_SEARCH_MAX_OFS);
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
    /* the reference's loops run while the offsets stay inside [l0, l1] / [r0, r1]: 25 steps each unless the step does not divide
     * the span (it does: span = 12 steps) */
    const int n_left = (l1 - l0) / scan_step + 1, n_right = (r1 - r0) / scan_step + 1;
    const int nl = n_left < P1_SEARCH_STEP_CNT ? n_left : P1_SEARCH_STEP_CNT, nr = n_right < P1_SEARCH_STEP_CNT ? n_right : P1_SEARCH_STEP_CNT;
    const int n_cand = nl * nr;
    const bool entry_forced = l.forced_bad;
    uint32_t first_coll = 0xFFFFFFFFu;
    SDV_WAVE_SYNC();
    auto note = [&](const L1 &t, int q) {
        uint8_t hy = t.hyst;
        if (t.picked_l != 0 && t.picked_r != 0) hy = 0x0E; else if (t.picked_r != 0) hy = 0x0D; else if (t.picked_l != 0) hy = 0x0C;
        lds.grid[q] = (uint32_t)(uint16_t)(t.v & 0xFFFF) | ((uint32_t)(hy & 0xF) << 16) | ((uint32_t)(t.shift & 0xF) << 20) | ((uint32_t)(crc_valid(t) ? 1 : 0) << 24);
        if (t.forced_bad && !entry_forced) first_coll = first_coll < (uint32_t)q ? first_coll : (uint32_t)q;
    };
    const uint8_t low0 = get_low_level(l.ref_level, 0), high0 = get_high_level(l.ref_level, 0);
    if (SDV_P1_SHARED_READS && scan_step == 1 && nl == P1_GRID && nr == P1_GRID && shift_lim == SHIFT_STAGES_SAFE && !l.ref_sweeped && low0 > l.black && high0 < l.white) {
        /* the grid along its diagonals (see read_pcm_data_shared): this lane's run of cells, diagonal d = row + col, rows upwards */
        int u = c_p1_runs[lane]; const int u_end = c_p1_runs[lane + 1];
        int d = 0, row = 0;
        if (u < u_end) { int left = u; for (;;) { const int len = (d < P1_GRID ? d : 2 * (P1_GRID - 1) - d) + 1; if (left < len) break; left -= len; d++; } row = (d < P1_GRID ? 0 : d - (P1_GRID - 1)) + left; }     /* (a lane without a run has no cell to find) */
        RawRead1 w_down, w_at, w_up;            /* the reads at start - 1, start, start + 1 of the cell that is evaluated next */
        w_down.v = w_at.v = w_up.v = 0; w_down.crc = w_at.crc = w_up.crc = 0;
        int k = 0;                              /* reads made for the current stretch of a diagonal */
        L1 geo = l;                             /* carries the cell pitch of the diagonal */
        while (__ballot(u < u_end) != 0ull) {
            if (u < u_end) {
                if (k == 0) { Coords dc; coords_set(dc, (int16_t)(l0 + row), (int16_t)(r1 - (d - row))); set_ppb(geo, dc); }
                w_down = w_at; w_at = w_up;
                w_up = raw_read_pcm1(geo, lds.w.px, (int32_t)(l0 + row) - 1 + (k < 2 ? k : 2), low0, high0);    /* `row` is the cell evaluated next */
                k++;
                if (k >= 3) {
                    const int col = d - row, q = row * nr + col;
                    L1 t = l;
                    coords_set(t.coords, (int16_t)(l0 + row), (int16_t)(r1 - col));
                    read_pcm_data_shared(c, t, w_at, w_up, w_down);
                    note(t, q);
                    u++;
                    const int row_last = d < P1_GRID ? d : P1_GRID - 1;
                    if (row == row_last) { d++; row = d < P1_GRID ? 0 : d - (P1_GRID - 1); k = 0; }
                    else row++;
                }
            }
        }
    } else
    for (int q = lane; q < n_cand; q += 64) {
        const int row = q / nr, col = q - row * nr;
        L1 t = l;
        coords_set(t.coords, (int16_t)(l0 + row * scan_step), (int16_t)(r1 - col * scan_step));
        read_pcm_data<false>(c, t, lds.w.px, hyst_lim, shift_lim);
        note(t, q);
    }
    first_coll = wave_min_u32(first_coll);
    SDV_WAVE_SYNC();
    /* rows of the grid with a read that is valid (behind the first collision nothing is) */
    uint64_t rows_live;
    {
        bool live = false;
        if (lane < nl) for (int i = 0; i < nr; i++) { const int q = lane * nr + i; live = live || (((lds.grid[q] >> 24) & 1u) != 0 && (uint32_t)q < first_coll); }
        rows_live = __ballot(live);
    }
    /* the votes: row after row, the columns of a row on the lanes (row_vote, pick_in_row); the vote over the rows on lane 0 */
    {
        uint8_t valid_left = 0;         /* lane 0's */
        if (lane == 0) stats_reset(lds.lstats, MAX_COLL_CRCS);
        if (lane < P1_SEARCH_STEP_CNT) lds.w.sweep[P1_LEFT_BASE + lane] = sweep_blank();
        SDV_WAVE_SYNC();
        for (int row = 0; row < nl; row++) {
            SweepEnt le = sweep_blank(); le.result = REF_BAD_CRC; le.crc = 0; le.hyst = HYST_DEPTH_MAX; le.shift = SHIFT_STAGES_MAX;     /* nothing reads in this row, or its vote is void */
            if ((rows_live >> row) & 1ull) {
                const int q = row * nr + lane;
                const uint32_t g = lane < nr ? lds.grid[q] : 0u;
                /* behind the first Bit Picker collision the line object is forced bad: nothing reads valid any more */
                const uint32_t m = (uint32_t)__ballot(lane < nr && ((g >> 24) & 1u) != 0 && (uint32_t)q < first_coll);
                const uint32_t crc = g & 0xFFFFu, hy = (g >> 16) & 0xFu, sh = (g >> 20) & 0xFu;
                uint32_t tcnt = 0, tfirst = 0;
                const uint32_t okm = m ? row_vote(m, crc, &lds.grid[row * nr], 1, __ffs((int)m) - 1, 31 - __clz((int)m), tcnt, tfirst) : 0u;
                uint8_t right_ofs = 0xFF;
                if (okm && pick_in_row(lane < 32 && ((okm >> (lane & 31)) & 1u), hy, sh, 0, P1_SEARCH_STEP_CNT - 1, right_ofs)) {
                    le.result = REF_CRC_OK; le.crc = (uint16_t)row_read(crc, right_ofs); le.hyst = (uint8_t)row_read(hy, right_ofs); le.shift = (uint8_t)row_read(sh, right_ofs);
                    le.start = (int16_t)(l0 + row * scan_step); le.stop = (int16_t)(r1 - (int)right_ofs * scan_step);
                    /* the row's winner enters the statistics of the left coordinate once per column that carried it */
                    const uint16_t top_crc = (uint16_t)row_read(crc, (int)tfirst); const uint8_t top_h = (uint8_t)row_read(hy, (int)tfirst), top_s = (uint8_t)row_read(sh, (int)tfirst);
                    if (lane == 0) {
                        bool found_it = false;
                        for (uint8_t i = 1; i <= valid_left; i++) if (lds.lstats[i].crc == top_crc) { lds.lstats[i].result = (uint8_t)(lds.lstats[i].result + tcnt); found_it = true; break; }
                        if (!found_it) { valid_left++; lds.lstats[valid_left].crc = top_crc; lds.lstats[valid_left].hyst = top_h; lds.lstats[valid_left].shift = top_s; lds.lstats[valid_left].result = (uint8_t)tcnt; }
                    }
                }
            }
            if (lane == 0) lds.w.sweep[P1_LEFT_BASE + row] = le;
        }
        SDV_WAVE_SYNC();
        if (lane == 0) {
            uint8_t left_ofs = 0xFF;
            if (valid_left > 0) {
                stats_most_frequent(lds.lstats, valid_left);
                sweep_invalidate_non_frequent(lds.w, P1_LEFT_BASE, P1_LEFT_BASE + P1_SEARCH_STEP_CNT - 1, valid_left, lds.lstats[0].crc);
                if (valid_left > 0)
                    if (pick_level_by_crc_stats(lds.w, &left_ofs, P1_LEFT_BASE, P1_LEFT_BASE + P1_SEARCH_STEP_CNT - 1, REF_CRC_OK, 0x0F, SHIFT_STAGES_MAX) != SPAN_OK) valid_left = 0;
            }
            lds.vote[0] = valid_left > 0 ? 1 : 0;
            if (valid_left > 0) { lds.vote[1] = lds.w.sweep[left_ofs].start; lds.vote[2] = lds.w.sweep[left_ofs].stop; }
        }
    }
    SDV_WAVE_SYNC();
    const bool found = lds.vote[0] != 0;
    const int f_start = lds.vote[1], f_stop = lds.vote[2];
    SDV_WAVE_SYNC();
    /* what the last candidate leaves in the line object, forced bad if a collision happened on the way */
    if (n_cand > 0) {
        if (first_coll != 0xFFFFFFFFu && first_coll < (uint32_t)(n_cand - 1)) l.forced_bad = true;
        coords_set(l.coords, (int16_t)(l0 + (nl - 1) * scan_step), (int16_t)(r1 - (nr - 1) * scan_step));
        read_pcm_data<true>(c, l, lds.w.px, hyst_lim, shift_lim);
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

/* findPCM1Coordinates (binarizer.cpp:5601-5813) */
__device__ inline void find_pcm1_coordinates(BinCtx &c, L1 &l, P1Lds &lds, const Coords &history, uint8_t &hyst_lim, uint8_t &shift_lim)
{
    Coords dc; coords_clear(dc);
    const int ss = c.scan_start, se = c.scan_end;
    const int margin = (uint16_t)(se - ss) / 16;
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
    search_pcm1_data(c, l, lds, dc, hyst_lim, shift_lim);
    hyst_lim = in_hyst; shift_lim = in_shift;
}

/* ---- reference level sweep (MODE_INSANE only for this format, binarizer.cpp:1105-1112) -------------------------------------------
 * The sweep's table (one entry per level) has to outlive the coordinate searches that run inside it, and those use the first 58
 * (PCM-1) / 150 (PCM-16x0) entries of WaveLds::sweep.  While the sweep runs the table is kept packed in two words per level: one in
 * the brightness histogram (not in use at that point), one behind entry 160 of the sweep array; it is unpacked in place when the
 * sweep is done. */
enum { RS_BASE = 160 };
static_assert(sizeof(SweepEnt) * (256 - RS_BASE) >= 256 * 4, "the second words of the packed table fit behind the search's entries");
__device__ inline uint32_t *rs_words(WaveLds &w) { return reinterpret_cast<uint32_t *>(&w.sweep[RS_BASE]); }
__device__ inline void rs_store(WaveLds &w, int lvl, const SweepEnt &e)
{
    w.hist[lvl] = (uint32_t)e.crc | ((uint32_t)(e.hyst & 0xF) << 16) | ((uint32_t)(e.shift & 0xF) << 20) | ((uint32_t)e.result << 24);
    rs_words(w)[lvl] = (uint32_t)(uint16_t)e.start | ((uint32_t)(uint16_t)e.stop << 16);
}
__device__ inline void rs_unpack(WaveLds &w)
{
    const int lane = lane_id();
    uint32_t a[4], b[4];
    SDV_WAVE_SYNC();
    for (int k = 0; k < 4; k++) { a[k] = w.hist[lane + 64 * k]; b[k] = rs_words(w)[lane + 64 * k]; }
    SDV_WAVE_SYNC();
    for (int k = 0; k < 4; k++) {
        SweepEnt e = sweep_blank();
        e.crc = (uint16_t)(a[k] & 0xFFFF); e.hyst = (uint8_t)((a[k] >> 16) & 0xF); e.shift = (uint8_t)((a[k] >> 20) & 0xF); e.result = (uint8_t)(a[k] >> 24);
        e.start = (int16_t)(b[k] & 0xFFFF); e.stop = (int16_t)(b[k] >> 16);
        w.sweep[lane + 64 * k] = e;
    }
    SDV_WAVE_SYNC();
}

/* Binarizer::sweepRefLevel (binarizer.cpp:3551-3817) with a PCM1Line as the trial line.  The first-try read from the preset
 * coordinates never runs for this format (skip_bin stays false, :3647-3679): every level runs the whole coordinate search, levels one
 * after the other (the search itself is spread over the lanes).  clear() through the PCMLine pointer is the base clear(): cells and
 * picked bits of the trial line persist from level to level. */
__device__ inline void sweep_ref_level_p1(BinCtx &c, const Bin &b, P1Lds &lds, const L1 &pcm_line, bool vl_doubled)
{
    Coords forced; calc_forced_coords(b, c.ps, forced);
    uint8_t low_lvl = (uint8_t)(pcm_line.black + 1), high_lvl = (uint8_t)(pcm_line.white - 1);
    if (c.ps.min_ref_lvl > low_lvl) low_lvl = c.ps.min_ref_lvl;
    if (c.ps.max_ref_lvl < high_lvl) high_lvl = c.ps.max_ref_lvl;
    L1 t; p1_clear(t);
    for (int lvl = (int)high_lvl; lvl >= (int)low_lvl; lvl--) {
        base_clear(t);
        if (c.scan_end > 0 && P1_BITS <= c.scan_end) { t.pixel_start = 0; t.pixel_stop = c.scan_end; }     /* setSourcePixels(0, size - 1) */
        t.coords.doubled = vl_doubled;
        t.black = low_lvl; t.white = high_lvl; t.ref_level = (uint8_t)lvl;
        uint8_t hyst_lim = 0, shift_lim = SHIFT_STAGES_SAFE;        /* calcRefLevelBySweep, :3847-3851 */
        if (!crc_valid(t)) {
            if (!coords_valid(forced)) find_pcm1_coordinates(c, t, lds, b.in_coord, hyst_lim, shift_lim);
            else { t.coords = forced; t.coords_set = true; }
            if (t.coords_set) read_pcm_data<true>(c, t, lds.w.px, hyst_lim, shift_lim);
        }
        if (t.picked_l != 0 && t.picked_r != 0) t.hyst = (uint8_t)(t.hyst + HYST_DEPTH_MAX + 3);         /* :3735-3752 */
        else if (t.picked_r != 0) t.hyst = (uint8_t)(t.hyst + HYST_DEPTH_MAX + 2);
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
__device__ inline void calc_ref_level_by_sweep_p1(BinCtx &c, const Bin &b, P1Lds &lds, L1 &l, bool vl_doubled, uint8_t &hyst_lim, uint8_t &shift_lim)
{
    const int lane = lane_id();
    const uint8_t fast_ref = pick_center_ref_level(c.ps, l.black, l.white);
    const uint8_t blk1 = (uint8_t)(l.black + 1), wht1 = (uint8_t)(l.white - 1);
    hyst_lim = 0; shift_lim = SHIFT_STAGES_SAFE;
    SDV_WAVE_SYNC();
    { const SweepEnt z = sweep_blank(); for (int i = lane; i < 256; i += 64) rs_store(lds.w, i, z); }
    SDV_WAVE_SYNC();
    sweep_ref_level_p1(c, b, lds, l, vl_doubled);
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

/* findBlackWhite (binarizer.cpp:3116-3473) over the PCM-1 part of the line (findPCM1BW, :2560-2600) */
__device__ inline bool find_black_white_p1(const BinCtx &c, WaveLds &lds, L1 &line, bool &was_bw_scanned, bool sweep_flag)
{
    uint16_t pixel_limit = (uint16_t)(c.scan_end - c.scan_start);
    const uint16_t search_end = (uint16_t)(c.scan_end - (uint16_t)(pixel_limit / 32));
    pixel_limit = (uint16_t)(c.scan_start + (uint16_t)(pixel_limit / 8));
    hist_clear(lds);
    hist_add_range(lds, pixel_limit, search_end);

    const BwLevels bw = bw_from_spread(c.ps, spread_levels(c.ps, lds), sweep_flag);      /* Binarizer::do_ref_lvl_sweep: left by the last line that got as far as :1104 */
    was_bw_scanned = true;
    line.black = bw.black; line.white = bw.white;
    line.bw_set = bw.set;
    return line.bw_set;
}

__device__ inline void set_service(L1 &l, uint8_t srv) { base_clear(l); l.service = srv; }      /* PCMLine::setServiceLine: base clear() only */

__device__ inline void emit_rec(const L1 &l, uint32_t frame, uint16_t line_no, bool from_doubled, sdv_pcm1_bin_rec *dst)
{
    if (lane_id() != 0) return;
    sdv_pcm1_bin_rec r;
    r.frame_number = frame; r.line_number = line_no;
    for (int k = 0; k < 7; k++) r.words[k] = get_word(l, k);
    r.calc_crc = l.calc_crc;
    r.data_start = l.coords.start; r.data_stop = l.coords.stop;
    r.black_level = l.black; r.white_level = l.white; r.ref_low = l.ref_low; r.ref_level = l.ref_level; r.ref_high = l.ref_high;
    r.hysteresis_depth = l.hyst; r.shift_stage = l.shift; r.service_type = l.service;
    r.picked_bits_left = l.picked_l; r.picked_bits_right = l.picked_r;
    r.flags = (uint8_t)((l.ref_sweeped ? SDV_LF_REF_SWEEPED : 0) | (l.coords_sweeped ? SDV_LF_COORDS_SWEEPED : 0) | (l.by_ext_tune ? SDV_LF_BY_EXT_TUNE : 0) | (l.bw_set ? SDV_LF_BW_SET : 0) |
                        (l.coords_set ? SDV_LF_COORDS_SET : 0) | (l.forced_bad ? SDV_LF_FORCED_BAD : 0) | (crc_valid(l) ? SDV_LF_CRC_VALID : 0) |
                        (from_doubled ? SDV_LF_FROM_DOUBLED : 0));
    r._pad[0] = r._pad[1] = r._pad[2] = 0;
    *dst = r;
}

__device__ inline void stage_row(uint8_t *px, const uint8_t *row, int width)
{
    const int lane = lane_id();
    SDV_WAVE_SYNC();
    if (((((uintptr_t)row) | (uintptr_t)width) & 15) == 0) {            /* 16 bytes per lane */
        for (int p = lane * 16; p < width; p += 64 * 16) *(uint4 *)&px[p] = *(const uint4 *)(row + p);
    } else {
        for (int p = lane; p < width; p += 64) px[p] = row[p];
    }
    SDV_WAVE_SYNC();
}

/* Stage STG_INPUT_ALL of processLine alone (binarizer.cpp:774-931), for the lean build of the frame kernel: a line whose reference level and
 * coordinates are preset - its levels measured first when they are not (the first line of a frame behind the prescan, the line behind one that did
 * not read) - is read with them, ladder and Bit Picker included; true when that ends in STG_DATA_OK, i.e. the line is done.  false: the line needs the
 * stages behind it (nothing is decided, `out` is not to be used) - the frame goes to the full build. */
__device__ inline bool input_all_p1(const BinCtx &c, const Bin &b, WaveLds &lds, L1 &out, bool vl_doubled)
{
    if (c.ps.en_force_coords) return false;
    if (!(is_ref_level_preset(b, c.ps) && coords_valid(b.in_coord))) return false;
    p1_clear(out);
    out.coords.doubled = vl_doubled;
    if (c.scan_end > c.scan_start && P1_BITS <= (c.scan_end - c.scan_start)) { out.pixel_start = c.scan_start; out.pixel_stop = c.scan_end; }
    bool was_bw_scanned = false;
    if (are_bw_levels_preset(b, c.ps)) { out.black = b.in_black; out.white = b.in_white; out.bw_set = true; }
    if (!out.bw_set) find_black_white_p1(c, lds, out, was_bw_scanned, b.do_ref_lvl_sweep);
    out.coords = b.in_coord;
    out.ref_level = b.in_ref;
    if (!out.bw_set) return false;
    if (b.in_ref >= out.white || b.in_ref <= out.black) return false;
    read_pcm_data<true>(c, out, lds.px, b.in_max_hyst, b.in_max_shift);
    if (!crc_valid(out)) return false;
    out.by_ext_tune = true;
    if (out.forced_bad) return false;
    if (has_header(out)) set_service(out, SDV_SRV_HEADER_LINE);
    return true;
}

/* Binarizer::processLine (binarizer.cpp:443-1724), PCM1Line output, for the video line staged in lds.w.px: the stage machine from
 * what the caller has preset on its Binarizer (`b`: in_black / in_white / in_ref / in_coord, limits by mode) to the finished line.
 * A pure function of the pixels, the presets, the mode and the fine settings. */
/* kInsane: the build for MODE_INSANE, the only one that holds the reference level sweep (kept out of the kernels of the other modes:
 * with it in the same function their register allocation and their rates suffer - measured 736 k -> 574-648 k frames/s for the PCM-1
 * frame driver, 372 k -> 240 k for the PCM-16x0 one) */
template <bool kInsane>
__device__ __forceinline__ void process_line_p1(BinCtx &c, Bin &b, bool coord_search, P1Lds &lds, L1 &out, bool vl_doubled)
{
    p1_clear(out);
    out.coords.doubled = vl_doubled;
    if (c.scan_end > c.scan_start && P1_BITS <= (c.scan_end - c.scan_start)) { out.pixel_start = c.scan_start; out.pixel_stop = c.scan_end; }   /* setSourcePixels */
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
            if (!out.bw_set) find_black_white_p1(c, lds.w, out, was_bw_scanned, b.do_ref_lvl_sweep);
            if (!coords_valid(forced)) out.coords = b.in_coord;
            out.ref_level = b.in_ref;
            if (!out.bw_set) state = STG_NO_GOOD;
            else if (b.in_ref >= out.white || b.in_ref <= out.black) state = STG_REF_FIND;
            else {
                read_pcm_data<true>(c, out, lds.w.px, hyst_lim, shift_lim);
                if (crc_valid(out)) { out.by_ext_tune = true; state = STG_DATA_OK; } else state = STG_REF_FIND;
            }
        } else if (state == STG_INPUT_LEVEL) {                  /* :932-1072 */
            if (!was_bw_scanned) find_black_white_p1(c, lds.w, out, was_bw_scanned, b.do_ref_lvl_sweep);
            if (!coords_valid(forced)) coords_set(out.coords, (int16_t)c.scan_start, (int16_t)c.scan_end);
            out.ref_level = b.in_ref;
            state = out.bw_set ? STG_REF_FIND : STG_NO_GOOD;
        } else if (state == STG_REF_FIND) {                     /* :1073-1390 */
            if (!was_bw_scanned) find_black_white_p1(c, lds.w, out, was_bw_scanned, b.do_ref_lvl_sweep);
            if (!out.bw_set) state = STG_NO_GOOD;
            else if ((b.do_ref_lvl_sweep = kInsane)) {          /* = (mode == MODE_INSANE), :1104-1133: the member keeps this until the next line gets here; STG_REF_SWEEP_RUN, :1391-1400 */
                if (kInsane) calc_ref_level_by_sweep_p1(c, b, lds, out, vl_doubled, hyst_lim, shift_lim);
                state = STG_READ_PCM;
            } else {
                hyst_lim = HYST_DEPTH_SAFE; shift_lim = SHIFT_STAGES_MIN;
                state = STG_READ_PCM;
                out.ref_level = pick_center_ref_level(c.ps, out.black, out.white);
                if (coords_valid(forced)) { out.coords = forced; out.coords_set = true; }
                else {
                    if (!coords_valid(b.in_coord)) coords_set(out.coords, (int16_t)c.scan_start, (int16_t)c.scan_end);
                    else out.coords = b.in_coord;
                    if (c.ps.en_coord_search && coord_search) find_pcm1_coordinates(c, out, lds, b.in_coord, hyst_lim, shift_lim);
                }
                if (!out.coords_set) { hyst_lim = HYST_DEPTH_SAFE; shift_lim = SHIFT_STAGES_MIN; }
                else { hyst_lim = b.in_max_hyst; shift_lim = b.in_max_shift; }
            }
        } else if (state == STG_READ_PCM) {                     /* :1401-1533 */
            if (coords_valid(forced)) { hyst_lim = HYST_DEPTH_SAFE; shift_lim = SHIFT_STAGES_MIN; }
            if (out.coords_set) read_pcm_data<true>(c, out, lds.w.px, hyst_lim, shift_lim);
            if (crc_valid(out)) state = STG_DATA_OK;
            if (state != STG_DATA_OK) {
                if (coords_valid(b.in_coord) && !coords_valid(forced) && !b.do_ref_lvl_sweep && !out.forced_bad && !out.coords_set) {
                    if (coords_ne(out.coords, b.in_coord)) {
                        out.coords = b.in_coord;
                        read_pcm_data<true>(c, out, lds.w.px, hyst_lim, shift_lim);
                        if (crc_valid(out)) state = STG_DATA_OK;
                    }
                }
                if (state != STG_DATA_OK) state = STG_NO_GOOD;
            }
        } else if (state == STG_DATA_OK) {                      /* :1534-1621 */
            if (out.forced_bad) state = STG_NO_GOOD;
            else {
                if (has_header(out)) set_service(out, SDV_SRV_HEADER_LINE);
                break;
            }
        } else {                                                /* STG_NO_GOOD, :1622-1669 */
            if (crc_valid(out)) set_invalid_crc(out);
            break;
        }
        if (stage_count > STG_MAX) break;
    }
}

/* one line of sdv_pcm1_binarize_lines (service lines and empty lines are the caller's: they carry no pixels and are not sent to the
 * device) */
template <bool kInsane>
__device__ inline void line_body(const LineArgs1 &a, P1Lds &lds, size_t li)
{
    stage_row(lds.w.px, a.luma + li * a.row_stride, a.width);

    BinCtx c; c.ps = a.preset; c.mode = a.mode; c.scan_start = 0; c.scan_end = (uint16_t)(a.width - 1);
    c.force_bit_picker = true;      /* the Binarizer is constructed with it set (binarizer.cpp:82) and nothing clears it */
    Bin b;                          /* the presets, in the form the shared helpers take */
    b.in_black = b.in_white = b.in_ref = 0; coords_clear(b.in_coord);
    if (a.states) {
        const sdv_bin_state s = a.states[li];
        b.in_black = s.in_def_black; b.in_white = s.in_def_white; b.in_ref = s.in_def_reference;
        b.in_coord.start = s.in_def_start; b.in_coord.stop = s.in_def_stop; b.in_coord.doubled = s.in_def_from_doubled != 0;
    }
    bin_set_mode(b, a.mode);
    b.scan_start = c.scan_start; b.scan_end = c.scan_end; b.vl_doubled = a.doubled != 0;
    b.do_ref_lvl_sweep = a.states ? a.states[li].do_ref_lvl_sweep != 0 : false;
    L1 out;
    process_line_p1<kInsane>(c, b, a.coord_search != 0, lds, out, a.doubled != 0);
    emit_rec(out, a.frame_number, (uint16_t)(a.first_line + li * a.line_step), a.doubled != 0, &a.out[li]);
}

/* The lean build of the line (the pattern of the STC-007 frame kernel): a line whose caller preset levels, reference level and
 * coordinates is read with them - stage STG_INPUT_ALL of processLine and nothing else, which needs a fraction of the registers - and
 * a line that does not come out of that with a valid CRC (or has no complete presets, or forced coordinates) is put on the list of
 * the full kernel, which decodes it from the start. */
__device__ inline void lean_body(const LineArgs1 &a, uint8_t *px, size_t li)
{
    bool done = false;
    const sdv_bin_state s = a.states[li];
    BinCtx c; c.ps = a.preset; c.mode = a.mode; c.scan_start = 0; c.scan_end = (uint16_t)(a.width - 1); c.force_bit_picker = true;
    Bin b;
    b.in_black = s.in_def_black; b.in_white = s.in_def_white; b.in_ref = s.in_def_reference;
    b.in_coord.start = s.in_def_start; b.in_coord.stop = s.in_def_stop; b.in_coord.doubled = s.in_def_from_doubled != 0;
    bin_set_mode(b, a.mode);
    if (!c.ps.en_force_coords && are_bw_levels_preset(b, c.ps) && is_ref_level_preset(b, c.ps) && coords_valid(b.in_coord)
        && b.in_ref < b.in_white && b.in_ref > b.in_black) {
        stage_row(px, a.luma + li * a.row_stride, a.width);
        L1 out; p1_clear(out);
        if (c.scan_end > c.scan_start && P1_BITS <= (c.scan_end - c.scan_start)) { out.pixel_start = c.scan_start; out.pixel_stop = c.scan_end; }
        out.black = b.in_black; out.white = b.in_white; out.bw_set = true;
        out.coords = b.in_coord;
        out.ref_level = b.in_ref;
        read_pcm_data<true>(c, out, px, b.in_max_hyst, b.in_max_shift);
        if (crc_valid(out)) {
            out.by_ext_tune = true;
            if (has_header(out)) set_service(out, SDV_SRV_HEADER_LINE);
            emit_rec(out, a.frame_number, (uint16_t)(a.first_line + li * a.line_step), a.doubled != 0, &a.out[li]);
            done = true;
        }
    }
    if (!done && lane_id() == 0) a.list[atomicAdd(&a.counters[0], 1)] = (int)li;
}

} // namespace sdvp1b

#ifndef SDV_P1B_WAVES_PER_EU
#define SDV_P1B_WAVES_PER_EU 4
#endif
#ifndef SDV_P1B_LEAN_WAVES_PER_EU
#define SDV_P1B_LEAN_WAVES_PER_EU 8
#endif
/* the full line: over all lines (grid-stride), or over the list the lean kernel left */
template <bool kInsane>
__device__ inline void lines_kernel_body(const sdvp1b::LineArgs1 &a, sdvp1b::P1Lds &lds)
{
    if (a.list) {
        /* entry blockIdx.x first, then whatever is next behind the grid: no atomic at all while the list is shorter than the grid */
        const int n = a.counters[0];
        int i = (int)blockIdx.x;
        while (i < n) {
            sdvp1b::line_body<kInsane>(a, lds, (size_t)a.list[i]);
            SDV_WAVE_SYNC();
            if (sdv::lane_id() == 0) lds.vote[3] = (int)gridDim.x + atomicAdd(&a.counters[1], 1);
            SDV_WAVE_SYNC();
            i = lds.vote[3];
        }
    } else {
        for (size_t li = blockIdx.x; li < a.n_lines; li += gridDim.x) sdvp1b::line_body<kInsane>(a, lds, li);
    }
}
__global__ void __launch_bounds__(64, SDV_P1B_WAVES_PER_EU) sdv_k_pcm1_lines(sdvp1b::LineArgs1 a)
{
    __shared__ sdvp1b::P1Lds lds;
    lines_kernel_body<false>(a, lds);
}
__global__ void __launch_bounds__(64, SDV_P1B_WAVES_PER_EU) sdv_k_pcm1_lines_insane(sdvp1b::LineArgs1 a)      /* MODE_INSANE */
{
    __shared__ sdvp1b::P1Lds lds;
    lines_kernel_body<true>(a, lds);
}
__global__ void __launch_bounds__(64, SDV_P1B_LEAN_WAVES_PER_EU) sdv_k_pcm1_lines_lean(sdvp1b::LineArgs1 a)
{
    __shared__ alignas(16) uint8_t px[SDV_PX_BYTES];
    sdvp1b::lean_body(a, px, (size_t)blockIdx.x);
}
