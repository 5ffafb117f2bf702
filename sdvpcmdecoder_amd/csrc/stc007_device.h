/*
 * stc007_device.h - HIP device code of the STC-007 binarize path for gfx950 (wave64).
 *
 * One wavefront decodes one video frame: it walks the frame's scanlines in the order
 * VideoInFFMPEG::spliceFrame emits them (vin_ffmpeg.cpp:213-364) and carries the
 * VideoToDigital/Binarizer feedback state from line to line (videotodigital.cpp:825-1717),
 * because in the reference every line's result depends on the previous good line
 * (Binarizer::setGoodParameters, binarizer.cpp:353-377).  Frames of a batch run in parallel
 * from speculated incoming states that the host validates afterwards.
 *
 * Per scanline: the row is read from HBM once (coalesced 16-byte loads), staged in LDS, and every
 * retry of the closed-loop binarizer (hysteresis x pixel-shift ladder, 24 marker hysteresis levels,
 * the <=234-level reference sweep) re-reads LDS only.
 *
 * Wave-level building blocks:
 *   - bit extraction: each lane samples 2 of the 128 bit cells; the reference's 2-state hysteresis
 *     automaton (binarizer.cpp:7377-7399) is solved for all 128 cells at once from four v_cmp ballots
 *     (bit-parallel "last constant + toggle parity" formulation, see fill_bits());
 *   - CRC-16/CCITT-FALSE (pcmline.cpp:461-487): linear over GF(2) => 16 lanes each take one CRC bit as
 *     parity(popcount(bits & K_j)), result assembled with one ballot;
 *   - brightness histograms: LDS atomics, one row per wave;
 *   - marker search at 24 hysteresis levels: one lane per level;
 *   - reference-level sweep: one lane per level (4 rounds of 64), statistics/vote replayed serially.
 *
 * Collectives are only used in wave-uniform control flow.  The same source is compiled by g++ with
 * tests/emu/hip_emu.h (-DSDV_EMU) so the logic can be exercised without a GPU in tests.
 *
 * Reference citations are file:line in Fagear/SDVPCMdecoder v0.99.7.
 */
#ifndef SDV_STC007_DEVICE_H
#define SDV_STC007_DEVICE_H

#include "../../include/sdvpcm.h"

#ifndef SDV_EMU
#include <hip/hip_runtime.h>
#endif

#define SDV_MAX_WIDTH 1536          /* staged scanline bytes per wave (720 px SD, 1440 px doubled) */
#ifndef SDV_BATCH_LINES
#define SDV_BATCH_LINES 2           /* batch loop: scanlines per iteration (their decode chains are independent and interleave); 1 = off */
#endif
#ifndef SDV_NT_RECORDS
#define SDV_NT_RECORDS 1           /* the batch path writes its records with streaming stores: 0.756 -> 0.740 ms per 10 000 frames (means of eight alternating runs) */
#endif
#ifndef SDV_CAPTURE_D
#define SDV_CAPTURE_D 4             /* capture loop: row pairs in flight */
#endif
#ifndef SDV_RB_GENERAL
#define SDV_RB_GENERAL 1            /* the batches of a tape on a later shift stage (rung_hint) in the general build too */
#endif
#ifndef SDV_PREFETCH_ITERS
#define SDV_PREFETCH_ITERS 1         /* multi-line loop: iterations of rows in flight ahead of the one being decoded */
#endif
#define SDV_ROWQ ((SDV_BATCH_LINES > 1 ? SDV_BATCH_LINES : 2) * SDV_PREFETCH_ITERS - 1)   /* rows queued behind `row` */
#define SDV_PX_BYTES (1024 * (SDV_BATCH_LINES > 2 ? SDV_BATCH_LINES : 2))   /* LDS bytes for staged scanlines: rows at a pitch of 1024, each written by 64 lanes x 16 bytes */
#define SDV_MAX_HEIGHT 640          /* LINES_PER_FRAME_MAX, config.h:79 */

namespace sdv {

/* ---- constants of the reference (pcmline.h, stc007line.h, binarizer.h, videotodigital.h) ---- */
enum { BITS_IN_LINE = 137, BITS_DATA = 128, BITS_BETWEEN = 132, BITS_START = 4 };
enum { MARK_ST_START = 0, MARK_ST_TOP_1, MARK_ST_BOT_1, MARK_ST_TOP_2, MARK_ST_BOT_2 };
enum { MARK_ED_START = 0, MARK_ED_TOP, MARK_ED_BOT, MARK_ED_LEN_OK };
enum { HYST_DEPTH_SAFE = 4, HYST_DEPTH_MAX = 10, SHIFT_STAGES_MIN = 0, SHIFT_STAGES_SAFE = 2, SHIFT_STAGES_MAX = 4 };
enum { STG_INPUT_ALL = 0, STG_INPUT_LEVEL, STG_REF_FIND, STG_REF_SWEEP_RUN, STG_READ_PCM, STG_DATA_OK, STG_NO_GOOD, STG_MAX };
enum { REF_NO_PCM = 0, REF_BAD_CRC, REF_CRC_COLL, REF_CRC_OK };
enum { SPAN_NOT_FOUND = 0, SPAN_TOO_NARROW, SPAN_OK };
enum { MAX_COLL_CRCS = 32, MIN_VALID_CRCS = 5 };
enum { FIELD_INIT = 0, FIELD_NEW, FIELD_SAFE, FIELD_UNSAFE };
enum { COORD_HISTORY_DEPTH = 9, COORD_LONG_HISTORY = 16, BIT_DIFF_THRES_DIV = 32 };
enum { NO_COORD_LEFT = -32768, NO_COORD_RIGHT = 32767 };
enum { CRC_SILENT = 0xA96A };

/* ---- CRC-16/CCITT-FALSE as a GF(2)-linear map of the 112 data bits -------------------------- */
struct CrcTables { uint64_t klo[16], khi[16]; uint16_t init; };

constexpr uint16_t crc16_step(uint16_t crc, int bit)
{
    bool msb = (crc & 0x8000) != 0;
    crc = (uint16_t)(crc << 1);
    if (msb != (bit != 0)) crc ^= 0x1021;
    return crc;
}
constexpr CrcTables make_crc_tables()
{
    CrcTables t{};
    /* contribution of the init value: CRC of 112 zero bits from 0xFFFF */
    uint16_t c = 0xFFFF;
    for (int i = 0; i < 112; i++) c = crc16_step(c, 0);
    t.init = c;
    for (int b = 0; b < 112; b++) {
        /* CRC (from init 0) of the message with only data bit b set */
        uint16_t v = 0;
        for (int i = 0; i < 112; i++) v = crc16_step(v, i == b);
        for (int j = 0; j < 16; j++)
            if (v & (1u << j)) { if (b < 64) t.klo[j] |= (1ull << b); else t.khi[j] |= (1ull << (b - 64)); }
    }
    return t;
}
#ifdef SDV_EMU
static const CrcTables c_crc = make_crc_tables();
#else
__device__ __constant__ const CrcTables c_crc = make_crc_tables();
#endif

/* ---- per-wave LDS --------------------------------------------------------------------------- */
struct SweepEnt { uint8_t result, hyst, shift, pad; uint16_t crc; int16_t start, stop; uint16_t pad2; };
struct CrcStat { uint8_t result, hyst, shift, idx; uint16_t crc; };
struct WaveLds {
    alignas(16) uint8_t px[SDV_PX_BYTES];     /* one staged scanline, or two of up to 1024 bytes side by side (the pair loop) */
    uint32_t hist[256];
    SweepEnt sweep[256];
    uint8_t park_room[256];                  /* (the whole-frame capture parks the second field's lines over px .. here: five chunks of 64 lines for a 640-line frame) */
    CrcStat crc_stats[MAX_COLL_CRCS + 1];
    uint32_t lv_keys[COORD_HISTORY_DEPTH];   /* last_valid_coord_list as sort keys (videotodigital.cpp:707) */
    uint32_t long_keys[COORD_LONG_HISTORY];  /* long_valid_coords (videotodigital.cpp:710) */
    uint32_t state_in[32];                   /* the chain state the frame was started from (sdv_v2d_state, a dword per lane: v2d_stage_state_in) */
};

/* ---- launch parameters ---------------------------------------------------------------------- */
/* the chain after a frame: 0 its successor was started from exactly what it produced, 1 a lean wave gave the frame up,
 * 2 the successor was started from something else */
enum { VF_OK = 0, VF_ABORTED = 1, VF_BREAK = 2,
       VF_KIND = 0x0F,
       VF_HIST = 0x10,              /* (with VF_BREAK) the state the successor was started from differs from this frame's in the 16-frame coordinate history: the scheduler's
                                     * "the history moves on" has something to do behind this link even when the frame itself only re-tuned its levels */
       VF_SLOW = 0x20,              /* lines of the frame went through the general path (full kernel only): the frame did need that kernel - what the scheduler's "worn tape" is decided on */
       VF_MOVED = 0x40,             /* the frame leaves the chain with other coordinates / histories than the model makes of what it was started from */
       VF_RETUNED = 0x80 };         /* ... with other black / white / reference levels */
struct FrameArgs {
    const uint8_t *luma;            /* frame f, row r at luma + f*frame_stride + r*row_stride */
    size_t frame_stride, row_stride;
    int width, height;
    uint32_t first_frame_no;        /* frame_number of frame index 0 */
    int frame_lo, frame_hi;         /* frames [lo, hi) are processed by this launch ... */
    const int *frame_list;          /* ... or, when set, the frames frame_list[0 .. grid) */
    uint8_t *flag;                  /* [n_total] VF_*: how frame f left the chain (written by the frame itself when it is done) */
    uint8_t *refs;                  /* [3 * n_total] or NULL: the reference level frame f was started from, the one it hands on, and whether it pushed one pair
                                     * into its coordinate history (the scheduler's guess at what a frame does with another state: engine.inc, "a level that
                                     * passes through", "the history moves on") */
    /* Trajectory snapshots of the general kernel (tc_*, below: "a pass that meets the last one"): per frame two sets of TC_ENTRIES snapshots, a header of
     * two words (which set holds the last complete pass + 1, or 0; its entries), and a second list of coordinate keys beside `scratch`.  NULL: off. */
    struct TcSnap *tc_snaps; uint32_t *tc_hdr; uint32_t *tc_keys;
    const uint8_t *skip;            /* [n_total] or NULL: frames of this launch that need not be decoded again (v2d_relink) */
    uint8_t *sig;                   /* [n_total] or NULL: where a frame the lean kernel gave up saw the line begin that it gave up on (give_up_signature) - frames that
                                     * gave up side by side over a window that jumped are told apart by it (engine.inc, the crowd rule) */
    int n_total;                    /* frames of the call */
    int new_file_frame;             /* frame index that is preceded by a NEW_FILE service line, or -1 */
    int end_file_frame;             /* frame index of the filler frame that closes the file (no pixels: FILLER lines, END_FILE), or -1 */
    uint8_t doubled, mode, check_line_copy, coordinate_damper, m2_format;
    sdv_bin_preset preset;
    const sdv_v2d_state *states_in; /* [n frames] speculated incoming chain state */
    sdv_v2d_state *states_out;      /* [n frames] outgoing chain state */
    sdv_line_rec *recs;             /* frame f: recs + f*(height+3) (+1 for every frame after new_file_frame...) */
    sdv_frame_stats *stats;         /* [n frames] */
    uint32_t *scratch;              /* per frame 2*height u32 (frame_valid / frame_invalid coordinate keys) */
    const uint8_t *frame_flags;     /* [n frames] SDV_FRAME_* of the caller (sdv_set_frame_flags), or NULL */
    struct SweepMemo *memo;         /* outcomes of reference-level sweeps and requests for more (stc007_sweep_device.h): the pool, ... */
    int32_t *memo_head;             /* ... [n_total * height] the newest entry of a line (-1: none), ... */
    int32_t *memo_count; int32_t memo_cap;   /* ... entries handed out (may run past the capacity: those requests were dropped) */
    unsigned long long *bw_memo;    /* [n_total * height] or NULL: what findBlackWhite found on a line, kept from one decode of a frame to the next (find_black_white) */
    struct SweepEnt *fat_levels;    /* sdv_k_stc007_frames_fat: 256 entries per workgroup of the launch (stc007_sweep_device.h, fat_sweep) */
    /* the fused entry (sdv_decode_frames), a tape that plays: a frame the whole-frame capture takes from end to end goes straight into the stitcher's field
     * buffers - 32-byte lines in field order, what sdv_k_stitch_analyze would make of the frame's 48-byte records - and leaves a summary instead of records */
    void *direct_fields;            /* the stitcher's field buffers (SLine[segments][2][direct_pitch]), or NULL */
    struct DirectFrame *direct_frames;      /* [n_total] what frame f left there (flag 0: nothing, its records are in recs) */
    int direct_seg_ofs, direct_pitch;       /* frame f is the stitcher's segment f + direct_seg_ofs; lines from one field buffer to the next */
    int direct_lines;                       /* lines a field buffer holds (the stitcher's BUF_FIELD: a longer field goes through its records, as on the record path) */
    /* The first round of a call on a tape that plays: every frame is started from the model's state (predict_half) behind ONE known state, `base` at frame
     * base_frame - the wave makes it itself instead of reading what a kernel in front of this one wrote to states_in (which then holds nothing yet: the
     * engine fills it in when the round was not the last, engine.inc). */
    uint8_t predict_in_kernel; int base_frame; sdv_v2d_state base;
};
/* what a frame that went straight into the field buffers tells the analysis kernel (stc007_stitch_device.h, analyze_body) */
struct DirectFrame { uint32_t frame_number; uint16_t n[2], bad[2]; uint8_t ref, flag, _pad[2]; };
enum { DSL_FORCED_BAD = 1, DSL_COORDS_VALID = 2, DSL_BW_SET = 4 };      /* SLine::flags as the stitcher names them (SL_*, checked there) */
/* a dropped frame: VideoInFFMPEG::insertDummyFrame(false, true) sends its lines as empty VideoLines (vin_ffmpeg.cpp:367-522) */
__device__ __forceinline__ bool frame_is_empty(const FrameArgs &a, int f) { return a.frame_flags && f < a.n_total && f != a.end_file_frame && (a.frame_flags[f] & SDV_FRAME_EMPTY); }

/* ---- uniform per-wave state (kept in registers; identical in all lanes) ---------------------- */
struct Coords { int16_t start, stop; bool doubled; };

struct Line {                       /* PCMLine + STC007Line (pcmline.h:137-166, stc007line.h:153-165) */
    uint32_t frame_number; uint16_t line_number;
    uint8_t black, white, ref_low, ref_level, ref_high;
    Coords coords;
    uint8_t hyst, shift;
    bool ref_sweeped, by_ext_tune;
    uint16_t calc_crc;
    bool bw_set, coords_set, forced_bad;
    uint8_t service;
    uint16_t pixel_start, pixel_stop;
    int16_t pso; uint32_t psm, hpsm;
    uint8_t mark_st, mark_ed;
    uint16_t m_st_bg, m_st_ed, m_sp_ed;
    bool m2;
    bool word_crc07, word_valid07, word_crc8, word_valid8;   /* flags of words 0..7 / of the CRC word */
    uint16_t words[9];
};

struct Bin {                        /* Binarizer (binarizer.h:306-337) */
    uint8_t in_black, in_white, in_ref;
    Coords in_coord;
    uint8_t in_max_hyst, in_max_shift;
    bool do_ref_lvl_sweep;
    uint8_t mode, hyst_lim, shift_lim;
    uint16_t line_length, scan_start, scan_end, mark_start_max, mark_end_min, estimated_ppb;
    bool was_bw_scanned;
    bool vl_doubled;
};

struct Markers { uint8_t st_stage, ed_stage; uint16_t st1s, st1e, st3e, ed_start, ed_end; bool has_start; };

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
/* Every kernel of the binarize paths is a workgroup of ONE wavefront.  What its phases need between them is that the LDS writes of some lanes come
 * before the LDS reads of others - which the LDS gives a wave for nothing (its instructions are carried out in order); the compiler only has to keep
 * the order.  __syncthreads() does more: it waits for every memory operation the wave has in flight (s_waitcnt vmcnt(0)) - the next row on its way
 * from HBM, the last record on its way out - before every phase of every line: a line taken one at a time cost 38 000 cycles, 33 000 of them
 * waiting at these barriers.  SDV_WAVE_SYNC orders without waiting (as stc007_stitch_device.h has done since round 2). */
#ifdef SDV_EMU
#define SDV_WAVE_SYNC() __syncthreads()
#else
#define SDV_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
/* ... and where the wave does have to wait for what it has in flight (lanes that read from global memory what other lanes of the wave stored): all that
 * __syncthreads() is for a workgroup of one wave (the compiler drops the barrier itself there), spelled without the barrier - the same code in the one-wave
 * kernels, and no rendezvous with the waves beside it in sdv_k_stc007_frames_fat, whose barriers are its protocol (stc007_sweep_device.h, fat_sweep). */
#ifdef SDV_EMU
#define SDV_BLOCK_SYNC() __syncthreads()
#else
#define SDV_BLOCK_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif
#ifndef SDV_OPAQUE
#ifdef SDV_EMU
#define SDV_OPAQUE(x) ((void)0)
#else
#define SDV_OPAQUE(x) asm volatile("" : "+v"(x))      /* the compiler knows nothing about x from here on: what is derived from it is worked out behind this point */
#endif
#endif
#ifdef SDV_K1_STAMPS        /* developer aid (variant builds only): cycles per part of a frame, summed over the frames of a launch (0..7 the frame loop, 8..15 the general path) */
__device__ unsigned long long sdv_k1_cycles[24];
/* summed per frame in LDS and added to the totals once, at the end of the frame: an atomic per stamp on one address held up every load
 * behind it (a stamped build ran 10-70 % slower than the plain one) */
__shared__ unsigned long long sdv_k1_lds[24];
#define K1_T(var) const unsigned long long var = __builtin_readcyclecounter()
#define K1_ADD(i, t0, t1) do { if (lane_id() == 0) sdv_k1_lds[i] += (t1) - (t0); } while (0)
#define K1_BEGIN() do { if (lane_id() < 24) sdv_k1_lds[lane_id()] = 0; SDV_WAVE_SYNC(); } while (0)
#define K1_FLUSH() do { SDV_WAVE_SYNC(); if (lane_id() < 24 && lane_id() != 7 && sdv_k1_lds[lane_id()] != 0) atomicAdd(&sdv_k1_cycles[lane_id()], sdv_k1_lds[lane_id()]); } while (0)
#else
#define K1_T(var) do { } while (0)
#define K1_ADD(i, t0, t1) do { } while (0)
#define K1_BEGIN() do { } while (0)
#define K1_FLUSH() do { } while (0)
#endif

/* Wave-uniform values that reach us through vector memory (global/LDS/scratch loads) are re-declared
 * uniform so the compiler keeps them in SGPRs and branches on them with scalar branches. */
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uniu(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ bool coords_valid(const Coords &c)    /* frametrimset.cpp:153-156 */
{
    return (c.start != NO_COORD_LEFT) && (c.stop != NO_COORD_RIGHT) && (c.start < c.stop);
}
__device__ __forceinline__ void coords_clear(Coords &c) { c.start = NO_COORD_LEFT; c.stop = NO_COORD_RIGHT; c.doubled = false; }
__device__ __forceinline__ bool coords_set(Coords &c, int16_t s, int16_t e) { if (e > s) { c.start = s; c.stop = e; return true; } return false; }
__device__ __forceinline__ bool coords_ne(const Coords &a, const Coords &b) { return a.start != b.start || a.stop != b.stop || a.doubled != b.doubled; }
/* sort key of CoordinatePair::operator< (frametrimset.cpp:63-98): start ascending, then stop descending
 * (reference field is always 0 on this path) */
__device__ __forceinline__ uint32_t coords_key(int16_t s, int16_t e) { return ((uint32_t)(uint16_t)(s + 32768) << 16) | (uint32_t)(uint16_t)(0xFFFF - (uint16_t)(e + 32768)); }
__device__ __forceinline__ int16_t key_start(uint32_t k) { return (int16_t)((int)(k >> 16) - 32768); }
__device__ __forceinline__ int16_t key_stop(uint32_t k) { return (int16_t)((int)(0xFFFF - (k & 0xFFFF)) - 32768); }

__device__ __forceinline__ uint8_t get_low_level(uint8_t l, uint8_t d) { return (l > d) ? (uint8_t)(l - d) : (uint8_t)1; }            /* binarizer.cpp:3476-3487 */
__device__ __forceinline__ uint8_t get_high_level(uint8_t l, uint8_t d) { return (l < (255 - d)) ? (uint8_t)(l + d) : (uint8_t)254; }  /* :3490-3501 */

__device__ inline uint8_t pick_center_ref_level(const sdv_bin_preset &ps, uint8_t lvl_black, uint8_t lvl_white)   /* :3504-3548 */
{
    uint8_t br_delta = (uint8_t)(lvl_white - lvl_black), res;
    if (br_delta >= ps.min_contrast) {
        br_delta = br_delta / 2;
        res = (uint8_t)(br_delta + lvl_black);
        if (res < ps.min_ref_lvl) res = ps.min_ref_lvl;
        else if (res > ps.max_ref_lvl) res = ps.max_ref_lvl;
    } else {
        res = (lvl_white < ps.max_ref_lvl) ? ps.max_ref_lvl : ps.min_ref_lvl;
    }
    return res;
}

/* ---- CRC helpers (bit-serial form for the few scalar uses) ----------------------------------- */
__device__ inline uint16_t crc16_words(const uint16_t *w)     /* stc007line.cpp:245-251 */
{
    uint16_t crc = 0xFFFF;
    for (int k = 0; k < 8; k++)
        for (int bit = 13; bit >= 0; bit--) crc = crc16_step(crc, (w[k] >> bit) & 1);
    return crc;
}

/* ---- Line helpers ---------------------------------------------------------------------------- */
__device__ inline void pcmline_clear(Line &l)      /* PCMLine::clear, pcmline.cpp:96-116 */
{
    l.frame_number = 0; l.line_number = 0;
    l.black = l.white = 0; l.ref_low = l.ref_level = l.ref_high = 0;
    coords_clear(l.coords);
    l.hyst = l.shift = 0;
    l.ref_sweeped = l.by_ext_tune = false;
    l.calc_crc = 0;
    l.bw_set = l.coords_set = l.forced_bad = false;
    l.service = SDV_SRV_NO;
    l.pixel_start = 0; l.pixel_stop = 1; l.pso = 0; l.psm = 128; l.hpsm = 64;
}
__device__ inline void stc_clear(Line &l)          /* STC007Line::clear, stc007line.cpp:64-93 */
{
    pcmline_clear(l);
    l.mark_st = MARK_ST_START; l.mark_ed = MARK_ED_START;
    l.m_st_bg = l.m_st_ed = l.m_sp_ed = 0;
    l.m2 = false;
    for (int i = 0; i < 8; i++) l.words[i] = 0;
    l.word_crc07 = l.word_valid07 = l.word_crc8 = l.word_valid8 = false;
    l.calc_crc = CRC_SILENT;
    l.words[8] = (uint16_t)~l.calc_crc;
}
__device__ __forceinline__ bool crc_valid_ignore_forced(const Line &l) { return l.calc_crc == l.words[8]; }
__device__ __forceinline__ bool crc_valid(const Line &l) { return !l.forced_bad && crc_valid_ignore_forced(l); }
__device__ __forceinline__ bool has_start(const Line &l) { return l.mark_st == MARK_ST_BOT_2; }
__device__ __forceinline__ bool has_stop(const Line &l) { return l.mark_ed == MARK_ED_LEN_OK; }
__device__ __forceinline__ bool has_markers(const Line &l) { return has_start(l) && has_stop(l); }
__device__ __forceinline__ void set_invalid_crc(Line &l) { l.words[8] = (uint16_t)~l.calc_crc; }
__device__ __forceinline__ void apply_crc_state_per_word(Line &l) { bool v = crc_valid(l); l.word_crc07 = l.word_valid07 = l.word_crc8 = l.word_valid8 = v; }
__device__ inline void set_source_pixels(Line &l, uint16_t a, uint16_t b)   /* pcmline.cpp:202-213 */
{
    if (b > a) if (BITS_BETWEEN <= (b - a)) { l.pixel_start = a; l.pixel_stop = b; }
}
__device__ inline void set_service(Line &l, uint8_t srv)   /* pcmline.cpp:118-171, 489-503 (base clear only) */
{
    uint32_t f = l.frame_number; uint16_t n = l.line_number;
    pcmline_clear(l);
    l.frame_number = f; l.line_number = n; l.service = srv;
}
__device__ inline int16_t stc_get_sample(const Line &l, int index)   /* stc007line.cpp:282-326 */
{
    uint16_t w = l.words[index];
    if (!l.m2) w = (uint16_t)(w << 2);
    else {
        if ((w & (1 << 13)) == 0) w = (uint16_t)(w << 3);
        else {
            bool pos = (w & (1 << 12)) == 0;
            w = (uint16_t)(w & ~(1 << 13));
            if (!pos) w |= (1 << 15) | (1 << 14) | (1 << 13);
        }
    }
    return (int16_t)w;
}
__device__ inline bool stc_is_almost_silent(const Line &l)   /* stc007line.cpp:568-599 */
{
    int n = 0;
    for (int i = 0; i <= 5; i++) { int16_t s = stc_get_sample(l, i); if (!(s >= 16) && !(s < -16)) n++; }
    return n >= 2;
}
__device__ inline bool has_control_block(const Line &l)      /* stc007line.cpp:493-504 */
{
    return l.words[0] == 0x3333 && l.words[1] == 0x0CCC && l.words[2] == 0x3333 && l.words[3] == 0x0CCC && l.words[4] == 0 && (l.words[7] & 0x0FF0) == 0;
}
__device__ inline void set_serv_ctrl_blk(Line &l)             /* stc007line.cpp:96-129 */
{
    uint16_t w4 = l.words[4], w5 = l.words[5], w6 = l.words[6], w7 = l.words[7];
    uint32_t f = l.frame_number; uint16_t n = l.line_number;
    stc_clear(l);
    l.frame_number = f; l.line_number = n;
    l.words[4] = w4; l.words[5] = w5; l.words[6] = w6; l.words[7] = w7;
    l.calc_crc = crc16_words(l.words);
    l.words[8] = l.calc_crc;
    l.service = SDV_SRV_CTRL_BLOCK;
}

__device__ inline void line_to_rec(const Line &l, sdv_line_rec *r)
{
    r->frame_number = l.frame_number; r->line_number = l.line_number;
    for (int i = 0; i < 9; i++) r->words[i] = l.words[i];
    r->calc_crc = l.calc_crc;
    r->data_start = l.coords.start; r->data_stop = l.coords.stop;
    r->marker_start_bg_coord = l.m_st_bg; r->marker_start_ed_coord = l.m_st_ed; r->marker_stop_ed_coord = l.m_sp_ed;
    r->black_level = l.black; r->white_level = l.white; r->ref_low = l.ref_low; r->ref_level = l.ref_level; r->ref_high = l.ref_high;
    r->hysteresis_depth = l.hyst; r->shift_stage = l.shift; r->service_type = l.service;
    r->mark_st_stage = l.mark_st; r->mark_ed_stage = l.mark_ed;
    uint8_t f = 0;
    if (l.ref_sweeped) f |= SDV_LF_REF_SWEEPED;
    if (l.by_ext_tune) f |= SDV_LF_BY_EXT_TUNE;
    if (l.bw_set) f |= SDV_LF_BW_SET;
    if (l.coords_set) f |= SDV_LF_COORDS_SET;
    if (l.forced_bad) f |= SDV_LF_FORCED_BAD;
    if (crc_valid(l)) f |= SDV_LF_CRC_VALID;
    if (l.coords.doubled) f |= SDV_LF_FROM_DOUBLED;
    r->flags = f;
    uint8_t ws = 0;
    if (!l.forced_bad && l.word_crc07) ws |= SDV_WS_WORD_CRC;
    if (!l.forced_bad && l.word_valid07) ws |= SDV_WS_WORD_VALID;
    r->word_state = ws;
}

/* ======================================================================================== */
/* Bit extraction                                                                            */
/* ======================================================================================== */

/* PCMLine::setPPB (pcmline.cpp:506-519) */
__device__ inline void set_ppb(Line &l, const Coords &c)
{
    l.psm = (uint32_t)((int)c.stop - (int)c.start);
    l.psm = (l.psm * 128u + BITS_BETWEEN / 2) / BITS_BETWEEN;
    l.pso = c.start;
    l.hpsm = (l.psm + 1) / 2;
}
/* PCMLine::getVideoPixeBylCalc (pcmline.cpp:249-311) for data bit `bit` (+3 START bits), before the
 * per-stage pixel shift */
__device__ __forceinline__ int32_t bit_center(const Line &l, int bit)
{
    uint32_t pcm_bit = (uint32_t)(bit + (BITS_START - 1));
    int32_t vp = (int32_t)(pcm_bit * l.psm + l.hpsm);
    vp = vp / 128;
    return vp + l.pso;
}
__device__ __forceinline__ int32_t shift_clamp(const Line &l, int32_t vp, int stage)
{
    /* PIX_SH_BG_TBL == PIX_SH_ED_TBL == {0,+1,-1,+2,-2} (pcmline.h:63-71): the shift is uniform */
    int sh = (stage == 0) ? 0 : ((stage & 1) ? ((stage + 1) >> 1) : -(stage >> 1));
    vp += sh;
    if (vp < (int32_t)l.pixel_start) vp = l.pixel_start;
    else if (vp >= (int32_t)l.pixel_stop) vp = (int32_t)l.pixel_stop - 1;
    return vp;
}

/* number of set bits of `m` strictly below this lane */
__device__ __forceinline__ int prefix_count(uint64_t m)
{
    return __popcll(m & ((1ull << lane_id()) - 1ull));
}

/* Solves s[b] = s[b-1] ? B[b] : A[b], s[-1] = 0, for the 128 cells (the hysteresis automaton of
 * Binarizer::fillSTC007, binarizer.cpp:7377-7399: A = "px > low", B = "px >= high").
 * A==B cells force the state, A!=B cells keep (A=0,B=1) or toggle (A=1,B=0) it, so
 *   s[b] = C[j] ^ parity(toggles in (j, b])   with j = last forcing cell <= b (or the initial 0).
 * With PT = inclusive prefix parity of the toggle mask: s = PT ^ fill_forward((C ^ PT) at forcing cells).
 * The fill-forward is done with one 128-bit integer addition (carry ripples through the non-forcing runs). */
__device__ inline void solve_automaton(uint64_t a_lo, uint64_t a_hi, uint64_t b_lo, uint64_t b_hi, uint64_t &s_lo, uint64_t &s_hi)
{
    uint64_t e_lo = ~(a_lo ^ b_lo), e_hi = ~(a_hi ^ b_hi);       /* forcing cells */
    uint64_t t_lo = a_lo & ~b_lo, t_hi = a_hi & ~b_hi;           /* toggling cells */
    /* inclusive prefix parity of t, lane i owns cells i and i+64 */
    int lane = lane_id();
    const uint64_t le = ((1ull << lane) - 1ull) | (1ull << lane);       /* inclusive */
    int c_lo = __popcll(t_lo & le);
    int c_hi = __popcll(t_lo) + __popcll(t_hi & le);
    uint64_t pt_lo = __ballot(c_lo & 1);
    uint64_t pt_hi = __ballot(c_hi & 1);
    uint64_t u_lo = ((a_lo & b_lo) ^ pt_lo) & e_lo, u_hi = ((a_hi & b_hi) ^ pt_hi) & e_hi;   /* head values */
    uint64_t x_lo = u_lo | ~e_lo, x_hi = u_hi | ~e_hi;
#ifdef SDV_EMU
    uint64_t y_lo = x_lo + u_lo;
    uint64_t carry = (y_lo < x_lo) ? 1ull : 0ull;
    uint64_t y_hi = x_hi + u_hi + carry;
#else
    /* one scalar add-with-carry chain over the four 32-bit limbs (there is no 64-bit unsigned compare on the scalar unit to
     * recover the carry of a 64-bit add, and the compiler re-materialises SCC between the limbs when left to itself) */
    uint32_t y0, y1, y2, y3;
    asm("s_add_u32 %0, %4, %8\n\ts_addc_u32 %1, %5, %9\n\ts_addc_u32 %2, %6, %10\n\ts_addc_u32 %3, %7, %11"
                 : "=&s"(y0), "=&s"(y1), "=&s"(y2), "=&s"(y3)
                 : "s"((uint32_t)x_lo), "s"((uint32_t)(x_lo >> 32)), "s"((uint32_t)x_hi), "s"((uint32_t)(x_hi >> 32)),
                   "s"((uint32_t)u_lo), "s"((uint32_t)(u_lo >> 32)), "s"((uint32_t)u_hi), "s"((uint32_t)(u_hi >> 32))
                 : "scc");
    uint64_t y_lo = ((uint64_t)y1 << 32) | y0, y_hi = ((uint64_t)y3 << 32) | y2;
#endif
    uint64_t d_lo = ((y_lo ^ x_lo) & ~e_lo) | u_lo;
    uint64_t d_hi = ((y_hi ^ x_hi) & ~e_hi) | u_hi;
    s_lo = d_lo ^ pt_lo;
    s_hi = d_hi ^ pt_hi;
}
/* the same for one LANE on its own masks (no collectives): the prefix parity by doubling shifts */
__device__ __forceinline__ uint64_t prefix_xor64(uint64_t x) { x ^= x << 1; x ^= x << 2; x ^= x << 4; x ^= x << 8; x ^= x << 16; x ^= x << 32; return x; }
__device__ inline void solve_automaton_lane(uint64_t a_lo, uint64_t a_hi, uint64_t b_lo, uint64_t b_hi, uint64_t &s_lo, uint64_t &s_hi)
{
    const uint64_t e_lo = ~(a_lo ^ b_lo), e_hi = ~(a_hi ^ b_hi), t_lo = a_lo & ~b_lo, t_hi = a_hi & ~b_hi;
    const uint64_t pt_lo = prefix_xor64(t_lo), pt_hi = prefix_xor64(t_hi) ^ ((__popcll(t_lo) & 1) ? ~0ull : 0ull);
    const uint64_t u_lo = ((a_lo & b_lo) ^ pt_lo) & e_lo, u_hi = ((a_hi & b_hi) ^ pt_hi) & e_hi;
    const uint64_t x_lo = u_lo | ~e_lo, x_hi = u_hi | ~e_hi;
    const uint64_t y_lo = x_lo + u_lo, y_hi = x_hi + u_hi + ((y_lo < x_lo) ? 1ull : 0ull);
    s_lo = (((y_lo ^ x_lo) & ~e_lo) | u_lo) ^ pt_lo;
    s_hi = (((y_hi ^ x_hi) & ~e_hi) | u_hi) ^ pt_hi;
}

/* The two comparison masks (px > low, px >= high) of kCount <= 32 consecutive cells for ONE lane on its own (the candidate reads of the
 * marker-less coordinate searches): cell k is sampled at pixel (acc >> 7) clamped to [lo, hi], acc advancing by psm per cell (acc holds
 * cell * psm + psm / 2 + 128 * (data start + pixel shift): the cell centre of getVideoPixeBylCalc, pcmline.cpp:249-311, in 1/128 pixel).
 * Written for the issue rate: the bytes are fetched eight at a time before they are used, a comparison is one subtraction whose sign is
 * shifted into the mask (v_alignbit), no 64-bit shifts.  First cell in bit 0. */
__device__ __forceinline__ uint32_t shl1_sign(uint32_t m, int32_t t)
{
#ifdef SDV_EMU
    return (m << 1) | ((uint32_t)t >> 31);
#else
    return __builtin_amdgcn_alignbit(m, (uint32_t)t, 31);
#endif
}
template <int kN>
__device__ __forceinline__ void compare_cells_block(const uint8_t *px_row, int32_t &acc, int32_t psm, int32_t lo, int32_t hi, int32_t ref_low, int32_t rh1,
                                                    uint32_t &am, uint32_t &bm)
{
    int32_t px[kN];
#pragma unroll
    for (int j = 0; j < kN; j++) {
        int32_t vp = acc >> 7; acc += psm;
        vp = vp < lo ? lo : (vp > hi ? hi : vp);
        px[j] = px_row[vp];
    }
#pragma unroll
    for (int j = 0; j < kN; j++) { am = shl1_sign(am, ref_low - px[j]); bm = shl1_sign(bm, rh1 - px[j]); }
}
#ifndef SDV_CELLS_UNROLL
#define SDV_CELLS_UNROLL 4          /* blocks of eight cells unrolled per mask (4 = all; 1 = a loop: 1-3 % slower on the PCM frame drivers, measured) */
#endif
template <int kCount>
__device__ __forceinline__ void compare_cells32(const uint8_t *px_row, int32_t &acc, int32_t psm, int32_t lo, int32_t hi, int32_t ref_low, int32_t ref_high,
                                                uint32_t &a, uint32_t &b)
{
    static_assert(kCount > 0 && kCount <= 32, "one 32-bit mask");
    uint32_t am = 0, bm = 0;
    const int32_t rh1 = ref_high - 1;           /* px >= high  <=>  high - 1 - px < 0 */
#pragma unroll SDV_CELLS_UNROLL
    for (int blk = 0; blk < kCount / 8; blk++) compare_cells_block<8>(px_row, acc, psm, lo, hi, ref_low, rh1, am, bm);
    if (kCount % 8) compare_cells_block<(kCount % 8) ? (kCount % 8) : 1>(px_row, acc, psm, lo, hi, ref_low, rh1, am, bm);
    a = __brev(am) >> (32 - kCount); b = __brev(bm) >> (32 - kCount);
}
__device__ __forceinline__ int32_t shift_of_stage(int stage) { return stage == 0 ? 0 : (stage == 1 ? 1 : (stage == 2 ? -1 : (stage == 3 ? 2 : -2))); }

__device__ __forceinline__ uint16_t rev14(uint32_t v) { return (uint16_t)(__brev(v) >> 18); }
__device__ __forceinline__ uint16_t rev16(uint32_t v) { return (uint16_t)(__brev(v) >> 16); }

/* Binarizer::fillSTC007 (binarizer.cpp:7322-7445) for the whole wave: samples the 128 cells from the
 * LDS-staged scanline at pixel-shift `stage`, packs the words MSB first and recomputes the CRC. */
__device__ inline void fill_stc007(Line &l, const WaveLds &lds, int32_t vp0, int32_t vp1, int stage)
{
    uint8_t p0 = lds.px[shift_clamp(l, vp0, stage)];
    uint8_t p1 = lds.px[shift_clamp(l, vp1, stage)];
    uint64_t a_lo = __ballot(p0 > l.ref_low), b_lo = __ballot(p0 >= l.ref_high);
    uint64_t a_hi = __ballot(p1 > l.ref_low), b_hi = __ballot(p1 >= l.ref_high);
    uint64_t s_lo, s_hi;
    solve_automaton(a_lo, a_hi, b_lo, b_hi, s_lo, s_hi);
    /* words: cell 14k..14k+13 -> word k, first cell = MSB */
    l.words[0] = rev14((uint32_t)(s_lo & 0x3FFF));
    l.words[1] = rev14((uint32_t)((s_lo >> 14) & 0x3FFF));
    l.words[2] = rev14((uint32_t)((s_lo >> 28) & 0x3FFF));
    l.words[3] = rev14((uint32_t)((s_lo >> 42) & 0x3FFF));
    l.words[4] = rev14((uint32_t)(((s_lo >> 56) | (s_hi << 8)) & 0x3FFF));
    l.words[5] = rev14((uint32_t)((s_hi >> 6) & 0x3FFF));
    l.words[6] = rev14((uint32_t)((s_hi >> 20) & 0x3FFF));
    l.words[7] = rev14((uint32_t)((s_hi >> 34) & 0x3FFF));
    l.words[8] = rev16((uint32_t)((s_hi >> 48) & 0xFFFF));
    l.word_crc07 = l.word_valid07 = false;            /* setWord(index, word) clears the flags of words 0..7 */
    int lane = lane_id();
    uint64_t klo = (lane < 16) ? c_crc.klo[lane & 15] : 0ull;
    uint64_t khi = (lane < 16) ? c_crc.khi[lane & 15] : 0ull;
    int par = (__popcll(s_lo & klo) + __popcll(s_hi & khi)) & 1;
    uint64_t cb = __ballot(par);
    l.calc_crc = (uint16_t)((uint16_t)(cb & 0xFFFF) ^ c_crc.init);
}

/* Binarizer::fillDataWords (binarizer.cpp:7560-7691) */
__device__ inline bool fill_data_words(Line &l, const WaveLds &lds, int32_t vp0, int32_t vp1, uint8_t ref_delta, uint8_t shift_stg)
{
    if (ref_delta > HYST_DEPTH_MAX) return false;
    if (shift_stg > SHIFT_STAGES_MAX) return false;
    uint8_t low_ref = get_low_level(l.ref_level, ref_delta), high_ref = get_high_level(l.ref_level, ref_delta);
    l.ref_low = low_ref; l.ref_high = high_ref;
    if (low_ref <= l.black) { set_invalid_crc(l); return false; }
    if (high_ref >= l.white) { set_invalid_crc(l); return false; }
    l.hyst = ref_delta; l.shift = shift_stg;
    fill_stc007(l, lds, vp0, vp1, shift_stg);
    return true;
}

/* Binarizer::readPCMdata (binarizer.cpp:7695-8055).  The CRC bookkeeping arrays of the reference
 * (shift_crcs/hyst_crcs/crc_stats) reduce, for a single candidate, to "first (hysteresis, shift) pair in
 * lexicographic order with a valid CRC, else (0,0)"; the ladder stops at the first hysteresis depth whose
 * levels leave (black, white). */
__device__ inline void read_pcm_data(Bin &b, Line &l, const WaveLds &lds, bool ladder_known_to_fail = false)
{
    set_ppb(l, l.coords);
    int lane = lane_id();
    int32_t vp0 = bit_center(l, lane), vp1 = bit_center(l, lane + 64);
    if (b.hyst_lim > HYST_DEPTH_MAX) b.hyst_lim = HYST_DEPTH_MAX;
    if (b.shift_lim > SHIFT_STAGES_MAX) b.shift_lim = SHIFT_STAGES_MAX;
    uint8_t valid_delta, valid_shift;
    if (!l.ref_sweeped) {
        valid_delta = 0; valid_shift = 0;
        bool found = false, last_is_target = false;
        /* (ladder_known_to_fail: the frame loop has been through exactly these reads for this line - fast_line - and none came out valid; what is left is the closing read) */
        for (int h = 0; h <= (int)b.hyst_lim && !found && !ladder_known_to_fail; h++) {
            bool invalid_hyst = false;
            for (int s = 0; s <= (int)b.shift_lim; s++) {
                if (!fill_data_words(l, lds, vp0, vp1, (uint8_t)h, (uint8_t)s)) { invalid_hyst = true; break; }
                if (crc_valid(l)) { found = true; valid_delta = (uint8_t)h; valid_shift = (uint8_t)s; last_is_target = true; break; }
            }
            if (invalid_hyst) break;
        }
        if (last_is_target) return;    /* the final re-read with the winning pair is idempotent */
    } else {
        valid_delta = b.hyst_lim; valid_shift = b.shift_lim;
    }
    fill_data_words(l, lds, vp0, vp1, valid_delta, valid_shift);
}

/* ======================================================================================== */
/* Marker search                                                                             */
/* ======================================================================================== */

/* Binarizer::searchSTC007Markers (binarizer.cpp:5275-5595) for one hysteresis level; pure function of the
 * staged scanline, the line's ref_level and the scan limits.  Runs independently in every lane. */
enum { SWEEP_WINDOW_MAX = 192 };    /* pixels of a marker search window the mask forms below cover (a 720-px line: 68 / 73; doubled: 136 / 146) */
__device__ __forceinline__ int m192_first_set(const uint64_t *w, int p)        /* the lowest set bit at or above p; SWEEP_WINDOW_MAX when there is none */
{
    int r = SWEEP_WINDOW_MAX;
#pragma unroll
    for (int k = 2; k >= 0; k--) {
        const int lo = p - 64 * k;
        if (lo > 63) continue;
        uint64_t x = w[k];
        if (lo > 0) x &= ~0ull << lo;
        if (x) r = 64 * k + __ffsll((unsigned long long)x) - 1;
    }
    return r;
}
__device__ __forceinline__ int m192_first_clear(const uint64_t *w, int p)
{
    const uint64_t n[3] = { ~w[0], ~w[1], ~w[2] };
    return m192_first_set(n, p);
}

__device__ inline Markers search_markers_px(const Bin &b, const sdv_bin_preset &ps, const uint8_t *px, uint8_t ref_level, uint8_t hyst_lvl)
{
    Markers m;
    uint8_t stage = MARK_ST_START, pv;
    uint8_t bin_level = ref_level;
    uint8_t bin_low = get_low_level(bin_level, hyst_lvl);
    if (bin_low < ps.min_ref_lvl) bin_low = ps.min_ref_lvl;
    uint8_t bin_high = bin_level;
    uint16_t st1s = 0, st1e = 0, st3s = 0, st3e = 0;
    uint16_t pixel_limit = (uint16_t)(b.mark_start_max + b.estimated_ppb * 5);
    if (pixel_limit > b.line_length) pixel_limit = b.line_length;
    uint16_t pixel = b.scan_start;
    while (pixel < pixel_limit) {
        pv = px[pixel];
        if (stage == MARK_ST_START) {
            if (pixel > b.mark_start_max) break;
            if (pv >= bin_low) { st1s = pixel; stage = MARK_ST_TOP_1; }
        } else if (stage == MARK_ST_TOP_1) {
            if (pv < bin_low) { st1e = pixel; stage = MARK_ST_BOT_1; }
        } else if (stage == MARK_ST_BOT_1) {
            if (pv >= bin_high) {
                st3s = pixel;
                if (((st3s - st1e) > (b.estimated_ppb * 2)) || ((st3s - st1e) < (b.estimated_ppb / 2))) stage = MARK_ST_START;
                else stage = MARK_ST_TOP_2;
            }
        } else if (stage == MARK_ST_TOP_2) {
            if (pv < bin_high) {
                st3e = pixel;
                if (((st3e - st3s) > (b.estimated_ppb * 2)) || ((st3e - st3s) < (b.estimated_ppb / 2))) stage = MARK_ST_START;
                else { stage = MARK_ST_BOT_2; break; }
            }
        }
        pixel++;
    }
    m.st_stage = stage; m.st1s = st1s; m.st1e = st1e; m.st3e = st3e;
    m.has_start = (stage == MARK_ST_BOT_2);
    stage = MARK_ED_START;
    uint16_t ed_start = 0, ed_end = 0;
    if (m.has_start) {
        bin_low = bin_level;
        if (b.mark_end_min > (b.estimated_ppb * 6)) pixel_limit = (uint16_t)(b.mark_end_min - b.estimated_ppb * 6);
        else pixel_limit = 0;
        pixel = b.scan_end;
        while (pixel > pixel_limit) {
            pv = px[pixel];
            if (stage == MARK_ED_START) {
                if (pixel < b.mark_end_min) break;
                if (pv >= bin_low) { ed_end = (uint16_t)(pixel + 1); stage = MARK_ED_TOP; }
            } else if (stage == MARK_ED_TOP) {
                if (pv < bin_high) {
                    ed_start = (uint16_t)(pixel + 1);
                    stage = MARK_ED_BOT;
                    if (((ed_end - ed_start) >= (b.estimated_ppb * 2)) && ((ed_end - ed_start) <= (b.estimated_ppb * 5))) { stage = MARK_ED_LEN_OK; break; }
                    else stage = MARK_ED_START;
                }
            }
            pixel--;
        }
    }
    m.ed_stage = stage; m.ed_start = ed_start; m.ed_end = ed_end;
    return m;
}

/* applies a search result to a line exactly as the tail of searchSTC007Markers does (:5459-5594) */
__device__ inline void apply_markers(Line &l, const Markers &m)
{
    l.mark_st = m.st_stage; l.m_st_bg = m.st1s; l.m_st_ed = m.st3e;
    if (m.has_start) l.mark_ed = m.ed_stage;
    coords_set(l.coords, (int16_t)m.st1e, (int16_t)m.ed_start);
    l.m_sp_ed = m.ed_end;
    l.coords_set = has_markers(l);
}

/* searchSTC007Markers for the 24 hysteresis depths of one reference level at once, lane = depth: the same two state machines as search_markers_px,
 * stepping from edge to edge on threshold masks instead of from pixel to pixel (a lane read some 170 pixels from LDS one behind the other: 17 000
 * cycles a search).  The masks - pixel >= threshold over the START window [0, n_start) and, mirrored, over the STOP window - are made by the wave
 * (a lane per pixel, a ballot per threshold: the 24 low levels and the reference level itself); every lane keeps the one of its depth.  What the
 * machines leave behind when they do not get through (stage, the coordinates of earlier attempts) is carried along as the pixel loop does. */
__device__ inline Markers search_markers_masks(const Bin &b, const sdv_bin_preset &ps, const uint8_t *px, uint8_t ref_level, int n_start, int n_stop)
{
    const int lane = lane_id();
    const int ppb = b.estimated_ppb, scan_end = b.scan_end;
    uint64_t L[3] = { 0, 0, 0 }, H[3] = { 0, 0, 0 }, R[3] = { 0, 0, 0 };
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (64 * c >= n_start && 64 * c >= n_stop) break;
        const int p = 64 * c + lane;
        const int vs = p < n_start ? (int)px[b.scan_start + p] : -1, ve = p < n_stop ? (int)px[scan_end - p] : -1;
        H[c] = __ballot(vs >= (int)ref_level); R[c] = __ballot(ve >= (int)ref_level);
        for (int d = 0; d < 24; d++) {
            int t = get_low_level(ref_level, (uint8_t)d);
            if (t < ps.min_ref_lvl) t = ps.min_ref_lvl;
            const uint64_t m = __ballot(vs >= t);
            L[c] = lane == d ? m : L[c];
        }
    }
    Markers m;
    uint8_t stage = MARK_ST_START;
    int st1s = -(int)b.scan_start, st1e = -(int)b.scan_start, st3e = -(int)b.scan_start;     /* (0 once the window's start is added: never assigned) */
    const int max_rel = (int)b.mark_start_max - (int)b.scan_start;
    for (int p = 0;;) {                                     /* forward: "1010" (binarizer.cpp:5310-5405) */
        const int q1 = m192_first_set(L, p);
        if (q1 >= n_start || q1 > max_rel) { stage = MARK_ST_START; break; }
        st1s = q1; stage = MARK_ST_TOP_1;
        const int q2 = m192_first_clear(L, q1 + 1);
        if (q2 >= n_start) break;
        st1e = q2; stage = MARK_ST_BOT_1;
        const int q3 = m192_first_set(H, q2 + 1);
        if (q3 >= n_start) break;
        if ((q3 - q2) > ppb * 2 || (q3 - q2) < ppb / 2) { stage = MARK_ST_START; p = q3 + 1; if (p >= n_start) break; continue; }
        stage = MARK_ST_TOP_2;
        const int q4 = m192_first_clear(H, q3 + 1);
        if (q4 >= n_start) break;
        st3e = q4;
        if ((q4 - q3) > ppb * 2 || (q4 - q3) < ppb / 2) { stage = MARK_ST_START; p = q4 + 1; if (p >= n_start) break; continue; }
        stage = MARK_ST_BOT_2;
        break;
    }
    m.st_stage = stage; m.st1s = (uint16_t)(b.scan_start + st1s); m.st1e = (uint16_t)(b.scan_start + st1e); m.st3e = (uint16_t)(b.scan_start + st3e);
    m.has_start = stage == MARK_ST_BOT_2;
    stage = MARK_ED_START;
    int ed_start = 0, ed_end = 0;
    if (m.has_start) {                                      /* backward: "01111" (:5410-5455) */
        const int i_max = scan_end - (int)b.mark_end_min;
        for (int i = 0;;) {
            const int j1 = m192_first_set(R, i);
            if (j1 >= n_stop || j1 > i_max) { stage = MARK_ED_START; break; }
            ed_end = scan_end - j1 + 1; stage = MARK_ED_TOP;
            const int j2 = m192_first_clear(R, j1 + 1);
            if (j2 >= n_stop) break;
            ed_start = scan_end - j2 + 1;
            if ((j2 - j1) >= ppb * 2 && (j2 - j1) <= ppb * 5) { stage = MARK_ED_LEN_OK; break; }
            stage = MARK_ED_START;
            i = j2 + 1;
            if (i >= n_stop) break;
        }
    }
    m.ed_stage = stage; m.ed_start = (uint16_t)ed_start; m.ed_end = (uint16_t)ed_end;
    return m;
}

/* Binarizer::findSTC007Coordinates (binarizer.cpp:6047-6113): lanes 0..23 each try one hysteresis level */
__device__ inline void find_coordinates_wave(const Bin &b, const sdv_bin_preset &ps, const WaveLds &lds, Line &l)
{
    int lane = lane_id();
    uint8_t h = (uint8_t)(lane < 24 ? lane : 23);
    /* the windows of the two searches (binarizer.cpp:5300-5308, :5408-5418) */
    int n_start = (b.mark_start_max + b.estimated_ppb * 5) & 0xFFFF; if (n_start > b.line_length) n_start = b.line_length;
    n_start -= b.scan_start;
    const int end_limit = b.mark_end_min > b.estimated_ppb * 6 ? b.mark_end_min - b.estimated_ppb * 6 : 0;
    const int n_stop = (int)b.scan_end - end_limit;
    Markers m = (n_start > 0 && n_start <= SWEEP_WINDOW_MAX && n_stop > 0 && n_stop <= SWEEP_WINDOW_MAX) ? search_markers_masks(b, ps, lds.px, l.ref_level, n_start, n_stop)
                                                                                                       : search_markers_px(b, ps, lds.px, l.ref_level, h);
    bool ok = (lane < 24) && m.has_start && (m.ed_stage == MARK_ED_LEN_OK);
    uint32_t k = coords_key((int16_t)m.st1e, (int16_t)m.ed_start);
    uint64_t okm = __ballot(ok);
    int best = 0;
    if (okm != 0) {
        /* minimum key, lowest lane on ties: tournament over the (at most 24) candidates */
        uint32_t best_key = 0xFFFFFFFFu; bool have = false;
        for (int i = 0; i < 24; i++) {
            uint32_t ki = (uint32_t)__shfl((int)k, i);
            if ((okm >> i) & 1ull) { if (!have || ki < best_key) { best_key = ki; best = i; have = true; } }
        }
    }
    /* final search with the chosen level == that lane's result */
    Markers r;
    r.st_stage = (uint8_t)__shfl((int)m.st_stage, best);
    r.ed_stage = (uint8_t)__shfl((int)m.ed_stage, best);
    r.st1s = (uint16_t)__shfl((int)m.st1s, best);
    r.st1e = (uint16_t)__shfl((int)m.st1e, best);
    r.st3e = (uint16_t)__shfl((int)m.st3e, best);
    r.ed_start = (uint16_t)__shfl((int)m.ed_start, best);
    r.ed_end = (uint16_t)__shfl((int)m.ed_end, best);
    r.has_start = (r.st_stage == MARK_ST_BOT_2);
    apply_markers(l, r);
}


/* ---- 256 levels as four 64-bit words: the selections below walk over runs of set bits, not over levels ---- */
struct M256 { uint64_t w[4]; };
__device__ __forceinline__ M256 m256_zero() { M256 m; m.w[0] = m.w[1] = m.w[2] = m.w[3] = 0ull; return m; }
__device__ __forceinline__ M256 m256_not(const M256 &a) { M256 m; m.w[0] = ~a.w[0]; m.w[1] = ~a.w[1]; m.w[2] = ~a.w[2]; m.w[3] = ~a.w[3]; return m; }
__device__ __forceinline__ M256 m256_and(const M256 &a, const M256 &b) { M256 m; m.w[0] = a.w[0] & b.w[0]; m.w[1] = a.w[1] & b.w[1]; m.w[2] = a.w[2] & b.w[2]; m.w[3] = a.w[3] & b.w[3]; return m; }
__device__ __forceinline__ bool m256_any(const M256 &a) { return (a.w[0] | a.w[1] | a.w[2] | a.w[3]) != 0ull; }
__device__ __forceinline__ int m256_count(const M256 &a) { return __popcll(a.w[0]) + __popcll(a.w[1]) + __popcll(a.w[2]) + __popcll(a.w[3]); }
__device__ __forceinline__ void m256_set(M256 &m, int i)
{
#pragma unroll
    for (int g = 0; g < 4; g++) m.w[g] |= (g == (i >> 6)) ? (1ull << (i & 63)) : 0ull;
}
__device__ __forceinline__ bool m256_test(const M256 &m, int i)
{
    uint64_t x = 0ull;
#pragma unroll
    for (int g = 0; g < 4; g++) x = (g == (i >> 6)) ? m.w[g] : x;
    return ((x >> (i & 63)) & 1ull) != 0ull;
}
/* bits lo..hi (inclusive; empty when hi < lo) */
__device__ inline M256 m256_range(int lo, int hi)
{
    M256 m = m256_zero();
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int a = lo - 64 * g, b = hi - 64 * g;         /* the range in this word's own bit numbers */
        if (b < 0 || a > 63 || hi < lo) continue;
        const uint64_t up = b >= 63 ? ~0ull : ((2ull << b) - 1ull), dn = a <= 0 ? ~0ull : (~0ull << a);
        m.w[g] = up & dn;
    }
    return m;
}
/* the lowest set bit at or above p, 256 when there is none */
__device__ inline int m256_bottom_ge(const M256 &m, int p)
{
    int r = 256;
#pragma unroll
    for (int g = 3; g >= 0; g--) {
        const int lo = p - 64 * g;
        if (lo > 63) continue;
        uint64_t x = m.w[g];
        if (lo > 0) x &= ~0ull << lo;
        if (x) r = 64 * g + __ffsll((unsigned long long)x) - 1;
    }
    return r;
}
/* the highest set bit at or below p (p < 0: none), -1 when there is none */
__device__ inline int m256_top_le(const M256 &m, int p)
{
    int r = -1;
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const int hi = p - 64 * g;
        if (hi < 0) continue;
        uint64_t x = m.w[g];
        if (hi < 63) x &= (2ull << hi) - 1ull;
        if (x) r = 64 * g + 63 - __clzll(x);
    }
    return r;
}

__device__ inline uint32_t wave_min_u32(uint32_t v) { for (int d = 1; d < 64; d <<= 1) { uint32_t o = (uint32_t)__shfl((int)v, lane_id() ^ d); v = o < v ? o : v; } return v; }
__device__ inline uint32_t wave_max_u32(uint32_t v) { for (int d = 1; d < 64; d <<= 1) { uint32_t o = (uint32_t)__shfl((int)v, lane_id() ^ d); v = o > v ? o : v; } return v; }

/* ======================================================================================== */
/* AGC: BLACK / WHITE levels                                                                 */
/* ======================================================================================== */

__device__ inline void hist_clear(WaveLds &lds)
{
    SDV_WAVE_SYNC();
    for (int i = lane_id(); i < 256; i += 64) lds.hist[i] = 0;
    SDV_WAVE_SYNC();
}
/* adds pixels [from, to) (ascending) to the histogram, lanes striding */
__device__ inline void hist_add_range(WaveLds &lds, int from, int to)
{
    for (int p = from + lane_id(); p < to; p += 64) atomicAdd(&lds.hist[lds.px[p]], 1u);
    SDV_WAVE_SYNC();
}

/* brightness-spread counters are uint16_t in the reference (binarizer.cpp:3125): a line is at most
 * 65535 px (uint16_t line_length), so no counter can wrap - 32-bit LDS counters are equivalent.
 *
 * The three questions the reference asks of a finished brightness spread, for the whole wave at once (lane i holds levels i, i + 64, i + 128,
 * i + 192): getMostFrequentBrightnessCount (binarizer.cpp:2450-2468) = the maximum over the lanes; getUsefullLowLevel (:2471-2513) = the lowest
 * level below max_black_lvl that more than 1/64 of that many pixels have (0 when there is none: the reference's second, unfiltered pass starts where
 * the first one ended - at max_black_lvl - and does nothing); getUsefullHighLevel (:2516-2557) = the highest such level from 255 down to
 * min_white_lvl (255 when there is none; its unfiltered pass stops at once on the level the filtered one found).  Needs the spread complete and
 * visible (hist_add_range ends with a barrier). */
struct SpreadLevels { uint16_t most_frequent; uint8_t low, high; const uint32_t *counts; /* the spread itself (LDS) */ };
/* the pixel count of one level.  (Read from LDS: a scan that takes the counts out of the lanes' registers with v_readlane instead was measured at
 * twice the time of this one - 89 000 against 49 000 cycles per findBlackWhite.) */
__device__ __forceinline__ uint32_t spread_at(const SpreadLevels &s, int lev) { return s.counts[lev]; }
__device__ inline SpreadLevels spread_levels(const sdv_bin_preset &ps, const WaveLds &lds)
{
    const int lane = lane_id();
    SpreadLevels r;
    r.counts = lds.hist;
    uint32_t h[4];
#pragma unroll
    for (int g = 0; g < 4; g++) h[g] = lds.hist[64 * g + lane];
    uint32_t m = h[0] > h[1] ? h[0] : h[1]; { const uint32_t m2 = h[2] > h[3] ? h[2] : h[3]; m = m > m2 ? m : m2; }
    r.most_frequent = (uint16_t)uniu(wave_max_u32(m));
    const uint32_t min_freq = r.most_frequent / 64;
    M256 A;
#pragma unroll
    for (int g = 0; g < 4; g++) A.w[g] = __ballot(h[g] > min_freq);
    int lo = 256;
#pragma unroll
    for (int g = 3; g >= 0; g--) if (A.w[g]) lo = 64 * g + __ffsll((unsigned long long)A.w[g]) - 1;
    r.low = lo < (int)ps.max_black_lvl ? (uint8_t)lo : (uint8_t)0;
    const int hi = m256_top_le(A, 255);
    r.high = hi >= (int)ps.min_white_lvl ? (uint8_t)hi : (uint8_t)255;
    return r;
}

/* The second half of Binarizer::findBlackWhite (binarizer.cpp:3290-3473), the same for the three formats: the black peak upwards from the lowest
 * useful level and the white peak downwards from the highest one - the running maximum, taken once it tops 1/64 of the most frequent count, until the
 * scan is 10 % / 12 % of the range past it - and the checks of the pair (contrast, limits; sweep_flag = Binarizer::do_ref_lvl_sweep, sticky).  The
 * counts come out of the lanes' registers (spread_at). */
struct BwLevels { uint8_t black, white; bool set; };
/* The walks over the brightness spread that look for a peak (binarizer.cpp:2450-2558, :3280-3400): from `start` in steps of `step` over `n` levels, the
 * level with the highest count so far is marked whenever a new highest count is also above `min_count`; the walk ends `stop_dist` levels behind the
 * mark.  found = a level was marked; level = the last mark ahead of the end of the walk.  One level per lane (n <= 64): the highest count so far as a
 * prefix maximum across the wave, the marks and the end of the walk from two ballots - the serial form read the spread one level at a time from
 * LDS, some 20 levels per walk and three walks per findBlackWhite (6 000 of its 9 500 cycles). */
struct PeakWalk { bool found; uint8_t level; };
__device__ inline PeakWalk peak_walk(const SpreadLevels &sl, int start, int step, int n, uint32_t min_count, int stop_dist)
{
    PeakWalk r; r.found = false; r.level = 0;
    if (n <= 0) return r;
    if (n > 64) {               /* (a spread wider than 192 levels: as the reference walks) */
        uint32_t top = 0;
        for (int t = 0; t < n; t++) {
            const int lev = start + step * t;
            const uint32_t cnt = spread_at(sl, lev);
            if (cnt > top) { top = cnt; if (top > min_count) { r.level = (uint8_t)lev; r.found = true; } }
            if (r.found) { const int d = (lev - (int)r.level) * step; if (d >= stop_dist) break; }
        }
        return r;
    }
    const int lane = lane_id();
    const bool act = lane < n;
    const uint32_t cnt = act ? spread_at(sl, start + step * lane) : 0u;
    uint32_t pm = cnt;                                  /* inclusive prefix maximum */
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl((int)pm, lane >= d ? lane - d : lane); if (lane >= d && o > pm) pm = o; }
    uint32_t prev = (uint32_t)__shfl((int)pm, lane > 0 ? lane - 1 : 0); if (lane == 0) prev = 0u;
    const uint64_t marks = __ballot(act && cnt > prev && cnt > min_count);
    const uint64_t upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
    const uint64_t mine = marks & upto;
    const int last = mine ? 63 - (int)__clzll((long long)mine) : -1;
    const uint64_t ends = __ballot(act && last >= 0 && (lane - last) >= stop_dist);
    const uint64_t seen = ends ? marks & ((ends & (0ull - ends)) | ((ends & (0ull - ends)) - 1ull)) : marks;        /* the marks up to the lane the walk ends in */
    if (seen) { r.found = true; r.level = (uint8_t)(start + step * (63 - (int)__clzll((long long)seen))); }
    return r;
}

__device__ inline BwLevels bw_from_spread(const sdv_bin_preset &ps, const SpreadLevels &sl, bool sweep_flag)
{
    uint8_t brt_lev, br_black, br_white, useful_low, useful_high, low_scan_limit, high_scan_limit, range_limit, bin_low, bin_high;
    uint32_t black_lvl_count, white_lvl_count, temp_calc;
    uint16_t search_lim;
    bool black_level_detected, white_level_detected;
    useful_low = low_scan_limit = br_black = sl.low;
    useful_high = high_scan_limit = br_white = sl.high;
    range_limit = (uint8_t)(high_scan_limit - low_scan_limit);
    low_scan_limit = (uint8_t)(low_scan_limit + (range_limit / 3));
    high_scan_limit = (uint8_t)(high_scan_limit - (range_limit / 3));
    temp_calc = range_limit; temp_calc = temp_calc * 10 / 100; bin_low = (uint8_t)temp_calc;
    temp_calc = range_limit; temp_calc = temp_calc * 12 / 100; bin_high = (uint8_t)temp_calc;
    search_lim = sl.most_frequent;
    search_lim = search_lim / 64;
    (void)brt_lev; (void)black_lvl_count; (void)white_lvl_count;
    {   /* the black peak, upwards from the lowest useful level (:3330-3362) */
        const PeakWalk w = peak_walk(sl, useful_low, 1, (int)low_scan_limit - (int)useful_low + 1, search_lim, bin_low);
        black_level_detected = w.found; if (w.found) br_black = w.level;
    }
    white_level_detected = false;
    if (black_level_detected) {     /* the white peak, downwards - not closer to the black level than the least contrast (:3364-3400) */
        int lowest = high_scan_limit; if (lowest < (int)br_black + (int)ps.min_contrast) lowest = (int)br_black + (int)ps.min_contrast;
        const PeakWalk w = peak_walk(sl, useful_high, -1, (int)useful_high - lowest + 1, search_lim, bin_high);
        white_level_detected = w.found; if (w.found) br_white = w.level;
    }
    if (black_level_detected && white_level_detected) {
        bool invalidate = false;
        if (br_white < br_black) invalidate = true;
        else if (((int)br_white - (int)br_black) < (int)ps.min_contrast) invalidate = true;
        else if (sweep_flag && (((int)br_white - (int)br_black) < (int)ps.min_valid_crcs)) invalidate = true;
        else if (br_black > ps.max_black_lvl) invalidate = true;
        else if (br_white < ps.min_white_lvl) invalidate = true;
        if (invalidate) { black_level_detected = white_level_detected = false; br_black = useful_low; br_white = useful_high; }
    }
    BwLevels r; r.black = br_black; r.white = br_white; r.set = black_level_detected && white_level_detected;
    return r;
}

/* Binarizer::findSTC007BW (binarizer.cpp:2684-3070); leaves the brightness spread to analyse in lds.hist */
__device__ inline bool find_stc007_bw(Bin &b, const sdv_bin_preset &ps, WaveLds &lds, Line &line)    /* returns: a white level was seen (the line's STOP-marker fields were written) */
{
    uint8_t brt_lev, stage, br_mark_white, useful_low, useful_high, high_scan_limit, low_scan_limit, range_limit, bin_level, bin_low, bin_high, pv;
    uint16_t pixel, pixel_limit, ed_start, ed_end, search_lim;
    uint32_t white_lvl_count, temp_calc;
    bool white_level_detected;

    hist_clear(lds);
    search_lim = (uint16_t)(b.scan_start + b.estimated_ppb * 10);
    hist_add_range(lds, b.scan_start, search_lim);
    search_lim = (uint16_t)(b.scan_end - b.estimated_ppb * 20);
    hist_add_range(lds, search_lim, (int)b.scan_end + 1);

    const SpreadLevels sl = spread_levels(ps, lds);
    useful_low = low_scan_limit = sl.low; useful_high = high_scan_limit = br_mark_white = sl.high;
    range_limit = (uint8_t)(high_scan_limit - low_scan_limit);
    high_scan_limit = (uint8_t)(high_scan_limit - (range_limit / 4));
    bin_high = range_limit / 8;
    (void)brt_lev; (void)white_lvl_count;
    {   /* the white peak of the two ends of the line, downwards over the top quarter of the range (:2760-2790) */
        const PeakWalk w = peak_walk(sl, useful_high, -1, (int)useful_high - (int)high_scan_limit + 1, 0u, bin_high);
        white_level_detected = w.found; if (w.found) br_mark_white = w.level;
    }
    pixel_limit = (uint16_t)(b.scan_end - b.scan_start);
    temp_calc = pixel_limit / 8;
    pixel_limit = (uint16_t)(b.scan_start + (uint16_t)temp_calc);
    search_lim = (uint16_t)(b.scan_end - (uint16_t)temp_calc);
    hist_clear(lds);
    hist_add_range(lds, pixel_limit, search_lim);

    stage = MARK_ED_START; ed_start = ed_end = 0;
    if (white_level_detected) {
        bin_level = pick_center_ref_level(ps, useful_low, br_mark_white);
        bin_high = bin_low = bin_level;
        if (b.mark_end_min > (b.estimated_ppb * 6)) pixel_limit = (uint16_t)(b.mark_end_min - b.estimated_ppb * 6);
        else pixel_limit = 0;
        const int n_stop = (int)b.scan_end - (int)pixel_limit;           /* pixels scan_end, scan_end - 1, ... the search may look at */
        if (n_stop <= 256) {
            /* the comparison of the whole window at once (bit i = pixel scan_end - i is at or above the level), the state machine from edge to edge */
            M256 R;
#pragma unroll
            for (int g = 0; g < 4; g++) { const int i = 64 * g + lane_id(); R.w[g] = __ballot(i < n_stop && lds.px[i < n_stop ? (int)b.scan_end - i : 0] >= bin_level); }
            const M256 nR = m256_not(R);
            const int i_max = (int)b.scan_end - (int)b.mark_end_min;
            for (int i = 0;;) {
                const int j1 = m256_bottom_ge(R, i);
                if (j1 >= n_stop || j1 > i_max) { stage = MARK_ED_START; break; }
                ed_end = (uint16_t)(b.scan_end - j1 + 1); stage = MARK_ED_TOP;
                const int j2 = m256_bottom_ge(nR, j1 + 1);
                if (j2 >= n_stop) break;
                ed_start = (uint16_t)(b.scan_end - j2 + 1);
                if ((j2 - j1) >= (b.estimated_ppb * 2)) { stage = MARK_ED_LEN_OK; break; }
                stage = MARK_ED_START; i = j2 + 1;
            }
        } else {
        pixel = b.scan_end;
        while (pixel > pixel_limit) {
            pv = lds.px[pixel];
            if (stage == MARK_ED_START) {
                if (pixel < b.mark_end_min) break;
                if (pv >= bin_low) { ed_end = (uint16_t)(pixel + 1); stage = MARK_ED_TOP; }
            } else if (stage == MARK_ED_TOP) {
                if (pv < bin_high) {
                    ed_start = (uint16_t)(pixel + 1);
                    stage = MARK_ED_BOT;
                    if ((ed_end - ed_start) >= (b.estimated_ppb * 2)) { stage = MARK_ED_LEN_OK; break; }
                    else stage = MARK_ED_START;
                }
            }
            pixel--;
        }
        }
        line.mark_ed = stage;
        line.coords.stop = (int16_t)ed_start;
        line.m_sp_ed = ed_end;
        if (has_stop(line)) {
            search_lim = (uint16_t)(b.estimated_ppb * 64);
            if (search_lim > ed_start) search_lim = b.mark_start_max;
            else search_lim = (uint16_t)(ed_start - search_lim);
            hist_clear(lds);
            /* pixels ed_start-1 down to search_lim+1 */
            int cnt = ((int)ed_start - 1) - (int)search_lim;
            if (cnt < 0) cnt = 0;
            hist_add_range(lds, (int)search_lim + 1, (int)ed_start);
            if (cnt < 32) {
                pixel_limit = (uint16_t)(b.scan_end - b.scan_start);
                pixel_limit = pixel_limit / 8;
                search_lim = (uint16_t)(b.scan_end - pixel_limit);
                hist_add_range(lds, pixel_limit, search_lim);
            }
        }
    }
    return white_level_detected;
}

/* Binarizer::findBlackWhite (binarizer.cpp:3116-3473), STC-007 branch.
 * memo: this line's slot of FrameArgs::bw_memo, or NULL.  What the search finds depends on the line's pixels and on one bit of the chain (the sticky
 * do_ref_lvl_sweep) - not on the tuning a frame was started from - and a frame of a worn tape is decoded several times (passes that wait for sweeps,
 * rounds of the speculation): the first decode leaves the outcome in the slot, the later ones take it from there (47 000 cycles -> one 8-byte load).
 * Slot: bit 0 present, 1 the do_ref_lvl_sweep it was made with, 2 levels set, 3 STOP-marker fields written; black << 8, white << 16, stage << 24, the two
 * STOP-marker coordinates in the upper half. */
__device__ inline bool find_black_white(Bin &b, const sdv_bin_preset &ps, WaveLds &lds, Line &line, unsigned long long *memo = nullptr)
{
    if (memo) {
        const uint2 m = *(const uint2 *)memo;
        const uint32_t lo = uniu(m.x), hi = uniu(m.y);
        if ((lo & 1u) && ((lo >> 1) & 1u) == (b.do_ref_lvl_sweep ? 1u : 0u)) {
            b.was_bw_scanned = true;
            line.black = (uint8_t)(lo >> 8); line.white = (uint8_t)(lo >> 16); line.bw_set = (lo & 4u) != 0;
            if (lo & 8u) { line.mark_ed = (uint8_t)(lo >> 24); line.coords.stop = (int16_t)(hi & 0xFFFFu); line.m_sp_ed = (uint16_t)(hi >> 16); }
            return line.bw_set;
        }
    }
    const bool wrote_stop = find_stc007_bw(b, ps, lds, line);
    const BwLevels bw = bw_from_spread(ps, spread_levels(ps, lds), b.do_ref_lvl_sweep);
    b.was_bw_scanned = true;
    line.black = bw.black; line.white = bw.white;
    line.bw_set = bw.set;
    if (memo && lane_id() == 0) {
        const uint32_t lo = 1u | (b.do_ref_lvl_sweep ? 2u : 0u) | (bw.set ? 4u : 0u) | (wrote_stop ? 8u : 0u) | ((uint32_t)bw.black << 8) | ((uint32_t)bw.white << 16) | ((uint32_t)line.mark_ed << 24);
        const uint32_t hi = (uint32_t)(uint16_t)line.coords.stop | ((uint32_t)line.m_sp_ed << 16);
        uint2 m; m.x = lo; m.y = hi;
        *(uint2 *)memo = m;
    }
    return line.bw_set;
}

/* ======================================================================================== */
/* CRC statistics (binarizer.cpp:1771-2383) on LDS arrays; serial, wave-uniform               */
/* ======================================================================================== */

__device__ inline void crc_stats_reset(WaveLds &lds, int count)
{
    for (int i = 0; i < count; i++) { lds.crc_stats[i].result = 0; lds.crc_stats[i].crc = 0; lds.crc_stats[i].hyst = lds.crc_stats[i].shift = 0x0f; lds.crc_stats[i].idx = 0; }
}
__device__ inline void crc_stats_update(WaveLds &lds, uint16_t crc, uint8_t hyst, uint8_t shift, uint8_t &valid_cnt)   /* :1789-1826 */
{
    bool found = false;
    if (valid_cnt >= MAX_COLL_CRCS) valid_cnt = MAX_COLL_CRCS - 1;
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (lds.crc_stats[i].crc == crc) { lds.crc_stats[i].result++; found = true; break; }
    if (!found) {
        valid_cnt++;
        if (valid_cnt < MAX_COLL_CRCS) {
            lds.crc_stats[valid_cnt].crc = crc; lds.crc_stats[valid_cnt].hyst = hyst; lds.crc_stats[valid_cnt].shift = shift;
            lds.crc_stats[valid_cnt].result++;
        }
    }
}
__device__ inline void crc_stats_most_frequent(WaveLds &lds, uint8_t &valid_cnt)   /* :1829-1928, skip_equal = true */
{
    CrcStat *a = lds.crc_stats;
    a[0].result = 0; a[0].idx = 0; a[0].hyst = 0; a[0].shift = 0;
    if (valid_cnt >= MAX_COLL_CRCS) valid_cnt = MAX_COLL_CRCS - 1;
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (a[i].result > a[0].result) { a[0].result = a[i].result; a[0].crc = a[i].crc; a[0].hyst = a[i].hyst; a[0].shift = a[i].shift; a[0].idx = i; }
    for (uint8_t i = 1; i <= valid_cnt; i++)
        if (a[0].idx != i)
            if ((int)a[0].result <= (2 * (int)a[i].result)) { a[0].result = 0; a[0].hyst = 0; a[0].shift = 0; break; }
    if (a[0].result == 0) valid_cnt = 0;
}
__device__ inline void sweep_invalidate_non_frequent(WaveLds &lds, uint8_t low_level, uint8_t high_level, uint8_t valid_cnt, uint16_t target_crc)   /* :1931-1982 */
{
    uint8_t index = high_level;
    while (index >= low_level) {
        if (lds.sweep[index].result == REF_CRC_OK)
            if ((valid_cnt == 0) || (lds.sweep[index].crc != target_crc)) lds.sweep[index].result = REF_CRC_COLL;
        if (index == low_level) break;
        index--;
    }
}
/* Second half of pickLevelByCRCStats (binarizer.cpp:2060-2140): E = the levels that carry the target result at the lowest (depth, stage),
 * none below low_lvl; high_ref = the highest of them.  The reference walks down from there: the run that starts at high_ref is the span to
 * beat, every further run that a non-member CLOSES inside [low_lvl, ..] replaces it when it is at least as long (so of equally long runs
 * the lowest wins, and a run still open at low_lvl is never looked at).  Here: from run to run with count-leading-zeros. */
__device__ inline void pick_longest_run(const M256 &E, int low_lvl, int &low_ref, int &high_ref)
{
    const M256 nE = m256_not(E);
    int z = m256_top_le(nE, high_ref);                      /* the non-member that closes the first run (below low_lvl: it stays open) */
    low_ref = z + 1 > low_lvl ? z + 1 : low_lvl;
    while (z >= low_lvl) {
        const int h = m256_top_le(E, z - 1);
        if (h < low_lvl) break;
        z = m256_top_le(nE, h);
        if (z < low_lvl) break;                             /* open at the low end */
        if ((h - (z + 1)) >= (high_ref - low_ref)) { low_ref = z + 1; high_ref = h; }
    }
}
/* First half of pickLevelByCRCStatsOpt (binarizer.cpp:2178-2262): the widest run of usable levels inside [low_lvl, high_lvl], of equally
 * wide ones the lowest; a run of two or more levels that reaches low_lvl is taken whatever its width, a single level there is not. */
__device__ inline bool pick_widest_region(const M256 &good, int low_lvl, int high_lvl, int &reg_lo, int &reg_hi)
{
    const M256 ngood = m256_not(good);
    bool lock = false;
    reg_lo = reg_hi = 0;
    for (int pos = high_lvl; pos >= low_lvl;) {
        const int h = m256_top_le(good, pos);
        if (h < low_lvl) break;
        const int z = m256_top_le(ngood, h);
        if (z < low_lvl) { if (h > low_lvl) { reg_lo = low_lvl; reg_hi = h; lock = true; } break; }
        if ((h - z) >= (reg_hi - reg_lo + 1)) { reg_lo = z + 1; reg_hi = h; lock = true; }
        pos = z;
    }
    return lock;
}
/* Second half of pickLevelByCRCStatsOpt (binarizer.cpp:2263-2383) over the usable levels from high_lvl down, `at(i)` = (depth << 8 | stage) of level i:
 * a lower depth, or the same depth with a lower stage, restarts the span there; levels equal to the best extend it downwards, five of them end
 * the walk, and so do five levels that are worse since the last restart. */
template <class At>
__device__ inline bool pick_opt_walk(const M256 &good, int low_lvl, int high_lvl, uint8_t max_ref_lvl, At at, uint8_t *ref_result)
{
    bool any = false;
    uint32_t best = 0xFFFFFFFFu;
    int low_ref = max_ref_lvl, high_ref = max_ref_lvl, hold = 0, same = MIN_VALID_CRCS;
    for (int pos = high_lvl; pos >= low_lvl;) {
        const int i = m256_top_le(good, pos);
        if (i < low_lvl) break;
        const uint32_t k = at(i);
        any = true;
        if (k < best) {
            if ((k >> 8) == (best >> 8)) same = MIN_VALID_CRCS;    /* the same depth, a lower stage */
            best = k; low_ref = high_ref = i; hold = MIN_VALID_CRCS;
        } else if (k == best) { low_ref = i; if (--same == 0) break; }
        else if (--hold == 0) break;
        pos = i - 1;
    }
    if (!any) return false;
    *ref_result = (uint8_t)(low_ref + (uint8_t)(high_ref - low_ref) / 2);
    return true;
}

/* pickLevelByCRCStats (binarizer.cpp:1985-2140) on a table in memory, for one lane on its own: the levels with the target result within the limits
 * as a mask, the lowest (depth, stage) among them, then pick_longest_run over those that have it.
 * crcs: the table the indices count from (WaveLds::sweep, or a row inside it) */
__device__ inline uint8_t pick_level_by_crc_stats_at(const SweepEnt *crcs, uint8_t *ref_result, uint8_t low_lvl, uint8_t high_lvl,
                                                     uint8_t target_result, uint8_t max_hyst, uint8_t max_shift)
{
    uint32_t best = 0xFFFFFFFFu;
    for (int i = low_lvl; i <= (int)high_lvl; i++) {
        const SweepEnt e = crcs[i];
        const uint32_t k = ((uint32_t)e.hyst << 8) | e.shift;
        if (e.result == target_result && e.hyst <= max_hyst && e.shift <= max_shift && k < best) best = k;
    }
    if (best == 0xFFFFFFFFu) return SPAN_NOT_FOUND;
    M256 E = m256_zero();
    for (int i = low_lvl; i <= (int)high_lvl; i++) {
        const SweepEnt e = crcs[i];
        if (e.result == target_result && (((uint32_t)e.hyst << 8) | e.shift) == best) m256_set(E, i);
    }
    int high_ref = m256_top_le(E, high_lvl), low_ref;
    pick_longest_run(E, low_lvl, low_ref, high_ref);
    *ref_result = (uint8_t)(low_ref + (uint8_t)(high_ref - low_ref) / 2);
    return SPAN_OK;
}
__device__ inline uint8_t pick_level_by_crc_stats(const WaveLds &lds, uint8_t *ref_result, uint8_t low_lvl, uint8_t high_lvl,
                                                  uint8_t target_result, uint8_t max_hyst, uint8_t max_shift)
{
    return pick_level_by_crc_stats_at(lds.sweep, ref_result, low_lvl, high_lvl, target_result, max_hyst, max_shift);
}
/* pickLevelByCRCStatsOpt (binarizer.cpp:2143-2383) on WaveLds::sweep, for one lane on its own */
__device__ inline uint8_t pick_level_by_crc_stats_opt(const sdv_bin_preset &ps, const WaveLds &lds, uint8_t *ref_result, uint8_t low_lvl, uint8_t high_lvl,
                                                      uint8_t target_result, uint8_t max_hyst, uint8_t max_shift)
{
    const SweepEnt *crcs = lds.sweep;
    M256 good = m256_zero();
    for (int i = low_lvl; i <= (int)high_lvl; i++) {
        const SweepEnt e = crcs[i];
        if (e.result == target_result && e.hyst <= max_hyst && e.shift <= max_shift) m256_set(good, i);
    }
    int lo = low_lvl, hi = high_lvl, rl, rh;
    if (pick_widest_region(good, lo, hi, rl, rh)) { lo = rl; hi = rh; }
    return pick_opt_walk(good, lo, hi, ps.max_ref_lvl, [&](int i) -> uint32_t { return ((uint32_t)crcs[i].hyst << 8) | crcs[i].shift; }, ref_result) ? SPAN_OK : SPAN_NOT_FOUND;
}

/* ======================================================================================== */
/* Reference level sweep (binarizer.cpp:3551-4120): see stc007_sweep_device.h                */
/* ======================================================================================== */

__device__ inline void calc_forced_coords(const Bin &b, const sdv_bin_preset &ps, Coords &fc)   /* :631-641 */
{
    coords_clear(fc);
    if (ps.en_force_coords) {
        fc.start = (int16_t)((int16_t)b.scan_start + ps.horiz_start);
        fc.stop = (int16_t)((int16_t)b.scan_end - ps.horiz_stop);
        if (b.vl_doubled) { fc.start = (int16_t)(fc.start + ps.horiz_start); fc.stop = (int16_t)(fc.stop - ps.horiz_stop); }
    }
}
__device__ inline void bin_set_mode(Bin &b, uint8_t m);
__device__ inline bool pod_coords_valid(int16_t s, int16_t e) { return s != NO_COORD_LEFT && e != NO_COORD_RIGHT && s < e; }
} // namespace sdv
#include "stc007_sweep_device.h"
namespace sdv {

/* Binarizer::calcRefLevelBySweep (binarizer.cpp:3821-4120) in the frame kernel: the sweep itself, its vote and its pick are taken from the
 * outcome the sweep kernels left for this line (stc007_sweep_device.h); what is done here is what the reference does with the level picked.
 * Without an outcome the line leaves a request and goes on as if the sweep had found nothing: the frame is decoded again once the request is
 * settled, so what is made of the line here never reaches the caller. */
template <int kWho>        /* (see process_line) */
__device__ inline void calc_ref_level_by_sweep(Bin &b, const sdv_bin_preset &ps, WaveLds &lds, Line &l, SweepHook &hook)
{
    Coords forced_coords;
    const uint8_t fast_ref = pick_center_ref_level(ps, l.black, l.white);
    b.hyst_lim = b.in_max_hyst; b.shift_lim = b.in_max_shift;
    calc_forced_coords(b, ps, forced_coords);
    SweepOutcome o;
    if (!sweep_lookup(hook, l.black, l.white, b.in_coord, o)) {
        const int slot = sweep_request(hook, l.black, l.white, b.in_coord);
        bool have = false;
        if (kWho == 3 && hook.fat && slot >= 0) {       /* sdv_k_stc007_frames_fat: settled by the waves beside this one, now */
            fat_sweep(hook, slot);
            have = sweep_lookup(hook, l.black, l.white, b.in_coord, o);
            if (have) hook.pending = false;
        }
        if (!have) {
            if (b.in_ref < ps.min_ref_lvl) hook.stop = true;
            o.span1 = o.span2 = SPAN_NOT_FOUND; o.ref_level = 0; o.t_hyst = o.t_shift = 0; o.t_start = o.t_stop = 0;
        }
    }
    if (o.span1 == SPAN_OK) {
        l.ref_level = o.ref_level;
        l.ref_sweeped = true;
        coords_set(l.coords, o.t_start, o.t_stop);
        l.coords_set = true;
        if (!coords_valid(forced_coords)) find_coordinates_wave(b, ps, lds, l);
        b.hyst_lim = o.t_hyst;
        if (b.hyst_lim > HYST_DEPTH_MAX) b.hyst_lim = HYST_DEPTH_MAX;
        b.shift_lim = o.t_shift;
    } else {
        if (o.span1 == SPAN_TOO_NARROW) l.forced_bad = true;
        if (o.span2 == SPAN_OK) {
            l.ref_level = o.ref_level;
            coords_set(l.coords, o.t_start, o.t_stop);
            l.coords_set = true;
            if (!coords_valid(forced_coords)) find_coordinates_wave(b, ps, lds, l);
        } else if (b.in_ref >= ps.min_ref_lvl) {
            l.ref_level = b.in_ref;
            if (coords_valid(b.in_coord)) l.coords = b.in_coord;
        } else {
            l.ref_level = fast_ref;
            if (!coords_valid(b.in_coord)) coords_set(l.coords, (int16_t)(b.scan_start + b.estimated_ppb), (int16_t)(b.scan_end - (4 * b.estimated_ppb)));
            else l.coords = b.in_coord;
        }
        b.hyst_lim = 0; b.shift_lim = SHIFT_STAGES_MIN;
    }
}

/* ======================================================================================== */
/* Binarizer setters (binarizer.cpp:240-438)                                                 */
/* ======================================================================================== */
__device__ inline void bin_set_bw_levels(Bin &b, const sdv_bin_preset &ps, uint8_t blk, uint8_t wht)
{
    if ((blk < wht) && (blk < ps.max_black_lvl) && (wht > ps.min_white_lvl) && (wht != 0)) { b.in_black = blk; b.in_white = wht; }
    else b.in_black = b.in_white = 0;
}
__device__ inline void bin_set_data_coordinates(Bin &b, const Coords &c) { if (coords_valid(c)) b.in_coord = c; else coords_clear(b.in_coord); }
__device__ inline void bin_set_data_coordinates2(Bin &b, int16_t s, int16_t e)
{
    Coords t; coords_clear(t);
    if ((s < e) && (e != 0) && (s != NO_COORD_LEFT) && (e != NO_COORD_RIGHT)) coords_set(t, s, e);
    bin_set_data_coordinates(b, t);
}
__device__ inline void bin_set_good_parameters_reset(Bin &b, const sdv_bin_preset &ps)
{
    b.in_ref = 0; bin_set_data_coordinates2(b, 0, 0); bin_set_bw_levels(b, ps, 0, 0);
}
__device__ inline void bin_set_good_parameters(Bin &b, const sdv_bin_preset &ps, const Line &l)
{
    if (crc_valid_ignore_forced(l)) { b.in_ref = l.ref_level; bin_set_data_coordinates(b, l.coords); bin_set_bw_levels(b, ps, l.black, l.white); }
}
__device__ __forceinline__ bool is_ref_level_preset(const Bin &b, const sdv_bin_preset &ps) { return b.in_ref >= ps.min_ref_lvl; }
__device__ inline bool are_bw_levels_preset(const Bin &b, const sdv_bin_preset &ps)
{
    if ((b.in_white > ps.min_white_lvl) && (b.in_black < ps.max_black_lvl)) {
        if (is_ref_level_preset(b, ps)) if ((b.in_ref <= b.in_black) || (b.in_ref >= b.in_white)) return false;
        return true;
    }
    return false;
}
__device__ inline void bin_set_mode(Bin &b, uint8_t m)   /* :120-152 */
{
    if (m == SDV_MODE_DRAFT) { b.mode = m; b.in_max_hyst = HYST_DEPTH_SAFE; b.in_max_shift = SHIFT_STAGES_MIN; }
    else if (m == SDV_MODE_FAST) { b.mode = m; b.in_max_hyst = 7; b.in_max_shift = SHIFT_STAGES_SAFE; }
    else if (m == SDV_MODE_INSANE) { b.mode = m; b.in_max_hyst = HYST_DEPTH_MAX; b.in_max_shift = SHIFT_STAGES_MAX; }
    else { b.mode = SDV_MODE_NORMAL; b.in_max_hyst = HYST_DEPTH_SAFE; b.in_max_shift = SHIFT_STAGES_SAFE; }
}

/* ======================================================================================== */
/* Binarizer::processLine (binarizer.cpp:443-1724) for a regular (non-service, non-empty) line */
/* that is already staged in lds.px.  Returns LB_RET_*.                                        */
/* ======================================================================================== */
/* kWho: which kernel the copy belongs to (0 the general frame kernel, 1 its build without snapshots, 2 the per-line kernel, 3 the frame kernel of small rounds).  Every kernel has its own copy of the
 * general path (slow_line<kWho> and what it calls once), for the compiler's sake: a function with two callers is no longer inlined into them - the binarizer,
 * the line and the sweep hook this one takes by reference then live in scratch memory - and a function shared by kernels of different launch bounds is compiled
 * for the registers of one of them and paid for by all (seen: a fifth of the general kernel's speed on the damaged tapes; 248 registers and one wave per SIMD
 * for every kernel that shared the path with a kernel without a bound). */
template <int kWho>
__device__ inline int process_line(Bin &b, const sdv_bin_preset &ps, WaveLds &lds, Line &out, uint32_t frame_no, uint16_t line_no, int width, bool doubled, SweepHook &hook)
{
    Coords forced_coords;
    stc_clear(out);
    out.frame_number = frame_no; out.line_number = line_no;
    bin_line_geometry(b, ps, width, doubled);
    out.coords.doubled = doubled;
    set_source_pixels(out, b.scan_start, b.scan_end);
    if (b.line_length < BITS_IN_LINE) return SDV_ERR_SHORT_LINE;
    coords_set(out.coords, (int16_t)b.scan_start, (int16_t)b.scan_end);
    calc_forced_coords(b, ps, forced_coords);
    if (ps.en_force_coords && coords_valid(forced_coords)) { out.coords = forced_coords; out.coords_set = true; }

    int state = STG_REF_FIND;
    b.was_bw_scanned = false;
    if (are_bw_levels_preset(b, ps)) { out.black = b.in_black; out.white = b.in_white; out.bw_set = true; }
    if (is_ref_level_preset(b, ps)) state = coords_valid(b.in_coord) ? STG_INPUT_ALL : STG_INPUT_LEVEL;
    b.hyst_lim = b.in_max_hyst; b.shift_lim = b.in_max_shift;

    int stage_count = 0;
    for (;;) {
        stage_count++;
        if (state == STG_INPUT_ALL) {                       /* :774-931 */
            if (!out.bw_set) { K1_T(t0_); find_black_white(b, ps, lds, out, hook.bw_slot); K1_T(t1_); K1_ADD(9, t0_, t1_); }
            if (!coords_valid(forced_coords)) out.coords = b.in_coord;
            out.ref_level = b.in_ref;
            out.ref_sweeped = false;
            if (!out.bw_set) state = STG_NO_GOOD;
            else if ((b.in_ref >= out.white) || (b.in_ref <= out.black)) state = STG_REF_FIND;
            else {
                bool force_level_find = false;
                if (!coords_valid(forced_coords) && !ps.en_good_no_marker) {      /* do_coord_search is always on for STC-007 (videotodigital.cpp:953) */
                    find_coordinates_wave(b, ps, lds, out);
                    out.coords = b.in_coord;
                    if (!has_markers(out)) force_level_find = true;
                }
                { K1_T(t0_); read_pcm_data(b, out, lds, hook.ladder_failed && !force_level_find); K1_T(t1_); K1_ADD(11, t0_, t1_); }
                if (crc_valid(out)) {
                    if (!force_level_find) { out.by_ext_tune = true; state = STG_DATA_OK; }
                    else state = STG_REF_FIND;
                } else state = !coords_valid(forced_coords) ? STG_INPUT_LEVEL : STG_REF_FIND;
            }
        } else if (state == STG_INPUT_LEVEL) {              /* :932-1072 */
            if (!b.was_bw_scanned) { K1_T(t0_); find_black_white(b, ps, lds, out, hook.bw_slot); K1_T(t1_); K1_ADD(9, t0_, t1_); }
            if (!coords_valid(forced_coords)) coords_set(out.coords, (int16_t)b.scan_start, (int16_t)b.scan_end);
            out.ref_level = b.in_ref;
            out.ref_sweeped = false;
            if (!out.bw_set) state = STG_NO_GOOD;
            else {
                state = STG_REF_FIND;
                if ((b.in_ref < out.white) && (b.in_ref > out.black)) {
                    { K1_T(t0_); find_coordinates_wave(b, ps, lds, out); K1_T(t1_); K1_ADD(10, t0_, t1_); }
                    if (has_markers(out)) {
                        if (!coords_valid(b.in_coord) || coords_ne(out.coords, b.in_coord)) {
                            read_pcm_data(b, out, lds);
                            if (crc_valid(out)) { out.by_ext_tune = true; state = STG_DATA_OK; }
                        }
                    }
                }
            }
        } else if (state == STG_REF_FIND) {                 /* :1073-1390 */
            if (!b.was_bw_scanned) find_black_white(b, ps, lds, out, hook.bw_slot);
            if (!out.bw_set) state = STG_NO_GOOD;
            else {
                b.do_ref_lvl_sweep = (b.mode == SDV_MODE_NORMAL) || (b.mode == SDV_MODE_INSANE);
                if (b.do_ref_lvl_sweep) state = STG_REF_SWEEP_RUN;
                else {
                    b.hyst_lim = HYST_DEPTH_SAFE; b.shift_lim = SHIFT_STAGES_MIN;
                    state = STG_READ_PCM;
                    out.ref_level = pick_center_ref_level(ps, out.black, out.white);
                    if (coords_valid(forced_coords)) { out.coords = forced_coords; out.coords_set = true; }
                    else {
                        find_coordinates_wave(b, ps, lds, out);
                        if (!has_markers(out)) { b.hyst_lim = HYST_DEPTH_SAFE; b.shift_lim = SHIFT_STAGES_MIN; }
                        else { b.hyst_lim = b.in_max_hyst; b.shift_lim = b.in_max_shift; }
                    }
                }
            }
        } else if (state == STG_REF_SWEEP_RUN) {            /* :1391-1400 */
            { K1_T(t0_); calc_ref_level_by_sweep<kWho>(b, ps, lds, out, hook); K1_T(t1_); K1_ADD(12, t0_, t1_); }
            state = STG_READ_PCM;
        } else if (state == STG_READ_PCM) {                 /* :1401-1533 */
            if (coords_valid(forced_coords)) { b.hyst_lim = HYST_DEPTH_SAFE; b.shift_lim = SHIFT_STAGES_MIN; }
            if (out.coords_set) read_pcm_data(b, out, lds);
            if (crc_valid(out)) state = STG_DATA_OK;
            if (state != STG_DATA_OK) {
                if (coords_valid(b.in_coord) && !coords_valid(forced_coords) && !b.do_ref_lvl_sweep && !out.forced_bad && !out.coords_set) {
                    if (coords_ne(out.coords, b.in_coord)) {
                        out.coords = b.in_coord;
                        out.m_st_bg = 0; out.m_st_ed = 0; out.m_sp_ed = 0;
                        read_pcm_data(b, out, lds);
                        if (crc_valid(out)) state = STG_DATA_OK;
                    }
                }
                if (state != STG_DATA_OK) state = STG_NO_GOOD;
            }
        } else if (state == STG_DATA_OK) {                  /* :1534-1621 */
            if (out.forced_bad) state = STG_NO_GOOD;
            else {
                if (!ps.en_good_no_marker && !has_markers(out)) { out.forced_bad = true; state = STG_NO_GOOD; continue; }
                apply_crc_state_per_word(out);
                if (has_control_block(out)) set_serv_ctrl_blk(out);
                out.coords.doubled = doubled;
                break;
            }
        } else if (state == STG_NO_GOOD) {                  /* :1622-1669 */
            if (crc_valid(out)) set_invalid_crc(out);
            apply_crc_state_per_word(out);
            if (coords_valid(forced_coords) && out.bw_set) { out.mark_st = MARK_ST_BOT_2; out.mark_ed = MARK_ED_LEN_OK; }
            out.coords.doubled = doubled;
            break;
        } else break;
        if (stage_count > STG_MAX) break;
    }
    return SDV_OK;
}

/* ======================================================================================== */
/* VideoToDigital::doBinarize per-frame driver (videotodigital.cpp:698-1815), STC-007          */
/* ======================================================================================== */
struct V2D {
    Bin bin;
    uint8_t field_state;
    bool reset_stats;
    uint16_t good_coords_in_field, pcm_lines_in_field, line_in_field_cnt;
    Coords frame_avg;
    int n_last, n_long, nfv, nfi;
    int long_pushes;                        /* pairs pushed into the full 16-frame history by this frame (the scheduler's "the history moves on") */
    uint16_t last_words[8];                 /* last_stc007_line.words (only the words are ever compared) */
    /* FrameBinDescriptor signal_quality */
    uint16_t q_line_length, q_odd, q_even, q_pcm_odd, q_pcm_even, q_bad_odd, q_bad_even, q_dup_odd, q_dup_even;
};

/* VideoToDigital::medianCoordinates (videotodigital.cpp:348-371): element of rank n/2 under
 * CoordinatePair::operator<.  keys[] readable by every lane (LDS or own global stores).
 * The keys are 32-bit numbers in that order, so the element of a given rank is found bit by bit from the top (a radix select): per bit the lanes
 * count, among the keys that still match the bits chosen so far, those with a 0 there - 32 passes over n / 64 keys per lane, whatever n is.
 * (Through round 3 every lane ranked its key against all n: n * n / 64 reads - a frame whose lines had found two different windows, 486 keys,
 * took 14 million cycles for this one number and held its whole launch up; the sub-line lists of PCM-16x0 are three times as long.) */
struct MedianKey { bool ok; uint32_t key; };
__device__ inline MedianKey median_keys_of(const uint32_t *keys, int n)
{
    MedianKey out; out.ok = false; out.key = 0;
    if (n <= 0) return out;
    /* (the lane number made opaque here: the function is inlined at five places of the frame loop, and what its loops derive from the lane number
     * - unrolled strides, trip counts - would be worked out once ahead of the loop and kept in registers the loop does not have) */
    int lane = lane_id();
    SDV_OPAQUE(lane);
    /* common case: every entry identical (steady tuning) -> one ballot */
    const uint32_t k0 = keys[0];
    bool differs = false;
#pragma unroll 1
    for (int j = lane; j < n; j += 64) differs = differs || (keys[j] != k0);
    if (__ballot(differs) == 0ull) { out.ok = true; out.key = uniu(k0); return out; }
    uint32_t target = (uint32_t)(n / 2);
    if (n <= 64) {              /* the short histories (9 lines, 16 frames): a lane per key, its rank by one pass over the others */
        const uint32_t ki = keys[lane < n ? lane : 0];
        uint32_t less = 0, leq = 0;
        for (int j = 0; j < n; j++) { const uint32_t kj = keys[j]; less += (kj < ki); leq += (kj <= ki); }
        const uint64_t m = __ballot(lane < n && less <= target && target < leq);
        out.key = uniu((uint32_t)__shfl((int)ki, m ? __ffsll((unsigned long long)m) - 1 : 0));
        out.ok = m != 0ull;
        return out;
    }
    uint32_t prefix = 0, mask = 0;
    for (int bit = 31; bit >= 0; bit--) {
        uint32_t zeros = 0;
#pragma unroll 2
        for (int j = lane; j < n; j += 64) { const uint32_t k = keys[j]; zeros += ((k & mask) == prefix && ((k >> bit) & 1u) == 0u) ? 1u : 0u; }
        uint32_t total = 0;     /* the lanes' counts (< 64 each for n < 4096) summed bit plane by bit plane: six ballots instead of six shuffles */
#pragma unroll
        for (int q = 0; q < 6; q++) total += (uint32_t)__popcll(__ballot((zeros >> q) & 1u)) << q;
        if (target >= total) { target -= total; prefix |= 1u << bit; }
        mask |= 1u << bit;
    }
    out.ok = true; out.key = prefix;
    return out;
}
__device__ __forceinline__ bool median_keys(const uint32_t *keys, int n, uint32_t *out_key) { const MedianKey m = median_keys_of(keys, n); *out_key = m.key; return m.ok; }

__device__ inline Coords key_to_coords(uint32_t k, bool doubled) { Coords c; c.start = key_start(k); c.stop = key_stop(k); c.doubled = doubled; return c; }

/* getWordsDiffBitCount (stc007line.cpp:329-357): the XOR is truncated to uint8_t in the reference */
__device__ inline uint8_t words_diff_bit_count(const uint16_t *a, const uint16_t *bw)
{
    uint8_t cnt = 0;
    for (int i = 0; i < 8; i++) cnt = (uint8_t)(cnt + __popc((uint32_t)(uint8_t)(a[i] ^ bw[i])));
    return cnt;
}

__device__ inline void v2d_begin_frame(V2D &v, const FrameArgs &a, WaveLds &lds)   /* :772-822 */
{
    v.field_state = FIELD_NEW;
    v.good_coords_in_field = v.pcm_lines_in_field = 0;
    if (v.reset_stats) {
        v.reset_stats = false;
        v.n_last = v.nfv = v.nfi = v.n_long = 0;
        coords_clear(v.frame_avg);
        bin_set_good_parameters_reset(v.bin, a.preset);
    }
    coords_clear(v.frame_avg);
    if (!a.preset.en_force_coords) {
        uint32_t k;
        if (median_keys(lds.long_keys, v.n_long, &k)) v.frame_avg = key_to_coords(k, a.doubled != 0);
        if (coords_valid(v.frame_avg)) bin_set_data_coordinates2(v.bin, v.frame_avg.start, v.frame_avg.stop);
    }
}

/* service lines: Binarizer::processLine :539-568 + VideoToDigital :1006-1114 */
__device__ inline void v2d_service_line(V2D &v, const FrameArgs &a, WaveLds &lds, Line &wl, uint32_t frame_no, uint16_t line_no, uint8_t srv)
{
    stc_clear(wl);
    wl.frame_number = frame_no; wl.line_number = line_no;
    set_service(wl, srv);
    if (srv == SDV_SRV_NEW_FILE || srv == SDV_SRV_END_FILE) {
        v.line_in_field_cnt = 0;
        v.n_last = v.nfv = v.nfi = v.n_long = 0;
        if (srv == SDV_SRV_END_FILE || !coords_valid(v.frame_avg)) bin_set_good_parameters_reset(v.bin, a.preset);
    } else if (srv == SDV_SRV_END_FIELD) {
        v.field_state = FIELD_NEW;
        v.line_in_field_cnt = 0;
        v.good_coords_in_field = 0; v.pcm_lines_in_field = 0;
        for (int i = 0; i < 8; i++) v.last_words[i] = 0;
    }
}

/* regular line, after Binarizer::processLine: VideoToDigital :1115-1634 */
__device__ inline void v2d_post_line(V2D &v, const FrameArgs &a, WaveLds &lds, Line &wl, uint32_t *fv_keys, uint32_t *fi_keys, bool even_line)
{
    const sdv_bin_preset &ps = a.preset;
    if (wl.service != SDV_SRV_NO) {
        /* Control Block (setServCtrlBlk inside processLine) */
        if (wl.service == SDV_SRV_CTRL_BLOCK && v.field_state == FIELD_NEW) v.field_state = FIELD_SAFE;
        return;
    }
    bool count_has_data = has_markers(wl);
    bool count_has_pcm = crc_valid(wl) || count_has_data;
    wl.m2 = a.m2_format != 0;
    if (count_has_pcm && v.field_state == FIELD_NEW) v.field_state = FIELD_UNSAFE;
    if (crc_valid(wl)) {
        v.good_coords_in_field++;
        v.q_line_length = (uint16_t)a.width;
        if (a.check_line_copy) {
            if (v.field_state == FIELD_UNSAFE) {
                bin_set_good_parameters(v.bin, ps, wl);
                if (ps.en_first_line_dup) wl.forced_bad = true;
            } else {
                uint8_t diff = words_diff_bit_count(wl.words, v.last_words);
                bool same_words = diff <= (BITS_DATA / BIT_DIFF_THRES_DIV);
                if (!stc_is_almost_silent(wl) && same_words) { wl.forced_bad = true; if (!even_line) v.q_dup_odd++; else v.q_dup_even++; }
            }
        }
        if (crc_valid_ignore_forced(wl)) {
            uint32_t key = coords_key(wl.coords.start, wl.coords.stop);
            /* last_valid_coord_list: push_back, keep the newest COORD_HISTORY_DEPTH */
            SDV_WAVE_SYNC();
            if (lane_id() == 0) {
                if (v.n_last == COORD_HISTORY_DEPTH) for (int i = 0; i < COORD_HISTORY_DEPTH - 1; i++) lds.lv_keys[i] = lds.lv_keys[i + 1];
                lds.lv_keys[v.n_last == COORD_HISTORY_DEPTH ? COORD_HISTORY_DEPTH - 1 : v.n_last] = key;
            }
            if (v.n_last < COORD_HISTORY_DEPTH) v.n_last++;
            SDV_WAVE_SYNC();
            fv_keys[v.nfv++] = key;
            if (a.coordinate_damper && !ps.en_force_coords && (v.n_last > (COORD_HISTORY_DEPTH / 2))) {
                Coords target; coords_clear(target);
                uint32_t k;
                if (median_keys(lds.lv_keys, v.n_last, &k)) target = key_to_coords(k, false);
                if (!coords_valid(target)) target = v.frame_avg;
                if (coords_valid(target)) {
                    int16_t ds = (int16_t)(wl.coords.start - target.start), de = (int16_t)(wl.coords.stop - target.stop);
                    uint8_t in_delta = (uint8_t)(((uint8_t)(wl.psm / 128u)) * 3);
                    bool warn = ((int)ds <= -(int)in_delta) || ((int)ds >= (int)in_delta) || ((int)de <= -(int)in_delta) || ((int)de >= (int)in_delta);
                    if (warn) wl.forced_bad = true;
                }
            }
        }
        if (crc_valid(wl)) bin_set_good_parameters(v.bin, ps, wl);
        else { if (!even_line) v.q_bad_odd++; else v.q_bad_even++; }
        v.field_state = FIELD_INIT;
    } else {
        if (v.q_line_length == 0) v.q_line_length = (uint16_t)a.width;
        if (coords_valid(wl.coords)) fi_keys[v.nfi++] = coords_key(wl.coords.start, wl.coords.stop);
        if (count_has_data) {
            Coords preset_coords; coords_clear(preset_coords);
            if (!even_line) v.q_bad_odd++; else v.q_bad_even++;
            if (!ps.en_force_coords) {
                uint32_t k;
                if (median_keys(lds.lv_keys, v.n_last, &k)) preset_coords = key_to_coords(k, a.doubled != 0);
                if (!coords_valid(preset_coords)) preset_coords = v.frame_avg;
            }
            v.field_state = FIELD_INIT;
            bin_set_data_coordinates(v.bin, preset_coords);
            bin_set_bw_levels(v.bin, ps, 0, 0);
        } else {
            bin_set_bw_levels(v.bin, ps, 0, 0);
        }
    }
    if (!even_line) v.q_odd++; else v.q_even++;
    if (count_has_pcm) {
        if (!even_line) v.q_pcm_odd++; else v.q_pcm_even++;
        v.pcm_lines_in_field++;
        for (int i = 0; i < 8; i++) v.last_words[i] = wl.words[i];
    }
    v.line_in_field_cnt++;
}

/* END_FRAME bookkeeping (videotodigital.cpp:1636-1714) */
/* uniform_key: every entry of fv_keys is known to hold this key (a frame that was captured whole) - its median without reading them */
__device__ inline void v2d_end_frame(V2D &v, const FrameArgs &a, WaveLds &lds, uint32_t frame_no, const uint32_t *fv_keys, const uint32_t *fi_keys, sdv_frame_stats *out,
                                  bool have_uniform_key = false, uint32_t uniform_key = 0)
{
    SDV_BLOCK_SYNC();           /* the coordinate keys of the frame's lines were stored to global memory by other lanes than the ones that read them below: wait for them */
    if (v.q_pcm_odd > v.q_odd) v.q_pcm_odd = v.q_odd;
    if (v.q_pcm_even > v.q_even) v.q_pcm_even = v.q_even;
    if (v.q_bad_odd > v.q_odd) v.q_bad_odd = v.q_odd;
    if (v.q_bad_even > v.q_even) v.q_bad_even = v.q_even;
    bool not_sure = false;
    uint32_t k;
    coords_clear(v.frame_avg);
    if (have_uniform_key && v.nfv > 0) v.frame_avg = key_to_coords(uniform_key, a.doubled != 0);
    else if (median_keys(fv_keys, v.nfv, &k)) v.frame_avg = key_to_coords(k, a.doubled != 0);
    if (coords_valid(v.frame_avg)) {
        SDV_WAVE_SYNC();
        if (lane_id() == 0) {
            if (v.n_long == COORD_LONG_HISTORY) for (int i = 0; i < COORD_LONG_HISTORY - 1; i++) lds.long_keys[i] = lds.long_keys[i + 1];
            lds.long_keys[v.n_long == COORD_LONG_HISTORY ? COORD_LONG_HISTORY - 1 : v.n_long] = coords_key(v.frame_avg.start, v.frame_avg.stop);
        }
        if (v.n_long < COORD_LONG_HISTORY) v.n_long++; else v.long_pushes++;
        SDV_WAVE_SYNC();
    } else {
        coords_clear(v.frame_avg);
        if (median_keys(fi_keys, v.nfi, &k)) v.frame_avg = key_to_coords(k, a.doubled != 0);
        if (!coords_valid(v.frame_avg)) { coords_clear(v.frame_avg); if (median_keys(lds.long_keys, v.n_long, &k)) v.frame_avg = key_to_coords(k, a.doubled != 0); }
        not_sure = true;
    }
    v.nfv = v.nfi = 0;
    if (lane_id() == 0) {
        sdv_frame_stats s;
        s.frame_id = frame_no; s.line_length = v.q_line_length;
        s.lines_odd = v.q_odd; s.lines_even = v.q_even; s.lines_pcm_odd = v.q_pcm_odd; s.lines_pcm_even = v.q_pcm_even;
        s.lines_bad_odd = v.q_bad_odd; s.lines_bad_even = v.q_bad_even; s.lines_dup_odd = v.q_dup_odd; s.lines_dup_even = v.q_dup_even;
        s.data_start = v.frame_avg.start; s.data_stop = v.frame_avg.stop;
        s.data_from_doubled = v.frame_avg.doubled ? 1 : 0; s.data_not_sure = not_sure ? 1 : 0;
        s._pad[0] = s._pad[1] = s._pad[2] = s._pad[3] = 0;
        *out = s;
    }
    v.q_line_length = v.q_odd = v.q_even = v.q_pcm_odd = v.q_pcm_even = v.q_bad_odd = v.q_bad_even = v.q_dup_odd = v.q_dup_even = 0;
}

/* re-declare the whole hot state wave-uniform (after loading it from memory) */
__device__ inline void v2d_make_uniform(V2D &v)
{
    Bin &b = v.bin;
    b.in_black = (uint8_t)uni(b.in_black); b.in_white = (uint8_t)uni(b.in_white); b.in_ref = (uint8_t)uni(b.in_ref);
    b.in_coord.start = (int16_t)uni(b.in_coord.start); b.in_coord.stop = (int16_t)uni(b.in_coord.stop); b.in_coord.doubled = uni(b.in_coord.doubled) != 0;
    b.in_max_hyst = (uint8_t)uni(b.in_max_hyst); b.in_max_shift = (uint8_t)uni(b.in_max_shift);
    b.do_ref_lvl_sweep = uni(b.do_ref_lvl_sweep) != 0; b.mode = (uint8_t)uni(b.mode);
    b.hyst_lim = (uint8_t)uni(b.hyst_lim); b.shift_lim = (uint8_t)uni(b.shift_lim);
    b.line_length = (uint16_t)uni(b.line_length); b.scan_start = (uint16_t)uni(b.scan_start); b.scan_end = (uint16_t)uni(b.scan_end);
    b.mark_start_max = (uint16_t)uni(b.mark_start_max); b.mark_end_min = (uint16_t)uni(b.mark_end_min); b.estimated_ppb = (uint16_t)uni(b.estimated_ppb);
    b.was_bw_scanned = uni(b.was_bw_scanned) != 0; b.vl_doubled = uni(b.vl_doubled) != 0;
    v.field_state = (uint8_t)uni(v.field_state); v.reset_stats = uni(v.reset_stats) != 0;
    v.good_coords_in_field = (uint16_t)uni(v.good_coords_in_field); v.pcm_lines_in_field = (uint16_t)uni(v.pcm_lines_in_field);
    v.line_in_field_cnt = (uint16_t)uni(v.line_in_field_cnt);
    v.frame_avg.start = (int16_t)uni(v.frame_avg.start); v.frame_avg.stop = (int16_t)uni(v.frame_avg.stop); v.frame_avg.doubled = uni(v.frame_avg.doubled) != 0;
    v.n_last = uni(v.n_last); v.n_long = uni(v.n_long); v.nfv = uni(v.nfv); v.nfi = uni(v.nfi);
    for (int i = 0; i < 8; i++) v.last_words[i] = (uint16_t)uni(v.last_words[i]);
    v.q_line_length = (uint16_t)uni(v.q_line_length); v.q_odd = (uint16_t)uni(v.q_odd); v.q_even = (uint16_t)uni(v.q_even);
    v.q_pcm_odd = (uint16_t)uni(v.q_pcm_odd); v.q_pcm_even = (uint16_t)uni(v.q_pcm_even); v.q_bad_odd = (uint16_t)uni(v.q_bad_odd);
    v.q_bad_even = (uint16_t)uni(v.q_bad_even); v.q_dup_odd = (uint16_t)uni(v.q_dup_odd); v.q_dup_even = (uint16_t)uni(v.q_dup_even);
}

/* The model of the chain (engine.inc, chain speculation): the state `m` frames behind s0 when every line of those frames decodes with the inherited
 * tuning - presets unchanged, the 9-entry window saturated with the one coordinate pair, one entry of it per frame pushed into the 16-frame history. */
/* (halfword by halfword: the frame kernels make a state a dword per lane, the predict kernel a whole state per thread - one model for both) */
enum { V2D_STATE_HALVES = sizeof(sdv_v2d_state) / 2, V2D_H_LAST = 9, V2D_H_LONG = 27 };      /* halfword 9: last_valid[0], halfword 27: long_valid[0] */
static_assert(offsetof(sdv_v2d_state, last_valid) == 2 * V2D_H_LAST && offsetof(sdv_v2d_state, long_valid) == 2 * V2D_H_LONG && offsetof(sdv_v2d_state, n_last_valid) == 12 &&
              offsetof(sdv_v2d_state, long_valid_doubled_mask) == 16 && sizeof(sdv_v2d_state) == 120, "the chain state is put together halfword by halfword");
__device__ inline uint16_t predict_half(const sdv_v2d_state &s0, int m, bool doubled, uint8_t min_ref_lvl, int h)
{
    const uint16_t *raw = reinterpret_cast<const uint16_t *>(&s0);
    /* What the frames in between will measure: what the last frame measured - the newest entry of the multi-frame history (on a
     * tape that plays every entry is the same; after a jump of the data window the history is a mix for 16 frames, and a frame
     * leaves the binarizer tuned to what it saw, not to the median it was started with, videotodigital.cpp:808-821, :1369) */
    int16_t cs = s0.bin.in_def_start, ce = s0.bin.in_def_stop;
    const int n_long0 = s0.n_long_valid;
    if (n_long0 > 0) { cs = (int16_t)raw[V2D_H_LONG + 2 * (n_long0 - 1)]; ce = (int16_t)raw[V2D_H_LONG + 2 * (n_long0 - 1) + 1]; }
    const bool steady = (s0.bin.in_def_reference >= min_ref_lvl) && pod_coords_valid(cs, ce) && !s0.reset_stats;
    const uint16_t same = raw[h];
    if (!steady) return same;
    const int total = n_long0 + m;
    const int keep = total > COORD_LONG_HISTORY ? COORD_LONG_HISTORY : total;
    const int drop = total - keep;             /* oldest entries that fell out of the window */
    if (h == 2) return (uint16_t)cs;
    if (h == 3) return (uint16_t)ce;
    if (h == 4) return (uint16_t)((same & 0xFF00u) | (doubled ? 1u : 0u));                          /* in_def_from_doubled */
    if (h == 6) return (uint16_t)(COORD_HISTORY_DEPTH | (keep << 8));                               /* n_last_valid, n_long_valid */
    if (h == 7) return doubled ? (uint16_t)((1u << COORD_HISTORY_DEPTH) - 1u) : (uint16_t)0;        /* last_valid_doubled_mask */
    if (h == 8) return doubled ? (uint16_t)((1u << keep) - 1u) : (uint16_t)0;                       /* long_valid_doubled_mask */
    if (h >= V2D_H_LAST && h < V2D_H_LONG) return (uint16_t)(((h - V2D_H_LAST) & 1) ? ce : cs);
    if (h >= V2D_H_LONG && h < V2D_H_LONG + 2 * COORD_LONG_HISTORY) {
        const int i = (h - V2D_H_LONG) >> 1, which = (h - V2D_H_LONG) & 1, src = i + drop;
        if (i >= keep) return 0;
        if (src < n_long0) return raw[V2D_H_LONG + 2 * src + which];
        return (uint16_t)(which ? ce : cs);
    }
    return same;
}
__device__ inline sdv_v2d_state predict_state(const sdv_v2d_state &s0, int m, bool doubled, uint8_t min_ref_lvl)
{
    sdv_v2d_state p;
    uint16_t *out = reinterpret_cast<uint16_t *>(&p);
    for (int h = 0; h < (int)V2D_STATE_HALVES; h++) out[h] = predict_half(s0, m, doubled, min_ref_lvl, h);
    return p;
}

/* The state frame f is started from, a dword per lane in WaveLds::state_in: what the chain speculation put into states_in[f], or (FrameArgs::predict_in_kernel)
 * the model's state f - base_frame frames behind `base`, made here.  dword `dw` of the state frame f + 1 was / will be started from: v2d_next_state_dword. */
__device__ inline uint32_t v2d_model_dword(const FrameArgs &a, int f, int dw)
{
    const int m = f - a.base_frame;
    if (m <= 0) return reinterpret_cast<const uint32_t *>(&a.base)[dw];
    return (uint32_t)predict_half(a.base, m, a.doubled != 0, a.preset.min_ref_lvl, 2 * dw) | ((uint32_t)predict_half(a.base, m, a.doubled != 0, a.preset.min_ref_lvl, 2 * dw + 1) << 16);
}
__device__ inline void v2d_stage_state_in(const FrameArgs &a, WaveLds &lds, int f)
{
    enum { NDW = sizeof(sdv_v2d_state) / 4 };
    static_assert(NDW <= 32, "WaveLds::state_in holds a chain state");
    const int lane = lane_id();
    SDV_WAVE_SYNC();
    if (lane < NDW) lds.state_in[lane] = a.predict_in_kernel ? v2d_model_dword(a, f, lane) : reinterpret_cast<const uint32_t *>(&a.states_in[f])[lane];
    SDV_WAVE_SYNC();
}
__device__ inline const sdv_v2d_state *v2d_state_in(const WaveLds &lds) { return reinterpret_cast<const sdv_v2d_state *>(lds.state_in); }

/* ---- chain state <-> registers ---------------------------------------------------------------- */
__device__ inline void v2d_load_state(V2D &v, WaveLds &lds, const sdv_v2d_state *s, const FrameArgs &a)
{
    v.bin.in_black = s->bin.in_def_black; v.bin.in_white = s->bin.in_def_white; v.bin.in_ref = s->bin.in_def_reference;
    v.bin.in_coord.start = s->bin.in_def_start; v.bin.in_coord.stop = s->bin.in_def_stop; v.bin.in_coord.doubled = s->bin.in_def_from_doubled != 0;
    v.bin.do_ref_lvl_sweep = s->do_ref_lvl_sweep != 0;
    bin_set_mode(v.bin, a.mode);
    v.bin.hyst_lim = 0; v.bin.shift_lim = 0;
    v.bin.line_length = 0; v.bin.scan_start = v.bin.scan_end = 0; v.bin.mark_start_max = 0; v.bin.mark_end_min = 0xFFFF; v.bin.estimated_ppb = 0;
    v.bin.was_bw_scanned = false; v.bin.vl_doubled = false;
    v.reset_stats = s->reset_stats != 0;
    v.n_last = s->n_last_valid; v.n_long = s->n_long_valid; v.long_pushes = 0;
    for (int i = 0; i < COORD_HISTORY_DEPTH; i++) lds.lv_keys[i] = coords_key(s->last_valid[i].data_start, s->last_valid[i].data_stop);
    for (int i = 0; i < COORD_LONG_HISTORY; i++) lds.long_keys[i] = coords_key(s->long_valid[i].data_start, s->long_valid[i].data_stop);
    v.field_state = FIELD_INIT;
    v.good_coords_in_field = v.pcm_lines_in_field = v.line_in_field_cnt = 0;
    coords_clear(v.frame_avg);
    v.nfv = v.nfi = 0;
    for (int i = 0; i < 8; i++) v.last_words[i] = 0;
    v.q_line_length = v.q_odd = v.q_even = v.q_pcm_odd = v.q_pcm_even = v.q_bad_odd = v.q_bad_even = v.q_dup_odd = v.q_dup_even = 0;
    v2d_make_uniform(v);
}
/* halfword h of the outgoing chain state */
__device__ inline uint32_t v2d_state_half(const V2D &v, const WaveLds &lds, const FrameArgs &a, int h)
{
    if (h >= V2D_H_LAST && h < V2D_H_LONG + 2 * COORD_LONG_HISTORY) {
        const bool lng = h >= V2D_H_LONG;
        const int k = h - (lng ? V2D_H_LONG : V2D_H_LAST), e = k >> 1;
        if (e >= (lng ? v.n_long : v.n_last)) return 0;
        const uint32_t key = lng ? lds.long_keys[e] : lds.lv_keys[e];
        return (uint32_t)(uint16_t)((k & 1) ? key_stop(key) : key_start(key));
    }
    switch (h) {
    case 0: return (uint32_t)v.bin.in_black | ((uint32_t)v.bin.in_white << 8);
    case 1: return (uint32_t)v.bin.in_ref;                                                  /* (sdv_bin_state::do_ref_lvl_sweep = 0: the flag has its own field) */
    case 2: return (uint32_t)(uint16_t)v.bin.in_coord.start;
    case 3: return (uint32_t)(uint16_t)v.bin.in_coord.stop;
    case 4: return v.bin.in_coord.doubled ? 1u : 0u;
    case 5: return (v.bin.do_ref_lvl_sweep ? 1u : 0u) | (v.reset_stats ? 0x100u : 0u);
    case 6: return (uint32_t)(uint8_t)v.n_last | ((uint32_t)(uint8_t)v.n_long << 8);
    case 7: return a.doubled ? (uint32_t)(uint16_t)((1u << v.n_last) - 1u) : 0u;
    case 8: return a.doubled ? (uint32_t)(uint16_t)((1u << v.n_long) - 1u) : 0u;
    default: return 0;                                                                      /* _pad */
    }
}
/* The check of the chain, by the frame itself: was the next frame started from the state this one hands on (`mine`: dword `lane` of it)?  (and: is it the state
 * the frame itself was started from - a frame that hands on what it got tells nothing new, one that does not has most likely tuned itself to its own pixels) */
__device__ inline uint8_t v2d_link_flags(const FrameArgs &a, const WaveLds &lds, int f, uint32_t mine)
{
    enum { NDW = sizeof(sdv_v2d_state) / 4 };
    const int lane = lane_id();
    const bool in = lane < NDW;
    const int dw = in ? lane : 0;
    uint8_t fl = VF_OK;
    bool hist_off = false;
    if (f + 1 < a.n_total) {
        const uint32_t next = a.predict_in_kernel ? v2d_model_dword(a, f + 1, dw) : reinterpret_cast<const uint32_t *>(&a.states_in[f + 1])[dw];
        if (__ballot(in && next != mine) != 0ull) {
            fl = VF_BREAK;
            /* long_valid[]: halfwords V2D_H_LONG .. V2D_H_LONG + 31 = the high half of dword 13, dwords 14 .. 28, the low half of dword 29 */
            const uint32_t d = next ^ mine;
            hist_off = __ballot(in && ((dw == 13 && (d & 0xFFFF0000u)) || (dw >= 14 && dw <= 28 && d) || (dw == 29 && (d & 0x0000FFFFu)))) != 0ull;
        }
    }
    if (fl == VF_BREAK) {       /* (only asked of a frame whose link broke: on a tape that plays this is skipped) */
        /* ... "what it was started from" as the model sees it: one frame on, with the inherited tuning */
        const sdv_v2d_state &was = *v2d_state_in(lds);
        const uint32_t own = (uint32_t)predict_half(was, 1, a.doubled != 0, a.preset.min_ref_lvl, 2 * dw) |
                             ((uint32_t)predict_half(was, 1, a.doubled != 0, a.preset.min_ref_lvl, 2 * dw + 1) << 16);
        /* dword 0: in_def_black, in_def_white, in_def_reference (+ a pad byte); dword 2, byte 2: do_ref_lvl_sweep - the rest are coordinates and histories */
        const uint32_t d = mine ^ own;
        if (__ballot(in && dw >= 1 && (dw == 2 ? d & 0xFF00FFFFu : d) != 0) != 0ull) fl |= VF_MOVED;
        if (__ballot(in && ((dw == 0 && d != 0) || (dw == 2 && (d & 0x00FF0000u) != 0))) != 0ull) fl |= VF_RETUNED;
        if (hist_off) fl |= VF_HIST;
    }
    return fl;
}
/* A frame the round need not decode again (FrameArgs::skip: the state it is started from is the one its last decode was started from, and that decode went to
 * the end of the frame): records, descriptor and the state it hands on are in place; what may have changed is what its successor is started from. */
__device__ inline void v2d_relink(const FrameArgs &a, const WaveLds &lds, int f)
{
    enum { NDW = sizeof(sdv_v2d_state) / 4 };
    const int lane = lane_id();
    const uint32_t mine = reinterpret_cast<const uint32_t *>(&a.states_out[f])[lane < NDW ? lane : 0];
    const uint8_t keep = (uint8_t)(uniu(a.flag[f]) & VF_SLOW);
    const uint8_t fl = v2d_link_flags(a, lds, f, mine);
    SDV_WAVE_SYNC();
    if (lane == 0) a.flag[f] = (uint8_t)(fl | keep);
}
/* The state goes out a dword per lane (lanes 0 .. 29), and the check of the chain is a ballot.  (It
 * was put together by lane 0 alone on the stack before: the histories are indexed by how full they are, and that was the lean kernel's scratch memory.) */
__device__ inline void v2d_store_state(const V2D &v, const WaveLds &lds, sdv_v2d_state *s, const FrameArgs &a, bool unsettled = false, uint8_t extra_flags = 0)
{
    enum { NDW = sizeof(sdv_v2d_state) / 4 };
    const int lane = lane_id();
    const bool in = lane < NDW;
    const int dw = in ? lane : 0;
    SDV_WAVE_SYNC();            /* (the histories were written by lane 0) */
    const uint32_t mine = v2d_state_half(v, lds, a, 2 * dw) | (v2d_state_half(v, lds, a, 2 * dw + 1) << 16);
    const int f = (int)(s - a.states_out);
    if (in) reinterpret_cast<uint32_t *>(s)[dw] = mine;
    /* the last frame of the call: its state also behind the flags, where the host's one read-back per round finds it (engine.inc: tail_ofs) */
    if (in && f == a.n_total - 1) reinterpret_cast<uint32_t *>(a.flag + (((size_t)a.n_total + 15) & ~(size_t)15))[dw] = mine;
    uint8_t fl = v2d_link_flags(a, lds, f, mine);
    if (unsettled) fl = VF_ABORTED;         /* a sweep is owed to this frame: to be decoded again, from the same state */
    if (lane == 0) {
        a.flag[f] = (uint8_t)(fl | extra_flags);
        if (a.refs) {
            /* (one pair pushed into a full history, the rest moved down a slot: counted where it happens - reading the incoming history again here cost 3 % of the kernel) */
            const bool pushed = v.long_pushes == 1 && v.n_long == COORD_LONG_HISTORY && !a.doubled;
            a.refs[3 * f] = v2d_state_in(lds)->bin.in_def_reference; a.refs[3 * f + 1] = v.bin.in_ref; a.refs[3 * f + 2] = pushed ? 1 : 0;
        }
    }
}
/* Where the line a lean wave gives its frame up on begins: the first pixel of the staged row at or above the reference level (the rising edge of the START
 * marker on a line that holds PCM), in pixels of an undoubled line, 0xFF when there is none below 254.  Not a measurement anything is decoded with: frames that
 * give up side by side because the data window is no longer where their state says tell the scheduler by it whether they look at one window or at several
 * (engine.inc: only the first frame of each goes to the general kernel, the others wait for what it finds). */
__device__ inline uint8_t give_up_signature(const FrameArgs &a, const WaveLds &lds, uint8_t ref_level)
{
    const int lane = lane_id();
    const int base = 16 * lane;
    uint32_t first = 0xFFFFu;
    if (base < a.width && base < 1024) {
#pragma unroll 1
        for (int i = 15; i >= 0; i--) { const int x = base + i; if (x < a.width && lds.px[x] >= ref_level) first = (uint32_t)x; }
    }
    first = wave_min_u32(first);
    if (a.doubled) first >>= 1;
    return first < 254u ? (uint8_t)first : (uint8_t)0xFF;
}
/* a frame given up: its state goes out as it came in, marked (dword 29, byte 2 = _pad[0]) */
__device__ inline void v2d_give_up(const FrameArgs &a, const WaveLds &lds, int f, uint8_t sig = 0xFF)
{
    enum { NDW = sizeof(sdv_v2d_state) / 4 };
    static_assert(offsetof(sdv_v2d_state, _pad) == 118, "the mark of a frame given up");
    const int lane = lane_id();
    if (lane < NDW) {
        uint32_t d = lds.state_in[lane];
        if (lane == NDW - 1) d = (d & 0xFF00FFFFu) | ((uint32_t)0xA5 << 16);      /* STATE_ABORTED */
        reinterpret_cast<uint32_t *>(&a.states_out[f])[lane] = d;
    }
    if (lane == 0) { a.flag[f] = VF_ABORTED; if (a.sig) a.sig[f] = sig; }
}

/* ---------------------------------------------------------------------------------------------
 * A pass that meets the last one (general kernel, damaged tapes).  A frame of a damaged tape is decoded several times - from a guessed state, again when its
 * sweeps are settled, again when its predecessor turned out to hand on another state - and every pass walks the whole frame although the passes differ only
 * up to the first line that tunes the binarizer to its own pixels again (a lost line takes the black and white presets back, a sweep sets the reference
 * level): from the line on where the state of this pass equals the state the last pass had there, this pass would only do the same thing again, record for record.
 * So a complete pass leaves snapshots of the state ahead of the lines it took one by one (where the state can change), and a later pass that arrives at such a
 * line with the same state stops there: the records behind are in place, the counters and coordinate keys of the rest of the frame are added from what the
 * last pass noted, and the frame ends from the last pass's final state.
 * What "state" is: everything a line's decode and its bookkeeping read - presets of the binarizer, field state, the 9-line window, the previous line's words,
 * the frame's start coordinates (TcSnap dwords 0 .. TC_CMP_DWORDS-1).  What is only counted (frame statistics, the lists of coordinate keys) is carried as
 * differences.  A pass is "complete" when it ran to the end of the frame with every sweep it asked for at hand (the memo only grows, so the same lookups hit again).
 * --------------------------------------------------------------------------------------------- */
enum { TC_ENTRIES = 64, TC_SNAP_DWORDS = 32, TC_CMP_DWORDS = 17, TC_USED_DWORDS = 25, TC_FINAL = TC_ENTRIES - 1, TC_POS_NONE = 0xFFFF };
struct TcSnap { uint32_t d[TC_SNAP_DWORDS]; };
/* dword `dw` of the snapshot of the state as it is now, ahead of line `pos` (= field * 1024 + index in the field) */
__device__ inline uint32_t tc_dword(const V2D &v, const WaveLds &lds, int dw, uint32_t pos, bool used_general)
{
    switch (dw) {
    case 0: case 1: case 2: case 3: case 4: case 5: case 6: case 7: case 8: return dw < v.n_last ? lds.lv_keys[dw] : 0u;
    case 9: return (uint32_t)v.last_words[0] | ((uint32_t)v.last_words[1] << 16);
    case 10: return (uint32_t)v.last_words[2] | ((uint32_t)v.last_words[3] << 16);
    case 11: return (uint32_t)v.last_words[4] | ((uint32_t)v.last_words[5] << 16);
    case 12: return (uint32_t)v.last_words[6] | ((uint32_t)v.last_words[7] << 16);
    case 13: return (uint32_t)(uint16_t)v.bin.in_coord.start | ((uint32_t)(uint16_t)v.bin.in_coord.stop << 16);
    case 14: return (uint32_t)(uint16_t)v.frame_avg.start | ((uint32_t)(uint16_t)v.frame_avg.stop << 16);
    case 15: return (uint32_t)v.bin.in_black | ((uint32_t)v.bin.in_white << 8) | ((uint32_t)v.bin.in_ref << 16) |
                    ((v.bin.in_coord.doubled ? 1u : 0u) | (v.bin.do_ref_lvl_sweep ? 2u : 0u) | (v.frame_avg.doubled ? 4u : 0u)) << 24;
    case 16: return (uint32_t)v.field_state | ((uint32_t)(uint8_t)v.n_last << 8);
    /* (17: reserved) ... what was counted so far: */
    case 18: return (pos & 0xFFFFu) | (used_general ? 0x10000u : 0u);
    case 19: return (uint32_t)(uint16_t)v.nfv | ((uint32_t)(uint16_t)v.nfi << 16);
    case 20: return (uint32_t)v.q_line_length | ((uint32_t)v.q_odd << 16);
    case 21: return (uint32_t)v.q_even | ((uint32_t)v.q_pcm_odd << 16);
    case 22: return (uint32_t)v.q_pcm_even | ((uint32_t)v.q_bad_odd << 16);
    case 23: return (uint32_t)v.q_bad_even | ((uint32_t)v.q_dup_odd << 16);
    case 24: return (uint32_t)v.q_dup_even;
    default: return 0u;
    }
}
__device__ inline void tc_write(TcSnap *dst, const V2D &v, const WaveLds &lds, uint32_t pos, bool used_general)
{
    const int lane = lane_id();
    SDV_WAVE_SYNC();                    /* (the 9-line window is written by lane 0) */
    if (lane < TC_USED_DWORDS) dst->d[lane] = tc_dword(v, lds, lane, pos, used_general);
}
/* the counters of a snapshot staged in LDS (w[0 .. 31]) */
struct TcCounts { int nfv, nfi; uint16_t q[9]; bool used_general; };
__device__ inline TcCounts tc_counts(const uint32_t *w)
{
    TcCounts c;
    c.used_general = (w[18] & 0x10000u) != 0;
    c.nfv = (int)(w[19] & 0xFFFF); c.nfi = (int)(w[19] >> 16);
    c.q[0] = (uint16_t)w[20]; c.q[1] = (uint16_t)(w[20] >> 16); c.q[2] = (uint16_t)w[21]; c.q[3] = (uint16_t)(w[21] >> 16); c.q[4] = (uint16_t)w[22];
    c.q[5] = (uint16_t)(w[22] >> 16); c.q[6] = (uint16_t)w[23]; c.q[7] = (uint16_t)(w[23] >> 16); c.q[8] = (uint16_t)w[24];
    return c;
}

/* Scanline staging: HBM -> registers (coalesced 16-byte loads, issued one line AHEAD so the HBM latency
 * hides under the decode of the current line) -> LDS.  Rows that are not 16-byte aligned take the slow
 * byte path at commit time. */
struct RowPrefetch {
    uint4 v0, v1; const uint8_t *row;
    uint4 vq[SDV_ROWQ]; const uint8_t *rowq[SDV_ROWQ];
    int nq;             /* multi-line loop: the nq rows after `row` in decode order that are already fetched */
    bool vec;           /* every row of the frame starts 16-byte aligned: 16-byte vectors, else the byte path */
    int i0;             /* this lane's vector of a row, clamped to the last one (lanes past the row reload it: no exec juggling) */
};

/* `row` is always a readable row of the frame (the caller passes any valid row when there is no next one) */
__device__ inline void row_prefetch(RowPrefetch &pf, const uint8_t *row, int width)
{
    pf.row = row;
    if (pf.vec) {
        const uint4 *src = (const uint4 *)row;
        pf.v0 = src[pf.i0];
        if (width > 1024) { int lane = lane_id(), nvec = width >> 4; if (lane + 64 < nvec) pf.v1 = src[lane + 64]; }
    }
}
__device__ inline void row_commit(WaveLds &lds, const RowPrefetch &pf, int width)
{
    SDV_WAVE_SYNC();
    int lane = lane_id();
    const uint8_t *row = pf.row;
    if (pf.vec) {
        uint4 *dst = (uint4 *)lds.px;
        dst[lane] = pf.v0;                      /* lanes past the row write into the unused end of px (64 x 16 <= SDV_MAX_WIDTH) */
        if (width > 1024) { int nvec = width >> 4; if (lane + 64 < nvec) dst[lane + 64] = pf.v1; }
        /* (the byte loops are the odd geometries' path: what they derive from the lane number is worked out here, not ahead of the frame loop, which
         * has no registers to keep it in) */
        if (width & 15) {
            int i0 = (width & ~15) + lane;
            SDV_OPAQUE(i0);
#pragma unroll 1
            for (int i = i0; i < width; i += 64) lds.px[i] = row[i];
        }
    } else {
        int i0 = lane;
        SDV_OPAQUE(i0);
#pragma unroll 1
        for (int i = i0; i < width; i += 64) lds.px[i] = row[i];
    }
    SDV_WAVE_SYNC();
}

__device__ inline void emit_record(const Line &wl, sdv_line_rec *dst)
{
    if (lane_id() == 0) {
        sdv_line_rec r; line_to_rec(wl, &r);
        /* (a service line's record is constants but for three fields, and the compiler sets such a record up dwords ahead of the loops it is stored
         * in - in registers the frame loop has to spill.  With one field it cannot see through, the record is made where it is stored.) */
        uint32_t fno = r.frame_number; SDV_OPAQUE(fno); r.frame_number = fno;
        uint32_t ref = r.ref_level; SDV_OPAQUE(ref); r.ref_level = (uint8_t)ref;
        uint32_t crc = r.calc_crc; SDV_OPAQUE(crc); r.calc_crc = (uint16_t)crc;
        *dst = r;
    }
}

/* ======================================================================================== */
/* Kernel: one wavefront (= one workgroup of 64) per frame                                   */
/* ======================================================================================== */
/* ---------------------------------------------------------------------------------------------
 * Fast path: the steady state of a good recording.  Everything the previous good line tuned is preset
 * (B/W levels, reference level, data coordinates), so Binarizer::processLine goes STG_INPUT_ALL ->
 * readPCMdata -> STG_DATA_OK (binarizer.cpp:774-931, 1534-1621) and VideoToDigital only does its
 * per-line bookkeeping (videotodigital.cpp:1155-1396, 1524-1633).  This function restates exactly that
 * branch with the minimum of live state; whenever any condition of the branch does not hold it returns
 * false WITHOUT side effects and the general slow_line() decodes the line from scratch.
 * --------------------------------------------------------------------------------------------- */
struct Geo { int16_t start, stop; uint32_t psm; int32_t vp0, vp1; bool valid; };      /* cached bit-cell centres */
struct LaneConst { uint64_t klo, khi; };                                              /* CRC parity masks of this lane */
struct FastBits { uint64_t s_lo, s_hi; uint16_t calc_crc; uint8_t ref_low, ref_high, h, s; bool ctrl_block; };

/* Control Block pattern (stc007line.cpp:493-504: words 0..3 = 0x3333 0x0CCC 0x3333 0x0CCC, word 4 = 0, bits 4..11 of word 7 = 0)
 * tested on the raw cells: cell 14k+i holds bit 13-i of word k, lane i owns cells i (s_lo) and i+64 (s_hi). */
constexpr uint64_t ctrl_cells_0_55()
{
    const uint16_t w[4] = { 0x3333, 0x0CCC, 0x3333, 0x0CCC };
    uint64_t m = 0;
    for (int k = 0; k < 4; k++) for (int i = 0; i < 14; i++) if ((w[k] >> (13 - i)) & 1) m |= 1ull << (14 * k + i);
    return m;
}
__device__ __forceinline__ bool ctrl_block_cells(uint64_t s_lo, uint64_t s_hi)
{
    constexpr uint64_t K = ctrl_cells_0_55();
    if ((uint32_t)s_lo != (uint32_t)K) return false;            /* one compare settles nearly every line */
    return (s_lo & ((1ull << 56) - 1)) == K && (s_lo >> 56) == 0 && (s_hi & 0x3Full) == 0 && (s_hi & (0xFFull << 36)) == 0;
}

/* are the conditions of the STG_INPUT_ALL branch met for the coming line? (wave-uniform, no side effects) */
__device__ inline bool fast_eligible(const FrameArgs &a, const Bin &b)
{
    const sdv_bin_preset &ps = a.preset;
    if (ps.en_force_coords || !ps.en_good_no_marker) return false;
    if (!(are_bw_levels_preset(b, ps) && is_ref_level_preset(b, ps) && coords_valid(b.in_coord))) return false;
    if (a.width < BITS_IN_LINE || (a.width - 1) < BITS_BETWEEN) return false;
    return true;
}

/* readPCMdata under the preset tuning for the scanline staged in LDS: the (hysteresis, shift) ladder of
 * binarizer.cpp:7769-7954 with the wave-parallel fill.  Returns false when no pair gives a valid CRC. */
__device__ inline bool fast_decode(const FrameArgs &a, const WaveLds &lds, const Bin &b, uint8_t black, uint8_t white, Geo &g, const LaneConst &lc, FastBits &o)
{
    const int lane = lane_id();
    const int32_t pixel_start = 0, pixel_stop = a.width - 1;
    if (!g.valid || g.start != b.in_coord.start || g.stop != b.in_coord.stop) {
        Line t; t.pixel_start = (uint16_t)pixel_start; t.pixel_stop = (uint16_t)pixel_stop;
        set_ppb(t, b.in_coord);
        g.start = b.in_coord.start; g.stop = b.in_coord.stop; g.psm = t.psm; g.valid = true;
        g.vp0 = bit_center(t, lane); g.vp1 = bit_center(t, lane + 64);
    }
    uint8_t hyst_lim = b.in_max_hyst, shift_lim = b.in_max_shift;
    if (hyst_lim > HYST_DEPTH_MAX) hyst_lim = HYST_DEPTH_MAX;
    if (shift_lim > SHIFT_STAGES_MAX) shift_lim = SHIFT_STAGES_MAX;
    uint64_t s_lo = 0, s_hi = 0; uint16_t calc_crc = 0; uint8_t ref_low = 0, ref_high = 0; int fh = 0, fs = 0;
    bool found = false;
    for (int h = 0; h <= (int)hyst_lim && !found; h++) {
        ref_low = get_low_level(b.in_ref, (uint8_t)h); ref_high = get_high_level(b.in_ref, (uint8_t)h);
        if (ref_low <= black || ref_high >= white) break;       /* fillDataWords level clipping */
        for (int st = 0; st <= (int)shift_lim; st++) {
            int sh = (st == 0) ? 0 : ((st & 1) ? ((st + 1) >> 1) : -(st >> 1));
            int32_t x0 = g.vp0 + sh, x1 = g.vp1 + sh;
            x0 = x0 < pixel_start ? pixel_start : (x0 >= pixel_stop ? pixel_stop - 1 : x0);
            x1 = x1 < pixel_start ? pixel_start : (x1 >= pixel_stop ? pixel_stop - 1 : x1);
            uint8_t p0 = lds.px[x0], p1 = lds.px[x1];
            uint64_t a_lo = __ballot(p0 > ref_low), b_lo = __ballot(p0 >= ref_high);
            uint64_t a_hi = __ballot(p1 > ref_low), b_hi = __ballot(p1 >= ref_high);
            solve_automaton(a_lo, a_hi, b_lo, b_hi, s_lo, s_hi);
            int par = (__popcll(s_lo & lc.klo) + __popcll(s_hi & lc.khi)) & 1;
            calc_crc = (uint16_t)((uint16_t)(__ballot(par) & 0xFFFF) ^ c_crc.init);
            if (calc_crc == rev16((uint32_t)((s_hi >> 48) & 0xFFFF))) { found = true; fh = h; fs = st; break; }
        }
    }
    if (!found) return false;
    o.s_lo = s_lo; o.s_hi = s_hi; o.calc_crc = calc_crc; o.ref_low = ref_low; o.ref_high = ref_high; o.h = (uint8_t)fh; o.s = (uint8_t)fs;
    o.ctrl_block = ctrl_block_cells(s_lo, s_hi);
    return true;
}

/* What does not change while a batch of lines is decoded with the inherited tuning: the sampling positions of the two cells a
 * lane owns at shift stage 0 and the hysteresis-depth-0 levels.  fast_try0 is the first rung of fast_decode's ladder with these
 * hoisted; a line that does not pass it goes through fast_decode itself. */
struct FastPre { int32_t x0, x1; uint32_t ref_low, ref_high; bool ok; };
__device__ inline FastPre fast_pre(const FrameArgs &a, const Bin &b, Geo &g)
{
    FastPre p;
    const int lane = lane_id();
    const int32_t pixel_start = 0, pixel_stop = a.width - 1;
    if (!g.valid || g.start != b.in_coord.start || g.stop != b.in_coord.stop) {
        Line t; t.pixel_start = (uint16_t)pixel_start; t.pixel_stop = (uint16_t)pixel_stop;
        set_ppb(t, b.in_coord);
        g.start = b.in_coord.start; g.stop = b.in_coord.stop; g.psm = t.psm; g.valid = true;
        g.vp0 = bit_center(t, lane); g.vp1 = bit_center(t, lane + 64);
    }
    p.ref_low = get_low_level(b.in_ref, 0); p.ref_high = get_high_level(b.in_ref, 0);
    p.ok = !(p.ref_low <= b.in_black || p.ref_high >= b.in_white);
    int32_t x0 = g.vp0, x1 = g.vp1;
    p.x0 = x0 < pixel_start ? pixel_start : (x0 >= pixel_stop ? pixel_stop - 1 : x0);
    p.x1 = x1 < pixel_start ? pixel_start : (x1 >= pixel_stop ? pixel_stop - 1 : x1);
    return p;
}
/* the two cells a lane owns, sampled from the row staged at px + ofs; fast_try0 below takes them (split so that the LDS reads of
 * several rows can be issued before the first row is worked on) */
struct FastCells { uint8_t p0, p1; };
__device__ __forceinline__ FastCells fast_sample(const WaveLds &lds, const FastPre &p, int ofs) { FastCells c; c.p0 = lds.px[p.x0 + ofs]; c.p1 = lds.px[p.x1 + ofs]; return c; }
__device__ inline bool fast_try0(const FastCells &cells, const FastPre &p, const LaneConst &lc, FastBits &o)     /* needs p.ok */
{
    uint8_t p0 = cells.p0, p1 = cells.p1;
    uint64_t a_lo = __ballot(p0 > p.ref_low), b_lo = __ballot(p0 >= p.ref_high);
    uint64_t a_hi = __ballot(p1 > p.ref_low), b_hi = __ballot(p1 >= p.ref_high);
    uint64_t s_lo, s_hi;
    solve_automaton(a_lo, a_hi, b_lo, b_hi, s_lo, s_hi);
    int par = (__popcll(s_lo & lc.klo) + __popcll(s_hi & lc.khi)) & 1;
    uint16_t calc_crc = (uint16_t)((uint16_t)(__ballot(par) & 0xFFFF) ^ c_crc.init);
    if (calc_crc != rev16((uint32_t)((s_hi >> 48) & 0xFFFF))) return false;
    o.s_lo = s_lo; o.s_hi = s_hi; o.calc_crc = calc_crc; o.ref_low = (uint8_t)p.ref_low; o.ref_high = (uint8_t)p.ref_high; o.h = 0; o.s = 0;
    o.ctrl_block = false;                   /* not looked at here: the batch loop does ctrl_block_maybe() */
    return true;
}
/* fast_try0 for two rows at once, the two decode chains written step by step side by side: every step of row B is independent
 * of the same step of row A, so the in-order issue of a wave always has a second instruction that does not wait for the first */
#ifndef SDV_EMU
#define SDV_CARRY_ADD(y0, y1, y2, y3, x_lo, x_hi, u_lo, u_hi) \
    asm("s_add_u32 %0, %4, %8\n\ts_addc_u32 %1, %5, %9\n\ts_addc_u32 %2, %6, %10\n\ts_addc_u32 %3, %7, %11" \
        : "=&s"(y0), "=&s"(y1), "=&s"(y2), "=&s"(y3) \
        : "s"((uint32_t)(x_lo)), "s"((uint32_t)((x_lo) >> 32)), "s"((uint32_t)(x_hi)), "s"((uint32_t)((x_hi) >> 32)), \
          "s"((uint32_t)(u_lo)), "s"((uint32_t)((u_lo) >> 32)), "s"((uint32_t)(u_hi)), "s"((uint32_t)((u_hi) >> 32)) \
        : "scc")
#endif
__device__ inline void fast_try0_x2(const FastCells &ca, const FastCells &cb, const FastPre &p, const LaneConst &lc,
                                    FastBits &oa, FastBits &ob, bool &ok_a, bool &ok_b)
{
    const int lane = lane_id();
    const uint64_t le = ((1ull << lane) - 1ull) | (1ull << lane);       /* lanes up to and including this one: inclusive prefix counts in one popcount */
    const uint64_t aA_lo = __ballot(ca.p0 > p.ref_low), aB_lo = __ballot(cb.p0 > p.ref_low);
    const uint64_t bA_lo = __ballot(ca.p0 >= p.ref_high), bB_lo = __ballot(cb.p0 >= p.ref_high);
    const uint64_t aA_hi = __ballot(ca.p1 > p.ref_low), aB_hi = __ballot(cb.p1 > p.ref_low);
    const uint64_t bA_hi = __ballot(ca.p1 >= p.ref_high), bB_hi = __ballot(cb.p1 >= p.ref_high);
    const uint64_t eA_lo = ~(aA_lo ^ bA_lo), eB_lo = ~(aB_lo ^ bB_lo), eA_hi = ~(aA_hi ^ bA_hi), eB_hi = ~(aB_hi ^ bB_hi);
    const uint64_t tA_lo = aA_lo & ~bA_lo, tB_lo = aB_lo & ~bB_lo, tA_hi = aA_hi & ~bA_hi, tB_hi = aB_hi & ~bB_hi;
    const int cA_lo = __popcll(tA_lo & le), cB_lo = __popcll(tB_lo & le);
    const int cA_hi = __popcll(tA_lo) + __popcll(tA_hi & le);
    const int cB_hi = __popcll(tB_lo) + __popcll(tB_hi & le);
    const uint64_t ptA_lo = __ballot(cA_lo & 1), ptB_lo = __ballot(cB_lo & 1), ptA_hi = __ballot(cA_hi & 1), ptB_hi = __ballot(cB_hi & 1);
    const uint64_t uA_lo = ((aA_lo & bA_lo) ^ ptA_lo) & eA_lo, uB_lo = ((aB_lo & bB_lo) ^ ptB_lo) & eB_lo;
    const uint64_t uA_hi = ((aA_hi & bA_hi) ^ ptA_hi) & eA_hi, uB_hi = ((aB_hi & bB_hi) ^ ptB_hi) & eB_hi;
    const uint64_t xA_lo = uA_lo | ~eA_lo, xB_lo = uB_lo | ~eB_lo, xA_hi = uA_hi | ~eA_hi, xB_hi = uB_hi | ~eB_hi;
    uint64_t yA_lo, yA_hi, yB_lo, yB_hi;
#ifdef SDV_EMU
    yA_lo = xA_lo + uA_lo; yA_hi = xA_hi + uA_hi + ((yA_lo < xA_lo) ? 1ull : 0ull);
    yB_lo = xB_lo + uB_lo; yB_hi = xB_hi + uB_hi + ((yB_lo < xB_lo) ? 1ull : 0ull);
#else
    { uint32_t y0, y1, y2, y3; SDV_CARRY_ADD(y0, y1, y2, y3, xA_lo, xA_hi, uA_lo, uA_hi); yA_lo = ((uint64_t)y1 << 32) | y0; yA_hi = ((uint64_t)y3 << 32) | y2; }
    { uint32_t y0, y1, y2, y3; SDV_CARRY_ADD(y0, y1, y2, y3, xB_lo, xB_hi, uB_lo, uB_hi); yB_lo = ((uint64_t)y1 << 32) | y0; yB_hi = ((uint64_t)y3 << 32) | y2; }
#endif
    const uint64_t sA_lo = (((yA_lo ^ xA_lo) & ~eA_lo) | uA_lo) ^ ptA_lo, sB_lo = (((yB_lo ^ xB_lo) & ~eB_lo) | uB_lo) ^ ptB_lo;
    const uint64_t sA_hi = (((yA_hi ^ xA_hi) & ~eA_hi) | uA_hi) ^ ptA_hi, sB_hi = (((yB_hi ^ xB_hi) & ~eB_hi) | uB_hi) ^ ptB_hi;
    const int parA = (__popcll(sA_lo & lc.klo) + __popcll(sA_hi & lc.khi)) & 1, parB = (__popcll(sB_lo & lc.klo) + __popcll(sB_hi & lc.khi)) & 1;
    const uint16_t crcA = (uint16_t)((uint16_t)(__ballot(parA) & 0xFFFF) ^ c_crc.init), crcB = (uint16_t)((uint16_t)(__ballot(parB) & 0xFFFF) ^ c_crc.init);
    ok_a = crcA == rev16((uint32_t)((sA_hi >> 48) & 0xFFFF)); ok_b = crcB == rev16((uint32_t)((sB_hi >> 48) & 0xFFFF));
    oa.s_lo = sA_lo; oa.s_hi = sA_hi; oa.calc_crc = crcA; oa.ref_low = (uint8_t)p.ref_low; oa.ref_high = (uint8_t)p.ref_high; oa.h = 0; oa.s = 0; oa.ctrl_block = false;
    ob.s_lo = sB_lo; ob.s_hi = sB_hi; ob.calc_crc = crcB; ob.ref_low = (uint8_t)p.ref_low; ob.ref_high = (uint8_t)p.ref_high; ob.h = 0; ob.s = 0; ob.ctrl_block = false;
}
/* the first 32 cells of the Control Block pattern: necessary for a Control Block, and a false alarm only once in 2^32 lines */
__device__ __forceinline__ bool ctrl_block_maybe(uint64_t s_lo) { return (uint32_t)s_lo == (uint32_t)ctrl_cells_0_55(); }

__device__ inline void bits_to_words(uint64_t s_lo, uint64_t s_hi, uint16_t *w)
{
    w[0] = rev14((uint32_t)(s_lo & 0x3FFF));
    w[1] = rev14((uint32_t)((s_lo >> 14) & 0x3FFF));
    w[2] = rev14((uint32_t)((s_lo >> 28) & 0x3FFF));
    w[3] = rev14((uint32_t)((s_lo >> 42) & 0x3FFF));
    w[4] = rev14((uint32_t)(((s_lo >> 56) | (s_hi << 8)) & 0x3FFF));
    w[5] = rev14((uint32_t)((s_hi >> 6) & 0x3FFF));
    w[6] = rev14((uint32_t)((s_hi >> 20) & 0x3FFF));
    w[7] = rev14((uint32_t)((s_hi >> 34) & 0x3FFF));
}

/* one line through the fast path (decode already staged in LDS); false = not handled, nothing changed.
 * kMeasure (full kernel only): the line behind one that did not read.  The worker took the black and white presets back then
 * (videotodigital.cpp:1468-1521), so Binarizer::processLine measures the two levels from the line's own pixels (findBlackWhite) before it reads with
 * the reference level and the coordinates that are still preset (STG_INPUT_ALL, binarizer.cpp:774-931) - the fast path with one step in front.
 * ladder_failed: no (depth, stage) of the ladder read the line with the inherited tuning - the general path need not try them again. */
template <bool kMeasure>
__device__ inline bool fast_line(const FrameArgs &a, WaveLds &lds, V2D &v, Geo &g, const LaneConst &lc,
                                 uint32_t frame_no, uint16_t line_num, uint32_t *fv_keys, sdv_line_rec *rec, bool *ladder_failed, unsigned long long *bw_slot = nullptr,
                                 int *rung_out = nullptr /* the line was taken: its shift stage when it read at hysteresis depth 0, else -1 */)
{
    const sdv_bin_preset &ps = a.preset;
    Bin &b = v.bin;
    uint8_t black = b.in_black, white = b.in_white, found_mark_ed = MARK_ED_START; uint16_t found_sp_ed = 0;
    *ladder_failed = false;
    if (!kMeasure) { if (!fast_eligible(a, b)) return false; }
    else {
        if (ps.en_force_coords || !ps.en_good_no_marker || are_bw_levels_preset(b, ps) || !is_ref_level_preset(b, ps) || !coords_valid(b.in_coord)) return false;
        if (a.width < BITS_IN_LINE || (a.width - 1) < BITS_BETWEEN) return false;
        Bin tb = b; Line t;
        stc_clear(t);
        bin_line_geometry(tb, ps, a.width, a.doubled != 0);
        tb.was_bw_scanned = false;
        K1_T(t_m0);
        const bool bw_found = find_black_white(tb, ps, lds, t, bw_slot);
        K1_T(t_m1);
        K1_ADD(22, t_m0, t_m1); K1_ADD(23, 0ull, 1ull);
        if (!bw_found) return false;
        black = t.black; white = t.white; found_mark_ed = t.mark_ed; found_sp_ed = t.m_sp_ed;      /* (findSTC007BW leaves what it saw of the STOP marker in the line) */
        if (b.in_ref >= white || b.in_ref <= black) return false;
    }
    FastBits fb;
    K1_T(t_fd0);
    if (!fast_decode(a, lds, b, black, white, g, lc, fb)) { *ladder_failed = !kMeasure; return false; }
    K1_T(t_fd1);
    K1_ADD(6, t_fd0, t_fd1);
    if (fb.ctrl_block) return false;
    if (rung_out) *rung_out = fb.h == 0 ? (int)fb.s : -1;
    const int lane = lane_id();
    uint16_t w[9];
    bits_to_words(fb.s_lo, fb.s_hi, w);
    w[8] = fb.calc_crc;

    /* ---- VideoToDigital bookkeeping for a line with valid CRC (videotodigital.cpp:1155-1396, 1524-1633) ---- */
    K1_T(t_bk0);
    const bool even_line = (line_num % 2) == 0;
    const bool doubled = a.doubled != 0;
    bool forced_bad = false;
    if (v.field_state == FIELD_NEW) v.field_state = FIELD_UNSAFE;
    v.good_coords_in_field++;
    v.q_line_length = (uint16_t)a.width;
    if (a.check_line_copy) {
        if (v.field_state == FIELD_UNSAFE) {
            b.in_coord.doubled = doubled;                   /* setGoodParameters(): same levels, same pair */
            if (kMeasure) bin_set_bw_levels(b, ps, black, white);
            if (ps.en_first_line_dup) forced_bad = true;
        } else {
            uint8_t diff = words_diff_bit_count(w, v.last_words);
            Line t; t.m2 = a.m2_format != 0; for (int i = 0; i < 6; i++) t.words[i] = w[i];
            if (!stc_is_almost_silent(t) && diff <= (BITS_DATA / BIT_DIFF_THRES_DIV)) { forced_bad = true; if (!even_line) v.q_dup_odd++; else v.q_dup_even++; }
        }
    }
    K1_T(t_bk1);
    {
        uint32_t key = coords_key(b.in_coord.start, b.in_coord.stop);
        SDV_WAVE_SYNC();
        if (lane == 0) {
            if (v.n_last == COORD_HISTORY_DEPTH) for (int i = 0; i < COORD_HISTORY_DEPTH - 1; i++) lds.lv_keys[i] = lds.lv_keys[i + 1];
            lds.lv_keys[v.n_last == COORD_HISTORY_DEPTH ? COORD_HISTORY_DEPTH - 1 : v.n_last] = key;
        }
        if (v.n_last < COORD_HISTORY_DEPTH) v.n_last++;
        SDV_WAVE_SYNC();
        fv_keys[v.nfv++] = key;
        K1_T(t_bk2);
        K1_ADD(16, t_bk0, t_bk1); K1_ADD(17, t_bk1, t_bk2);
        if (a.coordinate_damper && (v.n_last > (COORD_HISTORY_DEPTH / 2))) {
            Coords target; coords_clear(target);
            uint32_t k;
            if (median_keys(lds.lv_keys, v.n_last, &k)) target = key_to_coords(k, false);
            if (!coords_valid(target)) target = v.frame_avg;
            if (coords_valid(target)) {
                int16_t ds = (int16_t)(b.in_coord.start - target.start), de = (int16_t)(b.in_coord.stop - target.stop);
                uint8_t in_delta = (uint8_t)(((uint8_t)(g.psm / 128u)) * 3);
                if (((int)ds <= -(int)in_delta) || ((int)ds >= (int)in_delta) || ((int)de <= -(int)in_delta) || ((int)de >= (int)in_delta)) forced_bad = true;
            }
        }
        K1_T(t_bk2b);
        K1_ADD(18, t_bk2, t_bk2b);
    }
    K1_T(t_bk3);
    if (!forced_bad) { b.in_coord.doubled = doubled; if (kMeasure) bin_set_bw_levels(b, ps, black, white); }           /* setGoodParameters(work_line) */
    else { if (!even_line) v.q_bad_odd++; else v.q_bad_even++; }
    v.field_state = FIELD_INIT;
    if (!even_line) { v.q_odd++; v.q_pcm_odd++; } else { v.q_even++; v.q_pcm_even++; }
    v.pcm_lines_in_field++;
    for (int i = 0; i < 8; i++) v.last_words[i] = w[i];
    v.line_in_field_cnt++;
    b.line_length = (uint16_t)a.width;

    if (lane == 0) {
        sdv_line_rec r;
        r.frame_number = frame_no; r.line_number = line_num;
        for (int i = 0; i < 9; i++) r.words[i] = w[i];
        r.calc_crc = fb.calc_crc;
        r.data_start = b.in_coord.start; r.data_stop = b.in_coord.stop;
        r.marker_start_bg_coord = 0; r.marker_start_ed_coord = 0; r.marker_stop_ed_coord = found_sp_ed;
        r.black_level = black; r.white_level = white; r.ref_low = fb.ref_low; r.ref_level = b.in_ref; r.ref_high = fb.ref_high;
        r.hysteresis_depth = fb.h; r.shift_stage = fb.s; r.service_type = SDV_SRV_NO;
        r.mark_st_stage = MARK_ST_START; r.mark_ed_stage = found_mark_ed;
        r.flags = (uint8_t)(SDV_LF_BY_EXT_TUNE | SDV_LF_BW_SET | (forced_bad ? SDV_LF_FORCED_BAD : SDV_LF_CRC_VALID) | (doubled ? SDV_LF_FROM_DOUBLED : 0));
        r.word_state = forced_bad ? 0 : (uint8_t)(SDV_WS_WORD_CRC | SDV_WS_WORD_VALID);
        *rec = r;
    }
    K1_T(t_bk4);
    K1_ADD(19, t_bk3, t_bk4);
    return true;
}

struct BatchLaneOut { uint64_t s_lo, s_hi; uint16_t crc; };
/* ---------------------------------------------------------------------------------------------
 * Whole-frame capture (round 2).  On a tape that plays every line of a frame is read with the tuning the frame inherits, so what a
 * line contributes is only the four comparison masks of its 128 bit cells.  The capture loop takes the rows of BOTH fields together -
 * rows 2k and 2k+1 are neighbours in memory, so the frame is streamed through once, front to back, and the 128-byte lines two rows
 * share are fetched once - and per line does nothing but the cell gather, four compares and parking the masks in "its" lane
 * (v_writelane).  Every 64 row pairs the lanes solve the hysteresis automaton and the CRC of their own line (capture_solve), the
 * field-0 lines go through the per-line bookkeeping at once (batch_finish), the field-1 lines wait in LDS until field 0 has ended.
 * A line that does not read on the first rung ends the capture: what was decoded up to it stays, the frame loop below takes over
 * from that line with the row-staging paths of round 1.
 * --------------------------------------------------------------------------------------------- */
#ifdef SDV_EMU
__device__ __forceinline__ uint32_t write_lane(uint32_t old, uint32_t val, int lane) { return lane_id() == lane ? val : old; }
__device__ __forceinline__ uint8_t stream_load_u8(const uint8_t *p) { return *p; }
#define SDV_ISSUE_AFTER8(oa, ob, m0, m1, m2, m3, m4, m5, m6, m7) ((void)0)
#else
/* a byte that is read once: the load does not allocate in the caches on its way */
__device__ __forceinline__ uint8_t stream_load_u8(const uint8_t *p) { return __builtin_nontemporal_load(p); }
/* the uniform offsets oa, ob become known only once the eight uniform masks are: loads addressed through them are issued after the
 * compares that produce the masks (no instruction is emitted) */
#define SDV_ISSUE_AFTER8(oa, ob, m0, m1, m2, m3, m4, m5, m6, m7) \
    asm volatile("" : "+s"(oa), "+s"(ob) : "s"(m0), "s"(m1), "s"(m2), "s"(m3), "s"(m4), "s"(m5), "s"(m6), "s"(m7))
/* clang has no builtin for v_writelane_b32; the LLVM intrinsic is reachable by its name */
extern "C" __device__ uint32_t sdv_llvm_writelane(uint32_t val, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");
__device__ __forceinline__ uint32_t write_lane(uint32_t old, uint32_t val, int lane) { return sdv_llvm_writelane(val, (uint32_t)lane, old); }
#endif
struct CaptureRaw { uint32_t a0, a1, a2, a3, b0, b1, b2, b3; };     /* of "my" line: A = px > low (cells 0..127), B = px >= high */
__device__ __forceinline__ void capture_park(CaptureRaw &r, uint64_t a_lo, uint64_t a_hi, uint64_t b_lo, uint64_t b_hi, int j)
{
    r.a0 = write_lane(r.a0, (uint32_t)a_lo, j); r.a1 = write_lane(r.a1, (uint32_t)(a_lo >> 32), j);
    r.a2 = write_lane(r.a2, (uint32_t)a_hi, j); r.a3 = write_lane(r.a3, (uint32_t)(a_hi >> 32), j);
    r.b0 = write_lane(r.b0, (uint32_t)b_lo, j); r.b1 = write_lane(r.b1, (uint32_t)(b_lo >> 32), j);
    r.b2 = write_lane(r.b2, (uint32_t)b_hi, j); r.b3 = write_lane(r.b3, (uint32_t)(b_hi >> 32), j);
}
/* solve_automaton + the CRC of fill_stc007 for one line per LANE (every lane its own masks): returns the cells and whether the line
 * reads (CRC as read == CRC calculated, and not the start of a Control Block) */
/* (kAnyPattern: the CRC alone - the caller looks at the Control Block pattern itself) */
template <bool kAnyPattern = false>
__device__ inline bool capture_solve(const CaptureRaw &r, BatchLaneOut &o)
{
    const uint64_t a_lo = (uint64_t)r.a0 | ((uint64_t)r.a1 << 32), a_hi = (uint64_t)r.a2 | ((uint64_t)r.a3 << 32);
    const uint64_t b_lo = (uint64_t)r.b0 | ((uint64_t)r.b1 << 32), b_hi = (uint64_t)r.b2 | ((uint64_t)r.b3 << 32);
    const uint64_t e_lo = ~(a_lo ^ b_lo), e_hi = ~(a_hi ^ b_hi), t_lo = a_lo & ~b_lo, t_hi = a_hi & ~b_hi;
    const uint64_t pt_lo = prefix_xor64(t_lo), pt_hi = prefix_xor64(t_hi) ^ ((__popcll(t_lo) & 1) ? ~0ull : 0ull);
    const uint64_t u_lo = ((a_lo & b_lo) ^ pt_lo) & e_lo, u_hi = ((a_hi & b_hi) ^ pt_hi) & e_hi;
    const uint64_t x_lo = u_lo | ~e_lo, x_hi = u_hi | ~e_hi;
    const uint64_t y_lo = x_lo + u_lo, y_hi = x_hi + u_hi + ((y_lo < x_lo) ? 1ull : 0ull);
    const uint64_t s_lo = ((((y_lo ^ x_lo) & ~e_lo) | u_lo) ^ pt_lo), s_hi = ((((y_hi ^ x_hi) & ~e_hi) | u_hi) ^ pt_hi);
    uint32_t crc = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) crc |= (uint32_t)((__popcll(s_lo & c_crc.klo[j]) + __popcll(s_hi & c_crc.khi[j])) & 1) << j;
    crc ^= c_crc.init;
    o.s_lo = s_lo; o.s_hi = s_hi; o.crc = (uint16_t)crc;
    return (uint16_t)crc == rev16((uint32_t)((s_hi >> 48) & 0xFFFF)) && (kAnyPattern || !ctrl_block_maybe(s_lo));
}

/* ... as a call (general build: its frame loop has no registers to spare for the solve's 64-bit arithmetic; the masks go in and the cells come out in registers) */
struct SolveOut { uint32_t d0, d1, d2, d3, crc_reads; };          /* crc_reads: the CRC calculated | (it is the one read) << 16 */
#ifndef SDV_EMU
__device__ __attribute__((noinline)) SolveOut capture_solve_call(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
#else
__device__ inline SolveOut capture_solve_call(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3)
#endif
{
    CaptureRaw c; c.a0 = a0; c.a1 = a1; c.a2 = a2; c.a3 = a3; c.b0 = b0; c.b1 = b1; c.b2 = b2; c.b3 = b3;
    BatchLaneOut o;
    const bool reads = capture_solve<true>(c, o);
    SolveOut r; r.d0 = (uint32_t)o.s_lo; r.d1 = (uint32_t)(o.s_lo >> 32); r.d2 = (uint32_t)o.s_hi; r.d3 = (uint32_t)(o.s_hi >> 32);
    r.crc_reads = (uint32_t)o.crc | (reads ? 0x10000u : 0u);
    return r;
}

/* ---------------------------------------------------------------------------------------------
 * Line-batch fast path.  In the steady state the tuning does not change from line to line, so up to
 * 64 consecutive lines of a field are decoded one after another by the whole wave (phase A), each
 * line's 128 raw cells + CRC parked in "its" lane; then the per-line VideoToDigital bookkeeping -
 * word packing, duplicate-line test against the previous line, flags, the 48-byte record - runs for
 * all of them at once, one lane per line (phase B).  Preconditions (checked per batch) make the
 * bookkeeping of line k independent of lines < k except through the previous line's words:
 *   - the STG_INPUT_ALL conditions hold (fast_eligible);
 *   - last_valid_coord_list is full and holds only the preset pair, so pushing it again changes
 *     nothing and the coordinate damper sees a zero delta (videotodigital.cpp:1295-1363).
 * The first line that does not decode this way ends the batch and goes through the sequential path.
 * --------------------------------------------------------------------------------------------- */
struct BatchLane { uint32_t d0, d1, d2, d3, meta; };      /* raw cells + (calc_crc | h<<16 | s<<20) of "my" line */

__device__ inline bool batch_eligible(const FrameArgs &a, const WaveLds &lds, const V2D &v, const Geo &g)
{
    if (!fast_eligible(a, v.bin)) return false;
    if (v.n_last != COORD_HISTORY_DEPTH) return false;
    uint32_t key = coords_key(v.bin.in_coord.start, v.bin.in_coord.stop);
    bool differs = (lane_id() < COORD_HISTORY_DEPTH) && (lds.lv_keys[lane_id() < COORD_HISTORY_DEPTH ? lane_id() : 0] != key);
    if (__ballot(differs) != 0ull) return false;
    if (a.coordinate_damper) {
        /* in_delta = getPPB()*3 as uint8_t must be > 0 or the damper would flag a zero delta */
        uint32_t psm = (uint32_t)((int)v.bin.in_coord.stop - (int)v.bin.in_coord.start);
        psm = (psm * 128u + BITS_BETWEEN / 2) / BITS_BETWEEN;
        if ((uint8_t)(((uint8_t)(psm / 128u)) * 3) == 0) return false;
    }
    return true;
}

/* Where phase B stages the records of a batch on their way out: 64 slots of 16 bytes in the per-wave LDS, behind the lines the
 * whole-frame capture parks (five chunks of 64 lines x four words) and inside the room of the general path's sweep table, which no
 * path holds anything in across a batch. */
enum { REC_STAGE_OFS = 5120, REC_STAGE_BYTES = 1024 };
static_assert(REC_STAGE_OFS >= ((SDV_MAX_HEIGHT / 2 + 63) / 64) * 4 * 64 * sizeof(uint32_t), "the parked lines of a field of SDV_MAX_HEIGHT / 2 lines end in front of the record stage");
static_assert(REC_STAGE_OFS >= SDV_PX_BYTES && REC_STAGE_OFS + REC_STAGE_BYTES <= offsetof(WaveLds, crc_stats), "the record stage lies behind the staged scanlines and in front of the histories");
#ifndef SDV_COALESCED_RECORDS
#define SDV_COALESCED_RECORDS 1     /* 0: every lane stores its own record (three 16-byte stores at a pitch of 48 bytes) */
#endif

/* phase B for `n` (1..64) decoded lines: lane k owns line k of the batch.  write_keys: the lines' coordinate keys go to the frame's list (the
 * whole-frame capture leaves them out: while it lasts every line has the one pair the frame was started with, and the list is filled in
 * only when the capture ends early) */
__device__ inline void batch_finish(const FrameArgs &a, WaveLds &lds, V2D &v, const BatchLane &bl, int n, uint32_t frame_no, uint16_t first_line_num,
                                    uint32_t *fv_keys, sdv_line_rec *rec, bool write_keys = true, uint4 *direct_dst = nullptr, int *n_forced = nullptr)
{
    const sdv_bin_preset &ps = a.preset;
    Bin &b = v.bin;
    const int lane = lane_id();
    const bool active = lane < n;
    const bool doubled = a.doubled != 0;
    const bool even_line = (first_line_num % 2) == 0;           /* the same for every line of a field */
    uint64_t s_lo = (uint64_t)bl.d0 | ((uint64_t)bl.d1 << 32), s_hi = (uint64_t)bl.d2 | ((uint64_t)bl.d3 << 32);
    uint16_t w[8];
    bits_to_words(s_lo, s_hi, w);
    uint32_t p01 = (uint32_t)w[0] | ((uint32_t)w[1] << 16), p23 = (uint32_t)w[2] | ((uint32_t)w[3] << 16);
    uint32_t p45 = (uint32_t)w[4] | ((uint32_t)w[5] << 16), p67 = (uint32_t)w[6] | ((uint32_t)w[7] << 16);
    /* words of the previous PCM line: the lane below, or the line before the batch */
    int src = lane > 0 ? lane - 1 : 0;
    uint32_t q01 = (uint32_t)__shfl((int)p01, src), q23 = (uint32_t)__shfl((int)p23, src);
    uint32_t q45 = (uint32_t)__shfl((int)p45, src), q67 = (uint32_t)__shfl((int)p67, src);
    if (lane == 0) {
        q01 = (uint32_t)v.last_words[0] | ((uint32_t)v.last_words[1] << 16); q23 = (uint32_t)v.last_words[2] | ((uint32_t)v.last_words[3] << 16);
        q45 = (uint32_t)v.last_words[4] | ((uint32_t)v.last_words[5] << 16); q67 = (uint32_t)v.last_words[6] | ((uint32_t)v.last_words[7] << 16);
    }
    /* getWordsDiffBitCount: XOR truncated to uint8_t per word (stc007line.cpp:329-357) */
    uint32_t m = 0x00FF00FFu;
    int diff = __popc((p01 ^ q01) & m) + __popc((p23 ^ q23) & m) + __popc((p45 ^ q45) & m) + __popc((p67 ^ q67) & m);
    Line t; t.m2 = a.m2_format != 0; for (int i = 0; i < 6; i++) t.words[i] = w[i];
    bool silent = stc_is_almost_silent(t);
    uint8_t fs0 = v.field_state;
    if (fs0 == FIELD_NEW) fs0 = FIELD_UNSAFE;
    bool first_unsafe = (lane == 0) && (fs0 == FIELD_UNSAFE);
    bool forced_bad = false, dup = false;
    if (a.check_line_copy) {
        if (first_unsafe) forced_bad = ps.en_first_line_dup != 0;
        else { dup = !silent && diff <= (BITS_DATA / BIT_DIFF_THRES_DIV); forced_bad = dup; }
    }
    uint64_t dup_m = __ballot(active && dup), bad_m = __ballot(active && forced_bad);
    uint64_t setgood_m = __ballot(active && ((a.check_line_copy && first_unsafe) || !forced_bad));
    /* the record (sdv_line_rec, 48 bytes) as its twelve dwords */
    const uint32_t h = (bl.meta >> 16) & 0xF, sft = (bl.meta >> 20) & 0xF, crc = bl.meta & 0xFFFF;
    const uint32_t lf = SDV_LF_BY_EXT_TUNE | SDV_LF_BW_SET | (forced_bad ? SDV_LF_FORCED_BAD : SDV_LF_CRC_VALID) | (doubled ? SDV_LF_FROM_DOUBLED : 0);
    const uint32_t wst = forced_bad ? 0u : (uint32_t)(SDV_WS_WORD_CRC | SDV_WS_WORD_VALID);
    uint4 g0, g1, g2;
    g0.x = frame_no;
    g0.y = (uint32_t)(uint16_t)(first_line_num + 2 * lane) | (p01 << 16);
    g0.z = (p01 >> 16) | (p23 << 16);
    g0.w = (p23 >> 16) | (p45 << 16);
    g1.x = (p45 >> 16) | (p67 << 16);
    g1.y = (p67 >> 16) | (crc << 16);                            /* words[7], words[8] = the CRC as read (= as calculated: the line read) */
    g1.z = crc | ((uint32_t)(uint16_t)b.in_coord.start << 16);   /* calc_crc, data_start */
    g1.w = (uint32_t)(uint16_t)b.in_coord.stop;                  /* data_stop, marker_start_bg_coord = 0 */
    g2.x = 0;                                                    /* marker_start_ed_coord, marker_stop_ed_coord */
    g2.y = (uint32_t)b.in_black | ((uint32_t)b.in_white << 8) | ((uint32_t)get_low_level(b.in_ref, (uint8_t)h) << 16) | ((uint32_t)b.in_ref << 24);
    g2.z = (uint32_t)get_high_level(b.in_ref, (uint8_t)h) | (h << 8) | (sft << 16) | ((uint32_t)SDV_SRV_NO << 24);
    g2.w = (uint32_t)MARK_ST_START | ((uint32_t)MARK_ED_START << 8) | (lf << 16) | (wst << 24);
    static_assert(sizeof(sdv_line_rec) == 48 && offsetof(sdv_line_rec, words) == 6 && offsetof(sdv_line_rec, calc_crc) == 24 && offsetof(sdv_line_rec, data_stop) == 28 &&
                  offsetof(sdv_line_rec, black_level) == 36 && offsetof(sdv_line_rec, ref_high) == 40 && offsetof(sdv_line_rec, mark_st_stage) == 44 && offsetof(sdv_line_rec, word_state) == 47,
                  "the record is put together dword by dword");
    if (n_forced) *n_forced = __popcll(bad_m);
    if (direct_dst) {
        /* the line as the stitcher keeps it (SLine, 32 bytes: the record's first six dwords, then calc_crc | word_crc mask, word_valid mask | flags |
         * reference level - sline_from_raw, stc007_stitch_device.h), straight into the field buffer: two pieces of 16 bytes per line through the stage */
        const uint32_t wm = forced_bad ? 0u : 0x1FFu, slf = (forced_bad ? (uint32_t)DSL_FORCED_BAD : 0u) | (uint32_t)DSL_COORDS_VALID | (uint32_t)DSL_BW_SET;
        uint4 h1;
        h1.x = g1.x; h1.y = g1.y; h1.z = crc | (wm << 16); h1.w = wm | (slf << 16) | ((uint32_t)b.in_ref << 24);
        uint4 *const stage = (uint4 *)((uint8_t *)&lds + REC_STAGE_OFS);
        const int pieces = 2 * n;
        for (int c = 0; 64 * c < pieces; c++) {
            SDV_WAVE_SYNC();
            const int p0 = 2 * lane - 64 * c;
            if (active && (unsigned)p0 < 64u) { stage[p0] = g0; stage[p0 + 1] = h1; }
            SDV_WAVE_SYNC();
#if SDV_NT_RECORDS && !defined(SDV_EMU)
            if (64 * c + lane < pieces) { const uint4 q = stage[lane]; typedef uint32_t v4u __attribute__((ext_vector_type(4))); v4u t; t.x = q.x; t.y = q.y; t.z = q.z; t.w = q.w; __builtin_nontemporal_store(t, (v4u *)(direct_dst + 64 * c + lane)); }
#else
            if (64 * c + lane < pieces) direct_dst[64 * c + lane] = stage[lane];
#endif
        }
    } else {
    uint4 *const dst = (uint4 *)rec;
#if SDV_NT_RECORDS && !defined(SDV_EMU)
    /* the records are written once and read by a later kernel: streaming stores */
    auto put = [](const uint4 &val, uint4 *where) { typedef unsigned int nt_u32x4 __attribute__((ext_vector_type(4))); nt_u32x4 t = { val.x, val.y, val.z, val.w }; __builtin_nontemporal_store(t, (nt_u32x4 *)where); };
#else
    auto put = [](const uint4 &val, uint4 *where) { *where = val; };
#endif
#if SDV_COALESCED_RECORDS
    {   /* The n records are 3 n pieces of 16 bytes; piece p lies in lane p / 3.  They change lanes through the stage, 64 pieces at a time, so that
         * a store instruction covers 1 KB of consecutive addresses (lane i writes piece 64 c + i) - stored from the lanes that made them a store
         * covers every third piece of 3 KB, and the three stores of a record reach the memory side as three partial lines. */
        uint4 *const stage = (uint4 *)((uint8_t *)&lds + REC_STAGE_OFS);
        const int pieces = 3 * n;
        for (int c = 0; 64 * c < pieces; c++) {
            SDV_WAVE_SYNC();
            const int p0 = 3 * lane - 64 * c;           /* my first piece, counted from the start of this pass */
            if (active) {
                if ((unsigned)p0 < 64u) stage[p0] = g0;
                if ((unsigned)(p0 + 1) < 64u) stage[p0 + 1] = g1;
                if ((unsigned)(p0 + 2) < 64u) stage[p0 + 2] = g2;
            }
            SDV_WAVE_SYNC();
            if (64 * c + lane < pieces) put(stage[lane], dst + 64 * c + lane);
        }
    }
#else
    if (active) { put(g0, dst + 3 * lane); put(g1, dst + 3 * lane + 1); put(g2, dst + 3 * lane + 2); }
#endif
    }
    if (active && write_keys) fv_keys[v.nfv + lane] = coords_key(b.in_coord.start, b.in_coord.stop);
    /* wave-uniform state after the n lines */
    int nd = __popcll(dup_m), nb = __popcll(bad_m);
    v.nfv += n;
    v.good_coords_in_field = (uint16_t)(v.good_coords_in_field + n);
    v.pcm_lines_in_field = (uint16_t)(v.pcm_lines_in_field + n);
    v.line_in_field_cnt = (uint16_t)(v.line_in_field_cnt + n);
    v.q_line_length = (uint16_t)a.width;
    if (!even_line) { v.q_odd = (uint16_t)(v.q_odd + n); v.q_pcm_odd = (uint16_t)(v.q_pcm_odd + n); v.q_dup_odd = (uint16_t)(v.q_dup_odd + nd); v.q_bad_odd = (uint16_t)(v.q_bad_odd + nb); }
    else { v.q_even = (uint16_t)(v.q_even + n); v.q_pcm_even = (uint16_t)(v.q_pcm_even + n); v.q_dup_even = (uint16_t)(v.q_dup_even + nd); v.q_bad_even = (uint16_t)(v.q_bad_even + nb); }
    if (setgood_m != 0ull) b.in_coord.doubled = doubled;
    v.field_state = FIELD_INIT;
    b.line_length = (uint16_t)a.width;
    int last = n - 1;
    uint32_t l01 = (uint32_t)__shfl((int)p01, last), l23 = (uint32_t)__shfl((int)p23, last), l45 = (uint32_t)__shfl((int)p45, last), l67 = (uint32_t)__shfl((int)p67, last);
    l01 = uniu(l01); l23 = uniu(l23); l45 = uniu(l45); l67 = uniu(l67);
    v.last_words[0] = (uint16_t)(l01 & 0xFFFF); v.last_words[1] = (uint16_t)(l01 >> 16); v.last_words[2] = (uint16_t)(l23 & 0xFFFF); v.last_words[3] = (uint16_t)(l23 >> 16);
    v.last_words[4] = (uint16_t)(l45 & 0xFFFF); v.last_words[5] = (uint16_t)(l45 >> 16); v.last_words[6] = (uint16_t)(l67 & 0xFFFF); v.last_words[7] = (uint16_t)(l67 >> 16);
}

/* General path for one regular line, out of line so that its register appetite (reference sweep, marker
 * searches) does not spill into the hot loop.  State crosses the call in memory. */
struct SlowCtx { FrameArgs a; V2D v; Line wl; SweepHook hook; bool line_only; /* (the per-line kernel: the line as processLine leaves it, none of the worker's bookkeeping) */ };

template <int kWho>
__device__ __attribute__((noinline)) void slow_line(SlowCtx *c, WaveLds *lds, uint32_t frame_no, uint16_t line_num,
                                                    uint32_t *fv_keys, uint32_t *fi_keys, sdv_line_rec *rec)
{
    K1_T(t0_);
    bin_set_mode(c->v.bin, c->a.mode);
    process_line<kWho>(c->v.bin, c->a.preset, *lds, c->wl, frame_no, line_num, c->a.width, c->a.doubled != 0, c->hook);
    K1_T(t1_);
    if (!c->line_only) v2d_post_line(c->v, c->a, *lds, c->wl, fv_keys, fi_keys, (line_num % 2) == 0);
    emit_record(c->wl, rec);
    K1_T(t2_);
    K1_ADD(8, t0_, t2_); K1_ADD(13, t1_, t2_); K1_ADD(15, 0ull, 1ull);
}

/* A line of a dropped frame: Binarizer::processLine answers an empty VideoLine with a cleared line - silent words, CRC invalid
 * (binarizer.cpp:569-570, :1689-1700) - and the worker books it like any line that did not read (the line's length counts as 0:
 * VideoLine::setEmpty drops the pixels, videoline.cpp:71-75). */
__device__ __attribute__((noinline)) void empty_line(SlowCtx *c, WaveLds *lds, uint32_t frame_no, uint16_t line_num,
                                                     uint32_t *fv_keys, uint32_t *fi_keys, sdv_line_rec *rec)
{
    stc_clear(c->wl);
    c->wl.frame_number = frame_no; c->wl.line_number = line_num;
    const uint16_t ql = c->v.q_line_length;
    v2d_post_line(c->v, c->a, *lds, c->wl, fv_keys, fi_keys, (line_num % 2) == 0);
    c->v.q_line_length = ql;
    emit_record(c->wl, rec);
}

/* kLean: the build of the frame loop without the general path.  The general path (reference sweep, marker searches at 24
 * hysteresis levels) is rare on a tape that plays, but as a callee its registers count for the whole kernel and hold the
 * occupancy at 3 waves per SIMD; without it the loop fits 5.  A lean wave that meets a line it cannot take through the fast
 * paths gives the frame up: it marks its outgoing state (sdv_v2d_state::_pad[0]) and the engine decodes from that frame on
 * with the full kernel. */
enum { STATE_ABORTED = 0xA5 };       /* (v2d_give_up) */
/* sc (full kernel): the context of the general path, in LDS - one copy for the wave.  (It lived on the stack of the frame loop before: scratch memory,
 * every field the general path read of it a trip to global memory, a few dozen of them one behind the other per line: C3 PAL tape 28.4 -> 25.6 ms.)
 * The general path is wave-uniform code: every lane computes the same values and stores them to the same place, in lockstep.  The emulator runs the
 * lanes as threads of their own between collectives, where a shared read-modify-write would be seen half done: it keeps a copy per lane. */
#ifdef SDV_EMU
#define SDV_SLOW_CTX(c) SlowCtx c; sc_args_set = false
#else
#define SDV_SLOW_CTX(c) SlowCtx &c = *sc
#endif
/* kMeet (general build): with the trajectory snapshots ("a pass that meets the last one", TcSnap) and the batches of a tape on a later shift stage.  Both cost
 * the general build registers it does not have (168 with 45 of them spilled, against 154 and none): a sixth of its speed on a tape where neither helps - damage in
 * every few dozen lines that re-tunes the binarizer for good each time (4 % of the decodes meet their last pass).  The engine picks the build by
 * what the last call showed (engine.inc, plain_general). */
template <bool kLean, bool kMeet = true, bool kFat = false>
__device__ inline void frame_body(const FrameArgs &a, WaveLds &lds, int f, SlowCtx *sc = nullptr, FatLds *fat = nullptr, SweepEnt *fat_levels = nullptr)
{
    bool sc_args_set = false;           /* sc->a = a: once per frame */
    /* every decode of a frame says anew where its lines went (the last one counts): this build at the end of the frame, with everything else it writes - a
     * store at the head of the frame is a write in the middle of the other frames' reads; the general build, which decodes the frames this one gave up, here */
    if (!kLean && a.direct_frames && lane_id() == 0) a.direct_frames[f].flag = 0;
    DirectFrame dsum; dsum.frame_number = 0; dsum.n[0] = dsum.n[1] = dsum.bad[0] = dsum.bad[1] = 0; dsum.ref = 0; dsum.flag = 0; dsum._pad[0] = dsum._pad[1] = 0;
    K1_BEGIN();
    K1_T(t_begin);
    V2D v; Line wl;
    v2d_stage_state_in(a, lds, f);
    if (a.skip && uniu(a.skip[f])) { v2d_relink(a, lds, f); return; }
    v2d_load_state(v, lds, v2d_state_in(lds), a);
    const uint32_t frame_no = a.first_frame_no + (uint32_t)f;
    const uint8_t *frame = a.luma + (size_t)f * a.frame_stride;
    uint32_t *fv_keys = a.scratch + (size_t)f * 2u * (size_t)a.height;
    uint32_t *fi_keys = fv_keys + a.height;
    size_t rec_base = (size_t)f * (size_t)(a.height + 3);
    if (a.new_file_frame >= 0 && f > a.new_file_frame) rec_base += 1;
    sdv_line_rec *rec = a.recs + rec_base;
    Geo geo; geo.valid = false; geo.start = geo.stop = 0; geo.psm = 0; geo.vp0 = geo.vp1 = 0;
    LaneConst lc;
    lc.klo = (lane_id() < 16) ? c_crc.klo[lane_id() & 15] : 0ull;
    lc.khi = (lane_id() < 16) ? c_crc.khi[lane_id() & 15] : 0ull;
    const int lane = lane_id();
    /* a pass that meets the last one (TcSnap, above): general kernel only.  (A frame this kernel has decoded is not given to the lean kernel again within the
     * call - engine.inc, hard[] - so nothing else writes the frame's records between two passes of this kernel.) */
    bool tc_on = false, tc_met = false; uint32_t tc_prev = 0, tc_prev_n = 0; int tc_w = 0, tc_wbuf = 0;
    TcSnap *tc_wr = nullptr; const TcSnap *tc_rd = nullptr; const uint32_t *tc_rd_keys = nullptr;
    uint32_t tc_my_pos = TC_POS_NONE;       /* lane i: the line entry i of the last pass stands ahead of */

    v2d_begin_frame(v, a, lds);
    if (frame_is_empty(a, f)) {
        if (kLean) {            /* to the full kernel: the bookkeeping of lines that do not read lives there */
            v2d_give_up(a, lds, f);
            return;
        } else {
            uint16_t ln = 0;
            if (f == a.new_file_frame) { v2d_service_line(v, a, lds, wl, frame_no, 0, SDV_SRV_NEW_FILE); emit_record(wl, rec++); }
            for (int field = 0; field < 2; field++) {
                const int nl = field == 0 ? (a.height + 1) / 2 : a.height / 2;
                for (int i = 0; i < nl; i++) {
                    ln = (uint16_t)(field + 1 + 2 * i);
                    SDV_SLOW_CTX(c);
                    SDV_WAVE_SYNC();
                    if (!sc_args_set) { c.a = a; sc_args_set = true; }
                    c.v = v;
                    SDV_WAVE_SYNC();
                    empty_line(&c, &lds, frame_no, ln, fv_keys, fi_keys, rec++);
                    SDV_WAVE_SYNC();
                    v = c.v;
                    v2d_make_uniform(v);
                }
                ln = (uint16_t)(ln + 2);
                v2d_service_line(v, a, lds, wl, frame_no, ln, SDV_SRV_END_FIELD);
                emit_record(wl, rec++);
            }
            ln = (uint16_t)(ln + 2);
            v2d_service_line(v, a, lds, wl, frame_no, ln, SDV_SRV_END_FRAME);
            v2d_end_frame(v, a, lds, frame_no, fv_keys, fi_keys, &a.stats[f]);
            emit_record(wl, rec++);
            v2d_store_state(v, lds, &a.states_out[f], a);
            return;
        }
    }
    if (f == a.end_file_frame) {
        /* VideoInFFMPEG::insertDummyFrame(true, false) (vin_ffmpeg.cpp:367-523) through the worker: FILLER lines in field order,
         * END_FIELD after each field, END_FILE, END_FRAME */
        uint16_t ln = 0;
        for (int field = 0; field < 2; field++) {
            const int nl = field == 0 ? (a.height + 1) / 2 : a.height / 2;
            for (int i = 0; i < nl; i++) {
                ln = (uint16_t)(field + 1 + 2 * i);
                v2d_service_line(v, a, lds, wl, frame_no, ln, SDV_SRV_FILLER);
                emit_record(wl, rec++);
            }
            ln = (uint16_t)(ln + 2);
            v2d_service_line(v, a, lds, wl, frame_no, ln, SDV_SRV_END_FIELD);
            emit_record(wl, rec++);
        }
        ln = (uint16_t)(ln + 2);
        v2d_service_line(v, a, lds, wl, frame_no, ln, SDV_SRV_END_FILE);
        emit_record(wl, rec++);
        ln = (uint16_t)(ln + 2);
        v2d_service_line(v, a, lds, wl, frame_no, ln, SDV_SRV_END_FRAME);
        v2d_end_frame(v, a, lds, frame_no, fv_keys, fi_keys, &a.stats[f]);
        emit_record(wl, rec++);
        v2d_store_state(v, lds, &a.states_out[f], a);
        return;
    }
    if (f == a.new_file_frame) { v2d_service_line(v, a, lds, wl, frame_no, 0, SDV_SRV_NEW_FILE); emit_record(wl, rec++); }
    if (kFat && a.tc_hdr && lane == 0) a.tc_hdr[2 * f] = 0;     /* (a pass of the build without snapshots: what the last pass with them left no longer describes the frame's records) */
    if (!kLean && kMeet && a.tc_hdr) {
        tc_on = true;
        tc_prev = uniu(a.tc_hdr[2 * f]); tc_prev_n = uniu(a.tc_hdr[2 * f + 1]);
        if (tc_prev > 2u || tc_prev_n > (uint32_t)TC_FINAL) tc_prev = 0;
        tc_wbuf = tc_prev == 1u ? 1 : 0;         /* this pass writes the other set: the last pass's is read while this one goes */
        tc_wr = a.tc_snaps + ((size_t)f * 2u + (size_t)tc_wbuf) * TC_ENTRIES;
        if (tc_prev) {
            tc_rd = a.tc_snaps + ((size_t)f * 2u + (size_t)(tc_prev - 1u)) * TC_ENTRIES;
            tc_my_pos = lane < (int)tc_prev_n ? (tc_rd[lane].d[18] & 0xFFFFu) : (uint32_t)TC_POS_NONE;
            tc_rd_keys = (tc_prev == 2u ? a.tc_keys : a.scratch) + (size_t)f * 2u * (size_t)a.height;
        }
        if (tc_wbuf) { fv_keys = a.tc_keys + (size_t)f * 2u * (size_t)a.height; fi_keys = fv_keys + a.height; }
        SDV_WAVE_SYNC();
        if (lane == 0) a.tc_hdr[2 * f] = 0;     /* what this pass leaves counts only once it has ended complete (below) */
    }
    /* decode order of VideoInFFMPEG::spliceFrame (vin_ffmpeg.cpp:281-347): field 0 = rows 0,2,4.., field 1 = rows 1,3,5.. */
    const int n_field[2] = { (a.height + 1) / 2, a.height / 2 };
    RowPrefetch pf;
    pf.v0 = uint4{0, 0, 0, 0}; pf.v1 = uint4{0, 0, 0, 0};
    for (int u = 0; u < SDV_ROWQ; u++) { pf.vq[u] = uint4{0, 0, 0, 0}; pf.rowq[u] = frame; }
    pf.nq = 0;
    pf.vec = ((((uintptr_t)frame) | (uintptr_t)a.row_stride) & 15) == 0 && a.width >= 16;
    { int nvec = a.width >> 4; pf.i0 = lane < nvec ? lane : (nvec > 0 ? nvec - 1 : 0); }
    row_prefetch(pf, frame, a.width);
    const size_t row_step = 2 * a.row_stride;                  /* the next row of a field */
    uint16_t line_num = 0;
    int start_field = 0, start_idx = 0;
    bool all_captured = false; uint32_t captured_key = 0;
    bool frame_direct = false;              /* the frame's lines went straight into the stitcher's field buffers (no records) */
    bool sweep_pending = false;             /* a line of this frame went on without the sweep it asked for: the frame is decoded again */
    bool used_general = false;              /* a line of this frame took the general path */
#ifdef SDV_K1_STAMPS
    int n_slow_lines = 0;
#endif
    /* whole-frame capture (see capture_solve): only in the lean build, on the geometry the batch loop takes */
    if (kLean && pf.vec && a.width <= 1024 && (a.width & 15) == 0 && n_field[1] > 0 && (uint64_t)a.row_stride * (uint64_t)a.height < (1ull << 31) &&
        batch_eligible(a, lds, v, geo)) {
        const FastPre pre = fast_pre(a, v.bin, geo);
        if (pre.ok) {
            K1_T(t_cap);
            const int n0 = n_field[0], n1 = n_field[1];
            const int n_chunks = (n0 + 63) / 64;
            uint32_t *park = (uint32_t *)lds.px;                /* field-1 lines wait here: [chunk][4 words][lane], over px + hist + the front of sweep (up to REC_STAGE_OFS) */
            uint64_t ok0_packed = 0, ok1_packed = 0;            /* lines of field 0 / 1 that read, per chunk (7 bits each) */
            /* Field 0's lines wait too, in registers (the cells of chunk c in hold[c], picked by compares: the chunk loop stays a loop): every record of
             * the frame is written when the capture has ended, in one burst.  (Written chunk by chunk - eight bursts of 3 KB spread over the frame - the
             * same bytes cost more: the memory side turns from reading to writing and back for each of them; tools/probe_capture.hip, variant 5.) */
            enum { HOLD_CHUNKS = (SDV_MAX_HEIGHT / 2 + 63) / 64 };
            uint32_t hold[HOLD_CHUNKS][4];
#pragma unroll
            for (int u = 0; u < HOLD_CHUNKS; u++) hold[u][0] = hold[u][1] = hold[u][2] = hold[u][3] = 0;
            int n_done = 0;                                     /* chunks captured */
            const uint32_t rs = (uint32_t)a.row_stride;         /* offsets inside a frame fit 32 bits (checked above) */
            const uint32_t lo = pre.ref_low, hi = pre.ref_high;
            const uint32_t x0 = (uint32_t)pre.x0, x1 = (uint32_t)pre.x1;
            constexpr int D = SDV_CAPTURE_D;                    /* row pairs in flight */
            static_assert(64 % D == 0, "the prefetch queue must stay aligned with the 64-line chunks");
            /* rows 2k, 2k+1 of the frame; the luma is read once and never again: non-temporal, so that it does not push the
             * record buffer out of the last-level cache (profiles/r02_probe_capture_ceiling.txt) */
            auto pair_offsets = [&](int kn, uint32_t &oa, uint32_t &ob) {
                const uint32_t k = (uint32_t)(kn < n0 ? kn : n0 - 1);
                oa = 2u * k * rs; ob = (int)k < n1 ? oa + rs : oa; };
            uint8_t q[D][4];
#pragma unroll
            for (int d = 0; d < D; d++) {
                uint32_t oa, ob; pair_offsets(d, oa, ob);
                q[d][0] = stream_load_u8(frame + (oa + x0)); q[d][1] = stream_load_u8(frame + (oa + x1));
                q[d][2] = stream_load_u8(frame + (ob + x0)); q[d][3] = stream_load_u8(frame + (ob + x1));
            }
            bool whole = true;
            for (int c = 0; c < n_chunks && whole; c++) {
                const int cn0 = n0 - 64 * c < 64 ? n0 - 64 * c : 64, cn1 = n1 - 64 * c < 64 ? (n1 - 64 * c > 0 ? n1 - 64 * c : 0) : 64;
                CaptureRaw r0, r1;
                r0.a0 = r0.a1 = r0.a2 = r0.a3 = r0.b0 = r0.b1 = r0.b2 = r0.b3 = 0;
                r1 = r0;
                /* the body runs for whole groups of D pairs: past the end of the last chunk it parks clamped rows in lanes nobody reads */
                for (int j0 = 0; j0 < cn0; j0 += D) {
#pragma unroll
                    for (int d = 0; d < D; d++) {
                        const int j = j0 + d;
                        const uint8_t p0 = q[d][0], p1 = q[d][1], p2 = q[d][2], p3 = q[d][3];
                        const uint64_t aA_lo = __ballot(p0 > lo), bA_lo = __ballot(p0 >= hi), aA_hi = __ballot(p1 > lo), bA_hi = __ballot(p1 >= hi);
                        const uint64_t aB_lo = __ballot(p2 > lo), bB_lo = __ballot(p2 >= hi), aB_hi = __ballot(p3 > lo), bB_hi = __ballot(p3 >= hi);
                        {   /* the pair D places ahead (the chunks follow each other without a gap: j0 is a multiple of D). Its
                             * request waits for the compares above, so that the bytes land in the registers those just freed */
                            uint32_t oa, ob; pair_offsets(64 * c + j + D, oa, ob);
                            SDV_ISSUE_AFTER8(oa, ob, aA_lo, bA_lo, aA_hi, bA_hi, aB_lo, bB_lo, aB_hi, bB_hi);
                            q[d][0] = stream_load_u8(frame + (oa + x0)); q[d][1] = stream_load_u8(frame + (oa + x1));
                            q[d][2] = stream_load_u8(frame + (ob + x0)); q[d][3] = stream_load_u8(frame + (ob + x1));
                        }
                        capture_park(r0, aA_lo, aA_hi, bA_lo, bA_hi, j);
                        capture_park(r1, aB_lo, aB_hi, bB_lo, bB_hi, j);
                    }
                }
                /* every lane its own line: automaton + CRC */
                K1_T(t_s0);
                BatchLaneOut o0, o1;
                const bool ok0 = capture_solve(r0, o0), ok1 = capture_solve(r1, o1);
                const uint64_t okm0 = __ballot(ok0 || lane >= cn0), okm1 = __ballot(ok1 || lane >= cn1);
                const int n_ok0 = okm0 == ~0ull ? cn0 : (__ffsll((unsigned long long)~okm0) - 1);
                const int n_ok1 = okm1 == ~0ull ? cn1 : (__ffsll((unsigned long long)~okm1) - 1);
                ok0_packed |= (uint64_t)n_ok0 << (7 * c); ok1_packed |= (uint64_t)n_ok1 << (7 * c);
                K1_T(t_s1);
                K1_ADD(5, t_s0, t_s1);
                /* field 1 waits for the end of field 0: the cells only (of a line that read the CRC as calculated is the one in its cells, and
                 * only such lines are taken from here) */
                park[(c * 4 + 0) * 64 + lane] = (uint32_t)o1.s_lo; park[(c * 4 + 1) * 64 + lane] = (uint32_t)(o1.s_lo >> 32);
                park[(c * 4 + 2) * 64 + lane] = (uint32_t)o1.s_hi; park[(c * 4 + 3) * 64 + lane] = (uint32_t)(o1.s_hi >> 32);
#pragma unroll
                for (int u = 0; u < HOLD_CHUNKS; u++)
                    if (c == u) { hold[u][0] = (uint32_t)o0.s_lo; hold[u][1] = (uint32_t)(o0.s_lo >> 32); hold[u][2] = (uint32_t)o0.s_hi; hold[u][3] = (uint32_t)(o0.s_hi >> 32); }
                n_done = c + 1;
                if (n_ok0 < cn0) { whole = false; start_field = 0; start_idx = 64 * c + n_ok0; }
            }
            /* Straight into the stitcher's field buffers (FrameArgs::direct_fields) when the capture took every line of both fields: known here, before
             * anything of the frame is written.  Not the last frame of the call (the stitcher keeps that one's records for the next call), not a frame with
             * a service line of its own in front of it. */
            bool direct = false;
            if (a.direct_fields && whole && f + 1 < a.n_total && f != a.new_file_frame && n0 <= a.direct_lines && n1 <= a.direct_lines) {
                direct = true;
                for (int c = 0; c < n_chunks; c++) {
                    const int cn1 = n1 - 64 * c < 64 ? (n1 - 64 * c > 0 ? n1 - 64 * c : 0) : 64;
                    if ((int)((ok1_packed >> (7 * c)) & 0x7F) != cn1) direct = false;
                }
            }
            uint4 *const dfield0 = direct ? (uint4 *)a.direct_fields + ((size_t)(f + a.direct_seg_ofs) * 2u) * (size_t)a.direct_pitch * 2u : nullptr;      /* (a line is two uint4) */
            uint4 *const dfield1 = direct ? dfield0 + (size_t)a.direct_pitch * 2u : nullptr;
            int forced0 = 0, forced1 = 0;
            /* field 0 goes through the per-line bookkeeping */
            K1_T(t_s1b);
            for (int c = 0; c < n_done; c++) {
                const int n_ok0 = (int)((ok0_packed >> (7 * c)) & 0x7F);
                BatchLane bl; bl.d0 = bl.d1 = bl.d2 = bl.d3 = 0;
#pragma unroll
                for (int u = 0; u < HOLD_CHUNKS; u++)
                    if (c == u) { bl.d0 = hold[u][0]; bl.d1 = hold[u][1]; bl.d2 = hold[u][2]; bl.d3 = hold[u][3]; }
                bl.meta = (uint32_t)rev16(bl.d3 >> 16);
                int nf = 0;
                if (n_ok0 > 0) { batch_finish(a, lds, v, bl, n_ok0, frame_no, (uint16_t)(1 + 2 * (64 * c)), fv_keys, rec, false, direct ? dfield0 + 128 * c : nullptr, &nf); rec += n_ok0; }
                forced0 += nf;
            }
            K1_T(t_s2);
            K1_ADD(6, t_s1b, t_s2);
            if (whole) {
                line_num = (uint16_t)(1 + 2 * n0);
                v2d_service_line(v, a, lds, wl, frame_no, line_num, SDV_SRV_END_FIELD);
                if (!direct) emit_record(wl, rec);
                rec++;
                start_field = 1; start_idx = 0;
                SDV_WAVE_SYNC();
                for (int c = 0; c < n_chunks && whole; c++) {
                    const int cn1 = n1 - 64 * c < 64 ? (n1 - 64 * c > 0 ? n1 - 64 * c : 0) : 64;
                    if (cn1 <= 0) break;
                    const int n_ok1 = (int)((ok1_packed >> (7 * c)) & 0x7F);
                    BatchLane bl;
                    bl.d0 = park[(c * 4 + 0) * 64 + lane]; bl.d1 = park[(c * 4 + 1) * 64 + lane]; bl.d2 = park[(c * 4 + 2) * 64 + lane];
                    bl.d3 = park[(c * 4 + 3) * 64 + lane]; bl.meta = (uint32_t)rev16(bl.d3 >> 16);
                    int nf = 0;
                    if (n_ok1 > 0) { batch_finish(a, lds, v, bl, n_ok1, frame_no, (uint16_t)(2 + 2 * (64 * c)), fv_keys, rec, false, direct ? dfield1 + 128 * c : nullptr, &nf); rec += n_ok1; }
                    forced1 += nf;
                    if (n_ok1 < cn1) { whole = false; start_idx = 64 * c + n_ok1; }
                }
                SDV_WAVE_SYNC();
                if (whole) {
                    line_num = (uint16_t)(2 + 2 * n1);
                    v2d_service_line(v, a, lds, wl, frame_no, line_num, SDV_SRV_END_FIELD);
                    if (!direct) emit_record(wl, rec);
                    rec++;
                    start_field = 2;
                    all_captured = true; captured_key = coords_key(v.bin.in_coord.start, v.bin.in_coord.stop);
                    frame_direct = direct;
                    if (direct) {
                        dsum.frame_number = frame_no; dsum.n[0] = (uint16_t)n0; dsum.n[1] = (uint16_t)n1; dsum.bad[0] = (uint16_t)forced0; dsum.bad[1] = (uint16_t)forced1;
                        dsum.ref = v.bin.in_ref; dsum.flag = 1;
                    }
                }
            }
            if (start_field < 2) {      /* the row-staging loop below takes over: its first row */
                /* (the coordinate keys of the lines taken so far, left out above: all of them the pair the frame was started with) */
                const uint32_t key = coords_key(v.bin.in_coord.start, v.bin.in_coord.stop);
#pragma unroll 1
                for (int i = lane; i < v.nfv; i += 64) fv_keys[i] = key;
                pf.nq = 0;
                row_prefetch(pf, frame + (size_t)(2 * start_idx + start_field) * a.row_stride, a.width);
            }
            K1_T(t_cap1);
            K1_ADD(4, t_cap, t_cap1);
        }
    }
    /* ... by a line nothing was tuned for: what is behind it hangs on what the sweep finds, the pass ends here (SweepHook::stop).  (Through the loop
     * conditions, not with a return from inside the loop: with the return, hipcc 7.2 built a full kernel that lost one `rec++` - every record of the
     * frame one slot early on the GPU, at -O1 to -O3 alike, while the same source was right under the emulator.) */
    bool stop_frame = false;
    int last_miss_field = -1, last_miss_idx = 0;
    /* Lines that read, but not on the first rung of the ladder (a data window a pixel or two beside the preset coordinates: the lines of a frame
     * started from a state ahead of a small jump of the window, or a tape that sits like that): the two-lines-a-turn loop of the batch ends on every
     * one of them and the line went through the one-line path - 10 000 cycles each, a batch attempt included.  Once a line was read like that the
     * batches take one line a turn and walk the ladder on the staged row themselves (phase B books any rung); eight lines in a row on the first
     * rung bring the fast loop back. */
    bool sticky_rung = false; int first_rung_run = 0;
    int calm_lines = 0;         /* lines taken by batches since the last line that went through the general path */
    int rb_lines = 16;          /* lines the next batch of that kind takes on: a batch decodes all of its lines before it knows where it ends, so a tape whose lines also fail
                                 * now and then (every one ends a batch) would decode most lines twice or more with batches of 64 - doubled by every batch that went through */
    int rung_hint = 0;          /* while sticky: the shift stage (at hysteresis depth 0) the last line taken on its own read at, 0 when it needed a deeper hysteresis */
    /* ahead of line `pos`: leave a snapshot; has the last complete pass stood here with the same state?  Then the rest of the frame is its - see TcSnap. */
    auto tc_point = [&](uint32_t pos) -> bool {
        if (!tc_on) return false;
        if (tc_w < TC_FINAL) { tc_write(tc_wr + tc_w, v, lds, pos, used_general); tc_w++; }
        if (!tc_prev) return false;
        const uint64_t at_m = __ballot(tc_my_pos == pos);
        if (at_m == 0ull) return false;
        const int r = __ffsll((unsigned long long)at_m) - 1;
        const uint32_t theirs = lane < TC_SNAP_DWORDS ? tc_rd[r].d[lane < TC_SNAP_DWORDS ? lane : 0] : 0u;
        const uint32_t mine = tc_dword(v, lds, lane < TC_USED_DWORDS ? lane : 0, pos, used_general);
#if defined(SDV_EMU) && defined(SDV_DEV_AIDS)
        if (getenv("SDV_TC_DEBUG")) {
            const uint64_t dm = __ballot(lane < TC_CMP_DWORDS && mine != theirs);
            if (lane == 0) fprintf(stderr, "[tc] frame %d pos %u: differing dwords %05llx%s\n", f, pos, (unsigned long long)dm, dm ? "" : "  -> met");
            if (dm && ((dm >> lane) & 1)) fprintf(stderr, "[tc]    dword %d: now %08x, last pass %08x\n", lane, mine, theirs);
        }
#endif
        if (__ballot(lane < TC_CMP_DWORDS && mine != theirs) != 0ull) return false;
        /* met.  The last pass's snapshot here and its final one, staged where every lane can read them (the brightness spread's room) */
        uint32_t *const stg = lds.hist;
        SDV_WAVE_SYNC();
        if (lane < TC_SNAP_DWORDS) { stg[lane] = theirs; stg[TC_SNAP_DWORDS + lane] = tc_rd[TC_FINAL].d[lane]; }
        SDV_WAVE_SYNC();
        const TcCounts was = tc_counts(stg), fin = tc_counts(stg + TC_SNAP_DWORDS);
        /* the coordinate keys of the lines behind: the last pass's, behind this pass's */
#pragma unroll 1
        for (int i = was.nfv + lane; i < fin.nfv; i += 64) fv_keys[v.nfv + (i - was.nfv)] = tc_rd_keys[i];
#pragma unroll 1
        for (int i = was.nfi + lane; i < fin.nfi; i += 64) fi_keys[v.nfi + (i - was.nfi)] = tc_rd_keys[a.height + i];
        /* the last pass's later snapshots are this pass's too, with what this pass has counted otherwise up to here (dwords 19 .. 24: two 16-bit counters each) */
        {
            const uint32_t d_lo = (mine & 0xFFFFu) - (theirs & 0xFFFFu), d_hi = (mine >> 16) - (theirs >> 16);
            const bool counted = lane >= 19 && lane < TC_USED_DWORDS;
#pragma unroll 1
            for (int e = r + 1; e < (int)tc_prev_n && tc_w < TC_FINAL; e++) {
                uint32_t x = lane < TC_SNAP_DWORDS ? tc_rd[e].d[lane < TC_SNAP_DWORDS ? lane : 0] : 0u;
                if (counted) x = (((x & 0xFFFFu) + d_lo) & 0xFFFFu) | ((((x >> 16) + d_hi) & 0xFFFFu) << 16);
                if (lane == 18 && used_general) x |= 0x10000u;
                if (lane < TC_USED_DWORDS) tc_wr[tc_w].d[lane] = x;
                tc_w++;
            }
        }
        /* the frame's counters at its end ... */
        v.nfv += fin.nfv - was.nfv; v.nfi += fin.nfi - was.nfi;
        if (fin.q[0] > v.q_line_length) v.q_line_length = fin.q[0];          /* (the line length is noted, not counted: 0 or the width) */
        v.q_odd = (uint16_t)(v.q_odd + fin.q[1] - was.q[1]); v.q_even = (uint16_t)(v.q_even + fin.q[2] - was.q[2]);
        v.q_pcm_odd = (uint16_t)(v.q_pcm_odd + fin.q[3] - was.q[3]); v.q_pcm_even = (uint16_t)(v.q_pcm_even + fin.q[4] - was.q[4]);
        v.q_bad_odd = (uint16_t)(v.q_bad_odd + fin.q[5] - was.q[5]); v.q_bad_even = (uint16_t)(v.q_bad_even + fin.q[6] - was.q[6]);
        v.q_dup_odd = (uint16_t)(v.q_dup_odd + fin.q[7] - was.q[7]); v.q_dup_even = (uint16_t)(v.q_dup_even + fin.q[8] - was.q[8]);
        used_general = used_general || fin.used_general;
        /* ... and its state there (the frame's start coordinates do not change within a frame, the histories of frames are this pass's own) */
        const uint32_t *const fs = stg + TC_SNAP_DWORDS;
        v.n_last = (int)((fs[16] >> 8) & 0xFFu); v.field_state = (uint8_t)(fs[16] & 0xFFu);
        SDV_WAVE_SYNC();
        if (lane < COORD_HISTORY_DEPTH) lds.lv_keys[lane] = fs[lane];
        SDV_WAVE_SYNC();
        for (int i = 0; i < 4; i++) { v.last_words[2 * i] = (uint16_t)fs[9 + i]; v.last_words[2 * i + 1] = (uint16_t)(fs[9 + i] >> 16); }
        v.bin.in_coord.start = (int16_t)(uint16_t)fs[13]; v.bin.in_coord.stop = (int16_t)(uint16_t)(fs[13] >> 16);
        v.bin.in_black = (uint8_t)fs[15]; v.bin.in_white = (uint8_t)(fs[15] >> 8); v.bin.in_ref = (uint8_t)(fs[15] >> 16);
        v.bin.in_coord.doubled = ((fs[15] >> 24) & 1u) != 0; v.bin.do_ref_lvl_sweep = ((fs[15] >> 24) & 2u) != 0;
        v2d_make_uniform(v);
        if (lane == 0 && a.memo_count) atomicAdd(a.memo_count + 1, 1);      /* (sdv_run_info::frames_met) */
        return true;
    };
    bool tc_due = !kLean && kMeet && tc_on;          /* the state may have changed since the last look: at the head of the frame, behind every line taken on its own */
    for (int field = start_field; field < 2 && !stop_frame && !tc_met; field++) {
        const int nl = n_field[field];
        int idx = field == start_field ? start_idx : 0;
        while (idx < nl && !stop_frame && !tc_met) {
            if (!kLean && kMeet && tc_due) { tc_due = false; if (tc_point((uint32_t)(field * 1024 + idx))) { tc_met = true; break; } }
            bool staged = false, batch_gave_way = false;
            /* the batch loop is the 16-byte-vector, single-vector-per-lane case (rows aligned, width a multiple of 16 up to 1024:
             * SD video); everything else takes the sequential path below */
            FastPre pre; pre.ok = false;
            if (pf.vec && a.width <= 1024 && (a.width & 15) == 0 && batch_eligible(a, lds, v, geo)) pre = fast_pre(a, v.bin, geo);
            if (pre.ok) {
                K1_T(t_batch);
                int nb = nl - idx; if (nb > 64) nb = 64;
                BatchLane bl; bl.d0 = bl.d1 = bl.d2 = bl.d3 = bl.meta = 0;
                int j = 0;
                bool redo = false;                              /* a line of a pair did not pass: stage it again for the sequential path */
                /* row of line k of this field in decode order; past the field: the other field's rows, then any valid row */
                auto row_in_order = [&](int k) -> const uint8_t * {
                    if (k < nl) return frame + (size_t)(2 * k + field) * a.row_stride;
                    if (field == 0 && (k - nl) < n_field[1]) return frame + (size_t)(2 * (k - nl) + 1) * a.row_stride;
                    return frame;
                };
                /* A tape that sits a few pixels beside the coordinates the binarizer holds reads every line on the same later shift stage, and the coordinates
                 * stay as they are (a line that reads hands its coordinates on).  Taking such lines one a turn through the ladder of reads cost 2.2 us a line - a
                 * chain of dependent steps per rung, 10 000 such frames 2.7 ms against 0.7 ms on the first rung.  Here a batch is taken the capture's way: per
                 * line the comparison masks of the shift stages 0 .. rung_hint (hysteresis depth 0) are parked in the lane that owns the line, then every lane
                 * solves its own line stage by stage; the first stage whose CRC holds is the line's (binarizer.cpp:7769-7954: the first pair that reads
                 * wins).  A line that reads on none of them ends the batch and takes the sequential path with the whole ladder. */
                /* (general build: only on a stretch of the frame that reads - 48 lines since the last one through the general path.  On a tape with damage every
                 * few dozen lines the stretches are short, lines that read on a later stage are single events there, and taking every line of the next batch
                 * through all the stages for them cost the C3 tape a fifth of its speed; the frames this build decodes behind a jump of the window - general
                 * path on their first lines, then 480 lines on the stage the new coordinates put them on - are the ones it is here for.) */
                if ((kLean || (kMeet && SDV_RB_GENERAL && calm_lines >= 48)) && sticky_rung && rung_hint >= 1 && rung_hint <= SHIFT_STAGES_MAX && rung_hint <= (int)v.bin.in_max_shift) {
                    const int n_rungs = rung_hint + 1;
                    /* (the masks wait in LDS, eight words per line and stage, behind the staged row - up to the end of the sweep table's room, which nothing holds
                     * anything in across a batch; kept in registers like the capture's they were 40 registers the lean build does not have) */
                    enum { RB_OFS = 1024, RB_SLOTS = (REC_STAGE_OFS + REC_STAGE_BYTES - RB_OFS) / 32 };
                    static_assert(RB_OFS + RB_SLOTS * 32 <= offsetof(WaveLds, crc_stats), "the parked masks end in front of the histories");
                    uint4 *const rb = (uint4 *)((uint8_t *)&lds + RB_OFS);          /* [line][stage][2] */
                    if (nb > RB_SLOTS / n_rungs) nb = RB_SLOTS / n_rungs;
                    if (nb > rb_lines) nb = rb_lines;
                    const int32_t x_hi_lim = a.width - 2;          /* fast_decode: clamped to [pixel_start, pixel_stop - 1] */
                    for (int jj = 0; jj < nb; jj++) {
                        SDV_WAVE_SYNC();
                        ((uint4 *)lds.px)[lane] = pf.v0;
                        SDV_WAVE_SYNC();
                        {   /* the next row in decode order (as the one-line loop below) */
                            const int k = idx + jj + 1;
                            if (pf.nq > 0) {
                                pf.row = pf.rowq[0]; pf.v0 = pf.vq[0];
#pragma unroll
                                for (int u = 0; u + 1 < SDV_ROWQ; u++) { pf.rowq[u] = pf.rowq[u + 1]; pf.vq[u] = pf.vq[u + 1]; }
                                pf.nq--;
                            } else { pf.row = row_in_order(k); pf.v0 = ((const uint4 *)pf.row)[pf.i0]; }
                        }
                        uint8_t p0[SHIFT_STAGES_MAX + 1], p1[SHIFT_STAGES_MAX + 1];
#pragma unroll
                        for (int r = 0; r <= SHIFT_STAGES_MAX; r++) {       /* all reads of the row first */
                            const int32_t sh = shift_of_stage(r);
                            int32_t xa = geo.vp0 + sh, xb = geo.vp1 + sh;
                            xa = xa < 0 ? 0 : (xa > x_hi_lim ? x_hi_lim : xa); xb = xb < 0 ? 0 : (xb > x_hi_lim ? x_hi_lim : xb);
                            p0[r] = r < n_rungs ? lds.px[xa] : (uint8_t)0; p1[r] = r < n_rungs ? lds.px[xb] : (uint8_t)0;
                        }
#pragma unroll
                        for (int r = 0; r <= SHIFT_STAGES_MAX; r++)
                            if (r < n_rungs) {
                                const uint64_t a_lo = __ballot(p0[r] > pre.ref_low), a_hi = __ballot(p1[r] > pre.ref_low);
                                const uint64_t b_lo = __ballot(p0[r] >= pre.ref_high), b_hi = __ballot(p1[r] >= pre.ref_high);
                                if (lane == 0) {
                                    uint4 *const d = rb + (size_t)(jj * n_rungs + r) * 2u;
                                    d[0] = uint4{(uint32_t)a_lo, (uint32_t)(a_lo >> 32), (uint32_t)a_hi, (uint32_t)(a_hi >> 32)};
                                    d[1] = uint4{(uint32_t)b_lo, (uint32_t)(b_lo >> 32), (uint32_t)b_hi, (uint32_t)(b_hi >> 32)};
                                }
                            }
                    }
                    SDV_WAVE_SYNC();
                    int my_rung = -1; BatchLaneOut mine; mine.s_lo = mine.s_hi = 0; mine.crc = 0;
                    {
                        const uint4 *const my = rb + (size_t)((lane < nb ? lane : 0) * n_rungs) * 2u;
#pragma unroll 1
                        for (int r = 0; r < n_rungs; r++) {
                            const uint4 ma = my[2 * r], mb = my[2 * r + 1];
                            BatchLaneOut o; bool reads;
                            if (kLean) {
                                CaptureRaw c; c.a0 = ma.x; c.a1 = ma.y; c.a2 = ma.z; c.a3 = ma.w; c.b0 = mb.x; c.b1 = mb.y; c.b2 = mb.z; c.b3 = mb.w;
                                reads = capture_solve<true>(c, o);
                            } else {        /* (the general build lives at the edge of its register budget: inlined, the solve cost it 48 more spilled registers and the damaged tapes a quarter of their speed) */
                                const SolveOut so = capture_solve_call(ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w);
                                o.s_lo = (uint64_t)so.d0 | ((uint64_t)so.d1 << 32); o.s_hi = (uint64_t)so.d2 | ((uint64_t)so.d3 << 32); o.crc = (uint16_t)so.crc_reads;
                                reads = (so.crc_reads & 0x10000u) != 0;
                            }
                            if (my_rung < 0 && reads) { my_rung = r; mine = o; }
                        }
                    }
                    SDV_WAVE_SYNC();            /* (the room is batch_finish's record stage next) */
                    const bool taken = my_rung >= 0 && !ctrl_block_maybe(mine.s_lo);
                    const uint64_t taken_m = __ballot(taken || lane >= nb);
                    const int n_ok = taken_m == ~0ull ? nb : (__ffsll((unsigned long long)~taken_m) - 1);
                    bl.d0 = (uint32_t)mine.s_lo; bl.d1 = (uint32_t)(mine.s_lo >> 32); bl.d2 = (uint32_t)mine.s_hi; bl.d3 = (uint32_t)(mine.s_hi >> 32);
                    bl.meta = (uint32_t)mine.crc | ((uint32_t)(my_rung < 0 ? 0 : my_rung) << 20);
                    j = n_ok;
#if defined(SDV_EMU) && defined(SDV_DEV_AIDS)
                    if (lane == 0 && getenv("SDV_RB_DEBUG")) fprintf(stderr, "[rb] frame %d field %d idx %d: %d of %d lines taken on %d stages (%s)\n", f, field, idx, n_ok, nb, n_rungs, kLean ? "lean" : "general");
#endif
                    {   /* the stage the lines taken needed at most: the next batch tries no further; a batch of first-stage lines ends the sticky mode */
                        const uint32_t top = wave_max_u32(lane < n_ok && my_rung > 0 ? (uint32_t)my_rung : 0u);
                        if (n_ok >= 8 && top == 0u) { sticky_rung = false; first_rung_run = 0; }
                        else if (n_ok >= 8 && (int)top < rung_hint) rung_hint = (int)top;
                    }
                    if (n_ok < nb) { redo = true; pf.nq = 0; row_prefetch(pf, row_in_order(idx + n_ok), a.width); rb_lines = 16; }
                    else if (rb_lines < 64) rb_lines *= 2;
                }
                if (SDV_BATCH_LINES > 1 && !sticky_rung && !redo && j == 0) {
                    constexpr int NLA = SDV_ROWQ + 1 + (SDV_BATCH_LINES > 1 ? SDV_BATCH_LINES : 2);
                    const uint8_t *after[NLA];                  /* the rows that follow this field in decode order */
#pragma unroll
                    for (int q = 0; q < NLA; q++) after[q] = row_in_order(nl + q);
                    /* NL lines per iteration: inside a batch every line is decoded with the same inherited tuning, so the
                     * decode chains (LDS gather, ballots, automaton, CRC) are independent and interleave */
                    constexpr int NL = SDV_BATCH_LINES > 1 ? SDV_BATCH_LINES : 2;
                    /* R[0] = pf.v0 / pf.row is the row of line idx + j, R[1 + u] = pf.vq[u] the rows behind it: NL * SDV_PREFETCH_ITERS rows in
                     * registers or on their way */
                    constexpr int NR = SDV_ROWQ + 1;
                    auto beyond = [&](int d) -> const uint8_t * { const uint8_t *r = frame;
#pragma unroll
                        for (int q = 0; q < NLA; q++) r = d == q ? after[q] : r;
                        return r; };
                    for (; j + NL - 1 < nb; j += NL) {
                        /* rows that are not in flight yet (start of a batch after sequential lines) */
#pragma unroll
                        for (int i = 1; i < NR; i++)
                            if (i - 1 >= pf.nq) {
                                const int k = idx + j + i;
                                pf.rowq[i - 1] = k < nl ? row_in_order(k) : beyond(k - nl);
                                pf.vq[i - 1] = ((const uint4 *)pf.rowq[i - 1])[pf.i0];
                            }
                        pf.nq = NR - 1;
                        SDV_WAVE_SYNC();
                        ((uint4 *)lds.px)[lane] = pf.v0;
#pragma unroll
                        for (int u = 0; u < NL - 1; u++) ((uint4 *)lds.px)[64 * (u + 1) + lane] = pf.vq[u];
                        SDV_WAVE_SYNC();
                        {   /* the queue moves up by NL rows and NL new rows are requested at its end: one pointer step inside the field,
                             * past its end one of the rows looked up per batch */
                            const uint8_t *prev = NR > 1 ? pf.rowq[NR - 2] : pf.row;       /* the last row requested so far */
                            if (NR > NL) { constexpr int q0 = NR > NL ? NL - 1 : 0; pf.row = pf.rowq[q0]; pf.v0 = pf.vq[q0]; }
#pragma unroll
                            for (int i = 1; i + NL < NR; i++) { pf.rowq[i - 1] = pf.rowq[i - 1 + NL]; pf.vq[i - 1] = pf.vq[i - 1 + NL]; }
#pragma unroll
                            for (int u = 0; u < NL; u++) {
                                const int i = NR - NL + u;                                  /* slot in R of the new row */
                                const int k = idx + j + NL + i;                              /* its decode position */
                                const uint8_t *r = k < nl ? prev + row_step : beyond(k - nl);
                                const uint4 val = ((const uint4 *)r)[pf.i0];
                                if (i == 0) { pf.row = r; pf.v0 = val; } else { pf.rowq[i - 1] = r; pf.vq[i - 1] = val; }
                                prev = r;
                            }
                        }
                        FastBits fx[NL]; bool okx[NL]; FastCells cx[NL];
#pragma unroll
                        for (int u = 0; u < NL; u++) cx[u] = fast_sample(lds, pre, 1024 * u);      /* all LDS reads first */
                        if (NL == 2) fast_try0_x2(cx[0], cx[1], pre, lc, fx[0], fx[1], okx[0], okx[1]);
                        else {
#pragma unroll
                            for (int u = 0; u < NL; u++) okx[u] = fast_try0(cx[u], pre, lc, fx[u]);
                        }
                        int good = 0;
#pragma unroll
                        for (int u = 0; u < NL; u++) {
                            if (good == u && okx[u] && !ctrl_block_maybe(fx[u].s_lo)) {
                                good = u + 1;
                                bool mine = lane == j + u;
                                bl.d0 = mine ? (uint32_t)fx[u].s_lo : bl.d0; bl.d1 = mine ? (uint32_t)(fx[u].s_lo >> 32) : bl.d1;
                                bl.d2 = mine ? (uint32_t)fx[u].s_hi : bl.d2; bl.d3 = mine ? (uint32_t)(fx[u].s_hi >> 32) : bl.d3;
                                bl.meta = mine ? ((uint32_t)fx[u].calc_crc | ((uint32_t)fx[u].h << 16) | ((uint32_t)fx[u].s << 20)) : bl.meta;
                            }
                        }
                        if (good < NL) { j += good; redo = true; break; }
                    }
                    if (redo) {                                 /* line idx + j goes through the sequential path: fetch it again (it is in L2) */
                        pf.nq = 0;
                        row_prefetch(pf, row_in_order(idx + j), a.width);
                    }
                }
                for (; !redo && j < nb; j++) {
                    {   /* row_commit / row_prefetch of the plain case, without their case distinctions */
                        SDV_WAVE_SYNC();
                        ((uint4 *)lds.px)[lane] = pf.v0;
                        SDV_WAVE_SYNC();
                        int k = idx + j + 1;                    /* next row in decode order */
                        if (pf.nq > 0) {                        /* already fetched by the multi-line loop */
                            pf.row = pf.rowq[0]; pf.v0 = pf.vq[0];
#pragma unroll
                            for (int u = 0; u + 1 < SDV_ROWQ; u++) { pf.rowq[u] = pf.rowq[u + 1]; pf.vq[u] = pf.vq[u + 1]; }
                            pf.nq--;
                        }
                        else {
                            const uint8_t *nxt = (k < nl) ? pf.row + row_step
                                                          : (field == 0 && n_field[1] > 0 ? frame + a.row_stride : frame);
                            pf.row = nxt;
                            pf.v0 = ((const uint4 *)nxt)[pf.i0];
                        }
                    }
                    FastBits fb;
                    /* the first rung of the ladder; while lines need other rungs (sticky_rung) the whole ladder, on the row as it is staged.  A line
                     * that reads on no rung ends the batch and takes the sequential path below. */
                    const bool first_rung = fast_try0(fast_sample(lds, pre, 0), pre, lc, fb);
                    if (!first_rung && !(sticky_rung && fast_decode(a, lds, v.bin, v.bin.in_black, v.bin.in_white, geo, lc, fb))) break;
                    if (ctrl_block_maybe(fb.s_lo)) break;
                    if (!first_rung) first_rung_run = 0; else if (sticky_rung && ++first_rung_run >= 8) sticky_rung = false;
                    bool mine = lane == j;
                    bl.d0 = mine ? (uint32_t)fb.s_lo : bl.d0; bl.d1 = mine ? (uint32_t)(fb.s_lo >> 32) : bl.d1;
                    bl.d2 = mine ? (uint32_t)fb.s_hi : bl.d2; bl.d3 = mine ? (uint32_t)(fb.s_hi >> 32) : bl.d3;
                    bl.meta = mine ? ((uint32_t)fb.calc_crc | ((uint32_t)fb.h << 16) | ((uint32_t)fb.s << 20)) : bl.meta;
                }
                K1_T(t_loop);
                K1_ADD(1, t_batch, t_loop);
                if (j > 0) {
                    batch_finish(a, lds, v, bl, j, frame_no, (uint16_t)(field + 1 + 2 * idx), fv_keys, rec);
                    K1_T(t_fin);
                    K1_ADD(2, t_loop, t_fin);
                    rec += j; idx += j; calm_lines += j;
                }
                if (j == nb) continue;
                staged = !redo;                                 /* line idx sits in LDS and needs the sequential path */
                batch_gave_way = true;
            }
            K1_T(t_st0);
            if (!staged) {
                pf.nq = 0;
                row_commit(lds, pf, a.width);
                int k = idx + 1;
                const uint8_t *nxt = (k < nl) ? pf.row + row_step
                                              : (field == 0 && n_field[1] > 0 ? frame + a.row_stride : frame);
                row_prefetch(pf, nxt, a.width);
            }
            line_num = (uint16_t)(field + 1 + 2 * idx);
            K1_T(t_fl0);
            K1_ADD(21, t_st0, t_fl0);
            bool ladder_failed = false;
            int line_rung = -1;
            bool took_fast = fast_line<false>(a, lds, v, geo, lc, frame_no, line_num, fv_keys, rec, &ladder_failed, nullptr, &line_rung);
            if (!kLean && !took_fast && !ladder_failed) { bool lf2; took_fast = fast_line<true>(a, lds, v, geo, lc, frame_no, line_num, fv_keys, rec, &lf2, a.bw_memo ? a.bw_memo + ((size_t)f * (size_t)a.height + (size_t)(2 * idx + field)) : nullptr); }
            K1_T(t_fl1);
            if (took_fast && batch_gave_way) {                  /* (a line the batch ended on, read by the ladder) */
                if (!sticky_rung) { sticky_rung = true; first_rung_run = 0; }
                rung_hint = line_rung > 0 ? line_rung : 0;
            }
            if (!took_fast) K1_ADD(14, t_fl0, t_fl1);
            if (!kLean && took_fast) { K1_ADD(4, t_fl0, t_fl1); K1_ADD(5, 0ull, 1ull); }
            if (!took_fast) {
                if (kLean) {
                    v2d_give_up(a, lds, f, a.sig ? give_up_signature(a, lds, v.bin.in_ref) : (uint8_t)0xFF);      /* (the line sits in LDS: staged by the batch that ended on it, or just above) */
                    return;
                } else {
                    K1_T(t_sc0);
                    used_general = true; calm_lines = 0;
                    SDV_SLOW_CTX(c);
                    SDV_WAVE_SYNC();
                    if (!sc_args_set) { c.a = a; sc_args_set = true; }
                    c.v = v; c.line_only = false;
                    c.hook.memo = a.memo; c.hook.head = a.memo_head; c.hook.count = a.memo_count; c.hook.cap = a.memo_cap;
                    c.hook.frame = f; c.hook.row = (uint16_t)(2 * idx + field); c.hook.line = f * a.height + (2 * idx + field); c.hook.pending = false; c.hook.stop = false; c.hook.ladder_failed = ladder_failed;
                    c.hook.bw_slot = a.bw_memo ? a.bw_memo + ((size_t)f * (size_t)a.height + (size_t)(2 * idx + field)) : nullptr;
                    c.hook.fat = kFat ? fat : nullptr; c.hook.fa = &c.a; c.hook.fat_levels = fat_levels;
                    SDV_WAVE_SYNC();
                    slow_line<(kFat ? 3 : kMeet ? 0 : 1)>(&c, &lds, frame_no, line_num, fv_keys, fi_keys, rec);
#ifdef SDV_K1_STAMPS
                    n_slow_lines++;
#endif
                    SDV_WAVE_SYNC();
                    v = c.v;
                    v2d_make_uniform(v);
                    K1_T(t_sc1);
                    K1_ADD(20, t_sc0, t_sc1);
                    const bool missed = uni(c.hook.pending) != 0;
                    sweep_pending = sweep_pending || missed;
                    stop_frame = uni(c.hook.stop) != 0;
                    /* Two sweeps owed within a few lines of a field: the first of them may be one that finds a level the line reads with, and with it
                     * the tuning every line behind it needs - going on "as if it had found nothing" can leave all of those unreadable, each of them
                     * asking for a sweep of its own (seen: 342 lines of one frame, 46 ms, on a tape whose level drifts within the frame).  The pass
                     * over the frame ends here; it comes again when the two are settled. */
                    if (missed) { if (last_miss_field == field && idx - last_miss_idx < 8) stop_frame = true; last_miss_field = field; last_miss_idx = idx; }
                }
            }
            rec++; idx++;
            tc_due = !kLean && kMeet && tc_on;
        }
        if (stop_frame || tc_met) break;
        /* spliceFrame: END_FIELD carries the number the next line of the field would have had */
        line_num = (uint16_t)(field + 1 + 2 * nl);
        v2d_service_line(v, a, lds, wl, frame_no, line_num, SDV_SRV_END_FIELD);
        emit_record(wl, rec++);
    }
    if (stop_frame) {                       /* (what was decoded of the frame is not final anyway) */
        if (lane == 0) a.flag[f] = VF_ABORTED;
        K1_FLUSH();
        return;
    }
    if (kMeet && tc_met) {      /* the rest of the frame is the last pass's: on to the END_FRAME record */
        rec = a.recs + rec_base + (f == a.new_file_frame ? 1 : 0) + (size_t)a.height + 2u;
        line_num = (uint16_t)(2 + 2 * n_field[1]);
    }
    if (kMeet && tc_on && !sweep_pending) {      /* a complete pass: its snapshots are the ones the next pass of this frame looks at */
        tc_write(tc_wr + TC_FINAL, v, lds, TC_POS_NONE, used_general);
        if (lane == 0) { a.tc_hdr[2 * f + 1] = (uint32_t)tc_w; a.tc_hdr[2 * f] = (uint32_t)(tc_wbuf + 1); }
    }
    line_num = (uint16_t)(line_num + 2);
    v2d_service_line(v, a, lds, wl, frame_no, line_num, SDV_SRV_END_FRAME);
    K1_T(t_ef0);
    v2d_end_frame(v, a, lds, frame_no, fv_keys, fi_keys, &a.stats[f], all_captured, captured_key);
    if (!frame_direct) emit_record(wl, rec);
    rec++;
    if (a.direct_frames && (kLean || frame_direct) && lane_id() == 0) a.direct_frames[f] = dsum;
    v2d_store_state(v, lds, &a.states_out[f], a, sweep_pending, used_general ? (uint8_t)VF_SLOW : (uint8_t)0);
    K1_T(t_end);
    K1_ADD(3, t_ef0, t_end);
    K1_ADD(0, t_begin, t_end);
    K1_FLUSH();
#ifdef SDV_K1_STAMPS
    if (lane_id() == 0) atomicMax(&sdv_k1_cycles[7], ((unsigned long long)(t_end - t_begin) << 24) | ((unsigned long long)(n_slow_lines & 0x3FF) << 14) | (unsigned long long)(f & 0x3FFF));
#endif
}

/* ---- one line per wave: Binarizer::processLine with an STC007Line as output (sdv_binarize_lines) ------------------------------
 * What the frame kernels do per line of a frame, without the frame around it: the caller says what its Binarizer had been preset with before the line
 * (sdv_bin_state), the line comes back as processLine leaves it - before the bookkeeping VideoToDigital adds (duplicate lines, the coordinate damper), which
 * belongs to the frame entry.  A reference-level sweep is asked for and looked up like in the frame kernel (SweepHook; a list head per line): a line that
 * waits for one is left undone and decoded again when the engine has settled the round's requests. */
struct LineArgs7 {
    const uint8_t *luma; size_t row_stride; int width; uint32_t n_lines;
    const sdv_bin_state *states;        /* per line, or NULL: nothing preset */
    uint32_t frame_number; uint16_t first_line, line_step;
    uint8_t doubled, mode;
    sdv_bin_preset preset;
    sdv_line_rec *out;
    uint8_t *done;                      /* per line: its record is final */
    struct SweepMemo *memo; int32_t *memo_head; int32_t *memo_count; int32_t memo_cap;
};
enum { LINES_PER_MEMO_FRAME = 16384 };  /* (a sweep request names its pixels by frame and row: line i of the call is row i % 16384 of "frame" i / 16384) */
__device__ inline void stc_line_body(const LineArgs7 &a, WaveLds &lds, SlowCtx *sc, uint32_t i)
{
    if (uniu(a.done[i])) return;
    const int lane = lane_id();
    const uint8_t *row = a.luma + (size_t)i * a.row_stride;
    RowPrefetch pf;
    pf.v0 = uint4{0, 0, 0, 0}; pf.v1 = uint4{0, 0, 0, 0};
    for (int u = 0; u < SDV_ROWQ; u++) { pf.vq[u] = uint4{0, 0, 0, 0}; pf.rowq[u] = row; }
    pf.nq = 0;
    pf.vec = ((((uintptr_t)row) | (uintptr_t)a.row_stride) & 15) == 0 && a.width >= 16;
    { int nvec = a.width >> 4; pf.i0 = lane < nvec ? lane : (nvec > 0 ? nvec - 1 : 0); }
    row_prefetch(pf, row, a.width);
    row_commit(lds, pf, a.width);
    /* the general path's context (in LDS, one for the wave - the emulator keeps a copy per lane: SDV_SLOW_CTX) with what the line needs of it */
#ifdef SDV_EMU
    SlowCtx c;
#else
    SlowCtx &c = *sc;
#endif
    SDV_WAVE_SYNC();
    c.a.width = a.width; c.a.doubled = a.doubled; c.a.mode = a.mode; c.a.preset = a.preset; c.a.m2_format = 0;
    Bin &b = c.v.bin;
    b.in_black = b.in_white = b.in_ref = 0; coords_clear(b.in_coord); b.do_ref_lvl_sweep = false;
    if (a.states) {         /* setReferenceLevel, setDataCoordinates, setBWLevels (binarizer.cpp:240-350) */
        const sdv_bin_state st = a.states[i];
        b.in_ref = st.in_def_reference;
        Coords cc; cc.start = st.in_def_start; cc.stop = st.in_def_stop; cc.doubled = st.in_def_from_doubled != 0;
        bin_set_data_coordinates(b, cc);
        bin_set_bw_levels(b, a.preset, st.in_def_black, st.in_def_white);
        b.do_ref_lvl_sweep = st.do_ref_lvl_sweep != 0;
    }
    bin_set_mode(b, a.mode);
    b.hyst_lim = 0; b.shift_lim = 0;
    b.line_length = 0; b.scan_start = b.scan_end = 0; b.mark_start_max = 0; b.mark_end_min = 0xFFFF; b.estimated_ppb = 0;
    b.was_bw_scanned = false; b.vl_doubled = false;
    c.hook.memo = a.memo; c.hook.head = a.memo_head; c.hook.count = a.memo_count; c.hook.cap = a.memo_cap;
    c.hook.frame = (int32_t)(i / (uint32_t)LINES_PER_MEMO_FRAME); c.hook.row = (uint16_t)(i % (uint32_t)LINES_PER_MEMO_FRAME); c.hook.line = (int32_t)i;
    c.hook.pending = false; c.hook.bw_slot = nullptr; c.hook.ladder_failed = false; c.hook.stop = false;
    c.line_only = true;
    SDV_WAVE_SYNC();
    slow_line<2>(&c, &lds, a.frame_number, (uint16_t)(a.first_line + i * a.line_step), nullptr, nullptr, &a.out[i]);
    SDV_WAVE_SYNC();
    if (uni(c.hook.pending) != 0) return;             /* comes again with the sweep's outcome at hand (the record left now is written over then) */
    if (lane == 0) a.done[i] = 1;
}

} // namespace sdv

#ifndef SDV_WAVES_PER_EU
#define SDV_WAVES_PER_EU 3   /* round 6: the general build needs 168 registers with the trajectory snapshots; asked for more waves than that allows (8, the setting of rounds 1-5) the compiler
                              * takes 170 and the kernel drops to two waves per SIMD (8 000 PAL frames with lost lines 2.52 against 2.15 ms, profiles/r06_tuning_notes.md) */
#endif
__global__ void __launch_bounds__(64, SDV_WAVES_PER_EU) sdv_k_stc007_frames(sdv::FrameArgs a)
{
    __shared__ sdv::WaveLds lds;
    __shared__ sdv::SlowCtx slow_ctx;
    int f = a.frame_list ? a.frame_list[blockIdx.x] : a.frame_lo + (int)blockIdx.x;
    if (a.frame_list || f < a.frame_hi) sdv::frame_body<false>(a, lds, f, &slow_ctx);
}
#ifndef SDV_LEAN_WAVES_PER_EU
#define SDV_LEAN_WAVES_PER_EU 4   /* 128 registers: the capture holds a field's cells in registers without spilling.  Round 5, one box, ms per 10 000 frames: 0.688 (4) against
                                   * 0.765 (5: 96 registers, 6 of them spilled); the capture itself does not care (round 5's first build: 0.782 with 4 and with 5) */
#endif
/* the general build without snapshots and later-stage batches (frame_body, kMeet) */
#ifndef SDV_PLAIN_WAVES_PER_EU
#define SDV_PLAIN_WAVES_PER_EU 8    /* (asked for more waves than it can have the compiler settles on fewer registers and no spills: the setting of rounds 1-5 for this build) */
#endif
__global__ void __launch_bounds__(64, SDV_PLAIN_WAVES_PER_EU) sdv_k_stc007_frames_plain(sdv::FrameArgs a)
{
    __shared__ sdv::WaveLds lds;
    __shared__ sdv::SlowCtx slow_ctx;
    int f = a.frame_list ? a.frame_list[blockIdx.x] : a.frame_lo + (int)blockIdx.x;
    if (a.frame_list || f < a.frame_hi) sdv::frame_body<false, false>(a, lds, f, &slow_ctx);
}
/* ... and of small rounds: the frame's wave and four that settle the sweeps it misses (stc007_sweep_device.h, fat_sweep) */
__global__ void __launch_bounds__(64 * (1 + sdv::FAT_WORKERS)) sdv_k_stc007_frames_fat(sdv::FrameArgs a)
{
    __shared__ sdv::WaveLds lds;
    __shared__ sdv::SlowCtx slow_ctx;
    __shared__ sdv::FatLds fat;
    const int f = a.frame_list ? a.frame_list[blockIdx.x] : a.frame_lo + (int)blockIdx.x;
    sdv::SweepEnt *const levels = a.fat_levels + (size_t)blockIdx.x * 256u;
#ifndef SDV_EMU
    if (threadIdx.x >= 64) { sdv::fat_worker(a, fat, levels, (int)(threadIdx.x >> 6) - 1); return; }
#endif
    if (a.frame_list || f < a.frame_hi) sdv::frame_body<false, false, true>(a, lds, f, &slow_ctx, &fat, levels);
    sdv::fat_exit(fat);
}
__global__ void __launch_bounds__(64, SDV_LEAN_WAVES_PER_EU) sdv_k_stc007_frames_lean(sdv::FrameArgs a)
{
    __shared__ sdv::WaveLds lds;
    int f = a.frame_list ? a.frame_list[blockIdx.x] : a.frame_lo + (int)blockIdx.x;
    if (a.frame_list || f < a.frame_hi) sdv::frame_body<true>(a, lds, f);
}
/* (the bound of the general kernels: the general path is one function for all of them and is given the registers of its most generous caller - which every
 * caller then has to set aside) */
__global__ void __launch_bounds__(64, SDV_WAVES_PER_EU) sdv_k_stc007_lines(sdv::LineArgs7 a)
{
    __shared__ sdv::WaveLds lds;
    __shared__ sdv::SlowCtx slow_ctx;
    for (uint32_t i = blockIdx.x; i < a.n_lines; i += gridDim.x) { sdv::stc_line_body(a, lds, &slow_ctx, i); __syncthreads(); }
}
#endif
