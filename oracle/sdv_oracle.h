/*
 * sdv_oracle.h - CPU restatement (plain C) of the SDVPCMdecoder hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, link or call it,
 * and only as the checker / CPU baseline.  The product (sdvpcmdecoder_amd/) never links it.
 *
 * Every function cites the reference file:line (Fagear/SDVPCMdecoder v0.99.7) that it follows.
 * Parity pinning: see oracle/README.md (reference KATs from pcmtester.cpp + bit-exact comparison
 * against the reference itself compiled into oracle/_ref/ by oracle/Makefile.ref).
 */
#ifndef SDV_ORACLE_H
#define SDV_ORACLE_H

#include <stdint.h>
#include <stdbool.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants (reference: pcmline.h:60-102, stc007line.h:71-151, binarizer.h:194-288) ---- */
enum { ORC_PS_STAGES = 5 };                       /* PCM_LINE_MAX_PS_STAGES pcmline.h:60 */
enum { ORC_CRC_INIT = 0xFFFF, ORC_CRC_POLY = 0x1021 };
enum { ORC_INT_CALC_MULT = 128 };
enum { ORC_NO_COORD_LEFT = -32768, ORC_NO_COORD_RIGHT = 32767 };  /* frametrimset.h:35-38 */

/* service tags, PCMLine (pcmline.h:104-115); VideoLine uses the first six (videoline.h:41-49) */
enum {
    ORC_SRV_NO = 0, ORC_SRV_NEW_FILE, ORC_SRV_END_FILE, ORC_SRV_FILLER, ORC_SRV_END_FIELD,
    ORC_SRV_END_FRAME, ORC_SRV_HEADER_LINE, ORC_SRV_CTRL_BLOCK
};

/* STC-007 constants (stc007line.h:71-151) */
enum {
    ORC_STC_BITS_PER_WORD = 14, ORC_STC_WORD_MASK = 0x3FFF, ORC_STC_BITS_START = 4,
    ORC_STC_BITS_DATA = 128, ORC_STC_BITS_STOP = 5, ORC_STC_BITS_IN_LINE = 137,
    ORC_STC_BITS_LEFT_SHIFT = 24, ORC_STC_BITS_RIGHT_SHIFT = 76,
    ORC_STC_WORD_CNT = 9, ORC_STC_WORD_Q = 7, ORC_STC_WORD_CRC = 8,
    ORC_STC_CRC_SILENT = 0xA96A
};
enum { ORC_MARK_ST_START = 0, ORC_MARK_ST_TOP_1, ORC_MARK_ST_BOT_1, ORC_MARK_ST_TOP_2, ORC_MARK_ST_BOT_2 };
enum { ORC_MARK_ED_START = 0, ORC_MARK_ED_TOP, ORC_MARK_ED_BOT, ORC_MARK_ED_LEN_OK };

/* Binarizer modes / limits (binarizer.h:207-247) */
enum { ORC_MODE_DRAFT = 0, ORC_MODE_FAST, ORC_MODE_NORMAL, ORC_MODE_INSANE };
enum { ORC_MIN_VALID_CRCS = 5, ORC_MAX_COLL_CRCS = 32 };
enum { ORC_HYST_DEPTH_MIN = 0, ORC_HYST_DEPTH_SAFE = 4, ORC_HYST_DEPTH_MAX = 10 };
enum { ORC_SHIFT_STAGES_MIN = 0, ORC_SHIFT_STAGES_SAFE = 2, ORC_SHIFT_STAGES_MAX = 4 };
enum { ORC_LB_RET_OK = 0, ORC_LB_RET_NULL_VIDEO, ORC_LB_RET_NULL_PCM, ORC_LB_RET_SHORT_LINE, ORC_LB_RET_NO_COORD };
enum { ORC_STG_INPUT_ALL = 0, ORC_STG_INPUT_LEVEL, ORC_STG_REF_FIND, ORC_STG_REF_SWEEP_RUN,
       ORC_STG_READ_PCM, ORC_STG_DATA_OK, ORC_STG_NO_GOOD, ORC_STG_MAX };
enum { ORC_REF_NO_PCM = 0, ORC_REF_BAD_CRC, ORC_REF_CRC_COLL, ORC_REF_CRC_OK };
enum { ORC_SPAN_NOT_FOUND = 0, ORC_SPAN_TOO_NARROW, ORC_SPAN_OK };

/* ---- CoordinatePair (frametrimset.h:29-66) ---- */
typedef struct {
    uint8_t reference;
    int16_t data_start, data_stop;
    bool from_doubled, not_sure;
} orc_coords;

/* ---- VideoLine view (videoline.h:37-88) ---- */
typedef struct {
    uint32_t frame_number;
    uint16_t line_number;
    const uint8_t *pixels;
    uint16_t length;
    bool empty, doubled;
    uint8_t service_type;
    bool scan_done;             /* VideoLine::scan_done (videoline.h:58): set by the PCM-16x0 coordinate search, which then runs once per video line */
} orc_video_line;

/* ---- STC007Line : PCMLine (pcmline.h:137-166, stc007line.h:153-165) ---- */
typedef struct {
    /* PCMLine public */
    uint32_t frame_number; uint16_t line_number;
    uint8_t black_level, white_level, ref_low, ref_level, ref_high;
    orc_coords coords;
    uint8_t hysteresis_depth, shift_stage;
    bool ref_level_sweeped, coords_sweeped, data_by_ext_tune;
    /* PCMLine protected/private */
    uint16_t calc_crc;
    bool blk_wht_set, coords_set, forced_bad;
    uint8_t service_type;
    uint16_t pixel_start, pixel_stop;
    int16_t pixel_start_offset;
    uint32_t pixel_size_mult, halfpixel_size_mult;
    /* STC007Line */
    uint8_t mark_st_stage, mark_ed_stage;
    uint16_t marker_start_bg_coord, marker_start_ed_coord, marker_stop_ed_coord;
    bool m2_format;
    uint16_t pixel_coordinates[ORC_PS_STAGES][ORC_STC_BITS_DATA];
    bool word_crc[ORC_STC_WORD_CNT], word_valid[ORC_STC_WORD_CNT];
    uint16_t words[ORC_STC_WORD_CNT];
} orc_stc_line;

/* ---- bin_preset_t (binarizer.h:163-186) ---- */
typedef struct {
    uint8_t max_black_lvl, min_white_lvl, min_contrast, min_ref_lvl, max_ref_lvl, min_valid_crcs;
    uint8_t mark_max_dist, left_bit_pick, right_bit_pick;
    orc_coords horiz_coords;
    bool en_force_coords, en_coord_search, en_first_line_dup, en_good_no_marker;
} orc_bin_preset;

/* ---- crc_handler_t (binarizer.h:151-160) ---- */
typedef struct {
    uint8_t result; uint16_t crc; uint8_t hyst_dph, shift_stg; int16_t data_start, data_stop;
} orc_crc_handler;

/* ---- Binarizer (binarizer.h:306-337) ---- */
typedef struct {
    orc_bin_preset digi_set;
    const orc_video_line *video_line;   /* scan_done is written through it by the PCM-16x0 path (bin_pcm16.c) */
    orc_stc_line *out_pcm_line;
    uint8_t in_def_black, in_def_white, in_def_reference;
    orc_coords in_def_coord;
    uint8_t in_max_hysteresis_depth, in_max_shift_stages;
    bool do_coord_search, do_start_mark_sweep, do_ref_lvl_sweep, force_bit_picker;
    uint8_t proc_state, bin_mode, line_part_mode, hysteresis_depth_lim, shift_stages_lim;
    uint16_t line_length, scan_start, scan_end, mark_start_max, mark_end_min, estimated_ppb;
    bool was_BW_scanned;
    bool p1_scan_done;                  /* VideoLine::scan_done as left by findPCM1Coordinates (binarizer.cpp:5810) */
    orc_crc_handler shift_crcs[ORC_SHIFT_STAGES_MAX + 1];
    /* hyst_crcs is [HYST_DEPTH_MAX+1]; the reference reads one element past it
     * (binarizer.cpp:8005 with hyst_cnt+1) which lands on crc_stats[0] in the class layout
     * (binarizer.h:335-337).  The two arrays are kept contiguous here for the same effect. */
    orc_crc_handler hyst_crcs[ORC_HYST_DEPTH_MAX + 1];
    orc_crc_handler crc_stats[ORC_MAX_COLL_CRCS + 1];
} orc_binarizer;

/* ---- CRC (pcmline.cpp:455-487) ---- */
uint16_t orc_crc16_update(uint16_t crc, uint16_t in_data, uint8_t bit_cnt);
uint16_t orc_stc_crc_words(const uint16_t *words8);   /* stc007line.cpp:245-251 */
uint16_t orc_crc16_bytes(const uint8_t *data, size_t n); /* check value helper, pcmline.h:88-97 */

/* ---- CoordinatePair ---- */
void orc_coords_clear(orc_coords *c);
bool orc_coords_set(orc_coords *c, int16_t start, int16_t stop);
bool orc_coords_valid(const orc_coords *c);
bool orc_coords_lt(const orc_coords *a, const orc_coords *b);

/* ---- STC007Line ---- */
void orc_stc_clear(orc_stc_line *l);
void orc_stc_calc_crc(orc_stc_line *l);
void orc_stc_set_silent(orc_stc_line *l);
void orc_stc_set_invalid_crc(orc_stc_line *l);
void orc_stc_calc_ppb(orc_stc_line *l, orc_coords in_coords);
bool orc_stc_crc_valid_ignore_forced(const orc_stc_line *l);
bool orc_stc_crc_valid(const orc_stc_line *l);
bool orc_stc_has_markers(const orc_stc_line *l);
bool orc_stc_has_control_block(const orc_stc_line *l);
void orc_stc_set_serv_ctrl_blk(orc_stc_line *l);
void orc_stc_apply_crc_state_per_word(orc_stc_line *l);
int16_t orc_stc_get_sample(const orc_stc_line *l, uint8_t index);
uint8_t orc_stc_words_diff_bit_count(const orc_stc_line *l, const orc_stc_line *other);
bool orc_stc_is_almost_silent(const orc_stc_line *l);
bool orc_stc_is_silent(const orc_stc_line *l);
void orc_stc_set_service(orc_stc_line *l, uint8_t service_type);

/* ---- Binarizer ---- */
void orc_bin_preset_reset(orc_bin_preset *p);
void orc_binarizer_init(orc_binarizer *b);
void orc_binarizer_set_mode(orc_binarizer *b, uint8_t mode);
void orc_binarizer_set_bw_levels(orc_binarizer *b, uint8_t black, uint8_t white);
void orc_binarizer_set_reference_level(orc_binarizer *b, uint8_t ref);
void orc_binarizer_set_data_coordinates(orc_binarizer *b, orc_coords c);
void orc_binarizer_set_data_coordinates2(orc_binarizer *b, int16_t start, int16_t stop);
void orc_binarizer_set_good_parameters(orc_binarizer *b, const orc_stc_line *line /* NULL = reset */);
uint8_t orc_binarizer_process_line(orc_binarizer *b);

#ifdef __cplusplus
}
#endif
/* A dropped frame (VideoInFFMPEG::insertDummyFrame(false, true), vin_ffmpeg.cpp:367-522): the frame's lines arrive as VideoLines marked empty
 * (no service tag, the length of a line, no pixels looked at).  Set by the *_run loops for the frame they are about to feed (orc_set_empty_frames). */
extern int orc_g_empty_frame;
void orc_set_empty_frames(const uint8_t *mask, size_t n);

#endif
