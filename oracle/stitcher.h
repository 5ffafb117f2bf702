/* stitcher.h - STC007DataStitcher restatement (oracle/stitcher.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_STITCHER_H
#define ORC_STITCHER_H
#include "sdv_oracle.h"
#include "deint.h"
#include "../include/sdvpcm.h"
#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_VID_UNKNOWN = 0, ORC_VID_PAL, ORC_VID_NTSC, ORC_VID_MAX };
enum { ORC_ORDER_UNK = 0, ORC_ORDER_TFF, ORC_ORDER_BFF, ORC_ORDER_MAX };
enum { ORC_LINES_PF_NTSC = 245, ORC_LINES_PF_PAL = 294, ORC_LINES_PF_MAX_PAL = 294 + 16, ORC_LINES_PF_MAX_NTSC = 294 - 32 };
enum { ORC_BUF_SIZE_TRIM = 3 * 640 * 3, ORC_BUF_SIZE_FIELD = 294, ORC_MIN_GOOD_LINES_PF = 245 - 8, ORC_MIN_FILL_LINES_PF = 56 };
enum { ORC_MAX_PADDING_14BIT = 32, ORC_MAX_PADDING_16BIT = 16, ORC_MAX_BURST_SILENCE = 8, ORC_MAX_BURST_BROKEN = 1,
       ORC_MAX_BURST_UNCH_DELTA = 8, ORC_MAX_BURST_UNCH_14BIT = 0x40, ORC_MAX_BURST_UNCH_16BIT = 0x20, ORC_UNCH_MASK_DURATION = 128,
       ORC_STATS_DEPTH = 65 };
enum { ORC_SAMPLE_RES_UNKNOWN = 0, ORC_SAMPLE_RES_14BIT, ORC_SAMPLE_RES_16BIT, ORC_SAMPLE_RES_MAX };
enum { ORC_DS_RET_NO_DATA = 0, ORC_DS_RET_SILENCE, ORC_DS_RET_BROKE, ORC_DS_RET_NO_PAD, ORC_DS_RET_OK };

/* FrameAsmSTC007 (frametrimset.h:116-275) */
typedef struct {
    uint32_t frame_number;
    uint16_t odd_std_lines, even_std_lines, odd_data_lines, even_data_lines, odd_valid_lines, even_valid_lines;
    uint16_t odd_top_data, odd_bottom_data, even_top_data, even_bottom_data, odd_sample_rate, even_sample_rate;
    uint8_t field_order; bool odd_emphasis, even_emphasis; uint8_t odd_ref, even_ref;
    uint16_t blocks_total, blocks_drop, samples_drop; bool drawn, order_preset, order_guessed; uint8_t service_type;
    uint8_t video_standard, tff_cnt, bff_cnt, odd_resolution, even_resolution;
    uint16_t inner_padding, outer_padding;
    bool trim_ok, inner_padding_ok, outer_padding_ok, inner_silence, outer_silence, vid_std_preset, vid_std_guessed;
    uint16_t blocks_broken_field, blocks_broken_seam, blocks_fix_p, blocks_fix_q, blocks_fix_cwd;
    int8_t ctrl_index, ctrl_hour, ctrl_minute, ctrl_second, ctrl_field;
} orc_frasm;

typedef struct { uint16_t index, valid, silent, unchecked, broken; } orc_stitch_stats;      /* FieldStitchStats */
typedef struct { uint8_t data[ORC_STATS_DEPTH]; size_t fill_cnt, head_i, tail_i; bool is_full; } orc_circ65;   /* circarray<uint8_t,65> */
typedef struct { orc_stc_line *v; size_t head, n, cap; } orc_line_deque;

/* PCMSamplePair as a POD (pcmsamplepair.h:31-143). 16 bytes. */
typedef struct {
    int16_t audio_word[2];
    uint8_t data_block_ok[2], word_valid[2], word_fixed[2], word_masked[2];
    uint16_t sample_rate;
    uint8_t emphasis, service_type;
} orc_sample_pair;

typedef struct {
    orc_stc_block padding_block;
    orc_deint pad_checker, lines_to_block;
    orc_frasm frasm_f0, frasm_f1, frasm_f2;
    orc_line_deque in_lines, padding_queue, conv_queue;
    orc_stc_line *trim_buf, *frame1_even, *frame1_odd, *frame2_even, *frame2_odd;
    uint8_t max_unchecked_14b_blocks, max_unchecked_16b_blocks, broken_mask_dur, broken_countdown;
    uint8_t preset_video_mode, preset_field_order, preset_audio_res, last_pad_counter;
    orc_circ65 stats_field_order, stats_resolution;
    uint16_t trim_fill, f1_max_line, f2_max_line, preset_sample_rate;
    bool file_start, file_end, ignore_CRC, enable_P_code, enable_Q_code, enable_CWD, mode_m2, fix_cut_above, mask_seams;
    /* outputs */
    orc_sample_pair *out; size_t out_n, out_cap;
    orc_frasm *frames; size_t frames_n, frames_cap;        /* guiUpdFrameAsm emissions */
    orc_stc_block *blocks; size_t blocks_n, blocks_cap;    /* newBlockProcessed emissions (outputDataBlock :6626), kept when keep_blocks is set */
    bool keep_blocks;
    orc_stc_line *asm_lines; size_t asm_n, asm_cap; size_t *asm_frame_n; size_t asm_frames, asm_frames_cap;   /* newLineProcessed emissions and how many each turn made */
} orc_stitcher;

void orc_stitcher_init(orc_stitcher *s);
void orc_stitcher_free(orc_stitcher *s);
void orc_stitcher_push_line(orc_stitcher *s, const orc_stc_line *l);
/* one turn of the doFrameReassemble loop: returns true if a frame was processed */
bool orc_stitcher_step(orc_stitcher *s);
#ifdef __cplusplus
}
#endif
#endif
