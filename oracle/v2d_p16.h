/* v2d_p16.h - types of the PCM-16x0 VideoToDigital restatement (oracle/v2d_p16.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_V2D_P16_H
#define ORC_V2D_P16_H
#include "sdv_oracle.h"
#include "bin_pcm16.h"
#include "v2d.h"
#ifdef __cplusplus
extern "C" {
#endif

/* VideoToDigital members (videotodigital.h:112-135) + the locals of doBinarize that live across frames (videotodigital.cpp:700-720) */
typedef struct {
    orc_bin_preset fine_bin_preset;
    orc_binarizer line_converter;
    uint8_t binarization_mode;
    bool check_line_copy, coordinate_damper, reset_stats;
    orc_frame_stats signal_quality;
    uint8_t field_state, prescan_ref;
    uint16_t line_in_field_cnt, good_coords_in_field, pcm_lines_in_field;
    orc_coords frame_avg, target_coord;
    orc_coord_list last_valid_coord_list, frame_valid_coord_list, frame_invalid_coord_list, long_valid_coords;
    orc_p16_line pcm16x0_line, last_pcm16x0_line[3];
} orc_v2d16;

void orc_v2d16_init(orc_v2d16 *v);
void orc_v2d16_free(orc_v2d16 *v);
void orc_v2d16_set_fine_settings(orc_v2d16 *v, const orc_bin_preset *p);
/* one frame; returns the records written: 3 per video line + 3 service lines (+1 with new_file); the filler frame: height + 4 */
int orc_v2d16_frame(orc_v2d16 *v, const uint8_t *luma, size_t stride, int width, int height, uint32_t frame_no,
                    bool new_file, bool doubled, bool filler, sdv_pcm16x0_bin_rec *out, orc_frame_stats *out_stats);
#ifdef __cplusplus
}
#endif
#endif
