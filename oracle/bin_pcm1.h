/* TEST INFRASTRUCTURE ONLY.  PCM-1 front half of the oracle (oracle/bin_pcm1.c): PCM1Line and the PCM-1 paths of Binarizer. */
#ifndef ORC_BIN_PCM1_H
#define ORC_BIN_PCM1_H
#include "sdv_oracle.h"

enum { ORC_LB_RET_UNSUPPORTED = 100 };      /* not a reference code: a path the oracle does not restate (PCM-1 in MODE_INSANE) */

/* PCM1Line : PCMLine (pcmline.h:137-166, pcm1line.h:59-110) */
typedef struct {
    uint32_t frame_number; uint16_t line_number;
    uint8_t black_level, white_level, ref_low, ref_level, ref_high;
    orc_coords coords;
    uint8_t hysteresis_depth, shift_stage;
    bool ref_level_sweeped, coords_sweeped, data_by_ext_tune;
    uint16_t calc_crc;
    bool blk_wht_set, coords_set, forced_bad;
    uint8_t service_type;
    uint16_t pixel_start, pixel_stop;
    int16_t pixel_start_offset;
    uint32_t pixel_size_mult, halfpixel_size_mult;
    uint8_t picked_bits_left, picked_bits_right;
    uint16_t pixel_coordinates[ORC_PS_STAGES][94];
    uint16_t words[7];
} orc_p1_line;

void orc_p1_clear(orc_p1_line *l);
bool orc_p1_has_header(const orc_p1_line *l);
bool orc_p1_crc_valid_ignore_forced(const orc_p1_line *l);
bool orc_p1_crc_valid(const orc_p1_line *l);
void orc_binarizer_set_good_parameters_p1(orc_binarizer *b, const orc_p1_line *line /* NULL = reset */);
uint8_t orc_binarizer_process_line_p1(orc_binarizer *b, orc_p1_line *out);

#endif
