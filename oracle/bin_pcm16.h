/* TEST INFRASTRUCTURE ONLY.  PCM-16x0 front half of the oracle (oracle/bin_pcm16.c): PCM16X0SubLine and the PCM-16x0 paths of Binarizer. */
#ifndef ORC_BIN_PCM16_H
#define ORC_BIN_PCM16_H
#include "sdv_oracle.h"
#include "bin_pcm1.h"

enum { ORC_P16_BITS_IN_LINE = 193, ORC_P16_BITS_PCM_DATA = 64, ORC_P16_SUBLINES = 3 };
enum { ORC_PART_FULL_LINE = 0, ORC_PART_PCM16X0_LEFT, ORC_PART_PCM16X0_MIDDLE, ORC_PART_PCM16X0_RIGHT };     /* Binarizer::FULL_LINE.. (binarizer.h:217-224) */

/* PCM16X0SubLine : PCMLine (pcmline.h:137-166, pcm16x0subline.h:113-125) */
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
    bool control_bit;
    uint8_t line_part, picked_bits_left, picked_bits_right;
    uint16_t queue_order;
    uint16_t pixel_coordinates[ORC_PS_STAGES][ORC_P16_BITS_IN_LINE];
    uint16_t words[4];
} orc_p16_line;

void orc_p16_clear(orc_p16_line *l);
bool orc_p16_crc_valid_ignore_forced(const orc_p16_line *l);
bool orc_p16_crc_valid(const orc_p16_line *l);
uint16_t orc_p16_crc_words(const uint16_t *w3);
void orc_binarizer_set_good_parameters_p16(orc_binarizer *b, const orc_p16_line *line /* NULL = reset */);
/* b->video_line->scan_done is read and written (findPCM16X0Coordinates, binarizer.cpp:5846, 6039) */
uint8_t orc_binarizer_process_line_p16(orc_binarizer *b, orc_p16_line *out);

#endif
