/* TEST INFRASTRUCTURE ONLY.  Helpers of oracle/binarizer.c that do not depend on the line type, shared with the PCM-1
 * restatement (oracle/bin_pcm1.c).  Reference lines are cited at the definitions. */
#ifndef ORC_BIN_INTERNAL_H
#define ORC_BIN_INTERNAL_H
#include "sdv_oracle.h"

void reset_crc_stats(orc_crc_handler *a, uint16_t count, uint8_t *valid_cnt);
void update_crc_stats(orc_crc_handler *a, orc_crc_handler in, uint8_t *valid_cnt);
void find_most_frequent_crc(orc_crc_handler *a, uint8_t *valid_cnt, bool skip_equal);
void invalidate_non_frequent_crcs(orc_crc_handler *a, uint8_t low_level, uint8_t high_level, uint8_t valid_cnt, uint16_t target_crc);
uint8_t pick_level_by_crc_stats(const orc_crc_handler *crcs, uint8_t *ref_result, uint8_t low_lvl, uint8_t high_lvl,
                                uint8_t target_result, uint8_t max_hyst, uint8_t max_shift);
uint8_t pick_level_by_crc_stats_opt(const orc_binarizer *b, const orc_crc_handler *crcs, uint8_t *ref_result, uint8_t low_lvl, uint8_t high_lvl,
                                    uint8_t target_result, uint8_t max_hyst, uint8_t max_shift);
uint16_t most_frequent_brightness_count(const uint16_t *s);
uint8_t usefull_low_level(const orc_binarizer *b, const uint16_t *s);
uint8_t usefull_high_level(const orc_binarizer *b, const uint16_t *s);
uint8_t get_low_level(uint8_t in_lvl, uint8_t diff);
uint8_t get_high_level(uint8_t in_lvl, uint8_t diff);
uint8_t pick_center_ref_level(const orc_binarizer *b, uint8_t lvl_black, uint8_t lvl_white);
bool is_ref_level_preset(const orc_binarizer *b);
bool are_bw_levels_preset(const orc_binarizer *b);
void calc_forced_coords(const orc_binarizer *b, orc_coords *fc);
bool coords_ne(const orc_coords *a, const orc_coords *b);

#endif
