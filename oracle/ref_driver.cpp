/*
 * ref_driver.cpp - thin C-ABI shim around the REAL reference classes (compiled from
 * /root/reference by oracle/Makefile.ref into oracle/_ref/libsdvref.so).
 *
 * TEST INFRASTRUCTURE ONLY.  This file is our own code; it only *calls* the reference's public
 * API (Binarizer, STC007Line, VideoLine, STC007Deinterleaver, VideoToDigital, STC007DataStitcher).
 * No reference source is copied.  The resulting library is used (a) to validate the C restatement
 * in oracle/ bit-for-bit, (b) to generate the golden fixtures under tests/golden/, and (c) as the
 * `cpu_baseline.kind = "reference"` leg of bench.py when it loads on the GPU box.
 */
#include <cstring>
#include <cstdint>
#include <vector>
#include <deque>
#include <string>

#include "binarizer.h"
#include "stc007line.h"
#include "videoline.h"
#include "stc007datablock.h"
#include "stc007deinterleaver.h"

#include "../include/sdvpcm.h"

static void line_to_rec(STC007Line &l, sdv_line_rec *r)
{
    memset(r, 0, sizeof(*r));
    r->frame_number = l.frame_number;
    r->line_number = l.line_number;
    for (int i = 0; i < 9; i++) r->words[i] = l.getWord(i);
    r->calc_crc = l.getCalculatedCRC();
    r->data_start = l.coords.data_start;
    r->data_stop = l.coords.data_stop;
    r->marker_start_bg_coord = l.marker_start_bg_coord;
    r->marker_start_ed_coord = l.marker_start_ed_coord;
    r->marker_stop_ed_coord = l.marker_stop_ed_coord;
    r->black_level = l.black_level; r->white_level = l.white_level;
    r->ref_low = l.ref_low; r->ref_level = l.ref_level; r->ref_high = l.ref_high;
    r->hysteresis_depth = l.hysteresis_depth; r->shift_stage = l.shift_stage;
    uint8_t st = SDV_SRV_NO;
    if (l.isServNewFile()) st = SDV_SRV_NEW_FILE;
    else if (l.isServEndFile()) st = SDV_SRV_END_FILE;
    else if (l.isServFiller()) st = SDV_SRV_FILLER;
    else if (l.isServEndField()) st = SDV_SRV_END_FIELD;
    else if (l.isServEndFrame()) st = SDV_SRV_END_FRAME;
    else if (l.isServCtrlBlk()) st = SDV_SRV_CTRL_BLOCK;
    else if (l.isServiceLine()) st = 0xFF;
    r->service_type = st;
    r->mark_st_stage = l.mark_st_stage; r->mark_ed_stage = l.mark_ed_stage;
    uint8_t f = 0;
    if (l.isDataByRefSweep()) f |= SDV_LF_REF_SWEEPED;
    if (l.isDataByCoordSweep()) f |= SDV_LF_COORDS_SWEEPED;
    if (l.isDataBySkip()) f |= SDV_LF_BY_EXT_TUNE;
    if (l.hasBWSet()) f |= SDV_LF_BW_SET;
    if (l.hasDataCoordSet()) f |= SDV_LF_COORDS_SET;
    if (l.isForcedBad()) f |= SDV_LF_FORCED_BAD;
    if (l.isCRCValid()) f |= SDV_LF_CRC_VALID;
    if (l.isSourceDoubleWidth()) f |= SDV_LF_FROM_DOUBLED;
    r->flags = f;
    /* isWordCRCOk/isWordValid are masked by forced_bad in the getters (stc007line.cpp:656-680);
     * report them that way - it is what any consumer of the object can observe. */
    uint8_t ws = 0;
    if (l.isWordCRCOk(0)) ws |= SDV_WS_WORD_CRC;
    if (l.isWordValid(0)) ws |= SDV_WS_WORD_VALID;
    r->word_state = ws;
}

struct RefBin {
    Binarizer bin;
    VideoLine vline;
    STC007Line out;
};

extern "C" {

uint16_t ref_crc_stc007(const uint16_t *words8)
{
    STC007Line l;
    for (int i = 0; i < 8; i++) l.setWord(i, words8[i]);
    l.calcCRC();
    return l.getCalculatedCRC();
}

void *ref_bin_new(void) { return new RefBin(); }
void ref_bin_free(void *h) { delete (RefBin *)h; }
void ref_bin_set_mode(void *h, int mode) { ((RefBin *)h)->bin.setMode((uint8_t)mode); }
void ref_bin_set_coord_search(void *h, int on) { ((RefBin *)h)->bin.setCoordinatesSearch(on != 0); }

void ref_bin_set_preset(void *h, const sdv_bin_preset *p)
{
    RefBin *r = (RefBin *)h;
    bin_preset_t s = r->bin.getDefaultFineSettings();
    s.max_black_lvl = p->max_black_lvl; s.min_white_lvl = p->min_white_lvl; s.min_contrast = p->min_contrast;
    s.min_ref_lvl = p->min_ref_lvl; s.max_ref_lvl = p->max_ref_lvl; s.min_valid_crcs = p->min_valid_crcs;
    s.mark_max_dist = p->mark_max_dist; s.left_bit_pick = p->left_bit_pick; s.right_bit_pick = p->right_bit_pick;
    s.en_force_coords = p->en_force_coords; s.en_coord_search = p->en_coord_search;
    s.en_first_line_dup = p->en_first_line_dup; s.en_good_no_marker = p->en_good_no_marker;
    s.horiz_coords.data_start = p->horiz_start; s.horiz_coords.data_stop = p->horiz_stop;
    r->bin.setFineSettings(s);
}

/* Binarizer::setGoodParameters(NULL) */
void ref_bin_reset_good(void *h) { ((RefBin *)h)->bin.setGoodParameters(NULL); }
/* Binarizer::setGoodParameters(&last output line) */
void ref_bin_set_good_from_last(void *h) { RefBin *r = (RefBin *)h; r->bin.setGoodParameters(&r->out); }
/* explicit preset (setReferenceLevel / setDataCoordinates / setBWLevels, same order as setGoodParameters) */
void ref_bin_set_state(void *h, const sdv_bin_state *s)
{
    RefBin *r = (RefBin *)h;
    r->bin.setReferenceLevel(s->in_def_reference);
    CoordinatePair c;
    c.data_start = s->in_def_start; c.data_stop = s->in_def_stop; c.from_doubled = s->in_def_from_doubled != 0;
    r->bin.setDataCoordinates(c);
    r->bin.setBWLevels(s->in_def_black, s->in_def_white);
}

int ref_bin_process(void *h, const uint8_t *px, int len, uint32_t frame, uint16_t line, int service, int doubled, int empty,
                    sdv_line_rec *out)
{
    RefBin *r = (RefBin *)h;
    r->vline.clear();
    r->vline.frame_number = frame;
    r->vline.line_number = line;
    if (service == SDV_SRV_NO) {
        r->vline.setEmpty(empty != 0);          /* setEmpty() also clears pixel_data */
        if (!empty) r->vline.pixel_data.assign(px, px + len);
        r->vline.setDoubleWidth(doubled != 0);
    } else if (service == SDV_SRV_NEW_FILE) r->vline.setServNewFile("synthetic");
    else if (service == SDV_SRV_END_FILE) r->vline.setServEndFile();
    else if (service == SDV_SRV_FILLER) r->vline.setServFiller();
    else if (service == SDV_SRV_END_FIELD) r->vline.setServEndField();
    else if (service == SDV_SRV_END_FRAME) r->vline.setServEndFrame();
    r->bin.setSource(&r->vline);
    r->bin.setOutput(&r->out);
    int ret = r->bin.processLine();
    line_to_rec(r->out, out);
    return ret;
}

} /* extern "C" */
