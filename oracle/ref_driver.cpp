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

/* ------------------------------------------------------------------ VideoToDigital level */
#include <thread>
#include <functional>
#include <chrono>
#include <QMutex>
#include <QObject>
#include "videotodigital.h"

template <class LineT> struct RefV2DT {
    VideoToDigital v2d;
    std::deque<VideoLine> in_q;
    std::deque<LineT> out_q;
    QMutex in_mtx, out_mtx;
    std::deque<FrameBinDescriptor> stats_q;
    QMutex stats_mtx;
    std::thread th;
    bool started;
    RefV2DT() : started(false) {}
};
typedef RefV2DT<STC007Line> RefV2D;

static void stats_to_pod(FrameBinDescriptor &q, sdv_frame_stats *s)
{
    memset(s, 0, sizeof(*s));
    s->frame_id = q.frame_id; s->line_length = q.line_length;
    s->lines_odd = q.lines_odd; s->lines_even = q.lines_even;
    s->lines_pcm_odd = q.lines_pcm_odd; s->lines_pcm_even = q.lines_pcm_even;
    s->lines_bad_odd = q.lines_bad_odd; s->lines_bad_even = q.lines_bad_even;
    s->lines_dup_odd = q.lines_dup_odd; s->lines_dup_even = q.lines_dup_even;
    s->data_start = q.data_coord.data_start; s->data_stop = q.data_coord.data_stop;
    s->data_from_doubled = q.data_coord.from_doubled; s->data_not_sure = q.data_coord.not_sure;
}

extern "C" {

void *ref_v2d_new(void)
{
    RefV2D *r = new RefV2D();
    r->v2d.setInputPointers(&r->in_q, &r->in_mtx);
    r->v2d.setOutSTC007Pointers(&r->out_q, &r->out_mtx);
    r->v2d.setPCMType(VideoToDigital::TYPE_STC007);
    QObject::connect(&r->v2d, &VideoToDigital::guiUpdFrameBin, [r](FrameBinDescriptor d) {
        r->stats_mtx.lock(); r->stats_q.push_back(d); r->stats_mtx.unlock();
    });
    return r;
}
void ref_v2d_delete(void *h)
{
    RefV2D *r = (RefV2D *)h;
    if (r->started) { r->v2d.stop(); r->th.join(); }
    delete r;
}
void ref_v2d_set_mode(void *h, int mode) { ((RefV2D *)h)->v2d.setBinarizationMode((uint8_t)mode); }
void ref_v2d_set_check_line_dup(void *h, int on) { ((RefV2D *)h)->v2d.setCheckLineDup(on != 0); }
void ref_v2d_set_m2(void *h, int on) { ((RefV2D *)h)->v2d.setPCMType(on ? VideoToDigital::TYPE_M2 : VideoToDigital::TYPE_STC007); }
static bin_preset_t to_bin_preset(const sdv_bin_preset *p);
void ref_v2d_set_preset(void *h, const sdv_bin_preset *p) { ((RefV2D *)h)->v2d.setFineSettings(to_bin_preset(p)); }
}
static bin_preset_t to_bin_preset(const sdv_bin_preset *p)
{
    bin_preset_t s;
    s.max_black_lvl = p->max_black_lvl; s.min_white_lvl = p->min_white_lvl; s.min_contrast = p->min_contrast;
    s.min_ref_lvl = p->min_ref_lvl; s.max_ref_lvl = p->max_ref_lvl; s.min_valid_crcs = p->min_valid_crcs;
    s.mark_max_dist = p->mark_max_dist; s.left_bit_pick = p->left_bit_pick; s.right_bit_pick = p->right_bit_pick;
    s.en_force_coords = p->en_force_coords; s.en_coord_search = p->en_coord_search;
    s.en_first_line_dup = p->en_first_line_dup; s.en_good_no_marker = p->en_good_no_marker;
    s.horiz_coords.data_start = p->horiz_start; s.horiz_coords.data_stop = p->horiz_stop;
    return s;
}
extern "C" {

static void push_service(std::deque<VideoLine> &q, int kind, uint32_t frame, uint16_t line)
{
    VideoLine s;
    if (kind == SDV_SRV_NEW_FILE) s.setServNewFile("synthetic.avi");
    else if (kind == SDV_SRV_END_FIELD) s.setServEndField();
    else if (kind == SDV_SRV_END_FRAME) s.setServEndFrame();
    else if (kind == SDV_SRV_END_FILE) s.setServEndFile();
    else if (kind == SDV_SRV_FILLER) s.setServFiller();
    s.frame_number = frame; s.line_number = line;
    q.push_back(s);
}

} /* extern "C" */

/* dropped frames of the next *_run call: mask[f] != 0 = frame f of that call arrives as VideoInFFMPEG::insertDummyFrame(false, true) makes it
 * (vin_ffmpeg.cpp:367-522): lines without a service tag, of the length of a line, marked empty.  Consumed by the call. */
static std::vector<uint8_t> g_empty_frames;
extern "C" void ref_set_empty_frames(const uint8_t *mask, size_t n) { g_empty_frames.assign(mask, mask + (mask ? n : 0)); }

/* Feeds n_frames frames in VideoInFFMPEG::spliceFrame order (vin_ffmpeg.cpp:213-364) through the REAL
 * VideoToDigital worker loop and collects its STC007Line output and FrameBinDescriptor emissions. */
template <class LineT, class RecT>
static long v2d_run_t(RefV2DT<LineT> *r, const uint8_t *luma, size_t stride, int width, int height, int n_frames, uint32_t first_frame_no,
                      int new_file, int doubled, RecT *out, sdv_frame_stats *stats, void (*to_rec)(LineT &, RecT *), int recs_per_line)
{
    if (!r->started) { r->started = true; r->th = std::thread([r]() { r->v2d.doBinarize(); }); }
    const bool end_file = (new_file & 2) != 0;       /* bit 1: append VideoInFFMPEG::insertDummyFrame(true, false) */
    new_file &= 1;
    const int n_real = n_frames;
    if (end_file) n_frames++;
    long expect = (long)n_real * ((long)height * recs_per_line + 3) + (new_file ? 1 : 0) + (end_file ? height + 4 : 0);
    long got = 0; int fed = 0; int nstats = 0;
    while (got < expect || nstats < n_frames) {
        /* keep a few frames queued */
        r->in_mtx.lock();
        size_t qs = r->in_q.size();
        r->in_mtx.unlock();
        while (fed < n_frames && qs < (size_t)(3 * (height + 4))) {
            std::deque<VideoLine> tmp;
            uint32_t fno = first_frame_no + (uint32_t)fed;
            const uint8_t *fr = luma + (size_t)fed * stride * (size_t)height;
            if (new_file && fed == 0) push_service(tmp, SDV_SRV_NEW_FILE, fno, 0);
            uint16_t line_num = 0;
            if (fed >= n_real) {                      /* the filler frame that closes the file (vin_ffmpeg.cpp:367-523) */
                for (int field = 0; field < 2; field++) {
                    int line_offset = field;
                    for (;;) {
                        line_num = (uint16_t)(line_offset + 1);
                        push_service(tmp, SDV_SRV_FILLER, fno, line_num);
                        if (line_offset < (height - 2)) line_offset += 2;
                        else { line_num += 2; break; }
                    }
                    push_service(tmp, SDV_SRV_END_FIELD, fno, line_num);
                }
                line_num += 2; push_service(tmp, SDV_SRV_END_FILE, fno, line_num);
                line_num += 2; push_service(tmp, SDV_SRV_END_FRAME, fno, line_num);
                r->in_mtx.lock();
                for (auto &l : tmp) r->in_q.push_back(l);
                qs = r->in_q.size();
                r->in_mtx.unlock();
                fed++;
                continue;
            }
            for (int field = 0; field < 2; field++) {
                int line_offset = field;
                line_num = (uint16_t)(line_offset + 1);
                for (;;) {
                    VideoLine v;
                    v.frame_number = fno; v.line_number = line_num;
                    v.setDoubleWidth(doubled != 0);
                    if ((size_t)fed < g_empty_frames.size() && g_empty_frames[(size_t)fed]) { v.setServNo(); v.setLength((uint16_t)width); v.setEmpty(true); }     /* dummy_line of a dropped frame */
                    else v.pixel_data.assign(fr + (size_t)line_offset * stride, fr + (size_t)line_offset * stride + width);
                    tmp.push_back(v);
                    if (line_offset < (height - 2)) line_offset += 2;
                    else { line_num += 2; break; }
                    line_num += 2;
                }
                push_service(tmp, SDV_SRV_END_FIELD, fno, line_num);
            }
            line_num += 2;
            push_service(tmp, SDV_SRV_END_FRAME, fno, line_num);
            r->in_mtx.lock();
            for (auto &l : tmp) r->in_q.push_back(l);
            qs = r->in_q.size();
            r->in_mtx.unlock();
            fed++;
        }
        /* drain */
        r->out_mtx.lock();
        while (!r->out_q.empty() && got < expect) { to_rec(r->out_q.front(), &out[got++]); r->out_q.pop_front(); }
        r->out_mtx.unlock();
        r->stats_mtx.lock();
        while (!r->stats_q.empty() && nstats < n_frames) { if (stats) stats_to_pod(r->stats_q.front(), &stats[nstats]); nstats++; r->stats_q.pop_front(); }
        r->stats_mtx.unlock();
        if (got < expect || nstats < n_frames) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    g_empty_frames.clear();
    return got;
}

extern "C" long ref_v2d_run(void *h, const uint8_t *luma, size_t stride, int width, int height, int n_frames, uint32_t first_frame_no,
                            int new_file, int doubled, sdv_line_rec *out, sdv_frame_stats *stats)
{
    return v2d_run_t<STC007Line, sdv_line_rec>((RefV2D *)h, luma, stride, width, height, n_frames, first_frame_no, new_file, doubled, out, stats, line_to_rec, 1);
}


/* ------------------------------------------------------------------ deinterleaver level */
static void block_to_rec(STC007DataBlock &b, sdv_block_rec *r)
{
    memset(r, 0, sizeof(*r));
    for (int i = 0; i < 8; i++) {
        r->w_frame[i] = b.w_frame[i]; r->w_line[i] = b.w_line[i]; r->words[i] = b.getWord(i);
        if (b.isWordLineCRCOk(i)) r->line_crc |= (uint8_t)(1u << i);
        if (b.isWordCWDFixed(i)) r->cwd_fixed |= (uint8_t)(1u << i);
        if (b.isWordValid(i)) r->word_valid |= (uint8_t)(1u << i);
    }
    r->resolution = b.getResolution(); r->audio_state = b.getAudioState(); r->cwd_applied = b.cwd_applied; r->sample_rate = b.sample_rate;
}

extern "C" int ref_deint_run(const sdv_deint_line *lines, size_t n_lines, const sdv_deint_settings *st, sdv_block_rec *out, size_t n_blocks)
{
    std::deque<STC007Line> q;
    for (size_t i = 0; i < n_lines; i++) {
        const sdv_deint_line &in = lines[i];
        STC007Line l;
        l.frame_number = in.frame_number; l.line_number = in.line_number;
        for (uint8_t k = 0; k < 8; k++) l.setWord(k, in.words[k], (in.word_crc_ok >> k) & 1);
        l.calcCRC();
        l.setSourceCRC(l.getCalculatedCRC());
        if (in.flags & SDV_DL_FIXED_BY_CWD) {
            bool done = false;
            for (uint8_t k = 0; k < 8 && !done; k++) if (!l.isWordCRCOk(k)) { l.setFixed(k); done = true; }
            if (!done) { l.setWord(0, in.words[0], false); l.setFixed(0); }
        } else {
            bool any = false;
            for (uint8_t k = 0; k < 8; k++) if (!l.isWordCRCOk(k)) any = true;
            if (any) l.setInvalidCRC();
        }
        if (in.flags & SDV_DL_COORDS_BW_OK) { l.setBWLevelsState(true); l.coords.setCoordinates(10, 700); }
        q.push_back(l);
    }
    STC007Deinterleaver d;
    d.setInput(&q);
    d.setResMode(st->res_mode); d.setIgnoreCRC(st->ignore_crc); d.setForcedErrorCheck(st->force_ecc_check);
    /* set the three switches directly in the dependency-safe order of the GUI (stitcher: setPCorrection/Q/CWD) */
    d.setPCorrection(true); d.setQCorrection(st->en_q_code); d.setCWDCorrection(st->en_cwd);
    if (!st->en_p_code) d.setPCorrection(false);
    int rc = STC007Deinterleaver::DI_RET_OK;
    for (size_t s = 0; s < n_blocks; s++) {
        STC007DataBlock b;
        d.setOutput(&b);
        rc = d.processBlock((uint16_t)s);
        if (rc != STC007Deinterleaver::DI_RET_OK) break;
        block_to_rec(b, &out[s]);
    }
    return rc;
}

/* ------------------------------------------------------------------ stitcher level */
#include "stc007datastitcher.h"
#include "pcmsamplepair.h"

static void rec_to_line(const sdv_line_rec &r, STC007Line &l)     /* the binding of INTEGRATION.md */
{
    l.clear();
    l.frame_number = r.frame_number; l.line_number = r.line_number;
    switch (r.service_type) {
        case SDV_SRV_NEW_FILE: l.setServNewFile("synthetic.avi"); return;
        case SDV_SRV_END_FILE: l.setServEndFile(); return;
        case SDV_SRV_FILLER: l.setServFiller(); return;
        case SDV_SRV_END_FIELD: l.setServEndField(); return;
        case SDV_SRV_END_FRAME: l.setServEndFrame(); return;
        default: break;
    }
    for (uint8_t i = 0; i < 8; i++) l.setWord(i, r.words[i]);
    l.setSourceCRC(r.words[8]); l.calcCRC();
    l.black_level = r.black_level; l.white_level = r.white_level;
    l.ref_low = r.ref_low; l.ref_level = r.ref_level; l.ref_high = r.ref_high;
    l.coords.data_start = r.data_start; l.coords.data_stop = r.data_stop;
    l.setFromDoubledState((r.flags & SDV_LF_FROM_DOUBLED) != 0);
    l.hysteresis_depth = r.hysteresis_depth; l.shift_stage = r.shift_stage;
    l.setSweepedReference((r.flags & SDV_LF_REF_SWEEPED) != 0);
    l.data_by_ext_tune = (r.flags & SDV_LF_BY_EXT_TUNE) != 0;
    l.setBWLevelsState((r.flags & SDV_LF_BW_SET) != 0);
    l.setDataCoordinatesState((r.flags & SDV_LF_COORDS_SET) != 0);
    l.mark_st_stage = r.mark_st_stage; l.mark_ed_stage = r.mark_ed_stage;
    l.marker_start_bg_coord = r.marker_start_bg_coord; l.marker_start_ed_coord = r.marker_start_ed_coord; l.marker_stop_ed_coord = r.marker_stop_ed_coord;
    if (r.flags & SDV_LF_FORCED_BAD) l.setForcedBad();
    l.applyCRCStatePerWord();
    if (r.service_type == SDV_SRV_CTRL_BLOCK) l.setServCtrlBlk();
}

static void frasm_to_pod(FrameAsmSTC007 &f, sdv_frame_asm *o)
{
    memset(o, 0, sizeof(*o));
    o->frame_number = f.frame_number;
    o->odd_std_lines = f.odd_std_lines; o->even_std_lines = f.even_std_lines; o->odd_data_lines = f.odd_data_lines; o->even_data_lines = f.even_data_lines;
    o->odd_valid_lines = f.odd_valid_lines; o->even_valid_lines = f.even_valid_lines;
    o->odd_top_data = f.odd_top_data; o->odd_bottom_data = f.odd_bottom_data; o->even_top_data = f.even_top_data; o->even_bottom_data = f.even_bottom_data;
    o->odd_sample_rate = f.odd_sample_rate; o->even_sample_rate = f.even_sample_rate;
    o->blocks_total = f.blocks_total; o->blocks_drop = f.blocks_drop; o->samples_drop = f.samples_drop;
    o->inner_padding = f.inner_padding; o->outer_padding = f.outer_padding;
    o->blocks_broken_field = f.blocks_broken_field; o->blocks_broken_seam = f.blocks_broken_seam;
    o->blocks_fix_p = f.blocks_fix_p; o->blocks_fix_q = f.blocks_fix_q; o->blocks_fix_cwd = f.blocks_fix_cwd;
    o->field_order = f.field_order; o->odd_ref = f.odd_ref; o->even_ref = f.even_ref;
    o->service_type = f.isServNewFile() ? 1 : (f.isServEndFile() ? 2 : 0);
    o->video_standard = f.video_standard; o->tff_cnt = f.tff_cnt; o->bff_cnt = f.bff_cnt; o->odd_resolution = f.odd_resolution; o->even_resolution = f.even_resolution;
    o->flags = (uint8_t)((f.isOrderPreset() ? SDV_FA_ORDER_PRESET : 0) | (f.isOrderGuessed() ? SDV_FA_ORDER_GUESSED : 0) | (f.trim_ok ? SDV_FA_TRIM_OK : 0) |
                         (f.inner_padding_ok ? SDV_FA_INNER_OK : 0) | (f.outer_padding_ok ? SDV_FA_OUTER_OK : 0) | (f.inner_silence ? SDV_FA_INNER_SILENCE : 0) |
                         (f.outer_silence ? SDV_FA_OUTER_SILENCE : 0) | (f.vid_std_preset ? SDV_FA_VID_STD_PRESET : 0));
    o->flags2 = (uint8_t)((f.odd_emphasis ? SDV_FA2_ODD_EMPHASIS : 0) | (f.even_emphasis ? SDV_FA2_EVEN_EMPHASIS : 0) | (f.vid_std_guessed ? SDV_FA2_VID_STD_GUESSED : 0));
    o->ctrl_index = f.ctrl_index; o->ctrl_hour = f.ctrl_hour; o->ctrl_minute = f.ctrl_minute; o->ctrl_second = f.ctrl_second; o->ctrl_field = f.ctrl_field;
}

static std::vector<sdv_asm_line_rec> g_asm; static std::vector<uint32_t> g_asm_turns; static uint32_t g_asm_in_turn;
extern "C" void ref_stitch_last_asm_lines(sdv_asm_line_rec *out, size_t cap, size_t *n_lines, uint32_t *per_turn, size_t turns_cap, size_t *n_turns)
{
    for (size_t i = 0; i < g_asm.size() && i < cap; i++) out[i] = g_asm[i];
    for (size_t i = 0; i < g_asm_turns.size() && i < turns_cap; i++) per_turn[i] = g_asm_turns[i];
    if (n_lines) *n_lines = g_asm.size();
    if (n_turns) *n_turns = g_asm_turns.size();
}
static long ref_stitch_worker(const sdv_line_rec *recs, size_t n_recs, const sdv_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                               sdv_frame_asm *frames, size_t frames_cap, size_t *n_frames, sdv_block_rec *blocks, size_t blocks_cap, size_t *n_blocks)
{
    STC007DataStitcher *ds = new STC007DataStitcher();
    std::deque<STC007Line> in_q;
    std::deque<PCMSamplePair> out_q;
    QMutex in_mtx, out_mtx, fr_mtx;
    std::vector<FrameAsmSTC007> fr;
    ds->setInputPointers(&in_q, &in_mtx);
    ds->setOutputPointers(&out_q, &out_mtx);
    QObject::connect(ds, &STC007DataStitcher::guiUpdFrameAsm, [&](FrameAsmSTC007 d) { fr_mtx.lock(); fr.push_back(d); fr_mtx.unlock(); });
    size_t blocks_seen = 0;           /* newBlockProcessed: the block as outputDataBlock hands it to the visualiser (:6626), on the stitcher's thread */
    if (blocks) QObject::connect(ds, &STC007DataStitcher::newBlockProcessed, [&](STC007DataBlock b) { if (blocks_seen < blocks_cap) block_to_rec(b, &blocks[blocks_seen]); blocks_seen++; });
    if (blocks) {       /* ... and the assembled lines (newLineProcessed, :6696), kept for ref_stitch_last_asm_lines; a turn ends with its guiUpdFrameAsm */
        g_asm.clear(); g_asm_turns.clear(); g_asm_in_turn = 0;
        QObject::connect(ds, &STC007DataStitcher::newLineProcessed, [&](STC007Line l) {
            sdv_asm_line_rec r; memset(&r, 0, sizeof(r));
            r.frame_number = l.frame_number; r.line_number = l.line_number; r.calc_crc = l.getCalculatedCRC();
            for (int i = 0; i < 9; i++) { r.words[i] = l.getWord(i); if (l.isWordCRCOk(i)) r.word_crc_ok |= (uint16_t)(1u << i); if (l.isWordValid(i)) r.word_valid |= (uint16_t)(1u << i); }
            r.flags = (uint8_t)((l.isForcedBad() ? SDV_AL_FORCED_BAD : 0) | (l.hasMarkers() ? SDV_AL_MARKERS : 0) | (l.isCRCValid() ? SDV_AL_CRC_VALID : 0));
            g_asm.push_back(r); g_asm_in_turn++;
        });
        QObject::connect(ds, &STC007DataStitcher::guiUpdFrameAsm, [&](FrameAsmSTC007 d) { if (!d.isServNewFile() && !d.isServEndFile()) { g_asm_turns.push_back(g_asm_in_turn); g_asm_in_turn = 0; } });
    }
    ds->setVideoStandard(st->video_standard); ds->setFieldOrder(st->field_order);
    ds->setPCorrection(st->enable_p); ds->setQCorrection(st->enable_q); ds->setCWDCorrection(st->enable_cwd);
    ds->setM2SampleFormat(st->m2_format); ds->setResolutionPreset(st->resolution_preset); ds->setSampleRatePreset(st->sample_rate_preset);
    ds->setFineMaxUnch14(st->max_unch_14); ds->setFineMaxUnch16(st->max_unch_16); ds->setFineUseECC(st->use_ecc);
    ds->setFineMaskSeams(st->mask_seams); ds->setFineBrokeMask(st->broke_mask); ds->setFineTopLineFix(st->top_line_fix);
    std::thread th([ds]() { ds->doFrameReassemble(); });
    size_t fed = 0; long got = 0; bool overflow = false;
    size_t idle = 0;
    STC007Line l;
    while (true) {
        /* feed up to the reference's input queue limit, frame by frame */
        in_mtx.lock();
        size_t qs = in_q.size();
        while (fed < n_recs && qs < (size_t)(MAX_PCMLINE_QUEUE_SIZE - 1)) { rec_to_line(recs[fed], l); in_q.push_back(l); fed++; qs++; }
        in_mtx.unlock();
        out_mtx.lock();
        size_t drained = out_q.size();
        while (!out_q.empty()) {
            PCMSamplePair &p = out_q.front();
            if ((size_t)got < out_cap) {
                sdv_sample_pair *o = &out[got];
                memset(o, 0, sizeof(*o));
                for (int c = 0; c < 2; c++) {
                    o->audio_word[c] = p.samples[c].audio_word;
                    o->sample_flags[c] = (uint8_t)((p.samples[c].data_block_ok ? SDV_SF_BLOCK_OK : 0) | (p.samples[c].word_valid ? SDV_SF_WORD_VALID : 0) |
                                                   (p.samples[c].word_fixed ? SDV_SF_WORD_FIXED : 0) | (p.samples[c].word_masked ? SDV_SF_WORD_MASKED : 0));
                }
                o->sample_rate = p.sample_rate; o->emphasis = p.emphasis; o->service_type = p.service_type;
            } else overflow = true;
            got++;
            out_q.pop_front();
        }
        out_mtx.unlock();
        /* done when everything is fed and the stitcher has stopped making progress (it keeps < 2 frames queued) */
        in_mtx.lock(); qs = in_q.size(); in_mtx.unlock();
        fr_mtx.lock(); if (fr.size() > frames_cap + 64) overflow = true; fr_mtx.unlock();      /* ... or reports frames without end */
        if (overflow) break;        /* a stitcher that never lets go of a frame (a queue head it cannot pop) fills any buffer: report -1 instead of hanging */
        if (fed == n_recs && drained == 0) { idle++; if (idle > 150) break; } else idle = 0;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    ds->stop();
    th.join();
    size_t nf = fr.size() < frames_cap ? fr.size() : frames_cap;
    for (size_t i = 0; i < nf; i++) frasm_to_pod(fr[i], &frames[i]);
    if (n_frames) *n_frames = fr.size();
    if (n_blocks) *n_blocks = blocks_seen;
    delete ds;
    return overflow ? -1 : got;
}

extern "C" long ref_stitch_run(const sdv_line_rec *recs, size_t n_recs, const sdv_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                               sdv_frame_asm *frames, size_t frames_cap, size_t *n_frames)
{
    return ref_stitch_worker(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, NULL, 0, NULL);
}
extern "C" long ref_stitch_run_blocks(const sdv_line_rec *recs, size_t n_recs, const sdv_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                      sdv_frame_asm *frames, size_t frames_cap, size_t *n_frames, sdv_block_rec *blocks, size_t blocks_cap, size_t *n_blocks)
{
    return ref_stitch_worker(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, blocks, blocks_cap, n_blocks);
}

/* ------------------------------------------------------------------ PCM-1 back half: the real PCM1DataStitcher */
/* ---- PCM-1 front half: the real Binarizer with a PCM1Line as output ------------------------------------------------ */
struct RefBin1 {
    Binarizer bin;
    VideoLine vline;
    PCM1Line out;
};

static void p1_line_to_rec(PCM1Line &l, sdv_pcm1_bin_rec *r)
{
    memset(r, 0, sizeof(*r));
    r->frame_number = l.frame_number; r->line_number = l.line_number;
    for (int i = 0; i < 7; i++) r->words[i] = l.getWord(i);
    r->calc_crc = l.getCalculatedCRC();
    r->data_start = l.coords.data_start; r->data_stop = l.coords.data_stop;
    r->black_level = l.black_level; r->white_level = l.white_level;
    r->ref_low = l.ref_low; r->ref_level = l.ref_level; r->ref_high = l.ref_high;
    r->hysteresis_depth = l.hysteresis_depth; r->shift_stage = l.shift_stage;
    uint8_t st = SDV_SRV_NO;
    if (l.isServNewFile()) st = SDV_SRV_NEW_FILE; else if (l.isServEndFile()) st = SDV_SRV_END_FILE;
    else if (l.isServFiller()) st = SDV_SRV_FILLER; else if (l.isServEndField()) st = SDV_SRV_END_FIELD;
    else if (l.isServEndFrame()) st = SDV_SRV_END_FRAME; else if (l.isServHeader()) st = SDV_SRV_HEADER_LINE;
    r->service_type = st;
    r->picked_bits_left = l.picked_bits_left; r->picked_bits_right = l.picked_bits_right;
    r->flags = (uint8_t)((l.isDataByRefSweep() ? SDV_LF_REF_SWEEPED : 0) | (l.isDataByCoordSweep() ? SDV_LF_COORDS_SWEEPED : 0) |
                         (l.isDataBySkip() ? SDV_LF_BY_EXT_TUNE : 0) | (l.hasBWSet() ? SDV_LF_BW_SET : 0) |
                         (l.hasDataCoordSet() ? SDV_LF_COORDS_SET : 0) | (l.isForcedBad() ? SDV_LF_FORCED_BAD : 0) |
                         (l.isCRCValid() ? SDV_LF_CRC_VALID : 0) | (l.isSourceDoubleWidth() ? SDV_LF_FROM_DOUBLED : 0));
}

extern "C" {
void *ref_bin1_new(void) { return new RefBin1(); }
void ref_bin1_free(void *h) { delete (RefBin1 *)h; }
void ref_bin1_set_mode(void *h, int mode) { ((RefBin1 *)h)->bin.setMode((uint8_t)mode); }
void ref_bin1_set_coord_search(void *h, int on) { ((RefBin1 *)h)->bin.setCoordinatesSearch(on != 0); }
void ref_bin1_set_preset(void *h, const sdv_bin_preset *p)
{
    RefBin1 *r = (RefBin1 *)h;
    bin_preset_t s = r->bin.getDefaultFineSettings();
    s.max_black_lvl = p->max_black_lvl; s.min_white_lvl = p->min_white_lvl; s.min_contrast = p->min_contrast;
    s.min_ref_lvl = p->min_ref_lvl; s.max_ref_lvl = p->max_ref_lvl; s.min_valid_crcs = p->min_valid_crcs;
    s.mark_max_dist = p->mark_max_dist; s.left_bit_pick = p->left_bit_pick; s.right_bit_pick = p->right_bit_pick;
    s.en_force_coords = p->en_force_coords; s.en_coord_search = p->en_coord_search;
    s.en_first_line_dup = p->en_first_line_dup; s.en_good_no_marker = p->en_good_no_marker;
    s.horiz_coords.data_start = p->horiz_start; s.horiz_coords.data_stop = p->horiz_stop;
    r->bin.setFineSettings(s);
}
void ref_bin1_reset_good(void *h) { ((RefBin1 *)h)->bin.setGoodParameters(NULL); }
void ref_bin1_set_good_from_last(void *h) { RefBin1 *r = (RefBin1 *)h; r->bin.setGoodParameters(&r->out); }
void ref_bin1_set_state(void *h, const sdv_bin_state *s)
{
    RefBin1 *r = (RefBin1 *)h;
    r->bin.setReferenceLevel(s->in_def_reference);
    CoordinatePair c;
    c.data_start = s->in_def_start; c.data_stop = s->in_def_stop; c.from_doubled = s->in_def_from_doubled != 0;
    r->bin.setDataCoordinates(c);
    r->bin.setBWLevels(s->in_def_black, s->in_def_white);
}
int ref_bin1_scan_done(void *h) { return ((RefBin1 *)h)->vline.scan_done ? 1 : 0; }
int ref_bin1_process(void *h, const uint8_t *px, int len, uint32_t frame, uint16_t line, int service, int doubled, int empty,
                     sdv_pcm1_bin_rec *out)
{
    RefBin1 *r = (RefBin1 *)h;
    r->vline.clear();
    r->vline.frame_number = frame;
    r->vline.line_number = line;
    if (service == SDV_SRV_NO) {
        r->vline.setEmpty(empty != 0);
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
    p1_line_to_rec(r->out, out);
    return ret;
}
} /* extern "C" */

extern "C" uint16_t ref_pcm1_crc(const uint16_t *w6)
{
    PCM1Line l;
    for (uint8_t i = 0; i < 6; i++) l.setWord(i, w6[i]);
    l.calcCRC();
    return l.getCalculatedCRC();
}

static void rec_to_pcm1_line(const sdv_pcm1_line_rec &r, PCM1Line &l)
{
    l.clear();
    l.frame_number = r.frame_number; l.line_number = r.line_number;
    switch (r.service_type) {
        case SDV_SRV_NEW_FILE: l.setServNewFile("synthetic.avi"); return;
        case SDV_SRV_END_FILE: l.setServEndFile(); return;
        case SDV_SRV_FILLER: l.setServFiller(); return;
        case SDV_SRV_END_FIELD: l.setServEndField(); return;
        case SDV_SRV_END_FRAME: l.setServEndFrame(); return;
        case SDV_SRV_HEADER_LINE: l.setServHeader(); return;
        default: break;
    }
    for (uint8_t i = 0; i < 6; i++) l.setWord(i, r.words[i]);
    l.setSourceCRC(r.words[6]); l.calcCRC();
    l.ref_level = r.ref_level; l.picked_bits_left = r.picked_bits_left; l.picked_bits_right = r.picked_bits_right;
    l.setBWLevelsState((r.flags & SDV_LF_BW_SET) != 0);
    if (r.flags & SDV_LF_FORCED_BAD) l.setForcedBad();
}

static void frasm1_to_pod(FrameAsmPCM1 &f, sdv_frame_asm_pcm1 *o)
{
    memset(o, 0, sizeof(*o));
    o->frame_number = f.frame_number;
    o->odd_std_lines = f.odd_std_lines; o->even_std_lines = f.even_std_lines; o->odd_data_lines = f.odd_data_lines; o->even_data_lines = f.even_data_lines;
    o->odd_valid_lines = f.odd_valid_lines; o->even_valid_lines = f.even_valid_lines;
    o->odd_top_data = f.odd_top_data; o->odd_bottom_data = f.odd_bottom_data; o->even_top_data = f.even_top_data; o->even_bottom_data = f.even_bottom_data;
    o->odd_sample_rate = f.odd_sample_rate; o->even_sample_rate = f.even_sample_rate;
    o->blocks_total = f.blocks_total; o->blocks_drop = f.blocks_drop; o->samples_drop = f.samples_drop;
    o->odd_top_padding = f.odd_top_padding; o->odd_bottom_padding = f.odd_bottom_padding; o->even_top_padding = f.even_top_padding; o->even_bottom_padding = f.even_bottom_padding;
    o->blocks_fix_bp = f.blocks_fix_bp;
    o->field_order = f.field_order; o->odd_ref = f.odd_ref; o->even_ref = f.even_ref;
    o->service_type = f.isServNewFile() ? 1 : (f.isServEndFile() ? 2 : 0);
    o->flags = (uint8_t)((f.isOrderPreset() ? SDV_FA_ORDER_PRESET : 0) | (f.isOrderGuessed() ? SDV_FA_ORDER_GUESSED : 0) |
                         (f.odd_emphasis ? SDV_FA1_ODD_EMPHASIS : 0) | (f.even_emphasis ? SDV_FA1_EVEN_EMPHASIS : 0));
}

/* a PCM1DataBlock through its public interface (what the visualiser can ask of it) */
static void pcm1_block_to_rec(PCM1DataBlock &b, sdv_pcm1_block_rec *o)
{
    memset(o, 0, sizeof(*o));
    o->frame_number = b.frame_number; o->start_line = b.start_line; o->stop_line = b.stop_line; o->interleave_num = b.interleave_num;
    o->flags = (uint8_t)((b.isShortLength() ? SDV_P1B_SHORT : 0) | (b.hasEmphasis() ? SDV_P1B_EMPHASIS : 0));
    o->sample_rate = b.sample_rate;
    const bool was_short = b.isShortLength();
    b.setNormalLength();                   /* (getWord and the flags answer for words 182, 183 of a short block as well that way; a copy is worked on) */
    for (int i = 0; i < PCM1DataBlock::WORD_CNT; i++) {
        o->words[i] = b.getWord((uint8_t)i);
        o->word_flags[i] = (uint8_t)((b.isWordCRCOk((uint8_t)i) ? SDV_P1W_CRC_OK : 0) | (b.hasPickedSample((uint8_t)i) ? SDV_P1W_PICKED_LEFT : 0) |
                                     (b.hasPickedWord((uint8_t)i) ? SDV_P1W_PICKED_WORD : 0));
    }
    if (was_short) b.setShortLength();
}
static long ref_pcm1_stitch_run_impl(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                     sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm1_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                                     sdv_pcm1_asm_line_rec *vlines, size_t vlines_cap, size_t *n_vlines);
extern "C" long ref_pcm1_stitch_run(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                    sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames)
{
    return ref_pcm1_stitch_run_impl(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, NULL, 0, NULL, NULL, 0, NULL);
}
/* ... with what the stitcher hands to the visualiser: newBlockProcessed (16 blocks per frame) and newLineProcessed (the sub-lines of its queue that carry
 * the frame's number, in the order they are handed over: the engine's 1470 places per frame without the ones marked SDV_P1S_SKIP). */
extern "C" long ref_pcm1_stitch_run_vis(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                        sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm1_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                                        sdv_pcm1_asm_line_rec *vlines, size_t vlines_cap, size_t *n_vlines)
{
    return ref_pcm1_stitch_run_impl(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, blocks, blocks_cap, n_blocks, vlines, vlines_cap, n_vlines);
}
static long ref_pcm1_stitch_run_impl(const sdv_pcm1_line_rec *recs, size_t n_recs, const sdv_pcm1_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                     sdv_frame_asm_pcm1 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm1_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                                     sdv_pcm1_asm_line_rec *vlines, size_t vlines_cap, size_t *n_vlines)
{
    PCM1DataStitcher *ds = new PCM1DataStitcher();
    size_t blocks_seen = 0, vlines_seen = 0;
    if (blocks || vlines) {
        QObject::connect(ds, &PCM1DataStitcher::newBlockProcessed, [&](PCM1DataBlock b) { if (blocks && blocks_seen < blocks_cap) pcm1_block_to_rec(b, &blocks[blocks_seen]); blocks_seen++; });
        QObject::connect(ds, &PCM1DataStitcher::newLineProcessed, [&](PCM1SubLine l) {
            if (vlines && vlines_seen < vlines_cap) {
                sdv_pcm1_asm_line_rec *o = &vlines[vlines_seen];
                memset(o, 0, sizeof(*o));
                o->frame_number = l.frame_number; o->line_number = l.line_number; o->words[0] = l.getLeft(); o->words[1] = l.getRight();
                o->picked_bits_left = l.picked_bits_left; o->picked_bits_right = l.picked_bits_right; o->line_part = l.getLinePart();
                o->flags = (uint8_t)((l.hasBWSet() ? SDV_P1S_BW_SET : 0) | (l.isCRCValid() ? SDV_P1S_CRC_VALID : 0));
            }
            vlines_seen++;
        });
    }
    std::deque<PCM1Line> in_q;
    std::deque<PCMSamplePair> out_q;
    QMutex in_mtx, out_mtx, fr_mtx;
    std::vector<FrameAsmPCM1> fr;
    ds->setInputPointers(&in_q, &in_mtx);
    ds->setOutputPointers(&out_q, &out_mtx);
    QObject::connect(ds, &PCM1DataStitcher::guiUpdFrameAsm, [&](FrameAsmPCM1 d) { fr_mtx.lock(); fr.push_back(d); fr_mtx.unlock(); });
    ds->setFieldOrder(st->field_order); ds->setAutoLineOffset(st->auto_offset != 0);
    ds->setOddLineOffset(st->odd_offset); ds->setEvenLineOffset(st->even_offset); ds->setFineUseECC(st->use_ecc != 0);
    std::thread th([ds]() { ds->doFrameReassemble(); });
    size_t fed = 0; long got = 0; bool overflow = false;
    size_t idle = 0;
    PCM1Line l;
    while (true) {
        in_mtx.lock();
        size_t qs = in_q.size();
        while (fed < n_recs && qs < (size_t)(MAX_PCMLINE_QUEUE_SIZE - 1)) { rec_to_pcm1_line(recs[fed], l); in_q.push_back(l); fed++; qs++; }
        in_mtx.unlock();
        out_mtx.lock();
        size_t drained = out_q.size();
        while (!out_q.empty()) {
            PCMSamplePair &p = out_q.front();
            if ((size_t)got < out_cap) {
                sdv_sample_pair *o = &out[got];
                memset(o, 0, sizeof(*o));
                for (int c = 0; c < 2; c++) {
                    o->audio_word[c] = p.samples[c].audio_word;
                    o->sample_flags[c] = (uint8_t)((p.samples[c].data_block_ok ? SDV_SF_BLOCK_OK : 0) | (p.samples[c].word_valid ? SDV_SF_WORD_VALID : 0) |
                                                   (p.samples[c].word_fixed ? SDV_SF_WORD_FIXED : 0) | (p.samples[c].word_masked ? SDV_SF_WORD_MASKED : 0));
                }
                o->sample_rate = p.sample_rate; o->emphasis = p.emphasis; o->service_type = p.service_type;
            } else overflow = true;
            got++;
            out_q.pop_front();
        }
        out_mtx.unlock();
        fr_mtx.lock(); if (fr.size() > frames_cap + 64) overflow = true; fr_mtx.unlock();      /* ... or reports frames without end */
        if (overflow) break;        /* a stitcher that never lets go of a frame (a queue head it cannot pop) fills any buffer: report -1 instead of hanging */
        if (fed == n_recs && drained == 0) { idle++; if (idle > 150) break; } else idle = 0;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    ds->stop();
    th.join();
    size_t nf = fr.size() < frames_cap ? fr.size() : frames_cap;
    for (size_t i = 0; i < nf; i++) frasm1_to_pod(fr[i], &frames[i]);
    if (n_frames) *n_frames = fr.size();
    if (n_blocks) *n_blocks = blocks_seen;
    if (n_vlines) *n_vlines = vlines_seen;
    delete ds;
    return overflow ? -1 : got;
}

/* ---- PCM-1 frames through the real VideoToDigital worker (setPCMType(TYPE_PCM1)), PCM1Line queue drained ------------------- */
typedef RefV2DT<PCM1Line> RefV2D1;
extern "C" {
void *ref_v2d1_new(void)
{
    RefV2D1 *r = new RefV2D1();
    r->v2d.setInputPointers(&r->in_q, &r->in_mtx);
    r->v2d.setOutPCM1Pointers(&r->out_q, &r->out_mtx);
    r->v2d.setPCMType(VideoToDigital::TYPE_PCM1);
    QObject::connect(&r->v2d, &VideoToDigital::guiUpdFrameBin, [r](FrameBinDescriptor d) {
        r->stats_mtx.lock(); r->stats_q.push_back(d); r->stats_mtx.unlock();
    });
    return r;
}
void ref_v2d1_delete(void *h)
{
    RefV2D1 *r = (RefV2D1 *)h;
    if (r->started) { r->v2d.stop(); r->th.join(); }
    delete r;
}
void ref_v2d1_set_mode(void *h, int mode) { ((RefV2D1 *)h)->v2d.setBinarizationMode((uint8_t)mode); }
void ref_v2d1_set_check_line_dup(void *h, int on) { ((RefV2D1 *)h)->v2d.setCheckLineDup(on != 0); }
void ref_v2d1_set_preset(void *h, const sdv_bin_preset *p) { ((RefV2D1 *)h)->v2d.setFineSettings(to_bin_preset(p)); }
long ref_v2d1_run(void *h, const uint8_t *luma, size_t stride, int width, int height, int n_frames, uint32_t first_frame_no,
                  int new_file, int doubled, sdv_pcm1_bin_rec *out, sdv_frame_stats *stats)
{
    return v2d_run_t<PCM1Line, sdv_pcm1_bin_rec>((RefV2D1 *)h, luma, stride, width, height, n_frames, first_frame_no, new_file, doubled, out, stats, p1_line_to_rec, 1);
}
}

/* ---- PCM-16x0 front half: the real Binarizer with a PCM16X0SubLine as output, one pass per line part ----------------------- */
#include "pcm16x0subline.h"
struct RefBin16 {
    Binarizer bin;
    VideoLine vline;
    PCM16X0SubLine out;
};
static void p16_line_to_rec(PCM16X0SubLine &l, sdv_pcm16x0_bin_rec *r)
{
    memset(r, 0, sizeof(*r));
    r->frame_number = l.frame_number; r->line_number = l.line_number;
    for (int i = 0; i < 4; i++) r->words[i] = l.getWord(i);
    r->calc_crc = l.getCalculatedCRC();
    r->data_start = l.coords.data_start; r->data_stop = l.coords.data_stop;
    r->queue_order = l.queue_order;
    r->black_level = l.black_level; r->white_level = l.white_level;
    r->ref_low = l.ref_low; r->ref_level = l.ref_level; r->ref_high = l.ref_high;
    r->hysteresis_depth = l.hysteresis_depth; r->shift_stage = l.shift_stage;
    uint8_t st = SDV_SRV_NO;
    if (l.isServNewFile()) st = SDV_SRV_NEW_FILE; else if (l.isServEndFile()) st = SDV_SRV_END_FILE;
    else if (l.isServFiller()) st = SDV_SRV_FILLER; else if (l.isServEndField()) st = SDV_SRV_END_FIELD;
    else if (l.isServEndFrame()) st = SDV_SRV_END_FRAME;
    r->service_type = st;
    r->picked_bits_left = l.picked_bits_left; r->picked_bits_right = l.picked_bits_right;
    r->flags = (uint8_t)((l.isDataByRefSweep() ? SDV_LF_REF_SWEEPED : 0) | (l.isDataByCoordSweep() ? SDV_LF_COORDS_SWEEPED : 0) |
                         (l.isDataBySkip() ? SDV_LF_BY_EXT_TUNE : 0) | (l.hasBWSet() ? SDV_LF_BW_SET : 0) |
                         (l.hasDataCoordSet() ? SDV_LF_COORDS_SET : 0) | (l.isForcedBad() ? SDV_LF_FORCED_BAD : 0) |
                         (l.isCRCValid() ? SDV_LF_CRC_VALID : 0) | (l.isSourceDoubleWidth() ? SDV_LF_FROM_DOUBLED : 0));
    r->line_part = l.line_part; r->control_bit = l.control_bit ? 1 : 0;
}
extern "C" {
void *ref_bin16_new(void) { return new RefBin16(); }
void ref_bin16_free(void *h) { delete (RefBin16 *)h; }
void ref_bin16_set_mode(void *h, int mode) { ((RefBin16 *)h)->bin.setMode((uint8_t)mode); }
void ref_bin16_set_coord_search(void *h, int on) { ((RefBin16 *)h)->bin.setCoordinatesSearch(on != 0); }
void ref_bin16_set_preset(void *h, const sdv_bin_preset *p) { ((RefBin16 *)h)->bin.setFineSettings(to_bin_preset(p)); }
void ref_bin16_reset_good(void *h) { ((RefBin16 *)h)->bin.setGoodParameters(NULL); }
void ref_bin16_set_good_from_last(void *h) { RefBin16 *r = (RefBin16 *)h; r->bin.setGoodParameters(&r->out); }
void ref_bin16_set_state(void *h, const sdv_bin_state *s)
{
    RefBin16 *r = (RefBin16 *)h;
    r->bin.setReferenceLevel(s->in_def_reference);
    CoordinatePair c;
    c.data_start = s->in_def_start; c.data_stop = s->in_def_stop; c.from_doubled = s->in_def_from_doubled != 0;
    r->bin.setDataCoordinates(c);
    r->bin.setBWLevels(s->in_def_black, s->in_def_white);
}
int ref_bin16_scan_done(void *h) { return ((RefBin16 *)h)->vline.scan_done ? 1 : 0; }
int ref_bin16_process(void *h, const uint8_t *px, int len, uint32_t frame, uint16_t line, int service, int doubled, int empty,
                      int part, int new_line, sdv_pcm16x0_bin_rec *out)
{
    RefBin16 *r = (RefBin16 *)h;
    if (new_line) {
        r->vline.clear();
        r->vline.frame_number = frame;
        r->vline.line_number = line;
        if (service == SDV_SRV_NO) {
            r->vline.setEmpty(empty != 0);
            if (!empty) r->vline.pixel_data.assign(px, px + len);
            r->vline.setDoubleWidth(doubled != 0);
        } else if (service == SDV_SRV_NEW_FILE) r->vline.setServNewFile("synthetic");
        else if (service == SDV_SRV_END_FILE) r->vline.setServEndFile();
        else if (service == SDV_SRV_FILLER) r->vline.setServFiller();
        else if (service == SDV_SRV_END_FIELD) r->vline.setServEndField();
        else if (service == SDV_SRV_END_FRAME) r->vline.setServEndFrame();
    }
    r->bin.setSource(&r->vline);
    r->bin.setOutput(&r->out);
    r->bin.setLinePartMode((uint8_t)part);
    int ret = r->bin.processLine();
    p16_line_to_rec(r->out, out);
    return ret;
}
uint16_t ref_pcm16x0_crc(const uint16_t *w3)
{
    PCM16X0SubLine l;
    for (int i = 0; i < 3; i++) l.setWord(i, w3[i]);
    l.calcCRC();
    return l.getCalculatedCRC();
}
}

/* ---- PCM-16x0 frames through the real VideoToDigital worker (setPCMType(TYPE_PCM16X0)), PCM16X0SubLine queue drained -------- */
typedef RefV2DT<PCM16X0SubLine> RefV2D16;
extern "C" {
void *ref_v2d16_new(void)
{
    RefV2D16 *r = new RefV2D16();
    r->v2d.setInputPointers(&r->in_q, &r->in_mtx);
    r->v2d.setOutPCM16X0Pointers(&r->out_q, &r->out_mtx);
    r->v2d.setPCMType(VideoToDigital::TYPE_PCM16X0);
    QObject::connect(&r->v2d, &VideoToDigital::guiUpdFrameBin, [r](FrameBinDescriptor d) {
        r->stats_mtx.lock(); r->stats_q.push_back(d); r->stats_mtx.unlock();
    });
    return r;
}
void ref_v2d16_delete(void *h)
{
    RefV2D16 *r = (RefV2D16 *)h;
    if (r->started) { r->v2d.stop(); r->th.join(); }
    delete r;
}
void ref_v2d16_set_mode(void *h, int mode) { ((RefV2D16 *)h)->v2d.setBinarizationMode((uint8_t)mode); }
void ref_v2d16_set_check_line_dup(void *h, int on) { ((RefV2D16 *)h)->v2d.setCheckLineDup(on != 0); }
void ref_v2d16_set_preset(void *h, const sdv_bin_preset *p) { ((RefV2D16 *)h)->v2d.setFineSettings(to_bin_preset(p)); }
long ref_v2d16_run(void *h, const uint8_t *luma, size_t stride, int width, int height, int n_frames, uint32_t first_frame_no,
                   int new_file, int doubled, sdv_pcm16x0_bin_rec *out, sdv_frame_stats *stats)
{
    return v2d_run_t<PCM16X0SubLine, sdv_pcm16x0_bin_rec>((RefV2D16 *)h, luma, stride, width, height, n_frames, first_frame_no, new_file, doubled, out, stats, p16_line_to_rec, 3);
}
}

/* ---- PCM-16x0 back half: the real PCM16X0Deinterleaver and PCM16X0DataStitcher ------------------------------------------------ */
#include "pcm16.h"
static void rec_to_p16_line(const sdv_pcm16x0_bin_rec &r, PCM16X0SubLine &l)
{
    l.clear();
    l.frame_number = r.frame_number; l.line_number = r.line_number;
    switch (r.service_type) {
        case SDV_SRV_NEW_FILE: l.setServNewFile("synthetic.avi"); return;
        case SDV_SRV_END_FILE: l.setServEndFile(); return;
        case SDV_SRV_FILLER: l.setServFiller(); return;
        case SDV_SRV_END_FIELD: l.setServEndField(); return;
        case SDV_SRV_END_FRAME: l.setServEndFrame(); return;
        default: break;
    }
    for (uint8_t i = 0; i < 3; i++) l.setWord(i, r.words[i]);
    l.setSourceCRC(r.words[3]); l.calcCRC();
    l.coords.data_start = r.data_start; l.coords.data_stop = r.data_stop;
    l.ref_level = r.ref_level; l.picked_bits_left = r.picked_bits_left; l.picked_bits_right = r.picked_bits_right;
    l.control_bit = r.control_bit != 0; l.line_part = r.line_part; l.queue_order = r.queue_order;
    l.setBWLevelsState((r.flags & SDV_LF_BW_SET) != 0);
    if (r.flags & SDV_LF_FORCED_BAD) l.setForcedBad();
    /* what the stitcher does not look at but hands on to the visualiser with the sub-line (newLineProcessed): the rest of the binarizer's findings */
    l.black_level = r.black_level; l.white_level = r.white_level; l.ref_low = r.ref_low; l.ref_high = r.ref_high;
    l.hysteresis_depth = r.hysteresis_depth; l.shift_stage = r.shift_stage;
    l.setDataCoordinatesState((r.flags & SDV_LF_COORDS_SET) != 0);
    l.setSweepedReference((r.flags & SDV_LF_REF_SWEEPED) != 0); l.setSweepedCoordinates((r.flags & SDV_LF_COORDS_SWEEPED) != 0);
    l.data_by_ext_tune = (r.flags & SDV_LF_BY_EXT_TUNE) != 0;
    l.setFromDoubledState((r.flags & SDV_LF_FROM_DOUBLED) != 0);
}

extern "C" void ref_pcm16x0_deint_blocks(const sdv_pcm16x0_bin_rec *lines, size_t n_lines, int ei_format, int force_check, int p_code, int ignore_crc,
                                         int first_shift, int first_even, orc_p16_block_rec *out, size_t n_blocks)
{
    std::vector<PCM16X0SubLine> q(n_lines);
    for (size_t i = 0; i < n_lines; i++) rec_to_p16_line(lines[i], q[i]);
    PCM16X0Deinterleaver di;
    PCM16X0DataBlock b;
    di.setInput(&q); di.setOutput(&b);
    di.setForcedErrorCheck(force_check != 0); di.setPCorrection(p_code != 0); di.setIgnoreCRC(ignore_crc != 0);
    if (ei_format) di.setEIFormat(); else di.setSIFormat();
    bool even = first_even != 0;
    for (size_t k = 0; k < n_blocks; k++) {
        b.clear();
        uint8_t ret = di.processBlock((uint16_t)(first_shift + (int)k), even);
        orc_p16_block_rec *o = &out[k];
        memset(o, 0, sizeof(*o));
        o->frame_number = b.frame_number; o->start_line = b.start_line; o->stop_line = b.stop_line; o->queue_order = b.queue_order;
        o->start_part = b.start_part; o->stop_part = b.stop_part;
        for (uint8_t i = 0; i < 3; i++) {
            for (uint8_t w = 0; w < 3; w++) { o->words[i][w] = b.getWord(i, w); o->word_crc[i][w] = b.isWordCRCOk(i, w); o->word_valid[i][w] = b.isWordValid(i, w); }
            o->picked_left[i] = b.hasPickedLeft(i); o->picked_crc[i] = b.hasPickedCRC(i); o->audio_state[i] = b.getAudioState(i);
        }
        o->order_even = b.isOrderEven(); o->ret = ret;
        even = !even;
    }
}

static void frasm16_to_pod(FrameAsmPCM16x0 &f, sdv_frame_asm_pcm16x0 *o)
{
    memset(o, 0, sizeof(*o));
    o->frame_number = f.frame_number;
    o->odd_std_lines = f.odd_std_lines; o->even_std_lines = f.even_std_lines; o->odd_data_lines = f.odd_data_lines; o->even_data_lines = f.even_data_lines;
    o->odd_valid_lines = f.odd_valid_lines; o->even_valid_lines = f.even_valid_lines;
    o->odd_top_data = f.odd_top_data; o->odd_bottom_data = f.odd_bottom_data; o->even_top_data = f.even_top_data; o->even_bottom_data = f.even_bottom_data;
    o->odd_sample_rate = f.odd_sample_rate; o->even_sample_rate = f.even_sample_rate;
    o->blocks_total = f.blocks_total; o->blocks_drop = f.blocks_drop; o->samples_drop = f.samples_drop;
    o->odd_top_padding = f.odd_top_padding; o->odd_bottom_padding = f.odd_bottom_padding; o->even_top_padding = f.even_top_padding; o->even_bottom_padding = f.even_bottom_padding;
    o->blocks_broken = f.blocks_broken; o->blocks_fix_bp = f.blocks_fix_bp; o->blocks_fix_p = f.blocks_fix_p; o->blocks_fix_cwd = f.blocks_fix_cwd;
    o->field_order = f.field_order; o->odd_ref = f.odd_ref; o->even_ref = f.even_ref;
    o->service_type = f.isServNewFile() ? 1 : (f.isServEndFile() ? 2 : 0);
    o->flags = (uint8_t)((f.isOrderPreset() ? SDV_FA_ORDER_PRESET : 0) | (f.isOrderGuessed() ? SDV_FA_ORDER_GUESSED : 0) |
                         (f.odd_emphasis ? SDV_FA1_ODD_EMPHASIS : 0) | (f.even_emphasis ? SDV_FA1_EVEN_EMPHASIS : 0) |
                         (f.silence ? SDV_FA16_SILENCE : 0) | (f.padding_ok ? SDV_FA16_PADDING_OK : 0) | (f.ei_format ? SDV_FA16_EI_FORMAT : 0));
}

/* a PCM16X0DataBlock through its public interface, by line as the record holds it (getWordToLine, pcm16x0datablock.cpp:1029-1155, restated: the class
 * answers by word) */
static void pcm16_block_to_rec(PCM16X0DataBlock &b, sdv_pcm16x0_block_rec *o)
{
    memset(o, 0, sizeof(*o));
    const bool even = b.isOrderEven();
    for (uint8_t i = 0; i < 3; i++) {
        for (uint8_t w = 0; w < 3; w++) {
            const bool l_first = ((i & 1) != 0) != even;
            const int line = w == PCM16X0DataBlock::WORD_P ? 1 : (((w == PCM16X0DataBlock::WORD_L) == l_first) ? 0 : 2);
            o->words[i][line] = b.getWord(i, w);
            if (b.isWordCRCOk(i, w)) o->word_crc |= (uint16_t)(1u << (3 * i + line));
            if (b.isWordValid(i, w)) o->word_valid |= (uint16_t)(1u << (3 * i + line));
        }
        if (b.hasPickedLeft(i)) o->picked_left |= (uint8_t)(1u << i);
        if (b.hasPickedCRC(i)) o->picked_crc |= (uint8_t)(1u << i);
        o->audio_state[i] = b.getAudioState(i);
    }
    o->flags = (uint8_t)((even ? SDV_P16B_EVEN_ORDER : 0) | (b.isInEIFormat() ? SDV_P16B_EI_FORMAT : 0) | (b.hasEmphasis() ? SDV_P16B_EMPHASIS : 0) | (b.hasCode() ? SDV_P16B_CODE : 0));
    o->sample_rate = b.sample_rate;
}
/* hooks of the run for the visualiser's feed: every block the stitcher emits with newBlockProcessed, every frame it reports (no file tag) */
struct ref_p16_hooks { std::function<void(PCM16X0DataBlock &)> on_block; std::function<void(uint32_t)> on_frame; std::function<void(PCM16X0SubLine &)> on_line; };
long ref_pcm16x0_stitch_run_hooks(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                  sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, ref_p16_hooks *hooks);
extern "C" long ref_pcm16x0_stitch_run(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                       sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames)
{
    return ref_pcm16x0_stitch_run_hooks(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, NULL);
}
extern "C" long ref_pcm16x0_stitch_run_vis(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                           sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm16x0_block_rec *blocks, size_t blocks_cap, size_t *n_blocks)
{
    size_t seen = 0;
    ref_p16_hooks h;
    h.on_block = [&](PCM16X0DataBlock &b) { if (blocks && seen < blocks_cap) pcm16_block_to_rec(b, &blocks[seen]); seen++; };
    const long n = ref_pcm16x0_stitch_run_hooks(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, &h);
    if (n_blocks) *n_blocks = seen;
    return n;
}
/* ... with the assembled sub-lines newLineProcessed hands over (pcm16x0datastitcher.cpp:5203) as records, an END_FRAME record where MainWindow would emit
 * newFrameAssembled (mainwindow.cpp:3956): what sdv_set_pcm16x0_stitch_line_output writes */
extern "C" long ref_pcm16x0_stitch_run_feeds(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                             sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, sdv_pcm16x0_block_rec *blocks, size_t blocks_cap, size_t *n_blocks,
                                             sdv_pcm16x0_bin_rec *lines, size_t lines_cap, size_t *n_lines)
{
    size_t seen = 0, lseen = 0;
    ref_p16_hooks h;
    h.on_block = [&](PCM16X0DataBlock &b) { if (blocks && seen < blocks_cap) pcm16_block_to_rec(b, &blocks[seen]); seen++; };
    h.on_line = [&](PCM16X0SubLine &l) { if (lines && lseen < lines_cap) p16_line_to_rec(l, &lines[lseen]); lseen++; };
    h.on_frame = [&](uint32_t frame) {
        if (lines && lseen < lines_cap) {
            sdv_pcm16x0_bin_rec *r = &lines[lseen];
            memset(r, 0, sizeof(*r));
            r->frame_number = frame; r->words[3] = (uint16_t)~0x0E10; r->data_start = -32768; r->data_stop = 32767; r->control_bit = 1; r->service_type = SDV_SRV_END_FRAME;
        }
        lseen++;
    };
    const long n = ref_pcm16x0_stitch_run_hooks(recs, n_recs, st, out, out_cap, frames, frames_cap, n_frames, &h);
    if (n_blocks) *n_blocks = seen;
    if (n_lines) *n_lines = lseen;
    return n;
}
long ref_pcm16x0_stitch_run_hooks(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                  sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, ref_p16_hooks *hooks)
{
    PCM16X0DataStitcher *ds = new PCM16X0DataStitcher();
    if (hooks && hooks->on_block) QObject::connect(ds, &PCM16X0DataStitcher::newBlockProcessed, [hooks](PCM16X0DataBlock b) { hooks->on_block(b); });
    if (hooks && hooks->on_line) QObject::connect(ds, &PCM16X0DataStitcher::newLineProcessed, [hooks](PCM16X0SubLine l) { hooks->on_line(l); });
    if (hooks && hooks->on_frame) QObject::connect(ds, &PCM16X0DataStitcher::guiUpdFrameAsm, [hooks](FrameAsmPCM16x0 d) { if (!d.isServNewFile() && !d.isServEndFile()) hooks->on_frame(d.frame_number); });
    std::deque<PCM16X0SubLine> in_q;
    std::deque<PCMSamplePair> out_q;
    QMutex in_mtx, out_mtx, fr_mtx;
    std::vector<FrameAsmPCM16x0> fr;
    ds->setInputPointers(&in_q, &in_mtx);
    ds->setOutputPointers(&out_q, &out_mtx);
    QObject::connect(ds, &PCM16X0DataStitcher::guiUpdFrameAsm, [&](FrameAsmPCM16x0 d) { fr_mtx.lock(); fr.push_back(d); fr_mtx.unlock(); });
    ds->setFormat(st->format); ds->setFieldOrder(st->field_order); ds->setPCorrection(st->p_correction != 0);
    ds->setSampleRatePreset(st->sample_rate_preset);
    ds->setFineUseECC(st->use_ecc != 0); ds->setFineMaskSeams(st->mask_seams != 0); ds->setFineBrokeMask(st->broke_mask);
    std::thread th([ds]() { ds->doFrameReassemble(); });
    size_t fed = 0; long got = 0; bool overflow = false;
    size_t idle = 0;
    PCM16X0SubLine l;
    while (true) {
        in_mtx.lock();
        size_t qs = in_q.size();
        while (fed < n_recs && qs < (size_t)(MAX_PCMLINE_QUEUE_SIZE * 3 - 1)) { rec_to_p16_line(recs[fed], l); in_q.push_back(l); fed++; qs++; }
        in_mtx.unlock();
        out_mtx.lock();
        size_t drained = out_q.size();
        while (!out_q.empty()) {
            PCMSamplePair &p = out_q.front();
            if ((size_t)got < out_cap) {
                sdv_sample_pair *o = &out[got];
                memset(o, 0, sizeof(*o));
                for (int c = 0; c < 2; c++) {
                    o->audio_word[c] = p.samples[c].audio_word;
                    o->sample_flags[c] = (uint8_t)((p.samples[c].data_block_ok ? SDV_SF_BLOCK_OK : 0) | (p.samples[c].word_valid ? SDV_SF_WORD_VALID : 0) |
                                                   (p.samples[c].word_fixed ? SDV_SF_WORD_FIXED : 0) | (p.samples[c].word_masked ? SDV_SF_WORD_MASKED : 0));
                }
                o->sample_rate = p.sample_rate; o->emphasis = p.emphasis; o->service_type = p.service_type;
            } else overflow = true;
            got++;
            out_q.pop_front();
        }
        out_mtx.unlock();
        fr_mtx.lock(); if (fr.size() > frames_cap + 64) overflow = true; fr_mtx.unlock();      /* ... or reports frames without end */
        if (overflow) break;        /* a stitcher that never lets go of a frame (a queue head it cannot pop) fills any buffer: report -1 instead of hanging */
        if (fed == n_recs && drained == 0) { idle++; if (idle > 150) break; } else idle = 0;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    ds->stop();
    th.join();
    size_t nf = fr.size() < frames_cap ? fr.size() : frames_cap;
    for (size_t i = 0; i < nf; i++) frasm16_to_pod(fr[i], &frames[i]);
    if (n_frames) *n_frames = fr.size();
    delete ds;
    return overflow ? -1 : got;
}
