/*
 * v2d.c - CPU restatement of VideoToDigital::doBinarize (videotodigital.cpp:698-1815) for
 * STC-007: the per-frame driver around the per-line binarizer (feedback of good parameters,
 * duplicate-line detection, coordinate damper, frame statistics).
 * TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).
 *
 * The Qt worker loop, queues and sleeps are not restated; one call = one frame that
 * waitForOneFrame() (videotodigital.cpp:84-145) would have moved into frame_buf, in the line
 * order VideoInFFMPEG::spliceFrame produces (vin_ffmpeg.cpp:213-364).
 */
#include "sdv_oracle.h"
#include "v2d.h"
#include <string.h>
#include <stdlib.h>

void orc_line_to_rec(const orc_stc_line *l, sdv_line_rec *r);   /* api.c */

enum { FIELD_INIT = 0, FIELD_NEW, FIELD_SAFE, FIELD_UNSAFE };   /* videotodigital.h:90-96 */
enum { COORD_HISTORY_DEPTH = 9, COORD_LONG_HISTORY = 16, BIT_DIFF_THRES_DIV = 32 };   /* :99-110 */

/* ---- small deque<CoordinatePair> stand-in ---- */
static void cl_clear(orc_coord_list *l) { l->n = 0; }
static void cl_push(orc_coord_list *l, orc_coords c)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? l->cap * 2 : 64;
        l->v = (orc_coords *)realloc(l->v, (size_t)l->cap * sizeof(orc_coords));
    }
    l->v[l->n++] = c;
}
static void cl_pop_front(orc_coord_list *l) { memmove(l->v, l->v + 1, (size_t)(l->n - 1) * sizeof(orc_coords)); l->n--; }

static int cmp_coords(const void *a, const void *b)
{
    const orc_coords *x = (const orc_coords *)a, *y = (const orc_coords *)b;
    if (orc_coords_lt(x, y)) return -1;
    if (orc_coords_lt(y, x)) return 1;
    return 0;
}

/* videotodigital.cpp:348-371 medianCoordinates: nth_element at size/2 under CoordinatePair::operator<.
 * Elements that compare equal are interchangeable in every field that is read afterwards
 * (start, stop, reference), so a full sort returns the same value. */
static orc_coords median_coordinates(const orc_coord_list *l)
{
    orc_coords dummy; orc_coords_clear(&dummy);
    if (l->n > 0) {
        orc_coords *tmp = (orc_coords *)malloc((size_t)l->n * sizeof(orc_coords));
        memcpy(tmp, l->v, (size_t)l->n * sizeof(orc_coords));
        qsort(tmp, (size_t)l->n, sizeof(orc_coords), cmp_coords);
        dummy = tmp[l->n / 2];
        free(tmp);
    }
    return dummy;
}

static void stats_clear(orc_frame_stats *s)   /* frametrimset.cpp FrameBinDescriptor::clear */
{
    memset(s, 0, sizeof(*s));
    orc_coords_clear(&s->data_coord);
}

void orc_v2d_init(orc_v2d *v)   /* videotodigital.cpp:3-25 + doBinarize locals :727-732 */
{
    memset(v, 0, sizeof(*v));
    orc_binarizer_init(&v->line_converter);
    v->binarization_mode = ORC_MODE_NORMAL;
    v->coordinate_damper = true;
    v->check_line_copy = true;
    v->reset_stats = true;
    v->m2_format = false;
    orc_bin_preset_reset(&v->fine_bin_preset);
    v->line_converter.digi_set = v->fine_bin_preset;
    orc_coords_clear(&v->frame_avg);
    orc_coords_clear(&v->target_coord);
    stats_clear(&v->signal_quality);
    orc_stc_clear(&v->last_stc007_line);
    orc_stc_clear(&v->stc007_line);
    v->field_state = FIELD_INIT;
}

void orc_v2d_free(orc_v2d *v)
{
    free(v->last_valid_coord_list.v); free(v->frame_valid_coord_list.v);
    free(v->frame_invalid_coord_list.v); free(v->long_valid_coords.v);
    memset(v, 0, sizeof(*v));
}

void orc_v2d_set_fine_settings(orc_v2d *v, const orc_bin_preset *p)   /* :667-676 */
{
    v->reset_stats = true;
    v->fine_bin_preset = *p;
    v->line_converter.digi_set = *p;
}

/* :778-822 start-of-frame work */
void orc_v2d_begin_frame(orc_v2d *v)
{
    v->field_state = FIELD_NEW;
    v->good_coords_in_field = v->pcm_lines_in_field = 0;
    if (v->reset_stats) {
        v->reset_stats = false;
        cl_clear(&v->last_valid_coord_list); cl_clear(&v->frame_valid_coord_list);
        cl_clear(&v->frame_invalid_coord_list); cl_clear(&v->long_valid_coords);
        orc_coords_clear(&v->target_coord);
        orc_coords_clear(&v->frame_avg);
        orc_binarizer_set_good_parameters(&v->line_converter, NULL);
    }
    orc_coords_clear(&v->frame_avg);
    if (!v->fine_bin_preset.en_force_coords) {
        /* prescanCoordinates() returns at once for STC-007 (:175-200) */
        if (!orc_coords_valid(&v->frame_avg)) v->frame_avg = median_coordinates(&v->long_valid_coords);
        if (orc_coords_valid(&v->frame_avg))
            orc_binarizer_set_data_coordinates2(&v->line_converter, v->frame_avg.data_start, v->frame_avg.data_stop);
    }
}

/* :825-1717 body of the per-line loop for one VideoLine; emits exactly one record.
 * Returns true when the line was END_FRAME (stats filled). */
bool orc_v2d_line(orc_v2d *v, const orc_video_line *src, sdv_line_rec *out_rec, orc_frame_stats *out_stats)
{
    orc_binarizer *lc = &v->line_converter;
    orc_stc_line *wl = &v->stc007_line;
    bool even_line = ((src->line_number % 2) == 0);
    bool force_bad_line = false;
    bool frame_end = false;

    orc_binarizer_set_mode(lc, v->binarization_mode);
    lc->video_line = src;
    lc->line_part_mode = 0;
    lc->do_coord_search = true;
    lc->out_pcm_line = wl;
    orc_binarizer_process_line(lc);

    if (wl->service_type != ORC_SRV_NO) {                                   /* :1006-1114 */
        if (wl->service_type == ORC_SRV_NEW_FILE || wl->service_type == ORC_SRV_END_FILE) {
            v->line_in_field_cnt = 0;
            cl_clear(&v->last_valid_coord_list); cl_clear(&v->frame_valid_coord_list);
            cl_clear(&v->frame_invalid_coord_list); cl_clear(&v->long_valid_coords);
            orc_coords_clear(&v->target_coord);
            if (wl->service_type == ORC_SRV_END_FILE || !orc_coords_valid(&v->frame_avg))
                orc_binarizer_set_good_parameters(lc, NULL);
        } else if (wl->service_type == ORC_SRV_END_FIELD) {
            v->field_state = FIELD_NEW;
            v->line_in_field_cnt = 0;
            v->good_coords_in_field = 0; v->pcm_lines_in_field = 0;
            orc_stc_clear(&v->last_stc007_line);
        } else if (wl->service_type == ORC_SRV_CTRL_BLOCK) {
            if (v->field_state == FIELD_NEW) v->field_state = FIELD_SAFE;
        }
    } else {                                                                 /* :1115-1634 */
        bool count_has_data = orc_stc_has_markers(wl);
        bool count_has_pcm = orc_stc_crc_valid(wl) || count_has_data;
        wl->m2_format = v->m2_format;
        if (count_has_pcm && v->field_state == FIELD_NEW) v->field_state = FIELD_UNSAFE;
        if (orc_stc_crc_valid(wl) && force_bad_line) wl->forced_bad = true;
        if (orc_stc_crc_valid(wl)) {                                         /* :1182-1396 */
            v->good_coords_in_field++;
            v->signal_quality.line_length = src->length;
            if (v->check_line_copy) {
                if (v->field_state == FIELD_UNSAFE) {
                    orc_binarizer_set_good_parameters(lc, wl);
                    if (v->fine_bin_preset.en_first_line_dup) { wl->forced_bad = true; force_bad_line = true; }
                } else {
                    uint8_t bit_diff_cnt = orc_stc_words_diff_bit_count(wl, &v->last_stc007_line);
                    bool same_words = (bit_diff_cnt <= (ORC_STC_BITS_DATA / BIT_DIFF_THRES_DIV));
                    if (!orc_stc_is_almost_silent(wl) && same_words) {
                        wl->forced_bad = true;
                        if (!even_line) v->signal_quality.lines_dup_odd++; else v->signal_quality.lines_dup_even++;
                    }
                }
            }
            if (orc_stc_crc_valid_ignore_forced(wl)) {
                cl_push(&v->last_valid_coord_list, wl->coords);
                cl_push(&v->frame_valid_coord_list, wl->coords);
                while (v->last_valid_coord_list.n > COORD_HISTORY_DEPTH) cl_pop_front(&v->last_valid_coord_list);
                if (v->coordinate_damper && !v->fine_bin_preset.en_force_coords && (v->last_valid_coord_list.n > (COORD_HISTORY_DEPTH / 2))) {
                    v->target_coord = median_coordinates(&v->last_valid_coord_list);
                    if (!orc_coords_valid(&v->target_coord)) v->target_coord = v->frame_avg;
                    if (orc_coords_valid(&v->target_coord)) {
                        orc_coords d = wl->coords;
                        /* CoordinatePair::operator- (frametrimset.cpp:48-60), int16 wrap */
                        d.data_start = (int16_t)(d.data_start - v->target_coord.data_start);
                        d.data_stop = (int16_t)(d.data_stop - v->target_coord.data_stop);
                        /* getPPB() is uint8_t (pcmline.cpp:235-238); *3 passed through a uint8_t parameter */
                        uint8_t in_delta = (uint8_t)(((uint8_t)(wl->pixel_size_mult / ORC_INT_CALC_MULT)) * 3);
                        bool warn = (d.data_start <= -in_delta) || (d.data_start >= in_delta) || (d.data_stop <= -in_delta) || (d.data_stop >= in_delta);
                        if (warn) { wl->forced_bad = true; force_bad_line = true; }
                    }
                }
            }
            if (orc_stc_crc_valid(wl)) orc_binarizer_set_good_parameters(lc, wl);
            else { if (!even_line) v->signal_quality.lines_bad_odd++; else v->signal_quality.lines_bad_even++; }
            v->field_state = FIELD_INIT;
        } else {                                                             /* :1398-1523 */
            if (v->signal_quality.line_length == 0) v->signal_quality.line_length = src->length;
            if (orc_coords_valid(&wl->coords)) cl_push(&v->frame_invalid_coord_list, wl->coords);
            if (count_has_data) {
                orc_coords preset_coords; orc_coords_clear(&preset_coords);
                if (!even_line) v->signal_quality.lines_bad_odd++; else v->signal_quality.lines_bad_even++;
                if (!v->fine_bin_preset.en_force_coords) {
                    preset_coords = median_coordinates(&v->last_valid_coord_list);
                    if (!orc_coords_valid(&preset_coords)) preset_coords = v->frame_avg;
                }
                v->field_state = FIELD_INIT;
                orc_binarizer_set_data_coordinates(lc, preset_coords);
                orc_binarizer_set_bw_levels(lc, 0, 0);
            } else {
                orc_binarizer_set_bw_levels(lc, 0, 0);
            }
        }
        if (!even_line) v->signal_quality.lines_odd++; else v->signal_quality.lines_even++;
        if (count_has_pcm) {
            if (!even_line) v->signal_quality.lines_pcm_odd++; else v->signal_quality.lines_pcm_even++;
            v->pcm_lines_in_field++;
            v->last_stc007_line = *wl;
        }
        v->line_in_field_cnt++;
    }
    (void)force_bad_line;
    if (wl->service_type == ORC_SRV_END_FRAME) {                             /* :1636-1714 */
        orc_frame_stats *q = &v->signal_quality;
        if (q->lines_pcm_odd > q->lines_odd) q->lines_pcm_odd = q->lines_odd;
        if (q->lines_pcm_even > q->lines_even) q->lines_pcm_even = q->lines_even;
        if (q->lines_bad_odd > q->lines_odd) q->lines_bad_odd = q->lines_odd;
        if (q->lines_bad_even > q->lines_even) q->lines_bad_even = q->lines_even;
        q->frame_id = wl->frame_number;
        v->frame_avg = median_coordinates(&v->frame_valid_coord_list);
        if (orc_coords_valid(&v->frame_avg)) {
            q->data_coord = v->frame_avg;
            cl_push(&v->long_valid_coords, v->frame_avg);
            while (v->long_valid_coords.n > COORD_LONG_HISTORY) cl_pop_front(&v->long_valid_coords);
        } else {
            v->frame_avg = median_coordinates(&v->frame_invalid_coord_list);
            if (!orc_coords_valid(&v->frame_avg)) v->frame_avg = median_coordinates(&v->long_valid_coords);
            q->data_coord = v->frame_avg;
            q->data_coord.not_sure = true;
        }
        cl_clear(&v->frame_valid_coord_list); cl_clear(&v->frame_invalid_coord_list);
        if (out_stats) *out_stats = *q;
        stats_clear(q);
        frame_end = true;
    }
    orc_line_to_rec(wl, out_rec);                                            /* outNewLine :1717 */
    return frame_end;
}

/* One frame in spliceFrame() order (vin_ffmpeg.cpp:213-364): rows 0,2,4.. (line numbers 1,3,5..),
 * END_FIELD, rows 1,3,5.. (line numbers 2,4,6..), END_FIELD, END_FRAME.  If new_file, a NEW_FILE
 * service line (line_number 0) goes first (vin_ffmpeg.cpp:275-280, 525-565).
 * Returns the number of records written (height + 3 [+1]). */
int orc_v2d_frame(orc_v2d *v, const uint8_t *luma, size_t stride, int width, int height, uint32_t frame_no,
                  bool new_file, bool doubled, sdv_line_rec *out, orc_frame_stats *out_stats)
{
    int n = 0;
    orc_video_line vl;
    uint16_t line_num = 0;
    orc_v2d_begin_frame(v);
    memset(&vl, 0, sizeof(vl));
    vl.frame_number = frame_no;
    if (new_file) {
        vl.line_number = 0; vl.service_type = ORC_SRV_NEW_FILE; vl.empty = true; vl.pixels = NULL; vl.length = 0;
        orc_v2d_line(v, &vl, &out[n++], out_stats);
    }
    for (int field = 0; field < 2; field++) {
        int line_offset = field;
        line_num = (uint16_t)(line_offset + 1);
        for (;;) {
            vl.line_number = line_num; vl.service_type = ORC_SRV_NO; vl.empty = false; vl.doubled = doubled;
            vl.pixels = luma + (size_t)line_offset * stride; vl.length = (uint16_t)width;
            if (orc_g_empty_frame) { vl.empty = true; vl.pixels = NULL; vl.length = 0; }       /* a dropped frame: dummy_line.setLength(); setDoubleWidth(); setEmpty(true) (vin_ffmpeg.cpp:372-374, :428) */
            orc_v2d_line(v, &vl, &out[n++], out_stats);
            if (line_offset < (height - 2)) line_offset += 2;
            else { line_num = (uint16_t)(line_num + 2); break; }
            line_num = (uint16_t)(line_num + 2);
        }
        vl.line_number = line_num; vl.service_type = ORC_SRV_END_FIELD; vl.empty = true; vl.doubled = false; vl.pixels = NULL; vl.length = 0;
        orc_v2d_line(v, &vl, &out[n++], out_stats);
    }
    line_num = (uint16_t)(line_num + 2);
    vl.line_number = line_num; vl.service_type = ORC_SRV_END_FRAME;
    orc_v2d_line(v, &vl, &out[n++], out_stats);
    return n;
}

/* The frame VideoInFFMPEG::insertDummyFrame(true, false) (vin_ffmpeg.cpp:367-523) appends after the last frame of a file, as it
 * passes through the worker: height FILLER service lines in field order, END_FIELD after each field, then END_FILE and END_FRAME.
 * Returns the number of records written (height + 4). */
int orc_v2d_end_file_frame(orc_v2d *v, int height, uint32_t frame_no, sdv_line_rec *out, orc_frame_stats *out_stats)
{
    int n = 0;
    orc_video_line vl;
    uint16_t line_num = 0;
    orc_v2d_begin_frame(v);
    memset(&vl, 0, sizeof(vl));
    vl.frame_number = frame_no; vl.empty = true; vl.pixels = NULL; vl.length = 0;
    for (int field = 0; field < 2; field++) {
        int line_offset = field;
        for (;;) {
            line_num = (uint16_t)(line_offset + 1);
            vl.line_number = line_num; vl.service_type = ORC_SRV_FILLER;
            orc_v2d_line(v, &vl, &out[n++], out_stats);
            if (line_offset < (height - 2)) line_offset += 2;
            else { line_num = (uint16_t)(line_num + 2); break; }
        }
        vl.line_number = line_num; vl.service_type = ORC_SRV_END_FIELD;
        orc_v2d_line(v, &vl, &out[n++], out_stats);
    }
    line_num = (uint16_t)(line_num + 2);
    vl.line_number = line_num; vl.service_type = ORC_SRV_END_FILE;
    orc_v2d_line(v, &vl, &out[n++], out_stats);
    line_num = (uint16_t)(line_num + 2);
    vl.line_number = line_num; vl.service_type = ORC_SRV_END_FRAME;
    orc_v2d_line(v, &vl, &out[n++], out_stats);
    return n;
}
