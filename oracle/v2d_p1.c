/*
 * v2d_p1.c - CPU restatement of VideoToDigital::doBinarize (videotodigital.cpp:698-1815) and
 * VideoToDigital::prescanCoordinates (:148-345) for PCM-1: the per-frame driver around the PCM-1 line
 * binarizer of bin_pcm1.c (frame prescan of the data coordinates, feedback of good parameters,
 * coordinate-search switch of the real-time modes, Header lines, duplicate-line detection, coordinate
 * damper, frame statistics).
 * TEST INFRASTRUCTURE ONLY (see sdv_oracle.h).
 *
 * As in v2d.c the Qt worker loop, queues and sleeps are not restated; one call = one frame that
 * waitForOneFrame() (:84-145) would have moved into frame_buf, in VideoInFFMPEG::spliceFrame order
 * (vin_ffmpeg.cpp:213-364).
 */
#include "sdv_oracle.h"
#include "bin_pcm1.h"
#include "v2d_p1.h"
#include <string.h>
#include <stdlib.h>

enum { FIELD_INIT = 0, FIELD_NEW, FIELD_SAFE, FIELD_UNSAFE };   /* videotodigital.h:90-96 */
enum { COORD_CHECK_LINES = 4, COORD_CHECK_PARTS = COORD_CHECK_LINES + 2, COORD_HISTORY_DEPTH = 9, COORD_LONG_HISTORY = 16,
       BIT_DIFF_THRES_DIV = 32 };                                /* :99-110 */
enum { P1_BITS_PCM_DATA = 94, P1_LINES_PF = 245 };               /* pcm1line.h:71, pcm1datastitcher.h:103 */
enum { P1_BIT_RANGE_POS = 1 << 12, P1_BIT_SIGN_POS = 1 << 11 };  /* pcm1line.h:79-80 */

/* ---- deque<CoordinatePair> stand-in (as in v2d.c) ---- */
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
/* medianCoordinates (:348-371): nth_element at size/2; equal elements are interchangeable in what is read afterwards */
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
static void stats_clear(orc_frame_stats *s) { memset(s, 0, sizeof(*s)); orc_coords_clear(&s->data_coord); }

/* ---- PCM1Line helpers the frame driver uses ---- */
static int16_t p1_get_sample(const orc_p1_line *l, int index)   /* pcm1line.cpp:188-222 */
{
    uint16_t w = l->words[index];
    if ((w & P1_BIT_RANGE_POS) == 0) w = (uint16_t)(w << 4);
    else {
        bool pos = (w & P1_BIT_SIGN_POS) == 0;
        w = (uint16_t)(w & ~P1_BIT_RANGE_POS);
        w = (uint16_t)(w << 2);
        if (!pos) w |= (1 << 15) | (1 << 14);
    }
    return (int16_t)w;
}
static bool p1_is_almost_silent(const orc_p1_line *l)           /* :336-365: +-8 around zero, two or more of the six words */
{
    int n = 0;
    for (int i = 0; i < 6; i++) { int16_t s = p1_get_sample(l, i); if (!(s >= 8) && !(s < -8)) n++; }
    return n >= 2;
}
static uint8_t p1_words_diff_bit_count(const orc_p1_line *a, const orc_p1_line *b)   /* :236-263: the XOR is truncated to uint8_t */
{
    uint8_t cnt = 0;
    for (int i = 0; i < 6; i++) {
        uint8_t d = (uint8_t)(a->words[i] ^ b->words[i]);
        for (int bit = 0; bit < 8; bit++) if (d & (1 << bit)) cnt++;
    }
    return cnt;
}

void orc_v2d1_init(orc_v2d1 *v)   /* videotodigital.cpp:3-25 + doBinarize locals :727-732 */
{
    memset(v, 0, sizeof(*v));
    orc_binarizer_init(&v->line_converter);
    v->binarization_mode = ORC_MODE_NORMAL;
    v->coordinate_damper = true;
    v->check_line_copy = true;
    v->reset_stats = true;
    orc_bin_preset_reset(&v->fine_bin_preset);
    v->line_converter.digi_set = v->fine_bin_preset;
    orc_coords_clear(&v->frame_avg);
    orc_coords_clear(&v->target_coord);
    stats_clear(&v->signal_quality);
    orc_p1_clear(&v->pcm1_line);
    orc_p1_clear(&v->last_pcm1_line);
    v->field_state = FIELD_INIT;
    v->prescan_ref = 128;                /* :730 */
    /* setPCMType (:557-604): setGoodParameters() + reset_stats - the state the constructor leaves anyway */
}

void orc_v2d1_free(orc_v2d1 *v)
{
    free(v->last_valid_coord_list.v); free(v->frame_valid_coord_list.v);
    free(v->frame_invalid_coord_list.v); free(v->long_valid_coords.v);
    memset(v, 0, sizeof(*v));
}

void orc_v2d1_set_fine_settings(orc_v2d1 *v, const orc_bin_preset *p)   /* :667-676 */
{
    v->reset_stats = true;
    v->fine_bin_preset = *p;
    v->line_converter.digi_set = *p;
}

/* prescanCoordinates (:148-345): four lines spread over the frame buffer are decoded from scratch with the coordinate search on;
 * the median of the coordinates (and of the reference levels) of those that read valid is the frame's target */
static void prescan_coordinates(orc_v2d1 *v, const orc_video_line *frame_buf, int lines_cnt, orc_coords *out_coords, uint8_t *out_ref)
{
    orc_binarizer *lc = &v->line_converter;
    orc_p1_line pcm1_line;
    orc_coords coord_list[COORD_CHECK_LINES];
    uint8_t refs_list[COORD_CHECK_LINES];
    int n = 0;
    if (lines_cnt <= COORD_CHECK_PARTS || v->fine_bin_preset.en_force_coords) return;
    const uint16_t gap_length = (uint16_t)(lines_cnt / (COORD_CHECK_PARTS - 1));
    orc_binarizer_set_good_parameters_p1(lc, NULL);
    lc->do_coord_search = true;
    lc->line_part_mode = 0;
    for (int index = 0; index < COORD_CHECK_LINES; index++) {
        const orc_video_line *src = &frame_buf[(index + 1) * gap_length];
        if (src->service_type != ORC_SRV_NO) continue;
        lc->video_line = src;
        orc_binarizer_set_mode(lc, v->binarization_mode);
        orc_binarizer_process_line_p1(lc, &pcm1_line);
        if (orc_p1_crc_valid(&pcm1_line)) { coord_list[n] = pcm1_line.coords; refs_list[n] = pcm1_line.ref_level; n++; }
    }
    if (n > 0) {
        qsort(coord_list, (size_t)n, sizeof(orc_coords), cmp_coords);
        *out_coords = coord_list[n / 2];
        for (int i = 1; i < n; i++) { uint8_t x = refs_list[i]; int j = i; while (j > 0 && refs_list[j - 1] > x) { refs_list[j] = refs_list[j - 1]; j--; } refs_list[j] = x; }
        *out_ref = refs_list[n / 2];
    }
}

/* :772-822 start-of-frame work */
static void begin_frame(orc_v2d1 *v, const orc_video_line *frame_buf, int lines_cnt)
{
    orc_binarizer *lc = &v->line_converter;
    v->field_state = FIELD_NEW;
    v->good_coords_in_field = v->pcm_lines_in_field = 0;
    if (v->reset_stats) {
        v->reset_stats = false;
        cl_clear(&v->last_valid_coord_list); cl_clear(&v->frame_valid_coord_list);
        cl_clear(&v->frame_invalid_coord_list); cl_clear(&v->long_valid_coords);
        orc_coords_clear(&v->target_coord);
        orc_coords_clear(&v->frame_avg);
        orc_binarizer_set_good_parameters_p1(lc, NULL);
    }
    orc_coords_clear(&v->frame_avg);
    if (!v->fine_bin_preset.en_force_coords) {
        if (v->binarization_mode != ORC_MODE_DRAFT) prescan_coordinates(v, frame_buf, lines_cnt, &v->frame_avg, &v->prescan_ref);
        if (!orc_coords_valid(&v->frame_avg)) v->frame_avg = median_coordinates(&v->long_valid_coords);
        else orc_binarizer_set_reference_level(lc, v->prescan_ref);
        if (orc_coords_valid(&v->frame_avg)) orc_binarizer_set_data_coordinates2(lc, v->frame_avg.data_start, v->frame_avg.data_stop);
    }
}

void orc_p1_line_to_rec(const orc_p1_line *l, sdv_pcm1_bin_rec *r);   /* api.c */

/* :825-1717 body of the per-line loop for one VideoLine; emits exactly one record */
static void v2d1_line(orc_v2d1 *v, const orc_video_line *src, sdv_pcm1_bin_rec *out_rec, orc_frame_stats *out_stats)
{
    orc_binarizer *lc = &v->line_converter;
    orc_p1_line *wl = &v->pcm1_line;
    const bool even_line = ((src->line_number % 2) == 0);
    bool force_bad_line = false;

    orc_binarizer_set_mode(lc, v->binarization_mode);
    lc->video_line = src;
    lc->line_part_mode = 0;
    /* :853-884: the real-time modes stop searching once a field has produced enough good lines */
    if (v->binarization_mode == ORC_MODE_DRAFT || v->binarization_mode == ORC_MODE_FAST)
        lc->do_coord_search = !(v->good_coords_in_field > 2 || v->pcm_lines_in_field > 2);
    else lc->do_coord_search = true;
    orc_binarizer_process_line_p1(lc, wl);

    if (wl->service_type != ORC_SRV_NO) {                                   /* :1006-1114 */
        if (wl->service_type == ORC_SRV_NEW_FILE || wl->service_type == ORC_SRV_END_FILE) {
            v->line_in_field_cnt = 0;
            cl_clear(&v->last_valid_coord_list); cl_clear(&v->frame_valid_coord_list);
            cl_clear(&v->frame_invalid_coord_list); cl_clear(&v->long_valid_coords);
            orc_coords_clear(&v->target_coord);
            if (wl->service_type == ORC_SRV_END_FILE || !orc_coords_valid(&v->frame_avg))
                orc_binarizer_set_good_parameters_p1(lc, NULL);
        } else if (wl->service_type == ORC_SRV_END_FIELD) {
            v->field_state = FIELD_NEW;
            v->line_in_field_cnt = 0;
            v->good_coords_in_field = 0; v->pcm_lines_in_field = 0;
            orc_p1_clear(&v->last_pcm1_line);
        } else if (wl->service_type == ORC_SRV_HEADER_LINE) {               /* :1064-1088 */
            if (v->field_state == FIELD_NEW) v->field_state = FIELD_SAFE;
        }
    } else {                                                                 /* :1115-1634 */
        const bool count_has_data = wl->blk_wht_set;
        const bool count_has_pcm = orc_p1_crc_valid(wl) || count_has_data;
        if (count_has_pcm && v->field_state == FIELD_NEW) v->field_state = FIELD_UNSAFE;
        if (orc_p1_crc_valid(wl) && force_bad_line) wl->forced_bad = true;
        if (orc_p1_crc_valid(wl)) {                                          /* :1182-1396 */
            v->good_coords_in_field++;
            v->signal_quality.line_length = src->length;
            if (v->check_line_copy) {
                if (v->field_state == FIELD_UNSAFE) {
                    orc_binarizer_set_good_parameters_p1(lc, wl);
                    if (v->fine_bin_preset.en_first_line_dup) { wl->forced_bad = true; force_bad_line = true; }
                } else {
                    const uint8_t bit_diff_cnt = p1_words_diff_bit_count(wl, &v->last_pcm1_line);
                    const bool same_words = (bit_diff_cnt <= (P1_BITS_PCM_DATA / BIT_DIFF_THRES_DIV));
                    if (!p1_is_almost_silent(wl) && same_words) {
                        wl->forced_bad = true;
                        if (!even_line) v->signal_quality.lines_dup_odd++; else v->signal_quality.lines_dup_even++;
                    }
                }
            }
            if (orc_p1_crc_valid_ignore_forced(wl)) {
                cl_push(&v->last_valid_coord_list, wl->coords);
                cl_push(&v->frame_valid_coord_list, wl->coords);
                while (v->last_valid_coord_list.n > COORD_HISTORY_DEPTH) cl_pop_front(&v->last_valid_coord_list);
                if (v->coordinate_damper && !v->fine_bin_preset.en_force_coords && (v->last_valid_coord_list.n > (COORD_HISTORY_DEPTH / 2))) {
                    v->target_coord = median_coordinates(&v->last_valid_coord_list);
                    if (!orc_coords_valid(&v->target_coord)) v->target_coord = v->frame_avg;
                    if (orc_coords_valid(&v->target_coord)) {
                        orc_coords d = wl->coords;
                        d.data_start = (int16_t)(d.data_start - v->target_coord.data_start);     /* CoordinatePair::operator-, frametrimset.cpp:48-60 */
                        d.data_stop = (int16_t)(d.data_stop - v->target_coord.data_stop);
                        const uint8_t in_delta = (uint8_t)(((uint8_t)(wl->pixel_size_mult / ORC_INT_CALC_MULT)) * 3);    /* getPPB()*3 through uint8_t */
                        const bool warn = (d.data_start <= -in_delta) || (d.data_start >= in_delta) || (d.data_stop <= -in_delta) || (d.data_stop >= in_delta);
                        if (warn) { wl->forced_bad = true; force_bad_line = true; }
                    }
                }
            }
            if (orc_p1_crc_valid(wl)) orc_binarizer_set_good_parameters_p1(lc, wl);
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
            v->last_pcm1_line = *wl;
        }
        v->line_in_field_cnt++;
    }
    (void)force_bad_line;
    if (wl->service_type == ORC_SRV_END_FRAME) {                             /* :1636-1714 */
        orc_frame_stats *q = &v->signal_quality;
        q->lines_odd = q->lines_even = P1_LINES_PF;                          /* :1638-1642 */
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
    }
    orc_p1_line_to_rec(wl, out_rec);                                         /* outNewLine :1717 */
}

/* The frame buffer of one frame in spliceFrame order: [NEW_FILE,] rows 0,2,.. END_FIELD rows 1,3,.. END_FIELD END_FRAME; with
 * `filler` the frame VideoInFFMPEG::insertDummyFrame(true, false) appends to a file (vin_ffmpeg.cpp:367-523): FILLER lines,
 * END_FIELD twice, END_FILE, END_FRAME. */
static int build_frame_buf(orc_video_line *buf, const uint8_t *luma, size_t stride, int width, int height, uint32_t frame_no, bool new_file,
                           bool doubled, bool filler)
{
    int n = 0;
    uint16_t line_num = 0;
    orc_video_line vl;
    memset(&vl, 0, sizeof(vl));
    vl.frame_number = frame_no;
    if (new_file) { vl.line_number = 0; vl.service_type = ORC_SRV_NEW_FILE; vl.empty = true; buf[n++] = vl; }
    for (int field = 0; field < 2; field++) {
        int line_offset = field;
        line_num = (uint16_t)(line_offset + 1);
        for (;;) {
            vl.line_number = line_num;
            if (filler) { vl.service_type = ORC_SRV_FILLER; vl.empty = true; vl.doubled = false; vl.pixels = NULL; vl.length = 0; }
            else { vl.service_type = ORC_SRV_NO; vl.empty = false; vl.doubled = doubled; vl.pixels = luma + (size_t)line_offset * stride; vl.length = (uint16_t)width;
                   if (orc_g_empty_frame) { vl.empty = true; vl.pixels = NULL; vl.length = 0; } }      /* a dropped frame (vin_ffmpeg.cpp:372-374, :428) */
            buf[n++] = vl;
            if (line_offset < (height - 2)) line_offset += 2;
            else { line_num = (uint16_t)(line_num + 2); break; }
            line_num = (uint16_t)(line_num + 2);
        }
        vl.line_number = line_num; vl.service_type = ORC_SRV_END_FIELD; vl.empty = true; vl.doubled = false; vl.pixels = NULL; vl.length = 0;
        buf[n++] = vl;
    }
    if (filler) { line_num = (uint16_t)(line_num + 2); vl.line_number = line_num; vl.service_type = ORC_SRV_END_FILE; buf[n++] = vl; }
    line_num = (uint16_t)(line_num + 2);
    vl.line_number = line_num; vl.service_type = ORC_SRV_END_FRAME;
    buf[n++] = vl;
    return n;
}

/* One frame through the worker.  Returns the number of records written: height + 3 (+1 with new_file); the filler frame: height + 4. */
int orc_v2d1_frame(orc_v2d1 *v, const uint8_t *luma, size_t stride, int width, int height, uint32_t frame_no,
                   bool new_file, bool doubled, bool filler, sdv_pcm1_bin_rec *out, orc_frame_stats *out_stats)
{
    orc_video_line *buf = (orc_video_line *)malloc((size_t)(height + 8) * sizeof(orc_video_line));
    const int n = build_frame_buf(buf, luma, stride, width, height, frame_no, new_file, doubled, filler);
    begin_frame(v, buf, n);
    for (int i = 0; i < n; i++) v2d1_line(v, &buf[i], &out[i], out_stats);
    free(buf);
    return n;
}
