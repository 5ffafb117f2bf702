/*
 * pcm1_frames_device.h - PCM-1 frame driver: the PCM-1 branch of VideoToDigital::doBinarize (videotodigital.cpp:698-1815) and
 * VideoToDigital::prescanCoordinates (:148-345) around the PCM-1 line binarizer of pcm1_bin_device.h (SURVEY.md section 8 row
 * a11 for PCM-1).
 *
 * Two kernels per round:
 *   sdv_k_pcm1_prescan   one wave per (frame, prescan line): the four lines the worker decodes from scratch before every frame
 *                        (all modes but DRAFT) to find the frame's data coordinates and reference level.  The worker resets its
 *                        Binarizer first, so the result is a pure function of the pixels - all frames of a batch at once.
 *   sdv_k_pcm1_frames    one wave per frame, the pattern of the STC-007 frame kernel: the lines of a frame one after the other
 *                        in VideoInFFMPEG::spliceFrame order, every line decoded from what the lines before it left preset on the
 *                        Binarizer (setGoodParameters / setDataCoordinates / setBWLevels), with the per-line bookkeeping of the
 *                        worker (Header lines, duplicate-line detection, coordinate damper, statistics).  Frames run in parallel
 *                        from predicted incoming states; every frame checks the link to its successor itself (stc007_device.h,
 *                        v2d_store_state) and the host repeats the frames behind broken links (pcm1_frames_engine.inc).
 */
#pragma once
#include "pcm1_bin_device.h"

namespace sdvp1f {
using namespace sdv;
using namespace sdvp1b;

enum { COORD_CHECK_LINES = 4, COORD_CHECK_PARTS = COORD_CHECK_LINES + 2 };      /* videotodigital.h:101-102 */
enum { P1_LINES_PF = 245 };                                                     /* PCM1DataStitcher::LINES_PF, pcm1datastitcher.h:103 */

struct PrescanRes { int16_t start, stop; uint8_t ref, valid, pad[2]; };         /* one prescan line: the coordinates and level it read valid with */
/* pad[1]: the line found levels, i.e. it went through binarizer.cpp:1104 and left Binarizer::do_ref_lvl_sweep at "the mode is MODE_INSANE".
 * That member is the one thing a prescan line takes over from whatever the Binarizer did before (it enters the level validation, :3409),
 * and it only matters when min_valid_crcs > min_contrast: then every prescan line is decoded for both values of it (second half of the
 * array) and the frame picks, line after line, the variant its own flag calls for. */
__device__ __forceinline__ bool sweep_flag_matters(const sdv_bin_preset &ps) { return ps.min_valid_crcs > ps.min_contrast; }

struct FrameArgs1 {
    FrameArgs f;                    /* geometry, flags, states, stats, scratch as for STC-007 (f.recs unused) */
    sdv_pcm1_bin_rec *recs1;        /* frame k: recs1 + k*(height+3) (+1 behind the NEW_FILE frame) */
    PrescanRes *prescan;            /* [n_total][COORD_CHECK_LINES] */
    uint2 *frame_med;               /* [n_total] what frame f pushed into the multi-frame history: {coordinate key, 1} or {0, 0} (nothing) */
};

/* sdv_v2d_state::_pad[1] carries prescan_ref (videotodigital.cpp:703, 730: a local of the worker, 128 at its start) as its distance
 * from 128, so that an all-zero pad is the fresh worker for every format */
__device__ __forceinline__ uint8_t prescan_ref_of(const sdv_v2d_state &s) { return (uint8_t)(s._pad[1] ^ 128); }

/* lines in the frame buffer of frame f (waitForOneFrame, :84-145): the rows, END_FIELD twice, END_FRAME, the NEW_FILE line in
 * front of a file's first frame, END_FILE in the filler frame behind its last */
__device__ __forceinline__ int frame_buf_lines(const FrameArgs &a, int f) { return a.height + 3 + (f == a.new_file_frame ? 1 : 0) + (f == a.end_file_frame ? 1 : 0); }
/* does the worker prescan frame f? (prescanCoordinates :171-200, called in every mode but DRAFT, :796-801) */
__device__ __forceinline__ bool prescan_runs(const FrameArgs &a, int f)
{
    return !a.preset.en_force_coords && a.mode != SDV_MODE_DRAFT && frame_buf_lines(a, f) > COORD_CHECK_PARTS;
}
/* row of the picture at index i of frame f's buffer, or -1 for a service line */
__device__ inline int frame_buf_row(const FrameArgs &a, int f, int i)
{
    if (f == a.end_file_frame) return -1;                   /* FILLER lines carry no pixels */
    if (frame_is_empty(a, f)) return -1;                    /* nor do the lines of a dropped frame */
    if (f == a.new_file_frame) { if (i == 0) return -1; i--; }
    const int n0 = (a.height + 1) / 2, n1 = a.height / 2;
    if (i < n0) return 2 * i;
    i -= n0 + 1;
    if (i >= 0 && i < n1) return 2 * i + 1;
    return -1;
}

__device__ inline void ctx_for_line(const FrameArgs &a, BinCtx &c, Bin &b)
{
    c.ps = a.preset; c.mode = a.mode; c.scan_start = 0; c.scan_end = (uint16_t)(a.width - 1);
    c.force_bit_picker = true;      /* binarizer.cpp:82 */
    bin_set_mode(b, a.mode);
    b.scan_start = c.scan_start; b.scan_end = c.scan_end; b.vl_doubled = a.doubled != 0;
}

/* ---- prescan: block = (frame, k) ------------------------------------------------------------------------------------------ */
template <bool kInsane>
__device__ inline void prescan_body(const FrameArgs1 &a1, P1Lds &lds, int f, int k)
{
    const FrameArgs &a = a1.f;
    K1_BEGIN(); K1_T(tq0_);
    PrescanRes r; r.start = r.stop = 0; r.ref = 0; r.valid = 0; r.pad[0] = r.pad[1] = 0;
    if (prescan_runs(a, f)) {
        const int gap = frame_buf_lines(a, f) / (COORD_CHECK_PARTS - 1);
        const int row = frame_buf_row(a, f, (k + 1) * gap);
        if (row >= 0) {
            stage_row(lds.w.px, a.luma + (size_t)f * a.frame_stride + (size_t)row * a.row_stride, a.width);
            K1_T(tq1_); K1_ADD(16, tq0_, tq1_);
#pragma unroll 1
            for (int variant = 0; variant < (sweep_flag_matters(a.preset) ? 2 : 1); variant++) {
                BinCtx c; Bin b;
                b.in_black = b.in_white = b.in_ref = 0; coords_clear(b.in_coord);       /* setGoodParameters() (:221) */
                ctx_for_line(a, c, b);
                b.do_ref_lvl_sweep = variant != 0;
                L1 out;
                process_line_p1<kInsane>(c, b, true, lds, out, a.doubled != 0);
                PrescanRes q = r;
                if (crc_valid(out)) { q.start = out.coords.start; q.stop = out.coords.stop; q.ref = out.ref_level; q.valid = 1; }
                q.pad[1] = out.bw_set || out.service == SDV_SRV_HEADER_LINE ? 1 : 0;      /* (a Header line is only recognised behind the levels) */
                if (lane_id() == 0) a1.prescan[((size_t)variant * a.n_total + f) * COORD_CHECK_LINES + k] = q;
                SDV_WAVE_SYNC();
            }
            K1_T(tq2_); K1_ADD(23, tq0_, tq2_); K1_FLUSH();
            return;
        }
    }
    if (lane_id() == 0) {
        a1.prescan[(size_t)f * COORD_CHECK_LINES + k] = r;
        if (sweep_flag_matters(a.preset)) a1.prescan[((size_t)a.n_total + f) * COORD_CHECK_LINES + k] = r;
    }
}

/* ---- state of a PCM-1 frame wave ------------------------------------------------------------------------------------------ */
struct V2D1 {
    V2D v;                          /* the part shared with STC-007 (last_words: the six data words of last_pcm1_line) */
    uint8_t prescan_ref;
};

__device__ inline void v2d1_load_state(V2D1 &w, WaveLds &lds, const sdv_v2d_state *s, const FrameArgs &a)
{
    v2d_load_state(w.v, lds, s, a);
    w.prescan_ref = (uint8_t)uni(prescan_ref_of(*s));
}

/* Was frame f+1 started from what frame f handed on?  Like for STC-007 (byte for byte), with one difference: when the worker
 * prescans frame f+1 it resets its Binarizer first (prescanCoordinates :221), so what frame f left preset there does not reach
 * frame f+1 and does not count. */
__device__ inline bool link_holds1(const FrameArgs &a, int f, const sdv_v2d_state &out, sdv_v2d_state next_in)
{
    if (prescan_runs(a, f + 1)) next_in.bin = out.bin;
    uint32_t x[sizeof(sdv_v2d_state) / 4], y[sizeof(sdv_v2d_state) / 4];
    __builtin_memcpy(x, &out, sizeof(out));
    __builtin_memcpy(y, &next_in, sizeof(next_in));
    bool same = true;
    for (unsigned i = 0; i < sizeof(sdv_v2d_state) / 4; i++) same = same && (x[i] == y[i]);
    return same;
}

/* The chain after frame f; the link to frame f+1 is checked by the frame itself: when the worker prescans frame
 * f+1 it resets its Binarizer first (prescanCoordinates :221), so what frame f left preset there does not reach frame f+1 and does
 * not count. */
__device__ inline void v2d1_store_state(const V2D1 &w, const WaveLds &lds, sdv_v2d_state *s, const FrameArgs &a)
{
    if (lane_id() != 0) return;
    const V2D &v = w.v;
    sdv_v2d_state o;
    o.bin.in_def_black = v.bin.in_black; o.bin.in_def_white = v.bin.in_white; o.bin.in_def_reference = v.bin.in_ref; o.bin.do_ref_lvl_sweep = 0;
    o.bin.in_def_start = v.bin.in_coord.start; o.bin.in_def_stop = v.bin.in_coord.stop;
    o.bin.in_def_from_doubled = v.bin.in_coord.doubled ? 1 : 0; o.bin._pad2 = 0;
    o.do_ref_lvl_sweep = v.bin.do_ref_lvl_sweep ? 1 : 0; o.reset_stats = v.reset_stats ? 1 : 0;
    o.n_last_valid = (uint8_t)v.n_last; o.n_long_valid = (uint8_t)v.n_long;
    const uint16_t lm = a.doubled ? (uint16_t)((1u << v.n_last) - 1u) : 0, gm = a.doubled ? (uint16_t)((1u << v.n_long) - 1u) : 0;
    o.last_valid_doubled_mask_lo = (uint8_t)(lm & 0xFF); o.last_valid_doubled_mask_hi = (uint8_t)(lm >> 8);
    o.long_valid_doubled_mask = gm;
    for (int i = 0; i < COORD_HISTORY_DEPTH; i++) {
        if (i < v.n_last) { o.last_valid[i].data_start = key_start(lds.lv_keys[i]); o.last_valid[i].data_stop = key_stop(lds.lv_keys[i]); }
        else { o.last_valid[i].data_start = 0; o.last_valid[i].data_stop = 0; }
    }
    for (int i = 0; i < COORD_LONG_HISTORY; i++) {
        if (i < v.n_long) { o.long_valid[i].data_start = key_start(lds.long_keys[i]); o.long_valid[i].data_stop = key_stop(lds.long_keys[i]); }
        else { o.long_valid[i].data_start = 0; o.long_valid[i].data_stop = 0; }
    }
    o._pad[0] = 0; o._pad[1] = (uint8_t)(w.prescan_ref ^ 128);
    *s = o;
    const int f = (int)(s - a.states_out);
    a.flag[f] = (f + 1 < a.n_total && !link_holds1(a, f, o, a.states_in[f + 1])) ? VF_BREAK : VF_OK;
}

/* :772-822 start-of-frame work, with the prescan results of this frame */
__device__ inline void v2d1_begin_frame(V2D1 &w, const FrameArgs1 &a1, WaveLds &lds, int f)
{
    const FrameArgs &a = a1.f;
    V2D &v = w.v;
    v.field_state = FIELD_NEW;
    v.good_coords_in_field = v.pcm_lines_in_field = 0;
    if (v.reset_stats) {
        v.reset_stats = false;
        v.n_last = v.nfv = v.nfi = v.n_long = 0;
        coords_clear(v.frame_avg);
        bin_set_good_parameters_reset(v.bin, a.preset);
    }
    coords_clear(v.frame_avg);
    if (!a.preset.en_force_coords) {
        if (prescan_runs(a, f)) {
            bin_set_good_parameters_reset(v.bin, a.preset);                             /* :221 */
            /* the lines that read valid, sorted like std::sort under CoordinatePair::operator< / by level; the middle ones (:322-336) */
            uint32_t keys[COORD_CHECK_LINES]; uint8_t refs[COORD_CHECK_LINES]; int n = 0;
            for (int k = 0; k < COORD_CHECK_LINES; k++) {
                const int variant = (sweep_flag_matters(a.preset) && v.bin.do_ref_lvl_sweep) ? 1 : 0;
                const PrescanRes r = a1.prescan[((size_t)variant * a.n_total + f) * COORD_CHECK_LINES + k];
                if (uni(r.valid)) { keys[n] = uniu(coords_key(r.start, r.stop)); refs[n] = (uint8_t)uni(r.ref); n++; }
                if (uni(r.pad[1])) v.bin.do_ref_lvl_sweep = a.mode == SDV_MODE_INSANE;
            }
            for (int i = 1; i < n; i++)
                for (int j = i; j > 0; j--) {
                    if (keys[j - 1] > keys[j]) { const uint32_t t = keys[j]; keys[j] = keys[j - 1]; keys[j - 1] = t; }
                    if (refs[j - 1] > refs[j]) { const uint8_t t = refs[j]; refs[j] = refs[j - 1]; refs[j - 1] = t; }
                }
            if (n > 0) { v.frame_avg = key_to_coords(keys[n / 2], a.doubled != 0); w.prescan_ref = refs[n / 2]; }
        }
        if (!coords_valid(v.frame_avg)) { uint32_t k; if (median_keys(lds.long_keys, v.n_long, &k)) v.frame_avg = key_to_coords(k, a.doubled != 0); }
        else v.bin.in_ref = w.prescan_ref;                                              /* setReferenceLevel(prescan_ref) */
        if (coords_valid(v.frame_avg)) bin_set_data_coordinates2(v.bin, v.frame_avg.start, v.frame_avg.stop);
    }
}

__device__ inline void set_good_parameters_p1(Bin &b, const sdv_bin_preset &ps, const L1 &l)   /* binarizer.cpp:353-377 */
{
    if (crc_valid_ignore_forced(l)) { b.in_ref = l.ref_level; bin_set_data_coordinates(b, l.coords); bin_set_bw_levels(b, ps, l.black, l.white); }
}

/* service lines: Binarizer::processLine :539-568 + VideoToDigital :1006-1114 */
__device__ inline void v2d1_service_line(V2D1 &w, const FrameArgs &a, L1 &wl, uint8_t srv)
{
    V2D &v = w.v;
    p1_clear(wl);
    set_service(wl, srv);
    if (srv == SDV_SRV_NEW_FILE || srv == SDV_SRV_END_FILE) {
        v.line_in_field_cnt = 0;
        v.n_last = v.nfv = v.nfi = v.n_long = 0;
        if (srv == SDV_SRV_END_FILE || !coords_valid(v.frame_avg)) bin_set_good_parameters_reset(v.bin, a.preset);
    } else if (srv == SDV_SRV_END_FIELD) {
        v.field_state = FIELD_NEW;
        v.line_in_field_cnt = 0;
        v.good_coords_in_field = 0; v.pcm_lines_in_field = 0;
        for (int i = 0; i < 8; i++) v.last_words[i] = 0;           /* last_pcm1_line.clear(): the data words are zero */
    }
}

__device__ inline int16_t p1_get_sample(uint16_t w)     /* pcm1line.cpp:188-222 */
{
    if ((w & (1 << 12)) == 0) w = (uint16_t)(w << 4);
    else {
        const bool pos = (w & (1 << 11)) == 0;
        w = (uint16_t)(w & ~(1 << 12));
        w = (uint16_t)(w << 2);
        if (!pos) w |= (1 << 15) | (1 << 14);
    }
    return (int16_t)w;
}

/* regular line, after Binarizer::processLine: VideoToDigital :1115-1634 (PCM-1 branches) */
__device__ inline void v2d1_post_line(V2D1 &w, const FrameArgs &a, WaveLds &lds, L1 &wl, uint32_t *fv_keys, uint32_t *fi_keys, bool even_line)
{
    V2D &v = w.v;
    const sdv_bin_preset &ps = a.preset;
    if (wl.service != SDV_SRV_NO) {
        if (wl.service == SDV_SRV_HEADER_LINE && v.field_state == FIELD_NEW) v.field_state = FIELD_SAFE;     /* :1064-1088 */
        return;
    }
    const bool count_has_data = wl.bw_set;
    const bool count_has_pcm = crc_valid(wl) || count_has_data;
    if (count_has_pcm && v.field_state == FIELD_NEW) v.field_state = FIELD_UNSAFE;
    if (crc_valid(wl)) {
        v.good_coords_in_field++;
        v.q_line_length = (uint16_t)a.width;
        if (a.check_line_copy) {
            if (v.field_state == FIELD_UNSAFE) {
                set_good_parameters_p1(v.bin, ps, wl);
                if (ps.en_first_line_dup) wl.forced_bad = true;
            } else {
                int diff = 0, silent = 0;
                for (int i = 0; i < 6; i++) {
                    const uint16_t wd = get_word(wl, i);
                    diff += __popc((uint32_t)(uint8_t)(wd ^ v.last_words[i]));          /* the XOR is truncated to uint8_t (pcm1line.cpp:236-263) */
                    const int16_t smp = p1_get_sample(wd);
                    if (!(smp >= 8) && !(smp < -8)) silent++;
                }
                if (!(silent >= 2) && diff <= (P1_BITS / BIT_DIFF_THRES_DIV)) { wl.forced_bad = true; if (!even_line) v.q_dup_odd++; else v.q_dup_even++; }
            }
        }
        if (crc_valid_ignore_forced(wl)) {
            const uint32_t key = coords_key(wl.coords.start, wl.coords.stop);
            SDV_WAVE_SYNC();
            {   /* the window moves up by one when it is full: every lane carries one entry (no serial chain through LDS) */
                const int ln = lane_id();
                const bool full = v.n_last == COORD_HISTORY_DEPTH;
                const uint32_t moved = (full && ln < COORD_HISTORY_DEPTH - 1) ? lds.lv_keys[ln + 1] : 0u;
                SDV_WAVE_SYNC();
                if (full && ln < COORD_HISTORY_DEPTH - 1) lds.lv_keys[ln] = moved;
                if (ln == 0) lds.lv_keys[full ? COORD_HISTORY_DEPTH - 1 : v.n_last] = key;
            }
            if (v.n_last < COORD_HISTORY_DEPTH) v.n_last++;
            SDV_WAVE_SYNC();
            fv_keys[v.nfv++] = key;
            if (a.coordinate_damper && !ps.en_force_coords && (v.n_last > (COORD_HISTORY_DEPTH / 2))) {
                Coords target; coords_clear(target);
                uint32_t k;
                if (median_keys(lds.lv_keys, v.n_last, &k)) target = key_to_coords(k, false);
                if (!coords_valid(target)) target = v.frame_avg;
                if (coords_valid(target)) {
                    const int16_t ds = (int16_t)(wl.coords.start - target.start), de = (int16_t)(wl.coords.stop - target.stop);
                    const uint8_t in_delta = (uint8_t)(get_ppb(wl) * 3);
                    if (((int)ds <= -(int)in_delta) || ((int)ds >= (int)in_delta) || ((int)de <= -(int)in_delta) || ((int)de >= (int)in_delta)) wl.forced_bad = true;
                }
            }
        }
        if (crc_valid(wl)) set_good_parameters_p1(v.bin, ps, wl);
        else { if (!even_line) v.q_bad_odd++; else v.q_bad_even++; }
        v.field_state = FIELD_INIT;
    } else {
        if (v.q_line_length == 0) v.q_line_length = (uint16_t)a.width;
        if (coords_valid(wl.coords)) fi_keys[v.nfi++] = coords_key(wl.coords.start, wl.coords.stop);
        if (count_has_data) {
            Coords preset_coords; coords_clear(preset_coords);
            if (!even_line) v.q_bad_odd++; else v.q_bad_even++;
            if (!ps.en_force_coords) {
                uint32_t k;
                if (median_keys(lds.lv_keys, v.n_last, &k)) preset_coords = key_to_coords(k, a.doubled != 0);
                if (!coords_valid(preset_coords)) preset_coords = v.frame_avg;
            }
            v.field_state = FIELD_INIT;
            bin_set_data_coordinates(v.bin, preset_coords);
            bin_set_bw_levels(v.bin, ps, 0, 0);
        } else {
            bin_set_bw_levels(v.bin, ps, 0, 0);
        }
    }
    if (!even_line) v.q_odd++; else v.q_even++;
    if (count_has_pcm) {
        if (!even_line) v.q_pcm_odd++; else v.q_pcm_even++;
        v.pcm_lines_in_field++;
        for (int i = 0; i < 6; i++) v.last_words[i] = get_word(wl, i);
    }
    v.line_in_field_cnt++;
}

/* ---- the tape plays: lines that read from what their predecessor left preset ----------------------------------------------------------- */
#ifndef SDV_P1_BATCH
#define SDV_P1_BATCH 1              /* runs of lines that read from one tuning take batch1, single ones lean_line1 */
#endif
#ifndef SDV_P1_CAPTURE_D
#define SDV_P1_CAPTURE_D 8          /* batch1: lines in flight */
#endif
#ifdef SDV_EMU
__device__ __forceinline__ uint8_t lean_load_u8(const uint8_t *p) { return *p; }
#else
__device__ __forceinline__ uint8_t lean_load_u8(const uint8_t *p) { return __builtin_nontemporal_load(p); }      /* read once */
#endif
#ifdef SDV_EMU
__device__ inline uint32_t lane_read32(uint32_t x, uint32_t idx) { return (uint32_t)__shfl((int)x, (int)idx); }
#else
__device__ inline uint32_t lane_read32(uint32_t x, uint32_t idx) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)uniu(idx)); }
extern "C" __device__ uint32_t sdv_llvm_writelane1(uint32_t val, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");
#endif
__device__ __forceinline__ uint32_t park_lane(uint32_t old, uint32_t val, int lane)
{
#ifdef SDV_EMU
    return lane_id() == lane ? val : old;
#else
    return sdv_llvm_writelane1(val, (uint32_t)lane, old);
#endif
}
/* What depends on the tuning only: the pixels a lane samples (cells lane and lane + 64), the levels with their clipping test, the cells
 * the picture cuts off at either end (pickCutBitsUpPCM1, binarizer.cpp:6116-6596).  Kept while the tuning stays the same. */
struct Lean1 {
    uint32_t key_coords, key_levels;
    uint16_t x0, x1;
    uint8_t low, high, bits_l, bits_r;
    bool usable;
    uint32_t psm, hpsm; int16_t pso;
};
__device__ inline void lean1_reset(Lean1 &n) { n.key_levels = 0xFFFFFFFFu; n.key_coords = 0; n.usable = false; }
__device__ inline bool lean1_prepare(Lean1 &n, const BinCtx &c, const Bin &b)
{
    if (c.ps.en_force_coords) return false;
    if (!(are_bw_levels_preset(b, c.ps) && is_ref_level_preset(b, c.ps) && coords_valid(b.in_coord))) return false;
    if (!(b.in_ref < b.in_white && b.in_ref > b.in_black)) return false;
    if (!(c.scan_end > c.scan_start && P1_BITS <= (c.scan_end - c.scan_start))) return false;
    const uint32_t kc = coords_key(b.in_coord.start, b.in_coord.stop), kl = (uint32_t)b.in_black | ((uint32_t)b.in_white << 8) | ((uint32_t)b.in_ref << 16) | ((uint32_t)(b.in_coord.doubled ? 1 : 0) << 24);
    if (kc != n.key_coords || kl != n.key_levels) {
        n.key_coords = kc; n.key_levels = kl;
        L1 t; p1_clear(t);
        t.pixel_start = c.scan_start; t.pixel_stop = c.scan_end;
        t.coords = b.in_coord;
        set_ppb(t, t.coords);
        n.psm = t.psm; n.hpsm = t.hpsm; n.pso = t.pso;
        const int lane = lane_id();
        n.x0 = (uint16_t)pixel_of(t, lane, 0);
        n.x1 = (uint16_t)pixel_of(t, lane + 64 < P1_BITS ? lane + 64 : P1_BITS - 1, 0);
        n.low = get_low_level(b.in_ref, 0); n.high = get_high_level(b.in_ref, 0);
        n.usable = !(n.low <= b.in_black) && !(n.high >= b.in_white);
        int max_cut = c.ps.left_bit_pick; if (c.mode == SDV_MODE_DRAFT) max_cut /= 2;
        int first = c.scan_start, left_bits = 0, right_bits = 0;
        const int half_ppb = ((int)get_ppb(t) + 1) / 2;
        for (int i = 0; i < max_cut; i++) { const int cur = pixel_of(t, i, 0); if ((cur - first) >= half_ppb) break; if (i == 0) first = cur; left_bits = i + 1; }
        first = c.scan_end;
        max_cut = c.ps.right_bit_pick; if (c.mode == SDV_MODE_DRAFT) max_cut /= 2;
        for (int i = 0; i < max_cut; i++) { const int cur = pixel_of(t, P1_BITS - 1 - i, 0); if ((first - cur) >= half_ppb) break; if (i == 0) first = cur; right_bits = i + 1; }
        n.bits_l = (uint8_t)left_bits; n.bits_r = (uint8_t)right_bits;
    }
    return n.usable && c.force_bit_picker;
}
/* one read of a line under the tuning in n: the cells as two masks (cell b = bit b), the CRC over them; false when the CRC does not match
 * or the line is a Header (*header says which: a Header is read, all 94 cells of it - the first rung of the ladder takes it, pick_cut_bits
 * leaves a line alone that is valid, and a Header is valid whatever its CRC, pcm1line.cpp:314-323) */
__device__ inline bool lean_read1(const Lean1 &n, uint8_t p0, uint8_t p1, uint64_t &s_lo, uint64_t &s_hi, u128 &cells, uint16_t &crc, bool *header = nullptr)
{
    const int lane = lane_id();
    const bool second = lane + 64 < P1_BITS;
    const uint64_t a_lo = __ballot(p0 > n.low), b_lo = __ballot(p0 >= n.high);
    const uint64_t a_hi = __ballot(second && p1 > n.low), b_hi = __ballot(second && p1 >= n.high);
    solve_automaton(a_lo, a_hi, b_lo, b_hi, s_lo, s_hi);
    s_hi &= (1ull << (P1_BITS - 64)) - 1ull;
    cells = ((u128)__brevll(s_lo) << 30) | (u128)(__brevll(s_hi) >> 34);
    const uint64_t klo = (lane < 16) ? c_crc1.klo[lane & 15] : 0ull, khi = (lane < 16) ? c_crc1.khi[lane & 15] : 0ull;
    const int par = (__popcll(s_lo & klo) + __popcll(s_hi & khi)) & 1;
    crc = (uint16_t)((uint16_t)(__ballot(par) & 0xFFFF) ^ c_crc1.base);
    L1 t; t.v = cells;
    const bool hdr = has_header(t);
    if (header) *header = hdr;
    if (hdr) return false;
    return crc == (uint16_t)(cells & 0xFFFF);
}
__device__ inline void lean_fill_line1(const Lean1 &n, const FrameArgs &a, const Bin &b, u128 cells, uint16_t crc, L1 &out)
{
    p1_clear(out);
    out.pixel_start = 0; out.pixel_stop = (uint16_t)(a.width - 1);
    out.coords = b.in_coord;
    out.black = b.in_black; out.white = b.in_white; out.bw_set = true;
    out.ref_level = b.in_ref; out.ref_low = n.low; out.ref_high = n.high;
    out.hyst = 0; out.shift = 0;
    out.psm = n.psm; out.hpsm = n.hpsm; out.pso = n.pso;
    out.v = cells; out.calc_crc = crc;
    out.picked_l = n.bits_l; out.picked_r = n.bits_r;
    out.by_ext_tune = true;
}
/* Binarizer::processLine for a line that reads from its presets on the first rung of the ladder: stage STG_INPUT_ALL and nothing else */
__device__ inline bool lean_line1(Lean1 &n, const BinCtx &c, const FrameArgs &a, const Bin &b, const uint8_t *px_row, L1 &out)
{
    if (!lean1_prepare(n, c, b)) return false;
    uint64_t s_lo, s_hi; u128 cells; uint16_t crc; bool header = false;
    if (!lean_read1(n, px_row[n.x0], px_row[n.x1], s_lo, s_hi, cells, crc, &header)) {
        if (!header) return false;
        /* a Header line: read like any line (STG_INPUT_ALL, valid on the first rung), then made a service line (STG_DATA_OK, binarizer.cpp:1534-1621:
         * PCMLine::setServiceLine clears the base class's members - levels, coordinates, flags, the calculated CRC - and leaves the words and the
         * picked bits) */
        lean_fill_line1(n, a, b, cells, crc, out);
        set_service(out, SDV_SRV_HEADER_LINE);
        return true;
    }
    lean_fill_line1(n, a, b, cells, crc, out);
    return true;
}
/* A run of up to 64 lines that all read from the same presets (the pattern of batch16, pcm16_frames_device.h): phase A reads them - two
 * byte gathers per lane straight from the frame, four ballots, the automaton, the CRC - and parks each line's cells in the lane that
 * owns it; phase B does VideoToDigital's per-line bookkeeping (:1115-1634) a lane per line.  Returns the number of lines taken. */
__device__ inline int batch1(V2D1 &w, const FrameArgs &a, WaveLds &lds, const Lean1 &n, const uint8_t *frame, int field, int idx, int nl,
                             uint32_t frame_no, sdv_pcm1_bin_rec *rec, uint32_t *fv_keys, L1 &wl)
{
    V2D &v = w.v;
    const int lane = lane_id();
    if (v.field_state != FIELD_INIT) return 0;
    if ((uint8_t)((uint8_t)(n.psm / 128u) * 3) == 0) return 0;         /* in_delta = getPPB() * 3 as uint8_t: at 0 the damper flags a delta of zero */
    const uint32_t key = coords_key(v.bin.in_coord.start, v.bin.in_coord.stop);
    if (__ballot(lane < v.n_last && lds.lv_keys[lane < COORD_HISTORY_DEPTH ? lane : 0] != key) != 0ull) return 0;
    int n_lines = nl - idx; if (n_lines > 64) n_lines = 64;
    uint32_t m0 = 0, m1 = 0, m2 = 0;
    int n_ok = 0;
    {   /* Phase A, the pattern of the STC-007 capture (stc007_device.h): per line two byte gathers per lane straight from the frame (D lines in
         * flight: a line costs its share of the memory's bandwidth, not one trip to it), four ballots, and the masks parked in the lane that owns the
         * line; then every lane solves its own line - automaton and CRC.  A line that does not read ends the run: what was fetched behind it is
         * fetched again by whatever takes that line. */
        constexpr int D = SDV_P1_CAPTURE_D;
        static_assert(64 % D == 0, "whole groups of D lines make a run of up to 64");
        const uint32_t rs2 = 2u * (uint32_t)a.row_stride;
        const uint8_t *row0 = frame + (size_t)(2 * idx + field) * a.row_stride;
        const uint32_t x0 = n.x0, x1 = n.x1, lo = n.low, hi = n.high;
        const bool second = lane + 64 < P1_BITS;
        uint8_t q[D][2];
#pragma unroll
        for (int d = 0; d < D; d++) { const uint32_t o = (uint32_t)(d < n_lines ? d : n_lines - 1) * rs2; q[d][0] = lean_load_u8(row0 + (o + x0)); q[d][1] = lean_load_u8(row0 + (o + x1)); }
        uint32_t ra0 = 0, ra1 = 0, ra2 = 0, rb0 = 0, rb1 = 0, rb2 = 0;
        for (int j0 = 0; j0 < n_lines; j0 += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int j = j0 + d;
                const uint8_t p0 = q[d][0], p1 = q[d][1];
                const uint64_t a_lo = __ballot(p0 > lo), b_lo = __ballot(p0 >= hi), a_hi = __ballot(second && p1 > lo), b_hi = __ballot(second && p1 >= hi);
                { const int jn = j + D < n_lines ? j + D : n_lines - 1; const uint32_t o = (uint32_t)jn * rs2; q[d][0] = lean_load_u8(row0 + (o + x0)); q[d][1] = lean_load_u8(row0 + (o + x1)); }
                ra0 = park_lane(ra0, (uint32_t)a_lo, j); ra1 = park_lane(ra1, (uint32_t)(a_lo >> 32), j); ra2 = park_lane(ra2, (uint32_t)a_hi, j);
                rb0 = park_lane(rb0, (uint32_t)b_lo, j); rb1 = park_lane(rb1, (uint32_t)(b_lo >> 32), j); rb2 = park_lane(rb2, (uint32_t)b_hi, j);
            }
        }
        uint64_t s_lo, s_hi;
        solve_automaton_lane((uint64_t)ra0 | ((uint64_t)ra1 << 32), (uint64_t)ra2, (uint64_t)rb0 | ((uint64_t)rb1 << 32), (uint64_t)rb2, s_lo, s_hi);
        s_hi &= (1ull << (P1_BITS - 64)) - 1ull;
        const u128 cells = ((u128)__brevll(s_lo) << 30) | (u128)(__brevll(s_hi) >> 34);
        const uint16_t crc = crc_of_masks(s_lo, s_hi);
        L1 t; t.v = cells;
        const bool ok = crc == (uint16_t)(cells & 0xFFFF) && !has_header(t);
        const uint64_t okm = __ballot(ok || lane >= n_lines);
        n_ok = okm == ~0ull ? n_lines : (__ffsll((unsigned long long)~okm) - 1);
        m0 = (uint32_t)s_lo; m1 = (uint32_t)(s_lo >> 32); m2 = (uint32_t)s_hi;
    }
    if (n_ok == 0) return 0;
    /* phase B: lane j = line j of the run */
    const uint64_t my_lo = ((uint64_t)m1 << 32) | m0, my_hi = m2;
    const u128 mine = ((u128)__brevll(my_lo) << 30) | (u128)(__brevll(my_hi) >> 34);
    L1 t; t.v = mine;
    uint16_t wd[6];
#pragma unroll
    for (int k = 0; k < 6; k++) wd[k] = get_word(t, k);
    if (a.check_line_copy) {
        const int src = lane >= 1 ? lane - 1 : 0;
        const uint32_t p0 = (uint32_t)__shfl((int)m0, src), p1 = (uint32_t)__shfl((int)m1, src), p2 = (uint32_t)__shfl((int)m2, src);
        L1 ab; ab.v = ((u128)__brevll(((uint64_t)p1 << 32) | p0) << 30) | (u128)(__brevll((uint64_t)p2) >> 34);
        int diff = 0, silent = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const uint16_t above = lane == 0 ? v.last_words[k] : get_word(ab, k);
            diff += __popc((uint32_t)(uint8_t)(wd[k] ^ above));
            const int16_t smp = p1_get_sample(wd[k]);
            if (!(smp >= 8) && !(smp < -8)) silent++;
        }
        const uint64_t dup = __ballot(lane < n_ok && !(silent >= 2) && diff <= (P1_BITS / BIT_DIFF_THRES_DIV));
        if (dup) { n_ok = __ffsll((unsigned long long)dup) - 1; if (n_ok == 0) return 0; }
    }
    const bool even_line = field == 1;
    const uint8_t ref = v.bin.in_ref;
    if (lane < n_ok) {
        sdv_pcm1_bin_rec r;
        r.frame_number = frame_no; r.line_number = (uint16_t)(field + 1 + 2 * (idx + lane));
#pragma unroll
        for (int k = 0; k < 6; k++) r.words[k] = wd[k];
        r.words[6] = (uint16_t)(mine & 0xFFFF);
        r.calc_crc = (uint16_t)(mine & 0xFFFF);
        r.data_start = v.bin.in_coord.start; r.data_stop = v.bin.in_coord.stop;
        r.black_level = v.bin.in_black; r.white_level = v.bin.in_white; r.ref_low = n.low; r.ref_level = ref; r.ref_high = n.high;
        r.hysteresis_depth = 0; r.shift_stage = 0; r.service_type = SDV_SRV_NO;
        r.picked_bits_left = n.bits_l; r.picked_bits_right = n.bits_r;
        r.flags = (uint8_t)(SDV_LF_BY_EXT_TUNE | SDV_LF_BW_SET | SDV_LF_CRC_VALID | (a.doubled ? SDV_LF_FROM_DOUBLED : 0));
        r._pad[0] = r._pad[1] = r._pad[2] = 0;
        rec[lane] = r;
        fv_keys[v.nfv + lane] = key;
    }
    v.good_coords_in_field = (uint16_t)(v.good_coords_in_field + n_ok);
    v.q_line_length = (uint16_t)a.width;
    if (!even_line) { v.q_odd = (uint16_t)(v.q_odd + n_ok); v.q_pcm_odd = (uint16_t)(v.q_pcm_odd + n_ok); }
    else { v.q_even = (uint16_t)(v.q_even + n_ok); v.q_pcm_even = (uint16_t)(v.q_pcm_even + n_ok); }
    v.pcm_lines_in_field = (uint16_t)(v.pcm_lines_in_field + n_ok);
    v.line_in_field_cnt = (uint16_t)(v.line_in_field_cnt + n_ok);
    v.nfv += n_ok;
    {
        int fill_to = v.n_last + n_ok; if (fill_to > COORD_HISTORY_DEPTH) fill_to = COORD_HISTORY_DEPTH;
        SDV_WAVE_SYNC();
        if (lane >= v.n_last && lane < fill_to) lds.lv_keys[lane] = key;
        SDV_WAVE_SYNC();
        v.n_last = fill_to;
    }
    {   /* what the last line leaves behind */
        const uint64_t l_lo = ((uint64_t)lane_read32(m1, (uint32_t)(n_ok - 1)) << 32) | lane_read32(m0, (uint32_t)(n_ok - 1)), l_hi = lane_read32(m2, (uint32_t)(n_ok - 1));
        const u128 last = ((u128)__brevll(l_lo) << 30) | (u128)(__brevll(l_hi) >> 34);
        lean_fill_line1(n, a, v.bin, last, (uint16_t)(last & 0xFFFF), wl);
        for (int k = 0; k < 6; k++) v.last_words[k] = get_word(wl, k);
    }
    return n_ok;
}

/* one frame.  kLean: the build without Binarizer::processLine's search stages - a frame with a line the tuning it inherits does not read is given up
 * (its flag says VF_ABORTED) and decoded again by the full build (pcm1_frames_engine.inc); the pattern of the STC-007 frame kernels. */
template <bool kInsane, bool kLean = false>
__device__ inline void frame_body1(const FrameArgs1 &a1, P1Lds &lds, int f)
{
    const FrameArgs &a = a1.f;
    if (kLean && (f == a.end_file_frame || frame_is_empty(a, f))) {          /* frames without pixels: the full build's */
        if (lane_id() == 0) a.flag[f] = VF_ABORTED;
        return;
    }
    V2D1 w; L1 wl;
    v2d1_load_state(w, lds.w, &a.states_in[f], a);
    V2D &v = w.v;
    const uint32_t frame_no = a.first_frame_no + (uint32_t)f;
    const uint8_t *frame = a.luma + (size_t)f * a.frame_stride;
    uint32_t *fv_keys = a.scratch + (size_t)f * 2u * (size_t)a.height;
    uint32_t *fi_keys = fv_keys + a.height;
    size_t rec_base = (size_t)f * (size_t)(a.height + 3);
    if (a.new_file_frame >= 0 && f > a.new_file_frame) rec_base += 1;
    sdv_pcm1_bin_rec *rec = a1.recs1 + rec_base;
    const bool doubled = a.doubled != 0;

    v2d1_begin_frame(w, a1, lds.w, f);
    const bool empty_frame = frame_is_empty(a, f);
    Lean1 lean; lean1_reset(lean);
    const int n_field[2] = { (a.height + 1) / 2, a.height / 2 };
    uint16_t line_num = 0;
    if (f == a.new_file_frame) { v2d1_service_line(w, a, wl, SDV_SRV_NEW_FILE); emit_rec(wl, frame_no, 0, false, rec++); }
    for (int field = 0; field < 2; field++) {
        const int nl = n_field[field];
        for (int idx = 0; idx < nl; idx++) {
            line_num = (uint16_t)(field + 1 + 2 * idx);
            if (f == a.end_file_frame) {            /* VideoInFFMPEG::insertDummyFrame(true, false): FILLER lines (vin_ffmpeg.cpp:367-523) */
                v2d1_service_line(w, a, wl, SDV_SRV_FILLER);
                emit_rec(wl, frame_no, line_num, false, rec++);
                continue;
            }
            if (empty_frame) {                      /* a dropped frame: an empty VideoLine comes back as a cleared line (binarizer.cpp:1689-1700), its length counts as 0 */
                p1_clear(wl);
                const uint16_t ql = v.q_line_length;
                v2d1_post_line(w, a, lds.w, wl, fv_keys, fi_keys, (line_num % 2) == 0);
                v.q_line_length = ql;
                emit_rec(wl, frame_no, line_num, false, rec++);
                continue;
            }
            BinCtx c;
            ctx_for_line(a, c, v.bin);
#if SDV_P1_BATCH
            if (lean1_prepare(lean, c, v.bin)) {
                const int took = batch1(w, a, lds.w, lean, frame, field, idx, nl, frame_no, rec, fv_keys, wl);
                if (took > 0) { rec += took; idx += took - 1; continue; }
            }
#endif
            stage_row(lds.w.px, frame + (size_t)(2 * idx + field) * a.row_stride, a.width);
            /* :853-884: the real-time modes stop searching once a field has produced enough good lines */
            bool coord_search = true;
            if (a.mode == SDV_MODE_DRAFT || a.mode == SDV_MODE_FAST) coord_search = !(v.good_coords_in_field > 2 || v.pcm_lines_in_field > 2);
            if (!(SDV_P1_BATCH && lean_line1(lean, c, a, v.bin, lds.w.px, wl))) {
                if (kLean) { if (!input_all_p1(c, v.bin, lds.w, wl, doubled)) { if (lane_id() == 0) a.flag[f] = VF_ABORTED; return; } }
                else process_line_p1<kInsane>(c, v.bin, coord_search, lds, wl, doubled);
            }
            v2d1_post_line(w, a, lds.w, wl, fv_keys, fi_keys, (line_num % 2) == 0);
            emit_rec(wl, frame_no, line_num, doubled, rec++);
        }
        line_num = (uint16_t)(field + 1 + 2 * nl);
        v2d1_service_line(w, a, wl, SDV_SRV_END_FIELD);
        emit_rec(wl, frame_no, line_num, false, rec++);
    }
    if (f == a.end_file_frame) {
        line_num = (uint16_t)(line_num + 2);
        v2d1_service_line(w, a, wl, SDV_SRV_END_FILE);
        emit_rec(wl, frame_no, line_num, false, rec++);
    }
    line_num = (uint16_t)(line_num + 2);
    v2d1_service_line(w, a, wl, SDV_SRV_END_FRAME);
    v.q_odd = v.q_even = P1_LINES_PF;                               /* :1638-1642 */
    {   /* the median of the frame's valid coordinates is what v2d_end_frame pushes into long_valid_coords (:1668-1682) */
        __syncthreads();        /* (keys stored to global memory by other lanes than the ones that read them now) */
        uint32_t mk = 0; bool pushed = median_keys(fv_keys, v.nfv, &mk);
        if (pushed) { const Coords mc = key_to_coords(mk, false); pushed = coords_valid(mc); }
        if (lane_id() == 0) { uint2 m; m.x = pushed ? mk : 0u; m.y = pushed ? 1u : 0u; a1.frame_med[f] = m; }
    }
    v2d_end_frame(v, a, lds.w, frame_no, fv_keys, fi_keys, &a.stats[f]);
    emit_rec(wl, frame_no, line_num, false, rec++);
    v2d1_store_state(w, lds.w, &a.states_out[f], a);
}

/* ---- prediction of the incoming states ------------------------------------------------------------------------------------ */
/* states[k] for the frames behind an anchor (first_of as for STC-007, engine.inc): what the worker carries from frame to frame is the
 * coordinate history - the last nine valid lines, the medians of the last sixteen frames - and prescan_ref.  On a tape that plays
 * every line of a frame reads with the coordinates its prescan found, so all of that follows from the prescan results, which are
 * known before any frame is decoded.  DRAFT mode has no prescan: there the state is handed on as it is, like for STC-007. */
struct PredictArgs1 { sdv_v2d_state *states; const PrescanRes *prescan; int first, hi; const int *first_of; FrameArgs f; };

/* the median of a state's window of last valid coordinates (videotodigital.cpp:348-371), or false when it is empty */
__device__ inline bool last_valid_median1(const sdv_v2d_state &s0, sdv_coord *out)
{
    const int n = s0.n_last_valid > COORD_HISTORY_DEPTH ? COORD_HISTORY_DEPTH : s0.n_last_valid;
    if (n == 0 || s0.reset_stats) return false;
    uint32_t keys[COORD_HISTORY_DEPTH];
    for (int i = 0; i < n; i++) keys[i] = coords_key(s0.last_valid[i].data_start, s0.last_valid[i].data_stop);
    for (int i = 1; i < n; i++) { const uint32_t x = keys[i]; int j = i; while (j > 0 && keys[j - 1] > x) { keys[j] = keys[j - 1]; j--; } keys[j] = x; }
    out->data_start = key_start(keys[n / 2]); out->data_stop = key_stop(keys[n / 2]);
    return true;
}
/* sticky = the frames in between are taken to decode with the coordinates the stream already carries (the median of the window of
 * last valid coordinates) instead of the ones their own prescan finds: what happens on a tape without Header lines, where the first
 * line of a field is marked bad (:1193-1211) and the worker falls back on its history for the lines behind it (:1431-1451) */
__device__ inline sdv_v2d_state predict_state1(const PredictArgs1 &a, int k, int base, bool sticky = false)
{
    const sdv_v2d_state s0 = a.states[base];
    sdv_v2d_state p = s0;
    sdv_coord carried; carried.data_start = 0; carried.data_stop = 0;
    const bool use_carried = sticky && last_valid_median1(s0, &carried);
    const uint8_t dbl = a.f.doubled;
    int n_long = s0.reset_stats ? 0 : s0.n_long_valid;       /* a worker that starts over clears its histories first (:778-790) */
    sdv_coord lg[COORD_LONG_HISTORY];
    for (int i = 0; i < COORD_LONG_HISTORY; i++) lg[i] = s0.long_valid[i];
    bool touched = false;
    sdv_coord last; last.data_start = 0; last.data_stop = 0;
    uint8_t pref = prescan_ref_of(s0);
    /* only the last sixteen frames in between can still be seen in the history */
    int j0 = base; if (k - j0 > COORD_LONG_HISTORY + 1) j0 = k - (COORD_LONG_HISTORY + 1);
    for (int j = j0; j < k; j++) {
        if (!prescan_runs(a.f, j)) continue;
        uint32_t keys[COORD_CHECK_LINES]; uint8_t refs[COORD_CHECK_LINES]; int n = 0;
        for (int q = 0; q < COORD_CHECK_LINES; q++) {
            const PrescanRes r = a.prescan[(size_t)j * COORD_CHECK_LINES + q];
            if (r.valid) { keys[n] = coords_key(r.start, r.stop); refs[n] = r.ref; n++; }
            if (r.pad[1]) p.do_ref_lvl_sweep = a.f.mode == SDV_MODE_INSANE ? 1 : 0;
        }
        if (n == 0) continue;
        for (int i = 1; i < n; i++)
            for (int q = i; q > 0; q--) {
                if (keys[q - 1] > keys[q]) { const uint32_t t = keys[q]; keys[q] = keys[q - 1]; keys[q - 1] = t; }
                if (refs[q - 1] > refs[q]) { const uint8_t t = refs[q]; refs[q] = refs[q - 1]; refs[q - 1] = t; }
            }
        last.data_start = key_start(keys[n / 2]); last.data_stop = key_stop(keys[n / 2]);
        if (use_carried) last = carried;
        pref = refs[n / 2];
        touched = true;
        if (n_long == COORD_LONG_HISTORY) { for (int i = 0; i + 1 < COORD_LONG_HISTORY; i++) lg[i] = lg[i + 1]; n_long--; }
        lg[n_long++] = last;
    }
    if (touched) {
        p.reset_stats = 0;
        p.n_last_valid = COORD_HISTORY_DEPTH;
        for (int i = 0; i < COORD_HISTORY_DEPTH; i++) p.last_valid[i] = last;
        p.n_long_valid = (uint8_t)n_long;
        for (int i = 0; i < COORD_LONG_HISTORY; i++) { if (i < n_long) p.long_valid[i] = lg[i]; else { p.long_valid[i].data_start = 0; p.long_valid[i].data_stop = 0; } }
        const uint16_t lm = dbl ? (uint16_t)((1u << COORD_HISTORY_DEPTH) - 1u) : 0, gm = dbl ? (uint16_t)((1u << n_long) - 1u) : 0;
        p.last_valid_doubled_mask_lo = (uint8_t)(lm & 0xFF); p.last_valid_doubled_mask_hi = (uint8_t)(lm >> 8);
        p.long_valid_doubled_mask = gm;
        p._pad[1] = (uint8_t)(pref ^ 128);
        p.bin.in_def_start = last.data_start; p.bin.in_def_stop = last.data_stop; p.bin.in_def_from_doubled = dbl;
    } else if (!touched && !s0.reset_stats && a.f.mode == SDV_MODE_DRAFT) {
        /* DRAFT: the tuning is handed on; a frame that plays fills the histories with the pair it inherited (the STC-007 model) */
        const int16_t cs = s0.bin.in_def_start, ce = s0.bin.in_def_stop;
        if (s0.bin.in_def_reference >= a.f.preset.min_ref_lvl && (cs != NO_COORD_LEFT && ce != NO_COORD_RIGHT && cs < ce)) {
            const int m = k - base;
            p.bin.in_def_from_doubled = dbl;
            p.n_last_valid = COORD_HISTORY_DEPTH;
            for (int i = 0; i < COORD_HISTORY_DEPTH; i++) { p.last_valid[i].data_start = cs; p.last_valid[i].data_stop = ce; }
            const int total = (int)s0.n_long_valid + m;
            const int keep = total > COORD_LONG_HISTORY ? COORD_LONG_HISTORY : total, drop = total - keep;
            for (int i = 0; i < COORD_LONG_HISTORY; i++) {
                const int src = i + drop;
                if (i >= keep) { p.long_valid[i].data_start = 0; p.long_valid[i].data_stop = 0; }
                else if (src < (int)s0.n_long_valid) p.long_valid[i] = s0.long_valid[src];
                else { p.long_valid[i].data_start = cs; p.long_valid[i].data_stop = ce; }
            }
            p.n_long_valid = (uint8_t)keep;
            const uint16_t lm = dbl ? (uint16_t)((1u << COORD_HISTORY_DEPTH) - 1u) : 0, gm = dbl ? (uint16_t)((1u << keep) - 1u) : 0;
            p.last_valid_doubled_mask_lo = (uint8_t)(lm & 0xFF); p.last_valid_doubled_mask_hi = (uint8_t)(lm >> 8);
            p.long_valid_doubled_mask = gm;
        }
    }
    return p;
}
__device__ inline void predict_body1(const PredictArgs1 &a, int k)
{
    const int base = a.first_of ? a.first_of[k - a.first] : a.first;
    if (base != k) a.states[k] = predict_state1(a, k, base);
}

/* Repair of a run of broken links (pcm1_frames_engine.inc): the frame behind the first link of the run has been given what its
 * predecessor really handed on (sdv_k_anchor).  A frame list[i] further into the run, whose run starts at frame head[i]:
 *   DRAFT mode (the whole tuning is handed on): predicted again from its run's head - or, when that tells nothing new, its own
 *   predecessor's outcome;
 *   the other modes, first attempt (sticky[i]): predicted again from the head with the coordinates the stream carries (predict_state1);
 *   later attempts: its own predecessor's outcome (what a frame hands on depends little on what it was handed), except for the
 *   multi-frame history, which only passes through the frames - that is rebuilt from the head's true state and what the frames
 *   since then have pushed themselves, so that one wrong median does not need sixteen rounds to leave the chain. */
struct RepairArgs1 { PredictArgs1 p; const sdv_v2d_state *states_out; const int *list, *head; const uint8_t *sticky; int n; const uint2 *frame_med; };
__device__ inline void repair_body1(const RepairArgs1 &a, int i)
{
    const int k = a.list[i], h = a.head[i];
    if (a.p.f.mode == SDV_MODE_DRAFT || a.sticky[i]) {
        sdv_v2d_state p = predict_state1(a.p, k, h, a.p.f.mode != SDV_MODE_DRAFT);
        const sdv_v2d_state cur = a.p.states[k];
        uint32_t x[sizeof(sdv_v2d_state) / 4], y[sizeof(sdv_v2d_state) / 4];
        __builtin_memcpy(x, &p, sizeof(p));
        __builtin_memcpy(y, &cur, sizeof(cur));
        bool same = true;
        for (unsigned q = 0; q < sizeof(sdv_v2d_state) / 4; q++) same = same && (x[q] == y[q]);
        a.p.states[k] = same ? a.states_out[k - 1] : p;
        return;
    }
    sdv_v2d_state p = a.states_out[k - 1];
    const sdv_v2d_state h_in = a.p.states[h];
    int n_long = h_in.reset_stats ? 0 : h_in.n_long_valid;
    sdv_coord lg[COORD_LONG_HISTORY];
    for (int q = 0; q < COORD_LONG_HISTORY; q++) lg[q] = h_in.long_valid[q];
    for (int j = h; j < k; j++) {
        const uint2 m = a.frame_med[j];
        if (!m.y) continue;
        if (n_long == COORD_LONG_HISTORY) { for (int q = 0; q + 1 < COORD_LONG_HISTORY; q++) lg[q] = lg[q + 1]; n_long--; }
        lg[n_long].data_start = key_start(m.x); lg[n_long].data_stop = key_stop(m.x); n_long++;
    }
    p.n_long_valid = (uint8_t)n_long;
    for (int q = 0; q < COORD_LONG_HISTORY; q++) { if (q < n_long) p.long_valid[q] = lg[q]; else { p.long_valid[q].data_start = 0; p.long_valid[q].data_stop = 0; } }
    p.long_valid_doubled_mask = a.p.f.doubled ? (uint16_t)((1u << n_long) - 1u) : 0;
    a.p.states[k] = p;
}
/* the links of the chain after a round: flag[k] for k in [0, n - 1) */
struct VerifyArgs1 { FrameArgs f; };
__device__ inline void verify_body1(const VerifyArgs1 &a, int k)
{
    a.f.flag[k] = link_holds1(a.f, k, a.f.states_out[k], a.f.states_in[k + 1]) ? VF_OK : VF_BREAK;
}

} // namespace sdvp1f

/* two builds of the two kernels: MODE_INSANE (with the reference level sweep) and every other mode (process_line_p1) */
#ifndef SDV_P1PRE_WAVES_PER_EU
#define SDV_P1PRE_WAVES_PER_EU SDV_P1B_WAVES_PER_EU
#endif
#define SDV_P1F_KERNELS(SUFFIX, INSANE) \
__global__ void __launch_bounds__(64, SDV_P1PRE_WAVES_PER_EU) sdv_k_pcm1_prescan##SUFFIX(sdvp1f::FrameArgs1 a) \
{ \
    __shared__ sdvp1b::P1Lds lds; \
    const int i = (int)blockIdx.x, f = a.f.frame_list ? a.f.frame_list[i / sdvp1f::COORD_CHECK_LINES] : a.f.frame_lo + i / sdvp1f::COORD_CHECK_LINES; \
    sdvp1f::prescan_body<INSANE>(a, lds, f, i % sdvp1f::COORD_CHECK_LINES); \
} \
__global__ void __launch_bounds__(64, SDV_P1B_WAVES_PER_EU) sdv_k_pcm1_frames_bin##SUFFIX(sdvp1f::FrameArgs1 a) \
{ \
    __shared__ sdvp1b::P1Lds lds; \
    const int f = a.f.frame_list ? a.f.frame_list[blockIdx.x] : a.f.frame_lo + (int)blockIdx.x; \
    sdvp1f::frame_body1<INSANE>(a, lds, f); \
}
SDV_P1F_KERNELS(, false)
SDV_P1F_KERNELS(_insane, true)
#ifndef SDV_P1F_LEAN_WAVES_PER_EU
#define SDV_P1F_LEAN_WAVES_PER_EU 4
#endif
/* the lean build of the frame kernel (every mode: what it holds - lines that read from what they inherit - is the same in all of them) */
__global__ void __launch_bounds__(64, SDV_P1F_LEAN_WAVES_PER_EU) sdv_k_pcm1_frames_lean(sdvp1f::FrameArgs1 a)
{
    __shared__ sdvp1b::P1Lds lds;
    const int f = a.f.frame_list ? a.f.frame_list[blockIdx.x] : a.f.frame_lo + (int)blockIdx.x;
    sdvp1f::frame_body1<false, true>(a, lds, f);
}
#ifndef SDV_EMU
__global__ void sdv_k_pcm1_predict(sdvp1f::PredictArgs1 a)
{
    const int k = a.first + (a.first_of ? 0 : 1) + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < a.hi) sdvp1f::predict_body1(a, k);
}
__global__ void sdv_k_pcm1_repair(sdvp1f::RepairArgs1 a)
{
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < a.n) sdvp1f::repair_body1(a, i);
}
__global__ void sdv_k_pcm1_verify(sdvp1f::VerifyArgs1 a)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k + 1 < a.f.n_total) sdvp1f::verify_body1(a, k);
}
#endif
