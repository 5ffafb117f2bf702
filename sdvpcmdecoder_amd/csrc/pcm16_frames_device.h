/*
 * pcm16_frames_device.h - PCM-16x0 frame driver: the PCM-16x0 branch of VideoToDigital::doBinarize (videotodigital.cpp:698-1815)
 * and VideoToDigital::prescanCoordinates (:148-345) around the line binarizer of pcm16_bin_device.h (SURVEY.md section 8 row a11
 * for PCM-16x0).  Same two kernels per round and the same scheduling as PCM-1 (pcm1_frames_device.h); what differs:
 *   - a video line is read three times (left, middle, right third, :902-925) and queued as three PCM16X0SubLines; the parts of a
 *     line hand levels and coordinates to each other (:1455-1511) and a part forced bad takes the parts behind it along (:1168-1180);
 *   - the prescan reads the right third of its four lines (:253-259) and leaves VideoLine::scan_done set on them, so the main pass
 *     does not search those lines again (binarizer.cpp:5846);
 *   - the window of the last valid coordinates holds 27 entries (9 lines x 3 parts, :1265-1273), so the chain state is longer.
 */
#pragma once
#ifndef SDV_P16_STAMPS
#define SDV_P16_STAMPS 0         /* developer aid: the frame's statistics row is overwritten with four cycle sums (rows staged, parts decoded, per-part bookkeeping, records stored) */
#endif
#if SDV_P16_STAMPS && !defined(SDV_EMU)
#define P16_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define P16_ADD(i, t0, t1) (p16_stamp[i] += (t1) - (t0))
#else
#define P16_T(x) ((void)0)
#define P16_ADD(i, t0, t1) ((void)0)
#endif
#ifndef SDV_P16_BATCH
#define SDV_P16_BATCH 1          /* runs of lines that read from one tuning take batch16 */
#endif
#ifndef SDV_P16_LEAN
#define SDV_P16_LEAN 1           /* parts that read from their presets take lean_part16 */
#endif
#ifndef SDV_P16_ABLATE
#define SDV_P16_ABLATE 0        /* developer aid: 1 no line decode, 2 no per-part bookkeeping, 3 no record store (outputs are wrong) */
#endif
#include "pcm16_bin_device.h"
#include "pcm1_frames_device.h"

namespace sdvp16f {
using namespace sdv;
using namespace sdvp16;
using sdvp1f::PrescanRes;
using sdvp1f::prescan_ref_of;
using sdvp1f::frame_buf_lines;
using sdvp1f::frame_buf_row;
using sdvp1f::prescan_runs;
using sdvp1f::ctx_for_line;
using sdvp1f::COORD_CHECK_LINES;
using sdvp1f::COORD_CHECK_PARTS;
using sdvp1f::sweep_flag_matters;

enum { P16_LINES_PF = 245 };                                    /* PCM16X0DataStitcher::LINES_PF, pcm16x0datastitcher.h:124 */
enum { LV16 = COORD_HISTORY_DEPTH * P16_SUBLINES };             /* 27 */

/* the chain state of a PCM-16x0 stream: sdv_v2d_state with a last-valid window of 27 (entries 9.. in `more`) */
struct State16 { sdv_v2d_state s; sdv_coord more[LV16 - COORD_HISTORY_DEPTH]; };

struct FrameArgs16 {
    FrameArgs f;                    /* geometry, flags, stats, scratch as for STC-007 (f.recs, f.states_in/out unused) */
    const State16 *states_in; State16 *states_out;
    sdv_pcm16x0_bin_rec *recs16;    /* frame k: recs16 + k*(3*height+3) (+1 behind the NEW_FILE frame) */
    PrescanRes *prescan;            /* [n_total][COORD_CHECK_LINES]; pad[0] = VideoLine::scan_done after the prescan read */
    uint2 *frame_med;               /* [n_total] what frame f pushed into the multi-frame history: {coordinate key, 1} or {0, 0} (nothing) */
};
struct Lds16 { P16Lds p; uint32_t lv_keys16[LV16]; };

/* ---- prescan: block = (frame, k) ------------------------------------------------------------------------------------------ */
template <bool kInsane>
__device__ inline void prescan_body(const FrameArgs16 &a16, Lds16 &lds, int f, int k)
{
    const FrameArgs &a = a16.f;
    K1_BEGIN(); K1_T(tq0_);
    PrescanRes r; r.start = r.stop = 0; r.ref = 0; r.valid = 0; r.pad[0] = r.pad[1] = 0;
    if (prescan_runs(a, f)) {
        const int gap = frame_buf_lines(a, f) / (COORD_CHECK_PARTS - 1);
        const int row = frame_buf_row(a, f, (k + 1) * gap);
        if (row >= 0) {
            sdvp1b::stage_row(lds.p.w.px, a.luma + (size_t)f * a.frame_stride + (size_t)row * a.row_stride, a.width);
            K1_T(tq1_); K1_ADD(16, tq0_, tq1_);
            /* both values of the Binarizer's sticky sweep flag where it matters (pcm1_frames_device.h, PrescanRes) */
            for (int variant = 0; variant < (sweep_flag_matters(a.preset) ? 2 : 1); variant++) {
                BinCtx c; Bin b;
                b.in_black = b.in_white = b.in_ref = 0; coords_clear(b.in_coord);
                ctx_for_line(a, c, b);
                b.do_ref_lvl_sweep = variant != 0;
                L16 out;
                bool scan_done = false;
                process_line_p16<kInsane>(c, b, true, PART_RIGHT, scan_done, lds.p, out, a.doubled != 0);
                PrescanRes q = r;
                if (crc_valid(out)) { q.start = out.coords.start; q.stop = out.coords.stop; q.ref = out.ref_level; q.valid = 1; }
                q.pad[0] = scan_done ? 1 : 0;
                q.pad[1] = out.bw_set ? 1 : 0;
                if (lane_id() == 0) a16.prescan[((size_t)variant * a.n_total + f) * COORD_CHECK_LINES + k] = q;
                SDV_WAVE_SYNC();
            }
            K1_T(tq2_); K1_ADD(23, tq0_, tq2_); K1_FLUSH();
            return;
        }
    }
    if (lane_id() == 0) {
        a16.prescan[(size_t)f * COORD_CHECK_LINES + k] = r;
        if (sweep_flag_matters(a.preset)) a16.prescan[((size_t)a.n_total + f) * COORD_CHECK_LINES + k] = r;
    }
}

/* ---- state of a PCM-16x0 frame wave --------------------------------------------------------------------------------------- */
struct V2D16 {
    V2D v;                          /* the part shared with STC-007; last_words unused */
    uint8_t prescan_ref;
    uint8_t pre_variant;            /* bit k: prescan line k of this frame was taken from the second variant */
    uint64_t lw0, lw1, lw2;         /* the data words of last_pcm16x0_p0/p1/p2_line (the 48 data cells; named members: an indexed array would put the whole state into scratch) */
};

/* one of three values by an index 0..2, by masks: written as `i == 0 ? a : (i == 1 ? b : c)` over members of one object the compiler makes it one
 * load through a computed address - and an object that is addressed like that lives in scratch memory, all of it (the frame's whole state did) */
__device__ __forceinline__ uint64_t pick3_u64(int i, uint64_t a, uint64_t b, uint64_t c)
{
    const uint64_t ma = i == 0 ? ~0ull : 0ull, mb = i == 1 ? ~0ull : 0ull, mc = (i != 0 && i != 1) ? ~0ull : 0ull;
    return (a & ma) | (b & mb) | (c & mc);
}
__device__ inline void load_state16(V2D16 &w, Lds16 &lds, const State16 *s, const FrameArgs &a)
{
    v2d_load_state(w.v, lds.p.w, &s->s, a);
    w.prescan_ref = (uint8_t)uni(prescan_ref_of(s->s));
    for (int i = 0; i < LV16; i++) {
        const sdv_coord cc = i < COORD_HISTORY_DEPTH ? s->s.last_valid[i] : s->more[i - COORD_HISTORY_DEPTH];
        lds.lv_keys16[i] = coords_key(cc.data_start, cc.data_stop);
    }
    w.lw0 = w.lw1 = w.lw2 = 0;
}

__device__ inline bool link_holds16(const FrameArgs &a, int f, const State16 &out, State16 next_in)
{
    if (prescan_runs(a, f + 1)) next_in.s.bin = out.s.bin;             /* reset before it is read (prescanCoordinates :221) */
    uint32_t x[sizeof(State16) / 4], y[sizeof(State16) / 4];
    __builtin_memcpy(x, &out, sizeof(out));
    __builtin_memcpy(y, &next_in, sizeof(next_in));
    bool same = true;
    for (unsigned i = 0; i < sizeof(State16) / 4; i++) same = same && (x[i] == y[i]);
    return same;
}

/* The chain after frame f, a dword per lane (State16 is 48 dwords: the pattern of v2d_store_state, stc007_device.h - put together by one lane on the
 * stack it was the frame kernel's scratch memory); the link to frame f+1 is checked by the frame itself, as link_holds16 does. */
__device__ inline uint32_t state16_half(const V2D16 &w, const Lds16 &lds, const FrameArgs &a, int h)
{
    const V2D &v = w.v;
    enum { H_MORE = sizeof(sdv_v2d_state) / 2 };        /* halfword 60: more[0] */
    if ((h >= V2D_H_LAST && h < V2D_H_LONG) || (h >= H_MORE && h < H_MORE + 2 * (LV16 - COORD_HISTORY_DEPTH))) {
        const int k = h < V2D_H_LONG ? h - V2D_H_LAST : h - H_MORE + 2 * COORD_HISTORY_DEPTH, e = k >> 1;      /* entry e of the window of 27 */
        if (e >= v.n_last) return 0;
        const uint32_t key = lds.lv_keys16[e];
        return (uint32_t)(uint16_t)((k & 1) ? key_stop(key) : key_start(key));
    }
    if (h >= V2D_H_LONG && h < V2D_H_LONG + 2 * COORD_LONG_HISTORY) {
        const int k = h - V2D_H_LONG, e = k >> 1;
        if (e >= v.n_long) return 0;
        const uint32_t key = lds.p.w.long_keys[e];
        return (uint32_t)(uint16_t)((k & 1) ? key_stop(key) : key_start(key));
    }
    const int n9 = v.n_last < COORD_HISTORY_DEPTH ? v.n_last : COORD_HISTORY_DEPTH;
    switch (h) {
    case 0: return (uint32_t)v.bin.in_black | ((uint32_t)v.bin.in_white << 8);
    case 1: return (uint32_t)v.bin.in_ref;
    case 2: return (uint32_t)(uint16_t)v.bin.in_coord.start;
    case 3: return (uint32_t)(uint16_t)v.bin.in_coord.stop;
    case 4: return v.bin.in_coord.doubled ? 1u : 0u;
    case 5: return (v.bin.do_ref_lvl_sweep ? 1u : 0u) | (v.reset_stats ? 0x100u : 0u);
    case 6: return (uint32_t)(uint8_t)v.n_last | ((uint32_t)(uint8_t)v.n_long << 8);
    case 7: return a.doubled ? (uint32_t)(uint16_t)((1u << n9) - 1u) : 0u;
    case 8: return a.doubled ? (uint32_t)(uint16_t)((1u << v.n_long) - 1u) : 0u;
    case 59: return (uint32_t)(uint8_t)(w.prescan_ref ^ 128) << 8;      /* _pad[0] = 0, _pad[1] = prescan_ref */
    default: return 0;
    }
}
__device__ inline void store_state16(const V2D16 &w, const Lds16 &lds, const FrameArgs16 &a16, int f)
{
    enum { NDW = sizeof(State16) / 4 };
    static_assert(NDW <= 64 && offsetof(State16, more) == sizeof(sdv_v2d_state) && sizeof(sdv_bin_state) == 10, "a dword of the state per lane");
    const FrameArgs &a = a16.f;
    const int lane = lane_id();
    const bool in = lane < NDW;
    const int dw = in ? lane : 0;
    SDV_WAVE_SYNC();            /* (the histories were written by lane 0) */
    const uint32_t mine = state16_half(w, lds, a, 2 * dw) | (state16_half(w, lds, a, 2 * dw + 1) << 16);
    if (in) reinterpret_cast<uint32_t *>(&a16.states_out[f])[dw] = mine;
    uint8_t fl = VF_OK;
    if (f + 1 < a.n_total) {
        const uint32_t next = reinterpret_cast<const uint32_t *>(&a16.states_in[f + 1])[dw];
        /* when the worker prescans frame f+1 it resets its Binarizer first (prescanCoordinates :221): what this frame left preset there - the first ten
         * bytes of the state - does not reach frame f+1 and does not count */
        const uint32_t cmp = prescan_runs(a, f + 1) ? (dw < 2 ? 0u : (dw == 2 ? 0xFFFF0000u : 0xFFFFFFFFu)) : 0xFFFFFFFFu;
        if (__ballot(in && ((next ^ mine) & cmp) != 0) != 0ull) fl = VF_BREAK;
    }
    if (lane == 0) a.flag[f] = fl;
}

__device__ inline void begin_frame16(V2D16 &w, const FrameArgs16 &a16, Lds16 &lds, int f)   /* :772-822 */
{
    const FrameArgs &a = a16.f;
    V2D &v = w.v;
    w.pre_variant = 0;
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
            bin_set_good_parameters_reset(v.bin, a.preset);
            uint32_t keys[COORD_CHECK_LINES]; uint8_t refs[COORD_CHECK_LINES]; int n = 0;
            for (int k = 0; k < COORD_CHECK_LINES; k++) {
                const int variant = (sweep_flag_matters(a.preset) && v.bin.do_ref_lvl_sweep) ? 1 : 0;
                w.pre_variant = (uint8_t)(w.pre_variant | (variant << k));
                const PrescanRes r = a16.prescan[((size_t)variant * a.n_total + f) * COORD_CHECK_LINES + k];
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
        if (!coords_valid(v.frame_avg)) { uint32_t k; if (median_keys(lds.p.w.long_keys, v.n_long, &k)) v.frame_avg = key_to_coords(k, a.doubled != 0); }
        else v.bin.in_ref = w.prescan_ref;
        if (coords_valid(v.frame_avg)) bin_set_data_coordinates2(v.bin, v.frame_avg.start, v.frame_avg.stop);
    }
}

__device__ inline void set_good_parameters_p16(Bin &b, const sdv_bin_preset &ps, const L16 &l)   /* binarizer.cpp:353-377 */
{
    if (crc_valid_ignore_forced(l)) { b.in_ref = l.ref_level; bin_set_data_coordinates(b, l.coords); bin_set_bw_levels(b, ps, l.black, l.white); }
}

__device__ inline void service_line16(V2D16 &w, const FrameArgs &a, L16 &wl, uint8_t srv)       /* :1006-1114 */
{
    V2D &v = w.v;
    p16_clear(wl);
    set_service(wl, srv);
    if (srv == SDV_SRV_NEW_FILE || srv == SDV_SRV_END_FILE) {
        v.line_in_field_cnt = 0;
        v.n_last = v.nfv = v.nfi = v.n_long = 0;
        if (srv == SDV_SRV_END_FILE || !coords_valid(v.frame_avg)) bin_set_good_parameters_reset(v.bin, a.preset);
    } else if (srv == SDV_SRV_END_FIELD) {
        v.field_state = FIELD_NEW;
        v.line_in_field_cnt = 0;
        v.good_coords_in_field = 0; v.pcm_lines_in_field = 0;
        w.lw0 = w.lw1 = w.lw2 = 0;
    }
}

/* one part of a regular line, after Binarizer::processLine: VideoToDigital :1115-1634 (PCM-16x0 branches) */
__device__ inline void post_part16(V2D16 &w, const FrameArgs &a, Lds16 &lds, L16 &wl, uint32_t *fv_keys, uint32_t *fi_keys, bool even_line,
                                   bool &force_bad_line, bool scan_done)
{
    V2D &v = w.v;
    const sdv_bin_preset &ps = a.preset;
    const bool count_has_data = wl.bw_set;
    const bool count_has_pcm = crc_valid(wl) || count_has_data;
    const int part = wl.line_part;
    wl.queue_order = v.line_in_field_cnt;                               /* :1141 */
    if (count_has_pcm && v.field_state == FIELD_NEW) v.field_state = FIELD_UNSAFE;
    if (crc_valid(wl) && force_bad_line) wl.forced_bad = true;          /* :1168-1180 */
    if (crc_valid(wl)) {
        v.good_coords_in_field++;
        v.q_line_length = (uint16_t)a.width;
        if (a.check_line_copy) {
            if (v.field_state == FIELD_UNSAFE) {
                set_good_parameters_p16(v.bin, ps, wl);
                if (ps.en_first_line_dup) { wl.forced_bad = true; force_bad_line = true; }
            } else {
                const uint64_t lw = pick3_u64(part, w.lw0, w.lw1, w.lw2);
                const int diff = __popcll(((wl.v >> 16) ^ lw) & 0x00FF00FF00FFull);     /* per word the XOR is truncated to uint8_t */
                const int16_t s0 = (int16_t)get_word(wl, 0), s2 = (int16_t)get_word(wl, 2);
                const bool almost_silent = (!(s0 >= 4) && !(s0 < -4)) || (!(s2 >= 4) && !(s2 < -4));                      /* pcm16x0subline.cpp:291-318 */
                if (!almost_silent && diff <= (P16_DATA / BIT_DIFF_THRES_DIV)) { wl.forced_bad = true; if (!even_line) v.q_dup_odd++; else v.q_dup_even++; }
            }
        }
        if (crc_valid_ignore_forced(wl)) {
            const uint32_t key = coords_key(wl.coords.start, wl.coords.stop);
            SDV_WAVE_SYNC();
            {   /* the window moves up by one when it is full: every lane carries one entry (no serial chain through LDS) */
                const int ln = lane_id();
                const bool full = v.n_last == LV16;
                const uint32_t moved = (full && ln < LV16 - 1) ? lds.lv_keys16[ln + 1] : 0u;
                SDV_WAVE_SYNC();
                if (full && ln < LV16 - 1) lds.lv_keys16[ln] = moved;
                if (ln == 0) lds.lv_keys16[full ? LV16 - 1 : v.n_last] = key;
            }
            if (v.n_last < LV16) v.n_last++;
            SDV_WAVE_SYNC();
            fv_keys[v.nfv++] = key;
            if (a.coordinate_damper && !ps.en_force_coords && (v.n_last > (COORD_HISTORY_DEPTH / 2))) {
                Coords target; coords_clear(target);
                uint32_t k;
                if (median_keys(lds.lv_keys16, v.n_last, &k)) target = key_to_coords(k, false);
                if (!coords_valid(target)) target = v.frame_avg;
                if (coords_valid(target)) {
                    const int16_t ds = (int16_t)(wl.coords.start - target.start), de = (int16_t)(wl.coords.stop - target.stop);
                    const uint8_t in_delta = (uint8_t)(get_ppb(wl) * 3);
                    if (((int)ds <= -(int)in_delta) || ((int)ds >= (int)in_delta) || ((int)de <= -(int)in_delta) || ((int)de >= (int)in_delta)) { wl.forced_bad = true; force_bad_line = true; }
                }
            }
        }
        if (crc_valid(wl)) set_good_parameters_p16(v.bin, ps, wl);
        else { if (!even_line) v.q_bad_odd++; else v.q_bad_even++; }
        if (part == 2) v.field_state = FIELD_INIT;
    } else {
        if (v.q_line_length == 0) v.q_line_length = (uint16_t)a.width;
        if (coords_valid(wl.coords)) fi_keys[v.nfi++] = coords_key(wl.coords.start, wl.coords.stop);
        if (count_has_data) {
            Coords preset_coords; coords_clear(preset_coords);
            if (!even_line) v.q_bad_odd++; else v.q_bad_even++;
            if (!ps.en_force_coords) {
                uint32_t k;
                if (median_keys(lds.lv_keys16, v.n_last, &k)) preset_coords = key_to_coords(k, a.doubled != 0);
                if (!coords_valid(preset_coords)) preset_coords = v.frame_avg;
            }
            if (part == 2) {
                v.field_state = FIELD_INIT;
                bin_set_data_coordinates(v.bin, preset_coords);
                bin_set_bw_levels(v.bin, ps, 0, 0);
            } else {                                                    /* :1455-1511: the next part of the same line */
                v.bin.in_ref = wl.ref_level;
                bin_set_bw_levels(v.bin, ps, wl.black, wl.white);
                if (scan_done) bin_set_data_coordinates(v.bin, wl.coords);
                else bin_set_data_coordinates(v.bin, preset_coords);
            }
        } else {
            bin_set_bw_levels(v.bin, ps, 0, 0);
        }
    }
    if (!even_line) v.q_odd++; else v.q_even++;
    if (count_has_pcm) {
        if (!even_line) v.q_pcm_odd++; else v.q_pcm_even++;
        v.pcm_lines_in_field++;
        const uint64_t nw = wl.v >> 16;
        w.lw0 = part == 0 ? nw : w.lw0; w.lw1 = part == 1 ? nw : w.lw1; w.lw2 = part == 2 ? nw : w.lw2;
    }
    v.line_in_field_cnt++;
}

#ifdef SDV_EMU
__device__ inline uint32_t lane_read32(uint32_t x, uint32_t idx) { return (uint32_t)__shfl((int)x, (int)idx); }
#else
__device__ inline uint32_t lane_read32(uint32_t x, uint32_t idx) { return (uint32_t)__builtin_amdgcn_readlane((int)x, (int)uniu(idx)); }
#endif
/* ---- the tape plays: a part that reads from what its predecessor left preset ------------------------------------------------------- */
/* Binarizer::processLine for a part whose levels, reference level and coordinates are preset and that reads valid on the first rung of
 * the ladder (hysteresis 0, shift 0) is stage STG_INPUT_ALL and nothing else (binarizer.cpp:774-931): one LDS byte per lane, two ballots,
 * the automaton, the CRC, the Control Bit, the Bit Picker's count of cut-off cells.  What depends on the tuning only - the pixel every
 * lane samples in each of the three parts, the levels with their clipping test, the cut-off counts - is kept while the tuning stays the
 * same (on a tape that plays: for the whole frame).  Anything else - no complete presets, forced coordinates, a part that does not read
 * at once - goes through process_line_p16 from the start. */
struct Lean16 {
    uint32_t key_coords, key_levels;        /* what `x` etc. were computed for; key_levels = 0xFFFFFFFF: nothing yet */
    uint16_t x0, x1, x2, x_ctrl;            /* lane's pixel in the left / middle / right part (named members: an array indexed by the part puts the frame's state into scratch memory); the Control Bit's pixel */
    uint8_t low, high, bits_l, bits_r;
    bool usable;
    uint32_t psm, hpsm; int16_t pso;
};
__device__ inline void lean16_reset(Lean16 &n) { n.key_levels = 0xFFFFFFFFu; n.key_coords = 0; n.usable = false; }
/* true when parts can be read the lean way under the tuning in b (and n describes that tuning) */
__device__ inline bool lean16_prepare(Lean16 &n, const BinCtx &c, const Bin &b)
{
    if (c.ps.en_force_coords) return false;
    if (!(are_bw_levels_preset(b, c.ps) && is_ref_level_preset(b, c.ps) && coords_valid(b.in_coord))) return false;
    if (!(b.in_ref < b.in_white && b.in_ref > b.in_black)) return false;
    if (!(c.scan_end > c.scan_start && P16_BITS <= (c.scan_end - c.scan_start))) return false;
    const uint32_t kc = coords_key(b.in_coord.start, b.in_coord.stop) , kl = (uint32_t)b.in_black | ((uint32_t)b.in_white << 8) | ((uint32_t)b.in_ref << 16) | ((uint32_t)(b.in_coord.doubled ? 1 : 0) << 24);
    if (kc != n.key_coords || kl != n.key_levels) {
        n.key_coords = kc; n.key_levels = kl;
        L16 t; p16_clear(t);
        t.pixel_start = c.scan_start; t.pixel_stop = c.scan_end;
        t.coords = b.in_coord;
        set_ppb(t, t.coords);
        n.psm = t.psm; n.hpsm = t.hpsm; n.pso = t.pso;
        const int lane = lane_id();
        n.x0 = (uint16_t)pixel_of(t, part_start_bit((uint8_t)PART_LEFT) + lane, 0); n.x1 = (uint16_t)pixel_of(t, part_start_bit((uint8_t)PART_MIDDLE) + lane, 0); n.x2 = (uint16_t)pixel_of(t, part_start_bit((uint8_t)PART_RIGHT) + lane, 0);
        n.x_ctrl = (uint16_t)pixel_of(t, 2 * P16_DATA, 0);
        n.low = get_low_level(b.in_ref, 0); n.high = get_high_level(b.in_ref, 0);
        n.usable = !(n.low <= b.in_black) && !(n.high >= b.in_white);
        /* pickCutBitsUpPCM16X0 (binarizer.cpp:6599-7013): how many cells the picture cuts off at either end */
        t.black = b.in_black; t.white = b.in_white; t.ref_level = b.in_ref;
        for (int side = 0; side < 2; side++) {
            const bool left = side == 0;
            int max_cut = left ? c.ps.left_bit_pick : c.ps.right_bit_pick; if (c.mode == SDV_MODE_DRAFT) max_cut /= 2;
            int first = left ? c.scan_start : c.scan_end, bits = 0;
            const int half_ppb = ((int)get_ppb(t) + 1) / 2;
            for (int i = 0; i < max_cut; i++) {
                const int cur = pixel_of(t, left ? i : P16_BITS - 1 - i, 0);
                if ((left ? (cur - first) : (first - cur)) >= half_ppb) break;
                if (i == 0) first = cur;
                bits = i + 1;
            }
            if (left) n.bits_l = (uint8_t)bits; else n.bits_r = (uint8_t)bits;
        }
    }
    return n.usable && c.force_bit_picker;
}
__device__ inline bool lean_part16(Lean16 &n, const BinCtx &c, const Bin &b, uint8_t part, const uint8_t *px_row, L16 &out)
{
    if (!lean16_prepare(n, c, b)) return false;
    const int lane = lane_id();
    const int q = part == PART_LEFT ? 0 : (part == PART_MIDDLE ? 1 : 2);
    const uint32_t xq = (uint32_t)pick3_u64(q, n.x0, n.x1, n.x2);
    const uint8_t p0 = px_row[xq];
    const uint64_t a_lo = __ballot(p0 > n.low), b_lo = __ballot(p0 >= n.high);
    uint64_t s_lo, s_hi;
    solve_automaton(a_lo, 0ull, b_lo, 0ull, s_lo, s_hi);
    const uint64_t v = __brevll(s_lo);
    const int par = __popcll(v & c_crc16.k[lane & 15]) & 1;
    const uint16_t crc = (uint16_t)((uint16_t)(__ballot(par) & 0xFFFF) ^ c_crc16.base);
    if (crc != (uint16_t)(v & 0xFFFF)) return false;
    /* the line as processLine leaves it (STG_INPUT_ALL -> STG_DATA_OK) */
    p16_clear(out);
    out.line_part = (uint8_t)q;
    out.pixel_start = c.scan_start; out.pixel_stop = c.scan_end;
    out.coords = b.in_coord;
    out.black = b.in_black; out.white = b.in_white; out.bw_set = true;
    out.ref_level = b.in_ref; out.ref_low = n.low; out.ref_high = n.high;
    out.hyst = 0; out.shift = 0;
    out.psm = n.psm; out.hpsm = n.hpsm; out.pso = n.pso;
    out.v = v; out.calc_crc = crc;
    out.control_bit = !(px_row[n.x_ctrl] < b.in_ref);
    out.picked_l = q == 0 ? n.bits_l : 0; out.picked_r = q == 2 ? n.bits_r : 0;
    out.by_ext_tune = true; out.coords_set = true;
    return true;
}

/* ---- a run of lines that all read from the same presets: 21 video lines = 63 parts at a time -------------------------------------- */
/* While a tape plays every part is decoded with the tuning its predecessor was decoded with and hands that same tuning on, so the parts
 * of a run of lines do not depend on each other: phase A reads them - per line four byte gathers straight from the frame (the three
 * cells a lane owns and the Control Bit's pixel; no LDS staging, no barrier), two ballots, the automaton and the CRC per part - and parks
 * each part's 64 cells in the lane that owns it; phase B does VideoToDigital's per-part bookkeeping (:1115-1634) for all parts at once, a
 * lane per part: duplicate-line test against the same part of the line above, the record, the counters as sums.  Preconditions:
 * the field is past its first valid part, the window of last valid coordinates holds nothing but the preset coordinates (so the
 * coordinate damper sees a delta of zero).  The first line with a part that does not read, or that repeats the line above, ends the run:
 * it and what follows take the part-by-part path.  Returns the number of video lines taken. */
enum { BATCH16_LINES = 21 };
#ifndef SDV_P16_CAPTURE_D
#define SDV_P16_CAPTURE_D 7         /* batch16: lines in flight (a divisor of BATCH16_LINES) */
#endif
__device__ inline int batch16(V2D16 &w, const FrameArgs16 &a16, Lds16 &lds, const Lean16 &n, const uint8_t *frame, int field, int idx, int nl,
                              uint32_t frame_no, sdv_pcm16x0_bin_rec *rec, uint32_t *fv_keys, L16 &wl)
{
    const FrameArgs &a = a16.f;
    V2D &v = w.v;
    const int lane = lane_id();
    if (v.field_state != FIELD_INIT) return 0;
    if ((uint8_t)((uint8_t)(n.psm / 128u) * 3) == 0) return 0;         /* in_delta = getPPB() * 3 as uint8_t: at 0 the damper flags a delta of zero */
    const uint32_t key = coords_key(v.bin.in_coord.start, v.bin.in_coord.stop);
    if (__ballot(lane < v.n_last && lds.lv_keys16[lane < LV16 ? lane : 0] != key) != 0ull) return 0;
    int n_lines = nl - idx; if (n_lines > BATCH16_LINES) n_lines = BATCH16_LINES;
    const uint8_t low = n.low, high = n.high, ref = v.bin.in_ref;
    /* Phase A, the pattern of the STC-007 capture (stc007_device.h): per line four byte gathers per lane straight from the frame, D lines in flight
     * (a line costs its share of the memory's bandwidth, not a trip to it), two ballots per part, the masks parked in the lane that owns the part;
     * then every lane solves its own part - automaton and CRC. */
    uint32_t v_lo = 0, v_hi = 0, cb = 0;
    int n_ok = 0;
    {
        constexpr int D = SDV_P16_CAPTURE_D;
        static_assert(BATCH16_LINES % D == 0, "whole groups of D lines make a run of up to BATCH16_LINES");
        const uint32_t rs2 = 2u * (uint32_t)a.row_stride;
        const uint8_t *row0 = frame + (size_t)(2 * idx + field) * a.row_stride;
        const uint32_t x0 = n.x0, x1 = n.x1, x2 = n.x2, xc = n.x_ctrl;
        uint8_t q[D][4];
#pragma unroll
        for (int d = 0; d < D; d++) {
            const uint32_t o = (uint32_t)(d < n_lines ? d : n_lines - 1) * rs2;
            q[d][0] = sdvp1f::lean_load_u8(row0 + (o + x0)); q[d][1] = sdvp1f::lean_load_u8(row0 + (o + x1)); q[d][2] = sdvp1f::lean_load_u8(row0 + (o + x2)); q[d][3] = sdvp1f::lean_load_u8(row0 + (o + xc));
        }
        uint32_t ra0 = 0, ra1 = 0, rb0 = 0, rb1 = 0;
        for (int j0 = 0; j0 < n_lines; j0 += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int j = j0 + d;
                const uint8_t p0 = q[d][0], p1 = q[d][1], p2 = q[d][2], pc = q[d][3];
                const uint64_t a0 = __ballot(p0 > low), b0 = __ballot(p0 >= high), a1 = __ballot(p1 > low), b1 = __ballot(p1 >= high), a2 = __ballot(p2 > low), b2 = __ballot(p2 >= high);
                const uint32_t cbit = uniu(pc < ref ? 0u : 1u);
                {
                    const int jn = j + D < n_lines ? j + D : n_lines - 1; const uint32_t o = (uint32_t)jn * rs2;
                    q[d][0] = sdvp1f::lean_load_u8(row0 + (o + x0)); q[d][1] = sdvp1f::lean_load_u8(row0 + (o + x1)); q[d][2] = sdvp1f::lean_load_u8(row0 + (o + x2)); q[d][3] = sdvp1f::lean_load_u8(row0 + (o + xc));
                }
                ra0 = write_lane(ra0, (uint32_t)a0, 3 * j); ra1 = write_lane(ra1, (uint32_t)(a0 >> 32), 3 * j); rb0 = write_lane(rb0, (uint32_t)b0, 3 * j); rb1 = write_lane(rb1, (uint32_t)(b0 >> 32), 3 * j);
                ra0 = write_lane(ra0, (uint32_t)a1, 3 * j + 1); ra1 = write_lane(ra1, (uint32_t)(a1 >> 32), 3 * j + 1); rb0 = write_lane(rb0, (uint32_t)b1, 3 * j + 1); rb1 = write_lane(rb1, (uint32_t)(b1 >> 32), 3 * j + 1);
                ra0 = write_lane(ra0, (uint32_t)a2, 3 * j + 2); ra1 = write_lane(ra1, (uint32_t)(a2 >> 32), 3 * j + 2); rb0 = write_lane(rb0, (uint32_t)b2, 3 * j + 2); rb1 = write_lane(rb1, (uint32_t)(b2 >> 32), 3 * j + 2);
                cb = write_lane(cb, cbit, 3 * j); cb = write_lane(cb, cbit, 3 * j + 1); cb = write_lane(cb, cbit, 3 * j + 2);
            }
        }
        uint64_t s_lo, s_hi;
        solve_automaton_lane((uint64_t)ra0 | ((uint64_t)ra1 << 32), 0ull, (uint64_t)rb0 | ((uint64_t)rb1 << 32), 0ull, s_lo, s_hi);
        const uint64_t cells = __brevll(s_lo);
        uint32_t crc = 0;
#pragma unroll
        for (int k = 0; k < 16; k++) crc |= (uint32_t)(__popcll(cells & c_crc16.k[k]) & 1) << k;
        crc ^= c_crc16.base;
        const bool ok = (uint16_t)crc == (uint16_t)(cells & 0xFFFF);
        const uint64_t okm = __ballot(ok || lane >= 3 * n_lines);
        n_ok = okm == ~0ull ? n_lines : (__ffsll((unsigned long long)~okm) - 1) / 3;
        v_lo = (uint32_t)cells; v_hi = (uint32_t)(cells >> 32);
    }
    if (n_ok == 0) return 0;
    /* phase B: lane j = part j % 3 of line j / 3 */
    const int part = lane % 3, line_in = lane / 3;
    const uint64_t mine = ((uint64_t)v_hi << 32) | v_lo;
    const uint64_t words = mine >> 16;
    if (a.check_line_copy) {
        const int src = lane >= 3 ? lane - 3 : lane;
        uint64_t above = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(words >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)words, src);
        if (lane < 3) above = pick3_u64(lane, w.lw0, w.lw1, w.lw2);
        const int diff = __popcll((words ^ above) & 0x00FF00FF00FFull);
        const int16_t s0 = (int16_t)(uint16_t)(mine >> 48), s2 = (int16_t)(uint16_t)(mine >> 16);
        const bool almost_silent = (!(s0 >= 4) && !(s0 < -4)) || (!(s2 >= 4) && !(s2 < -4));
        const uint64_t dup = __ballot(lane < 3 * n_ok && !almost_silent && diff <= (P16_DATA / BIT_DIFF_THRES_DIV));
        if (dup) { n_ok = (__ffsll((unsigned long long)dup) - 1) / 3; if (n_ok == 0) return 0; }
    }
    const int n_sub = 3 * n_ok;
    const bool even_line = field == 1;
    const uint16_t q0 = v.line_in_field_cnt;
    if (lane < n_sub) {
        sdv_pcm16x0_bin_rec r;
        r.frame_number = frame_no; r.line_number = (uint16_t)(field + 1 + 2 * (idx + line_in));
        r.words[0] = (uint16_t)(mine >> 48); r.words[1] = (uint16_t)(mine >> 32); r.words[2] = (uint16_t)(mine >> 16); r.words[3] = (uint16_t)mine;
        r.calc_crc = (uint16_t)mine;
        r.data_start = v.bin.in_coord.start; r.data_stop = v.bin.in_coord.stop;
        r.queue_order = (uint16_t)(q0 + lane);
        r.black_level = v.bin.in_black; r.white_level = v.bin.in_white; r.ref_low = low; r.ref_level = ref; r.ref_high = high;
        r.hysteresis_depth = 0; r.shift_stage = 0; r.service_type = SDV_SRV_NO;
        r.picked_bits_left = part == 0 ? n.bits_l : 0; r.picked_bits_right = part == 2 ? n.bits_r : 0;
        r.flags = (uint8_t)(SDV_LF_BY_EXT_TUNE | SDV_LF_BW_SET | SDV_LF_COORDS_SET | SDV_LF_CRC_VALID | (a.doubled ? SDV_LF_FROM_DOUBLED : 0));
        r.line_part = (uint8_t)part; r.control_bit = (uint8_t)cb; r._pad = 0;
        rec[lane] = r;
        fv_keys[v.nfv + lane] = key;
    }
    /* the counters of n_sub parts that read (post_part16 with a valid CRC, field state FIELD_INIT) */
    v.good_coords_in_field = (uint16_t)(v.good_coords_in_field + n_sub);
    v.q_line_length = (uint16_t)a.width;
    if (!even_line) { v.q_odd = (uint16_t)(v.q_odd + n_sub); v.q_pcm_odd = (uint16_t)(v.q_pcm_odd + n_sub); }
    else { v.q_even = (uint16_t)(v.q_even + n_sub); v.q_pcm_even = (uint16_t)(v.q_pcm_even + n_sub); }
    v.pcm_lines_in_field = (uint16_t)(v.pcm_lines_in_field + n_sub);
    v.line_in_field_cnt = (uint16_t)(v.line_in_field_cnt + n_sub);
    v.nfv += n_sub;
    {   /* the window of last valid coordinates: n_sub more entries of the same key */
        int fill_to = v.n_last + n_sub; if (fill_to > LV16) fill_to = LV16;
        SDV_WAVE_SYNC();
        if (lane >= v.n_last && lane < fill_to) lds.lv_keys16[lane] = key;
        SDV_WAVE_SYNC();
        v.n_last = fill_to;
    }
    {   /* what the last line leaves behind: its words for the duplicate test, and the line object of its last part */
        const uint32_t h0 = lane_read32((uint32_t)(words >> 32), (uint32_t)(n_sub - 3)), l0 = lane_read32((uint32_t)words, (uint32_t)(n_sub - 3));
        const uint32_t h1 = lane_read32((uint32_t)(words >> 32), (uint32_t)(n_sub - 2)), l1 = lane_read32((uint32_t)words, (uint32_t)(n_sub - 2));
        const uint32_t h2 = lane_read32((uint32_t)(words >> 32), (uint32_t)(n_sub - 1)), l2 = lane_read32((uint32_t)words, (uint32_t)(n_sub - 1));
        w.lw0 = ((uint64_t)h0 << 32) | l0; w.lw1 = ((uint64_t)h1 << 32) | l1; w.lw2 = ((uint64_t)h2 << 32) | l2;
        const uint64_t last = ((uint64_t)lane_read32(v_hi, (uint32_t)(n_sub - 1)) << 32) | lane_read32(v_lo, (uint32_t)(n_sub - 1));
        p16_clear(wl);
        wl.line_part = 2;
        wl.pixel_start = 0; wl.pixel_stop = (uint16_t)(a.width - 1);
        wl.coords = v.bin.in_coord;
        wl.black = v.bin.in_black; wl.white = v.bin.in_white; wl.bw_set = true;
        wl.ref_level = ref; wl.ref_low = low; wl.ref_high = high;
        wl.psm = n.psm; wl.hpsm = n.hpsm; wl.pso = n.pso;
        wl.v = last; wl.calc_crc = (uint16_t)last;
        wl.control_bit = lane_read32(cb, (uint32_t)(n_sub - 1)) != 0;
        wl.picked_r = n.bits_r;
        wl.by_ext_tune = true; wl.coords_set = true;
        wl.queue_order = (uint16_t)(q0 + n_sub - 1);
    }
    return n_ok;
}

/* kLean: the build without Binarizer::processLine's search stages - a frame with a part the tuning it inherits does not read is given up (its flag says
 * VF_ABORTED) and decoded again by the full build (pcm16_frames_engine.inc); the pattern of the STC-007 frame kernels. */
template <bool kInsane, bool kLean = false>
__device__ inline void frame_body16(const FrameArgs16 &a16, Lds16 &lds, int f)
{
    const FrameArgs &a = a16.f;
    if (kLean && (f == a.end_file_frame || frame_is_empty(a, f))) {          /* frames without pixels: the full build's */
        if (lane_id() == 0) a.flag[f] = VF_ABORTED;
        return;
    }
    V2D16 w; L16 wl;
    load_state16(w, lds, &a16.states_in[f], a);
    V2D &v = w.v;
    const uint32_t frame_no = a.first_frame_no + (uint32_t)f;
    const uint8_t *frame = a.luma + (size_t)f * a.frame_stride;
    uint32_t *fv_keys = a.scratch + (size_t)f * 6u * (size_t)a.height;         /* three sub-lines per line, valid + invalid lists */
    uint32_t *fi_keys = fv_keys + 3 * a.height;
    size_t rec_base = (size_t)f * (size_t)(3 * a.height + 3);
    if (a.new_file_frame >= 0 && f > a.new_file_frame) rec_base += 1;
    sdv_pcm16x0_bin_rec *rec = a16.recs16 + rec_base;
    const bool doubled = a.doubled != 0;

    begin_frame16(w, a16, lds, f);
    /* the rows the prescan has read: their VideoLine carries scan_done into the main pass */
    int pre_row[COORD_CHECK_LINES]; bool pre_done[COORD_CHECK_LINES];
    for (int k = 0; k < COORD_CHECK_LINES; k++) { pre_row[k] = -1; pre_done[k] = false; }
    if (prescan_runs(a, f)) {
        const int gap = frame_buf_lines(a, f) / (COORD_CHECK_PARTS - 1);
        for (int k = 0; k < COORD_CHECK_LINES; k++) { pre_row[k] = frame_buf_row(a, f, (k + 1) * gap); pre_done[k] = uni(a16.prescan[((size_t)((w.pre_variant >> k) & 1) * a.n_total + f) * COORD_CHECK_LINES + k].pad[0]) != 0; }
    }
    const int n_field[2] = { (a.height + 1) / 2, a.height / 2 };
    uint16_t line_num = 0;
    Lean16 lean; lean16_reset(lean);
    const bool empty_frame = frame_is_empty(a, f);
#if SDV_P16_STAMPS && !defined(SDV_EMU)
    unsigned long long p16_stamp[4] = { 0, 0, 0, 0 };
#endif
    if (f == a.new_file_frame) { service_line16(w, a, wl, SDV_SRV_NEW_FILE); emit_rec(wl, frame_no, 0, false, rec++); }
    for (int field = 0; field < 2; field++) {
        const int nl = n_field[field];
        for (int idx = 0; idx < nl; idx++) {
            line_num = (uint16_t)(field + 1 + 2 * idx);
            if (f == a.end_file_frame) {
                service_line16(w, a, wl, SDV_SRV_FILLER);
                emit_rec(wl, frame_no, line_num, false, rec++);
                continue;
            }
#if SDV_P16_BATCH
            {
                BinCtx cb0;
                ctx_for_line(a, cb0, v.bin);
                P16_T(t_g);
                if (lean16_prepare(lean, cb0, v.bin)) {
                    const int took = batch16(w, a16, lds, lean, frame, field, idx, nl, frame_no, rec, fv_keys, wl);
                    P16_T(t_h); P16_ADD(1, t_g, t_h);
                    if (took > 0) { rec += 3 * took; idx += took - 1; continue; }
                }
            }
#endif
            const int row = 2 * idx + field;
            if (empty_frame) {                      /* a dropped frame: three passes over an empty VideoLine, each a cleared sub-line (binarizer.cpp:1689-1700) */
                bool scan_done_e = false, force_bad_e = false;
                for (int sub = 0; sub < P16_SUBLINES; sub++) {
                    p16_clear(wl); wl.line_part = (uint8_t)sub;
                    const uint16_t ql = v.q_line_length;
                    post_part16(w, a, lds, wl, fv_keys, fi_keys, (line_num % 2) == 0, force_bad_e, scan_done_e);
                    v.q_line_length = ql;
                    emit_rec(wl, frame_no, line_num, false, rec++);
                }
                continue;
            }
            P16_T(t_a);
            sdvp1b::stage_row(lds.p.w.px, frame + (size_t)row * a.row_stride, a.width);
            P16_T(t_b); P16_ADD(0, t_a, t_b);
            bool scan_done = false;
            for (int k = 0; k < COORD_CHECK_LINES; k++) if (pre_row[k] == row && pre_done[k]) scan_done = true;
            bool force_bad_line = false;
            for (int sub = 0; sub < P16_SUBLINES; sub++) {
                BinCtx c;
                ctx_for_line(a, c, v.bin);
                bool coord_search = true;                               /* :927-950 */
                if (a.mode == SDV_MODE_DRAFT || a.mode == SDV_MODE_FAST) coord_search = !(v.good_coords_in_field > 9 || v.pcm_lines_in_field > 15);
                P16_T(t_c);
#if SDV_P16_ABLATE == 1
                p16_clear(wl);
#else
                if (!(SDV_P16_LEAN && lean_part16(lean, c, v.bin, (uint8_t)(PART_LEFT + sub), lds.p.w.px, wl))) {
                    if (kLean) { if (!input_all_p16(c, v.bin, (uint8_t)(PART_LEFT + sub), lds.p.w, wl, doubled)) { if (lane_id() == 0) a.flag[f] = VF_ABORTED; return; } }
                    else process_line_p16<kInsane>(c, v.bin, coord_search, (uint8_t)(PART_LEFT + sub), scan_done, lds.p, wl, doubled);
                }
#endif
                P16_T(t_d); P16_ADD(1, t_c, t_d);
#if SDV_P16_ABLATE != 2
                post_part16(w, a, lds, wl, fv_keys, fi_keys, (line_num % 2) == 0, force_bad_line, scan_done);
#endif
                P16_T(t_e); P16_ADD(2, t_d, t_e);
#if SDV_P16_ABLATE != 3
                emit_rec(wl, frame_no, line_num, doubled, rec++);
#else
                rec++;
#endif
                P16_T(t_f); P16_ADD(3, t_e, t_f);
            }
        }
        line_num = (uint16_t)(field + 1 + 2 * nl);
        service_line16(w, a, wl, SDV_SRV_END_FIELD);
        emit_rec(wl, frame_no, line_num, false, rec++);
    }
    if (f == a.end_file_frame) {
        line_num = (uint16_t)(line_num + 2);
        service_line16(w, a, wl, SDV_SRV_END_FILE);
        emit_rec(wl, frame_no, line_num, false, rec++);
    }
    line_num = (uint16_t)(line_num + 2);
    service_line16(w, a, wl, SDV_SRV_END_FRAME);
    v.q_odd = v.q_even = P16_LINES_PF;                                  /* :1643-1652 */
    v.q_pcm_odd = (uint16_t)(v.q_pcm_odd / P16_SUBLINES); v.q_pcm_even = (uint16_t)(v.q_pcm_even / P16_SUBLINES);
    v.q_bad_odd = (uint16_t)(v.q_bad_odd / P16_SUBLINES); v.q_bad_even = (uint16_t)(v.q_bad_even / P16_SUBLINES);
    {   /* the median of the frame's valid coordinates is what v2d_end_frame pushes into long_valid_coords (:1668-1682) */
        __syncthreads();        /* (keys stored to global memory by other lanes than the ones that read them now) */
        uint32_t mk = 0; bool pushed = median_keys(fv_keys, v.nfv, &mk);
        if (pushed) { const Coords mc = key_to_coords(mk, false); pushed = coords_valid(mc); }
        if (lane_id() == 0) { uint2 m; m.x = pushed ? mk : 0u; m.y = pushed ? 1u : 0u; a16.frame_med[f] = m; }
    }
    v2d_end_frame(v, a, lds.p.w, frame_no, fv_keys, fi_keys, &a.stats[f]);
    emit_rec(wl, frame_no, line_num, false, rec++);
    store_state16(w, lds, a16, f);
#if SDV_P16_STAMPS && !defined(SDV_EMU)
    if (lane_id() == 0) { unsigned long long *dst = (unsigned long long *)&a.stats[f]; for (int i = 0; i < 4; i++) dst[i] = p16_stamp[i]; }
#endif
}

/* ---- prediction of the incoming states (the PCM-1 model, pcm1_frames_device.h, with the longer window) ------------------------ */
struct PredictArgs16 { State16 *states; const PrescanRes *prescan; int first, hi; FrameArgs f; };

/* the median of a state's window of last valid coordinates (videotodigital.cpp:348-371), or false when it is empty */
__device__ inline bool last_valid_median16(const State16 &s0, sdv_coord *out)
{
    const int n = s0.s.n_last_valid > LV16 ? LV16 : s0.s.n_last_valid;
    if (n == 0 || s0.s.reset_stats) return false;
    uint32_t keys[LV16];
    for (int i = 0; i < n; i++) { const sdv_coord cc = i < COORD_HISTORY_DEPTH ? s0.s.last_valid[i] : s0.more[i - COORD_HISTORY_DEPTH]; keys[i] = coords_key(cc.data_start, cc.data_stop); }
    for (int i = 1; i < n; i++) { const uint32_t x = keys[i]; int j = i; while (j > 0 && keys[j - 1] > x) { keys[j] = keys[j - 1]; j--; } keys[j] = x; }
    out->data_start = key_start(keys[n / 2]); out->data_stop = key_stop(keys[n / 2]);
    return true;
}
/* sticky = the frames in between are taken to decode with the coordinates the stream already carries (the median of the window of
 * last valid coordinates) instead of the ones their own prescan finds: what happens when the first line of a field is marked bad
 * (no Header in this format, :1193-1211) and the worker falls back on its history for the lines behind it (:1431-1451) */
__device__ inline State16 predict_state16(const PredictArgs16 &a, int k, int base, bool sticky = false)
{
    const State16 s0 = a.states[base];
    State16 p = s0;
    sdv_coord carried; carried.data_start = 0; carried.data_stop = 0;
    const bool use_carried = sticky && last_valid_median16(s0, &carried);
    const uint8_t dbl = a.f.doubled;
    int n_long = s0.s.reset_stats ? 0 : s0.s.n_long_valid;
    sdv_coord lg[COORD_LONG_HISTORY];
    for (int i = 0; i < COORD_LONG_HISTORY; i++) lg[i] = s0.s.long_valid[i];
    bool touched = false;
    sdv_coord last; last.data_start = 0; last.data_stop = 0;
    uint8_t pref = prescan_ref_of(s0.s);
    int j0 = base; if (k - j0 > COORD_LONG_HISTORY + 1) j0 = k - (COORD_LONG_HISTORY + 1);
    for (int j = j0; j < k; j++) {
        if (!prescan_runs(a.f, j)) continue;
        uint32_t keys[COORD_CHECK_LINES]; uint8_t refs[COORD_CHECK_LINES]; int n = 0;
        for (int q = 0; q < COORD_CHECK_LINES; q++) {
            const PrescanRes r = a.prescan[(size_t)j * COORD_CHECK_LINES + q];
            if (r.valid) { keys[n] = coords_key(r.start, r.stop); refs[n] = r.ref; n++; }
            if (r.pad[1]) p.s.do_ref_lvl_sweep = a.f.mode == SDV_MODE_INSANE ? 1 : 0;
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
        p.s.reset_stats = 0;
        p.s.n_last_valid = LV16;
        for (int i = 0; i < COORD_HISTORY_DEPTH; i++) p.s.last_valid[i] = last;
        for (int i = 0; i < LV16 - COORD_HISTORY_DEPTH; i++) p.more[i] = last;
        p.s.n_long_valid = (uint8_t)n_long;
        for (int i = 0; i < COORD_LONG_HISTORY; i++) { if (i < n_long) p.s.long_valid[i] = lg[i]; else { p.s.long_valid[i].data_start = 0; p.s.long_valid[i].data_stop = 0; } }
        const uint16_t lm = dbl ? (uint16_t)((1u << COORD_HISTORY_DEPTH) - 1u) : 0, gm = dbl ? (uint16_t)((1u << n_long) - 1u) : 0;
        p.s.last_valid_doubled_mask_lo = (uint8_t)(lm & 0xFF); p.s.last_valid_doubled_mask_hi = (uint8_t)(lm >> 8);
        p.s.long_valid_doubled_mask = gm;
        p.s._pad[1] = (uint8_t)(pref ^ 128);
        p.s.bin.in_def_start = last.data_start; p.s.bin.in_def_stop = last.data_stop; p.s.bin.in_def_from_doubled = dbl;
    } else if (!s0.s.reset_stats && a.f.mode == SDV_MODE_DRAFT) {
        const int16_t cs = s0.s.bin.in_def_start, ce = s0.s.bin.in_def_stop;
        if (s0.s.bin.in_def_reference >= a.f.preset.min_ref_lvl && (cs != NO_COORD_LEFT && ce != NO_COORD_RIGHT && cs < ce)) {
            const int m = k - base;
            p.s.bin.in_def_from_doubled = dbl;
            p.s.n_last_valid = LV16;
            for (int i = 0; i < COORD_HISTORY_DEPTH; i++) { p.s.last_valid[i].data_start = cs; p.s.last_valid[i].data_stop = ce; }
            for (int i = 0; i < LV16 - COORD_HISTORY_DEPTH; i++) { p.more[i].data_start = cs; p.more[i].data_stop = ce; }
            const int total = (int)s0.s.n_long_valid + m;
            const int keep = total > COORD_LONG_HISTORY ? COORD_LONG_HISTORY : total, drop = total - keep;
            for (int i = 0; i < COORD_LONG_HISTORY; i++) {
                const int src = i + drop;
                if (i >= keep) { p.s.long_valid[i].data_start = 0; p.s.long_valid[i].data_stop = 0; }
                else if (src < (int)s0.s.n_long_valid) p.s.long_valid[i] = s0.s.long_valid[src];
                else { p.s.long_valid[i].data_start = cs; p.s.long_valid[i].data_stop = ce; }
            }
            p.s.n_long_valid = (uint8_t)keep;
            const uint16_t lm = dbl ? (uint16_t)((1u << COORD_HISTORY_DEPTH) - 1u) : 0, gm = dbl ? (uint16_t)((1u << keep) - 1u) : 0;
            p.s.last_valid_doubled_mask_lo = (uint8_t)(lm & 0xFF); p.s.last_valid_doubled_mask_hi = (uint8_t)(lm >> 8);
            p.s.long_valid_doubled_mask = gm;
        }
    }
    return p;
}
struct RepairArgs16 { PredictArgs16 p; const State16 *states_out; const int *list, *head; const uint8_t *sticky; int n; const uint2 *frame_med; };
/* Repair of a run of broken links (pcm16_frames_engine.inc).  Heads (head[i] == list[i]) take their predecessor's real outcome; they
 * come first in the list and are written by an earlier launch than the others read them.  A frame further into the run:
 *   DRAFT mode (the whole tuning is handed on): predicted again from its run's head - or, when that tells nothing new, its own
 *   predecessor's outcome;
 *   the other modes, first attempt (sticky[i]): predicted again from the head with the coordinates the stream carries (predict_state16);
 *   later attempts: its own predecessor's outcome (what a frame hands on depends little on what it was handed), except for the
 *   multi-frame history, which only passes through the frames - that is rebuilt from the head's true state and what the frames
 *   since then have pushed themselves, so that one wrong median does not need sixteen rounds to leave the chain. */
__device__ inline void repair_body16(const RepairArgs16 &a, int i)
{
    const int k = a.list[i], h = a.head[i];
    if (h == k) { a.p.states[k] = a.states_out[k - 1]; return; }
    if (a.p.f.mode == SDV_MODE_DRAFT || a.sticky[i]) {
        State16 p = predict_state16(a.p, k, h, a.p.f.mode != SDV_MODE_DRAFT);
        const State16 cur = a.p.states[k];
        uint32_t x[sizeof(State16) / 4], y[sizeof(State16) / 4];
        __builtin_memcpy(x, &p, sizeof(p));
        __builtin_memcpy(y, &cur, sizeof(cur));
        bool same = true;
        for (unsigned q = 0; q < sizeof(State16) / 4; q++) same = same && (x[q] == y[q]);
        a.p.states[k] = same ? a.states_out[k - 1] : p;
        return;
    }
    State16 p = a.states_out[k - 1];
    const State16 h_in = a.p.states[h];
    int n_long = h_in.s.reset_stats ? 0 : h_in.s.n_long_valid;
    sdv_coord lg[COORD_LONG_HISTORY];
    for (int q = 0; q < COORD_LONG_HISTORY; q++) lg[q] = h_in.s.long_valid[q];
    for (int j = h; j < k; j++) {
        const uint2 m = a.frame_med[j];
        if (!m.y) continue;
        if (n_long == COORD_LONG_HISTORY) { for (int q = 0; q + 1 < COORD_LONG_HISTORY; q++) lg[q] = lg[q + 1]; n_long--; }
        lg[n_long].data_start = key_start(m.x); lg[n_long].data_stop = key_stop(m.x); n_long++;
    }
    p.s.n_long_valid = (uint8_t)n_long;
    for (int q = 0; q < COORD_LONG_HISTORY; q++) { if (q < n_long) p.s.long_valid[q] = lg[q]; else { p.s.long_valid[q].data_start = 0; p.s.long_valid[q].data_stop = 0; } }
    p.s.long_valid_doubled_mask = a.p.f.doubled ? (uint16_t)((1u << n_long) - 1u) : 0;
    a.p.states[k] = p;
}
struct VerifyArgs16 { FrameArgs f; const State16 *states_in, *states_out; };
__device__ inline void verify_body16(const VerifyArgs16 &a, int k)
{
    a.f.flag[k] = link_holds16(a.f, k, a.states_out[k], a.states_in[k + 1]) ? VF_OK : VF_BREAK;
}

} // namespace sdvp16f

#ifndef SDV_P16_WAVES_PER_EU
#define SDV_P16_WAVES_PER_EU 3
#endif
/* two builds of the two kernels: MODE_INSANE (with the reference level sweep) and every other mode (process_line_p1) */
#define SDV_P16F_KERNELS(SUFFIX, INSANE) \
__global__ void __launch_bounds__(64, SDV_P16_WAVES_PER_EU) sdv_k_pcm16_prescan##SUFFIX(sdvp16f::FrameArgs16 a) \
{ \
    __shared__ sdvp16f::Lds16 lds; \
    const int i = (int)blockIdx.x, f = a.f.frame_list ? a.f.frame_list[i / sdvp16f::COORD_CHECK_LINES] : a.f.frame_lo + i / sdvp16f::COORD_CHECK_LINES; \
    sdvp16f::prescan_body<INSANE>(a, lds, f, i % sdvp16f::COORD_CHECK_LINES); \
} \
__global__ void __launch_bounds__(64, SDV_P16_WAVES_PER_EU) sdv_k_pcm16_frames_bin##SUFFIX(sdvp16f::FrameArgs16 a) \
{ \
    __shared__ sdvp16f::Lds16 lds; \
    const int f = a.f.frame_list ? a.f.frame_list[blockIdx.x] : a.f.frame_lo + (int)blockIdx.x; \
    sdvp16f::frame_body16<INSANE>(a, lds, f); \
}
SDV_P16F_KERNELS(, false)
SDV_P16F_KERNELS(_insane, true)
#ifndef SDV_P16F_LEAN_WAVES_PER_EU
#define SDV_P16F_LEAN_WAVES_PER_EU 4
#endif
/* the lean build of the frame kernel (every mode: what it holds - parts that read from what they inherit - is the same in all of them) */
__global__ void __launch_bounds__(64, SDV_P16F_LEAN_WAVES_PER_EU) sdv_k_pcm16_frames_lean(sdvp16f::FrameArgs16 a)
{
    __shared__ sdvp16f::Lds16 lds;
    const int f = a.f.frame_list ? a.f.frame_list[blockIdx.x] : a.f.frame_lo + (int)blockIdx.x;
    sdvp16f::frame_body16<false, true>(a, lds, f);
}
#ifndef SDV_EMU
__global__ void sdv_k_pcm16_predict(sdvp16f::PredictArgs16 a)
{
    const int k = a.first + 1 + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k < a.hi) a.states[k] = sdvp16f::predict_state16(a, k, a.first, true);
}
__global__ void sdv_k_pcm16_repair(sdvp16f::RepairArgs16 a, int lo, int hi)
{
    const int i = lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i < hi) sdvp16f::repair_body16(a, i);
}
__global__ void sdv_k_pcm16_verify(sdvp16f::VerifyArgs16 a)
{
    const int k = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (k + 1 < a.f.n_total) sdvp16f::verify_body16(a, k);
}
#endif
