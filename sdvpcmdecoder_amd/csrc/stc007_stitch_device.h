/*
 * stc007_stitch_device.h - HIP device code of the STC-007 stitch stage: STC007DataStitcher
 * (stc007datastitcher.cpp:3-7488) from binarized line records to PCMSamplePair records.
 *
 * One wave64 per stitcher turn ("step": frame A = segment k of the record stream, frame B = segment k+1):
 *   sdv_k_stitch_analyze  frame-local work, once per frame: findFramesTrim (:259-734), splitFramesToFields (:737-985),
 *                         getFieldResolution (:996-1211)
 *   sdv_k_stitch_step     everything that depends on the previous turn: findFieldStitching (:2929-4275) with
 *                         tryPadding / findPadding (:1417-2054), fillFrameForOutput (:4588-5387), the CWD pre-scan
 *                         (:5905-6452) and performDeinterleave (:6675-6885) with the PCMSamplePair output (:6525-6569)
 *   sdv_k_stitch_compact  packs the per-step outputs into the contiguous streams the reference's queues hold
 * The control flow of a turn is wave-uniform (every lane follows the reference's state machines with the same values);
 * the lanes split the data-block decodes (tryPadding windows, resolution probes, final deinterleave) and the line
 * copies.  The CWD pre-scan runs one lane per residue class modulo 16: a block at offset i only touches lines
 * i + 16k, so the 16 classes are independent and each is processed in the reference's order.
 *
 * What links the turns (previous frame's FrameAsmSTC007, padding counter, broken countdown, the last 112 assembled
 * lines) travels in StepChain; the engine (stitch_engine.inc) iterates the turns of a batch in parallel until every
 * turn has been run from exactly what its predecessor produced ("stitch stage").
 */
#ifndef SDV_STC007_STITCH_DEVICE_H
#define SDV_STC007_STITCH_DEVICE_H

#include "stc007_deint_device.h"

namespace sdvs {
using sdvd::Block;

enum { VID_UNKNOWN = 0, VID_PAL, VID_NTSC, VID_MAX };
enum { ORDER_UNK = 0, ORDER_TFF, ORDER_BFF, ORDER_MAX };
enum { LINES_PF_NTSC = 245, LINES_PF_PAL = 294, LINES_PF_MAX_PAL = 294 + 16, LINES_PF_MAX_NTSC = 294 - 32 };   /* config.h:80-81, stc007datastitcher.h */
enum { BUF_FIELD = 294, FIELD_PITCH = 296 /* lines from one field buffer to the next: BUF_FIELD rounded up to a whole number of 128-byte lines */, BUF_TRIM = 3 * 640 * 3, MIN_GOOD_LINES_PF = 245 - 8, MIN_FILL_LINES_PF = 56 };
enum { MAX_PADDING_14BIT = 32, MAX_PADDING_16BIT = 16, MAX_BURST_SILENCE = 8, MAX_BURST_BROKEN = 1, MAX_BURST_UNCH_DELTA = 8 };
enum { SRES_UNKNOWN = 0, SRES_14BIT, SRES_16BIT, SRES_MAX };
enum { DS_NO_DATA = 0, DS_SILENCE, DS_BROKE, DS_NO_PAD, DS_OK };
enum { CRC_SILENT = 0xA96A, CRC_POLY = 0x1021, CRC_INIT = 0xFFFF };
enum { MIN_DEINT = sdvd::MIN_DEINT_DATA, ILV = sdvd::INTERLEAVE_OFS };
enum { QCAP = 1024 };               /* conv_queue capacity per wave: 112 + 80 + 2 x 294 + 112 + CWD look-ahead 112 */
enum { PAIR_SLOT = 2352, FRASM_SLOT = 3 };   /* per-step output slots: 3 pairs x (QCAP - 112 - 128) blocks + 2 service pairs */
enum { MST_BOT_2 = 4, MED_LEN_OK = 3 };      /* STC007Line::MARK_ST_BOT_2 / MARK_ED_LEN_OK */
enum { NO_COORD_L = -32768, NO_COORD_R = 32767 };

/* ---- a line as the stitcher needs it (STC007Line minus the pixel side). 32 bytes. ----------------------------- */
enum { SL_FORCED_BAD = 1, SL_COORDS_VALID = 2, SL_BW_SET = 4, SL_MARKERS = 8 /* hasMarkers(): only the visualiser's feed looks at it */ };
struct alignas(16) SLine {
    uint32_t frame; uint16_t line; uint16_t words[9]; uint16_t calc_crc;
    uint16_t wcrc, wvalid;          /* bit i = word_crc[i] / word_valid[i], i = 0..8 */
    uint8_t flags, ref_level;
};

__device__ inline uint16_t crc16_update(uint16_t crc, uint16_t data, int bits)   /* pcmline.cpp:461-487 */
{
    for (int i = 0; i < bits; i++) {
        bool msb = (crc & 0x8000) != 0, inb = (data & (1 << (bits - 1))) != 0;
        crc = (uint16_t)(crc << 1);
        if (msb != inb) crc ^= CRC_POLY;
        data = (uint16_t)(data << 1);
    }
    return crc;
}
__device__ inline uint16_t crc_words(const uint16_t *w)   /* stc007line.cpp:245-251 */
{
    uint16_t crc = CRC_INIT;
    for (int i = 0; i < 8; i++) crc = crc16_update(crc, w[i], 14);
    return crc;
}
__device__ inline SLine sline_empty(uint32_t frame, uint16_t line)   /* STC007Line::clear, stc007line.cpp:64-93 */
{
    SLine l;
    l.frame = frame; l.line = line;
    for (int i = 0; i < 8; i++) l.words[i] = 0;
    l.words[8] = (uint16_t)~CRC_SILENT; l.calc_crc = CRC_SILENT;
    l.wcrc = l.wvalid = 0; l.flags = 0; l.ref_level = 0;
    return l;
}
/* what toLine() of INTEGRATION.md makes of a record (STC007Line setters + applyCRCStatePerWord, stc007line.cpp:198-204) */
__device__ inline SLine sline_from_rec(const sdv_line_rec &r)
{
    if (r.service_type != SDV_SRV_NO && r.service_type != SDV_SRV_CTRL_BLOCK) {
        SLine l = sline_empty(r.frame_number, r.line_number);      /* filler: PCMLine::clear() on a cleared line */
        l.calc_crc = 0;
        return l;
    }
    SLine l;
    l.frame = r.frame_number; l.line = r.line_number;
    for (int i = 0; i < 9; i++) l.words[i] = r.words[i];
    l.calc_crc = r.calc_crc;
    bool forced = (r.flags & SDV_LF_FORCED_BAD) != 0;
    bool cv = r.data_start != NO_COORD_L && r.data_stop != NO_COORD_R && r.data_start < r.data_stop;
    l.flags = (uint8_t)((forced ? SL_FORCED_BAD : 0) | (cv ? SL_COORDS_VALID : 0) | ((r.flags & SDV_LF_BW_SET) ? SL_BW_SET : 0) |
                        ((r.mark_st_stage == MST_BOT_2 && r.mark_ed_stage == MED_LEN_OK) ? SL_MARKERS : 0));
    bool v = !forced && l.calc_crc == l.words[8];
    l.wcrc = l.wvalid = v ? 0x1FF : 0;
    l.ref_level = r.ref_level;
    return l;
}
__device__ inline uint16_t sl_word(const SLine &l, int i) { uint16_t r = 0; for (int k = 0; k < 8; k++) r = k == i ? l.words[k] : r; return r; }
__device__ inline void sl_set_word(SLine &l, int i, uint16_t v) { for (int k = 0; k < 8; k++) l.words[k] = k == i ? v : l.words[k]; }
__device__ inline bool crc_valid_if(const SLine &l) { return l.calc_crc == l.words[8]; }                 /* isCRCValidIgnoreForced */
__device__ inline bool crc_valid(const SLine &l) { return !(l.flags & SL_FORCED_BAD) && crc_valid_if(l); }   /* isCRCValid */
__device__ inline sdv_deint_line view(const SLine &l)     /* what STC007Deinterleaver::setWordData reads of a line */
{
    sdv_deint_line d;
    d.frame_number = l.frame; d.line_number = l.line;
    for (int i = 0; i < 8; i++) d.words[i] = l.words[i];
    d.word_crc_ok = (l.flags & SL_FORCED_BAD) ? 0 : (uint8_t)(l.wcrc & 0xFF);
    bool fixed = crc_valid(l) && ((~l.wcrc & l.wvalid & 0xFF) != 0);                                      /* isFixedByCWD, stc007line.cpp:628-641 */
    d.flags = (uint8_t)((fixed ? SDV_DL_FIXED_BY_CWD : 0) | (((l.flags & SL_COORDS_VALID) && (l.flags & SL_BW_SET)) ? SDV_DL_COORDS_BW_OK : 0));
    return d;
}

/* ---- FrameAsmSTC007 (frametrimset.h:116-275, frametrimset.cpp:383-959) ------------------------------------------ */
struct Frasm {
    uint32_t frame_number;
    uint16_t odd_std_lines, even_std_lines, odd_data_lines, even_data_lines, odd_valid_lines, even_valid_lines;
    uint16_t odd_top_data, odd_bottom_data, even_top_data, even_bottom_data, odd_sample_rate, even_sample_rate;
    uint16_t blocks_total, blocks_drop, samples_drop, inner_padding, outer_padding;
    uint16_t blocks_broken_field, blocks_broken_seam, blocks_fix_p, blocks_fix_q, blocks_fix_cwd;
    uint8_t field_order, odd_ref, even_ref, service_type, video_standard, tff_cnt, bff_cnt, odd_resolution, even_resolution;
    uint8_t odd_emphasis, even_emphasis, order_preset, order_guessed, trim_ok, inner_padding_ok, outer_padding_ok;
    uint8_t inner_silence, outer_silence, vid_std_preset, vid_std_guessed;
    int8_t ctrl_index, ctrl_hour, ctrl_minute, ctrl_second, ctrl_field;
    uint8_t _pad[3];                /* no implicit padding: the hand-over between turns is compared as words */
};
static_assert(sizeof(Frasm) == 76, "Frasm layout");
__host__ __device__ inline void frasm_clear_asm_stats(Frasm &f)
{
    f.odd_ref = f.even_ref = 0; f.blocks_total = f.blocks_drop = f.samples_drop = 0;
    f.blocks_broken_field = f.blocks_broken_seam = f.blocks_fix_p = f.blocks_fix_q = f.blocks_fix_cwd = 0;
}
__host__ __device__ inline void frasm_clear_misc(Frasm &f)
{
    f.odd_std_lines = f.even_std_lines = f.odd_data_lines = f.even_data_lines = f.odd_valid_lines = f.even_valid_lines = 0;
    f.odd_sample_rate = f.even_sample_rate = 0;
    f.field_order = ORDER_UNK; f.odd_emphasis = f.even_emphasis = 0; f.order_preset = f.order_guessed = 0; f.service_type = 0;
    f.video_standard = VID_UNKNOWN; f.tff_cnt = f.bff_cnt = 0; f.odd_resolution = f.even_resolution = 0;
    f.inner_padding = f.outer_padding = 0;
    f.trim_ok = 0; f.inner_padding_ok = f.outer_padding_ok = 0; f.inner_silence = f.outer_silence = 1;
    f.vid_std_preset = f.vid_std_guessed = 0;
    f.ctrl_index = f.ctrl_hour = f.ctrl_minute = f.ctrl_second = f.ctrl_field = -1;
    frasm_clear_asm_stats(f);
}
__host__ __device__ inline void frasm_clear(Frasm &f)
{
    f.frame_number = 0; f.odd_top_data = 0; f.odd_bottom_data = 0xFFFF; f.even_top_data = 0; f.even_bottom_data = 0xFFFF;
    f._pad[0] = f._pad[1] = f._pad[2] = 0;
    frasm_clear_misc(f);
}
__device__ inline bool order_set(const Frasm &f) { return f.field_order == ORDER_TFF || f.field_order == ORDER_BFF; }
__device__ inline void preset_order(Frasm &f, uint8_t o) { f.order_preset = 1; f.order_guessed = 0; f.field_order = o; }
__device__ inline void set_order_unknown(Frasm &f) { if (!f.order_preset) { f.field_order = ORDER_UNK; f.order_guessed = 0; } }
__device__ inline void set_order(Frasm &f, uint8_t o) { if (!f.order_preset) f.field_order = o; }
__device__ inline void frasm_to_pod(const Frasm &f, sdv_frame_asm &o)
{
    o.frame_number = f.frame_number;
    o.odd_std_lines = f.odd_std_lines; o.even_std_lines = f.even_std_lines; o.odd_data_lines = f.odd_data_lines; o.even_data_lines = f.even_data_lines;
    o.odd_valid_lines = f.odd_valid_lines; o.even_valid_lines = f.even_valid_lines;
    o.odd_top_data = f.odd_top_data; o.odd_bottom_data = f.odd_bottom_data; o.even_top_data = f.even_top_data; o.even_bottom_data = f.even_bottom_data;
    o.odd_sample_rate = f.odd_sample_rate; o.even_sample_rate = f.even_sample_rate;
    o.blocks_total = f.blocks_total; o.blocks_drop = f.blocks_drop; o.samples_drop = f.samples_drop;
    o.inner_padding = f.inner_padding; o.outer_padding = f.outer_padding;
    o.blocks_broken_field = f.blocks_broken_field; o.blocks_broken_seam = f.blocks_broken_seam;
    o.blocks_fix_p = f.blocks_fix_p; o.blocks_fix_q = f.blocks_fix_q; o.blocks_fix_cwd = f.blocks_fix_cwd;
    o.field_order = f.field_order; o.odd_ref = f.odd_ref; o.even_ref = f.even_ref; o.service_type = f.service_type;
    o.video_standard = f.video_standard; o.tff_cnt = f.tff_cnt; o.bff_cnt = f.bff_cnt; o.odd_resolution = f.odd_resolution; o.even_resolution = f.even_resolution;
    o.flags = (uint8_t)((f.order_preset ? SDV_FA_ORDER_PRESET : 0) | (f.order_guessed ? SDV_FA_ORDER_GUESSED : 0) | (f.trim_ok ? SDV_FA_TRIM_OK : 0) |
                        (f.inner_padding_ok ? SDV_FA_INNER_OK : 0) | (f.outer_padding_ok ? SDV_FA_OUTER_OK : 0) | (f.inner_silence ? SDV_FA_INNER_SILENCE : 0) |
                        (f.outer_silence ? SDV_FA_OUTER_SILENCE : 0) | (f.vid_std_preset ? SDV_FA_VID_STD_PRESET : 0));
    o.flags2 = (uint8_t)((f.odd_emphasis ? SDV_FA2_ODD_EMPHASIS : 0) | (f.even_emphasis ? SDV_FA2_EVEN_EMPHASIS : 0) | (f.vid_std_guessed ? SDV_FA2_VID_STD_GUESSED : 0));
    o.ctrl_index = f.ctrl_index; o.ctrl_hour = f.ctrl_hour; o.ctrl_minute = f.ctrl_minute; o.ctrl_second = f.ctrl_second; o.ctrl_field = f.ctrl_field;
}

/* ---- per-frame facts that do not depend on other frames (sdv_k_stitch_analyze) ---------------------------------- */
enum { FL_NEW_FILE = 1, FL_END_FILE = 2, FL_TRIM_OK = 4, FL_BAD_NUMBERS = 8 };
struct FrameLocal {
    uint32_t frame_number;          /* of the END_FRAME line that closes the segment */
    uint32_t seg_start, seg_len;    /* the frame's records, END_FRAME included */
    uint16_t top[2], bottom[2];     /* [0] odd lines, [1] even lines */
    uint16_t data_lines[2], valid_lines[2];
    uint16_t max_line;
    uint8_t ref[2];
    uint8_t flags;
    uint8_t field_res[2];           /* getFieldResolution: SRES_* */
    int8_t ctrl[5];
};

/* the record stream of a call: what was left over from the previous call, then the caller's records */
struct RecSrc {
    const sdv_line_rec *carry; uint32_t n_carry; const sdv_line_rec *recs;
    __device__ inline const sdv_line_rec &at(uint32_t i) const { return i < n_carry ? carry[i] : recs[i - n_carry]; }
};

/* STC007DataStitcher settings as the kernels need them */
struct Cfg {
    uint8_t preset_video_mode, preset_field_order, preset_audio_res, en_p, en_q, en_cwd, m2, ignore_crc;
    uint8_t max_unch_14, max_unch_16, mask_seams, broken_mask_dur, fix_cut_above, _pad;
    uint16_t preset_sample_rate;
};

/* ---- block helpers (STC007DataBlock, stc007datablock.cpp) ------------------------------------------------------- */
__device__ inline uint8_t ind_mask(const Block &b) { return b.resolution == SDV_RES_16BIT ? 0x7F : 0xFF; }
__device__ inline int errors_audio_fixed(const Block &b) { return __popc((uint32_t)(~b.word_valid & 0x3F)); }
__device__ inline bool blk_valid(const Block &b) { return (b.word_valid & 0x3F) == 0x3F; }                                   /* isBlockValid :305-312 */
__device__ inline int errors_total_cwd(const Block &b) { return __popc((uint32_t)(~b.line_crc & ~b.cwd_fixed & ind_mask(b))); }   /* :653-678 */
__device__ inline bool can_force_check(const Block &b)                                                                         /* :246-272 */
{
    if (b.audio_state == SDV_AUD_BROKEN) return false;
    return b.resolution == SDV_RES_14BIT ? errors_total_cwd(b) <= 1 : errors_total_cwd(b) == 0;
}
__device__ inline int16_t get_sample(const Block &b, int i, bool m2)                                                          /* :507-562 */
{
    if (!m2) return b.resolution == SDV_RES_16BIT ? (int16_t)b.w(i) : (int16_t)(b.w(i) << 2);
    uint16_t w = b.w(i);
    if ((w & (1 << 13)) == 0) w = (uint16_t)(w << 3);
    else {
        bool pos = (w & (1 << 12)) == 0;
        w = (uint16_t)(w & ~(1 << 13));
        if (!pos) w |= (1 << 15) | (1 << 14) | (1 << 13);
    }
    return (int16_t)w;
}
__device__ inline bool blk_silent(const Block &b, bool m2) { for (int i = 0; i < 6; i++) if (get_sample(b, i, m2) != 0) return false; return true; }   /* :465-478 */
__device__ inline void mark_unsafe(Block &b)                                                                                   /* :168-201 */
{
    if (b.audio_state == SDV_AUD_BROKEN) return;
    uint8_t m = ind_mask(b);
    b.word_valid = (uint8_t)((b.word_valid & ~m) | (b.line_crc & m));
    b.line_crc &= (uint8_t)~m; b.cwd_fixed &= (uint8_t)~m;
    b.audio_state = SDV_AUD_ORIG; b.cwd_applied = false;
}

/* resolution rules for a seam (stc007datastitcher.cpp:1214-1269) */
__device__ inline uint8_t res_mode_for_seam(uint8_t r1, uint8_t r2)
{
    uint8_t fin = SDV_RES_MODE_16BIT_AUTO;
    if (r1 == r2) { fin = r1; if (r1 == SDV_RES_MODE_14BIT_AUTO) fin = SDV_RES_MODE_14BIT; else if (r1 == SDV_RES_MODE_16BIT_AUTO) fin = SDV_RES_MODE_16BIT; }
    else if (r1 == SDV_RES_MODE_14BIT) { if (r2 == SDV_RES_MODE_14BIT_AUTO) fin = SDV_RES_MODE_14BIT_AUTO; }
    else if (r1 == SDV_RES_MODE_14BIT_AUTO) { if (r2 == SDV_RES_MODE_14BIT) fin = SDV_RES_MODE_14BIT_AUTO; }
    else if (r1 == SDV_RES_MODE_16BIT) { if (r2 == SDV_RES_MODE_14BIT) fin = SDV_RES_MODE_14BIT_AUTO; }
    return fin;
}
__device__ inline uint8_t res_for_seam(uint8_t r1, uint8_t r2)
{
    uint8_t fin = res_mode_for_seam(r1, r2);
    return (fin == SDV_RES_MODE_16BIT || fin == SDV_RES_MODE_16BIT_AUTO) ? SDV_RES_16BIT : SDV_RES_14BIT;
}

/* effective deinterleaver switches after the reference's setter coupling (stc007deinterleaver.cpp:210-260) */
__device__ inline sdv_deint_settings deint_cfg(uint8_t res_mode, bool ignore_crc, bool force, bool p, bool q, bool cwd)
{
    sdv_deint_settings st;
    st.res_mode = res_mode; st.ignore_crc = ignore_crc; st.force_ecc_check = force;
    st.en_q_code = q; st.en_p_code = p || q; st.en_cwd = cwd;
    st._pad[0] = st._pad[1] = 0;
    return st;
}

/* ---- line sources for processBlock ------------------------------------------------------------------------------- */
/* one of the four field buffers (frame1_odd/even, frame2_odd/even) */
struct Field {
    const SLine *lines; int size;               /* size = data lines */
    __device__ inline SLine get(int i) const { return lines[i]; }
};
/* the field buffers of all frames: BUF_FIELD lines (at a pitch of FIELD_PITCH) for (frame k, odd) then (frame k, even), written by the analysis pass */
__device__ inline const SLine *field_lines(const SLine *fields, uint32_t k, int parity) { return fields + ((size_t)k * 2 + (size_t)parity) * FIELD_PITCH; }
struct FieldSrc {
    Field f;
    __device__ inline sdv_deint_line line(size_t i) const { return view(f.get((int)i)); }
};
/* tryPadding's queue: tail of field 1, `pad` empty lines, head of field 2 (stc007datastitcher.cpp:1455-1520) */
struct PadQueue {
    Field f1, f2; int a0, n0, npad, n2; uint32_t pad_frame; uint16_t pad_line0; bool m2;
    __device__ inline int size() const { return n0 + npad + n2; }
    __device__ inline const SLine *src(int i) const        /* where place i is read from; NULL: a padding line */
    {
        if (i < n0) return f1.lines + (a0 + i);
        if (i < n0 + npad) return nullptr;
        return f2.lines + (i - n0 - npad);
    }
    __device__ inline SLine get(int i) const
    {
        if (i < n0) return f1.get(a0 + i);
        if (i < n0 + npad) {
            SLine e = sline_empty(pad_frame, (uint16_t)(pad_line0 + 2 * (i - n0)));
            if (m2) { for (int w = 0; w < 8; w++) e.words[w] = 1 << 13; e.calc_crc = crc_words(e.words); }
            return e;
        }
        return f2.get(i - n0 - npad);
    }
    __device__ inline sdv_deint_line line(size_t i) const { return view(get((int)i)); }
};
struct WsSrc {
    const SLine *q;
    __device__ inline sdv_deint_line line(size_t i) const { return view(q[i]); }
};

/* ================================================================================================================== */
/* sdv_k_stitch_analyze: one wave per frame segment                                                                    */
/* ================================================================================================================== */
struct FrameBrief { uint32_t frame_number; uint8_t flags, field_res[2], _pad; };     /* what the host needs of a FrameLocal */
struct AnalyzeArgs {
    RecSrc src; const uint32_t *seg_end; uint32_t n_seg;      /* seg_end[k] = index of the k-th END_FRAME record */
    uint32_t *ctl;                                            /* pipelined call: n_seg is the launch width, ctl[CTL_NSEG] the count (NULL otherwise) */
    Cfg cfg; FrameLocal *fl; FrameBrief *brief; SLine *fields;
    unsigned long long *timing;     /* optional: 8 cycle stamps per frame (SDV_STITCH_TIMING=1), NULL otherwise */
    /* the fused entry: segment k = direct_ofs + f is frame f of the binarize call; where direct[f].flag is set the frame kernel has put the frame's lines into
     * fields itself and left no records (stc007_device.h, FrameArgs::direct_fields) */
    const sdv::DirectFrame *direct; uint32_t direct_ofs, direct_n;
};
static_assert((int)sdv::DSL_FORCED_BAD == (int)SL_FORCED_BAD && (int)sdv::DSL_COORDS_VALID == (int)SL_COORDS_VALID && (int)sdv::DSL_BW_SET == (int)SL_BW_SET, "the frame kernel writes SLine::flags");

__device__ inline uint64_t lanemask_lt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }
/* the turn's control values are the same in every lane: say so, and they live in scalar registers */
__device__ inline uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <class T> __device__ inline void make_uniform(T &x)
{
    static_assert(sizeof(T) % 4 == 0, "whole words");
    uint32_t w[sizeof(T) / 4];
    __builtin_memcpy(w, &x, sizeof(T));
    for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = uni(w[i]);
    __builtin_memcpy(&x, w, sizeof(T));
}

/* A window of the queue in LDS for the loops that decode block i from lines i, i + 16, ..., i + 112, 64 blocks a step: the 176
 * lines of a step as sdv_deint_line (24 B) in a ring of 256, so that the 64 lines of the next step can arrive while this one is
 * decoded.  One round trip to the queue (L2 / Infinity Cache) per step instead of eight gathers per lane. */
enum { RING = 256, RING_SPAN = 64 + MIN_DEINT, RING_SMALL = 192 };    /* RING_SMALL: the analysis kernel's, in the LDS its first phases used */
/* The lanes of the one wave of a workgroup exchange data through LDS: the hardware runs a wave's LDS operations in order, all that is
 * needed is that the compiler keeps them in order too.  (__syncthreads would also wait for every load in flight - the prefetch.) */
#ifdef SDV_EMU
#define SDV_LDS_WAVE_SYNC() __syncthreads()
#else
#define SDV_LDS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

template <int N> struct RingSrcN {
    const sdv_deint_line *ring;
    __device__ inline const sdv_deint_line &line(size_t i) const { return ring[(uint32_t)i % (uint32_t)N]; }
};
typedef RingSrcN<RING> RingSrc;

/* What getFieldResolution asks of a block decoded in one of the two modes - "valid, checkable, not silent" and "BROKEN" - has a short form for the
 * block a tape that plays is made of: every word passed its CRC (in 16-bit mode: and the S bits of its line).  processBlock (forced check, P code,
 * no Q code, no CWD, fixed resolution) then only compares the P word with the XOR of the six samples: equal -> the block is valid as it stands (and
 * silent when all six are zero), not equal -> BROKEN (stc007deinterleaver.cpp:286-1123 via the first short cut of sdvd::process_block). */
/* STC007DataStitcher::getFieldResolution (stc007datastitcher.cpp:996-1211) for one field buffer, blocks split over the lanes.
 * The field is staged once in LDS in the form the trial needs of a line - per word k < 7 sixteen bits (the 14-bit word and the two S bits a 16-bit sample
 * takes from the line's Q word), and the line's per-word CRC flags - transposed, so that the 64 blocks of a step read consecutive halfwords: block i
 * takes word k of line i + 16 k.  A block whose words all passed their CRC is decided from those (probe of the clean block, above: the P word against
 * the XOR of the samples, in either resolution); only a block with a failed word goes through processBlock itself, on lines gathered from the field
 * buffer.  (Through round 4 every block of every field went through two processBlock calls on lines kept in an LDS ring: 130 000 cycles a frame.) */
enum { RES_PITCH = BUF_FIELD + 2 };
__device__ inline uint8_t field_resolution(const Cfg &cfg, const Field &f, int lane, uint32_t *lds_words, unsigned long long *tm = nullptr)
{
    if (cfg.preset_audio_res == SRES_14BIT) return SRES_14BIT;
    if (cfg.preset_audio_res == SRES_16BIT) return SRES_16BIT;
    if (f.size > BUF_FIELD || f.size <= MIN_DEINT) return SRES_UNKNOWN;
    const int test = f.size - MIN_DEINT;
    uint16_t *e16 = (uint16_t *)lds_words;            /* [k][line] for k < 7, then the flags: bits 0..7 word_crc_ok, bit 15: not a line the short form can take */
    FieldSrc src; src.f = f;
    SDV_LDS_WAVE_SYNC();                                              /* whoever used the staging area before is done with it */
    /* (the lines of a lane asked for together, ahead of the first use: one trip to the field buffer instead of five, one behind the other) */
    constexpr int STAGE_TRIPS = (BUF_FIELD + 63) / 64;
    SLine ls[STAGE_TRIPS];
#pragma unroll
    for (int u = 0; u < STAGE_TRIPS; u++) { const int j = lane + 64 * u; if (j < f.size) ls[u] = f.get(j); }
#pragma unroll
    for (int u = 0; u < STAGE_TRIPS; u++) {
        const int j = lane + 64 * u;
        if (j >= f.size) break;
        const sdv_deint_line d = view(ls[u]);
        uint32_t odd = cfg.m2 ? 1u : 0u;
#pragma unroll
        for (int k = 0; k < 7; k++) {
            odd |= (uint32_t)d.words[k] >> 14;
            e16[k * RES_PITCH + j] = (uint16_t)((d.words[k] & 0x3FFFu) | (((uint32_t)d.words[sdvd::WORD_Q0] >> (12 - 2 * k)) & 3u) << 14);
        }
        e16[7 * RES_PITCH + j] = (uint16_t)(d.word_crc_ok | (odd ? 0x8000u : 0u));
    }
    SDV_LDS_WAVE_SYNC();
#ifndef SDV_EMU
    if (tm && lane == 0) tm[5] = (unsigned long long)__builtin_readcyclecounter();
    unsigned slow_chunks = 0;
#endif
    uint16_t res[2] = { 0, 0 };
    for (int c = 0; c * 64 < test; c++) {
        const int i = c * 64 + lane;
        const bool act = i < test;
        /* per mode: the words that failed (bit k), the XOR of the seven words, the six samples */
        uint32_t bad14 = 0, bad16 = 0, plain = act ? 1u : 0u, p14 = 0, p16 = 0, w14v[6] = { 0, 0, 0, 0, 0, 0 }, any16 = 0;
        if (act) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t fl = e16[7 * RES_PITCH + i + ILV * k];
                const uint32_t ck = (fl >> k) & 1u, cq = (fl >> sdvd::WORD_Q0) & 1u;
                plain &= (fl >> 15) ^ 1u;
                bad14 |= (ck ^ 1u) << k;
                if (k < 7) {
                    bad16 |= ((ck & cq) ^ 1u) << k;
                    const uint32_t e = e16[k * RES_PITCH + i + ILV * k];
                    const uint32_t w14 = e & 0x3FFFu, w16 = ((w14 << 2) + (e >> 14)) & 0xFFFFu;
                    p14 ^= w14; p16 ^= w16;
                    if (k < 6) { w14v[k] = w14; any16 |= w16; }
                }
            }
        }
        /* The short forms (processBlock with forced check, P code, no Q code, no CWD, one attempt - stc007deinterleaver.cpp:286-1123):
         *   no failed word        P syndrome 0 -> valid as it stands; else BROKEN
         *   one failed word, 14 bit: an audio word is restored from P (never BROKEN, one error may still be checked: counts unless silent); the P word:
         *                         nothing to check, the audio stands; the Q word: as with no failed word
         *   one failed word, 16 bit: restored or left alone, but a block with an error cannot be checked: neither counted nor BROKEN
         * anything else goes through processBlock. */
        const int n14 = __popc(bad14), n16 = __popc(bad16);
        bool g14 = false, k14 = false, g16 = false, k16 = false;
        const bool short14 = act && plain && n14 <= 1, short16 = act && plain && n16 <= 1;
        if (short14) {
            uint32_t any = 0;
            const int k0 = n14 ? __ffs((int)bad14) - 1 : 8;
#pragma unroll
            for (int k = 0; k < 6; k++) any |= (k == k0) ? (p14 ^ w14v[k]) : w14v[k];      /* (a failed audio word: the value the P code restores) */
            if (k0 < 7) g14 = any != 0;
            else { k14 = p14 != 0; g14 = !k14 && any != 0; }
        }
        if (short16) { if (n16 == 0) { k16 = p16 != 0; g16 = !k16 && any16 != 0; } }
        uint64_t good[2], brk[2];
        const bool slow = act && !(short14 && short16);               /* the lines themselves, from the field buffer */
        sdvd::Lines8 l8;
#ifdef SDV_EMU
        if (act) sdvd::gather8(src, (size_t)i, l8);                   /* (the emulator checks every short form against processBlock) */
#else
        if (slow) sdvd::gather8(src, (size_t)i, l8);
        if (__ballot(slow)) slow_chunks++;
#endif
        for (int m = 0; m < 2; m++) {
            bool g = m == 0 ? g14 : g16, k = m == 0 ? k14 : k16;
            const bool short_form = m == 0 ? short14 : short16;
#ifdef SDV_EMU
            if (act) {
#else
            if (act && !short_form) {
#endif
                Block b;
                sdvd::process_block(deint_cfg(m == 0 ? SDV_RES_MODE_14BIT : SDV_RES_MODE_16BIT, false, true, true, false, false), l8, 0, b);
                const bool g2 = blk_valid(b) && can_force_check(b) && !blk_silent(b, cfg.m2), k2 = b.audio_state == SDV_AUD_BROKEN;
#ifdef SDV_EMU
                if (short_form && (g != g2 || k != k2)) {
                    fprintf(stderr, "field_resolution: short form of block %d mode %d says (%d, %d), processBlock (%d, %d); failed words %x / %x\n", i, m, (int)g, (int)k, (int)g2, (int)k2, bad14, bad16);
                    abort();
                }
#endif
                g = g2; k = k2;
            }
            good[m] = __ballot(g); brk[m] = __ballot(k);
        }
        /* The counters (:1147-1168): a block that counts adds one, a BROKEN one takes one away - never below zero.  Block after block that is
         * x <- max(x + d, 0); over a step of 64 blocks: x + S_n or S_n - min S_k, whichever is larger (S_k the running sum of the d up to block k) -
         * every lane its S_k from two population counts, the minimum across the wave.  (A 14-bit tape gives a BROKEN block at nearly every place of
         * its 16-bit trial: the loop over the blocks ran in every step, 15 000 cycles each.) */
        int cnt = test - c * 64; if (cnt > 64) cnt = 64;
#pragma unroll
        for (int m = 0; m < 2; m++) {
            if (brk[m] == 0) { res[m] = (uint16_t)(res[m] + __popcll(good[m])); continue; }
            const uint64_t upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
            int sk = (int)__popcll(good[m] & upto) - (int)__popcll(brk[m] & upto);
            const int sn = __shfl(sk, cnt - 1);
            if (lane >= cnt) sk = 0x7FFFFFFF;
            for (int d = 1; d < 64; d <<= 1) { const int o = __shfl(sk, lane ^ d); sk = o < sk ? o : sk; }
            const int a = (int)res[m] + sn, b = sn - sk;
            res[m] = (uint16_t)(a > b ? a : b);
        }
    }
#ifndef SDV_EMU
    if (tm && lane == 0) { tm[6] = (unsigned long long)__builtin_readcyclecounter(); tm[7] = slow_chunks; }
#endif
    if (res[0] > (ILV * 2)) {
        uint16_t t = (uint16_t)(res[1] * 128);
        t = (uint16_t)(t / res[0]);
        return t > 32 ? SRES_16BIT : SRES_14BIT;
    }
    return SRES_UNKNOWN;
}

/* What the trim search and the field split need of one record, one word: line number | flags | reference level.  The frame's
 * records are read from HBM once (four 64-record chunks of three 16-byte loads in flight per turn) and kept in LDS in this form;
 * only the lines that go into a field buffer are read again (for their nine words). */
enum { AM_G = 1 << 16,          /* data line, CRC valid, not forced bad */
       AM_CI = 1 << 17,         /* data line, CRC valid ignoring "forced bad" */
       AM_MK = 1 << 18,         /* data line with both markers found */
       AM_DATA = 1 << 19, AM_FILLER = 1 << 20, AM_NEW_FILE = 1 << 21, AM_END_FILE = 1 << 22, AM_CTRL = 1 << 23,
       ANALYZE_LDS = 1024,      /* records staged per frame; longer segments compute the word from the record on every access */
       ANALYZE_LDS_WORDS = 1184 };    /* ... and the same LDS holds the staged field of the resolution trials afterwards (8 x RES_PITCH halfwords) */
struct Rec48 { uint4 q0, q1, q2; };      /* an sdv_line_rec as three 16-byte loads */
static_assert(sizeof(sdv_line_rec) == 48, "record layout");
/* sline_from_rec() of a record held in registers: the first 26 bytes of the two layouts are the same */
__device__ inline SLine sline_from_raw(const Rec48 &r)
{
    static_assert(sizeof(SLine) == 32, "line layout");
    const uint32_t srv = r.q2.z >> 24, flags = (r.q2.w >> 16) & 0xFF, mst = r.q2.w & 0xFF, med = (r.q2.w >> 8) & 0xFF, ref = r.q2.y >> 24;
    uint32_t w[8];
    if (srv != SDV_SRV_NO && srv != SDV_SRV_CTRL_BLOCK) {      /* filler: PCMLine::clear() on a cleared line */
        w[0] = r.q0.x; w[1] = r.q0.y & 0xFFFFu; w[2] = w[3] = w[4] = 0; w[5] = (uint32_t)(uint16_t)~CRC_SILENT << 16; w[6] = 0; w[7] = 0;
    } else {
        const int16_t ds = (int16_t)(r.q1.z >> 16), de = (int16_t)(r.q1.w & 0xFFFF);
        const bool forced = (flags & SDV_LF_FORCED_BAD) != 0;
        const bool cv = ds != NO_COORD_L && de != NO_COORD_R && ds < de;
        const uint32_t lf = (forced ? SL_FORCED_BAD : 0u) | (cv ? SL_COORDS_VALID : 0u) | ((flags & SDV_LF_BW_SET) ? SL_BW_SET : 0u) | ((mst == MST_BOT_2 && med == MED_LEN_OK) ? SL_MARKERS : 0u);
        const uint32_t calc = r.q1.z & 0xFFFFu, w8 = r.q1.y >> 16;
        const uint32_t wm = (!forced && calc == w8) ? 0x1FFu : 0u;
        w[0] = r.q0.x; w[1] = r.q0.y; w[2] = r.q0.z; w[3] = r.q0.w; w[4] = r.q1.x; w[5] = r.q1.y;
        w[6] = calc | (wm << 16); w[7] = wm | (lf << 16) | (ref << 24);
    }
    SLine l; __builtin_memcpy(&l, w, 32);
    return l;
}
__device__ inline uint32_t rec_meta(const Rec48 &r, uint32_t fnum, bool &bad_number)
{
    const uint32_t line = r.q0.y & 0xFFFF, w8 = r.q1.y >> 16, crc = r.q1.z & 0xFFFF, srv = r.q2.z >> 24;
    const uint32_t mst = r.q2.w & 0xFF, med = (r.q2.w >> 8) & 0xFF, flags = (r.q2.w >> 16) & 0xFF, ref = r.q2.y >> 24;
    bad_number = r.q0.x != fnum;
    uint32_t m = line;
    if (srv == SDV_SRV_NO) {
        const bool ci = crc == w8;
        m |= AM_DATA | (ci ? AM_CI : 0u) | ((ci && !(flags & SDV_LF_FORCED_BAD)) ? AM_G : 0u) | ((mst == MST_BOT_2 && med == MED_LEN_OK) ? AM_MK : 0u) | (ref << 24);
    } else if (srv == SDV_SRV_FILLER) m |= AM_FILLER;
    else if (srv == SDV_SRV_NEW_FILE) m |= AM_NEW_FILE;
    else if (srv == SDV_SRV_END_FILE) m |= AM_END_FILE;
    else if (srv == SDV_SRV_CTRL_BLOCK) m |= AM_CTRL;
    return m;
}

#ifdef SDV_EMU
#define AN_STAMP(i) ((void)0)
#else
#define AN_STAMP(i) do { if (a.timing && lane == 0) a.timing[(size_t)k * 8 + (i)] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
#endif
template <bool kLds>
__device__ inline void analyze_body(const AnalyzeArgs &a, uint32_t k, int lane, uint32_t *meta)
{
    AN_STAMP(0);
    const uint32_t start = k == 0 ? 0u : a.seg_end[k - 1] + 1u, end = a.seg_end[k];      /* [start, end) + END_FRAME at end */
    const uint32_t n = end - start;
    FrameLocal *fl = &a.fl[k];
    if (a.direct && k >= a.direct_ofs && k - a.direct_ofs < a.direct_n && a.direct[k - a.direct_ofs].flag) {
        /* A frame the frame kernel's capture took from end to end: every line a data line that read (CRC valid, forced bad or not), in field order in
         * the field buffers already.  What the three passes below find on such a frame follows from the counts: the trim is the first and the last line
         * of either field (every line passes findFramesTrim's test), every line is kept, the valid ones are those not forced bad, they all carry the one
         * reference level, no Control Block, no file mark.  Left to do: the resolution trials. */
        const sdv::DirectFrame d = a.direct[k - a.direct_ofs];
        const uint32_t n0 = d.n[0], n1 = d.n[1];
        if (lane == 0) {
            fl->frame_number = d.frame_number; fl->seg_start = start; fl->seg_len = n + 1;
            fl->top[0] = 1; fl->bottom[0] = (uint16_t)(1 + 2 * (n0 - 1)); fl->top[1] = 2; fl->bottom[1] = (uint16_t)(2 + 2 * (n1 - 1));
            fl->data_lines[0] = (uint16_t)n0; fl->data_lines[1] = (uint16_t)n1;
            fl->valid_lines[0] = (uint16_t)(n0 - d.bad[0]); fl->valid_lines[1] = (uint16_t)(n1 - d.bad[1]);
            fl->ref[0] = d.ref; fl->ref[1] = d.ref;
            fl->max_line = (uint16_t)((1 + 2 * (n0 - 1)) > (2 + 2 * (n1 - 1)) ? (1 + 2 * (n0 - 1)) : (2 + 2 * (n1 - 1)));
            fl->flags = (uint8_t)(FL_TRIM_OK | (n > BUF_TRIM / 2 ? FL_BAD_NUMBERS : 0));
            for (int i = 0; i < 5; i++) fl->ctrl[i] = -1;
        }
        __syncthreads();
        AN_STAMP(1); AN_STAMP(2); AN_STAMP(3);
        uint8_t fres[2];
        for (int p = 0; p < 2; p++) {
            Field f; f.lines = field_lines(a.fields, k, p); f.size = (int)(p == 0 ? n0 : n1);
            fres[p] = field_resolution(a.cfg, f, lane, meta, (p == 0 && a.timing) ? a.timing + (size_t)k * 8 : nullptr);
        }
        if (lane == 0) {
            fl->field_res[0] = fres[0]; fl->field_res[1] = fres[1];
            FrameBrief br; br.frame_number = d.frame_number; br.flags = (uint8_t)(FL_TRIM_OK | (n > BUF_TRIM / 2 ? FL_BAD_NUMBERS : 0)); br.field_res[0] = fres[0]; br.field_res[1] = fres[1]; br._pad = 0;
            a.brief[k] = br;
        }
        AN_STAMP(4);
        return;
    }
    const uint32_t fnum = a.src.at(end).frame_number;
    auto meta_at = [&](uint32_t i) -> uint32_t {
        if (kLds) return meta[i];
        const Rec48 *src = (const Rec48 *)&a.src.at(start + i);
        Rec48 r; r.q0 = src->q0; r.q1 = src->q1; r.q2 = src->q2;
        bool bn; return rec_meta(r, fnum, bn);
    };
    /* pass 1 (findFramesTrim, first loop :300-470): good lines per field, service flags, first Control Block */
    uint32_t good[2] = { 0, 0 }, first_good = 0xFFFFFFFFu, ctrl_pos = 0xFFFFFFFFu;
    bool new_file = false, end_file = false, bad_numbers = false;
    __syncthreads();                /* LDS of the workgroup is free again (kernels that run several frames per workgroup) */
    for (uint32_t c4 = 0; c4 < n; c4 += 256) {
        Rec48 raw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = c4 + 64u * (uint32_t)u + (uint32_t)lane;
            const Rec48 *src = (const Rec48 *)&a.src.at(start + (i < n ? i : n - 1));
            raw[u].q0 = src->q0; raw[u].q1 = src->q1; raw[u].q2 = src->q2;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t c = c4 + 64u * (uint32_t)u, i = c + (uint32_t)lane;
            if (c >= n) break;
            const bool act = i < n;
            bool bn = false;
            uint32_t m = rec_meta(raw[u], fnum, bn);
            if (!act) { m = 0; bn = false; }
            if (kLds && act) meta[i] = m;
            const bool g = (m & AM_G) != 0, odd = (m & 1u) != 0;
            uint64_t mg = __ballot(g), mo = __ballot(g && odd), mc = __ballot((m & AM_CTRL) != 0);
            new_file = new_file || __ballot((m & AM_NEW_FILE) != 0) != 0; end_file = end_file || __ballot((m & AM_END_FILE) != 0) != 0;
            bad_numbers = bad_numbers || __ballot(bn) != 0;
            good[0] += (uint32_t)__popcll(mo); good[1] += (uint32_t)__popcll(mg & ~mo);
            if (mg != 0 && first_good == 0xFFFFFFFFu) first_good = c + (uint32_t)(__ffsll((unsigned long long)mg) - 1);
            /* a Control Block counts while no good line has been seen yet: keep the last such one */
            while (mc != 0) {
                uint32_t pos = c + (uint32_t)(__ffsll((unsigned long long)mc) - 1);
                mc &= mc - 1;
                if (pos < first_good) ctrl_pos = pos;
            }
        }
    }
    __syncthreads();                /* the staged words are read by other lanes from here on */
    AN_STAMP(1);
    bool skip[2] = { good[0] > MIN_GOOD_LINES_PF, good[1] > MIN_GOOD_LINES_PF };
    /* pass 2 (second loop :480-700): top / bottom data line of each field, in stream order */
    uint16_t top[2] = { 0, 0 }, bottom[2] = { 0, 0 }, max_line = 0;
    bool have[2] = { false, false };
    for (uint32_t c = 0; c < n; c += 64) {
        uint32_t i = c + (uint32_t)lane;
        bool act = i < n, q = false, odd = false;
        uint16_t ln = 0;
        if (act) {
            const uint32_t m = meta_at(i);
            ln = (uint16_t)(m & 0xFFFF); odd = (ln % 2) != 0;
            if (m & AM_DATA) {
                bool crc_if = (m & AM_CI) != 0;
                bool markers = (m & AM_MK) != 0;
                q = skip[odd ? 0 : 1] ? crc_if : (markers || crc_if);
            }
        }
        for (int p = 0; p < 2; p++) {
            uint64_t m = __ballot(q && (odd == (p == 0)));
            if (m != 0) {
                int lo = __ffsll((unsigned long long)m) - 1, hi = 63 - __clzll((unsigned long long)m);
                uint16_t l_lo = (uint16_t)__shfl((int)ln, lo), l_hi = (uint16_t)__shfl((int)ln, hi);
                if (!have[p]) { top[p] = l_lo; have[p] = true; }
                bottom[p] = l_hi;
            }
        }
    }
    bool trim_ok = have[0] && have[1];
    AN_STAMP(2);
    /* pass 3 (splitFramesToFields :737-985): the field buffers, valid counts, mean reference level */
    uint32_t cnt[2] = { 0, 0 }, valid[2] = { 0, 0 }, ref_all[2] = { 0, 0 }, ref_ok[2] = { 0, 0 };
    /* (the records of four steps asked for at once, as in pass 1: a step's lines are stored where the steps before it have counted to,
     * one round trip to the L2 per step would be all this pass spends its time on) */
    for (uint32_t c4 = 0; c4 < n; c4 += 256) {
        SLine made[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = c4 + 64u * (uint32_t)u + (uint32_t)lane;
            const Rec48 *src = (const Rec48 *)&a.src.at(start + (i < n ? i : n - 1));
            Rec48 r; r.q0 = src->q0; r.q1 = src->q1; r.q2 = src->q2;
            made[u] = sline_from_raw(r);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t c = c4 + 64u * (uint32_t)u, i = c + (uint32_t)lane;
            if (c >= n) break;
            bool act = i < n, sel = false, odd = false, ok = false;
            uint16_t ln = 0; uint32_t ref = 0;
            if (act) {
                const uint32_t m = meta_at(i);
                if (m & (AM_DATA | AM_FILLER)) {
                    ln = (uint16_t)(m & 0xFFFF); odd = (ln % 2) != 0;
                    int p = odd ? 0 : 1;
                    bool in = ln >= top[p] && ln <= bottom[p];
                    if (!odd) in = in && ((top[1] != bottom[1]) || (top[1] != 0));
                    sel = in;
                    if (m & AM_DATA) { ok = (m & AM_G) != 0; ref = m >> 24; }
                }
            }
            /* f_max_line and the reference-level sums: per lane here, across the lanes once after the pass */
            if (ln > max_line) max_line = ln;
            for (int p = 0; p < 2; p++) {
                bool mine = sel && (odd == (p == 0));
                uint64_t m = __ballot(mine);
                uint32_t rank = cnt[p] + (uint32_t)__popcll(m & lanemask_lt(lane));
                bool kept = mine && rank < BUF_FIELD;
                if (kept) a.fields[((size_t)k * 2 + (size_t)p) * FIELD_PITCH + rank] = made[u];
                uint64_t mk = __ballot(kept), mv = __ballot(kept && ok);
                ref_all[p] += kept ? ref : 0; ref_ok[p] += (kept && ok) ? ref : 0;
                cnt[p] += (uint32_t)__popcll(mk); valid[p] += (uint32_t)__popcll(mv);
            }
        }
    }
    for (int ofs = 32; ofs > 0; ofs >>= 1) {
        uint16_t o = (uint16_t)__shfl((int)max_line, (lane + ofs) & 63); if (o > max_line) max_line = o;
        for (int p = 0; p < 2; p++) {
            ref_all[p] += (uint32_t)__shfl((int)ref_all[p], (lane + ofs) & 63);
            ref_ok[p] += (uint32_t)__shfl((int)ref_ok[p], (lane + ofs) & 63);
        }
    }
    int8_t ctrl[5] = { -1, -1, -1, -1, -1 };
    if (ctrl_pos != 0xFFFFFFFFu) {
        const sdv_line_rec &r = a.src.at(start + ctrl_pos);          /* stc007line.cpp:375-445 */
        ctrl[0] = (int8_t)((r.words[5] >> 8) & 0x3F);
        ctrl[1] = (int8_t)((r.words[5] >> 4) & 0x0F);
        ctrl[2] = (int8_t)(((r.words[6] >> 12) & 0x03) + ((r.words[5] & 0x0F) << 2));
        ctrl[3] = (int8_t)((r.words[6] >> 6) & 0x3F);
        ctrl[4] = (int8_t)(r.words[6] & 0x3F);
    }
    if (lane == 0) {
        fl->frame_number = fnum; fl->seg_start = start; fl->seg_len = n + 1;
        for (int p = 0; p < 2; p++) {
            fl->top[p] = top[p]; fl->bottom[p] = bottom[p]; fl->data_lines[p] = (uint16_t)cnt[p]; fl->valid_lines[p] = (uint16_t)valid[p];
            fl->ref[p] = valid[p] > 0 ? (uint8_t)(ref_ok[p] / valid[p]) : (cnt[p] > 0 ? (uint8_t)(ref_all[p] / cnt[p]) : 0);
        }
        fl->max_line = max_line;
        fl->flags = (uint8_t)((new_file ? FL_NEW_FILE : 0) | (end_file ? FL_END_FILE : 0) | (trim_ok ? FL_TRIM_OK : 0) | ((bad_numbers || n > BUF_TRIM / 2) ? FL_BAD_NUMBERS : 0));
        for (int i = 0; i < 5; i++) fl->ctrl[i] = ctrl[i];
    }
    __syncthreads();            /* the field buffers are read back through processBlock below */
    AN_STAMP(3);
    uint8_t fres[2];
    for (int p = 0; p < 2; p++) {
        Field f; f.lines = field_lines(a.fields, k, p); f.size = (int)cnt[p];
        fres[p] = field_resolution(a.cfg, f, lane, meta, (p == 0 && a.timing) ? a.timing + (size_t)k * 8 : nullptr);
    }
    if (lane == 0) {
        fl->field_res[0] = fres[0]; fl->field_res[1] = fres[1];
        FrameBrief br; br.frame_number = fnum; br.flags = fl->flags; br.field_res[0] = fres[0]; br.field_res[1] = fres[1]; br._pad = 0;
        a.brief[k] = br;
    }
    AN_STAMP(4);
}
} // namespace sdvs

namespace sdvs {
/* ================================================================================================================== */
/* sdv_k_stitch_step: one wave per stitcher turn                                                                       */
/* ================================================================================================================== */
/* what one turn hands to the next (STC007DataStitcher members that survive a turn of doFrameReassemble) */
struct StepChain {
    Frasm f0;                               /* frasm_f0 */
    uint8_t last_pad_counter, broken_countdown;
    uint16_t tail_n;                        /* lines left in conv_queue (<= 112) */
    uint32_t _pad[3];
    SLine tail[MIN_DEINT];
};
enum { SI_OVERFLOW = 1, SI_STEADY = 2, SI_MISS = 4 /* direct output: the turn wanted more room than was guessed */ };
struct StepInfo { uint32_t n_pairs; uint8_t n_frasm, changed, push_order, bits; };
struct StepArgs {
    const SLine *fields; const FrameLocal *fl; Cfg cfg;
    uint32_t *next_work;                    /* work queue head: the resident waves take turns as they finish (turn times differ) */
    const uint32_t *work; uint32_t n_work;  /* steps to run: k | which[k-1] << 30 | which[k] << 31 (buffer holding the current output) */
    const StepChain *chain0;                /* the stream's state before step 0 */
    StepChain *chain[2];
    const uint8_t *prob_order, *prob_res;   /* getProbableFieldOrder() before the step's own push / getProbableResolution() after its pushes */
    SLine *ws;                              /* QCAP lines per wave */
    sdv_sample_pair *pairs; sdv_frame_asm *frasm; StepInfo *info;
    /* direct output: when every turn of the stream is known to emit guess_pairs pairs and guess_frasm descriptors (a tape that plays),
     * the turns write straight into the caller's buffers at k * guess; NULL = per-turn slots, packed by the compact kernel */
    sdv_sample_pair *direct_pairs; sdv_frame_asm *direct_frasm; uint32_t guess_pairs, guess_frasm;
    uint32_t first_round;
    const uint32_t *ctl; uint32_t est_seg;  /* pipelined call: the turns are 0 .. n_seg - 2 of the counted segments, `work` is NULL (every turn, first buffers), */
    uint8_t pipe_order, pipe_res;           /* ... and prob_order / prob_res are these two for every turn */
    unsigned long long *timing;             /* optional: 8 cycle stamps per step (SDV_STITCH_TIMING=1), NULL otherwise */
    sdv_block_rec *blocks; const uint32_t *block_ofs;   /* optional (the visualiser's feed): turn k's data blocks go to blocks[block_ofs[k] ..], newBlockProcessed :6626 */
    uint32_t *asm_cnt; sdv_asm_line_rec *asm_lines; const uint32_t *asm_ofs;   /* optional: how many assembled lines turn k hands to the visualiser (newLineProcessed :6696) / where they go */
};

struct FieldStitchStats { uint16_t index, valid, silent, unchecked, broken; };    /* frametrimset.h:278-300 */
__device__ inline void stats_clear(FieldStitchStats &t) { t.index = t.valid = 0; t.silent = t.unchecked = t.broken = 0xFF; }
__device__ inline bool stats_lt(const FieldStitchStats &a, const FieldStitchStats &b)   /* frametrimset.cpp:312-371 */
{
    if (a.broken != b.broken) return a.broken < b.broken;
    if (a.valid != b.valid) return a.valid > b.valid;
    if (a.unchecked != b.unchecked) return a.unchecked < b.unchecked;
    if (a.silent != b.silent) return a.silent < b.silent;
    return a.index < b.index;
}
/* the two smallest entries, which is all findPadding reads of the sorted array */
__device__ inline void stats_best2(const FieldStitchStats *sd, int n, FieldStitchStats &s0, FieldStitchStats &s1)
{
    int i0 = 0;
    for (int i = 1; i < n; i++) if (stats_lt(sd[i], sd[i0])) i0 = i;
    int i1 = i0 == 0 ? 1 : 0;
    for (int i = 0; i < n; i++) if (i != i0 && i != i1 && stats_lt(sd[i], sd[i1])) i1 = i;
    s0 = sd[i0]; s1 = sd[i1];
}

enum { STG_TRY_PREVIOUS = 0, STG_TRY_TFF_TO_TFF, STG_TRY_BFF_TO_BFF, STG_A_PREPARE, STG_A_PAD_TFF, STG_A_PAD_BFF, STG_AB_UNK_PREPARE,
       STG_AB_TFF_TO_TFF, STG_AB_TFF_TO_BFF, STG_AB_BFF_TO_BFF, STG_AB_BFF_TO_TFF, STG_PAD_NO_GOOD, STG_PAD_SILENCE, STG_PAD_OK, STG_PAD_MAX };

struct Step {
    Cfg cfg; const SLine *fields; uint32_t k; FrameLocal l1, l2;      /* frame A, frame B */
    Frasm f0, f1, f2;
    uint8_t last_pad_counter, broken_countdown, prob_order, prob_res, push_order;
    bool file_start, file_end;
    SLine *q; int qn;                       /* conv_queue */
    sdv_deint_line *ring;                   /* LDS, RING entries (RingSrc) */
    int lane;

    __device__ inline Field field(int frame, int parity) const
    {
        Field f; f.lines = field_lines(fields, frame == 1 ? k : k + 1, parity);
        const Frasm &fr = frame == 1 ? f1 : f2;
        f.size = parity == 0 ? fr.odd_data_lines : fr.even_data_lines;
        return f;
    }
    /* resolution a line was detected with (getDataBlockResolution's per-line lookup, :1290-1400) */
    __device__ inline uint8_t line_res(uint32_t frame, uint16_t line) const
    {
        /* (both fields read, then blended: a conditional read would keep the whole turn state in memory) */
        const uint32_t m = (line % 2) == 0 ? 0xFFu : 0u;
        if (frame == f2.frame_number) return (uint8_t)((f2.even_resolution & m) | (f2.odd_resolution & ~m));
        if (frame == f1.frame_number) return (uint8_t)((f1.even_resolution & m) | (f1.odd_resolution & ~m));
        if (frame == f0.frame_number) return (uint8_t)((f0.even_resolution & m) | (f0.odd_resolution & ~m));
        return SDV_RES_MODE_14BIT;
    }
    __device__ inline uint8_t block_res_mode(const SLine &first, const SLine &last) const
    {
        return res_mode_for_seam(line_res(first.frame, first.line), line_res(last.frame, last.line));
    }

    /* ---- tryPadding (:1417-1740) ---- */
    __device__ inline uint8_t try_padding(const Field &fa, const Field &fb, uint16_t padding, FieldStitchStats *st)
    {
        if (fa.size > BUF_FIELD || fb.size > BUF_FIELD) return DS_NO_DATA;
        PadQueue pq; pq.f1 = fa; pq.f2 = fb; pq.m2 = cfg.m2 != 0;
        int keep = MIN_DEINT + ILV / 2 - (int)padding;
        pq.a0 = fa.size > keep ? fa.size - keep : 0;
        pq.n0 = fa.size - pq.a0;
        pq.npad = padding;
        pq.pad_frame = 0; pq.pad_line0 = 2;
        pq.n2 = fb.size > (MIN_DEINT + ILV / 2) ? (MIN_DEINT + ILV / 2) : fb.size;
        const int n = pq.size();
        if (n < MIN_DEINT) return DS_NO_DATA;
        const uint8_t unchecked_lim = cfg.en_q ? cfg.max_unch_14 : cfg.max_unch_16;
#if SDV_ST_TRY_BATCH
        /* one round trip for all that is read before the first block: the field lines among the first 176 places of the queue and the
         * line the padding counts on from (places 0 and 112, which say what resolution the trial runs in, are among the former) */
        SLine got[3], lastl;
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int i = lane + 64 * u;
            got[u].frame = 0; got[u].line = 0;
            if (i < RING_SPAN && i < n) { const SLine *sp = pq.src(i); if (sp) got[u] = *sp; }
        }
        if (fa.size > 0) { lastl = fa.get(fa.size - 1); pq.pad_frame = uni(lastl.frame); pq.pad_line0 = (uint16_t)uni((uint32_t)(lastl.line + 2)); }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int i = lane + 64 * u;
            if (i < RING_SPAN && i < n && !pq.src(i)) got[u] = pq.get(i);
        }
        uint8_t mode = SDV_RES_MODE_14BIT;
        if (!cfg.m2) {
            if (n <= MIN_DEINT) mode = (uint8_t)SDV_RES_MODE_14BIT_AUTO;
            else {
                static_assert(MIN_DEINT == 112, "place 112 = lane 48 of the second step");
                const uint32_t fr0 = (uint32_t)__shfl((int)got[0].frame, 0), ln0 = (uint32_t)__shfl((int)got[0].line, 0);
                const uint32_t fr1 = (uint32_t)__shfl((int)got[1].frame, MIN_DEINT - 64), ln1 = (uint32_t)__shfl((int)got[1].line, MIN_DEINT - 64);
                mode = (uint8_t)uni(res_mode_for_seam(line_res(fr0, (uint16_t)ln0), line_res(fr1, (uint16_t)ln1)));
            }
        }
#else
        if (fa.size > 0) { SLine l = fa.get(fa.size - 1); pq.pad_frame = uni(l.frame); pq.pad_line0 = (uint16_t)uni((uint32_t)(l.line + 2)); }
        uint8_t mode = SDV_RES_MODE_14BIT;
        if (!cfg.m2) mode = n <= MIN_DEINT ? (uint8_t)SDV_RES_MODE_14BIT_AUTO : (uint8_t)uni(block_res_mode(pq.get(0), pq.get(MIN_DEINT)));
#endif
        const sdv_deint_settings ds = deint_cfg(mode, cfg.ignore_crc, true, cfg.en_p, cfg.en_q, false);
        const int nblk = n - MIN_DEINT;
        uint16_t valid_cnt = 0, silence_cnt = 0, uncheck_cnt = 0, broken_count = 0, valid_max = 0, silence_max = 0, uncheck_max = 0;
        RingSrc rs; rs.ring = ring;
        SDV_LDS_WAVE_SYNC();                                          /* whoever used the ring before is done with it */
#if SDV_ST_TRY_BATCH
#pragma unroll
        for (int u = 0; u < 3; u++) { const int i = lane + 64 * u; if (i < RING_SPAN && i < n) ring[i] = view(got[u]); }
#else
        for (int i = lane; i < RING_SPAN && i < n; i += 64) ring[i] = pq.line((size_t)i);
#endif
        for (int c = 0; c * 64 < nblk; c++) {
            int i = c * 64 + lane;
            bool v = false, sl = false, u = false, br = false;
            const int nx = c * 64 + RING_SPAN + lane;                 /* the next step's new lines: asked for now, stored behind this step's decodes */
            const bool has_nx = nx < n && (c + 1) * 64 < nblk;
            sdv_deint_line nxl; if (has_nx) nxl = pq.line((size_t)nx);
            SDV_LDS_WAVE_SYNC();
            if (i < nblk) {
                Block b; sdvd::Lines8 l8;
                sdvd::gather8(rs, (size_t)i, l8);
                sdvd::process_block(ds, l8, 0, b);
                bool silent = blk_silent(b, cfg.m2), force = can_force_check(b);
                v = blk_valid(b) && !silent && force;
                sl = silent;
                u = cfg.en_q ? (!force || b.audio_state == SDV_AUD_FIX_Q) : (b.audio_state == SDV_AUD_FIX_P);
                br = b.audio_state == SDV_AUD_BROKEN;
            }
            uint64_t mv = __ballot(v), ms = __ballot(sl), mu = __ballot(u), mb = __ballot(br);
            int cnt = nblk - c * 64; if (cnt > 64) cnt = 64;
            const uint64_t full = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            if ((ms | mu | mb) == 0 && (mv & full) == full) {        /* a chunk of valid, checked, sounding blocks: the run goes on */
                valid_cnt = (uint16_t)(valid_cnt + cnt);
                if (silence_cnt > silence_max) silence_max = silence_cnt;
                silence_cnt = 0;
                if (uncheck_cnt > uncheck_max) uncheck_max = uncheck_cnt;
                uncheck_cnt = 0;
            } else
            for (int j = 0; j < cnt; j++) {
                if ((mv >> j) & 1) valid_cnt++; else if (valid_cnt > valid_max) valid_max = valid_cnt;
                if ((ms >> j) & 1) { silence_cnt++; if (silence_cnt >= MAX_BURST_SILENCE) valid_cnt = 0; }
                else { if (silence_cnt > silence_max) silence_max = silence_cnt; silence_cnt = 0; }
                if ((mu >> j) & 1) { uncheck_cnt++; if (uncheck_cnt >= unchecked_lim) valid_cnt = 0; }
                else { if (uncheck_cnt > uncheck_max) uncheck_max = uncheck_cnt; uncheck_cnt = 0; }
                if ((mb >> j) & 1) { broken_count++; if (broken_count >= MAX_BURST_BROKEN) valid_cnt = 0; }
            }
            if (has_nx) ring[nx & (RING - 1)] = nxl;
        }
        if (valid_cnt > valid_max) valid_max = valid_cnt;
        if (silence_cnt > silence_max) silence_max = silence_cnt;
        if (uncheck_cnt > uncheck_max) uncheck_max = uncheck_cnt;
        if (st) { st->index = padding; st->valid = valid_max; st->silent = silence_max; st->unchecked = uncheck_max; st->broken = broken_count; }
        if (broken_count >= MAX_BURST_BROKEN) return DS_BROKE;
        if (silence_max > MAX_BURST_SILENCE) return DS_SILENCE;
        if (uncheck_max > unchecked_lim) return DS_NO_PAD;
        if (valid_max == 0) return DS_NO_PAD;
        return DS_OK;
    }

    /* ---- findPadding (:1743-2054) and the single tryPadding of the "same as last frame" path, behind one call site ----
     * find = false: tryPadding(fa, fb, pad_in) -> its return code.
     * find = true : findPadding(fa, fb, in_std, in_resolution, &padding); *found receives what the reference leaves in *padding. */
    __device__ inline uint8_t pad_search(bool find, const Field &fa, const Field &fb, uint16_t pad_in, uint8_t in_std, uint8_t in_resolution, uint16_t *found)
    {
        uint8_t stitch_res = DS_NO_PAD;
        int max_padding = 1; uint8_t unchecked_lim = cfg.max_unch_14;
        if (find) {
            const uint16_t pad = (uint16_t)fa.size;
            if (in_std == VID_PAL) *found = pad > LINES_PF_PAL ? 0 : (uint16_t)(LINES_PF_PAL - pad);
            else if (in_std == VID_NTSC) *found = pad > LINES_PF_NTSC ? 0 : (uint16_t)(LINES_PF_NTSC - pad);
            else *found = 0;
            max_padding = MAX_PADDING_14BIT;
            if (in_resolution == SDV_RES_16BIT || !cfg.en_q) { max_padding = MAX_PADDING_16BIT; unchecked_lim = cfg.max_unch_16; }
            last_pad_counter = 0xFF;
            if (!(cfg.en_p || cfg.en_q)) return stitch_res;
        }
        /* findPadding only reads the two best entries of its sorted statistics; paddings never tried (early break) keep
         * FieldStitchStats' cleared value and take part in the ranking like the reference's untouched slots */
        FieldStitchStats cur, s0, s1, no_brk;
        stats_clear(no_brk);
        uint16_t min_broken = 0xFFFF;
        int p_end = 0;
        for (int p = 0; p < max_padding; p++) {
            const uint8_t code = try_padding(fa, fb, find ? (uint16_t)p : pad_in, &cur);
            if (!find) return code;
            p_end = p + 1;
            sd_store(p, cur);
            if (min_broken > cur.broken) { min_broken = cur.broken; if (min_broken == 0) no_brk = cur; }
            else if (min_broken == 0) {
                if (no_brk.valid > 0 && no_brk.unchecked < unchecked_lim && cur.broken > 0) break;
            }
        }
        /* ranking: tried entries from the table, the rest are cleared entries */
        sd_best2(p_end, max_padding, false, 0, unchecked_lim, s0, s1);
        last_pad_counter = (uint8_t)s0.broken;
        if (s0.silent < MAX_BURST_SILENCE) {
            if (s0.unchecked < unchecked_lim) {
                if (s0.broken < 2 && s0.broken < s1.broken) { stitch_res = DS_OK; *found = s0.index; }
                else if ((((int16_t)s0.valid - (int16_t)s1.valid) > MAX_BURST_UNCH_DELTA) && s0.broken == 0) { stitch_res = DS_OK; *found = s0.index; }
            } else {
                sd_best2(p_end, max_padding, true, min_broken, unchecked_lim, s0, s1);
                if (s0.unchecked < unchecked_lim)
                    if (((int16_t)s0.valid - (int16_t)s1.valid) > MAX_BURST_UNCH_DELTA) { stitch_res = DS_OK; *found = s0.index; }
            }
        } else stitch_res = DS_SILENCE;
        return stitch_res;
    }
    /* findPadding's statistics table lives in LDS (one wave per workgroup): lane-uniform values, written by lane 0 */
    __device__ static inline FieldStitchStats *sd_tab() { __shared__ FieldStitchStats s_sd[MAX_PADDING_14BIT]; return s_sd; }
    __device__ inline void sd_store(int p, const FieldStitchStats &v)
    {
        if (lane == 0) sd_tab()[p] = v;
        __syncthreads();
    }
    /* the two smallest entries of the table under FieldStitchStats::operator< ; remap = the second ranking of findPadding
     * (:1990-2030): broken := min_broken, or 0xFF where unchecked >= limit - applied to untouched entries too */
    __device__ inline void sd_best2(int n_tried, int n_total, bool remap, uint16_t min_broken, uint8_t unchecked_lim, FieldStitchStats &s0, FieldStitchStats &s1)
    {
        FieldStitchStats b0, b1; bool h0 = false, h1 = false;
        for (int i = 0; i < n_total; i++) {
            FieldStitchStats e;
            if (i < n_tried) e = sd_tab()[i]; else stats_clear(e);
            if (remap) { e.broken = min_broken; if (e.unchecked >= unchecked_lim) e.broken = 0xFF; }
            if (!h0) { b0 = e; h0 = true; }
            else if (stats_lt(e, b0)) { b1 = b0; h1 = true; b0 = e; }
            else if (!h1 || stats_lt(e, b1)) { b1 = e; h1 = true; }
        }
        if (!h1) stats_clear(b1);
        s0 = b0; s1 = b1;
    }

    /* ---- detectAudioResolution (:2207-2763); the statistics pushes are replayed by the engine ---- */
    __device__ static inline void set_pair(uint8_t known_res, uint8_t &known, uint8_t &other)
    {
        if (known_res == SRES_16BIT) { known = SDV_RES_MODE_16BIT; other = SDV_RES_MODE_16BIT_AUTO; }
        else { known = SDV_RES_MODE_14BIT; other = SDV_RES_MODE_14BIT_AUTO; }
    }
    __device__ inline void detect_audio_resolution()
    {
        if (cfg.m2) { f1.odd_resolution = f1.even_resolution = f2.odd_resolution = f2.even_resolution = SDV_RES_MODE_14BIT; return; }
        const uint8_t f1o = l1.field_res[0], f1e = l1.field_res[1], f2o = l2.field_res[0], f2e = l2.field_res[1];
        const uint8_t prob_mode = prob_res == SRES_16BIT ? SDV_RES_MODE_16BIT_AUTO : SDV_RES_MODE_14BIT_AUTO;
        if (f1o == SRES_UNKNOWN && f1e == SRES_UNKNOWN) {
            if (f2o == SRES_UNKNOWN && f2e == SRES_UNKNOWN) f1.odd_resolution = f1.even_resolution = f2.odd_resolution = f2.even_resolution = prob_mode;
            else if (f2o == SRES_UNKNOWN) {
                if (f2e == SRES_16BIT) { f2.even_resolution = SDV_RES_MODE_16BIT; f1.odd_resolution = f1.even_resolution = f2.odd_resolution = SDV_RES_MODE_16BIT_AUTO; }
                else { f2.even_resolution = SDV_RES_MODE_14BIT; f1.odd_resolution = f1.even_resolution = f2.odd_resolution = SDV_RES_MODE_14BIT_AUTO; }
            } else if (f2e == SRES_UNKNOWN) {
                if (f2o == SRES_16BIT) { f2.odd_resolution = SDV_RES_MODE_16BIT; f1.odd_resolution = f1.even_resolution = f2.even_resolution = SDV_RES_MODE_16BIT_AUTO; }
                else { f2.odd_resolution = SDV_RES_MODE_14BIT; f1.odd_resolution = f1.even_resolution = f2.even_resolution = SDV_RES_MODE_14BIT_AUTO; }
            } else if (f2o == f2e && f2o == SRES_16BIT) { f2.odd_resolution = f2.even_resolution = SDV_RES_MODE_16BIT; f1.odd_resolution = f1.even_resolution = SDV_RES_MODE_16BIT_AUTO; }
            else {
                f2.odd_resolution = f2o == SRES_16BIT ? SDV_RES_MODE_16BIT : SDV_RES_MODE_14BIT;
                f2.even_resolution = f2e == SRES_16BIT ? SDV_RES_MODE_16BIT : SDV_RES_MODE_14BIT;
                f1.odd_resolution = f1.even_resolution = SDV_RES_MODE_14BIT_AUTO;
            }
        } else {
            if (f1o == SRES_UNKNOWN) set_pair(f1e, f1.even_resolution, f1.odd_resolution);
            else if (f1e == SRES_UNKNOWN) set_pair(f1o, f1.odd_resolution, f1.even_resolution);
            else {
                f1.odd_resolution = f1o == SRES_16BIT ? SDV_RES_MODE_16BIT : SDV_RES_MODE_14BIT;
                f1.even_resolution = f1e == SRES_16BIT ? SDV_RES_MODE_16BIT : SDV_RES_MODE_14BIT;
            }
            if (f2o == SRES_UNKNOWN && f2e == SRES_UNKNOWN) f2.odd_resolution = f2.even_resolution = prob_mode;
            else if (f2o == SRES_UNKNOWN) set_pair(f2e, f2.even_resolution, f2.odd_resolution);
            else if (f2e == SRES_UNKNOWN) set_pair(f2o, f2.odd_resolution, f2.even_resolution);
            else {
                f2.odd_resolution = f2o == SRES_16BIT ? SDV_RES_MODE_16BIT : SDV_RES_MODE_14BIT;
                f2.even_resolution = f2e == SRES_16BIT ? SDV_RES_MODE_16BIT : SDV_RES_MODE_14BIT;
            }
        }
    }

    /* ---- detectVideoStandard (:2773-2925) ---- */
    __device__ inline void detect_video_standard()
    {
        f1.video_standard = VID_UNKNOWN; f1.odd_std_lines = f1.even_std_lines = 0;
        if (cfg.preset_video_mode == VID_UNKNOWN) {
            f1.vid_std_preset = 0;
            uint16_t a = f1.odd_data_lines, b = f1.even_data_lines, c = f2.odd_data_lines, d = f2.even_data_lines;
            if (a > LINES_PF_MAX_PAL || b > LINES_PF_MAX_PAL || c > LINES_PF_MAX_PAL || d > LINES_PF_MAX_PAL) f1.video_standard = VID_UNKNOWN;
            else if (a > LINES_PF_MAX_NTSC || b > LINES_PF_MAX_NTSC || c > LINES_PF_MAX_NTSC || d > LINES_PF_MAX_NTSC) f1.video_standard = VID_PAL;
            else f1.video_standard = l1.max_line <= ((LINES_PF_PAL - ILV) * 2) ? VID_NTSC : VID_PAL;
        } else { f1.vid_std_preset = 1; f1.video_standard = cfg.preset_video_mode; }
        if (f1.video_standard == VID_UNKNOWN) f1.video_standard = f0.video_standard;
        if (f1.video_standard == VID_NTSC) f1.odd_std_lines = f1.even_std_lines = LINES_PF_NTSC;
        else if (f1.video_standard == VID_PAL) f1.odd_std_lines = f1.even_std_lines = LINES_PF_PAL;
        if (cfg.preset_field_order == ORDER_TFF) { preset_order(f1, ORDER_TFF); preset_order(f2, ORDER_TFF); }
        else if (cfg.preset_field_order == ORDER_BFF) { preset_order(f1, ORDER_BFF); preset_order(f2, ORDER_BFF); }
        else { f2.order_preset = 0; set_order_unknown(f2); }
    }

    /* ---- findFieldStitching (:2929-4275) ----
     * Same stages as the reference; every stage first says which padding search it needs (if any), the search runs at one
     * call site, then the stage consumes its result - so the padding search exists once in the kernel's code. */
    __device__ inline void find_field_stitching()
    {
        bool en_sw_order = true;
        uint8_t proc_state = STG_TRY_PREVIOUS, stage_count = 0, f_res;
        detect_audio_resolution();
        detect_video_standard();
        const Field f1o = field(1, 0), f1e = field(1, 1), f2o = field(2, 0), f2e = field(2, 1);
        for (;;) {
            stage_count++;
            /* -- 1. the search this stage asks for: kind 0 none, 1 tryPadding(fa, fb, pad_in), 2 findPadding(fa, fb) */
            int kind = 0, a_sel = 0, b_sel = 0;          /* field selectors: 0 f1o, 1 f1e, 2 f2o, 3 f2e */
            uint16_t pad_in = 0; uint8_t sres = SDV_RES_14BIT;
            const bool f1_small = f1.odd_data_lines < MIN_FILL_LINES_PF && f1.even_data_lines < MIN_FILL_LINES_PF;
            const bool f2_small = f2.odd_data_lines < MIN_FILL_LINES_PF && f2.even_data_lines < MIN_FILL_LINES_PF;
            if (proc_state == STG_TRY_PREVIOUS) {
                if (f0.odd_data_lines == f1.odd_data_lines && f0.even_data_lines == f1.even_data_lines && f0.inner_padding_ok && f0.outer_padding_ok &&
                    (!f1.order_preset || f0.field_order == f1.field_order) && !f1_small && order_set(f0)) {
                    kind = 1; pad_in = f0.inner_padding;
                    if (f0.field_order == ORDER_TFF) { a_sel = 0; b_sel = 1; } else { a_sel = 1; b_sel = 0; }
                }
            } else if (proc_state == STG_TRY_TFF_TO_TFF) {
                if (f2.odd_data_lines >= MIN_FILL_LINES_PF) { kind = 1; a_sel = 1; b_sel = 2; pad_in = f0.outer_padding; }
            } else if (proc_state == STG_TRY_BFF_TO_BFF) {
                if (f2.even_data_lines >= MIN_FILL_LINES_PF) { kind = 1; a_sel = 0; b_sel = 3; pad_in = f0.outer_padding; }
            } else if (proc_state == STG_A_PAD_TFF) { kind = 2; a_sel = 0; b_sel = 1; sres = res_for_seam(f1.odd_resolution, f1.even_resolution); }
            else if (proc_state == STG_A_PAD_BFF) { kind = 2; a_sel = 1; b_sel = 0; sres = res_for_seam(f1.even_resolution, f1.odd_resolution); }
            else if (proc_state == STG_AB_TFF_TO_TFF) {
                if (!f2_small && f2.odd_data_lines >= MIN_FILL_LINES_PF) { kind = 2; a_sel = 1; b_sel = 2; sres = res_for_seam(f1.even_resolution, f2.odd_resolution); }
            } else if (proc_state == STG_AB_BFF_TO_BFF) {
                if (!f2_small && f2.even_data_lines >= MIN_FILL_LINES_PF) { kind = 2; a_sel = 0; b_sel = 3; sres = res_for_seam(f1.odd_resolution, f2.even_resolution); }
            } else if (proc_state == STG_AB_TFF_TO_BFF) { kind = 2; a_sel = 1; b_sel = 3; sres = res_for_seam(f1.even_resolution, f2.even_resolution); }
            else if (proc_state == STG_AB_BFF_TO_TFF) { kind = 2; a_sel = 0; b_sel = 2; sres = res_for_seam(f1.odd_resolution, f2.odd_resolution); }
            uint16_t found = 0;
            f_res = DS_NO_PAD;
            if (kind != 0) {
                Field fa, fb;
                fa.lines = a_sel == 0 ? f1o.lines : f1e.lines; fa.size = a_sel == 0 ? f1o.size : f1e.size;
                fb.lines = b_sel == 0 ? f1o.lines : (b_sel == 1 ? f1e.lines : (b_sel == 2 ? f2o.lines : f2e.lines));
                fb.size = b_sel == 0 ? f1o.size : (b_sel == 1 ? f1e.size : (b_sel == 2 ? f2o.size : f2e.size));
                f_res = pad_search(kind == 2, fa, fb, pad_in, f1.video_standard, sres, &found);
            }
            /* -- 2. the stage */
            if (proc_state == STG_TRY_PREVIOUS) {
                proc_state = STG_A_PREPARE;
                if (f0.odd_data_lines == f1.odd_data_lines && f0.even_data_lines == f1.even_data_lines && f0.inner_padding_ok && f0.outer_padding_ok) {
                    if (!f1.order_preset || f0.field_order == f1.field_order) {
                        f1.inner_silence = f1.outer_silence = f2.inner_silence = f2.outer_silence = 1;
                        f2.inner_padding_ok = f2.outer_padding_ok = 0; f2.inner_padding = f2.outer_padding = 0;
                        if (f1_small) {
                            set_order_unknown(f1);
                            f1.inner_padding_ok = f1.outer_padding_ok = 0; f1.inner_padding = f1.outer_padding = 0;
                            proc_state = STG_PAD_NO_GOOD;
                        } else if (f_res == DS_OK) {
                            if (!f1.vid_std_preset && f0.video_standard < VID_MAX) f1.video_standard = f0.video_standard;
                            f1.field_order = f0.field_order;
                            f1.inner_padding = f0.inner_padding; f1.inner_padding_ok = 1; f1.inner_silence = 0;
                            if (f1.field_order == ORDER_TFF) { f1.tff_cnt = last_pad_counter; proc_state = STG_TRY_TFF_TO_TFF; }
                            else { f1.bff_cnt = last_pad_counter; proc_state = STG_TRY_BFF_TO_BFF; }
                        }
                    }
                }
            } else if (proc_state == STG_TRY_TFF_TO_TFF || proc_state == STG_TRY_BFF_TO_BFF) {
                const bool tt = proc_state == STG_TRY_TFF_TO_TFF;
                if (f_res == DS_OK) { f1.outer_padding = f0.outer_padding; f1.outer_padding_ok = 1; set_order(f2, tt ? ORDER_TFF : ORDER_BFF); f1.outer_silence = 0; proc_state = STG_PAD_OK; }
                else { proc_state = tt ? STG_AB_TFF_TO_TFF : STG_AB_BFF_TO_BFF; en_sw_order = false; }
            } else if (proc_state == STG_A_PREPARE) {
                f1.inner_padding_ok = f1.outer_padding_ok = 0; f1.inner_padding = f1.outer_padding = 0; f1.tff_cnt = f1.bff_cnt = 0;
                if (f1_small) {
                    if (!f1.order_preset) set_order_unknown(f1);
                    proc_state = STG_PAD_NO_GOOD;
                } else if (f1.even_data_lines < MIN_FILL_LINES_PF) {
                    if (f1.field_order == ORDER_TFF) { f1.outer_padding_ok = 0; f1.outer_padding = 0; proc_state = STG_PAD_NO_GOOD; }
                    else { proc_state = STG_AB_BFF_TO_BFF; en_sw_order = false; }
                } else if (f1.odd_data_lines < MIN_FILL_LINES_PF) {
                    if (f1.field_order == ORDER_BFF) { f1.outer_padding_ok = 0; f1.outer_padding = 0; proc_state = STG_PAD_NO_GOOD; }
                    else { proc_state = STG_AB_TFF_TO_TFF; en_sw_order = false; }
                } else {
                    if (f1.field_order == ORDER_BFF) { proc_state = STG_A_PAD_BFF; en_sw_order = false; }
                    else if (f1.field_order == ORDER_TFF) { proc_state = STG_A_PAD_TFF; en_sw_order = false; }
                    else { proc_state = prob_order == ORDER_BFF ? STG_A_PAD_BFF : STG_A_PAD_TFF; en_sw_order = true; }
                }
            } else if (proc_state == STG_A_PAD_TFF || proc_state == STG_A_PAD_BFF) {
                const bool tff = proc_state == STG_A_PAD_TFF;
                f1.inner_padding = found;
                if (tff) f1.tff_cnt = last_pad_counter; else f1.bff_cnt = last_pad_counter;
                f1.inner_silence = 0;
                if (f_res == DS_OK) {
                    set_order(f1, tff ? ORDER_TFF : ORDER_BFF);
                    f1.inner_padding_ok = 1;
                    proc_state = tff ? STG_AB_TFF_TO_TFF : STG_AB_BFF_TO_BFF; en_sw_order = false;
                } else if (f_res == DS_SILENCE) {
                    f1.inner_silence = 1; f1.outer_silence = 1; f1.inner_padding_ok = 0; f1.inner_padding = 0;
                    proc_state = STG_PAD_SILENCE;
                } else {
                    f1.inner_padding = 0;
                    if ((tff && f1.field_order == ORDER_TFF) || (!tff && f1.field_order == ORDER_BFF)) {
                        f1.inner_padding_ok = 0;
                        proc_state = tff ? STG_AB_TFF_TO_TFF : STG_AB_BFF_TO_BFF; en_sw_order = false;
                    } else if (en_sw_order) { proc_state = tff ? STG_A_PAD_BFF : STG_A_PAD_TFF; en_sw_order = false; }
                    else proc_state = STG_AB_UNK_PREPARE;
                }
            } else if (proc_state == STG_AB_UNK_PREPARE) {
                f1.inner_padding = 0; f1.inner_padding_ok = 0; set_order_unknown(f1);
                proc_state = prob_order == ORDER_BFF ? STG_AB_BFF_TO_BFF : STG_AB_TFF_TO_TFF;
                en_sw_order = true;
            } else if (proc_state == STG_AB_TFF_TO_TFF || proc_state == STG_AB_BFF_TO_BFF) {
                const bool tt = proc_state == STG_AB_TFF_TO_TFF;
                const uint16_t need = tt ? f2.odd_data_lines : f2.even_data_lines, other = tt ? f2.even_data_lines : f2.odd_data_lines;
                if (f2_small) {
                    f1.outer_padding = 0; f1.outer_padding_ok = 0; f2.inner_padding_ok = 0; proc_state = STG_PAD_NO_GOOD;
                } else if (need < MIN_FILL_LINES_PF) {
                    if (!f1.order_preset) proc_state = tt ? STG_AB_TFF_TO_BFF : STG_AB_BFF_TO_TFF;
                    else { f1.outer_padding = 0; f1.outer_padding_ok = 0; f2.inner_padding_ok = 0; proc_state = STG_PAD_NO_GOOD; }
                } else {
                    f1.outer_padding = found;
                    f1.outer_silence = 0;
                    if (f_res == DS_OK) {
                        f1.outer_padding_ok = 1;
                        set_order(f2, tt ? ORDER_TFF : ORDER_BFF);
                        proc_state = STG_PAD_OK;
                        if (!order_set(f1)) set_order(f1, tt ? ORDER_TFF : ORDER_BFF);
                        else if ((tt && f1.field_order == ORDER_BFF) || (!tt && f1.field_order == ORDER_TFF)) { f1.outer_padding_ok = 0; proc_state = STG_PAD_NO_GOOD; }
                    } else if (f_res == DS_SILENCE) {
                        f1.outer_silence = 1; f1.outer_padding = 0; f1.outer_padding_ok = 0; proc_state = STG_PAD_SILENCE;
                    } else {
                        if (other < MIN_FILL_LINES_PF) { f1.outer_padding = 0; f1.outer_padding_ok = 0; f2.inner_padding_ok = 0; proc_state = STG_PAD_NO_GOOD; }
                        else if (!f1.order_preset) proc_state = tt ? STG_AB_TFF_TO_BFF : STG_AB_BFF_TO_TFF;
                        else { f1.outer_padding = 0; f1.outer_padding_ok = 0; proc_state = STG_PAD_NO_GOOD; }
                    }
                }
            } else if (proc_state == STG_AB_TFF_TO_BFF || proc_state == STG_AB_BFF_TO_TFF) {
                const bool tb = proc_state == STG_AB_TFF_TO_BFF;
                f1.outer_padding = found;
                f1.outer_silence = 0;
                if (f_res == DS_OK) {
                    f1.outer_padding_ok = 1;
                    set_order(f2, tb ? ORDER_BFF : ORDER_TFF);
                    proc_state = STG_PAD_OK;
                    if (!order_set(f1)) set_order(f1, tb ? ORDER_TFF : ORDER_BFF);
                    else if ((tb && f1.field_order == ORDER_BFF) || (!tb && f1.field_order == ORDER_TFF)) { f1.outer_padding_ok = 0; proc_state = STG_PAD_NO_GOOD; }
                } else if (f_res == DS_SILENCE) {
                    f1.outer_silence = 1; f1.outer_padding = 0; f1.outer_padding_ok = 0; f2.inner_padding_ok = 0; proc_state = STG_PAD_SILENCE;
                } else {
                    f1.outer_padding = 0; f1.outer_padding_ok = 0; f2.inner_padding_ok = 0;
                    if (en_sw_order && f1.even_data_lines >= MIN_FILL_LINES_PF) { proc_state = tb ? STG_AB_BFF_TO_BFF : STG_AB_TFF_TO_TFF; en_sw_order = false; }
                    else proc_state = STG_PAD_NO_GOOD;
                }
            } else break;
            if (stage_count > STG_PAD_MAX) break;
        }
    }

    /* ---- getAssemblyFieldOrder (:4278-4380) ---- */
    __device__ inline uint8_t get_assembly_field_order()
    {
        uint8_t cur = ORDER_UNK;
        if (order_set(f1)) { cur = f1.field_order; if (!f1.order_preset) push_order = cur; }
        else {
            if (f2.order_preset && order_set(f2)) cur = f2.field_order;
            else if (order_set(f0) && f0.outer_padding_ok) cur = f0.field_order;
        }
        if (cur != ORDER_TFF && cur != ORDER_BFF) {
            if (prob_order == ORDER_TFF || prob_order == ORDER_BFF) cur = prob_order;
            else if (f1.tff_cnt < f1.bff_cnt) cur = ORDER_TFF;
            else if (f1.tff_cnt > f1.bff_cnt) cur = ORDER_BFF;
            else cur = ORDER_TFF;
        }
        if (!order_set(f1)) { f1.field_order = cur; if (!f1.order_preset) f1.order_guessed = 1; }
        return cur;
    }

    /* ---- conv_queue writers (addLinesFromField :4452-4518, addFieldPadding :4521-4571) ---- */
    /* could performCWD change this line or be changed by it: a word that failed its CRC on a line that is still unrepaired and eligible,
     * or repaired already (perform_cwd's `mine`, without its frame test).  No such line in the queue (padding lines are none): nothing to scan. */
    bool cwd_cand;
    __device__ static inline bool cwd_candidate(const SLine &l)
    {
        const bool forced = (l.flags & SL_FORCED_BAD) != 0, failed = forced || ((l.wcrc & 0xFF) != 0xFF);
        return failed && ((!crc_valid_if(l) && (l.flags & SL_COORDS_VALID) && !forced) || crc_valid(l));
    }
/* How many steps' lines the turn kernel asks for at once where it copies lines (1 = a step at a time).  The analysis kernel gains a lot
 * from asking for four steps at once (its pass 3); the turn kernel loses by it, every time: it runs at its 128-register limit and what the
 * batches hold in registers is spilled elsewhere (per 10 000-frame call, stitch kernels together: none 0.68 ms, queue fill x4 0.70,
 * tail + hand-over x2 0.72, the trial's first 176 lines x3 0.73; profiles/r04_tuning_notes.md section 7) */
#ifndef SDV_ST_FILL_U
#define SDV_ST_FILL_U 1
#endif
#ifndef SDV_ST_TAIL_U
#define SDV_ST_TAIL_U 1
#endif
#ifndef SDV_ST_RING_U
#define SDV_ST_RING_U 1
#endif
#ifndef SDV_ST_TRY_BATCH
#define SDV_ST_TRY_BATCH 0
#endif
    /* the frame number of the queue's last line when the writers below know it without asking the memory (prescan_frame's question) */
    uint32_t q_last_frame; bool q_last_known;
    __device__ inline uint16_t add_lines(const Field &f, uint16_t start, uint16_t count, uint16_t &last_line)
    {
        if (!(BUF_FIELD >= (int)start && BUF_FIELD >= (int)start + (int)count)) return 0;
        if (count == 0) return 0;
        /* four steps' lines asked for at once (a field is one such batch); the last of them says where the numbering goes on */
        const int e = (int)count - 1;
        uint32_t e_line = 0, e_frame = 0;
        for (int i0 = 0; i0 < (int)count; i0 += 64 * SDV_ST_FILL_U) {
            SLine l[SDV_ST_FILL_U];
#pragma unroll
            for (int u = 0; u < SDV_ST_FILL_U; u++) { const int i = i0 + 64 * u + lane; l[u].frame = 0; l[u].line = 0; if (i < (int)count) l[u] = f.get(start + i); }
#pragma unroll
            for (int u = 0; u < SDV_ST_FILL_U; u++) {
                const int i = i0 + 64 * u + lane;
                if (i < (int)count && qn + i < QCAP) { cwd_cand |= cwd_candidate(l[u]); q[qn + i] = l[u]; }
                if (i == e) { e_line = l[u].line; e_frame = l[u].frame; }
            }
        }
        last_line = (uint16_t)((uint32_t)__shfl((int)e_line, e & 63) + 2u);
        q_last_frame = (uint32_t)__shfl((int)e_frame, e & 63); q_last_known = true;
        qn += count; if (qn > QCAP) { qn = QCAP; overflow = true; q_last_known = false; }
        return count;
    }
    __device__ inline uint16_t add_padding(uint32_t frame, uint16_t count, uint16_t &last_line)
    {
        for (int i = lane; i < (int)count; i += 64) if (qn + i < QCAP) q[qn + i] = sline_empty(frame, (uint16_t)(last_line + 2 * i));
        last_line = (uint16_t)(last_line + 2 * count);
        if (count > 0) { q_last_frame = frame; q_last_known = true; }
        qn += count; if (qn > QCAP) { qn = QCAP; overflow = true; q_last_known = false; }
        return count;
    }
    bool overflow;

    /* ---- fillFrameForOutput (:4588-5387) ----
     * Every branch of the reference's decision tree adds the same kinds of things in the same order - [padding a], lines of the first
     * field, [padding b], lines of the second field, [padding c], [padding d] - so the tree only says how much of each, and the
     * queue is written at one place behind it. */
    __device__ inline void fill_frame_for_output()
    {
        uint16_t c1, c2, last_line = 0, lines_to_fill = 0, added_inner = 0, added_outer = 0;
        const uint8_t order = get_assembly_field_order();
        Field p1, p2;
        if (order == ORDER_TFF) {
            p1 = field(1, 0); p2 = field(1, 1);
            if (order_set(f0) && f0.field_order != ORDER_TFF) f0.outer_padding_ok = 0;
        } else {
            p1 = field(1, 1); p2 = field(1, 0);
            if (order_set(f0) && f0.field_order != ORDER_BFF) f0.outer_padding_ok = 0;
        }
        c1 = (uint16_t)p1.size; c2 = (uint16_t)p2.size;
        const int target = f1.video_standard == VID_PAL ? LINES_PF_PAL : LINES_PF_NTSC;
        if (c1 > target) c1 = (uint16_t)target;
        if (c2 > target) c2 = (uint16_t)target;
        const bool insert_top_line = cfg.fix_cut_above != 0;
        const uint32_t fr = f1.frame_number;
        const uint16_t first_ln = order == ORDER_TFF ? 1 : 2, second_ln = order == ORDER_TFF ? 2 : 1;
        uint16_t pa = 0, s1 = 0, n1 = 0, pb = 0, s2 = 0, n2 = 0, pc = 0, pd = 0;       /* the plan */
#define L1(st, cnt) (s1 = (uint16_t)(st), n1 = (uint16_t)(cnt))
#define L2(st, cnt) (s2 = (uint16_t)(st), n2 = (uint16_t)(cnt))
#define PA(cnt) (pa = (uint16_t)(cnt))
#define PB(cnt) (pb = (uint16_t)(cnt))
#define PC(cnt) (pc = (uint16_t)(cnt))
#define PD(cnt) (pd = (uint16_t)(cnt))
        if (file_start) {
            f0.frame_number = 0;
            f0.even_resolution = f0.odd_resolution = order == ORDER_TFF ? f1.odd_resolution : f1.even_resolution;
            last_line = f1.video_standard == VID_PAL ? LINES_PF_PAL : LINES_PF_NTSC;
            const uint8_t add_count = 80;             /* STC007DataBlock::LINE_R2 */
            last_line = (uint16_t)((last_line * 2) - (add_count * 2));
            add_padding(0, add_count, last_line);
            last_line = 0;
        }
        if (f0.outer_padding_ok) {
            if (f1.inner_padding_ok) {
                if (f1.outer_padding_ok) {
                    lines_to_fill = (uint16_t)(c1 + c2 + f1.inner_padding + f1.outer_padding);
                    if ((target * 2) == lines_to_fill) {
                        L1(0, c1); added_inner = PB(f1.inner_padding);
                        L2(0, c2); added_outer = PC(f1.outer_padding);
                    } else if ((target * 2) > lines_to_fill) {
                        lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                        L1(0, c1); added_inner = PB(f1.inner_padding);
                        L2(0, c2); added_outer = PC(f1.outer_padding); added_outer = (uint16_t)(added_outer + PD(lines_to_fill));
                        f1.outer_padding_ok = 0; set_order_unknown(f2);
                    } else {
                        lines_to_fill = (uint16_t)(c1 + c2 + f1.inner_padding);
                        if ((target * 2) >= lines_to_fill) {
                            lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                            L1(0, c1); added_inner = PB(f1.inner_padding);
                            L2(0, c2); added_outer = PC(lines_to_fill);
                        } else {
                            lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                            L1(0, c1); added_inner = PB(f1.inner_padding);
                            L2(0, c2 - lines_to_fill);
                        }
                        f1.outer_padding_ok = 0; set_order_unknown(f2);
                    }
                } else {
                    lines_to_fill = (uint16_t)(c1 + c2 + f1.inner_padding);
                    if ((target * 2) >= lines_to_fill) {
                        lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                        L1(0, c1); added_inner = PB(f1.inner_padding);
                        L2(0, c2); added_outer = PC(lines_to_fill);
                    } else {
                        lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                        L1(0, c1); added_inner = PB(f1.inner_padding);
                        L2(0, c2 - lines_to_fill);
                    }
                }
            } else if (f1.outer_padding_ok) {
                lines_to_fill = (uint16_t)(c1 + c2 + f1.outer_padding);
                if ((target * 2) >= lines_to_fill) {
                    lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                    L1(0, c1); added_inner = PB(lines_to_fill);
                    L2(0, c2); added_outer = PC(f1.outer_padding);
                } else {
                    lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                    L1(0, c1);
                    L2(lines_to_fill, c2 - lines_to_fill); added_outer = PC(f1.outer_padding);
                }
            } else {
                lines_to_fill = (uint16_t)(c1 + c2);
                if ((target * 2) >= lines_to_fill) {
                    L1(0, c1); added_inner = PB(target - c1);
                    L2(0, c2); added_outer = PC(target - c2);
                } else {
                    lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                    L1(0, c1);
                    L2(0, c2 - lines_to_fill);
                }
            }
        } else if (f1.inner_padding_ok) {
            if (f1.outer_padding_ok) {
                lines_to_fill = (uint16_t)(c1 + c2 + f1.inner_padding + f1.outer_padding);
                if ((target * 2) >= lines_to_fill) {
                    lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                    added_inner = PA(lines_to_fill); L1(0, c1); added_inner = (uint16_t)(added_inner + PB(f1.inner_padding));
                    L2(0, c2); added_outer = PC(f1.outer_padding);
                } else {
                    lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                    L1(lines_to_fill, c1 - lines_to_fill); added_inner = PB(f1.inner_padding);
                    L2(0, c2); added_outer = PC(f1.outer_padding);
                }
            } else {
                lines_to_fill = (uint16_t)(c1 + c2 + f1.inner_padding);
                if ((target * 2) >= lines_to_fill) {
                    lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                    L1(0, c1); added_inner = PB(f1.inner_padding);
                    L2(0, c2); added_outer = PC(lines_to_fill);
                } else {
                    lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                    L1(0, c1); added_inner = PB(f1.inner_padding);
                    L2(0, c2 - lines_to_fill);
                }
            }
        } else if (f1.outer_padding_ok) {
            lines_to_fill = (uint16_t)(c1 + c2 + f1.outer_padding);
            if ((target * 2) >= lines_to_fill) {
                lines_to_fill = (uint16_t)((target * 2) - lines_to_fill);
                L1(0, c1); added_inner = PB(lines_to_fill);
                L2(0, c2); added_outer = PC(f1.outer_padding);
            } else {
                lines_to_fill = (uint16_t)(lines_to_fill - (target * 2));
                L1(0, c1 - lines_to_fill);
                L2(0, c2); added_outer = PC(f1.outer_padding);
            }
        } else {
            lines_to_fill = (uint16_t)(c1 + c2);
            if ((target * 2) >= lines_to_fill) {
                if (insert_top_line && c1 > 0 && c2 > 0) {
                    if (order == ORDER_BFF) {
                        added_outer = PA(1); L1(0, c1); c1++; added_inner = PB(target - c1);
                        L2(0, c2); added_outer = (uint16_t)(added_outer + PC(target - c2));
                    } else {
                        L1(0, c1); added_inner = PB(target - c1 + 1);
                        L2(0, c2); c2++; added_outer = PC(target - c2);
                    }
                } else {
                    L1(0, c1); added_inner = PB(target - c1);
                    L2(0, c2); added_outer = PC(target - c2);
                }
            } else {
                if (c1 < target) { L1(0, c1); added_inner = PB(target - c1); } else L1(0, target);
                if (c2 < target) { L2(0, c2); added_outer = PC(target - c2); } else L2(0, target);
            }
        }
#undef L1
#undef L2
#undef PA
#undef PB
#undef PC
#undef PD
        last_line = first_ln;
        add_padding(fr, pa, last_line); add_lines(p1, s1, n1, last_line); add_padding(fr, pb, last_line);
        last_line = second_ln;
        add_lines(p2, s2, n2, last_line); add_padding(fr, pc, last_line); add_padding(fr, pd, last_line);
        if (file_end) { last_line = 1; add_padding(f2.frame_number, MIN_DEINT, last_line); }
        f1.inner_padding = added_inner;
        f1.outer_padding = added_outer;
    }

    /* ---- CWD pre-scan (prescanFrame :6401-6452, performCWD :5905-6398) ---- */
    __device__ inline uint8_t queue_res_mode() const      /* getDataBlockResolution(queue, 0) */
    {
        if (cfg.m2) return SDV_RES_MODE_14BIT;
        if (qn <= MIN_DEINT) return SDV_RES_MODE_14BIT_AUTO;
        return (uint8_t)uni(block_res_mode(q[0], q[MIN_DEINT]));
    }
    __device__ static inline void cwd_after_patch(SLine &l, bool &fixed)
    {
        if (crc_valid_if(l)) { l.wvalid = 0x1FF; fixed = true; }
    }
    __device__ inline bool perform_cwd()
    {
        const sdv_deint_settings ds = deint_cfg(queue_res_mode(), cfg.ignore_crc, !cfg.ignore_crc, cfg.en_p, cfg.en_q, true);
        const int nblk = qn - MIN_DEINT;
        bool fixed_any = false;
        WsSrc src; src.q = q;
        /* Every change below is made to a line whose word failed its CRC (block.line_crc false) and that is either still
         * unrepaired and eligible (bad CRC, valid coordinates, not forced bad, not of frame B) or already repaired (CRC valid
         * again: it may get forced bad).  A block none of whose eight lines is such a line cannot change anything, so each
         * class only decodes the blocks that touch one: candidate bits of the queue by ballot, then per class. */
        __shared__ uint64_t s_bad[QCAP / 64];
        bool any_bad = false;
        if (!cfg.ignore_crc && !__ballot(cwd_cand)) return false;     /* no line that entered the queue is a candidate: the common case on a clean tape */
        uint32_t mine = 0;                      /* bit c: line c * 64 + lane is a candidate (loads first, ballots after: they pipeline) */
        for (int c = 0; c * 64 < qn; c++) {
            const int i = c * 64 + lane;
            if (i < qn) {
                const SLine &l = q[i];
                const bool forced = (l.flags & SL_FORCED_BAD) != 0, failed = forced || ((l.wcrc & 0xFF) != 0xFF);
                const bool eligible = !crc_valid_if(l) && (l.flags & SL_COORDS_VALID) && !forced && l.frame != f2.frame_number;
                if (failed && (eligible || crc_valid(l))) mine |= 1u << c;
            }
        }
        any_bad = __ballot(mine != 0) != 0;
        if (!any_bad && !cfg.ignore_crc) return false;          /* nothing in the queue can change: the common case on a clean tape */
        for (int c = 0; c * 64 < qn; c++) {
            const uint64_t m = __ballot((mine >> c) & 1);
            if (lane == 0) s_bad[c] = m;
        }
        __syncthreads();
        if (lane < ILV) {
            uint64_t need = ~0ull;
            if (!cfg.ignore_crc) {
                uint64_t cb = 0;
                for (int j = 0; lane + ILV * j < qn; j++) { const int i = lane + ILV * j; cb |= ((s_bad[i >> 6] >> (i & 63)) & 1ull) << j; }
                need = cb | (cb >> 1) | (cb >> 2) | (cb >> 3) | (cb >> 4) | (cb >> 5) | (cb >> 6) | (cb >> 7);
            }
            for (int j = 0; lane + ILV * j < nblk; j++) {
                if (!((need >> j) & 1)) continue;
                const int ofs = lane + ILV * j;
                Block b; sdvd::Lines8 l8;
                sdvd::gather8(src, (size_t)ofs, l8);
                sdvd::process_block(ds, l8, 0, b);
                const int max_fixable = (!cfg.en_q || b.resolution == SDV_RES_16BIT) ? sdvd::WORD_P0 : sdvd::WORD_Q0;
                const bool data_fixed = (~b.line_crc & b.word_valid & 0xFF) != 0;          /* isDataFixed, stc007datablock.cpp:371-384 */
                if (!(blk_valid(b) && data_fixed)) continue;
                for (int wi = 0; wi <= max_fixable; wi++) {
                    if (sdvd::bit(b.line_crc, wi)) continue;
                    SLine l = q[ofs + ILV * wi];
                    const bool forced = (l.flags & SL_FORCED_BAD) != 0;
                    if (!crc_valid_if(l) && (l.flags & SL_COORDS_VALID) && !forced && l.frame != f2.frame_number) {
                        const uint16_t wbit = (uint16_t)(1u << wi);
                        if (b.resolution == SDV_RES_14BIT) {
                            if (sl_word(l, wi) != b.w(wi)) {
                                sl_set_word(l, wi, (uint16_t)(b.w(wi) & 0x3FFF));      /* setWord keeps the word's CRC flag */
                                l.calc_crc = crc_words(l.words);
                                l.wvalid |= wbit;
                                cwd_after_patch(l, fixed_any);
                            } else l.wvalid |= wbit;
                            if (!crc_valid_if(l)) {
                                if ((l.wvalid & 0xFF) == 0xFF) { l.calc_crc = crc_words(l.words); l.words[8] = l.calc_crc; l.wvalid |= 0x100; fixed_any = true; }
                            }
                        } else {
                            const uint16_t old_word = sl_word(l, wi);
                            uint16_t old_bitword = l.words[7], new_word = b.w(wi), new_bitword = (uint16_t)(new_word & 3);
                            new_word = (uint16_t)(new_word >> 2);
                            const int ofs_b = 12 - 2 * wi;
                            new_bitword = (uint16_t)(new_bitword << ofs_b);
                            old_bitword = (uint16_t)(old_bitword & (3 << ofs_b));
                            if (old_word != new_word) {
                                /* setWord(index, word, isWordCRCOk(index)) also rewrites word_valid with the CRC flag (stc007line.cpp:158-173) */
                                const bool wc = !forced && (l.wcrc & wbit);
                                sl_set_word(l, wi, (uint16_t)(new_word & 0x3FFF));
                                l.wcrc = wc ? (l.wcrc | wbit) : (l.wcrc & ~wbit);
                                l.calc_crc = crc_words(l.words);
                                l.wvalid |= wbit;
                                cwd_after_patch(l, fixed_any);
                            }
                            if (!crc_valid_if(l)) {
                                if (old_bitword != new_bitword) {
                                    old_bitword = (uint16_t)(l.words[7] & ~(3 << ofs_b));
                                    const bool qc = !forced && (l.wcrc & 0x80);
                                    l.words[7] = (uint16_t)((old_bitword | new_bitword) & 0x3FFF);
                                    l.wcrc = qc ? (l.wcrc | 0x80) : (l.wcrc & ~0x80);
                                    l.wvalid = qc ? (l.wvalid | 0x80) : (l.wvalid & ~0x80);
                                    l.calc_crc = crc_words(l.words);
                                    cwd_after_patch(l, fixed_any);
                                }
                            }
                        }
                        q[ofs + ILV * wi] = l;
                    } else if (crc_valid(l)) {
                        if (b.resolution == SDV_RES_14BIT && sl_word(l, wi) != b.w(wi)) {
                            l.flags |= SL_FORCED_BAD; q[ofs + ILV * wi] = l;
                            const int jl = j + wi;                      /* the line now counts as failed for the blocks still to come */
                            need |= jl >= 7 ? (0xFFull << (jl - 7)) : (0xFFull >> (7 - jl));
                        }
                    }
                }
            }
        }
        return __ballot(fixed_any) != 0;
    }
    __device__ inline void prescan_frame()
    {
        if (!cfg.en_cwd) return;
        bool next = false;
        const int qn_own = qn;                                        /* the queue without frame B's look-ahead lines */
        const bool own_known = q_last_known && qn > 0; const uint32_t own_frame = q_last_frame;
        if (f1.outer_padding_ok && order_set(f1)) {                   /* fillNextFieldForCWD :5390-5456 */
            uint16_t last_line = f1.field_order == ORDER_TFF ? 1 : 2;
            Field p = f1.field_order == ORDER_TFF ? field(2, 0) : field(2, 1);
            uint16_t cnt = (uint16_t)p.size;
            if (cnt > MIN_DEINT) cnt = MIN_DEINT;
            add_lines(p, 0, cnt, last_line);
            next = true;
        }
        __syncthreads();
        for (;;) { bool more = perform_cwd(); __syncthreads(); if (!more) break; }
        if (next) {                                                   /* removeNextFieldAfterCWD: every trailing line of frame B */
            qn = qn_own;                                              /* the look-ahead lines are frame B's by construction ... */
            /* ... and so is the end-of-file flush, if any (the writers usually know the frame of the line the queue ended with: no need to ask) */
            if (!(own_known && own_frame != f2.frame_number))
                while (qn > 0 && uni(q[qn - 1].frame) == f2.frame_number) qn--;
        }
    }

    /* ---- performDeinterleave (:6675-6885) + outputSamplePair (:6525-6569) ---- */
    sdv_sample_pair *out_pairs; uint32_t n_pairs;
    uint32_t pair_cap; bool clipped;       /* room for this turn's pairs; clipped: it wanted more */
    __device__ static inline sdv_sample_pair service_pair(uint8_t srv)
    {
        sdv_sample_pair p;
        p.audio_word[0] = p.audio_word[1] = 0; p.sample_flags[0] = p.sample_flags[1] = 0; p.sample_rate = 44056; p.emphasis = 0; p.service_type = srv; p._pad = 0;
        return p;
    }
    uint32_t *pairbuf;                      /* LDS: 64 x 3 pairs as dwords */
    sdv_block_rec *blocks_out;              /* this turn's place in the block stream, or NULL */
    uint32_t *asm_cnt_out; sdv_asm_line_rec *asm_out;      /* the same for the assembled lines: a count to leave, or a place to write to */
    __device__ static inline void pack_pair(const sdv_sample_pair &p, uint32_t *o)
    {
        o[0] = (uint32_t)(uint16_t)p.audio_word[0] | ((uint32_t)(uint16_t)p.audio_word[1] << 16);
        o[1] = (uint32_t)p.sample_flags[0] | ((uint32_t)p.sample_flags[1] << 8) | ((uint32_t)p.sample_rate << 16);
        o[2] = (uint32_t)p.emphasis | ((uint32_t)p.service_type << 8) | ((uint32_t)p._pad << 16);
    }
    __device__ inline sdv_sample_pair make_pair(const Block &b, int il, int ir, uint16_t rate) const
    {
        sdv_sample_pair p = service_pair(SDV_PAIR_SRV_NO);
        if (rate < 44101) p.sample_rate = rate;
        bool block_state = false, wl = false, wr = false, fl = false, fr = false;
        if (b.audio_state != SDV_AUD_BROKEN) {
            block_state = blk_valid(b);
            if (block_state) { fl = sdvd::bit(b.line_crc, il); fr = sdvd::bit(b.line_crc, ir); }
            wl = sdvd::bit(b.word_valid, il); wr = sdvd::bit(b.word_valid, ir);
        }
        p.audio_word[0] = get_sample(b, il, cfg.m2); p.audio_word[1] = get_sample(b, ir, cfg.m2);
        p.sample_flags[0] = (uint8_t)((block_state ? SDV_SF_BLOCK_OK : 0) | (wl ? SDV_SF_WORD_VALID : 0) | (fl ? SDV_SF_WORD_FIXED : 0));
        p.sample_flags[1] = (uint8_t)((block_state ? SDV_SF_BLOCK_OK : 0) | (wr ? SDV_SF_WORD_VALID : 0) | (fr ? SDV_SF_WORD_FIXED : 0));
        return p;
    }
    __device__ inline void perform_deinterleave()
    {
        const int nblk = qn > MIN_DEINT ? qn - MIN_DEINT : 0;
        if (asm_cnt_out || asm_out) {
            /* "dump the whole line buffer out (for visualization)" (:6689-6704): the lines of frame A and frame B as they stand behind the CWD pass */
            uint32_t made = 0;
            for (int c = 0; c * 64 < qn; c++) {
                const int i = c * 64 + lane;
                SLine l; bool mine = false;
                if (i < qn) { l = q[i]; mine = l.frame == f1.frame_number || l.frame == f2.frame_number; }
                const uint64_t mm = __ballot(mine);
                if (asm_out && mine) {
                    sdv_asm_line_rec r;
                    const bool forced = (l.flags & SL_FORCED_BAD) != 0;
                    r.frame_number = l.frame; r.line_number = l.line;
#pragma unroll
                    for (int w = 0; w < 9; w++) r.words[w] = l.words[w];
                    r.calc_crc = l.calc_crc;
                    r.word_crc_ok = forced ? (uint16_t)0 : (uint16_t)(l.wcrc & 0x1FF); r.word_valid = forced ? (uint16_t)0 : (uint16_t)(l.wvalid & 0x1FF);
                    r.flags = (uint8_t)((forced ? SDV_AL_FORCED_BAD : 0) | ((l.flags & SL_MARKERS) ? SDV_AL_MARKERS : 0) | (crc_valid(l) ? SDV_AL_CRC_VALID : 0));
                    r._pad = 0;
                    asm_out[made + (uint32_t)__popcll(mm & lanemask_lt(lane))] = r;
                }
                made += (uint32_t)__popcll(mm);
            }
            if (asm_cnt_out && lane == 0) *asm_cnt_out = made;
        }
        RingSrc src; src.ring = ring;
        __syncthreads();                                              /* whoever used the ring before is done with it */
        {
#if SDV_ST_RING_U == 3
            SLine got[3];
#pragma unroll
            for (int u = 0; u < 3; u++) { const int i = lane + 64 * u; if (i < RING_SPAN && i < qn) got[u] = q[i]; }
#pragma unroll
            for (int u = 0; u < 3; u++) { const int i = lane + 64 * u; if (i < RING_SPAN && i < qn) ring[i] = view(got[u]); }
#else
            for (int i = lane; i < RING_SPAN && i < qn; i += 64) ring[i] = view(q[i]);
#endif
        }
        uint16_t rate = (cfg.preset_sample_rate == 44100 || cfg.preset_sample_rate == 44056) ? cfg.preset_sample_rate
                        : (f1.video_standard == VID_NTSC ? (uint16_t)44056 : (uint16_t)44100);        /* setBlockSampleRate :6455-6480 */
        uint8_t cd = broken_countdown;
        uint32_t fix_p = 0, fix_q = 0, fix_cwd = 0, drop = 0, sdrop = 0, brk_field = 0;
        for (int c = 0; c * 64 < nblk; c++) {
            const int i = c * 64 + lane;
            const bool act = i < nblk;
            /* the next step's new lines: asked for now, stored behind this step's decodes */
            const int nx = c * 64 + RING_SPAN + lane;
            const bool has_nx = nx < qn && (c + 1) * 64 < nblk;
            SLine nxl; if (has_nx) nxl = q[nx];
            SDV_LDS_WAVE_SYNC();
            Block b; sdvd::blk_clear(b);
            bool ns = false, seam = false, brk = false;
            if (act) {
                sdvd::Lines8 l8;
                sdvd::gather8(src, (size_t)i, l8);
                const uint8_t mode = cfg.m2 ? (uint8_t)SDV_RES_MODE_14BIT
                                     : res_mode_for_seam(line_res(l8.l[0].frame_number, l8.l[0].line_number), line_res(l8.l[7].frame_number, l8.l[7].line_number));
                sdvd::process_block(deint_cfg(mode, cfg.ignore_crc, !cfg.ignore_crc, cfg.en_p, cfg.en_q, cfg.en_cwd), l8, 0, b);
                ns = !blk_silent(b, cfg.m2);
                if (ns && cfg.mask_seams) {
                    if (!f1.inner_padding_ok && !f1.inner_silence)
                        if (b.w_line[0] > b.w_line[7] && b.w_frame[0] == f1.frame_number && b.w_frame[0] == b.w_frame[7]) seam = true;
                    if (!f0.outer_padding_ok && !f0.outer_silence)
                        if (b.w_frame[0] != b.w_frame[7] && b.w_frame[0] == f0.frame_number && b.w_frame[7] == f1.frame_number) seam = true;
                }
                brk = b.audio_state == SDV_AUD_BROKEN;
                if (seam) mark_unsafe(b);
            }
            const uint64_t m_ns = __ballot(ns), m_seam = __ballot(seam), m_brk = __ballot(brk);
            uint64_t m_cd = 0;
            int cnt = nblk - c * 64; if (cnt > 64) cnt = 64;
            /* nothing counts down and no broken block starts a countdown in this chunk: nothing to replay (the usual case) */
            if (!(cd == 0 && (cfg.broken_mask_dur == 0 || (m_brk & m_ns & ~m_seam) == 0)))
            for (int j = 0; j < cnt; j++) {
                if (((m_ns >> j) & 1) && !((m_seam >> j) & 1)) {
                    if (cfg.broken_mask_dur > 0 && cd == 0 && ((m_brk >> j) & 1)) cd = cfg.broken_mask_dur;
                    if (cd != 0) m_cd |= 1ull << j;
                }
                if (cd > 0) cd--;
            }
            bool rep = false, valid = false;
            int errs = 0;
            if (act) {
                if ((m_cd >> lane) & 1) mark_unsafe(b);
                rep = !((file_start && b.w_frame[0] == f0.frame_number) || (file_end && b.w_frame[7] == f2.frame_number));      /* isBlockNoReport */
                valid = blk_valid(b);
                errs = errors_audio_fixed(b);
            }
            fix_p += (uint32_t)__popcll(__ballot(rep && valid && b.audio_state == SDV_AUD_FIX_P));
            fix_q += (uint32_t)__popcll(__ballot(rep && valid && b.audio_state == SDV_AUD_FIX_Q));
            fix_cwd += (uint32_t)__popcll(__ballot(rep && valid && b.cwd_applied && b.cwd_fixed != 0));
            const uint64_t m_drop = __ballot(rep && !valid);
            if (m_drop) {                                   /* dropped blocks are the exception: their statistics only then */
                drop += (uint32_t)__popcll(m_drop);
                brk_field += (uint32_t)__popcll(__ballot(rep && !valid && b.audio_state == SDV_AUD_BROKEN));
                for (int e = 1; e <= 6; e++) sdrop += (uint32_t)e * (uint32_t)__popcll(__ballot(rep && !valid && errs == e));
            }
            if (blocks_out && act) {                /* the block as outputDataBlock hands it to the visualiser (newBlockProcessed, :6626) */
                uint32_t r[18];                     /* an sdv_block_rec as 18 dwords: nine 8-byte stores instead of 28 narrow ones */
#pragma unroll
                for (int w = 0; w < 8; w++) r[w] = b.w_frame[w];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    r[8 + w] = (uint32_t)b.w_line[2 * w] | ((uint32_t)b.w_line[2 * w + 1] << 16);
                    r[12 + w] = (uint32_t)b.w(2 * w) | ((uint32_t)b.w(2 * w + 1) << 16);
                }
                r[16] = (uint32_t)b.line_crc | ((uint32_t)b.cwd_fixed << 8) | ((uint32_t)b.word_valid << 16) | ((uint32_t)b.resolution << 24);
                r[17] = (uint32_t)b.audio_state | ((b.cwd_applied ? 1u : 0u) << 8) | ((uint32_t)rate << 16);
                uint2 *o2 = (uint2 *)(blocks_out + i);
#pragma unroll
                for (int w = 0; w < 9; w++) { uint2 v; v.x = r[2 * w]; v.y = r[2 * w + 1]; o2[w] = v; }
            }
            /* the 64 blocks' 192 pairs are one stretch of 2 304 bytes: through LDS, then whole dwords side by side */
            if (act) {
                uint32_t *pb = pairbuf + 9 * lane;
                pack_pair(make_pair(b, 0, 1, rate), pb); pack_pair(make_pair(b, 2, 3, rate), pb + 3); pack_pair(make_pair(b, 4, 5, rate), pb + 6);
            }
            SDV_LDS_WAVE_SYNC();
            {
                const uint32_t first = n_pairs + 192u * (uint32_t)c;
                uint32_t *o32 = (uint32_t *)out_pairs + 3u * (size_t)first;
                for (uint32_t d = (uint32_t)lane; d < 9u * (uint32_t)cnt; d += 64u)
                    if (first + 3u * (d / 9u) + 3u <= pair_cap) o32[d] = pairbuf[d];          /* a block's three pairs fit or none of them is written */
            }
            if (has_nx) ring[nx & (RING - 1)] = view(nxl);             /* slots of lines this step no longer reads */
        }
        if (nblk > 0) {
            f1.blocks_total = (uint16_t)(f1.blocks_total + nblk);
            f1.odd_sample_rate = f1.even_sample_rate = rate;
            f1.blocks_fix_p = (uint16_t)(f1.blocks_fix_p + fix_p); f1.blocks_fix_q = (uint16_t)(f1.blocks_fix_q + fix_q); f1.blocks_fix_cwd = (uint16_t)(f1.blocks_fix_cwd + fix_cwd);
            f1.blocks_drop = (uint16_t)(f1.blocks_drop + drop); f1.samples_drop = (uint16_t)(f1.samples_drop + sdrop);
            f1.blocks_broken_field = (uint16_t)(f1.blocks_broken_field + brk_field);
            n_pairs += 3u * (uint32_t)nblk; if (n_pairs > pair_cap) { n_pairs = pair_cap; clipped = true; }
        }
        broken_countdown = cd;
        /* what stays in conv_queue: the last (at most) 112 lines */
        tail_ofs = nblk;
    }
    int tail_ofs;
};

__device__ inline void frasm_set_trim(Frasm &fr, const FrameLocal &fl)      /* what findFramesTrim left in the descriptor */
{
    fr.odd_top_data = fl.top[0]; fr.odd_bottom_data = fl.bottom[0]; fr.even_top_data = fl.top[1]; fr.even_bottom_data = fl.bottom[1];
    fr.trim_ok = (fl.flags & FL_TRIM_OK) != 0;
    fr.ctrl_index = fl.ctrl[0]; fr.ctrl_hour = fl.ctrl[1]; fr.ctrl_minute = fl.ctrl[2]; fr.ctrl_second = fl.ctrl[3]; fr.ctrl_field = fl.ctrl[4];
}
__device__ inline void frasm_set_counts(Frasm &fr, const FrameLocal &fl)    /* what splitFramesToFields counted */
{
    fr.odd_data_lines = fl.data_lines[0]; fr.even_data_lines = fl.data_lines[1]; fr.odd_valid_lines = fl.valid_lines[0]; fr.even_valid_lines = fl.valid_lines[1];
}
__device__ inline void reset_state(Step &s)      /* resetState :69-89 (the statistics rings live in the engine) */
{
    s.qn = 0; s.last_pad_counter = 0xFF; s.broken_countdown = 0;
    frasm_clear(s.f0); frasm_clear_misc(s.f1); frasm_clear_misc(s.f2);
}

/* one turn of doFrameReassemble (:7284-7457) for step k */
#ifdef SDV_EMU
#define ST_STAMP(i) ((void)0)
#else
#define ST_STAMP(i) do { if (a.timing && lane == 0) a.timing[(size_t)k * 8 + (i)] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
#endif
__device__ inline void step_body(const StepArgs &a, uint32_t work, uint32_t slot, int lane, sdv_deint_line *ring, uint32_t *pairbuf, SLine *q_lds = NULL)
{
    const uint32_t k = work & 0x3FFFFFFFu;
    const int w_prev = (work >> 30) & 1, w_cur = (work >> 31) & 1;
    ST_STAMP(0);
    const StepChain *in = k == 0 ? a.chain0 : &a.chain[w_prev][k - 1];
    StepChain *out = &a.chain[w_cur ^ 1][k];
    const StepChain *old = &a.chain[w_cur][k];
    Step s;
    s.cfg = a.cfg; s.fields = a.fields; s.k = k; s.lane = lane;
    s.l1 = a.fl[k]; s.l2 = a.fl[k + 1]; make_uniform(s.l1); make_uniform(s.l2);
    s.q = q_lds ? q_lds : a.ws + (size_t)slot * QCAP; s.overflow = false; s.ring = ring; s.pairbuf = pairbuf;
    s.blocks_out = a.blocks ? a.blocks + a.block_ofs[k] : NULL;
    s.asm_cnt_out = a.asm_cnt ? a.asm_cnt + k : NULL; s.asm_out = a.asm_lines ? a.asm_lines + a.asm_ofs[k] : NULL;
    s.prob_order = a.ctl ? a.pipe_order : (uint8_t)uni(a.prob_order[k]); s.prob_res = a.ctl ? a.pipe_res : (uint8_t)uni(a.prob_res[k]); s.push_order = ORDER_UNK;
    const bool direct = a.direct_pairs != NULL;
    s.out_pairs = direct ? a.direct_pairs + (size_t)k * a.guess_pairs : a.pairs + (size_t)k * PAIR_SLOT; s.n_pairs = 0;
    s.pair_cap = direct ? a.guess_pairs : (uint32_t)PAIR_SLOT; s.clipped = false;
    s.f0 = in->f0; make_uniform(s.f0);
    s.last_pad_counter = (uint8_t)uni(in->last_pad_counter); s.broken_countdown = (uint8_t)uni(in->broken_countdown);
    s.qn = (int)uni(in->tail_n);
    s.cwd_cand = false; s.q_last_known = false; s.q_last_frame = 0;
    {   /* the lines the turn before left over: at most 112, both steps asked for at once */
        static_assert(MIN_DEINT <= 128, "two steps");
        SLine t[2];
#if SDV_ST_TAIL_U == 2
#pragma unroll
        for (int u = 0; u < 2; u++) { const int i = lane + 64 * u; t[u].frame = 0; if (i < s.qn) t[u] = in->tail[i]; }
#pragma unroll
        for (int u = 0; u < 2; u++) { const int i = lane + 64 * u; if (i < s.qn) { s.cwd_cand |= Step::cwd_candidate(t[u]); s.q[i] = t[u]; } }
#else
        for (int u = 0; u < 2; u++) { const int i = lane + 64 * u; t[u].frame = 0; if (i < s.qn) { t[u] = in->tail[i]; s.cwd_cand |= Step::cwd_candidate(t[u]); s.q[i] = t[u]; } }
#endif
        if (s.qn > 0) {
            const int e = s.qn - 1;
            const uint32_t fr = e < 64 ? t[0].frame : t[1].frame;
            s.q_last_frame = (uint32_t)__shfl((int)fr, e & 63); s.q_last_known = true;
        }
    }
    /* waitForTwoFrames / findFramesTrim / splitFramesToFields results come from the analysis pass */
    frasm_clear(s.f1); frasm_clear(s.f2);
    s.f1.frame_number = s.l1.frame_number; s.f2.frame_number = s.l2.frame_number;
    s.file_start = (s.l1.flags & FL_NEW_FILE) != 0;
    s.file_end = ((s.l1.flags | s.l2.flags) & FL_END_FILE) != 0;
    frasm_set_trim(s.f1, s.l1); frasm_set_trim(s.f2, s.l2);
    if (s.file_start) { reset_state(s); s.q_last_known = false; }
    frasm_set_counts(s.f1, s.l1); frasm_set_counts(s.f2, s.l2);
    frasm_clear_asm_stats(s.f1);
    s.f1.odd_ref = s.l1.ref[0]; s.f1.even_ref = s.l1.ref[1];
    ST_STAMP(1);
    s.find_field_stitching();
    ST_STAMP(2);
    sdv_frame_asm *fo = direct ? a.direct_frasm + (size_t)k * a.guess_frasm : a.frasm + (size_t)k * FRASM_SLOT;
    const uint32_t frasm_cap = direct ? a.guess_frasm : (uint32_t)FRASM_SLOT;
    uint8_t n_frasm = 0;
    if (s.file_start) {
        Frasm sd; frasm_clear(sd); sd.service_type = 1;
        if (n_frasm < frasm_cap && s.pair_cap >= 1) { if (lane == 0) { frasm_to_pod(sd, fo[n_frasm]); s.out_pairs[0] = Step::service_pair(SDV_PAIR_SRV_NEW_FILE); } }
        else s.clipped = true;
        n_frasm++; s.n_pairs = 1;
    }
    s.fill_frame_for_output();
    ST_STAMP(3);
    s.prescan_frame();
    __syncthreads();
    ST_STAMP(4);
    s.perform_deinterleave();
    ST_STAMP(5);
    if (n_frasm < frasm_cap) { if (lane == 0) frasm_to_pod(s.f1, fo[n_frasm]); } else s.clipped = true;
    n_frasm++;
    s.f0 = s.f1;
    /* the next turn reads frasm_f0's geometry, order, paddings and resolutions only: keep the per-frame statistics out of
     * the hand-over so that it does not differ between rounds without consequence */
    frasm_clear_asm_stats(s.f0); s.f0.odd_sample_rate = s.f0.even_sample_rate = 0; s.f0.odd_valid_lines = s.f0.even_valid_lines = 0;
    int tail_n = s.qn - s.tail_ofs;
    if (s.file_end) {
        Frasm sd; frasm_clear(sd); sd.service_type = 2;
        if (n_frasm < frasm_cap && s.n_pairs < s.pair_cap) { if (lane == 0) { frasm_to_pod(sd, fo[n_frasm]); s.out_pairs[s.n_pairs] = Step::service_pair(SDV_PAIR_SRV_END_FILE); } }
        else s.clipped = true;
        n_frasm++; s.n_pairs++;
        reset_state(s); tail_n = 0;
    }
    /* hand over to the next turn; note whether anything differs from what this turn produced last time */
    bool diff = false;
    {
        uint32_t xw[sizeof(Frasm) / 4], yw[sizeof(Frasm) / 4];
        const Frasm of = old->f0;
        __builtin_memcpy(xw, &s.f0, sizeof(Frasm)); __builtin_memcpy(yw, &of, sizeof(Frasm));
        for (unsigned i = 0; i < sizeof(Frasm) / 4; i++) diff = diff || xw[i] != yw[i];
        diff = diff || old->last_pad_counter != s.last_pad_counter || old->broken_countdown != s.broken_countdown || old->tail_n != (uint16_t)tail_n;
#if SDV_ST_TAIL_U == 2
        SLine tl[2], tol[2];
#pragma unroll
        for (int h = 0; h < 2; h++) { const int i = lane + 64 * h; if (i < tail_n) { tl[h] = s.q[s.tail_ofs + i]; tol[h] = old->tail[i]; } }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int i = lane + 64 * h;
            if (i < tail_n) {
                uint32_t u[8], v[8];
                __builtin_memcpy(u, &tl[h], 32); __builtin_memcpy(v, &tol[h], 32);
                for (int w = 0; w < 8; w++) diff = diff || u[w] != v[w];
                out->tail[i] = tl[h];
            }
        }
#else
        for (int i = lane; i < tail_n; i += 64) {
            const SLine l = s.q[s.tail_ofs + i], ol = old->tail[i];
            uint32_t u[8], v[8];
            __builtin_memcpy(u, &l, 32); __builtin_memcpy(v, &ol, 32);
            for (int w = 0; w < 8; w++) diff = diff || u[w] != v[w];
            out->tail[i] = l;
        }
#endif
        if (lane == 0) { out->f0 = s.f0; out->last_pad_counter = s.last_pad_counter; out->broken_countdown = s.broken_countdown; out->tail_n = (uint16_t)tail_n; out->_pad[0] = out->_pad[1] = out->_pad[2] = 0; }
    }
    const bool changed = __ballot(diff) != 0;
    if (lane == 0) {
        StepInfo inf; inf.n_pairs = s.n_pairs; inf.n_frasm = n_frasm; inf.changed = changed ? 1 : 0; inf.push_order = s.push_order;
        inf.bits = (uint8_t)(((s.overflow || (!direct && s.clipped)) ? SI_OVERFLOW : 0) | ((direct && s.clipped) ? SI_MISS : 0) | ((s.f0.inner_padding_ok && s.f0.outer_padding_ok && order_set(s.f0)) ? SI_STEADY : 0));
        a.info[k] = inf;
    }
    __syncthreads();
    ST_STAMP(6);
}


/* ---- sdv_k_stitch_predict: what turn j will most likely hand over, for all j at once --------------------------------
 * A tape in steady state keeps its field order and paddings (the reference's own fast path, STG_TRY_PREVIOUS): turn j's
 * outcome is the template's decisions on frame j's own geometry, and its hand-over lines are the unpatched tail of
 * frame j assembled with those paddings.  The prediction only has to be right often: every turn compares what it really
 * produced with the prediction its successor was started from, and the successor is re-run on any difference. */
struct PredictStArgs { const SLine *fields; const FrameLocal *fl; const StepChain *tmpl; StepChain *out; uint32_t first, n; const uint32_t *ctl; uint32_t est_seg; };
__device__ inline void predict_st_body(const PredictStArgs &a, uint32_t j, int lane)
{
    StepChain *o = &a.out[j];
    Frasm f = a.tmpl->f0;
    const uint8_t lpc = a.tmpl->last_pad_counter;
    const FrameLocal *fl = &a.fl[j];
    const bool steady = f.inner_padding_ok && f.outer_padding_ok && order_set(f);
    int tail_n = 0;
    if (!steady) { frasm_clear(f); }
    else {
        f.frame_number = fl->frame_number;
        f.odd_top_data = fl->top[0]; f.odd_bottom_data = fl->bottom[0]; f.even_top_data = fl->top[1]; f.even_bottom_data = fl->bottom[1];
        f.odd_data_lines = fl->data_lines[0]; f.even_data_lines = fl->data_lines[1];
        f.trim_ok = (fl->flags & FL_TRIM_OK) != 0;
        f.ctrl_index = fl->ctrl[0]; f.ctrl_hour = fl->ctrl[1]; f.ctrl_minute = fl->ctrl[2]; f.ctrl_second = fl->ctrl[3]; f.ctrl_field = fl->ctrl[4];
        f.tff_cnt = f.field_order == ORDER_TFF ? lpc : 0; f.bff_cnt = f.field_order == ORDER_BFF ? lpc : 0;
        /* tail of [first field, inner padding, second field, outer padding] */
        const int p1 = f.field_order == ORDER_TFF ? 0 : 1, p2 = 1 - p1;
        const int target = f.video_standard == VID_PAL ? LINES_PF_PAL : LINES_PF_NTSC;
        int c1 = fl->data_lines[p1], c2 = fl->data_lines[p2];
        if (c1 > target) c1 = target;
        if (c2 > target) c2 = target;
        const int in_pad = f.inner_padding, out_pad = f.outer_padding, total = c1 + in_pad + c2 + out_pad;
        const SLine *l1 = field_lines(a.fields, j, p1), *l2 = field_lines(a.fields, j, p2);
        const uint16_t ln1 = c1 > 0 ? (uint16_t)(l1[c1 - 1].line + 2) : (uint16_t)(p1 == 0 ? 1 : 2);
        const uint16_t ln2 = c2 > 0 ? (uint16_t)(l2[c2 - 1].line + 2) : (uint16_t)(p2 == 0 ? 1 : 2);
        tail_n = total < MIN_DEINT ? total : MIN_DEINT;
        for (int i = lane; i < tail_n; i += 64) {
            const int pos = total - tail_n + i;
            SLine l;
            if (pos < c1) l = l1[pos];
            else if (pos < c1 + in_pad) l = sline_empty(fl->frame_number, (uint16_t)(ln1 + 2 * (pos - c1)));
            else if (pos < c1 + in_pad + c2) l = l2[pos - c1 - in_pad];
            else l = sline_empty(fl->frame_number, (uint16_t)(ln2 + 2 * (pos - c1 - in_pad - c2)));
            o->tail[i] = l;
        }
    }
    if (lane == 0) {
        o->f0 = f; o->last_pad_counter = steady ? lpc : (uint8_t)0xFF; o->broken_countdown = 0; o->tail_n = (uint16_t)tail_n;
        o->_pad[0] = o->_pad[1] = o->_pad[2] = 0;
    }
}

/* ---- sdv_k_stitch_compact: per-step slots -> contiguous streams ---------------------------------------------------- */
struct CompactArgs {
    const sdv_sample_pair *pairs; const sdv_frame_asm *frasm; const StepInfo *info;
    const uint64_t *pair_ofs; const uint32_t *frasm_ofs;      /* exclusive prefix sums over the steps (host) */
    sdv_sample_pair *out_pairs; sdv_frame_asm *out_frasm; uint32_t n_steps;
};
__device__ inline void compact_body(const CompactArgs &a, uint32_t k, int lane, int width)
{
    const StepInfo inf = a.info[k];
    const uint32_t *src = (const uint32_t *)(a.pairs + (size_t)k * PAIR_SLOT);
    uint32_t *dst = (uint32_t *)(a.out_pairs + a.pair_ofs[k]);
    for (uint32_t i = (uint32_t)lane; i < inf.n_pairs * 3u; i += (uint32_t)width) dst[i] = src[i];
    const uint32_t *fs = (const uint32_t *)(a.frasm + (size_t)k * FRASM_SLOT);
    uint32_t *fd = (uint32_t *)(a.out_frasm + a.frasm_ofs[k]);
    for (uint32_t i = (uint32_t)lane; i < inf.n_frasm * 16u; i += (uint32_t)width) fd[i] = fs[i];
}

/* ---- END_FRAME search: positions of the END_FRAME records, in stream order -------------------------------------- */
/* Pass 0 counts the END_FRAMEs per chunk and leaves every record's service type in a byte array; after the host's prefix sum
 * pass 1 writes the segment ends from those bytes (5 MB instead of the 235 MB of records of a 10 000-frame batch). */
struct LayoutArgs { uint32_t *ctl, *seg_end; uint32_t n_seg, n_carry, carry_frames, recs_per_frame; };     /* n_seg = frames of the call + the carried one */
struct SegArgs { RecSrc src; uint32_t n_recs; uint8_t *svc; uint32_t *block_count; const uint32_t *block_ofs; uint32_t *seg_end; int write; uint32_t seg_cap; /* 0 = no bound on the seg_end index */ };
enum { SEG_CHUNK = 1024 };
__device__ inline void seg_body(const SegArgs &a, uint32_t blk, int lane)
{
    const uint32_t lo = blk * SEG_CHUNK;
    uint32_t hi = lo + SEG_CHUNK; if (hi > a.n_recs) hi = a.n_recs;
    uint32_t cnt = 0;
    const uint32_t base = a.write ? a.block_ofs[blk] : 0u;
#pragma unroll 4
    for (uint32_t c = lo; c < hi; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        uint8_t srv = SDV_SRV_NO;
        if (i < hi) { if (a.write) srv = a.svc[i]; else { srv = a.src.at(i).service_type; a.svc[i] = srv; } }
        const bool ef = srv == SDV_SRV_END_FRAME;
        const uint64_t m = __ballot(ef);
        if (a.write && ef) {
            const uint32_t at = base + cnt + (uint32_t)__popcll(m & lanemask_lt(lane));
            if (a.seg_cap == 0 || at < a.seg_cap) a.seg_end[at] = i;
        }
        cnt += (uint32_t)__popcll(m);
    }
    if (!a.write && lane == 0) a.block_count[blk] = cnt;
}
/* The pipelined call (stitch_engine.inc, "a stream that plays"): the host does not wait for the counts - one wave turns them into offsets and leaves
 * the number of frame segments in ctl[CTL_NSEG]; the kernels behind read it from there, launched as wide as the host's estimate. */
enum { CTL_NSEG = 0, CTL_ABORT = 1, CTL_NEXT = 2 /* the turn kernel's work queue head */, CTL_WORDS = 3 };
struct ScanArgs { const uint32_t *block_count; uint32_t *block_ofs; uint32_t nblk; uint32_t *ctl; uint32_t *next_work; };
__device__ inline void seg_scan_body(const ScanArgs &a, int lane)
{
    /* every lane takes a contiguous run of the counts; one scan over the 64 run totals in between */
    const uint32_t per = (a.nblk + 63u) / 64u;
    uint32_t lo = (uint32_t)lane * per, hi = lo + per;
    if (lo > a.nblk) lo = a.nblk;
    if (hi > a.nblk) hi = a.nblk;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += a.block_count[i];
    uint32_t incl = sum;
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl((int)incl, lane >= d ? lane - d : lane); if (lane >= d) incl += o; }
    uint32_t run = incl - sum;
    for (uint32_t i = lo; i < hi; i++) { a.block_ofs[i] = run; run += a.block_count[i]; }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    if (lane == 0) { a.ctl[CTL_NSEG] = total; a.ctl[CTL_ABORT] = 0; *a.next_work = 0; }
}
/* how many segments / turns the kernels of a pipelined call work on: the counted ones, as far as the launch covers them */
__device__ inline uint32_t ctl_nseg(const uint32_t *ctl, uint32_t est) { const uint32_t n = ctl[CTL_NSEG]; return n < est ? n : est; }
} // namespace sdvs

__global__ void __launch_bounds__(64) sdv_k_stitch_segments(sdvs::SegArgs a) { sdvs::seg_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_stitch_seg_scan(sdvs::ScanArgs a) { sdvs::seg_scan_body(a, (int)threadIdx.x); }
/* records of known layout (the fused entry): the control words and the segment ends are arithmetic - made here, by a kernel in stream order (a copy from the
 * host between two kernels costs two hand-overs between the compute queue and the copy engine) */
__global__ void __launch_bounds__(64) sdv_k_stitch_layout(sdvs::LayoutArgs a)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i == 0) { a.ctl[sdvs::CTL_NSEG] = a.n_seg; a.ctl[sdvs::CTL_ABORT] = 0; a.ctl[sdvs::CTL_NEXT] = 0; }
    if (i < a.n_seg) a.seg_end[i] = a.carry_frames ? (i == 0 ? a.n_carry - 1u : a.n_carry + i * a.recs_per_frame - 1u) : a.n_carry + (i + 1u) * a.recs_per_frame - 1u;
}
#ifndef SDV_AN_WAVES
#define SDV_AN_WAVES 3
#endif
__global__ void __launch_bounds__(64, SDV_AN_WAVES) sdv_k_stitch_analyze(sdvs::AnalyzeArgs a)
{
    __shared__ uint32_t meta[sdvs::ANALYZE_LDS_WORDS];
    static_assert(sizeof(meta) >= 8 * sdvs::RES_PITCH * sizeof(uint16_t), "the staged field of the resolution trials lives in the staging area");
    /* The last frame of a call goes first: in the fused entry it is the one frame that is analysed from its records (the frame kernel writes every other frame
     * into the field buffers itself, AnalyzeArgs::direct), ten times the work of the others - started last it was the kernel's tail, 50 us on its own. */
    const uint32_t n_seg = a.ctl ? sdvs::ctl_nseg(a.ctl, a.n_seg) : a.n_seg;
    if (blockIdx.x >= n_seg) return;
    const uint32_t k = blockIdx.x == 0 ? n_seg - 1 : blockIdx.x - 1;
    const uint32_t n = a.seg_end[k] - (k == 0 ? 0u : a.seg_end[k - 1] + 1u);
    if (n <= sdvs::ANALYZE_LDS) sdvs::analyze_body<true>(a, k, (int)threadIdx.x, meta);
    else sdvs::analyze_body<false>(a, k, (int)threadIdx.x, meta);
    /* frames the turns must not be run on before the host has seen them (foreign lines, file boundaries): the kernels behind stand down */
    if (a.ctl && threadIdx.x == 0 && (a.brief[k].flags & (sdvs::FL_BAD_NUMBERS | sdvs::FL_END_FILE | sdvs::FL_NEW_FILE))) atomicOr(&a.ctl[sdvs::CTL_ABORT], 1u);
}
#ifndef SDV_ST_WAVES
#define SDV_ST_WAVES 4   /* 128 VGPRs + 100 B scratch: 4 096 resident waves share the turns; 1.73 ms per 10 000-frame call vs 1.83 (3 waves, no scratch) and 1.78 (5) */
#endif
__global__ void __launch_bounds__(64, SDV_ST_WAVES) sdv_k_stitch_step(sdvs::StepArgs a)
{
    __shared__ sdv_deint_line ring[sdvs::RING];
    __shared__ uint32_t pairbuf[64 * 9];
    uint32_t n_work = a.n_work;
    if (a.ctl) {
        const uint32_t ns = sdvs::ctl_nseg(a.ctl, a.est_seg);
        if (a.ctl[sdvs::CTL_ABORT] || ns < 2) return;
        n_work = ns - 1;
    }
    /* the turns are shared through a queue; a wave's first turn is its own number (the queue hands out what lies behind the launch width: 4 096 waves asking
     * one address for their first turn at the same moment stood in line for tens of microseconds) */
    for (bool first_turn = true;; first_turn = false) {
        uint32_t w = blockIdx.x;
        if (!first_turn) {
            if (threadIdx.x == 0) w = gridDim.x + atomicAdd(a.next_work, 1u);
            w = (uint32_t)__shfl((int)w, 0);
        }
        if (w >= n_work) break;
#if defined(SDV_ST_QUEUE_LDS) && !defined(SDV_EMU)
        __shared__ sdvs::SLine q_lds[sdvs::QCAP];          /* experiment: conv_queue of the turn in LDS (32 KB per wave) */
        sdvs::step_body(a, a.work ? a.work[w] : w, blockIdx.x, (int)threadIdx.x, ring, pairbuf, q_lds);
#else
        sdvs::step_body(a, a.work ? a.work[w] : w, blockIdx.x, (int)threadIdx.x, ring, pairbuf);
#endif
    }
}
__global__ void __launch_bounds__(64) sdv_k_stitch_predict(sdvs::PredictStArgs a)
{
    if (a.ctl) {        /* pipelined call: hand-overs for the turns 0 .. n_seg - 3 of the counted segments */
        const uint32_t ns = sdvs::ctl_nseg(a.ctl, a.est_seg);
        if (a.ctl[sdvs::CTL_ABORT] || ns < 2 || a.first + blockIdx.x >= ns - 2) return;
    }
    sdvs::predict_st_body(a, a.first + blockIdx.x, (int)threadIdx.x);
}
__global__ void __launch_bounds__(64) sdv_k_stitch_compact(sdvs::CompactArgs a) { sdvs::compact_body(a, blockIdx.x, (int)threadIdx.x, 64); }
#endif
