/*
 * pcm1_stitch_device.h - the PCM-1 back half on the device: PCM1DataStitcher::doFrameReassemble
 * (pcm1datastitcher.cpp:1578-1772) with PCM1Deinterleaver::processBlock (pcm1deinterleaver.cpp:69-278).
 *
 * PCM-1 has no error correction and the reference stitches every frame from scratch (resetState only drops flags that
 * findFrameTrim sets again), so frames are independent: one wave per frame, no speculation.  A frame is
 *   1. one sweep over its line records for what findFrameTrim (:202-568) collects - every item is a first/last/any/count
 *      over the records in stream order, i.e. a wave reduction;
 *   2. a second sweep that ranks the lines of either field between the trim marks (splitFrameToFields, :609-806) -
 *      ballot + prefix count; the line's record index goes to LDS, the 3 sub-lines per line are never materialised;
 *   3. paddings (findFramePadding, :809-923), then the output: the deinterleaver moves whole sub-lines (an L/R word pair)
 *      around, so every PCMSamplePair of a field is exactly one sub-line of the padded field
 *          pair p of block b  <-  sub-line  92*b + (b even == p odd ? 0 : 46) + p/2,
 *      735 per field (blocks 0-6: 92 pairs, block 7: 91), with the block's validity = all of its sub-lines valid.
 * Algorithmic bytes per frame: 32 B per line record in (490 lines: 15.7 KB) + 1470 x 12 B pairs + 52 B descriptor out.
 */
#ifndef SDV_PCM1_STITCH_DEVICE_H
#define SDV_PCM1_STITCH_DEVICE_H
#include "../../include/sdvpcm.h"
#include "stc007_stitch_device.h"

namespace sdvp1 {
using sdvs::lanemask_lt;
using sdvs::uni;

enum { LINES_PF = 245, SUBLINES_PF = 735, MIN_GOOD = 245 * 4 / 5, BUF_TRIM = 3 * 640, BIT_RANGE = 1 << 12, BIT_SIGN = 1 << 11, WORD_MASK = (1 << 13) - 1 };
enum { ORDER_TFF = 1, ORDER_BFF = 2 };
/* per-frame marks found by the flags pass */
enum { FF_NEW_FILE = 1, FF_END_FILE = 2, FF_FOREIGN = 4 };
/* reasons a frame cannot be stitched statelessly (the reference would read sub-lines left over from earlier frames, or hold lines back) */
enum { FE_FOREIGN = 1, FE_TOO_LONG = 2, FE_STALE = 4, FE_SHORT_QUEUE = 8 };

struct RecSrc1 {
    const sdv_pcm1_line_rec *carry; uint32_t n_carry; const sdv_pcm1_line_rec *recs;
    __device__ inline const sdv_pcm1_line_rec &at(uint32_t i) const { return i < n_carry ? carry[i] : recs[i - n_carry]; }
};
struct Cfg1 { uint8_t field_order, auto_offset, ignore_crc; int8_t odd_offset, even_offset; };

/* a service line is a cleared PCM1Line that keeps frame and line number (PCMLine::setServiceLine, pcmline.cpp:490-502) */
__device__ inline bool r_service(const sdv_pcm1_line_rec &r) { return r.service_type != SDV_SRV_NO; }
__device__ inline bool r_crc_if(const sdv_pcm1_line_rec &r) { return !r_service(r) && r.calc_crc == r.words[6]; }   /* isCRCValidIgnoreForced */
__device__ inline bool r_crc(const sdv_pcm1_line_rec &r) { return !(r.flags & SDV_LF_FORCED_BAD) && r_crc_if(r); }    /* isCRCValid, pcmline.cpp:360-367 */
__device__ inline bool r_bw(const sdv_pcm1_line_rec &r) { return !r_service(r) && (r.flags & SDV_LF_BW_SET) != 0; }

/* ---- segments: positions of the END_FRAME records (same two-pass scheme as the STC-007 stitch stage) ------------- */
struct SegArgs1 { RecSrc1 src; uint32_t n_recs; uint32_t *block_count; const uint32_t *block_ofs; uint32_t *seg_end; int write; };
enum { SEG_CHUNK1 = 4096 };
__device__ inline void seg_body(const SegArgs1 &a, uint32_t blk, int lane)
{
    const uint32_t lo = blk * SEG_CHUNK1;
    uint32_t hi = lo + SEG_CHUNK1; if (hi > a.n_recs) hi = a.n_recs;
    uint32_t cnt = 0;
    for (uint32_t c = lo; c < hi; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        const bool ef = i < hi && a.src.at(i).service_type == SDV_SRV_END_FRAME;
        const uint64_t m = __ballot(ef);
        if (a.write && ef) a.seg_end[a.block_ofs[blk] + cnt + (uint32_t)__popcll(m & lanemask_lt(lane))] = i;
        cnt += (uint32_t)__popcll(m);
    }
    if (!a.write && lane == 0) a.block_count[blk] = cnt;
}

/* ---- file marks per frame: a NEW_FILE / END_FILE record belongs to the frame whose END_FRAME follows it ------------ */
struct MarkArgs1 { RecSrc1 src; uint32_t n_recs; const uint32_t *seg_end; uint32_t n_seg; uint32_t *marks; uint32_t *stat; };
__device__ inline void mark_body(const MarkArgs1 &a, uint32_t i)
{
    if (i >= a.n_recs) return;
    const sdv_pcm1_line_rec &r = a.src.at(i);
    const uint8_t srv = r.service_type;
    uint32_t lo = 0, hi = a.n_seg;                     /* first segment whose END_FRAME index is >= i */
    if (srv != SDV_SRV_NEW_FILE && srv != SDV_SRV_END_FILE && srv != SDV_SRV_END_FRAME) {
        /* ordinary lines only need the check for lines of a later frame queued ahead of this frame's END_FRAME: rare, so look at the
         * neighbour first and search only when the frame number changes inside a segment */
        if (i + 1 >= a.n_recs) return;
        const sdv_pcm1_line_rec &nx = a.src.at(i + 1);
        if (nx.frame_number >= r.frame_number) return;
    }
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (a.seg_end[mid] < i) lo = mid + 1; else hi = mid; }
    if (lo >= a.n_seg) return;                          /* waits for its END_FRAME in the carry */
    const uint32_t frame = a.src.at(a.seg_end[lo]).frame_number;
    uint32_t m = 0;
    if (r.frame_number == frame) { if (srv == SDV_SRV_NEW_FILE) m = FF_NEW_FILE; else if (srv == SDV_SRV_END_FILE) m = FF_END_FILE; }
    else if (r.frame_number > frame) m = FF_FOREIGN;    /* the reference would keep this line queued for a later turn */
    if (m) { atomicOr(&a.marks[lo], m); atomicAdd(&a.stat[2], 1u); }
}

/* ---- output offsets: exclusive scan of the per-frame output counts, one workgroup --------------------------------- */
struct ScanArgs1 { const uint32_t *marks; uint32_t n_seg; uint64_t *pair_ofs; uint32_t *frasm_ofs; const uint32_t *stat; };   /* n_seg + 1 entries each */
__device__ inline void frame_counts(uint32_t marks, uint32_t &pairs, uint32_t &frasm)
{
    if (marks & FF_END_FILE) { pairs = 1; frasm = 1; }                         /* outputFileStop only (:1723-1729) */
    else { pairs = 2 * SUBLINES_PF + ((marks & FF_NEW_FILE) ? 1 : 0); frasm = 1 + ((marks & FF_NEW_FILE) ? 1 : 0); }
}
/* Most batches hold no file marks at all: then frame k simply starts at pair 1470 k, descriptor k, and only the totals are written
 * (stat[2] counts the marked frames).  Otherwise every lane sums a contiguous run of frames, the wave scans the 64 sums, and the
 * lanes write their runs. */
__device__ inline void scan_body(const ScanArgs1 &a, int lane)
{
    if (a.stat[2] == 0) {
        if (lane == 0) { a.pair_ofs[a.n_seg] = (uint64_t)a.n_seg * (2 * SUBLINES_PF); a.frasm_ofs[a.n_seg] = a.n_seg; }
        return;
    }
    const uint32_t run = (a.n_seg + 63) / 64, k0 = (uint32_t)lane * run;
    uint32_t k1 = k0 + run; if (k1 > a.n_seg) k1 = a.n_seg;
    uint64_t psum = 0; uint32_t fsum = 0;
    for (uint32_t k = k0; k < k1; k++) { uint32_t p, f; frame_counts(a.marks[k], p, f); psum += p; fsum += f; }
    uint64_t ps = psum; uint32_t fs = fsum;                     /* inclusive wave scan */
    for (int d = 1; d < 64; d <<= 1) {
        const int src = lane >= d ? lane - d : lane;
        const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)ps, src), hi = (uint32_t)__shfl((int)(uint32_t)(ps >> 32), src), of = (uint32_t)__shfl((int)fs, src);
        if (lane >= d) { ps += ((uint64_t)hi << 32) | lo; fs += of; }
    }
    uint64_t pb = ps - psum; uint32_t fb = fs - fsum;
    for (uint32_t k = k0; k < k1; k++) { uint32_t p, f; frame_counts(a.marks[k], p, f); a.pair_ofs[k] = pb; a.frasm_ofs[k] = fb; pb += p; fb += f; }
    if (lane == 63) { a.pair_ofs[a.n_seg] = ps; a.frasm_ofs[a.n_seg] = fs; }
}

/* ---- the frame ---------------------------------------------------------------------------------------------------- */
struct FrameArgs1 {
    RecSrc1 src; const uint32_t *seg_end; uint32_t n_seg; Cfg1 cfg;
    const uint32_t *marks; const uint64_t *pair_ofs; const uint32_t *frasm_ofs;
    sdv_sample_pair *out_pairs; uint64_t pairs_cap; sdv_frame_asm_pcm1 *out_frames; uint32_t frames_cap;
    uint32_t *stat;             /* [0] = OR of FE_*, [1] = first frame index with an error, [2] = marks set (0: the plain layout) */
};

__device__ inline uint32_t wmin(uint32_t v, int lane) { for (int d = 1; d < 64; d <<= 1) { uint32_t o = (uint32_t)__shfl((int)v, lane ^ d); v = o < v ? o : v; } return v; }
__device__ inline uint32_t wmax(uint32_t v, int lane) { for (int d = 1; d < 64; d <<= 1) { uint32_t o = (uint32_t)__shfl((int)v, lane ^ d); v = o > v ? o : v; } return v; }
__device__ inline uint32_t wsum(uint32_t v, int lane) { for (int d = 1; d < 64; d <<= 1) v += (uint32_t)__shfl((int)v, lane ^ d); return v; }

__device__ inline void frasm1_clear(sdv_frame_asm_pcm1 &f)     /* FrameAsmPCM1::clear, frametrimset.cpp:455-464, 727-744 */
{
    f = sdv_frame_asm_pcm1();
    f.odd_bottom_data = f.even_bottom_data = 0xFFFF;
}
__device__ inline void service_pair(sdv_sample_pair &p, uint8_t srv)
{
    p.audio_word[0] = p.audio_word[1] = 0; p.sample_flags[0] = p.sample_flags[1] = 0; p.sample_rate = 44056; p.emphasis = 0; p.service_type = srv; p._pad = 0;
}
/* PCM1DataBlock::getSample (pcm1datablock.cpp:309-348) */
__device__ inline int16_t p1_sample(uint16_t w)
{
    if ((w & BIT_RANGE) == 0) return (int16_t)(uint16_t)(w << 4);
    const bool pos = (w & BIT_SIGN) == 0;
    w = (uint16_t)((w & ~BIT_RANGE) << 2);
    if (!pos) w |= (1 << 15) | (1 << 14);
    return (int16_t)w;
}

#define P1_NONE 0xFFFFFFFFu

/* one wave, one frame.  lds: 2 x 245 record indices (+1: relative to the segment start) */
__device__ inline void frame_body(const FrameArgs1 &a, uint32_t k, int lane, uint16_t (*field_idx)[LINES_PF + 3])
{
    const uint32_t lo = k == 0 ? 0u : a.seg_end[k - 1] + 1u, hi = a.seg_end[k];
    const uint32_t frame = a.src.at(hi).frame_number;
    const bool plain = a.stat[2] == 0;
    const uint32_t marks = plain ? 0u : a.marks[k];
    const uint64_t pofs = plain ? (uint64_t)k * (2 * SUBLINES_PF) : a.pair_ofs[k];
    const uint32_t fofs = plain ? k : a.frasm_ofs[k];
    const Cfg1 cfg = a.cfg;
    uint32_t err = (marks & FF_FOREIGN) ? FE_FOREIGN : 0;
    const uint32_t n = hi - lo;
    if (n > BUF_TRIM) err |= FE_TOO_LONG;

    if (marks & FF_END_FILE) {                      /* the frame that carries the END_FILE tag only closes the file (:1721-1729) */
        if (lane == 0) {
            if (fofs < a.frames_cap) { sdv_frame_asm_pcm1 d; frasm1_clear(d); d.service_type = SDV_PAIR_SRV_END_FILE; a.out_frames[fofs] = d; }
            if (pofs < a.pairs_cap) { sdv_sample_pair p; service_pair(p, SDV_PAIR_SRV_END_FILE); a.out_pairs[pofs] = p; }
            if (err) { atomicOr(&a.stat[0], err); atomicMin(&a.stat[1], k); }
        }
        return;
    }

    /* 1. findFrameTrim: per parity (index 0 = odd lines, 1 = even lines) */
    uint32_t good[2] = { 0, 0 }, first_valid[2] = { P1_NONE, P1_NONE }, last_valid[2] = { 0, 0 }, first_hdr[2] = { P1_NONE, P1_NONE }, last_hdr[2] = { 0, 0 };
    uint32_t first_bw[2] = { P1_NONE, P1_NONE }, last_bw[2] = { 0, 0 }, first_ci[2] = { P1_NONE, P1_NONE }, last_ci[2] = { 0, 0 };   /* last_*: index + 1, 0 = none */
    for (uint32_t c = 0; c < n; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        if (i < n) {
            const sdv_pcm1_line_rec &r = a.src.at(lo + i);
            if (r.frame_number == frame) {
                const int par = (r.line_number & 1) ? 0 : 1;
                const bool v = !r_service(r) && r_crc(r), hd = r.service_type == SDV_SRV_HEADER_LINE, bw = r_bw(r), ci = r_crc_if(r);
#pragma unroll
                for (int p = 0; p < 2; p++) {       /* compile-time indices: the counters stay in registers */
                    const bool m = par == p;
                    if (m && v) { good[p]++; if (first_valid[p] == P1_NONE) first_valid[p] = i; last_valid[p] = i + 1; }
                    if (m && hd) { if (first_hdr[p] == P1_NONE) first_hdr[p] = i; last_hdr[p] = i + 1; }
                    if (m && bw) { if (first_bw[p] == P1_NONE) first_bw[p] = i; last_bw[p] = i + 1; }
                    if (m && ci) { if (first_ci[p] == P1_NONE) first_ci[p] = i; last_ci[p] = i + 1; }
                }
            }
        }
    }
    bool header_present = false, emphasis_set = false;
    uint32_t top[2], bottom[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        good[p] = wsum(good[p], lane);
        first_valid[p] = wmin(first_valid[p], lane); last_valid[p] = wmax(last_valid[p], lane);
        first_hdr[p] = wmin(first_hdr[p], lane); last_hdr[p] = wmax(last_hdr[p], lane);
        /* a header line counts while no valid data line of its field has been seen: from the top for "header present" (:262-285),
         * from the bottom for the emphasis flag (:306-352; the backward scan stops at the second field's last valid line, which is
         * never past the own field's) */
        if (first_hdr[p] != P1_NONE && first_hdr[p] < first_valid[p]) header_present = true;
        if (last_hdr[p] > last_valid[p]) emphasis_set = true;
        const bool skip_bad = good[p] > MIN_GOOD;
        const uint32_t fi = wmin(skip_bad ? first_ci[p] : first_bw[p], lane), la = wmax(skip_bad ? last_ci[p] : last_bw[p], lane);
        top[p] = fi != P1_NONE ? a.src.at(lo + fi).line_number : 0u;
        bottom[p] = la != 0 ? a.src.at(lo + la - 1).line_number : 0u;
    }
    if (!cfg.auto_offset) {                          /* :388-412 */
        top[0] = cfg.odd_offset > 0 ? (uint32_t)(2 * cfg.odd_offset + 1) : 1u;
        top[1] = cfg.even_offset > 0 ? (uint32_t)(2 * cfg.even_offset + 2) : 2u;
    }
    if (marks & FF_NEW_FILE) header_present = emphasis_set = false;            /* resetState after the trim search (:1688-1692, :63-76) */

    /* 2. splitFrameToFields: rank the lines of either field, 245 at most */
    uint32_t cnt[2] = { 0, 0 }, valid[2] = { 0, 0 }, ref_ok[2] = { 0, 0 }, ref_all[2] = { 0, 0 };
    const bool even_open = top[1] != bottom[1] || top[1] != 0;
    for (uint32_t c = 0; c < n; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        bool in[2] = { false, false };
        bool ok = false, filler = false; uint32_t ref = 0;
        if (i < n) {
            const sdv_pcm1_line_rec &r = a.src.at(lo + i);
            filler = r.service_type == SDV_SRV_FILLER;
            if (r.frame_number == frame && (!r_service(r) || filler)) {
                const uint32_t ln = r.line_number;
                in[0] = (ln & 1) != 0 && ln >= top[0] && ln <= bottom[0];
                in[1] = (ln & 1) == 0 && ln >= top[1] && ln <= bottom[1] && even_open;
                ok = r_crc(r); ref = filler ? 0u : r.ref_level;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint64_t m = __ballot(in[p]);
            const uint32_t rank = cnt[p] + (uint32_t)__popcll(m & lanemask_lt(lane));
            if (in[p] && rank < LINES_PF) {
                field_idx[p][rank] = (uint16_t)(i + 1);
                ref_all[p] += ref;
                if (ok) { valid[p]++; ref_ok[p] += ref; }
            }
            cnt[p] += (uint32_t)__popcll(m);
        }
    }
    sdv_frame_asm_pcm1 f; frasm1_clear(f);
    f.frame_number = frame;
    uint32_t data[2], ref_level[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if (cnt[p] > LINES_PF) cnt[p] = LINES_PF;
        valid[p] = wsum(valid[p], lane); ref_ok[p] = wsum(ref_ok[p], lane); ref_all[p] = wsum(ref_all[p], lane);
        data[p] = 3 * cnt[p];
        /* the reference sums the level once per sub-line and divides by the sub-line count (:767-798) */
        ref_level[p] = valid[p] > 0 ? ((3 * ref_ok[p]) / (3 * valid[p])) & 0xFF : (cnt[p] > 0 ? ((3 * ref_all[p]) / (3 * cnt[p])) & 0xFF : 0u);
    }

    /* 3. findFramePadding (:809-923); uint16_t arithmetic as the reference's fields */
    uint16_t top_pad[2], bot_pad[2];
    if (cfg.auto_offset) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint16_t pad = (uint16_t)((SUBLINES_PF - data[p]) / 3);
            top_pad[p] = header_present ? 0 : pad; bot_pad[p] = header_present ? pad : 0;
        }
    } else {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const int ofs = p == 0 ? cfg.odd_offset : cfg.even_offset;
            top_pad[p] = ofs > 0 ? 0 : (uint16_t)(0 - ofs);
            uint16_t bp = (uint16_t)(((int)bottom[p] - (int)top[p]) / 2 + 1);
            bp = (uint16_t)(bp + top_pad[p]);
            if (bp > LINES_PF) {
                bp = (uint16_t)(bp - LINES_PF);
                bottom[p] = (uint16_t)(bottom[p] - bp * 2);
                uint16_t dl = (uint16_t)(((int)bottom[p] - (int)top[p]) / 2 + 1);
                dl = (uint16_t)(dl * 3);
                if (dl > data[p]) err |= FE_STALE;   /* the reference would output sub-lines left in its field buffer by earlier frames */
                data[p] = dl;
            }
            bot_pad[p] = (uint16_t)((SUBLINES_PF - (int)data[p]) / 3 - top_pad[p]);
        }
    }
    const uint8_t order = cfg.field_order == ORDER_BFF ? ORDER_BFF : ORDER_TFF;

    /* 4. the two fields in output order (:1076-1218, :1382-1453) */
    uint64_t po = pofs;
    if (marks & FF_NEW_FILE) {
        if (lane == 0) {
            if (fofs < a.frames_cap) { sdv_frame_asm_pcm1 d; frasm1_clear(d); d.service_type = SDV_PAIR_SRV_NEW_FILE; a.out_frames[fofs] = d; }
            if (po < a.pairs_cap) { sdv_sample_pair p; service_pair(p, SDV_PAIR_SRV_NEW_FILE); a.out_pairs[po] = p; }
        }
        po++;
    }
    uint32_t blocks_drop = 0, samples_drop = 0, blocks_fix_bp = 0;
    for (int fld = 0; fld < 2; fld++) {
        const bool odd_field = (order == ORDER_TFF) == (fld == 0);
        const uint16_t *fidx = odd_field ? field_idx[0] : field_idx[1];
        const uint32_t f_top = odd_field ? top_pad[0] : top_pad[1], f_bot = odd_field ? bot_pad[0] : bot_pad[1], f_data = odd_field ? data[0] : data[1];
        const uint32_t q_top = 3u * f_top, q_data = f_data <= SUBLINES_PF ? f_data : 0u /* addLinesFromField refuses (:960) */;
        if ((uint64_t)q_top + q_data + 3ull * f_bot < SUBLINES_PF) err |= FE_SHORT_QUEUE;   /* DI_RET_NO_DATA: 8 cleared blocks of 92 pairs */
        for (uint32_t blk = 0; blk < 8; blk++) {
            const uint32_t len = blk == 7 ? 91u : 92u;         /* the last block is short (PCM1DataBlock::setShortLength) */
            sdv_sample_pair sp[2]; uint32_t dst[2]; bool act[2];
            uint32_t bad = 0; bool any_picked = false;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t t = (uint32_t)(h * 64 + lane);       /* sub-line of the block: [0,46) first stripe, [46,92) second */
                act[h] = t < len;
                const uint32_t sub = blk * 92 + t;
                uint16_t wl = BIT_RANGE, wr = BIT_RANGE; bool ok = false, picked = false;
                if (act[h] && sub >= q_top && sub < q_top + q_data) {
                    const uint32_t j = sub - q_top, line = j / 3, part = j - 3 * line;
                    const sdv_pcm1_line_rec &r = a.src.at(lo + fidx[line] - 1);
                    if (!r_service(r)) {            /* a filler line is a cleared line: silent, invalid */
                        wl = (uint16_t)(r.words[2 * part] & WORD_MASK); wr = (uint16_t)(r.words[2 * part + 1] & WORD_MASK);
                        ok = cfg.ignore_crc ? r_bw(r) : r_crc(r);
                        picked = (part == 0 && r.picked_bits_left > 0) || r.picked_bits_right > 0;
                    }
                }
                bad += (uint32_t)__popcll(__ballot(act[h] && !ok));
                any_picked = any_picked || __ballot(act[h] && picked) != 0;
                /* setWordData (:150-278): the stripe that starts at word 2 (odd pairs) reads the first 46 sub-lines in even blocks and
                 * the second 46 in odd blocks; the stripe that starts at word 0 the others */
                const bool first = t < 46;
                const uint32_t wp = first ? t : t - 46;
                const bool odd_pair = ((blk & 1) == 0) == first;
                dst[h] = blk * 92 + 2 * wp + (odd_pair ? 1u : 0u);
                sp[h].audio_word[0] = p1_sample(wl); sp[h].audio_word[1] = p1_sample(wr);
                sp[h].sample_flags[0] = sp[h].sample_flags[1] = (uint8_t)(ok ? SDV_SF_WORD_VALID : 0);
                sp[h].sample_rate = 44100; sp[h].emphasis = emphasis_set ? 1 : 0; sp[h].service_type = 0; sp[h]._pad = 0;
            }
            const bool valid_blk = bad == 0;
            if (!valid_blk) { blocks_drop++; samples_drop += (2 * bad) & 0xFF; }
            else if (any_picked) blocks_fix_bp++;
#pragma unroll
            for (int h = 0; h < 2; h++)
                if (act[h]) {
                    if (valid_blk) { sp[h].sample_flags[0] |= SDV_SF_BLOCK_OK; sp[h].sample_flags[1] |= SDV_SF_BLOCK_OK; }
                    if (po + dst[h] < a.pairs_cap) a.out_pairs[po + dst[h]] = sp[h];
                }
        }
        po += SUBLINES_PF;
    }
    if (lane == 0) {
        f.odd_std_lines = f.even_std_lines = LINES_PF;
        f.odd_data_lines = (uint16_t)(data[0] / 3); f.even_data_lines = (uint16_t)(data[1] / 3);
        f.odd_valid_lines = (uint16_t)valid[0]; f.even_valid_lines = (uint16_t)valid[1];
        f.odd_top_data = (uint16_t)top[0]; f.odd_bottom_data = (uint16_t)bottom[0]; f.even_top_data = (uint16_t)top[1]; f.even_bottom_data = (uint16_t)bottom[1];
        f.odd_sample_rate = f.even_sample_rate = 44100;
        f.blocks_total = 16; f.blocks_drop = (uint16_t)blocks_drop; f.samples_drop = (uint16_t)samples_drop; f.blocks_fix_bp = (uint16_t)blocks_fix_bp;
        f.odd_top_padding = top_pad[0]; f.odd_bottom_padding = bot_pad[0]; f.even_top_padding = top_pad[1]; f.even_bottom_padding = bot_pad[1];
        f.field_order = order; f.odd_ref = (uint8_t)ref_level[0]; f.even_ref = (uint8_t)ref_level[1];
        f.flags = (uint8_t)(SDV_FA_ORDER_PRESET | (emphasis_set ? (SDV_FA1_ODD_EMPHASIS | SDV_FA1_EVEN_EMPHASIS) : 0));
        const uint32_t fo = fofs + ((marks & FF_NEW_FILE) ? 1u : 0u);
        if (fo < a.frames_cap) a.out_frames[fo] = f;
        if (err) { atomicOr(&a.stat[0], err); atomicMin(&a.stat[1], k); }
    }
}
} // namespace sdvp1

__global__ void __launch_bounds__(64) sdv_k_pcm1_segments(sdvp1::SegArgs1 a) { sdvp1::seg_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm1_marks(sdvp1::MarkArgs1 a) { sdvp1::mark_body(a, blockIdx.x * 64u + threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm1_scan(sdvp1::ScanArgs1 a) { sdvp1::scan_body(a, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm1_frames(sdvp1::FrameArgs1 a)
{
    __shared__ uint16_t field_idx[2][sdvp1::LINES_PF + 3];
    sdvp1::frame_body(a, blockIdx.x, (int)threadIdx.x, field_idx);
}
#endif
