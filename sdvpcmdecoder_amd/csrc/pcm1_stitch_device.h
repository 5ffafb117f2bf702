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
enum { FE_FOREIGN = 1, FE_TOO_LONG = 2, FE_MARKS = 16 };

struct RecSrc1 {
    const sdv_pcm1_line_rec *carry; uint32_t n_carry; const sdv_pcm1_line_rec *recs;
    __device__ inline const sdv_pcm1_line_rec &at(uint32_t i) const { return i < n_carry ? carry[i] : recs[i - n_carry]; }
};
struct Cfg1 { uint8_t field_order, auto_offset, ignore_crc; int8_t odd_offset, even_offset; };

/* ---- segments: positions of the END_FRAME records, and the file tags of every frame ------------------------------- */
/* Pass 0 counts the END_FRAMEs of every 4096-record chunk and leaves each record's service type in a byte array (5 MB for a
 * 10 000-frame batch instead of the 157 MB of records); after the host's prefix sum, pass 1 works from those bytes: it writes
 * the segment ends and tags segment k (= the number of END_FRAMEs ahead of a record) with the NEW_FILE / END_FILE records in it.
 * The frame kernel, which sees the frame numbers, confirms the tags. */
struct SegArgs1 { RecSrc1 src; uint32_t n_recs; uint8_t *svc; uint32_t *block_count; const uint32_t *block_ofs; uint32_t *seg_end; uint32_t n_seg; uint32_t *marks; uint32_t *stat; int write; };
#ifndef SDV_P1_SEG_CHUNK
#define SDV_P1_SEG_CHUNK 1024
#endif
enum { SEG_CHUNK1 = SDV_P1_SEG_CHUNK };
__device__ inline void seg_body(const SegArgs1 &a, uint32_t blk, int lane)
{
    const uint32_t lo = blk * SEG_CHUNK1;
    uint32_t hi = lo + SEG_CHUNK1; if (hi > a.n_recs) hi = a.n_recs;
    uint32_t cnt = 0;
    const uint32_t base = a.write ? a.block_ofs[blk] : 0u;
#pragma unroll 4
    for (uint32_t c = lo; c < hi; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        uint8_t srv = SDV_SRV_NO;
        if (i < hi) { if (a.write) srv = a.svc[i]; else { srv = a.src.at(i).service_type; a.svc[i] = srv; } }
        const uint64_t m = __ballot(srv == SDV_SRV_END_FRAME);
        if (a.write) {
            const uint32_t seg = base + cnt + (uint32_t)__popcll(m & lanemask_lt(lane));
            if (srv == SDV_SRV_END_FRAME) a.seg_end[seg] = i;
            else if ((srv == SDV_SRV_NEW_FILE || srv == SDV_SRV_END_FILE) && seg < a.n_seg) {
                atomicOr(&a.marks[seg], srv == SDV_SRV_NEW_FILE ? (uint32_t)FF_NEW_FILE : (uint32_t)FF_END_FILE);
                atomicAdd(&a.stat[2], 1u);
            }
        }
        cnt += (uint32_t)__popcll(m);
    }
    if (!a.write && lane == 0) a.block_count[blk] = cnt;
}

/* ---- output offsets: exclusive scan of the per-frame output counts, one workgroup --------------------------------- */
struct ScanArgs1 { uint32_t *marks; uint32_t n_seg; uint64_t *pair_ofs; uint32_t *frasm_ofs; const uint32_t *stat; RecSrc1 src; const uint32_t *seg_end; };   /* n_seg + 1 entries each */
/* how many records of the segment [lo, lo + n) fillUntilFullFrame keeps: the lines that carry the frame's number, BUF_SIZE_TRIM of them at most (:150-196) -
 * the index behind the last one kept.  One lane. */
__device__ inline uint32_t kept_records(const RecSrc1 &src, uint32_t lo, uint32_t n, uint32_t frame)
{
    if (n <= (uint32_t)BUF_TRIM) return n;
    uint32_t kept = 0;
    for (uint32_t i = 0; i < n; i++) if (src.at(lo + i).frame_number == frame) { if (++kept == (uint32_t)BUF_TRIM) return i + 1; }
    return n;
}
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
    /* The segments pass marked every frame in whose records a file tag lies.  What the reference's frame sees is less: the tags that carry the
     * frame's own number, among the lines it keeps (a tag with an older number is popped with the frame and never looked at, :1617-1637) -
     * the few marked frames are read once more for that */
    for (uint32_t k = k0; k < k1; k++)
        if (a.marks[k]) {
            const uint32_t lo = k == 0 ? 0u : a.seg_end[k - 1] + 1u, n = a.seg_end[k] - lo, frame = a.src.at(lo + n).frame_number;
            const uint32_t keep = kept_records(a.src, lo, n, frame);
            uint32_t m = 0;
            for (uint32_t i = 0; i < keep; i++) {
                const sdv_pcm1_line_rec &r = a.src.at(lo + i);
                if (r.frame_number == frame) m |= r.service_type == SDV_SRV_NEW_FILE ? (uint32_t)FF_NEW_FILE : (r.service_type == SDV_SRV_END_FILE ? (uint32_t)FF_END_FILE : 0u);
            }
            a.marks[k] = m;
        }
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
    unsigned long long *timing; /* developer aid (SDV_STITCH_TIMING): 8 cycle stamps per frame */
    uint32_t *stat;             /* [0] = OR of FE_*, [1] = first frame index with an error, [2] = marks set (0: the plain layout) */
    /* The reference's field buffers outlive a frame (frame1_odd / frame1_even, pcm1datastitcher.h): with manual line offsets a field can be told
     * to hold more lines than the frame delivered, and then puts out what earlier frames left at those places (:896-909 -> addLinesFromField).
     * Manual mode therefore runs the kernel twice: the first pass (cnt_out set) stops behind the field split and leaves every frame's line counts
     * and the records its fields were made of; the second pass (cnt_in set) looks the missing places up - the last earlier frame whose field
     * reached that far, or `hist`, the buffers as the calls before left them. */
    uint32_t *cnt_out, *kept_out;               /* per frame: odd | even << 16 lines split; per frame 2 x 245 record indices */
    const uint32_t *cnt_in, *kept_in; const void *hist;     /* hist: 2 x 245 Line16 */
    /* the visualiser's feeds (sdv_set_pcm1_stitch_block_output / _line_output): 16 blocks and 1470 sub-lines per frame, or NULL */
    sdv_pcm1_block_rec *out_blocks; uint64_t blocks_cap; sdv_pcm1_asm_line_rec *out_asm; uint64_t asm_cap;
};


__device__ inline void frasm1_clear(sdv_frame_asm_pcm1 &f)     /* FrameAsmPCM1::clear, frametrimset.cpp:455-464, 727-744 */
{
    f = sdv_frame_asm_pcm1();
    f.odd_bottom_data = f.even_bottom_data = 0xFFFF;
}
/* a PCMSamplePair as its three 32-bit words (layout of sdv_sample_pair): built in registers, stored as one 12-byte write */
struct Pair3 { uint32_t a, b, c; };
static_assert(sizeof(sdv_sample_pair) == 12 && sizeof(Pair3) == 12, "pair layout");
__device__ inline Pair3 make_pair3(int16_t l, int16_t r, uint32_t flags, uint32_t rate, bool emphasis, uint32_t srv)
{
    Pair3 p;
    p.a = (uint32_t)(uint16_t)l | ((uint32_t)(uint16_t)r << 16);
    p.b = flags | (flags << 8) | (rate << 16);
    p.c = (emphasis ? 1u : 0u) | (srv << 8);
    return p;
}
#ifndef SDV_PAIR_NT_STORES
#define SDV_PAIR_NT_STORES 1
#endif
/* the sample pairs are written once and read by a later stage (or by the host): streaming stores in the product build */
__device__ inline void store_pair(sdv_sample_pair *dst, const Pair3 &p)
{
#if defined(SDV_EMU) || !SDV_PAIR_NT_STORES
    *(Pair3 *)dst = p;
#else
    uint32_t *d = (uint32_t *)dst;
    __builtin_nontemporal_store(p.a, d); __builtin_nontemporal_store(p.b, d + 1); __builtin_nontemporal_store(p.c, d + 2);
#endif
}
__device__ inline void service_pair(sdv_sample_pair *dst, uint8_t srv) { store_pair(dst, make_pair3(0, 0, 0, 44056, false, srv)); }
/* PCM1DataBlock::getSample (pcm1datablock.cpp:309-348) */
__device__ inline int16_t p1_sample(uint16_t w)
{
    /* fine range (range bit clear): the 12 bits times 16; coarse range: the 12 bits sign-extended, times 4 - i.e. the same
     * left-aligned value shifted back arithmetically by 2 */
    const int16_t x = (int16_t)(uint16_t)(w << 4);
    return (int16_t)(x >> ((w >> 11) & 2));
}

#define P1_NONE 0xFFFFFFFFu
#ifdef SDV_EMU
#define P1_STAMP(i) ((void)0)
#else
#define P1_STAMP(i) do { if (a.timing && lane == 0) a.timing[(size_t)k * 8 + (i)] = (unsigned long long)__builtin_readcyclecounter(); } while (0)
#endif

/* What the frame needs of one line record, 16 bytes: the frame's records are read from HBM once and kept in LDS in this form. */
struct Line16 { uint16_t w[6]; uint16_t line; uint8_t fl; uint8_t ref; };
enum { LF_VALID = 1,        /* isCRCValid() of a data line of this frame */
       LF_BW = 2,           /* hasBWSet() */
       LF_CI = 4,           /* isCRCValidIgnoreForced() */
       LF_DATA = 8,         /* takes part in splitFrameToFields: a data line or a filler of this frame */
       LF_HDR = 16,         /* isServHeader() */
       LF_OK = 32,          /* what the deinterleaver takes for "valid": B/W levels when CRCs are ignored, the CRC otherwise */
       LF_PICK = 64,        /* picked_bits_right > 0 */
       LF_PICKL = 128 };    /* picked_bits_left > 0 */
#ifndef SDV_P1_LDS_LINES
#define SDV_P1_LDS_LINES 544
#endif
enum { LDS_LINES = SDV_P1_LDS_LINES };   /* a 525-line frame's 490 PCM lines + service tags with room to spare (16 frames per CU); longer segments are read from
                                          * global memory in every sweep */

/* an sdv_pcm1_line_rec as the two 16-byte loads it is read with:
 *   a = { frame_number, line_number | words[0] << 16, words[1] | words[2] << 16, words[3] | words[4] << 16 }
 *   b = { words[5] | words[6] << 16, calc_crc | ref_level << 16 | picked_bits_left << 24, picked_bits_right | service_type << 8 | flags << 16, pad } */
struct Raw32 { uint4 a, b; };
static_assert(sizeof(sdv_pcm1_line_rec) == 32, "record layout");
/* words 2 part, 2 part + 1 of a line, by masks: as `l.w[2 * part]`, or as a choice between members by `part`, the compiler reads the line through a computed
 * address - and a line that is addressed like that lives in scratch memory (the visualiser's build of the frame kernel: 256 registers and 328 bytes of it) */
__device__ __forceinline__ uint32_t line_word_pair(const Line16 &l, uint32_t part)
{
    const uint32_t p0 = (uint32_t)l.w[0] | ((uint32_t)l.w[1] << 16), p1 = (uint32_t)l.w[2] | ((uint32_t)l.w[3] << 16), p2 = (uint32_t)l.w[4] | ((uint32_t)l.w[5] << 16);
    const uint32_t m0 = part == 0 ? ~0u : 0u, m1 = part == 1 ? ~0u : 0u, m2 = part >= 2 ? ~0u : 0u;
    return (p0 & m0) | (p1 & m1) | (p2 & m2);
}
struct LineBits { uint32_t w01, w23, w45, meta; };      /* the bytes of a Line16: meta = line | fl << 16 | ref << 24 */
static_assert(sizeof(Line16) == 16 && sizeof(LineBits) == 16, "line layout");

__device__ inline LineBits compact_raw(const uint4 &a, const uint4 &b, uint32_t frame, bool ignore_crc, uint32_t &seen)
{
    const uint32_t srv = (b.z >> 8) & 0xFF;
    const bool match = a.x == frame, data = match && srv == SDV_SRV_NO;
    if (a.x > frame) seen |= FF_FOREIGN;
    if (match && srv == SDV_SRV_NEW_FILE) seen |= FF_NEW_FILE;
    if (match && srv == SDV_SRV_END_FILE) seen |= FF_END_FILE;
    const uint32_t M = (uint32_t)WORD_MASK | ((uint32_t)WORD_MASK << 16), S = (uint32_t)BIT_RANGE | ((uint32_t)BIT_RANGE << 16);   /* a service line is a cleared line */
    LineBits l;
    l.w01 = data ? ((uint32_t)(((((uint64_t)a.z) << 32) | a.y) >> 16) & M) : S;
    l.w23 = data ? ((uint32_t)(((((uint64_t)a.w) << 32) | a.z) >> 16) & M) : S;
    l.w45 = data ? ((uint32_t)(((((uint64_t)b.x) << 32) | a.w) >> 16) & M) : S;
    const bool ci = data && (b.y & 0xFFFF) == (b.x >> 16), crc = ci && !(b.z & ((uint32_t)SDV_LF_FORCED_BAD << 16)), bw = data && (b.z & ((uint32_t)SDV_LF_BW_SET << 16)) != 0;
    const uint32_t fl = (crc ? LF_VALID : 0u) | (bw ? LF_BW : 0u) | (ci ? LF_CI : 0u) | ((data || (match && srv == SDV_SRV_FILLER)) ? LF_DATA : 0u) |
                        ((match && srv == SDV_SRV_HEADER_LINE) ? LF_HDR : 0u) | ((ignore_crc ? bw : crc) ? LF_OK : 0u) |
                        ((data && (b.z & 0xFF) != 0) ? LF_PICK : 0u) | ((data && (b.y >> 24) != 0) ? LF_PICKL : 0u);
    l.meta = (a.y & 0xFFFF) | (fl << 16) | (data ? (b.y << 8) & 0xFF000000u : 0u);
    return l;
}
__device__ inline Line16 compact(const sdv_pcm1_line_rec &r, uint32_t frame, bool ignore_crc, uint32_t &seen)
{
    const Raw32 *src = (const Raw32 *)&r;
    const LineBits b = compact_raw(src->a, src->b, frame, ignore_crc, seen);
    Line16 l;
    __builtin_memcpy(&l, &b, sizeof(l));
    return l;
}

/* one wave, one frame.  kLds: the frame's lines are staged in `lines` (LDS) by the first sweep; otherwise later sweeps read the records again. */
/* kVis: the visualiser's feeds are written as well (a build of its own: the frames of a caller who does not ask for them pay nothing) */
template <bool kLds, bool kVis>
__device__ inline bool frame_body(const FrameArgs1 &a, uint32_t k, int lane, uint32_t lo, uint32_t n, Line16 *lines, uint16_t (*field_idx)[LINES_PF + 3])
{
    const uint32_t frame = a.src.at(lo + n).frame_number;           /* the END_FRAME record */
    const bool plain = a.stat[2] == 0;
    const uint32_t marks = plain ? 0u : a.marks[k];
    const uint64_t pofs = plain ? (uint64_t)k * (2 * SUBLINES_PF) : a.pair_ofs[k];
    const uint32_t fofs = plain ? k : a.frasm_ofs[k];
    const Cfg1 cfg = a.cfg;
    uint32_t err = 0u, seen = 0;
    /* (a frame too long for the LDS staging reads its records again on every access; `lines` is then free and holds the lines earlier frames left in
     * the field buffers, if the frame is told to put any out: field_idx entries with bit 15 set point there) */
    auto line_at = [&](uint32_t i) -> Line16 {
        if (kLds) return lines[i];
        if (i & 0x8000u) return lines[i & 0x7FFFu];
        uint32_t dummy = 0; return compact(a.src.at(lo + i), frame, cfg.ignore_crc != 0, dummy);
    };

    P1_STAMP(0);
    /* 1. findFrameTrim: every item is a first / last / count over the lines in stream order; index 0 = odd lines, 1 = even lines */
    /* The sweep itself only loads, compacts, stores to LDS and notes which of the eight kinds of line
     *     q = 4 * (even line) + {0: CRC valid, 1: B/W levels, 2: CRC valid ignoring "forced bad", 3: header line}
     * occur at all (plus the count of valid lines per field).  The first and the last line of every kind that does occur is then
     * searched from either end of the staged lines - found in the first chunk looked at on any ordinary frame.
     * Four chunks of 64 records per turn: the eight 16-byte loads of a turn are issued back to back, so a 490-line frame costs
     * two memory round trips instead of eight. */
    uint32_t any8 = 0, goods = 0;                    /* goods: valid odd lines | valid even lines << 16 */
    /* fillUntilFullFrame keeps BUF_SIZE_TRIM of the frame's lines at most (:178-195): what lies behind the last one kept is not looked at */
    uint32_t n_scan = n;
    if (n > (uint32_t)BUF_TRIM) { if (lane == 0) n_scan = kept_records(a.src, lo, n, frame); n_scan = (uint32_t)__shfl((int)n_scan, 0); }
    auto kinds = [](uint32_t meta) -> uint32_t { const uint32_t fl = meta >> 16, bits = (fl & 7u) | ((fl >> 1) & 8u); return (meta & 1u) ? bits : bits << 4; };
    for (uint32_t c4 = 0; c4 < n_scan; c4 += 256) {
        Raw32 raw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = c4 + 64u * (uint32_t)u + (uint32_t)lane;
            const Raw32 *src = (const Raw32 *)&a.src.at(lo + (i < n_scan ? i : n_scan - 1));
            raw[u].a = src->a; raw[u].b = src->b;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t i = c4 + 64u * (uint32_t)u + (uint32_t)lane;
            const LineBits l = compact_raw(raw[u].a, raw[u].b, frame, cfg.ignore_crc != 0, seen);
            if (i < n_scan) {
                if (kLds) *(LineBits *)&lines[i] = l;
                const uint32_t m8 = kinds(l.meta);
                any8 |= m8;
                goods += (m8 & 1u) | ((m8 & 16u) << 12);
            }
        }
    }
    for (int d = 1; d < 64; d <<= 1) goods += (uint32_t)__shfl((int)goods, lane ^ d);
    __syncthreads();            /* the staged lines are read by other lanes from here on: LDS writes of the sweep made visible */
    uint32_t present = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) if (__ballot((any8 >> q) & 1u)) present |= 1u << q;
    uint32_t f0 = P1_NONE, f1 = P1_NONE, f2 = P1_NONE, f3 = P1_NONE, f4 = P1_NONE, f5 = P1_NONE, f6 = P1_NONE, f7 = P1_NONE;
    uint32_t l0 = 0, l1 = 0, l2 = 0, l3 = 0, l4 = 0, l5 = 0, l6 = 0, l7 = 0;          /* index + 1, 0 = none */
    auto kinds_at = [&](uint32_t i) -> uint32_t {
        if (i >= n_scan) return 0u;
        if (kLds) return kinds(((const LineBits *)&lines[i])->meta);
        uint32_t dummy = 0; const Raw32 *src = (const Raw32 *)&a.src.at(lo + i);
        return kinds(compact_raw(src->a, src->b, frame, cfg.ignore_crc != 0, dummy).meta);
    };
#define P1_FIRST(q, f) if ((pending >> (q)) & 1u) { const uint64_t m = __ballot((m8 >> (q)) & 1u); if (m) { f = c + (uint32_t)__ffsll((unsigned long long)m) - 1u; pending &= ~(1u << (q)); } }
#define P1_LAST(q, l) if ((pending >> (q)) & 1u) { const uint64_t m = __ballot((m8 >> (q)) & 1u); if (m) { l = c + 64u - (uint32_t)__clzll((unsigned long long)m); pending &= ~(1u << (q)); } }
    uint32_t pending = present;
    for (uint32_t c = 0; pending != 0 && c < n_scan; c += 64) {
        const uint32_t m8 = kinds_at(c + (uint32_t)lane);
        P1_FIRST(0, f0) P1_FIRST(1, f1) P1_FIRST(2, f2) P1_FIRST(3, f3) P1_FIRST(4, f4) P1_FIRST(5, f5) P1_FIRST(6, f6) P1_FIRST(7, f7)
    }
    pending = present;
    for (uint32_t c = n_scan ? ((n_scan - 1) / 64) * 64 : 0u; pending != 0; c -= 64) {
        const uint32_t m8 = kinds_at(c + (uint32_t)lane);
        P1_LAST(0, l0) P1_LAST(1, l1) P1_LAST(2, l2) P1_LAST(3, l3) P1_LAST(4, l4) P1_LAST(5, l5) P1_LAST(6, l6) P1_LAST(7, l7)
        if (c == 0) break;
    }
#undef P1_FIRST
#undef P1_LAST
    const uint32_t good[2] = { goods & 0xFFFF, goods >> 16 };
    const uint32_t first_valid[2] = { f0, f4 }, first_bw[2] = { f1, f5 }, first_ci[2] = { f2, f6 }, first_hdr[2] = { f3, f7 };
    const uint32_t last_valid[2] = { l0, l4 }, last_bw[2] = { l1, l5 }, last_ci[2] = { l2, l6 }, last_hdr[2] = { l3, l7 };
    P1_STAMP(1);
    seen = (uint32_t)(__ballot(seen & FF_FOREIGN) ? FF_FOREIGN : 0) | (uint32_t)(__ballot(seen & FF_NEW_FILE) ? FF_NEW_FILE : 0) | (uint32_t)(__ballot(seen & FF_END_FILE) ? FF_END_FILE : 0);
    if (seen & FF_FOREIGN) err |= FE_FOREIGN;
    if ((seen ^ marks) & (FF_NEW_FILE | FF_END_FILE)) err |= FE_MARKS;      /* a file tag with another frame's number: the offsets were laid out for it */

    if (marks & FF_END_FILE) {                      /* the frame that carries the END_FILE tag only closes the file (:1721-1729) */
        if (a.cnt_out) { if (lane == 0) a.cnt_out[k] = 0; return false; }    /* ... and does not touch the field buffers */
        if (lane == 0) {
            if (fofs < a.frames_cap) { sdv_frame_asm_pcm1 d; frasm1_clear(d); d.service_type = SDV_PAIR_SRV_END_FILE; a.out_frames[fofs] = d; }
            if (pofs < a.pairs_cap) service_pair(&a.out_pairs[pofs], SDV_PAIR_SRV_END_FILE);
            if (err) { atomicOr(&a.stat[0], err); atomicMin(&a.stat[1], k); }
        }
        return false;
    }

    bool header_present = false, emphasis_set = false;
    uint32_t top[2], bottom[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        /* a header line counts while no valid data line of its field has been seen: from the top for "header present" (:262-285),
         * from the bottom for the emphasis flag (:306-352; the backward scan stops at the second field's last valid line, which is
         * never past the own field's) */
        if (first_hdr[p] != P1_NONE && first_hdr[p] < first_valid[p]) header_present = true;
        if (last_hdr[p] > last_valid[p]) emphasis_set = true;
        const bool skip_bad = good[p] > MIN_GOOD;
        const uint32_t fi = skip_bad ? first_ci[p] : first_bw[p], la = skip_bad ? last_ci[p] : last_bw[p];
        top[p] = fi != P1_NONE ? (kLds ? lines[fi].line : a.src.at(lo + fi).line_number) : 0u;
        bottom[p] = la != 0 ? (kLds ? lines[la - 1].line : a.src.at(lo + la - 1).line_number) : 0u;
    }
    if (!cfg.auto_offset) {                          /* :388-412 */
        top[0] = cfg.odd_offset > 0 ? (uint32_t)(2 * cfg.odd_offset + 1) : 1u;
        top[1] = cfg.even_offset > 0 ? (uint32_t)(2 * cfg.even_offset + 2) : 2u;
    }
    if (marks & FF_NEW_FILE) header_present = emphasis_set = false;            /* resetState after the trim search (:1688-1692, :63-76) */

    P1_STAMP(2);
    /* 2. splitFrameToFields: rank the lines of either field, 245 at most */
    uint32_t cnt[2] = { 0, 0 }, valid[2] = { 0, 0 }, refs[2] = { 0, 0 };       /* refs: sum over valid lines << 16 | sum over all lines (245 x 255 fits) */
    const bool even_open = top[1] != bottom[1] || top[1] != 0;
    for (uint32_t c = 0; c < n_scan; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        bool in[2] = { false, false };
        bool ok = false; uint32_t ref = 0;
        if (i < n_scan) {
            const Line16 l = line_at(i);
            if (l.fl & LF_DATA) {
                const uint32_t ln = l.line;
                in[0] = (ln & 1) != 0 && ln >= top[0] && ln <= bottom[0];
                in[1] = (ln & 1) == 0 && ln >= top[1] && ln <= bottom[1] && even_open;
                ok = (l.fl & LF_VALID) != 0; ref = l.ref;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint64_t m = __ballot(in[p]);
            const uint32_t rank = cnt[p] + (uint32_t)__popcll(m & lanemask_lt(lane));
            const bool take = in[p] && rank < LINES_PF;
            if (take) { field_idx[p][rank] = (uint16_t)i; refs[p] += ref + (ok ? ref << 16 : 0u); }
            valid[p] += (uint32_t)__popcll(__ballot(take && ok));
            cnt[p] += (uint32_t)__popcll(m);
        }
    }
    __syncthreads();            /* field_idx[] complete before the output stage gathers through it */
    if (a.cnt_out) {            /* first pass of manual mode: what this frame writes into the field buffers */
        const uint32_t c0 = cnt[0] > LINES_PF ? (uint32_t)LINES_PF : cnt[0], c1 = cnt[1] > LINES_PF ? (uint32_t)LINES_PF : cnt[1];
        if (lane == 0) a.cnt_out[k] = c0 | (c1 << 16);
        for (uint32_t q = (uint32_t)lane; q < c0; q += 64) a.kept_out[(size_t)k * (2 * LINES_PF) + q] = lo + field_idx[0][q];
        for (uint32_t q = (uint32_t)lane; q < c1; q += 64) a.kept_out[(size_t)k * (2 * LINES_PF) + LINES_PF + q] = lo + field_idx[1][q];
        return false;
    }
    P1_STAMP(3);
    sdv_frame_asm_pcm1 f; frasm1_clear(f);
    f.frame_number = frame;
    uint32_t data[2], ref_level[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if (cnt[p] > LINES_PF) cnt[p] = LINES_PF;
        for (int d = 1; d < 64; d <<= 1) refs[p] += (uint32_t)__shfl((int)refs[p], lane ^ d);
        data[p] = 3 * cnt[p];
        /* the reference adds the level once per sub-line and divides by the sub-line count (:767-798): the same quotient.  Sums and
         * counts are below 2^16: the float quotient is exact enough for the floor (never within 1/245 of the next integer) */
        const uint32_t num = valid[p] > 0 ? refs[p] >> 16 : refs[p] & 0xFFFF, den = valid[p] > 0 ? valid[p] : cnt[p];
        ref_level[p] = den > 0 ? (uint32_t)((float)num / (float)den) & 0xFF : 0u;
    }

    /* 3. findFramePadding (:809-923); uint16_t arithmetic as the reference's fields */
    uint32_t top_pad[2], bot_pad[2];        /* hold uint16_t values */
    uint32_t stale_to[2] = { 0, 0 };        /* field positions [cnt, stale_to) come from earlier frames */
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
            top_pad[p] = ofs > 0 ? 0u : (uint32_t)(uint16_t)(0 - ofs);
            uint16_t bp = (uint16_t)(((int)bottom[p] - (int)top[p]) / 2 + 1);
            bp = (uint16_t)(bp + top_pad[p]);
            if (bp > LINES_PF) {
                bp = (uint16_t)(bp - LINES_PF);
                bottom[p] = (uint16_t)(bottom[p] - bp * 2);
                uint16_t dl = (uint16_t)(((int)bottom[p] - (int)top[p]) / 2 + 1);
                dl = (uint16_t)(dl * 3);
                if (dl > data[p]) stale_to[p] = dl / 3u;     /* the field buffer is read past what this frame wrote: earlier frames' lines */
                data[p] = dl;
            }
            bot_pad[p] = (uint32_t)(uint16_t)((SUBLINES_PF - (int)data[p]) / 3 - (int)top_pad[p]);
        }
    }
    if (stale_to[0] | stale_to[1]) {
        /* the lines earlier frames left at those places join the frame's staged lines (behind its own records) */
        const uint32_t s0 = stale_to[0] > cnt[0] ? stale_to[0] - cnt[0] : 0u, s1 = stale_to[1] > cnt[1] ? stale_to[1] - cnt[1] : 0u;
        const uint32_t room_from = kLds ? n_scan : 0u;      /* where `lines` is free */
        if (kLds && room_from + s0 + s1 > (uint32_t)LDS_LINES) return true;        /* no room behind the frame's own lines: once more on the path that leaves `lines` free (nothing was written yet) */
        /* There the room always suffices: a field is told to read at most LINES_PF places of its buffer - the manual offsets are int8, so the padding
         * above a field is at most 128 lines and the trimmed field (245 lines minus that padding, pcm1datastitcher.cpp:896-909) never wraps - and
         * LDS_LINES holds two fields.  (For the same reason the queue handed to PCM1Deinterleaver is always exactly one field:
         * its DI_RET_NO_DATA, pcm1deinterleaver.cpp:104 / 119, cannot be reached through the stitcher - tests/test_pcm1.py walks all offsets.) */
        static_assert(LDS_LINES >= 2 * LINES_PF, "the lines two fields can be told to take from earlier frames fit the staging buffer");
        {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                const uint32_t base = room_from + (p ? s0 : 0u);
                for (uint32_t q = cnt[p] + (uint32_t)lane; q < stale_to[p]; q += 64) {
                    Line16 l;
                    uint32_t j = k;
                    while (j > 0 && ((a.cnt_in[j - 1] >> (16 * p)) & 0xFFFFu) <= q) j--;
                    if (j > 0) {
                        const sdv_pcm1_line_rec &r = a.src.at(a.kept_in[(size_t)(j - 1) * (2 * LINES_PF) + (size_t)p * LINES_PF + q]);
                        uint32_t dummy = 0;
                        l = compact(r, r.frame_number, cfg.ignore_crc != 0, dummy);
                    } else l = ((const Line16 *)a.hist)[p * LINES_PF + q];
                    lines[base + q - cnt[p]] = l;
                    field_idx[p][q] = (uint16_t)((base + q - cnt[p]) | (kLds ? 0u : 0x8000u));
                }
            }
        }
        __syncthreads();
    }
    const uint8_t order = cfg.field_order == ORDER_BFF ? ORDER_BFF : ORDER_TFF;

    P1_STAMP(4);
    /* 4. the two fields in output order (:1076-1218, :1382-1453) */
    uint64_t po = pofs;
    if (marks & FF_NEW_FILE) {
        if (lane == 0) {
            if (fofs < a.frames_cap) { sdv_frame_asm_pcm1 d; frasm1_clear(d); d.service_type = SDV_PAIR_SRV_NEW_FILE; a.out_frames[fofs] = d; }
            if (po < a.pairs_cap) service_pair(&a.out_pairs[po], SDV_PAIR_SRV_NEW_FILE);
        }
        po++;
    }
    uint32_t blocks_drop = 0, samples_drop = 0, blocks_fix_bp = 0;
    /* pairs are addressed as a 32-bit byte offset from the field's (uniform) base; `lim`: how many of the field's pairs fit the buffer */
    for (int fld = 0; fld < 2; fld++) {
        const bool odd_field = (order == ORDER_TFF) == (fld == 0);
        const uint16_t *fidx = odd_field ? field_idx[0] : field_idx[1];
        const uint32_t f_top = odd_field ? top_pad[0] : top_pad[1], f_bot = odd_field ? bot_pad[0] : bot_pad[1], f_data = odd_field ? data[0] : data[1];
        const uint32_t f_lines = f_data <= SUBLINES_PF ? f_data / 3 : 0u;      /* addLinesFromField refuses more than a field (:960) */
        /* The padded field is 245 lines of 3 sub-lines; a lane takes a line.  First the lines' flags as three 245-bit masks ... */
        uint64_t okm[4], pkm[4], plm[4], nvm[4];
        uint32_t li[4];
        /* (a place of the field buffer that no frame has ever written holds a cleared sub-line, part 0: the stitcher's line counter does not move on
         * behind it, addLinesFromField :974-978 - only the visualiser's feeds carry those numbers) */
        auto stale_at = [&](uint32_t i) -> bool { return kLds ? i >= n_scan : (i & 0x8000u) != 0; };
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t pl = (uint32_t)(c * 64 + lane);
            const bool is_line = pl >= f_top && pl - f_top < f_lines && pl < LINES_PF;
            li[c] = is_line ? fidx[pl - f_top] : P1_NONE;
            const uint32_t fl = is_line ? (kLds ? (uint32_t)lines[li[c]].fl : (uint32_t)line_at(li[c]).fl) : 0u;
            okm[c] = __ballot((fl & LF_OK) != 0); pkm[c] = __ballot((fl & LF_PICK) != 0); plm[c] = __ballot((fl & LF_PICKL) != 0);
            nvm[c] = kVis ? __ballot(is_line && stale_at(li[c]) && (fl & LF_DATA) == 0) : 0ull;
        }
        auto rows_unnumbered_below = [&](uint32_t row) -> uint32_t {
            uint32_t n = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                if (row >= 64u * (uint32_t)(w + 1)) n += (uint32_t)__popcll(nvm[w]);
                else if (row > 64u * (uint32_t)w) n += (uint32_t)__popcll(nvm[w] & ((1ull << (row - 64u * (uint32_t)w)) - 1ull));
            }
            return n;
        };
        okm[3] |= ~0ull << (LINES_PF - 192);                    /* lines past the field do not exist: not "bad" */
        if (fld == 0) P1_STAMP(5);
        /* ... then the blocks: block b is sub-lines [92 b, 92 b + 92) (the last one: 91), i.e. whole lines plus a partial line at
         * either end.  All ranges are compile-time constants once the loop is unrolled. */
        uint32_t valid_blocks = 0;
#pragma unroll
        for (int blk = 0; blk < 8; blk++) {
            const int s0 = 92 * blk, s1 = blk == 7 ? SUBLINES_PF : s0 + 92, l0 = s0 / 3, l1 = (s1 - 1) / 3;
            uint32_t bad_lines = 0, any_pick = 0;
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const int lo_b = l0 > 64 * w ? l0 - 64 * w : 0, hi_b = l1 < 64 * w + 63 ? l1 - 64 * w : 63;     /* bits of word w inside [l0, l1] */
                if (lo_b <= hi_b && hi_b >= 0) {
                    const uint64_t m = (~0ull >> (63 - hi_b)) & (~0ull << lo_b);
                    bad_lines += (uint32_t)__popcll(~okm[w] & m);
                    any_pick |= (pkm[w] & m) != 0 ? 1u : 0u;
                    /* picked_bits_left belongs to the line's first sub-line only: lines whose sub-line 3 L lies inside the block */
                    const int f0 = (s0 + 2) / 3, fl_lo = f0 > 64 * w ? f0 - 64 * w : 0;
                    if (fl_lo <= hi_b) any_pick |= (plm[w] & (~0ull >> (63 - hi_b)) & (~0ull << fl_lo)) != 0 ? 1u : 0u;
                }
            }
            const uint32_t cut0 = (uint32_t)(s0 - 3 * l0), cut1 = (uint32_t)(3 * l1 + 3 - s1);          /* sub-lines of the edge lines outside the block */
            const uint32_t bad0 = (uint32_t)(~okm[l0 / 64] >> (l0 % 64)) & 1u, bad1 = (uint32_t)(~okm[l1 / 64] >> (l1 % 64)) & 1u;
            const uint32_t bad = 3 * bad_lines - cut0 * bad0 - cut1 * bad1;
            if (bad) { blocks_drop++; samples_drop += (2 * bad) & 0xFF; }
            else { valid_blocks |= 1u << blk; if (any_pick) blocks_fix_bp++; }
        }
        /* ... and the pairs in output order, a lane per pair: consecutive lanes write consecutive 12-byte pairs (one 768-byte run per
         * store instruction) and gather their sub-line from LDS.  setWordData (:150-278): the stripe that starts at word 2 (odd pairs)
         * reads the first 46 sub-lines of the block in even blocks and the second 46 in odd blocks; the stripe that starts at word 0
         * the others */
        char *const base = (char *)(a.out_pairs + po);
        const uint32_t lim = po >= a.pairs_cap ? 0u : (a.pairs_cap - po >= SUBLINES_PF ? (uint32_t)SUBLINES_PF : (uint32_t)(a.pairs_cap - po));
        for (uint32_t c = 0; c < SUBLINES_PF; c += 64) {
            const uint32_t o = c + (uint32_t)lane;
            const uint32_t blk = o / 92, pr = o - 92 * blk;
            const uint32_t sub = 92 * blk + ((((blk & 1) == 0) == ((pr & 1) != 0)) ? 0u : 46u) + (pr >> 1);
            const uint32_t pl = sub / 3, part = sub - 3 * pl;
            uint32_t ww = (uint32_t)BIT_RANGE | ((uint32_t)BIT_RANGE << 16); bool ok = false;
            if (o < lim && pl >= f_top && pl - f_top < f_lines) {
                const uint32_t li1 = fidx[pl - f_top];
                if (kLds) { ww = ((const uint32_t *)lines[li1].w)[part]; ok = (lines[li1].fl & LF_OK) != 0; }
                else { const Line16 l = line_at(li1); ww = line_word_pair(l, part); ok = (l.fl & LF_OK) != 0; }
            }
            const uint32_t flags = ((valid_blocks >> blk) & 1u ? (uint32_t)SDV_SF_BLOCK_OK : 0u) | (ok ? (uint32_t)SDV_SF_WORD_VALID : 0u);
            if (o < lim)
                store_pair((sdv_sample_pair *)(base + o * 12u), make_pair3(p1_sample((uint16_t)(ww & 0xFFFF)), p1_sample((uint16_t)(ww >> 16)), flags, 44100, emphasis_set, 0));
        }
        /* the visualiser's feeds.  newBlockProcessed (:1333): the block as PCM1Deinterleaver::setWordData filled it - pair pr of block blk is its words
         * 2 pr, 2 pr + 1 (what the loop above reads); newLineProcessed (:1392-1407): the queue itself, sub-line after sub-line.  The stitcher numbers the
         * lines of its queue anew (addFieldPadding / addLinesFromField, :952-1073): 1, 3, 5 ... down the odd field, 2, 4, 6 ... down the even one. */
        if (kVis) {
            const uint64_t d = (pofs - fofs) / (uint64_t)(2 * SUBLINES_PF - 1);       /* frames ahead of this one that are no file tags: their pairs minus their descriptors */
            const uint32_t first_line = odd_field ? 1u : 2u;
            if (a.out_blocks) {
                const uint64_t bb = d * 16u + 8u * (uint32_t)fld;
#pragma unroll 1
                for (uint32_t c = 0; c < SUBLINES_PF + 1; c += 64) {
                    const uint32_t o = c + (uint32_t)lane;                              /* o = 735: the two words the last block does not have */
                    const uint32_t blk = o < SUBLINES_PF ? o / 92 : 7u, pr = o - 92 * blk;
                    if (o > SUBLINES_PF || bb + blk >= a.blocks_cap) continue;
                    sdv_pcm1_block_rec *b = &a.out_blocks[bb + blk];
                    uint32_t ww = (uint32_t)BIT_RANGE | ((uint32_t)BIT_RANGE << 16), f0 = 0, f1 = 0;
                    if (o < SUBLINES_PF) {
                        const uint32_t sub = 92 * blk + ((((blk & 1) == 0) == ((pr & 1) != 0)) ? 0u : 46u) + (pr >> 1);
                        const uint32_t pl = sub / 3, part = sub - 3 * pl;
                        if (pl >= f_top && pl - f_top < f_lines) {
                            const Line16 l = line_at(fidx[pl - f_top]);
                            ww = line_word_pair(l, part);
                            const uint32_t ok = (l.fl & LF_OK) ? (uint32_t)SDV_P1W_CRC_OK : 0u, pc = (l.fl & LF_PICK) ? (uint32_t)SDV_P1W_PICKED_WORD : 0u;
                            f0 = ok | pc | ((part == 0 && (l.fl & LF_PICKL)) ? (uint32_t)(SDV_P1W_PICKED_LEFT | SDV_P1W_PICKED_WORD) : 0u); f1 = ok | pc;
                        }
                    }
                    *(uint32_t *)&b->words[2 * pr] = ww;
                    b->word_flags[2 * pr] = (uint8_t)f0; b->word_flags[2 * pr + 1] = (uint8_t)f1;
                    if (pr == 0) {
                        const uint32_t r0 = (92 * blk) / 3, r1 = (92 * blk + 91) / 3;
                        /* the number of the block's first sub-line: the frame's - a filler line of the odd field went into the field buffer as a cleared
                         * PCM1Line, number 0 (splitFrameToFields, :713-717); lines earlier frames left in the field buffers count as this frame's */
                        uint32_t fno = frame;
                        if (r0 >= f_top && r0 - f_top < f_lines) {
                            const uint32_t l0i = fidx[r0 - f_top];
                            if (odd_field && !stale_at(l0i) && a.src.at(lo + l0i).service_type == SDV_SRV_FILLER) fno = 0;
                        }
                        b->frame_number = fno;
                        b->start_line = (uint16_t)(first_line + 2 * (r0 - rows_unnumbered_below(r0))); b->stop_line = (uint16_t)(first_line + 2 * (r1 - rows_unnumbered_below(r1)));
                        b->interleave_num = (uint8_t)blk; b->flags = (uint8_t)((blk == 7 ? SDV_P1B_SHORT : 0) | (emphasis_set ? SDV_P1B_EMPHASIS : 0));
                        b->sample_rate = 44100;
                        for (int i = 0; i < 12; i++) b->_pad[i] = 0;
                    }
                }
            }
            if (a.out_asm) {
                const uint64_t lb = d * (uint64_t)(2 * SUBLINES_PF) + (uint64_t)SUBLINES_PF * (uint32_t)fld;
#pragma unroll 1
                for (uint32_t c = 0; c < SUBLINES_PF; c += 64) {
                    const uint32_t sub = c + (uint32_t)lane, pl = sub / 3, part = sub - 3 * pl;
                    if (sub >= SUBLINES_PF || lb + sub >= a.asm_cap) continue;
                    sdv_pcm1_asm_line_rec r;
                    r.frame_number = frame; r.line_number = (uint16_t)(first_line + 2 * (pl - rows_unnumbered_below(pl))); r.words[0] = r.words[1] = (uint16_t)BIT_RANGE;
                    r.picked_bits_left = r.picked_bits_right = 0; r.line_part = (uint8_t)part; r.flags = 0; r._pad[0] = r._pad[1] = 0;
                    if (pl >= f_top && pl - f_top < f_lines) {
                        const uint32_t li1 = fidx[pl - f_top];
                        const bool own = kLds ? li1 < n_scan : (li1 & 0x8000u) == 0;       /* else: a line an earlier frame left in the field buffer */
                        if (!own) { r.flags = SDV_P1S_SKIP; if ((line_at(li1).fl & LF_DATA) == 0) r.line_part = 0; }
                        else {
                            const Line16 l = line_at(li1);
                            const sdv_pcm1_line_rec &src = a.src.at(lo + li1);
                            if (odd_field && src.service_type == SDV_SRV_FILLER) r.flags = SDV_P1S_SKIP;     /* went in with frame number 0 (:713-717): not handed over */
                            else {
                                { const uint32_t wp = line_word_pair(l, part); r.words[0] = (uint16_t)wp; r.words[1] = (uint16_t)(wp >> 16); }
                                r.flags = (uint8_t)(((l.fl & LF_BW) ? SDV_P1S_BW_SET : 0) | ((l.fl & LF_VALID) ? SDV_P1S_CRC_VALID : 0));
                                if (l.fl & (LF_PICK | LF_PICKL)) { r.picked_bits_left = part == 0 ? src.picked_bits_left : 0; r.picked_bits_right = src.picked_bits_right; }
                            }
                        }
                    }
                    a.out_asm[lb + sub] = r;
                }
            }
        }
        po += SUBLINES_PF;
    }
    P1_STAMP(6);
    if (lane == 0) {
        f.odd_std_lines = f.even_std_lines = LINES_PF;
        f.odd_data_lines = (uint16_t)(data[0] / 3); f.even_data_lines = (uint16_t)(data[1] / 3);
        f.odd_valid_lines = (uint16_t)valid[0]; f.even_valid_lines = (uint16_t)valid[1];
        f.odd_top_data = (uint16_t)top[0]; f.odd_bottom_data = (uint16_t)bottom[0]; f.even_top_data = (uint16_t)top[1]; f.even_bottom_data = (uint16_t)bottom[1];
        f.odd_sample_rate = f.even_sample_rate = 44100;
        f.blocks_total = 16; f.blocks_drop = (uint16_t)blocks_drop; f.samples_drop = (uint16_t)samples_drop; f.blocks_fix_bp = (uint16_t)blocks_fix_bp;
        f.odd_top_padding = (uint16_t)top_pad[0]; f.odd_bottom_padding = (uint16_t)bot_pad[0]; f.even_top_padding = (uint16_t)top_pad[1]; f.even_bottom_padding = (uint16_t)bot_pad[1];
        f.field_order = order; f.odd_ref = (uint8_t)ref_level[0]; f.even_ref = (uint8_t)ref_level[1];
        f.flags = (uint8_t)(SDV_FA_ORDER_PRESET | (emphasis_set ? (SDV_FA1_ODD_EMPHASIS | SDV_FA1_EVEN_EMPHASIS) : 0));
        const uint32_t fo = fofs + ((marks & FF_NEW_FILE) ? 1u : 0u);
        if (fo < a.frames_cap) a.out_frames[fo] = f;
        if (err) { atomicOr(&a.stat[0], err); atomicMin(&a.stat[1], k); }
    }
    return false;
}
/* the field buffers as this call leaves them: place q of field p holds the line of the last frame whose field reached that far */
struct HistArgs1 { RecSrc1 src; const uint32_t *cnt, *kept; uint32_t n_seg; uint8_t ignore_crc; void *hist; };
__device__ inline void hist_body(const HistArgs1 &a, int lane)
{
    for (uint32_t i = (uint32_t)lane; i < 2 * LINES_PF; i += 64) {
        const uint32_t p = i / LINES_PF, q = i % LINES_PF;
        uint32_t j = a.n_seg;
        while (j > 0 && ((a.cnt[j - 1] >> (16 * p)) & 0xFFFFu) <= q) j--;
        if (j > 0) {
            const sdv_pcm1_line_rec &r = a.src.at(a.kept[(size_t)(j - 1) * (2 * LINES_PF) + (size_t)p * LINES_PF + q]);
            uint32_t dummy = 0;
            ((Line16 *)a.hist)[i] = compact(r, r.frame_number, a.ignore_crc != 0, dummy);
        }
    }
}
/* a fresh stitcher's field buffers: default-constructed PCM1SubLines (silent words, nothing valid; pcm1subline.cpp:47-64) */
__device__ inline void hist_clear_body(void *hist, int lane)
{
    for (uint32_t i = (uint32_t)lane; i < 2 * LINES_PF; i += 64) {
        Line16 l; for (int w = 0; w < 6; w++) l.w[w] = BIT_RANGE; l.line = 0; l.fl = 0; l.ref = 0;
        ((Line16 *)hist)[i] = l;
    }
}
/* ---- front half -> back half: what PCM1DataStitcher reads of a PCM1Line (sdv_pcm1_line_rec) out of the record the frame driver writes ---- */
struct ConvArgs1 { const sdv_pcm1_bin_rec *in; sdv_pcm1_line_rec *out; size_t n; };
__device__ inline void conv_body(const ConvArgs1 &a, size_t i)
{
    if (i >= a.n) return;
    const sdv_pcm1_bin_rec r = a.in[i];
    sdv_pcm1_line_rec o = sdv_pcm1_line_rec();
    o.frame_number = r.frame_number; o.line_number = r.line_number;
    for (int w = 0; w < 7; w++) o.words[w] = r.words[w];
    o.calc_crc = r.calc_crc; o.ref_level = r.ref_level; o.picked_bits_left = r.picked_bits_left; o.picked_bits_right = r.picked_bits_right;
    o.service_type = r.service_type; o.flags = (uint8_t)(r.flags & (SDV_LF_BW_SET | SDV_LF_FORCED_BAD));
    a.out[i] = o;
}
} // namespace sdvp1

__global__ void __launch_bounds__(64) sdv_k_pcm1_bin_to_line(sdvp1::ConvArgs1 a) { sdvp1::conv_body(a, (size_t)blockIdx.x * 64u + threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm1_segments(sdvp1::SegArgs1 a) { sdvp1::seg_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm1_scan(sdvp1::ScanArgs1 a) { sdvp1::scan_body(a, (int)threadIdx.x); }
/* What the host wants to know at the end of a stitch call (PCM-1 and PCM-16x0 engines) lies in four places; this puts it behind the four status
 * words so that one small copy brings all of it: stat[4..5] = pairs of the call, stat[6] = frame descriptors, stat[7] = index of the last END_FRAME. */
struct sdv_stitch_tail_args { uint32_t *stat; const uint64_t *pair_total; const uint32_t *frasm_total; const uint32_t *last_end; };
__global__ void __launch_bounds__(64) sdv_k_stitch_tail(sdv_stitch_tail_args a)
{
    if (threadIdx.x != 0) return;
    const uint64_t p = *a.pair_total;
    a.stat[4] = (uint32_t)p; a.stat[5] = (uint32_t)(p >> 32); a.stat[6] = *a.frasm_total; a.stat[7] = *a.last_end;
}
__global__ void __launch_bounds__(64) sdv_k_pcm1_hist(sdvp1::HistArgs1 a) { sdvp1::hist_body(a, (int)threadIdx.x); }
struct sdv_p1_hist_clear_args { void *hist; };
__global__ void __launch_bounds__(64) sdv_k_pcm1_hist_clear(sdv_p1_hist_clear_args a) { sdvp1::hist_clear_body(a.hist, (int)threadIdx.x); }
__global__ void __launch_bounds__(64, 4) sdv_k_pcm1_frames(sdvp1::FrameArgs1 a)
{
    alignas(16) __shared__ sdvp1::Line16 lines[sdvp1::LDS_LINES];
    __shared__ uint16_t field_idx[2][sdvp1::LINES_PF + 3];
    const uint32_t k = blockIdx.x;
    const uint32_t lo = k == 0 ? 0u : a.seg_end[k - 1] + 1u, n = a.seg_end[k] - lo;
    bool again = true;
    if (n <= sdvp1::LDS_LINES) again = sdvp1::frame_body<true, false>(a, k, (int)threadIdx.x, lo, n, lines, field_idx);
    if (again) { __syncthreads(); (void)sdvp1::frame_body<false, false>(a, k, (int)threadIdx.x, lo, n, lines, field_idx); }
}
/* ... with the visualiser's feeds (sdv_set_pcm1_stitch_block_output / _line_output) */
__global__ void __launch_bounds__(64) sdv_k_pcm1_frames_vis(sdvp1::FrameArgs1 a)
{
    alignas(16) __shared__ sdvp1::Line16 lines[sdvp1::LDS_LINES];
    __shared__ uint16_t field_idx[2][sdvp1::LINES_PF + 3];
    const uint32_t k = blockIdx.x;
    const uint32_t lo = k == 0 ? 0u : a.seg_end[k - 1] + 1u, n = a.seg_end[k] - lo;
    bool again = true;
    if (n <= sdvp1::LDS_LINES) again = sdvp1::frame_body<true, true>(a, k, (int)threadIdx.x, lo, n, lines, field_idx);
    if (again) { __syncthreads(); (void)sdvp1::frame_body<false, true>(a, k, (int)threadIdx.x, lo, n, lines, field_idx); }
}
#endif
