/*
 * pcm16_stitch_device.h - the PCM-16x0 back half on the device: PCM16X0DataStitcher::doFrameReassemble
 * (pcm16x0datastitcher.cpp:5652-5856) with PCM16X0Deinterleaver::processBlock (pcm16x0deinterleaver.cpp:128-708).
 *
 * The reference stitches one frame at a time; what a frame inherits from its predecessors is small: the padding that won most
 * often in the last 65 decisions (getProbablePadding, tried first) and the Control Bit majorities of the last 65 frames (used when
 * a frame's own Control Bits do not read).  Everything expensive is a pure function of the frame:
 *   K-A  sdv_k_pcm16_analyse   one wave per frame, all frames at once: the frame's sub-lines are compacted into LDS (12 bytes each),
 *        trimmed (findFrameTrim :213, the record walk of the reference on lane 0), ranked into the two fields (:566), scanned for
 *        false-positive CRCs (:753) and then checked under EVERY padding the sweep could try: SI 35 paddings x 5 interleave blocks x 35
 *        data blocks per field (trySIPadding :1129), EI 81 paddings x ~490 blocks per frame (tryEIPadding :2380) - a lane per
 *        (padding, interleave block) runs its burst counters over its own sequence of P-code checks.  Result: a table of
 *        FieldStitchStats per padding + the Control Bit offsets (findZeroControlBitOffset :868).
 *   K-B  sdv_k_pcm16_choose    one wave, the frames in order: the padding history (a histogram in registers) picks the padding every frame
 *        locks on - a handful of scalar steps per frame;  sdv_k_pcm16_finish, a lane per frame: the reference's arithmetic from there
 *        (findSIPadding :1557, findEIFrameStitching :3588, conditionEIFramePadding :2997, findEIDataAlignment :3467).
 *   K-C  sdv_k_pcm16_ctrl      one wave per frame: the Control Bits of the padded frame (collectCtrlBitStats :4745).
 *   K-D  sdv_k_pcm16_flags     one wave, the frames in order: Control Bit history -> sample rate / emphasis / code of every frame.
 *   K-E  sdv_k_pcm16_emit      one wave per frame: 490 data blocks through processBlock (a lane per block), seam and BROKEN masking
 *        as scans over the blocks (performDeinterleave :5165), 1470 PCMSamplePairs and the FrameAsmPCM16x0.
 *   K-B' sdv_k_pcm16_carry     one wave, the frames in order: a frame whose padded size is not 1470 sub-lines (sub-lines missing or
 *        doubled in a line: the reference's "WRONG COUNT" path, :4664-4703) leaves a remainder in conv_queue that the next frame's
 *        blocks start in (performDeinterleave pops whole interleave blocks only, :5216, :5431-5443).  What every frame finds in the queue
 *        and how many blocks it puts out is a segmented prefix sum over the frame sizes; the remainder itself is always the tail of the
 *        frame before, so K-C / K-E read it from there.
 * Algorithmic bytes per frame: 36 B per sub-line record in (1470: 52.9 KB) + 1470 x 12 B pairs + 56 B descriptor out.
 */
#ifndef SDV_PCM16_STITCH_DEVICE_H
#define SDV_PCM16_STITCH_DEVICE_H
#include "../../include/sdvpcm.h"
#include "stc007_stitch_device.h"
#include "pcm1_stitch_device.h"

namespace sdvp16 {
using sdvs::lanemask_lt;
using sdvp1::Pair3;
using sdvp1::make_pair3;
using sdvp1::store_pair;

enum { LINES_PF = 245, SUBLINES_PF = 735, SI_OFS = 35, EI_OFS = 490, SI_TRUE = 105, EI_TRUE = 490, IBLK_PF = 7, FRAME_SUBS = 1470 };
enum { BUF_TRIM = 3 * 640 * 3, MIN_GOOD_SUB = 35 * 6 * 3, MIN_FILL_SI = 105, MIN_FILL_EI = 82 * 3 };
enum { IBLK_DELIM = 45, MAX_PAD_SI = 35, MAX_PAD_EI = 81, MAX_SIL_SI = 34, MAX_SIL_EI = 81 * 3, MAX_BROKEN = 1, MAX_UNCH_SI = 34, MAX_UNCH_EI = 81 * 3,
       MIN_VALID_SI = 17, MIN_VALID_EI = 490 / 3, INVALID_PAD = 0xFF, STATS_DEPTH = 65 };
enum { BIT_EMPH = 0, BIT_RATE = 3, BIT_MODE = 6, BIT_CODE = 9 };
enum { DS_NO_DATA, DS_SILENCE, DS_BROKE, DS_NO_PAD, DS_OK };
enum { ORDER_TFF = 1, ORDER_BFF = 2 };
enum { FF_NEW_FILE = 1, FF_END_FILE = 2 };
/* reasons a frame cannot be stitched by this engine */
enum { FE_FOREIGN = 1, FE_TOO_LONG = 2, FE_MARKS = 16 };


/* ---- a sub-line as the stitcher needs it, 12 bytes ---------------------------------------------------------------- */
enum { SF_CRC = 1,          /* isCRCValid() (the stitcher's own "forced bad" marks included) */
       SF_CI = 2,           /* isCRCValidIgnoreForced() */
       SF_BW = 4,           /* hasBWSet() */
       SF_OKIGN = 8,        /* coords.areValid() && hasBWSet(): "valid" when CRCs are ignored (setWordData, pcm16x0deinterleaver.cpp:726-733) */
       SF_CTRL = 16,        /* control_bit */
       SF_PICKR = 32,       /* hasPickedRight() */
       SF_MATCH = 64,       /* carries the frame's number */
       SF_SKIP = 128 };     /* a service line other than a filler */
struct Sub { uint16_t w[3]; uint16_t line; uint8_t fl, part, ref, pickl; };       /* part: line_part, bit 7 set on a filler */
static_assert(sizeof(Sub) == 12, "sub-line layout");
enum { PART_FILLER = 0x80, PART_FORCED = 0x40 /* setForcedBad() by prescanForFalsePosCRCs: only the assembled-lines feed asks */ };
__device__ inline uint32_t sub_part(const Sub &s) { return s.part & 0x3Fu; }
__device__ inline Sub sub_empty()       /* a cleared PCM16X0SubLine (pcm16x0subline.cpp:63-86): silent words, CRC off, Control Bit set */
{
    Sub s; s.w[0] = s.w[1] = s.w[2] = 0; s.line = 0; s.fl = SF_CTRL; s.part = 0; s.ref = 0; s.pickl = 0; return s;
}
static_assert(sizeof(sdv_pcm16x0_bin_rec) == 36, "record layout");
__device__ inline Sub compact(const sdv_pcm16x0_bin_rec &r, uint32_t frame, uint32_t &seen)
{
    Sub s = sub_empty();
    const bool match = r.frame_number == frame;
    if (!match) seen |= FE_FOREIGN << 8;
    s.line = r.line_number;
    if (r.service_type != SDV_SRV_NO) {
        /* a service line is a cleared line that keeps its numbers (PCMLine::setServiceLine, pcmline.cpp:490-502) */
        if (match && r.service_type == SDV_SRV_NEW_FILE) seen |= FF_NEW_FILE;
        if (match && r.service_type == SDV_SRV_END_FILE) seen |= FF_END_FILE;
        s.fl = (uint8_t)(SF_CTRL | (match ? SF_MATCH : 0) | (r.service_type != SDV_SRV_FILLER ? SF_SKIP : 0));
        if (r.service_type == SDV_SRV_FILLER) s.part = PART_FILLER;
        return s;
    }
    s.w[0] = r.words[0]; s.w[1] = r.words[1]; s.w[2] = r.words[2];
    const bool ci = r.calc_crc == r.words[3], crc = ci && !(r.flags & SDV_LF_FORCED_BAD), bw = (r.flags & SDV_LF_BW_SET) != 0;
    const bool coords = r.data_start != -32768 && r.data_stop != 32767 && r.data_start < r.data_stop;
    s.fl = (uint8_t)((crc ? SF_CRC : 0) | (ci ? SF_CI : 0) | (bw ? SF_BW : 0) | ((coords && bw) ? SF_OKIGN : 0) | (r.control_bit ? SF_CTRL : 0) |
                     (r.picked_bits_right ? SF_PICKR : 0) | (match ? SF_MATCH : 0));
    s.part = r.line_part; s.ref = r.ref_level; s.pickl = r.picked_bits_left;
    return s;
}

struct RecSrc16 {
    const sdv_pcm16x0_bin_rec *carry; uint32_t n_carry; const sdv_pcm16x0_bin_rec *recs;
    __device__ inline const sdv_pcm16x0_bin_rec &at(uint32_t i) const { return i < n_carry ? carry[i] : recs[i - n_carry]; }
};
struct Cfg16 { uint8_t format, field_order, p_correction, ignore_crc, mask_seams, broke_mask; uint16_t sample_rate_preset; };

/* ---- segments (as for PCM-1: END_FRAME positions, file tags per segment) ------------------------------------------------ */
struct SegArgs16 { RecSrc16 src; uint32_t n_recs; uint8_t *svc; uint32_t *block_count; const uint32_t *block_ofs; uint32_t *seg_end; uint32_t n_seg; uint32_t *marks; uint32_t *stat; int write; };
enum { SEG_CHUNK16 = 1024 };
__device__ inline void seg_body(const SegArgs16 &a, uint32_t blk, int lane)
{
    const uint32_t lo = blk * SEG_CHUNK16;
    uint32_t hi = lo + SEG_CHUNK16; if (hi > a.n_recs) hi = a.n_recs;
    uint32_t cnt = 0;
    const uint32_t base = a.write ? a.block_ofs[blk] : 0u;
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
/* ---- PCM16X0DataBlock + PCM16X0Deinterleaver::processBlock on three sub-lines ----------------------------------------------- */
enum { L1 = 0, L2 = 1, L3 = 2, W_L = 0, W_R = 1, W_P = 2 };
enum { AUD_ORIG, AUD_FIX_P, AUD_BROKEN };
struct DiCfg { bool force_ecc_check, en_p_code, ignore_crc; };
/* the block as bit sets: bit 3 * sub-block + line */
struct Blk {
    uint16_t w[3][3];           /* [sub-block][line] */
    uint32_t crc, valid;        /* word_crc / word_valid, 9 bits */
    uint32_t pleft, pcrc;       /* picked_left / picked_crc per line, 3 bits */
    uint8_t state[3];
    bool even;
};
__device__ inline int word_line(const Blk &b, int blk, int word)       /* getWordToLine (pcm16x0datablock.cpp:1029-1155) */
{
    if (word == W_P) return L2;
    const bool l_first = ((blk & 1) != 0) != b.even;
    return (word == W_L) == l_first ? L1 : L3;
}
__device__ inline bool b_crc(const Blk &b, int blk, int line) { return (b.crc >> (3 * blk + line)) & 1u; }
__device__ inline bool b_val(const Blk &b, int blk, int line) { return (b.valid >> (3 * blk + line)) & 1u; }
__device__ inline uint16_t b_word(const Blk &b, int blk, int word) { return b.w[blk][word_line(b, blk, word)]; }
__device__ inline void b_fix(Blk &b, int blk, int word, uint16_t v) { const int l = word_line(b, blk, word); b.w[blk][l] = v; b.valid |= 1u << (3 * blk + l); }
__device__ inline void b_mark_bad(Blk &b, int blk, int line) { b.crc &= ~(1u << (3 * blk + line)); b.valid &= ~(1u << (3 * blk + line)); b.pleft &= ~(1u << line); }
__device__ inline void b_mark_broken(Blk &b, int blk)         /* blk = 3: all sub-blocks */
{
#pragma unroll
    for (int i = 0; i < 3; i++) if (blk >= 3 || i == blk) { b.crc &= ~(7u << (3 * i)); b.valid &= ~(7u << (3 * i)); b.state[i] = AUD_BROKEN; }
}
__device__ inline int b_err_total(const Blk &b, int blk) { return 3 - (int)__popc((b.crc >> (3 * blk)) & 7u); }
__device__ inline int b_err_audio(const Blk &b, int blk) { return 2 - (int)__popc((b.crc >> (3 * blk)) & 5u); }
__device__ inline int b_err_fixed_audio_all(const Blk &b) { return 6 - (int)__popc(b.valid & 0x16Du); }        /* lines 1 and 3 of the three sub-blocks */
__device__ inline bool b_valid_sub(const Blk &b, int blk) { return ((b.valid >> (3 * blk)) & 5u) == 5u; }
__device__ inline bool b_valid_all(const Blk &b) { return (b.valid & 0x16Du) == 0x16Du; }
__device__ inline bool b_broken_any(const Blk &b) { return b.state[0] == AUD_BROKEN || b.state[1] == AUD_BROKEN || b.state[2] == AUD_BROKEN; }
__device__ inline bool b_fixed_any(const Blk &b) { return b.state[0] == AUD_FIX_P || b.state[1] == AUD_FIX_P || b.state[2] == AUD_FIX_P; }
__device__ inline bool b_can_force(const Blk &b) { return !b_broken_any(b) && (b.crc & 0x1FFu) == 0x1FFu; }
__device__ inline bool b_silent(const Blk &b)
{
    return (b.w[0][L1] | b.w[0][L3] | b.w[1][L1] | b.w[1][L3] | b.w[2][L1] | b.w[2][L3]) == 0;
}
__device__ inline int b_picked_audio(const Blk &b, int blk)     /* getPickedAudioSamples (:458-475) */
{
    if (blk != 0) return 0;
    return (int)((b.pleft >> word_line(b, 0, W_L)) & 1u) + (int)((b.pleft >> word_line(b, 0, W_R)) & 1u);
}
__device__ inline bool b_picked_parity(const Blk &b, int blk) { return (blk == 0 && ((b.pleft >> L2) & 1u)) || ((b.pcrc >> L2) & 1u); }
__device__ inline void b_mark_unsafe(Blk &b)                   /* markAsUnsafe (:186-228) */
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const bool full_bad = !b_crc(b, i, L2) && b_err_audio(b, i) > 0;
        if (b.state[i] != AUD_BROKEN) {
            const uint32_t keep = full_bad ? 0u : ((b.crc >> (3 * i)) & 5u);
            b.valid = (b.valid & ~(5u << (3 * i))) | (keep << (3 * i));
            b.state[i] = AUD_ORIG;
        }
    }
}
/* processBlock (:128-708) on the lines l1, l2, l3 */
__device__ inline void process_block(const DiCfg &d, const Sub &l1, const Sub &l2, const Sub &l3, bool even_order, Blk &b)
{
    b.even = even_order; b.crc = b.valid = b.pleft = b.pcrc = 0;
    const Sub *ls[3] = { &l1, &l2, &l3 };
#pragma unroll
    for (int line = 0; line < 3; line++) {         /* setWordData (:711-787) */
        const Sub &l = *ls[line];
        const bool ok = d.ignore_crc ? (l.fl & SF_OKIGN) != 0 : (l.fl & SF_CRC) != 0;
#pragma unroll
        for (int blk = 0; blk < 3; blk++) { b.w[blk][line] = l.w[blk]; if (ok) { b.crc |= 1u << (3 * blk + line); b.valid |= 1u << (3 * blk + line); } }
        if (l.pickl != 0) b.pleft |= 1u << line;
        if (l.fl & SF_PICKR) b.pcrc |= 1u << line;
    }
    const uint8_t pick_cnt = (uint8_t)(l1.pickl + l2.pickl + l3.pickl);
    b.state[0] = b.state[1] = b.state[2] = AUD_ORIG;
    enum { STG_CRC_CHECK, STG_P_CORR, STG_BAD_BLOCK, STG_NO_CHECK, STG_DATA_OK, STG_CONVERT_MAX };
    for (int blk = 0; blk < 3; blk++) {
        int state = STG_CRC_CHECK, stage_count = 0;
        const int err_total = b_err_total(b, blk), err_audio = b_err_audio(b, blk);
        uint16_t pick_mask = 0;
        for (;;) {
            stage_count++;
            if (state == STG_CRC_CHECK) {
                if (err_total > 1) state = STG_BAD_BLOCK;
                else if (d.en_p_code) {
                    if (d.force_ecc_check) state = STG_P_CORR;
                    else if (err_total > 0) state = err_audio > 0 ? STG_P_CORR : STG_DATA_OK;
                    else state = STG_DATA_OK;
                } else {
                    if (err_audio > 0) state = STG_BAD_BLOCK;
                    else if (d.force_ecc_check) state = STG_NO_CHECK;
                    else state = STG_DATA_OK;
                }
            } else if (state == STG_P_CORR) {
                int bad_ptr = 64;
                if (!b_crc(b, blk, word_line(b, blk, W_L))) bad_ptr = W_L;
                else if (!b_crc(b, blk, word_line(b, blk, W_R))) bad_ptr = W_R;
                else if (!b_crc(b, blk, L2)) bad_ptr = W_P;
                if (bad_ptr != W_P) {
                    /* fixByP (:806-912) */
                    const uint16_t check = (uint16_t)(b.w[blk][L1] ^ b.w[blk][L2] ^ b.w[blk][L3]);
                    int fix;        /* 0 not needed, 1 broken, 2 done */
                    if (check == 0) { if (bad_ptr != 64) b_fix(b, blk, bad_ptr, b_word(b, blk, bad_ptr)); fix = 0; }
                    else if (bad_ptr == 64) fix = 1;
                    else if ((pick_mask & check) == 0) { b_fix(b, blk, bad_ptr, (uint16_t)(check ^ b_word(b, blk, bad_ptr))); fix = 2; }
                    else fix = 1;
                    if (fix == 1) {
                        const int picked = b_picked_audio(b, blk);
                        if (picked > 1) { b_mark_bad(b, blk, L1); b_mark_bad(b, blk, L3); state = STG_BAD_BLOCK; }
                        else if (picked == 1) {
                            if (b_picked_parity(b, blk)) { b_mark_bad(b, blk, L1); b_mark_bad(b, blk, L3); state = STG_BAD_BLOCK; }
                            else {
                                if ((b.pleft >> L1) & 1u) { b_mark_bad(b, blk, L1); state = STG_P_CORR; }
                                else if ((b.pleft >> L3) & 1u) { b_mark_bad(b, blk, L3); state = STG_P_CORR; }
                                else { state = STG_BAD_BLOCK; b_mark_broken(b, 3); }
                                if (pick_cnt > 0) {
                                    const uint16_t m = (uint16_t)(16 - pick_cnt);
                                    pick_mask = (uint16_t)((uint16_t)(1u << (m & 31)) - 1);     /* a shift count past the word wraps modulo 32 on the reference's x86 */
                                }
                            }
                        } else {
                            if (b_picked_parity(b, blk)) { b_mark_bad(b, blk, L2); state = STG_NO_CHECK; }
                            else { state = STG_BAD_BLOCK; b_mark_broken(b, blk); }
                        }
                    } else if (fix == 0) state = STG_DATA_OK;
                    else { state = STG_DATA_OK; b.state[blk] = AUD_FIX_P; }
                } else state = STG_NO_CHECK;
            } else break;
            if (stage_count > STG_CONVERT_MAX) break;
        }
    }
}

/* the burst bookkeeping of trySIPadding (:1173-1388) and tryEIPadding (:2426-2564) */
struct Bursts { uint16_t vc, sc, uc, bc, vm, sm, um, bm; };
__device__ inline void bursts_block(Bursts &u, const Blk &b, uint16_t max_sil, uint16_t max_unch)
{
    const bool silent = b_silent(b), can = b_can_force(b);
    if (b_valid_all(b) && !silent && can) u.vc++;
    else if (u.vc > u.vm) u.vm = u.vc;
    if (silent) { u.sc++; if (u.sc >= max_sil) u.vc = 0; }
    else { if (u.sc > u.sm) u.sm = u.sc; u.sc = 0; }
    if (!can || b_fixed_any(b)) { u.uc++; if (u.uc > max_unch) u.vc = 0; }
    else { if (u.uc > u.um) u.um = u.uc; u.uc = 0; }
    if (b_broken_any(b)) { u.bc++; if (u.bc >= MAX_BROKEN) u.vc = 0; }
    else { if (u.bc > u.bm) u.bm = u.bc; u.bc = 0; }
}
__device__ inline void bursts_end(Bursts &u)
{
    if (u.vc > u.vm) u.vm = u.vc;
    if (u.sc > u.sm) u.sm = u.sc;
    if (u.uc > u.um) u.um = u.uc;
    if (u.bc > u.bm) u.bm = u.bc;
}
/* The same bookkeeping for up to 64 blocks at once: bit i of V / S / U / B = what bursts_block would see of block i (valid_all && !silent && can /
 * silent / !can || fixed / broken), n = how many blocks the words hold.  Equivalent to n calls of bursts_block in bit order - the counters carry
 * from word to word - so a run of blocks is judged with a handful of mask operations instead of one pass through processBlock's bookkeeping per block.
 * Works on wave-uniform words (the EI sweep: a lane per block, the words are ballots) and on per-lane words (the SI sweep: a lane per padding) alike. */
struct BurstsW { uint32_t vc, sc, uc, bc, vm, sm, um, bm; };
__device__ inline uint64_t low_mask64(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }
__device__ inline uint32_t longest_run64(uint64_t x) { uint32_t n = 0; while (x) { x &= x << 1; n++; } return n; }
__device__ inline uint64_t runs_ge64(uint64_t w, uint32_t k)        /* the positions where a run of at least k ones (all of it inside the word) ends */
{
    if (k == 0) return ~0ull;
    if (k > 64u) return 0ull;
    uint64_t res = ~0ull, p = w; uint32_t len = 0, plen = 1;
    while (k) {
        if (k & 1u) { res &= len ? (p << len) : p; len += plen; }
        k >>= 1;
        if (k) { p &= p << plen; plen <<= 1; }
    }
    return res;
}
/* one of the run counters (silent / unchecked / broken) over a word: c = the run that reaches into the word from below, m = the longest run that has
 * ended so far; returns the positions at which the run (with what it brought along) is at least k long */
__device__ inline uint64_t bursts_run_word(uint32_t &c, uint32_t &m, uint64_t w, uint64_t mask_n, uint32_t n, uint32_t k)
{
    w &= mask_n;
    if (w == mask_n) {
        const uint64_t t = (c + n >= k) ? (mask_n & ~low_mask64(k > c ? k - c - 1u : 0u)) : 0ull;
        c += n;
        return t;
    }
    const uint64_t zeros = ~w & mask_n;
    const uint32_t lo = (uint32_t)__ffsll((unsigned long long)zeros) - 1u;          /* ones from bit 0 up */
    const uint32_t z = 63u - (uint32_t)__clzll((unsigned long long)zeros);          /* the highest zero */
    uint64_t t = (c + lo >= k) ? (low_mask64(lo) & ~low_mask64(k > c ? k - c - 1u : 0u)) : 0ull;
    if (c + lo > m) m = c + lo;
    const uint64_t rest = w & ~low_mask64(lo);
    const uint32_t inner = longest_run64(rest & low_mask64(z));
    if (inner > m) m = inner;
    t |= runs_ge64(rest, k) & mask_n;
    c = n - 1u - z;
    return t;
}
__device__ inline void bursts_word(BurstsW &u, uint64_t V, uint64_t S, uint64_t U, uint64_t B, uint32_t n, uint32_t max_sil, uint32_t max_unch)
{
    if (n == 0) return;
    const uint64_t mask_n = low_mask64(n);
    V &= mask_n;
    /* where the count of valid blocks starts over: a silent run of max_sil, an unchecked run past max_unch, any BROKEN block (MAX_BROKEN = 1) */
    uint64_t R = bursts_run_word(u.sc, u.sm, S, mask_n, n, max_sil);
    R |= bursts_run_word(u.uc, u.um, U, mask_n, n, max_unch + 1u);
    R |= bursts_run_word(u.bc, u.bm, B, mask_n, n, (uint32_t)MAX_BROKEN);
    /* the valid count is only looked at by a block that is not valid itself - before that block's own resets - and at the very end */
    if (V == 0) { if (u.vc > u.vm) u.vm = u.vc; if (R) u.vc = 0; return; }
    uint32_t pos = 0;
    while (R) {
        const uint32_t r = (uint32_t)__ffsll((unsigned long long)R) - 1u;
        const uint64_t seg = low_mask64(r + 1u) & ~low_mask64(pos), nv = ~V & seg;
        if (nv) { const uint32_t j = 63u - (uint32_t)__clzll((unsigned long long)nv), c = u.vc + (uint32_t)__popcll((unsigned long long)(V & seg & low_mask64(j))); if (c > u.vm) u.vm = c; }
        u.vc = 0; pos = r + 1u; R &= R - 1ull;
    }
    const uint64_t seg = mask_n & ~low_mask64(pos), nv = ~V & seg;
    if (nv) { const uint32_t j = 63u - (uint32_t)__clzll((unsigned long long)nv), c = u.vc + (uint32_t)__popcll((unsigned long long)(V & seg & low_mask64(j))); if (c > u.vm) u.vm = c; }
    u.vc += (uint32_t)__popcll((unsigned long long)(V & seg));
}
__device__ inline void bursts_end_w(BurstsW &u)
{
    if (u.vc > u.vm) u.vm = u.vc;
    if (u.sc > u.sm) u.sm = u.sc;
    if (u.uc > u.um) u.um = u.uc;
    if (u.bc > u.bm) u.bm = u.bc;
}
struct Stats { uint16_t valid, silent, unchecked, broken; };        /* FieldStitchStats without its index: the table slot is the index */
__device__ inline bool stats_less(const Stats &a, uint32_t ia, const Stats &b, uint32_t ib)      /* frametrimset.cpp:312-370 */
{
    if (a.broken != b.broken) return a.broken < b.broken;
    if (a.valid != b.valid) return a.valid > b.valid;
    if (a.unchecked != b.unchecked) return a.unchecked < b.unchecked;
    if (a.silent != b.silent) return a.silent < b.silent;
    return ia < ib;
}
__device__ inline uint8_t stats_verdict(const Stats &m, bool ei)     /* the return value of trySIPadding (:1528-1552) / tryEIPadding (:2621-2645) */
{
    if (m.unchecked > (ei ? MAX_UNCH_EI : MAX_UNCH_SI)) return DS_NO_PAD;
    if (m.valid == 0) return DS_NO_PAD;
    if (m.silent > (ei ? MAX_SIL_EI : MAX_SIL_SI)) return DS_SILENCE;
    if (m.broken >= MAX_BROKEN) return DS_BROKE;
    return DS_OK;
}

/* ---- what K-A leaves per frame --------------------------------------------------------------------------------------------- */
struct Ana16 {
    uint32_t frame, err;
    uint16_t top[2], bottom[2];             /* [0] odd lines, [1] even lines */
    uint16_t data[2], valid[2];             /* sub-lines in the field buffers, sub-lines with a valid CRC */
    int16_t zero_top[2], zero_bot[2];       /* findZeroControlBitOffset from the top (with the :1663-1688 step) / from the bottom */
    uint8_t iblk_top[2], iblk_bot[2];       /* estimateBlockNumber of those */
    uint8_t ref[2], marks, ran_ei;
    /* the sweep's winner (the front of the sorted paddings that share the fewest BROKEN blocks, :1869-1896 / :2799-2826): SI per field, EI [0] */
    Stats best[2]; uint16_t best_pad[2], best_min_broken[2]; uint8_t best_found[2]; uint8_t _pad[6];
    Stats st[MAX_PAD_EI];                   /* SI: [35 * field + padding]; EI: [padding], valid = 0 where the run did not start */
};
/* ... K-B ... */
struct Dec16 {
    uint16_t top_pad[2], bot_pad[2];        /* lines */
    uint16_t data[2];                       /* sub-lines taken from the field buffers (after cuts) */
    uint16_t cut[2];                        /* sub-lines cut from the top of the field buffers */
    uint16_t extra[2];                      /* lines of last-resort padding behind the [odd, even] field */
    uint8_t field_order, padding_ok, silence, err;
    uint16_t srate; uint8_t emph, code;     /* filled by K-D */
    uint32_t total;                         /* sub-lines fillFrameForOutput queues for this frame (1470 unless lines came with sub-lines missing or doubled) */
    uint16_t rem_in, n_it;                  /* filled by K-B': sub-lines already waiting in conv_queue, interleave-block rounds of performDeinterleave */
};
/* what the in-order part of K-B needs of a frame (written by K-A), 32 bytes, and what it decides */
struct Pick16 {
    uint64_t ok[2];                         /* paddings whose run ends in DS_RET_OK - SI: per field; EI: 81 bits over both words */
    uint8_t best_pad[2], sweep_lock[2];     /* the sweep's winner and whether the sweep locks on it (SI per field, EI [0]) */
    uint8_t elig[2];                        /* SI: the field is long enough for padding detection and P-code checks are on; EI [0]: the stages lead to findEIPadding */
    uint8_t marks, _pad[9];
};
static_assert(sizeof(Pick16) == 32, "Pick16 layout");
struct Choice16 { uint8_t mode[2], pad[2]; };     /* mode: 0 nothing locked, 1 the probable padding passed, 2 the sweep's winner */
/* ... K-C ... */
struct Ctrl16 { uint8_t even_order, emph, code, _pad; uint16_t rate; uint16_t _pad2; };
/* stream state of the stitcher (resetState :64-85) */
struct State16 {
    uint8_t pad_ring[STATS_DEPTH]; uint8_t emph_ring[STATS_DEPTH], code_ring[STATS_DEPTH]; uint16_t srate_ring[STATS_DEPTH];
    int32_t pad_pos, ctrl_pos;              /* slot the next push() overwrites */
    uint16_t f1_srate; uint8_t f1_emph, f1_code;
    uint32_t rem_n;                         /* sub-lines left in conv_queue behind the last frame (their content: the engine's remainder buffer) */
};

struct FrameArgs16s {
    RecSrc16 src; const uint32_t *seg_end; uint32_t n_seg, seg_base, n_batch; Cfg16 cfg;
    const uint32_t *marks; uint64_t *pair_ofs; uint32_t *frasm_ofs;      /* offsets: [k] written by K-B' of the batch that holds frame k, [k + 1] the running total */
    const Sub *rem_in; Sub *rem_out;        /* conv_queue's remainder in front of the batch's first frame / behind its last one (FRAME_SUBS entries each) */
    Ana16 *ana; Pick16 *pick; Choice16 *choice; Dec16 *dec; Ctrl16 *ctrl; Sub *fields;          /* per frame of the batch; fields: [frame][2][735] */
    State16 *state;
    sdv_sample_pair *out_pairs; uint64_t pairs_cap; sdv_frame_asm_pcm16x0 *out_frames; uint32_t frames_cap;
    uint32_t *stat;             /* [0] = OR of FE_*, [1] = first frame index with an error, [2] = file tags seen */
    /* the visualiser's feed (sdv_set_pcm16x0_stitch_block_output): the blocks next to the pairs; vblk_ofs like frasm_ofs, in blocks */
    uint32_t *vblk_ofs; sdv_pcm16x0_block_rec *out_blocks; uint64_t blocks_cap;
    /* ... and the assembled sub-lines (sdv_set_pcm16x0_stitch_line_output): records per frame, an END_FRAME record behind them; field_src: the record
     * every place of the field buffers was made from ([frame][2][735] like `fields`) */
    uint32_t *vline_ofs; sdv_pcm16x0_bin_rec *out_lines; uint64_t lines_cap; uint32_t *field_src;
    const State16 *state_snap;  /* EI: the stream's state when the call began (NULL: the analysis makes full padding tables) */
};

#ifndef SDV_P16_LDS_SUBS
#define SDV_P16_LDS_SUBS 1536
#endif
enum { LDS_SUBS = SDV_P16_LDS_SUBS };       /* a 525-line frame: 1470 sub-lines + service tags; longer segments are read from global memory */
enum { CLS_WORDS = 10,                      /* SI: block classes over the field positions 0..639 */
       SLOW_CAP = 512 };
struct AnaLds {
    Sub lines[LDS_SUBS];                    /* the frame's sub-lines in stream order; behind the field split: the two field buffers, [0..735) odd, [735..1470) even */
    union {
        struct {
            uint64_t cls[2][4][CLS_WORDS];  /* SI: [order of the block's first word: odd / even][valid, silent, unchecked, broken] one bit per field position the block starts at */
            Stats tab[MAX_PAD_SI][5];       /* SI: per padding, interleave blocks 1..5 */
        } si;
        struct {
            uint64_t m[MAX_PAD_EI][4];      /* EI: the same four questions for 64 consecutive blocks, under every padding */
            uint32_t slow[SLOW_CAP];        /* EI: (padding << 16 | block) of blocks that hold a sub-line the Bit Picker touched: decoded by processBlock itself */
        } ei;
    };
    int32_t uni[16];
};

/* What trySIPadding / tryEIPadding ask of a block (forced P-code check, :1203-1215 / :2456-2468) when none of its three sub-lines was touched by the Bit
 * Picker, in closed form (processBlock :128-708 with force_ecc_check and en_p_code, no picked bits): with all three lines good the block is BROKEN when a
 * P-code check fails and valid when none does; with one bad line that line is restored from the other two (or, the parity line, left alone) and the block
 * counts as unchecked; with more it is left as it is, unchecked as well.  Bits: 1 valid, 2 silent, 4 unchecked, 8 broken; 16: picked bits - ask processBlock. */
enum { CL_V = 1, CL_S = 2, CL_U = 4, CL_B = 8, CL_SLOW = 16 };
__device__ inline uint32_t classify_block(bool ignore_crc, const Sub &l1, const Sub &l2, const Sub &l3)
{
    if ((l1.pickl | l2.pickl | l3.pickl) != 0 || ((l1.fl | l2.fl | l3.fl) & SF_PICKR)) return CL_SLOW;
    const uint32_t okf = ignore_crc ? (uint32_t)SF_OKIGN : (uint32_t)SF_CRC;
    const bool ok1 = (l1.fl & okf) != 0, ok2 = (l2.fl & okf) != 0, ok3 = (l3.fl & okf) != 0;
    const uint32_t a = (uint32_t)(l1.w[0] | l1.w[1] | l1.w[2]), m = (uint32_t)(l2.w[0] | l2.w[1] | l2.w[2]), c = (uint32_t)(l3.w[0] | l3.w[1] | l3.w[2]);
    const int nb = (ok1 ? 0 : 1) + (ok2 ? 0 : 1) + (ok3 ? 0 : 1);
    if (nb == 0) {
        const bool brk = ((l1.w[0] ^ l2.w[0] ^ l3.w[0]) | (l1.w[1] ^ l2.w[1] ^ l3.w[1]) | (l1.w[2] ^ l2.w[2] ^ l3.w[2])) != 0, sil = (a | c) == 0;
        return (brk ? (uint32_t)(CL_U | CL_B) : (sil ? 0u : (uint32_t)CL_V)) | (sil ? (uint32_t)CL_S : 0u);
    }
    /* the audio words as they stand afterwards: a restored word is the XOR of the other two */
    bool sil;
    if (nb == 1 && !ok1) sil = (m | c) == 0;
    else if (nb == 1 && !ok3) sil = (a | m) == 0;
    else sil = (a | c) == 0;
    return (uint32_t)CL_U | (sil ? (uint32_t)CL_S : 0u);
}
__device__ inline uint32_t classify_slow(const DiCfg &d, const Sub &l1, const Sub &l2, const Sub &l3, bool even_order)
{
    Blk b;
    process_block(d, l1, l2, l3, even_order, b);
    const bool silent = b_silent(b), can = b_can_force(b);
    return ((b_valid_all(b) && !silent && can) ? (uint32_t)CL_V : 0u) | (silent ? (uint32_t)CL_S : 0u) | ((!can || b_fixed_any(b)) ? (uint32_t)CL_U : 0u) |
           (b_broken_any(b) ? (uint32_t)CL_B : 0u);
}
/* 35 bits of a bitmap from position q on */
__device__ inline uint64_t bits_from(const uint64_t *w, uint32_t q)
{
    const uint32_t i = q >> 6, sh = q & 63u;
    uint64_t x = w[i] >> sh;
    if (sh > 64u - SI_OFS && i + 1 < (uint32_t)CLS_WORDS) x |= w[i + 1] << (64u - sh);
    return x & ((1ull << SI_OFS) - 1ull);
}

__device__ inline uint32_t wave_max_u32(uint32_t v, int lane) { for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl((int)v, lane ^ d); v = o > v ? o : v; } return v; }
/* findZeroControlBitOffset (:868-1055) over a field: lane 0 */
template <typename F>
__device__ inline int16_t find_zero_ctrl(F field, uint16_t f_size, bool from_top)
{
    uint8_t best_cnt = 0, run_cnt = 0; int16_t best_ofs = 0, start;
    if (!from_top) {
        start = (int16_t)f_size; start++;
        while (start >= 3) {
            start = (int16_t)(start - 3);
            uint8_t zero_cnt = 0;
            for (int iblk = 0; iblk < IBLK_PF; iblk++) {
                const int16_t so = (int16_t)(start - iblk * SI_TRUE);
                if (so < 0) break;
                const Sub s = field((uint16_t)so);
                if (sub_part(s) != 1) { zero_cnt = 0; break; }
                if ((s.fl & SF_CRC) && !(s.fl & SF_CTRL)) zero_cnt++;
            }
            if (zero_cnt > best_cnt) { best_cnt = zero_cnt; best_ofs = (int16_t)(start - 1); }
            run_cnt++;
            if (run_cnt > (SI_OFS * 3 / 2)) break;
        }
    } else {
        start = 0; start++;
        while (start < ((int)f_size - 3)) {
            start = (int16_t)(start + 3);
            uint8_t zero_cnt = 0;
            for (int iblk = 0; iblk < IBLK_PF; iblk++) {
                const int16_t so = (int16_t)(start + iblk * SI_TRUE);
                if (so >= (int)f_size) break;
                const Sub s = field((uint16_t)so);
                if (sub_part(s) != 1) { zero_cnt = 0; break; }
                if ((s.fl & SF_CRC) && !(s.fl & SF_CTRL)) zero_cnt++;
            }
            if (zero_cnt > best_cnt) { best_cnt = zero_cnt; best_ofs = (int16_t)(start - 1); }
            run_cnt++;
            if (run_cnt > (SI_OFS * 3 / 2)) break;
        }
    }
    return best_cnt > 0 ? best_ofs : (int16_t)-1;
}
/* the same search with a lane per starting position (the reference tries at most 53 of them, three sub-lines apart, and keeps the first one with the
 * most cleared Control Bits) */
template <typename F>
__device__ inline int16_t find_zero_ctrl_wave(F field, uint16_t f_size, bool from_top, int lane)
{
    const int t = lane;
    int start; bool exists;
    if (!from_top) { exists = (int)f_size + 1 - 3 * t >= 3; start = (int)f_size + 1 - 3 * (t + 1); }
    else { exists = 1 + 3 * t < (int)f_size - 3; start = 1 + 3 * (t + 1); }
    exists = exists && t <= (SI_OFS * 3 / 2);
    uint32_t zero_cnt = 0;
    if (exists) {
        for (int iblk = 0; iblk < IBLK_PF; iblk++) {
            const int so = from_top ? start + iblk * SI_TRUE : start - iblk * SI_TRUE;
            if (from_top ? so >= (int)f_size : so < 0) break;
            const Sub s = field((uint16_t)so);
            if (sub_part(s) != 1) { zero_cnt = 0; break; }
            if ((s.fl & SF_CRC) && !(s.fl & SF_CTRL)) zero_cnt++;
        }
    }
    const uint32_t key = exists ? ((zero_cnt << 8) | (uint32_t)(63 - t)) : 0u;
    const uint32_t best = wave_max_u32(key, lane);
    if ((best >> 8) == 0) return (int16_t)-1;
    const int bt = 63 - (int)(best & 0xFF);
    const int bstart = from_top ? 1 + 3 * (bt + 1) : (int)f_size + 1 - 3 * (bt + 1);
    return (int16_t)(bstart - 1);
}
template <typename F>
__device__ inline uint8_t estimate_block_number(F field, uint16_t f_size, int16_t zero_ofs)    /* :1058-1126 */
{
    uint8_t out = IBLK_PF - 1;
    if (zero_ofs < (int)f_size) {
        if (zero_ofs < 0) out = 0;
        else {
            const uint16_t ln = field((uint16_t)zero_ofs).line;
            for (int k = 0; k <= 5; k++) if (ln < IBLK_DELIM + k * (2 * SI_OFS)) { out = (uint8_t)k; break; }
        }
    }
    return out;
}

__device__ inline uint32_t wave_min_u32(uint32_t v, int lane) { for (int d = 1; d < 64; d <<= 1) { const uint32_t o = (uint32_t)__shfl((int)v, lane ^ d); v = o < v ? o : v; } return v; }
/* the front of std::sort over the paddings that share the fewest BROKEN blocks and checked anything at all (:1869-1896, :2799-2826):
 * tab[0..n) with FieldStitchStats::operator<.  Returns false when there is no such padding. */
__device__ inline bool best_padding(const Stats *tab, int n, int lane, Stats &m, uint32_t &mi, uint16_t &min_broken)
{
    uint32_t brk = 0xFFFFFFFFu;
    for (int p = lane; p < n; p += 64) if (tab[p].broken < brk) brk = tab[p].broken;
    min_broken = (uint16_t)wave_min_u32(brk, lane);
    uint32_t hi = 0xFFFFFFFFu, lo = 0xFFFFFFFFu;
    for (int p = lane; p < n; p += 64) {
        const Stats t = tab[p];
        if (t.broken == min_broken && t.valid > 0) {
            const uint32_t h2 = ((uint32_t)(0xFFFF - t.valid) << 16) | t.unchecked, l2 = ((uint32_t)t.silent << 16) | (uint32_t)p;
            if (h2 < hi || (h2 == hi && l2 < lo)) { hi = h2; lo = l2; }
        }
    }
    const uint32_t bh = wave_min_u32(hi, lane);
    const uint32_t bl = wave_min_u32(hi == bh ? lo : 0xFFFFFFFFu, lane);
    if (bh == 0xFFFFFFFFu && bl == 0xFFFFFFFFu) return false;
    m.valid = (uint16_t)(0xFFFF - (bh >> 16)); m.unchecked = (uint16_t)(bh & 0xFFFF); m.silent = (uint16_t)(bl >> 16); m.broken = min_broken; mi = bl & 0xFFFF;
    return true;
}

/* K-A: one wave, one frame.  kLds: the frame's sub-lines are staged in LDS; otherwise every access compacts the record again. */
/* (the padding history in registers: the in-order pass K-B keeps it, the analysis asks it what a tape that plays will lock on) */
/* the value lane `idx` holds, idx the same in every lane: v_readlane_b32 (no trip through LDS) */
#ifdef SDV_EMU
__device__ inline uint32_t lane_read(uint32_t v, uint32_t idx) { return (uint32_t)__shfl((int)v, (int)idx); }
#else
__device__ inline uint32_t lane_read(uint32_t v, uint32_t idx) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)sdvs::uni(idx)); }
#endif
/* stats_padding (circarray<uint8_t, 65>, :243) in registers: lane i holds slot i (slot 64 is uniform), and a histogram of the values in
 * the ring - lane v counts value v and value 64 + v - so that getProbablePadding (:4368-4423) is one wave reduction */
struct PadHist { uint32_t ring_lo, ring64, hist0, hist1; int pos, nvalid; };
__device__ inline void hist_load(PadHist &h, const State16 &s, int lane)
{
    h.ring_lo = s.pad_ring[lane]; h.ring64 = s.pad_ring[64]; h.pos = s.pad_pos;
    uint32_t h0 = 0, h1 = 0; int nv = 0;
    for (int i = 0; i < STATS_DEPTH; i++) {
        const uint32_t v = s.pad_ring[i];
        nv += v != INVALID_PAD ? 1 : 0; h0 += v == (uint32_t)lane ? 1u : 0u; h1 += v == (uint32_t)lane + 64u ? 1u : 0u;
    }
    h.hist0 = h0; h.hist1 = h1; h.nvalid = nv;
}
__device__ inline void hist_store(const PadHist &h, State16 &s, int lane)
{
    s.pad_ring[lane] = (uint8_t)h.ring_lo;
    if (lane == 0) { s.pad_ring[64] = (uint8_t)h.ring64; s.pad_pos = h.pos; }
}
__device__ inline void hist_reset(PadHist &h) { h.ring_lo = h.ring64 = INVALID_PAD; h.hist0 = h.hist1 = 0; h.pos = 0; h.nvalid = 0; }    /* clearPadStats (:4349-4352) */
__device__ inline void push_padding(PadHist &h, uint32_t v, int lane)       /* updatePadStats(v, true) (:4355-4365) */
{
    const uint32_t in_lo = lane_read(h.ring_lo, (uint32_t)(h.pos < 64 ? h.pos : 0));
    const uint32_t old = h.pos < 64 ? in_lo : h.ring64;
    h.ring_lo = (h.pos < 64 && lane == h.pos) ? v : h.ring_lo;
    h.ring64 = h.pos < 64 ? h.ring64 : v;
    /* plain arithmetic on both counters (no conditional stores: they would be turned into a store through a selected address) */
    const uint32_t lane_u = (uint32_t)lane;
    h.hist0 = h.hist0 + (v == lane_u ? 1u : 0u) - ((old != INVALID_PAD && old == lane_u) ? 1u : 0u);
    h.hist1 = h.hist1 + (v == lane_u + 64u ? 1u : 0u) - ((old != INVALID_PAD && old == lane_u + 64u) ? 1u : 0u);
    h.nvalid += old == INVALID_PAD ? 1 : 0;
    h.pos = (h.pos + 1) % STATS_DEPTH;
}
__device__ inline uint8_t probable_padding(const PadHist &h, int lane)      /* the value that occurs most often, the smallest of them on a tie */
{
    if (h.nvalid == 0) return INVALID_PAD;
    /* the largest count, bit by bit (counts are below 128): lanes whose count lacks a bit that another one has drop out */
    uint64_t a0 = ~0ull, a1 = (1ull << (MAX_PAD_EI - 64)) - 1;
#pragma unroll
    for (int bit = 6; bit >= 0; bit--) {
        const uint64_t m0 = __ballot((h.hist0 >> bit) & 1u) & a0, m1 = __ballot((h.hist1 >> bit) & 1u) & a1;
        if (m0 | m1) { a0 = m0; a1 = m1; }
    }
    return a0 ? (uint8_t)(__ffsll((unsigned long long)a0) - 1) : (uint8_t)(64 + __ffsll((unsigned long long)a1) - 1);
}

template <bool kLds>
__device__ inline void analyse_body(const FrameArgs16s &a, uint32_t kb, int lane, uint32_t lo, uint32_t n, AnaLds &lds)
{
    const uint32_t k = a.seg_base + kb;
    const uint32_t frame = a.src.at(lo + n).frame_number;               /* the END_FRAME record */
    const Cfg16 cfg = a.cfg;
    const bool ei = cfg.format == SDV_P16_FORMAT_EI;
    const uint32_t n_scan = n < BUF_TRIM ? n : (uint32_t)BUF_TRIM;
    uint32_t seen = 0;
    auto line_at = [&](uint32_t i) -> Sub { if (kLds) return lds.lines[i]; uint32_t dummy = 0; return compact(a.src.at(lo + i), frame, dummy); };
    /* 1. stage */
    for (uint32_t c = 0; c < n_scan; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        if (i < n_scan) { const Sub s = compact(a.src.at(lo + i), frame, seen); if (kLds) lds.lines[i] = s; }
    }
    seen = (uint32_t)(__ballot(seen & FF_NEW_FILE) ? FF_NEW_FILE : 0) | (uint32_t)(__ballot(seen & FF_END_FILE) ? FF_END_FILE : 0) | (uint32_t)(__ballot(seen >> 8) ? (FE_FOREIGN << 8) : 0);
    __syncthreads();
    /* 2. findFrameTrim (:213-563).  The reference walks the records with a step that depends on what it finds (three sub-lines on from a line that
     * counted, else one).  On a stream of whole lines - every record either a service line or one of three consecutive sub-lines with parts
     * 0, 1, 2 - that walk looks at every line's first sub-line and nowhere else does anything, whatever it steps by; then its counts and its
     * first / last lines are ballots over the records.  Anything else (a sub-line missing or doubled, filler lines) takes the walk itself. */
    bool whole_lines = kLds;
    uint32_t good_par[2] = { 0, 0 };
    if (kLds) {
        bool bad = false;
        for (uint32_t c = 0; c < n_scan; c += 64) {
            const uint32_t i = c + (uint32_t)lane;
            bool counts = false; uint32_t par = 0;
            if (i < n_scan) {
                const Sub s0 = lds.lines[i];
                if (!(s0.fl & SF_SKIP)) {
                    const uint32_t part = s0.part;          /* (a filler carries PART_FILLER: not 0, 1 or 2) */
                    auto plain = [&](uint32_t j, uint32_t want) -> bool { if (j >= n_scan) return false; const Sub t = lds.lines[j]; return !(t.fl & SF_SKIP) && t.part == want; };
                    if (part == 0) bad = bad || !(plain(i + 1, 1) && plain(i + 2, 2));
                    else if (part == 1) bad = bad || !(i >= 1 && plain(i - 1, 0) && plain(i + 1, 2));
                    else if (part == 2) bad = bad || !(i >= 2 && plain(i - 2, 0) && plain(i - 1, 1));
                    else bad = true;
                    if (part == 0 && !bad && (s0.fl & SF_MATCH)) {
                        counts = ((s0.fl | lds.lines[i + 1].fl | lds.lines[i + 2].fl) & SF_CRC) != 0;
                        par = (s0.line % 2) == 0 ? 1u : 0u;
                    }
                }
            }
            good_par[0] += 3u * (uint32_t)__popcll(__ballot(counts && par == 0));
            good_par[1] += 3u * (uint32_t)__popcll(__ballot(counts && par == 1));
        }
        whole_lines = __ballot(bad) == 0;
    }
    if (whole_lines) {
        const bool skip_par[2] = { good_par[0] > (uint32_t)MIN_GOOD_SUB, good_par[1] > (uint32_t)MIN_GOOD_SUB };
        uint32_t first_i[2] = { 0xFFFFFFFFu, 0xFFFFFFFFu }, last_i[2] = { 0, 0 };      /* record index of the first line that counts; index + 1 of the last one */
        for (uint32_t c = 0; c < n_scan; c += 64) {
            const uint32_t i = c + (uint32_t)lane;
            bool hv = false; uint32_t par = 0;
            if (i < n_scan) {
                const Sub s0 = lds.lines[i];
                if (!(s0.fl & SF_SKIP) && s0.part == 0 && (s0.fl & SF_MATCH)) {
                    par = (s0.line % 2) == 0 ? 1u : 0u;
                    const uint8_t any = (uint8_t)(s0.fl | lds.lines[i + 1].fl | lds.lines[i + 2].fl);
                    hv = (any & (skip_par[par] ? SF_CI : SF_BW)) != 0;
                }
            }
#pragma unroll
            for (uint32_t p = 0; p < 2; p++) {
                const uint64_t m = __ballot(hv && par == p);
                if (m) {
                    if (first_i[p] == 0xFFFFFFFFu) first_i[p] = c + (uint32_t)__ffsll((unsigned long long)m) - 1u;
                    last_i[p] = c + 64u - (uint32_t)__clzll((unsigned long long)m);
                }
            }
        }
        if (lane == 0) {
            for (int p = 0; p < 2; p++) {
                lds.uni[p] = first_i[p] != 0xFFFFFFFFu ? (int32_t)lds.lines[first_i[p]].line : 0;
                lds.uni[2 + p] = (first_i[p] != 0xFFFFFFFFu && last_i[p] - 1u > first_i[p]) ? (int32_t)lds.lines[last_i[p] - 1u].line : 0;      /* the line that set the top does not set the bottom */
            }
        }
    } else if (lane == 0) {
        uint32_t i = 0, o_good = 0, e_good = 0;
        bool o_skip = false, e_skip = false, o_top = false, e_top = false;
        uint16_t top[2] = { 0, 0 }, bottom[2] = { 0, 0 };
        while (i < n_scan) {
            bool has_valid = false;
            const Sub s = line_at(i);
            if ((s.fl & SF_MATCH) && !(s.fl & SF_SKIP) && !(s.part & PART_FILLER)) {       /* a data line of this frame (a filler is a service line) */
                if ((i + 3) <= n_scan && s.part == 0) for (uint32_t q = 0; q < 3; q++) has_valid = has_valid || (line_at(i + q).fl & SF_CRC) != 0;
                if (has_valid) {
                    if ((s.line % 2) == 0) { e_good += 3; if (e_good > MIN_GOOD_SUB) e_skip = true; }
                    else { o_good += 3; if (o_good > MIN_GOOD_SUB) o_skip = true; }
                }
            }
            i += has_valid ? 3u : 1u;
        }
        bool subline_skip = false;
        i = 0;
        while (i < n_scan) {
            const Sub s = line_at(i);
            if (s.fl & SF_SKIP) { i++; continue; }
            bool has_valid = false;
            if (s.fl & SF_MATCH) {
                const int p = (s.line % 2) == 0 ? 1 : 0;
                const bool skip = p ? e_skip : o_skip;
                bool &tp = p ? e_top : o_top;
                const bool avail = (i + 3) <= n_scan && sub_part(s) == 0;
                if (avail) for (uint32_t q = 0; q < 3; q++) { const uint8_t fl = line_at(i + q).fl; has_valid = has_valid || (skip ? (fl & SF_CI) != 0 : (fl & SF_BW) != 0); }
                if (!tp) { if (has_valid) { top[p] = s.line; subline_skip = tp = true; } }
                else { if (!avail) subline_skip = false; if (has_valid) bottom[p] = s.line; }
            }
            i += subline_skip ? 3u : 1u;
        }
        lds.uni[0] = top[0]; lds.uni[1] = top[1]; lds.uni[2] = bottom[0]; lds.uni[3] = bottom[1];
    }
    __syncthreads();
    const uint32_t top[2] = { (uint32_t)lds.uni[0], (uint32_t)lds.uni[1] }, bottom[2] = { (uint32_t)lds.uni[2], (uint32_t)lds.uni[3] };
    /* 3. splitFrameToFields (:566-750): rank the sub-lines of either field, 735 at most */
    Sub *const fout = a.fields + (size_t)kb * (2 * SUBLINES_PF);
    uint32_t cnt[2] = { 0, 0 }, valid[2] = { 0, 0 }, refs_all[2] = { 0, 0 }, refs_ok[2] = { 0, 0 };
    const bool even_open = top[1] != bottom[1] || top[1] != 0;
    for (uint32_t c = 0; c < n_scan; c += 64) {
        const uint32_t i = c + (uint32_t)lane;
        bool in[2] = { false, false }; bool ok = false; uint32_t ref = 0;
        Sub sl = sub_empty();
        if (i < n_scan) {
            sl = line_at(i);
            if ((sl.fl & SF_MATCH) && !(sl.fl & SF_SKIP)) {
                const uint32_t ln = sl.line;
                in[0] = (ln & 1) != 0 && ln >= top[0] && ln <= bottom[0];
                in[1] = (ln & 1) == 0 && ln >= top[1] && ln <= bottom[1] && even_open;
                ok = (sl.fl & SF_CRC) != 0; ref = sl.ref;
            }
        }
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const uint64_t m = __ballot(in[p]);
            const uint32_t rank = cnt[p] + (uint32_t)__popcll(m & lanemask_lt(lane));
            const bool take = in[p] && rank < SUBLINES_PF;
            if (take) { fout[p * SUBLINES_PF + rank] = sl; refs_all[p] += ref; if (ok) refs_ok[p] += ref; }     /* the field buffers (global: K-B' .. K-E read them) */
            if (take && a.field_src) a.field_src[(size_t)kb * (2 * SUBLINES_PF) + p * SUBLINES_PF + rank] = lo + i;
            valid[p] += (uint32_t)__popcll(__ballot(take && ok));
            cnt[p] += (uint32_t)__popcll(m);
        }
    }
    uint32_t data[2], ref_level[2];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        if (cnt[p] > SUBLINES_PF) cnt[p] = SUBLINES_PF;
        for (int d = 1; d < 64; d <<= 1) { refs_all[p] += (uint32_t)__shfl((int)refs_all[p], lane ^ d); refs_ok[p] += (uint32_t)__shfl((int)refs_ok[p], lane ^ d); }
        data[p] = cnt[p];
        ref_level[p] = valid[p] > 0 ? (refs_ok[p] / valid[p]) & 0xFF : (cnt[p] > 0 ? (refs_all[p] / cnt[p]) & 0xFF : 0u);
    }
    __syncthreads();
    /* From here on only the field buffers are looked at: they take the place of the staged records in LDS (read back from global memory:
     * __syncthreads orders the wave's own stores before the loads) */
    for (uint32_t c = 0; c < 2 * SUBLINES_PF; c += 64) {
        const uint32_t u = c + (uint32_t)lane;
        if (u < 2 * SUBLINES_PF) { const uint32_t p = u >= SUBLINES_PF ? 1u : 0u; if (u - p * SUBLINES_PF < data[p]) lds.lines[u] = fout[u]; }
    }
    __syncthreads();
    auto fld_ref = [&](int p, uint32_t u) -> Sub & { return lds.lines[p * SUBLINES_PF + u]; };
    /* 4. prescanForFalsePosCRCs (:753-833): whole video lines whose only valid sub-line is one the Bit Picker completed */
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const uint32_t n_tri = data[p] / 3;
        uint32_t first_bad = n_tri;             /* the scan stops at the first triple that is not one video line */
        for (uint32_t c = 0; c < n_tri; c += 64) {
            const uint32_t t = c + (uint32_t)lane;
            bool bad = false;
            if (t < n_tri) bad = !(fld_ref(p, 3 * t).line == fld_ref(p, 3 * t + 1).line && fld_ref(p, 3 * t + 1).line == fld_ref(p, 3 * t + 2).line);
            const uint64_t m = __ballot(bad);
            if (m) { first_bad = c + (uint32_t)__ffsll((unsigned long long)m) - 1u; break; }
        }
        for (uint32_t c = 0; c < first_bad; c += 64) {
            const uint32_t t = c + (uint32_t)lane;
            if (t < first_bad) {
                Sub &s0 = fld_ref(p, 3 * t), &s1 = fld_ref(p, 3 * t + 1), &s2 = fld_ref(p, 3 * t + 2);
                const bool c0 = (s0.fl & SF_CRC) != 0, c1 = (s1.fl & SF_CRC) != 0, c2 = (s2.fl & SF_CRC) != 0;
                if ((c0 && !c1 && !c2 && s0.pickl != 0) || (!c0 && !c1 && c2 && (s2.fl & SF_PICKR))) {
                    s0.fl &= (uint8_t)~SF_CRC; s1.fl &= (uint8_t)~SF_CRC; s2.fl &= (uint8_t)~SF_CRC;
                    s0.part |= PART_FORCED; s1.part |= PART_FORCED; s2.part |= PART_FORCED;
                    Sub *g = fout + p * SUBLINES_PF + 3 * t;
                    g[0].fl = s0.fl; g[1].fl = s1.fl; g[2].fl = s2.fl; g[0].part = s0.part; g[1].part = s1.part; g[2].part = s2.part;
                }
            }
        }
    }
    __syncthreads();
    auto field_at = [&](int p, int u) -> Sub { return (u >= 0 && u < (int)data[p]) ? fld_ref(p, (uint32_t)u) : sub_empty(); };
    Ana16 *const out = &a.ana[kb];
    int ei_hint = -1;           /* EI: the one padding the table was made for (-1: all of them) */
    /* 5. the padding tables */
    const DiCfg pad_cfg = { true, true, ei ? cfg.ignore_crc != 0 : false };
    if (!ei) {
#pragma unroll 1
        for (int p = 0; p < 2; p++) {
            /* padding_queue under padding `pad`: pad lines of nothing, the field, nothing up to 735 sub-lines (:1613-1646, :1825-1833); the data block
             * of line li of interleave block iblk starts at field position q = li + 105 iblk - 3 pad - whatever the padding, a block that starts at
             * q is the sub-lines q, q + 35, q + 70, so every block the sweep looks at (35 paddings x 5 interleave blocks x 35 lines) is one of the
             * 640 x 2 (the order of its words alternates with li) that are judged here once, a lane each */
            for (int w = 0; w < CLS_WORDS; w++) {
                const int q = 64 * w + lane;
                const Sub l1 = field_at(p, q), l2 = field_at(p, q + SI_OFS), l3 = field_at(p, q + 2 * SI_OFS);
                uint32_t c0 = classify_block(pad_cfg.ignore_crc, l1, l2, l3), c1 = c0;
                if (c0 & CL_SLOW) { c0 = classify_slow(pad_cfg, l1, l2, l3, false); c1 = classify_slow(pad_cfg, l1, l2, l3, true); }
#pragma unroll
                for (int f = 0; f < 4; f++) {
                    const uint64_t m0 = __ballot((c0 >> f) & 1u), m1 = __ballot((c1 >> f) & 1u);
                    if (lane == 0) { lds.si.cls[0][f][w] = m0; lds.si.cls[1][f][w] = m1; }
                }
            }
            __syncthreads();
            for (int task0 = 0; task0 < MAX_PAD_SI * 5; task0 += 64) {
                const int task = task0 + lane;
                if (task < MAX_PAD_SI * 5) {
                    const int pad = task / 5, iblk = 1 + task % 5;
                    const uint32_t q0 = (uint32_t)(iblk * SI_TRUE - 3 * pad);
                    const uint64_t ev = 0x5555555555555555ull, od = ~ev;         /* line li of an interleave block: even_order = li odd */
                    uint64_t m[4];
#pragma unroll
                    for (int f = 0; f < 4; f++) m[f] = (bits_from(lds.si.cls[0][f], q0) & ev) | (bits_from(lds.si.cls[1][f], q0) & od);
                    BurstsW u = { 0, 0, 0, 0, 0, 0, 0, 0 };
                    bursts_word(u, m[0], m[1], m[2], m[3], SI_OFS, MAX_SIL_SI, MAX_UNCH_SI);
                    bursts_end_w(u);
                    Stats st; st.valid = (uint16_t)u.vm; st.silent = (uint16_t)u.sm; st.unchecked = (uint16_t)u.um; st.broken = (uint16_t)u.bm;
                    lds.si.tab[pad][iblk - 1] = st;
                }
            }
            __syncthreads();
            if (lane < MAX_PAD_SI) {       /* interleave blocks 1..5 share the worst BROKEN burst; the best of them stands for the padding (:1456-1514) */
                uint16_t top_broken = 0;
                for (int i = 0; i < 5; i++) if (lds.si.tab[lane][i].broken > top_broken) top_broken = lds.si.tab[lane][i].broken;
                Stats m = lds.si.tab[lane][0]; m.broken = top_broken; uint32_t mi = 1;
                for (int i = 1; i < 5; i++) { Stats c = lds.si.tab[lane][i]; c.broken = top_broken; if (stats_less(c, (uint32_t)(i + 1), m, mi)) { m = c; mi = (uint32_t)(i + 1); } }
                out->st[MAX_PAD_SI * p + lane] = m;
            }
            __syncthreads();
            {
                Stats m; m.valid = m.silent = m.unchecked = m.broken = 0; uint32_t mi = 0; uint16_t mb = 0;
                const bool found = best_padding(&out->st[MAX_PAD_SI * p], MAX_PAD_SI, lane, m, mi, mb);
                if (lane == 0) { out->best[p] = m; out->best_pad[p] = (uint16_t)mi; out->best_min_broken[p] = mb; out->best_found[p] = found ? 1 : 0; }
            }
        }
    } else {
        /* padding_queue = first field, `pad` lines of nothing, second field (:2708-2765); block i = lines i, i + 490, i + 980.  64 consecutive blocks
         * at a time, a lane per block: under every padding the four questions as ballots (a block's first line, and its second one while that
         * still lies in the first field, do not depend on the padding: fetched once per 64 blocks); then a lane per padding takes its counters
         * over the 64 blocks on the ballots. */
        const int p1 = cfg.field_order == ORDER_BFF ? 1 : 0, p2 = 1 - p1;
        const int c1 = (int)data[p1], c2 = (int)data[p2];
        auto n_blocks_of = [&](int pad) -> int { const int size = c1 + 3 * pad + c2; return size >= EI_TRUE ? size - 2 * EI_OFS - 1 : 0; };
        auto q_at = [&](int pad, int i) -> Sub { return i < c1 ? field_at(p1, i) : (i < c1 + 3 * pad ? sub_empty() : field_at(p2, i - c1 - 3 * pad)); };
        auto sweep = [&](int pad_lo, int pad_hi) {
        const int n_max = n_blocks_of(pad_hi);
        BurstsW u0 = { 0, 0, 0, 0, 0, 0, 0, 0 }, u1 = u0;      /* the counters of padding `lane` and padding `lane + 64` */
        for (int i0 = 0; i0 < n_max; i0 += 64) {
            const int i = i0 + lane;
            const bool fix1 = i < c1, fix2 = i + EI_OFS < c1;
            const Sub f1 = fix1 ? field_at(p1, i) : sub_empty(), f2 = fix2 ? field_at(p1, i + EI_OFS) : sub_empty();
            uint32_t n_slow = 0;
            auto flush_slow = [&]() {               /* the blocks with picked bits, side by side */
                __syncthreads();
                for (uint32_t k0 = 0; k0 < n_slow; k0 += 64) {
                    const uint32_t k = k0 + (uint32_t)lane;
                    if (k < n_slow) {
                        const uint32_t e = lds.ei.slow[k]; const int pad = (int)(e >> 16), bi = (int)(e & 0xFFFF);
                        const uint32_t c = classify_slow(pad_cfg, q_at(pad, bi), q_at(pad, bi + EI_OFS), q_at(pad, bi + 2 * EI_OFS), (bi & 1) != 0);
#pragma unroll
                        for (int f = 0; f < 4; f++) if ((c >> f) & 1u) atomicOr((unsigned long long *)&lds.ei.m[pad][f], 1ull << (bi & 63));
                    }
                }
                __syncthreads();
                n_slow = 0;
            };
#pragma unroll 1
            for (int pad = pad_lo; pad <= pad_hi; pad++) {
                const int nb = n_blocks_of(pad);
                uint32_t c = 0;
                if (i < nb) {
                    const Sub l1 = fix1 ? f1 : q_at(pad, i), l2 = fix2 ? f2 : q_at(pad, i + EI_OFS), l3 = q_at(pad, i + 2 * EI_OFS);
                    c = classify_block(pad_cfg.ignore_crc, l1, l2, l3);
                }
                const uint64_t sm = __ballot((c & CL_SLOW) != 0);
                if (sm) {
                    if (n_slow + 64u > (uint32_t)SLOW_CAP) flush_slow();
                    if (c & CL_SLOW) lds.ei.slow[n_slow + (uint32_t)__popcll(sm & lanemask_lt(lane))] = ((uint32_t)pad << 16) | (uint32_t)i;
                    n_slow += (uint32_t)__popcll(sm);
                }
#pragma unroll
                for (int f = 0; f < 4; f++) { const uint64_t m = __ballot((c >> f) & 1u); if (lane == 0) lds.ei.m[pad][f] = m; }
            }
            flush_slow();
            {
                const int nb0 = (lane >= pad_lo && lane <= pad_hi) ? n_blocks_of(lane) - i0 : 0, nb1 = (lane + 64 >= pad_lo && lane + 64 <= pad_hi) ? n_blocks_of(lane + 64) - i0 : 0;
                if (nb0 > 0) bursts_word(u0, lds.ei.m[lane][0], lds.ei.m[lane][1], lds.ei.m[lane][2], lds.ei.m[lane][3], (uint32_t)(nb0 < 64 ? nb0 : 64), MAX_SIL_EI, MAX_UNCH_EI);
                if (nb1 > 0) { const int pd = lane + 64; bursts_word(u1, lds.ei.m[pd][0], lds.ei.m[pd][1], lds.ei.m[pd][2], lds.ei.m[pd][3], (uint32_t)(nb1 < 64 ? nb1 : 64), MAX_SIL_EI, MAX_UNCH_EI); }
            }
            __syncthreads();
        }
        bursts_end_w(u0); bursts_end_w(u1);
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int pad = lane + 64 * half;
            if (pad >= pad_lo && pad <= pad_hi) {
                const BurstsW &u = half ? u1 : u0;
                Stats st; st.valid = (uint16_t)u.vm; st.silent = (uint16_t)u.sm; st.unchecked = (uint16_t)u.um; st.broken = (uint16_t)u.bm;
                if (n_blocks_of(pad) <= 0) { st.valid = 0; st.silent = st.unchecked = st.broken = 0xFF; }
                out->st[pad] = st;
            }
        }
        };
        /* A tape that plays locks on the padding its history holds, frame after frame, and asks nothing else of the sweep: with a history at hand when
         * the call began (a.state_snap) the table is made for that padding alone, and for all 81 only where it does not pass (or the frame carries a
         * file tag).  The in-order pass says when a frame so treated would have needed more - another probable padding than the one assumed -, and the
         * call is made again with full tables (pcm16_engine.inc). */
        int hint = -1;
        if (a.state_snap && !(seen & 3u)) {
            PadHist hs; hist_load(hs, *a.state_snap, lane);
            const uint32_t r = probable_padding(hs, lane);
            if (hs.nvalid == STATS_DEPTH && r < (uint32_t)MAX_PAD_EI) hint = (int)r;
        }
        if (hint >= 0) {
            if (lane < 64) { Stats z; z.valid = 0; z.silent = z.unchecked = z.broken = 0xFF; out->st[lane] = z; if (lane + 64 < MAX_PAD_EI) out->st[lane + 64] = z; }
            __syncthreads();
            sweep(hint, hint);
            __syncthreads();
            if (stats_verdict(out->st[hint], true) != DS_OK) hint = -1;
        }
        if (hint < 0) sweep(0, MAX_PAD_EI - 1);
        ei_hint = hint;
        __syncthreads();
        {
            Stats m; m.valid = m.silent = m.unchecked = m.broken = 0; uint32_t mi = 0; uint16_t mb = 0;
            const bool found = best_padding(out->st, MAX_PAD_EI, lane, m, mi, mb);
            if (lane == 0) { out->best[0] = m; out->best_pad[0] = (uint16_t)mi; out->best_min_broken[0] = mb; out->best_found[0] = found ? 1 : 0; out->best_found[1] = 0; }
        }
    }
    __syncthreads();
    {   /* what the in-order pass needs: which paddings pass, and whether the sweep locks */
        Pick16 pk; pk.ok[0] = pk.ok[1] = 0; pk.best_pad[0] = pk.best_pad[1] = 0; pk.sweep_lock[0] = pk.sweep_lock[1] = 0; pk.elig[0] = pk.elig[1] = 0;
        pk.marks = (uint8_t)(seen & 3u);
        for (int i = 0; i < 9; i++) pk._pad[i] = 0;
        if (!ei) {
#pragma unroll
            for (int p = 0; p < 2; p++) {
                pk.ok[p] = __ballot(lane < MAX_PAD_SI && stats_verdict(out->st[MAX_PAD_SI * p + (lane < MAX_PAD_SI ? lane : 0)], false) == DS_OK);
                const Stats m = out->best[p];
                pk.best_pad[p] = (uint8_t)out->best_pad[p];
                pk.sweep_lock[p] = out->best_found[p] && m.unchecked <= MAX_UNCH_SI && m.silent < MAX_SIL_SI;
                pk.elig[p] = data[p] >= MIN_FILL_SI && cfg.p_correction;
            }
        } else {
            const int p1 = cfg.field_order == ORDER_BFF ? 1 : 0;
            pk.ok[0] = __ballot(stats_verdict(out->st[lane], true) == DS_OK);
            pk.ok[1] = __ballot(lane + 64 < MAX_PAD_EI && stats_verdict(out->st[lane + 64 < MAX_PAD_EI ? lane + 64 : 0], true) == DS_OK);
            const Stats m = out->best[0];
            pk.best_pad[0] = (uint8_t)out->best_pad[0];
            pk.sweep_lock[0] = out->best_found[0] && m.unchecked <= MAX_UNCH_EI && m.silent < MAX_SIL_EI;
            pk.elig[0] = !((data[0] < MIN_FILL_EI && data[1] < MIN_FILL_EI) || (data[0] + data[1]) < (2 * MIN_FILL_EI)) && data[p1] >= MIN_FILL_EI && cfg.p_correction;
        }
        if (ei && ei_hint >= 0) { pk._pad[0] = 1; pk._pad[1] = (uint8_t)ei_hint; }      /* ok[] says something about this padding only; no sweep winner */
#ifdef SDV_EMU
        if (ei && ei_hint >= 0 && getenv("SDV_P16_HINT_SKEW")) pk.ok[0] = pk.ok[1] = 0;        /* test hook: the in-order pass must find the table wanting and the call run again */
#endif
        if (lane == 0) a.pick[kb] = pk;
    }
    /* 6. Control Bit offsets: the searches a lane per starting position, the rest on lane 0 */
    int16_t z_top[2], z_bot[2];
#pragma unroll 1
    for (int p = 0; p < 2; p++) {
        auto fld = [&](uint16_t u) -> Sub { return field_at(p, (int)u); };
        z_top[p] = find_zero_ctrl_wave(fld, (uint16_t)data[p], true, lane);
        z_bot[p] = find_zero_ctrl_wave(fld, (uint16_t)data[p], false, lane);
    }
    if (lane == 0) {
        out->frame = frame; out->err = seen >> 8; out->marks = (uint8_t)(seen & 3u);
#pragma unroll 1
        for (int p = 0; p < 2; p++) {
            auto fld = [&](uint16_t u) -> Sub { return field_at(p, (int)u); };
            const uint16_t cntp = (uint16_t)data[p];
            int16_t z = z_top[p];
            if (z >= 0 && (z + 3 + 1) < (int)cntp) { const Sub s = fld((uint16_t)(z + 4)); if ((s.fl & SF_CRC) && !(s.fl & SF_CTRL)) z = (int16_t)(z + 3); }
            out->zero_top[p] = z; out->iblk_top[p] = estimate_block_number(fld, cntp, z);
            const int16_t zb = z_bot[p];
            out->zero_bot[p] = zb; out->iblk_bot[p] = estimate_block_number(fld, cntp, zb);
            out->top[p] = (uint16_t)top[p]; out->bottom[p] = (uint16_t)bottom[p]; out->data[p] = cntp; out->valid[p] = (uint16_t)valid[p]; out->ref[p] = (uint8_t)ref_level[p];
        }
        out->ran_ei = ei ? 1 : 0;
    }
}

/* ---- K-B: the decisions, frames in order, one wave ------------------------------------------------------------------------------- */
/* cutFieldTop (:836-865) as far as the sizes go: the field buffer itself is read through `cut` */
__device__ inline void cut_field_top(uint16_t &f_size, uint16_t &cut, uint16_t cut_cnt)
{
    cut_cnt = (uint16_t)(cut_cnt * 3);
    if (cut_cnt > 0) { cut = (uint16_t)(cut + cut_cnt); f_size = (uint16_t)(f_size - cut_cnt); }
}
/* findSIPadding (:1557-2243) over the table of one field */
__device__ inline uint8_t find_si_padding(bool probable_ok, uint8_t probable_pad, const Cfg16 &cfg, bool best_found, const Stats &best, uint32_t best_pad, uint16_t best_min_broken, int16_t zero_ofs, uint8_t iblk_num, uint16_t &f_size, uint16_t &cut,
                                          uint16_t &top_padding, uint16_t &bottom_padding)
{
    uint8_t res = DS_NO_PAD;
    uint16_t count, pad_top = 0, pad_bottom = 0;
    int16_t last_ofs;
    bool lock = false;
    bottom_padding = 0;
    top_padding = (uint16_t)((SUBLINES_PF - f_size) / 3);
    if (f_size < MIN_FILL_SI) return DS_NO_DATA;
    count = f_size;
    pad_bottom = (uint16_t)(SUBLINES_PF - count);                   /* sub-lines the queue was filled up with at the bottom (:1628-1646) */
    if (cfg.p_correction) {
        if (probable_ok) {              /* the padding most frames before this one took also passes here (:1705-1746) */
            lock = true;
            pad_top = probable_pad;
            pad_bottom = pad_bottom >= pad_top ? (uint16_t)(pad_bottom - pad_top) : 0;
            res = DS_OK;
        }
        if (!lock) {                    /* the full sweep (:1800-2013) */
            const Stats m = best; const uint32_t mi = best_pad; const uint16_t min_broken = best_min_broken;
            if (best_found && m.unchecked <= MAX_UNCH_SI) {
                if (m.silent < MAX_SIL_SI) {
                    if (min_broken == 0) res = m.valid > MIN_VALID_SI ? DS_OK : DS_NO_PAD;
                    else res = DS_BROKE;
                    lock = true;
                    pad_top = (uint16_t)mi;
                    pad_bottom = pad_bottom >= pad_top ? (uint16_t)(pad_bottom - pad_top) : 0;
                } else res = DS_SILENCE;
            }
        }
    }
    if (lock) {
        last_ofs = (int16_t)(iblk_num * SI_OFS);
        if (last_ofs < (int)pad_top) {
            last_ofs = (int16_t)((iblk_num + 1) * SI_OFS);
            last_ofs = (int16_t)(last_ofs - pad_top);
            pad_top = 0;
            cut_field_top(f_size, cut, (uint16_t)last_ofs);
            count = f_size;
        } else if (last_ofs > (int)pad_top) {
            last_ofs = (int16_t)((iblk_num - 1) * SI_OFS);
            pad_top = (uint16_t)(pad_top + last_ofs);
        }
        pad_top = (uint16_t)(pad_top * 3);
        pad_bottom = (uint16_t)(SUBLINES_PF - pad_top);
        if (pad_bottom >= count) pad_bottom = (uint16_t)(pad_bottom - count);
        else { pad_bottom = (uint16_t)(count - pad_bottom); count = (uint16_t)(count - pad_bottom); pad_bottom = 0; }
    } else if (zero_ofs >= 0) {
        pad_top = pad_bottom = 0;
        last_ofs = (int16_t)(3 + iblk_num * SI_TRUE);
        last_ofs = (int16_t)(last_ofs - zero_ofs);
        if (last_ofs > 0) pad_top = (uint16_t)last_ofs;
        else if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); cut_field_top(f_size, cut, (uint16_t)(last_ofs / 3)); count = f_size; }
        last_ofs = (int16_t)pad_top;
        last_ofs = (int16_t)(last_ofs + count);
        last_ofs = (int16_t)(SUBLINES_PF - last_ofs);
        if (last_ofs > 0) pad_bottom = (uint16_t)last_ofs;
        else if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); count = (uint16_t)(count - last_ofs); }
    } else { pad_bottom = 0; pad_top = (uint16_t)(SUBLINES_PF - count); }
    top_padding = (uint16_t)(pad_top / 3);
    bottom_padding = (uint16_t)(pad_bottom / 3);
    f_size = count;
    return res;
}

/* conditionEIFramePadding (:2997-3464); zero1/zero2: findZeroControlBitOffset(from the bottom) of field1/field2, iblk2: estimateBlockNumber of field2's */
__device__ inline void condition_ei_frame_padding(int16_t zero1, int16_t zero2, uint8_t iblk2, uint16_t &f1_size, uint16_t &f2_size,
                                                  uint16_t &f1_top, uint16_t &f1_bottom, uint16_t &f2_top, uint16_t &f2_bottom)
{
    const uint16_t inter = f1_bottom;
    int16_t zero_ofs, last_ofs;
    bool pos_lock = false;
    zero_ofs = zero2;
    if (zero_ofs >= 0) {
        pos_lock = true;
        const uint8_t iblk_num = iblk2;
        zero_ofs = (int16_t)(f2_size - zero_ofs);
        last_ofs = (int16_t)((SI_OFS - 2) * 3 - zero_ofs);
        if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); f2_size = (uint16_t)(f2_size - last_ofs); }
        else if (last_ofs > 0) f2_bottom = (uint16_t)(f2_bottom + last_ofs / 3);
        last_ofs = (int16_t)((IBLK_PF - iblk_num - 1) * SI_TRUE);
        f2_bottom = (uint16_t)(f2_bottom + last_ofs / 3);
        last_ofs = (int16_t)(LINES_PF - f2_size / 3);
        last_ofs = (int16_t)(last_ofs - f2_bottom);
        if (last_ofs < 0) {
            last_ofs = (int16_t)(0 - last_ofs);
            zero_ofs = (int16_t)((last_ofs / SI_OFS) + 1);
            zero_ofs = (int16_t)(zero_ofs * SI_OFS);
            last_ofs = (int16_t)(f2_bottom - zero_ofs);
            if (last_ofs < 0) { f2_top = f2_bottom = 0; pos_lock = false; }
            else {
                f2_bottom = (uint16_t)last_ofs;
                last_ofs = (int16_t)(LINES_PF - f2_size / 3);
                last_ofs = (int16_t)(last_ofs - f2_bottom);
            }
        }
        if (last_ofs > (int)inter) {
            if ((last_ofs - (int)inter) < 2) { f2_top = inter; f2_bottom = (uint16_t)(f2_bottom + (last_ofs - inter)); }
            else { f2_top = f2_bottom = 0; pos_lock = false; }
        } else if (pos_lock) f2_top = (uint16_t)last_ofs;
    }
    if (pos_lock) {
        zero_ofs = (int16_t)(inter - f2_top);
        f1_bottom = (uint16_t)zero_ofs;
        zero_ofs = (int16_t)((f1_size + f2_size) / 3);
        zero_ofs = (int16_t)(zero_ofs + f1_bottom + f2_top);
        zero_ofs = (int16_t)(zero_ofs + f2_bottom);
        zero_ofs = (int16_t)((2 * LINES_PF) - zero_ofs);
        if (zero_ofs < 0) { f1_top = f1_bottom = f2_top = f2_bottom = 0; pos_lock = false; }
        else f1_top = (uint16_t)zero_ofs;
    }
    if (!pos_lock) {
        zero_ofs = zero1;
        if (zero_ofs >= 0) {
            pos_lock = true;
            const uint8_t iblk_cnt = (uint8_t)(zero_ofs / SI_TRUE);
            zero_ofs = (int16_t)(zero_ofs - iblk_cnt * SI_TRUE);
            zero_ofs = (int16_t)((SUBLINES_PF + 2 * 3) - zero_ofs);
            zero_ofs = (int16_t)(zero_ofs / 3);
            f1_top = (uint16_t)zero_ofs;
            zero_ofs = (int16_t)(LINES_PF - f1_top);
            zero_ofs = (int16_t)(zero_ofs - f1_size / 3);
            if (zero_ofs < 0) pos_lock = false;
            else {
                f1_bottom = (uint16_t)zero_ofs;
                zero_ofs = (int16_t)(inter - f1_bottom);
                if (zero_ofs < 0) pos_lock = false;
                else {
                    f2_top = (uint16_t)zero_ofs;
                    zero_ofs = (int16_t)(f2_size / 3);
                    zero_ofs = (int16_t)(zero_ofs + f2_top);
                    zero_ofs = (int16_t)(LINES_PF - zero_ofs);
                    if (zero_ofs < 0) { f2_bottom = 0; f2_size = (uint16_t)(f2_size - (0 - zero_ofs) * 3); }
                    else f2_bottom = (uint16_t)zero_ofs;
                }
            }
        }
    }
    if (!pos_lock) {
        zero_ofs = (int16_t)(inter / 2);
        f2_top = (uint16_t)zero_ofs;
        zero_ofs = (int16_t)(inter * 3);
        zero_ofs = (int16_t)(zero_ofs - f2_top * 3);
        f1_bottom = (uint16_t)(zero_ofs / 3);
        zero_ofs = (int16_t)(f1_size / 3);
        zero_ofs = (int16_t)(zero_ofs + f1_bottom);
        zero_ofs = (int16_t)(LINES_PF - zero_ofs);
        if (zero_ofs < 0) {
            f1_top = 0;
            zero_ofs = (int16_t)(f1_size / 3);
            zero_ofs = (int16_t)(LINES_PF - zero_ofs);
            f1_bottom = (uint16_t)zero_ofs;
            zero_ofs = (int16_t)(inter - f1_bottom);
            f2_top = (uint16_t)zero_ofs;
        } else f1_top = (uint16_t)zero_ofs;
        zero_ofs = (int16_t)(f2_size / 3);
        zero_ofs = (int16_t)(zero_ofs + f2_top);
        zero_ofs = (int16_t)(LINES_PF - zero_ofs);
        if (zero_ofs < 0) { f2_bottom = 0; f2_size = (uint16_t)(f2_size - (0 - zero_ofs) * 3); }
        else f2_bottom = (uint16_t)zero_ofs;
    }
}
/* findEIDataAlignment (:3467-3585) */
__device__ inline uint8_t find_ei_alignment(int16_t zero_bot, uint8_t iblk_num, uint16_t &f_size, uint16_t &cut, uint16_t &top_pad, uint16_t &bottom_pad)
{
    int16_t zero_ofs = zero_bot, last_ofs;
    if (zero_ofs < 0) return DS_NO_PAD;
    top_pad = bottom_pad = 0;
    zero_ofs = (int16_t)(f_size - zero_ofs);
    last_ofs = (int16_t)((SI_OFS - 2) * 3 - zero_ofs);
    if (last_ofs < 0) { last_ofs = (int16_t)(0 - last_ofs); f_size = (uint16_t)(f_size - last_ofs); }
    else if (last_ofs > 0) bottom_pad = (uint16_t)(bottom_pad + last_ofs / 3);
    last_ofs = (int16_t)((IBLK_PF - iblk_num - 1) * SI_TRUE);
    bottom_pad = (uint16_t)(bottom_pad + last_ofs / 3);
    last_ofs = (int16_t)(LINES_PF - f_size / 3);
    last_ofs = (int16_t)(last_ofs - bottom_pad);
    if (last_ofs < 0) {
        last_ofs = (int16_t)(0 - last_ofs);
        if (last_ofs < SI_OFS && last_ofs < (int)f_size) { cut_field_top(f_size, cut, (uint16_t)last_ofs); return DS_OK; }
        return DS_NO_PAD;
    }
    top_pad = (uint16_t)(top_pad + last_ofs);
    return DS_OK;
}

/* one frame's decisions: findSIDataAlignment (:2246-2377) / findEIFrameStitching (:3588-4115), then the sizes fillFrameForOutput (:4594-4697) adds */
template <int P1>      /* P1: the first field in playback order as an index into [odd, even] (compile-time, so that the small arrays stay in registers) */
__device__ inline void finish_frame(const Choice16 &ch, const Cfg16 &cfg, const Ana16 &an, Dec16 &d)
{
    uint16_t data[2] = { an.data[0], an.data[1] }, cut[2] = { 0, 0 }, top_pad[2] = { 0, 0 }, bot_pad[2] = { 0, 0 };
    bool padding_ok = false, silence = true;
    const uint8_t order = P1 == 1 ? ORDER_BFF : ORDER_TFF;
    if (cfg.format != SDV_P16_FORMAT_EI) {
        const uint8_t odd_res = find_si_padding(ch.mode[0] == 1, ch.pad[0], cfg, an.best_found[0] != 0, an.best[0], an.best_pad[0], an.best_min_broken[0], an.zero_top[0], an.iblk_top[0], data[0], cut[0], top_pad[0], bot_pad[0]);
        if (odd_res == DS_OK) { padding_ok = true; silence = false; } else { padding_ok = false; silence = odd_res == DS_SILENCE; }
        const uint8_t even_res = find_si_padding(ch.mode[1] == 1, ch.pad[1], cfg, an.best_found[1] != 0, an.best[1], an.best_pad[1], an.best_min_broken[1], an.zero_top[1], an.iblk_top[1], data[1], cut[1], top_pad[1], bot_pad[1]);
        if (even_res != DS_OK) { padding_ok = false; if (odd_res == DS_SILENCE) silence = true; }
    } else {
        enum { STG_TRY_PREVIOUS, STG_FULL_PREPARE, STG_INTERPAD, STG_ALIGN, STG_FB_CTRL_EST, STG_PAD_NO_GOOD, STG_PAD_OK, STG_PAD_MAX = 12 };
        constexpr int p1 = P1, p2 = 1 - P1;     /* first and second field in playback order as indices into [odd, even] */
        int state = STG_TRY_PREVIOUS, stage_count = 0;
        for (;;) {
            stage_count++;
            if (state == STG_TRY_PREVIOUS) {
                if (ch.mode[0] == 1) {              /* the padding most frames before this one took also passes here (:3637-3724) */
                    const uint8_t r = ch.pad[0];
                    bot_pad[p1] = r; top_pad[p2] = 0; silence = false; state = STG_ALIGN;
                } else state = STG_FULL_PREPARE;
            } else if (state == STG_FULL_PREPARE) {
                top_pad[0] = top_pad[1] = bot_pad[0] = bot_pad[1] = 0;
                if ((data[0] < MIN_FILL_EI && data[1] < MIN_FILL_EI) || (data[0] + data[1]) < (2 * MIN_FILL_EI)) state = STG_FB_CTRL_EST;
                else state = STG_INTERPAD;
            } else if (state == STG_INTERPAD) {
                if (data[p1] < MIN_FILL_EI) state = STG_FB_CTRL_EST;
                else {
                    /* findEIPadding (:2649-2994) */
                    uint8_t res = DS_NO_PAD, field_padding = 0; bool lock = false;
                    bot_pad[0] = bot_pad[1] = 0;
                    top_pad[0] = (uint16_t)((SUBLINES_PF - data[0]) / 3); top_pad[1] = (uint16_t)((SUBLINES_PF - data[1]) / 3);
                    if (cfg.p_correction) {
                        const Stats m = an.best[0]; const uint32_t mi = an.best_pad[0]; const uint16_t min_broken = an.best_min_broken[0];
                        if (an.best_found[0] && m.unchecked <= MAX_UNCH_EI) {
                            if (m.silent < MAX_SIL_EI) {
                                if (min_broken == 0) res = m.valid > MIN_VALID_EI ? DS_OK : DS_NO_PAD;
                                else res = DS_BROKE;
                                lock = true; field_padding = (uint8_t)mi;
                            } else res = DS_SILENCE;
                        }
                    }
                    if (lock) { bot_pad[p1] = field_padding; top_pad[p2] = 0; }
                    silence = false; padding_ok = false;
                    if (res == DS_OK) state = STG_ALIGN;
                    else { if (res == DS_SILENCE) silence = true; bot_pad[p1] = 0; state = STG_FB_CTRL_EST; }
                }
            } else if (state == STG_ALIGN) {
                condition_ei_frame_padding(an.zero_bot[p1], an.zero_bot[p2], an.iblk_bot[p2], data[p1], data[p2], top_pad[p1], bot_pad[p1], top_pad[p2], bot_pad[p2]);
                padding_ok = true; state = STG_PAD_OK;
            } else if (state == STG_FB_CTRL_EST) {
                state = STG_PAD_OK;
                if (find_ei_alignment(an.zero_bot[0], an.iblk_bot[0], data[0], cut[0], top_pad[0], bot_pad[0]) != DS_OK) {
                    bot_pad[0] = 0; top_pad[0] = (uint16_t)((SUBLINES_PF - data[0]) / 3); state = STG_PAD_NO_GOOD;
                }
                if (find_ei_alignment(an.zero_bot[1], an.iblk_bot[1], data[1], cut[1], top_pad[1], bot_pad[1]) != DS_OK) {
                    bot_pad[1] = 0; top_pad[1] = (uint16_t)((SUBLINES_PF - data[1]) / 3); state = STG_PAD_NO_GOOD;
                }
            } else break;
            if (stage_count > STG_PAD_MAX) break;
        }
    }
    /* fillFrameForOutput: what each field adds to conv_queue.  The reference counts in 16 bits; a shortfall is counted in sub-lines and
     * handed over as a count of lines (:4664-4673) */
    const uint32_t lines0 = data[0] <= SUBLINES_PF ? data[0] : 0u, lines1 = data[1] <= SUBLINES_PF ? data[1] : 0u;     /* addLinesFromField refuses more than the buffer holds (:4466) */
    const uint32_t added0 = 3u * top_pad[0] + lines0 + 3u * bot_pad[0], added1 = 3u * top_pad[1] + lines1 + 3u * bot_pad[1];
    const uint16_t extra0 = (uint16_t)added0 < SUBLINES_PF ? (uint16_t)(SUBLINES_PF - (uint16_t)added0) : 0, extra1 = (uint16_t)added1 < SUBLINES_PF ? (uint16_t)(SUBLINES_PF - (uint16_t)added1) : 0;
    d.extra[0] = extra0; d.extra[1] = extra1;
    const uint32_t total = added0 + 3u * extra0 + added1 + 3u * extra1;
    d.top_pad[0] = top_pad[0]; d.top_pad[1] = top_pad[1]; d.bot_pad[0] = bot_pad[0]; d.bot_pad[1] = bot_pad[1];
    d.data[0] = data[0]; d.data[1] = data[1]; d.cut[0] = cut[0]; d.cut[1] = cut[1];
    d.field_order = order; d.padding_ok = padding_ok; d.silence = silence;
    d.err = 0; d.total = total; d.rem_in = 0; d.n_it = 0;
}

/* K-B, in order: which padding every frame locks on.  One wave; lane j holds the Pick16 of frame c0 + j in registers and the loop
 * reads frame j's words with v_readlane, so that the per-frame steps are scalar */
__device__ inline void choose_body(const FrameArgs16s &a, int lane)
{
    PadHist st;
    hist_load(st, *a.state, lane);
    const bool ei = a.cfg.format == SDV_P16_FORMAT_EI;
    bool need_full = false;
    /* (the picks of the chunk behind the one being decided are on their way while it is: a chunk is a few hundred dependent instructions of one wave, and
     * with the load at its head every chunk began with a trip to memory - a third of the kernel's 117 us per 1 024 frames) */
    uint4 nq0 = { 0, 0, 0, 0 }, nq1 = { 0, 0, 0, 0 };
    if ((uint32_t)lane < a.n_batch) { const uint4 *src = (const uint4 *)&a.pick[lane]; nq0 = src[0]; nq1 = src[1]; }
    for (uint32_t c0 = 0; c0 < a.n_batch; c0 += 64) {
        const uint32_t nc = a.n_batch - c0 < 64 ? a.n_batch - c0 : 64u;
        const uint4 q0 = nq0, q1 = nq1;                                 /* ok[0], ok[1] | best_pad, sweep_lock, elig, marks */
        nq0 = uint4{ 0, 0, 0, 0 }; nq1 = uint4{ 0, 0, 0, 0 };
        if (c0 + 64u + (uint32_t)lane < a.n_batch) { const uint4 *src = (const uint4 *)&a.pick[c0 + 64u + (uint32_t)lane]; nq0 = src[0]; nq1 = src[1]; }
        uint32_t mine = 0;                      /* the Choice16 of frame c0 + lane, as a word */
        /* A tape that plays: the history holds one padding, 65 times over, and every frame's checks pass on it.  Then nothing of what the
         * frames do depends on their order - each locks on that padding (mode 1) and pushes it where it already is - and the 64 frames
         * of the chunk are decided at once, a lane each; anything else (a file tag, a field whose check fails on it, a history that is not
         * saturated) goes through the in-order loop below. */
        {
            const uint64_t s0 = __ballot(st.hist0 == (uint32_t)STATS_DEPTH), s1 = __ballot(st.hist1 == (uint32_t)STATS_DEPTH);
            const bool saturated = st.nvalid == STATS_DEPTH && ((s0 != 0) != (s1 != 0));
            const uint32_t P = s0 ? (uint32_t)__ffsll((unsigned long long)s0) - 1u : 64u + (uint32_t)(s1 ? __ffsll((unsigned long long)s1) - 1 : 0);
            const uint32_t marks_l = (q1.y >> 16) & 0xFF;
            bool fits = saturated && P < (ei ? (uint32_t)MAX_PAD_EI : (uint32_t)MAX_PAD_SI) && !(marks_l & (FF_NEW_FILE | FF_END_FILE));
            if ((q1.y >> 24) & 1u) fits = fits && P == (q1.z & 0xFFu);         /* (a table made for one padding answers for that padding only) */
            uint32_t word = 0, pushes = 0;
            const int n_fields = ei ? 1 : 2;
            for (int p = 0; p < n_fields; p++) {
                const uint32_t elig = (q1.y >> (8 * p)) & 0xFF;
                if (ei || elig) {
                    const uint32_t wi = ei ? (P >> 5) : (uint32_t)(2 * p) + (P >> 5);
                    const uint32_t w = wi == 0 ? q0.x : (wi == 1 ? q0.y : (wi == 2 ? q0.z : q0.w));
                    fits = fits && ((w >> (P & 31)) & 1u);
                    word |= (1u << (8 * p)) | (P << (16 + 8 * p));
                    pushes++;
                }
            }
            if (__ballot((uint32_t)lane < nc && !fits) == 0 && saturated) {
                uint32_t total = (uint32_t)lane < nc ? pushes : 0u;
                for (int d = 1; d < 64; d <<= 1) total += (uint32_t)__shfl((int)total, lane ^ d);
                st.pos = (int)(((uint32_t)st.pos + total) % (uint32_t)STATS_DEPTH);
                if ((uint32_t)lane < nc) *(uint32_t *)&a.choice[c0 + lane] = word;
                continue;
            }
        }
        /* SI with different paddings in the two fields: the history holds the two values A (field 0) and B (field 1) in turn.  As long as
         * every frame pushes A, then B - its field either passes on the probable padding when that is its own, or fails on the other one and
         * locks on the sweep's winner, which is its own - the counts of the two values before every push follow from the ring alone (one A
         * or B goes in, what sat 65 pushes back comes out), so the probable padding at each of the 64 pushes of 32 frames is known up
         * front, a lane checks its frame against it, and if all agree the 32 frames are decided at once. */
        bool sub_done[2] = { false, false };
        if (!ei) for (int sub = 0; sub < 2; sub++) {
            const uint32_t base = 32u * (uint32_t)sub;
            if (base >= nc) { sub_done[sub] = true; continue; }
            const uint32_t m = nc - base < 32u ? nc - base : 32u;
            const int pm2 = (st.pos + STATS_DEPTH - 2) % STATS_DEPTH, pm1 = (st.pos + STATS_DEPTH - 1) % STATS_DEPTH;
            const uint32_t A = pm2 < 64 ? lane_read(st.ring_lo, (uint32_t)pm2) : st.ring64, B = pm1 < 64 ? lane_read(st.ring_lo, (uint32_t)pm1) : st.ring64;
            const uint64_t mA_lo = __ballot(st.ring_lo == A), mAB_lo = __ballot(st.ring_lo == A || st.ring_lo == B);
            const bool mA_hi = st.ring64 == A, only_two = mAB_lo == ~0ull && (st.ring64 == A || st.ring64 == B);
            const uint32_t cA0 = (uint32_t)__popcll((unsigned long long)mA_lo) + (mA_hi ? 1u : 0u);
            const unsigned __int128 M = (unsigned __int128)mA_lo | ((unsigned __int128)(mA_hi ? 1u : 0u) << 64);
            const uint64_t R = (uint64_t)((M >> st.pos) | (M << (STATS_DEPTH - st.pos)));       /* bit t: the entry that push t of this run evicts is an A */
            const bool in = (uint32_t)lane >= base && (uint32_t)lane < base + m;
            const uint32_t i = (uint32_t)lane - base;
            auto low = [](uint32_t t) -> uint64_t { return t >= 64u ? ~0ull : ((1ull << t) - 1ull); };
            const uint32_t cA_t0 = cA0 + i - (uint32_t)__popcll((unsigned long long)(R & low(2u * i)));
            const uint32_t cA_t1 = cA0 + i + 1u - (uint32_t)__popcll((unsigned long long)(R & low(2u * i + 1u)));
            const bool r0A = cA_t0 * 2u > (uint32_t)STATS_DEPTH, r1A = cA_t1 * 2u > (uint32_t)STATS_DEPTH;    /* the probable padding is A (else B) */
            const uint64_t ok0 = (uint64_t)q0.x | ((uint64_t)q0.y << 32), ok1 = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
            const uint32_t best0 = q1.x & 0xFF, best1 = (q1.x >> 8) & 0xFF, lock0 = (q1.x >> 16) & 0xFF, lock1 = (q1.x >> 24) & 0xFF;
            const uint32_t elig0 = q1.y & 0xFF, elig1 = (q1.y >> 8) & 0xFF, marks_l = (q1.y >> 16) & 0xFF;
            const bool f0 = r0A ? ((ok0 >> (A & 63u)) & 1ull) : (!((ok0 >> (B & 63u)) & 1ull) && lock0 && best0 == A);
            const bool f1 = !r1A ? ((ok1 >> (B & 63u)) & 1ull) : (!((ok1 >> (A & 63u)) & 1ull) && lock1 && best1 == B);
            const bool fits = f0 && f1 && elig0 && elig1 && !(marks_l & (FF_NEW_FILE | FF_END_FILE));
            const bool all = st.nvalid == STATS_DEPTH && only_two && A != B && A < (uint32_t)MAX_PAD_SI && B < (uint32_t)MAX_PAD_SI && __ballot(in && !fits) == 0;
            if (!all) break;            /* the rest of the chunk goes through the loop below, in order */
            if (in) mine = (r0A ? 1u : 2u) | ((r1A ? 2u : 1u) << 8) | (A << 16) | (B << 24);
            /* the history after 2m pushes of A, B, A, B, ... */
            const int dl = (lane - st.pos + STATS_DEPTH) % STATS_DEPTH, d64 = (64 - st.pos + STATS_DEPTH) % STATS_DEPTH;
            if ((uint32_t)dl < 2u * m) st.ring_lo = (dl & 1) ? B : A;
            if ((uint32_t)d64 < 2u * m) st.ring64 = (d64 & 1) ? B : A;
            const uint32_t cA = cA0 + m - (uint32_t)__popcll((unsigned long long)(R & low(2u * m))), cB = (uint32_t)STATS_DEPTH - cA;
            st.hist0 = ((uint32_t)lane == A ? cA : 0u) + ((uint32_t)lane == B ? cB : 0u);
            st.hist1 = ((uint32_t)lane + 64u == A ? cA : 0u) + ((uint32_t)lane + 64u == B ? cB : 0u);
            st.pos = (int)(((uint32_t)st.pos + 2u * m) % (uint32_t)STATS_DEPTH);
            sub_done[sub] = true;
        }
        for (uint32_t j = 0; j < nc; j++) {
            if (sub_done[j >> 5]) continue;
            const uint32_t meta = lane_read(q1.x, j), meta2 = lane_read(q1.y, j);      /* best_pad[2] sweep_lock[2] | elig[2] marks */
            const uint32_t marks = (meta2 >> 16) & 0xFF;
            uint32_t word = 0;
            if (marks & FF_NEW_FILE) hist_reset(st);                        /* resetState ahead of the frame (:5768-5772) */
            if (marks & FF_END_FILE) hist_reset(st);                        /* ... and behind the END_FILE frame (:5822-5828) */
            else {
                const int n_fields = ei ? 1 : 2;
                for (int p = 0; p < n_fields; p++) {
                    const uint32_t elig = (meta2 >> (8 * p)) & 0xFF, lock = (meta >> (16 + 8 * p)) & 0xFF, best = (meta >> (8 * p)) & 0xFF;
                    uint32_t mode = 0, pad = 0;
                    if (ei || elig) {
                        const uint32_t r = probable_padding(st, lane);
                        bool ok = false;
                        if (r != INVALID_PAD) {
                            /* SI: ok[p] = words 2p, 2p + 1; EI: 81 bits over the four words */
                            const uint32_t wi = ei ? (r >> 5) : (uint32_t)(2 * p) + (r >> 5);
                            const uint32_t w = wi == 0 ? lane_read(q0.x, j) : (wi == 1 ? lane_read(q0.y, j) : (wi == 2 ? lane_read(q0.z, j) : lane_read(q0.w, j)));
                            ok = r < (ei ? (uint32_t)MAX_PAD_EI : (uint32_t)MAX_PAD_SI) && ((w >> (r & 31)) & 1u);
                        }
                        if (ok) { mode = 1; pad = r; }
                        else if (elig && lock) { mode = 2; pad = best; }
                        if (mode) push_padding(st, pad, lane);
                    }
                    word |= (mode << (8 * p)) | (pad << (16 + 8 * p));
                    if (((meta2 >> 24) & 1u) && mode != 1) need_full = true;        /* the frame's table was made for the padding the call began with: this frame needed more */
                }
            }
            if ((uint32_t)lane == j) mine = word;
        }
        if ((uint32_t)lane < nc) *(uint32_t *)&a.choice[c0 + lane] = mine;
    }
    hist_store(st, *a.state, lane);
    if (need_full && lane == 0) a.stat[3] = 1;          /* the call is made again with full tables (pcm16_engine.inc) */
}
/* K-B, the rest: a lane per frame */
__device__ inline void finish_body(const FrameArgs16s &a, uint32_t kb)
{
    if (kb >= a.n_batch) return;
    const Ana16 &an = a.ana[kb];
    Dec16 d;
    d.srate = 0; d.emph = d.code = 0;
    if (an.marks & FF_END_FILE) {
        d.top_pad[0] = d.top_pad[1] = d.bot_pad[0] = d.bot_pad[1] = d.data[0] = d.data[1] = d.cut[0] = d.cut[1] = d.extra[0] = d.extra[1] = 0;
        d.field_order = 0; d.padding_ok = d.silence = 0; d.err = 0; d.total = 0; d.rem_in = 0; d.n_it = 0;
    } else if (a.cfg.field_order == ORDER_BFF) finish_frame<1>(a.choice[kb], a.cfg, an, d);
    else finish_frame<0>(a.choice[kb], a.cfg, an, d);
    a.dec[kb] = d;
}

/* the padded frame as fillFrameForOutput queues it: sub-line `pos` of the 1470 */
__device__ inline Sub conv_at(const Dec16 &d, const Sub *fields, uint32_t pos)
{
    const int p_first = d.field_order == ORDER_BFF ? 1 : 0;
    for (int f = 0; f < 2; f++) {
        const int p = f == 0 ? p_first : 1 - p_first;
        const uint32_t n_top = 3u * d.top_pad[p], n_data = d.data[p] <= SUBLINES_PF ? d.data[p] : 0u, n_bot = 3u * d.bot_pad[p], n_extra = 3u * d.extra[p];
        if (pos < n_top) return sub_empty();
        pos -= n_top;
        if (pos < n_data) { const uint32_t u = pos + d.cut[p]; return u < SUBLINES_PF ? fields[p * SUBLINES_PF + u] : sub_empty(); }
        pos -= n_data;
        if (pos < n_bot + n_extra) return sub_empty();
        pos -= n_bot + n_extra;
    }
    return sub_empty();
}

/* conv_queue as frame kb finds it after fillFrameForOutput: what the frames before left behind (always a tail of the previous frame's
 * padded sub-lines - less than one interleave round of it stays, and every frame queues at least 1470), then its own padded sub-lines */
struct Queue16 {
    Dec16 d, dprev; const Sub *fields, *fields_prev, *rem; uint32_t rem_n, prev_from;
    __device__ inline Sub at(uint32_t q) const
    {
        if (q >= rem_n) return conv_at(d, fields, q - rem_n);
        return fields_prev ? conv_at(dprev, fields_prev, prev_from + q) : rem[q];
    }
};
__device__ inline Queue16 queue_of(const FrameArgs16s &a, uint32_t kb, const Dec16 &d)
{
    Queue16 q;
    q.d = d; q.fields = a.fields + (size_t)kb * (2 * SUBLINES_PF); q.rem_n = d.rem_in; q.fields_prev = NULL; q.rem = a.rem_in; q.prev_from = 0; q.dprev = d;
    if (d.rem_in != 0 && kb > 0) { q.dprev = a.dec[kb - 1]; q.fields_prev = q.fields - 2 * SUBLINES_PF; q.prev_from = q.dprev.total - d.rem_in; }
    return q;
}

/* ---- K-B': what waits in conv_queue (performDeinterleave :5216-5446 pops whole rounds - 105 sub-lines SI, 1470 EI - and leaves the rest) ------ */
__device__ inline void carry_body(const FrameArgs16s &a, int lane)
{
    const bool ei = a.cfg.format == SDV_P16_FORMAT_EI;
    const uint32_t lim = ei ? (uint32_t)FRAME_SUBS : (uint32_t)SI_TRUE, blk_it = ei ? (uint32_t)EI_OFS : (uint32_t)SI_OFS;
    uint32_t R = a.state->rem_n;
    uint64_t pbase = a.pair_ofs[a.seg_base]; uint32_t fbase = a.frasm_ofs[a.seg_base];
    uint32_t bbase = a.vblk_ofs ? a.vblk_ofs[a.seg_base] : 0u;
    uint32_t lbase = a.vline_ofs ? a.vline_ofs[a.seg_base] : 0u;
    for (uint32_t c0 = 0; c0 < a.n_batch; c0 += 64) {
        const uint32_t kb = c0 + (uint32_t)lane; const bool act = kb < a.n_batch;
        const uint32_t marks = act ? a.ana[kb].marks : 0u, S = act ? a.dec[kb].total : 0u;      /* the tags that carry the frame's own number: what its trim search saw (:300-330) */
        const bool endf = (marks & FF_END_FILE) != 0, newf = (marks & FF_NEW_FILE) != 0;       /* both empty the queue (resetState :72), END_FILE queues nothing */
        uint32_t f = (act && (endf || newf)) ? 1u : 0u, x = (!act || endf) ? 0u : S % lim;
        for (int d = 1; d < 64; d <<= 1) {              /* segmented sum of the frame sizes modulo a round */
            const int src = lane >= d ? lane - d : lane;
            const uint32_t xf = (uint32_t)__shfl((int)f, src), xv = (uint32_t)__shfl((int)x, src);
            if (lane >= d) { if (!f) x = (x + xv) % lim; f |= xf; }
        }
        const uint32_t r_out = f ? x : (x + R) % lim;
        const uint32_t r_prev = (uint32_t)__shfl((int)r_out, lane > 0 ? lane - 1 : 0);
        const uint32_t r_in = (newf || endf) ? 0u : (lane == 0 ? R : r_prev);
        const uint32_t n_it = endf ? 0u : (r_in + S) / lim;
        const uint32_t pairs = !act ? 0u : (endf ? 1u : 3u * blk_it * n_it + (newf ? 1u : 0u)), frasm = !act ? 0u : (endf ? 1u : 1u + (newf ? 1u : 0u));
        const uint32_t vblocks = (!act || endf) ? 0u : blk_it * n_it;
        const uint32_t vlines = (!act || endf) ? 0u : S + 1u;          /* what the frame queues and the END_FRAME record behind it */
        uint64_t ps = pairs; uint32_t fs = frasm, bs = vblocks, ls = vlines;
        for (int d = 1; d < 64; d <<= 1) {
            const int src = lane >= d ? lane - d : lane;
            const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)ps, src), hi = (uint32_t)__shfl((int)(uint32_t)(ps >> 32), src), of = (uint32_t)__shfl((int)fs, src);
            const uint32_t ob = (uint32_t)__shfl((int)bs, src), ol = (uint32_t)__shfl((int)ls, src);
            if (lane >= d) { ps += ((uint64_t)hi << 32) | lo; fs += of; bs += ob; ls += ol; }
        }
        if (act) {
            a.dec[kb].rem_in = (uint16_t)r_in; a.dec[kb].n_it = (uint16_t)n_it;
            a.pair_ofs[a.seg_base + kb] = pbase + ps - pairs; a.frasm_ofs[a.seg_base + kb] = fbase + fs - frasm;
            if (a.vblk_ofs) a.vblk_ofs[a.seg_base + kb] = bbase + bs - vblocks;
            if (a.vline_ofs) a.vline_ofs[a.seg_base + kb] = lbase + ls - vlines;
        }
        const uint32_t last = (a.n_batch - c0 < 64u ? a.n_batch - c0 : 64u) - 1u;
        R = (uint32_t)__shfl((int)r_out, (int)last);
        pbase += ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(ps >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)ps, 63);
        fbase += (uint32_t)__shfl((int)fs, 63);
        bbase += (uint32_t)__shfl((int)bs, 63);
        lbase += (uint32_t)__shfl((int)ls, 63);
    }
    if (lane == 0) { a.pair_ofs[a.seg_base + a.n_batch] = pbase; a.frasm_ofs[a.seg_base + a.n_batch] = fbase; a.state->rem_n = R; if (a.vblk_ofs) a.vblk_ofs[a.seg_base + a.n_batch] = bbase; if (a.vline_ofs) a.vline_ofs[a.seg_base + a.n_batch] = lbase; }
    if (R != 0) {               /* the batch's last frame leaves sub-lines behind: a copy for the frame that follows it (next batch or next call) */
        const uint32_t kb = a.n_batch - 1;
        const Dec16 d = a.dec[kb];
        const Sub *fields = a.fields + (size_t)kb * (2 * SUBLINES_PF);
        for (uint32_t i = (uint32_t)lane; i < R; i += 64) a.rem_out[i] = conv_at(d, fields, d.total - R + i);
    }
}

/* ---- the assembled sub-lines for the visualiser: what performDeinterleave hands to newLineProcessed (:5196-5213) -------------------------------
 * The sub-lines fillFrameForOutput queued for the frame (:4594-4690) as records of the binarizer's own type, so that the lines window draws them
 * (RenderPCM::renderNewLine(PCM16X0SubLine) is the renderer of both windows): a line of a field buffer is its source record with the queue order
 * addLinesFromField gave it (:4470) and the stitcher's forced-bad mark (:800-820); a padding line is a cleared sub-line with the frame's number,
 * the line number addFieldPadding counted to (:4552-4556: on from the last PART_RIGHT sub-line copied, or from 1 / 2) and its part. */
__device__ inline sdv_pcm16x0_bin_rec asm_blank_rec(uint32_t frame, uint16_t line)
{
    sdv_pcm16x0_bin_rec r;
    r.frame_number = frame; r.line_number = line;
    r.words[0] = r.words[1] = r.words[2] = 0; r.words[3] = (uint16_t)~0x0E10; r.calc_crc = 0x0E10;       /* PCM16X0SubLine::clear: silent words, the CRC of silence inverted */
    r.data_start = -32768; r.data_stop = 32767; r.queue_order = 0;
    r.black_level = r.white_level = r.ref_low = r.ref_level = r.ref_high = 0; r.hysteresis_depth = r.shift_stage = 0;
    r.service_type = SDV_SRV_NO; r.picked_bits_left = r.picked_bits_right = 0; r.flags = 0; r.line_part = 0; r.control_bit = 1; r._pad = 0;
    return r;
}
__device__ inline void emit_lines_body(const FrameArgs16s &a, uint32_t kb, int lane)
{
    if (!a.out_lines) return;
    const Dec16 d = a.dec[kb];
    const Ana16 *an = &a.ana[kb];
    if (an->marks & FF_END_FILE) return;            /* queues nothing */
    const uint32_t frame = an->frame;
    const uint64_t base = a.vline_ofs[a.seg_base + kb];
    const Sub *fields = a.fields + (size_t)kb * (2 * SUBLINES_PF);
    const uint32_t *fsrc = a.field_src + (size_t)kb * (2 * SUBLINES_PF);
    const int p_first = d.field_order == ORDER_BFF ? 1 : 0;
    uint32_t at = 0;
    for (int f = 0; f < 2; f++) {
        const int p = f == 0 ? p_first : 1 - p_first;
        const uint32_t n_top = 3u * d.top_pad[p], n_data = d.data[p] <= SUBLINES_PF ? d.data[p] : 0u, n_bot = 3u * ((uint32_t)d.bot_pad[p] + d.extra[p]);
        uint32_t last_line = (d.field_order == ORDER_TFF) == (f == 0) ? 1u : 2u;       /* getFirstFieldLineNum / getSecondFieldLineNum */
        for (uint32_t i = (uint32_t)lane; i < n_top; i += 64) {
            sdv_pcm16x0_bin_rec r = asm_blank_rec(frame, (uint16_t)(last_line + 2u * (i / 3u)));
            r.line_part = (uint8_t)(i % 3u); r.queue_order = (uint16_t)(1u + i);
            if (base + at + i < a.lines_cap) a.out_lines[base + at + i] = r;
        }
        last_line += 2u * d.top_pad[p]; at += n_top;
        /* the data: source records; the last PART_RIGHT one says where the numbering goes on */
        uint32_t right_at = 0xFFFFFFFFu;
        for (uint32_t c = 0; c < n_data; c += 64) {
            const uint32_t i = c + (uint32_t)lane;
            bool right = false;
            if (i < n_data) {
                const uint32_t u = i + d.cut[p];
                if (u < SUBLINES_PF) {
                    const Sub sb = fields[p * SUBLINES_PF + u];
                    sdv_pcm16x0_bin_rec r = a.src.at(fsrc[p * SUBLINES_PF + u]);
                    r.queue_order = (uint16_t)(1u + n_top + i);
                    if (sb.part & PART_FORCED) r.flags |= SDV_LF_FORCED_BAD;
                    const bool crc_now = r.service_type == SDV_SRV_NO && !(r.flags & SDV_LF_FORCED_BAD) && r.calc_crc == r.words[3];      /* isCRCValid() as it stands now */
                    r.flags = (uint8_t)((r.flags & ~SDV_LF_CRC_VALID) | (crc_now ? SDV_LF_CRC_VALID : 0));
                    right = r.line_part == 2;
                    if (base + at + i < a.lines_cap) a.out_lines[base + at + i] = r;
                } else {                /* (conv_at's answer past the buffer: a cleared sub-line) */
                    sdv_pcm16x0_bin_rec r = asm_blank_rec(0, 0); r.queue_order = (uint16_t)(1u + n_top + i);
                    if (base + at + i < a.lines_cap) a.out_lines[base + at + i] = r;
                }
            }
            const uint64_t m = __ballot(right);
            if (m) right_at = c + 63u - (uint32_t)__clzll((unsigned long long)m);
        }
        if (right_at != 0xFFFFFFFFu) last_line = (uint32_t)a.src.at(fsrc[p * SUBLINES_PF + right_at + d.cut[p]]).line_number + 2u;
        at += n_data;
        for (uint32_t i = (uint32_t)lane; i < n_bot; i += 64) {
            sdv_pcm16x0_bin_rec r = asm_blank_rec(frame, (uint16_t)(last_line + 2u * (i / 3u)));
            r.line_part = (uint8_t)(i % 3u); r.queue_order = (uint16_t)(1u + n_top + n_data + i);
            if (base + at + i < a.lines_cap) a.out_lines[base + at + i] = r;
        }
        at += n_bot;
    }
    if (lane == 0) {
        sdv_pcm16x0_bin_rec r = asm_blank_rec(frame, 0);
        r.calc_crc = 0; r.service_type = SDV_SRV_END_FRAME;       /* a service line: PCMLine::setServiceLine on a cleared sub-line */
        if (base + at < a.lines_cap) a.out_lines[base + at] = r;
    }
}

/* ---- K-C: collectCtrlBitStats (:4745-4912) ------------------------------------------------------------------------------------ */
__device__ inline void ctrl_body(const FrameArgs16s &a, uint32_t kb, int lane)
{
    const Dec16 d = a.dec[kb];
    const Queue16 queue = queue_of(a, kb, d);
    const int iblk = lane / 3, which = lane % 3;
    bool ok = false, zero = false;
    if (lane < 3 * 2 * IBLK_PF && !d.err && !(a.ana[kb].marks & FF_END_FILE)) {
        const uint32_t pos = (uint32_t)iblk * SI_TRUE + 1u + (which == 0 ? (uint32_t)BIT_EMPH : (which == 1 ? (uint32_t)BIT_RATE : (uint32_t)BIT_CODE));
        const Sub s = queue.at(pos);            /* from the front of the queue, whatever waits there (:4752-4792) */
        ok = (s.fl & SF_CRC) != 0; zero = ok && !(s.fl & SF_CTRL);
    }
    const uint64_t okm = __ballot(ok), zm = __ballot(zero);
    if (lane == 0) {
        uint64_t sel[3] = { 0, 0, 0 };
        for (int i = 0; i < 2 * IBLK_PF; i++) for (int w = 0; w < 3; w++) sel[w] |= 1ull << (3 * i + w);
        const int emph_cnt = __popcll(okm & sel[0]), rate_cnt = __popcll(okm & sel[1]), code_cnt = __popcll(okm & sel[2]);
        const int emph = __popcll(zm & sel[0]), rate = __popcll(zm & sel[1]), code = __popcll(zm & sel[2]);
        Ctrl16 c;
        c.emph = emph > emph_cnt / 2; c.rate = rate > rate_cnt / 2 ? 44100 : 44056; c.code = code > code_cnt / 2;
        c.even_order = emph_cnt >= 2 && rate_cnt >= 2 && code_cnt >= 2; c._pad = 0; c._pad2 = 0;
        a.ctrl[kb] = c;
    }
}

/* ---- K-D: Control Bit history (updateCtrlBitStats :4126-4166, fillFrameForOutput :4710-4741) ---------------------------------------- */
/* The three histories (circarray<.., 65>) sit in registers, slot i in lane i and slot 64 in a uniform value; the majority questions
 * (getProbableSampleRate :4289, getProbableEmphasesBit :4169, getProbableCodeBit :4229) are ballots.  One ring of packed entries:
 * bits 0-1 emphasis (0 unknown, 1 off, 2 on), 2-3 content (0 unknown, 1 audio, 2 code), 4-5 rate (0 unknown, 1 = 44056, 2 = 44100). */
enum { FLAG_CHUNK = 64 };
struct FlagLds { Ctrl16 c[FLAG_CHUNK]; uint8_t marks[FLAG_CHUNK]; };
__device__ inline uint32_t ctrl_pack(uint32_t emph, uint32_t code, uint32_t rate) { return emph | (code << 2) | (rate << 4); }
__device__ inline void flags_body(const FrameArgs16s &a, int lane, FlagLds &lds)
{
    State16 &st = *a.state;
    auto packed = [&](int i) -> uint32_t { return ctrl_pack(st.emph_ring[i], st.code_ring[i], st.srate_ring[i] == 44056 ? 1u : (st.srate_ring[i] == 44100 ? 2u : 0u)); };
    uint32_t ring_lo = packed(lane), ring64 = packed(64);
    int pos = st.ctrl_pos;
    uint32_t f1_srate = st.f1_srate, f1_emph = st.f1_emph, f1_code = st.f1_code;
    for (uint32_t c0 = 0; c0 < a.n_batch; c0 += FLAG_CHUNK) {
        const uint32_t nc = a.n_batch - c0 < FLAG_CHUNK ? a.n_batch - c0 : (uint32_t)FLAG_CHUNK;
        __syncthreads();
        Ctrl16 mine; mine.even_order = 0; mine.emph = mine.code = mine._pad = 0; mine.rate = 0; mine._pad2 = 0;
        uint32_t my_marks = 0;
        if ((uint32_t)lane < nc) { mine = a.ctrl[c0 + lane]; my_marks = a.ana[c0 + lane].marks; lds.c[lane] = mine; lds.marks[lane] = (uint8_t)my_marks; }
        /* A tape that plays: the Control Bits of every frame of the chunk read, no file tag among them.  Then no frame looks at the history - each takes
         * its own bits (:4716-4721) - and all the history sees is one push per frame: the chunk is decided at once, a lane per frame. */
        if (__ballot((uint32_t)lane < nc && (!mine.even_order || (my_marks & (FF_NEW_FILE | FF_END_FILE)))) == 0) {
            const uint32_t entry = ctrl_pack(mine.emph ? 2u : 1u, mine.code ? 2u : 1u, mine.rate == 44100 ? 2u : 1u);
            if ((uint32_t)lane < nc) { Dec16 &d = a.dec[c0 + lane]; d.srate = mine.rate; d.emph = mine.emph; d.code = mine.code; }
            /* slot s of the ring takes the push of frame (s - pos) mod 65 */
            const uint32_t j_lo = (uint32_t)((lane - pos + STATS_DEPTH) % STATS_DEPTH), j_64 = (uint32_t)((64 - pos + STATS_DEPTH) % STATS_DEPTH);
            const uint32_t e_lo = (uint32_t)__shfl((int)entry, (int)(j_lo < 64u ? j_lo : 0u)), e_64 = (uint32_t)__shfl((int)entry, (int)(j_64 < 64u ? j_64 : 0u));
            if (j_lo < nc) ring_lo = e_lo;
            if (j_64 < nc) ring64 = e_64;
            pos = (int)(((uint32_t)pos + nc) % (uint32_t)STATS_DEPTH);
            const uint32_t last = nc - 1u;
            f1_srate = (uint32_t)__shfl((int)(uint32_t)mine.rate, (int)last); f1_emph = (uint32_t)__shfl((int)(uint32_t)mine.emph, (int)last); f1_code = (uint32_t)__shfl((int)(uint32_t)mine.code, (int)last);
            continue;
        }
        __syncthreads();
        for (uint32_t j = 0; j < nc; j++) {
            const uint32_t marks = lds.marks[j];
            if (marks & (FF_NEW_FILE | FF_END_FILE)) { ring_lo = ring64 = 0; pos = 0; f1_srate = 44056; f1_emph = f1_code = 0; }   /* clearCtrlBitStats + resetState */
            if (marks & FF_END_FILE) continue;
            const Ctrl16 c = lds.c[j];
            const uint32_t entry = c.even_order ? ctrl_pack(c.emph ? 2u : 1u, c.code ? 2u : 1u, c.rate == 44100 ? 2u : 1u) : 0u;
            ring_lo = (pos < 64 && lane == pos) ? entry : ring_lo;
            ring64 = pos < 64 ? ring64 : entry;
            pos = (pos + 1) % STATS_DEPTH;
            if (c.even_order) { f1_srate = c.rate; f1_emph = c.emph; f1_code = c.code; }
            else {
                auto count = [&](uint32_t shift, uint32_t val) -> int { return __popcll(__ballot(((ring_lo >> shift) & 3u) == val)) + ((((ring64 >> shift) & 3u) == val) ? 1 : 0); };
                const int r056 = count(4, 1), r100 = count(4, 2), e_off = count(0, 1), e_on = count(0, 2), c_audio = count(2, 1), c_code = count(2, 2);
                f1_srate = (r056 > 0 || r100 > 0) ? (r056 < r100 ? 44100u : 44056u) : 44056u;
                f1_emph = (e_off > 0 || e_on > 0) ? (e_off < e_on ? 0u : 1u) : 1u;      /* the BIT (set = emphasis off) is what lands in f1_emph (:4733) */
                f1_code = (c_code > 0 || c_audio > 0) ? (c_code < c_audio ? 1u : 0u) : 1u;    /* the same for the code bit (set = audio) (:4734) */
            }
            if (lane == 0) { Dec16 &d = a.dec[c0 + j]; d.srate = (uint16_t)f1_srate; d.emph = (uint8_t)f1_emph; d.code = (uint8_t)f1_code; }
        }
    }
    __syncthreads();
    {
        const uint32_t e = ring_lo;
        st.emph_ring[lane] = (uint8_t)(e & 3u); st.code_ring[lane] = (uint8_t)((e >> 2) & 3u); st.srate_ring[lane] = ((e >> 4) & 3u) == 1 ? 44056 : (((e >> 4) & 3u) == 2 ? 44100 : 0);
        if (lane == 0) {
            st.emph_ring[64] = (uint8_t)(ring64 & 3u); st.code_ring[64] = (uint8_t)((ring64 >> 2) & 3u); st.srate_ring[64] = ((ring64 >> 4) & 3u) == 1 ? 44056 : (((ring64 >> 4) & 3u) == 2 ? 44100 : 0);
            st.ctrl_pos = pos; st.f1_srate = (uint16_t)f1_srate; st.f1_emph = (uint8_t)f1_emph; st.f1_code = (uint8_t)f1_code;
        }
    }
}

/* ---- K-E: performDeinterleave (:5165-5447) + outputDataBlock (:4973-5117) ----------------------------------------------------------- */
__device__ inline void frasm16_clear(sdv_frame_asm_pcm16x0 &f)     /* FrameAsmPCM16x0::clear (frametrimset.cpp:803-822) */
{
    f = sdv_frame_asm_pcm16x0();
    f.odd_bottom_data = f.even_bottom_data = 0xFFFF;
    f.flags = SDV_FA16_SILENCE;
}
__device__ inline void emit_body(const FrameArgs16s &a, uint32_t kb, int lane)
{
    const uint32_t k = a.seg_base + kb;
    const Ana16 &an = a.ana[kb];
    const Dec16 d = a.dec[kb];
    const Cfg16 cfg = a.cfg;
    const uint32_t marks = an.marks;
    const uint64_t pofs = a.pair_ofs[k];
    const uint32_t fofs = a.frasm_ofs[k];
    uint32_t err = an.err | d.err;
    if (marks & FF_END_FILE) {
        if (lane == 0) {
            if (fofs < a.frames_cap) { sdv_frame_asm_pcm16x0 s; frasm16_clear(s); s.service_type = SDV_PAIR_SRV_END_FILE; a.out_frames[fofs] = s; }
            if (pofs < a.pairs_cap) sdvp1::service_pair(&a.out_pairs[pofs], SDV_PAIR_SRV_END_FILE);
            if (err) { atomicOr(&a.stat[0], err); atomicMin(&a.stat[1], k); }
        }
        return;
    }
    if (err) { if (lane == 0) { atomicOr(&a.stat[0], err); atomicMin(&a.stat[1], k); } return; }
    uint64_t po = pofs;
    if (marks & FF_NEW_FILE) {
        if (lane == 0) {
            if (fofs < a.frames_cap) { sdv_frame_asm_pcm16x0 s; frasm16_clear(s); s.service_type = SDV_PAIR_SRV_NEW_FILE; a.out_frames[fofs] = s; }
            if (po < a.pairs_cap) sdvp1::service_pair(&a.out_pairs[po], SDV_PAIR_SRV_NEW_FILE);
        }
        po++;
    }
    const Queue16 queue = queue_of(a, kb, d);
    const bool ei = cfg.format == SDV_P16_FORMAT_EI;
    const uint32_t lim = ei ? (uint32_t)FRAME_SUBS : (uint32_t)SI_TRUE, blk_it = ei ? (uint32_t)EI_OFS : (uint32_t)SI_OFS;
    const uint32_t n_blocks = blk_it * d.n_it;          /* 490, unless the queue holds more or less than a frame (:5216) */
    const DiCfg di = { !cfg.ignore_crc, cfg.p_correction != 0, cfg.ignore_crc != 0 };
    const uint16_t rate = (cfg.sample_rate_preset == 44100 || cfg.sample_rate_preset == 44056) ? cfg.sample_rate_preset : d.srate;   /* setBlockSampleRate (:4915-4928) */
    const bool seam_mask = cfg.mask_seams && !d.padding_ok && !d.silence;
    uint32_t valid_cnt = 0; int32_t last_broken = -100000;
    uint32_t drop = 0, broken = 0, fix_p = 0, fix_bp = 0, samples_drop = 0;
    char *const base = (char *)(a.out_pairs + po);
    const uint64_t room = po >= a.pairs_cap ? 0 : a.pairs_cap - po;
    for (uint32_t c = 0; c < n_blocks; c += 64) {
        const uint32_t t = c + (uint32_t)lane;
        const bool act = t < n_blocks;
        Blk b;
        {
            const uint32_t tt = act ? t : 0u;
            const uint32_t i = tt % blk_it, p0 = (tt / blk_it) * lim + i;
            process_block(di, queue.at(p0), queue.at(p0 + blk_it), queue.at(p0 + 2 * blk_it), (i & 1u) != 0, b);
        }
        const bool nonsilent = act && !b_silent(b);
        /* seam masking: the first blocks of the frame, until three have checked out (:5256-5284) */
        const bool counts = nonsilent && b_valid_all(b) && !(((b.pleft | b.pcrc) & 7u) != 0);
        const uint64_t cm = __ballot(counts);
        const uint32_t incl = valid_cnt + (uint32_t)__popcll(cm & (lanemask_lt(lane) | (1ull << lane)));
        valid_cnt += (uint32_t)__popcll(cm);
        if (nonsilent && seam_mask && incl < 3) b_mark_unsafe(b);
        /* BROKEN masking: every block up to broke_mask - 1 blocks behind a BROKEN one (:5286-5320, :5407-5419) */
        const bool brk = nonsilent && cfg.broke_mask > 0 && b_broken_any(b);
        const uint64_t bm = __ballot(brk) & (lanemask_lt(lane) | (1ull << lane));
        const int32_t lb = bm ? (int32_t)(c + 63u - (uint32_t)__clzll((unsigned long long)bm)) : last_broken;
        {
            const uint64_t all = __ballot(brk);
            if (all) last_broken = (int32_t)(c + 63u - (uint32_t)__clzll((unsigned long long)all));
        }
        if (nonsilent && (int32_t)t - lb < (int32_t)cfg.broke_mask) b_mark_unsafe(b);
        if (act && a.out_blocks) {          /* newBlockProcessed (:5116): the block as it is now */
            const uint64_t bi = (uint64_t)a.vblk_ofs[k] + t;
            if (bi < a.blocks_cap) {
                sdv_pcm16x0_block_rec r;
#pragma unroll
                for (int s = 0; s < 3; s++) {
#pragma unroll
                    for (int l = 0; l < 3; l++) r.words[s][l] = b.w[s][l];
                    r.audio_state[s] = b.state[s];
                }
                r.word_crc = (uint16_t)(b.crc & 0x1FFu); r.word_valid = (uint16_t)(b.valid & 0x1FFu);
                r.picked_left = (uint8_t)(b.pleft & 7u); r.picked_crc = (uint8_t)(b.pcrc & 7u);
                r.flags = (uint8_t)((b.even ? SDV_P16B_EVEN_ORDER : 0) | (ei ? SDV_P16B_EI_FORMAT : 0) | (d.emph ? SDV_P16B_EMPHASIS : 0) | (d.code ? SDV_P16B_CODE : 0));
                r.sample_rate = rate; r._pad[0] = r._pad[1] = 0;
                a.out_blocks[bi] = r;
            }
        }
        if (act) {
            const bool all_valid = b_valid_all(b);
#pragma unroll
            for (int s = 0; s < 3; s++) {
                if (!b_valid_sub(b, s)) drop++;
                if (b.state[s] == AUD_BROKEN) broken++;
                if (b.state[s] == AUD_FIX_P) fix_p++;
                /* isDataFixedByBP (:542-562) */
                const bool pl = s == 0 && (((b.pleft >> L1) | (b.pleft >> L3)) & 1u), pc = (((b.pcrc >> L1) | (b.pcrc >> L3)) & 1u) || (((b.pcrc >> L2) & 1u) && b.state[s] == AUD_FIX_P);
                if (b_valid_sub(b, s) && (pl || pc)) fix_bp++;
            }
            if (!all_valid) samples_drop += (uint32_t)b_err_fixed_audio_all(b);
#pragma unroll
            for (int s = 0; s < 3; s++) {
                const int ll = word_line(b, s, W_L), lr = word_line(b, s, W_R);
                const bool ok = b.state[s] != AUD_BROKEN;
                const bool state = ok && all_valid;
                const uint32_t fl = (state ? (uint32_t)SDV_SF_BLOCK_OK : 0u) | ((ok && b_val(b, s, ll)) ? (uint32_t)SDV_SF_WORD_VALID : 0u) | ((state && b_crc(b, s, ll)) ? (uint32_t)SDV_SF_WORD_FIXED : 0u);
                const uint32_t fr = (state ? (uint32_t)SDV_SF_BLOCK_OK : 0u) | ((ok && b_val(b, s, lr)) ? (uint32_t)SDV_SF_WORD_VALID : 0u) | ((state && b_crc(b, s, lr)) ? (uint32_t)SDV_SF_WORD_FIXED : 0u);
                Pair3 p;
                p.a = (uint32_t)b.w[s][ll] | ((uint32_t)b.w[s][lr] << 16);
                p.b = fl | (fr << 8) | ((uint32_t)rate << 16);
                p.c = d.emph ? 1u : 0u;
                const uint32_t o = 3u * t + (uint32_t)s;
                if (o < room) store_pair((sdv_sample_pair *)(base + (size_t)o * 12u), p);
            }
        }
    }
    for (int dd = 1; dd < 64; dd <<= 1) {
        drop += (uint32_t)__shfl((int)drop, lane ^ dd); broken += (uint32_t)__shfl((int)broken, lane ^ dd); fix_p += (uint32_t)__shfl((int)fix_p, lane ^ dd);
        fix_bp += (uint32_t)__shfl((int)fix_bp, lane ^ dd); samples_drop += (uint32_t)__shfl((int)samples_drop, lane ^ dd);
    }
    if (lane == 0) {
        sdv_frame_asm_pcm16x0 f; frasm16_clear(f);
        f.frame_number = an.frame;
        f.odd_std_lines = f.even_std_lines = LINES_PF;
        f.odd_data_lines = (uint16_t)(d.data[0] / 3); f.even_data_lines = (uint16_t)(d.data[1] / 3);
        f.odd_valid_lines = (uint16_t)(an.valid[0] / 3); f.even_valid_lines = (uint16_t)(an.valid[1] / 3);
        f.odd_top_data = an.top[0]; f.odd_bottom_data = an.bottom[0]; f.even_top_data = an.top[1]; f.even_bottom_data = an.bottom[1];
        f.odd_sample_rate = f.even_sample_rate = rate;
        f.blocks_total = (uint16_t)(3u * n_blocks); f.blocks_drop = (uint16_t)drop; f.samples_drop = (uint16_t)samples_drop;
        f.odd_top_padding = d.top_pad[0]; f.odd_bottom_padding = d.bot_pad[0]; f.even_top_padding = d.top_pad[1]; f.even_bottom_padding = d.bot_pad[1];
        f.blocks_broken = (uint16_t)broken; f.blocks_fix_bp = (uint16_t)fix_bp; f.blocks_fix_p = (uint16_t)fix_p; f.blocks_fix_cwd = 0;
        f.field_order = d.field_order; f.odd_ref = an.ref[0]; f.even_ref = an.ref[1];
        f.flags = (uint8_t)(SDV_FA_ORDER_PRESET | (d.emph ? (SDV_FA1_ODD_EMPHASIS | SDV_FA1_EVEN_EMPHASIS) : 0) | (d.silence ? SDV_FA16_SILENCE : 0) |
                            (d.padding_ok ? SDV_FA16_PADDING_OK : 0) | (ei ? SDV_FA16_EI_FORMAT : 0));
        const uint32_t fo = fofs + ((marks & FF_NEW_FILE) ? 1u : 0u);
        if (fo < a.frames_cap) a.out_frames[fo] = f;
    }
}
} // namespace sdvp16

__global__ void __launch_bounds__(64) sdv_k_pcm16_segments(sdvp16::SegArgs16 a) { sdvp16::seg_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm16_analyse(sdvp16::FrameArgs16s a)
{
    __shared__ sdvp16::AnaLds lds;
    const uint32_t kb = blockIdx.x, k = a.seg_base + kb;
    const uint32_t lo = k == 0 ? 0u : a.seg_end[k - 1] + 1u, n = a.seg_end[k] - lo;
    if (n <= sdvp16::LDS_SUBS) sdvp16::analyse_body<true>(a, kb, (int)threadIdx.x, lo, n, lds);
    else sdvp16::analyse_body<false>(a, kb, (int)threadIdx.x, lo, n, lds);
}
__global__ void __launch_bounds__(64) sdv_k_pcm16_choose(sdvp16::FrameArgs16s a) { sdvp16::choose_body(a, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm16_finish(sdvp16::FrameArgs16s a) { sdvp16::finish_body(a, blockIdx.x * 64u + threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm16_carry(sdvp16::FrameArgs16s a) { sdvp16::carry_body(a, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm16_ctrl(sdvp16::FrameArgs16s a) { sdvp16::ctrl_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm16_flags(sdvp16::FrameArgs16s a)
{
    __shared__ sdvp16::FlagLds lds;
    sdvp16::flags_body(a, (int)threadIdx.x, lds);
}
__global__ void __launch_bounds__(64) sdv_k_pcm16_emit(sdvp16::FrameArgs16s a) { sdvp16::emit_body(a, blockIdx.x, (int)threadIdx.x); }
__global__ void __launch_bounds__(64) sdv_k_pcm16_emit_lines(sdvp16::FrameArgs16s a) { sdvp16::emit_lines_body(a, blockIdx.x, (int)threadIdx.x); }
#endif
