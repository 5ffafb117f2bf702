/*
 * stc007_deint_device.h - HIP device code of the STC-007 deinterleave + P/Q error-correction stage:
 * STC007Deinterleaver::processBlock (stc007deinterleaver.cpp:286-1123) with setWordData (:1126-1294),
 * fixByP (:1376-1464), fixByQ (:1468-2048) and the STC007DataBlock bookkeeping it drives
 * (stc007datablock.cpp:53-730).  One thread per data block: block s gathers word k from assembled line
 * s + 16k (stc007datablock.h:40-58), so consecutive threads read consecutive 24-byte line records.
 *
 * GF(2)^14 arithmetic of the b-adjacent Q code: T = multiply by x modulo x^14 + x^8 + 1 (what the reference's
 * T^1 row table encodes, stc007deinterleaver.cpp:8-11), so T^k / T^-k are k shift-and-fold steps;
 * (T^k + I)^-1, k = 1..5, are 14 row masks each, generated at compile time by Gauss-Jordan elimination and
 * applied as AND + parity.  The 21 (first_bad, second_bad) branches of fixByQ collapse to
 *   second == P :  e1 = T^-(6-first) Sq
 *   otherwise   :  e1 = (T^(second-first) + I)^-1 (T^-(6-second) Sq + Sp),  e2 = e1 + Sp.
 */
#ifndef SDV_STC007_DEINT_DEVICE_H
#define SDV_STC007_DEINT_DEVICE_H

#include "../../include/sdvpcm.h"
#ifndef SDV_EMU
#include <hip/hip_runtime.h>
#endif

namespace sdvd {

enum { WORD_P0 = 6, WORD_Q0 = 7, INTERLEAVE_OFS = 16, MIN_DEINT_DATA = 112 };
enum { STG_DATA_FILL = 0, STG_ERROR_CHECK, STG_TASK_SELECTION, STG_CWD_CORR, STG_P_CORR, STG_Q_CORR, STG_BAD_BLOCK, STG_NO_CHECK, STG_DATA_OK, STG_CONVERT_MAX };
enum { FIX_NOT_NEED = 0, FIX_SWITCH_P, FIX_BROKEN, FIX_NA, FIX_DONE };
enum { NO_ERR_INDEX = 64, MAX_PASSES = 3 };

constexpr uint16_t t_mul(uint16_t v) { return (uint16_t)(((v << 1) & 0x3FFF) ^ (((v >> 13) & 1) ? 0x0101 : 0)); }
constexpr uint16_t t_inv(uint16_t w) { return (uint16_t)((((w ^ ((w & 1) ? 0x0101 : 0)) >> 1) | ((w & 1) << 13)) & 0x3FFF); }
constexpr uint16_t t_pow(uint16_t v, int k)
{
    v &= 0x3FFF;
    for (; k > 0; k--) v = t_mul(v);
    for (; k < 0; k++) v = t_inv(v);
    return v;
}
struct TkiTables { uint16_t row[6][14]; };
constexpr TkiTables make_tki()
{
    TkiTables t{};
    for (int k = 1; k <= 5; k++) {
        uint16_t a[14] = {}, inv[14] = {};
        for (int i = 0; i < 14; i++) inv[i] = (uint16_t)(1u << i);
        for (int j = 0; j < 14; j++) {
            uint16_t col = (uint16_t)(t_pow((uint16_t)(1u << j), k) ^ (1u << j));
            for (int i = 0; i < 14; i++) if (col & (1u << i)) a[i] |= (uint16_t)(1u << j);
        }
        for (int c = 0; c < 14; c++) {
            int p = -1;
            for (int r = c; r < 14; r++) if (a[r] & (1u << c)) { p = r; break; }
            if (p < 0) continue;
            uint16_t x = a[p]; a[p] = a[c]; a[c] = x; x = inv[p]; inv[p] = inv[c]; inv[c] = x;
            for (int r = 0; r < 14; r++) if (r != c && (a[r] & (1u << c))) { a[r] ^= a[c]; inv[r] ^= inv[c]; }
        }
        for (int i = 0; i < 14; i++) t.row[k][i] = inv[i];
    }
    return t;
}
#ifdef SDV_EMU
static const TkiTables c_tki = make_tki();
#else
__device__ __constant__ const TkiTables c_tki = make_tki();
#endif

__device__ inline uint16_t tki_inv(uint16_t v, int k)
{
    uint16_t r = 0;
    for (int i = 0; i < 14; i++) r |= (uint16_t)((__popc((uint32_t)(c_tki.row[k][i] & v)) & 1) << i);
    return r;
}

/* STC007DataBlock in registers: three 8-bit masks instead of three bool[8]; the eight words packed into two 64-bit
 * values so that the error-position-dependent reads and writes of fixByP / fixByQ are shifts, not indexed memory */
struct Block {
    uint32_t w_frame[8]; uint16_t w_line[8];
    uint64_t wlo, whi;              /* words 0..3, 4..7: 16 bits each */
    uint8_t line_crc, cwd_fixed, word_valid, resolution, audio_state; bool cwd_applied;
    __device__ inline uint16_t w(int i) const { return (uint16_t)(((i & 4) ? whi : wlo) >> (16 * (i & 3))); }
    __device__ inline void setw(int i, uint16_t v)
    {
        const int sh = 16 * (i & 3);
        const uint64_t m = ~(0xFFFFull << sh), x = (uint64_t)v << sh;
        if (i & 4) whi = (whi & m) | x; else wlo = (wlo & m) | x;
    }
};
__device__ inline void blk_clear(Block &b)
{
    for (int i = 0; i < 8; i++) { b.w_frame[i] = 0; b.w_line[i] = 0; }
    b.wlo = b.whi = 0;
    b.line_crc = b.cwd_fixed = b.word_valid = 0; b.resolution = SDV_RES_14BIT; b.audio_state = SDV_AUD_ORIG; b.cwd_applied = false;
}
__device__ inline void blk_set_word(Block &b, int i, uint16_t w, bool line_valid, bool cwd_fixed)
{
    uint8_t m = (uint8_t)(1u << i);
    b.setw(i, w);
    b.line_crc = line_valid ? (b.line_crc | m) : (b.line_crc & ~m);
    b.word_valid = line_valid ? (b.word_valid | m) : (b.word_valid & ~m);
    b.cwd_fixed = cwd_fixed ? (b.cwd_fixed | m) : (b.cwd_fixed & ~m);
}
__device__ inline void blk_set_fixed(Block &b, int i) { b.word_valid |= (uint8_t)(1u << i); }
__device__ inline void blk_set_valid(Block &b, int i) { b.word_valid |= (uint8_t)(1u << i); b.cwd_fixed &= (uint8_t)~(1u << i); }
__device__ inline void blk_clear_cwd(Block &b, int i) { if (i < 8) b.cwd_fixed &= (uint8_t)~(1u << i); }
__device__ inline bool bit(uint8_t m, int i) { return (m >> i) & 1; }
__device__ inline uint8_t total_mask(const Block &b) { return b.resolution == SDV_RES_16BIT ? 0x7F : 0xFF; }
__device__ inline void blk_mark_broken(Block &b)
{
    uint8_t m = total_mask(b);
    b.word_valid &= (uint8_t)~m; b.line_crc &= (uint8_t)~m; b.cwd_fixed &= (uint8_t)~m;
    b.audio_state = SDV_AUD_BROKEN; b.cwd_applied = false;
}
__device__ inline uint16_t calc_p(const Block &b) { return (uint16_t)(b.w(0) ^ b.w(1) ^ b.w(2) ^ b.w(3) ^ b.w(4) ^ b.w(5)); }
__device__ inline uint16_t calc_q(const Block &b)
{
    /* Horner: T(T(T(T(T(T L0 + R0) + L1) + R1) + L2) + R2) = T^6 L0 + T^5 R0 + ... + T R2 (14-bit row masks: upper bits ignored) */
    uint16_t q = 0;
    for (int k = 0; k < 6; k++) q = t_mul((uint16_t)((q ^ b.w(k)) & 0x3FFF));
    return q;
}

/* where the assembled lines come from: a plain array (the deinterleave kernel) or one of the stitch stage's queues */
struct PtrSrc {
    const sdv_deint_line *p;
    __device__ inline const sdv_deint_line &line(size_t i) const { return p[i]; }
};

/* the eight lines of one block (line base + 16k -> slot k), gathered once so that a resolution re-try or the fast path
 * below does not go back to memory */
struct Lines8 {
    sdv_deint_line l[8];
    __device__ inline const sdv_deint_line &line(size_t i) const { return l[i / INTERLEAVE_OFS]; }
};
template <class Src>
__device__ inline void gather8(const Src &src, size_t base, Lines8 &out)
{
    for (int k = 0; k < 8; k++) out.l[k] = src.line(base + (size_t)INTERLEAVE_OFS * k);
}

template <class Src>
__device__ inline void set_word_data(const sdv_deint_settings &st, const Src &lines, size_t base, Block &b, uint8_t res)
{
    for (int k = 0; k < 8; k++) {
        const sdv_deint_line l = lines.line(base + (size_t)INTERLEAVE_OFS * k);
        bool bw_ok = (l.flags & SDV_DL_COORDS_BW_OK) != 0;
        bool ok = !st.ignore_crc ? bit(l.word_crc_ok, k) : bw_ok;
        bool cwd = (l.flags & SDV_DL_FIXED_BY_CWD) != 0;
        if (res == SDV_RES_14BIT) blk_set_word(b, k, l.words[k], ok, cwd);
        else if (k < 7) {
            bool sok = !st.ignore_crc ? bit(l.word_crc_ok, WORD_Q0) : bw_ok;
            uint16_t f1 = (uint16_t)(l.words[k] << 2), s = (uint16_t)((l.words[WORD_Q0] >> (12 - 2 * k)) & 3);
            blk_set_word(b, k, (uint16_t)(f1 + s), ok && sok, cwd);
        } else blk_set_word(b, WORD_Q0, 0, true, false);
        b.w_frame[k] = l.frame_number; b.w_line[k] = l.line_number;
    }
    b.resolution = res;
}

__device__ inline void recalc_p(Block &b)
{
    uint16_t p = calc_p(b);
    if (b.w(WORD_P0) != p) { blk_set_word(b, WORD_P0, p, bit(b.line_crc, WORD_P0), false); blk_set_fixed(b, WORD_P0); }
    else blk_set_valid(b, WORD_P0);
}
__device__ inline uint8_t fix_by_p(Block &b, uint8_t first_bad)
{
    b.audio_state = SDV_AUD_ORIG;
    uint16_t check = (uint16_t)(calc_p(b) ^ b.w(WORD_P0));
    if (check == 0) { if (first_bad != NO_ERR_INDEX) blk_set_valid(b, first_bad); return FIX_NOT_NEED; }
    if (first_bad == NO_ERR_INDEX) return FIX_BROKEN;
    uint16_t fix = (uint16_t)(check ^ b.w(first_bad));
    blk_set_word(b, first_bad, fix, false, bit(b.word_valid, first_bad));
    blk_set_fixed(b, first_bad);
    return FIX_DONE;
}
__device__ inline uint8_t fix_by_q(Block &b, uint8_t first_bad, uint8_t second_bad)
{
    uint16_t synd_p = 0, synd_q, e1 = 0, e2 = 0;
    b.audio_state = SDV_AUD_ORIG;
    if (second_bad == NO_ERR_INDEX && !bit(b.word_valid, WORD_P0)) second_bad = WORD_P0;
    synd_q = (uint16_t)(calc_q(b) ^ b.w(WORD_Q0));
    if (second_bad == WORD_P0) {
        if (synd_q == 0) { if (first_bad != NO_ERR_INDEX) blk_set_valid(b, first_bad); recalc_p(b); return FIX_NOT_NEED; }
    } else {
        synd_p = (uint16_t)(calc_p(b) ^ b.w(WORD_P0));
        if (synd_p == 0 && synd_q == 0) { blk_set_valid(b, first_bad); blk_set_valid(b, second_bad); return FIX_NOT_NEED; }
    }
    if (second_bad != WORD_P0 && !bit(b.word_valid, WORD_P0)) return FIX_NA;
    if (first_bad == NO_ERR_INDEX) return FIX_BROKEN;
    if (second_bad == NO_ERR_INDEX) return FIX_SWITCH_P;
    if (first_bad <= 5 && second_bad == WORD_P0) e1 = t_pow(synd_q, -(6 - (int)first_bad));
    else if (first_bad <= 4 && second_bad <= 5 && second_bad > first_bad) {
        e1 = (uint16_t)(t_pow(synd_q, -(6 - (int)second_bad)) ^ synd_p);
        e1 = tki_inv(e1, (int)second_bad - (int)first_bad);
        e2 = (uint16_t)(e1 ^ synd_p);
    } else return FIX_BROKEN;
    uint16_t old1 = b.w(first_bad);
    if (e1 != 0) { blk_set_word(b, first_bad, (uint16_t)(old1 ^ e1), false, bit(b.cwd_fixed, first_bad)); blk_set_fixed(b, first_bad); }
    else blk_set_valid(b, first_bad);
    uint16_t old2 = b.w(second_bad);
    if (second_bad == WORD_P0) e2 = (uint16_t)(old2 ^ calc_p(b));
    if (e2 != 0) { blk_set_word(b, second_bad, (uint16_t)(old2 ^ e2), false, bit(b.cwd_fixed, second_bad)); blk_set_fixed(b, second_bad); }
    else blk_set_valid(b, second_bad);
    return (e1 == 0 && e2 == 0) ? FIX_NOT_NEED : FIX_DONE;
}

/* the full state machine of STC007Deinterleaver::processBlock (process_block() below tries its two short cuts first) */
__device__ inline void process_block_fsm(const sdv_deint_settings &st, const Lines8 &lines, Block &out)
{
    const size_t base = 0;
    uint8_t run_res, stage_count = 0, fill_passes, all_errs = 0, aud_errs = 0, first_bad = NO_ERR_INDEX, second_bad = NO_ERR_INDEX, fix, state = STG_DATA_FILL;
    if (st.res_mode == SDV_RES_MODE_14BIT) { run_res = SDV_RES_14BIT; fill_passes = MAX_PASSES; }
    else if (st.res_mode == SDV_RES_MODE_14BIT_AUTO) { run_res = SDV_RES_14BIT; fill_passes = 0; }
    else if (st.res_mode == SDV_RES_MODE_16BIT_AUTO) { run_res = SDV_RES_16BIT; fill_passes = 0; }
    else { run_res = SDV_RES_16BIT; fill_passes = MAX_PASSES; }
    for (;;) {
        stage_count++;
        if (state == STG_DATA_FILL) {
            blk_clear(out);
            set_word_data(st, lines, base, out, run_res);
            out.audio_state = SDV_AUD_ORIG;
            fill_passes++;
            state = STG_ERROR_CHECK;
        } else if (state == STG_ERROR_CHECK) {
            first_bad = second_bad = NO_ERR_INDEX;
            for (uint8_t i = 0; i <= 5; i++)
                if (!bit(out.line_crc, i)) {
                    if (first_bad == NO_ERR_INDEX) first_bad = i;
                    else if (second_bad == NO_ERR_INDEX) { second_bad = i; break; }
                }
            aud_errs = (uint8_t)__popc((uint32_t)(~out.line_crc & 0x3F));
            all_errs = (uint8_t)__popc((uint32_t)(~out.line_crc & total_mask(out)));
            state = STG_TASK_SELECTION;
        } else if (state == STG_TASK_SELECTION) {
            state = STG_BAD_BLOCK;
            if (all_errs <= 2) {
                if (aud_errs == 0) state = !st.force_ecc_check ? STG_DATA_OK : (st.en_p_code ? STG_P_CORR : STG_NO_CHECK);
                else if (aud_errs == 1) { if (st.en_p_code) state = STG_P_CORR; }
                else if (aud_errs == 2) {
                    if (run_res == SDV_RES_14BIT) { if (st.en_q_code) state = STG_Q_CORR; }
                    else if (st.en_cwd && !(out.cwd_applied && out.cwd_fixed != 0)) state = STG_CWD_CORR;
                }
            } else if (st.en_cwd && !out.cwd_applied) state = STG_CWD_CORR;
        } else if (state == STG_CWD_CORR) {
            state = STG_BAD_BLOCK;
            if (out.cwd_fixed != 0) {
                out.word_valid |= out.cwd_fixed; out.cwd_applied = true;
                first_bad = second_bad = NO_ERR_INDEX;
                all_errs = (uint8_t)__popc((uint32_t)(~out.word_valid & 0xFF));
                aud_errs = 0;
                for (uint8_t i = 0; i <= 5; i++)
                    if (!bit(out.word_valid, i)) {
                        aud_errs++;
                        if (first_bad == NO_ERR_INDEX) first_bad = i;
                        else if (second_bad == NO_ERR_INDEX) second_bad = i;
                    }
                state = STG_TASK_SELECTION;
            }
        } else if (state == STG_P_CORR) {
            state = STG_BAD_BLOCK;
            if (bit(out.word_valid, WORD_P0)) {
                fix = fix_by_p(out, first_bad);
                if (fix == FIX_BROKEN) blk_mark_broken(out);
                else {
                    state = STG_DATA_OK;
                    blk_clear_cwd(out, first_bad);
                    if (fix == FIX_DONE) out.audio_state = SDV_AUD_FIX_P;
                    else if (fix == FIX_NOT_NEED && first_bad < WORD_P0) out.audio_state = SDV_AUD_FIX_P;
                    if (run_res == SDV_RES_14BIT && st.en_q_code) {
                        if (bit(out.word_valid, WORD_Q0)) {
                            if (st.force_ecc_check && (uint16_t)(calc_q(out) ^ out.w(WORD_Q0)) != 0) { state = STG_BAD_BLOCK; blk_mark_broken(out); }
                        } else {
                            uint16_t q = calc_q(out);
                            if (out.w(WORD_Q0) != q) { blk_set_word(out, WORD_Q0, q, bit(out.line_crc, WORD_Q0), false); blk_set_fixed(out, WORD_Q0); }
                            else blk_set_valid(out, WORD_Q0);
                        }
                    }
                }
            } else if (run_res == SDV_RES_14BIT) {
                if (st.en_q_code) state = STG_Q_CORR;
                else if (aud_errs == 0) state = STG_NO_CHECK;
            } else if (aud_errs == 0) state = STG_NO_CHECK;
        } else if (state == STG_Q_CORR) {
            state = STG_BAD_BLOCK;
            if (bit(out.word_valid, WORD_Q0)) {
                fix = fix_by_q(out, first_bad, second_bad);
                if (!bit(out.line_crc, WORD_P0)) second_bad = WORD_P0;
                if (fix == FIX_DONE) { state = STG_DATA_OK; blk_clear_cwd(out, first_bad); blk_clear_cwd(out, second_bad); out.audio_state = SDV_AUD_FIX_Q; }
                else if (fix == FIX_NOT_NEED) { state = STG_DATA_OK; blk_clear_cwd(out, first_bad); blk_clear_cwd(out, second_bad); if (first_bad < WORD_P0) out.audio_state = SDV_AUD_FIX_Q; }
                else if (fix == FIX_SWITCH_P) state = STG_P_CORR;
                else if (fix == FIX_BROKEN) blk_mark_broken(out);
            } else if (first_bad == NO_ERR_INDEX) {
                state = STG_NO_CHECK;
                uint16_t ecc = calc_p(out);
                blk_set_word(out, WORD_P0, ecc, false, bit(out.cwd_fixed, WORD_P0)); blk_set_fixed(out, WORD_P0);
                ecc = calc_q(out);
                blk_set_word(out, WORD_Q0, ecc, false, bit(out.cwd_fixed, WORD_Q0)); blk_set_fixed(out, WORD_Q0);
            }
        } else if (state == STG_BAD_BLOCK) {
            out.cwd_applied = false;
            if (fill_passes >= MAX_PASSES) break;
            run_res = (run_res == SDV_RES_16BIT) ? SDV_RES_14BIT : SDV_RES_16BIT;
            state = STG_DATA_FILL;
        } else break;
        if (stage_count > (STG_CONVERT_MAX * MAX_PASSES)) break;
    }
}

/* STC007Deinterleaver::processBlock for the block whose eight lines were gathered into `lines` (base = 0) */
__device__ inline void process_block(const sdv_deint_settings &st, const Lines8 &lines, size_t base, Block &out)
{
    uint8_t run_res;
    run_res = (st.res_mode == SDV_RES_MODE_14BIT || st.res_mode == SDV_RES_MODE_14BIT_AUTO) ? SDV_RES_14BIT : SDV_RES_16BIT;
    /* Short cut for the block a clean tape is made of: no word failed its CRC.  The stages then reduce to
     * ERROR_CHECK (no errors) -> TASK_SELECTION -> [P_CORR: P syndrome 0 -> FIX_NOT_NEED -> Q syndrome 0] -> DATA_OK
     * and leave the block exactly as DATA_FILL made it.  Anything else takes the full state machine below. */
    blk_clear(out);
    set_word_data(st, lines, base, out, run_res);
    out.audio_state = SDV_AUD_ORIG;
    const uint8_t bad = (uint8_t)(~out.line_crc & total_mask(out));
    if (bad == 0) {
        if (!st.force_ecc_check || !st.en_p_code) return;
        if ((uint16_t)(calc_p(out) ^ out.w(WORD_P0)) == 0) {
            if (!(run_res == SDV_RES_14BIT && st.en_q_code)) return;
            if ((uint16_t)(calc_q(out) ^ out.w(WORD_Q0)) == 0) return;
        }
        /* all words passed their CRC but a syndrome does not vanish: P_CORR / the Q check mark the block broken; with a fixed
         * resolution there is no second attempt (what a 16-bit probe of 14-bit material runs into block after block) */
        if (st.res_mode == SDV_RES_MODE_14BIT || st.res_mode == SDV_RES_MODE_16BIT) { blk_mark_broken(out); return; }
    } else if (run_res == SDV_RES_14BIT && (bad & (bad - 1)) == 0) {
        /* Second short cut, the block next to a lost line: exactly one word failed its CRC (14-bit mode).
         *   audio word k: TASK_SELECTION -> P_CORR: fixByP(k) (FIX_DONE or FIX_NOT_NEED) -> audio state FIX_P -> Q syndrome check
         *   P word      : P_CORR finds P invalid -> Q_CORR: fixByQ(none, P): Q syndrome 0 -> recalcP -> FIX_NOT_NEED
         *   Q word      : P_CORR: fixByP(none): P syndrome 0 -> FIX_NOT_NEED -> Q word recomputed
         * Whatever does not end well here (a syndrome that does not vanish) restarts in the state machine below. */
        if (bad & 0x3F) {
            if (st.en_p_code) {
                const uint8_t k = (uint8_t)(__ffs((int)bad) - 1);
                (void)fix_by_p(out, k);
                blk_clear_cwd(out, k);
                out.audio_state = SDV_AUD_FIX_P;
                if (!st.en_q_code) return;
                if (!(st.force_ecc_check && (uint16_t)(calc_q(out) ^ out.w(WORD_Q0)) != 0)) return;
            }
        } else if (bad == (1u << WORD_P0)) {
            if (!st.force_ecc_check || !st.en_p_code || !st.en_q_code) return;      /* DATA_OK / NO_CHECK / P_CORR -> NO_CHECK */
            if ((uint16_t)(calc_q(out) ^ out.w(WORD_Q0)) == 0) { recalc_p(out); blk_clear_cwd(out, WORD_P0); return; }
        } else {
            if (!st.force_ecc_check || !st.en_p_code) return;
            if ((uint16_t)(calc_p(out) ^ out.w(WORD_P0)) == 0) {
                if (st.en_q_code) {
                    const uint16_t q = calc_q(out);
                    if (out.w(WORD_Q0) != q) { blk_set_word(out, WORD_Q0, q, false, false); blk_set_fixed(out, WORD_Q0); }
                    else blk_set_valid(out, WORD_Q0);
                }
                return;
            }
        }
    }
    process_block_fsm(st, lines, out);
}

struct DeintArgs { const sdv_deint_line *lines; size_t n_blocks; sdv_deint_settings st; sdv_block_rec *out; };

__device__ inline void deint_body(const DeintArgs &a, size_t s)
{
    Block b;
    PtrSrc src; src.p = a.lines;
    Lines8 l8;
    gather8(src, s, l8);
    process_block(a.st, l8, 0, b);
    sdv_block_rec r;
    for (int i = 0; i < 8; i++) { r.w_frame[i] = b.w_frame[i]; r.w_line[i] = b.w_line[i]; r.words[i] = b.w(i); }
    r.line_crc = b.line_crc; r.cwd_fixed = b.cwd_fixed; r.word_valid = b.word_valid; r.resolution = b.resolution;
    r.audio_state = b.audio_state; r.cwd_applied = b.cwd_applied ? 1 : 0; r.sample_rate = 44056;   /* STC007DataBlock::clear, stc007datablock.cpp:55 */
    a.out[s] = r;
}
} // namespace sdvd

__global__ void __launch_bounds__(256) sdv_k_stc007_deint(sdvd::DeintArgs a)
{
    size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < a.n_blocks) sdvd::deint_body(a, s);
}
#endif
