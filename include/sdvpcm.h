/*
 * sdvpcm.h - C-ABI of the MI355X-native SDVPCMdecoder decode engine (libsdvpcm_hip.so).
 *
 * Plain C, POD only, caller-allocated buffers.  This is the boundary a maintainer of
 * Fagear/SDVPCMdecoder binds: the Qt workers VideoToDigital and
 * STC007DataStitcher stay, their inner per-line / per-block calls are replaced by whole-batch
 * calls into this library.
 *
 * Each entry point cites the reference interface it replaces (file:line in the reference tree).
 * All "device" pointers are HIP device pointers (hipMalloc / torch CUDA tensors); the *_host
 * convenience variants take host pointers and stage through PCIe.
 */
#ifndef SDVPCM_H
#define SDVPCM_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDV_ABI_VERSION 5   /* 2: output capacities on sdv_binarize_frames / sdv_pcm1_binarize_lines; 3: sdv_audio_process, sdv_wav_pack, sdv_wav_header, sdv_decode_frames (additions only);
                             * 4: sdv_pcm16x0_binarize_lines, sdv_audio_stalled, sdv_set_frame_flags, sdv_double_width, sdv_vis_render_lines (additions); the calls that used to refuse PCM-16x0 frames of the wrong size and the
                             * AudioProcessor's dead ends now follow the reference; the PCM-16x0 stitch state blob grew by conv_queue's remainder;
                             * 5: sdv_run_info grew by frames_met (at its end), sdv_binarize_lines (addition) */

/* ---- status codes ---------------------------------------------------------------------------
 * 0..4 mirror Binarizer::LB_RET_* (binarizer.h:268-275); 16.. mirror STC007Deinterleaver::DI_RET_*
 * (stc007deinterleaver.h:97-103); negative = engine/runtime errors (no reference equivalent:
 * the reference has no device boundary). */
enum {
    SDV_OK = 0,
    SDV_ERR_NULL_VIDEO = 1,     /* LB_RET_NULL_VIDEO */
    SDV_ERR_NULL_PCM = 2,       /* LB_RET_NULL_PCM */
    SDV_ERR_SHORT_LINE = 3,     /* LB_RET_SHORT_LINE: line shorter than the format's bit count */
    SDV_ERR_NO_COORD = 4,       /* LB_RET_NO_COORD */
    SDV_ERR_NULL_LINES = 16,    /* DI_RET_NULL_LINES */
    SDV_ERR_NULL_BLOCK = 17,    /* DI_RET_NULL_BLOCK */
    SDV_ERR_NO_DATA = 18,       /* DI_RET_NO_DATA */
    SDV_ERR_BAD_ARG = -1,
    SDV_ERR_HIP = -2,           /* HIP runtime failure, see sdv_last_error() */
    SDV_ERR_NO_DEVICE = -3,
    SDV_ERR_UNSUPPORTED = -4
};

/* PCM formats (PCMLine::TYPE_*, pcmline.h:77-84) */
enum { SDV_PCM_PCM1 = 0, SDV_PCM_PCM16X0 = 1, SDV_PCM_STC007 = 2 };
/* Binarizer::MODE_* (binarizer.h:207-214) */
enum { SDV_MODE_DRAFT = 0, SDV_MODE_FAST = 1, SDV_MODE_NORMAL = 2, SDV_MODE_INSANE = 3 };
/* PCMLine::SRVLINE_* (pcmline.h:104-115) */
enum {
    SDV_SRV_NO = 0, SDV_SRV_NEW_FILE, SDV_SRV_END_FILE, SDV_SRV_FILLER, SDV_SRV_END_FIELD,
    SDV_SRV_END_FRAME, SDV_SRV_HEADER_LINE, SDV_SRV_CTRL_BLOCK
};

/* sdv_line_rec.flags */
enum {
    SDV_LF_REF_SWEEPED = 1 << 0,    /* PCMLine::ref_level_sweeped   */
    SDV_LF_COORDS_SWEEPED = 1 << 1, /* PCMLine::coords_sweeped      */
    SDV_LF_BY_EXT_TUNE = 1 << 2,    /* PCMLine::data_by_ext_tune    */
    SDV_LF_BW_SET = 1 << 3,         /* PCMLine::hasBWSet()          */
    SDV_LF_COORDS_SET = 1 << 4,     /* PCMLine::hasDataCoordSet()   */
    SDV_LF_FORCED_BAD = 1 << 5,     /* PCMLine::isForcedBad()       */
    SDV_LF_CRC_VALID = 1 << 6,      /* PCMLine::isCRCValid()        */
    SDV_LF_FROM_DOUBLED = 1 << 7    /* PCMLine::isSourceDoubleWidth() */
};
/* sdv_line_rec.word_state */
enum { SDV_WS_WORD_CRC = 1 << 0, SDV_WS_WORD_VALID = 1 << 1 };

/* One binarized STC-007 line: everything needed to re-hydrate an STC007Line through its
 * public API (stc007line.h:153-221, pcmline.h:137-186).  48 bytes, no padding. */
typedef struct sdv_line_rec {
    uint32_t frame_number;          /* PCMLine::frame_number */
    uint16_t line_number;           /* PCMLine::line_number (1-based; 0 for trailing service lines) */
    uint16_t words[9];              /* STC007Line::words: 8 x 14-bit + CRCC as read */
    uint16_t calc_crc;              /* PCMLine::getCalculatedCRC() */
    int16_t data_start, data_stop;  /* PCMLine::coords */
    uint16_t marker_start_bg_coord, marker_start_ed_coord, marker_stop_ed_coord;
    uint8_t black_level, white_level, ref_low, ref_level, ref_high;
    uint8_t hysteresis_depth, shift_stage;
    uint8_t service_type;           /* SDV_SRV_* */
    uint8_t mark_st_stage, mark_ed_stage;
    uint8_t flags;                  /* SDV_LF_* */
    uint8_t word_state;             /* SDV_WS_* (uniform over the 9 words at binarizer output) */
} sdv_line_rec;

/* bin_preset_t (binarizer.h:163-186) */
typedef struct sdv_bin_preset {
    uint8_t max_black_lvl, min_white_lvl, min_contrast, min_ref_lvl, max_ref_lvl, min_valid_crcs;
    uint8_t mark_max_dist, left_bit_pick, right_bit_pick;
    uint8_t en_force_coords, en_coord_search, en_first_line_dup, en_good_no_marker;
    uint8_t _pad;
    int16_t horiz_start, horiz_stop;    /* bin_preset_t::horiz_coords */
} sdv_bin_preset;

/* Preset incoming "good parameters" of a Binarizer (Binarizer::setGoodParameters, binarizer.cpp:353-377):
 * all-zero = nothing preset. */
typedef struct sdv_bin_state {
    uint8_t in_def_black, in_def_white, in_def_reference;
    uint8_t do_ref_lvl_sweep;               /* Binarizer::do_ref_lvl_sweep as the previous line left it (sticky, binarizer.cpp:1104-1128): read by
                                             * the level detection of the next line (:3409).  Only the per-line entries take it from here (for
                                             * PCM-1 it is set by a line that is not SDV_LF_BY_EXT_TUNE and has SDV_LF_BW_SET, to "the mode is
                                             * MODE_INSANE"); inside sdv_v2d_state the flag has its own field and this byte is 0 */
    int16_t in_def_start, in_def_stop;      /* CoordinatePair::NO_COORD_LEFT/RIGHT when unset */
    uint8_t in_def_from_doubled, _pad2;
} sdv_bin_state;

/* FrameBinDescriptor (frametrimset.h:69-97) as emitted by VideoToDigital at END_FRAME
 * (videotodigital.cpp:1636-1714); time_odd/time_even are wall clock in the reference and are
 * not reproduced. 32 bytes. */
typedef struct sdv_frame_stats {
    uint32_t frame_id;
    uint16_t line_length;
    uint16_t lines_odd, lines_even;
    uint16_t lines_pcm_odd, lines_pcm_even;
    uint16_t lines_bad_odd, lines_bad_even;
    uint16_t lines_dup_odd, lines_dup_even;
    int16_t data_start, data_stop;      /* data_coord */
    uint8_t data_from_doubled, data_not_sure;
    uint8_t _pad[4];
} sdv_frame_stats;

/* The frame-to-frame state VideoToDigital::doBinarize carries (videotodigital.cpp:706-720 locals +
 * the Binarizer presets in_def_*, binarizer.h:310-313).  It is everything a frame's result depends
 * on besides its own pixels, which is what lets the engine decode frames of one stream in parallel
 * and still return the sequential reference result (DESIGN.md "chain speculation"). 120 bytes. */
typedef struct sdv_coord { int16_t data_start, data_stop; } sdv_coord;
typedef struct sdv_v2d_state {
    sdv_bin_state bin;                  /* line_converter presets after the last line */
    uint8_t do_ref_lvl_sweep;           /* Binarizer::do_ref_lvl_sweep (sticky, binarizer.cpp:1104-1128) */
    uint8_t reset_stats;                /* VideoToDigital::reset_stats (videotodigital.cpp:778-790) */
    uint8_t n_last_valid;               /* last_valid_coord_list.size() <= COORD_HISTORY_DEPTH (9) */
    uint8_t n_long_valid;               /* long_valid_coords.size()     <= COORD_LONG_HISTORY (16) */
    uint8_t last_valid_doubled_mask_lo, last_valid_doubled_mask_hi; /* from_doubled bit per entry */
    uint16_t long_valid_doubled_mask;
    sdv_coord last_valid[9];
    sdv_coord long_valid[16];
    uint8_t _pad[2];
} sdv_v2d_state;

/* ---- deinterleave / error-correction stage ---------------------------------------------------- */
/* STC007Deinterleaver::RES_MODE_* (stc007deinterleaver.h:106-113) */
enum { SDV_RES_MODE_14BIT = 0, SDV_RES_MODE_14BIT_AUTO = 1, SDV_RES_MODE_16BIT_AUTO = 2, SDV_RES_MODE_16BIT = 3 };
/* STC007DataBlock::RES_* / AUD_* (stc007datablock.h:95-111) */
enum { SDV_RES_14BIT = 0, SDV_RES_16BIT = 1 };
enum { SDV_AUD_ORIG = 0, SDV_AUD_FIX_P = 1, SDV_AUD_FIX_Q = 2, SDV_AUD_BROKEN = 3 };

/* What STC007Deinterleaver::setWordData reads of one assembled STC007Line (stc007deinterleaver.cpp:1126-1294).
 * 24 bytes. */
enum { SDV_DL_FIXED_BY_CWD = 1 << 0,    /* STC007Line::isFixedByCWD() */
       SDV_DL_COORDS_BW_OK = 1 << 1 };  /* coords.areValid() && hasBWSet(): the word validity when CRCs are ignored */
typedef struct sdv_deint_line {
    uint32_t frame_number;
    uint16_t line_number;
    uint16_t words[8];          /* STC007Line::getWord(0..7) */
    uint8_t word_crc_ok;        /* bit i = STC007Line::isWordCRCOk(i), i = 0..7 */
    uint8_t flags;              /* SDV_DL_* */
} sdv_deint_line;

/* STC007Deinterleaver settings (stc007deinterleaver.h:151-161; setters :133-282) */
typedef struct sdv_deint_settings {
    uint8_t res_mode;           /* SDV_RES_MODE_* (setResMode) */
    uint8_t ignore_crc;         /* setIgnoreCRC */
    uint8_t force_ecc_check;    /* setForcedErrorCheck */
    uint8_t en_p_code, en_q_code, en_cwd;   /* setPCorrection / setQCorrection / setCWDCorrection */
    uint8_t _pad[2];
} sdv_deint_settings;

/* One STC007DataBlock (stc007datablock.h:144-160). 72 bytes. Bit i of the three masks = word i. */
typedef struct sdv_block_rec {
    uint32_t w_frame[8];
    uint16_t w_line[8];
    uint16_t words[8];
    uint8_t line_crc, cwd_fixed, word_valid;
    uint8_t resolution;         /* SDV_RES_* */
    uint8_t audio_state;        /* SDV_AUD_* */
    uint8_t cwd_applied;
    uint16_t sample_rate;
} sdv_block_rec;

/* ---- stitch stage ------------------------------------------------------------------------------ */
/* PCMSamplePair (pcmsamplepair.h:77-143) / PCMSample (:31-74). 12 bytes. */
enum { SDV_SF_BLOCK_OK = 1 << 0,    /* PCMSample::data_block_ok */
       SDV_SF_WORD_VALID = 1 << 1,  /* PCMSample::word_valid    */
       SDV_SF_WORD_FIXED = 1 << 2,  /* PCMSample::word_fixed    */
       SDV_SF_WORD_MASKED = 1 << 3  /* PCMSample::word_masked   */ };
enum { SDV_PAIR_SRV_NO = 0, SDV_PAIR_SRV_NEW_FILE = 1, SDV_PAIR_SRV_END_FILE = 2 };   /* PCMSamplePair::SRV_* */
typedef struct sdv_sample_pair {
    int16_t audio_word[2];          /* CH_LEFT, CH_RIGHT */
    uint8_t sample_flags[2];        /* SDV_SF_* per channel */
    uint16_t sample_rate;           /* 44056 / 44100 */
    uint8_t emphasis;
    uint8_t service_type;           /* SDV_PAIR_SRV_* */
    uint16_t _pad;
} sdv_sample_pair;

/* FrameAsmSTC007 (frametrimset.h:116-275) as emitted with guiUpdFrameAsm (stc007datastitcher.cpp:7431). 64 bytes. */
typedef struct sdv_frame_asm {
    uint32_t frame_number;
    uint16_t odd_std_lines, even_std_lines, odd_data_lines, even_data_lines, odd_valid_lines, even_valid_lines;
    uint16_t odd_top_data, odd_bottom_data, even_top_data, even_bottom_data, odd_sample_rate, even_sample_rate;
    uint16_t blocks_total, blocks_drop, samples_drop;
    uint16_t inner_padding, outer_padding;
    uint16_t blocks_broken_field, blocks_broken_seam, blocks_fix_p, blocks_fix_q, blocks_fix_cwd;
    uint8_t field_order, odd_ref, even_ref, service_type;
    uint8_t video_standard, tff_cnt, bff_cnt, odd_resolution, even_resolution;
    uint8_t flags;                  /* SDV_FA_* */
    uint8_t flags2;
    int8_t ctrl_index, ctrl_hour, ctrl_minute, ctrl_second, ctrl_field;
} sdv_frame_asm;
enum { SDV_FA_ORDER_PRESET = 1 << 0, SDV_FA_ORDER_GUESSED = 1 << 1, SDV_FA_TRIM_OK = 1 << 2, SDV_FA_INNER_OK = 1 << 3,
       SDV_FA_OUTER_OK = 1 << 4, SDV_FA_INNER_SILENCE = 1 << 5, SDV_FA_OUTER_SILENCE = 1 << 6, SDV_FA_VID_STD_PRESET = 1 << 7 };
enum { SDV_FA2_ODD_EMPHASIS = 1 << 0, SDV_FA2_EVEN_EMPHASIS = 1 << 1, SDV_FA2_VID_STD_GUESSED = 1 << 2 };

/* STC007DataStitcher settings (slots stc007datastitcher.h:331-350, defaults :20-31, :7228-7236) */
typedef struct sdv_stitch_settings {
    uint8_t video_standard;         /* setVideoStandard: FrameAsmDescriptor::VID_* (0 = auto) */
    uint8_t field_order;            /* setFieldOrder: ORDER_* (0 = auto) */
    uint8_t enable_p, enable_q, enable_cwd;     /* setPCorrection / setQCorrection / setCWDCorrection */
    uint8_t m2_format;              /* setM2SampleFormat */
    uint8_t resolution_preset;      /* setResolutionPreset: SAMPLE_RES_* (0 = auto) */
    uint8_t max_unch_14, max_unch_16;           /* setFineMaxUnch14/16 */
    uint8_t use_ecc;                /* setFineUseECC (ignore_CRC = !use_ecc) */
    uint8_t mask_seams;             /* setFineMaskSeams */
    uint8_t broke_mask;             /* setFineBrokeMask */
    uint8_t top_line_fix;           /* setFineTopLineFix */
    uint8_t _pad;
    uint16_t sample_rate_preset;    /* setSampleRatePreset (1 = auto) */
} sdv_stitch_settings;

typedef struct sdv_engine sdv_engine;

/* ---- engine lifetime ------------------------------------------------------------------------ */
/* Creates an engine bound to HIP device `device` (one engine per GPU / per process rank).
 * Replaces: construction of VideoToDigital + its member Binarizer (videotodigital.h:114,
 * videotodigital.cpp:3-25).  Returns NULL on failure; sdv_last_error(NULL) then holds the reason
 * (per calling thread).  Every entry point switches to its engine's device for the duration of the call and
 * restores the caller's current device before it returns.  An engine is used by one thread at a time, like the
 * reference's workers. */
sdv_engine *sdv_engine_create(int device);
void sdv_engine_destroy(sdv_engine *e);
const char *sdv_last_error(const sdv_engine *e);
int sdv_abi_version(void);

/* ---- settings ------------------------------------------------------------------------------- */
/* Binarizer::getDefaultFineSettings (binarizer.cpp:380-385) */
void sdv_default_bin_preset(sdv_bin_preset *out);
/* Binarizer::setFineSettings (binarizer.cpp:394-403) */
int sdv_set_bin_preset(sdv_engine *e, const sdv_bin_preset *p);
/* Binarizer::setMode (binarizer.cpp:120-177) via VideoToDigital::setBinarizationMode */
int sdv_set_mode(sdv_engine *e, int mode);

/* VideoToDigital::setCheckLineDup (videotodigital.cpp:645-664) */
int sdv_set_check_line_dup(sdv_engine *e, int on);
/* VideoToDigital::setPCMType (videotodigital.cpp:557-604); m2_sample_format = TYPE_M2 */
int sdv_set_pcm_type(sdv_engine *e, int pcm_type, int m2_sample_format);
/* Forget the stream: equivalent to a freshly constructed VideoToDigital (reset_stats = true). */
int sdv_reset_stream(sdv_engine *e);
/* The feedback state after the last frame decoded so far (and a way to resume from a saved one). */
int sdv_get_chain_state(const sdv_engine *e, sdv_v2d_state *out);
int sdv_set_chain_state(sdv_engine *e, const sdv_v2d_state *in);
/* The same for a PCM-16x0 stream, whose window of last valid coordinates holds 27 entries (three sub-lines per video line): an opaque
 * blob of sdv_pcm16x0_chain_state_size() bytes (192).  PCM-1 streams use sdv_get/set_chain_state. */
size_t sdv_pcm16x0_chain_state_size(void);
int sdv_get_pcm16x0_chain_state(const sdv_engine *e, void *out, size_t cap);
int sdv_set_pcm16x0_chain_state(sdv_engine *e, const void *in, size_t n);

/* flags of sdv_binarize_frames */
enum {
    SDV_FLAG_NEW_FILE = 1u << 0,    /* first frame of a source: a NEW_FILE service line precedes it (vin_ffmpeg.cpp:275-280) */
    SDV_FLAG_DOUBLED = 1u << 1,     /* rows were width-doubled upstream (VideoLine::isDoubleWidth, ffmpegwrapper.cpp:179-186) */
    SDV_FLAG_END_FILE = 1u << 2     /* last frames of a source: the filler frame + END_FILE line the input plugin appends follow them
                                     * (VideoInFFMPEG::insertDummyFrame(true, false), vin_ffmpeg.cpp:367-523): height + 4 more records,
                                     * one more sdv_frame_stats row */
};

/* How the last sdv_binarize_frames call was scheduled (chain speculation). */
typedef struct sdv_run_info {
    uint32_t frames;            /* frames in the call */
    uint32_t rounds;            /* speculation rounds (launches of the frame kernel) needed; 1 = fully parallel */
    uint32_t frames_launched;   /* frame decodes executed, including re-decodes after a misprediction */
    uint32_t frames_general;    /* of those, by the full kernel (the lean one has no general path and gives such frames up) */
    float kernel_ms;            /* HIP-event time of the frame kernel launches of this call (sdv_set_profiling) */
    uint32_t sweeps;            /* reference-level sweeps (Binarizer::calcRefLevelBySweep) settled by the sweep kernels for this call */
    uint32_t frames_met;        /* of frames_general: decodes that ended early, on a line where the state was the one the frame's last complete decode had there
                                 * (the rest of the frame could only come out the same: DESIGN.md, "a pass that meets the last one") */
} sdv_run_info;
int sdv_get_run_info(const sdv_engine *e, sdv_run_info *out);
/* Bracket every frame-kernel launch with hipEvents on the caller's stream and report the sum in sdv_run_info. */
int sdv_set_profiling(sdv_engine *e, int on);

/* records emitted per frame: `height` scanlines + 2 END_FIELD + 1 END_FRAME service lines */
/* Per-frame marks of the caller for the NEXT frame entry call on this engine (sdv_binarize_frames, sdv_pcm1_binarize_frames,
 * sdv_pcm16x0_binarize_frames, sdv_decode_frames): flags[i] belongs to frame i of that call, the call consumes them (whether it succeeds or not).
 * SDV_FRAME_EMPTY: the frame was dropped by the video input - VideoInFFMPEG::insertDummyFrame(false, true) (vin_ffmpeg.cpp:367-522) sends its lines as
 * empty VideoLines, Binarizer::processLine answers each with a silent line of invalid CRC (binarizer.cpp:569-570, :1689-1700) and the worker books them
 * as lines that did not read; the pixels of such a frame are not looked at.  flags is a HOST array of n bytes (n == 0: no marks). */
enum { SDV_FRAME_EMPTY = 1 };
int sdv_set_frame_flags(sdv_engine *e, const uint8_t *flags, size_t n);

/* The 2x width doubling the input plugin applies to narrow sources before the lines reach the Binarizer (FFMPEGWrapper::needsDoubleWidth /
 * getFinalWidth, ffmpegwrapper.cpp:179-197: widths between MIN_DBL_WIDTH 10 and MAX_DBL_WIDTH 959), as an integer pixel replication on the device:
 * dst[r][2 x] = dst[r][2 x + 1] = src[r][x] for `rows` rows of `width` pixels (row strides in bytes; dst rows are 2 * width pixels).  The frames it
 * makes go to the frame entries with SDV_FLAG_DOUBLED.  NOT the reference's pixels: the reference lets libswscale resize with SWS_GAUSS
 * (ffmpegwrapper.cpp:236-241), which is outside the rebuilt path and not reproducible without that library (parity unpinned there);
 * this is the integer doubler SURVEY section 8f-3 names.  sdv_needs_double_width is the reference's rule for when to apply it.
 * Device pointers; asynchronous on `stream`; src and dst must not overlap. */
int sdv_needs_double_width(int width);
int sdv_double_width(sdv_engine *e, const uint8_t *src, size_t src_row_stride, int width, size_t rows, uint8_t *dst, size_t dst_row_stride, void *stream);

size_t sdv_records_per_frame(int height);
/* records one sdv_binarize_frames call emits: n_frames * (height + 3), + 1 with SDV_FLAG_NEW_FILE, + height + 4 with SDV_FLAG_END_FILE */
size_t sdv_binarize_records(int height, int n_frames, unsigned flags);

/* ---- hot path: binarize + bit-extract + CRC for a batch of whole frames ------------------------
 * Replaces the body of VideoToDigital::doBinarize (videotodigital.cpp:698-1815) including every call
 * it makes into its Binarizer (setMode/setSource/setOutput/processLine/setGoodParameters/
 * setDataCoordinates/setBWLevels, videotodigital.cpp:834-1003, 1198, 1369, 1468-1521) for `n_frames`
 * consecutive frames of one stream.  Continues from the state the previous call left
 * (sdv_reset_stream() to start over).
 *
 *  luma         device pointer; frame f, row r at luma + f*frame_stride + r*row_stride, `width` bytes
 *               of 8-bit luma per row (the pixel_data of the reference's VideoLine, videoline.h:37-88).
 *               Rows should be 16-byte aligned for full-rate loads (any alignment is accepted).
 *  out_lines    device pointer to lines_cap records; the call writes sdv_binarize_records(height, n_frames, flags) of them and
 *               refuses (SDV_ERR_BAD_ARG, nothing written) when lines_cap is less: (height+3)*n_frames records (+1 leading NEW_FILE record with
 *               SDV_FLAG_NEW_FILE), in exactly the order VideoToDigital pushes STC007Line objects
 *               into its output queue: odd-field rows (line numbers 1,3,..), END_FIELD, even-field
 *               rows (2,4,..), END_FIELD, END_FRAME (vin_ffmpeg.cpp:281-350).  With SDV_FLAG_END_FILE height+4 more:
 *               the filler frame (frame number first_frame_no + n_frames): FILLER lines in the same field order, END_FIELD
 *               twice, END_FILE, END_FRAME.
 *  out_stats    device pointer to stats_cap rows, one FrameBinDescriptor per frame (signal guiUpdFrameBin); one more row with
 *               SDV_FLAG_END_FILE (the worker reports the filler frame too).
 *  frame_stride at least (height-1)*row_stride + width when n_frames > 1 (frames do not overlap), else SDV_ERR_BAD_ARG.
 *  stream       hipStream_t (NULL = default stream).  The call returns after the device work of the
 *               batch has been validated (it synchronises `stream` at least once).
 * Returns SDV_OK or an SDV_ERR_* code; invalid input is refused up front like the reference's early
 * returns (binarizer.cpp:465-478, 582-589). */
int sdv_binarize_frames(sdv_engine *e, const uint8_t *luma, size_t row_stride, size_t frame_stride, int width, int height,
                        int n_frames, uint32_t first_frame_no, unsigned flags,
                        sdv_line_rec *out_lines, size_t lines_cap, sdv_frame_stats *out_stats, size_t stats_cap, void *stream);

/* ---- hot path: deinterleave + P/Q error correction for a batch of data blocks -------------------
 * Replaces STC007Deinterleaver::processBlock(line_shift) (stc007deinterleaver.cpp:286-1123) called for
 * line_shift = 0 .. n_blocks-1 over one buffer of assembled lines - the loops of STC007DataStitcher::tryPadding,
 * performCWD and performDeinterleave (stc007datastitcher.cpp:1546-1561, 5919-5939, 6682-6717).
 * Block s is assembled from lines s, s+16, .., s+112 (stc007datablock.h:40-58), so n_lines must exceed
 * 112 + (n_blocks-1), else SDV_ERR_NO_DATA (DI_RET_NO_DATA).  Device pointers; asynchronous on `stream`. */
void sdv_default_deint_settings(sdv_deint_settings *st);
int sdv_deinterleave_blocks(sdv_engine *e, const sdv_deint_line *lines, size_t n_lines, const sdv_deint_settings *settings,
                            sdv_block_rec *out_blocks, size_t n_blocks, void *stream);

/* ---- stitch stage: STC007DataStitcher (stc007datastitcher.h:74-353) ------------------------------------------- */
/* How the last sdv_stitch_frames call was scheduled (parallel turns in rounds). */
typedef struct sdv_stitch_info {
    uint32_t steps;             /* stitcher turns (frame pairs) completed by the call */
    uint32_t rounds;            /* parallel rounds until every turn had run from its predecessor's final output */
    uint32_t steps_launched;    /* turn executions over all rounds */
    uint32_t pipelined;         /* 1: the call ran its analysis and first round without waiting for the host (a stream that plays);
                                 * 2: ... and the field order and resolution histories were saturated, so the host's check needed no replay of them;
                                 * + 4: (sdv_decode_frames) the stage's kernels were queued right behind the frame kernel, ahead of the host's look at its round;
                                 * + 8: ... they had been, the frame kernel's round was not its last, and the stage was run again on the final records */
    float device_ms;            /* analysis + rounds + packing on the device (profiling on) */
    uint32_t direct_frames;     /* sdv_decode_frames: frames whose lines the frame kernel wrote into the stitch stage's field buffers itself (no line records) */
} sdv_stitch_info;

/* Defaults of the STC007DataStitcher constructor / setDefaultFineSettings (stc007datastitcher.cpp:20-31, 7228-7236),
 * with P and Q correction on as the application sets them. */
void sdv_default_stitch_settings(sdv_stitch_settings *st);
/* setVideoStandard / setFieldOrder / setPCorrection / setQCorrection / setCWDCorrection / setM2SampleFormat /
 * setResolutionPreset / setSampleRatePreset / setFine* slots (stc007datastitcher.h:331-350) in one call. */
int sdv_set_stitch_settings(sdv_engine *e, const sdv_stitch_settings *st);
/* A freshly constructed STC007DataStitcher: statistics, previous-frame memory and queued lines are dropped. */
int sdv_reset_stitcher(sdv_engine *e);
/* Like every entry point of an engine this is a call of the engine's one worker thread (the engine is as little re-entrant as the
 * Binarizer it replaces): `direct_frames` of a fused call is counted here, on first asking, with one small synchronous read-back
 * from the device - not a getter to poll from another thread while a decode call runs.  After a call that returned an error
 * every field is zero. */
int sdv_get_stitch_info(const sdv_engine *e, sdv_stitch_info *out);

/* The stitcher's stream state as an opaque blob (previous frame's descriptor, statistics rings, the 112 hand-over lines):
 * checkpointing, and handing a tape over to the engine of the next GPU when one stream is sharded across GPUs (the "field
 * seam" that travels by all-gather).  sdv_set_stitch_state drops lines that still wait for a successor
 * frame.  sdv_saturate_stitch_stats fills the two statistics rings with their current majority - for an engine that joined
 * the stream after a short warm-up and is about to compare its state with the true one. */
size_t sdv_stitch_state_size(void);
int sdv_get_stitch_state(sdv_engine *e, void *out, size_t cap);
int sdv_set_stitch_state(sdv_engine *e, const void *in, size_t n);
int sdv_saturate_stitch_stats(sdv_engine *e);

/* STC007DataStitcher::doFrameReassemble (stc007datastitcher.cpp:7239-7488) over a span of the binarized line stream.
 * `lines` is what VideoToDigital puts into the stitcher's input deque<STC007Line> (sdv_binarize_frames' out_lines,
 * service lines included; a file ends with the filler frame + END_FILE + END_FRAME the input plugin appends,
 * vin_ffmpeg.cpp:367-523).  Every frame that has a successor in the stream is reassembled: trim, field order and
 * padding detection, assembly, CWD, deinterleave + P/Q correction, masking of broken seams.  The PCMSamplePair
 * stream goes to out_pairs (what outputSamplePair / outputFileStart / outputFileStop push into the output deque,
 * :6483-6672), one FrameAsmSTC007 per guiUpdFrameAsm emission to out_frames.  The last frame of the span waits
 * inside the engine for the next call, exactly as it would wait in the reference's queue for its successor.
 * Restrictions (SDV_ERR_UNSUPPORTED otherwise): the lines of a frame carry one frame number and the numbers increase
 * along the stream (what VideoInFFMPEG produces); records behind an END_FILE frame are discarded like the queue flush
 * at :7380-7400 - feed one source per stream.
 * All buffers are device pointers.  The call returns when the outputs are complete; *n_pairs / *n_frames receive the
 * counts (also when SDV_ERR_BAD_ARG reports that pairs_cap / frames_cap were too small). */
int sdv_stitch_frames(sdv_engine *e, const sdv_line_rec *lines, size_t n_lines, sdv_sample_pair *out_pairs, size_t pairs_cap,
                      size_t *n_pairs, sdv_frame_asm *out_frames, size_t frames_cap, size_t *n_frames, void *stream);

/* ---- STC-007, line by line -------------------------------------------------------------------------------------------
 * Binarizer::processLine (binarizer.h:361, binarizer.cpp:443-1724) with an STC007Line as output, for n_lines video lines in one call: the per-line contract of
 * the reference's Binarizer itself (setSource / setOutput / processLine, videotodigital.cpp:834-1003), without the bookkeeping VideoToDigital wraps around it -
 * a host that keeps the reference's own per-line loop binds this.  Line i = luma + i*row_stride (width bytes), numbered first_line + i*line_step of frame
 * frame_number.  presets[i] is what the caller had set on its Binarizer before that line (setGoodParameters / setReferenceLevel / setDataCoordinates /
 * setBWLevels, binarizer.cpp:240-377; all zero or presets == NULL: nothing preset); mode and fine settings are the engine's (sdv_set_mode,
 * sdv_set_bin_preset); flags: SDV_FLAG_DOUBLED.  The record is the line as processLine leaves it (the duplicate-line mark and the coordinate damper are
 * VideoToDigital's: sdv_binarize_frames).  Service lines and empty lines carry no pixels and are the caller's to pass through.
 * Returns SDV_ERR_SHORT_LINE for lines under 137 px (LB_RET_SHORT_LINE), SDV_ERR_BAD_ARG when out_lines (lines_cap records) cannot take n_lines.
 * Device pointers.  The call returns when the records are complete (the reference-level sweeps of the lines that need one run between two passes over the lines). */
int sdv_binarize_lines(sdv_engine *e, const uint8_t *luma, size_t row_stride, int width, size_t n_lines,
                       const sdv_bin_state *presets, uint32_t frame_number, uint16_t first_line, uint16_t line_step,
                       unsigned flags, sdv_line_rec *out_lines, size_t lines_cap, void *stream);

/* ---- PCM-1 front half: one PCM1Line as Binarizer::processLine leaves it (pcm1line.h:59-146, pcmline.h:137-186) --------- */
/* 40 bytes.  Record of the oracle and of the reference driver today; the engine's PCM-1 binarize
 * entry will emit the same record. */
typedef struct sdv_pcm1_bin_rec {
    uint32_t frame_number;
    uint16_t line_number;
    uint16_t words[7];              /* L2 R2 L4 R4 L6 R6 (13 bit) + CRCC as read (or as the Bit Picker completed it) */
    uint16_t calc_crc;
    int16_t data_start, data_stop;  /* PCMLine::coords */
    uint8_t black_level, white_level, ref_low, ref_level, ref_high;
    uint8_t hysteresis_depth, shift_stage;
    uint8_t service_type;           /* SDV_SRV_* (SDV_SRV_HEADER_LINE when the header pattern was read) */
    uint8_t picked_bits_left, picked_bits_right;
    uint8_t flags;                  /* SDV_LF_* */
    uint8_t _pad[3];
} sdv_pcm1_bin_rec;

/* Binarizer::processLine (binarizer.h:361, binarizer.cpp:443-1724) with a PCM1Line as output, for n_lines video lines in one
 * launch: line i = luma + i*row_stride (width bytes), numbered first_line + i*line_step of frame frame_number.
 * presets[i] is what the caller had set on its Binarizer before that line (setGoodParameters / setReferenceLevel /
 * setDataCoordinates / setBWLevels, binarizer.cpp:240-377; all zero or presets == NULL: nothing preset); mode and fine settings are
 * the engine's (sdv_set_mode, sdv_set_bin_preset), coord_search is Binarizer::setCoordinatesSearch; flags: SDV_FLAG_DOUBLED.
 * Service lines and empty lines carry no pixels and are the caller's to pass through.  Returns SDV_ERR_SHORT_LINE for lines
 * under 94 px (LB_RET_SHORT_LINE), SDV_ERR_BAD_ARG when out_lines (lines_cap records) cannot take n_lines.  SDV_MODE_INSANE adds the reference level sweep
 * (sweepRefLevel / calcRefLevelBySweep, binarizer.cpp:3551-4120) to every line that does not read from its presets: a coordinate search per level.
 * Device pointers; asynchronous on `stream`. */
int sdv_pcm1_binarize_lines(sdv_engine *e, const uint8_t *luma, size_t row_stride, int width, size_t n_lines,
                            const sdv_bin_state *presets, uint32_t frame_number, uint16_t first_line, uint16_t line_step,
                            unsigned flags, int coord_search, sdv_pcm1_bin_rec *out_lines, size_t lines_cap, void *stream);

/* ---- PCM-16x0 front half: one PCM16X0SubLine as Binarizer::processLine leaves it (pcm16x0subline.h:113-125, pcmline.h:137-186) ---
 * A PCM-16x0 video line carries three sub-lines of 3 x 16 bit + CRCC (and one control bit between the second and the third); the
 * reference runs its Binarizer three times over a line, once per third (Binarizer::setLinePartMode, videotodigital.cpp:902-925), and
 * queues one PCM16X0SubLine per pass - three records per video line here, one for a service line.  36 bytes. */
typedef struct sdv_pcm16x0_bin_rec {
    uint32_t frame_number;
    uint16_t line_number;
    uint16_t words[4];              /* R1/P1/L1, L2/P2/R2, R3/P3/L3 (16 bit) + CRCC as read (or as the Bit Picker completed it) */
    uint16_t calc_crc;
    int16_t data_start, data_stop;  /* PCMLine::coords */
    uint16_t queue_order;           /* PCM16X0SubLine::queue_order: position of the video line in its field (videotodigital.cpp:1141) */
    uint8_t black_level, white_level, ref_low, ref_level, ref_high;
    uint8_t hysteresis_depth, shift_stage;
    uint8_t service_type;           /* SDV_SRV_* */
    uint8_t picked_bits_left, picked_bits_right;
    uint8_t flags;                  /* SDV_LF_* */
    uint8_t line_part;              /* PCM16X0SubLine::PART_LEFT / PART_MIDDLE / PART_RIGHT = 0 / 1 / 2 */
    uint8_t control_bit;            /* PCM16X0SubLine::control_bit */
    uint8_t _pad;
} sdv_pcm16x0_bin_rec;

/* Binarizer::processLine (binarizer.h:361, binarizer.cpp:443-1724) with a PCM16X0SubLine as output, for n_lines video lines in one launch:
 * the three passes the reference makes over one VideoLine (setLinePartMode PART_PCM16X0_LEFT / _MIDDLE / _RIGHT, videotodigital.cpp:902-925),
 * in that order, on line i = luma + i*row_stride (width bytes), numbered first_line + i*line_step of frame frame_number.  presets[3 i + part]
 * is what the caller had set on its Binarizer before that pass (setGoodParameters / setReferenceLevel / setDataCoordinates / setBWLevels,
 * binarizer.cpp:240-377, and the sticky do_ref_lvl_sweep member; presets == NULL: nothing preset); the passes of a line share the line's
 * VideoLine::scan_done mark like in the reference (a line whose coordinate search has run is not searched again, binarizer.cpp:5819-6042),
 * which starts cleared.  out_lines takes three records per line (lines_cap >= 3 n_lines), out_scan_done (or NULL) the mark behind each pass.
 * Mode, fine settings, coord_search, flags, service / empty lines, SDV_MODE_INSANE and errors as for sdv_pcm1_binarize_lines;
 * SDV_ERR_SHORT_LINE under 193 px.  The twin of sdv_pcm1_binarize_lines for callers that keep VideoToDigital's frame loop. */
int sdv_pcm16x0_binarize_lines(sdv_engine *e, const uint8_t *luma, size_t row_stride, int width, size_t n_lines,
                               const sdv_bin_state *presets, uint32_t frame_number, uint16_t first_line, uint16_t line_step,
                               unsigned flags, int coord_search, sdv_pcm16x0_bin_rec *out_lines, size_t lines_cap, uint8_t *out_scan_done, void *stream);

/* VideoToDigital::doBinarize (videotodigital.cpp:698-1815) with setPCMType(TYPE_PCM16X0) for a batch of whole frames: the frame
 * prescan (prescanCoordinates, :148-345: the right third of four lines, every mode but DRAFT), three Binarizer passes per video line
 * with the hand-over between the parts of a line (:1455-1511) and from line to line, forced-bad propagation inside a line
 * (:1168-1180), duplicate-line detection per part, coordinate damper and frame statistics.  out_lines takes
 * sdv_pcm16x0_binarize_records(height, n_frames, flags) records: three per video line, one per service line - what the worker pushes
 * into the deque<PCM16X0SubLine> that PCM16X0DataStitcher reads.  Everything else as for sdv_binarize_frames; SDV_ERR_SHORT_LINE under
 * 193 px.  SDV_MODE_INSANE: the reference level sweep for every pass that does not read from what was handed on (binarizer.cpp:1113-1120). */
size_t sdv_pcm16x0_binarize_records(int height, int n_frames, unsigned flags);
int sdv_pcm16x0_binarize_frames(sdv_engine *e, const uint8_t *luma, size_t row_stride, size_t frame_stride, int width, int height,
                                int n_frames, uint32_t first_frame_no, unsigned flags,
                                sdv_pcm16x0_bin_rec *out_lines, size_t lines_cap, sdv_frame_stats *out_stats, size_t stats_cap, void *stream);

/* VideoToDigital::doBinarize (videotodigital.cpp:698-1815) with setPCMType(TYPE_PCM1) for a batch of whole frames: the frame prescan
 * of the data coordinates (prescanCoordinates, :148-345; every mode but DRAFT), every line through Binarizer::processLine with
 * what the lines before it left preset, the coordinate-search switch of the real-time modes (:853-884), Header lines, duplicate-line
 * detection, coordinate damper and frame statistics.  Arguments, record order (one sdv_pcm1_bin_rec per video line and service line),
 * flags, capacities, stream state (sdv_reset_stream / sdv_get_chain_state / sdv_set_chain_state) and error codes as for
 * sdv_binarize_frames; SDV_ERR_SHORT_LINE under 94 px.  SDV_MODE_INSANE: the reference level sweep for every line that does not read
 * from what was handed on (binarizer.cpp:1105-1112).  out_lines is what the worker pushes
 * into the deque<PCM1Line> that PCM1DataStitcher reads. */
int sdv_pcm1_binarize_frames(sdv_engine *e, const uint8_t *luma, size_t row_stride, size_t frame_stride, int width, int height,
                             int n_frames, uint32_t first_frame_no, unsigned flags,
                             sdv_pcm1_bin_rec *out_lines, size_t lines_cap, sdv_frame_stats *out_stats, size_t stats_cap, void *stream);

/* ---- PCM-1 back half: PCM1DataStitcher (pcm1datastitcher.h:94-201) ------------------------------------------------- */
/* What PCM1DataStitcher reads of one PCM1Line (pcm1line.h:59-146, pcmline.h:137-186).  32 bytes. */
typedef struct sdv_pcm1_line_rec {
    uint32_t frame_number;          /* PCMLine::frame_number */
    uint16_t line_number;           /* PCMLine::line_number */
    uint16_t words[7];              /* PCM1Line::words: L2 R2 L4 R4 L6 R6 (13 bit) + CRCC as read */
    uint16_t calc_crc;              /* PCMLine::getCalculatedCRC() */
    uint8_t ref_level;              /* PCMLine::ref_level */
    uint8_t picked_bits_left, picked_bits_right;    /* PCM1Line::picked_bits_* (Bit Picker) */
    uint8_t service_type;           /* SDV_SRV_* (SDV_SRV_HEADER_LINE = PCM1Line::isServHeader()) */
    uint8_t flags;                  /* SDV_LF_BW_SET, SDV_LF_FORCED_BAD */
    uint8_t _pad[5];
} sdv_pcm1_line_rec;

/* PCM1DataStitcher slots (pcm1datastitcher.h:183-190); defaults of the constructor (pcm1datastitcher.cpp:18-31) */
typedef struct sdv_pcm1_stitch_settings {
    uint8_t field_order;            /* setFieldOrder: 1 = TFF (default), 2 = BFF */
    uint8_t auto_offset;            /* setAutoLineOffset (default on) */
    uint8_t use_ecc;                /* setFineUseECC (ignore_CRC = !use_ecc) */
    int8_t odd_offset, even_offset; /* setOddLineOffset / setEvenLineOffset */
    uint8_t _pad[3];
} sdv_pcm1_stitch_settings;

/* FrameAsmPCM1 (frametrimset.h:116-224) as emitted with guiUpdFrameAsm.  52 bytes. */
typedef struct sdv_frame_asm_pcm1 {
    uint32_t frame_number;
    uint16_t odd_std_lines, even_std_lines, odd_data_lines, even_data_lines, odd_valid_lines, even_valid_lines;
    uint16_t odd_top_data, odd_bottom_data, even_top_data, even_bottom_data, odd_sample_rate, even_sample_rate;
    uint16_t blocks_total, blocks_drop, samples_drop;
    uint16_t odd_top_padding, odd_bottom_padding, even_top_padding, even_bottom_padding, blocks_fix_bp;
    uint8_t field_order, odd_ref, even_ref, service_type;
    uint8_t flags;                  /* SDV_FA_ORDER_PRESET, SDV_FA_ORDER_GUESSED, SDV_FA1_ODD_EMPHASIS, SDV_FA1_EVEN_EMPHASIS */
    uint8_t _pad[3];
} sdv_frame_asm_pcm1;
enum { SDV_FA1_ODD_EMPHASIS = 1 << 2, SDV_FA1_EVEN_EMPHASIS = 1 << 3 };

void sdv_default_pcm1_stitch_settings(sdv_pcm1_stitch_settings *st);
int sdv_set_pcm1_stitch_settings(sdv_engine *e, const sdv_pcm1_stitch_settings *st);
/* PCM1DataStitcher::doFrameReassemble (pcm1datastitcher.cpp:1578-1772) over a span of the PCM-1 line stream: every complete
 * frame (lines up to its END_FRAME) is trimmed, split into sub-lines and fields, padded to 735 sub-lines per field and
 * deinterleaved into 8 blocks per field (PCM1Deinterleaver::processBlock, pcm1deinterleaver.cpp:69-278; PCM-1 has no error
 * correction); 1470 PCMSamplePairs and one FrameAsmPCM1 per frame (plus the NEW_FILE / END_FILE tags).  With automatic line
 * offsets frames are independent of each other, and there is no stream state apart from records that wait for their END_FRAME; with manual
 * offsets the stitcher's field buffers (what earlier frames left in them, pcm1datastitcher.cpp:896-909, 952-1016) live in the engine between
 * calls; sdv_set_pcm1_stitch_settings starts a fresh stitcher.
 * Same conventions as sdv_stitch_frames (device pointers, counts returned, SDV_ERR_UNSUPPORTED for lines of a later frame ahead of an END_FRAME).
 * A call that fails leaves the stream untouched - the lines of the call are not taken, lines that waited still wait - so it
 * can be repeated with the buffer sizes *n_pairs / *n_frames report. */
int sdv_pcm1_stitch_frames(sdv_engine *e, const sdv_pcm1_line_rec *lines, size_t n_lines, sdv_sample_pair *out_pairs, size_t pairs_cap,
                           size_t *n_pairs, sdv_frame_asm_pcm1 *out_frames, size_t frames_cap, size_t *n_frames, void *stream);

/* ---- what PCM1DataStitcher hands to the visualiser ----------------------------------------------------------------- */
/* One PCM1DataBlock as outputDataBlock emits it with newBlockProcessed (pcm1datastitcher.cpp:1333; pcm1datablock.h:84-98): the 184 words of an
 * interleave block (182 in the last block of a field) with the per-word flags PCM1Deinterleaver::setWordData left (pcm1deinterleaver.cpp:150-278).
 * 576 bytes.  stop_line of a field's last block is not comparable with the reference: it reads the sub-line one past the end of its queue
 * (:204-211 with stripe_len 46 for block 7 - index 735 of 735); the record holds the number the next line would have had. */
enum { SDV_P1B_SHORT = 1 << 0,          /* isShortLength(): words 182, 183 do not exist */
       SDV_P1B_EMPHASIS = 1 << 1 };     /* hasEmphasis() */
enum { SDV_P1W_CRC_OK = 1 << 0,         /* isWordCRCOk() / isWordValid() */
       SDV_P1W_PICKED_LEFT = 1 << 1,    /* hasPickedSample(): the sub-line's left word had bits picked (first sub-line of a line only) */
       SDV_P1W_PICKED_WORD = 1 << 2 };  /* hasPickedWord(): that, or the line's CRC was picked */
typedef struct sdv_pcm1_block_rec {
    uint32_t frame_number;          /* of the block's first sub-line */
    uint16_t start_line, stop_line; /* line numbers as the stitcher re-numbers the lines of its queue (addLinesFromField, :952-1016) */
    uint8_t interleave_num;         /* 0..7 within the field */
    uint8_t flags;                  /* SDV_P1B_* */
    uint16_t sample_rate;
    uint16_t words[184];            /* getWord(): 13-bit words, L R L R ... */
    uint8_t word_flags[184];        /* SDV_P1W_* */
    uint8_t _pad[12];
} sdv_pcm1_block_rec;
/* One PCM1SubLine of the stitcher's queue as performDeinterleave hands it to the visualiser (newLineProcessed, :1392-1407; pcm1subline.h:83-93):
 * a third of a PCM-1 line after trimming and padding.  16 bytes.  A frame has 2 x 735 places; the places of lines earlier frames left in the
 * field buffers (manual line offsets) are not handed over by the reference (their frame number is another one): SDV_P1S_SKIP. */
enum { SDV_P1S_BW_SET = 1 << 0, SDV_P1S_CRC_VALID = 1 << 1, SDV_P1S_SKIP = 1 << 7 };
typedef struct sdv_pcm1_asm_line_rec {
    uint32_t frame_number;
    uint16_t line_number;           /* re-numbered by the stitcher: 1, 3, 5 ... / 2, 4, 6 ... down the padded field */
    uint16_t words[2];              /* getLeft(), getRight() */
    uint8_t picked_bits_left, picked_bits_right;
    uint8_t line_part;              /* PCM1SubLine::PART_LEFT / _MIDDLE / _RIGHT */
    uint8_t flags;                  /* SDV_P1S_* */
    uint8_t _pad[2];
} sdv_pcm1_asm_line_rec;
/* With a block buffer set (device memory; NULL: off, the default) every sdv_pcm1_stitch_frames call also writes the blocks of its frames, 16 per
 * frame in the order of the sample pairs (block j of a frame = its pairs 92 j .. 92 j + 91, file tags aside); with a line buffer the 1470 sub-lines
 * per frame.  The counts of the last call (or what it needed, when it failed with SDV_ERR_BAD_ARG for lack of room). */
int sdv_set_pcm1_stitch_block_output(sdv_engine *e, sdv_pcm1_block_rec *out_blocks, size_t blocks_cap);
size_t sdv_pcm1_stitch_block_count(sdv_engine *e);
int sdv_set_pcm1_stitch_line_output(sdv_engine *e, sdv_pcm1_asm_line_rec *out_lines, size_t lines_cap);
size_t sdv_pcm1_stitch_line_count(sdv_engine *e);

/* The PCM1Line queue between the two halves: sdv_pcm1_binarize_frames writes sdv_pcm1_bin_rec (everything Binarizer::processLine leaves in a
 * PCM1Line), sdv_pcm1_stitch_frames reads sdv_pcm1_line_rec (what PCM1DataStitcher looks at).  Record i of `in` -> record i of `out`; device
 * pointers, asynchronous on `stream`. */
int sdv_pcm1_bin_to_line_recs(sdv_engine *e, const sdv_pcm1_bin_rec *in, size_t n, sdv_pcm1_line_rec *out, void *stream);

/* ---- PCM-16x0 back half: PCM16X0DataStitcher (pcm16x0datastitcher.h:100-327) ------------------------------------------- */
/* PCM16X0DataStitcher slots (pcm16x0datastitcher.h:304-314); defaults of the constructor and of setDefaultFineSettings
 * (pcm16x0datastitcher.cpp:3-33, 5636-5641) */
enum { SDV_P16_FORMAT_AUTO = 0, SDV_P16_FORMAT_SI = 1, SDV_P16_FORMAT_EI = 2 };     /* PCM16X0Deinterleaver::FORMAT_* (AUTO is handled as SI, :5784) */
typedef struct sdv_pcm16x0_stitch_settings {
    uint8_t format;                 /* setFormat: SDV_P16_FORMAT_* (default SI) */
    uint8_t field_order;            /* setFieldOrder: 1 = TFF (default), 2 = BFF */
    uint8_t p_correction;           /* setPCorrection (default on) */
    uint8_t use_ecc;                /* setFineUseECC (ignore_CRC = !use_ecc; default on) */
    uint8_t mask_seams;             /* setFineMaskSeams (default on) */
    uint8_t broke_mask;             /* setFineBrokeMask: blocks masked after a BROKEN one (default 81) */
    uint16_t sample_rate_preset;    /* setSampleRatePreset (1 = auto, 44056, 44100) */
} sdv_pcm16x0_stitch_settings;

/* FrameAsmPCM16x0 (frametrimset.h:116-249) as emitted with guiUpdFrameAsm.  56 bytes. */
typedef struct sdv_frame_asm_pcm16x0 {
    uint32_t frame_number;
    uint16_t odd_std_lines, even_std_lines, odd_data_lines, even_data_lines, odd_valid_lines, even_valid_lines;
    uint16_t odd_top_data, odd_bottom_data, even_top_data, even_bottom_data, odd_sample_rate, even_sample_rate;
    uint16_t blocks_total, blocks_drop, samples_drop;
    uint16_t odd_top_padding, odd_bottom_padding, even_top_padding, even_bottom_padding;
    uint16_t blocks_broken, blocks_fix_bp, blocks_fix_p, blocks_fix_cwd;
    uint8_t field_order, odd_ref, even_ref, service_type;
    uint8_t flags;                  /* SDV_FA_ORDER_PRESET, SDV_FA_ORDER_GUESSED, SDV_FA1_*_EMPHASIS, SDV_FA16_* */
    uint8_t _pad;
} sdv_frame_asm_pcm16x0;
enum { SDV_FA16_SILENCE = 1 << 4, SDV_FA16_PADDING_OK = 1 << 5, SDV_FA16_EI_FORMAT = 1 << 6 };

void sdv_default_pcm16x0_stitch_settings(sdv_pcm16x0_stitch_settings *st);
/* Applies the settings and starts a fresh PCM16X0DataStitcher (statistics and queued lines are dropped). */
int sdv_set_pcm16x0_stitch_settings(sdv_engine *e, const sdv_pcm16x0_stitch_settings *st);
/* PCM16X0DataStitcher's stream state as an opaque blob (the padding and Control Bit statistics rings, oldest entry first, the
 * Control Bit values of the last frame and what is left in conv_queue): checkpoints, and the hand-over to the engine of the next GPU of a sharded tape (DESIGN.md
 * section 7).  sdv_set_pcm16x0_stitch_state drops sub-lines of an unfinished frame.  sdv_saturate_pcm16x0_stitch_stats fills every ring
 * with its most frequent entry (an engine that joined the stream after a short warm-up, about to compare its state with the true one). */
size_t sdv_pcm16x0_stitch_state_size(void);
int sdv_get_pcm16x0_stitch_state(sdv_engine *e, void *out, size_t cap);
int sdv_set_pcm16x0_stitch_state(sdv_engine *e, const void *in, size_t n);
int sdv_saturate_pcm16x0_stitch_stats(sdv_engine *e);

/* PCM16X0DataStitcher::doFrameReassemble (pcm16x0datastitcher.cpp:5652-5856) over a span of the PCM-16x0 sub-line stream
 * (sdv_pcm16x0_binarize_frames' out_lines, service lines included): every complete frame (records up to its END_FRAME) is trimmed
 * (findFrameTrim :213), split into fields (:566), scanned for false-positive CRCs of the Bit Picker (:753), aligned - SI format:
 * per-field padding sweep with P-code checks and Control Bit positions (findSIPadding :1557); EI format: the padding between the
 * fields of the frame (findEIFrameStitching :3588) - , assembled with padding (fillFrameForOutput :4594) and deinterleaved with
 * P-code correction (PCM16X0Deinterleaver::processBlock, pcm16x0deinterleaver.cpp:128-708; performDeinterleave :5165) into three
 * PCMSamplePairs per data block: 1470 per frame, plus the NEW_FILE / END_FILE tags; one FrameAsmPCM16x0 per frame.  Stream state
 * (the padding and Control Bit histories, sub-lines that wait for their END_FRAME or for the rest of their interleave block)
 * lives in the engine.  Same conventions as sdv_pcm1_stitch_frames.
 * A frame whose lines arrive with sub-lines missing or doubled is queued with the reference's "WRONG COUNT" (:4699-4703): its fields
 * do not add up to 1470 sub-lines, performDeinterleave takes whole interleave rounds (105 sub-lines SI, 1470 EI; :5216) and the rest
 * stays in conv_queue, in front of the next frame - from there on every frame's blocks start in its predecessor, and a frame puts out
 * an extra round (105 / 1470 pairs more) whenever the rest has grown to one.  Reproduced as is; size `pairs_cap` for it (a call that does
 * not fit is refused with the needed sizes in *n_pairs / *n_frames and takes nothing). */
int sdv_pcm16x0_stitch_frames(sdv_engine *e, const sdv_pcm16x0_bin_rec *lines, size_t n_lines, sdv_sample_pair *out_pairs, size_t pairs_cap,
                              size_t *n_pairs, sdv_frame_asm_pcm16x0 *out_frames, size_t frames_cap, size_t *n_frames, void *stream);

/* One PCM16X0DataBlock as outputDataBlock hands it to the visualiser with newBlockProcessed (pcm16x0datastitcher.cpp:5116; pcm16x0datablock.h:120-140):
 * three sub-blocks of three words (two samples and their parity word, on three sub-lines 35 / 490 apart), after P-code correction and the
 * stitcher's seam / BROKEN masking.  Stored the way the class stores it, by line of the interleave (LINE_1, LINE_2, LINE_3); which line holds a
 * sub-block's left and right sample follows from the order flag (getWordToLine, pcm16x0datablock.cpp:1029-1155: with odd order sub-blocks 1 and 3 have
 * R on LINE_1 and L on LINE_3, sub-block 2 the other way round; even order swaps all three; LINE_2 is the parity word).  32 bytes.
 * Where the words came from (frame_number, start_line ... queue_order of the object) is not carried: the stitch kernels do not track it and RenderPCM
 * does not look at it. */
enum { SDV_P16B_EVEN_ORDER = 1 << 0, SDV_P16B_EI_FORMAT = 1 << 1, SDV_P16B_EMPHASIS = 1 << 2, SDV_P16B_CODE = 1 << 3 };
typedef struct sdv_pcm16x0_block_rec {
    uint16_t words[3][3];           /* [sub-block][line] */
    uint16_t word_crc;              /* bit 3 * sub-block + line: the word's sub-line passed its CRC (isWordCRCOk) */
    uint16_t word_valid;            /* ... the word is valid after correction and masking (isWordValid) */
    uint8_t picked_left;            /* bit line: hasPickedLeft(line) */
    uint8_t picked_crc;             /* bit line: hasPickedCRC(line) */
    uint8_t audio_state[3];         /* per sub-block: 0 AUD_ORIG, 1 AUD_FIX_P, 2 AUD_BROKEN */
    uint8_t flags;                  /* SDV_P16B_* */
    uint16_t sample_rate;
    uint8_t _pad[2];
} sdv_pcm16x0_block_rec;
/* With a block buffer set (device memory; NULL: off, the default) every sdv_pcm16x0_stitch_frames call also writes the blocks it turns into sample
 * pairs: block j of the call = its pairs 3 j .. 3 j + 2, file tags aside; sdv_frame_asm_pcm16x0::blocks_total / 3 of them belong to a frame
 * (the descriptor counts sub-blocks).  The count of the last call (or what it needed, when it failed with SDV_ERR_BAD_ARG for lack of room). */
int sdv_set_pcm16x0_stitch_block_output(sdv_engine *e, sdv_pcm16x0_block_rec *out_blocks, size_t blocks_cap);
size_t sdv_pcm16x0_stitch_block_count(sdv_engine *e);
/* The assembled sub-lines: what performDeinterleave hands to newLineProcessed before it decodes the frame (pcm16x0datastitcher.cpp:5196-5213) - the
 * sub-lines fillFrameForOutput queued for the frame (:4594-4690), trimmed and padded to 2 x 735.  They are written as records of the binarizer's own
 * type, because the window that shows them is drawn by the renderer of the lines window (renderAssembled->startPCM1600Frame + renderNewLine(PCM16X0SubLine),
 * mainwindow.cpp:2040-2044): sdv_vis_render_lines(SDV_VIS_PCM16X0_LINES) on this buffer draws the reference's "re-assembled" window.  A sub-line that
 * came from the stream is its input record with the queue order addLinesFromField gave it (:4470) and, where prescanForFalsePosCRCs forced it bad
 * (:800-820), SDV_LF_FORCED_BAD set - SDV_LF_CRC_VALID says what isCRCValid() answers now; a padding sub-line is a cleared PCM16X0SubLine (silent words, CRC word inverted,
 * Control Bit set, no levels or coordinates) with the frame's number, its part and the line number addFieldPadding counted to (:4552-4556).  Behind the
 * sub-lines of every frame one SDV_SRV_END_FRAME record (where MainWindow emits newFrameAssembled, :3956).  A frame that carries END_FILE queues nothing
 * and writes nothing.  Not covered: sub-lines a frame with the same number left in the queue (the reference would show them again).
 * Device memory; NULL: off, the default.  The count of the last call (or what it needed, when it failed with SDV_ERR_BAD_ARG for lack of room). */
int sdv_set_pcm16x0_stitch_line_output(sdv_engine *e, sdv_pcm16x0_bin_rec *out_lines, size_t lines_cap);
size_t sdv_pcm16x0_stitch_line_count(sdv_engine *e);


/* ---- AudioProcessor: dropout masking on the PCMSamplePair stream ---------------------------- */
/* AudioProcessor::DROP_* (audioprocessor.h:83-93), set with setMasking (audioprocessor.cpp:1532-1574) */
enum { SDV_DROP_IGNORE = 0, SDV_DROP_MUTE_BLOCK = 1, SDV_DROP_MUTE_WORD = 2, SDV_DROP_HOLD_BLOCK = 3, SDV_DROP_HOLD_WORD = 4,
       SDV_DROP_INTER_LIN_BLOCK = 5, SDV_DROP_INTER_LIN_WORD = 6, SDV_DROP_MAX = 7 };
/* The window constants of the class (audioprocessor.h:62-71): BUF_SIZE, MIN_VALID_BEFORE, MAX_RAMP_DOWN, MAX_RAMP_UP */
enum { SDV_AP_BUF_SIZE = 512, SDV_AP_MIN_VALID_BEFORE = 3, SDV_AP_MAX_RAMP_DOWN = 192, SDV_AP_MAX_RAMP_UP = 32 };

/* One purgePipeline() of the worker (audioprocessor.cpp:1716-1745: the output file is released, newSource is emitted and the
 * sample index starts over).  first_pair = how many pairs of this call's output had been put out when it happened: the pairs
 * from there up to the next event (or the end of the output) carry PCMSample::index 0, 1, 2, ... and, when the event is a
 * NEW_FILE tag, go into that source's WAV file.  16 bytes. */
enum { SDV_AP_PURGE_NEW_FILE = 1,   /* the NEW_FILE tag (:120-137): what waited in the window is put out as it is */
       SDV_AP_PURGE_END_FILE = 2,   /* the file ended (:138-152, :1338-1342) after the last window was scanned with the end-of-file rule */
       SDV_AP_PURGE_STOP = 3        /* stop() (:1655-1660) */ };
typedef struct sdv_audio_purge {
    uint64_t first_pair;
    uint32_t tag_index;             /* the position of the tag in this call's `pairs` (n_pairs for SDV_AP_PURGE_STOP) */
    uint8_t kind;                   /* SDV_AP_PURGE_* */
    uint8_t _pad[3];
} sdv_audio_purge;

/* setMasking(mode).  Like the slot it takes effect with the next window; it does not touch what waits in the window. */
int sdv_set_audio_masking(sdv_engine *e, int drop_mode);
/* A freshly constructed AudioProcessor (audioprocessor.cpp:3-36): empty window, sample index 0, masking as set. */
int sdv_reset_audio(sdv_engine *e);
/* How many pairs wait in the worker's window (prebuffer.size()), i.e. have been taken but not put out yet. */
size_t sdv_audio_pending(const sdv_engine *e);
/* 1 while the worker takes no more input: its window is full and the first pairs can never leave (fillUntilBufferFull :108). */
int sdv_audio_stalled(const sdv_engine *e);
/* PCMSample::index of the next pair that will be put out (it counts from 0 behind every purge and runs on across calls). */
uint64_t sdv_audio_next_index(const sdv_engine *e);

/* AudioProcessor::processAudio (audioprocessor.cpp:1621-1713) over a burst of the PCMSamplePair stream, i.e. what one of the
 * stitch entry points wrote (NEW_FILE / END_FILE tags included): the worker's loop of fillUntilBufferFull (:70-200: the window of
 * 512 pairs is topped up, pairs get their index, word validity is replaced by block validity in the *_BLOCK modes),
 * scanBuffer / fixBadSamples per channel (:740-1178: invalid regions found back to front; ramps of 192 / 32 samples into and out
 * of a mute for long ones, one region for short ones, filled by rangeMute / rangeLevelHold / rangeLinearInterpolation
 * :511-737; the end-of-file rule :1122-1172) and outputAudio (:1287-1357: pairs leave from the front while the first four are
 * valid or masked, three stay as look-behind), with purgePipeline at the tags.
 *
 * Feed schedule.  The reference polls its queue, so what it writes depends on how much the queue held at each turn.  This entry
 * fixes the schedule to the one a decoder faster than real time produces: the burst is in the queue before the worker's next
 * turn and the queue runs dry at the end of the call - every turn fills the window completely except the last one of the call,
 * which takes what is left (and, as in the reference, is scanned only when the window holds 227 pairs or the file ended).
 * A call with `stop` != 0 ends with stop(): the window is purged as it is (its last pair is dropped, dumpBuffer :1423-1434).
 *
 * out_pairs receives the pairs in the order of outputWordPair (:1265-1284) - service_type 0, flags as the worker left them
 * (SDV_SF_WORD_VALID set on everything a scan has seen, SDV_SF_WORD_MASKED on what was altered); out_purges the purge events in
 * order; *n_masked the sum of the guiAddMask reports.  *n_out / *n_purges receive the counts; when out_cap / purges_cap are too
 * small the call fails with SDV_ERR_BAD_ARG, the counts say what is needed (n_pairs + 512 + tags is always enough) and the
 * stream state is untouched, so the call can be repeated.  With room for n_pairs + 512 + tags pairs in out_pairs the pairs are worked on
 * there (what lies behind *n_out is scratch then) and cross HBM once in each direction; out_pairs may be `pairs` itself (or overlap it):
 * the burst then goes through a buffer of the engine first.
 * The two dead ends of the worker are reproduced as they are: an END_FILE that finds fewer than three pairs in the window neither
 * purges nor starts a new source (outputAudio returns early, :1298-1301) - the one or two pairs are scanned as the end of a file, stay
 * in the window and the stream goes on behind them; a full window whose first pairs can never leave (a stream that starts with invalid
 * samples and no NEW_FILE tag) makes the worker stop taking input for good (:108) - the call still succeeds, everything behind that
 * window - pairs and tags, of this call and of every later one - is dropped unread as the reference's queue would keep it forever,
 * sdv_audio_stalled() says so, and only stop() (the window is purged as it is) or sdv_reset_audio() get the stream going again.
 * Not supported (SDV_ERR_UNSUPPORTED, the stream state is left untouched): service tags other than NEW_FILE / END_FILE (PCMSamplePair has
 * no others, pcmsamplepair.h:99-104) and more than 65 536 tags in one burst.
 * All buffers are device pointers; the call returns when the outputs are complete. */
int sdv_audio_process(sdv_engine *e, const sdv_sample_pair *pairs, size_t n_pairs, int stop, sdv_sample_pair *out_pairs, size_t out_cap,
                      size_t *n_out, sdv_audio_purge *out_purges, size_t purges_cap, size_t *n_purges, uint64_t *n_masked, void *stream);

/* ---- SamplesToWAV: the file the GUI writes -------------------------------------------------- */
/* SamplesToWAV::saveAudio (samples2wav.cpp:306-323) for n pairs: left and right audio_word, little-endian, 4 bytes per pair, to
 * `pcm` (device pointers; asynchronous on `stream`). */
int sdv_wav_pack(sdv_engine *e, const sdv_sample_pair *pairs, size_t n, int16_t *pcm, void *stream);
/* The 44-byte RIFF header as SamplesToWAV leaves it (default_header :4-21, updateHeader :111-206) on a file that holds n_pairs
 * pairs whose last one had sample_rate `last_sample_rate` (setSampleRate :257-289: 44056 stays, anything else reads 44100).
 * Host memory. */
void sdv_wav_header(uint8_t hdr[44], uint64_t n_pairs, uint16_t last_sample_rate);

/* ---- the workers back to back: video frames -> PCMSamplePair (-> masked PCMSamplePair) in one call ------------------- */
/* SURVEY 8b's "sdv_decode_frames": the format's VideoToDigital worker, its data stitcher and - on request - the AudioProcessor, one
 * after the other on `stream`, with the line records (and the raw pair stream) in buffers the engine owns, so that only the luma
 * goes in and only the sample pairs come out.  pcm_type selects the chain (what setPCMType selects in the application):
 *   SDV_PCM_STC007   sdv_binarize_frames          -> sdv_stitch_frames           out_frames: sdv_frame_asm[]         (64 B)
 *   SDV_PCM_PCM1     sdv_pcm1_binarize_frames     -> sdv_pcm1_bin_to_line_recs -> sdv_pcm1_stitch_frames
 *                                                                                out_frames: sdv_frame_asm_pcm1[]    (52 B)
 *   SDV_PCM_PCM16X0  sdv_pcm16x0_binarize_frames  -> sdv_pcm16x0_stitch_frames   out_frames: sdv_frame_asm_pcm16x0[] (56 B)
 * Arguments up to `flags` as for sdv_binarize_frames (SDV_FLAG_NEW_FILE / SDV_FLAG_END_FILE put the file tags into the stream);
 * every setting comes from the same setters as for the separate calls (sdv_set_mode, sdv_set_bin_preset, sdv_set_stitch_settings,
 * sdv_set_pcm1_stitch_settings, sdv_set_pcm16x0_stitch_settings, sdv_set_audio_masking), and the stream state of every stage lives on
 * in the engine exactly as if the separate entry points had been called.  out_stats (n_frames records, + 1 with SDV_FLAG_END_FILE) may
 * be NULL.  with_audio != 0: the pair stream goes through sdv_audio_process (stop = audio_stop) before it is handed out; out_purges /
 * n_purges / n_masked then receive what that call reports (NULL otherwise).  The result is what the separate calls give - the parity
 * tests compare the two.  A failure of a later stage leaves the earlier stages' stream state advanced (the frames were binarized):
 * after an error other than SDV_ERR_BAD_ARG for a too small buffer of the last stage, reset the streams. */
int sdv_decode_frames(sdv_engine *e, int pcm_type, const uint8_t *luma, size_t row_stride, size_t frame_stride, int width, int height,
                      int n_frames, uint32_t first_frame_no, unsigned flags,
                      sdv_sample_pair *out_pairs, size_t pairs_cap, size_t *n_pairs, void *out_frames, size_t frames_cap, size_t *n_frames_out,
                      sdv_frame_stats *out_stats, size_t stats_cap,
                      int with_audio, int audio_stop, sdv_audio_purge *out_purges, size_t purges_cap, size_t *n_purges, uint64_t *n_masked, void *stream);

/* One assembled STC007Line as STC007DataStitcher hands it to the visualiser (newLineProcessed, stc007datastitcher.cpp:6689-6704): a line of conv_queue
 * after padding and CWD, with the per-word states the window colours by.  32 bytes. */
enum { SDV_AL_FORCED_BAD = 1 << 0,      /* STC007Line::isForcedBad() */
       SDV_AL_MARKERS = 1 << 1,         /* hasMarkers() */
       SDV_AL_CRC_VALID = 1 << 2 };     /* isCRCValid() */
typedef struct sdv_asm_line_rec {
    uint32_t frame_number;
    uint16_t line_number;
    uint16_t words[9];              /* getWord(0..8) */
    uint16_t calc_crc;
    uint16_t word_crc_ok;           /* bit i = isWordCRCOk(i), i = 0..8 (all clear on a line that is forced bad) */
    uint16_t word_valid;            /* bit i = isWordValid(i) */
    uint8_t flags;                  /* SDV_AL_* */
    uint8_t _pad;
} sdv_asm_line_rec;

/* The data blocks of the stitch stage, for the visualiser: STC007DataStitcher::outputDataBlock hands every block it has turned into three sample
 * pairs to `newBlockProcessed(STC007DataBlock)` (stc007datastitcher.cpp:6626).  With a block buffer set (device memory, blocks_cap records; NULL:
 * off, the default), every sdv_stitch_frames call also writes those blocks - as it leaves them: seam / BROKEN masking applied, sample rate set -
 * in stream order: block j belongs to the sample pairs 3 j .. 3 j + 2 of the call, file-tag pairs not counted.  sdv_stitch_block_count: how many the
 * last call wrote (or needed, when it failed with SDV_ERR_BAD_ARG for lack of room).  Costs one more launch of the turn kernel per call. */
int sdv_set_stitch_block_output(sdv_engine *e, sdv_block_rec *out_blocks, size_t blocks_cap);
size_t sdv_stitch_block_count(sdv_engine *e);
/* ... and the assembled lines the stitcher hands to the visualiser at the start of every turn's deinterleave (newLineProcessed: the lines of conv_queue that
 * belong to the turn's two frames, after padding and CWD; 490 per NTSC turn on a tape that plays): with a line buffer set every sdv_stitch_frames call also
 * writes them, turn after turn.  sdv_stitch_line_count: how many the last call wrote (or needed); sdv_stitch_line_counts: how many each turn of the last
 * call made (a host array of `cap` entries is filled; returns the number of turns) - one turn per frame descriptor that is no file tag.  Costs two more
 * launches of the turn kernel per call. */
int sdv_set_stitch_line_output(sdv_engine *e, sdv_asm_line_rec *out_lines, size_t lines_cap);
size_t sdv_stitch_line_count(sdv_engine *e);
size_t sdv_stitch_line_counts(sdv_engine *e, uint32_t *per_turn, size_t cap);

/* ---- visualiser feed: the canvases of RenderPCM's "binarized lines" window ---------------------------------------------------
 * Replaces RenderPCM::renderNewLine(STC007Line / PCM1Line / PCM16X0SubLine) (renderpcm.cpp:939-1169, 489-624, 743-936) as MainWindow drives it
 * for its binarized-lines visualiser (mainwindow.cpp:1949-1990): every line VideoToDigital queues that is no service line, fillers
 * included (videotodigital.cpp:398-402, 452-456, 507-511), is drawn on the next row of one canvas, bit by bit in the colours of
 * renderpcm.h:52-65 (5 / 8 / 4 pixels per bit); at every binarized frame (an END_FRAME record here) the canvas is handed out and the
 * next frame starts at row 0 again (prepareNewFrame, renderpcm.cpp:176-186) - on the same canvas, which is never cleared.
 *   kind                     records (device)          canvas (sdv_vis_canvas_size)
 *   SDV_VIS_STC007_LINES     sdv_line_rec              685 x 650   (startSTC007NTSCFrame + setLineCount(VID_UNKNOWN))
 *   SDV_VIS_PCM1_LINES       sdv_pcm1_bin_rec          752 x 490   (startPCM1Frame)
 *   SDV_VIS_PCM16X0_LINES    sdv_pcm16x0_bin_rec       772 x 490   (startPCM1600Frame; three sub-lines and the control bit per row)
 * - the record streams the frame entries (sdv_binarize_frames, sdv_pcm1_binarize_frames, sdv_pcm16x0_binarize_frames) put out.  One call
 * draws all frames that end in `recs`: out_canvases[f] is frame f's canvas (height x width pixels of 32 bit, QImage::Format_RGB32 as the
 * reference fills it: 0xFFRRGGBB, and the value 2 where the reference writes its VIS_BIT0_BLK, which is the enumerator Qt::black).  The engine
 * keeps the last canvas per kind for the next call like RenderPCM keeps its QImage; a canvas nothing was drawn on yet is 0xFF000000 (the
 * reference leaves those rows uninitialised).  Records behind the last END_FRAME are not drawn: pass whole frames.  *n_frames = frames in
 * `recs`; more than canvases_cap: SDV_ERR_BAD_ARG, nothing drawn.  Device pointers; the call reads one small array back (the frame count)
 * and leaves the drawing running on `stream`.  sdv_vis_reset: a new canvas (RenderPCM::startNewFrame).
 *   SDV_VIS_PCM1_ASM         sdv_pcm1_asm_line_rec     624 x 490   (startPCM1SubFrame: the assembled-lines window of PCM-1, renderNewLine(PCM1SubLine),
 *                                                                   renderpcm.cpp:626-741, on what sdv_set_pcm1_stitch_line_output wrote: 1470 records are a
 *                                                                   frame - no END_FRAME records here -, records marked SDV_P1S_SKIP are not drawn)
 * The block canvases and the assembled-lines canvas of STC-007 follow below. */
enum { SDV_VIS_STC007_LINES = 0, SDV_VIS_PCM1_LINES = 1, SDV_VIS_PCM16X0_LINES = 2, SDV_VIS_STC007_BLOCKS_NTSC = 3, SDV_VIS_STC007_BLOCKS_PAL = 4,
       SDV_VIS_STC007_ASM_NTSC = 5, SDV_VIS_STC007_ASM_PAL = 6,
       SDV_VIS_PCM1_BLOCKS = 7,         /* sdv_vis_render_blocks on sdv_pcm1_block_rec: 858 x 368 (startPCM1DBFrame), 23 rows of 8 words per block */
       SDV_VIS_PCM1_ASM = 8,            /* sdv_vis_render_lines on sdv_pcm1_asm_line_rec: 624 x 490 (startPCM1SubFrame), three sub-lines per row */
       SDV_VIS_PCM16X0_BLOCKS = 9,      /* sdv_vis_render_blocks on sdv_pcm16x0_block_rec: 678 x 490 (startPCM1600DBFrame), one row per block */
       SDV_VIS_M2_SAMPLES = 0x100 };   /* or-ed to a block canvas: the blocks hold M2 samples (STC007DataBlock::setM2Format; getSample's M2 branch, stc007datablock.cpp:527-556) */
int sdv_vis_canvas_size(int kind, uint32_t *width, uint32_t *height);
int sdv_vis_reset(sdv_engine *e, int kind, void *stream);
int sdv_vis_render_lines(sdv_engine *e, int kind, const void *recs, size_t n_recs, uint32_t *out_canvases, size_t canvases_cap, size_t *n_frames, void *stream);
/* The data blocks window: RenderPCM::renderNewBlock(STC007DataBlock) (renderpcm.cpp:1770-2051) as MainWindow drives it (mainwindow.cpp:2070-2113):
 * one row per block - six status bits (P / Q / CWD corrections, block validity, near silence), the six 16-bit samples with every bit in the colour
 * of its word's state, seven tail bits (seam, emphasis, BROKEN), 6 pixels per bit - and a prepareNewFrame per assembled frame.  kind:
 * SDV_VIS_STC007_BLOCKS_NTSC (654 x 490) or _PAL (654 x 588; the canvas follows the video standard, setLineCount).  blocks: the device
 * buffer sdv_set_stitch_block_output filled; frame_blocks: HOST array, how many of them belong to each of the n_frames frames
 * (sdv_frame_asm::blocks_total of the descriptors that are no file tags); blocks past the canvas' rows are dropped like in the reference.
 * out_canvases[f] = the canvas after frame f; the engine keeps the last one per kind.  kind | SDV_VIS_M2_SAMPLES: the samples are drawn as
 * getSample() expands M2 words (12 bits + range bit) and "near silence" is judged on the 16-bit scale, as for blocks with setM2Format(true).
 * Not covered: emphasis (the reference never sets it for STC-007, stc007datastitcher.cpp:6719).  Device pointers but frame_blocks; asynchronous on `stream`.
 * SDV_VIS_PCM1_BLOCKS: renderNewBlock(PCM1DataBlock) (renderpcm.cpp:1171-1400) on the sdv_pcm1_block_rec buffer sdv_set_pcm1_stitch_block_output
 * filled (frame_blocks[f] = 16): a block is 23 rows of eight words - per row a status bar (picked / invalid marks per word, block validity, near
 * silence), the eight 16-bit samples, the block's parity in the field and the emphasis mark.
 * SDV_VIS_PCM16X0_BLOCKS: renderNewBlock(PCM16X0DataBlock) (:1403-1768) on sdv_pcm16x0_block_rec (sdv_set_pcm16x0_stitch_block_output; frame_blocks[f] =
 * sdv_frame_asm_pcm16x0::blocks_total / 3 - the descriptor counts sub-blocks): one row per block - Bit Picker / P-correction marks per sub-block, the six samples, format and BROKEN marks.
 * `blocks` points to the record type of the kind. */
int sdv_vis_render_blocks(sdv_engine *e, int kind, const void *blocks, size_t n_blocks, const uint32_t *frame_blocks, size_t n_frames,
                          uint32_t *out_canvases, size_t canvases_cap, void *stream);
/* The assembled-lines window (renderAssembled, mainwindow.cpp:2000-2052): RenderPCM::renderNewLine(STC007Line) on the lines the stitcher hands over -
 * the line buffer sdv_set_stitch_line_output filled - where every word has its own state after the CWD pass (grey: read, green: repaired, yellow / red:
 * failed with / without markers, magenta: line forced bad).  kind: SDV_VIS_STC007_ASM_NTSC (685 x 490) / _PAL (685 x 588); frame_lines: HOST array,
 * lines per frame (sdv_stitch_line_counts).  Otherwise as sdv_vis_render_blocks. */
int sdv_vis_render_asm_lines(sdv_engine *e, int kind, const sdv_asm_line_rec *lines, size_t n_lines, const uint32_t *frame_lines, size_t n_frames,
                             uint32_t *out_canvases, size_t canvases_cap, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SDVPCM_H */
