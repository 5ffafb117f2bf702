/*
 * ref_render_driver.cpp - thin C-ABI shim around the REAL RenderPCM (renderpcm.cpp, compiled from /root/reference by oracle/Makefile.ref
 * into oracle/_ref/libsdvref.so).  TEST INFRASTRUCTURE ONLY.  Our own code: it only calls the reference's public slots the way
 * MainWindow wires its "binarized" visualiser (mainwindow.cpp:1949-1990): start<format>Frame, a renderNewLine per line that is not a service
 * line (fillers are drawn), prepareNewFrame per frame; frames leave through the renderedFrame(QImage) signal.
 */
#include <cstring>
#include <cstdint>
#include <vector>
#include <QCoreApplication>
#include <QImage>
#include "renderpcm.h"
#include "frametrimset.h"
#include "../include/sdvpcm.h"

static int q_argc = 1;
static char q_arg0[] = "sdvref";
static char *q_argv[] = { q_arg0, NULL };

static void service_of(PCMLine &l, uint8_t st)
{
    if (st == SDV_SRV_FILLER) l.setServFiller();
}

static void to_line(const sdv_line_rec &r, STC007Line &l)
{
    l.clear();
    l.frame_number = r.frame_number; l.line_number = r.line_number;
    if (r.service_type == SDV_SRV_FILLER) { l.setServFiller(); return; }
    for (uint8_t i = 0; i < 8; i++) l.setWord(i, r.words[i]);
    l.setSourceCRC(r.words[8]); l.calcCRC();
    l.setBWLevelsState((r.flags & SDV_LF_BW_SET) != 0);
    l.setDataCoordinatesState((r.flags & SDV_LF_COORDS_SET) != 0);
    l.mark_st_stage = r.mark_st_stage; l.mark_ed_stage = r.mark_ed_stage;
    if (r.flags & SDV_LF_FORCED_BAD) l.setForcedBad();
    l.applyCRCStatePerWord();
}

static void to_line(const sdv_pcm1_bin_rec &r, PCM1Line &l)
{
    l.clear();
    l.frame_number = r.frame_number; l.line_number = r.line_number;
    if (r.service_type == SDV_SRV_FILLER) { l.setServFiller(); return; }
    for (uint8_t i = 0; i < 6; i++) l.setWord(i, r.words[i]);
    l.setSourceCRC(r.words[6]); l.calcCRC();
    l.picked_bits_left = r.picked_bits_left; l.picked_bits_right = r.picked_bits_right;
    l.setBWLevelsState((r.flags & SDV_LF_BW_SET) != 0);
    l.setDataCoordinatesState((r.flags & SDV_LF_COORDS_SET) != 0);
    if (r.flags & SDV_LF_FORCED_BAD) l.setForcedBad();
}

static void to_line(const sdv_pcm16x0_bin_rec &r, PCM16X0SubLine &l)
{
    l.clear();
    l.frame_number = r.frame_number; l.line_number = r.line_number;
    if (r.service_type == SDV_SRV_FILLER) { l.setServFiller(); return; }
    for (uint8_t i = 0; i < 3; i++) l.setWord(i, r.words[i]);
    l.setSourceCRC(r.words[3]); l.calcCRC();
    l.picked_bits_left = r.picked_bits_left; l.picked_bits_right = r.picked_bits_right;
    l.control_bit = r.control_bit != 0; l.line_part = r.line_part; l.queue_order = r.queue_order;
    l.setBWLevelsState((r.flags & SDV_LF_BW_SET) != 0);
    l.setDataCoordinatesState((r.flags & SDV_LF_COORDS_SET) != 0);
    if (r.flags & SDV_LF_FORCED_BAD) l.setForcedBad();
}

/* kind: 0 STC-007 (sdv_line_rec), 1 PCM-1 (sdv_pcm1_bin_rec), 2 PCM-16x0 (sdv_pcm16x0_bin_rec).  The canvas of every frame (at its END_FRAME
 * record) goes to out[frame] as width*height 32-bit pixels; rows no frame has reached yet are whatever `new QImage` left there.  Returns frames. */
extern "C" long ref_vis_render_lines(int kind, const void *recs, size_t n_recs, uint32_t *out, size_t out_cap, uint32_t *width, uint32_t *height)
{
    if (!QCoreApplication::instance()) new QCoreApplication(q_argc, q_argv);
    RenderPCM ren;
    long frames = 0;
    uint32_t w = 0, h = 0;
    QObject::connect(&ren, &RenderPCM::renderedFrame, [&](QImage img) {
        w = (uint32_t)img.width(); h = (uint32_t)img.height();
        if ((size_t)frames < out_cap)
            for (uint32_t y = 0; y < h; y++) memcpy(out + ((size_t)frames * h + y) * w, img.constScanLine((int)y), (size_t)w * 4);
        frames++;
    });
    ren.setLivePlay(false);
    if (kind == 8) {        /* the assembled-lines window of PCM-1 (mainwindow.cpp:2034-2040): renderNewLine(PCM1SubLine) on the sub-lines the stitcher handed over
                             * (the records not marked SDV_P1S_SKIP), prepareNewFrame behind every frame's 1470 places */
        ren.startPCM1SubFrame();
        const sdv_pcm1_asm_line_rec *l = (const sdv_pcm1_asm_line_rec *)recs;
        PCM1SubLine sl;
        for (size_t i = 0; i < n_recs; i++) {
            if (!(l[i].flags & SDV_P1S_SKIP)) {
                sl.clear();
                sl.frame_number = l[i].frame_number; sl.line_number = l[i].line_number;
                sl.picked_bits_left = l[i].picked_bits_left; sl.picked_bits_right = l[i].picked_bits_right;
                sl.setLinePart(l[i].line_part); sl.setLeft(l[i].words[0]); sl.setRight(l[i].words[1]);
                sl.setBWLevels((l[i].flags & SDV_P1S_BW_SET) != 0); sl.setCRCValid((l[i].flags & SDV_P1S_CRC_VALID) != 0);
                ren.renderNewLine(sl);
            }
            if ((i + 1) % 1470 == 0) { ren.prepareNewFrame((uint32_t)(i / 1470)); ren.displayIsReady(); }
        }
        if (width) *width = w;
        if (height) *height = h;
        return frames;
    }
    if (kind == 0) { ren.startSTC007NTSCFrame(); ren.setLineCount(FrameAsmDescriptor::VID_UNKNOWN); }
    else if (kind == 1) ren.startPCM1Frame();
    else ren.startPCM1600Frame();
    STC007Line l0; PCM1Line l1; PCM16X0SubLine l2;
    for (size_t i = 0; i < n_recs; i++) {
        uint8_t srv; uint32_t frame;
        if (kind == 0) { const sdv_line_rec &r = ((const sdv_line_rec *)recs)[i]; srv = r.service_type; frame = r.frame_number; }
        else if (kind == 1) { const sdv_pcm1_bin_rec &r = ((const sdv_pcm1_bin_rec *)recs)[i]; srv = r.service_type; frame = r.frame_number; }
        else { const sdv_pcm16x0_bin_rec &r = ((const sdv_pcm16x0_bin_rec *)recs)[i]; srv = r.service_type; frame = r.frame_number; }
        if (srv == SDV_SRV_END_FRAME) { ren.prepareNewFrame(frame); ren.displayIsReady(); continue; }
        if (srv != SDV_SRV_NO && srv != SDV_SRV_FILLER) continue;          /* videotodigital.cpp:398-402, 452-456, 507-511 */
        if (kind == 0) { to_line(((const sdv_line_rec *)recs)[i], l0); ren.renderNewLine(l0); }
        else if (kind == 1) { to_line(((const sdv_pcm1_bin_rec *)recs)[i], l1); ren.renderNewLine(l1); }
        else { to_line(((const sdv_pcm16x0_bin_rec *)recs)[i], l2); ren.renderNewLine(l2); }
    }
    if (width) *width = w;
    if (height) *height = h;
    return frames;
}

/* ---- the data blocks window: renderNewBlock(STC007DataBlock) as MainWindow wires it (mainwindow.cpp:2070-2113): startSTC007DBFrame, setLineCount
 * with the video standard, a renderNewBlock per block the stitcher put out, prepareNewFrame per assembled frame. ----------------------------- */
#include "stc007datablock.h"
/* an sdv_block_rec back into the reference's object through its public interface; returns false for a record the interface cannot express
 * (a word whose line passed its CRC but that is not valid) */
static bool to_block(const sdv_block_rec &r, STC007DataBlock &b)
{
    bool ok = true;
    b.clear();
    b.setResolution(r.resolution);
    for (uint8_t i = 0; i < 8; i++) {
        const bool crc = (r.line_crc >> i) & 1, cwd = (r.cwd_fixed >> i) & 1, valid = (r.word_valid >> i) & 1;
        b.setWord(i, r.words[i], crc, cwd);             /* word_valid = line_crc */
        if (valid && !crc) b.setFixed(i);
        if (!valid && crc) ok = false;
        b.setSource(i, r.w_frame[i], r.w_line[i]);
    }
    b.setAudioState(r.audio_state);
    b.cwd_applied = r.cwd_applied != 0; b.sample_rate = r.sample_rate;
    return ok;
}

/* an sdv_pcm1_block_rec back into a PCM1DataBlock through its public interface: setWord takes the two Bit Picker flags apart (picked_left, picked_crc);
 * the record holds hasPickedSample (= picked_left) and hasPickedWord (= either) - picked_crc = hasPickedWord gives the same two answers */
#include "pcm1datablock.h"
static void to_pcm1_block(const sdv_pcm1_block_rec &r, PCM1DataBlock &b)
{
    b.clear();
    b.frame_number = r.frame_number; b.start_line = r.start_line; b.stop_line = r.stop_line; b.interleave_num = r.interleave_num; b.sample_rate = r.sample_rate;
    if (r.flags & SDV_P1B_SHORT) b.setShortLength(); else b.setNormalLength();     /* (first: setShortLength wipes the words from 181 on, pcm1datablock.cpp:87-98) */
    for (int i = 0; i < PCM1DataBlock::WORD_CNT; i++)                               /* (setWord ignores what a short block does not have) */
        b.setWord((uint8_t)i, r.words[i], (r.word_flags[i] & SDV_P1W_CRC_OK) != 0, (r.word_flags[i] & SDV_P1W_PICKED_LEFT) != 0, (r.word_flags[i] & SDV_P1W_PICKED_WORD) != 0);
    b.setEmphasis((r.flags & SDV_P1B_EMPHASIS) != 0);
}
/* kind: 3 NTSC (490 rows), 4 PAL (588 rows), 7 PCM-1 (sdv_pcm1_block_rec, 23 rows per block).  frame_blocks[f] blocks belong to frame f.  Returns the
 * frames, -2 if a record cannot be expressed. */
extern "C" long ref_vis_render_blocks(int kind, const void *blocks_, size_t n_blocks, const uint32_t *frame_blocks, size_t n_frames,
                                      uint32_t *out, size_t out_cap, uint32_t *width, uint32_t *height)
{
    const sdv_block_rec *blocks = (const sdv_block_rec *)blocks_;
    if (!QCoreApplication::instance()) new QCoreApplication(q_argc, q_argv);
    RenderPCM ren;
    long frames = 0;
    uint32_t w = 0, h = 0;
    QObject::connect(&ren, &RenderPCM::renderedFrame, [&](QImage img) {
        w = (uint32_t)img.width(); h = (uint32_t)img.height();
        if ((size_t)frames < out_cap)
            for (uint32_t y = 0; y < h; y++) memcpy(out + ((size_t)frames * h + y) * w, img.constScanLine((int)y), (size_t)w * 4);
        frames++;
    });
    ren.setLivePlay(false);
    if (kind == 7) {        /* mainwindow.cpp:2097-2101: startPCM1DBFrame, renderNewBlock(PCM1DataBlock), prepareNewFrame per assembled frame */
        ren.startPCM1DBFrame();
        const sdv_pcm1_block_rec *pb = (const sdv_pcm1_block_rec *)blocks_;
        PCM1DataBlock b1;
        size_t at1 = 0;
        for (size_t f = 0; f < n_frames; f++) {
            for (uint32_t i = 0; i < frame_blocks[f] && at1 < n_blocks; i++, at1++) { to_pcm1_block(pb[at1], b1); ren.renderNewBlock(b1); }
            ren.prepareNewFrame((uint32_t)f); ren.displayIsReady();
        }
        if (width) *width = w;
        if (height) *height = h;
        return frames;
    }
    ren.startSTC007DBFrame();
    const bool m2 = (kind & 0x100) != 0;           /* SDV_VIS_M2_SAMPLES: the blocks of a stream in M2 sample format (STC007DataStitcher sets it on every block) */
    kind &= 0xFF;
    ren.setLineCount(kind == 4 ? FrameAsmDescriptor::VID_PAL : FrameAsmDescriptor::VID_NTSC);
    STC007DataBlock b;
    size_t at = 0;
    for (size_t f = 0; f < n_frames; f++) {
        for (uint32_t i = 0; i < frame_blocks[f] && at < n_blocks; i++, at++) {
            if (!to_block(blocks[at], b)) return -2;
            b.setM2Format(m2);
            ren.renderNewBlock(b);
        }
        ren.prepareNewFrame((uint32_t)f); ren.displayIsReady();
    }
    if (width) *width = w;
    if (height) *height = h;
    return frames;
}


/* ---- the assembled-lines window: renderNewLine(STC007Line) on the stitcher's lines, as MainWindow wires renderAssembled (mainwindow.cpp:2000-2052) -- */
static bool to_asm_line(const sdv_asm_line_rec &r, STC007Line &l)
{
    bool ok = true;
    l.clear();
    l.frame_number = r.frame_number; l.line_number = r.line_number;
    for (uint8_t i = 0; i < 9; i++) {
        const bool crc = (r.word_crc_ok >> i) & 1, valid = (r.word_valid >> i) & 1;
        l.setWord(i, r.words[i], crc);                  /* word_crc = word_valid = crc */
        if (valid && !crc) l.setFixed(i);
        if (!valid && crc) ok = false;
    }
    l.calcCRC();
    if (r.flags & SDV_AL_MARKERS) l.forceMarkersOk();
    if (r.flags & SDV_AL_FORCED_BAD) {
        l.setForcedBad();
        if (r.word_crc_ok | r.word_valid) ok = false;
    }
    if (((r.flags & SDV_AL_CRC_VALID) != 0) != l.isCRCValid()) ok = false;
    return ok;
}
/* kind: 5 NTSC (490 rows), 6 PAL (588 rows).  Returns the frames; -2: a record the line object cannot express, -3: its CRC state differs. */
extern "C" long ref_vis_render_asm_lines(int kind, const sdv_asm_line_rec *lines, size_t n_lines, const uint32_t *frame_lines, size_t n_frames,
                                         uint32_t *out, size_t out_cap, uint32_t *width, uint32_t *height)
{
    if (!QCoreApplication::instance()) new QCoreApplication(q_argc, q_argv);
    RenderPCM ren;
    long frames = 0;
    uint32_t w = 0, h = 0;
    QObject::connect(&ren, &RenderPCM::renderedFrame, [&](QImage img) {
        w = (uint32_t)img.width(); h = (uint32_t)img.height();
        if ((size_t)frames < out_cap)
            for (uint32_t y = 0; y < h; y++) memcpy(out + ((size_t)frames * h + y) * w, img.constScanLine((int)y), (size_t)w * 4);
        frames++;
    });
    ren.setLivePlay(false);
    ren.startSTC007NTSCFrame();
    ren.setLineCount(kind == 6 ? FrameAsmDescriptor::VID_PAL : FrameAsmDescriptor::VID_NTSC);
    STC007Line l;
    size_t at = 0;
    for (size_t f = 0; f < n_frames; f++) {
        for (uint32_t i = 0; i < frame_lines[f] && at < n_lines; i++, at++) {
            if (!to_asm_line(lines[at], l)) return -2;
            ren.renderNewLine(l);
        }
        ren.prepareNewFrame((uint32_t)f); ren.displayIsReady();
    }
    if (width) *width = w;
    if (height) *height = h;
    return frames;
}

/* ---- the data blocks window of PCM-16x0, both ends real: PCM16X0DataStitcher's newBlockProcessed blocks go straight into RenderPCM::renderNewBlock, a
 * prepareNewFrame per frame the stitcher reports (mainwindow.cpp:2102-2106: startPCM1600DBFrame; newFrameAssembled -> prepareNewFrame).  The
 * object cannot be put together again from a record through its public interface (a word that passed its CRC and is not valid comes out of
 * markAsUnsafe only), so the canvases are made where the objects are. */
#include <functional>
#include "pcm16x0datastitcher.h"
struct ref_p16_hooks { std::function<void(PCM16X0DataBlock &)> on_block; std::function<void(uint32_t)> on_frame; std::function<void(PCM16X0SubLine &)> on_line; };
long ref_pcm16x0_stitch_run_hooks(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st, sdv_sample_pair *out, size_t out_cap,
                                  sdv_frame_asm_pcm16x0 *frames, size_t frames_cap, size_t *n_frames, ref_p16_hooks *hooks);
extern "C" long ref_vis_pcm16x0_stitch_block_canvases(const sdv_pcm16x0_bin_rec *recs, size_t n_recs, const sdv_pcm16x0_stitch_settings *st,
                                                      uint32_t *out, size_t out_cap, uint32_t *width, uint32_t *height)
{
    if (!QCoreApplication::instance()) new QCoreApplication(q_argc, q_argv);
    RenderPCM ren;
    long frames = 0;
    uint32_t w = 0, h = 0;
    QObject::connect(&ren, &RenderPCM::renderedFrame, [&](QImage img) {
        w = (uint32_t)img.width(); h = (uint32_t)img.height();
        if ((size_t)frames < out_cap)
            for (uint32_t y = 0; y < h; y++) memcpy(out + ((size_t)frames * h + y) * w, img.constScanLine((int)y), (size_t)w * 4);
        frames++;
    });
    ren.setLivePlay(false);
    ren.startPCM1600DBFrame();
    ref_p16_hooks hk;
    hk.on_block = [&](PCM16X0DataBlock &b) { ren.renderNewBlock(b); };
    hk.on_frame = [&](uint32_t frame_no) { ren.prepareNewFrame(frame_no); ren.displayIsReady(); };
    const size_t nfr = n_recs / 100 + 16;
    std::vector<sdv_sample_pair> pairs(n_recs * 3 + 4096);
    std::vector<sdv_frame_asm_pcm16x0> fr(nfr);
    size_t nf = 0;
    const long n = ref_pcm16x0_stitch_run_hooks(recs, n_recs, st, pairs.data(), pairs.size(), fr.data(), fr.size(), &nf, &hk);
    if (n < 0) return -1;
    if (width) *width = w;
    if (height) *height = h;
    return frames;
}
