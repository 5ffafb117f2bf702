/*
 * ref_audio_driver.cpp - C-ABI shim around the REAL AudioProcessor / SamplesToWAV of the reference (compiled from
 * /root/reference by oracle/Makefile.ref into oracle/_ref/libsdvref.so).
 *
 * TEST INFRASTRUCTURE ONLY.  Our own code: it constructs the reference's class, loads its input queue the way
 * sdv_audio_process defines the feed schedule (a burst is in the queue before the worker's next turn; the queue runs dry at
 * the end of a burst), runs the reference's own processAudio() loop and collects what it emits (outSamples, newSource,
 * guiAddMask) and, on request, lets it write its WAV files.  No reference source is copied.
 */
#include <cstring>
#include <cstdint>
#include <vector>
#include <deque>
#include <string>
#include <thread>
#include <chrono>
#include <atomic>
#include <QMutex>
#include <QObject>
#include <QCoreApplication>
#include <QEventLoop>
#include <QMetaObject>
#include "audioprocessor.h"
#include "../include/sdvpcm.h"

static void pod_to_pair(const sdv_sample_pair &r, PCMSamplePair &p, const std::string &dir, size_t k)
{
    p.clear();
    if (r.service_type == SDV_PAIR_SRV_NEW_FILE) { p.setServNewFile(dir + "/src" + std::to_string(k) + ".avi"); return; }
    if (r.service_type == SDV_PAIR_SRV_END_FILE) { p.setServEndFile(); return; }
    for (int c = 0; c < 2; c++) {
        p.samples[c].audio_word = r.audio_word[c];
        p.samples[c].data_block_ok = (r.sample_flags[c] & SDV_SF_BLOCK_OK) != 0;
        p.samples[c].word_valid = (r.sample_flags[c] & SDV_SF_WORD_VALID) != 0;
        p.samples[c].word_fixed = (r.sample_flags[c] & SDV_SF_WORD_FIXED) != 0;
        p.samples[c].word_masked = (r.sample_flags[c] & SDV_SF_WORD_MASKED) != 0;
    }
    p.sample_rate = r.sample_rate; p.emphasis = r.emphasis != 0;
}

/* bursts[i] = end of burst i in `pairs` (ascending, the last one = n).  wav_dir: NULL, or a directory the reference writes
 * src<k>_v<APP_VERSION>.wav into for the k-th NEW_FILE tag.  Returns the number of pairs put out (before stop() unless `stop`),
 * -1 if out_cap was too small. */
extern "C" long ref_audio_run(const sdv_sample_pair *pairs, size_t n, const uint64_t *bursts, size_t n_bursts, int mask_mode, int stop,
                              sdv_sample_pair *out, uint64_t *out_index, size_t out_cap, uint64_t *purges, size_t purges_cap, size_t *n_purges,
                              uint64_t *n_masked, const char *wav_dir)
{
    std::deque<PCMSamplePair> in_q;
    QMutex in_mtx;
    std::vector<PCMSamplePair> got;
    std::vector<uint64_t> pur;
    std::atomic<uint64_t> masked(0);
    std::atomic<AudioProcessor *> apptr(NULL);
    const std::string dir = wav_dir ? wav_dir : "/nonexistent-sdv";
    /* In the application stop() reaches the worker as a queued slot call, i.e. inside the processEvents() of its loop, and that is
     * what makes the loop purge before it ends (audioprocessor.cpp:1653-1660).  For that the worker's thread needs Qt's event
     * dispatcher, which Qt only creates when an application object exists. */
    static int q_argc = 1; static char q_arg0[] = "sdvref"; static char *q_argv[] = { q_arg0, NULL };
    if (!QCoreApplication::instance()) new QCoreApplication(q_argc, q_argv);
    /* the object lives in the thread that runs its loop, so that its own signal/slot connections are direct calls */
    std::thread th([&]() {
        QEventLoop dispatcher_for_this_thread;
        AudioProcessor *ap = new AudioProcessor();
        ap->setInputPointers(&in_q, &in_mtx);
        ap->setMasking((uint8_t)mask_mode);
        ap->setOutputToFile(wav_dir != NULL);
        ap->setOutputToLive(false);
        QObject::connect(ap, &AudioProcessor::outSamples, [&](PCMSamplePair p) { got.push_back(p); });
        QObject::connect(ap, &AudioProcessor::newSource, [&]() { pur.push_back((uint64_t)got.size()); });
        QObject::connect(ap, &AudioProcessor::guiAddMask, [&](uint16_t c) { masked += c; });
        apptr = ap;
        ap->processAudio();
        delete ap;
    });
    while (apptr.load() == NULL) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    size_t fed = 0, nfile = 0;
    PCMSamplePair p;
    for (size_t b = 0; b < n_bursts; b++) {
        in_mtx.lock();                                  /* the whole burst under the lock: the worker sees all of it or none */
        for (; fed < bursts[b] && fed < n; fed++) {
            pod_to_pair(pairs[fed], p, dir, nfile);
            if (pairs[fed].service_type == SDV_PAIR_SRV_NEW_FILE) nfile++;
            in_q.push_back(p);
        }
        in_mtx.unlock();
        /* wait for the queue to run dry (or for a worker that has stopped taking input), then for the turn that emptied it */
        size_t last = (size_t)-1; int same = 0;
        while (true) {
            in_mtx.lock(); const size_t qs = in_q.size(); in_mtx.unlock();
            if (qs == 0) break;
            if (qs == last) { if (++same > 40) break; } else { same = 0; last = qs; }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
        }
        /* the worker sleeps 50 ms when it finds the queue empty and a turn takes well under a millisecond: a quarter of a second of margin for a loaded machine
         * (only the scenarios with several bursts pay it more than once) */
        std::this_thread::sleep_for(std::chrono::milliseconds(b + 1 < n_bursts ? 250 : 130));
    }
    const size_t before_stop = got.size(), pur_before = pur.size();
    const uint64_t masked_before = masked.load();
    QMetaObject::invokeMethod(apptr.load(), "stop", Qt::QueuedConnection);
    th.join();
    const size_t n_got = stop ? got.size() : before_stop, n_pur = stop ? pur.size() : pur_before;
    for (size_t i = 0; i < n_got && i < out_cap; i++) {
        sdv_sample_pair *o = &out[i];
        PCMSamplePair &q = got[i];
        memset(o, 0, sizeof(*o));
        for (int c = 0; c < 2; c++) {
            o->audio_word[c] = q.samples[c].audio_word;
            o->sample_flags[c] = (uint8_t)((q.samples[c].data_block_ok ? SDV_SF_BLOCK_OK : 0) | (q.samples[c].word_valid ? SDV_SF_WORD_VALID : 0) |
                                           (q.samples[c].word_fixed ? SDV_SF_WORD_FIXED : 0) | (q.samples[c].word_masked ? SDV_SF_WORD_MASKED : 0));
        }
        o->sample_rate = q.sample_rate; o->emphasis = q.emphasis; o->service_type = q.service_type;
        if (out_index) out_index[i] = q.samples[0].index;
    }
    for (size_t i = 0; i < n_pur && i < purges_cap; i++) purges[i] = pur[i];
    if (n_purges) *n_purges = n_pur;
    if (n_masked) *n_masked = stop ? masked.load() : masked_before;
    return n_got > out_cap ? -1 : (long)n_got;
}
