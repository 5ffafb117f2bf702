/*
 * decode_tape_sharded.cpp - one STC-007 tape decoded by several GPUs of a node from plain C++ host code: one process per GPU, contiguous frame
 * ranges per rank, the luma never leaves its GPU, and the only thing that crosses between the ranks is what the reference's two workers carry from
 * frame to frame, all-gathered with RCCL (ncclAllGather over xGMI):
 *     VideoToDigital        sdv_v2d_state, 120 bytes                              (sdv_get_chain_state / sdv_set_chain_state)
 *     STC007DataStitcher    previous descriptor, statistics rings and the 112 assembled lines of conv_queue - the field seam -, 3.8 KB
 *                           (sdv_get_stitch_state / sdv_set_stitch_state)
 * The gathered states are used to VERIFY a guess, the same speculation the engine runs inside a batch, one level up (DESIGN.md section 7;
 * the Python harness of the same loop: sdvpcmdecoder_amd/sharded.py):
 *   0. rank 0 decodes the first frames of its range and publishes its state (the binarizer's sticky levels), the others start their warm-up from it;
 *   1. rank r > 0 decodes a short warm-up just before its range and keeps the state it ends in - its prediction of what rank
 *      r - 1 will hand over;
 *   2. every rank decodes its range from that state and runs the stitcher over the records straight away (warmed up the same way), then ONE all-gather
 *      carries what both workers assumed and what they ended in, so that every rank knows every rank's verdict;
 *   3. a rank whose prediction differs from what its predecessor really ended in decodes (or stitches) its range again from the true state; repeated
 *      until every rank started from exactly its predecessor's final state.  The result is the sequential decode of the whole tape, whatever the
 *      guesses were.  A tape that plays takes two all-gathers (step 0 and step 2).
 *
 *   RANK=r WORLD_SIZE=n [LOCAL_RANK=d] decode_tape_sharded <luma.raw> <width> <height> <n_frames> <out prefix> [rccl | file:<dir>] [warm-up frames [stitcher warm-up turns]]
 *       writes <out prefix>.rank<r>.pairs / .frames: the PCMSamplePair records and FrameAsmSTC007 descriptors of this rank's frames; concatenated in
 *       rank order they are what `decode_tape stc007` writes for the same file.
 *   Collective back ends: `rccl` (default; rank 0 publishes the ncclUniqueId in <out prefix>.ncclid - per run, see RcclComm; give all ranks of a run the
 *   same SDV_RUN_ID unless a launcher already sets TORCHELASTIC_RUN_ID / MASTER_PORT) and `file:<dir>` (the all-gather through files
 *   of a shared directory: for debugging and for the CPU test of this loop, which links the test-only emulator build of the engine and has no RCCL).
 *
 * Build: build.py build_example_sharded (g++, -lsdvpcm_hip -lamdhip64 -lrccl); the CPU test build: -DSDV_EXAMPLE_HOST_MEMORY against
 * tests/emu/libsdvpcm_emu.so (the same C-ABI on host pointers).
 */
#ifndef SDV_EXAMPLE_HOST_MEMORY
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#endif
#include <chrono>
#include <ctime>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "sdvpcm.h"

/* ---- buffers: device memory (product) or host memory (the emulator build of the CPU test) ------------------------------------------------ */
#ifndef SDV_EXAMPLE_HOST_MEMORY
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
static void *dev_alloc(size_t n) { void *p = NULL; HIP_OK(hipMalloc(&p, n ? n : 1)); return p; }
static void dev_free(void *p) { if (p) (void)hipFree(p); }
static void h2d(void *d, const void *h, size_t n) { if (n) HIP_OK(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); }
static void d2h(void *h, const void *d, size_t n) { if (n) HIP_OK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); }
static void d2d(void *d, const void *s, size_t n) { if (n) HIP_OK(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice)); }
static void dev_sync() { HIP_OK(hipDeviceSynchronize()); }
#else
static void *dev_alloc(size_t n) { return malloc(n ? n : 1); }
static void dev_free(void *p) { free(p); }
static void h2d(void *d, const void *h, size_t n) { memcpy(d, h, n); }
static void d2h(void *h, const void *d, size_t n) { memcpy(h, d, n); }
static void d2d(void *d, const void *s, size_t n) { memcpy(d, s, n); }
static void dev_sync() {}
#endif
#define SDV_OKAY(x) do { int r_ = (x); if (r_ != SDV_OK) { fprintf(stderr, "rank %d: %s = %d: %s\n", g_rank, #x, r_, sdv_last_error(eng)); exit(3); } } while (0)
static int g_rank = 0;

/* ---- the collective: every rank contributes `bytes`, every rank receives all contributions in rank order ----------------------------------- */
struct Comm {
    int rank, world;
    virtual ~Comm() {}
    virtual void all_gather(const void *send, void *recv, size_t bytes) = 0;
};
#ifndef SDV_EXAMPLE_HOST_MEMORY
#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); exit(4); } } while (0)
/* The rendezvous is per RUN: the file that carries rank 0's ncclUniqueId is named after the run (run_tag(): SDV_RUN_ID, else what the launchers set for
 * all ranks of a job - TORCHELASTIC_RUN_ID, MASTER_ADDR:MASTER_PORT), starts with that tag, is removed by rank 0 before it publishes a new id and again
 * once every rank has joined (ncclCommInitRank returns when all have), and a rank that finds a file of another run - or, without any tag, one written
 * before it was started itself - keeps waiting.  An id file left by a job that died can therefore not be picked up by the next one. */
static std::string run_tag()
{
    if (const char *v = getenv("SDV_RUN_ID")) return v;
    std::string t;
    if (const char *v = getenv("TORCHELASTIC_RUN_ID")) t = v;
    if (const char *a = getenv("MASTER_ADDR")) { t += "@"; t += a; if (const char *p = getenv("MASTER_PORT")) { t += ":"; t += p; } }
    return t;
}
struct IdFile { char magic[8]; char tag[120]; ncclUniqueId id; };
struct RcclComm : Comm {
    ncclComm_t comm; hipStream_t stream; uint8_t *d_send, *d_recv; size_t cap;
    RcclComm(int r, int w, const std::string &id_file) : comm(NULL), stream(NULL), d_send(NULL), d_recv(NULL), cap(0)
    {
        rank = r; world = w;
        const std::string tag = run_tag();
        const time_t started = time(NULL);
        IdFile rec;
        memset(&rec, 0, sizeof(rec));
        if (rank == 0) {
            (void)remove(id_file.c_str());                    /* whatever an earlier run left */
            memcpy(rec.magic, "SDVNCCL1", 8);
            snprintf(rec.tag, sizeof(rec.tag), "%s", tag.c_str());
            NCCL_OK(ncclGetUniqueId(&rec.id));
            const std::string tmp = id_file + ".tmp" + std::to_string((long)getpid());
            FILE *f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&rec, sizeof(rec), 1, f) != 1) { fprintf(stderr, "cannot write %s\n", tmp.c_str()); exit(4); }
            fclose(f);
            rename(tmp.c_str(), id_file.c_str());
        } else {
            for (int tries = 0;; tries++) {
                FILE *f = fopen(id_file.c_str(), "rb");
                if (f) {
                    bool ok = fread(&rec, sizeof(rec), 1, f) == 1 && memcmp(rec.magic, "SDVNCCL1", 8) == 0 && tag == std::string(rec.tag, strnlen(rec.tag, sizeof(rec.tag)));
                    struct stat sb;
                    /* ... and only a file written about when this rank was started: since then when there is no run tag to tell the runs apart; with a tag, not
                     * more than half a minute before (the ranks of a run are started together, rank 0 may be ahead by a little) - an id left under the same
                     * tag by a run that died (a constant SDV_RUN_ID, the same MASTER_ADDR:MASTER_PORT again) is older than that, and rank 0 removes it first
                     * thing anyway */
                    if (ok) ok = fstat(fileno(f), &sb) == 0 && sb.st_mtime + (tag.empty() ? 1 : 30) >= started;
                    fclose(f);
                    if (ok) break;
                }
                if (tries > 6000) { fprintf(stderr, "rank %d: no ncclUniqueId of this run in %s (set SDV_RUN_ID to the same value for all ranks of a run)\n", rank, id_file.c_str()); exit(4); }
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
        }
        HIP_OK(hipStreamCreate(&stream));
        NCCL_OK(ncclCommInitRank(&comm, world, rec.id, rank));
        if (rank == 0) (void)remove(id_file.c_str());         /* everybody has joined: nobody needs it any more */
    }
    ~RcclComm() { if (comm) ncclCommDestroy(comm); dev_free(d_send); dev_free(d_recv); if (stream) (void)hipStreamDestroy(stream); }
    void all_gather(const void *send, void *recv, size_t bytes) override
    {
        if (bytes > cap) { dev_free(d_send); dev_free(d_recv); d_send = (uint8_t *)dev_alloc(bytes); d_recv = (uint8_t *)dev_alloc(bytes * (size_t)world); cap = bytes; }
        HIP_OK(hipMemcpyAsync(d_send, send, bytes, hipMemcpyHostToDevice, stream));
        NCCL_OK(ncclAllGather(d_send, d_recv, bytes, ncclChar, comm, stream));       /* latency-bound: a few KB per rank over xGMI */
        HIP_OK(hipMemcpyAsync(recv, d_recv, bytes * (size_t)world, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }
};
#endif
struct FileComm : Comm {
    std::string dir, tag; unsigned seq;
    /* files of one run only: named after the run (SDV_RUN_ID / the launcher's variables, as above), a rank's file of gather k - 1 removed once gather k
     * is through (every rank wrote its file of gather k after it had read all of k - 1) */
    FileComm(int r, int w, const std::string &d) : dir(d), seq(0)
    {
        rank = r; world = w;
        std::string t;
        if (const char *v = getenv("SDV_RUN_ID")) t = v;
        else { if (const char *v = getenv("TORCHELASTIC_RUN_ID")) t = v; if (const char *p = getenv("MASTER_PORT")) { t += "p"; t += p; } }
        for (char &c : t) if (!((c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '-' || c == '_')) c = '_';
        tag = t.empty() ? std::string("g") : "g" + t + "_";
    }
    std::string name(unsigned k, int r) const { return dir + "/" + tag + std::to_string(k) + ".r" + std::to_string(r); }
    /* (the files of the very last gather stay: a rank cannot know that the others have read them, and waiting for them at exit would tie a fast rank to
     * the slowest one's repair; they carry the run's name, another run never reads them) */
    void all_gather(const void *send, void *recv, size_t bytes) override
    {
        const std::string mine = name(seq, rank);
        FILE *f = fopen((mine + ".tmp").c_str(), "wb");
        if (!f || fwrite(send, 1, bytes, f) != bytes) { fprintf(stderr, "rank %d: cannot write %s\n", rank, mine.c_str()); exit(4); }
        fclose(f);
        rename((mine + ".tmp").c_str(), mine.c_str());
        for (int r = 0; r < world; r++) {
            const std::string theirs = name(seq, r);
            for (int tries = 0;; tries++) {
                FILE *g = fopen(theirs.c_str(), "rb");
                if (g) { const bool ok = fread((uint8_t *)recv + (size_t)r * bytes, 1, bytes, g) == bytes; fclose(g); if (ok) break; }
                if (tries > 60000) { fprintf(stderr, "rank %d: rank %d never wrote %s\n", rank, r, theirs.c_str()); exit(4); }
                std::this_thread::sleep_for(std::chrono::milliseconds(2));
            }
        }
        if (seq >= 1) (void)remove(name(seq - 1, rank).c_str());
        seq++;
    }
};

static bool write_file(const std::string &path, const void *p, size_t n)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(p, 1, n, f) == n;
    fclose(f);
    return ok;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: see the header of examples/decode_tape_sharded.cpp\n"); return 1; }
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    const int local = getenv("LOCAL_RANK") ? atoi(getenv("LOCAL_RANK")) : rank;
    g_rank = rank;
    const char *luma_path = argv[1];
    const int width = atoi(argv[2]), height = atoi(argv[3]), n_frames = atoi(argv[4]);
    const std::string prefix = argv[5], comm_arg = argc > 6 ? argv[6] : "rccl";
    const int warmup = argc > 7 ? atoi(argv[7]) : 20, stitch_warmup = argc > 8 ? atoi(argv[8]) : 4;
    if (rank < 0 || rank >= world || n_frames < world || width <= 0 || height <= 0) { fprintf(stderr, "bad arguments\n"); return 1; }
#ifndef SDV_EXAMPLE_HOST_MEMORY
    HIP_OK(hipSetDevice(local));
#endif
    sdv_engine *eng = sdv_engine_create(local);
    if (!eng) { fprintf(stderr, "sdv_engine_create: %s\n", sdv_last_error(NULL)); return 2; }
    Comm *comm = NULL;
    if (comm_arg.rfind("file:", 0) == 0) comm = new FileComm(rank, world, comm_arg.substr(5));
#ifndef SDV_EXAMPLE_HOST_MEMORY
    else comm = new RcclComm(rank, world, prefix + ".ncclid");
#else
    else { fprintf(stderr, "this build has the file back end only\n"); return 1; }
#endif

    /* ---- this rank's part of the tape: `lead` warm-up frames, its own range [lo, hi), one successor frame for the last stitcher turn ---------- */
    const int lo = (int)((long long)n_frames * rank / world), hi = (int)((long long)n_frames * (rank + 1) / world);
    const int lead = warmup < lo ? warmup : lo, look = hi < n_frames ? 1 : 0, n_own = hi - lo;
    const bool last = rank == world - 1;
    const size_t frame_bytes = (size_t)width * (size_t)height, n_in = (size_t)(lead + n_own + look);
    std::vector<uint8_t> host(n_in * frame_bytes);
    {
        FILE *f = fopen(luma_path, "rb");
        if (!f || fseek(f, (long)((size_t)(lo - lead) * frame_bytes), SEEK_SET) != 0 || fread(host.data(), 1, host.size(), f) != host.size()) { fprintf(stderr, "rank %d: cannot read %s\n", rank, luma_path); return 1; }
        fclose(f);
    }
    uint8_t *d_luma = (uint8_t *)dev_alloc(host.size());
    h2d(d_luma, host.data(), host.size());
    const size_t rpf = sdv_records_per_frame(height);
    const unsigned own_flags = (rank == 0 ? SDV_FLAG_NEW_FILE : 0u) | (last ? SDV_FLAG_END_FILE : 0u);
    const size_t n_warm = (size_t)lead * rpf, n_own_recs = sdv_binarize_records(height, n_own, own_flags), n_extra = (size_t)look * rpf;
    sdv_line_rec *d_warm = (sdv_line_rec *)dev_alloc(n_warm * sizeof(sdv_line_rec));
    sdv_line_rec *d_whole = (sdv_line_rec *)dev_alloc((n_own_recs + n_extra) * sizeof(sdv_line_rec));       /* own range, then the successor frame */
    sdv_frame_stats *d_stats = (sdv_frame_stats *)dev_alloc((n_in + 2) * sizeof(sdv_frame_stats));
    SDV_OKAY(sdv_set_mode(eng, SDV_MODE_NORMAL));

    /* ---- binarize stage (the VideoToDigital worker) ------------------------------------------------------------------------------------------- */
    unsigned gathers = 0, binarize_redo = 0, stitch_redo = 0;
    SDV_OKAY(sdv_reset_stream(eng));
    sdv_v2d_state predicted = {}, fin = {};          /* (both travel in the all-gather blob: a rank without a warm-up sends zeros, as sharded.py does) */
    /* The binarizer's levels are sticky - a line that reads from the levels it inherits does not measure them again - so on a tape that plays they are
     * what the first lines of the TAPE measured, which no warm-up further down can find out.  Rank 0 decodes the first frames of its range first and
     * publishes the state it has then; the other ranks start their warm-up from it (a warm-up of 20 frames refills the histories either way). */
    int k0 = 0; size_t head_recs = 0;
    if (world > 1) {
        sdv_v2d_state early;
        memset(&early, 0, sizeof(early));
        if (rank == 0) {
            k0 = warmup > 0 ? warmup : 1; if (k0 > n_own) k0 = n_own;
            const unsigned fl = SDV_FLAG_NEW_FILE | ((last && k0 == n_own) ? SDV_FLAG_END_FILE : 0u);
            head_recs = sdv_binarize_records(height, k0, fl);
            SDV_OKAY(sdv_binarize_frames(eng, d_luma, (size_t)width, frame_bytes, width, height, k0, 1u, fl, d_whole, head_recs, d_stats, n_in + 2, NULL));
            SDV_OKAY(sdv_get_chain_state(eng, &early));
        }
        std::vector<sdv_v2d_state> earlies((size_t)world);
        comm->all_gather(&early, earlies.data(), sizeof(early));
        gathers++;
        if (rank > 0 && lo - lead > 0) SDV_OKAY(sdv_set_chain_state(eng, &earlies[0]));     /* (a warm-up that begins with the tape is the tape's own start) */
    }
    if (lead) {
        SDV_OKAY(sdv_binarize_frames(eng, d_luma, (size_t)width, frame_bytes, width, height, lead, (uint32_t)(1 + lo - lead), 0, d_warm, n_warm, d_stats, n_in + 2, NULL));
        SDV_OKAY(sdv_get_chain_state(eng, &predicted));
    }
    /* ---- the range: both workers back to back inside the engine (sdv_decode_frames), and one all-gather for both -------------------------------------
     * The stitcher runs straight behind the binarizer, before anybody knows whether the range was decoded from the right state: on a tape that plays it
     * was, and then one all-gather carries what both workers assumed and what they ended with - every rank works out every rank's verdict from it, no
     * second gather to agree on going on.  A rank that assumed wrong runs its range again from the true states. */
    const size_t n_whole = n_own_recs + n_extra;
    const size_t pairs_cap = n_whole * 4 + 8192, frames_cap = (size_t)n_own + 16, state_n = sdv_stitch_state_size();
    sdv_sample_pair *d_pairs = (sdv_sample_pair *)dev_alloc(pairs_cap * sizeof(sdv_sample_pair));
    sdv_frame_asm *d_frames = (sdv_frame_asm *)dev_alloc(frames_cap * sizeof(sdv_frame_asm));
    sdv_stitch_settings st; sdv_default_stitch_settings(&st);
    SDV_OKAY(sdv_set_stitch_settings(eng, &st));
    const size_t nb = sizeof(sdv_v2d_state), blob_n = 2 * nb + 2 * state_n;
    std::vector<uint8_t> s_pred(state_n, 0), s_final(state_n, 0), blob(blob_n), blobs(blob_n * (size_t)world);
    size_t n_pairs = 0, n_fr = 0;
    const int s_lead = stitch_warmup < lead ? stitch_warmup : lead;
    /* s_from: the stitcher state to start from instead of the warm-up's guess (NULL: a fresh stitcher); the binarizer's state is what the engine holds */
    auto run_range = [&](const uint8_t *s_from) {
        int start = 0;
        n_pairs = 0; n_fr = 0;
        auto fused = [&](int first, int count, unsigned fl) {       /* frames [first, first + count) of this rank's input through both workers, appended */
            size_t np = 0, nf = 0, npur = 0; uint64_t masked = 0;
            SDV_OKAY(sdv_decode_frames(eng, SDV_PCM_STC007, d_luma + (size_t)first * frame_bytes, (size_t)width, frame_bytes, width, height, count,
                                       (uint32_t)(1 + lo - lead + first), fl, d_pairs + n_pairs, pairs_cap - n_pairs, &np, d_frames + n_fr, frames_cap - n_fr, &nf,
                                       d_stats, n_in + 2, 0, 0, NULL, 0, &npur, &masked, NULL));
            n_pairs += np; n_fr += nf;
        };
        if (s_from) {
            SDV_OKAY(sdv_set_stitch_state(eng, s_from, state_n));       /* drops the waiting frame */
            memcpy(s_pred.data(), s_from, state_n);
        } else {
            SDV_OKAY(sdv_reset_stitcher(eng));
            std::fill(s_pred.begin(), s_pred.end(), 0);
            if (k0 > 0) {       /* rank 0 of several: the frames it decoded first go to the stitcher as records */
                SDV_OKAY(sdv_stitch_frames(eng, d_whole, head_recs, d_pairs, pairs_cap, &n_pairs, d_frames, frames_cap, &n_fr, NULL));
                start = k0;
            } else if (s_lead) {
                /* warm-up turns lo - s_lead .. lo - 1 (output discarded); frame lo then waits inside the engine for its successor */
                const unsigned fl = (last && n_own == 1) ? SDV_FLAG_END_FILE : 0u;
                const size_t n_first = sdv_binarize_records(height, 1, fl), n_cat = (size_t)s_lead * rpf + n_first;
                sdv_line_rec *d_cat = (sdv_line_rec *)dev_alloc(n_cat * sizeof(sdv_line_rec));
                d2d(d_cat, d_warm + (size_t)(lead - s_lead) * rpf, (size_t)s_lead * rpf * sizeof(sdv_line_rec));
                SDV_OKAY(sdv_binarize_frames(eng, d_luma + (size_t)lead * frame_bytes, (size_t)width, frame_bytes, width, height, 1, (uint32_t)(1 + lo), fl,
                                             d_cat + (size_t)s_lead * rpf, n_first, d_stats, n_in + 2, NULL));
                size_t np = 0, nf = 0;
                SDV_OKAY(sdv_stitch_frames(eng, d_cat, n_cat, d_pairs, pairs_cap, &np, d_frames, frames_cap, &nf, NULL));
                SDV_OKAY(sdv_saturate_stitch_stats(eng));
                SDV_OKAY(sdv_get_stitch_state(eng, s_pred.data(), state_n));
                dev_free(d_cat);
                start = 1;
            }
        }
        if (start < n_own) fused(lead + start, n_own - start, ((rank == 0 && start == 0) ? SDV_FLAG_NEW_FILE : 0u) | (last ? SDV_FLAG_END_FILE : 0u));
        SDV_OKAY(sdv_get_chain_state(eng, &fin));
        if (look) fused(lead + n_own, 1, 0u);       /* the successor of this range's last stitcher turn (rank r + 1 decodes it again, as its first frame) */
        SDV_OKAY(sdv_get_stitch_state(eng, s_final.data(), state_n));
    };
    run_range(NULL);
    for (;;) {
        memcpy(blob.data(), &predicted, nb); memcpy(blob.data() + nb, &fin, nb);
        memcpy(blob.data() + 2 * nb, s_pred.data(), state_n); memcpy(blob.data() + 2 * nb + state_n, s_final.data(), state_n);
        comm->all_gather(blob.data(), blobs.data(), blob_n);
        gathers++;
        auto part = [&](int r, size_t ofs) { return blobs.data() + (size_t)r * blob_n + ofs; };
        bool all_bin = true, all_st = true, my_bin = true, my_st = true;
        for (int r = 1; r < world; r++) {
            const bool b_ok = memcmp(part(r, 0), part(r - 1, nb), nb) == 0, t_ok = memcmp(part(r, 2 * nb), part(r - 1, 2 * nb + state_n), state_n) == 0;
            all_bin = all_bin && b_ok; all_st = all_st && t_ok;
            if (r == rank) { my_bin = b_ok; my_st = t_ok; }
        }
        if (all_bin && all_st) break;
        if (!my_bin) {
            binarize_redo++;
            memcpy(&predicted, part(rank - 1, nb), nb);
            SDV_OKAY(sdv_set_chain_state(eng, &predicted));
            run_range(NULL);
        } else if (all_bin && !my_st) {         /* (while a binarizer still decodes again, the stitcher states behind it are not final) */
            stitch_redo++;
            SDV_OKAY(sdv_set_chain_state(eng, &predicted));       /* the records stayed inside the engine: the range runs again, from both true states */
            std::vector<uint8_t> from(part(rank - 1, 2 * nb + state_n), part(rank - 1, 2 * nb + state_n) + state_n);
            run_range(from.data());
        }
    }
    dev_sync();

    std::vector<sdv_sample_pair> h_pairs(n_pairs); std::vector<sdv_frame_asm> h_frames(n_fr);
    d2h(h_pairs.data(), d_pairs, n_pairs * sizeof(sdv_sample_pair));
    d2h(h_frames.data(), d_frames, n_fr * sizeof(sdv_frame_asm));
    const std::string base = prefix + ".rank" + std::to_string(rank);
    if (!write_file(base + ".pairs", h_pairs.data(), n_pairs * sizeof(sdv_sample_pair)) || !write_file(base + ".frames", h_frames.data(), n_fr * sizeof(sdv_frame_asm))) { fprintf(stderr, "cannot write %s.*\n", base.c_str()); return 4; }
    printf("rank %d of %d: frames [%d, %d) (+ %d warm-up, + %d successor) -> %zu sample pairs, %zu frame descriptors; %u all-gathers, ranges decoded again: binarize %u, stitch %u\n",
           rank, world, lo, hi, lead, look, n_pairs, n_fr, gathers, binarize_redo, stitch_redo);
    delete comm;
    dev_free(d_luma); dev_free(d_warm); dev_free(d_whole); dev_free(d_stats); dev_free(d_pairs); dev_free(d_frames);
    sdv_engine_destroy(eng);
    return 0;
}
