/*
 * decode_tape.cpp - the drop-in boundary used the way the Qt application would use it: plain C++ host code, the HIP runtime for
 * the buffers, the C-ABI of include/sdvpcm.h for the work.  No Python, no torch.
 *
 *   decode_tape stc007 <luma.raw> <width> <height> <n_frames> <pairs.out> <frames.out>
 *       8-bit luma frames (what VideoInFFMPEG hands to VideoToDigital, vin_ffmpeg.cpp:281-350) -> sdv_binarize_frames
 *       (the VideoToDigital worker's body) -> sdv_stitch_frames (the STC007DataStitcher worker's body) -> PCMSamplePair records
 *   decode_tape pcm1 <lines.raw> <pairs.out> <frames.out>
 *       sdv_pcm1_line_rec records (the PCM1DataStitcher worker's input deque) -> sdv_pcm1_stitch_frames
 *   decode_tape pcm16x0 <luma.raw> <width> <height> <n_frames> <si|ei> <pairs.out> <frames.out>
 *       8-bit luma frames of a PCM-1600/1610/1630 tape -> sdv_pcm16x0_binarize_frames (VideoToDigital with TYPE_PCM16X0) ->
 *       sdv_pcm16x0_stitch_frames (the PCM16X0DataStitcher worker's body); the sub-line records never leave the device
 *
 *   decode_tape wav <luma.raw> <width> <height> <n_frames> <out.wav> [<mask mode 0..6>]
 *       the whole chain of the application for an STC-007 file: sdv_binarize_frames -> sdv_stitch_frames -> sdv_audio_process (the
 *       AudioProcessor worker's loop, linear interpolation of dropouts by default) -> sdv_wav_pack + sdv_wav_header: the file SamplesToWAV
 *       writes, byte for byte; nothing but the luma goes to the device and nothing but the 16-bit PCM comes back
 *
 * Build (host code only, any C++ compiler): g++ -std=c++17 -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/decode_tape.cpp
 *        -Lsdvpcmdecoder_amd -lsdvpcm_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,'$ORIGIN/../sdvpcmdecoder_amd' (build.py: build_example).
 */
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "sdvpcm.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SDV_OKAY(x) do { int r_ = (x); if (r_ != SDV_OK) { fprintf(stderr, "%s = %d: %s\n", #x, r_, sdv_last_error(eng)); return 3; } } while (0)

static bool read_file(const char *path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    out.resize((size_t)n);
    bool ok = fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}
static bool write_file(const char *path, const void *p, size_t n)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    bool ok = fwrite(p, 1, n, f) == n;
    fclose(f);
    return ok;
}
template <class T> static int download(const T *dev, size_t n, const char *path)
{
    std::vector<T> host(n);
    if (n) HIP_OK(hipMemcpy(host.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
    return write_file(path, host.data(), n * sizeof(T)) ? 0 : 4;
}

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: see the header of examples/decode_tape.cpp\n"); return 1; }
    const std::string mode = argv[1];
    sdv_engine *eng = sdv_engine_create(0);
    if (!eng) { fprintf(stderr, "sdv_engine_create: %s\n", sdv_last_error(NULL)); return 2; }
    std::vector<uint8_t> in;
    sdv_sample_pair *d_pairs = NULL;
    size_t n_pairs = 0, n_frames = 0;
    int rc = 0;
    if (mode == "stc007" && argc == 8) {
        const int width = atoi(argv[3]), height = atoi(argv[4]), n = atoi(argv[5]);
        if (!read_file(argv[2], in) || in.size() != (size_t)width * height * n) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
        uint8_t *d_luma = NULL; sdv_line_rec *d_lines = NULL; sdv_frame_stats *d_stats = NULL; sdv_frame_asm *d_frames = NULL;
        /* one file from its first to its last frame: NEW_FILE tag ahead, filler frame + END_FILE tag behind */
        const size_t n_lines = sdv_binarize_records(height, n, SDV_FLAG_NEW_FILE | SDV_FLAG_END_FILE);
        const size_t pairs_cap = n_lines * 4 + 8192, frames_cap = (size_t)n + 16;
        HIP_OK(hipMalloc((void **)&d_luma, in.size()));
        HIP_OK(hipMalloc((void **)&d_lines, n_lines * sizeof(sdv_line_rec)));
        HIP_OK(hipMalloc((void **)&d_stats, ((size_t)n + 1) * sizeof(sdv_frame_stats)));
        HIP_OK(hipMalloc((void **)&d_pairs, pairs_cap * sizeof(sdv_sample_pair)));
        HIP_OK(hipMalloc((void **)&d_frames, frames_cap * sizeof(sdv_frame_asm)));
        HIP_OK(hipMemcpy(d_luma, in.data(), in.size(), hipMemcpyHostToDevice));
        SDV_OKAY(sdv_set_mode(eng, SDV_MODE_NORMAL));
        SDV_OKAY(sdv_binarize_frames(eng, d_luma, (size_t)width, (size_t)width * height, width, height, n, 1,
                                     SDV_FLAG_NEW_FILE | SDV_FLAG_END_FILE, d_lines, n_lines, d_stats, (size_t)n + 1, NULL));
        sdv_stitch_settings st; sdv_default_stitch_settings(&st);
        SDV_OKAY(sdv_set_stitch_settings(eng, &st));
        SDV_OKAY(sdv_stitch_frames(eng, d_lines, n_lines, d_pairs, pairs_cap, &n_pairs, d_frames, frames_cap, &n_frames, NULL));
        HIP_OK(hipDeviceSynchronize());
        rc = download(d_pairs, n_pairs, argv[6]); if (!rc) rc = download(d_frames, n_frames, argv[7]);
        sdv_run_info info; sdv_get_run_info(eng, &info);
        printf("stc007: %d frames -> %zu line records -> %zu sample pairs, %zu frame descriptors (binarize rounds %u)\n", n, n_lines, n_pairs, n_frames, info.rounds);
        (void)hipFree(d_luma); (void)hipFree(d_lines); (void)hipFree(d_stats); (void)hipFree(d_frames);
    } else if (mode == "wav" && (argc == 7 || argc == 8)) {
        const int width = atoi(argv[3]), height = atoi(argv[4]), n = atoi(argv[5]);
        const int mask_mode = argc == 8 ? atoi(argv[7]) : SDV_DROP_INTER_LIN_WORD;
        if (!read_file(argv[2], in) || in.size() != (size_t)width * height * n) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
        uint8_t *d_luma = NULL; sdv_line_rec *d_lines = NULL; sdv_frame_stats *d_stats = NULL; sdv_frame_asm *d_frames = NULL;
        sdv_sample_pair *d_audio = NULL; sdv_audio_purge *d_purges = NULL; int16_t *d_pcm = NULL;
        const size_t n_lines = sdv_binarize_records(height, n, SDV_FLAG_NEW_FILE | SDV_FLAG_END_FILE);
        const size_t pairs_cap = n_lines * 4 + 8192, frames_cap = (size_t)n + 16, purges_cap = 8;
        HIP_OK(hipMalloc((void **)&d_luma, in.size()));
        HIP_OK(hipMalloc((void **)&d_lines, n_lines * sizeof(sdv_line_rec)));
        HIP_OK(hipMalloc((void **)&d_stats, ((size_t)n + 1) * sizeof(sdv_frame_stats)));
        HIP_OK(hipMalloc((void **)&d_pairs, pairs_cap * sizeof(sdv_sample_pair)));
        HIP_OK(hipMalloc((void **)&d_frames, frames_cap * sizeof(sdv_frame_asm)));
        HIP_OK(hipMalloc((void **)&d_audio, (pairs_cap + 1024) * sizeof(sdv_sample_pair)));
        HIP_OK(hipMalloc((void **)&d_purges, purges_cap * sizeof(sdv_audio_purge)));
        HIP_OK(hipMalloc((void **)&d_pcm, (pairs_cap + 1024) * 2 * sizeof(int16_t)));
        HIP_OK(hipMemcpy(d_luma, in.data(), in.size(), hipMemcpyHostToDevice));
        SDV_OKAY(sdv_set_mode(eng, SDV_MODE_NORMAL));
        SDV_OKAY(sdv_binarize_frames(eng, d_luma, (size_t)width, (size_t)width * height, width, height, n, 1,
                                     SDV_FLAG_NEW_FILE | SDV_FLAG_END_FILE, d_lines, n_lines, d_stats, (size_t)n + 1, NULL));
        sdv_stitch_settings st; sdv_default_stitch_settings(&st);
        SDV_OKAY(sdv_set_stitch_settings(eng, &st));
        SDV_OKAY(sdv_stitch_frames(eng, d_lines, n_lines, d_pairs, pairs_cap, &n_pairs, d_frames, frames_cap, &n_frames, NULL));
        /* the pair stream (NEW_FILE ... END_FILE) stays on the device and goes straight into the audio stage; stop = the application closing */
        size_t n_audio = 0, n_purges = 0; uint64_t n_masked = 0;
        SDV_OKAY(sdv_set_audio_masking(eng, mask_mode));
        SDV_OKAY(sdv_audio_process(eng, d_pairs, n_pairs, 1, d_audio, pairs_cap + 1024, &n_audio, d_purges, purges_cap, &n_purges, &n_masked, NULL));
        std::vector<sdv_audio_purge> purges(n_purges);
        if (n_purges) HIP_OK(hipMemcpy(purges.data(), d_purges, n_purges * sizeof(sdv_audio_purge), hipMemcpyDeviceToHost));
        /* the file of the first source: the pairs between its NEW_FILE purge and the next purge */
        size_t a = 0, b = 0; bool found = false;
        for (size_t k = 0; k < n_purges && !found; k++) if (purges[k].kind == SDV_AP_PURGE_NEW_FILE) { a = (size_t)purges[k].first_pair; b = k + 1 < n_purges ? (size_t)purges[k + 1].first_pair : n_audio; found = true; }
        if (!found || b <= a) { fprintf(stderr, "no audio came out\n"); rc = 5; }
        else {
            SDV_OKAY(sdv_wav_pack(eng, d_audio + a, b - a, d_pcm, NULL));
            HIP_OK(hipDeviceSynchronize());
            sdv_sample_pair last;
            HIP_OK(hipMemcpy(&last, d_audio + (b - 1), sizeof(last), hipMemcpyDeviceToHost));
            std::vector<uint8_t> file(44 + 4 * (b - a));
            sdv_wav_header(file.data(), b - a, last.sample_rate);
            HIP_OK(hipMemcpy(file.data() + 44, d_pcm, 4 * (b - a), hipMemcpyDeviceToHost));
            rc = write_file(argv[6], file.data(), file.size()) ? 0 : 4;
            printf("wav: %d frames -> %zu sample pairs -> %zu after the audio stage (%llu samples masked, %zu purges) -> %zu bytes at %u Hz\n", n, n_pairs, n_audio,
                   (unsigned long long)n_masked, n_purges, file.size(), last.sample_rate == 44056 ? 44056u : 44100u);
        }
        (void)hipFree(d_luma); (void)hipFree(d_lines); (void)hipFree(d_stats); (void)hipFree(d_frames); (void)hipFree(d_audio); (void)hipFree(d_purges); (void)hipFree(d_pcm);
    } else if (mode == "pcm1" && argc == 5) {
        if (!read_file(argv[2], in) || in.size() % sizeof(sdv_pcm1_line_rec)) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
        const size_t n_lines = in.size() / sizeof(sdv_pcm1_line_rec);
        sdv_pcm1_line_rec *d_lines = NULL; sdv_frame_asm_pcm1 *d_frames = NULL;
        const size_t pairs_cap = n_lines * 3 + 4096, frames_cap = n_lines / 32 + 64;
        HIP_OK(hipMalloc((void **)&d_lines, in.size() ? in.size() : 32));
        HIP_OK(hipMalloc((void **)&d_pairs, pairs_cap * sizeof(sdv_sample_pair)));
        HIP_OK(hipMalloc((void **)&d_frames, frames_cap * sizeof(sdv_frame_asm_pcm1)));
        HIP_OK(hipMemcpy(d_lines, in.data(), in.size(), hipMemcpyHostToDevice));
        sdv_pcm1_stitch_settings st; sdv_default_pcm1_stitch_settings(&st);
        SDV_OKAY(sdv_set_pcm1_stitch_settings(eng, &st));
        SDV_OKAY(sdv_pcm1_stitch_frames(eng, d_lines, n_lines, d_pairs, pairs_cap, &n_pairs, d_frames, frames_cap, &n_frames, NULL));
        HIP_OK(hipDeviceSynchronize());
        rc = download(d_pairs, n_pairs, argv[3]); if (!rc) rc = download(d_frames, n_frames, argv[4]);
        printf("pcm1: %zu line records -> %zu sample pairs, %zu frame descriptors\n", n_lines, n_pairs, n_frames);
        (void)hipFree(d_lines); (void)hipFree(d_frames);
    } else if (mode == "pcm16x0" && argc == 9) {
        const int width = atoi(argv[3]), height = atoi(argv[4]), n = atoi(argv[5]);
        const bool ei = std::string(argv[6]) == "ei";
        if (!read_file(argv[2], in) || in.size() != (size_t)width * height * n) { fprintf(stderr, "cannot read %s\n", argv[2]); return 1; }
        uint8_t *d_luma = NULL; sdv_pcm16x0_bin_rec *d_lines = NULL; sdv_frame_stats *d_stats = NULL; sdv_frame_asm_pcm16x0 *d_frames = NULL;
        const unsigned flags = SDV_FLAG_NEW_FILE | SDV_FLAG_END_FILE;
        const size_t n_lines = sdv_pcm16x0_binarize_records(height, n, flags);
        const size_t pairs_cap = (size_t)(n + 1) * 1470 + 16, frames_cap = (size_t)n + 16;
        HIP_OK(hipMalloc((void **)&d_luma, in.size()));
        HIP_OK(hipMalloc((void **)&d_lines, n_lines * sizeof(sdv_pcm16x0_bin_rec)));
        HIP_OK(hipMalloc((void **)&d_stats, ((size_t)n + 1) * sizeof(sdv_frame_stats)));
        HIP_OK(hipMalloc((void **)&d_pairs, pairs_cap * sizeof(sdv_sample_pair)));
        HIP_OK(hipMalloc((void **)&d_frames, frames_cap * sizeof(sdv_frame_asm_pcm16x0)));
        HIP_OK(hipMemcpy(d_luma, in.data(), in.size(), hipMemcpyHostToDevice));
        SDV_OKAY(sdv_set_pcm_type(eng, SDV_PCM_PCM16X0, 0));
        SDV_OKAY(sdv_set_mode(eng, SDV_MODE_NORMAL));
        SDV_OKAY(sdv_pcm16x0_binarize_frames(eng, d_luma, (size_t)width, (size_t)width * height, width, height, n, 1, flags, d_lines, n_lines,
                                             d_stats, (size_t)n + 1, NULL));
        sdv_pcm16x0_stitch_settings st; sdv_default_pcm16x0_stitch_settings(&st);
        st.format = ei ? SDV_P16_FORMAT_EI : SDV_P16_FORMAT_SI;
        SDV_OKAY(sdv_set_pcm16x0_stitch_settings(eng, &st));
        SDV_OKAY(sdv_pcm16x0_stitch_frames(eng, d_lines, n_lines, d_pairs, pairs_cap, &n_pairs, d_frames, frames_cap, &n_frames, NULL));
        HIP_OK(hipDeviceSynchronize());
        rc = download(d_pairs, n_pairs, argv[7]); if (!rc) rc = download(d_frames, n_frames, argv[8]);
        sdv_run_info info; sdv_get_run_info(eng, &info);
        printf("pcm16x0 (%s): %d frames -> %zu sub-line records -> %zu sample pairs, %zu frame descriptors (binarize rounds %u)\n", ei ? "EI" : "SI", n, n_lines, n_pairs, n_frames, info.rounds);
        (void)hipFree(d_luma); (void)hipFree(d_lines); (void)hipFree(d_stats); (void)hipFree(d_frames);
    } else { fprintf(stderr, "usage: see the header of examples/decode_tape.cpp\n"); rc = 1; }
    if (d_pairs) (void)hipFree(d_pairs);
    sdv_engine_destroy(eng);
    return rc;
}
