/* audio.h - AudioProcessor / SamplesToWAV restatement (oracle/audio.c). TEST INFRASTRUCTURE ONLY. */
#ifndef ORC_AUDIO_H
#define ORC_AUDIO_H
#include "sdv_oracle.h"
#include "../include/sdvpcm.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct orc_audio orc_audio;
orc_audio *orc_audio_new(int mask_mode);
void orc_audio_free(orc_audio *h);
void orc_audio_set_masking(orc_audio *h, int mode);
size_t orc_audio_pending(const orc_audio *h);
int orc_audio_hit_unsupported(const orc_audio *h);
long orc_audio_process(orc_audio *h, const sdv_sample_pair *pairs, size_t n_pairs, int stop, sdv_sample_pair *out, uint64_t *out_index, size_t out_cap,
                       sdv_audio_purge *purges, size_t purges_cap, size_t *n_purges, uint64_t *n_masked);
long orc_audio_run(const sdv_sample_pair *pairs, size_t n, const uint64_t *bursts, size_t n_bursts, int mask_mode, int stop,
                   sdv_sample_pair *out, uint64_t *out_index, size_t out_cap, sdv_audio_purge *purges, size_t purges_cap, size_t *n_purges,
                   uint64_t *n_masked, int *hit_unsupported);
void orc_wav_header(uint8_t hdr[44], uint64_t n_pairs, uint16_t last_sample_rate);
void orc_wav_pack(const sdv_sample_pair *pairs, size_t n, int16_t *pcm);
#ifdef __cplusplus
}
#endif
#endif
