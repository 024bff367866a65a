/* minimal.c -- the C-ABI of libdsabf.so from plain C (C99): one gemm-unit of the reference's DEBUG geometry through
 * the streaming entry points, exactly the calls a maintainer substitutes into src/beamformer.cu:
 *   bf_create -> bf_set_weights -> bf_submit_block (H2D) -> bf_enqueue_gemm_unit (expand+GEMM+detect, D2H)
 *   -> bf_enqueue_dedisperse -> bf_stream_sync -> bf_destroy.
 * Build:  hipcc -x c -std=c99 -Iinclude examples/minimal.c -o minimal -Ldsabeamformer_amd -ldsabf -Wl,-rpath,$PWD/dsabeamformer_amd
 * (any C compiler works; the program must also link the HIP runtime, which hipcc adds). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsabf.h"
#include "dsabf_host.h"

#define CHECK(x)                                                                  \
    do {                                                                          \
        int rc_ = (x);                                                            \
        if (rc_ < 0) {                                                            \
            fprintf(stderr, "GPUassert: %s (%s:%d)\n", bf_last_error(), __FILE__, __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

int main(void)
{
    bf_config cfg;
    bf_handle *h = NULL;
    void *block = NULL, *out = NULL, *ded = NULL;
    int8_t *w;
    float *pos, *dir;
    size_t block_bytes, i;
    int n_dev = 0;

    CHECK(bf_config_default(&cfg, /*debug=*/1));
    if (bf_device_count(&n_dev) != BF_OK || n_dev == 0) {
        printf("no gfx950 device: %s\n", bf_last_error());
        return 2;
    }
    CHECK(bf_create(&cfg, 0, &h));

    /* steering weights of the default linear geometry (src/beamformer.cu:135-147, 230-241) */
    pos = (float *)malloc(sizeof(float) * 3 * (size_t)cfg.n_ant);
    dir = (float *)malloc(sizeof(float) * 2 * (size_t)cfg.n_beams);
    w = (int8_t *)malloc((size_t)cfg.n_freq * cfg.n_ant * cfg.n_beams * 2);
    CHECK(bfh_default_positions(cfg.n_ant, pos));
    CHECK(bfh_default_directions(cfg.n_beams, dir));
    CHECK(bfh_make_weights(cfg.n_beams, cfg.n_ant, cfg.n_freq, 0, 0, pos, dir, w));
    CHECK(bf_set_weights(h, w));

    /* one PSRDADA-sized block of BOGUS_DATA 0x70 = (7 + 0j) everywhere (src/test_data_generator.hh:8) */
    block_bytes = bf_bytes_per_block(&cfg);
    CHECK(bf_alloc_pinned(&block, block_bytes));
    CHECK(bf_alloc_pinned(&out, bf_floats_per_detect(&cfg) * sizeof(float)));
    CHECK(bf_alloc_pinned(&ded, (size_t)cfg.n_beams * sizeof(float)));
    memset(block, 0x70, block_bytes);

    CHECK(bf_submit_block(h, /*slot=*/0, block, block_bytes, NULL));
    CHECK(bf_enqueue_gemm_unit(h, /*stream=*/0, /*slot=*/0, /*time_slice=*/0, (float *)out));
    CHECK(bf_enqueue_dedisperse(h, 0, (float *)ded));
    CHECK(bf_stream_sync(h, -1));

    /* (7 + 0j) on every antenna is a plane wave from the boresight: the beam pattern (main lobe and the grating lobes of
     * the 7.9 m antenna spacing) stands far above the mean, and mirror-image beams B-1-b and b see the same power */
    {
        const float *d = (const float *)ded;
        double mean = 0;
        int best = 0;
        for (i = 0; i < (size_t)cfg.n_beams; i++) {
            mean += d[i] / cfg.n_beams;
            if (d[i] > d[best]) best = (int)i;
        }
        printf("dedispersed power: peak beam %d = %g, mean %g (%s)\n", best, d[best], mean, bf_version());
        if (!(d[best] > 10 * mean) || d[best] != d[cfg.n_beams - 1 - best]) {
            fprintf(stderr, "unexpected beam pattern\n");
            return 1;
        }
    }
    CHECK(bf_destroy(h));           /* drains the queues; the host buffers of enqueued units outlive it (include/dsabf.h) */
    CHECK(bf_free_pinned(block));
    CHECK(bf_free_pinned(out));
    CHECK(bf_free_pinned(ded));
    free(pos);
    free(dir);
    free(w);
    printf("ok\n");
    return 0;
}
