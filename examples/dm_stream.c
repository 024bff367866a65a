/* dm_stream.c -- the DM-trial stage of the observation loop from plain C (C99): what a maintainer puts where the reference's
 * loop has its DM-0 collapse (src/beamformer.cu:492-511).  Four PSRDADA-sized blocks of the DEBUG geometry go through
 *   bf_submit_block -> bf_queue_stream + bf_dm_stream_reserve (the place of the block's rows INSIDE the DM stage's buffer)
 *   -> bf_enqueue_block_to (one fused launch per block writes its powers there; they are also copied to the host)
 *   -> bf_dm_stream_push (the rows are in place: the stage only launches its kernels, on the same queue -- no copy in between,
 *      as the reference's collapse reads what detect wrote)
 * and every chunk [dm][t][beam] the stream emits is checked, bit for bit, against the same ascending-f float sum computed
 * here from the detected powers -- across the block boundaries: the largest delay is longer than one block.
 * Build:  hipcc -x c -std=c99 -Iinclude examples/dm_stream.c -o dm_stream -Ldsabeamformer_amd -ldsabf -Wl,-rpath,$PWD/dsabeamformer_amd */
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

#define N_BLOCKS 4
#define N_DM 5

int main(void)
{
    bf_config cfg;
    bf_handle *h = NULL;
    bf_dm_stream *dm = NULL;
    void *block = NULL, *det = NULL, *chunk = NULL;
    int8_t *w;
    float *pos, *dir;
    int32_t *delays;
    size_t block_bytes, per_unit, i;
    int n_dev = 0, rows_per_block, n_rows, b, k, f, max_delay = 0;
    uint64_t next_t = 0, x = 0x9E3779B97F4A7C15ull;

    CHECK(bf_config_default(&cfg, /*debug=*/1));
    cfg.n_freq = 16;            /* a small sub-band keeps the host check quick */
    cfg.n_beams = 64;
    cfg.n_gemms_per_block = 4;  /* 4 gemm-units x 8 outputs = 32 beam-blocks (rows) per block */
    cfg.n_streams = 2;
    if (bf_device_count(&n_dev) != BF_OK || n_dev == 0) {
        printf("no gfx950 device: %s\n", bf_last_error());
        return 2;
    }
    CHECK(bf_create(&cfg, 0, &h));
    pos = (float *)malloc(sizeof(float) * 3 * (size_t)cfg.n_ant);
    dir = (float *)malloc(sizeof(float) * 2 * (size_t)cfg.n_beams);
    w = (int8_t *)malloc((size_t)cfg.n_freq * cfg.n_ant * cfg.n_beams * 2);
    CHECK(bfh_default_positions(cfg.n_ant, pos));
    CHECK(bfh_default_directions(cfg.n_beams, dir));
    CHECK(bfh_make_weights(cfg.n_beams, cfg.n_ant, cfg.n_freq, 0, 0, pos, dir, w));
    CHECK(bf_set_weights(h, w));

    rows_per_block = cfg.n_gemms_per_block * cfg.n_out_per_gemm;
    n_rows = N_BLOCKS * rows_per_block;
    per_unit = bf_floats_per_detect(&cfg);
    /* delay[dm][f]: grows with the trial, falls with f (channel 0 is the highest frequency); the last trial's window (45 rows)
     * is longer than a block (32 rows) */
    delays = (int32_t *)malloc(sizeof(int32_t) * N_DM * (size_t)cfg.n_freq);
    for (k = 0; k < N_DM; k++)
        for (f = 0; f < cfg.n_freq; f++) {
            delays[k * cfg.n_freq + f] = (int32_t)(k * 3 * (cfg.n_freq - 1 - f) / 4);
            if (delays[k * cfg.n_freq + f] > max_delay) max_delay = delays[k * cfg.n_freq + f];
        }
    CHECK(bf_dm_stream_create(h, delays, N_DM, cfg.n_freq, rows_per_block, &dm));
    if (bf_dm_stream_max_delay(dm) != max_delay || max_delay <= rows_per_block) {
        fprintf(stderr, "unexpected max delay %d\n", max_delay);
        return 1;
    }

    block_bytes = bf_bytes_per_block(&cfg);
    CHECK(bf_alloc_pinned(&block, block_bytes));
    CHECK(bf_alloc_pinned(&det, (size_t)n_rows * cfg.n_freq * cfg.n_beams * sizeof(float)));   /* the whole detected series */
    CHECK(bf_alloc_pinned(&chunk, (size_t)N_DM * rows_per_block * cfg.n_beams * sizeof(float)));

    for (b = 0; b < N_BLOCKS; b++) {
        float *host_ptrs[64];
        float *d_rows = NULL;
        void *queue = NULL;
        const int q = b % 2;    /* whole blocks alternate between two queues */
        uint64_t first_t = 0;
        int n_t = 0, t, bm;
        for (i = 0; i < block_bytes; i++) {   /* pseudo-random nibbles (xorshift) */
            x ^= x << 13, x ^= x >> 7, x ^= x << 17;
            ((unsigned char *)block)[i] = (unsigned char)(x >> 24);
        }
        for (i = 0; i < (size_t)cfg.n_gemms_per_block; i++)
            host_ptrs[i] = (float *)det + ((size_t)b * cfg.n_gemms_per_block + i) * per_unit;
        CHECK(bf_submit_block(h, b % cfg.n_blocks_on_gpu, block, block_bytes, NULL));
        CHECK(bf_stream_sync(h, -1));   /* (this example reuses one host block; the loop proper waits on the transfer event) */
        /* the DM stage owns the place of this block's rows ([unit][o][f][b] = [row][f][b]): the launch writes them there */
        CHECK(bf_queue_stream(h, q, &queue));
        CHECK(bf_dm_stream_reserve(dm, rows_per_block, &d_rows, queue));
        CHECK(bf_enqueue_block_to(h, q, b % cfg.n_blocks_on_gpu, 0, cfg.n_gemms_per_block, d_rows, host_ptrs));
        CHECK(bf_dm_stream_push(dm, d_rows, rows_per_block, (float *)chunk, &first_t, &n_t, queue));
        CHECK(bf_stream_sync(h, q));
        if (first_t != next_t) {
            fprintf(stderr, "block %d: chunk starts at %llu, expected %llu\n", b, (unsigned long long)first_t, (unsigned long long)next_t);
            return 1;
        }
        printf("block %d: rows %d .. %d pushed, output times %llu .. %llu complete\n", b, b * rows_per_block, (b + 1) * rows_per_block - 1,
               (unsigned long long)first_t, (unsigned long long)(first_t + (uint64_t)n_t) - 1);
        /* every emitted sum against the same ascending-f float chain over the detected powers (which reach back into
         * earlier blocks: the stream carried those rows over on the device) */
        for (k = 0; k < N_DM; k++)
            for (t = 0; t < n_t; t++)
                for (bm = 0; bm < cfg.n_beams; bm++) {
                    volatile float acc = 0.0f;   /* (volatile: one rounding per add, no reassociation) */
                    for (f = 0; f < cfg.n_freq; f++) {
                        const size_t row = (size_t)first_t + (size_t)t + (size_t)delays[k * cfg.n_freq + f];
                        acc = acc + ((const float *)det)[(row * cfg.n_freq + (size_t)f) * cfg.n_beams + (size_t)bm];
                    }
                    if (((const float *)chunk)[((size_t)k * n_t + (size_t)t) * cfg.n_beams + (size_t)bm] != acc) {
                        fprintf(stderr, "mismatch at block %d trial %d time %d beam %d\n", b, k, t, bm);
                        return 1;
                    }
                }
        next_t += (uint64_t)n_t;
    }
    if (next_t != (uint64_t)(n_rows - max_delay)) {
        fprintf(stderr, "emitted %llu output times, expected %d\n", (unsigned long long)next_t, n_rows - max_delay);
        return 1;
    }
    CHECK(bf_dm_stream_destroy(dm));   /* the stage before its handle, pinned memory after it (include/dsabf.h) */
    CHECK(bf_destroy(h));
    CHECK(bf_free_pinned(block));
    CHECK(bf_free_pinned(det));
    CHECK(bf_free_pinned(chunk));
    free(pos);
    free(dir);
    free(w);
    free(delays);
    printf("%d blocks, %d trials, %llu output times: every sum bit-equal to the host chain\nok\n", N_BLOCKS, N_DM, (unsigned long long)next_t);
    return 0;
}
