/* sharded.c -- one rank of a frequency-sharded run from plain C (C99): the multi-GPU half of the C-ABI (include/dsabf.h,
 * "Multi-GPU").  Every rank is its own process and owns n_freq / world channels of the reference's DEBUG geometry:
 *   bf_create (local channel count) -> weights of ITS channels -> bf_comm_create (RCCL, id handed over through a file)
 *   -> bf_submit_block + bf_enqueue_block (one launch for the block, powers stay in HBM)
 *   -> bf_gather_detected to rank 0 in the reference's [o][f][b] order, straight into the handle's gather buffer
 *   -> rank 0: D2H of the gathered block (bf_enqueue_d2h) and a look at it.
 * usage: sharded <rank> <world> <id-file> [device]        (start world processes; rank 0 writes the id file)
 * Build: hipcc -x c -std=c99 -Iinclude examples/sharded.c -o sharded -Ldsabeamformer_amd -ldsabf -Wl,-rpath,$PWD/dsabeamformer_amd
 * With world == 1 no RCCL is touched (bf_comm_create needs no id) and the program is a single-GPU run. */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "dsabf.h"
#include "dsabf_host.h"

#define CHECK(x)                                                                  \
    do {                                                                          \
        int rc_ = (x);                                                            \
        if (rc_ < 0) {                                                            \
            fprintf(stderr, "rank %d: %s (%s:%d)\n", rank, bf_last_error(), __FILE__, __LINE__); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

static int read_id(const char *path, char *id)
{
    int tries;
    for (tries = 0; tries < 12000; tries++) { /* rank 0 renames the file into place: it is complete once it exists */
        FILE *fp = fopen(path, "rb");
        if (fp) {
            const size_t n = fread(id, 1, BF_COMM_ID_BYTES, fp);
            fclose(fp);
            if (n == BF_COMM_ID_BYTES) return 0;
        }
        {
            struct timespec ts = {0, 10 * 1000 * 1000};
            nanosleep(&ts, NULL);
        }
    }
    return -1;
}

int main(int argc, char **argv)
{
    int rank = 0, world = 1, device = 0, n_dev = 0;
    bf_config cfg;
    bf_handle *h = NULL;
    bf_comm *comm = NULL;
    char id[BF_COMM_ID_BYTES];
    float *pos, *dir, *d_local = NULL, *d_full = NULL;
    int8_t *w;
    void *block = NULL, *stream = NULL, *host_full = NULL;
    size_t block_bytes, n_rows, row_floats, i;
    int n_freq_total;

    if (argc < 4) {
        fprintf(stderr, "usage: %s <rank> <world> <id-file> [device [staged]]\n", argv[0]);
        return 2;
    }
    rank = atoi(argv[1]);
    world = atoi(argv[2]);
    if (argc > 4) device = atoi(argv[4]);
    if (bf_device_count(&n_dev) != BF_OK || n_dev == 0) {
        printf("no gfx950 device: %s\n", bf_last_error());
        return 2;
    }

    CHECK(bf_config_default(&cfg, /*debug=*/1));
    n_freq_total = cfg.n_freq;
    if (world < 1 || rank < 0 || rank >= world || n_freq_total % world) {
        fprintf(stderr, "need 0 <= rank < world and world dividing %d channels\n", n_freq_total);
        return 2;
    }
    cfg.n_freq = n_freq_total / world;              /* this rank's channels: [rank * n_freq, (rank + 1) * n_freq) */
    cfg.n_gemms_per_block = 4;                      /* a small block keeps the example quick */
    CHECK(bf_create(&cfg, device, &h));

    pos = (float *)malloc(sizeof(float) * 3 * (size_t)cfg.n_ant);
    dir = (float *)malloc(sizeof(float) * 2 * (size_t)cfg.n_beams);
    w = (int8_t *)malloc((size_t)cfg.n_freq * cfg.n_ant * cfg.n_beams * 2);
    CHECK(bfh_default_positions(cfg.n_ant, pos));
    CHECK(bfh_default_directions(cfg.n_beams, dir));
    CHECK(bfh_make_weights(cfg.n_beams, cfg.n_ant, cfg.n_freq, /*chan0=*/rank * cfg.n_freq, /*gpu=*/0, pos, dir, w));
    CHECK(bf_set_weights(h, w));

    /* the communicator: rank 0 draws the id and hands its 128 bytes to the others (here: a file) */
    if (world > 1) {
        if (rank == 0) {
            char tmp[4096];
            FILE *fp;
            CHECK(bf_comm_unique_id(id));
            snprintf(tmp, sizeof tmp, "%s.tmp", argv[3]);
            fp = fopen(tmp, "wb");
            if (!fp || fwrite(id, 1, BF_COMM_ID_BYTES, fp) != BF_COMM_ID_BYTES || fclose(fp) != 0 || rename(tmp, argv[3]) != 0) {
                fprintf(stderr, "cannot write %s\n", argv[3]);
                return 1;
            }
        } else if (read_id(argv[3], id) != 0) {
            fprintf(stderr, "rank %d: no id in %s\n", rank, argv[3]);
            return 1;
        }
    }
    CHECK(bf_comm_create(rank, world, world > 1 ? id : NULL, device, &comm));

    /* this rank's slice of one block of BOGUS_DATA 0x70 = (7 + 0j) everywhere (src/test_data_generator.hh:8) */
    block_bytes = bf_bytes_per_block(&cfg);
    CHECK(bf_alloc_pinned(&block, block_bytes));
    memset(block, 0x70, block_bytes);
    CHECK(bf_submit_block(h, /*slot=*/0, block, block_bytes, NULL));
    CHECK(bf_enqueue_block(h, /*stream=*/0, /*slot=*/0, /*first_unit=*/0, cfg.n_gemms_per_block, NULL));

    /* the one collective: every rank's rows to rank 0, received at their place in [unit][o][f over the band][b] */
    n_rows = (size_t)cfg.n_gemms_per_block * cfg.n_out_per_gemm;
    row_floats = (size_t)cfg.n_freq * cfg.n_beams;
    CHECK(bf_block_output_device(h, 0, &d_local));
    CHECK(bf_queue_stream(h, 0, &stream));
    if (rank == 0) CHECK(bf_block_gather_device(h, 0, world, &d_full));
    if (argc > 5 && !strcmp(argv[5], "staged")) {
        /* the same [o][f][b] by the second transport: ONE message per sender into a staging area, then one device re-layout pass */
        float *d_stage = NULL;
        if (rank == 0) CHECK(bf_block_gather_stage_device(h, 0, world, &d_stage));
        CHECK(bf_gather_detected_staged(comm, d_local, n_rows, row_floats, /*root=*/0, d_full, d_stage, stream));
    } else {
        CHECK(bf_gather_detected(comm, d_local, n_rows, row_floats, /*root=*/0, BF_GATHER_LAYOUT_FREQ_MAJOR, d_full, stream));
    }

    if (rank == 0) {
        const size_t full_floats = n_rows * (size_t)world * row_floats;
        CHECK(bf_alloc_pinned(&host_full, full_floats * sizeof(float)));
        CHECK(bf_enqueue_d2h(h, 0, d_full, (float *)host_full, full_floats));
    }
    CHECK(bf_stream_sync(h, -1));

    if (rank == 0) {
        /* every gemm-unit saw the same voltages, so every row of the band equals row 0; and the boresight plane wave
         * puts the same power into mirror-image beams */
        const float *full = (const float *)host_full;
        const size_t band = (size_t)world * row_floats;
        double *ded = (double *)calloc((size_t)cfg.n_beams, sizeof(double));
        size_t f;
        int best = 0;
        for (i = 1; i < n_rows; i++)
            if (memcmp(full, full + i * band, band * sizeof(float)) != 0) {
                fprintf(stderr, "row %zu of the gathered band differs from row 0\n", i);
                return 1;
            }
        for (f = 0; f < (size_t)n_freq_total; f++)          /* DM 0 over the whole band, row 0 */
            for (i = 0; i < (size_t)cfg.n_beams; i++) ded[i] += full[f * cfg.n_beams + i];
        for (i = 0; i < (size_t)cfg.n_beams; i++)
            if (ded[i] > ded[best]) best = (int)i;
        printf("world %d: gathered %zu rows x %d channels x %d beams; band-summed peak beam %d = %g\n", world, n_rows,
               n_freq_total, cfg.n_beams, best, ded[best]);
        if (ded[best] != ded[cfg.n_beams - 1 - best] || !(ded[best] > 0)) {
            fprintf(stderr, "unexpected beam pattern\n");
            return 1;
        }
        free(ded);
    }
    CHECK(bf_comm_destroy(comm));
    CHECK(bf_destroy(h));           /* pinned memory after the handle (include/dsabf.h) */
    if (rank == 0) CHECK(bf_free_pinned(host_full));
    CHECK(bf_free_pinned(block));
    free(pos);
    free(dir);
    free(w);
    printf("rank %d ok\n", rank);
    return 0;
}
