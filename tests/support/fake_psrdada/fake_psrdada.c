/* fake_psrdada.c -- TEST INFRASTRUCTURE ONLY: a functional stand-in for the part of libpsrdada that the reference's
 * src/dada_handler.hh (and therefore dsabeamformer_amd/csrc/bf_dada.cpp) calls, so that the -DDSABF_WITH_PSRDADA branch of the
 * adapter can be EXECUTED in an image that has no libpsrdada: one reader (the beamformer), one writer (the test), a header
 * block and a ring of nbufs data blocks in one POSIX shared-memory segment named after the hex key.  It implements the
 * declarations of tests/support/psrdada_api/ (the names and argument lists of psrdada's public headers as the reference uses
 * them) with the semantics the reference relies on: connect / lock_read, one header block read and cleared, data blocks handed
 * out in write order with their valid byte count, a block is free for the writer again once the reader closed it, a block
 * shorter than the buffer size ends the data.  Nothing here is shipped, linked into libdsabf.so or used to build a reference
 * binary; it says nothing about real PSRDADA's performance or its SysV IPC details.
 * The writer's side (fakedada_*) is what dada_db + dada_junkdb would be (makefile:28-33). */
#define _GNU_SOURCE
#include <fcntl.h>
#include <stdarg.h>
#include <stdatomic.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "dada_hdu.h"

#define FAKE_HDR_BYTES 4096
#define FAKE_MAX_BUFS 64

typedef struct {
    ipcsync_t data_sync;                 /* what ring->sync points at: key, nbufs, bufsz */
    ipcsync_t hdr_sync;
    _Atomic uint64_t w_count, r_count;   /* data blocks written / closed by the reader */
    _Atomic int header_written, header_cleared, reader_connected;
    uint64_t bytes[FAKE_MAX_BUFS];       /* valid bytes of the block in each buffer */
    char pad[256];
} fake_ctl;

struct multilog_s {
    FILE *out[4];
    int n;
    char name[64];
};

typedef struct {
    dada_hdu_t hdu;     /* first member: what callers hold */
    key_t key;
    fake_ctl *ctl;
    size_t map_bytes;
    ipcio_t data;
    ipcbuf_t header;
    char *data_ptr[FAKE_MAX_BUFS];
    char *hdr_ptr[1];
} fake_hdu;

static void seg_name(key_t key, char *buf, size_t n) { snprintf(buf, n, "/dsabf_fakedada_%x", (unsigned)key); }
static size_t seg_bytes(uint64_t nbufs, uint64_t bufsz) { return sizeof(fake_ctl) + FAKE_HDR_BYTES + (size_t)nbufs * (size_t)bufsz; }
static void nap(void)
{
    struct timespec ts = {0, 200000};
    nanosleep(&ts, NULL);
}

/* ---- multilog.h --------------------------------------------------------------------------------------------------------- */
multilog_t *multilog_open(const char *program_name, char syslog_)
{
    multilog_t *m = (multilog_t *)calloc(1, sizeof *m);
    (void)syslog_;
    if (m) snprintf(m->name, sizeof m->name, "%s", program_name ? program_name : "");
    return m;
}
int multilog_add(multilog_t *m, FILE *fptr)
{
    if (!m || m->n >= 4) return -1;
    m->out[m->n++] = fptr;
    return 0;
}
int multilog(multilog_t *m, int priority, const char *format, ...)
{
    int i;
    (void)priority;
    for (i = 0; m && i < m->n; i++) {
        va_list ap;
        va_start(ap, format);
        fprintf(m->out[i], "%s: ", m->name);
        vfprintf(m->out[i], format, ap);
        va_end(ap);
    }
    return 0;
}

/* ---- dada_hdu.h --------------------------------------------------------------------------------------------------------- */
dada_hdu_t *dada_hdu_create(multilog_t *log)
{
    fake_hdu *f = (fake_hdu *)calloc(1, sizeof *f);
    if (!f) return NULL;
    f->hdu.log = log;
    return &f->hdu;
}
void dada_hdu_set_key(dada_hdu_t *hdu, key_t key) { ((fake_hdu *)hdu)->key = key; }

int dada_hdu_connect(dada_hdu_t *hdu)
{
    fake_hdu *f = (fake_hdu *)hdu;
    char name[64];
    struct stat st;
    uint64_t b;
    int fd = -1, tries;
    seg_name(f->key, name, sizeof name);
    for (tries = 0; tries < 25000 && fd < 0; tries++) {   /* the ring may be created a moment later: up to 5 s */
        fd = shm_open(name, O_RDWR, 0600);
        if (fd < 0) nap();
    }
    if (fd < 0) return -1;
    for (tries = 0; tries < 25000; tries++) {             /* ... and sized a moment after that */
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(fake_ctl)) break;
        nap();
    }
    f->map_bytes = (size_t)st.st_size;
    f->ctl = (fake_ctl *)mmap(NULL, f->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (f->ctl == MAP_FAILED) return -1;
    if (seg_bytes(f->ctl->data_sync.nbufs, f->ctl->data_sync.bufsz) != f->map_bytes) return -1;
    f->hdr_ptr[0] = (char *)f->ctl + sizeof(fake_ctl);
    for (b = 0; b < f->ctl->data_sync.nbufs; b++) f->data_ptr[b] = (char *)f->ctl + sizeof(fake_ctl) + FAKE_HDR_BYTES + (size_t)b * f->ctl->data_sync.bufsz;
    f->data.buf.sync = &f->ctl->data_sync;
    f->data.buf.buffer = f->data_ptr;
    f->header.sync = &f->ctl->hdr_sync;
    f->header.buffer = f->hdr_ptr;
    f->hdu.data_block = &f->data;
    f->hdu.header_block = &f->header;
    return 0;
}
int dada_hdu_disconnect(dada_hdu_t *hdu)
{
    fake_hdu *f = (fake_hdu *)hdu;
    if (f->ctl && f->ctl != MAP_FAILED) munmap(f->ctl, f->map_bytes);
    f->ctl = NULL;
    return 0;
}
int dada_hdu_lock_read(dada_hdu_t *hdu)
{
    fake_hdu *f = (fake_hdu *)hdu;
    int expected = 0;
    return atomic_compare_exchange_strong(&f->ctl->reader_connected, &expected, 1) ? 0 : -1;   /* one reader */
}
int dada_hdu_unlock_read(dada_hdu_t *hdu)
{
    fake_hdu *f = (fake_hdu *)hdu;
    if (!f->ctl) return -1;
    atomic_store(&f->ctl->reader_connected, 0);
    return 0;
}
void dada_hdu_destroy(dada_hdu_t *hdu)
{
    if (!hdu) return;
    dada_hdu_disconnect(hdu);
    free(hdu);
}

/* ---- ipcbuf.h / ipcio.h --------------------------------------------------------------------------------------------------- */
static fake_ctl *ctl_of(ipcbuf_t *id)   /* both ipcsync_t live inside the control block: the header's is the second member */
{
    char *s = (char *)id->sync;
    if (id->buffer[0] == s - offsetof(fake_ctl, hdr_sync) + sizeof(fake_ctl)) return (fake_ctl *)(s - offsetof(fake_ctl, hdr_sync));
    return (fake_ctl *)(s - offsetof(fake_ctl, data_sync));
}

char *ipcbuf_get_next_read(ipcbuf_t *id, uint64_t *bytes)   /* used on the HEADER block only (src/dada_handler.hh:67) */
{
    fake_ctl *c = ctl_of(id);
    int tries;
    for (tries = 0; tries < 150000 && !atomic_load(&c->header_written); tries++) nap();   /* up to 30 s */
    if (!atomic_load(&c->header_written)) return NULL;
    if (bytes) *bytes = FAKE_HDR_BYTES;
    return id->buffer[0];
}
int ipcbuf_mark_cleared(ipcbuf_t *id)
{
    atomic_store(&ctl_of(id)->header_cleared, 1);
    return 0;
}
uint64_t ipcbuf_get_bufsz(ipcbuf_t *id) { return id->sync->bufsz; }
int ipcbuf_lock(ipcbuf_t *id)
{
    (void)id;   /* shared memory is not swapped out from under a mapping that is about to be page-locked by the caller */
    return 0;
}
int ipcbuf_get_device(ipcbuf_t *id)
{
    (void)id;
    return -1;   /* host memory */
}

char *ipcio_open_block_read(ipcio_t *ipc, uint64_t *curbufsz, uint64_t *block_id)
{
    fake_ctl *c = ctl_of(&ipc->buf);
    const uint64_t r = atomic_load(&c->r_count);
    long tries;
    for (tries = 0; tries < 600000 && atomic_load(&c->w_count) <= r; tries++) nap();   /* blocks while the ring is empty (2 min) */
    if (atomic_load(&c->w_count) <= r) return NULL;
    if (curbufsz) *curbufsz = c->bytes[r % c->data_sync.nbufs];
    if (block_id) *block_id = r;
    return ipc->buf.buffer[r % c->data_sync.nbufs];
}
ssize_t ipcio_close_block_read(ipcio_t *ipc, uint64_t bytes)
{
    fake_ctl *c = ctl_of(&ipc->buf);
    atomic_fetch_add(&c->r_count, 1);   /* the writer may reuse the buffer from here on */
    return (ssize_t)bytes;
}

/* ---- the writer's side: dada_db -k key -n nbufs -b bufsz, then dada_junkdb-like writes ------------------------------------ */
typedef struct {
    fake_ctl *ctl;
    size_t map_bytes;
    key_t key;
} fakedada_writer;

fakedada_writer *fakedada_create(unsigned key, uint64_t nbufs, uint64_t bufsz, const char *header_text)
{
    char name[64];
    fakedada_writer *w;
    int fd;
    if (nbufs < 1 || nbufs > FAKE_MAX_BUFS) return NULL;
    seg_name((key_t)key, name, sizeof name);
    shm_unlink(name);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return NULL;
    w = (fakedada_writer *)calloc(1, sizeof *w);
    w->map_bytes = seg_bytes(nbufs, bufsz);
    w->key = (key_t)key;
    if (ftruncate(fd, (off_t)w->map_bytes) != 0) {
        close(fd);
        free(w);
        return NULL;
    }
    w->ctl = (fake_ctl *)mmap(NULL, w->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (w->ctl == MAP_FAILED) {
        free(w);
        return NULL;
    }
    w->ctl->data_sync.key = (key_t)key;
    w->ctl->data_sync.nbufs = nbufs;
    w->ctl->data_sync.bufsz = bufsz;
    w->ctl->hdr_sync.key = (key_t)key + 1;
    w->ctl->hdr_sync.nbufs = 1;
    w->ctl->hdr_sync.bufsz = FAKE_HDR_BYTES;
    snprintf((char *)w->ctl + sizeof(fake_ctl), FAKE_HDR_BYTES, "%s", header_text ? header_text : "HDR_SIZE 4096\n");
    atomic_store(&w->ctl->header_written, 1);
    return w;
}

/* bytes < bufsz ends the data (0: an empty last block); blocks while the ring is full; -1 after 2 minutes without a free block */
int fakedada_write(fakedada_writer *w, const void *data, uint64_t bytes)
{
    fake_ctl *c = w->ctl;
    const uint64_t n = atomic_load(&c->w_count), nb = c->data_sync.nbufs;
    long tries;
    if (bytes > c->data_sync.bufsz) return -2;
    for (tries = 0; tries < 600000 && n - atomic_load(&c->r_count) >= nb; tries++) nap();
    if (n - atomic_load(&c->r_count) >= nb) return -1;
    if (bytes) memcpy((char *)c + sizeof(fake_ctl) + FAKE_HDR_BYTES + (size_t)(n % nb) * c->data_sync.bufsz, data, bytes);
    c->bytes[n % nb] = bytes;
    atomic_fetch_add(&c->w_count, 1);
    return 0;
}
uint64_t fakedada_blocks_read(fakedada_writer *w) { return atomic_load(&w->ctl->r_count); }
int fakedada_header_cleared(fakedada_writer *w) { return atomic_load(&w->ctl->header_cleared); }
void fakedada_destroy(fakedada_writer *w)
{
    char name[64];
    if (!w) return;
    seg_name(w->key, name, sizeof name);
    munmap(w->ctl, w->map_bytes);
    shm_unlink(name);
    free(w);
}
