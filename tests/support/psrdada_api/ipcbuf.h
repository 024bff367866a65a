/* declarations only (tests/support/psrdada_api/README.md); used at src/dada_handler.hh:67,75,83,129-151 */
#ifndef DSABF_TEST_IPCBUF_H
#define DSABF_TEST_IPCBUF_H
#include <stdint.h>
#include <sys/types.h>
typedef struct {
    key_t key;
    uint64_t nbufs; /* the number of buffers in the ring */
    uint64_t bufsz; /* the size of each buffer */
} ipcsync_t;
typedef struct {
    int state;
    ipcsync_t *sync; /* pointer to sync structure in shared memory */
    char **buffer;   /* base addresses of the sub-blocks */
} ipcbuf_t;
char *ipcbuf_get_next_read(ipcbuf_t *id, uint64_t *bytes);
int ipcbuf_mark_cleared(ipcbuf_t *id);
uint64_t ipcbuf_get_bufsz(ipcbuf_t *id);
int ipcbuf_lock(ipcbuf_t *id);
int ipcbuf_get_device(ipcbuf_t *id);
#endif
