/* declarations only (tests/support/psrdada_api/README.md); used at src/dada_handler.hh:93,97 */
#ifndef DSABF_TEST_IPCIO_H
#define DSABF_TEST_IPCIO_H
#include "ipcbuf.h"
typedef struct {
    ipcbuf_t buf; /* first member: the reference casts ipcio_t* to ipcbuf_t* (src/dada_handler.hh:83,129) */
} ipcio_t;
char *ipcio_open_block_read(ipcio_t *ipc, uint64_t *curbufsz, uint64_t *block_id);
ssize_t ipcio_close_block_read(ipcio_t *ipc, uint64_t bytes);
#endif
