/* declarations only (tests/support/psrdada_api/README.md); used at src/dada_handler.hh:30-44,119-122 */
#ifndef DSABF_TEST_DADA_HDU_H
#define DSABF_TEST_DADA_HDU_H
#include "ipcio.h"
#include "multilog.h"
typedef struct {
    multilog_t *log;
    ipcio_t *data_block;
    ipcbuf_t *header_block;
} dada_hdu_t;
dada_hdu_t *dada_hdu_create(multilog_t *log);
void dada_hdu_set_key(dada_hdu_t *hdu, key_t key);
int dada_hdu_connect(dada_hdu_t *hdu);
int dada_hdu_disconnect(dada_hdu_t *hdu);
int dada_hdu_lock_read(dada_hdu_t *hdu);
int dada_hdu_unlock_read(dada_hdu_t *hdu);
void dada_hdu_destroy(dada_hdu_t *hdu);
#endif
