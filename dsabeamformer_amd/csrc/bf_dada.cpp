// bf_dada.cpp -- the PSRDADA input adapter: dsabf::dada_block_source is the reference's dada_handler
// (src/dada_handler.hh:1-177) behind the block_source interface the observation loop reads from, statement for statement.
//
// Compiled only with -DDSABF_WITH_PSRDADA (DSABF_WITH_PSRDADA=1 [PSRDADA_INCLUDE=... PSRDADA_LIB=...] python -m
// dsabeamformer_amd.build): libpsrdada is not part of this image, so the default build carries the shared-memory stand-in
// (bf_shmring.cpp) instead and this file is empty.  With the flag, `beam -k <hex key>` connects to a real PSRDADA ring
// (dada_db -k baab ..., makefile:31-33) exactly as the reference's `bin/beam -k baab` does; `beam -k <name>` with a name that
// is not a hex number still opens a shm_ring.
#ifdef DSABF_WITH_PSRDADA
#include <sched.h>

#include <cstdio>
#include <ostream>

// psrdada's public C headers (makefile:5, DADA_INCLUDE)
extern "C" {
#include "dada_hdu.h"
#include "ipcbuf.h"
#include "ipcio.h"
#include "multilog.h"
}

#include "../../include/dsabf.h"
#include "../../include/dsabf_host.hpp"

namespace dsabf {

// src/dada_handler.hh:25-60
dada_block_source::dada_block_source(const char* name, int core, unsigned in_key, std::ostream& log_) : out(log_)
{
    log = multilog_open(name, 0);                                  // :26
    multilog_add(static_cast<multilog_t*>(log), stderr);           // :27
    multilog(static_cast<multilog_t*>(log), LOG_INFO, "creating hdu\n");
    dada_hdu_t* hdu = dada_hdu_create(static_cast<multilog_t*>(log));   // :30
    dada_hdu_set_key(hdu, (key_t)in_key);                          // :31
    if (dada_hdu_connect(hdu) < 0) {                               // :33-36 (the reference exits; here ok() turns false)
        out << "Error: could not connect to dada buffer" << std::endl;
        dada_hdu_destroy(hdu);
        return;
    }
    if (dada_hdu_lock_read(hdu) < 0) {                             // :39-42
        out << "Error: could not lock to dada buffer (try relaxing memlock limits in /etc/security/limits.conf)" << std::endl;
        dada_hdu_disconnect(hdu);
        dada_hdu_destroy(hdu);
        return;
    }
    hdu_in = hdu;
    if (dbregister() < 0) {                                        // :44-47 (the reference exits; the copies still work
        out << "Error: could not pin dada buffer" << std::endl;    //  from pageable memory, so this build carries on unpinned)
    }
    if (core >= 0) {                                               // :50-56 dada_bind_thread_to_core
        out << "binding to core " << core << std::endl;
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(core, &set);
        if (::sched_setaffinity(0, sizeof(set), &set) != 0) out << "failed to bind to core " << core << std::endl;
    }
}

// :62-64 (+ the unlock / destroy of dsaX_dbgpu_cleanup, :118-124, which the reference only runs on its error paths)
dada_block_source::~dada_block_source()
{
    if (!hdu_in) return;
    dbunregister();
    cleanup();
}

void dada_block_source::cleanup()                                  // :118-124
{
    dada_hdu_t* hdu = static_cast<dada_hdu_t*>(hdu_in);
    if (dada_hdu_unlock_read(hdu) < 0) multilog(static_cast<multilog_t*>(log), LOG_ERR, "could not unlock read on hdu_in\n");
    dada_hdu_destroy(hdu);
    hdu_in = nullptr;
}

void dada_block_source::read_headers()                             // :66-90
{
    dada_hdu_t* hdu = static_cast<dada_hdu_t*>(hdu_in);
    char* header_in = ipcbuf_get_next_read(hdu->header_block, &header_size);
    if (!header_in) {
        multilog(static_cast<multilog_t*>(log), LOG_ERR, "main: could not read next header\n");
        failed = true;                                             // the reference: cleanup + exit(-1)
        return;
    }
    if (ipcbuf_mark_cleared(hdu->header_block) < 0) {
        multilog(static_cast<multilog_t*>(log), LOG_ERR, "could not mark header block cleared\n");
        failed = true;
        return;
    }
    block_size = ipcbuf_get_bufsz((ipcbuf_t*)hdu->data_block);     // size of block in dada buffer
    out << "block size is: " << block_size << std::endl;
}

char* dada_block_source::read()                                    // :92-94
{
    return ipcio_open_block_read(static_cast<dada_hdu_t*>(hdu_in)->data_block, &bytes_read, &block_id);
}

void dada_block_source::close()                                    // :96-98
{
    ipcio_close_block_read(static_cast<dada_hdu_t*>(hdu_in)->data_block, bytes_read);
}

bool dada_block_source::check_transfers_complete()                 // :100-116
{
    if (expected_bytes && bytes_read != expected_bytes)
        out << "ERROR: Async, Bytes Read: " << bytes_read << ", Should also be " << expected_bytes << std::endl;
    return bytes_read < block_size;                                // short block: the observation ends
}

// Page-locking of the ring's data blocks for DMA (the reference's dada_cuda_dbregister / dada_cuda_dbunregister,
// src/dada_handler.hh:127-177, with hipHostRegister behind bf_host_register): one walk over the blocks serves both directions.
// A ring whose blocks already live in device memory (ipcbuf_get_device >= 0) is left alone, as in the reference.
static int for_each_ring_block(void* hdu_void, bool pin, uint64_t* done)
{
    ipcbuf_t* ring = reinterpret_cast<ipcbuf_t*>(static_cast<dada_hdu_t*>(hdu_void)->data_block);
    if (ipcbuf_get_device(ring) >= 0) return 0;
    const uint64_t n_blocks = ring->sync->nbufs;
    const size_t block_bytes = ring->sync->bufsz;
    for (uint64_t b = 0; b < n_blocks; b++) {
        void* block = ring->buffer[b];
        const int rc = pin ? bf_host_register(block, block_bytes) : bf_host_unregister(block);
        if (rc != BF_OK) return -1;
        if (done) *done = b + 1;
    }
    return 0;
}

int dada_block_source::dbregister()
{
    ipcbuf_t* ring = reinterpret_cast<ipcbuf_t*>(static_cast<dada_hdu_t*>(hdu_in)->data_block);
    if (ipcbuf_lock(ring) < 0) {                                   // the blocks must be locked in shared memory first (:131-134)
        perror("dada_dbregister: ipcbuf_lock failed\n");
        return -1;
    }
    uint64_t pinned = 0;
    if (for_each_ring_block(hdu_in, /*pin=*/true, &pinned) != 0) {
        fprintf(stderr, "dada_dbregister: hipHostRegister failed: %s\n", bf_last_error());
        for (uint64_t b = 0; b < pinned; b++) bf_host_unregister(ring->buffer[b]);   // leave nothing half-registered
        return -1;
    }
    registered = true;
    return 0;
}

int dada_block_source::dbunregister()
{
    if (!registered) return 0;
    registered = false;
    if (for_each_ring_block(hdu_in, /*pin=*/false, nullptr) != 0) {
        fprintf(stderr, "dada_dbunregister: hipHostUnregister failed: %s\n", bf_last_error());
        return -1;
    }
    return 0;
}

}  // namespace dsabf
#endif  // DSABF_WITH_PSRDADA
