// bf_shmring.cpp -- a POSIX shared-memory block ring with the subset of PSRDADA's ipcbuf / ipcio semantics the
// reference's dada_handler relies on (src/dada_handler.hh:25-116), and the block_source on top of it.
//
// libpsrdada is not in this build, so this is the stand-in for the instrument's input ring:
// a writer process (csrc/junkdb_main.cpp, the reference's `dada_junkdb`, makefile:28-29) fills fixed-size blocks, the
// beamformer process attaches, pins the blocks for DMA (dada_cuda_dbregister, src/dada_handler.hh:127-158) and
// consumes them through read()/close(); a block shorter than the block size ends the observation (:100-116).
// A libpsrdada build replaces shm_block_source by the four ipcio_*/ipcbuf_* calls and keeps everything else.
#include <fcntl.h>
#include <semaphore.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <new>
#include <string>

#include "../../include/dsabf_host.hpp"

namespace dsabf {

namespace {
constexpr uint64_t kMagic = 0x44534142465247ULL;  // "DSABFRG"
constexpr size_t kAlign = 4096;

std::string shm_name(const char* name)
{
    std::string s = name && name[0] == '/' ? name : std::string("/") + (name ? name : "");
    return s;
}
}  // namespace

struct shm_ring::control {
    uint64_t magic;
    uint64_t n_blocks, block_size, header_size;
    uint64_t data_offset;              // from the start of the mapping
    uint64_t written, read;            // blocks closed by the writer / by the reader
    sem_t full, empty;                 // filled blocks waiting / free blocks (process-shared)
    uint64_t bytes[kMaxRingBlocks];    // valid bytes of each slot
    char header[kRingHeaderBytes];     // ASCII header (PSRDADA style), NUL padded
};

shm_ring::shm_ring() {}

shm_ring::~shm_ring()
{
    if (ctl) ::munmap(ctl, map_bytes);
}

shm_ring* shm_ring::create(const char* name, uint64_t n_blocks, uint64_t block_size, const char* header_text)
{
    if (!name || n_blocks == 0 || n_blocks > kMaxRingBlocks || block_size == 0) return nullptr;
    const std::string nm = shm_name(name);
    ::shm_unlink(nm.c_str());  // a stale ring of the same name (dada_db -d)
    const int fd = ::shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return nullptr;
    const size_t ctl_bytes = (sizeof(control) + kAlign - 1) / kAlign * kAlign;
    const size_t total = ctl_bytes + (size_t)n_blocks * block_size;
    if (::ftruncate(fd, (off_t)total) != 0) {
        ::close(fd);
        ::shm_unlink(nm.c_str());
        return nullptr;
    }
    void* p = ::mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) {
        ::shm_unlink(nm.c_str());
        return nullptr;
    }
    shm_ring* r = new (std::nothrow) shm_ring();
    if (!r) {
        ::munmap(p, total);
        return nullptr;
    }
    r->ctl = static_cast<control*>(p);
    r->map_bytes = total;
    control* c = r->ctl;
    ::memset(c, 0, sizeof(control));
    c->n_blocks = n_blocks;
    c->block_size = block_size;
    c->header_size = kRingHeaderBytes;
    c->data_offset = ctl_bytes;
    if (header_text) ::strncpy(c->header, header_text, kRingHeaderBytes - 1);
    ::sem_init(&c->full, 1, 0);
    ::sem_init(&c->empty, 1, (unsigned)n_blocks);
    __atomic_store_n(&c->magic, kMagic, __ATOMIC_RELEASE);  // last: attachers wait for it
    return r;
}

shm_ring* shm_ring::attach(const char* name, int timeout_ms)
{
    const std::string nm = shm_name(name);
    int fd = -1;
    for (int waited = 0;; waited += 10) {  // the writer may still be creating it (dada_hdu_connect fails instead)
        fd = ::shm_open(nm.c_str(), O_RDWR, 0600);
        if (fd >= 0) {
            struct stat st;
            if (::fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(control)) break;
            ::close(fd);
            fd = -1;
        }
        if (waited >= timeout_ms) return nullptr;
        ::usleep(10000);
    }
    struct stat st;
    ::fstat(fd, &st);
    void* p = ::mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    ::close(fd);
    if (p == MAP_FAILED) return nullptr;
    control* c = static_cast<control*>(p);
    for (int waited = 0; __atomic_load_n(&c->magic, __ATOMIC_ACQUIRE) != kMagic; waited += 10) {
        if (waited >= timeout_ms) {
            ::munmap(p, (size_t)st.st_size);
            return nullptr;
        }
        ::usleep(10000);
    }
    shm_ring* r = new (std::nothrow) shm_ring();
    if (!r) {
        ::munmap(p, (size_t)st.st_size);
        return nullptr;
    }
    r->ctl = c;
    r->map_bytes = (size_t)st.st_size;
    return r;
}

int shm_ring::unlink(const char* name) { return ::shm_unlink(shm_name(name).c_str()); }

uint64_t shm_ring::get_n_blocks() const { return ctl->n_blocks; }
uint64_t shm_ring::get_block_size() const { return ctl->block_size; }
uint64_t shm_ring::get_header_size() const { return ctl->header_size; }
const char* shm_ring::get_header() const { return ctl->header; }
uint64_t shm_ring::get_blocks_written() const { return __atomic_load_n(&ctl->written, __ATOMIC_ACQUIRE); }
uint64_t shm_ring::get_blocks_read() const { return __atomic_load_n(&ctl->read, __ATOMIC_ACQUIRE); }
char* shm_ring::block(uint64_t slot) const
{
    return reinterpret_cast<char*>(ctl) + ctl->data_offset + (size_t)(slot % ctl->n_blocks) * ctl->block_size;
}

static int wait_sem(sem_t* s)
{
    int rc;
    while ((rc = ::sem_wait(s)) != 0 && errno == EINTR) {}
    return rc;
}

char* shm_ring::open_block_write()
{
    if (wait_sem(&ctl->empty) != 0) return nullptr;
    return block(ctl->written);
}

void shm_ring::close_block_write(uint64_t bytes)
{
    ctl->bytes[ctl->written % ctl->n_blocks] = bytes;
    __atomic_store_n(&ctl->written, ctl->written + 1, __ATOMIC_RELEASE);
    ::sem_post(&ctl->full);
}

char* shm_ring::open_block_read(uint64_t* bytes, uint64_t* block_id)  // ipcio_open_block_read
{
    if (wait_sem(&ctl->full) != 0) return nullptr;
    const uint64_t id = ctl->read;
    if (bytes) *bytes = ctl->bytes[id % ctl->n_blocks];
    if (block_id) *block_id = id;
    return block(id);
}

void shm_ring::close_block_read()  // ipcio_close_block_read
{
    __atomic_store_n(&ctl->read, ctl->read + 1, __ATOMIC_RELEASE);
    ::sem_post(&ctl->empty);
}

// ---- shm_block_source: dada_handler on top of the ring ----------------------------------------------------------------
shm_block_source::shm_block_source(const char* name, int core, bool pin, std::ostream& log_) : log(log_)
{
    ring = shm_ring::attach(name);  // dada_hdu_connect + lock_read, src/dada_handler.hh:32-43
    if (!ring) {
        log << "Error: could not connect to dada buffer" << std::endl;
        return;
    }
    if (pin) {  // dada_cuda_dbregister, :127-158: one registration per block
        registered = true;
        for (uint64_t i = 0; i < ring->get_n_blocks(); i++)
            if (bf_host_register(ring->block(i), ring->get_block_size()) != BF_OK) {
                for (uint64_t j = 0; j < i; j++) bf_host_unregister(ring->block(j));
                registered = false;
                // the reference exits here ("could not pin dada buffer"); the copies still work from pageable memory,
                // only slower, so this build says so and carries on
                log << "Warning: could not pin dada buffer (" << bf_last_error() << "); continuing unpinned" << std::endl;
                break;
            }
    }
    if (core >= 0) {  // dada_bind_thread_to_core, :51-56
        log << "binding to core " << core << std::endl;
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(core, &set);
        if (::sched_setaffinity(0, sizeof(set), &set) != 0) log << "failed to bind to core " << core << std::endl;
    }
}

shm_block_source::~shm_block_source()
{
    if (ring && registered)  // dada_cuda_dbunregister, :160-177
        for (uint64_t i = 0; i < ring->get_n_blocks(); i++) bf_host_unregister(ring->block(i));
    delete ring;
}

void shm_block_source::read_headers()  // :66-90
{
    header_size = ring->get_header_size();
    block_size = ring->get_block_size();
    log << "block size is: " << block_size << std::endl;
}

char* shm_block_source::read() { return ring->open_block_read(&bytes_read, &block_id); }  // :92-94
void shm_block_source::close() { ring->close_block_read(); }                                // :96-98

bool shm_block_source::check_transfers_complete()  // :100-116
{
    if (expected_bytes && bytes_read != expected_bytes)
        log << "ERROR: Async, Bytes Read: " << bytes_read << ", Should also be " << expected_bytes << std::endl;
    return bytes_read < block_size;
}

}  // namespace dsabf
