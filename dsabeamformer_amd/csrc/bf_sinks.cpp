// bf_sinks.cpp -- host mirror, part 4: consumers of the detected stream: detected_sink and its file /
// shared-memory-ring implementations.
#include "../../include/dsabf_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <thread>
#include <fcntl.h>
#include <unistd.h>

#include "../../include/dsabf_host.h"
#include "bf_host_internal.h"

namespace dsabf {

// ---- detected-stream sinks ------------------------------------------------------------------------------------------
detected_sink::detected_sink(const bf_config& cfg, uint64_t slots, bool asynchronous)
    : floats_per_gemm(bf_floats_per_detect(&cfg)), n_slots(slots ? slots : slots_for(cfg)), async(asynchronous)
{
    const size_t bytes = floats_per_gemm * n_slots * sizeof(float);
    void* p = nullptr;
    if (bf_alloc_pinned(&p, bytes) == BF_OK)
        pinned = true;
    else
        p = ::malloc(bytes);  // no device (CPU tests of the ring logic)
    ring = static_cast<float*>(p);
    if (ring && async) worker = std::thread([this] { run(); });
}

detected_sink::~detected_sink()
{
    drain_and_stop();   // (a derived destructor has done this already; here it is a no-op)
    if (!ring) return;
    if (pinned)
        bf_free_pinned(ring);
    else
        ::free(ring);
}

// The delivery thread: hands committed gemm-units to deliver() in index order, one at a time.
void detected_sink::run()
{
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
        work.wait(lk, [this] { return stop || delivered < next_commit; });
        if (delivered == next_commit) return;   // stop requested and nothing left
        const uint64_t g = delivered;
        const uint64_t k = std::min(next_commit - delivered, n_slots - g % n_slots);   // ready AND contiguous in the ring
        lk.unlock();
        const bool ok_ = deliver_many(g, ring + (size_t)(g % n_slots) * floats_per_gemm, floats_per_gemm, k);
        lk.lock();
        if (!ok_) failed = true;
        delivered += k;
        done.notify_all();
    }
}

bool detected_sink::deliver_many(uint64_t first_gemm, const float* data, size_t n_floats_each, uint64_t count)
{
    bool ok_ = true;
    for (uint64_t i = 0; i < count; i++) ok_ = deliver(first_gemm + i, data + (size_t)i * n_floats_each, n_floats_each) && ok_;
    return ok_;
}

void detected_sink::drain_and_stop()
{
    if (!worker.joinable()) return;
    {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
    }
    work.notify_all();
    worker.join();   // run() returns only when everything committed has been delivered
}

bool detected_sink::ok()
{
    std::lock_guard<std::mutex> lk(m);
    return ring != nullptr && !failed;
}

uint64_t detected_sink::get_delivered()
{
    std::lock_guard<std::mutex> lk(m);
    return delivered;
}

float* detected_sink::acquire(uint64_t gemm_index)
{
    std::unique_lock<std::mutex> lk(m);
    if (!ring || gemm_index < next_commit || gemm_index >= next_commit + n_slots) return nullptr;
    // the slot's last occupant (gemm_index - n_slots) was committed; an asynchronous sink may still be delivering it
    if (async) done.wait(lk, [&] { return gemm_index < delivered + n_slots; });
    return ring + (size_t)(gemm_index % n_slots) * floats_per_gemm;
}

bool detected_sink::commit(uint64_t gemm_index)
{
    std::unique_lock<std::mutex> lk(m);
    if (!ring || gemm_index != next_commit) return false;
    next_commit++;
    if (async) {
        const bool was_ok = !failed;
        lk.unlock();
        work.notify_one();
        return was_ok;
    }
    lk.unlock();   // synchronous: deliver here, on the caller's thread
    const bool ok_ = deliver(gemm_index, ring + (size_t)(gemm_index % n_slots) * floats_per_gemm, floats_per_gemm);
    lk.lock();
    if (!ok_) failed = true;
    delivered++;
    return !failed;
}

file_sink::file_sink(const bf_config& cfg, const char* path, int gpu, uint64_t slots) : detected_sink(cfg, slots, true)
{
    fd = ::open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) return;
    char header[kHeaderBytes];
    ::memset(header, 0, sizeof(header));
    ::snprintf(header, sizeof(header),
               "HDR_VERSION 1.0\nHDR_SIZE %zu\nINSTRUMENT DSA\nCONTENT detected_power\nDTYPE float32\nENDIAN little\n"
               "ORDER gemm,output,frequency,beam\nN_BEAMS %d\nN_FREQUENCIES %d\nN_OUTPUTS_PER_GEMM %d\nN_ANTENNAS %d\n"
               "N_POL %d\nN_AVERAGING %d\nN_GEMMS_PER_BLOCK %d\nGPU %d\nFLOATS_PER_GEMM %zu\n",
               kHeaderBytes, cfg.n_beams, cfg.n_freq, cfg.n_out_per_gemm, cfg.n_ant, cfg.n_pol, cfg.n_avg,
               cfg.n_gemms_per_block, gpu, get_floats_per_gemm());
    if (::pwrite(fd, header, sizeof(header), 0) != (ssize_t)sizeof(header)) {
        ::close(fd);
        fd = -1;
    }
    if (const char* e = lab_getenv("DSABF_SINK_THREADS")) write_threads = std::max(1, atoi(e));
}

file_sink::~file_sink()
{
    drain_and_stop();
    finish();
}

// every gemm-unit has its place in the file (header + index * size): positional writes, in any order, from any thread
static bool pwrite_all(int fd, const char* p, size_t n, off_t at)
{
    while (n) {
        const ssize_t w = ::pwrite(fd, p, n, at);
        if (w <= 0) return false;
        p += w;
        n -= (size_t)w;
        at += w;
    }
    return true;
}

bool file_sink::deliver(uint64_t gemm_index, const float* data, size_t n_floats)
{
    return deliver_many(gemm_index, data, n_floats, 1);
}

bool file_sink::deliver_many(uint64_t first_gemm, const float* data, size_t n_floats_each, uint64_t count)
{
    if (fd < 0) return false;
    const size_t bytes = (size_t)count * n_floats_each * sizeof(float);
    const off_t at = (off_t)kHeaderBytes + (off_t)first_gemm * (off_t)(n_floats_each * sizeof(float));
    const char* p = reinterpret_cast<const char*>(data);
    const int nt = (int)std::min<size_t>((size_t)write_threads, bytes >> 22);   // at least 4 MiB per thread
    if (nt <= 1) return pwrite_all(fd, p, bytes, at);
    const size_t slice = ((bytes / nt) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    std::vector<char> ok_((size_t)nt, 1);
    int t = 0;
    for (size_t off = 0; off < bytes; off += slice, t++)
        th.emplace_back([=, &ok_] { ok_[(size_t)t] = pwrite_all(fd, p + off, std::min(slice, bytes - off), at + (off_t)off); });
    for (auto& x : th) x.join();
    return std::all_of(ok_.begin(), ok_.end(), [](char c) { return c != 0; });
}

void file_sink::finish()
{
    if (fd >= 0) ::close(fd);
    fd = -1;
}

ring_sink::ring_sink(const bf_config& cfg, const char* ring_name, uint64_t ring_blocks, int gpu, uint64_t slots)
    : detected_sink(cfg, slots, true), name(ring_name ? ring_name : "")
{
    char header[kRingHeaderBytes];
    ::snprintf(header, sizeof(header),
               "HDR_VERSION 1.0\nHDR_SIZE %zu\nINSTRUMENT DSA\nCONTENT detected_power\nDTYPE float32\nENDIAN little\n"
               "ORDER output,frequency,beam\nN_BEAMS %d\nN_FREQUENCIES %d\nN_OUTPUTS_PER_GEMM %d\nGPU %d\n",
               kRingHeaderBytes, cfg.n_beams, cfg.n_freq, cfg.n_out_per_gemm, gpu);
    out = shm_ring::create(name.c_str(), ring_blocks, get_floats_per_gemm() * sizeof(float), header);
}

ring_sink::~ring_sink()
{
    drain_and_stop();
    finish();
}

bool ring_sink::deliver(uint64_t, const float* data, size_t n_floats)
{
    if (!out) return false;
    char* b = out->open_block_write();  // blocks while the consumer is behind by a whole ring
    if (!b) return false;
    ::memcpy(b, data, n_floats * sizeof(float));
    out->close_block_write(n_floats * sizeof(float));
    return true;
}

void ring_sink::finish()
{
    if (!out) return;
    if (out->open_block_write()) out->close_block_write(0);  // short block: end of data
    // wait for the consumer to drain, then remove the ring (dada_db -d)
    for (int waited = 0; out->get_blocks_read() < out->get_blocks_written() && waited < 10000; waited += 5) ::usleep(5000);
    delete out;
    out = nullptr;
    shm_ring::unlink(name.c_str());
}

// ---- DM-trial chunks ---------------------------------------------------------------------------------------------------
dm_file_sink::dm_file_sink(const bf_config& cfg, int n_freq_total, int n_dm, int max_delay, const char* path, int gpu, int first_trial)
{
    fd = ::open(path, O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (fd < 0) return;
    char header[kHeaderBytes];
    ::memset(header, 0, sizeof(header));
    ::snprintf(header, sizeof(header),
               "HDR_VERSION 1.0\nHDR_SIZE %zu\nINSTRUMENT DSA\nCONTENT dedispersed_power\nDTYPE float32\nENDIAN little\n"
               "ORDER chunk(dm,time,beam)\nRECORD_HEADER_BYTES %zu\nN_DM %d\nDM_FIRST_TRIAL %d\nN_BEAMS %d\nN_FREQUENCIES %d\nMAX_DELAY %d\n"
               "N_OUTPUTS_PER_GEMM %d\nN_GEMMS_PER_BLOCK %d\nN_AVERAGING %d\nGPU %d\n",
               kHeaderBytes, kRecordBytes, n_dm, first_trial, cfg.n_beams, n_freq_total, max_delay, cfg.n_out_per_gemm,
               cfg.n_gemms_per_block, cfg.n_avg, gpu);
    if (!pwrite_all(fd, header, sizeof(header), 0) || ::lseek(fd, (off_t)kHeaderBytes, SEEK_SET) < 0) {
        ::close(fd);
        fd = -1;
    }
}

dm_file_sink::~dm_file_sink() { close(); }

static bool write_all(int fd, const char* p, size_t n)
{
    while (n) {
        const ssize_t w = ::write(fd, p, n);
        if (w <= 0) return false;
        p += w;
        n -= (size_t)w;
    }
    return true;
}

bool dm_file_sink::deliver(uint64_t first_t, int n_t, int n_dm, int n_beams, const float* data)
{
    if (fd < 0 || n_t <= 0) return fd >= 0;
    if (first_t != written_t) return false;   // chunks follow each other without gaps
    char rec[kRecordBytes];
    ::memset(rec, 0, sizeof rec);
    const uint32_t v[3] = {(uint32_t)n_t, (uint32_t)n_dm, (uint32_t)n_beams};
    ::memcpy(rec, &first_t, 8);
    ::memcpy(rec + 8, v, 12);
    if (!write_all(fd, rec, sizeof rec) ||
        !write_all(fd, reinterpret_cast<const char*>(data), (size_t)n_dm * n_t * n_beams * sizeof(float)))
        return false;
    written_t += (uint64_t)n_t;
    chunks++;
    return true;
}

void dm_file_sink::close()
{
    if (fd >= 0) ::close(fd);
    fd = -1;
}

dm_ring_sink::dm_ring_sink(const bf_config& cfg, int n_freq_total, int n_dm, int max_delay, int max_rows, const char* ring_name,
                           uint64_t ring_blocks, int gpu, int first_trial)
    : name(ring_name ? ring_name : "")
{
    block_bytes = dm_file_sink::kRecordBytes + (size_t)n_dm * max_rows * cfg.n_beams * sizeof(float);
    char header[kRingHeaderBytes];
    ::snprintf(header, sizeof(header),
               "HDR_VERSION 1.0\nHDR_SIZE %zu\nINSTRUMENT DSA\nCONTENT dedispersed_power\nDTYPE float32\nENDIAN little\n"
               "ORDER chunk(dm,time,beam)\nRECORD_HEADER_BYTES %zu\nN_DM %d\nDM_FIRST_TRIAL %d\nN_BEAMS %d\nN_FREQUENCIES %d\nMAX_DELAY %d\n"
               "MAX_TIMES_PER_CHUNK %d\nGPU %d\n",
               kRingHeaderBytes, dm_file_sink::kRecordBytes, n_dm, first_trial, cfg.n_beams, n_freq_total, max_delay, max_rows, gpu);
    out = shm_ring::create(name.c_str(), ring_blocks, block_bytes, header);
}

dm_ring_sink::~dm_ring_sink() { close(); }

bool dm_ring_sink::deliver(uint64_t first_t, int n_t, int n_dm, int n_beams, const float* data)
{
    if (!out) return false;
    const size_t payload = (size_t)n_dm * n_t * n_beams * sizeof(float);
    if (n_t <= 0 || dm_file_sink::kRecordBytes + payload > block_bytes) return n_t <= 0;
    char* b = out->open_block_write();   // blocks while the consumer is behind by a whole ring
    if (!b) return false;
    ::memset(b, 0, dm_file_sink::kRecordBytes);
    const uint32_t v[3] = {(uint32_t)n_t, (uint32_t)n_dm, (uint32_t)n_beams};
    ::memcpy(b, &first_t, 8);
    ::memcpy(b + 8, v, 12);
    ::memcpy(b + dm_file_sink::kRecordBytes, data, payload);
    out->close_block_write(block_bytes);   // always a whole block: a short one means end of data
    chunks++;
    return true;
}

void dm_ring_sink::close()
{
    if (!out) return;
    if (out->open_block_write()) out->close_block_write(0);   // short block: end of data
    for (int waited = 0; out->get_blocks_read() < out->get_blocks_written() && waited < 10000; waited += 5) ::usleep(5000);
    delete out;
    out = nullptr;
    shm_ring::unlink(name.c_str());
}

}  // namespace dsabf
