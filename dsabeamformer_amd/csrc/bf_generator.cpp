// bf_generator.cpp -- host mirror, part 2: test_data_generator (src/test_data_generator.hh:11-108) and
// the in-memory junk block source (the dada_junkdb stand-in of the production loop).
//
// Floating-point fidelity: the reference's trig expressions are written with unqualified sin/cos/round on float
// operands, which under g++ resolve to the double C functions.  Every promotion is spelled out
// below so the bytes match the reference's CPU path exactly; this file is compiled with -ffp-contract=off.
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
#include <unistd.h>

#include "../../include/dsabf_host.h"
#include "bf_host_internal.h"

namespace dsabf {

// ---- test_data_generator -------------------------------------------------------------------------------------
test_data_generator::test_data_generator(const bf_config& c, int per_batch, bool pin)
    : cfg(c), n_sources_per_batch(per_batch)
{
    const size_t n = input_data_size();
    void* p = nullptr;
    if (pin && bf_alloc_pinned(&p, n) == BF_OK) {  // cudaHostAlloc, src/test_data_generator.hh:35
        pinned = true;
    } else {
        p = ::malloc(n);
        pinned = false;
    }
    data = static_cast<char*>(p);
    if (data) ::memset(data, kBogusData, n);  // :36
}

test_data_generator::~test_data_generator()
{
    if (!data) return;
    if (pinned)
        bf_free_pinned(data);
    else
        ::free(data);
}

size_t test_data_generator::input_data_size() const { return bf_bytes_per_gemm(&cfg) * (size_t)n_sources_per_batch; }

void test_data_generator::set_source_directions(const beam_direction* src, int n)
{
    if (!use_source_catalog) {
        n_pt_sources = n;
        sources.assign(src, src + n);
        use_source_catalog = true;
        n_source_batches = (n_pt_sources + n_sources_per_batch - 1) / n_sources_per_batch;  // CEILING, :54
    }
}

int test_data_generator::read_in_source_directions(const char* file_name)
{
    if (use_source_catalog) return 0;
    std::ifstream input_file;
    input_file.open(file_name);
    if (!input_file.is_open()) return -1;
    int n = 0;
    input_file >> n;
    if (n < 0) n = 0;
    std::vector<beam_direction> s((size_t)n);
    for (int beam_idx = 0; beam_idx < n; beam_idx++) input_file >> s[beam_idx];
    set_source_directions(s.data(), n);
    if (cfg.verbose) std::cout << "Read in " << n_pt_sources << " source directions" << std::endl;
    return 0;
}

void test_data_generator::generate_test_data(const antenna pos[], int gpu)
{
    const int na = cfg.n_ant, nf = cfg.n_freq, nt = bf_n_timesteps_per_gemm(&cfg);
    const size_t per_gemm = bf_bytes_per_gemm(&cfg);
    parallel_for(n_sources_per_batch, [&](long lo, long hi) {
        for (long direction = lo; direction < hi; direction++) {
            const int source_look_up = (int)direction + source_batch_counter * n_sources_per_batch;  // :77
            for (int i = 0; i < nf; i++) {
                float freq = channel_frequency_generator(gpu, i);  // :72
                float wavelength = kCSpeed / (1E9 * freq);         // :74
                char* slab = data + (size_t)direction * per_gemm + (size_t)i * nt * na;
                for (int k = 0; k < na; k++) {
                    char byte = 0;  // :85
                    if (source_look_up < n_pt_sources && source_look_up < (int)sources.size()) {  // (no catalogue: zero bytes)
                        const double proj = (double)pos[k].x * ::sin((double)sources[source_look_up].theta) +
                                            (double)pos[k].y * ::sin((double)sources[source_look_up].phi);
                        const char high = (char)::round(kSigMaxVal * ::cos(2 * kPi * proj / (double)wavelength));  // :80
                        const char low = (char)::round(kSigMaxVal * ::sin(2 * kPi * proj / (double)wavelength));   // :81
                        byte = (char)(((int)high * 16) | (0x0F & (int)low));                                      // :83
                    }
                    slab[k] = byte;
                }
                // the reference evaluates the same expression for every time column j (it has no j in it):
                // replicate column 0 -- identical bytes, 1/n_time of the trig calls
                for (int j = 1; j < nt; j++) ::memcpy(slab + (size_t)j * na, slab, (size_t)na);
            }
        }
    });
    source_batch_counter++;  // :94
}

bool test_data_generator::check_need_to_generate_more_input_data(int blocks_transfered)
{
    return (use_source_catalog &&
            (blocks_transfered == (source_batch_counter * n_sources_per_batch) / cfg.n_gemms_per_block));  // :100
}

bool test_data_generator::check_data_ready_for_transfer(int blocks_transfer_queue)
{
    if (!use_source_catalog && (source_batch_counter == 0)) source_batch_counter = 1;             // :104-106
    return (blocks_transfer_queue < (source_batch_counter * n_sources_per_batch) / cfg.n_gemms_per_block);  // :107
}

// ---- junk_block_source -------------------------------------------------------------------------------------------
junk_block_source::junk_block_source(const bf_config& c, uint64_t nb, int rb, uint64_t seed)
    : block_size(bf_bytes_per_block(&c)), n_blocks(nb), ring_blocks(rb < 1 ? 1 : rb)
{
    const size_t total = (size_t)block_size * ring_blocks;
    void* p = nullptr;
    if (bf_alloc_pinned(&p, total) == BF_OK) {  // dada_cuda_dbregister pins the shm blocks, src/dada_handler.hh:127-177
        pinned = true;
    } else {
        p = ::malloc(total);
    }
    ring = static_cast<char*>(p);
    if (!ring) return;
    junk_fill(c, ring_blocks, seed, ring);
}

void junk_fill(const bf_config& c, int ring_blocks, uint64_t seed, char* ring)
{
    // every byte value (all 16 nibble codes in both halves), distinct blocks: 64-bit xorshift* per 8 bytes
    const size_t total = (size_t)bf_bytes_per_block(&c) * ring_blocks;
    const long parts = (long)ring_blocks * 64;
    const size_t n8 = total / 8 / (size_t)parts;       // 8-byte words per part; the LAST part also takes the remainder
    parallel_for(parts, [&](long lo, long hi) {
        for (long part = lo; part < hi; part++) {
            uint64_t x = seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(part + 1) * 0xBF58476D1CE4E5B9ULL;
            const size_t first = (size_t)part * n8, words = part == parts - 1 ? total / 8 - first : n8;
            uint64_t* q = reinterpret_cast<uint64_t*>(ring) + first;
            for (size_t i = 0; i < words; i++) {
                x ^= x >> 12;
                x ^= x << 25;
                x ^= x >> 27;
                q[i] = x * 0x2545F4914F6CDD1DULL;
            }
            if (part == parts - 1)                      // and the < 8 trailing bytes of a block size that is no multiple of 8
                for (size_t b = total / 8 * 8; b < total; b++) ring[b] = (char)(x >> (8 * (b & 7)));
        }
    });
}

junk_block_source::~junk_block_source()
{
    if (!ring) return;
    if (pinned)
        bf_free_pinned(ring);
    else
        ::free(ring);
}

char* junk_block_source::read()
{
    if (served < n_blocks) {
        bytes_read = block_size;
        return ring + (size_t)(served++ % (uint64_t)ring_blocks) * block_size;
    }
    bytes_read = 0;  // short block: end of data
    return ring;
}

bool junk_block_source::check_transfers_complete() { return bytes_read < block_size; }  // src/dada_handler.hh:105-113

}  // namespace dsabf
