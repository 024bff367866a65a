// junkdb_main.cpp -- the writer side of the shared-memory input ring: what `dada_db` + `dada_junkdb` do for the
// reference (makefile:28-33, README.md:151-175): create the ring, fill it with n blocks of pseudo-random 4-bit
// voltages at an optional rate, write a short block (end of data), wait until the reader has drained it, delete it.
//
//   junkdb -k name [-n blocks] [-r ring_blocks] [-b block_bytes] [-s seed] [-d distinct] [-R MB/s] [-H header_file] [-T copy_threads]
//
// Block i carries the bytes of block (i % distinct) of dsabf::junk_fill(seed), so a test can recompute every block.
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dsabf_host.hpp"

int main(int argc, char* argv[])
{
    using namespace dsabf;
    bf_config cfg;
    bf_config_default(&cfg, /*debug=*/0);
    std::string name = "dsabf", header_file;
    long n_blocks = 8, ring_blocks = 4, distinct = 4;
    uint64_t seed = 0xD5A, block_bytes = 0;
    int copy_threads = 4;  // threads that copy one block into the ring (1 = the single memcpy of dada_junkdb)
    double rate_mbs = 0;  // 0 = as fast as the reader takes them (the instrument: 4050 MB/s, makefile:29)
    int arg;
    while ((arg = getopt(argc, argv, "k:n:r:b:s:d:R:H:T:a:h")) != -1) {
        switch (arg) {
            case 'k': name = optarg; break;
            case 'n': n_blocks = atol(optarg); break;
            case 'r': ring_blocks = atol(optarg); break;
            case 'b': block_bytes = strtoull(optarg, nullptr, 0); break;
            case 's': seed = strtoull(optarg, nullptr, 0); break;
            case 'd': distinct = atol(optarg); break;
            case 'R': rate_mbs = atof(optarg); break;
            case 'H': header_file = optarg; break;
            case 'T': copy_threads = atoi(optarg); break;
            default:
                std::cout << "junkdb -k name [-n blocks] [-r ring_blocks] [-b block_bytes] [-s seed] [-d distinct] "
                             "[-R MB/s] [-H header_file] [-T copy_threads]\n";
                return arg == 'h' ? 0 : 1;
        }
    }
    if (distinct < 1) distinct = 1;
    if (!block_bytes) block_bytes = bf_bytes_per_block(&cfg);
    // junk_fill works on cfg's block size: describe block_bytes as gemm-units of 1 byte each
    bf_config fill_cfg = cfg;
    if (block_bytes != bf_bytes_per_block(&cfg)) {
        fill_cfg.n_gemms_per_block = 1;
        fill_cfg.n_freq = 1;
        fill_cfg.n_ant = 1;
        fill_cfg.n_pol = 1;
        fill_cfg.n_avg = 1;
        fill_cfg.n_out_per_gemm = (int)block_bytes;
        if (bf_bytes_per_block(&fill_cfg) != block_bytes) {
            fprintf(stderr, "junkdb: unsupported block size\n");
            return 1;
        }
    }
    std::string header = "HDR_VERSION 1.0\nHDR_SIZE 4096\nINSTRUMENT DSAX\nMODE RAW\nNBIT 4\nNPOL 2\nSOURCE JUNK\n";
    if (!header_file.empty()) {
        std::ifstream in(header_file);
        if (!in) {
            fprintf(stderr, "junkdb: cannot read %s\n", header_file.c_str());
            return 1;
        }
        std::stringstream ss;
        ss << in.rdbuf();
        header = ss.str();
    }
    std::vector<char> junk((size_t)block_bytes * distinct);
    junk_fill(fill_cfg, (int)distinct, seed, junk.data());

    shm_ring* ring = shm_ring::create(name.c_str(), (uint64_t)ring_blocks, block_bytes, header.c_str());
    if (!ring) {
        fprintf(stderr, "junkdb: could not create ring %s\n", name.c_str());
        return 1;
    }
    std::cout << "junkdb: ring " << name << ": " << ring_blocks << " blocks of " << block_bytes << " bytes" << std::endl;
    const auto t0 = std::chrono::steady_clock::now();
    for (long i = 0; i < n_blocks; i++) {
        char* b = ring->open_block_write();
        if (!b) return 1;
        const char* src = junk.data() + (size_t)(i % distinct) * block_bytes;
        if (copy_threads <= 1) {
            std::memcpy(b, src, block_bytes);
        } else {   // one core's memcpy (~30 GB/s) is slower than the reader's H2D (~55 GB/s): split the block
            std::vector<std::thread> th;
            const int nth = copy_threads > 64 ? 64 : copy_threads;      // -T is clamped to [1, 64]
            size_t slice = ((size_t)block_bytes / nth + 4095) & ~(size_t)4095;
            if (slice < 4096) slice = 4096;                              // never 0: tiny blocks are one slice
            for (size_t off = 0; off < block_bytes; off += slice)
                th.emplace_back([=] { std::memcpy(b + off, src + off, std::min(slice, (size_t)block_bytes - off)); });
            for (auto& t : th) t.join();
        }
        if (rate_mbs > 0) {
            const double due = (double)(i + 1) * block_bytes / (rate_mbs * 1e6);
            std::this_thread::sleep_until(t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(
                                                   std::chrono::duration<double>(due)));
        }
        ring->close_block_write(block_bytes);
    }
    ring->open_block_write();
    ring->close_block_write(0);  // short block: end of data (src/dada_handler.hh:105-113)
    const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "junkdb: wrote " << n_blocks << " blocks in " << el << " s (" << n_blocks * (double)block_bytes / el / 1e6
              << " MB/s)" << std::endl;
    while (ring->get_blocks_read() < (uint64_t)n_blocks + 1) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    delete ring;
    shm_ring::unlink(name.c_str());
    return 0;
}
