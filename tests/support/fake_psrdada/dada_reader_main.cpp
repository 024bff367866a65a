// dada_reader_main.cpp -- TEST INFRASTRUCTURE ONLY: drives dsabf::dada_block_source (csrc/bf_dada.cpp, the reference's dada_handler
// behind -DDSABF_WITH_PSRDADA) through exactly the calls the observation loop makes (src/beamformer.cu:334,384-401) against the
// stand-in library of this directory, and prints what it saw.  No GPU is needed: without one the page-locking of the ring blocks
// fails and the adapter carries on unpinned (as its constructor says).
//   usage: dada_reader <hex key> <expected block bytes>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "dsabf_host.hpp"

int main(int argc, char** argv)
{
    if (argc < 3) return 64;
    const unsigned key = (unsigned)strtoul(argv[1], nullptr, 16);
    const uint64_t expect = strtoull(argv[2], nullptr, 10);
    dsabf::dada_block_source src("dada_reader", -1, key, std::cout);   // src/dada_handler.hh:25-60
    if (!src.ok()) return 2;
    src.expect_block_bytes(expect);
    src.read_headers();                                                  // :66-90
    if (!src.ok()) return 3;
    std::cout << "pinned " << (src.is_pinned() ? 1 : 0) << " block_size " << src.get_block_size() << std::endl;
    for (int n = 0;; n++) {
        const char* block = src.read();                                  // :92-94
        if (!block) return 4;
        const bool done = src.check_transfers_complete();                // :100-116: a short block ends the observation
        uint64_t h = 1469598103934665603ull;                             // FNV-1a of the valid bytes
        for (uint64_t i = 0; i < src.get_bytes_read(); i++) h = (h ^ (unsigned char)block[i]) * 1099511628211ull;
        std::cout << "block " << n << " bytes " << src.get_bytes_read() << " fnv " << std::hex << h << std::dec << (done ? " last" : "") << std::endl;
        src.close();                                                     // :96-98
        if (done) break;
    }
    std::cout << "done" << std::endl;
    return 0;
}
