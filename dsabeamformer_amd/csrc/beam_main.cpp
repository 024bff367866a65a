// beam_main.cpp -- the `beam` driver: command line and lifecycle of the reference's main() (src/beamformer.cu:12-157,
// 539-571; usage() src/beamformer.hh:222-243) on top of libdsabf.so.
//
//   beam [-g gpu] [-p position_file] [-d direction_file] [-s source_file] [-o data.py] [-D device] [-a n_avg] [-u] [-v] [-h | -H]
//        (-u: the reference's launch pattern, one launch + copy per gemm-unit, instead of one per block; same data.py)
//   beam -j n_blocks [-g gpu] [-p ...] [-d ...]     production geometry, observation loop fed by the in-memory
//                                                   dada_junkdb stand-in (soak / data-rate run, makefile:28-29)
//   beam -j n_blocks -R world -r rank -I idfile     one frequency SHARD of a sub-band: this process beamforms channels
//                                                   [rank * 256/world, (rank+1) * 256/world); after every block the
//                                                   shards' detected powers are gathered to rank 0 (RCCL over xGMI), which
//                                                   alone writes -w / -K.  idfile: rank 0 publishes the 128-byte RCCL
//                                                   unique id there, the others wait for it.  beam_replicas -S starts
//                                                   the `world` processes.  (The reference's own scaling -- 8 independent
//                                                   sub-bands selected by -g, README.md:168 -- is beam_replicas without -S.)
//   beam -j n_blocks -M dm_max [-N n_dm] [-T tsamp_ms] [-W dm_file | -Q dm_ring]
//                                                   the DM stage, where the reference's loop has its DM-0
//                                                   collapse (src/beamformer.cu:492-511): the ladder of sandbox/Dispersion
//                                                   Theory.ipynb from 0 to dm_max (at most n_dm of its trials, evenly picked),
//                                                   every analysed block through a bf_dm_stream with the delay window carried
//                                                   over on the device, chunks [dm][t][beam] to dm_file.  With -R: on the
//                                                   gather root, over the gathered band.  With -R and -X: the shards' powers go to
//                                                   EVERY rank (the one collective becomes an all-gather) and rank r dedisperses its
//                                                   share of the ladder, writing dm_file.<r>: the DM work scales with the GPUs.
//
// With the reference's `make debug` geometry (default) it generates synthetic point-source voltages on the CPU,
// streams them through the observation loop and writes bin/data.py (dedispersed beam responses, one row per source)
// exactly like the reference's DEBUG build.  The PSRDADA observation mode (-c core, -k key) needs libpsrdada, which is
// not part of this build; the options are accepted and reported.
#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <fstream>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "../../include/dsabf_host.hpp"

int main(int argc, char* argv[])
{
    using namespace dsabf;
    bf_config cfg;
    bf_config_default(&cfg, /*debug=*/1);
    debug_run_options opt;
    bool per_unit = false;
    std::string positions, directions, sources, output = "bin/data.py", detected_path, out_ring;
    std::string ring_key;
    int core = -1;
    long junk_blocks = -1;
    int world = 1, rank = 0;
    std::string id_file;
    double dm_max = 0.0, tsamp_ms = 0.131;   // (the notebook's sample time, cell 5)
    int n_dm_cap = 0;
    bool dm_split = false;
    std::string dm_path, dm_ring;   // -W file / -Q shared-memory ring for the DM chunks

    int arg = 0;
    while ((arg = getopt(argc, argv, "s:g:p:d:o:D:a:c:k:K:j:w:R:r:I:M:N:T:W:Q:XuvhH")) != -1) {  // src/beamformer.cu:41-43 (+ -o -D -a -v)
        switch (arg) {
            case 's': sources = optarg; break;                 // :77-89
            case 'g': opt.gpu = atoi(optarg); break;           // :92-100
            case 'p': positions = optarg; break;               // :102-111
            case 'd': directions = optarg; break;              // :113-121
            case 'o': output = optarg; break;
            case 'D': opt.device = atoi(optarg); break;
            case 'a': cfg.n_avg = atoi(optarg); break;
            case 'j': junk_blocks = atol(optarg); break;
            case 'w': detected_path = optarg; break;
            case 'K': out_ring = optarg; break;
            case 'R': world = atoi(optarg); break;
            case 'r': rank = atoi(optarg); break;
            case 'I': id_file = optarg; break;
            case 'M': dm_max = atof(optarg); break;
            case 'N': n_dm_cap = atoi(optarg); break;
            case 'T': tsamp_ms = atof(optarg); break;
            case 'W': dm_path = optarg; break;
            case 'X': dm_split = true; break;
            case 'Q': dm_ring = optarg; break;
            case 'u': per_unit = true; break;                   // the reference's launch pattern: one launch per gemm-unit
            case 'v': opt.verbose = true; cfg.verbose = 1; break;
            case 'c': core = atoi(optarg); break;              // :59-65
            case 'k': ring_key = optarg; break;                // :66-75 (a shared-memory ring name instead of a hex key)
            case 'h': usage(true, std::cout); return EXIT_SUCCESS;  // :123-125
            case 'H':   // the reference's text, then what this build adds to its command line
                usage(true, std::cout);
                std::cout << "extensions of this build (no counterpart in the reference):\n"
                             " -o file                 where the DEBUG run writes its table [bin/data.py]\n"
                             " -D device               HIP device index [0]\n"
                             " -a n_avg                N_AVERAGING of the DEBUG geometry [1]\n"
                             " -u                      the reference's launch pattern: one launch + copy per gemm-unit\n"
                             " -v                      verbose (the reference's -DVERBOSE)\n"
                             " -j n_blocks             observation mode: production geometry, in-memory dada_junkdb source\n"
                             " -k name | hexkey        observation mode: blocks from a shared-memory ring (hex key: PSRDADA builds)\n"
                             " -c core                 bind to a CPU core (with -k)\n"
                             " -w file | -K ring       keep the detected stream: file, or shared-memory ring to another process\n"
                             " -R world -r rank -I id  one frequency shard of a sub-band; powers gathered to rank 0 (RCCL)\n"
                             " -M dm_max [-N n_dm] [-T tsamp_ms]   the DM-trial stage inside the loop (notebook ladder 0 .. dm_max)\n"
                             " -W file | -Q ring       its chunks [dm][t][beam]: file of records, or shared-memory ring\n"
                             " -X                      with -R: gather to every shard, shard r dedisperses its share of the trials\n"
                             " -H                      this text\n";
                return EXIT_SUCCESS;
            default: usage(true, std::cerr); return EXIT_FAILURE;
        }
    }
    opt.positions = positions.empty() ? nullptr : positions.c_str();
    opt.directions = directions.empty() ? nullptr : directions.c_str();
    opt.sources = sources.empty() ? nullptr : sources.c_str();
    opt.output = output.c_str();
    opt.block_launch = !per_unit;

    int n_dev = 0;
    if (bf_device_count(&n_dev) != BF_OK || n_dev == 0) {
        fprintf(stderr, "GPUassert: %s\n", bf_last_error());  // src/beamformer.cuh:26
        return EXIT_FAILURE;
    }
    char name[256];
    if (bf_device_name(opt.device, name, sizeof name) == BF_OK) std::cout << "Selected: " << name << std::endl;

    if (junk_blocks >= 0 || !ring_key.empty()) {  // observation (production) mode: N_AVERAGING 16
        bf_config pcfg;
        bf_config_default(&pcfg, /*debug=*/0);
        pcfg.verbose = cfg.verbose;
        if (world < 1 || rank < 0 || rank >= world || pcfg.n_freq % world) {
            fprintf(stderr, "beam: -R %d -r %d: need 0 <= rank < world and world dividing %d channels\n", world, rank, pcfg.n_freq);
            return EXIT_FAILURE;
        }
        bf_config full_cfg = pcfg;        // the whole sub-band (what the gather root's sink receives)
        pcfg.n_freq /= world;             // this rank's shard
        bf_comm* comm = nullptr;
        if (world > 1 || !id_file.empty()) {
            char id[BF_COMM_ID_BYTES];
            if (id_file.empty()) {
                fprintf(stderr, "beam: -R needs -I idfile (where rank 0 publishes the RCCL unique id)\n");
                return EXIT_FAILURE;
            }
            if (rank == 0) {
                if (bf_comm_unique_id(id) != BF_OK) {
                    fprintf(stderr, "GPUassert: %s\n", bf_last_error());
                    return EXIT_FAILURE;
                }
                // A stale id (a crashed run, a reused name) would make the other ranks join a communicator that no longer
                // exists and hang: rank 0 removes whatever is there, writes a fresh file it alone created (O_EXCL |
                // O_NOFOLLOW: no symlink is followed) and renames it into place -- readers see nothing or the whole id.
                const std::string tmp = id_file + ".tmp";
                (void)unlink(id_file.c_str());
                (void)unlink(tmp.c_str());
                const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
                const bool ok = fd >= 0 && write(fd, id, sizeof id) == (ssize_t)sizeof id;
                if (fd >= 0) close(fd);
                if (!ok || rename(tmp.c_str(), id_file.c_str()) != 0) {
                    perror("beam: publishing the unique id");
                    return EXIT_FAILURE;
                }
            } else {
                bool got = false;
                for (int tries = 0; tries < 1200 && !got; tries++) {   // up to 2 minutes
                    std::ifstream in(id_file, std::ios::binary);
                    got = in && in.read(id, sizeof id) && in.gcount() == (std::streamsize)sizeof id;
                    if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
                }
                if (!got) {
                    fprintf(stderr, "beam: rank %d never saw the unique id in %s\n", rank, id_file.c_str());
                    return EXIT_FAILURE;
                }
            }
            if (bf_comm_create(rank, world, id, opt.device, &comm) != BF_OK) {
                fprintf(stderr, "GPUassert: %s\n", bf_last_error());
                return EXIT_FAILURE;
            }
            std::cout << "Shard " << rank << " of " << world << ": channels " << rank * pcfg.n_freq << " .. "
                      << (rank + 1) * pcfg.n_freq - 1 << std::endl;
        }
        std::vector<antenna> pos((size_t)pcfg.n_ant);
        std::vector<beam_direction> dir((size_t)pcfg.n_beams);
        if (!opt.positions || read_in_position_locations(opt.positions, pcfg.n_ant, pos.data()) != 0)
            default_positions(pcfg.n_ant, pos.data());
        if (!opt.directions || read_in_beam_directions(opt.directions, pcfg.n_beams, dir.data()) != 0)
            default_directions(pcfg.n_beams, dir.data());
        std::unique_ptr<block_source> src;
        observation_options oopt;
#ifdef DSABF_WITH_PSRDADA
        unsigned in_key = 0;
        char trailing = 0;
        if (!ring_key.empty() && sscanf(ring_key.c_str(), "%x%c", &in_key, &trailing) == 1) {   // -k baab: src/beamformer.cu:66-75
            char name[] = "beam";
            dada_block_source* s = new dada_block_source(name, core, in_key, std::cout);        // :132
            src.reset(s);
            if (!s->ok()) return EXIT_FAILURE;
            s->expect_block_bytes(bf_bytes_per_block(&pcfg));
            oopt.burn_in = kBurnIn;                                                              // BURNIN reads, :348-355
        } else
#endif
        if (!ring_key.empty()) {  // -k: blocks from the shared-memory ring (the PSRDADA stand-in), no burn-in reads
            shm_block_source* s = new shm_block_source(ring_key.c_str(), core, /*pin=*/true, std::cout);
            src.reset(s);
            if (!s->ok()) return EXIT_FAILURE;  // "Error: could not connect to dada buffer", src/dada_handler.hh:35-38
            s->expect_block_bytes(bf_bytes_per_block(&pcfg));
        } else {                  // -j: the in-memory dada_junkdb source
            junk_block_source* s = new junk_block_source(pcfg, (uint64_t)junk_blocks);
            src.reset(s);
            if (!s->ok()) {
                fprintf(stderr, "beam: could not allocate the junk ring\n");
                return EXIT_FAILURE;
            }
            oopt.burn_in = kBurnIn;
        }
        oopt.gpu = opt.gpu;
        oopt.device = opt.device;
        oopt.verbose = opt.verbose;
        oopt.world = world;
        oopt.rank = rank;
        oopt.comm = comm;
        const bf_config& sink_cfg = comm ? full_cfg : pcfg;
        if (comm && rank != 0) {          // only the gather root has a consumer
            out_ring.clear();
            detected_path.clear();
        }
        std::unique_ptr<detected_sink> sink;
        std::string sink_name;
        if (!out_ring.empty()) {  // -K: hand the detected stream to another process through a shared-memory ring
            ring_sink* rs = new ring_sink(sink_cfg, out_ring.c_str(), 8, opt.gpu);
            sink.reset(rs);
            sink_name = "ring " + out_ring;
            if (!rs->ok() || !rs->is_open()) {
                fprintf(stderr, "beam: could not create ring %s\n", out_ring.c_str());
                return EXIT_FAILURE;
            }
        } else if (!detected_path.empty()) {  // -w: keep the detected stream in a file (the reference drops it, README.md:149)
            file_sink* fs = new file_sink(sink_cfg, detected_path.c_str(), opt.gpu);
            sink.reset(fs);
            sink_name = detected_path;
            if (!fs->ok() || !fs->is_open()) {
                fprintf(stderr, "beam: could not open %s\n", detected_path.c_str());
                return EXIT_FAILURE;
            }
        }
        oopt.sink = sink.get();
        // -M: the DM stage.  Ladder and delay law of sandbox/Dispersion Theory.ipynb (cells 1-2 and 5) over the WHOLE sub-band this
        // run covers (world x n_freq channels), referred to its highest frequency (channel 0): every delay is >= 0.
        std::vector<int32_t> delays;
        std::unique_ptr<dm_file_sink> dm_sink;
        std::unique_ptr<dm_ring_sink> dm_rsink;
        int n_dm = 0, my_trials = 0;
        if (dm_max > 0.0) {
            std::vector<double> dms = dm_trials(0.0, dm_max);
            if (n_dm_cap > 0 && (int)dms.size() > n_dm_cap) {
                std::vector<double> pick;
                for (int i = 0; i < n_dm_cap; i++) pick.push_back(dms[(size_t)((double)i * (dms.size() - 1) / (n_dm_cap > 1 ? n_dm_cap - 1 : 1))]);
                dms.swap(pick);
            }
            n_dm = (int)dms.size();
            std::vector<float> freq((size_t)full_cfg.n_freq);
            for (int c = 0; c < full_cfg.n_freq; c++) freq[(size_t)c] = channel_frequency_weights(opt.gpu, c);
            delays.resize((size_t)n_dm * full_cfg.n_freq);
            dm_delays(dms.data(), n_dm, freq.data(), full_cfg.n_freq, freq[0], tsamp_ms, delays.data());
            int dmax = 0;
            for (int32_t d : delays) dmax = d > dmax ? d : dmax;
            std::cout << "DM stage: " << n_dm << " trials 0 .. " << dms.back() << " pc/cc, largest delay " << dmax << " samples of "
                      << tsamp_ms << " ms" << std::endl;
            oopt.dm_delays = delays.data();
            oopt.n_dm = n_dm;
            int my_first = 0, my_count = n_dm;
            if (dm_split && comm) {       // -X: every rank receives the band and takes its share of the trials
                oopt.gather_root = BF_GATHER_ROOT_ALL;
                oopt.dm_split_trials = true;
                dm_trial_share(n_dm, world, rank, &my_first, &my_count);
                dmax = 0;
                for (size_t i = (size_t)my_first * full_cfg.n_freq; i < (size_t)(my_first + my_count) * full_cfg.n_freq; i++) dmax = delays[i] > dmax ? delays[i] : dmax;
                if (!dm_path.empty()) dm_path += "." + std::to_string(rank);
                std::cout << "Shard " << rank << " dedisperses trials " << my_first << " .. " << my_first + my_count - 1 << std::endl;
            }
            my_trials = my_count;
            if (!dm_path.empty() && my_count > 0 && (!comm || rank == 0 || oopt.dm_split_trials)) {
                dm_sink.reset(new dm_file_sink(pcfg, full_cfg.n_freq, my_count, dmax, dm_path.c_str(), opt.gpu, my_first));
                if (!dm_sink->is_open()) {
                    fprintf(stderr, "beam: could not open %s\n", dm_path.c_str());
                    return EXIT_FAILURE;
                }
                oopt.dm_sink = dm_sink.get();
            } else if (!dm_ring.empty() && my_count > 0 && (!comm || rank == 0 || oopt.dm_split_trials)) {
                // -Q: hand the chunks to another process (the downstream search) through a shared-memory ring
                if (oopt.dm_split_trials) dm_ring += "." + std::to_string(rank);
                dm_rsink.reset(new dm_ring_sink(pcfg, full_cfg.n_freq, my_count, dmax, pcfg.n_gemms_per_block * pcfg.n_out_per_gemm,
                                                dm_ring.c_str(), 4, opt.gpu, my_first));
                if (!dm_rsink->is_open()) {
                    fprintf(stderr, "beam: could not create ring %s\n", dm_ring.c_str());
                    return EXIT_FAILURE;
                }
                oopt.dm_sink = dm_rsink.get();
            }
        }
        observation_result ores;
        int orc = run_observation(pcfg, oopt, *src, pos.data(), dir.data(), &ores, std::cout);
        if (sink) std::cout << "Wrote " << sink->get_delivered() << " gemm-units of detected powers to " << sink_name << std::endl;
        if (dm_sink) std::cout << "Wrote " << dm_sink->get_times_written() << " dedispersed samples x " << my_trials << " trials to " << dm_path << std::endl;
        bf_comm_destroy(comm);
        if (orc != BF_OK) {
            fprintf(stderr, "GPUassert: %s (%d)\n", bf_last_error(), orc);
            return EXIT_FAILURE;
        }
        return 0;
    }

    debug_run_result res;
    int rc = run_debug_observation(cfg, opt, &res, nullptr, std::cout);
    if (rc != BF_OK) {
        fprintf(stderr, "GPUassert: %s (%d)\n", bf_last_error(), rc);
        return EXIT_FAILURE;  // the reference exit()s inside gpuErrchk; the driver keeps that policy here
    }
    std::cout << "Freeing CUDA Structures" << std::endl;
    std::cout << "Freed GPU memory" << std::endl;
    std::cout << "Freed CPU memory" << std::endl;
    return 0;
}
