// bf_geometry.cpp -- host mirror, part 1: antenna / beam types and config readers (src/beamformer.hh:183-311), usage and
// defines print-out, default geometry (src/beamformer.cu:135-147), channel frequencies, the steering weights
// (src/beamformer.cu:230-241) and the dedispersion trial ladder / delays (row f4).
// Plain host C++: no device code here.
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

// ---- antenna / beam_direction (src/beamformer.hh:183-213) -------------------------------------------------
std::istream& operator>>(std::istream& in, antenna& a) { return in >> a.x >> a.y >> a.z; }
std::ostream& operator<<(std::ostream& out, const antenna& a)
{
    return out << "(" << a.x << ", " << a.y << ", " << a.z << ")" << std::endl;
}
std::istream& operator>>(std::istream& in, beam_direction& a) { return in >> a.theta >> a.phi; }
std::ostream& operator<<(std::ostream& out, const beam_direction& a)
{
    return out << "(" << a.theta << ", " << a.phi << ")" << std::endl;
}

int read_in_beam_directions(const char* file_name, int expected_beams, beam_direction* dir)
{
    std::ifstream input_file;
    input_file.open(file_name);
    if (!input_file.is_open()) return -1;
    int nbeam = 0;
    input_file >> nbeam;
    if (nbeam != expected_beams) {
        std::cout << "Number of beams in file (" << nbeam << ") does not match expected (" << expected_beams << ")"
                  << std::endl;
        std::cout << "Excess beams will be ignored, missing beams will be set to 0." << std::endl;
    }
    for (int beam_idx = 0; beam_idx < expected_beams; beam_idx++) {
        beam_direction d;  // a failed extraction leaves this and later entries at their zero defaults
        input_file >> d;
        dir[beam_idx] = d;
    }
    return 0;
}

int read_in_position_locations(const char* file_name, int n_antennas, antenna* pos)
{
    std::ifstream input_file;
    input_file.open(file_name);
    if (!input_file.is_open()) return -1;
    int nant = 0;
    input_file >> nant;
    if (nant != n_antennas) {
        std::cout << "Number of antennas in file (" << nant << ") does not match N_ANTENNAS (" << n_antennas << ")"
                  << std::endl;
        std::cout << "Excess antennas will be ignored, missing antennas will be set to 0." << std::endl;
    }
    for (int ant = 0; ant < n_antennas; ant++) {
        antenna a;
        input_file >> a;
        pos[ant] = a;
    }
    return 0;
}

int write_array_to_disk_as_python_file(const float* data_out, int rows, int cols, const char* output_filename)
{
    std::ofstream f;
    f.open(output_filename);
    if (!f.is_open()) return -1;
    f << "A = [[";
    for (int jj = 0; jj < rows; jj++) {
        for (int ii = 0; ii < cols; ii++) {
            f << data_out[(size_t)jj * cols + ii];
            if (ii != cols - 1) f << ",";
        }
        if (jj != rows - 1)
            f << "],\n[";
        else
            f << "]]" << std::endl;
    }
    f.close();
    return 0;
}

void print_all_defines(const bf_config& c, std::ostream& out)
{
    const long long n_ipo = bf_n_inputs_per_output(&c), n_time = bf_n_timesteps_per_gemm(&c);
    const long long n_cx_in = (long long)c.n_ant * c.n_freq * n_time, n_cx_out = (long long)c.n_beams * c.n_freq * n_time;
    out << "N_BEAMS: " << c.n_beams << "\n";
    out << "N_ANTENNAS: " << c.n_ant << "\n";
    out << "N_FREQUENCIES: " << c.n_freq << "\n";
    out << "N_AVERAGING: " << c.n_avg << "\n";
    out << "N_POL: " << c.n_pol << "\n";
    out << "N_CX: " << 2 << "\n";
    out << "N_GEMMS_PER_GPU: " << c.n_gemms_per_block * c.n_blocks_on_gpu << "\n";
    out << "N_OUTPUTS_PER_GEMM: " << c.n_out_per_gemm << "\n";
    out << "N_GEMMS_PER_BLOCK: " << c.n_gemms_per_block << "\n";
    out << "N_INPUTS_PER_OUTPUT: " << n_ipo << "\n";
    out << "N_TIMESTEPS_PER_GEMM: " << n_time << "\n";
    out << "N_BLOCKS_ON_GPU: " << c.n_blocks_on_gpu << "\n";
    out << "N_CX_IN_PER_GEMM: " << n_cx_in << "\n";
    out << "N_CX_OUT_PER_GEMM: " << n_cx_out << "\n";
    out << "N_BYTES_POST_EXPANSION_PER_GEMM: " << n_cx_in * 2 << "\n";
    out << "N_BYTES_PRE_EXPANSION_PER_GEMM: " << bf_bytes_per_gemm(&c) << "\n";
    out << "N_BYTES_PRE_EXPANSION_PER_BLOCK: " << bf_bytes_per_block(&c) << "\n";
    out << "N_GPUS: " << kNGpus << "\n";
    out << "TOT_CHANNELS: " << kTotChannels << "\n";
    out << "START_F: " << kStartF << "\n";
    out << "END_F: " << kEndF << "\n";
    out << "ZERO_PT: " << kZeroPt << "\n";
    out << "BW_PER_CHANNEL: " << ((kEndF - kStartF) / kTotChannels) << "\n";
    out << "C_SPEED: " << kCSpeed << "\n";
    out << "PI: " << kPi << "\n";
    out << "N_BITS: " << 8 << "\n";
    out << "MAX_VAL: " << kMaxVal << "\n";
    out << "SIG_BITS: " << 4 << "\n";
    out << "SIG_MAX_VAL: " << kSigMaxVal << "\n";
    out << "N_STREAMS: " << c.n_streams << "\n";
    out << "N_SOURCES_PER_BATCH: " << kSourcesPerBatch << "\n";
    out << std::endl;
}

void usage(bool debug_mode, std::ostream& out)
{
    if (debug_mode) {
        out << "dsaX_beamformer_DEBUG_MODE [options]\n"
               " -g gpu                  select a predefined frequency range\n"
               " -p position_filename    file where the antenna positions are stored\n"
               " -d direction_filename   file where the beam directions are stored\n"
               " -s source_filename      file where the source directions are stored\n"
               " -h                      print usage\n";
    } else {
        out << "dsaX_beamformer [options]\n"
               " -c core                 bind process to CPU core\n"
               " -k key                  [default dada]\n"
               " -g gpu                  select a predefined frequency range\n"
               " -p position_filename    file where the antenna positions are stored\n"
               " -d direction_filename   file where the beam directions are stored\n"
               " -h                      print usage\n";
    }
}

void default_positions(int n_antennas, antenna* pos)
{
    for (int i = 0; i < n_antennas; i++) {
        pos[i] = antenna();
        pos[i].x = i * 500.0 / (n_antennas - 1) - 250.0;  // src/beamformer.cu:138
    }
}

void default_directions(int n_beams, beam_direction* dir)
{
    const double deg2rad_2fov = (2 * kHalfFov) * kPi / 180.0;  // DEG2RAD(2*HALF_FOV)
    const double deg2rad_fov = (kHalfFov)*kPi / 180.0;         // DEG2RAD(HALF_FOV)
    for (int i = 0; i < n_beams; i++) {
        dir[i] = beam_direction();
        dir[i].theta = i * deg2rad_2fov / (n_beams - 1) - deg2rad_fov;  // src/beamformer.cu:145
    }
}

float channel_frequency_weights(int gpu, int chan)
{
    float bw_per_channel = (kEndF - kStartF) / kTotChannels;  // src/beamformer.cu:173
    float freq = kEndF - (kZeroPt + gpu * kTotChannels / (kNGpus - 1) + chan) * bw_per_channel;  // :233
    return freq;
}

float channel_frequency_generator(int gpu, int chan)
{
    float freq = kEndF - (kZeroPt + gpu * kTotChannels / (kNGpus - 1) + chan) * ((kEndF - kStartF) / kTotChannels);
    return freq;  // src/test_data_generator.hh:72
}

void parallel_for(long n, const std::function<void(long, long)>& body)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (const char* e = getenv("DSABF_THREADS"))   // like OMP_NUM_THREADS for the reference's -fopenmp build (makefile:16)
        if (atoi(e) > 0) nt = (unsigned)atoi(e);
    if (nt == 0) nt = 1;
    if ((long)nt > n) nt = (unsigned)std::max<long>(1, n);
    if (nt <= 1) {
        body(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(body, n * t / nt, n * (t + 1) / nt);
    for (auto& x : th) x.join();
}

void generate_fourier_coefficients(int n_beams, int n_antennas, int n_freq, int chan0, int gpu, const antenna* pos,
                                   const beam_direction* dir, int8_t* out)
{
    parallel_for(n_freq, [&](long lo, long hi) {
        for (long i = lo; i < hi; i++) {
            float freq = channel_frequency_weights(gpu, chan0 + (int)i);
            float wavelength = kCSpeed / (1E9 * freq);  // src/beamformer.cu:234
            for (int j = 0; j < n_antennas; j++) {
                for (int k = 0; k < n_beams; k++) {
                    const double proj = (double)pos[j].x * ::sin((double)dir[k].theta) +
                                        (double)pos[j].y * ::sin((double)dir[k].phi);
                    int8_t* o = out + 2 * ((size_t)i * n_antennas * n_beams + (size_t)j * n_beams + k);
                    o[0] = (int8_t)::round(kMaxVal * ::cos(-2 * kPi * proj / (double)wavelength));  // :237
                    o[1] = (int8_t)::round(kMaxVal * ::sin(-2 * kPi * proj / (double)wavelength));  // :238
                }
            }
        }
    });
}

// ---- dedispersion trial ladder and delays (sandbox/Dispersion Theory.ipynb) ------------------------------------------
std::vector<double> dm_trials(double dm0, double dm_max, int nchan, double epsilon, double nu_ghz, double chan_bw_mhz,
                              double ti_us, double tscat_us, double tsamp_us)
{
    const double n2 = (double)nchan * (double)nchan;                               // cell 1
    const double alpha = 1.0 / (16 + n2);
    const double beta = ti_us * ti_us + tscat_us * tscat_us + tsamp_us * tsamp_us;
    const double k = (nu_ghz * nu_ghz * nu_ghz) / (8.3 * chan_bw_mhz);
    std::vector<double> dms{dm0};
    double dm_prev = dm0;
    while (dm_prev < dm_max) {                                                      // cell 2
        dm_prev = n2 * alpha * dm_prev + 4 * std::sqrt(alpha * (epsilon * epsilon - n2 * alpha) * dm_prev * dm_prev +
                                                       alpha * beta * (epsilon * epsilon - 1) * (k * k));
        dms.push_back(dm_prev);
    }
    return dms;
}

void dm_delays(const double* dms, int n_dm, const float* freq_ghz, int n_freq, double f_ref_ghz, double tsamp_ms,
               int32_t* out)
{
    for (int d = 0; d < n_dm; d++)
        for (int f = 0; f < n_freq; f++) {
            const double fr = (double)freq_ghz[f];                                  // cell 5
            out[(size_t)d * n_freq + f] =
                (int32_t)(4.15 * dms[d] * (-1.0 / (f_ref_ghz * f_ref_ghz) + 1.0 / (fr * fr)) / tsamp_ms);
        }
}

}  // namespace dsabf
