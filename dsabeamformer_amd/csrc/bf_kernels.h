// bf_kernels.h -- internal launch interface between the C-ABI runtime (bf_runtime.cpp) and the gfx950
// kernels (bf_kernels.hip).  Not installed; the public surface is include/dsabf.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace dsabf {

constexpr int kRowsPerChunk = 128;   // time samples per LDS buffer

struct Geometry {
    int n_beams, n_ant, n_freq, n_ipo, n_out, n_time;  // n_time = n_out * n_ipo (per gemm-unit)
    int n_ctiles;                                       // 16-beam column tiles: ceil(n_beams / 16)
    bool fast_detect;                                   // BF_DETECT_FAST requested (honoured by fused16_kernel, n_ipo >= 16)
    bool contracted_detect = false;                     // BF_DETECT_CONTRACTED requested
    bool paired = false;                                // weights verified conjugate-symmetric: beam B-1-b = conj(beam b)
    // Test / measurement switches, read from the environment ONCE per handle (read_env_switches, at bf_create) -- they decide
    // the layout of the weight images as well as the kernel, so a handle must not see them change between two calls:
    bool plain_wg_waves = false;                        // DSABF_WG_WAVES=4: 4-wave workgroups everywhere
    bool plain_col_tiles = false;                       // DSABF_COL_TILES=4: 4 output slots per wave everywhere
    // ... and three that only pick among launches of the same kernels; bf_set_switch changes them per handle for A/B runs
    int rtw_kout = 0;                                   // windows per stream of the run-time-window launch (0: fused_launch_shape decides)
    int tsplit = 0;                                     // DSABF_TSPLIT=n: time splits per frequency (0: fused_launch_shape decides)
    int lds_pad = 0;                                    // DSABF_LDS_PAD=bytes: extra dynamic LDS (fewer resident workgroups); clamped
    bool dm_wide = true;                                // DSABF_DM_WIDE=0: the per-thread-window DM kernel alone
    bool force_generic = false;                         // DSABF_GENERIC=1: fusedg_kernel (bf_fusedg.hip) for every geometry
    bool no_rtw = false;                                // DSABF_RTW=0: no run-time-window instantiations of fused16_kernel (-> fusedg_kernel)
    bool no_deep = false;                               // DSABF_DEEP=0: no three / four k-step classes of fused16_kernel (-> fusedg_kernel)
};
constexpr int kLdsPerCuBytes = 160 * 1024;
void read_env_switches(Geometry& g);
// getenv(name) in a process that says DSABF_LAB=1, NULL anywhere else: every measurement / test switch of the library reads the
// environment through this (the production allow-list -- DSABF_RCCL_LIB, DSABF_THREADS, DSABF_COALESCE, DSABF_PAIRED -- does not)
bool lab_mode();
const char* lab_getenv(const char* name);

// Bytes of the MFMA-fragment weight image: [freq][16-beam tile][re|im row][re|im operand][k-step][lane] x 16 B.
size_t weight_image_bytes(const Geometry& g);

// Conjugate-pair image (fused16_kernel<..., PAIRED>): [freq][pair tile][Wr|Wi|-Wi][lane] x 16 B; 0 bytes when the
// geometry has no paired kernel.
bool pairing_supported(const Geometry& g);
size_t weight_pair_image_bytes(const Geometry& g);

// True if the fused kernel has an instantiation for this geometry; `why` (optional) explains a refusal.
bool fused_supported(const Geometry& g, const char** why);

struct LaunchShape {
    int grid, block, lds_bytes, n_tsplit, chunks_total;
    int rt_kout;     // run-time-window class: whole windows per lane-group stream (0 otherwise)
    int n_bgroups;   // workgroups along the beam axis: ceil(n_beams / (64 * waves per workgroup))
};
// Waves per workgroup of the fused kernel for this geometry: 4, or 8 (two k-steps, n_ipo >= 16, an even number of 256-beam
// groups).  write_c: the stage-parity launch (always 4).
int fused_wg_waves(const Geometry& g, bool write_c = false);
// 16-beam output slots per wave: 4, or 8 (conjugate-pair kernel, two k-steps, n_ipo >= 16, n_beams a multiple of 512).
int fused_col_tiles(const Geometry& g, bool paired);
LaunchShape fused_launch_shape(const Geometry& g, int n_units, int n_cus, bool write_c = false);
// Whole windows per lane-group stream of a launch with run-time window boundaries (fused16_kernel<NIPO = 0>, fusedg_kernel): S samples
// per frequency, base = workgroups per time split (bf_kernels.hip).
int rtw_kout(const Geometry& g, long long S, long long base, int n_cus);

// Reference-layout weights [f][a][b]{re,im} (device) -> fragment image (device).  Sets *d_bad to non-zero if
// any imaginary part is -128 (its negation does not fit int8).  With d_pair_image (pairing_supported geometries) also
// builds the conjugate-pair image and sets d_bad[1] to non-zero unless W[f][a][B-1-b] == conj(W[f][a][b]) everywhere.
hipError_t launch_weight_relayout(const Geometry& g, const int8_t* d_w, void* d_image, void* d_pair_image, int* d_bad,
                                  hipStream_t s);

// Fused expand -> int8 MFMA -> detect over n_units gemm-units (g.paired selects the conjugate-pair kernel + image).
hipError_t launch_fused(const Geometry& g, const void* d_image, const void* d_pair_image, const void* d_packed,
                        int n_units, float* d_out, int n_cus, hipStream_t s);

// Same pipeline but stores the scaled complex GEMM result c[f][t][b]{re,im} for ONE gemm-unit (stage parity).
hipError_t launch_gemm_only(const Geometry& g, const void* d_image, const void* d_packed, float* d_c, int n_cus,
                            hipStream_t s);

hipError_t launch_expand(const void* d_in, size_t nbytes, void* d_out, hipStream_t s);
// back-to-back v_mfma_i32_16x16x64_i8 on the caller's operand bytes (the measured peak of SURVEY.md 8d)
constexpr size_t kMfmaPeakSrcBytes = (size_t)(512 + 256) * 256 * 16;   // 3 MiB read as operands
constexpr size_t kMfmaPeakSinkBytes = (size_t)4096 * 256 * sizeof(int);   // scratch for up to 1024 CUs
hipError_t launch_mfma_peak(const void* d_src, void* d_sink, int iters, int n_cus, double* ops, hipStream_t s);
// full[row][rank][row_floats] = stage[rank][row][row_floats] for every rank but skip_rank (-1: none); 16-byte aligned pointers,
// row_floats a multiple of 4 (the staged transport of bf_gather_detected_staged)
hipError_t launch_gather_relayout(const float* d_stage, float* d_full, size_t held, int world, size_t row_floats, int skip_rank,
                                  int n_cus, hipStream_t s);
hipError_t launch_dedisperse(const Geometry& g, const float* d_out_unit, float* d_ded, hipStream_t s);
// ded[u][b] = ascending-f fp32 sum of output 0 of unit u, units `unit_stride` floats apart: one launch for a whole block
hipError_t launch_dedisperse_units(const Geometry& g, const float* d_out_units, size_t unit_stride, int n_units, float* d_ded,
                                   hipStream_t s);
// out[dm][t][b] = sum_f series[t + delays[dm][f]][f][b]  (series: n_t beam-blocks [f][b]; t < n_t_out)
// d_flags: kDmScratchBytes of scratch owned by the caller, its last 512 bytes zero (one per group of kDwTrials trials; NULL: the per-thread-window
// kernel alone).  Groups whose delays fit run dedisperse_dm_wide_kernel (bf_dm_wide.hip), the rest dedisperse_dm_kernel.
hipError_t launch_dedisperse_dm(const Geometry& g, const float* d_series, int n_t, const int* d_delays, int n_dm,
                                int n_t_out, float* d_out, int* d_flags, hipStream_t s);
constexpr int kDwWavesPerWg = 16;            // waves per workgroup of the shared-window DM kernel
constexpr int kDwTrials = 2 * kDwWavesPerWg;   // trials per tile of the wide kernel (two per wave)
constexpr int kDwMaxGroups = 4096;   // flag ints a caller provides ...
constexpr size_t kDmScratchBytes = kDwMaxGroups * sizeof(int) + 512;   // ... followed by a 512-byte row of zeros (zero-filled by the caller)
constexpr int kDwMaxFreq = 1024;     // channels the wide kernel's LDS tables hold
bool dm_wide_supported(const Geometry& g, int n_dm);
hipError_t launch_dedisperse_dm_wide(const Geometry& g, const float* d_series, int n_t, const int* d_delays, int n_dm,
                                     int n_t_out, float* d_out, int* d_flags, hipStream_t s);

// ---- fusedg_kernel (bf_fusedg.hip): every geometry of the reference's contract the specialised instantiations do not cover --
constexpr int kGenericMaxAnt = 2048;
bool use_generic(const Geometry& g);      // does this geometry run fusedg_kernel?  (bf_kernels.hip)
bool generic_supported(const Geometry& g, const char** why);
int generic_ksteps(const Geometry& g);
int generic_interleave(const Geometry& g);
LaunchShape generic_launch_shape(const Geometry& g, int n_units, int n_cus);
hipError_t launch_fused_generic(const Geometry& g, const void* d_image, const void* d_packed, int n_units, float* d_out, int n_cus,
                                bool write_c, hipStream_t s);
int generic_vgprs(const Geometry& g);
size_t generic_image_extra_bytes(const Geometry& g);     // the offset-nibble corrections behind the fragment image
hipError_t launch_generic_colsum(const Geometry& g, const int8_t* d_w, void* d_image, hipStream_t s);

int fused_vgprs(const Geometry& g);  // from hipFuncGetAttributes, for reports
// The kernel instantiation a geometry runs (write_c: its stage-parity launch), spelled as the demangled kernel symbol spells it:
// "fused16_kernel<-1, 32, false, 0, true, 4, 4>" / "fusedg_kernel<true, 0, false>"; "" if there is none.  Host arithmetic only.
const char* fused_variant_key(const Geometry& g, bool write_c, char* buf, size_t n);
const char* generic_variant_key(const Geometry& g, bool write_c, char* buf, size_t n);
const char* fused_kernel_name(const Geometry& g, char* buf, size_t n);

}  // namespace dsabf
