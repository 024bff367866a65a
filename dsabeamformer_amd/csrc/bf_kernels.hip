// bf_kernels.hip -- hand-written gfx950 (CDNA4, MI355X) kernels for the DSA beamformer hot path.
//
// Replaces the reference's three device stages with ONE kernel, fused16_kernel:
//   expand_input (src/beamformer.cuh:66-109)  -> nibble expand in registers while staging into LDS
//   cublasGemmStridedBatchedEx (src/beamformer.cu:470-477) -> v_mfma_i32_16x16x64_i8 on a real-embedded K
//   detect_sum (src/beamformer.cuh:115-155)   -> power detect + time/pol accumulate in the accumulator VGPRs
// so the reference's d_B (expanded voltages) and d_C (complex fp32 beams, 16 MiB per beam-block in production)
// never touch HBM.  This is not a translation of those kernels; the design notes are in DESIGN.md section 3.
//
// Work decomposition
//   workgroup = 256 threads = 4 wave64, owns (frequency f, group of 256 beams, contiguous range of time chunks)
//   wave w    = 64 beams: its weight fragments live in VGPRs for the whole kernel; it streams every time tile of
//               the workgroup's range through MFMA.
//   time      = the MFMA row (M) axis, beams = the column (N) axis: each lane owns ONE beam per column tile and holds
//               4 consecutive time samples of it per accumulator, so the detect/accumulate is a sequential in-register
//               fp32 add chain in exactly the reference's order (bit-exact for every n_ipo, not only n_ipo = 2).
//
// Exact-arithmetic tricks (all proven in tests/test_numerics_tricks.py on the CPU):
//   * a packed byte b = (re << 4 | im & 15) is expanded to the int8 pair (b & 0xF0, (b << 4) & 0xF0) = (16*re,
//     16*im): two's complement places the signed nibble in the top of the byte, no sign-extension ops needed.
//     The MFMA therefore accumulates 16 * n (|16 n| <= 4,161,536 with 128 antennas).
//   * the accumulator is seeded with the int32 0x4B400000, the bit pattern of the float 1.5 * 2^23; adding the
//     integer 16 n to it yields the bit pattern of the float K + 16 n (K = 12582912) exactly, so no
//     v_cvt_f32_i32 is needed.
//   * x = fl(n * c), c = fl(1/127) (the reference's alpha, src/beamformer.cu:191) is obtained with ONE fma:
//     fma(K + 16 n, c/16, -K*c/16) -- K*c/16 = 6340995 * 2^-10 is exactly representable, so the fma rounds the
//     exact real n*c once, identical to (float)n * c.
//   * re^2 + im^2 is two multiplies and one add (compiled with -ffp-contract=off), then one add into the running
//     sum: the reference's `shmem += x*x + y*y` evaluated without contraction.
#include "bf_kernels.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#ifndef DSABF_PAIRED
#define DSABF_PAIRED 1    // build the conjugate-pair variants of fused16_kernel (used when the weights allow it)
#endif
#ifndef DSABF_PAIR_MFMA
#define DSABF_PAIR_MFMA 4 // MFMAs per conjugate pair tile: 4 (+-P2, +-P4 on the VALU), 5 (real part chained on the MFMA), 6
#endif
#ifndef DSABF_INTERLEAVE
#define DSABF_INTERLEAVE 1 // deal beams to a wave's column tiles 4 (pairs: 2) at a time -> 16- / 8-byte stores (beam_of_tile)
#endif
#ifndef DSABF_CLOCKPROBE
#define DSABF_CLOCKPROBE 0 // diagnostic build only (tools/clock_probe.sh): every workgroup overwrites out[blockIdx.x] with its
#endif                     // in-kernel shader clock in GHz (s_memtime / s_memrealtime around the chunk loop); results invalid
#ifndef DSABF_FASTADDR
#define DSABF_FASTADDR 1  // scalar chunk addressing in fused16_kernel when gemm-units are a multiple of the chunk span
#endif
#ifndef DSABF_OCC16
#define DSABF_OCC16 3     // register budget of the 64-antenna variants (168): the general kernel needs 153 VGPRs; the paired
                          // one (101) reaches 4 workgroups per CU by itself; capping at 128 spills for no gain
#endif

namespace dsabf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

constexpr unsigned kMagicBits = 0x4B400000u;        // float 12582912 = 1.5 * 2^23
constexpr float kMagic = 12582912.0f;
constexpr float kAlpha = (float)(1.0 / 127.0);      // h_inv_max_value.x, src/beamformer.cu:191
constexpr float kAlpha16 = kAlpha * 0.0625f;        // exact (power-of-two scaling)
constexpr float kNegMagicAlpha16 = -(kMagic * kAlpha16);
static_assert((double)kMagic * (double)kAlpha16 == (double)(kMagic * kAlpha16),
              "K * alpha/16 must be exactly representable for the single-fma conversion");

struct FusedArgs {
    const uint8_t* __restrict__ in;  // packed voltages [unit][f][t][a]
    const v4i* __restrict__ wimg;    // weight fragment image
    float* __restrict__ out;         // detected [unit*n_out + o][f][b]   (WRITE_C: c[f][t][b]{re,im})
    int n_freq, n_beams, n_btiles, n_bgroups;
    int T;                           // time samples per gemm-unit
    int t_shift;                     // log2(T) if T is a power of two, else -1
    unsigned S;                      // total time samples per frequency in this launch (n_units * T)
    int chunks_total;                // 128-sample chunks per frequency in this launch
    int n_tsplit;                    // workgroups along time
    int interleave;                  // beams are dealt to the column tiles of a wave 4 (pairs: 2) at a time: see beam_of_tile
};

// Which beam MFMA column c of a wave's column tile t computes.  A wave owns 64 consecutive beams (paired: 32 base
// beams and their mirror images).  Interleaved (n_beams % 64 == 0): beam = first + 4c + t (paired: 2c + t), so a lane's
// four results of one output are 4 (2 + 2) consecutive floats -> one 16-byte (two 8-byte) stores per lane, 256 (128)
// contiguous bytes per lane group, instead of four scattered 64-byte rows.  Otherwise: tile t = beams first + 16t + c.
__host__ __device__ inline int beam_of_tile(int interleave, int paired, int tile, int c)
{
    const int per = paired ? 2 : 4;                    // column tiles per wave
    if (!interleave) return tile * 16 + c;
    return (tile / per) * (16 * per) + per * c + tile % per;
}

// blockIdx -> (frequency f, beam group bg, time split ts).  Workgroups are dealt round-robin over the 8 XCDs, each
// with its own L2, so blocks b and b+8 share an L2: the low 3 bits of the block index select f % 8 (a frequency
// always lands on the same XCD), and the beam groups / time splits of one frequency are the NEXT-fastest index, so
// every workgroup that needs a frequency's 64-KiB weight panel (and, across beam groups, the same voltages) is
// resident at the same time on the same XCD: the panel is fetched once instead of once per time split
// (FETCH_SIZE 373 MB -> measured in profiles/).  Placement is only a speed matter; any mapping is correct.
__device__ __forceinline__ void decode_block(const FusedArgs& a, int& f, int& bg, int& ts)
{
    int bid = blockIdx.x;
    if ((a.n_freq & 7) == 0) {
        const int lo = bid & 7;
        bid >>= 3;
        bg = bid % a.n_bgroups;
        bid /= a.n_bgroups;
        ts = bid % a.n_tsplit;
        f = (bid / a.n_tsplit) * 8 + lo;
    } else {
        f = bid % a.n_freq;
        bid /= a.n_freq;
        bg = bid % a.n_bgroups;
        ts = bid / a.n_bgroups;
    }
}

// ---------------------------------------------------------------------------------------------------------
// a1 alone (API parity with expand_input): byte b -> (int8)(b >> 4), (int8)((int8)(b << 4) >> 4), order kept.
// HBM-bound: 16 B in, 32 B out per thread-iteration, fully coalesced.
__device__ __forceinline__ unsigned sext4x4(unsigned nib)  // four 4-bit values in the low nibbles of 4 bytes
{
    return ((nib ^ 0x88888888u) - 0x08080808u) ^ 0x80808080u;
}

__global__ void expand_kernel(const v4i* __restrict__ in, v4i* __restrict__ out, size_t n_vec)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        const v4i v = __builtin_nontemporal_load(in + i);
        v4i o0, o1;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const unsigned w = (unsigned)v[d];
            const unsigned hi = sext4x4((w >> 4) & 0x0F0F0F0Fu);
            const unsigned lo = sext4x4(w & 0x0F0F0F0Fu);
            // bytes (hi0, lo0, hi1, lo1) and (hi2, lo2, hi3, lo3)
            const unsigned e0 = __builtin_amdgcn_perm(lo, hi, 0x05010400u);
            const unsigned e1 = __builtin_amdgcn_perm(lo, hi, 0x07030602u);
            if (d < 2) {
                o0[2 * d] = (int)e0;
                o0[2 * d + 1] = (int)e1;
            } else {
                o1[2 * (d - 2)] = (int)e0;
                o1[2 * (d - 2) + 1] = (int)e1;
            }
        }
        __builtin_nontemporal_store(o0, out + 2 * i);
        __builtin_nontemporal_store(o1, out + 2 * i + 1);
    }
}

// a8: ded[b] = sum over f (ascending, fp32) of out[0][f][b]; one thread per beam.
__global__ void dedisperse_kernel(const float* __restrict__ out_unit, float* __restrict__ ded, int n_freq, int n_beams)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_beams) return;
    float acc = 0.0f;
    int f = 0;
    for (; f + 16 <= n_freq; f += 16) {  // 16 independent loads in flight, then the adds in ascending-f order
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = out_unit[(size_t)(f + i) * n_beams + b];
#pragma unroll
        for (int i = 0; i < 16; i++) acc = acc + v[i] * 1.0f;
    }
    for (; f < n_freq; f++) acc = acc + out_unit[(size_t)f * n_beams + b] * 1.0f;
    ded[b] = acc;
}

// 8f-4: incoherent dedispersion of a detected series: out[dm][t][b] = sum over f (ascending, fp32) of
// series[t + delay[dm][f]][f][b]; rows past the end of the series contribute nothing (adding +0 is the same thing: a
// running sum that starts at +0 never becomes -0).
//
// One thread = one beam x kDmTb consecutive output times x a block of kDmBlock consecutive DM trials.  Neighbouring
// trials need almost the same input rows (their delays differ by a few samples per channel), so for each frequency the
// thread loads ONE window of kDmTb + kDmSpan consecutive rows (first row = the delay of the block's first trial) and
// every trial of the block adds its kDmTb values out of that window: 24 coalesced row loads instead of 64 per frequency.
// The trial's offset into the window is only known at run time, and registers cannot be indexed dynamically, so the
// window lives in LDS -- as a PRIVATE column per thread ([row][thread]: conflict-free, no barrier: a thread only ever
// reads what it wrote).  The next frequency's rows are already in flight while the current window is consumed.  A
// trial whose delay falls outside the window (widely spaced or non-monotonic trials) reads its rows directly.
// Measured (profiles/r01_stage_kernels.json): 0.78 ms for 64 trials x 901 samples x 256 x 256 vs 1.09 ms for the
// one-trial-per-workgroup version it replaced; tile shapes (8,4,8) ... (16,8,16) were within +-20 % of this one.
constexpr int kDmTb = 8, kDmBlock = 8, kDmSpan = 16, kDmWin = kDmTb + kDmSpan, kDmThreads = 256;

__global__ __launch_bounds__(kDmThreads) void dedisperse_dm_kernel(const float* __restrict__ series,
                                                                   const int* __restrict__ delays, float* __restrict__ out,
                                                                   int n_t, int n_freq, int n_beams, int n_t_out, int n_dm)
{
    __shared__ float win[kDmWin][kDmThreads];
    const int dm0 = blockIdx.x * kDmBlock;
    const int t0 = blockIdx.y * kDmTb;
    const int tid = threadIdx.x;
    const int b = blockIdx.z * kDmThreads + tid;
    if (b >= n_beams) return;
    float acc[kDmBlock][kDmTb];
#pragma unroll
    for (int k = 0; k < kDmBlock; k++)
#pragma unroll
        for (int i = 0; i < kDmTb; i++) acc[k][i] = 0.0f;
    const size_t row_stride = (size_t)n_freq * n_beams;
    const int* dl0 = delays + (size_t)dm0 * n_freq;
    float nxt[kDmWin];
    auto load_window = [&](int f) {
        const float* p = series + (size_t)f * n_beams + b;
        const int first = t0 + __builtin_amdgcn_readfirstlane(dl0[f]);
#pragma unroll
        for (int j = 0; j < kDmWin; j++) {
            const int r = first + j;
            nxt[j] = (r >= 0 && r < n_t) ? p[(size_t)r * row_stride] : 0.0f;
        }
    };
    load_window(0);
    for (int f = 0; f < n_freq; f++) {
#pragma unroll
        for (int j = 0; j < kDmWin; j++) win[j][tid] = nxt[j];
        const int base = __builtin_amdgcn_readfirstlane(dl0[f]);
        if (f + 1 < n_freq) load_window(f + 1);
#pragma unroll
        for (int k = 0; k < kDmBlock; k++) {
            if (dm0 + k >= n_dm) break;
            const int dl = __builtin_amdgcn_readfirstlane(delays[(size_t)(dm0 + k) * n_freq + f]);
            const int d = dl - base;  // wave-uniform
            if (d >= 0 && d <= kDmSpan) {
                const float* w = &win[d][tid];
#pragma unroll
                for (int i = 0; i < kDmTb; i++) acc[k][i] = acc[k][i] + w[i * kDmThreads];
            } else {  // outside the window: direct loads
                const float* p = series + (size_t)f * n_beams + b;
#pragma unroll
                for (int i = 0; i < kDmTb; i++) {
                    const int r = t0 + dl + i;
                    const float v = (r >= 0 && r < n_t) ? p[(size_t)r * row_stride] : 0.0f;
                    acc[k][i] = acc[k][i] + v;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kDmBlock; k++)
#pragma unroll
        for (int i = 0; i < kDmTb; i++)
            if (dm0 + k < n_dm && t0 + i < n_t_out) out[((size_t)(dm0 + k) * n_t_out + t0 + i) * n_beams + b] = acc[k][i];
}

// =========================================================================================================
// fused16_kernel -- expand + complex int8 GEMM + detect in one kernel, built on v_mfma_i32_16x16x64_i8.
//
// Why this shape: on random int8 operands the chip holds a higher clock on the 16x16x64 instruction than on
// 32x32x32 (tools/ubench_shape.hip: 129-142 ns vs 149-157 ns per 262,144 MACs per SIMD) and the composite tile
// (MFMA + LDS fragment reads + canonical detect) is 7 % faster (tools/ubench_tile16.hip); a 32x32x32 implementation
// of the same design was measured 4 % (64 antennas) to 5 % (100 antennas) slower on the whole kernel and removed
// (git history, profiles/r01_variants_log.txt).  The 4-register accumulator tile lets one wave cover 64 beams, which
// halves the LDS fragment traffic per MFMA, and the detect of one tile interleaves with the MFMAs of the next in the
// wave's own in-order stream.
//
// Mapping, per group of 64 antennas (K' = 128 = 64 re | 64 im = two MFMAs of K = 64 chained through srcC):
//   A operand: 16 time rows; lane l supplies row l&15, bytes 16*(l>>4).. of the re (s = 0) or im (s = 1) half (LDS piece
//              4*s + (l>>4)).
//   D tile   : lane (column c = l&15, group g = l>>4) holds rows 4g..4g+3 in 4 registers.  Row 4g+r of a tile is
//              position 4*q + r of STREAM g, so every lane accumulates one output at a time, in time order, and a
//              128-row chunk holds 4 streams x 32 positions (n_ipo >= 32) or 2 x 4 streams x 16 positions.
//   A wave   : 4 column tiles = 64 beams; workgroup = 4 waves = 256 beams; chunk = 8 row tiles of 16.
template <int NIPO>
__device__ __forceinline__ int lds_row16(int t8, int rho)  // row of the chunk image read by A-row rho of tile t8
{
    if constexpr (NIPO >= 32)
        return (rho >> 2) * 32 + 4 * t8 + (rho & 3);
    else
        return (t8 >> 2) * 64 + (rho >> 2) * 16 + 4 * (t8 & 3) + (rho & 3);
}

template <int NIPO>
__device__ __forceinline__ int swz16(int chunk, int row)  // 8 chunks of 16 B per 128-B row; conflict-free both ways
{
    constexpr int LR = NIPO >= 32 ? 32 : 16;  // rows per stream in a chunk
    return chunk ^ ((((row >> 1) & 1) | (((row / LR) & 3) << 1)) ^ ((row & 1) << 2));
}

constexpr int kWaves16 = 4;                  // waves per workgroup of fused16_kernel
constexpr int kThreads16 = 64 * kWaves16;
constexpr int kColTiles16 = 4;               // 16-beam column tiles per wave

// PAIRED: the steering weights of beam B-1-b are the complex conjugates of those of beam b for every (frequency,
// antenna) -- true for any beam set that is symmetric about the boresight, e.g. the reference's linear fan and 16x16
// grid (checked exactly by pair_check_kernel when the weights are set).  Then with the four REAL K=64 products
//   P1 = sum Wr*Vr, P2 = sum Wi*Vi, P3 = sum Wr*Vi, P4 = sum Wi*Vr        (one 16x16x64 MFMA each)
// C(b) = (P1 - P2) + j(P3 + P4) and C(B-1-b) = (P1 + P2) + j(P3 - P4): two beams for the MFMA work of one, exact in
// int32 (the +-P2 / +-P4 are 4 integer VALU ops per sample pair; P1 and P3 carry the float seed, P2 and P4 start at 0).
//
// AIN = antennas per time sample (64, 100 or 128).  More than 64 antennas are two k-steps of 64: the LDS chunk image
// becomes two 128-row planes (antennas 0-63 | 64-127), every product is a chain of two MFMAs, and the detect -- whose
// cost does not depend on the antenna count -- is amortised over twice the MACs.  100 antennas run as 128 with zero
// weights behind antenna 99; their packed rows (100 B) are only dword-aligned, so they are staged in 4-byte pieces.
template <int AIN, int NIPO, bool WRITE_C, bool FAST = false, bool PAIRED = false>
__global__ __launch_bounds__(kThreads16, AIN > 64 ? 2 : DSABF_OCC16) void fused16_kernel(FusedArgs a)
{
    static_assert(!FAST || (NIPO >= 16 && !WRITE_C), "the fast detect exists for n_ipo >= 16 only");
    static_assert(!(PAIRED && WRITE_C), "the stage-parity path always runs the general kernel");
    static_assert(AIN % 4 == 0 && AIN <= 128, "antenna count");
    constexpr int A = AIN, RB = 128;
    constexpr int KS = AIN > 64 ? 2 : 1;                 // k-steps of 64 antennas
    constexpr int PLANE = kRowsPerChunk * RB;            // LDS bytes of one k-step's chunk image
    constexpr int BUF = KS * PLANE;
    constexpr bool DW = (AIN % 16) != 0;                 // rows only dword-aligned: 4-byte staging pieces
    constexpr int PB = DW ? 4 : 16;                      // bytes per staging piece
    constexpr int PPR = AIN / PB;                        // pieces per time sample
    constexpr int TOTALP = kRowsPerChunk * PPR;          // pieces per chunk
    constexpr int NS = kColTiles16;                      // 16-beam output slots per lane (beams per wave = 16 * NS)
    constexpr int NT = PAIRED ? NS / 2 : NS;             // MFMA column tiles per wave (a paired tile feeds 2 slots)
    constexpr bool LONG = NIPO >= 16;
    constexpr int L = LONG ? NIPO : 16;                  // samples per stream
    constexpr int LR = NIPO >= 32 ? 32 : 16;             // stream rows held by one chunk
    constexpr int CPG = L > 32 ? L / 32 : 1;             // chunks per group of 4 streams
    constexpr int PPT = (TOTALP + kThreads16 - 1) / kThreads16;  // pieces per thread per chunk (2; 4; 13 for 100 antennas)
    using stage_t = std::conditional_t<DW, int, v4i>;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 buffers x KS planes x 128 rows x 128 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4;   // lane group = stream within the tile / k-block of the operands
    const int c16 = lane & 15;  // column within a 16-beam tile / A row

    int f, bg, ts;
    decode_block(a, f, bg, ts);
    const int units_total = a.chunks_total / CPG;
    const int c_begin = (int)(((long long)units_total * ts) / a.n_tsplit) * CPG;
    const int c_end = (int)(((long long)units_total * (ts + 1)) / a.n_tsplit) * CPG;

    // ---- which beams this lane produces, and the weight fragments ------------------------------------------
    int slot_beam[NS];                                    // beam index of output slot s (>= n_beams: none)
    constexpr int NPC = DSABF_PAIR_MFMA >= 5 ? 3 : 2;     // paired fragments per tile: Wr, Wi (, -Wi)
    v4i bw[NT][PAIRED ? NPC : 4][KS];                     // general: [ct][2*rho + s][k-step]; paired: [pct][Wr, Wi, -Wi][k-step]
    bool wave_active;
    if constexpr (PAIRED) {
        const int n_pct = a.n_btiles;                     // pair tiles of 16 base beams = n_beams / 32
        const int pct0 = (bg * kWaves16 + wave) * NT;
        wave_active = pct0 < n_pct;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int bb = beam_of_tile(a.interleave, 1, pct0 + t, c16);  // base beam (< n_beams / 2)
            const bool ok = pct0 + t < n_pct;
            slot_beam[2 * t] = ok ? bb : a.n_beams;
            slot_beam[2 * t + 1] = ok ? a.n_beams - 1 - bb : a.n_beams;
#pragma unroll
            for (int comp = 0; comp < NPC; comp++)
#pragma unroll
                for (int h = 0; h < KS; h++)
                    bw[t][comp][h] =
                        ok ? a.wimg[((((size_t)f * n_pct + pct0 + t) * 3 + comp) * KS + h) * 64 + lane] : v4i{0, 0, 0, 0};
        }
    } else {
        const int n_ctiles = a.n_btiles * 2;
        const int ct0 = (bg * kWaves16 + wave) * NT;      // first 16-beam column tile of this wave
        wave_active = ct0 < n_ctiles;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const bool ok = ct0 + t < n_ctiles;
            slot_beam[t] = ok ? beam_of_tile(a.interleave, 0, ct0 + t, c16) : a.n_beams;
#pragma unroll
            for (int k = 0; k < 4; k++)                    // k = 2*rho + s
#pragma unroll
                for (int h = 0; h < KS; h++)
                    bw[t][k][h] =
                        ok ? a.wimg[((((size_t)f * n_ctiles + ct0 + t) * 4 + k) * KS + h) * 64 + lane] : v4i{0, 0, 0, 0};
        }
    }

    v4i kc = {(int)kMagicBits, (int)kMagicBits, (int)kMagicBits, (int)kMagicBits};
    asm volatile("" : "+v"(kc));
    const v4i kzero = {0, 0, 0, 0};

    // ---- staging (the chunk's 128 samples are contiguous in time for n_ipo <= 32) ------------------------------
    auto run_sample0 = [&](int c, int run) -> unsigned {   // first global sample of stream-run `run` of chunk c
        if constexpr (NIPO >= 32)
            return (4u * (unsigned)(c / CPG) + (unsigned)run) * (unsigned)L + 32u * (unsigned)(c % CPG);
        else
            return (unsigned)c * 128u + 16u * (unsigned)run;
    };
    stage_t stage[PPT];
    // Fast addressing: when a chunk's sample span (128 samples, 256 for n_ipo = 64) never straddles a gemm-unit, the
    // unit / time split of the chunk is wave-uniform -- a scalar base that advances by one span per chunk -- and the
    // per-lane part (row and piece) is a constant 32-bit offset: no vector integer arithmetic (the generic
    // path costs ~17 VALU ops, 6 of them quarter-rate 32-bit multiplies, per load).
    constexpr unsigned SPAN = (NIPO == 64) ? 256u : 128u;
    const bool fast_addr = DSABF_FASTADDR && ((unsigned)a.T % SPAN) == 0;
    // Piece k of this thread is piece pc = tid + 256 k of the chunk: row pc / PPR, position pc % PPR.  Its byte offset
    // from the chunk's first sample is PB * pc (rows are PPR * PB bytes and consecutive) -- except for n_ipo = 64, whose
    // chunk rows are four runs of 32 samples, 64 apart.  The 16*im image sits 4 pieces away from the 16*re image (after it
    // in plane 0, before it in plane 1), and the swizzle only XORs the 3 piece bits, so its LDS offset is the re offset ^ 64.
    [[maybe_unused]] unsigned lane_off64[NIPO == 64 ? PPT : 1];
    int lds_re[PPT];                      // LDS byte offset (inside one buffer) of the piece's 16*re image
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int pc = tid + k * kThreads16;
        const int row = (pc / PPR) % kRowsPerChunk, pi = pc % PPR;   // (% keeps the unused tail pieces in range)
        if constexpr (NIPO == 64) lane_off64[k] = (unsigned)((row / LR) * L + (row % LR)) * A + pi * PB;
        const int blk = DW ? pi / 4 : pi;                            // 16-antenna block of the piece
        const int h = blk / 4, kp = blk % 4, sub = DW ? 4 * (pi % 4) : 0;
        lds_re[k] = h * PLANE + row * RB + 16 * swz16<NIPO>(kp + 4 * (h & 1), row) + sub;  // plane 1: halves swapped
    }
    auto lane_off = [&](int k) -> unsigned {
        if constexpr (NIPO == 64)
            return lane_off64[k];
        else
            return (unsigned)PB * (unsigned)(tid + k * kThreads16);
    };
    auto piece_live = [&](int k) { return (TOTALP % kThreads16 == 0) || (tid + k * kThreads16 < TOTALP); };
    int ld_span = -1;                 // span the scalar state below describes
    unsigned ld_u = 0, ld_t0 = 0;     // its gemm-unit and first sample inside the unit
    auto load_chunk = [&](int c) {
        if (fast_addr) {
            const int span = (NIPO == 64) ? c / 2 : c;
            if (ld_span < 0) {
                const unsigned s_c = (unsigned)span * SPAN;
                ld_u = a.t_shift >= 0 ? (s_c >> a.t_shift) : (s_c / (unsigned)a.T);
                ld_t0 = s_c - ld_u * (unsigned)a.T;
                ld_span = span;
            }
            while (ld_span < span) {  // at most one step: chunks are loaded in order
                ld_t0 += SPAN;
                if (ld_t0 >= (unsigned)a.T) {
                    ld_t0 = 0;
                    ld_u++;
                }
                ld_span++;
            }
            const bool valid = (unsigned)span * SPAN < a.S;
            const unsigned half = (NIPO == 64) ? 32u * (unsigned)(c & 1) : 0u;
            const uint8_t* base = a.in + ((size_t)((size_t)ld_u * a.n_freq + f) * a.T + ld_t0 + half) * A;
#pragma unroll
            for (int k = 0; k < PPT; k++) {
                stage[k] = stage_t{};
                if (valid && piece_live(k)) stage[k] = *reinterpret_cast<const stage_t*>(base + lane_off(k));
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int pc = tid + k * kThreads16;
            const int row = (pc / PPR) % kRowsPerChunk, pi = pc % PPR;
            const unsigned s0 = run_sample0(c, row / LR) + (unsigned)(row % LR);
            stage[k] = stage_t{};
            if (s0 < a.S && piece_live(k)) {
                const unsigned u = a.t_shift >= 0 ? (s0 >> a.t_shift) : (s0 / (unsigned)a.T);
                const unsigned t = s0 - u * (unsigned)a.T;
                stage[k] = *reinterpret_cast<const stage_t*>(a.in + ((size_t)((size_t)u * a.n_freq + f) * a.T + t) * A + pi * PB);
            }
        }
    };
    auto write_chunk = [&](char* buf) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            if (!piece_live(k)) continue;
            if constexpr (DW) {
                const unsigned w = (unsigned)stage[k];
                *reinterpret_cast<int*>(buf + lds_re[k]) = (int)(w & 0xF0F0F0F0u);
                *reinterpret_cast<int*>(buf + (lds_re[k] ^ 64)) = (int)((w << 4) & 0xF0F0F0F0u);
            } else {
                v4i re, im;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned w = (unsigned)stage[k][d];
                    re[d] = (int)(w & 0xF0F0F0F0u);
                    im[d] = (int)((w << 4) & 0xF0F0F0F0u);
                }
                *reinterpret_cast<v4i*>(buf + lds_re[k]) = re;
                *reinterpret_cast<v4i*>(buf + (lds_re[k] ^ 64)) = im;
            }
        }
    };

    const size_t FB = (size_t)a.n_freq * a.n_beams;
    float sum[NS];                         // running sum of this lane's current output, per slot
#pragma unroll
    for (int sl = 0; sl < NS; sl++) sum[sl] = 0.0f;
    constexpr int PEND = LONG ? (L >= 32 ? 1 : 2) : 1;   // outputs completed per chunk per lane (LONG)
    float pend[PEND][NS];
    int pend_chunk[PEND];                  // chunk whose finished sums sit in pend[gi] (-1: none); tracked per entry
#pragma unroll                             // because entry 0 of chunk c can be parked before entry 1 of chunk c-1 left
    for (int gi = 0; gi < PEND; gi++) pend_chunk[gi] = -1;
    // x[sl] -> row[beam of slot sl]; `row` points at beam 0 of one output's frequency row.  Interleaved tiles give every
    // lane consecutive beams: vector stores.
    auto store_slots = [&](float* row, const float (&x)[NS]) {
        if (a.interleave) {
            if constexpr (PAIRED) {   // slots 0, 2 = base beams bb, bb + 1; slots 1, 3 = their mirrors B-1-bb, B-2-bb
                *reinterpret_cast<v2f*>(row + slot_beam[0]) = v2f{x[0], x[2]};
                *reinterpret_cast<v2f*>(row + slot_beam[3]) = v2f{x[3], x[1]};
            } else {
                *reinterpret_cast<v4f*>(row + slot_beam[0]) = v4f{x[0], x[1], x[2], x[3]};
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < NS; sl++)
                if (slot_beam[sl] < a.n_beams) row[slot_beam[sl]] = x[sl];
        }
    };
    auto flush_pending = [&]() {
        if constexpr (LONG && !WRITE_C) {
#pragma unroll
            for (int gi = 0; gi < PEND; gi++) {
                if (pend_chunk[gi] >= 0 && wave_active) {
                    const unsigned grp = (NIPO >= 32) ? (unsigned)(pend_chunk[gi] / CPG) : (2u * pend_chunk[gi] + gi);
                    float* ub = a.out + ((size_t)(4u * grp) * FB + (size_t)f * a.n_beams);  // wave-uniform part
                    const unsigned o = 4u * grp + (unsigned)g4;
                    if (o * (unsigned)L < a.S) store_slots(ub + (size_t)g4 * FB, pend[gi]);
                }
                pend_chunk[gi] = -1;
            }
        }
    };

    if (c_begin >= c_end) return;
#if DSABF_CLOCKPROBE
    const unsigned long long probe_t0 = __builtin_amdgcn_s_memtime(), probe_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    load_chunk(c_begin);
    write_chunk(smem);
    if (c_begin + 1 < c_end) load_chunk(c_begin + 1);
    __syncthreads();

    for (int c = c_begin; c < c_end; c++) {
        char* cur = smem + ((c - c_begin) & 1) * BUF;
        char* nxt = smem + ((c - c_begin + 1) & 1) * BUF;
        if (!wave_active) {
            if (c + 1 < c_end) write_chunk(nxt);
            if (c + 2 < c_end) load_chunk(c + 2);
        } else {
            [[maybe_unused]] float ov[2][NS];   // n_ipo < 16: the outputs the current tile completed, per slot
            // detect + accumulate the 4 samples (fr, fi: accumulator bit patterns K + 16 n) of output slot sl
            auto detect = [&](const int t8, const v4f fr, const v4f fi, const int sl) {
                // stream position of this tile's rows and whether it starts / ends an output
                const int gi = (NIPO >= 32) ? 0 : (t8 >> 2);          // group inside the chunk (L = 16)
                const int q4 = (NIPO >= 32) ? (32 * (c % CPG) + 4 * t8) : 4 * (t8 & 3);  // position of register 0
                const unsigned grp = (NIPO >= 32) ? (unsigned)(c / CPG) : (2u * (unsigned)c + gi);
                const unsigned o = 4u * grp + (unsigned)g4;           // this lane's stream (output index if LONG)
                const int beam = slot_beam[sl];
                if constexpr (WRITE_C) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const unsigned sidx = o * (unsigned)L + (unsigned)(q4 + r);
                        if (o * (unsigned)L < a.S && beam < a.n_beams) {
                            v2f cv = {__builtin_fmaf(fr[r], kAlpha16, kNegMagicAlpha16),
                                      __builtin_fmaf(fi[r], kAlpha16, kNegMagicAlpha16)};
                            *reinterpret_cast<v2f*>(a.out + 2 * (((size_t)f * a.T + sidx) * a.n_beams + beam)) = cv;
                        }
                    }
                } else if constexpr (FAST) {
                    // BF_DETECT_FAST: d = 16 n exactly (one subtract), acc = fma(d, d, acc): 4 ops per sample;
                    // the (alpha/16)^2 scale is applied once per output when it is parked for the store.
                    float sacc = (q4 == 0) ? 0.0f : sum[sl];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float dr = fr[r] - kMagic, di = fi[r] - kMagic;
                        sacc = __builtin_fmaf(dr, dr, sacc);
                        sacc = __builtin_fmaf(di, di, sacc);
                    }
                    asm volatile("" : "+v"(sacc));
                    sum[sl] = sacc;
                    if (q4 + 4 == L) {
                        pend[gi][sl] = sacc * (kAlpha16 * kAlpha16);
                        pend_chunk[gi] = c;
                    }
                } else {
                    float p[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float x = __builtin_fmaf(fr[r], kAlpha16, kNegMagicAlpha16);
                        const float y = __builtin_fmaf(fi[r], kAlpha16, kNegMagicAlpha16);
                        const float xx = x * x;
                        const float yy = y * y;
                        p[r] = xx + yy;
                    }
                    if constexpr (LONG) {
                        float sacc = (q4 == 0) ? p[0] : (sum[sl] + p[0]);
                        sacc = sacc + p[1];
                        sacc = sacc + p[2];
                        sacc = sacc + p[3];
                        asm volatile("" : "+v"(sacc));
                        sum[sl] = sacc;
                        if (q4 + 4 == L) {
                            pend[gi][sl] = sacc;
                            pend_chunk[gi] = c;
                        }
                    } else {
                        // 16-sample stream = 16/NIPO outputs; registers r hold positions q4 + r.  Finished outputs are
                        // collected per slot (ov) and stored together after the tile's last column tile.
                        if constexpr (NIPO == 2) {
                            ov[0][sl] = p[0] + p[1];
                            ov[1][sl] = p[2] + p[3];
                        } else if constexpr (NIPO == 4) {
                            float sacc = p[0] + p[1];
                            sacc = sacc + p[2];
                            ov[0][sl] = sacc + p[3];
                        } else {  // NIPO == 8
                            float sacc = (q4 % 8 == 0) ? p[0] : (sum[sl] + p[0]);
                            sacc = sacc + p[1];
                            sacc = sacc + p[2];
                            sacc = sacc + p[3];
                            asm volatile("" : "+v"(sacc));
                            sum[sl] = sacc;
                            ov[0][sl] = sacc;
                        }
                    }
                }
            };
            // stores of the outputs a short-window (n_ipo < 16) tile completed
            auto store_short = [&](const int t8) {
                if constexpr (!LONG && !WRITE_C) {
                    const int gi = t8 >> 2, q4 = 4 * (t8 & 3);
                    const unsigned o = 4u * (2u * (unsigned)c + gi) + (unsigned)g4;   // this lane's 16-sample stream
                    if (o * 16u < a.S) {
                        float* base = a.out + ((size_t)o * (16 / NIPO)) * FB + (size_t)f * a.n_beams;
                        if constexpr (NIPO == 2) {
                            store_slots(base + (size_t)(q4 / 2) * FB, ov[0]);
                            store_slots(base + (size_t)(q4 / 2 + 1) * FB, ov[1]);
                        } else if constexpr (NIPO == 4) {
                            store_slots(base + (size_t)(q4 / 4) * FB, ov[0]);
                        } else {
                            if (q4 % 8 == 4) store_slots(base + (size_t)(q4 / 8) * FB, ov[0]);
                        }
                    }
                }
            };

            // LDS fragments of row-tile t8: a0[h] = 16*re, a1[h] = 16*im of 16 antennas x 16 samples per lane group, k-step h
            auto read_frag = [&](const int t8, v4i (&a0)[KS], v4i (&a1)[KS]) {
                const int row = lds_row16<NIPO>(t8, c16);
#pragma unroll
                for (int h = 0; h < KS; h++) {  // plane 1 keeps (im | re): the two planes' staging writes then never collide
                    a0[h] = *reinterpret_cast<const v4i*>(cur + h * PLANE + row * RB + 16 * swz16<NIPO>(g4 + 4 * (h & 1), row));
                    a1[h] = *reinterpret_cast<const v4i*>(cur + h * PLANE + row * RB + 16 * swz16<NIPO>(g4 + 4 * ((h & 1) ^ 1), row));
                }
            };
            // acc = seed + sum over the k-steps of x[h] * w[h]  (one MFMA per k-step, chained through srcC)
            auto dot = [&](const v4i (&x)[KS], const v4i (&w)[KS], v4i acc) {
#pragma unroll
                for (int h = 0; h < KS; h++) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(x[h], w[h], acc, 0, 0, 0);
                return acc;
            };
            // One step = the MFMAs of column tile t on row-tile fragments (a0, a1); its SPS output slots land in
            // re[] / im[] as accumulator bit patterns K + 16 n.
            constexpr int SPS = PAIRED ? 2 : 1;                 // output slots per step
            auto issue = [&](const v4i (&a0)[KS], const v4i (&a1)[KS], const int t, v4i (&re)[SPS], v4i (&im)[SPS]) {
                if constexpr (PAIRED) {
                    const v4i p1 = dot(a0, bw[t][0], kc);     // Wr*Vr + K
                    const v4i p3 = dot(a1, bw[t][0], kc);     // Wr*Vi + K
                    if constexpr (DSABF_PAIR_MFMA >= 5) {   // +-P2 chained on the MFMA pipe (bw[t][2] = -Wi)
                        re[0] = dot(a1, bw[t][2], p1);
                        re[1] = dot(a1, bw[t][1], p1);
                    } else {
                        const v4i p2 = dot(a1, bw[t][1], kzero);  // Wi*Vi
                        re[0] = p1 - p2;
                        re[1] = p1 + p2;
                    }
                    if constexpr (DSABF_PAIR_MFMA >= 6) {
                        im[0] = dot(a0, bw[t][1], p3);
                        im[1] = dot(a0, bw[t][2], p3);
                    } else {
                        const v4i p4 = dot(a0, bw[t][1], kzero);  // Wi*Vr
                        im[0] = p3 + p4;
                        im[1] = p3 - p4;
                    }
                } else {
                    re[0] = dot(a1, bw[t][1], dot(a0, bw[t][0], kc));
                    im[0] = dot(a1, bw[t][3], dot(a0, bw[t][2], kc));
                }
            };
            auto consume = [&](const int t8, const int t, const v4i (&re)[SPS], const v4i (&im)[SPS]) {
#pragma unroll
                for (int e = 0; e < SPS; e++)   // paired: slot 2t = beam b, slot 2t+1 = beam B-1-b
                    detect(t8, __builtin_bit_cast(v4f, re[e]), __builtin_bit_cast(v4f, im[e]), SPS * t + e);
            };
            // staging work in the shadow of the MFMA stream: the next chunk's LDS image after tile 1, the parked stores
            // of the previous chunk and the prefetch of chunk c+2 after tile 3
            auto staging = [&](const int t8) {
                if (t8 == 1 && c + 1 < c_end) write_chunk(nxt);
                if (t8 == 3) {
                    flush_pending();
                    if (c + 2 < c_end) load_chunk(c + 2);
                }
            };
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) {
                v4i a0[KS], a1[KS];
                read_frag(t8, a0, a1);
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    v4i re[SPS], im[SPS];
                    issue(a0, a1, t, re, im);
                    consume(t8, t, re, im);
                }
                store_short(t8);
                staging(t8);
            }
        }
        __syncthreads();
    }
    flush_pending();
#if DSABF_CLOCKPROBE
    __syncthreads();
    if (tid == 0) {
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - probe_t0, dr = __builtin_amdgcn_s_memrealtime() - probe_r0;
        a.out[blockIdx.x] = (float)((double)dt / (double)dr * 0.1);  // s_memrealtime ticks at 100 MHz
    }
#endif
}

// Conjugate-pair test: *flag stays 0 iff W[f][a][B-1-b] == conj(W[f][a][b]) for every f, a and b < B/2.
__global__ void pair_check_kernel(const int8_t* __restrict__ w, size_t n_fa, int n_beams, int* __restrict__ flag)
{
    const size_t total = n_fa * (size_t)(n_beams / 2);
    bool bad = false;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const size_t fa = idx / (n_beams / 2);
        const int b = (int)(idx % (n_beams / 2));
        const int8_t* e0 = w + 2 * (fa * n_beams + b);
        const int8_t* e1 = w + 2 * (fa * n_beams + (n_beams - 1 - b));
        bad |= (e0[0] != e1[0]) || (e0[1] != -e1[1]);
    }
    if (bad) *flag = 1;
}

// Paired weight image: image[f][pct][comp][h][lane] (16 bytes): lane = 16*kb + c; byte i = Wr (comp 0), Wi (comp 1) or
// -Wi (comp 2) of antenna 64*h + 16*kb + i (zero behind the last antenna) for base beam beam_of_tile(pct, c) (< n_beams / 2).
__global__ void weight_relayout16p_kernel(const int8_t* __restrict__ w, v4i* __restrict__ image, int n_freq, int n_ant,
                                          int n_beams, int ks, int interleave)
{
    const int n_pct = n_beams / 32;
    const size_t total = (size_t)n_freq * n_pct * 3 * ks * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        size_t r = idx >> 6;
        const int h = (int)(r % ks);
        r /= ks;
        const int comp = (int)(r % 3);
        r /= 3;
        const int pct = (int)(r % n_pct);
        const int f = (int)(r / n_pct);
        const int kb = lane >> 4, b = beam_of_tile(interleave, 1, pct, lane & 15);
        unsigned d[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
            const int ant = 64 * h + kb * 16 + i;
            int v = 0;
            if (ant < n_ant) v = w[2 * (((size_t)f * n_ant + ant) * n_beams + b) + (comp ? 1 : 0)];
            if (comp == 2) v = -v;
            d[i >> 2] |= ((unsigned)v & 0xFFu) << (8 * (i & 3));
        }
        image[idx] = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
    }
}

// 16x16x64 weight image: image[f][ct16][rho][s][h][lane] (16 bytes): lane = 16*kb + c; byte i multiplies LDS chunk
// 4*s + kb of the A row of k-step h = component s (0 = re, 1 = im) of antenna 64*h + 16*kb + i (zero behind the last
// antenna), for output row rho of beam beam_of_tile(ct16, c).
__global__ void weight_relayout16_kernel(const int8_t* __restrict__ w, v4i* __restrict__ image, int n_freq, int n_ant,
                                         int n_beams, int ks, int interleave, int* __restrict__ bad)
{
    const int n_ct = n_beams / 16;
    const size_t total = (size_t)n_freq * n_ct * 2 * 2 * ks * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        size_t r = idx >> 6;
        const int h = (int)(r % ks);
        r /= ks;
        const int sk = (int)(r & 1);
        r >>= 1;
        const int rho = (int)(r & 1);
        r >>= 1;
        const int ct = (int)(r % n_ct);
        const int f = (int)(r / n_ct);
        const int kb = lane >> 4, b = beam_of_tile(interleave, 0, ct, lane & 15);
        unsigned d[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
            const int ant = 64 * h + kb * 16 + i;
            int v = 0;
            if (ant < n_ant && b < n_beams) {
                const int8_t* e = w + 2 * (((size_t)f * n_ant + ant) * n_beams + b);
                const int wr = e[0], wi = e[1];
                if (wi == -128) *bad = 1;
                v = (rho == 0) ? (sk == 0 ? wr : -wi) : (sk == 0 ? wi : wr);
            }
            d[i >> 2] |= ((unsigned)v & 0xFFu) << (8 * (i & 3));
        }
        image[idx] = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
    }
}

template <int AIN, int NIPO, bool WRITE_C, bool FAST = false, bool PAIRED = false>
hipError_t launch_fused16_t(const FusedArgs& args, const LaunchShape& ls, hipStream_t s)
{
    auto kern = fused16_kernel<AIN, NIPO, WRITE_C, FAST, PAIRED>;
    if (ls.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           ls.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ls.grid), dim3(ls.block), ls.lds_bytes, s, args);
    return hipGetLastError();
}

// the fused16_kernel instantiation for (n_ant, n_ipo, mode); nullptr if there is none
template <int AIN, bool WRITE_C, bool FAST, bool PAIRED>
const void* fused16_fn_ant(int n_ipo, hipError_t (**launch)(const FusedArgs&, const LaunchShape&, hipStream_t))
{
#define DSABF_CASE16(nipo_)                                                              \
    case nipo_:                                                                          \
        if constexpr (!FAST || nipo_ >= 16) {                                            \
            *launch = launch_fused16_t<AIN, nipo_, WRITE_C, FAST, PAIRED>;               \
            return reinterpret_cast<const void*>(fused16_kernel<AIN, nipo_, WRITE_C, FAST, PAIRED>); \
        }                                                                                \
        return nullptr;
    switch (n_ipo) {
        DSABF_CASE16(2)
        DSABF_CASE16(32)
        default: break;
    }
    if constexpr (AIN == 64) {
        switch (n_ipo) {
            DSABF_CASE16(4)
            DSABF_CASE16(8)
            DSABF_CASE16(16)
            DSABF_CASE16(64)
            default: break;
        }
    }
#undef DSABF_CASE16
    return nullptr;
}

template <bool WRITE_C, bool FAST, bool PAIRED>
const void* fused16_fn(int n_ant, int n_ipo, hipError_t (**launch)(const FusedArgs&, const LaunchShape&, hipStream_t))
{
    switch (n_ant) {
        case 64: return fused16_fn_ant<64, WRITE_C, FAST, PAIRED>(n_ipo, launch);
        case 16: return fused16_fn_ant<16, WRITE_C, FAST, PAIRED>(n_ipo, launch);
        case 32: return fused16_fn_ant<32, WRITE_C, FAST, PAIRED>(n_ipo, launch);
        case 100: return fused16_fn_ant<100, WRITE_C, FAST, PAIRED>(n_ipo, launch);
        case 128: return fused16_fn_ant<128, WRITE_C, FAST, PAIRED>(n_ipo, launch);
        default: return nullptr;
    }
}

// picks the variant the geometry runs (fast detect and pairing only where they exist)
template <bool WRITE_C>
const void* fused16_select(const Geometry& g, hipError_t (**launch)(const FusedArgs&, const LaunchShape&, hipStream_t))
{
    if constexpr (!WRITE_C) {
        const bool fast = g.fast_detect && g.n_ipo >= 16;
        if (g.paired && fast) return fused16_fn<false, true, true>(g.n_ant, g.n_ipo, launch);
        if (g.paired) return fused16_fn<false, false, true>(g.n_ant, g.n_ipo, launch);
        if (fast) return fused16_fn<false, true, false>(g.n_ant, g.n_ipo, launch);
    }
    return fused16_fn<WRITE_C, false, false>(g.n_ant, g.n_ipo, launch);
}

// geometries with an instantiation: 64 antennas with n_ipo 2..64, 16 / 32 / 100 / 128 antennas with n_ipo 2 and 32
bool use16(const Geometry& g)
{
    if (g.n_ant == 64) return g.n_ipo == 2 || g.n_ipo == 4 || g.n_ipo == 8 || g.n_ipo == 16 || g.n_ipo == 32 || g.n_ipo == 64;
    return (g.n_ant == 16 || g.n_ant == 32 || g.n_ant == 100 || g.n_ant == 128) && (g.n_ipo == 2 || g.n_ipo == 32);
}
int ksteps16(const Geometry& g) { return g.n_ant > 64 ? 2 : 1; }
int interleaved(const Geometry& g) { return DSABF_INTERLEAVE && g.n_beams % 64 == 0; }  // every wave owns 64 whole beams

int ilog2_exact(int v)
{
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) s++;
    return s;
}

template <bool WRITE_C>
hipError_t dispatch_fused(const Geometry& g, const FusedArgs& args, const LaunchShape& ls, hipStream_t s)
{
    hipError_t (*launch)(const FusedArgs&, const LaunchShape&, hipStream_t) = nullptr;
    if (!use16(g) || !fused16_select<WRITE_C>(g, &launch) || !launch) return hipErrorInvalidValue;
    return launch(args, ls, s);
}

}  // namespace

size_t weight_image_bytes(const Geometry& g)
{
    return (size_t)g.n_freq * g.n_btiles * 2 * 4 * ksteps16(g) * 64 * 16;  // [f][ct16][rho][s][k-step][lane] x 16 B
}

bool pairing_supported(const Geometry& g) { return DSABF_PAIRED && use16(g); }
size_t weight_pair_image_bytes(const Geometry& g)
{
    return pairing_supported(g) ? (size_t)g.n_freq * g.n_btiles * 3 * ksteps16(g) * 64 * 16 : 0;
}

bool fused_supported(const Geometry& g, const char** why)
{
    const char* dummy;
    if (!why) why = &dummy;
    if (g.n_beams <= 0 || g.n_beams % 32) { *why = "n_beams must be a positive multiple of 32"; return false; }
    if (g.n_ant % 4) { *why = "N_ANTENNAS must be divisible by 4"; return false; }
    if (g.n_ipo < 16 && g.n_time % 16) { *why = "n_out_per_gemm * n_pol * n_avg must be a multiple of 16"; return false; }
    if (use16(g)) return true;
    *why = "no kernel instantiation for this (n_ant, n_pol*n_avg); supported: n_ant 16/32/100/128 with n_ipo 2/32, "
           "n_ant 64 with n_ipo 2/4/8/16/32/64";
    return false;
}

LaunchShape fused_launch_shape(const Geometry& g, int n_units, int n_cus)
{
    LaunchShape ls{};
    const long long S = (long long)n_units * g.n_time;
    long long rows;
    int cpg = 1;  // chunks per output group: a workgroup's chunk range must cover whole groups
    if (g.n_ipo >= 16) {
        const long long groups = (S / g.n_ipo + 3) / 4;  // a lane group carries one output: 4 outputs advance together
        rows = groups * 4 * g.n_ipo;
        if (4 * g.n_ipo > kRowsPerChunk) cpg = 4 * g.n_ipo / kRowsPerChunk;
    } else {
        rows = (S + 15) / 16 * 16;                       // 16-sample runs
    }
    ls.chunks_total = (int)((rows + kRowsPerChunk - 1) / kRowsPerChunk);
    ls.chunks_total = (ls.chunks_total + cpg - 1) / cpg * cpg;
    const int base = g.n_freq * g.n_bgroups;
    // Time splits per frequency.  Measured (profiles/r01_variants_log.txt): the kernel is fastest with ~16-32 chunks
    // per workgroup (long enough to amortise the weight-fragment load and the prologue, short enough that the tail
    // of the launch is fine-grained); small launches still get ~2 workgroups per resident slot, but never fewer
    // than 2 chunks each.
    const int groups_avail = ls.chunks_total / cpg;
    const int max_split = groups_avail >= 2 ? groups_avail / 2 : 1;
    int want = (groups_avail + 10) / 20;                                       // ~20 chunk-groups per workgroup
    // short windows (n_ipo < 16) are store-bound: fewer, longer workgroups measured better (C2: 2 per CU 0.54 of the
    // HBM peak, 8 per CU 0.49); the MFMA-bound shapes want ~2 resident sets of 4
    const int target_wgs_per_cu = g.n_ipo < 16 ? 2 : 2 * (16 / kWaves16);
    int want_fill = (target_wgs_per_cu * n_cus + base - 1) / base;            // enough workgroups to fill the chip
    if (want_fill > max_split) want_fill = max_split;
    if (want < want_fill) want = want_fill;
    if (const char* e = getenv("DSABF_TSPLIT")) want = atoi(e);  // tuning override (time splits per frequency)
    if (want < 1) want = 1;
    if (want > ls.chunks_total / cpg) want = ls.chunks_total / cpg;
    ls.n_tsplit = want;
    ls.grid = base * ls.n_tsplit;
    ls.block = kThreads16;
    ls.lds_bytes = 2 * ksteps16(g) * kRowsPerChunk * 128;  // double buffer x k-step planes x 128 rows x (64 re | 64 im)
    return ls;
}

static FusedArgs make_args(const Geometry& g, const void* d_image, const void* d_packed, int n_units, float* d_out,
                           const LaunchShape& ls)
{
    FusedArgs a{};
    a.in = static_cast<const uint8_t*>(d_packed);
    a.wimg = static_cast<const v4i*>(d_image);
    a.out = d_out;
    a.n_freq = g.n_freq;
    a.n_beams = g.n_beams;
    a.n_btiles = g.n_btiles;
    a.n_bgroups = g.n_bgroups;
    a.T = g.n_time;
    a.t_shift = ilog2_exact(g.n_time);
    a.S = (unsigned)((long long)n_units * g.n_time);
    a.chunks_total = ls.chunks_total;
    a.n_tsplit = ls.n_tsplit;
    a.interleave = interleaved(g);
    return a;
}

hipError_t launch_fused(const Geometry& g, const void* d_image, const void* d_pair_image, const void* d_packed,
                        int n_units, float* d_out, int n_cus, hipStream_t s)
{
    if (n_units <= 0) return hipSuccess;
    if ((long long)n_units * g.n_time > 0x7fffffffLL / 2) return hipErrorInvalidValue;
    if (g.paired && !(pairing_supported(g) && d_pair_image)) return hipErrorInvalidValue;
    const LaunchShape ls = fused_launch_shape(g, n_units, n_cus);
    const FusedArgs a = make_args(g, g.paired ? d_pair_image : d_image, d_packed, n_units, d_out, ls);
    return dispatch_fused<false>(g, a, ls, s);
}

hipError_t launch_gemm_only(const Geometry& g, const void* d_image, const void* d_packed, float* d_c, int n_cus,
                            hipStream_t s)
{
    Geometry gg = g;
    gg.paired = false;  // the stage-parity path always runs the general kernel on the general image
    const LaunchShape ls = fused_launch_shape(gg, 1, n_cus);
    const FusedArgs a = make_args(gg, d_image, d_packed, 1, d_c, ls);
    return dispatch_fused<true>(gg, a, ls, s);
}

hipError_t launch_weight_relayout(const Geometry& g, const int8_t* d_w, void* d_image, void* d_pair_image, int* d_bad,
                                  hipStream_t s)
{
    const size_t total = weight_image_bytes(g) / 16;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (pairing_supported(g) && d_pair_image) {
        const size_t n_fa = (size_t)g.n_freq * g.n_ant;
        int pgrid = (int)((n_fa * (g.n_beams / 2) + 255) / 256);
        if (pgrid > 4096) pgrid = 4096;
        hipLaunchKernelGGL(pair_check_kernel, dim3(pgrid), dim3(256), 0, s, d_w, n_fa, g.n_beams, d_bad + 1);
        int rgrid = (int)((weight_pair_image_bytes(g) / 16 + 255) / 256);
        if (rgrid > 4096) rgrid = 4096;
        hipLaunchKernelGGL(weight_relayout16p_kernel, dim3(rgrid), dim3(256), 0, s, d_w, static_cast<v4i*>(d_pair_image),
                           g.n_freq, g.n_ant, g.n_beams, ksteps16(g), interleaved(g));
    }
    hipLaunchKernelGGL(weight_relayout16_kernel, dim3(grid), dim3(256), 0, s, d_w, static_cast<v4i*>(d_image), g.n_freq,
                       g.n_ant, g.n_beams, ksteps16(g), interleaved(g), d_bad);
    return hipGetLastError();
}

hipError_t launch_expand(const void* d_in, size_t nbytes, void* d_out, hipStream_t s)
{
    const size_t n_vec = nbytes / 16;
    if (n_vec == 0) return hipSuccess;
    size_t grid = (n_vec + 255) / 256;
    if (grid > 65536) grid = 65536;  // measured best on MI355X (4.4 TB/s algorithmic vs 3.8 at 2048 blocks)
    hipLaunchKernelGGL(expand_kernel, dim3((unsigned)grid), dim3(256), 0, s, static_cast<const v4i*>(d_in),
                       static_cast<v4i*>(d_out), n_vec);
    return hipGetLastError();
}

hipError_t launch_dedisperse(const Geometry& g, const float* d_out_unit, float* d_ded, hipStream_t s)
{
    hipLaunchKernelGGL(dedisperse_kernel, dim3((g.n_beams + 63) / 64), dim3(64), 0, s, d_out_unit, d_ded, g.n_freq,
                       g.n_beams);
    return hipGetLastError();
}

hipError_t launch_dedisperse_dm(const Geometry& g, const float* d_series, int n_t, const int* d_delays, int n_dm,
                                int n_t_out, float* d_out, hipStream_t s)
{
    if (n_dm <= 0 || n_t_out <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_dm + kDmBlock - 1) / kDmBlock), (unsigned)((n_t_out + kDmTb - 1) / kDmTb),
                    (unsigned)((g.n_beams + kDmThreads - 1) / kDmThreads));
    if (grid.y > 65535u || grid.z > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(dedisperse_dm_kernel, grid, dim3(kDmThreads), 0, s, d_series, d_delays, d_out, n_t, g.n_freq,
                       g.n_beams, n_t_out, n_dm);
    return hipGetLastError();
}

const char* fused_kernel_name(const Geometry& g, char* buf, size_t n)
{
    snprintf(buf, n, "dsabf::fused16_kernel<ANT=%d,NIPO=%d%s%s> (v_mfma_i32_16x16x64_i8)", g.n_ant, g.n_ipo,
             (g.fast_detect && g.n_ipo >= 16) ? ",FAST" : "", g.paired ? ",PAIRED" : "");
    return buf;
}

int fused_vgprs(const Geometry& g)
{
    hipFuncAttributes attr{};
    const void* fn = nullptr;
    hipError_t (*launch)(const FusedArgs&, const LaunchShape&, hipStream_t) = nullptr;
    if (use16(g)) fn = fused16_select<false>(g, &launch);
    if (!fn || hipFuncGetAttributes(&attr, fn) != hipSuccess) return -1;
    return attr.numRegs;
}

}  // namespace dsabf
