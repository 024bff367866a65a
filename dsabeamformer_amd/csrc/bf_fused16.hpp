// bf_fused16.hpp -- the fused expand + complex int8 GEMM + detect kernel (fused16_kernel) as a template, so that its
// instantiations can be spread over several translation units that compile in parallel (bf_fused16_*.hip: one per antenna
// class).  Internal to libdsabf.so; design notes in DESIGN.md section 3 and at the top of bf_kernels.hip.
#pragma once
#include "bf_kernels.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#ifndef DSABF_CLOCKPROBE
#define DSABF_CLOCKPROBE 0 // the ONE compile-time switch left in this kernel: a diagnostic build that tools/clock_probe.sh makes beside
#endif                     // the product (-DDSABF_CLOCKPROBE=1: every workgroup overwrites out[blockIdx.x] with its in-kernel shader clock
                           // in GHz, s_memtime / s_memrealtime around the chunk loop; results invalid).  The experiment arms of rounds
                           // 1-5 (5 / 6 MFMAs per conjugate-pair tile, the 4-fragment general image, vector chunk addressing,
                           // sign-extended nibbles in the deep classes, the timing ablations, -DDSABF_WAVES / _NS / _OCC16 builds) were
                           // measured, lost, and are gone: profiles/r0[1-5]_variants_log.txt, docs/LOG_r06.md.

namespace dsabf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// Detect variants (bf_config.detect_mode): how one sample's power enters the running sum.
constexpr int kDetCanonical = 0;   // xx = x*x; yy = y*y; p = xx + yy; acc += p        (6 VALU ops per complex sample)
constexpr int kDetFast = 1;        // acc = fma(d, d, acc) on the unscaled integers        (4)
constexpr int kDetContracted = 2;  // yy = y*y; p = fma(x, x, yy); acc += p  (nvcc -fmad)  (5)

// Antenna classes (template parameter AIN).  A positive value is a compile-time antenna count: since round 6 only 100 (BASELINE
// config 5, whose 100-byte rows are dword-staged: 3 - 7 % faster than the run-time class that covers them); the compile-time
// classes of 64 / 128 / 192 / 256 antennas measured inside the box noise of the run-time ones and were folded into them
// (profiles/r06_class_fold_ab.txt, r06_ab_fold_c3.txt).  The negative classes take the count from FusedArgs::n_ant at run time:
constexpr int kAntK1P16 = -1;  // <= 64 antennas, n_ant % 16 == 0: one k-step, 16-byte staging pieces
constexpr int kAntK1P4 = -2;   // <= 64 antennas, n_ant % 4 == 0:  one k-step, 4-byte staging pieces
constexpr int kAntK2P16 = -3;  // 65..128 antennas, n_ant % 16 == 0: two k-steps
constexpr int kAntK2P4 = -4;   // 65..128 antennas, n_ant % 4 == 0
constexpr int kAntK4P16 = -5;  // 193..256 antennas, n_ant % 16 == 0: four k-steps (the "deep" classes, round 4)
constexpr int kAntK3P16 = -6;  // 129..192 antennas, n_ant % 16 == 0: three k-steps
constexpr int kAntK4P4 = -7;   // 193..256 antennas, n_ant % 4 == 0: four k-steps, 4-byte staging pieces (round 5)
constexpr int kAntK3P4 = -8;   // 129..192 antennas, n_ant % 4 == 0

// Weight fragments per tile in the two images (weight_relayout16_kernel / weight_relayout16p_kernel, bf_kernels.hip):
constexpr int kGeneralComps = 3;   // Wr, -Wi, Wi: the real row multiplies (Vr | Vi) by (Wr | -Wi), the imaginary row by (Wi | Wr) -- Wr serves both
constexpr int kPairComps = 2;      // Wr, Wi: the conjugate-pair kernel forms +-P2, +-P4 on the VALU

constexpr unsigned kMagicBits = 0x4B400000u;        // float 12582912 = 1.5 * 2^23
constexpr float kMagic = 12582912.0f;
constexpr float kAlpha = (float)(1.0 / 127.0);      // h_inv_max_value.x, src/beamformer.cu:191
constexpr float kAlpha16 = kAlpha * 0.0625f;        // exact (power-of-two scaling)
constexpr float kNegMagicAlpha16 = -(kMagic * kAlpha16);
constexpr float kNegMagicAlpha = -(kMagic * kAlpha);   // the deep classes stage TRUE nibbles (|16 n| would leave the seed's range)
static_assert((double)kMagic * (double)kAlpha == (double)(kMagic * kAlpha), "K * alpha must be exactly representable");
static_assert((double)kMagic * (double)kAlpha16 == (double)(kMagic * kAlpha16),
              "K * alpha/16 must be exactly representable for the single-fma conversion");

struct FusedArgs {
    const uint8_t* __restrict__ in;  // packed voltages [unit][f][t][a]
    const v4i* __restrict__ wimg;    // weight fragment image
    float* __restrict__ out;         // detected [unit*n_out + o][f][b]   (WRITE_C: c[f][t][b]{re,im})
    int n_freq, n_beams, n_bgroups;
    int n_ctiles;                    // 16-beam column tiles = ceil(n_beams / 16) (the last one may be partly filled)
    int n_ptiles;                    // conjugate-pair tiles of 16 base beams = n_beams / 32 (paired kernel only)
    int n_ant;                       // antennas per time sample (the run-time antenna classes read it; AIN > 0 ignores it)
    int T;                           // time samples per gemm-unit
    int t_shift;                     // log2(T) if T is a power of two, else -1
    unsigned S;                      // total time samples per frequency in this launch (n_units * T)
    int chunks_total;                // 128-sample chunks per frequency in this launch
    int n_tsplit;                    // workgroups along time
    int interleave;                  // MFMA column tiles per wave if beams are dealt to them round-robin (beam_of_tile), else 0
    // run-time accumulation window (template NIPO = 0): n_ipo, samples per stream = rt_kout * rt_L, windows per stream, chunks per stream group
    int rt_L, rt_Ls, rt_kout, rt_cpg;
};

// Which beam MFMA column c of MFMA column tile `tile` computes.  A wave owns 16 * NS consecutive beams (paired: 8 * NS base
// beams and their mirror images) in `per` MFMA column tiles (per = NS, paired NS / 2).  Interleaved (per > 0; n_beams a multiple
// of 16 * NS): beam = first + per * c + t, so a lane's results of one output are `per` consecutive floats (paired: per + per)
// -> 16-byte (8-byte for 2) stores, whole 128-byte lines per store instruction, instead of scattered 64-byte rows.
// per = 0: tile t = beams first + 16 t + c.
__host__ __device__ inline int beam_of_tile(int per, int tile, int c)
{
    if (!per) return tile * 16 + c;
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
//              (two-k-step classes, where the beams allow: 8 column tiles = 128 beams per wave for the conjugate-pair
//              kernel -- template parameter NS --, else 8 waves = 512 beams per workgroup -- WAVES; bf_kernels.hip)
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
constexpr int kWavesWide16 = 8;              // ... of the two-k-step classes where the beam count allows (fused_wg_waves)
constexpr int kColTiles16 = 4;               // 16-beam column tiles (output slots) per wave
constexpr int kColTilesWide16 = 8;           // ... of the two-k-step conjugate-pair kernels where the beams allow (fused_col_tiles)

// PAIRED: the steering weights of beam B-1-b are the complex conjugates of those of beam b for every (frequency,
// antenna) -- true for any beam set that is symmetric about the boresight, e.g. the reference's linear fan and 16x16
// grid (checked exactly by pair_check_kernel when the weights are set).  Then with the four REAL K=64 products
//   P1 = sum Wr*Vr, P2 = sum Wi*Vi, P3 = sum Wr*Vi, P4 = sum Wi*Vr        (one 16x16x64 MFMA each)
// C(b) = (P1 - P2) + j(P3 + P4) and C(B-1-b) = (P1 + P2) + j(P3 - P4): two beams for the MFMA work of one, exact in
// int32 (the +-P2 / +-P4 are 4 integer VALU ops per sample pair; P1 and P3 carry the float seed, P2 and P4 start at 0).
//
// AIN = antenna class (above).  More than 64 antennas are two k-steps of 64: the LDS chunk image
// becomes two 128-row planes (antennas 0-63 | 64-127), every product is a chain of two MFMAs, and the detect -- whose
// cost does not depend on the antenna count -- is amortised over twice the MACs.  100 antennas run as 128 with zero
// weights behind antenna 99; their packed rows (100 B) are only dword-aligned, so they are staged in 4-byte pieces.
template <int AIN>
constexpr int ant_ksteps()
{
    if (AIN > 0) return (AIN + 63) / 64;
    return (AIN == kAntK4P16 || AIN == kAntK4P4) ? 4 : (AIN == kAntK3P16 || AIN == kAntK3P4) ? 3 : (AIN == kAntK2P16 || AIN == kAntK2P4) ? 2 : 1;
}
template <int AIN>
constexpr bool ant_two_ksteps() { return ant_ksteps<AIN>() == 2; }
// Three and four k-steps (129 ... 256 antennas; round 4): the same weight-stationary kernel, always on 8-wave workgroups (their
// LDS image -- 2 buffers x 3 | 4 planes of 16 KiB -- leaves one workgroup per CU).  A wave owns TWO output slots (general kernel:
// 24 weight registers per k-step; four slots = two pair tiles for the conjugate-pair kernel), its voltages are staged as true
// nibble values (sign-extended once per workgroup: the 16 x nibble operands of the shallower classes would carry |16 n| past the
// 2^22 the float seed covers), and the detect -- whose cost does not depend on the antenna count -- is amortised over 3 - 4 x
// the MACs of the 64-antenna kernel.
template <int AIN>
constexpr bool ant_deep() { return ant_ksteps<AIN>() > 2; }

// Waves per SIMD the register allocation is held to (= resident workgroups per CU).  Two k-steps: 2 (their 64 KiB of LDS
// allow no more).  One k-step: 3 -- except where 4 fit without a spill: since round 3's 3-fragment weight image the general
// kernel of the 16-byte-staged classes needs 124 VGPRs for n_ipo 8 / 16 / 32 (the conjugate-pair kernel always did), and the
// fourth resident workgroup is worth 2 % (canonical) to 3 % (contracted) on C3 (profiles/r03_ab_c3_general_occ4.txt).  The
// other window lengths and the dword-staged class would spill at 128 and stay at 3.
template <int AIN, int NIPO, bool WRITE_C, int NS = kColTiles16>
constexpr int fused_min_waves()
{
    if (NS == 8 || ant_ksteps<AIN>() >= 2) return 2;
    if (AIN == kAntK1P16 && (NIPO == 8 || NIPO == 16 || NIPO == 32) && !WRITE_C) return 4;
    return 3;   // 168 VGPRs
}

//
// WAVES = waves per workgroup (4, or 8 where fused_wg_waves() in bf_kernels.hip says so): a workgroup stages one frequency's
// voltages for 64 * WAVES beams.
// NS = 16-beam output slots per wave (4, or 8 where fused_col_tiles() says so): a wave's LDS fragment reads feed NS (paired: NS / 2)
// MFMA column tiles.
template <int AIN, int NIPO, bool WRITE_C, int MODE = kDetCanonical, bool PAIRED = false, int WAVES = kWaves16, int NS = kColTiles16>
__global__ __launch_bounds__(64 * WAVES, (fused_min_waves<AIN, NIPO, WRITE_C, NS>())) void fused16_kernel(FusedArgs a)
{
    constexpr int THREADS = 64 * WAVES;
    constexpr bool FAST = MODE == kDetFast;
    constexpr bool CONTRACTED = MODE == kDetContracted;
    // NIPO = 0: the accumulation window is a RUN-TIME value (any n_pol * n_avg that is not one of the compile-time windows; round 4).
    // A lane group's 32 rows of a chunk are then rt_kout whole windows back to back (or one window over rt_cpg chunks), padded
    // to 32 rows -- the stream scheme of fusedg_kernel in this kernel's weight-stationary loop; window starts and ends are
    // wave-uniform run-time flags per row.
    constexpr bool RTW = NIPO == 0;
    constexpr int MAPN = RTW ? 32 : NIPO;                // row <-> stream mapping and LDS swizzle: as for windows >= 32
    static_assert(!FAST || ((NIPO >= 16 || RTW) && !WRITE_C), "the fast detect exists for n_ipo >= 16 only");
    static_assert(!RTW || (WAVES == kWaves16 && NS == kColTiles16 && !ant_deep<AIN>()), "run-time windows: the plain launch shape");
    static_assert(!(PAIRED && WRITE_C), "the stage-parity path always runs the general kernel");
    static_assert(AIN >= kAntK3P4 && AIN != 0 && (AIN < 0 || AIN % 4 == 0) && AIN <= 256, "antenna class");
    static_assert(!ant_deep<AIN>() || (WAVES == 8 && NIPO >= 16 && !WRITE_C), "deep classes: 8 waves, long windows");
    // the deep classes' operands count in units of the nibble value itself, not 16 x, staged as OFFSET nibbles v + 8 in [0, 15] with
    // the correction in the accumulator seeds (sign-extended nibbles cost 9 VALU per dword and the pipe holds a lower clock on them:
    // profiles/r04_ubench_encoding.txt)
    constexpr bool OFFSET_NIB = ant_deep<AIN>();
    constexpr float kA = OFFSET_NIB ? kAlpha : kAlpha16;   // accumulator unit -> alpha
    constexpr float kNKA = OFFSET_NIB ? kNegMagicAlpha : kNegMagicAlpha16;
    constexpr bool RT = AIN < 0;                         // antenna count known only at run time
    constexpr int RB = 128;
    constexpr int KS = ant_ksteps<AIN>();                // k-steps of 64 antennas
    constexpr int PLANE = kRowsPerChunk * RB;            // LDS bytes of one k-step's chunk image
    constexpr int BUF = KS * PLANE;
    constexpr bool DW = RT ? (AIN == kAntK1P4 || AIN == kAntK2P4 || AIN == kAntK3P4 || AIN == kAntK4P4) : (AIN % 16) != 0;  // rows only dword-aligned: 4-byte pieces
    constexpr int PB = DW ? 4 : 16;                      // bytes per staging piece
    constexpr int AMAX = RT ? 64 * KS : AIN;             // most antennas this instantiation can meet
    const int A = RT ? a.n_ant : AIN;                    // antennas per time sample (constant-folded unless RT)
    const int PPR = A / PB;                              // pieces per time sample
    const int TOTALP = kRowsPerChunk * PPR;              // pieces per chunk
    constexpr int TOTALP_MAX = kRowsPerChunk * (AMAX / PB);
    constexpr int NT = PAIRED ? NS / 2 : NS;             // MFMA column tiles per wave (a paired tile feeds 2 slots)
    constexpr bool LONG = NIPO >= 16;
    constexpr int L = LONG ? NIPO : 16;                  // samples per stream
    constexpr int LR = (NIPO >= 32 || RTW) ? 32 : 16;    // stream rows held by one chunk
    constexpr int CPG = L > 32 ? L / 32 : 1;             // chunks per group of 4 streams
    constexpr int PPT = (TOTALP_MAX + THREADS - 1) / THREADS;  // pieces per thread per chunk (2; 4; 13 for 100 antennas)
    using stage_t = std::conditional_t<DW, int, v4i>;

    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 buffers x KS planes x 128 rows x 128 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4;   // lane group = stream within the tile / k-block of the operands
    const int c16 = lane & 15;  // column within a 16-beam tile / A row

    int f, bg, ts;
    decode_block(a, f, bg, ts);
    const int cpg_rt = RTW ? a.rt_cpg : CPG;            // (constant-folded unless the window is a run-time value)
    const int units_total = a.chunks_total / cpg_rt;
    const int c_begin = (int)(((long long)units_total * ts) / a.n_tsplit) * cpg_rt;
    const int c_end = (int)(((long long)units_total * (ts + 1)) / a.n_tsplit) * cpg_rt;

    // ---- which beams this lane produces, and the weight fragments ------------------------------------------
    int slot_beam[NS];                                    // beam index of output slot s (>= n_beams: none)
    constexpr int NPC = kPairComps;                       // paired fragments per tile: Wr, Wi
    constexpr int NGC = kGeneralComps;                    // general fragments per tile: Wr, -Wi, Wi (the im row's Wr IS comp 0)
    v4i bw[NT][PAIRED ? NPC : NGC][KS];                   // general: [ct][Wr, -Wi, Wi][k-step]; paired: [pct][Wr, Wi][k-step]
    bool wave_active;
    if constexpr (PAIRED) {
        const int n_pct = a.n_ptiles;                     // pair tiles of 16 base beams = n_beams / 32
        const int pct0 = (bg * WAVES + wave) * NT;
        wave_active = pct0 < n_pct;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const int bb = beam_of_tile(a.interleave ? NT : 0, pct0 + t, c16);  // base beam (< n_beams / 2)
            const bool ok = pct0 + t < n_pct;
            slot_beam[2 * t] = ok ? bb : a.n_beams;
            slot_beam[2 * t + 1] = ok ? a.n_beams - 1 - bb : a.n_beams;
#pragma unroll
            for (int comp = 0; comp < NPC; comp++)
#pragma unroll
                for (int h = 0; h < KS; h++)
                    bw[t][comp][h] =
                        ok ? a.wimg[((((size_t)f * n_pct + pct0 + t) * NPC + comp) * KS + h) * 64 + lane] : v4i{0, 0, 0, 0};
        }
    } else {
        const int n_ctiles = a.n_ctiles;
        const int ct0 = (bg * WAVES + wave) * NT;      // first 16-beam column tile of this wave
        wave_active = ct0 < n_ctiles;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const bool ok = ct0 + t < n_ctiles;
            slot_beam[t] = ok ? beam_of_tile(a.interleave ? NT : 0, ct0 + t, c16) : a.n_beams;
#pragma unroll
            for (int k = 0; k < NGC; k++)
#pragma unroll
                for (int h = 0; h < KS; h++)
                    bw[t][k][h] =
                        ok ? a.wimg[((((size_t)f * n_ctiles + ct0 + t) * NGC + k) * KS + h) * 64 + lane] : v4i{0, 0, 0, 0};
        }
    }

    v4i kc = {(int)kMagicBits, (int)kMagicBits, (int)kMagicBits, (int)kMagicBits};
    asm volatile("" : "+v"(kc));
    const v4i kzero = {0, 0, 0, 0};
    // Offset nibbles: sum W (v + 8) = sum W v + 8 sum W, so the chains start at seed - 8 * (this lane's column sum of the weight
    // fragment they multiply): summed here from the B fragments themselves (16 antennas per lane and k-step, the four lane groups
    // hold the other 48), once per workgroup.  General: sd[t][0] -> re = Wr Vr - Wi Vi, sd[t][1] -> im = Wi Vr + Wr Vi;
    // paired: sd[t][0] -> P1, P3 (Wr), sd[t][1] -> P2, P4 (Wi, no magic).
    [[maybe_unused]] v4i sd[OFFSET_NIB ? NT : 1][2];
    if constexpr (OFFSET_NIB) {
        auto colsum = [&](const v4i (&w)[KS]) {
            int sacc = 0;
#pragma unroll
            for (int h = 0; h < KS; h++)
#pragma unroll
                for (int d = 0; d < 4; d++)
#pragma unroll
                    for (int b = 0; b < 4; b++) sacc += (int)(signed char)((unsigned)w[h][d] >> (8 * b));
            sacc += __shfl_xor(sacc, 16);
            sacc += __shfl_xor(sacc, 32);
            return sacc;
        };
#pragma unroll
        for (int t = 0; t < NT; t++) {
            int s0, s1;
            if constexpr (PAIRED) {
                s0 = (int)kMagicBits - 8 * colsum(bw[t][0]);
                s1 = -8 * colsum(bw[t][1]);
            } else {
                const int wr = colsum(bw[t][0]), nwi = colsum(bw[t][1]), wi = colsum(bw[t][2]);
                s0 = (int)kMagicBits - 8 * (wr + nwi);
                s1 = (int)kMagicBits - 8 * (wi + wr);
            }
            sd[t][0] = v4i{s0, s0, s0, s0};
            sd[t][1] = v4i{s1, s1, s1, s1};
            asm volatile("" : "+v"(sd[t][0]), "+v"(sd[t][1]));
        }
    }

    // ---- staging (the chunk's 128 samples are contiguous in time for n_ipo <= 32) ------------------------------
    auto run_sample0 = [&](int c, int run) -> unsigned {   // first global sample of stream-run `run` of chunk c
        if constexpr (RTW)
            return (4u * (unsigned)(c / cpg_rt) + (unsigned)run) * (unsigned)a.rt_Ls + 32u * (unsigned)(c % cpg_rt);
        else if constexpr (NIPO >= 32)
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
    const bool fast_addr = !RTW && ((unsigned)a.T % SPAN) == 0;
    // Piece k of this thread is piece pc = tid + 256 k of the chunk: row pc / PPR, position pc % PPR.  Its byte offset
    // from the chunk's first sample is PB * pc (rows are PPR * PB bytes and consecutive) -- except for n_ipo = 64, whose
    // chunk rows are four runs of 32 samples, 64 apart.  The 16*im image sits 4 pieces away from the 16*re image (after it
    // in plane 0, before it in plane 1), and the swizzle only XORs the 3 piece bits, so its LDS offset is the re offset ^ 64.
    [[maybe_unused]] unsigned lane_off64[NIPO == 64 ? PPT : 1];
    int lds_re[PPT];                      // LDS byte offset (inside one buffer) of the piece's 16*re image
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int pc = tid + k * THREADS;
        const int row = (pc / PPR) % kRowsPerChunk, pi = pc % PPR;   // (% keeps the unused tail pieces in range)
        if constexpr (NIPO == 64) lane_off64[k] = (unsigned)((row / LR) * L + (row % LR)) * A + pi * PB;
        const int blk = DW ? pi / 4 : pi;                            // 16-antenna block of the piece
        const int h = blk / 4, kp = blk % 4, sub = DW ? 4 * (pi % 4) : 0;
        lds_re[k] = h * PLANE + row * RB + 16 * swz16<MAPN>(kp + 4 * (h & 1), row) + sub;  // plane 1: halves swapped
    }
    auto lane_off = [&](int k) -> unsigned {
        if constexpr (NIPO == 64)
            return lane_off64[k];
        else
            return (unsigned)PB * (unsigned)(tid + k * THREADS);
    };
    auto piece_live = [&](int k) { return (!RT && TOTALP_MAX % THREADS == 0) || (tid + k * THREADS < TOTALP); };
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
                if (valid && piece_live(k)) stage[k] = *reinterpret_cast<const stage_t*>(base + lane_off(k));   // (nontemporal: +-0.5 %, r02_variants_log)
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int pc = tid + k * THREADS;       // (run-time antenna classes: a real division per piece, but this is
            const int row = (pc / PPR) % kRowsPerChunk, pi = pc % PPR;   //  the path of small DEBUG-style gemm-units only)
            const unsigned s0 = run_sample0(c, row / LR) + (unsigned)(row % LR);
            stage[k] = stage_t{};
            bool row_ok = s0 < a.S;
            if constexpr (RTW) row_ok = row_ok && 32u * (unsigned)(c % cpg_rt) + (unsigned)(row % LR) < (unsigned)a.rt_Ls;   // not a padding row
            if (row_ok && piece_live(k)) {
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
                if constexpr (OFFSET_NIB) {          // (the deep classes' encodings: see the 16-byte pieces below)
                    const unsigned x = w ^ 0x88888888u;
                    *reinterpret_cast<int*>(buf + lds_re[k]) = (int)((x >> 4) & 0x0F0F0F0Fu);
                    *reinterpret_cast<int*>(buf + (lds_re[k] ^ 64)) = (int)(x & 0x0F0F0F0Fu);
                } else {
                    *reinterpret_cast<int*>(buf + lds_re[k]) = (int)(w & 0xF0F0F0F0u);
                    *reinterpret_cast<int*>(buf + (lds_re[k] ^ 64)) = (int)((w << 4) & 0xF0F0F0F0u);
                }
            } else {
                v4i re, im;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned w = (unsigned)stage[k][d];
                    if constexpr (OFFSET_NIB) {   // v + 8 = the nibble's bits with the top one flipped
                        const unsigned x = w ^ 0x88888888u;
                        re[d] = (int)((x >> 4) & 0x0F0F0F0Fu);
                        im[d] = (int)(x & 0x0F0F0F0Fu);
                    } else {
                        re[d] = (int)(w & 0xF0F0F0F0u);
                        im[d] = (int)((w << 4) & 0xF0F0F0F0u);
                    }
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
            // Nontemporal: the powers are written once and read by nobody on this GPU before the D2H / gather.  Every
            // store instruction covers whole 128-byte lines (16 lanes x 16 B, or 16 x 8 B), so streaming them past L2
            // costs nothing at C3 / C5 (+0.3 %) and lifts the store-bound DEBUG geometry from 0.60 to 0.73 of 8 TB/s
            // (profiles/r02_variants_log.txt).  The scalar stores of non-interleaved tiles cover partial lines: plain.
            if constexpr (PAIRED && NS == 2) {   // one pair tile per wave (deep classes, beams not in groups of 512): beam bb and its mirror
                row[slot_beam[0]] = x[0];
                row[slot_beam[1]] = x[1];
            } else if constexpr (PAIRED && NS == 4) {   // slots 0, 2 = base beams bb, bb + 1; slots 1, 3 = their mirrors B-1-bb, B-2-bb
                __builtin_nontemporal_store(v2f{x[0], x[2]}, reinterpret_cast<v2f*>(row + slot_beam[0]));
                __builtin_nontemporal_store(v2f{x[3], x[1]}, reinterpret_cast<v2f*>(row + slot_beam[3]));
            } else if constexpr (PAIRED) {       // NS == 8: four base beams ascending, their four mirrors descending
                __builtin_nontemporal_store(v4f{x[0], x[2], x[4], x[6]}, reinterpret_cast<v4f*>(row + slot_beam[0]));
                __builtin_nontemporal_store(v4f{x[7], x[5], x[3], x[1]}, reinterpret_cast<v4f*>(row + slot_beam[7]));
            } else if constexpr (NS == 2) {       // two neighbouring beams per lane
                __builtin_nontemporal_store(v2f{x[0], x[1]}, reinterpret_cast<v2f*>(row + slot_beam[0]));
            } else {
#pragma unroll
                for (int q = 0; q < NS; q += 4)
                    __builtin_nontemporal_store(v4f{x[q], x[q + 1], x[q + 2], x[q + 3]}, reinterpret_cast<v4f*>(row + slot_beam[q]));
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
            // run-time window (RTW): where the windows of this lane group's stream start and end in the current row tile, the
            // sums that ended there, and the stream-relative index of their outputs -- wave-uniform
            [[maybe_unused]] bool rt_st[4] = {false, false, false, false}, rt_en[4] = {false, false, false, false};
            [[maybe_unused]] unsigned rt_o[4] = {0, 0, 0, 0};
            [[maybe_unused]] float rt_x[4][NS];
            [[maybe_unused]] int rt_m = 0;            // position of the next row inside its window
            [[maybe_unused]] unsigned rt_oq = 0;      // windows of the stream that ended before the next row
            if constexpr (RTW) {
                const unsigned p0 = 32u * (unsigned)(c % cpg_rt);
                rt_m = (int)(p0 % (unsigned)a.rt_L);
                rt_oq = p0 / (unsigned)a.rt_L;
            }
            // detect + accumulate the 4 samples (fr, fi: accumulator bit patterns K + 16 n) of output slot sl
            auto detect = [&](const int t8, const v4f fr, const v4f fi, const int sl) {
                // stream position of this tile's rows and whether it starts / ends an output
                const int gi = (NIPO >= 32) ? 0 : (t8 >> 2);          // group inside the chunk (L = 16)
                const int q4 = (NIPO >= 32) ? (32 * (c % CPG) + 4 * t8) : 4 * (t8 & 3);  // position of register 0
                const unsigned grp = (NIPO >= 32) ? (unsigned)(c / CPG) : (2u * (unsigned)c + gi);
                const unsigned o = 4u * grp + (unsigned)g4;           // this lane's stream (output index if LONG)
                const int beam = slot_beam[sl];
                if constexpr (WRITE_C && RTW) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const unsigned pos = 32u * (unsigned)(c % cpg_rt) + 4u * t8 + r;
                        const unsigned sidx = (4u * (unsigned)(c / cpg_rt) + (unsigned)g4) * (unsigned)a.rt_Ls + pos;
                        if (pos < (unsigned)a.rt_Ls && sidx < a.S && beam < a.n_beams) {
                            v2f cv = {__builtin_fmaf(fr[r], kA, kNKA), __builtin_fmaf(fi[r], kA, kNKA)};
                            *reinterpret_cast<v2f*>(a.out + 2 * (((size_t)f * a.T + sidx) * a.n_beams + beam)) = cv;
                        }
                    }
                } else if constexpr (RTW) {
                    // run-time window: rt_st[r] / rt_en[r] say whether a window starts / ends at register r of this row tile
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        if constexpr (FAST) {
                            const float dr = fr[r] - kMagic, di = fi[r] - kMagic;
                            float sacc = rt_st[r] ? 0.0f : sum[sl];
                            sacc = __builtin_fmaf(dr, dr, sacc);
                            sum[sl] = __builtin_fmaf(di, di, sacc);
                            if (rt_en[r]) rt_x[r][sl] = sum[sl] * (kA * kA);
                        } else {
                            const float x = __builtin_fmaf(fr[r], kA, kNKA);
                            const float y = __builtin_fmaf(fi[r], kA, kNKA);
                            const float yy = y * y;
                            float pp;
                            if constexpr (CONTRACTED) {
                                pp = __builtin_fmaf(x, x, yy);
                            } else {
                                const float xx = x * x;
                                pp = xx + yy;
                            }
                            // (a window's first sample: 0 * sum + pp = pp, any other: 1 * sum + pp in ONE rounding = sum + pp -- the
                            //  select folded into the add; the sums are finite and >= +0)
                            sum[sl] = __builtin_fmaf(sum[sl], rt_st[r] ? 0.0f : 1.0f, pp);
                            if (rt_en[r]) rt_x[r][sl] = sum[sl];
                        }
                    }
                } else if constexpr (WRITE_C) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const unsigned sidx = o * (unsigned)L + (unsigned)(q4 + r);
                        if (o * (unsigned)L < a.S && beam < a.n_beams) {
                            v2f cv = {__builtin_fmaf(fr[r], kA, kNKA), __builtin_fmaf(fi[r], kA, kNKA)};
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
                        pend[gi][sl] = sacc * (kA * kA);
                        pend_chunk[gi] = c;
                    }
                } else {
                    float p[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float x = __builtin_fmaf(fr[r], kA, kNKA);
                        const float y = __builtin_fmaf(fi[r], kA, kNKA);
                        const float yy = y * y;
                        if constexpr (CONTRACTED) {
                            p[r] = __builtin_fmaf(x, x, yy);   // nvcc's reading of x*x + y*y (-fmad=true): mul, then fma
                        } else {
                            const float xx = x * x;
                            p[r] = xx + yy;
                        }
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
                if constexpr (!LONG && !WRITE_C && !RTW) {
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
                const int row = lds_row16<MAPN>(t8, c16);
#pragma unroll
                for (int h = 0; h < KS; h++) {  // plane 1 keeps (im | re): the two planes' staging writes then never collide
                    a0[h] = *reinterpret_cast<const v4i*>(cur + h * PLANE + row * RB + 16 * swz16<MAPN>(g4 + 4 * (h & 1), row));
                    a1[h] = *reinterpret_cast<const v4i*>(cur + h * PLANE + row * RB + 16 * swz16<MAPN>(g4 + 4 * ((h & 1) ^ 1), row));
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
                const v4i k0 = OFFSET_NIB ? sd[OFFSET_NIB ? t : 0][0] : kc;                         // chain seeds (see sd above)
                const v4i k1 = OFFSET_NIB ? sd[OFFSET_NIB ? t : 0][1] : (PAIRED ? kzero : kc);
                if constexpr (PAIRED) {
                    const v4i p1 = dot(a0, bw[t][0], k0);     // Wr*Vr + K
                    const v4i p3 = dot(a1, bw[t][0], k0);     // Wr*Vi + K
                    const v4i p2 = dot(a1, bw[t][1], k1);     // Wi*Vi   (+-P2, +-P4 on the VALU: chaining them on the MFMA
                    const v4i p4 = dot(a0, bw[t][1], k1);     // Wi*Vr    pipe -- 5 or 6 MFMAs per pair tile -- lost, r03_ab_c3_pairmfma)
                    re[0] = p1 - p2;
                    re[1] = p1 + p2;
                    im[0] = p3 + p4;
                    im[1] = p3 - p4;
                } else {
                    re[0] = dot(a1, bw[t][1], dot(a0, bw[t][0], k0));              // Wr*Vr - Wi*Vi
                    im[0] = dot(a1, bw[t][0], dot(a0, bw[t][2], k1));              // Wi*Vr + Wr*Vi
                }
            };
            auto consume = [&](const int t8, const int t, const v4i (&re)[SPS], const v4i (&im)[SPS]) {
#pragma unroll
                for (int e = 0; e < SPS; e++) {  // paired: slot 2t = beam b, slot 2t+1 = beam B-1-b
                    detect(t8, __builtin_bit_cast(v4f, re[e]), __builtin_bit_cast(v4f, im[e]), SPS * t + e);
                }
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
            for (int t8 = 0; t8 < 8; t8++) {   // (requesting tile t8+1's LDS fragments one tile early was tried: pair kernel
                v4i a0[KS], a1[KS];            //  -2 % (129 VGPRs: 3 instead of 4 waves per SIMD), general +-0, r02 variants log;
                read_frag(t8, a0, a1);         //  deep classes, requested and pinned one tile early: +-0.5 %, r04 variants log)
                if constexpr (RTW) {
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        rt_st[r] = rt_m == 0;
                        rt_en[r] = ++rt_m == a.rt_L;
                        if (rt_en[r]) {
                            rt_m = 0;
                            rt_o[r] = rt_oq++;
                            rt_en[r] = rt_o[r] < (unsigned)a.rt_kout;   // (a "window" of padding rows behind the stream's last one is nobody's)
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < NT; t++) {   // (the compiler issues the first MFMAs of all chains before the dependent
                    v4i re[SPS], im[SPS];        //  second ones by itself; forcing that order changed nothing, r02 variants log)
                    issue(a0, a1, t, re, im);
                    consume(t8, t, re, im);
                }
                if constexpr (RTW && !WRITE_C) {   // the windows that ended in this row tile: their sums leave at once
                    const unsigned sigma = 4u * (unsigned)(c / cpg_rt) + (unsigned)g4;      // this lane's stream
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (rt_en[r]) {
                            const unsigned o = sigma * (unsigned)a.rt_kout + rt_o[r];
                            if ((unsigned long long)o * (unsigned)a.rt_L < a.S) store_slots(a.out + (size_t)o * FB + (size_t)f * a.n_beams, rt_x[r]);
                        }
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

template <int AIN, int NIPO, bool WRITE_C, int MODE, bool PAIRED, int WAVES, int NS>
hipError_t launch_fused16_t(const FusedArgs& args, const LaunchShape& ls, hipStream_t s)
{
    auto kern = fused16_kernel<AIN, NIPO, WRITE_C, MODE, PAIRED, WAVES, NS>;
    if (ls.block != 64 * WAVES || (args.interleave && args.interleave != (PAIRED ? NS / 2 : NS))) return hipErrorInvalidValue;
    if (ls.lds_bytes > 48 * 1024) {   // once per kernel and device, not per launch (the two-k-step image is always 64 KiB)
        static std::atomic<unsigned> done_mask{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned bit = 1u << (dev & 31);
        if (dev >= 32 || !(done_mask.load(std::memory_order_acquire) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               ls.lds_bytes);
            if (e != hipSuccess) return e;
            done_mask.fetch_or(bit, std::memory_order_release);
        }
    }
    (void)hipGetLastError();   // clear what earlier, unrelated calls left behind: return this launch's own status
    hipLaunchKernelGGL(kern, dim3(ls.grid), dim3(ls.block), ls.lds_bytes, s, args);
    return hipGetLastError();
}


// One entry of the variant table: the kernel symbol (for hipFuncGetAttributes) and its launcher.
using fused_launch_fn = hipError_t (*)(const FusedArgs&, const LaunchShape&, hipStream_t);
struct FusedVariant {
    const void* fn = nullptr;
    fused_launch_fn launch = nullptr;
    // the instantiation's template arguments, in the kernel's own order: what the census (bf_variant_key, tests/test_census_cpu.py,
    // profiles/r06_instantiations.txt) compares with the kernel symbols of the shipped library
    int ain = 0, nipo = 0, write_c = 0, mode = 0, paired = 0, waves = 0, ns = 0;
};

template <int AIN, int NIPO, bool WRITE_C, int MODE, bool PAIRED, int WAVES, int NS = kColTiles16>
FusedVariant make_variant()
{
    return FusedVariant{reinterpret_cast<const void*>(fused16_kernel<AIN, NIPO, WRITE_C, MODE, PAIRED, WAVES, NS>),
                        launch_fused16_t<AIN, NIPO, WRITE_C, MODE, PAIRED, WAVES, NS>, AIN, NIPO, WRITE_C, MODE, PAIRED, WAVES, NS};
}

template <int AIN, int NIPO, int WAVES>
FusedVariant fused16_variant_nipo(bool write_c, int mode, bool paired)
{
    if (write_c) {   // stage parity: general kernel, canonical scale, 4 waves
        if constexpr (WAVES == kWaves16) return make_variant<AIN, NIPO, true, kDetCanonical, false, WAVES>();
        return FusedVariant{};
    }
    if constexpr (NIPO >= 16 || NIPO == 0) {   // (a run-time window: the caller asks for the fast detect only from 16 samples on)
        if (mode == kDetFast)
            return paired ? make_variant<AIN, NIPO, false, kDetFast, true, WAVES>() : make_variant<AIN, NIPO, false, kDetFast, false, WAVES>();
    }
    if (mode == kDetContracted)
        return paired ? make_variant<AIN, NIPO, false, kDetContracted, true, WAVES>()
                      : make_variant<AIN, NIPO, false, kDetContracted, false, WAVES>();
    return paired ? make_variant<AIN, NIPO, false, kDetCanonical, true, WAVES>() : make_variant<AIN, NIPO, false, kDetCanonical, false, WAVES>();
}

// Every instantiation of one antenna class (n_ipo 2 ... 64, general / conjugate-pair, three detect modes, stage parity).
// mode: kDet*; the fast detect falls back to canonical below n_ipo = 16 (include/dsabf.h).
template <int AIN>
FusedVariant fused16_variant(int n_ipo, bool write_c, int mode, bool paired)
{
    switch (n_ipo) {
        case 2: return fused16_variant_nipo<AIN, 2, kWaves16>(write_c, mode, paired);
        case 4: return fused16_variant_nipo<AIN, 4, kWaves16>(write_c, mode, paired);
        case 8: return fused16_variant_nipo<AIN, 8, kWaves16>(write_c, mode, paired);
        case 16: return fused16_variant_nipo<AIN, 16, kWaves16>(write_c, mode, paired);
        case 32: return fused16_variant_nipo<AIN, 32, kWaves16>(write_c, mode, paired);
        case 64: return fused16_variant_nipo<AIN, 64, kWaves16>(write_c, mode, paired);
        default: return fused16_variant_nipo<AIN, 0, kWaves16>(write_c, mode, paired);   // run-time window (the caller passes n_ipo = 0)
    }
}

// The conjugate-pair kernel with 8 output slots per wave (two-k-step classes, n_ipo >= 16; fused_col_tiles() in bf_kernels.hip).
// ... where its 236-256 registers hold without a spill: not the run-time dword-staged class (13 staging pieces per thread:
// 60-412 bytes of scratch per lane) and not 100 antennas at n_ipo 64 (140); those keep 4 slots on 8-wave workgroups.
template <int AIN, int NIPO>
constexpr bool ns8_fits() { return AIN == kAntK2P16 || (AIN == 100 && NIPO < 64); }

template <int AIN, int NIPO>
FusedVariant fused16_variant_ns8_nipo(int mode)
{
    if constexpr (!ns8_fits<AIN, NIPO>()) return FusedVariant{};
    else {
    if (mode == kDetFast) return make_variant<AIN, NIPO, false, kDetFast, true, kWaves16, kColTilesWide16>();
    if (mode == kDetContracted) return make_variant<AIN, NIPO, false, kDetContracted, true, kWaves16, kColTilesWide16>();
    return make_variant<AIN, NIPO, false, kDetCanonical, true, kWaves16, kColTilesWide16>();
    }
}

// The wide launches of the two-k-step classes (n_ipo >= 16), each kind in translation units of its own because each wants a
// different instruction scheduling strategy (dsabeamformer_amd/build.py): 8-wave workgroups (fused_wg_waves() in
// bf_kernels.hip), general kernel ...
template <int AIN, int NIPO, bool PAIRED>
FusedVariant fused16_variant_w8_nipo(int mode)
{
    if (mode == kDetFast) return make_variant<AIN, NIPO, false, kDetFast, PAIRED, kWavesWide16>();
    if (mode == kDetContracted) return make_variant<AIN, NIPO, false, kDetContracted, PAIRED, kWavesWide16>();
    return make_variant<AIN, NIPO, false, kDetCanonical, PAIRED, kWavesWide16>();
}

// (PAIRED = false: bf_fused16_*_w8.hip; true, the conjugate-pair kernel on 8-wave workgroups: bf_fused16_*_w8p.hip)
template <int AIN, bool PAIRED>
FusedVariant fused16_variant_w8(int n_ipo, int mode)
{
    static_assert(ant_two_ksteps<AIN>(), "the wide launches exist for the two-k-step classes only");
    switch (n_ipo) {
        case 16: return fused16_variant_w8_nipo<AIN, 16, PAIRED>(mode);
        case 32: return fused16_variant_w8_nipo<AIN, 32, PAIRED>(mode);
        case 64: return fused16_variant_w8_nipo<AIN, 64, PAIRED>(mode);
        default: return FusedVariant{};
    }
}

// ... and the conjugate-pair kernel on 4-wave workgroups whose waves own 8 output slots (fused_col_tiles()).
template <int AIN>
FusedVariant fused16_variant_s8(int n_ipo, int mode)
{
    static_assert(ant_two_ksteps<AIN>(), "the wide launches exist for the two-k-step classes only");
    switch (n_ipo) {
        case 16: return fused16_variant_ns8_nipo<AIN, 16>(mode);
        case 32: return fused16_variant_ns8_nipo<AIN, 32>(mode);
        case 64: return fused16_variant_ns8_nipo<AIN, 64>(mode);
        default: return FusedVariant{};
    }
}

// The deep classes (three / four k-steps): 8-wave workgroups; general kernel with 2 output slots per wave, conjugate-pair kernel
// with 4 (two pair tiles); n_ipo 16 / 32 / 64.
template <int AIN, int NIPO>
FusedVariant fused16_variant_deep_nipo(int mode, bool paired, int ns)
{
    if (paired && ns == 2) {   // one pair tile per wave: 8 waves x 32 beams = 256 beams per workgroup
        if (mode == kDetFast) return make_variant<AIN, NIPO, false, kDetFast, true, kWavesWide16, 2>();
        if (mode == kDetContracted) return make_variant<AIN, NIPO, false, kDetContracted, true, kWavesWide16, 2>();
        return make_variant<AIN, NIPO, false, kDetCanonical, true, kWavesWide16, 2>();
    }
    if (paired) {
        if (mode == kDetFast) return make_variant<AIN, NIPO, false, kDetFast, true, kWavesWide16, 4>();
        if (mode == kDetContracted) return make_variant<AIN, NIPO, false, kDetContracted, true, kWavesWide16, 4>();
        return make_variant<AIN, NIPO, false, kDetCanonical, true, kWavesWide16, 4>();
    }
    if (mode == kDetFast) return make_variant<AIN, NIPO, false, kDetFast, false, kWavesWide16, 2>();
    if (mode == kDetContracted) return make_variant<AIN, NIPO, false, kDetContracted, false, kWavesWide16, 2>();
    return make_variant<AIN, NIPO, false, kDetCanonical, false, kWavesWide16, 2>();
}
template <int AIN>
FusedVariant fused16_variant_deep(int n_ipo, int mode, bool paired, int ns)
{
    static_assert(ant_deep<AIN>(), "three or four k-steps");
    switch (n_ipo) {
        case 16: return fused16_variant_deep_nipo<AIN, 16>(mode, paired, ns);
        case 32: return fused16_variant_deep_nipo<AIN, 32>(mode, paired, ns);
        case 64:   // (the dword-staged classes would spill at this window -- 12 / 16 staging pieces per thread -- and lose to fusedg_kernel:
                   //  profiles/r05_deep_p4_perf.txt; deep_class() leaves them there)
            if constexpr (AIN == kAntK3P4 || AIN == kAntK4P4) return FusedVariant{};
            else return fused16_variant_deep_nipo<AIN, 64>(mode, paired, ns);
        default: return FusedVariant{};
    }
}
// ns: output slots per wave (general 2; conjugate-pair 4 where the beams come in groups of 512, else 2)
FusedVariant fused16_variant_k4p16(int n_ipo, int mode, bool paired, int ns);
FusedVariant fused16_variant_k3p16(int n_ipo, int mode, bool paired, int ns);
FusedVariant fused16_variant_k4p4(int n_ipo, int mode, bool paired, int ns);
FusedVariant fused16_variant_k3p4(int n_ipo, int mode, bool paired, int ns);

// One definition per antenna class, each in its own translation unit (bf_fused16_*.hip).
FusedVariant fused16_variant_a100(int n_ipo, bool write_c, int mode, bool paired);
FusedVariant fused16_variant_k1p16(int n_ipo, bool write_c, int mode, bool paired);
FusedVariant fused16_variant_k1p4(int n_ipo, bool write_c, int mode, bool paired);
FusedVariant fused16_variant_k2p16(int n_ipo, bool write_c, int mode, bool paired);
FusedVariant fused16_variant_k2p4(int n_ipo, bool write_c, int mode, bool paired);
// ... and three per two-k-step class for its wide launches (bf_fused16_*_w8.hip, bf_fused16_*_w8p.hip, bf_fused16_*_s8.hip)
FusedVariant fused16_variant_a100_w8(int n_ipo, int mode);
FusedVariant fused16_variant_k2p16_w8(int n_ipo, int mode);
FusedVariant fused16_variant_k2p4_w8(int n_ipo, int mode);
FusedVariant fused16_variant_a100_w8p(int n_ipo, int mode);
FusedVariant fused16_variant_k2p16_w8p(int n_ipo, int mode);
FusedVariant fused16_variant_k2p4_w8p(int n_ipo, int mode);
FusedVariant fused16_variant_a100_s8(int n_ipo, int mode);
FusedVariant fused16_variant_k2p16_s8(int n_ipo, int mode);
FusedVariant fused16_variant_k2p4_s8(int n_ipo, int mode);

}  // namespace dsabf
