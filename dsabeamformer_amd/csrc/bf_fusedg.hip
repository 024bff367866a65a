// bf_fusedg.hip -- fusedg_kernel: the fused expand + complex int8 GEMM + detect kernel for EVERY geometry of the reference's
// contract (src/beamformer.hh:47-60,155-156: N_BEAMS % 4 == 0, N_ANTENNAS % 4 == 0, any N_AVERAGING) that the specialised
// instantiations of fused16_kernel (bf_fused16.hpp: <= 128 antennas, n_ipo a power of two <= 64) do not cover: more than 128
// antennas (any number of k-steps of 64), any accumulation window n_ipo = n_pol * n_avg, gemm-units of any length.
//
// Same arithmetic, same bits (tests/test_gpu_round4.py holds it to the oracle and, on the geometries both cover, to fused16_kernel):
// v_mfma_i32_16x16x64_i8 on the complex -> real embedding (Vr | Vi) x (Wr | -Wi), (Wi | Wr), exact int32 sums, the reference's
// fp32 detect in the reference's order.  What differs is the loop structure -- fused16_kernel keeps a wave's WEIGHT fragments in
// registers for the whole kernel and its accumulators for one tile; that needs 48 registers per k-step and stops at two:
//
//   * ACCUMULATOR-stationary.  A wave owns 128 time rows x 32 beams (8 row tiles x 2 column tiles x {re, im} = 128 accumulator
//     registers) for one chunk and walks the k-steps: per k-step the workgroup stages ONE 128-row plane of 64 antennas into
//     LDS (double-buffered single planes: 32 KiB whatever the antenna count), every wave reads its 16 A fragments from it and
//     streams its 6 B (weight) fragments of that k-step from the fragment image in L2 (one k-step ahead, in registers):
//     64 MFMAs between two barriers.  B traffic is 96 B per MFMA (24 B/clk/CU), A reads 0.25 ds_read_b128 per MFMA.
//   * The detect runs once per chunk, after the last k-step, on the finished accumulators -- its cost does not depend on the
//     antenna count, so it is amortised over ks x the MACs: the more antennas, the closer to the matrix pipe's rate.
//   * Voltages are staged as TRUE nibble values (sign-extended, -8..7), not x16 as in fused16_kernel: the accumulator then holds
//     seed + n with |n| <= 2032 * n_ant < 2^22 up to 2064 antennas, so the magic-seed conversion (int32 0x4B400000 + n = the bits of
//     the float 1.5 * 2^23 + n, then ONE fma for fl(n / 127)) holds for every supported antenna count; the sign extension is
//     paid once per workgroup while staging, not per wave.
//   * Any n_ipo.  A lane group's 32 rows of a chunk are one STREAM: `kout` whole accumulation windows back to back
//     (kout = 32 / n_ipo windows of n_ipo samples when they fit, else one window over cpg = ceil(n_ipo / 32) chunks), so a
//     window never straddles two lane groups and its sum is one lane's sequential fp32 chain in time order, as the reference
//     sums it (src/beamformer.cuh:150-152).  Rows behind the last whole window of a run are padding (zero voltages, never
//     stored): efficiency kout * n_ipo / 32, e.g. 30/32 for n_ipo = 6 or 10, 24/32 for 12 or 24 (the power-of-two windows of
//     fused16_kernel have none).  Window starts and ends are wave-uniform, so they cost scalar branches, not lane masks.
#include "bf_fused16.hpp"

#include <cstdio>

namespace dsabf {

namespace {

constexpr int kGWaves = 8;                 // waves per workgroup
constexpr int kGThreads = 64 * kGWaves;
constexpr int kGNT = 2;                    // 16-beam column tiles per wave: 8 waves x 32 beams = 256 beams per workgroup
constexpr int kGPlane = kRowsPerChunk * 128;   // LDS bytes of one staged plane: 128 rows x (64 re | 64 im)
constexpr float kNegMagicAlpha = -(kMagic * kAlpha);
static_assert((double)kMagic * (double)kAlpha == (double)(kMagic * kAlpha), "K * alpha must be exactly representable");

struct GenArgs {
    const uint8_t* __restrict__ in;   // packed voltages [unit][f][t][a]
    const v4i* __restrict__ wimg;     // general weight fragment image [f][ct16][Wr, -Wi, Wi][k-step][lane]
    float* __restrict__ out;          // detected [unit*n_out + o][f][b]   (WRITE_C: c[f][t][b]{re,im})
    int n_freq, n_beams, n_bgroups, n_ctiles, n_ant, ks;
    int T;                            // time samples per gemm-unit
    int L;                            // n_ipo: samples per output
    int Ls;                           // samples per stream = kout * L
    int kout;                         // outputs per stream
    int cpg;                          // chunks per group of 4 streams = ceil(Ls / 32)
    unsigned S;                       // total time samples per frequency in this launch
    int chunks_total, n_tsplit, interleave;
};

__device__ __forceinline__ unsigned sext_nibbles(unsigned nib)   // four 4-bit two's complement values in the low nibbles of 4 bytes
{
    return ((nib ^ 0x88888888u) - 0x08080808u) ^ 0x80808080u;
}

// P16: packed rows are 16-byte aligned (n_ant % 16 == 0): one 16-byte piece per thread and plane; else four dwords.
template <bool P16, int MODE, bool WRITE_C>
__global__ __launch_bounds__(kGThreads, 2) void fusedg_kernel(GenArgs a)
{
    constexpr bool FAST = MODE == kDetFast;
    constexpr bool CONTRACTED = MODE == kDetContracted;
    constexpr int PPT = P16 ? 1 : 4;                       // staging pieces per thread per plane
    using stage_t = std::conditional_t<P16, v4i, int>;

    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 planes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4;
    const int c16 = lane & 15;

    // blockIdx -> (f, beam group, time split): as fused16_kernel (a frequency's workgroups share one XCD's L2)
    int f, bg, ts;
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
    const int groups_total = a.chunks_total / a.cpg;
    const int c_begin = (int)(((long long)groups_total * ts) / a.n_tsplit) * a.cpg;
    const int c_end = (int)(((long long)groups_total * (ts + 1)) / a.n_tsplit) * a.cpg;
    if (c_begin >= c_end) return;

    const int A = a.n_ant, KS = a.ks;
    const int ct0 = (bg * kGWaves + wave) * kGNT;
    const bool wave_active = ct0 < a.n_ctiles;
    int slot_beam[kGNT];
    bool slot_ok[kGNT];
#pragma unroll
    for (int t = 0; t < kGNT; t++) {
        slot_ok[t] = ct0 + t < a.n_ctiles;
        slot_beam[t] = slot_ok[t] ? beam_of_tile(a.interleave ? kGNT : 0, ct0 + t, c16) : a.n_beams;
    }
    // B fragments of k-step h: bw[t][Wr, -Wi, Wi]
    auto load_b = [&](v4i (&bw)[kGNT][3], int h) {
#pragma unroll
        for (int t = 0; t < kGNT; t++)
#pragma unroll
            for (int k = 0; k < 3; k++)
                bw[t][k] = slot_ok[t] ? a.wimg[((((size_t)f * a.n_ctiles + ct0 + t) * 3 + k) * KS + h) * 64 + lane] : v4i{0, 0, 0, 0};
    };

    // ---- staging: this thread's pieces of a plane ---------------------------------------------------------------------------
    int lds_re[PPT];       // LDS byte offset of the piece's re image inside a plane; the im image is at ^ 64
    int poff[PPT];         // byte offset of the piece inside the 64 antennas of a k-step
    int prow[PPT];         // chunk row of the piece
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int pc = tid + k * kGThreads;
        const int row = P16 ? pc >> 2 : pc >> 4, pi = P16 ? pc & 3 : pc & 15;
        const int kp = P16 ? pi : pi >> 2, sub = P16 ? 0 : 4 * (pi & 3);
        prow[k] = row;
        poff[k] = (P16 ? 16 : 4) * pi;
        lds_re[k] = row * 128 + 16 * swz16<32>(kp, row) + sub;
    }
    size_t rowoff[PPT];    // byte offset of the row's first antenna in `in` for the chunk being loaded
    bool rowok[PPT];
    auto row_meta = [&](int c) {
        const unsigned grp = (unsigned)(c / a.cpg), cc = (unsigned)(c % a.cpg);
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const unsigned run = (unsigned)prow[k] >> 5, j = (unsigned)prow[k] & 31u;
            const unsigned pos = 32u * cc + j;
            const unsigned long long s = (unsigned long long)(4u * grp + run) * (unsigned)a.Ls + pos;
            rowok[k] = pos < (unsigned)a.Ls && s < a.S;
            const unsigned ss = rowok[k] ? (unsigned)s : 0u;
            const unsigned u = ss / (unsigned)a.T, t = ss - u * (unsigned)a.T;
            rowoff[k] = ((size_t)((size_t)u * a.n_freq + f) * a.T + t) * (size_t)A;
        }
    };
    stage_t stage[PPT];
    int ld_c = c_begin, ld_h = 0;     // the plane the next load_plane() fetches
    bool ld_more = true;
    auto load_plane = [&]() {
        if (!ld_more) return;
        if (ld_h == 0) row_meta(ld_c);
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            stage[k] = stage_t{};
            const int ab = 64 * ld_h + poff[k];
            if (rowok[k] && ab < A) stage[k] = *reinterpret_cast<const stage_t*>(a.in + rowoff[k] + ab);
        }
        if (++ld_h == KS) {
            ld_h = 0;
            ld_more = ++ld_c < c_end;
        }
    };
    auto write_plane = [&](char* buf) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            if constexpr (P16) {
                v4i re, im;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned w = (unsigned)stage[k][d];
                    re[d] = (int)sext_nibbles((w >> 4) & 0x0F0F0F0Fu);
                    im[d] = (int)sext_nibbles(w & 0x0F0F0F0Fu);
                }
                *reinterpret_cast<v4i*>(buf + lds_re[k]) = re;
                *reinterpret_cast<v4i*>(buf + (lds_re[k] ^ 64)) = im;
            } else {
                const unsigned w = (unsigned)stage[k];
                *reinterpret_cast<int*>(buf + lds_re[k]) = (int)sext_nibbles((w >> 4) & 0x0F0F0F0Fu);
                *reinterpret_cast<int*>(buf + (lds_re[k] ^ 64)) = (int)sext_nibbles(w & 0x0F0F0F0Fu);
            }
        }
    };

    // ---- accumulators and the running sums -----------------------------------------------------------------------------------
    v4i acc[8][kGNT][2];                  // [row tile][column tile][re, im]: seed + n
    v4i kc = {(int)kMagicBits, (int)kMagicBits, (int)kMagicBits, (int)kMagicBits};
    asm volatile("" : "+v"(kc));
    float sum[kGNT] = {0.0f, 0.0f};
    const size_t FB = (size_t)a.n_freq * a.n_beams;

    auto store_slots = [&](float* row, const float (&x)[kGNT]) {
        if (a.interleave) {   // a lane's two beams are neighbours: 8-byte stores, whole 128-byte lines per lane group
            __builtin_nontemporal_store(v2f{x[0], x[1]}, reinterpret_cast<v2f*>(row + slot_beam[0]));
        } else {
#pragma unroll
            for (int t = 0; t < kGNT; t++)
                if (slot_beam[t] < a.n_beams) row[slot_beam[t]] = x[t];
        }
    };

    v4i bw[kGNT][3];                       // B fragments of the current k-step
    load_b(bw, 0);
    load_plane();
    write_plane(smem);
    load_plane();
    __syncthreads();

    int p = 0;                             // plane counter (parity = LDS buffer)
    for (int c = c_begin; c < c_end; c++) {
        for (int h = 0; h < KS; h++, p++) {
            char* cur = smem + (p & 1) * kGPlane;
            char* nxt = smem + ((p + 1) & 1) * kGPlane;
            const bool last_plane = (c + 1 == c_end) && (h + 1 == KS);
            if (h == 0) {
#pragma unroll
                for (int t8 = 0; t8 < 8; t8++)
#pragma unroll
                    for (int t = 0; t < kGNT; t++) acc[t8][t][0] = acc[t8][t][1] = kc;
            }
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) {
                if (wave_active) {
                    const int row = lds_row16<32>(t8, c16);
                    const v4i a0 = *reinterpret_cast<const v4i*>(cur + row * 128 + 16 * swz16<32>(g4, row));       // Vr
                    const v4i a1 = *reinterpret_cast<const v4i*>(cur + row * 128 + 16 * swz16<32>(g4 + 4, row));   // Vi
#pragma unroll
                    for (int t = 0; t < kGNT; t++) {
                        acc[t8][t][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bw[t][0], acc[t8][t][0], 0, 0, 0);   // + Wr Vr
                        acc[t8][t][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bw[t][2], acc[t8][t][1], 0, 0, 0);   // + Wi Vr
                        acc[t8][t][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bw[t][1], acc[t8][t][0], 0, 0, 0);   // - Wi Vi
                        acc[t8][t][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bw[t][0], acc[t8][t][1], 0, 0, 0);   // + Wr Vi
                    }
                }
                if (t8 == 1 && !last_plane) write_plane(nxt);
                if (t8 == 3) load_plane();
                // keep the scheduler from hoisting every row tile's fragment reads to the top of the plane (64 registers
                // next to 128 accumulators): two tiles in flight are enough to cover the LDS latency
                if (t8 & 1) __builtin_amdgcn_sched_barrier(0);
            }
            // the next k-step's B fragments: the MFMAs above have read the registers; the loads fly during the detect / the barrier
            if (!last_plane) load_b(bw, h + 1 == KS ? 0 : h + 1);

            if (h + 1 == KS && wave_active) {
                // ---- detect: the chunk's 128 rows x 32 beams are complete ------------------------------------------------
                const unsigned grp = (unsigned)(c / a.cpg), cc = (unsigned)(c % a.cpg);
                const unsigned sigma = 4u * grp + (unsigned)g4;          // this lane's stream
                int m = (int)((32u * cc) % (unsigned)a.L);               // position inside the window of this run's row 0
                unsigned oq = (32u * cc) / (unsigned)a.L;                // windows of the stream that ended before it
#pragma unroll
                for (int t8 = 0; t8 < 8; t8++) {
                    float pw[kGNT][4];
#pragma unroll
                    for (int t = 0; t < kGNT; t++) {
                        const v4f fr = __builtin_bit_cast(v4f, acc[t8][t][0]), fi = __builtin_bit_cast(v4f, acc[t8][t][1]);
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            if constexpr (WRITE_C) {
                                const unsigned pos = 32u * cc + 4u * t8 + r;
                                const unsigned long long s = (unsigned long long)sigma * (unsigned)a.Ls + pos;
                                if (pos < (unsigned)a.Ls && s < a.S && slot_beam[t] < a.n_beams) {
                                    v2f cv = {__builtin_fmaf(fr[r], kAlpha, kNegMagicAlpha), __builtin_fmaf(fi[r], kAlpha, kNegMagicAlpha)};
                                    *reinterpret_cast<v2f*>(a.out + 2 * (((size_t)f * a.T + (size_t)s) * a.n_beams + slot_beam[t])) = cv;
                                }
                                pw[t][r] = 0.0f;
                            } else if constexpr (FAST) {
                                pw[t][r] = 0.0f;   // (unused: the fast detect chains its fmas below)
                            } else {
                                const float x = __builtin_fmaf(fr[r], kAlpha, kNegMagicAlpha);
                                const float y = __builtin_fmaf(fi[r], kAlpha, kNegMagicAlpha);
                                const float yy = y * y;
                                if constexpr (CONTRACTED) {
                                    pw[t][r] = __builtin_fmaf(x, x, yy);
                                } else {
                                    const float xx = x * x;
                                    pw[t][r] = xx + yy;
                                }
                            }
                        }
                    }
                    if constexpr (!WRITE_C) {
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            // wave-uniform: does a window start / end at this row?
                            const bool start = m == 0;
#pragma unroll
                            for (int t = 0; t < kGNT; t++) {
                                if constexpr (FAST) {
                                    const v4f fr = __builtin_bit_cast(v4f, acc[t8][t][0]), fi = __builtin_bit_cast(v4f, acc[t8][t][1]);
                                    const float dr = fr[r] - kMagic, di = fi[r] - kMagic;
                                    float s0 = start ? 0.0f : sum[t];
                                    s0 = __builtin_fmaf(dr, dr, s0);
                                    sum[t] = __builtin_fmaf(di, di, s0);
                                } else {
                                    sum[t] = start ? pw[t][r] : sum[t] + pw[t][r];
                                }
                            }
                            if (++m == a.L) {
                                m = 0;
                                const unsigned o = sigma * (unsigned)a.kout + oq;     // this lane's output (over the whole launch)
                                oq++;
                                if ((unsigned long long)o * (unsigned)a.L < a.S) {
                                    float x[kGNT];
#pragma unroll
                                    for (int t = 0; t < kGNT; t++) x[t] = FAST ? sum[t] * (kAlpha * kAlpha) : sum[t];
                                    store_slots(a.out + (size_t)o * FB + (size_t)f * a.n_beams, x);
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

template <bool P16, int MODE, bool WRITE_C>
hipError_t launch_g(const GenArgs& args, const LaunchShape& ls, hipStream_t s)
{
    (void)hipGetLastError();
    hipLaunchKernelGGL((fusedg_kernel<P16, MODE, WRITE_C>), dim3(ls.grid), dim3(ls.block), ls.lds_bytes, s, args);
    return hipGetLastError();
}

template <bool P16>
hipError_t launch_g_mode(int mode, bool write_c, const GenArgs& args, const LaunchShape& ls, hipStream_t s)
{
    if (write_c) return launch_g<P16, kDetCanonical, true>(args, ls, s);
    if (mode == kDetFast) return launch_g<P16, kDetFast, false>(args, ls, s);
    if (mode == kDetContracted) return launch_g<P16, kDetContracted, false>(args, ls, s);
    return launch_g<P16, kDetCanonical, false>(args, ls, s);
}

template <bool P16>
const void* kernel_g_mode(int mode)
{
    if (mode == kDetFast) return reinterpret_cast<const void*>(fusedg_kernel<P16, kDetFast, false>);
    if (mode == kDetContracted) return reinterpret_cast<const void*>(fusedg_kernel<P16, kDetContracted, false>);
    return reinterpret_cast<const void*>(fusedg_kernel<P16, kDetCanonical, false>);
}

int generic_mode(const Geometry& g) { return g.fast_detect ? kDetFast : g.contracted_detect ? kDetContracted : kDetCanonical; }

}  // namespace

int generic_ksteps(const Geometry& g) { return (g.n_ant + 63) / 64; }
int generic_interleave(const Geometry& g) { return g.n_beams % (16 * kGNT) == 0 ? kGNT : 0; }

bool generic_supported(const Geometry& g, const char** why)
{
    const char* dummy;
    if (!why) why = &dummy;
    if (g.n_ant > kGenericMaxAnt) { *why = "more than 2048 antennas: the int32 sums would leave the exactly convertible range"; return false; }
    if (g.n_ipo <= 0 || g.n_time % g.n_ipo) { *why = "n_time must be n_out * n_pol * n_avg"; return false; }
    return true;
}

// Streams: kout whole windows of n_ipo samples per 32-row run when they fit, else one window over cpg chunks.
LaunchShape generic_launch_shape(const Geometry& g, int n_units, int n_cus)
{
    LaunchShape ls{};
    const int L = g.n_ipo;
    const int kout = L <= 32 ? 32 / L : 1;
    const long long Ls = (long long)kout * L;
    const int cpg = (int)((Ls + 31) / 32);
    const long long S = (long long)n_units * g.n_time;
    const long long n_streams = (S + Ls - 1) / Ls;
    const long long groups = (n_streams + 3) / 4;
    ls.chunks_total = (int)(groups * cpg);
    ls.n_bgroups = (g.n_beams + 16 * kGNT * kGWaves - 1) / (16 * kGNT * kGWaves);
    const long long base = (long long)g.n_freq * ls.n_bgroups;
    // one 8-wave workgroup is resident per CU (its 128 accumulator registers): two rounds of workgroups fill the chip's tail,
    // but a workgroup should keep >= 2 chunk groups (the prologue and the B fragments of k-step 0 are paid per workgroup)
    long long want = (2LL * n_cus + base - 1) / base;
    const long long max_split = groups >= 2 ? groups / 2 : 1;
    if (want > max_split) want = max_split;
    if (g.tsplit > 0) want = g.tsplit;
    if (want < 1) want = 1;
    if (want > groups) want = groups;
    ls.n_tsplit = (int)want;
    ls.grid = (int)(base * ls.n_tsplit);
    ls.block = kGThreads;
    ls.lds_bytes = 2 * kGPlane;
    if (g.lds_pad > 0) ls.lds_bytes += g.lds_pad < kLdsPerCuBytes - ls.lds_bytes ? g.lds_pad : kLdsPerCuBytes - ls.lds_bytes;
    return ls;
}

hipError_t launch_fused_generic(const Geometry& g, const void* d_image, const void* d_packed, int n_units, float* d_out, int n_cus,
                                bool write_c, hipStream_t s)
{
    if (n_units <= 0) return hipSuccess;
    if ((long long)n_units * g.n_time > 0x3fffffffLL) return hipErrorInvalidValue;
    const LaunchShape ls = generic_launch_shape(g, n_units, n_cus);
    GenArgs a{};
    a.in = static_cast<const uint8_t*>(d_packed);
    a.wimg = static_cast<const v4i*>(d_image);
    a.out = d_out;
    a.n_freq = g.n_freq;
    a.n_beams = g.n_beams;
    a.n_bgroups = ls.n_bgroups;
    a.n_ctiles = g.n_ctiles;
    a.n_ant = g.n_ant;
    a.ks = generic_ksteps(g);
    a.T = g.n_time;
    a.L = g.n_ipo;
    a.kout = g.n_ipo <= 32 ? 32 / g.n_ipo : 1;
    a.Ls = a.kout * a.L;
    a.cpg = (a.Ls + 31) / 32;
    a.S = (unsigned)((long long)n_units * g.n_time);
    a.chunks_total = ls.chunks_total;
    a.n_tsplit = ls.n_tsplit;
    a.interleave = generic_interleave(g);
    const int mode = generic_mode(g);
    if (ls.lds_bytes > 48 * 1024) {   // only with the lds_pad measurement switch
        for (int m = 0; m < 3; m++) {
            (void)hipFuncSetAttribute(kernel_g_mode<true>(m), hipFuncAttributeMaxDynamicSharedMemorySize, ls.lds_bytes);
            (void)hipFuncSetAttribute(kernel_g_mode<false>(m), hipFuncAttributeMaxDynamicSharedMemorySize, ls.lds_bytes);
        }
    }
    return g.n_ant % 16 == 0 ? launch_g_mode<true>(mode, write_c, a, ls, s) : launch_g_mode<false>(mode, write_c, a, ls, s);
}

int generic_vgprs(const Geometry& g)
{
    hipFuncAttributes attr{};
    const void* fn = g.n_ant % 16 == 0 ? kernel_g_mode<true>(generic_mode(g)) : kernel_g_mode<false>(generic_mode(g));
    if (hipFuncGetAttributes(&attr, fn) != hipSuccess) return -1;
    return attr.numRegs;
}

}  // namespace dsabf
