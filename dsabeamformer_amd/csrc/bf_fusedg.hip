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
//     Workgroups are 4 waves = 128 beams and TWO are resident per CU: they are not synchronised with each other, so one's
//     barrier, B-fragment waits and detect phase run in the shadow of the other's MFMAs.
//   * The detect runs once per chunk, after the last k-step, on the finished accumulators -- its cost does not depend on the
//     antenna count, so it is amortised over ks x the MACs: the more antennas, the closer to the matrix pipe's rate.
//   * Voltages are staged as OFFSET nibbles u = v + 8 (0..15: the packed nibble with its sign bit flipped -- an AND and an XOR
//     per four samples, no sign extension), not x16 as in fused16_kernel: sum W (u - 8) = sum W u - 8 sum W, and 8 sum W is a
//     constant per (frequency, beam, re | im) that weight_colsum_kernel tabulates when the weights are set.  A chunk's
//     accumulators START at seed - 8 sum W (one register move each, as a plain seed would cost), so the finished sums are
//     seed + n with |n| <= 2032 * n_ant < 2^22 up to 2064 antennas: the magic-seed conversion (int32 0x4B400000 + n = the bits
//     of the float 1.5 * 2^23 + n, then ONE fma for fl(n / 127)) holds for every supported antenna count.
//   * An accumulator's two MFMAs of a plane are issued back to back (the second continues the first inside the matrix unit):
//     what the pipe charges for is every accumulator that enters and leaves it (profiles/r04_ubench_chains.txt).
//   * Any n_ipo.  A lane group's rows are one STREAM: `kout` whole accumulation windows back to back over cpg = ceil(kout *
//     n_ipo / 32) chunks of 32 rows, so a window never straddles two lane groups and its sum is one lane's sequential fp32 chain
//     in time order, as the reference sums it (src/beamformer.cuh:150-152).  Rows behind the stream's last window are padding
//     (zero voltages, never stored): efficiency kout * n_ipo / (32 cpg).  A small launch takes what fits one chunk (30/32 for
//     n_ipo = 6 or 10, 24/32 for 12 or 24); one that fills the chip the kout with the least padding that still does -- whole
//     chunks (4 windows of 24 = 3 chunks) -- rtw_kout, bf_kernels.hip (round 5; the power-of-two windows have none either
//     way).  Window starts and ends are wave-uniform, so they cost scalar branches, not lane masks.
#include "bf_fused16.hpp"

#include <cstdio>

namespace dsabf {

namespace {

// (no compile-time switches: the round-robin accumulator order and the timing ablations of round 4 -- profiles/r04_generic_ablate.txt --
//  were measurement arms and are gone)
constexpr int kGWaves = 4;                 // waves per workgroup: TWO workgroups are resident per CU (204-235 registers), unsynchronised --
                                           // one's barriers, B-fragment waits and detect phases overlap the other's MFMAs
constexpr int kGThreads = 64 * kGWaves;
constexpr int kGNT = 2;                    // 16-beam column tiles per wave: 4 waves x 32 beams = 128 beams per workgroup
constexpr int kGPlane = kRowsPerChunk * 128;   // LDS bytes of one staged plane: 128 rows x (64 re | 64 im)

typedef int v2i_g __attribute__((ext_vector_type(2)));

struct GenArgs {
    const uint8_t* __restrict__ in;   // packed voltages [unit][f][t][a]
    const v4i* __restrict__ wimg;     // general weight fragment image [f][ct16][Wr, -Wi, Wi][k-step][lane]
    const v2i_g* __restrict__ corr;   // [f][ct16][column] {-8 sum_a (Wr - Wi), -8 sum_a (Wr + Wi)}: the offset-nibble correction
    unsigned long long wimg_bytes;    // bytes of the fragment image
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

// A thread stages two 16-byte pieces per plane.  P16: packed rows are 16-byte aligned (n_ant % 16 == 0): one 16-byte load per
// piece; else four dword loads.
template <bool P16, int MODE, bool WRITE_C>
__global__ __launch_bounds__(kGThreads, 2) void fusedg_kernel(GenArgs a)
{
    constexpr bool FAST = MODE == kDetFast;
    constexpr bool CONTRACTED = MODE == kDetContracted;
    constexpr int PPT = 512 / kGThreads;                   // 16-byte staging pieces per thread per plane
    using stage_t = v4i;

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
    // B fragments of k-step h: bw[t][Wr, -Wi, Wi].  EVERY memory operation of this kernel is unconditional (a column tile behind
    // the last one re-reads the wave's first one and its results are never stored; rows and pieces outside the data load a valid
    // address: their weights are zero or their outputs never stored): a load under a branch -- even a wave-uniform one -- leaves
    // the compiler's s_waitcnt pass unable to count what is in flight, and it answers with vmcnt(0) in front of every MFMA
    // group (measured: 3,700 of 5,700 cycles per plane were such waits).
    // Addressing stays on the scalar unit: one buffer descriptor per wave (base = the wave's first column tile of this
    // frequency), the lane's 16 bytes as the only vector offset, (tile, component, k-step) in the scalar offset.
    const int ct_base = min(ct0, a.n_ctiles - 1);
    const size_t b_base = ((size_t)f * a.n_ctiles + ct_base) * 3 * KS * 1024;          // bytes
    const unsigned long long b_left = a.wimg_bytes - b_base;
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wimg) + b_base), 0, (int)(b_left < 0xfffffff0ull ? b_left : 0xfffffff0ull), 0x00020000);
    const int b_lane = lane * 16;
    auto load_b = [&](v4i (&bw)[kGNT][3], int h) {
#pragma unroll
        for (int t = 0; t < kGNT; t++) {
            const int tt = (ct0 + t < a.n_ctiles) ? t : 0;     // wave-uniform
#pragma unroll
            for (int k = 0; k < 3; k++)
                bw[t][k] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(b_rsrc, b_lane, (((tt * 3 + k) * KS) + h) * 1024, 0));
        }
    };
    // a chunk's accumulators start at seed + the offset-nibble correction of this lane's column, per slot and component: the finished
    // sums are seed + sum W (v + 8) - 8 sum W = seed + n, and their bits ARE the float K + n (no integer add in the detect)
    v4i seed[kGNT][2];
#pragma unroll
    for (int t = 0; t < kGNT; t++) {
        const v2i_g cr = a.corr[((size_t)f * a.n_ctiles + min(ct0 + t, a.n_ctiles - 1)) * 16 + c16];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int sd = (int)kMagicBits + cr[k];
            seed[t][k] = v4i{sd, sd, sd, sd};
            asm volatile("" : "+v"(seed[t][k]));
        }
    }

    // ---- staging: this thread's pieces of a plane ---------------------------------------------------------------------------
    int lds_re[PPT];       // LDS byte offset of the piece's re image inside a plane; the im image is at ^ 64
    int poff[PPT];         // byte offset of the piece inside the 64 antennas of a k-step
    int prow[PPT];         // chunk row of the piece
#pragma unroll
    for (int k = 0; k < PPT; k++) {
        const int pc = tid + k * kGThreads;
        const int row = pc >> 2, pi = pc & 3;
        prow[k] = row;
        poff[k] = 16 * pi;
        lds_re[k] = row * 128 + 16 * swz16<32>(pi, row);
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
            rowoff[k] = ((size_t)((size_t)u * a.n_freq + f) * a.T + t) * (size_t)A;   // (a row outside the data: sample 0's)
        }
    };
    // One register set of staged (still packed) pieces, requested at row tile 3 of plane p for plane p + 2 and written to LDS at
    // row tiles 1 and 2 of plane p + 1.  (Two sets, requested 1.75 planes ahead, measured the same: the voltage stream's latency
    // is not what the kernel waits for.)
    stage_t stage1[PPT];
    int ld_c = c_begin, ld_h = 0;     // the plane the next load_plane() fetches; behind the last plane: the last one again
    auto load_plane = [&](stage_t (&stage)[PPT]) {
        if (ld_h == 0) row_meta(ld_c);
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int ab = 64 * ld_h + poff[k];     // bytes behind the last antenna meet zero weights: any value will do
            if constexpr (P16) {
                stage[k] = *reinterpret_cast<const v4i*>(a.in + rowoff[k] + min(ab, A - 16));
            } else {
#pragma unroll
                for (int d = 0; d < 4; d++) stage[k][d] = *reinterpret_cast<const int*>(a.in + rowoff[k] + min(ab + 4 * d, A - 4));
            }
        }
        if (++ld_h == KS) {
            ld_h = 0;
            if (ld_c + 1 < c_end) ld_c++;
        }
    };
    auto write_piece = [&](char* buf, const stage_t (&stage)[PPT], int k) {   // expand + store ONE 16-byte piece: (v + 8) nibbles
        v4i re, im;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const unsigned w = (unsigned)stage[k][d];
            re[d] = (int)(((w >> 4) & 0x0F0F0F0Fu) ^ 0x08080808u);
            im[d] = (int)((w & 0x0F0F0F0Fu) ^ 0x08080808u);
        }
        *reinterpret_cast<v4i*>(buf + lds_re[k]) = re;
        *reinterpret_cast<v4i*>(buf + (lds_re[k] ^ 64)) = im;
    };
    auto write_plane = [&](char* buf, const stage_t (&stage)[PPT]) {
#pragma unroll
        for (int k = 0; k < PPT; k++) write_piece(buf, stage, k);
    };

    // ---- accumulators and the running sums -----------------------------------------------------------------------------------
    v4i acc[8][kGNT][2];                  // [row tile][column tile][re, im]: seed + n
    float sum[kGNT] = {0.0f, 0.0f};
    [[maybe_unused]] const v4i kzero4 = {0, 0, 0, 0};
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

    v4i bw0[kGNT][3], bw1[kGNT][3];        // B fragments of the current k-step and of the next one (they swap roles per plane)
    load_b(bw0, 0);
    load_plane(stage1);               // plane 0: straight to LDS
    write_plane(smem, stage1);
    load_plane(stage1);               // plane 1 (written during plane 0)
    __syncthreads();

    int c = c_begin, h = 0, p = 0;         // the plane being computed: chunk, k-step, running index (parity = LDS buffer)
    const int n_planes = (c_end - c_begin) * KS;
    // one plane: 64 MFMAs per wave on LDS buffer p & 1 with the B fragments bc, while bn receives the next k-step's; after a
    // chunk's last k-step the detect.  Called with (bw0, bw1) for even planes and (bw1, bw0) for odd ones.
    auto plane = [&](const v4i (&bc)[kGNT][3], v4i (&bn)[kGNT][3], stage_t (&stage)[PPT]) {
        char* cur = smem + (p & 1) * kGPlane;
        char* nxt = smem + ((p + 1) & 1) * kGPlane;
        load_b(bn, h + 1 == KS ? 0 : h + 1);   // one plane ahead: lands behind this plane's MFMAs (behind the last plane: unused)
        // A fragments one row tile ahead (two register sets), the order pinned: the scheduler otherwise either reads every
        // tile's fragments at the top of the plane (64 registers beside 128 accumulators: spills) or each tile's just in time
        // (its LDS latency in front of every 8 MFMAs).
        auto tiles = [&]() {
            v4i fa[2][2];
            auto read_frag = [&](int t8, v4i (&fr)[2]) {
                const int row = lds_row16<32>(t8, c16);
                fr[0] = *reinterpret_cast<const v4i*>(cur + row * 128 + 16 * swz16<32>(g4, row));       // Vr + 8
                fr[1] = *reinterpret_cast<const v4i*>(cur + row * 128 + 16 * swz16<32>(g4 + 4, row));   // Vi + 8
            };
            read_frag(0, fa[0]);
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++) {
                if (t8 + 1 < 8) read_frag(t8 + 1, fa[(t8 + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                {   // (a wave behind the last beam computes too, on its first tile's weights: no branch around the MFMAs)
                    const v4i a0 = fa[t8 & 1][0], a1 = fa[t8 & 1][1];
                    // chain by chain: an accumulator's two MFMAs of this plane back to back -- the second continues the first inside
                    // the matrix unit instead of reading the accumulator back from the VGPRs (tools/ubench_chains.hip: 16 chains of
                    // 2 at two waves per SIMD, 0.585 round robin -> 0.676 chain by chain)
#pragma unroll
                    for (int t = 0; t < kGNT; t++) {
                        acc[t8][t][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bc[t][0], acc[t8][t][0], 0, 0, 0);   // + Wr Vr
                        __builtin_amdgcn_sched_barrier(0x7F6);
                        acc[t8][t][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bc[t][1], acc[t8][t][0], 0, 0, 0);   // - Wi Vi
                        __builtin_amdgcn_sched_barrier(0x7F6);
                        acc[t8][t][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, bc[t][2], acc[t8][t][1], 0, 0, 0);   // + Wi Vr
                        __builtin_amdgcn_sched_barrier(0x7F6);
                        acc[t8][t][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, bc[t][0], acc[t8][t][1], 0, 0, 0);   // + Wr Vi
                        __builtin_amdgcn_sched_barrier(0x7F6);
                    }
                }
                // plane p + 1 goes to LDS one piece per row tile (behind the last plane: a copy nobody reads), then plane p + 2
                // is requested into the registers just emptied
                if (t8 == 1) write_piece(nxt, stage, 0);
                if constexpr (PPT > 1)
                    if (t8 == 2) write_piece(nxt, stage, 1);
                if (t8 == 3) load_plane(stage);     // plane p + 2
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // (peeling the first k-step -- srcC = the seeds instead of 128 register moves per chunk -- was tried: four copies of the tile
        //  loop instead of two, and the register allocator spills 175 registers across the join)
        if (h == 0) {
#pragma unroll
            for (int t8 = 0; t8 < 8; t8++)
#pragma unroll
                for (int t = 0; t < kGNT; t++) {
                    acc[t8][t][0] = seed[t][0];
                    acc[t8][t][1] = seed[t][1];
                }
        }
        tiles();
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
                                sum[t] = __builtin_fmaf(sum[t], start ? 0.0f : 1.0f, pw[t][r]);   // (0 * sum + p = p; 1 * sum + p in one rounding)
                            }
                        }
                        if (++m == a.L) {
                            m = 0;
                            const unsigned o = sigma * (unsigned)a.kout + oq;     // this lane's output (over the whole launch)
                            // (a "window" of padding rows behind the stream's last one is nobody's)
                            if (oq++ < (unsigned)a.kout && (unsigned long long)o * (unsigned)a.L < a.S) {
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
        p++;
        if (++h == KS) {
            h = 0;
            c++;
        }
    };
    while (p < n_planes) {
        plane(bw0, bw1, stage1);
        if (p < n_planes) plane(bw1, bw0, stage1);
    }
}

// corr[f][ct16][column] = {-8 sum_a (Wr - Wi), -8 sum_a (Wr + Wi)} of the beam that column holds (beam_of_tile): what the
// offset nibbles (v + 8) add to the real / imaginary sums, taken back through the accumulator seeds.
__global__ void weight_colsum_kernel(const int8_t* __restrict__ w, v2i_g* __restrict__ corr, int n_freq, int n_ant, int n_beams,
                                     int n_ctiles, int interleave)
{
    const size_t total = (size_t)n_freq * n_ctiles * 16;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx & 15);
        const int ct = (int)((idx >> 4) % n_ctiles);
        const int f = (int)((idx >> 4) / n_ctiles);
        const int b = beam_of_tile(interleave, ct, c);
        int sr = 0, si = 0;
        if (b < n_beams)
            for (int ant = 0; ant < n_ant; ant++) {
                const int8_t* e = w + 2 * (((size_t)f * n_ant + ant) * n_beams + b);
                sr += e[0];
                si += e[1];
            }
        corr[idx] = v2i_g{-8 * (sr - si), -8 * (sr + si)};
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

int generic_mode(const Geometry& g) { return (g.fast_detect && g.n_ipo >= 16) ? kDetFast : g.contracted_detect ? kDetContracted : kDetCanonical; }

}  // namespace

int generic_ksteps(const Geometry& g) { return (g.n_ant + 63) / 64; }
// the fragment image is followed by the offset-nibble corrections (8 bytes per tile column)
static size_t generic_frag_bytes(const Geometry& g) { return (size_t)g.n_freq * g.n_ctiles * 3 * generic_ksteps(g) * 64 * 16; }
size_t generic_image_extra_bytes(const Geometry& g) { return (size_t)g.n_freq * g.n_ctiles * 16 * sizeof(v2i_g); }
hipError_t launch_generic_colsum(const Geometry& g, const int8_t* d_w, void* d_image, hipStream_t s)
{
    (void)hipGetLastError();
    const size_t total = (size_t)g.n_freq * g.n_ctiles * 16;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(weight_colsum_kernel, dim3(grid), dim3(256), 0, s, d_w,
                       reinterpret_cast<v2i_g*>(static_cast<char*>(d_image) + generic_frag_bytes(g)), g.n_freq, g.n_ant, g.n_beams,
                       g.n_ctiles, generic_interleave(g));
    return hipGetLastError();
}
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
    const long long S = (long long)n_units * g.n_time;
    ls.n_bgroups = (g.n_beams + 16 * kGNT * kGWaves - 1) / (16 * kGNT * kGWaves);
    const long long base = (long long)g.n_freq * ls.n_bgroups;
    const int kout = ls.rt_kout = rtw_kout(g, S, base, n_cus);   // (the stream with the least padding that still fills the chip)
    const long long Ls = (long long)kout * L;
    const int cpg = (int)((Ls + 31) / 32);
    const long long n_streams = (S + Ls - 1) / Ls;
    const long long groups = (n_streams + 3) / 4;
    ls.chunks_total = (int)(groups * cpg);
    // two 4-wave workgroups are resident per CU (128 accumulator registers per wave): two rounds of them fill the chip's tail,
    // but a workgroup should keep >= 2 chunk groups (the prologue and the B fragments of k-step 0 are paid per workgroup)
    long long want = (4LL * n_cus + base - 1) / base;
    // ... and the workgroups that are resident together should belong to FEW frequencies: every wave streams its B fragments
    // from the fragment image once per chunk, and a frequency's panel (48 KiB per k-step and 256 beams) is shared through its
    // XCD's L2 by the workgroups of that frequency only.  Block ids ascend beam group -> time split -> frequency within an XCD
    // (64 resident workgroups each): with >= 16 workgroups per frequency at most 4 panels are live per L2 (measured at 256
    // antennas, 8 gemm-units: 16 panels x 192 KiB per 4-MiB L2 -> the B stream came from the Infinity Cache, 0.39 of peak).
    const long long per_freq = ((long long)n_cus / 4 + ls.n_bgroups - 1) / ls.n_bgroups / 4;   // 16 / beam groups at 256 CUs
    if (want < per_freq) want = per_freq;
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
    a.wimg_bytes = generic_frag_bytes(g);
    a.corr = reinterpret_cast<const v2i_g*>(static_cast<const char*>(d_image) + generic_frag_bytes(g));
    a.out = d_out;
    a.n_freq = g.n_freq;
    a.n_beams = g.n_beams;
    a.n_bgroups = ls.n_bgroups;
    a.n_ctiles = g.n_ctiles;
    a.n_ant = g.n_ant;
    a.ks = generic_ksteps(g);
    a.T = g.n_time;
    a.L = g.n_ipo;
    a.kout = ls.rt_kout;
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

const char* generic_variant_key(const Geometry& g, bool write_c, char* buf, size_t n)
{
    snprintf(buf, n, "fusedg_kernel<%s, %d, %s>", g.n_ant % 16 == 0 ? "true" : "false", write_c ? kDetCanonical : generic_mode(g),
             write_c ? "true" : "false");
    return buf;
}

int generic_vgprs(const Geometry& g)
{
    hipFuncAttributes attr{};
    const void* fn = g.n_ant % 16 == 0 ? kernel_g_mode<true>(generic_mode(g)) : kernel_g_mode<false>(generic_mode(g));
    if (hipFuncGetAttributes(&attr, fn) != hipSuccess) return -1;
    return attr.numRegs;
}

}  // namespace dsabf
