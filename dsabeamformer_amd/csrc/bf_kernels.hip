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
//     sum: the reference's `shmem += x*x + y*y` evaluated without contraction (BF_DETECT_CANONICAL); BF_DETECT_CONTRACTED
//     evaluates it as nvcc's default -fmad=true does, fma(x, x, y*y); BF_DETECT_FAST see include/dsabf.h.
//
// The fused kernel itself is a template in bf_fused16.hpp, instantiated per antenna class in bf_fused16_*.hip; this file
// holds the small kernels (expand, dedisperse, weight re-layout) and the launch logic.
#include "bf_fused16.hpp"

#include <cstdio>
#include <cstdlib>

namespace dsabf {

namespace {

// ---------------------------------------------------------------------------------------------------------
// a1 alone (API parity with expand_input): byte b -> (int8)(b >> 4), (int8)((int8)(b << 4) >> 4), order kept.
// HBM-bound, 1 B in / 2 B out.  One lane = 8 packed bytes -> 16 expanded bytes, so that every wave instruction moves ONE
// contiguous run (512 B loaded, 1 KiB stored): the round-1 mapping (16 B in -> two 16-byte stores 32 B apart per lane) left
// each nontemporal store instruction covering every other 16 bytes of its lines and ran at 4.1-4.4 TB/s; this one runs at
// 6.3 TB/s algorithmic = 0.79 of 8 TB/s, the float4-copy rate of the chip (tools/ubench_expand.hip, profiles/r02_ubench_expand.txt).
__device__ __forceinline__ unsigned sext4x4(unsigned nib)  // four 4-bit values in the low nibbles of 4 bytes
{
    return ((nib ^ 0x88888888u) - 0x08080808u) ^ 0x80808080u;
}

typedef int v2i_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void expand_kernel(const v2i_t* __restrict__ in, v4i* __restrict__ out, size_t n_half)
{
    // a wave takes 128 consecutive 8-byte pieces per iteration, as two passes of 64
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, n_waves = (size_t)gridDim.x * blockDim.x / 64;
    const int lane = threadIdx.x & 63;
    for (size_t base = wave * 128; base < n_half; base += n_waves * 128) {
#pragma unroll
        for (int p = 0; p < 2; p++) {
            const size_t i = base + p * 64 + lane;
            if (i >= n_half) break;
            const v2i_t v = __builtin_nontemporal_load(in + i);
            v4i o;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const unsigned w = (unsigned)v[d];
                const unsigned hi = sext4x4((w >> 4) & 0x0F0F0F0Fu);
                const unsigned lo = sext4x4(w & 0x0F0F0F0Fu);
                o[2 * d] = (int)__builtin_amdgcn_perm(lo, hi, 0x05010400u);       // bytes (hi0, lo0, hi1, lo1)
                o[2 * d + 1] = (int)__builtin_amdgcn_perm(lo, hi, 0x07030602u);   //       (hi2, lo2, hi3, lo3)
            }
            __builtin_nontemporal_store(o, out + i);
        }
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

// a8 for every gemm-unit of a block in ONE launch (bf_enqueue_block_dedisperse): ded[u][b] = the same ascending-f fp32 sum
// over output 0 of unit u.  The order forbids splitting one beam's sum over threads, so the parallelism is units x beams:
// 32 units x 256 beams = 128 waves in one launch instead of 32 launches of 4 waves (7 us each, more than half the 12 us
// one-unit fused launch they follow in the reference's DEBUG loop).
__global__ void dedisperse_units_kernel(const float* __restrict__ out_units, size_t unit_stride, float* __restrict__ ded,
                                        int n_freq, int n_beams, int n_units)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const int u = blockIdx.y;
    if (b >= n_beams || u >= n_units) return;
    const float* out_unit = out_units + (size_t)u * unit_stride;
    float acc = 0.0f;
    int f = 0;
    for (; f + 16 <= n_freq; f += 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = out_unit[(size_t)(f + i) * n_beams + b];
#pragma unroll
        for (int i = 0; i < 16; i++) acc = acc + v[i] * 1.0f;
    }
    for (; f < n_freq; f++) acc = acc + out_unit[(size_t)f * n_beams + b] * 1.0f;
    ded[(size_t)u * n_beams + b] = acc;
}

// 8f-4: incoherent dedispersion of a detected series: out[dm][t][b] = sum over f (ascending, fp32) of
// series[t + delay[dm][f]][f][b]; rows past the end of the series contribute nothing (adding +0 is the same thing: a
// running sum that starts at +0 never becomes -0).
//
// One thread = one beam x kDmTb consecutive output times x a block of kDmBlock consecutive DM trials.  Neighbouring
// trials need almost the same input rows (their delays differ by a few samples per channel), so for each frequency the
// thread loads ONE window of kDmTb + kDmSpan consecutive rows (first row = the delay of the block's first trial) and
// every trial of the block adds its kDmTb values out of that window: 24 coalesced row loads instead of 64 per frequency
// (round 1: 8 times x 8 trials; now 16 x 4).
// The trial's offset into the window is only known at run time, and registers cannot be indexed dynamically, so the
// window lives in LDS -- as a PRIVATE column per thread ([row][thread]: conflict-free, no barrier: a thread only ever
// reads what it wrote).  The next frequency's rows are already in flight while the current window is consumed.
// Round 2 (0.76 -> 0.42 ms for 64 trials x 901 samples x 256 x 256, profiles/r02_dm_*): the round-1 loop was bound by
// instruction issue -- 258 scalar + 257 vector instructions per (wave, frequency) around 64 useful adds: 64-bit flat
// addresses per row, a range test per row, a scalar delay load + test per trial.  Now the loads are buffer loads (one
// descriptor per frequency, the row in the 32-bit offset, the series' end enforced by the descriptor's range check),
// the trials' window offsets are tabulated in LDS once per tile, and a tile whose trials all stay inside their windows
// (any fine DM ladder) runs a loop without a data-dependent branch; tiles are ordered so that the ones sharing rows run
// on one XCD (L2 hits 64 % -> 85 %, Infinity Cache / HBM fetches 2.0 -> 0.4 GB).  What binds now is LDS bandwidth
// (24 rows written + 48 read per 64 adds).  Tiles with widely spaced / non-monotonic trials or negative delays take
// the general loop (direct loads for a trial outside the window).
// 16 times x 4 trials per thread: 24 window rows written + 48 read per 64 adds (8 x 8: 24 + 56, 9 % slower)
constexpr int kDmTb = 16, kDmBlock = 4, kDmSpan = 8, kDmWin = kDmTb + kDmSpan, kDmThreads = 256;
constexpr int kDmMaxTableFreq = 1024;   // the trials' window offsets of a tile ([f][trial] ints) live in LDS up to this many channels

__global__ __launch_bounds__(kDmThreads, 4) void dedisperse_dm_kernel(const float* __restrict__ series,
                                                                   const int* __restrict__ delays, float* __restrict__ out,
                                                                   int n_t, int n_freq, int n_beams, int n_t_out, int n_dm,
                                                                   const int* __restrict__ taken)
{
    __shared__ float win[kDmWin][kDmThreads];
    // XCD-aware tile order: workgroup L runs on XCD L % 8 (round-robin dispatch), and each XCD has its own L2.  The trial
    // blocks of one time tile and the neighbouring time tiles read the same rows, so consecutive tiles (trial block
    // fastest, then time tile, then beam group) go to ONE XCD: the rows are fetched from Infinity Cache / HBM once, not once
    // per XCD, and the chain of dependent row loads runs at L2-hit latency.
    const int n_x = (n_dm + kDmBlock - 1) / kDmBlock, n_y = (n_t_out + kDmTb - 1) / kDmTb;
    const int per_xcd = gridDim.x / 8;
    const int v = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (v >= n_x * n_y * ((n_beams + kDmThreads - 1) / kDmThreads)) return;
    const int dm0 = (v % n_x) * kDmBlock;
    static_assert(kDwTrials % kDmBlock == 0, "a trial block lies inside one group of the wide kernel");
    if (taken && taken[dm0 / kDwTrials]) return;   // dedisperse_dm_wide_kernel (bf_dm_wide.hip) has this trial group
    const int t0 = ((v / n_x) % n_y) * kDmTb;
    const int tid = threadIdx.x;
    const int b_raw = (v / (n_x * n_y)) * kDmThreads + tid;
    const bool live = b_raw < n_beams;
    const int b = live ? b_raw : n_beams - 1;
    float acc[kDmBlock][kDmTb];
#pragma unroll
    for (int k = 0; k < kDmBlock; k++)
#pragma unroll
        for (int i = 0; i < kDmTb; i++) acc[k][i] = 0.0f;
    const size_t row_stride = (size_t)n_freq * n_beams;
    const int* dl0 = delays + (size_t)dm0 * n_freq;
    float nxt[kDmWin];
    // Addressing stays on the scalar unit: per frequency ONE buffer descriptor whose base is the first row of the window
    // (wave-uniform), the row as an SGPR offset (j * row stride), the beam as the one per-lane 32-bit offset -- a load
    // costs no VALU instruction (flat loads need a 64-bit v_mad per row).
    const int lane_bytes = b * 4;
    const unsigned stride_bytes = (unsigned)row_stride * 4u;   // launch_dedisperse_dm checks kDmWin * stride < 4 GiB

    // Is this tile REGULAR -- every window starting inside the series and every trial inside its window, for every
    // frequency?  (Always, except for widely spaced or non-monotonic trials or negative delays.)  One cooperative pass
    // over the block's delays decides, and leaves the trials' window offsets in LDS ([f][trial], bytes).  A regular tile
    // then runs a loop without a data-dependent branch and without a scalar load between its LDS reads (SMEM and LDS
    // share one counter and SMEM returns out of order: a delay fetched by s_load inside the trial loop forces
    // s_waitcnt lgkmcnt(0) per trial, which serialises the LDS round trips).  The kernel is bound by instruction issue
    // and by these latencies, and -- all workgroups being resident at once -- by its SLOWEST tile, so the ends of the
    // series must not be a slow path: rows past the end are dropped by the buffer descriptor's range check (they read as
    // +0, which is what the definition asks for), a partial trial block repeats its last trial (never stored).
    extern __shared__ __attribute__((aligned(16))) int trial_off[];   // [n_freq][kDmBlock]; only filled / used when the table fits (see the launcher)
    const int* dlk[kDmBlock];
#pragma unroll
    for (int k = 0; k < kDmBlock; k++) dlk[k] = delays + (size_t)min(dm0 + k, n_dm - 1) * n_freq;
    int irregular = n_freq > kDmMaxTableFreq;
    if (!irregular)
        for (int f = tid; f < n_freq; f += kDmThreads) {
            const int base = dl0[f];
            irregular |= t0 + base < 0;
#pragma unroll
            for (int k = 0; k < kDmBlock; k++) {
                const int d = dlk[k][f] - base;
                irregular |= d < 0 || d > kDmSpan;
                trial_off[f * kDmBlock + k] = d * (kDmThreads * 4);
            }
        }
    irregular = __syncthreads_or(irregular);

    if (!irregular) {
        static_assert(kDmBlock == 4 || kDmBlock == 8, "one or two 16-byte table reads per frequency");
        const char* my_col = reinterpret_cast<const char*>(&win[0][tid]);
        const size_t series_floats = (size_t)n_t * row_stride;
        auto load_window = [&](int first, int f) {
            // descriptor: base = row `first` of channel f, records = what is left of the series from there (capped: rows
            // of a window are < 4 GiB apart); the per-lane offset carries the row so that the range check sees it
            const size_t at = (size_t)first * row_stride + (size_t)f * n_beams;
            const size_t left = at < series_floats ? (series_floats - at) * 4 : 0;   // a window wholly past the end reads +0
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(series + (left ? at : 0)), 0, (int)(left < 0xfffffff0u ? left : 0xfffffff0u), 0x00020000);
#pragma unroll
            for (int j = 0; j < kDmWin; j++)
                nxt[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, lane_bytes + (int)(j * stride_bytes), 0, 0));
        };
        load_window(t0 + __builtin_amdgcn_readfirstlane(dl0[0]), 0);
        for (int f = 0; f < n_freq; f++) {
            const int fn = min(f + 1, n_freq - 1);   // the last iteration re-reads its own window: no branch
            const int first_next = t0 + __builtin_amdgcn_readfirstlane(dl0[fn]);
#pragma unroll
            for (int j = 0; j < kDmWin; j++) win[j][tid] = nxt[j];
#pragma unroll
            for (int i = 0; i < kDmTb; i++) acc[0][i] = acc[0][i] + nxt[i];   // trial 0 sits at offset 0: from registers
            load_window(first_next, fn);
            int off[kDmBlock];
#pragma unroll
            for (int q = 0; q < kDmBlock / 4; q++) {
                const v4i o = *reinterpret_cast<const v4i*>(&trial_off[f * kDmBlock + 4 * q]);
                off[4 * q] = o.x, off[4 * q + 1] = o.y, off[4 * q + 2] = o.z, off[4 * q + 3] = o.w;
            }
#pragma unroll
            for (int k = 1; k < kDmBlock; k++) {
                const float* w = reinterpret_cast<const float*>(my_col + off[k]);
#pragma unroll
                for (int i = 0; i < kDmTb; i++) acc[k][i] = acc[k][i] + w[i * kDmThreads];
            }
        }
    } else {
        auto load_window = [&](int f) {
            const float* col = series + (size_t)f * n_beams;   // uniform
            const int first = t0 + __builtin_amdgcn_readfirstlane(dl0[f]);
#pragma unroll
            for (int j = 0; j < kDmWin; j++) {
                const int r = first + j;
                nxt[j] = (r >= 0 && r < n_t) ? (col + (size_t)r * row_stride)[b] : 0.0f;
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
                        const float x = (r >= 0 && r < n_t) ? p[(size_t)r * row_stride] : 0.0f;
                        acc[k][i] = acc[k][i] + x;
                    }
                }
            }
        }
    }
    if (!live) return;
#pragma unroll
    for (int k = 0; k < kDmBlock; k++)
#pragma unroll
        for (int i = 0; i < kDmTb; i++)
            if (dm0 + k < n_dm && t0 + i < n_t_out) out[((size_t)(dm0 + k) * n_t_out + t0 + i) * n_beams + b] = acc[k][i];
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

// Paired weight image: image[f][pct][comp][h][lane] (16 bytes): lane = 16*kb + c; byte i = Wr (comp 0) or Wi (comp 1) of
// antenna 64*h + 16*kb + i (zero behind the last antenna) for base beam beam_of_tile(pct, c) (< n_beams / 2).
__global__ void weight_relayout16p_kernel(const int8_t* __restrict__ w, v4i* __restrict__ image, int n_freq, int n_ant,
                                          int n_beams, int ks, int interleave)
{
    const int n_pct = n_beams / 32;
    const size_t total = (size_t)n_freq * n_pct * kPairComps * ks * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        size_t r = idx >> 6;
        const int h = (int)(r % ks);
        r /= ks;
        const int comp = (int)(r % kPairComps);
        r /= kPairComps;
        const int pct = (int)(r % n_pct);
        const int f = (int)(r / n_pct);
        const int kb = lane >> 4, b = beam_of_tile(interleave, pct, lane & 15);
        unsigned d[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
            const int ant = 64 * h + kb * 16 + i;
            int v = 0;
            if (ant < n_ant) v = w[2 * (((size_t)f * n_ant + ant) * n_beams + b) + (comp ? 1 : 0)];
            d[i >> 2] |= ((unsigned)v & 0xFFu) << (8 * (i & 3));
        }
        image[idx] = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
    }
}

// 16x16x64 weight image: image[f][ct16][comp][h][lane] (16 bytes): lane = 16*kb + c; byte i = Wr (comp 0), -Wi (comp 1) or
// Wi (comp 2) of antenna 64*h + 16*kb + i (zero behind the last antenna) for beam beam_of_tile(ct16, c).  The real row of
// the embedding multiplies (Vr | Vi) by (Wr | -Wi), the imaginary row by (Wi | Wr): Wr serves both.
__global__ void weight_relayout16_kernel(const int8_t* __restrict__ w, v4i* __restrict__ image, int n_freq, int n_ant,
                                         int n_beams, int ks, int interleave, int* __restrict__ bad)
{
    const int n_ct = (n_beams + 15) / 16;   // the last tile may be partly filled: zero weights behind the last beam
    constexpr int NGC = kGeneralComps;
    const size_t total = (size_t)n_freq * n_ct * NGC * ks * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        size_t r = idx >> 6;
        const int h = (int)(r % ks);
        r /= ks;
        const int comp = (int)(r % NGC);
        r /= NGC;
        const int ct = (int)(r % n_ct);
        const int f = (int)(r / n_ct);
        const int kb = lane >> 4, b = beam_of_tile(interleave, ct, lane & 15);
        unsigned d[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
            const int ant = 64 * h + kb * 16 + i;
            int v = 0;
            if (ant < n_ant && b < n_beams) {
                const int8_t* e = w + 2 * (((size_t)f * n_ant + ant) * n_beams + b);
                const int wr = e[0], wi = e[1];
                if (wi == -128) *bad = 1;
                v = (comp == 1) ? -wi : (comp == 2) ? wi : wr;
            }
            d[i >> 2] |= ((unsigned)v & 0xFFu) << (8 * (i & 3));
        }
        image[idx] = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
    }
}


// ---- which instantiation a geometry runs ---------------------------------------------------------------------------------
int ksteps16(const Geometry& g) { return (g.n_ant + 63) / 64; }
// The deep classes of fused16_kernel (three / four k-steps, bf_fused16.hpp): 129 ... 256 antennas in 16-byte rows, windows of
// 16 / 32 / 64 samples.  (The stage-parity launch and every other geometry beyond 128 antennas run fusedg_kernel.)
bool deep_class(const Geometry& g)
{
    // (rows that are only dword-aligned -- 132, 140, ... antennas -- from round 5 on, in windows of 16 and 32 samples)
    return !g.force_generic && !g.no_deep && g.n_ant > 128 && g.n_ant <= 256 && g.n_ant % 4 == 0 &&
           (g.n_ipo == 16 || g.n_ipo == 32 || (g.n_ipo == 64 && g.n_ant % 16 == 0));
}
// MFMA column tiles per wave the beams are dealt to round-robin (beam_of_tile), 0 = tile t is beams 16 t ...: interleaved when every
// wave owns whole groups of 16 * NS beams.  paired: the layout of the conjugate-pair image / kernel (NS / 2 pair tiles per wave).
int interleaved(const Geometry& g, bool paired)
{
    if (use_generic(g)) return paired ? 0 : generic_interleave(g);
    const int ns = fused_col_tiles(g, paired);
    if (g.n_beams % (16 * ns)) return 0;
    return paired ? ns / 2 : ns;
}
bool nipo_supported(int n_ipo) { return n_ipo == 2 || n_ipo == 4 || n_ipo == 8 || n_ipo == 16 || n_ipo == 32 || n_ipo == 64; }
// Accumulation windows that have no compile-time instantiation (or short windows in gemm-units that are not whole 16-sample
// runs) run fused16_kernel's run-time-window instantiations (template NIPO = 0), up to 128 antennas; DSABF_RTW=0: fusedg_kernel.
bool rtw_class(const Geometry& g)
{
    return !g.force_generic && !g.no_rtw && g.n_ant <= 128 && g.n_ipo > 0 && (!nipo_supported(g.n_ipo) || (g.n_ipo < 16 && g.n_time % 16));
}
int detect_mode_of(const Geometry& g) { return g.fast_detect ? kDetFast : g.contracted_detect ? kDetContracted : kDetCanonical; }

// 100 antennas (BASELINE config 5) have compile-time instantiations: 3 - 7 % faster than the dword-staged run-time class that covers
// the count.  Every other multiple of 4 up to 256 -- the reference's 64 included -- runs the run-time-count class of its k-step count
// and row alignment: the compile-time classes of 64 / 128 / 192 / 256 antennas that rounds 1-5 carried measured inside the box noise
// of the run-time ones (-1.4 ... +1.9 %, one geometry nobody names excepted: profiles/r06_class_fold_ab.txt) and were folded into them.
FusedVariant select_variant(const Geometry& g, bool write_c)
{
    if (use_generic(g)) return FusedVariant{};
    if (deep_class(g)) {
        if (write_c) return FusedVariant{};
        const bool paired = g.paired;
        const int mode = detect_mode_of(g);
        const int ns = fused_col_tiles(g, paired);
        if (g.n_ant % 16) return g.n_ant > 192 ? fused16_variant_k4p4(g.n_ipo, mode, paired, ns) : fused16_variant_k3p4(g.n_ipo, mode, paired, ns);
        return g.n_ant > 192 ? fused16_variant_k4p16(g.n_ipo, mode, paired, ns) : fused16_variant_k3p16(g.n_ipo, mode, paired, ns);
    }
    if ((!nipo_supported(g.n_ipo) && !rtw_class(g)) || g.n_ant <= 0 || g.n_ant > 128 || g.n_ant % 4) return FusedVariant{};
    const bool paired = g.paired && !write_c;
    const int sel_ipo = rtw_class(g) ? 0 : g.n_ipo;     // 0: the run-time-window instantiations (every fused16_variant's default)
    const int mode = (detect_mode_of(g) == kDetFast && g.n_ipo < 16) ? kDetCanonical : detect_mode_of(g);   // fast: from 16 samples on
    if (fused_col_tiles(g, paired) == kColTilesWide16) {
        if (g.n_ant == 100) return fused16_variant_a100_s8(g.n_ipo, mode);
        return g.n_ant % 16 == 0 ? fused16_variant_k2p16_s8(g.n_ipo, mode) : fused16_variant_k2p4_s8(g.n_ipo, mode);
    }
    if (fused_wg_waves(g, write_c) == kWavesWide16) {
        if (paired) {
            if (g.n_ant == 100) return fused16_variant_a100_w8p(g.n_ipo, mode);
            return g.n_ant % 16 == 0 ? fused16_variant_k2p16_w8p(g.n_ipo, mode) : fused16_variant_k2p4_w8p(g.n_ipo, mode);
        }
        if (g.n_ant == 100) return fused16_variant_a100_w8(g.n_ipo, mode);
        return g.n_ant % 16 == 0 ? fused16_variant_k2p16_w8(g.n_ipo, mode) : fused16_variant_k2p4_w8(g.n_ipo, mode);
    }
    if (g.n_ant == 100) return fused16_variant_a100(sel_ipo, write_c, mode, paired);
    if (g.n_ant <= 64)
        return g.n_ant % 16 == 0 ? fused16_variant_k1p16(sel_ipo, write_c, mode, paired) : fused16_variant_k1p4(sel_ipo, write_c, mode, paired);
    return g.n_ant % 16 == 0 ? fused16_variant_k2p16(sel_ipo, write_c, mode, paired) : fused16_variant_k2p4(sel_ipo, write_c, mode, paired);
}

int ilog2_exact(int v)
{
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) s++;
    return s;
}

hipError_t dispatch_fused(const Geometry& g, bool write_c, const FusedArgs& args, const LaunchShape& ls, hipStream_t s)
{
    const FusedVariant v = select_variant(g, write_c);
    if (!v.launch) return hipErrorInvalidValue;
    return v.launch(args, ls, s);
}

}  // namespace

// fusedg_kernel takes what the specialised instantiations of fused16_kernel do not cover: more than two k-steps, accumulation
// windows that are not a power of two (or longer than 64), short windows in gemm-units that are not whole 16-sample runs.
bool use_generic(const Geometry& g)
{
    if (g.force_generic || (g.n_ant > 128 && !deep_class(g))) return true;
    return (!nipo_supported(g.n_ipo) || (g.n_ipo < 16 && g.n_time % 16)) && !rtw_class(g);
}

size_t weight_image_bytes(const Geometry& g)
{
    return (size_t)g.n_freq * g.n_ctiles * kGeneralComps * ksteps16(g) * 64 * 16 +   // [f][ct16][Wr, -Wi, Wi][k-step][lane] x 16 B
           ((use_generic(g) || deep_class(g)) ? generic_image_extra_bytes(g) : 0);
}

// the conjugate-pair kernel works on tiles of 16 base beams + their 16 mirror images
bool pairing_supported(const Geometry& g)
{
    return !use_generic(g) && g.n_beams % 32 == 0 && select_variant(g, false).launch != nullptr;
}
size_t weight_pair_image_bytes(const Geometry& g)
{
    return pairing_supported(g) ? (size_t)g.n_freq * (g.n_beams / 32) * kPairComps * ksteps16(g) * 64 * 16 : 0;
}

bool fused_supported(const Geometry& g, const char** why)
{
    const char* dummy;
    if (!why) why = &dummy;
    if (g.n_beams <= 0 || g.n_beams % 4) { *why = "N_BEAMS must be a positive multiple of 4"; return false; }      // src/beamformer.hh:155
    if (g.n_ant <= 0 || g.n_ant % 4) { *why = "N_ANTENNAS must be a positive multiple of 4"; return false; }       // src/beamformer.hh:156
    if (use_generic(g)) return generic_supported(g, why);
    return true;
}

// Measurement / test switches from the ENVIRONMENT: honoured only in a process that says DSABF_LAB=1 (tools/, the A/B tests).  A
// production host's stray DSABF_* variable selects nothing (tests/test_abi_cpu.py greps csrc/ for getenv against the allow-list
// of INTEGRATION.md); inside one process bf_set_switch (dsabf_bench.h) does the same per handle.
bool lab_mode()
{
    const char* lab = getenv("DSABF_LAB");
    return lab && lab[0] == '1' && !lab[1];
}
const char* lab_getenv(const char* name) { return lab_mode() ? getenv(name) : nullptr; }

void read_env_switches(Geometry& g)
{
    const char* w = lab_getenv("DSABF_WG_WAVES");
    const char* t = lab_getenv("DSABF_COL_TILES");
    g.plain_wg_waves = w && atoi(w) == kWaves16;
    g.plain_col_tiles = t && atoi(t) == kColTiles16;
    const char* ts = lab_getenv("DSABF_TSPLIT");
    const char* pad = lab_getenv("DSABF_LDS_PAD");
    const char* dw = lab_getenv("DSABF_DM_WIDE");
    g.tsplit = ts ? atoi(ts) : 0;
    if (g.tsplit < 0) g.tsplit = 0;
    g.lds_pad = pad ? atoi(pad) : 0;       // (clamped to what the CU has left where it is applied, fused_launch_shape)
    g.dm_wide = !(dw && dw[0] == '0');
    const char* gen = lab_getenv("DSABF_GENERIC");
    g.force_generic = gen && gen[0] == '1';
    const char* nd = lab_getenv("DSABF_DEEP");
    g.no_deep = nd && nd[0] == '0';
    const char* nr = lab_getenv("DSABF_RTW");
    g.no_rtw = nr && nr[0] == '0';
}

// Output slots (16 beams each) per wave.  The two-k-step conjugate-pair kernels hold 2 waves per SIMD whatever they do (64 KiB of
// LDS per workgroup), and with 8 slots instead of 4 a wave's LDS fragment reads feed twice the MFMAs (32 between two reads) at 241
// of its 256 registers: BASELINE config 5 runs 6.4-7.0 % faster than on 8-wave workgroups, 8.5 % faster than on the 4-slot
// 4-wave launch (profiles/r03_ab_c5_ns8.txt).  Needs whole workgroups of 4 x 128 beams; the general kernel would spill (192
// registers of weight fragments alone) and keeps 4; so do the run-time dword-staged class and 100 antennas at n_ipo 64 (they would
// spill, ns8_fits) and the one-k-step classes, where 8 slots cost two of the four resident waves per SIMD (+6 % time,
// profiles/r03_variants_log.txt).
int fused_col_tiles(const Geometry& g, bool paired)
{
    if (deep_class(g)) return (paired && g.n_beams % 512 == 0 && !g.plain_col_tiles) ? 4 : 2;   // pair: 2 pair tiles per wave where 8 waves x 64 beams fill
    // the instantiations that fit their registers (ns8_fits, bf_fused16.hpp): 16-byte-staged rows, or the compile-time 100 antennas
    const bool fits = g.n_ant % 16 == 0 || (g.n_ant == 100 && g.n_ipo < 64);
    const bool can = paired && fits && ksteps16(g) == 2 && g.n_ipo >= 16 && nipo_supported(g.n_ipo) && g.n_beams % 512 == 0;
    return (can && !g.plain_col_tiles) ? kColTilesWide16 : kColTiles16;
}

// A workgroup stages one frequency's voltages for all of its waves.  The two-k-step classes hold 2 waves per SIMD whatever
// the workgroup size (their registers), so where the beams fill them, 8-wave workgroups -- one per CU instead of two -- stage
// and read every voltage once per 512 beams instead of once per 256: BASELINE config 5 runs 2 % faster (pair and general
// kernel, profiles/r03_ab_c5_w8.txt).  The one-k-step classes keep 4 (their 3-4 resident workgroups overlap each other's
// barriers); so do the store-bound short windows, the stage-parity launch and the 8-slot pair kernel above.
int fused_wg_waves(const Geometry& g, bool write_c)
{
    if (deep_class(g) && !write_c) return kWavesWide16;
    const bool can = !write_c && ksteps16(g) == 2 && g.n_ipo >= 16 && nipo_supported(g.n_ipo) && ((g.n_beams + 255) / 256) % 2 == 0 &&
                     fused_col_tiles(g, g.paired) == kColTiles16;
    return (can && !g.plain_wg_waves) ? kWavesWide16 : kWaves16;
}

// Run-time window: a lane group's stream is kout whole windows = kout * L rows over cpg = ceil(kout L / 32) chunks of 32; the rows
// between the stream's end and 32 cpg are padding the MFMAs multiply all the same (L = 24 alone in a chunk: a quarter of the
// launch; L = 40 alone in two: three eighths).  kout * L a multiple of 32 wastes nothing (kout = 32 / gcd(L, 32) always is one),
// but a group of 4 streams is also the unit the launch is split by: the kout with the fewest chunks in total among those that
// still leave 8 groups per CU for the chip, the smallest such; a small launch keeps what fits one chunk.
int rtw_kout(const Geometry& g, long long S, long long base, int n_cus)
{
    const int L = g.n_ipo;
    if (g.rtw_kout > 0) return g.rtw_kout;
    const long long W = (S + L - 1) / L;   // windows per frequency
    auto groups = [&](int k) { return (W + 4LL * k - 1) / (4LL * k); };
    auto chunks = [&](int k) { return groups(k) * (((long long)k * L + 31) / 32); };
    int gcd = 32;
    while (L % gcd) gcd >>= 1;
    int best = L <= 32 ? 32 / L : 1;       // what fits one chunk (one window over ceil(L / 32) chunks): the finest split there is
    for (int k = best + 1; k <= 32 / gcd; k++)
        if (base * groups(k) >= 8LL * n_cus && chunks(k) < chunks(best)) best = k;
    return best;
}

LaunchShape fused_launch_shape(const Geometry& g, int n_units, int n_cus, bool write_c)
{
    if (use_generic(g)) return generic_launch_shape(g, n_units, n_cus);
    LaunchShape ls{};
    const int wg_waves = fused_wg_waves(g, write_c);
    const int beams_per_wg = wg_waves * 16 * fused_col_tiles(g, g.paired && !write_c);
    ls.n_bgroups = (g.n_beams + beams_per_wg - 1) / beams_per_wg;
    const long long S = (long long)n_units * g.n_time;
    long long rows;
    int cpg = 1;  // chunks per output group: a workgroup's chunk range must cover whole groups
    if (rtw_class(g)) {                                  // run-time window: streams of kout whole windows, 32 rows per chunk
        const int kout = ls.rt_kout = rtw_kout(g, S, (long long)g.n_freq * ls.n_bgroups, n_cus);
        const long long Ls = (long long)kout * g.n_ipo;
        cpg = (int)((Ls + 31) / 32);
        const long long groups = ((S + Ls - 1) / Ls + 3) / 4;
        rows = groups * cpg * kRowsPerChunk;
    } else if (g.n_ipo >= 16) {
        const long long groups = (S / g.n_ipo + 3) / 4;  // a lane group carries one output: 4 outputs advance together
        rows = groups * 4 * g.n_ipo;
        if (4 * g.n_ipo > kRowsPerChunk) cpg = 4 * g.n_ipo / kRowsPerChunk;
    } else {
        rows = (S + 15) / 16 * 16;                       // 16-sample runs
    }
    ls.chunks_total = (int)((rows + kRowsPerChunk - 1) / kRowsPerChunk);
    ls.chunks_total = (ls.chunks_total + cpg - 1) / cpg * cpg;
    const int base = g.n_freq * ls.n_bgroups;
    // Time splits per frequency.  Measured (profiles/r01_variants_log.txt): the kernel is fastest with ~16-32 chunks
    // per workgroup (long enough to amortise the weight-fragment load and the prologue, short enough that the tail
    // of the launch is fine-grained); small launches still get ~2 workgroups per resident slot, but never fewer
    // than 2 chunks each.
    const int groups_avail = ls.chunks_total / cpg;
    // At least 2 chunk-groups per workgroup (the weight-fragment load and the prologue are paid per workgroup; two
    // k-steps = twice the fragments: at least 4) -- unless that leaves fewer than 4 (2) workgroups per CU: a single
    // gemm-unit is 4 chunks per frequency, 1 chunk each is then 6 % faster (profiles/r02_launch_size.txt).
    const bool two_k = ksteps16(g) >= 2;
    const bool wide = wg_waves == kWavesWide16;   // one resident workgroup per CU
    const bool ns8 = fused_col_tiles(g, g.paired && !write_c) == kColTilesWide16;   // two resident, long like the wide ones
    const int min_groups = two_k ? 4 : 2;
    int max_split = groups_avail >= min_groups ? groups_avail / min_groups : 1;
    if ((long long)base * max_split < (wide ? 1LL : two_k ? 2LL : 4LL) * n_cus) max_split = groups_avail >= 1 ? groups_avail : 1;
    // ~20 chunk-groups per workgroup; the 8-wave workgroups (one resident per CU) the longer the better: BASELINE config 5 on
    // 1024 workgroups of 32 chunks -2.0 % (general kernel -3.1 %) against 4-wave workgroups, on 2048 of 16 -1.3 % (-2.1 %),
    // on 4096 of 8 +1.3 % (profiles/r03_ab_c5_w8.txt)
    int want = (wide || ns8) ? (groups_avail + 20) / 40 : (groups_avail + 10) / 20;
    // Short windows (n_ipo < 16) are store-bound: fewer, longer workgroups measured better (C2: 2 per CU 0.54 of the
    // HBM peak, 8 per CU 0.49).  The MFMA-bound one-k-step shapes want ~4 resident sets of 4: a 32-unit block (one
    // PSRDADA block, bf_enqueue_block) runs 4 % faster on 4096 workgroups than on 2048 (profiles/r02_launch_size.txt);
    // two-k-step shapes (2 resident per CU, 128 KiB of weight fragments each) want 2 sets of 2: a 16-unit launch of a
    // BASELINE-config-5 rank shard is 14 % faster on 1024 workgroups than on 4096 (profiles/r02_launch_size_c5shard.txt).
    // Their 8-wave workgroups, one resident per CU, want one set: the same shard 256 workgroups -7.0 %, 512 -4.9 %, 1024 -1.2 %
    // against the 4-wave launch (profiles/r03_ab_c5_w8.txt).
    // The 8-slot pair kernel (two resident per CU) wants one set of two: the shard on 512 workgroups -12.1 %, on 1024 -10.6 %, on
    // 256 (half the chip's slots empty) +35 % (profiles/r03_ab_c5_ns8.txt).
    const int target_wgs_per_cu = g.n_ipo < 16 ? 2 : wide ? 1 : ns8 ? 2 : (two_k ? 4 : 4 * (16 / kWaves16));
    int want_fill = (target_wgs_per_cu * n_cus + base - 1) / base;            // enough workgroups to fill the chip
    if (want_fill > max_split) want_fill = max_split;
    if (want < want_fill) want = want_fill;
    if (g.tsplit > 0) want = g.tsplit;  // tuning override (time splits per frequency): DSABF_TSPLIT at bf_create / bf_set_switch
    if (want < 1) want = 1;
    if (want > ls.chunks_total / cpg) want = ls.chunks_total / cpg;
    ls.n_tsplit = want;
    ls.grid = base * ls.n_tsplit;
    ls.block = 64 * wg_waves;
    ls.lds_bytes = 2 * ksteps16(g) * kRowsPerChunk * 128;  // double buffer x k-step planes x 128 rows x (64 re | 64 im)
    if (g.lds_pad > 0)   // measurement switch (DSABF_LDS_PAD / bf_set_switch): fewer resident workgroups per CU
        ls.lds_bytes += g.lds_pad < kLdsPerCuBytes - ls.lds_bytes ? g.lds_pad : kLdsPerCuBytes - ls.lds_bytes;
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
    a.n_ctiles = g.n_ctiles;
    a.n_ptiles = g.n_beams / 32;
    a.n_ant = g.n_ant;
    a.n_bgroups = ls.n_bgroups;
    a.T = g.n_time;
    a.t_shift = ilog2_exact(g.n_time);
    a.S = (unsigned)((long long)n_units * g.n_time);
    a.chunks_total = ls.chunks_total;
    a.n_tsplit = ls.n_tsplit;
    a.interleave = interleaved(g, g.paired);
    if (rtw_class(g)) {
        a.rt_L = g.n_ipo;
        a.rt_kout = ls.rt_kout;
        a.rt_Ls = a.rt_kout * a.rt_L;
        a.rt_cpg = (a.rt_Ls + 31) / 32;
    }
    return a;
}

hipError_t launch_fused(const Geometry& g, const void* d_image, const void* d_pair_image, const void* d_packed,
                        int n_units, float* d_out, int n_cus, hipStream_t s)
{
    if (n_units <= 0) return hipSuccess;
    if (use_generic(g)) return launch_fused_generic(g, d_image, d_packed, n_units, d_out, n_cus, false, s);
    if ((long long)n_units * g.n_time > 0x7fffffffLL / 2) return hipErrorInvalidValue;
    if (g.paired && !(pairing_supported(g) && d_pair_image)) return hipErrorInvalidValue;
    const LaunchShape ls = fused_launch_shape(g, n_units, n_cus);
    const FusedArgs a = make_args(g, g.paired ? d_pair_image : d_image, d_packed, n_units, d_out, ls);
    return dispatch_fused(g, false, a, ls, s);
}

hipError_t launch_gemm_only(const Geometry& g, const void* d_image, const void* d_packed, float* d_c, int n_cus,
                            hipStream_t s)
{
    Geometry gg = g;
    gg.paired = false;  // the stage-parity path always runs the general kernel on the general image
    if (use_generic(gg) || deep_class(gg))   // (the deep classes share fusedg_kernel's image: same k-steps, same 2-beam interleave)
        return launch_fused_generic(gg, d_image, d_packed, 1, d_c, n_cus, true, s);
    const LaunchShape ls = fused_launch_shape(gg, 1, n_cus, true);
    const FusedArgs a = make_args(gg, d_image, d_packed, 1, d_c, ls);
    return dispatch_fused(gg, true, a, ls, s);
}

// hipGetLastError() also returns errors left behind by EARLIER, unrelated calls of the thread (a failed
// hipGetDeviceProperties of a caller, say): every launcher clears the slot first so that what it returns is its own.
static inline void clear_stale_error() { (void)hipGetLastError(); }

hipError_t launch_weight_relayout(const Geometry& g, const int8_t* d_w, void* d_image, void* d_pair_image, int* d_bad,
                                  hipStream_t s)
{
    clear_stale_error();
    const bool with_corr = use_generic(g) || deep_class(g);   // the generic kernel's offset-nibble corrections behind the fragments
    const size_t total = (weight_image_bytes(g) - (with_corr ? generic_image_extra_bytes(g) : 0)) / 16;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    if (with_corr) {
        hipError_t e = launch_generic_colsum(g, d_w, d_image, s);
        if (e != hipSuccess) return e;
    }
    if (pairing_supported(g) && d_pair_image) {
        const size_t n_fa = (size_t)g.n_freq * g.n_ant;
        int pgrid = (int)((n_fa * (g.n_beams / 2) + 255) / 256);
        if (pgrid > 4096) pgrid = 4096;
        hipLaunchKernelGGL(pair_check_kernel, dim3(pgrid), dim3(256), 0, s, d_w, n_fa, g.n_beams, d_bad + 1);
        int rgrid = (int)((weight_pair_image_bytes(g) / 16 + 255) / 256);
        if (rgrid > 4096) rgrid = 4096;
        hipLaunchKernelGGL(weight_relayout16p_kernel, dim3(rgrid), dim3(256), 0, s, d_w, static_cast<v4i*>(d_pair_image),
                           g.n_freq, g.n_ant, g.n_beams, ksteps16(g), interleaved(g, true));
    }
    hipLaunchKernelGGL(weight_relayout16_kernel, dim3(grid), dim3(256), 0, s, d_w, static_cast<v4i*>(d_image), g.n_freq,
                       g.n_ant, g.n_beams, ksteps16(g), interleaved(g, false), d_bad);
    return hipGetLastError();
}

hipError_t launch_expand(const void* d_in, size_t nbytes, void* d_out, hipStream_t s)
{
    clear_stale_error();
    const size_t n_half = nbytes / 8;
    if (n_half == 0) return hipSuccess;
    size_t grid = (n_half + 511) / 512;   // one iteration per wave if the grid is not capped
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(expand_kernel, dim3((unsigned)grid), dim3(256), 0, s, static_cast<const v2i_t*>(d_in),
                       static_cast<v4i*>(d_out), n_half);
    return hipGetLastError();
}

// ---- the matrix pipe by itself ("a back-to-back v_mfma micro-benchmark; report utilisation against both nominal
// and measured peak") -------------------------------------------------------------------------------------------------------------
// 4 waves per SIMD, two accumulator chains of v_mfma_i32_16x16x64_i8 per wave issued chain by chain (8 dependent MFMAs on one
// accumulator, then 8 on the other) and nothing else in the loop: the order the pipe runs fastest in (tools/ubench_chains.hip,
// profiles/r04_ubench_chains.txt: 1-2 chains 0.94-0.95 of 5.0 POP/s, 16 chains of 2 round robin -- this kernel up to round 3 --
// 0.63).  The operands are the caller's bytes -- A as the fused kernel sees voltages (16 * nibble), B as it sees weights (any
// int8) -- because the clock the chip holds under this load depends on the operand bits (tools/ubench_shape.hip).  The
// accumulators run on across iterations (int32 wrap-around is harmless here) and are stored so that nothing is optimised away.
__global__ __launch_bounds__(256, 4) void mfma_peak_kernel(const v4i* __restrict__ src, int* __restrict__ sink, int iters)
{
    v4i a[4], b[8];
#pragma unroll
    for (int i = 0; i < 8; i++) b[i] = src[((blockIdx.x & 63) * 8 + i) * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const v4i r = src[(512 + (blockIdx.x & 63) * 4 + i) * 256 + threadIdx.x];
        a[i] = v4i{r[0] & (int)0xF0F0F0F0u, r[1] & (int)0xF0F0F0F0u, r[2] & (int)0xF0F0F0F0u, r[3] & (int)0xF0F0F0F0u};
    }
    v4i c[2] = {v4i{0, 0, 0, 0}, v4i{0, 0, 0, 0}};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int m = 0; m < 8; m++) {
                c[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[(t + m) & 3], b[(m + 3 * t) & 7], c[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[0][3] + c[1][1] + c[1][2];
}

// src: kMfmaPeakSrcBytes of caller data; sink: kMfmaPeakSinkBytes of scratch.  *ops = int8 ops (2 per MAC) the launch executes.
hipError_t launch_mfma_peak(const void* d_src, void* d_sink, int iters, int n_cus, double* ops, hipStream_t s)
{
    clear_stale_error();
    const int grid = n_cus * 4;   // 4 workgroups of 4 waves per CU = 4 waves per SIMD, one set
    if ((size_t)grid * 256 * sizeof(int) > kMfmaPeakSinkBytes || iters <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(grid), dim3(256), 0, s, static_cast<const v4i*>(d_src), static_cast<int*>(d_sink), iters);
    if (ops) *ops = (double)grid * 4.0 * iters * 16.0 * 2.0 * 16 * 16 * 64;
    return hipGetLastError();
}

// The staged transport of the gather (bf_gather_detected_staged): the shards arrive sub-band-major,
// stage[rank][row][row_floats] -- ONE message per sender -- and this pass puts them where the reference's [o][f][b] order has
// them, full[row][rank][row_floats] (f = rank * n_freq_local + f_local).  Pure HBM traffic: every float is read once and
// written once, in whole 16-byte pieces of 128-byte lines, nontemporal both ways (the data is not touched again here).
// A workgroup moves whole (rank, row) blocks, consecutive workgroups consecutive rows of one rank: reads stream through
// the stage linearly, writes land world * row_floats apart.  skip_rank: the receiver's own rows are copied straight from
// its kernel output to their final place and never staged.
__global__ __launch_bounds__(256) void gather_relayout_kernel(const v4i* __restrict__ stage, v4i* __restrict__ full, unsigned held,
                                                              unsigned world, unsigned row_vec, int skip_rank)
{
    const size_t n_blocks = (size_t)held * world;
    for (size_t blk = blockIdx.x; blk < n_blocks; blk += gridDim.x) {
        const unsigned r = (unsigned)(blk / held), h = (unsigned)(blk % held);
        if ((int)r == skip_rank) continue;
        const v4i* src = stage + blk * row_vec;
        v4i* dst = full + ((size_t)h * world + r) * row_vec;
        unsigned v = threadIdx.x;
        for (; v + 3 * 256 < row_vec; v += 4 * 256) {     // four 16-byte loads in flight per lane before the first store
            v4i x[4];
#pragma unroll
            for (int i = 0; i < 4; i++) x[i] = __builtin_nontemporal_load(src + v + i * 256);
#pragma unroll
            for (int i = 0; i < 4; i++) __builtin_nontemporal_store(x[i], dst + v + i * 256);
        }
        for (; v < row_vec; v += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(src + v), dst + v);
    }
}

hipError_t launch_gather_relayout(const float* d_stage, float* d_full, size_t held, int world, size_t row_floats, int skip_rank,
                                  int n_cus, hipStream_t s)
{
    if (!held || world <= 0 || !row_floats) return hipSuccess;
    if (row_floats % 4 || ((uintptr_t)d_stage & 15) || ((uintptr_t)d_full & 15) || held >= (1u << 31) || row_floats / 4 >= (1u << 31))
        return hipErrorInvalidValue;
    clear_stale_error();
    const size_t n_blocks = held * (size_t)world;
    const size_t cap = (size_t)n_cus * 32;   // enough workgroups per CU to keep the 16-byte requests that fill HBM in flight
    hipLaunchKernelGGL(gather_relayout_kernel, dim3((unsigned)(n_blocks < cap ? n_blocks : cap)), dim3(256), 0, s,
                       reinterpret_cast<const v4i*>(d_stage), reinterpret_cast<v4i*>(d_full), (unsigned)held, (unsigned)world,
                       (unsigned)(row_floats / 4), skip_rank);
    return hipGetLastError();
}

hipError_t launch_dedisperse(const Geometry& g, const float* d_out_unit, float* d_ded, hipStream_t s)
{
    clear_stale_error();
    hipLaunchKernelGGL(dedisperse_kernel, dim3((g.n_beams + 63) / 64), dim3(64), 0, s, d_out_unit, d_ded, g.n_freq,
                       g.n_beams);
    return hipGetLastError();
}

hipError_t launch_dedisperse_units(const Geometry& g, const float* d_out_units, size_t unit_stride, int n_units, float* d_ded,
                                   hipStream_t s)
{
    if (n_units <= 0) return hipSuccess;
    clear_stale_error();
    hipLaunchKernelGGL(dedisperse_units_kernel, dim3((g.n_beams + 63) / 64, n_units), dim3(64), 0, s, d_out_units, unit_stride,
                       d_ded, g.n_freq, g.n_beams, n_units);
    return hipGetLastError();
}

hipError_t launch_dedisperse_dm(const Geometry& g, const float* d_series, int n_t, const int* d_delays, int n_dm,
                                int n_t_out, float* d_out, int* d_flags, hipStream_t s)
{
    if (n_dm <= 0 || n_t_out <= 0) return hipSuccess;
    // g.dm_wide: measurement / test switch (DSABF_DM_WIDE=0 at bf_create, bf_set_switch): the per-thread-window kernel alone
    const bool wide = d_flags && dm_wide_supported(g, n_dm) && g.dm_wide &&
                      !((uintptr_t)d_series & 15) && !((uintptr_t)d_out & 15) &&   // its 16-byte LDS-DMA pieces / 16-byte stores
                      (size_t)n_t * g.n_freq * g.n_beams * sizeof(float) < ((size_t)1 << 32);   // ... and 32-bit byte offsets into the series
    if (wide) {
        hipError_t e = launch_dedisperse_dm_wide(g, d_series, n_t, d_delays, n_dm, n_t_out, d_out, d_flags, s);
        if (e != hipSuccess) return e;
    }
    clear_stale_error();
    if ((size_t)kDmWin * g.n_freq * g.n_beams * 4 >= ((size_t)1 << 32)) return hipErrorInvalidValue;   // SGPR row offsets are 32-bit
    const size_t tiles = (size_t)((n_dm + kDmBlock - 1) / kDmBlock) * (size_t)((n_t_out + kDmTb - 1) / kDmTb) *
                         (size_t)((g.n_beams + kDmThreads - 1) / kDmThreads);
    if (tiles > (size_t)1 << 30) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((tiles + 7) / 8 * 8));    // a multiple of the 8 XCDs (see the kernel's tile order)
    const size_t table = g.n_freq <= kDmMaxTableFreq ? (size_t)g.n_freq * kDmBlock * sizeof(int) : 0;
    hipLaunchKernelGGL(dedisperse_dm_kernel, grid, dim3(kDmThreads), table, s, d_series, d_delays, d_out, n_t, g.n_freq,
                       g.n_beams, n_t_out, n_dm, wide ? (const int*)d_flags : nullptr);
    return hipGetLastError();
}

const char* fused_kernel_name(const Geometry& g, char* buf, size_t n)
{
    if (use_generic(g)) {
        snprintf(buf, n, "dsabf::fusedg_kernel<ANT=%d (%d k-steps, %d-byte staging),NIPO=%d%s> (v_mfma_i32_16x16x64_i8)", g.n_ant,
                 generic_ksteps(g), g.n_ant % 16 ? 4 : 16, g.n_ipo, g.fast_detect ? ",FAST" : g.contracted_detect ? ",CONTRACTED" : "");
        return buf;
    }
    // (the geometry in words; which template instantiation that is -- antenna CLASS, run-time or compile-time -- fused_variant_key says)
    snprintf(buf, n, "dsabf::fused16_kernel<ANT=%d,NIPO=%d%s%s%s%s> (v_mfma_i32_16x16x64_i8)", g.n_ant, g.n_ipo, rtw_class(g) ? "(run-time)" : "", (g.fast_detect && g.n_ipo >= 16) ? ",FAST" : g.contracted_detect ? ",CONTRACTED" : "",
             g.paired ? ",PAIRED" : "",
             fused_col_tiles(g, g.paired) == kColTilesWide16 ? ",SLOTS=8" : fused_wg_waves(g) == kWavesWide16 ? ",WAVES=8" : "");
    return buf;
}

const char* fused_variant_key(const Geometry& g, bool write_c, char* buf, size_t n)
{
    if (!n) return buf;
    buf[0] = 0;
    Geometry gg = g;
    if (write_c) gg.paired = false;   // (launch_gemm_only: the stage-parity path always runs the general kernel)
    if (use_generic(gg) || (write_c && deep_class(gg))) return generic_variant_key(gg, write_c, buf, n);
    const FusedVariant v = select_variant(gg, write_c);
    if (v.launch)
        snprintf(buf, n, "fused16_kernel<%d, %d, %s, %d, %s, %d, %d>", v.ain, v.nipo, v.write_c ? "true" : "false", v.mode,
                 v.paired ? "true" : "false", v.waves, v.ns);
    return buf;
}

int fused_vgprs(const Geometry& g)
{
    if (use_generic(g)) return generic_vgprs(g);
    hipFuncAttributes attr{};
    const void* fn = select_variant(g, false).fn;
    if (!fn || hipFuncGetAttributes(&attr, fn) != hipSuccess) return -1;
    return attr.numRegs;
}

}  // namespace dsabf
