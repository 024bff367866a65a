// bf_dm_wide.hip -- DM-trial dedispersion, round-3 kernel: one workgroup = 32 trials x 16 output times x
// 128 beams, the window of input rows SHARED by all 32 trials through LDS.
//
//   out[dm][t][b] = sum over f (ascending, fp32, from +0) of series[t + delay[dm][f]][f][b]
//
// Why a second kernel.  dedisperse_dm_kernel (bf_kernels.hip: 4 trials x 16 times per THREAD, a private LDS column per
// thread) re-loads every input row once per block of 4 trials through the vector-memory path (64 B/clk/CU) and reads one
// LDS dword per add (ds_read_b32: 128 B/clk/CU): 0.42-0.44 ms for 64 trials x 901 samples x 256 x 256, 0.094 of the HBM
// roofline of its algorithmic bytes, LDS-bound (profiles/r02_dm_pmc_summary.txt).  Every add needs one 4-byte operand, so
// what matters is where it comes from: global 16 adds/clk/CU, LDS b32 32, LDS b64/b128 64, registers 128 (the VALU).
// Here
//   * a row is fetched from global memory ONCE per workgroup and serves all 32 trials of the tile (reuse 32 x 16 /
//     (16 + spread) instead of 4 x 16 / 24): the vector-memory path drops out of the picture;
//   * a wave owns TWO neighbouring trials with all its lanes x 2 beams per lane, and the two SHARE the rows they have in
//     common (round 4): the wave reads rows d_A .. d_A + 18 of the window once (19 ds_read_b64, whatever the trials' distance)
//     and trial B adds r[i + delta], delta = d_B - d_A in 0 ... 3 on any fine ladder -- wave-uniform, so a scalar branch over
//     four straight-line blocks with static register indices (round 3 gave each trial a half-wave and its own 16 reads; one
//     body per delta with delta-dependent READS made the compiler merge the bodies and spill -- here no memory operation
//     depends on delta);
//   * bounds: one fp32 add and one 4-byte LDS operand per (trial, time, beam, channel).  The adds: 64 lanes per clock and CU
//     (v_pk_add_f32 issues at half rate: the same) = 1024 cycles per channel and tile; the operands: round 3 1024 cycles at
//     256 B/clk/CU, now 19/32 of that.
//   * what actually binds the kernel (round 4, late): INSTRUCTION ISSUE.  The 16 waves run their per-channel instruction
//     streams in lock step between barriers, and every instruction a wave executes per channel costs ~0.7 % of the kernel's
//     time (measured by adding some).  LDS, VALU and the DMA path are each 30-35 % busy.  So the loop around the asm body is
//     kept as short as it gets -- 92 instructions per wave and channel, ~60 of them the adds, the row reads and their waits:
//     windows staged by buffer-addressed LDS-DMA from a byte offset the table pass precomputed (no range test, no vector
//     address), the count of issued DMAs kept instead of recomputed, a hand-written vmcnt switch, a wave's two offsets as one
//     16-bit read, the loop unrolled over the ring's three positions.  0.249 ms (round 3) -> 0.192 ms.
// The sum of one (trial, time, beam) still runs over ascending f in ONE register, one add per channel: the same bits as
// dedisperse_dm_kernel and as the oracle (orc_dedisperse_dm).
//
// A trial group "fits" when, at every channel, the delays of its <= 32 trials span no more than the window holds
// (every tile decides for its group, on the device: the delays are device data).  Groups that
// do not fit -- coarse or non-monotonic ladders, negative-going delays -- are left to dedisperse_dm_kernel, which skips the
// groups this kernel took.
#include "bf_kernels.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>

#include "bf_dm_wide_body.inc"   // the per-channel bodies as inline-asm text (tools/gen_dm_body.py)

namespace dsabf {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

#ifndef DSABF_DW_PROBE
#define DSABF_DW_PROBE 0   // the one compile-time switch of this file: a diagnostic build tools/dm_probe.py makes BESIDE the product
#endif                     // (s_memtime around the phases of a channel; results invalid)
constexpr int kDwTb = 16;   // output times per tile; x beams per lane = 64 accumulator registers
constexpr int kDwBpl = 64 / kDwTb;   // beams per lane: 4 (ds_read_b128) or 2 (ds_read_b64), both 256 B/clk/CU
static_assert(kDwTb == 16 || kDwTb == 32, "16 times x 4 beams or 32 times x 2 beams per lane");
constexpr int kDwWaves = kDwWavesPerWg;   // waves per workgroup, two trials each
constexpr int kDwThreads = 64 * kDwWaves;
constexpr int kDwBeams = 32 * kDwBpl; // beams per tile: a half-wave across
constexpr int kDwRowBytes = kDwBeams * 4;
constexpr int kDwRowsPerDma = 1024 / kDwRowBytes;   // window rows one LDS-DMA instruction moves (16 B per lane, 1 KiB): 2 or 4
typedef float vbf __attribute__((ext_vector_type(kDwBpl)));   // a lane's beams
constexpr int kDwMaxRows = 4 * kDwRowsPerDma * kDwWaves < 224 ? 4 * kDwRowsPerDma * kDwWaves : 224;   // rows per window buffer at most (byte offsets; <= 4 DMAs per wave)
constexpr int kDwSlackRows = 4;      // rows a wave may read past its window (never used: see the adds)
constexpr int kDwNbuf = 3;           // window buffers in LDS: channel f is consumed while f + 1 and f + 2 are landing
constexpr int kDwPairsPerWave = (kDwMaxRows / kDwRowsPerDma + kDwWaves - 1) / kDwWaves;   // LDS-DMA instructions a wave issues per window at most
constexpr int kDwLdsBytes = 160 * 1024 / (16 / kDwWaves) - 1024;   // the CU's LDS over the workgroups its 16 wave slots hold,
                                                                   // less the static words of __syncthreads_and (256 B)

// LDS carve-up for n_freq channels: [offs: kDwTrials x n_freq bytes][{base, rows}: n_freq x 2 int][kDwNbuf window buffers]
__host__ __device__ inline int dw_table_bytes(int n_freq) { return (n_freq * (kDwTrials + 8) + 511) & ~511; }
__host__ __device__ inline int dw_rows_cap(int n_freq)
{
    // (4 rows of slack behind the last buffer: a wave reads up to 20 rows from its first trial's offset, see the adds)
    const int r = ((kDwLdsBytes - dw_table_bytes(n_freq) - kDwSlackRows * kDwRowBytes) / (kDwNbuf * kDwRowBytes)) & ~(kDwRowsPerDma - 1);   // whole DMAs
    return r > kDwMaxRows ? kDwMaxRows : r;
}

// One lane's beams of one window row, as an explicit ds_read_b64 / ds_read_b128.  Left to itself the compiler fuses two
// 8-byte reads of neighbouring rows into ds_read2_b64, which runs at HALF the LDS rate of ds_read_b64 (128 instead of
// 256 B/clk/CU, MI355X_MICROARCH.md section LDS) -- and the LDS read rate is what this kernel's adds live on.  The compiler
// does not see these reads in lgkmcnt: dw_wait<N>() below is the matching s_waitcnt, tied to the registers it releases.
__device__ __forceinline__ void lds_row(vbf& v, const char* p)
{
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p;
    if constexpr (kDwBpl == 2)
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory");
    else
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
}
// wait until at most N LDS operations issued AFTER the batch `r` are outstanding (LDS returns in order): r is then valid
template <int N, int B>
__device__ __forceinline__ void dw_wait(vbf (&r)[B])
{
    static_assert(B == 4 || B == 8, "batch");
    if constexpr (B == 8)
        asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "n"(N) : "memory");
}

__global__ __launch_bounds__(kDwThreads) void dedisperse_dm_wide_kernel(const float* __restrict__ series,
                                                                        const int* __restrict__ delays,
                                                                        int* __restrict__ flags,
                                                                        const float* __restrict__ zero_row, float* __restrict__ out,
                                                                        int n_t, int n_freq, int n_beams, int n_t_out, int n_dm,
                                                                        int rows_cap)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    unsigned char* offs = reinterpret_cast<unsigned char*>(lds);                         // [trial pair][f][A | B]: delay - base[f] (< 256): a wave's two offsets are ONE 16-bit read
    v2i* tab = reinterpret_cast<v2i*>(lds + kDwTrials * n_freq);                          // [f] {smallest delay of the group, rows of the window}
    char* win = lds + dw_table_bytes(n_freq);                                             // kDwNbuf x rows_cap x 512 B

    // tile order: beam half fastest, then trial group, then time block; consecutive tiles on one XCD (they share rows)
    const int n_g = (n_dm + kDwTrials - 1) / kDwTrials, n_y = (n_t_out + kDwTb - 1) / kDwTb;
    const int n_bg = (n_beams + kDwBeams - 1) / kDwBeams;
    const int per_xcd = gridDim.x / 8;
    const int v = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if (v >= n_g * n_y * n_bg) return;
    const int bg = v % n_bg, g = (v / n_bg) % n_g, ty = v / (n_bg * n_g);
    const int dm0 = g * kDwTrials, nk = min(kDwTrials, n_dm - dm0), t0 = ty * kDwTb;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- tables: per channel the group's smallest delay and window height, per trial its offset into the window ----------
    int fits = 1;
    for (int f = tid; f < n_freq; f += kDwThreads) {
        int lo = 0x7fffffff, hi = -0x7fffffff - 1;
        for (int k = 0; k < nk; k++) {
            const int d = delays[(size_t)(dm0 + k) * n_freq + f];
            lo = min(lo, d);
            hi = max(hi, d);
        }
        // .y: rows of the window; bit 16: every row of every DMA piece of the window lies inside the series (the common case:
        // the staging loop then needs no range test and no per-lane address)
        const int rows = hi - lo + kDwTb, rows_up = (rows + kDwRowsPerDma - 1) / kDwRowsPerDma * kDwRowsPerDma;
        const bool inside = (long long)t0 + lo >= 0 && (long long)t0 + lo + rows_up <= n_t && (bg + 1) * kDwBeams <= n_beams;   // (and every beam of the tile live)
        // .x: the group's smallest delay -- or, for an inside window, the byte offset of its first row at this channel and tile
        // (what the staging loop's scalar offset starts from; 32 bits: the launcher keeps longer series away from this kernel)
        const unsigned first_b = (unsigned)(t0 + lo) * (unsigned)((size_t)n_freq * n_beams * sizeof(float)) + (unsigned)f * (unsigned)n_beams * (unsigned)sizeof(float);
        tab[f] = v2i{inside ? (int)first_b : lo, rows | (inside ? 0x10000 : 0)};
        fits &= (long long)hi - lo + kDwTb <= rows_cap && lo > -(1 << 30) && hi < (1 << 30);   // (and row arithmetic stays in int)
        for (int k = 0; k < kDwTrials; k++)               // a missing trial repeats the last one (computed, never stored)
            offs[((k >> 1) * n_freq + f) * 2 + (k & 1)] = (unsigned char)(delays[(size_t)(dm0 + min(k, nk - 1)) * n_freq + f] - lo);
    }
    // Does the group fit -- at every channel, do the delays of its trials span no more than a window holds?  Every tile of
    // the group reaches the same verdict from the same delays; one of them records it for dedisperse_dm_kernel, which runs
    // behind this kernel on the same stream and takes the groups that do not fit.
    fits = __syncthreads_and(fits);
    if (ty == 0 && bg == 0 && tid == 0) flags[g] = fits;
    if (!fits) return;

    // ---- staging by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, 1 KiB = TWO window rows per wave instruction, no
    // register on the way): wave w moves row pairs w, w + 16, ... of a window.  Rows outside the series -- and the beams
    // behind the last one -- are fetched from a 512-byte row of zeros (they contribute +0).
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // staging lane: 16 B (4 beams) of row (lane / lanes-per-row) of the DMA's rows; n_beams % 4 == 0: all four live or none
    constexpr int kLpr = kDwRowBytes / 16;                // lanes per row
    const int bs = bg * kDwBeams + 4 * (lane % kLpr);
    const size_t row_stride = (size_t)n_freq * n_beams;
    const int win_bytes = rows_cap * kDwRowBytes;
    // Addressing is split so that the per-lane part is loop-invariant: lane_src already holds the lane's row inside a DMA
    // (lane / lanes-per-row) and its beams; what changes per DMA -- first row of the piece, channel -- is wave-uniform and
    // stays on the scalar unit (a per-lane row * stride would be two quarter-rate 64-bit multiplies per instruction).  The
    // common case -- every row of the piece inside the series, every beam live -- needs no per-lane test either.
    // (byte offsets in 32 bits: the launcher sends series of 4 GiB and more to dedisperse_dm_kernel)
    const unsigned lane_off = (unsigned)(((size_t)(lane / kLpr) * row_stride + bs) * sizeof(float));
    const float* lane_zero = zero_row + 4 * (lane % kLpr);
    const bool beams_ok = bs < n_beams;
    const bool all_beams_ok = __builtin_amdgcn_readfirstlane((int)__all(beams_ok ? 1 : 0)) != 0;   // wave-uniform
    const unsigned row_stride_b = (unsigned)(row_stride * sizeof(float));   // uniform
    // the common case on the scalar unit alone: a buffer descriptor over the series, the lane's constant offset as the only vector
    // operand, (window row, channel, piece) in the scalar offset -- 5 scalar instructions and the DMA per piece, no range test
    const __amdgpu_buffer_rsrc_t s_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(series), 0, -1, 0x00020000);
    const unsigned wave_piece_b = (unsigned)(kDwRowsPerDma * wave) * row_stride_b;          // this wave's first piece, bytes from the window's first row
    // returns the number of DMA instructions THIS wave issued (what wait_dma_but has to leave in flight one iteration later)
    auto dma_window = [&](int f, int slot_b, v2i t) -> int { // t = tab[f], already in registers; slot_b = (f % kDwNbuf) * win_bytes
        int issued = 0;
        const int ty_ = __builtin_amdgcn_readfirstlane(t.y), tx_ = __builtin_amdgcn_readfirstlane(t.x);
        const int nr = ty_ & 0xFFFF;
        char* buf = win + slot_b;
        if (ty_ & 0x10000) {
            const unsigned soff0 = (unsigned)tx_ + wave_piece_b;
            // (nested, not four independent tests: a wave without a second piece -- the common case -- leaves after one more compare)
            auto piece = [&](int j) {
                const int pr = wave + kDwWaves * j;       // wave-uniform
                __builtin_amdgcn_raw_ptr_buffer_load_lds(s_rsrc, (lds_ptr)(buf + kDwRowsPerDma * pr * kDwRowBytes), 16, (int)lane_off,
                                                         (int)(soff0 + (unsigned)(kDwRowsPerDma * kDwWaves * j) * row_stride_b), 0, 0);
                issued = j + 1;
            };
            auto more = [&](int j) { return kDwRowsPerDma * (wave + kDwWaves * j) < nr; };
            static_assert(kDwPairsPerWave == 4, "the nest below is written for four pieces per wave");
            if (more(0)) {
                piece(0);
                if (more(1)) {
                    piece(1);
                    if (more(2)) {
                        piece(2);
                        if (more(3)) piece(3);
                    }
                }
            }
        } else {
        const int first = t0 + tx_;
        const unsigned col_b = (unsigned)f * (unsigned)n_beams * (unsigned)sizeof(float);   // uniform
#pragma unroll
        for (int j = 0; j < kDwPairsPerWave; j++) {
            const int pr = wave + kDwWaves * j;           // wave-uniform
            if (kDwRowsPerDma * pr < nr) {
                const int row0 = first + kDwRowsPerDma * pr;                       // uniform: first row of this piece
                const char* src;
                if (all_beams_ok && row0 >= 0 && row0 + kDwRowsPerDma <= n_t) {    // the common case: scalar base + the lane's constant offset
                    const char* sbase = reinterpret_cast<const char*>(series) + ((unsigned)row0 * row_stride_b + col_b);
                    src = sbase + lane_off;
                } else {                                                            // an end of the series / of the beams
                    const int row = row0 + lane / kLpr;
                    src = (row >= 0 && row < n_t && beams_ok)
                              ? reinterpret_cast<const char*>(series + ((size_t)row * row_stride + (size_t)f * n_beams + bs))
                              : reinterpret_cast<const char*>(lane_zero);
                }
                __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(src), (lds_ptr)(buf + kDwRowsPerDma * pr * kDwRowBytes), 16, 0, 0);
                issued = j + 1;
            }
        }
        }
        return issued;
    };
    // The compiler must neither drain the DMA queue at a barrier (a fence would: vmcnt(0)) nor move LDS reads across one:
    // raw s_barrier between compiler-level memory barriers, the waits written out.
    // (one asm statement: as a C++ switch the five cases cost ~15 scalar instructions of flag shuffling per channel, and every
    //  instruction a wave executes per channel is ~0.7 % of the kernel's time)
    auto wait_dma_but = [&](int newest) {                 // until at most `newest` of this wave's DMA instructions are in flight
        asm volatile(
            "s_cmp_lt_u32 %0, 2\n s_cbranch_scc1 .Ldww01_%=\n"
            "s_cmp_lt_u32 %0, 3\n s_cbranch_scc1 .Ldww2_%=\n"
            "s_cmp_lt_u32 %0, 4\n s_cbranch_scc1 .Ldww3_%=\n"
            "s_waitcnt vmcnt(4)\n s_branch .Ldwwe_%=\n"
            ".Ldww3_%=:\n s_waitcnt vmcnt(3)\n s_branch .Ldwwe_%=\n"
            ".Ldww2_%=:\n s_waitcnt vmcnt(2)\n s_branch .Ldwwe_%=\n"
            ".Ldww01_%=:\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 .Ldww0_%=\n s_waitcnt vmcnt(1)\n s_branch .Ldwwe_%=\n"
            ".Ldww0_%=:\n s_waitcnt vmcnt(0)\n"
            ".Ldwwe_%=:\n"
            :
            : "s"(newest)
            : "memory", "scc");
    };
    static_assert(kDwPairsPerWave <= 4, "wait_dma_but covers up to 4 instructions per window");
    auto block_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the buffer about to be refilled are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- the adds: wave w owns trials 2 w and 2 w + 1 with ALL its lanes (2 beams per lane), 16 output times --------------------
    // Round 4.  The two trials of a wave need rows d_A .. d_A + 15 and d_B .. d_B + 15 of the channel's window, and on any fine
    // DM ladder d_B - d_A = delta is 0 ... 3: the wave reads rows d_A .. d_A + 15 (+ 4 more if delta > 0) ONCE and trial B adds
    // row i + delta out of the same registers -- delta is wave-uniform, so that is a scalar branch over four straight-line
    // bodies with static register indices in each.  LDS reads per add fall from 1 to 0.5 - 0.63 (round 3: one half-wave per trial, 16 ds_read_b128 per channel and wave: the LDS read rate and the VALU
    // rate were both AT their floor of 1024 cycles per channel and tile and did not overlap: 0.25 ms).  A pair further apart
    // (coarse ladders that still fit a window) re-reads 16 rows for trial B.
    constexpr int kAddBpl = 2;                            // beams per lane in the adds
    typedef float v2a __attribute__((ext_vector_type(2)));
    const int kA = 2 * wave, kB = 2 * wave + 1;
    const unsigned short* offsAB = reinterpret_cast<const unsigned short*>(offs) + wave * n_freq;   // [f]: A's offset | B's << 8
    const int bq = bg * kDwBeams + kAddBpl * lane;        // this lane's beams (n_beams % 4 == 0 and bq even: both live or none)
    const int lane_col = lane * 4 * kAddBpl;
    // The accumulators -- [trial A | B][time i]{beam, beam + 1} = elements 2 i, 2 i + 1 -- and the two row batches are pinned to
    // the registers the asm bodies name (bf_dm_wide_body.inc): physical-register operands of ONE asm statement per channel.
    typedef float v8a __attribute__((ext_vector_type(8)));
    typedef float v32a __attribute__((ext_vector_type(32)));
    v32a accA, accB;
#pragma unroll
    for (int i = 0; i < 32; i++) accA[i] = accB[i] = 0.0f;
    static_assert(kDwTb == 16, "the shared-row bodies (tools/gen_dm_body.py) are written for 16 output times");
    static_assert(DW_ASM_BASE == 40, "the operand constraints below name the registers of tools/gen_dm_body.py");

    dma_window(0, 0, tab[0]);
    if (n_freq > 1) dma_window(1, (1 % kDwNbuf) * win_bytes, tab[1]);
    int oAB = offsAB[0];
    v2i tnext = tab[min(2, n_freq - 1)];                  // the window the first iteration will fetch
    wait_dma_but(0);
    block_barrier();
#if DSABF_DW_PROBE
    unsigned long long pt_body = 0, pt_dma = 0, pt_bar = 0, pt0 = __builtin_amdgcn_s_memtime();
#endif
    // one channel: f's window sits at byte offset `slot` of the ring, window f + 2 goes to `slot2`
    auto channel = [&](const int f, const int slot, const int slot2) {
        // bookkeeping reads first (the compiler's own LDS operations: everything after them is one of the bodies' row reads;
        // LDS returns in order, so the bodies' counted waits also cover whatever of these is still in flight)
        // (behind the last channel these read the next table's first bytes: inside the LDS allocation, never used)
        const int oAB_next = offsAB[f + 1];
        const v2i tafter = tab[f + 3];
        const int uAB = __builtin_amdgcn_readfirstlane(oAB);
        const int uA = uAB & 0xFF, uB = uAB >> 8;
        const int delta = uB - uA;                        // wave-uniform
        const unsigned pa = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(win + slot + uA * kDwRowBytes + lane_col);
        const unsigned pb = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)(win + slot + uB * kDwRowBytes + lane_col);
        // rows 0 .. 7 of trial A's window are requested at once ...
        v8a ra, rb;
        asm volatile(DW_BODY_HEAD : "={v[104:111]}"(ra), "={v[112:119]}"(rb) : [pa] "v"(pa) : "memory");
        // ... and the DMA of window f + 2 is issued in their shadow (ring slot of f + 2 was last read in iteration f - 1: every
        // wave is past that iteration's barrier)
        const int newest = f + 2 < n_freq ? dma_window(f + 2, slot2, tnext) : 0;
        asm volatile(DW_BODY_ALL
                     : "+{v[40:71]}"(accA), "+{v[72:103]}"(accB), "+{v[104:111]}"(ra), "+{v[112:119]}"(rb)
                     : [pa] "v"(pa), [pb] "v"(pb), [d] "s"(delta)
                     : "memory", "scc");
        oAB = oAB_next;
        tnext = tafter;
#if DSABF_DW_PROBE
        const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
        wait_dma_but(newest);                             // this wave's pieces of window f + 1 have landed ...
#if DSABF_DW_PROBE
        const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
        block_barrier();                                  // ... and so have everybody else's
#if DSABF_DW_PROBE
        const unsigned long long pt3 = __builtin_amdgcn_s_memtime();
        pt_body += pt1 - pt0, pt_dma += pt2 - pt1, pt_bar += pt3 - pt2, pt0 = pt3;
#endif
    };
    // A whole turn of the ring per trip: the ring positions are loop constants (no per-channel position arithmetic: every
    // instruction a wave executes per channel is ~0.7 % of the kernel's time), three copies of the body.
    static_assert(kDwNbuf == 3, "the loop below is unrolled over a ring of three windows");
    const int pos1 = win_bytes, pos2 = 2 * win_bytes;
    int f0 = 0;
    for (; f0 + kDwNbuf <= n_freq; f0 += kDwNbuf) {
        channel(f0, 0, pos2);
        channel(f0 + 1, pos1, 0);
        channel(f0 + 2, pos2, pos1);
    }
    if (f0 < n_freq) {
        channel(f0, 0, pos2);
        if (f0 + 1 < n_freq) channel(f0 + 1, pos1, 0);
    }
#if DSABF_DW_PROBE   // diagnostic build only (results invalid): tile 0's waves leave their per-channel cycle averages in out[wave * 4 ...]
    if (v == 0 && lane == 0) {
        out[wave * 4 + 0] = (float)pt_body / n_freq, out[wave * 4 + 1] = (float)pt_dma / n_freq, out[wave * 4 + 2] = (float)pt_bar / n_freq;
        out[wave * 4 + 3] = (float)(pt_body + pt_dma + pt_bar) / n_freq;
    }
    return;
#endif

    if (bq >= n_beams) return;
#pragma unroll
    for (int tr = 0; tr < 2; tr++) {
        const int k_mine = tr ? kB : kA;
        if (k_mine >= nk) continue;
        float* o = out + ((size_t)(dm0 + k_mine) * n_t_out + t0) * n_beams + bq;
#pragma unroll
        for (int i = 0; i < kDwTb; i++)
            if (t0 + i < n_t_out) *reinterpret_cast<v2a*>(o + (size_t)i * n_beams) = tr ? v2a{accB[2 * i], accB[2 * i + 1]} : v2a{accA[2 * i], accA[2 * i + 1]};
    }
}

}  // namespace

bool dm_wide_supported(const Geometry& g, int n_dm)
{
    return g.n_freq <= kDwMaxFreq && n_dm > 0 && (n_dm + kDwTrials - 1) / kDwTrials <= kDwMaxGroups;
}

// Runs the wide kernel on the trial groups that fit a window and records which did (d_flags[g], kDwMaxGroups ints owned by
// the handle) for dedisperse_dm_kernel.
hipError_t launch_dedisperse_dm_wide(const Geometry& g, const float* d_series, int n_t, const int* d_delays, int n_dm,
                                     int n_t_out, float* d_out, int* d_flags, hipStream_t s)
{
    const int n_g = (n_dm + kDwTrials - 1) / kDwTrials;
    const int rows_cap = dw_rows_cap(g.n_freq);
    (void)hipGetLastError();
    const size_t tiles = (size_t)n_g * (size_t)((n_t_out + kDwTb - 1) / kDwTb) * (size_t)((g.n_beams + kDwBeams - 1) / kDwBeams);
    if (tiles > (size_t)1 << 30) return hipErrorInvalidValue;
    const int lds = dw_table_bytes(g.n_freq) + kDwNbuf * rows_cap * kDwRowBytes + kDwSlackRows * kDwRowBytes;
    const float* zero_row = reinterpret_cast<const float*>(d_flags + kDwMaxGroups);   // 512 B of zeros behind the flags
    static std::atomic<unsigned> done_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (dev >= 32 || !(done_mask.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dedisperse_dm_wide_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, kDwLdsBytes);
        if (e != hipSuccess) return e;
        done_mask.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(dedisperse_dm_wide_kernel, dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(kDwThreads), lds, s, d_series,
                       d_delays, d_flags, zero_row, d_out, n_t, g.n_freq, g.n_beams, n_t_out, n_dm, rows_cap);
    return hipGetLastError();
}

}  // namespace dsabf
