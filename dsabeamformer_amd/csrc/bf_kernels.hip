// bf_kernels.hip -- hand-written gfx950 (CDNA4, MI355X) kernels for the DSA beamformer hot path.
//
// Replaces the reference's three device stages with ONE kernel:
//   expand_input (src/beamformer.cuh:66-109)  -> nibble expand in registers while staging into LDS
//   cublasGemmStridedBatchedEx (src/beamformer.cu:470-477) -> v_mfma_i32_32x32x32_i8 on a real-embedded K
//   detect_sum (src/beamformer.cuh:115-155)   -> power detect + time/pol accumulate in the accumulator VGPRs
// so the reference's d_B (expanded voltages) and d_C (complex fp32 beams, 16 MiB per beam-block in production)
// never touch HBM.  This is not a translation of those kernels; the design notes are in DESIGN.md section 3.
//
// Work decomposition
//   workgroup = 512 threads = 8 wave64, owns (frequency f, group of 256 beams, contiguous range of time chunks)
//   wave w    = one 32-beam tile: its weight fragments (re-row and im-row images, K' = 32*NKS int8) live in
//               VGPRs for the whole kernel; it streams every time tile of the workgroup's range through MFMA.
//   time      = the MFMA row (M) axis, beams = the column (N) axis: each lane owns ONE beam and holds 16 time
//               samples of it in its accumulator registers, so the detect/accumulate is a sequential in-register
//               fp32 add chain in exactly the reference's order (bit-exact for every n_ipo, not only n_ipo = 2).
//
// Exact-arithmetic tricks (all proven in tests/test_numerics_tricks.py on the CPU):
//   * a packed byte b = (re << 4 | im & 15) is expanded to the int8 pair (b & 0xF0, (b << 4) & 0xF0) = (16*re,
//     16*im): two's complement places the signed nibble in the top of the byte, no sign-extension ops needed.
//     The MFMA therefore accumulates 16 * n (|16 n| <= 2,080,768).
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

namespace dsabf {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

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
    int chunks_total;                // ceil(tiles / 4)
    int n_tsplit;                    // workgroups along time
};

// MFMA D-row r (0..31) <-> (lane half, accumulator register): rows held by half h, reg j are
// (j&3) + 8*(j>>2) + 4*h.  We want half h / reg j to mean "sample j of run h", so LDS row of D-row r is:
__device__ __forceinline__ int lds_row_of_mfma_row(int r) { return ((r >> 2) & 1) * 16 + (r >> 3) * 4 + (r & 3); }

// XOR swizzle of the 16-byte chunk index inside an LDS row so that both the ds_write_b128 of the staging pass
// and the ds_read_b128 of the fragment pass are bank-conflict free.
template <int RBC>
__device__ __forceinline__ int swz(int chunk, int row)
{
    return RBC == 8 ? (chunk ^ (((row >> 1) & 7) ^ ((row & 1) << 2))) : (chunk ^ (row & 15));
}

template <int NKS, int NIPO, bool WRITE_C>
__global__ __launch_bounds__(kWgThreads, (NKS <= 4 ? 4 : 2)) void fused_kernel(FusedArgs a)
{
    constexpr int RBC = (NKS <= 4) ? 8 : 16;            // 16-byte chunks per LDS row
    constexpr int RB = RBC * 16;                         // LDS row bytes: [16*re of ant 0.. | 16*im of ant 0..]
    constexpr int L = NIPO < 16 ? 16 : NIPO;             // samples per lane-half stream
    constexpr int R = L / 16;                            // row-tiles per output group
    constexpr int A = 16 * NKS;                          // packed bytes per time sample (= n_ant)
    constexpr int PIECES = kRowsPerChunk * NKS;          // 16-byte packed pieces per chunk
    constexpr int PPT = (PIECES + kWgThreads - 1) / kWgThreads;
    static_assert(kTilesPerChunk % R == 0, "chunk must hold whole output groups");

    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x 128 rows x RB

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hl = lane >> 5;   // lane half
    const int lc = lane & 31;   // MFMA column (beam within tile) / MFMA row for the A operand

    // blockIdx -> (f, beam group, time split); f fastest so that workgroups b and b+8 (same XCD under the
    // round-robin dispatch) share weight panels of the same frequencies in that XCD's L2.
    int bid = blockIdx.x;
    const int f = bid % a.n_freq;
    bid /= a.n_freq;
    const int bg = bid % a.n_bgroups;
    const int ts = bid / a.n_bgroups;
    const int c_begin = (int)(((long long)a.chunks_total * ts) / a.n_tsplit);
    const int c_end = (int)(((long long)a.chunks_total * (ts + 1)) / a.n_tsplit);

    const int bt = bg * kWavesPerWg + wave;  // this wave's 32-beam tile
    const bool wave_active = bt < a.n_btiles;
    const int beam = bt * 32 + lc;

    // ---- weight fragments -> registers (once) ------------------------------------------------------------
    v4i bre[NKS], bim[NKS];
    if (wave_active) {
        const v4i* wp = a.wimg + ((size_t)(f * a.n_btiles + bt) * 2 * NKS) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            bre[ks] = wp[(size_t)ks * 64];
            bim[ks] = wp[(size_t)(NKS + ks) * 64];
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < NKS; ks++) {
            bre[ks] = v4i{0, 0, 0, 0};
            bim[ks] = v4i{0, 0, 0, 0};
        }
    }

    v16i kc;
#pragma unroll
    for (int i = 0; i < 16; i++) kc[i] = (int)kMagicBits;
    asm volatile("" : "+v"(kc));  // keep the seed in registers; do not rematerialise 16 v_mov per tile

    // ---- staging helpers ---------------------------------------------------------------------------------
    v4i stage[PPT];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int pc = tid + k * kWgThreads;
            const int rr = pc / (16 * NKS);   // run within chunk (0..7): tile rr>>1, half rr&1
            const int pi = pc % (16 * NKS);   // 16-byte piece within the run
            const unsigned tile = (unsigned)c * kTilesPerChunk + (rr >> 1);
            const unsigned hs = 2u * (tile / R) + (rr & 1);
            const unsigned s0 = hs * L + 16u * (tile % R);
            stage[k] = v4i{0, 0, 0, 0};
            if (pc < PIECES && s0 < a.S) {
                const unsigned u = a.t_shift >= 0 ? (s0 >> a.t_shift) : (s0 / (unsigned)a.T);
                const unsigned t = s0 - u * (unsigned)a.T;
                const uint8_t* src = a.in + ((size_t)((size_t)u * a.n_freq + f) * a.T + t) * A + (size_t)pi * 16;
                stage[k] = *reinterpret_cast<const v4i*>(src);
            }
        }
    };
    auto write_chunk = [&](char* buf) {
#pragma unroll
        for (int k = 0; k < PPT; k++) {
            const int pc = tid + k * kWgThreads;
            if (pc < PIECES) {
                const int rr = pc / (16 * NKS);
                const int pi = pc % (16 * NKS);
                const int row = rr * 16 + pi / NKS;
                const int ks = pi % NKS;
                v4i re, im;
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    const unsigned w = (unsigned)stage[k][d];
                    re[d] = (int)(w & 0xF0F0F0F0u);          // 16 * real nibble, as int8 x4
                    im[d] = (int)((w << 4) & 0xF0F0F0F0u);   // 16 * imag nibble
                }
                *reinterpret_cast<v4i*>(buf + row * RB + 16 * swz<RBC>(ks, row)) = re;
                *reinterpret_cast<v4i*>(buf + row * RB + 16 * swz<RBC>(RBC / 2 + ks, row)) = im;
            }
        }
    };

    // ---- per-lane constants for the fragment reads and the epilogue --------------------------------------
    const int arow = lds_row_of_mfma_row(lc);
    int aoff[NKS];
#pragma unroll
    for (int ks = 0; ks < NKS; ks++) aoff[ks] = arow * RB + 16 * swz<RBC>(hl * (RBC / 2) + ks, arow);

    const size_t FB = (size_t)a.n_freq * a.n_beams;
    float carry = 0.0f;  // running sum of the current output when NIPO > 16

    if (c_begin >= c_end) return;

    // ---- prologue ------------------------------------------------------------------------------------------
    load_chunk(c_begin);
    write_chunk(smem);
    if (c_begin + 1 < c_end) load_chunk(c_begin + 1);
    __syncthreads();

    for (int c = c_begin; c < c_end; c++) {
        char* cur = smem + ((c - c_begin) & 1) * (kRowsPerChunk * RB);
        char* nxt = smem + ((c - c_begin + 1) & 1) * (kRowsPerChunk * RB);

        // stage chunk c+1 (its global loads were issued one iteration ago) and issue the loads of chunk c+2
        if (c + 1 < c_end) write_chunk(nxt);
        if (c + 2 < c_end) load_chunk(c + 2);

        if (wave_active) {
#pragma unroll
            for (int j = 0; j < kTilesPerChunk; j++) {
                // -- A fragments: 32 time rows x (16*NKS re | 16*NKS im) int8 --
                v4i af[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ks++)
                    af[ks] = *reinterpret_cast<const v4i*>(cur + j * 32 * RB + aoff[ks]);

                v16i are = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bre[0], kc, 0, 0, 0);
                v16i aim = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[0], bim[0], kc, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < NKS; ks++) {
                    are = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks], bre[ks], are, 0, 0, 0);
                    aim = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[ks], bim[ks], aim, 0, 0, 0);
                }

                // -- epilogue: lane (beam, half hl) holds samples s0 .. s0+15 of its half-stream --
                const unsigned tile = (unsigned)c * kTilesPerChunk + j;
                const unsigned hs = 2u * (tile / R) + hl;
                const unsigned s0 = hs * L + 16u * (tile % R);
                const bool valid = (s0 < a.S) && (beam < a.n_beams);

                // NOTE: bit-cast the WHOLE vector; __builtin_bit_cast(float, vec[i]) is miscompiled by ROCm 7.2
                // clang (it reads element 0 for every i).
                const v16f fre = __builtin_bit_cast(v16f, are);
                const v16f fim = __builtin_bit_cast(v16f, aim);
                float p[16];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const float xr = __builtin_fmaf(fre[i], kAlpha16, kNegMagicAlpha16);
                    const float xi = __builtin_fmaf(fim[i], kAlpha16, kNegMagicAlpha16);
                    if constexpr (WRITE_C) {
                        if (valid) {
                            v2f cv = {xr, xi};
                            *reinterpret_cast<v2f*>(a.out + 2 * (((size_t)f * a.T + (s0 + i)) * a.n_beams + beam)) = cv;
                        }
                    } else {
                        const float xx = xr * xr;
                        const float yy = xi * xi;
                        p[i] = xx + yy;
                    }
                }
                if constexpr (!WRITE_C) {
                    if constexpr (NIPO < 16) {
                        constexpr int OPR = 16 / NIPO;  // outputs per run
                        float* op = a.out + ((size_t)hs * OPR) * FB + (size_t)f * a.n_beams + beam;
#pragma unroll
                        for (int g = 0; g < OPR; g++) {
                            float s = p[g * NIPO];
#pragma unroll
                            for (int i = 1; i < NIPO; i++) s = s + p[g * NIPO + i];
                            asm volatile("" : "+v"(s));  // keep the detect outside the store predicate (see below)
                            if (valid) op[(size_t)g * FB] = s;
                        }
                    } else {
                        float s = (j % R == 0) ? p[0] : (carry + p[0]);
#pragma unroll
                        for (int i = 1; i < 16; i++) s = s + p[i];
                        // Pin the value here: otherwise the compiler sinks the whole detect under `if (valid)`,
                        // hoists the MFMAs of the next tile above that branch and doubles the live accumulators.
                        asm volatile("" : "+v"(s));
                        carry = s;
                        if ((j % R == R - 1) && valid) a.out[(size_t)hs * FB + (size_t)f * a.n_beams + beam] = s;
                    }
                }
                // One tile at a time per wave: 4 waves/SIMD overlap each other's MFMA and VALU phases; letting the
                // scheduler interleave two tiles doubles the live accumulators and spills at the 128-VGPR budget.
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Weight re-layout: reference [f][a][b]{re,im} -> MFMA B-operand fragment image
//   image[f][bt][rho][ks][lane] (16 bytes): lane = 32*h + c; byte i multiplies component h (0 = re, 1 = im) of
//   antenna 16*ks + i in the A operand, for output row rho of beam 32*bt + c:
//     rho = 0 (Re C): h=0 -> Wr, h=1 -> -Wi         rho = 1 (Im C): h=0 -> Wi, h=1 -> Wr
__global__ void weight_relayout_kernel(const int8_t* __restrict__ w, v4i* __restrict__ image, int n_freq, int n_ant,
                                       int n_beams, int n_btiles, int nks, int* __restrict__ bad)
{
    const size_t total = (size_t)n_freq * n_btiles * 2 * nks * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63);
        size_t r = idx >> 6;
        const int ks = (int)(r % nks);
        r /= nks;
        const int rho = (int)(r & 1);
        r >>= 1;
        const int bt = (int)(r % n_btiles);
        const int f = (int)(r / n_btiles);
        const int h = lane >> 5, b = bt * 32 + (lane & 31);
        unsigned d[4] = {0, 0, 0, 0};
        for (int i = 0; i < 16; i++) {
            const int ant = ks * 16 + i;
            int v = 0;
            if (ant < n_ant && b < n_beams) {
                const int8_t* e = w + 2 * (((size_t)f * n_ant + ant) * n_beams + b);
                const int wr = e[0], wi = e[1];
                if (wi == -128) *bad = 1;
                v = (rho == 0) ? (h == 0 ? wr : -wi) : (h == 0 ? wi : wr);
            }
            d[i >> 2] |= ((unsigned)v & 0xFFu) << (8 * (i & 3));
        }
        image[idx] = v4i{(int)d[0], (int)d[1], (int)d[2], (int)d[3]};
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
    for (int f = 0; f < n_freq; f++) acc = acc + out_unit[(size_t)f * n_beams + b] * 1.0f;
    ded[b] = acc;
}

template <int NKS, int NIPO, bool WRITE_C>
hipError_t launch_fused_t(const FusedArgs& args, const LaunchShape& ls, hipStream_t s)
{
    auto kern = fused_kernel<NKS, NIPO, WRITE_C>;
    if (ls.lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           ls.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(ls.grid), dim3(ls.block), ls.lds_bytes, s, args);
    return hipGetLastError();
}

int ilog2_exact(int v)
{
    if (v <= 0 || (v & (v - 1))) return -1;
    int s = 0;
    while ((1 << s) < v) s++;
    return s;
}

// (NKS, NIPO) instantiation table.
#define DSABF_FOR_EACH_VARIANT(X) \
    X(1, 2) X(1, 32) X(2, 2) X(2, 32) X(4, 2) X(4, 4) X(4, 8) X(4, 16) X(4, 32) X(4, 64) X(8, 2) X(8, 32)

template <bool WRITE_C>
hipError_t dispatch_fused(const Geometry& g, const FusedArgs& args, const LaunchShape& ls, hipStream_t s)
{
#define X(nks_, nipo_) \
    if (g.nks == nks_ && g.n_ipo == nipo_) return launch_fused_t<nks_, nipo_, WRITE_C>(args, ls, s);
    DSABF_FOR_EACH_VARIANT(X)
#undef X
    return hipErrorInvalidValue;
}

}  // namespace

size_t weight_image_bytes(const Geometry& g) { return (size_t)g.n_freq * g.n_btiles * 2 * g.nks * 64 * 16; }

bool fused_supported(const Geometry& g, const char** why)
{
    const char* dummy;
    if (!why) why = &dummy;
    if (g.n_beams <= 0 || g.n_beams % 32) { *why = "n_beams must be a positive multiple of 32"; return false; }
    if (g.n_ant != 16 * g.nks) { *why = "n_ant must be a multiple of 16 (16, 32, 64 or 128) in this build"; return false; }
    if (g.n_time % 16) { *why = "n_out_per_gemm * n_pol * n_avg must be a multiple of 16"; return false; }
    if (g.n_ipo > 16 && g.n_time % g.n_ipo) { *why = "n_time must be a multiple of n_ipo"; return false; }
#define X(nks_, nipo_) \
    if (g.nks == nks_ && g.n_ipo == nipo_) return true;
    DSABF_FOR_EACH_VARIANT(X)
#undef X
    *why = "no kernel instantiation for this (n_ant, n_pol*n_avg); supported: n_ant 16/32/64/128 with n_ipo 2/32, "
           "n_ant 64 with n_ipo 2/4/8/16/32/64";
    return false;
}

LaunchShape fused_launch_shape(const Geometry& g, int n_units, int n_cus)
{
    LaunchShape ls{};
    const int L = g.n_ipo < 16 ? 16 : g.n_ipo;
    const int R = L / 16;
    const long long S = (long long)n_units * g.n_time;
    const long long halfstreams = S / L;
    const long long groups = (halfstreams + 1) / 2;
    const long long tiles = groups * R;
    ls.chunks_total = (int)((tiles + kTilesPerChunk - 1) / kTilesPerChunk);
    const int base = g.n_freq * g.n_bgroups;
    // aim for ~2 resident workgroups per CU, but never less than 1 chunk per workgroup
    int want = (2 * n_cus + base - 1) / base;
    if (want < 1) want = 1;
    if (want > ls.chunks_total) want = ls.chunks_total;
    ls.n_tsplit = want;
    ls.grid = base * ls.n_tsplit;
    ls.block = kWgThreads;
    const int rbc = g.nks <= 4 ? 8 : 16;
    ls.lds_bytes = 2 * kRowsPerChunk * rbc * 16;
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
    return a;
}

hipError_t launch_fused(const Geometry& g, const void* d_image, const void* d_packed, int n_units, float* d_out,
                        int n_cus, hipStream_t s)
{
    if (n_units <= 0) return hipSuccess;
    if ((long long)n_units * g.n_time > 0x7fffffffLL / 2) return hipErrorInvalidValue;
    const LaunchShape ls = fused_launch_shape(g, n_units, n_cus);
    const FusedArgs a = make_args(g, d_image, d_packed, n_units, d_out, ls);
    return dispatch_fused<false>(g, a, ls, s);
}

hipError_t launch_gemm_only(const Geometry& g, const void* d_image, const void* d_packed, float* d_c, int n_cus,
                            hipStream_t s)
{
    const LaunchShape ls = fused_launch_shape(g, 1, n_cus);
    const FusedArgs a = make_args(g, d_image, d_packed, 1, d_c, ls);
    return dispatch_fused<true>(g, a, ls, s);
}

hipError_t launch_weight_relayout(const Geometry& g, const int8_t* d_w, void* d_image, int* d_bad, hipStream_t s)
{
    const size_t total = weight_image_bytes(g) / 16;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(weight_relayout_kernel, dim3(grid), dim3(256), 0, s, d_w, static_cast<v4i*>(d_image), g.n_freq,
                       g.n_ant, g.n_beams, g.n_btiles, g.nks, d_bad);
    return hipGetLastError();
}

hipError_t launch_expand(const void* d_in, size_t nbytes, void* d_out, hipStream_t s)
{
    const size_t n_vec = nbytes / 16;
    if (n_vec == 0) return hipSuccess;
    size_t grid = (n_vec + 255) / 256;
    if (grid > 2048) grid = 2048;
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

int fused_vgprs(const Geometry& g)
{
    hipFuncAttributes attr{};
    const void* fn = nullptr;
#define X(nks_, nipo_) \
    if (g.nks == nks_ && g.n_ipo == nipo_) fn = reinterpret_cast<const void*>(fused_kernel<nks_, nipo_, false>);
    DSABF_FOR_EACH_VARIANT(X)
#undef X
    if (!fn || hipFuncGetAttributes(&attr, fn) != hipSuccess) return -1;
    return attr.numRegs;
}

}  // namespace dsabf
