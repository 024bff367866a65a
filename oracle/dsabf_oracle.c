/* dsabf_oracle.c -- CPU ORACLE (test infrastructure only; see dsabf_oracle.h for scope and pinning status).
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -fopenmp -ffp-contract=off; NO -ffast-math, NO -mfma, so every
 * float operation below is one IEEE-754 binary32/binary64 operation in source order).
 * Citations are file:line under /root/reference.
 */
#include "dsabf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/beamformer.hh:63-72,77,80 */
#define ORC_N_GPUS 8
#define ORC_TOT_CHANNELS 2048
#define ORC_START_F 1.28
#define ORC_END_F 1.53
#define ORC_ZERO_PT 0
#define ORC_BW_PER_CHANNEL ((ORC_END_F - ORC_START_F) / ORC_TOT_CHANNELS)
#define ORC_C_SPEED 299792458.0
#define ORC_PI 3.14159265358979
#define ORC_MAX_VAL 127
#define ORC_SIG_MAX_VAL 7
#define ORC_HALF_FOV 3.5
#define ORC_DEG2RAD(x) ((x) * ORC_PI / 180.0)

static int g_threads = 0;
static int g_contract = ORC_CONTRACT_NONE; /* how `x*x + y*y` of src/beamformer.cuh:151 is evaluated, see the header */

void orc_set_threads(int n) { g_threads = n; }

int orc_get_threads(void)
{
#ifdef _OPENMP
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

#ifdef _OPENMP
#define ORC_NT() (g_threads > 0 ? g_threads : omp_get_max_threads())
#endif

int orc_n_ipo(const orc_geom *g) { return g->n_pol * g->n_avg; }
int orc_n_time(const orc_geom *g) { return g->n_out_per_gemm * g->n_pol * g->n_avg; }
size_t orc_bytes_per_gemm(const orc_geom *g)
{
    return (size_t)g->n_ant * (size_t)g->n_freq * (size_t)orc_n_time(g);
}

/* src/beamformer.cu:173,233: `float bw_per_channel = BW_PER_CHANNEL;`
 * `float freq = END_F - (ZERO_PT + gpu*TOT_CHANNELS/(N_GPUS-1) + i)*bw_per_channel;`
 * int * float -> float product; double - float -> double; narrowed to float on assignment. */
float orc_freq_weights(int gpu, int chan)
{
    float bw_per_channel = ORC_BW_PER_CHANNEL;
    float freq = ORC_END_F - (ORC_ZERO_PT + gpu * ORC_TOT_CHANNELS / (ORC_N_GPUS - 1) + chan) * bw_per_channel;
    return freq;
}

/* src/test_data_generator.hh:72: same expression with the double macro BW_PER_CHANNEL (int * double). */
float orc_freq_generator(int gpu, int chan)
{
    float freq = ORC_END_F - (ORC_ZERO_PT + gpu * ORC_TOT_CHANNELS / (ORC_N_GPUS - 1) + chan) * ORC_BW_PER_CHANNEL;
    return freq;
}

/* src/beamformer.cu:135-147 */
void orc_default_positions(int n_ant, float *pos)
{
    memset(pos, 0, sizeof(float) * 3 * (size_t)n_ant);
    for (int i = 0; i < n_ant; i++)
        pos[3 * i + 0] = i * 500.0 / (n_ant - 1) - 250.0;
}

void orc_default_directions(int n_beams, float *dir)
{
    memset(dir, 0, sizeof(float) * 2 * (size_t)n_beams);
    for (int i = 0; i < n_beams; i++)
        dir[2 * i + 0] = i * ORC_DEG2RAD(2 * ORC_HALF_FOV) / (n_beams - 1) - ORC_DEG2RAD(ORC_HALF_FOV);
}

/* a5 -- src/beamformer.cu:230-241.  Under g++ the unqualified sin/cos/round on float arguments are the
 * double C functions: float operands are promoted, the phase is evaluated in double. */
void orc_make_weights(const orc_geom *g, const float *pos, const float *dir, int gpu, int8_t *out)
{
    const int nb = g->n_beams, na = g->n_ant, nf = g->n_freq;
#ifdef _OPENMP
#pragma omp parallel for num_threads(ORC_NT()) schedule(static)
#endif
    for (int i = 0; i < nf; i++) {
        float freq = orc_freq_weights(gpu, i);
        float wavelength = ORC_C_SPEED / (1E9 * freq);
        for (int j = 0; j < na; j++) {
            float px = pos[3 * j + 0], py = pos[3 * j + 1];
            for (int k = 0; k < nb; k++) {
                float theta = dir[2 * k + 0], phi = dir[2 * k + 1];
                int8_t *o = out + 2 * ((size_t)i * na * nb + (size_t)j * nb + k);
                o[0] = (int8_t)round(ORC_MAX_VAL * cos(-2 * ORC_PI * (px * sin(theta) + py * sin(phi)) / wavelength));
                o[1] = (int8_t)round(ORC_MAX_VAL * sin(-2 * ORC_PI * (px * sin(theta) + py * sin(phi)) / wavelength));
            }
        }
    }
}

/* a6 -- src/test_data_generator.hh:63-95 */
void orc_generate_test_data(const orc_geom *g, const float *pos, const float *src, int n_src, int gpu,
                            int batch_counter, int n_units, int literal, uint8_t *out)
{
    const int na = g->n_ant, nf = g->n_freq, nt = orc_n_time(g);
    const size_t per_gemm = orc_bytes_per_gemm(g);
#ifdef _OPENMP
#pragma omp parallel for num_threads(ORC_NT()) schedule(dynamic, 4)
#endif
    for (long direction = 0; direction < n_units; direction++) {
        int source_look_up = (int)direction + batch_counter * n_units; /* :77 */
        for (int i = 0; i < nf; i++) {
            float freq = orc_freq_generator(gpu, i);       /* :72 */
            float wavelength = ORC_C_SPEED / (1E9 * freq); /* :74 */
            uint8_t *slab = out + (size_t)direction * per_gemm + (size_t)i * nt * na;
            int jmax = literal ? nt : 1;
            for (int j = 0; j < jmax; j++) {
                for (int k = 0; k < na; k++) {
                    if (source_look_up < n_src) {
                        float px = pos[3 * k + 0], py = pos[3 * k + 1];
                        float theta = src[2 * source_look_up + 0], phi = src[2 * source_look_up + 1];
                        signed char high = (signed char)round(
                            ORC_SIG_MAX_VAL * cos(2 * ORC_PI * (px * sin(theta) + py * sin(phi)) / wavelength)); /* :80 */
                        signed char low = (signed char)round(
                            ORC_SIG_MAX_VAL * sin(2 * ORC_PI * (px * sin(theta) + py * sin(phi)) / wavelength)); /* :81 */
                        /* :83 `(high << 4) | (0x0F & low)` narrowed to char */
                        slab[(size_t)j * na + k] = (uint8_t)((((int)high) * 16) | (0x0F & (int)low));
                    } else {
                        slab[(size_t)j * na + k] = 0; /* :85 */
                    }
                }
            }
            for (int j = jmax; j < nt; j++) /* columns are identical: the expression above has no j */
                memcpy(slab + (size_t)j * na, slab, (size_t)na);
        }
    }
}

/* a1 -- src/beamformer.cuh:92-103 */
void orc_expand(const uint8_t *in, size_t n, int8_t *out)
{
    for (size_t i = 0; i < n; i++) {
        signed char temp = (signed char)in[i];
        signed char high = (signed char)(temp >> 4);                   /* :96 arithmetic shift, sign-extending */
        signed char low = (signed char)((unsigned char)temp << 4);     /* :97 */
        low = (signed char)(low >> 4);                                 /* :98 */
        out[2 * i + 0] = high;                                         /* :101 real */
        out[2 * i + 1] = low;                                          /* :102 imag */
    }
}

/* Exact integer complex dot products for one (f, t): acc[b] = sum_a W[a][b] * V[a].
 * wf = W[f] as [a][b]{re,im}; v = one expanded column [a]{re,im}. */
static void cdot_column(int na, int nb, const int8_t *wf, const int8_t *v, int32_t *acc_re, int32_t *acc_im)
{
    memset(acc_re, 0, sizeof(int32_t) * (size_t)nb);
    memset(acc_im, 0, sizeof(int32_t) * (size_t)nb);
    for (int a = 0; a < na; a++) {
        const int32_t vr = v[2 * a + 0], vi = v[2 * a + 1];
        const int8_t *wa = wf + 2 * (size_t)a * nb;
        for (int b = 0; b < nb; b++) {
            const int32_t wr = wa[2 * b + 0], wi = wa[2 * b + 1];
            acc_re[b] += wr * vr - wi * vi;
            acc_im[b] += wr * vi + wi * vr;
        }
    }
}


/* One term of detect_sum's `shmem[beam_idx] += input[i].x*input[i].x + input[i].y*input[i].y` (src/beamformer.cuh:151)
 * under the three ways a compiler may evaluate it (header: orc_set_detect_contract). */
static inline float power_term(float x, float y, int contract)
{
    if (contract == ORC_CONTRACT_NVCC) {
        const float yy = y * y;
        return fmaf(x, x, yy);          /* mul t, y, y ; fma p, x, x, t */
    }
    if (contract == ORC_CONTRACT_NVCC_ALT) {
        const float xx = x * x;
        return fmaf(y, y, xx);          /* mul t, x, x ; fma p, y, y, t */
    }
    const float xx = x * x;
    const float yy = y * y;
    return xx + yy;
}

void orc_set_detect_contract(int mode) { g_contract = mode; }
int orc_get_detect_contract(void) { return g_contract; }

/* a2 -- src/beamformer.cu:470-477 with alpha = (float)(1.0/127) (:191), beta = 0 (:193-194). */
void orc_gemm(const orc_geom *g, const int8_t *w, const int8_t *v, float *c)
{
    const int nb = g->n_beams, na = g->n_ant, nf = g->n_freq, nt = orc_n_time(g);
    const float alpha = 1.0 / ORC_MAX_VAL; /* h_inv_max_value.x = 1.0/MAX_VAL, src/beamformer.cu:191 */
#ifdef _OPENMP
#pragma omp parallel num_threads(ORC_NT())
#endif
    {
        int32_t *ar = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb * 2);
        int32_t *ai = ar + nb;
#ifdef _OPENMP
#pragma omp for collapse(2) schedule(static)
#endif
        for (int f = 0; f < nf; f++) {
            for (int t = 0; t < nt; t++) {
                cdot_column(na, nb, w + 2 * (size_t)f * na * nb, v + 2 * ((size_t)f * nt + t) * na, ar, ai);
                float *cc = c + 2 * ((size_t)f * nt + t) * nb;
                for (int b = 0; b < nb; b++) {
                    cc[2 * b + 0] = (float)ar[b] * alpha;
                    cc[2 * b + 1] = (float)ai[b] * alpha;
                }
            }
        }
        free(ar);
    }
}

/* a3 -- src/beamformer.cuh:139-154 */
void orc_detect(const orc_geom *g, const float *c, float *out)
{
    const int nb = g->n_beams, nf = g->n_freq, no = g->n_out_per_gemm, n_avg = orc_n_ipo(g);
    const int contract = g_contract;
#ifdef _OPENMP
#pragma omp parallel for collapse(2) num_threads(ORC_NT()) schedule(static)
#endif
    for (int o = 0; o < no; o++) {
        for (int f = 0; f < nf; f++) {
            for (int b = 0; b < nb; b++) {
                float acc = 0; /* shmem[beam_idx] = 0  :139 */
                const size_t input_idx = (size_t)f * no * n_avg * nb + (size_t)o * n_avg * nb + b; /* :141-145 */
                for (size_t i = input_idx; i < input_idx + (size_t)n_avg * nb; i += nb) {          /* :150 */
                    const float x = c[2 * i + 0], y = c[2 * i + 1];
                    const float p = power_term(x, y, contract);
                    acc = acc + p; /* :151 */
                }
                out[(size_t)o * nf * nb + (size_t)f * nb + b] = acc; /* :147,154 */
            }
        }
    }
}

/* a1+a2+a3 fused, same operations in the same order per output element. */
void orc_beamform(const orc_geom *g, const int8_t *w, const uint8_t *packed, int n_units, float *out)
{
    const int nb = g->n_beams, na = g->n_ant, nf = g->n_freq, no = g->n_out_per_gemm;
    const int n_ipo = orc_n_ipo(g), nt = orc_n_time(g);
    const float alpha = 1.0 / ORC_MAX_VAL;
    const size_t per_gemm = orc_bytes_per_gemm(g);
    const size_t out_per_gemm = (size_t)no * nf * nb;
    const int contract = g_contract;
#ifdef _OPENMP
#pragma omp parallel num_threads(ORC_NT())
#endif
    {
        int32_t *ar = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb * 2);
        int32_t *ai = ar + nb;
        float *acc = (float *)malloc(sizeof(float) * (size_t)nb);
        int8_t *col = (int8_t *)malloc((size_t)na * 2);
#ifdef _OPENMP
#pragma omp for collapse(2) schedule(static)
#endif
        for (int u = 0; u < n_units; u++) {
            for (int f = 0; f < nf; f++) {
                const int8_t *wf = w + 2 * (size_t)f * na * nb;
                for (int o = 0; o < no; o++) {
                    for (int b = 0; b < nb; b++)
                        acc[b] = 0;
                    for (int i = 0; i < n_ipo; i++) {
                        const int t = o * n_ipo + i;
                        orc_expand(packed + (size_t)u * per_gemm + ((size_t)f * nt + t) * na, (size_t)na, col);
                        cdot_column(na, nb, wf, col, ar, ai);
                        for (int b = 0; b < nb; b++) {
                            const float x = (float)ar[b] * alpha;
                            const float y = (float)ai[b] * alpha;
                            const float p = power_term(x, y, contract);
                            acc[b] = acc[b] + p;
                        }
                    }
                    memcpy(out + (size_t)u * out_per_gemm + (size_t)o * nf * nb + (size_t)f * nb, acc,
                           sizeof(float) * (size_t)nb);
                }
            }
        }
        free(col);
        free(acc);
        free(ar);
    }
}

/* The same path with NO floating-point rounding until the end: out = alpha^2 * sum_i (re_i^2 + im_i^2), the integer sum
 * exact in int64 (< 2^41), alpha = (float)(1.0/127) as a double, product rounded once to double (2^-53).  This is the
 * value every float evaluation order of src/beamformer.cuh:150-152 approximates; tests bound canonical / nvcc-contracted /
 * fast results against it. */
void orc_beamform_exact(const orc_geom *g, const int8_t *w, const uint8_t *packed, int n_units, double *out)
{
    const int nb = g->n_beams, na = g->n_ant, nf = g->n_freq, no = g->n_out_per_gemm;
    const int n_ipo = orc_n_ipo(g), nt = orc_n_time(g);
    const float alpha = 1.0 / ORC_MAX_VAL;
    const double a2 = (double)alpha * (double)alpha;
    const size_t per_gemm = orc_bytes_per_gemm(g);
    const size_t out_per_gemm = (size_t)no * nf * nb;
#ifdef _OPENMP
#pragma omp parallel num_threads(ORC_NT())
#endif
    {
        int32_t *ar = (int32_t *)malloc(sizeof(int32_t) * (size_t)nb * 2);
        int32_t *ai = ar + nb;
        int64_t *acc = (int64_t *)malloc(sizeof(int64_t) * (size_t)nb);
        int8_t *col = (int8_t *)malloc((size_t)na * 2);
#ifdef _OPENMP
#pragma omp for collapse(2) schedule(static)
#endif
        for (int u = 0; u < n_units; u++) {
            for (int f = 0; f < nf; f++) {
                const int8_t *wf = w + 2 * (size_t)f * na * nb;
                for (int o = 0; o < no; o++) {
                    for (int b = 0; b < nb; b++)
                        acc[b] = 0;
                    for (int i = 0; i < n_ipo; i++) {
                        const int t = o * n_ipo + i;
                        orc_expand(packed + (size_t)u * per_gemm + ((size_t)f * nt + t) * na, (size_t)na, col);
                        cdot_column(na, nb, wf, col, ar, ai);
                        for (int b = 0; b < nb; b++)
                            acc[b] += (int64_t)ar[b] * ar[b] + (int64_t)ai[b] * ai[b];
                    }
                    double *o_ = out + (size_t)u * out_per_gemm + (size_t)o * nf * nb + (size_t)f * nb;
                    for (int b = 0; b < nb; b++)
                        o_[b] = (double)acc[b] * a2;
                }
            }
        }
        free(col);
        free(acc);
        free(ar);
    }
}

/* a8 -- src/beamformer.cu:498-504: y = 1.0 * A * ones + 0 * y, A = d_out (first output), lda = N_BEAMS. */
void orc_dedisperse(const orc_geom *g, const float *out_unit, float *ded)
{
    const int nb = g->n_beams, nf = g->n_freq;
    for (int b = 0; b < nb; b++) {
        float acc = 0;
        for (int f = 0; f < nf; f++)
            acc = acc + out_unit[(size_t)f * nb + b] * 1.0f;
        ded[b] = acc;
    }
}

/* 8f-4 -- sandbox/Dispersion Theory.ipynb cells 1-2 (pinned by executing the cells: tests/golden/make_dispersion_golden.py). */
int orc_dm_trials(double dm0, double dm_max, int nchan, double epsilon, double nu_ghz, double chan_bw_mhz, double ti_us,
                  double tscat_us, double tsamp_us, double *out, int cap)
{
    const double n2 = (double)nchan * (double)nchan;
    const double alpha = 1.0 / (16 + n2);
    const double beta = ti_us * ti_us + tscat_us * tscat_us + tsamp_us * tsamp_us;
    const double k = (nu_ghz * nu_ghz * nu_ghz) / (8.3 * chan_bw_mhz);
    double dm_prev = dm0;
    int n = 0;
    if (n < cap)
        out[n] = dm0;
    n++;
    while (dm_prev < dm_max) {
        dm_prev = n2 * alpha * dm_prev +
                  4 * sqrt(alpha * (epsilon * epsilon - n2 * alpha) * dm_prev * dm_prev +
                           alpha * beta * (epsilon * epsilon - 1) * (k * k));
        if (n < cap)
            out[n] = dm_prev;
        n++;
    }
    return n < cap ? n : cap;
}

/* cell 5: int(d * (-f1**(-2) + f**(-2)) / tsamp_ms), d = 4.15 * DM */
void orc_dm_delays(const double *dms, int n_dm, const float *freq_ghz, int n_freq, double f_ref_ghz, double tsamp_ms,
                   int32_t *out)
{
    for (int d = 0; d < n_dm; d++)
        for (int f = 0; f < n_freq; f++) {
            const double fr = (double)freq_ghz[f];
            const double v = 4.15 * dms[d] * (-1.0 / (f_ref_ghz * f_ref_ghz) + 1.0 / (fr * fr)) / tsamp_ms;
            out[(size_t)d * n_freq + f] = (int32_t)v;
        }
}

void orc_dedisperse_dm(const float *series, int n_t, int n_freq, int n_beams, const int32_t *delays, int n_dm,
                       int n_t_out, float *out)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < n_dm; d++)
        for (int t = 0; t < n_t_out; t++) {
            float *o = out + ((size_t)d * n_t_out + t) * n_beams;
            for (int b = 0; b < n_beams; b++)
                o[b] = 0.0f;
            for (int f = 0; f < n_freq; f++) {
                const long row = (long)t + delays[(size_t)d * n_freq + f];
                if (row < 0 || row >= n_t)
                    continue;
                const float *p = series + ((size_t)row * n_freq + f) * n_beams;
                for (int b = 0; b < n_beams; b++)
                    o[b] = o[b] + p[b];
            }
        }
}

/* src/beamformer.hh:250-284: `in >> count; for (i < expected) in >> entry;` -- whitespace-separated floats;
 * once extraction fails (EOF or a non-numeric token such as the U+200B at the end of
 * config/linear_directions.txt) every later entry keeps its zero default. */
static int read_floats(const char *path, int expected, int per_entry, float *dst)
{
    FILE *fp = fopen(path, "r");
    if (!fp)
        return -1;
    int count = 0;
    if (fscanf(fp, "%d", &count) != 1) {
        fclose(fp);
        return 0;
    }
    int ok = 1;
    for (int i = 0; i < expected && ok; i++) {
        for (int c = 0; c < per_entry; c++) {
            char tok[128];
            if (fscanf(fp, "%127s", tok) != 1) {
                ok = 0;
                break;
            }
            char *end = NULL;
            float v = strtof(tok, &end);
            if (end == tok) {
                ok = 0;
                break;
            }
            dst[(size_t)i * per_entry + c] = v;
        }
    }
    fclose(fp);
    return count;
}

int orc_read_positions(const char *path, int expected, float *pos) { return read_floats(path, expected, 3, pos); }
int orc_read_directions(const char *path, int expected, float *dir) { return read_floats(path, expected, 2, dir); }

int orc_count_entries(const char *path)
{
    FILE *fp = fopen(path, "r");
    if (!fp)
        return -1;
    int count = 0;
    if (fscanf(fp, "%d", &count) != 1)
        count = 0;
    fclose(fp);
    return count;
}

/* src/beamformer.hh:287-311 */
int orc_write_python_file(const float *data, int rows, int cols, const char *path)
{
    FILE *fp = fopen(path, "w");
    if (!fp)
        return -1;
    fputs("A = [[", fp);
    for (int jj = 0; jj < rows; jj++) {
        for (int ii = 0; ii < cols; ii++) {
            fprintf(fp, "%g", (double)data[(size_t)jj * cols + ii]);
            if (ii != cols - 1)
                fputc(',', fp);
        }
        if (jj != rows - 1)
            fputs("],\n[", fp);
        else
            fputs("]]\n", fp);
    }
    fclose(fp);
    return 0;
}

uint64_t orc_fnv1a64(const void *buf, size_t n)
{
    const unsigned char *p = (const unsigned char *)buf;
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < n; i++) {
        h ^= p[i];
        h *= 0x100000001b3ULL;
    }
    return h;
}
