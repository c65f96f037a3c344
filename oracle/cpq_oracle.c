/*
 * cpq_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See cpq_oracle.h for the parity status ("parity unpinned" for the convolver
 * and SVF band kernel; fastTanh / EQParameters pinned through oracle/_ref).
 *
 * Build: see oracle/Makefile  (-O2 -mavx2 -mfma -ffp-contract=off: FMA only
 * where the reference source writes an FMA intrinsic).
 */
#include "cpq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_LAYERS 3
#define ORC_L0_MAX_PARTS 32  /* kL0MaxParts, src/MKLNonUniformConvolver.h:392 */
#define ORC_L1_MAX_PARTS 64  /* kL1MaxParts, :393 */

static int    imin(int a, int b) { return a < b ? a : b; }
static int    imax(int a, int b) { return a > b ? a : b; }
static int    iclamp(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }   /* juce::jlimit */
static double dclamp(double lo, double hi, double v) { return v < lo ? lo : (v > hi ? hi : v); }
static double dmaxd(double a, double b) { return a > b ? a : b; }

static int next_pow2(int v) /* juce::nextPowerOfTwo */
{
    int p = 1;
    if (v <= 1) return 1;
    while (p < v) p <<= 1;
    return p;
}

static double* alloc_d(size_t n)
{
    void* p = NULL;
    if (n == 0) n = 1;
    if (posix_memalign(&p, 64, n * sizeof(double)) != 0) return NULL;
    memset(p, 0, n * sizeof(double));
    return (double*)p;
}

/* =============================================================== RNG ===== */

uint64_t orc_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

double orc_rand_pm1(uint64_t seed, uint64_t stream, uint64_t channel, uint64_t index)
{
    const uint64_t u = orc_splitmix64(seed ^ (stream << 40) ^ (channel << 32) ^ index);
    return (double)(u >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;
}

void orc_gen_pcm(double* x, int64_t n, uint64_t seed, int stream, int channel, int64_t start)
{
    for (int64_t i = 0; i < n; ++i)
        x[i] = 0.25 * orc_rand_pm1(seed, (uint64_t)stream, (uint64_t)channel, (uint64_t)(start + i));
}

void orc_gen_ir(double* h, int len, uint64_t seed, int stream, int channel)
{
    for (int i = 0; i < len; ++i)
        h[i] = 0.05 * orc_rand_pm1(seed, (uint64_t)stream, (uint64_t)channel, (uint64_t)i)
             * exp(-6.9 * (double)i / (double)len);
}

/* =============================================================== FFT ===== */

struct orc_fft {
    int     n;       /* real length */
    int     m;       /* n/2 complex points */
    int     logm;
    double* wr;      /* exp(-2 pi i j / m), j < m/2 */
    double* wi;
    double* ur;      /* exp(-2 pi i k / n), k <= m */
    double* ui;
    int*    rev;
    double* sre;     /* scratch, one transform at a time per plan */
    double* sim;
};

orc_fft* orc_fft_create(int n)
{
    if (n < 4 || (n & (n - 1)) != 0) return NULL;
    orc_fft* f = (orc_fft*)calloc(1, sizeof(orc_fft));
    if (!f) return NULL;
    f->n = n;
    f->m = n / 2;
    f->logm = 0;
    while ((1 << f->logm) < f->m) ++f->logm;
    f->wr = alloc_d((size_t)f->m / 2 + 1);
    f->wi = alloc_d((size_t)f->m / 2 + 1);
    f->ur = alloc_d((size_t)f->m + 1);
    f->ui = alloc_d((size_t)f->m + 1);
    f->rev = (int*)malloc(sizeof(int) * (size_t)f->m);
    f->sre = alloc_d((size_t)f->m);
    f->sim = alloc_d((size_t)f->m);
    if (!f->wr || !f->wi || !f->ur || !f->ui || !f->rev || !f->sre || !f->sim) { orc_fft_destroy(f); return NULL; }
    const long double twopi = 6.283185307179586476925286766559005768L;
    for (int j = 0; j < f->m / 2; ++j) {
        const long double a = -twopi * (long double)j / (long double)f->m;
        f->wr[j] = (double)cosl(a);
        f->wi[j] = (double)sinl(a);
    }
    for (int k = 0; k <= f->m; ++k) {
        const long double a = -twopi * (long double)k / (long double)n;
        f->ur[k] = (double)cosl(a);
        f->ui[k] = (double)sinl(a);
    }
    for (int i = 0; i < f->m; ++i) {
        int r = 0;
        for (int b = 0; b < f->logm; ++b)
            if (i & (1 << b)) r |= 1 << (f->logm - 1 - b);
        f->rev[i] = r;
    }
    return f;
}

void orc_fft_destroy(orc_fft* f)
{
    if (!f) return;
    free(f->wr); free(f->wi); free(f->ur); free(f->ui); free(f->rev); free(f->sre); free(f->sim);
    free(f);
}

/* in-place radix-2 DIT on m complex points held as separate re/im arrays; sign=-1 forward, +1 inverse */
static void cfft_inplace(const orc_fft* f, double* re, double* im, int sign)
{
    const int m = f->m;
    for (int i = 0; i < m; ++i) {
        const int r = f->rev[i];
        if (r > i) {
            double t = re[i]; re[i] = re[r]; re[r] = t;
            t = im[i]; im[i] = im[r]; im[r] = t;
        }
    }
    for (int half = 1; half < m; half <<= 1) {
        const int step = m / (2 * half);
        for (int base = 0; base < m; base += 2 * half) {
            for (int j = 0; j < half; ++j) {
                const double wr = f->wr[j * step];
                const double wi = (sign < 0) ? f->wi[j * step] : -f->wi[j * step];
                const int a = base + j, b = a + half;
                const double tr = re[b] * wr - im[b] * wi;
                const double ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] = re[a] + tr; im[a] = im[a] + ti;
            }
        }
    }
}

void orc_fft_fwd_ccs(orc_fft* f, const double* in, double* ccs)
{
    const int m = f->m;
    double* re = f->sre;
    double* im = f->sim;
    for (int i = 0; i < m; ++i) { re[i] = in[2 * i]; im[i] = in[2 * i + 1]; }
    cfft_inplace(f, re, im, -1);
    for (int k = 0; k <= m; ++k) {
        const int k1 = (k == m) ? 0 : k;
        const int k2 = (k == 0) ? 0 : m - k;
        const double zr = re[k1], zi = im[k1];
        const double cr = re[k2], ci = -im[k2];              /* conj(Z[m-k]) */
        const double er = 0.5 * (zr + cr), ei = 0.5 * (zi + ci);
        const double dr = 0.5 * (zr - cr), di = 0.5 * (zi - ci);
        /* O = -i * d */
        const double orr = di, oi = -dr;
        const double wr = f->ur[k], wi = f->ui[k];
        ccs[2 * k]     = er + (orr * wr - oi * wi);
        ccs[2 * k + 1] = ei + (orr * wi + oi * wr);
    }
    ccs[1] = 0.0;
    ccs[2 * m + 1] = 0.0;
}

void orc_fft_inv_ccs(orc_fft* f, const double* ccs, double* out)
{
    const int m = f->m;
    double* re = f->sre;
    double* im = f->sim;
    for (int k = 0; k < m; ++k) {
        const double xr = ccs[2 * k], xi = (k == 0) ? 0.0 : ccs[2 * k + 1];
        const int k2 = m - k;
        const double cr = ccs[2 * k2], ci = (k2 == m) ? 0.0 : -ccs[2 * k2 + 1];  /* conj(X[m-k]) */
        const double er = 0.5 * (xr + cr), ei = 0.5 * (xi + ci);
        const double dr = 0.5 * (xr - cr), di = 0.5 * (xi - ci);
        /* O = d * conj(W^k) */
        const double wr = f->ur[k], wi = -f->ui[k];
        const double orr = dr * wr - di * wi;
        const double oi  = dr * wi + di * wr;
        /* Z = E + i O */
        re[k] = er - oi;
        im[k] = ei + orr;
    }
    cfft_inplace(f, re, im, +1);
    const double s = 1.0 / (double)m;
    for (int i = 0; i < m; ++i) { out[2 * i] = re[i] * s; out[2 * i + 1] = im[i] * s; }
}

/* =============================================================== NUC ===== */

typedef struct {
    int fftSize, partSize, numParts, numPartsIR, fdlMask, complexSize;
    int isImmediate;
    double *irRe, *irIm;          /* [numParts][complexSize], partition order reversed (:959-985) */
    double *fdlRe, *fdlIm;        /* [2*numParts][complexSize] with mirror (:1275-1283) */
    double *timeBuf, *outBuf, *prevIn, *accRe, *accIm, *ccs, *inputAcc;
    int fdlIndex, inputPos;
    double* tailOut;
    int outputDelay, delayCap;
    double* delayBuf;
    uint64_t delayW, delayR;
    int partsPerCallback, nextPart, baseFdlIdxSaved, distributing;
    orc_fft* fft;
} orc_layer;

struct orc_nuc {
    orc_layer L[ORC_MAX_LAYERS];
    int numLayers, latency, ready;
    double* ring; int ringSize, ringMask, ringW, ringR, ringAvail, ringOverflow;
    int directTaps, directHistLen, directMaxBlock, directPending, directEnabled;
    double *directIRRev, *directHist, *directWin, *directOut;
    int tailEnabled, maxBlock;
    double tailStrength, layerGain[3];
    orc_nuc_plan plan;
};

static void layer_free(orc_layer* l)
{
    free(l->irRe); free(l->irIm); free(l->fdlRe); free(l->fdlIm);
    free(l->timeBuf); free(l->outBuf); free(l->prevIn); free(l->accRe); free(l->accIm);
    free(l->ccs); free(l->inputAcc); free(l->tailOut); free(l->delayBuf);
    orc_fft_destroy(l->fft);
    memset(l, 0, sizeof(*l));
}

static void nuc_release(orc_nuc* c)
{
    for (int i = 0; i < ORC_MAX_LAYERS; ++i) layer_free(&c->L[i]);
    free(c->ring); c->ring = NULL;
    free(c->directIRRev); free(c->directHist); free(c->directWin); free(c->directOut);
    c->directIRRev = c->directHist = c->directWin = c->directOut = NULL;
    c->numLayers = 0; c->ready = 0;
}

orc_nuc* orc_nuc_create(void) { return (orc_nuc*)calloc(1, sizeof(orc_nuc)); }
void orc_nuc_destroy(orc_nuc* c) { if (c) { nuc_release(c); free(c); } }
int  orc_nuc_latency(const orc_nuc* c) { return c ? c->latency : 0; }
int  orc_nuc_get_plan(const orc_nuc* c, orc_nuc_plan* p) { if (!c || !c->ready) return -1; *p = c->plan; return 0; }

/* src/MKLNonUniformConvolver.cpp:626-684 (tail profile), :689-695 (direct head),
 * :738-758 (layer lengths), :784-786, :988-994 (partsPerCallback), :1005-1024 (B13 delay) */
int orc_nuc_plan_compute(int irLen, int blockSize, int enableDirectHead,
                         const orc_filter_spec* spec, orc_nuc_plan* p)
{
    if (irLen <= 0 || blockSize <= 0 || !p) return -1;
    memset(p, 0, sizeof(*p));

    const int    tailMode   = spec ? iclamp(0, 2, spec->tailMode) : 1;
    const int    tailEnabled = (tailMode != 2) && (spec ? (spec->tailEnabled != 0) : 1);
    const double srTail     = spec ? spec->sampleRate : 48000.0;
    double       tailStartSec = spec ? dclamp(0.01, 0.80, spec->tailStartSeconds) : 0.085;
    const double userStrength = spec ? dclamp(0.0, 2.0, spec->tailStrength) : 1.0;
    double       tailStrength = userStrength;
    int          mult = spec ? iclamp(2, 16, spec->tailL1L2Multiplier) : 8;
    double g1 = 1.0, g2 = 1.0;
    const double s01 = dclamp(0.0, 1.0, userStrength * 0.5);

    if (!tailEnabled) {
        tailStrength = 0.0; g1 = 0.0; g2 = 0.0;
    } else if (tailMode == 0) {
        tailStartSec = dclamp(0.01, 0.80, dmaxd(tailStartSec, 0.055));
        mult = iclamp(2, 16, imax(mult, 6));
        tailStrength = dclamp(0.0, 2.0, userStrength);
        g1 = dclamp(0.0, 2.0, tailStrength * (0.95 - 0.25 * s01));
        g2 = dclamp(0.0, 2.0, tailStrength * (0.80 - 0.45 * s01));
    } else if (tailMode == 1) {
        tailStartSec = dclamp(0.01, 0.80, dmaxd(tailStartSec, 0.12));
        tailStrength = dclamp(0.0, 2.0, dmaxd(tailStrength, 1.25));
        mult = iclamp(2, 16, imax(mult, 8));
        g1 = dclamp(0.0, 2.0, tailStrength * (1.05 + 0.20 * s01));
        g2 = dclamp(0.0, 2.0, tailStrength * (0.82 + 0.12 * s01));
    } else {
        tailStrength = 0.0; g1 = 0.0; g2 = 0.0;
    }

    const int directPart = next_pow2(imax(blockSize, 64));
    p->directTaps = enableDirectHead ? imin(irLen, imin(directPart, 32)) : 0;

    const int l0Part = next_pow2(imax(blockSize, 64));
    const int l1Part = l0Part * mult;
    const int l2Part = l1Part * mult;
    const int l0MaxLen = ORC_L0_MAX_PARTS * l0Part;
    const int l0ByTail = (int)llround(tailStartSec * srTail);
    const int l0Target = iclamp(l0Part, l0MaxLen, l0ByTail);
    const int l0Len = imin(irLen, tailEnabled ? l0Target : l0MaxLen);
    const int l1Len = tailEnabled ? imax(0, imin(irLen - l0Len, ORC_L1_MAX_PARTS * l1Part)) : 0;
    const int l2Len = tailEnabled ? imax(0, irLen - l0Len - l1Len) : 0;

    const int offs[3]  = { 0, l0Len, l0Len + l1Len };
    const int lens[3]  = { l0Len, l1Len, l2Len };
    const int parts[3] = { l0Part, l1Part, l2Part };
    const double gains[3] = { 1.0, g1, g2 };

    int prevTotal = 0, n = 0;
    p->ltiValid = 1;
    for (int li = 0; li < 3; ++li) {
        if (lens[li] <= 0) continue;
        p->partSize[n] = parts[li];
        p->offset[n] = offs[li];
        p->len[n] = lens[li];
        p->numPartsIR[n] = (lens[li] + parts[li] - 1) / parts[li];
        p->numParts[n] = next_pow2(p->numPartsIR[n]);
        /* m_tailLayerGain is indexed by active-layer index in Get() (:1626-1628) but filled by
         * config index (:682-684); both agree because only trailing layers can be empty. */
        p->gain[n] = gains[li];
        if (li > 0) {
            const int bpp = (parts[li] + imax(blockSize, 1) - 1) / imax(blockSize, 1);
            int ppc = imax(1, (p->numPartsIR[n] + bpp - 1) / bpp);
            ppc = imin(ppc, p->numPartsIR[n]);
            p->partsPerCallback[n] = ppc;
        }
        p->outputDelay[n] = (prevTotal > 0) ? prevTotal : 0;
        if (n > 0) {
            /* A6: first tail block is written at callback c_done, read in the same callback */
            const int fillCb = (parts[li] + blockSize - 1) / blockSize - 1;
            const int ppc = p->partsPerCallback[n];
            const int cbs = (p->numPartsIR[n] + ppc - 1) / ppc;
            p->doneCallback[n] = fillCb + cbs - 1;
            p->lag[n] = p->doneCallback[n] * blockSize - p->offset[n];
            if (p->partSize[n] > p->outputDelay[n]) p->ltiValid = 0;
        }
        prevTotal += lens[li];
        ++n;
    }
    p->numLayers = n;
    p->latency = l0Part;
    /* A6 also assumes the caller feeds whole L0 partitions per callback */
    if (blockSize != l0Part) p->ltiValid = 0;
    return n > 0 ? 0 : -1;
}

int orc_nuc_heff(const double* ir, int irLen, int blockSize, double scale,
                 const orc_filter_spec* spec, double* heff, int cap)
{
    orc_nuc_plan p;
    if (orc_nuc_plan_compute(irLen, blockSize, 0, spec, &p) != 0) return -1;
    int need = p.len[0];
    for (int l = 1; l < p.numLayers; ++l)
        need = imax(need, p.offset[l] + p.lag[l] + p.len[l]);
    if (heff) {
        for (int i = 0; i < cap; ++i) heff[i] = 0.0;
        for (int l = 0; l < p.numLayers; ++l) {
            const int pos = p.offset[l] + p.lag[l];
            for (int i = 0; i < p.len[l]; ++i) {
                const int d = pos + i;
                if (d >= 0 && d < cap) heff[d] += p.gain[l] * (ir[p.offset[l] + i] * scale);
            }
        }
    }
    return need;
}

/* split-complex MAC, src/MKLNonUniformConvolver.cpp:150-195: mul/add, no FMA */
static void mac_split(const double* ar, const double* ai, const double* br, const double* bi,
                      double* dr, double* di, int n)
{
    for (int k = 0; k < n; ++k) {
        dr[k] = dr[k] + (ar[k] * br[k] - ai[k] * bi[k]);
        di[k] = di[k] + (ar[k] * bi[k] + ai[k] * br[k]);
    }
}

/* HC/LC spectral gains, src/MKLNonUniformConvolver.cpp:336-443 */
static void apply_spectrum_filter(orc_nuc* c, const orc_filter_spec* s)
{
    const double fs = s->sampleRate;
    const double nyq = fs * 0.5;
    const double hcStart = (fs <= 48000.0) ? 18000.0 : 22000.0;
    const double hcEnd = nyq;
    const double lcEnd = (s->lcMode == 1) ? 6.0 : 8.0;
    const double lcStart = (s->lcMode == 1) ? 15.0 : 18.0;
    for (int li = 0; li < c->numLayers; ++li) {
        orc_layer* l = &c->L[li];
        const int N = l->fftSize, halfN = N / 2, cs = l->complexSize;
        double* g = alloc_d((size_t)cs);
        for (int k = 0; k < cs; ++k) g[k] = 1.0;
        {
            const int kS = (int)round(hcStart * N / fs);
            const int kE = imin(halfN, (int)round(hcEnd * N / fs));
            for (int k = 0; k < cs; ++k) {
                if (k <= kS) continue;
                if (k <= kE) {
                    const double x = (double)(k - kS) / (double)(kE - kS);
                    if (s->hcMode == 0)      g[k] = 1.0 / sqrt(1.0 + pow(x, 8.0));
                    else if (s->hcMode == 1) g[k] = 0.5 * (1.0 + cos(M_PI * x));
                    else                     g[k] = exp(-4.60517 * x * x);
                }
            }
        }
        {
            const int kE = (int)round(lcEnd * N / fs);
            const int kS = (int)round(lcStart * N / fs);
            for (int k = 0; k < cs; ++k) {
                if (k <= kE) g[k] = 0.0;
                else if (k < kS) {
                    const double x = (double)(k - kE) / (double)imax(1, kS - kE);
                    g[k] *= 0.5 * (1.0 - cos(M_PI * x));
                }
            }
        }
        for (int p = 0; p < l->numParts; ++p) {
            double* re = l->irRe + (size_t)p * cs;
            double* im = l->irIm + (size_t)p * cs;
            for (int k = 0; k < cs; ++k) { re[k] *= g[k]; im[k] *= g[k]; }
        }
        free(g);
    }
}

int orc_nuc_set_impulse(orc_nuc* c, const double* ir, int irLen, int blockSize,
                        double scale, int enableDirectHead, const orc_filter_spec* spec)
{
    if (!c) return 0;
    c->ready = 0;
    if (!ir || irLen <= 0 || blockSize <= 0) return 0;
    nuc_release(c);

    orc_nuc_plan p;
    if (orc_nuc_plan_compute(irLen, blockSize, enableDirectHead, spec, &p) != 0) return 0;
    c->plan = p;

    const int    tailMode = spec ? iclamp(0, 2, spec->tailMode) : 1;
    const int    tailEnabled = (tailMode != 2) && (spec ? (spec->tailEnabled != 0) : 1);
    c->tailEnabled = tailEnabled;
    c->maxBlock = blockSize;
    c->layerGain[0] = 1.0; c->layerGain[1] = 1.0; c->layerGain[2] = 1.0;
    for (int l = 0; l < p.numLayers; ++l) c->layerGain[l] = p.gain[l];

    /* direct head, :689-718 */
    c->directTaps = p.directTaps;
    c->directHistLen = imax(0, c->directTaps - 1);
    c->directMaxBlock = imax(blockSize, 1);
    c->directPending = 0;
    c->directEnabled = c->directTaps > 0;
    if (c->directEnabled) {
        c->directIRRev = alloc_d((size_t)c->directTaps);
        c->directHist = alloc_d((size_t)c->directHistLen);
        c->directWin = alloc_d((size_t)(c->directHistLen + c->directMaxBlock));
        c->directOut = alloc_d((size_t)c->directMaxBlock);
        for (int i = 0; i < c->directTaps; ++i)
            c->directIRRev[i] = ir[c->directTaps - 1 - i] * scale;
    }

    double* irFft = alloc_d((size_t)irLen);
    memcpy(irFft, ir, sizeof(double) * (size_t)irLen);
    if (c->directEnabled) memset(irFft, 0, sizeof(double) * (size_t)c->directTaps);   /* :730-731 */

    for (int li = 0; li < p.numLayers; ++li) {
        orc_layer* l = &c->L[li];
        l->partSize = p.partSize[li];
        l->fftSize = 2 * l->partSize;
        l->isImmediate = (li == 0);
        l->complexSize = l->fftSize / 2 + 1;
        l->numPartsIR = p.numPartsIR[li];
        l->numParts = p.numParts[li];
        l->fdlMask = l->numParts - 1;
        l->fft = orc_fft_create(l->fftSize);
        const size_t cs = (size_t)l->complexSize;
        l->irRe = alloc_d((size_t)l->numParts * cs);
        l->irIm = alloc_d((size_t)l->numParts * cs);
        l->fdlRe = alloc_d((size_t)l->numParts * 2 * cs);
        l->fdlIm = alloc_d((size_t)l->numParts * 2 * cs);
        l->timeBuf = alloc_d((size_t)l->fftSize);
        l->outBuf = alloc_d((size_t)l->fftSize);
        l->prevIn = alloc_d((size_t)l->partSize);
        l->accRe = alloc_d(cs);
        l->accIm = alloc_d(cs);
        l->ccs = alloc_d((size_t)l->fftSize + 2);
        l->inputAcc = alloc_d((size_t)l->partSize);
        if (!l->isImmediate) l->tailOut = alloc_d((size_t)l->partSize);

        /* IR pre-FFT per partition, :919-946 */
        double* tt = alloc_d((size_t)l->fftSize);
        const double* src = irFft + p.offset[li];
        const int remain = p.len[li];
        for (int q = 0; q < l->numParts; ++q) {
            memset(tt, 0, sizeof(double) * (size_t)l->fftSize);
            if (q < l->numPartsIR) {
                const int cstart = q * l->partSize;
                const int clen = imin(l->partSize, remain - cstart);
                if (clen > 0) memcpy(tt, src + cstart, sizeof(double) * (size_t)clen);
            }
            orc_fft_fwd_ccs(l->fft, tt, l->ccs);
            if (fabs(scale - 1.0) > 1e-12)
                for (int k = 0; k < l->complexSize * 2; ++k) l->ccs[k] *= scale;      /* cblas_dscal :939-940 */
            for (int k = 0; k < l->complexSize; ++k) {
                l->irRe[(size_t)q * cs + k] = l->ccs[2 * k];
                l->irIm[(size_t)q * cs + k] = l->ccs[2 * k + 1];
            }
        }
        free(tt);
        /* reverse partition order, :959-985 */
        for (int pf = 0; pf < l->numPartsIR / 2; ++pf) {
            const int pb = l->numPartsIR - 1 - pf;
            for (int k = 0; k < l->complexSize; ++k) {
                double t = l->irRe[(size_t)pf * cs + k];
                l->irRe[(size_t)pf * cs + k] = l->irRe[(size_t)pb * cs + k];
                l->irRe[(size_t)pb * cs + k] = t;
                t = l->irIm[(size_t)pf * cs + k];
                l->irIm[(size_t)pf * cs + k] = l->irIm[(size_t)pb * cs + k];
                l->irIm[(size_t)pb * cs + k] = t;
            }
        }
        l->partsPerCallback = p.partsPerCallback[li];
        l->outputDelay = p.outputDelay[li];
        if (l->outputDelay > 0) {   /* :1005-1019 */
            l->delayCap = ((l->outputDelay + l->partSize + c->maxBlock + 15) / 16) * 16;
            l->delayBuf = alloc_d((size_t)l->delayCap);
        }
    }
    free(irFft);
    c->numLayers = p.numLayers;

    /* output ring, :1033-1053 */
    {
        const int l0p = c->L[0].partSize;
        const int nPartsIR = (irLen + blockSize - 1) / blockSize;
        const int nParts = next_pow2(nPartsIR);
        const int rSize = next_pow2(nParts * 2 + next_pow2(blockSize));
        const int minSize = next_pow2(l0p * 4 + blockSize * 4);
        c->ringSize = imax(rSize, minSize);
        c->ringMask = c->ringSize - 1;
        c->ring = alloc_d((size_t)c->ringSize);
        c->ringW = c->ringR = c->ringAvail = 0;
    }
    c->latency = c->L[0].partSize;

    if (spec && spec->applySpectrumFilter) {
        apply_spectrum_filter(c, spec);
        if (tailEnabled && tailMode == 0) {   /* air absorption damping, :1060-1097 */
            const double userStrength = dclamp(0.0, 2.0, spec->tailStrength);
            const double s01 = dclamp(0.0, 1.0, userStrength * 0.5);
            double tss = dclamp(0.01, 0.80, spec->tailStartSeconds);
            tss = dclamp(0.01, 0.80, dmaxd(tss, 0.055));
            const double startNorm = dclamp(0.65, 1.55, tss / 0.085);
            const double base = (0.35 + 1.10 * s01) * startNorm;
            for (int li = 1; li < c->numLayers; ++li) {
                orc_layer* l = &c->L[li];
                const double coeff = base * ((li == 1) ? 1.0 : 1.6);
                const double denom = (double)imax(1, l->complexSize - 1);
                for (int q = 0; q < l->numParts; ++q)
                    for (int k = 0; k < l->complexSize; ++k) {
                        const double fn = (double)k / denom;
                        const double gk = exp(-coeff * fn * fn);
                        l->irRe[(size_t)q * l->complexSize + k] *= gk;
                        l->irIm[(size_t)q * l->complexSize + k] *= gk;
                    }
            }
        }
    }
    c->ready = 1;
    return 1;
}

/* :1341-1371 */
static void ring_write(orc_nuc* c, const double* src, int n)
{
    if (n <= 0 || !c->ring) return;
    const int first = imin(n, c->ringSize - c->ringW);
    memcpy(c->ring + c->ringW, src, sizeof(double) * (size_t)first);
    if (n > first) memcpy(c->ring, src + first, sizeof(double) * (size_t)(n - first));
    c->ringW = (c->ringW + n) & c->ringMask;
    const int next = c->ringAvail + n;
    if (next > c->ringSize) {
        c->ringR = (c->ringR + (next - c->ringSize)) & c->ringMask;
        c->ringAvail = c->ringSize;
        ++c->ringOverflow;
    } else c->ringAvail = next;
}

/* :1376-1402 */
static int ring_read(orc_nuc* c, double* dst, int n)
{
    if (n <= 0 || !c->ring) return 0;
    const int toRead = imin(n, c->ringAvail);
    if (toRead == 0) { if (dst) memset(dst, 0, sizeof(double) * (size_t)n); return 0; }
    const int first = imin(toRead, c->ringSize - c->ringR);
    if (dst) {
        memcpy(dst, c->ring + c->ringR, sizeof(double) * (size_t)first);
        if (toRead > first) memcpy(dst + first, c->ring, sizeof(double) * (size_t)(toRead - first));
        if (toRead < n) memset(dst + toRead, 0, sizeof(double) * (size_t)(n - toRead));
    }
    c->ringR = (c->ringR + toRead) & c->ringMask;
    c->ringAvail -= toRead;
    return toRead;
}

/* overlap-save frame + forward FFT + FDL push with mirror, :1256-1283 / :1456-1479 */
static void layer_push_fdl(orc_layer* l)
{
    memcpy(l->timeBuf, l->prevIn, sizeof(double) * (size_t)l->partSize);
    memcpy(l->timeBuf + l->partSize, l->inputAcc, sizeof(double) * (size_t)l->partSize);
    memcpy(l->prevIn, l->inputAcc, sizeof(double) * (size_t)l->partSize);
    orc_fft_fwd_ccs(l->fft, l->timeBuf, l->ccs);
    const size_t cs = (size_t)l->complexSize;
    double* r0 = l->fdlRe + (size_t)l->fdlIndex * cs;
    double* i0 = l->fdlIm + (size_t)l->fdlIndex * cs;
    double* r1 = l->fdlRe + (size_t)(l->fdlIndex + l->numParts) * cs;
    double* i1 = l->fdlIm + (size_t)(l->fdlIndex + l->numParts) * cs;
    for (int k = 0; k < l->complexSize; ++k) {
        r0[k] = r1[k] = l->ccs[2 * k];
        i0[k] = i1[k] = l->ccs[2 * k + 1];
    }
}

static void layer_inverse(orc_layer* l)
{
    for (int k = 0; k < l->complexSize; ++k) { l->ccs[2 * k] = l->accRe[k]; l->ccs[2 * k + 1] = l->accIm[k]; }
    orc_fft_inv_ccs(l->fft, l->ccs, l->outBuf);
}

/* :1245-1336 */
static void process_layer_block(orc_nuc* c, orc_layer* l)
{
    layer_push_fdl(l);
    const size_t cs = (size_t)l->complexSize;
    memset(l->accRe, 0, sizeof(double) * cs);
    memset(l->accIm, 0, sizeof(double) * cs);
    const int linStart = l->fdlIndex - l->numPartsIR + 1 + l->numParts;
    for (int p = 0; p < l->numPartsIR; ++p) {
        const size_t idx = (size_t)(linStart + p);
        mac_split(l->fdlRe + idx * cs, l->fdlIm + idx * cs,
                  l->irRe + (size_t)p * cs, l->irIm + (size_t)p * cs, l->accRe, l->accIm, l->complexSize);
    }
    /* killDenormalV is a no-op in Release (src/DspNumericPolicy.h:189-204) */
    layer_inverse(l);
    ring_write(c, l->outBuf + l->partSize, l->partSize);
    l->fdlIndex = (l->fdlIndex + 1) & l->fdlMask;
}

/* :1639-1648 */
static void delay_write(orc_layer* l, const double* src, int n)
{
    const size_t off = (size_t)(l->delayW % (uint64_t)l->delayCap);
    const int first = imin(n, l->delayCap - (int)off);
    memcpy(l->delayBuf + off, src, sizeof(double) * (size_t)first);
    if (first < n) memcpy(l->delayBuf, src + first, sizeof(double) * (size_t)(n - first));
    l->delayW += (uint64_t)n;
}

/* :1653-1688 */
static void delay_read_add(orc_layer* l, double* dst, int n, double gain)
{
    if (!l->delayBuf || l->delayCap <= 0 || !dst) return;
    const uint64_t maxRead = (l->delayW >= (uint64_t)l->outputDelay) ? (l->delayW - (uint64_t)l->outputDelay) : 0;
    const uint64_t start = (l->delayR > maxRead) ? l->delayR : maxRead;
    if (start + (uint64_t)n > l->delayW) return;
    const size_t off = (size_t)(start % (uint64_t)l->delayCap);
    const int first = imin(n, l->delayCap - (int)off);
    const int unity = fabs(gain - 1.0) < 1.0e-12;
    for (int i = 0; i < first; ++i) dst[i] += unity ? l->delayBuf[off + i] : l->delayBuf[off + i] * gain;
    for (int i = first; i < n; ++i) dst[i] += unity ? l->delayBuf[i - first] : l->delayBuf[i - first] * gain;
    l->delayR = start + (uint64_t)n;
}

/* :1169-1232; two 4-lane FMA accumulators then the hsum of :108-115 */
static void process_direct(orc_nuc* c, const double* in, int n)
{
    if (!c->directEnabled || n <= 0) return;
    if (n > c->directMaxBlock) { c->directPending = 0; return; }
    memset(c->directOut, 0, sizeof(double) * (size_t)n);
    if (c->directHistLen > 0) memcpy(c->directWin, c->directHist, sizeof(double) * (size_t)c->directHistLen);
    if (in) memcpy(c->directWin + c->directHistLen, in, sizeof(double) * (size_t)n);
    else memset(c->directWin + c->directHistLen, 0, sizeof(double) * (size_t)n);
    const int v8 = (c->directTaps / 8) * 8;
    for (int s = 0; s < n; ++s) {
        const double* x = c->directWin + s;
        double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
        int k = 0;
        for (; k < v8; k += 8)
            for (int j = 0; j < 4; ++j) {
                a0[j] = fma(c->directIRRev[k + j], x[k + j], a0[j]);
                a1[j] = fma(c->directIRRev[k + 4 + j], x[k + 4 + j], a1[j]);
            }
        double v[4];
        for (int j = 0; j < 4; ++j) v[j] = a0[j] + a1[j];
        double y = (v[0] + v[2]) + (v[1] + v[3]);
        for (; k < c->directTaps; ++k) y += c->directIRRev[k] * x[k];
        if (!(y - y == 0.0) || fabs(y) < 1.0e-20) y = 0.0;    /* :1219, threshold kDenormThresholdDouble */
        c->directOut[s] = y;
    }
    if (c->directHistLen > 0) memcpy(c->directHist, c->directWin + n, sizeof(double) * (size_t)c->directHistLen);
    c->directPending = n;
}

void orc_nuc_add(orc_nuc* c, const double* in, int n)
{
    if (!c || !c->ready || n <= 0) return;
    process_direct(c, in, n);
    for (int li = 0; li < c->numLayers; ++li) {
        orc_layer* l = &c->L[li];
        int consumed = 0;
        while (consumed < n) {
            const int toFill = imin(n - consumed, l->partSize - l->inputPos);
            if (in) memcpy(l->inputAcc + l->inputPos, in + consumed, sizeof(double) * (size_t)toFill);
            else memset(l->inputAcc + l->inputPos, 0, sizeof(double) * (size_t)toFill);
            l->inputPos += toFill;
            consumed += toFill;
            if (l->inputPos >= l->partSize) {
                l->inputPos = 0;
                if (l->isImmediate) process_layer_block(c, l);
                else {
                    layer_push_fdl(l);
                    l->fdlIndex = (l->fdlIndex + 1) & l->fdlMask;
                    l->baseFdlIdxSaved = (l->fdlIndex - 1 + l->numParts) & l->fdlMask;
                    memset(l->accRe, 0, sizeof(double) * (size_t)l->complexSize);
                    memset(l->accIm, 0, sizeof(double) * (size_t)l->complexSize);
                    l->nextPart = 0;
                    l->distributing = 1;
                }
            }
        }
        /* distributed MAC, once per Add call: :1497-1545 */
        if (!l->isImmediate && l->distributing) {
            const size_t cs = (size_t)l->complexSize;
            const int endPart = imin(l->nextPart + l->partsPerCallback, l->numPartsIR);
            const int linStart = l->baseFdlIdxSaved - l->numPartsIR + 1 + l->numParts;
            for (int p = l->nextPart; p < endPart; ++p) {
                const size_t idx = (size_t)(linStart + p);
                mac_split(l->fdlRe + idx * cs, l->fdlIm + idx * cs,
                          l->irRe + (size_t)p * cs, l->irIm + (size_t)p * cs, l->accRe, l->accIm, l->complexSize);
            }
            l->nextPart = endPart;
            if (l->nextPart >= l->numPartsIR) {
                layer_inverse(l);
                memcpy(l->tailOut, l->outBuf + l->partSize, sizeof(double) * (size_t)l->partSize);
                if (l->delayBuf) delay_write(l, l->tailOut, l->partSize);
                l->distributing = 0;
                l->nextPart = 0;
            }
        }
    }
}

int orc_nuc_get(orc_nuc* c, double* out, int n)
{
    if (!c || !c->ready || n <= 0) {
        if (out && n > 0) memset(out, 0, sizeof(double) * (size_t)n);
        return 0;
    }
    const int got = ring_read(c, out, n);
    if (c->directEnabled && c->directOut) {
        const int toAdd = imin(n, c->directPending);
        if (toAdd > 0) {
            if (out) for (int i = 0; i < toAdd; ++i) out[i] += c->directOut[i];
            memset(c->directOut, 0, sizeof(double) * (size_t)toAdd);
            c->directPending = 0;
        }
    }
    for (int li = 1; li < c->numLayers; ++li) {
        orc_layer* l = &c->L[li];
        if (!l->delayBuf) continue;
        if (out) delay_read_add(l, out, n, c->tailEnabled ? c->layerGain[iclamp(0, 2, li)] : 0.0);
    }
    return got;
}

void orc_nuc_reset(orc_nuc* c)
{
    if (!c) return;
    for (int li = 0; li < c->numLayers; ++li) {
        orc_layer* l = &c->L[li];
        const size_t cs = (size_t)l->complexSize;
        memset(l->fdlRe, 0, sizeof(double) * (size_t)l->numParts * 2 * cs);
        memset(l->fdlIm, 0, sizeof(double) * (size_t)l->numParts * 2 * cs);
        memset(l->timeBuf, 0, sizeof(double) * (size_t)l->fftSize);
        memset(l->outBuf, 0, sizeof(double) * (size_t)l->fftSize);
        memset(l->prevIn, 0, sizeof(double) * (size_t)l->partSize);
        memset(l->accRe, 0, sizeof(double) * cs);
        memset(l->accIm, 0, sizeof(double) * cs);
        memset(l->inputAcc, 0, sizeof(double) * (size_t)l->partSize);
        if (l->tailOut) memset(l->tailOut, 0, sizeof(double) * (size_t)l->partSize);
        l->fdlIndex = l->inputPos = l->nextPart = l->baseFdlIdxSaved = l->distributing = 0;
        l->delayW = l->delayR = 0;
        if (l->delayBuf) memset(l->delayBuf, 0, sizeof(double) * (size_t)l->delayCap);
    }
    if (c->ring) memset(c->ring, 0, sizeof(double) * (size_t)c->ringSize);
    c->ringW = c->ringR = c->ringAvail = 0;
    if (c->directHist && c->directHistLen > 0) memset(c->directHist, 0, sizeof(double) * (size_t)c->directHistLen);
    if (c->directOut) memset(c->directOut, 0, sizeof(double) * (size_t)c->directMaxBlock);
    c->directPending = 0;
}

void orc_nuc_run(orc_nuc* c, const double* in, double* out, int blockSize, int nBlocks)
{
    for (int b = 0; b < nBlocks; ++b) {
        orc_nuc_add(c, in ? in + (size_t)b * blockSize : NULL, blockSize);
        const int got = orc_nuc_get(c, out + (size_t)b * blockSize, blockSize);
        if (got < blockSize && got > 0) { /* ring_read already zero-filled the remainder */ }
    }
}

void orc_direct_conv_at(const double* x, int64_t nx, const double* h, int nh,
                        const int64_t* idx, int nidx, double* y)
{
    for (int q = 0; q < nidx; ++q) {
        const int64_t n = idx[q];
        long double acc = 0.0L;
        for (int j = 0; j < nh; ++j) {
            const int64_t m = n - j;
            if (m < 0) break;
            if (m < nx) acc += (long double)h[j] * (long double)x[m];
        }
        y[q] = (double)acc;
    }
}

/* ================================================================ EQ ===== */

void orc_eq_params_default(orc_eq_params* p)
{
    static const float f[20] = { 20.0f, 32.0f, 50.0f, 80.0f, 125.0f, 200.0f, 315.0f, 500.0f, 800.0f, 1250.0f,
                                 2000.0f, 3150.0f, 5000.0f, 8000.0f, 12500.0f, 16000.0f, 19000.0f, 20000.0f,
                                 22000.0f, 24000.0f };
    memset(p, 0, sizeof(*p));
    for (int i = 0; i < 20; ++i) {
        p->bands[i].frequency = f[i];
        p->bands[i].gain = 0.0f;
        p->bands[i].q = 0.707f;
        p->bands[i].enabled = 1;
        p->bands[i].type = 1;
        p->bands[i].channelMode = 0;
    }
    p->totalGainDb = 0.0f;
    p->agcEnabled = 0;
    p->nonlinearSaturation = 0.2f;
    p->filterStructure = 0;
}

static void svf_bypass(orc_svf_coeffs* c)
{
    c->a1 = 1.0; c->a2 = 0.0; c->a3 = 0.0; c->m0 = 1.0; c->m1 = 0.0; c->m2 = 0.0;
}

void orc_svf_design(int type, float freq, float gainDb, float q, double sr, orc_svf_coeffs* c)
{
    memset(c, 0, sizeof(*c));
    c->m0 = 1.0;
    if (sr <= 0.0) { svf_bypass(c); return; }
    /* validateAndClampParameters, Coefficients.cpp:84-96 (float arithmetic) */
    const float nyquist = (float)(sr * 0.5);
    const float maxFreq = fminf(20000.0f, nyquist * 0.95f);
    freq = freq < 20.0f ? 20.0f : (freq > maxFreq ? maxFreq : freq);
    q = q < 0.01f ? 0.01f : (q > 20.0f ? 20.0f : q);
    gainDb = gainDb < -48.0f ? -48.0f : (gainDb > 48.0f ? 48.0f : gainDb);
    const double f = (double)freq, gdb = (double)gainDb, Q = (double)q;

    double A = 1.0, g, k;
    switch (type) {
        case 0: A = pow(10.0, gdb / 40.0); g = tan(M_PI * f / sr) / sqrt(A); k = 1.0 / Q; break;
        case 1: A = pow(10.0, gdb / 40.0); g = tan(M_PI * f / sr);           k = 1.0 / (Q * A); break;
        case 2: A = pow(10.0, gdb / 40.0); g = tan(M_PI * f / sr) * sqrt(A); k = 1.0 / Q; break;
        case 3: g = tan(M_PI * f / sr); k = 1.0 / Q; break;
        case 4: g = tan(M_PI * f / sr); k = 1.0 / Q; break;
        default: memset(c, 0, sizeof(*c)); c->m0 = 1.0; return;     /* `return {}` */
    }
    if (!isfinite(g) || !isfinite(k)) { svf_bypass(c); return; }
    const double den = 1.0 + g * (g + k);
    if (fabs(den) < 1.0e-15) { svf_bypass(c); return; }
    c->a1 = 1.0 / den;
    c->a2 = g * c->a1;
    c->a3 = g * c->a2;
    switch (type) {
        case 0: c->m0 = 1.0;   c->m1 = k * (A - 1.0);       c->m2 = A * A - 1.0; break;
        case 1: c->m0 = 1.0;   c->m1 = (A - 1.0 / A) / Q;   c->m2 = 0.0; break;
        case 2: c->m0 = A * A; c->m1 = k * (1.0 - A) * A;   c->m2 = 1.0 - A * A; break;
        case 3: c->m0 = 0.0;   c->m1 = 0.0;                 c->m2 = 1.0; break;
        case 4: c->m0 = 1.0;   c->m1 = -k;                  c->m2 = -1.0; break;
    }
}

double orc_fast_tanh_scalar(double x)
{
    if (x >= 4.5) return 1.0;
    if (x <= -4.5) return -1.0;
    const double x2 = x * x;
    return x * (27.0 + x2) / (27.0 + 9.0 * x2);
}

double orc_fast_tanh_v128(double x)
{
    /* _mm_max_pd(x, lo) then _mm_min_pd(., hi): NaN in x propagates the second operand of max -> lo */
    double xc = (x > -4.5) ? x : -4.5;
    xc = (xc < 4.5) ? xc : 4.5;
    const double x2 = xc * xc;
    const double num = xc * (27.0 + x2);
    const double den = 27.0 + 9.0 * x2;
    return num / den;
}

/* sanitizeFiniteInRangeV(v, 0, 1e15), Processing.cpp:90-101 */
static double sanitize(double v)
{
    const int finite = (v - v == 0.0);
    const double a = fabs(v);
    return (finite && a >= 0.0 && a < 1.0e15) ? v : 0.0;
}

void orc_svf_band_stereo_lane(double* data, int64_t n, const orc_svf_coeffs* c,
                              double* state, double saturation)
{
    double ic1 = state[0], ic2 = state[1];
    const double a1 = c->a1, a2 = c->a2, a3 = c->a3, m0 = c->m0, m1 = c->m1, m2 = c->m2;
    for (int64_t i = 0; i < n; ++i) {
        const double v0 = data[i];
        const double v3 = v0 - ic2;
        const double v1 = fma(a1, ic1, a2 * v3);
        const double v2 = fma(a2, ic1, fma(a3, v3, ic2));
        ic1 = fma(2.0, v1, -ic1);
        ic2 = fma(2.0, v2, -ic2);
        double y = fma(m0, v0, fma(m1, v1, m2 * v2));
        if (saturation > 0.0)
            y = (y * (1.0 - saturation)) + (orc_fast_tanh_v128(y) * saturation);
        y = sanitize(y);
        ic1 = sanitize(ic1);
        ic2 = sanitize(ic2);
        /* _mm_min_pd(_mm_max_pd(y, -100), 100) */
        y = (y > -100.0) ? y : -100.0;
        y = (y < 100.0) ? y : 100.0;
        data[i] = y;
    }
    state[0] = ic1; state[1] = ic2;
}

void orc_svf_band_mono(double* data, int64_t n, const orc_svf_coeffs* c,
                       double* state, double saturation)
{
    double ic1 = state[0], ic2 = state[1];
    const double a1 = c->a1, a2 = c->a2, a3 = c->a3, m0 = c->m0, m1 = c->m1, m2 = c->m2;
    for (int64_t i = 0; i < n; ++i) {
        const double v0 = data[i];
        const double v3 = v0 - ic2;
        const double v1 = a1 * ic1 + a2 * v3;
        const double v2 = ic2 + a2 * ic1 + a3 * v3;
        ic1 = 2.0 * v1 - ic1;
        ic2 = 2.0 * v2 - ic2;
        double y = m0 * v0 + m1 * v1 + m2 * v2;
        if (saturation > 0.0)
            y = y * (1.0 - saturation) + orc_fast_tanh_scalar(y) * saturation;
        y = sanitize(y);
        y = y < -100.0 ? -100.0 : (y > 100.0 ? 100.0 : y);   /* std::clamp */
        data[i] = y;
        if (!((ic1 - ic1 == 0.0) && fabs(ic1) < 1.0e15)) ic1 = 0.0;
        if (!((ic2 - ic2 == 0.0) && fabs(ic2) < 1.0e15)) ic2 = 0.0;
    }
    state[0] = ic1; state[1] = ic2;
}

/* calculateRMS, Processing.cpp:21-52: four FMA accumulator lanes, summed left to right, scalar tail, sqrt(sum/n) */
static double rms_avx_pattern(const double* d, int n)
{
    if (!d || n <= 0) return 0.0;
    double acc[4] = { 0, 0, 0, 0 };
    int i = 0;
    const int vEnd = n / 4 * 4;
    for (; i < vEnd; i += 4)
        for (int j = 0; j < 4; ++j) acc[j] = fma(d[i + j], d[i + j], acc[j]);
    double sumSq = acc[0] + acc[1] + acc[2] + acc[3];
    for (; i < n; ++i) sumSq += d[i] * d[i];
    return sqrt(sumSq / (double)n);
}

/* applyGainRamp_AVX2, Processing.cpp:279-337: four gain lanes advanced by repeated additions of 4*inc / 16*inc */
static void gain_ramp_avx_pattern(double* data, int n, double startGain, double inc)
{
    double vg[4] = { startGain, startGain + inc, startGain + 2.0 * inc, startGain + 3.0 * inc };
    const double inc4 = 4.0 * inc, inc16 = 16.0 * inc;
    int i = 0;
    const int vEnd16 = n / 16 * 16, vEnd4 = n / 4 * 4;
    for (; i < vEnd16; i += 16) {
        double g[4];
        for (int j = 0; j < 4; ++j) g[j] = vg[j];
        for (int q = 0; q < 4; ++q) {
            for (int j = 0; j < 4; ++j) data[i + 4 * q + j] *= g[j];
            if (q < 3) for (int j = 0; j < 4; ++j) g[j] = g[j] + inc4;
        }
        for (int j = 0; j < 4; ++j) vg[j] = vg[j] + inc16;
    }
    for (; i < vEnd4; i += 4) {
        for (int j = 0; j < 4; ++j) data[i + j] *= vg[j];
        for (int j = 0; j < 4; ++j) vg[j] = vg[j] + inc4;
    }
    double gain = startGain + (double)i * inc;
    for (; i < n; ++i) { data[i] *= gain; gain += inc; }
}

void orc_eq_process_stereo(double* dataL, double* dataR, int64_t n, int blockSize,
                           const orc_eq_params* p, double sr, double* state)
{
    orc_eq_process_stereo_ex(dataL, dataR, n, blockSize, p, sr, state, 0);
}

void orc_eq_process_stereo_ex(double* dataL, double* dataR, int64_t n, int blockSize,
                              const orc_eq_params* p, double sr, double* state, int forceBasicPath)
{
    orc_svf_coeffs co[20];
    int active[20];
    for (int b = 0; b < 20; ++b) {   /* createCoeffCache, ProcessingCache.cpp:71-90 */
        active[b] = p->bands[b].enabled && sr > 0.0;
        if (active[b]) orc_svf_design(p->bands[b].type, p->bands[b].frequency, p->bands[b].gain, p->bands[b].q, sr, &co[b]);
    }
    /* an active Mid/Side band sends the whole call to the basic process(block) (Processing.cpp:1036-1044), whose
     * band nodes are inactive for non-LP/HP bands within 0.01 dB of flat (createBandNode, Coefficients.cpp:48-53) */
    int basicPath = forceBasicPath != 0;
    for (int b = 0; b < 20; ++b) if (active[b] && p->bands[b].channelMode >= 3) basicPath = 1;
    if (basicPath)
        for (int b = 0; b < 20; ++b)
            if (active[b] && p->bands[b].type != 3 && p->bands[b].type != 4 && fabsf(p->bands[b].gain) < 0.01f) active[b] = 0;
    /* Mid / Side band states (states[2], states[3] of filterState[4][20][2]) live at state[88..127] / [128..167] */
    double* stMid = state + 88;
    double* stSide = state + 128;
    const double sat = (double)p->nonlinearSaturation;
    /* juce::Decibels::decibelsToGain<double>: > -100 dB ? 10^(dB/20) : 0 (EQProcessor.h:450) */
    const double gdb = (double)p->totalGainDb;
    const double gain = gdb > -100.0 ? pow(10.0, gdb * 0.05) : 0.0;
    /* AGC state lives behind the filter states: state[80..82] = envIn, envOut, currentGain-1 (stored as gain-1 so a
     * zeroed state means unity gain, rtAgcCurrentGainShadow's initial value) */
    for (int64_t off = 0; off < n; off += blockSize) {
        const int64_t len = (n - off < blockSize) ? (n - off) : blockSize;
        double inputRMS = 0.0;
        if (p->agcEnabled) {      /* Processing.cpp:1116-1127 */
            const double r0 = rms_avx_pattern(dataL + off, (int)len), r1 = rms_avx_pattern(dataR + off, (int)len);
            if (r0 > inputRMS) inputRMS = r0;
            if (r1 > inputRMS) inputRMS = r1;
        }
        if (p->filterStructure == 1) {
            /* FilterStructure::Parallel, Processing.cpp:1164-1226: out = src + sum over bands of (band(src) - src),
             * accumulated in band order as accum += work; accum -= src */
            double* srcL = alloc_d((size_t)len); double* srcR = alloc_d((size_t)len);
            double* accL = alloc_d((size_t)len); double* accR = alloc_d((size_t)len);
            double* wrk = alloc_d((size_t)len);
            memcpy(srcL, dataL + off, sizeof(double) * (size_t)len);
            memcpy(srcR, dataR + off, sizeof(double) * (size_t)len);
            for (int b = 0; b < 20; ++b) {
                if (!active[b]) continue;
                const int mode = p->bands[b].channelMode;
                if (mode >= 3) {       /* basic path, Processing.cpp:792-836 */
                    double* ms = alloc_d((size_t)len * 2);
                    for (int64_t i = 0; i < len; ++i) { ms[i] = (srcL[i] + srcR[i]) * 0.5; ms[len + i] = (srcL[i] - srcR[i]) * 0.5; }
                    if (mode == 3) orc_svf_band_mono(ms, len, &co[b], stMid + b * 2, sat);
                    else           orc_svf_band_mono(ms + len, len, &co[b], stSide + b * 2, sat);
                    for (int64_t i = 0; i < len; ++i) {
                        const double wl = ms[i] + ms[len + i], wr = ms[i] - ms[len + i];
                        accL[i] += wl - srcL[i];
                        accR[i] += wr - srcR[i];
                    }
                    free(ms);
                    continue;
                }
                for (int ch = 0; ch < 2; ++ch) {
                    if (!(mode == 0 || mode == 1 + ch)) continue;
                    const double* src = ch ? srcR : srcL;
                    double* acc = ch ? accR : accL;
                    memcpy(wrk, src, sizeof(double) * (size_t)len);
                    if (mode == 0) orc_svf_band_stereo_lane(wrk, len, &co[b], state + (ch * 20 + b) * 2, sat);
                    else           orc_svf_band_mono(wrk, len, &co[b], state + (ch * 20 + b) * 2, sat);
                    for (int64_t i = 0; i < len; ++i) { acc[i] = acc[i] + wrk[i]; acc[i] = acc[i] - src[i]; }
                }
            }
            for (int64_t i = 0; i < len; ++i) {
                dataL[off + i] = (srcL[i] + accL[i]);
                dataR[off + i] = (srcR[i] + accR[i]);
            }
            free(srcL); free(srcR); free(accL); free(accR); free(wrk);
        } else {
        for (int b = 0; b < 20; ++b) {
            if (!active[b]) continue;
            const int mode = p->bands[b].channelMode;
            if (mode == 0) {          /* processBandStereo packs L and R in one register: same arithmetic per lane */
                orc_svf_band_stereo_lane(dataL + off, len, &co[b], state + (0 * 20 + b) * 2, sat);
                orc_svf_band_stereo_lane(dataR + off, len, &co[b], state + (1 * 20 + b) * 2, sat);
            } else if (mode == 1) {
                orc_svf_band_mono(dataL + off, len, &co[b], state + (0 * 20 + b) * 2, sat);
            } else if (mode == 2) {
                orc_svf_band_mono(dataR + off, len, &co[b], state + (1 * 20 + b) * 2, sat);
            } else {                  /* Mid / Side: encode, filter one component, decode (Processing.cpp:690-739) */
                double* L = dataL + off; double* R = dataR + off;
                double* ms = alloc_d((size_t)len * 2);
                for (int64_t i = 0; i < len; ++i) { ms[i] = (L[i] + R[i]) * 0.5; ms[len + i] = (L[i] - R[i]) * 0.5; }
                if (mode == 3) orc_svf_band_mono(ms, len, &co[b], stMid + b * 2, sat);
                else           orc_svf_band_mono(ms + len, len, &co[b], stSide + b * 2, sat);
                for (int64_t i = 0; i < len; ++i) { L[i] = ms[i] + ms[len + i]; R[i] = ms[i] - ms[len + i]; }
                free(ms);
            }
        }
        }
        if (p->agcEnabled) {
            /* processAGC, Processing.cpp:367-445; block coefficients from the tables of prepareToPlay
             * (EQProcessor.Core.cpp:776-784): 1 - exp(-n / (sr * tau)), tau = 0.2 / 2.0 / 0.2 s */
            const double nn = (double)len;
            const double bAtt = 1.0 - exp(-nn / (sr * 0.2)), bRel = 1.0 - exp(-nn / (sr * 2.0)), bSm = 1.0 - exp(-nn / (sr * 0.2));
            double outputRMS = 0.0;
            const double r0 = rms_avx_pattern(dataL + off, (int)len), r1 = rms_avx_pattern(dataR + off, (int)len);
            if (r0 > outputRMS) outputRMS = r0;
            if (r1 > outputRMS) outputRMS = r1;
            if (!(inputRMS - inputRMS == 0.0) || inputRMS > 1000.0) inputRMS = 1000.0;
            if (!(outputRMS - outputRMS == 0.0) || outputRMS > 1000.0) outputRMS = 1000.0;
            double envIn = state[80], envOut = state[81], cur = state[82] + 1.0;
            const double inA = (inputRMS > envIn) ? bAtt : bRel, outA = (outputRMS > envOut) ? bAtt : bRel;
            envIn = envIn * (1.0 - inA) + inputRMS * inA;
            envOut = envOut * (1.0 - outA) + outputRMS * outA;
            if (envIn < 1.0e-20) envIn = 0.0;
            if (envOut < 1.0e-20) envOut = 0.0;
            double target = 1.0;       /* calculateAGCGain, :343-358 */
            if (!(envOut < 1e-6)) {
                const double ratio = envIn / envOut;
                if (!(ratio > 1.0 / 1.059 && ratio < 1.059))
                    target = ratio < (double)0.06f ? (double)0.06f : (ratio > (double)16.0f ? (double)16.0f : ratio);
            }
            const double next = cur * (1.0 - bSm) + target * bSm;
            state[80] = envIn; state[81] = envOut; state[82] = next - 1.0;
            const double incr = (next - cur) / nn;
            gain_ramp_avx_pattern(dataL + off, (int)len, cur, incr);
            gain_ramp_avx_pattern(dataR + off, (int)len, cur, incr);
        } else {
            /* total gain through smoothTotalGain (LinearRamp, src/DspNumericPolicy.h:319-421, 50 ms) and
             * applyGainRamp_AVX2 (Processing.cpp:1262-1274).  state[83] = initialised flag (prepareToPlay does
             * setCurrentAndTargetValue, Core.cpp:765), [84] current, [85] target, [86] step, [87] remaining */
            if (state[83] == 0.0) { state[83] = 1.0; state[84] = gain; state[85] = gain; state[86] = 0.0; state[87] = 0.0; }
            if (fabs(state[85] - gain) > 1e-6 && gain != state[85]) {          /* setTargetValue */
                int total = (int)(sr * 0.05 + 0.5);
                if (total <= 0) total = 1;
                const int steps = state[87] > 0.0 ? (int)state[87] : total;
                state[85] = gain;
                state[86] = (state[85] - state[84]) / (double)steps;
                state[87] = (double)steps;
            }
            const double startG = state[84];
            if (state[87] > 0.0) {                                               /* skip(numSamples) */
                if ((double)len >= state[87]) { state[84] = state[85]; state[87] = 0.0; }
                else { state[84] += state[86] * (double)len; state[87] -= (double)len; }
            }
            const double incr = (state[84] - startG) / (double)len;
            gain_ramp_avx_pattern(dataL + off, (int)len, startG, incr);
            gain_ramp_avx_pattern(dataR + off, (int)len, startG, incr);
        }
    }
}

/* ====================================================== OutputFilter ===== */

static orc_biquad bq_identity(void) { orc_biquad c = { 1.0, 0.0, 0.0, 0.0, 0.0 }; return c; }

static orc_biquad bq_lpf(double fc, double Q, double fs)
{
    const double nyq = fs * 0.4999;
    if (fc >= nyq || Q <= 0.0 || fs <= 0.0) return bq_identity();
    const double w0 = 2.0 * M_PI * fc / fs;
    const double sn = sin(w0), cs = cos(w0);
    const double alpha = sn / (2.0 * Q);
    const double a0inv = 1.0 / (1.0 + alpha);
    orc_biquad c;
    c.b0 = (1.0 - cs) * 0.5 * a0inv;
    c.b1 = (1.0 - cs) * a0inv;
    c.b2 = (1.0 - cs) * 0.5 * a0inv;
    c.a1 = (-2.0 * cs) * a0inv;
    c.a2 = (1.0 - alpha) * a0inv;
    return c;
}

static orc_biquad bq_hpf(double fc, double Q, double fs)
{
    const double nyq = fs * 0.4999;
    if (fc <= 0.0 || fc >= nyq || Q <= 0.0 || fs <= 0.0) return bq_identity();
    const double w0 = 2.0 * M_PI * fc / fs;
    const double sn = sin(w0), cs = cos(w0);
    const double alpha = sn / (2.0 * Q);
    const double a0inv = 1.0 / (1.0 + alpha);
    orc_biquad c;
    c.b0 = (1.0 + cs) * 0.5 * a0inv;
    c.b1 = -(1.0 + cs) * a0inv;
    c.b2 = (1.0 + cs) * 0.5 * a0inv;
    c.a1 = (-2.0 * cs) * a0inv;
    c.a2 = (1.0 - alpha) * a0inv;
    return c;
}

void orc_outfilter_design(int convIsLast, int hcMode, int lcMode, int lpMode, double fs, orc_biquad out[3])
{
    const double fc_hc = (fs <= 48000.0) ? 19000.0 : 22000.0;
    const double fc_lp = (fs <= 48000.0) ? 19000.0 : 24000.0;
    if (convIsLast) {
        out[0] = (lcMode == 1) ? bq_hpf(15.0, 0.5, fs) : bq_hpf(18.0, 0.70711, fs);
        if (hcMode == 0)      { out[1] = bq_lpf(fc_hc, 0.54120, fs); out[2] = bq_lpf(fc_hc, 1.30656, fs); }
        else if (hcMode == 2) { out[1] = bq_lpf(fc_hc, 0.5, fs);     out[2] = bq_identity(); }
        else                  { out[1] = bq_lpf(fc_hc, 0.70711, fs); out[2] = bq_lpf(fc_hc, 0.70711, fs); }
    } else {
        out[0] = bq_hpf(20.0, 0.70711, fs);
        const double q = (lpMode == 0) ? 1.0 : (lpMode == 2 ? 0.5 : 0.70711);
        out[1] = bq_lpf(fc_lp, q, fs);
        out[2] = bq_lpf(fc_lp, q, fs);
    }
}

static double bq_step(double x, const orc_biquad* c, double* w1, double* w2)
{
    const double y = fma(c->b0, x, *w1);
    const double n1 = fma(c->b1, x, fma(-c->a1, y, *w2));
    const double n2 = fma(-c->a2, y, c->b2 * x);
    *w1 = (fabs(n1) < 1.0e-20) ? 0.0 : n1;      /* _mm_andnot_pd(_mm_cmplt_pd(|w|, 1e-20), w) */
    *w2 = (fabs(n2) < 1.0e-20) ? 0.0 : n2;
    return y;
}

void orc_biquad_df2t_lane(double* data, int64_t n, const orc_biquad* c, double* state)
{
    double w1 = state[0], w2 = state[1];
    for (int64_t i = 0; i < n; ++i) data[i] = bq_step(data[i], c, &w1, &w2);
    state[0] = w1; state[1] = w2;
}

void orc_outfilter_process_stereo(double* dataL, double* dataR, int64_t n, const orc_biquad c[3], double* state)
{
    double* ch[2] = { dataL, dataR };
    for (int k = 0; k < 2; ++k) {
        double* st = state + k * 6;
        for (int64_t i = 0; i < n; ++i) {
            double x = ch[k][i];
            x = bq_step(x, &c[0], &st[0], &st[1]);
            x = bq_step(x, &c[1], &st[2], &st[3]);
            x = bq_step(x, &c[2], &st[4], &st[5]);
            ch[k][i] = x;
        }
    }
}

double orc_equal_power_sin(double x)
{
    const double t = x * (M_PI * 0.5);
    const double t2 = t * t;
    return t * (1.0 + t2 * (-1.0 / 6.0 + t2 * (1.0 / 120.0 + t2 * (-1.0 / 5040.0 + t2 * (1.0 / 362880.0)))));
}
