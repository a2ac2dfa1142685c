/* TEST INFRASTRUCTURE ONLY -- see cvtt_oracle.h.
 *
 * Plain-C, lane-by-lane restatement of the reference's SSE2 lane arithmetic
 * (SURVEY.md App. A): every lane of the reference is independent except for the
 * group-wide booleans computed with AnySet/AllSet, which are modelled explicitly.
 * All float arithmetic is IEEE binary32, one rounding per operation, no contraction
 * (compile with -ffp-contract=off -mfpmath=sse).
 */
#include "cvtt_oracle.h"
#include "cvtt_oracle_tables.h"
#include "cvtt_oracle_bc7sc.h"
#include "cvtt_oracle_s3tcsc.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <xmmintrin.h>
#include <emmintrin.h>

/* ------------------------------------------------------------------------------------
 * Lane arithmetic helpers (ConvectionKernels_ParallelMath.h, SSE2 branch)
 * ---------------------------------------------------------------------------------- */

/* MINPS / MAXPS operand semantics, ParallelMath.h:522-559 */
static inline float f_min(float a, float b) { return a < b ? a : b; }
static inline float f_max(float a, float b) { return a > b ? a : b; }
/* ParallelMath.h:561-567 */
static inline float f_clamp(float v, float lo, float hi) { return f_max(f_min(v, hi), lo); }
/* ParallelMath.h:472-475 */
static inline float f_safe_denom(float v) { return v == 0.0f ? 1.0f : v; }

/* CVTPS2DQ (round-half-even in the RN scope) + signed saturating pack to 16 bit,
 * ParallelMath.h:936-946 */
static inline int cvt_rne_s16(float v)
{
    int i = _mm_cvtss_si32(_mm_set_ss(v));
    if (i > 32767) i = 32767;
    if (i < -32768) i = -32768;
    return i;
}

size_t orc_sizeof_options(void) { return sizeof(orc_options); }
size_t orc_sizeof_bc7_plan(void) { return sizeof(orc_bc7_plan); }

void orc_probe_rcp(float out17[17])
{
    for (int i = 0; i <= 16; i++)
    {
        float v = (float)(i == 0 ? 1 : i);
        out17[i] = _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(v)));
    }
}

/* Util::FillWeights, ConvectionKernels_Util.cpp:62-73 */
static void fill_weights(const orc_options *o, float w[4])
{
    if (o->flags & ORC_FLAG_UNIFORM)
        w[0] = w[1] = w[2] = w[3] = 1.0f;
    else
    {
        w[0] = o->redWeight;
        w[1] = o->greenWeight;
        w[2] = o->blueWeight;
        w[3] = o->alphaWeight;
    }
}

/* Util::ComputeTweakFactors, ConvectionKernels_Util.cpp:75-84 */
static void tweak_factors(int tweak, int range, float tf[2])
{
    int totalUnits = range - 1;
    int minOutside = (tweak >> 1) & 1;
    int maxOutside = tweak & 1;
    int inside = totalUnits - minOutside - maxOutside;
    tf[0] = -(float)minOutside / (float)inside;
    tf[1] = (float)maxOutside / (float)inside + 1.0f;
}

/* ------------------------------------------------------------------------------------
 * EndpointSelector<N,8> + PackedCovarianceMatrix<N> + UnfinishedEndpoints<N>
 * (ConvectionKernels_EndpointSelector.h:33-149, _PackedCovarianceMatrix.h:29-59,
 *  _UnfinishedEndpoints.h:77-114)
 * ---------------------------------------------------------------------------------- */
typedef struct
{
    float base[4];
    float offset[4];
} unfinished_t;

/* values: pre-weighted pixels [*][stride>=n]; list of `len` pixel indices `frag`
 * (NULL = 0..len-1). */
static void pca_endpoints(const float *values, int stride, int n, const uint8_t *frag, int len,
                          const float *channelWeights, unfinished_t *out)
{
    float centroid[4] = {0, 0, 0, 0}, direction[4] = {0, 0, 0, 0};
    float cov[10];
    float weightTotal = 0.0f;
    float minDist = FLT_MAX, maxDist = -FLT_MAX;
    for (int i = 0; i < 10; i++) cov[i] = 0.0f;

    /* pass 0: centroid (EndpointSelector.h:73-87) */
    for (int i = 0; i < len; i++)
    {
        const float *v = values + (frag ? frag[i] : i) * stride;
        for (int ch = 0; ch < n; ch++)
            centroid[ch] = centroid[ch] + v[ch] * 1.0f;
        weightTotal = weightTotal + 1.0f;
    }
    {
        float denom = f_safe_denom(weightTotal);
        for (int ch = 0; ch < n; ch++)
            centroid[ch] = centroid[ch] / denom;
    }

    /* pass 1: covariance (EndpointSelector.h:89-96, PackedCovarianceMatrix.h:29-41) */
    for (int i = 0; i < len; i++)
    {
        const float *v = values + (frag ? frag[i] : i) * stride;
        float diff[4];
        for (int ch = 0; ch < n; ch++)
            diff[ch] = v[ch] - centroid[ch];
        int index = 0;
        for (int row = 0; row < n; row++)
            for (int col = 0; col <= row; col++)
            {
                cov[index] = cov[index] + diff[row] * diff[col] * 1.0f;
                index++;
            }
    }
    /* power iteration (EndpointSelector.h:98-130) */
    {
        float approx[4];
        for (int ch = 0; ch < n; ch++) approx[ch] = 1.0f;
        for (int it = 0; it < 8; it++)
        {
            float product[4];
            for (int row = 0; row < n; row++)
            {
                float sum = 0.0f;
                int index = (row * (row + 1)) >> 1;
                for (int col = 0; col < n; col++)
                {
                    sum = sum + approx[col] * cov[index];
                    if (col >= row)
                        index += col + 1;
                    else
                        index++;
                }
                product[row] = sum;
            }
            float largest = product[0];
            for (int ch = 1; ch < n; ch++)
                largest = f_max(largest, product[ch]);
            largest = f_safe_denom(largest);
            for (int ch = 0; ch < n; ch++)
                approx[ch] = product[ch] / largest;
        }
        float approxLen = 0.0f;
        for (int ch = 0; ch < n; ch++)
            approxLen = approxLen + approx[ch] * approx[ch];
        approxLen = sqrtf(approxLen);
        approxLen = f_safe_denom(approxLen);
        for (int ch = 0; ch < n; ch++)
            direction[ch] = approx[ch] / approxLen;
    }
    /* pass 2: extent along the axis (EndpointSelector.h:132-140) */
    for (int i = 0; i < len; i++)
    {
        const float *v = values + (frag ? frag[i] : i) * stride;
        float dist = 0.0f;
        for (int ch = 0; ch < n; ch++)
            dist = dist + direction[ch] * (v[ch] - centroid[ch]);
        minDist = f_min(minDist, dist);
        maxDist = f_max(maxDist, dist);
    }
    /* GetEndpoints (EndpointSelector.h:51-70): divides by the raw weight */
    for (int ch = 0; ch < n; ch++)
    {
        float mn = centroid[ch] + direction[ch] * minDist;
        float mx = centroid[ch] + direction[ch] * maxDist;
        out->base[ch] = mn / channelWeights[ch];
        out->offset[ch] = (mx - mn) / channelWeights[ch];
    }
}

/* UnfinishedEndpoints::FinishLDR, UnfinishedEndpoints.h:77-91 */
static void finish_ldr(const unfinished_t *u, int n, int tweak, int range, int ep0[4], int ep1[4])
{
    float tf[2];
    tweak_factors(tweak, range, tf);
    for (int ch = 0; ch < n; ch++)
    {
        float e0 = f_clamp(u->base[ch] + u->offset[ch] * tf[0], 0.0f, 255.0f);
        float e1 = f_clamp(u->base[ch] + u->offset[ch] * tf[1], 0.0f, 255.0f);
        ep0[ch] = cvt_rne_s16(e0);
        ep1[ch] = cvt_rne_s16(e1);
    }
}

/* ------------------------------------------------------------------------------------
 * IndexSelector<N> (ConvectionKernels_IndexSelector.h:27-131)
 * ---------------------------------------------------------------------------------- */
typedef struct
{
    int n;
    int ep[2][4];
    float origin[4];
    float axis[4];
    int range;
    float maxValue;
} selector_t;

static void selector_init(selector_t *s, int n, const float *channelWeights, int ep[2][4], int range)
{
    float epDiffWeighted[4];
    s->n = n;
    s->range = range;
    s->maxValue = (float)(range - 1);
    for (int ch = 0; ch < n; ch++)
    {
        s->ep[0][ch] = ep[0][ch];
        s->ep[1][ch] = ep[1][ch];
        s->origin[ch] = (float)ep[0][ch];
        float opposing = (float)ep[1][ch];
        epDiffWeighted[ch] = (opposing - s->origin[ch]) * channelWeights[ch];
    }
    float lenSquared = epDiffWeighted[0] * epDiffWeighted[0];
    for (int ch = 1; ch < n; ch++)
        lenSquared = lenSquared + epDiffWeighted[ch] * epDiffWeighted[ch];
    lenSquared = f_safe_denom(lenSquared);
    float mvdls = s->maxValue / lenSquared;
    for (int ch = 0; ch < n; ch++)
        s->axis[ch] = epDiffWeighted[ch] * channelWeights[ch] * mvdls;
}

static int selector_select(const selector_t *s, const float *pixel)
{
    float dist = (pixel[0] - s->origin[0]) * s->axis[0];
    for (int ch = 1; ch < s->n; ch++)
        dist = dist + (pixel[ch] - s->origin[ch]) * s->axis[ch];
    return cvt_rne_s16(f_clamp(dist, 0.0f, s->maxValue));
}

/* g_weightReciprocals[range] = round(32768/(range-1)), IndexSelector.cpp:43-62 */
static inline unsigned weight_reciprocal(int range)
{
    return (unsigned)((65536 + (range - 1)) / (2 * (range - 1)));
}

/* ReconstructLDR_BC7, IndexSelector.h:90-100 (16-bit wrapping arithmetic) */
static void selector_reconstruct_bc7(const selector_t *s, int index, int *pixel, int numRealChannels)
{
    uint16_t weight = (uint16_t)((uint16_t)((uint16_t)(weight_reciprocal(s->range) * (unsigned)index) + 256) >> 9);
    for (int ch = 0; ch < numRealChannels; ch++)
    {
        uint16_t e0 = (uint16_t)((uint16_t)(64 - weight) * (uint16_t)s->ep[0][ch]);
        uint16_t e1 = (uint16_t)(weight * (uint16_t)s->ep[1][ch]);
        pixel[ch] = (uint16_t)((uint16_t)(e0 + e1 + 32) >> 6);
    }
}

/* ReconstructLDRPrecise, IndexSelector.h:102-112 */
static void selector_reconstruct_precise(const selector_t *s, int index, int *pixel, int numRealChannels)
{
    uint16_t weight = (uint16_t)((uint16_t)((uint16_t)(weight_reciprocal(s->range) * (unsigned)index) + 64) >> 7);
    for (int ch = 0; ch < numRealChannels; ch++)
    {
        uint16_t e0 = (uint16_t)((uint16_t)(256 - weight) * (uint16_t)s->ep[0][ch]);
        uint16_t e1 = (uint16_t)(weight * (uint16_t)s->ep[1][ch]);
        pixel[ch] = (uint16_t)((uint16_t)(e0 + e1 + 128) >> 8);
    }
}

/* ------------------------------------------------------------------------------------
 * AggregatedError<N> + BCCommon::ComputeErrorLDR (AggregatedError.h:19-46, BCCommon.h:24-43)
 * ---------------------------------------------------------------------------------- */
typedef struct
{
    uint32_t err[4];
} aggerr_t;

static inline void agg_init(aggerr_t *a) { a->err[0] = a->err[1] = a->err[2] = a->err[3] = 0; }

static inline void agg_add_pixel(aggerr_t *a, const int *reconstructed, const int *original, int numRealChannels)
{
    for (int ch = 0; ch < numRealChannels; ch++)
    {
        uint16_t d = (uint16_t)(reconstructed[ch] - original[ch]);
        a->err[ch] += (uint16_t)(d * d); /* SqDiffUInt8, ParallelMath.h:987-994 */
    }
}

static float agg_finalize(const aggerr_t *a, int n, uint32_t flags, const float *weightsSq)
{
    if (flags & ORC_FLAG_UNIFORM)
    {
        uint32_t total = a->err[0];
        for (int ch = 1; ch < n; ch++)
            total += a->err[ch];
        return (float)(int32_t)total;
    }
    float total = (float)(int32_t)a->err[0] * weightsSq[0];
    for (int ch = 1; ch < n; ch++)
        total = total + (float)(int32_t)a->err[ch] * weightsSq[ch];
    return total;
}

static float error_ldr_simple(uint32_t flags, int n, const int *reconstructed, const int *original,
                              int numRealChannels, const float *weightsSq)
{
    aggerr_t a;
    agg_init(&a);
    agg_add_pixel(&a, reconstructed, original, numRealChannels);
    return agg_finalize(&a, n, flags, weightsSq);
}

/* ------------------------------------------------------------------------------------
 * EndpointRefiner<N> (ConvectionKernels_EndpointRefiner.h:38-175)
 * ---------------------------------------------------------------------------------- */
typedef struct
{
    int n;
    float tv[4], v[4];
    float tt, t, w;
    int wu;
    float rcpMaxIndex;
    float rcpChannelWeights[4];
    const float *rcpLUT;
} refiner_t;

static void refiner_init(refiner_t *r, int n, int indexRange, const float *channelWeights, const float *rcpLUT)
{
    r->n = n;
    for (int ch = 0; ch < 4; ch++)
        r->tv[ch] = r->v[ch] = 0.0f;
    r->tt = r->t = r->w = 0.0f;
    r->wu = 0;
    r->rcpMaxIndex = 1.0f / (float)(indexRange - 1);
    for (int ch = 0; ch < n; ch++)
    {
        r->rcpChannelWeights[ch] = 1.0f;
        if (channelWeights[ch] != 0.0f)
            r->rcpChannelWeights[ch] = 1.0f / channelWeights[ch];
    }
    r->rcpLUT = rcpLUT;
}

static void refiner_contribute_unweighted(refiner_t *r, const float *pwPixel, int index, int numRealChannels)
{
    float t = (float)index * r->rcpMaxIndex;
    for (int ch = 0; ch < numRealChannels; ch++)
    {
        float v = pwPixel[ch];
        r->tv[ch] = r->tv[ch] + t * v;
        r->v[ch] = r->v[ch] + v;
    }
    r->tt = r->tt + t * t;
    r->t = r->t + t;
    r->wu++;
}

/* RCPPS: on every path that reaches here the argument is an integer 1..16
 * (count of contributed pixels), so the host's instruction is a 16-entry table. */
static float refiner_rcp(const refiner_t *r, float w)
{
    int iw = (int)w;
    if ((float)iw == w && iw >= 1 && iw <= 16)
        return r->rcpLUT[iw];
    return _mm_cvtss_f32(_mm_rcp_ps(_mm_set1_ps(w)));
}

static void refiner_solve(const refiner_t *r, float endPoint[2][4])
{
    float w = r->w + (float)r->wu;
    w = f_safe_denom(w);
    float wRcp = refiner_rcp(r, w);
    float adenom = (r->tt * w - r->t * r->t) * wRcp;
    int adenomZero = (adenom == 0.0f);
    if (adenomZero)
        adenom = 1.0f;
    for (int ch = 0; ch < r->n; ch++)
    {
        float a = (r->tv[ch] - r->t * r->v[ch] * wRcp) / adenom;
        float b = (r->v[ch] - a * r->t) * wRcp;
        float p1 = b;
        float p2 = a + b;
        if (adenomZero)
        {
            p1 = r->v[ch] * wRcp;
            p2 = p1;
        }
        float inv = r->rcpChannelWeights[ch];
        endPoint[0][ch] = p1 * inv;
        endPoint[1][ch] = p2 * inv;
    }
}

static void refiner_get_ldr(const refiner_t *r, int ep[2][4])
{
    float fe[2][4];
    refiner_solve(r, fe);
    for (int epi = 0; epi < 2; epi++)
        for (int ch = 0; ch < r->n; ch++)
            ep[epi][ch] = cvt_rne_s16(f_clamp(fe[epi][ch], 0.0f, 255.0f));
}

/* ------------------------------------------------------------------------------------
 * BC7 endpoint quantisation (ConvectionKernels_BC67.cpp:829-938), 16-bit wrapping
 * ---------------------------------------------------------------------------------- */
static void bc7_quantize(int *c, int bits, int channels)
{
    for (int ch = 0; ch < channels; ch++)
    {
        uint16_t v = (uint16_t)c[ch];
        v = (uint16_t)((uint16_t)((uint16_t)((uint16_t)(v << bits) - v) + (uint16_t)(127 + (1 << (7 - bits)))) >> 8);
        c[ch] = v;
    }
}

static void bc7_quantize_p(int *c, int bits, int p, int channels)
{
    uint16_t addend = p ? (uint16_t)((1 << (8 - bits)) - 1) : 255;
    for (int ch = 0; ch < channels; ch++)
    {
        uint16_t v = (uint16_t)c[ch];
        v = (uint16_t)((uint16_t)((uint16_t)((uint16_t)(v << (bits + 1)) - v) + addend) >> 9);
        v = (uint16_t)((uint16_t)(v << 1) | (uint16_t)p);
        c[ch] = v;
    }
}

static void bc7_unquantize(int *c, int bits, int channels)
{
    for (int ch = 0; ch < channels; ch++)
    {
        uint16_t v = (uint16_t)c[ch];
        v = (uint16_t)(v << (8 - bits));
        c[ch] = (uint16_t)(v | (v >> bits));
    }
}

static void bc7_compress_endpoints(int mode, int ep[2][4], const int p[2])
{
    for (int j = 0; j < 2; j++)
    {
        switch (mode)
        {
        case 0:
            bc7_quantize_p(ep[j], 4, p[j], 3);
            bc7_unquantize(ep[j], 5, 3);
            ep[j][3] = 255;
            break;
        case 1:
            bc7_quantize_p(ep[j], 6, p[0], 3);
            bc7_unquantize(ep[j], 7, 3);
            ep[j][3] = 255;
            break;
        case 2:
            bc7_quantize(ep[j], 5, 3);
            bc7_unquantize(ep[j], 5, 3);
            ep[j][3] = 255;
            break;
        case 3:
            bc7_quantize_p(ep[j], 7, p[j], 3);
            ep[j][3] = 255;
            break;
        case 6:
            bc7_quantize_p(ep[j], 7, p[j], 4);
            break;
        case 7:
            bc7_quantize_p(ep[j], 5, p[j], 4);
            bc7_unquantize(ep[j], 6, 4);
            break;
        default:
            break;
        }
    }
}

/* BC7 mode table (format spec; ConvectionKernels_BC67.cpp:108-119) */
typedef struct
{
    int pBitMode; /* 0 per endpoint, 1 per subset, 2 none */
    int alphaMode; /* 0 combined, 1 separate, 2 none */
    int rgbBits, alphaBits, partitionBits, numSubsets, indexBits, alphaIndexBits, hasIndexSelector;
} bc7_mode_t;

static const bc7_mode_t bc7_modes[8] = {
    {0, 2, 4, 0, 4, 3, 3, 0, 0},
    {1, 2, 6, 0, 6, 2, 3, 0, 0},
    {2, 2, 5, 0, 6, 3, 2, 0, 0},
    {0, 2, 7, 0, 6, 2, 2, 0, 0},
    {2, 1, 5, 6, 0, 1, 2, 3, 1},
    {2, 1, 7, 8, 0, 1, 2, 2, 0},
    {0, 0, 7, 7, 0, 1, 4, 0, 0},
    {0, 0, 5, 5, 6, 2, 2, 0, 0},
};

typedef struct
{
    int mode;
    float error;
    int ep[3][2][4];
    int indexes[16];
    int indexes2[16];
    int partOrIS; /* union { m_partition ; m_isr.m_indexSelector }, BC67.cpp:67-75 */
    int rotation;
} bc7_work_t;

typedef struct
{
    uint32_t flags;
    float w[4];
    float wSq[4];
    const orc_bc7_plan *plan;
    int refineRounds;
    const float *rcp;
    int anyBlockHasAlpha; /* group-wide, BC67.cpp:1069 */
    int allowRGBModes;    /* group-wide, BC67.cpp:1072 */
} bc7_ctx_t;

/* Per-lane (= per block of the group) state of TrySinglePlane.
 * SinglePlaneTemporaries, BC67.cpp:803-811 (canonical oracle: zero-initialised). */
typedef struct
{
    int pixels[16][4];
    float floatPixels[16][4];
    float preWeighted[16][4];
    int isPunchThrough, blockHasNonMaxAlpha, blockHasNonZeroAlpha;
    unfinished_t unfinishedRGB[243];
    unfinished_t unfinishedRGBA[129];
    int fragmentBestIndexes[1612];
    int shapeBestEP[243][2][4];
    float shapeBestError[243];
    /* per-shape / per-trial scratch */
    float staticAlphaError;
    int tweakBaseEP[4][2][4];
    int punchThroughInvalid[4];
    int ep[2][4];
    int indexes[16];
    float shapeError;
    refiner_t ref;
    bc7_work_t work;
} bc7_lane_t;

/* one (endpoint set -> indexes, error, refiner sums) evaluation of a lane;
 * body of the refine loop, BC67.cpp:1319-1395 */
static void bc7_single_plane_trial(const bc7_ctx_t *cx, bc7_lane_t *ln, int mode, const int p[2], const uint8_t *frag,
                                   int shapeLength, int indexPrec, int numRealChannels, int isRGB, int lastRefine)
{
    const uint32_t flags = cx->flags;
    bc7_compress_endpoints(mode, ln->ep, p);

    float shapeError = 0.0f;
    selector_t sel;
    selector_init(&sel, 4, cx->w, ln->ep, 1 << indexPrec);
    refiner_init(&ln->ref, 4, 1 << indexPrec, cx->w, cx->rcp);

    aggerr_t agg;
    agg_init(&agg);
    for (int i = 0; i < shapeLength; i++)
    {
        const int px = frag[i];
        int reconstructed[4];
        int index = selector_select(&sel, ln->floatPixels[px]);
        selector_reconstruct_bc7(&sel, index, reconstructed, numRealChannels);

        if (flags & ORC_FLAG_BC7_FAST_INDEXING)
            agg_add_pixel(&agg, reconstructed, ln->pixels[px], numRealChannels);
        else
        {
            float error = error_ldr_simple(flags, 4, reconstructed, ln->pixels[px], numRealChannels, cx->wSq);
            int alt[2];
            alt[0] = (index > 1 ? index : 1) - 1;
            alt[1] = (index + 1 < (1 << indexPrec) - 1) ? index + 1 : (1 << indexPrec) - 1;
            for (int ii = 0; ii < 2; ii++)
            {
                selector_reconstruct_bc7(&sel, alt[ii], reconstructed, numRealChannels);
                float altError = error_ldr_simple(flags, 4, reconstructed, ln->pixels[px], numRealChannels, cx->wSq);
                int better = altError < error;
                error = f_min(error, altError);
                if (better) index = alt[ii];
            }
            shapeError = shapeError + error;
        }

        if (!lastRefine)
            refiner_contribute_unweighted(&ln->ref, ln->preWeighted[px], index, numRealChannels);
        ln->indexes[i] = index;
    }

    if (flags & ORC_FLAG_BC7_FAST_INDEXING)
        shapeError = agg_finalize(&agg, 4, flags, cx->wSq);
    if (isRGB)
        shapeError = shapeError + ln->staticAlphaError;
    ln->shapeError = shapeError;
}

/* BC7_TrySingleColor: call site BC67.cpp:1436-1570 + TrySingleColorRGBAMultiTable, BC67.cpp:940-1040.
 * Restated as the reference behaves, not as it was meant:
 *   - the shape average is taken over pixels[pxi] (the FIRST shapeLength pixels of the block),
 *     BC67.cpp:1446, not over the members of the shape;
 *   - `better = AndNot(pti, better)` is `pti & ~better` (ParallelMath.h:900-905), so a table is
 *     only ever taken by lanes that are punch-through-invalid AND not better; bestAverageError
 *     therefore stays FLT_MAX and, for finite weights, no table is ever taken: the candidate
 *     that reaches the error test is endpoints (0,0,0[,255]) with index 0.
 * The 8 lanes run in lock-step (AnySet guards). */
static void bc7_try_single_color(const bc7_ctx_t *cx, bc7_lane_t *lanes, int mode, int shape, const uint8_t *frag,
                                 int shapeStart, int shapeLength, int numRealChannels)
{
    const uint32_t flags = cx->flags;
    float average[8][4], bestAverageError[8];
    int intAverage[8][4], eps[8][2][4], reconstructed[8][4], index[8];
    const float rcpShapeLength = 1.0f / (float)shapeLength;
    for (int l = 0; l < 8; l++)
    {
        for (int ch = 0; ch < 4; ch++)
        {
            uint16_t total = 0;
            for (int pxi = 0; pxi < shapeLength; pxi++)
                total = (uint16_t)(total + lanes[l].pixels[pxi][ch]);
            average[l][ch] = (float)total * rcpShapeLength;
            intAverage[l][ch] = cvt_rne_s16(average[l][ch]);
        }
        bestAverageError[l] = FLT_MAX;
        for (int epi = 0; epi < 2; epi++)
        {
            eps[l][epi][0] = eps[l][epi][1] = eps[l][epi][2] = 0;
            eps[l][epi][3] = 255;
        }
        reconstructed[l][0] = reconstructed[l][1] = reconstructed[l][2] = 0;
        reconstructed[l][3] = 255;
        index[l] = 0;
    }

    const int first = orc_bc7sc_first[mode], count = orc_bc7sc_count[mode];
    for (int t = first; t < first + count; t++)
    {
        const int tableIndex = orc_bc7sc_info[t][0], tablePBits = orc_bc7sc_info[t][1];
        int candRec[8][4], candEP[8][2][4], better[8], any = 0;
        for (int l = 0; l < 8; l++)
        {
            float avgError = 0.0f;
            for (int ch = 0; ch < numRealChannels; ch++)
            {
                const unsigned char *e = orc_bc7sc_entries[t][intAverage[l][ch] & 255];
                candEP[l][0][ch] = e[0];
                candEP[l][1][ch] = e[1];
                candRec[l][ch] = e[2];
                const float delta = (float)candRec[l][ch] - average[l][ch];
                avgError = avgError + delta * delta * cx->wSq[ch];
            }
            const int isBetter = avgError < bestAverageError[l];
            better[l] = lanes[l].punchThroughInvalid[tablePBits] && !isBetter; /* AndNot(pti, better) */
            any = any || better[l];
            if (better[l])
                bestAverageError[l] = avgError; /* set below under AnySet; same lanes */
        }
        if (!any)
            continue;
        for (int l = 0; l < 8; l++)
        {
            if (!better[l])
                continue;
            index[l] = tableIndex;
            for (int ch = 0; ch < numRealChannels; ch++)
            {
                reconstructed[l][ch] = candRec[l][ch];
                eps[l][0][ch] = candEP[l][0][ch];
                eps[l][1][ch] = candEP[l][1][ch];
            }
        }
    }

    for (int l = 0; l < 8; l++)
    {
        bc7_lane_t *ln = &lanes[l];
        aggerr_t agg;
        agg_init(&agg);
        for (int pxi = 0; pxi < shapeLength; pxi++)
            agg_add_pixel(&agg, reconstructed[l], ln->pixels[frag[pxi]], numRealChannels);
        const float error = agg_finalize(&agg, 4, flags, cx->wSq) + ln->staticAlphaError;
        if (error < ln->shapeBestError[shape])
        {
            ln->shapeBestError[shape] = error; /* Min(best, error) with error < best */
            for (int epi = 0; epi < 2; epi++)
                for (int ch = 0; ch < numRealChannels; ch++)
                    ln->shapeBestEP[shape][epi][ch] = eps[l][epi][ch];
            for (int pxi = 0; pxi < shapeLength; pxi++)
                ln->fragmentBestIndexes[shapeStart + pxi] = index[l];
        }
    }
}

/* BC7Computer::TrySinglePlane, ConvectionKernels_BC67.cpp:1042-1662.
 * The 8 lanes run in lock-step because with BC7_RespectPunchThrough the commit rule of
 * BC67.cpp:1406-1428 couples them (AnySet guards + the operand order of
 * ParallelMath::AndNot, ParallelMath.h:900-905, which yields `invalid & ~better`). */
static void bc7_try_single_plane(const bc7_ctx_t *cx, bc7_lane_t *lanes)
{
    const orc_bc7_plan *plan = cx->plan;
    const uint32_t flags = cx->flags;
    int numRefineRounds = cx->refineRounds;
    if (numRefineRounds < 1)
        numRefineRounds = 1;

    const int anyBlockHasAlpha = cx->anyBlockHasAlpha;
    const int allowRGBModes = cx->allowRGBModes;
    const int allowMode7 = anyBlockHasAlpha || (plan->mode7RGBPartitionEnabled != 0);

    for (int l = 0; l < 8; l++)
    {
        bc7_lane_t *ln = &lanes[l];
        memset(ln->unfinishedRGB, 0, sizeof(ln->unfinishedRGB));
        memset(ln->unfinishedRGBA, 0, sizeof(ln->unfinishedRGBA));
        memset(ln->fragmentBestIndexes, 0, sizeof(ln->fragmentBestIndexes));
        memset(ln->shapeBestEP, 0, sizeof(ln->shapeBestEP));

        int maxAlpha = 0, minAlpha = 255, isPunchThrough = 1;
        for (int px = 0; px < 16; px++)
        {
            int a = ln->pixels[px][3];
            if (a > maxAlpha) maxAlpha = a;
            if (a < minAlpha) minAlpha = a;
            isPunchThrough = isPunchThrough && (a == 0 || a == 255);
        }
        ln->isPunchThrough = isPunchThrough;
        ln->blockHasNonMaxAlpha = minAlpha < 255;
        ln->blockHasNonZeroAlpha = 0 < maxAlpha;

        for (int px = 0; px < 16; px++)
            for (int ch = 0; ch < 4; ch++)
                ln->preWeighted[px][ch] = (float)ln->pixels[px][ch] * cx->w[ch];

        if (allowRGBModes)
        {
            for (int it = 0; it < plan->rgbNumShapesToEvaluate; it++)
            {
                int shape = plan->rgbShapeList[it];
                pca_endpoints(&ln->preWeighted[0][0], 4, 3, orc_fragments + orc_shape_start[shape], orc_shape_len[shape],
                              cx->w, &ln->unfinishedRGB[shape]);
            }
        }
        for (int it = 0; it < plan->rgbaNumShapesToEvaluate; it++)
        {
            int shape = plan->rgbaShapeList[it];
            if (anyBlockHasAlpha || !allowRGBModes)
                pca_endpoints(&ln->preWeighted[0][0], 4, 4, orc_fragments + orc_shape_start[shape], orc_shape_len[shape],
                              cx->w, &ln->unfinishedRGBA[shape]);
            else
            {
                /* ExpandTo<4>(255), UnfinishedEndpoints.h:93-114 */
                for (int ch = 0; ch < 3; ch++)
                {
                    ln->unfinishedRGBA[shape].base[ch] = ln->unfinishedRGB[shape].base[ch];
                    ln->unfinishedRGBA[shape].offset[ch] = ln->unfinishedRGB[shape].offset[ch];
                }
                ln->unfinishedRGBA[shape].base[3] = 255.0f;
                ln->unfinishedRGBA[shape].offset[3] = 0.0f;
            }
        }
    }

    for (int mode = 0; mode <= 7; mode++)
    {
        if (mode == 4 || mode == 5) continue;
        if (mode < 4 && !allowRGBModes) continue;
        if (mode == 7 && !allowMode7) continue;

        const bc7_mode_t *mi = &bc7_modes[mode];
        const int isRGB = mode < 4;
        const unsigned numPartitions = 1u << mi->partitionBits;
        const int numSubsets = mi->numSubsets;
        const int indexPrec = mi->indexBits;
        int parityBitMax = 1;
        if (mi->pBitMode == 0) parityBitMax = 4;
        else if (mi->pBitMode == 1) parityBitMax = 2;
        const int numRealChannels = isRGB ? 3 : 4;

        int numShapes;
        const uint8_t *shapeList;
        static const uint8_t list1[1] = {0};
        uint8_t list2[128];
        for (int i = 0; i < 128; i++) list2[i] = (uint8_t)(i + 1);
        if (numSubsets == 1) { numShapes = 1; shapeList = list1; }
        else if (numSubsets == 2) { numShapes = 128; shapeList = list2; }
        else if (numPartitions == 16) { numShapes = 36; shapeList = orc_shape_list3_short; }
        else { numShapes = 140; shapeList = orc_shape_list3; }

        for (int l = 0; l < 8; l++)
            for (int s = 0; s < 243; s++)
                lanes[l].shapeBestError[s] = FLT_MAX;

        for (int it = 0; it < numShapes; it++)
        {
            const int shape = shapeList[it];
            int numTweakRounds = isRGB ? plan->seedPointsForShapeRGB[shape] : plan->seedPointsForShapeRGBA[shape];
            if (numTweakRounds == 0) continue;
            if (numTweakRounds > 4) numTweakRounds = 4;

            const int shapeStart = orc_shape_start[shape];
            const int shapeLength = orc_shape_len[shape];
            const uint8_t *frag = orc_fragments + shapeStart;

            for (int l = 0; l < 8; l++)
            {
                bc7_lane_t *ln = &lanes[l];
                /* static alpha error of RGB modes on groups with alpha, BC67.cpp:1250-1264 */
                aggerr_t alphaAgg;
                agg_init(&alphaAgg);
                if (isRGB && anyBlockHasAlpha)
                {
                    for (int i = 0; i < shapeLength; i++)
                    {
                        int filled = 255;
                        agg_add_pixel(&alphaAgg, &filled, &ln->pixels[frag[i]][3], 1);
                    }
                }
                ln->staticAlphaError = agg_finalize(&alphaAgg, 1, flags, &cx->wSq[3]);

                for (int tweak = 0; tweak < numTweakRounds; tweak++)
                {
                    if (isRGB)
                    {
                        finish_ldr(&ln->unfinishedRGB[shape], 3, tweak, 1 << indexPrec, ln->tweakBaseEP[tweak][0], ln->tweakBaseEP[tweak][1]);
                        ln->tweakBaseEP[tweak][0][3] = ln->tweakBaseEP[tweak][1][3] = 255;
                    }
                    else
                        finish_ldr(&ln->unfinishedRGBA[shape], 4, tweak, 1 << indexPrec, ln->tweakBaseEP[tweak][0], ln->tweakBaseEP[tweak][1]);
                }

                for (int pIter = 0; pIter < parityBitMax; pIter++)
                {
                    ln->punchThroughInvalid[pIter] = 0;
                    if ((flags & ORC_FLAG_BC7_RESPECT_PUNCHTHROUGH) && (mode == 6 || mode == 7))
                    {
                        if (pIter == 0)
                            ln->punchThroughInvalid[pIter] = ln->isPunchThrough && ln->blockHasNonZeroAlpha;
                        else if (pIter == parityBitMax - 1)
                            ln->punchThroughInvalid[pIter] = ln->isPunchThrough && ln->blockHasNonMaxAlpha;
                        else
                            ln->punchThroughInvalid[pIter] = ln->isPunchThrough;
                    }
                }
            }

            for (int pIter = 0; pIter < parityBitMax; pIter++)
            {
                int allInvalid = 1, anyInvalid = 0;
                for (int l = 0; l < 8; l++)
                {
                    allInvalid = allInvalid && lanes[l].punchThroughInvalid[pIter];
                    anyInvalid = anyInvalid || lanes[l].punchThroughInvalid[pIter];
                }
                if (allInvalid)
                    continue;
                const int needPunchThroughCheck = anyInvalid;

                for (int tweak = 0; tweak < numTweakRounds; tweak++)
                {
                    const int p[2] = {pIter & 1, (pIter >> 1) & 1};
                    for (int l = 0; l < 8; l++)
                        memcpy(lanes[l].ep, lanes[l].tweakBaseEP[tweak], sizeof(lanes[l].ep));

                    for (int refine = 0; refine < numRefineRounds; refine++)
                    {
                        const int lastRefine = (refine == numRefineRounds - 1);
                        int better[8], anyBetter = 0;
                        for (int l = 0; l < 8; l++)
                        {
                            bc7_single_plane_trial(cx, &lanes[l], mode, p, frag, shapeLength, indexPrec, numRealChannels, isRGB, lastRefine);
                            better[l] = lanes[l].shapeError < lanes[l].shapeBestError[shape];
                            anyBetter = anyBetter || better[l];
                        }

                        if (anyBetter)
                        {
                            int punchThroughOK = 1;
                            if (needPunchThroughCheck)
                            {
                                /* AndNot(punchThroughInvalid, better) == invalid & ~better */
                                int any = 0;
                                for (int l = 0; l < 8; l++)
                                {
                                    better[l] = lanes[l].punchThroughInvalid[pIter] && !better[l];
                                    any = any || better[l];
                                }
                                if (!any)
                                    punchThroughOK = 0;
                            }
                            if (punchThroughOK)
                            {
                                for (int l = 0; l < 8; l++)
                                {
                                    if (!better[l]) continue;
                                    bc7_lane_t *ln = &lanes[l];
                                    ln->shapeBestError[shape] = ln->shapeError;
                                    for (int epi = 0; epi < 2; epi++)
                                        for (int ch = 0; ch < numRealChannels; ch++)
                                            ln->shapeBestEP[shape][epi][ch] = ln->ep[epi][ch];
                                    for (int i = 0; i < shapeLength; i++)
                                        ln->fragmentBestIndexes[shapeStart + i] = ln->indexes[i];
                                }
                            }
                        }

                        if (!lastRefine)
                            for (int l = 0; l < 8; l++)
                                refiner_get_ldr(&lanes[l].ref, lanes[l].ep);
                    }
                }
            }
            if (flags & ORC_FLAG_BC7_TRY_SINGLE_COLOR)
                bc7_try_single_color(cx, lanes, mode, shape, frag, shapeStart, shapeLength, numRealChannels);
        }

        /* partition argmin, BC67.cpp:1573-1660.  For mode 7 the reference assigns the
         * plan mask to a dead variable, so all 64 partitions are scanned. */
        uint64_t partitionsEnabledBits = 0xffffffffffffffffULL;
        switch (mode)
        {
        case 0: partitionsEnabledBits = plan->mode0PartitionEnabled; break;
        case 1: partitionsEnabledBits = plan->mode1PartitionEnabled; break;
        case 2: partitionsEnabledBits = plan->mode2PartitionEnabled; break;
        case 3: partitionsEnabledBits = plan->mode3PartitionEnabled; break;
        case 6: partitionsEnabledBits = plan->mode6Enabled ? 1 : 0; break;
        default: break;
        }

        for (unsigned partition = 0; partition < numPartitions; partition++)
        {
            if (((partitionsEnabledBits >> partition) & 1) == 0)
                continue;
            int partitionShapes[3] = {0, 0, 0};
            if (numSubsets == 2)
            {
                partitionShapes[0] = orc_shapes2[partition * 2 + 0];
                partitionShapes[1] = orc_shapes2[partition * 2 + 1];
            }
            else if (numSubsets == 3)
            {
                for (int s = 0; s < 3; s++)
                    partitionShapes[s] = orc_shapes3[partition * 3 + s];
            }
            for (int l = 0; l < 8; l++)
            {
                bc7_lane_t *ln = &lanes[l];
                bc7_work_t *work = &ln->work;
                float totalError = 0.0f;
                for (int s = 0; s < numSubsets; s++)
                    totalError = totalError + ln->shapeBestError[partitionShapes[s]];

                int better = totalError < work->error;
                if (mode == 7 && anyBlockHasAlpha)
                {
                    int rgbAllowed = ((plan->mode7RGBPartitionEnabled >> partition) & 1) != 0;
                    if (!rgbAllowed)
                        better = better && ln->blockHasNonMaxAlpha;
                }
                if (better)
                {
                    for (int s = 0; s < numSubsets; s++)
                    {
                        int shape = partitionShapes[s];
                        int ss = orc_shape_start[shape], sl = orc_shape_len[shape];
                        for (int epi = 0; epi < 2; epi++)
                            for (int ch = 0; ch < 4; ch++)
                                work->ep[s][epi][ch] = ln->shapeBestEP[shape][epi][ch];
                        for (int i = 0; i < sl; i++)
                            work->indexes[orc_fragments[ss + i]] = ln->fragmentBestIndexes[ss + i];
                    }
                    work->error = totalError;
                    work->mode = mode;
                    work->partOrIS = (int)partition;
                }
            }
        }
    }
}

/* BC7Computer::TweakAlpha, BC67.cpp:815-827 */
static void bc7_tweak_alpha(const int original[2], int tweak, int range, int result[2])
{
    float tf[2];
    tweak_factors(tweak, range, tf);
    float base = (float)original[0];
    float offs = (float)original[1] - base;
    result[0] = cvt_rne_s16(f_clamp(base + offs * tf[0], 0.0f, 255.0f));
    result[1] = cvt_rne_s16(f_clamp(base + offs * tf[1], 0.0f, 255.0f));
}

/* BC7Computer::TryDualPlane, BC67.cpp:1664-1965, one lane */
static void bc7_try_dual_plane(const bc7_ctx_t *cx, const int pixels[16][4], const float floatPixels[16][4],
                               bc7_work_t *work)
{
    const orc_bc7_plan *plan = cx->plan;
    const uint32_t flags = cx->flags;
    int numRefineRounds = cx->refineRounds;
    if (numRefineRounds < 1)
        numRefineRounds = 1;

    for (int mode = 4; mode <= 5; mode++)
    {
        int numSP[2] = {0, 0};
        for (int rotation = 0; rotation < 4; rotation++)
        {
            if (mode == 4)
            {
                numSP[0] = plan->mode4SP[rotation][0];
                numSP[1] = plan->mode4SP[rotation][1];
            }
            else
                numSP[0] = numSP[1] = plan->mode5SP[rotation];
            if (numSP[0] == 0 && numSP[1] == 0)
                continue;

            const int alphaChannel = (rotation + 3) & 3;
            const int redChannel = (rotation == 1) ? 3 : 0;
            const int greenChannel = (rotation == 2) ? 3 : 1;
            const int blueChannel = (rotation == 3) ? 3 : 2;

            int rotatedRGB[16][4];
            float floatRotatedRGB[16][3];
            for (int px = 0; px < 16; px++)
            {
                rotatedRGB[px][0] = pixels[px][redChannel];
                rotatedRGB[px][1] = pixels[px][greenChannel];
                rotatedRGB[px][2] = pixels[px][blueChannel];
                rotatedRGB[px][3] = 0;
                for (int ch = 0; ch < 3; ch++)
                    floatRotatedRGB[px][ch] = (float)rotatedRGB[px][ch];
            }
            const int maxIndexSelector = (mode == 4) ? 2 : 1;
            const float rotatedRGBWeights[3] = {cx->w[redChannel], cx->w[greenChannel], cx->w[blueChannel]};
            const float rotatedRGBWeightsSq[3] = {cx->wSq[redChannel], cx->wSq[greenChannel], cx->wSq[blueChannel]};
            const float rotatedAlphaWeightSq[1] = {cx->wSq[alphaChannel]};
            const float uniformWeight[1] = {1.0f};

            float preWeightedRotatedRGB[16][3];
            for (int px = 0; px < 16; px++)
                for (int ch = 0; ch < 3; ch++)
                    preWeightedRotatedRGB[px][ch] = (float)rotatedRGB[px][ch] * rotatedRGBWeights[ch];

            for (int indexSelector = 0; indexSelector < maxIndexSelector; indexSelector++)
            {
                int numTweakRounds = numSP[indexSelector];
                if (numTweakRounds <= 0) continue;
                if (numTweakRounds > 4) numTweakRounds = 4;

                unfinished_t unfinishedRGB;
                pca_endpoints(&preWeightedRotatedRGB[0][0], 3, 3, NULL, 16, rotatedRGBWeights, &unfinishedRGB);

                int alphaRange[2];
                alphaRange[0] = alphaRange[1] = pixels[0][alphaChannel];
                for (int px = 1; px < 16; px++)
                {
                    int a = pixels[px][alphaChannel];
                    if (a < alphaRange[0]) alphaRange[0] = a;
                    if (a > alphaRange[1]) alphaRange[1] = a;
                }

                int rgbPrec, alphaPrec;
                if (mode == 4)
                {
                    rgbPrec = indexSelector ? 3 : 2;
                    alphaPrec = indexSelector ? 2 : 3;
                }
                else
                    rgbPrec = alphaPrec = 2;

                float bestRGBError = FLT_MAX, bestAlphaError = FLT_MAX;
                int bestRGBIndexes[16], bestAlphaIndexes[16];
                int bestEP[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                for (int px = 0; px < 16; px++)
                    bestRGBIndexes[px] = bestAlphaIndexes[px] = 0;

                for (int tweak = 0; tweak < numTweakRounds; tweak++)
                {
                    int rgbEP[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
                    int alphaEP[2];
                    finish_ldr(&unfinishedRGB, 3, tweak, 1 << rgbPrec, rgbEP[0], rgbEP[1]);
                    bc7_tweak_alpha(alphaRange, tweak, 1 << alphaPrec, alphaEP);

                    for (int refine = 0; refine < numRefineRounds; refine++)
                    {
                        /* CompressEndpoints4/5, BC67.cpp:901-923 */
                        for (int j = 0; j < 2; j++)
                        {
                            if (mode == 4)
                            {
                                bc7_quantize(rgbEP[j], 5, 3);
                                bc7_unquantize(rgbEP[j], 5, 3);
                                bc7_quantize(alphaEP + j, 6, 1);
                                bc7_unquantize(alphaEP + j, 6, 1);
                            }
                            else
                            {
                                bc7_quantize(rgbEP[j], 7, 3);
                                bc7_unquantize(rgbEP[j], 7, 3);
                            }
                        }

                        selector_t alphaSel, rgbSel;
                        {
                            int alphaEPTemp[2][4] = {{alphaEP[0], 0, 0, 0}, {alphaEP[1], 0, 0, 0}};
                            selector_init(&alphaSel, 1, uniformWeight, alphaEPTemp, 1 << alphaPrec);
                        }
                        selector_init(&rgbSel, 3, rotatedRGBWeights, rgbEP, 1 << rgbPrec);

                        refiner_t rgbRef, alphaRef;
                        refiner_init(&rgbRef, 3, 1 << rgbPrec, rotatedRGBWeights, cx->rcp);
                        refiner_init(&alphaRef, 1, 1 << alphaPrec, uniformWeight, cx->rcp);

                        float errorRGB = 0.0f, errorA = 0.0f;
                        int rgbIndexes[16], alphaIndexes[16];
                        aggerr_t rgbAgg, alphaAgg;
                        agg_init(&rgbAgg);
                        agg_init(&alphaAgg);

                        for (int px = 0; px < 16; px++)
                        {
                            int rgbIndex = selector_select(&rgbSel, floatRotatedRGB[px]);
                            int alphaIndex = selector_select(&alphaSel, &floatPixels[px][alphaChannel]);
                            int recRGB[3], recA[1];
                            selector_reconstruct_bc7(&rgbSel, rgbIndex, recRGB, 3);
                            selector_reconstruct_bc7(&alphaSel, alphaIndex, recA, 1);

                            if (flags & ORC_FLAG_BC7_FAST_INDEXING)
                            {
                                agg_add_pixel(&rgbAgg, recRGB, rotatedRGB[px], 3);
                                agg_add_pixel(&alphaAgg, recA, &pixels[px][alphaChannel], 1);
                            }
                            else
                            {
                                float rgbError = error_ldr_simple(flags, 3, recRGB, rotatedRGB[px], 3, rotatedRGBWeightsSq);
                                float alphaError = error_ldr_simple(flags, 1, recA, &pixels[px][alphaChannel], 1, rotatedAlphaWeightSq);
                                int altRGB[2], altA[2];
                                altRGB[0] = (rgbIndex > 1 ? rgbIndex : 1) - 1;
                                altRGB[1] = (rgbIndex + 1 < (1 << rgbPrec) - 1) ? rgbIndex + 1 : (1 << rgbPrec) - 1;
                                altA[0] = (alphaIndex > 1 ? alphaIndex : 1) - 1;
                                altA[1] = (alphaIndex + 1 < (1 << alphaPrec) - 1) ? alphaIndex + 1 : (1 << alphaPrec) - 1;
                                for (int ii = 0; ii < 2; ii++)
                                {
                                    selector_reconstruct_bc7(&rgbSel, altRGB[ii], recRGB, 3);
                                    selector_reconstruct_bc7(&alphaSel, altA[ii], recA, 1);
                                    float altRGBError = error_ldr_simple(flags, 3, recRGB, rotatedRGB[px], 3, rotatedRGBWeightsSq);
                                    float altAlphaError = error_ldr_simple(flags, 1, recA, &pixels[px][alphaChannel], 1, rotatedAlphaWeightSq);
                                    int rgbBetter = altRGBError < rgbError;
                                    int alphaBetter = altAlphaError < alphaError;
                                    rgbError = f_min(altRGBError, rgbError);
                                    alphaError = f_min(altAlphaError, alphaError);
                                    if (rgbBetter) rgbIndex = altRGB[ii];
                                    if (alphaBetter) alphaIndex = altA[ii];
                                }
                                errorRGB = errorRGB + rgbError;
                                errorA = errorA + alphaError;
                            }

                            if (refine != numRefineRounds - 1)
                            {
                                refiner_contribute_unweighted(&rgbRef, preWeightedRotatedRGB[px], rgbIndex, 3);
                                refiner_contribute_unweighted(&alphaRef, &floatPixels[px][alphaChannel], alphaIndex, 1);
                            }
                            if (flags & ORC_FLAG_BC7_FAST_INDEXING)
                            {
                                errorRGB = agg_finalize(&rgbAgg, 3, flags, rotatedRGBWeightsSq);
                                errorA = agg_finalize(&alphaAgg, 1, flags, rotatedAlphaWeightSq);
                            }
                            rgbIndexes[px] = rgbIndex;
                            alphaIndexes[px] = alphaIndex;
                        }

                        if (errorRGB < bestRGBError)
                        {
                            bestRGBError = f_min(errorRGB, bestRGBError);
                            memcpy(bestRGBIndexes, rgbIndexes, sizeof(rgbIndexes));
                            for (int e = 0; e < 2; e++)
                                for (int ch = 0; ch < 3; ch++)
                                    bestEP[e][ch] = rgbEP[e][ch];
                        }
                        if (errorA < bestAlphaError)
                        {
                            bestAlphaError = f_min(errorA, bestAlphaError);
                            memcpy(bestAlphaIndexes, alphaIndexes, sizeof(alphaIndexes));
                            for (int e = 0; e < 2; e++)
                                bestEP[e][3] = alphaEP[e];
                        }

                        if (refine != numRefineRounds - 1)
                        {
                            refiner_get_ldr(&rgbRef, rgbEP);
                            int alphaEPTemp[2][4];
                            refiner_get_ldr(&alphaRef, alphaEPTemp);
                            alphaEP[0] = alphaEPTemp[0][0];
                            alphaEP[1] = alphaEPTemp[1][0];
                        }
                    }
                }

                float combinedError = bestRGBError + bestAlphaError;
                int better = combinedError < work->error;
                work->error = f_min(combinedError, work->error);
                if (better)
                {
                    work->mode = mode;
                    work->rotation = rotation;
                    work->partOrIS = indexSelector;
                    for (int px = 0; px < 16; px++)
                    {
                        work->indexes[px] = indexSelector ? bestAlphaIndexes[px] : bestRGBIndexes[px];
                        work->indexes2[px] = indexSelector ? bestRGBIndexes[px] : bestAlphaIndexes[px];
                    }
                    for (int e = 0; e < 2; e++)
                        for (int ch = 0; ch < 4; ch++)
                            work->ep[0][e][ch] = bestEP[e][ch];
                }
            }
        }
    }
}

/* PackingVector, BC67.cpp:652-698 */
typedef struct
{
    uint32_t v[5];
    int offset;
} packer_t;

static void pk_init(packer_t *pk) { memset(pk, 0, sizeof(*pk)); }

static void pk_pack(packer_t *pk, unsigned value, int bits)
{
    value &= 0xffffu; /* ScalarUInt16 parameter */
    int vOffset = pk->offset >> 5;
    int bitOffset = pk->offset & 0x1f;
    pk->v[vOffset] |= (uint32_t)value << bitOffset;
    int overflowBits = bitOffset + bits - 32;
    if (overflowBits > 0)
        pk->v[vOffset + 1] |= (uint32_t)value >> (bits - overflowBits);
    pk->offset += bits;
}

static void pk_flush(const packer_t *pk, uint8_t *out)
{
    for (int v = 0; v < 4; v++)
        for (int b = 0; b < 4; b++)
            out[v * 4 + b] = (uint8_t)((pk->v[v] >> (b * 8)) & 0xff);
}

/* per-block tail of BC7Computer::Pack, BC67.cpp:2003-2203 */
static void bc7_emit(const bc7_work_t *work, uint8_t *out)
{
    packer_t pv;
    pk_init(&pv);
    const int mode = work->mode;
    const int partition = work->partOrIS;
    const int indexSelector = work->partOrIS;
    const bc7_mode_t *mi = &bc7_modes[mode];

    int indexes[16], indexes2[16], endPoints[3][2][4];
    memcpy(indexes, work->indexes, sizeof(indexes));
    memcpy(indexes2, work->indexes2, sizeof(indexes2));
    memcpy(endPoints, work->ep, sizeof(endPoints));

    int fixups[3] = {0, 0, 0};
    if (mi->alphaMode == 1)
    {
        int flipRGB = (indexes[0] & (1 << (mi->indexBits - 1))) != 0;
        int flipAlpha = (indexes2[0] & (1 << (mi->alphaIndexBits - 1))) != 0;
        if (flipRGB)
        {
            int hi = (1 << mi->indexBits) - 1;
            for (int px = 0; px < 16; px++) indexes[px] = hi - indexes[px];
        }
        if (flipAlpha)
        {
            int hi = (1 << mi->alphaIndexBits) - 1;
            for (int px = 0; px < 16; px++) indexes2[px] = hi - indexes2[px];
        }
        if (indexSelector)
        {
            int t = flipRGB; flipRGB = flipAlpha; flipAlpha = t;
        }
        if (flipRGB)
            for (int ch = 0; ch < 3; ch++)
            {
                int t = endPoints[0][0][ch]; endPoints[0][0][ch] = endPoints[0][1][ch]; endPoints[0][1][ch] = t;
            }
        if (flipAlpha)
        {
            int t = endPoints[0][0][3]; endPoints[0][0][3] = endPoints[0][1][3]; endPoints[0][1][3] = t;
        }
    }
    else
    {
        if (mi->numSubsets == 2)
            fixups[1] = orc_anchor2[partition];
        else if (mi->numSubsets == 3)
        {
            fixups[1] = orc_anchor3[partition * 2 + 0];
            fixups[2] = orc_anchor3[partition * 2 + 1];
        }
        int flip[3] = {0, 0, 0};
        for (int s = 0; s < mi->numSubsets; s++)
            flip[s] = (indexes[fixups[s]] & (1 << (mi->indexBits - 1))) != 0;
        if (flip[0] || flip[1] || flip[2])
        {
            int hi = (1 << mi->indexBits) - 1;
            for (int px = 0; px < 16; px++)
            {
                int subset = 0;
                if (mi->numSubsets == 2)
                    subset = (orc_partition2[partition] >> px) & 1;
                else if (mi->numSubsets == 3)
                    subset = (orc_partition3[partition] >> (px * 2)) & 3;
                if (flip[subset])
                    indexes[px] = hi - indexes[px];
            }
            int maxCH = (mi->alphaMode == 0) ? 4 : 3;
            for (int s = 0; s < mi->numSubsets; s++)
                if (flip[s])
                    for (int ch = 0; ch < maxCH; ch++)
                    {
                        int t = endPoints[s][0][ch]; endPoints[s][0][ch] = endPoints[s][1][ch]; endPoints[s][1][ch] = t;
                    }
        }
    }

    pk_pack(&pv, (uint8_t)(1 << mode), mode + 1);
    if (mi->partitionBits)
        pk_pack(&pv, (unsigned)partition, mi->partitionBits);
    if (mi->alphaMode == 1)
        pk_pack(&pv, (unsigned)work->rotation, 2);
    if (mi->hasIndexSelector)
        pk_pack(&pv, (unsigned)indexSelector, 1);

    for (int ch = 0; ch < 3; ch++)
        for (int s = 0; s < mi->numSubsets; s++)
            for (int e = 0; e < 2; e++)
                pk_pack(&pv, (unsigned)(endPoints[s][e][ch] & 0xffff) >> (8 - mi->rgbBits), mi->rgbBits);
    if (mi->alphaMode != 2)
        for (int s = 0; s < mi->numSubsets; s++)
            for (int e = 0; e < 2; e++)
                pk_pack(&pv, (unsigned)(endPoints[s][e][3] & 0xffff) >> (8 - mi->alphaBits), mi->alphaBits);

    if (mi->pBitMode == 1)
    {
        for (int s = 0; s < mi->numSubsets; s++)
            pk_pack(&pv, ((unsigned)(endPoints[s][0][0] & 0xffff) >> (7 - mi->rgbBits)) & 1, 1);
    }
    else if (mi->pBitMode == 0)
    {
        for (int s = 0; s < mi->numSubsets; s++)
            for (int e = 0; e < 2; e++)
                pk_pack(&pv, ((unsigned)(endPoints[s][e][0] & 0xffff) >> (7 - mi->rgbBits)) & 1, 1);
    }

    for (int px = 0; px < 16; px++)
    {
        int bits = mi->indexBits;
        if (px == 0 || px == fixups[1] || px == fixups[2])
            bits--;
        pk_pack(&pv, (unsigned)indexes[px], bits);
    }
    if (mi->alphaMode == 1)
        for (int px = 0; px < 16; px++)
        {
            int bits = mi->alphaIndexBits;
            if (px == 0) bits--;
            pk_pack(&pv, (unsigned)indexes2[px], bits);
        }
    pk_flush(&pv, out);
}

/* BC7Computer::Pack for one group of 8 blocks, BC67.cpp:1975-2204 */
static void bc7_encode_group(const uint8_t *blocks, uint8_t *out, const orc_options *options,
                             const orc_bc7_plan *plan, const float *rcp)
{
    bc7_ctx_t cx;
    cx.flags = options->flags;
    fill_weights(options, cx.w);
    for (int ch = 0; ch < 4; ch++)
        cx.wSq[ch] = cx.w[ch] * cx.w[ch];
    cx.plan = plan;
    cx.refineRounds = options->refineRoundsBC7;
    cx.rcp = rcp;

    /* group-wide booleans (AnySet over the 8 lanes), BC67.cpp:1066-1072 */
    cx.anyBlockHasAlpha = 0;
    cx.allowRGBModes = 0;
    for (int b = 0; b < 8; b++)
    {
        int minAlpha = 255;
        for (int px = 0; px < 16; px++)
        {
            int a = blocks[b * 64 + px * 4 + 3];
            if (a < minAlpha) minAlpha = a;
        }
        if (minAlpha < 255) cx.anyBlockHasAlpha = 1;
        if (250 < minAlpha) cx.allowRGBModes = 1;
    }

    static __thread bc7_lane_t lanes[8];
    for (int b = 0; b < 8; b++)
    {
        bc7_lane_t *ln = &lanes[b];
        for (int px = 0; px < 16; px++)
            for (int ch = 0; ch < 4; ch++)
            {
                ln->pixels[px][ch] = blocks[b * 64 + px * 4 + ch];
                ln->floatPixels[px][ch] = (float)ln->pixels[px][ch];
            }
        memset(&ln->work, 0, sizeof(ln->work));
        ln->work.error = FLT_MAX;
    }
    bc7_try_single_plane(&cx, lanes);
    for (int b = 0; b < 8; b++)
    {
        bc7_try_dual_plane(&cx, lanes[b].pixels, lanes[b].floatPixels, &lanes[b].work);
        bc7_emit(&lanes[b].work, out + b * 16);
    }
}

/* ------------------------------------------------------------------------------------
 * threading over groups
 * ---------------------------------------------------------------------------------- */
typedef void (*group_fn)(const uint8_t *in, uint8_t *out, const void *a, const void *b, const float *rcp);

typedef struct
{
    group_fn fn;
    const uint8_t *in;
    uint8_t *out;
    size_t inStride, outStride;
    size_t groupBegin, groupEnd;
    const void *a, *b;
    const float *rcp;
} job_t;

static void *job_main(void *arg)
{
    job_t *j = (job_t *)arg;
    unsigned csr = _mm_getcsr();
    _mm_setcsr((csr & ~_MM_ROUND_MASK) | _MM_ROUND_NEAREST); /* RoundTowardNearestForScope */
    for (size_t g = j->groupBegin; g < j->groupEnd; g++)
        j->fn(j->in + g * j->inStride, j->out + g * j->outStride, j->a, j->b, j->rcp);
    _mm_setcsr(csr);
    return NULL;
}

static void run_groups(group_fn fn, const uint8_t *in, uint8_t *out, size_t numGroups, size_t inStride,
                       size_t outStride, const void *a, const void *b, const float *rcp, int threads)
{
    if (threads < 1) threads = 1;
    if ((size_t)threads > numGroups) threads = (int)(numGroups ? numGroups : 1);
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *tids = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++)
    {
        jobs[t].fn = fn;
        jobs[t].in = in;
        jobs[t].out = out;
        jobs[t].inStride = inStride;
        jobs[t].outStride = outStride;
        jobs[t].groupBegin = numGroups * (size_t)t / (size_t)threads;
        jobs[t].groupEnd = numGroups * (size_t)(t + 1) / (size_t)threads;
        jobs[t].a = a;
        jobs[t].b = b;
        jobs[t].rcp = rcp;
    }
    if (threads == 1)
        job_main(&jobs[0]);
    else
    {
        for (int t = 0; t < threads; t++)
            pthread_create(&tids[t], NULL, job_main, &jobs[t]);
        for (int t = 0; t < threads; t++)
            pthread_join(tids[t], NULL);
    }
    free(jobs);
    free(tids);
}

static void bc7_group_thunk(const uint8_t *in, uint8_t *out, const void *a, const void *b, const float *rcp)
{
    bc7_encode_group(in, out, (const orc_options *)a, (const orc_bc7_plan *)b, rcp);
}

int orc_encode_bc7(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const orc_options *options,
                   const orc_bc7_plan *plan, const float *rcp17, int threads)
{
    if (numBlocks % 8 != 0)
        return -1;
    float probed[17];
    if (!rcp17)
    {
        orc_probe_rcp(probed);
        rcp17 = probed;
    }
    run_groups(bc7_group_thunk, blocks, out, numBlocks / 8, 8 * 64, 8 * 16, options, plan, rcp17, threads);
    return 0;
}

#include "cvtt_oracle_bc1.inc"
#include "cvtt_oracle_s3tc.inc"
#include "cvtt_oracle_bc6h.inc"
#include "cvtt_oracle_etc2.inc"
