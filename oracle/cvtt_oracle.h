/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's per-block encoders.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load the
 * library built from this file.  The product path (convectionkernels_amd/) never does.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against the real
 * reference compiled from /root/reference (oracle/_ref, see oracle/Makefile) and against
 * the committed golden vectors under tests/golden/ (tests/test_oracle_*.py).
 */
#ifndef CVTT_ORACLE_H
#define CVTT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Byte-compatible images of the reference PODs (ConvectionKernels.h:73-103, 142-199). */
typedef struct orc_options
{
    uint32_t flags;
    float threshold;
    float redWeight, greenWeight, blueWeight, alphaWeight;
    int32_t refineRoundsBC7, refineRoundsBC6H, refineRoundsIIC, refineRoundsS3TC;
    int32_t seedPoints;
} orc_options;

typedef struct orc_bc7_plan
{
    uint64_t mode1PartitionEnabled;
    uint64_t mode2PartitionEnabled;
    uint64_t mode3PartitionEnabled;
    uint16_t mode0PartitionEnabled;
    uint64_t mode7RGBAPartitionEnabled;
    uint64_t mode7RGBPartitionEnabled;
    uint8_t mode4SP[4][2];
    uint8_t mode5SP[4];
    uint8_t mode6Enabled; /* C++ bool */
    uint8_t seedPointsForShapeRGB[243];
    uint8_t seedPointsForShapeRGBA[129];
    uint8_t rgbaShapeList[129];
    uint8_t rgbaNumShapesToEvaluate;
    uint8_t rgbShapeList[243];
    uint8_t rgbNumShapesToEvaluate;
} orc_bc7_plan;

enum
{
    ORC_FLAG_BC7_FAST_INDEXING = 0x008,
    ORC_FLAG_BC7_TRY_SINGLE_COLOR = 0x010,
    ORC_FLAG_BC7_RESPECT_PUNCHTHROUGH = 0x020,
    ORC_FLAG_BC6H_FAST_INDEXING = 0x040,
    ORC_FLAG_S3TC_EXHAUSTIVE = 0x080,
    ORC_FLAG_S3TC_PARANOID = 0x100,
    ORC_FLAG_UNIFORM = 0x200
};

size_t orc_sizeof_options(void);
size_t orc_sizeof_bc7_plan(void);

/* rcpps(i) for i = 1..16 on this host (entry 0 = entry 1); SURVEY App. A. */
void orc_probe_rcp(float out17[17]);

/* Group g = blocks [8g, 8g+8).  numBlocks must be a multiple of 8.
 * rcp17: the 17-entry reciprocal table to use (NULL = probe this host).
 * threads: worker threads over groups (<=1: run inline).
 * Returns 0, or a negative value for unsupported flag combinations. */
int orc_encode_bc7(uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                   const orc_options *options, const orc_bc7_plan *plan,
                   const float *rcp17, int threads);

int orc_encode_bc1(uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                   const orc_options *options, const float *rcp17, int threads);

/* blocksF16: numBlocks * 128 bytes of PixelBlockF16 (half bits as int16, RGBA; alpha ignored). */
int orc_encode_bc6h(uint8_t *out, const uint8_t *blocksF16, size_t numBlocks,
                    const orc_options *options, int isSigned, const float *rcp17, int threads);

/* mode 0: ETC2 RGB (8 B/block), 1: ETC2 RGBA = [EAC alpha | colour] (16 B/block), 2: EAC alpha (8 B/block) */
int orc_encode_etc2(uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                    const orc_options *options, int mode, int threads);
/* the same with the Options of AllocETC2Data given apart (their colour weights fix the chroma axes, ETC.cpp:3117-3145) */
int orc_encode_etc2_alloc(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const orc_options *options,
                          const orc_options *allocOptions, int mode, int threads);

/* BC2 / BC3 / BC4U / BC4S / BC5U / BC5S (format 2..7; signed formats take PixelBlockS8): 8 B (BC4) or 16 B per block */
int orc_encode_s3tc(uint8_t *out, const uint8_t *blocks, size_t numBlocks, const orc_options *options, int format,
                    const float *rcp17, int threads);

/* EAC R11 (cvtt::Kernels::EncodeETC2Alpha11): numBlocks * 16 int16 (PixelBlockScalarS16) -> 8 B/block */
int orc_encode_eac11(uint8_t *out, const int16_t *blocksS16, size_t numBlocks, int isSigned);

#ifdef __cplusplus
}
#endif
#endif
