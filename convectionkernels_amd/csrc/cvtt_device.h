// Shared host/device declarations of the MI355X encoder (product code).
#ifndef CVTTMI_DEVICE_H
#define CVTTMI_DEVICE_H

#include <stdint.h>
#include "../../include/cvtt_mi355x.h"

// Constant tables resident in HBM (uploaded once per context, read through the scalar
// cache: every access in the kernels uses a wave-uniform index).
struct CvttDeviceTables
{
    uint16_t shapeMask[243];   // pixel bitmask of every BC7 shape (tools/gen_tables.py)
    uint16_t partition2[64];   // two-subset partition bitmaps
    uint32_t partition3[64];   // three-subset partition maps, 2 bits per pixel
    uint8_t shapes2[64][2];    // shape ids of the subsets of a two-subset partition
    uint8_t shapes3[64][3];    // ... three-subset partition
    uint8_t anchor2[64];
    uint8_t anchor3[64][2];
    // Util::ComputeTweakFactors (reference ConvectionKernels_Util.cpp:75-84) evaluated on
    // the host in binary32 for range = 4, 8, 16 (index = log2(range) - 2)
    float tweakFactors[3][4][2];
    float rcpTable[17];        // host RCPPS(i), i = 1..16
    float rcpMaxIndex[5];      // 1.0f / ((1 << bits) - 1), bits = 0..4 (entry 0 unused)
    float tweakFactors3[2][4][2]; // same for range = 3, 4 (BC1; range 3 has 3 seed points)
    // BC6H mode table {mode id, two regions, transformed, endpoint bits, delta bits r,g,b, pad}
    // and header bit layout (field*16+bit; tools/gen_bc6h_layout.py)
    uint8_t bc6hModeInfo[14][8];
    uint8_t bc6hLayout[14][82];
    // ETC1 / ETC2 / EAC tables (tools/gen_etc_tables.py)
    int16_t etc1Modifiers[8][4];   // {-large, -small, +small, +large}
    int16_t thDistance[8];
    uint8_t eacPositive[16][4];
    uint8_t eacRounding[16][13];
    // the same two tables as words: the four positive modifiers of a table as bytes, its row of the rounding table as 13
    // two-bit fields (the EAC search inside the ETC2 RGBA colour kernel: a candidate's table is a per-lane value)
    uint32_t eacPosWord[16];
    uint32_t eacRoundBits[16];
    uint8_t clusterCount[8];
    uint16_t clusterStart[8];
    int16_t clusterOffsets[632];
    // the same list with what the base-colour pass of the cluster fit derives from the entry number: offset (bits 0-15) |
    // table << 16 | (first entry of its table) << 19 -- one load instead of a seven-step comparison chain per lane and trip
    uint32_t clusterEntry[640];
    // pixel bitmasks of subsets 1 and 2 of every three-subset partition (subset 1 of a
    // two-subset partition is partition2 itself)
    uint16_t subsetMask3[64][2];
    // BC1-family single-colour end points (tools/gen_s3tc_single_color.py): [paranoid*4 + (range==3)*2 + green][value] = {min, max, colour, span}
    uint8_t s3tcSingleColor[8][256][4];
    // ETC_UseFakeBT709 rounding table (tools/gen_fake709_rounding.py): [r << 8 | g << 4 | b] = nearest cell corner
    uint8_t fake709Rounding[4096];
    // Byte masks of the subsets the first-tier BC7 bounds sum over on the 8-bit grid (bc7_kernel.hip, maskedSums): pixel i
    // of the subset -> byte i % 4 of word i / 4 is 0xff.  [partition] = subset 1 of a two-subset partition,
    // [64 + 2 * partition + s] = subset 1 + s of a three-subset partition.
    uint32_t subsetByteMask[192][4];
};

// The caller's plan plus two bitmaps derived on the host: which shapes the plan's
// rgbShapeList / rgbaShapeList actually name.  The reference only computes PCA seeds for
// listed shapes (BC67.cpp:1085-1144); a shape with seed points that is NOT listed keeps the
// zero-initialised seeds of the canonical oracle build (hazard H5).
struct CvttBc7DevicePlan
{
    cvttmi_bc7_plan plan;
    uint32_t rgbListed[8];
    uint32_t rgbaListed[5];
};

// BC7 blocks whose mode-7 partitions are searched by a second launch (bc7_kernel.hip, HARD): what the first launch
// leaves per block, and what every wavefront of the second one leaves per partition slice.
constexpr int kHardWaves = 16;
struct CvttBc7HardRec
{
    uint32_t blockIndex;
    float err;    // best error without the handed-over partitions
    int32_t seq;  // its position in the reference's candidate order
    uint32_t aliveLo, aliveHi; // the partitions handed over
    uint32_t pad[3];
};
struct CvttBc7HardCand
{
    uint32_t packed[4];
    float err;    // FLT_MAX: the slice holds nothing better than the record
    int32_t seq;
    uint32_t pad[2];
};

// Per-launch uniform parameters (kernel argument, lives in SGPRs).
struct CvttBc7Args
{
    float w[4];      // Util::FillWeights (reference ConvectionKernels_Util.cpp:62-73)
    float wSq[4];    // w*w, rounded once (reference BC67.cpp:1047-1050)
    float rcpW[4];   // EndpointRefiner::Init m_rcpChannelWeights (EndpointRefiner.h:52-58)
    uint32_t flags;
    int32_t refineRounds;
    uint32_t numBlocks;
    // exact branch-and-bound (bc7_kernel.hip: shapeErrorLowerBound): 0 = exhaustive search
    uint32_t prune;
    float delta3;    // 0.5*sqrt(wSq[0]+wSq[1]+wSq[2]), rounded up
    float delta4;    // 0.5*sqrt(wSq[0..3]), rounded up
    float wSqSum3;   // wSq[0] + wSq[1] + wSq[2] (scale of the bound grids; summed on the host so that it arrives in a scalar register)
    // hand-over of blocks with many live mode-7 partitions to a second launch; hardCap = 0: off
    uint32_t hardCap;   // record slots of this launch
    uint32_t hardMin;   // live partitions of a wave from which its blocks are handed over, at the end of the grid
    uint32_t hardDiv;   // ... and one more for every hardDiv waves that follow it
    uint32_t *hardCount;     // slots claimed in THIS encode (zero when its first launch starts: see hardCountNext)
    uint32_t *hardCountNext; // the counter of the next encode on this context: zeroed by this encode's first launch
    CvttBc7HardRec *hardRec;
    CvttBc7HardCand *hardCand; // [hardCap][kHardWaves]
    // BC7_RespectPunchThrough with more refine rounds than the LDS trial table holds (bc7_kernel.hip, kMaxPTRefine): the
    // table of every wave of the launch in HBM, [wave][unit 32][chain 16][round]; NULL: the LDS table is used
    float *ptTrial;
};

// BC1 per-launch parameters.
struct CvttBc1Args
{
    float w[4];
    float wSq[4];
    float rcpW[4];
    uint32_t flags;
    int32_t refineRounds; // Options::refineRoundsS3TC
    int32_t seedPoints;   // Options::seedPoints
    int32_t threshold;    // floor(threshold * 255 + 0.5) as a signed 16-bit lane value (S3TC.cpp:748)
    uint32_t numBlocks;
    uint32_t alphaTest;   // 1: BC1 (EncodeBC1); 0: the colour half of BC2 / BC3 (range 4 only, every pixel weighs 1)
    uint32_t outStride;   // bytes between output blocks (8, or 16 inside BC2 / BC3)
    uint32_t outOffset;
};

// BC6H per-launch parameters.
struct CvttBc6hArgs
{
    float w[4];
    float wSq[4];
    float rcpW[4];
    uint32_t flags;
    int32_t refineRounds; // Options::refineRoundsBC6H (clamped to 1..3 by the kernel)
    int32_t seedPoints;   // Options::seedPoints (clamped to 1..4)
    uint32_t numBlocks;
};

// ETC2 / EAC per-launch parameters.
struct CvttEtcArgs
{
    float rw, gw, bw;   // Options weights, used directly (reference ETC.cpp:73-80, 2150-2152)
    float axis0[3];     // chroma-plane axes of ETC2CompressionDataInternal (reference ETC.cpp:3117-3145)
    float axis1[3];
    uint32_t flags;
    uint32_t numBlocks;
    uint32_t outStride; // bytes between consecutive output blocks (8, or 16 for RGBA)
    uint32_t outOffset; // byte offset of this kernel's 8 bytes inside the output block
    uint64_t debug;     // developer builds (-DCVTT_ETC_DEBUG): device pointer for per-stage errors, else 0
    uint32_t alphaThreshold; // punch-through: pixels with alpha below this are transparent (reference ETC.cpp:1672-1675)
};

#endif
