// cvtt/ConvectionKernels.h -- source-compatible C++ face of the MI355X encoder.
//
// A code base written against the reference's public header (reference ConvectionKernels.h:31-278)
// can include this file instead and link libcvtt_mi355x.so: the namespaces, PODs (same field
// order and sizes: Options 44 B, BC7EncodingPlan 808 B), flag values, pixel-block types and the
// cvtt::Kernels entry points of the hot path keep their names and call convention -- every
// Encode* call still consumes NumParallelBlocks input blocks and writes as many output blocks --
// but the work is done by the HIP kernels behind the C ABI of cvtt_mi355x.h.  A launch per 8
// blocks is only there for compatibility; throughput comes from the *Batch overloads below (or
// the C ABI directly), which take any multiple of 8 blocks.
//
//
// Error behaviour: the reference's functions return void and assert.  These abort() with a
// message on stderr when no gfx950 device is present or a call fails -- there is no CPU path.
#ifndef CVTT_MI355X_CONVECTION_KERNELS_H
#define CVTT_MI355X_CONVECTION_KERNELS_H

#include <stddef.h>
#include <stdint.h>

namespace cvtt
{
    namespace Flags
    {
        enum : uint32_t
        {
            BC7_FastIndexing = 0x008,
            BC7_TrySingleColor = 0x010,
            BC7_RespectPunchThrough = 0x020,
            BC6H_FastIndexing = 0x040,
            S3TC_Exhaustive = 0x080,
            S3TC_Paranoid = 0x100,
            Uniform = 0x200,
            ETC_UseFakeBT709 = 0x400,
            ETC_FakeBT709Accurate = 0x800,

            Fastest = BC6H_FastIndexing | BC7_FastIndexing | S3TC_Paranoid,
            Faster = Fastest,
            Fast = BC7_FastIndexing | S3TC_Paranoid,
            Default = Fast,
            Better = S3TC_Paranoid | S3TC_Exhaustive,
            Ultra = BC7_TrySingleColor | S3TC_Paranoid | S3TC_Exhaustive | ETC_FakeBT709Accurate
        };
    }

    const unsigned int NumParallelBlocks = 8;

    struct Options
    {
        uint32_t flags;
        float threshold;
        float redWeight, greenWeight, blueWeight, alphaWeight;
        int refineRoundsBC7, refineRoundsBC6H, refineRoundsIIC, refineRoundsS3TC;
        int seedPoints;
        Options(); // Default flags, threshold 0.5, Rec.709-derived weights, 2/3/8/2 refine rounds, 4 seed points
    };

    struct BC7FineTuningParams
    {
        // seed points (0 = off) per mode and partition / rotation / index selector
        uint8_t mode0SP[16];
        uint8_t mode1SP[64];
        uint8_t mode2SP[64];
        uint8_t mode3SP[64];
        uint8_t mode4SP[4][2];
        uint8_t mode5SP[4];
        uint8_t mode6SP;
        uint8_t mode7SP[64];
        BC7FineTuningParams(); // four seed points everywhere
    };

    struct BC7EncodingPlan
    {
        static const int kNumRGBAShapes = 129;
        static const int kNumRGBShapes = 243;
        uint64_t mode1PartitionEnabled, mode2PartitionEnabled, mode3PartitionEnabled;
        uint16_t mode0PartitionEnabled;
        uint64_t mode7RGBAPartitionEnabled, mode7RGBPartitionEnabled;
        uint8_t mode4SP[4][2];
        uint8_t mode5SP[4];
        bool mode6Enabled;
        uint8_t seedPointsForShapeRGB[kNumRGBShapes];
        uint8_t seedPointsForShapeRGBA[kNumRGBAShapes];
        uint8_t rgbaShapeList[kNumRGBAShapes];
        uint8_t rgbaNumShapesToEvaluate;
        uint8_t rgbShapeList[kNumRGBShapes];
        uint8_t rgbNumShapesToEvaluate;
        BC7EncodingPlan(); // every shape and partition enabled, four seed points: the maximum-quality plan
    };

    struct PixelBlockU8 { uint8_t m_pixels[16][4]; };
    struct PixelBlockS8 { int8_t m_pixels[16][4]; };
    struct PixelBlockScalarS16 { int16_t m_pixels[16]; };
    struct PixelBlockF16 { int16_t m_pixels[16][4]; }; // half-float bit patterns

    // Kept so that AllocETC2Data / ReleaseETC2Data call sites compile: the GPU kernels hold their
    // scratch in LDS, so the object only remembers the allocator context.
    class ETC2CompressionData { protected: ETC2CompressionData() {} };
    class ETC1CompressionData { protected: ETC1CompressionData() {} };

    namespace Kernels
    {
        typedef void *allocFunc_t(void *context, size_t size);
        typedef void freeFunc_t(void *context, void *ptr, size_t size);

        // 8 blocks in, 8 blocks out -- the reference's call convention
        void EncodeBC1(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options);
        void EncodeBC2(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options);
        void EncodeBC3(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options);
        void EncodeBC4U(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options);
        void EncodeBC4S(uint8_t *pBC, const PixelBlockS8 *pBlocks, const Options &options);
        void EncodeBC5U(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options);
        void EncodeBC5S(uint8_t *pBC, const PixelBlockS8 *pBlocks, const Options &options);
        void EncodeBC6HU(uint8_t *pBC, const PixelBlockF16 *pBlocks, const Options &options);
        void EncodeBC6HS(uint8_t *pBC, const PixelBlockF16 *pBlocks, const Options &options);
        void EncodeBC7(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, const BC7EncodingPlan &encodingPlan);
        void EncodeETC1(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC1CompressionData *compressionData);
        void EncodeETC2(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *compressionData);
        void EncodeETC2RGBA(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *compressionData);
        void EncodeETC2PunchthroughAlpha(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options, ETC2CompressionData *compressionData);
        void EncodeETC2Alpha(uint8_t *pBC, const PixelBlockU8 *pBlocks, const Options &options);

        void EncodeETC2Alpha11(uint8_t *pBC, const PixelBlockScalarS16 *pBlocks, bool isSigned, const Options &options);
        void DecodeBC7(PixelBlockU8 *pBlocks, const uint8_t *pBC);
        void DecodeBC6HU(PixelBlockF16 *pBlocks, const uint8_t *pBC);
        void DecodeBC6HS(PixelBlockF16 *pBlocks, const uint8_t *pBC);

        // quality 1..100 (clamped); byte-identical plans to the reference's.  Host-side, need no device.
        void ConfigureBC7EncodingPlanFromQuality(BC7EncodingPlan &encodingPlan, int quality);
        bool ConfigureBC7EncodingPlanFromFineTuningParams(BC7EncodingPlan &encodingPlan, const BC7FineTuningParams &params);

        // As in the reference (ETC.cpp:3117-3145) the two chroma axes of the T / H sector split belong to the Options given
        // HERE: the block AllocETC2Data returns keeps them, and EncodeETC2 / EncodeETC2RGBA / EncodeETC2PunchthroughAlpha use
        // them together with the error weights of the Options of the Encode call (cvttmi_encode_etc2_with_data).
        ETC2CompressionData *AllocETC2Data(allocFunc_t allocFunc, void *context, const Options &options);
        void ReleaseETC2Data(ETC2CompressionData *compressionData, freeFunc_t freeFunc);
        ETC1CompressionData *AllocETC1Data(allocFunc_t allocFunc, void *context);
        void ReleaseETC1Data(ETC1CompressionData *compressionData, freeFunc_t freeFunc);

        // numBlocks (a multiple of NumParallelBlocks) blocks per call; group g = blocks [8g, 8g+8)
        void EncodeBC1Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC2Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC3Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC4UBatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC4SBatch(uint8_t *pBC, const PixelBlockS8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC5UBatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC5SBatch(uint8_t *pBC, const PixelBlockS8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC6HUBatch(uint8_t *pBC, const PixelBlockF16 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC6HSBatch(uint8_t *pBC, const PixelBlockF16 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeBC7Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, const BC7EncodingPlan &encodingPlan);
        void EncodeETC1Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options);
        // (compressionData: what AllocETC2Data returned -- the chroma axes belong to the Options it was allocated with, as in
        // the reference, ConvectionKernels_ETC.cpp:3117-3145; NULL = the axes of `options`)
        void EncodeETC2Batch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, ETC2CompressionData *compressionData = 0);
        void EncodeETC2RGBABatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, ETC2CompressionData *compressionData = 0);
        void EncodeETC2PunchthroughAlphaBatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options, ETC2CompressionData *compressionData = 0);
        void EncodeETC2AlphaBatch(uint8_t *pBC, const PixelBlockU8 *pBlocks, size_t numBlocks, const Options &options);
        void EncodeETC2Alpha11Batch(uint8_t *pBC, const PixelBlockScalarS16 *pBlocks, size_t numBlocks, bool isSigned, const Options &options);
        void DecodeBC7Batch(PixelBlockU8 *pBlocks, const uint8_t *pBC, size_t numBlocks);
        void DecodeBC6HUBatch(PixelBlockF16 *pBlocks, const uint8_t *pBC, size_t numBlocks);
        void DecodeBC6HSBatch(PixelBlockF16 *pBlocks, const uint8_t *pBC, size_t numBlocks);
    }
}

#endif
