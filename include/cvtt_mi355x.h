/* cvtt_mi355x.h -- C ABI of the MI355X-native block-texture encoder.
 *
 * This is the drop-in boundary for the hot path of elasota/ConvectionKernels: the
 * per-block endpoint / partition / index search behind
 *     cvtt::Kernels::EncodeBC7        (reference ConvectionKernels.h:251, ConvectionKernels_API.cpp:41-54)
 *     cvtt::Kernels::EncodeBC1        (reference ConvectionKernels.h:242, ConvectionKernels_API.cpp:86-99)
 *     cvtt::Kernels::EncodeBC6HU/S    (reference ConvectionKernels.h:249-250, API.cpp:56-84)
 *     cvtt::Kernels::EncodeETC2[RGBA] (reference ConvectionKernels.h:253-254, API.cpp:216-229, 270-286)
 *
 * Plain pointers and sizes only.  The PODs below are byte-compatible with the reference's
 * cvtt::Options (ConvectionKernels.h:73-103, 44 bytes) and cvtt::BC7EncodingPlan
 * (ConvectionKernels.h:142-199, 808 bytes), so a reference user passes the address of the
 * structs they already have.
 *
 * Batch semantics: every encode call takes numBlocks (a multiple of 8) PixelBlocks stored
 * contiguously; group g = blocks [8g, 8g+8) is exactly what ONE call of the corresponding
 * cvtt::Kernels::Encode* consumes (ConvectionKernels.h:241), including the reference's
 * cross-lane coupling inside a group.  Output block i is bit-identical
 * to what the reference's SSE2 path writes for input block i.
 *
 * Error behaviour: the reference's entry points are void and assert on NULL
 * (ConvectionKernels_API.cpp:43-44).  Here every call returns 0 on success or a negative
 * CVTTMI_E_* code; nothing is written to the output on failure.  There is NO CPU fallback:
 * if no HIP device / kernel image is available the call fails with CVTTMI_E_NO_DEVICE.
 */
#ifndef CVTT_MI355X_H
#define CVTT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVTTMI_OK 0
#define CVTTMI_E_INVALID (-1)     /* NULL pointer, numBlocks not a multiple of 8, ... */
#define CVTTMI_E_UNSUPPORTED (-2) /* flag / format combination not implemented on the GPU path */
#define CVTTMI_E_NO_DEVICE (-3)   /* no HIP device, or device is not gfx950 */
#define CVTTMI_E_HIP (-4)         /* a HIP runtime call failed; see cvttmi_last_error() */

/* mirrors cvtt::Flags (ConvectionKernels.h:33-69) */
#define CVTTMI_FLAG_BC7_FAST_INDEXING 0x008u
#define CVTTMI_FLAG_BC7_TRY_SINGLE_COLOR 0x010u
#define CVTTMI_FLAG_BC7_RESPECT_PUNCHTHROUGH 0x020u
#define CVTTMI_FLAG_BC6H_FAST_INDEXING 0x040u
#define CVTTMI_FLAG_S3TC_EXHAUSTIVE 0x080u
#define CVTTMI_FLAG_S3TC_PARANOID 0x100u
#define CVTTMI_FLAG_UNIFORM 0x200u
#define CVTTMI_FLAG_ETC_USE_FAKE_BT709 0x400u
#define CVTTMI_FLAG_ETC_FAKE_BT709_ACCURATE 0x800u
#define CVTTMI_FLAGS_DEFAULT (CVTTMI_FLAG_BC7_FAST_INDEXING | CVTTMI_FLAG_S3TC_PARANOID)

/* byte image of cvtt::Options, ConvectionKernels.h:73-103 */
typedef struct cvttmi_options
{
    uint32_t flags;
    float threshold;
    float redWeight;
    float greenWeight;
    float blueWeight;
    float alphaWeight;
    int32_t refineRoundsBC7;
    int32_t refineRoundsBC6H;
    int32_t refineRoundsIIC;
    int32_t refineRoundsS3TC;
    int32_t seedPoints;
} cvttmi_options;

/* byte image of cvtt::BC7EncodingPlan, ConvectionKernels.h:142-199 */
typedef struct cvttmi_bc7_plan
{
    uint64_t mode1PartitionEnabled;
    uint64_t mode2PartitionEnabled;
    uint64_t mode3PartitionEnabled;
    uint16_t mode0PartitionEnabled;
    uint64_t mode7RGBAPartitionEnabled;
    uint64_t mode7RGBPartitionEnabled;
    uint8_t mode4SP[4][2];
    uint8_t mode5SP[4];
    uint8_t mode6Enabled;
    uint8_t seedPointsForShapeRGB[243];
    uint8_t seedPointsForShapeRGBA[129];
    uint8_t rgbaShapeList[129];
    uint8_t rgbaNumShapesToEvaluate;
    uint8_t rgbShapeList[243];
    uint8_t rgbNumShapesToEvaluate;
} cvttmi_bc7_plan;

/* byte image of cvtt::BC7FineTuningParams, ConvectionKernels.h:105-140: seed points (0 = off) per mode and
 * partition / rotation / index selector */
typedef struct cvttmi_bc7_fine_tuning
{
    uint8_t mode0SP[16];
    uint8_t mode1SP[64];
    uint8_t mode2SP[64];
    uint8_t mode3SP[64];
    uint8_t mode4SP[4][2];
    uint8_t mode5SP[4];
    uint8_t mode6SP;
    uint8_t mode7SP[64];
} cvttmi_bc7_fine_tuning;

typedef struct cvttmi_context cvttmi_context;

/* Default-constructed PODs (cvtt::Options(), cvtt::BC7EncodingPlan()). */
/* Build identity of this library: SHA-256 (hex) over its kernel / shim sources, public headers and compiler flags.  Not part
 * of the reference's API: measurement tooling uses it to tie rocprofv3 summaries to the library they were taken with. */
const char *cvttmi_source_sha256(void);

void cvttmi_default_options(cvttmi_options *out);
void cvttmi_default_bc7_plan(cvttmi_bc7_plan *out);
void cvttmi_default_bc7_fine_tuning(cvttmi_bc7_fine_tuning *out);

/* replace cvtt::Kernels::ConfigureBC7EncodingPlanFromQuality (ConvectionKernels_BC67.cpp:3291-3352; quality is
 * clamped to 1..100) and ConfigureBC7EncodingPlanFromFineTuningParams (BC67.cpp:3355-3483).  Host-side, once per job;
 * need no device.  The quality ladder is the reference's empirical ranking, carried as the observed plan changes per
 * quality step (csrc/bc7_quality_events.h, tools/gen_bc7_quality_events.py); plans are byte-identical to the reference's. */
int cvttmi_bc7_plan_from_quality(cvttmi_bc7_plan *plan, int quality);
int cvttmi_bc7_plan_from_fine_tuning(cvttmi_bc7_plan *plan, const cvttmi_bc7_fine_tuning *params);

/* Create a context on HIP device `device`: uploads the constant tables to HBM and
 * probes the host's RCPPS table (the reference's EndpointRefiner uses _mm_rcp_ps,
 * ConvectionKernels_EndpointRefiner.h:106, whose result is CPU-model specific; the
 * argument is always an integer 1..16, so the kernels take it as a 17-entry table and
 * "bit-exact vs. the CPU path on the same box" is well defined). */
int cvttmi_create(cvttmi_context **out, int device);
void cvttmi_destroy(cvttmi_context *ctx);
const char *cvttmi_last_error(const cvttmi_context *ctx);

/* Override / read the reciprocal table: lut[i] = rcpps(i), i = 1..16 (lut[0] ignored). */
int cvttmi_set_rcp_table(cvttmi_context *ctx, const float lut[17]);
int cvttmi_get_rcp_table(const cvttmi_context *ctx, float lut[17]);

/* Page-locked host memory.  The host-pointer entry points below (the ones the reference's own callers bind,
 * ConvectionKernels.h:242-256 take host pointers) move caller buffers that are page-locked straight over PCIe, in
 * chunks pipelined with the search; pageable buffers go through the context's staging buffers first (one extra CPU
 * copy each way).  cvttmi_host_alloc / cvttmi_host_free give such memory; cvttmi_host_register / _unregister
 * page-lock an existing allocation for as long as it is reused (registering costs about as much as one copy). */
int cvttmi_host_alloc(cvttmi_context *ctx, void **ptr, size_t bytes);
int cvttmi_host_free(cvttmi_context *ctx, void *ptr);
int cvttmi_host_register(cvttmi_context *ctx, void *ptr, size_t bytes);
int cvttmi_host_unregister(cvttmi_context *ctx, void *ptr);

/* ---- device-resident entry points: d_blocks / d_out are HBM pointers on the context's
 * device; the launch is asynchronous on `hipStream` (a hipStream_t, NULL = default).
 * Streams: a context owns ONE set of device work space (BC7 hand-over list, punch-through trial table and plan ring; up to 256 MB
 * for the trial table, see below).  The calls that use it (EncodeBC7 from half a million blocks or with BC7_RespectPunchThrough and
 * more than 2 refine rounds) are ordered by the library itself: such a call on another stream than the
 * previous one first makes its stream wait (hipStreamWaitEvent) for that call's launches; a plan slot is rewritten only
 * after every launch that read it, on whatever stream, has finished.  Calls that use no shared work space (BC1-BC5, ETC,
 * EAC, decode, tiling) are simply queued on the stream given.  A mutex serialises the host side of EVERY call on a
 * context, so any stream / thread mix is safe on one context; it is not concurrent: use one context per stream (or per
 * worker thread, the reference's caller model, etc2packer.cpp:215-281) to overlap independent jobs.  (EncodeBC6H keeps its
 * whole search state on the chip since round 4: no work buffer, no ordering between its calls.) ---- */

/* replaces cvtt::Kernels::EncodeBC7 (ConvectionKernels_API.cpp:41-54): numBlocks * 64 B
 * of PixelBlockU8 in, numBlocks * 16 B out. */
int cvttmi_encode_bc7_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                             const cvttmi_options *options, const cvttmi_bc7_plan *plan, void *hipStream);

/* replaces cvtt::Kernels::EncodeBC1 (ConvectionKernels_API.cpp:86-99 ->
 * S3TCComputer::PackRGB with alphaTest = true, ConvectionKernels_S3TC.cpp:717-1052):
 * numBlocks * 64 B of PixelBlockU8 in, numBlocks * 8 B out.  Uses options->threshold,
 * seedPoints, refineRoundsS3TC, the S3TC_Paranoid / Uniform flags and the channel weights. */
int cvttmi_encode_bc1_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                             const cvttmi_options *options, void *hipStream);

/* replaces cvtt::Kernels::EncodeBC6HU (isSigned = 0) / EncodeBC6HS (isSigned != 0)
 * (ConvectionKernels_API.cpp:56-84 -> BC6HComputer::Pack, ConvectionKernels_BC67.cpp:2665-3051):
 * numBlocks * 128 B of PixelBlockF16 (half bits as int16, RGBA, alpha ignored) in,
 * numBlocks * 16 B out.  Uses options->seedPoints, refineRoundsBC6H, the BC6H_FastIndexing /
 * Uniform flags and the channel weights.  Group g = blocks [8g, 8g+8) keeps the reference's
 * cross-lane coupling (duplicate-round skip, mode commit loop). */
int cvttmi_encode_bc6h_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                              const cvttmi_options *options, int isSigned, void *hipStream);

/* replace cvtt::Kernels::EncodeETC2 / EncodeETC2RGBA / EncodeETC2Alpha
 * (ConvectionKernels_API.cpp:216-229, 270-286, 246-256 -> ETCComputer::CompressETC2Block with
 * punchthroughAlpha = false, ConvectionKernels_ETC.cpp:1664-1887, and CompressETC2AlphaBlock,
 * 1889-2085).  numBlocks * 64 B of PixelBlockU8 in; 8 B (RGB, alpha) or 16 B (RGBA: EAC alpha
 * block then colour block) per block out.  The reference's ETC2CompressionData scratch
 * (AllocETC2Data / ReleaseETC2Data) has no counterpart: the kernels keep their scratch in
 * LDS.  These calls derive the two chroma axes of the sector split from options->{red,green,blue}Weight,
 * i.e. they behave like a caller who allocates with the options it encodes with; a caller whose two
 * Options differ uses cvttmi_encode_etc2_with_data[_device] below.
 * Uses the colour weights and the Uniform / ETC_UseFakeBT709 / ETC_FakeBT709Accurate flags. */
int cvttmi_encode_etc2_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                              const cvttmi_options *options, void *hipStream);
int cvttmi_encode_etc2_rgba_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                                   const cvttmi_options *options, void *hipStream);
int cvttmi_encode_etc2_alpha_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                                    const cvttmi_options *options, void *hipStream);

/* replaces cvtt::Kernels::EncodeETC1 (ConvectionKernels_API.cpp:201-214 -> ETCComputer::CompressETC1Block,
 * ConvectionKernels_ETC.cpp:2116-2126, CompressETC1BlockInternal 2624-2882 with both the individual and the
 * differential mode): numBlocks * 64 B of PixelBlockU8 in, 8 B per block out.  The ETC1CompressionData scratch
 * (AllocETC1Data / ReleaseETC1Data) has no counterpart.  Same flags as ETC2. */
int cvttmi_encode_etc1_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                              const cvttmi_options *options, void *hipStream);
int cvttmi_encode_etc1(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                       const cvttmi_options *options);

/* replaces cvtt::Kernels::EncodeETC2PunchthroughAlpha (ConvectionKernels_API.cpp:231-244 -> CompressETC2Block with
 * punchthroughAlpha = true, ConvectionKernels_ETC.cpp:1664-1887; EncodeVirtualTModePunchthrough 887-1264;
 * CompressETC1PunchthroughBlockInternal 2885-3082): GL_COMPRESSED_RGB8_PUNCHTHROUGH_ALPHA1_ETC2 blocks, 8 B each.
 * A pixel is transparent when alpha < floor(clamp(options->threshold, 0, 1) * 255 + 1).  Group g = blocks [8g, 8g+8)
 * keeps the reference's coupling: one transparent pixel anywhere in the group sends all eight blocks through the
 * punch-through modes as well, and the virtual-T candidate lists depend on the group's pixel counts. */
int cvttmi_encode_etc2_punchthrough_alpha_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                                                 const cvttmi_options *options, void *hipStream);
int cvttmi_encode_etc2_punchthrough_alpha(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                                          const cvttmi_options *options);

/* The reference fixes the two chroma axes when the scratch is ALLOCATED -- ETC2CompressionDataInternal's constructor
 * computes them from the Options given to AllocETC2Data (ConvectionKernels_ETC.cpp:3117-3145) -- while the error weights
 * come from the Options of each Encode call.  `allocOptions` = the Options the caller passed to AllocETC2Data (only its
 * red / green / blue weights are read; NULL = `options`).  kind: CVTTMI_ETC2_RGB = EncodeETC2, CVTTMI_ETC2_RGBA =
 * EncodeETC2RGBA, CVTTMI_ETC2_PUNCHTHROUGH = EncodeETC2PunchthroughAlpha (the three calls that take the scratch). */
#define CVTTMI_ETC2_RGB 0
#define CVTTMI_ETC2_RGBA 1
#define CVTTMI_ETC2_PUNCHTHROUGH 4
int cvttmi_encode_etc2_with_data_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks,
                                        const cvttmi_options *options, const cvttmi_options *allocOptions, int kind, void *hipStream);
int cvttmi_encode_etc2_with_data(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                                 const cvttmi_options *options, const cvttmi_options *allocOptions, int kind);

/* ---- host-buffer convenience entry points: stage through pinned memory, launch, copy
 * back, synchronise.  Same semantics as the *_device calls. ---- */
int cvttmi_encode_bc7(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                      const cvttmi_options *options, const cvttmi_bc7_plan *plan);

int cvttmi_encode_bc1(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                      const cvttmi_options *options);

int cvttmi_encode_bc6h(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                       const cvttmi_options *options, int isSigned);

int cvttmi_encode_etc2(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                       const cvttmi_options *options);
int cvttmi_encode_etc2_rgba(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                            const cvttmi_options *options);
int cvttmi_encode_etc2_alpha(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks,
                             const cvttmi_options *options);

/* Search strategy.  By default the kernels skip candidates (partitions / subsets) whose
 * rigorous error lower bound already exceeds the best candidate found so far -- an exact
 * branch-and-bound: the output is bit-identical to the exhaustive search the reference
 * performs.  exhaustive != 0 evaluates every candidate like the reference does (same output,
 * slower); the environment variable CVTTMI_EXHAUSTIVE=1 sets the default of new contexts. */
int cvttmi_set_exhaustive(cvttmi_context *ctx, int exhaustive);

/* ---- image -> PixelBlock tiling on the device (the caller side of the path: reference
 * etc2packer/etc2packer.cpp:215-247 cuts a linear image into groups of eight horizontally
 * adjacent 4x4 blocks, clamping reads at the right and bottom edges; 275-281 drops the blocks
 * that only pad the last group of a block row) ----
 * cvttmi_tiled_block_count: blocks the tiling produces = ceil(ceil(w/4)/8)*8 per block row
 *   (a multiple of 8, so every block row is made of whole groups) times ceil(h/4) rows.
 * cvttmi_tile_image_device: d_image = linear RGBA8 or RGBA16F image in HBM (rowPitchBytes
 *   between rows) -> d_blocks = PixelBlockU8 / PixelBlockF16 array of that many blocks.
 * cvttmi_compact_rows_device: packed blocks of the padded layout -> the ceil(w/4) real blocks of
 *   every row, contiguous (what a container writer stores).  A no-op copy when w % 32 == 0. */
#define CVTTMI_PIXELS_RGBA8 0
#define CVTTMI_PIXELS_RGBA16F 1
size_t cvttmi_tiled_block_count(uint32_t width, uint32_t height);
int cvttmi_tile_image_device(cvttmi_context *ctx, void *d_blocks, const void *d_image, uint32_t width, uint32_t height,
                             size_t rowPitchBytes, int pixelFormat, void *hipStream);
int cvttmi_compact_rows_device(cvttmi_context *ctx, void *d_out, const void *d_packed, uint32_t width, uint32_t height,
                               uint32_t bytesPerBlock, void *hipStream);

/* BC2 / BC3 / BC4 / BC5: cvtt::Kernels::EncodeBC2, EncodeBC3, EncodeBC4U/S, EncodeBC5U/S (reference
 * ConvectionKernels_API.cpp:101-199 -> S3TCComputer::PackRGB without alpha test, PackExplicitAlpha,
 * PackInterpolatedAlpha, ConvectionKernels_S3TC.cpp:306-715).  Input PixelBlockU8 (isSigned: PixelBlockS8, biased
 * like Util::BiasSignedInput); output 16 B per block (BC4: 8 B): BC2/BC3 = [alpha 8 B | colour 8 B], BC4 = red,
 * BC5 = [red | green].  Options used: seedPoints, refineRoundsIIC (alpha), refineRoundsS3TC + weights (colour). */
int cvttmi_encode_bc2_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream);
int cvttmi_encode_bc3_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, void *hipStream);
int cvttmi_encode_bc4_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, int isSigned, void *hipStream);
int cvttmi_encode_bc5_device(cvttmi_context *ctx, void *d_out, const void *d_blocks, size_t numBlocks, const cvttmi_options *options, int isSigned, void *hipStream);
int cvttmi_encode_bc2(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options);
int cvttmi_encode_bc3(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options);
int cvttmi_encode_bc4(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options, int isSigned);
int cvttmi_encode_bc5(cvttmi_context *ctx, uint8_t *out, const uint8_t *blocks, size_t numBlocks, const cvttmi_options *options, int isSigned);

/* EAC R11: cvtt::Kernels::EncodeETC2Alpha11 (reference ConvectionKernels_API.cpp:258-268 -> CompressEACBlock,
 * ConvectionKernels_ETC.cpp:2087-2114).  blocksS16: numBlocks x PixelBlockScalarS16 (16 int16: unsigned 0..2047,
 * signed -1023..1023, clamped like the reference) -> 8 bytes per block. */
int cvttmi_encode_etc2_alpha11_device(cvttmi_context *ctx, void *d_out, const void *d_blocksS16, size_t numBlocks, int isSigned,
                                      const cvttmi_options *options, void *hipStream);
int cvttmi_encode_etc2_alpha11(cvttmi_context *ctx, uint8_t *out, const int16_t *blocksS16, size_t numBlocks, int isSigned,
                               const cvttmi_options *options);

/* ---- decoders: cvtt::Kernels::DecodeBC7 / DecodeBC6HU / DecodeBC6HS (reference
 * ConvectionKernels_API.cpp:288-310, ConvectionKernels_BC67.cpp:2206-2423, 3058-3289), batched:
 * numBlocks (multiple of 8) packed 16-byte blocks -> PixelBlockU8 (64 B) / PixelBlockF16 (128 B,
 * half bit patterns, alpha = 1.0).  Reserved modes decode like the reference (zeros). ---- */
int cvttmi_decode_bc7_device(cvttmi_context *ctx, void *d_blocks, const void *d_bc, size_t numBlocks, void *hipStream);
int cvttmi_decode_bc7(cvttmi_context *ctx, uint8_t *blocks, const uint8_t *bc, size_t numBlocks);
int cvttmi_decode_bc6h_device(cvttmi_context *ctx, void *d_blocksF16, const void *d_bc, size_t numBlocks, int isSigned, void *hipStream);
int cvttmi_decode_bc6h(cvttmi_context *ctx, uint8_t *blocksF16, const uint8_t *bc, size_t numBlocks, int isSigned);

/* ---- one job on several devices (csrc/multi.cpp).  The north-star's "large images shard by block row across the GPUs of one
 * node", for callers of this C interface (the reference has no counterpart: its callers parallelise by threads over groups,
 * etc2packer.cpp:215-281).  Groups of 8 blocks are independent, so the search needs no exchange: the job is cut into
 * contiguous ranges of whole block rows, moved to the next group boundary where a row is no multiple of 8 blocks
 * (cvttmi_shard_block_rows -- the same rule as convectionkernels_amd/sharding.py), each range is encoded by its own context on
 * its own device from its own host thread, and the packed blocks land in `out` at the range's offset.  Output bytes equal the
 * single-device call's.  `devices` may name a device more than once (a context each).  blocksPerRow = blocks in one block row of
 * the tiled image (ceil(ceil(w/4)/8)*8, cvttmi_tiled_block_count); 0 = no rows, shard by groups.
 * cvttmi_multi_*: a handle owning one context per list entry.  A handle runs ONE job at a time (a mutex; the setters take it
 * too): concurrent callers of one handle -- including the C++ face's *Batch calls, which share the handle of the drop-in's device
 * list -- are serialised; use a handle per worker thread to overlap jobs.
 * cvttmi_encode_*_multi: stateless forms.  They keep one handle (contexts, staging) per distinct device list FOR THE LIFE OF THE
 * PROCESS -- nothing is released before exit -- and report failures through cvttmi_multi_last_error(NULL): the text of the calling
 * thread's most recent failed multi-device call.
 * No call of this section lets a C++ exception out; allocation failures come back as CVTTMI_E_HIP.
 * Between processes (one per GPU) the packed output is gathered with RCCL send/recv over xGMI instead: sharding.py,
 * bench.py --gpus N. ---- */
#define CVTTMI_FMT_BC7 0
#define CVTTMI_FMT_BC1 1
#define CVTTMI_FMT_BC6HU 2
#define CVTTMI_FMT_BC6HS 3
#define CVTTMI_FMT_ETC2_RGB 4
#define CVTTMI_FMT_ETC2_RGBA 5
typedef struct cvttmi_multi cvttmi_multi;
int cvttmi_shard_block_rows(size_t blockRows, size_t blocksPerRow, int rank, int world, size_t *firstBlock, size_t *lastBlock);
int cvttmi_multi_create(cvttmi_multi **out, const int *devices, int numDevices);
void cvttmi_multi_destroy(cvttmi_multi *m);
const char *cvttmi_multi_last_error(const cvttmi_multi *m);
int cvttmi_multi_num_devices(const cvttmi_multi *m);
cvttmi_context *cvttmi_multi_context(cvttmi_multi *m, int index);
int cvttmi_multi_last_shard(const cvttmi_multi *m, int index, size_t *firstBlock, size_t *lastBlock);
int cvttmi_multi_set_rcp_table(cvttmi_multi *m, const float lut[17]);
int cvttmi_multi_set_exhaustive(cvttmi_multi *m, int exhaustive);
/* format = CVTTMI_FMT_*; plan: BC7 only (NULL otherwise); host buffers (page-locked ones are transferred in place) */
int cvttmi_multi_encode(cvttmi_multi *m, int format, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                        const cvttmi_options *options, const cvttmi_bc7_plan *plan);
/* Device-resident callers (the tiling kernel or an upload of the caller's own has put every shard where it is searched):
 * d_shards[r] = HBM pointer ON devices[r] to the PixelBlocks of shard r, i.e. blocks [first_r, last_r) of the job as
 * cvttmi_shard_block_rows(numBlocks / blocksPerRow, blocksPerRow, r, numDevices) cuts it (NULL allowed for an empty shard);
 * d_out = numBlocks packed blocks ON devices[0].  Every device searches its shard on a stream of its own; shards on the root
 * device write their slice of d_out directly, the others write a buffer on their own device that is then copied into the slice
 * with hipMemcpyPeerAsync -- over xGMI, with peer access enabled where the pair allows it: the north-star's gather of the packed
 * output for a single process.  Returns when d_out is complete.  (CVTTMI_MULTI_FORCE_STAGE=1 in the environment when the handle is
 * created sends root-device shards through the staged route too: how a one-GPU box tests it.) */
int cvttmi_multi_encode_device(cvttmi_multi *m, int format, void *d_out, const void *const *d_shards, size_t numBlocks, size_t blocksPerRow,
                               const cvttmi_options *options, const cvttmi_bc7_plan *plan);
int cvttmi_encode_bc7_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                            const cvttmi_options *options, const cvttmi_bc7_plan *plan);
int cvttmi_encode_bc1_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                            const cvttmi_options *options);
int cvttmi_encode_bc6h_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                             const cvttmi_options *options, int isSigned);
int cvttmi_encode_etc2_rgba_multi(const int *devices, int numDevices, uint8_t *out, const uint8_t *blocks, size_t numBlocks, size_t blocksPerRow,
                                  const cvttmi_options *options);
/* The device list of the C++ face's *Batch entry points (cvtt::Kernels::EncodeBC7Batch, EncodeBC1Batch, EncodeBC6HU/SBatch,
 * EncodeETC2RGBABatch): more than one entry = calls of at least 65 536 blocks are sharded over the list as above (by groups).
 * Default: the environment variable CVTTMI_DEVICES ("0,1,2,3"), else the single device CVTTMI_DEVICE (default 0).
 * Configuration call: not to be made while another thread is inside a *Batch call (it replaces the handle those calls use). */
int cvttmi_dropin_set_devices(const int *devices, int numDevices);

/* Arithmetic self-test: evaluates binary32 divide and square root on `count` pseudo-random finite
 * operand patterns (zeros, denormals and both signs included) with the expressions and compiler
 * flags the encoders use, and counts results that differ from the host's DIVSS / SQRTSS -- the
 * arithmetic contract behind bit-exactness (reference ParallelMath.h:261-323, 959-965). */
int cvttmi_selftest_arith(cvttmi_context *ctx, uint64_t count, uint64_t seed, uint64_t *divMismatches, uint64_t *sqrtMismatches);

/* Time (ms, HIP events on the launch stream) and launch count of the kernels of the most
 * recent *_device call sequence since cvttmi_timing_reset(); used by bench.py's roofline. */
int cvttmi_timing_enable(cvttmi_context *ctx, int enable);
int cvttmi_timing_read(cvttmi_context *ctx, double *totalMs, uint64_t *launches);

#ifdef __cplusplus
}
#endif
#endif
