/*
 * lz4f_mi355x.h -- C ABI of liblz4f_mi355x.so: the MI355X-native LZ4 frame codec that drops in
 * under Codec.Compression.LZ4.Conduit (nh2/lz4-frame-conduit).
 *
 * PART 1 is the drop-in boundary: exactly the twelve LZ4F_* entry points the reference's
 * inline-c FFI binds (citations are /root/reference/src/Codec/Compression/LZ4/Conduit.hsc),
 * with the struct layouts its Storable instances poke (CTypes.hsc:155-232).  The symbols are
 * exported under BOTH the upstream names (LZ4F_compressUpdate ...) so that swapping the cabal
 * `c-sources` for `extra-libraries: lz4f_mi355x` is the whole integration, and
 * under an lz4f_mi355x_ prefix (same functions) for processes that also load liblz4.
 * Every block is encoded / decoded / checksummed by HIP kernels on the GPU; there is no CPU
 * fallback: with no usable device the calls return ERROR_GENERIC (and lz4f_mi355x_last_error()
 * says why).
 *
 * PART 2 is the bulk extension the GPU needs (SURVEY.md section 8b "Bulk extension"): whole
 * frames / many blocks per call, host-pointer and device-pointer variants.
 */
#ifndef LZ4F_MI355X_H
#define LZ4F_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LZ4F_MI355X_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * PART 1 -- the LZ4F-compatible streaming API (rows a1, a3, a6, a7 of SURVEY.md section 8a)
 * ------------------------------------------------------------------------------------------ */

#define LZ4F_VERSION          100   /* consumed at Conduit.hsc:199, :242, :568 */
#define LZ4F_HEADER_SIZE_MIN  7
#define LZ4F_HEADER_SIZE_MAX  19    /* consumed at Conduit.hsc:367, :466 */
#define LZ4F_BLOCK_HEADER_SIZE 4
#define LZ4F_BLOCK_CHECKSUM_SIZE 4
#define LZ4F_CONTENT_CHECKSUM_SIZE 4

typedef size_t LZ4F_errorCode_t;

/* CTypes.hsc:48-66 (BlockSizeID) */
typedef enum { LZ4F_default = 0, LZ4F_max64KB = 4, LZ4F_max256KB = 5, LZ4F_max1MB = 6, LZ4F_max4MB = 7 } LZ4F_blockSizeID_t;
/* CTypes.hsc:77-91 (BlockMode) */
typedef enum { LZ4F_blockLinked = 0, LZ4F_blockIndependent } LZ4F_blockMode_t;
/* CTypes.hsc:98-112 (ContentChecksum) */
typedef enum { LZ4F_noContentChecksum = 0, LZ4F_contentChecksumEnabled } LZ4F_contentChecksum_t;
/* CTypes.hsc:117-131 (BlockChecksum) */
typedef enum { LZ4F_noBlockChecksum = 0, LZ4F_blockChecksumEnabled } LZ4F_blockChecksum_t;
/* CTypes.hsc:136-150 (FrameType) */
typedef enum { LZ4F_frame = 0, LZ4F_skippableFrame } LZ4F_frameType_t;

/* CTypes.hsc:155-199 (FrameInfo): 32 bytes; offsets 0,4,8,12,16,24,28 */
typedef struct {
    LZ4F_blockSizeID_t     blockSizeID;
    LZ4F_blockMode_t       blockMode;
    LZ4F_contentChecksum_t contentChecksumFlag;
    LZ4F_frameType_t       frameType;
    unsigned long long     contentSize;
    unsigned               dictID;
    LZ4F_blockChecksum_t   blockChecksumFlag;
} LZ4F_frameInfo_t;

/* CTypes.hsc:202-232 (Preferences): 56 bytes; offsets 0,32,36,40,44 */
typedef struct {
    LZ4F_frameInfo_t frameInfo;
    int      compressionLevel;   /* only level <= 2 ("fast") exists here; the reference pins 0 (Conduit.hsc:260) */
    unsigned autoFlush;
    unsigned favorDecSpeed;
    unsigned reserved[3];
} LZ4F_preferences_t;

typedef struct { unsigned stableSrc; unsigned reserved[3]; } LZ4F_compressOptions_t;
typedef struct { unsigned stableDst; unsigned reserved[3]; } LZ4F_decompressOptions_t;

typedef struct LZ4F_cctx_s LZ4F_cctx;   /* CTypes.hsc:235 */
typedef struct LZ4F_dctx_s LZ4F_dctx;   /* CTypes.hsc:236 */

/* Error enum order == upstream lz4frame.h (names surface verbatim as "lz4frame error: <name>",
 * Conduit.hsc:160) */
typedef enum {
    LZ4F_OK_NoError = 0, LZ4F_ERROR_GENERIC, LZ4F_ERROR_maxBlockSize_invalid, LZ4F_ERROR_blockMode_invalid,
    LZ4F_ERROR_contentChecksumFlag_invalid, LZ4F_ERROR_compressionLevel_invalid, LZ4F_ERROR_headerVersion_wrong,
    LZ4F_ERROR_blockChecksum_invalid, LZ4F_ERROR_reservedFlag_set, LZ4F_ERROR_allocation_failed,
    LZ4F_ERROR_srcSize_tooLarge, LZ4F_ERROR_dstMaxSize_tooSmall, LZ4F_ERROR_frameHeader_incomplete,
    LZ4F_ERROR_frameType_unknown, LZ4F_ERROR_frameSize_wrong, LZ4F_ERROR_srcPtr_wrong,
    LZ4F_ERROR_decompressionFailed, LZ4F_ERROR_headerChecksum_invalid, LZ4F_ERROR_contentChecksum_invalid,
    LZ4F_ERROR_frameDecoding_alreadyStarted, LZ4F_ERROR_maxCode
} LZ4F_errorCodes;

/* replaces LZ4F_isError / LZ4F_getErrorName -- Conduit.hsc:149-153 (unsafe FFI call: never blocks) */
LZ4F_MI355X_API unsigned    LZ4F_isError(LZ4F_errorCode_t code);
LZ4F_MI355X_API const char* LZ4F_getErrorName(LZ4F_errorCode_t code);
LZ4F_MI355X_API unsigned    LZ4F_getVersion(void);

/* replaces LZ4F_createCompressionContext -- Conduit.hsc:199, :242 */
LZ4F_MI355X_API LZ4F_errorCode_t LZ4F_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version);
/* replaces LZ4F_freeCompressionContext -- Conduit.hsc:181, :210 (NULL is accepted) */
LZ4F_MI355X_API LZ4F_errorCode_t LZ4F_freeCompressionContext(LZ4F_cctx* cctx);
/* replaces LZ4F_compressBegin -- Conduit.hsc:292 */
LZ4F_MI355X_API size_t LZ4F_compressBegin(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_preferences_t* prefsPtr);
/* replaces LZ4F_compressBound -- Conduit.hsc:302 */
LZ4F_MI355X_API size_t LZ4F_compressBound(size_t srcSize, const LZ4F_preferences_t* prefsPtr);
/* replaces LZ4F_compressUpdate -- Conduit.hsc:311 (cOptPtr is always NULL there) */
LZ4F_MI355X_API size_t LZ4F_compressUpdate(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity,
                                           const void* srcBuffer, size_t srcSize, const LZ4F_compressOptions_t* cOptPtr);
/* LZ4F_flush: not bound by the reference; LZ4F_compressEnd is defined in terms of it */
LZ4F_MI355X_API size_t LZ4F_flush(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr);
/* replaces LZ4F_compressEnd -- Conduit.hsc:321 */
LZ4F_MI355X_API size_t LZ4F_compressEnd(LZ4F_cctx* cctx, void* dstBuffer, size_t dstCapacity, const LZ4F_compressOptions_t* cOptPtr);

/* replaces LZ4F_createDecompressionContext -- Conduit.hsc:568 */
LZ4F_MI355X_API LZ4F_errorCode_t LZ4F_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version);
/* replaces LZ4F_freeDecompressionContext -- Conduit.hsc:545 (NULL is accepted) */
LZ4F_MI355X_API LZ4F_errorCode_t LZ4F_freeDecompressionContext(LZ4F_dctx* dctx);
LZ4F_MI355X_API void   LZ4F_resetDecompressionContext(LZ4F_dctx* dctx);
LZ4F_MI355X_API size_t LZ4F_headerSize(const void* src, size_t srcSize);
/* replaces LZ4F_getFrameInfo -- Conduit.hsc:579 */
LZ4F_MI355X_API size_t LZ4F_getFrameInfo(LZ4F_dctx* dctx, LZ4F_frameInfo_t* frameInfoPtr, const void* srcBuffer, size_t* srcSizePtr);
/* replaces LZ4F_decompress -- Conduit.hsc:591 (dOptPtr is always NULL there) */
LZ4F_MI355X_API size_t LZ4F_decompress(LZ4F_dctx* dctx, void* dstBuffer, size_t* dstSizePtr,
                                       const void* srcBuffer, size_t* srcSizePtr, const LZ4F_decompressOptions_t* dOptPtr);

/* the same twelve (+3) under a private prefix, for processes that also map liblz4 */
LZ4F_MI355X_API unsigned    lz4f_mi355x_isError(size_t code);
LZ4F_MI355X_API const char* lz4f_mi355x_getErrorName(size_t code);
LZ4F_MI355X_API size_t lz4f_mi355x_createCompressionContext(LZ4F_cctx** cctxPtr, unsigned version);
LZ4F_MI355X_API size_t lz4f_mi355x_freeCompressionContext(LZ4F_cctx* cctx);
LZ4F_MI355X_API size_t lz4f_mi355x_compressBegin(LZ4F_cctx* cctx, void* dst, size_t cap, const LZ4F_preferences_t* prefs);
LZ4F_MI355X_API size_t lz4f_mi355x_compressBound(size_t srcSize, const LZ4F_preferences_t* prefs);
LZ4F_MI355X_API size_t lz4f_mi355x_compressUpdate(LZ4F_cctx* cctx, void* dst, size_t cap, const void* src, size_t n, const LZ4F_compressOptions_t* o);
LZ4F_MI355X_API size_t lz4f_mi355x_flush(LZ4F_cctx* cctx, void* dst, size_t cap, const LZ4F_compressOptions_t* o);
LZ4F_MI355X_API size_t lz4f_mi355x_compressEnd(LZ4F_cctx* cctx, void* dst, size_t cap, const LZ4F_compressOptions_t* o);
LZ4F_MI355X_API size_t lz4f_mi355x_createDecompressionContext(LZ4F_dctx** dctxPtr, unsigned version);
LZ4F_MI355X_API size_t lz4f_mi355x_freeDecompressionContext(LZ4F_dctx* dctx);
LZ4F_MI355X_API size_t lz4f_mi355x_getFrameInfo(LZ4F_dctx* dctx, LZ4F_frameInfo_t* fi, const void* src, size_t* srcSize);
LZ4F_MI355X_API size_t lz4f_mi355x_decompress(LZ4F_dctx* dctx, void* dst, size_t* dstSize, const void* src, size_t* srcSize, const LZ4F_decompressOptions_t* o);

/* The two finalizers the reference defines verbatim in C (Conduit.hsc:163-189, :539-553) and
 * takes the address of (Conduit.hsc:191, :555); shipped here so a binding that does not use
 * inline-c still finds them. */
LZ4F_MI355X_API void haskell_lz4_freeCompressionContext(LZ4F_cctx** ctxPtr);
LZ4F_MI355X_API void haskell_lz4_freeDecompressionContext(LZ4F_dctx** ctxPtr);

/* ------------------------------------------------------------------------------------------
 * PART 2 -- bulk extension (new; what the batched conduits call).  All functions return an
 * LZ4F-style size_t (test with LZ4F_isError).
 * ------------------------------------------------------------------------------------------ */

/* why the last failing call on this thread failed (HIP error text etc.); never NULL */
LZ4F_MI355X_API const char* lz4f_mi355x_last_error(void);
/* number of usable HIP devices (0 when there is none: every compute entry then fails loudly) */
LZ4F_MI355X_API int lz4f_mi355x_device_count(void);
/* which device the calling thread's contexts and engines are created on (default: 0, or the
 * LZ4F_MI355X_DEVICE environment variable) */
LZ4F_MI355X_API size_t lz4f_mi355x_set_device(int device);
/* The host-pointer entry points (the twelve LZ4F_* functions, compressFrame / decompressFrame, the conduits) borrow an engine
 * - a HIP stream, pinned staging, device workspace - from a process-wide pool for the duration of a call and give it back;
 * nothing is owned by a thread.  This frees the engines that are idle right now (the pool refills on demand). */
LZ4F_MI355X_API void lz4f_mi355x_release_engines(void);

/* Worst-case size of a whole frame for srcSize bytes (header + blocks + EndMark + checksum). */
LZ4F_MI355X_API size_t lz4f_mi355x_compressFrameBound(size_t srcSize, const LZ4F_preferences_t* prefs);

/* Host-pointer bulk calls: one complete frame per call.  The frame's blocks go to the GPU in slabs of consecutive blocks (64 MiB),
 * several slabs in flight - two engines per device, each on a host thread of its own - so that uploads, kernels and downloads
 * overlap; page-locked buffers (lz4f_mi355x_host_alloc) are read and written by the DMA engines directly, pageable ones through
 * pinned staging.  compressFrame == the bytes the streaming API would produce for the same input fed in whole blocks (up to the
 * match finder's choice of matches, which is not deterministic). */
LZ4F_MI355X_API size_t lz4f_mi355x_compressFrame(void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                                 const LZ4F_preferences_t* prefs);
/* BLOCK LIST of a finished frame held in host memory - a frame of this library's host paths (compressFrame, the LZ4F_* streaming
 * calls, the conduits) or of any other LZ4 encoder (an archive made by liblz4 years ago).  appendBlockList walks the frame's size words
 * once on the host (a 4-byte read per block, no GPU) and writes, at buf + frameSize, the skippable frame lz4f_mi355x_dev_decompressFrame
 * looks for at a stream's end: where every block's size word is.  The device decoder then checks those positions link by link in parallel
 * instead of walking them - a walk is one dependent read per block, 0.12 ms for 256 blocks of 4 MiB and 10x that for 64 KiB blocks - and
 * every other LZ4 reader skips the extra frame as the format says.  frameSize must be exactly the frame (header .. EndMark / content
 * checksum); returns frameSize + the bytes added (nothing is added to a frame without blocks), or an error code
 * (dstMaxSize_tooSmall: capacity; frameSize_wrong: the frame does not end at frameSize).  blockListSize: the bytes appendBlockList would add.
 * The frame must start 16-byte aligned in DEVICE memory when it is decoded for the list to be looked at (it is a hint: without it the
 * walk happens as before). */
LZ4F_MI355X_API size_t lz4f_mi355x_blockListSize(const void* frame, size_t frameSize);
LZ4F_MI355X_API size_t lz4f_mi355x_appendBlockList(void* buf, size_t frameSize, size_t capacity);
/* Decodes the first frame found in src. *srcConsumed (optional) = bytes of src used. */
LZ4F_MI355X_API size_t lz4f_mi355x_decompressFrame(void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                                                   size_t* srcConsumed);
/* The same without an output buffer of the caller's: `yield` gets the decoded bytes slab by slab, in order, on the calling
 * thread (each slab is valid during the call only).  Memory is bounded by the slabs in flight whatever the frame claims.
 * Returns the decoded size. */
typedef void (*lz4f_mi355x_yield_fn)(void* user, const void* data, size_t size);
LZ4F_MI355X_API size_t lz4f_mi355x_decompressFrameTo(lz4f_mi355x_yield_fn yield, void* user, const void* src, size_t srcSize,
                                                     size_t* srcConsumed);
/* A frame decoded BATCH BY BATCH, for callers that receive a stream and must not hold all of it (the bounded decompressBatched conduit;
 * the reference's `decompress` holds one max(hint, 16 KiB) buffer, Conduit.hsc:634-659).  The caller walks the size words over what arrives:
 *   fdec_create   parses a complete frame header (LZ4F_headerSize says how many bytes that is); returns its size, fills *info (may be NULL)
 *   fdec_blocks   a run of WHOLE blocks - [u32 size word | payload | u32 checksum if the header says so]* - is decoded through the bulk path
 *                 (slabs of blocks in flight over the devices lz4f_mi355x_use_devices named) and handed to `yield` in order; returns the decoded
 *                 bytes.  Linked frames: the last 64 KiB handed over are kept inside for the next run.  Memory: the run + the slabs in flight.
 *   fdec_end      `tail` = the EndMark and, if the header asks for one, the content checksum behind it: verifies contentSize and the checksum
 *                 over everything the runs produced; returns the bytes of tail consumed (4 or 8)
 *   fdec_free     always, also after an error */
typedef struct lz4f_mi355x_fdec lz4f_mi355x_fdec;
LZ4F_MI355X_API size_t lz4f_mi355x_fdec_create(lz4f_mi355x_fdec** out, const void* header, size_t headerBytes, LZ4F_frameInfo_t* info);
LZ4F_MI355X_API size_t lz4f_mi355x_fdec_blocks(lz4f_mi355x_fdec* d, lz4f_mi355x_yield_fn yield, void* user, const void* blocks, size_t blocksBytes);
LZ4F_MI355X_API size_t lz4f_mi355x_fdec_end(lz4f_mi355x_fdec* d, const void* tail, size_t tailBytes);
LZ4F_MI355X_API void   lz4f_mi355x_fdec_free(lz4f_mi355x_fdec* d);

/* How many GPUs the bulk calls above deal their slabs over (round-robin, starting at the calling thread's device; default 1).
 * Blocks of an independent-block frame need nothing from each other: no collective, the host puts the slabs' output in order.
 * Process-wide.  count must be <= the visible devices (test switch: with LZ4F_MI355X_LOGICAL_DEVICES=n in the environment up to
 * n "logical" devices are accepted and mapped onto the visible ones, d mod visible - the dealing code runs as on n GPUs). */
LZ4F_MI355X_API size_t lz4f_mi355x_use_devices(int count);
/* Page-locked host memory for the bulk calls' src / dst (what the batched conduits gather their chunks in). */
LZ4F_MI355X_API void*  lz4f_mi355x_host_alloc(size_t size);
LZ4F_MI355X_API void   lz4f_mi355x_host_free(void* p);

/* ---- device-resident engine: everything stays in HBM, nothing synchronises with the host ---- */
typedef struct lz4f_mi355x_engine lz4f_mi355x_engine;

/* borrowStream == 0: the engine creates (and owns) a non-blocking stream on `device`; hipStream is ignored.
 * borrowStream != 0: all work is enqueued on the caller's hipStream_t `hipStream` (NULL = HIP's default stream). */
LZ4F_MI355X_API size_t lz4f_mi355x_engine_create(lz4f_mi355x_engine** out, int device, void* hipStream, int borrowStream);
LZ4F_MI355X_API size_t lz4f_mi355x_engine_free(lz4f_mi355x_engine* e);
LZ4F_MI355X_API void*  lz4f_mi355x_engine_stream(lz4f_mi355x_engine* e);

/* DETERMINISM.  liblz4 maps equal input to equal bytes.  This encoder, by default, does not: the sixteen waves of a workgroup search
 * neighbouring slices of a 64 KiB tile at once and share one hash table, so which earlier occurrence a position finds depends on
 * which wave got there first.  Every frame is a valid LZ4 frame that decodes to the input, and the size varies by ~1e-5 between
 * runs (4 GiB of the bench input: 2 189 936 735 .. 2 189 966 976 bytes) - but two compressions of the same input are in general
 * NOT byte-identical.  Where that matters (reproducible archives, deduplication, content-addressed stores) switch the engine to
 * the deterministic search (round 4: csrc/encode_solo.cuh): one wave per 64 KiB chunk with a hash table of its own - nothing shared
 * between waves, so the records are a function of the input alone; equal input then gives equal bytes, on the bench input at 1.4x
 * the default match finder's time and the same ratio (text: 6x the time, 2 % of the ratio; until round 4: one wave per workgroup of the shared search
 * parsing in order, 10x), with the worst-case record workspace, 2 bytes per input byte.  Engines of the host-pointer calls
 * and of the LZ4F_* streaming functions read LZ4F_MI355X_DETERMINISTIC=1 from the environment when they are made.
 * Decoding is deterministic always. */
LZ4F_MI355X_API size_t lz4f_mi355x_engine_set_deterministic(lz4f_mi355x_engine* e, int enable);

/* Optional per-kernel timing with HIP events recorded on the engine's stream around each kernel of the last
 * compress / decompress call.  get_timing synchronises the stream and fills ms[] (milliseconds):
 *   [0] find_matches  [1] layout  [2] emit  [3] xxh32 (compress)  [4] walk  [5] xxh32 (verify)  [6] decode (all kernels)
 *   [7] finish  [8] decode: parse kernel  [9] decode: copy kernel
 *   [10] the whole compress call  [11] the whole decompress call (first kernel's start to last kernel's end on the engine's stream:
 *        the block-checksum verification runs beside the decode kernels on a stream of the engine's own, so [5] + [6] > [11])
 * entries of kernels that did not run are 0.
 * get_timing_n fills the first min(n, LZ4F_MI355X_TIMING_SLOTS) slots of a float[n] - the call to use: the slot count has grown
 * (10 in rounds 1-2, 12 since round 3) and may grow again.  get_timing (no capacity argument) keeps its ORIGINAL contract and
 * writes exactly LZ4F_MI355X_TIMING_SLOTS_V1 = 10 floats, so a caller built against an older header is never overrun. */
#define LZ4F_MI355X_TIMING_SLOTS 12
#define LZ4F_MI355X_TIMING_SLOTS_V1 10
LZ4F_MI355X_API size_t lz4f_mi355x_engine_set_timing(lz4f_mi355x_engine* e, int enable);
LZ4F_MI355X_API size_t lz4f_mi355x_engine_get_timing(lz4f_mi355x_engine* e, float* ms);
LZ4F_MI355X_API size_t lz4f_mi355x_engine_get_timing_n(lz4f_mi355x_engine* e, float* ms, size_t n);

/* result record the device writes; read it back after synchronising the stream */
typedef struct {
    uint64_t size;        /* compress: frame bytes written; decompress: decoded bytes */
    uint64_t consumed;    /* decompress: frame bytes consumed (incl. EndMark / checksum) */
    uint32_t status;      /* LZ4F_errorCodes value, 0 = ok */
    uint32_t n_blocks;
    uint32_t first_bad_block;
    uint32_t flags;       /* decoded FLG byte (decompress) */
} lz4f_mi355x_result;

/* Block table entry: where block i lives in the frame and in the output. */
typedef struct {
    uint64_t src_off;     /* offset of the payload (after the 4-byte size word) in the frame */
    uint64_t dst_off;     /* offset of the decoded block in the output */
    uint32_t word;        /* the size word as stored: bit31 = stored raw, bits 30..0 = payload bytes */
    uint32_t dst_size;    /* decoded bytes (filled by decode; blockSize-capacity on input) */
} lz4f_mi355x_block;

/* Bytes of device workspace the engine will hold for inputs up to srcSize (informational). */
LZ4F_MI355X_API size_t lz4f_mi355x_dev_workspace_size(size_t srcSize, const LZ4F_preferences_t* prefs);

/* d_src[0..srcSize) -> one LZ4 frame at d_dst (device pointers).  Asynchronous on the engine's
 * stream.  d_result (device, optional) receives size/status; d_table (device, optional,
 * >= srcSize/blockSize+1 entries) receives the block table, which dev_decompressBlocks accepts
 * back to skip the serial walk over the size words.
 * Content checksum (prefs->frameInfo.contentChecksumFlag): XXH32 over the whole input is one
 * dependent chain, so ONE wave computes it (k_xxh32_content: the four accumulators as four
 * lanes) at ~1.4 GB/s - far below the codec; the word lands behind the EndMark as liblz4's does.  */
LZ4F_MI355X_API size_t lz4f_mi355x_dev_compressFrame(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity,
                                                     const void* d_src, size_t srcSize, const LZ4F_preferences_t* prefs,
                                                     lz4f_mi355x_result* d_result, lz4f_mi355x_block* d_table);

/* First frame at d_frame[0..frameCapacity) -> d_dst.  Peeks the 7..19 header bytes (one small
 * device->host copy), then everything is asynchronous: a walk kernel chases the size words, block
 * checksums are verified on the GPU, blocks are decoded.  A content checksum present in the
 * frame is verified behind the decode (same single wave as above; a mismatch gives
 * ERROR_contentChecksum_invalid in result.status); an engine made with
 * LZ4F_MI355X_NO_CONTENT_CHECK set in the environment skips that, as liblz4 >= 1.9.4 can
 * (LZ4F_decompressOptions_t.skipChecksums).                                                   */
LZ4F_MI355X_API size_t lz4f_mi355x_dev_decompressFrame(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity,
                                                       const void* d_frame, size_t frameCapacity,
                                                       lz4f_mi355x_result* d_result);

/* Table-driven variant: the caller already has the block table (from dev_compressFrame, or from
 * walking the size words on the host), so there is no walk and no host synchronisation at all.
 * `info` carries blockSizeID / blockMode / blockChecksumFlag of the frame.                     */
LZ4F_MI355X_API size_t lz4f_mi355x_dev_decompressBlocks(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity,
                                                        const void* d_frame, size_t frameCapacity,
                                                        const lz4f_mi355x_block* d_table, uint32_t n_blocks,
                                                        const LZ4F_frameInfo_t* info, lz4f_mi355x_result* d_result);

/* Sequence index (optional side channel between this library's own compress and decompress; the frame itself stays a
 * plain LZ4 frame and decodes without it).  While it writes a block's payload the compressor notes, every 16 sequences,
 * where the token sits in the payload and which output position the sequence starts at; with that table the decoder
 * parses every 16 sequences on their own lane, finds the matches whose bytes sit in the payload itself, and copies in
 * parallel instead of walking one dependent chain per 4 MiB block (replaces the same inner loop of LZ4F_decompress,
 * Conduit.hsc:591).  The decoder still parses the payload itself -- the index only says where it may start -- and checks
 * that the pieces join up: a missing, stale or foreign index makes it fall back to the generic decoder, it cannot change
 * the bytes that come out.  Independent blocks only.  A stream with more sequences than the index has room for (about one
 * per 64 input bytes at the recommended size) gets an index marked unusable. */
LZ4F_MI355X_API size_t lz4f_mi355x_dev_index_size(size_t srcSize, const LZ4F_preferences_t* prefs);       /* recommended bytes for d_index */
/* IN-BAND: with d_index == NULL and indexCapacity == LZ4F_MI355X_INBAND (d_table and d_result may then be NULL too) the index
 * and the list of the blocks' positions travel in the byte stream itself, as a skippable frame (magic 0x184D2A5E) right behind
 * the LZ4 frame: result.size includes it, d_dst must be 16-byte aligned and hold compressFrameBound + trailer_bound bytes.
 * liblz4, the `lz4` tool and the reference's `decompress` (Conduit.hsc:598: it stops at the EndMark) decode such a stream to the
 * same bytes; lz4f_mi355x_dev_decompressFrame finds the trailer from the stream's last 32 bytes and, after checking it against
 * the frame itself, skips the walk over the size words and parses with the index. */
#define LZ4F_MI355X_INBAND ((size_t)-1)
/* result.flags: bits 0..7 the frame's FLG byte, bit 8 a skippable frame was skipped; bits 12.. say which way a decompress call
 * went - a pure function of the call's arguments, the frame's header and trailer and the switches the engine was made with,
 * never of earlier calls (tests/test_gpu_parity.py: test_decode_path_by_input_class) */
#define LZ4F_MI355X_PATH_TABLE_GIVEN   0x001u   /* the caller's block table: no walk */
#define LZ4F_MI355X_PATH_TRAILER       0x002u   /* the size words' positions came from the frame's trailer (checked link by link) */
#define LZ4F_MI355X_PATH_PARALLEL_WALK 0x004u   /* the size words were looked for in parallel (small blocks) */
#define LZ4F_MI355X_PATH_INDEXED       0x008u   /* sequence index: parse per entry, direct matches, copier workgroups */
#define LZ4F_MI355X_PATH_SELF_INDEX    0x010u   /* linked frame without an index: the decoder made one */
#define LZ4F_MI355X_PATH_DOUBLING      0x020u   /* dense frame: pointer doubling was set up (the device decides whether it runs) */
#define LZ4F_MI355X_PATH_TRACE_HOPS    0x040u   /* dense frame: the hop-by-hop tracer was set up */
#define LZ4F_MI355X_PATH_WINDOW        0x080u   /* linked frame: the single-workgroup window kernel was launched (it returns at once behind a successful indexed decode) */
#define LZ4F_MI355X_PATH_FUSED         0x100u   /* fused parse+copy workgroups were launched (alone, or as what the others fall back to) */
#define LZ4F_MI355X_PATH_WAVE_PER_BLOCK 0x200u  /* small independent blocks: a wave per block */
#define LZ4F_MI355X_PATH_WORKGROUP_PER_BLOCK 0x800u /* few big independent blocks that may be dense: a workgroup per block with the window in LDS was launched beside the fused ones (the payload's density decides on the device which of the two decodes) */
#define LZ4F_MI355X_PATH_INDEX_DROPPED 0x400u   /* set on the device: the indexed kernels refused the index, the generic ones decoded */
/* compress calls, result.flags bit 9: the encoder's record workspace (sized for a sequence per 5.3 input bytes on average - dense text has
 * one per 6..8 - instead of the format's worst case of one per 4: lz4f_mi355x_dev_workspace_size) was used up, and the 64 KiB tiles that
 * found it empty went out as literals.  The frame is valid and decodes to the input, it is only bigger than it could be.
 * LZ4F_MI355X_RECS_PER_TILE (1..16385 records per 64 KiB of input, read when an engine is made; default 12288 = 1.5 bytes of workspace
 * per input byte, 16385 = worst case for every tile, 1024 = 0.13 bytes per byte for data known to be sparse in matches) sizes it. */
#define LZ4F_MI355X_ENC_POOL_SHORT     0x200u
LZ4F_MI355X_API size_t lz4f_mi355x_trailer_bound(size_t srcSize, const LZ4F_preferences_t* prefs);
LZ4F_MI355X_API size_t lz4f_mi355x_dev_compressFrameIndexed(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize,
                                                            const LZ4F_preferences_t* prefs, lz4f_mi355x_result* d_result,
                                                            lz4f_mi355x_block* d_table, void* d_index, size_t indexCapacity);
LZ4F_MI355X_API size_t lz4f_mi355x_dev_decompressBlocksIndexed(lz4f_mi355x_engine* e, void* d_dst, size_t dstCapacity, const void* d_frame,
                                                               size_t frameCapacity, const lz4f_mi355x_block* d_table, uint32_t n_blocks,
                                                               const LZ4F_frameInfo_t* info, const void* d_index, size_t indexSize,
                                                               lz4f_mi355x_result* d_result);

/* Per-block XXH32 of n_blocks byte ranges: d_out[i] = XXH32(d_base + off[i], len[i], 0) (row a5). */
LZ4F_MI355X_API size_t lz4f_mi355x_dev_xxh32(lz4f_mi355x_engine* e, const void* d_base, const uint64_t* d_off,
                                             const uint32_t* d_len, uint32_t n_blocks, uint32_t* d_out);

/* ---- C++ mirror of the reference's conduits, driven through callbacks (host side above the
 *      C ABI; see lz4_frame_conduit_amd/csrc/conduit.hpp).  `await` returns the next input chunk
 *      (size 0 at end of stream: sets *data = NULL); `yield` receives each output ByteString. ---- */
typedef size_t (*lz4f_mi355x_await_fn)(void* user, const void** data);
/* compress = compressWithOutBufferSize 0 (Conduit.hsc:336-337, :457-533); prefs NULL = lz4DefaultPreferences */
LZ4F_MI355X_API int lz4f_mi355x_conduit_compress(size_t outBufferSize, const LZ4F_preferences_t* prefs,
                                                 lz4f_mi355x_await_fn await, lz4f_mi355x_yield_fn yield, void* user,
                                                 char* errbuf, size_t errcap);
/* compressYieldImmediately (Conduit.hsc:364-425) */
LZ4F_MI355X_API int lz4f_mi355x_conduit_compress_yield_immediately(const LZ4F_preferences_t* prefs,
                                                 lz4f_mi355x_await_fn await, lz4f_mi355x_yield_fn yield, void* user,
                                                 char* errbuf, size_t errcap);
/* decompress (Conduit.hsc:598-701) */
LZ4F_MI355X_API int lz4f_mi355x_conduit_decompress(lz4f_mi355x_await_fn await, lz4f_mi355x_yield_fn yield, void* user,
                                                 char* errbuf, size_t errcap);
/* batched conduits (new): gather >= batchBytes of input per GPU call */
LZ4F_MI355X_API int lz4f_mi355x_conduit_compress_batched(size_t batchBytes, const LZ4F_preferences_t* prefs,
                                                 lz4f_mi355x_await_fn await, lz4f_mi355x_yield_fn yield, void* user,
                                                 char* errbuf, size_t errcap);
/* the same, and behind the frame its BLOCK LIST as a skippable frame (see lz4f_mi355x_appendBlockList): `mi355x-lz4c --index` */
LZ4F_MI355X_API int lz4f_mi355x_conduit_compress_batched_listed(size_t batchBytes, const LZ4F_preferences_t* prefs,
                                                 lz4f_mi355x_await_fn await, lz4f_mi355x_yield_fn yield, void* user,
                                                 char* errbuf, size_t errcap);
LZ4F_MI355X_API int lz4f_mi355x_conduit_decompress_batched_bounded(size_t batchBytes, lz4f_mi355x_await_fn await, lz4f_mi355x_yield_fn yield, void* user,
                                                                  char* errbuf, size_t errcap);
LZ4F_MI355X_API int lz4f_mi355x_conduit_decompress_batched(lz4f_mi355x_await_fn await, lz4f_mi355x_yield_fn yield, void* user,
                                                 char* errbuf, size_t errcap);

#ifdef __cplusplus
}
#endif
#endif /* LZ4F_MI355X_H */
