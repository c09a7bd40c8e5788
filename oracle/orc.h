/*
 * oracle/orc.h -- CPU restatement of the LZ4 frame/block/xxHash32 arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker.  The product (lz4_frame_conduit_amd/) never links,
 * imports or calls this code.
 *
 * Where the algorithm comes from
 * ------------------------------
 * The reference (nh2/lz4-frame-conduit) keeps no arithmetic in its own tree: the hot
 * path is the third-party dependency lz4/lz4, vendored as the git submodule `lz4/`
 * (/root/reference/.gitmodules:1-3) and compiled from lz4/lib/{lz4,lz4frame,lz4hc,
 * xxhash}.c (/root/reference/lz4-frame-conduit.cabal:48-52).  That directory is EMPTY
 * in /root/reference and the pinned commit is unrecoverable (no .git); API use needs
 * lz4 >= v1.8.2 (/root/reference/src/Codec/Compression/LZ4/CTypes.hsc:229).
 * This file therefore restates the published LZ4 Block format, LZ4 Frame format
 * (v1.6.x) and XXH32 algorithms, following the behaviour of lz4 v1.9.3 (the liblz4
 * installed in this container), and anchors parity on the reference's call sites:
 *   LZ4F_compressBegin   Conduit.hsc:292     LZ4F_compressBound  Conduit.hsc:302
 *   LZ4F_compressUpdate  Conduit.hsc:311     LZ4F_compressEnd    Conduit.hsc:321
 *   LZ4F_getFrameInfo    Conduit.hsc:579     LZ4F_decompress     Conduit.hsc:591
 * and on the inputs of the reference's own tests (test/Main.hs:60-119).
 *
 * Pinning: PINNED.  The reference holds no known-answer bytes (its tests pipe through
 * the external `lz4` CLI), so the oracle is pinned against golden vectors minted from
 * liblz4 1.9.3 -- the same upstream library the reference bundles -- by
 * oracle/mint_golden.py, committed under tests/golden/ (SURVEY.md section 8c, G1-G10).
 */
#ifndef ORC_H
#define ORC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- xxHash32 (row a5) ---------------- */
uint32_t orc_xxh32(const void* data, size_t len, uint32_t seed);
typedef struct {
    uint32_t total_len_32, large_len;
    uint32_t v[4];
    uint8_t  mem[16];
    uint32_t memsize;
} orc_xxh32_state;
void     orc_xxh32_reset(orc_xxh32_state* s, uint32_t seed);
void     orc_xxh32_update(orc_xxh32_state* s, const void* data, size_t len);
uint32_t orc_xxh32_digest(const orc_xxh32_state* s);

/* ---------------- LZ4 block codec (rows a2, a4) ---------------- */
int orc_lz4_compress_bound(int n);

/* Persistent encoder state: restates LZ4_stream_t_internal (hash table + running index). */
typedef struct {
    uint32_t table[4096];      /* byU32: 4096 x u32 ; byU16 view: 8192 x u16 over the same 16 KiB */
    uint32_t current_offset;   /* index of the first byte of the next input */
    uint32_t dict_size;        /* bytes of contiguous history in front of the next input */
} orc_lz4_stream;
void orc_lz4_stream_reset(orc_lz4_stream* s);

/* Independent block, bit-exact with LZ4_compress_default() of lz4 1.9.3 when
 * dst_cap >= orc_lz4_compress_bound(n); with a smaller dst_cap it follows the
 * limitedOutput rules (returns 0 when the budget checks fail).            */
int orc_lz4_compress_default(const uint8_t* src, uint8_t* dst, int n, int dst_cap);

/* Linked block in prefix mode (history of `s->dict_size` bytes lies directly in front
 * of src): restates LZ4_compress_fast_continue(acceleration 1) as lz4frame drives it
 * when fed <= blockSize slices (the reference's 16 KiB slicing, Conduit.hsc:464). */
int orc_lz4_compress_continue(orc_lz4_stream* s, const uint8_t* src, uint8_t* dst, int n, int dst_cap);

/* Safe decoder; `dict_size` bytes of history lie directly in front of dst
 * (LZ4_decompress_safe / _withPrefix64k / _usingDict with contiguous dictionary).
 * Returns decoded size, or a negative value on malformed input. */
int orc_lz4_decompress_safe(const uint8_t* src, uint8_t* dst, int csize, int dst_cap, size_t dict_size);

/* Statistics over a valid block: number of sequences (incl. the last literal-only one). */
int orc_lz4_count_sequences(const uint8_t* src, int csize);

/* ---------------- LZ4 frame (rows a1, a3, a6, a7) ---------------- */
typedef struct {
    uint32_t blockSizeID;          /* 0 (default=64KB), 4..7 */
    uint32_t blockMode;            /* 0 linked, 1 independent */
    uint32_t contentChecksumFlag;
    uint32_t frameType;
    uint64_t contentSize;
    uint32_t dictID;
    uint32_t blockChecksumFlag;
} orc_frame_info;                  /* 32 bytes, same layout as LZ4F_frameInfo_t */
typedef struct {
    orc_frame_info frameInfo;
    int32_t  compressionLevel;
    uint32_t autoFlush;
    uint32_t favorDecSpeed;
    uint32_t reserved[3];
} orc_prefs;                       /* 56 bytes, same layout as LZ4F_preferences_t */

/* error codes are (size_t)-(code) like LZ4F */
enum {
    ORC_OK = 0, ORC_ERR_GENERIC, ORC_ERR_maxBlockSize_invalid, ORC_ERR_blockMode_invalid,
    ORC_ERR_contentChecksumFlag_invalid, ORC_ERR_compressionLevel_invalid,
    ORC_ERR_headerVersion_wrong, ORC_ERR_blockChecksum_invalid, ORC_ERR_reservedFlag_set,
    ORC_ERR_allocation_failed, ORC_ERR_srcSize_tooLarge, ORC_ERR_dstMaxSize_tooSmall,
    ORC_ERR_frameHeader_incomplete, ORC_ERR_frameType_unknown, ORC_ERR_frameSize_wrong,
    ORC_ERR_srcPtr_wrong, ORC_ERR_decompressionFailed, ORC_ERR_headerChecksum_invalid,
    ORC_ERR_contentChecksum_invalid, ORC_ERR_frameDecoding_alreadyStarted, ORC_ERR_maxCode
};
unsigned    orc_is_error(size_t code);
const char* orc_error_name(size_t code);

size_t orc_block_size(uint32_t blockSizeID);            /* 0 on invalid id */
size_t orc_compress_bound(size_t src_size, const orc_prefs* prefs);   /* LZ4F_compressBound */
size_t orc_write_header(uint8_t* dst, size_t cap, const orc_prefs* prefs);  /* LZ4F_compressBegin bytes */

/* Streaming compressor restating LZ4F_compressBegin/Update/End buffer behaviour. */
typedef struct orc_cctx orc_cctx;
orc_cctx* orc_cctx_create(void);
void      orc_cctx_free(orc_cctx* c);
size_t    orc_compress_begin(orc_cctx* c, uint8_t* dst, size_t cap, const orc_prefs* prefs);
size_t    orc_compress_update(orc_cctx* c, uint8_t* dst, size_t cap, const uint8_t* src, size_t n);
size_t    orc_compress_end(orc_cctx* c, uint8_t* dst, size_t cap);

/* The reference's compress conduit call pattern (Conduit.hsc:457-533) as one call:
 * input is cut in `slice` byte pieces (16384 in the reference). Returns frame size or error. */
size_t orc_conduit_compress(const uint8_t* src, size_t n, const orc_prefs* prefs, size_t slice,
                            uint8_t* dst, size_t cap);

/* One-shot frame decode (first frame only, like the reference's decompress conduit).
 * *consumed = bytes of src used.  Returns decoded size or an error code. */
size_t orc_decompress_frame(const uint8_t* src, size_t n, uint8_t* dst, size_t cap,
                            size_t* consumed, orc_frame_info* info_out);

#ifdef __cplusplus
}
#endif
#endif
