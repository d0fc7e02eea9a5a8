/*
 * oracle/zso_common.h — shared constants for the CPU oracle ("zso" = zstd oracle).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked into, imported by
 * or called from the product path (zstdsharp_amd/, include/).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and
 * there only as the checker / the timed CPU baseline.
 *
 * The oracle is a plain-C restatement of the algorithms in the reference
 * (ZstdSharp, a C# port of zstd v1.5.1).  Citations use the SURVEY.md
 * shorthand: U/ = /root/reference/src/ZstdSharp/Unsafe/.
 *
 * Pinning (see DESIGN.md "Oracle"): the reference cannot be built here (C#,
 * no dotnet/mono) and holds no golden compressed vectors; the oracle is
 * pinned by (1) the reference tests' known answers (T/ZstdNetTests.cs,
 * T/ZstdNetSteamingTests.cs: header bytes, checksum length, error codes,
 * round trips on the deterministic fixtures) and (2) committed .zst fixtures
 * produced in the authoring container by libzstd 1.4.8/1.4.9/1.5.7 (the codec
 * the reference's own T/ZstdTest.cs:69-90 treats as byte-equivalent), with
 * the generating script in tests/golden/make_golden.py.
 */
#ifndef ZSO_COMMON_H
#define ZSO_COMMON_H

#include <stddef.h>
#include <stdint.h>
#include <string.h>

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t  s16;

/* error codes: U/ZSTD_ErrorCode.cs ; error <=> value > (size_t)-120 (U/ErrorPrivate.cs:10-13) */
enum {
    ZSO_error_GENERIC = 1,
    ZSO_error_prefix_unknown = 10,
    ZSO_error_version_unsupported = 12,
    ZSO_error_frameParameter_unsupported = 14,
    ZSO_error_frameParameter_windowTooLarge = 16,
    ZSO_error_corruption_detected = 20,
    ZSO_error_checksum_wrong = 22,
    ZSO_error_dictionary_corrupted = 30,
    ZSO_error_dictionary_wrong = 32,
    ZSO_error_parameter_unsupported = 40,
    ZSO_error_parameter_outOfBound = 42,
    ZSO_error_tableLog_tooLarge = 44,
    ZSO_error_maxSymbolValue_tooLarge = 46,
    ZSO_error_maxSymbolValue_tooSmall = 48,
    ZSO_error_stage_wrong = 60,
    ZSO_error_memory_allocation = 64,
    ZSO_error_workSpace_tooSmall = 66,
    ZSO_error_dstSize_tooSmall = 70,
    ZSO_error_srcSize_wrong = 72,
    ZSO_error_dstBuffer_null = 74,
    ZSO_error_maxCode = 120
};
#define ZSO_ERR(name) ((size_t)0 - (size_t)ZSO_error_##name)
static inline int zso_isError(size_t c) { return c > (size_t)0 - (size_t)ZSO_error_maxCode; }

#define ZSO_MAGIC            0xFD2FB528u
#define ZSO_MAGIC_SKIPPABLE  0x184D2A50u
#define ZSO_BLOCKSIZE_MAX    (1u << 17)
#define ZSO_CONTENTSIZE_UNKNOWN ((u64)0 - 1)
#define ZSO_CONTENTSIZE_ERROR   ((u64)0 - 2)

#define ZSO_MaxLL 35
#define ZSO_MaxML 52
#define ZSO_MaxOff 31
#define ZSO_LLFSELog 9
#define ZSO_MLFSELog 9
#define ZSO_OffFSELog 8
#define ZSO_LL_DEFAULTNORMLOG 6
#define ZSO_ML_DEFAULTNORMLOG 6
#define ZSO_OF_DEFAULTNORMLOG 5

static inline u32 zso_readLE16(const void* p) { const u8* b = (const u8*)p; return (u32)b[0] | ((u32)b[1] << 8); }
static inline u32 zso_readLE24(const void* p) { const u8* b = (const u8*)p; return (u32)b[0] | ((u32)b[1] << 8) | ((u32)b[2] << 16); }
static inline u32 zso_readLE32(const void* p) { u32 v; memcpy(&v, p, 4); return v; }
static inline u64 zso_readLE64(const void* p) { u64 v; memcpy(&v, p, 8); return v; }
static inline void zso_writeLE16(void* p, u32 v) { u8* b = (u8*)p; b[0] = (u8)v; b[1] = (u8)(v >> 8); }
static inline void zso_writeLE24(void* p, u32 v) { u8* b = (u8*)p; b[0] = (u8)v; b[1] = (u8)(v >> 8); b[2] = (u8)(v >> 16); }
static inline void zso_writeLE32(void* p, u32 v) { memcpy(p, &v, 4); }
static inline void zso_writeLE64(void* p, u64 v) { memcpy(p, &v, 8); }
static inline u32 zso_highbit32(u32 v) { return 31u - (u32)__builtin_clz(v); }

/* U/ZstdInternal.cs:38 (LL_bits), :120 (ML_bits); U/ZstdDecompressInternal.cs:9,49,85,121 (bases) */
static const u8 ZSO_LL_bits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0, 1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static const u32 ZSO_LL_base[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,
                                     0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
static const u8 ZSO_ML_bits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                    1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static const u32 ZSO_ML_base[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                     35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };
/* U/ZstdInternal.cs:78,177,236 (default norms) */
static const s16 ZSO_LL_defaultNorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static const s16 ZSO_ML_defaultNorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                            1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static const s16 ZSO_OF_defaultNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

u64 zso_xxh64(const void* data, size_t len, u64 seed);

/* decoder (zso_dec.c) */
size_t zso_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize);
/* shared by the dictionary loaders of both sides (FSE_readNCount, U/EntropyCommon.cs:60-250; HUF_readStats, :292-402) */
size_t zso_readNCount(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const void* src, size_t srcSize);
size_t zso_huf_readStats(u8* weights, u32* nbSymbolsPtr, u32* tableLogPtr, u32* rankStats, const void* src, size_t srcSize);
size_t zso_decompress_usingDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* dict, size_t dictSize);
u64    zso_decompressBound(const void* src, size_t srcSize);
u64    zso_getFrameContentSize(const void* src, size_t srcSize);
size_t zso_findFrameCompressedSize(const void* src, size_t srcSize);

#endif
