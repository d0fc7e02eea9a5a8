// zmi_cparams.h — level -> compression parameters on the host.
//
// The table is the reference's ZSTD_defaultCParameters (U/Clevels.cs:8-941: four tiers by source size x levels 0..22); the
// resolution follows ZSTD_getCParams_internal (U/ZstdCompress.cs:7891-7927) and ZSTD_adjustCParams_internal (:2023-2094) for
// a known source size and no dictionary attached to the tables.  The product compresses independent chunks of at most 64 KiB,
// each its own frame, so the size a level is resolved against is the chunk's: the <= 128 KiB and <= 16 KiB tiers are the ones
// that occur.  What the kernels then honour of the seven values is stated at resolve_for_kernels().
#pragma once
#include <stdint.h>

namespace zmi {

struct CParams { uint32_t windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy; };

enum : uint32_t { kStratFast = 1, kStratDfast = 2, kStratGreedy = 3, kStratLazy = 4, kStratLazy2 = 5, kStratBtlazy2 = 6,
                  kStratBtopt = 7, kStratBtultra = 8, kStratBtultra2 = 9 };

// { windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy }, rows = levels 0 (base for negative levels) .. 22
static const CParams kDefaultCParams[4][23] = {
    {   // srcSize > 256 KiB (U/Clevels.cs:10-241)
        { 19, 12, 13, 1, 6,   1, 1 }, { 19, 13, 14, 1, 7,   0, 1 }, { 20, 15, 16, 1, 6,   0, 1 }, { 21, 16, 17, 1, 5,   0, 2 },
        { 21, 18, 18, 1, 5,   0, 2 }, { 21, 18, 19, 3, 5,   2, 3 }, { 21, 18, 19, 3, 5,   4, 4 }, { 21, 19, 20, 4, 5,   8, 4 },
        { 21, 19, 20, 4, 5,  16, 5 }, { 22, 20, 21, 4, 5,  16, 5 }, { 22, 21, 22, 5, 5,  16, 5 }, { 22, 21, 22, 6, 5,  16, 5 },
        { 22, 22, 23, 6, 5,  32, 5 }, { 22, 22, 22, 4, 5,  32, 6 }, { 22, 22, 23, 5, 5,  32, 6 }, { 22, 23, 23, 6, 5,  32, 6 },
        { 22, 22, 22, 5, 5,  48, 7 }, { 23, 23, 22, 5, 4,  64, 7 }, { 23, 23, 22, 6, 3,  64, 8 }, { 23, 24, 22, 7, 3, 256, 9 },
        { 25, 25, 23, 7, 3, 256, 9 }, { 26, 26, 24, 7, 3, 512, 9 }, { 27, 27, 25, 9, 3, 999, 9 },
    },
    {   // srcSize <= 256 KiB (U/Clevels.cs:243-474)
        { 18, 12, 13, 1, 5,   1, 1 }, { 18, 13, 14, 1, 6,   0, 1 }, { 18, 14, 14, 1, 5,   0, 2 }, { 18, 16, 16, 1, 4,   0, 2 },
        { 18, 16, 17, 3, 5,   2, 3 }, { 18, 17, 18, 5, 5,   2, 3 }, { 18, 18, 19, 3, 5,   4, 4 }, { 18, 18, 19, 4, 4,   4, 4 },
        { 18, 18, 19, 4, 4,   8, 5 }, { 18, 18, 19, 5, 4,   8, 5 }, { 18, 18, 19, 6, 4,   8, 5 }, { 18, 18, 19, 5, 4,  12, 6 },
        { 18, 19, 19, 7, 4,  12, 6 }, { 18, 18, 19, 4, 4,  16, 7 }, { 18, 18, 19, 4, 3,  32, 7 }, { 18, 18, 19, 6, 3, 128, 7 },
        { 18, 19, 19, 6, 3, 128, 8 }, { 18, 19, 19, 8, 3, 256, 8 }, { 18, 19, 19, 6, 3, 128, 9 }, { 18, 19, 19, 8, 3, 256, 9 },
        { 18, 19, 19, 10, 3, 512, 9 }, { 18, 19, 19, 12, 3, 512, 9 }, { 18, 19, 19, 13, 3, 999, 9 },
    },
    {   // srcSize <= 128 KiB (U/Clevels.cs:476-707): the tier of a full 64 KiB chunk
        { 17, 12, 12, 1, 5,   1, 1 }, { 17, 12, 13, 1, 6,   0, 1 }, { 17, 13, 15, 1, 5,   0, 1 }, { 17, 15, 16, 2, 5,   0, 2 },
        { 17, 17, 17, 2, 4,   0, 2 }, { 17, 16, 17, 3, 4,   2, 3 }, { 17, 16, 17, 3, 4,   4, 4 }, { 17, 16, 17, 3, 4,   8, 5 },
        { 17, 16, 17, 4, 4,   8, 5 }, { 17, 16, 17, 5, 4,   8, 5 }, { 17, 16, 17, 6, 4,   8, 5 }, { 17, 17, 17, 5, 4,   8, 6 },
        { 17, 18, 17, 7, 4,  12, 6 }, { 17, 18, 17, 3, 4,  12, 7 }, { 17, 18, 17, 4, 3,  32, 7 }, { 17, 18, 17, 6, 3, 256, 7 },
        { 17, 18, 17, 6, 3, 128, 8 }, { 17, 18, 17, 8, 3, 256, 8 }, { 17, 18, 17, 10, 3, 512, 8 }, { 17, 18, 17, 5, 3, 256, 9 },
        { 17, 18, 17, 7, 3, 512, 9 }, { 17, 18, 17, 9, 3, 512, 9 }, { 17, 18, 17, 11, 3, 999, 9 },
    },
    {   // srcSize <= 16 KiB (U/Clevels.cs:709-940)
        { 14, 12, 13, 1, 5,   1, 1 }, { 14, 14, 15, 1, 5,   0, 1 }, { 14, 14, 15, 1, 4,   0, 1 }, { 14, 14, 15, 2, 4,   0, 2 },
        { 14, 14, 14, 4, 4,   2, 3 }, { 14, 14, 14, 3, 4,   4, 4 }, { 14, 14, 14, 4, 4,   8, 5 }, { 14, 14, 14, 6, 4,   8, 5 },
        { 14, 14, 14, 8, 4,   8, 5 }, { 14, 15, 14, 5, 4,   8, 6 }, { 14, 15, 14, 9, 4,   8, 6 }, { 14, 15, 14, 3, 4,  12, 7 },
        { 14, 15, 14, 4, 3,  24, 7 }, { 14, 15, 14, 5, 3,  32, 8 }, { 14, 15, 15, 6, 3,  64, 8 }, { 14, 15, 15, 7, 3, 256, 8 },
        { 14, 15, 15, 5, 3,  48, 9 }, { 14, 15, 15, 6, 3, 128, 9 }, { 14, 15, 15, 7, 3, 256, 9 }, { 14, 15, 15, 8, 3, 256, 9 },
        { 14, 15, 15, 8, 3, 512, 9 }, { 14, 15, 15, 9, 3, 512, 9 }, { 14, 15, 15, 10, 3, 999, 9 },
    },
};

static inline uint32_t cp_highbit32(uint32_t v) { uint32_t r = 0; while (v >>= 1) ++r; return r; }

// ZSTD_cycleLog (U/ZstdCompress.cs:1970-1974): binary-tree strategies use half the chain for the cycle
static inline uint32_t cp_cycle_log(uint32_t chainLog, uint32_t strategy) { return chainLog - (strategy >= kStratBtlazy2 ? 1u : 0u); }

// ZSTD_getCParams_internal + ZSTD_adjustCParams_internal for a known source size, no dictionary, ZSTD_cpm_noAttachDict
static inline CParams get_cparams(int level, uint64_t srcSize)
{
    const uint32_t tier = (srcSize <= (256u << 10)) + (srcSize <= (128u << 10)) + (srcSize <= (16u << 10));
    const int row = level == 0 ? 3 : level < 0 ? 0 : level > 22 ? 22 : level;
    CParams cp = kDefaultCParams[tier][row];
    if (level < 0) {                                        // negative levels: acceleration in targetLength (:7915-7920)
        const int clamped = level < -(1 << 17) ? -(1 << 17) : level;
        cp.targetLength = (uint32_t)(-clamped);
    }
    if (srcSize < (1ull << 30)) {
        const uint32_t tSize = (uint32_t)srcSize;
        const uint32_t srcLog = tSize < 64 ? 6 : cp_highbit32(tSize - 1) + 1;
        if (cp.windowLog > srcLog) cp.windowLog = srcLog;
    }
    {   // dictAndWindowLog = windowLog without a dictionary (ZSTD_dictAndWindowLog, :1985-2013)
        const uint32_t cycleLog = cp_cycle_log(cp.chainLog, cp.strategy);
        if (cp.hashLog > cp.windowLog + 1) cp.hashLog = cp.windowLog + 1;
        if (cycleLog > cp.windowLog) cp.chainLog -= cycleLog - cp.windowLog;
    }
    if (cp.windowLog < 10) cp.windowLog = 10;
    return cp;
}

// ZSTD_literalsCompressionIsDisabled in its default (auto) mode, U/ZstdCompressInternal.cs:146-173
static inline bool literals_compression_disabled(const CParams& cp) { return cp.strategy == kStratFast && cp.targetLength > 0; }

// What the gfx950 match finders implement for a chunk (lz_fast.hip): tables of 2^13 buckets with 16-bit tags in LDS whatever the
// level asks for, a 6-byte hash for the fast strategy and 8-byte + 5-byte hashes for the others, a 4-byte verification.  These
// are the values a caller may set explicitly without being refused; anything else within bounds is parameter_unsupported.
constexpr uint32_t kKernelHashLog = 13;
static inline uint32_t kernel_min_match(uint32_t strategy) { return strategy == kStratFast ? 6u : 5u; }

} // namespace zmi
