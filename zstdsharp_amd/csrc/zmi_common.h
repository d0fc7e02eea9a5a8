// zmi_common.h — layouts shared by the host driver and the gfx950 kernels.
//
// Domain vocabulary follows the reference (ZstdSharp / zstd 1.5.1): chunks become frames, each frame holds
// one block; a block is a literals section + a sequences section (U/ZstdCompress.cs:3236-3354).
#pragma once
#include <stdint.h>
#include <stddef.h>

namespace zmi {

typedef uint8_t  u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t  s16;
typedef int32_t  s32;

// ---- framing: one independent zstd frame per 64 KiB chunk ----
constexpr u32 kChunkLog   = 16;
constexpr u32 kChunkSize  = 1u << kChunkLog;
// literal buffer of chunk c starts at c * kLitStride: the 320-byte skew keeps the chunks' writers and readers, which move in
// lockstep, from hitting addresses that agree in their low 16 bits (channel camping)
constexpr u32 kLitStride  = kChunkSize + 320;
constexpr u32 kSlotStride = kChunkSize + 512;      // >= ZSTD_compressBound(64 KiB) + frame/block headers + checksum
constexpr u32 kMaxSeq     = kChunkSize / 4;        // a sequence consumes >= 4 input bytes in this match finder

// one stored sequence: same meaning as the reference's seqDef_s (U/seqDef_s.cs)
struct Seq { u32 offBase; u16 litLength; u16 mlBase; };   // offBase 1..3 repcode, >=4 distance+3 ; mlBase = matchLength-3

enum LitMode : u32 { kLitRaw = 0, kLitRle = 1, kLitCompressed = 2 };

// per-chunk record that travels between the pipeline's kernels (HBM resident)
struct ChunkMeta {
    u32 srcSize;        // bytes of input in this chunk (<= 64 KiB)
    u32 nbSeq;          // sequences found by the match finder
    u32 litSize;        // literal bytes (trailing literals included)
    u32 fhSize;         // frame header bytes for this chunk
    // literals section, decided by huf_build
    u32 litMode;        // LitMode
    u32 litSingle;      // 1 = single stream
    u32 lhSize;         // literals section header bytes (1..5)
    u32 hufHdrSize;     // Huffman tree description bytes
    u32 streamSize[4];  // bytes per Huffman stream
    u32 litSectionSize; // whole literals section
    // result
    u32 bodySize;       // compressed block body (literals + sequences sections) or 0 if stored raw/RLE
    u32 blockType;      // 0 raw, 1 RLE, 2 compressed
    u32 outSize;        // frame bytes in the slot (header + block + optional checksum)
    u32 checksum;       // low 32 bits of XXH64 of the chunk when the checksum flag is set
    u32 pad[2];
};

// Huffman code table for one chunk (HBM): canonical codes as the reference assigns them (U/HufCompress.cs:750-788)
struct HufTable {
    u16 code[256];
    u8  nbBits[256];
    u8  hdr[132];       // tree description as written by HUF_writeCTable (<= 129 bytes)
    u32 maxSV, tableLog;
};

// ---- decoder side ----
// Literal scratch of frame f starts at its output offset + f * kLitSkew: without the skew all frames' streams write addresses
// that agree in their low 14 bits at any moment (four 16 KiB segments per 64 KiB frame, decoded in lockstep).
#ifndef ZMI_LITSKEW
#define ZMI_LITSKEW 320
#endif
constexpr u32 kLitSkew = ZMI_LITSKEW;

// A formatted dictionary as the decoder sees it (filled by dict_parse_kernel; ZSTD_loadDEntropy, U/ZstdDecompress.cs:1773-1875):
// offsets into the dictionary bytes of the Huffman description and of the three NCounts, the repcodes, where the content starts.
struct DictInfo {
    u32 err;                        // 0, or kErrDictionaryCorrupted
    u32 dictID;
    u32 hufOff, hufSize;
    u32 ofOff, mlOff, llOff;        // each NCount runs up to the next offset (llOff up to repOff)
    u32 repOff;
    u32 rep[3];
    u32 contentOff, contentSize;
    u32 pad[3];
};

struct FrameDesc {      // one per frame found by the frame walk (U/ZstdDecompress.cs:877-951)
    u64 srcOff;         // offset of the frame in the compressed input
    u64 dstOff;         // offset of its content in the output
    u32 srcSize;        // compressed frame size
    u32 dstSize;        // content size; for a frame without one (unsized = 1) the bound nbBlocks x blockSizeMax (U/ZstdDecompress.cs:877-951)
    u32 unsized;        // 1 = the header carries no content size: the decoder reports the regenerated size instead of checking it
    u32 pad;
};

// error codes: U/ZSTD_ErrorCode.cs
enum : u32 {
    kErrGeneric = 1, kErrPrefixUnknown = 10, kErrVersionUnsupported = 12, kErrFrameParameterUnsupported = 14,
    kErrWindowTooLarge = 16, kErrCorruption = 20, kErrChecksumWrong = 22, kErrDictionaryCorrupted = 30,
    kErrDictionaryWrong = 32, kErrParameterUnsupported = 40, kErrParameterOutOfBound = 42,
    kErrTableLogTooLarge = 44, kErrMaxSymbolValueTooLarge = 46, kErrMaxSymbolValueTooSmall = 48,
    kErrStageWrong = 60, kErrInitMissing = 62, kErrMemoryAllocation = 64, kErrWorkSpaceTooSmall = 66,
    kErrDstSizeTooSmall = 70, kErrSrcSizeWrong = 72, kErrDstBufferNull = 74, kErrMaxCode = 120
};

} // namespace zmi
