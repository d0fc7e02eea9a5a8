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
typedef int64_t  s64;

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
    u32 rleByte;        // the byte of an RLE literals section
    u32 litFromSrc;     // 1 = the chunk has no sequences and its literals were never copied: they ARE the chunk's source bytes
    u32 regionCursor;   // != 0: lz_kernel stopped behind the chunk's first tile (dense data); lz_region_kernel does the rest from
                        // parse cursor regionCursor - 1 (nbSeq / litSize / litFromSrc hold the state after that tile)
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

// The decoder's work lists (decode_walk.hip builds them).  A frame is a run of blocks; blocks are the
// unit every parallel stage works on: a block's literals and its three FSE state chains depend on nothing but the tables the
// pre-pass resolves for it (U/ZstdDecompressBlock.cs:197-207, 1780-1786), only the LZ execution is ordered inside a frame.
struct FrameDesc {      // one per frame found by the frame walk (U/ZstdDecompress.cs:877-951)
    u64 srcOff;         // offset of the frame in the compressed input
    u64 dstOff;         // offset of its content in the output (frames without a content size: after their regenerated sizes are known)
    u64 scratchOff;     // offset of its literals in the literal scratch: the content-size (or bound) prefix sum the walk produced
    u64 srcSize;        // compressed frame size
    u64 dstSize;        // content size; for a frame without one the bound nbBlocks x blockSizeMax until the regenerated size replaces it
    u32 firstBlock, nbBlocks;
    u32 unsized;        // 1 = the header carries no content size
    u32 checksum;       // 1 = a 4-byte XXH64 checksum follows the last block
    u32 bad;            // 1 = some stage failed in this frame (its error is in the status words): later stages leave it alone
    u32 hasSeq;         // 1 = at least one block holds sequences (the ordered executor has work in this frame)
    u32 viaOrigin;      // 1 = the frame's matches are resolved by the origin path (decode_origin.hip): exec_matches only checks its checksum
    u32 pad;
    u64 originOff;      // then: index of its first entry in the origin array
};

constexpr u32 kNoBlock   = 0xFFFFFFFFu;     // table source: nothing defined it (corruption, or the default where that is legal)
constexpr u32 kDictBlock = 0xFFFFFFFEu;     // table source: the formatted dictionary

// One block of a frame.  Filled in stages: the walk (position, type, size), block_parse (the sections of a compressed block),
// block_link (where repeat-mode tables and treeless literals come from, literal offsets, record offsets), seq_decode (regenerated
// size, repcode transfer), block_offsets (output offset, repcodes at the block's start).
struct BlockDesc {
    u64 srcOff;         // the block's body in the compressed input (absolute; behind its 3-byte header)
    u64 dstRel;         // output offset relative to its frame's
    u64 litRel;         // offset of its regenerated literals relative to the frame's scratch (Huffman-coded literals sections only)
    u64 seqBase;        // index of its first sequence record
    u32 frame;
    u32 bsz;            // body bytes in the input (an RLE block: 1)
    u32 outSize;        // regenerated bytes: raw/RLE from the header; compressed: litSize when it has no sequences, else from seq_decode
    u8  type, last, litType, litSingle;     // block type 0 raw 1 RLE 2 compressed; literals section type 0 raw 1 RLE 2 compressed 3 treeless
    u32 litSize, litCSize;
    u32 lhSize;         // literals section header bytes
    u32 hufSrc;         // block whose literals section holds this block's Huffman table description (itself, an earlier one, kDictBlock)
    u32 nbSeq;
    u32 modes;          // the symbol-compression-modes byte (LL << 6 | OF << 4 | ML << 2)
    u32 tblOff[3];      // LL, OF, ML table descriptions inside the body (RLE byte or NCount), where the mode has one
    u32 bitsOff;        // start of the sequence bitstream inside the body
    u32 tblSrc[3];      // block whose sequences header defines the table this block uses (itself unless repeat mode)
    // repcodes: out = f(in), slot by slot.  kind 0: the constant val; kind 1..3: max(in[kind - 1] - val, 1)  (U/ZstdDecompressBlock.cs:2387-2443)
    u32 repKind[3], repVal[3];
    u32 repIn[3];       // the three repcodes at the block's start (block_offsets)
    u32 err;            // first error found in this block (0 = none)
    u32 litInPlace;     // 1 = no sequences: the literal decoder writes the block's output itself (block_link decides; when the literal
                        // decoder runs beside seq_decode only a sized frame's first block can — the others' offsets are not known yet)
};

// One decoded sequence (U/ZstdDecompressBlock.cs:2360-2484) as seq_decode leaves it for the executors, packed into 8 bytes:
//   bits  0-16  litLength (<= 131071: LL_base[35] + 16 extra bits)
//   bits 17-33  matchLength - 3 (<= 131071: ML_base[52] + 16 extra bits - 3)
//   bits 34-63  bit 29 clear: the match offset (1 .. 2^29 - 1; a frame that reaches further back is refused with windowTooLarge);
//               bit 29 set: a repcode left symbolic — bits 27-28 = i (1..3), bits 0-26 = d: the offset is max(repIn[i - 1] - d, 1)
// The output position is not stored: consumers walk a block's records in order and carry the running sum of litLength + matchLength.
struct SeqRec { u32 lo, hi; };
constexpr u32 kRecOffMax = (1u << 29) - 1;

// status words shared by the decoder's kernels and the host
enum : u32 { kStFrames = 0, kStErr = 1, kStTotalLo = 2, kStTotalHi = 3, kStUsable = 4, kStUnsized = 5, kStBlocks = 6, kStSeqLo = 8, kStSeqHi = 9,
             kStErrKeyLo = 10, kStErrKeyHi = 11, kStActualLo = 12, kStActualHi = 13,
             kStOriginFrames = 14,      // frames the origin path took (origin_select_kernel)
             kStOriginLo = 16, kStOriginHi = 17,        // entries of the origin array handed out so far (u64)
             kStOriginChanged = 18,     // .. 18 + kOriginRounds: round r of the pointer jumping changed something
             kStLitLo = 54, kStLitHi = 55,              // regenerated bytes of all Huffman-coded literals sections (u64; block_link)
             kStBigBins = 56,           // 12 x u64: content bytes of the frames with sequences by size class, bin k = [2^(20+k), 2^(21+k)), below 2^30
             kStWords = 80 };
constexpr u32 kOriginRounds = 36;

// Stage timing (ZSTDMI_*_setProfiling): a launcher that issues several kernels marks the end of each on the context's timer, so
// that a stage time is ONE kernel's duration (bench.py prices the slowest kernel against the HBM roof).  No-op when profiling is off.
struct StageHook {
    void (*fn)(void* self, const char* name) = nullptr; void* self = nullptr;
    void operator()(const char* name) const { if (fn) fn(self, name); }
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
