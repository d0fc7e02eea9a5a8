/*
 * oracle/zso_enc.h — CPU oracle: encoder-side API.  TEST INFRASTRUCTURE ONLY
 * (see zso_common.h).
 */
#ifndef ZSO_ENC_H
#define ZSO_ENC_H
#include "zso_common.h"

/* U/ZSTD_strategy.cs */
enum { ZSO_fast = 1, ZSO_dfast = 2, ZSO_greedy = 3, ZSO_lazy = 4 };

typedef struct { u32 windowLog, chainLog, hashLog, searchLog, minMatch, targetLength, strategy; } zso_cparams;

/* One stored sequence, same meaning as the reference's seqDef_s (U/seqDef_s.cs):
 * offBase = offCode + 1 : 1..3 = repcode, >= 4 = distance + 3;  mlBase = matchLength - 3. */
typedef struct { u32 offBase; u16 litLength; u16 mlBase; } zso_seq;

zso_cparams zso_getCParams(int level, u64 srcSize);
size_t zso_compressBound(size_t srcSize);

/* Reference behaviour: one frame, blocks of min(128 KiB, window), history across blocks. */
size_t zso_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksumFlag);

/* The GPU path's framing: one independent frame per `chunkSize` bytes, concatenated. */
size_t zso_compress_chunked(void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                            int level, int checksumFlag, size_t chunkSize);

/* Dictionary as history in front of the input — raw content, or formatted (entropy tables and repcodes of its header
 * become the first block's previous state, ZSTD_loadCEntropy) — fast strategy only; see zso_enc.c for what this restates.
 * Generator of dictionary frames for decoder tests; "parity unpinned". */
size_t zso_compress_usingDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                              const void* dict, size_t dictSize, int level, int checksumFlag);

/* Writer of a FORMATTED dictionary for tests (see zso_enc.c); zso_compress_usingDict / zso_decompress_usingDict take it. */
size_t zso_make_dictionary(void* dst, size_t cap, const void* content, size_t contentSize,
                           const void* sample, size_t sampleSize, u32 dictID);

/* Stage hooks for kernel-level parity tests (SURVEY.md §8 a-4 … a-11). */

/* a-4: run the level's block match finder over ONE block with fresh state (rep = {1,4,8}, empty table).
 * Writes up to seqCap sequences and the literal bytes; returns nbSeq, *litSizePtr = total literal bytes
 * (trailing literals included). */
size_t zso_block_sequences(zso_seq* seqs, size_t seqCap, u8* lits, size_t* litSizePtr,
                           const void* src, size_t srcSize, int level);

/* a-7 … a-11: entropy-code a seqStore as the body of one compressed block with NO previous entropy state
 * (first block of a frame).  Returns the body size, 0 if "not compressible" by the reference's rules
 * (ZSTD_entropyCompressSeqStore, U/ZstdCompress.cs:3357-3392), or an error code. */
size_t zso_entropy_block(void* dst, size_t dstCapacity, const zso_seq* seqs, size_t nbSeq,
                         const u8* lits, size_t litSize, u32 strategy, size_t srcSize);

/* a-9 alone: Huffman code lengths the reference would assign for a histogram (HUF_buildCTable_wksp).
 * Returns the table's max code length, fills nbBits[0..maxSymbolValue]. */
size_t zso_huf_buildLengths(u8* nbBits, const u32* count, u32 maxSymbolValue, u32 maxNbBits);

#endif
