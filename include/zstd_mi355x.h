/*
 * zstd_mi355x.h — C ABI of libzstd_mi355x.so, the MI355X (gfx950) block compressor/decompressor that sits
 * behind ZstdSharp's one-shot path (Compressor.Wrap / Decompressor.Unwrap).
 *
 * Every ZSTD_* entry point below keeps the name, argument order, types and error convention of the function the
 * reference calls at that point, so the reference's own P/Invoke precedent
 * (/root/reference/src/Zstd.Extern/ExternMethods.cs:8-42: cdecl, IntPtr contexts/buffers, nuint sizes, 32-bit
 * enums) binds to this library unchanged, and a `Methods`-shaped shim lets Compressor.cs / Decompressor.cs
 * compile as they are.  Citations: S/ = src/ZstdSharp/, U/ = src/ZstdSharp/Unsafe/.
 *
 * Return convention (U/ErrorPrivate.cs:10-24): size_t result; it is an error iff > (size_t)-120, and then
 * (0 - result) is a ZSTD_ErrorCode (U/ZSTD_ErrorCode.cs).  A too-small destination yields exactly (size_t)-70,
 * which TryWrap/TryUnwrap test for (S/Compressor.cs:116-120, S/Decompressor.cs:105-109).
 *
 * Buffers may be host memory (as the C# callers pass, pinned only for the call) or device memory (HBM): the
 * library asks the HIP runtime which it is and stages host buffers itself.  Nothing is retained after return.
 * One context is used by one thread at a time; any number of contexts may be used concurrently.
 */
#ifndef ZSTD_MI355X_H
#define ZSTD_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ZSTD_CCtx_s ZSTD_CCtx;
typedef struct ZSTD_DCtx_s ZSTD_DCtx;

/* U/ZSTD_cParameter.cs / U/ZSTD_dParameter.cs values that the safe API uses */
enum {
    ZSTD_c_compressionLevel = 100, ZSTD_c_windowLog = 101, ZSTD_c_hashLog = 102, ZSTD_c_chainLog = 103,
    ZSTD_c_searchLog = 104, ZSTD_c_minMatch = 105, ZSTD_c_targetLength = 106, ZSTD_c_strategy = 107,
    ZSTD_c_contentSizeFlag = 200, ZSTD_c_checksumFlag = 201, ZSTD_c_dictIDFlag = 202, ZSTD_c_nbWorkers = 400,
    ZSTD_d_windowLogMax = 100
};

/* ---- compression context: S/Compressor.cs:32,60,138 -> U/ZstdCompress.cs:24-27, 43-62, 137-160 ---- */
ZSTD_CCtx* ZSTD_createCCtx(void);
size_t     ZSTD_freeCCtx(ZSTD_CCtx* cctx);                                   /* NULL is accepted */
/* S/Compressor.cs:48-54 -> U/ZstdCompress.cs:819-884, 1270-1283.  compressionLevel (negative levels included: fast strategy
 * with a probing step and raw literals, as U/ZstdCompress.cs:7915-7920 + U/ZstdCompressInternal.cs:146-173), checksumFlag,
 * dictIDFlag, strategy (1..9, mapped onto the three finders) and targetLength are honoured; windowLog >= 16, contentSizeFlag = 1,
 * nbWorkers = 0, and for hashLog / minMatch / chainLog / searchLog the value the kernels implement (13; 6 or 5; the level's
 * own) are accepted; anything else within bounds returns parameter_unsupported — nothing is silently ignored. */
size_t     ZSTD_CCtx_setParameter(ZSTD_CCtx* cctx, int param, int value);
size_t     ZSTD_CCtx_getParameter(const ZSTD_CCtx* cctx, int param, int* value);
/* S/Compressor.cs:43-56 (dictionary load) -> U/ZstdCompress.cs:1286-1330, 5465-5503.  RAW-CONTENT dictionaries (any bytes
 * that do not start with the magic 0xEC30A437): history in front of every frame, no dictID.  FORMATTED dictionaries (the
 * magic, a dictID, entropy tables, repcodes, content — what the trainer returns): the content is the history, the frames
 * carry the dictID (unless ZSTD_c_dictIDFlag = 0) and start from the dictionary's repcodes; its entropy tables are not
 * used (every block carries its own — valid for any decoder holding the dictionary).  A malformed header ->
 * dictionary_corrupted (at this call when a device is bound, else at first use).  NULL/0 = no dictionary; under 8 bytes =
 * ignored, as in the reference.  The pointer may be host or device memory; the bytes are copied. */
size_t     ZSTD_CCtx_loadDictionary(ZSTD_CCtx* cctx, const void* dict, size_t dictSize);
/* S/Compressor.cs:73-76 -> U/ZstdCompress.cs:19-22 */
size_t     ZSTD_compressBound(size_t srcSize);
/* S/Compressor.cs:94 -> U/ZstdCompress.cs:7138-7177 */
size_t     ZSTD_compress2(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);
/* B/Benchmark.cs:67, X/ExternMethods.cs:17-18 -> U/ZstdCompress.cs:5751-5776: the level alone — default frame parameters, no
 * dictionary even if one is loaded; the context's sticky parameters and dictionary are left as they are */
size_t     ZSTD_compressCCtx(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int compressionLevel);
/* S/Compressor.cs:8-9 -> U/ZstdCompress.cs:7762-7770 */
int        ZSTD_minCLevel(void);
int        ZSTD_maxCLevel(void);
int        ZSTD_defaultCLevel(void);

/* ---- decompression context: S/Decompressor.cs:12,25,124 -> U/ZstdDecompress.cs:326-395 ---- */
ZSTD_DCtx* ZSTD_createDCtx(void);
size_t     ZSTD_freeDCtx(ZSTD_DCtx* dctx);
size_t     ZSTD_DCtx_setParameter(ZSTD_DCtx* dctx, int param, int value);   /* S/Decompressor.cs:41-46 */
size_t     ZSTD_DCtx_getParameter(ZSTD_DCtx* dctx, int param, int* value);
/* S/Decompressor.cs:36-48 -> U/ZstdDecompress.cs:1909-1931, 1758-1875: raw-content dictionaries (history in front of every
 * frame) and formatted ones (frames start from the dictionary's Huffman/FSE tables and repcodes and must name its dictID or
 * none; a malformed header -> dictionary_corrupted) */
size_t     ZSTD_DCtx_loadDictionary(ZSTD_DCtx* dctx, const void* dict, size_t dictSize);
/* S/Decompressor.cs:53 -> U/ZstdDecompress.cs:971-993 ; error = (unsigned long long)-2 (S/ThrowHelper.cs:7-8) */
unsigned long long ZSTD_decompressBound(const void* src, size_t srcSize);
unsigned long long ZSTD_getFrameContentSize(const void* src, size_t srcSize);
size_t     ZSTD_findFrameCompressedSize(const void* src, size_t srcSize);
/* S/Decompressor.cs:86 -> U/ZstdDecompress.cs:1365-1368 */
size_t     ZSTD_decompressDCtx(ZSTD_DCtx* dctx, void* dst, size_t dstCapacity, const void* src, size_t srcSize);

/* ---- errors: S/ThrowHelper.cs:12-13 -> U/ErrorPrivate.cs:10-24, 35-120 ---- */
unsigned    ZSTD_isError(size_t code);
const char* ZSTD_getErrorName(size_t code);
/* S/ThrowHelper.cs:18-24 (EnsureZdictSuccess) -> U/Zdict.cs:11-19.  The trainer itself (ZDICT_trainFromBuffer, S/DictBuilder.cs)
 * is outside this library: a shim that keeps DictBuilder.cs routes that one call to the managed implementation. */
unsigned    ZDICT_isError(size_t code);
const char* ZDICT_getErrorName(size_t code);
unsigned    ZSTD_versionNumber(void);        /* 10501, as U/ZstdCommon.cs:11-21 */
const char* ZSTD_versionString(void);

/* ---- streaming entry points of the safe API (S/Compressor.cs:108-116 <- S/CompressionStream.cs:130-190;
 *      S/Decompressor.cs:97-106 <- S/DecompressionStream.cs:88-162; U/ZstdCompress.cs:6632-6861, U/ZstdDecompress.cs:2816-3205).
 *      Adapters on the batched engine: input is collected on the host and goes through the one-shot
 *      pipeline in batches (compress: 16 MiB or at flush/end; decompress: every whole frame received so far).  endOp is
 *      ZSTD_EndDirective (0 continue, 1 flush, 2 end); return values follow the reference (bytes left to flush / 0 at a
 *      frame boundary / hint), including the hostage-byte rule of U/ZstdDecompress.cs:3170-3194.  Host pointers only. ---- */
typedef struct { const void* src; size_t size; size_t pos; } ZSTD_inBuffer;
typedef struct { void* dst; size_t size; size_t pos; } ZSTD_outBuffer;
size_t ZSTD_compressStream2(ZSTD_CCtx* cctx, ZSTD_outBuffer* output, ZSTD_inBuffer* input, int endOp);
/* frames whose window (or single-segment content size) exceeds 1 << ZSTD_d_windowLogMax (default 27) are refused with
 * frameParameter_windowTooLarge as soon as their header has arrived (U/ZstdDecompress.cs:2965-2969) */
size_t ZSTD_decompressStream(ZSTD_DCtx* dctx, ZSTD_outBuffer* output, ZSTD_inBuffer* input);
/* buffer sizes the stream classes ask for: S/CompressionStream.cs:41 -> U/ZstdCompress.cs:6246-6249; S/DecompressionStream.cs:41 ->
 * U/ZstdCompress.cs:6241-6244; U/ZstdDecompress.cs:2096-2104 */
size_t ZSTD_CStreamInSize(void);
size_t ZSTD_CStreamOutSize(void);
size_t ZSTD_DStreamInSize(void);
size_t ZSTD_DStreamOutSize(void);

/* =====================================================================================================
 * Extensions (not in the reference): device selection, HBM-resident calls, per-stage timing, and test hooks.
 * ===================================================================================================== */
int    ZSTDMI_deviceCount(void);                       /* number of visible MI355X devices; 0 => every call fails loudly */
size_t ZSTDMI_CCtx_setDevice(ZSTD_CCtx* cctx, int device);
size_t ZSTDMI_DCtx_setDevice(ZSTD_DCtx* dctx, int device);
/* Several GPUs behind one context (north_star: "chunks partition naturally across the 8 GPUs of one node"): one device worker per
 * entry of `devices` (an ordinal may be listed more than once).  ZSTD_compress2 / ZSTD_compressCCtx / ZSTD_decompressDCtx and the
 * device-pointer calls then deal the call's frames to the workers in contiguous shares — each stages, compresses or decodes its share on
 * its own device and stream — and copy the results one behind the other into the caller's buffer.  The bytes written do not depend on
 * the number of workers.  n <= 1 returns to a single device.  (The torch.distributed path of bench.py — one process per GPU, RCCL
 * all-gather-v of the shards — is the other way to use a node; this one needs no process group.) */
size_t ZSTDMI_CCtx_setDevices(ZSTD_CCtx* cctx, const int* devices, int n);
size_t ZSTDMI_DCtx_setDevices(ZSTD_DCtx* dctx, const int* devices, int n);
/* run on a caller-owned HIP stream (e.g. torch's current stream); NULL restores the context's own stream */
size_t ZSTDMI_CCtx_setStream(ZSTD_CCtx* cctx, void* hipStream);
size_t ZSTDMI_DCtx_setStream(ZSTD_DCtx* dctx, void* hipStream);

/* inputs larger than one pass are compressed pass by pass (default 16384 chunks = 1 GiB, which bounds the HBM workspace) */
size_t ZSTDMI_CCtx_setPassChunks(ZSTD_CCtx* cctx, unsigned chunksPerPass);

/* Cross-chunk history (the window ZSTD_compress_frameChunk's block loop carries from block to block, U/ZstdCompress.cs:4705-4807,
 * U/ZstdCompressInternal.cs:787-813): historyBytes > 0 makes the blocks 64 KiB - historyBytes long, each matching into the
 * historyBytes of input in front of it, and groups them into multi-block frames of frameBytes of content (0 = keep, default
 * 256 KiB); 0 = independent single-block 64 KiB frames; < 0 = by level (default: 16 KiB at levels 3-4, 32 KiB at levels >= 5,
 * off at levels 1-2 unless ZSTD_c_windowLog > 16 is set).  Ignored while a dictionary is loaded. */
size_t ZSTDMI_CCtx_setHistory(ZSTD_CCtx* cctx, int historyBytes, unsigned frameBytes);

/* parse of chunks that are dense in matches (text, source code, structured data): 0 = region parse (default: after a chunk's
 * first 4 KiB tile found >= 384 matches, the candidates of every later position are looked up first and one lane per 64
 * positions walks them the way ZSTD_compressBlock_fast walks a block, U/ZstdFast.cs:130-260), 1 = the tile loop for every chunk
 * (each 4 KiB tile verified position by position and selected in parallel; slower on dense data, sizes within 0.5 %).
 * Both produce valid, deterministic streams. */
size_t ZSTDMI_CCtx_setParser(ZSTD_CCtx* cctx, unsigned mode);

/* literal (Huffman) decoder: 0 = chosen by frame count (default), 1 = serial, 4 lanes per frame (highest throughput when
 * thousands of frames are in flight), 2 = self-synchronising, 256 lanes per frame (lowest latency per frame),
 * 3 = serial with compact tables (2 KiB + pair table per frame: twice the frames in flight) */
size_t ZSTDMI_DCtx_setLiteralDecoder(ZSTD_DCtx* dctx, unsigned mode);

/* match execution of long frames (the reference's Compressor writes ONE frame per call, U/ZstdCompress.cs:4690-4815): 0 = by cost
 * (default: a frame whose ordered one-wave walk would take longer than a parallel sweep of all such frames is resolved by origin
 * pointers — decode_origin.hip —, the others are walked), 1 = always the walk, 2 = origin pointers for every frame of 1 MiB or more */
size_t ZSTDMI_DCtx_setLongFrames(ZSTD_DCtx* dctx, unsigned mode);
/* the literal decoder beside the sequence decoder on a second stream (they need nothing of each other): 0 = when a call has few
 * blocks (default: neither kernel fills the chip then), 1 = never, 2 = always */
size_t ZSTDMI_DCtx_setOverlap(ZSTD_DCtx* dctx, unsigned mode);
/* waves per frame in the match-execution stage: 0 = by the number of frames in the call (default: few frames get up to 16 waves
 * each — a batch of 64 x waves sequences per round of dependent copies —, more than 2048 frames one wave each), else 1, 2, 4, 8 or 16 */
size_t ZSTDMI_DCtx_setExecWaves(ZSTD_DCtx* dctx, unsigned waves);
/* diagnostic: 1 if the last decompress call listed its frames with the exact serial walk (frames naming a dictionary, frames
 * without a content size, damaged input) instead of the parallel one, 0 if not, -1 without a context */
int ZSTDMI_debugLastWalkSerial(const ZSTD_DCtx* dctx);

/* same contracts as ZSTD_compress2 / ZSTD_decompressDCtx, but src and dst MUST be device pointers (no staging) */
size_t ZSTDMI_compressDevice(ZSTD_CCtx* cctx, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize);
size_t ZSTDMI_decompressDevice(ZSTD_DCtx* dctx, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize);

/* per-stage HIP-event timing of the LAST call (enable first).  Fills up to `cap` entries, returns the count. */
size_t ZSTDMI_CCtx_setProfiling(ZSTD_CCtx* cctx, int enable);
size_t ZSTDMI_DCtx_setProfiling(ZSTD_DCtx* dctx, int enable);
int    ZSTDMI_CCtx_getStageTimes(const ZSTD_CCtx* cctx, float* ms, const char** names, int cap);
int    ZSTDMI_DCtx_getStageTimes(const ZSTD_DCtx* dctx, float* ms, const char** names, int cap);

/* test hooks (kernel-level parity against the oracle) */
typedef struct { unsigned offBase; unsigned short litLength; unsigned short mlBase; } ZSTDMI_Seq;
/* sequences + literals the match finder produced for chunk `chunkIdx` of the last ZSTDMI/ZSTD compress call */
size_t ZSTDMI_debugGetChunk(ZSTD_CCtx* cctx, size_t chunkIdx, ZSTDMI_Seq* seqs, size_t seqCap, size_t* nbSeq,
                            void* lits, size_t litCap, size_t* litSize);
/* entropy-code ONE caller-supplied seqStore with the GPU kernels; returns the compressed block body size
 * (0 = "store raw", the reference's ZSTD_entropyCompressSeqStore convention) */
size_t ZSTDMI_debugEntropyBlock(ZSTD_CCtx* cctx, void* dst, size_t dstCapacity, const ZSTDMI_Seq* seqs, size_t nbSeq,
                                const void* lits, size_t litSize, size_t srcSize);
/* the entropy stage of ONE chunk whose ChunkMeta is taken as given, WITHOUT the host's range checks, over a seqStore filled with the
 * byte `fill`: what a faulty match finder would hand over.  The kernels must bound every size themselves (meta_checked, the
 * bitstream room): returns the bytes the chunk's frame would take, never faults.  tests/test_gpu_boundary.py */
size_t ZSTDMI_debugPoisonedChunk(ZSTD_CCtx* cctx, unsigned nbSeq, unsigned litSize, unsigned srcSize, unsigned fill);

#ifdef __cplusplus
}
#endif
#endif
