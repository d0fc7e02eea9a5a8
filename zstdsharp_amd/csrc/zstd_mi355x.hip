// zstd_mi355x.hip — host side of libzstd_mi355x.so: contexts, parameters, error names, HBM workspaces and the
// launch sequences of the compress / decompress pipelines.  The C ABI is declared in include/zstd_mi355x.h.
//
// There is no CPU codec in this library: without a usable gfx950 device every compress/decompress call returns
// ZSTD_error_init_missing, loudly.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <vector>
#include <cstring>
#include <mutex>
#include <thread>
#include <new>
#include "zmi_common.h"
#include "zmi_cparams.h"
#include "../../include/zstd_mi355x.h"

namespace zmi {
// kernels (lz_fast.hip, huf_enc.hip, seq_enc.hip, frame.hip, decode.hip)
void launch_lz(u32 finder, const u8* src, u64 srcSize, u32 nChunks, Seq* seqs, u8* lits, ChunkMeta* meta, const u8* prefix, u32 prefixLen,
               u32 chunkBytes, u32 fhExtra, u32 minStrideLog, u32 frameBlocks, u16* cand, u16* chain, u32* regionList, u32 hcDepth, hipStream_t stream, StageHook hook, u32* claimCtr);
void launch_lz_probe(const u8* src, u64 srcSize, u64 front, u64 groupBytes, u32 nGroups, u32 tilesPerGroup, u32* out, hipStream_t stream);
void launch_huf_build(const u8* lits, ChunkMeta* meta, HufTable* tables, u8* slots, u32 nChunks, u32 rawLiterals, const u8* src, u32 chunkBytes,
                      hipStream_t stream, StageHook hook);
void launch_huf_encode(const u8* lits, const ChunkMeta* meta, const HufTable* tables, u8* slots, u8* dst, const u64* offsets, u64 dstCapacity,
                       u32 nChunks, const u8* src, u32 chunkBytes, hipStream_t stream);
void launch_seq_encode(Seq* seqs, ChunkMeta* meta, u8* slots, u32 nChunks, u32 strategy, u32 checksumFlag, u32 resolveReps,
                       u32 dictID, u32 dictIdBytes, const u32* initReps, u32 frameBlocks, u32 chunkBytes, u64 srcSize, hipStream_t stream);
void launch_scan_sizes(const ChunkMeta* meta, u32 nChunks, u64* offsets, u64* total, hipStream_t stream);
void launch_gather(const u8* src, u64 srcSize, const u8* slots, const ChunkMeta* meta, const u64* offsets, u8* dst, u64 dstCapacity,
                   u32 nChunks, u32 chunkBytes, hipStream_t stream);
void launch_xxh64(const u8* src, u64 srcSize, ChunkMeta* meta, u32 nChunks, u32 chunkBytes, u32 frameBlocks, hipStream_t stream);
// decoder (decode_walk.hip, decode_lit.hip, decode_seq.hip)
size_t decode_walk_workspace_bytes(u64 srcSize);
void launch_frame_walk_count(const u8* src, u64 srcSize, u32 maxFrames, u32* status, u8* walkWs, hipStream_t stream);
void launch_frame_walk_emit(const u8* src, u64 srcSize, FrameDesc* frames, BlockDesc* blocks, u8* walkWs, hipStream_t stream);
void launch_frame_walk_serial(const u8* src, u64 srcSize, FrameDesc* frames, BlockDesc* blocks, u32 maxFrames, u32* status, u32 dictID, u32 emit,
                              hipStream_t stream);
void launch_dict_parse(const u8* dict, u32 dictSize, DictInfo* out, hipStream_t stream);
void launch_block_prepass(const u8* src, FrameDesc* frames, BlockDesc* blocks, u32 nFrames, u32 nBlocks, u32 haveDict, u32 earlyLiterals, u32* status, hipStream_t stream);
void launch_seq_decode(const u8* src, const FrameDesc* frames, BlockDesc* blocks, u32 nBlocks, SeqRec* recs, u32* status,
                       const u8* dictFull, const DictInfo* di, hipStream_t stream);
void launch_block_offsets(FrameDesc* frames, BlockDesc* blocks, u32 nFrames, const DictInfo* di, u32 rescan, u64 dstCapacity, u32* status, hipStream_t stream);
void launch_decode_literals(const u8* src, u8* out, u8* scratch, const FrameDesc* frames, const BlockDesc* blocks, u32 nBlocks, u32* status,
                            u8* slowFlags, u32 mode, const u8* dictFull, const DictInfo* di, hipStream_t stream, StageHook hook);
void launch_place_literals(const u8* src, u8* out, const u8* scratch, const FrameDesc* frames, const BlockDesc* blocks, u32 nBlocks,
                           const SeqRec* recs, const u32* status, hipStream_t stream);
void launch_exec_matches(const u8* src, u8* out, const FrameDesc* frames, const BlockDesc* blocks, u32 nFrames, const SeqRec* recs, u32* status,
                         const u8* dict, u32 dictSize, hipStream_t stream, int wide);
void launch_origin_select(FrameDesc* frames, u32 nFrames, u64 minBytes, u32* list, u32 listCap, u64 originCap, u32* status, hipStream_t stream);
void launch_origin_init(const FrameDesc* frames, const BlockDesc* blocks, const u32* list, u32 listCap, u64 maxFrameBytes, const SeqRec* recs, u32* status,
                        u32* origin, u32 dictSize, hipStream_t stream);
void launch_origin_jump(const FrameDesc* frames, const u32* list, u32 listCap, u64 maxFrameBytes, u32* status, u32* origin, u32* done, u32 r0, u32 r1, hipStream_t stream);
void launch_origin_gather(const FrameDesc* frames, const u32* list, u32 listCap, u64 maxFrameBytes, const u32* status, const u32* origin, u8* out, const u8* dict, hipStream_t stream);
}

using namespace zmi;

#define ZERR(code) ((size_t)0 - (size_t)(code))
static inline bool isErr(size_t c) { return c > ZERR(kErrMaxCode); }
// No C++ exception may cross the C ABI (the caller is P/Invoke): host-side container growth is the only thing that throws here.
template <class F> static size_t guarded(F f)
{
    try { return f(); }
    catch (const std::bad_alloc&) { return ZERR(kErrMemoryAllocation); }
    catch (...) { return ZERR(kErrGeneric); }
}

namespace {

constexpr int kMaxStages = 24;

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    bool ensure(size_t n)
    {
        if (n <= cap) return true;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + (n >> 3) + 4096;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; if (hipMalloc(&p, n) != hipSuccess) { p = nullptr; return false; } want = n; }
        cap = want; return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct StageTimer {
    bool enabled = false;
    hipEvent_t ev[kMaxStages + 1] = {};
    const char* names[kMaxStages] = {};
    float ms[kMaxStages] = {};
    int n = 0; bool created = false; hipStream_t stream = nullptr;
    static void hook_fn(void* self, const char* name) { StageTimer* t = (StageTimer*)self; t->mark(name, t->stream); }
    StageHook hook() { StageHook h; if (enabled) { h.fn = hook_fn; h.self = this; } return h; }
    void begin(hipStream_t s) { n = 0; stream = s; if (!enabled) return; if (!created) { for (auto& e : ev) (void)hipEventCreate(&e); created = true; } (void)hipEventRecord(ev[0], s); }
    void mark(const char* name, hipStream_t s) { if (!enabled || n >= kMaxStages) return; names[n] = name; (void)hipEventRecord(ev[n + 1], s); n++; }
    void finish() { if (!enabled) return; for (int i = 0; i < n; i++) { float t = 0; (void)hipEventElapsedTime(&t, ev[i], ev[i + 1]); ms[i] = t; } }
    void destroy() { if (created) for (auto& e : ev) (void)hipEventDestroy(e); created = false; }
};

bool is_device_ptr(const void* p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// a zstd-format dictionary starts with the magic 0xEC30A437 (ZSTD_MAGIC_DICTIONARY); anything else is raw content
bool is_formatted_dictionary(const u8* p, size_t n)
{
    return n >= 8 && ((u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24)) == 0xEC30A437u;
}

// devices the kernels can run on: the leading run of gfx950 agents (device ordinals stay HIP's, so a context's device index means
// the same thing to the caller's runtime; the code objects in this library are gfx950 only)
int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (; ok < n; ++ok) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, ok) != hipSuccess) { (void)hipGetLastError(); break; }
        if (strncmp(p.gcnArchName, "gfx950", 6) != 0) break;
    }
    return ok;
}

} // namespace

// ======================================================================================================
struct ZSTD_CCtx_s {
    int level = 3;              // ZSTD_CLEVEL_DEFAULT
    int checksumFlag = 0, contentSizeFlag = 1, dictIDFlag = 1;
    int windowLog = 0, hashLog = 0, chainLog = 0, searchLog = 0, minMatch = 0, targetLength = 0, strategy = 0;
    int device = 0; bool deviceOk = false;
    hipStream_t ownStream = nullptr, stream = nullptr;
    DevBuf seqs, lits, meta, tables, slots, offsets, total, cand, probe, stageSrc, stageDst;
    u32 lastChunks = 0;         // chunks of the last pass (debug hook)
    const u8* lastSrc = nullptr; u32 lastChunkBytes = 0;     // its source (debug hook: chunks without sequences keep their literals there)
    u32 passChunks = 16384;     // chunks per pass: 1 GiB of input bounds the HBM workspace to ~4.2 GiB
    // cross-chunk history (row f-1): -1 = by level (on for the strategies above fast, i.e. levels >= 3: 16 KiB at levels 3-4, 32 KiB
    // above; and whenever the caller asks for a windowLog above 16), 0 = off (independent 64 KiB frames), else the bytes of
    // history per block (4 KiB units)
    int historyBytes = -1; u32 frameBytes = 256u << 10;
    u32 parser = 0;                      // ZSTDMI_CCtx_setParser
    // streaming adapter (ZSTD_compressStream2): host-side batching in front of the one-shot engine
    std::vector<u8> sIn, sOut; size_t sOutPos = 0; bool sWrote = false, sEnding = false; size_t sBatch = (size_t)16 << 20;
    StageTimer timer;
    float stageMs[kMaxStages] = {}; const char* stageNames[kMaxStages] = {}; int nStages = 0;
    // dictionary (ZSTD_CCtx_loadDictionary).  dictHost = the history bytes: the last kDictKeep bytes of a raw-content dictionary
    // or of a formatted dictionary's content; host copy + device copy made at the next compression.  A formatted dictionary
    // (dictFull, validated on the device into `info`) also gives the frames their dictID and the first repcodes; its entropy
    // tables are not used (every block carries its own), which any decoder holding the dictionary accepts.
    std::vector<u8> dictHost, dictFull; DevBuf dict, dictFullDev, dictInfoDev; bool dictDirty = false, dictFormatted = false;
    DictInfo info = {};
    u64 dictGen = 0;            // bumped by every ZSTD_CCtx_loadDictionary: device workers copy the dictionary when theirs is older
    // ZSTDMI_CCtx_setDevices: one worker context per listed device (its own stream and workspaces there); a call's frames are
    // dealt to them in contiguous shares (compress_multi).  Empty = the context's own device only.
    std::vector<ZSTD_CCtx_s*> workers;
    DevBuf gatherIn, gatherOut;     // a many-range plan: the ranges of a kind side by side, and their output (compress_plan)
};
// History per chunk lives in LDS beside the chunk: up to 32 KiB of dictionary in front of 32 KiB chunks, or up to 60 KiB when
// the whole input fits behind it in one chunk (small records, the usual dictionary case).
constexpr size_t kDictKeep = 60u << 10;
static u32 round_tile(size_t n) { return (u32)((n + 4095) & ~(size_t)4095); }
// -> bytes of dictionary used as history for an input of srcSize bytes (0 = none)
static u32 dict_prefix_len(const ZSTD_CCtx* c, size_t srcSize)
{
    const size_t have = c->dictHost.size();
    if (have < 8) return 0;                                  // ZSTD_compress_insertDictionary ignores dictionaries < 8 bytes, U/ZstdCompress.cs:5469-5477
    const size_t wide = have < kDictKeep ? have : kDictKeep;
    if (srcSize <= kChunkSize - round_tile(wide)) return (u32)wide;
    return (u32)(have < (32u << 10) ? have : (32u << 10));
}

struct ZSTD_DCtx_s {
    int windowLogMax = 27;
    int device = 0; bool deviceOk = false;
    hipStream_t ownStream = nullptr, stream = nullptr;
    hipStream_t aux = nullptr; hipEvent_t auxDone = nullptr;     // the literal decoder beside seq_decode (decompress_device)
    int overlapMode = 0;        // ZSTDMI_DCtx_setOverlap: 0 = by block count, 1 = never, 2 = always
    bool lastWalkSerial = false; // the last call's frames were listed by the serial walk (ZSTDMI_debugLastWalkSerial)
    int execWaves = 0;          // ZSTDMI_DCtx_setExecWaves: waves per frame in exec_matches, 0 = by the number of frames
    DevBuf frames, blocks, recs, status, scratch, walkWs, slowFlags, stageSrc, stageDst, origin, originList;
    int originMode = 0;         // ZSTDMI_DCtx_setLongFrames: 0 = by cost (see decompress_device), 1 = never, 2 = every frame of 1 MiB or more
    StageTimer timer;
    // streaming adapter (ZSTD_decompressStream): whole frames are collected on the host, decoded in batches
    std::vector<u8> dIn, dOut; size_t dOutPos = 0; bool hostage = false;
    u32 litDecoder = 0;         // 0 auto, 1 serial (4 lanes per frame), 2 self-synchronising (256 lanes per frame), 3 serial with compact tables
    // dictionary (ZSTD_DCtx_loadDictionary): host copy, uploaded at the next decompression.  Raw content: the bytes are the
    // history.  Formatted (magic 0xEC30A437): dict_parse_kernel validates the header and fills `info`; the history is the content.
    std::vector<u8> dictHost; DevBuf dict, dictInfoDev; bool dictDirty = false, dictFormatted = false;
    DictInfo info = {};
    u64 dictGen = 0;
    std::vector<ZSTD_DCtx_s*> workers;      // ZSTDMI_DCtx_setDevices (decompress_multi)
};


static size_t cctx_sync_dictionary(ZSTD_CCtx* c);
static size_t cctx_bind(ZSTD_CCtx* c)
{
    if (!c) return ZERR(kErrGeneric);
    if (!c->deviceOk) {
        if (device_count() <= c->device) return ZERR(kErrInitMissing);       // no gfx950 device: fail loudly, never fall back
        if (hipSetDevice(c->device) != hipSuccess) return ZERR(kErrInitMissing);
        if (!c->ownStream && hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess) return ZERR(kErrMemoryAllocation);
        if (!c->stream) c->stream = c->ownStream;
        c->deviceOk = true;
    } else if (hipSetDevice(c->device) != hipSuccess) return ZERR(kErrInitMissing);
    return 0;
}
static size_t dctx_sync_dictionary(ZSTD_DCtx* d);
static size_t dctx_bind(ZSTD_DCtx* d)
{
    if (!d) return ZERR(kErrGeneric);
    if (!d->deviceOk) {
        if (device_count() <= d->device) return ZERR(kErrInitMissing);
        if (hipSetDevice(d->device) != hipSuccess) return ZERR(kErrInitMissing);
        if (!d->ownStream && hipStreamCreateWithFlags(&d->ownStream, hipStreamNonBlocking) != hipSuccess) return ZERR(kErrMemoryAllocation);
        if (!d->aux && hipStreamCreateWithFlags(&d->aux, hipStreamNonBlocking) != hipSuccess) return ZERR(kErrMemoryAllocation);
        if (!d->auxDone && hipEventCreateWithFlags(&d->auxDone, hipEventDisableTiming) != hipSuccess) return ZERR(kErrMemoryAllocation);
        if (!d->stream) d->stream = d->ownStream;
        d->deviceOk = true;
    } else if (hipSetDevice(d->device) != hipSuccess) return ZERR(kErrInitMissing);
    return 0;
}

// The parameters one compression call runs with: the context's sticky ones (ZSTD_compress2, ZSTD_compressStream2) or, for
// ZSTD_compressCCtx, the level alone with default frame parameters and no dictionary (U/ZstdCompress.cs:5751-5776:
// compress_usingDict(NULL) builds its parameters from the level and leaves the context's requested ones untouched).
struct CallParams {
    int level = 3, checksumFlag = 0, contentSizeFlag = 1, dictIDFlag = 1, strategy = 0, targetLength = 0, windowLog = 0, searchLog = 0;
    int minMatch = 0, chainLog = 0;     // accepted by the setters only at the value the kernels implement for the level/strategy in force THEN: checked again per call
    bool useDict = true;
};
static CallParams sticky_params(const ZSTD_CCtx* c)
{
    CallParams p; p.level = c->level; p.checksumFlag = c->checksumFlag; p.contentSizeFlag = c->contentSizeFlag; p.dictIDFlag = c->dictIDFlag;
    p.strategy = c->strategy; p.targetLength = c->targetLength; p.windowLog = c->windowLog; p.searchLog = c->searchLog; p.minMatch = c->minMatch; p.chainLog = c->chainLog; p.useDict = true;
    return p;
}

// What a call resolves to: the reference's cParams for (level, chunk size) with the explicitly set strategy /
// targetLength on top (ZSTD_overrideCParams, U/ZstdCompress.cs:2096-2127), then the parts of them the kernels act on.
struct Resolved { CParams cp; u32 finder, minStrideLog, rawLiterals; };
static Resolved resolve_call(const CallParams& p, size_t srcSize, u32 chunkBytes)
{
    Resolved r;
    r.cp = get_cparams(p.level, srcSize < chunkBytes ? srcSize : chunkBytes);
    if (p.strategy) r.cp.strategy = (u32)p.strategy;
    if (p.targetLength) r.cp.targetLength = (u32)p.targetLength;
    if (p.searchLog) r.cp.searchLog = (u32)p.searchLog;
    // match finder by strategy (U/ZstdCompress.cs:3397-3417 selects the block compressor the same way): fast; doubleFast -> the
    // dual-hash finder; greedy and everything above it -> dual-hash + lazy deferral (no lazy2 / binary-tree / optimal parsers)
    r.finder = r.cp.strategy <= kStratFast ? 0u : r.cp.strategy == kStratDfast ? 1u : 2u;
    // ZSTD_fast probes two of every targetLength + 2 positions once a step is set (negative levels; U/ZstdFast.cs:101-103,
    // 130-136) but falls back to every position right after each match; the tile finder's counterpart is a floor under its
    // probing stride (a power of two, fixed per 4-16 KiB), set at half the reference's density so that it does not lose more
    // ratio than the reference's own step does (measured against the oracle at levels -5 and -20 in tests/test_gpu_boundary.py)
    r.minStrideLog = 0;
    if (r.cp.strategy == kStratFast && r.cp.targetLength > 5) {
        const u32 gap = (r.cp.targetLength + 2) / 4;
        r.minStrideLog = cp_highbit32(gap); if (r.minStrideLog > 4) r.minStrideLog = 4;
    }
    r.rawLiterals = literals_compression_disabled(r.cp) ? 1u : 0u;
    return r;
}

static bool cctx_workspace(ZSTD_CCtx* c, u32 nChunks)
{
    return c->seqs.ensure((size_t)nChunks * kMaxSeq * sizeof(Seq)) && c->lits.ensure((size_t)nChunks * kLitStride + 64) &&
           c->meta.ensure((size_t)nChunks * sizeof(ChunkMeta)) && c->tables.ensure((size_t)nChunks * sizeof(HufTable)) &&
           c->slots.ensure((size_t)nChunks * kSlotStride + 64) && c->offsets.ensure((size_t)nChunks * sizeof(u64)) &&
           c->total.ensure(64);
}
// the region parse of the fast strategy keeps one candidate position (u16) per input byte between its two steps (lz_fast.hip)
// region parse: candidates (u16 per position of a chunk's 64 KiB image) [+ the hash chains of the level >= 5 finder, same size] + the work list
static size_t cand_plane_bytes(u32 nChunks) { return (size_t)nChunks * kChunkSize * sizeof(u16) + 256; }
static bool cctx_cand_workspace(ZSTD_CCtx* c, u32 nChunks, bool chains) { return c->cand.ensure(cand_plane_bytes(nChunks) * (chains ? 2 : 1) + ((size_t)nChunks + 1) * sizeof(u32)); }

// upload a newly loaded dictionary; a formatted one is first validated on the device (ZSTD_loadCEntropy's checks are those of
// ZSTD_loadDEntropy plus the symbol-coverage rules that only matter to an encoder reusing the tables) -> dictionary_corrupted
static size_t cctx_sync_dictionary(ZSTD_CCtx* c)
{
    if (!c->dictDirty) return 0;
    hipStream_t s = c->stream;
    if (c->dictFormatted) {
        const size_t n = c->dictFull.size();
        if (!c->dictFullDev.ensure(n + 64) || !c->dictInfoDev.ensure(sizeof(DictInfo))) return ZERR(kErrMemoryAllocation);
        if (hipMemcpyAsync(c->dictFullDev.p, c->dictFull.data(), n, hipMemcpyHostToDevice, s) != hipSuccess) return ZERR(kErrGeneric);
        launch_dict_parse((const u8*)c->dictFullDev.p, (u32)n, (DictInfo*)c->dictInfoDev.p, s);
        if (hipMemcpyAsync(&c->info, c->dictInfoDev.p, sizeof(DictInfo), hipMemcpyDeviceToHost, s) != hipSuccess) return ZERR(kErrGeneric);
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
        if (c->info.err) { c->dictFull.clear(); c->dictHost.clear(); c->dictFormatted = false; c->dictDirty = false; return ZERR(kErrDictionaryCorrupted); }
        const size_t keep = c->info.contentSize < kDictKeep ? c->info.contentSize : kDictKeep;
        c->dictHost.assign(c->dictFull.end() - (ptrdiff_t)keep, c->dictFull.end());
    }
    if (!c->dictHost.empty()) {
        if (!c->dict.ensure(c->dictHost.size() + 64)) return ZERR(kErrMemoryAllocation);
        if (hipMemcpyAsync(c->dict.p, c->dictHost.data(), c->dictHost.size(), hipMemcpyHostToDevice, s) != hipSuccess) return ZERR(kErrGeneric);
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
    }
    c->dictDirty = false;
    return 0;
}

// How a range of paramSize bytes is cut into blocks and frames (a function of the parameters, the loaded dictionary and that size):
// bytes of dictionary in front of every chunk, bytes per block, blocks per frame (0 = every block a frame of its own), and what
// the level resolves to for it.
struct Framing { u32 prefixLen, chunkBytes, frameBlocks; Resolved rs; u32 indepWindowLog = 0; size_t span() const { return (size_t)chunkBytes * (frameBlocks ? frameBlocks : 1u); } };
static Framing resolve_framing(const ZSTD_CCtx* c, const CallParams& cp, size_t paramSize)
{
    const u32 prefixLen = cp.useDict ? dict_prefix_len(c, paramSize) : 0u;
    u32 chunkBytes = kChunkSize - round_tile(prefixLen);
    // ZSTD_c_windowLog 10 .. 15: independent frames of 1 << windowLog bytes (the reference cuts blocks at the window size and lets
    // no offset exceed it, U/ZstdCompress.cs:4690-4712, U/ZstdCompressInternal.cs:787-813; a frame that IS its own window does both)
    u32 indepWindowLog = 0;
    if (cp.windowLog >= 10 && cp.windowLog < (int)kChunkLog && chunkBytes > (1u << cp.windowLog)) { chunkBytes = 1u << cp.windowLog; indepWindowLog = (u32)cp.windowLog; }
    Resolved rs = resolve_call(cp, paramSize, chunkBytes);
    // Cross-chunk history (the window the block loop carries, U/ZstdCompress.cs:4705-4807): blocks of 64 KiB - hist
    // bytes, each with the hist bytes in front of it as match-only history in LDS, frameBlocks of them to a frame (so a
    // match never reaches out of its frame and frames stay independent units for the decoder and for sharding).  Without a
    // dictionary only (a dictionary's tail takes the same place in LDS).
    u32 frameBlocks = 0;
    // a frame never declares more than the window the caller asked for (its content size is its window)
    const u32 frameBytes = (cp.windowLog >= (int)kChunkLog && cp.windowLog < 31 && ((u64)1 << cp.windowLog) < c->frameBytes) ? (1u << cp.windowLog) : c->frameBytes;
    if (prefixLen == 0 && paramSize > kChunkSize && chunkBytes == kChunkSize && frameBytes > kChunkSize) {
        int hb = c->historyBytes;
        // by level: the doubleFast levels (3-4; 3 is the library's default level) stage 16 KiB of history per 48 KiB block (one
        // third more staging and hashing for three quarters of what 32 KiB buy), greedy and above 32 KiB per 32 KiB block
        if (hb < 0) hb = (rs.cp.strategy > kStratFast || cp.windowLog > (int)kChunkLog) ? (rs.cp.strategy == kStratDfast ? (16 << 10) : (32 << 10)) : 0;
        if (hb > 0) {
            // what the level resolves to at the frame's size decides the form: the fast strategy keeps full 64 KiB blocks and
            // finds far matches through its table (candidates in front of the block are verified against global memory, up to
            // 188 KiB back); the dual-hash finders' 16-bit tables cannot hold far positions, so their blocks shrink to
            // 64 KiB - hist and carry the hist bytes in front of them in LDS
            const Resolved rf = resolve_call(cp, paramSize < frameBytes ? paramSize : frameBytes, frameBytes);
            if (rf.finder == 0) chunkBytes = kChunkSize;
            else { const u32 histB = round_tile((size_t)hb) > (48u << 10) ? (48u << 10) : round_tile((size_t)hb); chunkBytes = kChunkSize - histB; }
            frameBlocks = frameBytes / chunkBytes; if (frameBlocks < 2) frameBlocks = 2;
            rs = rf;
        }
    }
    // Small calls at the fast strategy (64 KiB < size <= kSmallCall, everything left to the level): a call of 10 MiB is 160 chunks on
    // 256 CUs, and what it waits for is ONE chunk's serial chains — the tANS states of seq_encode and, on the way back, of seq_decode
    // (~270 ns a sequence, 1.0 of a 2.4 ms round trip), then the Huffman streams.  Frames stay 64 KiB (match execution is ordered per
    // frame) but are cut into four blocks of 16 KiB, each behind the frame's earlier blocks in LDS (the history form of the dual-hash
    // levels): same window, four times as many chains a quarter as long, for a block header, a Huffman table and the unknown-repcode
    // start per 16 KiB (text: + 1.5 % of size).  The finder restages the history per block — on a chip that a call this size leaves idle.
    constexpr size_t kSmallCall = (size_t)32 << 20;
    if (!frameBlocks && prefixLen == 0 && chunkBytes == kChunkSize && c->historyBytes < 0 && cp.windowLog == 0 && rs.finder == 0 && rs.minStrideLog == 0 &&
        paramSize > kChunkSize && paramSize <= kSmallCall) {
        chunkBytes = 16u << 10; frameBlocks = kChunkSize / chunkBytes;
    }
    // ZSTD_c_windowLog 10 .. 15 (continued): the blocks of 1 << windowLog bytes are independent of each other but share frames of 64 KiB
    // with a window descriptor of exactly that windowLog — a frame per block would cost 13 bytes per KiB on incompressible input, more
    // than ZSTD_compressBound grants (the reference spends a 3-byte block header per window)
    if (indepWindowLog && paramSize > chunkBytes) frameBlocks = kChunkSize / chunkBytes; else indepWindowLog = 0;
    Framing f; f.prefixLen = prefixLen; f.chunkBytes = chunkBytes; f.frameBlocks = frameBlocks; f.rs = rs; f.indepWindowLog = indepWindowLog;
    return f;
}

// the compress pipeline over device-resident buffers: one range of the input with one set of parameters (see compress_device)
// paramSize: the size the parameters are resolved for — the whole range's, of which [d_src, d_src + srcSize) may be a frame-aligned
// part (a device worker's share of the range, compress_multi): what is written for a stretch of frames depends on nothing else
// markAt / marks (optional): input offsets (multiples of the frame span, ascending) whose place in the output is wanted -> marks[i]
static size_t compress_range(ZSTD_CCtx* c, const CallParams& cp, u8* d_dst, size_t dstCapacity, const u8* d_src, size_t srcSize, size_t paramSize, bool& first,
                             const std::vector<size_t>* markAt = nullptr, std::vector<u64>* marks = nullptr)
{
    hipStream_t s = c->stream;
    if (srcSize == 0) {     // ZSTD_writeEpilogue on an empty frame: header (FCS=0, single segment) + empty raw last block
        u8 f[13]; size_t n = 0;
        f[n++] = 0x28; f[n++] = 0xB5; f[n++] = 0x2F; f[n++] = 0xFD;
        if (cp.contentSizeFlag) { f[n++] = (u8)((cp.checksumFlag ? 4 : 0) | 0x20); f[n++] = 0; }
        else { f[n++] = (u8)(cp.checksumFlag ? 4 : 0); f[n++] = (u8)(((cp.windowLog >= 10 ? cp.windowLog : 10) - 10) << 3); }   // window descriptor, no content size
        f[n++] = 1; f[n++] = 0; f[n++] = 0;
        if (cp.checksumFlag) { f[n++] = 0x99; f[n++] = 0xE9; f[n++] = 0xD8; f[n++] = 0x51; }   // XXH64("") low 32 bits = 0x51D8E999
        if (dstCapacity < n) return ZERR(kErrDstSizeTooSmall);
        if (hipMemcpyAsync(d_dst, f, n, hipMemcpyHostToDevice, s) != hipSuccess) return ZERR(kErrGeneric);
        (void)hipStreamSynchronize(s);
        return n;
    }
    if (cp.useDict) { const size_t e = cctx_sync_dictionary(c); if (isErr(e)) return e; }
    const Framing fr = resolve_framing(c, cp, paramSize);
    const u32 prefixLen = fr.prefixLen, chunkBytes = fr.chunkBytes, frameBlocks = fr.frameBlocks;
    const u32 lzFrameBlocks = frameBlocks | (fr.indepWindowLog ? 0x80000000u : 0u);       // (independent blocks: no history between them)
    const u32 hdrWindow = fr.indepWindowLog;
    const Resolved rs = fr.rs;
    // a formatted dictionary: its dictID in every frame header (unless ZSTD_c_dictIDFlag = 0), its repcodes in front of every frame
    const bool fmtDict = cp.useDict && c->dictFormatted;
    const u32 dictID = fmtDict ? c->info.dictID : 0u;
    const u32 dictIdBytes = (dictID && cp.dictIDFlag) ? (dictID < 256 ? 1u : dictID < 65536 ? 2u : 4u) : 0u;
    const u32 plainReps[3] = { 1, 4, 8 };
    const u32* const initReps = fmtDict ? c->info.rep : plainReps;
    const u8* prefix = prefixLen ? (const u8*)c->dict.p + (c->dictHost.size() - prefixLen) : nullptr;
    const u64 totalChunks = (srcSize + chunkBytes - 1) / chunkBytes;
    u32 passChunks = (u32)(totalChunks < c->passChunks ? totalChunks : c->passChunks);
    if (frameBlocks && passChunks < totalChunks) { passChunks -= passChunks % frameBlocks; if (!passChunks) passChunks = frameBlocks; }     // frames never straddle passes
    if (!cctx_workspace(c, passChunks)) return ZERR(kErrMemoryAllocation);
    const bool regionParse = rs.minStrideLog == 0 && !(rs.finder == 0 && frameBlocks && chunkBytes >= kChunkSize) && c->parser == 0;   // (not the far-candidate finder)
    const bool hcChains = regionParse && rs.finder >= 2;
    // attempts per position of the level >= 5 search: the reference's 1 << searchLog (U/ZstdLazy.cs:641-642), between 4 and 32; the
    // greedy and lazy strategies (levels 5-7) stop at 8 unless ZSTD_c_searchLog asks for more: measured on text, 32 attempts
    // instead of 8 cost twice the time for 1 % of size
    u32 hcDepth = rs.cp.searchLog < 2 ? 4u : rs.cp.searchLog > 5 ? 32u : 1u << rs.cp.searchLog;
    if (hcDepth > 8 && rs.cp.strategy <= 4 && cp.searchLog == 0) hcDepth = 8;
    if (regionParse && !cctx_cand_workspace(c, passChunks, hcChains)) return ZERR(kErrMemoryAllocation);
    const u32 strategy = rs.cp.strategy < kStratGreedy ? rs.cp.strategy : (u32)kStratGreedy;      // ZSTD_selectEncodingType's < lazy heuristic is the one seq_encode holds (U/ZstdCompressSequences.cs:400-469): levels whose strategy is lazy or above get greedy's constants
    size_t produced = 0;
    for (u64 c0 = 0; c0 < totalChunks; c0 += passChunks) {
        const u32 nChunks = (u32)((totalChunks - c0) < passChunks ? (totalChunks - c0) : passChunks);
        const u8* src = d_src + c0 * chunkBytes;
        const u64 n = (srcSize - c0 * chunkBytes) < (u64)nChunks * chunkBytes ? (srcSize - c0 * chunkBytes) : (u64)nChunks * chunkBytes;
        Seq* seqs = (Seq*)c->seqs.p; u8* lits = (u8*)c->lits.p; ChunkMeta* meta = (ChunkMeta*)c->meta.p;
        HufTable* tables = (HufTable*)c->tables.p; u8* slots = (u8*)c->slots.p; u64* offsets = (u64*)c->offsets.p; u64* total = (u64*)c->total.p;
        c->timer.begin(s);
        launch_lz(rs.finder, src, n, nChunks, seqs, lits, meta, prefix, prefixLen, chunkBytes, dictIdBytes | (cp.contentSizeFlag ? 0u : 0x100u) | (hdrWindow << 12), rs.minStrideLog, lzFrameBlocks, regionParse ? (u16*)c->cand.p : nullptr, hcChains ? (u16*)((u8*)c->cand.p + cand_plane_bytes(passChunks)) : nullptr,
                  regionParse ? (u32*)((u8*)c->cand.p + cand_plane_bytes(passChunks) * (hcChains ? 2 : 1)) : nullptr, hcDepth, s, c->timer.hook(), (u32*)(total + 4));      // (the claim counter: a word of `total`'s 64 bytes)
        launch_huf_build(lits, meta, tables, slots, nChunks, rs.rawLiterals, src, chunkBytes, s, c->timer.hook());
        if (cp.checksumFlag) { launch_xxh64(src, n, meta, nChunks, chunkBytes, frameBlocks, s);             c->timer.mark("xxh64", s); }
        launch_seq_encode(seqs, meta, slots, nChunks, strategy, (cp.checksumFlag ? 1u : 0u) | (cp.contentSizeFlag ? 0u : 2u) | (hdrWindow << 8), 1, dictID, dictIdBytes, initReps, frameBlocks, chunkBytes, n, s);   c->timer.mark("seq_encode", s);
        launch_scan_sizes(meta, nChunks, offsets, total, s);                       c->timer.mark("scan", s);
        const size_t room = dstCapacity > produced ? dstCapacity - produced : 0;
        // the literals section (most of the output) is encoded straight into its final place; gather moves the rest
        launch_huf_encode(lits, meta, tables, slots, d_dst + produced, offsets, room, nChunks, src, chunkBytes, s);   c->timer.mark("huf_encode", s);
        launch_gather(src, n, slots, meta, offsets, d_dst + produced, room, nChunks, chunkBytes, s);      c->timer.mark("gather", s);
        u64 passTotal = 0;
        if (hipMemcpyAsync(&passTotal, total, sizeof(u64), hipMemcpyDeviceToHost, s) != hipSuccess) return ZERR(kErrGeneric);
        if (markAt) for (size_t i = 0; i < markAt->size(); ++i) {           // output offsets of the chunks that start at the marked inputs
            const u64 ck = (*markAt)[i] / chunkBytes;
            if (ck >= c0 && ck < c0 + nChunks && hipMemcpyAsync(&(*marks)[i], offsets + (ck - c0), sizeof(u64), hipMemcpyDeviceToHost, s) != hipSuccess) return ZERR(kErrGeneric);
        }
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
        if (markAt) for (size_t i = 0; i < markAt->size(); ++i) { const u64 ck = (*markAt)[i] / chunkBytes; if (ck >= c0 && ck < c0 + nChunks) (*marks)[i] += produced; }
        // stage times of the call = sums over its passes (inputs above 1 GiB take several)
        c->timer.finish(); c->nStages = c->timer.n;
        for (int i = 0; i < c->timer.n; i++) { c->stageMs[i] = (first ? 0.f : c->stageMs[i]) + c->timer.ms[i]; c->stageNames[i] = c->timer.names[i]; }
        first = false;
        if (passTotal > room) return ZERR(kErrDstSizeTooSmall);
        produced += (size_t)passTotal;
        c->lastChunks = nChunks; c->lastSrc = src; c->lastChunkBytes = chunkBytes;
    }
    return produced;
}

// Input the match finder gets nothing out of (BASELINE's Zipf bytes, random or already compressed data): the history, smaller blocks
// and deeper search of the levels >= 3 only cost there — Zipf at level 5 came out 0.8 % LARGER than at level 1 (a frame header share
// and a Huffman table per 32 KiB instead of per 64 KiB) at an eighth of the speed, and in a mixed input such stretches took a third
// of the match finder's time.  So a call of 4 MiB or more that leaves strategy, window and history to the level is first looked at
// in groups of 16 frames (~4 MiB): lz_probe_kernel counts, in eight 4 KiB tiles per group, the positions that repeat an earlier one of
// their tile or of the 60 KiB in front of it (text: several hundred per tile; Zipf bytes: a handful; an eighth of the input is read).
// Runs of groups below 32 per tile — the whole call, or at least 8 MiB of it — are compressed as level 1 would (its finder,
// independent 64 KiB frames: the same bytes level 1 writes for them), the rest by the level's own path; a group boundary is a frame
// boundary of both.  The decision is a function of the data alone: probe (device, per group) -> plan (host) -> ranges.
constexpr u32 kProbeTiles = 8;
constexpr size_t kMinRunBytes = (size_t)8 << 20;     // a stretch without matches becomes a range of its own from this length on
struct PlanRange { size_t off, len; bool sparse; };

// does the call get probed, and in groups of how many bytes?  (0 = no probe: one range, the call's own parameters)
static size_t probe_group_bytes(ZSTD_CCtx* c, const CallParams& cp, size_t srcSize, size_t& err)
{
    err = 0;
    // (a caller-set targetLength keeps the level's own path: at the fast strategy it means raw literals, which the sparse ranges must not inherit)
    if (!(srcSize >= (4u << 20) && c->historyBytes < 0 && cp.strategy == 0 && cp.windowLog == 0 && cp.searchLog == 0 && cp.targetLength == 0)) return 0;
    if (cp.useDict) { err = cctx_sync_dictionary(c); if (isErr(err)) return 0; err = 0; }
    if (cp.useDict && dict_prefix_len(c, srcSize)) return 0;
    if (resolve_call(cp, srcSize, kChunkSize).cp.strategy <= kStratFast) return 0;
    const Framing fr = resolve_framing(c, cp, srcSize);
    return (size_t)16 * (fr.frameBlocks ? fr.span() : (size_t)kChunkSize * 4);      // a multiple of 64 KiB
}
// counts of the groups of [d_src, d_src + len); `front` = bytes of the input readable in front of d_src (a worker's share of a call)
static size_t probe_run(ZSTD_CCtx* c, const u8* d_src, size_t len, size_t front, size_t group, u32* counts)
{
    const u32 nGroups = (u32)((len + group - 1) / group);
    if (!c->probe.ensure((size_t)nGroups * sizeof(u32))) return ZERR(kErrMemoryAllocation);
    launch_lz_probe(d_src, len, front, group, nGroups, kProbeTiles, (u32*)c->probe.p, c->stream);
    if (hipMemcpyAsync(counts, c->probe.p, (size_t)nGroups * sizeof(u32), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return ZERR(kErrGeneric);
    if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
    return 0;
}
static void plan_ranges(const std::vector<u32>& counts, size_t group, size_t srcSize, std::vector<PlanRange>& out)
{
    const u32 nGroups = (u32)counts.size();
    // a range is a pass of its own: a stretch without matches counts if it is the whole call or at least kMinRunBytes long (round 2
    // asked for 64 MiB, because the passes of a many-range plan ran one after the other and none filled the chip: the mixed bench
    // input's 12.8 MiB pieces took 1.7 x the time as ranges of their own; now the ranges of a kind are compressed as one pass, compress_plan)
    std::vector<u8> sp(nGroups);
    for (u32 g = 0; g < nGroups; ++g) sp[g] = counts[g] < kProbeTiles * 32;
    const u32 minRun = (u32)((kMinRunBytes + group - 1) / group);
    for (u32 g = 0; g < nGroups; ) {
        u32 e = g + 1;
        while (e < nGroups && sp[e] == sp[g]) ++e;
        if (sp[g] && e - g < minRun && !(g == 0 && e == nGroups)) for (u32 k = g; k < e; ++k) sp[k] = 0;
        g = e;
    }
    for (u32 g = 0; g < nGroups; ) {
        const bool sparse = sp[g] != 0;
        u32 e = g + 1;
        while (e < nGroups && (sp[e] != 0) == sparse) ++e;
        PlanRange r; r.off = (size_t)g * group; r.len = (e == nGroups ? srcSize : (size_t)e * group) - r.off; r.sparse = sparse;
        out.push_back(r);
        g = e;
    }
}
static size_t check_call_params(const CallParams& cp)
{
    if (cp.minMatch || cp.chainLog) {       // set under another level or strategy than the call runs with: refuse, never ignore
        const CParams lv = get_cparams(cp.level, kChunkSize);
        if (cp.minMatch && cp.minMatch != (int)kernel_min_match(cp.strategy ? (u32)cp.strategy : lv.strategy)) return ZERR(kErrParameterUnsupported);
        if (cp.chainLog && cp.chainLog != (int)lv.chainLog) return ZERR(kErrParameterUnsupported);
    }
    return 0;
}

// run f(0 .. n - 1), one host thread each (f(0) on the caller's): a device worker's calls block on its own stream
// (nothing may leave a thread as an exception: -> false, and the caller reports memory_allocation)
template <class F> static bool run_on_workers(size_t n, F f)
{
    std::vector<std::thread> th;
    std::vector<u8> bad(n, 0);
    th.reserve(n);
    auto one = [&f, &bad](size_t i) { try { f(i); } catch (...) { bad[i] = 1; } };
    bool ok = true;
    for (size_t i = 1; i < n; ++i) { try { th.emplace_back(one, i); } catch (...) { ok = false; break; } }
    if (ok) one(0);
    for (auto& t : th) t.join();
    for (size_t i = 0; i < n; ++i) ok = ok && !bad[i];
    return ok;
}
static size_t compress_multi(ZSTD_CCtx* c, const CallParams& cp, void* dst, size_t dstCapacity, const void* src, size_t srcSize);

// A plan of several ranges (a mixed input: stretches the level's finder is for, stretches without matches) as it stands is a run of
// small passes — a handful of kernels over a few hundred chunks each, none of which fills the chip; side by side on several
// streams they only queue for the CUs (the finders hold a CU's LDS alone; measured: slower than one after the other).  So the
// ranges of a KIND are gathered into one contiguous buffer and compressed as ONE pass sequence per kind (range lengths are whole
// probe groups, i.e. whole frames of either framing; the call's tail is the last range of its kind), each range's place in that
// output is read back with the pass (marks), and the pieces are copied into the caller's buffer in the order of the input.  Two
// pass sequences instead of one per range; the bytes are those of the ranges one after the other.  paramSize: the whole call's.
static size_t compress_plan(ZSTD_CCtx* c, const CallParams& cp, const std::vector<PlanRange>& plan, size_t paramSize, u8* d_dst, size_t dstCapacity, const u8* d_src, bool& first)
{
    const size_t R = plan.size();
    CallParams cpSparse = cp; cpSparse.level = 1;
    size_t len[2] = {0, 0}, cnt[2] = {0, 0};
    for (const PlanRange& r : plan) { len[r.sparse] += r.len; cnt[r.sparse]++; }
    // the kinds' inputs, contiguous (a kind with one range is read where it lies)
    size_t inAt[2] = {0, len[0] + 256};
    const size_t inNeed = (cnt[0] > 1 ? len[0] + 256 : 0) + (cnt[1] > 1 ? len[1] + 256 : 0);
    if (inNeed && !c->gatherIn.ensure(inAt[1] + len[1] + 256)) return ZERR(kErrMemoryAllocation);
    size_t bound[2], outAt[2] = {0, 0};
    for (int k = 0; k < 2; ++k) bound[k] = len[k] ? ((ZSTD_compressBound(len[k]) + (len[k] >> 12) + 4096) & ~(size_t)255) : 0;
    outAt[1] = bound[0];
    if (!c->gatherOut.ensure(bound[0] + bound[1] + 256)) return ZERR(kErrMemoryAllocation);
    std::vector<size_t> markAt[2]; std::vector<u64> marks[2];
    { size_t fill[2] = {0, 0};
      for (const PlanRange& r : plan) {
          const int k = r.sparse;
          markAt[k].push_back(fill[k]);
          if (cnt[k] > 1 && hipMemcpyAsync((u8*)c->gatherIn.p + inAt[k] + fill[k], d_src + r.off, r.len, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return ZERR(kErrGeneric);
          fill[k] += r.len;
      } }
    size_t got[2] = {0, 0};
    for (int k = 0; k < 2; ++k) {
        if (!len[k]) continue;
        marks[k].assign(markAt[k].size(), 0);
        const u8* in = (const u8*)c->gatherIn.p + inAt[k];
        if (cnt[k] == 1) for (const PlanRange& r : plan) if ((int)r.sparse == k) in = d_src + r.off;
        const size_t n = compress_range(c, k ? cpSparse : cp, (u8*)c->gatherOut.p + outAt[k], bound[k], in, len[k], paramSize, first, &markAt[k], &marks[k]);
        if (isErr(n)) return n;
        got[k] = n;
    }
    if (got[0] + got[1] > dstCapacity) return ZERR(kErrDstSizeTooSmall);
    size_t pos = 0, seen[2] = {0, 0};
    for (size_t r = 0; r < R; ++r) {
        const int k = plan[r].sparse;
        const size_t i = seen[k]++;
        const u64 lo = marks[k][i], hi = i + 1 < marks[k].size() ? marks[k][i + 1] : got[k];
        if (hipMemcpyAsync(d_dst + pos, (const u8*)c->gatherOut.p + outAt[k] + lo, (size_t)(hi - lo), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) return ZERR(kErrGeneric);
        pos += (size_t)(hi - lo);
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
    c->lastChunks = 0;
    return pos;
}

static size_t compress_device(ZSTD_CCtx* c, const CallParams& cp, u8* d_dst, size_t dstCapacity, const u8* d_src, size_t srcSize)
{
    bool first = true;
    { const size_t e = check_call_params(cp); if (isErr(e)) return e; }
    size_t err = 0;
    const size_t group = probe_group_bytes(c, cp, srcSize, err);
    if (isErr(err)) return err;
    if (!group) return compress_range(c, cp, d_dst, dstCapacity, d_src, srcSize, srcSize, first);
    std::vector<u32> counts((srcSize + group - 1) / group);
    { const size_t e = probe_run(c, d_src, srcSize, 0, group, counts.data()); if (isErr(e)) return e; }
    std::vector<PlanRange> plan;
    plan_ranges(counts, group, srcSize, plan);
    if (plan.size() == 1) { CallParams one = cp; if (plan[0].sparse) one.level = 1; return compress_range(c, one, d_dst, dstCapacity, d_src, srcSize, srcSize, first); }
    return compress_plan(c, cp, plan, srcSize, d_dst, dstCapacity, d_src, first);
}


// ======================================================================================================
extern "C" {

ZSTD_CCtx* ZSTD_createCCtx(void) { return new (std::nothrow) ZSTD_CCtx_s(); }

size_t ZSTD_freeCCtx(ZSTD_CCtx* c)
{
    if (!c) return 0;
    for (ZSTD_CCtx* w : c->workers) (void)ZSTD_freeCCtx(w);
    c->workers.clear();
    if (c->deviceOk) {
        (void)hipSetDevice(c->device);
        if (c->ownStream) (void)hipStreamSynchronize(c->ownStream);
        c->seqs.release(); c->lits.release(); c->meta.release(); c->tables.release(); c->slots.release(); c->cand.release(); c->probe.release();
        c->gatherIn.release(); c->gatherOut.release(); c->offsets.release(); c->total.release(); c->stageSrc.release(); c->stageDst.release(); c->dict.release(); c->dictFullDev.release(); c->dictInfoDev.release();
        c->timer.destroy();
        if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    }
    delete c;
    return 0;
}

int ZSTD_minCLevel(void) { return -(1 << 17); }
int ZSTD_maxCLevel(void) { return 22; }
int ZSTD_defaultCLevel(void) { return 3; }

size_t ZSTD_CCtx_setParameter(ZSTD_CCtx* c, int param, int value)
{
    if (!c) return ZERR(kErrGeneric);
    switch (param) {
    case ZSTD_c_compressionLevel:           // clamped, 0 means default (U/ZstdCompress.cs:886-905)
        if (value < ZSTD_minCLevel()) value = ZSTD_minCLevel();
        if (value > ZSTD_maxCLevel()) value = ZSTD_maxCLevel();
        c->level = value == 0 ? 3 : value;
        return c->level >= 0 ? (size_t)c->level : 0;        /* a size_t cannot carry a negative level (U/ZstdCompress.cs:899-905) */
    case ZSTD_c_checksumFlag:    if (value < 0 || value > 1) return ZERR(kErrParameterOutOfBound); c->checksumFlag = value; return (size_t)value;
    case ZSTD_c_contentSizeFlag: if (value < 0 || value > 1) return ZERR(kErrParameterOutOfBound);   /* 0: window descriptor instead of the content size */
                                 c->contentSizeFlag = value; return (size_t)value;
    case ZSTD_c_dictIDFlag:      if (value < 0 || value > 1) return ZERR(kErrParameterOutOfBound); c->dictIDFlag = value; return (size_t)value;
    case ZSTD_c_nbWorkers:       if (value != 0) return ZERR(kErrParameterUnsupported); return 0;   /* as U/ZstdCompress.cs:1064-1072 */
    case ZSTD_c_windowLog:       if (value != 0 && (value < 10 || value > 31)) return ZERR(kErrParameterOutOfBound);
                                 c->windowLog = value; return (size_t)value;            /* 10 .. 15: frames of 1 << windowLog bytes; 17+: frames of at most that */
    // Match-finder parameters (bounds: ZSTD_cParam_getBounds, U/ZstdCompress.cs:444-700).  0 = "from the level".  A value the
    // kernels implement is accepted and stored; any other value within bounds is parameter_unsupported, never silently ignored.
    case ZSTD_c_strategy:        // every strategy maps onto one of the three finders (resolve_call)
        if (value < 0 || value > 9) return ZERR(kErrParameterOutOfBound);
        c->strategy = value; return (size_t)value;
    case ZSTD_c_targetLength:    // fast strategy: acceleration (the probing stride) and, when > 0, raw literals; the other strategies' finders have no
                                 // counterpart of it (accepted and unused there, as the reference's greedy/lazy levels 5-12 ignore it: U/ZstdLazy.cs)
        if (value < 0 || value > (1 << 17)) return ZERR(kErrParameterOutOfBound);
        c->targetLength = value; return (size_t)value;
    case ZSTD_c_hashLog:         // the LDS tables have 2^13 buckets whatever the level's row says
        if (value != 0 && (value < 6 || value > 30)) return ZERR(kErrParameterOutOfBound);
        if (value != 0 && value != (int)kKernelHashLog) return ZERR(kErrParameterUnsupported);
        c->hashLog = value; return (size_t)value;
    case ZSTD_c_minMatch: {      // width of the finder's hash: 6 (fast), 5 (the others)
        if (value != 0 && (value < 3 || value > 7)) return ZERR(kErrParameterOutOfBound);
        const CParams cp = get_cparams(c->level, kChunkSize);
        if (value != 0 && value != (int)kernel_min_match(c->strategy ? (u32)c->strategy : cp.strategy)) return ZERR(kErrParameterUnsupported);
        c->minMatch = value; return (size_t)value; }
    case ZSTD_c_chainLog: {      // the chain table covers the finder's whole 64 KiB window: only the level's own value is accepted
        if (value != 0 && (value < 6 || value > 30)) return ZERR(kErrParameterOutOfBound);
        const CParams cp = get_cparams(c->level, kChunkSize);
        if (value != 0 && value != (int)cp.chainLog) return ZERR(kErrParameterUnsupported);
        c->chainLog = value; return (size_t)value; }
    case ZSTD_c_searchLog:       // attempts per position of the greedy/lazy search = 1 << searchLog (used from 2 to 5: 4 .. 32 attempts)
        if (value != 0 && (value < 1 || value > 30)) return ZERR(kErrParameterOutOfBound);
        c->searchLog = value; return (size_t)value;
    default: return ZERR(kErrParameterUnsupported);
    }
}

size_t ZSTD_CCtx_getParameter(const ZSTD_CCtx* c, int param, int* value)
{
    if (!c || !value) return ZERR(kErrGeneric);
    switch (param) {
    case ZSTD_c_compressionLevel: *value = c->level; return 0;
    case ZSTD_c_checksumFlag: *value = c->checksumFlag; return 0;
    case ZSTD_c_contentSizeFlag: *value = c->contentSizeFlag; return 0;
    case ZSTD_c_dictIDFlag: *value = c->dictIDFlag; return 0;
    case ZSTD_c_nbWorkers: *value = 0; return 0;
    case ZSTD_c_windowLog: *value = c->windowLog; return 0;
    case ZSTD_c_hashLog: *value = c->hashLog; return 0;            // the requested values, 0 = from the level (U/ZstdCompress.cs:1100-1150)
    case ZSTD_c_chainLog: *value = c->chainLog; return 0;
    case ZSTD_c_searchLog: *value = c->searchLog; return 0;
    case ZSTD_c_minMatch: *value = c->minMatch; return 0;
    case ZSTD_c_targetLength: *value = c->targetLength; return 0;
    case ZSTD_c_strategy: *value = c->strategy; return 0;
    default: return ZERR(kErrParameterUnsupported);
    }
}

// ZSTD_compress_insertDictionary, U/ZstdCompress.cs:5465-5503: without the magic the bytes are raw content
// (ZSTD_loadDictionaryContent, :5126-5237) and the frames carry no dictID, exactly as the reference writes them; with it, a
// formatted dictionary (ZSTD_loadZstdDictionary, :5402-5463).
static size_t ZSTD_CCtx_loadDictionary_impl(ZSTD_CCtx* c, const void* dict, size_t dictSize)
{
    if (!c) return ZERR(kErrGeneric);
    if (!c->sIn.empty() || c->sEnding) return ZERR(kErrStageWrong);        /* not in the middle of a streaming frame session, U/ZstdCompress.cs:1273 */
    c->dictGen++;
    c->dictFormatted = false; c->dictFull.clear();
    if (dict == nullptr || dictSize == 0) { c->dictHost.clear(); c->dictDirty = true; return 0; }          /* "no dictionary" */
    if (dictSize > (size_t)1 << 30) return ZERR(kErrParameterUnsupported);
    u8 head[8] = {};
    const bool dev = is_device_ptr(dict);
    if (dev) { if (hipMemcpy(head, dict, dictSize < 8 ? dictSize : 8, hipMemcpyDeviceToHost) != hipSuccess) return ZERR(kErrGeneric); }
    else memcpy(head, dict, dictSize < 8 ? dictSize : 8);
    if (is_formatted_dictionary(head, dictSize)) {
        // ZSTD_loadZstdDictionary, U/ZstdCompress.cs:5402-5463: dictID + repcodes + content are used (see ZSTD_CCtx_s::dictHost)
        std::vector<u8> full(dictSize);
        if (dev) { if (hipMemcpy(full.data(), dict, dictSize, hipMemcpyDeviceToHost) != hipSuccess) return ZERR(kErrGeneric); }
        else memcpy(full.data(), dict, dictSize);
        c->dictFull.swap(full); c->dictHost.clear(); c->dictFormatted = true; c->dictDirty = true;
        if (!isErr(cctx_bind(c))) return cctx_sync_dictionary(c);          // validated now when a device is there, else at first use
        return 0;
    }
    const size_t keep = dictSize < kDictKeep ? dictSize : kDictKeep;
    std::vector<u8> h(dictSize < 8 ? dictSize : keep);
    const u8* tail = (const u8*)dict + (dictSize - h.size());
    if (dev) { if (hipMemcpy(h.data(), tail, h.size(), hipMemcpyDeviceToHost) != hipSuccess) return ZERR(kErrGeneric); }
    else memcpy(h.data(), tail, h.size());
    c->dictHost.swap(h);
    c->dictDirty = true;
    return 0;
}

size_t ZSTD_compressBound(size_t n) { return n + (n >> 8) + (n < (128u << 10) ? (((128u << 10) - n) >> 11) : 0); }

static size_t ZSTDMI_compressDevice_impl(ZSTD_CCtx* c, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize)
{
    size_t e = cctx_bind(c); if (isErr(e)) return e;
    if (srcSize && !d_src) return ZERR(kErrSrcSizeWrong);
    if (!d_dst && dstCapacity) return ZERR(kErrDstBufferNull);
    if (!d_dst) return ZERR(kErrDstSizeTooSmall);
    if (c->workers.size() > 1 && srcSize) return compress_multi(c, sticky_params(c), d_dst, dstCapacity, d_src, srcSize);
    return compress_device(c, sticky_params(c), (u8*)d_dst, dstCapacity, (const u8*)d_src, srcSize);
}

static size_t compress_any(ZSTD_CCtx* c, const CallParams& cp, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    size_t e = cctx_bind(c); if (isErr(e)) return e;
    if (srcSize && !src) return ZERR(kErrSrcSizeWrong);
    if (!dst) return ZERR(kErrDstSizeTooSmall);
    if (c->workers.size() > 1 && srcSize) return compress_multi(c, cp, dst, dstCapacity, src, srcSize);
    const bool srcDev = srcSize ? is_device_ptr(src) : true, dstDev = is_device_ptr(dst);
    const u8* d_src = (const u8*)src; u8* d_dst = (u8*)dst;
    size_t devCap = dstCapacity;
    if (!srcDev) {
        if (!c->stageSrc.ensure(srcSize + 64)) return ZERR(kErrMemoryAllocation);
        if (hipMemcpyAsync(c->stageSrc.p, src, srcSize, hipMemcpyHostToDevice, c->stream) != hipSuccess) return ZERR(kErrGeneric);
        d_src = (const u8*)c->stageSrc.p;
    }
    if (!dstDev) {
        const size_t worst = ZSTD_compressBound(srcSize) + 32;
        devCap = dstCapacity < worst ? dstCapacity : worst;
        if (!c->stageDst.ensure(devCap + 64)) return ZERR(kErrMemoryAllocation);
        d_dst = (u8*)c->stageDst.p;
    }
    const size_t r = compress_device(c, cp, d_dst, devCap, d_src, srcSize);
    if (isErr(r)) return r;
    if (!dstDev) {
        if (hipMemcpyAsync(dst, d_dst, r, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return ZERR(kErrGeneric);
        if (hipStreamSynchronize(c->stream) != hipSuccess) return ZERR(kErrGeneric);
    }
    return r;
}

size_t ZSTDMI_compressDevice(ZSTD_CCtx* c, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize) { return guarded([&] { return ZSTDMI_compressDevice_impl(c, d_dst, dstCapacity, d_src, srcSize); }); }
size_t ZSTD_compress2(ZSTD_CCtx* c, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    if (!c) return ZERR(kErrGeneric);
    return guarded([&] { return compress_any(c, sticky_params(c), dst, dstCapacity, src, srcSize); });
}

size_t ZSTD_compressCCtx(ZSTD_CCtx* c, void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level)
{
    if (!c) return ZERR(kErrGeneric);
    // ZSTD_compressCCtx = ZSTD_compress_usingDict(dict = NULL, level) (U/ZstdCompress.cs:5751-5776): the level alone, default
    // frame parameters (content size, no checksum), NO dictionary even if one is loaded; the context's sticky parameters and
    // its dictionary stay as they are for later ZSTD_compress2 calls.
    CallParams p; p.level = level == 0 ? 3 : (level < ZSTD_minCLevel() ? ZSTD_minCLevel() : level > ZSTD_maxCLevel() ? ZSTD_maxCLevel() : level);
    p.useDict = false;
    return guarded([&] { return compress_any(c, p, dst, dstCapacity, src, srcSize); });
}

/* S/CompressionStream.cs:41, S/DecompressionStream.cs:41 size their buffers with these (U/ZstdCompress.cs:6241-6249,
 * U/ZstdDecompress.cs:2096-2104): one block in, one compressed block + header + checksum out */
size_t ZSTD_CStreamInSize(void)  { return (size_t)1 << 17; }
size_t ZSTD_CStreamOutSize(void) { return ZSTD_compressBound((size_t)1 << 17) + 3 + 4; }
size_t ZSTD_DStreamInSize(void)  { return ((size_t)1 << 17) + 3; }
size_t ZSTD_DStreamOutSize(void) { return (size_t)1 << 17; }

// ---------------- decompression ----------------
ZSTD_DCtx* ZSTD_createDCtx(void) { return new (std::nothrow) ZSTD_DCtx_s(); }
size_t ZSTD_freeDCtx(ZSTD_DCtx* d)
{
    if (!d) return 0;
    for (ZSTD_DCtx* w : d->workers) (void)ZSTD_freeDCtx(w);
    d->workers.clear();
    if (d->deviceOk) {
        (void)hipSetDevice(d->device);
        if (d->ownStream) (void)hipStreamSynchronize(d->ownStream);
        d->frames.release(); d->blocks.release(); d->recs.release(); d->status.release(); d->scratch.release(); d->walkWs.release(); d->slowFlags.release(); d->stageSrc.release(); d->stageDst.release(); d->dict.release(); d->dictInfoDev.release(); d->origin.release(); d->originList.release();
        d->timer.destroy();
        if (d->aux) { (void)hipStreamSynchronize(d->aux); (void)hipStreamDestroy(d->aux); }
        if (d->auxDone) (void)hipEventDestroy(d->auxDone);
        if (d->ownStream) (void)hipStreamDestroy(d->ownStream);
    }
    delete d;
    return 0;
}
size_t ZSTD_DCtx_setParameter(ZSTD_DCtx* d, int param, int value)
{
    if (!d) return ZERR(kErrGeneric);
    if (param == ZSTD_d_windowLogMax) { if (value != 0 && (value < 10 || value > 31)) return ZERR(kErrParameterOutOfBound); d->windowLogMax = value ? value : 27; return 0; }
    return ZERR(kErrParameterUnsupported);
}
size_t ZSTD_DCtx_getParameter(ZSTD_DCtx* d, int param, int* value)
{
    if (!d || !value) return ZERR(kErrGeneric);
    if (param == ZSTD_d_windowLogMax) { *value = d->windowLogMax; return 0; }
    return ZERR(kErrParameterUnsupported);
}
// ZSTD_decompress_insertDictionary, U/ZstdDecompress.cs:1909-1931: without the magic the bytes are raw content, history in
// front of every frame (ZSTD_refDictContent, :1758-1771); with it (0xEC30A437) the header's Huffman and FSE tables and
// repcodes are what every frame starts from and frames must name its dictID or none (ZSTD_loadDEntropy, :1773-1875).
static size_t ZSTD_DCtx_loadDictionary_impl(ZSTD_DCtx* d, const void* dict, size_t dictSize)
{
    if (!d) return ZERR(kErrGeneric);
    d->dictGen++;
    if (dict == nullptr || dictSize == 0) { d->dictHost.clear(); d->dictDirty = true; return 0; }
    if (dictSize > (size_t)1 << 30) return ZERR(kErrParameterUnsupported);
    std::vector<u8> h(dictSize);
    if (is_device_ptr(dict)) {
        if (hipMemcpy(h.data(), dict, dictSize, hipMemcpyDeviceToHost) != hipSuccess) return ZERR(kErrGeneric);
    } else memcpy(h.data(), dict, dictSize);
    d->dictFormatted = is_formatted_dictionary(h.data(), dictSize);
    d->dictHost.swap(h);
    d->dictDirty = true;
    if (d->dictFormatted && !isErr(dctx_bind(d))) return dctx_sync_dictionary(d);      // validated now when a device is there, else at first use
    d->dictDirty = true;
    return 0;
}

// Host-side header walk for host buffers (ZSTD_findFrameSizeInfo, U/ZstdDecompress.cs:877-951): headers only, no payload.
static size_t host_frame_size_info(const u8* src, size_t srcSize, unsigned long long* bound)
{
    auto rd32 = [](const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); };
    if (srcSize >= 8 && (rd32(src) & 0xFFFFFFF0u) == 0x184D2A50u) {
        const u64 sz = (u64)rd32(src + 4) + 8;
        if (sz > srcSize) return ZERR(kErrSrcSizeWrong);
        *bound = 0; return (size_t)sz;
    }
    if (srcSize < 5) return ZERR(kErrSrcSizeWrong);
    if (rd32(src) != 0xFD2FB528u) return ZERR(kErrPrefixUnknown);
    const u8 fhd = src[4];
    static const size_t did[4] = { 0, 1, 2, 4 }, fcsB[4] = { 0, 2, 4, 8 };
    const u32 single = (fhd >> 5) & 1, fcsId = fhd >> 6;
    const size_t fhs = 5 + !single + did[fhd & 3] + fcsB[fcsId] + (single && !fcsId);
    if (srcSize < fhs) return ZERR(kErrSrcSizeWrong);
    if (fhd & 0x08) return ZERR(kErrFrameParameterUnsupported);
    size_t pos = 5; u64 windowSize = 0, fcs = ~0ull;
    if (!single) { const u8 wl = src[pos++]; const u32 wlog = (wl >> 3) + 10; if (wlog > 31) return ZERR(kErrWindowTooLarge); windowSize = 1ull << wlog; windowSize += (windowSize >> 3) * (wl & 7); }
    pos += did[fhd & 3];
    switch (fcsId) {
    case 0: if (single) fcs = src[pos]; break;
    case 1: fcs = (u64)((u32)src[pos] | ((u32)src[pos + 1] << 8)) + 256; break;
    case 2: fcs = rd32(src + pos); break;
    default: fcs = (u64)rd32(src + pos) | ((u64)rd32(src + pos + 4) << 32); break;
    }
    if (single) windowSize = fcs;
    const u64 blockSizeMax = windowSize < (1u << 17) ? windowSize : (1u << 17);
    const u8* ip = src + fhs; size_t remaining = srcSize - fhs; u64 nbBlocks = 0;
    for (;;) {
        if (remaining < 3) return ZERR(kErrSrcSizeWrong);
        const u32 bh = (u32)ip[0] | ((u32)ip[1] << 8) | ((u32)ip[2] << 16);
        const u32 last = bh & 1, type = (bh >> 1) & 3; u32 cSize = bh >> 3;
        if (type == 3) return ZERR(kErrCorruption);
        if (type == 1) cSize = 1;
        if (3 + (size_t)cSize > remaining) return ZERR(kErrSrcSizeWrong);
        ip += 3 + cSize; remaining -= 3 + cSize; nbBlocks++;
        if (last) break;
    }
    if ((fhd >> 2) & 1) { if (remaining < 4) return ZERR(kErrSrcSizeWrong); ip += 4; }
    *bound = fcs != ~0ull ? fcs : nbBlocks * blockSizeMax;
    return (size_t)(ip - src);
}

// Window size a frame header declares (ZSTD_getFrameHeader_advanced, U/ZstdDecompress.cs:462-634): the window descriptor, or the
// content size of a single-segment frame.  0 = not a zstd frame header, or not all of it is there yet.
static u64 host_frame_window(const u8* src, size_t srcSize)
{
    auto rd32 = [](const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); };
    if (srcSize < 5 || rd32(src) != 0xFD2FB528u) return 0;
    const u8 fhd = src[4];
    static const size_t did[4] = { 0, 1, 2, 4 }, fcsB[4] = { 0, 2, 4, 8 };
    const u32 single = (fhd >> 5) & 1, fcsId = fhd >> 6;
    const size_t fhs = 5 + !single + did[fhd & 3] + fcsB[fcsId] + (single && !fcsId);
    if (srcSize < fhs) return 0;
    if (!single) { const u8 wl = src[5]; const u32 wlog = (wl >> 3) + 10; if (wlog > 31) return ~0ull; const u64 w = 1ull << wlog; return w + (w >> 3) * (wl & 7); }
    const size_t pos = 5 + did[fhd & 3];
    switch (fcsId) {
    case 0: return src[pos];
    case 1: return (u64)((u32)src[pos] | ((u32)src[pos + 1] << 8)) + 256;
    case 2: return rd32(src + pos);
    default: return (u64)rd32(src + pos) | ((u64)rd32(src + pos + 4) << 32);
    }
}

static const u8* host_view(const void* src, size_t srcSize, std::vector<u8>& tmp)
{
    if (!is_device_ptr(src)) return (const u8*)src;
    tmp.resize(srcSize);
    if (hipMemcpy(tmp.data(), src, srcSize, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;
    return tmp.data();
}

static unsigned long long ZSTD_decompressBound_impl(const void* src, size_t srcSize)
{
    std::vector<u8> tmp; const u8* ip = srcSize ? host_view(src, srcSize, tmp) : (const u8*)src;
    if (srcSize && !ip) return (unsigned long long)0 - 2;
    unsigned long long bound = 0;
    while (srcSize > 0) {
        unsigned long long b = 0; const size_t cs = host_frame_size_info(ip, srcSize, &b);
        if (isErr(cs)) return (unsigned long long)0 - 2;
        ip += cs; srcSize -= cs; bound += b;
    }
    return bound;
}
static size_t ZSTD_findFrameCompressedSize_impl(const void* src, size_t srcSize)
{
    std::vector<u8> tmp; const u8* ip = host_view(src, srcSize, tmp);
    if (!ip) return ZERR(kErrSrcSizeWrong);
    unsigned long long b; return host_frame_size_info(ip, srcSize, &b);
}
static unsigned long long ZSTD_getFrameContentSize_impl(const void* src, size_t srcSize)
{
    std::vector<u8> tmp; const size_t look = srcSize < 18 ? srcSize : 18;
    const u8* ip = look ? host_view(src, look, tmp) : nullptr;
    if (!ip || look < 5) return (unsigned long long)0 - 2;
    auto rd32 = [](const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24); };
    if ((rd32(ip) & 0xFFFFFFF0u) == 0x184D2A50u) return 0;
    if (rd32(ip) != 0xFD2FB528u) return (unsigned long long)0 - 2;
    const u8 fhd = ip[4]; static const size_t did[4] = { 0, 1, 2, 4 }, fcsB[4] = { 0, 2, 4, 8 };
    const u32 single = (fhd >> 5) & 1, fcsId = fhd >> 6;
    const size_t fhs = 5 + !single + did[fhd & 3] + fcsB[fcsId] + (single && !fcsId);
    if (look < fhs) return (unsigned long long)0 - 2;
    size_t pos = 5 + !single + did[fhd & 3];
    switch (fcsId) {
    case 0: return single ? ip[pos] : (unsigned long long)0 - 1;
    case 1: return (u64)((u32)ip[pos] | ((u32)ip[pos + 1] << 8)) + 256;
    case 2: return rd32(ip + pos);
    default: return (u64)rd32(ip + pos) | ((u64)rd32(ip + pos + 4) << 32);
    }
}

// upload a newly loaded dictionary; a formatted one is validated on the device (ZSTD_loadDEntropy's checks) -> dictionary_corrupted
static size_t dctx_sync_dictionary(ZSTD_DCtx* d)
{
    if (!d->dictDirty) return 0;
    hipStream_t s = d->stream;
    if (!d->dictHost.empty()) {
        if (!d->dict.ensure(d->dictHost.size() + 64) || !d->dictInfoDev.ensure(sizeof(DictInfo))) return ZERR(kErrMemoryAllocation);
        if (hipMemcpyAsync(d->dict.p, d->dictHost.data(), d->dictHost.size(), hipMemcpyHostToDevice, s) != hipSuccess) return ZERR(kErrGeneric);
        if (d->dictFormatted) {
            launch_dict_parse((const u8*)d->dict.p, (u32)d->dictHost.size(), (DictInfo*)d->dictInfoDev.p, s);
            if (hipMemcpyAsync(&d->info, d->dictInfoDev.p, sizeof(DictInfo), hipMemcpyDeviceToHost, s) != hipSuccess) return ZERR(kErrGeneric);
        }
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
        if (d->dictFormatted && d->info.err) { d->dictHost.clear(); d->dictFormatted = false; d->dictDirty = false; return ZERR(kErrDictionaryCorrupted); }
    }
    d->dictDirty = false;
    return 0;
}

// The decompress pipeline over device-resident buffers.  Two host round trips size the work lists (frames + blocks after the
// counting walk, sequence records after the block pre-pass); everything else is one launch sequence:
//   walk (count) | walk (emit) -> block_parse -> block_link -> seq_scan | seq_decode -> block_offsets [-> frame_rescan]
//   -> decode_literals -> place_literals -> exec_matches
static size_t decompress_device(ZSTD_DCtx* d, u8* d_dst, size_t dstCapacity, const u8* d_src, size_t srcSize)
{
    hipStream_t s = d->stream;
    if (srcSize == 0) return 0;
    // a frame is at least 9 bytes; our own streams hold one per 64 KiB, foreign ones usually far fewer
    const u32 maxFrames = (u32)((srcSize / 9 + 1) < (1u << 26) ? (srcSize / 9 + 1) : (1u << 26));
    if (!d->status.ensure(kStWords * sizeof(u32)) || !d->walkWs.ensure(decode_walk_workspace_bytes(srcSize))) return ZERR(kErrMemoryAllocation);
    u32* status = (u32*)d->status.p;
    { const size_t e = dctx_sync_dictionary(d); if (isErr(e)) return e; }
    const bool fmt = d->dictFormatted && !d->dictHost.empty();
    const u8* const dictFull = fmt ? (const u8*)d->dict.p : nullptr;
    const DictInfo* const dinfo = fmt ? (const DictInfo*)d->dictInfoDev.p : nullptr;
    const u8* const dictContent = d->dictHost.empty() ? nullptr : (const u8*)d->dict.p + (fmt ? d->info.contentOff : 0u);
    const u32 dictContentSize = d->dictHost.empty() ? 0u : (fmt ? d->info.contentSize : (u32)d->dictHost.size());
    const u32 dictID = fmt ? d->info.dictID : 0u;
    auto read_status = [&](u32* st) -> bool {
        if (hipMemcpyAsync(st, status, kStWords * sizeof(u32), hipMemcpyDeviceToHost, s) != hipSuccess) return false;
        if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return false; }
        return true;
    };
    d->timer.begin(s);
    {   // error key = "none" (all ones); everything else zero
        u32 init[kStWords] = {}; init[kStErrKeyLo] = 0xFFFFFFFFu; init[kStErrKeyHi] = 0xFFFFFFFFu;
        if (hipMemcpyAsync(status, init, sizeof init, hipMemcpyHostToDevice, s) != hipSuccess) return ZERR(kErrGeneric);
    }
    u32 st[kStWords] = {};
    launch_frame_walk_count(d_src, srcSize, maxFrames, status, (u8*)d->walkWs.p, s);
    if (!read_status(st)) return ZERR(kErrGeneric);
    const bool serialWalk = !st[kStUsable];
    d->lastWalkSerial = serialWalk;
    if (serialWalk) {   // the segment links did not close: take the exact serial walk (it also yields the reference's error code)
        launch_frame_walk_serial(d_src, srcSize, nullptr, nullptr, maxFrames, status, dictID, 0, s);
        if (!read_status(st)) return ZERR(kErrGeneric);
    }
    if (st[kStErr]) return ZERR(st[kStErr]);
    const u32 nFrames = st[kStFrames], nBlocks = st[kStBlocks], nUnsized = st[kStUnsized];
    const u64 total = (u64)st[kStTotalLo] | ((u64)st[kStTotalHi] << 32);       // content sizes (bounds for frames without one)
    if (!nUnsized && total > dstCapacity) return ZERR(kErrDstSizeTooSmall);
    if (nFrames == 0) { d->timer.finish(); return 0; }
    if (!d->frames.ensure((size_t)nFrames * sizeof(FrameDesc)) || !d->blocks.ensure((size_t)nBlocks * sizeof(BlockDesc) + 64) ||
        !d->scratch.ensure((size_t)total + (size_t)nFrames * kLitSkew + 256) || !d->slowFlags.ensure((size_t)nBlocks + 64)) return ZERR(kErrMemoryAllocation);
    FrameDesc* frames = (FrameDesc*)d->frames.p; BlockDesc* blocks = (BlockDesc*)d->blocks.p;
    if (serialWalk) launch_frame_walk_serial(d_src, srcSize, frames, blocks, maxFrames, status, dictID, 1, s);
    else            launch_frame_walk_emit(d_src, srcSize, frames, blocks, (u8*)d->walkWs.p, s);
    d->timer.mark("frame_walk", s);
    // The literal decoder and seq_decode need nothing of each other (a block's Huffman streams and its FSE chains).  With few
    // blocks neither fills the chip — both are serial chains per block — so below kOverlapBlocks the literal decoder may run beside
    // seq_decode on a stream of its own (`early`: block_link then lets it write only the outputs whose place is known by now).
    // Whether it does is decided once the pre-pass has counted both kinds of work (below).
    constexpr u32 kOverlapBlocks = 12288;
    const bool early = d->overlapMode == 2 || (d->overlapMode == 0 && nBlocks <= kOverlapBlocks);
    launch_block_prepass(d_src, frames, blocks, nFrames, nBlocks, fmt ? 1u : 0u, early ? 1u : 0u, status, s);
    if (!read_status(st)) return ZERR(kErrGeneric);
    d->timer.mark("block_prepass", s);
    const u64 nSeq = (u64)st[kStSeqLo] | ((u64)st[kStSeqHi] << 32);
    if (!d->recs.ensure((size_t)(nSeq + 64) * sizeof(SeqRec))) return ZERR(kErrMemoryAllocation);
    // Long frames (decode_origin.hip).  The ordered walk of exec_matches moves a frame at about kWalkRate, all frames at
    // once; the origin path sweeps the frames it is given at about kSweepRate together.  A frame belongs on the origin path when its
    // own walk would outlast the sweep of every frame at least as long: the smallest size class 2^(20+k) with
    // 2^(20+k) / kWalkRate >= bytes(frames >= 2^(20+k)) / kSweepRate, from the per-class sums block_link filed.
    u64 originMin = 0, originBytes = 0, originLongest = 0; u32 originCap = 0;
    // (the walk has 64 x W sequences in flight per frame, W waves by the number of frames: measured on 1 - 4 MiB level-5 frames)
    const int execWaves = d->execWaves ? d->execWaves : nFrames <= 256 ? 16 : nFrames <= 512 ? 8 : nFrames <= 1024 ? 4 : nFrames <= 2048 ? 2 : 1;
    if (d->originMode != 1) {
        const double kWalkRate = execWaves >= 16 ? 0.42e9 : execWaves == 8 ? 0.33e9 : execWaves == 4 ? 0.24e9 : execWaves == 2 ? 0.19e9 : 0.15e9;
        constexpr double kSweepRate = 20e9;
        u64 above = 0;
        u64 sums[12];
        for (int k = 0; k < 12; ++k) sums[k] = (u64)st[kStBigBins + 2 * k] | ((u64)st[kStBigBins + 2 * k + 1] << 32);
        for (int k = 11; k >= 0; --k) {
            above += sums[k];
            if (!above) continue;
            const double size = (double)((u64)1 << (20 + k));
            if (d->originMode == 2 || size / kWalkRate >= (double)above / kSweepRate) { originMin = (u64)1 << (20 + k); originBytes = above; }
        }
        if (originMin) {
            // how many frames that can be (a frame of class k holds at least 2^(20+k) bytes) and how long the longest (below 2^(21+k))
            u64 cap = 0;
            for (int k = 0; k < 12; ++k) if (((u64)1 << (20 + k)) >= originMin && sums[k]) { cap += sums[k] >> (20 + k); originLongest = (u64)1 << (21 + k); }
            if (originLongest > originBytes) originLongest = originBytes;
            originCap = (u32)(cap < 65535 ? cap : 65535);       // (a grid dimension; more long frames than that keep the walk)
            // (+ one word per 1024 origins: origin_jump_kernel's finished regions)
            if (!d->origin.ensure((size_t)(originBytes + 1024 * (u64)originCap) * sizeof(u32) + (size_t)((originBytes + 1024 * (u64)originCap) / 1024 + 64) * sizeof(u32)) ||
                !d->originList.ensure((size_t)originCap * sizeof(u32))) { originMin = 0; (void)hipGetLastError(); }
        }
    }
    SeqRec* recs = (SeqRec*)d->recs.p;
    struct AuxGuard { hipStream_t a; bool on; ~AuxGuard() { if (on) (void)hipStreamSynchronize(a); } } auxGuard{ d->aux, false };   // nothing of this call outlives it
    // Beside each other only when both are substantial (from five coded literal bytes per sequence): a Huffman symbol costs its chain
    // about 26 ns per literal byte (four streams), a sequence about 270 ns — text (three or four literal bytes per sequence) has nothing to hide behind seq_decode and only loses LDS
    // bandwidth to the company (measured: 1 GiB of 1 MiB level-5 frames, mixed corpus 14.7 -> 13.4 ms, text 15.5 -> 15.7 ms), and
    // input without sequences (Zipf bytes) has no seq_decode to hide behind.
    const u64 litBytes = (u64)st[kStLitLo] | ((u64)st[kStLitHi] << 32);
    const bool beside = early && nSeq && (d->overlapMode == 2 || (litBytes >= 5 * nSeq && nSeq >= 64 * (u64)nBlocks));
    if (getenv("ZMI_DEBUG")) fprintf(stderr, "zmi: blocks %u seqs %llu coded literal bytes %llu early %d beside %d\n", nBlocks, (unsigned long long)nSeq, (unsigned long long)litBytes, (int)early, (int)beside);
    if (beside) {               // (the host has just waited for the pre-pass: everything the literal decoder reads is there)
        launch_decode_literals(d_src, d_dst, (u8*)d->scratch.p, frames, blocks, nBlocks, status, (u8*)d->slowFlags.p, d->litDecoder, dictFull, dinfo, d->aux, StageHook());
        if (hipEventRecord(d->auxDone, d->aux) != hipSuccess) return ZERR(kErrGeneric);
        auxGuard.on = true;
    }
    launch_seq_decode(d_src, frames, blocks, nBlocks, recs, status, dictFull, dinfo, s);            d->timer.mark("seq_decode", s);
    launch_block_offsets(frames, blocks, nFrames, dinfo, nUnsized ? 1u : 0u, dstCapacity, status, s);  d->timer.mark("block_offsets", s);
    if (auxGuard.on) { if (hipStreamWaitEvent(s, d->auxDone, 0) != hipSuccess) return ZERR(kErrGeneric); d->timer.mark("decode_literals", s); }     // (what of it seq_decode did not cover)
    else launch_decode_literals(d_src, d_dst, (u8*)d->scratch.p, frames, blocks, nBlocks, status, (u8*)d->slowFlags.p, d->litDecoder, dictFull, dinfo, s, d->timer.hook());
    launch_place_literals(d_src, d_dst, (const u8*)d->scratch.p, frames, blocks, nBlocks, recs, status, s);    d->timer.mark("place_literals", s);
    if (originMin) {
        const u64 entries = originBytes + 1024 * (u64)originCap, longest = originLongest;
        u32* const origin = (u32*)d->origin.p; const u32* const list = (const u32*)d->originList.p;
        launch_origin_select(frames, nFrames, originMin, (u32*)d->originList.p, originCap, entries, status, s);
        launch_origin_init(frames, blocks, list, originCap, longest, recs, status, origin, dictContent ? dictContentSize : 0u, s);    d->timer.mark("origin_init", s);
        // The rounds in groups of six, the host looking at the last one's verdict in between: ordinary data settles in about ten
        // rounds, and a round that only finds out that nothing is left still costs its launch (the kernels check the flag too).
        for (u32 r = 0; r < kOriginRounds; r += 6) {
            launch_origin_jump(frames, list, originCap, longest, status, origin, origin + entries, r, r + 6, s);
            u32 open = 0;
            const u32 lastRound = (r + 6 < kOriginRounds ? r + 6 : kOriginRounds) - 1;
            if (hipMemcpyAsync(&open, status + kStOriginChanged + lastRound, sizeof(u32), hipMemcpyDeviceToHost, s) != hipSuccess) return ZERR(kErrGeneric);
            if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
            if (!open) break;
        }
        d->timer.mark("origin_jump", s);
        launch_origin_gather(frames, list, originCap, longest, status, origin, d_dst, dictContent, s);    d->timer.mark("origin_gather", s);
    }
    launch_exec_matches(d_src, d_dst, frames, blocks, nFrames, recs, status, dictContent, dictContentSize, s, execWaves);  d->timer.mark("exec_matches", s);
    if (!read_status(st)) return ZERR(kErrGeneric);
    d->timer.finish();
    if (st[kStErrKeyLo] != 0xFFFFFFFFu || st[kStErrKeyHi] != 0xFFFFFFFFu) return ZERR(st[kStErrKeyLo] & 0xFFFFu);   // the first failing block's first error
    if (st[kStErr]) return ZERR(st[kStErr]);                  // regenerated sizes of unsized frames exceed the destination
    if (nUnsized) return (size_t)((u64)st[kStActualLo] | ((u64)st[kStActualHi] << 32));
    return (size_t)total;
}

static size_t decompress_multi(ZSTD_DCtx* d, void* dst, size_t dstCapacity, const void* src, size_t srcSize);
static size_t ZSTDMI_decompressDevice_impl(ZSTD_DCtx* d, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize)
{
    size_t e = dctx_bind(d); if (isErr(e)) return e;
    if (srcSize && !d_src) return ZERR(kErrSrcSizeWrong);
    if (d->workers.size() > 1 && srcSize) return decompress_multi(d, d_dst, dstCapacity, d_src, srcSize);
    return decompress_device(d, (u8*)d_dst, dstCapacity, (const u8*)d_src, srcSize);
}

static size_t ZSTD_decompressDCtx_impl(ZSTD_DCtx* d, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    size_t e = dctx_bind(d); if (isErr(e)) return e;
    if (srcSize && !src) return ZERR(kErrSrcSizeWrong);
    if (srcSize == 0) return 0;
    if (d->workers.size() > 1) return decompress_multi(d, dst, dstCapacity, src, srcSize);
    const bool srcDev = is_device_ptr(src), dstDev = dst ? is_device_ptr(dst) : false;
    const u8* d_src = (const u8*)src; u8* d_dst = (u8*)dst;
    if (!srcDev) {
        if (!d->stageSrc.ensure(srcSize + 64)) return ZERR(kErrMemoryAllocation);
        if (hipMemcpyAsync(d->stageSrc.p, src, srcSize, hipMemcpyHostToDevice, d->stream) != hipSuccess) return ZERR(kErrGeneric);
        d_src = (const u8*)d->stageSrc.p;
    }
    if (!dstDev) {
        if (!d->stageDst.ensure(dstCapacity + 64)) return ZERR(kErrMemoryAllocation);
        d_dst = (u8*)d->stageDst.p;
    }
    const size_t r = decompress_device(d, d_dst, dstCapacity, d_src, srcSize);
    if (isErr(r)) return r;
    if (!dstDev && r) {
        if (hipMemcpyAsync(dst, d_dst, r, hipMemcpyDeviceToHost, d->stream) != hipSuccess) return ZERR(kErrGeneric);
        if (hipStreamSynchronize(d->stream) != hipSuccess) return ZERR(kErrGeneric);
    }
    return r;
}

// ---------------- several devices behind one context (SURVEY.md section 8 e; ZSTDMI_*_setDevices) ----------------
// north_star: "chunks partition naturally across the 8 GPUs of one node".  Frames are the independent unit (a match never leaves its
// frame), so a call's frames are dealt to the device workers in contiguous shares: every worker stages its share on its own device,
// compresses it with the kernels above on its own stream, from a host thread of its own, and the shares' outputs are copied into the
// caller's buffer one behind the other.  What is written does not depend on the number of workers: shares are cut on frame
// boundaries (on probe-group boundaries when the sparse-input probe runs), parameters are resolved for the whole range, and the probe's
// plan is made once over all shares' counts.  No collective: the only exchange is the final copy (device to host, or peer to peer).
static size_t copy_any(void* dst, const void* src, size_t n, hipStream_t s)
{
    if (!n) return 0;
    if (hipMemcpyAsync(dst, src, n, hipMemcpyDefault, s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
    return 0;
}

static size_t compress_multi(ZSTD_CCtx* c, const CallParams& cp, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    { const size_t e = check_call_params(cp); if (isErr(e)) return e; }
    const size_t W = c->workers.size();
    bool ok = true;
    // the workers run with the parent's settings and dictionary
    if (cp.useDict) { const size_t e = cctx_sync_dictionary(c); if (isErr(e)) return e; }
    for (ZSTD_CCtx* w : c->workers) {
        w->historyBytes = c->historyBytes; w->frameBytes = c->frameBytes; w->parser = c->parser; w->passChunks = c->passChunks; w->timer.enabled = c->timer.enabled;
        if (w->dictGen != c->dictGen) {
            w->dictHost = c->dictHost; w->dictFull = c->dictFull; w->dictFormatted = c->dictFormatted; w->info = c->info; w->dictDirty = true; w->dictGen = c->dictGen;
        }
    }
    size_t err = 0;
    ZSTD_CCtx* const w0 = c->workers[0];
    { const size_t e = cctx_bind(w0); if (isErr(e)) return e; }
    if (cp.useDict) { const size_t e = cctx_sync_dictionary(w0); if (isErr(e)) return e; }
    const size_t group = probe_group_bytes(w0, cp, srcSize, err);
    if (isErr(err)) return err;
    // shares: whole probe groups, or whole frames of the one range
    const size_t unit = group ? group : resolve_framing(w0, cp, srcSize).span();
    const size_t nUnits = (srcSize + unit - 1) / unit;
    std::vector<size_t> lo(W + 1);
    for (size_t i = 0; i <= W; ++i) { const size_t u = nUnits * i / W; lo[i] = u * unit < srcSize ? u * unit : srcSize; }
    lo[W] = srcSize;
    std::vector<size_t> res(W, 0), front(W, 0);
    std::vector<u32> counts(group ? (srcSize + group - 1) / group : 0);
    // phase A: every worker stages its share (with the probe's 60 KiB window in front of it) and, if the call is probed, counts its groups
    ok = run_on_workers(W, [&](size_t i) {
        ZSTD_CCtx* w = c->workers[i];
        size_t e = cctx_bind(w); if (isErr(e)) { res[i] = e; return; }
        const size_t n = lo[i + 1] - lo[i];
        if (!n) return;
        front[i] = group ? (lo[i] < (60u << 10) ? lo[i] : (60u << 10)) : 0;
        if (!w->stageSrc.ensure(front[i] + n + 64) || !w->stageDst.ensure(ZSTD_compressBound(n) + (n / unit + 2) * 64 + 64)) { res[i] = ZERR(kErrMemoryAllocation); return; }
        e = copy_any(w->stageSrc.p, (const u8*)src + lo[i] - front[i], front[i] + n, w->stream); if (isErr(e)) { res[i] = e; return; }
        if (group) { e = probe_run(w, (const u8*)w->stageSrc.p + front[i], n, front[i], group, counts.data() + lo[i] / group); if (isErr(e)) res[i] = e; }
    });
    if (!ok) return ZERR(kErrMemoryAllocation);
    for (size_t i = 0; i < W; ++i) if (isErr(res[i])) return res[i];
    std::vector<PlanRange> plan;
    if (group) plan_ranges(counts, group, srcSize, plan);
    else { PlanRange r; r.off = 0; r.len = srcSize; r.sparse = false; plan.push_back(r); }
    CallParams cpSparse = cp; cpSparse.level = 1;
    // phase B: every worker compresses what the plan's ranges hold of its share (parameters resolved for the whole call, as one device does)
    std::vector<size_t> produced(W, 0);
    ok = run_on_workers(W, [&](size_t i) {
        ZSTD_CCtx* w = c->workers[i];
        if (isErr(cctx_bind(w))) return;
        bool first = true;
        std::vector<PlanRange> part;
        for (const PlanRange& r : plan) {
            const size_t a = r.off > lo[i] ? r.off : lo[i], b = (r.off + r.len) < lo[i + 1] ? (r.off + r.len) : lo[i + 1];
            if (a < b) { PlanRange q; q.off = a - lo[i]; q.len = b - a; q.sparse = r.sparse; part.push_back(q); }
        }
        if (part.empty()) return;
        const u8* base = (const u8*)w->stageSrc.p + front[i];
        size_t n;
        if (part.size() == 1) n = compress_range(w, part[0].sparse ? cpSparse : cp, (u8*)w->stageDst.p, w->stageDst.cap, base + part[0].off, part[0].len, srcSize, first);
        else n = compress_plan(w, cp, part, srcSize, (u8*)w->stageDst.p, w->stageDst.cap, base, first);
        if (isErr(n)) { res[i] = n; return; }
        produced[i] = n;
    });
    if (!ok) return ZERR(kErrMemoryAllocation);
    for (size_t i = 0; i < W; ++i) if (isErr(res[i])) return res[i];
    size_t total = 0;
    for (size_t i = 0; i < W; ++i) total += produced[i];
    if (total > dstCapacity) return ZERR(kErrDstSizeTooSmall);
    // the shares, one behind the other, into the caller's buffer (host: device-to-host copies side by side; device: peer copies)
    std::vector<size_t> at(W, 0);
    for (size_t i = 1; i < W; ++i) at[i] = at[i - 1] + produced[i - 1];
    ok = run_on_workers(W, [&](size_t i) {
        ZSTD_CCtx* w = c->workers[i];
        if (isErr(cctx_bind(w))) return;
        size_t e = copy_any((u8*)dst + at[i], w->stageDst.p, produced[i], w->stream);
        if (!isErr(e) && hipStreamSynchronize(w->stream) != hipSuccess) { (void)hipGetLastError(); e = ZERR(kErrGeneric); }
        if (isErr(e)) res[i] = e;
    });
    if (!ok) return ZERR(kErrMemoryAllocation);
    for (size_t i = 0; i < W; ++i) if (isErr(res[i])) return res[i];
    // stage times: the first worker's (every worker runs the same sequence over its share)
    c->nStages = w0->nStages;
    for (int i = 0; i < w0->nStages; i++) { c->stageMs[i] = w0->stageMs[i]; c->stageNames[i] = w0->stageNames[i]; }
    c->lastChunks = 0;
    (void)cctx_bind(c);
    return total;
}

// decompress: the frames of the input (a host-side header walk) in contiguous shares by compressed size
static size_t decompress_multi(ZSTD_DCtx* d, void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    const size_t W = d->workers.size();
    bool ok = true;
    std::vector<u8> tmp;
    const u8* const ip = host_view(src, srcSize, tmp);
    if (!ip) return ZERR(kErrGeneric);
    struct Piece { size_t off, len; unsigned long long bound; bool sized; };
    std::vector<Piece> frames;
    { size_t pos = 0;
      while (pos < srcSize) {
          unsigned long long b = 0; const size_t fs = host_frame_size_info(ip + pos, srcSize - pos, &b);
          if (isErr(fs)) { if (frames.empty() || fs != ZERR(kErrPrefixUnknown)) return fs; return ZERR(kErrSrcSizeWrong); }     // as ZSTD_decompressMultiFrame: garbage behind a frame
          Piece p; p.off = pos; p.len = fs; p.bound = b;
          p.sized = ZSTD_getFrameContentSize_impl(ip + pos, fs) < (unsigned long long)0 - 2;
          frames.push_back(p); pos += fs;
      } }
    // shares of about equal compressed size
    std::vector<size_t> lo(W + 1, frames.size());
    { size_t acc = 0, k = 0; lo[0] = 0;
      for (size_t f = 0; f < frames.size(); ++f) { while (k + 1 < W && acc >= srcSize * (k + 1) / W) lo[++k] = f; acc += frames[f].len; }
      while (k + 1 < W) lo[++k] = frames.size(); lo[W] = frames.size(); }
    std::vector<size_t> res(W, 0), got(W, 0);
    std::vector<unsigned long long> bound(W, 0);
    bool allSized = true;
    for (size_t i = 0; i < W; ++i) for (size_t f = lo[i]; f < lo[i + 1]; ++f) { bound[i] += frames[f].bound; allSized = allSized && frames[f].sized; }
    if (allSized) { unsigned long long t = 0; for (size_t i = 0; i < W; ++i) t += bound[i]; if (t > dstCapacity) return ZERR(kErrDstSizeTooSmall); }
    { const size_t e = dctx_sync_dictionary(d); if (isErr(e)) return e; }
    for (ZSTD_DCtx* w : d->workers) {
        w->litDecoder = d->litDecoder; w->originMode = d->originMode; w->overlapMode = d->overlapMode; w->timer.enabled = d->timer.enabled;
        if (w->dictGen != d->dictGen) { w->dictHost = d->dictHost; w->dictFormatted = d->dictFormatted; w->dictDirty = true; w->dictGen = d->dictGen; }
    }
    std::vector<size_t> at(W, 0);
    for (size_t i = 1; i < W; ++i) at[i] = at[i - 1] + (size_t)bound[i - 1];       // exact when every frame has a content size
    ok = run_on_workers(W, [&](size_t i) {
        ZSTD_DCtx* w = d->workers[i];
        size_t e = dctx_bind(w); if (isErr(e)) { res[i] = e; return; }
        if (lo[i] == lo[i + 1]) return;
        const size_t a = frames[lo[i]].off, n = frames[lo[i + 1] - 1].off + frames[lo[i + 1] - 1].len - a;
        if (!w->stageSrc.ensure(n + 64) || !w->stageDst.ensure((size_t)bound[i] + 64)) { res[i] = ZERR(kErrMemoryAllocation); return; }
        e = copy_any(w->stageSrc.p, ip + a, n, w->stream); if (isErr(e)) { res[i] = e; return; }
        const size_t r = decompress_device(w, (u8*)w->stageDst.p, (size_t)bound[i], (const u8*)w->stageSrc.p, n);
        if (isErr(r)) { res[i] = r; return; }
        got[i] = r;
        if (allSized) {         // its place in the caller's buffer is known
            e = copy_any((u8*)dst + at[i], w->stageDst.p, r, w->stream);
            if (!isErr(e) && hipStreamSynchronize(w->stream) != hipSuccess) { (void)hipGetLastError(); e = ZERR(kErrGeneric); }
            if (isErr(e)) res[i] = e;
        }
    });
    if (!ok) return ZERR(kErrMemoryAllocation);
    for (size_t i = 0; i < W; ++i) if (isErr(res[i])) return res[i];          // the first share's error is the first frame's
    size_t total = 0;
    for (size_t i = 0; i < W; ++i) total += got[i];
    if (!allSized) {            // frames without a content size: the shares' places follow from what they regenerated
        if (total > dstCapacity) return ZERR(kErrDstSizeTooSmall);
        size_t pos = 0;
        for (size_t i = 0; i < W; ++i) {
            ZSTD_DCtx* w = d->workers[i];
            if (isErr(dctx_bind(w))) return ZERR(kErrGeneric);
            const size_t e = copy_any((u8*)dst + pos, w->stageDst.p, got[i], w->stream); if (isErr(e)) return e;
            if (hipStreamSynchronize(w->stream) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
            pos += got[i];
        }
    }
    d->timer.n = d->workers[0]->timer.n;
    for (int i = 0; i < d->timer.n; i++) { d->timer.ms[i] = d->workers[0]->timer.ms[i]; d->timer.names[i] = d->workers[0]->timer.names[i]; }
    (void)dctx_bind(d);
    return total;
}

// ---------------- errors ----------------
unsigned ZSTD_isError(size_t code) { return isErr(code); }
const char* ZSTD_getErrorName(size_t code)
{
    if (!isErr(code)) return "No error detected";
    switch ((u32)(0 - code)) {          // strings as U/ErrorPrivate.cs:35-180
    case kErrGeneric: return "Error (generic)";
    case kErrPrefixUnknown: return "Unknown frame descriptor";
    case kErrVersionUnsupported: return "Version not supported";
    case kErrFrameParameterUnsupported: return "Unsupported frame parameter";
    case kErrWindowTooLarge: return "Frame requires too much memory for decoding";
    case kErrCorruption: return "Corrupted block detected";
    case kErrChecksumWrong: return "Restored data doesn't match checksum";
    case kErrParameterUnsupported: return "Unsupported parameter";
    case kErrParameterOutOfBound: return "Parameter is out of bound";
    case kErrInitMissing: return "Context should be init first";
    case kErrMemoryAllocation: return "Allocation error : not enough memory";
    case kErrWorkSpaceTooSmall: return "workSpace buffer is not large enough";
    case kErrStageWrong: return "Operation not authorized at current processing stage";
    case kErrTableLogTooLarge: return "tableLog requires too much memory : unsupported";
    case kErrMaxSymbolValueTooLarge: return "Unsupported max Symbol Value : too large";
    case kErrMaxSymbolValueTooSmall: return "Specified maxSymbolValue is too small";
    case kErrDictionaryCorrupted: return "Dictionary is corrupted";
    case kErrDictionaryWrong: return "Dictionary mismatch";
    case 34: return "Cannot create Dictionary from provided samples";
    case kErrDstSizeTooSmall: return "Destination buffer is too small";
    case kErrSrcSizeWrong: return "Src size is incorrect";
    case kErrDstBufferNull: return "Operation on NULL destination buffer";
    case 100: return "Frame index is too large";
    case 102: return "An I/O error occurred when reading/seeking";
    case 104: return "Destination buffer is wrong";
    case 105: return "Source buffer is wrong";
    default: return "Unspecified error code";
    }
}
/* S/ThrowHelper.cs:18-24 (EnsureZdictSuccess) -> U/Zdict.cs:11-19: the dictionary builder's error helpers are the common ones */
unsigned ZDICT_isError(size_t code) { return isErr(code); }
const char* ZDICT_getErrorName(size_t code) { return ZSTD_getErrorName(code); }
unsigned ZSTD_versionNumber(void) { return 10501; }
const char* ZSTD_versionString(void) { return "1.5.1"; }

// ---------------- streaming adapters on the batched engine ----------------
// ZSTD_compressStream2 (S/Compressor.cs:108-116 <- S/CompressionStream.cs:130-190; U/ZstdCompress.cs:6632-6861).
// Input is buffered on the host until a batch (16 MiB) is full or the caller flushes/ends; each batch goes through the
// one-shot pipeline and comes back as complete, independent 64 KiB frames, so a flush point is a frame boundary and the
// concatenation of everything emitted is one ordinary multi-frame zstd stream.  Return value as the reference's: for
// e_flush / e_end the number of bytes still to be flushed (0 = done), for e_continue a non-zero hint.
static size_t cstream_drain(ZSTD_CCtx* c, ZSTD_outBuffer* o)
{
    const size_t avail = c->sOut.size() - c->sOutPos, room = o->size - o->pos;
    const size_t n = avail < room ? avail : room;
    if (n) { memcpy((u8*)o->dst + o->pos, c->sOut.data() + c->sOutPos, n); o->pos += n; c->sOutPos += n; }
    if (c->sOutPos == c->sOut.size()) { c->sOut.clear(); c->sOutPos = 0; }
    return c->sOut.size() - c->sOutPos;
}
static size_t cstream_compress(ZSTD_CCtx* c, size_t n)      // first n buffered bytes -> appended to sOut
{
    const size_t cap = ZSTD_compressBound(n), at = c->sOut.size();
    c->sOut.resize(at + cap);
    const size_t r = ZSTD_compress2(c, c->sOut.data() + at, cap, c->sIn.data(), n);
    if (isErr(r)) { c->sOut.resize(at); return r; }
    c->sOut.resize(at + r);
    c->sIn.erase(c->sIn.begin(), c->sIn.begin() + (ptrdiff_t)n);
    return 0;
}
static size_t ZSTD_compressStream2_impl(ZSTD_CCtx* c, ZSTD_outBuffer* output, ZSTD_inBuffer* input, int endOp)
{
    if (!c || !output || !input) return ZERR(kErrGeneric);
    if (output->pos > output->size) return ZERR(104);          // dstBuffer_wrong
    if (input->pos > input->size) return ZERR(105);            // srcBuffer_wrong
    if ((unsigned)endOp > 2) return ZERR(kErrParameterOutOfBound);
    if (input->size > input->pos && !input->src) return ZERR(kErrSrcSizeWrong);
    if (output->size > output->pos && !output->dst) return ZERR(kErrDstBufferNull);
    if (cstream_drain(c, output)) return c->sOut.size() - c->sOutPos;       // output full: nothing consumed this time
    if (!c->sEnding) {
        const size_t n = input->size - input->pos;
        if (n) { c->sIn.insert(c->sIn.end(), (const u8*)input->src + input->pos, (const u8*)input->src + input->size); input->pos = input->size; c->sWrote = true; }
        size_t e = 0;
        if (endOp == 0) {                                      // ZSTD_e_continue: whole chunks only, so that frames stay 64 KiB
            if (c->sIn.size() >= c->sBatch) e = cstream_compress(c, c->sIn.size() / kChunkSize * kChunkSize);
        } else {
            if (!c->sIn.empty()) e = cstream_compress(c, c->sIn.size());
            else if (endOp == 2 && !c->sWrote) {               // ZSTD_e_end on an empty stream: the empty frame (U/ZstdCompress.cs:5598-5656)
                u8 tmp[16]; const size_t r = ZSTD_compress2(c, tmp, sizeof tmp, tmp, 0);
                if (isErr(r)) e = r; else c->sOut.insert(c->sOut.end(), tmp, tmp + r);
            }
            if (endOp == 2) c->sEnding = true;
        }
        if (isErr(e)) return e;
    }
    const size_t left = cstream_drain(c, output);
    if (c->sEnding && left == 0) { c->sEnding = false; c->sWrote = false; }    // frame session closed; the context may start another
    if (endOp == 0) return left ? left : (c->sBatch > c->sIn.size() ? c->sBatch - c->sIn.size() : 1);
    return left;
}

// ZSTD_decompressStream (S/Decompressor.cs:97-106 <- S/DecompressionStream.cs:88-162; U/ZstdDecompress.cs:2816-3205).
// Compressed bytes are collected until at least one whole frame is present (frame sizes come from the block headers,
// ZSTD_findFrameSizeInfo); all whole frames collected so far are decoded in one GPU batch into a pending buffer that is
// handed out as the caller's output space allows.  Returns 0 when a frame boundary is reached and everything is flushed,
// an error, or a non-zero hint.  As in the reference (U/ZstdDecompress.cs:3170-3194) the last input byte is held hostage
// while decoded data is still pending, so that a caller who stops feeding at end of input still gets called back.
static size_t dstream_drain(ZSTD_DCtx* d, ZSTD_outBuffer* o)
{
    const size_t avail = d->dOut.size() - d->dOutPos, room = o->size - o->pos;
    const size_t n = avail < room ? avail : room;
    if (n) { memcpy((u8*)o->dst + o->pos, d->dOut.data() + d->dOutPos, n); o->pos += n; d->dOutPos += n; }
    if (d->dOutPos == d->dOut.size()) { d->dOut.clear(); d->dOutPos = 0; }
    return d->dOut.size() - d->dOutPos;
}
static size_t ZSTD_decompressStream_impl(ZSTD_DCtx* d, ZSTD_outBuffer* output, ZSTD_inBuffer* input)
{
    if (!d || !output || !input) return ZERR(kErrGeneric);
    if (output->pos > output->size) return ZERR(104);
    if (input->pos > input->size) return ZERR(105);
    if (input->size > input->pos && !input->src) return ZERR(kErrSrcSizeWrong);
    if (output->size > output->pos && !output->dst) return ZERR(kErrDstBufferNull);
    if (d->hostage && input->pos < input->size) { input->pos++; d->hostage = false; }       // that byte was consumed earlier
    size_t pending = dstream_drain(d, output);
    if (!pending) {
        const size_t n = input->size - input->pos;
        if (n) { d->dIn.insert(d->dIn.end(), (const u8*)input->src + input->pos, (const u8*)input->src + input->size); input->pos = input->size; }
        size_t whole = 0; unsigned long long bound = 0;
        while (whole < d->dIn.size()) {
            unsigned long long b = 0;
            // ZSTD_d_windowLogMax bounds what a streamed frame may ask for (U/ZstdDecompress.cs:2966-2969, with the 1 KiB floor of
            // :2965): checked as soon as the header is there, before the frame is collected or anything is sized from it
            { u64 w = host_frame_window(d->dIn.data() + whole, d->dIn.size() - whole);
              if (w && w < 1024) w = 1024;
              if (w > (1ull << d->windowLogMax)) { d->dIn.clear(); return ZERR(kErrWindowTooLarge); } }
            const size_t fs = host_frame_size_info(d->dIn.data() + whole, d->dIn.size() - whole, &b);
            if (isErr(fs)) { if (fs == ZERR(kErrSrcSizeWrong)) break; return fs; }          // incomplete frame: wait for more input
            whole += fs; bound += b;
        }
        if (whole) {
            d->dOut.resize((size_t)bound); d->dOutPos = 0;
            const size_t r = ZSTD_decompressDCtx_impl(d, d->dOut.data(), d->dOut.size(), d->dIn.data(), whole);
            if (isErr(r)) { d->dOut.clear(); return r; }
            d->dOut.resize(r);
            d->dIn.erase(d->dIn.begin(), d->dIn.begin() + (ptrdiff_t)whole);
            pending = dstream_drain(d, output);
        }
    }
    if (pending) {
        if (!d->hostage && input->pos == input->size && input->pos > 0) { input->pos--; d->hostage = true; }
        return 1;
    }
    if (d->hostage) return 1;                                  // flushed, but the hostage byte has not been handed back yet
    return d->dIn.empty() ? 0 : 1;                             // 0 only on a frame boundary
}

// ---------------- extensions ----------------
int ZSTDMI_deviceCount(void) { return device_count(); }
size_t ZSTDMI_CCtx_setDevice(ZSTD_CCtx* c, int device) { if (!c) return ZERR(kErrGeneric); if (c->deviceOk && device != c->device) return ZERR(kErrStageWrong); c->device = device; return 0; }
size_t ZSTDMI_DCtx_setDevice(ZSTD_DCtx* d, int device) { if (!d) return ZERR(kErrGeneric); if (d->deviceOk && device != d->device) return ZERR(kErrStageWrong); d->device = device; return 0; }
// devices: one worker per entry (an ordinal may repeat: several workers share that device); n <= 1 = back to the context's own device
size_t ZSTDMI_CCtx_setDevices(ZSTD_CCtx* c, const int* devices, int n)
{
    if (!c || n < 0 || n > 64 || (n && !devices)) return ZERR(kErrParameterOutOfBound);
    for (int i = 0; i < n; ++i) if (devices[i] < 0 || devices[i] >= device_count()) return ZERR(kErrInitMissing);
    for (ZSTD_CCtx* w : c->workers) (void)ZSTD_freeCCtx(w);
    c->workers.clear();
    if (n <= 1) { if (n == 1) return ZSTDMI_CCtx_setDevice(c, devices[0]); return 0; }
    for (int i = 0; i < n; ++i) {
        ZSTD_CCtx* w = ZSTD_createCCtx();
        if (!w) return ZERR(kErrMemoryAllocation);
        w->device = devices[i]; w->dictGen = ~(u64)0;
        c->workers.push_back(w);
    }
    return 0;
}
size_t ZSTDMI_DCtx_setDevices(ZSTD_DCtx* d, const int* devices, int n)
{
    if (!d || n < 0 || n > 64 || (n && !devices)) return ZERR(kErrParameterOutOfBound);
    for (int i = 0; i < n; ++i) if (devices[i] < 0 || devices[i] >= device_count()) return ZERR(kErrInitMissing);
    for (ZSTD_DCtx* w : d->workers) (void)ZSTD_freeDCtx(w);
    d->workers.clear();
    if (n <= 1) { if (n == 1) return ZSTDMI_DCtx_setDevice(d, devices[0]); return 0; }
    for (int i = 0; i < n; ++i) {
        ZSTD_DCtx* w = ZSTD_createDCtx();
        if (!w) return ZERR(kErrMemoryAllocation);
        w->device = devices[i]; w->dictGen = ~(u64)0;
        d->workers.push_back(w);
    }
    return 0;
}
size_t ZSTDMI_CCtx_setStream(ZSTD_CCtx* c, void* st) { size_t e = cctx_bind(c); if (isErr(e)) return e; c->stream = st ? (hipStream_t)st : c->ownStream; return 0; }
size_t ZSTDMI_DCtx_setStream(ZSTD_DCtx* d, void* st) { size_t e = dctx_bind(d); if (isErr(e)) return e; d->stream = st ? (hipStream_t)st : d->ownStream; return 0; }
int ZSTDMI_debugLastWalkSerial(const ZSTD_DCtx* d) { return d ? (int)d->lastWalkSerial : -1; }
size_t ZSTDMI_DCtx_setExecWaves(ZSTD_DCtx* d, unsigned waves) { if (!d || (waves != 0 && waves != 1 && waves != 2 && waves != 4 && waves != 8 && waves != 16)) return ZERR(kErrParameterOutOfBound); d->execWaves = (int)waves; return 0; }
size_t ZSTDMI_DCtx_setOverlap(ZSTD_DCtx* d, unsigned mode) { if (!d || mode > 2) return ZERR(kErrParameterOutOfBound); d->overlapMode = (int)mode; return 0; }
size_t ZSTDMI_DCtx_setLongFrames(ZSTD_DCtx* d, unsigned mode) { if (!d || mode > 2) return ZERR(kErrParameterOutOfBound); d->originMode = (int)mode; return 0; }
size_t ZSTDMI_DCtx_setLiteralDecoder(ZSTD_DCtx* d, unsigned mode) { if (!d || mode > 3) return ZERR(kErrParameterOutOfBound); d->litDecoder = mode; return 0; }
size_t ZSTDMI_CCtx_setPassChunks(ZSTD_CCtx* c, unsigned chunks) { if (!c || chunks == 0 || chunks > (1u << 20)) return ZERR(kErrParameterOutOfBound); c->passChunks = chunks; return 0; }
// bytes: 0 = independent 64 KiB frames; > 0 = cross-chunk history of that many bytes per block (rounded to 4 KiB, at most 48 KiB);
// < 0 = by level.  frameBytes: content of one multi-block frame (64 KiB .. 16 MiB), 0 = keep.
size_t ZSTDMI_CCtx_setHistory(ZSTD_CCtx* c, int bytes, unsigned frameBytes)
{
    if (!c || bytes > (48 << 10) || (frameBytes && (frameBytes < kChunkSize || frameBytes > (16u << 20)))) return ZERR(kErrParameterOutOfBound);
    c->historyBytes = bytes; if (frameBytes) c->frameBytes = frameBytes; return 0;
}
size_t ZSTDMI_CCtx_setParser(ZSTD_CCtx* c, unsigned mode) { if (!c || mode > 1) return ZERR(kErrParameterOutOfBound); c->parser = mode; return 0; }
size_t ZSTDMI_CCtx_setProfiling(ZSTD_CCtx* c, int en) { if (!c) return ZERR(kErrGeneric); c->timer.enabled = en != 0; return 0; }
size_t ZSTDMI_DCtx_setProfiling(ZSTD_DCtx* d, int en) { if (!d) return ZERR(kErrGeneric); d->timer.enabled = en != 0; return 0; }
int ZSTDMI_CCtx_getStageTimes(const ZSTD_CCtx* c, float* ms, const char** names, int cap)
{
    if (!c) return 0;
    int n = c->nStages < cap ? c->nStages : cap;
    for (int i = 0; i < n; i++) { if (ms) ms[i] = c->stageMs[i]; if (names) names[i] = c->stageNames[i]; }
    return n;
}
int ZSTDMI_DCtx_getStageTimes(const ZSTD_DCtx* d, float* ms, const char** names, int cap)
{
    if (!d) return 0;
    int n = d->timer.n < cap ? d->timer.n : cap;
    for (int i = 0; i < n; i++) { if (ms) ms[i] = d->timer.ms[i]; if (names) names[i] = d->timer.names[i]; }
    return n;
}

size_t ZSTDMI_debugGetChunk(ZSTD_CCtx* c, size_t chunkIdx, ZSTDMI_Seq* seqs, size_t seqCap, size_t* nbSeq, void* lits, size_t litCap, size_t* litSize)
{
    size_t e = cctx_bind(c); if (isErr(e)) return e;
    if (chunkIdx >= c->lastChunks) return ZERR(kErrParameterOutOfBound);
    ChunkMeta m;
    if (hipMemcpy(&m, (ChunkMeta*)c->meta.p + chunkIdx, sizeof m, hipMemcpyDeviceToHost) != hipSuccess) return ZERR(kErrGeneric);
    *nbSeq = m.nbSeq; *litSize = m.litSize;
    const size_t ns = m.nbSeq < seqCap ? m.nbSeq : seqCap, nl = m.litSize < litCap ? m.litSize : litCap;
    if (ns && hipMemcpy(seqs, (Seq*)c->seqs.p + chunkIdx * kMaxSeq, ns * sizeof(Seq), hipMemcpyDeviceToHost) != hipSuccess) return ZERR(kErrGeneric);
    const u8* litDev = m.litFromSrc ? c->lastSrc + chunkIdx * (size_t)c->lastChunkBytes : (const u8*)c->lits.p + chunkIdx * kLitStride;
    if (nl && (!litDev || hipMemcpy(lits, litDev, nl, hipMemcpyDeviceToHost) != hipSuccess)) return ZERR(kErrGeneric);
    return 0;
}

size_t ZSTDMI_debugEntropyBlock(ZSTD_CCtx* c, void* dst, size_t dstCapacity, const ZSTDMI_Seq* seqs, size_t nbSeq,
                                const void* lits, size_t litSize, size_t srcSize)
{
    size_t e = cctx_bind(c); if (isErr(e)) return e;
    if (nbSeq > kMaxSeq || litSize > kChunkSize || srcSize > kChunkSize) return ZERR(kErrParameterOutOfBound);
    if (!cctx_workspace(c, 1)) return ZERR(kErrMemoryAllocation);
    hipStream_t s = c->stream;
    ChunkMeta m = {}; m.srcSize = (u32)srcSize; m.nbSeq = (u32)nbSeq; m.litSize = (u32)litSize;
    m.fhSize = 4 + 1 + (srcSize < 256 ? 1 : 2);
    if (nbSeq) (void)hipMemcpyAsync(c->seqs.p, seqs, nbSeq * sizeof(Seq), hipMemcpyHostToDevice, s);
    if (litSize) (void)hipMemcpyAsync(c->lits.p, lits, litSize, hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(c->meta.p, &m, sizeof m, hipMemcpyHostToDevice, s);
    launch_huf_build((u8*)c->lits.p, (ChunkMeta*)c->meta.p, (HufTable*)c->tables.p, (u8*)c->slots.p, 1, 0, nullptr, 0, s, StageHook());
    launch_huf_encode((u8*)c->lits.p, (ChunkMeta*)c->meta.p, (HufTable*)c->tables.p, (u8*)c->slots.p, nullptr, nullptr, 0, 1, nullptr, 0, s);
    { const u32 plainReps[3] = { 1, 4, 8 };
      const Resolved rs = resolve_call(sticky_params(c), srcSize, kChunkSize);
      launch_seq_encode((Seq*)c->seqs.p, (ChunkMeta*)c->meta.p, (u8*)c->slots.p, 1, rs.cp.strategy < kStratGreedy ? rs.cp.strategy : (u32)kStratGreedy, 0, 0, 0, 0, plainReps, 0, kChunkSize, 0, s); }
    if (hipMemcpyAsync(&m, c->meta.p, sizeof m, hipMemcpyDeviceToHost, s) != hipSuccess) return ZERR(kErrGeneric);
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
    if (m.blockType != 2) return 0;
    if (m.bodySize > dstCapacity) return ZERR(kErrDstSizeTooSmall);
    if (hipMemcpy(dst, (u8*)c->slots.p + m.fhSize + 3, m.bodySize, hipMemcpyDeviceToHost) != hipSuccess) return ZERR(kErrGeneric);
    c->lastChunks = 1;
    return m.bodySize;
}

size_t ZSTDMI_debugPoisonedChunk(ZSTD_CCtx* c, unsigned nbSeq, unsigned litSize, unsigned srcSize, unsigned fill)
{
    size_t e = cctx_bind(c); if (isErr(e)) return e;
    if (!cctx_workspace(c, 1)) return ZERR(kErrMemoryAllocation);
    hipStream_t s = c->stream;
    ChunkMeta m = {}; m.srcSize = srcSize; m.nbSeq = nbSeq; m.litSize = litSize; m.fhSize = 7;
    (void)hipMemsetAsync(c->seqs.p, (int)(fill & 0xFF), (size_t)kMaxSeq * sizeof(Seq), s);
    (void)hipMemsetAsync(c->lits.p, (int)(fill & 0xFF), kLitStride, s);
    (void)hipMemcpyAsync(c->meta.p, &m, sizeof m, hipMemcpyHostToDevice, s);
    launch_huf_build((u8*)c->lits.p, (ChunkMeta*)c->meta.p, (HufTable*)c->tables.p, (u8*)c->slots.p, 1, 0, (const u8*)c->lits.p, kChunkSize, s, StageHook());
    launch_huf_encode((u8*)c->lits.p, (ChunkMeta*)c->meta.p, (HufTable*)c->tables.p, (u8*)c->slots.p, nullptr, nullptr, 0, 1, (const u8*)c->lits.p, kChunkSize, s);
    { const u32 plainReps[3] = { 1, 4, 8 };
      launch_seq_encode((Seq*)c->seqs.p, (ChunkMeta*)c->meta.p, (u8*)c->slots.p, 1, 1, 0, 1, 0, 0, plainReps, 0, kChunkSize, srcSize < kChunkSize ? srcSize : kChunkSize, s); }
    if (hipMemcpyAsync(&m, c->meta.p, sizeof m, hipMemcpyDeviceToHost, s) != hipSuccess) return ZERR(kErrGeneric);
    if (hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return ZERR(kErrGeneric); }
    c->lastChunks = 0;
    return m.outSize;
}

// ---------------- entry points whose host-side containers may throw: guarded (see guarded()) ----------------
size_t ZSTD_CCtx_loadDictionary(ZSTD_CCtx* c, const void* dict, size_t dictSize) { return guarded([&] { return ZSTD_CCtx_loadDictionary_impl(c, dict, dictSize); }); }
size_t ZSTD_DCtx_loadDictionary(ZSTD_DCtx* d, const void* dict, size_t dictSize) { return guarded([&] { return ZSTD_DCtx_loadDictionary_impl(d, dict, dictSize); }); }
size_t ZSTD_findFrameCompressedSize(const void* src, size_t srcSize) { return guarded([&] { return ZSTD_findFrameCompressedSize_impl(src, srcSize); }); }
size_t ZSTD_decompressDCtx(ZSTD_DCtx* d, void* dst, size_t dstCapacity, const void* src, size_t srcSize) { return guarded([&] { return ZSTD_decompressDCtx_impl(d, dst, dstCapacity, src, srcSize); }); }
size_t ZSTDMI_decompressDevice(ZSTD_DCtx* d, void* d_dst, size_t dstCapacity, const void* d_src, size_t srcSize) { return guarded([&] { return ZSTDMI_decompressDevice_impl(d, d_dst, dstCapacity, d_src, srcSize); }); }
size_t ZSTD_compressStream2(ZSTD_CCtx* c, ZSTD_outBuffer* output, ZSTD_inBuffer* input, int endOp) { return guarded([&] { return ZSTD_compressStream2_impl(c, output, input, endOp); }); }
size_t ZSTD_decompressStream(ZSTD_DCtx* d, ZSTD_outBuffer* output, ZSTD_inBuffer* input) { return guarded([&] { return ZSTD_decompressStream_impl(d, output, input); }); }
unsigned long long ZSTD_decompressBound(const void* src, size_t srcSize)
{
    try { return ZSTD_decompressBound_impl(src, srcSize); } catch (...) { return (unsigned long long)0 - 2; }      /* ZSTD_CONTENTSIZE_ERROR */
}
unsigned long long ZSTD_getFrameContentSize(const void* src, size_t srcSize)
{
    try { return ZSTD_getFrameContentSize_impl(src, srcSize); } catch (...) { return (unsigned long long)0 - 2; }      /* ZSTD_CONTENTSIZE_ERROR */
}

} // extern "C"
