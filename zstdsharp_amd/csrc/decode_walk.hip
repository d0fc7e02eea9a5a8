// decode_walk.hip — the decoder's work lists on gfx950 (SURVEY.md §8 a-13, a-14, a-16's headers).
//
//   frame walk      : ZSTD_findFrameSizeInfo over the whole input (U/ZstdDecompress.cs:877-951, 971-993), headers only.
//                     Parallel form: the input is cut into 128 KiB segments; one wave per segment scans for the first frame
//                     magic, validates it by chaining frame -> frame until it leaves the segment, and a scan kernel checks
//                     that every segment's exit is the next segment's entry (anything else — embedded frames inside raw
//                     blocks, a corrupt header, a frame naming a dictionary — falls back to the exact serial walk, which
//                     also produces the reference's error).  Both forms count first (the host sizes the lists), then emit
//                     one FrameDesc per frame and one BlockDesc per block.
//   block_parse     : one lane per compressed block: the literals section header (ZSTD_decodeLiteralsBlock's header parse,
//                     U/ZstdDecompressBlock.cs:88-396) and the sequences section header (ZSTD_decodeSeqHeaders, :1845-1943) up
//                     to where the bitstream starts; the NCount descriptions are measured, not stored.
//   block_link      : one lane per frame: which earlier block a treeless literals section takes its Huffman table from
//                     (:197-207) and which one defines each FSE table used in repeat mode (:1780-1786) — the only state
//                     besides repcodes and history that the reference carries from block to block; literal offsets.
//   seq_scan        : exclusive scan of the blocks' sequence counts -> where each block's records go.
//   block_offsets   : one wave per frame, after seq_decode: output offset and starting repcodes of every block (prefix sums;
//                     repcodes by composing the blocks' transfer functions), regenerated size against the header's
//                     (U/ZstdDecompress.cs:1177-1184).
//   frame_rescan    : only when some frame carries no content size: output offsets of the frames from their regenerated sizes.
#include "zmi_decode.h"

namespace zmi {

// ------------------------------------------------------------------------------------------------
// one frame of the chain
// ------------------------------------------------------------------------------------------------
// frame (or skippable frame) at pos -> next position.  Returns 0 frame, 1 skippable, or an error code >= 2 (the reference's).
// `strict`: the parallel walk's view — a frame that names a dictionary is left to the serial walk (which knows the loaded one).
struct ChainOut { u64 next; u64 content; u64 window; u32 nbBlocks; u32 hdrSize; u32 checksum; u32 unsized; u32 dictID; };
// EMIT: the frame was validated by an earlier walk; this one also writes its blocks' descriptors (block header walks are chains of
// dependent loads — 0.3 us a block — so a walk that emits must not be a walk of its own behind the one that validates)
template <bool EMIT>
__device__ inline u32 chain_step_t(const u8* __restrict__ src, u64 srcSize, u64 pos, ChainOut& o, u32 frameIdx, u32 firstBlock, BlockDesc* __restrict__ blocks)
{
    o.content = 0; o.nbBlocks = 0; o.hdrSize = 0; o.checksum = 0; o.unsized = 0; o.dictID = 0; o.window = 0; o.next = pos;
    if (srcSize - pos < 5) return kErrSrcSizeWrong;
    const u8* p = src + pos; const u64 avail = srcSize - pos;
    const u32 magic = readLE32(p);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
        if (avail < 8) return kErrSrcSizeWrong;
        const u64 sz = (u64)readLE32(p + 4) + 8;
        if (sz > avail) return kErrSrcSizeWrong;
        o.next = pos + sz; return 1;
    }
    const FrameHeader h = parse_frame_header(p, avail);
    if (h.err) return h.err;
    u64 q = pos + h.headerSize; u32 nb = 0;
    for (;;) {
        if (srcSize - q < 3) return kErrSrcSizeWrong;
        const u32 bh = readLE24(src + q);
        const u32 last = bh & 1, type = (bh >> 1) & 3; u32 cSize = bh >> 3;
        if (type == 3) return kErrCorruption;
        if (type == 1) cSize = 1;
        if (3 + (u64)cSize > srcSize - q) return kErrSrcSizeWrong;
        if (EMIT) {
            BlockDesc b = {};
            b.srcOff = q + 3; b.frame = frameIdx; b.type = (u8)type; b.last = (u8)last;
            b.bsz = cSize; b.outSize = type == 2 ? 0u : (bh >> 3);
            b.hufSrc = kNoBlock; b.tblSrc[0] = b.tblSrc[1] = b.tblSrc[2] = kNoBlock;
            blocks[firstBlock + nb] = b;
        }
        q += 3 + cSize;
        if (++nb == 0xFFFFFFFFu) return kErrMemoryAllocation;
        if (last) break;
    }
    if (h.checksum) { if (srcSize - q < 4) return kErrSrcSizeWrong; q += 4; }
    o.next = q; o.nbBlocks = nb; o.hdrSize = h.headerSize; o.checksum = h.checksum; o.dictID = h.dictID; o.window = h.windowSize;
    o.unsized = h.contentSize == ~0ull;
    // a frame without a content size gets the bound ZSTD_findFrameSizeInfo gives it: nbBlocks x min(window, 128 KiB)
    o.content = o.unsized ? (u64)nb * (h.windowSize < (1u << 17) ? h.windowSize : (u64)(1u << 17)) : h.contentSize;
    return 0;
}

__device__ inline u32 chain_step(const u8* __restrict__ src, u64 srcSize, u64 pos, ChainOut& o) { return chain_step_t<false>(src, srcSize, pos, o, 0, 0, nullptr); }

// the frame at `pos` (validated by an earlier walk) into the lists: ONE walk over its block headers.  -> chain_step's result
__device__ inline u32 emit_frame(const u8* __restrict__ src, u64 srcSize, u64 pos, ChainOut& o, u32 frameIdx, u32 firstBlock, u64 dstOff,
                                 FrameDesc* __restrict__ frames, BlockDesc* __restrict__ blocks)
{
    const u32 st = chain_step_t<true>(src, srcSize, pos, o, frameIdx, firstBlock, blocks);
    if (st) return st;
    FrameDesc f; f.srcOff = pos; f.dstOff = dstOff; f.scratchOff = dstOff; f.srcSize = o.next - pos; f.dstSize = o.content;
    f.firstBlock = firstBlock; f.nbBlocks = o.nbBlocks; f.unsized = o.unsized; f.checksum = o.checksum; f.bad = 0; f.hasSeq = 0; f.viaOrigin = 0; f.pad = 0; f.originOff = 0;
    frames[frameIdx] = f;
    return 0;
}

// the exact serial walk (ZSTD_decompressMultiFrame's loop, U/ZstdDecompress.cs:1216-1315), one lane.  emit = 0: count frames and
// blocks, sum the content sizes, find the first error; emit = 1 (same input, lists allocated): write the lists.
__global__ void frame_walk_serial_kernel(const u8* __restrict__ src, u64 srcSize, FrameDesc* __restrict__ frames, BlockDesc* __restrict__ blocks,
                                         u32 maxFrames, u32* __restrict__ status, u32 dictID, u32 emit)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u64 pos = 0, dstOff = 0; u32 n = 0, err = 0, nUnsized = 0; u64 nBlocks = 0;
    while (srcSize - pos >= 5) {            // ZSTD_decompressMultiFrame loop condition (U/ZstdDecompress.cs:1228)
        ChainOut o;
        // (emit: the same input has been through the counting pass, which stopped at the first error and sized the lists)
        const u32 st = emit ? emit_frame(src, srcSize, pos, o, n, (u32)nBlocks, dstOff, frames, blocks) : chain_step(src, srcSize, pos, o);
        if (st == 1) { pos = o.next; continue; }
        if (st) { err = (st == kErrPrefixUnknown && n > 0) ? (u32)kErrSrcSizeWrong : st; break; }
        if (o.dictID && o.dictID != dictID) { err = kErrDictionaryWrong; break; }       // U/ZstdDecompress.cs:1404-1412 (dictID 0 = none loaded)
        if (n >= maxFrames || nBlocks + o.nbBlocks > 0xFFFFFFF0ull) { err = kErrMemoryAllocation; break; }
        n++; nUnsized += o.unsized; nBlocks += o.nbBlocks;
        dstOff += o.content; pos = o.next;
    }
    if (!err && pos != srcSize) err = kErrSrcSizeWrong;     // trailing garbage (U/ZstdDecompress.cs:1309-1312)
    if (emit) return;
    status[kStFrames] = n; status[kStErr] = err; status[kStTotalLo] = (u32)dstOff; status[kStTotalHi] = (u32)(dstOff >> 32);
    status[kStUnsized] = nUnsized; status[kStBlocks] = (u32)nBlocks;
}

// ------------------------------------------------------------------------------------------------
// parallel frame walk
// ------------------------------------------------------------------------------------------------
constexpr u32 kSegLog = 17;
struct SegInfo { u64 entry, exit, dstBytes; u32 count, valid, blocks, pad; };

__global__ __launch_bounds__(64) void walk_segments_kernel(const u8* __restrict__ src, u64 srcSize, SegInfo* __restrict__ segs, u32 nSeg)
{
    const u32 s = blockIdx.x, lane = threadIdx.x;
    if (s >= nSeg) return;
    const u64 segStart = (u64)s << kSegLog;
    const u64 segEnd = (segStart + (1ull << kSegLog)) < srcSize ? segStart + (1ull << kSegLog) : srcSize;
    SegInfo r; r.entry = 0; r.exit = 0; r.dstBytes = 0; r.count = 0; r.valid = 0; r.blocks = 0; r.pad = 0;
    u64 scan = segStart;
    while (scan < segEnd) {
        // 64 lanes x 4 byte positions: a position p is a candidate if the dword at p is a frame or skippable-frame magic
        const u64 base = scan + 4 * lane;
        u64 w = 0;
        if (base + 8 <= srcSize) w = readLE64(src + base);
        else for (u32 k = 0; k < 8; k++) if (base + k < srcSize) w |= (u64)src[base + k] << (8 * k);
        u32 hit = 4;
#pragma unroll
        for (int k = 3; k >= 0; k--) {
            const u32 v = (u32)(w >> (8 * k));
            if ((v == 0xFD2FB528u || (v & 0xFFFFFFF0u) == 0x184D2A50u) && base + k < segEnd) hit = k;
        }
        const u64 m = ballot(hit < 4);
        if (!m) { scan += 256; continue; }
        const u32 fl = ctz64(m);
        const u64 cand = scan + 4 * fl + read_lane(hit, fl);
        // validate by chaining until the chain leaves the segment (every lane walks the same chain: uniform)
        u64 pos = cand, dstBytes = 0, nBlocks = 0; u32 count = 0; bool ok = true;
        while (pos < segEnd) {
            ChainOut o;
            const u32 st = chain_step(src, srcSize, pos, o);
            // frames naming a dictionary and frames without a content size take the serial walk (it knows the loaded dictionary,
            // and the regenerated sizes of unsized frames decide where everything behind them goes)
            if (st >= 2 || (st == 0 && (o.dictID || o.unsized))) { ok = false; break; }
            if (st == 0) { count++; dstBytes += o.content; nBlocks += o.nbBlocks; }
            pos = o.next;
        }
        // A magic that turns up by chance inside a frame's payload (the sixteen skippable-frame magics: once per 270 MB of
        // incompressible payload) can chain out of the segment too — a skippable frame of any size that stays inside the input is
        // "valid" — and, where it lies in front of the segment's first real frame, the link check then sends the whole call to the
        // serial walk (40 ms for 16 384 frames).  A real chain leaves the segment AT a frame (or at the end of the input): a
        // candidate whose chain lands anywhere else is not one.
        if (ok && pos != srcSize) {
            u32 v = 0;
            if (srcSize - pos >= 4) v = readLE32(src + pos);
            if (!(v == 0xFD2FB528u || (v & 0xFFFFFFF0u) == 0x184D2A50u)) ok = false;
        }
        if (ok && nBlocks < 0xFFFFFFF0ull) { r.entry = cand; r.exit = pos; r.dstBytes = dstBytes; r.count = count; r.blocks = (u32)nBlocks; r.valid = 1; break; }
        scan = cand + 1;           // false positive (or a corrupt stream: the link check then sends us to the serial walk)
    }
    if (lane == 0) segs[s] = r;
}

// single workgroup: link check + prefix sums.  status: frames, total, blocks; [kStUsable] = 1 when the parallel walk is usable
__global__ __launch_bounds__(1024) void walk_link_kernel(const SegInfo* __restrict__ segs, u32 nSeg, u64 srcSize, u32 maxFrames,
                                                         u32* __restrict__ frameBase, u32* __restrict__ blockBase, u64* __restrict__ dstBase,
                                                         u32* __restrict__ status)
{
    __shared__ u64 sh64[16], shBlk[16]; __shared__ u32 sh32[16]; __shared__ s32 shLast[16]; __shared__ u32 bad;
    const u32 tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    if (tid == 0) bad = 0;
    __syncthreads();
    u64 carryDst = 0, carryBlk = 0; u32 carryCnt = 0; s32 carryLast = -1;
    for (u32 base = 0; base < nSeg; base += 1024) {
        const u32 i = base + tid;
        SegInfo g; g.valid = 0; g.count = 0; g.dstBytes = 0; g.entry = 0; g.exit = 0; g.blocks = 0;
        if (i < nSeg) g = segs[i];
        // inclusive scans inside the wave: counts, bytes, blocks, index of the last valid segment
        u32 c = g.valid ? g.count : 0; u64 b = g.valid ? g.dstBytes : 0, k = g.valid ? g.blocks : 0; s32 lastv = g.valid ? (s32)i : -1;
        u32 ci = c; u64 bi = b, ki = k; s32 li = lastv;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 tc = __shfl_up(ci, d); const u64 tb = __shfl_up(bi, d), tk = __shfl_up(ki, d); const s32 tl = __shfl_up(li, d);
            if ((int)lane >= d) { ci += tc; bi += tb; ki += tk; li = tl > li ? tl : li; }
        }
        if (lane == 63) { sh32[wave] = ci; sh64[wave] = bi; shBlk[wave] = ki; shLast[wave] = li; }
        __syncthreads();
        u32 cb = carryCnt; u64 bb = carryDst, kb = carryBlk; s32 lb = carryLast; u32 call = 0; u64 ball = 0, kall = 0; s32 lall = -1;
        for (u32 w = 0; w < 16; w++) {
            if (w < wave) { cb += sh32[w]; bb += sh64[w]; kb += shBlk[w]; lb = shLast[w] > lb ? shLast[w] : lb; }
            call += sh32[w]; ball += sh64[w]; kall += shBlk[w]; lall = shLast[w] > lall ? shLast[w] : lall;
        }
        // previous valid segment (exclusive): from the lanes before me in my wave, else from earlier waves / rounds
        s32 prevInWave = __shfl_up(li, 1); if (lane == 0) prevInWave = -1;
        const s32 prevValid = prevInWave > lb ? prevInWave : lb;
        if (i < nSeg && g.valid) {
            const u64 expect = prevValid >= 0 ? segs[prevValid].exit : 0;
            if (g.entry != expect) atomicOr(&bad, 1u);
            frameBase[i] = cb + ci - c; dstBase[i] = bb + bi - b; blockBase[i] = (u32)(kb + ki - k);
        }
        carryCnt += call; carryDst += ball; carryBlk += kall; carryLast = lall > carryLast ? lall : carryLast;
        __syncthreads();
    }
    if (tid == 0) {
        u32 usable = !bad;
        if (carryLast < 0) usable = 0; else if (segs[carryLast].exit != srcSize) usable = 0;
        if (carryCnt > maxFrames || carryBlk > 0xFFFFFFF0ull) usable = 0;
        status[kStFrames] = carryCnt; status[kStErr] = 0; status[kStTotalLo] = (u32)carryDst; status[kStTotalHi] = (u32)(carryDst >> 32);
        status[kStUsable] = usable; status[kStUnsized] = 0; status[kStBlocks] = (u32)carryBlk;
    }
}

__global__ __launch_bounds__(256) void walk_emit_kernel(const u8* __restrict__ src, u64 srcSize, const SegInfo* __restrict__ segs, u32 nSeg,
                                                        const u32* __restrict__ frameBase, const u32* __restrict__ blockBase, const u64* __restrict__ dstBase,
                                                        FrameDesc* __restrict__ frames, BlockDesc* __restrict__ blocks)
{
    const u32 s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nSeg) return;
    const SegInfo g = segs[s];
    if (!g.valid) return;
    u64 pos = g.entry, dstOff = dstBase[s]; u32 idx = frameBase[s], blk = blockBase[s];
    while (pos < g.exit) {
        ChainOut o;
        const u32 st = emit_frame(src, srcSize, pos, o, idx, blk, dstOff, frames, blocks);
        if (st >= 2) return;                                   // cannot happen: the chain was validated by walk_segments
        if (st == 0) { idx++; blk += o.nbBlocks; dstOff += o.content; }
        pos = o.next;
    }
}

size_t decode_walk_workspace_bytes(u64 srcSize)
{
    const u64 nSeg = (srcSize + (1ull << kSegLog) - 1) >> kSegLog;
    return (size_t)(nSeg * (sizeof(SegInfo) + 2 * sizeof(u32) + sizeof(u64)) + 256);
}

struct WalkWs { SegInfo* segs; u64* dstBase; u32* frameBase; u32* blockBase; u32 nSeg; };
static WalkWs walk_ws(u8* walkWs, u64 srcSize)
{
    WalkWs w; w.nSeg = (u32)((srcSize + (1ull << kSegLog) - 1) >> kSegLog);
    w.segs = reinterpret_cast<SegInfo*>(walkWs);
    w.dstBase = reinterpret_cast<u64*>(walkWs + (size_t)w.nSeg * sizeof(SegInfo));
    w.frameBase = reinterpret_cast<u32*>(walkWs + (size_t)w.nSeg * (sizeof(SegInfo) + sizeof(u64)));
    w.blockBase = w.frameBase + w.nSeg;
    return w;
}
// count: frames, blocks, content bytes -> status (the host then sizes the lists)
void launch_frame_walk_count(const u8* src, u64 srcSize, u32 maxFrames, u32* status, u8* walkWs, hipStream_t stream)
{
    const WalkWs w = walk_ws(walkWs, srcSize);
    hipLaunchKernelGGL(walk_segments_kernel, dim3(w.nSeg), dim3(64), 0, stream, src, srcSize, w.segs, w.nSeg);
    hipLaunchKernelGGL(walk_link_kernel, dim3(1), dim3(1024), 0, stream, w.segs, w.nSeg, srcSize, maxFrames, w.frameBase, w.blockBase, w.dstBase, status);
}
void launch_frame_walk_emit(const u8* src, u64 srcSize, FrameDesc* frames, BlockDesc* blocks, u8* walkWs, hipStream_t stream)
{
    const WalkWs w = walk_ws(walkWs, srcSize);
    hipLaunchKernelGGL(walk_emit_kernel, dim3((w.nSeg + 255) / 256), dim3(256), 0, stream, src, srcSize, w.segs, w.nSeg, w.frameBase, w.blockBase, w.dstBase, frames, blocks);
}
void launch_frame_walk_serial(const u8* src, u64 srcSize, FrameDesc* frames, BlockDesc* blocks, u32 maxFrames, u32* status, u32 dictID, u32 emit,
                              hipStream_t stream)
{
    hipLaunchKernelGGL(frame_walk_serial_kernel, dim3(1), dim3(64), 0, stream, src, srcSize, frames, blocks, maxFrames, status, dictID, emit);
}

// ------------------------------------------------------------------------------------------------
// block_parse: the two section headers of every compressed block, one lane per block
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void block_parse_kernel(const u8* __restrict__ src, BlockDesc* __restrict__ blocks, u32 nBlocks, u32* __restrict__ status)
{
    const u32 bi = blockIdx.x * 256 + threadIdx.x;
    if (bi >= nBlocks) return;
    BlockDesc& B = blocks[bi];
    if (B.type != 2) return;
    const u32 bsz = B.bsz;
    u32 err = 0;
    do {
        if (bsz >= kBlockMax) { err = kErrSrcSizeWrong; break; }          // ZSTD_decompressBlock_internal, U/ZstdDecompressBlock.cs:3095-3098
        if (bsz < 3) { err = kErrCorruption; break; }                     // MIN_CBLOCK_SIZE, :90-93
        const u8* const b = src + B.srcOff;
        const LitHeader lh = parse_lit_header(b, bsz);
        if (lh.err) { err = lh.err; break; }
        B.litType = (u8)lh.type; B.litSingle = (u8)lh.single; B.litSize = lh.litSize; B.litCSize = lh.litCSize; B.lhSize = lh.lhSize;
        u32 bp = lh.type >= 2 ? lh.lhSize + lh.litCSize : lh.type == 0 ? lh.lhSize + lh.litSize : lh.lhSize + 1;
        // ---- sequences header (ZSTD_decodeSeqHeaders, :1845-1943) ----
        if (bp >= bsz) { err = kErrSrcSizeWrong; break; }
        u32 nbSeq = b[bp++];
        if (!nbSeq) { if (bp != bsz) { err = kErrSrcSizeWrong; break; } B.nbSeq = 0; B.outSize = lh.litSize; break; }
        if (nbSeq > 0x7F) {
            if (nbSeq == 0xFF) { if (bp + 2 > bsz) { err = kErrSrcSizeWrong; break; } nbSeq = readLE16(b + bp) + 0x7F00; bp += 2; }
            else { if (bp >= bsz) { err = kErrSrcSizeWrong; break; } nbSeq = ((nbSeq - 0x80) << 8) + b[bp++]; }
        }
        if (bp + 1 > bsz) { err = kErrSrcSizeWrong; break; }
        const u32 modes = b[bp++];
        B.nbSeq = nbSeq; B.modes = modes;
        const u32 maxSym[3] = { 35, 31, 52 };
#pragma unroll
        for (u32 t = 0; t < 3; ++t) {                                      // LL, OF, ML in that order
            const u32 mode = (modes >> (6 - 2 * t)) & 3;
            B.tblOff[t] = bp;
            if (mode == 1) { if (bp >= bsz) { err = kErrCorruption; break; } bp += 1; }
            else if (mode == 2) {
                u32 maxSV = maxSym[t], tableLog = 0;
                const u32 hs = read_ncount_t<false>(nullptr, &maxSV, &tableLog, b + bp, bsz - bp);
                if (!hs) { err = kErrCorruption; break; }
                bp += hs;
            }
        }
        if (err) break;
        B.bitsOff = bp;
    } while (false);
    if (err) { B.err = err; report_error(status, bi, kStageParse, err); }
}

// ------------------------------------------------------------------------------------------------
// block_link: table provenance inside a frame, literal offsets; one wave per frame, 64 blocks at a time
// ------------------------------------------------------------------------------------------------
// frames of up to kLinkSmall blocks (a stream of single-block 64 KiB frames has 16 384 of them per GiB): one LANE per frame, the
// blocks one after the other — a wave per frame costs three times as much there; longer frames: block_link_kernel below
constexpr u32 kLinkSmall = 4;
__global__ __launch_bounds__(64) void block_link_small_kernel(FrameDesc* __restrict__ frames, BlockDesc* __restrict__ blocks, u32 nFrames, u32 haveDict,
                                                              u32 earlyLiterals, u32* __restrict__ status)
{
    const u32 f = blockIdx.x * 64 + threadIdx.x;
    if (f >= nFrames) return;
    FrameDesc& F = frames[f];
    if (F.nbBlocks > kLinkSmall) return;                       // (a wave of block_link_kernel takes it)
    // a formatted dictionary: every frame starts from its Huffman table and its three FSE tables (ZSTD_decompressBegin_usingDict,
    // U/ZstdDecompress.cs:1956-1990); without one nothing is defined before the frame's first block defines it
    u32 lastHuf = haveDict ? kDictBlock : kNoBlock;
    u32 lastTbl[3] = { lastHuf, lastHuf, lastHuf };
    u64 litAcc = 0; u32 hasSeq = 0;
    for (u32 k = 0; k < F.nbBlocks; ++k) {
        const u32 bi = F.firstBlock + k;
        BlockDesc& B = blocks[bi];
        if (B.type != 2 || B.err) continue;
        u32 err = 0;
        if (B.litType == 2) { lastHuf = bi; B.hufSrc = bi; }
        else if (B.litType == 3) { B.hufSrc = lastHuf; if (lastHuf == kNoBlock) err = kErrDictionaryCorrupted; }     // U/ZstdDecompressBlock.cs:197-207
        if (B.litType >= 2) {
            if (B.litSize > F.dstSize - litAcc) { err = err ? err : (u32)kErrCorruption; B.litRel = 0; }
            else { B.litRel = litAcc; litAcc += B.litSize; }
            B.litInPlace = (B.nbSeq == 0 && (!earlyLiterals || (k == 0 && !F.unsized))) ? 1u : 0u;
        }
        if (B.nbSeq) {
            hasSeq = 1;
#pragma unroll
            for (u32 t = 0; t < 3; ++t) {
                const u32 mode = (B.modes >> (6 - 2 * t)) & 3;
                if (mode == 3) { B.tblSrc[t] = lastTbl[t]; if (lastTbl[t] == kNoBlock) err = err ? err : (u32)kErrCorruption; }   // :1780-1786
                else { B.tblSrc[t] = bi; lastTbl[t] = bi; }
            }
        }
        if (err) { B.err = err; report_error(status, bi, B.litType >= 2 && (err == kErrDictionaryCorrupted || B.litSize > F.dstSize) ? kStageLiterals : kStageSequences, err); }
    }
    F.hasSeq = hasSeq;
    if (litAcc) atomicAdd(reinterpret_cast<unsigned long long*>(status + kStLitLo), (unsigned long long)litAcc);
    if (hasSeq && F.dstSize >= (1u << 20) && F.dstSize < (1ull << 30))
        atomicAdd(reinterpret_cast<unsigned long long*>(status + kStBigBins) + highbit32((u32)(F.dstSize >> 20)), (unsigned long long)F.dstSize);
}

// "the latest earlier block that ..." over the 64 blocks of a batch: the highest lane below mine in `mask`, else what earlier batches left
__device__ __forceinline__ u32 latest_before(u64 mask, u32 firstOfBatch, u32 carried, u32 lane)
{
    const u64 prior = mask & lanemask_lt();
    return prior ? firstOfBatch + (63u - (u32)__builtin_clzll(prior)) : carried;
}
__global__ __launch_bounds__(64) void block_link_kernel(FrameDesc* __restrict__ frames, BlockDesc* __restrict__ blocks, u32 nFrames, u32 haveDict,
                                                        u32 earlyLiterals, u32* __restrict__ status)
{
    const u32 f = blockIdx.x, lane = threadIdx.x;
    if (f >= nFrames) return;
    FrameDesc& F = frames[f];
    const u32 first = uniform(F.firstBlock), nb = uniform(F.nbBlocks);
    if (nb <= kLinkSmall) return;                              // (a lane of block_link_small_kernel takes it)
    const u64 dstSize = F.dstSize;
    // a formatted dictionary: every frame starts from its Huffman table and its three FSE tables (ZSTD_decompressBegin_usingDict,
    // U/ZstdDecompress.cs:1956-1990); without one nothing is defined before the frame's first block defines it
    u32 lastHuf = haveDict ? kDictBlock : kNoBlock;
    u32 lastTbl0 = lastHuf, lastTbl1 = lastHuf, lastTbl2 = lastHuf;
    u64 litAcc = 0; u32 hasSeq = 0;
    for (u32 k0 = 0; k0 < nb; k0 += 64) {
        const u32 bi = first + k0 + lane;
        const bool have = k0 + lane < nb;
        u32 litType = 0, litSize = 0, nbSeq = 0, modes = 0; bool live = false;
        if (have) {
            const BlockDesc& B = blocks[bi];
            live = B.type == 2 && !B.err;
            if (live) { litType = B.litType; litSize = B.litSize; nbSeq = B.nbSeq; modes = B.modes; }
        }
        u32 err = 0;
        // the Huffman table of a treeless literals section is the latest one defined in front of it (U/ZstdDecompressBlock.cs:197-207)
        const u64 defHuf = ballot(live && litType == 2);
        u32 hufSrc = kNoBlock;
        if (live && litType == 2) hufSrc = bi;
        else if (live && litType == 3) { hufSrc = latest_before(defHuf, first + k0, lastHuf, lane); if (hufSrc == kNoBlock) err = kErrDictionaryCorrupted; }
        if (defHuf) lastHuf = first + k0 + (63u - (u32)__builtin_clzll(defHuf));
        // regenerated literals of Huffman-coded sections, one after the other in the frame's scratch (dstSize long: a block that does
        // not fit takes none of it and is an error, so the bound holds for every block whatever later kernels do with the frame)
        const bool coded = live && litType >= 2;
        const u64 mine = coded ? litSize : 0u;
        u64 incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(incl, d); if ((int)lane >= d) incl += t; }
        u64 litRel = litAcc + incl - mine;
        if (coded && litRel + mine > dstSize) { err = err ? err : (u32)kErrCorruption; litRel = 0; }
        litAcc += __shfl(incl, 63);
        // repeat-mode FSE tables: the latest table of that kind defined in front of the block (:1780-1786)
        const bool seqs = live && nbSeq != 0;
        if (ballot(seqs)) hasSeq = 1;
        u32 src3[3];
#pragma unroll
        for (u32 t = 0; t < 3; ++t) {
            const u32 mode = (modes >> (6 - 2 * t)) & 3;
            const u64 def = ballot(seqs && mode != 3);
            u32& last = t == 0 ? lastTbl0 : t == 1 ? lastTbl1 : lastTbl2;
            src3[t] = bi;
            if (seqs && mode == 3) { src3[t] = latest_before(def, first + k0, last, lane); if (src3[t] == kNoBlock) err = err ? err : (u32)kErrCorruption; }
            if (def) last = first + k0 + (63u - (u32)__builtin_clzll(def));
        }
        if (live) {
            BlockDesc& B = blocks[bi];
            if (litType >= 2) {
                B.hufSrc = hufSrc; B.litRel = litRel;
                // earlyLiterals: the literal decoder runs beside seq_decode, i.e. before block_offsets: only where the output offset is
                // known by now — the first block of a frame with a content size — can it write the output itself
                B.litInPlace = (nbSeq == 0 && (!earlyLiterals || (k0 + lane == 0 && !F.unsized))) ? 1u : 0u;
            }
            if (seqs) { B.tblSrc[0] = src3[0]; B.tblSrc[1] = src3[1]; B.tblSrc[2] = src3[2]; }
            if (err) { B.err = err; report_error(status, bi, litType >= 2 && (err == kErrDictionaryCorrupted || litSize > dstSize) ? kStageLiterals : kStageSequences, err); }
        }
    }
    if (lane == 0) {
        F.hasSeq = hasSeq;
        if (litAcc) atomicAdd(reinterpret_cast<unsigned long long*>(status + kStLitLo), (unsigned long long)litAcc);
        // long frames by size class: what the host decides the origin path from (decode_origin.hip); a frame without a content size counts with its bound
        if (hasSeq && dstSize >= (1u << 20) && dstSize < (1ull << 30))
            atomicAdd(reinterpret_cast<unsigned long long*>(status + kStBigBins) + highbit32((u32)(dstSize >> 20)), (unsigned long long)dstSize);
    }
}

// ------------------------------------------------------------------------------------------------
// seq_scan: where every block's sequence records go (exclusive scan of nbSeq), single workgroup
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void seq_scan_kernel(BlockDesc* __restrict__ blocks, u32 nBlocks, u32* __restrict__ status)
{
    __shared__ u64 shW[16];
    const u32 tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    u64 carry = 0;
    for (u32 base = 0; base < nBlocks; base += 1024) {
        const u32 i = base + tid;
        u64 v = 0;
        if (i < nBlocks) { const BlockDesc& B = blocks[i]; v = (B.type == 2 && !B.err) ? B.nbSeq : 0u; }
        u64 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(incl, d); if ((int)lane >= d) incl += t; }
        if (lane == 63) shW[wave] = incl;
        __syncthreads();
        u64 before = carry, all = 0;
        for (u32 w = 0; w < 16; ++w) { if (w < wave) before += shW[w]; all += shW[w]; }
        if (i < nBlocks) blocks[i].seqBase = before + incl - v;
        carry += all;
        __syncthreads();
    }
    if (tid == 0) { status[kStSeqLo] = (u32)carry; status[kStSeqHi] = (u32)(carry >> 32); }
}

void launch_block_prepass(const u8* src, FrameDesc* frames, BlockDesc* blocks, u32 nFrames, u32 nBlocks, u32 haveDict, u32 earlyLiterals, u32* status, hipStream_t stream)
{
    hipLaunchKernelGGL(block_parse_kernel, dim3((nBlocks + 255) / 256), dim3(256), 0, stream, src, blocks, nBlocks, status);
    hipLaunchKernelGGL(block_link_small_kernel, dim3((nFrames + 63) / 64), dim3(64), 0, stream, frames, blocks, nFrames, haveDict, earlyLiterals, status);
    if (nBlocks > nFrames) hipLaunchKernelGGL(block_link_kernel, dim3(nFrames), dim3(64), 0, stream, frames, blocks, nFrames, haveDict, earlyLiterals, status);      // (some frame has several blocks)
    hipLaunchKernelGGL(seq_scan_kernel, dim3(1), dim3(1024), 0, stream, blocks, nBlocks, status);
}

// ------------------------------------------------------------------------------------------------
// block_offsets: output offsets and starting repcodes of the blocks of a frame (one wave per frame, after seq_decode)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void block_offsets_kernel(FrameDesc* __restrict__ frames, BlockDesc* __restrict__ blocks, u32 nFrames,
                                                           const DictInfo* __restrict__ di, u32* __restrict__ status)
{
    const u32 f = blockIdx.x, lane = threadIdx.x;
    if (f >= nFrames) return;
    const u32 first = frames[f].firstBlock, nb = frames[f].nbBlocks;
    u32 r0 = 1, r1 = 4, r2 = 8;                                // ZSTD_decompressBegin, U/ZstdDecompress.cs:1933-1954
    if (di) { r0 = uniform(di->rep[0]); r1 = uniform(di->rep[1]); r2 = uniform(di->rep[2]); }
    u64 acc = 0; u32 bad = 0;
    for (u32 k0 = 0; k0 < nb; k0 += 64) {
        const u32 k = k0 + lane; const bool have = k < nb;
        const u32 bi = first + (have ? k : 0);
        u32 outSize = 0, err = 0, kinds = 0, v0 = 0, v1 = 0, v2 = 0; bool hasRep = false;
        if (have) {
            const BlockDesc& B = blocks[bi];
            outSize = B.outSize; err = B.err;
            hasRep = B.type == 2 && B.nbSeq != 0 && !B.err;
            if (hasRep) { kinds = B.repKind[0] | (B.repKind[1] << 2) | (B.repKind[2] << 4); v0 = B.repVal[0]; v1 = B.repVal[1]; v2 = B.repVal[2]; }
        }
        if (ballot(err != 0)) bad = 1;
        // output offsets: exclusive prefix sum of the regenerated sizes
        u64 incl = outSize;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(incl, d); if ((int)lane >= d) incl += t; }
        if (have) blocks[bi].dstRel = acc + incl - outSize;
        acc += __shfl(incl, 63);
        // repcodes: the blocks' transfer functions applied in order (wave-uniform; blocks without sequences pass them through)
        u32 in0 = r0, in1 = r1, in2 = r2;                      // what this lane's block starts from
        u64 m = ballot(have);
        while (m) {
            const u32 l = ctz64(m); m &= m - 1;
            if (lane == l) { in0 = r0; in1 = r1; in2 = r2; }
            if (read_lane((u32)hasRep, l)) {
                const u32 kk = read_lane(kinds, l), a0 = read_lane(v0, l), a1 = read_lane(v1, l), a2 = read_lane(v2, l);
                auto apply = [&](u32 kind, u32 val) -> u32 {
                    if (kind == 0) return val;
                    const u32 in = kind == 1 ? r0 : kind == 2 ? r1 : r2;
                    return in > val ? in - val : 1u;
                };
                const u32 n0 = apply(kk & 3, a0), n1 = apply((kk >> 2) & 3, a1), n2 = apply((kk >> 4) & 3, a2);
                r0 = n0; r1 = n1; r2 = n2;
            }
        }
        if (have) { BlockDesc& B = blocks[bi]; B.repIn[0] = in0; B.repIn[1] = in1; B.repIn[2] = in2; }
    }
    if (lane == 0) {
        FrameDesc& F = frames[f];
        if (bad) F.bad = 1;
        else if (!F.unsized && acc != F.dstSize) {              // regenerated size must equal the header's (U/ZstdDecompress.cs:1177-1184)
            F.bad = 1; report_error(status, (u64)first + nb - 1, kStageFrameEnd, kErrCorruption);
        } else if (F.unsized) {
            if (acc > F.dstSize) { F.bad = 1; report_error(status, (u64)first + nb - 1, kStageFrameEnd, kErrCorruption); }   // more than nbBlocks x blockSizeMax: no valid encoder
            else F.dstSize = acc;
        }
    }
}

// frames without a content size: the frames' output offsets from their regenerated sizes (single workgroup); the total goes to
// status, and a total beyond the destination's capacity stops everything behind this kernel
__global__ __launch_bounds__(1024) void frame_rescan_kernel(FrameDesc* __restrict__ frames, u32 nFrames, u64 dstCapacity, u32* __restrict__ status)
{
    __shared__ u64 shW[16];
    const u32 tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    u64 carry = 0;
    for (u32 base = 0; base < nFrames; base += 1024) {
        const u32 i = base + tid;
        const u64 v = i < nFrames ? frames[i].dstSize : 0;
        u64 incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(incl, d); if ((int)lane >= d) incl += t; }
        if (lane == 63) shW[wave] = incl;
        __syncthreads();
        u64 before = carry, all = 0;
        for (u32 w = 0; w < 16; ++w) { if (w < wave) before += shW[w]; all += shW[w]; }
        if (i < nFrames) frames[i].dstOff = before + incl - v;
        carry += all;
        __syncthreads();
    }
    if (tid == 0) {
        status[kStActualLo] = (u32)carry; status[kStActualHi] = (u32)(carry >> 32);
        if (carry > dstCapacity) status[kStErr] = kErrDstSizeTooSmall;
    }
}

void launch_block_offsets(FrameDesc* frames, BlockDesc* blocks, u32 nFrames, const DictInfo* di, u32 rescan, u64 dstCapacity, u32* status, hipStream_t stream)
{
    hipLaunchKernelGGL(block_offsets_kernel, dim3(nFrames), dim3(64), 0, stream, frames, blocks, nFrames, di, status);
    if (rescan) hipLaunchKernelGGL(frame_rescan_kernel, dim3(1), dim3(1024), 0, stream, frames, nFrames, dstCapacity, status);
}

// ZSTD_loadDEntropy's checks (U/ZstdDecompress.cs:1773-1875) on one lane: where the Huffman description and the three NCounts
// sit, the repcodes, where the content starts; err = dictionary_corrupted if anything is off.
struct DictScratch { u8 weights[256]; s16 norm[256]; u16 symbolNext[256]; u16 wNewState[64]; u8 wSymbol[64]; u8 wNbBits[64]; };
__global__ __launch_bounds__(64) void dict_parse_kernel(const u8* __restrict__ dict, u32 dictSize, DictInfo* __restrict__ out)
{
    __shared__ DictScratch L;
    __shared__ s16 norm[64];
    const u32 lane = threadIdx.x;
    if (lane != 0) return;
    DictInfo d = {}; d.err = kErrDictionaryCorrupted;
    do {
        if (dictSize <= 8) break;
        d.dictID = readLE32(dict + 4);
        u32 nbSymbols = 0, tableLog = 0;
        const u32 hs = huf_read_stats(L, dict + 8, dictSize - 8, &nbSymbols, &tableLog);
        if (!hs || tableLog > 12) break;
        d.hufOff = 8; d.hufSize = hs;
        u32 p = 8 + hs, maxSV, log, h;
        maxSV = 31; h = read_ncount(norm, &maxSV, &log, dict + p, dictSize - p);
        if (!h || maxSV > 31 || log > 8) break;
        d.ofOff = p; p += h;
        maxSV = 52; h = read_ncount(norm, &maxSV, &log, dict + p, dictSize - p);
        if (!h || maxSV > 52 || log > 9) break;
        d.mlOff = p; p += h;
        maxSV = 35; h = read_ncount(norm, &maxSV, &log, dict + p, dictSize - p);
        if (!h || maxSV > 35 || log > 9) break;
        d.llOff = p; p += h;
        if (p + 12 > dictSize) break;
        d.repOff = p;
        d.contentOff = p + 12; d.contentSize = dictSize - d.contentOff;
        bool ok = true;
        for (u32 i = 0; i < 3; i++) { d.rep[i] = readLE32(dict + p + 4 * i); if (d.rep[i] == 0 || d.rep[i] > d.contentSize) ok = false; }
        if (!ok) break;
        d.err = 0;
    } while (false);
    *out = d;
}
void launch_dict_parse(const u8* dict, u32 dictSize, DictInfo* out, hipStream_t stream)
{
    hipLaunchKernelGGL(dict_parse_kernel, dim3(1), dim3(64), 0, stream, dict, dictSize, out);
}

} // namespace zmi
