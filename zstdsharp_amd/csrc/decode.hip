// decode.hip — zstd frame decoder on gfx950 (SURVEY.md §8 a-13 … a-17).
//
//   frame walk     : ZSTD_findFrameSizeInfo over the whole input (U/ZstdDecompress.cs:877-951, 971-993), headers only,
//                    producing the work list {srcOff, dstOff, srcSize, dstSize} per frame.  Parallel form: the input is
//                    cut into 128 KiB segments; one wave per segment scans for the first frame magic, validates it by
//                    chaining frame -> frame until it leaves the segment, and a scan kernel checks that every
//                    segment's exit is the next segment's entry (anything else — embedded frames inside raw blocks, a
//                    corrupt header — falls back to the exact serial walk, which also produces the reference's error).
//   literals kernels: frames are independent (ZSTD_decompressBegin resets repcodes/tables per frame,
//                    U/ZstdDecompress.cs:1933-1954; blocks inside a frame are not).  Per compressed block: Huffman
//                    weights + X1 table in LDS (HUF_readStats U/EntropyCommon.cs:292-402, HUF_readDTableX1
//                    U/HufDecompress.cs:80-251), then the literal streams (U/HufDecompress.cs:342-537), written to an HBM
//                    scratch.  Two forms: serial (8 frames per wave, a stream per lane, software-pipelined 64-bit
//                    window) for thousands of frames, and self-synchronising (a workgroup per frame, 64 lanes per
//                    stream) for few; see launch_decode_literals.
//   sequences kernel: one wave per frame: FSE tables built by the wave (U/ZstdDecompressBlock.cs:1571-1943), the sequence
//                    state chain on the scalar unit (:2360-2484), 64-lane literal / match execution (:2187-2262).
// Results are bit-exact with the reference decoder by construction of the format; error codes follow
// U/ZSTD_ErrorCode.cs (first failing frame wins).
#include "zmi_device.h"

namespace zmi {

// ------------------------------------------------------------------------------------------------
// frame walk
// ------------------------------------------------------------------------------------------------
struct FrameHeader { u64 contentSize; u64 windowSize; u32 headerSize; u32 checksum; u32 dictID; u32 err; };

__device__ inline FrameHeader parse_frame_header(const u8* p, u64 avail)
{
    FrameHeader h; h.contentSize = ~0ull; h.windowSize = 0; h.headerSize = 0; h.checksum = 0; h.dictID = 0; h.err = 0;
    if (avail < 5) { h.err = kErrSrcSizeWrong; return h; }
    if (readLE32(p) != 0xFD2FB528u) { h.err = kErrPrefixUnknown; return h; }
    const u8 fhd = p[4];
    const u32 didCode = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
    const u32 didSize = didCode == 3 ? 4 : didCode, fcsSize = fcsId == 0 ? (single ? 1 : 0) : (1u << fcsId);
    const u32 fhs = 5 + !single + didSize + fcsSize;
    if (avail < fhs) { h.err = kErrSrcSizeWrong; return h; }
    if (fhd & 0x08) { h.err = kErrFrameParameterUnsupported; return h; }
    u32 pos = 5;
    if (!single) {
        const u8 wl = p[pos++]; const u32 wlog = (wl >> 3) + 10;
        if (wlog > 31) { h.err = kErrWindowTooLarge; return h; }
        h.windowSize = 1ull << wlog; h.windowSize += (h.windowSize >> 3) * (wl & 7);
    }
    if (didCode == 1) h.dictID = p[pos]; else if (didCode == 2) h.dictID = readLE16(p + pos); else if (didCode == 3) h.dictID = readLE32(p + pos);
    pos += didSize;
    switch (fcsId) {
    case 0: if (single) h.contentSize = p[pos]; break;
    case 1: h.contentSize = (u64)readLE16(p + pos) + 256; break;
    case 2: h.contentSize = readLE32(p + pos); break;
    default: h.contentSize = readLE64(p + pos); break;
    }
    if (single) h.windowSize = h.contentSize;
    h.headerSize = fhs; h.checksum = (fhd >> 2) & 1;
    return h;
}

__global__ void frame_walk_serial_kernel(const u8* __restrict__ src, u64 srcSize, FrameDesc* __restrict__ frames, u32 maxFrames, u32* __restrict__ status,
                                         u32 dictID)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    u64 pos = 0, dstOff = 0; u32 n = 0, err = 0, nUnsized = 0;
    while (srcSize - pos >= 5) {            // ZSTD_decompressMultiFrame loop condition (U/ZstdDecompress.cs:1228)
        const u8* p = src + pos; const u64 avail = srcSize - pos;
        const u32 magic = readLE32(p);
        if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
            if (avail < 8) { err = kErrSrcSizeWrong; break; }
            const u64 sz = (u64)readLE32(p + 4) + 8;
            if (sz > avail) { err = kErrSrcSizeWrong; break; }
            pos += sz; continue;
        }
        const FrameHeader h = parse_frame_header(p, avail);
        if (h.err) { err = (h.err == kErrPrefixUnknown && n > 0) ? kErrSrcSizeWrong : h.err; break; }
        if (h.dictID && h.dictID != dictID) { err = kErrDictionaryWrong; break; }       // U/ZstdDecompress.cs:1404-1412 (dictID 0 = none loaded)
        u64 q = pos + h.headerSize; u64 nbBlocks = 0;
        for (;;) {
            if (srcSize - q < 3) { err = kErrSrcSizeWrong; break; }
            const u32 bh = readLE24(src + q);
            const u32 last = bh & 1, type = (bh >> 1) & 3; u32 cSize = bh >> 3;
            if (type == 3) { err = kErrCorruption; break; }
            if (type == 1) cSize = 1;
            if (3 + (u64)cSize > srcSize - q) { err = kErrSrcSizeWrong; break; }
            q += 3 + cSize; nbBlocks++;
            if (last) break;
        }
        if (err) break;
        if (h.checksum) { if (srcSize - q < 4) { err = kErrSrcSizeWrong; break; } q += 4; }
        // a frame without a content size gets the bound ZSTD_findFrameSizeInfo gives it: nbBlocks x min(window, 128 KiB)
        const bool unsizedF = h.contentSize == ~0ull;
        const u64 content = unsizedF ? nbBlocks * (h.windowSize < (1u << 17) ? h.windowSize : (u64)(1u << 17)) : h.contentSize;
        if (n >= maxFrames || (q - pos) > 0xFFFFFFFFull || content > 0xFFFFFFFFull) { err = kErrMemoryAllocation; break; }
        FrameDesc f; f.srcOff = pos; f.dstOff = dstOff; f.srcSize = (u32)(q - pos); f.dstSize = (u32)content; f.unsized = unsizedF; f.pad = 0;
        frames[n++] = f; nUnsized += unsizedF;
        dstOff += content; pos = q;
    }
    if (!err && pos != srcSize) err = kErrSrcSizeWrong;     // trailing garbage (U/ZstdDecompress.cs:1309-1312)
    status[0] = n; status[1] = err; status[2] = (u32)dstOff; status[3] = (u32)(dstOff >> 32); status[5] = nUnsized;
}

// ------------------------------------------------------------------------------------------------
// backward bit reader over global memory (U/Bitstream.cs:172-426)
// ------------------------------------------------------------------------------------------------
struct BackBits {
    const u8* base; s32 size;   // stream bytes
    s32 pos;                    // unread bits; < 0 after an over-read (the reference's BIT_DStream_overflow)
    u64 win; s32 wStart;        // 64 stream bits starting at bit wStart

    __device__ __forceinline__ void load_window(s32 endBit)        // window that ends at the byte holding endBit-1
    {
        s32 endByte = (endBit + 7) >> 3;
        if (endByte > size) endByte = size;
        s32 b0 = endByte - 8;
        if (b0 >= 0) { win = readLE64(base + b0); wStart = b0 * 8; }
        else {
            u64 v = 0;
            for (s32 i = 0; i < 8; i++) { const s32 k = b0 + i; if (k >= 0 && k < size) v |= (u64)base[k] << (8 * i); }
            win = v; wStart = b0 * 8;                                // negative start: low bits read as zero
        }
    }
    // returns false when the stream is malformed (empty, or no end mark)
    __device__ __forceinline__ bool init(const u8* p, s32 n)
    {
        base = p; size = n; pos = 0; win = 0; wStart = 0;
        if (n < 1) return false;
        const u32 last = p[n - 1];
        if (last == 0) return false;
        pos = (n - 1) * 8 + (s32)highbit32(last);
        load_window(pos);
        return true;
    }
    __device__ __forceinline__ u32 peek(u32 nb)                      // next nb bits (nb <= 32), zeros below bit 0
    {
        const s32 lo = pos - (s32)nb;
        if (lo < wStart) load_window(pos);
        const s32 sh = lo - wStart;
        const u64 v = sh >= 0 ? (win >> sh) : (win << (-sh));       // sh < 0 only when reading below the stream start
        return nb ? (u32)(v & ((1ull << nb) - 1)) : 0u;
    }
    __device__ __forceinline__ u32 read(u32 nb) { const u32 v = peek(nb); pos -= (s32)nb; return v; }
};

// ------------------------------------------------------------------------------------------------
// per-wave decoder state in LDS
// ------------------------------------------------------------------------------------------------
#ifdef ZMI_LZ_STAMPS
__device__ unsigned long long g_seqStamps[16];
#define ZMI_SSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stampAcc[i] += now_ - stampLast; stampLast = now_; } while (0)
#else
#define ZMI_SSTAMP(i) do { } while (0)
#endif

struct SeqSym { u16 nextState; u8 nbAddBits; u8 nbBits; u32 baseValue; };

// literals kernel: 5.5 KiB per wave
struct LitLds {
    u16 huf[2048];              // X1 table indexed by 11 bits: byte | nbBits << 8, or 0xF000 | pair for 12-bit codes
    u16 pair[128][2];           // tableLog 12 only: the two 12-bit symbols that share an 11-bit prefix
    u8  weights[256];
    s16 norm[256];
    u16 symbolNext[256];
    u32 rankStart[16];
    u32 hufLog, hufValid;
    u16 wNewState[64]; u8 wSymbol[64]; u8 wNbBits[64];    // FSE scratch for Huffman weights (tableLog <= 6)
};
// sequences kernel: 10.6 KiB per wave
struct SeqLds {
    SeqSym ll[512], ml[512], of[256];
    s16 norm[64];
    u16 symbolNext[64];
    u32 llLog, mlLog, ofLog;
    u32 llValid, mlValid, ofValid;
};

__constant__ u8  dLL_bits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
__constant__ u32 dLL_base[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
__constant__ u8  dML_bits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
__constant__ u32 dML_base[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                  35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };
__constant__ s16 dLL_defaultNorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
__constant__ s16 dML_defaultNorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                         1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
__constant__ s16 dOF_defaultNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

__device__ __forceinline__ u32 of_base(u32 code) { return code == 0 ? 0u : code == 1 ? 1u : (1u << code) - 3u; }   // OF_base, U/ZstdDecompressInternal.cs:85

// forward bit cursor (FSE_readNCount_body, U/EntropyCommon.cs:52-242); zeros beyond `size`
__device__ __forceinline__ u32 fwd_bits(const u8* p, u32 size, u32 bitpos, u32 n)
{
    u64 acc = 0; const u32 b0 = bitpos >> 3;
    if (b0 + 8 <= size) acc = readLE64(p + b0);
    else for (u32 i = 0; i < 8; i++) if (b0 + i < size) acc |= (u64)p[b0 + i] << (8 * i);
    return (u32)((acc >> (bitpos & 7)) & ((1ull << n) - 1));
}

// returns bytes consumed, 0 on error
__device__ inline u32 read_ncount(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const u8* ip, u32 srcSize)
{
    u32 bitpos, nbBits, remaining, threshold, charnum = 0; const u32 maxSV1 = *maxSVPtr + 1; bool previous0 = false;
    if (srcSize < 1) return 0;
    for (u32 s = 0; s < maxSV1; s++) norm[s] = 0;
    nbBits = fwd_bits(ip, srcSize, 0, 4) + 5;
    if (nbBits > 15) return 0;
    bitpos = 4; *tableLogPtr = nbBits;
    remaining = (1u << nbBits) + 1; threshold = 1u << nbBits; nbBits++;
    for (;;) {
        if (previous0) {
            for (;;) {
                const u32 r = fwd_bits(ip, srcSize, bitpos, 2);
                bitpos += 2; charnum += r;
                if (r != 3) break;
                if (bitpos > srcSize * 8 + 32) return 0;
            }
            if (charnum >= maxSV1) break;
        }
        {
            const u32 max = (2 * threshold - 1) - remaining; int count;
            const u32 low = fwd_bits(ip, srcSize, bitpos, nbBits - 1);
            if (low < max) { count = (int)low; bitpos += nbBits - 1; }
            else { u32 v = fwd_bits(ip, srcSize, bitpos, nbBits); if (v >= threshold) v -= max; count = (int)v; bitpos += nbBits; }
            count--;
            remaining -= count >= 0 ? (u32)count : 1u;
            norm[charnum++] = (s16)count;
            previous0 = count == 0;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = highbit32(remaining) + 1; threshold = 1u << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
        }
    }
    if (remaining != 1 || charnum > maxSV1) return 0;
    *maxSVPtr = charnum - 1;
    const u32 used = (bitpos + 7) >> 3;
    return used > srcSize ? 0 : used;
}

// ZSTD_buildFSETable_body (U/ZstdDecompressBlock.cs:1571-1710), one lane
__device__ __forceinline__ SeqSym seq_entry(u32 sym, u32 nextState, u32 tableLog, u32 tableSize, int kind)
{
    SeqSym e; e.nbBits = (u8)(tableLog - highbit32(nextState));
    e.nextState = (u16)((nextState << e.nbBits) - tableSize);
    if (kind == 0) { e.nbAddBits = dLL_bits[sym]; e.baseValue = dLL_base[sym]; }
    else if (kind == 1) { e.nbAddBits = (u8)sym; e.baseValue = of_base(sym); }
    else { e.nbAddBits = dML_bits[sym]; e.baseValue = dML_base[sym]; }
    return e;
}

// ZSTD_buildFSETable_body (U/ZstdDecompressBlock.cs:1571-1710) by all 64 lanes of the frame's wave; lane s stands for
// symbol s (at most 53 symbols).  The reference's three serial passes become:
//   low-probability symbols   -> the top cells, in symbol order (ballot rank);
//   spreading                 -> the reference visits (i*step) & mask for i = 0, 1, 2, ... and skips cells above
//                                highThreshold; the j-th cell it keeps goes to the symbol whose cumulative count covers j
//                                (prefix count of kept cells, then a binary search in the cumulative counts);
//   nextState = symbolNext++  -> cells are taken 64 at a time in index order; within a group the lanes that hold the
//                                same symbol are ranked with a ballot, and the per-symbol counter lives in that symbol's lane.
__device__ __forceinline__ void build_seq_dtable_wave(SeqSym* t, u16* cum, const s16* norm, u32 maxSV, u32 tableLog, int kind, u32 lane)
{
    const u32 tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    const int nrm = lane <= maxSV ? (int)norm[lane] : 0;
    const bool low = nrm == -1;
    const u64 lowMask = ballot(low);
    const u32 highThreshold = tableSize - 1 - popc64(lowMask);
    if (low) t[tableSize - 1 - popc64(lowMask & lanemask_lt())].baseValue = lane;
    const u32 cnt = nrm > 0 ? (u32)nrm : 0;
    const u32 incl = wave_scan_incl(cnt);
    cum[lane] = (u16)(incl - cnt);
    wave_lds_sync();
    u32 jBase = 0;
    for (u32 i0 = 0; i0 < tableSize; i0 += 64) {
        const u32 i = i0 + lane, p = (i * step) & mask;
        const bool place = i < tableSize && p <= highThreshold;
        const u64 bal = ballot(place);
        const u32 j = jBase + popc64(bal & lanemask_lt());
        jBase += popc64(bal);
        if (place) {
            u32 lo = 0, hi = 63;                 // largest symbol whose cumulative count is <= j
#pragma unroll
            for (u32 it = 0; it < 6; ++it) { const u32 mid = (lo + hi + 1) >> 1; if (cum[mid] <= j) lo = mid; else hi = mid - 1; }
            t[p].baseValue = lo;
        }
    }
    wave_lds_sync();
    u32 nxt = low ? 1u : cnt;                    // symbolNext of symbol `lane`
    for (u32 u0 = 0; u0 < tableSize; u0 += 64) {
        const u32 u = u0 + lane; const bool valid = u < tableSize;
        const u32 sym = valid ? t[u].baseValue : 0xFFFFu;
        u32 myNext = 0;
        u64 rem = ballot(valid);
        while (rem) {
            const u32 s0 = read_lane(sym, ctz64(rem));
            const u64 m = ballot(sym == s0);
            const u32 baseN = read_lane(nxt, s0);
            if (sym == s0) myNext = baseN + popc64(m & lanemask_lt());
            nxt = lane == s0 ? nxt + popc64(m) : nxt;
            rem &= ~m;
        }
        wave_lds_sync();                         // every lane has read its cell's symbol before the cells are overwritten
        if (valid) t[u] = seq_entry(sym, myNext, tableLog, tableSize, kind);
    }
    wave_lds_sync();
}

// ZSTD_buildSeqTable (U/ZstdDecompressBlock.cs:1746-1840), whole wave; only the NCount header is parsed by one lane.
// Returns bytes consumed or 0xFFFFFFFF on error (uniform).
__device__ __forceinline__ u32 set_seq_table(SeqLds& L, SeqSym* t, u32* logPtr, u32* validPtr, u32 type, u32 max, u32 maxLog,
                                    const u8* src, u32 srcSize, int kind, const s16* defNorm, u32 defLog, u32 defMax, u32 lane)
{
    switch (type) {
    case 1: {
        if (!srcSize) return 0xFFFFFFFFu;
        const u32 sym = uniform((u32)src[0]);
        if (sym > max) return 0xFFFFFFFFu;
        if (lane == 0) {
            SeqSym e = seq_entry(sym, 1, 0, 1, kind); e.nextState = 0; e.nbBits = 0;
            t[0] = e; *logPtr = 0; *validPtr = 1;
        }
        wave_lds_sync();
        return 1; }
    case 0:
        if (lane <= defMax) L.norm[lane] = defNorm[lane];
        wave_lds_sync();
        build_seq_dtable_wave(t, L.symbolNext, L.norm, defMax, defLog, kind, lane);
        if (lane == 0) { *logPtr = defLog; *validPtr = 1; }
        wave_lds_sync();
        return 0;
    case 3:
        return *validPtr ? 0 : 0xFFFFFFFFu;
    default: {
        u32 maxSV = max, tableLog = 0, hs = 0;
        if (lane == 0) hs = read_ncount(L.norm, &maxSV, &tableLog, src, srcSize);
        hs = uniform(hs); maxSV = uniform(maxSV); tableLog = uniform(tableLog);
        if (!hs || tableLog > maxLog) return 0xFFFFFFFFu;
        wave_lds_sync();
        build_seq_dtable_wave(t, L.symbolNext, L.norm, maxSV, tableLog, kind, lane);
        if (lane == 0) { *logPtr = tableLog; *validPtr = 1; }
        wave_lds_sync();
        return hs; }
    }
}

// HUF_readStats (weights) on lane 0; returns bytes consumed or 0 on error.  nbSymbols/tableLog out.
template <class Scratch>
__device__ inline u32 huf_read_stats(Scratch& L, const u8* ip, u32 srcSize, u32* nbSymbolsPtr, u32* tableLogPtr)
{
    if (!srcSize) return 0;
    u32 iSize = ip[0], oSize;
    if (iSize >= 128) {
        oSize = iSize - 127; iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return 0;
        for (u32 n = 0; n < oSize; n += 2) { L.weights[n] = ip[1 + n / 2] >> 4; L.weights[n + 1] = ip[1 + n / 2] & 15; }
    } else {
        if (iSize + 1 > srcSize) return 0;
        // FSE_decompress_wksp with maxLog 6 (U/FseDecompress.cs:230-446)
        u32 maxSV = 255, tableLog = 0;
        const u32 hs = read_ncount(L.norm, &maxSV, &tableLog, ip + 1, iSize);
        if (!hs || tableLog > 6) return 0;
        {   // FSE_buildDTable
            const u32 tableSize = 1u << tableLog; u32 highThreshold = tableSize - 1;
            for (u32 s = 0; s <= maxSV; s++) {
                if (L.norm[s] == -1) { L.wSymbol[highThreshold--] = (u8)s; L.symbolNext[s] = 1; } else L.symbolNext[s] = (u16)L.norm[s];
            }
            const u32 mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3; u32 pos = 0;
            for (u32 s = 0; s <= maxSV; s++)
                for (int i = 0; i < L.norm[s]; i++) { L.wSymbol[pos] = (u8)s; pos = (pos + step) & mask; while (pos > highThreshold) pos = (pos + step) & mask; }
            if (pos != 0) return 0;
            for (u32 u = 0; u < tableSize; u++) {
                const u32 sym = L.wSymbol[u], nextState = L.symbolNext[sym]++;
                const u32 nb = tableLog - highbit32(nextState);
                L.wNbBits[u] = (u8)nb; L.wNewState[u] = (u16)((nextState << nb) - tableSize);
            }
        }
        BackBits bd;
        if (!bd.init(ip + 1 + hs, (s32)(iSize - hs))) return 0;
        u32 s1 = bd.read(tableLog), s2 = bd.read(tableLog); u32 n = 0;
        for (;;) {
            if (n + 2 > 255) return 0;
            L.weights[n++] = L.wSymbol[s1]; s1 = L.wNewState[s1] + bd.read(L.wNbBits[s1]);
            if (bd.pos < 0) { L.weights[n++] = L.wSymbol[s2]; break; }
            if (n + 2 > 255) return 0;
            L.weights[n++] = L.wSymbol[s2]; s2 = L.wNewState[s2] + bd.read(L.wNbBits[s2]);
            if (bd.pos < 0) { L.weights[n++] = L.wSymbol[s1]; break; }
        }
        oSize = n;
    }
    u32 weightTotal = 0, rank1 = 0;
    for (u32 n = 0; n < oSize; n++) {
        const u32 w = L.weights[n];
        if (w > 12) return 0;
        weightTotal += (1u << w) >> 1; rank1 += w == 1;
    }
    if (!weightTotal) return 0;
    const u32 tableLog = highbit32(weightTotal) + 1;
    if (tableLog > 12) return 0;
    const u32 rest = (1u << tableLog) - weightTotal;
    if ((1u << highbit32(rest)) != rest) return 0;
    const u32 lastWeight = highbit32(rest) + 1;
    L.weights[oSize] = (u8)lastWeight; rank1 += lastWeight == 1;
    if (rank1 < 2 || (rank1 & 1)) return 0;
    *nbSymbolsPtr = oSize + 1; *tableLogPtr = tableLog;
    return iSize + 1;
}

// ------------------------------------------------------------------------------------------------
// Huffman literal streams
// ------------------------------------------------------------------------------------------------
// One stream on one lane (HUF_decodeStreamX1, U/HufDecompress.cs:264-309).  The reference keeps a 64-bit container and
// reloads it every few symbols (BIT_reloadDStream, U/Bitstream.cs:377-419); here the 8 bytes BELOW the container are
// fetched one reload ahead, so the HBM/L2 latency of the next reload is hidden behind the symbols of this one.
// Returns false on corruption (stream not consumed exactly).
template <bool TL12>
__device__ __forceinline__ bool huf_decode_stream(const LitLds& L, u32 tableLog, const u8* __restrict__ src, u32 srcSize, u8* __restrict__ out, u32 n)
{
    if (srcSize < 1) return false;
    const u32 last = src[srcSize - 1];
    if (!last) return false;
    s32 remaining = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);       // unread bits
    const u32 idxBits = TL12 ? 11u : tableLog;
    u32 i = 0; u64 acc = 0;
    auto emit = [&](u32 sym) {
        acc |= (u64)sym << (8 * (i & 7));
        if ((i & 7) == 7) { *(u64u*)(out + i - 7) = acc; acc = 0; }
        i++;
    };
    if (srcSize >= 16) {
        u32 ptr = srcSize - 8;
        u64 cont = readLE64(src + ptr), lower = readLE64(src + ptr - 8);
        u32 consumed = 64u - (u32)(remaining - (s32)(8 * ptr));
        bool lowerValid = true;
        while (i < n) {
            if (consumed > 52) {
                if (!lowerValid) break;
                const u32 k = consumed >> 3;
                cont = k == 8 ? lower : ((cont << (8 * k)) | (lower >> (64 - 8 * k)));
                ptr -= k; consumed -= 8 * k;
                if (ptr >= 8) lower = readLE64(src + ptr - 8); else lowerValid = false;
            }
            const u64 top = cont << consumed;
            u32 e = L.huf[(u32)(top >> (64 - idxBits))];
            if (TL12 && e >= 0xF000u) e = (u32)L.pair[e & 0xFFFu][(u32)(top >> 52) & 1u] | (12u << 8);
            consumed += e >> 8;
            emit(e & 0xFFu);
        }
        remaining = (s32)(8 * ptr) + 64 - (s32)consumed;
    }
    if (i < n) {                         // short stream, or the last bytes of a long one: plain bit reader
        BackBits bd; bd.base = src; bd.size = (s32)srcSize; bd.pos = remaining; bd.load_window(remaining);
        while (i < n) {
            u32 e;
            if (TL12) {
                const u32 v = bd.peek(12);
                e = L.huf[v >> 1];
                if (e >= 0xF000u) e = (u32)L.pair[e & 0xFFFu][v & 1u] | (12u << 8);
            } else e = L.huf[bd.peek(tableLog)];
            bd.pos -= (s32)(e >> 8);
            emit(e & 0xFFu);
        }
        remaining = bd.pos;
    }
    for (u32 k = 0; k < (n & 7); k++) out[(n & ~7u) + k] = (u8)(acc >> (8 * k));
    return remaining == 0;
}

// HUF_readDTableX1 (U/HufDecompress.cs:80-251): rank starts on lane 0, fill by all lanes.  tableLog 12 (legal, never
// produced for zstd literals) is folded into the 11-bit table: codes of 12 bits share an 11-bit prefix pairwise.
__device__ inline void huf_build_table(LitLds& L, u32 nbSymbols, u32 tableLog, u32 lane)
{
    if (lane == 0) {
        u32 cnt[13]; for (int i = 0; i < 13; i++) cnt[i] = 0;
        for (u32 n = 0; n < nbSymbols; n++) cnt[L.weights[n]]++;
        u32 next = 0;
        for (u32 w = 1; w <= tableLog; w++) { L.rankStart[w] = next; next += cnt[w] << (w - 1); }
        for (u32 n = 0; n < nbSymbols; n++) {
            const u32 w = L.weights[n];
            if (w) { L.symbolNext[n] = (u16)L.rankStart[w]; L.rankStart[w] += (1u << w) >> 1; }
        }
        L.hufLog = tableLog; L.hufValid = 1;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    const bool tl12 = tableLog == 12;
    for (u32 n = lane; n < nbSymbols; n += 64) {
        const u32 w = L.weights[n];
        if (!w) continue;
        const u32 len = (1u << w) >> 1, start = L.symbolNext[n];
        const u16 e = (u16)(n | ((tableLog + 1 - w) << 8));
        if (!tl12) { for (u32 u = 0; u < len; u++) L.huf[start + u] = e; }
        else if (w == 1) { L.pair[start >> 1][start & 1] = (u16)n; L.huf[start >> 1] = (u16)(0xF000u | (start >> 1)); }
        else { for (u32 u = 0; u < (len >> 1); u++) L.huf[(start >> 1) + u] = e; }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
}

constexpr u32 kBlockMax = 1u << 17;

struct LitHeader { u32 type, lhSize, litSize, litCSize, single, err; };
// ZSTD_decodeLiteralsBlock header parse (U/ZstdDecompressBlock.cs:88-396)
__device__ __forceinline__ LitHeader parse_lit_header(const u8* b, u32 bsz)
{
    LitHeader h; h.err = 0; h.single = 0; h.litCSize = 0;
    h.type = b[0] & 3; const u32 lhl = (b[0] >> 2) & 3;
    if (h.type >= 2) {
        if (bsz < 5) { h.err = kErrCorruption; return h; }
        const u32 lhc = readLE32(b);
        switch (lhl) {
        case 0: case 1: h.single = !lhl; h.lhSize = 3; h.litSize = (lhc >> 4) & 0x3FF; h.litCSize = (lhc >> 14) & 0x3FF; break;
        case 2: h.lhSize = 4; h.litSize = (lhc >> 4) & 0x3FFF; h.litCSize = lhc >> 18; break;
        default: h.lhSize = 5; h.litSize = (lhc >> 4) & 0x3FFFF; h.litCSize = (lhc >> 22) + ((u32)b[4] << 10); break;
        }
        if (h.litSize > kBlockMax || h.litCSize + h.lhSize > bsz) h.err = kErrCorruption;
    } else {
        switch (lhl) {
        case 0: case 2: h.lhSize = 1; h.litSize = b[0] >> 3; break;
        case 1: h.lhSize = 2; h.litSize = readLE16(b) >> 4; break;
        default: h.lhSize = 3; h.litSize = readLE24(b) >> 4; break;
        }
        if (h.litSize > kBlockMax) h.err = kErrCorruption;
        else if (h.type == 0 ? (h.lhSize + h.litSize > bsz) : (h.lhSize + 1 > bsz)) h.err = kErrCorruption;
    }
    return h;
}

// literals of every compressed block of one frame -> scratch (one wave; every branch is wave-uniform)
// dictFull / di: a formatted dictionary's bytes and layout (null without one): a treeless block that no block of the frame
// precedes takes the dictionary's Huffman table (dctx->litEntropy, U/ZstdDecompress.cs:1925-1929).
__device__ u32 decode_frame_literals(LitLds& L, const FrameDesc fd, const u8* __restrict__ fsrc, u8* __restrict__ litOut, const u32 lane,
                                     const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    const FrameHeader h = parse_frame_header(fsrc, fd.srcSize);
    u32 ip = h.headerSize, litOff = 0;
    if (lane == 0) L.hufValid = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (;;) {
        if (fd.srcSize - ip < 3) return kErrSrcSizeWrong;
        const u32 bh = readLE24(fsrc + ip);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
        ip += 3;
        if (type == 3) return kErrCorruption;
        if (type == 1) { if (1 > fd.srcSize - ip) return kErrSrcSizeWrong; ip += 1; }
        else {
            if (bsz > fd.srcSize - ip) return kErrSrcSizeWrong;
            if (type == 2) {
                if (bsz >= kBlockMax) return kErrSrcSizeWrong;
                if (bsz < 3) return kErrCorruption;
                const u8* const b = fsrc + ip;
                const LitHeader lh = parse_lit_header(b, bsz);
                if (lh.err) return lh.err;
                if (lh.type >= 2) {
                    if (lh.litSize > fd.dstSize - litOff) return kErrCorruption;
                    const u8* hsrc = b + lh.lhSize; u32 hlen = lh.litCSize;
                    if (lh.type == 2) {
                        u32 nbSymbols = 0, tableLog = 0, hs = 0;
                        if (lane == 0) hs = huf_read_stats(L, hsrc, hlen, &nbSymbols, &tableLog);
                        hs = uniform(hs); nbSymbols = uniform(nbSymbols); tableLog = uniform(tableLog);
                        if (!hs || hs >= hlen) return kErrCorruption;
                        huf_build_table(L, nbSymbols, tableLog, lane);
                        hsrc += hs; hlen -= hs;
                    } else if (!uniform(L.hufValid)) {
                        if (!di) return kErrDictionaryCorrupted;
                        u32 nbSymbols = 0, tableLog = 0, hs = 0;
                        if (lane == 0) hs = huf_read_stats(L, dictFull + di->hufOff, di->hufSize, &nbSymbols, &tableLog);
                        hs = uniform(hs); nbSymbols = uniform(nbSymbols); tableLog = uniform(tableLog);
                        if (!hs) return kErrDictionaryCorrupted;
                        huf_build_table(L, nbSymbols, tableLog, lane);
                    }
                    const u32 tableLog = uniform(L.hufLog);
                    u8* const dst = litOut + litOff;
                    bool ok = true;
                    if (lh.single) {
                        if (lane == 0) ok = tableLog == 12 ? huf_decode_stream<true>(L, tableLog, hsrc, hlen, dst, lh.litSize)
                                                           : huf_decode_stream<false>(L, tableLog, hsrc, hlen, dst, lh.litSize);
                    } else {
                        if (hlen < 10) return kErrCorruption;
                        const u32 l1 = readLE16(hsrc), l2 = readLE16(hsrc + 2), l3 = readLE16(hsrc + 4);
                        const u32 seg = (lh.litSize + 3) / 4;
                        if (6 + l1 + l2 + l3 > hlen) return kErrCorruption;
                        if (seg * 3 > lh.litSize) return kErrCorruption;
                        const u32 l4 = hlen - 6 - l1 - l2 - l3;
                        if (lane < 4) {
                            const u32 so = lane == 0 ? 6 : lane == 1 ? 6 + l1 : lane == 2 ? 6 + l1 + l2 : 6 + l1 + l2 + l3;
                            const u32 sl = lane == 0 ? l1 : lane == 1 ? l2 : lane == 2 ? l3 : l4;
                            const u32 on = lane < 3 ? seg : lh.litSize - 3 * seg;
                            ok = tableLog == 12 ? huf_decode_stream<true>(L, tableLog, hsrc + so, sl, dst + lane * seg, on)
                                                : huf_decode_stream<false>(L, tableLog, hsrc + so, sl, dst + lane * seg, on);
                        }
                    }
                    if (ballot(!ok)) return kErrCorruption;
                    litOff += lh.litSize;
                }
            }
            ip += bsz;
        }
        if (last) break;
    }
    return 0;
}

// slow path: one frame per wave, handles everything (incl. 12-bit Huffman tables); only runs for frames the quad kernel flagged
__global__ __launch_bounds__(64) void decode_literals_slow_kernel(const u8* __restrict__ src, u64 srcSize, const FrameDesc* __restrict__ frames,
                                                                  u32 nFrames, u32* __restrict__ frameErr, u8* __restrict__ litScratch, u64 dstCapacity,
                                                                  const u8* __restrict__ slowFlags,
                                                                  const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ LitLds L;
    const u32 f = blockIdx.x, lane = threadIdx.x;
    if (f >= nFrames || (slowFlags && !slowFlags[f])) return;       // slowFlags null: every frame (formatted dictionary loaded)
    const FrameDesc fd = frames[f];
    if (fd.srcOff + fd.srcSize > srcSize || fd.dstOff + fd.dstSize > dstCapacity) { if (lane == 0) atomicCAS(frameErr, 0u, (u32)kErrGeneric); return; }
    const u32 err = decode_frame_literals(L, fd, src + fd.srcOff, litScratch + fd.dstOff + (u64)f * kLitSkew, lane, dictFull, di);
    if (err && lane == 0) atomicCAS(frameErr, 0u, err);
}

// ZSTD_loadDEntropy's checks (U/ZstdDecompress.cs:1773-1875) on one wave: where the Huffman description and the three NCounts
// sit, the repcodes, where the content starts; err = dictionary_corrupted if anything is off.
__global__ __launch_bounds__(64) void dict_parse_kernel(const u8* __restrict__ dict, u32 dictSize, DictInfo* __restrict__ out)
{
    __shared__ LitLds L;
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


// ------------------------------------------------------------------------------------------------
// literals, fast path: kQuads frames per wave, 4 lanes (one per Huffman stream) per frame
// ------------------------------------------------------------------------------------------------
// The four streams of a block are four serial table-lookup chains, so a frame can keep only four lanes busy, and a
// frame needs its 4 KiB X1 table in LDS.  What bounds the kernel is therefore LDS capacity (32 frames = 128 busy
// lanes per CU) and the length of one lookup step; packing 8 frames into a wave lets one wave instruction advance
// 32 streams instead of 4, which is what the one-frame-per-wave form wasted its issue slots on.
constexpr u32 kQuads = 8;
struct QuadLds {
    u16 huf[2048];              // X1 table (tableLog <= 11).  Before it is filled, its storage holds the FSE scratch below.
    u8  weights[256];
    u16 start[256];             // first table index of each symbol
    u32 meta[4];                // hs, nbSymbols, tableLog, valid
};
struct QuadScratch {            // view used by huf_read_stats: FSE scratch aliased onto the (not yet built) table
    u8* weights; s16* norm; u16* symbolNext; u16* wNewState; u8* wSymbol; u8* wNbBits;
};

// 4 symbols per step into one dword.  The loop body is branch-free: the container is re-based on EVERY step (by
// consumed/8 bytes, possibly 0) from `lower`, the 8 stream bytes below it, whose load was issued one step earlier;
// bytes below the stream start read as zero.  Keeping the refill unconditional is what lets the compiler place a
// counted s_waitcnt in front of the use instead of draining every store (vmcnt counts loads and stores together).
__device__ __forceinline__ bool huf_decode_stream4(const u16* __restrict__ table, u32 tableLog, const u8* __restrict__ src, u32 srcSize,
                                                   u8* __restrict__ out, u32 n)
{
    if (srcSize < 1) return false;
    s32 remaining; u32 i = 0;
    if (srcSize >= 16) {
        s32 ptr = (s32)srcSize - 8;                                    // byte index of the container; may go negative at the very end
        u64 cont = readLE64(src + ptr);
        u64 raw = readLE64(src + ptr - 8); s32 lp = ptr - 8;           // the 8 bytes below the container, fetched one step ahead
        const u32 last = (u32)(cont >> 56);                            // (also makes the loop start with both loads retired)
        if (!last) return false;
        remaining = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);
        u32 consumed = 64u - (u32)(remaining - 8 * ptr);
        const u32 sh = 32 - tableLog;
        // one step = 4 symbols -> one dword, then re-base the container and prefetch the next 8 bytes below it
#define ZMI_HUF_STEP(word)                                                                                         \
        {                                                                                                           \
            u32 e;                                                                                                  \
            e = table[(u32)((cont << consumed) >> 32) >> sh]; consumed += e >> 8; word = e & 0xFFu;                 \
            e = table[(u32)((cont << consumed) >> 32) >> sh]; consumed += e >> 8; word |= (e & 0xFFu) << 8;         \
            e = table[(u32)((cont << consumed) >> 32) >> sh]; consumed += e >> 8; word |= (e & 0xFFu) << 16;        \
            e = table[(u32)((cont << consumed) >> 32) >> sh]; consumed += e >> 8; word |= (e & 0xFFu) << 24;        \
            const u64 lower = lp >= 0 ? raw : (lp > -8 ? (raw << (8 * (u32)(-lp))) : 0);                            \
            const u32 k = consumed >> 3;                                                                            \
            cont = (cont << (8 * k)) | (k ? (lower >> (64 - 8 * k)) : 0);                                           \
            ptr -= (s32)k; consumed -= 8 * k;                                                                       \
            lp = ptr - 8;                                                                                           \
            raw = readLE64(src + (lp > 0 ? lp : 0));                                                                \
        }
        // 8 steps per store: vmcnt orders loads and stores together, so every store sits in front of the next
        // prefetch wait; 32 symbols per (2 x 16 B) store keeps that exposure to once per 32 symbols
        while (i + 32 <= n) {
            u32 w0, w1, w2, w3, w4, w5, w6, w7;
            ZMI_HUF_STEP(w0) ZMI_HUF_STEP(w1) ZMI_HUF_STEP(w2) ZMI_HUF_STEP(w3)
            ZMI_HUF_STEP(w4) ZMI_HUF_STEP(w5) ZMI_HUF_STEP(w6) ZMI_HUF_STEP(w7)
            u32u* o = (u32u*)(out + i);
            o[0] = w0; o[1] = w1; o[2] = w2; o[3] = w3; o[4] = w4; o[5] = w5; o[6] = w6; o[7] = w7;
            i += 32;
        }
        while (i + 4 <= n) {
            u32 w;
            ZMI_HUF_STEP(w)
            *(u32u*)(out + i) = w;
            i += 4;
        }
#undef ZMI_HUF_STEP
        remaining = 8 * ptr + 64 - (s32)consumed;
    } else {
        const u32 last = src[srcSize - 1];
        if (!last) return false;
        remaining = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);
    }
    if (i < n && remaining > 0) {
        BackBits bd; bd.base = src; bd.size = (s32)srcSize; bd.pos = remaining; bd.load_window(remaining);
        while (i < n) { const u32 e = table[bd.peek(tableLog)]; bd.pos -= (s32)(e >> 8); out[i++] = (u8)e; }
        remaining = bd.pos;
    }
    return i == n && remaining == 0;
}

// One frame on the 4 lanes of a quad.  Header parsing is done redundantly by the 4 lanes (same loads, same values, so
// the quad's control flow is uniform without any cross-lane traffic); only the weight decoding runs on the quad leader.
// Returns an error code, or 0xFFFF to ask for the slow path (12-bit table).
__device__ u32 quad_decode_literals(QuadLds& Q, const FrameDesc fd, const u8* __restrict__ fsrc, u8* __restrict__ litOut, const u32 ql,
                                    const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    const FrameHeader h = parse_frame_header(fsrc, fd.srcSize);
    u32 ip = h.headerSize, litOff = 0;
    bool haveTable = false; u32 tableLog = 0;
    for (;;) {
        if (fd.srcSize - ip < 3) return kErrSrcSizeWrong;
        const u32 bh = readLE24(fsrc + ip);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
        ip += 3;
        if (type == 3) return kErrCorruption;
        if (type == 1) { if (1 > fd.srcSize - ip) return kErrSrcSizeWrong; ip += 1; }
        else {
            if (bsz > fd.srcSize - ip) return kErrSrcSizeWrong;
            if (type == 2) {
                if (bsz >= kBlockMax) return kErrSrcSizeWrong;
                if (bsz < 3) return kErrCorruption;
                const u8* const b = fsrc + ip;
                const LitHeader lh = parse_lit_header(b, bsz);
                if (lh.err) return lh.err;
                if (lh.type >= 2) {
                    if (lh.litSize > fd.dstSize - litOff) return kErrCorruption;
                    const u8* hsrc = b + lh.lhSize; u32 hlen = lh.litCSize;
                    const bool fromDict = lh.type == 3 && !haveTable && di != nullptr;      // a treeless first block takes the dictionary's table
                    const u8* const tsrc = fromDict ? dictFull + di->hufOff : hsrc; const u32 tlen = fromDict ? di->hufSize : hlen;
                    if (lh.type == 2 || fromDict) {
                        if (ql == 0) {
                            QuadScratch sc;
                            sc.weights = Q.weights; sc.norm = reinterpret_cast<s16*>(Q.huf); sc.symbolNext = Q.huf + 256;
                            sc.wNewState = Q.huf + 512; sc.wSymbol = reinterpret_cast<u8*>(Q.huf + 576); sc.wNbBits = reinterpret_cast<u8*>(Q.huf + 608);
                            u32 nbSymbols = 0, tl = 0;
                            const u32 hs = huf_read_stats(sc, tsrc, tlen, &nbSymbols, &tl);
                            if (hs && tl <= 11) {          // rank starts -> per-symbol first index (HUF_readDTableX1)
                                u32 cnt[13]; for (int i = 0; i < 13; i++) cnt[i] = 0;
                                for (u32 n = 0; n < nbSymbols; n++) cnt[Q.weights[n]]++;
                                u32 rs[13]; u32 next = 0;
                                for (u32 w = 1; w <= tl; w++) { rs[w] = next; next += cnt[w] << (w - 1); }
                                for (u32 n = 0; n < nbSymbols; n++) { const u32 w = Q.weights[n]; if (w) { Q.start[n] = (u16)rs[w]; rs[w] += (1u << w) >> 1; } }
                            }
                            Q.meta[0] = hs; Q.meta[1] = nbSymbols; Q.meta[2] = tl;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                        const u32 hs = Q.meta[0], nbSymbols = Q.meta[1]; tableLog = Q.meta[2];
                        if (!hs || (fromDict ? hs > tlen : hs >= hlen)) return fromDict ? kErrDictionaryCorrupted : kErrCorruption;
                        if (tableLog > 11) return 0xFFFFu;
                        for (u32 n = ql; n < nbSymbols; n += 4) {          // table fill, 4 lanes
                            const u32 w = Q.weights[n];
                            if (!w) continue;
                            const u32 len = (1u << w) >> 1, st = Q.start[n];
                            const u32 e = n | ((tableLog + 1 - w) << 8);
                            if (len >= 4) { const u64 e4 = (u64)(e | (e << 16)) * 0x100000001ull; for (u32 u = 0; u < len; u += 4) *reinterpret_cast<u64*>(&Q.huf[st + u]) = e4; }
                            else for (u32 u = 0; u < len; u++) Q.huf[st + u] = (u16)e;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                        haveTable = true;
                        if (!fromDict) { hsrc += hs; hlen -= hs; }
                    } else if (!haveTable) return kErrDictionaryCorrupted;
                    u8* const dst = litOut + litOff;
                    bool ok = true;
                    if (lh.single) {
                        if (ql == 0) ok = huf_decode_stream4(Q.huf, tableLog, hsrc, hlen, dst, lh.litSize);
                    } else {
                        if (hlen < 10) return kErrCorruption;
                        const u32 l1 = readLE16(hsrc), l2 = readLE16(hsrc + 2), l3 = readLE16(hsrc + 4);
                        const u32 seg = (lh.litSize + 3) / 4;
                        if (6 + l1 + l2 + l3 > hlen) return kErrCorruption;
                        if (seg * 3 > lh.litSize) return kErrCorruption;
                        const u32 l4 = hlen - 6 - l1 - l2 - l3;
                        const u32 so = ql == 0 ? 6 : ql == 1 ? 6 + l1 : ql == 2 ? 6 + l1 + l2 : 6 + l1 + l2 + l3;
                        const u32 sl = ql == 0 ? l1 : ql == 1 ? l2 : ql == 2 ? l3 : l4;
                        const u32 on = ql < 3 ? seg : lh.litSize - 3 * seg;
                        ok = huf_decode_stream4(Q.huf, tableLog, hsrc + so, sl, dst + ql * seg, on);
                    }
                    // any stream of the quad failing fails the frame: combine through LDS (the 4 lanes are converged here)
                    if (ql == 0) Q.meta[3] = 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                    if (!ok) Q.meta[3] = 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                    if (Q.meta[3]) return kErrCorruption;
                    litOff += lh.litSize;
                }
            }
            ip += bsz;
        }
        if (last) break;
    }
    return 0;
}

// =====================================================================================================================
// Literal decoder, serial form, COMPACT tables: what bounds the serial form is LDS (a 4 KiB table per frame admits 32
// frames per CU, so 16 384 frames take two rounds).  Here the table is indexed by at most 10 bits (2 KiB); for tableLog 11
// the longest codes (weight 1) come in pairs that share a 10-bit prefix: their entry is an escape (0xF000 | pair) and one
// more bit picks the symbol out of `sorted`, the list of symbols in (weight, symbol) order whose head IS the pair table.
// 2.4 KiB per frame -> 64 frames per CU -> one round.  Same acceptance as the other forms; tableLog 12 goes the slow way.
// =====================================================================================================================
struct CompactLds {
    u16 huf[1024];              // byte | nbBits << 8; the first n1/2 entries (owned by pairs of 11-bit codes): symbol(bit 0) | symbol(bit 1) << 8.  Before it is filled: FSE scratch (low 1280 B) and the weights (top 256 B)
    u8  sorted[256];            // symbols ordered by (weight, symbol), weight 0 excluded; its head = the 11-bit codes in table order
    u16 classStart[14];         // first index (in the tableLog-bit table) of weight class w; [tableLog + 1] = table size
    u16 classFirst[14];         // index into sorted[] of the first symbol of class w
    u32 meta[4];
};

// n1 = number of 11-bit codes when tableLog = 11 (else 0).  In the canonical order (HUF_readDTableX1: weight classes ascending)
// they own the first n1 entries of the 11-bit table, one each, i.e. the first n1/2 entries of the 10-bit table, two each: such
// an entry holds BOTH symbols (low byte: next bit 0, high byte: next bit 1) and "index < n1/2" says so.  One LDS read per
// symbol whatever the code length, no branch.
__device__ __forceinline__ bool huf_decode_stream4c(const u16* __restrict__ table, const u8* __restrict__ sorted, u32 tableLog, u32 n1,
                                                    const u8* __restrict__ src, u32 srcSize, u8* __restrict__ out, u32 n)
{
    if (srcSize < 1) return false;
    const u32 idxBits = tableLog > 10 ? 10u : tableLog;
    s32 remaining; u32 i = 0;
    const u32 nPair = n1 >> 1;
    (void)sorted;
    auto lookup = [&](u32 top32) -> u32 {                 // top32 = the next 32 stream bits
        const u32 idx = top32 >> (32 - idxBits);
        const u32 sh = (top32 >> 18) & 8u;                // 8 x the bit after the 10 index bits (only used when tableLog = 11)
        const u32 e = table[idx];
        const u32 pairSym = ((e >> sh) & 0xFFu) | (11u << 8);
        return idx < nPair ? pairSym : e;
    };
    if (srcSize >= 16) {
        s32 ptr = (s32)srcSize - 8;
        u64 cont = readLE64(src + ptr);
        u64 raw = readLE64(src + ptr - 8); s32 lp = ptr - 8;
        const u32 last = (u32)(cont >> 56);
        if (!last) return false;
        remaining = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);
        u32 consumed = 64u - (u32)(remaining - 8 * ptr);
#define ZMI_HUF_STEPC(word)                                                                                        \
        {                                                                                                           \
            u32 e;                                                                                                  \
            e = lookup((u32)((cont << consumed) >> 32)); consumed += e >> 8; word = e & 0xFFu;                      \
            e = lookup((u32)((cont << consumed) >> 32)); consumed += e >> 8; word |= (e & 0xFFu) << 8;              \
            e = lookup((u32)((cont << consumed) >> 32)); consumed += e >> 8; word |= (e & 0xFFu) << 16;             \
            e = lookup((u32)((cont << consumed) >> 32)); consumed += e >> 8; word |= (e & 0xFFu) << 24;             \
            const u64 lower = lp >= 0 ? raw : (lp > -8 ? (raw << (8 * (u32)(-lp))) : 0);                            \
            const u32 k = consumed >> 3;                                                                            \
            cont = (cont << (8 * k)) | (k ? (lower >> (64 - 8 * k)) : 0);                                           \
            ptr -= (s32)k; consumed -= 8 * k;                                                                       \
            lp = ptr - 8;                                                                                           \
            raw = readLE64(src + (lp > 0 ? lp : 0));                                                                \
        }
        while (i + 32 <= n) {
            u32 w0, w1, w2, w3, w4, w5, w6, w7;
            ZMI_HUF_STEPC(w0) ZMI_HUF_STEPC(w1) ZMI_HUF_STEPC(w2) ZMI_HUF_STEPC(w3)
            ZMI_HUF_STEPC(w4) ZMI_HUF_STEPC(w5) ZMI_HUF_STEPC(w6) ZMI_HUF_STEPC(w7)
            u32u* o = (u32u*)(out + i);
            o[0] = w0; o[1] = w1; o[2] = w2; o[3] = w3; o[4] = w4; o[5] = w5; o[6] = w6; o[7] = w7;
            i += 32;
        }
        while (i + 4 <= n) {
            u32 w;
            ZMI_HUF_STEPC(w)
            *(u32u*)(out + i) = w;
            i += 4;
        }
#undef ZMI_HUF_STEPC
        remaining = 8 * ptr + 64 - (s32)consumed;
    } else {
        const u32 last = src[srcSize - 1];
        if (!last) return false;
        remaining = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);
    }
    if (i < n && remaining > 0) {
        BackBits bd; bd.base = src; bd.size = (s32)srcSize; bd.pos = remaining; bd.load_window(remaining);
        while (i < n) { const u32 e = lookup(bd.peek(11) << 21); bd.pos -= (s32)(e >> 8); out[i++] = (u8)e; }
        remaining = bd.pos;
    }
    return i == n && remaining == 0;
}

// One frame on the 4 lanes of a quad (compact tables).  Returns an error code, or 0xFFFF to ask for the slow path.
__device__ u32 quad_decode_literals_c(CompactLds& Q, const FrameDesc fd, const u8* __restrict__ fsrc, u8* __restrict__ litOut, const u32 ql,
                                      const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    const FrameHeader h = parse_frame_header(fsrc, fd.srcSize);
    u32 ip = h.headerSize, litOff = 0;
    bool haveTable = false; u32 tableLog = 0, n1 = 0;
    for (;;) {
        if (fd.srcSize - ip < 3) return kErrSrcSizeWrong;
        const u32 bh = readLE24(fsrc + ip);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
        ip += 3;
        if (type == 3) return kErrCorruption;
        if (type == 1) { if (1 > fd.srcSize - ip) return kErrSrcSizeWrong; ip += 1; }
        else {
            if (bsz > fd.srcSize - ip) return kErrSrcSizeWrong;
            if (type == 2) {
                if (bsz >= kBlockMax) return kErrSrcSizeWrong;
                if (bsz < 3) return kErrCorruption;
                const u8* const b = fsrc + ip;
                const LitHeader lh = parse_lit_header(b, bsz);
                if (lh.err) return lh.err;
                if (lh.type >= 2) {
                    if (lh.litSize > fd.dstSize - litOff) return kErrCorruption;
                    const u8* hsrc = b + lh.lhSize; u32 hlen = lh.litCSize;
                    const bool fromDict = lh.type == 3 && !haveTable && di != nullptr;      // a treeless first block takes the dictionary's table
                    const u8* const tsrc = fromDict ? dictFull + di->hufOff : hsrc; const u32 tlen = fromDict ? di->hufSize : hlen;
                    if (lh.type == 2 || fromDict) {
                        u8* const weights = reinterpret_cast<u8*>(Q.huf + 896);     // top 256 B of the table area until the fill
                        if (ql == 0) {
                            QuadScratch sc;
                            sc.weights = weights; sc.norm = reinterpret_cast<s16*>(Q.huf); sc.symbolNext = Q.huf + 256;
                            sc.wNewState = Q.huf + 512; sc.wSymbol = reinterpret_cast<u8*>(Q.huf + 576); sc.wNbBits = reinterpret_cast<u8*>(Q.huf + 608);
                            u32 nbSymbols = 0, tl = 0;
                            const u32 hs = huf_read_stats(sc, tsrc, tlen, &nbSymbols, &tl);
                            if (hs && tl <= 11) {          // class extents (HUF_readDTableX1) and the symbols in (weight, symbol) order
                                for (u32 w = 0; w < 14; w++) { Q.classStart[w] = 0; Q.classFirst[w] = 0; }
                                for (u32 n = 0; n < nbSymbols; n++) Q.classFirst[weights[n]]++;             // counts, for now
                                u32 idx = 0, first = 0;
                                for (u32 w = 1; w <= tl; w++) {
                                    const u32 cnt = Q.classFirst[w];
                                    Q.classStart[w] = idx; Q.classFirst[w] = first;
                                    idx += cnt << (w - 1); first += cnt;
                                }
                                Q.classStart[tl + 1] = idx; Q.classFirst[tl + 1] = first;
                                // running cursors in classFirst[w] while placing; restored afterwards from the class sizes
                                for (u32 n = 0; n < nbSymbols; n++) { const u32 w = weights[n]; if (w) Q.sorted[Q.classFirst[w]++] = (u8)n; }
                                for (u32 w = tl; w >= 1; w--) Q.classFirst[w] = w == 1 ? 0u : Q.classFirst[w - 1];   // cursor of w-1 ended where class w begins
                            }
                            Q.meta[0] = hs; Q.meta[1] = nbSymbols; Q.meta[2] = tl;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                        const u32 hs = Q.meta[0]; tableLog = Q.meta[2];
                        if (!hs || (fromDict ? hs > tlen : hs >= hlen)) return fromDict ? kErrDictionaryCorrupted : kErrCorruption;
                        if (tableLog > 11) return 0xFFFFu;
                        {   // table fill by the quad's 4 lanes, one symbol of `sorted` at a time
                            const u32 nSorted = Q.classFirst[tableLog + 1];
                            const u32 drop = tableLog > 10 ? 1u : 0u;          // 11-bit codes: the table is indexed by the upper 10 bits
                            for (u32 k = ql; k < nSorted; k += 4) {
                                u32 w = 1;
                                for (u32 cw = 2; cw <= tableLog; ++cw) if (Q.classFirst[cw] <= k) w = cw;
                                const u32 start = Q.classStart[w] + ((k - Q.classFirst[w]) << (w - 1));     // index in the tableLog-bit table
                                const u32 sym = Q.sorted[k];
                                if (drop && w == 1) {                   // 11-bit codes: both symbols of the pair in one entry (see huf_decode_stream4c)
                                    if (!(start & 1)) Q.huf[start >> 1] = (u16)(sym | ((u32)Q.sorted[k + 1] << 8));
                                    continue;
                                }
                                const u32 len = ((1u << w) >> 1) >> drop, st = start >> drop;
                                const u32 e = sym | ((tableLog + 1 - w) << 8);
                                if (len >= 4) { const u64 e4 = (u64)(e | (e << 16)) * 0x100000001ull; for (u32 u = 0; u < len; u += 4) *reinterpret_cast<u64*>(&Q.huf[st + u]) = e4; }
                                else for (u32 u = 0; u < len; u++) Q.huf[st + u] = (u16)e;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                        haveTable = true;
                        n1 = tableLog > 10 ? Q.classFirst[2] : 0u;
                        if (!fromDict) { hsrc += hs; hlen -= hs; }
                    } else if (!haveTable) return kErrDictionaryCorrupted;
                    u8* const dst = litOut + litOff;
                    bool ok = true;
                    if (lh.single) {
                        if (ql == 0) ok = huf_decode_stream4c(Q.huf, Q.sorted, tableLog, n1, hsrc, hlen, dst, lh.litSize);
                    } else {
                        if (hlen < 10) return kErrCorruption;
                        const u32 l1 = readLE16(hsrc), l2 = readLE16(hsrc + 2), l3 = readLE16(hsrc + 4);
                        const u32 seg = (lh.litSize + 3) / 4;
                        if (6 + l1 + l2 + l3 > hlen) return kErrCorruption;
                        if (seg * 3 > lh.litSize) return kErrCorruption;
                        const u32 l4 = hlen - 6 - l1 - l2 - l3;
                        const u32 so = ql == 0 ? 6 : ql == 1 ? 6 + l1 : ql == 2 ? 6 + l1 + l2 : 6 + l1 + l2 + l3;
                        const u32 sl = ql == 0 ? l1 : ql == 1 ? l2 : ql == 2 ? l3 : l4;
                        const u32 on = ql < 3 ? seg : lh.litSize - 3 * seg;
                        ok = huf_decode_stream4c(Q.huf, Q.sorted, tableLog, n1, hsrc + so, sl, dst + ql * seg, on);
                    }
                    if (ql == 0) Q.meta[3] = 0;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                    if (!ok) Q.meta[3] = 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                    if (Q.meta[3]) return kErrCorruption;
                    litOff += lh.litSize;
                }
            }
            ip += bsz;
        }
        if (last) break;
    }
    return 0;
}

__global__ __launch_bounds__(64) void decode_literals_compact_kernel(const u8* __restrict__ src, u64 srcSize, const FrameDesc* __restrict__ frames,
                                                                     u32 nFrames, u32* __restrict__ frameErr, u8* __restrict__ litScratch, u64 dstCapacity,
                                                                     u8* __restrict__ slowFlags,
                                                                     const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ CompactLds Qs[kQuads];
    const u32 lane = threadIdx.x, q = lane >> 2, ql = lane & 3;
    const u32 f = blockIdx.x * kQuads + q;
    if (q >= kQuads || f >= nFrames) return;
    const FrameDesc fd = frames[f];
    if (ql == 0) slowFlags[f] = 0;
    if (fd.srcOff + fd.srcSize > srcSize || fd.dstOff + fd.dstSize > dstCapacity) { if (ql == 0) atomicCAS(frameErr, 0u, (u32)kErrGeneric); return; }
    const u32 err = quad_decode_literals_c(Qs[q], fd, src + fd.srcOff, litScratch + fd.dstOff + (u64)f * kLitSkew, ql, dictFull, di);
    if (ql == 0) {
        if (err == 0xFFFFu) slowFlags[f] = 1;
        else if (err) atomicCAS(frameErr, 0u, err);
    }
}

__global__ __launch_bounds__(64) void decode_literals_kernel(const u8* __restrict__ src, u64 srcSize, const FrameDesc* __restrict__ frames,
                                                             u32 nFrames, u32* __restrict__ frameErr, u8* __restrict__ litScratch, u64 dstCapacity,
                                                             u8* __restrict__ slowFlags,
                                                             const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ QuadLds Qs[kQuads];
    const u32 lane = threadIdx.x, q = lane >> 2, ql = lane & 3;
    const u32 f = blockIdx.x * kQuads + q;
    if (q >= kQuads || f >= nFrames) return;
    const FrameDesc fd = frames[f];
    if (ql == 0) slowFlags[f] = 0;
    if (fd.srcOff + fd.srcSize > srcSize || fd.dstOff + fd.dstSize > dstCapacity) { if (ql == 0) atomicCAS(frameErr, 0u, (u32)kErrGeneric); return; }
    const u32 err = quad_decode_literals(Qs[q], fd, src + fd.srcOff, litScratch + fd.dstOff + (u64)f * kLitSkew, ql, dictFull, di);
    if (ql == 0) {
        if (err == 0xFFFFu) slowFlags[f] = 1;
        else if (err) atomicCAS(frameErr, 0u, err);
    }
}

// ------------------------------------------------------------------------------------------------
// sequences
// ------------------------------------------------------------------------------------------------
// 64-lane copies inside one wave; both sides may be arbitrarily aligned (gfx950 handles unaligned 8-byte accesses)
__device__ __forceinline__ void wave_copy(u8* __restrict__ d, const u8* __restrict__ s, u32 n, u32 lane)
{
    if (n <= 64) { if (lane < n) d[lane] = s[lane]; return; }
    const u32 chunks = n >> 4;
    u32 i = lane;
    // four 16-byte pieces per lane in flight (4 KiB per wave) before the first store: one load latency per 4 KiB, not per 1 KiB
    for (; i + 192 < chunks; i += 256) {
        const u64 a0 = readLE64(s + 16 * i), b0 = readLE64(s + 16 * i + 8);
        const u64 a1 = readLE64(s + 16 * (i + 64)), b1 = readLE64(s + 16 * (i + 64) + 8);
        const u64 a2 = readLE64(s + 16 * (i + 128)), b2 = readLE64(s + 16 * (i + 128) + 8);
        const u64 a3 = readLE64(s + 16 * (i + 192)), b3 = readLE64(s + 16 * (i + 192) + 8);
        *(u64u*)(d + 16 * i) = a0; *(u64u*)(d + 16 * i + 8) = b0;
        *(u64u*)(d + 16 * (i + 64)) = a1; *(u64u*)(d + 16 * (i + 64) + 8) = b1;
        *(u64u*)(d + 16 * (i + 128)) = a2; *(u64u*)(d + 16 * (i + 128) + 8) = b2;
        *(u64u*)(d + 16 * (i + 192)) = a3; *(u64u*)(d + 16 * (i + 192) + 8) = b3;
    }
    for (; i < chunks; i += 64) {
        const u64 a = readLE64(s + 16 * i), b = readLE64(s + 16 * i + 8);
        *(u64u*)(d + 16 * i) = a; *(u64u*)(d + 16 * i + 8) = b;
    }
    const u32 done = chunks << 4;
    if (lane < n - done) d[done + lane] = s[done + lane];
}
// exact n-byte copy by ONE lane in 8-byte pieces (the last piece overlaps the one before instead of a byte tail);
// source and destination do not overlap
__device__ __forceinline__ void lane_copy(u8* __restrict__ d, const u8* __restrict__ s, u32 n)
{
    if (n >= 8) {
        for (u32 i = 0; i + 8 < n; i += 8) *(u64u*)(d + i) = readLE64(s + i);
        *(u64u*)(d + n - 8) = readLE64(s + n - 8);
    } else if (n >= 4) {
        const u32 a = readLE32(s), b = readLE32(s + n - 4);
        *(u32u*)d = a; *(u32u*)(d + n - 4) = b;
    } else if (n >= 2) {
        const u32 a = readLE16(s), b = readLE16(s + n - 2);
        writeLE16(d, a); writeLE16(d + n - 2, b);
    } else if (n) d[0] = s[0];
}
// one lane's match of n bytes at distance `offset` (ZSTD_execSequence's overlap semantics): 8-byte pieces when the
// distance allows it, bytes otherwise
__device__ __forceinline__ void lane_match_copy(u8* d, u32 offset, u32 n)
{
    const u8* s0 = d - offset;
    if (offset >= n) { lane_copy(d, s0, n); return; }
    if (offset >= 8) {
        // n > offset >= 8: pieces in order, each reading bytes that earlier pieces of this lane have written
        u32 i = 0;
        for (; i + 8 <= n; i += 8) *(u64u*)(d + i) = readLE64(s0 + i);
        for (; i < n; i++) d[i] = s0[i];
        return;
    }
    for (u32 i = 0; i < n; i++) d[i] = s0[i % offset];
}
// match copy with the byte-wise overlap semantics of ZSTD_execSequence (U/ZstdDecompressBlock.cs:2247-2259): byte i of
// the match equals the byte `offset` behind it, i.e. src0[i % offset] over the bytes that existed before the match.
__device__ __forceinline__ void wave_match_copy(u8* d, u32 offset, u32 n, u32 lane)
{
    const u8* s0 = d - offset;
    if (offset >= n) { wave_copy(d, s0, n, lane); return; }
    for (u32 i = lane; i < n; i += 64) d[i] = s0[i % offset];
}

// Wave-uniform reader of the backward sequence bitstream, for the state chain.  The stream is seen as dwords (dword d =
// stream bytes 4d..4d+3, zero outside the stream); lane l of `winCur` holds dword wbase + l and `winNext` the window 32
// dwords lower, fetched one rotation ahead.  Bits are taken straight out of the window with two v_readlane and a scalar
// 64-bit shift: no container to maintain, no refill branches, and the position arithmetic stays on the scalar unit
// (which issues beside the vector unit: the chain is issue-bound, so the work is split between the two on purpose).
struct SBits {
    const u8* s; s32 size;
    s32 pos;                    // stream bit index one past the next bit to read; may go negative (reads zeros)
    s32 wbase;
    u32 winCur, winNext;        // per lane
    __device__ __forceinline__ u32 load_dword_z(s32 d) const
    {
        const s32 b = 4 * d;
        if (size >= 4) {        // uniform; branch-free inside: clamp the address, then shift or zero what lies outside
            const s32 hiB = size - 4;
            const u32 v = readLE32(s + (b < 0 ? 0 : (b > hiB ? hiB : b)));
            const u32 part = b > hiB ? (b < size ? v >> (8 * (u32)(b - hiB)) : 0u) : v;
            return b < 0 ? 0u : part;
        }
        u32 v = 0;
        for (s32 i = 0; i < 4; i++) { const s32 k = b + i; if (k >= 0 && k < size) v |= (u32)s[k] << (8 * i); }
        return v;
    }
    __device__ __forceinline__ bool init(const u8* p, s32 n, u32 lane)
    {
        s = p; size = n; pos = 0; wbase = 0; winCur = 0; winNext = 0;
        if (n < 1) return false;
        const u32 last = uniform((u32)p[n - 1]);
        if (!last) return false;
        pos = (n - 1) * 8 + (s32)highbit32(last);
        wbase = ((pos - 1) >> 5) - 62;             // the dword holding the first bit sits at lane 62
        winCur = load_dword_z(wbase + (s32)lane); winNext = load_dword_z(wbase - 32 + (s32)lane);
        return true;
    }
    // the 64 stream bits from bit q upward, q >= pos - 96 (the window is rotated when q falls below it)
    __device__ __forceinline__ u64 peek(s32 q, u32 lane)
    {
        const s32 d = q >> 5;
        if (d < wbase) { winCur = winNext; wbase -= 32; winNext = load_dword_z(wbase - 32 + (s32)lane); }
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)winCur, d - wbase);
        const u32 hi = (u32)__builtin_amdgcn_readlane((int)winCur, d + 1 - wbase);
        return (((u64)hi << 32) | lo) >> (u32)(q & 31);
    }
    __device__ __forceinline__ u32 read(u32 nb, u32 lane)       // nb <= 32
    {
        pos -= (s32)nb;
        return nb ? (u32)peek(pos, lane) & (0xFFFFFFFFu >> (32 - nb)) : 0u;
    }
    // bits [p - nb, p) of the stream for an arbitrary lane-private p (the extra-bit fields, read by the sequence's own lane)
    __device__ __forceinline__ u32 field(s32 p, u32 nb) const
    {
        if (!nb) return 0;
        const s32 q = p - (s32)nb, by = q >> 3;
        u64 v;
        if (by >= 0 && by + 8 <= size) v = readLE64(s + by);
        else { v = 0; for (s32 i = 0; i < 8; i++) { const s32 k = by + i; if (k >= 0 && k < size) v |= (u64)s[k] << (8 * i); } }
        return (u32)(v >> (u32)(q & 7)) & (0xFFFFFFFFu >> (32 - nb));
    }
};
__device__ __forceinline__ u64 uniform64(u64 v) { return (u64)uniform((u32)v) | ((u64)uniform((u32)(v >> 32)) << 32); }


// Sequences of one frame on one wave.  Per block, 64 sequences at a time: the FSE state chain (ZSTD_decodeSequence,
// U/ZstdDecompressBlock.cs:2360-2484) runs wave-uniform on the scalar unit and records states + bit position per
// sequence; every lane extracts the fields of its own sequence; repcodes are resolved in order; then all 64 lanes execute
// the batch (ZSTD_execSequence, :2187-2262): output positions by prefix sum, literals (long runs in 16-byte pieces dealt
// round-robin), matches in dependency rounds.
__device__ u32 decode_frame_sequences(SeqLds& L, const FrameDesc fd, const u8* __restrict__ fsrc, u8* __restrict__ out,
                                      const u8* __restrict__ litIn, const u32 lane, u32* actualOut,
                                      const u8* __restrict__ dict, const u32 dictSize,
                                      const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
#define FAIL(code) return (code)
#ifdef ZMI_LZ_STAMPS
    unsigned long long stampAcc[8] = {0,0,0,0,0,0,0,0}; unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif
    const FrameHeader h = parse_frame_header(fsrc, fd.srcSize);
    u32 ip = h.headerSize, op = 0, litOff = 0;
    u32 rep0 = 1, rep1 = 4, rep2 = 8;
    if (lane == 0) { L.llValid = L.mlValid = L.ofValid = 0; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (di) {
        // a formatted dictionary: every frame starts from its three FSE tables and its repcodes (dctx->fseEntropy,
        // ZSTD_decompressBegin_usingDict, U/ZstdDecompress.cs:1956-1990)
        const u32 ofOff = uniform(di->ofOff), mlOff = uniform(di->mlOff), llOff = uniform(di->llOff), repOff = uniform(di->repOff);
        u32 r;
        r = set_seq_table(L, L.ll, &L.llLog, &L.llValid, 2, 35, 9, dictFull + llOff, repOff - llOff, 0, dLL_defaultNorm, 6, 35, lane);
        if (r == 0xFFFFFFFFu) FAIL(kErrDictionaryCorrupted);
        r = set_seq_table(L, L.of, &L.ofLog, &L.ofValid, 2, 31, 8, dictFull + ofOff, mlOff - ofOff, 1, dOF_defaultNorm, 5, 28, lane);
        if (r == 0xFFFFFFFFu) FAIL(kErrDictionaryCorrupted);
        r = set_seq_table(L, L.ml, &L.mlLog, &L.mlValid, 2, 52, 9, dictFull + mlOff, llOff - mlOff, 2, dML_defaultNorm, 6, 52, lane);
        if (r == 0xFFFFFFFFu) FAIL(kErrDictionaryCorrupted);
        rep0 = uniform(di->rep[0]); rep1 = uniform(di->rep[1]); rep2 = uniform(di->rep[2]);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    }
    for (;;) {
        if (fd.srcSize - ip < 3) FAIL(kErrSrcSizeWrong);
        const u32 bh = readLE24(fsrc + ip);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
        ip += 3;
        if (type == 3) FAIL(kErrCorruption);
        if (type == 0) {
            if (bsz > fd.srcSize - ip) FAIL(kErrSrcSizeWrong);
            if (bsz > fd.dstSize - op) FAIL(kErrCorruption);
            wave_copy(out + op, fsrc + ip, bsz, lane);
            op += bsz; ip += bsz;
        } else if (type == 1) {
            if (1 > fd.srcSize - ip) FAIL(kErrSrcSizeWrong);
            if (bsz > fd.dstSize - op) FAIL(kErrCorruption);
            const u8 b = fsrc[ip];
            for (u32 i = lane; i < bsz; i += 64) out[op + i] = b;
            op += bsz; ip += 1;
        } else {
            if (bsz > fd.srcSize - ip) FAIL(kErrSrcSizeWrong);
            if (bsz >= kBlockMax) FAIL(kErrSrcSizeWrong);
            if (bsz < 3) FAIL(kErrCorruption);
            const u8* const b = fsrc + ip; const u32 bend = bsz;
            const LitHeader lh = parse_lit_header(b, bsz);
            if (lh.err) FAIL(lh.err);
            const u32 litSize = lh.litSize;
            const u8* lit = nullptr; u32 rleByte = 0; bool litIsRle = false; u32 bp;
            if (lh.type >= 2) {
                if (litSize > fd.dstSize - litOff) FAIL(kErrCorruption);
                lit = litIn + litOff; litOff += litSize; bp = lh.lhSize + lh.litCSize;
            } else if (lh.type == 0) { lit = b + lh.lhSize; bp = lh.lhSize + litSize; }
            else { litIsRle = true; rleByte = b[lh.lhSize]; bp = lh.lhSize + 1; }
            // ---- sequences header (ZSTD_decodeSeqHeaders, :1845-1943) ----
            if (bp >= bend) FAIL(kErrSrcSizeWrong);
            u32 nbSeq = b[bp++];
            if (!nbSeq) { if (bp != bend) FAIL(kErrSrcSizeWrong); }
            else {
                if (nbSeq > 0x7F) {
                    if (nbSeq == 0xFF) { if (bp + 2 > bend) FAIL(kErrSrcSizeWrong); nbSeq = readLE16(b + bp) + 0x7F00; bp += 2; }
                    else { if (bp >= bend) FAIL(kErrSrcSizeWrong); nbSeq = ((nbSeq - 0x80) << 8) + b[bp++]; }
                }
                if (bp + 1 > bend) FAIL(kErrSrcSizeWrong);
                const u32 modes = b[bp++];
                u32 adv = 0;
                {
                    u32 p = bp, r;
                    r = set_seq_table(L, L.ll, &L.llLog, &L.llValid, modes >> 6, 35, 9, b + p, bend - p, 0, dLL_defaultNorm, 6, 35, lane);
                    if (r == 0xFFFFFFFFu) adv = r; else { p += r;
                    r = set_seq_table(L, L.of, &L.ofLog, &L.ofValid, (modes >> 4) & 3, 31, 8, b + p, bend - p, 1, dOF_defaultNorm, 5, 28, lane);
                    if (r == 0xFFFFFFFFu) adv = r; else { p += r;
                    r = set_seq_table(L, L.ml, &L.mlLog, &L.mlValid, (modes >> 2) & 3, 52, 9, b + p, bend - p, 2, dML_defaultNorm, 6, 52, lane);
                    if (r == 0xFFFFFFFFu) adv = r; else { p += r; adv = p - bp; } } }
                }
                if (adv == 0xFFFFFFFFu) FAIL(kErrCorruption);
                bp += adv;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
            }
            // ---- sequences (ZSTD_decompressSequences_body, :2668-2763), 64 at a time ----
            u32 litPos = 0;
            if (nbSeq) {
                // The state chain (ZSTD_decodeSequence's three FSE updates, :2360-2484) only needs the LENGTHS of the extra-bit
                // fields, so it runs wave-uniform on the scalar unit and records, per sequence, the three states and the bit
                // position; each lane then reads the fields of its own sequence; repcodes are resolved in order afterwards.
                SBits bd;
                if (!bd.init(b + bp, (s32)(bend - bp), lane)) FAIL(kErrCorruption);
                u32 sLL = bd.read(L.llLog, lane), sOF = bd.read(L.ofLog, lane), sML = bd.read(L.mlLog, lane);
                for (u32 base = 0; base < nbSeq; base += 64) {
                    const u32 cnt = nbSeq - base < 64 ? nbSeq - base : 64;
                    u32 recSt = 0; s32 recPos = 0;
                    ZMI_SSTAMP(0);
                    for (u32 k = 0; k < cnt; k++) {
                        // SeqSym = nextState:16 | nbAddBits:8 | nbBits:8 | baseValue:32 ; only the first dword matters here
                        const u32 vLL = *reinterpret_cast<const u32*>(&L.ll[sLL]), vML = *reinterpret_cast<const u32*>(&L.ml[sML]),
                                  vOF = *reinterpret_cast<const u32*>(&L.of[sOF]);
                        const u32 pk = sLL | (sML << 10) | (sOF << 20);
                        recSt = lane == k ? pk : recSt; recPos = lane == k ? bd.pos : recPos;
                        const u32 eLL = uniform(vLL), eML = uniform(vML), eOF = uniform(vOF);
                        const u32 nLL = eLL >> 24, nML = eML >> 24, nOF = eOF >> 24;
                        bd.pos -= (s32)(((eLL >> 16) & 0xFF) + ((eML >> 16) & 0xFF) + ((eOF >> 16) & 0xFF));   // the extra-bit fields
                        // the three state updates read LL, ML, OF bits in that order: one extraction, split afterwards
                        const u32 all = bd.read(nLL + nML + nOF, lane);
                        sLL = (vLL & 0xFFFF) + (all >> (nML + nOF));
                        sML = (vML & 0xFFFF) + ((all >> nOF) & ~(0xFFFFFFFFu << nML));
                        sOF = (vOF & 0xFFFF) + (all & ~(0xFFFFFFFFu << nOF));
                    }
                    ZMI_SSTAMP(1);
                    // ---- every lane: the fields of its own sequence ----
                    const bool have = lane < cnt;
                    u32 ll = 0, ml = 0, off = 1, code = 4;      // code 4 = a real offset; 0..3 = repcode selector
                    if (have) {
                        const SeqSym qLL = L.ll[recSt & 1023], qML = L.ml[(recSt >> 10) & 1023], qOF = L.of[recSt >> 20];
                        s32 p = recPos;
                        const u32 ofv = bd.field(p, qOF.nbAddBits); p -= qOF.nbAddBits;
                        const u32 mlv = bd.field(p, qML.nbAddBits); p -= qML.nbAddBits;
                        const u32 llv = bd.field(p, qLL.nbAddBits);
                        ll = qLL.baseValue + llv; ml = qML.baseValue + mlv;
                        if (qOF.nbAddBits > 1) off = qOF.baseValue + ofv;
                        else code = qOF.baseValue + (qLL.baseValue == 0) + ofv;     // ofv is 0 or the single extra bit
                    }
                    // ---- repcodes, in sequence order (wave-uniform) ----
                    {
                        const u64 repMask = ballot(have && code != 4);
                        if (!repMask && cnt >= 3) {
                            rep0 = read_lane(off, cnt - 1); rep1 = read_lane(off, cnt - 2); rep2 = read_lane(off, cnt - 3);
                        } else {
                            for (u32 k = 0; k < cnt; k++) {
                                const u32 cd = read_lane(code, k);
                                if (cd == 4) { const u32 o = read_lane(off, k); rep2 = rep1; rep1 = rep0; rep0 = o; }
                                else if (cd != 0) {
                                    u32 t = cd == 1 ? rep1 : (cd == 2 ? rep2 : rep0 - 1);
                                    t += !t;
                                    if (cd != 1) rep2 = rep1;
                                    rep1 = rep0; rep0 = t;
                                    off = lane == k ? t : off;
                                } else off = lane == k ? rep0 : off;
                            }
                        }
                    }
                    ZMI_SSTAMP(2);
                    const u32 inclOut = wave_scan_incl(ll + ml), inclLit = wave_scan_incl(ll);
                    const u32 totalOut = read_lane(inclOut, 63), totalLit = read_lane(inclLit, 63);
                    if (totalLit > litSize - litPos) FAIL(kErrCorruption);
                    if (totalOut > fd.dstSize - op) FAIL(kErrCorruption);
                    const u32 dLit = op + inclOut - ll - ml;            // where my literals go
                    const u32 dMatch = dLit + ll;                       // where my match goes
                    const u32 sLit = litPos + inclLit - ll;
                    if (ballot(have && (off > dMatch + dictSize || off == 0))) FAIL(kErrCorruption);
                    // literals: short runs by their own lane; long runs are cut into 16-byte pieces (the last one overlapping the
                    // one before, so every piece is whole) and ALL pieces of the batch are dealt to the lanes round-robin, four
                    // in flight per lane: the copy is paced by bandwidth, not by one load-store round trip per run
                    {
                        const bool longLit = ll > 32;
                        if (have && !longLit) {
                            if (litIsRle) for (u32 i = 0; i < ll; i++) out[dLit + i] = (u8)rleByte;
                            else lane_copy(out + dLit, lit + sLit, ll);
                        }
                        if (litIsRle) {
                            u64 lm = ballot(have && longLit);
                            while (lm) {
                                const u32 i = ctz64(lm); lm &= lm - 1;
                                const u32 d0 = read_lane(dLit, i), n0 = read_lane(ll, i);
                                for (u32 k2 = lane; k2 < n0; k2 += 64) out[d0 + k2] = (u8)rleByte;
                            }
                        } else if (ballot(have && longLit)) {
                            const u32 pc = (have && longLit) ? (ll + 15) >> 4 : 0;
                            const u32 pIncl = wave_scan_incl(pc), pExcl = pIncl - pc;
                            const u32 P = read_lane(pIncl, 63);
                            for (u32 q0 = 0; q0 < P; q0 += 256) {
                                u64 a[4], b2[4]; u32 dd[4]; bool ok[4];
#pragma unroll
                                for (u32 t = 0; t < 4; ++t) {
                                    const u32 q = q0 + t * 64 + lane;
                                    ok[t] = q < P;
                                    u32 j = 0;                       // the run that owns piece q: first lane whose inclusive count exceeds q
#pragma unroll
                                    for (u32 st = 32; st; st >>= 1) { const u32 v = __shfl(pIncl, (int)(j + st - 1)); if (v <= q) j += st; }
                                    const u32 nj = __shfl(ll, (int)j), sj = __shfl(sLit, (int)j), dj = __shfl(dLit, (int)j), ej = __shfl(pExcl, (int)j);
                                    u32 o = 16 * (q - ej); if (o + 16 > nj) o = nj - 16;
                                    a[t] = 0; b2[t] = 0; dd[t] = dj + o;
                                    if (ok[t]) { a[t] = readLE64(lit + sj + o); b2[t] = readLE64(lit + sj + o + 8); }
                                }
#pragma unroll
                                for (u32 t = 0; t < 4; ++t) if (ok[t]) { *(u64u*)(out + dd[t]) = a[t]; *(u64u*)(out + dd[t] + 8) = b2[t]; }
                            }
                        }
                    }
                    ZMI_SSTAMP(3);
                    // ---- matches, in dependency rounds ----
                    // A match may start once every earlier match of this batch whose output its source touches is complete
                    // (literals of the batch and everything before the batch already are).  Each round runs all matches that
                    // are ready: short ones by their own lane, long ones by the whole wave, then one fence.  The number of
                    // rounds is the depth of the dependency chain, not the number of dependent matches.
                    {
                        // a match that starts in the dictionary (ZSTD_execSequence's extDict branch, U/ZstdDecompressBlock.cs:2223-2250):
                        // its first bytes come from the dictionary's tail (read-only, no dependency), the rest is an ordinary
                        // match at the same distance whose source is the start of the frame
                        u32 dictN = 0;
                        if (dictSize) {                                 // uniform
                            if (have && off > dMatch) {
                                const u32 back = off - dMatch;
                                dictN = back < ml ? back : ml;
                                const u8* ds = dict + (dictSize - back);
                                for (u32 i = 0; i < dictN; i++) out[dMatch + i] = ds[i];
                            }
                        }
                        const u32 dMatchR = dMatch + dictN, mlR = ml - dictN;      // what remains for the rounds
                        const bool hasMatch = have && mlR != 0;
                        const u32 srcLo = dMatchR - off, srcHi = srcLo + (off < mlR ? off : mlR);
                        // earlier lanes whose match output overlaps my source: outputs are laid out in lane order, so they form
                        // a lane interval [jl, jh) found by two binary searches over the (monotone) per-lane bounds
                        const u64 mm = ballot(hasMatch);
                        const u32 endOutX = have ? dMatch + ml : 0xFFFFFFFFu, dMatchX = have ? dMatch : 0xFFFFFFFFu;
                        u32 jl = 0, jh = 0;
#pragma unroll
                        for (u32 st = 32; st; st >>= 1) {
                            const u32 v0 = __shfl(endOutX, (int)(jl + st - 1)), v1 = __shfl(dMatchX, (int)(jh + st - 1));
                            if (v0 <= srcLo) jl += st;
                            if (v1 < srcHi) jh += st;
                        }
                        const u64 dep = (jh > jl ? ((~0ull >> (64 - (jh - jl))) << jl) : 0ull) & mm & lanemask_lt();
                        u64 doneMask = ~mm;                            // lanes without a match never block anyone
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // output of earlier batches and this batch's literals visible
                        bool mine = hasMatch;
                        ZMI_SSTAMP(4);
                        while (doneMask != ~0ull) {
                            const bool ready = mine && (dep & ~doneMask) == 0;
                            const bool longM = mlR > 64;
                            if (ready && !longM) lane_match_copy(out + dMatchR, off, mlR);
                            u64 lm = ballot(ready && longM);
                            while (lm) {
                                const u32 i = ctz64(lm); lm &= lm - 1;
                                wave_match_copy(out + read_lane(dMatchR, i), read_lane(off, i), read_lane(mlR, i), lane);
                            }
                            const u64 r = ballot(ready);
                            doneMask |= r; mine = mine && !ready;
                            if (doneMask != ~0ull) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#ifdef ZMI_LZ_STAMPS
                            stampAcc[6] += 1;
#endif
                        }
#ifdef ZMI_LZ_STAMPS
                        stampAcc[7] += 1;
#endif
                    }
                    op += totalOut; litPos += totalLit;
                    ZMI_SSTAMP(5);
                }
                if (bd.pos > 0) FAIL(kErrCorruption);                 // bitstream not fully consumed (:2730-2733)
            }
            {
                const u32 lastLL = litSize - litPos;
                if (lastLL > fd.dstSize - op) FAIL(kErrCorruption);
                if (litIsRle) { for (u32 i = lane; i < lastLL; i += 64) out[op + i] = (u8)rleByte; }
                else wave_copy(out + op, lit + litPos, lastLL, lane);
                op += lastLL;
            }
            ip += bsz;
        }
        if (last) break;
    }
#ifdef ZMI_LZ_STAMPS
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_seqStamps[i], stampAcc[i]);
#endif
    if (!fd.unsized && op != fd.dstSize) FAIL(kErrCorruption);       // regenerated size must equal the header's FCS (U/ZstdDecompress.cs:1177-1184)
    *actualOut = op;
    if (h.checksum) {
        // XXH64 of the regenerated frame: accumulators on lanes 0..3 (U/ZstdDecompress.cs:1186-1208)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const u64 P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
        auto rotl = [](u64 x, int r) { return (x << r) | (x >> (64 - r)); };
        auto rnd = [&](u64 acc, u64 in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; };
        const u32 n = op, stripes = n >> 5, j = lane & 3;
        u64 v = j == 0 ? P1 + P2 : j == 1 ? P2 : j == 2 ? 0 : 0 - P1;
        if (lane < 4) for (u32 i = 0; i < stripes; i++) v = rnd(v, readLE64(out + 32 * i + 8 * j));
        const u64 v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        u32 bad = 0;
        if (lane == 0) {
            u64 hh;
            if (n >= 32) {
                hh = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
                auto mrg = [&](u64 acc, u64 x) { acc ^= rnd(0, x); return acc * P1 + P4; };
                hh = mrg(hh, v1); hh = mrg(hh, v2); hh = mrg(hh, v3); hh = mrg(hh, v4);
            } else hh = P5;
            hh += (u64)n;
            const u8* q = out + (stripes << 5); const u8* const end = out + n;
            while (q + 8 <= end) { hh ^= rnd(0, readLE64(q)); hh = rotl(hh, 27) * P1 + P4; q += 8; }
            if (q + 4 <= end) { hh ^= (u64)readLE32(q) * P1; hh = rotl(hh, 23) * P2 + P3; q += 4; }
            while (q < end) { hh ^= (*q) * P5; hh = rotl(hh, 11) * P1; q++; }
            hh ^= hh >> 33; hh *= P2; hh ^= hh >> 29; hh *= P3; hh ^= hh >> 32;
            if (fd.srcSize - ip < 4 || (u32)hh != readLE32(fsrc + ip)) bad = 1;
        }
        if (uniform(bad)) FAIL(kErrChecksumWrong);
    }
    return 0;
#undef FAIL
}

__global__ __launch_bounds__(64) void decode_sequences_kernel(const u8* __restrict__ src, u64 srcSize, u8* __restrict__ dst, u64 dstCapacity,
                                                              const FrameDesc* __restrict__ frames, u32 nFrames, u32* __restrict__ frameErr,
                                                              const u8* __restrict__ litScratch, u32* __restrict__ frameActual,
                                                              const u8* __restrict__ dict, u32 dictSize,
                                                              const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ SeqLds L;
    const u32 f = blockIdx.x, lane = threadIdx.x;
    if (f >= nFrames) return;
    const FrameDesc fd = frames[f];
    if (fd.srcOff + fd.srcSize > srcSize || fd.dstOff + fd.dstSize > dstCapacity) { if (lane == 0) atomicCAS(frameErr, 0u, (u32)kErrGeneric); return; }
    u32 actual = 0;
    const u32 err = decode_frame_sequences(L, fd, src + fd.srcOff, dst + fd.dstOff, litScratch + fd.dstOff + (u64)f * kLitSkew, lane, &actual, dict, dictSize, dictFull, di);
    if (err && lane == 0) atomicCAS(frameErr, 0u, err);
    if (frameActual && lane == 0) frameActual[f] = err ? 0u : actual;      // only asked for when some frame carries no content size
}

// ------------------------------------------------------------------------------------------------
// parallel frame walk
// ------------------------------------------------------------------------------------------------
constexpr u32 kSegLog = 17;
struct SegInfo { u64 entry, exit, dstBytes; u32 count, valid; };

// one step of the chain: frame (or skippable frame) at pos -> next position.  status: 0 frame, 1 skippable, 2 invalid/unsupported
__device__ inline u32 chain_step(const u8* __restrict__ src, u64 srcSize, u64 pos, u64* next, u32* content)
{
    if (srcSize - pos < 5) return 2;
    const u8* p = src + pos; const u64 avail = srcSize - pos;
    const u32 magic = readLE32(p);
    if ((magic & 0xFFFFFFF0u) == 0x184D2A50u) {
        if (avail < 8) return 2;
        const u64 sz = (u64)readLE32(p + 4) + 8;
        if (sz > avail) return 2;
        *next = pos + sz; *content = 0; return 1;
    }
    const FrameHeader h = parse_frame_header(p, avail);
    if (h.err || h.dictID || h.contentSize > 0xFFFFFFFFull) return 2;
    u64 q = pos + h.headerSize;
    for (;;) {
        if (srcSize - q < 3) return 2;
        const u32 bh = readLE24(src + q);
        const u32 last = bh & 1, type = (bh >> 1) & 3; u32 cSize = bh >> 3;
        if (type == 3) return 2;
        if (type == 1) cSize = 1;
        if (3 + (u64)cSize > srcSize - q) return 2;
        q += 3 + cSize;
        if (last) break;
    }
    if (h.checksum) { if (srcSize - q < 4) return 2; q += 4; }
    if (q - pos > 0xFFFFFFFFull) return 2;
    *next = q; *content = (u32)h.contentSize; return 0;
}

__global__ __launch_bounds__(64) void walk_segments_kernel(const u8* __restrict__ src, u64 srcSize, SegInfo* __restrict__ segs, u32 nSeg)
{
    const u32 s = blockIdx.x, lane = threadIdx.x;
    if (s >= nSeg) return;
    const u64 segStart = (u64)s << kSegLog;
    const u64 segEnd = (segStart + (1ull << kSegLog)) < srcSize ? segStart + (1ull << kSegLog) : srcSize;
    SegInfo r; r.entry = 0; r.exit = 0; r.dstBytes = 0; r.count = 0; r.valid = 0;
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
        u64 pos = cand, dstBytes = 0; u32 count = 0; bool ok = true;
        while (pos < segEnd) {
            u64 next = 0; u32 content = 0;
            const u32 st = chain_step(src, srcSize, pos, &next, &content);
            if (st == 2) { ok = false; break; }
            if (st == 0) { count++; dstBytes += content; }
            pos = next;
        }
        if (ok) { r.entry = cand; r.exit = pos; r.dstBytes = dstBytes; r.count = count; r.valid = 1; break; }
        scan = cand + 1;           // false positive (or a corrupt stream: the link check then sends us to the serial walk)
    }
    if (lane == 0) segs[s] = r;
}

// single workgroup: link check + prefix sums.  status: [0]=nFrames [1]=err [2..3]=total [4]=1 when the parallel walk is usable
__global__ __launch_bounds__(1024) void walk_link_kernel(const SegInfo* __restrict__ segs, u32 nSeg, u64 srcSize, u32 maxFrames,
                                                         u32* __restrict__ frameBase, u64* __restrict__ dstBase, u32* __restrict__ status)
{
    __shared__ u64 sh64[16]; __shared__ u32 sh32[16]; __shared__ s32 shLast[16]; __shared__ u32 bad;
    const u32 tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    if (tid == 0) bad = 0;
    __syncthreads();
    u64 carryDst = 0; u32 carryCnt = 0; s32 carryLast = -1;
    for (u32 base = 0; base < nSeg; base += 1024) {
        const u32 i = base + tid;
        SegInfo g; g.valid = 0; g.count = 0; g.dstBytes = 0; g.entry = 0; g.exit = 0;
        if (i < nSeg) g = segs[i];
        // inclusive scans inside the wave: counts, bytes, index of the last valid segment
        u32 c = g.valid ? g.count : 0; u64 b = g.valid ? g.dstBytes : 0; s32 lastv = g.valid ? (s32)i : -1;
        u32 ci = c; u64 bi = b; s32 li = lastv;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 tc = __shfl_up(ci, d); const u64 tb = __shfl_up(bi, d); const s32 tl = __shfl_up(li, d);
            if ((int)lane >= d) { ci += tc; bi += tb; li = tl > li ? tl : li; }
        }
        if (lane == 63) { sh32[wave] = ci; sh64[wave] = bi; shLast[wave] = li; }
        __syncthreads();
        u32 cb = carryCnt; u64 bb = carryDst; s32 lb = carryLast; u32 call = 0; u64 ball = 0; s32 lall = -1;
        for (u32 k = 0; k < 16; k++) {
            if (k < wave) { cb += sh32[k]; bb += sh64[k]; lb = shLast[k] > lb ? shLast[k] : lb; }
            call += sh32[k]; ball += sh64[k]; lall = shLast[k] > lall ? shLast[k] : lall;
        }
        // previous valid segment (exclusive): from the lanes before me in my wave, else from earlier waves / rounds
        s32 prevInWave = __shfl_up(li, 1); if (lane == 0) prevInWave = -1;
        const s32 prevValid = prevInWave > lb ? prevInWave : lb;
        if (i < nSeg && g.valid) {
            const u64 expect = prevValid >= 0 ? segs[prevValid].exit : 0;
            if (g.entry != expect) atomicOr(&bad, 1u);
            frameBase[i] = cb + ci - c; dstBase[i] = bb + bi - b;
        }
        carryCnt += call; carryDst += ball; carryLast = lall > carryLast ? lall : carryLast;
        __syncthreads();
    }
    if (tid == 0) {
        u32 usable = !bad;
        if (carryLast < 0) usable = 0; else if (segs[carryLast].exit != srcSize) usable = 0;
        if (carryCnt > maxFrames) usable = 0;
        status[0] = carryCnt; status[1] = 0; status[2] = (u32)carryDst; status[3] = (u32)(carryDst >> 32); status[4] = usable;
    }
}

__global__ __launch_bounds__(256) void walk_emit_kernel(const u8* __restrict__ src, u64 srcSize, const SegInfo* __restrict__ segs, u32 nSeg,
                                                        const u32* __restrict__ frameBase, const u64* __restrict__ dstBase,
                                                        const u32* __restrict__ status, FrameDesc* __restrict__ frames)
{
    const u32 s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nSeg || !status[4]) return;
    const SegInfo g = segs[s];
    if (!g.valid) return;
    u64 pos = g.entry, dstOff = dstBase[s]; u32 idx = frameBase[s];
    while (pos < g.exit) {
        u64 next = 0; u32 content = 0;
        const u32 st = chain_step(src, srcSize, pos, &next, &content);
        if (st == 2) return;                                   // cannot happen: the chain was validated by walk_segments
        if (st == 0) { FrameDesc f; f.srcOff = pos; f.dstOff = dstOff; f.srcSize = (u32)(next - pos); f.dstSize = content; f.unsized = 0; f.pad = 0; frames[idx++] = f; dstOff += content; }
        pos = next;
    }
}

size_t decode_walk_workspace_bytes(u64 srcSize)
{
    const u64 nSeg = (srcSize + (1ull << kSegLog) - 1) >> kSegLog;
    return (size_t)(nSeg * (sizeof(SegInfo) + sizeof(u32) + sizeof(u64)) + 256);
}

void launch_frame_walk(const u8* src, u64 srcSize, FrameDesc* frames, u32 maxFrames, u32* status, u8* walkWs, hipStream_t stream)
{
    const u32 nSeg = (u32)((srcSize + (1ull << kSegLog) - 1) >> kSegLog);
    SegInfo* segs = reinterpret_cast<SegInfo*>(walkWs);
    u64* dstBase = reinterpret_cast<u64*>(walkWs + (size_t)nSeg * sizeof(SegInfo));
    u32* frameBase = reinterpret_cast<u32*>(walkWs + (size_t)nSeg * (sizeof(SegInfo) + sizeof(u64)));
    hipLaunchKernelGGL(walk_segments_kernel, dim3(nSeg), dim3(64), 0, stream, src, srcSize, segs, nSeg);
    hipLaunchKernelGGL(walk_link_kernel, dim3(1), dim3(1024), 0, stream, segs, nSeg, srcSize, maxFrames, frameBase, dstBase, status);
    hipLaunchKernelGGL(walk_emit_kernel, dim3((nSeg + 255) / 256), dim3(256), 0, stream, src, srcSize, segs, nSeg, frameBase, dstBase, status, frames);
}
void launch_frame_walk_serial(const u8* src, u64 srcSize, FrameDesc* frames, u32 maxFrames, u32* status, u32 dictID, hipStream_t stream)
{
    hipLaunchKernelGGL(frame_walk_serial_kernel, dim3(1), dim3(64), 0, stream, src, srcSize, frames, maxFrames, status, dictID);
}
// =====================================================================================================================
// Literal decoder, self-synchronising form (rows a-15, a-16): one 256-thread workgroup per frame, wave w decodes Huffman
// stream w with ALL 64 lanes.  A Huffman stream has no random access, but decoding started at an arbitrary bit falls
// into step with the true codeword boundaries after a few symbols.  So the stream's bits are cut into 64 spans, one per
// lane; every lane decodes its span from a guessed start, then restarts from the position where the lane above it
// really ended, until no start changes any more (lane 0's start is exact, so this converges; two or three passes in
// practice).  The symbol counts then give every lane its output offset, and a last pass writes the symbols.  One 8 KiB
// table per frame serves 256 lookup chains instead of 4, which is what the serial form could not have (LDS capacity).
// Accepts what HUF_decompress4X1/1X1 accept (U/HufDecompress.cs:264-537): tableLog <= 12, every stream consumed exactly.
// =====================================================================================================================
constexpr u32 kStageBytes = 16384;
struct SyncLds {
    // one compressed stream per wave, staged with coalesced loads: the spans' containers are then refilled from LDS (per-lane
    // 8-byte global loads cost one cache-line request per lane per refill, which is what bounds the serial decoder too).
    u8  stage[4][kStageBytes];  // first member: 16-byte aligned
    u16 huf[4096];              // X1 table: byte | nbBits << 8.  Before it is filled its storage holds the FSE scratch.
    u8  weights[256];
    u8  sorted[256];            // symbols ordered by (weight, symbol), weight 0 excluded
    u32 classStart[14];         // first table index of weight class w; [tableLog + 1] = table size
    u32 classFirst[14];         // index into sorted[] of the first symbol of class w
    u32 meta[4];
    u32 err;
};

// One decode chain over the span (lo, p] of a stream: from a start position p (a codeword boundary or a guess) down to the
// first boundary at or below lo.  `sb` points at stream byte 0 (LDS stage + 8, or global memory when the stream did not
// fit the stage; then `size` bounds the reads).  The 4-symbol group is branch-free (a finished chain keeps looking up
// but stops advancing), so that two chains of a lane can be interleaved instruction by instruction.
template <bool STAGED>
struct SpanChain {
    s32 ptr; u32 consumed; u32 more; s32 lo; u32 cnt;
    __device__ __forceinline__ void init(s32 p, s32 lo_)
    {
        lo = lo_; cnt = 0; more = p > lo_;
        ptr = ((p + 7) >> 3) - 8;                           // container = stream bytes [ptr, ptr + 8), ptr >= -7 while the chain runs
        consumed = (u32)(8 * (ptr + 8) - p);                // bits of the container already used (0..7)
    }
    __device__ __forceinline__ s32 pos() const { return 8 * (ptr + 8) - (s32)consumed; }
    __device__ __forceinline__ u64 load8(const u8* __restrict__ sb, s32 size) const
    {
        const s32 idx = more ? ptr : 0;
        if (STAGED) {
            // LDS: three aligned dwords + v_alignbyte (an unaligned 8-byte LDS read is split into byte reads by the compiler).
            // sb is 16-byte aligned and the stage has 16 zero bytes below the stream and 8 of slack above it.
            const u32 a = (u32)(idx + 16);
            const u32* w32 = reinterpret_cast<const u32*>(sb - 16) + (a >> 2);
            const u32 d0 = w32[0], d1 = w32[1], d2 = w32[2], shb = a & 3;
            return (u64)__builtin_amdgcn_alignbyte(d1, d0, shb) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, shb) << 32);
        }
        if (idx >= 0 && idx + 8 <= size) return readLE64(sb + idx);
        u64 v = 0;
        for (s32 i = 0; i < 8; i++) { const s32 k = idx + i; if (k >= 0 && k < size) v |= (u64)sb[k] << (8 * i); }
        return v;
    }
};

constexpr u32 kChains = 2;              // independent lookup chains per lane (adjacent spans of the same stream); 4 measured slower (shorter spans, same LDS latency)

// up to 4 symbols of each of the lane's chains, interleaved instruction by instruction; packed symbols in w[], counts in k[]
template <bool STAGED, bool WRITE>
__device__ __forceinline__ void span_group(const u16* __restrict__ table, const u32 sh, const u8* __restrict__ sb, const s32 size,
                                           SpanChain<STAGED> (&C)[kChains], u32 (&w)[kChains], u32 (&k)[kChains])
{
    u64 cont[kChains]; u32 lim[kChains], con[kChains], m[kChains];
#pragma unroll
    for (u32 c = 0; c < kChains; ++c) {
        cont[c] = C[c].load8(sb, size);
        lim[c] = (u32)(8 * (C[c].ptr + 8) - C[c].lo); con[c] = C[c].consumed; m[c] = C[c].more;
        w[c] = 0; k[c] = 0;
    }
#pragma unroll
    for (u32 j = 0; j < 4; ++j) {
        u32 e[kChains];
#pragma unroll
        for (u32 c = 0; c < kChains; ++c) e[c] = table[(u32)((cont[c] << (con[c] & 63)) >> 32) >> sh];
#pragma unroll
        for (u32 c = 0; c < kChains; ++c) {
            con[c] += m[c] ? (e[c] >> 8) : 0u;
            k[c] += m[c];
            if (WRITE) w[c] |= (m[c] ? (e[c] & 0xFFu) : 0u) << (8 * j);
            m[c] = m[c] & (con[c] < lim[c]);
        }
    }
#pragma unroll
    for (u32 c = 0; c < kChains; ++c) { C[c].ptr -= (s32)(con[c] >> 3); C[c].consumed = con[c] & 7; C[c].more = m[c]; C[c].cnt += k[c]; }
}

// run all chains of a lane to the end of their spans; WRITE: symbols to out[c], 16 per store
template <bool STAGED, bool WRITE>
__device__ __forceinline__ void span_run(const u16* __restrict__ table, const u32 tableLog, const u8* __restrict__ sb, const s32 size,
                                         SpanChain<STAGED> (&C)[kChains], u8* (&out)[kChains])
{
    const u32 sh = 32 - tableLog;
    for (;;) {
        u32 any = 0;
#pragma unroll
        for (u32 c = 0; c < kChains; ++c) any |= C[c].more;
        if (!any) break;
        u32 w4[4][kChains], k4[4][kChains];
#pragma unroll
        for (u32 g = 0; g < 4; ++g) span_group<STAGED, WRITE>(table, sh, sb, size, C, w4[g], k4[g]);
        if (WRITE) {
#pragma unroll
            for (u32 c = 0; c < kChains; ++c) {
                const u32 nC = k4[0][c] + k4[1][c] + k4[2][c] + k4[3][c];
                if (nC == 16) { u32u* o = (u32u*)out[c]; o[0] = w4[0][c]; o[1] = w4[1][c]; o[2] = w4[2][c]; o[3] = w4[3][c]; }
                else { u32 t = 0; for (u32 g = 0; g < 4; ++g) for (u32 j = 0; j < 4; ++j) if (j < k4[g][c]) out[c][t++] = (u8)(w4[g][c] >> (8 * j)); }
                out[c] += nC;
            }
        }
    }
}

// one stream on one wave; true iff it decodes to exactly n symbols and is consumed to its first bit.  The stream's bits
// are cut into 64 x kChains spans, kChains ADJACENT ones per lane (independent lookup chains of a lane hide each other's
// LDS latency).  Every span first decodes a short run-in above its upper boundary to fall into step, so its first guess of
// its own start is almost always the true one; starts are then corrected from the span above until none changes.
template <bool STAGED>
__device__ __forceinline__ bool huf_stream_passes(const u16* __restrict__ table, u32 tableLog, const u8* __restrict__ sb, u32 srcSize, u32 last,
                                                  u8* __restrict__ out, u32 n, u32 lane)
{
    const s32 P0 = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);
    const s32 nSpans = 64 * kChains;
    s32 span = (P0 + nSpans - 1) / nSpans; if (span < 128) span = 128;  // >= 10 codewords per span
    const s32 kRunIn = 256;                                              // bits decoded above a span to synchronise (~40 codewords)
    s32 hi[kChains], lo[kChains];
#pragma unroll
    for (u32 c = 0; c < kChains; ++c) {
        hi[c] = P0 - (s32)(kChains * lane + c) * span;                   // upper boundary of my c-th span
        lo[c] = hi[c] - span > 0 ? hi[c] - span : 0;
    }
    SpanChain<STAGED> C[kChains];
    u8* none[kChains] = {};
#ifdef ZMI_LZ_STAMPS
    unsigned long long t0 = __builtin_amdgcn_s_memtime(); u32 nPass = 0;
#endif
    // run-in: first boundary at or below each upper boundary, reached from kRunIn bits above it (the very first span starts exactly)
#pragma unroll
    for (u32 c = 0; c < kChains; ++c) {
        const s32 g = hi[c] + kRunIn < P0 ? hi[c] + kRunIn : P0;
        if (hi[c] > 0) C[c].init(g, hi[c]); else C[c].init(hi[c], hi[c]);
    }
    span_run<STAGED, false>(table, tableLog, sb, (s32)srcSize, C, none);
    s32 start[kChains], end[kChains]; u32 cnt[kChains]; bool dirty[kChains];
#pragma unroll
    for (u32 c = 0; c < kChains; ++c) { start[c] = C[c].pos(); end[c] = start[c]; cnt[c] = 0; dirty[c] = true; }
    if (lane == 0) start[0] = P0;
    for (u32 pass = 0; pass < 64 * kChains + 2; ++pass) {
        // (a clean chain is re-initialised at its own end: nothing to do)
#pragma unroll
        for (u32 c = 0; c < kChains; ++c) C[c].init(dirty[c] ? start[c] : end[c], dirty[c] ? lo[c] : end[c]);
        span_run<STAGED, false>(table, tableLog, sb, (s32)srcSize, C, none);
#pragma unroll
        for (u32 c = 0; c < kChains; ++c) if (dirty[c]) { end[c] = C[c].pos(); cnt[c] = C[c].cnt; }
        s32 ns0 = __shfl_up(end[kChains - 1], 1);
        if (lane == 0) ns0 = P0;
        bool anyDirty = false;
#pragma unroll
        for (u32 c = 0; c < kChains; ++c) {
            const s32 ns = c == 0 ? ns0 : end[c - 1];
            dirty[c] = ns != start[c]; start[c] = ns; anyDirty = anyDirty || dirty[c];
        }
#ifdef ZMI_LZ_STAMPS
        ++nPass;
#endif
        if (!ballot(anyDirty)) break;
    }
#ifdef ZMI_LZ_STAMPS
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
#endif
    u32 cntLane = 0;
#pragma unroll
    for (u32 c = 0; c < kChains; ++c) cntLane += cnt[c];
    const u32 incl = wave_scan_incl(cntLane);
    const u32 total = read_lane(incl, 63);
    const s32 finalEnd = (s32)read_lane((u32)end[kChains - 1], 63);
    if (total != n || finalEnd != 0) return false;
    u8* outs[kChains]; u8* o = out + (incl - cntLane);
#pragma unroll
    for (u32 c = 0; c < kChains; ++c) { outs[c] = o; o += cnt[c]; C[c].init(start[c], lo[c]); }
    span_run<STAGED, true>(table, tableLog, sb, (s32)srcSize, C, outs);
#ifdef ZMI_LZ_STAMPS
    if (lane == 0) { atomicAdd(&g_seqStamps[8], (unsigned long long)nPass); atomicAdd(&g_seqStamps[9], 1ull); atomicAdd(&g_seqStamps[10], t1 - t0); atomicAdd(&g_seqStamps[11], __builtin_amdgcn_s_memtime() - t1); }
#endif
    return true;
}
__device__ __forceinline__ bool huf_decode_stream_sync(const u16* __restrict__ table, u32 tableLog, const u8* __restrict__ src, u32 srcSize,
                                       u8* __restrict__ out, u32 n, u32 lane, u8* __restrict__ stage)
{
    if (srcSize < 1) return false;
    const u32 last = uniform((u32)src[srcSize - 1]);
    if (!last) return false;
    if (srcSize + 32 <= kStageBytes) {
        // stage: 16 zero bytes, then the stream (16-byte aligned), then zeros up to the next 16-byte boundary + 16
        uint4* st4 = reinterpret_cast<uint4*>(stage);
        if (lane == 0) st4[0] = make_uint4(0, 0, 0, 0);
        const u32 pieces = (srcSize + 15) / 16 + 1;
        for (u32 i = lane; i < pieces; i += 64) {
            const u32 o = 16 * i; uint4 v = make_uint4(0, 0, 0, 0);
            if (o + 16 <= srcSize) { const u64 a = readLE64(src + o), b = readLE64(src + o + 8); v = make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32)); }
            else if (o < srcSize) {
                u32 w[4] = { 0, 0, 0, 0 };
                for (u32 k = o; k < srcSize; ++k) w[(k - o) >> 2] |= (u32)src[k] << (8 * ((k - o) & 3));
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
            st4[1 + i] = v;
        }
        wave_lds_sync();
        const bool ok = huf_stream_passes<true>(table, tableLog, stage + 16, srcSize, last, out, n, lane);
        wave_lds_sync();
        return ok;
    }
    return huf_stream_passes<false>(table, tableLog, src, srcSize, last, out, n, lane);
}

__device__ __forceinline__ u32 sync_decode_literals(SyncLds& L, const FrameDesc fd, const u8* __restrict__ fsrc, u8* __restrict__ litOut, const u32 tid,
                                                    const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    const u32 lane = tid & 63, wave = tid >> 6;
    const FrameHeader h = parse_frame_header(fsrc, fd.srcSize);
    u32 ip = h.headerSize, litOff = 0;
    bool haveTable = false; u32 tableLog = 0;
    for (;;) {
        if (fd.srcSize - ip < 3) return kErrSrcSizeWrong;
        const u32 bh = readLE24(fsrc + ip);
        const u32 last = bh & 1, type = (bh >> 1) & 3, bsz = bh >> 3;
        ip += 3;
        if (type == 3) return kErrCorruption;
        if (type == 1) { if (1 > fd.srcSize - ip) return kErrSrcSizeWrong; ip += 1; }
        else {
            if (bsz > fd.srcSize - ip) return kErrSrcSizeWrong;
            if (type == 2) {
                if (bsz >= kBlockMax) return kErrSrcSizeWrong;
                if (bsz < 3) return kErrCorruption;
                const u8* const b = fsrc + ip;
                const LitHeader lh = parse_lit_header(b, bsz);
                if (lh.err) return lh.err;
                if (lh.type >= 2) {
                    if (lh.litSize > fd.dstSize - litOff) return kErrCorruption;
                    const u8* hsrc = b + lh.lhSize; u32 hlen = lh.litCSize;
                    const bool fromDict = lh.type == 3 && !haveTable && di != nullptr;      // a treeless first block takes the dictionary's table
                    const u8* const tsrc = fromDict ? dictFull + di->hufOff : hsrc; const u32 tlen = fromDict ? di->hufSize : hlen;
                    if (lh.type == 2 || fromDict) {
                        __syncthreads();                               // the previous block's streams are done with the table
                        if (tid == 0) {
                            QuadScratch sc;
                            sc.weights = L.weights; sc.norm = reinterpret_cast<s16*>(L.huf); sc.symbolNext = L.huf + 256;
                            sc.wNewState = L.huf + 512; sc.wSymbol = reinterpret_cast<u8*>(L.huf + 576); sc.wNbBits = reinterpret_cast<u8*>(L.huf + 608);
                            u32 nbSymbols = 0, tl = 0;
                            const u32 hs = huf_read_stats(sc, tsrc, tlen, &nbSymbols, &tl);
                            L.meta[0] = hs; L.meta[1] = nbSymbols; L.meta[2] = tl; L.err = 0;
                        }
                        __syncthreads();
                        const u32 hs = L.meta[0], nbSymbols = L.meta[1]; tableLog = L.meta[2];
                        if (!hs || (fromDict ? hs > tlen : hs >= hlen)) return fromDict ? kErrDictionaryCorrupted : kErrCorruption;
                        if (tableLog > 12) return kErrTableLogTooLarge;
                        if (wave == 0) {       // HUF_readDTableX1_wksp (U/HufDecompress.cs:80-251): symbols by (weight, symbol), class extents
                            u32 wk[4], pos[4];
#pragma unroll
                            for (u32 k = 0; k < 4; ++k) { const u32 sI = k * 64 + lane; wk[k] = sI < nbSymbols ? L.weights[sI] : 0; pos[k] = 0; }
                            u32 symBase = 0, idxBase = 0;
                            for (u32 w = 1; w <= tableLog; ++w) {
                                if (lane == 0) { L.classStart[w] = idxBase; L.classFirst[w] = symBase; }
                                u32 acc = 0;
#pragma unroll
                                for (u32 k = 0; k < 4; ++k) {
                                    const u64 bm = ballot(wk[k] == w);
                                    if (wk[k] == w) pos[k] = symBase + acc + popc64(bm & lanemask_lt());
                                    acc += popc64(bm);
                                }
                                symBase += acc; idxBase += acc << (w - 1);
                            }
                            if (lane == 0) L.classStart[tableLog + 1] = idxBase;
#pragma unroll
                            for (u32 k = 0; k < 4; ++k) if (wk[k]) L.sorted[pos[k]] = (u8)(k * 64 + lane);
                        }
                        __syncthreads();
                        {
                            const u32 tableSize = 1u << tableLog;
                            if (L.classStart[tableLog + 1] != tableSize) return kErrCorruption;      // (huf_read_stats guarantees it; cheap to keep)
                            for (u32 e = tid; e < tableSize; e += 256) {
                                u32 w = 1;
                                for (u32 c = 2; c <= tableLog; ++c) if (L.classStart[c] <= e) w = c;
                                const u32 sym = L.sorted[L.classFirst[w] + ((e - L.classStart[w]) >> (w - 1))];
                                L.huf[e] = (u16)(sym | ((tableLog + 1 - w) << 8));
                            }
                        }
                        __syncthreads();
                        haveTable = true;
                        if (!fromDict) { hsrc += hs; hlen -= hs; }
                    } else if (!haveTable) return kErrDictionaryCorrupted;
                    u8* const dst = litOut + litOff;
                    bool ok = true;
                    if (lh.single) {
                        if (wave == 0) ok = huf_decode_stream_sync(L.huf, tableLog, hsrc, hlen, dst, lh.litSize, lane, L.stage[0]);
                    } else {
                        if (hlen < 10) return kErrCorruption;
                        const u32 l1 = readLE16(hsrc), l2 = readLE16(hsrc + 2), l3 = readLE16(hsrc + 4);
                        const u32 seg = (lh.litSize + 3) / 4;
                        if (6 + l1 + l2 + l3 > hlen) return kErrCorruption;
                        if (seg * 3 > lh.litSize) return kErrCorruption;
                        const u32 l4 = hlen - 6 - l1 - l2 - l3;
                        const u32 so = wave == 0 ? 6 : wave == 1 ? 6 + l1 : wave == 2 ? 6 + l1 + l2 : 6 + l1 + l2 + l3;
                        const u32 sl = wave == 0 ? l1 : wave == 1 ? l2 : wave == 2 ? l3 : l4;
                        const u32 on = wave < 3 ? seg : lh.litSize - 3 * seg;
                        ok = huf_decode_stream_sync(L.huf, tableLog, hsrc + so, sl, dst + wave * seg, on, lane, L.stage[wave]);
                    }
                    if (!ok && lane == 0) L.err = 1;
                    __syncthreads();
                    if (L.err) return kErrCorruption;
                    litOff += lh.litSize;
                }
            }
            ip += bsz;
        }
        if (last) break;
    }
    return 0;
}

__global__ __launch_bounds__(256) void decode_literals_sync_kernel(const u8* __restrict__ src, u64 srcSize, const FrameDesc* __restrict__ frames,
                                                                   u32 nFrames, u32* __restrict__ frameErr, u8* __restrict__ litScratch, u64 dstCapacity,
                                                                   const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    extern __shared__ __attribute__((aligned(16))) u8 syncLdsRaw[];
    SyncLds& L = *reinterpret_cast<SyncLds*>(syncLdsRaw);
    const u32 f = blockIdx.x, tid = threadIdx.x;
    if (f >= nFrames) return;
    const FrameDesc fd = frames[f];
    if (tid == 0) L.err = 0;
    if (fd.srcOff + fd.srcSize > srcSize || fd.dstOff + fd.dstSize > dstCapacity) { if (tid == 0) atomicCAS(frameErr, 0u, (u32)kErrGeneric); return; }
    __syncthreads();
    const u32 err = sync_decode_literals(L, fd, src + fd.srcOff, litScratch + fd.dstOff + (u64)f * kLitSkew, tid, dictFull, di);
    if (err && tid == 0) atomicCAS(frameErr, 0u, err);
}

// Three literal decoders (tools/lit_decoder_crossover.py, MI355X, 64 KiB Zipf frames):
//   serial   (4 lanes per frame, 4 KiB table):  32 frames per CU; a round of 8192 frames takes 1.1-1.7 ms;
//   compact  (4 lanes per frame, 2 KiB table + pair table): 64 frames per CU; a round of 16 384 frames takes 1.8-2.7 ms;
//   selfsync (256 lanes per frame): 0.15 ms up to 256 frames, 0.25 ms per 1000 frames beyond.
// mode: 0 = choose by frame count, 1 = serial, 2 = self-synchronising, 3 = compact.
void launch_decode_literals(const u8* src, u64 srcSize, const FrameDesc* frames, u32 nFrames, u32* frameErr, u8* litScratch, u64 dstCapacity,
                            u8* slowFlags, u32 mode, const u8* dictFull, const DictInfo* di, hipStream_t stream)
{
    if (mode == 0) {
        static int cus[64] = {};                     // per device
        int dev = 0; (void)hipGetDevice(&dev);
        if (!cus[dev & 63]) { int n = 0; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); cus[dev & 63] = n > 0 ? n : 256; }
        const u32 roundS = (u32)cus[dev & 63] * 32u, roundC = roundS * 2u;
        if (nFrames <= roundS * 3u / 4u) mode = 2;
        else {
            const float tS = (float)((nFrames + roundS - 1) / roundS) * 1.7f, tC = (float)((nFrames + roundC - 1) / roundC) * 2.7f;
            mode = tC < tS ? 3u : 1u;
        }
    }
    const bool sync = mode == 2;
    if (sync) {
        static bool attrSet[64] = {};               // per device
        int dev = 0; (void)hipGetDevice(&dev);
        if (!attrSet[dev & 63]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_literals_sync_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SyncLds)); attrSet[dev & 63] = true; }
        hipLaunchKernelGGL(decode_literals_sync_kernel, dim3(nFrames), dim3(256), sizeof(SyncLds), stream, src, srcSize, frames, nFrames, frameErr, litScratch, dstCapacity, dictFull, di);
        return;
    }
    if (mode == 3) hipLaunchKernelGGL(decode_literals_compact_kernel, dim3((nFrames + kQuads - 1) / kQuads), dim3(64), 0, stream, src, srcSize, frames, nFrames, frameErr, litScratch, dstCapacity, slowFlags, dictFull, di);
    else           hipLaunchKernelGGL(decode_literals_kernel, dim3((nFrames + kQuads - 1) / kQuads), dim3(64), 0, stream, src, srcSize, frames, nFrames, frameErr, litScratch, dstCapacity, slowFlags, dictFull, di);
    hipLaunchKernelGGL(decode_literals_slow_kernel, dim3(nFrames), dim3(64), 0, stream, src, srcSize, frames, nFrames, frameErr, litScratch, dstCapacity,
                       (const u8*)slowFlags, dictFull, di);
}
void launch_decode_sequences(const u8* src, u64 srcSize, u8* dst, u64 dstCapacity, const FrameDesc* frames, u32 nFrames, u32* frameErr,
                             const u8* litScratch, u32* frameActual, const u8* dict, u32 dictSize, const u8* dictFull, const DictInfo* di,
                             hipStream_t stream)
{
    // dict: a raw-content dictionary = history in front of EVERY frame (ZSTD_refDictContent, U/ZstdDecompress.cs:1758-1771); may be null
    hipLaunchKernelGGL(decode_sequences_kernel, dim3(nFrames), dim3(64), 0, stream, src, srcSize, dst, dstCapacity, frames, nFrames, frameErr, litScratch,
                       frameActual, dict, dict ? dictSize : 0u, dictFull, di);
}

#ifdef ZMI_LZ_STAMPS
extern "C" void ZSTDMI_debugReadSeqStamps(unsigned long long* out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_seqStamps), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seqStamps), z, sizeof z); }
}
#endif

} // namespace zmi
