// zmi_decode.h — device-side pieces shared by the decoder's kernels (decode_walk.hip, decode_lit.hip, decode_seq.hip):
// frame-header parse, bit readers, FSE_readNCount, the FSE decoding-table build, HUF_readStats, literals-section header,
// wave / lane copies.  Each follows the reference function it names; results are bit-exact with the reference decoder by
// construction of the format.  Everything here has internal linkage (one copy per translation unit).
#pragma once
#include "zmi_device.h"

namespace zmi {

// ---- error ranking: the reference stops at the first failing frame, block by block, literals before sequences.  Kernels of
// every stage run for all blocks, so each reports (block index, stage) with its code and the smallest key wins. ----
enum : u32 { kStageHeader = 0, kStageParse = 1, kStageLiterals = 2, kStageSequences = 3, kStageExec = 4, kStageFrameEnd = 5 };
__device__ __forceinline__ void report_error(u32* status, u64 blockIdx, u32 stage, u32 err)
{
    const unsigned long long key = ((unsigned long long)blockIdx << 24) | ((unsigned long long)stage << 16) | err;
    atomicMin(reinterpret_cast<unsigned long long*>(status + kStErrKeyLo), key);
}

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

struct SeqSym { u16 nextState; u8 nbAddBits; u8 nbBits; u32 baseValue; };

static __constant__ u8  dLL_bits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
static __constant__ u32 dLL_base[36] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,18,20,22,24,28,32,40,48,64,0x80,0x100,0x200,0x400,0x800,0x1000,0x2000,0x4000,0x8000,0x10000 };
static __constant__ u8  dML_bits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
static __constant__ u32 dML_base[53] = { 3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,32,33,34,
                                  35,37,39,41,43,47,51,59,67,83,99,0x83,0x103,0x203,0x403,0x803,0x1003,0x2003,0x4003,0x8003,0x10003 };
static __constant__ s16 dLL_defaultNorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
static __constant__ s16 dML_defaultNorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                         1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
static __constant__ s16 dOF_defaultNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

__device__ __forceinline__ u32 of_base(u32 code) { return code == 0 ? 0u : code == 1 ? 1u : (1u << code) - 3u; }   // OF_base, U/ZstdDecompressInternal.cs:85

// forward bit cursor (FSE_readNCount_body, U/EntropyCommon.cs:52-242); zeros beyond `size`
__device__ __forceinline__ u32 fwd_bits(const u8* p, u32 size, u32 bitpos, u32 n)
{
    u64 acc = 0; const u32 b0 = bitpos >> 3;
    if (b0 + 8 <= size) acc = readLE64(p + b0);
    else for (u32 i = 0; i < 8; i++) if (b0 + i < size) acc |= (u64)p[b0 + i] << (8 * i);
    return (u32)((acc >> (bitpos & 7)) & ((1ull << n) - 1));
}

// FSE_readNCount_body (U/EntropyCommon.cs:52-242).  Returns bytes consumed, 0 on error.  STORE = false only measures the
// description (block_parse needs to know where the next section starts, not the counts).
template <bool STORE>
__device__ inline u32 read_ncount_t(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const u8* ip, u32 srcSize)
{
    u32 bitpos, nbBits, remaining, threshold, charnum = 0; const u32 maxSV1 = *maxSVPtr + 1; bool previous0 = false;
    if (srcSize < 1) return 0;
    if (STORE) for (u32 s = 0; s < maxSV1; s++) norm[s] = 0;
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
            if (STORE) norm[charnum] = (s16)count;
            charnum++;
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

__device__ inline u32 read_ncount(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const u8* ip, u32 srcSize)
{
    return read_ncount_t<true>(norm, maxSVPtr, tableLogPtr, ip, srcSize);
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
// source and destination do not overlap.  Four pieces are loaded before the first is stored: a copy of up to 32 bytes costs one
// load latency, not one per piece (the lanes of a wave copy different runs, so this latency is what the wave waits for).
__device__ __forceinline__ void lane_copy(u8* __restrict__ d, const u8* __restrict__ s, u32 n)
{
    if (n >= 8 && n <= 16) {               // the common short run: two pieces
        const u64 a0 = readLE64(s), a1 = readLE64(s + n - 8);
        *(u64u*)d = a0; *(u64u*)(d + n - 8) = a1;
    } else if (n > 16) {
        u32 i = 0;
        for (; i + 32 < n; i += 32) {
            const u64 a0 = readLE64(s + i), a1 = readLE64(s + i + 8), a2 = readLE64(s + i + 16), a3 = readLE64(s + i + 24);
            *(u64u*)(d + i) = a0; *(u64u*)(d + i + 8) = a1; *(u64u*)(d + i + 16) = a2; *(u64u*)(d + i + 24) = a3;
        }
        // the last 1..32 bytes: pieces at i, i + 8, i + 16 (each pulled back to n - 8 where it would overrun) and n - 8
        const u32 e = n - 8;
        const u32 o0 = i < e ? i : e, o1 = i + 8 < e ? i + 8 : e, o2 = i + 16 < e ? i + 16 : e;
        const u64 a0 = readLE64(s + o0), a1 = readLE64(s + o1), a2 = readLE64(s + o2), a3 = readLE64(s + e);
        *(u64u*)(d + o0) = a0; *(u64u*)(d + o1) = a1; *(u64u*)(d + o2) = a2; *(u64u*)(d + e) = a3;
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
        // n > offset >= 8: pieces in order, each reading bytes that earlier pieces of this lane have written; with a distance of
        // 32 or more, four pieces at a time only read what earlier groups wrote
        u32 i = 0;
        if (offset >= 32) {
            for (; i + 32 <= n; i += 32) {
                const u64 a0 = readLE64(s0 + i), a1 = readLE64(s0 + i + 8), a2 = readLE64(s0 + i + 16), a3 = readLE64(s0 + i + 24);
                *(u64u*)(d + i) = a0; *(u64u*)(d + i + 8) = a1; *(u64u*)(d + i + 16) = a2; *(u64u*)(d + i + 24) = a3;
            }
        }
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
// SeqRec (zmi_common.h) <-> fields.  unpack: tag 0 = `off` is the offset; tag 1..3 = the offset is max(repIn[tag - 1] - off, 1)
__device__ __forceinline__ SeqRec rec_pack(u32 ll, u32 ml, u32 off, u32 tag)
{
    const u32 o30 = tag ? ((1u << 29) | (tag << 27) | (off & 0x7FFFFFFu)) : off;
    const u64 v = (u64)ll | ((u64)(ml - 3u) << 17) | ((u64)o30 << 34);
    SeqRec r; r.lo = (u32)v; r.hi = (u32)(v >> 32); return r;
}
__device__ __forceinline__ void rec_unpack(const SeqRec r, u32& ll, u32& ml, u32& off, u32& tag)
{
    const u64 v = (u64)r.lo | ((u64)r.hi << 32);
    ll = (u32)v & 0x1FFFFu; ml = ((u32)(v >> 17) & 0x1FFFFu) + 3u;
    const u32 o30 = r.hi >> 2;
    tag = (o30 >> 29) ? (o30 >> 27) & 3u : 0u;
    off = tag ? (o30 & 0x7FFFFFFu) : o30;
}
// one batch of up to 64 records of a block, one per lane, as the in-order consumers see it: lengths, the resolved offset, and the
// block-relative position of the sequence's literals (`outBase` = the running sum before the batch; returns the sum after it).
// `next` holds the batch's records (loaded one batch ahead by the caller).  Lanes >= cnt get ll = ml = 0.
struct SeqLane { u32 ll, ml, off, pos; };
__device__ __forceinline__ u32 seq_batch(const SeqRec r, bool have, u32 in0, u32 in1, u32 in2, u32 outBase, SeqLane& q)
{
    q.ll = 0; q.ml = 0; q.off = 1;
    if (have) {
        u32 tag; rec_unpack(r, q.ll, q.ml, q.off, tag);
        if (tag) { const u32 in = tag == 1 ? in0 : tag == 2 ? in1 : in2; q.off = in > q.off ? in - q.off : 1u; }
    }
    const u32 incl = wave_scan_incl(q.ll + q.ml);
    q.pos = outBase + incl - q.ll - q.ml;
    return outBase + read_lane(incl, 63);
}

__device__ __forceinline__ u64 uniform64(u64 v) { return (u64)uniform((u32)v) | ((u64)uniform((u32)(v >> 32)) << 32); }

} // namespace zmi
