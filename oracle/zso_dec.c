/*
 * oracle/zso_dec.c — CPU oracle: zstd frame decoder.  TEST INFRASTRUCTURE ONLY
 * (see zso_common.h).  Restates, in plain C, the decode path of the reference:
 *
 *   frame walk / header   U/ZstdDecompress.cs:427-446, 462-634, 877-993, 1062-1315
 *   block header          U/ZstdDecompressBlock.cs:19-46
 *   literals section      U/ZstdDecompressBlock.cs:88-396
 *   Huffman stats/table   U/EntropyCommon.cs:292-402, U/HufDecompress.cs:80-251
 *   Huffman 1X/4X decode  U/HufDecompress.cs:264-309, 342-537
 *   FSE NCount            U/EntropyCommon.cs:52-242
 *   FSE generic decode    U/FseDecompress.cs:25-176, 230-312
 *   sequence tables       U/ZstdDecompressBlock.cs:1571-1710, 1746-1943
 *   sequence decode/exec  U/ZstdDecompressBlock.cs:2187-2262, 2360-2484, 2668-2763
 *   bit reader            U/Bitstream.cs:172-426
 *
 * Only what changes results is restated; CPU speed tricks (X2 tables, wildcopy,
 * prefetch variants, split literal buffers) are not, as they do not change output.
 * No dictionary support (frames that name a dictID are refused).
 */
#include "zso_common.h"
#include <stdlib.h>

/* ---------- backward bit reader (U/Bitstream.cs:172-273, 377-419) ----------
 * The stream is the byte array read as one little-endian integer; the highest
 * set bit of the last byte is the end mark; symbols are read from just below
 * it, downward.  `pos` = number of unread bits.  Reading below bit 0 yields
 * zeros and drives pos negative (the reference's BIT_DStream_overflow). */
typedef struct { const u8* p; size_t size; int64_t pos; } zso_bitd;

static size_t bitd_init(zso_bitd* b, const void* src, size_t size)
{
    if (size < 1) { memset(b, 0, sizeof *b); return ZSO_ERR(srcSize_wrong); }
    b->p = (const u8*)src; b->size = size;
    {   u8 last = b->p[size - 1];
        if (last == 0) return ZSO_ERR(GENERIC);      /* end mark not present */
        b->pos = (int64_t)(size - 1) * 8 + zso_highbit32(last);
    }
    return size;
}
static inline u32 bitd_peekAt(const zso_bitd* b, int64_t bitpos, u32 n)
{   /* bits [bitpos, bitpos+n) of the stream, zeros outside */
    u64 acc = 0; int64_t byte0; int sh; int i;
    if (n == 0) return 0;
    if (bitpos < 0) {          /* partial/complete underflow: low bits are zeros */
        int64_t miss = -bitpos;
        if (miss >= (int64_t)n) return 0;
        return bitd_peekAt(b, 0, n - (u32)miss) << miss;
    }
    byte0 = bitpos >> 3; sh = (int)(bitpos & 7);
    for (i = 0; i < 8; i++) { int64_t k = byte0 + i; if (k < (int64_t)b->size) acc |= (u64)b->p[k] << (8 * i); }
    return (u32)((acc >> sh) & (((u64)1 << n) - 1));
}
static inline u32 bitd_read(zso_bitd* b, u32 n) { b->pos -= n; return bitd_peekAt(b, b->pos, n); }
/* look at the next n bits without consuming (Huffman lookup) */
static inline u32 bitd_look(const zso_bitd* b, u32 n) { return bitd_peekAt(b, b->pos - (int64_t)n, n); }

/* ---------- forward bit cursor for FSE_readNCount ---------- */
static inline u32 fwd_bits(const u8* p, size_t size, size_t bitpos, u32 n)
{
    u64 acc = 0; size_t byte0 = bitpos >> 3; int i;
    for (i = 0; i < 8; i++) { size_t k = byte0 + i; if (k < size) acc |= (u64)p[k] << (8 * i); }
    return (u32)((acc >> (bitpos & 7)) & (((u64)1 << n) - 1));
}

/* FSE_readNCount_body, U/EntropyCommon.cs:52-242 */
size_t zso_readNCount(s16* norm, u32* maxSVPtr, u32* tableLogPtr, const void* src, size_t srcSize)
{
    const u8* ip = (const u8*)src;
    size_t bitpos = 0;
    u32 nbBits, remaining, threshold, charnum = 0, maxSV1 = *maxSVPtr + 1;
    int previous0 = 0;
    if (srcSize < 1) return ZSO_ERR(srcSize_wrong);
    memset(norm, 0, (*maxSVPtr + 1) * sizeof(s16));
    nbBits = fwd_bits(ip, srcSize, 0, 4) + 5;               /* FSE_MIN_TABLELOG */
    if (nbBits > 15) return ZSO_ERR(tableLog_tooLarge);     /* FSE_TABLELOG_ABSOLUTE_MAX */
    bitpos = 4;
    *tableLogPtr = nbBits;
    remaining = (1u << nbBits) + 1;
    threshold = 1u << nbBits;
    nbBits++;
    for (;;) {
        if (previous0) {
            /* repeat flags: 2-bit groups, value 3 = "3 more zeros and continue" */
            for (;;) {
                u32 r = fwd_bits(ip, srcSize, bitpos, 2);
                bitpos += 2;
                charnum += r;
                if (r != 3) break;
                if (bitpos > srcSize * 8 + 32) return ZSO_ERR(corruption_detected);
            }
            if (charnum >= maxSV1) break;   /* reference: leaves loop, then fails on remaining != 1 or charnum > maxSV1 */
        }
        {   u32 const max = (2 * threshold - 1) - remaining;
            int count;
            u32 low = fwd_bits(ip, srcSize, bitpos, nbBits - 1);
            if (low < max) { count = (int)low; bitpos += nbBits - 1; }
            else {
                u32 v = fwd_bits(ip, srcSize, bitpos, nbBits);
                if (v >= threshold) v -= max;
                count = (int)v; bitpos += nbBits;
            }
            count--;     /* extra accuracy: -1 means "low probability" */
            if (count >= 0) remaining -= (u32)count; else remaining -= 1;
            norm[charnum++] = (s16)count;
            previous0 = !count;
            if (remaining < threshold) {
                if (remaining <= 1) break;
                nbBits = zso_highbit32(remaining) + 1;
                threshold = 1u << (nbBits - 1);
            }
            if (charnum >= maxSV1) break;
        }
    }
    if (remaining != 1) return ZSO_ERR(corruption_detected);
    if (charnum > maxSV1) return ZSO_ERR(maxSymbolValue_tooSmall);
    *maxSVPtr = charnum - 1;
    {   size_t used = (bitpos + 7) >> 3;
        if (used > srcSize) return ZSO_ERR(srcSize_wrong);
        return used;
    }
}

/* ---------- generic FSE decoding table (U/FseDecompress.cs:25-176) ---------- */
typedef struct { u16 newState; u8 symbol; u8 nbBits; } zso_fse_dentry;

static size_t zso_fse_buildDTable(zso_fse_dentry* dt, const s16* norm, u32 maxSV, u32 tableLog)
{
    u32 const tableSize = 1u << tableLog;
    u32 highThreshold = tableSize - 1;
    u16 symbolNext[256];
    u32 s;
    if (maxSV > 255) return ZSO_ERR(maxSymbolValue_tooLarge);
    if (tableLog > 12) return ZSO_ERR(tableLog_tooLarge);
    for (s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { dt[highThreshold--].symbol = (u8)s; symbolNext[s] = 1; }
        else symbolNext[s] = (u16)norm[s];
    }
    {   u32 const mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
        u32 pos = 0;
        for (s = 0; s <= maxSV; s++) {
            int i;
            for (i = 0; i < norm[s]; i++) {
                dt[pos].symbol = (u8)s;
                pos = (pos + step) & mask;
                while (pos > highThreshold) pos = (pos + step) & mask;
            }
        }
        if (pos != 0) return ZSO_ERR(GENERIC);
    }
    {   u32 u;
        for (u = 0; u < tableSize; u++) {
            u8 const sym = dt[u].symbol;
            u32 const nextState = symbolNext[sym]++;
            dt[u].nbBits = (u8)(tableLog - zso_highbit32(nextState));
            dt[u].newState = (u16)((nextState << dt[u].nbBits) - tableSize);
        }
    }
    return 0;
}

/* FSE_decompress_wksp (two interleaved states), U/FseDecompress.cs:230-312.
 * Used only for Huffman weight headers.  Stops when the bit reader overflows. */
static size_t zso_fse_decompress(u8* dst, size_t maxDst, const void* src, size_t srcSize, u32 maxLog)
{
    s16 norm[256]; u32 maxSV = 255, tableLog;
    zso_fse_dentry dt[1 << 12];
    const u8* ip = (const u8*)src;
    size_t const hs = zso_readNCount(norm, &maxSV, &tableLog, src, srcSize);
    zso_bitd bd; u32 s1, s2; size_t n = 0;
    if (zso_isError(hs)) return hs;
    if (tableLog > maxLog) return ZSO_ERR(tableLog_tooLarge);
    {   size_t e = zso_fse_buildDTable(dt, norm, maxSV, tableLog); if (zso_isError(e)) return e; }
    {   size_t e = bitd_init(&bd, ip + hs, srcSize - hs); if (zso_isError(e)) return e; }
    s1 = bitd_read(&bd, tableLog);
    s2 = bitd_read(&bd, tableLog);
    for (;;) {
        if (n + 2 > maxDst) return ZSO_ERR(dstSize_tooSmall);
        dst[n++] = dt[s1].symbol; s1 = dt[s1].newState + bitd_read(&bd, dt[s1].nbBits);
        if (bd.pos < 0) { dst[n++] = dt[s2].symbol; break; }
        if (n + 2 > maxDst) return ZSO_ERR(dstSize_tooSmall);
        dst[n++] = dt[s2].symbol; s2 = dt[s2].newState + bitd_read(&bd, dt[s2].nbBits);
        if (bd.pos < 0) { dst[n++] = dt[s1].symbol; break; }
    }
    return n;
}

/* ---------- Huffman (X1 only; X2 is a speed variant with identical output) ---------- */
typedef struct { u8 byte; u8 nbBits; } zso_huf_dentry;
typedef struct { u32 tableLog; zso_huf_dentry e[1 << 12]; int valid; } zso_huf_dtable;

/* HUF_readStats_body, U/EntropyCommon.cs:292-402.  Returns bytes consumed. */
size_t zso_huf_readStats(u8* weights, u32* nbSymbolsPtr, u32* tableLogPtr, u32* rankStats,
                                const void* src, size_t srcSize)
{
    const u8* ip = (const u8*)src;
    size_t iSize, oSize; u32 weightTotal = 0, n;
    if (!srcSize) return ZSO_ERR(srcSize_wrong);
    iSize = ip[0];
    if (iSize >= 128) {            /* raw 4-bit weights */
        oSize = iSize - 127;
        iSize = (oSize + 1) / 2;
        if (iSize + 1 > srcSize) return ZSO_ERR(srcSize_wrong);
        ip += 1;
        for (n = 0; n < oSize; n += 2) { weights[n] = ip[n / 2] >> 4; weights[n + 1] = ip[n / 2] & 15; }
    } else {                       /* FSE-compressed weights, tableLog <= 6 */
        if (iSize + 1 > srcSize) return ZSO_ERR(srcSize_wrong);
        oSize = zso_fse_decompress(weights, 255, ip + 1, iSize, 6);
        if (zso_isError(oSize)) return oSize;
    }
    memset(rankStats, 0, 13 * sizeof(u32));
    for (n = 0; n < oSize; n++) {
        if (weights[n] > 12) return ZSO_ERR(corruption_detected);
        rankStats[weights[n]]++;
        weightTotal += (1u << weights[n]) >> 1;
    }
    if (weightTotal == 0) return ZSO_ERR(corruption_detected);
    {   u32 const tableLog = zso_highbit32(weightTotal) + 1;
        u32 const total = 1u << tableLog, rest = total - weightTotal;
        u32 const verif = 1u << zso_highbit32(rest), lastWeight = zso_highbit32(rest) + 1;
        if (tableLog > 12) return ZSO_ERR(corruption_detected);
        *tableLogPtr = tableLog;
        if (verif != rest) return ZSO_ERR(corruption_detected);   /* last value must be a clean power of 2 */
        weights[oSize] = (u8)lastWeight;
        rankStats[lastWeight]++;
    }
    if (rankStats[1] < 2 || (rankStats[1] & 1)) return ZSO_ERR(corruption_detected);
    *nbSymbolsPtr = (u32)(oSize + 1);
    return iSize + 1;
}

/* HUF_readDTableX1_wksp, U/HufDecompress.cs:80-251 (same table, built plainly) */
static size_t zso_huf_readDTable(zso_huf_dtable* dt, const void* src, size_t srcSize)
{
    u8 weights[256]; u32 rankStats[13], rankStart[13], nbSymbols, tableLog, n;
    size_t const iSize = zso_huf_readStats(weights, &nbSymbols, &tableLog, rankStats, src, srcSize);
    if (zso_isError(iSize)) return iSize;
    if (tableLog > 12) return ZSO_ERR(tableLog_tooLarge);
    dt->tableLog = tableLog;
    {   u32 next = 0;
        for (n = 1; n < tableLog + 1; n++) { rankStart[n] = next; next += rankStats[n] << (n - 1); }
    }
    for (n = 0; n < nbSymbols; n++) {
        u32 const w = weights[n];
        if (w) {
            u32 const len = (1u << w) >> 1, nb = tableLog + 1 - w;
            u32 u;
            for (u = rankStart[w]; u < rankStart[w] + len; u++) { dt->e[u].byte = (u8)n; dt->e[u].nbBits = (u8)nb; }
            rankStart[w] += len;
        }
    }
    dt->valid = 1;
    return iSize;
}

/* HUF_decompress1X1_usingDTable_internal_body, U/HufDecompress.cs:264-340 */
static size_t zso_huf_decode1X(u8* dst, size_t dstSize, const void* src, size_t srcSize, const zso_huf_dtable* dt)
{
    zso_bitd bd; size_t i;
    {   size_t e = bitd_init(&bd, src, srcSize); if (zso_isError(e)) return ZSO_ERR(corruption_detected); }
    for (i = 0; i < dstSize; i++) {
        u32 const idx = bitd_look(&bd, dt->tableLog);
        dst[i] = dt->e[idx].byte;
        bd.pos -= dt->e[idx].nbBits;
    }
    if (bd.pos != 0) return ZSO_ERR(corruption_detected);    /* BIT_endOfDStream */
    return dstSize;
}

/* HUF_decompress4X1_usingDTable_internal_body, U/HufDecompress.cs:342-537 */
static size_t zso_huf_decode4X(u8* dst, size_t dstSize, const void* src, size_t srcSize, const zso_huf_dtable* dt)
{
    const u8* ip = (const u8*)src;
    if (srcSize < 10) return ZSO_ERR(corruption_detected);
    {   size_t const l1 = zso_readLE16(ip), l2 = zso_readLE16(ip + 2), l3 = zso_readLE16(ip + 4);
        size_t const seg = (dstSize + 3) / 4;
        size_t l4;
        if (6 + l1 + l2 + l3 > srcSize) return ZSO_ERR(corruption_detected);
        l4 = srcSize - 6 - l1 - l2 - l3;
        if (seg * 3 > dstSize) return ZSO_ERR(corruption_detected);
        {   size_t e;
            e = zso_huf_decode1X(dst,           seg, ip + 6,                l1, dt); if (zso_isError(e)) return e;
            e = zso_huf_decode1X(dst + seg,     seg, ip + 6 + l1,           l2, dt); if (zso_isError(e)) return e;
            e = zso_huf_decode1X(dst + 2 * seg, seg, ip + 6 + l1 + l2,      l3, dt); if (zso_isError(e)) return e;
            e = zso_huf_decode1X(dst + 3 * seg, dstSize - 3 * seg, ip + 6 + l1 + l2 + l3, l4, dt); if (zso_isError(e)) return e;
        }
    }
    return dstSize;
}

/* ---------- sequence decoding tables (U/ZstdDecompressBlock.cs:1571-1710) ---------- */
typedef struct { u16 nextState; u8 nbAddBits; u8 nbBits; u32 baseValue; } zso_seqsym;
typedef struct { u32 tableLog; zso_seqsym e[512]; } zso_seqtable;

static const u32 ZSO_OF_base[32] = { 0,1,1,5,0xD,0x1D,0x3D,0x7D,0xFD,0x1FD,0x3FD,0x7FD,0xFFD,0x1FFD,0x3FFD,0x7FFD,
    0xFFFD,0x1FFFD,0x3FFFD,0x7FFFD,0xFFFFD,0x1FFFFD,0x3FFFFD,0x7FFFFD,0xFFFFFD,0x1FFFFFD,0x3FFFFFD,0x7FFFFFD,
    0xFFFFFFD,0x1FFFFFFD,0x3FFFFFFD,0x7FFFFFFD };
static const u8 ZSO_OF_bits[32] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31 };

static void zso_buildSeqTable(zso_seqtable* t, const s16* norm, u32 maxSV, const u32* base, const u8* bits, u32 tableLog)
{
    u32 const tableSize = 1u << tableLog;
    u32 highThreshold = tableSize - 1, s;
    u16 symbolNext[64];
    t->tableLog = tableLog;
    for (s = 0; s <= maxSV; s++) {
        if (norm[s] == -1) { t->e[highThreshold--].baseValue = s; symbolNext[s] = 1; }
        else symbolNext[s] = (u16)norm[s];
    }
    {   u32 const mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
        u32 pos = 0;
        for (s = 0; s <= maxSV; s++) {
            int i;
            for (i = 0; i < norm[s]; i++) {
                t->e[pos].baseValue = s;
                pos = (pos + step) & mask;
                while (pos > highThreshold) pos = (pos + step) & mask;
            }
        }
    }
    {   u32 u;
        for (u = 0; u < tableSize; u++) {
            u32 const sym = t->e[u].baseValue;
            u32 const nextState = symbolNext[sym]++;
            t->e[u].nbBits = (u8)(tableLog - zso_highbit32(nextState));
            t->e[u].nextState = (u16)((nextState << t->e[u].nbBits) - tableSize);
            t->e[u].nbAddBits = bits[sym];
            t->e[u].baseValue = base[sym];
        }
    }
}

typedef struct {
    zso_seqtable LL, OF, ML;          /* current tables (persist across blocks of a frame for set_repeat) */
    int llValid, ofValid, mlValid;
    zso_huf_dtable huf;
    u32 rep[3];
    const u8* dict; size_t dictSize;  /* dictionary CONTENT = history in front of every frame (ZSTD_refDictContent, U/ZstdDecompress.cs:1758-1771) */
    /* formatted dictionary (ZSTD_loadDEntropy, :1773-1875): entropy tables and repcodes every frame starts from */
    int dictEntropy; u32 dictID; u32 dictRep[3];
    zso_seqtable dLL, dOF, dML; zso_huf_dtable dHuf;
} zso_frame_state;

/* ZSTD_buildSeqTable, U/ZstdDecompressBlock.cs:1746-1840.  Returns bytes consumed. */
static size_t zso_setSeqTable(zso_seqtable* t, int* valid, u32 type, u32 max, u32 maxLog,
                              const u8* src, size_t srcSize, const u32* base, const u8* bits,
                              const s16* defNorm, u32 defLog)
{
    switch (type) {
    case 1: /* set_rle */
        if (!srcSize) return ZSO_ERR(srcSize_wrong);
        if (src[0] > max) return ZSO_ERR(corruption_detected);
        t->tableLog = 0;
        t->e[0].nextState = 0; t->e[0].nbBits = 0; t->e[0].nbAddBits = bits[src[0]]; t->e[0].baseValue = base[src[0]];
        *valid = 1;
        return 1;
    case 0: /* set_basic */
        zso_buildSeqTable(t, defNorm, max, base, bits, defLog);
        *valid = 1;
        return 0;
    case 3: /* set_repeat */
        if (!*valid) return ZSO_ERR(corruption_detected);
        return 0;
    default: { /* set_compressed */
        s16 norm[64]; u32 maxSV = max, tableLog;
        size_t const hs = zso_readNCount(norm, &maxSV, &tableLog, src, srcSize);
        if (zso_isError(hs)) return ZSO_ERR(corruption_detected);
        if (tableLog > maxLog) return ZSO_ERR(corruption_detected);
        zso_buildSeqTable(t, norm, maxSV, base, bits, tableLog);
        *valid = 1;
        return hs; }
    }
}

/* ZSTD_loadDEntropy, U/ZstdDecompress.cs:1773-1875: Huffman table, then the offset, match-length and literal-length
 * NCounts, then three repcodes; returns the size of this header (content follows) or dictionary_corrupted. */
static size_t zso_loadDEntropy(zso_frame_state* fs, const u8* dict, size_t dictSize)
{
    const u8* p = dict + 8; const u8* const end = dict + dictSize;
    if (dictSize <= 8) return ZSO_ERR(dictionary_corrupted);
    {   size_t const h = zso_huf_readDTable(&fs->dHuf, p, (size_t)(end - p));
        if (zso_isError(h)) return ZSO_ERR(dictionary_corrupted);
        p += h;
    }
    {   s16 norm[64]; u32 maxSV = 31, log;
        size_t const h = zso_readNCount(norm, &maxSV, &log, p, (size_t)(end - p));
        if (zso_isError(h) || maxSV > 31 || log > 8) return ZSO_ERR(dictionary_corrupted);
        zso_buildSeqTable(&fs->dOF, norm, maxSV, ZSO_OF_base, ZSO_OF_bits, log);
        p += h;
    }
    {   s16 norm[64]; u32 maxSV = 52, log;
        size_t const h = zso_readNCount(norm, &maxSV, &log, p, (size_t)(end - p));
        if (zso_isError(h) || maxSV > 52 || log > 9) return ZSO_ERR(dictionary_corrupted);
        zso_buildSeqTable(&fs->dML, norm, maxSV, ZSO_ML_base, ZSO_ML_bits, log);
        p += h;
    }
    {   s16 norm[64]; u32 maxSV = 35, log;
        size_t const h = zso_readNCount(norm, &maxSV, &log, p, (size_t)(end - p));
        if (zso_isError(h) || maxSV > 35 || log > 9) return ZSO_ERR(dictionary_corrupted);
        zso_buildSeqTable(&fs->dLL, norm, maxSV, ZSO_LL_base, ZSO_LL_bits, log);
        p += h;
    }
    if (p + 12 > end) return ZSO_ERR(dictionary_corrupted);
    {   size_t const contentSize = (size_t)(end - (p + 12)); int i;
        for (i = 0; i < 3; i++) {
            u32 const r = zso_readLE32(p); p += 4;
            if (r == 0 || r > contentSize) return ZSO_ERR(dictionary_corrupted);
            fs->dictRep[i] = r;
        }
    }
    return (size_t)(p - dict);
}

/* ---------- block decode (U/ZstdDecompressBlock.cs:3090-3154) ---------- */
static size_t zso_decodeBlock(zso_frame_state* fs, u8* const dstStart, u8* op, u8* const oend,
                              const u8* src, size_t srcSize, u8* litBuf)
{
    const u8* ip = src; const u8* const iend = src + srcSize;
    const u8* lit; size_t litSize;
    u8* const ostart = op;
    if (srcSize >= ZSO_BLOCKSIZE_MAX) return ZSO_ERR(srcSize_wrong);
    /* --- literals section, U/ZstdDecompressBlock.cs:88-396 --- */
    if (srcSize < 3) return ZSO_ERR(corruption_detected);
    {   u32 const type = ip[0] & 3, lhl = (ip[0] >> 2) & 3;
        if (type >= 2) {                     /* compressed (2) or treeless/repeat (3) */
            size_t lhSize, litCSize; int single = 0; u32 lhc;
            if (srcSize < 5) return ZSO_ERR(corruption_detected);
            if (type == 3 && !fs->huf.valid) return ZSO_ERR(dictionary_corrupted);
            lhc = zso_readLE32(ip);
            switch (lhl) {
            case 0: case 1: default: single = !lhl; lhSize = 3; litSize = (lhc >> 4) & 0x3FF; litCSize = (lhc >> 14) & 0x3FF; break;
            case 2: lhSize = 4; litSize = (lhc >> 4) & 0x3FFF; litCSize = lhc >> 18; break;
            case 3: lhSize = 5; litSize = (lhc >> 4) & 0x3FFFF; litCSize = (lhc >> 22) + ((size_t)ip[4] << 10); break;
            }
            if (litSize > ZSO_BLOCKSIZE_MAX) return ZSO_ERR(corruption_detected);
            if (litCSize + lhSize > srcSize) return ZSO_ERR(corruption_detected);
            {   size_t const cap = (size_t)(oend - op), expected = cap < ZSO_BLOCKSIZE_MAX ? cap : ZSO_BLOCKSIZE_MAX;
                if (expected < litSize) return ZSO_ERR(dstSize_tooSmall);
            }
            {   const u8* hsrc = ip + lhSize; size_t hlen = litCSize; size_t r;
                if (type == 2) {
                    size_t const hs = zso_huf_readDTable(&fs->huf, hsrc, hlen);
                    if (zso_isError(hs)) return ZSO_ERR(corruption_detected);
                    if (hs >= hlen) return ZSO_ERR(corruption_detected);
                    hsrc += hs; hlen -= hs;
                }
                r = single ? zso_huf_decode1X(litBuf, litSize, hsrc, hlen, &fs->huf)
                           : zso_huf_decode4X(litBuf, litSize, hsrc, hlen, &fs->huf);
                if (zso_isError(r)) return ZSO_ERR(corruption_detected);
            }
            lit = litBuf; ip += lhSize + litCSize;
        } else {                             /* raw (0) or RLE (1) */
            size_t lhSize;
            switch (lhl) {
            case 0: case 2: default: lhSize = 1; litSize = ip[0] >> 3; break;
            case 1: lhSize = 2; litSize = zso_readLE16(ip) >> 4; break;
            case 3: lhSize = 3; litSize = zso_readLE24(ip) >> 4; break;
            }
            if (litSize > ZSO_BLOCKSIZE_MAX) return ZSO_ERR(corruption_detected);
            {   size_t const cap = (size_t)(oend - op), expected = cap < ZSO_BLOCKSIZE_MAX ? cap : ZSO_BLOCKSIZE_MAX;
                if (expected < litSize) return ZSO_ERR(dstSize_tooSmall);
            }
            if (type == 0) {
                if (lhSize + litSize > srcSize) return ZSO_ERR(corruption_detected);
                lit = ip + lhSize; ip += lhSize + litSize;
            } else {
                if (lhSize + 1 > srcSize) return ZSO_ERR(corruption_detected);
                memset(litBuf, ip[lhSize], litSize);
                lit = litBuf; ip += lhSize + 1;
            }
        }
    }
    /* --- sequences header, U/ZstdDecompressBlock.cs:1845-1943 --- */
    {   int nbSeq;
        if (ip >= iend) return ZSO_ERR(srcSize_wrong);
        nbSeq = *ip++;
        if (!nbSeq) {
            if (ip != iend) return ZSO_ERR(srcSize_wrong);
        } else {
            if (nbSeq > 0x7F) {
                if (nbSeq == 0xFF) { if (ip + 2 > iend) return ZSO_ERR(srcSize_wrong); nbSeq = (int)zso_readLE16(ip) + 0x7F00; ip += 2; }
                else { if (ip >= iend) return ZSO_ERR(srcSize_wrong); nbSeq = ((nbSeq - 0x80) << 8) + *ip++; }
            }
            if (ip + 1 > iend) return ZSO_ERR(srcSize_wrong);
            {   u32 const LLtype = *ip >> 6, OFtype = (*ip >> 4) & 3, MLtype = (*ip >> 2) & 3;
                size_t h;
                ip++;
                h = zso_setSeqTable(&fs->LL, &fs->llValid, LLtype, ZSO_MaxLL, ZSO_LLFSELog, ip, (size_t)(iend - ip), ZSO_LL_base, ZSO_LL_bits, ZSO_LL_defaultNorm, ZSO_LL_DEFAULTNORMLOG);
                if (zso_isError(h)) return ZSO_ERR(corruption_detected);
                ip += h;
                h = zso_setSeqTable(&fs->OF, &fs->ofValid, OFtype, ZSO_MaxOff, ZSO_OffFSELog, ip, (size_t)(iend - ip), ZSO_OF_base, ZSO_OF_bits, ZSO_OF_defaultNorm, ZSO_OF_DEFAULTNORMLOG);
                if (zso_isError(h)) return ZSO_ERR(corruption_detected);
                ip += h;
                h = zso_setSeqTable(&fs->ML, &fs->mlValid, MLtype, ZSO_MaxML, ZSO_MLFSELog, ip, (size_t)(iend - ip), ZSO_ML_base, ZSO_ML_bits, ZSO_ML_defaultNorm, ZSO_ML_DEFAULTNORMLOG);
                if (zso_isError(h)) return ZSO_ERR(corruption_detected);
                ip += h;
            }
        }
        /* --- sequences, U/ZstdDecompressBlock.cs:2668-2763 --- */
        {   const u8* litPtr = lit; const u8* const litEnd = lit + litSize;
            if (nbSeq) {
                zso_bitd bd; u32 sLL, sOF, sML; int n;
                {   size_t e = bitd_init(&bd, ip, (size_t)(iend - ip)); if (zso_isError(e)) return ZSO_ERR(corruption_detected); }
                sLL = bitd_read(&bd, fs->LL.tableLog);
                sOF = bitd_read(&bd, fs->OF.tableLog);
                sML = bitd_read(&bd, fs->ML.tableLog);
                for (n = 0; n < nbSeq; n++) {
                    zso_seqsym const ll = fs->LL.e[sLL], ml = fs->ML.e[sML], of = fs->OF.e[sOF];
                    size_t offset, matchLength = ml.baseValue, litLength = ll.baseValue;
                    /* ZSTD_decodeSequence, :2360-2484 */
                    if (of.nbAddBits > 1) {
                        offset = of.baseValue + bitd_read(&bd, of.nbAddBits);
                        fs->rep[2] = fs->rep[1]; fs->rep[1] = fs->rep[0]; fs->rep[0] = (u32)offset;
                    } else {
                        u32 const ll0 = (ll.baseValue == 0);
                        if (of.nbAddBits == 0) {
                            offset = fs->rep[ll0];
                            fs->rep[1] = fs->rep[!ll0];
                            fs->rep[0] = (u32)offset;
                        } else {
                            u32 const code = of.baseValue + ll0 + bitd_read(&bd, 1);
                            u32 temp = (code == 3) ? fs->rep[0] - 1 : fs->rep[code];
                            temp += !temp;
                            if (code != 1) fs->rep[2] = fs->rep[1];
                            fs->rep[1] = fs->rep[0];
                            fs->rep[0] = temp; offset = temp;
                        }
                    }
                    if (ml.nbAddBits) matchLength += bitd_read(&bd, ml.nbAddBits);
                    if (ll.nbAddBits) litLength += bitd_read(&bd, ll.nbAddBits);
                    sLL = ll.nextState + bitd_read(&bd, ll.nbBits);
                    sML = ml.nextState + bitd_read(&bd, ml.nbBits);
                    sOF = of.nextState + bitd_read(&bd, of.nbBits);
                    /* ZSTD_execSequence, :2187-2262 */
                    if (litLength > (size_t)(litEnd - litPtr)) return ZSO_ERR(corruption_detected);
                    if (litLength + matchLength > (size_t)(oend - op)) return ZSO_ERR(dstSize_tooSmall);
                    memcpy(op, litPtr, litLength); op += litLength; litPtr += litLength;
                    if (offset > (size_t)(op - dstStart)) {
                        /* the match starts in the dictionary (ZSTD_execSequence's extDict branch, :2223-2250): the dictionary's
                           end is virtually contiguous with the start of the frame's output */
                        size_t const back = offset - (size_t)(op - dstStart);
                        size_t n1;
                        if (back > fs->dictSize) return ZSO_ERR(corruption_detected);
                        n1 = back < matchLength ? back : matchLength;
                        memcpy(op, fs->dict + fs->dictSize - back, n1);
                        op += n1; matchLength -= n1;
                    }
                    {   const u8* m = op - offset; size_t k;
                        for (k = 0; k < matchLength; k++) op[k] = m[k];   /* byte-wise: overlap semantics */
                        op += matchLength;
                    }
                }
                if (bd.pos > 0) return ZSO_ERR(corruption_detected);      /* not fully consumed, :2730-2733 */
            }
            {   size_t const last = (size_t)(litEnd - litPtr);
                if (last > (size_t)(oend - op)) return ZSO_ERR(dstSize_tooSmall);
                memcpy(op, litPtr, last); op += last;
            }
        }
    }
    return (size_t)(op - ostart);
}

/* ---------- frame header (U/ZstdDecompress.cs:427-446, 462-634) ---------- */
typedef struct { u64 contentSize; u64 windowSize; u32 blockSizeMax; u32 headerSize; u32 dictID; u32 checksum; int skippable; } zso_fh;

static size_t zso_frameHeaderSize(const u8* src, size_t srcSize)
{
    if (srcSize < 5) return ZSO_ERR(srcSize_wrong);
    {   u8 const fhd = src[4];
        static const size_t did[4] = { 0, 1, 2, 4 }, fcs[4] = { 0, 2, 4, 8 };
        u32 const single = (fhd >> 5) & 1, fcsId = fhd >> 6;
        return 5 + !single + did[fhd & 3] + fcs[fcsId] + (single && !fcsId);
    }
}
/* returns 0 ok, >0 = wanted srcSize, or error */
static size_t zso_getFrameHeader(zso_fh* h, const u8* src, size_t srcSize)
{
    memset(h, 0, sizeof *h);
    if (srcSize < 5) return 5;
    if (zso_readLE32(src) != ZSO_MAGIC) {
        if ((zso_readLE32(src) & 0xFFFFFFF0u) == ZSO_MAGIC_SKIPPABLE) {
            if (srcSize < 8) return 8;
            h->contentSize = zso_readLE32(src + 4); h->skippable = 1;
            return 0;
        }
        return ZSO_ERR(prefix_unknown);
    }
    {   size_t const fhs = zso_frameHeaderSize(src, srcSize);
        if (srcSize < fhs) return fhs;
        h->headerSize = (u32)fhs;
    }
    {   u8 const fhd = src[4]; size_t pos = 5;
        u32 const didCode = fhd & 3, single = (fhd >> 5) & 1, fcsId = fhd >> 6;
        u64 windowSize = 0, fcsv = ZSO_CONTENTSIZE_UNKNOWN;
        h->checksum = (fhd >> 2) & 1;
        if (fhd & 0x08) return ZSO_ERR(frameParameter_unsupported);
        if (!single) {
            u8 const wl = src[pos++]; u32 const windowLog = (wl >> 3) + 10;
            if (windowLog > 31) return ZSO_ERR(frameParameter_windowTooLarge);
            windowSize = (u64)1 << windowLog;
            windowSize += (windowSize >> 3) * (wl & 7);
        }
        switch (didCode) {
        case 1: h->dictID = src[pos]; pos += 1; break;
        case 2: h->dictID = zso_readLE16(src + pos); pos += 2; break;
        case 3: h->dictID = zso_readLE32(src + pos); pos += 4; break;
        default: break;
        }
        switch (fcsId) {
        case 0: if (single) fcsv = src[pos]; break;
        case 1: fcsv = (u64)zso_readLE16(src + pos) + 256; break;
        case 2: fcsv = zso_readLE32(src + pos); break;
        default: fcsv = zso_readLE64(src + pos); break;
        }
        if (single) windowSize = fcsv;
        h->contentSize = fcsv; h->windowSize = windowSize;
        h->blockSizeMax = (u32)(windowSize < ZSO_BLOCKSIZE_MAX ? windowSize : ZSO_BLOCKSIZE_MAX);
    }
    return 0;
}

/* ZSTD_findFrameSizeInfo, U/ZstdDecompress.cs:877-951 */
static size_t zso_findFrameSizeInfo(const u8* src, size_t srcSize, u64* boundPtr)
{
    if (srcSize >= 8 && (zso_readLE32(src) & 0xFFFFFFF0u) == ZSO_MAGIC_SKIPPABLE) {
        u64 const sz = (u64)zso_readLE32(src + 4) + 8;
        if (sz > srcSize) return ZSO_ERR(srcSize_wrong);
        if (boundPtr) *boundPtr = 0;
        return (size_t)sz;
    }
    {   zso_fh h; size_t const r = zso_getFrameHeader(&h, src, srcSize);
        const u8* ip = src; size_t remaining = srcSize; size_t nbBlocks = 0;
        if (zso_isError(r)) return r;
        if (r > 0) return ZSO_ERR(srcSize_wrong);
        ip += h.headerSize; remaining -= h.headerSize;
        for (;;) {
            u32 bh; u32 last, type, cSize;
            if (remaining < 3) return ZSO_ERR(srcSize_wrong);
            bh = zso_readLE24(ip); last = bh & 1; type = (bh >> 1) & 3; cSize = bh >> 3;
            if (type == 3) return ZSO_ERR(corruption_detected);
            if (type == 1) cSize = 1;
            if (3 + (size_t)cSize > remaining) return ZSO_ERR(srcSize_wrong);
            ip += 3 + cSize; remaining -= 3 + cSize; nbBlocks++;
            if (last) break;
        }
        if (h.checksum) { if (remaining < 4) return ZSO_ERR(srcSize_wrong); ip += 4; }
        if (boundPtr) *boundPtr = (h.contentSize != ZSO_CONTENTSIZE_UNKNOWN) ? h.contentSize : (u64)nbBlocks * h.blockSizeMax;
        return (size_t)(ip - src);
    }
}

size_t zso_findFrameCompressedSize(const void* src, size_t srcSize) { return zso_findFrameSizeInfo((const u8*)src, srcSize, NULL); }

/* ZSTD_decompressBound, U/ZstdDecompress.cs:971-993 */
u64 zso_decompressBound(const void* src, size_t srcSize)
{
    const u8* ip = (const u8*)src; u64 bound = 0;
    while (srcSize > 0) {
        u64 b; size_t const cs = zso_findFrameSizeInfo(ip, srcSize, &b);
        if (zso_isError(cs)) return ZSO_CONTENTSIZE_ERROR;
        ip += cs; srcSize -= cs; bound += b;
    }
    return bound;
}

u64 zso_getFrameContentSize(const void* src, size_t srcSize)
{
    zso_fh h;
    if (zso_getFrameHeader(&h, (const u8*)src, srcSize) != 0) return ZSO_CONTENTSIZE_ERROR;
    return h.skippable ? 0 : h.contentSize;
}

/* ZSTD_decompressFrame, U/ZstdDecompress.cs:1062-1214 */
static size_t zso_decompressFrame(u8* dst, size_t dstCapacity, const u8** srcPtr, size_t* srcSizePtr, zso_frame_state* fs, u8* litBuf)
{
    const u8* ip = *srcPtr; size_t remaining = *srcSizePtr;
    u8* op = dst; u8* const oend = dst + dstCapacity;
    zso_fh h;
    if (remaining < 6 + 3) return ZSO_ERR(srcSize_wrong);    /* min frame header + block header */
    {   size_t const fhs = zso_frameHeaderSize(ip, remaining);
        size_t r;
        if (zso_isError(fhs)) return fhs;
        if (remaining < fhs + 3) return ZSO_ERR(srcSize_wrong);
        r = zso_getFrameHeader(&h, ip, fhs);
        if (zso_isError(r)) return r;
        if (r > 0) return ZSO_ERR(srcSize_wrong);
        if (h.dictID && h.dictID != fs->dictID) return ZSO_ERR(dictionary_wrong);      /* :1404-1412 */
        ip += fhs; remaining -= fhs;
    }
    /* ZSTD_decompressBegin(_usingDict), :1933-1990: fresh entropy state and repcodes per frame, or the dictionary's */
    fs->llValid = fs->ofValid = fs->mlValid = 0; fs->huf.valid = 0;
    fs->rep[0] = 1; fs->rep[1] = 4; fs->rep[2] = 8;
    if (fs->dictEntropy) {
        fs->LL = fs->dLL; fs->OF = fs->dOF; fs->ML = fs->dML; fs->huf = fs->dHuf;
        fs->llValid = fs->ofValid = fs->mlValid = 1; fs->huf.valid = 1;
        fs->rep[0] = fs->dictRep[0]; fs->rep[1] = fs->dictRep[1]; fs->rep[2] = fs->dictRep[2];
    }
    for (;;) {
        u32 bh, last, type, cSize; size_t decoded;
        if (remaining < 3) return ZSO_ERR(srcSize_wrong);
        bh = zso_readLE24(ip); last = bh & 1; type = (bh >> 1) & 3; cSize = bh >> 3;
        if (type == 3) return ZSO_ERR(corruption_detected);
        ip += 3; remaining -= 3;
        if ((type == 1 ? 1u : cSize) > remaining) return ZSO_ERR(srcSize_wrong);
        switch (type) {
        case 2:
            decoded = zso_decodeBlock(fs, dst, op, oend, ip, cSize, litBuf);
            break;
        case 0:
            if (cSize > (size_t)(oend - op)) return ZSO_ERR(dstSize_tooSmall);
            if (cSize) memcpy(op, ip, cSize);
            decoded = cSize;
            break;
        default: /* RLE: cSize is the regenerated size */
            if (cSize > (size_t)(oend - op)) return ZSO_ERR(dstSize_tooSmall);
            if (cSize) memset(op, ip[0], cSize);
            decoded = cSize; cSize = 1;
            break;
        }
        if (zso_isError(decoded)) return decoded;
        op += decoded; ip += cSize; remaining -= cSize;
        if (last) break;
    }
    if (h.contentSize != ZSO_CONTENTSIZE_UNKNOWN && (u64)(op - dst) != h.contentSize) return ZSO_ERR(corruption_detected);
    if (h.checksum) {
        if (remaining < 4) return ZSO_ERR(checksum_wrong);
        if ((u32)zso_xxh64(dst, (size_t)(op - dst), 0) != zso_readLE32(ip)) return ZSO_ERR(checksum_wrong);
        ip += 4; remaining -= 4;
    }
    *srcPtr = ip; *srcSizePtr = remaining;
    return (size_t)(op - dst);
}

/* ZSTD_decompressMultiFrame, U/ZstdDecompress.cs:1216-1315 */
size_t zso_decompress(void* dst, size_t dstCapacity, const void* src, size_t srcSize)
{
    return zso_decompress_usingDict(dst, dstCapacity, src, srcSize, NULL, 0);
}

/* ZSTD_decompress_usingDict (ZSTD_decompress_insertDictionary, U/ZstdDecompress.cs:1909-1931): without the magic the bytes
 * are raw content (ZSTD_refDictContent, :1758-1771); with it, dictID + entropy tables + repcodes precede the content. */
size_t zso_decompress_usingDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize, const void* dict, size_t dictSize)
{
    const u8* ip = (const u8*)src; u8* op = (u8*)dst;
    int moreThan1Frame = 0;
    zso_frame_state* fs = (zso_frame_state*)malloc(sizeof *fs);
    u8* litBuf = (u8*)malloc(ZSO_BLOCKSIZE_MAX + 64);
    size_t result = 0;
    if (!fs || !litBuf) { free(fs); free(litBuf); return ZSO_ERR(memory_allocation); }
    fs->dict = (const u8*)dict; fs->dictSize = dict ? dictSize : 0; fs->dictEntropy = 0; fs->dictID = 0;
    if (dict && dictSize >= 8 && zso_readLE32(dict) == 0xEC30A437u) {
        size_t const e = zso_loadDEntropy(fs, (const u8*)dict, dictSize);
        if (zso_isError(e)) { free(fs); free(litBuf); return e; }
        fs->dictID = zso_readLE32((const u8*)dict + 4);
        fs->dictEntropy = 1;
        fs->dict = (const u8*)dict + e; fs->dictSize = dictSize - e;
    }
    while (srcSize >= 5) {
        u32 const magic = zso_readLE32(ip);
        if ((magic & 0xFFFFFFF0u) == ZSO_MAGIC_SKIPPABLE) {
            size_t skip;
            if (srcSize < 8) { result = ZSO_ERR(srcSize_wrong); goto done; }
            skip = (size_t)zso_readLE32(ip + 4) + 8;
            if (skip > srcSize) { result = ZSO_ERR(srcSize_wrong); goto done; }
            ip += skip; srcSize -= skip;
            continue;
        }
        {   size_t const r = zso_decompressFrame(op, dstCapacity, &ip, &srcSize, fs, litBuf);
            if (zso_isError(r)) {
                /* reference: a bad magic after at least one good frame reads as trailing garbage, :1285-1291 */
                result = (r == ZSO_ERR(prefix_unknown) && moreThan1Frame) ? ZSO_ERR(srcSize_wrong) : r;
                goto done;
            }
            op += r; dstCapacity -= r;
            moreThan1Frame = 1;
        }
    }
    if (srcSize) { result = ZSO_ERR(srcSize_wrong); goto done; }
    result = (size_t)(op - (u8*)dst);
done:
    free(fs); free(litBuf);
    return result;
}
