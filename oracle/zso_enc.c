/*
 * oracle/zso_enc.c — CPU oracle: zstd frame encoder for the "fast" and "doubleFast" strategies (levels 1-4), greedy/lazy over the
 * row-hash match finder (level 5 and the greedy tiers of level 4) plus the strategy-independent entropy stage.  TEST INFRASTRUCTURE ONLY (see zso_common.h).
 * Restates, in plain C, the compress path of the reference:
 *
 *   level -> cParams          U/ZstdCompress.cs:7891-7927, 2023-2094 ; U/Clevels.cs:10,243,476,709
 *   frame header / epilogue   U/ZstdCompress.cs:4817-4929, 5598-5656
 *   block loop                U/ZstdCompress.cs:4690-4815, 4528-4582, 3432-3530
 *   fast match finder         U/ZstdFast.cs:96-288 ; hashes U/ZstdCompressInternal.cs:340-434
 *   doubleFast match finder   U/ZstdDoubleFast.cs:51-247
 *   greedy/lazy + row hash    U/ZstdLazy.cs:788-1309, 1743-2032
 *   seqStore / codes          U/ZstdCompressInternal.cs:20-38, 204-246 ; U/ZstdCompress.cs:3069-3098
 *   literals section          U/ZstdCompressLiterals.cs:8-185
 *   Huffman                   U/Hist.cs ; U/HufCompress.cs:40-235, 377-823, 989-1355, 1360-1543
 *   sequences section         U/ZstdCompress.cs:3127-3392 ; U/ZstdCompressSequences.cs:400-704
 *   FSE                       U/FseCompress.cs:13-360, 384-720 ; U/Fse.cs:10-57 ; U/Bitstream.cs:15-160
 *
 * The goal is byte-for-byte the reference's output for these levels (dictionary-less, single thread);
 * workspace carving, SIMD/unrolling and other CPU mechanics that do not change bytes are not restated.
 */
#include "zso_enc.h"
#include <stdlib.h>

/* ------------------------------------------------------------------ */
/*  parameters                                                         */
/* ------------------------------------------------------------------ */
/* rows 0..5 of the four size tiers, U/Clevels.cs:10 (>256K), :243 (<=256K), :476 (<=128K), :709 (<=16K) */
static const zso_cparams ZSO_levels[4][6] = {
  { {19,12,13,1,6,1,ZSO_fast}, {19,13,14,1,7,0,ZSO_fast}, {20,15,16,1,6,0,ZSO_fast},
    {21,16,17,1,5,0,ZSO_dfast}, {21,18,18,1,5,0,ZSO_dfast}, {21,18,19,3,5,2,ZSO_greedy} },
  { {18,12,13,1,5,1,ZSO_fast}, {18,13,14,1,6,0,ZSO_fast}, {18,14,14,1,5,0,ZSO_dfast},
    {18,16,16,1,4,0,ZSO_dfast}, {18,16,17,3,5,2,ZSO_greedy}, {18,17,18,5,5,2,ZSO_greedy} },
  { {17,12,12,1,5,1,ZSO_fast}, {17,12,13,1,6,0,ZSO_fast}, {17,13,15,1,5,0,ZSO_fast},
    {17,15,16,2,5,0,ZSO_dfast}, {17,17,17,2,4,0,ZSO_dfast}, {17,16,17,3,4,2,ZSO_greedy} },
  { {14,12,13,1,5,1,ZSO_fast}, {14,14,15,1,5,0,ZSO_fast}, {14,14,15,1,4,0,ZSO_fast},
    {14,14,15,2,4,0,ZSO_dfast}, {14,14,14,4,4,2,ZSO_greedy}, {14,14,14,3,4,4,ZSO_lazy} },
};

zso_cparams zso_getCParams(int level, u64 srcSize)
{
    u32 const tableID = (srcSize <= 256 * 1024) + (srcSize <= 128 * 1024) + (srcSize <= 16 * 1024);
    int row = level == 0 ? 3 : level < 0 ? 0 : level > 5 ? 5 : level;   /* oracle covers rows 0..5 only */
    zso_cparams cp = ZSO_levels[tableID][row];
    if (level < 0) cp.targetLength = (u32)(-level);
    /* ZSTD_adjustCParams_internal, no dictionary */
    if (srcSize < ((u64)1 << 30)) {
        u32 const tSize = (u32)srcSize;
        u32 const srcLog = (tSize < 64) ? 6 : zso_highbit32(tSize - 1) + 1;
        if (cp.windowLog > srcLog) cp.windowLog = srcLog;
    }
    {   u32 const cycleLog = cp.chainLog;   /* ZSTD_cycleLog: chainLog - (strategy >= btlazy2) */
        if (cp.hashLog > cp.windowLog + 1) cp.hashLog = cp.windowLog + 1;
        if (cycleLog > cp.windowLog) cp.chainLog -= (cycleLog - cp.windowLog);
    }
    if (cp.windowLog < 10) cp.windowLog = 10;
    return cp;
}

/* ZSTD_compressBound, U/ZstdCompress.cs:19-22 */
size_t zso_compressBound(size_t n) { return n + (n >> 8) + (n < (128 << 10) ? ((128 << 10) - n) >> 11 : 0); }

/* ------------------------------------------------------------------ */
/*  forward bit writer  (BIT_CStream_t)                                */
/* ------------------------------------------------------------------ */
typedef struct { u8* start; u8* ptr; u8* end; u64 acc; u32 nbits; int overflow; } zso_bitc;

static size_t bitc_init(zso_bitc* b, void* dst, size_t cap)
{
    if (cap <= 8) return ZSO_ERR(dstSize_tooSmall);
    b->start = b->ptr = (u8*)dst; b->end = b->start + cap - 8;   /* the reference keeps one container of slack */
    b->acc = 0; b->nbits = 0; b->overflow = 0;
    return 0;
}
static inline void bitc_add(zso_bitc* b, u64 value, u32 n)
{
    if (n) value &= (((u64)1 << n) - 1); else value = 0;
    b->acc |= value << b->nbits; b->nbits += n;
    while (b->nbits >= 8) {
        if (b->ptr < b->end) *b->ptr++ = (u8)b->acc; else b->overflow = 1;
        b->acc >>= 8; b->nbits -= 8;
    }
}
static size_t bitc_close(zso_bitc* b)
{
    bitc_add(b, 1, 1);                                  /* end mark */
    if (b->overflow || b->ptr >= b->end) return 0;
    if (b->nbits) { *b->ptr = (u8)b->acc; return (size_t)(b->ptr - b->start) + 1; }
    return (size_t)(b->ptr - b->start);
}

/* ------------------------------------------------------------------ */
/*  histogram  (U/Hist.cs)                                             */
/* ------------------------------------------------------------------ */
static u32 zso_hist(u32* count, u32* maxSVPtr, const u8* src, size_t n)
{
    u32 maxSV = *maxSVPtr, s, largest = 0; size_t i;
    memset(count, 0, (maxSV + 1) * sizeof(u32));
    if (!n) { *maxSVPtr = 0; return 0; }
    for (i = 0; i < n; i++) count[src[i]]++;
    while (!count[maxSV]) maxSV--;
    *maxSVPtr = maxSV;
    for (s = 0; s <= maxSV; s++) if (count[s] > largest) largest = count[s];
    return largest;
}

/* ------------------------------------------------------------------ */
/*  FSE compression tables                                             */
/* ------------------------------------------------------------------ */
typedef struct { int deltaFindState; u32 deltaNbBits; } zso_symTT;
typedef struct { u32 tableLog; u32 maxSV; u16 stateTable[512]; zso_symTT tt[256]; } zso_fse_ct;

static u32 zso_fse_minTableLog(size_t srcSize, u32 maxSV)
{
    u32 const a = zso_highbit32((u32)srcSize) + 1, b = zso_highbit32(maxSV) + 2;
    return a < b ? a : b;
}
/* FSE_optimalTableLog_internal, U/FseCompress.cs:397-430 */
static u32 zso_fse_optimalTableLog(u32 maxTableLog, size_t srcSize, u32 maxSV, u32 minus)
{
    u32 const maxBitsSrc = zso_highbit32((u32)(srcSize - 1)) - minus;
    u32 tableLog = maxTableLog, minBits = zso_fse_minTableLog(srcSize, maxSV);
    if (tableLog == 0) tableLog = 11;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}

/* FSE_normalizeM2, U/FseCompress.cs:443-561 */
static size_t zso_fse_normalizeM2(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSV, s16 lowProbCount)
{
    s16 const NOT_YET = -2;
    u32 s, distributed = 0, ToDistribute;
    u32 const lowThreshold = (u32)(total >> tableLog);
    u32 lowOne = (u32)((total * 3) >> (tableLog + 1));
    for (s = 0; s <= maxSV; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NOT_YET;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0) return 0;
    if ((total / ToDistribute) > lowOne) {
        lowOne = (u32)((total * 3) / (ToDistribute * 2));
        for (s = 0; s <= maxSV; s++)
            if (norm[s] == NOT_YET && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        ToDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSV + 1) {
        u32 maxV = 0, maxC = 0;
        for (s = 0; s <= maxSV; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (s16)ToDistribute;
        return 0;
    }
    if (total == 0) {
        for (s = 0; ToDistribute > 0; s = (s + 1) % (maxSV + 1)) if (norm[s] > 0) { ToDistribute--; norm[s]++; }
        return 0;
    }
    {   u64 const vStepLog = 62 - tableLog, mid = ((u64)1 << (vStepLog - 1)) - 1;
        u64 const rStep = ((((u64)1 << vStepLog) * ToDistribute) + mid) / (u32)total;
        u64 tmpTotal = mid;
        for (s = 0; s <= maxSV; s++) {
            if (norm[s] == NOT_YET) {
                u64 const end = tmpTotal + (count[s] * rStep);
                u32 const sStart = (u32)(tmpTotal >> vStepLog), sEnd = (u32)(end >> vStepLog), weight = sEnd - sStart;
                if (weight < 1) return ZSO_ERR(GENERIC);
                norm[s] = (s16)weight; tmpTotal = end;
            }
        }
    }
    return 0;
}

/* FSE_normalizeCount, U/FseCompress.cs:574-665 ; rtbTable U/Arrays.cs:8 */
static size_t zso_fse_normalizeCount(s16* norm, u32 tableLog, const u32* count, size_t total, u32 maxSV, u32 useLowProbCount)
{
    static const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    if (tableLog == 0) tableLog = 11;
    if (tableLog < 5) return ZSO_ERR(GENERIC);
    if (tableLog > 12) return ZSO_ERR(tableLog_tooLarge);
    if (tableLog < zso_fse_minTableLog(total, maxSV)) return ZSO_ERR(GENERIC);
    {   s16 const lowProbCount = useLowProbCount ? -1 : 1;
        u64 const scale = 62 - tableLog, step = ((u64)1 << 62) / (u32)total, vStep = (u64)1 << (scale - 20);
        int stillToDistribute = 1 << tableLog;
        u32 s, largest = 0; s16 largestP = 0;
        u32 const lowThreshold = (u32)(total >> tableLog);
        for (s = 0; s <= maxSV; s++) {
            if (count[s] == total) return 0;                       /* rle special case */
            if (count[s] == 0) { norm[s] = 0; continue; }
            if (count[s] <= lowThreshold) { norm[s] = lowProbCount; stillToDistribute--; }
            else {
                s16 proba = (s16)((count[s] * step) >> scale);
                if (proba < 8) {
                    u64 const restToBeat = vStep * rtb[proba];
                    proba += (count[s] * step) - ((u64)proba << scale) > restToBeat;
                }
                if (proba > largestP) { largestP = proba; largest = s; }
                norm[s] = proba; stillToDistribute -= proba;
            }
        }
        if (-stillToDistribute >= (norm[largest] >> 1)) {
            size_t const e = zso_fse_normalizeM2(norm, tableLog, count, total, maxSV, lowProbCount);
            if (zso_isError(e)) return e;
        } else norm[largest] += (s16)stillToDistribute;
    }
    return tableLog;
}

/* FSE_writeNCount_generic, U/FseCompress.cs:203-336 (capacity handling reduced to a bound check) */
static size_t zso_fse_writeNCount(void* header, size_t cap, const s16* norm, u32 maxSV, u32 tableLog)
{
    u8 tmp[512]; u8* out = tmp;
    int nbBits = (int)tableLog + 1, remaining = (1 << tableLog) + 1, threshold = 1 << tableLog;
    u32 bitStream = 0; int bitCount = 0; u32 symbol = 0; u32 const alphabetSize = maxSV + 1; int previousIs0 = 0;
    if (tableLog > 12) return ZSO_ERR(tableLog_tooLarge);
    if (tableLog < 5) return ZSO_ERR(GENERIC);
    bitStream += (tableLog - 5) << bitCount; bitCount += 4;
    while (symbol < alphabetSize && remaining > 1) {
        if (previousIs0) {
            u32 start = symbol;
            while (symbol < alphabetSize && !norm[symbol]) symbol++;
            if (symbol == alphabetSize) break;
            while (symbol >= start + 24) {
                start += 24; bitStream += 0xFFFFu << bitCount;
                out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16;
            }
            while (symbol >= start + 3) { start += 3; bitStream += 3u << bitCount; bitCount += 2; }
            bitStream += (symbol - start) << bitCount; bitCount += 2;
            if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
        }
        {   int count = norm[symbol++];
            int const max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (u32)count << bitCount;
            bitCount += nbBits; bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return ZSO_ERR(GENERIC);
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return ZSO_ERR(GENERIC);
    out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += (bitCount + 7) / 8;
    {   size_t const n = (size_t)(out - tmp);
        if (n > cap) return ZSO_ERR(dstSize_tooSmall);
        memcpy(header, tmp, n);
        return n;
    }
}

/* FSE_buildCTable_wksp, U/FseCompress.cs:13-191 */
static size_t zso_fse_buildCTable(zso_fse_ct* ct, const s16* norm, u32 maxSV, u32 tableLog)
{
    u32 const tableSize = 1u << tableLog, tableMask = tableSize - 1;
    u32 const step = (tableSize >> 1) + (tableSize >> 3) + 3;
    u16 cumul[258]; u8 tableSymbol[4096];
    u32 highThreshold = tableSize - 1, u;
    if (tableLog > 9 && maxSV > 52) { if (tableLog > 12) return ZSO_ERR(tableLog_tooLarge); }
    ct->tableLog = tableLog; ct->maxSV = maxSV;
    cumul[0] = 0;
    for (u = 1; u <= maxSV + 1; u++) {
        if (norm[u - 1] == -1) { cumul[u] = cumul[u - 1] + 1; tableSymbol[highThreshold--] = (u8)(u - 1); }
        else cumul[u] = cumul[u - 1] + (u16)norm[u - 1];
    }
    cumul[maxSV + 1] = (u16)(tableSize + 1);
    {   u32 position = 0, symbol;     /* the reference's "no low-prob" fast path lays out the same table */
        for (symbol = 0; symbol <= maxSV; symbol++) {
            int n;
            for (n = 0; n < norm[symbol]; n++) {
                tableSymbol[position] = (u8)symbol;
                position = (position + step) & tableMask;
                while (position > highThreshold) position = (position + step) & tableMask;
            }
        }
        if (position != 0) return ZSO_ERR(GENERIC);
    }
    for (u = 0; u < tableSize; u++) { u8 const s = tableSymbol[u]; ct->stateTable[cumul[s]++] = (u16)(tableSize + u); }
    {   u32 total = 0, s;
        for (s = 0; s <= maxSV; s++) {
            switch (norm[s]) {
            case 0: ct->tt[s].deltaNbBits = ((tableLog + 1) << 16) - (1u << tableLog); ct->tt[s].deltaFindState = 0; break;
            case -1: case 1:
                ct->tt[s].deltaNbBits = (tableLog << 16) - (1u << tableLog);
                ct->tt[s].deltaFindState = (int)(total - 1); total++; break;
            default: {
                u32 const maxBitsOut = tableLog - zso_highbit32((u32)norm[s] - 1);
                u32 const minStatePlus = (u32)norm[s] << maxBitsOut;
                ct->tt[s].deltaNbBits = (maxBitsOut << 16) - minStatePlus;
                ct->tt[s].deltaFindState = (int)(total - (u32)norm[s]);
                total += (u32)norm[s]; }
            }
        }
    }
    return 0;
}
/* FSE_buildCTable_rle, U/FseCompress.cs:706-720 */
static void zso_fse_buildCTable_rle(zso_fse_ct* ct, u8 symbol)
{
    ct->tableLog = 0; ct->maxSV = symbol; ct->stateTable[0] = 0; ct->stateTable[1] = 0;
    ct->tt[symbol].deltaNbBits = 0; ct->tt[symbol].deltaFindState = 0;
}

/* FSE_CState_t ops, U/Fse.cs:10-57 */
typedef struct { ptrdiff_t value; const zso_fse_ct* ct; } zso_cstate;
static void cstate_init2(zso_cstate* st, const zso_fse_ct* ct, u32 symbol)
{
    zso_symTT const tt = ct->tt[symbol];
    u32 const nbBitsOut = (tt.deltaNbBits + (1 << 15)) >> 16;
    st->ct = ct;
    st->value = (ptrdiff_t)((nbBitsOut << 16) - tt.deltaNbBits);
    st->value = ct->stateTable[(st->value >> nbBitsOut) + tt.deltaFindState];
}
static inline void cstate_encode(zso_bitc* b, zso_cstate* st, u32 symbol)
{
    zso_symTT const tt = st->ct->tt[symbol];
    u32 const nbBitsOut = (u32)(((size_t)st->value + tt.deltaNbBits) >> 16);
    bitc_add(b, (u64)st->value, nbBitsOut);
    st->value = st->ct->stateTable[(st->value >> nbBitsOut) + tt.deltaFindState];
}
static inline void cstate_flush(zso_bitc* b, const zso_cstate* st) { bitc_add(b, (u64)st->value, st->ct->tableLog); }

/* FSE_compress_usingCTable_generic, U/FseCompress.cs:722-820: symbol i goes to state (i & 1), last symbol first */
static size_t zso_fse_compress(void* dst, size_t cap, const u8* src, size_t n, const zso_fse_ct* ct)
{
    zso_bitc b; zso_cstate st[2]; size_t i;
    if (n <= 2) return 0;
    if (zso_isError(bitc_init(&b, dst, cap))) return 0;
    cstate_init2(&st[(n - 1) & 1], ct, src[n - 1]);
    cstate_init2(&st[(n - 2) & 1], ct, src[n - 2]);
    for (i = n - 2; i-- > 0; ) cstate_encode(&b, &st[i & 1], src[i]);
    cstate_flush(&b, &st[1]);       /* CState2 first, then CState1 */
    cstate_flush(&b, &st[0]);
    return bitc_close(&b);
}

/* ------------------------------------------------------------------ */
/*  Huffman                                                            */
/* ------------------------------------------------------------------ */
typedef struct { u32 count; u16 parent; u8 byte; u8 nbBits; } zso_node;
typedef struct { u8 nbBits[256]; u16 val[256]; u32 tableLog; u32 maxSV; int valid; } zso_huf_ct;   /* valid = a table exists */
enum { HUF_repeat_none = 0, HUF_repeat_check = 1, HUF_repeat_valid = 2 };

/* HUF_sort helpers, U/HufCompress.cs:520-680 */
static u32 huf_getIndex(u32 count) { return count < 165 ? count : zso_highbit32(count) + 158; }
static void huf_insertionSort(zso_node* a, int low, int high)
{
    int i, size = high - low + 1; a += low;
    for (i = 1; i < size; i++) {
        zso_node const key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
static int huf_partition(zso_node* a, int low, int high)
{
    u32 const pivot = a[high].count; int i = low - 1, j;
    for (j = low; j < high; j++) if (a[j].count > pivot) { zso_node t; i++; t = a[i]; a[i] = a[j]; a[j] = t; }
    { zso_node t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
    return i + 1;
}
static void huf_quickSort(zso_node* a, int low, int high)
{
    if (high - low < 8) { huf_insertionSort(a, low, high); return; }
    while (low < high) {
        int const idx = huf_partition(a, low, high);
        if (idx - low < high - idx) { huf_quickSort(a, low, idx - 1); low = idx + 1; }
        else { huf_quickSort(a, idx + 1, high); high = idx - 1; }
    }
}
static void huf_sort(zso_node* huffNode, const u32* count, u32 maxSV)
{
    struct { u16 base, curr; } rp[192];
    u32 n; u32 const maxSV1 = maxSV + 1;
    memset(rp, 0, sizeof rp);
    for (n = 0; n < maxSV1; n++) rp[huf_getIndex(count[n])].base++;
    for (n = 191; n > 0; n--) { rp[n - 1].base += rp[n].base; rp[n - 1].curr = rp[n - 1].base; }
    for (n = 0; n < maxSV1; n++) {
        u32 const c = count[n], r = huf_getIndex(c) + 1, pos = rp[r].curr++;
        huffNode[pos].count = c; huffNode[pos].byte = (u8)n;
    }
    for (n = 165; n < 191; n++) {
        u32 const bucketSize = rp[n].curr - rp[n].base, bucketStart = rp[n].base;
        if (bucketSize > 1) huf_quickSort(huffNode + bucketStart, 0, (int)bucketSize - 1);
    }
}
/* HUF_buildTree, U/HufCompress.cs:689-738 */
static int huf_buildTree(zso_node* huffNode, u32 maxSV)
{
    zso_node* const huffNode0 = huffNode - 1;
    int nonNullRank = (int)maxSV, lowS, lowN, nodeNb = 256, n, nodeRoot;
    while (huffNode[nonNullRank].count == 0) nonNullRank--;
    lowS = nonNullRank; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (n = nodeNb; n <= nodeRoot; n++) huffNode[n].count = 1u << 30;
    huffNode0[0].count = 1u << 31;     /* sentinel at huffNode[-1] */
    while (nodeNb <= nodeRoot) {
        int const n1 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        int const n2 = (huffNode[lowS].count < huffNode[lowN].count) ? lowS-- : lowN++;
        huffNode[nodeNb].count = huffNode[n1].count + huffNode[n2].count;
        huffNode[n1].parent = huffNode[n2].parent = (u16)nodeNb;
        nodeNb++;
    }
    huffNode[nodeRoot].nbBits = 0;
    for (n = nodeRoot - 1; n >= 256; n--) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    for (n = 0; n <= nonNullRank; n++) huffNode[n].nbBits = huffNode[huffNode[n].parent].nbBits + 1;
    return nonNullRank;
}
/* HUF_setMaxHeight, U/HufCompress.cs:377-514 */
static u32 huf_setMaxHeight(zso_node* huffNode, u32 lastNonNull, u32 maxNbBits)
{
    u32 const largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= maxNbBits) return largestBits;
    {   int totalCost = 0; u32 const baseCost = 1u << (largestBits - maxNbBits); int n = (int)lastNonNull;
        while (huffNode[n].nbBits > maxNbBits) {
            totalCost += (int)(baseCost - (1u << (largestBits - huffNode[n].nbBits)));
            huffNode[n].nbBits = (u8)maxNbBits; n--;
        }
        while (huffNode[n].nbBits == maxNbBits) --n;
        totalCost >>= (largestBits - maxNbBits);
        {   u32 const noSymbol = 0xF0F0F0F0; u32 rankLast[14]; int pos; u32 currentNbBits = maxNbBits;
            for (pos = 0; pos < 14; pos++) rankLast[pos] = noSymbol;
            for (pos = n; pos >= 0; pos--) {
                if (huffNode[pos].nbBits >= currentNbBits) continue;
                currentNbBits = huffNode[pos].nbBits;
                rankLast[maxNbBits - currentNbBits] = (u32)pos;
            }
            while (totalCost > 0) {
                u32 nBitsToDecrease = zso_highbit32((u32)totalCost) + 1;
                for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                    u32 const highPos = rankLast[nBitsToDecrease], lowPos = rankLast[nBitsToDecrease - 1];
                    if (highPos == noSymbol) continue;
                    if (lowPos == noSymbol) break;
                    {   u32 const highTotal = huffNode[highPos].count, lowTotal = 2 * huffNode[lowPos].count;
                        if (highTotal <= lowTotal) break;
                    }
                }
                while (nBitsToDecrease <= 12 && rankLast[nBitsToDecrease] == noSymbol) nBitsToDecrease++;
                totalCost -= 1 << (nBitsToDecrease - 1);
                huffNode[rankLast[nBitsToDecrease]].nbBits++;
                if (rankLast[nBitsToDecrease - 1] == noSymbol) rankLast[nBitsToDecrease - 1] = rankLast[nBitsToDecrease];
                if (rankLast[nBitsToDecrease] == 0) rankLast[nBitsToDecrease] = noSymbol;
                else {
                    rankLast[nBitsToDecrease]--;
                    if (huffNode[rankLast[nBitsToDecrease]].nbBits != maxNbBits - nBitsToDecrease) rankLast[nBitsToDecrease] = noSymbol;
                }
            }
            while (totalCost < 0) {
                if (rankLast[1] == noSymbol) {
                    while (huffNode[n].nbBits == maxNbBits) n--;
                    huffNode[n + 1].nbBits--;
                    rankLast[1] = (u32)(n + 1);
                    totalCost++;
                    continue;
                }
                huffNode[rankLast[1] + 1].nbBits--;
                rankLast[1]++;
                totalCost++;
            }
        }
    }
    return maxNbBits;
}
/* HUF_buildCTable_wksp + HUF_buildCTableFromTree, U/HufCompress.cs:750-823 */
static size_t huf_buildCTable(zso_huf_ct* ct, const u32* count, u32 maxSV, u32 maxNbBits)
{
    zso_node nodes[513]; zso_node* const huffNode = nodes + 1;
    int nonNullRank, n; u16 nbPerRank[13] = {0}, valPerRank[13] = {0};
    if (maxNbBits == 0) maxNbBits = 11;
    if (maxSV > 255) return ZSO_ERR(maxSymbolValue_tooLarge);
    memset(nodes, 0, sizeof nodes);
    huf_sort(huffNode, count, maxSV);
    nonNullRank = huf_buildTree(huffNode, maxSV);
    maxNbBits = huf_setMaxHeight(huffNode, (u32)nonNullRank, maxNbBits);
    if (maxNbBits > 12) return ZSO_ERR(GENERIC);
    for (n = 0; n <= nonNullRank; n++) nbPerRank[huffNode[n].nbBits]++;
    {   u16 min = 0;
        for (n = (int)maxNbBits; n > 0; n--) { valPerRank[n] = min; min += nbPerRank[n]; min >>= 1; }
    }
    memset(ct->nbBits, 0, sizeof ct->nbBits); memset(ct->val, 0, sizeof ct->val);
    for (n = 0; n <= (int)maxSV; n++) ct->nbBits[huffNode[n].byte] = huffNode[n].nbBits;
    for (n = 0; n <= (int)maxSV; n++) ct->val[n] = valPerRank[ct->nbBits[n]]++;
    ct->tableLog = maxNbBits; ct->maxSV = maxSV; ct->valid = 1;
    return maxNbBits;
}
size_t zso_huf_buildLengths(u8* nbBits, const u32* count, u32 maxSV, u32 maxNbBits)
{
    zso_huf_ct ct; size_t const r = huf_buildCTable(&ct, count, maxSV, maxNbBits);
    if (!zso_isError(r)) memcpy(nbBits, ct.nbBits, maxSV + 1);
    return r;
}

/* HUF_compressWeights, U/HufCompress.cs:40-125 */
static size_t huf_compressWeights(void* dst, size_t cap, const u8* weights, size_t wtSize)
{
    u8* const ostart = (u8*)dst; u8* op = ostart; u8* const oend = ostart + cap;
    u32 maxSV = 12, tableLog = 6, count[13]; s16 norm[13]; zso_fse_ct ct;
    if (wtSize <= 1) return 0;
    {   u32 const maxCount = zso_hist(count, &maxSV, weights, wtSize);
        if (maxCount == wtSize) return 1;     /* only a single symbol: rle */
        if (maxCount == 1) return 0;          /* each symbol present at most once: not compressible */
    }
    tableLog = zso_fse_optimalTableLog(tableLog, wtSize, maxSV, 2);
    {   size_t const e = zso_fse_normalizeCount(norm, tableLog, count, wtSize, maxSV, 0); if (zso_isError(e)) return e; }
    {   size_t const h = zso_fse_writeNCount(op, (size_t)(oend - op), norm, maxSV, tableLog); if (zso_isError(h)) return h; op += h; }
    {   size_t const e = zso_fse_buildCTable(&ct, norm, maxSV, tableLog); if (zso_isError(e)) return e; }
    {   size_t const c = zso_fse_compress(op, (size_t)(oend - op), weights, wtSize, &ct);
        if (zso_isError(c)) return c;
        if (c == 0) return 0;
        op += c;
    }
    return (size_t)(op - ostart);
}
/* HUF_writeCTable_wksp, U/HufCompress.cs:168-235 */
static size_t huf_writeCTable(void* dst, size_t maxDst, const zso_huf_ct* ct, u32 maxSV, u32 huffLog)
{
    u8 bitsToWeight[13], huffWeight[256]; u8* op = (u8*)dst; u32 n;
    bitsToWeight[0] = 0;
    for (n = 1; n < huffLog + 1; n++) bitsToWeight[n] = (u8)(huffLog + 1 - n);
    for (n = 0; n < maxSV; n++) huffWeight[n] = bitsToWeight[ct->nbBits[n]];
    if (maxDst < 1) return ZSO_ERR(dstSize_tooSmall);
    {   size_t const hSize = huf_compressWeights(op + 1, maxDst - 1, huffWeight, maxSV);
        if (zso_isError(hSize)) return hSize;
        if (hSize > 1 && hSize < maxSV / 2) { op[0] = (u8)hSize; return hSize + 1; }
    }
    if (maxSV > 128) return ZSO_ERR(GENERIC);
    if (((maxSV + 1) / 2) + 1 > maxDst) return ZSO_ERR(dstSize_tooSmall);
    op[0] = (u8)(128 + (maxSV - 1));
    huffWeight[maxSV] = 0;
    for (n = 0; n < maxSV; n += 2) op[(n / 2) + 1] = (u8)((huffWeight[n] << 4) + huffWeight[n + 1]);
    return ((maxSV + 1) / 2) + 1;
}
/* HUF_compress1X_usingCTable_internal_body, U/HufCompress.cs:1056-1189: symbols last-to-first, then end mark */
static size_t huf_compress1X(void* dst, size_t cap, const u8* src, size_t n, const zso_huf_ct* ct)
{
    zso_bitc b; size_t i;
    if (cap < 8) return 0;
    if (zso_isError(bitc_init(&b, dst, cap))) return 0;
    for (i = n; i-- > 0; ) bitc_add(&b, ct->val[src[i]], ct->nbBits[src[i]]);
    return bitc_close(&b);
}
/* HUF_compress4X_usingCTable_internal, U/HufCompress.cs:1221-1321 */
static size_t huf_compress4X(void* dst, size_t cap, const u8* src, size_t n, const zso_huf_ct* ct)
{
    size_t const seg = (n + 3) / 4; const u8* ip = src; const u8* const iend = src + n;
    u8* const ostart = (u8*)dst; u8* const oend = ostart + cap; u8* op = ostart; int k;
    if (cap < 6 + 1 + 1 + 1 + 8) return 0;
    if (n < 12) return 0;
    op += 6;
    for (k = 0; k < 4; k++) {
        size_t const len = (k < 3) ? seg : (size_t)(iend - ip);
        size_t const c = huf_compress1X(op, (size_t)(oend - op), ip, len, ct);
        if (c == 0 || c > 65535) return 0;
        if (k < 3) zso_writeLE16(ostart + 2 * k, (u32)c);
        op += c; ip += len;
    }
    return (size_t)(op - ostart);
}
/* HUF_validateCTable / HUF_estimateCompressedSize, U/HufCompress.cs:825-860 */
static int huf_validateCTable(const zso_huf_ct* ct, const u32* count, u32 maxSV)
{
    u32 s; int bad = 0;
    for (s = 0; s <= maxSV; s++) bad |= (count[s] != 0) & (ct->nbBits[s] == 0);
    return !bad;
}
static size_t huf_estimate(const zso_huf_ct* ct, const u32* count, u32 maxSV)
{
    size_t nb = 0; u32 s;
    for (s = 0; s <= maxSV; s++) nb += (size_t)ct->nbBits[s] * count[s];
    return nb >> 3;
}
/* HUF_compressCTable_internal, U/HufCompress.cs:1336-1358 */
static size_t huf_body(u8* ostart, u8* op, u8* oend, const u8* src, size_t srcSize, int singleStream, const zso_huf_ct* ct)
{
    size_t const c = singleStream ? huf_compress1X(op, (size_t)(oend - op), src, srcSize, ct)
                                  : huf_compress4X(op, (size_t)(oend - op), src, srcSize, ct);
    if (c == 0) return 0;
    op += c;
    if ((size_t)(op - ostart) >= srcSize - 1) return 0;
    return (size_t)(op - ostart);
}
/* HUF_compress_internal, U/HufCompress.cs:1360-1543 */
static size_t huf_compress(void* dst, size_t dstSize, const u8* src, size_t srcSize, int singleStream,
                           zso_huf_ct* oldTable, int* repeat, int preferRepeat, u32 suspectUncompressible)
{
    u8* const ostart = (u8*)dst; u8* const oend = ostart + dstSize; u8* op = ostart;
    u32 count[256], maxSV = 255, huffLog = 11; zso_huf_ct table;
    if (!srcSize || !dstSize) return 0;
    if (srcSize > 128 * 1024) return ZSO_ERR(srcSize_wrong);
    if (preferRepeat && *repeat == HUF_repeat_valid) return huf_body(ostart, op, oend, src, srcSize, singleStream, oldTable);
    if (suspectUncompressible && srcSize >= 4096 * 10) {
        u32 m1 = 255, m2 = 255; size_t largestTotal = 0;
        largestTotal += zso_hist(count, &m1, src, 4096);
        largestTotal += zso_hist(count, &m2, src + srcSize - 4096, 4096);
        if (largestTotal <= ((2 * 4096) >> 7) + 4) return 0;
    }
    {   size_t const largest = zso_hist(count, &maxSV, src, srcSize);
        if (largest == srcSize) { *ostart = src[0]; return 1; }
        if (largest <= (srcSize >> 7) + 4) return 0;
    }
    if (*repeat == HUF_repeat_check && !huf_validateCTable(oldTable, count, maxSV)) *repeat = HUF_repeat_none;
    if (preferRepeat && *repeat != HUF_repeat_none) return huf_body(ostart, op, oend, src, srcSize, singleStream, oldTable);
    huffLog = zso_fse_optimalTableLog(huffLog, srcSize, maxSV, 1);
    {   size_t const maxBits = huf_buildCTable(&table, count, maxSV, huffLog);
        if (zso_isError(maxBits)) return maxBits;
        huffLog = (u32)maxBits;
    }
    {   size_t const hSize = huf_writeCTable(op, dstSize, &table, maxSV, huffLog);
        if (zso_isError(hSize)) return hSize;
        if (*repeat != HUF_repeat_none) {
            size_t const oldSize = huf_estimate(oldTable, count, maxSV), newSize = huf_estimate(&table, count, maxSV);
            if (oldSize <= hSize + newSize || hSize + 12 >= srcSize) return huf_body(ostart, op, oend, src, srcSize, singleStream, oldTable);
        }
        if (hSize + 12 >= srcSize) return 0;
        op += hSize;
        *repeat = HUF_repeat_none;
        *oldTable = table;
    }
    return huf_body(ostart, op, oend, src, srcSize, singleStream, &table);
}

/* ------------------------------------------------------------------ */
/*  block entropy state                                                */
/* ------------------------------------------------------------------ */
enum { FSE_repeat_none = 0, FSE_repeat_check = 1, FSE_repeat_valid = 2 };
typedef struct {
    zso_huf_ct huf; int hufRepeat;
    zso_fse_ct ll, of, ml; int llRepeat, ofRepeat, mlRepeat;
    u32 rep[3];
} zso_bstate;

static void bstate_reset(zso_bstate* s)     /* ZSTD_reset_compressedBlockState, U/ZstdCompress.cs:2419-2433 */
{
    memset(s, 0, sizeof *s);
    s->rep[0] = 1; s->rep[1] = 4; s->rep[2] = 8;
    s->hufRepeat = HUF_repeat_none; s->llRepeat = s->ofRepeat = s->mlRepeat = FSE_repeat_none;
}

static size_t zso_minGain(size_t srcSize) { return (srcSize >> 6) + 2; }   /* strategies < btultra */

/* ZSTD_noCompressLiterals / ZSTD_compressRleLiteralsBlock, U/ZstdCompressLiterals.cs:8-83 */
static size_t lit_raw(void* dst, size_t cap, const u8* src, size_t n)
{
    u8* const o = (u8*)dst; u32 const flSize = 1 + (n > 31) + (n > 4095);
    if (n + flSize > cap) return ZSO_ERR(dstSize_tooSmall);
    switch (flSize) {
    case 1: o[0] = (u8)(0 + (n << 3)); break;
    case 2: zso_writeLE16(o, (u32)(0 + (1 << 2) + (n << 4))); break;
    default: zso_writeLE32(o, (u32)(0 + (3 << 2) + (n << 4))); break;
    }
    memcpy(o + flSize, src, n);
    return n + flSize;
}
static size_t lit_rle(void* dst, size_t cap, const u8* src, size_t n)
{
    u8* const o = (u8*)dst; u32 const flSize = 1 + (n > 31) + (n > 4095);
    (void)cap;
    switch (flSize) {
    case 1: o[0] = (u8)(1 + (n << 3)); break;
    case 2: zso_writeLE16(o, (u32)(1 + (1 << 2) + (n << 4))); break;
    default: zso_writeLE32(o, (u32)(1 + (3 << 2) + (n << 4))); break;
    }
    o[flSize] = src[0];
    return flSize + 1;
}
/* ZSTD_compressLiterals, U/ZstdCompressLiterals.cs:86-185 */
static size_t zso_compressLiterals(const zso_bstate* prev, zso_bstate* next, u32 strategy, int disableLiteralCompression,
                                   void* dst, size_t cap, const u8* src, size_t srcSize, u32 suspectUncompressible)
{
    size_t const minGain = zso_minGain(srcSize);
    size_t const lhSize = 3 + (srcSize >= 1024) + (srcSize >= 16384);
    u8* const ostart = (u8*)dst; int singleStream = srcSize < 256; u32 hType = 2; size_t cLitSize;
    next->huf = prev->huf; next->hufRepeat = prev->hufRepeat;
    if (disableLiteralCompression) return lit_raw(dst, cap, src, srcSize);
    {   size_t const minLitSize = (prev->hufRepeat == HUF_repeat_valid) ? 6 : 63;
        if (srcSize <= minLitSize) return lit_raw(dst, cap, src, srcSize);
    }
    if (cap < lhSize + 1) return ZSO_ERR(dstSize_tooSmall);
    {   int repeat = prev->hufRepeat;
        int const preferRepeat = strategy < ZSO_lazy ? srcSize <= 1024 : 0;
        if (repeat == HUF_repeat_valid && lhSize == 3) singleStream = 1;
        cLitSize = huf_compress(ostart + lhSize, cap - lhSize, src, srcSize, singleStream, &next->huf, &repeat, preferRepeat, suspectUncompressible);
        if (repeat != HUF_repeat_none) hType = 3;     /* reused the previous table */
    }
    if (cLitSize == 0 || cLitSize >= srcSize - minGain || zso_isError(cLitSize)) {
        next->huf = prev->huf; next->hufRepeat = prev->hufRepeat;
        return lit_raw(dst, cap, src, srcSize);
    }
    if (cLitSize == 1) {
        next->huf = prev->huf; next->hufRepeat = prev->hufRepeat;
        return lit_rle(dst, cap, src, srcSize);
    }
    if (hType == 2) next->hufRepeat = HUF_repeat_check;
    switch (lhSize) {
    case 3: zso_writeLE24(ostart, hType + ((u32)!singleStream << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 14)); break;
    case 4: zso_writeLE32(ostart, hType + (2 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 18)); break;
    default: zso_writeLE32(ostart, hType + (3 << 2) + ((u32)srcSize << 4) + ((u32)cLitSize << 22)); ostart[4] = (u8)(cLitSize >> 10); break;
    }
    return lhSize + cLitSize;
}

/* ------------------------------------------------------------------ */
/*  sequences                                                          */
/* ------------------------------------------------------------------ */
static const u8 ZSO_LL_Code[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
    22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
static const u8 ZSO_ML_Code[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
    32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
    40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
    42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };
static inline u32 zso_LLcode(u32 ll) { return ll > 63 ? zso_highbit32(ll) + 19 : ZSO_LL_Code[ll]; }
static inline u32 zso_MLcode(u32 ml) { return ml > 127 ? zso_highbit32(ml) + 36 : ZSO_ML_Code[ml]; }

typedef struct {
    zso_seq* seqs; size_t nbSeq, cap;
    u8* lit; size_t litSize;
    u32 longLengthType, longLengthPos;      /* 0 none, 1 literalLength, 2 matchLength (ZSTD_longLengthType_e) */
    u8 *llCode, *mlCode, *ofCode;
} zso_seqstore;

/* ZSTD_selectEncodingType (strategy < lazy branch), U/ZstdCompressSequences.cs:400-469 */
static u32 zso_selectEncodingType(int* repeatMode, size_t mostFrequent, size_t nbSeq, u32 defaultNormLog, int isDefaultAllowed, u32 strategy)
{
    if (mostFrequent == nbSeq) {
        *repeatMode = FSE_repeat_none;
        if (isDefaultAllowed && nbSeq <= 2) return 0;   /* set_basic */
        return 1;                                        /* set_rle */
    }
    /* strategy >= lazy would compare estimated costs; the oracle only restates levels whose strategy is < lazy */
    if (isDefaultAllowed) {
        size_t const mult = 10 - strategy, dynamicFse_nbSeq_min = (((size_t)1 << defaultNormLog) * mult) >> 3;
        if (*repeatMode == FSE_repeat_valid && nbSeq < 1000) return 3;   /* set_repeat */
        if (nbSeq < dynamicFse_nbSeq_min || mostFrequent < (nbSeq >> (defaultNormLog - 1))) { *repeatMode = FSE_repeat_none; return 0; }
    }
    *repeatMode = FSE_repeat_check;
    return 2;   /* set_compressed */
}
/* ZSTD_buildCTable, U/ZstdCompressSequences.cs:471-582 */
static size_t zso_buildSeqCTable(void* dst, size_t cap, zso_fse_ct* next, u32 FSELog, u32 type, u32* count, u32 max,
                                 const u8* codeTable, size_t nbSeq, const s16* defaultNorm, u32 defaultNormLog, u32 defaultMax,
                                 const zso_fse_ct* prev)
{
    u8* op = (u8*)dst;
    switch (type) {
    case 1: zso_fse_buildCTable_rle(next, (u8)max); if (!cap) return ZSO_ERR(dstSize_tooSmall); *op = codeTable[0]; return 1;
    case 3: *next = *prev; return 0;
    case 0: { size_t const e = zso_fse_buildCTable(next, defaultNorm, defaultMax, defaultNormLog); return zso_isError(e) ? e : 0; }
    default: {
        s16 norm[64]; size_t nbSeq_1 = nbSeq;
        u32 const tableLog = zso_fse_optimalTableLog(FSELog, nbSeq, max, 2);
        if (count[codeTable[nbSeq - 1]] > 1) { count[codeTable[nbSeq - 1]]--; nbSeq_1--; }
        {   size_t const e = zso_fse_normalizeCount(norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048); if (zso_isError(e)) return e; }
        {   size_t const n = zso_fse_writeNCount(op, cap, norm, max, tableLog);
            if (zso_isError(n)) return n;
            {   size_t const e = zso_fse_buildCTable(next, norm, max, tableLog); if (zso_isError(e)) return e; }
            return n;
        } }
    }
}

/* ZSTD_encodeSequences_body, U/ZstdCompressSequences.cs:585-704 (64-bit, no long offsets) */
static size_t zso_encodeSequences(void* dst, size_t cap, const zso_fse_ct* ctML, const u8* mlCode,
                                  const zso_fse_ct* ctOF, const u8* ofCode, const zso_fse_ct* ctLL, const u8* llCode,
                                  const zso_seq* seqs, size_t nbSeq)
{
    zso_bitc b; zso_cstate sML, sOF, sLL; size_t n;
    if (zso_isError(bitc_init(&b, dst, cap))) return ZSO_ERR(dstSize_tooSmall);
    cstate_init2(&sML, ctML, mlCode[nbSeq - 1]);
    cstate_init2(&sOF, ctOF, ofCode[nbSeq - 1]);
    cstate_init2(&sLL, ctLL, llCode[nbSeq - 1]);
    bitc_add(&b, seqs[nbSeq - 1].litLength, ZSO_LL_bits[llCode[nbSeq - 1]]);
    bitc_add(&b, seqs[nbSeq - 1].mlBase, ZSO_ML_bits[mlCode[nbSeq - 1]]);
    bitc_add(&b, seqs[nbSeq - 1].offBase, ofCode[nbSeq - 1]);
    for (n = nbSeq - 1; n-- > 0; ) {
        u8 const ll = llCode[n], of = ofCode[n], ml = mlCode[n];
        cstate_encode(&b, &sOF, of);
        cstate_encode(&b, &sML, ml);
        cstate_encode(&b, &sLL, ll);
        bitc_add(&b, seqs[n].litLength, ZSO_LL_bits[ll]);
        bitc_add(&b, seqs[n].mlBase, ZSO_ML_bits[ml]);
        bitc_add(&b, seqs[n].offBase, of);
    }
    cstate_flush(&b, &sML); cstate_flush(&b, &sOF); cstate_flush(&b, &sLL);
    {   size_t const sz = bitc_close(&b); if (!sz) return ZSO_ERR(dstSize_tooSmall); return sz; }
}

/* ZSTD_entropyCompressSeqStore_internal, U/ZstdCompress.cs:3236-3354 */
static size_t zso_entropyCompressSeqStore_internal(zso_seqstore* ss, const zso_bstate* prev, zso_bstate* next,
                                                   u32 strategy, int disableLiteralCompression, void* dst, size_t cap)
{
    u8* const ostart = (u8*)dst; u8* const oend = ostart + cap; u8* op = ostart;
    size_t const nbSeq = ss->nbSeq; size_t lastCountSize = 0; u32 count[64];
    {   u32 const suspect = (nbSeq == 0) || (ss->litSize / nbSeq >= 20);
        size_t const c = zso_compressLiterals(prev, next, strategy, disableLiteralCompression, op, cap, ss->lit, ss->litSize, suspect);
        if (zso_isError(c)) return c;
        op += c;
    }
    if (oend - op < 3 + 1) return ZSO_ERR(dstSize_tooSmall);
    if (nbSeq < 128) *op++ = (u8)nbSeq;
    else if (nbSeq < 0x7F00) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; op += 2; }
    else { op[0] = 0xFF; zso_writeLE16(op + 1, (u32)(nbSeq - 0x7F00)); op += 3; }
    if (nbSeq == 0) {
        next->ll = prev->ll; next->of = prev->of; next->ml = prev->ml;
        next->llRepeat = prev->llRepeat; next->ofRepeat = prev->ofRepeat; next->mlRepeat = prev->mlRepeat;
        return (size_t)(op - ostart);
    }
    {   u8* const seqHead = op++; u32 LLtype, OFtype, MLtype; size_t n;
        /* ZSTD_seqToCodes, U/ZstdCompress.cs:3069-3098 */
        for (n = 0; n < nbSeq; n++) {
            ss->llCode[n] = (u8)zso_LLcode(ss->seqs[n].litLength);
            ss->ofCode[n] = (u8)zso_highbit32(ss->seqs[n].offBase);
            ss->mlCode[n] = (u8)zso_MLcode(ss->seqs[n].mlBase);
        }
        if (ss->longLengthType == 1) ss->llCode[ss->longLengthPos] = ZSO_MaxLL;
        if (ss->longLengthType == 2) ss->mlCode[ss->longLengthPos] = ZSO_MaxML;
        /* ZSTD_buildSequencesStatistics, U/ZstdCompress.cs:3127-3233 */
        {   u32 max = ZSO_MaxLL; size_t const mf = zso_hist(count, &max, ss->llCode, nbSeq); size_t cs;
            next->llRepeat = prev->llRepeat;
            LLtype = zso_selectEncodingType(&next->llRepeat, mf, nbSeq, ZSO_LL_DEFAULTNORMLOG, 1, strategy);
            cs = zso_buildSeqCTable(op, (size_t)(oend - op), &next->ll, ZSO_LLFSELog, LLtype, count, max, ss->llCode, nbSeq, ZSO_LL_defaultNorm, ZSO_LL_DEFAULTNORMLOG, ZSO_MaxLL, &prev->ll);
            if (zso_isError(cs)) return cs;
            if (LLtype == 2) lastCountSize = cs;
            op += cs;
        }
        {   u32 max = ZSO_MaxOff; size_t const mf = zso_hist(count, &max, ss->ofCode, nbSeq); size_t cs;
            int const defaultAllowed = max <= 28;
            next->ofRepeat = prev->ofRepeat;
            OFtype = zso_selectEncodingType(&next->ofRepeat, mf, nbSeq, ZSO_OF_DEFAULTNORMLOG, defaultAllowed, strategy);
            cs = zso_buildSeqCTable(op, (size_t)(oend - op), &next->of, ZSO_OffFSELog, OFtype, count, max, ss->ofCode, nbSeq, ZSO_OF_defaultNorm, ZSO_OF_DEFAULTNORMLOG, 28, &prev->of);
            if (zso_isError(cs)) return cs;
            if (OFtype == 2) lastCountSize = cs;
            op += cs;
        }
        {   u32 max = ZSO_MaxML; size_t const mf = zso_hist(count, &max, ss->mlCode, nbSeq); size_t cs;
            next->mlRepeat = prev->mlRepeat;
            MLtype = zso_selectEncodingType(&next->mlRepeat, mf, nbSeq, ZSO_ML_DEFAULTNORMLOG, 1, strategy);
            cs = zso_buildSeqCTable(op, (size_t)(oend - op), &next->ml, ZSO_MLFSELog, MLtype, count, max, ss->mlCode, nbSeq, ZSO_ML_defaultNorm, ZSO_ML_DEFAULTNORMLOG, ZSO_MaxML, &prev->ml);
            if (zso_isError(cs)) return cs;
            if (MLtype == 2) lastCountSize = cs;
            op += cs;
        }
        *seqHead = (u8)((LLtype << 6) + (OFtype << 4) + (MLtype << 2));
    }
    {   size_t const bs = zso_encodeSequences(op, (size_t)(oend - op), &next->ml, ss->mlCode, &next->of, ss->ofCode, &next->ll, ss->llCode, ss->seqs, nbSeq);
        if (zso_isError(bs)) return bs;
        op += bs;
        if (lastCountSize && (lastCountSize + bs) < 4) return 0;   /* 1.3.4 decoder quirk, :3346-3350 */
    }
    return (size_t)(op - ostart);
}
/* ZSTD_entropyCompressSeqStore, U/ZstdCompress.cs:3357-3392 */
static size_t zso_entropyCompressSeqStore(zso_seqstore* ss, const zso_bstate* prev, zso_bstate* next, u32 strategy,
                                          int disableLiteralCompression, void* dst, size_t cap, size_t srcSize)
{
    size_t const c = zso_entropyCompressSeqStore_internal(ss, prev, next, strategy, disableLiteralCompression, dst, cap);
    if (c == 0) return 0;
    if (c == ZSO_ERR(dstSize_tooSmall) && srcSize <= cap) return 0;
    if (zso_isError(c)) return c;
    if (c >= srcSize - zso_minGain(srcSize)) return 0;
    return c;
}

/* ------------------------------------------------------------------ */
/*  fast match finder                                                  */
/* ------------------------------------------------------------------ */
static inline size_t zso_hashPtr(const u8* p, u32 hBits, u32 mls)
{
    switch (mls) {
    default:
    case 4: return (zso_readLE32(p) * 2654435761U) >> (32 - hBits);
    case 5: return (size_t)(((zso_readLE64(p) << (64 - 40)) * 889523592379ULL) >> (64 - hBits));
    case 6: return (size_t)(((zso_readLE64(p) << (64 - 48)) * 227718039650203ULL) >> (64 - hBits));
    case 7: return (size_t)(((zso_readLE64(p) << (64 - 56)) * 58295818150454627ULL) >> (64 - hBits));
    case 8: return (size_t)((zso_readLE64(p) * 0xCF1BBCDCB7A56463ULL) >> (64 - hBits));
    }
}
static size_t zso_count(const u8* ip, const u8* match, const u8* iend)
{
    const u8* const start = ip;
    while (ip < iend && *ip == *match) { ip++; match++; }
    return (size_t)(ip - start);
}
static void zso_storeSeq(zso_seqstore* ss, size_t litLength, const u8* literals, u32 offCode, size_t mlBase)
{
    memcpy(ss->lit + ss->litSize, literals, litLength); ss->litSize += litLength;
    if (litLength > 0xFFFF) { ss->longLengthType = 1; ss->longLengthPos = (u32)ss->nbSeq; }
    ss->seqs[ss->nbSeq].litLength = (u16)litLength;
    ss->seqs[ss->nbSeq].offBase = offCode + 1;
    if (mlBase > 0xFFFF) { ss->longLengthType = 2; ss->longLengthPos = (u32)ss->nbSeq; }
    ss->seqs[ss->nbSeq].mlBase = (u16)mlBase;
    ss->nbSeq++;
}

/* the match state that persists across the blocks of one frame */
typedef struct {
    zso_cparams cp;
    const u8* base;            /* index i <-> base + i ; the first source byte has index 2 (U/ZstdCompressInternal.cs:723-732) */
    u32 dictLimit, lowLimit;
    u32* hashTable;
    u32* chainTable;           /* doubleFast: the short-match table (1 << chainLog) */
    u16* tagTable;             /* row-hash match finder: 1 << hashLog u16 (head byte + tags per row) */
    u32 rowHashLog, nextToUpdate, hashCache[8];
} zso_mstate;

static u32 ms_lowestPrefixIndex(const zso_mstate* ms, u32 curr, u32 windowLog)
{
    u32 const maxDistance = 1u << windowLog, lowestValid = ms->dictLimit;
    return (curr - lowestValid > maxDistance) ? curr - maxDistance : lowestValid;
}

/* ZSTD_compressBlock_fast_noDict_generic, U/ZstdFast.cs:96-288.  Returns the trailing-literals size. */
static size_t zso_compressBlock_fast(zso_mstate* ms, zso_seqstore* ss, u32 rep[3], const u8* src, size_t srcSize)
{
    u32* const hashTable = ms->hashTable;
    u32 const hlog = ms->cp.hashLog, mls = ms->cp.minMatch;
    size_t const stepSize = (ms->cp.targetLength > 1) ? (ms->cp.targetLength + !ms->cp.targetLength + 1) : 2;
    const u8* const base = ms->base; const u8* const istart = src;
    u32 const endIndex = (u32)((size_t)(istart - base) + srcSize);
    u32 const prefixStartIndex = ms_lowestPrefixIndex(ms, endIndex, ms->cp.windowLog);
    const u8* const prefixStart = base + prefixStartIndex;
    const u8* const iend = istart + srcSize; const u8* const ilimit = iend - 8;
    const u8* anchor = istart; const u8* ip0 = istart; const u8 *ip1, *ip2, *ip3;
    u32 current0, rep1 = rep[0], rep2 = rep[1], offsetSaved = 0;
    size_t hash0, hash1; u32 idx, mval, offcode; const u8* match0; size_t mLength, step; const u8* nextStep;
    size_t const kStepIncr = 1 << 7;
    if (srcSize < 8) return srcSize;   /* not reachable from the block loop (blocks < 7 bytes are stored raw, and ilimit guards the rest) */
    ip0 += (ip0 == prefixStart);
    {   u32 const curr = (u32)(ip0 - base), windowLow = ms_lowestPrefixIndex(ms, curr, ms->cp.windowLog), maxRep = curr - windowLow;
        if (rep2 > maxRep) { offsetSaved = rep2; rep2 = 0; }
        if (rep1 > maxRep) { offsetSaved = rep1; rep1 = 0; }
    }
    for (;;) {      /* _start */
        int found = 0;     /* 1 = repcode match at ip2, 2 = hash-table match at ip0 */
        step = stepSize; nextStep = ip0 + kStepIncr;
        ip1 = ip0 + 1; ip2 = ip0 + step; ip3 = ip2 + 1;
        if (ip3 >= ilimit) break;
        hash0 = zso_hashPtr(ip0, hlog, mls); hash1 = zso_hashPtr(ip1, hlog, mls);
        idx = hashTable[hash0];
        do {
            u32 const rval = zso_readLE32(ip2 - rep1);
            current0 = (u32)(ip0 - base); hashTable[hash0] = current0;
            if (zso_readLE32(ip2) == rval && rep1 > 0) {
                ip0 = ip2; match0 = ip0 - rep1;
                mLength = ip0[-1] == match0[-1];
                ip0 -= mLength; match0 -= mLength;
                offcode = 0; mLength += 4;
                found = 1; break;
            }
            mval = (idx >= prefixStartIndex) ? zso_readLE32(base + idx) : zso_readLE32(ip0) ^ 1;
            if (zso_readLE32(ip0) == mval) { found = 2; break; }
            idx = hashTable[hash1]; hash0 = hash1; hash1 = zso_hashPtr(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip3;
            current0 = (u32)(ip0 - base); hashTable[hash0] = current0;
            mval = (idx >= prefixStartIndex) ? zso_readLE32(base + idx) : zso_readLE32(ip0) ^ 1;
            if (zso_readLE32(ip0) == mval) { found = 2; break; }
            idx = hashTable[hash1]; hash0 = hash1; hash1 = zso_hashPtr(ip2, hlog, mls);
            ip0 = ip1; ip1 = ip2; ip2 = ip0 + step; ip3 = ip1 + step;
            if (ip2 >= nextStep) { step++; nextStep += kStepIncr; }
        } while (ip3 < ilimit);
        if (!found) break;      /* _cleanup */
        if (found == 2) {       /* _offset */
            match0 = base + idx;
            rep2 = rep1; rep1 = (u32)(ip0 - match0);
            offcode = rep1 + 2; mLength = 4;
            while (ip0 > anchor && match0 > prefixStart && ip0[-1] == match0[-1]) { ip0--; match0--; mLength++; }
        }
        /* _match */
        mLength += zso_count(ip0 + mLength, match0 + mLength, iend);
        zso_storeSeq(ss, (size_t)(ip0 - anchor), anchor, offcode, mLength - 3);
        ip0 += mLength; anchor = ip0;
        if (ip1 < ip0) hashTable[hash1] = (u32)(ip1 - base);
        if (ip0 <= ilimit) {
            hashTable[zso_hashPtr(base + current0 + 2, hlog, mls)] = current0 + 2;
            hashTable[zso_hashPtr(ip0 - 2, hlog, mls)] = (u32)(ip0 - 2 - base);
            if (rep2 > 0) {
                while (ip0 <= ilimit && zso_readLE32(ip0) == zso_readLE32(ip0 - rep2)) {
                    size_t const rLength = zso_count(ip0 + 4, ip0 + 4 - rep2, iend) + 4;
                    { u32 const t = rep2; rep2 = rep1; rep1 = t; }
                    hashTable[zso_hashPtr(ip0, hlog, mls)] = (u32)(ip0 - base);
                    ip0 += rLength;
                    zso_storeSeq(ss, 0, anchor, 0, rLength - 3);
                    anchor = ip0;
                }
            }
        }
    }
    rep[0] = rep1 ? rep1 : offsetSaved;
    rep[1] = rep2 ? rep2 : offsetSaved;
    return (size_t)(iend - anchor);
}


/* ZSTD_compressBlock_doubleFast_noDict_generic, U/ZstdDoubleFast.cs:51-247.  Returns the trailing-literals size. */
static size_t zso_compressBlock_doubleFast(zso_mstate* ms, zso_seqstore* ss, u32 rep[3], const u8* src, size_t srcSize)
{
    u32* const hashLong = ms->hashTable; u32 const hBitsL = ms->cp.hashLog;
    u32* const hashSmall = ms->chainTable; u32 const hBitsS = ms->cp.chainLog;
    u32 const mls = ms->cp.minMatch;
    const u8* const base = ms->base; const u8* const istart = src; const u8* anchor = istart;
    u32 const endIndex = (u32)((size_t)(istart - base) + srcSize);
    u32 const prefixLowestIndex = ms_lowestPrefixIndex(ms, endIndex, ms->cp.windowLog);
    const u8* const prefixLowest = base + prefixLowestIndex;
    const u8* const iend = istart + srcSize; const u8* const ilimit = iend - 8;
    u32 offset_1 = rep[0], offset_2 = rep[1], offsetSaved = 0;
    size_t mLength; u32 offset = 0, curr = 0;
    size_t const kStepIncr = 1 << 8;
    const u8* nextStep; size_t step; size_t hl0, hl1; u32 idxl0, idxl1;
    const u8 *matchl0, *matchs0, *matchl1; const u8* ip = istart; const u8* ip1;
    if (srcSize < 8) return srcSize;
    ip += ((ip - prefixLowest) == 0);
    {   u32 const current = (u32)(ip - base), windowLow = ms_lowestPrefixIndex(ms, current, ms->cp.windowLog), maxRep = current - windowLow;
        if (offset_2 > maxRep) { offsetSaved = offset_2; offset_2 = 0; }
        if (offset_1 > maxRep) { offsetSaved = offset_1; offset_1 = 0; }
    }
    for (;;) {
        int how = 0;    /* 1 = repcode (already stored), 2 = long at ip, 3 = search-next-long */
        step = 1; nextStep = ip + kStepIncr; ip1 = ip + step;
        if (ip1 > ilimit) break;
        hl0 = zso_hashPtr(ip, hBitsL, 8); idxl0 = hashLong[hl0]; matchl0 = base + idxl0;
        do {
            size_t const hs0 = zso_hashPtr(ip, hBitsS, mls);
            u32 const idxs0 = hashSmall[hs0];
            curr = (u32)(ip - base); matchs0 = base + idxs0;
            hashLong[hl0] = hashSmall[hs0] = curr;
            if (offset_1 > 0 && zso_readLE32(ip + 1 - offset_1) == zso_readLE32(ip + 1)) {
                mLength = zso_count(ip + 1 + 4, ip + 1 + 4 - offset_1, iend) + 4;
                ip++;
                zso_storeSeq(ss, (size_t)(ip - anchor), anchor, 0, mLength - 3);
                how = 1; break;
            }
            hl1 = zso_hashPtr(ip1, hBitsL, 8);
            if (idxl0 > prefixLowestIndex && zso_readLE64(matchl0) == zso_readLE64(ip)) {
                mLength = zso_count(ip + 8, matchl0 + 8, iend) + 8;
                offset = (u32)(ip - matchl0);
                while (ip > anchor && matchl0 > prefixLowest && ip[-1] == matchl0[-1]) { ip--; matchl0--; mLength++; }
                how = 2; break;
            }
            idxl1 = hashLong[hl1]; matchl1 = base + idxl1;
            if (idxs0 > prefixLowestIndex && zso_readLE32(matchs0) == zso_readLE32(ip)) { how = 3; break; }
            if (ip1 >= nextStep) { step++; nextStep += kStepIncr; }
            ip = ip1; ip1 += step;
            hl0 = hl1; idxl0 = idxl1; matchl0 = matchl1;
        } while (ip1 <= ilimit);
        if (!how) break;
        if (how == 3) {         /* _search_next_long */
            if (idxl1 > prefixLowestIndex && zso_readLE64(matchl1) == zso_readLE64(ip1)) {
                ip = ip1;
                mLength = zso_count(ip + 8, matchl1 + 8, iend) + 8;
                offset = (u32)(ip - matchl1);
                while (ip > anchor && matchl1 > prefixLowest && ip[-1] == matchl1[-1]) { ip--; matchl1--; mLength++; }
            } else {
                mLength = zso_count(ip + 4, matchs0 + 4, iend) + 4;
                offset = (u32)(ip - matchs0);
                while (ip > anchor && matchs0 > prefixLowest && ip[-1] == matchs0[-1]) { ip--; matchs0--; mLength++; }
            }
        }
        if (how != 1) {         /* _match_found */
            offset_2 = offset_1; offset_1 = offset;
            if (step < 4) hashLong[hl1] = (u32)(ip1 - base);
            zso_storeSeq(ss, (size_t)(ip - anchor), anchor, offset + 2, mLength - 3);
        }
        /* _match_stored */
        ip += mLength; anchor = ip;
        if (ip <= ilimit) {
            {   u32 const indexToInsert = curr + 2;
                hashLong[zso_hashPtr(base + indexToInsert, hBitsL, 8)] = indexToInsert;
                hashLong[zso_hashPtr(ip - 2, hBitsL, 8)] = (u32)(ip - 2 - base);
                hashSmall[zso_hashPtr(base + indexToInsert, hBitsS, mls)] = indexToInsert;
                hashSmall[zso_hashPtr(ip - 1, hBitsS, mls)] = (u32)(ip - 1 - base);
            }
            while (ip <= ilimit && offset_2 > 0 && zso_readLE32(ip) == zso_readLE32(ip - offset_2)) {
                size_t const rLength = zso_count(ip + 4, ip + 4 - offset_2, iend) + 4;
                u32 const tmpOff = offset_2; offset_2 = offset_1; offset_1 = tmpOff;
                hashSmall[zso_hashPtr(ip, hBitsS, mls)] = (u32)(ip - base);
                hashLong[zso_hashPtr(ip, hBitsL, 8)] = (u32)(ip - base);
                zso_storeSeq(ss, 0, anchor, 0, rLength - 3);
                ip += rLength; anchor = ip;
            }
        }
    }
    rep[0] = offset_1 ? offset_1 : offsetSaved;
    rep[1] = offset_2 ? offset_2 : offsetSaved;
    return (size_t)(iend - anchor);
}


/* ------------------------------------------------------------------ */
/*  greedy / lazy parse over the row-hash match finder (levels 4-5+)   */
/* ------------------------------------------------------------------ */
/* U/ZstdLazy.cs:788-1309 (row helpers + ZSTD_RowFindBestMatch, noDict), :1743-2032 (ZSTD_compressBlock_lazy_generic). */
#define ZSO_ROW_TAG_OFFSET 16
static u32 row_nextIndex(u8* tagRow, u32 rowMask) { u32 const next = (u32)(*tagRow - 1) & rowMask; *tagRow = (u8)next; return next; }
static u32 row_hash(const zso_mstate* ms, u32 idx, u32 mls) { return (u32)zso_hashPtr(ms->base + idx, ms->rowHashLog + 8, mls); }
static void row_fillHashCache(zso_mstate* ms, u32 mls, u32 idx, const u8* iLimit)
{
    u32 const maxElems = (ms->base + idx) > iLimit ? 0 : (u32)(iLimit - (ms->base + idx) + 1);
    u32 const lim = idx + (8 < maxElems ? 8 : maxElems);
    for (; idx < lim; ++idx) ms->hashCache[idx & 7] = row_hash(ms, idx, mls);
}
static u32 row_nextCachedHash(zso_mstate* ms, u32 idx, u32 mls)
{
    u32 const newHash = row_hash(ms, idx + 8, mls);
    u32 const hash = ms->hashCache[idx & 7];
    ms->hashCache[idx & 7] = newHash;
    return hash;
}
static void row_update_impl(zso_mstate* ms, u32 start, u32 end, u32 mls, u32 rowLog, u32 rowMask, int useCache)
{
    for (; start < end; ++start) {
        u32 const hash = useCache ? row_nextCachedHash(ms, start, mls) : row_hash(ms, start, mls);
        u32 const relRow = (hash >> 8) << rowLog;
        u32* const row = ms->hashTable + relRow;
        u8* const tagRow = (u8*)(ms->tagTable + relRow);
        u32 const pos = row_nextIndex(tagRow, rowMask);
        tagRow[pos + ZSO_ROW_TAG_OFFSET] = (u8)hash;
        row[pos] = start;
    }
}
static void row_update(zso_mstate* ms, const u8* ip, u32 mls, u32 rowLog, u32 rowMask, int useCache)
{
    u32 idx = ms->nextToUpdate; u32 const target = (u32)(ip - ms->base);
    if (useCache && target - idx > 384) {
        row_update_impl(ms, idx, idx + 96, mls, rowLog, rowMask, useCache);
        idx = target - 32;
        row_fillHashCache(ms, mls, idx, ip + 1);
    }
    row_update_impl(ms, idx, target, mls, rowLog, rowMask, useCache);
    ms->nextToUpdate = target;
}
static u64 row_getMatchMask(const u8* tagRow, u8 tag, u32 head, u32 rowEntries)
{
    u64 m = 0; u32 i;
    for (i = 0; i < rowEntries; i++) if (tagRow[ZSO_ROW_TAG_OFFSET + i] == tag) m |= (u64)1 << i;
    if (head == 0) return m;
    if (rowEntries == 64) return (m >> head) | (m << (64 - head));
    return ((m >> head) | (m << (rowEntries - head))) & (((u64)1 << rowEntries) - 1);
}
/* ZSTD_RowFindBestMatch, noDict */
static size_t row_findBestMatch(zso_mstate* ms, const u8* ip, const u8* iLimit, size_t* offsetPtr, u32 mls, u32 rowLog)
{
    const u8* const base = ms->base;
    u32 const curr = (u32)(ip - base), maxDistance = 1u << ms->cp.windowLog, lowestValid = ms->lowLimit;
    u32 const lowLimit = (curr - lowestValid > maxDistance) ? curr - maxDistance : lowestValid;
    u32 const rowEntries = 1u << rowLog, rowMask = rowEntries - 1;
    u32 const cappedSearchLog = ms->cp.searchLog < rowLog ? ms->cp.searchLog : rowLog;
    u32 nbAttempts = 1u << cappedSearchLog;
    size_t ml = 4 - 1;
    row_update(ms, ip, mls, rowLog, rowMask, 1);
    {   u32 const hash = row_nextCachedHash(ms, curr, mls);
        u32 const relRow = (hash >> 8) << rowLog; u8 const tag = (u8)hash;
        u32* const row = ms->hashTable + relRow; u8* const tagRow = (u8*)(ms->tagTable + relRow);
        u32 const head = *tagRow & rowMask;
        u32 matchBuffer[64]; size_t numMatches = 0, currMatch;
        u64 matches = row_getMatchMask(tagRow, tag, head, rowEntries);
        for (; matches > 0 && nbAttempts > 0; --nbAttempts, matches &= (matches - 1)) {
            u32 const matchPos = (head + (u32)__builtin_ctzll(matches)) & rowMask;
            u32 const matchIndex = row[matchPos];
            if (matchIndex < lowLimit) break;
            matchBuffer[numMatches++] = matchIndex;
        }
        {   u32 const pos = row_nextIndex(tagRow, rowMask);
            tagRow[pos + ZSO_ROW_TAG_OFFSET] = tag;
            row[pos] = ms->nextToUpdate++;
        }
        for (currMatch = 0; currMatch < numMatches; ++currMatch) {
            u32 const matchIndex = matchBuffer[currMatch];
            const u8* const match = base + matchIndex;
            size_t currentMl = 0;
            if (match[ml] == ip[ml]) currentMl = zso_count(ip, match, iLimit);
            if (currentMl > ml) {
                ml = currentMl; *offsetPtr = curr - matchIndex + 2;
                if (ip + currentMl == iLimit) break;
            }
        }
    }
    return ml;
}
/* ZSTD_compressBlock_lazy_generic, noDict + row hash; depth 0 = greedy, 1 = lazy, 2 = lazy2 */
static size_t zso_compressBlock_lazy_row(zso_mstate* ms, zso_seqstore* ss, u32 rep[3], const u8* src, size_t srcSize, u32 depth)
{
    const u8* const istart = src; const u8* ip = istart; const u8* anchor = istart;
    const u8* const iend = istart + srcSize; const u8* const ilimit = iend - 8 - 8;
    const u8* const base = ms->base;
    u32 const prefixLowestIndex = ms->dictLimit; const u8* const prefixLowest = base + prefixLowestIndex;
    u32 offset_1 = rep[0], offset_2 = rep[1], savedOffset = 0;
    u32 const sl = ms->cp.searchLog, rowLog = sl < 4 ? 4 : (sl > 6 ? 6 : sl);
    u32 const mls = ms->cp.minMatch < 4 ? 4 : (ms->cp.minMatch > 6 ? 6 : ms->cp.minMatch);
    if (srcSize < 16) return srcSize;
    ip += ((ip - prefixLowest) == 0);
    {   u32 const curr = (u32)(ip - base), windowLow = ms_lowestPrefixIndex(ms, curr, ms->cp.windowLog), maxRep = curr - windowLow;
        if (offset_2 > maxRep) { savedOffset = offset_2; offset_2 = 0; }
        if (offset_1 > maxRep) { savedOffset = offset_1; offset_1 = 0; }
    }
    row_fillHashCache(ms, mls, ms->nextToUpdate, ilimit);
    while (ip < ilimit) {
        size_t matchLength = 0, offset = 0; const u8* start = ip + 1; int storeNow = 0;
        if (offset_1 > 0 && zso_readLE32(ip + 1 - offset_1) == zso_readLE32(ip + 1)) {
            matchLength = zso_count(ip + 1 + 4, ip + 1 + 4 - offset_1, iend) + 4;
            if (depth == 0) storeNow = 1;
        }
        if (!storeNow) {
            {   size_t offsetFound = 999999999;
                size_t const ml2 = row_findBestMatch(ms, ip, iend, &offsetFound, mls, rowLog);
                if (ml2 > matchLength) { matchLength = ml2; start = ip; offset = offsetFound; }
            }
            if (matchLength < 4) { ip += ((ip - anchor) >> 8) + 1; continue; }
            if (depth >= 1) {
                while (ip < ilimit) {
                    ip++;
                    if (offset && offset_1 > 0 && zso_readLE32(ip) == zso_readLE32(ip - offset_1)) {
                        size_t const mlRep = zso_count(ip + 4, ip + 4 - offset_1, iend) + 4;
                        int const gain2 = (int)(mlRep * 3), gain1 = (int)(matchLength * 3 - zso_highbit32((u32)offset + 1) + 1);
                        if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offset = 0; start = ip; }
                    }
                    {   size_t offset2 = 999999999;
                        size_t const ml2 = row_findBestMatch(ms, ip, iend, &offset2, mls, rowLog);
                        int const gain2 = (int)(ml2 * 4 - zso_highbit32((u32)offset2 + 1)), gain1 = (int)(matchLength * 4 - zso_highbit32((u32)offset + 1) + 4);
                        if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offset = offset2; start = ip; continue; }
                    }
                    if (depth == 2 && ip < ilimit) {
                        ip++;
                        if (offset && offset_1 > 0 && zso_readLE32(ip) == zso_readLE32(ip - offset_1)) {
                            size_t const mlRep = zso_count(ip + 4, ip + 4 - offset_1, iend) + 4;
                            int const gain2 = (int)(mlRep * 4), gain1 = (int)(matchLength * 4 - zso_highbit32((u32)offset + 1) + 1);
                            if (mlRep >= 4 && gain2 > gain1) { matchLength = mlRep; offset = 0; start = ip; }
                        }
                        {   size_t offset2 = 999999999;
                            size_t const ml2 = row_findBestMatch(ms, ip, iend, &offset2, mls, rowLog);
                            int const gain2 = (int)(ml2 * 4 - zso_highbit32((u32)offset2 + 1)), gain1 = (int)(matchLength * 4 - zso_highbit32((u32)offset + 1) + 7);
                            if (ml2 >= 4 && gain2 > gain1) { matchLength = ml2; offset = offset2; start = ip; continue; }
                        }
                    }
                    break;
                }
            }
            if (offset) {       /* catch up */
                while (start > anchor && start - (offset - 2) > prefixLowest && start[-1] == (start - (offset - 2))[-1]) { start--; matchLength++; }
                offset_2 = offset_1; offset_1 = (u32)(offset - 2);
            }
        }
        /* _storeSequence */
        zso_storeSeq(ss, (size_t)(start - anchor), anchor, (u32)offset, matchLength - 3);
        anchor = ip = start + matchLength;
        while (ip <= ilimit && offset_2 > 0 && zso_readLE32(ip) == zso_readLE32(ip - offset_2)) {
            matchLength = zso_count(ip + 4, ip + 4 - offset_2, iend) + 4;
            offset = offset_2; offset_2 = offset_1; offset_1 = (u32)offset;
            zso_storeSeq(ss, 0, anchor, 0, matchLength - 3);
            ip += matchLength; anchor = ip;
        }
    }
    rep[0] = offset_1 ? offset_1 : savedOffset;
    rep[1] = offset_2 ? offset_2 : savedOffset;
    return (size_t)(iend - anchor);
}

/* ------------------------------------------------------------------ */
/*  frame                                                              */
/* ------------------------------------------------------------------ */
/* ZSTD_writeFrameHeader, U/ZstdCompress.cs:4817-4929 (contentSizeFlag = 1; dictID 0 = none) */
static size_t zso_writeFrameHeader(u8* op, size_t cap, u32 windowLog, u64 pledged, int checksumFlag, u32 dictID)
{
    u32 const windowSize = 1u << windowLog;
    u32 const single = windowSize >= pledged;
    u32 const fcsCode = (pledged >= 256) + (pledged >= 65536 + 256) + (pledged >= 0xFFFFFFFFu);
    u32 const didCode = (dictID > 0) + (dictID >= 256) + (dictID >= 65536);
    size_t pos = 0;
    if (cap < 18) return ZSO_ERR(dstSize_tooSmall);
    zso_writeLE32(op, ZSO_MAGIC); pos = 4;
    op[pos++] = (u8)(didCode + ((u32)(checksumFlag > 0) << 2) + (single << 5) + (fcsCode << 6));
    if (!single) op[pos++] = (u8)((windowLog - 10) << 3);
    switch (didCode) {
    case 1: op[pos++] = (u8)dictID; break;
    case 2: zso_writeLE16(op + pos, dictID); pos += 2; break;
    case 3: zso_writeLE32(op + pos, dictID); pos += 4; break;
    default: break;
    }
    switch (fcsCode) {
    case 0: if (single) op[pos++] = (u8)pledged; break;
    case 1: zso_writeLE16(op + pos, (u32)(pledged - 256)); pos += 2; break;
    case 2: zso_writeLE32(op + pos, (u32)pledged); pos += 4; break;
    default: zso_writeLE64(op + pos, pledged); pos += 8; break;
    }
    return pos;
}

static int zso_isRLE(const u8* src, size_t n) { size_t i; for (i = 1; i < n; i++) if (src[i] != src[0]) return 0; return 1; }

static void seqstore_alloc(zso_seqstore* ss, size_t blockSize)
{
    size_t const maxNbSeq = blockSize / 3 + 1;      /* the reference divides by 3 for minMatch 3, else 4 (U/ZstdCompress.cs:2570) */
    memset(ss, 0, sizeof *ss);
    ss->seqs = (zso_seq*)malloc(maxNbSeq * sizeof(zso_seq)); ss->cap = maxNbSeq;
    ss->lit = (u8*)malloc(blockSize + 64);
    ss->llCode = (u8*)malloc(maxNbSeq); ss->mlCode = (u8*)malloc(maxNbSeq); ss->ofCode = (u8*)malloc(maxNbSeq);
}
static void seqstore_free(zso_seqstore* ss) { free(ss->seqs); free(ss->lit); free(ss->llCode); free(ss->mlCode); free(ss->ofCode); }
static void seqstore_reset(zso_seqstore* ss) { ss->nbSeq = 0; ss->litSize = 0; ss->longLengthType = 0; ss->longLengthPos = 0; }

static size_t zso_blockCompressor(zso_mstate* ms, zso_seqstore* ss, u32 rep[3], const u8* src, size_t srcSize)
{
    /* fast, doubleFast and greedy/lazy over the row-hash finder are restated; anything else is refused by the callers */
    if (ms->cp.strategy == ZSO_dfast) return zso_compressBlock_doubleFast(ms, ss, rep, src, srcSize);
    if (ms->cp.strategy == ZSO_greedy) return zso_compressBlock_lazy_row(ms, ss, rep, src, srcSize, 0);
    if (ms->cp.strategy == ZSO_lazy) return zso_compressBlock_lazy_row(ms, ss, rep, src, srcSize, 1);
    return zso_compressBlock_fast(ms, ss, rep, src, srcSize);
}

/* prefixLen > 0: the `prefixLen` bytes in front of `src` are a raw-content dictionary, loaded the way
 * ZSTD_loadDictionaryContent does for the fast strategy (window extended over it, ZSTD_fillHashTable with dtlm_fast,
 * U/ZstdCompress.cs:5126-5237, U/ZstdFast.cs:9-46) with the input contiguous behind it, so that the noDict block
 * compressor sees it as plain history.  This is the reference's prefix/raw-dictionary semantics, not the CDict
 * attach/copy path `ZSTD_CCtx_loadDictionary` + `ZSTD_compress2` would take (dictMatchState / extDict finders, which
 * this oracle does not restate): the frames are valid dictionary frames for DECODER tests; their bytes are
 * "parity unpinned" (the reference holds no dictionary fixtures — its tests train dictionaries at run time). */
/* HUF_readCTable, U/HufCompress.cs:237-290: weights -> code lengths -> canonical codes; *hasZeroWeights as the reference reports it */
static size_t zso_huf_readCTable(zso_huf_ct* ct, u32* maxSVPtr, const void* src, size_t srcSize, u32* hasZeroWeights)
{
    u8 weights[256]; u32 rankStats[13], nbSymbols = 0, tableLog = 0, n; u16 nbPerRank[14] = {0}, valPerRank[14] = {0};
    size_t const readSize = zso_huf_readStats(weights, &nbSymbols, &tableLog, rankStats, src, srcSize);
    if (zso_isError(readSize)) return readSize;
    if (tableLog > 12) return ZSO_ERR(tableLog_tooLarge);
    if (nbSymbols > *maxSVPtr + 1) return ZSO_ERR(maxSymbolValue_tooSmall);
    memset(ct->nbBits, 0, sizeof ct->nbBits); memset(ct->val, 0, sizeof ct->val);
    *hasZeroWeights = 0;
    for (n = 0; n < nbSymbols; n++) {
        u32 const w = weights[n];
        *hasZeroWeights |= (w == 0);
        ct->nbBits[n] = (u8)((tableLog + 1 - w) & -(int)(w != 0));
    }
    for (n = 0; n < nbSymbols; n++) nbPerRank[ct->nbBits[n]]++;
    valPerRank[tableLog + 1] = 0;
    {   u16 min = 0; u32 r;
        for (r = tableLog; r > 0; r--) { valPerRank[r] = min; min += nbPerRank[r]; min >>= 1; }
    }
    for (n = 0; n < nbSymbols; n++) ct->val[n] = valPerRank[ct->nbBits[n]]++;
    ct->tableLog = tableLog; ct->maxSV = nbSymbols - 1; ct->valid = 1;
    *maxSVPtr = nbSymbols - 1;
    return readSize;
}

/* ZSTD_dictNCountRepeat, U/ZstdCompress.cs:5239-5257 */
static int zso_dictNCountRepeat(const s16* norm, u32 dictMaxSV, u32 maxSV)
{
    u32 s;
    if (dictMaxSV < maxSV) return FSE_repeat_check;
    for (s = 0; s <= maxSV; s++) if (norm[s] == 0) return FSE_repeat_check;
    return FSE_repeat_valid;
}

/* ZSTD_loadCEntropy, U/ZstdCompress.cs:5259-5400.  Returns the size of the dictionary's header (content follows). */
static size_t zso_loadCEntropy(zso_bstate* bs, const u8* dict, size_t dictSize)
{
    const u8* p = dict + 8; const u8* const end = dict + dictSize;
    s16 ofNorm[64], mlNorm[64], llNorm[64]; u32 ofMax = 31, mlMax = 52, llMax = 35, log;
    bs->hufRepeat = HUF_repeat_check;
    {   u32 maxSV = 255, hasZero = 1;
        size_t const h = zso_huf_readCTable(&bs->huf, &maxSV, p, (size_t)(end - p), &hasZero);
        if (!hasZero) bs->hufRepeat = HUF_repeat_valid;
        if (zso_isError(h) || maxSV < 255) return ZSO_ERR(dictionary_corrupted);
        p += h;
    }
    {   size_t const h = zso_readNCount(ofNorm, &ofMax, &log, p, (size_t)(end - p));
        if (zso_isError(h) || log > 8) return ZSO_ERR(dictionary_corrupted);
        if (zso_isError(zso_fse_buildCTable(&bs->of, ofNorm, 31, log))) return ZSO_ERR(dictionary_corrupted);
        p += h;
    }
    {   size_t const h = zso_readNCount(mlNorm, &mlMax, &log, p, (size_t)(end - p));
        if (zso_isError(h) || log > 9) return ZSO_ERR(dictionary_corrupted);
        if (zso_isError(zso_fse_buildCTable(&bs->ml, mlNorm, mlMax, log))) return ZSO_ERR(dictionary_corrupted);
        bs->mlRepeat = zso_dictNCountRepeat(mlNorm, mlMax, 52);
        p += h;
    }
    {   size_t const h = zso_readNCount(llNorm, &llMax, &log, p, (size_t)(end - p));
        if (zso_isError(h) || log > 9) return ZSO_ERR(dictionary_corrupted);
        if (zso_isError(zso_fse_buildCTable(&bs->ll, llNorm, llMax, log))) return ZSO_ERR(dictionary_corrupted);
        bs->llRepeat = zso_dictNCountRepeat(llNorm, llMax, 35);
        p += h;
    }
    if (p + 12 > end) return ZSO_ERR(dictionary_corrupted);
    bs->rep[0] = zso_readLE32(p); bs->rep[1] = zso_readLE32(p + 4); bs->rep[2] = zso_readLE32(p + 8);
    p += 12;
    {   size_t const contentSize = (size_t)(end - p); u32 offcodeMax = 31, u;
        if (contentSize <= 0xFFFFFFFFu - (128u << 10)) offcodeMax = zso_highbit32((u32)contentSize + (128u << 10));
        bs->ofRepeat = zso_dictNCountRepeat(ofNorm, ofMax, offcodeMax < 31 ? offcodeMax : 31);
        for (u = 0; u < 3; u++) if (bs->rep[u] == 0 || bs->rep[u] > contentSize) return ZSO_ERR(dictionary_corrupted);
    }
    return (size_t)(p - dict);
}

/* entropyDict != NULL: a formatted dictionary's header (magic, dictID, tables, repcodes); its content is the prefix */
static size_t zso_compress_internal(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksumFlag,
                                    size_t prefixLen, const u8* entropyDict, size_t entropyDictSize)
{
    zso_cparams const cp = zso_getCParams(level, srcSize + prefixLen);
    u8* const ostart = (u8*)dst; u8* op = ostart; const u8* ip = (const u8*)src; size_t remaining = srcSize;
    size_t blockSize = (size_t)1 << cp.windowLog; size_t result;
    zso_mstate ms; zso_seqstore ss; zso_bstate *prev, *next; int isFirstBlock = 1, wroteBlock = 0;
    int const disableLit = (cp.strategy == ZSO_fast) && (cp.targetLength > 0);
    if (blockSize > ZSO_BLOCKSIZE_MAX) blockSize = ZSO_BLOCKSIZE_MAX;
    /* greedy/lazy use the row-hash finder only when windowLog > 14 (ZSTD_resolveRowMatchFinderMode, U/ZstdCompress.cs:221-252);
       the hash-chain finder of the small-window tiers is not restated: say so, never substitute */
    if (cp.strategy >= ZSO_greedy && cp.windowLog <= 14) return ZSO_ERR(parameter_unsupported);
    if (prefixLen && cp.strategy != ZSO_fast) return ZSO_ERR(parameter_unsupported);
    {   size_t const h = zso_writeFrameHeader(op, dstCapacity, cp.windowLog, srcSize, checksumFlag, entropyDict ? zso_readLE32(entropyDict + 4) : 0);
        if (zso_isError(h)) return h;
        op += h; dstCapacity -= h;
    }
    ms.cp = cp; ms.base = ip - prefixLen - 2; ms.dictLimit = ms.lowLimit = 2;
    ms.hashTable = (u32*)calloc((size_t)1 << cp.hashLog, sizeof(u32));
    ms.chainTable = (u32*)calloc((size_t)1 << cp.chainLog, sizeof(u32));
    ms.tagTable = (u16*)calloc((size_t)1 << cp.hashLog, sizeof(u16));
    {   u32 const sl = cp.searchLog, rowLog = sl < 4 ? 4 : (sl > 6 ? 6 : sl); ms.rowHashLog = cp.hashLog - rowLog; }
    ms.nextToUpdate = 2; memset(ms.hashCache, 0, sizeof ms.hashCache);
    if (prefixLen > 8) {            /* ZSTD_fillHashTable(ms, dictEnd, ZSTD_dtlm_fast) */
        const u8* p = ms.base + ms.nextToUpdate; const u8* const fend = ip - 8;
        for (; p + 3 < fend + 2; p += 3) ms.hashTable[zso_hashPtr(p, cp.hashLog, cp.minMatch)] = (u32)(p - ms.base);
        ms.nextToUpdate = (u32)(ip - ms.base);
    }
    seqstore_alloc(&ss, blockSize);
    prev = (zso_bstate*)malloc(sizeof *prev); next = (zso_bstate*)malloc(sizeof *next);
    bstate_reset(prev); bstate_reset(next);
    if (entropyDict) {
        size_t const e = zso_loadCEntropy(prev, entropyDict, entropyDictSize);
        if (zso_isError(e)) { result = e; goto done; }
    }
    while (remaining) {             /* ZSTD_compress_frameChunk */
        u32 const lastBlock = blockSize >= remaining;
        u32 const maxDist = 1u << cp.windowLog;
        size_t cSize;
        if (dstCapacity < 3 + 3) { result = ZSO_ERR(dstSize_tooSmall); goto done; }
        if (remaining < blockSize) blockSize = remaining;
        {   /* ZSTD_window_enforceMaxDist(window, ip, maxDist) — note: evaluated at the block START */
            u32 const blockIdx = (u32)(ip - ms.base);
            if (blockIdx > maxDist) {
                u32 const newLow = blockIdx - maxDist;
                if (ms.lowLimit < newLow) ms.lowLimit = newLow;
                if (ms.dictLimit < ms.lowLimit) ms.dictLimit = ms.lowLimit;
            }
        }
        if (ms.nextToUpdate < ms.lowLimit) ms.nextToUpdate = ms.lowLimit;
        /* ZSTD_compressBlock_internal */
        if (blockSize < 1 + 1 + 1 + 3 + 1) cSize = 0;          /* ZSTD_buildSeqStore: too small, don't even try */
        else {
            size_t lastLL; int i;
            seqstore_reset(&ss);
            {   u32 const curr = (u32)(ip - ms.base);     /* ZSTD_buildSeqStore: limit catch-up after a long uncompressed stretch */
                if (curr > ms.nextToUpdate + 384) { u32 const d = curr - ms.nextToUpdate - 384; ms.nextToUpdate = curr - (192 < d ? 192 : d); }
            }
            for (i = 0; i < 3; i++) next->rep[i] = prev->rep[i];
            lastLL = zso_blockCompressor(&ms, &ss, next->rep, ip, blockSize);
            memcpy(ss.lit + ss.litSize, ip + blockSize - lastLL, lastLL); ss.litSize += lastLL;
            cSize = zso_entropyCompressSeqStore(&ss, prev, next, cp.strategy, disableLit, op + 3, dstCapacity - 3, blockSize);
            if (!isFirstBlock && cSize < 25 && zso_isRLE(ip, blockSize)) { cSize = 1; op[3] = ip[0]; }
        }
        if (!zso_isError(cSize) && cSize > 1) { zso_bstate* t = prev; prev = next; next = t; }   /* confirm repcodes + entropy tables */
        if (prev->ofRepeat == FSE_repeat_valid) prev->ofRepeat = FSE_repeat_check;
        if (zso_isError(cSize)) { result = cSize; goto done; }
        if (cSize == 0) {           /* ZSTD_noCompressBlock */
            if (blockSize + 3 > dstCapacity) { result = ZSO_ERR(dstSize_tooSmall); goto done; }
            zso_writeLE24(op, lastBlock + (0u << 1) + (u32)(blockSize << 3));
            memcpy(op + 3, ip, blockSize);
            cSize = 3 + blockSize;
        } else {
            u32 const h = cSize == 1 ? lastBlock + (1u << 1) + (u32)(blockSize << 3) : lastBlock + (2u << 1) + (u32)(cSize << 3);
            zso_writeLE24(op, h);
            cSize += 3;
        }
        ip += blockSize; remaining -= blockSize; op += cSize; dstCapacity -= cSize;
        isFirstBlock = 0; wroteBlock = 1;
    }
    /* ZSTD_writeEpilogue */
    if (!wroteBlock) {
        if (dstCapacity < 4) { result = ZSO_ERR(dstSize_tooSmall); goto done; }
        zso_writeLE24(op, 1); op += 3; dstCapacity -= 3;
    }
    if (checksumFlag) {
        if (dstCapacity < 4) { result = ZSO_ERR(dstSize_tooSmall); goto done; }
        zso_writeLE32(op, (u32)zso_xxh64(src, srcSize, 0)); op += 4;
    }
    result = (size_t)(op - ostart);
done:
    free(ms.hashTable); free(ms.chainTable); free(ms.tagTable); seqstore_free(&ss); free(prev); free(next);
    return result;
}

size_t zso_compress(void* dst, size_t dstCapacity, const void* src, size_t srcSize, int level, int checksumFlag)
{
    return zso_compress_internal(dst, dstCapacity, src, srcSize, level, checksumFlag, 0, NULL, 0);
}

size_t zso_compress_usingDict(void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                              const void* dict, size_t dictSize, int level, int checksumFlag)
{
    u8* buf; size_t r;
    if (dict == NULL || dictSize < 8) return zso_compress(dst, dstCapacity, src, srcSize, level, checksumFlag);   /* :5469-5477 */
    buf = (u8*)malloc(dictSize + srcSize + 8);
    if (!buf) return ZSO_ERR(memory_allocation);
    memcpy(buf, dict, dictSize); memcpy(buf + dictSize, src, srcSize); memset(buf + dictSize + srcSize, 0, 8);
    if (zso_readLE32(dict) == 0xEC30A437u) {
        /* formatted dictionary (ZSTD_loadZstdDictionary, :5402-5463): the header's tables and repcodes become the previous
           block state, the content behind it the prefix */
        zso_bstate* probe = (zso_bstate*)malloc(sizeof *probe);
        size_t e;
        if (!probe) { free(buf); return ZSO_ERR(memory_allocation); }
        bstate_reset(probe);
        e = zso_loadCEntropy(probe, (const u8*)dict, dictSize);
        free(probe);
        if (zso_isError(e)) { free(buf); return e; }
        r = zso_compress_internal(dst, dstCapacity, buf + dictSize, srcSize, level, checksumFlag, dictSize - e, buf, dictSize);
    } else
        r = zso_compress_internal(dst, dstCapacity, buf + dictSize, srcSize, level, checksumFlag, dictSize, NULL, 0);
    free(buf);
    return r;
}

size_t zso_compress_chunked(void* dst, size_t dstCapacity, const void* src, size_t srcSize,
                            int level, int checksumFlag, size_t chunkSize)
{
    const u8* ip = (const u8*)src; u8* op = (u8*)dst; size_t remaining = srcSize;
    if (srcSize == 0) return zso_compress(dst, dstCapacity, src, 0, level, checksumFlag);
    while (remaining) {
        size_t const n = remaining < chunkSize ? remaining : chunkSize;
        size_t const c = zso_compress(op, dstCapacity, ip, n, level, checksumFlag);
        if (zso_isError(c)) return c;
        op += c; dstCapacity -= c; ip += n; remaining -= n;
    }
    return (size_t)(op - (u8*)dst);
}

/* A formatted dictionary for TESTS: magic, dictID, Huffman table, offset / match-length / literal-length NCounts, repcodes
 * {1,4,8}, content (the layout ZSTD_loadZstdDictionary parses, U/ZstdCompress.cs:5402-5463).  The statistics are those of
 * the sample's level-1 parse with every symbol counted once more, so that every table covers its whole alphabet — a
 * stand-in for the reference's trainer (ZDICT_finalizeDictionary, U/Zdict.cs), which is out of scope. */
size_t zso_make_dictionary(void* dst, size_t cap, const void* content, size_t contentSize,
                           const void* sample, size_t sampleSize, u32 dictID)
{
    u8* const ostart = (u8*)dst; u8* op = ostart; u8* const oend = ostart + cap;
    u32 litCount[256], ofCount[32], mlCount[64], llCount[64], n, maxSV;
    zso_seq* seqs; u8* lits; size_t litSize = 0, nbSeq;
    if (cap < 8 + 512 + 12 + contentSize || contentSize < 8 || sampleSize < 16 || sampleSize > ZSO_BLOCKSIZE_MAX) return ZSO_ERR(GENERIC);
    seqs = (zso_seq*)malloc((sampleSize / 3 + 2) * sizeof(zso_seq)); lits = (u8*)malloc(sampleSize + 64);
    if (!seqs || !lits) { free(seqs); free(lits); return ZSO_ERR(memory_allocation); }
    nbSeq = zso_block_sequences(seqs, sampleSize / 3 + 1, lits, &litSize, sample, sampleSize, 1);
    for (n = 0; n < 256; n++) litCount[n] = 1;
    for (n = 0; n < 32; n++) ofCount[n] = 1;
    for (n = 0; n < 64; n++) { mlCount[n] = 1; llCount[n] = 1; }
    for (n = 0; n < litSize; n++) litCount[lits[n]]++;
    for (n = 0; n < nbSeq; n++) {
        ofCount[zso_highbit32(seqs[n].offBase) & 31]++;
        mlCount[zso_MLcode(seqs[n].mlBase)]++;
        llCount[zso_LLcode(seqs[n].litLength)]++;
    }
    free(seqs); free(lits);
    zso_writeLE32(op, 0xEC30A437u); zso_writeLE32(op + 4, dictID); op += 8;
    {   zso_huf_ct ct; size_t h;
        size_t const e = huf_buildCTable(&ct, litCount, 255, 11);
        if (zso_isError(e)) return e;
        h = huf_writeCTable(op, (size_t)(oend - op), &ct, 255, ct.tableLog);
        if (zso_isError(h)) return h;
        op += h;
    }
    {   struct { u32* count; u32 max, log; } const t[3] = { { ofCount, 30, 8 }, { mlCount, 52, 9 }, { llCount, 35, 9 } };
        int k;
        for (k = 0; k < 3; k++) {
            s16 norm[64]; size_t total = 0, e, h;
            for (n = 0; n <= t[k].max; n++) total += t[k].count[n];
            maxSV = t[k].max;
            e = zso_fse_normalizeCount(norm, t[k].log, t[k].count, total, maxSV, 0);
            if (zso_isError(e)) return e;
            h = zso_fse_writeNCount(op, (size_t)(oend - op), norm, maxSV, t[k].log);
            if (zso_isError(h)) return h;
            op += h;
        }
    }
    zso_writeLE32(op, 1); zso_writeLE32(op + 4, 4); zso_writeLE32(op + 8, 8); op += 12;
    if ((size_t)(oend - op) < contentSize) return ZSO_ERR(dstSize_tooSmall);
    memcpy(op, content, contentSize); op += contentSize;
    return (size_t)(op - ostart);
}

/* ---------- stage hooks ---------- */
size_t zso_block_sequences(zso_seq* seqs, size_t seqCap, u8* lits, size_t* litSizePtr,
                           const void* src, size_t srcSize, int level)
{
    zso_cparams const cp = zso_getCParams(level, srcSize);
    zso_mstate ms; zso_seqstore ss; u32 rep[3] = { 1, 4, 8 }; size_t lastLL, n;
    ms.cp = cp; ms.base = (const u8*)src - 2; ms.dictLimit = ms.lowLimit = 2;
    ms.hashTable = (u32*)calloc((size_t)1 << cp.hashLog, sizeof(u32));
    ms.chainTable = (u32*)calloc((size_t)1 << cp.chainLog, sizeof(u32));
    ms.tagTable = (u16*)calloc((size_t)1 << cp.hashLog, sizeof(u16));
    {   u32 const sl = cp.searchLog, rowLog = sl < 4 ? 4 : (sl > 6 ? 6 : sl); ms.rowHashLog = cp.hashLog - rowLog; }
    ms.nextToUpdate = 2; memset(ms.hashCache, 0, sizeof ms.hashCache);
    seqstore_alloc(&ss, srcSize + 8);
    lastLL = srcSize < 8 ? srcSize : zso_blockCompressor(&ms, &ss, rep, (const u8*)src, srcSize);
    memcpy(ss.lit + ss.litSize, (const u8*)src + srcSize - lastLL, lastLL); ss.litSize += lastLL;
    n = ss.nbSeq < seqCap ? ss.nbSeq : seqCap;
    memcpy(seqs, ss.seqs, n * sizeof(zso_seq));
    memcpy(lits, ss.lit, ss.litSize); *litSizePtr = ss.litSize;
    n = ss.nbSeq;
    free(ms.hashTable); free(ms.chainTable); free(ms.tagTable); seqstore_free(&ss);
    return n;
}

size_t zso_entropy_block(void* dst, size_t dstCapacity, const zso_seq* seqs, size_t nbSeq,
                         const u8* lits, size_t litSize, u32 strategy, size_t srcSize)
{
    zso_seqstore ss; zso_bstate *prev = (zso_bstate*)malloc(sizeof *prev), *next = (zso_bstate*)malloc(sizeof *next);
    size_t r;
    seqstore_alloc(&ss, (litSize > nbSeq * 3 ? litSize : nbSeq * 3) + 8);
    bstate_reset(prev); bstate_reset(next);
    memcpy(ss.seqs, seqs, nbSeq * sizeof(zso_seq)); ss.nbSeq = nbSeq;
    memcpy(ss.lit, lits, litSize); ss.litSize = litSize;
    r = zso_entropyCompressSeqStore(&ss, prev, next, strategy, 0, dst, dstCapacity, srcSize);
    seqstore_free(&ss); free(prev); free(next);
    return r;
}
