// zmi_fse.h — device-side FSE (tANS) table construction: the short serial sections run by ONE lane per block, the table itself by the wave.
//
// These are the per-block serial sections of the entropy stage: a few dozen symbols
// and tables of <= 512 states.  They follow the reference step for step so that the emitted headers and tables
// are bit-identical to what ZstdSharp writes for the same histogram:
//   FSE_optimalTableLog_internal  U/FseCompress.cs:397-430
//   FSE_normalizeCount / M2       U/FseCompress.cs:443-665 (rtbTable U/Arrays.cs:8)
//   FSE_writeNCount_generic       U/FseCompress.cs:203-336
//   FSE_buildCTable_wksp          U/FseCompress.cs:13-191
//   FSE_initCState2/encodeSymbol  U/Fse.cs:26-57
#pragma once
#include "zmi_device.h"

namespace zmi {

struct SymTT { s32 deltaFindState; u32 deltaNbBits; };

__device__ __forceinline__ u32 fse_min_table_log(u32 srcSize, u32 maxSV)
{
    const u32 a = highbit32(srcSize) + 1, b = highbit32(maxSV) + 2;
    return a < b ? a : b;
}
__device__ __forceinline__ u32 fse_optimal_table_log(u32 maxTableLog, u32 srcSize, u32 maxSV, u32 minus)
{
    const u32 maxBitsSrc = highbit32(srcSize - 1) - minus;
    u32 tableLog = maxTableLog;
    const u32 minBits = fse_min_table_log(srcSize, maxSV);
    if (tableLog == 0) tableLog = 11;
    if (maxBitsSrc < tableLog) tableLog = maxBitsSrc;
    if (minBits > tableLog) tableLog = minBits;
    if (tableLog < 5) tableLog = 5;
    if (tableLog > 12) tableLog = 12;
    return tableLog;
}

// returns false on the (unreachable for valid input) error paths
__device__ inline bool fse_normalize_m2(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSV, s16 lowProbCount)
{
    const s16 NOT_YET = -2;
    u32 distributed = 0, ToDistribute;
    const u32 lowThreshold = total >> tableLog;
    u32 lowOne = (u32)(((u64)total * 3) >> (tableLog + 1));
    for (u32 s = 0; s <= maxSV; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; distributed++; total -= count[s]; continue; }
        if (count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; continue; }
        norm[s] = NOT_YET;
    }
    ToDistribute = (1u << tableLog) - distributed;
    if (ToDistribute == 0) return true;
    if ((total / ToDistribute) > lowOne) {
        lowOne = (u32)(((u64)total * 3) / (ToDistribute * 2));
        for (u32 s = 0; s <= maxSV; s++)
            if (norm[s] == NOT_YET && count[s] <= lowOne) { norm[s] = 1; distributed++; total -= count[s]; }
        ToDistribute = (1u << tableLog) - distributed;
    }
    if (distributed == maxSV + 1) {
        u32 maxV = 0, maxC = 0;
        for (u32 s = 0; s <= maxSV; s++) if (count[s] > maxC) { maxV = s; maxC = count[s]; }
        norm[maxV] += (s16)ToDistribute;
        return true;
    }
    if (total == 0) {
        for (u32 s = 0; ToDistribute > 0; s = (s + 1) % (maxSV + 1)) if (norm[s] > 0) { ToDistribute--; norm[s]++; }
        return true;
    }
    {
        const u64 vStepLog = 62 - tableLog, mid = (1ull << (vStepLog - 1)) - 1;
        const u64 rStep = (((1ull << vStepLog) * ToDistribute) + mid) / total;
        u64 tmpTotal = mid;
        for (u32 s = 0; s <= maxSV; s++) {
            if (norm[s] == NOT_YET) {
                const u64 end = tmpTotal + (count[s] * rStep);
                const u32 sStart = (u32)(tmpTotal >> vStepLog), sEnd = (u32)(end >> vStepLog), weight = sEnd - sStart;
                if (weight < 1) return false;
                norm[s] = (s16)weight; tmpTotal = end;
            }
        }
    }
    return true;
}

__device__ inline bool fse_normalize_count(s16* norm, u32 tableLog, const u32* count, u32 total, u32 maxSV, u32 useLowProbCount)
{
    const u32 rtb[8] = { 0, 473195, 504333, 520860, 550000, 700000, 750000, 830000 };
    const s16 lowProbCount = useLowProbCount ? -1 : 1;
    const u64 scale = 62 - tableLog, step = (1ull << 62) / total, vStep = 1ull << (scale - 20);
    int stillToDistribute = 1 << tableLog;
    u32 largest = 0; s16 largestP = 0;
    const u32 lowThreshold = total >> tableLog;
    for (u32 s = 0; s <= maxSV; s++) {
        if (count[s] == 0) { norm[s] = 0; continue; }
        if (count[s] <= lowThreshold) { norm[s] = lowProbCount; stillToDistribute--; }
        else {
            s16 proba = (s16)((count[s] * step) >> scale);
            if (proba < 8) {
                const u64 restToBeat = vStep * rtb[proba];
                proba += (count[s] * step) - ((u64)proba << scale) > restToBeat;
            }
            if (proba > largestP) { largestP = proba; largest = s; }
            norm[s] = proba; stillToDistribute -= proba;
        }
    }
    if (-stillToDistribute >= (norm[largest] >> 1)) return fse_normalize_m2(norm, tableLog, count, total, maxSV, lowProbCount);
    norm[largest] += (s16)stillToDistribute;
    return true;
}

// returns the header size in bytes (0 on the unreachable error path); `out` needs 2 bytes of slack
__device__ inline u32 fse_write_ncount(u8* out, const s16* norm, u32 maxSV, u32 tableLog)
{
    u8* const ostart = out;
    int nbBits = (int)tableLog + 1, remaining = (1 << tableLog) + 1, threshold = 1 << tableLog;
    u32 bitStream = 0; int bitCount = 0; u32 symbol = 0; const u32 alphabetSize = maxSV + 1; int previousIs0 = 0;
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
        {
            int count = norm[symbol++];
            const int max = (2 * threshold - 1) - remaining;
            remaining -= count < 0 ? -count : count;
            count++;
            if (count >= threshold) count += max;
            bitStream += (u32)count << bitCount;
            bitCount += nbBits; bitCount -= (count < max);
            previousIs0 = (count == 1);
            if (remaining < 1) return 0;
            while (remaining < threshold) { nbBits--; threshold >>= 1; }
        }
        if (bitCount > 16) { out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += 2; bitStream >>= 16; bitCount -= 16; }
    }
    if (remaining != 1) return 0;
    out[0] = (u8)bitStream; out[1] = (u8)(bitStream >> 8); out += (bitCount + 7) / 8;
    return (u32)(out - ostart);
}

// FSE_buildCTable_wksp (U/FseCompress.cs:20-160) by the 64 lanes of the chunk's wave; lane s stands for symbol s.  Same
// restatement as the decoder's table build (decode.hip): low-probability symbols take the top cells in symbol order; the
// reference's spreading visits (i*step) & mask for i = 0, 1, ... and skips cells above highThreshold, so the j-th cell it
// keeps belongs to the symbol whose cumulative count covers j; `stateTable[cumul[s]++] = tableSize + u` in cell order is a
// ballot rank per symbol with the counter kept in that symbol's lane; the per-symbol transforms are independent.
__device__ __forceinline__ void fse_build_ctable_wave(u16* stateTable, SymTT* tt, const s16* norm, u32 maxSV, u32 tableLog,
                                                      u16* cumR, u8* tableSymbol, u32 lane)
{
    const u32 tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    const int nc = lane <= maxSV ? (int)norm[lane] : 0;
    const bool low = nc == -1;
    const u64 lowMask = ballot(low);
    const u32 highThreshold = tableSize - 1 - popc64(lowMask);
    if (low) tableSymbol[tableSize - 1 - popc64(lowMask & lanemask_lt())] = (u8)lane;
    const u32 cntAll = low ? 1u : (nc > 0 ? (u32)nc : 0u), cntReg = nc > 0 ? (u32)nc : 0u;
    const u32 inclAll = wave_scan_incl(cntAll), inclReg = wave_scan_incl(cntReg);
    const u32 total = inclAll - cntAll;                    // cumul[s]: first state-table slot of symbol s
    cumR[lane] = (u16)(inclReg - cntReg);
    wave_lds_sync();
    u32 jBase = 0;
    for (u32 i0 = 0; i0 < tableSize; i0 += 64) {
        const u32 i = i0 + lane, p = (i * step) & mask;
        const bool place = i < tableSize && p <= highThreshold;
        const u64 bal = ballot(place);
        const u32 j = jBase + popc64(bal & lanemask_lt());
        jBase += popc64(bal);
        if (place) {
            u32 lo = 0, hi = 63;
#pragma unroll
            for (u32 it = 0; it < 6; ++it) { const u32 mid = (lo + hi + 1) >> 1; if (cumR[mid] <= j) lo = mid; else hi = mid - 1; }
            tableSymbol[p] = (u8)lo;
        }
    }
    wave_lds_sync();
    u32 nxt = total;                                       // cumul[lane], advanced as cells of symbol `lane` are met
    for (u32 u0 = 0; u0 < tableSize; u0 += 64) {
        const u32 u = u0 + lane; const bool valid = u < tableSize;
        const u32 sym = valid ? tableSymbol[u] : 0xFFFFu;
        u64 rem = ballot(valid);
        while (rem) {
            const u32 s0 = read_lane(sym, ctz64(rem));
            const u64 m = ballot(sym == s0);
            const u32 baseN = read_lane(nxt, s0);
            if (sym == s0) stateTable[baseN + popc64(m & lanemask_lt())] = (u16)(tableSize + u);
            nxt = lane == s0 ? nxt + popc64(m) : nxt;
            rem &= ~m;
        }
    }
    if (lane <= maxSV) {
        SymTT t;
        if (nc == 0) { t.deltaNbBits = ((tableLog + 1) << 16) - (1u << tableLog); t.deltaFindState = 0; }
        else if (nc == -1 || nc == 1) { t.deltaNbBits = (tableLog << 16) - (1u << tableLog); t.deltaFindState = (s32)(total - 1); }
        else {
            const u32 maxBitsOut = tableLog - highbit32((u32)nc - 1);
            t.deltaNbBits = (maxBitsOut << 16) - ((u32)nc << maxBitsOut);
            t.deltaFindState = (s32)(total - (u32)nc);
        }
        tt[lane] = t;
    }
    wave_lds_sync();
}

__device__ __forceinline__ u32 fse_init_state2(const u16* stateTable, const SymTT* tt, u32 symbol)
{
    const SymTT t = tt[symbol];
    const u32 nbBitsOut = (t.deltaNbBits + (1u << 15)) >> 16;
    const u32 v = (nbBitsOut << 16) - t.deltaNbBits;
    return stateTable[(s32)(v >> nbBitsOut) + t.deltaFindState];
}

} // namespace zmi
