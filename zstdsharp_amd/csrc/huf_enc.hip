// huf_enc.hip — literals section of a block on gfx950.
//
//   huf_hist_kernel   : one 256-thread workgroup per chunk.  Four waves build the four per-stream byte histograms in
//                       LDS (HIST_count, U/Hist.cs:67-166; 16-byte loads, parity-split copies, the hottest byte
//                       counted by ballot), take the compressible / RLE / raw verdict of HUF_compress_internal
//                       (U/HufCompress.cs:1360-1543) and run HUF_sort (:520-680: parallel bucket placement, the log2
//                       buckets quick-sorted on separate lanes exactly as the reference does).  Sorted leaves and
//                       histograms go to the next kernel through the chunk's output slot.
//   huf_tree_kernel   : one wave per chunk (5 KiB of LDS, ~28 chunks per CU).  The serial constructions, exactly as
//                       the reference: HUF_buildTree / HUF_setMaxHeight / HUF_buildCTableFromTree (:377-823), the tree
//                       description (HUF_writeCTable_wksp + HUF_compressWeights, :40-235) and every raw / RLE /
//                       compressed decision of ZSTD_compressLiterals (U/ZstdCompressLiterals.cs:86-185).  Because
//                       stream sizes follow from histogram x code length, the whole literals section is sized before
//                       a byte is encoded.
//   huf_encode_kernel : one 256-thread workgroup per chunk, wave w encodes stream w
//                       (HUF_compress4X_usingCTable_internal, U/HufCompress.cs:1221-1321): symbols are taken last
//                       to first, 8 per lane, their codes concatenated in registers, bit offsets come from a wave
//                       prefix scan, and the lanes OR their bits into an LDS tile that is flushed with coalesced
//                       dword stores.
// Given the same literals and sequence count, the bytes produced equal the oracle's
// (tests/test_gpu_parity.py::test_entropy_stage_is_byte_identical_to_oracle).
#include "zmi_device.h"
#include "zmi_fse.h"

namespace zmi {

#ifdef ZMI_LZ_STAMPS
__device__ unsigned long long g_hufStamps[16];
#define ZMI_HSTAMP(i) do { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stampAcc[i] += now_ - stampLast; stampLast = now_; } } while (0)
#else
#define ZMI_HSTAMP(i) do { } while (0)
#endif

struct Node { u32 count; u16 parent; u8 byte; u8 nbBits; };

// hand-off from huf_hist_kernel to huf_tree_kernel, kept in the chunk's output slot (overwritten later by huf_encode)
struct HufWork {
    u32 leafCount[256];         // HUF_sort's result: counts in descending order ...
    u8  leafByte[256];          // ... and their symbols
    u16 hist[4][256];           // per-stream histograms
    u32 flags[4];               // compressible?, maxSymbolValue, RLE?, RLE byte
};
static_assert(sizeof(HufWork) <= kSlotStride, "hand-off fits the slot");

struct HufTreeLds {
    Node nodes[513];
    u8  nbBits[256];
    u8  weights[256];
    u32 wcount[13]; s16 wnorm[13]; u16 wstate[64]; SymTT wtt[13]; u16 wcumul[15]; u8 wsym[64];
    u32 valPerRank[13];
    u32 rankLast[14];
    u32 sh[8];
};

struct HufBuildLds {
    u32 hist[4][2][256];         // two copies per wave (more copies cost more in residency than they save in same-address atomics: DESIGN 5)
    u32 sample[2][256];
    u32 count[256];
    Node nodes[513];
    u16 rankBase[192], rankCurr[192];
    u32 sampleMax[2];
    u32 redMaxSV[4], redLargest[4];
    u8  sortIdx[256];
    s16 qsStack[26][64];        // one explicit quicksort stack per log2 bucket (buckets are sorted by separate lanes); values -1..256
    u32 sh[8];                  // decisions shared by the workgroup: see enum below
};
enum { kShCompressed = 0, kShMaxSV, kShRle, kShRleByte, kShHuffLog, kShNonNull, kShRoot, kShHSize };

// ---- HUF_sort (U/HufCompress.cs:520-680): bucket sort by count, quicksort inside the log2 buckets ----
__device__ __forceinline__ u32 huf_get_index(u32 count) { return count < 165 ? count : highbit32(count) + 158; }

__device__ inline void huf_insertion_sort(Node* a, int low, int high)
{
    const int size = high - low + 1; a += low;
    for (int i = 1; i < size; i++) {
        const Node key = a[i]; int j = i - 1;
        while (j >= 0 && a[j].count < key.count) { a[j + 1] = a[j]; j--; }
        a[j + 1] = key;
    }
}
__device__ inline int huf_partition(Node* a, int low, int high)
{
    const u32 pivot = a[high].count; int i = low - 1;
    for (int j = low; j < high; j++) if (a[j].count > pivot) { i++; const Node t = a[i]; a[i] = a[j]; a[j] = t; }
    { const Node t = a[i + 1]; a[i + 1] = a[high]; a[high] = t; }
    return i + 1;
}
// HUF_simpleQuickSort: a call checks the insertion-sort threshold once, then partitions in a loop, recursing
// (threshold checked again) into the smaller side and continuing the loop (threshold NOT checked) on the larger.
// Sub-ranges are disjoint, so an explicit stack of {low, high, isCall} reproduces the result exactly.
__device__ inline void huf_quick_sort(Node* a, int low0, int high0, s16* stack)
{
    int sp = 0;
    stack[sp++] = (s16)low0; stack[sp++] = (s16)high0; stack[sp++] = 1;
    while (sp) {
        const int isCall = stack[--sp]; const int high = stack[--sp]; const int low = stack[--sp];
        if (isCall && high - low < 8) { huf_insertion_sort(a, low, high); continue; }
        if (!(low < high)) continue;
        const int idx = huf_partition(a, low, high);
        if (idx - low < high - idx) {
            stack[sp++] = (s16)(idx + 1); stack[sp++] = (s16)high;      stack[sp++] = 0;
            stack[sp++] = (s16)low;       stack[sp++] = (s16)(idx - 1); stack[sp++] = 1;
        } else {
            stack[sp++] = (s16)low;       stack[sp++] = (s16)(idx - 1); stack[sp++] = 0;
            stack[sp++] = (s16)(idx + 1); stack[sp++] = (s16)high;      stack[sp++] = 1;
        }
    }
}

// the log2 buckets (extents computed by the kernel's parallel placement) are sorted on separate lanes
__device__ inline void huf_sort_bucket(HufBuildLds& L, u32 b /* 0..25 */)
{
    const u32 n = 165 + b;
    const u32 bucketSize = L.rankCurr[n] - L.rankBase[n], bucketStart = L.rankBase[n];
    if (bucketSize > 1) huf_quick_sort(L.nodes + 1 + bucketStart, 0, (int)bucketSize - 1, L.qsStack[b]);
}

// ---- HUF_buildTree (U/HufCompress.cs:689-738) ----
__device__ inline int huf_build_tree(Node* huffNode, u32 maxSV, int* rootOut)
{
    Node* const huffNode0 = huffNode - 1;
    int nonNullRank = (int)maxSV, lowS, lowN, nodeNb = 256, nodeRoot;
    while (huffNode[nonNullRank].count == 0) nonNullRank--;
    lowS = nonNullRank; nodeRoot = nodeNb + lowS - 1; lowN = nodeNb;
    huffNode[nodeNb].count = huffNode[lowS].count + huffNode[lowS - 1].count;
    huffNode[lowS].parent = huffNode[lowS - 1].parent = (u16)nodeNb;
    nodeNb++; lowS -= 2;
    for (int n = nodeNb; n <= nodeRoot; n++) huffNode[n].count = 1u << 30;
    huffNode0[0].count = 1u << 31;
    // two-queue merge; the heads of both queues are kept in registers so that each pick costs one LDS read
    u32 cS = huffNode[lowS].count, cN = huffNode[lowN].count;
    while (nodeNb <= nodeRoot) {
        int n1, n2; u32 c1, c2;
        // (a head equal to nodeNb reads the 1<<30 placeholder, exactly as the reference's second comparison does)
        if (cS < cN) { n1 = lowS--; c1 = cS; cS = huffNode[lowS].count; } else { n1 = lowN++; c1 = cN; cN = huffNode[lowN].count; }
        if (cS < cN) { n2 = lowS--; c2 = cS; cS = huffNode[lowS].count; } else { n2 = lowN++; c2 = cN; cN = huffNode[lowN].count; }
        huffNode[nodeNb].count = c1 + c2;
        if (lowN == nodeNb) cN = c1 + c2;          // the node just created is the next head of the node queue
        huffNode[n1].parent = huffNode[n2].parent = (u16)nodeNb;
        nodeNb++;
    }
    // code lengths (the reference's two top-down loops) are computed by the caller, one leaf per lane
    *rootOut = nodeRoot;
    return nonNullRank;
}

// ---- HUF_setMaxHeight (U/HufCompress.cs:377-514), called by all 64 lanes of the chunk's wave ----
// The three scans of the reference (clamp the over-long tail and total its cost; skip the run already at maxNbBits; find
// the last symbol of every shorter length) are evaluated with ballots over the four positions each lane holds; only the
// repayment loops, whose steps depend on each other, run on lane 0.  rankLast lives in LDS (it is indexed dynamically).
template <class LDS>
__device__ inline u32 huf_set_max_height_wave(LDS& L, Node* huffNode, u32 lastNonNull, u32 maxNbBits)
{
    const u32 lane = lane_id();
    const u32 largestBits = huffNode[lastNonNull].nbBits;
    if (largestBits <= maxNbBits) return largestBits;
    const u32 noSymbol = 0xF0F0F0F0u;
    u32 nb[4];
    for (u32 k = 0; k < 4; ++k) { const u32 pos = k * 64 + lane; nb[k] = pos <= lastNonNull ? huffNode[pos].nbBits : 0xFFu; }
    auto highest = [&](bool p0, bool p1, bool p2, bool p3) -> int {       // highest position whose predicate holds, -1 if none
        const u64 b3 = ballot(p3), b2 = ballot(p2), b1 = ballot(p1), b0 = ballot(p0);
        if (b3) return 192 + 63 - __builtin_clzll(b3);
        if (b2) return 128 + 63 - __builtin_clzll(b2);
        if (b1) return 64 + 63 - __builtin_clzll(b1);
        if (b0) return 63 - __builtin_clzll(b0);
        return -1;
    };
    // loop 1: the tail of positions above n1 is longer than allowed
    const int n1 = highest(nb[0] <= maxNbBits, nb[1] <= maxNbBits, nb[2] <= maxNbBits, nb[3] <= maxNbBits);
    const u32 baseCost = 1u << (largestBits - maxNbBits);
    int cost = 0;
    for (u32 k = 0; k < 4; ++k) {
        const int pos = (int)(k * 64 + lane);
        if (pos > n1 && pos <= (int)lastNonNull) { cost += (int)(baseCost - (1u << (largestBits - nb[k]))); huffNode[pos].nbBits = (u8)maxNbBits; }
    }
    int totalCost = (int)wave_sum((u32)cost);
    totalCost >>= (largestBits - maxNbBits);
    // loop 2: n = last position (<= n1) not already at maxNbBits
    auto upTo = [&](u32 k, int lim) { return (int)(k * 64 + lane) <= lim; };
    const int n = highest(upTo(0, n1) && nb[0] != maxNbBits, upTo(1, n1) && nb[1] != maxNbBits, upTo(2, n1) && nb[2] != maxNbBits, upTo(3, n1) && nb[3] != maxNbBits);
    // loop 3: rankLast[maxNbBits - b] = last position of length b, if nothing shorter or equal sits above it
    if (lane < 14) L.rankLast[lane] = noSymbol;
    wave_lds_sync();
    {
        int above = -1;        // highest position (<= n) of any length smaller than b
        for (u32 b = 1; b < maxNbBits; ++b) {
            const int hi = highest(upTo(0, n) && nb[0] == b, upTo(1, n) && nb[1] == b, upTo(2, n) && nb[2] == b, upTo(3, n) && nb[3] == b);
            if (hi >= 0 && hi > above && lane == 0) L.rankLast[maxNbBits - b] = (u32)hi;
            if (hi > above) above = hi;
        }
    }
    wave_lds_sync();
    if (lane == 0) {
        u32* rankLast = L.rankLast; int nn = n;
        while (totalCost > 0) {
            u32 nBitsToDecrease = highbit32((u32)totalCost) + 1;
            for (; nBitsToDecrease > 1; nBitsToDecrease--) {
                const u32 highPos = rankLast[nBitsToDecrease], lowPos = rankLast[nBitsToDecrease - 1];
                if (highPos == noSymbol) continue;
                if (lowPos == noSymbol) break;
                const u32 highTotal = huffNode[highPos].count, lowTotal = 2 * huffNode[lowPos].count;
                if (highTotal <= lowTotal) break;
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
                while (huffNode[nn].nbBits == maxNbBits) nn--;
                huffNode[nn + 1].nbBits--;
                rankLast[1] = (u32)(nn + 1);
                totalCost++;
                continue;
            }
            huffNode[rankLast[1] + 1].nbBits--;
            rankLast[1]++;
            totalCost++;
        }
    }
    wave_lds_sync();
    return maxNbBits;
}

// ---- HUF_compressWeights (U/HufCompress.cs:40-125); returns bytes written, 0 = not compressible, 1 = single symbol ----
// Called by all 64 lanes of the chunk's wave (the result is uniform).  Lane 0 takes the decisions, FSE_normalizeCount and
// FSE_writeNCount (13 symbols); the table is built by the wave; then FSE_compress_usingCTable_generic (U/FseCompress.cs:722-820):
// symbol i uses state (i & 1), last symbol first — two independent chains, so lane p walks the symbols of parity p (each lane
// first fetches the transforms of its four symbols, so a chain step is one dependent LDS read) and leaves (bits, count) per
// symbol; the fields are placed by a prefix sum over the emission order and OR-ed into an LDS bit buffer, the two final states
// and the end mark behind them (BIT_closeCStream), and the bytes copied out.  (On one lane with a serial bit writer this was
// 300 000 cycles per chunk — the longest stretch of huf_tree_kernel; byte-identical by tests/test_gpu_parity.py.)
// Scratch: the tree's node array, dead by now.
__device__ inline u32 huf_compress_weights_wave(HufTreeLds& L, u8* dst, u32 wtSize, u32 lane)
{
    u32* const fields = reinterpret_cast<u32*>(L.nodes);            // [256] value | nbBits << 16, by symbol index
    SymTT* const tts = reinterpret_cast<SymTT*>(fields + 256);      // [256] the symbols' transforms
    u32* const bitbuf = reinterpret_cast<u32*>(tts + 256);          // [64]
    u16* const cumR = reinterpret_cast<u16*>(bitbuf + 64);          // [64] scratch of the table build
    static_assert(256 * 4 + 256 * sizeof(SymTT) + 64 * 4 + 64 * 2 <= sizeof(L.nodes), "scratch fits the node array");
    u32 res = 0xFFFFFFFFu, maxSV = 12, tableLog = 0, hs = 0;
    if (lane == 0) {
        do {
            if (wtSize <= 1) { res = 0; break; }
            while (!L.wcount[maxSV]) maxSV--;
            u32 maxCount = 0;
            for (u32 s = 0; s <= maxSV; s++) if (L.wcount[s] > maxCount) maxCount = L.wcount[s];
            if (maxCount == wtSize) { res = 1; break; }
            if (maxCount == 1) { res = 0; break; }
            tableLog = fse_optimal_table_log(6, wtSize, maxSV, 2);
            if (!fse_normalize_count(L.wnorm, tableLog, L.wcount, wtSize, maxSV, 0)) { res = 0; break; }
            hs = fse_write_ncount(dst, L.wnorm, maxSV, tableLog);
            if (!hs) { res = 0; break; }
            if (wtSize <= 2) { res = 0; break; }
        } while (false);
    }
    res = uniform(res); maxSV = uniform(maxSV); tableLog = uniform(tableLog); hs = uniform(hs);
    if (res != 0xFFFFFFFFu) return res;
    wave_lds_sync();                                                // (wnorm)
    fse_build_ctable_wave(L.wstate, L.wtt, L.wnorm, maxSV, tableLog, cumR, L.wsym, lane);
#pragma unroll
    for (u32 k = 0; k < 4; ++k) { const u32 i = k * 64 + lane; if (i < wtSize) tts[i] = L.wtt[L.weights[i]]; }
    if (lane < 64) bitbuf[lane] = 0;
    wave_lds_sync();
    if (lane < 2) {                                                 // the chain of parity `lane`
        const int top = (int)(wtSize - 1) - ((((wtSize - 1) & 1u) != lane) ? 1 : 0);
        u32 v = fse_init_state2(L.wstate, L.wtt, L.weights[top]);
        for (int i = top - 2; i >= 0; i -= 2) {
            const SymTT t = tts[i];
            const u32 nb = (v + t.deltaNbBits) >> 16;
            fields[i] = (v & ((1u << nb) - 1u)) | (nb << 16);
            v = L.wstate[(s32)(v >> nb) + t.deltaFindState];
        }
        L.sh[6 + lane] = v;                                         // (kShRoot and kShHSize: free by now)
    }
    wave_lds_sync();
    // emission order: symbol wtSize - 3 first, symbol 0 last; rank r = wtSize - 3 - i
    const u32 nEmit = wtSize - 2;
    u32 carry = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u32 r = k * 64 + lane;
        const bool have = r < nEmit;
        const u32 f = have ? fields[nEmit - 1 - r] : 0u;
        const u32 nb = f >> 16, val = f & 0xFFFFu;
        const u32 incl = wave_scan_incl(nb);
        const u32 at = carry + incl - nb;
        if (nb) {
            const u64 sh = (u64)val << (at & 31u);
            atomicOr(&bitbuf[at >> 5], (u32)sh);
            if ((u32)(sh >> 32)) atomicOr(&bitbuf[(at >> 5) + 1], (u32)(sh >> 32));
        }
        carry += read_lane(incl, 63);
    }
    wave_lds_sync();
    if (lane == 0) {                                                // FSE_flushCState x2 (state 2 first) + the end mark
        const u32 st0 = L.sh[6], st1 = L.sh[7], mask = (1u << tableLog) - 1u;
        const u64 tail = (u64)(st1 & mask) | ((u64)(st0 & mask) << tableLog) | (1ull << (2 * tableLog));
        const u32 at = carry;
        const u64 lo = tail << (at & 31u);                          // (2 * 6 + 1 bits shifted by at most 31: fits 64)
        bitbuf[at >> 5] |= (u32)lo;
        if ((u32)(lo >> 32)) bitbuf[(at >> 5) + 1] |= (u32)(lo >> 32);
    }
    wave_lds_sync();
    const u32 nBytes = (carry + 2 * tableLog + 1 + 7) >> 3;
    const u8* const bytes = reinterpret_cast<const u8*>(bitbuf);
#pragma unroll
    for (u32 k = 0; k < 4; ++k) { const u32 i = k * 64 + lane; if (i < nBytes) dst[hs + i] = bytes[i]; }
    return hs + nBytes;
}

__device__ __forceinline__ u32 min_gain(u32 srcSize) { return (srcSize >> 6) + 2; }   // ZSTD_minGain, strategies < btultra

// Front half: everything that is parallel over the literals or over the 256 symbols.  Hands the sorted leaves, the four
// per-stream histograms and the verdicts to huf_tree_kernel through the chunk's (still unused) output slot.
__global__ __launch_bounds__(256) void huf_hist_kernel(const u8* __restrict__ lits, const ChunkMeta* __restrict__ meta,
                                                       u8* __restrict__ slots, const u32 rawLiterals, const u8* __restrict__ src, const u32 chunkBytes)
{
    __shared__ HufBuildLds L;
    const u32 c = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const ChunkMeta m0 = meta_checked(meta[c]);
    const u32 litSize = m0.litSize, nbSeqIn = m0.nbSeq;
    HufWork* __restrict__ W = reinterpret_cast<HufWork*>(slots + (u64)c * kSlotStride);
    // (a chunk without sequences never copied its literals: they are its source bytes, lz_fast.hip)
    const u8* __restrict__ lit = m0.litFromSrc ? src + (u64)c * chunkBytes : lits + (u64)c * kLitStride;
#ifdef ZMI_LZ_STAMPS
    unsigned long long stampAcc[10] = {0,0,0,0,0,0,0,0,0,0}; unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif

    // ZSTD_compressLiterals: <= 63 literals are stored raw (no previous table in a one-block frame)
    // (rawLiterals: literal compression is off — the fast strategy with a step, i.e. negative levels; U/ZstdCompressInternal.cs:146-173)
    if (litSize <= 63 || rawLiterals) return;  // huf_tree_kernel stores them raw
    for (u32 i = tid; i < 8 * 256; i += 256) (&L.hist[0][0][0])[i] = 0;
    for (u32 i = tid; i < 2 * 256; i += 256) (&L.sample[0][0])[i] = 0;
    for (u32 i = tid; i < 513; i += 256) { Node z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0; L.nodes[i] = z; }
    __syncthreads();
    const u32 seg = (litSize + 3) / 4;
    {   // wave w counts segment w (these are also the per-stream histograms that size the four streams)
        const u32 s0 = wave * seg, s1 = (s0 + seg < litSize) ? s0 + seg : litSize;
        u32* H = L.hist[wave][lane & 1];      // two copies per wave, by lane parity: halves same-address atomic serialisation
        if (s0 < s1) {
            // 16 bytes per lane per load (aligned: the literal buffer is, a caller's source need not be): a byte-per-lane loop is
            // bound by one global-load latency per 64 bytes
            u32 a0 = s0 + ((0u - (u32)(uintptr_t)(lit + s0)) & 15u); if (a0 > s1) a0 = s1;
            if (s0 + lane < a0) atomicAdd(&H[lit[s0 + lane]], 1u);
            const u32 nVec = (s1 - a0) >> 4;
            const uint4* v4 = reinterpret_cast<const uint4*>(lit + a0);
            // Skewed data serialises the atomics of the lanes that hold the most frequent byte.  Guess that byte from 64
            // samples and count it with a ballot instead (scalar add, no LDS traffic); everything else takes the atomic.
            u32 hot = 256, hotCnt = 0;
            if (nVec >= 64) {
                const u32 b0 = v4[lane].x & 0xFFu;
                u32 best = 0;
#pragma unroll
                for (u32 k = 0; k < 8; ++k) {
                    const u32 cand = read_lane(b0, k * 8);
                    const u32 cn = popc64(ballot(b0 == cand));
                    if (cn > best) { best = cn; hot = cand; }
                }
                if (best < 6) hot = 256;
            }
            // (four loads in flight per lane: a wave's 16 atomics per load are far shorter than the load's way from HBM, and 32 waves
            //  per CU do not cover it — the loop used to wait out every load: 0.73 ms per GiB of Zipf bytes against 0.13 for reading them)
            auto count16 = [&](const uint4 v) {
                const u32 d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
#pragma unroll
                    for (u32 b = 0; b < 4; ++b) {
                        const u32 sym = (d[k] >> (8 * b)) & 0xFFu;
                        const bool isHot = sym == hot;
                        hotCnt += popc64(ballot(isHot));
                        if (!isHot) atomicAdd(&H[sym], 1u);
                    }
                }
            };
            //  (the loads are unconditional — past the end they re-read the segment's first piece —: behind a branch the compiler
            //   waits for every load in flight instead of the oldest)
            uint4 q[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) { const u32 ix = lane + 64 * k; q[k] = v4[ix < nVec ? ix : 0u]; }
            for (u32 i = lane; i < nVec; i += 256) {
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
                    const uint4 v = q[k];
                    const u32 nx = i + 64 * k + 256;
                    q[k] = v4[nx < nVec ? nx : 0u];
                    if (i + 64 * k < nVec) count16(v);
                }
            }
            if (lane == 0 && hot < 256) atomicAdd(&H[hot], hotCnt);
            const u32 t0 = a0 + (nVec << 4);
            if (t0 + lane < s1) atomicAdd(&H[lit[t0 + lane]], 1u);
        }
    }
    const u32 suspect = (nbSeqIn == 0) || (litSize / nbSeqIn >= 20);
    const bool doSample = suspect && litSize >= 4096 * 10;
    if (doSample && wave < 2) {       // HUF_compress_internal's 2 x 4 KiB pre-check (U/HufCompress.cs:1412-1446)
        const u8* sp = wave == 0 ? lit : lit + litSize - 4096;
        for (u32 i = lane; i < 4096; i += 64) atomicAdd(&L.sample[wave][sp[i]], 1u);
    }
    __syncthreads();
    {
        u32 tot = 0;
#pragma unroll
        for (u32 w = 0; w < 4; ++w) { const u32 v = L.hist[w][0][tid] + L.hist[w][1][tid]; W->hist[w][tid] = (u16)v; tot += v; }
        L.count[tid] = tot;
    }
    if (doSample && wave < 2) {
        u32 mx = 0;
        for (u32 i = lane; i < 256; i += 64) { const u32 v = L.sample[wave][i]; mx = v > mx ? v : mx; }
        mx = wave_max(mx);
        if (lane == 0) L.sampleMax[wave] = mx;
    }
    __syncthreads();

    ZMI_HSTAMP(0);
    {   // compressible at all?  (HUF_compress_internal, U/HufCompress.cs:1412-1462) — every thread reaches the same verdict
        const u32 cnt = L.count[tid];
        const u64 nz = ballot(cnt != 0);
        const u32 wmx = wave_max(cnt);
        if (lane == 0) { L.redMaxSV[wave] = nz ? wave * 64 + 63 - (u32)__builtin_clzll(nz) : 0; L.redLargest[wave] = wmx; }
        __syncthreads();
        u32 maxSV0 = 0, largest = 0;
        for (u32 w = 0; w < 4; ++w) { maxSV0 = L.redMaxSV[w] > maxSV0 ? L.redMaxSV[w] : maxSV0; largest = L.redLargest[w] > largest ? L.redLargest[w] : largest; }
        u32 compressed = 1, rle = 0;
        if (doSample && L.sampleMax[0] + L.sampleMax[1] <= ((2 * 4096) >> 7) + 4) { compressed = 0; maxSV0 = 255; }
        else if (largest == litSize) { rle = 1; compressed = 0; }
        else if (largest <= (litSize >> 7) + 4) compressed = 0;
        if (tid == 0) { L.sh[kShCompressed] = compressed; L.sh[kShMaxSV] = maxSV0; L.sh[kShRle] = rle; L.sh[kShRleByte] = rle ? lit[0] : 0; }
        if (compressed) {
            // HUF_sort's bucket placement (U/HufCompress.cs:635-680) without the serial counters: a symbol lands behind
            // every symbol of a higher bucket and behind the lower-numbered symbols of its own bucket
            const u32 idx = huf_get_index(cnt);
            L.sortIdx[tid] = (u8)idx;
            __syncthreads();
            if (tid <= maxSV0) {
                u32 pos = 0;
                for (u32 mS = 0; mS <= maxSV0; ++mS) { const u32 im = L.sortIdx[mS]; pos += (im > idx) || (im == idx && mS < tid); }
                Node nd; nd.count = cnt; nd.parent = 0; nd.byte = (u8)tid; nd.nbBits = 0;
                L.nodes[1 + pos] = nd;
            }
            if (tid < 26) {            // extent of log2 bucket n (symbols whose index is n-1), for huf_sort_bucket
                const u32 nB = 165 + tid; u32 start = 0, size = 0;
                for (u32 mS = 0; mS <= maxSV0; ++mS) { const u32 im = L.sortIdx[mS]; start += im >= nB; size += im == nB - 1; }
                L.rankBase[nB] = (u16)start; L.rankCurr[nB] = (u16)(start + size);
            }
        }
    }
    ZMI_HSTAMP(1);
    __syncthreads();
    if (L.sh[kShCompressed]) {
        if (tid < 26) huf_sort_bucket(L, tid);                 // HUF_sort's per-bucket quicksorts, one bucket per lane
        __syncthreads();
        W->leafCount[tid] = L.nodes[1 + tid].count; W->leafByte[tid] = L.nodes[1 + tid].byte;
    }
    if (tid < 4) W->flags[tid] = L.sh[tid];                    // compressed, maxSV, rle, rleByte
    ZMI_HSTAMP(2);
#ifdef ZMI_LZ_STAMPS
    if (tid == 0) for (int i = 0; i < 3; i++) atomicAdd(&g_hufStamps[i], stampAcc[i]);
#endif
}

// Back half: the serial constructions (HUF_buildTree, HUF_setMaxHeight, HUF_compressWeights) and the decisions.  One
// wave per chunk and ~6 KiB of LDS, so that ~25 chunks per CU hide each other's LDS latency; per-symbol steps run
// four symbols per lane.
__global__ __launch_bounds__(64) void huf_tree_kernel(ChunkMeta* __restrict__ meta, HufTable* __restrict__ tables, const u8* __restrict__ slots,
                                                      const u32 rawLiterals)
{
    __shared__ HufTreeLds L;
    const u32 c = blockIdx.x, lane = threadIdx.x, tid = lane;
    ChunkMeta m = meta_checked(meta[c]);
    const u32 litSize = m.litSize;
    const u32 lhSizeRaw = 1 + (litSize > 31) + (litSize > 4095);
    const HufWork* __restrict__ W = reinterpret_cast<const HufWork*>(slots + (u64)c * kSlotStride);
#ifdef ZMI_LZ_STAMPS
    unsigned long long stampAcc[10] = {0,0,0,0,0,0,0,0,0,0}; unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif
    // ZSTD_compressLiterals: <= 63 literals are stored raw (no previous table in a one-block frame)
    if (litSize <= 63 || rawLiterals) {        // (or ZSTD_noCompressLiterals because literal compression is disabled, U/ZstdCompressLiterals.cs:99-101)
        if (tid == 0) { m.litMode = kLitRaw; m.lhSize = lhSizeRaw; m.litSectionSize = lhSizeRaw + litSize; meta[c] = m; }
        return;
    }
    const u32 lhSize = 3 + (litSize >= 1024) + (litSize >= 16384);
    const u32 single = litSize < 256;
    HufTable* T = tables + c;
    const u32 shCompressed = W->flags[0], maxSV = W->flags[1], shRle = W->flags[2], shRleByte = W->flags[3];
    u32 huffLog = 0;
    u32 streamBits[4] = { 0, 0, 0, 0 };
    if (shCompressed) {
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {
            const u32 pos = k * 64 + lane;
            Node nd; nd.count = W->leafCount[pos]; nd.parent = 0; nd.byte = W->leafByte[pos]; nd.nbBits = 0;
            L.nodes[1 + pos] = nd;
            Node z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0;
            L.nodes[257 + pos] = z;
            L.nbBits[pos] = 0;
        }
        if (lane == 0) { Node z; z.count = 0; z.parent = 0; z.byte = 0; z.nbBits = 0; L.nodes[0] = z; }
        if (lane < 13) L.wcount[lane] = 0;
        wave_lds_sync();
        Node* huffNode = L.nodes + 1;
        if (lane == 0) { int root = 0; L.sh[kShNonNull] = (u32)huf_build_tree(huffNode, maxSV, &root); L.sh[kShRoot] = (u32)root; }
        wave_lds_sync();
        ZMI_HSTAMP(3);
        const u32 nonNull = L.sh[kShNonNull], root = L.sh[kShRoot];
#pragma unroll
        for (u32 k = 0; k < 4; ++k) {   // depth of every leaf = number of parent links up to the root (HUF_buildTree's nbBits loops)
            const u32 pos = k * 64 + lane;
            if (pos <= nonNull) { u32 node = pos, d = 0; while (node != root) { node = huffNode[node].parent; d++; } huffNode[pos].nbBits = (u8)d; }
        }
        wave_lds_sync();
        ZMI_HSTAMP(4);
        {
            u32 hl = fse_optimal_table_log(11, litSize, maxSV, 1);
            hl = huf_set_max_height_wave(L, huffNode, nonNull, hl);
            // HUF_buildCTableFromTree: symbols per length, then the first code value of every length
            u32 nbPerRank[13];
            {
                u32 nbk[4];
#pragma unroll
                for (u32 k = 0; k < 4; ++k) { const u32 pos = k * 64 + lane; nbk[k] = pos <= nonNull ? huffNode[pos].nbBits : 0xFFu; }
#pragma unroll
                for (u32 r = 0; r < 13; ++r) nbPerRank[r] = popc64(ballot(nbk[0] == r)) + popc64(ballot(nbk[1] == r)) + popc64(ballot(nbk[2] == r)) + popc64(ballot(nbk[3] == r));
            }
            if (lane == 0) {
                u32 mn = 0;
#pragma unroll
                for (int r = 12; r > 0; r--) { if (r <= (int)hl) { L.valPerRank[r] = mn; mn += nbPerRank[r]; mn >>= 1; } }
                L.valPerRank[0] = 0;
                L.sh[kShHuffLog] = hl;
            }
        }
        wave_lds_sync();
        ZMI_HSTAMP(5);
        huffLog = L.sh[kShHuffLog];
#pragma unroll
        for (u32 k = 0; k < 4; ++k) { const u32 pos = k * 64 + lane; if (pos <= maxSV) L.nbBits[huffNode[pos].byte] = huffNode[pos].nbBits; }   // HUF_buildCTableFromTree
        wave_lds_sync();
        {   // canonical codes: symbols of one length get consecutive values in symbol order (U/HufCompress.cs:766-785)
            u32 nb[4], pre[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) { const u32 sIdx = k * 64 + lane; nb[k] = sIdx <= maxSV ? L.nbBits[sIdx] : 0; pre[k] = 0; }
            for (u32 r = 1; r <= 12; r++) {
                u32 acc = 0;
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
                    const u64 b = ballot(nb[k] == r);
                    if (nb[k] == r) pre[k] = acc + popc64(b & lanemask_lt());
                    acc += popc64(b);
                }
            }
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                const u32 sIdx = k * 64 + lane;
                const u32 code = nb[k] ? L.valPerRank[nb[k]] + pre[k] : 0;
                T->nbBits[sIdx] = (u8)nb[k]; T->code[sIdx] = (u16)code;
                if (sIdx < maxSV) { const u32 wt = nb[k] ? huffLog + 1 - nb[k] : 0; L.weights[sIdx] = (u8)wt; atomicAdd(&L.wcount[wt], 1u); }   // HUF_writeCTable_wksp's bitsToWeight + the weights' histogram
            }
            // stream sizes = sum(count x nbBits) per segment (HUF_compress1X_usingCTable_internal + HUF_closeCStream)
#pragma unroll
            for (u32 w = 0; w < 4; ++w) {
                u32 bits = 0;
#pragma unroll
                for (u32 k = 0; k < 4; ++k) bits += (u32)W->hist[w][k * 64 + lane] * nb[k];
                streamBits[w] = wave_sum(bits);
            }
        }
        wave_lds_sync();
        ZMI_HSTAMP(6);
    }
    u32 wsWave = 0;
    if (shCompressed) wsWave = huf_compress_weights_wave(L, T->hdr + 1, maxSV, lane);      // (uniform branch, uniform result)
    if (tid != 0) return;

    // ---------------- remaining serial section: tree description + decisions ----------------
    bool compressed = shCompressed != 0;
    const bool rle = shRle != 0;
    u32 hSize = 0, cLitSize = 0;
    u32 streamSize[4] = { 0, 0, 0, 0 };
    if (compressed) {
        T->maxSV = maxSV; T->tableLog = huffLog;
        const u32 ws = wsWave;
        if (ws > 1 && ws < maxSV / 2) { T->hdr[0] = (u8)ws; hSize = ws + 1; }
        else if (maxSV > 128) { compressed = false; }     // HUF_writeCTable_wksp fails -> ZSTD_compressLiterals stores raw
        else {
            T->hdr[0] = (u8)(128 + (maxSV - 1));
            L.weights[maxSV] = 0;
            for (u32 n = 0; n < maxSV; n += 2) T->hdr[(n / 2) + 1] = (u8)((L.weights[n] << 4) + L.weights[n + 1]);
            hSize = ((maxSV + 1) / 2) + 1;
        }
        if (compressed && hSize + 12 >= litSize) compressed = false;
    }
    if (compressed) {
        if (single) {
            const u32 bits = streamBits[0] + streamBits[1] + streamBits[2] + streamBits[3];
            streamSize[0] = (bits >> 3) + 1;
            cLitSize = hSize + streamSize[0];
        } else {
            if (litSize < 12) compressed = false;
            cLitSize = hSize + 6;
#pragma unroll
            for (u32 w = 0; w < 4; w++) {
                streamSize[w] = (streamBits[w] >> 3) + 1;
                if (streamSize[w] > 65535) compressed = false;    // (a zero-length stream cannot occur: the end mark is a byte)
                cLitSize += streamSize[w];
            }
        }
        if (compressed && cLitSize >= litSize - 1) compressed = false;                     // HUF_compressCTable_internal
        if (compressed && cLitSize >= litSize - min_gain(litSize)) compressed = false;     // ZSTD_compressLiterals
    }
    if (compressed) {
        m.litMode = kLitCompressed; m.litSingle = single; m.lhSize = lhSize; m.hufHdrSize = hSize;
        m.streamSize[0] = streamSize[0]; m.streamSize[1] = streamSize[1]; m.streamSize[2] = streamSize[2]; m.streamSize[3] = streamSize[3];
        m.litSectionSize = lhSize + cLitSize;
    } else if (rle) {
        m.litMode = kLitRle; m.lhSize = lhSizeRaw; m.litSectionSize = lhSizeRaw + 1; m.rleByte = shRleByte;
    } else {
        m.litMode = kLitRaw; m.lhSize = lhSizeRaw; m.litSectionSize = lhSizeRaw + litSize;
    }
    meta[c] = m;
    ZMI_HSTAMP(7);
#ifdef ZMI_LZ_STAMPS
    for (int i = 3; i < 10; i++) atomicAdd(&g_hufStamps[i], stampAcc[i]);
#endif
}

#ifdef ZMI_LZ_STAMPS
extern "C" void ZSTDMI_debugReadHufStamps(unsigned long long* out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_hufStamps), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_hufStamps), z, sizeof z); }
}
#endif

// ------------------------------------------------------------------------------------------------
constexpr u32 kSymPerLane = 8;
constexpr u32 kTileSyms   = 64 * kSymPerLane;             // 512 symbols per wave step, <= 5632 bits
constexpr u32 kTileWords  = (kTileSyms * 11 + 31) / 32 + 2;

struct HufEncLds {
    u32 ct[256];                       // code | nbBits << 16
    u32 tile[4][kTileWords];
};

// `dst` != nullptr: the literals section goes straight to its final place in the output (offsets[] from the scan; the
// sequences section and the headers follow through gather_kernel); nullptr: into the chunk's slot (test hook).
__global__ __launch_bounds__(256) void huf_encode_kernel(const u8* __restrict__ lits, const ChunkMeta* __restrict__ meta,
                                                         const HufTable* __restrict__ tables, u8* __restrict__ slots,
                                                         u8* __restrict__ dst, const u64* __restrict__ offsets, u64 dstCapacity,
                                                         const u8* __restrict__ src, const u32 chunkBytes)
{
    __shared__ HufEncLds L;
    const u32 c = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const ChunkMeta m = meta_checked(meta[c]);
    const u32 litSize = m.litSize;
    const u8* __restrict__ lit = m.litFromSrc ? src + (u64)c * chunkBytes : lits + (u64)c * kLitStride;
    u8* __restrict__ body = slots + (u64)c * kSlotStride + m.fhSize + 3;      // block body starts after frame + block header
    if (dst) {
        if (m.blockType != 2) return;                                        // stored raw: gather copies the source bytes
        const u64 off = offsets[c];
        if (off + m.outSize > dstCapacity) return;                           // host reports dstSize_tooSmall from the scanned total
        body = dst + off + m.fhSize + 3;
    }

    if (m.litMode != kLitCompressed) {
        // ZSTD_noCompressLiterals / ZSTD_compressRleLiteralsBlock (U/ZstdCompressLiterals.cs:8-83)
        const u32 type = m.litMode == kLitRle ? 1u : 0u;
        if (tid == 0) {
            switch (m.lhSize) {
            case 1: body[0] = (u8)(type + (litSize << 3)); break;
            case 2: writeLE16(body, type + (1u << 2) + (litSize << 4)); break;
            default: writeLE24(body, type + (3u << 2) + (litSize << 4)); break;
            }
            if (type == 1) body[m.lhSize] = (u8)m.rleByte;
        }
        if (type == 0) for (u32 i = tid; i < litSize; i += 256) body[m.lhSize + i] = lit[i];
        return;
    }
    const HufTable* __restrict__ T = tables + c;
    L.ct[tid] = (u32)T->code[tid] | ((u32)T->nbBits[tid] << 16);
    if (tid == 0) {
        const u32 cLitSize = m.litSectionSize - m.lhSize;
        switch (m.lhSize) {       // ZSTD_compressLiterals header (U/ZstdCompressLiterals.cs:150-182)
        case 3: writeLE24(body, 2u + ((m.litSingle ? 0u : 1u) << 2) + (litSize << 4) + (cLitSize << 14)); break;
        case 4: writeLE32(body, 2u + (2u << 2) + (litSize << 4) + (cLitSize << 18)); break;
        default: writeLE32(body, 2u + (3u << 2) + (litSize << 4) + (cLitSize << 22)); body[4] = (u8)(cLitSize >> 10); break;
        }
    }
    for (u32 i = tid; i < m.hufHdrSize; i += 256) body[m.lhSize + i] = T->hdr[i];
    u8* payload = body + m.lhSize + m.hufHdrSize;
    if (!m.litSingle && tid < 3) writeLE16(payload + 2 * tid, meta[c].streamSize[tid]);       // jump table
    __syncthreads();

    const u32 nStreams = m.litSingle ? 1 : 4;
    if (wave >= nStreams) return;
    const u32 seg = m.litSingle ? litSize : (litSize + 3) / 4;
    const u32 s0 = wave * seg;
    const u32 len = (wave == nStreams - 1) ? litSize - s0 : seg;
    u8* out = payload + (m.litSingle ? 0 : 6);
    for (u32 w = 0; w < wave; w++) out += meta[c].streamSize[w];
    const u8* __restrict__ sym = lit + s0;
    u32* tile = L.tile[wave];

    u32 carry = 0, carryBits = 0;        // bits of a partially filled dword carried into the next tile
    u32 outWords = 0;                    // dwords already flushed to `out`
    // The lane's 8 symbols of a tile sit in 8 consecutive bytes (descending): one unaligned 8-byte load when all are in range —
    // issued a tile AHEAD (its way from HBM was the longest part of a tile's time); the tile belongs to this wave alone, so its LDS
    // traffic needs program order only (a wavefront fence: the workgroup fences that stood here waited for every store to reach
    // L2, twice per tile).  0.74 -> 0.59 ms per GiB of Zipf bytes; writing the tile out a step later changed nothing more.
    auto load_pack = [&](u32 k0) -> u64 {
        const u32 kb = k0 + lane * kSymPerLane;
        return (k0 < len && kb + kSymPerLane <= len) ? *reinterpret_cast<const u64u*>(sym + (len - kSymPerLane - kb)) : 0ull;
    };
    static_assert(kTileWords <= 3 * 64, "a tile is flushed in at most three stores per lane");
    u64 packNext = load_pack(0);
    for (u32 k0 = 0; k0 < len; k0 += kTileSyms) {
        for (u32 i = lane; i < kTileWords; i += 64) tile[i] = 0;
        // lane handles reversed indices k0 + lane*8 .. +7  ->  source bytes len-1-k, descending
        const u32 kb = k0 + lane * kSymPerLane;
        const bool full8 = kb + kSymPerLane <= len;
        const u64 pack = packNext;
        packNext = load_pack(k0 + kTileSyms);
        // codes are at most 11 bits: symbols 0..4 fit one 64-bit accumulator (<= 55 bits), symbols 5..7 another (<= 33),
        // so the per-symbol step is a plain shift-or; the two halves are joined once
        u64 accA = 0, accB = 0; u32 nA = 0, nB = 0;
#pragma unroll
        for (u32 j = 0; j < kSymPerLane; j++) {
            const u32 k = kb + j;
            const u32 s8 = full8 ? (u32)(pack >> (8 * (kSymPerLane - 1 - j))) & 0xFFu : (k < len ? (u32)sym[len - 1 - k] : 0u);
            const u32 e = k < len ? L.ct[s8] : 0u;
            const u64 code = e & 0xFFFF; const u32 b = e >> 16;
            if (j < 5) { accA |= code << nA; nA += b; } else { accB |= code << nB; nB += b; }
        }
        const u32 nb = nA + nB;
        const u64 lo = accA | (nA < 64 ? accB << nA : 0ull);
        const u64 hi = nA ? accB >> (64 - nA) : 0ull;
        const u32 incl = wave_scan_incl(nb);
        const u32 tileBits = read_lane(incl, 63);
        const u32 bitOff = carryBits + incl - nb;
        if (nb) {
            const u32 w0 = bitOff >> 5, sh = bitOff & 31;
            // up to 88 bits shifted by <32 -> spans at most 4 dwords
            const u64 a0 = lo << sh;
            const u64 a1 = sh ? ((lo >> (64 - sh)) | (hi << sh)) : hi;
            atomicOr(&tile[w0], (u32)a0);
            if ((u32)(a0 >> 32)) atomicOr(&tile[w0 + 1], (u32)(a0 >> 32));
            if ((u32)a1) atomicOr(&tile[w0 + 2], (u32)a1);
            if ((u32)(a1 >> 32)) atomicOr(&tile[w0 + 3], (u32)(a1 >> 32));
        }
        if (lane == 0 && carryBits) atomicOr(&tile[0], carry);
        const u32 total = carryBits + tileBits;
        const u32 fullWords = total >> 5;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (u32 r = 0; r < 3; ++r) { const u32 i = lane + 64 * r; if (i < fullWords) *(u32u*)(out + 4 * (outWords + i)) = tile[i]; }
        carry = tile[fullWords]; carryBits = total & 31;       // every lane reads the same LDS word
        outWords += fullWords;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {       // end mark + tail bytes (HUF_closeCStream, U/HufCompress.cs:964-979)
        u32 v = carry | (1u << carryBits);
        const u32 nbytes = (carryBits + 1 + 7) >> 3;
        u8* p = out + 4 * outWords;
        for (u32 i = 0; i < nbytes; i++) { p[i] = (u8)v; v >>= 8; }
    }
}

void launch_huf_build(const u8* lits, ChunkMeta* meta, HufTable* tables, u8* slots, u32 nChunks, u32 rawLiterals, const u8* src, u32 chunkBytes,
                      hipStream_t stream, StageHook hook)
{
    hipLaunchKernelGGL(huf_hist_kernel, dim3(nChunks), dim3(256), 0, stream, lits, meta, slots, rawLiterals, src, chunkBytes);
    hook("huf_hist");
    hipLaunchKernelGGL(huf_tree_kernel, dim3(nChunks), dim3(64), 0, stream, meta, tables, slots, rawLiterals);
    hook("huf_tree");
}
void launch_huf_encode(const u8* lits, const ChunkMeta* meta, const HufTable* tables, u8* slots, u8* dst, const u64* offsets, u64 dstCapacity,
                       u32 nChunks, const u8* src, u32 chunkBytes, hipStream_t stream)
{
    hipLaunchKernelGGL(huf_encode_kernel, dim3(nChunks), dim3(256), 0, stream, lits, meta, tables, slots, dst, offsets, dstCapacity, src, chunkBytes);
}

} // namespace zmi
