// decode_seq.hip — sequences of a frame on gfx950, in three stages:
//
//   seq_decode      : one wave per compressed block, ALL blocks of all frames at once.  The block's three FSE tables are built by
//                     the whole wave (ZSTD_buildFSETable, U/ZstdDecompressBlock.cs:1571-1710) from its own sequences header or
//                     from the earlier block / dictionary that block_link named for a repeat-mode table (:1746-1840); the state
//                     chain (ZSTD_decodeSequence, :2360-2484) runs on the scalar unit; every lane extracts the fields of its own
//                     sequence.  Repcodes cannot be resolved yet — they come from the previous block — so they are resolved
//                     SYMBOLICALLY: every offset is a number, or "the block's starting repcode i, minus d (floor 1)", and the
//                     block as a whole maps its three starting repcodes to its three ending ones the same way.  Out: one SeqRec
//                     per sequence (offset, lengths, block-relative output position), the block's regenerated size and transfer.
//   place_literals  : one wave per block, all blocks at once, after block_offsets has turned sizes into output offsets: raw
//                     and RLE blocks, the literals of every sequence and the trailing literals go to their final place.
//   exec_matches    : one wave per frame, blocks in order (the only ordered stage): the matches (ZSTD_execSequence, :2187-2262),
//                     64 at a time in dependency rounds; then the frame's checksum (U/ZstdDecompress.cs:1186-1208).
#include "zmi_decode.h"

namespace zmi {

#ifdef ZMI_LZ_STAMPS
__device__ unsigned long long g_seqStamps[16];
#define ZMI_SSTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stampAcc[i] += now_ - stampLast; stampLast = now_; } while (0)
extern "C" void ZSTDMI_debugReadSeqStamps(unsigned long long* out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_seqStamps), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seqStamps), z, sizeof z); }
}
#else
#define ZMI_SSTAMP(i) do { } while (0)
#endif

// ---- seq_decode: kPack blocks per workgroup of four waves ----
// The FSE state chain of a block (ZSTD_decodeSequence's three state updates, U/ZstdDecompressBlock.cs:2360-2484) is serial and
// does the same few operations per sequence whatever runs it: on one wave per block every instruction moves ONE sequence
// forward, and the kernel is bound by instruction issue (45 per sequence).  So a workgroup takes kPack consecutive blocks:
//   A  their tables are built by the four waves, one (block, table) pair per wave at a time, 64 lanes each;
//   B  wave 0 walks the chains, lane s = block s — kPack sequences per instruction, each lane with its own table slot in
//      LDS and its own bit window — and leaves one (states, bit position) record per sequence, 64 sequences per block at a time;
//   C  waves 1-3 turn the records of the PREVIOUS 64 sequences of each block into SeqRecs meanwhile (fields, repcodes, prefix
//      sums: 64 lanes per block), and keep each block's bitstream staged in an LDS ring ahead of its chain (a global load in
//      the chain costs it several hundred cycles per sequence, an LDS read rides along with the table reads); one barrier
//      per 64 sequences hands the record buffers and the ring positions over.
constexpr u32 kPack = 5;                 // 5 x (5 KiB of tables + 2 KiB of stream + records): four workgroups per CU
constexpr u32 kRing = 2048, kRingDw = kRing / 4;
constexpr u32 kTabLL = 0, kTabML = 512, kTabOF = 1024, kTabWords = 1280;

struct SeqLds {
    u32 tab[kPack][kTabWords];           // one entry per state: nextState:16 | nbBits:4 (bit 16) | nbBits + nbAddBits:6 (bit 20: what a step consumes) | symbol:6 (bit 26)
    u32 recSt[2][kPack][64];             // phase B -> C (double-buffered): the three states before each of a block's 64 sequences ...
    u16 recPos[2][kPack][64];            // ... and the bit position, as posBase - position
    s32 posBase[2][kPack];               // bit position at the start of the batch
    // the stream of block s: byte x (from the stream's start) lives at ring[s][x & 2047] while it is staged.  ringLo = lowest
    // staged byte the chain may rely on in a batch, curByte = byte the chain stands at when a batch starts (both double-buffered:
    // written during one batch for the next)
    union {
        u32 ring[kPack][kRingDw + 4];     // (+ the first three dwords again behind the end: a window of four never wraps)
        struct { s16 norm[4][64]; u16 symbolNext[4][64]; } build;     // table-build scratch, one per wave (phase A only)
    };
    s32 ringLo[2][kPack], curByte[2][kPack];
    u32 tblErr[kPack][3];
    u32 chainErr[kPack];
    s32 endPos[kPack];
};
__device__ __forceinline__ u64 read_lane64(u64 v, u32 l) { return (u64)read_lane((u32)v, l) | ((u64)read_lane((u32)(v >> 32), l) << 32); }

__device__ __forceinline__ u32 pack_entry(u32 sym, u32 nextState, u32 tableLog, u32 tableSize, int kind)
{
    const u32 nb = tableLog - highbit32(nextState);
    const u32 add = kind == 0 ? (u32)dLL_bits[sym] : kind == 1 ? sym : (u32)dML_bits[sym];
    return (((nextState << nb) - tableSize) & 0xFFFFu) | (nb << 16) | ((nb + add) << 20) | (sym << 26);
}

// ZSTD_buildFSETable_body (U/ZstdDecompressBlock.cs:1571-1710) by the whole wave, as build_seq_dtable_wave (zmi_decode.h) but
// into packed entries; until the last pass a cell holds its symbol.
__device__ __forceinline__ void build_seq_ptable_wave(u32* t, u16* cum, const s16* norm, u32 maxSV, u32 tableLog, int kind, u32 lane)
{
    const u32 tableSize = 1u << tableLog, mask = tableSize - 1, step = (tableSize >> 1) + (tableSize >> 3) + 3;
    const int nrm = lane <= maxSV ? (int)norm[lane] : 0;
    const bool low = nrm == -1;
    const u64 lowMask = ballot(low);
    const u32 highThreshold = tableSize - 1 - popc64(lowMask);
    if (low) t[tableSize - 1 - popc64(lowMask & lanemask_lt())] = lane;
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
            t[p] = lo;
        }
    }
    wave_lds_sync();
    u32 nxt = low ? 1u : cnt;                    // symbolNext of symbol `lane`
    for (u32 u0 = 0; u0 < tableSize; u0 += 64) {
        const u32 u = u0 + lane; const bool valid = u < tableSize;
        const u32 sym = valid ? t[u] : 0xFFFFu;
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
        wave_lds_sync();
        if (valid) t[u] = pack_entry(sym, myNext, tableLog, tableSize, kind);
    }
    wave_lds_sync();
}

// One table of ZSTD_buildSeqTable (U/ZstdDecompressBlock.cs:1746-1840), whole wave; only the NCount header is parsed by one lane.
// type: 0 predefined, 1 RLE, 2 compressed (repeat mode was resolved to the block that defined the table).  Returns the table's
// log, or 0xFFFFFFFF when the description is corrupt.
__device__ __forceinline__ u32 set_seq_table(s16* norm, u16* symbolNext, u32* t, u32 type, u32 max, u32 maxLog,
                                             const u8* src, u32 srcSize, int kind, const s16* defNorm, u32 defLog, u32 defMax, u32 lane)
{
    switch (type) {
    case 1: {
        if (!srcSize) return 0xFFFFFFFFu;
        const u32 sym = uniform((u32)src[0]);
        if (sym > max) return 0xFFFFFFFFu;
        if (lane == 0) t[0] = pack_entry(sym, 1, 0, 1, kind) & ~0x000FFFFFu;      // one state: no bits, stays where it is
        wave_lds_sync();
        return 0; }
    case 0:
        if (lane <= defMax) norm[lane] = defNorm[lane];
        wave_lds_sync();
        build_seq_ptable_wave(t, symbolNext, norm, defMax, defLog, kind, lane);
        return defLog;
    default: {
        u32 maxSV = max, tableLog = 0, hs = 0;
        if (lane == 0) hs = read_ncount(norm, &maxSV, &tableLog, src, srcSize);
        hs = uniform(hs); maxSV = uniform(maxSV); tableLog = uniform(tableLog);
        if (!hs || tableLog > maxLog) return 0xFFFFFFFFu;
        wave_lds_sync();
        build_seq_ptable_wave(t, symbolNext, norm, maxSV, tableLog, kind, lane);
        return tableLog; }
    }
}

// a repcode slot during the symbolic resolution: kind 0 = the constant val; kind 1..3 = max(in[kind - 1] - val, 1)
struct RepSlot { u32 kind, val; };
__device__ __forceinline__ RepSlot rep_minus_one(RepSlot s)       // offset = rep0 - 1, 0 -> 1 (U/ZstdDecompressBlock.cs:2421-2433)
{
    if (s.kind == 0) { s.val = s.val > 1 ? s.val - 1 : 1u; return s; }
    s.val += 1; return s;
}

// dword d (stream-relative, may lie outside) of a backward bitstream of `size` bytes at s; zeros outside the stream
__device__ __forceinline__ u32 stream_dword_z(const u8* s, s32 size, s32 d)
{
    const s32 b = 4 * d;
    if (b >= 0 && b + 4 <= size) return readLE32(s + b);
    u32 v = 0;
    for (s32 i = 0; i < 4; i++) { const s32 k = b + i; if (k >= 0 && k < size) v |= (u32)s[k] << (8 * i); }
    return v;
}
// bits [p - nb, p) of the stream (nb <= 32), lane-private
__device__ __forceinline__ u32 stream_field(const u8* s, s32 size, s32 p, u32 nb)
{
    if (!nb) return 0;
    const s32 q = p - (s32)nb, by = q >> 3;
    u64 v;
    if (by >= 0 && by + 8 <= size) v = readLE64(s + by);
    else { v = 0; for (s32 i = 0; i < 8; i++) { const s32 k = by + i; if (k >= 0 && k < size) v |= (u64)s[k] << (8 * i); } }
    return (u32)(v >> (u32)(q & 7)) & (0xFFFFFFFFu >> (32 - nb));
}

// phase C for one block: records `buf` of its sequences [base, base + 64) -> SeqRecs; the block's running state (symbolic repcodes,
// output and literal totals, error) is wave-uniform and lives in the caller's registers
struct SlotState { RepSlot s0, s1, s2; u32 outBase, litUsed, err; };
__device__ __forceinline__ void seq_fields_batch(const SeqLds& L, u32 buf, u32 sl, u32 base, u32 nbSeq, const u8* sp, s32 size, u32 litSize,
                                                 SeqRec* __restrict__ rec, SlotState& S, u32 lane)
{
    const u32 cnt = nbSeq - base < 64 ? nbSeq - base : 64;
    const u32* const T = L.tab[sl];
    // ---- every lane: the fields of its own sequence ----
    const bool have = lane < cnt;
    u32 ll = 0, ml = 0, off = 1, code = 4;      // code 4 = a real offset; 0..3 = repcode selector
    if (have) {
        const u32 st = L.recSt[buf][sl][lane];
        const u32 qLL = T[kTabLL + (st & 1023u)], qML = T[kTabML + ((st >> 10) & 1023u)], qOF = T[kTabOF + (st >> 20)];
        const u32 aLL = ((qLL >> 20) & 63u) - ((qLL >> 16) & 15u), aML = ((qML >> 20) & 63u) - ((qML >> 16) & 15u), aOF = ((qOF >> 20) & 63u) - ((qOF >> 16) & 15u);
        const u32 bLL = dLL_base[qLL >> 26], bML = dML_base[qML >> 26], bOF = of_base(qOF >> 26);
        s32 p = L.posBase[buf][sl] - (s32)L.recPos[buf][sl][lane];
        const u32 ofv = stream_field(sp, size, p, aOF); p -= (s32)aOF;
        const u32 mlv = stream_field(sp, size, p, aML); p -= (s32)aML;
        const u32 llv = stream_field(sp, size, p, aLL);
        ll = bLL + llv; ml = bML + mlv;
        if (aOF > 1) off = bOF + ofv;
        else code = bOF + (bLL == 0) + ofv;     // ofv is 0 or the single extra bit
    }
    // ---- repcodes, in sequence order (wave-uniform), relative to the block's starting ones ----
    u32 tag = 0;
    RepSlot s0 = S.s0, s1 = S.s1, s2 = S.s2;
    {
        const u64 repMask = ballot(have && code != 4);
        if (!repMask && cnt >= 3) {
            s0.kind = 0; s0.val = read_lane(off, cnt - 1); s1.kind = 0; s1.val = read_lane(off, cnt - 2); s2.kind = 0; s2.val = read_lane(off, cnt - 3);
        } else {
            for (u32 k = 0; k < cnt; k++) {
                const u32 cd = read_lane(code, k);
                if (cd == 4) { const u32 o = read_lane(off, k); s2 = s1; s1 = s0; s0.kind = 0; s0.val = o; }
                else {
                    RepSlot t;
                    if (cd == 0) t = s0;
                    else {
                        t = cd == 1 ? s1 : (cd == 2 ? s2 : rep_minus_one(s0));
                        if (cd != 1) s2 = s1;
                        s1 = s0; s0 = t;
                    }
                    if (lane == k) { off = t.val; tag = t.kind; }
                }
            }
        }
    }
    const u32 totalOut = wave_sum(ll + ml), totalLit = wave_sum(ll);
    if (totalLit > litSize - S.litUsed) { S.err = kErrCorruption; return; }
    if (totalOut > 0xFFFFFFFFu - S.outBase - litSize) { S.err = kErrCorruption; return; }     // (no valid block regenerates 4 GiB)
    if (ballot(have && !tag && off > kRecOffMax)) { S.err = kErrWindowTooLarge; return; }      // beyond what a record holds (windows above 512 MiB)
    if (have) rec[base + lane] = rec_pack(ll, ml, off, tag);
    S.s0 = s0; S.s1 = s1; S.s2 = s2; S.outBase += totalOut; S.litUsed += totalLit;
}

__global__ __launch_bounds__(256) void seq_decode_kernel(const u8* __restrict__ src, const FrameDesc* __restrict__ frames, BlockDesc* __restrict__ blocks,
                                                         u32 nBlocks, SeqRec* __restrict__ recs, u32* __restrict__ status,
                                                         const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ SeqLds L;
    const u32 lane = lane_id(), wave = uniform(wave_id()), b0 = blockIdx.x * kPack;
#ifdef ZMI_LZ_STAMPS
    unsigned long long stampAcc[8] = {0,0,0,0,0,0,0,0}; unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif
    // ---- phase A: the tables, each from the block that defined it (ZSTD_decodeSeqHeaders, :1845-1943); task = (slot, table) ----
    for (u32 task = wave; task < kPack * 3; task += 4) {
        const u32 sl = task / 3, t = task % 3;                      // t: 0 LL, 1 OF, 2 ML
        const u32 bi = b0 + sl;
        u32 res = 0;                                                // tableLog, or an error code << 8 | 0xFF
        if (bi < nBlocks) {
            const BlockDesc& B = blocks[bi];
            if (uniform((u32)B.type) == 2 && uniform(B.nbSeq) != 0 && !uniform(B.err)) {
                const u32 s = uniform(B.tblSrc[t]);
                const u8* d; u32 avail, mode;
                if (s == kDictBlock) {                              // dctx->fseEntropy from the dictionary (U/ZstdDecompress.cs:1956-1990)
                    const u32 o0 = uniform(t == 0 ? di->llOff : t == 1 ? di->ofOff : di->mlOff), o1 = uniform(t == 0 ? di->repOff : t == 1 ? di->mlOff : di->llOff);
                    d = dictFull + o0; avail = o1 - o0; mode = 2;
                } else {
                    const BlockDesc& S = blocks[s];
                    mode = (uniform(S.modes) >> (6 - 2 * t)) & 3;
                    const u32 o = uniform(S.tblOff[t]);
                    d = src + S.srcOff + o; avail = uniform(S.bsz) - o;
                }
                u32 lg;
                if (t == 0)      lg = set_seq_table(L.build.norm[wave], L.build.symbolNext[wave], L.tab[sl] + kTabLL, mode, 35, 9, d, avail, 0, dLL_defaultNorm, 6, 35, lane);
                else if (t == 1) lg = set_seq_table(L.build.norm[wave], L.build.symbolNext[wave], L.tab[sl] + kTabOF, mode, 31, 8, d, avail, 1, dOF_defaultNorm, 5, 28, lane);
                else             lg = set_seq_table(L.build.norm[wave], L.build.symbolNext[wave], L.tab[sl] + kTabML, mode, 52, 9, d, avail, 2, dML_defaultNorm, 6, 52, lane);
                res = lg == 0xFFFFFFFFu ? ((s == kDictBlock ? (u32)kErrDictionaryCorrupted : (u32)kErrCorruption) << 8) | 0xFFu : lg;
            }
        }
        if (lane == 0) L.tblErr[sl][t] = res;
    }
    __syncthreads();
    ZMI_SSTAMP(0);
    // ---- what every wave needs of the blocks (lane s < kPack = slot s) ----
    const u32 myBi = (lane < kPack && b0 + lane < nBlocks) ? b0 + lane : b0;
    BlockDesc& MB = blocks[myBi];
    u32 mNbSeq = 0, mErr = 0, myLogs = 0;
    if (lane < kPack && b0 + lane < nBlocks && MB.type == 2 && MB.nbSeq != 0 && !MB.err) {
        const u32 r0 = L.tblErr[lane][0], r1 = L.tblErr[lane][1], r2 = L.tblErr[lane][2];
        // the reference stops at the first corrupt table in LL, OF, ML order
        if ((r0 & 0xFF) == 0xFF) mErr = r0 >> 8; else if ((r1 & 0xFF) == 0xFF) mErr = r1 >> 8; else if ((r2 & 0xFF) == 0xFF) mErr = r2 >> 8;
        else { mNbSeq = MB.nbSeq; myLogs = r0 | (r1 << 8) | (r2 << 16); }
    }
    const u32 mBsz = MB.bsz, mBitsOff = MB.bitsOff, mLitSize = MB.litSize;
    const u8* const mSp = src + MB.srcOff + mBitsOff;
    const s32 mSize = (s32)(mBsz - mBitsOff);
    const u64 mSeqBase = MB.seqBase;
    s32 pos = 0; u32 sLL = 0, sML = 0, sOF = 0;
    if (mNbSeq) {
        const u32 last = (mBitsOff < mBsz) ? (u32)mSp[mSize - 1] : 0u;
        if (mBitsOff >= mBsz || !last) { mErr = kErrCorruption; mNbSeq = 0; }
        else {
            pos = (mSize - 1) * 8 + (s32)highbit32(last);
            const u32 llLog = myLogs & 0xFF, ofLog = (myLogs >> 8) & 0xFF, mlLog = (myLogs >> 16) & 0xFF;
            sLL = stream_field(mSp, mSize, pos, llLog); pos -= (s32)llLog;
            sOF = stream_field(mSp, mSize, pos, ofLog); pos -= (s32)ofLog;
            sML = stream_field(mSp, mSize, pos, mlLog); pos -= (s32)mlLog;
        }
    }
    const u32 maxSeq = wave_max(mNbSeq);                            // (the same in every wave)
    const u32 nBatch = (maxSeq + 63) >> 6;
    // phase C: wave w (1..3) owns slots w - 1 and w + 2 (the latter only while it exists)
    const u32 slA = wave ? wave - 1 : 0, slB = (wave && wave + 2 < kPack) ? wave + 2 : kPack;      // kPack = none
    SlotState SA = { { 1, 0 }, { 2, 0 }, { 3, 0 }, 0, 0, 0 }, SB = SA;       // the block's starting repcodes, symbolically
    const u32 nbA = read_lane(mNbSeq, slA), nbB = slB < kPack ? read_lane(mNbSeq, slB) : 0u;
    const u8* const spA = reinterpret_cast<const u8*>(read_lane64((u64)(uintptr_t)mSp, slA));
    const u8* const spB = reinterpret_cast<const u8*>(read_lane64((u64)(uintptr_t)mSp, slB));
    const s32 sizeA = (s32)read_lane((u32)mSize, slA), sizeB = (s32)read_lane((u32)mSize, slB);
    const u32 litA = read_lane(mLitSize, slA), litB = read_lane(mLitSize, slB);
    SeqRec* __restrict__ const recA = recs + read_lane64(mSeqBase, slA);
    SeqRec* __restrict__ const recB = recs + read_lane64(mSeqBase, slB);
    const u32* const mT = L.tab[lane < kPack ? lane : 0];
    // ---- the top 2 KiB of every stream into its ring (zeros outside the stream) ----
    s32 loA = 0, loB = 0;                                           // helper waves: lowest staged byte of their blocks (multiples of 4)
    auto stage = [&](u32 sl, const u8* sp, s32 size, s32 from, s32 to) {       // bytes [from, to), multiples of 4
        for (s32 x = from + 4 * (s32)lane; x < to; x += 256) {
            const u32 ix = (u32)(x >> 2) & (kRingDw - 1), v = stream_dword_z(sp, size, x >> 2);
            L.ring[sl][ix] = v;
            if (ix < 3) L.ring[sl][kRingDw + ix] = v;
        }
    };
    if (wave) {
        if (nbA) { const s32 hi = (sizeA + 3) & ~3; loA = hi - (s32)kRing; stage(slA, spA, sizeA, loA, hi); }
        if (nbB) { const s32 hi = (sizeB + 3) & ~3; loB = hi - (s32)kRing; stage(slB, spB, sizeB, loB, hi); }
        if (lane == 0) { L.ringLo[0][slA] = loA; if (slB < kPack) L.ringLo[0][slB] = loB; }
    } else if (lane < kPack) L.curByte[0][lane] = pos >> 3;
    __syncthreads();
    for (u32 bt = 0; bt <= nBatch; ++bt) {
        const u32 base = bt << 6;
        if (wave == 0) {
            // ---- phase B: 64 steps of every chain.  Per sequence a lane reads its three table entries (LDS) and, independently of
            // them, the 128 stream bits below its bit position (global memory; a stream is walked downward, so these loads stay in
            // one cache line for a dozen sequences); the extra-bit fields are only skipped here ----
            // (this wave's dependent chain competes for issue slots with the helper waves of the workgroups sharing its SIMD: it goes first)
            __builtin_amdgcn_s_setprio(3);
            if (base < mNbSeq) {
                const u32 steps = mNbSeq - base < 64 ? mNbSeq - base : 64, buf = bt & 1;
                const s32 ringLo = L.ringLo[buf][lane], pos0 = pos;
                const u32* const R = L.ring[lane];
                L.posBase[buf][lane] = pos0;
                // The window: stream dwords dl .. dl + 3, at least bits [pos - 97, pos), out of the ring.  (Staged, always: a valid
                // stream never reaches more than 16 bytes below its start, and the ring is staged down to 64 below.  A damaged stream
                // that runs away further reads the lowest staged dwords instead — in bounds, wrong bits, states stay inside their
                // tables by construction; the fields phase sees the negative positions and reports corruption, as it did when this
                // case read zeros.)  A step's seven reads are issued at the END of the step before it, as soon as the states and the
                // position they depend on exist: both LDS latencies of a step overlap instead of following each other.
                s32 dl = ((pos - 1) >> 5) - 3;
                const u32* Rw = R + ((u32)(4 * dl >= ringLo ? dl : (ringLo >> 2)) & (kRingDw - 1));
                u32 w0 = Rw[0], w1 = Rw[1], w2 = Rw[2], w3 = Rw[3];
                u32 eLL = mT[kTabLL + sLL], eML = mT[kTabML + sML], eOF = mT[kTabOF + sOF];
                for (u32 k = 0; k < steps; ++k) {
                    L.recSt[buf][lane][k] = sLL | (sML << 10) | (sOF << 20); L.recPos[buf][lane][k] = (u16)(pos0 - pos);
                    const u32 nLL = (eLL >> 16) & 15u, nML = (eML >> 16) & 15u, nOF = (eOF >> 16) & 15u, nbTot = nLL + nML + nOF;
                    const s32 q = pos - (s32)(((eLL >> 20) & 63u) + ((eML >> 20) & 63u) + ((eOF >> 20) & 63u));
                    const u32 rr = (u32)(q - 32 * dl), ix = rr >> 5;     // q >= pos - 89 >= 32 dl + 8
                    const u32 lo = ix == 0 ? w0 : ix == 1 ? w1 : ix == 2 ? w2 : w3;
                    const u32 hi = ix == 0 ? w1 : ix == 1 ? w2 : ix == 2 ? w3 : 0u;
                    const u32 all = __builtin_amdgcn_alignbit(hi, lo, rr & 31u) & ~(0xFFFFFFFFu << nbTot);
                    sLL = (eLL & 0xFFFFu) + (all >> (nML + nOF));
                    sML = (eML & 0xFFFFu) + ((all >> nOF) & ~(0xFFFFFFFFu << nML));
                    sOF = (eOF & 0xFFFFu) + (all & ~(0xFFFFFFFFu << nOF));
                    pos = q;
                    // the next step's reads (harmless after the last one: states and window stay in bounds)
                    dl = ((pos - 1) >> 5) - 3;
                    Rw = R + ((u32)(4 * dl >= ringLo ? dl : (ringLo >> 2)) & (kRingDw - 1));
                    w0 = Rw[0]; w1 = Rw[1]; w2 = Rw[2]; w3 = Rw[3];
                    eLL = mT[kTabLL + sLL]; eML = mT[kTabML + sML]; eOF = mT[kTabOF + sOF];
                }
            }
            __builtin_amdgcn_s_setprio(0);
            if (lane < kPack) L.curByte[(bt + 1) & 1][lane] = pos >> 3;
        } else {
            // ---- phase C for the previous 64 sequences of this wave's blocks ----
            if (bt) {
                const u32 pbase = base - 64, buf = (bt - 1) & 1;
                if (pbase < nbA && !SA.err) seq_fields_batch(L, buf, slA, pbase, nbA, spA, sizeA, litA, recA, SA, lane);
                if (pbase < nbB && !SB.err) seq_fields_batch(L, buf, slB, pbase, nbB, spB, sizeB, litB, recB, SB, lane);
            }
            // ---- their streams: stage down to 2 KiB below where the chain stands now.  What that overwrites in the ring lies 16
            // bytes or more above the chain's position (dead: it only moves down); a batch consumes at most 712 bytes and the
            // chain reaches 16 below its position, so the NEXT batch finds everything it can touch staged by this one ----
            if (nbA) { s32 nl = (L.curByte[bt & 1][slA] + 16 - (s32)kRing + 3) & ~3; if (nl < -64) nl = -64; if (nl < loA) { stage(slA, spA, sizeA, nl, loA); loA = nl; } }
            if (nbB) { s32 nl = (L.curByte[bt & 1][slB] + 16 - (s32)kRing + 3) & ~3; if (nl < -64) nl = -64; if (nl < loB) { stage(slB, spB, sizeB, nl, loB); loB = nl; } }
            if (lane == 0) { L.ringLo[(bt + 1) & 1][slA] = loA; if (slB < kPack) L.ringLo[(bt + 1) & 1][slB] = loB; }
        }
        __syncthreads();
    }
    ZMI_SSTAMP(1);
    // ---- results ----
    if (wave == 0 && lane < kPack) { L.chainErr[lane] = mErr; L.endPos[lane] = pos; }
    __syncthreads();
    if (wave && lane == 0) {
#pragma unroll
        for (u32 h = 0; h < 2; ++h) {
            const u32 sl = h ? slB : slA; const SlotState& S = h ? SB : SA;
            const u32 bi = b0 + sl;
            if (sl >= kPack || bi >= nBlocks) continue;
            BlockDesc& B = blocks[bi];
            if (B.type != 2 || B.nbSeq == 0 || B.err) continue;
            u32 err = L.chainErr[sl];                               // tables, end mark
            if (!err) err = S.err;
            if (!err && L.endPos[sl] > 0) err = kErrCorruption;     // bitstream not fully consumed (:2730-2733)
            if (err) { B.err = err; report_error(status, bi, kStageSequences, err); continue; }
            B.outSize = S.outBase + (B.litSize - S.litUsed);
            B.repKind[0] = S.s0.kind; B.repVal[0] = S.s0.val; B.repKind[1] = S.s1.kind; B.repVal[1] = S.s1.val; B.repKind[2] = S.s2.kind; B.repVal[2] = S.s2.val;
        }
    }
    ZMI_SSTAMP(2);
#ifdef ZMI_LZ_STAMPS
    if (wave == 0 && lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_seqStamps[i], stampAcc[i]);
#endif
}

void launch_seq_decode(const u8* src, const FrameDesc* frames, BlockDesc* blocks, u32 nBlocks, SeqRec* recs, u32* status,
                       const u8* dictFull, const DictInfo* di, hipStream_t stream)
{
    hipLaunchKernelGGL(seq_decode_kernel, dim3((nBlocks + kPack - 1) / kPack), dim3(256), 0, stream, src, frames, blocks, nBlocks, recs, status, dictFull, di);
}

// ------------------------------------------------------------------------------------------------
// place_literals: everything that is not a match goes to its final place; one wave per block
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void place_literals_kernel(const u8* __restrict__ src, u8* __restrict__ out, const u8* __restrict__ scratch,
                                                            const FrameDesc* __restrict__ frames, const BlockDesc* __restrict__ blocks, u32 nBlocks,
                                                            const SeqRec* __restrict__ recs, const u32* __restrict__ status)
{
    const u32 bi = blockIdx.x, lane = threadIdx.x;
    if (bi >= nBlocks || status[kStErr]) return;
    const BlockDesc& B = blocks[bi];
    const FrameDesc& F = frames[B.frame];
    if (F.bad || B.err) return;
    u8* __restrict__ const o = out + F.dstOff + B.dstRel;
    const u32 type = uniform((u32)B.type);
    if (type == 0) { wave_copy(o, src + B.srcOff, uniform(B.bsz), lane); return; }           // ZSTD_copyRawBlock, U/ZstdDecompress.cs:1004-1027
    if (type == 1) {                                                                          // ZSTD_setRleBlock, :1029-1052
        const u32 n = uniform(B.outSize); const u8 v = src[B.srcOff];
        const u64 v8 = 0x0101010101010101ull * v;
        for (u32 i = lane * 8; i + 8 <= n; i += 512) *(u64u*)(o + i) = v8;
        if (lane < (n & 7)) o[(n & ~7u) + lane] = v;
        return;
    }
    const u32 litType = uniform((u32)B.litType), litSize = uniform(B.litSize), nbSeq = uniform(B.nbSeq);
    if (uniform(B.litInPlace)) return;                          // decoded in place by the literal decoder
    const u8* __restrict__ lit = nullptr; u32 rleByte = 0; const bool litIsRle = litType == 1;
    if (litType >= 2) lit = scratch + F.scratchOff + B.litRel + (u64)B.frame * kLitSkew;
    else if (litType == 0) lit = src + B.srcOff + B.lhSize;
    else rleByte = src[B.srcOff + B.lhSize];
    u32 litPos = 0, outEnd = 0;
    const SeqRec* __restrict__ const rec = recs + B.seqBase;
    SeqRec rNext; rNext.lo = 0; rNext.hi = 0;
    if (lane < nbSeq) rNext = rec[lane];                        // records come one batch ahead
    for (u32 base = 0; base < nbSeq; base += 64) {
        const u32 cnt = nbSeq - base < 64 ? nbSeq - base : 64;
        const bool have = lane < cnt;
        const SeqRec r = rNext;
        if (base + 64 + lane < nbSeq) rNext = rec[base + 64 + lane];
        SeqLane q;
        const u32 outNext = seq_batch(r, have, 1, 1, 1, outEnd, q);     // (offsets are of no interest here)
        const u32 ll = q.ll, pos = q.pos;
        const u32 inclLit = wave_scan_incl(ll);
        const u32 totalLit = read_lane(inclLit, 63);
        const u32 sLit = litPos + inclLit - ll;
        outEnd = outNext;
        // literals: short runs by their own lane; long runs are cut into 16-byte pieces (the last one overlapping the one before,
        // so every piece is whole) and ALL pieces of the batch are dealt to the lanes round-robin, four in flight per lane: the
        // copy is paced by bandwidth, not by one load-store round trip per run
        const bool longLit = ll > 32;
        if (have && !longLit) {
            if (litIsRle) for (u32 i = 0; i < ll; i++) o[pos + i] = (u8)rleByte;
            else lane_copy(o + pos, lit + sLit, ll);
        }
        if (litIsRle) {
            u64 lm = ballot(have && longLit);
            while (lm) {
                const u32 i = ctz64(lm); lm &= lm - 1;
                const u32 d0 = read_lane(pos, i), n0 = read_lane(ll, i);
                for (u32 k2 = lane; k2 < n0; k2 += 64) o[d0 + k2] = (u8)rleByte;
            }
        } else if (ballot(have && longLit)) {
            const u32 pc = (have && longLit) ? (ll + 15) >> 4 : 0;
            const u32 pIncl = wave_scan_incl(pc), pExcl = pIncl - pc;
            const u32 P = read_lane(pIncl, 63);
            for (u32 q0 = 0; q0 < P; q0 += 256) {
                u64 a[4], b2[4]; u32 dd[4]; bool okp[4];
#pragma unroll
                for (u32 t = 0; t < 4; ++t) {
                    const u32 q = q0 + t * 64 + lane;
                    okp[t] = q < P;
                    u32 j = 0;                       // the run that owns piece q: first lane whose inclusive count exceeds q
#pragma unroll
                    for (u32 st = 32; st; st >>= 1) { const u32 v = __shfl(pIncl, (int)(j + st - 1)); if (v <= q) j += st; }
                    const u32 nj = __shfl(ll, (int)j), sj = __shfl(sLit, (int)j), dj = __shfl(pos, (int)j), ej = __shfl(pExcl, (int)j);
                    u32 oo = 16 * (q - ej); if (oo + 16 > nj) oo = nj - 16;
                    a[t] = 0; b2[t] = 0; dd[t] = dj + oo;
                    if (okp[t]) { a[t] = readLE64(lit + sj + oo); b2[t] = readLE64(lit + sj + oo + 8); }
                }
#pragma unroll
                for (u32 t = 0; t < 4; ++t) if (okp[t]) { *(u64u*)(o + dd[t]) = a[t]; *(u64u*)(o + dd[t] + 8) = b2[t]; }
            }
        }
        litPos += totalLit;
    }
    {   // trailing literals (:2747-2760); for a block without sequences: all of them
        const u32 lastLL = litSize - litPos;
        if (litIsRle) { for (u32 i = lane; i < lastLL; i += 64) o[outEnd + i] = (u8)rleByte; }
        else wave_copy(o + outEnd, lit + litPos, lastLL, lane);
    }
}

void launch_place_literals(const u8* src, u8* out, const u8* scratch, const FrameDesc* frames, const BlockDesc* blocks, u32 nBlocks,
                           const SeqRec* recs, const u32* status, hipStream_t stream)
{
    hipLaunchKernelGGL(place_literals_kernel, dim3(nBlocks), dim3(64), 0, stream, src, out, scratch, frames, blocks, nBlocks, recs, status);
}

// ------------------------------------------------------------------------------------------------
// exec_matches: the ordered stage, one wave per frame
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void exec_matches_kernel(const u8* __restrict__ src, u8* __restrict__ out, const FrameDesc* __restrict__ frames,
                                                          const BlockDesc* __restrict__ blocks, u32 nFrames, const SeqRec* __restrict__ recs,
                                                          u32* __restrict__ status, const u8* __restrict__ dict, const u32 dictSize)
{
    const u32 f = blockIdx.x, lane = threadIdx.x;
    if (f >= nFrames || status[kStErr]) return;
    const FrameDesc& F = frames[f];
    if (F.bad || !(F.hasSeq || F.checksum)) return;
    const bool viaOrigin = uniform(F.viaOrigin) != 0;           // its matches are in place already (decode_origin.hip): the checksum is what is left
    if (viaOrigin && !F.checksum) return;
    u8* const fout = out + F.dstOff;
    const u32 first = uniform(F.firstBlock), nb = uniform(F.nbBlocks);
#ifdef ZMI_LZ_STAMPS
    unsigned long long stampAcc[8] = {0,0,0,0,0,0,0,0}; unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif
    u32 err = 0, errBlock = first;
    for (u32 k = 0; k < nb && !err && F.hasSeq && !viaOrigin; ++k) {
        const BlockDesc& B = blocks[first + k];
        const u32 nbSeq = uniform(B.type == 2 ? B.nbSeq : 0u);
        if (!nbSeq) continue;
        const u64 bRel = B.dstRel;                               // frame-relative start of the block's output
        u8* const o = fout + bRel;
        const u32 in0 = uniform(B.repIn[0]), in1 = uniform(B.repIn[1]), in2 = uniform(B.repIn[2]);
        const SeqRec* __restrict__ const rec = recs + B.seqBase;
        // output of earlier blocks (other waves' literals included: kernel boundary) and of this wave so far is visible
        SeqRec rNext; rNext.lo = 0; rNext.hi = 0;
        if (lane < nbSeq) rNext = rec[lane];                    // records come one batch ahead: their load rides under the copies
        u32 outBase = 0;
        for (u32 base = 0; base < nbSeq; base += 64) {
            const u32 cnt = nbSeq - base < 64 ? nbSeq - base : 64;
            const bool have = lane < cnt;
            const SeqRec r = rNext;
            if (base + 64 + lane < nbSeq) rNext = rec[base + 64 + lane];
            SeqLane q;
            outBase = seq_batch(r, have, in0, in1, in2, outBase, q);
            const u32 ll = q.ll, ml = q.ml, pos = q.pos, off = q.off;
            ZMI_SSTAMP(0);
            const u32 dMatch = pos + ll;                          // block-relative start of my match
            // offset beyond everything produced so far in this frame (+ the dictionary): corruption (:2218-2223)
            if (ballot(have && ((u64)off > bRel + dMatch + dictSize || off == 0))) { err = kErrCorruption; errBlock = first + k; break; }
            // ---- matches, in dependency rounds ----
            // A match may start once every earlier match of this batch whose output its source touches is complete (literals and
            // everything before the batch already are).  Each round runs all matches that are ready: short ones by their own
            // lane, long ones by the whole wave, then one fence.  The number of rounds is the depth of the dependency chain.
            u32 dictN = 0;
            if (dictSize) {                                 // uniform
                // a match that starts in the dictionary (ZSTD_execSequence's extDict branch, :2223-2250): its first bytes come from
                // the dictionary's tail (read-only, no dependency), the rest is an ordinary match whose source is the frame's start
                if (have && (u64)off > bRel + dMatch) {
                    const u32 back = (u32)((u64)off - (bRel + dMatch));
                    dictN = back < ml ? back : ml;
                    const u8* ds = dict + (dictSize - back);
                    for (u32 i = 0; i < dictN; i++) o[dMatch + i] = ds[i];
                }
            }
            const u32 dMatchR = dMatch + dictN, mlR = ml - dictN;      // what remains for the rounds
            const bool hasMatch = have && mlR != 0;
            // block-relative source interval; negative = in earlier blocks (complete)
            const s64 srcLo = (s64)dMatchR - (s64)off, srcHi = srcLo + (s64)(off < mlR ? off : mlR);
            // earlier lanes whose match output overlaps my source: outputs are laid out in lane order, so they form a lane
            // interval [jl, jh) found by two binary searches over the (monotone) per-lane bounds
            const u64 mm = ballot(hasMatch);
            const s64 endOutX = have ? (s64)dMatch + ml : (s64)0x7FFFFFFFFFFFll, dMatchX = have ? (s64)dMatch : (s64)0x7FFFFFFFFFFFll;
            u32 jl = 0, jh = 0;
#pragma unroll
            for (u32 st = 32; st; st >>= 1) {
                const s64 v0 = __shfl(endOutX, (int)(jl + st - 1)), v1 = __shfl(dMatchX, (int)(jh + st - 1));
                if (v0 <= srcLo) jl += st;
                if (v1 < srcHi) jh += st;
            }
            const u64 dep = (jh > jl ? ((~0ull >> (64 - (jh - jl))) << jl) : 0ull) & mm & lanemask_lt();
            u64 doneMask = ~mm;                            // lanes without a match never block anyone
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // output of earlier batches visible
            bool mine = hasMatch;
            ZMI_SSTAMP(1);
            while (doneMask != ~0ull) {
                const bool ready = mine && (dep & ~doneMask) == 0;
                const bool longM = mlR > 64;
                if (ready && !longM) lane_match_copy(o + dMatchR, off, mlR);
                u64 lm = ballot(ready && longM);
                while (lm) {
                    const u32 i = ctz64(lm); lm &= lm - 1;
                    wave_match_copy(o + read_lane(dMatchR, i), read_lane(off, i), read_lane(mlR, i), lane);
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
            ZMI_SSTAMP(2);
        }
    }
#ifdef ZMI_LZ_STAMPS
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_seqStamps[8 + i], stampAcc[i]);
#endif
    if (err) { if (lane == 0) report_error(status, errBlock, kStageExec, err); return; }
    if (F.checksum) {
        // XXH64 of the regenerated frame: accumulators on lanes 0..3 (U/ZstdDecompress.cs:1186-1208)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const u64 P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
        auto rotl = [](u64 x, int r) { return (x << r) | (x >> (64 - r)); };
        auto rnd = [&](u64 acc, u64 in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; };
        const u64 n = F.dstSize, stripes = n >> 5; const u32 j = lane & 3;
        u64 v = j == 0 ? P1 + P2 : j == 1 ? P2 : j == 2 ? 0 : 0 - P1;
        if (lane < 4) for (u64 i = 0; i < stripes; i++) v = rnd(v, readLE64(fout + 32 * i + 8 * j));
        const u64 v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        u32 bad = 0;
        if (lane == 0) {
            u64 hh;
            if (n >= 32) {
                hh = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
                auto mrg = [&](u64 acc, u64 x) { acc ^= rnd(0, x); return acc * P1 + P4; };
                hh = mrg(hh, v1); hh = mrg(hh, v2); hh = mrg(hh, v3); hh = mrg(hh, v4);
            } else hh = P5;
            hh += n;
            const u8* q = fout + (stripes << 5); const u8* const end = fout + n;
            while (q + 8 <= end) { hh ^= rnd(0, readLE64(q)); hh = rotl(hh, 27) * P1 + P4; q += 8; }
            if (q + 4 <= end) { hh ^= (u64)readLE32(q) * P1; hh = rotl(hh, 23) * P2 + P3; q += 4; }
            while (q < end) { hh ^= (*q) * P5; hh = rotl(hh, 11) * P1; q++; }
            hh ^= hh >> 33; hh *= P2; hh ^= hh >> 29; hh *= P3; hh ^= hh >> 32;
            if ((u32)hh != readLE32(src + F.srcOff + F.srcSize - 4)) bad = 1;
        }
        if (uniform(bad) && lane == 0) report_error(status, (u64)first + nb - 1, kStageFrameEnd, kErrChecksumWrong);
    }
}

// ------------------------------------------------------------------------------------------------
// exec_matches on FEW frames: W waves per frame.  With one wave per frame what a frame costs is its number of dependency rounds
// (a round = stores acknowledged, barrier, loads: about a microsecond, whatever it copies), and 1024 frames of 1 MiB leave three
// quarters of the chip's wave slots empty.  Here a batch is 64 x W sequences, one per thread of a W-wave workgroup: the chain of
// matches that read each other's output grows far slower than the batch (oracle level-5 text, rounds per sequence: 0.050 at
// 64, 0.027 at 256, 0.021 at 512), so a frame needs about half the rounds at W = 4, two fifths at W = 8.
// Same rules as above: outputs lie in sequence order, a match depends on the earlier matches of its batch whose output its source
// touches (an index interval, found by binary search over the batch's bounds in LDS), a round runs every match whose interval is
// complete.  The done bits are double-buffered by round parity, so one barrier per round is enough.
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64 * W) void exec_matches_wide_kernel(const u8* __restrict__ src, u8* __restrict__ out, const FrameDesc* __restrict__ frames,
                                                                   const BlockDesc* __restrict__ blocks, u32 nFrames, const SeqRec* __restrict__ recs,
                                                                   u32* __restrict__ status, const u8* __restrict__ dict, const u32 dictSize)
{
    constexpr u32 kB = 64u * W;                                  // sequences per batch
    __shared__ u32 endOfOut[kB], startOfOut[kB];                 // block-relative bounds of the batch's match outputs (monotone; 0xFFFFFFFF: no sequence)
    __shared__ u64 doneBits[2][W];
    __shared__ u32 waveSum[W];
    __shared__ u32 errFlag;
    const u32 f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = uniform(tid >> 6);
    if (f >= nFrames || status[kStErr]) return;
    const FrameDesc& F = frames[f];
    if (F.bad || !(F.hasSeq || F.checksum)) return;
    const bool viaOrigin = uniform(F.viaOrigin) != 0;
    if (viaOrigin && !F.checksum) return;
    u8* const fout = out + F.dstOff;
    const u32 first = uniform(F.firstBlock), nb = uniform(F.nbBlocks);
    if (tid == 0) errFlag = 0;
    __syncthreads();
    u32 err = 0, errBlock = first;
    for (u32 k = 0; k < nb && !err && F.hasSeq && !viaOrigin; ++k) {
        const BlockDesc& B = blocks[first + k];
        const u32 nbSeq = uniform(B.type == 2 ? B.nbSeq : 0u);
        if (!nbSeq) continue;
        const u64 bRel = B.dstRel;
        u8* const o = fout + bRel;
        const u32 in0 = uniform(B.repIn[0]), in1 = uniform(B.repIn[1]), in2 = uniform(B.repIn[2]);
        const SeqRec* __restrict__ const rec = recs + B.seqBase;
        SeqRec rNext; rNext.lo = 0; rNext.hi = 0;
        if (tid < nbSeq) rNext = rec[tid];
        u32 outBase = 0;
        for (u32 base = 0; base < nbSeq; base += kB) {
            const bool have = base + tid < nbSeq;
            const SeqRec r = rNext;
            if (base + kB + tid < nbSeq) rNext = rec[base + kB + tid];
            u32 ll = 0, ml = 0, off = 1;
            if (have) {
                u32 tag; rec_unpack(r, ll, ml, off, tag);
                if (tag) { const u32 in = tag == 1 ? in0 : tag == 2 ? in1 : in2; off = in > off ? in - off : 1u; }
            }
            const u32 incl = wave_scan_incl(ll + ml);
            if (lane == 63) waveSum[wave] = incl;
            __syncthreads();                                     // (also: nobody still reads the previous batch's bounds or done bits)
            u32 before = 0, total = 0;
#pragma unroll
            for (u32 w = 0; w < (u32)W; ++w) { const u32 v = waveSum[w]; if (w < wave) before += v; total += v; }
            const u32 pos = outBase + before + incl - ll - ml;
            outBase += total;
            const u32 dMatch = pos + ll;
            if (have && ((u64)off > bRel + dMatch + dictSize || off == 0)) errFlag = 1;       // (:2218-2223)
            u32 dictN = 0;
            if (dictSize) {                                      // uniform: a match that starts in the dictionary (:2223-2250)
                if (have && (u64)off > bRel + dMatch && (u64)off <= bRel + dMatch + dictSize) {
                    const u32 back = (u32)((u64)off - (bRel + dMatch));
                    dictN = back < ml ? back : ml;
                    const u8* ds = dict + (dictSize - back);
                    for (u32 i = 0; i < dictN; i++) o[dMatch + i] = ds[i];
                }
            }
            const u32 dMatchR = dMatch + dictN, mlR = ml - dictN;
            const bool hasMatch = have && mlR != 0;
            const s64 srcLo = (s64)dMatchR - (s64)off, srcHi = srcLo + (s64)(off < mlR ? off : mlR);
            endOfOut[tid] = have ? dMatch + ml : 0xFFFFFFFFu; startOfOut[tid] = have ? dMatch : 0xFFFFFFFFu;
            {
                const u64 mm = ballot(hasMatch);
                if (lane == 0) doneBits[0][wave] = ~mm;           // lanes without a match never block anyone
            }
            __syncthreads();                                     // bounds, done bits, errFlag; output of earlier batches visible
            if (errFlag) { err = kErrCorruption; errBlock = first + k; break; }      // uniform
            // earlier sequences of the batch whose match output overlaps my source: [jl, jh)
            u32 jl = 0, jh = 0;
            if (hasMatch && srcHi > 0) {
                const u32 lo32 = srcLo > 0 ? (u32)srcLo : 0u, hi32 = (u32)srcHi;
#pragma unroll
                for (u32 st = kB >> 1; st; st >>= 1) {
                    if (endOfOut[jl + st - 1] <= lo32) jl += st;
                    if (startOfOut[jh + st - 1] < hi32) jh += st;
                }
                if (jh > tid) jh = tid;
            }
            bool mine = hasMatch;
            const bool longM = mlR > 64;
#if defined(ZMI_EXP_EXEC) && ZMI_EXP_EXEC == 1
            continue;                                            // (ablation build: no rounds, no copies)
#endif
            for (u32 round = 0; ; ++round) {
                const u64* const D = doneBits[round & 1];
                u64 all = ~0ull, own = 0;
#pragma unroll
                for (u32 w = 0; w < (u32)W; ++w) { const u64 d = D[w]; all &= d; if (w == wave) own = d; }
                if (all == ~0ull) break;                         // uniform: everybody reads the same words
                bool ready = mine;
                if (ready && jh > jl) {
                    for (u32 w = jl >> 6; w <= ((jh - 1) >> 6); ++w) {
                        u64 bits = ~0ull;
                        if (w == (jl >> 6)) bits &= ~0ull << (jl & 63u);
                        if (w == ((jh - 1) >> 6)) bits &= ~0ull >> (63u - ((jh - 1) & 63u));
                        if (bits & ~D[w]) { ready = false; break; }
                    }
                }
#if defined(ZMI_EXP_EXEC) && ZMI_EXP_EXEC == 2
                u64 lm = 0;                                      // (ablation build: the rounds without their copies)
#else
                if (ready && !longM) lane_match_copy(o + dMatchR, off, mlR);
                u64 lm = ballot(ready && longM);
#endif
                while (lm) {
                    const u32 i = ctz64(lm); lm &= lm - 1;
                    wave_match_copy(o + read_lane(dMatchR, i), read_lane(off, i), read_lane(mlR, i), lane);
                }
                const u64 rdy = ballot(ready);
                mine = mine && !ready;
                if (lane == 0) doneBits[(round + 1) & 1][wave] = own | rdy;
                __syncthreads();                                 // this round's bytes are visible to the workgroup
            }
        }
    }
    if (err) { if (tid == 0) report_error(status, errBlock, kStageExec, err); return; }
    if (F.checksum && wave == 0) {
        // XXH64 of the regenerated frame, as in the one-wave kernel (U/ZstdDecompress.cs:1186-1208)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        const u64 P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL, P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
        auto rotl = [](u64 x, int r) { return (x << r) | (x >> (64 - r)); };
        auto rnd = [&](u64 acc, u64 in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; };
        const u64 n = F.dstSize, stripes = n >> 5; const u32 j = lane & 3;
        u64 v = j == 0 ? P1 + P2 : j == 1 ? P2 : j == 2 ? 0 : 0 - P1;
        if (lane < 4) for (u64 i = 0; i < stripes; i++) v = rnd(v, readLE64(fout + 32 * i + 8 * j));
        const u64 v1 = __shfl(v, 0), v2 = __shfl(v, 1), v3 = __shfl(v, 2), v4 = __shfl(v, 3);
        u32 bad = 0;
        if (lane == 0) {
            u64 hh;
            if (n >= 32) {
                hh = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
                auto mrg = [&](u64 acc, u64 x) { acc ^= rnd(0, x); return acc * P1 + P4; };
                hh = mrg(hh, v1); hh = mrg(hh, v2); hh = mrg(hh, v3); hh = mrg(hh, v4);
            } else hh = P5;
            hh += n;
            const u8* q = fout + (stripes << 5); const u8* const end = fout + n;
            while (q + 8 <= end) { hh ^= rnd(0, readLE64(q)); hh = rotl(hh, 27) * P1 + P4; q += 8; }
            if (q + 4 <= end) { hh ^= (u64)readLE32(q) * P1; hh = rotl(hh, 23) * P2 + P3; q += 4; }
            while (q < end) { hh ^= (*q) * P5; hh = rotl(hh, 11) * P1; q++; }
            hh ^= hh >> 33; hh *= P2; hh ^= hh >> 29; hh *= P3; hh ^= hh >> 32;
            if ((u32)hh != readLE32(src + F.srcOff + F.srcSize - 4)) bad = 1;
        }
        if (uniform(bad) && lane == 0) report_error(status, (u64)first + nb - 1, kStageFrameEnd, kErrChecksumWrong);
    }
}

// wide: waves per frame.  Measured on 1 MiB level-5 frames (exec_matches, ms; tools/exec_waves_time.py): 256 frames: 1 wave 5.0,
// 16 waves 2.4; 1024 frames: 6.8 / 5.5 / 4.3 / 5.7 / 7.3 at 1 / 2 / 4 / 8 / 16; 2048 frames: 8.3 / 6.5 / 8.0 at 1 / 2 / 4; 4096
// frames: 9.8 / 12.3 at 1 / 2 — i.e. as many waves as keep frames x waves at about 4096 (the caller's rule, decompress_device)
void launch_exec_matches(const u8* src, u8* out, const FrameDesc* frames, const BlockDesc* blocks, u32 nFrames, const SeqRec* recs, u32* status,
                         const u8* dict, u32 dictSize, hipStream_t stream, int wide)
{
    const u32 ds = dict ? dictSize : 0u;
    switch (wide) {
    case 16: hipLaunchKernelGGL(exec_matches_wide_kernel<16>, dim3(nFrames), dim3(1024), 0, stream, src, out, frames, blocks, nFrames, recs, status, dict, ds); return;
    case 8:  hipLaunchKernelGGL(exec_matches_wide_kernel<8>,  dim3(nFrames), dim3(512),  0, stream, src, out, frames, blocks, nFrames, recs, status, dict, ds); return;
    case 4:  hipLaunchKernelGGL(exec_matches_wide_kernel<4>,  dim3(nFrames), dim3(256),  0, stream, src, out, frames, blocks, nFrames, recs, status, dict, ds); return;
    case 2:  hipLaunchKernelGGL(exec_matches_wide_kernel<2>,  dim3(nFrames), dim3(128),  0, stream, src, out, frames, blocks, nFrames, recs, status, dict, ds); return;
    default: break;
    }
    // dict: a raw-content dictionary (or a formatted one's content) = history in front of EVERY frame (ZSTD_refDictContent,
    // U/ZstdDecompress.cs:1758-1771); may be null
    hipLaunchKernelGGL(exec_matches_kernel, dim3(nFrames), dim3(64), 0, stream, src, out, frames, blocks, nFrames, recs, status, dict, dict ? dictSize : 0u);
}

} // namespace zmi
