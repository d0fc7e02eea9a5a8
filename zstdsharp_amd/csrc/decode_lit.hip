// decode_lit.hip — Huffman literal decoders on gfx950 (SURVEY.md §8 a-15, a-16's weights).
//
// Work item = one compressed block whose literals section is Huffman-coded (block_parse found it, block_link resolved where a
// treeless section's table comes from: an earlier block of the frame or the formatted dictionary, U/ZstdDecompressBlock.cs:197-207).
// Per block: Huffman weights + X1 table in LDS (HUF_readStats U/EntropyCommon.cs:292-402, HUF_readDTableX1
// U/HufDecompress.cs:80-251), then the literal streams (U/HufDecompress.cs:342-537).  The regenerated literals go to the
// literal scratch, or straight to their place in the output when the block has no sequences (its literals ARE its output).
// Three forms + a catch-all, see launch_decode_literals:
//   serial   : 8 blocks per wave, a stream per lane, 4 KiB table per block;
//   compact  : the same with a 2 KiB table (twice the blocks in flight per CU);
//   selfsync : a workgroup per block, 64 lanes per stream;
//   slow     : one wave per block, everything the others pass on (12-bit tables).
#include "zmi_decode.h"

namespace zmi {

// what a literal decoder needs to know about its block (computed redundantly by every lane that works on the block)
struct LitJob {
    const u8* tsrc; u32 tlen;       // Huffman tree description: this block's, an earlier block's, or the dictionary's
    bool own;                       // the description sits in front of this block's own streams
    u32 tErr;                       // what a bad description is called (corruption_detected, or dictionary_corrupted for the dictionary's)
    const u8* hsrc; u32 hlen;       // this block's compressed literals (behind the section header; description included when own)
    u8* dst; u32 litSize; u32 single;
};
__device__ __forceinline__ bool lit_job(LitJob& J, u32 bi, const BlockDesc* __restrict__ blocks, const FrameDesc* __restrict__ frames,
                                        const u8* __restrict__ src, u8* __restrict__ out, u8* __restrict__ scratch,
                                        const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    const BlockDesc& B = blocks[bi];
    if (B.type != 2 || B.litType < 2 || B.err) return false;
    const FrameDesc& F = frames[B.frame];
    if (F.bad) return false;
    J.hsrc = src + B.srcOff + B.lhSize; J.hlen = B.litCSize; J.litSize = B.litSize; J.single = B.litSingle;
    J.own = B.hufSrc == bi; J.tErr = kErrCorruption;
    if (J.own) { J.tsrc = J.hsrc; J.tlen = J.hlen; }
    else if (B.hufSrc == kDictBlock) { J.tsrc = dictFull + di->hufOff; J.tlen = di->hufSize; J.tErr = kErrDictionaryCorrupted; }
    else { const BlockDesc& S = blocks[B.hufSrc]; J.tsrc = src + S.srcOff + S.lhSize; J.tlen = S.litCSize; }
    // a block without sequences regenerates exactly its literals: they are decoded in place
    J.dst = B.litInPlace ? out + F.dstOff + B.dstRel : scratch + F.scratchOff + B.litRel + (u64)B.frame * kLitSkew;
    return true;
}

#ifdef ZMI_LZ_STAMPS
__device__ unsigned long long g_litStamps[16];      // diagnostic build only: passes / streams / cycles of the self-synchronising decoder
extern "C" void ZSTDMI_debugReadLitStamps(unsigned long long* out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_litStamps), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_litStamps), z, sizeof z); }
}
#endif

// per-wave decoder state in LDS (slow path)
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

#ifndef ZMI_LIT_QUADSTORE
#define ZMI_LIT_QUADSTORE 1
#endif
// true iff the predicate holds on all 4 lanes of this lane's quad (the quad's lanes are converged here)
__device__ __forceinline__ bool quad_all(bool p, u32 ql)
{
    const u64 b = ballot(p);
    const u32 base = lane_id() & ~3u;
    (void)ql;
    return ((b >> base) & 0xFull) == 0xFull;
}

// the four streams of a block: offsets and lengths from the 6-byte jump table (U/HufDecompress.cs:344-389); false = malformed
struct Streams4 { u32 so, sl, on, seg; };
__device__ __forceinline__ bool streams4(Streams4& S, const u8* hsrc, u32 hlen, u32 litSize, u32 which)
{
    if (hlen < 10) return false;
    const u32 l1 = readLE16(hsrc), l2 = readLE16(hsrc + 2), l3 = readLE16(hsrc + 4);
    S.seg = (litSize + 3) / 4;
    if (6 + l1 + l2 + l3 > hlen) return false;
    if (S.seg * 3 > litSize) return false;
    const u32 l4 = hlen - 6 - l1 - l2 - l3;
    S.so = which == 0 ? 6 : which == 1 ? 6 + l1 : which == 2 ? 6 + l1 + l2 : 6 + l1 + l2 + l3;
    S.sl = which == 0 ? l1 : which == 1 ? l2 : which == 2 ? l3 : l4;
    S.on = which < 3 ? S.seg : litSize - 3 * S.seg;
    return true;
}

// slow path: one block per wave, handles everything (incl. 12-bit Huffman tables); only runs for blocks the quad kernels flagged
// (slowFlags null: every block)
__global__ __launch_bounds__(64) void decode_literals_slow_kernel(const u8* __restrict__ src, u8* __restrict__ out, u8* __restrict__ scratch,
                                                                  const FrameDesc* __restrict__ frames, const BlockDesc* __restrict__ blocks, u32 nBlocks,
                                                                  u32* __restrict__ status, const u8* __restrict__ slowFlags,
                                                                  const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ LitLds L;
    const u32 bi = blockIdx.x, lane = threadIdx.x;
    if (bi >= nBlocks || (slowFlags && !slowFlags[bi]) || status[kStErr]) return;
    LitJob J;
    if (!lit_job(J, bi, blocks, frames, src, out, scratch, dictFull, di)) return;
    u32 err = 0;
    do {
        u32 nbSymbols = 0, tableLog = 0, hs = 0;
        if (lane == 0) hs = huf_read_stats(L, J.tsrc, J.tlen, &nbSymbols, &tableLog);
        hs = uniform(hs); nbSymbols = uniform(nbSymbols); tableLog = uniform(tableLog);
        if (!hs || (J.own ? hs >= J.hlen : hs > J.tlen)) { err = J.tErr; break; }
        huf_build_table(L, nbSymbols, tableLog, lane);
        const u8* hsrc = J.hsrc; u32 hlen = J.hlen;
        if (J.own) { hsrc += hs; hlen -= hs; }
        bool ok = true;
        if (J.single) {
            if (lane == 0) ok = tableLog == 12 ? huf_decode_stream<true>(L, tableLog, hsrc, hlen, J.dst, J.litSize)
                                               : huf_decode_stream<false>(L, tableLog, hsrc, hlen, J.dst, J.litSize);
        } else {
            Streams4 S;
            if (!streams4(S, hsrc, hlen, J.litSize, lane & 3)) { err = kErrCorruption; break; }
            if (lane < 4) ok = tableLog == 12 ? huf_decode_stream<true>(L, tableLog, hsrc + S.so, S.sl, J.dst + lane * S.seg, S.on)
                                              : huf_decode_stream<false>(L, tableLog, hsrc + S.so, S.sl, J.dst + lane * S.seg, S.on);
        }
        if (ballot(!ok)) err = kErrCorruption;
    } while (false);
    if (err && lane == 0) report_error(status, bi, kStageLiterals, err);
}

// ------------------------------------------------------------------------------------------------
// One stream on one lane, the form both quad decoders use (HUF_decodeStreamX1, U/HufDecompress.cs:264-309).
//
// What bounds a lane is the dependent chain from one symbol's table entry to the next symbol's table index, so that chain is
// kept to three vector instructions around the LDS read:
//     t = w >> (31 - IDX)          the index bits (and the bit behind them) of the 32-bit window w
//     a = (t & mask) | tableBase   LDS address
//     e = lds[a]                   entry: low byte = 32 - nbBits, high byte = symbol
//     w = alignbit(w, r, e)        v_alignbit_b32 takes its shift from the entry's low 5 bits as they are: {w, r} >> (32 - nbBits)
// (r, q), the 64 bits behind the window, the output packing and the bit count follow off the chain.  After 8 symbols (at most
// 88 of the 96 bits) the window is rebuilt from the byte stream, see the function body.
// IDX = index bits of the table (tables of a smaller tableLog are replicated up to it).  PAIRS: tableLog = IDX + 1 is held in the
// same table: the longest codes come in pairs sharing an IDX-bit prefix and own the first entries of the table (canonical order:
// weight classes ascending); such an entry holds BOTH symbols (low byte: next bit 0, high byte: next bit 1), "index below the
// pair count" says so before the entry has arrived, and its length is the constant IDX + 1: one select on the chain, no second read.
// The table is addressed by its LDS offset (aligned to its size), so that index and base combine in one v_and_or_b32.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 huf_entry(u32 sym, u32 nbBits) { return (32u - nbBits) | (sym << 8); }
typedef __attribute__((address_space(3))) const u16 lds_u16_t;
#define ZMI_LDS_U16(off) (*(lds_u16_t*)(size_t)(off))
__device__ __forceinline__ u32 lds_offset(const void* p) { return (u32)(size_t)(__attribute__((address_space(3))) const u8*)p; }

// quadStore (uniform over the 4 lanes of a quad that decode the 4 streams of one block): the first nQuad symbols (a multiple of
// 64, the same for the 4 lanes) are written through a 4 x 4 transpose inside the quad (v_mov_dpp quad_perm, no LDS): lane j
// then stores bytes [16 j, 16 j + 16) of stream p's 64-byte piece, p = 0 .. 3, so that every store instruction writes 64
// contiguous bytes per stream instead of 16 — whole 64-byte sectors reach L2 / HBM (16-byte pieces were written back 2.4 times
// over).  seg = distance between the output bases of consecutive streams, ql = this lane's stream.
template <u32 IDX, bool PAIRS>
__device__ __forceinline__ bool huf_decode_stream_fs(const u32 tOff, const u32 nPair2,
                                                     const u8* __restrict__ src, u32 srcSize, u8* __restrict__ out, u32 n,
                                                     const bool quadStore = false, const u32 nQuad = 0, const u32 ql = 0, const u32 seg = 0)
{
    if (srcSize < 1) return false;
    s32 remaining; u32 i = 0;
    constexpr u32 kMask = ((1u << IDX) - 1u) << 1;
    // one symbol out of the 96-bit window (w, r, q): returns it, advances the window, adds 32 - its bits to `spare`
    // (nPair2 = twice the number of pair entries: t, which carries the bit behind the index, is below it exactly for them)
    auto sym1 = [&](u32& w, u32& r, u32& q, u32& spare) -> u32 {
        const u32 t = w >> (31 - IDX);
        const u32 e = ZMI_LDS_U16(tOff | (t & kMask));
        u32 e2 = e, sh = 8;
        if (PAIRS) { const bool isPair = t < nPair2; e2 = isPair ? 31u - IDX : e; sh = isPair ? (t & 1u) << 3 : 8u; }
        w = __builtin_amdgcn_alignbit(w, r, e2);
        r = __builtin_amdgcn_alignbit(r, q, e2);
        q = __builtin_amdgcn_alignbit(q, 0u, e2);
        spare += e2 & 0xFFu;
        return (e >> sh) & 0xFFu;
    };
    if (srcSize >= 16) {
        // The stream is read from its last byte down.  cont = stream bytes [ptr, ptr + 8); (lowHi, lowLo) = the 16 bytes below it,
        // [ptr - 8, ptr) and [ptr - 16, ptr - 8), loaded one step (8 symbols) ahead of their use; bytes below the stream's start
        // read as zero.  A step decodes 8 symbols (at most 88 bits) out of the 96 bits behind the `consumed` (<= 8) bits already
        // used of cont, then re-bases the registers by the whole bytes consumed (<= 12) — from registers, so that the next step
        // starts at once — and issues the load that replaces (lowHi, lowLo) by the exact 16 bytes below the new cont.  The load
        // has 8 symbols' time to arrive (an L2 hit takes about 4 symbols' time), and between two store groups (64 symbols) it
        // is not behind any store: vmcnt counts loads and stores together, in order, so a load issued after a store waits for it.
        s32 ptr = (s32)srcSize - 8;
        u64 cont = readLE64(src + ptr);
        u64 lowHi = 0, lowLo = 0;
        auto load_low = [&]() {
            const s32 lp = ptr - 16;
            const u8* a = src + (lp > 0 ? lp : 0);
            lowLo = readLE64(a); lowHi = readLE64(a + 8);
        };
        auto fix_low = [&]() {           // bytes below the stream's start are zero (only the last steps of a stream get here)
            const s32 lp = ptr - 16;
            if (lp < 0) {
                if (lp <= -16) { lowHi = 0; lowLo = 0; }
                else {
                    const u32 sh = 8u * (u32)(-lp);                 // 8 .. 120: the 16 loaded bytes are [0, 16), the wanted ones [lp, lp + 16)
                    if (sh >= 64) { lowHi = lowLo << (sh - 64); lowLo = 0; }
                    else { lowHi = (lowHi << sh) | (lowLo >> (64 - sh)); lowLo <<= sh; }
                }
            }
        };
        load_low();
        const u32 last = (u32)(cont >> 56);
        if (!last) return false;
        remaining = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);
        u32 consumed = 64u - (u32)(remaining - 8 * ptr);               // 1 .. 8
        u32 hiTop;                                                     // the 4 bytes right below cont: all a window needs beyond cont
        {   // (computed on a copy: the registers themselves are fixed where they are used, in the first step's re-base)
            const u64 h0 = lowHi, l0 = lowLo;
            fix_low(); hiTop = (u32)(lowHi >> 32);
            lowHi = h0; lowLo = l0;
        }
        // one step = 8 symbols -> two dwords
        auto step = [&](u32& word0, u32& word1) {
            const u64 top = cont << consumed;                                          // the 96-bit window = (cont : hiTop) << consumed
            const u64 mid = ((((u64)(u32)cont) << 32) | hiTop) << consumed;
            u32 w = (u32)(top >> 32), r = (u32)(mid >> 32), q = (u32)mid, spare = 0;
            word0 = sym1(w, r, q, spare); word0 |= sym1(w, r, q, spare) << 8; word0 |= sym1(w, r, q, spare) << 16; word0 |= sym1(w, r, q, spare) << 24;
            word1 = sym1(w, r, q, spare); word1 |= sym1(w, r, q, spare) << 8; word1 |= sym1(w, r, q, spare) << 16; word1 |= sym1(w, r, q, spare) << 24;
            consumed += 8u * 32u - spare;
            // re-base by k whole bytes out of the 24 bytes in registers (now exact: the load issued a step ago has arrived)
            fix_low();
            const u32 k = consumed >> 3, j8 = 8u * (k & 7u);
            const bool far = k >= 8;
            const u64 X = far ? lowHi : cont, Y = far ? lowLo : lowHi, Z = far ? 0ull : lowLo;
            cont = (X << j8) | (j8 ? Y >> (64 - j8) : 0ull);
            const u64 hn = (Y << j8) | (j8 ? Z >> (64 - j8) : 0ull);     // (its top 4 bytes are exact for every k <= 12)
            hiTop = (u32)(hn >> 32);
            ptr -= (s32)k; consumed &= 7u;
            load_low();
        };
        if (quadStore) {
            const bool odd1 = ql & 1u, odd2 = ql & 2u;
            u8* const tbase = out - (size_t)ql * seg + 16u * ql;       // stream 0's base + this lane's 16-byte column
            while (i + 64 <= nQuad) {
                u32 d[16];
#pragma unroll
                for (u32 g = 0; g < 8; ++g) step(d[2 * g], d[2 * g + 1]);
                // 4 x 4 transpose of 16-byte pieces over the quad's lanes: piece p of lane l <-> piece l of lane p
                u32 b[16], c[16];
#pragma unroll
                for (u32 p = 0; p < 4; ++p)
#pragma unroll
                    for (u32 k = 0; k < 4; ++k) {
                        const u32 nb1 = (u32)__builtin_amdgcn_mov_dpp((int)d[4 * (p ^ 1u) + k], 0xB1, 0xF, 0xF, true);    // lane ^ 1's piece p ^ 1
                        b[4 * p + k] = (odd1 == (bool)(p & 1u)) ? d[4 * p + k] : nb1;
                    }
#pragma unroll
                for (u32 p = 0; p < 4; ++p)
#pragma unroll
                    for (u32 k = 0; k < 4; ++k) {
                        const u32 nb2 = (u32)__builtin_amdgcn_mov_dpp((int)b[4 * (p ^ 2u) + k], 0x4E, 0xF, 0xF, true);    // lane ^ 2's piece p ^ 2
                        c[4 * p + k] = (odd2 == (bool)(p & 2u)) ? b[4 * p + k] : nb2;
                    }
#pragma unroll
                for (u32 p = 0; p < 4; ++p) {
                    u32u* o = (u32u*)(tbase + (size_t)p * seg + i);
                    o[0] = c[4 * p]; o[1] = c[4 * p + 1]; o[2] = c[4 * p + 2]; o[3] = c[4 * p + 3];
                }
                i += 64;
            }
        }
        while (i + 64 <= n) {
            u32 d[16];
#pragma unroll
            for (u32 g = 0; g < 8; ++g) step(d[2 * g], d[2 * g + 1]);
            u32u* o = (u32u*)(out + i);
#pragma unroll
            for (u32 g = 0; g < 16; ++g) o[g] = d[g];
            i += 64;
        }
        while (i + 8 <= n) {
            u32 d0, d1;
            step(d0, d1);
            u32u* o = (u32u*)(out + i); o[0] = d0; o[1] = d1;
            i += 8;
        }
        remaining = 8 * ptr + 64 - (s32)consumed;
    } else {
        const u32 last = src[srcSize - 1];
        if (!last) return false;
        remaining = (s32)(srcSize - 1) * 8 + (s32)highbit32(last);
    }
    if (i < n && remaining > 0) {        // short stream, or the last symbols of a long one: plain bit reader
        BackBits bd; bd.base = src; bd.size = (s32)srcSize; bd.pos = remaining; bd.load_window(remaining);
        while (i < n) {
            u32 w = bd.peek(IDX + 1) << (31 - IDX), r = 0, q = 0, spare = 0;
            out[i++] = (u8)sym1(w, r, q, spare);
            bd.pos -= (s32)(32u - spare);
        }
        remaining = bd.pos;
    }
    return i == n && remaining == 0;
}

// ------------------------------------------------------------------------------------------------
// literals, fast path: kQuads frames per wave, 4 lanes (one per Huffman stream) per frame
// ------------------------------------------------------------------------------------------------
// The four streams of a block are four serial table-lookup chains, so a frame can keep only four lanes busy, and a
// frame needs its 4 KiB X1 table in LDS.  What bounds the kernel is therefore LDS capacity (32 frames = 128 busy
// lanes per CU) and the length of one lookup step; packing 8 frames into a wave lets one wave instruction advance
// 32 streams instead of 4, which is what the one-frame-per-wave form wasted its issue slots on.
constexpr u32 kQuads = 8;
// (the X1 tables live in their own LDS array, aligned to their size: huf_decode_stream_fs; before a table is filled its storage
//  holds the FSE scratch of huf_read_stats)
struct QuadLds {
    u8  weights[256];
    u16 start[256];             // first table index of each symbol
    u32 meta[4];                // hs, nbSymbols, tableLog, valid
};
struct QuadScratch {            // view used by huf_read_stats: FSE scratch aliased onto the (not yet built) table
    u8* weights; s16* norm; u16* symbolNext; u16* wNewState; u8* wSymbol; u8* wNbBits;
};

// rank starts -> per-symbol first index (HUF_readDTableX1), on the quad leader, after huf_read_stats
__device__ __forceinline__ void quad_symbol_starts(QuadLds& Q, u32 nbSymbols, u32 tl)
{
    u32 cnt[13]; for (int i = 0; i < 13; i++) cnt[i] = 0;
    for (u32 n = 0; n < nbSymbols; n++) cnt[Q.weights[n]]++;
    u32 rs[13]; u32 next = 0;
    for (u32 w = 1; w <= tl; w++) { rs[w] = next; next += cnt[w] << (w - 1); }
    for (u32 n = 0; n < nbSymbols; n++) { const u32 w = Q.weights[n]; if (w) { Q.start[n] = (u16)rs[w]; rs[w] += (1u << w) >> 1; } }
}

// One block on the 4 lanes of a quad.  The job is computed redundantly by the 4 lanes (same loads, same values, so the quad's
// control flow is uniform without any cross-lane traffic); only the weight decoding runs on the quad leader.
// Returns an error code, or 0xFFFF to ask for the slow path (12-bit table).
__device__ u32 quad_decode_literals(u16* __restrict__ huf, QuadLds& Q, const LitJob& J, const u32 ql)
{
    if (ql == 0) {
        QuadScratch sc;
        sc.weights = Q.weights; sc.norm = reinterpret_cast<s16*>(huf); sc.symbolNext = huf + 256;
        sc.wNewState = huf + 512; sc.wSymbol = reinterpret_cast<u8*>(huf + 576); sc.wNbBits = reinterpret_cast<u8*>(huf + 608);
        u32 nbSymbols = 0, tl = 0;
        const u32 hs = huf_read_stats(sc, J.tsrc, J.tlen, &nbSymbols, &tl);
        if (hs && tl <= 11) quad_symbol_starts(Q, nbSymbols, tl);
        Q.meta[0] = hs; Q.meta[1] = nbSymbols; Q.meta[2] = tl;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    const u32 hs = Q.meta[0], nbSymbols = Q.meta[1], tableLog = Q.meta[2];
    if (!hs || (J.own ? hs >= J.hlen : hs > J.tlen)) return J.tErr;
    if (tableLog > 11) return 0xFFFFu;
    const u32 up = 11 - tableLog;                      // the table is indexed by 11 bits: a smaller one is replicated 2^up times
    for (u32 n = ql; n < nbSymbols; n += 4) {          // table fill, 4 lanes
        const u32 w = Q.weights[n];
        if (!w) continue;
        const u32 len = ((1u << w) >> 1) << up, st = (u32)Q.start[n] << up;
        const u32 e = huf_entry(n, tableLog + 1 - w);
        if (len >= 4) { const u64 e4 = (u64)(e | (e << 16)) * 0x100000001ull; for (u32 u = 0; u < len; u += 4) *reinterpret_cast<u64*>(&huf[st + u]) = e4; }
        else for (u32 u = 0; u < len; u++) huf[st + u] = (u16)e;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    const u8* hsrc = J.hsrc; u32 hlen = J.hlen;
    if (J.own) { hsrc += hs; hlen -= hs; }
    bool ok = true;
    {   // one call site: a single-stream block is "stream 0 of 1" on the quad leader
        Streams4 S; S.so = 0; S.sl = hlen; S.on = J.litSize; S.seg = 0;
        if (!J.single && !streams4(S, hsrc, hlen, J.litSize, ql)) return kErrCorruption;
        // (the four streams hold seg, seg, seg and <= seg symbols, and every one of them at least 16 bytes of input when the
        //  transposed stores are used: the group loop then runs the same number of times on the 4 lanes)
        const u32 lastOn = J.litSize - 3 * S.seg;
        const bool quadStore = !J.single && ZMI_LIT_QUADSTORE && quad_all(S.sl >= 16, ql);
        if (!J.single || ql == 0) ok = huf_decode_stream_fs<11, false>(lds_offset(huf), 0u, hsrc + S.so, S.sl, J.dst + ql * S.seg, S.on, quadStore, lastOn & ~63u, ql, S.seg);
    }
    // any stream of the quad failing fails the block: combine through LDS (the 4 lanes are converged here)
    if (ql == 0) Q.meta[3] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (!ok) Q.meta[3] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    return Q.meta[3] ? (u32)kErrCorruption : 0u;
}

// =====================================================================================================================
// Literal decoder, serial form, COMPACT tables: what bounds the serial form is LDS (a 4 KiB table per block admits 32 blocks
// per CU, so 16 384 blocks take two rounds).  Here the table is indexed by at most 10 bits (2 KiB); for tableLog 11 the longest
// codes (weight 1) come in pairs that share a 10-bit prefix, and a pair's entry holds both symbols (huf_decode_stream_fs).
// 2.4 KiB per block -> 64 blocks per CU -> one round, as FOUR workgroups of 16 blocks: a wave works with all its 64 lanes (16
// quads), i.e. half the instructions of two half-filled waves, and has its SIMD's issue slots to itself.
// Same acceptance as the other forms; tableLog 12 goes the slow way.
// =====================================================================================================================
constexpr u32 kQuadsC = 16;
struct CompactLds {             // (the tables live in their own LDS array, aligned to their size; before a table is filled its
    u8  sorted[256];            //  storage holds the FSE scratch, low 1280 B, and the weights, top 256 B)
                                // sorted: symbols ordered by (weight, symbol), weight 0 excluded; its head = the 11-bit codes in table order
    u16 classStart[14];         // first index (in the tableLog-bit table) of weight class w; [tableLog + 1] = table size
    u16 classFirst[14];         // index into sorted[] of the first symbol of class w
    u32 meta[4];
};

// One block on the 4 lanes of a quad (compact tables).  Returns an error code, or 0xFFFF to ask for the slow path.
__device__ u32 quad_decode_literals_c(u16* __restrict__ huf, CompactLds& Q, const LitJob& J, const u32 ql)
{
    u8* const weights = reinterpret_cast<u8*>(huf + 896);       // top 256 B of the table area until the fill
    if (ql == 0) {
        QuadScratch sc;
        sc.weights = weights; sc.norm = reinterpret_cast<s16*>(huf); sc.symbolNext = huf + 256;
        sc.wNewState = huf + 512; sc.wSymbol = reinterpret_cast<u8*>(huf + 576); sc.wNbBits = reinterpret_cast<u8*>(huf + 608);
        u32 nbSymbols = 0, tl = 0;
        const u32 hs = huf_read_stats(sc, J.tsrc, J.tlen, &nbSymbols, &tl);
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
    const u32 hs = Q.meta[0], tableLog = Q.meta[2];
    if (!hs || (J.own ? hs >= J.hlen : hs > J.tlen)) return J.tErr;
    if (tableLog > 11) return 0xFFFFu;
    {   // table fill by the quad's 4 lanes, one symbol of `sorted` at a time
        const u32 nSorted = Q.classFirst[tableLog + 1];
        const u32 drop = tableLog > 10 ? 1u : 0u;          // 11-bit codes: the table is indexed by the upper 10 bits
        const u32 up = tableLog < 10 ? 10 - tableLog : 0u; // a smaller table is replicated up to 10 index bits
        for (u32 k = ql; k < nSorted; k += 4) {
            u32 w = 1;
            for (u32 cw = 2; cw <= tableLog; ++cw) if (Q.classFirst[cw] <= k) w = cw;
            const u32 start = Q.classStart[w] + ((k - Q.classFirst[w]) << (w - 1));     // index in the tableLog-bit table
            const u32 sym = Q.sorted[k];
            if (drop && w == 1) {                   // 11-bit codes: both symbols of the pair in one entry (see huf_decode_stream_fs)
                if (!(start & 1)) huf[start >> 1] = (u16)(sym | ((u32)Q.sorted[k + 1] << 8));
                continue;
            }
            const u32 len = (((1u << w) >> 1) >> drop) << up, st = (start >> drop) << up;
            const u32 e = huf_entry(sym, tableLog + 1 - w);
            if (len >= 4) { const u64 e4 = (u64)(e | (e << 16)) * 0x100000001ull; for (u32 u = 0; u < len; u += 4) *reinterpret_cast<u64*>(&huf[st + u]) = e4; }
            else for (u32 u = 0; u < len; u++) huf[st + u] = (u16)e;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    const bool pairs = tableLog > 10;
    const u32 nPair2 = pairs ? (u32)Q.classFirst[2] : 0u;      // the 11-bit codes = weight class 1: n1 of them, n1 / 2 pair entries, t < n1
    const u8* hsrc = J.hsrc; u32 hlen = J.hlen;
    if (J.own) { hsrc += hs; hlen -= hs; }
    bool ok = true;
    {   // one call site per table form: a single-stream block is "stream 0 of 1" on the quad leader
        Streams4 S; S.so = 0; S.sl = hlen; S.on = J.litSize; S.seg = 0;
        if (!J.single && !streams4(S, hsrc, hlen, J.litSize, ql)) return kErrCorruption;
        const u32 lastOn = J.litSize - 3 * S.seg;
        const bool quadStore = !J.single && ZMI_LIT_QUADSTORE && quad_all(S.sl >= 16, ql);
        if (!J.single || ql == 0) {
            if (pairs) ok = huf_decode_stream_fs<10, true>(lds_offset(huf), nPair2, hsrc + S.so, S.sl, J.dst + ql * S.seg, S.on, quadStore, lastOn & ~63u, ql, S.seg);
            else       ok = huf_decode_stream_fs<10, false>(lds_offset(huf), 0u, hsrc + S.so, S.sl, J.dst + ql * S.seg, S.on, quadStore, lastOn & ~63u, ql, S.seg);
        }
    }
    if (ql == 0) Q.meta[3] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    if (!ok) Q.meta[3] = 1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
    return Q.meta[3] ? (u32)kErrCorruption : 0u;
}

__global__ __launch_bounds__(64) void decode_literals_compact_kernel(const u8* __restrict__ src, u8* __restrict__ out, u8* __restrict__ scratch,
                                                                     const FrameDesc* __restrict__ frames, const BlockDesc* __restrict__ blocks, u32 nBlocks,
                                                                     u32* __restrict__ status, u8* __restrict__ slowFlags,
                                                                     const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ __attribute__((aligned(2048))) u16 tables[kQuadsC][1024];
    __shared__ CompactLds Qs[kQuadsC];
    const u32 lane = threadIdx.x, q = lane >> 2, ql = lane & 3;
    const u32 bi = blockIdx.x * kQuadsC + q;
    if (bi >= nBlocks) return;
    if (ql == 0) slowFlags[bi] = 0;
    if (status[kStErr]) return;
    LitJob J;
    if (!lit_job(J, bi, blocks, frames, src, out, scratch, dictFull, di)) return;
    const u32 err = quad_decode_literals_c(tables[q], Qs[q], J, ql);
    if (ql == 0) {
        if (err == 0xFFFFu) slowFlags[bi] = 1;
        else if (err) report_error(status, bi, kStageLiterals, err);
    }
}

__global__ __launch_bounds__(64) void decode_literals_kernel(const u8* __restrict__ src, u8* __restrict__ out, u8* __restrict__ scratch,
                                                             const FrameDesc* __restrict__ frames, const BlockDesc* __restrict__ blocks, u32 nBlocks,
                                                             u32* __restrict__ status, u8* __restrict__ slowFlags,
                                                             const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    __shared__ __attribute__((aligned(4096))) u16 tables[kQuads][2048];
    __shared__ QuadLds Qs[kQuads];
    const u32 lane = threadIdx.x, q = lane >> 2, ql = lane & 3;
    const u32 bi = blockIdx.x * kQuads + q;
    if (q >= kQuads || bi >= nBlocks) return;          // (a wave holds kQuads quads: its upper lanes have no table to work with)
    if (ql == 0) slowFlags[bi] = 0;
    if (status[kStErr]) return;
    LitJob J;
    if (!lit_job(J, bi, blocks, frames, src, out, scratch, dictFull, di)) return;
    const u32 err = quad_decode_literals(tables[q], Qs[q], J, ql);
    if (ql == 0) {
        if (err == 0xFFFFu) slowFlags[bi] = 1;
        else if (err) report_error(status, bi, kStageLiterals, err);
    }
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
    if (lane == 0) { atomicAdd(&g_litStamps[8], (unsigned long long)nPass); atomicAdd(&g_litStamps[9], 1ull); atomicAdd(&g_litStamps[10], t1 - t0); atomicAdd(&g_litStamps[11], __builtin_amdgcn_s_memtime() - t1); }
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

// one block on a 256-thread workgroup; every branch is workgroup-uniform
__device__ __forceinline__ u32 sync_decode_literals(SyncLds& L, const LitJob& J, const u32 tid)
{
    const u32 lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        QuadScratch sc;
        sc.weights = L.weights; sc.norm = reinterpret_cast<s16*>(L.huf); sc.symbolNext = L.huf + 256;
        sc.wNewState = L.huf + 512; sc.wSymbol = reinterpret_cast<u8*>(L.huf + 576); sc.wNbBits = reinterpret_cast<u8*>(L.huf + 608);
        u32 nbSymbols = 0, tl = 0;
        const u32 hs = huf_read_stats(sc, J.tsrc, J.tlen, &nbSymbols, &tl);
        L.meta[0] = hs; L.meta[1] = nbSymbols; L.meta[2] = tl; L.err = 0;
    }
    __syncthreads();
    const u32 hs = L.meta[0], nbSymbols = L.meta[1], tableLog = L.meta[2];
    if (!hs || (J.own ? hs >= J.hlen : hs > J.tlen)) return J.tErr;
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
    const u8* hsrc = J.hsrc; u32 hlen = J.hlen;
    if (J.own) { hsrc += hs; hlen -= hs; }
    bool ok = true;
    if (J.single) {
        if (wave == 0) ok = huf_decode_stream_sync(L.huf, tableLog, hsrc, hlen, J.dst, J.litSize, lane, L.stage[0]);
    } else {
        Streams4 S;
        if (!streams4(S, hsrc, hlen, J.litSize, wave)) return kErrCorruption;
        ok = huf_decode_stream_sync(L.huf, tableLog, hsrc + S.so, S.sl, J.dst + wave * S.seg, S.on, lane, L.stage[wave]);
    }
    if (!ok && lane == 0) L.err = 1;
    __syncthreads();
    return L.err ? (u32)kErrCorruption : 0u;
}

__global__ __launch_bounds__(256) void decode_literals_sync_kernel(const u8* __restrict__ src, u8* __restrict__ out, u8* __restrict__ scratch,
                                                                   const FrameDesc* __restrict__ frames, const BlockDesc* __restrict__ blocks, u32 nBlocks,
                                                                   u32* __restrict__ status, const u8* __restrict__ dictFull, const DictInfo* __restrict__ di)
{
    extern __shared__ __attribute__((aligned(16))) u8 syncLdsRaw[];
    SyncLds& L = *reinterpret_cast<SyncLds*>(syncLdsRaw);
    const u32 bi = blockIdx.x, tid = threadIdx.x;
    if (bi >= nBlocks || status[kStErr]) return;
    LitJob J;
    if (!lit_job(J, bi, blocks, frames, src, out, scratch, dictFull, di)) return;
    const u32 err = sync_decode_literals(L, J, tid);
    if (err && tid == 0) report_error(status, bi, kStageLiterals, err);
}

// Three literal decoders (tools/lit_decoder_crossover.py, MI355X, 64 KiB Zipf frames):
//   serial   (4 lanes per block, 4 KiB table):  32 blocks per CU; a round of 8192 blocks takes 1.1-1.7 ms;
//   compact  (4 lanes per block, 2 KiB table + pair table): 64 blocks per CU; a round of 16 384 blocks takes 1.8-2.7 ms;
//   selfsync (256 lanes per block): 0.15 ms up to 256 blocks, 0.25 ms per 1000 blocks beyond.
// mode: 0 = choose by block count, 1 = serial, 2 = self-synchronising, 3 = compact.
void launch_decode_literals(const u8* src, u8* out, u8* scratch, const FrameDesc* frames, const BlockDesc* blocks, u32 nBlocks, u32* status,
                            u8* slowFlags, u32 mode, const u8* dictFull, const DictInfo* di, hipStream_t stream, StageHook hook)
{
    if (mode == 0) {
        static int cus[64] = {};                     // per device
        int dev = 0; (void)hipGetDevice(&dev);
        if (!cus[dev & 63]) { int n = 0; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); cus[dev & 63] = n > 0 ? n : 256; }
        const u32 roundS = (u32)cus[dev & 63] * 32u, roundC = roundS * 2u;
        if (nBlocks <= roundS * 3u / 4u) mode = 2;
        else {
            const float tS = (float)((nBlocks + roundS - 1) / roundS) * 1.7f, tC = (float)((nBlocks + roundC - 1) / roundC) * 2.7f;
            mode = tC < tS ? 3u : 1u;
        }
    }
    if (mode == 2) {
        static bool attrSet[64] = {};               // per device
        int dev = 0; (void)hipGetDevice(&dev);
        if (!attrSet[dev & 63]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_literals_sync_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(SyncLds)); attrSet[dev & 63] = true; }
        hipLaunchKernelGGL(decode_literals_sync_kernel, dim3(nBlocks), dim3(256), sizeof(SyncLds), stream, src, out, scratch, frames, blocks, nBlocks, status, dictFull, di);
        hook("decode_literals");
        return;
    }
    if (mode == 3) hipLaunchKernelGGL(decode_literals_compact_kernel, dim3((nBlocks + kQuadsC - 1) / kQuadsC), dim3(64), 0, stream, src, out, scratch, frames, blocks, nBlocks, status, slowFlags, dictFull, di);
    else           hipLaunchKernelGGL(decode_literals_kernel, dim3((nBlocks + kQuads - 1) / kQuads), dim3(64), 0, stream, src, out, scratch, frames, blocks, nBlocks, status, slowFlags, dictFull, di);
    hook("decode_literals");
    hipLaunchKernelGGL(decode_literals_slow_kernel, dim3(nBlocks), dim3(64), 0, stream, src, out, scratch, frames, blocks, nBlocks, status, (const u8*)slowFlags, dictFull, di);
    hook("decode_literals_slow");
}

} // namespace zmi
