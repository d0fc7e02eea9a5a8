// lz_fast.hip — the LZ match finders + parse for gfx950: one 1024-thread workgroup per 64 KiB chunk, chunk resident in LDS.
//
// Takes the place of ZSTD_compressBlock_fast_noDict_generic (U/ZstdFast.cs:96-288), of doubleFast
// (U/ZstdDoubleFast.cs:51-247) and of the greedy/lazy search (U/ZstdLazy.cs:1743-2032) + ZSTD_storeSeq
// (U/ZstdCompressInternal.cs:204-246) for one block.  It is NOT those functions' parse: the reference walks the block
// serially, inserting only the positions it visits.  Here, per 4096-position tile:
//
//   probe    every position is hashed by its own lane (24-bit multiplies over the first 6 bytes; the dual finder adds an
//            8-byte and a 5-byte hash) and looks up two LDS tables — the latest occurrence in EARLIER tiles and the first
//            occurrence in THIS tile (atomicMin with a tile stamp, so it needs no clearing); entries carry a 16-bit tag;
//   verify   candidates whose tag matches are compared and extended up to 32 bytes (the reference's 4-byte check,
//            ZstdFast.cs:179-191); periods 1..4 (runs) are tested in registers; the lazy finder lets a match yield to the
//            one starting a byte later (gain rule of U/ZstdLazy.cs:1836-1870);
//   select   the greedy left-to-right parse "next = first match at or after the end of the current one" is the orbit
//            of a jump function over the tile: tiles with <= 64 matches are walked by wave 0; in dense tiles every wave
//            turns its 256 positions into an entry -> exit function (pointer doubling in registers, by shuffles), the
//            sixteen functions are chained after one barrier and every wave marks its own part of the orbit;
//   finish   matches that hit the 32-byte cap are completed by wave 0 with 64 lanes x 8 bytes per step, in order,
//            dropping the selections they swallow;
//   emit     every selected match computes its own sequence in parallel (rank by popcount prefix, literal length from
//            the previous selected match's end, backward extension as ZstdFast.cs:242-247);
//   literals bytes not covered by a selected match are compacted 4 per lane; every wave scans the 64 coverage words
//            itself, so the compaction needs no barrier.
//
// Offsets are stored raw (distance + 3); the repcode assignment (U/ZstdDecompressBlock.cs:2387-2443 run forward) is done
// by seq_encode.  All table updates are commutative atomics or single-writer stores and every parallel step is
// order-independent, so the output is deterministic.
//
// LDS (one workgroup per CU, 158 KiB): chunk 64 KiB (+pad) | 64 KiB of hash tables | tile arrays.
// HBM traffic per chunk: read n, write literals (<= n) + 8 B per sequence.
#include <type_traits>
#include "zmi_device.h"

namespace zmi {

#ifdef ZMI_LZ_STAMPS
// diagnostic build only (make STAMPS=1): per-phase shader-clock sums of thread 0 of every workgroup
__device__ unsigned long long g_lzStamps[24];     // 0-9 tile loop, 10-15 region parse, 16-23 hash-chain search (level >= 5)
#define ZMI_STAMP(i) do { if ((tid & 63u) == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stampAcc[i] += now_ - stampLast; stampLast = now_; } } while (0)
// (the region parse is a function of its own: it times itself from its entry and adds to the global sums directly)
#define ZMI_DSTAMP(i) do { if ((tid & 63u) == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); dAcc[(i) - 10] += now_ - dLast; dLast = now_; } } while (0)
#else
#define ZMI_STAMP(i) do { } while (0)
#define ZMI_DSTAMP(i) do { } while (0)
#endif

constexpr u32 kHashLog  = 13;            // the reference's hashLog for level 1 at <= 128 KiB (U/Clevels.cs:488)
constexpr u32 kTile     = 1024;          // threads per workgroup
constexpr u32 kPPT      = 4;             // positions per thread per tile
constexpr u32 kTilePos  = kTile * kPPT;  // 4096 positions per tile
constexpr u32 kTileLog  = 12;
constexpr u32 kGroups   = kTilePos / 64; // 64-position groups per tile
constexpr u32 kLenCap   = 32;            // per-lane forward extension cap; capped matches are finished by wave 0
constexpr u32 kInPad    = 64;
constexpr u32 kDenseMin = 384;           // matches in a chunk's first tile from which the rest of the chunk goes to the region parse
constexpr u32 kPassPos  = 30720;         // positions per pass of the region parse: their candidates (u16) fill the 64 KiB of the tables
constexpr u32 kRegions  = kPassPos / 64; // 64-position regions per pass, one lane each
// FAR mode (row f-1 for the fast strategy): the table also holds positions of the input IN FRONT of the block, up to kFarMax
// bytes back inside the block's frame; table entries are (rel + 1) << 14 | tag14 with rel = kFarMax + block position
constexpr u32 kFarMax   = (192u << 10) - 4096;
constexpr u32 kFarStep  = 2;             // every 2nd position of the far history is inserted (the reference's own table is sparser)
static_assert((1u << kTileLog) == kTilePos && kGroups == 64, "tile geometry");

struct LzLds {
    u8  in[kChunkSize + kInPad];
    // 64 KiB of hash tables, laid out per finder (see lz_kernel):
    //   fast  : table u32[8192] (position+1 of the latest occurrence in EARLIER tiles; 0 = empty)
    //           first u32[8192] (first occurrence inside the CURRENT tile: ((15-tile) << 12) | index)
    //   dual  : firstL u32[4096] | firstS u32[4096] (as `first`, indexed by the hash's upper 12 bits)
    //           tableL u16[8192] | tableS u16[8192] (position+1 of the hash's first occurrence in the latest earlier tile that had it)
    u32 tabMem[2u << kHashLog];
    u8  tileLen[kTilePos];               // match length at each position of the tile (0 = none, kLenCap = "at least"); FAR: offset bits 16-17 on top
    u16 tileOff[kTilePos];
    u16 jump[kTilePos];                  // pointer-doubling array; afterwards: full length of capped selected matches
    u64 matchMask[kGroups];              // bit = position holds a match
    u64 capMask[kGroups];                // bit = that match hit the cap
    u64 selMask[kGroups];                // bit = match is on the greedy orbit (selected)
    u64 covMask[2][kGroups];             // bit = byte covered by a selected match (tile-local); slot = tile parity: a sparse tile has
                                         // a single barrier, so the next tile clears its masks while this tile's literals are still read
    u32 wordRank[kGroups + 1];           // selected matches before each group
    u16 jumpB[kTilePos];                 // second buffer of the doubling rounds; once they end it is reused as
                                         // endOf[r+1] = absolute end of the r-th selected match of the tile (u32[kTilePos/4+2]), endOf[0] = anchor
    // per-tile counters, slot = tile mod 3: between two consecutive verify barriers one slot is read (tile t), one is
    // accumulated (tile t+1, whose fused probe/verify may already run) and the third is reset for tile t+2
    u64 nzWords[3];                      // bit g = matchMask[g] != 0 (accumulated with atomicOr during verify)
    u32 matchCount[3];                   // matches in the current tile (decides sparse / dense selection)
    u16 sparseList[64];                  // sparse path: the tile's matches in position order
    u32 claim;                           // the chunk this workgroup takes next (lz_kernel with a claim counter)
};
static_assert(sizeof(LzLds) <= 160u * 1024u, "a workgroup's LDS");

// hash of the 6 bytes at a position (the reference's ZSTD_hash6 needs a 64x64-bit multiply, four quarter-rate VALU
// multiplies per lane; two 32-bit multiplies over the same six bytes mix as well for a 13-bit table)
// (v_mul_u32_u24 is a full-rate instruction, the 32-bit multiply is not: three input bytes per multiply)
__device__ __forceinline__ u32 hash6p(u64 w)
{
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    return __umul24(lo & 0xFFFFFFu, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbyte(hi, lo, 3) & 0xFFFFFFu, 0x85EBCBu);
}
// a hash product gives the table index (top kHashLog bits) and a 16-bit tag (the bits below): entries carry the tag in
// their low half, so a candidate whose tag differs is dropped without touching the input bytes
__device__ __forceinline__ u32 hidx(u32 prod) { return prod >> (32 - kHashLog); }
__device__ __forceinline__ u32 htag(u32 prod) { return (prod >> (16 - kHashLog)) & 0xFFFFu; }
// long hash of the dual finder: all 8 bytes (the reference's long table is hash8, U/ZstdDoubleFast.cs:60-75)
__device__ __forceinline__ u32 hash8p(u64 w)
{
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    return __umul24(lo & 0xFFFFFFu, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbyte(hi, lo, 3) & 0xFFFFFFu, 0x85EBCBu) + __umul24(hi >> 16, 0xC2B2AFu);
}
// short hash of the dual finder: 5 or 4 bytes (minMatch of U/Clevels.cs:490-492 at <= 128 KiB)
template <int BYTES> __device__ __forceinline__ u32 hash_shortp(u64 w)
{
    const u32 lo = (u32)w, hi = (u32)(w >> 32);
    if (BYTES == 5) return __umul24(lo & 0xFFFFFFu, 0x9E3779u) + __umul24(__builtin_amdgcn_alignbyte(hi, lo, 3) & 0xFFFFu, 0x85EBCBu);
    return lo * 2654435761u;
}
__device__ __forceinline__ u64 read_lane64(u64 v, u32 l) { return (u64)read_lane((u32)v, l) | ((u64)read_lane((u32)(v >> 32), l) << 32); }

// 8 / 4 input bytes at an arbitrary LDS position, fetched as ALIGNED dwords + v_alignbyte: consecutive lanes read
// consecutive positions, so four lanes share each dword (broadcast, conflict-free), which an unaligned ds_read_b64
// per lane is not
__device__ __forceinline__ u64 lds_load8(const u8* base, u32 pos)
{
    const u32* w32 = reinterpret_cast<const u32*>(base) + (pos >> 2);
    const u32 sh = pos & 3, d0 = w32[0], d1 = w32[1], d2 = w32[2];
    return (u64)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
}
__device__ __forceinline__ u32 lds_load4(const u8* base, u32 pos)
{
    const u32* w32 = reinterpret_cast<const u32*>(base) + (pos >> 2);
    return __builtin_amdgcn_alignbyte(w32[1], w32[0], pos & 3);
}

// 16 input bytes at an arbitrary LDS position: five aligned dwords, four v_alignbyte
__device__ __forceinline__ void lds_load16(const u8* base, u32 pos, u64& lo, u64& hi)
{
    const u32* w32 = reinterpret_cast<const u32*>(base) + (pos >> 2);
    const u32 sh = pos & 3, d0 = w32[0], d1 = w32[1], d2 = w32[2], d3 = w32[3], d4 = w32[4];
    lo = (u64)__builtin_amdgcn_alignbyte(d1, d0, sh) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, sh) << 32);
    hi = (u64)__builtin_amdgcn_alignbyte(d3, d2, sh) | ((u64)__builtin_amdgcn_alignbyte(d4, d3, sh) << 32);
}

// length of the match between position p (its first 16 bytes are w, w2) and cpos < p, capped; 0 if shorter than 4 (the reference's
// MEM_read32 check, ZstdFast.cs:179-191).  The first 16 bytes are compared without a branch; only longer matches loop.
__device__ __forceinline__ u32 match_len(const LzLds& L, u32 p, u32 cpos, u64 w, u64 w2, u32 n)
{
    u64 c1, c2;
    lds_load16(L.in, cpos, c1, c2);
    const u64 x1 = w ^ c1, x2 = w2 ^ c2;
    u32 l = x1 ? (ctz64(x1) >> 3) : 8u + (x2 ? (ctz64(x2) >> 3) : 8u);
    if (l == 16) {
        while (l < kLenCap) {
            const u64 x = lds_load8(L.in, p + l) ^ lds_load8(L.in, cpos + l);
            if (x) { l += ctz64(x) >> 3; break; }
            l += 8;
        }
    }
    if (l > n - p) l = n - p;
    if (l > kLenCap) l = kLenCap;
    return l >= 4 ? l : 0;
}

// the same against a candidate in front of the block: its bytes come from global memory (g = the candidate's address; it lies
// at least one byte before position p of the source, so every read below stays inside the source while p + l + 8 <= n)
__device__ __forceinline__ u32 match_len_far(const LzLds& L, u32 p, const u8* __restrict__ g, u64 w, u32 n)
{
    const u64 cw = readLE64(g);
    if ((u32)cw != (u32)w) return 0;
    u64 x = w ^ cw;
    u32 l = x ? (ctz64(x) >> 3) : 8;
    if (!x) {
        while (l < kLenCap && p + l + 8 <= n) {
            x = lds_load8(L.in, p + l) ^ readLE64(g + l);
            if (x) { l += ctz64(x) >> 3; break; }
            l += 8;
        }
    }
    if (l > n - p) l = n - p;
    if (l > kLenCap) l = kLenCap;
    return l >= 4 ? l : 0;
}


// ---------------------------------------------------------------------------------------------------------------------
// Region parse (fast strategy, dense data).  The tile loop above computes a verified match at EVERY position and then selects
// one position in eight; on dense data that verification and the selection machinery are most of its time.  Here the rest of
// a chunk (tiles [fromTile, nTiles)) is done in two steps:
//   I   per tile, as above: every position is hashed and inserted, but only the POSITION of its best-looking candidate (a
//       period, else the same-tile first occurrence, else the latest earlier one — tags checked, bytes not) is kept: 2 bytes
//       per position, through global memory (L2), because the tables still fill LDS;
//   II  passes of 30 720 positions whose candidates now take the tables' place in LDS: the positions are cut into regions of
//       64, ONE LANE PER REGION walks its region the way the reference's finder walks a block (U/ZstdFast.cs:130-260): look
//       at the candidate, verify, extend backward over pending literals, take the match, jump behind it; only the positions it
//       visits cost anything.  A match ends at its region's end; the next region continues it (same offset, through its
//       leading literals) in a second, parallel step, so that a match is only cut where the next region had found something
//       else.  Ranks, literal runs and coverage come from wave scans over the regions; every lane emits its own sequences.
// Any parse is a valid parse: what this loses against the exact greedy orbit is a match start within the last three positions
// of a region (found two or three bytes late by the next one).
// ---------------------------------------------------------------------------------------------------------------------
struct DenseLds {                        // laid over the tile arrays (tileLen .. jump), which the tile loop no longer needs
    u64 keep[kRegions];                  // bit = byte is a literal (not covered by a match, not before the entry cursor)
    u32 keepExcl[kRegions];              // literals of the pass before this region
    u64 covHi[kRegions];                 // bytes of the NEXT region that this lane's matches cover (its stretch runs up to espec)
    u32 espec[kRegions];                 // where the speculative parse leaves the region
    u32 lastEnd[kRegions];               // end of the lane's last match, 0 = no match
    u16 lastOff[kRegions];               // ... and its offset
    u16 cont[kRegions];                  // length of a first match that continues the match before it (absorbed by that one)
    u8  more[kRegions];                  // the lane has matches of its own besides
    u32 waveTot[3][16];                  // cross-wave scan: matches, kept literals, last end
};

// Tiles [tFrom, tTo) go into the hash tables exactly as the tile loop of lz_kernel would put them there (every position; the
// latest occurrence per bucket for later tiles, the first occurrence within the tile for the tile itself), without verifying or
// selecting anything.  From tile candFrom on, the POSITION (+ 1; 0 = none) of every position's candidate is written to candG
// (a period 1..4, else the table candidates as the finder ranks them; tags checked, bytes not — the dual finders, whose table
// entries carry no tag, keep the one of their four whose first 8 bytes agree furthest).
// A thread takes FOUR CONSECUTIVE positions: their 8-byte windows (and the 4 bytes in front, for the period test) come out of
// four aligned dwords with v_alignbyte, and their four candidates leave as one 8-byte store.
// Used for: the history / dictionary tiles in front of a block (no candidates), and step I of the region parse.
template <int MODE>
__device__ __forceinline__ void insert_tiles(LzLds& L, const u32 n, const u32 tFrom, const u32 tTo, const u32 candFrom, const u32 lowLimit,
                                             u16* __restrict__ candG, const u32 tid)
{
    u32* const table = L.tabMem;                           // fast (see lz_kernel for the layouts)
    u32* const first = L.tabMem + (1u << kHashLog);
    u32* const firstL = L.tabMem;                          // dual
    u32* const firstS = L.tabMem + (1u << (kHashLog - 1));
    u16* const tableL = reinterpret_cast<u16*>(L.tabMem + (1u << kHashLog));
    u16* const tableS = tableL + (1u << kHashLog);
    for (u32 t = tFrom; t < tTo; ++t) {
        const u32 tileStart = t * kTilePos;
        const bool wantCand = t >= candFrom;              // (uniform) earlier tiles only fill the tables
        const u32 stamp = ((kChunkSize / kTilePos) - t - 1) << kTileLog;
        const u32 q0 = 4 * tid, p0 = tileStart + q0;
        const u32* const d32 = reinterpret_cast<const u32*>(L.in) + (p0 >> 2);
        const u32 dm1 = p0 ? d32[-1] : 0u, d0 = d32[0], d1 = d32[1], d2 = d32[2];
        u64 w[kPPT]; u32 h[kPPT], h2[kPPT], cnd[kPPT]; bool valid[kPPT];
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            const u32 p = p0 + j;
            valid[j] = p + 8 <= n && p >= lowLimit; h[j] = 0; h2[j] = 0; cnd[j] = 0;
            w[j] = (u64)__builtin_amdgcn_alignbyte(d1, d0, j) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, j) << 32);
            if (valid[j]) {
                if (MODE == 0) {
                    h[j] = hash6p(w[j]);
                    cnd[j] = table[hidx(h[j])];
                    atomicMin(&first[hidx(h[j])], ((stamp + q0 + j) << 16) | htag(h[j]));
                } else {
                    h[j] = hash8p(w[j]); h2[j] = hash_shortp<5>(w[j]);
                    const u32 hL = hidx(h[j]), hS = hidx(h2[j]);
                    cnd[j] = (u32)tableL[hL] | ((u32)tableS[hS] << 16);
                    atomicMin(&firstL[hL >> 1], ((stamp + q0 + j) << 16) | htag(h[j])); atomicMin(&firstS[hS >> 1], ((stamp + q0 + j) << 16) | htag(h2[j]));
                }
            }
        }
        lds_barrier();
        u64 out4 = 0;
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            const u32 q = q0 + j, p = p0 + j;
            u32 cp = 0;                                   // candidate position + 1
            if (valid[j]) {
                const u32 lo = (u32)w[j], hi = (u32)(w[j] >> 32);
                const bool i4 = lo == hi;
                const bool i3 = __builtin_amdgcn_alignbyte(hi, lo, 3) == lo && ((hi ^ (hi >> 24)) & 0xFFu) == 0;
                u32 per = 0;
                if ((i4 | i3) && wantCand && p >= lowLimit + 4) {     // periods 1..4, as in the tile loop
                    const u32 prev4 = __builtin_amdgcn_alignbyte(d0, dm1, j);
                    const bool i2 = i4 && ((lo ^ (lo >> 16)) & 0xFFFFu) == 0, i1 = i2 && ((lo ^ (lo >> 8)) & 0xFFu) == 0;
                    if (i4 && prev4 == lo) per = 4;
                    if (i3 && (prev4 >> 8) == (lo & 0xFFFFFFu)) per = 3;
                    if (i2 && (prev4 >> 16) == (lo & 0xFFFFu)) per = 2;
                    if (i1 && (prev4 >> 24) == (lo & 0xFFu)) per = 1;
                }
                if (MODE == 0) {
                    const u32 tag = htag(h[j]);
                    atomicMax(&table[hidx(h[j])], ((p + 1) << 16) | tag);
                    if (per) cp = p - per + 1;
                    else if (wantCand) {
                        const u32 f = first[hidx(h[j])];
                        const u32 fq = (f >> 16) - stamp;
                        if (fq < q && (f & 0xFFFFu) == tag) cp = tileStart + fq + 1;
                        else if (cnd[j] && (cnd[j] & 0xFFFFu) == tag) cp = cnd[j] >> 16;
                    }
                } else {
                    const u32 hL = hidx(h[j]), hS = hidx(h2[j]);
                    const u32 eL = firstL[hL >> 1], eS = firstS[hS >> 1];
                    const u32 fL = eL >> 16, fS = eS >> 16;
                    if (fL == stamp + q) tableL[hL] = (u16)(p + 1);          // one writer per entry, as in the tile loop
                    if (fS == stamp + q) tableS[hS] = (u16)(p + 1);
                    if (per) cp = p - per + 1;
                    else if (wantCand) {
                        // the four candidates of the dual finder (same-tile / earlier, 8-byte / 5-byte hash); the table entries
                        // carry no tag, so the one whose first 8 bytes agree furthest is kept (in this order on ties), from 4 up
                        u32 cc[4];
                        cc[0] = (fL - stamp < q && (eL & 0xFFFFu) == htag(h[j])) ? tileStart + (fL - stamp) + 1 : 0u;
                        cc[1] = cnd[j] & 0xFFFFu;
                        cc[2] = (fS - stamp < q && (eS & 0xFFFFu) == htag(h2[j])) ? tileStart + (fS - stamp) + 1 : 0u;
                        cc[3] = cnd[j] >> 16;
                        u64 x[4];                          // all four reads in flight together (no candidate: position 0, result dropped)
#pragma unroll
                        for (u32 k = 0; k < 4; ++k) x[k] = w[j] ^ lds_load8(L.in, cc[k] ? cc[k] - 1 : 0u);
                        u32 best = 0, bestLen = 3;
#pragma unroll
                        for (u32 k = 0; k < 4; ++k) {
                            const u32 l8 = x[k] ? (ctz64(x[k]) >> 3) : 8u;
                            if (cc[k] && l8 > bestLen) { bestLen = l8; best = cc[k]; }
                        }
                        cp = best;
                    }
                }
            }
            out4 |= (u64)(cp & 0xFFFFu) << (16 * j);
        }
        if (wantCand && p0 < n) *reinterpret_cast<u64*>(candG + p0) = out4;      // (entries behind the data's end are zero: no candidate)
        lds_barrier();                                    // (the tables; the candidates only have to have arrived before step II reads them)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Hash-chain search for level >= 5 (MODE 2), the place of ZSTD_HcFindBestMatch / ZSTD_RowFindBestMatch behind the greedy and
// lazy parsers (U/ZstdLazy.cs:619-760, 1101-1309; searchLog 3 at this level and size, U/Clevels.cs): every position is linked to
// an earlier position with the same 5-byte hash — the most recent one in front of its 64-position group — and a position's
// candidate is the LONGEST match among the first `depth` (1 << searchLog, as the reference's nbAttempts) positions down its chain (the reference's chain table,
// ZSTD_insertAndFindFirstIndex_internal, U/ZstdLazy.cs:572-617, walked as in :660-720).
// The reference builds the chain by walking the input in order.  Here all threads hash a tile's 4096 positions, then ONE wave
// takes the tile's 64 groups of 64 consecutive positions in order: read the table (latest occurrence so far) -> that is the
// link; then ds_max the group's own positions into the table.  LDS operations of one wave execute in program order, so a group
// sees every group before it and nothing of itself: deterministic, no sort, 64 x 4 LDS instructions per tile.  (Exact links —
// a stable radix sort of the tile by hash, neighbours in sorted order — were built first and measured: 7.3 us per tile against
// 1.5 for this, for 0.2 - 1.4 % of compressed size on the CPU model of this parse, tests/lab/finder_lab.c.)
// Links of the current tile and the three before it stay in LDS, all links go to global memory (2 bytes per position, L2), where
// the walk follows them further back.
// Measured with tests/lab/finder_lab.c (same parse on the CPU): chains of depth 8 against one candidate per position: -7 % compressed
// size on text, -11 % on Python sources.
// ---------------------------------------------------------------------------------------------------------------------
constexpr u32 kHcCont  = 12;             // a position whose left neighbour holds a match this long takes its continuation unsearched
constexpr u32 kHcRing = 4;               // tiles whose links stay in LDS: the current one and the three before it
struct HcLds {                           // laid over the 64 KiB of the hash tables
    u32 table[1u << kHashLog];           // position + 1 of the hash's latest occurrence so far (0 = none)
    u16 links[kHcRing * kTilePos];       // links (position + 1, 0 = none) of the last 16384 positions: the one of position p at p & 16383
};
static_assert(sizeof(HcLds) == sizeof(u32) * (2u << kHashLog), "HcLds fills the tables' place");

__device__ __forceinline__ void hc_tiles(LzLds& L, const u32 n, const u32 tFrom, const u32 tTo, const u32 candFrom, const u32 lowLimit, const u32 depth,
                                         u16* __restrict__ candG, u16* __restrict__ chainG, const u32 tid, const u32 lane, const u32 wave
#ifdef ZMI_LZ_STAMPS
                                         , unsigned long long* dAcc, unsigned long long& dLast
#endif
                                         )
{
    HcLds& Hc = *reinterpret_cast<HcLds*>(L.tabMem);
    // the tile's hashes (0xFFFF: not a position to link): over the tile loop's arrays, idle here
    u16* const hashes = reinterpret_cast<u16*>(&L.tileLen[0]);
    static_assert(offsetof(LzLds, tileLen) + kTilePos * sizeof(u16) <= sizeof(LzLds), "the hashes must fit behind the tables");
    {   // no occurrences yet
        uint4* const t4 = reinterpret_cast<uint4*>(Hc.table);
        t4[tid] = uint4{0u, 0u, 0u, 0u}; t4[tid + kTile] = uint4{0u, 0u, 0u, 0u};
    }
    for (u32 t = tFrom; t < tTo; ++t) {
        const u32 tileStart = t * kTilePos;
        u16* const tilePred = Hc.links + (tileStart & (kHcRing * kTilePos - 1));
        const u32 q0 = 4 * tid, p0 = tileStart + q0;
        const u32* const d32 = reinterpret_cast<const u32*>(L.in) + (p0 >> 2);
        const u32 dm1 = p0 ? d32[-1] : 0u, d0 = d32[0], d1 = d32[1], d2 = d32[2], d3 = d32[3], d4 = d32[4];
        u64 a1[4], a2[4]; bool valid[4];
        {   // ---- hashes: four consecutive positions per thread ----
            u64 h4 = 0;
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {
                const u32 p = p0 + j;
                a1[j] = (u64)__builtin_amdgcn_alignbyte(d1, d0, j) | ((u64)__builtin_amdgcn_alignbyte(d2, d1, j) << 32);
                a2[j] = (u64)__builtin_amdgcn_alignbyte(d3, d2, j) | ((u64)__builtin_amdgcn_alignbyte(d4, d3, j) << 32);
                valid[j] = p + 8 <= n && p >= lowLimit;
                h4 |= (u64)(valid[j] ? hidx(hash_shortp<5>(a1[j])) : 0xFFFFu) << (16 * j);
            }
            *reinterpret_cast<u64*>(&hashes[q0]) = h4;
        }
        ZMI_DSTAMP(16);
        lds_barrier();
        ZMI_DSTAMP(17);
        // ---- links: one wave, group after group (see above); four groups' reads and updates per step ----
        if (wave == 0) {                                   // eight groups per step, the next step's hashes already on their way
            u32 hn[8];
#pragma unroll
            for (u32 k = 0; k < 8; ++k) hn[k] = hashes[k * 64 + lane];
#pragma unroll 1
            for (u32 g = 0; g < 64; g += 8) {
                u32 h[8], pv[8];
#pragma unroll
                for (u32 k = 0; k < 8; ++k) h[k] = hn[k];
                if (g + 8 < 64) {
#pragma unroll
                    for (u32 k = 0; k < 8; ++k) hn[k] = hashes[(g + 8 + k) * 64 + lane];
                }
                u32 hx[8], val[8];                         // (branch-free: a position that is not linked adds 0 to bucket 0)
#pragma unroll
                for (u32 k = 0; k < 8; ++k) { const bool link = h[k] != 0xFFFFu; hx[k] = link ? h[k] : 0u; val[k] = link ? tileStart + (g + k) * 64 + lane + 1 : 0u; }
#pragma unroll
                for (u32 k = 0; k < 8; ++k) { pv[k] = Hc.table[hx[k]]; atomicMax(&Hc.table[hx[k]], val[k]); }
#pragma unroll
                for (u32 k = 0; k < 8; ++k) tilePred[(g + k) * 64 + lane] = (u16)(val[k] ? pv[k] : 0u);
            }
        }
        ZMI_DSTAMP(18);
        lds_barrier();
        ZMI_DSTAMP(19);
        if (tid < 512) reinterpret_cast<uint4*>(chainG + tileStart)[tid] = reinterpret_cast<const uint4*>(tilePred)[tid];
        // ---- search ----
        if (t >= candFrom) {                               // uniform
            u32 c[4], best[4], bestLen[4];
#pragma unroll
            for (u32 j = 0; j < 4; ++j) {
                const u32 p = p0 + j;
                best[j] = 0; bestLen[j] = 3; c[j] = valid[j] ? (u32)tilePred[q0 + j] : 0u;
                const u32 lo = (u32)a1[j], hi = (u32)(a1[j] >> 32);
                const bool i4 = lo == hi;
                const bool i3 = __builtin_amdgcn_alignbyte(hi, lo, 3) == lo && ((hi ^ (hi >> 24)) & 0xFFu) == 0;
                if ((i4 | i3) && valid[j] && p >= lowLimit + 4) {   // periods 1..4 (runs, tiny patterns): the candidate is the period itself
                    const u32 prev4 = __builtin_amdgcn_alignbyte(d0, dm1, j);
                    const bool i2 = i4 && ((lo ^ (lo >> 16)) & 0xFFFFu) == 0, i1 = i2 && ((lo ^ (lo >> 8)) & 0xFFu) == 0;
                    u32 per = 0;
                    if (i4 && prev4 == lo) per = 4;
                    if (i3 && (prev4 >> 8) == (lo & 0xFFFFFFu)) per = 3;
                    if (i2 && (prev4 >> 16) == (lo & 0xFFFFu)) per = 2;
                    if (i1 && (prev4 >> 24) == (lo & 0xFFu)) per = 1;
                    if (per) { best[j] = p - per + 1; bestLen[j] = 64; c[j] = 0; }
                }
            }
            // A candidate can only beat the best so far if it agrees with the position on the four bytes that END one past the
            // best length (the reference's own filter, U/ZstdLazy.cs:690-696): one unaligned dword per candidate instead of sixteen
            // bytes; only those that pass are measured in full.  pw = those four bytes of the position.
            // A position whose left neighbour found kHcCont bytes or more takes that match's continuation (same offset, one byte
            // shorter) without a search of its own — the parse only ever looks at such a position for the lazy step, which a match
            // that long does not lose (tests/lab/finder_lab.c: 0.1 - 0.4 % of compressed size).
            // What the walk costs is instruction issue under divergence: in nearly every step SOME lane of the wave has a candidate
            // to measure in full, so every step pays that path (measured: skipping 40 % of the positions, or two chains per lane in
            // flight instead of one, both leave the time where it is; an attempt costs 1.0 - 1.7 ms per GiB).
            // Positions 0 and 2 of a thread are walked side by side, then 1 and 3 (which may take their left neighbour's
            // continuation): two chains in flight per lane hide each other's LDS latency.
            // (one LDS read whatever the tile — an index, not a choice between arrays: a pointer chosen among LDS and global memory
            //  makes the load a flat one, which waits for both memories; only positions further back take the branch to L2)
            const u32 ringLow = tileStart >= (kHcRing - 1) * kTilePos ? tileStart - (kHcRing - 1) * kTilePos : 0u;
            //  (the pointers carry their address spaces: left generic, the compiler turns the branch into a choice of pointer again)
            const __attribute__((address_space(3))) u16* const linksL = (const __attribute__((address_space(3))) u16*)Hc.links;
            const __attribute__((address_space(1))) u16* const linksG = (const __attribute__((address_space(1))) u16*)chainG;
            auto link = [&](u32 cpos) -> u32 {
                const u32 near = linksL[cpos & (kHcRing * kTilePos - 1)];
                u32 far = 0;                               // (registers of their own: neither load waits for the other)
                if (cpos < ringLow) far = linksG[cpos];
                return cpos < ringLow ? far : near;
            };
            auto measure = [&](auto J, u32 cpos, u32& pw) {           // the candidate passed the filter: its full length, up to 64
                constexpr u32 j = decltype(J)::value;
                const u32 p = p0 + j;
                u64 c1, c2;
                lds_load16(L.in, cpos, c1, c2);
                const u64 x1 = a1[j] ^ c1, x2 = a2[j] ^ c2;
                u32 l = x1 ? (ctz64(x1) >> 3) : 8u + (x2 ? (ctz64(x2) >> 3) : 8u);
                if (l == 16) {
                    while (l < 64) {
                        const u64 x = lds_load8(L.in, p + l) ^ lds_load8(L.in, cpos + l);
                        if (x) { l += ctz64(x) >> 3; break; }
                        l += 8;
                    }
                    if (l > 64) l = 64;
                }
                if (l > n - p) l = n - p;
                if (l > bestLen[j]) { bestLen[j] = l; best[j] = cpos + 1; pw = lds_load4(L.in, p + l - 3); }
            };
            auto pair_walk = [&](auto JA, auto JB) {
                constexpr u32 ja = decltype(JA)::value, jb = decltype(JB)::value;
                u32 pwa = (u32)a1[ja], pwb = (u32)a1[jb];
#pragma unroll 1
                for (u32 d = 0; d < depth && (c[ja] | c[jb]); ++d) {
                    const bool ha = c[ja] != 0, hb = c[jb] != 0;
                    const u32 ca = ha ? c[ja] - 1 : 0u, cb = hb ? c[jb] - 1 : 0u;      // (idle: position 0, in bounds, result unused)
                    const u32 chka = lds_load4(L.in, ca + bestLen[ja] - 3), chkb = lds_load4(L.in, cb + bestLen[jb] - 3);
                    const u32 nxa = link(ca), nxb = link(cb);
                    if (ha) { if (chka == pwa) measure(JA, ca, pwa); c[ja] = bestLen[ja] >= 64 ? 0u : nxa; }
                    if (hb) { if (chkb == pwb) measure(JB, cb, pwb); c[jb] = bestLen[jb] >= 64 ? 0u : nxb; }
                }
            };
            auto take_over = [&](auto J) {                            // the left neighbour's match, one byte shorter
                constexpr u32 j = decltype(J)::value;
                if (c[j] && best[j - 1] && bestLen[j - 1] >= kHcCont && bestLen[j - 1] < 64) { best[j] = best[j - 1] + 1; bestLen[j] = bestLen[j - 1] - 1; c[j] = 0; }
            };
            using I0 = std::integral_constant<u32, 0>; using I1 = std::integral_constant<u32, 1>; using I2 = std::integral_constant<u32, 2>; using I3 = std::integral_constant<u32, 3>;
            pair_walk(I0{}, I2{});
            take_over(I1{}); take_over(I3{});
            pair_walk(I1{}, I3{});
            u64 out4 = 0;
#pragma unroll
            for (u32 j = 0; j < 4; ++j) out4 |= (u64)(best[j] & 0xFFFFu) << (16 * j);
            if (p0 < n) *reinterpret_cast<u64*>(candG + p0) = out4;
        }
        ZMI_DSTAMP(20);
        // (the search is done with the hashes and with the ring slot the next tile takes over)
        // the links in global memory are read by tiles at least kHcRing after this one: every kHcRing-th tile waits for the stores
        if (t % kHcRing == kHcRing - 1) __syncthreads(); else lds_barrier();
        ZMI_DSTAMP(22);
    }
}

// MODE as in lz_kernel (0 fast, 1 dual-hash, 2 dual-hash + lazy deferral).  hist / lowLimit: the history in front of the block in
// LDS (positions below lowLimit are padding); insertFrom: first tile whose positions still have to go into the tables (lz_region_kernel
// starts from empty tables: the history tiles and the block's first tile; inlined in lz_kernel: fromTile).
template <int MODE>
__device__ __forceinline__ void dense_rest(LzLds& L, const u32 n, const u32 insertFrom, const u32 fromTile, const u32 nTiles, const u32 hist, const u32 lowLimit,
                                           u16* __restrict__ candG, u16* __restrict__ chainG, const u32 hcDepth,
                                           Seq* __restrict__ seqOut, u8* __restrict__ litOut,
                                           u32& cursor, u32& nbSeq, u32& litBase, bool& deferred, const u32 tid, const u32 lane, const u32 wave)
{
    static_assert(sizeof(DenseLds) <= sizeof(L.tileLen) + sizeof(L.tileOff) + sizeof(L.jump), "DenseLds must fit over tileLen .. jump");
    DenseLds& D = *reinterpret_cast<DenseLds*>(&L.tileLen[0]);
    // positions of a pass's matches by rank: behind DenseLds, over what is left of the tile loop's arrays (all idle by now)
    constexpr u32 kMaxPassSeq = (kPassPos + 64) / 4 + 8;
    static_assert(offsetof(LzLds, tileLen) + sizeof(DenseLds) + kMaxPassSeq * sizeof(u16) <= sizeof(LzLds), "rank list must fit behind DenseLds");
    u16* const rankPos = reinterpret_cast<u16*>(reinterpret_cast<u8*>(&D) + sizeof(DenseLds));
#ifdef ZMI_LZ_STAMPS
    unsigned long long dLast = __builtin_amdgcn_s_memtime(); unsigned long long dAcc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#if defined(ZMI_EXP_STOP) && ZMI_EXP_STOP == 1       // ablation builds (diagnostic, make variant DEFS=-DZMI_EXP_STOP=n): stop after a phase
    return;
#endif
    // ---------------- I: candidates of every position, tile by tile ----------------
#if defined(ZMI_EXP_STOP) && ZMI_EXP_STOP == 3
    if constexpr (MODE == 2) { hc_tiles(L, n, lowLimit >> kTileLog, hist >> kTileLog, fromTile, lowLimit, hcDepth, candG, chainG, tid, lane, wave); __syncthreads(); return; }
#endif
    if constexpr (MODE == 2) hc_tiles(L, n, lowLimit >> kTileLog, nTiles, fromTile, lowLimit, hcDepth, candG, chainG, tid, lane, wave
#ifdef ZMI_LZ_STAMPS
                                      , dAcc, dLast
#endif
                                      );
    else insert_tiles<MODE>(L, n, insertFrom, nTiles, fromTile, lowLimit, candG, tid);      // (MODE 2 always comes with its chain plane)
    __syncthreads();
    ZMI_DSTAMP(10);
#if defined(ZMI_EXP_STOP) && ZMI_EXP_STOP == 2
    return;
#endif
    if (deferred) {                                       // (uniform) the bytes counted so far, out of LDS (see the tile loop)
        for (u32 q16 = tid * 16; q16 < litBase; q16 += kTile * 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(L.in + hist + q16);
            u32u* o = (u32u*)(litOut + q16);
            o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        }
        deferred = false;
    }
    // ---------------- II: passes of kPassPos positions ----------------
    u16* const C = reinterpret_cast<u16*>(L.tabMem);
    for (u32 lo = fromTile * kTilePos; lo < n; lo += kPassPos) {
        const u32 hi = lo + kPassPos < n ? lo + kPassPos : n;
        const u32 nReg = (hi - lo + 63) >> 6;
        {   // candidates of the pass (+ one region: a lane's stretch may reach that far): global (L2) -> LDS, 16 bytes per lane,
            // the four loads of a thread in flight together
            const u32 cnt = hi - lo + 64;
            uint4 v[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) { const u32 i = (tid + k * kTile) * 8; v[k] = *reinterpret_cast<const uint4*>(candG + lo + (i < cnt ? i : 0)); }
#pragma unroll
            for (u32 k = 0; k < 4; ++k) { const u32 i = (tid + k * kTile) * 8; if (i < cnt) *reinterpret_cast<uint4*>(C + i) = v[k]; }
        }
        lds_barrier();
        ZMI_DSTAMP(11);
        const u32 rs = lo + tid * 64, re = rs + 64 < hi ? rs + 64 : hi;
        const bool mine = tid < nReg;
        // the candidate at position p verified against the input: match length (0 = none), at most 64 and never past the data
        auto verify = [&](u32 p, u32 cpos) -> u32 {
            u64 a1, a2, c1, c2;
            lds_load16(L.in, p, a1, a2); lds_load16(L.in, cpos, c1, c2);
            const u64 x1 = a1 ^ c1, x2 = a2 ^ c2;
            u32 l = x1 ? (ctz64(x1) >> 3) : 8u + (x2 ? (ctz64(x2) >> 3) : 8u);
            if (l == 16) {
                while (l < 64) {
                    const u64 x = lds_load8(L.in, p + l) ^ lds_load8(L.in, cpos + l);
                    if (x) { l += ctz64(x) >> 3; break; }
                    l += 8;
                }
                if (l > 64) l = 64;
            }
            if (l > n - p) l = n - p;
            return l >= 4 ? l : 0u;
        };
        // lazy deferral (MODE 2; U/ZstdLazy.cs:1836-1870): the match at p yields to the one at p + 1 while that one gains more
        // (4 bits per byte saved, minus log2 of the offset, plus 4 for the literal it costs)
        auto lazy = [&](u32& p, u32& l, u32& cpos, const u32 bound) {
            for (;;) {
                if (p + 1 >= bound || p + 9 > n) break;
                const u32 c2 = C[p + 1 - lo];
                if (!c2) break;
                const u32 l2 = verify(p + 1, c2 - 1);
                if (!l2) break;
                const int g1 = (int)(l * 4) - (int)highbit32(p - cpos + 1) + 4;
                const int g2 = (int)(l2 * 4) - (int)highbit32(p + 1 - (c2 - 1) + 1);
                if (g2 <= g1) break;
                ++p; l = l2; cpos = c2 - 1;
            }
        };
        // ---- step 1, speculative: where does the parse LEAVE this region?  The lane starts kWarm positions in front of its region
        // (the first region of a pass at the cursor it was handed): a greedy parse forgets where it started within a few matches,
        // so it usually leaves the region exactly where the parse that comes through the regions before it will.  Nothing is written ----
        constexpr u32 kWarm = 16;
        const bool lastReg = tid + 1 == nReg;
        u32 eSpec = 0;
#ifdef ZMI_EXP_SPEC
        for (u32 rep = 0; rep < ZMI_EXP_SPEC; ++rep)
#endif
        if (mine) {
            u32 p = tid == 0 ? rs : rs - kWarm;
            if (p < cursor) p = cursor;
            while (p < re) {
                // four candidates per LDS round trip: positions without one (literals) cost no round trip of their own
                const u64 c4 = *reinterpret_cast<const u64u*>(C + (p - lo));
                if (!c4) { p += 4; continue; }
                const u32 skip = ctz64(c4) >> 4;
                p += skip;
                if (p >= re) break;
                const u32 cp = (u32)(c4 >> (16 * skip)) & 0xFFFFu;
                u32 l = (p + 8 <= n) ? verify(p, cp - 1) : 0u;
                if (MODE == 2 && l) { u32 cq = cp - 1; lazy(p, l, cq, re); }
                p += l ? l : 1u;
            }
            if (p < re) p = re;
            eSpec = p;
            D.espec[tid] = eSpec;
        }
        ZMI_DSTAMP(12);
        lds_barrier();
        ZMI_DSTAMP(12);
        // ---- step 2, for real: from where the region before this one is left by ITS speculation to exactly where this region's
        // own speculation left it (there the next lane starts: no byte is parsed twice, none is skipped; the pass's last region
        // simply runs to its end).  A match is cut where the lane's stretch ends.  A taken match is recorded where its candidate
        // was (offset) and one entry further (length); a match that starts where the one before it ended, with the same offset,
        // is that match going on (the verification stops at 64 bytes) ----
        u64 selLo = 0, selHi = 0, covLo = 0, covHi = 0;     // Lo: positions of this region; Hi: the stretch beyond it, up to eSpec
        u32 lastEnd = 0, lastOff = 0, lastStart = 0, firstOff = 0, firstLen = 0, firstStart = 0;
        if (mine) {
            u32 p = tid == 0 ? rs : D.espec[tid - 1];
            if (p < cursor) p = cursor;
            const u32 bound = lastReg ? re : eSpec;
            u32 anchor = p;
            while (p < bound) {
                const u64 c4 = *reinterpret_cast<const u64u*>(C + (p - lo));
                if (!c4) { p += 4; continue; }
                const u32 skip = ctz64(c4) >> 4;
                p += skip;
                if (p >= bound) break;
                const u32 cp = (u32)(c4 >> (16 * skip)) & 0xFFFFu;
                if (p + 8 > n) { ++p; continue; }
                u32 cpos = cp - 1;
                u32 l = verify(p, cpos);
                if (MODE == 2 && l) lazy(p, l, cpos, bound);
                if (!lastReg && p + l > eSpec) l = eSpec - p;
                if (l < 4) { ++p; continue; }
                const u32 off = p - cpos;
                while (p > anchor && cpos > lowLimit && L.in[p - 1] == L.in[cpos - 1]) { --p; --cpos; ++l; }     // ZstdFast.cs:242-247
                const u32 q = p - rs;                                    // 0 .. 126
                {   // bytes [q, q + l) of the lane's stretch (a backward extension can make a match longer than the 64 verified bytes)
                    const u32 e = q + l;
                    if (q < 64) covLo |= (~0ull << q) & (e >= 64 ? ~0ull : ((1ull << e) - 1));
                    if (e > 64) { const u32 hs = q > 64 ? q - 64 : 0u, he = e - 64; covHi |= (~0ull << hs) & (he >= 64 ? ~0ull : ((1ull << he) - 1)); }
                }
                if (lastEnd == p && lastOff == off && (selLo | selHi)) {
                    C[lastStart - lo + 1] = (u16)(C[lastStart - lo + 1] + l);      // the match before it, going on
                    if (lastStart == firstStart) firstLen += l;
                } else {
                    C[p - lo] = (u16)off; C[p - lo + 1] = (u16)l;
                    if (!(selLo | selHi)) { firstOff = off; firstLen = l; firstStart = p; }
                    if (q < 64) selLo |= 1ull << q; else selHi |= 1ull << (q - 64);
                    lastStart = p;
                }
                p += l; anchor = p; lastOff = off; lastEnd = p;
            }
            D.lastEnd[tid] = lastEnd; D.lastOff[tid] = (u16)lastOff; D.covHi[tid] = covHi;
        }
        ZMI_DSTAMP(13);
        // ---- where the pass's last match ends (the next pass's cursor) ----
        {
            const u32 wE = wave_max(lastEnd);
            if (lane == 0) D.waveTot[2][wave] = wE;
        }
        lds_barrier();
        ZMI_DSTAMP(13);
        u32 totE = cursor;
#pragma unroll
        for (u32 k2 = 0; k2 < 16; ++k2) { const u32 e = D.waveTot[2][k2]; totE = e > totE ? e : totE; }
        // ---- links: a lane's first match that starts exactly where the match before it ends, with the same offset, is that
        // match going on (a long match or a run comes out of the lanes as a chain of pieces): the chain's head owns it ----
        u32 absorbed = 0;
        if (mine) {
            if (tid > 0) covLo |= D.covHi[tid - 1];                            // what the lane before this one did beyond its region
            if (cursor > rs) covLo |= cursor - rs >= 64 ? ~0ull : ((1ull << (cursor - rs)) - 1);     // a match from before the pass
            if ((selLo | selHi) && tid > 0 && firstStart == D.lastEnd[tid - 1] && firstOff == (u32)D.lastOff[tid - 1]) {
                absorbed = firstLen;
                if (selLo) selLo &= selLo - 1; else selHi &= selHi - 1;
            }
            D.cont[tid] = (u16)absorbed; D.more[tid] = (selLo | selHi) ? (u8)1 : (u8)0;
        }
        lds_barrier();
        // ---- what every lane contributes: sequences; what every region holds: literals ----
        u32 ext = 0;
        if (mine && (selLo | selHi)) {
            for (u32 e = tid + 1; e < nReg; ++e) { const u32 ce = D.cont[e]; if (!ce) break; ext += ce; if (D.more[e]) break; }
        }
        const u32 validBits = mine ? re - rs : 0u;
        const u64 keep = mine ? ~covLo & (validBits >= 64 ? ~0ull : ((1ull << validBits) - 1)) : 0ull;
        const u32 nMatch = popc64(selLo) + popc64(selHi), nKeep = popc64(keep);
        const u32 inclM = wave_scan_incl(nMatch), inclK = wave_scan_incl(nKeep);
        if (lane == 63) { D.waveTot[0][wave] = inclM; D.waveTot[1][wave] = inclK; }
        lds_barrier();
        u32 baseM = 0, baseK = 0, totM = 0, totK = 0;
#pragma unroll
        for (u32 k2 = 0; k2 < 16; ++k2) {
            const u32 m = D.waveTot[0][k2], kk = D.waveTot[1][k2];
            if (k2 < wave) { baseM += m; baseK += kk; }
            totM += m; totK += kk;
        }
        ZMI_DSTAMP(14);
        if (mine) { D.keep[tid] = keep; D.keepExcl[tid] = baseK + inclK - nKeep; }
        // ---- sequences.  A lane's matches have consecutive ranks, so lanes storing their own records hit 64 different cache lines
        // per store instruction (measured: a tenth of the kernel on text).  Instead every lane files the POSITIONS of its matches
        // by rank (and adds to its last match what the lanes after it continue), and after a barrier the whole workgroup builds
        // the records rank by rank: consecutive lanes, consecutive records, whole lines.  The literals in front of a match run
        // from the end of the match one rank below it (the first of the pass: from the cursor the pass was entered with) ----
        if (mine && nMatch) {
            u32 r = baseM + inclM - nMatch, lastQ = 0;
            u64 bLo = selLo, bHi = selHi;
            while (bLo) { lastQ = ctz64(bLo); bLo &= bLo - 1; rankPos[r++] = (u16)(rs - lo + lastQ); }
            while (bHi) { lastQ = 64u + ctz64(bHi); bHi &= bHi - 1; rankPos[r++] = (u16)(rs - lo + lastQ); }
            if (ext) C[rs - lo + lastQ + 1] = (u16)(C[rs - lo + lastQ + 1] + ext);
        }
        lds_barrier();
        for (u32 k = tid; k < totM; k += kTile) {
            const u32 pos = rankPos[k], prev = k ? rankPos[k - 1] : 0u;
            const u32 off = C[pos], len = C[pos + 1], plen = C[prev + 1];
            const u32 pe = k ? lo + prev + plen : cursor;
            Seq sq; sq.offBase = off + 3; sq.litLength = (u16)(lo + pos - pe); sq.mlBase = (u16)(len - 3);
            // (streaming stores, here and for the literals: what this kernel should keep in L2 is its candidate plane)
            __builtin_nontemporal_store((u64)sq.offBase | ((u64)sq.litLength << 32) | ((u64)sq.mlBase << 48), reinterpret_cast<u64*>(seqOut + nbSeq + k));
        }
        lds_barrier();
        // ---- literals: compacted in LDS first (the candidates' place: they are spent), then written out in whole 16-byte pieces:
        // on dense data the runs between matches are three or four bytes long, and a byte store to global memory each is what
        // this step would otherwise consist of.  32 positions per thread, two threads per region ----
        {
            u8* const S = reinterpret_cast<u8*>(L.tabMem);
            const u32 g = tid >> 1, half = tid & 1u;
            if (g < nReg) {
                const u64 kg = D.keep[g];
                u32 k32 = half ? (u32)(kg >> 32) : (u32)kg;
                if (k32) {
                    u8* o = S + D.keepExcl[g] + (half ? popc64(kg & 0xFFFFFFFFull) : 0u);
                    const u32 pos0 = lo + g * 64 + half * 32;
                    if (k32 == 0xFFFFFFFFu) {             // nothing matched here: 32 bytes straight through
                        const uint4 v0 = *reinterpret_cast<const uint4*>(L.in + pos0), v1 = *reinterpret_cast<const uint4*>(L.in + pos0 + 16);
                        u32u* o4 = (u32u*)o; o4[0] = v0.x; o4[1] = v0.y; o4[2] = v0.z; o4[3] = v0.w; o4[4] = v1.x; o4[5] = v1.y; o4[6] = v1.z; o4[7] = v1.w;
                    } else {                              // dword by dword out of registers: stores only, nothing waits on LDS
                        const uint4 v0 = *reinterpret_cast<const uint4*>(L.in + pos0), v1 = *reinterpret_cast<const uint4*>(L.in + pos0 + 16);
                        const u32 d[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
#pragma unroll
                        for (u32 k = 0; k < 8; ++k) {
                            const u32 keep = (k32 >> (4 * k)) & 0xFu, w4 = d[k];
                            if (keep == 0xFu) { *(u32u*)o = w4; o += 4; }
                            else { if (keep & 1) *o++ = (u8)w4; if (keep & 2) *o++ = (u8)(w4 >> 8); if (keep & 4) *o++ = (u8)(w4 >> 16); if (keep & 8) *o++ = (u8)(w4 >> 24); }
                        }
                    }
                }
            }
            lds_barrier();
            u8* const dst = litOut + litBase;
            // (litBase is whatever earlier tiles left: the head brings the destination to 16-byte alignment)
            u32 head = (u32)((16 - ((uintptr_t)dst & 15)) & 15); if (head > totK) head = totK;
            if (tid < head) dst[tid] = S[tid];
            const u32 body = (totK - head) >> 4;
            for (u32 i = tid; i < body; i += kTile) {
                uint4 v; const u8* sp = S + head + 16 * i;
                v.x = *(const u32u*)sp; v.y = *(const u32u*)(sp + 4); v.z = *(const u32u*)(sp + 8); v.w = *(const u32u*)(sp + 12);
                typedef u32 __attribute__((ext_vector_type(4))) v4u;
                __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w}, reinterpret_cast<v4u*>(dst + head + 16 * i));
            }
            const u32 done = head + (body << 4);
            if (tid < totK - done) dst[done + tid] = S[done + tid];
        }
        lds_barrier();
        ZMI_DSTAMP(15);
        cursor = totE; nbSeq += totM; litBase += totK;
    }
#ifdef ZMI_LZ_STAMPS
    if ((tid & 63u) == 0) for (int i = 0; i < 14; i++) atomicAdd(&g_lzStamps[10 + i], dAcc[i]);
#endif
}

// MODE 0 = fast strategy (one 6-byte hash; levels 1-2 and the negative levels); 1 = doubleFast strategy (8-byte + SHORT-byte
// hashes, four candidates per position; levels 3-4: the place of U/ZstdDoubleFast.cs:51-247); 2 = greedy and above (the dual
// candidates + one-step lazy deferral; levels >= 5: the place of U/ZstdLazy.cs:1743-2032).  The host maps strategy -> MODE.
// DICT: a dictionary prefix is present (its bounds checks fold away otherwise)
// FAR: matches may start in the input in front of the block (same frame): those candidates are verified against global memory
template <int MODE, int SHORT, bool DICT, bool FAR>
__global__ __launch_bounds__(1024) void lz_kernel(const u8* __restrict__ src, u64 srcSize,
                                                  Seq* __restrict__ seqs, u8* __restrict__ lits,
                                                  ChunkMeta* __restrict__ meta,
                                                  const u8* __restrict__ prefixArg, const u32 prefixLenArg, const u32 chunkBytes,
                                                  const u32 fhExtra, const u32 minStrideLog, const u32 frameBlocksArg, u16* __restrict__ candAll, u16* __restrict__ chainAll, u32* __restrict__ regionList, const u32 nChunks,
                                                  u32* __restrict__ claimCtr)
{
    extern __shared__ __attribute__((aligned(16))) u8 ldsRaw[];
    LzLds& L = *reinterpret_cast<LzLds*>(ldsRaw);
    const u32 tid = threadIdx.x, lane = lane_id(), wave = uniform(wave_id());
    // With one workgroup per CU (LDS) nothing else hides the load latency at the start of a chunk, so the NEXT chunk's bytes are
    // fetched into registers while this one is parsed (plain chunks only: a dictionary or history in front of the chunk keeps the
    // direct path).  Which chunk is next: claimCtr == null: c + gridDim.x.  Otherwise (a grid of one workgroup per CU) the
    // workgroups CLAIM chunks from a counter as they go — chunks of unequal cost balance by themselves, and every chunk but a
    // workgroup's first arrives prefetched (a stamped build had 42 % of an unprefetched Zipf chunk's time in this load).  The claim
    // is made one chunk ahead of the prefetch (thread 0 holds the answer in a register through a chunk), so nobody waits for it.
    constexpr bool kPrefetch = MODE == 0 && !DICT && !FAR;      // (the dual-hash finders have no 16 registers to spare)
    uint4 pf0 = {0, 0, 0, 0}, pf1 = pf0, pf2 = pf0, pf3 = pf0; bool pfValid = false;
    const bool claiming = kPrefetch && claimCtr != nullptr;     // uniform
    u32 claimed = 0;                                            // thread 0: the chunk after the next
    if (claiming && tid == 0) claimed = atomicAdd(claimCtr, 1u) + gridDim.x;
    u32 cNext = 0;
    for (u32 c = blockIdx.x; c < nChunks; c = cNext) {
    // Raw-content dictionary (row f-4; ZSTD_loadDictionaryContent, U/ZstdCompress.cs:5126-5237): its last `prefixLen` bytes
    // sit in front of the chunk in LDS, ending at a tile boundary (hist = whole tiles of history, positions below lowLimit
    // are padding and never referenced).  History tiles only go into the tables (insert_tiles); the parse starts
    // with the cursor at `hist`, so everything after it is untouched: offsets simply reach back into the history.
    // chunkBytes = 64 KiB - hist (64 KiB without a dictionary).
    // Cross-chunk history (row f-1; the block loop's window, U/ZstdCompress.cs:4705-4807): frameBlocks > 0 makes every
    // `frameBlocks` consecutive chunks the blocks of ONE frame, and a block's history is the input in front of it — up to
    // `hist` bytes, as far back as its frame reaches — staged in the same place a dictionary's tail would be.
    const u32 cb = DICT ? chunkBytes : kChunkSize;
    const u32 hist = DICT ? kChunkSize - ((chunkBytes + kTilePos - 1) & ~(kTilePos - 1)) : 0u;      // (chunks below a tile: ZSTD_c_windowLog 10, 11)
    const u64 base = (u64)c * cb;
    const u8* __restrict__ in = src + base;
    // (frameBlocksArg bit 31: the frame's blocks are independent of each other — windows below 64 KiB, where a block IS the window:
    //  no history; a dictionary is history of the frame's first block only, whose image is the layout the decoder sees)
    const u32 frameBlocks = frameBlocksArg & 0x7FFFFFFFu;
    const bool indep = (frameBlocksArg >> 31) != 0;
    const u32 bf = ((DICT || FAR) && frameBlocks) ? c % frameBlocks : 0u;               // block index inside its frame
    const u32 farAvail = FAR ? ((u64)bf * cb < kFarMax ? bf * cb : kFarMax) : 0u;      // bytes of far history in front of the block
    u32 prefixLen = prefixLenArg; const u8* __restrict__ prefix = prefixArg;
    if (DICT && frameBlocks && !indep) { const u64 back = (u64)bf * cb; prefixLen = back < hist ? (u32)back : hist; prefix = in - prefixLen; }
    if (DICT && indep && bf) prefixLen = 0;
    const u32 lowLimit = DICT ? hist - prefixLen : 0u;
    const u32 nData = (u32)((srcSize - base) < cb ? (srcSize - base) : cb);
    const u32 n = hist + nData;                            // end of the data in LDS
#ifdef ZMI_LZ_STAMPS
    unsigned long long stampAcc[14] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif

    // ---- stage the chunk: 16 B per lane when the source is 16-byte aligned ----
    if (DICT && frameBlocks && !indep && ((((uintptr_t)prefix) | lowLimit) & 15) == 0) {
        // cross-chunk history: the history and the block are ONE contiguous piece of the input (at most 64 KiB): four 16-byte
        // pieces per thread, all loads in flight before the first store (pieces past the end re-read piece 0 and are not stored)
        const uint4* in4 = reinterpret_cast<const uint4*>(prefix);
        uint4* l4 = reinterpret_cast<uint4*>(L.in + lowLimit);
        const u32 bytes = prefixLen + nData, full = bytes >> 4;
        uint4 v0, v1, v2, v3;
        const u32 i0 = tid, i1 = tid + kTile, i2 = tid + 2 * kTile, i3 = tid + 3 * kTile;
        v0 = in4[i0 < full ? i0 : 0]; v1 = in4[i1 < full ? i1 : 0]; v2 = in4[i2 < full ? i2 : 0]; v3 = in4[i3 < full ? i3 : 0];
        if (i0 < full) l4[i0] = v0;
        if (i1 < full) l4[i1] = v1;
        if (i2 < full) l4[i2] = v2;
        if (i3 < full) l4[i3] = v3;
        for (u32 i = (full << 4) + tid; i < bytes; i += kTile) L.in[lowLimit + i] = prefix[i];
        for (u32 i = tid; i < lowLimit; i += kTile) L.in[i] = 0;
    } else {
    if (kPrefetch && pfValid) {                            // (uniform) a full aligned chunk, already in registers
        uint4* l4 = reinterpret_cast<uint4*>(L.in);
        l4[tid] = pf0; l4[tid + kTile] = pf1; l4[tid + 2 * kTile] = pf2; l4[tid + 3 * kTile] = pf3;
    } else if ((((uintptr_t)in) & 15) == 0) {
        const uint4* in4 = reinterpret_cast<const uint4*>(in);
        uint4* l4 = reinterpret_cast<uint4*>(L.in + hist);
        const u32 full = nData >> 4;
        if (full) {                                 // a full chunk is four 16-byte pieces per thread: all loads in flight, then the stores
            uint4 v0, v1, v2, v3;                   // (pieces past the end re-read piece 0 and are not stored)
            const u32 i0 = tid, i1 = tid + kTile, i2 = tid + 2 * kTile, i3 = tid + 3 * kTile;
            v0 = in4[i0 < full ? i0 : 0]; v1 = in4[i1 < full ? i1 : 0]; v2 = in4[i2 < full ? i2 : 0]; v3 = in4[i3 < full ? i3 : 0];
            if (i0 < full) l4[i0] = v0;
            if (i1 < full) l4[i1] = v1;
            if (i2 < full) l4[i2] = v2;
            if (i3 < full) l4[i3] = v3;
        }
        for (u32 i = (full << 4) + tid; i < nData; i += kTile) L.in[hist + i] = in[i];
    } else {
        for (u32 i = tid; i < nData; i += kTile) L.in[hist + i] = in[i];
    }
    if (DICT) for (u32 i = tid; i < hist; i += kTile) L.in[i] = i >= lowLimit ? prefix[i - lowLimit] : (u8)0;
    }
    for (u32 i = n + tid; i < kChunkSize + kInPad; i += kTile) L.in[i] = 0;
    u32* const endOf = reinterpret_cast<u32*>(L.jumpB);
    u32* const table = L.tabMem;                           // fast
    u32* const first = L.tabMem + (1u << kHashLog);
    u32* const firstL = L.tabMem;                          // dual
    u32* const firstS = L.tabMem + (1u << (kHashLog - 1));
    u16* const tableL = reinterpret_cast<u16*>(L.tabMem + (1u << kHashLog));
    u16* const tableS = tableL + (1u << kHashLog);
    {   // fast: table = 0 (empty), first = ~0; dual: firstL|firstS = ~0, tableL|tableS = 0 — 32 KiB each, 16 bytes per store
        uint4* const t4 = reinterpret_cast<uint4*>(L.tabMem);
        const uint4 z = {0u, 0u, 0u, 0u}, f = {~0u, ~0u, ~0u, ~0u};
#pragma unroll
        for (u32 k = 0; k < 2; ++k) { t4[tid + k * kTile] = MODE == 0 ? z : f; t4[2 * kTile + tid + k * kTile] = MODE == 0 ? f : z; }
    }
    if (tid == 0) { L.nzWords[0] = 0; L.nzWords[1] = 0; L.nzWords[2] = 0; L.matchCount[0] = 0; L.matchCount[1] = 0; L.matchCount[2] = 0; }
    pfValid = false;
    cNext = c + gridDim.x;
    if (claiming && tid == 0) { L.claim = claimed; claimed = atomicAdd(claimCtr, 1u) + gridDim.x; }
    __syncthreads();
    if (claiming) cNext = L.claim;
    if (kPrefetch) {
        const u32 cN = cNext;
        const u8* __restrict__ inN = src + (u64)cN * kChunkSize;
        if (cN < nChunks && srcSize - (u64)cN * kChunkSize >= kChunkSize && (((uintptr_t)inN) & 15) == 0) {      // uniform
            const uint4* n4 = reinterpret_cast<const uint4*>(inN);
            pf0 = n4[tid]; pf1 = n4[tid + kTile]; pf2 = n4[tid + 2 * kTile]; pf3 = n4[tid + 3 * kTile];
            pfValid = true;
        }
    }
    if (FAR && farAvail) {
        // the table starts out holding the input in front of the block (what the reference's table still holds from the blocks
        // before, U/ZstdFast.cs:9-46): latest occurrence per bucket, straight from global memory
        for (u32 i = tid * kFarStep; i < farAvail; i += kTile * kFarStep) {
            const u32 hp = hash6p(readLE64(in - farAvail + i));
            atomicMax(&table[hidx(hp)], ((kFarMax - farAvail + i + 1) << 14) | (htag(hp) >> 2));
        }
        __syncthreads();
    }
    ZMI_STAMP(0);

    Seq* __restrict__ seqOut = seqs + (u64)c * kMaxSeq;
    u8* __restrict__ litOut = lits + (u64)c * kLitStride;
    // parse state, kept identically in every thread's registers (all updates come from LDS values read after a barrier)
    u32 cursor = hist;   // absolute position where the parse of the previous tiles ended (= end of the last selected match)
    u32 nbSeq = 0, litBase = 0;
    // As long as the chunk has no sequence its literals are its own bytes from the start: they are counted, not copied.  The first
    // selected match makes up for it (one copy out of LDS); a chunk that ends without sequences never writes its literals at all
    // and the entropy stage reads them from the source (ChunkMeta::litFromSrc) — incompressible-by-LZ data, e.g. BASELINE's Zipf bytes.
    bool deferred = true;

    // one selected match -> its sequence + its coverage bits (used by the dense and the sparse path)
    u32 covPar = 0;                      // coverage-mask slot of the current tile
    u64* cov = L.covMask[0];             // coverage words of the current tile (64, or 256 for a super-tile)
    u32 span = kTilePos;                 // positions the current tile covers
    // ix = index into the tile arrays (the position, or the probe slot of a super-tile), q = tile-relative position
    auto tile_len = [&](u32 ix) -> u32 { return FAR ? (u32)L.tileLen[ix] & 63u : (u32)L.tileLen[ix]; };
    auto tile_off = [&](u32 ix) -> u32 { return FAR ? (u32)L.tileOff[ix] | (((u32)L.tileLen[ix] >> 6) << 16) : (u32)L.tileOff[ix]; };
    auto emit_match = [&](u32 tileStart, u32 ix, u32 q, u32 rank, u32 end) {
        u32 p = tileStart + q;
        const u32 off = tile_off(ix);
        const u32 litStart = endOf[rank];
        const u32 floorPos = litStart > tileStart ? litStart : tileStart;     // literals of earlier tiles are already emitted
        // backward: give bytes of the pending literal run to the match while they agree (ZstdFast.cs:242-247)
        // (four bytes per step: the dwords in front of the match and of its source, compared from the top)
        if (FAR && off > p) {                                          // the source lies in front of the block: byte steps against global memory
            while (p > floorPos && off - p < farAvail && L.in[p - 1] == in[(s64)p - 1 - (s64)off]) --p;
        } else for (;;) {
            u32 room = p - floorPos;                                   // bytes the pending literal run can give
            const u32 srcRoom = p - off - lowLimit;                    // bytes in front of the source
            if (srcRoom < room) room = srcRoom;
            if (room == 0) break;
            if (p - off < 4) {                                         // source within 4 bytes of the start: byte steps
                if (L.in[p - 1] != L.in[p - off - 1]) break;
                --p; continue;
            }
            const u32 x = lds_load4(L.in, p - 4) ^ lds_load4(L.in, p - off - 4);
            u32 m = x ? (u32)__builtin_clz(x) >> 3 : 4u;               // byte p-1 is the dword's top byte
            if (m > room) m = room;
            p -= m;
            if (m < 4) break;
        }
        Seq sq; sq.offBase = off + 3; sq.litLength = (u16)(p - litStart); sq.mlBase = (u16)(end - p - 3);
        seqOut[nbSeq + rank] = sq;
        const u32 r0 = p - tileStart, r1 = (end - tileStart) < span ? (end - tileStart) : span;
        for (u32 wI = r0 >> 6; wI <= ((r1 - 1) >> 6); ++wI) {
            u64 m = ~0ull;
            if (wI == (r0 >> 6)) m &= ~0ull << (r0 & 63);
            if (wI == ((r1 - 1) >> 6)) m &= ~0ull >> (63 - ((r1 - 1) & 63));
            atomicOr((unsigned long long*)&cov[wI], (unsigned long long)m);
        }
    };
    // full length of a match that hit the cap: 64 lanes x 8 bytes per step (whole wave, uniform arguments)
    auto finish_capped = [&](u32 p, u32 off) -> u32 {
        u32 e = p + kLenCap;
        for (;;) {
            const u32 pos = e + 8 * lane;              // reads past n land in the table region: harmless, clamped below
            const s32 sp = (s32)pos - (s32)off;                        // FAR: a negative source position is in front of the block
            const u64 x = lds_load8(L.in, pos) ^ ((FAR && sp < 0) ? readLE64(in + sp) : lds_load8(L.in, (u32)sp));
            const u64 bad = ballot(x != 0 || pos + 8 > n);
            if (bad == 0) { e += 512; continue; }
            const u32 fl = ctz64(bad);
            const u32 cnt = x ? (ctz64(x) >> 3) : 8;
            e += 8 * fl + read_lane(cnt, fl);
            break;
        }
        return e > n ? n : e;
    };

    // Matches may start where 8 bytes are still readable (the reference stops at iend-8, ZstdFast.cs:110).
    const u32 nTiles = (n + kTilePos - 1) / kTilePos;
    // Where the previous tile found next to nothing, this tile probes every 2nd or 4th position only — the reference's own
    // acceleration (ZSTD_fast's step = 1 + ((ip - anchor) >> kSearchStrength), U/ZstdFast.cs:130-136).  A match that starts
    // between probed positions is still picked up one or two bytes later and grown backward at emission.
    u32 regionCursor = 0;                // != 0: the chunk goes on in lz_region_kernel (parse cursor + 1)
    u32 prevDensity = 0xFFFFFFFFu;       // matches per 4096 positions in the previous tile (scaled by its stride)
    u32 prevStride = 0;                  // the previous iteration's stride
    u64* const superCov = reinterpret_cast<u64*>(L.jump);      // up to 1024 coverage words of a super-tile (L.jump is idle outside dense tiles)
    // history / dictionary tiles (those wholly below lowLimit are padding) only fill the tables: searched, never parsed
    if (DICT && lowLimit < hist) insert_tiles<MODE>(L, n, lowLimit >> kTileLog, hist >> kTileLog, 0xFFFFFFFFu, lowLimit, nullptr, tid);
    ZMI_STAMP(8);                        // (dense-tile select shares the slot: told apart by the workload)
#ifdef ZMI_EXP_LZSTOP
    for (u32 t = hist >> kTileLog, it = 0; t < nTiles && it < ZMI_EXP_LZSTOP - 1; ++it) {      // (ablation build: the first N - 1 iterations only)
#else
    for (u32 t = hist >> kTileLog, it = 0; t < nTiles; ++it) {
#endif
        const u32 tileStart = t * kTilePos;
        // stride: every 2nd / 4th position after a sparse tile; where a strided iteration found next to nothing either, every
        // 16th (the reference's step grows faster still while nothing matches: 64 after 16 KiB, U/ZstdFast.cs:130-136)
        u32 strideSel = prevDensity < 8 ? 2u : (prevDensity < 32 ? 1u : 0u);            // uniform
        if (prevDensity < 4 && prevStride >= 2) strideSel = 4u;       // (straight to 16 — the rest of the chunk in one iteration: Zipf bytes 1.16 -> 1.08 ms per GiB)
        else if (prevDensity == 0 && t != 0) strideSel = 3;                // nothing at all in the previous tile(s), whatever their stride
        // negative levels (ZSTD_fast with a step, U/ZstdFast.cs:101-103): never denser than the step asks for; history is still inserted in full
        if (strideSel < minStrideLog) strideSel = minStrideLog;                          // uniform
        // Super-tile: where only every 2nd / 4th (8th, 16th) position is probed, TWO / FOUR tiles (as many as are left in
        // full) are taken in one iteration — up to 4096 probes, four per thread as in a dense tile, so their LDS latencies
        // overlap and the two barriers are paid once per 8 / 16 KiB.  The tile arrays are then indexed by probe slot
        // (slot order = position order), coverage has up to 1024 words, and the selection is the serial walk of wave 0
        // whatever the number of matches (capped; what is left out stays literals).
        // 4096 probes per iteration whatever the stride, up to the rest of the chunk at 16: an iteration costs its three barriers
        // (3.3 us; 0.2 ms per GiB) far more than its probes — measured against at most 16 KiB per iteration (the stride adapting
        // again after that): Zipf bytes 1.37 -> 1.15 ms per GiB, random bytes 1.07 -> 0.86, the mixed corpus 1.71 -> 1.60, same sizes
        u32 nSubT = 1u << strideSel;
        { const u32 fullLeft = (n - tileStart) >> kTileLog; if (nSubT > fullLeft) nSubT = fullLeft; }
        const bool super = nSubT >= 2;                                                  // uniform
        if (!super) { nSubT = 1; if (strideSel > 2) strideSel = 2; }                    // (a lone tile knows strides 1, 2, 4 only)
        const u32 strideLog = strideSel;
        span = nSubT << kTileLog;
        const u32 slots = span >> strideLog;                                            // probes of this iteration (<= 4096)
        // first-occurrence entries: base + tile-relative position, 16 bits, smaller for later tiles (atomicMin keeps the
        // current tile's earliest).  Tile t owns [(15-t) << 12, +4096); a super-tile owns the ranges of the tiles it covers.
        const u32 stamp = ((kChunkSize / kTilePos) - t - nSubT) << kTileLog;
        const u32 nPass = (slots + kTile - 1) / kTile, par = it % 3;
        covPar = it & 1;
        cov = super ? superCov : L.covMask[covPar];
        // (A sparse tile could fuse probe and verify and defer its table inserts behind the verify barrier — one barrier
        // less, measured 7 % faster — but the next tile's probes would then race with those inserts and the output would
        // depend on wave timing.  Determinism is part of the contract, so the two-barrier form stays.)
        constexpr bool fused = false;
        // probed position of lattice cell c = j * kTile + tid: c * stride + a pseudo-random residue, so that a repeat of
        // earlier data lines up with inserted positions one time in `stride` whatever its distance (a fixed lattice would
        // never see a repeat whose distance is not a multiple of the stride)
        auto slot_pos = [&](u32 cI) -> u32 {
            if (strideLog == 0) return cI;                   // (uniform) dense tile: every position
            return (cI << strideLog) + ((((tileStart >> kTileLog) * kTilePos + cI) * 2654435761u >> 27) & ((1u << strideLog) - 1));
        };
        auto probed = [&](u32 j) -> u32 { return slot_pos(j * kTile + tid); };
        // index into tileLen/tileOff/the match masks: the position, or the probe slot in a super-tile
        auto arr_ix = [&](u32 j, u32 q) -> u32 { return super ? j * kTile + tid : q; };
        // ---------------- probe ----------------
        // fast: h = hash product, cand = table entry.  dual: h = long product, h2 = short product, cand = tableL | tableS << 16
        u64 w[kPPT], w2[kPPT]; u32 h[kPPT], h2[kPPT], cand[kPPT]; bool valid[kPPT];
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            const u32 q = probed(j), p = tileStart + q;
            valid[j] = j < nPass && j * kTile + tid < slots && p + 8 <= n && p >= lowLimit; w[j] = 0; w2[j] = 0; h[j] = 0; h2[j] = 0; cand[j] = 0;
            if (j >= nPass) continue;            // uniform
            if (valid[j]) {
                lds_load16(L.in, p, w[j], w2[j]);
                if (MODE == 0) {
                    h[j] = hash6p(w[j]);
                    cand[j] = table[hidx(h[j])];
                    if (!fused) atomicMin(&first[hidx(h[j])], ((stamp + q) << 16) | htag(h[j]));
                } else {
                    h[j] = hash8p(w[j]); h2[j] = hash_shortp<SHORT>(w[j]);
                    const u32 hL = hidx(h[j]), hS = hidx(h2[j]);
                    cand[j] = (u32)tableL[hL] | ((u32)tableS[hS] << 16);
                    atomicMin(&firstL[hL >> 1], ((stamp + q) << 16) | htag(h[j])); atomicMin(&firstS[hS >> 1], ((stamp + q) << 16) | htag(h2[j]));
                }
            }
        }
        ZMI_STAMP(1);
        // (LDS-only barriers in this loop: __syncthreads() also waits for the wave's outstanding global loads — the NEXT chunk's
        //  bytes, fetched on purpose behind this chunk's work — and for every sequence and literal store to be acknowledged)
        if (!fused) lds_barrier();             // every probe of this tile precedes every insert of this tile
        ZMI_STAMP(2);
        u64 mmJ[kPPT], cmJ[kPPT];
        const bool slotMasks = strideLog == 0 || super;      // (uniform) array index = j * kTile + tid: a wave's ballots ARE its mask words
        if (!slotMasks && lane < 4) {            // strided tile: a wave's probes fall into its own four groups; matching lanes set bits below
            const u32 g = strideLog == 2 ? wave * 4 + lane : (lane >> 1) * 32 + wave * 2 + (lane & 1);
            L.matchMask[g] = 0; L.capMask[g] = 0; L.selMask[g] = 0; L.covMask[covPar][g] = 0;
        }
        if (super && tid < nSubT * kGroups) superCov[tid] = 0;
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            mmJ[j] = 0; cmJ[j] = 0;
            if (j >= nPass) continue;            // uniform
            const u32 q = probed(j), p = tileStart + q;
            u32 len = 0, off = 0;
            if (valid[j]) {
                if (MODE == 0 && !fused) atomicMax(&table[hidx(h[j])], FAR ? ((kFarMax + p + 1) << 14) | (htag(h[j]) >> 2) : ((p + 1) << 16) | htag(h[j]));
                // (1) periods 1..4: bytes p..p+7 repeat with period d and the d bytes before p agree — runs and tiny
                //     patterns, which neither table can see inside one tile
                u32 per = 0;
                {
                    // prefilter with two 32-bit tests: period 4 over the 8 bytes (implied by periods 1 and 2) or period 3
                    const u32 lo = (u32)w[j], hi = (u32)(w[j] >> 32);
                    const bool i4 = lo == hi;
                    const bool i3 = __builtin_amdgcn_alignbyte(hi, lo, 3) == lo && ((hi ^ (hi >> 24)) & 0xFFu) == 0;
                    if ((i4 | i3) && p >= lowLimit + 4) {                     // rare outside runs: only then look at the bytes before p
                        const u32 prev4 = lds_load4(L.in, p - 4);
                        const bool i2 = i4 && ((lo ^ (lo >> 16)) & 0xFFFFu) == 0, i1 = i2 && ((lo ^ (lo >> 8)) & 0xFFu) == 0;
                        if (i4 && prev4 == lo) per = 4;
                        if (i3 && (prev4 >> 8) == (lo & 0xFFFFFFu)) per = 3;
                        if (i2 && (prev4 >> 16) == (lo & 0xFFFFu)) per = 2;
                        if (i1 && (prev4 >> 24) == (lo & 0xFFu)) per = 1;
                    }
                }
                if (MODE == 0) {
                    if (per) { len = match_len(L, p, p - per, w[j], w2[j], n); off = per; }
                    else {
                        // (2) same-tile first occurrence, (3) latest occurrence in earlier tiles: keep the longer, nearer on ties
                        const u32 tag = htag(h[j]);
                        const u32 f = fused ? 0xFFFFFFFFu : first[hidx(h[j])];
                        const u32 fq = (f >> 16) - stamp;                 // (an entry seen here was written by this tile: see `stamp`)
                        if (!fused && fq < q && (f & 0xFFFFu) == tag) {
                            const u32 cpos = tileStart + fq;
                            len = match_len(L, p, cpos, w[j], w2[j], n); off = p - cpos;
                        }
                        if (FAR) {
                            if (cand[j] && (cand[j] & 0x3FFFu) == (tag >> 2) && len < kLenCap) {
                                const u32 crel = (cand[j] >> 14) - 1;                 // kFarMax + position; below kFarMax: in front of the block
                                const u32 l2 = crel >= kFarMax ? match_len(L, p, crel - kFarMax, w[j], w2[j], n)
                                                               : match_len_far(L, p, in - (kFarMax - crel), w[j], n);
                                if (l2 > len) { len = l2; off = kFarMax + p - crel; }
                            }
                        } else if (cand[j] && (cand[j] & 0xFFFFu) == tag && len < kLenCap) {
                            const u32 cpos = (cand[j] >> 16) - 1;
                            const u32 l2 = match_len(L, p, cpos, w[j], w2[j], n);
                            if (l2 > len) { len = l2; off = p - cpos; }
                        }
                    }
                } else {
                    const u32 hL = hidx(h[j]), hS = hidx(h2[j]);
                    const u32 eL = firstL[hL >> 1], eS = firstS[hS >> 1];
                    const u32 fL = eL >> 16, fS = eS >> 16;
                    // the first position of a bucket in this tile becomes the bucket's entry for later tiles: one writer per
                    // entry, no atomic (two positions with the same 13-bit hash share the 12-bit bucket)
                    if (fL == stamp + q) tableL[hL] = (u16)(p + 1);
                    if (fS == stamp + q) tableS[hS] = (u16)(p + 1);
                    if (per) { len = match_len(L, p, p - per, w[j], w2[j], n); off = per; }
                    // candidates, longest wins, nearer on ties: in-tile long, earlier-tile long, in-tile short, earlier-tile short
                    u32 c0p = 0xFFFFFFFFu, c1p = 0xFFFFFFFFu;
                    if (len < kLenCap && fL - stamp < q && (eL & 0xFFFFu) == htag(h[j])) {
                        c0p = tileStart + (fL - stamp);
                        const u32 l2 = match_len(L, p, c0p, w[j], w2[j], n);
                        if (l2 > len || (l2 == len && l2 && p - c0p < off)) { len = l2; off = p - c0p; }
                    }
                    if (len < kLenCap && (cand[j] & 0xFFFFu)) {
                        c1p = (cand[j] & 0xFFFFu) - 1;
                        const u32 l2 = match_len(L, p, c1p, w[j], w2[j], n);
                        if (l2 > len) { len = l2; off = p - c1p; }
                    }
                    if (len < kLenCap && fS - stamp < q && (eS & 0xFFFFu) == htag(h2[j])) {
                        const u32 cp = tileStart + (fS - stamp);
                        if (cp != c0p) {
                            const u32 l2 = match_len(L, p, cp, w[j], w2[j], n);
                            if (l2 > len || (l2 == len && l2 && p - cp < off)) { len = l2; off = p - cp; }
                        }
                    }
                    if (len < kLenCap && (cand[j] >> 16)) {
                        const u32 cp = (cand[j] >> 16) - 1;
                        if (cp != c1p) {
                            const u32 l2 = match_len(L, p, cp, w[j], w2[j], n);
                            if (l2 > len) { len = l2; off = p - cp; }
                        }
                    }
                }
            }
            if (MODE != 0 && SHORT == 4) {
                // a 4-byte match far away costs more than its literals (offset bits + three codes against ~5 bits a byte)
                if (len == 4 && off >= 256) len = 0;
            }
            if (MODE == 2 && strideLog == 0) {
                // lazy deferral (U/ZstdLazy.cs:1836-1870): a match yields to the one starting one byte later when that one
                // gains more (4 bits per byte saved, minus log2 of the offset, plus 4 for the literal it costs).  The next
                // position lives in the next lane; the last lane of a wave keeps its match.
                const u32 nLen = __shfl_down(len, 1), nOff = __shfl_down(off, 1);
                if (len && len < kLenCap && lane < 63 && nLen) {
                    const int g1 = (int)(len * 4) - (int)highbit32(off + 1) + 4;
                    const int g2 = (int)(nLen * 4) - (int)highbit32(nOff + 1);
                    if (g2 > g1) len = 0;
                }
            }
            if (len) { const u32 ix = arr_ix(j, q); L.tileLen[ix] = (u8)(FAR ? len | ((off >> 16) << 6) : len); L.tileOff[ix] = (u16)off; }      // only read where matchMask has the bit
            mmJ[j] = ballot(len != 0); cmJ[j] = ballot(len == kLenCap);
            if (!slotMasks) {
                if (len) {
                    atomicOr((unsigned long long*)&L.matchMask[q >> 6], 1ull << (q & 63));
                    if (len == kLenCap) atomicOr((unsigned long long*)&L.capMask[q >> 6], 1ull << (q & 63));
                    atomicOr((unsigned long long*)&L.nzWords[par], 1ull << (q >> 6));
                }
                if (lane == 0 && mmJ[j]) atomicAdd(&L.matchCount[par], popc64(mmJ[j]));
            }
        }
        if (slotMasks) {        // the wave's four groups of masks are written together by lanes 0..3 (one predicated block instead of four)
            u64 mmL = mmJ[0], cmL = cmJ[0];
#pragma unroll
            for (u32 k = 1; k < kPPT; ++k) { mmL = lane == k ? mmJ[k] : mmL; cmL = lane == k ? cmJ[k] : cmL; }
            const u32 nMatch = popc64(mmJ[0]) + popc64(mmJ[1]) + popc64(mmJ[2]) + popc64(mmJ[3]);
            if (lane < kPPT) {
                const u32 g = lane * 16 + wave;
                L.matchMask[g] = mmL; L.capMask[g] = cmL; L.selMask[g] = 0; L.covMask[covPar][g] = 0;
                if (mmL) atomicOr((unsigned long long*)&L.nzWords[par], 1ull << g);
                if (lane == 0 && nMatch) atomicAdd(&L.matchCount[par], nMatch);
            }
        }
        ZMI_STAMP(3);
        lds_barrier();
        ZMI_STAMP(4);
        const u32 c0 = cursor > tileStart ? cursor - tileStart : 0;       // entry cursor, tile-relative
        const u32 matchCount = L.matchCount[par];
        if (tid == 0) { const u32 nx = (it + 2) % 3; L.nzWords[nx] = 0; L.matchCount[nx] = 0; }    // slot of the iteration after next: idle until the next barrier
        if (fused) {                           // the deferred inserts of a sparse tile (every probe of the tile came before the barrier)
#pragma unroll
            for (u32 j = 0; j < kPPT; ++j) if (valid[j]) atomicMax(&table[hidx(h[j])], ((tileStart + probed(j) + 1) << 16) | htag(h[j]));
        }
        // matches per 4096 positions had every position been probed
        prevDensity = (matchCount << strideLog) / nSubT;
        prevStride = strideLog;
        const bool any = matchCount != 0 && c0 < span;                   // uniform
        const bool dense = any && matchCount > 64 && !super;
        if (dense) {
            // ---------------- select: the orbit of the greedy parse, segment by segment ----------------
            // The parse is the orbit of "next match at or after the end of this one" from the entry cursor.  A segment is 64
            // positions, a wave owns four consecutive ones (lane = position inside the segment).  Because a match is at most
            // kLenCap = 32 long here, the cursor enters a segment at one of its first 32 positions, so a segment IS a function
            // entry offset -> exit offset (into the next segment).  Each wave computes it for all entries at once by pointer
            // doubling in registers (five shuffle rounds: a step advances by >= 4 from a match, 2^5 steps leave the segment),
            // composes its four segments, and publishes 32 bytes.  After ONE barrier every wave chains the functions of the
            // waves before it (readlane on values held in registers), then walks its own segments from the now known entry
            // with the single-step jumps: at most 16 matches per segment, a readlane each.
            u8* const segFn = reinterpret_cast<u8*>(L.jumpB + 3584);          // [16][32]; jumpB holds: endOf u32[1026] | selPos u16[1024] at 2304 | segFn
            u32* const segExit0 = reinterpret_cast<u32*>(L.jumpB + 3584 + 256);
            u16* const selPos = L.jumpB + 2304;
            const u32 w0 = c0 >> 8, k0 = (c0 >> 6) & 3u, e0 = c0 & 63u;       // (uniform) where the entry cursor sits
            u32 J1[4], X[4]; u64 mmS[4];
#pragma unroll
            for (u32 k = 0; k < 4; ++k) {
                const u32 g = wave * 4 + k;
                const u64 mmv = L.matchMask[g];
                const u64 mm = (u64)uniform((u32)mmv) | ((u64)uniform((u32)(mmv >> 32)) << 32);
                mmS[k] = mm;
                const bool has = (mm >> lane) & 1ull;
                const u32 len = has ? tile_len(g * 64 + lane) : 0u;
                const u32 t = lane + len;                                     // a match: its end; no match here: the cursor itself
                u32 j = t;                                                    // >= 64: leaves the segment at offset t - 64 (< 32)
                if (t < 64) { const u64 rest = mm >> t; j = rest ? t + ctz64(rest) : 64u; }
                J1[k] = j; X[k] = j;
            }
#pragma unroll
            for (u32 r = 0; r < 5; ++r) {
#pragma unroll
                for (u32 k = 0; k < 4; ++k) { const u32 nx = (u32)__shfl((int)X[k], (int)(X[k] & 63u)); X[k] = X[k] < 64 ? nx : X[k]; }
            }
            {   // the wave's four segments composed: entry offset (lane, < 32) -> exit offset into the next wave's first segment
                u32 y = X[0] - 64;
#pragma unroll
                for (u32 k = 1; k < 4; ++k) y = (u32)__shfl((int)X[k], (int)y) - 64;
                if (lane < 32) segFn[wave * 32 + lane] = (u8)y;
                if (wave == w0) {                                             // (uniform) the wave the entry cursor falls into: from segment k0, offset e0
                    u32 z = 0;
#pragma unroll
                    for (u32 k = 0; k < 4; ++k) {
                        if (k == k0) z = read_lane(X[k], e0) - 64;
                        else if (k > k0) z = read_lane(X[k], z) - 64;
                    }
                    if (lane == 0) *segExit0 = z;
                }
            }
            ZMI_STAMP(8);
            lds_barrier();
            ZMI_STAMP(8);
            if (wave >= w0) {                                                 // (uniform) earlier waves lie before the cursor: nothing selected
                u32 ent = e0, kStart = k0;
                if (wave > w0) {
                    u32 fv[16];
#pragma unroll
                    for (u32 k = 0; k < 16; ++k) fv[k] = segFn[k * 32 + (lane & 31u)];
                    ent = uniform(*segExit0); kStart = 0;
#pragma unroll
                    for (u32 k = 1; k < 15; ++k) if (k > w0 && k < wave) ent = read_lane(fv[k], ent);
                }
                ZMI_STAMP(8);
#pragma unroll
                for (u32 k = 0; k < 4; ++k) {
                    if (k < kStart) continue;                                 // uniform
                    const u64 rest = mmS[k] >> ent;
                    u32 p = rest ? ent + ctz64(rest) : 64u;
                    u64 mark = 0;
                    while (p < 64) { mark |= 1ull << p; p = read_lane(J1[k], p); }
                    ent = p - 64;
                    if (lane == 0) L.selMask[wave * 4 + k] = mark;
                }
            }
            ZMI_STAMP(8);
            lds_barrier();
            ZMI_STAMP(8);
            if (wave == 0) {
                // ---- finish capped matches in order; drop the selections they swallow ----
                u32 from = 0;                  // consider selected capped matches at tile positions >= from
                while (from < kTilePos) {
                    u64 wm = L.selMask[lane] & L.capMask[lane];       // lane = group index; re-read every time
                    if (lane < (from >> 6)) wm = 0; else if (lane == (from >> 6)) wm &= ~0ull << (from & 63);
                    const u64 have = ballot(wm != 0);
                    if (!have) break;
                    const u32 g = ctz64(have);
                    const u32 q = g * 64 + ctz64(read_lane64(wm, g));
                    const u32 p = tileStart + q;
                    const u32 e = finish_capped(p, tile_off(q));
                    if (lane == 0) L.jump[q] = (u16)(e - p > 0xFFFFu ? 0xFFFFu : e - p);
                    const u32 r0 = q + 1, r1 = (e - tileStart) < kTilePos ? (e - tileStart) : kTilePos;   // swallowed: [r0, r1)
                    if (r1 > r0) {
                        const u32 w0 = r0 >> 6, w1 = (r1 - 1) >> 6;
                        if (lane >= w0 && lane <= w1) {
                            u64 m = ~0ull;
                            if (lane == w0) m &= ~0ull << (r0 & 63);
                            if (lane == w1) m &= ~0ull >> (63 - ((r1 - 1) & 63));
                            L.selMask[lane] &= ~m;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                    }
                    from = r1 > r0 ? r1 : r0;
                }
                // ---- ranks: selected matches before each group ----
                const u32 cnt = popc64(L.selMask[lane]);
                const u32 incl = wave_scan_incl(cnt);
                L.wordRank[lane] = incl - cnt;
                if (lane == 63) L.wordRank[64] = incl;
                if (lane == 0) endOf[0] = cursor;
            }
            lds_barrier();
            ZMI_STAMP(9);
            // every selected match files its position and its end under its rank; after the barrier the matches are emitted BY RANK,
            // one per lane from lane 0 up (a tile selects a few hundred of its 4096 positions: emitting by position would run the
            // emission code sixty-four times per thread row with a handful of lanes active)
#pragma unroll
            for (u32 j = 0; j < kPPT; ++j) {
                const u32 q = j * kTile + tid;
                const u64 sm = L.selMask[q >> 6];
                if ((sm >> (q & 63)) & 1ull) {
                    const u32 rank = L.wordRank[q >> 6] + popc64(sm & ((1ull << (q & 63)) - 1));
                    const u32 len = tile_len(q);
                    endOf[rank + 1] = tileStart + q + (len == kLenCap ? (u32)L.jump[q] : len);
                    selPos[rank] = (u16)q;
                }
            }
            lds_barrier();
            {
                const u32 nSelT = L.wordRank[64];
                if (tid < nSelT) { const u32 q = selPos[tid]; emit_match(tileStart, q, q, tid, endOf[tid + 1]); }
            }
        } else if (any) {
            // ---------------- sparse tile (<= 64 matches) or super-tile: wave 0 parses it alone, exactly greedy ----------------
            // (a super-tile may hold more than 64 matches: they are walked 64 at a time, at most kSuperMax of them — the
            //  rest stays literals; it happens where sparse data turns dense, and the next tile is a dense one)
            if (wave == 0) {
                constexpr u32 kSuperMax = 1024, kFrugalLen = 12;
                const u32 mcount = matchCount < kSuperMax ? matchCount : kSuperMax;
                const bool frugal = matchCount <= 2;
                const u64 mmw = L.matchMask[lane];                     // lane = group of 64 array slots
                const u32 cntW = popc64(mmw);
                const u32 rankW = wave_scan_incl(cntW) - cntW;          // matches before this lane's group
                u32 cur = c0, nSelTot = 0;
                if (lane == 0) endOf[0] = cursor;
                for (u32 b0 = 0; b0 < mcount; b0 += 64) {
                    {   // matches of rank [b0, b0 + 64) in position order -> one per lane
                        u32 r = rankW;
                        for (u64 b = mmw; b; b &= b - 1, ++r) if (r - b0 < 64u) L.sparseList[r - b0] = (u16)(lane * 64 + ctz64(b));
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                    const u32 nB = mcount - b0 < 64u ? mcount - b0 : 64u;
                    const bool have = lane < nB;
                    const u32 ix = have ? L.sparseList[lane] : 0;
                    const u32 q = super ? slot_pos(ix) : ix;
                    const u32 len = have ? tile_len(ix) : 0, off = have ? tile_off(ix) : 0;
                    u32 myEnd = 0; u64 selBits = 0;
                    for (u32 i = 0; i < nB; ++i) {
                        const u32 qi = read_lane(q, i);
                        if (qi < cur) continue;
                        const u32 li = read_lane(len, i);
                        // one or two short matches in 4-16 KiB do not pay for a sequences section (a lone sequence costs about
                        // four bytes, a 6-byte match saves about as much): they stay literals, and a chunk with nothing else
                        // stays a pure literals block — no sequence stages on either side, literals decoded in place
                        if (frugal && li < kFrugalLen) continue;
                        u32 e = tileStart + qi + li;
                        if (li == kLenCap) e = finish_capped(tileStart + qi, read_lane(off, i));
                        if (lane == i) myEnd = e;
                        selBits |= 1ull << i;
                        cur = e - tileStart;
                    }
                    const bool sel = (selBits >> lane) & 1ull;
                    const u32 rank = nSelTot + popc64(selBits & lanemask_lt());
                    if (sel) endOf[rank + 1] = myEnd;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                    if (sel) emit_match(tileStart, ix, q, rank, myEnd);
                    nSelTot += popc64(selBits);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
                }
                if (lane == 0) L.wordRank[64] = nSelTot;
            }
        }
        ZMI_STAMP(5);
        if (any) lds_barrier();                // (uniform) a tile without a selection has nothing to hand over
        ZMI_STAMP(6);
        const u32 nSel = any ? L.wordRank[64] : 0u;
        // ---------------- literals of this tile: not covered by a selected match, not behind the entry cursor ----------------
        const u32 nSub = span >> kTileLog;          // 1, or up to 16 in a super-tile: the compaction below runs per 4096 positions
        // sixteen positions per thread: a tile takes 256 threads, four sub-tiles of a super-tile all 1024 (one pass per four)
        if (nSel == 0 && c0 == 0) {
            // nothing selected, nothing carried in: every byte of the tile (what there is of it) is a literal
            const u32 bytes = tileStart + span <= n ? span : n - tileStart;
            if (!deferred) {                     // (uniform) copied straight through
                for (u32 q16 = tid * 16; q16 < bytes; q16 += kTile * 16) {
                    const uint4 v = *reinterpret_cast<const uint4*>(L.in + tileStart + q16);
                    u8* o1 = litOut + litBase + q16;
                    if (q16 + 16 <= bytes) { u32u* o = (u32u*)o1; o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
                    else for (u32 k = 0; q16 + k < bytes; ++k) o1[k] = L.in[tileStart + q16 + k];
                }
            }
            litBase += bytes;
        } else {
            if (deferred) {         // (uniform) the first tile that is not all literals: the bytes counted so far, out of LDS
                for (u32 q16 = tid * 16; q16 < litBase; q16 += kTile * 16) {
                    const uint4 v = *reinterpret_cast<const uint4*>(L.in + hist + q16);
                    u32u* o = (u32u*)(litOut + q16);             // (litBase is a multiple of 4096 here: whole tiles only)
                    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
                }
                deferred = false;
            }
            const u32 lastEnd = nSel ? endOf[nSel] : cursor;
            // Every wave scans the group words itself (one LDS read per lane and sub-tile), so the compaction offsets need
            // neither a cross-wave table nor another barrier.  keepG(sub, g) = bytes of group g of sub-tile `sub` that are
            // literals.  Four sub-tiles of a super-tile are scanned side by side.
            auto keepG = [&](u32 sub, u32 g) -> u64 {
                const u32 g0 = sub * kTilePos + g * 64;             // tile-relative
                u64 k = ~cov[sub * kGroups + g];
                const u32 pG = tileStart + g0;
                if (pG >= n) k = 0; else if (n - pG < 64) k &= (1ull << (n - pG)) - 1;
                if (c0 > g0) k = (c0 - g0 >= 64) ? 0ull : (k & (~0ull << (c0 - g0)));      // before the entry cursor: inside an earlier match
                return k;
            };
            for (u32 s0 = 0; s0 < nSub; s0 += 4) {           // uniform
                u32 gcnt[4], gincl[4];
#pragma unroll
                for (u32 k = 0; k < 4; ++k) gcnt[k] = s0 + k < nSub ? popc64(keepG(s0 + k, lane)) : 0u;       // (uniform predicate)
#pragma unroll
                for (u32 k = 0; k < 4; ++k) gincl[k] = s0 + k < nSub ? wave_scan_incl(gcnt[k]) : 0u;
                const u32 tot0 = read_lane(gincl[0], 63), tot1 = read_lane(gincl[1], 63), tot2 = read_lane(gincl[2], 63), tot3 = read_lane(gincl[3], 63);
                const u32 q16 = s0 * kTilePos + tid * 16;
                if (q16 < span) {                            // whole waves: 256 threads per sub-tile
                    const u32 kSub = tid >> 8, gl = (tid >> 2) & 63u, sh = (tid & 3u) * 16u;
                    const u32 myExcl = kSub == 0 ? gincl[0] - gcnt[0] : kSub == 1 ? gincl[1] - gcnt[1] : kSub == 2 ? gincl[2] - gcnt[2] : gincl[3] - gcnt[3];
                    const u32 subBase = kSub == 0 ? 0u : kSub == 1 ? tot0 : kSub == 2 ? tot0 + tot1 : tot0 + tot1 + tot2;
                    const u32 gexcl = __shfl(myExcl, (int)gl);
                    const u64 kgw = keepG(s0 + kSub, gl);
                    const u32 keep16 = (u32)(kgw >> sh) & 0xFFFFu;
                    if (keep16) {
                        u8* o = litOut + litBase + subBase + gexcl + popc64(kgw & ((1ull << sh) - 1));
                        const uint4 v = *reinterpret_cast<const uint4*>(L.in + tileStart + q16);
                        if (keep16 == 0xFFFFu) { u32u* o4 = (u32u*)o; o4[0] = v.x; o4[1] = v.y; o4[2] = v.z; o4[3] = v.w; }
                        else {
                            const u32 d[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
                            for (u32 k = 0; k < 4; ++k) {
                                const u32 keep = (keep16 >> (4 * k)) & 0xFu, w4 = d[k];
                                if (keep == 0xFu) { *(u32u*)o = w4; o += 4; }
                                else { if (keep & 1) *o++ = (u8)w4; if (keep & 2) *o++ = (u8)(w4 >> 8); if (keep & 4) *o++ = (u8)(w4 >> 16); if (keep & 8) *o++ = (u8)(w4 >> 24); }
                            }
                        }
                    }
                }
                litBase += tot0 + tot1 + tot2 + tot3;
            }
            nbSeq += nSel; cursor = lastEnd;
        }
        t += nSub;
        ZMI_STAMP(7);
        // a chunk whose first tile is dense in matches (text, source code, structured data) is finished by lz_region_kernel: this
        // loop verifies every position to keep one in eight.  (A kernel of its own: with the region parse inlined here the tile
        // loop itself ran 8 % slower on sparse data — code size.)
        if (!FAR && regionList && tileStart == hist && !super && strideLog == 0 && matchCount >= kDenseMin && nTiles > (hist >> kTileLog) + 2) {      // uniform
            if constexpr ((MODE == 0 && !DICT) || MODE == 2) { regionCursor = cursor + 1; break; }     // (kSplit of launch_one)
            else {
                __syncthreads();
                dense_rest<MODE>(L, n, t, t, nTiles, hist, lowLimit, candAll + (u64)c * kChunkSize, nullptr, 0, seqOut, litOut, cursor, nbSeq, litBase, deferred, tid, lane, wave);
                break;
            }
        }
    }
#ifdef ZMI_LZ_STAMPS
    if ((tid & 63u) == 0) for (int i = 0; i < 14; i++) atomicAdd(&g_lzStamps[i], stampAcc[i]);
#endif
    if (tid == 0) {
        ChunkMeta m = {};
        m.srcSize = nData; m.nbSeq = nbSeq; m.litSize = litBase;
        if ((DICT || FAR) && frameBlocks) {   // only a frame's first block carries the frame header, sized for the whole frame's content
            const u64 fStart = base - (u64)bf * cb, fMax = (u64)frameBlocks * cb;
            const u64 fLen = (srcSize - fStart) < fMax ? (srcSize - fStart) : fMax;
            const u32 fcsField = fLen < 256 ? 0u : fLen < 65536 + 256 ? 2u : fLen <= 0xFFFFFFFFull ? 4u : 8u;       // (behind a window descriptor)
            m.fhSize = bf == 0 ? ((fhExtra >> 12) ? 6u + ((fhExtra & 0x100u) ? 0u : fcsField) : (fhExtra & 0x100u) ? 6u : frame_header_size64(fLen)) + (fhExtra & 7u) : 0u;
        } else m.fhSize = ((fhExtra & 0x100u) ? 6u : frame_header_size(nData)) + (fhExtra & 7u);      // fhExtra: bits 0-2 bytes of the dictID field (formatted dictionary); bit 8: window descriptor instead of a content size (magic, descriptor, window byte); bits 12-16: an explicit windowLog (multi-block frames only: descriptor AND content size)
        m.litFromSrc = deferred ? 1u : 0u;       // (then nbSeq = 0 and litBase = nData: the literals are the chunk itself)
        m.regionCursor = regionCursor;
        meta[c] = m;
        if (regionCursor) regionList[1 + atomicAdd(&regionList[0], 1u)] = c;       // work list of lz_region_kernel: [count, chunks...]
    }
    __syncthreads();                                       // the next chunk takes over LDS
    }
}

// The rest of a dense chunk (see dense_rest) as a kernel of its own — the fast finder on plain chunks (with the region parse
// inlined its tile loop ran 8 % slower on sparse data: code size) and the level >= 5 finder (whose hash-chain search wants the
// registers the tile loop's state would occupy: inlined, it spilled 288 bytes per lane): the chunk, and the history or dictionary
// tail in front of it, is staged again; the fast finder's table gets the first tile's positions (the latest occurrence per
// bucket, as the tile loop left it; the first-occurrence table starts empty: it only ever serves the tile that filled it); and
// the region parse takes the tiles after the first from the state lz_kernel recorded.
template <int MODE, bool DICT>
__global__ __launch_bounds__(1024) void lz_region_kernel(const u8* __restrict__ src, u64 srcSize, Seq* __restrict__ seqs, u8* __restrict__ lits,
                                                         ChunkMeta* __restrict__ meta, u16* __restrict__ candAll, u16* __restrict__ chainAll, const u32* __restrict__ regionList,
                                                         const u8* __restrict__ prefixArg, const u32 prefixLenArg, const u32 chunkBytes, const u32 frameBlocksArg, const u32 hcDepth)
{
    const u32 frameBlocks = frameBlocksArg & 0x7FFFFFFFu;
    const bool indep = (frameBlocksArg >> 31) != 0;
    extern __shared__ __attribute__((aligned(16))) u8 ldsRaw[];
    LzLds& L = *reinterpret_cast<LzLds*>(ldsRaw);
    const u32 tid = threadIdx.x, lane = lane_id(), wave = uniform(wave_id());
    // one workgroup per CU walks the work list lz_kernel filled (its order is whatever the atomics made it; chunks are independent)
    const u32 count = regionList[0];
    for (u32 li = blockIdx.x; li < count; li += gridDim.x) {
    const u32 c = regionList[1 + li];
    const u32 rc = meta[c].regionCursor;
    // geometry as in lz_kernel
    const u32 cb = DICT ? chunkBytes : kChunkSize;
    const u32 hist = DICT ? kChunkSize - ((chunkBytes + kTilePos - 1) & ~(kTilePos - 1)) : 0u;
    const u64 base = (u64)c * cb;
    const u8* __restrict__ in = src + base;
    const u32 bf = (DICT && frameBlocks) ? c % frameBlocks : 0u;
    u32 prefixLen = prefixLenArg; const u8* __restrict__ prefix = prefixArg;
    if (DICT && frameBlocks && !indep) { const u64 back = (u64)bf * cb; prefixLen = back < hist ? (u32)back : hist; prefix = in - prefixLen; }
    if (DICT && indep && bf) prefixLen = 0;
    const u32 lowLimit = DICT ? hist - prefixLen : 0u;
    const u32 nData = (u32)((srcSize - base) < cb ? (srcSize - base) : cb);
    const u32 n = hist + nData;
    {   // the image [lowLimit, n): history (or dictionary tail) + block.  Cross-chunk history is one contiguous piece of the input
        const bool onePiece = !DICT || (frameBlocks != 0 && !indep);
        const u8* __restrict__ from = onePiece ? in - prefixLen : in;
        const u32 at = onePiece ? lowLimit : hist, bytes = onePiece ? prefixLen + nData : nData;
        if ((((uintptr_t)from) | at) % 16 == 0) {
            uint4* l4 = reinterpret_cast<uint4*>(L.in + at);
            const u32 full = bytes >> 4;
            uint4 v0, v1, v2, v3;
            const u32 i0 = tid, i1 = tid + kTile, i2 = tid + 2 * kTile, i3 = tid + 3 * kTile;
            {   // (streaming loads: see the stores of dense_rest)
                typedef u32 __attribute__((ext_vector_type(4))) v4u;
                const v4u* n4 = reinterpret_cast<const v4u*>(from);
                const v4u a0 = __builtin_nontemporal_load(n4 + (i0 < full ? i0 : 0)), a1 = __builtin_nontemporal_load(n4 + (i1 < full ? i1 : 0));
                const v4u a2 = __builtin_nontemporal_load(n4 + (i2 < full ? i2 : 0)), a3 = __builtin_nontemporal_load(n4 + (i3 < full ? i3 : 0));
                v0 = uint4{a0.x, a0.y, a0.z, a0.w}; v1 = uint4{a1.x, a1.y, a1.z, a1.w}; v2 = uint4{a2.x, a2.y, a2.z, a2.w}; v3 = uint4{a3.x, a3.y, a3.z, a3.w};
            }
            if (i0 < full) l4[i0] = v0;
            if (i1 < full) l4[i1] = v1;
            if (i2 < full) l4[i2] = v2;
            if (i3 < full) l4[i3] = v3;
            for (u32 i = (full << 4) + tid; i < bytes; i += kTile) L.in[at + i] = from[i];
        } else {
            for (u32 i = tid; i < bytes; i += kTile) L.in[at + i] = from[i];
        }
        if (DICT && !onePiece) for (u32 i = lowLimit + tid; i < hist; i += kTile) L.in[i] = prefix[i - lowLimit];
        for (u32 i = tid; i < lowLimit; i += kTile) L.in[i] = 0;
    }
    for (u32 i = n + tid; i < kChunkSize + kInPad; i += kTile) L.in[i] = 0;
    {   // tables as lz_kernel starts them (the level >= 5 search lays its own over them)
        uint4* const t4 = reinterpret_cast<uint4*>(L.tabMem);
        const uint4 z = {0u, 0u, 0u, 0u}, f = {~0u, ~0u, ~0u, ~0u};
#pragma unroll
        for (u32 k = 0; k < 2; ++k) { t4[tid + k * kTile] = MODE == 0 ? z : f; t4[2 * kTile + tid + k * kTile] = MODE == 0 ? f : z; }
    }
    __syncthreads();
    u32 insertFrom = lowLimit >> kTileLog;
    if (MODE == 0 && !DICT) {                              // the first tile's positions into the table
        u32* const table = L.tabMem;
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            const u32 p = j * kTile + tid;
            if (p + 8 <= n) { const u32 hp = hash6p(lds_load8(L.in, p)); atomicMax(&table[hidx(hp)], ((p + 1) << 16) | htag(hp)); }
        }
        __syncthreads();
        insertFrom = 1;
    }
    const ChunkMeta m0 = meta[c];
    u32 cursor = rc - 1, nbSeq = m0.nbSeq, litBase = m0.litSize; bool deferred = m0.litFromSrc != 0;
    const u32 nTiles = (n + kTilePos - 1) / kTilePos;
    // (candidate and link planes belong to the WORKGROUP, not to the chunk: written and read back within a chunk's time, the same
    //  128 KiB per CU again and again stay in L2 / the memory-side cache instead of travelling to HBM and back)
    dense_rest<MODE>(L, n, insertFrom, (hist >> kTileLog) + 1, nTiles, hist, lowLimit, candAll + (u64)blockIdx.x * kChunkSize, chainAll ? chainAll + (u64)blockIdx.x * kChunkSize : nullptr, hcDepth,
                     seqs + (u64)c * kMaxSeq, lits + (u64)c * kLitStride, cursor, nbSeq, litBase, deferred, tid, lane, wave);
    if (tid == 0) { meta[c].nbSeq = nbSeq; meta[c].litSize = litBase; meta[c].litFromSrc = deferred ? 1u : 0u; }
    __syncthreads();                                       // the next chunk takes over LDS
    }
}

size_t lz_fast_lds_bytes() { return sizeof(LzLds); }

// How much is there for a match finder to find?  The input in groups of groupBytes; per group, tilesPerGroup tiles of 4 KiB spread
// evenly over it; per tile, two counts go into out[group]: (A) the positions whose 6 bytes' hash was seen earlier in the same tile
// with the same tag and the same first four bytes (what the fast finder's first-occurrence table would offer them), and (B) the
// positions of the 60 KiB in front of the tile — the reach of the history the levels >= 3 stage beside a block — that agree with
// a position of the tile on 8 bytes: redundancy that only shows at a distance (repeated incompressible records, duplicate pages)
// repeats nothing inside 4 KiB.  Text: several hundred per tile; Zipf or random bytes: a handful (an 8-byte agreement between
// independent Zipf(1.1) strings has probability 2e-10; the 6-byte hash with its tag lets a hundredth of a pair per tile through).
// (One workgroup per tile, reads 64 KiB: an eighth of the input at eight tiles per 4 MiB.)
constexpr u32 kProbeBack = 60u << 10;
// front: bytes of the input readable in front of src (a device worker's share of a call: the window in front of its first tiles)
__global__ __launch_bounds__(256) void lz_probe_kernel(const u8* __restrict__ src, u64 srcSize, u64 front, u64 groupBytes, u32 tilesPerGroup, u32* __restrict__ out)
{
    __shared__ u32 firstSeen[1u << kHashLog];
    const u32 tid = threadIdx.x;
    const u32 g = blockIdx.x / tilesPerGroup, k = blockIdx.x % tilesPerGroup;
    const u64 gStart = (u64)g * groupBytes, gLen = (srcSize - gStart) < groupBytes ? (srcSize - gStart) : groupBytes;
    const u64 off = gStart + (((gLen / tilesPerGroup) * k) & ~(u64)(kTilePos - 1));
    const u8* __restrict__ in = src + off;
    const u32 avail = (u32)((srcSize - off) < kTilePos + 8 ? (srcSize - off) : kTilePos + 8);      // bytes readable from `in`
    for (u32 i = tid; i < (1u << kHashLog); i += 256) firstSeen[i] = 0xFFFFFFFFu;
    __syncthreads();
    u32 h[16];
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
        const u32 q = j * 256 + tid;
        h[j] = 0;
        if (q + 8 <= avail) { h[j] = hash6p(readLE64(in + q)); atomicMin(&firstSeen[hidx(h[j])], (q << 16) | htag(h[j])); }
    }
    __syncthreads();
    u32 cnt = 0;
#pragma unroll
    for (u32 j = 0; j < 16; ++j) {
        const u32 q = j * 256 + tid;
        if (q + 8 <= avail) {
            const u32 f = firstSeen[hidx(h[j])], fq = f >> 16;
            if (fq < q && (f & 0xFFFFu) == htag(h[j]) && readLE32(in + fq) == readLE32(in + q)) ++cnt;
        }
    }
    // (B) the window in front of the tile against the tile's table
    const u32 back = avail < 16 ? 0u : off + front < kProbeBack ? (u32)(off + front) : kProbeBack;       // (the last read of the window reaches 7 bytes into the tile)
    const u8* __restrict__ win = in - back;
    for (u32 p = tid; p < back; p += 256) {
        const u64 w = readLE64(win + p);
        const u32 hp = hash6p(w);
        const u32 f = firstSeen[hidx(hp)];
        if ((f & 0xFFFFu) == htag(hp) && f != 0xFFFFFFFFu && readLE64(in + (f >> 16)) == w) ++cnt;
    }
    cnt = wave_sum(cnt);
    if (lane_id() == 0 && cnt) atomicAdd(&out[g], cnt);
}

void launch_lz_probe(const u8* src, u64 srcSize, u64 front, u64 groupBytes, u32 nGroups, u32 tilesPerGroup, u32* out, hipStream_t stream)
{
    (void)hipMemsetAsync(out, 0, (size_t)nGroups * sizeof(u32), stream);
    hipLaunchKernelGGL(lz_probe_kernel, dim3(nGroups * tilesPerGroup), dim3(256), 0, stream, src, srcSize, front, groupBytes, tilesPerGroup, out);
}

#ifdef ZMI_LZ_STAMPS
extern "C" void ZSTDMI_debugReadLzStamps(unsigned long long* out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lzStamps), 24 * sizeof(unsigned long long));      // (24 entries)
    if (reset) { unsigned long long z[24] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lzStamps), z, sizeof z); }
}
#endif

template <int MODE, int SHORT, bool DICT, bool FAR = false>
static void launch_one(const u8* src, u64 srcSize, u32 nChunks, Seq* seqs, u8* lits, ChunkMeta* meta, const u8* prefix, u32 prefixLen,
                       u32 chunkBytes, u32 fhExtra, u32 minStrideLog, u32 frameBlocks, u16* cand, u16* chain, u32* regionList, u32 hcDepth, hipStream_t stream, StageHook hook, u32* claimCtr)
{
    // the region parse of dense chunks: a second kernel behind a work list (see lz_region_kernel), inlined for the others
    constexpr bool kSplit = (MODE == 0 && !DICT && !FAR) || MODE == 2;
    // (the attribute is per device: a process may hold contexts on several GPUs)
    static bool attrSet[64] = {};
    int dev = 0; (void)hipGetDevice(&dev);
    if (!attrSet[dev & 63]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz_kernel<MODE, SHORT, DICT, FAR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzLds));
        if constexpr (kSplit) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lz_region_kernel<MODE, DICT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzLds));
        attrSet[dev & 63] = true;
    }
    if (FAR) cand = nullptr;
    if (cand && kSplit) (void)hipMemsetAsync(regionList, 0, sizeof(u32), stream);
    // the fast finder on plain chunks, more chunks than CUs: one workgroup per CU, chunks claimed from a counter (see lz_kernel)
    static u32 cuCount[64] = {};
    if (!cuCount[dev & 63]) { int n = 0; (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); cuCount[dev & 63] = n > 0 ? (u32)n : 256u; }
    const bool claim = MODE == 0 && !DICT && !FAR && claimCtr && nChunks > cuCount[dev & 63];
    if (claim) (void)hipMemsetAsync(claimCtr, 0, sizeof(u32), stream);
    const u32 grid = claim ? cuCount[dev & 63] : nChunks;
    hipLaunchKernelGGL((lz_kernel<MODE, SHORT, DICT, FAR>), dim3(grid), dim3(kTile), sizeof(LzLds), stream, src, srcSize, seqs, lits, meta, prefix, prefixLen, chunkBytes, fhExtra, minStrideLog, frameBlocks, cand, cand ? chain : nullptr, cand ? regionList : nullptr, nChunks,
                       claim ? claimCtr : nullptr);
    hook("lz_fast");
    if constexpr (kSplit) if (cand) {                      // the dense chunks' rest: 256 workgroups (one per CU) walk the list
        hipLaunchKernelGGL((lz_region_kernel<MODE, DICT>), dim3(nChunks < 256 ? nChunks : 256), dim3(kTile), sizeof(LzLds), stream, src, srcSize, seqs, lits, meta, cand, chain, regionList,
                           prefix, prefixLen, chunkBytes, frameBlocks, hcDepth);
        hook("lz_region");
    }
}

// finder: 0 = fast, 1 = dual (8-byte + 5-byte hashes), 2 = dual + lazy deferral.  (A 4-byte short hash, the reference's
// minMatch at levels 4+, was measured and lost ratio on every corpus tried: far 4-byte matches cost more than literals.)
// prefix/prefixLen: the dictionary bytes every chunk sees as history (null/0 without one); chunkBytes = 64 KiB minus prefixLen
// rounded up to whole 4 KiB tiles.  frameBlocks > 0: cross-chunk history instead (no dictionary): `frameBlocks` chunks of chunkBytes
// form one frame and each sees up to 64 KiB - chunkBytes of the input in front of it.  chunkBytes < 64 KiB with neither: independent
// frames of chunkBytes each (ZSTD_c_windowLog 10 .. 15: a frame is its own window), on the same instantiation with an empty history.
// cand / regionList (null: off): workspace of the region parse, 65536 u16 per chunk and 1 + nChunks u32.
void launch_lz(u32 finder, const u8* src, u64 srcSize, u32 nChunks, Seq* seqs, u8* lits, ChunkMeta* meta, const u8* prefix, u32 prefixLen,
               u32 chunkBytes, u32 fhExtra, u32 minStrideLog, u32 frameBlocks, u16* cand, u16* chain, u32* regionList, u32 hcDepth, hipStream_t stream, StageHook hook, u32* claimCtr)
{
    if (chunkBytes >= kChunkSize && frameBlocks && finder == 0) {       // fast strategy with cross-chunk history: full 64 KiB blocks, far candidates
        launch_one<0, 5, false, true>(src, srcSize, nChunks, seqs, lits, meta, nullptr, 0, kChunkSize, fhExtra, minStrideLog, frameBlocks, nullptr, nullptr, nullptr, 0, stream, hook, claimCtr);
        return;
    }
    if (chunkBytes >= kChunkSize) {
        switch (finder) {
        case 0:  launch_one<0, 5, false>(src, srcSize, nChunks, seqs, lits, meta, nullptr, 0, kChunkSize, fhExtra, minStrideLog, 0, cand, chain, regionList, hcDepth, stream, hook, claimCtr); break;
        case 1:  launch_one<1, 5, false>(src, srcSize, nChunks, seqs, lits, meta, nullptr, 0, kChunkSize, fhExtra, minStrideLog, 0, cand, chain, regionList, hcDepth, stream, hook, claimCtr); break;
        default: launch_one<2, 5, false>(src, srcSize, nChunks, seqs, lits, meta, nullptr, 0, kChunkSize, fhExtra, minStrideLog, 0, cand, chain, regionList, hcDepth, stream, hook, claimCtr); break;
        }
        return;
    }
    switch (finder) {
    case 0:  launch_one<0, 5, true>(src, srcSize, nChunks, seqs, lits, meta, prefix, prefixLen, chunkBytes, fhExtra, minStrideLog, frameBlocks, cand, chain, regionList, hcDepth, stream, hook, claimCtr); break;
    case 1:  launch_one<1, 5, true>(src, srcSize, nChunks, seqs, lits, meta, prefix, prefixLen, chunkBytes, fhExtra, minStrideLog, frameBlocks, cand, chain, regionList, hcDepth, stream, hook, claimCtr); break;
    default: launch_one<2, 5, true>(src, srcSize, nChunks, seqs, lits, meta, prefix, prefixLen, chunkBytes, fhExtra, minStrideLog, frameBlocks, cand, chain, regionList, hcDepth, stream, hook, claimCtr); break;
    }
}

} // namespace zmi
