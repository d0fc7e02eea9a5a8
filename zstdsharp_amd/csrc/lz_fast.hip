// lz_fast.hip — level-1 ("fast" strategy) LZ match finder for gfx950: one 1024-thread workgroup per 64 KiB chunk.
//
// Takes the place of ZSTD_compressBlock_fast_noDict_generic (U/ZstdFast.cs:96-288) + ZSTD_storeSeq
// (U/ZstdCompressInternal.cs:204-246) for one block.  It is NOT that function's parse: the reference walks
// the block serially, inserting only the positions it visits; here every position of a 1024-byte tile is
// hashed (hash6, U/ZstdCompressInternal.cs:380-392), probed and verified by its own lane against the
// LDS-resident table of all earlier tiles, and wave 0 then performs the greedy left-to-right selection over
// the tile with ballots.  The produced sequences are therefore valid but not identical to the reference's
// (SURVEY.md §7 "valid zstd frames, not byte-identical frames"); repcode assignment follows the decoder's
// history rule (U/ZstdDecompressBlock.cs:2387-2443) so every emitted offBase decodes to the intended offset.
//
// Candidates inside the same tile are found through a second table that keeps, per hash, the FIRST position of
// the current tile (atomicMin with a tile stamp, so it needs no clearing and is order-independent): a lane whose
// hash was already seen in its own tile matches against that nearer occurrence, which is what catches runs and
// short-period data.  Both tables are updated with commutative atomics only, so the output is deterministic.
//
// LDS (one workgroup per CU): chunk bytes 64 KiB (+pad) | 2 hash tables u32[8192] 64 KiB | tile arrays 3 KiB |
// coverage bitmask 8 KiB.  HBM traffic per chunk: read n, write literals (<= n) + 8 B per sequence.
#include "zmi_device.h"

namespace zmi {

constexpr u32 kHashLog  = 13;            // the reference's hashLog for level 1 at <= 128 KiB (U/Clevels.cs:488)
constexpr u32 kTile     = 1024;          // threads per workgroup
constexpr u32 kPPT      = 4;             // positions per thread per tile (independent probes in flight per lane)
constexpr u32 kTilePos  = kTile * kPPT;  // positions per tile
constexpr u32 kTileLog  = 12;            // log2(kTilePos)
static_assert((1u << kTileLog) == kTilePos, "tile geometry");
constexpr u32 kLenCap   = 32;            // per-lane forward extension cap; the selecting wave extends the rest
constexpr u32 kInPad    = 64;

struct LzLds {
    u8  in[kChunkSize + kInPad];
    u32 table[1u << kHashLog];           // position+1 of the latest occurrence of the hash in EARLIER tiles; 0 = empty
    u32 first[1u << kHashLog];           // first occurrence of the hash inside the CURRENT tile: ((15-tile) << 12) | index
    u16 tileOff[kTilePos];
    u8  tileLen[kTilePos];
    u32 cov[kChunkSize / 32];            // bit p set <=> byte p is covered by a selected match
    u64 tileMask[kTilePos / 64];         // per 64-position group of the current tile: lanes that hold a match
    u32 waveCnt[2][16];
    u32 nbSeq, anchorEnd;
};

__device__ __forceinline__ u32 hash6(u64 w) { return (u32)(((w << 16) * 227718039650203ULL) >> (64 - kHashLog)); }

struct Walk { u32 cur, anchor, nbSeq, rep0, rep1, rep2; };

// Greedy selection over one tile, executed by wave 0 with all 64 lanes (control flow is wave-uniform).
__device__ __forceinline__ void walk_tile(LzLds& L, u32 n, u32 tileStart, Seq* __restrict__ seqOut, Walk& st)
{
    const u32 lane = lane_id();
    // groups of this tile that hold at least one match: the walker only visits those
    u64 groups = ballot(lane < kTilePos / 64 && L.tileMask[lane % (kTilePos / 64)] != 0);
    while (groups) {
        const u32 g = ctz64(groups);
        groups &= groups - 1;
        const u32 gbase = tileStart + g * 64;
        if (st.cur >= gbase + 64) continue;
        const u32 myLen = L.tileLen[g * 64 + lane];
        const u32 myOff = L.tileOff[g * 64 + lane];
        u64 mask = ballot(myLen != 0);
        if (st.cur > gbase) mask &= ~0ull << (st.cur - gbase);
        while (mask) {
            const u32 f = ctz64(mask);
            u32 pos = gbase + f;
            u32 len = read_lane(myLen, f);
            const u32 off = read_lane(myOff, f);
            // forward: a capped lane length means "at least kLenCap": finish it with 64 lanes x 8 bytes per step
            if (len == kLenCap) {
                u32 e = pos + len;
                for (;;) {
                    const u32 q = e + 8 * lane;                       // reads past n land in the table region: harmless, clamped below
                    const u64 x = readLE64(L.in + q) ^ readLE64(L.in + q - off);
                    const u64 bad = ballot(x != 0 || q + 8 > n);
                    if (bad == 0) { e += 512; continue; }
                    const u32 fl = ctz64(bad);
                    const u32 cnt = x ? (ctz64(x) >> 3) : 8;
                    e += 8 * fl + read_lane(cnt, fl);
                    break;
                }
                if (e > n) e = n;
                len = e - pos;
            }
            // backward: give bytes of the pending literal run to the match while they agree (ZstdFast.cs:242-247)
            for (;;) {
                u32 maxBack = pos - st.anchor;
                const u32 cpos = pos - off;
                if (cpos < maxBack) maxBack = cpos;
                if (maxBack > 64) maxBack = 64;
                const bool eq = lane < maxBack && L.in[pos - 1 - lane] == L.in[cpos - 1 - lane];
                const u64 b = ballot(eq);
                const u32 back = (~b == 0) ? 64 : ctz64(~b);
                pos -= back; len += back;
                if (back < 64) break;
            }
            if (st.nbSeq < kMaxSeq) {
                const u32 litLen = pos - st.anchor;
                const u32 ll0 = litLen == 0;
                u32 code;
                if (!ll0) code = off == st.rep0 ? 1 : off == st.rep1 ? 2 : off == st.rep2 ? 3 : off + 3;
                else      code = off == st.rep1 ? 1 : off == st.rep2 ? 2 : (off == st.rep0 - 1 && st.rep0 > 1) ? 3 : off + 3;
                if (code > 3) { st.rep2 = st.rep1; st.rep1 = st.rep0; st.rep0 = off; }
                else {
                    const u32 idx = code - 1 + ll0;
                    if (idx == 1) { const u32 t = st.rep1; st.rep1 = st.rep0; st.rep0 = t; }
                    else if (idx == 2) { const u32 t = st.rep2; st.rep2 = st.rep1; st.rep1 = st.rep0; st.rep0 = t; }
                    else if (idx == 3) { const u32 t = st.rep0 - 1; st.rep2 = st.rep1; st.rep1 = st.rep0; st.rep0 = t; }
                }
                if (lane == 0) { Seq s; s.offBase = code; s.litLength = (u16)litLen; s.mlBase = (u16)(len - 3); seqOut[st.nbSeq] = s; }
                st.nbSeq++;
                // mark [pos, pos+len) covered
                {
                    const u32 last = pos + len - 1, w0 = pos >> 5, w1 = last >> 5;
                    for (u32 w = w0 + lane; w <= w1; w += 64) {
                        u32 m = ~0u;
                        if (w == w0) m &= ~0u << (pos & 31);
                        if (w == w1) m &= ~0u >> (31 - (last & 31));
                        L.cov[w] |= m;
                    }
                }
                st.anchor = st.cur = pos + len;
            } else {
                st.cur = pos + 1;          // sequence budget exhausted: the rest of the chunk stays literal
                mask = 0;
                break;
            }
            mask = (st.cur - gbase < 64) ? (mask & (~0ull << (st.cur - gbase))) : 0;
        }
    }
}

__global__ __launch_bounds__(1024) void lz_fast_kernel(const u8* __restrict__ src, u64 srcSize,
                                                       Seq* __restrict__ seqs, u8* __restrict__ lits,
                                                       ChunkMeta* __restrict__ meta)
{
    extern __shared__ __attribute__((aligned(16))) u8 ldsRaw[];
    LzLds& L = *reinterpret_cast<LzLds*>(ldsRaw);
    const u32 c = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    const u64 base = (u64)c << kChunkLog;
    const u32 n = (u32)((srcSize - base) < kChunkSize ? (srcSize - base) : kChunkSize);
    const u8* __restrict__ in = src + base;

    // ---- stage the chunk: 16 B per lane when the source is 16-byte aligned ----
    if ((((uintptr_t)in) & 15) == 0) {
        const uint4* in4 = reinterpret_cast<const uint4*>(in);
        uint4* l4 = reinterpret_cast<uint4*>(L.in);
        const u32 full = n >> 4;
        for (u32 i = tid; i < full; i += kTile) l4[i] = in4[i];
        for (u32 i = (full << 4) + tid; i < n; i += kTile) L.in[i] = in[i];
    } else {
        for (u32 i = tid; i < n; i += kTile) L.in[i] = in[i];
    }
    for (u32 i = n + tid; i < kChunkSize + kInPad; i += kTile) L.in[i] = 0;
    for (u32 i = tid; i < (1u << kHashLog); i += kTile) { L.table[i] = 0; L.first[i] = 0xFFFFFFFFu; }
    for (u32 i = tid; i < kChunkSize / 32; i += kTile) L.cov[i] = 0;
    __syncthreads();

    Walk st; st.cur = 0; st.anchor = 0; st.nbSeq = 0; st.rep0 = 1; st.rep1 = 4; st.rep2 = 8;
    Seq* __restrict__ seqOut = seqs + (u64)c * kMaxSeq;

    // Matches may start where 8 bytes are still readable (the reference stops at iend-8, ZstdFast.cs:110).
    const u32 nTiles = (n + kTilePos - 1) / kTilePos;
    for (u32 t = 0; t < nTiles; ++t) {
        const u32 stamp = ((kChunkSize / kTilePos - 1) - t) << kTileLog;
        u64 w[kPPT]; u32 h[kPPT], cand[kPPT]; bool valid[kPPT];
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            const u32 q = j * kTile + tid, p = t * kTilePos + q;
            valid[j] = p + 8 <= n; w[j] = 0; h[j] = 0; cand[j] = 0;
            if (valid[j]) { w[j] = readLE64(L.in + p); h[j] = hash6(w[j]); cand[j] = L.table[h[j]]; atomicMin(&L.first[h[j]], stamp | q); }
        }
        __syncthreads();                       // every probe of this tile precedes every insert of this tile
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            const u32 q = j * kTile + tid, p = t * kTilePos + q;
            if (valid[j]) {
                atomicMax(&L.table[h[j]], p + 1);
                const u32 f = L.first[h[j]];   // same-tile first occurrence: nearer than anything in the cross-tile table
                if ((f >> kTileLog) == (stamp >> kTileLog) && (f & (kTilePos - 1)) < q) cand[j] = t * kTilePos + (f & (kTilePos - 1)) + 1;
            }
        }
#pragma unroll
        for (u32 j = 0; j < kPPT; ++j) {
            const u32 q = j * kTile + tid, p = t * kTilePos + q;
            u32 len = 0, off = 0;
            if (cand[j]) {
                const u32 cpos = cand[j] - 1;
                u64 x = w[j] ^ readLE64(L.in + cpos);
                if ((u32)x == 0) {             // >= 4 equal bytes, as the reference's MEM_read32 check (ZstdFast.cs:179-191)
                    u32 l = x ? (ctz64(x) >> 3) : 8;
                    if (!x) {
                        while (l < kLenCap) {
                            x = readLE64(L.in + p + l) ^ readLE64(L.in + cpos + l);
                            if (x) { l += ctz64(x) >> 3; break; }
                            l += 8;
                        }
                    }
                    if (l > n - p) l = n - p;
                    if (l > kLenCap) l = kLenCap;
                    if (l >= 4) { len = l; off = p - cpos; }
                }
            }
            L.tileLen[q] = (u8)len; L.tileOff[q] = (u16)off;
            { const u64 mm = ballot(len != 0); if (lane == 0) L.tileMask[j * (kTile / 64) + wave] = mm; }
        }
        __syncthreads();                       // tile arrays and inserts visible
        if (wave == 0) walk_tile(L, n, t * kTilePos, seqOut, st);
        // the other 15 waves run ahead into the next tile's probes; they meet wave 0 at that tile's first barrier
    }
    if (tid == 0) { L.nbSeq = st.nbSeq; }
    __syncthreads();

    // ---- literals: every byte not covered by a selected match, in order; 16 positions per thread per round ----
    u8* __restrict__ litOut = lits + ((u64)c << kChunkLog);
    u32 litBase = 0;
    const u32 nRounds = (n + 16 * kTile - 1) / (16 * kTile);
    for (u32 r = 0; r < nRounds; ++r) {
        const u32 p = (r * kTile + tid) * 16;
        u32 keep = 0;                                         // bit k set <=> byte p+k is a literal
        if (p < n) {
            const u32 covw = L.cov[p >> 5] >> (p & 31);       // p is a multiple of 16: the 16 bits sit in one word
            keep = ~covw & 0xFFFFu;
            if (n - p < 16) keep &= (1u << (n - p)) - 1;
        }
        const u32 cnt = __builtin_popcount(keep);
        const u32 incl = wave_scan_incl(cnt);
        if (lane == 63) L.waveCnt[r & 1][wave] = incl;
        __syncthreads();
        u32 before = 0, total = 0;
#pragma unroll
        for (u32 k = 0; k < 16; ++k) { const u32 v = L.waveCnt[r & 1][k]; total += v; if (k < wave) before += v; }
        u8* o = litOut + litBase + before + incl - cnt;
        if (keep == 0xFFFFu) {
            const uint4 v = *reinterpret_cast<const uint4*>(L.in + p);
            *(u32u*)(o) = v.x; *(u32u*)(o + 4) = v.y; *(u32u*)(o + 8) = v.z; *(u32u*)(o + 12) = v.w;
        } else {
            while (keep) { const u32 k = __builtin_ctz(keep); keep &= keep - 1; *o++ = L.in[p + k]; }
        }
        litBase += total;
    }
    if (tid == 0) {
        ChunkMeta m = {};
        m.srcSize = n; m.nbSeq = L.nbSeq; m.litSize = litBase; m.fhSize = frame_header_size(n);
        meta[c] = m;
    }
}

size_t lz_fast_lds_bytes() { return sizeof(LzLds); }

void launch_lz_fast(const u8* src, u64 srcSize, u32 nChunks, Seq* seqs, u8* lits, ChunkMeta* meta, hipStream_t stream)
{
    static bool attrSet = false;
    if (!attrSet) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(lz_fast_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LzLds));
        attrSet = true;
    }
    hipLaunchKernelGGL(lz_fast_kernel, dim3(nChunks), dim3(kTile), sizeof(LzLds), stream, src, srcSize, seqs, lits, meta);
}

} // namespace zmi
