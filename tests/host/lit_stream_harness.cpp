#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
typedef uint8_t u8; typedef uint16_t u16; typedef uint32_t u32; typedef uint64_t u64; typedef int32_t s32;
typedef u64 __attribute__((aligned(1))) u64u; typedef u32 __attribute__((aligned(1))) u32u; typedef u16 __attribute__((aligned(1))) u16u;
static inline u32 highbit32(u32 v) { return 31u - (u32)__builtin_clz(v); }
static inline u64 readLE64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
static inline u32 __builtin_amdgcn_alignbit(u32 hi, u32 lo, u32 s) { s &= 31; return (u32)((((u64)hi << 32) | lo) >> s); }
static inline int __builtin_amdgcn_mov_dpp(int v, int, int, int, bool) { return v; }     // quad-transposed stores: GPU only, never taken here
#define ZMI_LIT_EXPERIMENT 0
static u8 g_lds[1 << 16];                      // stands in for the LDS: tables are addressed by offset
#define ZMI_LDS_U16(off) (*(const u16*)(g_lds + (off)))
static inline u32 huf_entry(u32 sym, u32 nbBits) { return (32u - nbBits) | (sym << 8); }
#include "bb.inc"
#include "fs.inc"
// build a canonical zstd Huffman code from weights (HUF_readDTableX1 order), tables for IDX bits, encode symbols backward
struct Code { u32 nb[256]; u32 val[256]; u32 tableLog; };
int main(int argc, char** argv) {
    srand(7);
    int bad = 0;
    for (int iter = 0; iter < 30000; ++iter) {
        // random weights with sum of 2^(w-1) a power of two
        u32 tableLog = 5 + rand() % 7;         // 5..11
        u32 nsym = 2 + rand() % 200;
        std::vector<u32> w(256, 0);
        // start with all symbols weight... build by splitting: simple approach: assign lengths via random tree splits
        std::vector<u32> lens; lens.push_back(0);
        while (lens.size() < nsym) { size_t k = rand() % lens.size(); if (lens[k] >= tableLog) { bool any=false; for (auto L: lens) if (L<tableLog) any=true; if(!any) break; continue; } u32 L = lens[k] + 1; lens[k] = L; lens.push_back(L); }
        u32 maxL = 0; for (auto L : lens) maxL = std::max(maxL, L);
        tableLog = maxL; if (tableLog < 1) continue;
        nsym = lens.size();
        std::vector<u32> perm(256); for (int i = 0; i < 256; i++) perm[i] = i; std::random_shuffle(perm.begin(), perm.end());
        Code c; memset(&c, 0, sizeof c); c.tableLog = tableLog;
        for (u32 i = 0; i < nsym; i++) { c.nb[perm[i]] = lens[i]; w[perm[i]] = tableLog + 1 - lens[i]; }
        // table in tableLog bits: classes by weight ascending, symbols ascending
        std::vector<u16> full(1u << tableLog); std::vector<u8> sorted; std::vector<u32> startOf(256);
        u32 idx = 0;
        for (u32 ww = 1; ww <= tableLog; ww++) for (u32 sI = 0; sI < 256; sI++) if (w[sI] == ww) { startOf[sI] = idx; u32 len = 1u << (ww - 1); for (u32 u = 0; u < len; u++) full[idx + u] = (u16)sI; idx += len; sorted.push_back((u8)sI); }
        if (idx != (1u << tableLog)) { printf("bad tree\n"); return 1; }
        for (u32 sI = 0; sI < 256; sI++) if (w[sI]) c.val[sI] = startOf[sI] >> (w[sI] - 1);   // code value = top nb bits of index
        sorted.resize(256);
        // random symbols
        u32 n = (iter & 1) ? 1 + rand() % 200 : 1 + rand() % 5000;
        std::vector<u8> syms(n); std::vector<u32> present; for (u32 sI = 0; sI < 256; sI++) if (w[sI]) present.push_back(sI);
        for (u32 i = 0; i < n; i++) syms[i] = (u8)present[rand() % present.size()];
        // encode: symbols last to first into an LSB-first stream; decoder reads from the top: first decoded = last written
        std::vector<u8> out((size_t)n * 2 + 16, 0); u64 bitpos = 0;
        auto put = [&](u32 v, u32 nb) { for (u32 b = 0; b < nb; b++) { if ((v >> b) & 1) out[(bitpos) >> 3] |= 1 << (bitpos & 7); bitpos++; } };
        for (int i = (int)n - 1; i >= 0; i--) put(c.val[syms[i]], c.nb[syms[i]]);
        put(1, 1);
        u32 sz = (u32)((bitpos + 7) >> 3);
        std::vector<u8> stream(out.begin(), out.begin() + sz);
        // decoder tables
        for (int form = 0; form < 2; form++) {
            std::vector<u16> tab; bool pairs = false; u32 IDX = form == 0 ? 11 : 10;
            if (tableLog > IDX + (form == 1 ? 1 : 0)) continue;
            tab.assign(1u << IDX, 0);
            const u32 tOff = 8192;                                   // aligned to the table size, like the kernels' LDS arrays
            if (tableLog <= IDX) { u32 up = IDX - tableLog; for (u32 i = 0; i < (1u << tableLog); i++) for (u32 u = 0; u < (1u << up); u++) tab[(i << up) + u] = (u16)huf_entry(full[i], c.nb[full[i]]); }
            u32 n1 = 0;
            if (tableLog > IDX) { pairs = true; for (u32 sI = 0; sI < 256; sI++) n1 += w[sI] == 1; for (u32 i = 0; i < (1u << tableLog); i += 2) { if (i < n1) tab[i >> 1] = (u16)(full[i] | (full[i + 1] << 8)); else tab[i >> 1] = (u16)huf_entry(full[i], c.nb[full[i]]); } }
            std::vector<u8> buf(32 + sz + 32, 0xAA); memcpy(buf.data() + 32, stream.data(), sz);      // guard bytes around the stream
            std::vector<u8> dec(n + 64, 0);
            bool ok;
            memcpy(g_lds + tOff, tab.data(), tab.size() * 2);
            if (form == 0) ok = huf_decode_stream_fs<11, false>(tOff, 0u, buf.data() + 32, sz, dec.data(), n);
            else ok = pairs ? huf_decode_stream_fs<10, true>(tOff, n1, buf.data() + 32, sz, dec.data(), n)
                            : huf_decode_stream_fs<10, false>(tOff, 0u, buf.data() + 32, sz, dec.data(), n);
            if (!ok || memcmp(dec.data(), syms.data(), n)) { u32 k = 0; while (k < n && dec[k] == syms[k]) k++; printf("iter %d form %d tableLog %u n %u sz %u ok %d firstbad %u\n", iter, form, tableLog, n, sz, ok, k); if (++bad > 10) return 1; }
        }
    }
    printf("done bad=%d\n", bad);
}
