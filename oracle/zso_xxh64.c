/*
 * oracle/zso_xxh64.c — XXH64, restating U/Xxhash.cs:378-600 (one-shot form).
 * TEST INFRASTRUCTURE ONLY (see zso_common.h).
 */
#include "zso_common.h"

#define P1 0x9E3779B185EBCA87ULL
#define P2 0xC2B2AE3D27D4EB4FULL
#define P3 0x165667B19E3779F9ULL
#define P4 0x85EBCA77C2B2AE63ULL
#define P5 0x27D4EB2F165667C5ULL

static inline u64 rotl(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
static inline u64 xround(u64 acc, u64 in) { acc += in * P2; acc = rotl(acc, 31); return acc * P1; }
static inline u64 xmerge(u64 acc, u64 v) { acc ^= xround(0, v); return acc * P1 + P4; }

u64 zso_xxh64(const void* data, size_t len, u64 seed)
{
    const u8* p = (const u8*)data; const u8* const end = p + len; u64 h;
    if (len >= 32) {
        const u8* const limit = end - 32;
        u64 v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
        do {
            v1 = xround(v1, zso_readLE64(p)); v2 = xround(v2, zso_readLE64(p + 8));
            v3 = xround(v3, zso_readLE64(p + 16)); v4 = xround(v4, zso_readLE64(p + 24));
            p += 32;
        } while (p <= limit);
        h = rotl(v1, 1) + rotl(v2, 7) + rotl(v3, 12) + rotl(v4, 18);
        h = xmerge(h, v1); h = xmerge(h, v2); h = xmerge(h, v3); h = xmerge(h, v4);
    } else h = seed + P5;
    h += (u64)len;
    while (p + 8 <= end) { h ^= xround(0, zso_readLE64(p)); h = rotl(h, 27) * P1 + P4; p += 8; }
    if (p + 4 <= end) { h ^= (u64)zso_readLE32(p) * P1; h = rotl(h, 23) * P2 + P3; p += 4; }
    while (p < end) { h ^= (*p) * P5; h = rotl(h, 11) * P1; p++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    return h;
}
