// frame.hip — output assembly on gfx950: per-chunk frame sizes -> exclusive scan -> gather into one contiguous
// stream (the "variable-length output" step of SURVEY.md §7), plus the optional XXH64 content checksum
// (U/Xxhash.cs:378-600, written by ZSTD_writeEpilogue U/ZstdCompress.cs:5641-5652).
#include "zmi_device.h"

namespace zmi {

// ---- exclusive scan of ChunkMeta::outSize (one workgroup; nChunks is at most a few hundred thousand) ----
__global__ __launch_bounds__(1024) void scan_sizes_kernel(const ChunkMeta* __restrict__ meta, u32 nChunks,
                                                          u64* __restrict__ offsets, u64* __restrict__ total)
{
    __shared__ u32 waveSum[16];
    const u32 tid = threadIdx.x, lane = lane_id(), wave = wave_id();
    u64 carry = 0;
    for (u32 base = 0; base < nChunks; base += 1024) {
        const u32 i = base + tid;
        const u32 v = i < nChunks ? meta[i].outSize : 0;
        const u32 incl = wave_scan_incl(v);
        if (lane == 63) waveSum[wave] = incl;
        __syncthreads();
        u32 before = 0, all = 0;
#pragma unroll
        for (u32 k = 0; k < 16; k++) { const u32 s = waveSum[k]; all += s; if (k < wave) before += s; }
        if (i < nChunks) offsets[i] = carry + before + incl - v;
        carry += all;
        __syncthreads();
    }
    if (tid == 0) *total = carry;
}

// ---- gather: copy each chunk's frame from its slot (or, for stored blocks, header + source bytes) to dst ----
__device__ __forceinline__ void copy_bytes(u8* __restrict__ d, const u8* __restrict__ s, u32 n, u32 tid, u32 nthreads)
{
    // head: bring d to 16-byte alignment, then 16 B stores fed by unaligned loads
    u32 head = (u32)((16 - ((uintptr_t)d & 15)) & 15);
    if (head > n) head = n;
    if (tid < head) d[tid] = s[tid];
    const u32 body = (n - head) >> 4;
    uint4* d4 = reinterpret_cast<uint4*>(d + head);
    const u8* sb = s + head;
    for (u32 i = tid; i < body; i += nthreads) {
        uint4 v;
        v.x = readLE32(sb + 16 * i); v.y = readLE32(sb + 16 * i + 4); v.z = readLE32(sb + 16 * i + 8); v.w = readLE32(sb + 16 * i + 12);
        d4[i] = v;
    }
    const u32 done = head + (body << 4);
    if (tid < n - done) d[done + tid] = s[done + tid];
}

__global__ __launch_bounds__(256) void gather_kernel(const u8* __restrict__ src, u64 srcSize, const u8* __restrict__ slots,
                                                     const ChunkMeta* __restrict__ meta, const u64* __restrict__ offsets,
                                                     u8* __restrict__ dst, u64 dstCapacity, u32 chunkBytes)
{
    const u32 c = blockIdx.x, tid = threadIdx.x;
    const ChunkMeta m = meta_checked(meta[c]);
    const u64 off = offsets[c];
    if (off + m.outSize > dstCapacity) return;           // host reports dstSize_tooSmall from the scanned total
    const u8* slot = slots + (u64)c * kSlotStride;
    u8* d = dst + off;
    const u32 tail = m.outSize - (m.fhSize + 3) - (m.blockType == 2 ? m.bodySize : m.srcSize);   // 0 or 4 (checksum)
    if (m.blockType == 2) {
        // the literals section is already in place (huf_encode writes it there); headers and the sequences section follow
        const u32 head = m.fhSize + 3, seqAt = head + m.litSectionSize;
        if (tid < head) d[tid] = slot[tid];
        copy_bytes(d + seqAt, slot + seqAt, head + m.bodySize - seqAt, tid, 256);
    } else {
        if (tid < m.fhSize + 3) d[tid] = slot[tid];
        copy_bytes(d + m.fhSize + 3, src + (u64)c * chunkBytes, m.srcSize, tid, 256);
    }
    if (tail && tid < 4) d[m.outSize - 4 + tid] = (u8)(m.checksum >> (8 * tid));
}

// ---- XXH64 (seed 0): 4 lanes per chunk, one per accumulator ----
constexpr u64 P1 = 0x9E3779B185EBCA87ULL, P2 = 0xC2B2AE3D27D4EB4FULL, P3 = 0x165667B19E3779F9ULL,
              P4 = 0x85EBCA77C2B2AE63ULL, P5 = 0x27D4EB2F165667C5ULL;
__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 xxh_round(u64 acc, u64 in) { acc += in * P2; acc = rotl64(acc, 31); return acc * P1; }
__device__ __forceinline__ u64 xxh_merge(u64 acc, u64 v) { acc ^= xxh_round(0, v); return acc * P1 + P4; }

// (one frame = frameBlocks chunks of chunkBytes; the checksum is filed with the frame's last block, which carries it)
__global__ __launch_bounds__(256) void xxh64_kernel(const u8* __restrict__ src, u64 srcSize, ChunkMeta* __restrict__ meta, u32 nChunks, u32 chunkBytes,
                                                    u32 frameBlocks)
{
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    const u32 f = t >> 2, j = t & 3;
    if ((u64)f * frameBlocks >= nChunks) return;           // whole groups of 4 lanes leave together
    const u64 frameBytes = (u64)frameBlocks * chunkBytes;
    const u64 base = (u64)f * frameBytes;
    const u32 n = (u32)((srcSize - base) < frameBytes ? (srcSize - base) : frameBytes);
    const u32 c = (f + 1) * frameBlocks <= nChunks ? (f + 1) * frameBlocks - 1 : nChunks - 1;
    const u8* p = src + base;
    u64 h;
    const u32 stripes = n >> 5;
    u64 v = j == 0 ? P1 + P2 : j == 1 ? P2 : j == 2 ? 0 : 0 - P1;
    for (u32 i = 0; i < stripes; i++) v = xxh_round(v, readLE64(p + 32 * i + 8 * j));
    const u32 l0 = threadIdx.x & 60;    // first lane of this group within the wave (groups never straddle waves)
    const u64 v1 = __shfl(v, (l0 & 63) + 0), v2 = __shfl(v, (l0 & 63) + 1), v3 = __shfl(v, (l0 & 63) + 2), v4 = __shfl(v, (l0 & 63) + 3);
    if (j != 0) return;
    if (n >= 32) {
        h = rotl64(v1, 1) + rotl64(v2, 7) + rotl64(v3, 12) + rotl64(v4, 18);
        h = xxh_merge(h, v1); h = xxh_merge(h, v2); h = xxh_merge(h, v3); h = xxh_merge(h, v4);
    } else h = P5;
    h += (u64)n;
    const u8* q = p + (stripes << 5); const u8* const end = p + n;
    while (q + 8 <= end) { h ^= xxh_round(0, readLE64(q)); h = rotl64(h, 27) * P1 + P4; q += 8; }
    if (q + 4 <= end) { h ^= (u64)readLE32(q) * P1; h = rotl64(h, 23) * P2 + P3; q += 4; }
    while (q < end) { h ^= (*q) * P5; h = rotl64(h, 11) * P1; q++; }
    h ^= h >> 33; h *= P2; h ^= h >> 29; h *= P3; h ^= h >> 32;
    meta[c].checksum = (u32)h;
}

void launch_scan_sizes(const ChunkMeta* meta, u32 nChunks, u64* offsets, u64* total, hipStream_t stream)
{
    hipLaunchKernelGGL(scan_sizes_kernel, dim3(1), dim3(1024), 0, stream, meta, nChunks, offsets, total);
}
void launch_gather(const u8* src, u64 srcSize, const u8* slots, const ChunkMeta* meta, const u64* offsets, u8* dst, u64 dstCapacity,
                   u32 nChunks, u32 chunkBytes, hipStream_t stream)
{
    hipLaunchKernelGGL(gather_kernel, dim3(nChunks), dim3(256), 0, stream, src, srcSize, slots, meta, offsets, dst, dstCapacity, chunkBytes);
}
void launch_xxh64(const u8* src, u64 srcSize, ChunkMeta* meta, u32 nChunks, u32 chunkBytes, u32 frameBlocks, hipStream_t stream)
{
    if (!frameBlocks) frameBlocks = 1;
    const u32 nFrames = (nChunks + frameBlocks - 1) / frameBlocks;
    hipLaunchKernelGGL(xxh64_kernel, dim3((nFrames * 4 + 255) / 256), dim3(256), 0, stream, src, srcSize, meta, nChunks, chunkBytes, frameBlocks);
}

} // namespace zmi
