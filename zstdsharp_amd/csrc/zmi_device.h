// zmi_device.h — device-side helpers for the gfx950 kernels (wave = 64 lanes; no other target is supported).
#pragma once
#include <hip/hip_runtime.h>
#include "zmi_common.h"

namespace zmi {

// gfx950 (amdhsa) runs with unaligned access enabled for global and LDS; hipcc lowers these to single
// global_load_dwordx2 / ds_read_b64 instructions.
typedef u64 __attribute__((aligned(1))) u64u;
typedef u32 __attribute__((aligned(1))) u32u;
typedef u16 __attribute__((aligned(1))) u16u;

constexpr u32 kWave = 64;

__device__ __forceinline__ u32 lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ u32 wave_id() { return threadIdx.x >> 6; }
__device__ __forceinline__ u32 uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u32 read_lane(u32 v, u32 l) { return (u32)__builtin_amdgcn_readlane((int)v, (int)uniform(l)); }
__device__ __forceinline__ u64 ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ u32 popc64(u64 x) { return (u32)__builtin_popcountll(x); }
__device__ __forceinline__ u32 ctz64(u64 x) { return (u32)__builtin_ctzll(x); }
__device__ __forceinline__ u32 highbit32(u32 v) { return 31u - (u32)__builtin_clz(v); }
__device__ __forceinline__ u64 lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ u32 wave_scan_incl(u32 v)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { u32 t = __shfl_up(v, d); if ((int)lane_id() >= d) v += t; }
    return v;
}
// make one wave's LDS writes visible to its other lanes (single-wave workgroups need no s_barrier)
// workgroup barrier that orders LDS only: __syncthreads() also waits until every global store of the wave has been acknowledged
// (vmcnt), which the region parse does not need where it only writes results out
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void wave_lds_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
__device__ __forceinline__ u32 wave_sum(u32 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ u32 wave_max(u32 v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { u32 t = __shfl_xor(v, d); v = t > v ? t : v; }
    return v;
}

__device__ __forceinline__ u32 readLE16(const u8* p) { return *(const u16u*)p; }
__device__ __forceinline__ u32 readLE24(const u8* p) { return (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16); }
__device__ __forceinline__ u32 readLE32(const u8* p) { return *(const u32u*)p; }
__device__ __forceinline__ u64 readLE64(const u8* p) { return *(const u64u*)p; }
__device__ __forceinline__ void writeLE16(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); }
__device__ __forceinline__ void writeLE24(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); }
__device__ __forceinline__ void writeLE32(u8* p, u32 v) { p[0] = (u8)v; p[1] = (u8)(v >> 8); p[2] = (u8)(v >> 16); p[3] = (u8)(v >> 24); }

// What the entropy stage may rely on in a ChunkMeta whatever the match finder left there (a finder bug, or an experiment build
// without its emit phase, must not turn into out-of-bounds addresses downstream): sizes inside the chunk's buffers; a record that
// breaks them is taken as "no sequences".  Every consumer applies the same function, so all of them see the same chunk.
__device__ __forceinline__ ChunkMeta meta_checked(ChunkMeta m)
{
    if (m.srcSize > kChunkSize) m.srcSize = kChunkSize;
    if (m.nbSeq > kMaxSeq || m.litSize > m.srcSize) { m.nbSeq = 0; if (m.litSize > m.srcSize) m.litSize = m.srcSize; }
    if (m.fhSize > 18) m.fhSize = 18;
    return m;
}

// frame header size for a chunk of n bytes: magic + FHD + FCS, single-segment (U/ZstdCompress.cs:4817-4929)
__host__ __device__ __forceinline__ u32 frame_header_size(u32 n) { return 4 + 1 + (n < 256 ? 1 : (n < 65536 + 256 ? 2 : 4)); }
__host__ __device__ __forceinline__ u32 frame_header_size64(u64 n) { return 4 + 1 + (n < 256 ? 1 : (n < 65536 + 256 ? 2 : (n <= 0xFFFFFFFFull ? 4 : 8))); }

} // namespace zmi
