// decode_origin.hip — match execution for frames that are too long for one wave (ZSTD_execSequence,
// U/ZstdDecompressBlock.cs:2187-2262, over ALL sequences of a frame at once).
//
// What Compressor.Wrap of the reference writes for any input is ONE frame of 128 KiB blocks chained by the window
// (U/ZstdCompress.cs:4690-4815): every match may copy from anywhere in the window behind it, so match execution is ordered
// inside a frame, and exec_matches (decode_seq.hip) walks a frame on one wave — about 150 MB/s.  A call of a few long frames
// would run on a few waves.  For such frames the copies are turned into a pointer problem that has no order:
//
//   origin[i]   for every output byte i of the frame: the frame position its value comes from.  A literal byte is its own
//               origin (a root: place_literals has put its value there already); byte j of a match at d with offset `off`
//               starts out pointing at d - off + (j mod off) — the byte-wise overlap semantics of :2247-2259 — or, past the
//               frame's start, into the dictionary (ZSTD_execSequence's extDict branch, :2223-2250), which is a root too.
//   jump        origin[i] = origin[origin[i]], in place, for every byte at once: after r rounds every byte whose chain of copies
//               is at most 2^r long points at a root.  Rounds stop when one changes nothing (a chain is at most the frame long:
//               30 rounds bound it).  In-place is safe: whatever a racing read returns is an ancestor of the byte, old or new.
//   gather      out[i] = out[origin[i]] for every non-root byte; roots are never written here, so nothing is ordered.  (Chains that
//               the rounds left unfinished are walked to their root: origins strictly decrease, the walk always ends.)
//
// O(n log depth) work instead of O(n), four bytes of origin per output byte — but every step is a plain parallel sweep at HBM
// speed, where the ordered walk is bound by the latency of one wave.  The host sends a frame here when the sweep over all such
// frames is shorter than the walk of that frame alone (decompress_device); everything else keeps the one-wave walk, which wins
// whenever a call has thousands of frames.
#include "zmi_decode.h"

namespace zmi {

constexpr u32 kOriginDict = 0x80000000u;        // origin values with this bit: index into the dictionary content
constexpr u32 kOriginSettled = 0x40000000u;     // ... with this one: the position they name is a root (frames on this path are below 1 GiB)

// frames of at least minBytes with sequences take the path: an entry in `list`, a range of the origin array
__global__ __launch_bounds__(256) void origin_select_kernel(FrameDesc* __restrict__ frames, u32 nFrames, u64 minBytes, u32* __restrict__ list, u32 listCap,
                                                            u64 originCap, u32* __restrict__ status)
{
    for (u32 f = blockIdx.x * 256 + threadIdx.x; f < nFrames; f += gridDim.x * 256) {
        FrameDesc& F = frames[f];
        if (F.bad || !F.hasSeq || F.dstSize < minBytes || F.dstSize >= (1ull << 30)) continue;
        const u64 need = (F.dstSize + 1023) & ~(u64)1023;       // whole regions of origin_jump_kernel
        const u64 at = atomicAdd(reinterpret_cast<unsigned long long*>(status + kStOriginLo), (unsigned long long)need);
        if (at + need > originCap) continue;                 // (cannot happen: the host sized the array from the same frames' bounds)
        const u32 idx = atomicAdd(&status[kStOriginFrames], 1u);
        if (idx >= listCap) continue;
        list[idx] = f; F.viaOrigin = 1; F.originOff = at;
    }
}

// origin[i] = i
__global__ __launch_bounds__(256) void origin_fill_kernel(const FrameDesc* __restrict__ frames, const u32* __restrict__ list, const u32* __restrict__ status,
                                                          u32* __restrict__ origin)
{
    if (blockIdx.y >= status[kStOriginFrames]) return;
    const FrameDesc& F = frames[list[blockIdx.y]];
    u32* __restrict__ const P = origin + F.originOff;
    const u32 n4 = (u32)((F.dstSize + 3) >> 2);                 // (the range is padded to 1024 entries)
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        uint4 v; v.x = 4 * i; v.y = 4 * i + 1; v.z = 4 * i + 2; v.w = 4 * i + 3;
        reinterpret_cast<uint4*>(P)[i] = v;
    }
}

// the match bytes' first origins: one wave per block, 64 sequences at a time, the batch's match bytes dealt to the lanes in order
__global__ __launch_bounds__(256) void origin_init_kernel(const FrameDesc* __restrict__ frames, const BlockDesc* __restrict__ blocks, const u32* __restrict__ list,
                                                          const SeqRec* __restrict__ recs, u32* __restrict__ status, u32* __restrict__ origin, const u32 dictSize)
{
    if (blockIdx.y >= status[kStOriginFrames]) return;
    const FrameDesc& F = frames[list[blockIdx.y]];
    u32* __restrict__ const P = origin + F.originOff;
    const u32 lane = lane_id(), first = uniform(F.firstBlock), nb = uniform(F.nbBlocks);
    for (u32 k = blockIdx.x * 4 + uniform(wave_id()); k < nb; k += gridDim.x * 4) {
        const BlockDesc& B = blocks[first + k];
        const u32 nbSeq = uniform(B.type == 2 ? B.nbSeq : 0u);
        if (!nbSeq || uniform(B.err)) continue;
        const u32 bRel = (u32)B.dstRel;                          // (frames on this path are below 1 GiB)
        const u32 in0 = uniform(B.repIn[0]), in1 = uniform(B.repIn[1]), in2 = uniform(B.repIn[2]);
        const SeqRec* __restrict__ const rec = recs + B.seqBase;
        SeqRec rNext; rNext.lo = 0; rNext.hi = 0;
        if (lane < nbSeq) rNext = rec[lane];
        u32 outBase = 0;
        for (u32 base = 0; base < nbSeq; base += 64) {
            const u32 cnt = nbSeq - base < 64 ? nbSeq - base : 64;
            const bool have = lane < cnt;
            const SeqRec r = rNext;
            if (base + 64 + lane < nbSeq) rNext = rec[base + 64 + lane];
            SeqLane q;
            outBase = seq_batch(r, have, in0, in1, in2, outBase, q);
            const u32 d = bRel + q.pos + q.ll;                   // frame position of my match
            // offset beyond everything in front of the match (+ the dictionary): corruption (:2218-2223)
            if (ballot(have && ((u64)q.off > (u64)d + dictSize || q.off == 0))) { if (lane == 0) report_error(status, (u64)first + k, kStageExec, kErrCorruption); break; }
            const u32 ml = have ? q.ml : 0u;
            const u32 incl = wave_scan_incl(ml), excl = incl - ml, total = read_lane(incl, 63);
            for (u32 t0 = 0; t0 < total; t0 += 64) {
                const u32 t = t0 + lane;
                u32 j = 0;                                       // the sequence that owns match byte t: first lane whose inclusive count exceeds t
#pragma unroll
                for (u32 st = 32; st; st >>= 1) { const u32 v = __shfl(incl, (int)(j + st - 1)); if (v <= t) j += st; }
                const u32 dj = __shfl(d, (int)j), oj = __shfl(q.off, (int)j), ej = __shfl(excl, (int)j);
                if (t < total) {
                    u32 w = t - ej;                              // byte of the match; its source repeats with period `off` (:2247-2259)
                    if (w >= oj) w %= oj;
                    const s64 sp = (s64)dj - (s64)oj + (s64)w;
                    P[dj + (t - ej)] = sp >= 0 ? (u32)sp : (kOriginDict | (u32)((s64)dictSize + sp));
                }
            }
        }
    }
}

// one round of origin[i] = origin[origin[i]]; four entries per thread, 1024 per workgroup and step = one REGION.
// An entry found to point at a root is marked (kOriginSettled) and never looked up again: a lookup is a scattered 4-byte read, the
// expensive part of a round, and most bytes of ordinary data sit one or two copies away from a literal.  A region whose entries
// are all roots or settled is finished for good and says so in `done` (one word per region, behind the origins): text settles in
// six or seven rounds, only stretches with deep chains (runs, periodic data: every match copies the one before it) go on, so a
// round costs what is still moving, not the whole frame.
__global__ __launch_bounds__(256) void origin_jump_kernel(const FrameDesc* __restrict__ frames, const u32* __restrict__ list, u32* __restrict__ status,
                                                          u32* __restrict__ origin, u32* __restrict__ done, const u32 round)
{
    if (blockIdx.y >= status[kStOriginFrames]) return;
    if (round && !status[kStOriginChanged + round - 1]) return;             // the round before left nothing unsettled
    const FrameDesc& F = frames[list[blockIdx.y]];
    u32* const P = origin + F.originOff;
    u32* const D = done + (F.originOff >> 10);
    const u32 n4 = (u32)((F.dstSize + 3) >> 2);
    bool open = false;
    constexpr u32 kFinal = kOriginDict | kOriginSettled;
    for (u32 i0w = blockIdx.x * 256; i0w < n4; i0w += gridDim.x * 256) {    // (uniform: the barrier below is met by every thread)
        const u32 region = i0w >> 8, i = i0w + threadIdx.x;
        if (round && D[region]) continue;
        bool mine = false;
        if (i < n4) {
            uint4 v = reinterpret_cast<const uint4*>(P)[i];
            const u32 i0 = 4 * i;
            // entries still on their way: not a root, not final.  All lookups of the thread are issued before any is used; four
            // entries that name four consecutive positions (the inside of a match) are looked up with one 16-byte read
            const bool n0 = v.x != i0 && !(v.x & kFinal), n1 = v.y != i0 + 1 && !(v.y & kFinal), n2 = v.z != i0 + 2 && !(v.z & kFinal), n3 = v.w != i0 + 3 && !(v.w & kFinal);
            if (n0 | n1 | n2 | n3) {
                uint4 q;
                if (n0 && n1 && n2 && n3 && v.y == v.x + 1 && v.z == v.x + 2 && v.w == v.x + 3) {
                    const u32* a = P + v.x;
                    q.x = a[0]; q.y = a[1]; q.z = a[2]; q.w = a[3];           // (one global_load_dwordx4: the address is dword-aligned)
                } else {
                    q.x = P[n0 ? v.x : i0]; q.y = P[n1 ? v.y : i0]; q.z = P[n2 ? v.z : i0]; q.w = P[n3 ? v.w : i0];
                }
                // the entry named is a root: settled where it is; else take over what it names (a dictionary position and a settled entry are final as they are)
                if (n0) { if (q.x == v.x) v.x |= kOriginSettled; else { v.x = q.x; mine |= !(q.x & kFinal); } }
                if (n1) { if (q.y == v.y) v.y |= kOriginSettled; else { v.y = q.y; mine |= !(q.y & kFinal); } }
                if (n2) { if (q.z == v.z) v.z |= kOriginSettled; else { v.z = q.z; mine |= !(q.z & kFinal); } }
                if (n3) { if (q.w == v.w) v.w |= kOriginSettled; else { v.w = q.w; mine |= !(q.w & kFinal); } }
                reinterpret_cast<uint4*>(P)[i] = v;
            }
        }
        const int any = __syncthreads_or(mine ? 1 : 0);
        if (threadIdx.x == 0) D[region] = any ? 0u : 1u;
        open |= any != 0;
    }
    if (open && threadIdx.x == 0) status[kStOriginChanged + round] = 1;
}

// out[i] = out[origin[i]]; four bytes per thread
__global__ __launch_bounds__(256) void origin_gather_kernel(const FrameDesc* __restrict__ frames, const u32* __restrict__ list, const u32* __restrict__ status,
                                                            const u32* __restrict__ origin, u8* __restrict__ out, const u8* __restrict__ dict)
{
    if (blockIdx.y >= status[kStOriginFrames] || status[kStErr]) return;
    const FrameDesc& F = frames[list[blockIdx.y]];
    if (F.bad) return;
    const u32* __restrict__ const P = origin + F.originOff;
    u8* const fout = out + F.dstOff;
    const u32 n = (u32)F.dstSize, n4 = (n + 3) >> 2;
    auto value = [&](u32 p) -> u32 {                             // the byte a position with origin entry p != itself takes
        if (!(p & (kOriginDict | kOriginSettled)))               // (what the rounds left: origins strictly decrease, the walk ends)
            for (;;) { const u32 q = P[p]; if (q == p) break; if (q & (kOriginDict | kOriginSettled)) { p = q; break; } p = q; }
        return (p & kOriginDict) ? (u32)dict[p & ~kOriginDict] : (u32)fout[p & ~kOriginSettled];
    };
    for (u32 i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
        const uint4 v = reinterpret_cast<const uint4*>(P)[i];
        const u32 i0 = 4 * i;
        const bool r0 = v.x == i0, r1 = v.y == i0 + 1, r2 = v.z == i0 + 2, r3 = v.w == i0 + 3;
        if (r0 && r1 && r2 && r3) continue;                      // four literal bytes
        if (i0 + 4 <= n) {
            const u32 b0 = r0 ? (u32)fout[i0] : value(v.x), b1 = r1 ? (u32)fout[i0 + 1] : value(v.y);
            const u32 b2 = r2 ? (u32)fout[i0 + 2] : value(v.z), b3 = r3 ? (u32)fout[i0 + 3] : value(v.w);
            *(u32u*)(fout + i0) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        } else {
            if (!r0 && i0 < n) fout[i0] = (u8)value(v.x);
            if (!r1 && i0 + 1 < n) fout[i0 + 1] = (u8)value(v.y);
            if (!r2 && i0 + 2 < n) fout[i0 + 2] = (u8)value(v.z);
        }
    }
}

void launch_origin_select(FrameDesc* frames, u32 nFrames, u64 minBytes, u32* list, u32 listCap, u64 originCap, u32* status, hipStream_t stream)
{
    const u32 grid = (nFrames + 255) / 256;
    hipLaunchKernelGGL(origin_select_kernel, dim3(grid < 1024 ? grid : 1024), dim3(256), 0, stream, frames, nFrames, minBytes, list, listCap, originCap, status);
}

// Geometry of the sweeps: x = workgroups per frame (each strides over the frame's regions), y = frames.  About 8192 workgroups in all
// (a launch of 65 536 that only find out that there is nothing left to do takes 170 us), at least 64 per frame, never more than the
// longest frame has regions.
static dim3 origin_grid(u64 maxFrameBytes, u32 listCap)
{
    u64 wg = (maxFrameBytes + 4095) / 4096;                      // a workgroup's step covers 1024 entries
    const u64 share = 8192 / listCap > 64 ? 8192 / listCap : 64;
    if (wg > share) wg = share;
    if (wg > 4096) wg = 4096;
    if (wg < 1) wg = 1;
    return dim3((u32)wg, listCap);
}
// select + fill + init (queued behind block_offsets; the literals are not needed yet)
void launch_origin_init(const FrameDesc* frames, const BlockDesc* blocks, const u32* list, u32 listCap, u64 maxFrameBytes, const SeqRec* recs, u32* status,
                        u32* origin, u32 dictSize, hipStream_t stream)
{
    const dim3 grid = origin_grid(maxFrameBytes, listCap), tb(256);
    u64 bw = (maxFrameBytes / (128u << 10) + 4) / 4;             // one wave per block of 128 KiB
    if (bw > 4096) bw = 4096;
    hipLaunchKernelGGL(origin_fill_kernel, grid, tb, 0, stream, frames, list, (const u32*)status, origin);
    hipLaunchKernelGGL(origin_init_kernel, dim3((u32)bw, listCap), tb, 0, stream, frames, blocks, list, recs, status, origin, dictSize);
}
// rounds [r0, r1) of the pointer jumping; done: one word per 1024 origins, any contents before round 0
void launch_origin_jump(const FrameDesc* frames, const u32* list, u32 listCap, u64 maxFrameBytes, u32* status, u32* origin, u32* done, u32 r0, u32 r1, hipStream_t stream)
{
    const dim3 grid = origin_grid(maxFrameBytes, listCap), tb(256);
    for (u32 r = r0; r < r1 && r < kOriginRounds; ++r) hipLaunchKernelGGL(origin_jump_kernel, grid, tb, 0, stream, frames, list, status, origin, done, r);
}
void launch_origin_gather(const FrameDesc* frames, const u32* list, u32 listCap, u64 maxFrameBytes, const u32* status, const u32* origin, u8* out, const u8* dict, hipStream_t stream)
{
    hipLaunchKernelGGL(origin_gather_kernel, origin_grid(maxFrameBytes, listCap), dim3(256), 0, stream, frames, list, status, origin, out, dict);
}

} // namespace zmi
