// seq_enc.hip — sequences section + block/frame finalisation on gfx950.
//
// One wave per chunk, four chunks per 256-thread workgroup.  The 64 lanes turn raw offsets into repcodes (the
// decoder's history rule, U/ZstdDecompressBlock.cs:2387-2443, written as a lane-parallel recurrence) and build the
// LL/OF/ML code histograms in LDS (ZSTD_seqToCodes U/ZstdCompress.cs:3069-3098 + HIST_countFast); lane 0 takes
// ZSTD_selectEncodingType's strategy<lazy decisions (U/ZstdCompressSequences.cs:400-469) and runs FSE_normalizeCount /
// FSE_writeNCount; the whole wave builds each FSE_buildCTable (:471-582); the three interleaved tANS state chains of
// ZSTD_encodeSequences_body (:585-704) run on three lanes while 64 lanes pack the bit fields — straight into the
// chunk's output slot behind the literals section that huf_tree already sized.  Lane 0 then applies
// ZSTD_entropyCompressSeqStore's "compressed enough?" rule (U/ZstdCompress.cs:3357-3392) and writes the block header
// (U/ZstdCompress.cs:4792) and the single-segment frame header (U/ZstdCompress.cs:4817-4929).
// The state chains are serial per block by construction (U/Fse.cs:41-49); the GPU gets its parallelism from the
// 16 384 independent chunks per GiB, which is why this kernel keeps LDS small.
#include "zmi_device.h"
#include "zmi_fse.h"

namespace zmi {

__constant__ u8 cLL_Code[64] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,16,17,17,18,18,19,19,20,20,20,20,21,21,21,21,
    22,22,22,22,22,22,22,22,23,23,23,23,23,23,23,23,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24,24 };
__constant__ u8 cML_Code[128] = { 0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31,
    32,32,33,33,34,34,35,35,36,36,36,36,37,37,37,37,38,38,38,38,38,38,38,38,39,39,39,39,39,39,39,39,
    40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,40,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,41,
    42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42,42 };
__constant__ u8 cLL_bits[36] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,2,2,3,3,4,6,7,8,9,10,11,12,13,14,15,16 };
__constant__ u8 cML_bits[53] = { 0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,
                                 1,1,1,1,2,2,3,3,4,4,5,7,8,9,10,11,12,13,14,15,16 };
__constant__ s16 cLL_defaultNorm[36] = { 4,3,2,2,2,2,2,2,2,2,2,2,2,1,1,1,2,2,2,2,2,2,2,2,2,3,2,1,1,1,1,1,-1,-1,-1,-1 };
__constant__ s16 cML_defaultNorm[53] = { 1,4,3,2,2,2,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,
                                         1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1,-1,-1 };
__constant__ s16 cOF_defaultNorm[29] = { 1,1,1,1,1,1,2,2,2,1,1,1,1,1,1,1,1,1,1,1,1,1,1,1,-1,-1,-1,-1,-1 };

__device__ __forceinline__ u32 ll_code(u32 ll) { return ll > 63 ? highbit32(ll) + 19 : cLL_Code[ll]; }
__device__ __forceinline__ u32 ml_code(u32 ml) { return ml > 127 ? highbit32(ml) + 36 : cML_Code[ml]; }

#ifdef ZMI_LZ_STAMPS
__device__ unsigned long long g_sencStamps[16];
#define ZMI_ESTAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stampAcc[i] += now_ - stampLast; stampLast = now_; } while (0)
#else
#define ZMI_ESTAMP(i) do { } while (0)
#endif

struct SeqWaveLds {
    u32 count[3][64];                   // LL, OF, ML code histograms
    u16 llState[512]; u16 mlState[512]; u16 ofState[256];
    SymTT llTT[36]; SymTT mlTT[53]; SymTT ofTT[32];
    s16 norm[64]; u16 cumul[66]; u8 tableSymbol[512];
    // one batch of 64 sequences: codes in, (state bits, count) out of the three chains, and the packing tile
    u8  bCode[3][64];
    SymTT bTT[3][64];                   // the symbol transforms of the batch's codes, looked up by the 64 lanes (the chains only add and shift)
    u32 bBits[3][64];                   // low 16: value, high 16: number of bits
    u32 tile[192];
};

struct SeqTable { const u16* state; const SymTT* tt; u32 tableLog; };

// ZSTD_selectEncodingType + ZSTD_buildCTable for one of LL/OF/ML, whole wave: the decisions, FSE_normalizeCount and
// FSE_writeNCount run on lane 0 (short), the table itself is built by all lanes.  Returns bytes written to `op` (uniform).
__device__ __forceinline__ u32 build_seq_table(SeqWaveLds& W, u8* op, u32* count, u32 maxPossible, u32 FSELog, u32 nbSeq, u32 lastCode,
                                               const s16* defaultNorm, u32 defaultNormLog, u32 defaultMax, bool defaultAllowedByMax,
                                               u32 strategy, u16* stateTable, SymTT* tt, u32* typeOut, u32* tableLogOut, u32* lastCountSize, u32 lane)
{
    u32 type = 0, max = 0, tableLog = 0, n = 0;
    if (lane == 0) {
        u32 mostFrequent = 0; max = maxPossible;
        while (!count[max]) max--;
        for (u32 s = 0; s <= max; s++) if (count[s] > mostFrequent) mostFrequent = count[s];
        const bool isDefaultAllowed = defaultAllowedByMax ? (max <= defaultMax) : true;
        if (mostFrequent == nbSeq) type = (isDefaultAllowed && nbSeq <= 2) ? 0 : 1;
        else {
            type = 2;
            if (isDefaultAllowed) {
                const u32 mult = 10 - strategy, dynamicFse_nbSeq_min = ((1u << defaultNormLog) * mult) >> 3;
                if (nbSeq < dynamicFse_nbSeq_min || mostFrequent < (nbSeq >> (defaultNormLog - 1))) type = 0;
            }
        }
        if (type == 1) {        // set_rle: FSE_buildCTable_rle
            stateTable[0] = 0; stateTable[1] = 0;
            tt[max].deltaNbBits = 0; tt[max].deltaFindState = 0;
            op[0] = (u8)lastCode;   // the reference writes codeTable[0]; every code is equal here
            n = 1;
        } else if (type == 0) { // set_basic
            for (u32 s = 0; s <= defaultMax; s++) W.norm[s] = defaultNorm[s];
            max = defaultMax; tableLog = defaultNormLog;
        } else {                // set_compressed
            u32 nbSeq_1 = nbSeq;
            tableLog = fse_optimal_table_log(FSELog, nbSeq, max, 2);
            if (count[lastCode] > 1) { count[lastCode]--; nbSeq_1--; }
            fse_normalize_count(W.norm, tableLog, count, nbSeq_1, max, nbSeq_1 >= 2048);
            n = fse_write_ncount(op, W.norm, max, tableLog);
        }
    }
    type = uniform(type); max = uniform(max); tableLog = uniform(tableLog); n = uniform(n);
    wave_lds_sync();
    if (type != 1) fse_build_ctable_wave(stateTable, tt, W.norm, max, tableLog, W.cumul, W.tableSymbol, lane);
    *typeOut = type; *tableLogOut = tableLog;
    if (type == 2) *lastCountSize = n;
    return n;
}

__global__ __launch_bounds__(256) void seq_encode_kernel(Seq* __restrict__ seqs, ChunkMeta* __restrict__ meta,
                                                         u8* __restrict__ slots, u32 nChunks, u32 strategy, u32 checksumFlag, u32 resolveReps,
                                                         u32 dictID, u32 dictIdBytes, u32 initRep0, u32 initRep1, u32 initRep2,
                                                         const u32 frameBlocks, const u32 chunkBytes, const u64 srcSize)
{
    __shared__ SeqWaveLds Ws[4];
    __shared__ u32 sBatchSeq[4];          // sequences each of the four chunks sends through the state chains (0: none)
    __shared__ u32 sFinalState[4][3];
    const u32 lane = lane_id(), wave = uniform(wave_id());
    const u32 c = blockIdx.x * 4 + wave;
    const bool live = c < nChunks;        // (a wave without a chunk still meets the workgroup's barriers below)
    SeqWaveLds& W = Ws[wave];
    ChunkMeta m = {};
    if (live) m = meta_checked(meta[c]);
    const u32 nbSeq = m.nbSeq, n = m.srcSize;
    // Multi-block frames (row f-1): chunk c is block bf of frame c / frameBlocks.  A later block never relies on the repcodes the
    // blocks before it leave behind — whether one of them ends up stored raw (and so leaves the decoder's history untouched,
    // U/ZstdCompress.cs:3620-3640) is only known once it has been encoded — so its history starts as "unknown" (0 matches no
    // offset): offsets are written in full until the block itself has defined the repcode they would use.  That is always a
    // valid encoding (the decoder pushes a full offset whatever it equals) and costs a few bits per block.
    const u32 bf = frameBlocks ? c % frameBlocks : 0u;
    u64 frameLen = n; bool lastBlock = true;
    if (frameBlocks) {
        const u64 fStart = (u64)(c - bf) * chunkBytes, fMax = (u64)frameBlocks * chunkBytes;
        frameLen = (srcSize - fStart) < fMax ? (srcSize - fStart) : fMax;
        lastBlock = (u64)(bf + 1) * chunkBytes >= frameLen;
        if (bf) { initRep0 = 0; initRep1 = 0; initRep2 = 0; }
    }
    Seq* __restrict__ sq = seqs + (u64)c * kMaxSeq;
    u8* const slot = slots + (u64)c * kSlotStride;
    u8* const body = slot + m.fhSize + 3;

#ifdef ZMI_LZ_STAMPS
    unsigned long long stampAcc[8] = {0,0,0,0,0,0,0,0}; unsigned long long stampLast = __builtin_amdgcn_s_memtime();
#endif
    for (u32 i = lane; i < 3 * 64; i += 64) (&W.count[0][0])[i] = 0;
    if (resolveReps) {
        // The match finder stores raw offsets (distance + 3).  Turning them into repcodes is a serial state machine over
        // the sequences — the decoder's history rule (U/ZstdDecompressBlock.cs:2387-2443) run forward — so it lives
        // here: 64 sequences are loaded per step (coalesced), the chain itself runs on scalar values.
        // Written as a recurrence on the history BEFORE each sequence, the rule needs no serial walk:
        //   rep0 before k  = offset of sequence k-1, always (every case of the rule leaves the used offset in front);
        //   rep1 before k  = rep0 before j, j = the latest earlier sequence that was not a plain rep0 hit;
        //   rep2 before k  = rep1 before j, j = the latest earlier sequence that neither hit rep0 nor swapped with rep1.
        // "Latest earlier sequence with a property" is a ballot and a count-leading-zeros; the value comes by shuffle.
        u32 R0 = initRep0, R1 = initRep1, R2 = initRep2;      // history before the batch (uniform): {1,4,8}, or a formatted dictionary's repcodes
        for (u32 b0 = 0; b0 < nbSeq; b0 += 64) {
            const u32 i = b0 + lane;
            Seq s; s.offBase = 4; s.litLength = 0; s.mlBase = 0;
            if (i < nbSeq) s = sq[i];
            const u32 cnt = nbSeq - b0 < 64 ? nbSeq - b0 : 64;
            const bool valid = lane < cnt;
            const u32 off = s.offBase - 3; const bool ll0 = s.litLength == 0;
            u32 rep0b = __shfl_up(off, 1); if (lane == 0) rep0b = R0;
            const bool hit0 = valid && !ll0 && off == rep0b;
            const u64 lt = lanemask_lt();
            const u64 nh0 = ballot(valid && !hit0);
            const u64 m1 = nh0 & lt;
            const u32 g1 = __shfl(rep0b, m1 ? 63 - (int)__builtin_clzll(m1) : 0);
            const u32 rep1b = m1 ? g1 : R1;
            const bool e1 = off == rep1b, hit1 = valid && e1 && !hit0;
            const u64 ns2 = ballot(valid && !(hit0 || hit1));
            const u64 m2 = ns2 & lt;
            const u32 g2 = __shfl(rep1b, m2 ? 63 - (int)__builtin_clzll(m2) : 0);
            const u32 rep2b = m2 ? g2 : R2;
            const bool e2 = off == rep2b, e3 = off == rep0b - 1 && rep0b > 1;
            u32 myCode = off + 3;
            if (!ll0) myCode = hit0 ? 1 : e1 ? 2 : e2 ? 3 : off + 3;
            else      myCode = e1 ? 1 : e2 ? 2 : e3 ? 3 : off + 3;
            // history after the batch = history "before sequence cnt"
            {
                const u64 a1 = cnt < 64 ? nh0 & ((1ull << cnt) - 1) : nh0, a2 = cnt < 64 ? ns2 & ((1ull << cnt) - 1) : ns2;
                const u32 nR1 = a1 ? read_lane(rep0b, 63 - (u32)__builtin_clzll(a1)) : R1;
                const u32 nR2 = a2 ? read_lane(rep1b, 63 - (u32)__builtin_clzll(a2)) : R2;
                R0 = read_lane(off, cnt - 1); R1 = nR1; R2 = nR2;
            }
            if (i < nbSeq) sq[i].offBase = myCode;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    ZMI_ESTAMP(0);
    for (u32 i = lane; i < nbSeq; i += 64) {
        const Seq s = sq[i];
        atomicAdd(&W.count[0][ll_code(s.litLength)], 1u);
        atomicAdd(&W.count[1][highbit32(s.offBase)], 1u);
        atomicAdd(&W.count[2][ml_code(s.mlBase)], 1u);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    ZMI_ESTAMP(1);
    // ---------------- serial section (lane 0): section header + the three table descriptions ----------------
    u8* op = body + m.litSectionSize;
    bool giveUp = n < 7;                 // ZSTD_buildSeqStore: blocks under MIN_CBLOCK_SIZE+header are never compressed
    u32 bodyTablesEnd = 0, typesOk = 0, tLLlog = 0, tOFlog = 0, tMLlog = 0, tLastCount = 0;
    if (!giveUp) {
        const u32 hdr = nbSeq < 128 ? 1u : (nbSeq < 0x7F00 ? 2u : 3u);
        if (lane == 0) {
            if (hdr == 1) op[0] = (u8)nbSeq;
            else if (hdr == 2) { op[0] = (u8)((nbSeq >> 8) + 0x80); op[1] = (u8)nbSeq; }
            else { op[0] = 0xFF; writeLE16(op + 1, nbSeq - 0x7F00); }
        }
        op += hdr;
    }
    if (!giveUp && nbSeq) {             // (uniform: every lane walks the same pointers, lane 0 stores)
        u8* const seqHead = op++;
        const Seq last = sq[nbSeq - 1];
        const u32 lastLL = ll_code(last.litLength), lastOF = highbit32(last.offBase), lastML = ml_code(last.mlBase);
        u32 LLtype, OFtype, MLtype, llLog, ofLog, mlLog, lastCountSize = 0;
        op += build_seq_table(W, op, W.count[0], 35, 9, nbSeq, lastLL, cLL_defaultNorm, 6, 35, false, strategy, W.llState, W.llTT, &LLtype, &llLog, &lastCountSize, lane);
        op += build_seq_table(W, op, W.count[1], 31, 8, nbSeq, lastOF, cOF_defaultNorm, 5, 28, true,  strategy, W.ofState, W.ofTT, &OFtype, &ofLog, &lastCountSize, lane);
        op += build_seq_table(W, op, W.count[2], 52, 9, nbSeq, lastML, cML_defaultNorm, 6, 52, false, strategy, W.mlState, W.mlTT, &MLtype, &mlLog, &lastCountSize, lane);
        // lastCountSize must be the size of the LAST compressed table description (U/ZstdCompress.cs:3196-3224)
        if (lane == 0) *seqHead = (u8)((LLtype << 6) + (OFtype << 4) + (MLtype << 2));

        bodyTablesEnd = (u32)(op - body);
        typesOk = 1; tLLlog = llLog; tOFlog = ofLog; tMLlog = mlLog; tLastCount = lastCountSize;
    }
    // ---- hand the table geometry to the whole wave ----
    bodyTablesEnd = uniform(bodyTablesEnd); typesOk = uniform(typesOk);
    tLLlog = uniform(tLLlog); tOFlog = uniform(tOFlog); tMLlog = uniform(tMLlog); tLastCount = uniform(tLastCount);
    u32 bitstreamSize = 0;
    ZMI_ESTAMP(2);
    {
        // ZSTD_encodeSequences_body (U/ZstdCompressSequences.cs:585-704), 64 sequences per step, last sequence first.  Every wave
        // prepares the batch of its own chunk (codes and their symbol transforms, one sequence per lane) and later packs that
        // batch's bit fields in the reference's order [OF state, ML state, LL state, LL extra, ML extra, OF extra] (a wave prefix
        // sum places them); in between, the state chains — serial, a handful of instructions per step whatever runs them — of
        // ALL FOUR chunks of the workgroup run on twelve lanes of wave 0 (lane 3 w + t = chunk w, table t: LL, OF, ML), so a
        // chain step is issued once per workgroup instead of once per chunk.  Two barriers per batch hand the LDS arrays over.
        if (lane == 0) sBatchSeq[wave] = typesOk ? nbSeq : 0u;
        __syncthreads();
        const u32 mySeq = typesOk ? nbSeq : 0u;
        const u32 q0 = sBatchSeq[0], q1 = sBatchSeq[1], q2 = sBatchSeq[2], q3 = sBatchSeq[3];
        const u32 q01 = q0 > q1 ? q0 : q1, q23 = q2 > q3 ? q2 : q3, maxSeq = q01 > q23 ? q01 : q23;       // uniform over the workgroup
        u8* const out = body + bodyTablesEnd;
        // chain lanes (wave 0): their chunk's tables
        const u32 cw = lane < 12 ? lane / 3 : 0, ct = lane < 12 ? lane % 3 : 0;
        SeqWaveLds& CW = Ws[cw];
        const u16* const stT = ct == 0 ? CW.llState : ct == 1 ? CW.ofState : CW.mlState;
        const SymTT* const ttT = ct == 0 ? CW.llTT : ct == 1 ? CW.ofTT : CW.mlTT;
        const u32 chainSeq = (wave == 0 && lane < 12) ? sBatchSeq[cw] : 0u;
        u32 state = 0;
        u32 carry = 0, carryBits = 0, outWords = 0;
        // room for the bitstream in the chunk's slot: a stream that does not fit is longer than the block it encodes (sequences no
        // finder can produce) — it is cut short and the block stored raw
        const u32 roomWords = (kSlotStride - 64u - (m.fhSize + 3u + bodyTablesEnd)) >> 2;
        bool ovf = false;
        for (u32 done = 0; done < maxSeq; done += 64) {
            const bool act = done < mySeq;                      // uniform per wave
            const u32 cnt = act ? (mySeq - done < 64 ? mySeq - done : 64) : 0;
            const bool have = lane < cnt;
            Seq sv; sv.offBase = 1; sv.litLength = 0; sv.mlBase = 0;
            if (have) sv = sq[nbSeq - 1 - done - lane];
            const u32 llc = ll_code(sv.litLength), ofc = highbit32(sv.offBase), mlc = ml_code(sv.mlBase);
            if (act) {
                W.bCode[0][lane] = (u8)llc; W.bCode[1][lane] = (u8)ofc; W.bCode[2][lane] = (u8)mlc;
                W.bTT[0][lane] = W.llTT[llc]; W.bTT[1][lane] = W.ofTT[ofc]; W.bTT[2][lane] = W.mlTT[mlc];
                for (u32 i = lane; i < 192; i += 64) W.tile[i] = 0;
            }
            __syncthreads();
            ZMI_ESTAMP(3);
            if (done < chainSeq) {
                __builtin_amdgcn_s_setprio(3);          // the serial chains go ahead of the wave-parallel work sharing their SIMD
                // a chain step is: nb = (state + deltaNbBits) >> 16, emit the low nb bits, state = table[(state >> nb) + deltaFindState]
                // (FSE_encodeSymbol, U/Fse.cs:41-49); the transforms come four at a time, ahead of the states that need them
                const u32 ccnt = chainSeq - done < 64 ? chainSeq - done : 64;
                u32 k = 0;
                if (done == 0) { state = fse_init_state2(stT, ttT, CW.bCode[ct][0]); CW.bBits[ct][0] = 0; k = 1; }   // FSE_initCState2: no bits
                for (; k < ccnt && (k & 3); k++) {
                    const SymTT tt = CW.bTT[ct][k];
                    const u32 nb = (state + tt.deltaNbBits) >> 16;
                    CW.bBits[ct][k] = (state & ((1u << nb) - 1)) | (nb << 16);
                    state = stT[(s32)(state >> nb) + tt.deltaFindState];
                }
                for (; k < ccnt; k += 4) {              // k is a multiple of 4: the four transforms lie inside the batch's 64 slots
                    const SymTT t0 = CW.bTT[ct][k], t1 = CW.bTT[ct][k + 1], t2 = CW.bTT[ct][k + 2], t3 = CW.bTT[ct][k + 3];
                    u32 nb = (state + t0.deltaNbBits) >> 16;
                    CW.bBits[ct][k] = (state & ((1u << nb) - 1)) | (nb << 16);
                    state = stT[(s32)(state >> nb) + t0.deltaFindState];
                    if (k + 1 < ccnt) {
                        nb = (state + t1.deltaNbBits) >> 16;
                        CW.bBits[ct][k + 1] = (state & ((1u << nb) - 1)) | (nb << 16);
                        state = stT[(s32)(state >> nb) + t1.deltaFindState];
                    }
                    if (k + 2 < ccnt) {
                        nb = (state + t2.deltaNbBits) >> 16;
                        CW.bBits[ct][k + 2] = (state & ((1u << nb) - 1)) | (nb << 16);
                        state = stT[(s32)(state >> nb) + t2.deltaFindState];
                    }
                    if (k + 3 < ccnt) {
                        nb = (state + t3.deltaNbBits) >> 16;
                        CW.bBits[ct][k + 3] = (state & ((1u << nb) - 1)) | (nb << 16);
                        state = stT[(s32)(state >> nb) + t3.deltaFindState];
                    }
                }
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
            ZMI_ESTAMP(4);
            if (!act) continue;                                 // uniform per wave; the barriers above are met by every wave
            u64 lo = 0, hi = 0; u32 nbTot = 0;
            auto put = [&](u32 v, u32 nb) {
                if (!nb) return;
                const u64 vv = v & ((nb >= 32) ? 0xFFFFFFFFu : ((1u << nb) - 1));
                if (nbTot < 64) { lo |= vv << nbTot; if (nbTot + nb > 64) hi |= vv >> (64 - nbTot); }
                else hi |= vv << (nbTot - 64);
                nbTot += nb;
            };
            if (have) {
                const u32 bOF = W.bBits[1][lane], bML = W.bBits[2][lane], bLL = W.bBits[0][lane];
                put(bOF & 0xFFFF, bOF >> 16); put(bML & 0xFFFF, bML >> 16); put(bLL & 0xFFFF, bLL >> 16);
                put(sv.litLength, cLL_bits[llc]); put(sv.mlBase, cML_bits[mlc]); put(sv.offBase, ofc);
            }
            const u32 incl = wave_scan_incl(nbTot);
            const u32 batchBits = read_lane(incl, 63);
            const u32 bitOff = carryBits + incl - nbTot;
            if (nbTot) {
                const u32 w0 = bitOff >> 5, sh = bitOff & 31;
                const u64 a0 = lo << sh;
                const u64 a1 = sh ? ((lo >> (64 - sh)) | (hi << sh)) : hi;
                const u32 a2 = sh ? (u32)(hi >> (64 - sh)) : 0;
                atomicOr(&W.tile[w0], (u32)a0);
                if ((u32)(a0 >> 32)) atomicOr(&W.tile[w0 + 1], (u32)(a0 >> 32));
                if ((u32)a1) atomicOr(&W.tile[w0 + 2], (u32)a1);
                if ((u32)(a1 >> 32)) atomicOr(&W.tile[w0 + 3], (u32)(a1 >> 32));
                if (a2) atomicOr(&W.tile[w0 + 4], a2);
            }
            if (lane == 0 && carryBits) atomicOr(&W.tile[0], carry);
            const u32 total = carryBits + batchBits;
            const u32 fullWords = total >> 5;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
            if (outWords + fullWords > roomWords) ovf = true;          // (uniform)
            if (!ovf) for (u32 i = lane; i < fullWords; i += 64) *(u32u*)(out + 4 * (outWords + i)) = W.tile[i];
            carry = W.tile[fullWords]; carryBits = total & 31;
            if (!ovf) outWords += fullWords;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier();
        }
        ZMI_ESTAMP(3);
        if (wave == 0 && lane < 12) sFinalState[cw][ct] = state;
        __syncthreads();
        // FSE_flushCState x3 (ML, OF, LL) + end mark (BIT_closeCStream)
        if (typesOk && lane == 0) {
            const u32 stLL = sFinalState[wave][0], stOF = sFinalState[wave][1], stML = sFinalState[wave][2];
            u64 acc = carry; u32 nb = carryBits;
            acc |= (u64)(stML & ((1u << tMLlog) - 1)) << nb; nb += tMLlog;
            acc |= (u64)(stOF & ((1u << tOFlog) - 1)) << nb; nb += tOFlog;
            acc |= (u64)(stLL & ((1u << tLLlog) - 1)) << nb; nb += tLLlog;
            acc |= 1ull << nb; nb += 1;
            u8* p = out + 4 * outWords;
            const u32 nbytes = (nb + 7) >> 3;
            for (u32 i = 0; i < nbytes; i++) { p[i] = (u8)acc; acc >>= 8; }
            bitstreamSize = 4 * outWords + nbytes;
        }
        bitstreamSize = uniform(bitstreamSize);
        if (ovf) giveUp = true;
    }
#ifdef ZMI_LZ_STAMPS
    if (lane == 0) for (int i = 0; i < 8; i++) atomicAdd(&g_sencStamps[i], stampAcc[i]);
#endif
    if (lane != 0 || !live) return;
    if (typesOk) {
        op = body + bodyTablesEnd + bitstreamSize;
        const u32 lastCountSize = tLastCount;
        if (lastCountSize && lastCountSize + bitstreamSize < 4) giveUp = true;      // 1.3.4 decoder quirk (:3346-3350)
    }
    u32 cSize = (u32)(op - body);
    if (!giveUp && cSize >= n - ((n >> 6) + 2)) giveUp = true;                      // ZSTD_minGain

    // ---- frame header in front of the frame's first block (ZSTD_writeFrameHeader, U/ZstdCompress.cs:4817-4929): single segment with
    // the content size, or — ZSTD_c_contentSizeFlag = 0 (checksumFlag bit 1) — a window descriptor and no content size: the smallest
    // power of two that holds the frame (>= 1 KiB), so that no offset and no block exceeds the declared window ----
    if (bf == 0) {
        writeLE32(slot, 0xFD2FB528u);
        const u32 didCode = dictIdBytes == 4 ? 3u : dictIdBytes;          // dictID field of 0, 1, 2 or 4 bytes (U/ZstdCompress.cs:4843-4849, 4896-4918)
        const u32 wlSet = (checksumFlag >> 8) & 31u;          // an explicit window (frames of independent blocks, each a window long)
        if ((checksumFlag & 2u) || wlSet) {
            u32 wl = wlSet;
            if (!wl) { wl = 10; while (((u64)1 << wl) < frameLen) ++wl; }
            const u32 fcsCode = (checksumFlag & 2u) ? 0u : (frameLen >= 256) + (frameLen >= 65536 + 256) + (frameLen > 0xFFFFFFFFull);
            slot[4] = (u8)(didCode + ((checksumFlag & 1u) << 2) + (fcsCode << 6));
            slot[5] = (u8)((wl - 10) << 3);
            for (u32 i = 0; i < dictIdBytes; i++) slot[6 + i] = (u8)(dictID >> (8 * i));
            u8* const fcs = slot + 6 + dictIdBytes;               // (not a single segment: a content size below 256 has no field)
            if (fcsCode == 1) writeLE16(fcs, (u32)frameLen - 256);
            else if (fcsCode == 2) writeLE32(fcs, (u32)frameLen);
            else if (fcsCode == 3) { writeLE32(fcs, (u32)frameLen); writeLE32(fcs + 4, (u32)(frameLen >> 32)); }
        } else {
            const u32 fcsCode = (frameLen >= 256) + (frameLen >= 65536 + 256) + (frameLen > 0xFFFFFFFFull);
            slot[4] = (u8)(didCode + ((checksumFlag & 1u) << 2) + (1u << 5) + (fcsCode << 6));
            for (u32 i = 0; i < dictIdBytes; i++) slot[5 + i] = (u8)(dictID >> (8 * i));
            u8* const fcs = slot + 5 + dictIdBytes;
            if (fcsCode == 0) fcs[0] = (u8)frameLen;
            else if (fcsCode == 1) writeLE16(fcs, (u32)frameLen - 256);
            else if (fcsCode == 2) writeLE32(fcs, (u32)frameLen);
            else { writeLE32(fcs, (u32)frameLen); writeLE32(fcs + 4, (u32)(frameLen >> 32)); }
        }
    }
    u8* const bh = slot + m.fhSize;
    const u32 lastBit = lastBlock ? 1u : 0u;                  // Last_Block (U/ZstdCompress.cs:4757-4760)
    if (giveUp) {            // ZSTD_noCompressBlock: the gather kernel copies the n source bytes behind this header
        writeLE24(bh, lastBit + (0u << 1) + (n << 3));
        m.blockType = 0; m.bodySize = 0; cSize = n;
    } else {
        writeLE24(bh, lastBit + (2u << 1) + (cSize << 3));
        m.blockType = 2; m.bodySize = cSize;
    }
    m.outSize = m.fhSize + 3 + cSize + (((checksumFlag & 1u) && lastBlock) ? 4 : 0);
    meta[c] = m;
}

void launch_seq_encode(Seq* seqs, ChunkMeta* meta, u8* slots, u32 nChunks, u32 strategy, u32 checksumFlag, u32 resolveReps,
                       u32 dictID, u32 dictIdBytes, const u32* initReps, u32 frameBlocks, u32 chunkBytes, u64 srcSize, hipStream_t stream)
{
    hipLaunchKernelGGL(seq_encode_kernel, dim3((nChunks + 3) / 4), dim3(256), 0, stream, seqs, meta, slots, nChunks, strategy, checksumFlag, resolveReps,
                       dictID, dictIdBytes, initReps[0], initReps[1], initReps[2], frameBlocks, chunkBytes, srcSize);
}

#ifdef ZMI_LZ_STAMPS
extern "C" void ZSTDMI_debugReadSeqEncStamps(unsigned long long* out16, int reset)
{
    (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sencStamps), 16 * sizeof(unsigned long long));
    if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sencStamps), z, sizeof z); }
}
#endif

} // namespace zmi
