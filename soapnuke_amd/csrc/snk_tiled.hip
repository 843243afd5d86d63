// snk_tiled.hip -- the wave-tiled fast kernel of the SOAPnuke-filter hot path (gfx950).
//
// One wavefront owns a tile of 64 read pairs and walks it in three phases and a hand-over:
//
//  phase 1  The tile's read bytes arrive in LDS by DMA (global_load_lds, 16 B/lane, next chunk in flight).  Per read
//           the wave reads its row twice: as dwords -- lane l holds bases and qualities 4l..4l+3, and every predicate is
//           evaluated on the four bytes at once and shifted into packed collectors (2-bit character codes, "quality above
//           lowQual", in the FULL variant "quality >= the low-quality-end thresholds" and the byte sums of the mean-quality
//           filter by v_sad_u8; a per-READ flag "some character is not the letter of its code" travels through the carry)
//           -- and as bytes, lane = position, one 64-position strip per instruction, for the raw per-position quality
//           histogram in LDS (lane l owns positions l, l+64, ...: the adds of one instruction never collide).  The
//           collectors are parked every 4 / 8 reads.  LDS reads / adds / waits are hand-placed asm, two reads ahead.
//  hand-over  The parked dwords hold, per byte, the bits of 8 (codes: 4) reads at one position.  Byte transposes + wide
//           LDS writes/reads give every lane = position its 64 read bits per plane; their popcounts ARE the raw base
//           histogram of the position (one LDS add per letter, strip and tile), and 64 x 64 bit-matrix transposes
//           (snk_bittr.hip.h: v_permlane32/16_swap + DPP) turn them into the per-read bit planes of phase 2.
//  phase 2  lane = READ.  Each lane now holds its read as bit planes (one bit per
//           position).  Adapter search (src/read_filter.cpp:707-790) runs bit-sliced over
//           all candidate offsets at once: for the first S-1 adapter characters (S =
//           segMatchThr) no run of S matches can complete, so the only possible event is
//           "mismatch budget exceeded" -> a unary mismatch counter per candidate kept in
//           bit planes, one funnel shift + a few logic ops per 32 candidates per step.
//           The handful of survivors is decided exactly with the closed form of the
//           reference's early-exit scan (first run of S ones vs (budget+1)-th zero, SURVEY
//           appendix D).  Low-quality-end / polyG / polyX runs are ctz/clz on the planes;
//           the discard cascade uses host-precomputed integer thresholds (no fp32 here).
//  phase 3  lane = position again, only for reads that were discarded or trimmed: their
//           (removed part of the) contribution goes to a second LDS histogram, so that
//           clean = raw - removed needs no second pass over the surviving 80 %.
//
// LDS holds, per mate, raw and removed {base[pos][5], qual[pos][nq]} histograms as 16-bit
// counters packed two per dword (strips 2k and 2k+1 share dwords, so one strip never
// hits a dword twice); the workgroup flushes them to the global uint64 block before a
// counter can overflow; the few hot trimming-position counters live in a per-workgroup
// uint32 copy in HBM.  Everything is integer/byte work: no MFMA; bound by HBM in theory, in
// practice by the latency of the few KB per wave that the LDS budget lets the DMA keep in
// flight.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include "snk_common.hip.h"
#include "snk_bittr.hip.h"

using namespace snk;

// ablation builds for profiling only (tools/ablate.sh, results are wrong by design): 1 = phase 1 + hand-over only,
// 2 = phases 1+2, 3 = no phase 3, 4 = no adapter search, 5 = no trimming-position counters, 6 = no exact
// decision of adapter candidates, 7 = no adapter screening, 8 / 9 / 10 = no polyX / no low-quality-end + polyG /
// no trim_finish (FULL variant), 11 = no LDS histogram adds, 12 = no DMA, 14 = DMA never waited for, 15 = 12-like
// phase 1 with an L2-resident source (see SNK_L2SRC), 16 = no per-read quality sum (FULL), 17 = no low-quality-end collectors (FULL).
// The shipped library is built with 0.  SNK_ONLY_NW=5 compiles the 129..160-position instances only (ablation builds).
#ifndef SNK_ABL
#define SNK_ABL 0
#endif
// tile of wave w of this workgroup within one round: 0 = wave-major (neighbouring workgroups take neighbouring
// tiles), 1 = workgroup-major (the waves of a workgroup take neighbouring tiles: few pages per CU)
#ifndef SNK_L2SRC
#define SNK_L2SRC 0      // profiling only: with SNK_ABL == 15
#endif
#ifndef SNK_ORDER
#define SNK_ORDER 1
#endif
// wave priorities of phase 1 / hand-over / adapter search + pair level / phase 3, as a 4-digit number (tools/ab.sh experiments)
#ifndef SNK_PAIR
// Two reads share the LDS ops of their short last quality strip (129..160 positions: 7 instead of 8 LDS ops and one clamp + address
// pair less per read).  1 (default): on the static-shape loop only, where the shared strip costs no address select -- a second base
// register, the row an immediate (round 5; the variant round 4's budget asked for: octet loop 124 -> 116 VALU, 64 -> 56 LDS ops);
// 2: on the run-time-shape loop too, whose address selects made it 4 % SLOWER when it was measured (2.99 vs 2.87 ms,
// profiles/r02_pair_ab.txt); 0: off.  tools/ab.sh builds the other two for an A/B.
#define SNK_PAIR 1
#endif
#ifndef SNK_PRIO
#define SNK_PRIO 3210
#endif
#if SNK_ORDER
#define SNK_TILE_OF(w) ((long)blockIdx.x * Wc + (long)(w))
#else
#define SNK_TILE_OF(w) ((long)(w) * gridDim.x + blockIdx.x)
#endif

#include "snk_adapter_bits.hip.h"

#include "snk_gfx950.hip.h"      // the hand-placed instructions: LDS reads / adds / waits, the LDS DMA, v_writelane

namespace {

// 4 x 4 byte transpose: out[k] = bytes k of in[0..3] (8 v_perm)
__device__ __forceinline__ void byte_tr4(u32 i0, u32 i1, u32 i2, u32 i3, u32 (&o)[4]) {
    const u32 t0 = __builtin_amdgcn_perm(i1, i0, 0x05010400u), t1 = __builtin_amdgcn_perm(i1, i0, 0x07030602u);
    const u32 t2 = __builtin_amdgcn_perm(i3, i2, 0x05010400u), t3 = __builtin_amdgcn_perm(i3, i2, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}
typedef u32 v4u __attribute__((ext_vector_type(4)));
typedef u32 v2u __attribute__((ext_vector_type(2)));
typedef u32 v16u __attribute__((ext_vector_type(16)));
typedef u32 v8u __attribute__((ext_vector_type(8)));

// compile-time strip loop (the strip number feeds immediate offsets of the asm above)
template <int V> struct IntC { static constexpr int v = V; };
template <int... I, class F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F &&f) { (f(IntC<I>{}), ...); }

// sum over the 64 lanes, uniform result: DPP row scan + two row broadcasts (6 VALU, no LDS round trips)
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);    // row_shr:8  -> lane 15 of a row = row total
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);    // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);    // row_bcast:31 into rows 2 and 3
    return rl(v, 63);
}

// OR over the 64 lanes, uniform result
__device__ __forceinline__ u32 wave_or(u32 x) {
    int v = (int)x;
    v |= __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);
    v |= __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);
    return (u32)rl(v, 63);
}

// A mate's ReadState between its phase 2 and the pair level, four registers instead of fourteen (the first mate's state has to
// survive the second mate's phases 1 and 2; kept whole it was spilled at the top of every mate iteration: 26 scratch stores per
// lane and tile-mate, 2.1 GB of scratch writes per 10 M pairs in the FULL variant).  Every count is at most the tile kernel's 256
// positions (9 bits); inc_ada <=> adacut >= 0; hd_h / hd_t are parameters (trim_finish) and are not stored.
// ... and of the two mates' states only ONE is carried in registers: the first mate's is parked in its own 16-byte output record
// (which the pair level fills later anyway: one coalesced 1 KB store per tile and one load instead of six spilled registers stored
// at the top and reloaded at the bottom of EVERY mate iteration) together with the mate's input-error code.
struct PackedRS { u64 a; u32 lq; int sumq; };
__device__ __forceinline__ PackedRS rs_pack(const ReadState &r, int estat) {
    PackedRS p;
    const u32 lo = (u32)r.len | ((u32)r.n_a << 9) | ((u32)r.n_n << 18) | ((u32)r.lowq << 27);                  // lowq: low 5 bits here
    const u32 hi = ((u32)r.lowq >> 5) | ((u32)r.clen << 4) | ((u32)r.start << 13) | ((u32)(r.adacut + 1) << 22) | ((u32)r.polyx << 31);
    p.a = ((u64)hi << 32) | lo;
    p.lq = ((u32)(r.lq_h + 1) & 0x3FFu) | (((u32)(r.lq_t + 1) & 0x3FFu) << 10) | ((u32)estat << 20);      // -1 .. 256 each; SNK_E_* input errors are 0 .. 4
    p.sumq = r.sumq;
    return p;
}
template <class PT>
__device__ __forceinline__ void rs_unpack(const PT &P, int mate, const PackedRS &p, ReadState &r) {
    const u32 lo = (u32)p.a, hi = (u32)(p.a >> 32);
    r.len = (int)(lo & 511u);
    r.n_a = (int)((lo >> 9) & 511u);
    r.n_n = (int)((lo >> 18) & 511u);
    r.lowq = (int)((lo >> 27) | ((hi & 15u) << 5));
    r.clen = (int)((hi >> 4) & 511u);
    r.start = (int)((hi >> 13) & 511u);
    r.adacut = (int)((hi >> 22) & 511u) - 1;
    r.polyx = (int)(hi >> 31);
    r.inc_ada = r.adacut >= 0 ? 1 : 0;
    r.lq_h = (int)(p.lq & 0x3FFu) - 1;
    r.lq_t = (int)((p.lq >> 10) & 0x3FFu) - 1;
    r.sumq = p.sumq;
    const bool hard = P.trim_on && P.has_hard;                         // trim_finish(), snk_common.hip.h
    r.hd_h = hard ? P.hard[P.paired ? 2 * mate : 0] : -1;
    r.hd_t = hard ? P.hard[P.paired ? 2 * mate + 1 : 1] : -1;
}

__device__ __forceinline__ int rs_estat(const PackedRS &p) { return (int)(p.lq >> 20); }
__device__ __forceinline__ void rs_park(snk_read_result *out, long i, const PackedRS &p) {
    *reinterpret_cast<uint4 *>(out + i) = make_uint4((u32)p.a, (u32)(p.a >> 32), p.lq, (u32)p.sumq);
}
__device__ __forceinline__ PackedRS rs_fetch(const snk_read_result *out, long i) {
    const uint4 v = *reinterpret_cast<const uint4 *>(out + i);
    PackedRS p;
    p.a = ((u64)v.y << 32) | v.x;
    p.lq = v.z;
    p.sumq = (int)v.w;
    return p;
}

struct TileGeom {
    int lcap, nq, Lh, lg, WB, WQ, SET;   // Lh = 1 << lg dwords per histogram bin row
    int pairq;                           // 129..160 positions, staged path: the raw quality rows' slots of positions 160..191 count positions 128..159 too (PAIR, phase 1)
    // LDS staging of the read bytes (global_load_lds, 16 B/lane): per wave 2 buffers x
    // {bases, qualities} x cba bytes, a chunk = rb consecutive reads.  rb == 0: disabled.
    int rb, cba, stg_off, stg_wave;
    // hand-over scratch: 2 KB per wave (8 rows of 256 B); the wave's own staging buffers when they exist (idle by then)
    int scr_off, scr_wave;
};
// LDS words behind the four histogram sets: 64 per-lane scratch words, 80 misc counters
constexpr int SNK_LDS_TAIL = 64 + 80;

// Static shape of the staged path: row pitch, bytes per staging array and reads per chunk known at compile time (the PE150 / PE250
// batches of BASELINE configs[1..4]: pitch 160 / 256).  Every LDS read of phase 1 then is `base register + immediate` -- no address
// arithmetic per read -- and the chunk bookkeeping (which read closes a chunk, which buffer a row sits in) folds away: 3 VALU,
// ~12 SALU and 3 branches per read less than with run-time values (RB == 0: run-time shape, any pitch).
template <int PITCH_, int CBA_, int RB_> struct TileShape { static constexpr int PITCH = PITCH_, CBA = CBA_, RB = RB_; };
typedef TileShape<0, 0, 0> ShapeRT;

// The kernel's argument block: ONE struct by value, so that it sits at offset 0 of the kernarg segment and the kernel can look at
// it through a constant-address-space pointer of its own (SNK_KERNARG_PTR).  Round 5: the four structs used to be four by-value
// arguments; everything loaded from them and everything derived from that is loop-invariant, so the compiler loaded and derived all
// of it at the kernel's entry and carried it through the tile loop -- 230 scalar registers spilled into VGPR lanes, 600 v_readlane
// reloads in the code (tools/isa_spills.py).  Now each region of a tile (phases 1-2 of the mates, the pair level with phase 3,
// the flush) takes a FRESH look (SNK_FRESH_ARGS: the pointer passes through an empty asm): its scalar loads and address arithmetic
// happen where they are used, on the scalar unit (which has slots to spare: the kernel is bound by VALU issue), and die there.
struct TiledArgs { DevParams P; DevBatch B; DevStats st; TileGeom G; int iters, flush_every; };
typedef __attribute__((address_space(4))) TiledArgs CTiledArgs;

// One tile = up to 64 pairs starting at t0, processed by one wave.
// The mate loops are deliberately NOT unrolled (one copy of phases 1-3 in the instruction cache);
// per-mate results are handed over in the two ReadState values r0 / r1.
template <int NW, bool FULL, bool STAGED, class SH>
__device__ __forceinline__ void process_tile(const CTiledArgs *ka, u32 *lds, long t0, int cnt) {      // (one call site per kernel instance; as a call its uniform arguments would arrive in VGPRs)
    constexpr int NS = (NW + 1) / 2;
    int lane = threadIdx.x & 63;
    asm volatile("" : "+v"(lane));   // keep per-lane address math local to the tile (no hoisting out of the tile loop)
    const bool lanev = lane < cnt;
    PackedRS pl = {0ull, 0u, 0};                      // the state of the mate the loop ran last (written by every iteration: nothing is carried)
    {   // ------------------------------------------------------------ phases 1 and 2 of the mates: their own look at the arguments
    const CTiledArgs *ka1 = ka;
    SNK_FRESH_ARGS(ka1);
    const auto &P = ka1->P;
    const auto &B = ka1->B;
    const auto &G = ka1->G;
    const int mates = P.paired ? 2 : 1;
    const int phred = P.phred, nq = G.nq, lowQ = P.low_qual;
    const int lgb = G.lg + 2;                         // log2(bytes per histogram bin row)
    const bool oobH = (0 - phred) < P.lq_head_q, oobT = (0 - phred) < P.lq_tail_q;

    const uint8_t *const seq0 = B.seq[0], *const seq1 = B.seq[1], *const qual0 = B.qual[0], *const qual1 = B.qual[1];
#pragma unroll 1
    for (int m = 0; m < mates; ++m) {
        asm volatile("" : "+v"(lane));
        const uint8_t *seq = m ? seq1 : seq0, *qual = m ? qual1 : qual0;
        int mylen = 0;
        if (lanev) mylen = B.len[m] ? (int)B.len[m][t0 + lane] : B.fixed_len[m];
        const int clen_v = min(mylen, G.lcap);
        // ------------------------------------------------------------ phase 1
        // Straight-line per strip: one LDS histogram add (quality) and four one-bit collector updates.
        // Nothing else: bits past a read's end are masked per lane in phase 2, base and low-quality counts are
        // popcounts of the planes, reads containing anything but ACGT (N, lower case, garbage) show up in the
        // "exact ACGT" plane and are repaired in a rare fix-up pass, and an out-of-range quality lands in an
        // overflow / underflow row that the flush checks.
        u32 X[4][NW], XN[NW], FG[NW], EQ[NW], LQH[NW], LQT[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) XN[j] = FG[j] = EQ[j] = LQH[j] = LQT[j] = 0;
        int v_sumq = 0;
        // Bit collectors, FOUR POSITIONS PER LANE: lane l holds bases / qualities 4l..4l+3 of the current read as one dword
        // each (one ds_read_b32), every predicate is evaluated on the four bytes at once and shifted into a packed
        // accumulator -- one bit (or bit pair) per byte and read:
        //   aC   2-bit code of the character (bits 1-2: A 00, C 01, T 10, G 11): aC = aC << 2 | code, parked every 4 reads
        //   aQ   quality >= lowQual + 1 (bit 7 of quality + 128 - threshold; no carry between valid bytes, which are
        //        < 128): aQ = aQ >> 1 | bit 7, parked every 8 reads
        //   aA / aT (FULL)  quality >= head / tail threshold of the low-quality-end trim, like aQ
        //   sbad one bit per READ, a SCALAR: some lane's dword of the read holds a character that is not exactly the letter its code stands for
        //        (v_perm picks that letter for the four bytes at once), shifted in through the carry
        // The parked dwords (PC: 16, PQ / PA / PT: 8) cross over to lane = read in the hand-over below.  Bytes past the
        // read's end are forced to 'A' first, so they raise no flag and count as code 00 (subtracted at the hand-over).
        v16u PC = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // (16-element vectors: the compiler parks into them with s_set_gpr_idx; 8-element ones get a v_cndmask chain)
        v16u PQA = PC;                     // [0,8): aQ, [8,16): aA
        v8u PTT = {0, 0, 0, 0, 0, 0, 0, 0}; // aT
        u32 aC = 0, aQ = 0, aA = 0, aT = 0;
        u64 sbad = 0;                      // read r of the tile: bit (reads walked so far - 1 - r)
        u32 aS = 0;                        // FULL, mean-quality filter: byte sums of this lane's four qualities, reads 2k (low half) and 2k+1
        const bool has_px = FULL && __builtin_amdgcn_readfirstlane(P.polyX_num) != -1;
        const bool has_lq = FULL && SNK_ABL != 17 && __builtin_amdgcn_readfirstlane(P.has_lq) != 0;
        // a side of the low-quality-end trim whose limit is 0 cuts nothing: its plane is not collected (BASELINE configs[2] has trimBadTail only)
        const bool has_lqh = has_lq && __builtin_amdgcn_readfirstlane(P.lq_head_len) > 0;
        const bool has_lqt = has_lq && __builtin_amdgcn_readfirstlane(P.lq_tail_len) > 0;
        const int has_meanq = __builtin_amdgcn_readfirstlane(P.has_meanq);
        const int len0 = rl(clen_v, 0);
        const bool fixed = __all(!lanev || clen_v == len0);
        // every read of the tile fills the whole capacity: lanes past the end fall into histogram
        // slots of positions >= lcap, which are never flushed -> no validity masking of the histogram adds
        const bool fulllen = fixed && len0 == G.lcap && G.lcap >= 4;
        const u32 rawBw = (u32)((m * 2 + 0) * G.SET), rawQw = rawBw + (u32)G.WB;
        const u32 lds0 = SNK_LDS_ADDR(lds);                   // absolute LDS address of the histograms
        const u32 laneB = lds0 + (rawBw + (u32)lane) * 4u;                   // base bin row 0
        // quality rows: the character itself is clamped to [phred-1, phred+nq] -> rows -1 (underflow, the spare
        // sixth base row) .. nq (overflow); the flush reports both
        const u32 laneQc = lds0 + (rawQw + (u32)lane) * 4u - ((u32)phred << lgb);
        u32 qlo_v = (u32)(phred - 1);
        asm volatile("" : "+v"(qlo_v));                                     // v_med3 takes one scalar operand only
        const u32 qhi = (u32)(phred + nq);
        // byte-parallel thresholds: bit 7 of (quality + 128 - k) <=> quality >= k, for k in [0, 128]
        auto kbytes = [](int k) { return (u32)(128 - min(max(k, 0), 128)) * 0x01010101u; };
        const u32 KQ = kbytes(phred + lowQ + 1), KA = kbytes(phred + P.lq_head_q), KT = kbytes(phred + P.lq_tail_q);
        const u32 dumB = lds0 + ((u32)(4 * G.SET) + (u32)lane) * 4u;   // per-lane scratch word (variable-length tiles)
        uint8_t *ldsb = reinterpret_cast<uint8_t *>(lds);
        const int lane4 = 4 * lane;
        auto bytemask = [&](const int len_r) -> u32 {      // bytes k of this lane's dword with 4*lane + k < len_r
            const int d = len_r - lane4;
            return d >= 4 ? 0xFFFFFFFFu : (d <= 0 ? 0u : ((1u << (8 * d)) - 1u));
        };
        const u32 padm = bytemask(fixed ? len0 : G.lcap);                    // constant over a fixed-length tile
        // (moving the dword reads of the lanes at and past the end of a fixed-length read back onto its last four characters
        // would save the mask, but the unaligned ds_read_b32 of those lanes costs far more: 2.9 -> 3.8 ms)
        const int lenF = fixed ? len0 : G.lcap;
        constexpr bool nomask = false;
        const int l4 = lane4;                                                // byte offset of this lane's dword in a row
        // PAIR (whole tiles of full-length reads with 129..160 positions, staged path): the third strip of a read has at most
        // 32 live lanes, so ONE ds_read_u8 and ONE ds_add serve the third strips of reads r (even; lanes 0-31) and r + 1
        // (lanes 32-63, which fetch the next row): 7 instead of 8 LDS ops per read.  The add is the ordinary one -- lanes
        // 32-63 land in the slots of positions 160..191, which such a batch does not have; the flush books them under
        // positions 128..159 (TileGeom::pairq).  pq: that strip as fetched with read r.
        // after an odd read: the two packed quality sums cross the wave and land in the lanes of their reads
        auto sum_flush = [&](const int r) {
            int hm = FULL ? has_meanq : 0;
            SNK_OPAQUE_S(hm);
            if (FULL && hm != 0 && SNK_ABL != 16) {
                const u32 tot = (u32)wave_sum((int)aS);
                v_sumq = wl(v_sumq, (int)(tot & 0xFFFFu), r - 1);
                v_sumq = wl(v_sumq, (int)(tot >> 16), r);
                aS = 0;
            }
        };
        auto do_read = [&](auto FL, auto JC, auto PR, const int r, u32 c4, const u32 q4, const u32 (&cq)[NS], const u32 pq) {
            constexpr bool FULLLEN = decltype(FL)::value;
            constexpr bool PAIR = decltype(PR)::value;
            constexpr int jq = decltype(JC)::v & 3;        // place of the read in its group of four
            constexpr bool odd = decltype(JC)::v & 1;
            const int len_r = FULLLEN ? G.lcap : (fixed ? len0 : rl(clen_v, r));
            {   // ---- collectors (4 positions per lane)
                u32 vm = padm;                             // bytes of this lane's dword inside the read
                if (!FULLLEN && !fixed) {                  // variable-length tile: the partial dword's mask is uniform
                    const int lr4 = len_r & ~3;
                    const u32 pm = (1u << (8 * (len_r & 3))) - 1u;
                    vm = lane4 < lr4 ? 0xFFFFFFFFu : (lane4 == lr4 ? pm : 0u);
                }
                const u32 t = c4 & vm & 0x06060606u;                           // 2 * code of every byte (A 0, C 2, T 4, G 6); past the end: A
                if (jq == 0) aC = t >> 1;                                      // read j of the group: bits 2j, 2j+1 of every byte
                else aC = (t << (2 * jq - 1)) | aC;                            // v_lshl_or
                const u32 ex = __builtin_amdgcn_perm(0x00470054u, 0x00430041u, t);   // the letter each code stands for
                const u32 df = (ex ^ c4) & vm;                                 // one v_bitop3
                // one bit per read, wave-wide: the vector compare leaves the lanes with such a byte as a scalar mask (one VALU
                // instruction), "any lane" and the shift into the tile's flag word are scalar-unit work.  (Per-lane flag words --
                // compare + select + shift-or, three VALU instructions per read, and two wave-wide OR reductions per tile-mate at the
                // hand-over -- were what this replaced in round 5.)
                sbad = (sbad << 1) | (u64)mask_nonzero(__ballot(df != 0u));
                aQ = (aQ >> 1) | ((q4 + KQ) & 0x80808080u);                   // v_add, v_lshrrev, v_and_or
                if (FULL) {
                    // Real scalar branches (the empty asm keeps the compiler from turning them back into selects: left to itself it
                    // computes both collectors for every read and SELECTS -- configs[2] trims tails only: four of its VALU
                    // instructions per read were the head collector it does not have and two selects)
                    if (has_lqh) { aA = (aA >> 1) | ((q4 + KA) & 0x80808080u); asm volatile("" : "+v"(aA)); }
                    if (has_lqt) { aT = (aT >> 1) | ((q4 + KT) & 0x80808080u); asm volatile("" : "+v"(aT)); }
                    int hm = has_meanq;                // (the mean-quality filter selects the FULL variant)
                    SNK_OPAQUE_S(hm);       // a plain scalar compare + branch per read (hoisted, the flag turns into lane masks)
                    if (hm != 0 && SNK_ABL != 16) {
                        // quality sum of the read (src/read_filter.cpp:299-308): v_sad_u8 adds this lane's four bytes, two reads share
                        // a dword (a read's characters sum to at most 256 * 255 < 2^16), one wave reduction per two reads; the Phred
                        // offset comes off per read in phase 2
                        const u32 qm = q4 & vm;
                        if (!odd) aS = __builtin_amdgcn_sad_u8(qm, 0u, 0u);
                        else aS = __builtin_amdgcn_sad_hi_u8(qm, 0u, aS);
                        asm volatile("" : "+v"(aS));       // (a branch, not a select)
                    }
                }
            }
            // ---- raw per-position quality histogram (src/peprocess.cpp:1182-1201), lane = position; the base histogram
            // is counted from the collected planes once per tile (hand-over)
            static_for(std::make_integer_sequence<int, NS>{}, [&](auto sc) {
                constexpr int s = decltype(sc)::v;
                if constexpr (!(PAIR && odd && s == NS - 1)) {     // (PAIR: the even read's last strip holds both reads' bytes)
                    const int pos = 64 * s + lane;
                    const u32 qb = cq[s];
                    // clamp + row address in ONE asm statement: the compiler pads every asm result that the next VALU instruction
                    // reads with an s_nop (it has to assume a dst_sel forwarding hazard), three per read
                    u32 aQa;
                    if (FULLLEN && SNK_ABL != 11) {                // ... and the fire-and-forget add behind them
                        aQa = clamp_row_addr_add<256 * (s >> 1)>(qb, qlo_v, qhi, lgb, laneQc, (s & 1) ? 0x10000u : 1u);
                    } else {
                        aQa = clamp_row_addr(qb, qlo_v, qhi, lgb, laneQc);
                        if (!FULLLEN) aQa = pos < len_r ? aQa : dumB - 256u * (s >> 1);
                        if (SNK_ABL == 11) { asm volatile("" ::"v"(aQa)); }
                        else lds_add_u32<256 * (s >> 1)>(aQa, (s & 1) ? 0x10000u : 1u);
                    }
                }
            });
            if (FULL && odd) sum_flush(r);
        };
        // a read that does not exist (last tile of a batch): keep the collectors aligned
        auto skip_read = [&](auto JC, const int r) {
            aQ >>= 1; sbad <<= 1;
            if (FULL) {
                aA >>= 1; aT >>= 1;
                if (decltype(JC)::v & 1) sum_flush(r);          // (read r - 1 may exist)
            }
        };
        // parks after reads 8o+3 and 8o+7 (uniform o); read 32 starts the second flag word
        auto park4 = [&](const int slot) {
            PC[slot] = aC;
            aC = 0;
        };
        auto park8 = [&](const int o) {
            PQA[o] = aQ;
            aQ = 0;
            if (FULL) {
                if (has_lq) { PQA[8 + o] = aA; PTT[o] = aT; aA = aT = 0; }
            }
        };
        auto run_phase1 = [&](auto FL, auto C64) {
            constexpr bool CNT64 = decltype(C64)::value;      // a whole tile: no per-read existence test
            constexpr bool FULLLEN = decltype(FL)::value;
            const int nocts = CNT64 ? 8 : (cnt + 7) >> 3;
            if (STAGED) {
                // bytes arrive in LDS by DMA (16 B/lane), the next chunk of rb reads in flight
                // behind the collectors of the current one
                constexpr bool SS = SH::RB != 0;              // static shape: pitch, staging size and chunk length are compile-time constants
                const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
                const int rb = SS ? SH::RB : G.rb, cba = SS ? SH::CBA : G.cba, pitch = SS ? SH::PITCH : B.pitch;
                const int stg_wave = SS ? 4 * SH::CBA : G.stg_wave;
                uint8_t *stg = ldsb + G.stg_off + wave * stg_wave;
                const int nchunks = (SS && CNT64) ? 64 / (SS ? SH::RB : 1) : (cnt + rb - 1) / rb;
                // chunk sources advance by scalar adds; the lane offset is the same for every full chunk
                const int chunkB = rb * pitch;
                // (ablation 15: every wave streams one of 64 tiles over and over -> the DMA hits L2)
                const long t0s = (SNK_ABL == 15 || SNK_L2SRC) ? (long)((blockIdx.x * 16 + wave) & 63) * 64 : t0;
                const uint8_t *gs = seq + t0s * (long)pitch, *gq = qual + t0s * (long)pitch;
                const int offF = min(lane * 16, chunkB - 16);
                const bool dlane = lane * 16 < cba;
                auto issue = [&](const int k) {
                    uint8_t *dst = stg + (k & 1) * 2 * cba;
                    int off = offF;
                    if (!(SS && CNT64) && (k + 1) * rb > cnt) off = min(lane * 16, (cnt - k * rb) * pitch - 16);     // last chunk of the last tile
                    SNK_WAVE_SYNC();                       // (every lane has its rows of the buffer's previous chunk in registers)
                    if (SNK_ABL != 12 && dlane) {          // same instruction count every chunk (counted vmcnt below)
                        dma_to_lds16(gs + (u32)off, dst);      // (scalar base + zero-extended lane offset: the saddr form, no 64-bit add per lane)
                        dma_to_lds16(gq + (u32)off, dst + cba);
                    }
                    gs += chunkB;
                    gq += chunkB;
                };
                // LDS -> register reads run TWO reads ahead of the collectors, across chunk boundaries: the read
                // queue is never drained inside a tile (four register sets rotate, no moves).  Reads, histogram adds and
                // their waits are hand-placed asm: the compiler cannot see through the loop-carried LDS queue and would
                // drain it (lgkmcnt(0)) right after issuing the prefetch.  LDS ops return in order, so "all but the newest
                // K" is exact: behind the row of read r+1 come the row of read r+2 (2 + NS reads) and the NS histogram
                // adds of read r.  Every step issues the same ops (rows past the tile's last read are staging bytes that
                // are never used).
                // DMA: when the row of the last read of chunk k has been ISSUED (two reads before it is used), the next
                // row to fetch is row 0 of chunk k+1, whose DMA is waited for there; one step later that last row sits in
                // registers, and the DMA of chunk k+2 goes into the buffer of chunk k.
                constexpr bool PAIR = (SNK_PAIR == 2 || (SNK_PAIR == 1 && SS)) && NW == 5 && FULLLEN && CNT64;
                constexpr int ROWOPS = 2 + NS;
                constexpr int K = (SNK_ABL == 11 ? 0 : NS) + ROWOPS;
                // PAIR: an even read's row fetch has all NS strips (the last one shared with the odd read behind it), an odd
                // read's one less; likewise the histogram adds
                constexpr int ROWE = 2 + NS, ROWO = 2 + NS - 1, ADDE = SNK_ABL == 11 ? 0 : NS, ADDO = SNK_ABL == 11 ? 0 : NS - 1;
                const u32 stgA = lds0 + (u32)(G.stg_off + wave * stg_wave);
                const u32 lc4 = (u32)l4, l1 = (u32)cba + (u32)lane, lq4 = (u32)cba + (u32)l4;
                // the shared strip: lanes 0-31 positions 64*(NS-1).. of this row, lanes 32-63 the same positions of the next row
                const u32 l2 = (u32)cba + (u32)(64 * (NS - 1)) + (u32)(lane & 31) + (lane >= 32 ? (u32)pitch : 0u);
                issue(0);
                if (nchunks > 1) issue(1);
                if (SNK_ABL != 14) {
                    if (nchunks > 1) vmem_wait<2>();
                    else vmem_wait<0>();
                }
                auto lds_rd = [&](auto OD, u32 &c4, u32 &q4, u32 (&q)[NS], const u32 row) {
                    lds_read_b32(c4, row + lc4);
                    lds_read_b32(q4, row + lq4);
                    const u32 a2 = row + l2;
                    if constexpr (PAIR) {
                        lds_read_qstrips<0, NS - 1>(q, row + l1);
                        if constexpr (!decltype(OD)::value) lds_read_u8(q[NS - 1], a2);
                    } else {
                        lds_read_qstrips<0, NS>(q, row + l1);
                    }
                };
                // static shape: two base registers for the whole tile (dword view / byte view of the wave's staging area), the row
                // is an immediate: buffer parity * 2 * CBA + row in chunk * PITCH (+ CBA for the qualities, + 64 s for strip s)
                u32 vC = stgA + lc4, vQ = stgA + (u32)lane;
                // PAIR: the shared last strip -- lanes 0-31 this row's positions 64 (NS - 1) + lane, lanes 32-63 the same positions of the
                // NEXT row (an even read and the odd one behind it sit in one chunk: RB is even)
                u32 vQ2 = vQ + ((PAIR && lane >= 32) ? (u32)((SS ? SH::PITCH : 0) - 32) : 0u);
                asm volatile("" : "+v"(vC), "+v"(vQ), "+v"(vQ2));
                auto lds_rd_s = [&](auto OFFC, auto OD, u32 &c4, u32 &q4, u32 (&q)[NS]) {
                    constexpr int O = decltype(OFFC)::v;
                    constexpr int CB = SS ? SH::CBA : 0;
                    lds_read_b32_at<O>(c4, vC);
                    lds_read_b32_at<O + CB>(q4, vC);
                    if constexpr (PAIR) {
                        lds_read_qstrips_at<0, NS - 1, O + CB>(q, vQ);
                        if constexpr (!decltype(OD)::value) lds_read_qstrips_at<NS - 1, NS, O + CB>(q, vQ2);
                    } else {
                        lds_read_qstrips_at<0, NS, O + CB>(q, vQ);
                    }
                };
                u32 C4[4], Q4[4], QS[4][NS];                // register set of read r: r & 3
                u32 row = stgA;                             // LDS address of the newest prefetched row (scalar)
                const int rbm = rb - 1, lgrb = 31 - __builtin_clz((unsigned)rb);     // rb is a power of two >= 2 (launch())
                if constexpr (SS) {
                    lds_rd_s(IntC<0>{}, std::false_type{}, C4[0], Q4[0], QS[0]);
                    lds_rd_s(IntC<(SS ? SH::PITCH : 0)>{}, std::true_type{}, C4[1], Q4[1], QS[1]);
                } else {
                    lds_rd(std::false_type{}, C4[0], Q4[0], QS[0], row);
                    row += (u32)pitch;                      // (row 1 is in chunk 0: rb >= 2)
                    lds_rd(std::true_type{}, C4[1], Q4[1], QS[1], row);
                }
                // The rows are written by the LDS unit BEHIND the compiler's back: to it an asm read's outputs are defined the moment
                // the read is issued.  Whatever it does with such a register before the s_waitcnt that covers it -- a copy on a loop
                // edge (phi), a spill -- moves stale bits (seen once: read 1 of a tile, whose row was still in flight at the loop
                // entry, in one build of the FULL variant).  So nothing is in flight on a loop edge: both rows have landed before
                // the loop is entered, and the last read of an octet waits for both rows fetched ahead (only its adds stay out).
                // tools/isa_lint.py checks the generated code for any use of a row register ahead of its wait.
                lds_wait<0>(C4[0], Q4[0], QS[0]);
                lds_wait<0>(C4[1], Q4[1], QS[1]);
                for (int o = 0; o < nocts; ++o) {           // reads 8o .. 8o+7
                    // which reads of this octet are the last of a chunk (rb is a power of two): one bit test per read
                    const u32 evm = rb == 2 ? 0xAAu : rb == 4 ? 0x88u : ((((8 * (o + 1)) & rbm) == 0) ? 0x80u : 0u);
                    static_for(std::make_integer_sequence<int, 8>{}, [&](auto jc) {
                        constexpr int j = decltype(jc)::v;
                        const int r = 8 * o + j;
                        if (CNT64 || r < cnt) {
                            if constexpr (SS) {
                                // RB divides 8: the chunk position of read r + 2 and its buffer are those of j + 2
                                constexpr int RBs = SS ? SH::RB : 1, R2 = j + 2;
                                constexpr bool closes = (R2 % RBs) == 0;               // read r+1 is the last of chunk k
                                const int k = (8 * o) / RBs + R2 / RBs - 1;
                                if constexpr (closes) {
                                    if (k + 1 < nchunks && SNK_ABL != 14) vmem_wait<0>();
                                }
                                constexpr int par = (R2 / RBs) & 1, rowi = R2 % RBs;
                                lds_rd_s(IntC<par * 2 * (SS ? SH::CBA : 0) + rowi * (SS ? SH::PITCH : 0)>{}, std::integral_constant<bool, (j & 1) != 0>{},
                                         C4[(j + 2) & 3], Q4[(j + 2) & 3], QS[(j + 2) & 3]);
                                do_read(FL, IntC<j>{}, std::integral_constant<bool, PAIR>{}, r, C4[j & 3], Q4[j & 3], QS[j & 3], 0u);
                                // behind the row of read r+1: the row of read r+2 and the adds of read r (both of this read's parity)
                                if constexpr (j == 7) {
                                    lds_wait<PAIR ? ADDO : K - ROWOPS>(C4[0], Q4[0], QS[0]);       // loop edge: rows of reads r+1 and r+2 both in
                                    lds_wait<PAIR ? ADDO : K - ROWOPS>(C4[1], Q4[1], QS[1]);
                                } else {
                                    lds_wait<PAIR ? ((j & 1) ? ROWO + ADDO : ROWE + ADDE) : K>(C4[(j + 1) & 3], Q4[(j + 1) & 3], QS[(j + 1) & 3]);
                                }
                                if constexpr (closes) {
                                    if (k + 2 < nchunks) issue(k + 2);                 // every row of chunk k sits in registers now
                                }
                            } else {
                            // read r+1 is the last of chunk k (never for j = 7: read 0 of an octet closes no chunk)
                            const bool closes = j < 7 && ((evm >> ((j + 1) & 7)) & 1u);
                            const int k = ((r + 2) >> lgrb) - 1;
                            if (closes) {                   // row of read r+2 = row 0 of chunk k+1
                                if (k + 1 < nchunks && SNK_ABL != 14) vmem_wait<0>();
                                row = stgA + (u32)(((k + 1) & 1) * 2 * cba);
                            } else {
                                row += (u32)pitch;
                            }
                            lds_rd(std::integral_constant<bool, (j & 1) != 0>{}, C4[(j + 2) & 3], Q4[(j + 2) & 3], QS[(j + 2) & 3], row);
                            do_read(FL, IntC<j>{}, std::integral_constant<bool, PAIR>{}, r, C4[j & 3], Q4[j & 3], QS[j & 3], QS[j & 2][NS - 1]);
                            // behind the row of read r+1: the row of read r+2 and the adds of read r (same parity)
                            if constexpr (j == 7) {          // loop edge: rows of reads r+1 and r+2 both in (see above)
                                lds_wait<PAIR ? ADDO : K - ROWOPS>(C4[0], Q4[0], QS[0]);
                                lds_wait<PAIR ? ADDO : K - ROWOPS>(C4[1], Q4[1], QS[1]);
                            } else {
                                lds_wait<PAIR ? ((j & 1) ? ROWO + ADDO : ROWE + ADDE) : K>(C4[(j + 1) & 3], Q4[(j + 1) & 3], QS[(j + 1) & 3]);
                            }
                            if (closes && k + 2 < nchunks) issue(k + 2);     // every row of chunk k sits in registers now
                            }
                        } else skip_read(IntC<j>{}, r);
                        if (j == 3) park4(2 * o);
                        if (j == 7) { park4(2 * o + 1); park8(o); }
                    });
                }
                // the two rows fetched past the last read: nothing may be in flight into registers the compiler reuses
                lds_wait<0>(C4[0], Q4[0], QS[0]);
                lds_wait<0>(C4[1], Q4[1], QS[1]);
                lds_wait<0>(C4[2], Q4[2], QS[2]);
                lds_wait<0>(C4[3], Q4[3], QS[3]);
            } else {
                // register path (pitch not a multiple of 16): loads run one read ahead
                u32 offq[NS], nc4, nq4, nqb[NS];
                const bool l4ok = l4 < B.pitch;
#pragma unroll
                for (int s = 0; s < NS; ++s) offq[s] = (u32)min(64 * s + lane, B.pitch - 1);
                auto load = [&](const long rd) {
                    const uint8_t *sp = seq + rd * (long)B.pitch, *qp = qual + rd * (long)B.pitch;
                    nc4 = l4ok ? *reinterpret_cast<const u32 *>(sp + l4) : 0u;
                    nq4 = l4ok ? *reinterpret_cast<const u32 *>(qp + l4) : 0u;
#pragma unroll
                    for (int s = 0; s < NS; ++s) nqb[s] = qp[offq[s]];
                };
                load(t0);
                for (int o = 0; o < nocts; ++o) {
                    static_for(std::make_integer_sequence<int, 8>{}, [&](auto jc) {
                        constexpr int j = decltype(jc)::v;
                        const int r = 8 * o + j;
                        if (CNT64 || r < cnt) {
                            u32 cq[NS];
                            const u32 c4 = nc4, q4 = nq4;
#pragma unroll
                            for (int s = 0; s < NS; ++s) cq[s] = nqb[s];
                            if (r + 1 < cnt) load(t0 + r + 1);
                            do_read(FL, IntC<j>{}, std::false_type{}, r, c4, q4, cq, 0u);
                        } else skip_read(IntC<j>{}, r);
                        if (j == 3) park4(2 * o);
                        if (j == 7) { park4(2 * o + 1); park8(o); }
                    });
                }
            }
        };
        // Wave priorities follow the phases (3 = phase 1 ... 0 = phase 3): the 16 waves of a CU are spread over all
        // phases, and the arbiter's default (oldest first) lets the ALU-dense phases of some waves starve the LDS
        // round trips of the waves in phase 1.  Phase 1 first, phase 3 last: 3.78 -> 3.36 ms.
        __builtin_amdgcn_s_setprio((SNK_PRIO / 1000) % 10);
        if (cnt == 64) {
            if (fulllen) run_phase1(std::true_type{}, std::true_type{});
            else run_phase1(std::false_type{}, std::true_type{});
        } else {
            run_phase1(std::false_type{}, std::false_type{});
        }
        const int walked = cnt == 64 ? 64 : 8 * ((cnt + 7) >> 3);       // reads phase 1 walked (whole octets): read r sits at bit walked - 1 - r of sbad
        __builtin_amdgcn_s_setprio((SNK_PRIO / 100) % 10);            // hand-over, planes, fix-up
        // ------------------------------------------------------------ hand-over: lane = 4 positions -> lane = read
        ReadState R;
        rs_init(R, clen_v);
        int v_adja = 0, v_nn = 0, v_bad = 0;
        u32 VP[NW], QP[NW];               // exact-ACGT positions, low-quality positions
        bool badread;
        {
            // Parked dwords go to the wave's LDS scratch, one 256-byte row each: byte p of row g = the bits of position
            // p for reads 8g..8g+7 (code planes: reads 4g..4g+3, two bits each).  A lane = position gathers its byte of
            // 8 rows into 64 read bits, and the 64 x 64 bit transposes (snk_bittr.hip.h) turn them into per-read planes.
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            // Hand-over scratch of the wave (2 KB): 8 bytes per position.  The owner of positions 4l..4l+3 turns its 8 parked
            // dwords (byte k = 8 read bits of position 4l+k) into four 8-byte rows by two 4 x 4 byte transposes and stores
            // them with two 16-byte writes; the lane = position reads its 64 read bits back with one 8-byte read.
            uint8_t *scr = ldsb + G.scr_off + wave * G.scr_wave;
            auto write8 = [&](const v8u &V) {
                u32 lo[4], hi[4];
                byte_tr4(V[0], V[1], V[2], V[3], lo);
                byte_tr4(V[4], V[5], V[6], V[7], hi);
                v4u *dst = reinterpret_cast<v4u *>(scr + 32 * lane);
                SNK_WAVE_SYNC();              // (the gathers of the rows written before are done ...
                dst[0] = v4u{lo[0], hi[0], lo[1], hi[1]};
                dst[1] = v4u{lo[2], hi[2], lo[3], hi[3]};
                SNK_WAVE_SYNC();              //  ... and these rows are there for every lane of the wave)
            };
            auto gather = [&](const int s, u32 &w0, u32 &w1) {
                const v2u v = *reinterpret_cast<const v2u *>(scr + 8 * (64 * s + lane));
                w0 = v[0];
                w1 = v[1];
            };
            // one-bit planes (8 parked dwords): gather + transpose per strip -> out[NW], lane = read, bit = position
            auto cross1 = [&](const v8u &V, u32 (&out)[NW]) {
                write8(V);
                static_for(std::make_integer_sequence<int, NS>{}, [&](auto sc) {
                    constexpr int s = decltype(sc)::v;
                    u32 w0, w1;
                    gather(s, w0, w1);
                    if (2 * s + 1 < NW) {
                        bit_transpose64(w0, w1, lane);
                        out[(2 * s + 1 < NW) ? 2 * s + 1 : 0] = w1;
                    } else {
                        w0 = bit_transpose64_lo(w0, w1, lane);
                    }
                    out[2 * s] = w0;
                });
            };
            // code planes: rows 0..7 hold reads 0..31 (matrix A), rows 8..15 reads 32..63 (matrix B), two bits per read
            u32 cw[NS][4];                 // lane = position: its 128 code bits
            {
                v8u lo8, hi8;
#pragma unroll
                for (int g = 0; g < 8; ++g) { lo8[g] = PC[g]; hi8[g] = PC[8 + g]; }
                write8(lo8);
                static_for(std::make_integer_sequence<int, NS>{}, [&](auto sc) { constexpr int s = decltype(sc)::v; gather(s, cw[s][0], cw[s][1]); });
                write8(hi8);
                static_for(std::make_integer_sequence<int, NS>{}, [&](auto sc) { constexpr int s = decltype(sc)::v; gather(s, cw[s][2], cw[s][3]); });
            }
            // raw per-position base histogram (src/peprocess.cpp:1145-1180): lane = position holds the codes of all
            // reads at its position, so the count of a letter is a popcount -> one LDS add per letter, strip and TILE.
            // LDS base rows are ordered by the code (A 00, C 01, T 10, G 11, then N; the flush swaps rows 2/3 back to
            // ACGT order).  Every character is counted by its code here (N and n as G, lower case as its letter, bytes
            // past a read's end as the 'A' they were replaced by): A = reads covering the position - C - T - G, and
            // the fix-up pass moves the N of the reads that have any.
            if (SNK_ABL != 11) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    u32 cover = (u32)cnt;
                    if (!fulllen) {
                        // variable lengths: bit r of lane p = position p lies inside read r (made here, strip by strip: five words
                        // computed ahead of the loop were spilled across the transposes)
                        u32 i0 = lanev ? lowmask32(clen_v - 64 * s) : 0u, i1 = (2 * s + 1 < NW && lanev) ? lowmask32(clen_v - 64 * s - 32) : 0u;
                        bit_transpose64(i0, i1, lane);
                        cover = __popc(i0) + __popc(i1);
                    }
                    // C = code 01, T = 10, G = 11 of a bit pair: with P = all set bits, PL = set low bits and G = pairs with both set,
                    // C = PL - G and T = (P - PL) - G: three popcounts and three logic operations per word (round 4: nine)
                    u32 pP = 0, pL = 0, nG = 0;
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 x = cw[s][w];
                        pP += __popc(x);
                        pL += __popc(x & 0x55555555u);
                        nG += __popc(x & (x >> 1) & 0x55555555u);
                    }
                    u32 nC = pL - nG, nT = pP - pL - nG;
                    if (!fulllen && nomask && 64 * s + lane >= lenF) nC = nT = nG = 0;   // (fixed-length tile: duplicates past the end)
                    const u32 nA = cover - nC - nT - nG;
                    const u32 sh = (s & 1) ? 16u : 0u;
                    u32 *rowp = lds + rawBw + 64 * (s >> 1) + lane;
                    atomicAdd(rowp, nA << sh);
                    atomicAdd(rowp + (1u << G.lg), nC << sh);
                    atomicAdd(rowp + (2u << G.lg), nT << sh);
                    atomicAdd(rowp + (3u << G.lg), nG << sh);
                }
            }
            // transposes: lane i of matrix A / B then holds 64 positions of ONE plane of ONE read -- bit i of a code word
            // is plane (i & 1) of read i >> 1 (+ 32 for B) -- and lane = read fetches its two planes from there (ds_bpermute)
            u32 LP[NW], HP[NW];
            {
                const int rr = lane & 31;
                const int srcL = 4 * (2 * rr);                                 // byte address of the source lane
                const bool fromB = lane >= 32;
                static_for(std::make_integer_sequence<int, NS>{}, [&](auto sc) {
                    constexpr int s = decltype(sc)::v;
                    u32 a0 = cw[s][0], a1 = cw[s][1], b0 = cw[s][2], b1 = cw[s][3];
                    if (2 * s + 1 < NW) {
                        bit_transpose64(a0, a1, lane);
                        bit_transpose64(b0, b1, lane);
                        const u32 la = (u32)__builtin_amdgcn_ds_bpermute(srcL, (int)a1), lb = (u32)__builtin_amdgcn_ds_bpermute(srcL, (int)b1);
                        const u32 ha = (u32)__builtin_amdgcn_ds_bpermute(srcL + 4, (int)a1), hb = (u32)__builtin_amdgcn_ds_bpermute(srcL + 4, (int)b1);
                        LP[(2 * s + 1 < NW) ? 2 * s + 1 : 0] = fromB ? lb : la;
                        HP[(2 * s + 1 < NW) ? 2 * s + 1 : 0] = fromB ? hb : ha;
                    } else {
                        a0 = bit_transpose64_lo(a0, a1, lane);
                        b0 = bit_transpose64_lo(b0, b1, lane);
                    }
                    const u32 la = (u32)__builtin_amdgcn_ds_bpermute(srcL, (int)a0), lb = (u32)__builtin_amdgcn_ds_bpermute(srcL, (int)b0);
                    const u32 ha = (u32)__builtin_amdgcn_ds_bpermute(srcL + 4, (int)a0), hb = (u32)__builtin_amdgcn_ds_bpermute(srcL + 4, (int)b0);
                    LP[2 * s] = fromB ? lb : la;
                    HP[2 * s] = fromB ? hb : ha;
                });
            }
            v8u PQ, PA, PT;
#pragma unroll
            for (int g = 0; g < 8; ++g) { PQ[g] = PQA[g]; PA[g] = PQA[8 + g]; PT[g] = PTT[g]; }
            cross1(PQ, QP);
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                QP[j] = ~QP[j];                 // collected: quality > lowQual
                VP[j] = 0xFFFFFFFFu;            // exact ACGT everywhere, unless the read is flagged (fix-up pass)
                X[0][j] = ~(HP[j] | LP[j]);
                X[1][j] = LP[j] & ~HP[j];
                X[2][j] = HP[j] & LP[j];
                X[3][j] = HP[j] & ~LP[j];
            }
            // reads with a character that is not exactly A/C/G/T: flag word bit 31 - (r & 31), OR over the lanes
            badread = lane < walked && ((sbad >> (walked - 1 - lane)) & 1ull) != 0;
            if (FULL) {
                if (has_px) {
                    // "same character as the previous position" (polyX) from the planes: same code bits and both exact
                    // ACGT; reads with anything else get their exact plane in the fix-up pass
#pragma unroll
                    for (int j = 0; j < NW; ++j) {
                        const u32 lp = (LP[j] << 1) | (j ? LP[j ? j - 1 : 0] >> 31 : 0u), hp = (HP[j] << 1) | (j ? HP[j ? j - 1 : 0] >> 31 : 0u);
                        EQ[j] = ~(LP[j] ^ lp) & ~(HP[j] ^ hp) & (j ? 0xFFFFFFFFu : 0xFFFFFFFEu);
                    }
                }
                if (has_lqh) {
                    cross1(PA, LQH);
#pragma unroll
                    for (int j = 0; j < NW; ++j) LQH[j] = ~LQH[j];                          // collected: quality >= threshold
                }
                if (has_lqt) {
                    cross1(PT, LQT);
#pragma unroll
                    for (int j = 0; j < NW; ++j) LQT[j] = ~LQT[j];
                }
            }
            lds_wait_all();      // the scratch may be the next mate's DMA target
        }
        if ((SNK_ABL == 1 || SNK_ABL >= 11)) {
            if (SNK_ABL == 14) vmem_wait<0>();
#pragma unroll
            for (int j = 0; j < NW; ++j) asm volatile("" ::"v"(X[0][j]), "v"(X[1][j]), "v"(X[2][j]), "v"(X[3][j]), "v"(VP[j]), "v"(QP[j]));
            asm volatile("" ::"v"(v_sumq));
            continue;
        }
        // ------------------------------------------------------------ phase 2 (this mate)
#define SNK_PUT(PL, VAL)                                                                   \
    {                                                                                      \
        const u64 val_ = (VAL);                                                            \
        PL[2 * s] = wl(PL[2 * s], (int)(u32)val_, r);                                      \
        if (2 * s + 1 < NW) PL[(2 * s + 1 < NW) ? 2 * s + 1 : 0] = wl(PL[(2 * s + 1 < NW) ? 2 * s + 1 : 0], (int)(u32)(val_ >> 32), r); \
    }
        // fix-up pass (rare): reads with N / lower case / garbage.  Their exact-ACGT plane, exact counts, the N plane,
        // the folded-G plane, and the histogram move "bin of the code -> N bin".
        {
            u64 fix = __ballot(badread && lanev);
            while (fix) {
                const int r = __ffsll((long long)fix) - 1;
                fix &= fix - 1;
                const int len_r = rl(clen_v, r);
                const uint8_t *sp = seq + (t0 + r) * (long)B.pitch;
                int adjA = 0, nN = 0, bad = 0;
                u32 prev_last = 0xFFFFFFFFu;
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int pos = 64 * s + lane;
                    const bool valid = pos < len_r;
                    u32 c = 0;
                    if (valid) c = sp[pos];
                    if (FULL && has_px) {
                        // character of the previous position: wave_shr:1, lane 0 keeps `old` = last of the previous strip
                        const u32 pc = (u32)__builtin_amdgcn_update_dpp((int)prev_last, (int)c, 0x138, 0xF, 0xF, false);
                        prev_last = (u32)rl((int)c, 63);
                        SNK_PUT(EQ, __ballot(valid && c == pc))
                    }
                    const u32 cu = c & 0xDFu;
                    const bool exact = c == 'A' || c == 'C' || c == 'G' || c == 'T';
                    SNK_PUT(VP, __ballot(!valid || exact))
                    adjA += __popcll(__ballot(valid && cu == 'A' && c != 'A'));
                    const bool isn = valid && cu == 'N';
                    nN += __popcll(__ballot(isn));
                    if (__any(valid && !(cu == 'A' || cu == 'C' || cu == 'G' || cu == 'T' || cu == 'N'))) bad = 1;
                    // the hand-over counted every character in the row of its bits 1-2: N / n sit in row 3 and belong in row 4
                    // (lower-case letters share the row of their upper-case form, as the reference's switch does)
                    if (isn) {
                        u32 *rowp = reinterpret_cast<u32 *>(ldsb + (laneB - lds0) + 256u * (s >> 1));
                        atomicSub(rowp + (3u << G.lg), (s & 1) ? 0x10000u : 1u);
                        atomicAdd(rowp + (4u << G.lg), (s & 1) ? 0x10000u : 1u);
                    }
                    if (FULL) {
                        SNK_PUT(XN, __ballot(valid && c == 'N'))
                        SNK_PUT(FG, __ballot(valid && cu == 'G'))
                    }
                }
                v_adja = wl(v_adja, adjA, r);
                v_nn = wl(v_nn, nN, r);
                v_bad = wl(v_bad, bad, r);
            }
        }
        int nlowq = 0;
        {   // mask the garbage past each read; exact planes: no bit where the character is not that letter
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const u32 in = lowmask32(R.len - 32 * j);
#pragma unroll
                for (int k = 0; k < 4; ++k) X[k][j] &= in & VP[j];
                nlowq += __popc(QP[j] & in);
                if (FULL) {
                    FG[j] = badread ? FG[j] : X[2][j];
                    EQ[j] &= in;
                    LQH[j] = (LQH[j] & in) | (oobH ? ~in : 0u);
                }
            }
        }
        const int estat = !lanev ? 0 : (mylen > G.lcap ? SNK_E_TOO_LONG : (mylen == 0 ? SNK_E_EMPTY_SEQ : (v_bad ? SNK_E_BAD_BASE : 0)));
        {   // counts from the planes, then "beyond the read = matches anything" for the adapter search
            int na = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const u32 in = lowmask32(R.len - 32 * j);
                na += __popc(X[0][j]);
#pragma unroll
                for (int k = 0; k < 4; ++k) X[k][j] |= ~in;
                if (FULL) XN[j] |= ~in;
            }
            R.n_a = na + v_adja;
            R.n_n = v_nn;
            R.lowq = nlowq;
            R.sumq = has_meanq ? v_sumq - phred * R.len : 0;      // (phase 1 summed the characters)
        }
        int hix = 0, tix = 0, polyg = 0;
        if (FULL) {
            if (SNK_ABL != 8 && P.polyX_num != -1) {        // contig_base >= polyX_num  <=>  run of polyX_num-1 "same as previous"
                const int need = P.polyX_num - 1;
                if (need <= 0) R.polyx = 1;
                else {
                    int have = 1;
                    while (have < need) {
                        const int stp = min(have, need - have);
                        u32 T[NW];
#pragma unroll
                        for (int j = 0; j < NW; ++j) T[j] = EQ[j];
                        shr_plane<NW>(T, stp);
#pragma unroll
                        for (int j = 0; j < NW; ++j) EQ[j] &= T[j];
                        have += stp;
                    }
                    R.polyx = any_bit(EQ) ? 1 : 0;
                }
            }
            if (SNK_ABL != 9 && P.has_lq) {                  // src/read_filter.cpp:409-424
                const int ru = run_up<NW>(LQH);
                hix = (ru >= 32 * NW && oobH) ? P.lq_head_len : min(ru, P.lq_head_len);
                hix = max(hix, 0);
                int rd = run_down<NW>(LQT, R.len);
                if (rd >= R.len && oobT) rd = 0x7FFFFFFF;           // runs off the front: reads '\0'
                tix = max(min(rd, P.lq_tail_len), 0);
            }
            if (SNK_ABL != 9 && P.has_polyG) polyg = run_down<NW>(FG, R.len);     // src/read_filter.cpp:472-482
        }
        __builtin_amdgcn_s_setprio((SNK_PRIO / 10) % 10);            // adapter search, trimming, pair level
        const bool good = lanev && !estat;
        int ada_pos = -1;
        const int nada = P.n_ada[m];
        for (int i = 0; i < (SNK_ABL == 4 ? 0 : nada); ++i) {      // src/read_filter.cpp:175-188
            const bool todo = good && ada_pos < 0;
            if (!__any(todo)) break;
            // the compact descriptors of the whole list sit in a device array read through the constant address space: scalar
            // loads, invariant for the compiler, like the kernel arguments the first four used to travel in
            const CTileAdapter &Acur = ((const CTileAdapter *)(uintptr_t)P.tile_ada)[m * P.ada_stride + i];
            const int pp = adapter_tile<NW, FULL>(Acur, P.ada[m * P.ada_stride + i], X, XN, R.len, todo,
                                            seq + (t0 + lane) * (long)B.pitch, true, true, Acur.has_lower != 0 && badread);
            if (todo && pp >= 0) ada_pos = pp;
        }
        if (ada_pos >= 0) { R.inc_ada = 1; R.adacut = R.len - ada_pos; }
        if (SNK_ABL != 10 && P.trim_on) trim_finish(P, m, R, hix, tix, polyg);
        pl = rs_pack(R, estat);
        if (m + 1 < mates && lanev) rs_park(B.out[0], t0 + lane, pl);
    }
    }

    if ((SNK_ABL == 1 || SNK_ABL >= 11)) return;
    if (SNK_ABL == 2) {
        asm volatile("" ::"v"(pl.a), "v"(pl.lq), "v"(pl.sumq));
        return;
    }
    // ---------------------------------------------------------------- pair level
    // (quality-range errors are found by the flush through the overflow bin)
    asm volatile("" : "+v"(lane));
    const CTiledArgs *ka2 = ka;                       // (its own look at the arguments, as the flush: see TiledArgs)
    SNK_FRESH_ARGS(ka2);
    const auto &P = ka2->P;
    const auto &B = ka2->B;
    const auto &st = ka2->st;
    const auto &G = ka2->G;
    const int mates = P.paired ? 2 : 1;
    const int phred = P.phred, nq = G.nq;
    const int lgb = G.lg + 2;
    const long fb = file_block(G.lcap, nq);
    const int pe = mates - 1;
    PackedRS p0 = pl, p1 = pl;
    if (pe && lanev) p0 = rs_fetch(B.out[0], t0 + lane);
    const int e0 = lanev ? rs_estat(p0) : 0, e1 = lanev ? rs_estat(p1) : 0;
    const u64 gidx = B.first_index + (u64)(t0 + lane);
    if (lanev && (e0 || (pe && e1))) report_err(st, gidx, e0 ? 0 : 1, e0 ? e0 : e1);
    const bool live = lanev && !(e0 || (pe && e1));
    int v = 0, reason = SNK_KEEP;
    if (live) {
        ReadState r0, r1;
        rs_unpack(P, 0, p0, r0);
        rs_unpack(P, 1, p1, r1);
        const int dup = B.dup ? (int)B.dup[t0 + lane] : 0;
        const int cfv = B.cf ? (int)B.cf[t0 + lane] : 0;      // contaminant verdicts (snk_contam_kernel), rare configuration
        reason = pe ? discard_reason(P, r0, r1, dup, v, cfv & 3, (cfv >> 2) & 3) : discard_reason(P, r0, r0, dup, v, cfv & 3, cfv & 3);
        store_rec(B.out[0], t0 + lane, r0, reason, v);
        if (pe) store_rec(B.out[1], t0 + lane, r1, reason, v);
    } else if (pe && lanev) {
        rs_park(B.out[0], t0 + lane, PackedRS{0ull, 0u, 0});      // (a pair with an input error gets no record: not the parked state either)
    }
    // reason counters: one LDS add per (family, tile); the workgroup flushes them (chip-wide atomics on
    // a handful of hot addresses once per tile cost a third of the kernel)
    u32 *misc = lds + 4 * G.SET + 64;                 // [0,64) filter counters, [64,68) reads_number
    u64 *maxk = reinterpret_cast<u64 *>(misc + 72);   // last-read keys of the 4 file stats
    {
        u32 *fs = misc;
        if (__any(live && reason != SNK_KEEP)) {
            // per-lane LDS adds (a dozen lanes, same-address conflicts are cheaper than walking the families)
            if (live && reason == SNK_R_DUP) atomicAdd(&fs[SNK_FS_DUP], 1u);
            if (live && reason == SNK_R_TILE) atomicAdd(&fs[SNK_FS_TILE], 1u);
            if (live && reason == SNK_R_FOV) atomicAdd(&fs[SNK_FS_FOV], 1u);
            const int fam = reason_family(reason);
            if (live && fam >= 0) {
                atomicAdd(&fs[fam], 1u);
                if (v & 1) atomicAdd(&fs[fam + 1], 1u);
                if (v & 2) atomicAdd(&fs[fam + 2], 1u);
                if (v == 3) atomicAdd(&fs[fam + 3], 1u);
            }
        }
    }
    const bool kept = live && reason == SNK_KEEP;
    const u64 liveM = __ballot(live), keptM = __ballot(kept);
    // ---------------------------------------------------------------- per mate: trimming-position
    // counters (rare), reads_number, last-read key; then phase 3: clean = raw - removed, only
    // discarded / trimmed reads are walked again
#pragma unroll 1
    for (int m = 0; m < mates; ++m) {
        asm volatile("" : "+v"(lane));
        ReadState R;
        rs_unpack(P, m, m ? p1 : p0, R);
        u64 *fraw = st.sum + SNK_FS_N + m * fb, *fcl = st.sum + SNK_FS_N + (2 + m) * fb;
        // trimming-position counters: a few dozen hot addresses -> the workgroup's private uint32 copy
        // (chip-wide atomics on them cost 4 ms per 10 M pairs with trimBadTail on); drained at the end
        u32 *wts = st.tsw + (size_t)blockIdx.x * (4 * SNK_TS_N);
        if (__any(live && (max(max(R.hd_h, R.lq_h), max(R.hd_t, R.lq_t)) > 0 || R.adacut >= 0)) && lane == 0) misc[68] = 1u;   // something to drain (the fields default to -1)
        if (SNK_ABL != 5 && live && P.copy_back)
            ts_update(wts + m * SNK_TS_N, R.hd_h, R.lq_h, R.hd_t, R.lq_t, R.adacut, (pe && m == 1) ? R.len : 0, !pe);
        if (SNK_ABL != 5 && kept)
            ts_update(wts + (2 + m) * SNK_TS_N, R.hd_h, R.lq_h, R.hd_t, R.lq_t, R.adacut, (pe && m == 1) ? R.clen : R.len, !pe);
        if (liveM) {
            const int last = 63 - __clzll((long long)liveM);
            const int ll = rl(R.len, last);
            if (lane == 0) {
                atomicAdd(&misc[64 + m], (u32)__popcll(liveM));
                atomicMax(&maxk[m], ((B.first_index + (u64)(t0 + last) + 1) << 16) | (u64)ll);
            }
        }
        if (keptM) {
            const int last = 63 - __clzll((long long)keptM);
            const int ll = rl(R.clen, last);
            if (lane == 0) {
                atomicAdd(&misc[66 + m], (u32)__popcll(keptM));
                atomicMax(&maxk[2 + m], ((B.first_index + (u64)(t0 + last) + 1) << 16) | (u64)ll);
            }
        }
        const uint8_t *seq = B.seq[m], *qual = B.qual[m];
        u32 *remB = lds + (m * 2 + 1) * G.SET, *remQ = remB + G.WB;
        u64 *gbs = fcl + SNK_GS_N, *gqs = fcl + SNK_GS_N + (long)G.lcap * 5;
        __builtin_amdgcn_s_setprio(SNK_PRIO % 10);            // phase 3
        const bool modl = live && (reason != SNK_KEEP || R.clen != R.len);
        u64 mod = __ballot(modl);
        if (SNK_ABL == 3 || !mod) continue;
        // what the walk needs of a read, packed into one register (one v_readlane per read)
        const u32 pk = (u32)R.len | ((u32)R.clen << 9) | ((u32)R.start << 18) | ((reason != SNK_KEEP ? 1u : 0u) << 27) |
                       ((R.n_n > 0 ? 1u : 0u) << 28);
        const uint8_t *seqt = seq + t0 * (long)B.pitch, *qualt = qual + t0 * (long)B.pitch;
        const u32 lds0 = SNK_LDS_ADDR(lds);
        const u32 remBa = lds0 + ((u32)((m * 2 + 1) * G.SET) + (u32)lane) * 4u, remQa = remBa + (u32)G.WB * 4u;
        const u32 dumR = lds0 + ((u32)(4 * G.SET) + (u32)lane) * 4u;
        const u32 remQc = remQa - ((u32)phred << lgb);      // clamped character -> quality row, as in phase 1
        u32 qlo_v = (u32)(phred - 1);
        asm volatile("" : "+v"(qlo_v));
        const u32 qhi = (u32)(phred + nq);
        u32 offp[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) offp[s] = (u32)min(64 * s + lane, B.pitch - 1);
        while (mod) {
            // up to 4 reads per trip: their 8*NS byte loads are all in flight before the first use
            int rr[4];
            u32 cb[4][NS], qb[4][NS];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                rr[b] = -1;
                if (mod) {
                    rr[b] = __ffsll((long long)mod) - 1;
                    mod &= mod - 1;
                    const u32 ro = (u32)rr[b] * (u32)B.pitch;
                    const uint8_t *sp = seqt + ro, *qp = qualt + ro;
#pragma unroll
                    for (int s = 0; s < NS; ++s) { cb[b][s] = sp[offp[s]]; qb[b][s] = qp[offp[s]]; }
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (rr[b] < 0) continue;
                const int r = rr[b];
                const u32 w = (u32)rl((int)pk, r);
                const int len_r = (int)(w & 511u), clen_r = (int)((w >> 9) & 511u), start_r = (int)((w >> 18) & 511u);
                const bool disc = (w >> 27) & 1u, hasn = (w >> 28) & 1u;
                const bool shifted = !disc && start_r > 0;
                const int rm_lo = (disc || shifted) ? 0 : clen_r;       // removed raw positions [rm_lo, len)
                // straight-line like phase 1: N / garbage go to the row of their bits 1-2 and are moved
                // below, an out-of-range quality goes to the overflow bin (the flush reports it)
                if (rm_lo == 0 && len_r == G.lcap) {
                    // whole read, full capacity: lanes past the end hit slots of positions >= lcap (never flushed)
                    static_for(std::make_integer_sequence<int, NS>{}, [&](auto sc) {
                        constexpr int s = decltype(sc)::v;
                        const u32 c = cb[b][s];
                        const u32 qa = clamp_row_addr(qb[b][s], qlo_v, qhi, lgb, remQc);
                        lds_add_u32<256 * (s >> 1)>(((c & 6u) << (lgb - 1)) + remBa, (s & 1) ? 0x10000u : 1u);
                        lds_add_u32<256 * (s >> 1)>(qa, (s & 1) ? 0x10000u : 1u);
                    });
                } else {
                    const u32 span = (u32)(len_r - rm_lo);
                    static_for(std::make_integer_sequence<int, NS>{}, [&](auto sc) {
                        constexpr int s = decltype(sc)::v;
                        const u32 c = cb[b][s];
                        const u32 qa = clamp_row_addr(qb[b][s], qlo_v, qhi, lgb, remQc);
                        const bool inr = (u32)(64 * s + lane - rm_lo) < span;
                        const u32 aB = inr ? ((c & 6u) << (lgb - 1)) + remBa : dumR - 256u * (s >> 1);
                        const u32 aQ = inr ? qa : dumR - 256u * (s >> 1);
                        lds_add_u32<256 * (s >> 1)>(aB, (s & 1) ? 0x10000u : 1u);
                        lds_add_u32<256 * (s >> 1)>(aQ, (s & 1) ? 0x10000u : 1u);
                    });
                }
                if (hasn || shifted) {                                   // rare
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const int pos = 64 * s + lane;
                        const u32 c = cb[b][s];
                        const int q = (int)qb[b][s] - phred;
                        const bool isn = (c & 0xDFu) == 'N';
                        if (isn && pos >= rm_lo && pos < len_r) {          // bits 1-2 of 'N' / 'n': row 3
                            atomicSub(&remB[(3u << G.lg) + 64 * (s >> 1) + lane], (s & 1) ? 0x10000u : 1u);
                            atomicAdd(&remB[(4u << G.lg) + 64 * (s >> 1) + lane], (s & 1) ? 0x10000u : 1u);
                        }
                        if (shifted && pos >= start_r && pos < start_r + clen_r && (u32)q < (u32)nq) {
                            const u32 cls = isn ? 4u : __builtin_amdgcn_ubfe(0xB4u, c & 6u, 2u);      // stats order A C G T N
                            atomicAdd(&gbs[(pos - start_r) * 5 + cls], 1ull);     // head-trimmed survivor
                            atomicAdd(&gqs[(long)(pos - start_r) * nq + q], 1ull);
                        }
                    }
                }
            }
        }
    }
}

template <int NW, bool FULL, bool STAGED, int MAXW = 16, class SH = ShapeRT>
__global__ void __launch_bounds__(MAXW * 64)
snk_tiled_kernel(const TiledArgs A) {
    // the arguments travel by value in the kernarg segment and are looked at through a constant-address-space pointer, region by
    // region (TiledArgs above): scalar loads where a value is used, nothing carried across the tile loop but this pointer
    HIP_DYNAMIC_SHARED(u32, lds)
    constexpr int NS = (NW + 1) / 2;
    const CTiledArgs *const ka = SNK_KERNARG_PTR(CTiledArgs, A);
    const int lane = threadIdx.x & 63, W = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: tile index and addresses stay scalar
    const int iters = ka->iters, flush_every = ka->flush_every;
    {
        const int nwords = 4 * ka->G.SET + SNK_LDS_TAIL;    // histograms, per-lane scratch words, misc counters
        for (int i = threadIdx.x; i < nwords; i += blockDim.x) lds[i] = 0;
    }
    __syncthreads();
    // LDS histogram word of (bin b, position p = 64*s + l) = b*Lh + 64*(s>>1) + l, half-word s&1:
    // every address/increment of a strip is lane + compile-time constants (no per-lane tables), and
    // the 64 lanes of one ds_add hit 64 consecutive dwords (conflict-free).
    const int Wc = W;
    const long GW = (long)gridDim.x * W;
    const long n_pairs = ka->B.n;
    int flush_lo = 0;
    for (int it = 0; it < iters; ++it) {
        const long tile = (long)it * GW + SNK_TILE_OF(wave);
        const long t0 = tile * 64;
        long rem = n_pairs - t0;
        const int cnt = rem >= 64 ? 64 : (rem > 0 ? (int)rem : 0);
        if (cnt > 0) process_tile<NW, FULL, STAGED, SH>(ka, lds, t0, cnt);
        if ((it + 1) % flush_every == 0 || it + 1 == iters) {
            lds_wait_all();      // the asm histogram adds
            __syncthreads();
            const CTiledArgs *kf = ka;               // the flush's own look at the arguments
            SNK_FRESH_ARGS(kf);
            const auto &P = kf->P;
            const auto &B = kf->B;
            const auto &st = kf->st;
            const auto &G = kf->G;
            const int mates = P.paired ? 2 : 1;
            const long fb = file_block(G.lcap, G.nq);
            // flush: the workgroup's histogram words are added to its own slice of DevStats::part (plain adds: nobody else touches
            // it); snk_tiled_reduce_kernel sums the slices behind this kernel (global raw += raw ; global clean += raw - removed).
            // One 64-bit atomic per word and workgroup on the same 28 k counters (6 M atomics per launch, 256 deep per address)
            // cost 0.14 of the 2.89 ms of a 10 M-pair launch.
            int ovf = 0;
            for (int m = 0; m < mates; ++m) {
                u32 *raw = lds + (m * 2 + 0) * G.SET, *remv = lds + (m * 2 + 1) * G.SET;
                uint2 *praw = reinterpret_cast<uint2 *>(st.part) + ((size_t)blockIdx.x * 4 + (m * 2 + 0)) * G.SET;
                uint2 *prem = reinterpret_cast<uint2 *>(st.part) + ((size_t)blockIdx.x * 4 + (m * 2 + 1)) * G.SET;
                for (int w = threadIdx.x; w < G.SET; w += blockDim.x) {
                    const u32 a = raw[w], b = remv[w];
                    if (a | b) {
                        int pm, bin;
                        if (w < G.WB) { bin = w >> G.lg; pm = w - (bin << G.lg); }
                        else { const int ww = w - G.WB; bin = ww >> G.lg; pm = ww - (bin << G.lg); }
                        // slots of positions >= lcap hold the spill-over of lanes past the read end: dropped
                        // (quality rows of 129..160-position batches: phase 1 counts the third strip of the odd reads in the slots of
                        // positions 160..191 -- PAIR -- which such a batch does not have)
                        const bool pq = G.pairq && (w >= G.WB || bin == 5) && pm >= 96;   // raw set only: the removed set's slots there hold spill-over
                        const u32 alo = a & 0xFFFFu, ahi = pq ? 0u : (a >> 16);
                        const int plo = 128 * (pm >> 6) + (pm & 63) - (pq ? 32 : 0), phi = plo + 64;
                        const bool lo_ok = plo < G.lcap, hi_ok = phi < G.lcap;
                        if ((w >= G.WB && bin == G.nq) || (w < G.WB && bin == 5)) {
                            if ((alo && lo_ok) || (ahi && hi_ok)) ovf = 1;       // quality outside [0,nq): overflow / underflow row
                        } else if (flush_lo == 0) {                             // the launch's first flush (mostly its only one): the slice is zero
                            if (a) praw[w] = make_uint2(a & 0xFFFFu, a >> 16);
                            if (b) prem[w] = make_uint2(b & 0xFFFFu, b >> 16);
                        } else {
                            if (a) { uint2 x = praw[w]; x.x += a & 0xFFFFu; x.y += a >> 16; praw[w] = x; }
                            if (b) { uint2 x = prem[w]; x.x += b & 0xFFFFu; x.y += b >> 16; prem[w] = x; }
                        }
                        raw[w] = 0;
                        remv[w] = 0;
                    }
                }
            }
            {   // per-workgroup scalar counters
                u32 *misc = lds + 4 * G.SET + 64;
                u64 *maxk = reinterpret_cast<u64 *>(misc + 72);
                const int i = threadIdx.x;
                if (i < 68) {
                    const u32 v = misc[i];
                    if (v) {
                        if (i < 64) atomicAdd(&st.sum[i], (u64)v);
                        else atomicAdd(&st.sum[SNK_FS_N + (i - 64) * fb + SNK_GS_READS], (u64)v);
                        misc[i] = 0;
                    }
                } else if (i < 72) {
                    const u64 v = maxk[i - 68];
                    if (v) { atomicMax(&st.maxb[i - 68], v); maxk[i - 68] = 0; }
                }
            }
            if (__syncthreads_or(ovf)) {
                // error path: some quality since the last flush was out of range -> find the reads
                // (the reference corrupts its heap here, src/peprocess.cpp:1196; we report the first)
                for (int it2 = flush_lo; it2 <= it; ++it2) {
                    const long t2 = ((long)it2 * GW + SNK_TILE_OF(wave)) * 64;
                    const long rem2 = B.n - t2;
                    const int cnt2 = rem2 >= 64 ? 64 : (rem2 > 0 ? (int)rem2 : 0);
                    for (int m = 0; m < mates; ++m)
                        for (int r = 0; r < cnt2; ++r) {
                            int len_r = B.len[m] ? (int)B.len[m][t2 + r] : B.fixed_len[m];
                            len_r = min(len_r, G.lcap);
                            const uint8_t *qp = B.qual[m] + (t2 + r) * (long)B.pitch;
                            bool bq = false;
                            for (int pos = lane; pos < len_r; pos += 64) bq = bq || (u32)((int)qp[pos] - P.phred) >= (u32)G.nq;
                            if (__any(bq) && lane == 0) report_err(st, B.first_index + (u64)(t2 + r), m, SNK_E_QUAL_RANGE);
                        }
                }
            }
            flush_lo = it + 1;
        }
    }
    // drain this workgroup's trimming-position counters into the bound stats block (only if a tile
    // touched them; indices live within lcap+1 of the five array boundaries, src/peprocess.cpp:1124-1140).
    // Its own global atomics above must have landed: L2 is the point of coherence for both, the barrier
    // orders the workgroup.
    __threadfence();
    __syncthreads();
    const CTiledArgs *kd = ka;
    SNK_FRESH_ARGS(kd);
    const auto &P = kd->P;
    const auto &st = kd->st;
    const auto &G = kd->G;
    const long fb = file_block(G.lcap, G.nq);
    if (lds[4 * G.SET + 64 + 68]) {
        u32 *wts = st.tsw + (size_t)blockIdx.x * (4 * SNK_TS_N);
        const long ts_off = SNK_GS_N + (long)G.lcap * 5 + (long)G.lcap * G.nq;
        // every index is (array boundary c*1000) + v with |v| <= the longest read / trim setting + 1
        int hw = max(max(max(P.hard[0], P.hard[1]), max(P.hard[2], P.hard[3])), max(P.lq_head_len, P.lq_tail_len));
        hw = min(500, max(hw, G.lcap) + 2);                   // 500: the windows tile the whole block
        const int win = 2 * hw;
        for (int j = threadIdx.x; j < 4 * 6 * win; j += blockDim.x) {
            const int f = j / (6 * win), r = j - f * 6 * win, c = r / win, k = c * 1000 + (r - c * win) - hw;
            if (k < 0 || k >= SNK_TS_N) continue;
            const int i = f * SNK_TS_N + k;
            if (__hip_atomic_load(&wts[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                const u32 v = atomicExch(&wts[i], 0u);
                if (v) atomicAdd(&st.sum[SNK_FS_N + f * fb + ts_off + k], (u64)v);
            }
        }
    }
}

// Behind the tiled kernel: the workgroups' flushed histogram words (DevStats::part: [workgroup][mate x {raw, removed}][word]
// [half]) summed over the workgroups -- 16 words x 16 workgroup lanes per block, an LDS tree across the lanes -- mapped to their
// counters like the flush used to and added to the bound statistics: global raw += raw, global clean += raw - removed.  The
// slices are zero again afterwards.
__global__ void __launch_bounds__(256) snk_tiled_reduce_kernel(const DevStats st, const TileGeom G, const int nwg, const int mates) {
    __shared__ u32 acc[4][16][17];
    const int wx = threadIdx.x & 15, gy = threadIdx.x >> 4;
    const int per = (G.SET + 15) / 16;
    const int m = blockIdx.x / per, w = (blockIdx.x - m * per) * 16 + wx;
    u32 alo = 0, ahi = 0, blo = 0, bhi = 0;
    if (w < G.SET) {
        for (int g = gy; g < nwg; g += 16) {
            uint2 *praw = reinterpret_cast<uint2 *>(st.part) + ((size_t)g * 4 + (m * 2 + 0)) * G.SET + w;
            uint2 *prem = reinterpret_cast<uint2 *>(st.part) + ((size_t)g * 4 + (m * 2 + 1)) * G.SET + w;
            const uint2 a = *praw, b = *prem;
            if (a.x | a.y) *praw = make_uint2(0u, 0u);
            if (b.x | b.y) *prem = make_uint2(0u, 0u);
            alo += a.x; ahi += a.y; blo += b.x; bhi += b.y;     // (a workgroup flushes before 65 536 adds: 256 of them stay below 2^32)
        }
    }
    acc[0][wx][gy] = alo; acc[1][wx][gy] = ahi; acc[2][wx][gy] = blo; acc[3][wx][gy] = bhi;
    __syncthreads();
    if (gy != 0 || w >= G.SET) return;
    u64 A0 = 0, A1 = 0, B0 = 0, B1 = 0;
    for (int k = 0; k < 16; ++k) { A0 += acc[0][wx][k]; A1 += acc[1][wx][k]; B0 += acc[2][wx][k]; B1 += acc[3][wx][k]; }
    if (!(A0 | A1 | B0 | B1)) return;
    const long fb = file_block(G.lcap, G.nq);
    u64 *fraw = st.sum + SNK_FS_N + m * fb, *fcl = st.sum + SNK_FS_N + (2 + m) * fb;
    long off;
    int pm, bin;
    if (w < G.WB) { bin = w >> G.lg; pm = w - (bin << G.lg); off = SNK_GS_N + (long)(bin ^ ((bin >> 1) & (bin < 4))); }
    else { const int ww = w - G.WB; bin = ww >> G.lg; pm = ww - (bin << G.lg); off = SNK_GS_N + (long)G.lcap * 5 + bin; }
    if ((w >= G.WB && bin == G.nq) || (w < G.WB && bin == 5)) return;     // overflow / underflow rows: the tiled kernel reported them
    const long stride = w < G.WB ? 5 : G.nq;
    // slots of positions >= lcap hold the spill-over of lanes past the read end: dropped (quality rows of 129..160-position
    // batches: phase 1 counts the third strip of the odd reads in the slots of positions 160..191 -- PAIR -- which such a batch
    // does not have; raw set only: the removed set's slots there hold spill-over)
    const bool pq = G.pairq && (w >= G.WB || bin == 5) && pm >= 96;
    if (pq) { B0 = 0; A1 = 0; B1 = 0; }
    const int plo = 128 * (pm >> 6) + (pm & 63) - (pq ? 32 : 0), phi = plo + 64;
    const bool lo_ok = plo < G.lcap, hi_ok = phi < G.lcap;
    if (A0 && lo_ok) atomicAdd(&fraw[off + plo * stride], A0);
    if (A0 != B0 && lo_ok) atomicAdd(&fcl[off + plo * stride], A0 - B0);
    if (A1 && hi_ok) atomicAdd(&fraw[off + phi * stride], A1);
    if (A1 != B1 && hi_ok) atomicAdd(&fcl[off + phi * stride], A1 - B1);
}

template <int NW, bool FULL, bool STAGED, int MAXW = 16, class SH = ShapeRT>
void go(const DevParams &hp, const TileAdapters &ta, const DevBatch &b, const DevStats &st, const TileGeom &G, int iters,
        int flush_every, unsigned wgs, int threads, size_t shmem, void *stream) {
    static bool attr_done = false;
    auto kern = snk_tiled_kernel<NW, FULL, STAGED, MAXW, SH>;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_done = true;
    }
    TiledArgs A;
    memset(&A, 0, sizeof A);
    A.P = hp; A.B = b; A.st = st; A.G = G; A.iters = iters; A.flush_every = flush_every;
    (void)ta;                                    // (the descriptors are read from DevParams::tile_ada: lists of any length)
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(threads), shmem, (hipStream_t)stream, A);
    const int mates = hp.paired ? 2 : 1, per = (G.SET + 15) / 16;
    hipLaunchKernelGGL(snk_tiled_reduce_kernel, dim3((unsigned)(mates * per)), dim3(256), 0, (hipStream_t)stream, st, G, (int)wgs, mates);
}

// reads per staging chunk: a power of two <= 32, so that read 32 (where the bit collectors are parked) opens a chunk
inline int pow2_floor(int v) {
    int r = 0;
    for (int p = 1; p <= 32 && p <= v; p <<= 1) r = p;
    return r;
}

template <int NW, bool FULL>
int launch(const DevParams &hp, const TileAdapters &ta, const DevBatch &b, const DevStats &st, TileGeom G, int n_cu,
           void *stream) {
    const size_t hist = ((size_t)2 * 2 * G.SET + SNK_LDS_TAIL) * sizeof(u32);
    // staging needs 16-byte rows; otherwise the register path is used
    const bool can_stage = (b.pitch % 16 == 0) && b.pitch <= 1024 &&
                           (((uintptr_t)b.seq[0] | (uintptr_t)b.qual[0] | (uintptr_t)b.seq[1] | (uintptr_t)b.qual[1]) % 16 == 0);
    int W = 16;
    // (12 waves of up to 168 VGPRs -- no spills in the FULL variant, 47 with 16 waves -- are slower: C3 4.11 vs 3.84 ms, C2 3.26 vs 2.89)
    constexpr int MAXW = 16;
    G.rb = 0; G.cba = 0; G.stg_off = (int)hist; G.stg_wave = 0;
    constexpr int SCR = 2048;          // hand-over scratch per wave: 8 rows of 256 B
    if (can_stage) {
        // bytes per array per buffer: one DMA of (cba/16) lanes x 16 B; the largest chunk that still
        // lets 16 waves share the CU's LDS with the histograms.  The wave's staging buffers double as its
        // hand-over scratch, so they are at least that large.
        for (G.cba = 1024; G.cba >= 512; G.cba -= 256) {
            G.rb = pow2_floor(G.cba / b.pitch);
            G.stg_wave = 2 * 2 * G.cba;
            if (G.rb >= 2 && hist + (size_t)W * G.stg_wave + 2048 <= 160 * 1024) break;
            G.rb = 0;
        }
        if (G.rb == 0) {
            G.cba = 1024; G.rb = pow2_floor(G.cba / b.pitch); G.stg_wave = 2 * 2 * G.cba;
            while (W > 4 && hist + (size_t)W * G.stg_wave + 2048 > 160 * 1024) W -= 4;
            if (hist + (size_t)W * G.stg_wave + 2048 > 160 * 1024 || G.rb < 2) { G.rb = 0; W = 16; }   // reads pair up inside a chunk
        }
    }
    if (G.rb) { G.scr_off = G.stg_off; G.scr_wave = G.stg_wave; }
    else {
        G.scr_off = (int)hist; G.scr_wave = SCR;
        while (W > 4 && hist + (size_t)W * SCR > 160 * 1024) W -= 4;
        if (hist + (size_t)W * SCR > 160 * 1024) return 0;
    }
    const size_t shmem = hist + (G.rb ? (size_t)W * G.stg_wave + 2048 : (size_t)W * SCR);   // + slack for strip / prefetch over-reads
    const long tiles = (b.n + 63) / 64;
    long wgs = (tiles + W - 1) / W;
    if (wgs > n_cu) wgs = n_cu;
    // test hooks (tests/test_gpu_parity.py::test_multi_flush_launches): fewer workgroups / an earlier flush make a small batch walk
    // the multi-iteration and the read-modify-write flush branches that otherwise need > 16.5 M pairs per launch
    static const int hook_wgs = getenv("SNK_TEST_MAX_WGS") ? atoi(getenv("SNK_TEST_MAX_WGS")) : 0;
    static const int hook_flush = getenv("SNK_TEST_FLUSH_EVERY") ? atoi(getenv("SNK_TEST_FLUSH_EVERY")) : 0;
    if (hook_wgs > 0 && wgs > hook_wgs) wgs = hook_wgs;
    const long GW = wgs * W;
    const int iters = (int)((tiles + GW - 1) / GW);
    int flush_every = 65535 / (W * 64);
    if (hook_flush > 0 && hook_flush < flush_every) flush_every = hook_flush;
    // the shapes of BASELINE's configurations get their own instances (TileShape): PE150 at pitch 160, PE250 at pitch 256
    static const bool no_static = getenv("SNK_TILED_RUNTIME_SHAPE") != nullptr;       // (A/B and tests: the run-time shape for every batch)
    // the flush books the slots of positions 160..191 under 128..159 exactly when the instance that runs pairs its strips (otherwise
    // those slots hold the spill-over of lanes past the read's end)
    const bool static160 = NW == 5 && G.rb && !no_static && b.pitch == 160 && G.cba == 768 && G.rb == 4;
    G.pairq = (NW == 5 && G.rb && (SNK_PAIR == 2 || (SNK_PAIR == 1 && static160))) ? 1 : 0;
    bool launched = false;
    if constexpr (NW == 5) {
        if (G.rb && !no_static && b.pitch == 160 && G.cba == 768 && G.rb == 4) {
            go<NW, FULL, true, MAXW, TileShape<160, 768, 4>>(hp, ta, b, st, G, iters, flush_every, (unsigned)wgs, W * 64, shmem, stream);
            launched = true;
        }
    }
    if constexpr (NW == 8) {
        if (G.rb && !no_static && b.pitch == 256 && G.cba == 768 && G.rb == 2) {
            go<NW, FULL, true, MAXW, TileShape<256, 768, 2>>(hp, ta, b, st, G, iters, flush_every, (unsigned)wgs, W * 64, shmem, stream);
            launched = true;
        }
    }
    if (!launched) {
        if (G.rb) go<NW, FULL, true, MAXW>(hp, ta, b, st, G, iters, flush_every, (unsigned)wgs, W * 64, shmem, stream);
        else go<NW, FULL, false, MAXW>(hp, ta, b, st, G, iters, flush_every, (unsigned)wgs, W * 64, shmem, stream);
    }
    return 1;
}

}  // namespace

// bytes of DevStats::part: per workgroup (n_cu of them at most) 4 histogram sets x SET words x 2 halves
size_t snk_tiled_part_bytes(int lcap, int nq, int n_cu) {
    const int Lh = 64 * (((lcap + 63) / 64 + 1) / 2);
    return (size_t)n_cu * 4 * ((size_t)Lh * 6 + (size_t)Lh * (nq + 1)) * 2 * sizeof(u32);
}

int snk_launch_tiled(const DevParams &hp, const TileAdapters &ta, const DevBatch &b, const DevStats &st,
                     int lcap, int nq, int n_cu, void *stream) {
    if (!hp.tile_ok || lcap > 256 || b.n <= 0 || !st.part) return 0;
    TileGeom G;
    G.lcap = lcap;
    G.nq = nq;
    G.Lh = 64 * (((lcap + 63) / 64 + 1) / 2);       // dwords per bin row: strips pair up in one dword
    G.lg = G.Lh == 64 ? 6 : 7;
    G.WB = G.Lh * 6;                 // A C T G N + the quality underflow row (sits right below quality bin 0)
    G.WQ = G.Lh * (nq + 1);          // bin nq collects qualities >= nq
    G.SET = G.WB + G.WQ;
    if (((size_t)2 * 2 * G.SET + SNK_LDS_TAIL) * sizeof(u32) > 160 * 1024) return 0;
    G.rb = G.cba = G.stg_off = G.stg_wave = G.scr_off = G.scr_wave = 0;
    G.pairq = 0;
    const bool full = hp.need_n || hp.has_polyG || hp.polyX_num != -1 || hp.has_lq || hp.has_meanq;
    const int nw = (lcap + 31) / 32;        // dwords per bit plane
#define SNK_GO(NW_)                                                                    \
    return full ? launch<NW_, true>(hp, ta, b, st, G, n_cu, stream)                    \
                : launch<NW_, false>(hp, ta, b, st, G, n_cu, stream);
#ifdef SNK_ONLY_NW
    if (nw != SNK_ONLY_NW) return 0;
    SNK_GO(SNK_ONLY_NW)
#else
    if (nw <= 2) { SNK_GO(2) }
    else if (nw <= 4) { SNK_GO(4) }
    else if (nw <= 5) { SNK_GO(5) }
    else if (nw <= 6) { SNK_GO(6) }
    else { SNK_GO(8) }
#endif
#undef SNK_GO
}

// ---- include/snk_selftest.h: the hand-over primitive on its own
namespace {
__global__ void bittr_selftest_kernel(const u32 *in, u32 *out, u32 *out_lo) {
    const int lane = threadIdx.x & 63;
    const size_t m = blockIdx.x;
    u32 lo = in[(m * 64 + lane) * 2], hi = in[(m * 64 + lane) * 2 + 1];
    out_lo[m * 64 + lane] = bit_transpose64_lo(lo, hi, lane);
    bit_transpose64(lo, hi, lane);
    out[(m * 64 + lane) * 2] = lo;
    out[(m * 64 + lane) * 2 + 1] = hi;
}
}  // namespace

int snk_launch_bittr_selftest(const unsigned *d_in, int n, unsigned *d_out, unsigned *d_out_lo) {
    hipLaunchKernelGGL(bittr_selftest_kernel, dim3(n), dim3(64), 0, 0, d_in, d_out, d_out_lo);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
