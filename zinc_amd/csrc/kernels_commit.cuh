// Commit-side kernels: RAA encode of each witness row fused with the BLAKE3 leaf
// hashes and the lowest Merkle levels; batched upper Merkle levels.
//
// Reference loops replaced (paths relative to the reference repository):
//   repeat / shuffle / accumulate x2   src/zip/code_raa.rs:89-105,142-171, src/zip/utils.rs:139-142
//   encode_rows                        src/zip/pcs/commit.rs:158-183
//   compute_leaves_hashes              src/zip/pcs/utils.rs:87-93
//   merklize_leaves_hashes             src/zip/pcs/utils.rs:95-118
//
// HBM layout:
//   evals   int64  [rows][row_len]                     (row-major witness matrix)
//   rows    u64    [rows][cw][4]  little-endian limbs   (== MultilinearZipData.rows)
//   layers  32 B   [rows][2*cw]   per-row flat tree: level k at hash offset
//                  2cw - (2cw >> k); root at 2cw-2, slot 2cw-1 is padding
//                  (== MerkleTree.layers, src/zip/pcs/utils.rs:77, + root + pad)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "blake3.cuh"

namespace zipk {

typedef __int128 i128;
typedef unsigned __int128 u128;

#ifdef ZIPK_DEBUG_STAMPS
constexpr uint32_t kStampRec = 64;
__device__ __forceinline__ unsigned long long stamp_hw_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, 0, 4) / hwreg(HW_REG_HW_ID = 4, 0, 32)
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20), hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    return (unsigned long long)xcc | ((unsigned long long)hw << 8);
}
#endif

struct CommitArgs {
    const int64_t *evals;
    const uint32_t *perm1;
    const uint32_t *perm2;
    uint64_t *rows;    // [rows][cw][4] (Int<4>), or [rows][cw][2] when compact_rows
    uint32_t compact_rows;  // 16-byte row entries (w0, w1, w2, sign): the values fit 96 bits, the rest is sign extension
    uint32_t *layers;  // 8 words per hash
    uint32_t *roots;   // [rows][8]
    uint32_t row_len;
    uint32_t cw;
    uint32_t nact;  // active threads per workgroup = cw / E
    uint32_t num_rows;          // rows of this ctx (the kernel is persistent: row = blockIdx.x + i * gridDim.x)
    uint32_t rounds_per_chunk;  // a chunk = this many consecutive rounds of gridDim.x rows ...
    uint64_t chunk_ends;        // ... or, when non-zero, bit r set = a chunk ends with round r (unequal chunks, <= 64 rounds)
    // zip_commit_open, packed openings (MODE kStorePacked): what the hinted openings read of the row entries and of
    // tree levels 0..2 is stored DENSELY, per row [values 16 B each | level-0 | level-1 | level-2 nodes 32 B each] in
    // ascending index order, instead of at its sparse natural place (where every 32 bytes read cost a 128-byte line)
    // Since round 4 both the packed blocks and the tree levels such a commit still stores in `layers` (3 and up) are
    // ROW-INTERLEAVED in groups of four rows (rows 4q .. 4q+3): member `rank` of a section sits at
    //   pk + q * 4 * pk_stride + 4 * section_offset + (rank * 4 + (row & 3)) * size        (size 16: values, 32: nodes)
    // and node slot s (= level_off + index) of row r at byte  ((q * 2cw + s) * 4 + (r & 3)) * 32  of `layers`,
    // so that what ONE opening reads of four consecutive rows at one level is ONE whole 128-byte line (the gather's
    // 32-byte reads of lines 512 KB apart were 55.7 M requests per launch at 2^24, profiles/round3_gather_pmc.md).
    uint8_t *pk;                // [ceil(num_rows / 4)][4 * pk_stride]
    uint32_t pk_stride, pk_off0, pk_off1, pk_off2;  // bytes: row stride, start of the three node sections (of ONE row)
    const uint32_t *pk_tab;     // [waves][16] words: per wave the 32 16-bit section ranks its lanes' stores start at
    uint32_t *chunk_done;       // [chunks] arrival counters, or null
    // A commit of ONE round with several workgroups per CU (2^20: four of 256 threads): 1 = as always; k > 1 = k classes
    // of workgroups (commit_class) at DIFFERENT wave priorities, so that the workgroups sharing a CU finish one after the
    // other instead of together; class c's rows are chunk c (chunk_done[c]), whose openings are gathered beside the
    // hashing of the classes still at work -- a single round has no later round for a gather to run beside.
    uint32_t classes;
    // Opening hint (zip_commit_hinted): bitmaps of what an open of the hinted columns will ever read, or null =
    // store everything.  Words: [V: cw bits, entry j is opened][N0: cw bits][N1: cw/2][N2: cw/4], N_l bit i = node i
    // of level l is the sibling of an opened path ((i ^ 1) == c >> l for an opened column c).  Levels >= 3 are
    // always stored (the in-kernel upper levels read level 3 back, and above it nearly every node is needed).
    const uint32_t *need;
    // measurement hook (zip_ctx_set_profiling): workgroup 0 stamps {s_memtime, s_memrealtime} at its start and at
    // its end into clock[0..3] -- the shader clock the chip actually held during THIS launch (the VALU roofline
    // of bench.py is priced at that clock, not at a datasheet figure)
    unsigned long long *clock;
#ifdef ZIPK_DEBUG_STAMPS
    // tools/wg_spread.py, tools/ubench_pipeline.hip: kStampRec words per workgroup (100 MHz wall clock):
    // [0] start [1] end [2] XCC id | HW_ID << 8 [3..5] time in the scan passes / hash phase / chunk ends [6] rows done
    // [8 + i] end of the workgroup's i-th row (i < kStampRec - 8)
    unsigned long long *stamps;
#endif
};

__device__ __forceinline__ i128 shfl_up_i96(i128 x, int off) {
    uint32_t d0 = (uint32_t)x, d1 = (uint32_t)((u128)x >> 32), d2 = (uint32_t)((u128)x >> 64);
    d0 = __shfl_up(d0, off, 64);
    d1 = __shfl_up(d1, off, 64);
    d2 = __shfl_up(d2, off, 64);
    const int64_t hi = (int64_t)(int32_t)d2;  // sign-extend bit 95
    return (i128)(((u128)(uint64_t)hi << 64) | ((u128)d1 << 32) | d0);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory
// counter (s_waitcnt vmcnt(0)), i.e. waits for every outstanding global STORE of the wave to be
// acknowledged; between the passes of a row only LDS data is exchanged, and waiting on the row
// and hash stores there makes the kernel sensitive to memory latency for nothing.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One DPP move of a 96-bit value: lanes without a source lane (or outside row_mask) read 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ i128 dpp_i96(i128 x) {
    const int d0 = (int)(uint32_t)x, d1 = (int)(uint32_t)((u128)x >> 32), d2 = (int)(uint32_t)((u128)x >> 64);
    const uint32_t y0 = (uint32_t)__builtin_amdgcn_update_dpp(0, d0, CTRL, ROW_MASK, 0xF, false);
    const uint32_t y1 = (uint32_t)__builtin_amdgcn_update_dpp(0, d1, CTRL, ROW_MASK, 0xF, false);
    const uint32_t y2 = (uint32_t)__builtin_amdgcn_update_dpp(0, d2, CTRL, ROW_MASK, 0xF, false);
    const int64_t hi = (int64_t)(int32_t)y2;  // sign-extend bit 95
    return (i128)(((u128)(uint64_t)hi << 64) | ((u128)y1 << 32) | y0);
}

// Exclusive prefix over the workgroup of one (<= 96-bit signed) value per thread.
// wave_tot: LDS scratch of 2 x 16 entries; `which` (0/1) selects the half, and consecutive calls must
// alternate so that ONE (LDS-only) barrier per call suffices: a half is rewritten only two calls later.
// Both levels are DPP scans (row_shr 1/2/4/8 inside the 16-lane rows, then row_bcast:15 / :31 across
// them): VALU-rate data movement instead of dependent ds_bpermute round trips and a serial loop over
// the wave totals, on the critical path between the barriers of every row.
__device__ __forceinline__ i128 block_exclusive_scan_i96(i128 total, i128 *wave_tot, int which) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    i128 x = total;
    x += dpp_i96<0x111, 0xF>(x);  // row_shr:1
    x += dpp_i96<0x112, 0xF>(x);  // row_shr:2
    x += dpp_i96<0x114, 0xF>(x);  // row_shr:4
    x += dpp_i96<0x118, 0xF>(x);  // row_shr:8
    x += dpp_i96<0x142, 0xA>(x);  // row_bcast:15 -> rows 1, 3
    x += dpp_i96<0x143, 0xC>(x);  // row_bcast:31 -> rows 2, 3
    i128 *tot = wave_tot + 16 * which;
    if (lane == 63) tot[wid] = x;
    lds_barrier();
    // every wave scans the (<= 16) wave totals in its first row of lanes and picks its predecessor's
    i128 t = lane < 16 ? tot[lane] : (i128)0;
    t += dpp_i96<0x111, 0xF>(t);
    t += dpp_i96<0x112, 0xF>(t);
    t += dpp_i96<0x114, 0xF>(t);
    t += dpp_i96<0x118, 0xF>(t);
    i128 base = 0;
    if (wid > 0) {
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)t, wid - 1);
        const uint32_t b1 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((u128)t >> 32), wid - 1);
        const uint32_t b2 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((u128)t >> 64), wid - 1);
        base = (i128)(((u128)(uint64_t)(int64_t)(int32_t)b2 << 64) | ((u128)b1 << 32) | b0);
    }
    return base + (x - total);
}

__device__ __forceinline__ void store_hash(uint32_t *dst, const uint32_t (&h)[8]) {
#ifdef ZIPK_EXP_NOSTORE  // timing build (hipcc -DZIPK_EXP_NOSTORE -o other.so; ZIP_HIP_LIB_PATH=other.so tools/exp_scans_only.py or bench.py): the kernel without its tree stores
    if (h[0] != 0x12345678u || h[1] != 0x9ABCDEF0u) return;
#endif
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    d[0] = make_uint4(h[0], h[1], h[2], h[3]);
    d[1] = make_uint4(h[4], h[5], h[6], h[7]);
}
__device__ __forceinline__ void load_hash(const uint32_t *src, uint32_t (&h)[8]) {
    const uint4 *s = reinterpret_cast<const uint4 *>(src);
    uint4 a = s[0], b = s[1];
    h[0] = a.x; h[1] = a.y; h[2] = a.z; h[3] = a.w;
    h[4] = b.x; h[5] = b.y; h[6] = b.z; h[7] = b.w;
}

// hash offset of level k inside one tree of 2*cw slots
__device__ __forceinline__ uint32_t level_off(uint32_t cw, uint32_t k) { return 2u * cw - ((2u * cw) >> k); }

// Where the tree of `row` starts in `layers` and how many 32-bit words separate consecutive node slots: the natural
// layout ([row][2cw] hashes), or the row-interleaved one of packed commits (CommitArgs.pk: [row / 4][2cw][row % 4]).
template <bool ILV>
__device__ __forceinline__ uint32_t *tree_of(uint32_t *layers, uint32_t cw, uint32_t row) {
    return ILV ? layers + ((size_t)(row >> 2) * (2u * cw) * 4 + (row & 3u)) * 8 : layers + (size_t)row * (2u * cw) * 8;
}
template <bool ILV>
constexpr uint32_t kNodeWords = ILV ? 32u : 8u;

// Which row of a round a workgroup takes.  Workgroups b and b + 8 share an XCD (round-robin placement: observed, for
// speed only), so XCD x takes the x-th eighth of the round's rows, in order: the four rows of an interleave group
// (and the 32 rows of a gather workgroup) are then produced at the same time by workgroups that share an L2, where
// their quarter-line stores meet before they are written back.  Any bijection of [0, G) is correct.
// `classes` > 1 (CommitArgs.classes: a commit of ONE round whose workgroups share their CUs, 2^20): the l-th workgroup of
// an XCD belongs to class l / (G / 8 / classes) -- with round-robin placement the workgroups that share a CU are one of
// each class -- and class k takes the k-th part of the rows, each XCD a contiguous piece of it: a class is a chunk of
// consecutive rows that is published, and gathered, on its own (commit_class).
__device__ __forceinline__ uint32_t round_slot(uint32_t b, uint32_t G, uint32_t classes = 1) {
    if (G & 7u) return b;
    const uint32_t x = b & 7u, l = b >> 3;
    if (classes <= 1u) return x * (G >> 3) + l;
    const uint32_t per = (G >> 3) / classes, k = l / per;
    return k * (G / classes) + x * per + (l - k * per);
}
__device__ __forceinline__ uint32_t commit_class(uint32_t b, uint32_t G, uint32_t classes) {
    return classes <= 1u ? 0u : (b >> 3) / ((G >> 3) / classes);
}

// Hash of the complete subtree over the 2^LVL inputs [E0, E0 + 2^LVL) of one thread,
// written as compile-time recursion so that every hash lives in named registers
// (a runtime-indexed stack would be placed in scratch memory).  Src provides
//   leaf(e, h)            -> hash of input e (and stores it if it is a new node)
//   store(lvl, idx, h)    -> stores node `idx` of the thread's local level `lvl`
template <int LVL, int E0, class Src>
__device__ __forceinline__ void subtree_hash(Src &src, uint32_t (&h)[8]) {
    if constexpr (LVL == 0) {
        src.template leaf<E0>(h);
    } else {
        uint32_t l[8], r[8];
        subtree_hash<LVL - 1, E0>(src, l);
        subtree_hash<LVL - 1, E0 + (1 << (LVL - 1))>(src, r);
        blake3_node(l, r, h);
        src.store(LVL, E0 >> LVL, h);
    }
}

// Strided ownership for the output phase: at step e lane t owns codeword entry
// j = e*T + t, so the 32-byte row entries, leaf hashes and nodes that a wave stores in one
// instruction are adjacent in memory (4 lanes per 128-byte line instead of one lane per
// line).  Sibling leaves then sit in NEIGHBOUR LANES; the in-thread subtree becomes a
// butterfly: at level l a lane exchanges one child hash with lane t ^ 2^(l-1) and ends up
// with the node of step E0 + (t mod 2^l).  Every lane still hashes E leaves, E/2 ... 1 nodes.
// MODE 0: every entry and node is stored.  MODE 1 (zip_commit_hinted): stores no opening of the hinted columns reads
// are predicated off, at the natural places (ZIP_HIP_PACKED=0, and what hinted commits below codeword 512 ... do not
// have: they store everything).  MODE 3 (hinted commits): as 1, but what the openings read of the entries and of
// levels 0..2 goes to the packed per-row block (CommitArgs.pk).  A lane's place in a
// section is the rank of its entry / node among the section's members: the members owned by the lanes of one wave at
// one store site are consecutive in index order (per parity class for level 1, per tid mod 4 for level 2), so the
// rank is a per-wave base (pk_tab, row invariant, held in scalar registers) + the number of storing lanes below.
constexpr int kStoreAll = 0, kStoreHinted = 1, kStorePacked = 3;

// LAZY: the lane's entries are not held in registers but read from LDS when their turn comes (entry of step e at
// lz_lo / lz_hi [e * lz_stride]): the LDS planes are idle during the hash phase, and 3 E registers less are live at
// the deepest point of the butterfly.
template <int E, int MODE = kStoreAll, bool LAZY = false>
struct StridedLeaves {
    static constexpr bool MASKED = MODE != kStoreAll;
    uint32_t w0[E], w1[E], w2[E];  // 96-bit two's-complement values of the lane's E entries (!LAZY)
    const uint64_t *lz_lo = nullptr;
    const uint32_t *lz_hi = nullptr;
    uint32_t lz_stride = 0;
    template <int E0>
    __device__ __forceinline__ void entry(uint32_t &a, uint32_t &b, uint32_t &c) const {
        if (LAZY) {
            const uint64_t lo = lz_lo[E0 * lz_stride];
            a = (uint32_t)lo;
            b = (uint32_t)(lo >> 32);
            c = lz_hi[E0 * lz_stride];
        } else {
            a = w0[E0];
            b = w1[E0];
            c = w2[E0];
        }
    }
    uint64_t *out_row;
    uint32_t *tree;
    uint32_t cw, T, tid;
    uint32_t base = 0;  // first entry of the strip the butterfly covers (a multiple of E * T)
    uint32_t compact = 0;  // CommitArgs.compact_rows (wave-uniform)
    // MASKED (zip_commit_hinted): which of this lane's stores an opening of the hinted columns can ever read
    // (row invariant, see store_mask below): bit e = row entry of step e, bit 8 + e = its leaf hash, bits 16.. =
    // the lane's level-1 nodes (E0 / 2), bits 24.. = its level-2 nodes (E0 / 4).
    uint32_t smask = 0xFFFFFFFFu;
    // MODE 3: this row's lane of its group's packed block (pk_v: 16-byte values, pk_n: 32-byte nodes; the four rows
    // of a group are interleaved member by member, CommitArgs.pk), the section offsets (of the group: 4 x a row's) and
    // the wave's 32 base ranks (two per word)
    static constexpr uint32_t NW = MODE == kStorePacked ? 32u : 8u;  // words between node slots of `tree` (kNodeWords)
    uint8_t *pk_v = nullptr, *pk_n = nullptr;
    uint32_t pk_off0 = 0, pk_off1 = 0, pk_off2 = 0;
    __device__ __forceinline__ void set_packed(const CommitArgs &a, uint32_t row) {
        uint8_t *g = a.pk + (size_t)(row >> 2) * 4 * a.pk_stride;
        pk_v = g + (row & 3u) * 16u;
        pk_n = g + (row & 3u) * 32u;
        pk_off0 = 4u * a.pk_off0;
        pk_off1 = 4u * a.pk_off1;
        pk_off2 = 4u * a.pk_off2;
    }
    uint32_t ptab[16] = {};
    template <int K>
    __device__ __forceinline__ uint32_t pbase() const { return (ptab[K >> 1] >> ((K & 1) * 16)) & 0xFFFFu; }
    // number of lanes below this one among those of `m` (the lanes executing a predicated store)
    static __device__ __forceinline__ uint32_t lanes_below(uint64_t m) {
        return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    }
    template <int E0>
    __device__ __forceinline__ void store_row_of(uint32_t d0, uint32_t d1, uint32_t d2) {
        if (MASKED && !(smask & (1u << E0))) return;
        const uint32_t j = base + E0 * T + tid;
        const uint32_t s = (uint32_t)((int32_t)d2 >> 31);
#ifdef ZIPK_EXP_NOSTORE
        if (d0 != 0x12345678u || d1 != 0x9ABCDEF0u) return;
#endif
        if (MODE == kStorePacked) {
            const uint32_t pos = pbase<E0>() + lanes_below(__builtin_amdgcn_ballot_w64(true));
            *reinterpret_cast<uint4 *>(pk_v + (size_t)pos * 64) = make_uint4(d0, d1, d2, s);
        } else if (compact) {
            *reinterpret_cast<uint4 *>(out_row + (size_t)j * 2) = make_uint4(d0, d1, d2, s);
        } else {
            uint4 *o = reinterpret_cast<uint4 *>(out_row + (size_t)j * 4);
            o[0] = make_uint4(d0, d1, d2, s);  // sign extension to Int<4>
            o[1] = make_uint4(s, s, s, s);
        }
    }
    template <int E0>
    __device__ __forceinline__ void store_row() {
        uint32_t d0, d1, d2;
        entry<E0>(d0, d1, d2);
        store_row_of<E0>(d0, d1, d2);
    }
    template <int E0>
    __device__ __forceinline__ void leaf(uint32_t (&h)[8]) {
        uint32_t d0, d1, d2;
        entry<E0>(d0, d1, d2);
        store_row_of<E0>(d0, d1, d2);
        blake3_leaf_sext96(d0, d1, d2, h);
        if (!MASKED || (smask & (0x100u << E0))) {
            if (MODE == kStorePacked) {
                const uint32_t pos = pbase<8 + E0>() + lanes_below(__builtin_amdgcn_ballot_w64(true));
                store_hash(reinterpret_cast<uint32_t *>(pk_n + pk_off0) + (size_t)pos * 32, h);
            } else {
                store_hash(tree + (size_t)(base + E0 * T + tid) * 8, h);
            }
        }
    }
    // node of local level LVL computed by this lane: step E0 + (tid mod 2^LVL)
    template <int LVL, int E0>
    __device__ __forceinline__ void store(const uint32_t (&h)[8]) {
        if (MASKED && LVL <= 2 && !(smask & ((LVL == 1 ? 0x10000u : 0x1000000u) << (E0 >> LVL)))) return;
        const uint32_t e = E0 + (tid & ((1u << LVL) - 1u));
        const uint32_t n = (base + e * T + tid) >> LVL;
        if (MODE == kStorePacked && LVL == 1) {  // even lanes own the first half of the group's nodes, odd the second
            const uint64_t act = __builtin_amdgcn_ballot_w64(true);
            const bool odd = tid & 1u;
            const uint32_t pos = (odd ? pbase<16 + (E0 >> 1) * 2 + 1>() : pbase<16 + (E0 >> 1) * 2>()) +
                                 lanes_below(act & (odd ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull));
            store_hash(reinterpret_cast<uint32_t *>(pk_n + pk_off1) + (size_t)pos * 32, h);
        } else if (MODE == kStorePacked && LVL == 2) {  // tid mod 4 = q owns the q-th quarter
            const uint64_t act = __builtin_amdgcn_ballot_w64(true);
            const uint32_t q = tid & 3u;
            constexpr int K0 = 24 + ((E0 >> 2) & 1) * 4;
            const uint32_t b01 = (q & 1u) ? pbase<K0 + 1>() : pbase<K0>(), b23 = (q & 1u) ? pbase<K0 + 3>() : pbase<K0 + 2>();
            const uint32_t pos = ((q & 2u) ? b23 : b01) + lanes_below(act & (0x1111111111111111ull << q));
            store_hash(reinterpret_cast<uint32_t *>(pk_n + pk_off2) + (size_t)pos * 32, h);
        } else {
            store_hash(tree + ((size_t)level_off(cw, LVL) + n) * NW, h);
        }
    }
};

// The row-invariant store mask of one lane (StridedLeaves::smask) from the hint bitmaps of CommitArgs.need:
// the lane's entry of step e is j = base + e * T + tid, its level-l node of group E0 (a multiple of 2^l) is
// node (base + (E0 + tid mod 2^l) * T + tid) >> l.
template <int E>
__device__ __forceinline__ uint32_t store_mask(const uint32_t *need, uint32_t cw, uint32_t base, uint32_t T, uint32_t tid) {
    if (!need) return 0xFFFFFFFFu;
    const uint32_t *nv = need, *n0 = nv + (cw + 31) / 32, *n1 = n0 + (cw + 31) / 32, *n2 = n1 + (cw / 2 + 31) / 32;
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
        const uint32_t j = base + e * T + tid;
        m |= ((nv[j >> 5] >> (j & 31)) & 1u) << e;
        m |= ((n0[j >> 5] >> (j & 31)) & 1u) << (8 + e);
    }
#pragma unroll
    for (int g = 0; g < E / 2; g++) {
        const uint32_t i = (base + (2 * g + (tid & 1u)) * T + tid) >> 1;
        m |= ((n1[i >> 5] >> (i & 31)) & 1u) << (16 + g);
    }
#pragma unroll
    for (int g = 0; g < E / 4; g++) {
        const uint32_t i = (base + (4 * g + (tid & 3u)) * T + tid) >> 2;
        m |= ((n2[i >> 5] >> (i & 31)) & 1u) << (24 + g);
    }
    return m;
}

template <int E, int E0, class Src>
__device__ __forceinline__ void store_rows_only(Src &src) {
    if constexpr (E0 < E) {
        src.template store_row<E0>();
        store_rows_only<E, E0 + 1>(src);
    }
}

template <int LVL, int E0, class Src>
__device__ __forceinline__ void bfly_hash(Src &src, uint32_t (&h)[8]) {
    if constexpr (LVL == 0) {
        src.template leaf<E0>(h);
    } else {
        uint32_t A[8], B[8], m[16];
        bfly_hash<LVL - 1, E0>(src, A);
        bfly_hash<LVL - 1, E0 + (1 << (LVL - 1))>(src, B);
        const bool up = (src.tid >> (LVL - 1)) & 1;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint32_t snd = up ? A[i] : B[i];
#ifdef ZIPK_EXP_NOXCHG  // timing experiment: the butterfly without its lane exchanges (wrong trees)
            const uint32_t rcv = snd ^ 0x5A5A5A5Au;
#else
            const uint32_t rcv = __shfl_xor(snd, 1 << (LVL - 1), 64);
#endif
            m[i] = up ? rcv : A[i];      // left child
            m[8 + i] = up ? B[i] : rcv;  // right child
        }
        blake3_block64(m, h);
        src.template store<LVL, E0>(h);
    }
}

// ---- end of a chunk of rows of one workgroup of a persistent commit kernel --------------------------------------
// The butterflies leave level `first_level - 1` of every row in global memory; what remains are the levels
// first_level .. depth of the chunk's rows, the roots, and the publication of the chunk.  Round 2 did all of it at
// the chunk end, level by level with a barrier each: 47 us per chunk of four rows at 2^24 (a quarter useful work, the
// rest a chain of ten dependent compressions by ever fewer lanes, the release fence and the re-start of the row
// pipeline), 13 % of the kernel (profiles/round3_wg_spread.md); round 3 kept one stage of two levels ("the head") at
// the chunk end behind a barrier.  Now (round 4) a chunk end with a next row does NOTHING at the chunk end:
//   stages  of two levels each -- every lane hashing the complete 4-leaf subtree over four consecutive nodes (three
//           compressions, no exchange between lanes) -- AFTER the hash phase of the NEXT row, by the two oldest waves of
//           each SIMD (ChunkFinisher::after_hash): they finish a hash phase ~45 / ~30 us before the youngest and would
//           only wait at the next barrier.  The chunk is published ~60-70 us after its last row, which the consumer's
//           slack absorbs.
//   last    the LAST chunk of the kernel has no next row: one stage at once with every lane (chunk_end), then the levels
//           above it one by one through LDS (after_loop).
// Stages hand their nodes over through global memory (L2): a storing wave waits for its stores (vmcnt(0)) before the
// barrier or the LDS counter that the loading wave passes afterwards.
// Levels [lvl, lvl + nl) (nl = 1 or 2) of the rows of rounds first .. first + nrows_c - 1 of this workgroup from their
// level lvl - 1 nodes; lane `lane` of `nlanes` takes every nlanes-th group of 2^nl nodes.
template <bool ILV>
__device__ __forceinline__ void upper_stage(const CommitArgs &a, uint32_t first, uint32_t nrows_c, uint32_t lvl, uint32_t nl,
                                            uint32_t lane, uint32_t nlanes) {
    constexpr uint32_t NW = kNodeWords<ILV>;
    const uint32_t cw = a.cw, depth = 31u - __builtin_clz(cw);
    const uint32_t u_shift = depth - (lvl - 1u) - nl;  // log2(groups per row)
    const uint32_t total = nrows_c << u_shift;
    const uint32_t slot0 = round_slot(blockIdx.x, gridDim.x, a.classes);
    for (uint32_t idx = lane; idx < total; idx += nlanes) {
        const uint32_t ri = idx >> u_shift, i = idx & ((1u << u_shift) - 1u);
        const uint32_t r = slot0 + (first + ri) * gridDim.x;
        uint32_t *tree = tree_of<ILV>(a.layers, cw, r);
        const uint32_t *ch = tree + ((size_t)level_off(cw, lvl - 1u) + ((size_t)i << nl)) * NW;
        uint32_t c0[8], c1[8], h0[8];
        load_hash(ch, c0);
        load_hash(ch + NW, c1);
        if (nl == 1u) {
            blake3_node(c0, c1, h0);
            store_hash(tree + ((size_t)level_off(cw, lvl) + i) * NW, h0);
            if (lvl == depth) store_hash(a.roots + (size_t)r * 8, h0);
        } else {
            uint32_t c2[8], c3[8], h1[8], h2[8];
            blake3_node(c0, c1, h0);
            uint32_t *o = tree + ((size_t)level_off(cw, lvl) + 2u * i) * NW;
            store_hash(o, h0);
            // (compiler barrier: the second pair is loaded only now -- all four children at once are 16 more live
            // registers at the kernel's tightest point; the latency hides behind the other waves)
            asm volatile("" ::: "memory");
            load_hash(ch + 2 * NW, c2);
            load_hash(ch + 3 * NW, c3);
            blake3_node(c2, c3, h1);
            store_hash(o + NW, h1);
            blake3_node(h0, h1, h2);
            store_hash(tree + ((size_t)level_off(cw, lvl + 1u) + i) * NW, h2);
            if (lvl + 1u == depth) store_hash(a.roots + (size_t)r * 8, h2);
        }
    }
}

// The chunk schedule of a persistent commit kernel, tracked per workgroup in scalar registers.
struct ChunkCursor {
    uint32_t first = 0, index = 0;  // first round and number of the chunk in progress
    __device__ __forceinline__ bool ends_with(const CommitArgs &a, uint32_t round, bool last) const {
        if (last) return true;
        if (a.chunk_ends) return (a.chunk_ends >> round) & 1u;
        return (round + 1) % a.rounds_per_chunk == 0;
    }
    __device__ __forceinline__ void advance(uint32_t round) {
        first = round + 1;
        index++;
    }
};

// LDS words a commit kernel reserves behind its own buffers for ChunkFinisher (one counter per deferred stage)
constexpr uint32_t kFinisherFlagWords = 16;

template <bool HASH, bool ILV = false>
struct ChunkFinisher {
    // what the workgroup still owes of a finished chunk (wave-uniform): the levels from p_lvl up and the publication
    bool pending = false;
    uint32_t p_first = 0, p_nrows = 0, p_lvl = 0, p_index = 0;
    uint32_t p_base = 0;  // the level the deferred stages of every chunk but the last one start at
    uint32_t n_deferred = 0;  // chunks finished the deferred way so far (the flag counters count them)
    uint32_t *flags = nullptr;  // LDS, kFinisherFlagWords zeroed words

    __device__ __forceinline__ void init(uint32_t *lds_flags, uint32_t tid) {
        flags = lds_flags;
        if (tid < kFinisherFlagWords) lds_flags[tid] = 0;  // (every row has barriers long before the first use)
    }

    // top of a row iteration, every wave: its stores of a chunk that is still owed are in L2 before the barriers of
    // this row (and before it issues this row's loads)
    __device__ __forceinline__ void top_of_row() const {
        if (HASH && pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    static __device__ __forceinline__ void publish(const CommitArgs &a, uint32_t index, uint32_t lane) {
        if (!a.chunk_done) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
#ifndef ZIPK_EXP_NOFENCE  // timing experiment only (the gather may then read stale lines)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-back has landed before the count moves
#endif
            __hip_atomic_fetch_add(&a.chunk_done[index], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // this wave's stores are in L2: count it in
    static __device__ __forceinline__ void signal(uint32_t *flag, uint32_t lane) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // a wave parks until `target` waves have counted in.  The waves it waits for never wait for it (see after_hash), and
    // they run: the spin is bounded only so that a bug ends in a trap, not in a hung GPU.
    static __device__ __forceinline__ void wait_for(uint32_t *flag, uint32_t target) {
        uint32_t spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 24)) __builtin_trap();
        }
    }

    // After the hash phase of the row that FOLLOWS a chunk end.  The oldest wave of every SIMD (waves 0..3) finishes a
    // hash phase ~45 us before the youngest (then it only waits at the next row's first barrier), and work added to a
    // hash phase issues at the SIMD's mixed rate (3.4 cycles per instruction) where a lone wave at a chunk end runs at
    // its dependent-issue rate (6).  So EVERYTHING above the butterflies' last level is paid here (round 4: the head's
    // stage too -- at the chunk end it cost a barrier that collapsed the waves' stagger and ~35 us per chunk), spread
    // over the SIMDs, two levels per stage:
    //   a stage with >= 64 groups per row   waves 0..W-1 together, a W-th of the groups each; each counts in on flag j
    //   a smaller stage                     ONE wave (round robin), once stage j-1 has counted in on flag j-1 -> flag j
    //   publication                         the wave of the last stage.
    // A waiting wave has finished its own part of the stage before, so nobody waits for a waiter.
    // The flags count up over the chunks: stage j of this chunk is complete at flag j == target(j) = what the chunks
    // before added (they all started at level p_base, the kernel's first_level) + this chunk's own waves.
    static __device__ __forceinline__ bool stage_wide(uint32_t cw, uint32_t lvl, uint32_t nl, uint32_t j) {
        return j == 0u || (cw >> (lvl - 1u + nl)) >= 64u;
    }
    static __device__ __forceinline__ uint32_t stage_waves(uint32_t cw, uint32_t depth, uint32_t lvl0, uint32_t j, uint32_t W) {
        const uint32_t lvl = lvl0 + 2u * j;
        if (lvl > depth) return 0u;
        return stage_wide(cw, lvl, depth - lvl >= 1u ? 2u : 1u, j) ? W : 1u;
    }
    __device__ __forceinline__ uint32_t target(uint32_t cw, uint32_t depth, uint32_t j, uint32_t W) const {
        return (n_deferred - 1u) * stage_waves(cw, depth, p_base, j, W) + stage_waves(cw, depth, p_lvl, j, W);
    }
    __device__ __forceinline__ void after_hash(const CommitArgs &a, uint32_t wave, uint32_t lane, uint32_t T) {
        if (!(HASH && pending)) return;
        pending = false;
        n_deferred++;
        // the finishing waves: the TWO oldest of every SIMD (waves 0..7 of a 1024-thread workgroup).  Four (round 3) leave
        // the oldest wave of a SIMD with 12 + 3 compressions per lane and chunk and the load latencies between them --
        // more than its ~45 us lead: it became the last wave of that row; twelve or sixteen take the work to the
        // youngest waves, which have no lead to spend (step 1.477 with 8 against 1.491 / 1.527 / 1.544 ms with 4 / 12 / 16).
#ifndef ZIPK_FIN_WAVES
#define ZIPK_FIN_WAVES 8u
#endif
        const uint32_t W = T >= 64u * ZIPK_FIN_WAVES ? ZIPK_FIN_WAVES : T >= 256u ? 4u : (T + 63u) / 64u;
        if (wave >= W) return;
        const uint32_t cw = a.cw, depth = 31u - __builtin_clz(cw);
        uint32_t lvl = p_lvl, j = 0, owner = 0;
        bool wide = true;
        while (lvl <= depth) {
            const uint32_t nl = depth - lvl >= 1u ? 2u : 1u;
            wide = stage_wide(cw, lvl, nl, j);
            if (wide) {
                if (j) wait_for(flags + (j - 1u), target(cw, depth, j - 1u, W));
                upper_stage<ILV>(a, p_first, p_nrows, lvl, nl, lane + 64u * wave, 64u * W);
                signal(flags + j, lane);
            } else {
                owner = j % W;
                if (wave == owner) {
                    wait_for(flags + (j - 1u), target(cw, depth, j - 1u, W));
                    upper_stage<ILV>(a, p_first, p_nrows, lvl, nl, lane, 64u);
                    signal(flags + j, lane);
                }
            }
            lvl += nl;
            j++;
        }
        if (j == 0u || wide) owner = 0u;  // (no stage at all, or the last one was everybody's: wave 0 publishes)
        if (wave == owner) {
            if (j && wide) wait_for(flags + (j - 1u), target(cw, depth, j - 1u, W));
            publish(a, p_index, lane);
        }
    }

    // behind the row loop: the rest of the LAST chunk.  Nothing is left to hide it behind and every wave is free, so the
    // levels go one by one with ALL lanes and the nodes stay in LDS between them (the row buffers are dead by now): one
    // read of the head's nodes from L2, then per level a conflict-free LDS read, one compression, an LDS write and the
    // store to the tree -- ~2 us a level where the staged hand-over of after_hash (made for ONE free wave per SIMD) pays
    // a flag, an L2 round trip and three lone-wave compressions per two levels.  `lds`: scratch of lds_bytes (16-byte
    // aligned); too small for this chunk's nodes -> the staged path.
    __device__ __forceinline__ void after_loop(const CommitArgs &a, uint32_t wave, uint32_t lane, uint32_t T, unsigned char *lds,
                                               uint32_t lds_bytes) {
        if (!(HASH && pending)) return;
        __syncthreads();  // (the head's stores are in L2: what top_of_row + a row's barriers do otherwise)
        const uint32_t cw = a.cw, depth = 31u - __builtin_clz(cw);
        const uint32_t tid = lane + 64u * wave;
        const uint32_t w_in0 = cw >> (p_lvl - 1u);           // nodes per row at the level below p_lvl
        const uint32_t n_in0 = p_nrows * w_in0;
#ifndef ZIPK_NO_LDS_TOP
        if (p_lvl <= depth && (n_in0 + n_in0 / 2u) * 32u <= lds_bytes && n_in0 >= 2u) {
            constexpr uint32_t NW = kNodeWords<ILV>;
            pending = false;
            uint4 *bufA = reinterpret_cast<uint4 *>(lds), *bufB = bufA + 2u * n_in0;  // 2 uint4 per node
            const uint32_t slot0 = round_slot(blockIdx.x, gridDim.x, a.classes);
            const uint32_t sh0 = 31u - __builtin_clz(w_in0);
            for (uint32_t idx = tid; idx < n_in0; idx += T) {
                const uint32_t ri = idx >> sh0, i = idx & (w_in0 - 1u);
                const uint32_t r = slot0 + (p_first + ri) * gridDim.x;
                const uint4 *src = reinterpret_cast<const uint4 *>(tree_of<ILV>(a.layers, cw, r) + ((size_t)level_off(cw, p_lvl - 1u) + i) * NW);
                bufA[2u * idx] = src[0];
                bufA[2u * idx + 1u] = src[1];
            }
            __syncthreads();
            uint4 *in = bufA, *out = bufB;
            for (uint32_t lvl = p_lvl; lvl <= depth; lvl++) {
                const uint32_t w = cw >> lvl, sh = 31u - __builtin_clz(w), m = p_nrows * w;
                for (uint32_t idx = tid; idx < m; idx += T) {
                    const uint32_t ri = idx >> sh, i = idx & (w - 1u);
                    const uint32_t r = slot0 + (p_first + ri) * gridDim.x;
                    const uint4 a0 = in[4u * idx], a1 = in[4u * idx + 1u], b0 = in[4u * idx + 2u], b1 = in[4u * idx + 3u];
                    const uint32_t c0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const uint32_t c1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    uint32_t h[8];
                    blake3_node(c0, c1, h);
                    store_hash(tree_of<ILV>(a.layers, cw, r) + ((size_t)level_off(cw, lvl) + i) * NW, h);
                    if (lvl == depth) store_hash(a.roots + (size_t)r * 8, h);
                    out[2u * idx] = make_uint4(h[0], h[1], h[2], h[3]);
                    out[2u * idx + 1u] = make_uint4(h[4], h[5], h[6], h[7]);
                }
                __syncthreads();  // (also drains this level's stores: the publication below needs them all in L2)
                uint4 *t = in;
                in = out;
                out = t;
            }
            if (wave == 0u) publish(a, p_index, lane);
            return;
        }
#endif
        after_hash(a, wave, lane, T);
    }

    // the row of `round` ended chunk cc (every wave calls this)
    __device__ __forceinline__ void chunk_end(const CommitArgs &a, uint32_t first_level, const ChunkCursor &cc, uint32_t round,
                                              bool last, uint32_t tid, uint32_t T) {
        const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
        if (!HASH) {  // encode_rows: nothing to hash, chunks are only ever published for a consumer of the rows
            if (a.chunk_done) {
                __syncthreads();
                if (wave == 0) publish(a, cc.index, lane);
            }
            return;
        }
        const uint32_t cw = a.cw, depth = 31u - __builtin_clz(cw);
        const uint32_t first = cc.first, nrows_c = round - first + 1;
        uint32_t lvl = first_level;
        // A chunk with a next row hands ALL its upper levels to after_hash() of that row: no barrier here, the waves keep
        // their stagger (every wave drains its stores before the next row's last scan barrier: top_of_row()).  The LAST
        // chunk (and a tree with nothing above the butterflies) pays the head's stage now, all lanes at work, and
        // after_loop() the rest.  (-DZIPK_HEAD_AT_END: every chunk the round-3 way, for A/B runs.)
#ifdef ZIPK_HEAD_AT_END
        const bool now = true;
#else
        const bool now = last || lvl > depth;
#endif
#ifndef ZIPK_EXP_NOFINISH  // timing experiment (tools/wg_spread.py): what do the upper levels cost?
        if (now) {
            __syncthreads();  // every wave's nodes of level first_level - 1 are in L2
            if (lvl <= depth) {  // the head: one stage with every lane at work
                const uint32_t nl = depth - lvl >= 1u ? 2u : 1u;
                upper_stage<ILV>(a, first, nrows_c, lvl, nl, tid, T);
                lvl += nl;
            }
        }
#else
        __syncthreads();
        lvl = depth + 1;
#endif
        if (depth == 0 && tid == 0) {  // a one-leaf tree: the root is the leaf hash
            for (uint32_t ri = 0; ri < nrows_c; ri++) {
                const uint32_t r = round_slot(blockIdx.x, gridDim.x, a.classes) + (first + ri) * gridDim.x;
                uint32_t h[8];
                load_hash(a.layers + (size_t)r * (2u * cw) * 8, h);  // (never interleaved: packed commits have depth >= 3)
                store_hash(a.roots + (size_t)r * 8, h);
            }
        }
        if (!last) p_base = lvl;  // (every chunk but the last one starts its deferred stages at the same level)
        pending = true;
        p_first = first;
        p_nrows = nrows_c;
        p_lvl = lvl;
        p_index = cc.index;
        // (the last chunk of the kernel has no next row to hide the rest behind: after_loop() pays it at once)
    }
};

__device__ __forceinline__ void stamp_clock(const CommitArgs &a, int slot) {
    if (a.clock && blockIdx.x == 0 && threadIdx.x == 0) {
        a.clock[2 * slot] = __builtin_amdgcn_s_memtime();
        a.clock[2 * slot + 1] = __builtin_amdgcn_s_memrealtime();
    }
}

// Returns 0 in a way the optimiser cannot see through.  Adding it to the per-row index
// arithmetic keeps loop-invariant code motion from hoisting ~50 registers of permutation
// indices / addresses out of the persistent row loop (which pushed the kernel to 128 VGPRs,
// i.e. a full register file and no co-residency).
__device__ __forceinline__ uint32_t opaque_zero(uint32_t dep) {
    uint32_t z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z) : "s"(dep));
    return z;
}

// Persistent commit kernel: gridDim.x workgroups (one per CU), workgroup g encodes and
// hashes rows g, g + G, g + 2G, ...  Within a row thread t owns the E consecutive codeword
// entries [t*E, t*E+E) during the two scans.  Values never exceed 64 + 2*log2(cw) + 1 <= 96
// bits (width assertion src/zip/code_raa.rs:53-72), so the scans run on i128 lanes and the
// 256-bit result is the sign extension.
// The witness row and the intermediate codeword t2 live in LDS (cw*12 + row_len*8 bytes; up to
// cw = 8192 -- cw = 16384 has its own kernel below).  In LDS, entry j sits at slot (j % E) * (T + 32/E) + j / E: the thread-contiguous writes
// of a wave are bank-conflict free, and so are the strided reads (entry e*T + t) of the
// output phase; the pi2 gather is random either way.
//
// Being resident on every CU for the whole commit, the kernel is never displaced by the
// consumers that run beside it on other streams (upper Merkle levels, column gather): those
// only ever get the wave slots / LDS this kernel leaves free.  Rows are grouped into chunks of
// `rounds_per_chunk` rounds; when a workgroup has finished its rows of a chunk it publishes
// them (agent-scope release) and bumps chunk_done[chunk]; a chunk is complete when every
// workgroup that owns rows in it has done so.
//
// Register budget: 1024 threads are 4 waves per SIMD; capping the kernel at 512/5 -> 96 VGPRs
// (second __launch_bounds__ argument = waves per SIMD) leaves a quarter of every SIMD register
// file to the consumer kernels.  At 128 VGPRs the file is full and nothing can co-reside
// (measured: a probe kernel on another stream then only runs when this kernel ends).
// MASKED = the commit carries an opening hint (CommitArgs.need): stores the hinted openings can never read are
// predicated off by a per-lane, row-invariant bit mask.  That variant needs one VGPR more than the 96 of the
// plain one and is allowed 104 (4 x 104 of the 512 per SIMD still leave room for three gather waves, and the
// gather's LDS image limits it to two workgroups per CU beside this kernel anyway).
template <int E, bool HASH, int MODE = kStoreAll>
__global__ void __launch_bounds__(1024, 4) raa_commit_kernel(CommitArgs a) {
    constexpr bool MASKED = MODE != kStoreAll;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int LOGE = (E == 1) ? 0 : (E == 2) ? 1 : (E == 4) ? 2 : 3;
    static_assert((1 << LOGE) == E && E <= 8, "E must be 1, 2, 4 or 8");

    const uint32_t tid0 = threadIdx.x, T = blockDim.x;
    constexpr uint32_t PAD = 32 / E;
    const uint32_t PS = T + PAD;  // plane stride (slots)
    const uint32_t cw = a.cw, row_len = a.row_len;
    const bool active = tid0 < a.nact;
    stamp_clock(a, 0);

    i128 *wave_tot = reinterpret_cast<i128 *>(smem);                  // 2 x 16 entries
    uint64_t *t2lo = reinterpret_cast<uint64_t *>(smem + 512);         // E planes of PS slots
    uint32_t *t2hi = reinterpret_cast<uint32_t *>(t2lo + E * PS);
    int64_t *rowbuf = reinterpret_cast<int64_t *>(t2hi + E * PS);
    // The thread's permutation indices (packed: pi1 source index | pi2 LDS slot << 16), loaded ONCE and held for the
    // whole kernel: 8 registers, 93 VGPRs in all (96 allocated: 128 of a SIMD lane's 512 stay free for the two gather
    // waves of 48 and the row combinations beside this kernel).  Round 3 re-read them every row to free those 8 registers
    // (85 VGPRs, when a gather wave took 32 and the budget was tighter) -- but vmcnt counts in order, so the first use of
    // the re-read indices waited for every store of the previous row's hash phase to be acknowledged, on the critical
    // path of every row: a timing build without the stores runs 0.095 ms faster, and with the indices in registers the
    // kernel takes 1.271 instead of 1.289 ms alone, 1.376-1.382 instead of 1.410-1.419 ms in the step, the step
    // 1.528-1.537 instead of 1.564-1.574 ms (alternated four times).  (-DZIPK_PIDX_REREAD: the round-3 form, for A/B runs.)
    // The unpacking is kept inside the row loop (`^ z`, an opaque 0): hoisted, it is 16 more registers (109).
    uint32_t pidx[E];
    auto load_pidx = [&](uint32_t t) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            uint32_t v1 = 0, v2 = 0;
            if (active) {
                v1 = a.perm1[t * E + e] & (row_len - 1);  // repeat: t0[j] = row[j mod row_len]
                const uint32_t p2 = a.perm2[t * E + e];
                v2 = (p2 & (E - 1)) * PS + (p2 >> LOGE);
            }
            pidx[e] = v1 | (v2 << 16);
        }
    };
#ifndef ZIPK_PIDX_REREAD
    load_pidx(tid0);
#endif
    // ... the lane's store mask under an opening hint ...
    const uint32_t smask = (MASKED && active) ? store_mask<E>(a.need, cw, 0u, a.nact, tid0) : 0xFFFFFFFFu;
    // ... and (packed openings) the wave's base ranks, wave-uniform
    uint32_t ptab[16] = {};
    if (MODE == kStorePacked) {
        const uint32_t *wt = a.pk_tab + (size_t)__builtin_amdgcn_readfirstlane(tid0 >> 6) * 16;
#pragma unroll
        for (int k = 0; k < 16; k++) ptab[k] = __builtin_amdgcn_readfirstlane(wt[k]);
    }
    // ... and the NEXT witness row, fetched during the scan passes of the current one (rep = 2
    // geometry: row_len == NPF * blockDim; anything else takes the direct path).
    constexpr int NPF = (E >= 2) ? E / 2 : 1;
    const bool prefetch = row_len == NPF * T;
    int64_t nxt[NPF];

#ifdef ZIPK_DEBUG_STAMPS
    unsigned long long ph_a = 0, ph_b = 0, ph_c = 0, ph_t;
    if (tid0 == 0 && a.stamps) {
        a.stamps[(size_t)kStampRec * blockIdx.x] = wall_clock64();
        a.stamps[(size_t)kStampRec * blockIdx.x + 2] = stamp_hw_id();
    }
#define ZIPK_PH(acc) do { const unsigned long long now_ = wall_clock64(); acc += now_ - ph_t; ph_t = now_; } while (0)
#else
#define ZIPK_PH(acc) do { } while (0)
#endif
    uint32_t round = 0;
    ChunkCursor cc;
    if (a.classes > 1u) {  // (one round: the class is the chunk; the earlier the class, the higher its waves' priority)
        const uint32_t c = commit_class(blockIdx.x, gridDim.x, a.classes);
        cc.index = c;
        const uint32_t prio = (a.classes - 1u - c) * 3u / (a.classes - 1u);  // 3 .. 0 over the classes
        if (prio == 3u) __builtin_amdgcn_s_setprio(3);
        else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
        else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    }
    ChunkFinisher<HASH, MODE == kStorePacked> fin;
    fin.init(reinterpret_cast<uint32_t *>(rowbuf + row_len), tid0);
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    for (uint32_t row = round_slot(blockIdx.x, gridDim.x, a.classes); row < a.num_rows; row += gridDim.x, round++) {
#ifdef ZIPK_DEBUG_STAMPS
        ph_t = wall_clock64();
#endif
        // (fin.top_of_row(): moved down, just before this row's last scan barrier -- see there)
        const uint32_t z = opaque_zero(row);
        const uint32_t tid = tid0 + z;
        const int64_t *in = a.evals + (size_t)row * row_len;
        uint64_t *out_row = a.rows + (size_t)row * cw * (a.compact_rows ? 2 : 4);

        const bool has_next = row + gridDim.x < a.num_rows;
        if (!(prefetch && round)) {
            if (round) lds_barrier();  // the previous row's LDS image has been consumed
            for (uint32_t i = tid; i < row_len; i += T) rowbuf[i] = in[i];
            lds_barrier();
        }
        if (prefetch && has_next) {  // the NEXT witness row: in flight during the two scan passes
            const int64_t *nin = a.evals + (size_t)(row + gridDim.x) * row_len;
#pragma unroll
            for (int k = 0; k < NPF; k++) nxt[k] = nin[k * T + tid];
#ifdef ZIPK_EXP_NONXT  // timing experiment: no prefetch of the next witness row (every row computes on stale data)
#pragma unroll
            for (int k = 0; k < NPF; k++) nxt[k] = (int64_t)(k + z);
#endif
        }

        i128 v[E];
#ifdef ZIPK_PIDX_REREAD
        load_pidx(tid);  // (`tid` is opaque: the loads stay inside the loop)
#endif
        // ---- pass 1: repeat + permute(pi1) + accumulate ------------------------
        if (active) {
#pragma unroll
            for (int e = 0; e < E; e++) v[e] = (i128)rowbuf[(pidx[e] ^ z) & 0xFFFFu];  // (^ opaque 0: the unpacking stays in the loop)
#pragma unroll
            for (int e = 1; e < E; e++) v[e] += v[e - 1];
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) v[e] = 0;
        }
        {
            const i128 pre = block_exclusive_scan_i96(v[E - 1], wave_tot, 0);
            if (active) {
#pragma unroll
                for (int e = 0; e < E; e++) {
                    v[e] += pre;
                    const uint32_t slot = e * PS + tid;
                    t2lo[slot] = (uint64_t)v[e];
                    t2hi[slot] = (uint32_t)((u128)v[e] >> 64);
                }
            }
        }
        lds_barrier();
        // ---- pass 2: permute(pi2) + accumulate ---------------------------------
        if (active) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t slot = (pidx[e] ^ z) >> 16;
                const uint64_t lo = t2lo[slot];
                const int64_t hi = (int64_t)(int32_t)t2hi[slot];
                v[e] = (i128)(((u128)(uint64_t)hi << 64) | lo);
            }
#pragma unroll
            for (int e = 1; e < E; e++) v[e] += v[e - 1];
        }
        {
            // the barriers inside also order the t2 reads above before the stores below
            const i128 pre = block_exclusive_scan_i96(v[E - 1], wave_tot, 1);
#pragma unroll
            for (int e = 0; e < E; e++) v[e] += pre;
        }

        // ---- transpose to strided ownership through LDS, then rows + hashes ------
        if (active) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t slot = e * PS + tid;
                t2lo[slot] = (uint64_t)v[e];
                t2hi[slot] = (uint32_t)((u128)v[e] >> 64);
            }
        }
        if (prefetch && has_next) {
            // rowbuf was last read in pass 1 (two barriers ago): stage the prefetched row now, under the
            // barrier below, so that NO barrier follows the hash phase -- the waves of a SIMD finish
            // hashing one after the other, and the early ones go straight on into the next row's pass 1
#pragma unroll
            for (int k = 0; k < NPF; k++) rowbuf[k * T + tid] = nxt[k];
        }
        // the stores of a chunk end that is still owed (its head, one row ago) must be in L2 before the oldest waves read
        // them back after THIS row's hash phase: drained here, two scan passes after they were issued, the wait is for
        // nothing -- at the top of the row (round 3) the last wave to arrive stood in it on the critical path
        fin.top_of_row();
        lds_barrier();
        ZIPK_PH(ph_a);
#ifdef ZIPK_EXP_SCANS_ONLY  // timing build (-DZIPK_EXP_SCANS_ONLY, tools/exp_scans_only.py): a row without its hash phase and chunk ends
        if (tid0 == 0 && a.roots) a.roots[row * 8] = (uint32_t)t2lo[0];
        continue;
#endif
        if (active) {
            StridedLeaves<E, MODE, true> src;
            src.out_row = out_row;
            src.compact = a.compact_rows;
            src.tree = HASH ? tree_of<MODE == kStorePacked>(a.layers, cw, row) : nullptr;
            src.cw = cw;
            src.T = a.nact;
            src.tid = tid;
            src.smask = smask;
            if (MODE == kStorePacked) {
                src.set_packed(a, row);
#pragma unroll
                for (int k = 0; k < 16; k++) src.ptab[k] = ptab[k];
            }
            {
                // entry j = e * nact + tid sits at slot (j % E) * PS + j / E; nact is a multiple of E (or E = 1), so the
                // slot of step e is the slot of step 0 + e * nact / E: read when its turn comes (StridedLeaves LAZY)
                const uint32_t slot0 = (tid & (E - 1)) * PS + (tid >> LOGE);
                src.lz_lo = t2lo + slot0;
                src.lz_hi = t2hi + slot0;
                src.lz_stride = a.nact >> LOGE;
            }
            if (HASH) {
                uint32_t top[8];
                bfly_hash<LOGE, 0>(src, top);
            } else {
                store_rows_only<E, 0>(src);
            }
        }
        ZIPK_PH(ph_b);
        // ---- wave 0: the tail of the previous chunk; then, if this row ends a chunk, its head ------
        fin.after_hash(a, wave0, tid0 & 63u, T);
        const bool last = row + gridDim.x >= a.num_rows;
        if (cc.ends_with(a, round, last)) {
            fin.chunk_end(a, LOGE + 1, cc, round, last, tid, T);
            cc.advance(round);
        }
        ZIPK_PH(ph_c);
#ifdef ZIPK_DEBUG_STAMPS
        if (tid0 == 0 && a.stamps && round < kStampRec - 8) a.stamps[(size_t)kStampRec * blockIdx.x + 8 + round] = ph_t;
#endif
    }
    // (the t2 planes: the last row's entries have been read, every wave is past its hash phase after the barrier inside)
    fin.after_loop(a, wave0, tid0 & 63u, T, smem + 512, (uint32_t)(E * PS * 12u));
    stamp_clock(a, 1);
#ifdef ZIPK_DEBUG_STAMPS
    if (tid0 == 0 && a.stamps) {
        unsigned long long *o = a.stamps + (size_t)kStampRec * blockIdx.x;
        o[1] = wall_clock64();
        o[3] = ph_a; o[4] = ph_b; o[5] = ph_c; o[6] = round;
    }
#endif
}

// ---------------------------------------------------------------------------------------
// cw = 16384 (the 2^26 geometry: row_len 8192, 1024 threads x 16 entries).  The intermediate
// codeword t2 does not fit LDS as 96-bit values (196 KB), but it does in a compact form:
//   t2[j] = P[j/16] + L[j]   (P = exclusive prefix of the thread that owns j, |L| < 2^67)
// so bits 64.. of t2[j] differ from those of P[j/16] by at most +-9.  LDS holds the low 64 bits per
// entry (planes, 128 KB), that difference as one byte per entry (16 KB) and the high word of P per
// thread (4 KB): 148.5 KB, and pass 2 gathers pi2 from LDS like the smaller geometries do -- no
// round trip of t2 through L2 and no store-draining barriers.  The witness row (64 KB) no longer
// fits beside it and is gathered from global memory (L2) at the start of pass 1.
//
// Output phase: the 196 KB of final values fit LDS even less, so there is NO workgroup-wide
// transposition here.  A thread owns 16 consecutive entries = two aligned 8-entry subtrees, and the
// eight lanes 8m..8m+7 of a wave own 16 such subtrees; an 8x8 transpose INSIDE each 8-lane group
// (three xor-shuffle stages, registers only) hands lane 8m+i, at step s, entry i of the subtree of
// lane 8m+s: sibling leaves sit in neighbour lanes, which is all the butterfly of raa_commit_kernel
// needs, and the eight lanes of a group store 128 contiguous bytes of rows / 256 of leaf hashes per
// instruction.  The row is emitted in two phases (entries 0..7 and 8..15 of every thread); the
// second half waits in the thread's own slots of the dead t2 planes (written and read back by the
// same thread: no barrier) instead of in 48 registers -- the previous variant (two 8192-entry strips
// through LDS, three more barriers per row) spilled 59 VGPRs to scratch.
// ---------------------------------------------------------------------------------------
// T threads x 16 entries: T = 1024 for cw = 16384 (one workgroup per CU), T = 512 for cw = 8192, where TWO
// workgroups fit a CU (2 x 74.8 KB of LDS): while one of them is in its scan passes or in the latency-bound top
// of its tree, the other one hashes.
constexpr size_t c16_lds_bytes(uint32_t T) {  // wave totals | t2lo [16][T+2] | t2dh [16][T+2] | ghi [T] | finisher flags
    return 512 + (size_t)16 * (T + 2) * 8 + (size_t)16 * (T + 2) + (size_t)T * 4 + 64;
}

// One stage of the 8x8 transpose over the lanes of an 8-lane group: registers r and r | 2^B are
// exchanged with lane ^ 2^B (lane bit B clear keeps x[r], set keeps x[r | 2^B]).  After B = 0, 1, 2
// lane i holds in x[s] what lane s held in x[i].
// (compile-time recursion instead of loops: with constant indices from the start the arrays become registers in
// the first SROA run; behind a not yet unrolled loop the optimiser first turns `up ? x[r] : x[r2]` into a load
// from a selected ADDRESS, and the arrays then stay in scratch memory for good)
template <int B, int P = 0>
__device__ __forceinline__ void transpose8_stage(uint32_t (&x)[8], bool up) {
    if constexpr (P < 4) {
        constexpr int r = ((P >> B) << (B + 1)) | (P & ((1 << B) - 1)), r2 = r | (1 << B);
        const uint32_t lo = x[r], hi = x[r2];
        const uint32_t rcv = __shfl_xor(up ? lo : hi, 1 << B, 64);
        x[r] = up ? rcv : lo;
        x[r2] = up ? hi : rcv;
        transpose8_stage<B, P + 1>(x, up);
    }
}
__device__ __forceinline__ void transpose8(uint32_t (&x)[8], uint32_t lane) {
    transpose8_stage<0>(x, lane & 1u);
    transpose8_stage<1>(x, (lane >> 1) & 1u);
    transpose8_stage<2>(x, (lane >> 2) & 1u);
}

template <uint32_t T, bool HASH, int MODE = kStoreAll>
__global__ void __launch_bounds__(T, 4) raa_commit16_kernel(CommitArgs a) {
    constexpr bool MASKED = MODE != kStoreAll;
    static_assert(MODE == kStoreAll || MODE == kStoreHinted || MODE == kStorePacked, "store mode");
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int E = 16;
    constexpr uint32_t PS = T + 2;  // plane stride of the 16 t2 planes
    const uint32_t tid0 = threadIdx.x;
    const uint32_t cw = a.cw, row_len = a.row_len;  // 16 T, 8 T
    stamp_clock(a, 0);

    i128 *wave_tot = reinterpret_cast<i128 *>(smem);
    uint64_t *t2lo = reinterpret_cast<uint64_t *>(smem + 512);           // [16][PS]
    int8_t *t2dh = reinterpret_cast<int8_t *>(t2lo + (size_t)E * PS);     // [16][PS]
    int32_t *ghi = reinterpret_cast<int32_t *>(t2dh + (size_t)E * PS);    // [T]: bits 64.. of P[thread]
    // second half of the outputs (phase B) waits in the dead t2 planes
    uint64_t *park_lo = reinterpret_cast<uint64_t *>(smem + 512);         // [8][PS]
    uint32_t *park_hi = reinterpret_cast<uint32_t *>(park_lo + (size_t)8 * PS);  // [8][PS]

    // row-invariant store masks of the two output phases under an opening hint
    const uint32_t smask0 = MASKED ? store_mask<8>(a.need, cw, 0u, 16u, (tid0 & ~7u) * 16u + (tid0 & 7u)) : 0xFFFFFFFFu;
    const uint32_t smask1 = MASKED ? store_mask<8>(a.need, cw, 0u, 16u, (tid0 & ~7u) * 16u + 8u + (tid0 & 7u)) : 0xFFFFFFFFu;
    // packed openings: the wave's base ranks of both output phases, wave-uniform (CommitArgs.pk_tab)
    // (two arrays and a select, not one array indexed by the phase: see transpose8_stage)
    uint32_t ptab0[16] = {}, ptab1[16] = {};
    if (MODE == kStorePacked) {
        const uint32_t *wt = a.pk_tab + (size_t)__builtin_amdgcn_readfirstlane(tid0 >> 6) * 32;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            ptab0[k] = __builtin_amdgcn_readfirstlane(wt[k]);
            ptab1[k] = __builtin_amdgcn_readfirstlane(wt[16 + k]);
        }
    }

    uint32_t round = 0;
    ChunkCursor cc;
    ChunkFinisher<HASH, MODE == kStorePacked> fin;
    fin.init(reinterpret_cast<uint32_t *>(ghi + T), tid0);
    const uint32_t wave0 = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    for (uint32_t row = round_slot(blockIdx.x, gridDim.x, a.classes); row < a.num_rows; row += gridDim.x, round++) {
        fin.top_of_row();
        const uint32_t z = opaque_zero(row);
        const uint32_t tid = tid0 + z;
        const int64_t *in = a.evals + (size_t)row * row_len;
        uint64_t *out_row = a.rows + (size_t)row * cw * (a.compact_rows ? 2 : 4);

        // The thread's permutation indices are re-read every row (2 x 64 bytes per thread from tables that live
        // in L2; `tid` is opaque so the loads stay in the loop): held in registers across the hash phase, their
        // 16 VGPRs were what pushed this kernel over its 128-register budget into scratch memory.
        uint32_t p2[E];
        i128 v[E];
        // ---- pass 1: repeat + permute(pi1) + accumulate (witness from global / L2) ----
        {
            const uint4 *q1 = reinterpret_cast<const uint4 *>(a.perm1 + (size_t)tid * E);
            const uint4 *q2 = reinterpret_cast<const uint4 *>(a.perm2 + (size_t)tid * E);
#pragma unroll
            for (int k = 0; k < E / 4; k++) {
                const uint4 s1 = q1[k], s2 = q2[k];
                v[4 * k + 0] = (i128)in[s1.x & (row_len - 1)];
                v[4 * k + 1] = (i128)in[s1.y & (row_len - 1)];
                v[4 * k + 2] = (i128)in[s1.z & (row_len - 1)];
                v[4 * k + 3] = (i128)in[s1.w & (row_len - 1)];
                p2[4 * k + 0] = s2.x;
                p2[4 * k + 1] = s2.y;
                p2[4 * k + 2] = s2.z;
                p2[4 * k + 3] = s2.w;
            }
        }
#pragma unroll
        for (int e = 1; e < E; e++) v[e] += v[e - 1];
        {
            // (a thread reaches the barrier inside only after it has read back its parked outputs of the
            // previous row, so the t2 stores below cannot overtake anybody's reads)
            const i128 pre = block_exclusive_scan_i96(v[E - 1], wave_tot, 0);
            const int32_t phi = (int32_t)(uint32_t)((u128)pre >> 64);
            ghi[tid] = phi;
#pragma unroll
            for (int e = 0; e < E; e++) {
                const i128 t = v[e] + pre;
                t2lo[e * PS + tid] = (uint64_t)t;
                t2dh[e * PS + tid] = (int8_t)((int32_t)(uint32_t)((u128)t >> 64) - phi);
            }
        }
        lds_barrier();
        // ---- pass 2: permute(pi2) + accumulate ----
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t j = p2[e];
            const uint32_t slot = (j & 15u) * PS + (j >> 4);
            const uint64_t lo = t2lo[slot];
            const int64_t hi = (int64_t)(ghi[j >> 4] + (int32_t)t2dh[slot]);
            v[e] = (i128)(((u128)(uint64_t)hi << 64) | lo);
        }
#pragma unroll
        for (int e = 1; e < E; e++) v[e] += v[e - 1];
        {
            // the barrier inside also orders everybody's t2 reads above before the parking stores below
            const i128 pre = block_exclusive_scan_i96(v[E - 1], wave_tot, 1);
#pragma unroll
            for (int e = 0; e < E; e++) v[e] += pre;
        }
        // park entries 8..15 in this thread's own slots
#pragma unroll
        for (int e = 8; e < E; e++) {
            park_lo[(e - 8) * PS + tid] = (uint64_t)v[e];
            park_hi[(e - 8) * PS + tid] = (uint32_t)((u128)v[e] >> 64);
        }
        // ---- output: two phases of 8 entries per thread, 8x8 lane-group transpose, butterfly ----
        uint32_t x0[8], x1[8], x2[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            x0[e] = (uint32_t)(uint64_t)v[e];
            x1[e] = (uint32_t)((uint64_t)v[e] >> 32);
            x2[e] = (uint32_t)((u128)v[e] >> 64);
        }
        // (both phases unrolled: arrays carried around a rolled loop stay in scratch memory)
#pragma unroll
        for (uint32_t q = 0; q < 2; q++) {
            if (q) {
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const uint64_t lo = park_lo[e * PS + tid];
                    x0[e] = (uint32_t)lo;
                    x1[e] = (uint32_t)(lo >> 32);
                    x2[e] = park_hi[e * PS + tid];
                }
            }
            transpose8(x0, tid);
            transpose8(x1, tid);
            transpose8(x2, tid);
            StridedLeaves<8, MODE> src;
            if (MODE == kStorePacked) {
                src.set_packed(a, row);
#pragma unroll
                for (int k = 0; k < 16; k++) src.ptab[k] = q ? ptab1[k] : ptab0[k];
            }
            src.out_row = out_row;
            src.compact = a.compact_rows;
            src.tree = HASH ? tree_of<MODE == kStorePacked>(a.layers, cw, row) : nullptr;
            src.cw = cw;
            src.T = 16;  // entry of step s = 16 s + src.tid (the subtree of lane 8m+s, see above)
            // wave base + 128 (lane / 8) + 8 q + (lane % 8): the low three bits are the lane's, as the butterfly needs
            src.tid = (tid & ~7u) * 16u + q * 8u + (tid & 7u);
            src.smask = q ? smask1 : smask0;
#pragma unroll
            for (int e = 0; e < 8; e++) {
                src.w0[e] = x0[e];
                src.w1[e] = x1[e];
                src.w2[e] = x2[e];
            }
            if (HASH) {
                uint32_t top[8];
                bfly_hash<3, 0>(src, top);
            } else {
                store_rows_only<8, 0>(src);
            }
        }
        fin.after_hash(a, wave0, tid0 & 63u, T);
        const bool last = row + gridDim.x >= a.num_rows;
        if (cc.ends_with(a, round, last)) {
            fin.chunk_end(a, 4u, cc, round, last, tid, T);
            cc.advance(round);
        }
    }
    fin.after_loop(a, wave0, tid0 & 63u, T, smem + 512, (uint32_t)(E * PS * 9u));  // (t2lo + t2dh)
    stamp_clock(a, 1);
}

// Digest of a witness (zip_commit's speculative hint, zip_hip.hip): out[0] += sum w[i], out[1] += sum w[i] * (2 i + 1),
// both mod 2^64.  Not cryptographic: it guards against a caller that overwrote its DEVICE witness between a commit and
// the transparent re-run of that commit -- any single changed word changes out[0], a permuted array changes out[1].
// One workgroup per CU, 16 bytes per lane and load, four loads in flight: it runs beside the commit kernel and must
// not take the co-residency slots of the row combinations and the gathers (2048 workgroups of 8-byte loads took 340 us
// there and doubled the time of the row combinations).
__global__ void __launch_bounds__(256) witness_digest_kernel(const uint64_t *w, uint64_t n, unsigned long long *out) {
    unsigned long long a = 0, b = 0;
    const uint64_t n2 = n / 2, stride = (uint64_t)gridDim.x * blockDim.x;
    const ulonglong2 *w2 = reinterpret_cast<const ulonglong2 *>(w);
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        ulonglong2 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = w2[i + k * stride];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint64_t j = 2 * (i + k * stride);
            a += v[k].x + v[k].y;
            b += v[k].x * (2ull * j + 1ull) + v[k].y * (2ull * j + 3ull);
        }
    }
    for (; i < n2; i += stride) {
        const ulonglong2 v = w2[i];
        a += v.x + v.y;
        b += v.x * (4ull * i + 1ull) + v.y * (4ull * i + 3ull);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        a += w[n - 1];
        b += w[n - 1] * (2ull * (n - 1) + 1ull);
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out, a);
        atomicAdd(out + 1, b);
    }
}

// [n][2] compact row entries (w0, w1, w2, sign as four 32-bit words) -> [n][4] Int<4> limbs
__global__ void __launch_bounds__(256) expand_rows_kernel(const uint4 *in, uint4 *out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        out[2 * i] = v;
        out[2 * i + 1] = make_uint4(v.w, v.w, v.w, v.w);
    }
}

// One wave that parks a stream until `*counter >= target` (a chunk of the persistent commit
// kernel is complete).  The spin is bounded: after `timeout_ticks` of the 100 MHz wall clock
// it gives up and raises *timeout_flag, so a lost producer cannot hang the GPU.
// The poll is an atomic read-modify-write (add 0): device-scope atomics execute at the memory
// side, whereas a plain or sc1 load can keep hitting a stale copy of the line in this XCD's L2
// for as long as the producer kernel runs (observed: the wait only ended at the producer's end).
// (`zero` is a runtime 0: a literal would let the compiler fold the RMW back into a load.)
__global__ void __launch_bounds__(64) wait_counter_kernel(uint32_t *counter, uint32_t target, uint32_t zero,
                                                          uint32_t *timeout_flag, unsigned long long timeout_ticks) {
    // Four older, always-ready hashing waves share this SIMD: without a priority bump this
    // wave only gets issue slots when they drain (observed: the wait ended with the producer).
    __builtin_amdgcn_s_setprio(3);
    if (threadIdx.x == 0) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_fetch_add(counter, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(64);
            if (wall_clock64() - t0 > timeout_ticks) {
                __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

// Inputs / node sink of one thread of merkle_upper_kernel.
struct UpperNodes {
    const uint32_t *in;
    uint32_t *tree, *root;
    uint32_t cw, level_in, depth, n0;  // n0 = index of the thread's first input node
    template <int E0>
    __device__ __forceinline__ void leaf(uint32_t (&h)[8]) {
        load_hash(in + (size_t)E0 * 8, h);
    }
    __device__ __forceinline__ void store(int lvl, uint32_t idx, const uint32_t (&h)[8]) {
        const uint32_t lv = level_in + lvl;
        store_hash(tree + ((size_t)level_off(cw, lv) + (n0 >> lvl) + idx) * 8, h);
        if (lv == depth) store_hash(root, h);
    }
};

// Batched upper Merkle levels: each thread reduces 2^NL consecutive nodes of
// level `level_in` of one row tree to one node of level `level_in + NL`, writing
// every intermediate node.  All lanes stay busy at every level, unlike a
// per-workgroup tree reduction.  When the top is reached the root is also copied
// to roots[row] (MultilinearZipCommitment.roots, src/zip/pcs/structs.rs:42-45).
template <int NL>
__global__ void __launch_bounds__(256) merkle_upper_kernel(uint32_t *layers, uint32_t *roots,
                                                           uint32_t num_rows, uint32_t cw,
                                                           uint32_t level_in, uint32_t depth) {
    const uint32_t width_in = cw >> level_in;
    const uint32_t groups = width_in >> NL;
    const uint32_t log_groups = 31u - __builtin_clz(groups);
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= num_rows * groups) return;
    const uint32_t r = gid >> log_groups, g = gid & (groups - 1);
    uint32_t *tree = layers + (size_t)r * (2u * cw) * 8;
    const uint32_t *src_nodes = tree + ((size_t)level_off(cw, level_in) + ((size_t)g << NL)) * 8;
    UpperNodes src{src_nodes, tree, roots + (size_t)r * 8, cw, level_in, depth, g << NL};
    uint32_t top[8];
    subtree_hash<NL, 0>(src, top);
}

// depth == 0 trees (a single leaf) have root == leaf hash
__global__ void copy_roots_depth0_kernel(const uint32_t *layers, uint32_t *roots, uint32_t num_rows,
                                         uint32_t cw) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= num_rows) return;
    const uint32_t *tree = layers + (size_t)r * (2u * cw) * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) roots[(size_t)r * 8 + i] = tree[i];
}

// Leaf hashes of arbitrary LIMBS-limb integers (standalone MerkleTree::new,
// src/zip/pcs/utils.rs:74-85; benches/zip_benches.rs:80-98 uses random Int<4>).
template <int LIMBS>
__global__ void __launch_bounds__(256) merkle_leaves_kernel(const uint64_t *leaves, uint32_t *layers,
                                                            uint32_t num_trees, uint32_t nleaves) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)num_trees * nleaves) return;
    const uint32_t t = (uint32_t)(gid / nleaves), i = (uint32_t)(gid % nleaves);
    uint64_t limb[LIMBS];
#pragma unroll
    for (int k = 0; k < LIMBS; k++) limb[k] = leaves[gid * LIMBS + k];
    uint32_t h[8];
    blake3_leaf_limbs<LIMBS>(limb, h);
    store_hash(layers + ((size_t)t * (2u * nleaves) + i) * 8, h);
}

}  // namespace zipk
