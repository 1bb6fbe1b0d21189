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

struct CommitArgs {
    const int64_t *evals;
    const uint32_t *perm1;
    const uint32_t *perm2;
    uint64_t *rows;
    uint32_t *layers;  // 8 words per hash
    uint32_t row_len;
    uint32_t cw;
    uint32_t nact;  // active threads per workgroup = cw / E
};

__device__ __forceinline__ i128 shfl_up_i96(i128 x, int off) {
    uint32_t d0 = (uint32_t)x, d1 = (uint32_t)((u128)x >> 32), d2 = (uint32_t)((u128)x >> 64);
    d0 = __shfl_up(d0, off, 64);
    d1 = __shfl_up(d1, off, 64);
    d2 = __shfl_up(d2, off, 64);
    const int64_t hi = (int64_t)(int32_t)d2;  // sign-extend bit 95
    return (i128)(((u128)(uint64_t)hi << 64) | ((u128)d1 << 32) | d0);
}

// Exclusive prefix over the workgroup of one (<= 96-bit signed) value per thread.
// wave_tot: LDS scratch of >= 16 entries.  Contains two barriers.
__device__ __forceinline__ i128 block_exclusive_scan_i96(i128 total, i128 *wave_tot) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    i128 x = total;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        i128 y = shfl_up_i96(x, off);
        if (lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wid] = x;
    __syncthreads();
    i128 base = 0;
    for (int w = 0; w < wid; w++) base += wave_tot[w];
    __syncthreads();
    return base + (x - total);
}

__device__ __forceinline__ void store_hash(uint32_t *dst, const uint32_t (&h)[8]) {
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

// Leaves / node sink of one thread of raa_commit_kernel: E consecutive codeword entries.
template <int E>
struct CommitLeaves {
    const i128 (&v)[E];
    uint32_t *tree;
    uint32_t cw, j0;
    template <int E0>
    __device__ __forceinline__ void leaf(uint32_t (&h)[8]) {
        blake3_leaf_sext96((uint32_t)v[E0], (uint32_t)((u128)v[E0] >> 32), (uint32_t)((u128)v[E0] >> 64), h);
        store_hash(tree + (size_t)(j0 + E0) * 8, h);
    }
    __device__ __forceinline__ void store(int lvl, uint32_t idx, const uint32_t (&h)[8]) {
        store_hash(tree + ((size_t)level_off(cw, lvl) + (j0 >> lvl) + idx) * 8, h);
    }
};

// Strided ownership for the output phase: at step e lane t owns codeword entry
// j = e*T + t, so the 32-byte row entries, leaf hashes and nodes that a wave stores in one
// instruction are adjacent in memory (4 lanes per 128-byte line instead of one lane per
// line).  Sibling leaves then sit in NEIGHBOUR LANES; the in-thread subtree becomes a
// butterfly: at level l a lane exchanges one child hash with lane t ^ 2^(l-1) and ends up
// with the node of step E0 + (t mod 2^l).  Every lane still hashes E leaves, E/2 ... 1 nodes.
template <int E>
struct StridedLeaves {
    uint32_t w0[E], w1[E], w2[E];  // 96-bit two's-complement values of the lane's E entries
    uint64_t *out_row;
    uint32_t *tree;
    uint32_t cw, T, tid;
    template <int E0>
    __device__ __forceinline__ void leaf(uint32_t (&h)[8]) {
        const uint32_t j = E0 * T + tid;
        const uint32_t s = (uint32_t)((int32_t)w2[E0] >> 31);
        uint4 *o = reinterpret_cast<uint4 *>(out_row + (size_t)j * 4);
        o[0] = make_uint4(w0[E0], w1[E0], w2[E0], s);  // sign extension to Int<4>
        o[1] = make_uint4(s, s, s, s);
        blake3_leaf_sext96(w0[E0], w1[E0], w2[E0], h);
        store_hash(tree + (size_t)j * 8, h);
    }
    __device__ __forceinline__ void store(int lvl, uint32_t e, const uint32_t (&h)[8]) {
        store_hash(tree + ((size_t)level_off(cw, lvl) + ((e * T + tid) >> lvl)) * 8, h);
    }
};

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
            const uint32_t rcv = __shfl_xor(snd, 1 << (LVL - 1), 64);
            m[i] = up ? rcv : A[i];      // left child
            m[8 + i] = up ? B[i] : rcv;  // right child
        }
        blake3_block(m, 64u, h);
        src.store(LVL, E0 + (src.tid & ((1u << LVL) - 1u)), h);
    }
}

// One workgroup per witness row; thread t owns the E consecutive codeword entries
// [t*E, t*E+E).  Values never exceed 64 + 2*log2(cw) + 1 <= 96 bits (width
// assertion src/zip/code_raa.rs:53-72), so scans run on i128 lanes and the 256-bit
// result is the sign extension.
//   T2_LDS = true : witness row and the intermediate codeword t2 live in LDS
//                   (cw*12 + row_len*8 bytes; up to cw = 8192).
//   T2_LDS = false: t2 is parked in the (not yet written) output row in HBM/L2
//                   and the witness row is gathered from global memory.
// In LDS, entry j sits at slot (j % E) * (T + 32/E) + j / E: the thread-contiguous writes
// of a wave are bank-conflict free, and so are the strided reads (entry e*T + t) of the
// output phase; the pi2 gather is random either way.
template <int E, bool HASH, bool T2_LDS>
__global__ void __launch_bounds__(1024) raa_commit_kernel(CommitArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int LOGE = (E == 1) ? 0 : (E == 2) ? 1 : (E == 4) ? 2 : (E == 8) ? 3 : 4;
    static_assert((1 << LOGE) == E, "E must be a power of two <= 16");

    const uint32_t tid = threadIdx.x, T = blockDim.x;
    constexpr uint32_t PAD = 32 / E;
    const uint32_t PS = T + PAD;  // plane stride (slots)
    const uint32_t row = blockIdx.x;
    const uint32_t cw = a.cw, row_len = a.row_len;
    const bool active = tid < a.nact;
    const uint32_t j0 = tid * E;

    i128 *wave_tot = reinterpret_cast<i128 *>(smem);                  // 16 entries
    uint64_t *t2lo = reinterpret_cast<uint64_t *>(smem + 256);         // E planes of PS slots
    uint32_t *t2hi = reinterpret_cast<uint32_t *>(t2lo + (T2_LDS ? E * PS : 0));
    int64_t *rowbuf = reinterpret_cast<int64_t *>(t2hi + (T2_LDS ? E * PS : 0));

    const int64_t *in = a.evals + (size_t)row * row_len;
    uint64_t *out_row = a.rows + (size_t)row * cw * 4;
    u128 *t2g = reinterpret_cast<u128 *>(out_row);  // T2_LDS == false: first half of the output row

    if (T2_LDS) {
        for (uint32_t i = tid; i < row_len; i += T) rowbuf[i] = in[i];
        __syncthreads();
    }

    i128 v[E];
    // ---- pass 1: repeat + permute(pi1) + accumulate ----------------------------
    if (active) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t src = a.perm1[j0 + e] & (row_len - 1);  // repeat: t0[j] = row[j mod row_len]
            v[e] = (i128)(T2_LDS ? rowbuf[src] : in[src]);
        }
#pragma unroll
        for (int e = 1; e < E; e++) v[e] += v[e - 1];
    } else {
#pragma unroll
        for (int e = 0; e < E; e++) v[e] = 0;
    }
    {
        const i128 pre = block_exclusive_scan_i96(v[E - 1], wave_tot);
        if (active) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                v[e] += pre;
                if (T2_LDS) {
                    const uint32_t slot = e * PS + tid;
                    t2lo[slot] = (uint64_t)v[e];
                    t2hi[slot] = (uint32_t)((u128)v[e] >> 64);
                } else {
                    t2g[j0 + e] = (u128)v[e];
                }
            }
        }
    }
    __syncthreads();
    // ---- pass 2: permute(pi2) + accumulate -------------------------------------
    if (active) {
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t src = a.perm2[j0 + e];
            if (T2_LDS) {
                const uint32_t slot = (src & (E - 1)) * PS + (src >> LOGE);
                const uint64_t lo = t2lo[slot];
                const int64_t hi = (int64_t)(int32_t)t2hi[slot];
                v[e] = (i128)(((u128)(uint64_t)hi << 64) | lo);
            } else {
                v[e] = (i128)t2g[src];
            }
        }
#pragma unroll
        for (int e = 1; e < E; e++) v[e] += v[e - 1];
    }
    {
        // the barriers inside also order the t2 reads above before the row stores below
        const i128 pre = block_exclusive_scan_i96(v[E - 1], wave_tot);
#pragma unroll
        for (int e = 0; e < E; e++) v[e] += pre;
    }
    if (T2_LDS) {
        // ---- transpose to strided ownership through LDS, then rows + hashes ----------
        // (the second barrier inside the scan above already ordered every pi2 gather of t2
        // before these writes)
        if (active) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t slot = e * PS + tid;
                t2lo[slot] = (uint64_t)v[e];
                t2hi[slot] = (uint32_t)((u128)v[e] >> 64);
            }
        }
        __syncthreads();
        if (!active) return;
        StridedLeaves<E> src;
        src.out_row = out_row;
        src.tree = HASH ? a.layers + (size_t)row * (2u * cw) * 8 : nullptr;
        src.cw = cw;
        src.T = a.nact;
        src.tid = tid;
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t j = e * a.nact + tid;
            const uint32_t slot = (j & (E - 1)) * PS + (j >> LOGE);
            const uint64_t lo = t2lo[slot];
            src.w0[e] = (uint32_t)lo;
            src.w1[e] = (uint32_t)(lo >> 32);
            src.w2[e] = t2hi[slot];
        }
        if (HASH) {
            uint32_t top[8];
            bfly_hash<LOGE, 0>(src, top);
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const uint32_t j = e * a.nact + tid;
                const uint32_t sg = (uint32_t)((int32_t)src.w2[e] >> 31);
                uint4 *o = reinterpret_cast<uint4 *>(out_row + (size_t)j * 4);
                o[0] = make_uint4(src.w0[e], src.w1[e], src.w2[e], sg);
                o[1] = make_uint4(sg, sg, sg, sg);
            }
        }
    } else {
        if (!active) return;
        // ---- thread-contiguous outputs (cw too large for the LDS transposition) -----
        uint4 *orow = reinterpret_cast<uint4 *>(out_row + (size_t)j0 * 4);
#pragma unroll
        for (int e = 0; e < E; e++) {
            const uint32_t d0 = (uint32_t)v[e], d1 = (uint32_t)((u128)v[e] >> 32),
                           d2 = (uint32_t)((u128)v[e] >> 64);
            const uint32_t sg = (uint32_t)((int32_t)d2 >> 31);
            orow[2 * e] = make_uint4(d0, d1, d2, sg);
            orow[2 * e + 1] = make_uint4(sg, sg, sg, sg);
        }
        if (HASH) {
            CommitLeaves<E> src{v, a.layers + (size_t)row * (2u * cw) * 8, cw, j0};
            uint32_t top[8];
            subtree_hash<LOGE, 0>(src, top);
        }
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
