/*
 * zip_oracle.h -- CPU restatement of the Zip PCS commit/open/verify path of
 * NethermindEth/zinc (reference snapshot 2025-08-24).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path (zinc_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * Parity status
 * -------------
 *  - BLAKE3 single-block hashing: pinned against the official BLAKE3 C
 *    implementation (LLVM's vendored copy, exported by libclang-cpp.so) through
 *    tests/golden/blake3_vectors.json, and against the BLAKE3("") KAT.
 *  - Keccak-256 / Fiat-Shamir transcript: permutation pinned against
 *    hashlib.sha3_256; challenge derivation pinned by the reference KAT at
 *    src/transcript.rs:214-234.
 *  - Montgomery arithmetic: pinned by the reference KAT src/field/config.rs:338-345
 *    and against Python big integers.
 *  - Wide integer scan / combine_rows / expand: pinned by the reference KATs
 *    src/zip/code_raa.rs:199-244, src/zip/pcs/utils.rs:301-337,
 *    src/zip/utils.rs:164-234.
 *  - rand 0.9 `StdRng::seed_from_u64` + `SliceRandom::shuffle`
 *    (orc_shuffle_seeded_perm): restated from the published algorithm of the
 *    un-vendored crates; the reference pins no permutation.  Pinned piece by
 *    piece by published vectors (tests/golden/rand_vectors.json): the ChaCha12
 *    block, StdRng's word/counter layout (rand's test_stdrng_construction),
 *    PCG32 (O'Neill's demo), the IncreasingUniform shuffle with Canon's-method
 *    random_range (rand's value_stability_slice), and rand_core's
 *    seed_from_u64 (PCG32 expansion of the u64 seed: rand_pcg's
 *    test_lcg64xsh32_construction).  The product ABI takes explicit
 *    permutation tables, so nothing on the GPU depends on any of this.
 *  - The reference itself (Rust) cannot be built in this image (no cargo/rustc,
 *    no vendored crates), so there is no oracle/_ref.
 *
 * All multi-limb integers are little-endian arrays of uint64_t limbs in two's
 * complement (src/field/int.rs:23-25).  Field elements are little-endian limb
 * arrays holding the Montgomery representation (src/field.rs:24-32).
 */
#ifndef ZIP_ORACLE_H
#define ZIP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_ERR_OVERFLOW (-1)   /* crypto-bigint checked_add/checked_mul would panic */
#define ORC_ERR_PARAM (-2)
#define ORC_ERR_PROOF (-3)      /* verifier rejected */
#define ORC_ERR_TRANSCRIPT (-4) /* proof stream exhausted */
#define ORC_ERR_ALLOC (-5)

#define ORC_MAX_FL 8 /* max field limbs handled */

/* ------------------------------------------------------------------ BLAKE3 */
/* blake3::hash(msg) for len <= 64 (one chunk, one block).  Public BLAKE3 spec;
 * call sites src/zip/pcs/utils.rs:90,107-112. */
int orc_blake3_hash_block(const uint8_t *msg, size_t len, uint8_t out[32]);

/* ----------------------------------------------------------------- Keccak */
typedef struct {
    uint64_t st[25];
    uint8_t buf[136];
    uint32_t buflen;
} orc_keccak;

void orc_keccak_init(orc_keccak *k);
void orc_keccak_update(orc_keccak *k, const uint8_t *data, size_t len);
/* finalises a COPY of the state (sha3::Digest::finalize on a clone); domain
 * byte 0x01 = Keccak-256 (sha3 crate `Keccak256`), 0x06 = NIST SHA3-256. */
void orc_keccak_finalize_copy(const orc_keccak *k, uint8_t domain, uint8_t out[32]);

/* KeccakTranscript, src/transcript.rs */
void orc_tr_get_random_bytes(orc_keccak *k, size_t length, uint8_t *out);     /* :40-55  */
uint64_t orc_tr_get_u64(orc_keccak *k);                                       /* :183-185 */
void orc_tr_get_integer_challenge(orc_keccak *k, uint32_t n_limbs, uint64_t *out); /* :142-155 */

/* ------------------------------------------------------------------ field */
typedef struct {
    uint32_t fl;                    /* number of 64-bit limbs */
    uint64_t modulus[ORC_MAX_FL];
    uint64_t r[ORC_MAX_FL];         /* R  mod q,  R = 2^(64*fl) */
    uint64_t r2[ORC_MAX_FL];        /* R^2 mod q */
    uint64_t inv;                   /* -q^{-1} mod 2^64 */
    int has_spare_bit;
} orc_field;

int orc_field_new(orc_field *f, uint32_t fl, const uint64_t *modulus);        /* config.rs:174-186 */
void orc_field_mul(const orc_field *f, uint64_t *a, const uint64_t *b);       /* config.rs:163-170 */
void orc_field_add(const orc_field *f, uint64_t *a, const uint64_t *b);       /* config.rs:53-58  */
void orc_field_sub(const orc_field *f, uint64_t *a, const uint64_t *b);       /* config.rs:60-66  */
void orc_field_neg(const orc_field *f, uint64_t *a);                          /* arithmetic.rs:130-149 */
void orc_field_from_i64(const orc_field *f, int64_t v, uint64_t *out);        /* conversion.rs:86-100, field.rs:536-568 */
void orc_field_from_u128(const orc_field *f, uint64_t lo, uint64_t hi, uint64_t *out); /* conversion.rs:9-46 */
/* map a signed n-limb integer (Int<n>) into the field */
void orc_field_from_int(const orc_field *f, const uint64_t *v, uint32_t n, uint64_t *out);
/* KeccakTranscript::get_challenge, transcript.rs:88-133 */
void orc_tr_get_challenge(orc_keccak *k, const orc_field *f, uint64_t *out);
/* RandomField::absorb_into_transcript (Initialized), field.rs:360-378 */
void orc_tr_absorb_field(orc_keccak *k, const orc_field *f, const uint64_t *val);
/* build_eq_x_r_vec, sumcheck/utils.rs:117-177: out has 2^nvars field elems */
int orc_build_eq_x_r(const orc_field *f, const uint64_t *r, uint32_t nvars, uint64_t *out);

/* ---------------------------------------------------------------- shuffle */
/* rand 0.9 restatement (see the header comment for what is pinned).  Fills perm with the permutation such
 * that shuffle_seeded(x, seed)[j] == x[perm[j]]  (zip/utils.rs:139-142). */
void orc_shuffle_seeded_perm(uint64_t seed, uint32_t len, uint32_t *perm);
/* known-answer hooks for its pieces (tests/test_oracle_kats.py, tests/golden/rand_vectors.json) */
void orc_kat_chacha12_block(const uint32_t key[8], uint64_t counter, uint32_t out[16]);
void orc_kat_stdrng_from_seed_u64(const uint8_t seed[32], uint32_t n, uint64_t *out);
void orc_kat_pcg32(uint64_t state, uint64_t stream, uint32_t n, uint32_t *out);
void orc_kat_shuffle_pcg32(uint64_t state, uint64_t stream, uint32_t len, uint32_t *perm);
void orc_kat_seed_from_u64(uint64_t state, uint32_t key_out[8]);

/* ---------------------------------------------------------------- RAA code */
/* RaaCode::encode_inner (code_raa.rs:89-105) on one row, explicit permutations.
 * in: row_len integers of in_limbs limbs; out: row_len*rep integers of out_limbs. */
int orc_raa_encode_row(const uint64_t *row, uint32_t in_limbs, uint32_t row_len,
                       uint32_t rep, const uint32_t *perm1, const uint32_t *perm2,
                       uint64_t *out, uint32_t out_limbs);
/* encode_f (code_raa.rs:133-138): same passes over field elements */
void orc_raa_encode_row_field(const orc_field *f, const uint64_t *row, uint32_t row_len,
                              uint32_t rep, const uint32_t *perm1, const uint32_t *perm2,
                              uint64_t *out);

/* ----------------------------------------------------------------- Merkle */
/* MerkleTree::new (pcs/utils.rs:74-118).  leaves: 2^depth integers of
 * leaf_limbs limbs.  layers: (2<<depth)-1 hashes of 32 bytes, INCLUDING the
 * root as the last entry (the reference pops it into .root). */
int orc_merkle_tree(uint32_t depth, const uint64_t *leaves, uint32_t leaf_limbs,
                    uint8_t *layers);
/* MerkleProof::create_proof (pcs/utils.rs:163-176): writes depth*32 bytes */
void orc_merkle_path(uint32_t depth, const uint8_t *layers, uint32_t leaf, uint8_t *path);
/* MerkleProof::verify (pcs/utils.rs:178-210) */
int orc_merkle_verify(uint32_t depth, const uint8_t *path, const uint8_t root[32],
                      const uint64_t *leaf, uint32_t leaf_limbs, uint32_t leaf_index);

/* ------------------------------------------------------------------ commit */
typedef struct {
    uint32_t num_vars, row_len, num_rows, codeword_len, rep, depth;
    uint32_t n_limbs, k_limbs, m_limbs; /* Int<N>, Int<4N>, Int<8N> */
    const uint32_t *perm1, *perm2;      /* codeword_len each */
    uint32_t num_column_opening;        /* 1000 for DefaultLinearCodeSpec */
    uint32_t num_proximity_testing;     /* 1 */
} orc_params;

/* RaaCode::new geometry + MultilinearZip::setup (code_raa.rs:35-86, structs.rs:79-91);
 * returns ORC_ERR_PARAM when the width assertion code_raa.rs:68-72 fails. */
int orc_params_init(orc_params *p, uint32_t num_vars, uint32_t n_limbs, uint32_t rep,
                    const uint32_t *perm1, const uint32_t *perm2);

/* MultilinearZip::commit (commit.rs:50-87).
 * rows: num_rows*cw*k_limbs u64; layers: num_rows * ((2<<depth)-1) * 32 bytes (root last
 * in each tree block); roots: num_rows*32.  layers/roots may be NULL (commit_no_merkle). */
int orc_commit(const orc_params *p, const uint64_t *evals, uint64_t *rows,
               uint8_t *layers, uint8_t *roots);

/* The same commit, kept only as far as a checker of a LARGE instance needs it (2^26: rows + layers would be 12 GiB):
 * every root (commit.rs:78-81) and, for the n_cols picked columns, the complete opening block that
 * open_merkle_trees_for_column writes for it (open_z.rs:124-143; pcs/utils.rs:163-176,220-233;
 * pcs_transcript.rs:198-211): num_rows values (k_limbs LE limbs each), then num_rows x { be64(depth), depth x 32 B }.
 * Row by row: encode, tree, root, the picked entries and paths; nothing else is retained.
 * blocks: n_cols * num_rows * (8 k_limbs + 8 + 32 depth) bytes. */
int orc_commit_open_columns(const orc_params *p, const uint64_t *evals, const uint32_t *cols, uint32_t n_cols,
                            uint8_t *roots, uint8_t *blocks);

/* combine_rows over Int<M> (zip/utils.rs:94-127 via open_z.rs:103-112). */
int orc_combine_rows_int(const uint64_t *coeffs, uint32_t coeff_limbs,
                         const uint64_t *evals, uint32_t eval_limbs,
                         uint32_t num_rows, uint32_t row_len,
                         uint64_t *out, uint32_t m_limbs);
/* map_to_field + combine_rows over F (open_z.rs:76-90). q0: num_rows field elems. */
void orc_combine_rows_field(const orc_field *f, const uint64_t *q0,
                            const uint64_t *evals, uint32_t eval_limbs,
                            uint32_t num_rows, uint32_t row_len, uint64_t *out);

/* proof length, commit.rs:712-737 */
size_t orc_proof_len(const orc_params *p, uint32_t fl);

/* MultilinearZip::open (open_z.rs:22-143): writes the proof stream, drives the
 * Fiat-Shamir transcript `fs` exactly as PcsTranscript does.  point: num_vars
 * field elements (Montgomery limbs).  If cols_out != NULL the squeezed column
 * indices are also returned (num_column_opening of them); if coeffs_out != NULL
 * the proximity coefficients (num_rows * n_limbs). */
int orc_open(const orc_params *p, const orc_field *f, const uint64_t *evals,
             const uint64_t *rows, const uint8_t *layers, const uint64_t *point,
             orc_keccak *fs, uint8_t *proof, size_t proof_cap, size_t *proof_len,
             uint32_t *cols_out, uint64_t *coeffs_out);

/* MultilinearZip::verify (verify_z.rs:19-188).  eval: claimed evaluation (field). */
int orc_verify(const orc_params *p, const orc_field *f, const uint8_t *roots,
               const uint64_t *point, const uint64_t *eval, orc_keccak *fs,
               const uint8_t *proof, size_t proof_len, int check_merkle);

/* DenseMultilinearExtension::evaluate over the field of integer evals
 * (prover.rs:317-319): sum_i eq(point)[i] * phi(evals[i]). */
void orc_mle_eval_field(const orc_field *f, const uint64_t *evals, uint32_t eval_limbs,
                        uint32_t num_vars, const uint64_t *point, uint64_t *out);

/* MLSumcheck::prove_as_subprotocol (src/sumcheck.rs:56-112) with prove_round
 * (src/sumcheck/prover.rs:62-180) and fix_variables (src/poly_f/mle/dense.rs:142-168) for the
 * combination function comb_fn(vals) = vals[0] * vals[1] * ... (ZincProver's second sumcheck,
 * src/zinc/prover.rs:300).  mles: n_mles tables of 2^nvars field elements (Montgomery limbs),
 * consumed in place.  msgs_out: nvars * (degree + 1) elements (ProverMsg.evaluations per round);
 * randomness_out: nvars elements (ProverState.randomness).  Returns ORC_ERR_PARAM for nvars == 0
 * (the reference returns an empty proof there). */
int orc_sumcheck_prove_product(const orc_field *f, uint64_t *mles, uint32_t n_mles, uint32_t nvars,
                               uint32_t degree, orc_keccak *transcript, uint64_t *msgs_out,
                               uint64_t *randomness_out);
/* The same with the combination function of ZincProver's first sumcheck
 * (sumcheck_polynomial_comb_fn_1, src/zinc/utils.rs:77-94):
 *   comb(vals) = (sum_t coeffs[t] * prod_{j in term_masks[t]} vals[j]) * vals[n_mles - 1]
 * term_masks[t]: bit j set = MLE j is a factor of term t (ccs.S[t]); coeffs: n_terms field elements
 * (ccs.c, Montgomery).  n_terms == 0 selects the plain product of all MLEs. */
int orc_sumcheck_prove(const orc_field *f, uint64_t *mles, uint32_t n_mles, uint32_t nvars, uint32_t degree,
                       uint32_t n_terms, const uint32_t *term_masks, const uint64_t *coeffs,
                       orc_keccak *transcript, uint64_t *msgs_out, uint64_t *randomness_out);

/* ---------------------------------------------------------------- Spartan (ZincProver / ZincVerifier) */
/* SparseMatrix<Int<1>> (src/sparse_matrix.rs:12-17) flattened to CSR: row_ptr has n_rows + 1 entries;
 * rows the reference's `coeffs` does not hold (pad_rows only bumps n_rows) are simply empty. */
typedef struct {
    uint32_t n_rows, n_cols;
    const uint32_t *row_ptr, *col_idx;
    const int64_t *values;
} orc_sparse;
/* CCS_Z (src/ccs/ccs_z.rs:30-52) + Statement_Z.constraints.  S_masks[i]: bit j set = matrix j in S[i].
 * Supported shape = what ZincProver itself supports: m == n == 2^s == 2^s_prime (compute_eval_table_sparse
 * asserts rx.len() == ccs.n, ccs_f.rs:133, sparse_matrix.rs:172), every c[i] != 0 and the concatenation of
 * the S[i] equal to 0..t-1 in order (sumcheck_polynomial_comb_fn_1 indexes the MLE list of
 * prepare_lin_sumcheck_polynomial by matrix number, zinc/utils.rs:66-70,84-88). */
typedef struct {
    uint32_t m, n, s, s_prime, t, q, d;
    const orc_sparse *M;
    const uint32_t *S_masks;
    const int64_t *c;
} orc_ccs;

void orc_field_inv(const orc_field *f, const uint64_t *a, uint64_t *out);
void orc_interpolate_uni_poly(const orc_field *f, const uint64_t *p_i, uint32_t len, const uint64_t *x,
                              uint64_t *out);                                   /* sumcheck/verifier.rs:161-303 */
int orc_sumcheck_verify(const orc_field *f, uint32_t nvars, uint32_t degree, const uint64_t *claimed_sum,
                        const uint64_t *msgs, orc_keccak *tr, uint64_t *point_out, uint64_t *expected_out);
int orc_ccs_mz(const orc_field *f, const orc_ccs *ccs, const int64_t *z, uint32_t z_len, uint64_t *mz_out);
int orc_ccs_second_table(const orc_field *f, const orc_ccs *ccs, const uint64_t *eq_rx, const uint64_t *gamma,
                         uint64_t *out);
int orc_spartan_prove(const orc_field *f, const orc_ccs *ccs, const int64_t *z, uint32_t z_len, orc_keccak *tr,
                      uint64_t *msgs1, uint64_t *r_x, uint64_t *msgs2, uint64_t *r_y, uint64_t *V_s);
int orc_spartan_verify(const orc_field *f, const orc_ccs *ccs, const uint64_t *msgs1, const uint64_t *msgs2,
                       const uint64_t *V_s, orc_keccak *tr, uint64_t *r_x, uint64_t *r_y, uint64_t *e_y,
                       uint64_t *gamma);
/* mle[M_k](r_x, r_y), k < t (DenseMultilinearExtension::from_matrix + evaluate, verifier.rs:248-261) */
int orc_ccs_eval_matrices(const orc_field *f, const orc_ccs *ccs, const uint64_t *r_x, const uint64_t *r_y,
                          uint64_t *v_xy);
int orc_spartan_final_check(const orc_field *f, const orc_ccs *ccs, const uint64_t *r_x, const uint64_t *r_y,
                            const uint64_t *gamma, const uint64_t *v, const uint64_t *e_y);

/* MLSumcheck::prove_as_subprotocol with rand_poly_comb_fn (src/sumcheck/utils.rs:67-78): sum_p coeffs[p] * prod_{j in
 * masks[p]} vals[j] -- the workload of benches/sumcheck_benches.rs (rand_poly: 7 products of 2-4 fresh MLEs) */
int orc_sumcheck_prove_products(const orc_field *f, uint64_t *mles, uint32_t n_mles, uint32_t nvars, uint32_t degree,
                                uint32_t n_products, const uint32_t *masks, const uint64_t *coeffs, orc_keccak *tr,
                                uint64_t *msgs_out, uint64_t *randomness_out);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
