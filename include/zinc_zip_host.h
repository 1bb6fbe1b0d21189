/*
 * zinc_zip_host.h -- C facade of libzinc_zip.so, the C++ mirror (zinc_amd/host/zinc_zip.hpp) of
 * the HOST side of zinc::zip: what stays in the Rust crate in the real integration (Fiat-Shamir
 * transcript, permutation-seed expansion, the eq tensor, parameter checks) driving the HIP
 * library through include/zip_hip.h.  The facade exists so that tests and tools can exercise
 * that layer through ctypes; it adds no behaviour of its own.
 *
 * Return codes: 0 ok; ZINC_ERR_INVALID_PARAM = zip::Error::InvalidPcsParam; ZINC_ERR_PANIC = a
 * place where the reference panics (assert!/expect); ZINC_ERR_DEVICE = HIP library failure.
 * zinc_last_error() returns the message of the calling thread's last failure.
 */
#ifndef ZINC_ZIP_HOST_H
#define ZINC_ZIP_HOST_H
#include <stddef.h>
#include <stdint.h>

#include "zip_hip.h" /* zip_sparse_matrix */
#ifdef __cplusplus
extern "C" {
#endif

#define ZINC_OK 0
#define ZINC_ERR_INVALID_PARAM (-1)
#define ZINC_ERR_PANIC (-2)
#define ZINC_ERR_DEVICE (-3)
#define ZINC_ERR_NULL (-4)

const char *zinc_last_error(void);

/* KeccakTranscript (src/transcript.rs) */
typedef struct zinc_transcript zinc_transcript;
zinc_transcript *zinc_transcript_new(void);
void zinc_transcript_free(zinc_transcript *t);
void zinc_transcript_absorb(zinc_transcript *t, const uint8_t *bytes, size_t len);
uint64_t zinc_transcript_get_u64(zinc_transcript *t);
void zinc_transcript_get_integer_challenges(zinc_transcript *t, size_t n, int64_t *out);
/* get_challenge::<RandomField<limbs>>: Montgomery limbs of the challenge */
int32_t zinc_transcript_get_challenge(zinc_transcript *t, const uint64_t *modulus, uint32_t limbs, uint64_t *out);

/* field helpers (src/field/config.rs, src/conversion.rs, src/sumcheck/utils.rs) */
int32_t zinc_field_constants(const uint64_t *modulus, uint32_t limbs, uint64_t *r, uint64_t *r2, uint64_t *inv);
int32_t zinc_field_mul(const uint64_t *modulus, uint32_t limbs, const uint64_t *a, const uint64_t *b, uint64_t *out);
int32_t zinc_map_to_field_i64(const uint64_t *modulus, uint32_t limbs, const int64_t *v, size_t n, uint64_t *out);
int32_t zinc_build_eq_x_r(const uint64_t *modulus, uint32_t limbs, const uint64_t *r, uint32_t nvars, uint64_t *out);

/* shuffle_seeded on the identity (src/zip/utils.rs:139-142; rand 0.9 restated, every piece held to a published
 * vector: tests/golden/rand_vectors.json) */
void zinc_shuffle_seeded_perm(uint64_t seed, uint32_t len, uint32_t *perm);
/* known-answer hook: the first n_words little-endian words rand_core::SeedableRng::seed_from_u64 expands `seed` to */
void zinc_kat_seed_from_u64(uint64_t seed, uint32_t *words, uint32_t n_words);

/* RaaCode::new (src/zip/code_raa.rs:35-86).  transcript == NULL uses MockTranscript (seeds 1, 2). */
typedef struct {
    uint32_t row_len, repetition_factor, num_column_opening, num_proximity_testing;
    uint64_t perm_1_seed, perm_2_seed;
} zinc_raa_code;
int32_t zinc_raa_code_new(uint64_t poly_size, zinc_transcript *transcript, zinc_raa_code *out);

/* MultilinearZip::setup / commit / open (src/zip/pcs/structs.rs:79-91, commit.rs:50-119, open_z.rs:22-40) */
typedef struct zinc_zip_params zinc_zip_params;
typedef struct zinc_zip_data zinc_zip_data;
int32_t zinc_zip_setup(uint64_t poly_size, const zinc_raa_code *code, int32_t device, zinc_zip_params **out);
void zinc_zip_params_free(zinc_zip_params *pp);
/* setup() caches device contexts per (device, geometry, seeds); this drops the cache and its device memory */
void zinc_zip_release_cached_contexts(void);
void zinc_zip_params_geometry(const zinc_zip_params *pp, uint32_t *num_vars, uint32_t *num_rows, uint32_t *row_len,
                              uint32_t *codeword_len);
/* roots_out: num_rows * 32 bytes (ignored when with_merkle == 0) */
int32_t zinc_zip_commit(const zinc_zip_params *pp, const int64_t *evals, size_t n_evals, uint32_t poly_num_vars,
                        int32_t with_merkle, uint8_t *roots_out, zinc_zip_data **out);
void zinc_zip_data_free(zinc_zip_data *d);
/* MultilinearZipData is `pub rows` / `pub rows_merkle_trees` in the reference (structs.rs:33-38) and its
 * tests read and mutate both: these move them between the host and the device handle.
 *   rows    num_rows * codeword_len Int<4>  (4 u64 each)
 *   layers  per row (2 * codeword_len - 2) hashes = MerkleTree.layers (pcs/utils.rs:67-71)
 *   roots   per row one hash.  Any pointer may be NULL on download; layers / roots NULL on upload = no trees. */
int32_t zinc_zip_data_download(const zinc_zip_data *d, uint64_t *rows_out, uint8_t *layers_out, uint8_t *roots_out);
int32_t zinc_zip_data_upload(const zinc_zip_params *pp, const uint64_t *rows, const uint8_t *layers,
                             const uint8_t *roots, zinc_zip_data **out);
/* MerkleTree::new(depth, leaves) (pcs/utils.rs:74-85) for n_leaves Int<leaf_limbs>; layers_out: (2 << depth) - 1
 * hashes, the root last.  ZINC_ERR_PANIC when n_leaves != 2^depth ("leaves.len().is_power_of_two()" / depth assert). */
int32_t zinc_merkle_tree_new(uint32_t depth, const uint64_t *leaves, size_t n_leaves, uint32_t leaf_limbs,
                             int32_t device, uint8_t *layers_out);

/* PcsTranscript (src/zip/pcs_transcript.rs) */
typedef struct zinc_pcs_transcript zinc_pcs_transcript;
zinc_pcs_transcript *zinc_pcs_transcript_new(void);
void zinc_pcs_transcript_free(zinc_pcs_transcript *t);
size_t zinc_pcs_transcript_len(const zinc_pcs_transcript *t);
void zinc_pcs_transcript_copy(const zinc_pcs_transcript *t, uint8_t *out); /* into_proof() */
/* the Fiat-Shamir state after the calls so far, as one more u64 squeeze (for equality checks) */
uint64_t zinc_pcs_transcript_probe(const zinc_pcs_transcript *t);

/* point: point_len field elements, Montgomery limbs */
int32_t zinc_zip_open(const zinc_zip_params *pp, const int64_t *evals, size_t n_evals, uint32_t poly_num_vars,
                      const zinc_zip_data *data, const uint64_t *point, size_t point_len, const uint64_t *modulus,
                      uint32_t limbs, zinc_pcs_transcript *transcript);

/* PcsTranscript::from_proof (src/zip/pcs_transcript.rs:28-35): the reading side */
zinc_pcs_transcript *zinc_pcs_transcript_from_proof(const uint8_t *proof, size_t len);
size_t zinc_pcs_transcript_position(const zinc_pcs_transcript *t);

/* MultilinearZip::verify (src/zip/pcs/verify_z.rs:19-38).  roots: num_rows * 32 bytes; eval: Montgomery limbs.
 * ZINC_OK = accepted; ZINC_ERR_INVALID_OPEN = rejected (zinc_last_error() carries the reference's message);
 * ZINC_ERR_PANIC where the reference panics. */
#define ZINC_ERR_INVALID_OPEN (-5)
int32_t zinc_zip_verify(const zinc_zip_params *vp, const uint8_t *roots, const uint64_t *point, size_t point_len,
                        const uint64_t *eval, const uint64_t *modulus, uint32_t limbs, zinc_pcs_transcript *transcript);

/* z_mle.map_to_field(config).evaluate(r_y) (src/zinc/prover.rs:317-319): out = limbs Montgomery limbs */
int32_t zinc_zip_evaluate(const zinc_zip_params *pp, const int64_t *evals, size_t n_evals, const uint64_t *point,
                          size_t point_len, const uint64_t *modulus, uint32_t limbs, uint64_t *out);

/* ZincProver::commit_z_mle_and_prove_evaluation (src/zinc/prover.rs:305-327) -> ZipProof {z_comm, v, pcs_proof}.
 * transcript: the prover's main KeccakTranscript (two permutation seeds are drawn from it).
 * Call once with proof_out == NULL to learn *proof_len and *n_roots; the result is kept in the handle. */
typedef struct zinc_zip_proof zinc_zip_proof;
int32_t zinc_commit_z_mle_and_prove_evaluation(const int64_t *z_evals, size_t m, const uint64_t *r_y, size_t r_y_len,
                                               zinc_transcript *transcript, const uint64_t *modulus, uint32_t limbs,
                                               int32_t device, zinc_zip_proof **out);
size_t zinc_zip_proof_len(const zinc_zip_proof *p);
size_t zinc_zip_proof_num_roots(const zinc_zip_proof *p);
void zinc_zip_proof_read(const zinc_zip_proof *p, uint8_t *roots_out, uint64_t *v_out, uint8_t *pcs_proof_out);
void zinc_zip_proof_free(zinc_zip_proof *p);

/* MLSumcheck::prove_as_subprotocol with comb_fn = product (src/sumcheck.rs:56-112, zinc/prover.rs:297-302).
 * mles: n_mles host tables of 2^nvars elements (Montgomery limbs).  msgs_out: nvars * (degree+1) * limbs;
 * randomness_out: nvars * limbs. */
int32_t zinc_sumcheck_prove_product(zinc_transcript *transcript, const uint64_t *const *mles, uint32_t n_mles,
                                    uint32_t nvars, uint32_t degree, const uint64_t *modulus, uint32_t limbs,
                                    int32_t device, uint64_t *msgs_out, uint64_t *randomness_out);

/* MLSumcheck::prove_as_subprotocol with sumcheck_polynomial_comb_fn_1 (src/zinc/utils.rs:77-94):
 * (sum_t c[t] * prod_{j in S[t]} vals[j]) * vals[n_mles - 1].  c: n_terms field elements (Montgomery);
 * s_masks[t]: the bits of ccs.S[t] as positions in `mles` (the eq() MLE is the last one). */
int32_t zinc_sumcheck_prove_ccs(zinc_transcript *transcript, const uint64_t *const *mles, uint32_t n_mles,
                                uint32_t nvars, uint32_t degree, uint32_t n_terms, const uint64_t *c,
                                const uint32_t *s_masks, const uint64_t *modulus, uint32_t limbs, int32_t device,
                                uint64_t *msgs_out, uint64_t *randomness_out);

/* MLSumcheck::prove_as_subprotocol with rand_poly_comb_fn (src/sumcheck/utils.rs:67-78), the workload of
 * benches/sumcheck_benches.rs: sum_p coeffs[p] * prod_{j in masks[p]} vals[j] (bit j of masks[p]: MLE j is a factor;
 * 1..4 factors per product, up to 32 MLEs).  coeffs: n_products field elements (Montgomery). */
int32_t zinc_sumcheck_prove_products(zinc_transcript *transcript, const uint64_t *const *mles, uint32_t n_mles,
                                     uint32_t nvars, uint32_t degree, uint32_t n_products, const uint64_t *coeffs,
                                     const uint32_t *masks, const uint64_t *modulus, uint32_t limbs, int32_t device,
                                     uint64_t *msgs_out, uint64_t *randomness_out);

/* MLSumcheck::verify_as_subprotocol (src/sumcheck.rs:116-160) with check_and_generate_subclaim
 * (src/sumcheck/verifier.rs:97-143), on the host (O(nvars * degree) field operations).  msgs: n_rounds messages of
 * evals_per_round elements each.  point_out: nvars elements, expected_out: one (either may be NULL).
 * ZINC_ERR_SPARTAN: SumCheckFailed / MaxDegreeExceeded / InvalidProofLength (zinc_last_error() says which). */
#define ZINC_ERR_SPARTAN (-6)
int32_t zinc_sumcheck_verify(zinc_transcript *transcript, uint32_t nvars, uint32_t degree, const uint64_t *claimed_sum,
                             const uint64_t *msgs, uint32_t n_rounds, uint32_t evals_per_round, const uint64_t *modulus,
                             uint32_t limbs, uint64_t *point_out, uint64_t *expected_out);

/* ZincProver (src/zinc/prover.rs): Prover::prove (:50-88) when with_pcs != 0, else
 * prepare_for_random_field_piop + SpartanProver::prove (:130-161, what benches/spartan_benches.rs times).
 *   constraints   Statement_Z.constraints as CSR (zip_sparse_matrix, include/zip_hip.h), t matrices of
 *                 2^s columns; CCS_Z {m = n = 2^s, s = s_prime = s, d, S = s_masks (bit j: matrix j), c}
 *   public_input / w_ccs   Statement_Z.public_input (l entries), Witness_Z.w_ccs: z = x || 1 || w
 *   outputs       SpartanProof.linearization_sumcheck: s * (d + 2) elements; .second_sumcheck: s * 3;
 *                 .V_s: t; r_y: s elements (all Montgomery limbs); ZipProof through the last argument when with_pcs != 0
 * ZINC_ERR_PANIC where the reference panics (shapes the prover's own assertions reject).
 *
 * zinc_prover_prepare: the circuit's matrices mapped to F_q and resident in HBM (what
 * prepare_for_random_field_piop recomputes per proof, prover.rs:188-189, depends only on circuit and field).
 * Pass the handle as `prepared` (then `constraints` may be NULL) to every proof of that circuit; NULL = build
 * and drop it inside the call, as the reference does. */
typedef struct zinc_prepared_ccs zinc_prepared_ccs;
int32_t zinc_prover_prepare(const zip_sparse_matrix *constraints, uint32_t t, uint32_t s, const uint64_t *modulus,
                            uint32_t limbs, int32_t device, zinc_prepared_ccs **out);
void zinc_prepared_ccs_free(zinc_prepared_ccs *p);
int32_t zinc_prover_prove(const zip_sparse_matrix *constraints, uint32_t t, uint32_t s, uint32_t d, uint32_t q,
                          const uint32_t *s_masks, const int64_t *c, const int64_t *public_input, size_t l,
                          const int64_t *w_ccs, size_t w_len, zinc_transcript *transcript, const uint64_t *modulus,
                          uint32_t limbs, int32_t device, zinc_prepared_ccs *prepared, int32_t with_pcs,
                          uint64_t *msgs1_out, uint64_t *msgs2_out, uint64_t *v_s_out, uint64_t *r_y_out,
                          zinc_zip_proof **zip_proof_out);

/* ZincVerifier (src/zinc/verifier.rs): SpartanVerifier::verify (:105-139), and with with_pcs != 0 also
 * verify_pcs_proof (:221-273: RaaCode::new from the transcript, MultilinearZip::verify, the matrix MLEs at
 * (r_x, r_y) on the device, the final equation) -- Verifier::verify without its draw_random_field check.
 * Proof fields as zinc_prover_prove returns them.  rx_ry_out: 2 s elements; e_y_out, gamma_out: one element (any
 * may be NULL).  ZINC_ERR_SPARTAN: a sumcheck or the final equation failed; ZINC_ERR_INVALID_OPEN: the Zip
 * verifier rejected; zinc_last_error() has the message. */
int32_t zinc_verifier_verify(const zip_sparse_matrix *constraints, uint32_t t, uint32_t s, uint32_t d, uint32_t q,
                             const uint32_t *s_masks, const int64_t *c, zinc_transcript *transcript, const uint64_t *modulus,
                             uint32_t limbs, int32_t device, zinc_prepared_ccs *prepared, const uint64_t *msgs1,
                             const uint64_t *msgs2, const uint64_t *v_s, int32_t with_pcs, const uint8_t *roots,
                             size_t n_roots, const uint64_t *v, const uint8_t *pcs_proof, size_t pcs_proof_len,
                             uint64_t *rx_ry_out, uint64_t *e_y_out, uint64_t *gamma_out);

#ifdef __cplusplus
}
#endif
#endif
