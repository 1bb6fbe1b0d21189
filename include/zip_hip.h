/*
 * zip_hip.h -- C ABI of libzip_hip.so: the MI355X (gfx950) implementation of the
 * Zip PCS commit/open hot path of NethermindEth/zinc.
 *
 * This is the drop-in boundary.  A thin Rust shim inside
 *   zinc::zip::pcs::MultilinearZip::{commit, commit_no_merkle, encode_rows, open}
 *   (src/zip/pcs/commit.rs:50,104,158; src/zip/pcs/open_z.rs:22)
 * binds these symbols (see INTEGRATION.md for the `extern "C"` block) and keeps
 * ZincProver / ZincVerifier unchanged.  Everything that is sequential and tiny
 * stays on the host side of the boundary: the Keccak Fiat-Shamir transcript
 * (src/transcript.rs), the rand-based permutation expansion
 * (src/zip/utils.rs:139-142), the eq-tensor q_0 (src/zip/pcs/utils.rs:279-292) and
 * the absorption of the evaluation row (src/zip/pcs_transcript.rs:107-113).
 *
 * Conventions
 *  - plain C types only; every call returns int32 (ZIP_OK == 0, negative = error)
 *  - no exceptions / unwinding cross the boundary; zip_strerror() names a code and
 *    zip_ctx_last_error() returns the detailed message of the last failing call
 *  - one zip_ctx per GPU and per geometry.  Several GPUs: either ONE process drives them all through the
 *    multi-device context below -- row shards behind one call, the roots gathered with in-process RCCL --, or one process
 *    per GPU gives every rank its own ctx with a row shard and exchanges roots / partial rows with RCCL above
 *    this ABI (zinc_amd/dist.py)
 *  - all work is enqueued on the ctx's HIP stream; calls that write HOST memory
 *    return after the data has landed, calls that only touch DEVICE memory return
 *    asynchronously (use zip_ctx_synchronize)
 *  - integers are little-endian arrays of 64-bit limbs in two's complement
 *    (src/field/int.rs:23-25); field elements are little-endian limb arrays of the
 *    Montgomery representation (src/field.rs:24-32)
 *  - a pointer argument with a zip_mem_kind may be host or device memory
 */
#ifndef ZIP_HIP_H
#define ZIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZIP_HIP_ABI_VERSION 3 /* 3: zip_ctx_set_speculation, zip_open_shard, zip_mctx_roots, zip_mctx_roots_path */

/* status codes */
#define ZIP_OK 0
#define ZIP_ERR_INVALID_PARAM (-1) /* zip::Error::InvalidPcsParam (src/zip/pcs/utils.rs:17-39) and the
                                      RaaCode width assertion (src/zip/code_raa.rs:68-72) */
#define ZIP_ERR_SHAPE (-2)         /* where the reference panics on a shape mismatch (commit.rs:56-63) */
#define ZIP_ERR_HIP (-3)           /* a HIP runtime call failed */
#define ZIP_ERR_NO_DEVICE (-4)     /* no usable gfx950 device: there is NO CPU fallback */
#define ZIP_ERR_UNSUPPORTED (-5)   /* geometry outside what the kernels implement */
#define ZIP_ERR_ALLOC (-6)
#define ZIP_ERR_NULL (-7)

typedef enum { ZIP_MEM_HOST = 0, ZIP_MEM_DEVICE = 1 } zip_mem_kind;

typedef struct zip_ctx zip_ctx;
typedef struct zip_commitment zip_commitment;

/* Geometry of one polynomial size: what RaaCode::new (src/zip/code_raa.rs:35-86) and
 * MultilinearZip::setup (src/zip/pcs/structs.rs:79-91) compute, plus the two
 * permutations that shuffle_seeded(perm_k_seed) applies (out[j] = in[perm[j]]). */
typedef struct {
    uint32_t num_vars;     /* log2(poly_size) */
    uint32_t row_len;      /* next_pow2(isqrt(poly_size))            code_raa.rs:43  */
    uint32_t num_rows;     /* next_pow2(poly_size / row_len)         structs.rs:82   */
    uint32_t codeword_len; /* row_len * rep                          code_raa.rs:113 */
    uint32_t rep;          /* repetition factor (power of two)       code.rs:235-237 */
    uint32_t n_limbs;      /* ZipTypes::N limbs, must be 1           traits/types.rs:225-240 */
    uint32_t k_limbs;      /* ZipTypes::K limbs, must be 4 */
    uint32_t m_limbs;      /* ZipTypes::M limbs, must be 8 */
    const uint32_t *perm1; /* HOST, codeword_len entries, a permutation of [0, codeword_len) */
    const uint32_t *perm2; /* HOST, codeword_len entries */
    int32_t device;        /* HIP device ordinal */
    uint32_t row_begin;    /* first row of this ctx's shard (0 for a whole polynomial) */
    uint32_t row_count;    /* rows in the shard; 0 means num_rows */
} zip_params;

/* Field configuration (src/field/config.rs:30-50).  Only the modulus crosses the
 * boundary; R^2 and -q^-1 mod 2^64 are recomputed (config.rs:174-214). */
typedef struct {
    uint32_t limbs;      /* FIELD_LIMBS: 2, 3 or 4 */
    uint64_t modulus[8]; /* little-endian limbs, odd */
} zip_field;

int32_t zip_abi_version(void);
const char *zip_strerror(int32_t code);
/* number of visible HIP devices (0 when there is no GPU); never initialises a context */
int32_t zip_device_count(void);
/* zip_sumcheck / zip_ccs handles live for one proof; their device blocks are kept (per device, at most 4 GiB)
 * for the next handle instead of going back to hipFree / hipMalloc.  This returns them to the driver. */
void zip_release_cached_memory(void);
/* Host buffers the caller pins once (a reused proof or witness buffer) are reached by the DMA engines directly:
 * every host transfer of this library first looks whether its buffer is pinned and only otherwise goes through the
 * library's pinned bounce buffers and a host-side copy (~33 GB/s instead of PCIe's ~55).  ZIP_ERR_ALLOC when the
 * range cannot be pinned (locked-memory limit); nothing else changes then. */
int32_t zip_host_register(void *p, size_t bytes);
void zip_host_unregister(void *p);

/* Threading: calls on one ctx (and on its commitments) are serialised inside the library -- they
 * share one pinned staging buffer and one set of streams; distinct contexts run concurrently. */
int32_t zip_ctx_create(const zip_params *params, zip_ctx **out);
void zip_ctx_destroy(zip_ctx *ctx);
const char *zip_ctx_last_error(const zip_ctx *ctx);
int32_t zip_ctx_synchronize(zip_ctx *ctx);
/* the ctx's hipStream_t, for callers that order their own device work against it */
void *zip_ctx_stream(zip_ctx *ctx);

/* ---- commit ------------------------------------------------------------------
 * MultilinearZip::commit (commit.rs:50-87) when with_merkle != 0,
 * commit_no_merkle / encode_rows (commit.rs:104-119,158-183) when with_merkle == 0.
 * evals: the shard's row_count * row_len witness values (Int<1>), row-major.
 * n_evals must equal row_count * row_len (ZIP_ERR_SHAPE otherwise, commit.rs:56-63).
 * roots_out: HOST, row_count * 32 bytes, may be NULL.
 * The returned handle owns the device-resident MultilinearZipData. */
int32_t zip_commit(zip_ctx *ctx, const int64_t *evals, size_t n_evals, zip_mem_kind evals_kind,
                   int32_t with_merkle, uint8_t *roots_out, zip_commitment **out);
void zip_commitment_free(zip_commitment *c);

/* zip_commit (with Merkle trees) for a caller that already knows which columns it is going to open.  In the
 * prover flow it does: ZincProver::commit_z_mle_and_prove_evaluation (src/zinc/prover.rs:305-328) opens with a
 * FRESH PcsTranscript (prover.rs:316) and write_integers / write_merkle_proof never absorb
 * (src/zip/pcs_transcript.rs:115-135,198-211), so the column indices of open (src/zip/pcs/open_z.rs:116-120) are
 * a function of the field configuration alone and the shim can squeeze them before it commits.
 * cols: HOST, n_cols indices < codeword_len (duplicates allowed).
 * Same roots and the same handle semantics as zip_commit; the hint only lets the commit kernel skip the HBM
 * stores no opening of those columns can read (about three quarters of the encoded rows and tree nodes at 1000
 * of 8192 columns) and put what they do read of the entries and of tree levels 0..2 into one dense block per row.
 * Whatever else is later asked of the handle -- an opening of any other column list (codeword_len >= 512: of
 * anything but exactly `cols`, in that order), zip_commit_download, the rows / layers device pointers -- first
 * re-runs the commit in full from the witness, transparently; for that a DEVICE
 * `evals` must stay valid and unchanged until the handle is freed.  Geometries below codeword_len 512 ignore
 * the hint. */
/* zip_commit and the opening hint: a ctx whose zip_open / zip_open_stream has named a column list hints its NEXT plain
 * zip_commit calls (with_merkle != 0) with that list on its own -- in the prover's flow (commit, then open on a fresh
 * PcsTranscript: src/zinc/prover.rs:315-320) the columns never change, so the two unchanged calls run at the hinted
 * commit's speed and produce the same bytes.  The handle's semantics stay zip_commit's: whatever it is asked for that
 * the hint did not keep completes it first (a transparent re-run of the commit from the witness).
 *   default      only commits of a HOST witness speculate (the Rust binding's case: poly.evaluations): the re-run reads
 *                the library's own device copy, nothing changes for the caller's buffers;
 *   on != 0      commits of a DEVICE witness speculate too.  LIFETIME RULE the caller accepts with this call: the device
 *                witness of such a commit must stay valid and unchanged until its handle is freed -- a re-run reads it
 *                again (from the witness zip_open is handed, or from `evals` of the commit, whose digest, taken beside
 *                the commit kernel, must then still match: ZIP_ERR_INVALID_PARAM otherwise);
 *   on == 0      no speculation on this ctx.
 * ZIP_HIP_SPECULATE=0 / 1 makes "off" / "device witnesses too" the process-wide default. */
int32_t zip_ctx_set_speculation(zip_ctx *ctx, int32_t on);

int32_t zip_commit_hinted(zip_ctx *ctx, const int64_t *evals, size_t n_evals, zip_mem_kind evals_kind,
                          const uint32_t *cols, uint32_t n_cols, uint8_t *roots_out, zip_commitment **out);

/* Device views of the handle (zero-copy interop).  rows: row_count*cw*4 u64 -- asking for them expands the
 * 16-byte entries a commitment keeps internally into this array once (a copy of row_count*cw*32 bytes on the
 * device); pass rows == NULL if only the layers / roots are wanted.
 * layers: per row 2*cw hashes of 32 B (level k at hash offset 2cw-(2cw>>k), root at
 * 2cw-2, last slot padding).  roots: row_count*32 B.  Any out pointer may be NULL. */
int32_t zip_commitment_device_ptrs(zip_commitment *c, uint64_t **rows, uint8_t **layers, uint8_t **roots);
/* Host copies in the reference's layout: rows_out row_count*cw*k_limbs u64
 * (MultilinearZipData.rows), layers_out row_count * ((2<<depth)-2) * 32 B
 * (MerkleTree.layers of every row, root popped), roots_out row_count*32 B. */
int32_t zip_commit_download(zip_commitment *c, uint64_t *rows_out, uint8_t *layers_out, uint8_t *roots_out);
/* Build a handle from host data in the layout above (the reference's tests mutate
 * MultilinearZipData before opening: commit.rs:389, open_z.rs:230-241). */
int32_t zip_commitment_upload(zip_ctx *ctx, const uint64_t *rows, const uint8_t *layers, const uint8_t *roots,
                              zip_commitment **out);

/* ---- open ----------------------------------------------------------------------
 * The three parts of MultilinearZip::open (open_z.rs:22-40) with the transcript
 * outputs as inputs: coeffs = get_integer_challenges(num_rows) (open_z.rs:104),
 * cols = the squeezed column indices (open_z.rs:118), q0_mont = the eq tensor of the
 * last log2(num_rows) point coordinates (pcs/utils.rs:279-292).  Shard-local slices
 * (row_count entries) when the ctx is a row shard. */

/* prove_testing_phase, proximity row (open_z.rs:103-112): uprime_out row_len * m_limbs u64. */
int32_t zip_open_testing(zip_ctx *ctx, const int64_t *evals, zip_mem_kind evals_kind, const int64_t *coeffs,
                         uint64_t *uprime_out, zip_mem_kind out_kind);
/* open_merkle_trees_for_column for n_cols columns (open_z.rs:116-120,124-143): wire_out gets
 * n_cols * row_count * (8*k_limbs + 8 + 32*depth) bytes in proof-stream format. */
int32_t zip_open_columns(zip_commitment *c, const uint32_t *cols, uint32_t n_cols, uint8_t *wire_out,
                         zip_mem_kind out_kind);
/* prove_evaluation_phase (open_z.rs:62-91): row_out row_len * field->limbs u64 (Montgomery limbs).
 * When num_rows == 1 q0_mont is ignored and row_out = map_to_field(evals).
 *
 * PARITY UNPINNED -- the signed-modulus quirk.  For a modulus q whose top bit is set in its limb width with
 * 2^(64 limbs) - q < 2^64 (e.g. q = 2^256 - 189, benches/spartan_benches.rs:134-137) the kernels first reduce |w|
 * modulo 2^(64 limbs) - q (`quirk_mod`), because the builder READS src/field.rs:550-557 as doing so: `F::I = Int<N>`
 * makes the modulus a NEGATIVE Int inside map_to_field's `%=`.  That reading of crypto-bigint 0.6's `Int::rem`
 * sign semantics has no published vector and could not be run here (no cargo); it changes results only for such
 * moduli (never for the 256-bit modulus of benches/zip_benches.rs:253: its top bit is set too, but 2^256 - q there
 * is far above 2^64, so a 64-bit |w| is never touched), it is applied
 * identically in zip_open_eval, zip_open, zip_verify, zip_mle_eval and zip_field_map_int256, and
 * integration/rust/fixture_dump.rs + tests/test_rust_fixtures.py pin it the first time a maintainer runs
 * `cargo test fixture_dump`. */
int32_t zip_open_eval(zip_ctx *ctx, const int64_t *evals, zip_mem_kind evals_kind, const uint64_t *q0_mont,
                      const zip_field *field, uint64_t *row_out, zip_mem_kind out_kind);
/* proof length in bytes for n_cols openings (commit.rs:712-737) */
size_t zip_proof_len(const zip_ctx *ctx, uint32_t n_cols, uint32_t field_limbs);
/* Whole proof stream of open() in one call, witness read once:
 *   [u' : row_len*m_limbs*8 B, only if num_rows > 1] [n_cols column openings] [row_len field
 *   elements, big-endian bytes of the Montgomery value (pcs_transcript.rs:107-113)].
 * The caller still absorbs the field elements into its transcript afterwards.
 * Only valid on an unsharded ctx. */
int32_t zip_open(zip_commitment *c, const int64_t *evals, zip_mem_kind evals_kind, const int64_t *coeffs,
                 const uint32_t *cols, uint32_t n_cols, const uint64_t *q0_mont, const zip_field *field,
                 uint8_t *proof_out, zip_mem_kind out_kind);

/* The open of ONE ROW SHARD in one call (ctx created with row_begin / row_count; SURVEY.md 8e, one process per GPU):
 * one pass over the shard's rows of the witness for BOTH partial row combinations -- partial u' (row_len * m_limbs
 * u64) and partial evaluation row (row_len * limbs u64, Montgomery limbs) over the shard's rows, to be all-gathered
 * and added with zip_sum_partials -- and the shard's rows of every opened column in wire order
 * ([n_cols][row_count * 32 B values | row_count records]), pipelined behind the shard's commit kernel.
 * coeffs / q0_mont: the shard's slices (row_count entries), HOST; everything else DEVICE.  Returns when the device work
 * has finished. */
int32_t zip_open_shard(zip_commitment *c, const int64_t *evals_d, const int64_t *coeffs, const uint32_t *cols, uint32_t n_cols,
                       const uint64_t *q0_mont, const zip_field *field, uint64_t *uprime_part_d, uint64_t *row_part_d,
                       uint8_t *wire_d);

/* commit + open in ONE call: what ZincProver::commit_z_mle_and_prove_evaluation (src/zinc/prover.rs:305-328) does with
 * its two calls, commit (commit.rs:50-87) and open (open_z.rs:22-40) on a fresh PcsTranscript -- the binding for that
 * function body.  Same roots and byte-identical proof stream as zip_commit followed by zip_open; internally
 * zip_commit_hinted + the pipelined open.
 *   roots_out  HOST, num_rows * 32 bytes, may be NULL
 *   proof_out  zip_proof_len() bytes, HOST or DEVICE per out_kind
 *   out        may be NULL (the handle is then freed); otherwise a handle like zip_commit_hinted's: anything later
 *              asked of it that is not in it first re-runs the commit in full, transparently -- a DEVICE `evals` must
 *              stay valid and unchanged until the handle is freed.
 * Unsharded ctx only. */
int32_t zip_commit_open(zip_ctx *ctx, const int64_t *evals, size_t n_evals, zip_mem_kind evals_kind, const int64_t *coeffs,
                        const uint32_t *cols, uint32_t n_cols, const uint64_t *q0_mont, const zip_field *field,
                        uint8_t *roots_out, uint8_t *proof_out, zip_mem_kind out_kind, zip_commitment **out);

/* zip_commit_open as a JOB, for a caller that proves several polynomials in a row (the reference's batch paths are
 * loops over polynomials, commit.rs:134-142, open_z.rs:43-58): _begin enqueues the whole commit + open and returns,
 * zip_job_wait collects it.  With a second _begin made before the first job is waited for, the commit kernel of the
 * second polynomial starts the moment the first one's ends and runs beside the tail of the first one's openings
 * (the last chunk's gather, which has the GPU to itself otherwise) -- about 10 % more proofs per second at 2^24.
 * At most TWO jobs per ctx are in flight; a third _begin returns ZIP_ERR_INVALID_PARAM.
 *   evals_d, proof_out_d   DEVICE memory, valid and untouched until the job has been waited for
 *   coeffs, cols, q0_mont  HOST, consumed before _begin returns
 *   roots_out              HOST, num_rows * 32 bytes, or NULL
 * zip_job_wait frees the job whatever it returns.  Unsharded ctx only. */
typedef struct zip_job zip_job;
int32_t zip_commit_open_begin(zip_ctx *ctx, const int64_t *evals_d, size_t n_evals, const int64_t *coeffs,
                              const uint32_t *cols, uint32_t n_cols, const uint64_t *q0_mont, const zip_field *field,
                              uint8_t *proof_out_d, zip_job **job);
int32_t zip_job_wait(zip_job *job, uint8_t *roots_out);

/* ---- streaming proof writer -------------------------------
 * zip_open with the stream delivered to `sink` in order, piece by piece (u', then groups of opened
 * columns, then the evaluation row), instead of into one contiguous buffer: the 1.74 GiB proof of a
 * 2^24 witness never needs a single host allocation, and the PCIe copy of one group overlaps the
 * gather of the next.  `bytes` points into pinned memory owned by the library and is valid only
 * during the call; a non-zero return from the sink aborts (ZIP_ERR_INVALID_PARAM).
 * chunk_bytes: target size of a column group (0 = 64 MiB).  The concatenation of all pieces is
 * byte-identical to what zip_open writes. */
typedef int32_t (*zip_proof_sink)(void *user, const uint8_t *bytes, size_t len);
int32_t zip_open_stream(zip_commitment *c, const int64_t *evals, zip_mem_kind evals_kind, const int64_t *coeffs,
                        const uint32_t *cols, uint32_t n_cols, const uint64_t *q0_mont, const zip_field *field,
                        zip_proof_sink sink, void *user, size_t chunk_bytes);

/* ---- verifier side ---------------------------------------------
 * MultilinearZip::verify (src/zip/pcs/verify_z.rs:19-188) on the device.  The caller (the Rust
 * shim / zinc_amd/host) keeps the Fiat-Shamir work: it squeezes `coeffs` (num_rows, only when
 * num_rows > 1) and the `cols`, builds (q0, q1) = point_to_tensor (pcs/utils.rs:252-276) and, after
 * the call, absorbs the row_len field elements at the end of the stream
 * (read_field_elements, pcs_transcript.rs:138-160).
 *   roots      HOST, num_rows * 32 bytes (MultilinearZipCommitment.roots)
 *   proof      the stream `open` wrote, zip_proof_len() bytes (longer is allowed, shorter is malformed)
 *   q1_mont    row_len field elements (may be NULL when row_len == 1: q_1 is empty there)
 *   eval_mont  the claimed evaluation, Montgomery limbs
 * report->verdict is the first check that fails in the reference's order, ZIP_VERIFY_ACCEPT if none.
 * Deliberate differences from the reference, both on the rejecting side:
 *  - Merkle paths ARE checked (the reference computes the check and drops the result,
 *    verify_z.rs:99, but then loses its place in the stream at the first bad path);
 *  - evaluation-row elements >= q that survive the evaluation-consistency check (which is
 *    representation independent and comes first, as in the reference) are rejected as malformed; the
 *    reference does not range-check them and its sequential modular additions in encode_f then
 *    depend on the representation.
 * The function result is only non-zero for usage / device errors. */
typedef enum {
    ZIP_VERIFY_ACCEPT = 0,
    ZIP_VERIFY_PROXIMITY_TESTING = 1, /* "Proximity failure", verify_z.rs:122-125 */
    ZIP_VERIFY_EVAL_CONSISTENCY = 2,  /* "Evaluation consistency failure", verify_z.rs:145-149 */
    ZIP_VERIFY_PROXIMITY_Q0 = 3,      /* "Proximity failure", verify_z.rs:184-186 */
    ZIP_VERIFY_MERKLE = 4,            /* a path does not hash to its row root */
    ZIP_VERIFY_MALFORMED = 5,         /* short stream, wrong path length prefix, non-canonical field element */
    ZIP_VERIFY_OVERFLOW = 6           /* encode_wide overflows Int<M>: the reference panics (int.rs:122-134) */
} zip_verify_verdict;
typedef struct {
    int32_t verdict;
    uint32_t column;           /* index into cols[] of the failing opening, when the verdict names one */
    uint32_t bad_merkle_paths; /* over all openings */
    uint32_t malformed_paths;
} zip_verify_report;
int32_t zip_verify(zip_ctx *ctx, const uint8_t *roots, const uint8_t *proof, zip_mem_kind proof_kind, size_t proof_len,
                   const int64_t *coeffs, const uint32_t *cols, uint32_t n_cols, const uint64_t *q0_mont,
                   const uint64_t *q1_mont, const uint64_t *eval_mont, const zip_field *field,
                   zip_verify_report *report);

/* ---- sumcheck prover --------------------------------------------
 * IPForMLSumcheck::prove_round (src/sumcheck/prover.rs:62-180) for the two combination functions
 * ZincProver uses -- comb == NULL: comb_fn(vals) = vals[0] * vals[1] * ... (second sumcheck,
 * src/zinc/prover.rs:297-302); comb != NULL: sumcheck_polynomial_comb_fn_1 (src/zinc/utils.rs:77-94),
 *   (sum_t coeff[t] * prod_{j in term_mask[t]} vals[j]) * vals[n_mles - 1]
 * with coeff = ccs.c (Montgomery limbs), term_mask[t] = the bits of ccs.S[t] (positions in the MLE list of
 * prepare_lin_sumcheck_polynomial, zinc/utils.rs:49-75) and the eq() MLE last -- one call per round; the
 * caller keeps the
 * transcript (MLSumcheck::prove_as_subprotocol, src/sumcheck.rs:56-112: absorb the evaluations, squeeze
 * the challenge, absorb it, hand it to the next round).
 *   mles      n_mles (1..4) tables of 2^num_vars field elements, Montgomery limbs, variable 0 = least
 *             significant index bit; HOST tables are copied, DEVICE tables are read in place (never
 *             written) and must outlive the handle
 *   degree    1..4: the round polynomial is returned as its values at 0..degree (ProverMsg.evaluations)
 *   r_prev    the verifier's challenge for the previous round (NULL in round 1): the tables are folded
 *             with it (fix_variables, src/poly_f/mle/dense.rs:142-168) in the same pass
 *   evaluations_out  HOST, (degree + 1) * field->limbs limbs */
typedef struct zip_sumcheck zip_sumcheck;
typedef struct {
    uint32_t n_terms;      /* 1..8 */
    uint32_t term_mask[8]; /* bit j: MLE j is a factor of the term */
    uint64_t coeff[8][8];  /* Montgomery limbs, little-endian, unused limbs zero */
} zip_sumcheck_comb;
int32_t zip_sumcheck_init(int32_t device, const uint64_t *const *mles, zip_mem_kind kind, uint32_t n_mles,
                          uint32_t num_vars, uint32_t degree, const zip_sumcheck_comb *comb, const zip_field *field,
                          zip_sumcheck **out);
int32_t zip_sumcheck_round(zip_sumcheck *s, const uint64_t *r_prev, uint64_t *evaluations_out);
/* The same round in two halves -- enqueue, then collect -- so that several handles (the products of a
 * sum-of-products polynomial, src/sumcheck/utils.rs:27-78: each product folds its own MLEs, the round messages
 * add up) work at the same time on their own streams. */
int32_t zip_sumcheck_round_begin(zip_sumcheck *s, const uint64_t *r_prev);
int32_t zip_sumcheck_round_end(zip_sumcheck *s, uint64_t *evaluations_out);
const char *zip_sumcheck_last_error(const zip_sumcheck *s);
void zip_sumcheck_free(zip_sumcheck *s);

/* ---- the field loops of SpartanProver::prove around its sumchecks (BASELINE configs[4]) ---------
 * A zip_ccs holds the constraint matrices of a CCS (Statement_Z.constraints, src/ccs/ccs_z.rs:155-158,
 * mapped to F_q as SparseMatrix::map_to_field does, src/sparse_matrix.rs:38-58) and the per-proof
 * tables in HBM.  With zip_sumcheck_* (tables handed over as ZIP_MEM_DEVICE) this is everything
 * src/zinc/prover.rs:130-161 computes; the caller keeps the transcript.
 *   zip_sparse_matrix   SparseMatrix<Int<1>> as CSR: row_ptr[n_rows + 1], col_idx / values[row_ptr[n_rows]].
 *                       n_cols must be 2^s and n_rows <= 2^s (missing rows are zero: pad_rows,
 *                       src/sparse_matrix.rs:104-108); else ZIP_ERR_SHAPE, where the reference panics
 *                       or returns LengthsNotEqual (src/ccs/utils.rs:52-59).
 *   zip_ccs_set_z       z = x || 1 || w (Statement_Z::get_z_vector), z_len <= 2^s, zero-extended
 *                       (prover.rs:230-232): builds z_ccs in F_q (:236) and the t tables M_k z
 *                       (calculate_Mz_mles, src/zinc/utils.rs:121-135)
 *   zip_ccs_eq_table    build_eq_x_r(r) (src/sumcheck/utils.rs:102-177), r = s Montgomery elements on
 *                       the HOST; slot 0 is for eq(beta) (prepare_lin_sumcheck_polynomial,
 *                       zinc/utils.rs:72), slot 1 for eq(r_x)
 *   zip_ccs_second_table  what sumcheck_2 needs (prover.rs:261-296): eq(r_x) into slot 1, the table
 *                       sum_k gamma^k * compute_eval_table_sparse(M_k, eq(r_x)) (src/sparse_matrix.rs:165-182),
 *                       and V_s[k] = (M_k z)(r_x) (calculate_V_s, prover.rs:330-347) to v_s_out (HOST, t elements)
 *   zip_ccs_eval_matrices  the verifier's V_xy (src/zinc/verifier.rs:248-261): mle[M_k](r_x, r_y) for every matrix,
 *                       DenseMultilinearExtension::from_matrix (src/poly_f/mle/dense.rs:69-87) evaluated without
 *                       building the dense 2^(2s) table; r_x, r_y: s elements each on the HOST; overwrites both eq slots
 *   zip_ccs_table       device pointer of a table of 2^s elements, valid until the next call that
 *                       rebuilds it or zip_ccs_free; index = matrix for ZIP_CCS_MZ, slot for ZIP_CCS_EQ
 *   zip_ccs_download    the same table copied to the HOST (tests) */
typedef struct zip_ccs zip_ccs;
typedef struct {
    uint32_t n_rows, n_cols;
    const uint32_t *row_ptr, *col_idx;
    const int64_t *values;
} zip_sparse_matrix;
typedef enum { ZIP_CCS_Z_FIELD = 0, ZIP_CCS_MZ = 1, ZIP_CCS_EQ = 2, ZIP_CCS_SECOND = 3 } zip_ccs_table_kind;
int32_t zip_ccs_create(int32_t device, const zip_sparse_matrix *matrices, uint32_t t, uint32_t s, const zip_field *field,
                       zip_ccs **out);
void zip_ccs_free(zip_ccs *c);
const char *zip_ccs_last_error(const zip_ccs *c);
int32_t zip_ccs_set_z(zip_ccs *c, const int64_t *z, size_t z_len, zip_mem_kind kind);
int32_t zip_ccs_eq_table(zip_ccs *c, const uint64_t *r, uint32_t slot);
int32_t zip_ccs_second_table(zip_ccs *c, const uint64_t *r_x, const uint64_t *gamma, uint64_t *v_s_out);
int32_t zip_ccs_eval_matrices(zip_ccs *c, const uint64_t *r_x, const uint64_t *r_y, uint64_t *v_xy_out);
int32_t zip_ccs_table(zip_ccs *c, zip_ccs_table_kind which, uint32_t index, const uint64_t **table_dev);
int32_t zip_ccs_download(zip_ccs *c, zip_ccs_table_kind which, uint32_t index, uint64_t *out);

/* Diagnostic: FieldMap for Int<4> (src/conversion.rs:86-100) of n arbitrary 256-bit values, as the
 * verifier applies it to column entries.  values: HOST n*4 limbs; out: HOST n*limbs Montgomery limbs. */
int32_t zip_field_map_int256(zip_ctx *ctx, const uint64_t *values, uint32_t n, const zip_field *field, uint64_t *out);

/* z_mle.map_to_field(config).evaluate(r_y) (src/zinc/prover.rs:317-319, poly_f/mle/dense.rs:35-41),
 * the step ZincProver runs between commit and open: <q0-combination of the rows, q1>.
 * q0_mont: num_rows elements (NULL when num_rows == 1), q1_mont: row_len elements (NULL when
 * row_len == 1).  value_out: HOST, field->limbs Montgomery limbs. */
int32_t zip_mle_eval(zip_ctx *ctx, const int64_t *evals, zip_mem_kind evals_kind, const uint64_t *q0_mont,
                     const uint64_t *q1_mont, const zip_field *field, uint64_t *value_out);
/* The same over the witness a host-side zip_commit left on the device with the commitment: the prover's
 * z_mle.evaluate(r_y) between commit and open (src/zinc/prover.rs:315-320) without a second upload. */
int32_t zip_commitment_mle_eval(zip_commitment *c, const uint64_t *q0_mont, const uint64_t *q1_mont,
                                const zip_field *field, uint64_t *value_out);

/* Row-sharded open: exact sum of n_parts partial results (after an all-gather).
 * uparts: n_parts * row_len * m_limbs u64 or NULL; fparts: n_parts * row_len * limbs u64 or NULL.
 * All pointers DEVICE memory. */
int32_t zip_sum_partials(zip_ctx *ctx, const uint64_t *uparts, const uint64_t *fparts, uint32_t n_parts,
                         const zip_field *field, uint64_t *uprime_out, uint64_t *row_out);

/* ---- several GPUs behind one call ---------------------------------------
 * The reference's callers are one process making one call (src/zinc/prover.rs:305-328): a zip_mctx gives that
 * process all the GPUs of the node.  It owns one row-shard context per entry of `devices` (contiguous blocks of
 * rows; an ordinal may repeat -- several shards on one GPU).  `params->device / row_begin / row_count` are ignored.
 *
 * zip_mctx_commit_open = zip_commit_open on the whole polynomial, the same roots and the same proof bytes:
 *   - every shard commits its rows (hinted persistent kernel) and opens THEM for every column, pipelined as on
 *     one GPU; no shard waits for another;
 *   - the row combinations are computed per shard and added on the lead device (the first of `devices`): the
 *     only exchange besides the roots, 96 bytes per witness column and shard, peer copies -- exact, the sums are
 *     integer / modular;
 *   - proof_out != NULL (HOST): u', then every shard's rows of every column straight from that shard's memory
 *     over its own PCIe link (two pitched copies per shard), then the evaluation row.  Pin the buffer
 *     (zip_host_register) for full link speed;
 *   - proof_out == NULL: the pieces stay on the devices (zip_mctx_shard_openings, zip_mctx_ends).
 *   evals      HOST, the whole witness, or NULL to use the slices zip_mctx_set_witness left on the devices
 *   coeffs, q0_mont, cols   the WHOLE challenge vectors, as for zip_open
 *   roots_out  HOST, num_rows * 32 bytes, may be NULL */
typedef struct zip_mctx zip_mctx;
int32_t zip_mctx_create(const zip_params *params, int32_t n_devices, const int32_t *devices, zip_mctx **out);
void zip_mctx_destroy(zip_mctx *m);
const char *zip_mctx_last_error(const zip_mctx *m);
uint32_t zip_mctx_shards(const zip_mctx *m);
zip_ctx *zip_mctx_shard_ctx(zip_mctx *m, uint32_t shard); /* the shard's context (profiling hooks, geometry); owned by m */
int32_t zip_mctx_set_witness(zip_mctx *m, const int64_t *evals, size_t n_evals);
int32_t zip_mctx_commit_open(zip_mctx *m, const int64_t *evals, const int64_t *coeffs, const uint32_t *cols,
                             uint32_t n_cols, const uint64_t *q0_mont, const zip_field *field, uint8_t *roots_out,
                             uint8_t *proof_out);
/* after zip_mctx_commit_open: shard's openings on ITS device, [n_cols][row_count * 32 B values | row_count records] */
int32_t zip_mctx_shard_openings(zip_mctx *m, uint32_t shard, uint8_t **ptr, size_t *bytes, uint32_t *row_begin,
                                uint32_t *row_count);
/* after zip_mctx_commit_open: u' (u_bytes) followed by the evaluation row (row_bytes) on the lead device */
int32_t zip_mctx_ends(zip_mctx *m, uint8_t **ptr, size_t *u_bytes, size_t *row_bytes);
/* after zip_mctx_commit_open: the commitment -- the roots of ALL rows, [num_rows][32] -- as shard's device holds it.
 * The concatenation of the per-row roots (src/zip/pcs/commit.rs:78-81) is the one exchange of a row-sharded commit:
 * over distinct devices it is one grouped ncclAllGather (ncclBroadcast per owner for uneven blocks) on in-process
 * RCCL communicators (ncclCommInitAll; librccl is bound with dlopen when such a zip_mctx is created), over repeated
 * ordinals -- or when RCCL is missing or refuses -- device copies.  zip_mctx_roots_path: "rccl" | "copies" | "none". */
int32_t zip_mctx_roots(zip_mctx *m, uint32_t shard, uint8_t **ptr);
const char *zip_mctx_roots_path(const zip_mctx *m);

/* ---- standalone Merkle tree --------------------------------------------------------
 * MerkleTree::new (pcs/utils.rs:74-85) over num_trees * 2^depth leaves of leaf_limbs
 * (1..8) limbs each.  layers_out: num_trees * ((2<<depth)-1) * 32 B, root last
 * (the reference pops it into .root).  Buffers HOST or DEVICE per kind. */
int32_t zip_merkle_trees(int32_t device, const uint64_t *leaves, uint32_t leaf_limbs, uint32_t depth,
                         uint32_t num_trees, zip_mem_kind kind, uint8_t *layers_out);

/* ---- measurement hooks ---------------------------------------------------------------
 * With profiling on (1), every kernel launch is bracketed by HIP events on the stream it is launched on; with on = 2
 * only the commit / encode kernel is (it has its stream to itself: between the kernels of ONE stream the events cost
 * every dependent launch about 12 us, which a timed region need not pay).
 * zip_ctx_profile_read synchronises, then fills up to cap entries (kernel name, launch
 * count, total ms) and resets the counters; returns the number of distinct kernels. */
typedef struct {
    const char *name;
    uint32_t launches;
    float total_ms;
} zip_kernel_time;
int32_t zip_ctx_set_profiling(zip_ctx *ctx, int32_t on);
/* Shader clock (MHz) the chip held during the most recent commit kernel launched with profiling on: the kernel's
 * first workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end.  0 if none.
 * bench.py prices the VALU roofline of the commit kernel at this clock. */
int32_t zip_ctx_commit_clock(zip_ctx *ctx, double *mhz_out);
int32_t zip_ctx_profile_read(zip_ctx *ctx, zip_kernel_time *out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ZIP_HIP_H */
