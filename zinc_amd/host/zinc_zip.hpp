// zinc_zip.hpp -- C++ mirror of the host side of zinc::zip for the commit/open path.
//
// In the real integration this layer is the Rust crate itself: `ZincProver` keeps calling
// `MultilinearZip::{setup, commit, open}` and a shim inside those functions calls the C ABI of
// libzip_hip.so.  This image has no Rust toolchain, so the same host logic is
// restated here in C++ with the reference's names, argument meaning and error behaviour:
//
//   zinc::KeccakTranscript          src/transcript.rs
//   zinc::FieldConfig / FieldElem   src/field/config.rs, src/field.rs (RandomField::Initialized)
//   zinc::map_to_field              src/conversion.rs:86-100, src/field.rs:536-568
//   zinc::build_eq_x_r              src/sumcheck/utils.rs:117-177
//   zinc::zip::shuffle_seeded       src/zip/utils.rs:139-142 (rand 0.9 restated, PARITY UNPINNED)
//   zinc::zip::RaaCode              src/zip/code_raa.rs:16-140
//   zinc::zip::PcsTranscript        src/zip/pcs_transcript.rs
//   zinc::zip::MultilinearZip       src/zip/pcs/structs.rs, commit.rs, open_z.rs
//
// Everything sequential and tiny stays here on the host (Fiat-Shamir, permutation expansion,
// the eq tensor, absorbing the evaluation row); every O(n) loop is a call into the HIP library.
// There is no CPU implementation of commit/open in this layer.
#pragma once
#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <type_traits>
#include <utility>
#include <stdexcept>
#include <string>
#include <vector>

#include "zip_hip.h"

namespace zinc {

// Byte vector whose resize() leaves new bytes uninitialised: the proof stream of a 2^24 witness is
// 1.74 GiB that the device overwrites entirely, and value-initialising it first costs a full extra
// pass over fresh pages on the host.
template <class T>
struct default_init_allocator {
    using value_type = T;
    template <class U> struct rebind { using other = default_init_allocator<U>; };
    default_init_allocator() = default;
    template <class U> default_init_allocator(const default_init_allocator<U> &) {}
    // Large blocks are 2 MiB aligned and marked for transparent huge pages: first-touch faults
    // (the kernel zero-filling fresh pages) are what bounds the arrival of a proof in host memory.
    T *allocate(size_t n) { return static_cast<T *>(big_alloc(n * sizeof(T))); }
    void deallocate(T *p, size_t n) { byte_stream_free_dispatch(p, n * sizeof(T)); }
    static void byte_stream_free_dispatch(void *p, size_t bytes);
    template <class U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
    bool operator==(const default_init_allocator &) const { return true; }
    bool operator!=(const default_init_allocator &) const { return false; }
    static void *big_alloc(size_t bytes);
};
void *byte_stream_alloc(size_t bytes);
void byte_stream_free(void *p, size_t bytes);   // big blocks go to a small process-wide cache
void byte_stream_release_cache();               // returns that cache to the system
template <class T>
void *default_init_allocator<T>::big_alloc(size_t bytes) { return byte_stream_alloc(bytes); }
template <class T>
void default_init_allocator<T>::byte_stream_free_dispatch(void *p, size_t bytes) { byte_stream_free(p, bytes); }
using ByteStream = std::vector<uint8_t, default_init_allocator<uint8_t>>;
using IntVec = std::vector<int64_t, default_init_allocator<int64_t>>;  // witness-sized: huge pages, no zero fill

constexpr uint32_t kMaxLimbs = 8;
using Limbs = std::array<uint64_t, kMaxLimbs>;  // little-endian, unused limbs zero

// ---------------------------------------------------------------------------- errors
// zip::Error (src/zip.rs:29-41).  Where the reference panics, this layer throws std::logic_error.
struct ZipError : std::runtime_error {
    enum Kind { InvalidPcsParam, InvalidPcsOpen, Transcript, Device } kind;
    ZipError(Kind k, const std::string &what) : std::runtime_error(what), kind(k) {}
};

// ---------------------------------------------------------------------------- Keccak-256
class Keccak256 {
  public:
    Keccak256() { std::memset(st_, 0, sizeof st_); }
    void update(const uint8_t *data, size_t len);
    void update(std::initializer_list<uint8_t> bytes) { update(bytes.begin(), bytes.size()); }
    std::array<uint8_t, 32> finalize() const;  // of a clone: the hasher keeps absorbing

  private:
    static void permute(uint64_t st[25]);
    void absorb_block(const uint8_t *blk);
    uint64_t st_[25];
    uint8_t buf_[136];
    uint32_t buflen_ = 0;
};

// ---------------------------------------------------------------------------- field
// FieldConfig (src/field/config.rs:30-50): Montgomery constants for an odd modulus of `limbs` limbs.
struct FieldConfig {
    uint32_t limbs = 0;
    Limbs modulus{}, r{}, r2{};
    uint64_t inv = 0;
    bool modulus_has_spare_bit = false;

    static FieldConfig make(const uint64_t *modulus, uint32_t limbs);  // Config::new, config.rs:174-186
    void mul_assign(Limbs &a, const Limbs &b) const;                   // config.rs:163-170
    void add_assign(Limbs &a, const Limbs &b) const;                   // config.rs:53-58
    void sub_assign(Limbs &a, const Limbs &b) const;                   // config.rs:60-66
    void neg(Limbs &a) const;                                          // arithmetic.rs:130-149
    Limbs inverse(const Limbs &a) const;                               // a^(q-2): the unique inverse (q prime)
    uint32_t num_bits() const;
    zip_field to_abi() const;

  private:
    void reduce_modulus(Limbs &a, bool carry) const;  // config.rs:68-76
};

// FieldMap for i64 / u128 (conversion.rs:9-46,86-100; field.rs:536-568): Montgomery form.
Limbs map_to_field(const FieldConfig &f, int64_t v);
Limbs map_to_field_u128(const FieldConfig &f, uint64_t lo, uint64_t hi);
// build_eq_x_r_vec (sumcheck/utils.rs:117-177); r: nvars field elements, result 2^nvars.
std::vector<Limbs> build_eq_x_r(const FieldConfig &f, const Limbs *r, uint32_t nvars);

// ---------------------------------------------------------------------------- Fiat-Shamir
class KeccakTranscript {
  public:
    void absorb(const uint8_t *v, size_t len) { hasher_.update(v, len); }
    std::vector<uint8_t> get_random_bytes(size_t length);          // transcript.rs:40-55
    void absorb_random_field(const FieldConfig &f, const Limbs &v);  // field.rs:360-378 (Initialized)
    Limbs get_challenge(const FieldConfig &f);                     // transcript.rs:88-133
    void get_integer_challenge(uint32_t n_limbs, uint64_t *out);   // transcript.rs:142-155
    std::vector<int64_t> get_integer_challenges_i64(size_t n);     // transcript.rs:158-160 with I = Int<1>
    uint64_t get_u64() {                                           // ZipTranscript::get_u64, transcript.rs:183-185
        uint64_t w;
        get_integer_challenge(1, &w);
        return w;
    }

  private:
    Keccak256 hasher_;
};

namespace zip {

// shuffle_seeded applied to the identity: shuffle_seeded(x, seed)[j] == x[perm[j]].
std::vector<uint32_t> shuffle_seeded_perm(uint64_t seed, uint32_t len);
// the seed bytes rand_core's seed_from_u64 expands `seed` to, as little-endian words (known-answer hook)
void kat_seed_from_u64(uint64_t seed, uint32_t *words, uint32_t n_words);

// LinearCodeSpec / DefaultLinearCodeSpec (src/zip/code.rs:217-242)
struct LinearCodeSpec {
    uint32_t num_column_opening = 1000;
    uint32_t repetition_factor = 2;
    uint32_t num_proximity_testing = 1;
};

// A source of permutation seeds: ZipTranscript::get_u64 (structs.rs:67-76).
struct SeedSource {
    virtual ~SeedSource() = default;
    virtual uint64_t get_u64() = 0;
};
struct KeccakSeedSource : SeedSource {
    explicit KeccakSeedSource(KeccakTranscript &t) : t_(t) {}
    uint64_t get_u64() override { return t_.get_u64(); }
    KeccakTranscript &t_;
};
struct MockTranscript : SeedSource {  // src/zip/pcs/tests.rs:24-37
    int64_t counter = 0;
    uint64_t get_u64() override { return (uint64_t)++counter; }
};

// RaaCode (src/zip/code_raa.rs:16-86), ZipTypes fixed to INT_LIMBS = 1 (N = Int<1>, K = Int<4>, M = Int<8>).
struct RaaCode {
    uint32_t row_len = 0, repetition_factor = 0, num_column_opening = 0, num_proximity_testing = 0;
    uint64_t perm_1_seed = 0, perm_2_seed = 0;

    // RaaCode::new: geometry, width assertion (throws std::logic_error like the reference's assert!),
    // then two seeds from the transcript.
    static RaaCode make(const LinearCodeSpec &spec, uint64_t poly_size, SeedSource &transcript);
    uint32_t codeword_len() const { return row_len * repetition_factor; }
};

// PcsTranscript (src/zip/pcs_transcript.rs:19-48): Fiat-Shamir state + the proof byte stream.
struct PcsTranscript {
    KeccakTranscript fs_transcript;
    ByteStream stream;
    size_t read_pos = 0;  // Cursor position of the reading side (PcsTranscript::from_proof, :28-35)

    const uint8_t *borrowed = nullptr;  // reading side over the caller's bytes (from_proof_view): nothing is copied
    size_t borrowed_len = 0;

    static PcsTranscript from_proof(const uint8_t *proof, size_t len) {
        PcsTranscript t;
        t.stream.assign(proof, proof + len);
        return t;
    }
    // the same without taking a copy of a proof that may be GiBs; the bytes must outlive the transcript
    static PcsTranscript from_proof_view(const uint8_t *proof, size_t len) {
        PcsTranscript t;
        t.borrowed = proof;
        t.borrowed_len = len;
        return t;
    }
    const uint8_t *read_data() const { return borrowed ? borrowed : stream.data(); }
    size_t read_size() const { return borrowed ? borrowed_len : stream.size(); }

    void write_field_elements(const FieldConfig &f, const Limbs *elems, size_t n);  // :76-113
    void append(const uint8_t *bytes, size_t n) { stream.insert(stream.end(), bytes, bytes + n); }
    size_t squeeze_challenge_idx(const FieldConfig &f, size_t cap);                // :174-179
    ByteStream into_proof() { return std::move(stream); }
};

// MultilinearZipParams (structs.rs:13-18) + the device context that serves this geometry.
struct MultilinearZipParams {
    uint32_t num_vars = 0, num_rows = 0;
    RaaCode linear_code;
    std::vector<uint32_t> perm1, perm2;
    std::shared_ptr<zip_ctx> ctx;  // created by setup()
};

// MultilinearZipData (structs.rs:33-38): rows + Merkle trees, device resident behind the handle.
struct MultilinearZipData {
    std::shared_ptr<zip_ctx> ctx;             // keeps the device context alive (declared first: destroyed last)
    std::shared_ptr<zip_commitment> handle;
};
// MultilinearZipCommitment (structs.rs:42-45)
struct MultilinearZipCommitment {
    std::vector<std::array<uint8_t, 32>> roots;
};

struct MultilinearZip {
    // structs.rs:79-91.  Device contexts (streams, pinned staging, the device memory pool, the
    // uploaded permutation tables) are cached per (device, geometry, seeds): ZincProver calls setup
    // for every proof, and a cold context costs more than a 2^20 commit.  The most recent few stay
    // alive; release_cached_contexts() drops them (and their device memory).
    static MultilinearZipParams setup(uint64_t poly_size, const RaaCode &code, int device = 0);
    static void release_cached_contexts();
    // commit.rs:50-87.  poly_num_vars is DenseMultilinearExtension::num_vars of the caller's polynomial.
    static std::pair<MultilinearZipData, MultilinearZipCommitment> commit(const MultilinearZipParams &pp,
                                                                          const int64_t *evals, size_t n_evals,
                                                                          uint32_t poly_num_vars);
    // commit.rs:104-119
    static MultilinearZipData commit_no_merkle(const MultilinearZipParams &pp, const int64_t *evals, size_t n_evals,
                                               uint32_t poly_num_vars);
    // open_z.rs:22-40.  point: num_vars field elements in Montgomery form.
    static void open(const MultilinearZipParams &pp, const int64_t *evals, size_t n_evals, uint32_t poly_num_vars,
                     const MultilinearZipData &commit_data, const Limbs *point, size_t point_len,
                     const FieldConfig &field, PcsTranscript &transcript);
    // verify_z.rs:19-38.  Throws ZipError{InvalidPcsOpen} with the reference's message when a check
    // fails ("Proximity failure", "Evaluation consistency failure"), std::logic_error where the
    // reference panics (encode_wide overflow).  Merkle paths are checked (zip_hip.h, zip_verify).
    static void verify(const MultilinearZipParams &vp, const MultilinearZipCommitment &comm, const Limbs *point,
                       size_t point_len, const Limbs &eval, PcsTranscript &transcript, const FieldConfig &field);
    // z_mle.map_to_field(config).evaluate(r_y, config) (zinc/prover.rs:317-319; poly_f/mle/dense.rs:35-41):
    // nullopt-like failure (wrong point length) throws ZipError{InvalidPcsParam}.
    static Limbs evaluate(const MultilinearZipParams &pp, const int64_t *evals, size_t n_evals, const Limbs *point,
                          size_t point_len, const FieldConfig &field);
};

// ZipProof (src/zinc/structs.rs:26-30) and the PCS step of the prover,
// ZincProver::commit_z_mle_and_prove_evaluation (src/zinc/prover.rs:305-327): code from the main
// transcript, setup, commit, v = z_mle(r_y), open on a FRESH PcsTranscript.
struct ZipProof {
    MultilinearZipCommitment z_comm;
    Limbs v{};
    ByteStream pcs_proof;
};
ZipProof commit_z_mle_and_prove_evaluation(const LinearCodeSpec &lc_spec, const int64_t *z_evals, size_t m,
                                           const Limbs *r_y, size_t r_y_len, KeccakTranscript &transcript,
                                           const FieldConfig &config, int device = 0);

}  // namespace zip

// ---------------------------------------------------------------------------- sumcheck
// MLSumcheck::prove_as_subprotocol (src/sumcheck.rs:56-112) for comb_fn = product of the MLE values
// (ZincProver::sumcheck_2, src/zinc/prover.rs:297-302): the rounds run on the device
// (zip_sumcheck_round), the transcript stays here.
namespace sumcheck {
struct SumcheckProof {                       // SumcheckProof(Vec<ProverMsg>) (src/sumcheck.rs:24-26)
    std::vector<std::vector<Limbs>> msgs;    // per round: evaluations at 0..degree
};
struct ProverOutput {
    SumcheckProof proof;
    std::vector<Limbs> randomness;           // ProverState.randomness
};
// mles[k]: 2^nvars field elements as flat little-endian Montgomery limbs (config.limbs each), host memory.
ProverOutput prove_as_subprotocol_product(KeccakTranscript &transcript, const std::vector<const uint64_t *> &mles,
                                          uint32_t nvars, uint32_t degree, const FieldConfig &config, int device = 0);
// The same with sumcheck_polynomial_comb_fn_1 (src/zinc/utils.rs:77-94; ZincProver::sumcheck_1,
// zinc/prover.rs:241-259): (sum_t c[t] * prod_{j in S[t]} vals[j]) * vals.last().  c: ccs.c (Montgomery),
// S: ccs.S as positions in `mles`, whose last entry is the eq() MLE (prepare_lin_sumcheck_polynomial).
ProverOutput prove_as_subprotocol_ccs(KeccakTranscript &transcript, const std::vector<const uint64_t *> &mles,
                                      uint32_t nvars, uint32_t degree, const std::vector<Limbs> &c,
                                      const std::vector<std::vector<uint32_t>> &S, const FieldConfig &config,
                                      int device = 0);
// The same for a sum of products, comb(vals) = sum_p coeff_p * prod_{j in indices_p} vals[j] (rand_poly_comb_fn,
// src/sumcheck/utils.rs:67-78; the workload of benches/sumcheck_benches.rs: 7 products of 2-4 fresh MLEs).  Each
// product runs as its own device prover (at most 4 multiplicands each), all of them enqueued before any is
// collected; the round message is the field sum of theirs.
ProverOutput prove_as_subprotocol_products(KeccakTranscript &transcript, const std::vector<const uint64_t *> &mles,
                                           uint32_t nvars, uint32_t degree,
                                           const std::vector<std::pair<Limbs, std::vector<uint32_t>>> &products,
                                           const FieldConfig &config, int device = 0);
// SumCheckError / SpartanError (src/sumcheck.rs:28-38, src/zinc/errors.rs): what the verifier returns as Err
struct SpartanError : std::runtime_error {
    enum Kind { SumCheckFailed, InvalidProofLength, MaxDegreeExceeded, PcsVerification } kind;
    SpartanError(Kind k, const std::string &what) : std::runtime_error(what), kind(k) {}
};
struct SubClaim {  // src/sumcheck/verifier.rs:33-40
    std::vector<Limbs> point;
    Limbs expected_evaluation{};
};
// interpolate_uni_poly (src/sumcheck/verifier.rs:161-303)
Limbs interpolate_uni_poly(const FieldConfig &config, const std::vector<Limbs> &p_i, const Limbs &x);
// MLSumcheck::verify_as_subprotocol (src/sumcheck.rs:116-160); throws SpartanError
SubClaim verify_as_subprotocol(KeccakTranscript &transcript, uint32_t num_vars, uint32_t degree, const Limbs &claimed_sum,
                               const SumcheckProof &proof, const FieldConfig &config);
}  // namespace sumcheck
using sumcheck::SpartanError;

// ---------------------------------------------------------------------------- CCS (src/ccs/ccs_z.rs)
namespace ccs {
// SparseMatrix<Int<1>> (src/sparse_matrix.rs:12-17) as CSR; `coeffs` may hold fewer than n_rows rows
// (pad_rows only bumps n_rows, :104-108): from_coeffs() leaves the missing rows empty.
struct SparseMatrix {
    uint32_t n_rows = 0, n_cols = 0;
    std::vector<uint32_t> row_ptr{0}, col_idx;
    std::vector<int64_t> values;
    static SparseMatrix from_coeffs(uint32_t n_rows, uint32_t n_cols,
                                    const std::vector<std::vector<std::pair<int64_t, uint32_t>>> &coeffs);
    zip_sparse_matrix to_abi() const { return {n_rows, n_cols, row_ptr.data(), col_idx.data(), values.data()}; }
};
struct CCS_Z {  // src/ccs/ccs_z.rs:30-52
    size_t m = 0, n = 0, l = 0, t = 0, q = 0, d = 0, s = 0, s_prime = 0;
    std::vector<std::vector<size_t>> S;
    std::vector<int64_t> c;
};
struct Statement_Z {  // :155-158
    std::vector<SparseMatrix> constraints;
    std::vector<int64_t> public_input;
    std::vector<int64_t> get_z_vector(const std::vector<int64_t> &w) const;  // x || 1 || w, :219-229
};
struct Witness_Z {  // :178-182
    std::vector<int64_t> w_ccs;
};
}  // namespace ccs

// ---------------------------------------------------------------------------- ZincProver (src/zinc/prover.rs)
// SpartanProof / ZincProof (src/zinc/structs.rs:13-30)
struct SpartanProof {
    sumcheck::SumcheckProof linearization_sumcheck, second_sumcheck;
    std::vector<Limbs> V_s;
};
struct ZincProof {
    SpartanProof spartan_proof;
    zip::ZipProof zip_proof;
};
// The transcript, the challenges and the proof objects live here; every O(n) loop runs on the device
// (zip_ccs_* for the matrix / eq tables, zip_sumcheck_* for the rounds, the Zip PCS for the last step).
// Shape: m == n == 2^s == 2^s_prime, every c[i] != 0, the S[i] concatenated == 0..t-1 -- what the
// reference's own code supports (compute_eval_table_sparse asserts rx.len() == ccs.n, ccs_f.rs:133;
// sumcheck_polynomial_comb_fn_1 indexes the MLE list by matrix number, zinc/utils.rs:84-88);
// anything else throws std::logic_error where the reference panics.
// The constraint matrices of one circuit in HBM (F_q values in CSR and CSC order) with the proof-time tables
// beside them.  What prepare_for_random_field_piop recomputes per proof in the reference
// (ccs.map_to_field / statement.map_to_field, prover.rs:188-189) depends only on the circuit and the field:
// build it once, hand it to every proof.  One proof at a time per object (internally locked).
class PreparedCcs {
  public:
    PreparedCcs(const ccs::Statement_Z &statement, const ccs::CCS_Z &ccs, const FieldConfig &config, int device = 0);
    // the same from borrowed CSR arrays (t matrices of 2^s columns): nothing is copied on the host
    PreparedCcs(const zip_sparse_matrix *matrices, uint32_t t, uint32_t s, const FieldConfig &config, int device = 0);
    ~PreparedCcs();
    PreparedCcs(const PreparedCcs &) = delete;
    PreparedCcs &operator=(const PreparedCcs &) = delete;

  private:
    friend class ZincProver;
    friend class ZincVerifier;
    zip_ccs *h_ = nullptr;
    std::mutex mu_;
    uint32_t t_ = 0, s_ = 0, limbs_ = 0;
    Limbs modulus_{};
    int device_ = 0;
};

class ZincProver {
  public:
    explicit ZincProver(zip::LinearCodeSpec spec = {}, int device = 0) : lc_spec_(spec), device_(device) {}
    // Prover::prove (prover.rs:50-88)
    // r_y_out: the second sumcheck's point (the verifier re-derives it; handed out for tests and tools)
    // prepared: the circuit's matrices already on the device (must come from the same statement and field)
    ZincProof prove(const ccs::Statement_Z &statement, const ccs::Witness_Z &wit, KeccakTranscript &transcript,
                    const ccs::CCS_Z &ccs, const FieldConfig &config, std::vector<Limbs> *r_y_out = nullptr,
                    PreparedCcs *prepared = nullptr) const;
    // the same from the z vector itself (x || 1 || w, at most ccs.m entries; shorter vectors are zero-extended)
    ZincProof prove_z(const ccs::Statement_Z &statement, const int64_t *z_ccs, size_t z_len, KeccakTranscript &transcript,
                      const ccs::CCS_Z &ccs, const FieldConfig &config, std::vector<Limbs> *r_y_out = nullptr,
                      PreparedCcs *prepared = nullptr) const;
    // prepare_for_random_field_piop (:172-191, the z vector) + SpartanProver::prove (:130-161)
    std::pair<SpartanProof, std::vector<Limbs>> spartan_prove(const ccs::Statement_Z &statement, const int64_t *z_ccs, size_t z_len,
                                                              KeccakTranscript &transcript, const ccs::CCS_Z &ccs,
                                                              const FieldConfig &config, PreparedCcs *prepared = nullptr) const;
    // get_z_ccs_and_z_mle (:222-239): x || 1 || w, zero-extended to ccs.m
    static IntVec get_z_ccs(const int64_t *x, size_t l, const int64_t *w, size_t w_len, size_t m);
    static IntVec get_z_ccs(const ccs::Statement_Z &statement, const ccs::Witness_Z &wit, const ccs::CCS_Z &ccs) {
        return get_z_ccs(statement.public_input.data(), statement.public_input.size(), wit.w_ccs.data(), wit.w_ccs.size(), ccs.m);
    }

  private:
    zip::LinearCodeSpec lc_spec_;
    int device_;
};

// VerificationPoints (src/zinc/verifier.rs:275-279)
struct VerificationPoints {
    std::vector<Limbs> rx_ry;
    Limbs e_y{}, gamma{};
};
// ZincVerifier (src/zinc/verifier.rs).  The sumcheck verifiers are O(s * d) field operations on the host; the
// Zip verifier and the evaluation of the matrix MLEs at (r_x, r_y) run on the device.  Not mirrored: the
// draw_random_field check of Verifier::verify (:53-57, prime generation is out of scope).
class ZincVerifier {
  public:
    explicit ZincVerifier(zip::LinearCodeSpec spec = {}, int device = 0) : lc_spec_(spec), device_(device) {}
    // Verifier::verify (:45-76) without the field check; throws SpartanError / ZipError{InvalidPcsOpen}
    void verify(const ccs::Statement_Z &statement, const ZincProof &proof, KeccakTranscript &transcript, const ccs::CCS_Z &ccs,
                const FieldConfig &config, PreparedCcs *prepared = nullptr) const;
    // SpartanVerifier::verify (:105-139)
    VerificationPoints spartan_verify(const SpartanProof &proof, const ccs::CCS_Z &ccs, KeccakTranscript &transcript,
                                      const FieldConfig &config) const;
    // verify_pcs_proof (:221-273)
    void verify_pcs_proof(const ccs::Statement_Z &statement, const zip::ZipProof &zip_proof, const VerificationPoints &points,
                          const ccs::CCS_Z &ccs, KeccakTranscript &transcript, const FieldConfig &config,
                          PreparedCcs *prepared = nullptr) const {
        verify_pcs_proof(statement, zip_proof.z_comm, zip_proof.v, zip_proof.pcs_proof.data(), zip_proof.pcs_proof.size(), points,
                         ccs, transcript, config, prepared);
    }
    // the same over borrowed proof bytes
    void verify_pcs_proof(const ccs::Statement_Z &statement, const zip::MultilinearZipCommitment &z_comm, const Limbs &v,
                          const uint8_t *pcs_proof, size_t pcs_proof_len, const VerificationPoints &points,
                          const ccs::CCS_Z &ccs, KeccakTranscript &transcript, const FieldConfig &config,
                          PreparedCcs *prepared = nullptr) const;

  private:
    zip::LinearCodeSpec lc_spec_;
    int device_;
};
}  // namespace zinc
