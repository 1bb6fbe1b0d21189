// The reference's src/zinc/tests.rs in the host language of the mirror: the same four tests through the C++ classes
// (zinc::ZincProver / zinc::ZincVerifier over ccs::{CCS_Z, Statement_Z, Witness_Z}), not through the C facade.
// Built and run by tests/test_gpu_cpp_mirror.py on the GPU box.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "zinc_zip.hpp"

using namespace zinc;

static FieldConfig field192() {  // field_config!(312829638388039969874974628075306023441, 3), tests.rs:28
    uint64_t m[3] = {0, 0, 0};
    // parse the decimal literal
    const char *dec = "312829638388039969874974628075306023441";
    for (const char *p = dec; *p; p++) {
        unsigned __int128 carry = (unsigned)(*p - '0');
        for (int i = 0; i < 3; i++) {
            const unsigned __int128 t = (unsigned __int128)m[i] * 10 + carry;
            m[i] = (uint64_t)t;
            carry = t >> 64;
        }
    }
    return FieldConfig::make(m, 3);
}

// get_dummy_ccs_Z_from_z (src/ccs/test_utils.rs:89-121): A = B = I, C = diag(z)
static void dummy_ccs(const std::vector<int64_t> &z, ccs::CCS_Z &ccs, ccs::Statement_Z &st, ccs::Witness_Z &wit) {
    const size_t n = z.size();
    size_t s = 0;
    while (((size_t)1 << s) < n) s++;
    ccs.m = ccs.n = n;
    ccs.l = 1;
    ccs.t = 3;
    ccs.q = 2;
    ccs.d = 2;
    ccs.s = ccs.s_prime = s;
    ccs.S = {{0, 1}, {2}};
    ccs.c = {1, -1};
    std::vector<std::vector<std::pair<int64_t, uint32_t>>> ident(n), diag(n);
    for (size_t i = 0; i < n; i++) {
        ident[i] = {{1, (uint32_t)i}};
        diag[i] = {{z[i], (uint32_t)i}};
    }
    st.constraints = {ccs::SparseMatrix::from_coeffs((uint32_t)n, (uint32_t)n, ident),
                      ccs::SparseMatrix::from_coeffs((uint32_t)n, (uint32_t)n, ident),
                      ccs::SparseMatrix::from_coeffs((uint32_t)n, (uint32_t)n, diag)};
    st.public_input = {z[0]};
    wit.w_ccs.assign(z.begin() + 2, z.end());
}

// get_test_ccs_stuff_Z (src/ccs/ccs_z.rs:231-318): x^3 + x + 5 = y, padded to 8 x 8
static void vitalik_ccs(int64_t x, bool break_witness, ccs::CCS_Z &ccs, ccs::Statement_Z &st, ccs::Witness_Z &wit) {
    using Row = std::vector<std::pair<int64_t, uint32_t>>;
    const std::vector<Row> A = {{{1, 0}}, {{1, 3}}, {{1, 0}, {1, 4}}, {{5, 1}, {1, 5}}};
    const std::vector<Row> B = {{{1, 0}}, {{1, 0}}, {{1, 1}}, {{1, 1}}};
    const std::vector<Row> C = {{{1, 3}}, {{1, 4}}, {{1, 5}}, {{1, 2}}};
    ccs.m = ccs.n = 8;
    ccs.l = 1;
    ccs.t = 3;
    ccs.q = 2;
    ccs.d = 2;
    ccs.s = ccs.s_prime = 3;
    ccs.S = {{0, 1}, {2}};
    ccs.c = {1, -1};
    st.constraints = {ccs::SparseMatrix::from_coeffs(8, 8, A), ccs::SparseMatrix::from_coeffs(8, 8, B),
                      ccs::SparseMatrix::from_coeffs(8, 8, C)};
    st.public_input = {x};
    wit.w_ccs = {x * x * x + x + 5, x * x, x * x * x, x * x * x + x};
    if (break_witness) wit.w_ccs[3] = 0;  // tests.rs:166-169
}

static bool spartan_round_trip(const ccs::CCS_Z &ccs, const ccs::Statement_Z &st, const ccs::Witness_Z &wit,
                               const FieldConfig &f) {
    const ZincProver prover;
    const ZincVerifier verifier;
    KeccakTranscript pt, vt;
    const IntVec z = ZincProver::get_z_ccs(st, wit, ccs);
    const auto proved = prover.spartan_prove(st, z.data(), z.size(), pt, ccs, f);
    try {
        const VerificationPoints pts = verifier.spartan_verify(proved.first, ccs, vt, f);
        for (size_t i = 0; i < ccs.s; i++)
            if (pts.rx_ry[ccs.s + i] != proved.second[i]) return false;  // the verifier re-derives r_y
        return true;
    } catch (const SpartanError &) {
        return false;
    }
}

#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            return 1;                                                    \
        }                                                                \
    } while (0)

int main() {
    const FieldConfig f = field192();
    // test_dummy_spartan_prover / test_dummy_spartan_verifier (tests.rs:22-57, 110-157): n = 2^13
    {
        std::vector<int64_t> z(1 << 13);
        uint64_t x = 0x5A494E43;
        for (auto &v : z) {  // SplitMix64
            x += 0x9E3779B97F4A7C15ull;
            uint64_t t = x;
            t = (t ^ (t >> 30)) * 0xBF58476D1CE4E5B9ull;
            t = (t ^ (t >> 27)) * 0x94D049BB133111EBull;
            v = (int64_t)(t ^ (t >> 31));
        }
        z[1] = 1;
        ccs::CCS_Z ccs;
        ccs::Statement_Z st;
        ccs::Witness_Z wit;
        dummy_ccs(z, ccs, st, wit);
        CHECK(spartan_round_trip(ccs, st, wit, f));
        // the full protocol with the PCS step, prepared circuit, through Prover::prove / Verifier::verify
        PreparedCcs prep(st, ccs, f);
        KeccakTranscript pt, vt;
        const ZincProof proof = ZincProver().prove(st, wit, pt, ccs, f, nullptr, &prep);
        CHECK(proof.zip_proof.pcs_proof.size() ==
              (size_t)128 * 64 + (size_t)1000 * 64 * (32 + 8 + 32 * 8) + (size_t)128 * 8 * f.limbs);  // commit.rs:712-775
        ZincVerifier().verify(st, proof, vt, ccs, f, &prep);                                 // throws on rejection
        ZincProof bad = proof;
        bad.zip_proof.v[0] ^= 1;
        bool rejected = false;
        try {
            KeccakTranscript t;
            ZincVerifier().verify(st, bad, t, ccs, f, &prep);
        } catch (const ZipError &e) {
            rejected = e.kind == ZipError::InvalidPcsOpen;
        }
        CHECK(rejected);
    }
    // test_spartan_verifier / test_failing_spartan_verifier (tests.rs:59-108, 159-209)
    {
        ccs::CCS_Z ccs;
        ccs::Statement_Z st;
        ccs::Witness_Z wit;
        vitalik_ccs(3, false, ccs, st, wit);
        CHECK(spartan_round_trip(ccs, st, wit, f));
        vitalik_ccs(3, true, ccs, st, wit);
        CHECK(!spartan_round_trip(ccs, st, wit, f));
    }
    zip::MultilinearZip::release_cached_contexts();
    std::puts("OK");
    return 0;
}
