// C facade over zinc_zip.hpp (include/zinc_zip_host.h): opaque handles + error-code mapping.
#include <string>

#include "zinc_zip.hpp"
#include "zinc_zip_host.h"

using namespace zinc;
using namespace zinc::zip;

struct zinc_transcript { KeccakTranscript t; };
struct zinc_zip_params { MultilinearZipParams pp; };
struct zinc_zip_data { MultilinearZipData d; };
struct zinc_pcs_transcript { PcsTranscript t; };

namespace {
thread_local std::string g_err;

template <class F>
int32_t guarded(F &&f) {
    try {
        f();
        return ZINC_OK;
    } catch (const zinc::SpartanError &e) {
        g_err = e.what();
        return ZINC_ERR_SPARTAN;
    } catch (const ZipError &e) {
        g_err = e.what();
        return e.kind == ZipError::InvalidPcsParam ? ZINC_ERR_INVALID_PARAM
               : (e.kind == ZipError::InvalidPcsOpen || e.kind == ZipError::Transcript) ? ZINC_ERR_INVALID_OPEN
                                                                                        : ZINC_ERR_DEVICE;
    } catch (const std::logic_error &e) {  // where the reference panics
        g_err = e.what();
        return ZINC_ERR_PANIC;
    } catch (const std::exception &e) {
        g_err = e.what();
        return ZINC_ERR_DEVICE;
    }
}
Limbs load(const uint64_t *p, uint32_t n) {
    Limbs l{};
    for (uint32_t i = 0; i < n; i++) l[i] = p[i];
    return l;
}
}  // namespace

extern "C" {

const char *zinc_last_error(void) { return g_err.c_str(); }

zinc_transcript *zinc_transcript_new(void) { return new zinc_transcript(); }
void zinc_transcript_free(zinc_transcript *t) { delete t; }
void zinc_transcript_absorb(zinc_transcript *t, const uint8_t *bytes, size_t len) { t->t.absorb(bytes, len); }
uint64_t zinc_transcript_get_u64(zinc_transcript *t) { return t->t.get_u64(); }
void zinc_transcript_get_integer_challenges(zinc_transcript *t, size_t n, int64_t *out) {
    const auto v = t->t.get_integer_challenges_i64(n);
    for (size_t i = 0; i < n; i++) out[i] = v[i];
}
int32_t zinc_transcript_get_challenge(zinc_transcript *t, const uint64_t *modulus, uint32_t limbs, uint64_t *out) {
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        const Limbs c = t->t.get_challenge(f);
        for (uint32_t i = 0; i < limbs; i++) out[i] = c[i];
    });
}

int32_t zinc_field_constants(const uint64_t *modulus, uint32_t limbs, uint64_t *r, uint64_t *r2, uint64_t *inv) {
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        for (uint32_t i = 0; i < limbs; i++) {
            r[i] = f.r[i];
            r2[i] = f.r2[i];
        }
        *inv = f.inv;
    });
}
int32_t zinc_field_mul(const uint64_t *modulus, uint32_t limbs, const uint64_t *a, const uint64_t *b, uint64_t *out) {
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        Limbs x = load(a, limbs);
        f.mul_assign(x, load(b, limbs));
        for (uint32_t i = 0; i < limbs; i++) out[i] = x[i];
    });
}
int32_t zinc_map_to_field_i64(const uint64_t *modulus, uint32_t limbs, const int64_t *v, size_t n, uint64_t *out) {
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        for (size_t k = 0; k < n; k++) {
            const Limbs x = map_to_field(f, v[k]);
            for (uint32_t i = 0; i < limbs; i++) out[k * limbs + i] = x[i];
        }
    });
}
int32_t zinc_build_eq_x_r(const uint64_t *modulus, uint32_t limbs, const uint64_t *r, uint32_t nvars, uint64_t *out) {
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<Limbs> rv(nvars);
        for (uint32_t t = 0; t < nvars; t++) rv[t] = load(r + (size_t)t * limbs, limbs);
        const auto eq = build_eq_x_r(f, rv.data(), nvars);
        for (size_t k = 0; k < eq.size(); k++)
            for (uint32_t i = 0; i < limbs; i++) out[k * limbs + i] = eq[k][i];
    });
}

void zinc_shuffle_seeded_perm(uint64_t seed, uint32_t len, uint32_t *perm) {
    const auto p = shuffle_seeded_perm(seed, len);
    for (uint32_t i = 0; i < len; i++) perm[i] = p[i];
}

void zinc_kat_seed_from_u64(uint64_t seed, uint32_t *words, uint32_t n_words) { kat_seed_from_u64(seed, words, n_words); }

int32_t zinc_raa_code_new(uint64_t poly_size, zinc_transcript *transcript, zinc_raa_code *out) {
    if (!out) return ZINC_ERR_NULL;
    return guarded([&] {
        RaaCode c;
        if (transcript) {
            KeccakSeedSource src(transcript->t);
            c = RaaCode::make(LinearCodeSpec{}, poly_size, src);
        } else {
            MockTranscript mock;
            c = RaaCode::make(LinearCodeSpec{}, poly_size, mock);
        }
        *out = {c.row_len, c.repetition_factor, c.num_column_opening, c.num_proximity_testing, c.perm_1_seed, c.perm_2_seed};
    });
}

int32_t zinc_zip_setup(uint64_t poly_size, const zinc_raa_code *code, int32_t device, zinc_zip_params **out) {
    if (!code || !out) return ZINC_ERR_NULL;
    *out = nullptr;
    return guarded([&] {
        RaaCode c;
        c.row_len = code->row_len;
        c.repetition_factor = code->repetition_factor;
        c.num_column_opening = code->num_column_opening;
        c.num_proximity_testing = code->num_proximity_testing;
        c.perm_1_seed = code->perm_1_seed;
        c.perm_2_seed = code->perm_2_seed;
        auto *pp = new zinc_zip_params{MultilinearZip::setup(poly_size, c, device)};
        *out = pp;
    });
}
void zinc_zip_params_free(zinc_zip_params *pp) { delete pp; }
void zinc_zip_release_cached_contexts(void) { MultilinearZip::release_cached_contexts(); }
void zinc_zip_params_geometry(const zinc_zip_params *pp, uint32_t *num_vars, uint32_t *num_rows, uint32_t *row_len,
                              uint32_t *codeword_len) {
    *num_vars = pp->pp.num_vars;
    *num_rows = pp->pp.num_rows;
    *row_len = pp->pp.linear_code.row_len;
    *codeword_len = pp->pp.linear_code.codeword_len();
}

int32_t zinc_zip_commit(const zinc_zip_params *pp, const int64_t *evals, size_t n_evals, uint32_t poly_num_vars,
                        int32_t with_merkle, uint8_t *roots_out, zinc_zip_data **out) {
    if (!pp || !out) return ZINC_ERR_NULL;
    *out = nullptr;
    return guarded([&] {
        if (with_merkle) {
            auto res = MultilinearZip::commit(pp->pp, evals, n_evals, poly_num_vars);
            if (roots_out) std::memcpy(roots_out, res.second.roots.data(), res.second.roots.size() * 32);
            *out = new zinc_zip_data{std::move(res.first)};
        } else {
            *out = new zinc_zip_data{MultilinearZip::commit_no_merkle(pp->pp, evals, n_evals, poly_num_vars)};
        }
    });
}
void zinc_zip_data_free(zinc_zip_data *d) { delete d; }

int32_t zinc_zip_data_download(const zinc_zip_data *d, uint64_t *rows_out, uint8_t *layers_out, uint8_t *roots_out) {
    if (!d) return ZINC_ERR_NULL;
    return guarded([&] {
        const int32_t rc = zip_commit_download(d->d.handle.get(), rows_out, layers_out, roots_out);
        if (rc) throw ZipError(ZipError::Device, std::string("zip_commit_download: ") + zip_ctx_last_error(d->d.ctx.get()));
    });
}

int32_t zinc_zip_data_upload(const zinc_zip_params *pp, const uint64_t *rows, const uint8_t *layers,
                             const uint8_t *roots, zinc_zip_data **out) {
    if (!pp || !rows || !out) return ZINC_ERR_NULL;
    *out = nullptr;
    return guarded([&] {
        zip_commitment *h = nullptr;
        const int32_t rc = zip_commitment_upload(pp->pp.ctx.get(), rows, layers, roots, &h);
        if (rc) throw ZipError(ZipError::Device, std::string("zip_commitment_upload: ") + zip_ctx_last_error(pp->pp.ctx.get()));
        MultilinearZipData data;
        data.ctx = pp->pp.ctx;
        data.handle = std::shared_ptr<zip_commitment>(h, zip_commitment_free);
        *out = new zinc_zip_data{std::move(data)};
    });
}

int32_t zinc_merkle_tree_new(uint32_t depth, const uint64_t *leaves, size_t n_leaves, uint32_t leaf_limbs,
                             int32_t device, uint8_t *layers_out) {
    if (!leaves || !layers_out) return ZINC_ERR_NULL;
    return guarded([&] {
        if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) throw std::logic_error("assertion failed: leaves.len().is_power_of_two()");
        if (n_leaves != ((size_t)1 << depth)) throw std::logic_error("assertion failed: leaves.len() == 1 << merkle_depth");
        const int32_t rc = zip_merkle_trees(device, leaves, leaf_limbs, depth, 1, ZIP_MEM_HOST, layers_out);
        if (rc) throw ZipError(ZipError::Device, std::string("zip_merkle_trees: ") + zip_strerror(rc));
    });
}

zinc_pcs_transcript *zinc_pcs_transcript_new(void) { return new zinc_pcs_transcript(); }
void zinc_pcs_transcript_free(zinc_pcs_transcript *t) { delete t; }
size_t zinc_pcs_transcript_len(const zinc_pcs_transcript *t) { return t->t.stream.size(); }
void zinc_pcs_transcript_copy(const zinc_pcs_transcript *t, uint8_t *out) {
    std::memcpy(out, t->t.stream.data(), t->t.stream.size());
}
uint64_t zinc_pcs_transcript_probe(const zinc_pcs_transcript *t) {
    KeccakTranscript copy = t->t.fs_transcript;
    return copy.get_u64();
}

int32_t zinc_zip_open(const zinc_zip_params *pp, const int64_t *evals, size_t n_evals, uint32_t poly_num_vars,
                      const zinc_zip_data *data, const uint64_t *point, size_t point_len, const uint64_t *modulus,
                      uint32_t limbs, zinc_pcs_transcript *transcript) {
    if (!pp || !data || !transcript) return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<Limbs> pt(point_len);
        for (size_t i = 0; i < point_len; i++) pt[i] = load(point + i * limbs, limbs);
        MultilinearZip::open(pp->pp, evals, n_evals, poly_num_vars, data->d, pt.data(), point_len, f, transcript->t);
    });
}

zinc_pcs_transcript *zinc_pcs_transcript_from_proof(const uint8_t *proof, size_t len) {
    auto *t = new zinc_pcs_transcript();
    t->t = PcsTranscript::from_proof(proof, len);
    return t;
}
size_t zinc_pcs_transcript_position(const zinc_pcs_transcript *t) { return t->t.read_pos; }

int32_t zinc_zip_verify(const zinc_zip_params *vp, const uint8_t *roots, const uint64_t *point, size_t point_len,
                        const uint64_t *eval, const uint64_t *modulus, uint32_t limbs, zinc_pcs_transcript *transcript) {
    if (!vp || !roots || !eval || !transcript) return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<Limbs> pt(point_len);
        for (size_t i = 0; i < point_len; i++) pt[i] = load(point + i * limbs, limbs);
        MultilinearZipCommitment comm;
        comm.roots.resize(vp->pp.num_rows);
        std::memcpy(comm.roots.data(), roots, (size_t)vp->pp.num_rows * 32);
        MultilinearZip::verify(vp->pp, comm, pt.data(), point_len, load(eval, limbs), transcript->t, f);
    });
}

int32_t zinc_zip_evaluate(const zinc_zip_params *pp, const int64_t *evals, size_t n_evals, const uint64_t *point,
                          size_t point_len, const uint64_t *modulus, uint32_t limbs, uint64_t *out) {
    if (!pp || !evals || !out) return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<Limbs> pt(point_len);
        for (size_t i = 0; i < point_len; i++) pt[i] = load(point + i * limbs, limbs);
        const Limbs v = MultilinearZip::evaluate(pp->pp, evals, n_evals, pt.data(), point_len, f);
        for (uint32_t i = 0; i < limbs; i++) out[i] = v[i];
    });
}

struct zinc_zip_proof {
    zinc::zip::ZipProof p;
    uint32_t limbs;
};
int32_t zinc_commit_z_mle_and_prove_evaluation(const int64_t *z_evals, size_t m, const uint64_t *r_y, size_t r_y_len,
                                               zinc_transcript *transcript, const uint64_t *modulus, uint32_t limbs,
                                               int32_t device, zinc_zip_proof **out) {
    if (!z_evals || !transcript || !out) return ZINC_ERR_NULL;
    *out = nullptr;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<Limbs> pt(r_y_len);
        for (size_t i = 0; i < r_y_len; i++) pt[i] = load(r_y + i * limbs, limbs);
        auto *res = new zinc_zip_proof{
            zinc::zip::commit_z_mle_and_prove_evaluation(LinearCodeSpec{}, z_evals, m, pt.data(), r_y_len, transcript->t, f, device),
            limbs};
        *out = res;
    });
}
size_t zinc_zip_proof_len(const zinc_zip_proof *p) { return p->p.pcs_proof.size(); }
size_t zinc_zip_proof_num_roots(const zinc_zip_proof *p) { return p->p.z_comm.roots.size(); }
void zinc_zip_proof_read(const zinc_zip_proof *p, uint8_t *roots_out, uint64_t *v_out, uint8_t *pcs_proof_out) {
    if (roots_out) std::memcpy(roots_out, p->p.z_comm.roots.data(), p->p.z_comm.roots.size() * 32);
    if (v_out)
        for (uint32_t i = 0; i < p->limbs; i++) v_out[i] = p->p.v[i];
    if (pcs_proof_out) std::memcpy(pcs_proof_out, p->p.pcs_proof.data(), p->p.pcs_proof.size());
}
void zinc_zip_proof_free(zinc_zip_proof *p) { delete p; }

int32_t zinc_sumcheck_prove_product(zinc_transcript *transcript, const uint64_t *const *mles, uint32_t n_mles,
                                    uint32_t nvars, uint32_t degree, const uint64_t *modulus, uint32_t limbs,
                                    int32_t device, uint64_t *msgs_out, uint64_t *randomness_out) {
    if (!transcript || !mles || !msgs_out || !randomness_out) return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<const uint64_t *> tables(mles, mles + n_mles);
        const auto res = zinc::sumcheck::prove_as_subprotocol_product(transcript->t, tables, nvars, degree, f, device);
        for (size_t r = 0; r < res.proof.msgs.size(); r++) {
            for (uint32_t e = 0; e <= degree; e++)
                for (uint32_t i = 0; i < limbs; i++)
                    msgs_out[(r * (degree + 1) + e) * limbs + i] = res.proof.msgs[r][e][i];
            for (uint32_t i = 0; i < limbs; i++) randomness_out[r * limbs + i] = res.randomness[r][i];
        }
    });
}

int32_t zinc_sumcheck_prove_ccs(zinc_transcript *transcript, const uint64_t *const *mles, uint32_t n_mles,
                                uint32_t nvars, uint32_t degree, uint32_t n_terms, const uint64_t *c,
                                const uint32_t *s_masks, const uint64_t *modulus, uint32_t limbs, int32_t device,
                                uint64_t *msgs_out, uint64_t *randomness_out) {
    if (!transcript || !mles || !c || !s_masks || !msgs_out || !randomness_out) return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<const uint64_t *> tables(mles, mles + n_mles);
        std::vector<Limbs> cv(n_terms);
        std::vector<std::vector<uint32_t>> S(n_terms);
        for (uint32_t t = 0; t < n_terms; t++) {
            cv[t] = load(c + (size_t)t * limbs, limbs);
            for (uint32_t j = 0; j < 32; j++)
                if ((s_masks[t] >> j) & 1u) S[t].push_back(j);
        }
        const auto res = zinc::sumcheck::prove_as_subprotocol_ccs(transcript->t, tables, nvars, degree, cv, S, f, device);
        for (size_t r = 0; r < res.proof.msgs.size(); r++) {
            for (uint32_t e = 0; e <= degree; e++)
                for (uint32_t i = 0; i < limbs; i++)
                    msgs_out[(r * (degree + 1) + e) * limbs + i] = res.proof.msgs[r][e][i];
            for (uint32_t i = 0; i < limbs; i++) randomness_out[r * limbs + i] = res.randomness[r][i];
        }
    });
}

int32_t zinc_sumcheck_prove_products(zinc_transcript *transcript, const uint64_t *const *mles, uint32_t n_mles,
                                     uint32_t nvars, uint32_t degree, uint32_t n_products, const uint64_t *coeffs,
                                     const uint32_t *masks, const uint64_t *modulus, uint32_t limbs, int32_t device,
                                     uint64_t *msgs_out, uint64_t *randomness_out) {
    if (!transcript || !mles || !coeffs || !masks || !msgs_out || !randomness_out) return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        std::vector<const uint64_t *> tables(mles, mles + n_mles);
        std::vector<std::pair<Limbs, std::vector<uint32_t>>> products(n_products);
        for (uint32_t p = 0; p < n_products; p++) {
            products[p].first = load(coeffs + (size_t)p * limbs, limbs);
            for (uint32_t j = 0; j < 32; j++)
                if ((masks[p] >> j) & 1u) products[p].second.push_back(j);
        }
        const auto res = zinc::sumcheck::prove_as_subprotocol_products(transcript->t, tables, nvars, degree, products, f, device);
        for (size_t r = 0; r < res.proof.msgs.size(); r++) {
            for (uint32_t e = 0; e <= degree; e++)
                for (uint32_t i = 0; i < limbs; i++)
                    msgs_out[(r * (degree + 1) + e) * limbs + i] = res.proof.msgs[r][e][i];
            for (uint32_t i = 0; i < limbs; i++) randomness_out[r * limbs + i] = res.randomness[r][i];
        }
    });
}

int32_t zinc_sumcheck_verify(zinc_transcript *transcript, uint32_t nvars, uint32_t degree, const uint64_t *claimed_sum,
                             const uint64_t *msgs, uint32_t n_rounds, uint32_t evals_per_round, const uint64_t *modulus,
                             uint32_t limbs, uint64_t *point_out, uint64_t *expected_out) {
    if (!transcript || !claimed_sum || (n_rounds && !msgs)) return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        zinc::sumcheck::SumcheckProof proof;
        for (uint32_t r = 0; r < n_rounds; r++) {
            proof.msgs.emplace_back();
            for (uint32_t e = 0; e < evals_per_round; e++)
                proof.msgs.back().push_back(load(msgs + ((size_t)r * evals_per_round + e) * limbs, limbs));
        }
        const zinc::sumcheck::SubClaim claim =
            zinc::sumcheck::verify_as_subprotocol(transcript->t, nvars, degree, load(claimed_sum, limbs), proof, f);
        if (point_out)
            for (size_t i = 0; i < claim.point.size(); i++)
                for (uint32_t k = 0; k < limbs; k++) point_out[i * limbs + k] = claim.point[i][k];
        if (expected_out)
            for (uint32_t k = 0; k < limbs; k++) expected_out[k] = claim.expected_evaluation[k];
    });
}

namespace {
zinc::ccs::CCS_Z square_ccs(uint32_t t, uint32_t s) {
    zinc::ccs::CCS_Z ccs;
    ccs.m = ccs.n = (size_t)1 << s;
    ccs.s = ccs.s_prime = s;
    ccs.t = t;
    return ccs;
}
}  // namespace

struct zinc_prepared_ccs {
    std::unique_ptr<zinc::PreparedCcs> p;
    uint32_t t, s;
};

int32_t zinc_prover_prepare(const zip_sparse_matrix *constraints, uint32_t t, uint32_t s, const uint64_t *modulus,
                            uint32_t limbs, int32_t device, zinc_prepared_ccs **out) {
    if (!constraints || !modulus || !out) return ZINC_ERR_NULL;
    *out = nullptr;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        *out = new zinc_prepared_ccs{std::make_unique<zinc::PreparedCcs>(constraints, t, s, f, device), t, s};
    });
}
void zinc_prepared_ccs_free(zinc_prepared_ccs *p) { delete p; }

int32_t zinc_prover_prove(const zip_sparse_matrix *constraints, uint32_t t, uint32_t s, uint32_t d, uint32_t q,
                          const uint32_t *s_masks, const int64_t *c, const int64_t *public_input, size_t l,
                          const int64_t *w_ccs, size_t w_len, zinc_transcript *transcript, const uint64_t *modulus,
                          uint32_t limbs, int32_t device, zinc_prepared_ccs *prepared, int32_t with_pcs,
                          uint64_t *msgs1_out, uint64_t *msgs2_out, uint64_t *v_s_out, uint64_t *r_y_out,
                          zinc_zip_proof **zip_proof_out) {
    if ((!constraints && !prepared) || !s_masks || !c || !transcript || !msgs1_out || !msgs2_out || !v_s_out || !r_y_out ||
        (l && !public_input) || (w_len && !w_ccs) || (with_pcs && !zip_proof_out))
        return ZINC_ERR_NULL;
    if (zip_proof_out) *zip_proof_out = nullptr;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        zinc::ccs::CCS_Z ccs = square_ccs(t, s);
        ccs.l = l;
        ccs.q = q;
        ccs.d = d;
        for (uint32_t i = 0; i < q; i++) {
            ccs.c.push_back(c[i]);
            ccs.S.emplace_back();
            for (uint32_t j = 0; j < 32; j++)
                if ((s_masks[i] >> j) & 1u) ccs.S.back().push_back(j);
        }
        zinc::ccs::Statement_Z st;
        st.constraints.resize(t);  // the matrices go to the device straight from the caller's arrays
        st.public_input.assign(public_input, public_input + l);
        std::unique_ptr<zinc::PreparedCcs> own;  // per proof, as the reference maps the matrices per proof
        if (!prepared) own = std::make_unique<zinc::PreparedCcs>(constraints, t, s, f, device);
        zinc::PreparedCcs *prep = prepared ? prepared->p.get() : own.get();
        const zinc::IntVec z = zinc::ZincProver::get_z_ccs(public_input, l, w_ccs, w_len, ccs.m);  // x || 1 || w
        const zinc::ZincProver prover(LinearCodeSpec{}, device);
        zinc::SpartanProof sp;
        std::vector<Limbs> r_y;
        if (with_pcs) {
            auto proof = prover.prove_z(st, z.data(), z.size(), transcript->t, ccs, f, &r_y, prep);
            sp = std::move(proof.spartan_proof);
            *zip_proof_out = new zinc_zip_proof{std::move(proof.zip_proof), limbs};
        } else {
            auto res = prover.spartan_prove(st, z.data(), z.size(), transcript->t, ccs, f, prep);
            sp = std::move(res.first);
            r_y = std::move(res.second);
        }
        auto put = [&](uint64_t *dst, const Limbs &v) {
            for (uint32_t i = 0; i < limbs; i++) dst[i] = v[i];
        };
        for (size_t r = 0; r < sp.linearization_sumcheck.msgs.size(); r++)
            for (uint32_t e = 0; e <= d + 1; e++) put(msgs1_out + (r * (d + 2) + e) * limbs, sp.linearization_sumcheck.msgs[r][e]);
        for (size_t r = 0; r < sp.second_sumcheck.msgs.size(); r++)
            for (uint32_t e = 0; e < 3; e++) put(msgs2_out + (r * 3 + e) * limbs, sp.second_sumcheck.msgs[r][e]);
        for (size_t k = 0; k < sp.V_s.size(); k++) put(v_s_out + k * limbs, sp.V_s[k]);
        for (size_t i = 0; i < r_y.size(); i++) put(r_y_out + i * limbs, r_y[i]);
    });
}

int32_t zinc_verifier_verify(const zip_sparse_matrix *constraints, uint32_t t, uint32_t s, uint32_t d, uint32_t q,
                             const uint32_t *s_masks, const int64_t *c, zinc_transcript *transcript, const uint64_t *modulus,
                             uint32_t limbs, int32_t device, zinc_prepared_ccs *prepared, const uint64_t *msgs1,
                             const uint64_t *msgs2, const uint64_t *v_s, int32_t with_pcs, const uint8_t *roots,
                             size_t n_roots, const uint64_t *v, const uint8_t *pcs_proof, size_t pcs_proof_len,
                             uint64_t *rx_ry_out, uint64_t *e_y_out, uint64_t *gamma_out) {
    if ((!constraints && !prepared && with_pcs) || !s_masks || !c || !transcript || !msgs1 || !msgs2 || !v_s ||
        (with_pcs && (!roots || !v || !pcs_proof)))
        return ZINC_ERR_NULL;
    return guarded([&] {
        const FieldConfig f = FieldConfig::make(modulus, limbs);
        zinc::ccs::CCS_Z ccs = square_ccs(t, s);
        ccs.q = q;
        ccs.d = d;
        for (uint32_t i = 0; i < q; i++) {
            ccs.c.push_back(c[i]);
            ccs.S.emplace_back();
            for (uint32_t j = 0; j < 32; j++)
                if ((s_masks[i] >> j) & 1u) ccs.S.back().push_back(j);
        }
        zinc::SpartanProof sp;
        auto take = [&](const uint64_t *src, size_t rounds, size_t per) {
            zinc::sumcheck::SumcheckProof p;
            for (size_t r = 0; r < rounds; r++) {
                p.msgs.emplace_back();
                for (size_t e = 0; e < per; e++) p.msgs.back().push_back(load(src + (r * per + e) * limbs, limbs));
            }
            return p;
        };
        sp.linearization_sumcheck = take(msgs1, s, d + 2);
        sp.second_sumcheck = take(msgs2, s, 3);
        for (uint32_t k = 0; k < t; k++) sp.V_s.push_back(load(v_s + (size_t)k * limbs, limbs));
        const zinc::ZincVerifier verifier(LinearCodeSpec{}, device);
        const zinc::VerificationPoints pts = verifier.spartan_verify(sp, ccs, transcript->t, f);
        auto put = [&](uint64_t *dst, const Limbs &x) {
            if (dst)
                for (uint32_t i = 0; i < limbs; i++) dst[i] = x[i];
        };
        if (rx_ry_out)
            for (size_t i = 0; i < pts.rx_ry.size(); i++) put(rx_ry_out + i * limbs, pts.rx_ry[i]);
        put(e_y_out, pts.e_y);
        put(gamma_out, pts.gamma);
        if (!with_pcs) return;
        zinc::ccs::Statement_Z st;
        st.constraints.resize(t);
        std::unique_ptr<zinc::PreparedCcs> own;
        if (!prepared) own = std::make_unique<zinc::PreparedCcs>(constraints, t, s, f, device);
        zinc::zip::MultilinearZipCommitment z_comm;
        z_comm.roots.resize(n_roots);
        std::memcpy(z_comm.roots.data(), roots, n_roots * 32);
        verifier.verify_pcs_proof(st, z_comm, load(v, limbs), pcs_proof, pcs_proof_len, pts, ccs, transcript->t, f,
                                  prepared ? prepared->p.get() : own.get());  // the proof bytes are read in place
    });
}

}  // extern "C"
