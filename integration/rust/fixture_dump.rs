//! Fixture dumper for the MI355X path's CPU oracle -- to be run ONCE by a maintainer on any machine with cargo
//! (no GPU needed).  It records what the REAL reference computes for the pieces `zinc_amd`'s oracle can only
//! restate (this repository has no Rust toolchain): the `rand`-based permutation tables, whole commit / open
//! results under `MockTranscript`, and `map_to_field` for a modulus with its top bit set.
//!
//! How to run (paths relative to the reference checkout):
//!   1. copy this file to `src/zip/pcs/fixture_dump.rs`
//!   2. add `#[cfg(test)] mod fixture_dump;` to `src/zip/pcs.rs` (next to `mod tests;`)
//!   3. `RAYON_NUM_THREADS=8 cargo test --release --features parallel fixture_dump -- --nocapture`
//!      (a power-of-two thread count: `encode_rows` leaves tail rows zero otherwise, commit.rs:164-170)
//!   4. copy `target/zinc_amd_fixtures.json` to `tests/golden/rust_fixtures.json` of the zinc_amd repository;
//!      `pytest tests/test_rust_fixtures.py` then compares the oracle with it (skipped while the file is absent).
//!
//! NOT COMPILED by the builder (no cargo in its image): written against the API as the reference's own tests and
//! benches use it (src/zip/pcs/tests.rs, benches/zip_benches.rs); a maintainer may need to fix an import.
//! It lives inside the crate because `shuffle_seeded` is `pub(super)` (src/zip/utils.rs:139).
#![cfg(test)]

use std::{fmt::Write as _, str::FromStr};

use sha3::{Digest, Sha3_256};

use crate::{
    define_random_field_zip_types,
    field::{BigInt, ConfigRef, FieldConfig, Int, RandomField},
    implement_random_field_zip_types,
    poly_z::mle::DenseMultilinearExtension,
    traits::{Field, FieldMap},
    zip::{
        code::{DefaultLinearCodeSpec, LinearCode},
        code_raa::RaaCode,
        pcs::{structs::MultilinearZip, tests::MockTranscript},
        pcs_transcript::PcsTranscript,
        utils::shuffle_seeded,
    },
};

const INT_LIMBS: usize = 1;
const FIELD_LIMBS: usize = 4;

define_random_field_zip_types!();
implement_random_field_zip_types!(INT_LIMBS);

type ZT = RandomFieldZipTypes<INT_LIMBS>;
type LC = RaaCode<ZT>;
type Zip = MultilinearZip<ZT, LC>;

const BENCH_MODULUS: &str = "106319353542452952636349991594949358997917625194731877894581586278529202198383"; // benches/zip_benches.rs:253
const TOP_BIT_MODULUS: &str = "115792089237316195423570985008687907853269984665640564039457584007913129639747"; // 2^256 - 189, benches/spartan_benches.rs:134-137

fn hex(bytes: &[u8]) -> String {
    let mut s = String::with_capacity(2 * bytes.len());
    for b in bytes {
        write!(s, "{b:02x}").unwrap();
    }
    s
}

fn sha3(bytes: &[u8]) -> String {
    hex(&Sha3_256::digest(bytes))
}

/// SplitMix64 stream `tests/_oracle.py::splitmix64(seed, n)`: element i (from 1) = mix(seed + i * golden gamma).
fn splitmix64(seed: u64, n: usize) -> Vec<i64> {
    (1..=n as u64)
        .map(|i| {
            let mut z = seed.wrapping_add(i.wrapping_mul(0x9E37_79B9_7F4A_7C15));
            z = (z ^ (z >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
            z = (z ^ (z >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
            (z ^ (z >> 31)) as i64
        })
        .collect()
}

#[test]
fn fixture_dump() {
    let mut out = String::from("{\n");

    // ---- 1. shuffle_seeded on the identity: table[j] = x[perm[j]] (src/zip/utils.rs:139-142)
    out.push_str(" \"perm_tables\": [\n");
    let cases: [(u64, usize); 6] = [(1, 512), (2, 512), (1, 13), (12345, 10), (1, 8192), (2, 8192)];
    for (k, (seed, len)) in cases.iter().enumerate() {
        let mut v: Vec<u32> = (0..*len as u32).collect();
        shuffle_seeded(&mut v, *seed);
        let bytes: Vec<u8> = v.iter().flat_map(|x| x.to_le_bytes()).collect();
        let head: Vec<String> = v.iter().take(16).map(|x| x.to_string()).collect();
        writeln!(
            out,
            "  {{\"seed\": {seed}, \"len\": {len}, \"first16\": [{}], \"sha3_256_le_u32\": \"{}\"}}{}",
            head.join(", "),
            sha3(&bytes),
            if k + 1 < cases.len() { "," } else { "" }
        )
        .unwrap();
    }
    out.push_str(" ],\n");

    // ---- 2. commit + open under MockTranscript (perm seeds 1 and 2), fresh PcsTranscript, point = [1; nv]
    let config = FieldConfig::new(BigInt::<FIELD_LIMBS>::from_str(BENCH_MODULUS).unwrap());
    let cfg = ConfigRef::from(&config);
    out.push_str(" \"commit_open\": [\n");
    let sizes = [12usize, 13, 14, 16];
    for (k, nv) in sizes.iter().enumerate() {
        let poly_size = 1usize << nv;
        let mut mock = MockTranscript::default();
        let code = LC::new(&DefaultLinearCodeSpec, poly_size, &mut mock);
        let pp = Zip::setup(poly_size, code);
        let evals: Vec<Int<INT_LIMBS>> = splitmix64(0x5A49_4E43, poly_size).into_iter().map(Int::from).collect();
        let poly = DenseMultilinearExtension::from_evaluations_vec(*nv, evals);
        let (data, comm) = Zip::commit::<RandomField<FIELD_LIMBS>>(&pp, &poly).expect("commit");
        let roots: Vec<u8> = comm.roots.iter().flat_map(|h| h.as_bytes().to_vec()).collect();
        let rows: Vec<u8> = data
            .rows
            .iter()
            .flat_map(|x| x.as_words().iter().flat_map(|w| w.to_le_bytes()).collect::<Vec<u8>>())
            .collect();
        let point: Vec<RandomField<FIELD_LIMBS>> = vec![1i64; *nv].map_to_field(cfg);
        let mut transcript = PcsTranscript::<RandomField<FIELD_LIMBS>>::new();
        Zip::open(&pp, &poly, &data, &point, cfg, &mut transcript).expect("open");
        let proof = transcript.into_proof();
        writeln!(
            out,
            "  {{\"num_vars\": {nv}, \"witness\": \"splitmix64(0x5A494E43)\", \"modulus\": \"{BENCH_MODULUS}\", \
             \"root0\": \"{}\", \"roots_sha3_256\": \"{}\", \"rows_sha3_256\": \"{}\", \"proof_len\": {}, \"proof_sha3_256\": \"{}\"}}{}",
            hex(&roots[..32]),
            sha3(&roots),
            sha3(&rows),
            proof.len(),
            sha3(&proof),
            if k + 1 < sizes.len() { "," } else { "" }
        )
        .unwrap();
    }
    out.push_str(" ],\n");

    // ---- 3. map_to_field of small and extreme witnesses (src/conversion.rs:86-100, src/field.rs:536-568):
    //         big-endian bytes of the Montgomery value, as write_field_element emits them (pcs_transcript.rs:107-113)
    out.push_str(" \"map_to_field\": [\n");
    let moduli = [BENCH_MODULUS, TOP_BIT_MODULUS];
    let inputs: [i64; 8] = [5, -5, 190, -190, 189, -189, i64::MAX, i64::MIN];
    for (k, m) in moduli.iter().enumerate() {
        let config = FieldConfig::new(BigInt::<FIELD_LIMBS>::from_str(m).unwrap());
        let cfg = ConfigRef::from(&config);
        let vals: Vec<String> = inputs
            .iter()
            .map(|w| {
                let fe: RandomField<FIELD_LIMBS> = Int::<INT_LIMBS>::from(*w).map_to_field(cfg);
                format!("\"{}\"", hex(fe.value().clone().to_bytes_be().as_ref()))
            })
            .collect();
        let ins: Vec<String> = inputs.iter().map(|w| w.to_string()).collect();
        writeln!(
            out,
            "  {{\"modulus\": \"{m}\", \"inputs\": [{}], \"montgomery_be\": [{}]}}{}",
            ins.join(", "),
            vals.join(", "),
            if k + 1 < moduli.len() { "," } else { "" }
        )
        .unwrap();
    }
    out.push_str(" ]\n}\n");

    let path = std::env::var("ZINC_AMD_FIXTURES").unwrap_or_else(|_| "target/zinc_amd_fixtures.json".into());
    std::fs::write(&path, &out).expect("write fixtures");
    println!("wrote {path}");
}
