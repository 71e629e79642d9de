//! dump_fixtures — close the parity gap in one sitting.
//!
//! The MI355X engine and its CPU oracle restate cfdkim, mailparse, rsa and regex-automata from their published
//! algorithms; the build image had no Rust toolchain, so nothing there could be checked against the crates themselves
//! ("parity unpinned": DESIGN.md §4).  This example is the other half: run it ONCE on a machine that has the
//! zkemail.rs workspace and cargo, and it writes what the reference itself computes for every case of
//! `tests/golden/ref_manifest.json` into `tests/golden/ref/` —
//!
//!   * `cfdkim::verify_email_with_key(..)`: Ok -> `DKIMResult::with_detail()`, Err -> the error's Display text
//!     (what `verify_dkim` unwraps and tests for "pass": core/src/email.rs:25-36);
//!   * `cfdkim::canonicalize_signed_email(raw)`: canonical header preimage, canonical body, signature bytes
//!     (core/src/circuits.rs:34-35), or the error;
//!   * `hash_bytes(from_domain)`, `hash_bytes(public_key.key)` (core/src/circuits.rs:16-17);
//!   * `remove_quoted_printable_soft_breaks(canonical body)` (core/src/email.rs:61-86);
//!   * per regex pattern: `dfa::regex::Regex::new(pattern)`, both `to_bytes_little_endian()` blobs with the padding stripped
//!     as `helpers/src/regex.rs:7-14` does — these are the first UNANCHORED, accelerated dense DFAs the engine's parser sees —
//!     and the `find_iter` spans over the canonical header and over the cleaned body (core/src/regex.rs:36);
//!   * `VerificationOutput::abi_encode()` of the outputs (core/src/io.rs:28-44).
//!
//! Then `python -m pytest tests/test_reference_fixtures.py` (CPU: the oracle; `-m gpu`: the engine) compares field by field and
//! names the first behaviour that differs — each of them sits behind a strictness flag of `zke_options` or a single named site.
//!
//!     cd zkemail.rs            # the workspace, with bindings/zkemail-amd-sys and bindings/zkemail-core-amd copied beside core/
//!     cargo run -p zkemail-core-amd --example dump_fixtures -- /path/to/zkemail.rs_amd/tests/golden
//!
//! Source only: it has never been compiled (no cargo in the build image).  It uses nothing beyond the reference's own
//! dependencies (Cargo.toml:5-30) and the functions the reference itself calls.
use std::fs;
use std::path::{Path, PathBuf};

use cfdkim::{canonicalize_signed_email, verify_email_with_key, DkimPublicKey};
use mailparse::parse_mail;
use regex_automata::dfa::regex::Regex as DFARegex;
use serde_json::{json, Value};
use zkemail_core::{hash_bytes, remove_quoted_printable_soft_breaks, EmailVerifierOutput, VerificationOutput};

fn hex(b: &[u8]) -> String {
    b.iter().map(|x| format!("{x:02x}")).collect()
}

fn unhex(s: &str) -> Vec<u8> {
    (0..s.len() / 2).map(|i| u8::from_str_radix(&s[2 * i..2 * i + 2], 16).expect("hex")).collect()
}

/// helpers/src/regex.rs:7-14
fn blobs(re: &DFARegex) -> (Vec<u8>, Vec<u8>) {
    let (fwd, fwd_pad) = re.forward().to_bytes_little_endian();
    let (bwd, bwd_pad) = re.reverse().to_bytes_little_endian();
    (fwd[fwd_pad..].to_vec(), bwd[bwd_pad..].to_vec())
}

fn spans(re: &DFARegex, hay: &[u8]) -> Value {
    // find_iter panics when the DFA quits (a Unicode word boundary on non-ASCII input): recorded as such
    let r = std::panic::catch_unwind(|| re.find_iter(hay).map(|m| json!([m.start(), m.end()])).collect::<Vec<_>>());
    match r {
        Ok(v) => json!(v),
        Err(_) => json!("panic"),
    }
}

fn main() {
    let golden = PathBuf::from(std::env::args().nth(1).expect("usage: dump_fixtures <tests/golden>"));
    let manifest: Value = serde_json::from_slice(&fs::read(golden.join("ref_manifest.json")).expect("ref_manifest.json")).expect("json");
    let out_dir = golden.join("ref");
    fs::create_dir_all(&out_dir).expect("mkdir ref");
    let logger = slog::Logger::root(slog::Discard, slog::o!());
    std::panic::set_hook(Box::new(|_| {}));           // the panics below are data, not noise

    for case in manifest["cases"].as_array().expect("cases") {
        let name = case["name"].as_str().unwrap();
        let raw = fs::read(golden.join(case["eml"].as_str().unwrap())).expect("eml");
        let from_domain = case["from_domain"].as_str().unwrap();
        let key_type = case["key_type"].as_str().unwrap();
        let key = unhex(case["key_hex"].as_str().unwrap());
        let mut out = json!({ "name": name });

        // ---- verify_dkim, step by step so that every panic site is told apart (core/src/email.rs:25-36)
        let parsed = std::panic::catch_unwind(|| parse_mail(&raw).map(|_| ()).map_err(|e| e.to_string()));
        out["parse_mail"] = match &parsed {
            Ok(Ok(())) => json!("ok"),
            Ok(Err(e)) => json!({ "error": e }),
            Err(_) => json!("panic"),
        };
        let key_ok = DkimPublicKey::try_from_bytes(&key, key_type).map(|_| ()).map_err(|e| e.to_string());
        out["public_key"] = match &key_ok {
            Ok(()) => json!("ok"),
            Err(e) => json!({ "error": e }),
        };
        if matches!(parsed, Ok(Ok(()))) && key_ok.is_ok() {
            let parsed = parse_mail(&raw).unwrap();
            let pk = DkimPublicKey::try_from_bytes(&key, key_type).unwrap();
            out["verify"] = match verify_email_with_key(&logger, from_domain, &parsed, pk, false) {
                Ok(res) => json!({ "with_detail": res.with_detail(), "pass": res.with_detail().starts_with("pass") }),
                Err(e) => json!({ "error": e.to_string() }),
            };
        }

        // ---- canonicalize_signed_email (core/src/circuits.rs:34-35)
        let canon = std::panic::catch_unwind(|| canonicalize_signed_email(&raw).map_err(|e| e.to_string()));
        let mut header: Vec<u8> = Vec::new();
        let mut cleaned: Vec<u8> = Vec::new();
        out["canonicalize"] = match canon {
            Ok(Ok((h, b, s))) => {
                let (c, _) = remove_quoted_printable_soft_breaks(b.clone());
                header = h.clone();
                cleaned = c.clone();
                json!({ "header_hex": hex(&h), "body_hex": hex(&b), "signature_hex": hex(&s), "cleaned_body_hex": hex(&c) })
            }
            Ok(Err(e)) => json!({ "error": e }),
            Err(_) => json!("panic"),
        };

        // ---- the output witnesses and their ABI encoding (core/src/circuits.rs:15-28, core/src/io.rs:28-44)
        let email_out = EmailVerifierOutput {
            from_domain_hash: hash_bytes(from_domain.as_bytes()),
            public_key_hash: hash_bytes(&key),
            external_inputs: vec![],
        };
        out["from_domain_hash_hex"] = json!(hex(&email_out.from_domain_hash));
        out["public_key_hash_hex"] = json!(hex(&email_out.public_key_hash));
        out["abi_encode_email_only_hex"] = json!(hex(&VerificationOutput::from_parts(
            EmailVerifierOutput {
                from_domain_hash: email_out.from_domain_hash.clone(),
                public_key_hash: email_out.public_key_hash.clone(),
                external_inputs: vec!["name".into(), "value".into()],
            },
            None
        )
        .abi_encode()));
        out["abi_encode_with_regex_hex"] =
            json!(hex(&VerificationOutput::from_parts(email_out, Some(vec!["match one".into(), "".into()])).abi_encode()));

        // ---- regex parts: the blobs helpers hands to core, and find_iter over both haystacks (core/src/regex.rs:32-36)
        let mut parts = Vec::new();
        for pat in case["patterns"].as_array().map(|v| v.as_slice()).unwrap_or(&[]) {
            let pattern = pat.as_str().unwrap();
            parts.push(match DFARegex::new(pattern) {
                Ok(re) => {
                    let (fwd, bwd) = blobs(&re);
                    json!({ "pattern": pattern, "fwd_hex": hex(&fwd), "bwd_hex": hex(&bwd),
                            "header_spans": spans(&re, &header), "cleaned_body_spans": spans(&re, &cleaned) })
                }
                Err(e) => json!({ "pattern": pattern, "error": e.to_string() }),
            });
        }
        out["regex"] = json!(parts);

        let path: &Path = &out_dir.join(format!("{name}.json"));
        fs::write(path, serde_json::to_vec_pretty(&out).unwrap()).expect("write");
        println!("{name}");
    }
}
