//! `zkemail-core-amd` — zkemail-core's verification API over the MI355X-native batched engine.
//!
//! ```ignore
//! // before:  use zkemail_core::{verify_email, verify_email_with_regex};
//! use zkemail_core_amd::{verify_email, verify_email_with_regex};
//! ```
//!
//! Same input and output structs (they are re-exported from `zkemail-core`, `core/src/structs.rs:8-75`), same
//! function signatures (`core/src/circuits.rs:9,31`) and the same failure behaviour: where the reference panics
//! (`assert!` `circuits.rs:13,45,54`; `expect` `:24`; `unwrap` `circuits.rs:35`, `email.rs:26,29,33`,
//! `regex.rs:32,33`) these functions panic too, with a message that names the site.  What differs is where the
//! work happens: header split, canonicalisation, SHA-256, RSA / Ed25519 and the DFA walk run on the GPU behind
//! `libzkemail_amd.so`.  A GPU pays off on batches, so the crate adds [`Engine::verify_emails`] /
//! [`Engine::verify_emails_with_regex`], which take slices and return one `Result` per e-mail instead of aborting
//! the batch on the first bad one.
//!
//! No Rust toolchain exists in the image this repository is built in: the crate is source that mirrors
//! `include/zkemail_core.hpp` (its compiled and GPU-tested C++ twin) call for call, and
//! `tests/test_rust_bindings.py` keeps the FFI layer under it identical to the C header.
#![deny(unsafe_op_in_unsafe_fn)]

use std::ffi::CStr;
use std::fmt;
use std::ptr;
use std::sync::OnceLock;

use zkemail_amd_sys as sys;

pub use zkemail_core::{
    hash_bytes, remove_quoted_printable_soft_breaks, CompiledRegex, Email, EmailVerifierOutput, EmailWithRegex,
    EmailWithRegexVerifierOutput, ExternalInput, PublicKey, RegexInfo, VerificationOutput, DFA,
};

/// The reference would have panicked on this e-mail: `status` names the panic site
/// (`ZKE_PARSE_FAIL` = `email.rs:26` … `ZKE_BODY_REGEX_FAIL` = `circuits.rs:54`), `detail` is cfdkim's reason.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub struct Panic {
    pub status: u32,
    pub detail: u32,
}

impl Panic {
    /// The reference source line whose `assert!` / `unwrap` / `expect` fires for this status.
    pub fn site(&self) -> &'static str {
        match self.status {
            sys::ZKE_PARSE_FAIL => "core/src/email.rs:26 mailparse::parse_mail(..).unwrap()",
            sys::ZKE_KEY_DECODE_FAIL => "core/src/email.rs:29 DkimPublicKey::try_from_bytes(..).unwrap()",
            sys::ZKE_DKIM_ERROR => "core/src/email.rs:33 verify_email_with_key(..).unwrap()",
            sys::ZKE_DKIM_NOT_PASS => "core/src/circuits.rs:13 assert!(verified)",
            sys::ZKE_EXTERNAL_INPUT_NULL => "core/src/circuits.rs:24 expect(\"Value cannot be null\")",
            sys::ZKE_CANON_FAIL => "core/src/circuits.rs:35 canonicalize_signed_email(..).unwrap()",
            sys::ZKE_DFA_DECODE_FAIL => "core/src/regex.rs:32-33 dense::DFA::from_bytes(..).unwrap()",
            sys::ZKE_HEADER_REGEX_FAIL => "core/src/circuits.rs:45 assert!(verified)",
            sys::ZKE_BODY_REGEX_FAIL => "core/src/circuits.rs:54 assert!(verified)",
            sys::ZKE_UNSUPPORTED => "input outside what the engine implements (never a silent mis-verify)",
            _ => "unknown status",
        }
    }
}

impl fmt::Display for Panic {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "zkemail_core would panic at {} (status {}, detail {})", self.site(), self.status, self.detail)
    }
}

impl std::error::Error for Panic {}

/// The call itself failed (no GPU, bad arguments, out of device memory) — not a property of an e-mail.
#[derive(Debug, Clone)]
pub struct EngineError {
    pub code: i32,
    pub message: String,
}

impl fmt::Display for EngineError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "zkemail_amd engine error {}: {}", self.code, self.message)
    }
}

impl std::error::Error for EngineError {}

/// One `Email` as the C side sees it: pointers into its own buffers (valid while the `Email` is borrowed).
fn email_ref(e: &Email) -> sys::zke_email_ref {
    sys::zke_email_ref {
        raw: e.raw_email.as_ptr(),
        raw_len: e.raw_email.len(),
        from_domain: e.from_domain.as_ptr() as *const std::os::raw::c_char,
        domain_len: e.from_domain.len(),
        key: e.public_key.key.as_ptr(),
        key_len: e.public_key.key.len(),
        key_type: key_type_code(&e.public_key.key_type) as u32,
        external_input_null: e.external_inputs.iter().any(|x| x.value.is_none()) as u32,
    }
}

fn key_type_code(key_type: &str) -> u8 {
    match key_type {
        "rsa" => sys::ZKE_KEY_RSA as u8,
        "ed25519" => sys::ZKE_KEY_ED25519 as u8,
        _ => sys::ZKE_KEY_OTHER as u8, // DkimPublicKey::try_from_bytes rejects it: ZKE_KEY_DECODE_FAIL
    }
}

/// Struct-of-arrays image of `&[Email]` as `zke_batch` wants it (CSR blobs, `include/zkemail_amd.h`).  `verify_emails` does not
/// need it (the engine gathers the e-mails from where they are); it is for a caller that submits the same batch more than once
/// or uploads it to HBM for `zke_verify_batch_device` (`Engine::verify_packed`).
pub struct Packed {
    raw_blob: Vec<u8>,
    raw_off: Vec<u64>,
    domain_blob: Vec<u8>,
    domain_off: Vec<u64>,
    key_blob: Vec<u8>,
    key_off: Vec<u64>,
    key_type: Vec<u8>,
    ext_null: Vec<u8>,
}

impl Packed {
    pub fn from_emails<'a, I: Iterator<Item = &'a Email>>(emails: I) -> Self {
        let mut p = Packed {
            raw_blob: Vec::new(),
            raw_off: vec![0],
            domain_blob: Vec::new(),
            domain_off: vec![0],
            key_blob: Vec::new(),
            key_off: vec![0],
            key_type: Vec::new(),
            ext_null: Vec::new(),
        };
        for e in emails {
            p.raw_blob.extend_from_slice(&e.raw_email);
            p.raw_off.push(p.raw_blob.len() as u64);
            p.domain_blob.extend_from_slice(e.from_domain.as_bytes());
            p.domain_off.push(p.domain_blob.len() as u64);
            p.key_blob.extend_from_slice(&e.public_key.key);
            p.key_off.push(p.key_blob.len() as u64);
            p.key_type.push(key_type_code(&e.public_key.key_type));
            // circuits.rs:24: a None value panics — after the DKIM assert, before any regex work
            p.ext_null.push(e.external_inputs.iter().any(|x| x.value.is_none()) as u8);
        }
        // the C side never dereferences an empty blob, but it wants non-null pointers
        for blob in [&mut p.raw_blob, &mut p.domain_blob, &mut p.key_blob] {
            if blob.is_empty() {
                blob.push(0);
            }
        }
        p
    }

    pub fn batch(&self, n: usize) -> sys::zke_batch {
        sys::zke_batch {
            n: n as u32,
            raw_blob: self.raw_blob.as_ptr(),
            raw_off: self.raw_off.as_ptr(),
            domain_blob: self.domain_blob.as_ptr(),
            domain_off: self.domain_off.as_ptr(),
            key_blob: self.key_blob.as_ptr(),
            key_off: self.key_off.as_ptr(),
            key_type: self.key_type.as_ptr(),
            ext_null: self.ext_null.as_ptr(),
            with_regex: 0,
            n_header_parts: 0,
            n_body_parts: 0,
            header_part_ids: ptr::null(),
            body_part_ids: ptr::null(),
            cap_off: ptr::null(),
            cap_str_off: ptr::null(),
            cap_blob: ptr::null(),
        }
    }
}

fn zeroed_result() -> sys::zke_result {
    sys::zke_result {
        status: 0,
        detail: 0,
        sig_index: 0,
        flags: 0,
        canon_header_len: 0,
        canon_body_len: 0,
        body_offset: 0,
        n_headers: 0,
        from_domain_hash: [0; 32],
        public_key_hash: [0; 32],
        body_hash: [0; 32],
        header_hash: [0; 32],
        regex_part: 0,
        match_count: 0,
        match_start: 0,
        match_end: 0,
        rsa_bits: 0,
        reserved: [0; 3],
    }
}

/// `EmailVerifierOutput` of a verified e-mail (circuits.rs:15-28): the two witnesses come from the record, the
/// external inputs are echoed `[name, value, ...]`.
fn email_output(email: &Email, r: &sys::zke_result) -> EmailVerifierOutput {
    EmailVerifierOutput {
        from_domain_hash: r.from_domain_hash.to_vec(),
        public_key_hash: r.public_key_hash.to_vec(),
        external_inputs: email
            .external_inputs
            .iter()
            .flat_map(|x| vec![x.name.clone(), x.value.clone().expect("Value cannot be null")])
            .collect(),
    }
}

/// `regex_matches` = header captures then body captures (circuits.rs:58-62); the strings are the *input* capture
/// strings (regex.rs:47), meaningful once the engine reported `ZKE_OK`.
fn regex_matches_of(info: &RegexInfo) -> Vec<String> {
    let mut out = Vec::new();
    for parts in [&info.header_parts, &info.body_parts] {
        if let Some(parts) = parts {
            for p in parts {
                if let Some(c) = &p.captures {
                    out.extend(c.iter().cloned());
                }
            }
        }
    }
    out
}

/// The default configuration (a zero-filled `zke_options` with the device set).
pub fn default_options(device: i32) -> sys::zke_options {
    sys::zke_options {
        device,
        slots: 0,
        max_sig_rounds: 0,
        disable_key_cache: 0,
        host_threads: 0,
        max_dfas: 0,
        rsa_lane_groups: 0,
        dfa_mapping: 0,
        replay_graphs: 0,
        enforce_expiry_x: 0,
        canon_takes_verified_signature: 0,
        canon_ignores_l: 0,
        i_must_be_subdomain: 0,
        b_removes_own_span_only: 0,
        reserved0: 0,
        now_unix: 0,
        reserved: [0; 4],
    }
}

/// One engine per GPU: submission slots (a stream, a workspace and a pinned staging image each), registered DFA
/// tables, the per-key Montgomery cache.  Re-entrant, as the reference's functions are (core/src/circuits.rs:9): the
/// C entry points take a slot by an atomic ticket and hold that slot's lock while they enqueue, so one `Engine` may be
/// shared by any number of threads.
pub struct Engine {
    raw: *mut sys::zke_engine,
}

// The handle owns device memory and streams; the C-ABI serialises what must be serialised (include/zkemail_amd.h, "Threading").
unsafe impl Send for Engine {}
unsafe impl Sync for Engine {}

impl Engine {
    /// `device`: HIP device ordinal, -1 = the current device.  Every other option at its default.
    pub fn new(device: i32) -> Result<Self, EngineError> {
        Self::with_options(default_options(device))
    }

    /// An engine with explicit `zke_options` (ABI 0.3: submission slots, host threads, kernel variants and the
    /// strictness flags — the readings of cfdkim that a maintainer with the crates at hand may want to flip).
    pub fn with_options(opt: sys::zke_options) -> Result<Self, EngineError> {
        let mut raw: *mut sys::zke_engine = ptr::null_mut();
        // SAFETY: `opt` and `raw` outlive the call; the callee writes a handle or leaves null.
        let rc = unsafe { sys::zke_engine_create(&opt, &mut raw) };
        if rc != 0 || raw.is_null() {
            return Err(EngineError { code: rc, message: "zke_engine_create failed (no HIP device? the engine has no CPU path)".into() });
        }
        Ok(Engine { raw })
    }

    fn last_error(&self, code: i32) -> EngineError {
        // SAFETY: the engine is alive; zke_last_error returns a NUL-terminated string it owns.
        let message = unsafe { CStr::from_ptr(sys::zke_last_error(self.raw)) }.to_string_lossy().into_owned();
        EngineError { code, message }
    }

    /// Pre-size `slots` submission slots for batches of up to `max_n` e-mails / `max_raw_total` raw bytes
    /// (`zke_engine_reserve`): nothing is allocated in the submit path afterwards.
    pub fn reserve(&self, max_n: u32, max_raw_total: u64, slots: u32, max_regex_parts: u32) -> Result<(), EngineError> {
        // SAFETY: plain values.
        let rc = unsafe { sys::zke_engine_reserve(self.raw, max_n, max_raw_total, slots, max_regex_parts) };
        if rc != 0 {
            return Err(self.last_error(rc));
        }
        Ok(())
    }

    /// The raw records of a batch packed beforehand (`Packed::from_emails`): `zke_verify_batch` over the three CSR blobs.
    pub fn verify_packed(&self, packed: &Packed, n: usize) -> Result<Vec<sys::zke_result>, EngineError> {
        let batch = packed.batch(n);
        let mut out = vec![zeroed_result(); n];
        // SAFETY: every pointer in `batch` refers into `packed`, borrowed for the whole (synchronous) call; `out` holds n records.
        let rc = unsafe { sys::zke_verify_batch(self.raw, &batch, out.as_mut_ptr(), ptr::null_mut()) };
        if rc != 0 {
            return Err(self.last_error(rc));
        }
        Ok(out)
    }

    /// `verify_email` over a slice: one `Result` per e-mail, in order.  Never aborts the batch on a bad e-mail.
    pub fn verify_emails(&self, emails: &[Email]) -> Result<Vec<Result<EmailVerifierOutput, Panic>>, EngineError> {
        if emails.is_empty() {
            return Ok(Vec::new());
        }
        // The e-mails stay where they are: one zke_email_ref per `Email`, pointing into its own Vec<u8> / Strings.  The engine
        // gathers them into its pinned staging image on its packing threads (zke_verify_emails) — no concatenation here.
        let refs: Vec<sys::zke_email_ref> = emails.iter().map(email_ref).collect();
        let mut out = vec![zeroed_result(); emails.len()];
        // SAFETY: every pointer in `refs` refers into `emails`, borrowed for the whole (synchronous) call; `out` holds n records.
        let rc = unsafe { sys::zke_verify_emails(self.raw, refs.as_ptr(), refs.len() as u32, out.as_mut_ptr()) };
        if rc != 0 {
            return Err(self.last_error(rc));
        }
        Ok(emails
            .iter()
            .zip(out.iter())
            .map(|(e, r)| if r.status == sys::ZKE_OK { Ok(email_output(e, r)) } else { Err(Panic { status: r.status, detail: r.detail }) })
            .collect())
    }

    /// `verify_email_with_regex` over a slice.  All inputs must share one part list (one `regex_config` per batch:
    /// the same DFA pairs in the same order); the captures are per e-mail.  Each DFA pair is parsed and staged on
    /// the device once (`zke_dfa_register`), not once per e-mail as `core/src/regex.rs:32-33` does.
    pub fn verify_emails_with_regex(
        &self,
        inputs: &[EmailWithRegex],
    ) -> Result<Vec<Result<EmailWithRegexVerifierOutput, Panic>>, EngineError> {
        if inputs.is_empty() {
            return Ok(Vec::new());
        }
        let empty: Vec<CompiledRegex> = Vec::new();
        let first = &inputs[0].regex_info;
        let ids = |engine: &Engine, parts: &Option<Vec<CompiledRegex>>| -> Result<Vec<u32>, EngineError> {
            let mut v = Vec::new();
            for p in parts.as_ref().unwrap_or(&empty) {
                let mut id = 0u32;
                // SAFETY: the slices outlive the call; the engine copies what it keeps.
                let rc = unsafe {
                    sys::zke_dfa_register(
                        engine.raw,
                        p.verify_re.fwd.as_ptr(),
                        p.verify_re.fwd.len(),
                        p.verify_re.bwd.as_ptr(),
                        p.verify_re.bwd.len(),
                        &mut id,
                    )
                };
                if rc != 0 {
                    return Err(engine.last_error(rc));
                }
                v.push(id); // registering an equal pair again returns the id it already has
            }
            Ok(v)
        };
        let hdr_ids = ids(self, &first.header_parts)?;
        let body_ids = ids(self, &first.body_parts)?;
        let n_parts = hdr_ids.len() + body_ids.len();
        // captures: two-level CSR — strings cap_str_off[cap_off[i*P+p] .. cap_off[i*P+p+1]) of e-mail i, part p
        let mut cap_off: Vec<u32> = vec![0];
        let mut cap_str_off: Vec<u32> = vec![0];
        let mut cap_blob: Vec<u8> = Vec::new();
        for inp in inputs {
            if ids(self, &inp.regex_info.header_parts)? != hdr_ids || ids(self, &inp.regex_info.body_parts)? != body_ids {
                return Err(EngineError { code: sys::ZKE_E_ARG, message: "a batch must share one part list; split it per regex_config".into() });
            }
            for parts in [&inp.regex_info.header_parts, &inp.regex_info.body_parts] {
                for p in parts.as_ref().unwrap_or(&empty) {
                    for c in p.captures.as_ref().map(|v| v.as_slice()).unwrap_or(&[]) {
                        cap_blob.extend_from_slice(c.as_bytes());
                        cap_str_off.push(cap_blob.len() as u32);
                    }
                    cap_off.push((cap_str_off.len() - 1) as u32);
                }
            }
        }
        if cap_blob.is_empty() {
            cap_blob.push(0);
        }
        // the e-mails stay in their own buffers (zke_verify_emails_with_regex gathers them); only the small tables are built here
        let refs: Vec<sys::zke_email_ref> = inputs.iter().map(|i| email_ref(&i.email)).collect();
        let lists = sys::zke_regex_lists {
            n_header_parts: hdr_ids.len() as u32,
            header_part_ids: hdr_ids.as_ptr(),
            n_body_parts: body_ids.len() as u32,
            body_part_ids: body_ids.as_ptr(),
            cap_off: if n_parts > 0 { cap_off.as_ptr() } else { ptr::null() },
            cap_str_off: cap_str_off.as_ptr(),
            cap_blob: cap_blob.as_ptr(),
        };
        let mut out = vec![zeroed_result(); inputs.len()];
        // SAFETY: as in verify_emails; the id lists and the capture tables live until the call returns.
        let rc = unsafe { sys::zke_verify_emails_with_regex(self.raw, refs.as_ptr(), refs.len() as u32, &lists, out.as_mut_ptr()) };
        if rc != 0 {
            return Err(self.last_error(rc));
        }
        Ok(inputs
            .iter()
            .zip(out.iter())
            .map(|(inp, r)| {
                if r.status == sys::ZKE_OK {
                    Ok(EmailWithRegexVerifierOutput { email: email_output(&inp.email, r), regex_matches: regex_matches_of(&inp.regex_info) })
                } else {
                    Err(Panic { status: r.status, detail: r.detail })
                }
            })
            .collect())
    }

    /// One e-mail through the single-e-mail C entry point `zke_verify_email` (core/src/circuits.rs:9).
    pub fn try_verify_email(&self, email: &Email) -> Result<Result<EmailVerifierOutput, Panic>, EngineError> {
        let mut r = zeroed_result();
        let ext_null = email.external_inputs.iter().any(|x| x.value.is_none()) as u32;
        // SAFETY: the slices outlive the (synchronous) call; lengths are passed with them.
        let rc = unsafe {
            sys::zke_verify_email(
                self.raw,
                email.raw_email.as_ptr(),
                email.raw_email.len(),
                email.from_domain.as_ptr().cast(),
                email.from_domain.len(),
                email.public_key.key.as_ptr(),
                email.public_key.key.len(),
                key_type_code(&email.public_key.key_type) as u32,
                ext_null,
                &mut r,
            )
        };
        if rc != 0 {
            return Err(self.last_error(rc));
        }
        Ok(if r.status == sys::ZKE_OK { Ok(email_output(email, &r)) } else { Err(Panic { status: r.status, detail: r.detail }) })
    }

    /// One `EmailWithRegex` through `zke_verify_email_with_regex` (core/src/circuits.rs:31).
    pub fn try_verify_email_with_regex(
        &self,
        input: &EmailWithRegex,
    ) -> Result<Result<EmailWithRegexVerifierOutput, Panic>, EngineError> {
        // RegexInfo -> two zke_regex_part lists; the pointer tables live until the call returns
        struct Part {
            ptrs: Vec<*const u8>,
            lens: Vec<usize>,
        }
        let empty: Vec<CompiledRegex> = Vec::new();
        let mut keep: Vec<Part> = Vec::new();
        let mut lists: [Vec<sys::zke_regex_part>; 2] = [Vec::new(), Vec::new()];
        for (side, parts) in [&input.regex_info.header_parts, &input.regex_info.body_parts].into_iter().enumerate() {
            for p in parts.as_ref().unwrap_or(&empty) {
                let caps: &[String] = p.captures.as_ref().map(|v| v.as_slice()).unwrap_or(&[]);
                keep.push(Part { ptrs: caps.iter().map(|c| c.as_ptr()).collect(), lens: caps.iter().map(|c| c.len()).collect() });
                let k = keep.last().unwrap();
                lists[side].push(sys::zke_regex_part {
                    fwd: p.verify_re.fwd.as_ptr(),
                    fwd_len: p.verify_re.fwd.len(),
                    bwd: p.verify_re.bwd.as_ptr(),
                    bwd_len: p.verify_re.bwd.len(),
                    n_captures: caps.len() as u32,
                    captures: k.ptrs.as_ptr(),       // the Vec's heap buffer does not move when `keep` grows
                    capture_lens: k.lens.as_ptr(),
                });
            }
        }
        let email = &input.email;
        let mut r = zeroed_result();
        let ext_null = email.external_inputs.iter().any(|x| x.value.is_none()) as u32;
        // SAFETY: every pointer refers into `input`, `keep` or `lists`, all alive until the synchronous call returns.
        let rc = unsafe {
            sys::zke_verify_email_with_regex(
                self.raw,
                email.raw_email.as_ptr(),
                email.raw_email.len(),
                email.from_domain.as_ptr().cast(),
                email.from_domain.len(),
                email.public_key.key.as_ptr(),
                email.public_key.key.len(),
                key_type_code(&email.public_key.key_type) as u32,
                ext_null,
                lists[0].as_ptr(),
                lists[0].len() as u32,
                lists[1].as_ptr(),
                lists[1].len() as u32,
                &mut r,
            )
        };
        if rc != 0 {
            return Err(self.last_error(rc));
        }
        Ok(if r.status == sys::ZKE_OK {
            Ok(EmailWithRegexVerifierOutput { email: email_output(email, &r), regex_matches: regex_matches_of(&input.regex_info) })
        } else {
            Err(Panic { status: r.status, detail: r.detail })
        })
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        // SAFETY: the handle came from zke_engine_create and is destroyed once.
        unsafe { sys::zke_engine_destroy(self.raw) };
    }
}

/// The process-wide engine behind the two free functions (device: the current HIP device).
/// Shared without a lock: the C-ABI's entry points are re-entrant (one submission slot per call in flight).
pub fn default_engine() -> &'static Engine {
    static ENGINE: OnceLock<Engine> = OnceLock::new();
    ENGINE.get_or_init(|| Engine::new(-1).expect("zkemail_amd: no usable GPU engine (there is no CPU fallback)"))
}

/// `zkemail_core::verify_email` (core/src/circuits.rs:9-29): same signature, same panics.
pub fn verify_email(email: &Email) -> EmailVerifierOutput {
    match default_engine().try_verify_email(email) {
        Err(e) => panic!("{e}"),
        Ok(Err(p)) => panic!("{p}"),
        Ok(Ok(out)) => out,
    }
}

/// `zkemail_core::verify_email_with_regex` (core/src/circuits.rs:31-68): same signature, same panics.
pub fn verify_email_with_regex(input: &EmailWithRegex) -> EmailWithRegexVerifierOutput {
    match default_engine().try_verify_email_with_regex(input) {
        Err(e) => panic!("{e}"),
        Ok(Err(p)) => panic!("{p}"),
        Ok(Ok(out)) => out,
    }
}
