/*
 * zkemail_amd.h — C-ABI of the MI355X-native batched email-verification engine.
 *
 * This is the drop-in boundary for zkemail_core's two public functions
 *
 *     pub fn verify_email(email: &Email) -> EmailVerifierOutput             (core/src/circuits.rs:9)
 *     pub fn verify_email_with_regex(input: &EmailWithRegex)
 *                                   -> EmailWithRegexVerifierOutput         (core/src/circuits.rs:31)
 *
 * The reference has no FFI of its own; these entry points are what a
 * Rust `-sys` shim for that path binds (INTEGRATION.md shows the `extern "C"` block).
 * Plain pointers and sizes only; no C++ or torch types.
 *
 * All multi-byte integers are host-endian (little-endian on the target).  Offsets
 * arrays are CSR style: entry i spans blob[off[i] .. off[i+1]).
 *
 * Failure model.  The reference panics (process abort under its release profile,
 * Cargo.toml:35).  A batch engine must not abort a batch, so every email gets a
 * `status` naming the reference panic site that would have fired.  The C++ mirror in
 * include/zkemail_core.hpp re-raises them to keep drop-in semantics.
 */
#ifndef ZKEMAIL_AMD_H
#define ZKEMAIL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status codes */
/* One per reference panic site. */
enum {
  ZKE_OK                  = 0,
  ZKE_PARSE_FAIL          = 1,  /* mailparse::parse_mail(..).unwrap()            core/src/email.rs:26 */
  ZKE_KEY_DECODE_FAIL     = 2,  /* DkimPublicKey::try_from_bytes(..).unwrap()    core/src/email.rs:29 */
  ZKE_DKIM_ERROR          = 3,  /* verify_email_with_key(..).unwrap()            core/src/email.rs:33 */
  ZKE_DKIM_NOT_PASS       = 4,  /* assert!(verified)                             core/src/circuits.rs:13 */
  ZKE_EXTERNAL_INPUT_NULL = 5,  /* .expect("Value cannot be null")               core/src/circuits.rs:24 */
  ZKE_CANON_FAIL          = 6,  /* canonicalize_signed_email(..).unwrap()        core/src/circuits.rs:35 */
  ZKE_DFA_DECODE_FAIL     = 7,  /* dense::DFA::from_bytes(..).unwrap()           core/src/regex.rs:32-33 */
  ZKE_HEADER_REGEX_FAIL   = 8,  /* assert!(verified) on header parts             core/src/circuits.rs:45 */
  ZKE_BODY_REGEX_FAIL     = 9,  /* assert!(verified) on body parts               core/src/circuits.rs:54 */
  ZKE_UNSUPPORTED         = 10  /* input is outside what this engine implements; never a silent mis-verify */
};

/* `detail` sub-codes.  For ZKE_DKIM_NOT_PASS they mirror cfdkim's DKIMError of the
 * last signature tried (the text after "fail (" in DKIMResult::with_detail()). */
enum {
  ZKE_D_NONE                 = 0,
  ZKE_D_NEUTRAL              = 1,  /* no DKIM-Signature with d= == from_domain was tried */
  ZKE_D_SIG_SYNTAX           = 2,  /* tag-list did not parse */
  ZKE_D_MISSING_TAG          = 3,  /* one of v a b bh d h s absent */
  ZKE_D_INCOMPATIBLE_VERSION = 4,  /* v != "1" */
  ZKE_D_DOMAIN_MISMATCH      = 5,  /* i= does not end with d= */
  ZKE_D_FROM_NOT_SIGNED      = 6,  /* h= lacks "from" */
  ZKE_D_BAD_QUERY_METHOD     = 7,  /* q= present and != "dns/txt" */
  ZKE_D_BAD_CANON            = 8,  /* c= not one of the six accepted spellings */
  ZKE_D_BAD_ALGO             = 9,  /* a= not rsa-sha1 / rsa-sha256 / ed25519-sha256 */
  ZKE_D_BAD_LENGTH           = 10, /* l= not a decimal usize */
  ZKE_D_BODY_HASH_MISMATCH   = 11, /* base64(SHA-256(canon body)) != bh= */
  ZKE_D_SIG_B64              = 12, /* b= is not canonical padded base64 */
  ZKE_D_SIG_MISMATCH         = 13, /* RSASSA-PKCS1-v1_5 / Ed25519 verification failed (an Ed25519 b= that is not 64 bytes included) */
  /* ZKE_PARSE_FAIL */
  ZKE_D_HDR_LEADING_SPACE    = 20, /* header line starts with ' ' */
  ZKE_D_HDR_LONE_CR          = 21, /* headers followed by a lone CR */
  ZKE_D_EMPTY_INPUT          = 22,
  ZKE_D_SUBPART_LEADING_SPACE = 23, /* the same two errors in the header block of a MIME subpart (parse_mail walks the multipart tree) */
  ZKE_D_SUBPART_LONE_CR      = 24,
  /* ZKE_KEY_DECODE_FAIL */
  ZKE_D_KEY_TYPE             = 30, /* key_type not "rsa"/"ed25519" */
  ZKE_D_KEY_DER              = 31, /* not a DER RSAPublicKey */
  ZKE_D_KEY_RANGE            = 32, /* modulus > 4096 bits, e < 2 or e > 2^33-1 */
  ZKE_D_KEY_ED25519_POINT    = 33, /* 32-byte Ed25519 key is not a curve point (VerifyingKey::from_bytes fails) */
  /* ZKE_CANON_FAIL */
  ZKE_D_NO_SIGNATURE         = 40, /* no DKIM-Signature header at all */
  /* ZKE_UNSUPPORTED */
  ZKE_D_U_ALGO_SHA1          = 50, /* (unused since a=rsa-sha1 is implemented) */
  ZKE_D_U_ALGO_ED25519       = 51, /* a= and key type disagree (a=ed25519-sha256 with an RSA key, a=rsa-* with an Ed25519
                                      key): cfdkim errors or verifies against the key type; reported, never guessed */
  ZKE_D_U_SIG_NON_ASCII      = 52, /* DKIM-Signature value has bytes >= 0x80 (reference goes through from_utf8_lossy) */
  ZKE_D_U_TOO_MANY_HEADERS   = 53, /* more than ZKE_MAX_HEADERS header fields */
  ZKE_D_U_PREIMAGE_OVERFLOW  = 54, /* canonicalised header preimage exceeds its scratch slot */
  ZKE_D_U_EVEN_MODULUS       = 55,
  ZKE_D_U_TOO_MANY_TAGS      = 56,
  ZKE_D_U_CAPTURE_FFFD       = 57, /* capture holds U+FFFD and the match text is not valid UTF-8 */
  ZKE_D_U_EMAIL_TOO_LARGE    = 58, /* raw email >= 2^31 bytes */
  /* regex */
  ZKE_D_RE_MATCH_COUNT       = 60, /* find_iter(..).count() != 1      core/src/regex.rs:37 */
  ZKE_D_RE_CAPTURE_MISSING   = 61, /* !matched_str.contains(capture)  core/src/regex.rs:44 */
  ZKE_D_RE_QUIT              = 62, /* DFA entered its quit state (find_iter panics in the reference) */
  /* ZKE_UNSUPPORTED, continued */
  ZKE_D_U_SIG_TOO_LONG       = 63, /* FWS-stripped tag values of one DKIM-Signature exceed ZKE_MAX_TAGBUF bytes */
  ZKE_D_U_TOO_MANY_SIGS      = 64, /* more failing same-domain signatures than the engine's signature rounds */
  ZKE_D_U_SIG_B_REPEATED     = 65, /* (no longer produced: both front ends remove every occurrence of the raw b= value, as the reference does) */
  ZKE_D_U_DOMAIN_FOLD        = 66, /* from_domain holds U+212A KELVIN SIGN, the one non-ASCII character whose to_lowercase() is ASCII ("k"):
                                      cfdkim compares d= and from_domain lower-cased as Unicode strings; the engine folds ASCII only,
                                      which is exact for every other domain (d= itself is ASCII whenever it gets that far) */
  ZKE_D_U_MIME_CTYPE         = 67, /* a Content-Type value that decides the subpart walk holds bytes >= 0x80 or an RFC 2047 encoded word */
  ZKE_D_U_MIME_BOUNDARY      = 68, /* multipart boundary parameter folded across lines, or given only in an RFC 2231 form (boundary*, boundary*0) */
  ZKE_D_U_MIME_DEPTH         = 69, /* multiparts nested more than 8 deep */
  /* ZKE_DKIM_NOT_PASS, continued */
  ZKE_D_SIG_EXPIRED          = 14, /* x= lies in the past (only with zke_options.enforce_expiry_x) */
  /* ZKE_DFA_DECODE_FAIL: the section of the regex-automata dense-DFA blob at which dense::DFA::from_bytes would have
   * given up (core/src/regex.rs:32-33).  70..79: the forward blob (verify_re.fwd); + 10: the reverse blob (verify_re.bwd).
   * The sections the committed regex-automata blobs do not pin (unanchored start block, accelerators, quit set)
   * have codes of their own so that the first real blob that fails says where the recalled layout is wrong. */
  ZKE_D_DFA_LABEL            = 70, /* label / padding ("rust-regex-automata-dfa-dense\0", NUL-padded to 32 bytes) */
  ZKE_D_DFA_ENDIAN_VERSION   = 71, /* endianness marker 0x0000FEFF, version 2, the unused word */
  ZKE_D_DFA_FLAGS            = 72, /* the flags word */
  ZKE_D_DFA_TRANSITIONS      = 73, /* state_len, stride2, byte classes, the transition table and its ids */
  ZKE_D_DFA_START_TABLE      = 74, /* start kind, start-byte map, stride, pattern_len, universal starts, the start ids */
  ZKE_D_DFA_MATCH_STATES     = 75, /* match-state slices, pattern_len, pattern ids */
  ZKE_D_DFA_SPECIAL          = 76, /* the eight special-state bounds */
  ZKE_D_DFA_ACCELS           = 77, /* accelerator count and records */
  ZKE_D_DFA_QUITSET          = 78, /* the 256-bit quit set */
  ZKE_D_DFA_UNREGISTERED     = 79  /* the part id names no registered pair (never registered, or unregistered since) */
};
#define ZKE_D_DFA_BWD_OFFSET 10u /* added to ZKE_D_DFA_LABEL .. ZKE_D_DFA_QUITSET when the reverse blob is the one that fails (80..88) */

#define ZKE_MAX_HEADERS 256u   /* header fields per email the device parser tables hold */
#define ZKE_MAX_TAGS    32u    /* tag-specs per DKIM-Signature */
#define ZKE_MAX_TAGBUF  2048u  /* bytes of FWS-stripped tag values per DKIM-Signature */
#define ZKE_MAX_RSA_BYTES 512u /* RSA-4096, the rsa crate's ceiling (rsa 0.9.6 RsaPublicKey::MAX_SIZE) */
#define ZKE_KEY_RSA 0u
#define ZKE_KEY_ED25519 1u
#define ZKE_KEY_OTHER 2u

/* flags in zke_result.flags */
#define ZKE_F_HDR_RELAXED  1u
#define ZKE_F_BODY_RELAXED 2u
#define ZKE_F_HAS_LENGTH   4u
#define ZKE_F_SHA1         8u   /* a=rsa-sha1: body_hash / header_hash hold 20-byte SHA-1 digests, zero padded */
#define ZKE_F_ED25519      16u  /* a=ed25519-sha256 with an Ed25519 key (RFC 8463): Ed25519 over the SHA-256 header hash */

/* ------------------------------------------------------------------ result record */
/* Fixed 192-byte record per email.  from_domain_hash / public_key_hash are the
 * EmailVerifierOutput witnesses (core/src/circuits.rs:16-17); body_hash, header_hash,
 * lengths and the match span are intermediates exposed so parity is checkable
 * (SURVEY.md §0 item 5). external_inputs / regex_matches are echoes of the inputs
 * (circuits.rs:18-27, regex.rs:47) and are reassembled by the host wrapper. */
typedef struct zke_result {
  uint32_t status;            /* ZKE_* */
  uint32_t detail;            /* ZKE_D_* */
  uint32_t sig_index;         /* index among DKIM-Signature headers (file order) that passed / was tried last */
  uint32_t flags;             /* ZKE_F_* of that signature */
  uint32_t canon_header_len;  /* bytes in the header-hash preimage */
  uint32_t canon_body_len;    /* bytes hashed for bh (after l=) */
  uint32_t body_offset;       /* offset of the body in raw_email (first CRLFCRLF + 4) */
  uint32_t n_headers;         /* header fields mailparse would return */
  uint8_t  from_domain_hash[32];
  uint8_t  public_key_hash[32];
  uint8_t  body_hash[32];     /* SHA-256(canon body[..l]) of the signature in sig_index */
  uint8_t  header_hash[32];   /* SHA-256(header preimage) */
  uint32_t regex_part;        /* part index (header parts first, then body parts) checked last; 0xFFFFFFFF if none */
  uint32_t match_count;       /* matches found in that part, saturating at 2 */
  uint32_t match_start;       /* span of its first match */
  uint32_t match_end;
  uint32_t rsa_bits;          /* modulus bit length */
  uint32_t reserved[3];
} zke_result;

/* ------------------------------------------------------------------ batch input */
/* Struct-of-arrays view of &[Email] / &[EmailWithRegex] (core/src/structs.rs:49-62).
 * Caller-owned, read-only for the call.  In zke_verify_batch / zke_verify_batch_async the
 * pointers are host memory and every offset array is checked to be non-decreasing before
 * anything is copied (ZKE_E_ARG otherwise); in zke_verify_batch_device they are device (HBM)
 * pointers and the offsets are trusted like the pointers themselves. */
typedef struct zke_batch {
  uint32_t n;                      /* emails */
  const uint8_t*  raw_blob;        /* Email.raw_email, concatenated            structs.rs:51 */
  const uint64_t* raw_off;         /* [n+1] */
  const uint8_t*  domain_blob;     /* Email.from_domain (UTF-8)                structs.rs:50 */
  const uint64_t* domain_off;      /* [n+1] */
  const uint8_t*  key_blob;        /* Email.public_key.key (PKCS#1 DER for rsa) structs.rs:9 */
  const uint64_t* key_off;         /* [n+1] */
  const uint8_t*  key_type;        /* [n] ZKE_KEY_*  (PublicKey.key_type)      structs.rs:10 */
  const uint8_t*  ext_null;        /* [n] or NULL: 1 if any ExternalInput.value is None (circuits.rs:24) */

  /* regex section (EmailWithRegex.regex_info, structs.rs:32-35,59-62).  The part list
   * is shared by the batch (one regex_config per batch); captures are per email. */
  uint32_t with_regex;             /* 0: verify_email; 1: verify_email_with_regex */
  uint32_t n_header_parts;
  uint32_t n_body_parts;
  const uint32_t* header_part_ids; /* [n_header_parts] ids from zke_dfa_register */
  const uint32_t* body_part_ids;   /* [n_body_parts] */
  /* captures of email i, part p (p over header parts then body parts, P = total):
   * strings cap_str_off[ cap_off[i*P+p] .. cap_off[i*P+p+1] ) in cap_blob. */
  const uint32_t* cap_off;         /* [n*P + 1] or NULL when P == 0 */
  const uint32_t* cap_str_off;     /* [n_strings + 1] */
  const uint8_t*  cap_blob;
} zke_batch;

/* Optional copies of the intermediates, for parity tests (host mode only).  Any
 * pointer may be NULL.  Slot i of a blob starts at i*stride. */
typedef struct zke_debug_out {
  uint8_t* canon_header; size_t canon_header_stride;  /* header-hash preimage */
  uint8_t* canon_body;   size_t canon_body_stride;    /* canonicalised body (before l=) */
  uint8_t* clean_body;   size_t clean_body_stride;    /* after remove_quoted_printable_soft_breaks (email.rs:61-86) */
  uint8_t* em;           size_t em_stride;            /* sig^e mod n, big-endian, k bytes */
  uint32_t* canon_body_full_len;                      /* [n] canonical body length before l= */
  uint32_t* rsa_route;                                /* [n] which RSA routine takes the e-mail's signatures: 4 four lanes per signature,
                                                         8 eight lanes (the key's constants were cached); else one signature per
                                                         wave and why: 0x100 no lane-group kernel for this size in the launch,
                                                         0x200 key not cached yet, 0x400 cache slot taken, 0x800 not eligible */
} zke_debug_out;

/* Engine options.  A zero-filled struct (or NULL) is the default configuration: every field is phrased so that 0 means
 * "what the engine does by default".  ABI 0.3: the fields have names (0.2 kept them in reserved[]). */
typedef struct zke_options {
  int32_t  device;              /* HIP device ordinal; -1 = the calling thread's current device.  NOTE: 0 is device 0. */
  uint32_t slots;               /* submission slots created with the engine (0 = 1); zke_engine_reserve can raise it later */
  uint32_t max_sig_rounds;      /* same-domain DKIM-Signature headers tried per e-mail, in file order, until one passes — cfdkim
                                   tries them all; 0 = 16, at most ZKE_MAX_HEADERS.  The first is tried in the batch's three
                                   launches, later ones inside the last of them by the e-mail's own wave.  An e-mail with more
                                   failing candidates reports ZKE_UNSUPPORTED / ZKE_D_U_TOO_MANY_SIGS. */
  uint32_t disable_key_cache;   /* 1: no per-key Montgomery-constant cache (R^2 mod n recomputed per signature) */
  uint32_t host_threads;        /* worker threads that copy host-entry batches into pinned staging memory (0 = 4; 1 = the caller's
                                   thread alone).  zke_verify_batch[_async] only. */
  uint32_t max_dfas;            /* DFA pairs the registry holds before it evicts pairs that zke_verify_email_with_regex registered
                                   on its own (0 = 4096) */
  /* kernel variants (0 = chosen by batch size; the parity tests force each) */
  uint32_t rsa_lane_groups;     /* 1: never the four- / eight-lanes-per-signature RSA routines; 2: always */
  uint32_t dfa_mapping;         /* 1: one e-mail per lane for every regex part; 2: one e-mail per wave for every part */
  uint32_t replay_graphs;       /* 1: zke_verify_batch_device replays a captured hipGraph when a slot sees the same descriptor
                                   again (measured slower than plain launches on MI355X: DESIGN.md §5) */
  /* Strictness flags: behaviours of the reference's un-vendored crates that could not be verified offline (SURVEY.md
   * Appendix B "open questions").  0 = this engine's reading of cfdkim@75af99fb; 1 = the other reading.  Each
   * flag switches ONE named site in the device front end (csrc/parse.hip.h, ZKE_STRICT_*) and the same site in the CPU
   * oracle (oracle/zke_oracle.c), so a maintainer with the Rust crates at hand flips a field, not a kernel. */
  uint32_t enforce_expiry_x;             /* 1: a signature whose x= tag lies before `now_unix` fails (ZKE_D_SIG_EXPIRED), as
                                            cloudflare/dkim's validate_header does; 0: x= is ignored (a zkVM guest has no clock) */
  uint32_t canon_takes_verified_signature; /* canonicalize_signed_email (core/src/circuits.rs:34-35) — 0: the FIRST DKIM-Signature
                                            header of the e-mail, whatever its d=; 1: the signature verify_dkim accepted */
  uint32_t canon_ignores_l;              /* 0: canonicalize_signed_email truncates the canonical body to l=, as the verify path
                                            does; 1: it returns the whole canonical body */
  uint32_t i_must_be_subdomain;          /* i= check — 0: i= ends with d= (a plain suffix test); 1: the domain of i= (behind its
                                            last '@') equals d= or ends with "." d=, case-insensitively (RFC 6376 §3.5) */
  uint32_t b_removes_own_span_only;      /* 0: the raw b= value is removed wherever it occurs in the header (String::replace);
                                            1: only the b= tag's own span is emptied */
  uint32_t reserved0;
  uint64_t now_unix;            /* the time x= is compared with (enforce_expiry_x); 0 = the host clock at submission */
  uint64_t reserved[4];         /* 0 */
} zke_options;

/* bits of the strictness mask the kernels and the oracle take (one per flag above, same order) */
#define ZKE_STRICT_EXPIRY_X        1u
#define ZKE_STRICT_CANON_VERIFIED  2u
#define ZKE_STRICT_CANON_IGNORES_L 4u
#define ZKE_STRICT_I_SUBDOMAIN     8u
#define ZKE_STRICT_B_OWN_SPAN      16u

typedef struct zke_engine zke_engine;

/* Device time of one batch's launches, microseconds (HIP events on the stream the batch ran on; zke_set_timing). */
typedef struct zke_timings {
  float front_end_us;     /* parse_kernel: header split, key decode, tag lists, header-hash preimage, body canonicalisation */
  float hash_modexp_us;   /* hash_modexp_kernel: the SHA-256 groups beside the RSA roles (one launch) */
  float ed_verdict_us;    /* ed_verdict_kernel: Ed25519 stage, verdicts, later signature rounds */
  float regex_prep_us;    /* verify_email_with_regex: canonicalize_signed_email pass + QP soft-break removal */
  float dfa_us;           /* ... the DFA launches and the regex verdict */
  float total_us;         /* first launch to last launch (copies excluded) */
  float h2d_us, d2h_us;   /* host entry only: the packed input image in, the records out */
} zke_timings;

/* All functions return 0 on success, or a negative code if the CALL failed (bad
 * arguments, device error, extension missing).  They never abort on a bad email. */
#define ZKE_E_ARG     (-1)
#define ZKE_E_DEVICE  (-2)
#define ZKE_E_NOMEM   (-3)
#define ZKE_E_DFA     (-4)

/* Process-wide set-up, optional.  HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the
 * variable once, when the runtime initialises; every submission slot wants a queue of its own, and MI355X runs 24 queues of a
 * process without time-slicing them.  zke_process_init(q) exports GPU_MAX_HW_QUEUES = q (0 = 23: 22 slots + the
 * null stream) unless the variable is already set.  It has an effect only before the process's first HIP call: call it first
 * thing in main().  zke_engine_create calls zke_process_init(0) itself, which covers hosts whose first HIP call is that
 * one; a host that has used HIP before (a framework, another library) keeps the pool it has and should call this itself,
 * earlier.  Nothing else in this library touches the process environment, and nothing runs at load time. */
int zke_process_init(uint32_t hw_queues);

int zke_engine_create(const zke_options* opt, zke_engine** out);
/* Waits for everything in flight; host batches nobody waited for are delivered to their `out` arrays on the way out. */
void zke_engine_destroy(zke_engine* e);
/* The message of the last failed call on this engine by the calling thread ("" if none). */
const char* zke_last_error(const zke_engine* e);

/* Threading.  The reference's two functions are re-entrant (core/src/circuits.rs:9: no state between calls), and so are
 * the entry points here: zke_verify_batch, zke_verify_batch_async / zke_batch_wait, zke_verify_batch_device,
 * zke_verify_email, zke_verify_email_with_regex and zke_dfa_register may be called from any number of host threads on ONE
 * engine at once.  Each call takes the next submission slot (an atomic ticket) and holds that slot's lock while it
 * enqueues; everything a call needs lives in its slot.  zke_engine_reserve, zke_dfa_unregister and zke_engine_destroy
 * are exclusive: they wait for the submissions in progress and hold new ones off. */

/* One engine per GPU.  An engine owns `slots` submission slots — a stream, a private workspace and (host entry) a pinned
 * staging image each — so that many batches can be in flight at once; what batches share (the per-key Montgomery constants
 * of the RSA kernels, the registered DFA tables, kernel attributes) exists once per engine.
 * zke_engine_reserve(e, max_n, max_raw_total, slots, max_regex_parts) raises the slot count to `slots` (1..64) and sizes
 * every slot's workspace for batches of up to max_n e-mails / max_raw_total raw bytes (max_regex_parts > 0: the
 * verify_email_with_regex buffers too), so that no allocation happens in the submit path afterwards.  A larger batch
 * still works: its slot grows (one synchronising reallocation). */
int zke_engine_reserve(zke_engine* e, uint32_t max_n, uint64_t max_raw_total, uint32_t slots, uint32_t max_regex_parts);
/* The same for the host entry's staging: every slot's pinned input image and record buffer (and their HBM twins) are sized
 * for batches of up to max_n e-mails whose inputs — raw e-mails + from_domains + keys (+ captures) — total max_input_bytes.
 * Pinned memory is a host resource (S slots x image bytes): a caller that only uses zke_verify_batch_device never pays it,
 * and without this call a slot's staging comes into being with its first host batch. */
int zke_engine_reserve_host(zke_engine* e, uint32_t max_n, uint64_t max_input_bytes);

/* Parse one regex-automata 0.4 dense-DFA pair once (replaces the per-email
 * dense::DFA::from_bytes of core/src/regex.rs:32-33), validate it and stage a repacked
 * transition table on the device.  Accepts unaligned input, so the align_slice shim
 * (core/src/regex.rs:5-13) is unnecessary.  A blob that from_bytes would reject still
 * gets an id; emails using it report ZKE_DFA_DECODE_FAIL with the failing section as `detail` (ZKE_D_DFA_*).
 * An equal pair registered again gets its old id (the registry is keyed by a hash of the pair). */
int zke_dfa_register(zke_engine* e, const uint8_t* fwd, size_t fwd_len,
                     const uint8_t* bwd, size_t bwd_len, uint32_t* out_id);
/* *detail = 0 when both blobs of pair `id` deserialise, else the ZKE_D_DFA_* section at which from_bytes gives up. */
int zke_dfa_status(zke_engine* e, uint32_t id, uint32_t* detail);
/* Drop a registered pair and free its tables (waits for the batches in flight).  The id may be given out again. */
int zke_dfa_unregister(zke_engine* e, uint32_t id);

/* Host-memory batch — what a drop-in caller has: `&[Email]` in RAM (core/src/circuits.rs:9 takes a host-resident &Email,
 * built at helpers/src/generator.rs:40-45).  The offsets, key types and the three blobs are packed into ONE image in the
 * slot's pinned staging buffer (by `host_threads` workers for large batches), cross PCIe as ONE copy on the slot's
 * stream, the three launches follow, and the n records come back as one copy into pinned memory.
 *   zke_verify_batch_async  returns as soon as everything is enqueued; `in` has been read completely (the caller may reuse
 *                           its buffers), `out` must stay valid until zke_batch_wait(e, *ticket) has returned.  With S slots
 *                           reserved, S batches are in flight; a slot whose previous batch was never waited for completes it
 *                           (its records are delivered) before it is reused.
 *   zke_batch_wait          blocks until that batch's records are in `out`.  Waiting twice, or for a ticket a later batch of
 *                           the same slot has already retired, returns 0 at once.
 *   zke_verify_batch        = async + wait; `dbg` != NULL additionally copies the parity intermediates back (tests). */
int zke_verify_batch(zke_engine* e, const zke_batch* in, zke_result* out, zke_debug_out* dbg);
int zke_verify_batch_async(zke_engine* e, const zke_batch* in, zke_result* out, uint64_t* ticket);
int zke_batch_wait(zke_engine* e, uint64_t ticket);

/* The reference's own layout: n separate Email values, each with buffers of its own (`&[Email]`: a Vec<u8> and two Strings per
 * e-mail, core/src/structs.rs:49-54) instead of three concatenated blobs.  The engine gathers them straight into its pinned
 * staging image on its packing threads — ONE copy between the caller's Vecs and the DMA engine — and computes the CSR offsets
 * itself; a caller that would only concatenate the buffers to call zke_verify_batch saves that pass (a single-threaded copy of
 * the whole batch).  Same pipeline, same records, same ticket protocol as zke_verify_batch_async; verify_email only (regex
 * batches carry per-e-mail capture tables: zke_batch). */
typedef struct zke_email_ref {
  const uint8_t* raw;         size_t raw_len;        /* Email.raw_email */
  const char*    from_domain; size_t domain_len;     /* Email.from_domain (UTF-8, no terminator needed) */
  const uint8_t* key;         size_t key_len;        /* Email.public_key.key */
  uint32_t key_type;                                 /* ZKE_KEY_* of Email.public_key.key_type */
  uint32_t external_input_null;                      /* != 0: some ExternalInput.value is None (circuits.rs:24) */
} zke_email_ref;
int zke_verify_emails(zke_engine* e, const zke_email_ref* emails, uint32_t n, zke_result* out);
int zke_verify_emails_async(zke_engine* e, const zke_email_ref* emails, uint32_t n, zke_result* out, uint64_t* ticket);
/* ... and verify_email_with_regex over `&[EmailWithRegex]` that share one part list (one regex_config per batch): the e-mails as
 * above, the part lists and the per-e-mail capture tables exactly as zke_batch has them (host arrays; small). */
typedef struct zke_regex_lists {
  uint32_t n_header_parts; const uint32_t* header_part_ids;   /* ids from zke_dfa_register, RegexInfo.header_parts order */
  uint32_t n_body_parts;   const uint32_t* body_part_ids;
  const uint32_t* cap_off;         /* [n*P + 1] or NULL: no captures anywhere (as zke_batch.cap_off) */
  const uint32_t* cap_str_off;     /* [n_strings + 1] */
  const uint8_t*  cap_blob;
} zke_regex_lists;
int zke_verify_emails_with_regex(zke_engine* e, const zke_email_ref* emails, uint32_t n, const zke_regex_lists* lists, zke_result* out);
int zke_verify_emails_with_regex_async(zke_engine* e, const zke_email_ref* emails, uint32_t n, const zke_regex_lists* lists,
                                       zke_result* out, uint64_t* ticket);

/* Device-resident batch: every pointer in `in` and `out_dev` is device memory (the part-id lists stay host arrays);
 * `raw_total`, `domain_total`, `key_total` are the blob sizes (the CSR tails), which the
 * host needs for workspace sizing without a device read.  Takes the engine's next submission slot (round-robin),
 * enqueues on `stream` (a hipStream_t; NULL = that slot's own stream) and returns without synchronising: with S slots
 * reserved, S consecutive calls run concurrently.  A slot is reused only behind its previous batch (stream order, or an
 * event wait when the caller's stream changed).  zke_engine_sync waits for every slot. */
int zke_verify_batch_device(zke_engine* e, const zke_batch* in, uint64_t raw_total,
                            uint64_t domain_total, uint64_t key_total,
                            zke_result* out_dev, void* stream);
int zke_engine_sync(zke_engine* e);
/* Device-side join, no host wait: whatever is enqueued on `stream` (a hipStream_t; here NULL is the device's null stream)
 * after this call runs behind every batch submitted so far, on whichever slot or stream it went.  For a consumer of the
 * result records that lives on a stream of its own (a copy, a collective): it can be enqueued while the batches still run. */
int zke_engine_join(zke_engine* e, void* stream);
/* Timings of the last batch that ran in `slot` / in the slot used most recently.  A slot that has not run a timed batch
 * reports all zeros. */
int zke_get_timings(zke_engine* e, zke_timings* t);
int zke_get_slot_timings(zke_engine* e, uint32_t slot, zke_timings* t);
/* Enable per-launch HIP-event timing (adds event records between the launches). */
int zke_set_timing(zke_engine* e, int enabled);

/* Single-email wrappers over a batch of one (config 1 / API-shape parity): the two functions of the reference,
 *     verify_email(&Email)                      core/src/circuits.rs:9
 *     verify_email_with_regex(&EmailWithRegex)  core/src/circuits.rs:31
 * with the struct fields spelled out as pointers and sizes.  `external_input_null` != 0: some ExternalInput.value is
 * None (circuits.rs:24 panics after the DKIM assert and before any regex work; the status order is the reference's). */
int zke_verify_email(zke_engine* e, const uint8_t* raw, size_t raw_len,
                     const char* from_domain, size_t domain_len,
                     const uint8_t* key, size_t key_len, uint32_t key_type,
                     uint32_t external_input_null, zke_result* out);

/* One CompiledRegex of RegexInfo.header_parts / body_parts (core/src/structs.rs:16-35). */
typedef struct zke_regex_part {
  const uint8_t* fwd; size_t fwd_len;        /* verify_re.fwd: regex-automata dense DFA, little-endian, padding stripped */
  const uint8_t* bwd; size_t bwd_len;        /* verify_re.bwd */
  uint32_t n_captures;                       /* captures: Some(v) -> v.len(); None -> 0 (core/src/regex.rs:41 skips the check) */
  const uint8_t* const* captures;            /* [n_captures] UTF-8 bytes, no terminator needed */
  const size_t* capture_lens;                /* [n_captures] */
} zke_regex_part;

/* Registers the DFA pairs it has not seen before (zke_dfa_register returns the old id for an equal pair), so calling
 * this per e-mail with the same regex_config parses every table once, not once per e-mail as core/src/regex.rs:32-33.
 * Pairs registered this way are the ones the registry evicts (least recently used first) when it is full. */
int zke_verify_email_with_regex(zke_engine* e, const uint8_t* raw, size_t raw_len,
                                const char* from_domain, size_t domain_len,
                                const uint8_t* key, size_t key_len, uint32_t key_type,
                                uint32_t external_input_null,
                                const zke_regex_part* header_parts, uint32_t n_header_parts,
                                const zke_regex_part* body_parts, uint32_t n_body_parts,
                                zke_result* out);

/* ---- serialised inputs -------------------------------------------------------------------------
 * The byte streams the zkVM hosts already produce for Email / EmailWithRegex (the derives at core/src/structs.rs:1-6): borsh
 * under the reference's `risc0` feature (u32 lengths), bincode 1.x default options over serde under `sp1` (u64 lengths).  Read
 * in place — every pointer of a zke_wire_email points into the caller's buffer, which must outlive the document. */
#define ZKE_WIRE_BORSH   0u
#define ZKE_WIRE_BINCODE 1u
typedef struct zke_wire_doc zke_wire_doc;
typedef struct zke_wire_email {                 /* Email (structs.rs:49-54) + RegexInfo (structs.rs:32-35) as the entry points take them */
  const uint8_t* raw; size_t raw_len;           /* Email.raw_email */
  const char* from_domain; size_t domain_len;   /* Email.from_domain (UTF-8, validated) */
  const uint8_t* key; size_t key_len;           /* Email.public_key.key */
  uint32_t key_type;                            /* ZKE_KEY_* of Email.public_key.key_type */
  uint32_t n_external_inputs;                   /* Email.external_inputs.len(); zke_wire_external_input reads one */
  uint32_t external_input_null;                 /* 1: some ExternalInput.value is None (circuits.rs:24) */
  uint32_t has_header_parts, has_body_parts;    /* Option tags of RegexInfo.header_parts / body_parts */
  uint32_t n_header_parts, n_body_parts;
  const zke_regex_part* header_parts;           /* [n_header_parts]; captures: None -> n_captures 0 */
  const zke_regex_part* body_parts;
} zke_wire_email;
/* Decode ONE record at bytes[0, len): Email (with_regex == 0) or EmailWithRegex.  *consumed = its size (records may follow
 * each other in a stream).  ZKE_E_ARG with a message (zke_last_error(NULL)) for a truncated or malformed stream. */
int zke_wire_decode(uint32_t format, const uint8_t* bytes, size_t len, uint32_t with_regex, zke_wire_doc** out, size_t* consumed);
void zke_wire_free(zke_wire_doc* d);
int zke_wire_view(const zke_wire_doc* d, zke_wire_email* out);
int zke_wire_external_input(const zke_wire_doc* d, uint32_t i, const uint8_t** name, size_t* name_len,
                            const uint8_t** value, size_t* value_len, uint32_t* is_null);
/* verify_email / verify_email_with_regex of one serialised record: decode + zke_verify_email[_with_regex].  The record must
 * fill bytes[0, len) exactly. */
int zke_verify_wire(zke_engine* e, uint32_t format, const uint8_t* bytes, size_t len, uint32_t with_regex, zke_result* out);

/* ---- one batch over N GPUs ----------------------------------------------------------------------------
 * Every e-mail is verified in isolation, so ranks take contiguous ranges of a batch balanced by cumulative raw BYTES (not by
 * count: a ragged batch then loads the ranks evenly) and the data path has no collective; the only exchange is the gather of
 * the per-e-mail witnesses.  zke_shard_bounds writes bounds[0 .. world]: rank r owns e-mails [bounds[r], bounds[r + 1]) of the
 * n whose CSR offsets are raw_off[0 .. n]; cut points sit where the cumulative byte count crosses r / world of the total.
 * Rank r then runs zke_verify_batch_device on its range — the offsets are absolute, so a range is `raw_off + bounds[r]` with
 * the same blobs — and contributes (bounds[r + 1] - bounds[r]) records.  Pure host code.  (Python:
 * zkemail_rs_amd.distributed.shard_bounds / ShardedVerifier, the latter with the RCCL all-gather of the witnesses.) */
int zke_shard_bounds(const uint64_t* raw_off, uint32_t n, uint32_t world, uint32_t* bounds);

/* Building blocks, exported for parity tests and micro-benchmarks.  Host pointers. */
/* n messages msg_blob[off[i]..off[i+1]) -> digests[32*i..]           (core/src/crypto.rs:3-7) */
int zke_sha256_batch(zke_engine* e, const uint8_t* msg_blob, const uint64_t* off,
                     uint32_t n, uint8_t* digests);
/* n RSA public-key operations: em[i] = sig[i]^e[i] mod n[i].  sig/mod big-endian,
 * `bytes` each (<= 512, multiple of 4); em big-endian `bytes` each.  ok[i] = 0 when
 * sig >= n or n even.  */
int zke_rsa_modexp_batch(zke_engine* e, const uint8_t* sig, const uint8_t* mod,
                         const uint64_t* exp, uint32_t bytes, uint32_t n,
                         uint8_t* em, uint8_t* ok);
/* n independent Ed25519 verifications under the rule cfdkim applies to k=ed25519 keys (ed25519-dalek 2.1.1
 * verify_strict, Cargo.lock:778).  keys n x 32 bytes, msgs n x msg_len bytes (msg_len <= 32: DKIM signs the
 * 32-byte header hash), sigs n x 64 bytes.  out[i]: 0 = key does not decode to a curve point
 * (VerifyingKey::from_bytes fails), 1 = signature rejected, 2 = valid. */
int zke_ed25519_verify_batch(zke_engine* e, const uint8_t* keys, const uint8_t* msgs, uint32_t msg_len,
                             const uint8_t* sigs, uint32_t n, uint32_t* out);
/* Device-resident SHA-256 micro-benchmark entry: messages already in HBM. */
int zke_sha256_batch_device(zke_engine* e, const uint8_t* msg_blob_dev, const uint64_t* off_dev,
                            uint32_t n, uint8_t* digests_dev, void* stream);

/* Solidity ABI encoding of a verifier output — what VerificationOutput::from_parts(email, matches).abi_encode()
 * returns (core/src/io.rs:28-44): abi.encode of
 *     struct SolEmailOutput          { bytes32 from_domain_hash; bytes32 public_key_hash; string[] external_inputs; }
 *     struct SolEmailWithRegexOutput { SolEmailOutput email; string[] matches; }               (core/src/io.rs:5-16)
 * with_matches == 0: EmailOnly; != 0: WithRegex (matches may then be empty).  Strings are UTF-8 bytes with lengths.
 * Pure host code, no engine and no GPU.  Writes *out_len = the encoding's size; returns ZKE_E_NOMEM (nothing written)
 * when out_cap is smaller — call with out == NULL, out_cap == 0 to size the buffer. */
int zke_abi_encode(const uint8_t* from_domain_hash /*[32]*/, const uint8_t* public_key_hash /*[32]*/,
                   const uint8_t* const* external_inputs, const size_t* external_input_lens, uint32_t n_external_inputs,
                   uint32_t with_matches, const uint8_t* const* matches, const size_t* match_lens, uint32_t n_matches,
                   uint8_t* out, size_t out_cap, size_t* out_len);

/* Library / build identification.  zke_abi_version() = 3 for this header (struct layouts and entry points of ABI 0.3). */
const char* zke_version(void);
uint32_t zke_abi_version(void);
/* The reference panic site a status stands for, as text ("assert!(verified)  core/src/circuits.rs:13"); "" for ZKE_OK,
 * "unknown status" beyond the enum.  A static string. */
const char* zke_status_name(uint32_t status);
/* 1 if a HIP device is usable from this process, else 0 (never falls back to a CPU path). */
int zke_device_available(void);

#ifdef __cplusplus
}
#endif
#endif /* ZKEMAIL_AMD_H */
