/*
 * zke_oracle.h — CPU oracle for the zkemail_core hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libzkemail_amd.so) never links or calls it.
 *
 * PARITY UNPINNED by the reference: /root/reference holds no golden vector, fixture or
 * hermetic test for this path and its Rust sources cannot be built
 * here (no cargo/rustc).  The arithmetic lives in un-vendored crates (cfdkim@75af99fb,
 * sha2 0.10.9, rsa 0.9.6, regex-automata 0.4.9, mailparse 0.15.0; Cargo.lock).  This
 * restatement follows their published algorithms (FIPS 180-4, RFC 8017, RFC 6376,
 * RFC 2045, regex-automata's dense-DFA wire format) and the reference's own call sites;
 * it is pinned by NIST / RFC known-answer vectors, the RFC 8463 Appendix A message, two dense
 * DFAs regex-automata itself serialised, and the independent Python signer in tests/synth.py
 * (hashlib, int pow, openssl, `re`).
 */
#ifndef ZKE_ORACLE_H
#define ZKE_ORACLE_H
#include "../include/zkemail_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* core/src/crypto.rs:3-7 hash_bytes */
void zko_sha256(const uint8_t* data, size_t len, uint8_t out[32]);
void zko_sha1(const uint8_t* data, size_t len, uint8_t out[20]);
/* 1 when the SHA-NI path is in use */
int zko_sha256_uses_shani(void);

/* rsa 0.9.6 rsa_encrypt: em = sig^e mod n (big-endian, `bytes` wide). returns 0 ok, -1 if sig>=n / n even */
int zko_rsa_modexp(const uint8_t* sig, const uint8_t* mod, uint32_t bytes, uint64_t e, uint8_t* em);
/* PKCS#1 DER RSAPublicKey -> modulus (big-endian, minimal) and exponent. 0 ok else ZKE_D_* */
int zko_parse_rsa_pkcs1(const uint8_t* der, size_t len, uint8_t* mod_out, uint32_t* mod_len, uint64_t* e_out);
/* RSASSA-PKCS1-v1_5 verify with SHA-256 DigestInfo (rsa 0.9.6 pkcs1v15::verify). 1 = valid */
int zko_rsa_pkcs1v15_sha256_verify(const uint8_t* mod, uint32_t mod_len, uint64_t e,
                                   const uint8_t* sig, uint32_t sig_len, const uint8_t hash[32],
                                   uint8_t* em_out /* may be NULL, mod_len bytes */);

/* oracle/zke_ed25519.c — SHA-512 and the ed25519-dalek 2.1.1 rules cfdkim applies to k=ed25519 keys */
void zko_sha512(const uint8_t* data, size_t len, uint8_t out[64]);
int zko_ed25519_key_decodes(const uint8_t key[32]);                     /* VerifyingKey::from_bytes */
int zko_ed25519_verify_strict(const uint8_t key[32], const uint8_t* msg, size_t msg_len, const uint8_t sig[64]);

/* base64 STANDARD (padded, canonical). dec returns length or -1 */
size_t zko_b64_encode(const uint8_t* in, size_t n, char* out);
long zko_b64_decode(const uint8_t* in, size_t n, uint8_t* out);

/* mailparse 0.15.0 parse_headers restatement: spans (key_start,key_end,val_start,val_end)
 * as 4 uint32 per header; returns header count or -ZKE_D_* on the errors parse_mail raises */
long zko_parse_headers(const uint8_t* raw, size_t len, uint32_t* spans, size_t max_headers,
                       size_t* body_ix);
/* mailparse 0.15.0 parse_mail_recursive (the MIME subpart walk): 0, or ZKE_PARSE_FAIL / ZKE_UNSUPPORTED with *detail */
uint32_t zko_mime_walk(const uint8_t* raw, size_t len, uint32_t* detail);

/* cfdkim canonicalisation (RFC 6376 §3.4).  Output buffers must hold len+4 bytes. */
size_t zko_canon_body(const uint8_t* body, size_t len, int relaxed, uint8_t* out);
size_t zko_canon_header(const uint8_t* key, size_t klen, const uint8_t* val, size_t vlen,
                        int relaxed, uint8_t* out);

/* core/src/email.rs:61-86 remove_quoted_printable_soft_breaks (cleaned bytes, zero padded) */
void zko_remove_qp_soft_breaks(const uint8_t* body, size_t len, uint8_t* out);

/* regex-automata 0.4.9 dense DFA (little-endian wire format). */
int zko_dfa_register(const uint8_t* fwd, size_t fwd_len, const uint8_t* bwd, size_t bwd_len,
                     uint32_t* out_id);
long zko_dfa_status(uint32_t id);   /* 0, or the ZKE_D_DFA_* section at which the pair does not deserialise; -1: no such id */
void zko_dfa_reset(void);
/* find_iter over `hay`: writes up to max_spans (start,end) pairs; returns the count, or -1 on quit,
 * -2 if the id is unknown / blob invalid */
long zko_regex_find_iter(uint32_t id, const uint8_t* hay, size_t len, uint32_t* spans, size_t max_spans);

/* verify_email / verify_email_with_regex over a batch (core/src/circuits.rs:9-68), one
 * email at a time exactly as the reference orders the work.  threads<=1: serial. */
int zko_verify_batch(const zke_batch* in, zke_result* out, zke_debug_out* dbg, int threads);
/* ... with the strictness flags of zke_options as a ZKE_STRICT_* mask (each switches the site of the same name here and
 * in csrc/parse.hip.h) and the time x= is compared with */
int zko_verify_batch_strict(const zke_batch* in, zke_result* out, zke_debug_out* dbg, int threads, uint32_t strict, uint64_t now);

#ifdef __cplusplus
}
#endif
#endif
