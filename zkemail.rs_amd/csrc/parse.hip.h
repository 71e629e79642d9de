// parse.hip.h — device-side restatement of the e-mail front end of verify_dkim, one e-mail per
// wavefront (blockDim = 64).
//
// Replaces, on the verify path (call sites core/src/email.rs:26-33, core/src/circuits.rs:34-35):
//   * mailparse 0.15.0 parse_headers / parse_header (header list) and parse_mail_recursive's walk over the MIME
//     subparts (mime.hip.h: its only effect on this path is the parse error of a malformed subpart header block);
//   * cfdkim validate_header, the tag-list grammar (RFC 6376 §3.2), d= / i= / h= / q= / v= checks,
//     c= / a= / l= parsing, header selection (§5.4.2, bottom-up) and header canonicalisation
//     (§3.4.1 / §3.4.2) producing the header-hash preimage (§3.7);
//   * rsa 0.9.6 RsaPublicKey::from_pkcs1_der + check_public (RFC 8017 A.1.1);
//   * base64 STANDARD decode of b=.
//
// Execution model.  Control flow is wave-uniform (one e-mail, one parser state); the 64 lanes
// are used as a 64-byte-wide scanner: each primitive looks at 64 consecutive bytes at once and
// turns per-byte predicates into 64-bit ballot masks (find first / find last / stream
// compaction by popcount prefix).  The head of the e-mail (3.5 KB: the header block of ordinary mail) is staged
// in LDS with 16-byte lane-contiguous loads; the per-e-mail tables (header spans, tag records, the FWS-stripped
// tag values) live in LDS too.  In round 0 the kernel also does the batch's bookkeeping (batch_prologue), and
// after the parse the same wave canonicalises the body (canon.hip.h) — one launch for the whole front end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "zkemail_amd.h"
#include "rsa.hip.h"
#include "sha256.hip.h"

namespace zke {

constexpr uint32_t OOB = 0x100;       // "no byte here"
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t WNONE = 0x80000000u;   // "no window loaded" (e-mails are < 2^31 bytes)

// Per-e-mail state shared by the kernels of one batch.
struct EmailMeta {
  uint32_t state;             // ST_*
  uint32_t status, detail;    // decided status when state == ST_FINAL
  uint32_t unsupported;       // last ZKE_D_U_* seen on a signature (0 = none)
  uint32_t last_touched_sig;  // oracle's sig_index when nothing passes
  uint32_t cand_total;        // same-domain signatures that reach the hash stage
  uint32_t cand_sig_index;    // signature index (file order) of this round's candidate
  uint32_t cand_hdr;          // header index of this round's candidate
  uint32_t cand_err;          // failure recorded for this round's candidate (finalize)
  uint32_t post_err;          // last non-candidate error located after this round's candidate
  uint32_t pre_err;           // last non-candidate error overall (used when there is no candidate)
  uint32_t flags;             // ZKE_F_*
  uint32_t len_tag_lo, len_tag_hi;
  uint32_t preimage_len;
  uint32_t body_off, body_len;
  uint32_t canon_full_len;    // canonical body length before l=
  uint32_t hashed_len;        // after l=
  uint32_t body_src_is_raw;   // simple canonicalisation hashes the raw bytes in place
  uint32_t n_headers, n_sigs;
  uint32_t first_sig_hdr;     // header index of the first DKIM-Signature (canonicalize_signed_email)
  uint32_t sig_b64_ok;
  uint32_t bh_len;
  uint8_t  bh[48];            // FWS-stripped bh= (first 48 chars)
  uint32_t key_ok;            // 1 = RSA key decoded, 2 = 32-byte Ed25519 key (its point check runs in ed25519_email_kernel)
  uint32_t even_modulus;
  uint32_t reuse;             // mode 1: the first DKIM-Signature is the verified one; scratch of the verify pass is reused
  uint32_t ed_key_bad;        // Ed25519 key is not a curve point (VerifyingKey::from_bytes fails): verdict = KEY_DECODE_FAIL
  uint32_t ed_ok;             // Ed25519 signature of this round's candidate verifies
  uint32_t rsa_route;         // RSA_F_QUAD / RSA_F_OCT: the key's Montgomery constants are cached and a lane-group kernel takes the modexp (round 0)
  uint32_t em_ok;             // the RSA role's result: EM = sig^e mod n has the EMSA-PKCS1-v1_5 shape for this signature's hash ...
  uint32_t em_tail[8];        // ... and EM's trailing digest bytes (little-endian limbs), compared with the header hash by verdict_kernel
};
static_assert(sizeof(EmailMeta) % 8 == 0, "EmailMeta alignment");

enum : uint32_t { ST_FINAL = 0, ST_CAND = 1, ST_PENDING = 2 };
enum : int { TG_V = 0, TG_A, TG_B, TG_BH, TG_D, TG_H, TG_S, TG_I, TG_Q, TG_C, TG_L, TG_X, TG_N };

// Batch view handed to the kernels (device pointers).
struct BatchDev {
  uint32_t n;
  const uint8_t* raw; const uint64_t* raw_off;
  const uint8_t* dom; const uint64_t* dom_off;
  const uint8_t* key; const uint64_t* key_off;
  const uint8_t* key_type; const uint8_t* ext_null;
  zke_result* results;
  EmailMeta* meta;
  RsaJob* rsa;
  ShaJob* sha;                // kind-major: sha[kind * n_pad + i], kinds 0 body, 1 header, 2 domain, 3 key
  uint32_t n_pad;
  uint8_t* scratch;           // per e-mail: region A (preimage) then region B (canonical body)
  uint64_t* scratch_off;      // [n+1]; region A = raw_len + PRE_SLACK bytes, region B = raw_len + 16 (written by the round-0 front end)
  uint64_t* clean_off;        // [n+1]; offsets of the QP-cleaned bodies (regex stage), written alongside
  uint32_t* pending;          // e-mails waiting for another signature round (reset by the round-0 front end)
  uint32_t* order;            // length buckets of the body and header-preimage hashes (sha256.hip.h, ShaOrder): 2 x 256 counters (zero at
                              // the start of a batch: the verdict launch of the previous one clears them), then 2 x n_pad keys; or nullptr
  const EmailMeta* meta_verify; // mode 1 only: the verify pass's meta
};
constexpr uint32_t PRE_SLACK = 1024;
constexpr uint32_t HDR_LDS_ENTRIES = 64;                // header spans kept in LDS; entries 64..255 overflow to the scratch slot
constexpr uint32_t HDR_OVF_BYTES = (ZKE_MAX_HEADERS - HDR_LDS_ENTRIES) * 16;
constexpr uint32_t SCR_PER_EMAIL = HDR_OVF_BYTES + PRE_SLACK + 64;      // fixed part of an e-mail's scratch slot
constexpr uint32_t CLEAN_PER_EMAIL = 32;

// slot i = [header-span overflow (HDR_OVF_BYTES)] [region A] [region B], at align16(2 * (raw_off[i] - raw_off[0])) + i * SCR_PER_EMAIL;
// scratch_off[i] is the offset of region A
// clean_off[i]   = (raw_off[i] - raw_off[0]) + i * CLEAN_PER_EMAIL
__host__ __device__ inline uint64_t scratch_offset(uint64_t rel, uint32_t i) { return ((2 * rel + 15) & ~15ull) + (uint64_t)i * SCR_PER_EMAIL + HDR_OVF_BYTES; }
__host__ __device__ inline uint64_t clean_offset(uint64_t rel, uint32_t i) { return rel + (uint64_t)i * CLEAN_PER_EMAIL; }

// Round-0 bookkeeping of a batch, done by the front-end kernel itself instead of three tiny launches (an offsets
// kernel and two memsets, each a dispatch that queues behind the other batches in flight): the thread that owns
// e-mail i publishes its two offsets; `lead` threads (t of nt, all in the block that owns e-mail 0) clear the
// SHA jobs of the padding lanes of each kind's last wave, the (n+1)-th offsets and the pending counter.
__device__ __forceinline__ void batch_prologue(const BatchDev& B, uint32_t i, bool owner, bool lead, uint32_t t, uint32_t nt) {
  const uint64_t base = B.raw_off[0];
  if (owner) {
    const uint64_t rel = B.raw_off[i] - base;
    B.scratch_off[i] = scratch_offset(rel, i);
    B.clean_off[i] = clean_offset(rel, i);
  }
  if (lead) {
    for (uint32_t k = 0; k < 4; k++)
      for (uint32_t o = B.n + t; o < B.n_pad; o += nt) { ShaJob z{0, 0, 0, 0}; B.sha[(size_t)k * B.n_pad + o] = z; }
    if (t == 0) {
      const uint64_t rel = B.raw_off[B.n] - base;
      B.scratch_off[B.n] = scratch_offset(rel, B.n);
      B.clean_off[B.n] = clean_offset(rel, B.n);
      *B.pending = 0;
    }
  }
}
constexpr uint32_t PARSE_STAGE_BYTES = 3568;   // header blocks beyond this are read from HBM past the staged part

// ------------------------------------------------------------------ byte strings and windows
struct Str {                  // logical string over global memory with one optional excision
  const uint8_t* base;
  uint32_t len;               // logical length
  uint32_t cut, skip;         // logical index >= cut reads base[index + skip]
  const uint8_t* lds;         // LDS copy of base[0 .. lds_len) (the staged head of the e-mail), or nullptr
  uint32_t lds_len;
};
__device__ __forceinline__ Str mkstr(const uint8_t* b, uint32_t len) { return Str{b, len, NONE, 0, nullptr, 0}; }
// s[a, b) as a string of its own (keeps the LDS window)
__device__ __forceinline__ Str substr(const Str& s, uint32_t a, uint32_t b) {
  Str r{s.base + a, b - a, NONE, 0, nullptr, 0};
  if (a < s.lds_len) { r.lds = s.lds + a; r.lds_len = s.lds_len - a; }
  return r;
}
__device__ __forceinline__ uint32_t ldb(const Str& s, uint32_t l) {
  if (l >= s.len) return OOB;
  const uint32_t phys = l + (l >= s.cut ? s.skip : 0u);
  const uint8_t* p = phys < s.lds_len ? s.lds + phys : s.base + phys;     // one flat load either way
  return (uint32_t)*p;
}
struct Win { uint32_t wpos; uint32_t c; };   // c = byte at logical wpos + lane (OOB beyond the end)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ void wload(const Str& s, Win& w, uint32_t pos) { w.wpos = pos; w.c = ldb(s, pos + lane_id()); }
__device__ __forceinline__ uint32_t at(const Str& s, Win& w, uint32_t pos) {   // uniform pos -> uniform byte
  if (pos - w.wpos >= 64u) wload(s, w, pos);
  return __builtin_amdgcn_readlane(w.c, pos - w.wpos);
}
// A value every lane holds alike (loaded from LDS or memory at a wave-uniform address), as a scalar: branches on it are
// then scalar branches, not exec-mask regions — the parser's control flow is wave-uniform, but the compiler can only see
// that where the values come out of v_readfirstlane / v_readlane.
__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) {     // set bits of m in the lanes below this one
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ uint64_t bits_from(uint32_t rel) { return rel >= 64 ? 0ull : (~0ull << rel); }
__device__ __forceinline__ uint64_t bits_below(uint32_t rel) { return rel >= 64 ? ~0ull : ((1ull << rel) - 1); }

// first l in [pos, end) with pred(byte), else end
template <class P>
__device__ __forceinline__ uint32_t wfind(const Str& s, Win& w, uint32_t pos, uint32_t end, P pred) {
  while (pos < end) {
    if (pos - w.wpos >= 64u) wload(s, w, pos);
    const uint32_t rel = pos - w.wpos;
    uint64_t m = __ballot(pred(w.c) && (w.wpos + lane_id()) < end) & bits_from(rel);
    if (m) return w.wpos + (uint32_t)__builtin_ctzll(m);
    pos = w.wpos + 64;
  }
  return end;
}
// last l in [start, end) with pred(byte), else NONE
template <class P>
__device__ __forceinline__ uint32_t wrfind(const Str& s, Win& w, uint32_t start, uint32_t end, P pred) {
  uint32_t pos = end;
  while (pos > start) {
    const uint32_t ws = (pos - start > 64u) ? pos - 64u : start;
    if (!(ws >= w.wpos && pos <= w.wpos + 64u)) wload(s, w, ws);
    const uint32_t l = w.wpos + lane_id();
    uint64_t m = __ballot(pred(w.c) && l >= ws && l < pos);
    if (m) return w.wpos + 63u - (uint32_t)__builtin_clzll(m);
    pos = ws;
  }
  return NONE;
}

__device__ __forceinline__ bool is_fws(uint32_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
__device__ __forceinline__ bool is_wsp(uint32_t c) { return c == ' ' || c == '\t'; }
__device__ __forceinline__ bool is_valchar(uint32_t c) { return (c >= 0x21 && c <= 0x3a) || (c >= 0x3c && c <= 0x7e); }
__device__ __forceinline__ bool is_alpha(uint32_t c) { return ((c | 0x20) >= 'a' && (c | 0x20) <= 'z') && c < 0x80; }
__device__ __forceinline__ bool is_alnumpunc(uint32_t c) { return is_alpha(c) || (c >= '0' && c <= '9') || c == '_'; }
__device__ __forceinline__ uint32_t lower(uint32_t c) { return (c >= 'A' && c <= 'Z') ? c + 32 : c; }

// case-insensitive equality of s[a, a+n) with a byte string in LDS / constant / global memory
__device__ __forceinline__ bool span_ieq(const Str& s, uint32_t a, uint32_t n, const uint8_t* name, uint32_t nlen) {
  if (n != nlen) return false;
  for (uint32_t o = 0; o < n; o += 64) {
    const uint32_t l = o + lane_id();
    bool bad = false;
    if (l < n) bad = lower(ldb(s, a + l)) != lower(name[l]);
    if (__ballot(bad)) return false;
  }
  return true;
}

// value of the neighbouring lane (0 beyond the wave): DPP wave shifts
__device__ __forceinline__ uint32_t lane_shl1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /*wave_shl:1: lane l <- l+1*/, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t lane_shr1(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138 /*wave_shr:1: lane l <- l-1*/, 0xf, 0xf, false); }

// Short literals (<= 16 bytes) as four constant words: byte `lane` of one comes out of two v_perm_b32 on constants —
// no per-lane load from constant memory (a dependent L2 round trip per comparison) and no hoisted per-lane address of
// every literal (two registers each, live across the whole signature loop: they were what the front end spilled).
struct Lit { uint32_t w0, w1, w2, w3, n; };
template <size_t N>
constexpr Lit LIT(const char (&s)[N]) {
  static_assert(N >= 1 && N - 1 <= 16, "literal too long");
  uint32_t w[4] = {0, 0, 0, 0};
  for (size_t i = 0; i + 1 < N; i++) w[i / 4] |= (uint32_t)(uint8_t)s[i] << (8 * (i % 4));
  return Lit{w[0], w[1], w[2], w[3], (uint32_t)(N - 1)};
}
__device__ __forceinline__ uint32_t lit_at(const Lit& t, uint32_t lane) {      // byte `lane` (< 16) of the literal
  const uint32_t sel = (lane & 7u) | 0x0c0c0c00u;
  const uint32_t lo = __builtin_amdgcn_perm(t.w1, t.w0, sel);
  if (t.n <= 8) return lo;
  const uint32_t hi = __builtin_amdgcn_perm(t.w3, t.w2, sel);
  return (lane & 8u) ? hi : lo;
}
// case-insensitive equality of s[a, a+n) with a lower-case literal
__device__ __forceinline__ bool span_ieq(const Str& s, uint32_t a, uint32_t n, const Lit& t) {
  if (n != t.n) return false;
  const uint32_t l = (uint32_t)lane_id();
  const bool bad = l < n && lower(ldb(s, a + l)) != lit_at(t, l);
  return __ballot(bad) == 0;
}

// ------------------------------------------------------------------ output stream (preimage)
struct Out { uint8_t* p; uint32_t o, cap; bool overflow; };
__device__ __forceinline__ void emit_lit(Out& out, const Lit& t) {
  if (out.o + t.n > out.cap) { out.overflow = true; return; }
  if ((uint32_t)lane_id() < t.n) out.p[out.o + lane_id()] = (uint8_t)lit_at(t, (uint32_t)lane_id());
  out.o += t.n;
}
template <class F>   // copy s[a,b) through a byte map
__device__ __forceinline__ void emit_map(Out& out, const Str& s, uint32_t a, uint32_t b, F f) {
  if (b <= a) return;
  if (out.o + (b - a) > out.cap) { out.overflow = true; return; }
  for (uint32_t base = a; base < b; base += 64) {            // uniform trip count: the loop control stays on the scalar unit
    const uint32_t l = base + (uint32_t)lane_id();
    if (l < b) out.p[out.o + (l - a)] = (uint8_t)f(ldb(s, l));
  }
  out.o += b - a;
}

// cfdkim canonicalize_header_relaxed value part: unfold (drop CRLF), WSP runs -> one SP, trim both ends.
// One forward pass, one load per lane and 64-byte chunk (issued a chunk ahead; the neighbours come over DPP):
// a WSP run becomes the single SP written IN FRONT OF the next kept non-WSP byte, so leading runs (nothing
// emitted yet) and trailing runs (no such byte) vanish without a backward scan for the last kept byte.
__device__ __forceinline__ uint64_t uni64(uint64_t x) {
  return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | (uint64_t)uni((uint32_t)x);      // (the builtin returns int: widen as unsigned)
}
__device__ __forceinline__ void emit_relaxed_value(Out& out, const Str& v) {
  const uint32_t L = v.len;
  if (L == 0) return;
  const int lane = lane_id();
  // (every loop-carried value below is wave-uniform; said so with v_readfirstlane once per step, or the compiler takes the
  // loop for a divergent one — counters in vector registers, an exec-mask ledger around every branch)
  uint64_t cin = 0;            // a kept WSP byte lies behind the last kept non-WSP byte in front of this chunk
  uint32_t started = 0;        // a non-WSP byte has been emitted
  uint32_t prev_last = OOB;    // raw byte in front of this chunk
  uint32_t o = uni(out.o);
  const uint32_t cap = uni(out.cap);
  uint32_t cur = ldb(v, (uint32_t)lane);
  for (uint32_t base = 0; base < L; base += 64) {
    const uint32_t nxt = ldb(v, base + 64 + lane);                 // next chunk, in flight during this one
    const uint32_t l = base + lane;
    uint32_t cp = lane_shr1(cur); if (lane == 0) cp = prev_last;
    const uint32_t nfirst = __builtin_amdgcn_readfirstlane(nxt);     // outside the lane test: all lanes active here
    uint32_t cn = lane_shl1(cur); if (lane == 63) cn = nfirst;
    const bool crlf = (cur == '\r' && cn == '\n') || (cur == '\n' && cp == '\r');
    const bool kept = l < L && !crlf;
    const bool wsp = is_wsp(cur);
    const bool nw = kept && !wsp;
    const uint64_t W = __ballot(kept && wsp), N = __ballot(nw);
    // "a kept WSP lies between byte i and the last kept non-WSP byte below it" is a carry chain over the chunk — a WSP byte
    // generates, a dropped byte (CRLF, beyond the end) propagates, a non-WSP byte kills — solved for all 64 bytes by ONE
    // 64-bit addition on the scalar unit (a = generate | propagate, b = generate: the carry INTO bit i is sum ^ a ^ b).
    const uint64_t a = ~N, b = W;
    const uint64_t sum = a + b + cin;
    const uint64_t C = sum ^ a ^ b;
    uint64_t S = N & C;                                            // non-WSP bytes that get the run's single SP in front
    S &= started ? ~0ull : ~(N & (0 - N));                         // ... except the first one of the value: leading WSP vanishes
    S = uni64(S);
    cin = uni64(((a & b) | ((a | b) & ~sum)) >> 63);               // the carry out of bit 63
    const uint32_t cnt = uni((uint32_t)__builtin_popcountll(N) + (uint32_t)__builtin_popcountll(S));
    if (o + cnt > cap) { out.o = o; out.overflow = true; return; }
    if (nw) {
      uint8_t* dst = out.p + o + lanes_below(N) + lanes_below(S);
      if ((S >> lane) & 1) { dst[0] = (uint8_t)' '; dst[1] = (uint8_t)cur; } else dst[0] = (uint8_t)cur;
    }
    o = uni(o + cnt);
    started = uni(started | (N != 0 ? 1u : 0u));
    prev_last = __builtin_amdgcn_readlane(cur, 63);
    cur = nxt;
  }
  out.o = o;
}

// cfdkim canonicalize_header_{relaxed,simple}(key, value); the CRLF is appended by the caller's choice
__device__ __forceinline__ void emit_header(Out& out, const Str& key, const Str& val, bool relaxed, bool crlf) {
  if (relaxed) {
    uint32_t ke = key.len;
    if (ke && is_wsp(uni(ldb(key, ke - 1)))) {            // WSP in front of the ':' (rare): the name ends at its last non-WSP byte
      Win w; w.wpos = WNONE; w.c = 0;
      ke = wrfind(key, w, 0, key.len, [](uint32_t c) { return !is_wsp(c); });
      ke = (ke == NONE) ? 0 : ke + 1;
    }
    if (ke < 64) {                                        // the lower-cased name and its ':' as one store
      if (out.o + ke + 1 > out.cap) { out.overflow = true; return; }
      const uint32_t l = (uint32_t)lane_id();
      if (l <= ke) out.p[out.o + l] = l < ke ? (uint8_t)lower(ldb(key, l)) : (uint8_t)':';
      out.o += ke + 1;
    } else {
      emit_map(out, key, 0, ke, [](uint32_t c) { return lower(c); });
      emit_lit(out, LIT(":"));
    }
    emit_relaxed_value(out, val);
  } else {
    emit_map(out, key, 0, key.len, [](uint32_t c) { return c; });
    emit_lit(out, LIT(": "));
    emit_map(out, val, 0, val.len, [](uint32_t c) { return c; });
  }
  if (crlf) emit_lit(out, LIT("\r\n"));
}

// ------------------------------------------------------------------ LDS image of one wave
constexpr uint32_t SEG_CAP = 40;       // tag-specs the lane-per-tag parse looks at: ZKE_MAX_TAGS + 1 decide everything
static_assert(SEG_CAP > ZKE_MAX_TAGS && SEG_CAP <= 64, "one lane per tag-spec");
struct ParseLds {
  uint32_t hdr[4 * HDR_LDS_ENTRIES];   // key_start, key_end, val_start, val_end of headers 0..127 (hdr_get / hdr_put)
  uint32_t tag[TG_N][4];               // raw_s, raw_e, val_off, val_len (last occurrence wins, as IndexMap::insert)
  uint8_t tagbuf[ZKE_MAX_TAGBUF];      // FWS-stripped tag values
  uint16_t seg[5][SEG_CAP];            // lane-per-tag parse: position of the k-th ';', and of tag-spec k: raw_s, raw_e, val_off, val_end
  __attribute__((aligned(16))) uint8_t stage[PARSE_STAGE_BYTES];   // head of the e-mail (header block), copied in 16-byte lanes
};

// Header span table: 64 entries in LDS (10 KB of LDS per wave would cap a CU at 16 front-end waves; 6.7 KB lets the
// register budget decide: 24), the rest — e-mails with more than 64 header fields — in the e-mail's scratch slot.
// The overflow is written and read back by the same wave: the reads go around L1 (agent-scope atomic loads).
struct HdrSpan { uint32_t ks, ke, vs, ve; };
__device__ __forceinline__ void hdr_put(ParseLds& L, uint32_t* ovf, uint32_t x, uint32_t ks, uint32_t ke, uint32_t vs, uint32_t ve) {
  if (lane_id() != 0) return;
  uint32_t* p = x < HDR_LDS_ENTRIES ? L.hdr + 4 * x : ovf + 4 * (x - HDR_LDS_ENTRIES);
  p[0] = ks; p[1] = ke; p[2] = vs; p[3] = ve;
}
__device__ __forceinline__ HdrSpan hdr_get(const ParseLds& L, const uint32_t* ovf, uint32_t x) {
  if (x < HDR_LDS_ENTRIES) return HdrSpan{uni(L.hdr[4 * x]), uni(L.hdr[4 * x + 1]), uni(L.hdr[4 * x + 2]), uni(L.hdr[4 * x + 3])};
  const uint32_t* p = ovf + 4 * (x - HDR_LDS_ENTRIES);
  auto ld = [](const uint32_t* q) { return uni(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
  return HdrSpan{ld(p), ld(p + 1), ld(p + 2), ld(p + 3)};
}

// strip FWS from v[rs,re) into tagbuf at *tb; returns false on overflow
__device__ __forceinline__ bool strip_to_lds(ParseLds& L, const Str& v, uint32_t rs, uint32_t re, uint32_t& tb) {
  for (uint32_t base = rs; base < re; base += 64) {
    const uint32_t l = base + lane_id();
    const uint32_t c = l < re ? ldb(v, l) : OOB;
    const bool k = l < re && !is_fws(c);
    const uint64_t m = __ballot(k);
    const uint32_t cnt = (uint32_t)__builtin_popcountll(m);
    if (tb + cnt > ZKE_MAX_TAGBUF) return false;
    if (k) L.tagbuf[tb + (uint32_t)__builtin_popcountll(m & bits_below(lane_id()))] = (uint8_t)c;
    tb += cnt;
  }
  return true;
}
__device__ __forceinline__ uint32_t tagf(const ParseLds& L, int id, int k) { return uni(L.tag[id][k]); }    // k: 0 raw_s, 1 raw_e, 2 val_off, 3 val_len
// a tag's stripped value, first 16 bytes: lane l holds byte l; compared with literals without touching memory again
struct TagVal { uint32_t len, c; };
__device__ __forceinline__ TagVal tagval(const ParseLds& L, int id) {
  TagVal v{tagf(L, id, 3), 0};
  if ((uint32_t)lane_id() < v.len && lane_id() < 16) v.c = L.tagbuf[tagf(L, id, 2) + lane_id()];
  return v;
}
__device__ __forceinline__ bool operator==(const TagVal& v, const Lit& t) {
  if (v.len != t.n) return false;
  const uint32_t l = (uint32_t)lane_id();
  return __ballot(l < t.n && v.c != lit_at(t, l)) == 0;
}
__device__ __forceinline__ bool tagval_eq(const ParseLds& L, int id, const Lit& t) { return tagval(L, id) == t; }

// tag-spec = [FWS] tag-name [FWS] "=" [FWS] tag-value [FWS]   (cfdkim parser::tag_spec)
// returns the position after the spec, or NONE.  err: 0 ok, else ZKE_D_U_*
__device__ __forceinline__ uint32_t parse_tag_spec(ParseLds& L, const Str& v, Win& w, uint32_t pos, uint32_t& present,
                                                   uint32_t& ntags, uint32_t& tb, uint32_t& err) {
  const uint32_t n = v.len;
  uint32_t p = wfind(v, w, pos, n, [](uint32_t c) { return !is_fws(c); });
  if (p >= n || !is_alpha(at(v, w, p))) return NONE;
  const uint32_t ns = p;
  const uint32_t ne = wfind(v, w, p, n, [](uint32_t c) { return !is_alnumpunc(c); });
  p = wfind(v, w, ne, n, [](uint32_t c) { return !is_fws(c); });
  if (p >= n || at(v, w, p) != '=') return NONE;
  const uint32_t rs = wfind(v, w, p + 1, n, [](uint32_t c) { return !is_fws(c); });
  const uint32_t rend = wfind(v, w, rs, n, [](uint32_t c) { return !(is_valchar(c) || is_fws(c)); });
  uint32_t re = wrfind(v, w, rs, rend, [](uint32_t c) { return is_valchar(c); });
  re = (re == NONE) ? rs : re + 1;
  ntags++;
  if (ntags > ZKE_MAX_TAGS) { err = ZKE_D_U_TOO_MANY_TAGS; return rend; }
  // classify the name (case-sensitive, exact)
  int id = -1;
  const uint32_t c0 = at(v, w, ns);
  if (ne - ns == 1) {
    switch (c0) {
      case 'v': id = TG_V; break; case 'a': id = TG_A; break; case 'b': id = TG_B; break; case 'd': id = TG_D; break;
      case 'h': id = TG_H; break; case 's': id = TG_S; break; case 'i': id = TG_I; break; case 'q': id = TG_Q; break;
      case 'c': id = TG_C; break; case 'l': id = TG_L; break; case 'x': id = TG_X; break; default: break;
    }
  } else if (ne - ns == 2 && c0 == 'b' && at(v, w, ns + 1) == 'h') {
    id = TG_BH;
  }
  const uint32_t off = tb;
  if (!strip_to_lds(L, v, rs, re, tb)) { err = ZKE_D_U_SIG_TOO_LONG; return rend; }
  if (id >= 0) {
    if (lane_id() == 0) { L.tag[id][0] = rs; L.tag[id][1] = re; L.tag[id][2] = off; L.tag[id][3] = tb - off; }
    present |= 1u << id;
  }
  return rend;
}

// The tag list, serially: one tag-spec after the other, each scan 64 bytes wide (the later signature rounds inside the
// verdict launch, and whatever the lane-per-tag parse below hands back).  0, ZKE_D_SIG_SYNTAX or ZKE_D_U_*
__device__ __forceinline__ uint32_t taglist_serial(ParseLds& L, const Str& v, uint32_t& present) {
  Win w; w.wpos = WNONE; w.c = 0;
  uint32_t ntags = 0, tb = 0, err = 0;
  present = 0;
  if (wfind(v, w, 0, v.len, [](uint32_t c) { return c >= 0x80 && c != OOB; }) < v.len) return ZKE_D_U_SIG_NON_ASCII;
  uint32_t p = parse_tag_spec(L, v, w, 0, present, ntags, tb, err);
  if (p == NONE) return ZKE_D_SIG_SYNTAX;
  while (!err && p < v.len && at(v, w, p) == ';') {
    const uint32_t q = parse_tag_spec(L, v, w, p + 1, present, ntags, tb, err);
    if (q == NONE) break;
    p = q;
  }
  return err;
}

constexpr uint32_t TL_SERIAL = 0xFFFFFFF0u;      // "take taglist_serial"

// The tag list, one LANE per tag-spec.  A tag value holds no ';' (valchar excludes it), a name and FWS do not either,
// so the tag-specs are exactly the ';'-separated pieces of v — up to the first byte that is neither valchar, FWS nor
// ';': the value it sits in ends there and nothing behind it is parsed (parse_tag_list stops silently), which is the
// parse of v cut off at that byte.  Three steps:
//   1. 64 bytes of v per step: the ';' write their positions into seg[0] by rank; first bad byte; any byte >= 0x80;
//   2. lane k walks the head of piece k ([FWS] name [FWS] "=" [FWS]) and its tail (FWS in front of the ';') byte by
//      byte — a handful of bytes each, all pieces at once; the first piece that is no tag-spec ends the list;
//   3. 64 bytes per step again: a byte looks its piece up by the number of ';' in front of it and is kept when it lies
//      in the piece's value and is not FWS (compaction into tagbuf); the bytes at raw_s / raw_e - 1 note where the
//      stripped value begins and ends.
// A piece whose head or tail needs more than WALK_BUDGET steps (long runs of FWS, long names) hands the whole list to
// taglist_serial, as does a value of 64 KB or more.
constexpr uint32_t WALK_BUDGET = 40;
__device__ __forceinline__ uint32_t taglist_lanes(ParseLds& L, const Str& v, uint32_t& present) {
  const uint32_t lane = (uint32_t)lane_id();
  const uint32_t n = v.len;
  present = 0;
  if (n > 0xFFFFu) return TL_SERIAL;
  // ---- 1
  uint32_t nsemi = 0, neff = n;
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t p = base + lane;
    const uint32_t c = ldb(v, p);                       // OOB behind the end
    if (__ballot(p < n && c >= 0x80)) return ZKE_D_U_SIG_NON_ASCII;
    if (base < neff) {
      const bool semi = c == ';';
      const uint64_t bm = __ballot(p < n && !(is_valchar(c) || is_fws(c) || semi));
      const uint32_t lim = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
      const uint64_t sm = __ballot(semi) & bits_below(lim);
      const uint32_t rank = nsemi + lanes_below(sm);
      if (semi && lane < lim && rank < SEG_CAP) L.seg[0][rank] = (uint16_t)p;
      nsemi += (uint32_t)__builtin_popcountll(sm);
      if (bm) neff = base + lim;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- 2
  const uint32_t nseg = nsemi + 1 < SEG_CAP ? nsemi + 1 : SEG_CAP;
  const bool active = lane < nseg;
  uint32_t s = 0, e = neff;
  if (active) {
    if (lane > 0) s = (uint32_t)L.seg[0][lane - 1] + 1u;
    if (lane < nsemi) e = (uint32_t)L.seg[0][lane];
  }
  uint32_t budget = active ? WALK_BUDGET : 0u;
  uint32_t p = s, c = OOB;
  auto walk = [&](auto pred) {        // p: first position in [p, e) whose byte fails pred; c: that byte (OOB at e)
    for (;;) {
      c = (p < e && budget) ? ldb(v, p) : OOB;
      if (c == OOB || !pred(c)) break;
      p++; budget--;
    }
  };
  bool ok = active;
  walk([](uint32_t x) { return is_fws(x); });
  ok = ok && is_alpha(c);
  const uint32_t ns = p, c0 = c;
  if (ok) { p++; walk([](uint32_t x) { return is_alnumpunc(x); }); }
  const uint32_t nl = p - ns;
  uint32_t c1 = 0;
  if (ok && nl == 2) c1 = ldb(v, ns + 1);
  if (ok) walk([](uint32_t x) { return is_fws(x); });
  ok = ok && c == '=';
  if (ok) { p++; walk([](uint32_t x) { return is_fws(x); }); }
  const uint32_t rs = p;
  uint32_t re = e;
  if (ok) {
    for (;;) {                         // everything in [rs, e) is valchar or FWS: the last valchar is the last non-FWS
      if (!(re > rs && budget)) break;
      if (!is_fws(ldb(v, re - 1))) break;
      re--; budget--;
    }
  }
  if (__ballot(active && budget == 0)) return TL_SERIAL;
  const uint64_t nm = __ballot(active && !ok);
  const uint32_t T = nm ? (uint32_t)__builtin_ctzll(nm) : nseg;
  if (T == 0) return ZKE_D_SIG_SYNTAX;
  const uint32_t Tcap = T < ZKE_MAX_TAGS ? T : ZKE_MAX_TAGS;
  const bool mine = lane < Tcap;
  // the name (case-sensitive, exact); the last tag of a name wins (IndexMap::insert)
  int id = -1;
  if (mine && nl == 1) {
    switch (c0) {
      case 'v': id = TG_V; break; case 'a': id = TG_A; break; case 'b': id = TG_B; break; case 'd': id = TG_D; break;
      case 'h': id = TG_H; break; case 's': id = TG_S; break; case 'i': id = TG_I; break; case 'q': id = TG_Q; break;
      case 'c': id = TG_C; break; case 'l': id = TG_L; break; case 'x': id = TG_X; break; default: break;
    }
  } else if (mine && nl == 2 && c0 == 'b' && c1 == 'h') {
    id = TG_BH;
  }
  bool winner = false;
#pragma unroll
  for (int x = 0; x < TG_N; x++) {
    const uint64_t m = __ballot(id == x);
    if (m) { present |= 1u << x; winner = winner || (id == x && lane == 63u - (uint32_t)__builtin_clzll(m)); }
  }
  if (mine) { L.seg[1][lane] = (uint16_t)rs; L.seg[2][lane] = (uint16_t)(re > rs ? re : rs); L.seg[3][lane] = 0; L.seg[4][lane] = 0; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- 3
  uint32_t tb = 0, before = 0;         // bytes kept, ';' passed
  for (uint32_t base = 0; base < neff && before < Tcap; base += 64) {
    const uint32_t q = base + lane;
    const uint32_t b = q < neff ? ldb(v, q) : OOB;
    const uint64_t sm = __ballot(b == ';');
    const uint32_t k = before + lanes_below(sm);
    before += (uint32_t)__builtin_popcountll(sm);
    uint32_t vs = 0, ve = 0;
    if (k < Tcap) { vs = L.seg[1][k]; ve = L.seg[2][k]; }
    const bool keep = q >= vs && q < ve && !is_fws(b);
    const uint64_t km = __ballot(keep);
    const uint32_t d = tb + lanes_below(km);
    if (keep && d < ZKE_MAX_TAGBUF) {
      L.tagbuf[d] = (uint8_t)b;
      if (q == vs) L.seg[3][k] = (uint16_t)d;
      if (q + 1 == ve) L.seg[4][k] = (uint16_t)(d + 1);
    }
    tb += (uint32_t)__builtin_popcountll(km);
  }
  if (tb > ZKE_MAX_TAGBUF) return ZKE_D_U_SIG_TOO_LONG;
  if (T > ZKE_MAX_TAGS) return ZKE_D_U_TOO_MANY_TAGS;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (winner) {
    const uint32_t off = L.seg[3][lane], end = L.seg[4][lane];
    L.tag[id][0] = rs; L.tag[id][1] = re > rs ? re : rs; L.tag[id][2] = off; L.tag[id][3] = end - off;
  }
  return 0;
}

// cfdkim validate_header over the header value v.  0 = valid, else ZKE_D_* (ZKE_D_U_SIG_NON_ASCII: a byte >= 0x80 —
// from_utf8_lossy would rewrite it — is reported, never guessed)
// `strict`: ZKE_STRICT_* (zke_options' strictness flags: the readings of cfdkim that could not be verified offline); `now`: the
// time x= is compared with.
template <bool FAST>
__device__ __forceinline__ uint32_t validate_sig(ParseLds& L, const Str& v, uint32_t& present, uint32_t strict, uint64_t now) {
  uint32_t err = FAST ? taglist_lanes(L, v, present) : TL_SERIAL;
  if (err == TL_SERIAL) err = taglist_serial(L, v, present);
  if (err) return err;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const uint32_t req = (1u << TG_V) | (1u << TG_A) | (1u << TG_B) | (1u << TG_BH) | (1u << TG_D) | (1u << TG_H) | (1u << TG_S);
  if ((present & req) != req) return ZKE_D_MISSING_TAG;
  if (!tagval_eq(L, TG_V, LIT("1"))) return ZKE_D_INCOMPATIBLE_VERSION;
  if (present & (1u << TG_I)) {
    // STRICTNESS SITE i_must_be_subdomain (oracle: validate_header, same name).  Default: user.ends_with(signing_domain), a plain
    // suffix test on the bytes.  ZKE_STRICT_I_SUBDOMAIN: the domain of i= (behind its last '@'; the whole value without one)
    // equals d= or ends with "." d=, ASCII case folded (RFC 6376 §3.5 "same as or a subdomain of").
    const uint32_t il = tagf(L, TG_I, 3), dl = tagf(L, TG_D, 3), io = tagf(L, TG_I, 2), dofs = tagf(L, TG_D, 2);
    const bool sub = (strict & ZKE_STRICT_I_SUBDOMAIN) != 0;
    uint32_t dom_s = 0;                 // where the domain of i= starts (sub only)
    if (sub) {
      for (uint32_t o = 0; o < il; o += 64) {
        const uint32_t l = o + lane_id();
        const uint64_t m = __ballot(l < il && L.tagbuf[io + l] == '@');
        if (m) dom_s = o + 64u - (uint32_t)__builtin_clzll(m);
      }
    }
    if (il - dom_s < dl) return ZKE_D_DOMAIN_MISMATCH;
    bool bad = false;
    for (uint32_t o = 0; o < dl; o += 64) {
      const uint32_t l = o + lane_id();
      if (l < dl) {
        const uint32_t a = L.tagbuf[io + il - dl + l], b = L.tagbuf[dofs + l];
        bad |= sub ? lower(a) != lower(b) : a != b;
      }
    }
    if (__ballot(bad)) return ZKE_D_DOMAIN_MISMATCH;
    if (sub && il - dom_s > dl && uni(L.tagbuf[io + il - dl - 1]) != '.') return ZKE_D_DOMAIN_MISMATCH;
  }
  {   // h= must name "from" (split on ':', lower-cased)
    const uint32_t ho = tagf(L, TG_H, 2), hl = tagf(L, TG_H, 3);
    bool found = false;
    for (uint32_t o = 0; o < hl; o += 64) {
      const uint32_t l = o + lane_id();
      if (l + 4 <= hl) {
        const uint8_t* h = L.tagbuf + ho;
        const bool st = (l == 0) || h[l - 1] == ':';
        const bool en = (l + 4 == hl) || h[l + 4] == ':';
        found |= st && en && lower(h[l]) == 'f' && lower(h[l + 1]) == 'r' && lower(h[l + 2]) == 'o' && lower(h[l + 3]) == 'm';
      }
    }
    if (!__ballot(found)) return ZKE_D_FROM_NOT_SIGNED;
  }
  if ((present & (1u << TG_Q)) && !tagval_eq(L, TG_Q, LIT("dns/txt"))) return ZKE_D_BAD_QUERY_METHOD;
  if ((strict & ZKE_STRICT_EXPIRY_X) && (present & (1u << TG_X))) {
    // STRICTNESS SITE enforce_expiry_x (oracle: validate_header, same name).  Default: x= is ignored — a zkVM guest has no clock.
    // ZKE_STRICT_EXPIRY_X: cloudflare/dkim's rule — x= parsed as i64 (str::parse: optional sign, digits, no overflow; anything
    // else counts as 0), fifteen minutes of drift allowed, expired when now > x + 900.
    const uint8_t* xs = L.tagbuf + tagf(L, TG_X, 2);
    const uint32_t xn = tagf(L, TG_X, 3);
    uint32_t k = 0;
    bool neg = false, okx = xn > 0;
    if (okx) { const uint32_t c = uni(xs[0]); if (c == '+' || c == '-') { neg = c == '-'; k = 1; okx = xn > 1; } }
    uint64_t mag = 0;
    const uint64_t lim = neg ? (1ull << 63) : (1ull << 63) - 1;
    for (; okx && k < xn; k++) {
      const uint32_t c = uni(xs[k]);
      if (c < '0' || c > '9') { okx = false; break; }
      const uint64_t d = c - '0';
      if (mag > (lim - d) / 10) { okx = false; break; }
      mag = mag * 10 + d;
    }
    const int64_t x = okx ? (neg ? (int64_t)(0 - mag) : (int64_t)mag) : 0;
    const int64_t deadline = x > INT64_MAX - 900 ? INT64_MAX : x + 900;
    if ((int64_t)now > deadline) return ZKE_D_SIG_EXPIRED;
  }
  return 0;
}

// usize::from_str (optional '+', decimal digits, no overflow) on a stripped tag value
__device__ __forceinline__ bool parse_usize_tag(const ParseLds& L, int id, uint64_t& out) {
  const uint8_t* s = L.tagbuf + tagf(L, id, 2);
  const uint32_t n = tagf(L, id, 3);
  uint32_t i = 0;
  if (n && uni(s[0]) == '+') i = 1;
  if (i >= n) return false;
  uint64_t v = 0;
  for (; i < n; i++) {
    const uint32_t c = uni(s[i]);
    if (c < '0' || c > '9') return false;
    const uint64_t d = c - '0';
    if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10) return false;
    v = v * 10 + d;
  }
  out = v;
  return true;
}

// ------------------------------------------------------------------ mailparse header split
// parse_headers / parse_header over `raw`: sink(ix, key_end, vs, ve) for every header field (false = stop: too many).
// Returns the header count or NONE with perr set; hdr_end = where the list stopped (the empty line, or the end).
template <class Sink>
__device__ __forceinline__ uint32_t scan_headers(const Str& raw, uint32_t& perr, uint32_t& hdr_end, Sink sink) {
  Win w; w.wpos = WNONE; w.c = 0;
  const uint32_t len = raw.len;
  uint32_t ix = 0, nh = 0;
  perr = 0;
  for (;;) {
    if (ix >= len) break;
    const uint32_t c0 = at(raw, w, ix);
    if (c0 == '\n') break;
    if (c0 == '\r') {
      if (ix + 1 < len && at(raw, w, ix + 1) == '\n') break;
      perr = ZKE_D_HDR_LONE_CR;
      return NONE;
    }
    if (c0 == ' ') { perr = ZKE_D_HDR_LEADING_SPACE; return NONE; }
    uint32_t key_end, vs, ve, next;
    const uint32_t p = wfind(raw, w, ix, len, [](uint32_t c) { return c == ':' || c == '\n'; });
    if (p >= len) {
      key_end = len; vs = ve = len; next = len;
    } else if (at(raw, w, p) == '\n') {
      key_end = p; vs = ve = p; next = p + 1;
    } else {
      key_end = p;
      vs = wfind(raw, w, p + 1, len, [](uint32_t c) { return c != ' '; });
      // header end: first LF at q >= vs whose successor is not SP / HTAB (or which ends the input)
      uint32_t q = vs;
      for (;;) {
        q = wfind(raw, w, q, len, [](uint32_t c) { return c == '\n'; });
        if (q >= len) break;
        const uint32_t nx = (q + 1 < len) ? at(raw, w, q + 1) : OOB;
        if (nx == ' ' || nx == '\t') { q++; continue; }
        break;
      }
      next = (q < len) ? q + 1 : len;
      const uint32_t lim = (q < len) ? q : len;
      const uint32_t lastv = wrfind(raw, w, vs, lim, [](uint32_t c) { return c != '\r' && c != '\n'; });
      ve = (lastv == NONE) ? vs : lastv + 1;
    }
    if (!sink(ix, key_end, vs, ve)) { perr = ZKE_D_U_TOO_MANY_HEADERS; return NONE; }
    nh++;
    ix = next;
  }
  hdr_end = ix;
  return nh;
}
// Fills L.hdr; returns the header count or NONE with perr set.
__device__ __forceinline__ uint32_t split_headers(ParseLds& L, uint32_t* ovf, const Str& raw, uint32_t& perr, uint32_t& hdr_end) {
  uint32_t nh = 0;
  const uint32_t r = scan_headers(raw, perr, hdr_end, [&](uint32_t ix, uint32_t key_end, uint32_t vs, uint32_t ve) {
    if (nh >= ZKE_MAX_HEADERS) return false;
    hdr_put(L, ovf, nh, ix, key_end, vs, ve);
    nh++;
    return true;
  });
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return r;
}

// The same split for a header block that lies inside the staged head (every ordinary e-mail), LINE-parallel: the serial
// form walked the block header by header with a handful of dependent single-byte LDS reads each (2.5 k of the front end's
// 9 k scalar instructions per e-mail, 22 % of its cycles).
//   pass A, 64 bytes per step: ballots of LF and ':'; every LF lane stores its position under its line number, every
//   lane holding the FIRST ':' of its line stores that; stops at the first empty line (LF followed by LF or CRLF), which
//   ends the header list whatever the state (an empty line never starts with SP / HTAB).
//   pass B, 64 lines per step, one lane per line: W = the line starts with SP / HTAB, C = it holds a ':'.  "Inside a
//   value after line i" is V[i] = C[i] | (W[i] & V[i-1]) — a carry chain, solved for all 64 lines by ONE 64-bit addition
//   (a = W | C, b = C: carry into bit i+1 = V[i]).  Line i continues a header iff W[i] & V[i-1]; every other line starts
//   one (or is an error: SP first, CR not followed by LF).  A header's lane stores key and value start, the lane of its
//   last line (the next line does not continue it) stores the value end.
// Returns false when it does not apply — no empty line inside the staged part, or more than LINE_CAP lines before it —
// and the caller runs split_headers above.
constexpr uint32_t LINE_CAP = ZKE_MAX_TAGBUF / 4;      // two u16 tables in the (still unused) tag buffer
__device__ __forceinline__ bool split_headers_lines(ParseLds& L, uint32_t* ovf, const Str& raw, uint32_t& nh_out, uint32_t& perr,
                                                    uint32_t& hdr_end) {
#ifdef ZKE_NO_LINE_SPLIT
  return false;
#endif
  const uint32_t len = raw.len, staged = raw.lds_len;
  const uint32_t lane = (uint32_t)lane_id();
  uint16_t* lfpos = (uint16_t*)L.tagbuf;               // line g = [g ? lfpos[g-1] + 1 : 0, lfpos[g]]
  uint16_t* colpos = lfpos + LINE_CAP;                 // its first ':' (0xFFFF: none)
  perr = 0;
  auto sb = [&](uint32_t pos) -> uint32_t { return pos < staged ? (uint32_t)L.stage[pos] : OOB; };
  if (len == 0) { nh_out = 0; hdr_end = 0; return true; }
  {
    const uint32_t c0 = uni(sb(0)), c1 = uni(sb(1));
    if (c0 == '\n' || (c0 == '\r' && c1 == '\n')) { nh_out = 0; hdr_end = 0; return true; }     // the list is empty
  }
  // ---- pass A
  uint32_t n_lines = 0, cut = NONE;
  bool colon_seen = false;                             // the line that runs into this chunk already holds a ':'
  if (lane == 0) colpos[0] = 0xFFFF;
  for (uint32_t base = 0; base < staged; base += 64) {
    const uint32_t l = base + lane;
    const uint32_t c = sb(l), n1 = sb(l + 1), n2 = sb(l + 2);
    uint64_t Lm = __ballot(c == '\n'), Cm = __ballot(c == ':');
    const uint64_t Bm = __ballot(c == '\n' && (n1 == '\n' || (n1 == '\r' && n2 == '\n')));
    uint32_t cutbit = 64;
    if (Bm) { cutbit = (uint32_t)__builtin_ctzll(Bm); Lm &= bits_below(cutbit + 1); Cm &= bits_below(cutbit); }
    const uint32_t line = n_lines + lanes_below(Lm);             // the line this byte lies in (its LF closes it)
    if ((Lm >> lane) & 1) {
      if (line < LINE_CAP) lfpos[line] = (uint16_t)l;
      if (line + 1 < LINE_CAP) colpos[line + 1] = 0xFFFF;
    }
    // "no ':' yet in this line" in front of every byte: a carry chain — an LF generates, a ':' kills, everything else passes
    // it on — solved by one 64-bit addition (a = generate | propagate = ~Cm, b = generate = Lm; the carry INTO bit i is sum ^ a ^ b)
    const uint64_t ca = ~Cm, cb = Lm, cin = colon_seen ? 0ull : 1ull;
    const uint64_t sum = ca + cb + cin;
    const uint64_t F = Cm & (sum ^ ca ^ cb);                       // the first ':' of each line
    if (((F >> lane) & 1) && line < LINE_CAP) colpos[line] = (uint16_t)l;
    colon_seen = (((ca & cb) | ((ca | cb) & ~sum)) >> 63) == 0;    // no carry out: a ':' behind the chunk's last LF
    n_lines += (uint32_t)__builtin_popcountll(Lm);
    if (Bm) { cut = base + cutbit; break; }
  }
  if (cut == NONE || n_lines > LINE_CAP) return false;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ---- pass B
  uint32_t nh = 0;
  uint64_t vcarry = 0;                                 // inside a value after the previous group's last line
  for (uint32_t G = 0; G < n_lines; G += 64) {
    const uint32_t g = G + lane;
    const bool valid = g < n_lines;
    uint32_t s = 0, e = 0, c0 = OOB, c1 = OOB, col = 0xFFFF, nx = OOB;
    if (valid) {
      s = g ? (uint32_t)lfpos[g - 1] + 1u : 0u;
      e = lfpos[g];
      c0 = L.stage[s]; c1 = L.stage[s + 1];            // a line in front of the first empty one holds a byte besides its LF
      col = colpos[g];
      nx = L.stage[e + 1];                             // staged: pass A read it
    }
    const bool W = c0 == ' ' || c0 == '\t', C = col != 0xFFFF;
    const uint64_t VM = __ballot(valid), Wm = __ballot(valid && W), Cm = __ballot(valid && C);
    const uint64_t a = Wm | Cm, b = Cm;
    const uint64_t Vprev = (a + b + vcarry) ^ a ^ b;   // bit i: inside a value after line i - 1
    const uint64_t cont = Wm & Vprev;
    const uint64_t HS = VM & ~cont;                    // lines that start a header
    const bool hs = (HS >> lane) & 1;
    const uint64_t below = bits_below(lane);
    const uint32_t hr = nh + (uint32_t)__builtin_popcountll(HS & below);
    const uint64_t bad = (HS & __ballot(valid && (c0 == ' ' || (c0 == '\r' && c1 != '\n')))) | __ballot(hs && hr >= ZKE_MAX_HEADERS);
    if (bad) {                                         // the first one in line order, as the serial walk meets them
      const uint32_t f = (uint32_t)__builtin_ctzll(bad);
      const uint32_t fc = __builtin_amdgcn_readlane(c0, f);
      // (a header-start line whose first byte is CR is not followed by LF: that would be the empty line)
      perr = fc == ' ' ? ZKE_D_HDR_LEADING_SPACE : fc == '\r' ? ZKE_D_HDR_LONE_CR : ZKE_D_U_TOO_MANY_HEADERS;
      nh_out = NONE;
      return true;
    }
    if (hs) {
      uint32_t ke = e, vs = e;
      if (C) { ke = col; vs = col + 1; while (L.stage[vs] == ' ') vs++; }      // ends at the line's LF at the latest
      uint32_t* p = hr < HDR_LDS_ENTRIES ? L.hdr + 4 * hr : ovf + 4 * (hr - HDR_LDS_ENTRIES);
      p[0] = s; p[1] = ke; p[2] = vs;
      if (!C) p[3] = e;                                // a line without ':' is a key with an empty value
    }
    const bool inval = hs ? C : true;                  // inside a value after this line
    if (valid && inval && !(nx == ' ' || nx == '\t')) {   // the last line of a header with a value: strip trailing CR / LF
      uint32_t q = e;
      while (q > s) { const uint32_t t = L.stage[q - 1]; if (t != '\r' && t != '\n') break; q--; }
      const uint32_t h = nh + (uint32_t)__builtin_popcountll(HS & bits_below(lane + 1)) - 1u;
      uint32_t* p = h < HDR_LDS_ENTRIES ? L.hdr + 4 * h : ovf + 4 * (h - HDR_LDS_ENTRIES);
      p[3] = q;
    }
    nh += (uint32_t)__builtin_popcountll(HS);
    vcarry = ((Cm >> 63) & 1) | (((Wm >> 63) & 1) & ((Vprev >> 63) & 1));
  }
  hdr_end = cut + 1;                                   // the empty line
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  nh_out = nh;
  return true;
}

// cfdkim get_body: everything after the first CRLFCRLF (empty when there is none).
// `from`: a position no CRLFCRLF starts in front of.  After split_headers that is hdr_end - 2: bytes p..p+3 =
// CRLFCRLF make p+2 a line start that begins with CRLF (the LF at p+1 cannot continue a folded line: CR follows),
// and the header split stops at the first such line start, so p + 2 >= hdr_end.
__device__ __forceinline__ uint32_t find_body(const Str& raw, uint32_t from) {
  const uint32_t len = raw.len;
  for (uint32_t base = from; base + 4 <= len; base += 61) {
    const uint32_t c = ldb(raw, base + lane_id());
    const uint64_t r = __ballot(c == '\r'), n = __ballot(c == '\n');
    uint64_t m = r & (n >> 1) & (r >> 2) & (n >> 3) & bits_below(61);
    if (m) return base + (uint32_t)__builtin_ctzll(m) + 4;
  }
  return len;
}

// ------------------------------------------------------------------ PKCS#1 RSAPublicKey DER
__device__ __forceinline__ uint32_t der_len(const Str& k, Win& w, uint32_t p, uint32_t avail, uint32_t& out) {
  if (avail < 1) return 0;
  const uint32_t b0 = at(k, w, p);
  if (b0 < 0x80) { out = b0; return 1; }
  const uint32_t nb = b0 & 0x7f;
  if (nb == 0 || nb > 4 || nb + 1 > avail) return 0;
  uint32_t v = 0;
  for (uint32_t i = 0; i < nb; i++) v = (v << 8) | at(k, w, p + 1 + i);
  if (at(k, w, p + 1) == 0) return 0;
  if (nb == 1 && v < 0x80) return 0;
  if (nb == 4 && (v >> 31)) return 0;
  out = v;
  return nb + 1;
}
// INTEGER at p: value span [vp, vp+vl) with one sign octet stripped; returns bytes used or 0
__device__ __forceinline__ uint32_t der_uint(const Str& k, Win& w, uint32_t p, uint32_t avail, uint32_t& vp, uint32_t& vl) {
  if (avail < 2 || at(k, w, p) != 0x02) return 0;
  uint32_t l, c = der_len(k, w, p + 1, avail - 1, l);
  if (!c || l == 0 || (uint64_t)1 + c + l > avail) return 0;
  uint32_t s = p + 1 + c;
  const uint32_t v0 = at(k, w, s);
  if (v0 & 0x80) return 0;
  if (l > 1 && v0 == 0 && !(at(k, w, s + 1) & 0x80)) return 0;
  vp = s; vl = l;
  if (l > 1 && v0 == 0) { vp = s + 1; vl = l - 1; }
  return 1 + c + l;
}
// returns 0 or ZKE_D_KEY_*; fills the RSA job's modulus / exponent
__device__ __forceinline__ uint32_t decode_rsa_key(const Str& k, RsaJob* J, uint32_t& bits_out, uint32_t& even, uint32_t& np_out,
                                                   uint32_t& nl_out, uint64_t& e_out) {
  Win w; w.wpos = WNONE; w.c = 0;
  const uint32_t len = k.len;
  if (len < 2 || at(k, w, 0) != 0x30) return ZKE_D_KEY_DER;
  uint32_t sl, c = der_len(k, w, 1, len - 1, sl);
  if (!c || (uint64_t)1 + c + sl != len) return ZKE_D_KEY_DER;
  uint32_t p = 1 + c, avail = sl, np, nl, ep, el;
  uint32_t used = der_uint(k, w, p, avail, np, nl);
  if (!used) return ZKE_D_KEY_DER;
  p += used; avail -= used;
  used = der_uint(k, w, p, avail, ep, el);
  if (!used || used != avail) return ZKE_D_KEY_DER;
  const uint32_t n0 = at(k, w, np);
  uint32_t bits = 0;
  if (!(nl == 1 && n0 == 0)) bits = nl * 8 - (uint32_t)(__builtin_clz(n0) - 24);
  if (bits > 4096) return ZKE_D_KEY_RANGE;
  if (el > 8) return ZKE_D_KEY_RANGE;
  uint64_t e = 0;
  for (uint32_t i = 0; i < el; i++) e = (e << 8) | at(k, w, ep + i);
  if (e < 2 || e > ((1ull << 33) - 1)) return ZKE_D_KEY_RANGE;
  // modulus, big-endian, right-aligned in the 512-byte field (zero fill in front)
  {
    // (the eight loads of a lane leave together; interleaved with the stores each waited for its own round trip)
    uint8_t mb[8];
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) {
      const uint32_t o = (uint32_t)lane_id() + 64 * t;
      mb[t] = o >= 512 - nl ? (uint8_t)ldb(k, np + (o - (512 - nl))) : (uint8_t)0;
    }
#pragma unroll
    for (uint32_t t = 0; t < 8; t++) J->mod[(uint32_t)lane_id() + 64 * t] = mb[t];
  }
  even = !(at(k, w, np + nl - 1) & 1) && bits != 0;
  if (lane_id() == 0) { J->e = e; J->k = nl; J->bits = bits; }
  bits_out = bits; np_out = np; nl_out = nl; e_out = e;
  return 0;
}

// Which RSA routine takes this key's signatures: RSA_F_QUAD / RSA_F_OCT when the lane-group kernel for its size is part of
// the batch's launch (mask bit 0 / 1) and the key's Montgomery constants are in the cache — the modulus [np, np + nl) of
// the DER key compared limb by limb with the entry, so a hit is exact; 0 = the one-signature-per-wave routine (which
// fills the cache for the next batch).
__device__ __forceinline__ uint32_t rsa_route(const Str& k, uint32_t np, uint32_t nl, uint32_t bits, const KeyCacheEntry* cache, uint32_t mask) {
  // (the bits above RSA_F_* say why a key was not routed: zke_debug_out.rsa_route shows them to the tests)
  if (bits < 512 || !(mask & (bits <= 2048 ? 1u : 2u))) return 0x100;
  const uint32_t lane = (uint32_t)lane_id();
  auto limb = [&](uint32_t L) -> uint32_t {       // little-endian 32-bit limb L of the big-endian modulus
    uint32_t v = 0;
#pragma unroll
    for (uint32_t b = 0; b < 4; b++) { const uint32_t pos = 4 * L + b; if (pos < nl) v |= ldb(k, np + nl - 1 - pos) << (8 * b); }
    return v;
  };
  const uint32_t l0 = limb(lane), l1 = limb(64 + lane);
  const KeyCacheEntry* E = cache + key_cache_slot(__builtin_amdgcn_readfirstlane(l0), __builtin_amdgcn_readlane(l0, 1));
  // All four loads leave before the first answer is looked at: each is an agent-scope load served at the coherence point, a
  // round trip of a microsecond or more, and three of them in a row (state, then bits, then the limbs) were a twentieth of the
  // wave's life.  An entry is written once and immutable from state == 2 on (rsa_kernel.hip.h): limbs served before the
  // publication beside a state served after it can only turn a hit into a miss — the wave routine then takes the signature.
  const uint32_t st = ld_agent(&E->state), eb = ld_agent(&E->bits), m0 = ld_agent(&E->mod[lane]), m1 = ld_agent(&E->mod[64 + lane]);
  if (st != 2u) return 0x200;                                       // not cached (yet)
  if (eb != bits) return 0x400;                                     // the slot belongs to another key
  const bool same = m0 == l0 && m1 == l1;
  return __ballot(!same) == 0 ? (bits <= 2048 ? (uint32_t)RSA_F_QUAD : (uint32_t)RSA_F_OCT) : 0x400u;
}

// base64 STANDARD decode of tagbuf[off, off+n) into J->sig (right-aligned).  false = not canonical base64.
__device__ __forceinline__ uint32_t b64v(uint32_t c) {
  if (c >= 'A' && c <= 'Z') return c - 'A';
  if (c >= 'a' && c <= 'z') return c - 'a' + 26;
  if (c >= '0' && c <= '9') return c - '0' + 52;
  if (c == '+') return 62;
  if (c == '/') return 63;
  return 64;
}
__device__ __forceinline__ bool decode_sig(const ParseLds& L, uint32_t off, uint32_t n, RsaJob* J, uint32_t& sig_len) {
  sig_len = 0;
  const uint8_t* s = L.tagbuf + off;
  uint32_t pad = 0, total = 0;
  const bool shape_ok = (n % 4) == 0;
  if (shape_ok && n) {
    pad = (uni(s[n - 1]) == '=') ? ((uni(s[n - 2]) == '=') ? 2u : 1u) : 0u;
    total = 3 * (n / 4) - pad;
  }
  {   // zero the part of the field the decoded bytes will not cover
    const uint32_t zend = (shape_ok && total <= 512) ? 512 - total : 512;
    for (uint32_t base = 0; base < zend; base += 64) { const uint32_t o = base + (uint32_t)lane_id(); if (o < zend) J->sig[o] = 0; }
  }
  if (!shape_ok) return false;
  if (n == 0) return true;
  bool bad = false;
  for (uint32_t q0 = 0; q0 < n / 4; q0 += 64) {
    const uint32_t q = q0 + lane_id();
    if (q < n / 4) {
      const bool last = (q + 1 == n / 4);
      const uint32_t a = b64v(s[4 * q]), b = b64v(s[4 * q + 1]);
      uint32_t c = b64v(s[4 * q + 2]), d = b64v(s[4 * q + 3]);
      uint32_t nb = 3;
      if (last && pad == 2) { c = 0; d = 0; nb = 1; if (b & 15) bad = true; }
      else if (last && pad == 1) { d = 0; nb = 2; if (c < 64 && (c & 3)) bad = true; }
      if (a > 63 || b > 63 || c > 63 || d > 63) bad = true;
      const uint32_t v = (a << 18) | (b << 12) | (c << 6) | d;
      if (total <= 512) {
        const uint32_t dst = 512 - total + 3 * q;
        J->sig[dst] = (uint8_t)(v >> 16);
        if (nb > 1) J->sig[dst + 1] = (uint8_t)(v >> 8);
        if (nb > 2) J->sig[dst + 2] = (uint8_t)v;
      }
    }
  }
  sig_len = total;
  return __ballot(bad) == 0;
}

// Does v[p, p+n) equal v[a, a+n)?
__device__ __forceinline__ bool same_bytes(const Str& v, uint32_t p, uint32_t a, uint32_t n) {
  for (uint32_t o = 0; o < n; o += 64) {
    const uint32_t l = o + lane_id();
    bool bad = l < n && ldb(v, p + l) != ldb(v, a + l);
    if (__ballot(bad)) return false;
  }
  return true;
}

}  // namespace zke
#include "mime.hip.h"
namespace zke {

// ------------------------------------------------------------------ the parse kernel
// mode 0: verify_email_with_key scan (round r picks the r-th same-domain candidate)
// mode 1: canonicalize_signed_email (first DKIM-Signature header, no domain filter; core/src/circuits.rs:34-35)
struct ParseArgs {
  BatchDev b; uint32_t round; uint32_t mode;
  uint32_t debug_stop;              // ZKE_DEV_KNOBS builds only (stage ablation for the profiles): 0 everywhere else
  uint32_t strict;                  // ZKE_STRICT_*: zke_options' strictness flags
  uint64_t now;                     // the time x= is compared with (ZKE_STRICT_EXPIRY_X)
  const KeyCacheEntry* cache;       // per-key Montgomery constants (nullptr: no cache, every signature takes the wave routine)
  uint32_t route_mask;              // bit 0 / 1: the four- / eight-lane RSA kernel is part of this batch's hash / modexp launch
  uint32_t* wave_count;             // job list of the one-signature-per-wave RSA routine: the e-mails not routed to a lane-group kernel
  uint32_t* wave_list;              // (nullptr: no list, the routine looks at every job).  Appended here, consumed by the next launch.
};

#ifdef ZKE_DEV_KNOBS
#define ZKE_DEV_STOP(k) do { if (A.debug_stop == (k)) return; } while (0)
#else
#define ZKE_DEV_STOP(k) do { } while (0)
#endif

// canon.hip.h
__device__ __forceinline__ void canon_body_wave(const BatchDev& B, uint32_t i, uint32_t mode, uint32_t flags, uint32_t boff,
                                                uint32_t blen, uint64_t len_tag, uint8_t* lds, bool ignore_l, bool bucket,
                                                uint32_t hdr_len = 0);
// File e-mail i's message of `len` bytes (kind 0 body, 1 header preimage) under its length class.  One lane per kind: lanes 0
// and 1 file the body and the header preimage with ONE atomic instruction at the very end of the front end — the positions
// come back after a round trip to the memory side that nothing else waits for.
__device__ __forceinline__ void sha_bucket(uint32_t* order, uint32_t kind, uint32_t n_pad, uint32_t i, uint32_t len) {
  const uint32_t cls = sha_len_class((len + 9 + 63) >> 6);
  order[sha_order_key(kind, n_pad) + i] = (cls << 24) | (atomicAdd(order + sha_order_cnt(kind) + cls, 1u) & 0xFFFFFFu);
}


// The front end of e-mail i by the calling wave (L: the wave's LDS image).  parse_kernel runs it for every e-mail of a
// batch (round 0, and mode 1); the verdict launch runs it again, round by round, for the rare e-mail whose candidate
// signature failed while a later same-domain signature is still untried (verdict.hip.h).
template <bool FAST = true, int MODE = 0>
__device__ __forceinline__ void parse_email(const ParseArgs& A, const uint32_t i, ParseLds& L) {
  const BatchDev& B = A.b;
  const int lane = lane_id();
  EmailMeta* M = B.meta + i;
  zke_result* R = B.results + i;
  RsaJob* J = B.rsa + i;
  const uint32_t round = A.round;
  if (round > 0 && MODE == 0) {
    // later signature rounds: retire this e-mail's jobs of the previous round first, so the SHA / RSA
    // launches of this round only touch e-mails that are still pending
    if (lane < 4) { ShaJob z{0, 0, 0, 0}; B.sha[(size_t)lane * B.n_pad + i] = z; }
    if (lane == 0) J->flags = 0;
    if (M->state != ST_PENDING) return;
  }

  const uint64_t r0 = B.raw_off[i], r1 = B.raw_off[i + 1];
  if (round == 0 && MODE == 0) batch_prologue(B, i, lane == 0, i == 0, (uint32_t)lane, 64);
  Str raw = mkstr(B.raw + r0, (uint32_t)(r1 - r0));
  auto stage_head = [&]() {
    // Stage the head of the e-mail in LDS with 16-byte lane-contiguous loads: every later scan of the header
    // block (split, tag lists, canonicalisation) then costs LDS latency instead of a dependent L2 round trip.
    const uint32_t want = raw.len < PARSE_STAGE_BYTES ? raw.len : PARSE_STAGE_BYTES;
    const uint32_t full = want & ~15u;
    for (uint32_t base = 0; base < full; base += 64 * 16) {
      const uint32_t o = base + (uint32_t)lane * 16;
      if (o < full) *(uint4*)(L.stage + o) = *(const uint4_unaligned*)(raw.base + o);
    }
    { const uint32_t o = full + (uint32_t)lane; if (o < want) L.stage[o] = raw.base[o]; }        // want - full < 16
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    raw.lds = L.stage; raw.lds_len = want;
  };
  if (MODE == 0) stage_head();
  const Str dom = mkstr(B.dom + B.dom_off[i], (uint32_t)(B.dom_off[i + 1] - B.dom_off[i]));
  const Str key = mkstr(B.key + B.key_off[i], (uint32_t)(B.key_off[i + 1] - B.key_off[i]));
  uint8_t* regA = B.scratch + scratch_offset(r0 - B.raw_off[0], i);
  const uint32_t capA = raw.len + PRE_SLACK;

  auto sha_job = [&](uint32_t kind, const void* src, uint32_t len, void* dst, uint32_t algo = 0) {
    if (lane == 0 && MODE == 0) {
      ShaJob j; j.src = (uint64_t)src; j.dst = (uint64_t)dst; j.len = len; j.pad = algo;
      B.sha[(size_t)kind * B.n_pad + i] = j;
    }
  };
  auto finish = [&](uint32_t status, uint32_t detail) {
    if (lane == 0) { M->state = ST_FINAL; M->status = status; M->detail = detail; }
  };
  const bool bucket = B.order && round == 0 && MODE == 0;        // (later rounds hash by the e-mail's own wave: verdict.hip.h)

  if (MODE == 1) {
    // canonicalize_signed_email runs only for e-mails whose verify_email succeeded (circuits.rs:32-35)
    for (uint32_t o = lane; o < sizeof(EmailMeta) / 4; o += 64) ((uint32_t*)M)[o] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (R->status != ZKE_OK) { finish(R->status, R->detail); return; }
    const EmailMeta* V = B.meta_verify + i;
    // STRICTNESS SITE canon_takes_verified_signature (oracle: verify_one, same name).  Default: canonicalize_signed_email takes
    // the FIRST DKIM-Signature header, whatever its d=.  ZKE_STRICT_CANON_VERIFIED: the signature verify_dkim accepted.
    if (V->first_sig_hdr == V->cand_hdr || (A.strict & ZKE_STRICT_CANON_VERIFIED)) {        // same signature: nothing to recompute
      if (lane == 0) {
        M->state = ST_CAND; M->reuse = 1; M->flags = V->flags; M->preimage_len = V->preimage_len;
        M->body_off = V->body_off; M->body_len = V->body_len; M->canon_full_len = V->canon_full_len;
        // STRICTNESS SITE canon_ignores_l (also canon_body_wave, mode 1): the whole canonical body instead of its first l= bytes
        M->hashed_len = (A.strict & ZKE_STRICT_CANON_IGNORES_L) ? V->canon_full_len : V->hashed_len;
        M->body_src_is_raw = V->body_src_is_raw;
        M->len_tag_lo = V->len_tag_lo; M->len_tag_hi = V->len_tag_hi;
      }
      return;
    }
  } else if (round == 0) {
    // result record and meta start from zero
    for (uint32_t o = lane; o < sizeof(zke_result) / 4; o += 64) ((uint32_t*)R)[o] = 0;
    for (uint32_t o = lane; o < sizeof(EmailMeta) / 4; o += 64) ((uint32_t*)M)[o] = 0;
    for (uint32_t k = 0; k < 4; k++) sha_job(k, nullptr, 0, nullptr);
    if (bucket && lane < 2) B.order[sha_order_key((uint32_t)lane, B.n_pad) + i] = SHA_KEY_NONE;
    if (lane == 0) { J->flags = 0; J->bits = 0; J->k = 0; J->sig_len = 0; J->e = 0; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    if (lane == 0) R->regex_part = 0xFFFFFFFFu;
    if (r1 - r0 >= (1ull << 31)) { finish(ZKE_UNSUPPORTED, ZKE_D_U_EMAIL_TOO_LARGE); return; }
  }

  if (MODE == 1) stage_head();      // (after the early exits above: an e-mail whose canonicalize pass reuses the verify pass never gets here)
  ZKE_DEV_STOP(1);
  // ---- mailparse::parse_mail (core/src/email.rs:26)
  uint32_t perr;
  uint32_t hdr_end = 0;
  uint32_t* hdr_ovf = (uint32_t*)(regA - HDR_OVF_BYTES);       // in front of region A
  uint32_t nh = 0;
  // (the rare later rounds, inlined into the verdict launch, take the serial split: less code there)
  if (!FAST || !split_headers_lines(L, hdr_ovf, raw, nh, perr, hdr_end)) nh = split_headers(L, hdr_ovf, raw, perr, hdr_end);
  if (nh == NONE) {
    finish(perr == ZKE_D_U_TOO_MANY_HEADERS ? ZKE_UNSUPPORTED : ZKE_PARSE_FAIL, perr);
    return;
  }
  ZKE_DEV_STOP(2);
  const uint32_t body_off = find_body(raw, hdr_end >= 2 ? hdr_end - 2 : 0);
  ZKE_DEV_STOP(3);
  if (lane == 0) {
    if (MODE == 0) { R->n_headers = nh; R->body_offset = body_off; }
    M->n_headers = nh; M->body_off = body_off; M->body_len = raw.len - body_off;
  }
  // lane x: where header field x's name starts and how long it is (fields 0..63; later ones are looked up one by one)
  uint32_t hks_lane = 0, hnl_lane = NONE;
  if ((uint32_t)lane < nh && lane < (int)HDR_LDS_ENTRIES) { hks_lane = L.hdr[4 * lane]; hnl_lane = L.hdr[4 * lane + 1] - hks_lane; }
  if (round == 0 && MODE == 0) {
    // ---- still parse_mail: the MIME subparts (mime.hip.h).  The first Content-Type header of the message decides.
    bool has_ct = false;
    uint32_t cvs = 0, cve = 0;
    for (uint64_t m = __ballot(hnl_lane == 12u); m; m &= m - 1) {
      const uint32_t x = (uint32_t)__builtin_ctzll(m);
      if (span_ieq(raw, __builtin_amdgcn_readlane(hks_lane, x), 12, LIT("content-type"))) { const HdrSpan hs = hdr_get(L, hdr_ovf, x); has_ct = true; cvs = hs.vs; cve = hs.ve; break; }
    }
    for (uint32_t x = HDR_LDS_ENTRIES; !has_ct && x < nh; x++) {
      const HdrSpan hs = hdr_get(L, hdr_ovf, x);
      if (span_ieq(raw, hs.ks, hs.ke - hs.ks, LIT("content-type"))) { has_ct = true; cvs = hs.vs; cve = hs.ve; }
    }
    if (has_ct) {
      uint32_t ixb = hdr_end;                      // behind the empty line that ended the header list
      if (ixb < raw.len) ixb += (uni(ldb(raw, ixb)) == '\r') ? 2u : 1u;
      uint32_t md = 0;
      const uint32_t mr = mime_walk((uint32_t*)L.tagbuf, raw, ixb, true, cvs, cve, md);     // the tag buffer is not in use yet
      if (mr) { finish(mr, md); return; }
    }
  }

  // ---- DkimPublicKey::try_from_bytes (core/src/email.rs:28-29)
  if (round == 0 && MODE == 0) {
    const uint32_t kt = B.key_type[i];
    if (kt == ZKE_KEY_ED25519) {
      // raw 32 bytes (helpers/src/dkim.rs:103-108); whether they are a curve point is decided by ed25519_email_kernel
      if (key.len != 32) { finish(ZKE_KEY_DECODE_FAIL, ZKE_D_KEY_DER); return; }
      if (lane == 0) M->key_ok = 2;
    } else {
      if (kt != ZKE_KEY_RSA) { finish(ZKE_KEY_DECODE_FAIL, ZKE_D_KEY_TYPE); return; }
      uint32_t bits = 0, even = 0, np = 0, nl = 0;
      uint64_t ekey = 0;
      const uint32_t kr = decode_rsa_key(key, J, bits, even, np, nl, ekey);
      if (kr) { finish(ZKE_KEY_DECODE_FAIL, kr); return; }
      uint32_t route = 0x800;           // no cache, no lane-group kernel in this launch, or an exponent / modulus they do not take
      if (A.cache && A.route_mask && ekey == 65537 && !even) route = rsa_route(key, np, nl, bits, A.cache, A.route_mask);
      if (lane == 0) { R->rsa_bits = bits; M->key_ok = 1; M->even_modulus = even; M->rsa_route = route; }
    }
    // the two output witnesses (core/src/circuits.rs:16-17)
    sha_job(2, dom.base, dom.len, R->from_domain_hash);
    sha_job(3, key.base, key.len, R->public_key_hash);
    {
      // cfdkim compares d= and from_domain lower-cased as Unicode strings; d= is ASCII whenever a signature gets that far,
      // so folding ASCII alone is exact unless from_domain holds U+212A KELVIN SIGN (to_lowercase() = "k", the one
      // non-ASCII character with an ASCII lower case): reported, never guessed
      bool kelvin = false;
      for (uint32_t base = 0; base + 2 < dom.len; base += 64) {
        const uint32_t o = base + (uint32_t)lane;
        kelvin = kelvin || (o + 2 < dom.len && ldb(dom, o) == 0xE2 && ldb(dom, o + 1) == 0x84 && ldb(dom, o + 2) == 0xAA);
      }
      if (__ballot(kelvin)) { finish(ZKE_UNSUPPORTED, ZKE_D_U_DOMAIN_FOLD); return; }
    }
  }

  ZKE_DEV_STOP(4);
  // ---- scan the DKIM-Signature headers in file order
  uint32_t sig_ix = 0, cand_count = 0, last_touched = 0, unsupported = 0;
  uint32_t err_all = 0;        // last non-candidate error anywhere
  uint32_t err_after = 0;      // last non-candidate error after this round's candidate
  bool have_cand = false;
  uint32_t first_sig_hdr = NONE;
  uint32_t cand_flags = 0, cand_hdr_len = 0;
  uint64_t cand_len_tag = 0;
  const uint64_t sig_len_mask = __ballot(hnl_lane == 14u);          // only a 14-byte name can be "DKIM-Signature"
  for (uint32_t hx = 0; hx < nh; hx++) {
    if (hx < 64 && !((sig_len_mask >> hx) & 1)) continue;
    const HdrSpan hs = hdr_get(L, hdr_ovf, hx);
    const uint32_t ks = hs.ks, ke = hs.ke, vs = hs.vs, ve = hs.ve;
    if (!span_ieq(raw, ks, ke - ks, LIT("dkim-signature"))) continue;
    const uint32_t this_ix = sig_ix++;
    if (first_sig_hdr == NONE) first_sig_hdr = hx;
    if (MODE == 1 && hx != first_sig_hdr) break;
    const Str v = substr(raw, vs, ve);
    auto note_err = [&](uint32_t e) { err_all = e; if (have_cand) err_after = e; last_touched = this_ix; };
    uint32_t present;
    const uint32_t verr = validate_sig<FAST>(L, v, present, A.strict, A.now);
    ZKE_DEV_STOP(5);
    if (verr == ZKE_D_U_SIG_NON_ASCII || verr == ZKE_D_U_TOO_MANY_TAGS || verr == ZKE_D_U_SIG_TOO_LONG) {
      unsupported = verr; last_touched = this_ix;
      if (MODE == 1) { finish(ZKE_UNSUPPORTED, verr); return; }
      continue;
    }
    if (verr) {
      if (MODE == 1) { finish(ZKE_CANON_FAIL, verr); return; }
      note_err(verr);
      continue;
    }
    if (MODE == 0) {
      // signing_domain.to_lowercase() == from_domain.to_lowercase()
      bool same = tagf(L, TG_D, 3) == dom.len;
      if (same) {
        bool bad = false;
        const uint32_t dofs = tagf(L, TG_D, 2);
        for (uint32_t o = 0; o < dom.len; o += 64) {
          const uint32_t l = o + lane;
          if (l < dom.len) bad |= lower(L.tagbuf[dofs + l]) != lower(ldb(dom, l));
        }
        same = __ballot(bad) == 0;
      }
      if (!same) continue;
    }
    last_touched = this_ix;
    // c=, a=, l=  (parser::parse_canonicalization, parse_hash_algo, compute_body_hash's length parse)
    uint32_t flags = 0;
    if (present & (1u << TG_C)) {
      const TagVal c = tagval(L, TG_C);
      if (c == LIT("relaxed/relaxed")) flags = ZKE_F_HDR_RELAXED | ZKE_F_BODY_RELAXED;
      else if (c == LIT("relaxed/simple") || c == LIT("relaxed")) flags = ZKE_F_HDR_RELAXED;
      else if (c == LIT("simple/simple") || c == LIT("simple")) flags = 0;
      else if (c == LIT("simple/relaxed")) flags = ZKE_F_BODY_RELAXED;
      else {
        if (MODE == 1) { finish(ZKE_CANON_FAIL, ZKE_D_BAD_CANON); return; }
        note_err(ZKE_D_BAD_CANON); continue;
      }
    }
    if (MODE == 0) {
      bool ed_alg = false;
      const TagVal a = tagval(L, TG_A);
      if (a == LIT("rsa-sha256")) {}
      else if (a == LIT("rsa-sha1")) flags |= ZKE_F_SHA1;
      else if (a == LIT("ed25519-sha256")) ed_alg = true;      // RFC 8463: SHA-256 hashes, Ed25519 signature
      else { note_err(ZKE_D_BAD_ALGO); continue; }
      // a= and the key type must name the same scheme
      if (ed_alg != (B.key_type[i] == ZKE_KEY_ED25519)) { unsupported = ZKE_D_U_ALGO_ED25519; continue; }
      if (ed_alg) flags |= ZKE_F_ED25519;
    }
    uint64_t len_tag = 0;
    if (present & (1u << TG_L)) {
      if (!parse_usize_tag(L, TG_L, len_tag)) {
        if (MODE == 1) { finish(ZKE_CANON_FAIL, ZKE_D_BAD_LENGTH); return; }
        note_err(ZKE_D_BAD_LENGTH); continue;
      }
      flags |= ZKE_F_HAS_LENGTH;
    }
    // this signature reaches the hash stage
    const uint32_t my_cand = cand_count++;
    if (my_cand != round) continue;
    have_cand = true; err_after = 0;

    // ---- the b= value: decode, and locate its raw span for removal from the preimage
    const uint32_t b_rs = tagf(L, TG_B, 0), b_re = tagf(L, TG_B, 1);
    if (MODE == 0) {
      uint32_t sig_len = 0;
      const bool b64ok = decode_sig(L, tagf(L, TG_B, 2), tagf(L, TG_B, 3), J, sig_len);
      const uint32_t bh_o = tagf(L, TG_BH, 2), bh_l = tagf(L, TG_BH, 3);
      if (lane == 0) {
        M->sig_b64_ok = b64ok ? 1u : 0u;
        J->sig_len = sig_len;
        M->bh_len = bh_l;
      }
      for (uint32_t o = lane; o < 48; o += 64) M->bh[o] = o < bh_l ? L.tagbuf[bh_o + o] : 0;
    }
    // String::replace removes EVERY occurrence of the raw b= value (leftmost first, non-overlapping).  Normally the
    // tag's own span is the only one and the header is read through a one-excision view; when the value occurs
    // again (only short, hand-made b= values do) the excised header is written out once to region B of the scratch
    // slot — free until the body is canonicalised — and read back from there.
    const uint32_t bl = b_re - b_rs;
    Str sv = v;
    if (bl) {
      const uint32_t fl = bl < 8 ? bl : 8;
      auto next_occurrence = [&](uint32_t from, uint32_t skip_at) -> uint32_t {      // first match at >= from, != skip_at
        for (uint32_t base = from; base + bl <= v.len; base += 64) {
          const uint32_t p = base + lane;
          bool cand = p + bl <= v.len && p != skip_at;
          for (uint32_t t = 0; t < fl && cand; t++) cand = ldb(v, p + t) == ldb(v, b_rs + t);
          uint64_t m = __ballot(cand);
          while (m) {
            const uint32_t q = base + (uint32_t)__builtin_ctzll(m);
            m &= m - 1;
            if (same_bytes(v, q, b_rs, bl)) return q;
          }
        }
        return NONE;
      };
      // STRICTNESS SITE b_removes_own_span_only (oracle: build_preimage, same name).  Default: String::replace — every occurrence
      // of the raw b= value goes.  ZKE_STRICT_B_OWN_SPAN: only the tag's own span is emptied.
      if ((A.strict & ZKE_STRICT_B_OWN_SPAN) || next_occurrence(0, b_rs) == NONE) {
        sv.len = v.len - bl; sv.cut = b_rs; sv.skip = bl;
      } else {
        uint8_t* tmp = regA + (((size_t)raw.len + PRE_SLACK + 15) & ~(size_t)15);    // region B: raw.len + 16 bytes
        uint32_t pos = 0, tn = 0;
        while (pos < v.len) {
          const uint32_t q = next_occurrence(pos, NONE);
          const uint32_t end = (q == NONE) ? v.len : q;
          for (uint32_t base = pos; base < end; base += 64) { const uint32_t l = base + (uint32_t)lane; if (l < end) tmp[tn + (l - pos)] = (uint8_t)ldb(v, l); }
          tn += end - pos;
          if (q == NONE) break;
          pos = q + bl;
        }
        // the wave reads these bytes back with ordinary loads: make them visible past this CU's L1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        sv = mkstr(tmp, tn);
      }
    }
    ZKE_DEV_STOP(6);
    // ---- header-hash preimage (cfdkim hash::compute_headers_hash)
    Out out{regA, 0, capA, false};
    const bool hrel = (flags & ZKE_F_HDR_RELAXED) != 0;
    {
      const uint8_t* h = L.tagbuf + tagf(L, TG_H, 2);
      const uint32_t hl = tagf(L, TG_H, 3);
      // h= is split at every ':' (FWS is already stripped; empty entries count).  The ':' after position `from`, 64 bytes
      // of the value per step, instead of a scalar walk over its bytes.
      // (an h= value of at most 64 bytes — five to ten names — has its ':' in ONE mask: no loop, no further reads)
      const bool one_win = hl <= 64;
      const uint64_t colons = one_win ? __ballot((uint32_t)lane < hl && h[(uint32_t)lane < hl ? lane : 0] == ':') : 0ull;
      auto colon_after = [&](uint32_t from) -> uint32_t {
        if (one_win) { const uint64_t m = colons & bits_from(from); return m ? (uint32_t)__builtin_ctzll(m) : hl; }
        for (uint32_t base = from & ~63u; base < hl; base += 64) {
          const uint32_t l = base + (uint32_t)lane;
          const uint64_t m = __ballot(l < hl && l >= from && h[l] == ':');
          if (m) return base + (uint32_t)__builtin_ctzll(m);
        }
        return hl;
      };
      // the last header field in front of index `start` whose name is name[0, len) (case-insensitive), or NONE:
      // cfdkim walks the header list bottom-up with a cursor per name.  Fields 0..63 are tested by name LENGTH first, all
      // at once (lane x holds field x's), so only fields whose name has the right length are compared.
      auto find_last = [&](const uint8_t* name, uint32_t len, uint32_t start) -> uint32_t {
        for (uint32_t x = start; x-- > 64;) {
          const HdrSpan h2 = hdr_get(L, hdr_ovf, x);
          if (span_ieq(raw, h2.ks, h2.ke - h2.ks, name, len)) return x;
        }
        uint64_t m = __ballot((uint32_t)lane < start && (uint32_t)lane < nh && hnl_lane == len);
        while (m) {
          const uint32_t x = 63u - (uint32_t)__builtin_clzll(m);
          m &= ~(1ull << x);
          if (span_ieq(raw, __builtin_amdgcn_readlane(hks_lane, x), len, name, len)) return x;
        }
        return NONE;
      };
      uint32_t st = 0;
      for (;;) {
        const uint32_t e = colon_after(st);                  // entry [st, e)
        // bottom-up cursor per (lower-cased) name: replay the selection of the earlier entries of h= with the same name
        uint32_t cur = nh;
        for (uint32_t st2 = 0; st2 < st;) {
          const uint32_t e2 = colon_after(st2);              // < st: position st - 1 holds a ':'
          bool same = (e2 - st2) == (e - st);
          if (same) {
            bool bad = false;
            for (uint32_t base = 0; base < e - st; base += 64) { const uint32_t o = base + (uint32_t)lane; if (o < e - st) bad |= lower(h[st2 + o]) != lower(h[st + o]); }
            same = __ballot(bad) == 0;
          }
          if (same) {
            const uint32_t found = find_last(h + st2, e2 - st2, cur);
            cur = (found == NONE) ? 0 : found;
          }
          st2 = e2 + 1;
        }
        const uint32_t found = find_last(h + st, e - st, cur);
        if (found != NONE) {
          const HdrSpan sp = hdr_get(L, hdr_ovf, found);
          emit_header(out, substr(raw, sp.ks, sp.ke), substr(raw, sp.vs, sp.ve), hrel, true);
        }
        if (e >= hl) break;
        st = e + 1;
      }
    }
    // the DKIM-Signature header itself, b= emptied, no CRLF (cfdkim uses its own spelling of the name)
    if (hrel) { emit_lit(out, LIT("dkim-signature:")); emit_relaxed_value(out, sv); }
    else { emit_lit(out, LIT("DKIM-Signature: ")); emit_map(out, sv, 0, sv.len, [](uint32_t c) { return c; }); }
    if (out.overflow) {
      unsupported = ZKE_D_U_PREIMAGE_OVERFLOW;
      if (MODE == 1) { finish(ZKE_UNSUPPORTED, unsupported); return; }
      have_cand = false; cand_count--;
      continue;
    }
    cand_flags = flags; cand_len_tag = len_tag;
    if (lane == 0) {
      M->cand_sig_index = this_ix; M->cand_hdr = hx; M->flags = flags;
      M->len_tag_lo = (uint32_t)len_tag; M->len_tag_hi = (uint32_t)(len_tag >> 32);
      M->preimage_len = out.o;
      if (MODE == 0) { R->flags = flags; R->canon_header_len = out.o; R->sig_index = this_ix; }
    }
    if (MODE == 0) sha_job(1, regA, out.o, R->header_hash, (flags & ZKE_F_SHA1) ? 1u : 0u);
    cand_hdr_len = out.o;
    if (MODE == 1) break;
  }
  if (MODE == 1) {
    if (!have_cand) { finish(ZKE_CANON_FAIL, first_sig_hdr == NONE ? ZKE_D_NO_SIGNATURE : ZKE_D_SIG_SYNTAX); return; }
    if (lane == 0) { M->state = ST_CAND; M->first_sig_hdr = first_sig_hdr; }
    return;
  }
  if (lane == 0) {
    M->n_sigs = sig_ix; M->cand_total = cand_count; M->unsupported = unsupported; M->last_touched_sig = last_touched;
    M->post_err = err_after; M->pre_err = err_all; M->first_sig_hdr = first_sig_hdr;
  }
  if (!have_cand) {
    // nothing (more) to hash: neutral / last error / unsupported (core/src/circuits.rs:13 panics either way)
    if (lane == 0) R->sig_index = last_touched;
    if (unsupported) finish(ZKE_UNSUPPORTED, unsupported);
    else finish(ZKE_DKIM_NOT_PASS, err_all ? err_all : (round == 0 ? ZKE_D_NEUTRAL : M->cand_err));
    return;
  }
  ZKE_DEV_STOP(7);        // ablation: full parse, nothing downstream
  if (lane == 0) {
    M->state = ST_CAND;
    // an Ed25519 candidate leaves the RSA job inactive; ed25519_email_kernel verifies it
    const uint32_t jf = (cand_flags & ZKE_F_ED25519) ? 0u : (RSA_F_ACTIVE | ((cand_flags & ZKE_F_SHA1) ? (uint32_t)RSA_F_SHA1 : 0u) |
                                                                  (A.route_mask ? M->rsa_route & (RSA_F_QUAD | RSA_F_OCT) : 0u));
    J->flags = jf;
    M->em_ok = 0;
    if ((jf & RSA_F_ACTIVE) && !(jf & (RSA_F_QUAD | RSA_F_OCT)) && A.wave_list) A.wave_list[atomicAdd(A.wave_count, 1u)] = i;
  }
  // ---- body canonicalisation of the candidate (cfdkim hash::compute_body_hash), same wave, no launch boundary
  canon_body_wave(B, i, 0, cand_flags, body_off, raw.len - body_off, cand_len_tag, L.stage, false, bucket, cand_hdr_len);   // parsing is over: the staged head is dead
}

#ifndef ZKE_PARSE_PRIO
#define ZKE_PARSE_PRIO 0
#endif
#ifndef ZKE_PARSE_WAVES
#define ZKE_PARSE_WAVES 5        // waves per SIMD the front end is compiled for: 96 registers, no spills (at 6 — 80 registers — it spills 300 B per
                                 // lane since the strictness sites and the length buckets joined it: 28.5 M e-mails/s against 30.1 M; LDS allows 22 waves per CU)
#endif
// ZKE_PARSE_WG_WAVES = W: W e-mails per workgroup, still one per wavefront and nothing shared between them (no barrier,
// an LDS area each).  W only sets the granularity at which the chip hands out LDS and registers to the front end: with
// W = 8 a CU holds two workgroups (2 x 8 x 7.2 KB of LDS), i.e. four 80-register waves per SIMD, and 192 registers per SIMD
// + 44 KB of LDS are always left for the hash / modexp launches of earlier batches (fused.hip.h: 187 registers, 14.6 KB).
#ifndef ZKE_PARSE_WG_WAVES
#define ZKE_PARSE_WG_WAVES 1
#endif
constexpr size_t PARSE_DYN_LDS = ZKE_PARSE_WG_WAVES > 1 ? ZKE_PARSE_WG_WAVES * sizeof(ParseLds) : 0;     // launch argument
#ifdef ZKE_PARSE_NUM_VGPR      // with W > 1 the compiler sees an LDS-bound occupancy and takes more registers; this holds them
#define ZKE_PARSE_VGPR_ATTR __attribute__((amdgpu_num_vgpr(ZKE_PARSE_NUM_VGPR)))
#else
#define ZKE_PARSE_VGPR_ATTR
#endif
__global__ __launch_bounds__(64 * ZKE_PARSE_WG_WAVES, ZKE_PARSE_WAVES) ZKE_PARSE_VGPR_ATTR void parse_kernel(ParseArgs A) {
#if ZKE_PARSE_WG_WAVES > 1
  // dynamic LDS (W * sizeof(ParseLds), given at launch): with the size in sight the compiler takes the kernel's occupancy
  // for LDS-bound at 4 waves per SIMD and spends 109 registers — the point of W is that the front end stays at 80
  extern __shared__ __attribute__((aligned(16))) uint8_t parse_lds_raw[];
  ParseLds* L = reinterpret_cast<ParseLds*>(parse_lds_raw);
#else
  __shared__ ParseLds L[1];
#endif
  const uint32_t wave = ZKE_PARSE_WG_WAVES > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const uint32_t email = blockIdx.x * ZKE_PARSE_WG_WAVES + wave;
  if (email >= A.b.n) return;
  if (ZKE_PARSE_PRIO) __builtin_amdgcn_s_setprio(ZKE_PARSE_PRIO);
  parse_email(A, email, L[wave]);
}

// CSR (blob, off[n+1]) -> ShaJob list with digests packed 32 B apart (building-block entry point)
__global__ void sha_jobs_from_csr_kernel(const uint8_t* blob, const uint64_t* off, uint32_t n, uint8_t* digests, ShaJob* jobs) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ShaJob j;
  j.src = (uint64_t)(blob + off[i]);
  j.dst = (uint64_t)(digests + (size_t)i * 32);
  j.len = (uint32_t)(off[i + 1] - off[i]);
  j.pad = 0;
  jobs[i] = j;
}

}  // namespace zke
