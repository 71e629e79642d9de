// front.hip.h — the e-mail front end of verify_dkim, ONE E-MAIL PER LANE (64 e-mails per wavefront).
//
// Replaces, on the verify path (call sites core/src/email.rs:26-33, core/src/circuits.rs:34-35):
//   * mailparse 0.15.0 parse_headers / parse_header (header list; the MIME subpart walk has no effect on
//     this path other than extra parse errors on malformed subparts — not restated);
//   * cfdkim validate_header, the tag-list grammar (RFC 6376 §3.2), d= / i= / h= / q= / v= checks,
//     c= / a= / l= parsing, header selection (§5.4.2, bottom-up) and header canonicalisation
//     (§3.4.1 / §3.4.2) producing the header-hash preimage (§3.7);
//   * rsa 0.9.6 RsaPublicKey::from_pkcs1_der + check_public (RFC 8017 A.1.1);
//   * base64 STANDARD decode of b=.
//
// Why a second front end.  Header parsing is branchy byte-serial work on ~1 KB per e-mail.  The default front
// end (parse.hip.h) runs one e-mail per wavefront with the lanes as a 64-byte-wide scanner: ~45 k
// wave-instructions and ~85 us per e-mail-wave, 1 024 waves per 1 024-e-mail batch.  This variant gives every
// LANE its own e-mail and runs the byte-serial parser 64-wide: ~25x fewer wave-instructions per e-mail, but a
// wave now lives ~1 ms (measured: ~50 VALU per byte step once window reloads, bounds checks and divergence are
// paid), so it only wins when a launch has thousands of e-mails to spread over the chip.  Measured on MI355X at
// 1 024 e-mails per batch, 16 batches in flight: 10.9 M e-mails/s (wavefront front end) vs 7.1 M (this one).
// It is kept selectable (ZKE_LANE_PARSE=1), parity-tested, as the starting point for a word-at-a-time
// (SWAR) scanner.  Each lane reads its e-mail's head from an LDS slab filled wave-cooperatively (16 B per lane,
// coalesced) through an 8-byte register window, keeps its header-span table, tag records and FWS-stripped tag
// values in a private slice of an HBM workspace, and writes the preimage with byte stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "parse.hip.h"

namespace zke {

typedef uint64_t __attribute__((aligned(1))) u64_unaligned;

constexpr uint32_t FRONT_STAGE = 2048;          // bytes of every e-mail's head staged in LDS
constexpr uint32_t FRONT_ROW = FRONT_STAGE + 16;  // row stride: lane l and l+16 share banks (2-way) at most

struct Rd {                      // sequential byte reader with an 8-byte register window
  const uint8_t* p;              // the string in HBM
  const uint8_t* lds;            // LDS copy of p[0 .. staged) (8-byte aligned), or nullptr
  uint32_t staged;
  uint32_t len;
  uint32_t w0;
  uint64_t win;
};
__device__ __forceinline__ void rd_init(Rd& r, const uint8_t* p, uint32_t len, const uint8_t* lds = nullptr, uint32_t staged = 0) {
  r.p = p; r.len = len; r.w0 = 0xFFFFFF00u; r.win = 0; r.lds = lds; r.staged = staged;
}
__device__ __forceinline__ uint32_t rd_get(Rd& r, uint32_t i) {       // requires i < r.len
  uint32_t d = i - r.w0;
  if (d >= 8u) {
    const uint32_t b = i & ~7u;
    if (b + 8u <= r.staged) {
      r.win = *(const uint64_t*)(r.lds + b);                           // ds_read_b64
    } else if (b + 8u <= r.len) {
      r.win = *(const u64_unaligned*)(r.p + b);
    } else {                                                           // never read past the end of the string
      uint64_t v = 0;
      for (uint32_t k = b; k < r.len; k++) v |= (uint64_t)r.p[k] << (8 * (k - b));
      r.win = v;
    }
    r.w0 = b;
    d = i - b;
  }
  return (uint32_t)(r.win >> (8 * d)) & 0xffu;
}
__device__ __forceinline__ uint32_t rd_at(Rd& r, uint32_t i) { return i < r.len ? rd_get(r, i) : OOB; }

struct Wr {                      // byte writer into the preimage region
  uint8_t* p;
  uint32_t o, cap;
  bool ovf;
};
__device__ __forceinline__ void wr_put(Wr& w, uint32_t c) {
  if (w.o < w.cap) w.p[w.o] = (uint8_t)c; else w.ovf = true;
  w.o++;
}

// per-e-mail private workspace (HBM)
struct LaneWs {
  uint32_t hdr[4 * ZKE_MAX_HEADERS];    // key_start, key_end, val_start, val_end
  uint32_t tag[TG_N][4];                // raw_s, raw_e, val_off, val_len (last occurrence wins)
  uint8_t tagbuf[ZKE_MAX_TAGBUF];       // FWS-stripped tag values
};

// ---- mailparse parse_headers / parse_header ---------------------------------------------------------
// returns the header count, or NONE with perr set
__device__ __forceinline__ uint32_t lane_split_headers(Rd& r, uint32_t* ht, uint32_t& perr) {
  const uint32_t len = r.len;
  uint32_t ix = 0, nh = 0;
  perr = 0;
  for (;;) {
    if (ix >= len) break;
    const uint32_t c0 = rd_get(r, ix);
    if (c0 == '\n') break;
    if (c0 == '\r') {
      if (ix + 1 < len && rd_get(r, ix + 1) == '\n') break;
      perr = ZKE_D_HDR_LONE_CR;
      return NONE;
    }
    if (c0 == ' ') { perr = ZKE_D_HDR_LEADING_SPACE; return NONE; }
    uint32_t p = ix, key_end, vs, ve;
    uint32_t c = c0;
    while (p < len && (c = rd_get(r, p)) != ':' && c != '\n') p++;
    if (p >= len) {
      key_end = len; vs = ve = len;
    } else if (c == '\n') {
      key_end = p; vs = ve = p; p = p + 1;
    } else {
      key_end = p;
      p++;
      while (p < len && rd_get(r, p) == ' ') p++;
      vs = ve = p;
      for (;;) {                                  // Value / ValueNewline states
        if (p >= len) break;
        const uint32_t cc = rd_get(r, p);
        if (cc == '\n') {
          const uint32_t nx = (p + 1 < len) ? rd_get(r, p + 1) : OOB;
          if (nx == ' ' || nx == '\t') { p++; continue; }
          p++;
          break;
        }
        if (cc != '\r') ve = p + 1;
        p++;
      }
    }
    if (nh >= ZKE_MAX_HEADERS) { perr = ZKE_D_U_TOO_MANY_HEADERS; return NONE; }
    ht[4 * nh] = ix; ht[4 * nh + 1] = key_end; ht[4 * nh + 2] = vs; ht[4 * nh + 3] = ve;
    nh++;
    ix = p;
  }
  return nh;
}

// cfdkim get_body: everything after the first CRLFCRLF
__device__ __forceinline__ uint32_t lane_find_body(Rd& r) {
  const uint32_t len = r.len;
  uint32_t st = 0;                     // how much of \r\n\r\n has been seen
  for (uint32_t i = 0; i < len; i++) {
    const uint32_t c = rd_get(r, i);
    if (c == '\r') st = (st == 2) ? 3 : 1;
    else if (c == '\n') { if (st == 1) st = 2; else if (st == 3) return i + 1; else st = 0; }
    else st = 0;
  }
  return len;
}

// ---- PKCS#1 RSAPublicKey DER (rsa 0.9.6 from_pkcs1_der + check_public) ----------------------------
__device__ __forceinline__ uint32_t lane_der_len(Rd& k, uint32_t p, uint32_t avail, uint32_t& out) {
  if (avail < 1) return 0;
  const uint32_t b0 = rd_get(k, p);
  if (b0 < 0x80) { out = b0; return 1; }
  const uint32_t nb = b0 & 0x7f;
  if (nb == 0 || nb > 4 || nb + 1 > avail) return 0;
  uint32_t v = 0;
  for (uint32_t i = 0; i < nb; i++) v = (v << 8) | rd_get(k, p + 1 + i);
  if (rd_get(k, p + 1) == 0) return 0;
  if (nb == 1 && v < 0x80) return 0;
  if (nb == 4 && (v >> 31)) return 0;
  out = v;
  return nb + 1;
}
__device__ __forceinline__ uint32_t lane_der_uint(Rd& k, uint32_t p, uint32_t avail, uint32_t& vp, uint32_t& vl) {
  if (avail < 2 || rd_get(k, p) != 0x02) return 0;
  uint32_t l, c = lane_der_len(k, p + 1, avail - 1, l);
  if (!c || l == 0 || (uint64_t)1 + c + l > avail) return 0;
  const uint32_t s = p + 1 + c;
  const uint32_t v0 = rd_get(k, s);
  if (v0 & 0x80) return 0;
  if (l > 1 && v0 == 0 && !(rd_get(k, s + 1) & 0x80)) return 0;
  vp = s; vl = l;
  if (l > 1 && v0 == 0) { vp = s + 1; vl = l - 1; }
  return 1 + c + l;
}
__device__ __forceinline__ uint32_t lane_decode_key(Rd& k, RsaJob* J, uint32_t& bits_out, uint32_t& even) {
  const uint32_t len = k.len;
  if (len < 2 || rd_get(k, 0) != 0x30) return ZKE_D_KEY_DER;
  uint32_t sl, c = lane_der_len(k, 1, len - 1, sl);
  if (!c || (uint64_t)1 + c + sl != len) return ZKE_D_KEY_DER;
  uint32_t p = 1 + c, avail = sl, np, nl, ep, el;
  uint32_t used = lane_der_uint(k, p, avail, np, nl);
  if (!used) return ZKE_D_KEY_DER;
  p += used; avail -= used;
  used = lane_der_uint(k, p, avail, ep, el);
  if (!used || used != avail) return ZKE_D_KEY_DER;
  const uint32_t n0 = rd_get(k, np);
  uint32_t bits = 0;
  if (!(nl == 1 && n0 == 0)) bits = nl * 8 - (uint32_t)(__builtin_clz(n0) - 24);
  if (bits > 4096) return ZKE_D_KEY_RANGE;
  if (el > 8) return ZKE_D_KEY_RANGE;
  uint64_t e = 0;
  for (uint32_t i = 0; i < el; i++) e = (e << 8) | rd_get(k, ep + i);
  if (e < 2 || e > ((1ull << 33) - 1)) return ZKE_D_KEY_RANGE;
  // modulus, big-endian, right-aligned in the 512-byte field, zero fill in front; written as dwords
  uint32_t* md = (uint32_t*)J->mod;
  for (uint32_t wd = 0; wd < 128; wd++) {
    uint32_t v = 0;
    const uint32_t o = 4 * wd;
    if (o + 4 > 512 - nl) {
      for (uint32_t b = 0; b < 4; b++) {
        const uint32_t off = o + b;
        if (off >= 512 - nl) v |= rd_get(k, np + (off - (512 - nl))) << (8 * b);
      }
    }
    md[wd] = v;
  }
  even = !(rd_get(k, np + nl - 1) & 1) && bits != 0;
  J->e = e; J->k = nl; J->bits = bits;
  bits_out = bits;
  return 0;
}

// ---- cfdkim tag-list (parser::tag_list / tag_spec) over the header value raw[vs, ve) ------------------
// returns the position after the spec, or NONE
__device__ __forceinline__ uint32_t lane_tag_spec(Rd& r, LaneWs* W, uint32_t vs, uint32_t ve, uint32_t pos, uint32_t& present,
                                               uint32_t& ntags, uint32_t& tb, uint32_t& err) {
  uint32_t p = pos;
  while (p < ve && is_fws(rd_get(r, p))) p++;
  if (p >= ve || !is_alpha(rd_get(r, p))) return NONE;
  const uint32_t ns = p;
  while (p < ve && is_alnumpunc(rd_get(r, p))) p++;
  const uint32_t ne = p;
  while (p < ve && is_fws(rd_get(r, p))) p++;
  if (p >= ve || rd_get(r, p) != '=') return NONE;
  p++;
  while (p < ve && is_fws(rd_get(r, p))) p++;
  const uint32_t rs = p;
  uint32_t re = p;
  const uint32_t off = tb;
  bool over = false;
  while (p < ve) {
    const uint32_t c = rd_get(r, p);
    const bool vc = is_valchar(c);
    if (!vc && !is_fws(c)) break;
    if (vc) {
      re = p + 1;
      if (tb < ZKE_MAX_TAGBUF) W->tagbuf[tb] = (uint8_t)c; else over = true;
      tb++;
    }
    p++;
  }
  ntags++;
  if (ntags > ZKE_MAX_TAGS) { err = ZKE_D_U_TOO_MANY_TAGS; return p; }
  if (over) { err = ZKE_D_U_SIG_TOO_LONG; return p; }
  int id = -1;
  const uint32_t c0 = rd_get(r, ns);
  if (ne - ns == 1) {
    switch (c0) {
      case 'v': id = TG_V; break; case 'a': id = TG_A; break; case 'b': id = TG_B; break; case 'd': id = TG_D; break;
      case 'h': id = TG_H; break; case 's': id = TG_S; break; case 'i': id = TG_I; break; case 'q': id = TG_Q; break;
      case 'c': id = TG_C; break; case 'l': id = TG_L; break; default: break;
    }
  } else if (ne - ns == 2 && c0 == 'b' && rd_get(r, ns + 1) == 'h') {
    id = TG_BH;
  }
  if (id >= 0) {
    W->tag[id][0] = rs - vs; W->tag[id][1] = re - vs; W->tag[id][2] = off; W->tag[id][3] = tb - off;
    present |= 1u << id;
  }
  return p;
}
__device__ __forceinline__ bool lane_tag_eq(const LaneWs* W, int id, const char* lit, uint32_t n) {
  if (W->tag[id][3] != n) return false;
  const uint8_t* s = W->tagbuf + W->tag[id][2];
  for (uint32_t i = 0; i < n; i++) if (s[i] != (uint8_t)lit[i]) return false;
  return true;
}
// cfdkim validate_header.  0 = valid, else ZKE_D_*
__device__ __forceinline__ uint32_t lane_validate_sig(Rd& r, LaneWs* W, uint32_t vs, uint32_t ve, uint32_t& present) {
  uint32_t ntags = 0, tb = 0, err = 0;
  present = 0;
  uint32_t p = lane_tag_spec(r, W, vs, ve, vs, present, ntags, tb, err);
  if (p == NONE) return ZKE_D_SIG_SYNTAX;
  while (!err && p < ve && rd_get(r, p) == ';') {
    const uint32_t q = lane_tag_spec(r, W, vs, ve, p + 1, present, ntags, tb, err);
    if (q == NONE) break;
    p = q;
  }
  if (err) return err;
  const uint32_t req = (1u << TG_V) | (1u << TG_A) | (1u << TG_B) | (1u << TG_BH) | (1u << TG_D) | (1u << TG_H) | (1u << TG_S);
  if ((present & req) != req) return ZKE_D_MISSING_TAG;
  if (!lane_tag_eq(W, TG_V, "1", 1)) return ZKE_D_INCOMPATIBLE_VERSION;
  if (present & (1u << TG_I)) {               // user.ends_with(signing_domain)
    const uint32_t il = W->tag[TG_I][3], dl = W->tag[TG_D][3];
    if (il < dl) return ZKE_D_DOMAIN_MISMATCH;
    const uint8_t* a = W->tagbuf + W->tag[TG_I][2] + il - dl;
    const uint8_t* b = W->tagbuf + W->tag[TG_D][2];
    for (uint32_t i = 0; i < dl; i++) if (a[i] != b[i]) return ZKE_D_DOMAIN_MISMATCH;
  }
  {                                           // h= must name "from"
    const uint8_t* h = W->tagbuf + W->tag[TG_H][2];
    const uint32_t hl = W->tag[TG_H][3];
    bool found = false;
    uint32_t st = 0;
    for (uint32_t i = 0; i <= hl; i++) {
      if (i == hl || h[i] == ':') {
        if (i - st == 4 && lower(h[st]) == 'f' && lower(h[st + 1]) == 'r' && lower(h[st + 2]) == 'o' && lower(h[st + 3]) == 'm') found = true;
        st = i + 1;
      }
    }
    if (!found) return ZKE_D_FROM_NOT_SIGNED;
  }
  if ((present & (1u << TG_Q)) && !lane_tag_eq(W, TG_Q, "dns/txt", 7)) return ZKE_D_BAD_QUERY_METHOD;
  return 0;
}
__device__ __forceinline__ bool lane_parse_usize(const LaneWs* W, int id, uint64_t& out) {
  const uint8_t* s = W->tagbuf + W->tag[id][2];
  const uint32_t n = W->tag[id][3];
  uint32_t i = 0;
  if (n && s[0] == '+') i = 1;
  if (i >= n) return false;
  uint64_t v = 0;
  for (; i < n; i++) {
    const uint32_t c = s[i];
    if (c < '0' || c > '9') return false;
    const uint64_t d = c - '0';
    if (v > (0xFFFFFFFFFFFFFFFFull - d) / 10) return false;
    v = v * 10 + d;
  }
  out = v;
  return true;
}

// base64 STANDARD decode of a stripped tag value into J->sig (right-aligned).  false = not canonical base64
__device__ __forceinline__ bool lane_decode_sig(const LaneWs* W, uint32_t off, uint32_t n, RsaJob* J, uint32_t& sig_len) {
  sig_len = 0;
  const uint8_t* s = W->tagbuf + off;
  uint32_t pad = 0, total = 0;
  const bool shape_ok = (n % 4) == 0;
  if (shape_ok && n) {
    pad = (s[n - 1] == '=') ? ((s[n - 2] == '=') ? 2u : 1u) : 0u;
    total = 3 * (n / 4) - pad;
  }
  uint32_t* sd = (uint32_t*)J->sig;
  for (uint32_t wd = 0; wd < 128; wd++) sd[wd] = 0;
  if (!shape_ok) return false;
  if (n == 0) return true;
  bool bad = false;
  for (uint32_t q = 0; q < n / 4; q++) {
    const bool last = (q + 1 == n / 4);
    const uint32_t a = b64v(s[4 * q]), b = b64v(s[4 * q + 1]);
    uint32_t c = b64v(s[4 * q + 2]), d = b64v(s[4 * q + 3]);
    uint32_t nb = 3;
    if (last && pad == 2) { c = 0; d = 0; nb = 1; if (b < 64 && (b & 15)) bad = true; }
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
  sig_len = total;
  return !bad;
}

// A logical view of raw[vs, ve) with every occurrence of the raw b= value removed (String::replace).
// Emitted through `f(byte)` in order.
template <class F>
__device__ __forceinline__ void lane_for_each_without_b(Rd& r, Rd& r2, uint32_t vs, uint32_t ve, uint32_t bs, uint32_t bl, F f) {
  uint32_t i = vs;
  while (i < ve) {
    if (bl && i + bl <= ve && rd_get(r, i) == rd_get(r2, bs)) {
      uint32_t k = 1;
      while (k < bl && rd_get(r, i + k) == rd_get(r2, bs + k)) k++;
      if (k == bl) { i += bl; continue; }
    }
    f(rd_get(r, i));
    i++;
  }
}

// cfdkim canonicalize_header_relaxed value part as a streaming filter: unfold, WSP runs -> SP, trim both ends
struct RelaxState { bool pending_sp, any, prev_cr; };
__device__ __forceinline__ void relax_feed(RelaxState& s, Wr& w, uint32_t c, uint32_t nextc) {
  // (c == '\r' && next == '\n') pairs vanish; the '\n' is dropped by prev_cr
  if (s.prev_cr) { s.prev_cr = false; if (c == '\n') return; }
  if (c == '\r' && nextc == '\n') { s.prev_cr = true; return; }
  if (c == ' ' || c == '\t') { s.pending_sp = true; return; }
  if (s.pending_sp && s.any) wr_put(w, ' ');
  s.pending_sp = false; s.any = true;
  wr_put(w, c);
}

__device__ __forceinline__ bool lane_key_ieq_lit(Rd& r, uint32_t ks, uint32_t ke, const uint8_t* name, uint32_t nl) {
  if (ke - ks != nl) return false;
  for (uint32_t i = 0; i < nl; i++) if (lower(rd_get(r, ks + i)) != lower(name[i])) return false;
  return true;
}

// cfdkim canonicalize_header_{relaxed,simple}(key, value) for a header of the e-mail
__device__ __forceinline__ void lane_emit_header(Rd& r, Wr& w, uint32_t ks, uint32_t ke, uint32_t vs, uint32_t ve, bool relaxed) {
  if (relaxed) {
    uint32_t kl = ke;
    while (kl > ks && is_wsp(rd_get(r, kl - 1))) kl--;
    for (uint32_t i = ks; i < kl; i++) wr_put(w, lower(rd_get(r, i)));
    wr_put(w, ':');
    RelaxState st{false, false, false};
    for (uint32_t i = vs; i < ve; i++) relax_feed(st, w, rd_get(r, i), i + 1 < ve ? rd_get(r, i + 1) : OOB);
  } else {
    for (uint32_t i = ks; i < ke; i++) wr_put(w, rd_get(r, i));
    wr_put(w, ':'); wr_put(w, ' ');
    for (uint32_t i = vs; i < ve; i++) wr_put(w, rd_get(r, i));
  }
  wr_put(w, '\r'); wr_put(w, '\n');
}

struct FrontArgs { BatchDev b; LaneWs* ws; uint32_t round; uint32_t mode; uint32_t debug_stop; };

// blockDim = 64; grid = ceil(n / 64).  mode 0: verify_email_with_key scan (round r picks the r-th same-domain
// candidate); mode 1: canonicalize_signed_email (first DKIM-Signature header, no domain filter).
__global__ __launch_bounds__(64) void front_kernel(FrontArgs A) {
  const BatchDev& B = A.b;
  extern __shared__ __attribute__((aligned(16))) uint8_t front_lds[];
  const uint32_t i = blockIdx.x * 64 + threadIdx.x;
  // ---- stage the head of each of the wave's 64 e-mails in LDS: whole wave per row, 16 B per lane, coalesced
  const uint8_t* my_stage = front_lds + (size_t)threadIdx.x * FRONT_ROW;
  uint32_t my_staged = 0;
  {
    const uint32_t base = blockIdx.x * 64;
    const int lane = threadIdx.x;
    for (uint32_t row = 0; row < 64; row++) {
      const uint32_t e = base + row;
      if (e >= B.n) break;
      const uint64_t q0 = B.raw_off[e];
      const uint32_t elen = (uint32_t)(B.raw_off[e + 1] - q0);
      const uint32_t want = elen < FRONT_STAGE ? elen : FRONT_STAGE;
      const uint32_t full = want & ~15u;               // whole 16-byte chunks only; the rest is read from HBM
      const uint8_t* src = B.raw + q0;
      for (uint32_t o = lane * 16; o < full; o += 64 * 16)
        *(uint4*)(front_lds + (size_t)row * FRONT_ROW + o) = *(const uint4_unaligned*)(src + o);
      if ((uint32_t)lane == row) my_staged = full;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (i >= B.n) return;
  if (A.debug_stop == 1) return;
  EmailMeta* M = B.meta + i;
  zke_result* R = B.results + i;
  RsaJob* J = B.rsa + i;
  LaneWs* W = A.ws + i;
  const uint32_t round = A.round;
  const bool verify = A.mode == 0;

  auto sha_job = [&](uint32_t kind, const void* src, uint32_t len, void* dst, uint32_t algo = 0) {
    if (verify) { ShaJob j; j.src = (uint64_t)src; j.dst = (uint64_t)dst; j.len = len; j.pad = algo; B.sha[(size_t)kind * B.n_pad + i] = j; }
  };
  auto finish = [&](uint32_t status, uint32_t detail) { M->state = ST_FINAL; M->status = status; M->detail = detail; };

  if (round > 0 && verify) {
    for (uint32_t k = 0; k < 4; k++) sha_job(k, nullptr, 0, nullptr);
    J->flags = 0;
    if (M->state != ST_PENDING) return;
  }
  const uint64_t r0 = B.raw_off[i], r1 = B.raw_off[i + 1];
  const uint8_t* raw = B.raw + r0;
  const uint32_t raw_len = (uint32_t)(r1 - r0);
  const uint8_t* dom = B.dom + B.dom_off[i];
  const uint32_t dom_len = (uint32_t)(B.dom_off[i + 1] - B.dom_off[i]);
  if (round == 0 && verify) batch_prologue(B, i, true, i == 0, 0, 1);
  uint8_t* regA = B.scratch + scratch_offset(r0 - B.raw_off[0], i);
  const uint32_t capA = raw_len + PRE_SLACK;

  if (!verify) {
    for (uint32_t o = 0; o < sizeof(EmailMeta) / 4; o++) ((uint32_t*)M)[o] = 0;
    if (R->status != ZKE_OK) { finish(R->status, R->detail); return; }       // circuits.rs:32 already panicked
    const EmailMeta* V = B.meta_verify + i;
    if (V->first_sig_hdr == V->cand_hdr) {                                     // same signature: nothing to recompute
      M->state = ST_CAND; M->reuse = 1; M->flags = V->flags; M->preimage_len = V->preimage_len;
      M->body_off = V->body_off; M->body_len = V->body_len; M->canon_full_len = V->canon_full_len;
      M->hashed_len = V->hashed_len; M->body_src_is_raw = V->body_src_is_raw;
      M->len_tag_lo = V->len_tag_lo; M->len_tag_hi = V->len_tag_hi;
      return;
    }
  } else if (round == 0) {
    for (uint32_t o = 0; o < sizeof(zke_result) / 4; o++) ((uint32_t*)R)[o] = 0;
    for (uint32_t o = 0; o < sizeof(EmailMeta) / 4; o++) ((uint32_t*)M)[o] = 0;
    for (uint32_t k = 0; k < 4; k++) sha_job(k, nullptr, 0, nullptr);
    J->flags = 0; J->bits = 0; J->k = 0; J->sig_len = 0; J->e = 0;
    R->regex_part = 0xFFFFFFFFu;
    if (r1 - r0 >= (1ull << 31)) { finish(ZKE_UNSUPPORTED, ZKE_D_U_EMAIL_TOO_LARGE); return; }
  }

  Rd r, r2;
  rd_init(r, raw, raw_len, my_stage, my_staged);
  rd_init(r2, raw, raw_len, my_stage, my_staged);   // second window: comparisons against an earlier span keep their own 8 bytes

  // ---- mailparse::parse_mail (core/src/email.rs:26)
  uint32_t perr;
  const uint32_t nh = lane_split_headers(r, W->hdr, perr);
  if (nh == NONE) { finish(perr == ZKE_D_U_TOO_MANY_HEADERS ? ZKE_UNSUPPORTED : ZKE_PARSE_FAIL, perr); return; }
  if (A.debug_stop == 2) return;
  const uint32_t body_off = lane_find_body(r);
  if (A.debug_stop == 3) return;
  if (verify) { R->n_headers = nh; R->body_offset = body_off; }
  M->n_headers = nh; M->body_off = body_off; M->body_len = raw_len - body_off;

  // ---- DkimPublicKey::try_from_bytes (core/src/email.rs:28-29)
  if (round == 0 && verify) {
    const uint32_t kt = B.key_type[i];
    const uint8_t* key = B.key + B.key_off[i];
    const uint32_t key_len = (uint32_t)(B.key_off[i + 1] - B.key_off[i]);
    if (kt == ZKE_KEY_ED25519) {
      // raw 32 bytes (helpers/src/dkim.rs:103-108); the curve-point check runs in ed25519_email_kernel
      if (key_len != 32) { finish(ZKE_KEY_DECODE_FAIL, ZKE_D_KEY_DER); return; }
      M->key_ok = 2;
    } else {
      if (kt != ZKE_KEY_RSA) { finish(ZKE_KEY_DECODE_FAIL, ZKE_D_KEY_TYPE); return; }
      Rd kr;
      rd_init(kr, key, key_len);
      uint32_t bits = 0, even = 0;
      const uint32_t kerr = lane_decode_key(kr, J, bits, even);
      if (kerr) { finish(ZKE_KEY_DECODE_FAIL, kerr); return; }
      R->rsa_bits = bits; M->key_ok = 1; M->even_modulus = even;
    }
    sha_job(2, dom, dom_len, R->from_domain_hash);                            // circuits.rs:16
    sha_job(3, key, key_len, R->public_key_hash);                             // circuits.rs:17
    for (uint32_t o = 0; o + 2 < dom_len; o++)                                // U+212A KELVIN SIGN lower-cases to ASCII "k": see parse.hip.h
      if (dom[o] == 0xE2 && dom[o + 1] == 0x84 && dom[o + 2] == 0xAA) { finish(ZKE_UNSUPPORTED, ZKE_D_U_DOMAIN_FOLD); return; }
  }

  if (A.debug_stop == 4) return;
  // ---- scan the DKIM-Signature headers in file order
  uint32_t sig_ix = 0, cand_count = 0, last_touched = 0, unsupported = 0, err_all = 0, err_after = 0;
  bool have_cand = false;
  uint32_t first_sig_hdr = NONE;
  for (uint32_t hx = 0; hx < nh; hx++) {
    const uint32_t ks = W->hdr[4 * hx], ke = W->hdr[4 * hx + 1], vs = W->hdr[4 * hx + 2], ve = W->hdr[4 * hx + 3];
    if (!lane_key_ieq_lit(r, ks, ke, DKIM_NAME, 14)) continue;
    const uint32_t this_ix = sig_ix++;
    if (first_sig_hdr == NONE) first_sig_hdr = hx;
    if (!verify && hx != first_sig_hdr) break;
    // from_utf8_lossy would rewrite invalid UTF-8: any byte >= 0x80 is reported, never guessed
    bool non_ascii = false;
    for (uint32_t p = vs; p < ve; p++) if (rd_get(r, p) >= 0x80) { non_ascii = true; break; }
    if (non_ascii) {
      unsupported = ZKE_D_U_SIG_NON_ASCII; last_touched = this_ix;
      if (!verify) { finish(ZKE_UNSUPPORTED, unsupported); return; }
      continue;
    }
    uint32_t present;
    const uint32_t verr = lane_validate_sig(r, W, vs, ve, present);
    if (A.debug_stop == 5) return;
    if (verr == ZKE_D_U_TOO_MANY_TAGS || verr == ZKE_D_U_SIG_TOO_LONG) {
      unsupported = verr; last_touched = this_ix;
      if (!verify) { finish(ZKE_UNSUPPORTED, verr); return; }
      continue;
    }
    if (verr) {
      if (!verify) { finish(ZKE_CANON_FAIL, verr); return; }
      err_all = verr; if (have_cand) err_after = verr; last_touched = this_ix;
      continue;
    }
    if (verify) {                      // signing_domain.to_lowercase() == from_domain.to_lowercase()
      bool same = W->tag[TG_D][3] == dom_len;
      const uint8_t* d = W->tagbuf + W->tag[TG_D][2];
      for (uint32_t k = 0; same && k < dom_len; k++) same = lower(d[k]) == lower(dom[k]);
      if (!same) continue;
    }
    last_touched = this_ix;
    uint32_t flags = 0;
    if (present & (1u << TG_C)) {
      if (lane_tag_eq(W, TG_C, "simple/simple", 13) || lane_tag_eq(W, TG_C, "simple", 6)) flags = 0;
      else if (lane_tag_eq(W, TG_C, "relaxed/simple", 14) || lane_tag_eq(W, TG_C, "relaxed", 7)) flags = ZKE_F_HDR_RELAXED;
      else if (lane_tag_eq(W, TG_C, "simple/relaxed", 14)) flags = ZKE_F_BODY_RELAXED;
      else if (lane_tag_eq(W, TG_C, "relaxed/relaxed", 15)) flags = ZKE_F_HDR_RELAXED | ZKE_F_BODY_RELAXED;
      else {
        if (!verify) { finish(ZKE_CANON_FAIL, ZKE_D_BAD_CANON); return; }
        err_all = ZKE_D_BAD_CANON; if (have_cand) err_after = err_all;
        continue;
      }
    }
    if (verify) {
      bool ed_alg = false;
      if (lane_tag_eq(W, TG_A, "rsa-sha256", 10)) {}
      else if (lane_tag_eq(W, TG_A, "rsa-sha1", 8)) flags |= ZKE_F_SHA1;
      else if (lane_tag_eq(W, TG_A, "ed25519-sha256", 14)) ed_alg = true;
      else { err_all = ZKE_D_BAD_ALGO; if (have_cand) err_after = err_all; continue; }
      if (ed_alg != (B.key_type[i] == ZKE_KEY_ED25519)) { unsupported = ZKE_D_U_ALGO_ED25519; continue; }
      if (ed_alg) flags |= ZKE_F_ED25519;
    }
    uint64_t len_tag = 0;
    if (present & (1u << TG_L)) {
      if (!lane_parse_usize(W, TG_L, len_tag)) {
        if (!verify) { finish(ZKE_CANON_FAIL, ZKE_D_BAD_LENGTH); return; }
        err_all = ZKE_D_BAD_LENGTH; if (have_cand) err_after = err_all;
        continue;
      }
      flags |= ZKE_F_HAS_LENGTH;
    }
    const uint32_t my_cand = cand_count++;
    if (my_cand != round) continue;
    have_cand = true; err_after = 0;

    if (verify) {
      uint32_t sig_len = 0;
      const bool b64ok = lane_decode_sig(W, W->tag[TG_B][2], W->tag[TG_B][3], J, sig_len);
      M->sig_b64_ok = b64ok ? 1u : 0u;
      J->sig_len = sig_len;
      const uint32_t bhl = W->tag[TG_BH][3];
      M->bh_len = bhl;
      for (uint32_t o = 0; o < 48; o++) M->bh[o] = o < bhl ? W->tagbuf[W->tag[TG_BH][2] + o] : 0;
    }
    if (A.debug_stop == 6) return;
    // ---- header-hash preimage (cfdkim hash::compute_headers_hash)
    Wr w{regA, 0, capA, false};
    const bool hrel = (flags & ZKE_F_HDR_RELAXED) != 0;
    {
      const uint8_t* h = W->tagbuf + W->tag[TG_H][2];
      const uint32_t hl = W->tag[TG_H][3];
      uint32_t st = 0;
      for (uint32_t e = 0; e <= hl; e++) {
        if (e != hl && h[e] != ':') continue;
        // bottom-up cursor per (lower-cased) name: replay the earlier entries of h= with the same name
        uint32_t cur = nh, st2 = 0;
        for (uint32_t e2 = 0; e2 < st; e2++) {
          if (h[e2] != ':') continue;
          bool same = (e2 - st2) == (e - st);
          for (uint32_t k = 0; same && k < e - st; k++) same = lower(h[st2 + k]) == lower(h[st + k]);
          if (same) {
            uint32_t found = NONE;
            for (uint32_t x = cur; x-- > 0;)
              if (lane_key_ieq_lit(r, W->hdr[4 * x], W->hdr[4 * x + 1], h + st2, e2 - st2)) { found = x; break; }
            cur = (found == NONE) ? 0 : found;
          }
          st2 = e2 + 1;
        }
        uint32_t found = NONE;
        for (uint32_t x = cur; x-- > 0;)
          if (lane_key_ieq_lit(r, W->hdr[4 * x], W->hdr[4 * x + 1], h + st, e - st)) { found = x; break; }
        if (found != NONE) {
          const uint32_t* sp = W->hdr + 4 * found;
          lane_emit_header(r, w, sp[0], sp[1], sp[2], sp[3], hrel);
        }
        st = e + 1;
      }
    }
    {   // the DKIM-Signature header itself, every occurrence of the raw b= value removed, no trailing CRLF
      const uint32_t bs = vs + W->tag[TG_B][0], bl = W->tag[TG_B][1] - W->tag[TG_B][0];
      if (hrel) {
        const char* nm = "dkim-signature:";
        for (uint32_t k = 0; k < 15; k++) wr_put(w, (uint8_t)nm[k]);
        RelaxState st{false, false, false};
        // one byte of lookahead for the CRLF test: feed with a one-byte delay
        uint32_t held = OOB;
        lane_for_each_without_b(r, r2, vs, ve, bs, bl, [&](uint32_t c) {
          if (held != OOB) relax_feed(st, w, held, c);
          held = c;
        });
        if (held != OOB) relax_feed(st, w, held, OOB);
      } else {
        const char* nm = "DKIM-Signature: ";
        for (uint32_t k = 0; k < 16; k++) wr_put(w, (uint8_t)nm[k]);
        lane_for_each_without_b(r, r2, vs, ve, bs, bl, [&](uint32_t c) { wr_put(w, c); });
      }
    }
    if (w.ovf) {
      unsupported = ZKE_D_U_PREIMAGE_OVERFLOW;
      if (!verify) { finish(ZKE_UNSUPPORTED, unsupported); return; }
      have_cand = false; cand_count--;
      continue;
    }
    M->cand_sig_index = this_ix; M->cand_hdr = hx; M->flags = flags;
    M->len_tag_lo = (uint32_t)len_tag; M->len_tag_hi = (uint32_t)(len_tag >> 32);
    M->preimage_len = w.o;
    if (verify) { R->flags = flags; R->canon_header_len = w.o; R->sig_index = this_ix; sha_job(1, regA, w.o, R->header_hash, (flags & ZKE_F_SHA1) ? 1u : 0u); }
    if (!verify) break;
  }
  if (!verify) {
    if (!have_cand) { finish(ZKE_CANON_FAIL, first_sig_hdr == NONE ? ZKE_D_NO_SIGNATURE : ZKE_D_SIG_SYNTAX); return; }
    M->state = ST_CAND; M->first_sig_hdr = first_sig_hdr;
    return;
  }
  M->n_sigs = sig_ix; M->cand_total = cand_count; M->unsupported = unsupported; M->last_touched_sig = last_touched;
  M->post_err = err_after; M->pre_err = err_all; M->first_sig_hdr = first_sig_hdr;
  if (!have_cand) {
    R->sig_index = last_touched;
    if (unsupported) finish(ZKE_UNSUPPORTED, unsupported);
    else finish(ZKE_DKIM_NOT_PASS, err_all ? err_all : (round == 0 ? ZKE_D_NEUTRAL : M->cand_err));
    return;
  }
  M->state = ST_CAND;
  J->flags = (M->flags & ZKE_F_ED25519) ? 0u : (RSA_F_ACTIVE | ((M->flags & ZKE_F_SHA1) ? (uint32_t)RSA_F_SHA1 : 0u));
}

}  // namespace zke
