// mime.hip.h — mailparse 0.15.0 parse_mail_recursive: the walk over the MIME subparts of a multipart message, whose only
// effect on the verify path is the Err (panic at core/src/email.rs:26) of a subpart whose header block is malformed.
// Included by parse.hip.h; runs in the front end's wave right after the top-level header split (round 0, mode 0).
//
//   part = headers, then: the first "Content-Type" header (eq_ignore_ascii_case) decides.  get_value() unfolds it
//   (lines(), each trim_start()ed, joined by one SP) and decodes RFC 2047 words; parse_param_content splits at EVERY ';',
//   trims, lower-cases the first token (the mimetype) and the parameter names, strips one pair of double quotes from a
//   value, last duplicate wins.  mimetype.starts_with("multipart/") && a "boundary" parameter && a non-empty body:
//   the body ends at the first LINE that starts with "--" + boundary; after each such line the next part runs from the
//   byte behind the next LF to the next line that starts with the boundary (none: the rest is not a part) and is parsed
//   recursively; "--" right behind a boundary ends the walk.
//
// Decided on the raw bytes, which is exact for ASCII values without encoded words whose boundary value lies on one line;
// the rest — bytes >= 0x80 or "=?" in a value that decides (str::trim and to_lowercase are Unicode-aware, a decoded word
// can hold anything), a boundary value with a line break in it, RFC 2231 boundary forms without a plain one, more than
// 8 nested multiparts — is ZKE_UNSUPPORTED (ZKE_D_U_MIME_*), never a guess.  Restated from recollection of the crate,
// like the header split; oracle/zke_oracle.c (mime_walk) and tests/mime_model.py state the same rules.
#pragma once

namespace zke {

constexpr uint32_t MIME_MAX_DEPTH = 8;

__device__ __forceinline__ bool rust_ws(uint32_t c) { return c == ' ' || (c >= 9 && c <= 13); }     // ASCII members of White_Space

// v[a, b) holds a byte >= 0x80 or "=?"
__device__ __forceinline__ bool mime_undecidable(const Str& v, uint32_t a, uint32_t b) {
  for (uint32_t base = a; base < b; base += 64) {
    const uint32_t l = base + (uint32_t)lane_id();
    bool bad = false;
    if (l < b) {
      const uint32_t c = ldb(v, l);
      bad = c >= 0x80 || (c == '=' && ldb(v, l + 1) == '?');
    }
    if (__ballot(bad)) return true;
  }
  return false;
}
// trim(): [s, e) without the white space at both ends
__device__ __forceinline__ void mime_trim(const Str& v, Win& w, uint32_t& s, uint32_t& e) {
  s = wfind(v, w, s, e, [](uint32_t c) { return !rust_ws(c); });
  const uint32_t last = wrfind(v, w, s, e, [](uint32_t c) { return !rust_ws(c); });
  e = (last == NONE) ? s : last + 1;
}
// parse_content_type(get_value()) as far as the walk needs it.  0: no multipart with a boundary; 1: the boundary value is
// v[bs, be); 0x80000000 | ZKE_D_U_MIME_*: not decided
__device__ __forceinline__ uint32_t mime_content_type(const Str& v, uint32_t& bs, uint32_t& be) {
  const uint32_t n = v.len;
  {
    // The usual case in one look at the first 64 bytes: the first token ends there (';' or the end of the value), holds
    // nothing undecidable and does not start with "multipart/" — a leaf.
    const uint32_t l = (uint32_t)lane_id();
    const uint32_t c = ldb(v, l);
    const uint64_t semi = __ballot(c == ';');
    const uint32_t t0 = semi ? (uint32_t)__builtin_ctzll(semi) : (n <= 64 ? n : 65u);
    if (t0 <= 63) {                                                       // (a token that fills the chunk goes the long way)
      const uint32_t nx = lane_shl1(c);                                   // lane 63 is outside the token
      const bool in = l < t0;
      const uint64_t und = __ballot(in && (c >= 0x80 || (c == '=' && nx == '?')));
      const uint64_t nws = __ballot(in && !rust_ws(c));
      if (!und) {
        if (!nws) return 0;
        const uint32_t s0 = (uint32_t)__builtin_ctzll(nws), e0 = 64u - (uint32_t)__builtin_clzll(nws);
        if (e0 - s0 < 10) return 0;
        if (__ballot(l >= s0 && l < s0 + 10 && lower(c) != lit_at(LIT("multipart/"), l - s0))) return 0;
      }
    }
  }
  Win w; w.wpos = WNONE; w.c = 0;
  const uint32_t t0e = wfind(v, w, 0, n, [](uint32_t c) { return c == ';'; });
  if (mime_undecidable(v, 0, t0e)) return 0x80000000u | ZKE_D_U_MIME_CTYPE;
  uint32_t s = 0, e = t0e;
  mime_trim(v, w, s, e);
  if (e - s < 10 || !span_ieq(v, s, 10, LIT("multipart/"))) return 0;
  if (mime_undecidable(v, 0, n)) return 0x80000000u | ZKE_D_U_MIME_CTYPE;
  bool have = false, starred = false;
  for (uint32_t p = t0e + 1; p <= n;) {
    const uint32_t q = wfind(v, w, p, n, [](uint32_t c) { return c == ';'; });
    const uint32_t eq = wfind(v, w, p, q, [](uint32_t c) { return c == '='; });
    if (eq < q) {
      uint32_t ks = p, ke = eq, vs = eq + 1, ve = q;
      mime_trim(v, w, ks, ke);
      mime_trim(v, w, vs, ve);
      if (ve - vs > 1 && at(v, w, vs) == '"' && at(v, w, ve - 1) == '"') { vs++; ve--; }
      if (ke - ks == 8 && span_ieq(v, ks, 8, LIT("boundary"))) { have = true; bs = vs; be = ve; }
      else if (ke - ks >= 9 && span_ieq(v, ks, 9, LIT("boundary*"))) starred = true;
    }
    if (q >= n) break;
    p = q + 1;
  }
  if (have) {
    if (wfind(v, w, bs, be, [](uint32_t c) { return c == '\n'; }) < be) return 0x80000000u | ZKE_D_U_MIME_BOUNDARY;
    return 1;
  }
  return starred ? (0x80000000u | ZKE_D_U_MIME_BOUNDARY) : 0u;
}

// find_from_u8_line_prefix: the first pos in [from, b - len] where "--" + raw[bs, be) starts a line.  (A part slice starts
// behind an LF, so "the start of the slice" and "behind an LF" are one test everywhere but at offset 0 of the e-mail.)
__device__ __forceinline__ uint32_t find_boundary_line(const Str& raw, uint32_t from, uint32_t b, uint32_t bs, uint32_t be) {
  const uint32_t vl = be - bs, need = 2 + vl;
  if (b < need) return NONE;
  const uint32_t last = b - need;                     // the last position a boundary still fits at
  const uint32_t b0 = vl ? ldb(raw, bs) : 0u;
  for (uint32_t base = from; base <= last; base += 64) {
    const uint32_t pos = base + (uint32_t)lane_id();
    bool cand = false;
    if (pos <= last) {
      const uint32_t c0 = ldb(raw, pos), c1 = ldb(raw, pos + 1);
      const uint32_t cp = pos ? ldb(raw, pos - 1) : (uint32_t)'\n';
      cand = c0 == '-' && c1 == '-' && cp == '\n' && (vl == 0 || ldb(raw, pos + 2) == b0);
    }
    uint64_t m = __ballot(cand);
    while (m) {
      const uint32_t q = base + (uint32_t)__builtin_ctzll(m);
      m &= m - 1;
      if (vl <= 1 || same_bytes(raw, q + 2, bs, vl)) return q;
    }
  }
  return NONE;
}

// The walk.  0, or ZKE_PARSE_FAIL / ZKE_UNSUPPORTED with `detail`.  The top-level part's headers are already split:
// ix_body = the byte behind its empty line, [ct_vs, ct_ve) = the value of its first Content-Type header (has_ct).
// stk: 4 words per open multipart (slice end, boundary value span, where the last boundary line's prefix ended), LDS.
__device__ __forceinline__ uint32_t mime_walk(uint32_t* stk, const Str& raw, uint32_t ix_body, bool has_ct, uint32_t ct_vs, uint32_t ct_ve,
                                              uint32_t& detail) {
  const int lane = lane_id();
  uint32_t depth = 0;
  // a part whose headers are known: open a frame when it is a multipart whose first boundary line exists
  auto enter = [&](uint32_t b, uint32_t ixb, bool ct, uint32_t cvs, uint32_t cve) -> uint32_t {
    if (!ct) return 0;
    uint32_t bs = 0, be = 0;
    const uint32_t m = mime_content_type(substr(raw, cvs, cve), bs, be);
    if (m & 0x80000000u) { detail = m & 0xFFFFu; return ZKE_UNSUPPORTED; }
    if (m == 0 || !(b > ixb)) return 0;
    if (depth >= MIME_MAX_DEPTH) { detail = ZKE_D_U_MIME_DEPTH; return ZKE_UNSUPPORTED; }
    bs += cvs; be += cvs;
    const uint32_t pos = find_boundary_line(raw, ixb, b, bs, be);
    if (pos == NONE) return 0;
    if (lane == 0) { uint32_t* f = stk + 4 * depth; f[0] = b; f[1] = bs; f[2] = be; f[3] = pos + 2 + (be - bs); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    depth++;
    return 0;
  };
  if (const uint32_t r = enter(raw.len, ix_body, has_ct, ct_vs, ct_ve)) return r;
  Win w; w.wpos = WNONE; w.c = 0;
  while (depth) {
    uint32_t* f = stk + 4 * (depth - 1);
    const uint32_t b = uni(f[0]), bs = uni(f[1]), be = uni(f[2]), bend = uni(f[3]);
    if (bend == NONE) { depth--; continue; }                                   // "--" followed the last boundary
    const uint32_t nl = wfind(raw, w, bend, b, [](uint32_t c) { return c == '\n'; });
    if (nl >= b) { depth--; continue; }
    const uint32_t ps = nl + 1;
    const uint32_t pe = find_boundary_line(raw, ps, b, bs, be);
    if (pe == NONE) { depth--; continue; }                                     // no line ends this part: it is not one
    const uint32_t nb = pe + 2 + (be - bs);
    const bool closing = nb + 1 < b && at(raw, w, nb) == '-' && at(raw, w, nb + 1) == '-';
    if (lane == 0) f[3] = closing ? NONE : nb;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // parse_mail_recursive(raw[ps, pe))
    const Str sub = substr(raw, ps, pe);
    uint32_t perr = 0, hdr_end = 0, cvs = 0, cve = 0;
    bool ct = false;
    const uint32_t nh = scan_headers(sub, perr, hdr_end, [&](uint32_t ix, uint32_t key_end, uint32_t vs, uint32_t ve) {
      if (!ct && key_end - ix == 12 && span_ieq(sub, ix, 12, LIT("content-type"))) { ct = true; cvs = vs; cve = ve; }
      return true;
    });
    if (nh == NONE) {
      detail = perr == ZKE_D_HDR_LONE_CR ? ZKE_D_SUBPART_LONE_CR : ZKE_D_SUBPART_LEADING_SPACE;
      return ZKE_PARSE_FAIL;
    }
    // hdr_end is where the list stopped: at the empty line (LF or CRLF) or at the end of the slice
    uint32_t ixb = hdr_end;
    if (ixb < sub.len) ixb += (uni(ldb(sub, ixb)) == '\r') ? 2u : 1u;
    if (const uint32_t r = enter(pe, ps + ixb, ct, ps + cvs, ps + cve)) return r;
  }
  return 0;
}

}  // namespace zke
