// wire.hip.h — the byte streams the zkVM hosts already produce for `Email` / `EmailWithRegex` (the
// derives at core/src/structs.rs:1-6), read in place and handed to the single-e-mail entry points: a host that holds the
// serialised input (what SP1Stdin::write / a risc0 env write emitted) does not have to rebuild the structs to use the engine.
//
//   borsh (cargo feature `risc0`, structs.rs:5): little-endian; String / Vec<T> = u32 length + items; Option<T> = u8 tag
//     (0 / 1) + value; usize as u64; struct fields in declaration order.
//   bincode 1.x default options over serde (feature `sp1`, structs.rs:6): the same with u64 lengths.
//
// Neither crate is vendored in the reference tree; the layouts are the formats' published specifications (pinned by hand-laid
// byte strings in tests/test_wire.py).  Pure host code: no GPU, no allocation beyond the part tables.  Included by engine.hip.
#pragma once

struct zke_wire_doc {
  zke_wire_email view{};
  std::vector<zke_regex_part> parts[2];                    // header, body
  std::vector<const uint8_t*> cap_ptrs;                    // capture tables of all parts, back to back
  std::vector<size_t> cap_lens;
  struct Ext { const uint8_t* name; size_t name_len; const uint8_t* value; size_t value_len; uint32_t is_null; uint64_t max_length; };
  std::vector<Ext> ext;
};

namespace {

struct WireReader {
  const uint8_t* b; size_t n, o = 0; bool wide;            // wide: u64 lengths (bincode)
  const char* err = nullptr;
  bool fail(const char* what) { if (!err) err = what; return false; }
  bool len(size_t& v) {
    const size_t w = wide ? 8 : 4;
    if (n - o < w) return fail("truncated length");
    uint64_t x = 0;
    for (size_t k = 0; k < w; k++) x |= (uint64_t)b[o + k] << (8 * k);
    o += w;
    if (x > n) return fail("implausible length");          // (no item is smaller than a byte)
    v = (size_t)x;
    return true;
  }
  bool bytes(const uint8_t*& p, size_t& l) {
    if (!len(l)) return false;
    if (n - o < l) return fail("truncated bytes");
    p = b + o; o += l;
    return true;
  }
  bool str(const uint8_t*& p, size_t& l) {                 // String: the bytes must be UTF-8 (borsh and serde both check)
    if (!bytes(p, l)) return false;
    for (size_t i = 0; i < l;) {
      const uint32_t c = p[i];
      uint32_t need, lo;
      if (c < 0x80) { i++; continue; }
      if (c >= 0xC2 && c <= 0xDF) { need = 1; lo = 0x80; }
      else if (c >= 0xE0 && c <= 0xEF) { need = 2; lo = 0x800; }
      else if (c >= 0xF0 && c <= 0xF4) { need = 3; lo = 0x10000; }
      else return fail("invalid UTF-8 in String");
      if (l - i <= need) return fail("invalid UTF-8 in String");
      uint32_t cp = c & (0x3Fu >> need);
      for (uint32_t k = 1; k <= need; k++) {
        if ((p[i + k] & 0xC0) != 0x80) return fail("invalid UTF-8 in String");
        cp = (cp << 6) | (p[i + k] & 0x3F);
      }
      if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return fail("invalid UTF-8 in String");
      i += need + 1;
    }
    return true;
  }
  bool tag(bool& some) {
    if (o >= n) return fail("truncated Option tag");
    const uint8_t t = b[o++];
    if (t > 1) return fail("bad Option tag");
    some = t == 1;
    return true;
  }
  bool u64(uint64_t& v) {
    if (n - o < 8) return fail("truncated usize");
    v = 0;
    for (int k = 0; k < 8; k++) v |= (uint64_t)b[o + k] << (8 * k);
    o += 8;
    return true;
  }
};

// Option<Vec<CompiledRegex>> (structs.rs:24-35)
bool wire_parts(WireReader& r, zke_wire_doc& d, int side, uint32_t& has) {
  bool some = false;
  if (!r.tag(some)) return false;
  has = some ? 1u : 0u;
  if (!some) return true;
  size_t np = 0;
  if (!r.len(np)) return false;
  for (size_t k = 0; k < np; k++) {
    zke_regex_part p{};
    if (!r.bytes(p.fwd, p.fwd_len) || !r.bytes(p.bwd, p.bwd_len)) return false;        // DFA { fwd, bwd }  structs.rs:16-19
    bool caps = false;
    if (!r.tag(caps)) return false;
    // captures: None -> the containment check is skipped (core/src/regex.rs:41): 0 captures and a null table
    p.n_captures = 0;
    p.captures = reinterpret_cast<const uint8_t* const*>((uintptr_t)d.cap_ptrs.size());   // index for now: the vectors may still grow
    if (caps) {
      size_t nc = 0;
      if (!r.len(nc)) return false;
      for (size_t c = 0; c < nc; c++) {
        const uint8_t* s; size_t sl;
        if (!r.str(s, sl)) return false;
        d.cap_ptrs.push_back(s); d.cap_lens.push_back(sl);
      }
      p.n_captures = (uint32_t)nc;
    }
    d.parts[side].push_back(p);
  }
  return true;
}

}  // namespace

extern "C" {

int zke_wire_decode(uint32_t format, const uint8_t* bytes, size_t len, uint32_t with_regex, zke_wire_doc** out, size_t* consumed) {
  if (!out || (len && !bytes) || format > ZKE_WIRE_BINCODE) return ZKE_E_ARG;
  *out = nullptr;
  zke_wire_doc* d = new zke_wire_doc();
  WireReader r{bytes, len, 0, format == ZKE_WIRE_BINCODE};
  zke_wire_email& v = d->view;
  const uint8_t *dom = nullptr, *kt = nullptr;
  size_t ktl = 0, next = 0;
  bool ok = r.str(dom, v.domain_len) && r.bytes(v.raw, v.raw_len) && r.bytes(v.key, v.key_len) && r.str(kt, ktl) && r.len(next);
  v.from_domain = reinterpret_cast<const char*>(dom);
  for (size_t k = 0; ok && k < next; k++) {                // Vec<ExternalInput>  structs.rs:40-44
    zke_wire_doc::Ext x{};
    bool some = false;
    ok = r.str(x.name, x.name_len) && r.tag(some);
    if (ok && some) ok = r.str(x.value, x.value_len);
    x.is_null = some ? 0u : 1u;
    ok = ok && r.u64(x.max_length);
    if (ok) d->ext.push_back(x);
    if (ok && !some) v.external_input_null = 1;            // circuits.rs:24 would panic
  }
  if (ok) {
    v.n_external_inputs = (uint32_t)d->ext.size();
    v.key_type = (ktl == 3 && !memcmp(kt, "rsa", 3)) ? ZKE_KEY_RSA : (ktl == 7 && !memcmp(kt, "ed25519", 7)) ? ZKE_KEY_ED25519 : ZKE_KEY_OTHER;
    if (with_regex) ok = wire_parts(r, *d, 0, v.has_header_parts) && wire_parts(r, *d, 1, v.has_body_parts);
  }
  if (!ok) {
    g_err = std::string("zke_wire_decode: ") + (r.err ? r.err : "malformed stream");
    delete d;
    return ZKE_E_ARG;
  }
  for (int side = 0; side < 2; side++)                     // the capture tables are final now: indices -> pointers
    for (zke_regex_part& p : d->parts[side]) {
      const size_t at = (size_t)(uintptr_t)p.captures;
      p.captures = p.n_captures ? d->cap_ptrs.data() + at : nullptr;
      p.capture_lens = p.n_captures ? d->cap_lens.data() + at : nullptr;
    }
  v.header_parts = d->parts[0].empty() ? nullptr : d->parts[0].data(); v.n_header_parts = (uint32_t)d->parts[0].size();
  v.body_parts = d->parts[1].empty() ? nullptr : d->parts[1].data(); v.n_body_parts = (uint32_t)d->parts[1].size();
  if (consumed) *consumed = r.o;
  *out = d;
  return 0;
}

void zke_wire_free(zke_wire_doc* d) { delete d; }

int zke_wire_view(const zke_wire_doc* d, zke_wire_email* out) {
  if (!d || !out) return ZKE_E_ARG;
  *out = d->view;
  return 0;
}

int zke_wire_external_input(const zke_wire_doc* d, uint32_t i, const uint8_t** name, size_t* name_len, const uint8_t** value,
                            size_t* value_len, uint32_t* is_null) {
  if (!d || i >= d->ext.size() || !name || !name_len || !value || !value_len || !is_null) return ZKE_E_ARG;
  const zke_wire_doc::Ext& x = d->ext[i];
  *name = x.name; *name_len = x.name_len; *value = x.value; *value_len = x.value_len; *is_null = x.is_null;
  return 0;
}

int zke_verify_wire(zke_engine* e, uint32_t format, const uint8_t* bytes, size_t len, uint32_t with_regex, zke_result* out) {
  if (!e || !out) return ZKE_E_ARG;
  zke_wire_doc* d = nullptr;
  size_t used = 0;
  if (int r = zke_wire_decode(format, bytes, len, with_regex, &d, &used)) return r;
  int r = 0;
  if (used != len) r = fail(e, ZKE_E_ARG, "zke_verify_wire: trailing bytes behind the record");
  const zke_wire_email& v = d->view;
  if (!r) {
    r = with_regex ? zke_verify_email_with_regex(e, v.raw, v.raw_len, v.from_domain, v.domain_len, v.key, v.key_len, v.key_type,
                                                 v.external_input_null, v.header_parts, v.n_header_parts, v.body_parts, v.n_body_parts, out)
                   : zke_verify_email(e, v.raw, v.raw_len, v.from_domain, v.domain_len, v.key, v.key_len, v.key_type, v.external_input_null, out);
  }
  zke_wire_free(d);
  return r;
}

}  // extern "C"
