// dfa_registry.hip.h — host side of zke_dfa_register: the regex-automata 0.4.9 dense-DFA wire format (little-endian,
// version 2) restated as dense::DFA::from_bytes reads it, and the engine's registry of parsed pairs.
// Replaces the per-e-mail from_bytes of core/src/regex.rs:32-33.  Included by engine.hip (single translation unit).
#pragma once

namespace {

uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

struct HostDfa {
  DfaDev d{};
  std::vector<uint32_t> table;
  uint32_t idle = 0xFFFFFFFFu;     // see dfa_idle_state
};

// The state an unanchored search idles in between matches: among the Start::Text state and the states most of its
// bytes lead to (two hops), the ordinary state (not dead / quit / match) with the most self-loops, if more than half
// of the byte values stay in it.  Only a hint for dfa_wave_kernel's chunk map: any answer is correct, a good one is fast.
uint32_t dfa_idle_state(const HostDfa& h) {
  const DfaDev& d = h.d;
  if (d.start_kind == 2 || h.table.empty()) return 0xFFFFFFFFu;
  auto ordinary = [&](uint32_t s) { return s != 0 && s != d.quit_id && !(d.min_match && d.min_match <= s && s <= d.max_match); };
  auto target = [&](uint32_t s, uint32_t byte) { return h.table[s + d.classes[byte]]; };
  auto majority = [&](uint32_t s) {
    uint32_t best = s, bestn = 0;
    for (uint32_t x = 0; x < 256; x++) {
      const uint32_t t = target(s, x);
      uint32_t cnt = 0;
      for (uint32_t y = 0; y < 256; y++) cnt += target(s, y) == t;
      if (cnt > bestn) { bestn = cnt; best = t; }
    }
    return best;
  };
  uint32_t cand[3];
  cand[0] = d.starts[2]; cand[1] = majority(cand[0]); cand[2] = majority(cand[1]);
  uint32_t idle = 0xFFFFFFFFu, bestn = 127;
  for (uint32_t c : cand) {
    if (!ordinary(c)) continue;
    uint32_t loops = 0;
    for (uint32_t y = 0; y < 256; y++) loops += target(c, y) == c;
    if (loops > bestn) { bestn = loops; idle = c; }
  }
  return idle;
}

// dense::DFA::from_bytes restated: structure, sizes and the id validity checks.  Returns 0 when the blob deserialises,
// else the section at which from_bytes gives up (ZKE_D_DFA_LABEL .. ZKE_D_DFA_QUITSET) — reported as the `detail` of
// ZKE_DFA_DECODE_FAIL, so the first real blob that does not load says which part of the recalled layout is wrong
// (the unanchored start block, the accelerators and the quit set are not pinned by a regex-automata-written blob: DESIGN.md §4).
uint32_t parse_dfa_blob(const uint8_t* b, size_t n, HostDfa& h) {
  static const char LABEL[] = "rust-regex-automata-dfa-dense";
  DfaDev& d = h.d;
  size_t p = 0;
  while (p < n && p < 7 && b[p] == 0) p++;
  auto need = [&](size_t k) { return n - p >= k; };
  if (!need(32) || memcmp(b + p, LABEL, 29) || b[p + 29] != 0) return ZKE_D_DFA_LABEL;
  p += 32;
  if (!need(4) || rd32(b + p) != 0xFEFF) return ZKE_D_DFA_ENDIAN_VERSION; p += 4;
  if (!need(4) || rd32(b + p) != 2) return ZKE_D_DFA_ENDIAN_VERSION; p += 4;
  if (!need(4)) return ZKE_D_DFA_ENDIAN_VERSION; p += 4;
  // Flags::from_bytes: ONE u32 bit set — bit 0 has_empty, bit 1 is_utf8, bit 2 is_always_start_anchored — as the blobs
  // regex-automata itself wrote show (tests/golden/regex_automata_*.dfa; SURVEY Appendix A.3 recalled three u32s)
  if (!need(4)) return ZKE_D_DFA_FLAGS;
  { const uint32_t fl = rd32(b + p); d.has_empty = fl & 1u; d.is_utf8 = (fl >> 1) & 1u; d.always_anchored = (fl >> 2) & 1u; }
  p += 4;
  if (!need(8 + 256)) return ZKE_D_DFA_TRANSITIONS;
  d.state_len = rd32(b + p); d.stride2 = rd32(b + p + 4); p += 8;
  memcpy(d.classes, b + p, 256); p += 256;
  if (d.stride2 < 1 || d.stride2 > 9) return ZKE_D_DFA_TRANSITIONS;
  d.alphabet_len = (uint32_t)d.classes[255] + 2;
  if (d.alphabet_len > (1u << d.stride2)) return ZKE_D_DFA_TRANSITIONS;
  for (int i = 0; i < 256; i++)          // ByteClasses::from_bytes: no class beyond the alphabet — the walk indexes table[sid + class]
    if (d.classes[i] >= d.alphabet_len) return ZKE_D_DFA_TRANSITIONS;   // and only columns < alphabet_len are id-checked below
  if (d.state_len > (1u << 26)) return ZKE_D_DFA_TRANSITIONS;
  const size_t tl = (size_t)d.state_len << d.stride2;
  if (!need(tl * 4)) return ZKE_D_DFA_TRANSITIONS;
  d.table_len = (uint32_t)tl;
  h.table.resize(tl);
  for (size_t i = 0; i < tl; i++) h.table[i] = rd32(b + p + 4 * i);
  p += tl * 4;
  const uint32_t stride = 1u << d.stride2;
  for (size_t s = 0; s < d.state_len; s++)
    for (uint32_t c = 0; c < d.alphabet_len; c++) {
      const uint32_t id = h.table[(s << d.stride2) + c];
      if (id >= tl || (id & (stride - 1))) return ZKE_D_DFA_TRANSITIONS;
    }
  if (!need(4 + 256 + 16)) return ZKE_D_DFA_START_TABLE;
  d.start_kind = rd32(b + p); p += 4;
  if (d.start_kind > 2) return ZKE_D_DFA_START_TABLE;
  memcpy(d.start_map, b + p, 256); p += 256;
  for (int i = 0; i < 256; i++) if (d.start_map[i] >= 6) return ZKE_D_DFA_START_TABLE;
  if (rd32(b + p) != 6) return ZKE_D_DFA_START_TABLE; p += 4;
  const uint32_t spl = rd32(b + p); p += 4;
  p += 8;
  const size_t npat = spl == 0xFFFFFFFFu ? 0 : spl;
  if (npat > (1u << 20)) return ZKE_D_DFA_START_TABLE;
  const size_t sl = 12 + 6 * npat;
  if (!need(sl * 4)) return ZKE_D_DFA_START_TABLE;
  for (size_t i = 0; i < sl; i++) {
    const uint32_t v = rd32(b + p + 4 * i);
    if (v >= tl || (v & (stride - 1))) return ZKE_D_DFA_START_TABLE;
    if (i < 12) d.starts[i] = v;
  }
  p += sl * 4;
  if (!need(4)) return ZKE_D_DFA_MATCH_STATES;
  const uint32_t ms_len = rd32(b + p); p += 4;
  if (ms_len > d.state_len) return ZKE_D_DFA_MATCH_STATES;
  if (!need((size_t)ms_len * 8 + 8)) return ZKE_D_DFA_MATCH_STATES;
  p += (size_t)ms_len * 8;
  p += 4;
  const uint32_t idlen = rd32(b + p); p += 4;
  if (idlen > (1u << 24) || !need((size_t)idlen * 4)) return ZKE_D_DFA_MATCH_STATES;
  p += (size_t)idlen * 4;
  if (!need(32)) return ZKE_D_DFA_SPECIAL;
  d.sp_max = rd32(b + p); d.quit_id = rd32(b + p + 4); d.min_match = rd32(b + p + 8); d.max_match = rd32(b + p + 12);
  const uint32_t min_accel = rd32(b + p + 16), max_accel = rd32(b + p + 20), min_start = rd32(b + p + 24), max_start = rd32(b + p + 28);
  p += 32;
  if (d.min_match > d.max_match || min_accel > max_accel || min_start > max_start) return ZKE_D_DFA_SPECIAL;
  if ((d.min_match == 0) != (d.max_match == 0)) return ZKE_D_DFA_SPECIAL;
  if (d.max_match > d.sp_max || max_accel > d.sp_max || max_start > d.sp_max) return ZKE_D_DFA_SPECIAL;
  if (tl && d.sp_max >= tl) return ZKE_D_DFA_SPECIAL;
  {
    const uint32_t nm = d.max_match ? ((d.max_match - d.min_match) >> d.stride2) + 1 : 0;
    if (nm != ms_len) return ZKE_D_DFA_MATCH_STATES;
  }
  if (!need(4)) return ZKE_D_DFA_ACCELS;
  const uint32_t acc = rd32(b + p); p += 4;
  if (acc > d.state_len || !need((size_t)acc * 8)) return ZKE_D_DFA_ACCELS;
  p += (size_t)acc * 8;
  if (!need(32)) return ZKE_D_DFA_QUITSET;
  memcpy(d.quitset, b + p, 32);
  d.quitset_nonempty = 0;
  for (int i = 0; i < 32; i++) if (d.quitset[i]) d.quitset_nonempty = 1;
  d.wide = tl > 65536 ? 1u : 0u;
  d.valid = 1;
  return 0;
}

// One registered pair.  Heap objects: the registry's vector holds pointers, so an entry does not move while batches use it;
// it is freed only by zke_dfa_unregister / an eviction, both of which first wait for the engine to drain.
struct RegisteredDfa {
  bool valid = false;       // both blobs deserialise (dense::DFA::from_bytes would succeed)
  uint32_t detail = 0;      // ZKE_D_DFA_* of the blob that does not (reverse blob: + ZKE_D_DFA_BWD_OFFSET)
  size_t lds_bytes = 0;     // repacked fwd + rev tables
  uint32_t idle = 0xFFFFFFFFu;   // forward automaton's idle state (dfa_idle_state)
  DevBuf blob;              // the repacked tables
  DevBuf dev;               // RegexDev image
  std::vector<uint8_t> fwd_copy, bwd_copy;     // the registered bytes: an equal pair registered again gets the old id
  uint64_t hash = 0;        // pair_hash of them: the registry's key
  bool transient = false;   // registered by zke_verify_email_with_regex on its own: may be evicted when the registry is full
  std::atomic<uint64_t> last_use{0};   // registry clock at the last registration / lookup hit
  std::atomic<uint32_t> pins{0};       // per-e-mail calls between their registration and the end of their batch: not evictable,
                                       // not unregistrable — an id must not change hands under a call that is about to use it
};

// 64-bit hash of a blob pair, eight bytes per step (a per-e-mail caller re-submits its part list with every e-mail: the
// lookup must cost far less than the batch; equality is still confirmed byte by byte on a hit).
uint64_t pair_hash(const uint8_t* fwd, size_t fl, const uint8_t* bwd, size_t bl) {
  auto mix = [](uint64_t h, uint64_t v) { h ^= v; h *= 0x9E3779B97F4A7C15ull; return h ^ (h >> 29); };
  auto run = [&](uint64_t h, const uint8_t* p, size_t n) {
    h = mix(h, (uint64_t)n);
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v; memcpy(&v, p + i, 8); h = mix(h, v); }
    uint64_t v = 0;
    for (size_t k = 0; i + k < n; k++) v |= (uint64_t)p[i + k] << (8 * k);
    return mix(h, v);
  };
  return run(run(0x243F6A8885A308D3ull, fwd, fl), bwd, bl);
}

// What a batch needs to know about one regex part, copied out of the registry while its lock is held.
struct PartInfo {
  const RegexDev* dev = nullptr;      // nullptr: the pair does not deserialise / the id is not registered
  size_t lds_bytes = 0;
  uint32_t idle = 0xFFFFFFFFu;
  uint32_t detail = ZKE_D_DFA_UNREGISTERED;
};

}  // namespace
