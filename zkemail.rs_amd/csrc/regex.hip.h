// regex.hip.h — the regex half of verify_email_with_regex (core/src/circuits.rs:37-62):
//   * remove_quoted_printable_soft_breaks (core/src/email.rs:61-86) as a wave-wide stream compaction;
//   * process_regex_parts (core/src/regex.rs:15-53): regex-automata 0.4.9 dense-DFA search with the
//     exact find_iter / leftmost-first / delayed-match / EOI semantics, "exactly one match", and the
//     capture containment test.
//
// The DFA blob is parsed, validated and repacked ONCE per registration on the host
// (zke_dfa_register, replacing the per-e-mail dense::DFA::from_bytes of regex.rs:32-33); the
// kernel stages the repacked transition tables (u16 entries when every premultiplied state id
// fits, else u32) and the byte-class maps in LDS and walks one e-mail per lane: a DFA walk is a
// serial chain of dependent table lookups, so the batch, not the input, supplies the parallelism.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "canon.hip.h"

namespace zke {

struct DfaDev {                 // device image of one dense DFA
  uint32_t valid;
  uint32_t stride2, alphabet_len, state_len, table_len;
  uint32_t wide;                // 0: u16 entries, 1: u32 entries
  uint32_t start_kind;          // 0 both, 1 unanchored, 2 anchored
  uint32_t sp_max, quit_id, min_match, max_match;
  uint32_t has_empty, is_utf8, always_anchored;
  uint32_t quitset_nonempty;
  uint32_t starts[12];          // unanchored[6] then anchored[6] (Start::{NonWordByte,WordByte,Text,LineLF,LineCR,Custom})
  uint8_t classes[256];
  uint8_t start_map[256];
  uint8_t quitset[32];
  uint64_t table;               // device address of the entries
};

struct RegexDev { DfaDev fwd, rev; };

struct PartRes { uint32_t code; uint32_t count; uint32_t start; uint32_t end; };
constexpr uint32_t PART_DECODE_FAIL = 0xFFFFu;
constexpr uint32_t PART_SKIPPED = 0xFFFEu;

// ---- remove_quoted_printable_soft_breaks ---------------------------------------------------
struct QpArgs { BatchDev b; const EmailMeta* meta_v; uint8_t* clean; const uint64_t* clean_off; const uint8_t* scratch_v; const uint64_t* scratch_v_off; };

__device__ __forceinline__ const uint8_t* region_b(const uint8_t* scratch, const uint64_t* off, uint32_t i, uint32_t raw_len) {
  return scratch + off[i] + (((size_t)raw_len + PRE_SLACK + 15) & ~(size_t)15);
}

// remove_quoted_printable_soft_breaks by the calling wave: n bytes of canonical body at src -> dst (zero padded to n)
__device__ __forceinline__ void qp_wave(const uint8_t* src, uint32_t n, uint8_t* dst) {
  const int lane = lane_id();
  uint32_t o = 0;
  uint32_t carry = 0;            // how many leading bytes of this chunk belong to a "=\r\n" begun in the previous one
  for (uint32_t base = 0; base < n; base += 64) {
    const uint32_t l = base + lane;
    const uint32_t c = l < n ? src[l] : OOB, c1 = l + 1 < n ? src[l + 1] : OOB, c2 = l + 2 < n ? src[l + 2] : OOB;
    const uint64_t D = __ballot(c == '=' && c1 == '\r' && c2 == '\n');
    uint64_t drop = D | (D << 1) | (D << 2);
    if (carry == 2) drop |= 3; else if (carry == 1) drop |= 1;
    carry = (D >> 63) ? 2u : ((D >> 62) & 1 ? 1u : 0u);
    const bool k = l < n && !((drop >> lane) & 1);
    const uint64_t Km = __ballot(k);
    if (k) dst[o + (uint32_t)__builtin_popcountll(Km & bits_below(lane))] = (uint8_t)c;
    o += (uint32_t)__builtin_popcountll(Km);
  }
  for (uint32_t l = o + lane; l < n; l += 64) dst[l] = 0;     // email.rs:79: pad back to the original length
}

// The preparation of the regex stage as ONE launch, one e-mail per wave (it was three: the front end in mode 1, the body
// canonicaliser, the QP filter — each a dispatch that queues behind the other batches in flight):
//   canonicalize_signed_email (core/src/circuits.rs:34-35) — for nearly every e-mail the first DKIM-Signature header IS the
//   verified one and the verify pass's preimage and canonical body are reused: such a wave never stages or parses anything,
//   it copies a dozen words of the verify pass's EmailMeta; otherwise the front end runs again for the first signature
//   (mode 1) and the body is canonicalised;
//   remove_quoted_printable_soft_breaks (circuits.rs:37) — only when somebody will read the cleaned body: a batch without
//   body parts never looks at it (circuits.rs:48-56), so `want_clean` is off unless body parts or a parity buffer ask for it.
struct PrepArgs { ParseArgs parse; QpArgs qp; uint32_t want_clean; };
__global__ __launch_bounds__(64, ZKE_PARSE_WAVES) void regex_prep_kernel(PrepArgs A) {
  __shared__ ParseLds L;
  const BatchDev& B = A.parse.b;
  const uint32_t i = blockIdx.x;
  if (i >= B.n) return;
  // what parse_email<_, 1> is going to decide, from the verify pass's state (written by earlier launches: plain loads)
  const EmailMeta* V = B.meta_verify + i;
  const bool ok = B.results[i].status == ZKE_OK;
  const bool reuse = ok && (V->first_sig_hdr == V->cand_hdr || (A.parse.strict & ZKE_STRICT_CANON_VERIFIED));
  parse_email<true, 1>(A.parse, i, L);
  if (!ok) return;
  const uint64_t r0 = B.raw_off[i];
  const uint32_t raw_len = (uint32_t)(B.raw_off[i + 1] - r0);
  uint8_t* dst = A.qp.clean + A.qp.clean_off[i];
  if (reuse) {
    if (!A.want_clean) return;
    const uint32_t n = (A.parse.strict & ZKE_STRICT_CANON_IGNORES_L) ? V->canon_full_len : V->hashed_len;
    const uint8_t* src = V->body_src_is_raw ? B.raw + r0 + V->body_off : region_b(A.qp.scratch_v, A.qp.scratch_v_off, i, raw_len);
    qp_wave(src, n, dst);
    return;
  }
  // The rare e-mail whose first DKIM-Signature is not the verified one.  The wave reads back what its lane 0 stored to
  // EmailMeta, and further down the canonical body it wrote: past this CU's L1 (agent-scope fences: an L2 write-back /
  // invalidate on this chip, affordable on a path this rare)
  EmailMeta* M = B.meta + i;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  auto ld = [](const uint32_t* q) { return uni(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); };
  if (ld(&M->state) != ST_CAND) return;
  canon_body_wave(B, i, 1, ld(&M->flags), ld(&M->body_off), ld(&M->body_len), ((uint64_t)ld(&M->len_tag_hi) << 32) | ld(&M->len_tag_lo),
                  L.stage, (A.parse.strict & ZKE_STRICT_CANON_IGNORES_L) != 0, false);
  if (!A.want_clean) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  const uint8_t* src = ld(&M->body_src_is_raw) ? B.raw + r0 + ld(&M->body_off) : region_b(B.scratch, B.scratch_off, i, raw_len);
  qp_wave(src, ld(&M->hashed_len), dst);
}

// ---- dense DFA search ------------------------------------------------------------------------
struct DfaLds {               // where one DFA lives during the kernel
  const void* table;          // LDS or global
  const uint8_t* classes;
  const uint8_t* start_map;
  uint32_t wide, alphabet_len, sp_max, quit_id, min_match, max_match, start_kind;
  uint32_t has_empty, is_utf8, always_anchored, quitset_nonempty;
  const DfaDev* dev;          // starts[] and quitset stay in global/constant memory (rarely read)
};

__device__ __forceinline__ uint32_t dfa_tr(const DfaLds& d, uint32_t sid, uint32_t cls) {
  return d.wide ? ((const uint32_t*)d.table)[sid + cls] : (uint32_t)((const uint16_t*)d.table)[sid + cls];
}
__device__ __forceinline__ bool dfa_is_match(const DfaLds& d, uint32_t s) { return s != 0 && d.min_match <= s && s <= d.max_match; }
__device__ __forceinline__ bool dfa_is_quit(const DfaLds& d, uint32_t s) { return s != 0 && s == d.quit_id; }

// Automaton::start_state.  returns false on quit / unsupported anchored mode
__device__ __forceinline__ bool dfa_start(const DfaLds& d, bool anchored, bool have_look, uint32_t look, uint32_t& sid) {
  uint32_t st = 2;   // Start::Text
  if (have_look) {
    if (d.quitset_nonempty && ((d.dev->quitset[look >> 3] >> (look & 7)) & 1)) return false;
    st = d.start_map[look];
  }
  if (!anchored) { if (d.start_kind == 2) return false; sid = d.dev->starts[st]; }
  else { if (d.start_kind == 1) return false; sid = d.dev->starts[6 + st]; }
  return true;
}

typedef uint4 __attribute__((aligned(1))) uint4_u1;

// Chunk map of one haystack for the wave-per-e-mail kernel: the haystack is cut into 64 chunks of C bytes, and bit l
// of `clean` says that a walk of chunk l that ENTERS in state X also LEAVES in state X and meets no match, dead or
// quit state on the way (64 lanes establish that in parallel, one chunk each).  A forward search that stands at a
// chunk boundary in state X with no match pending can therefore pass over every following clean chunk at once —
// exactly, whatever X is.  (X is the state the unanchored search idles in between matches, so in ordinary text
// almost every chunk is clean and the serial walk shrinks to the chunks around the matches.)
struct DfaAccel { uint64_t clean; uint32_t C, X; bool on; };

// dfa/search.rs find_fwd (leftmost, earliest = false).  1 match, 0 none, -1 quit
__device__ __forceinline__ int dfa_find_fwd(const DfaLds& d, const uint8_t* hay, uint32_t hlen, uint32_t start, uint32_t end,
                                            uint32_t& mend, const DfaAccel& acc = DfaAccel{0, 0, 0, false}) {
  if (start > end) return 0;
  uint32_t sid;
  if (!dfa_start(d, false, start > 0, start > 0 ? hay[start - 1] : 0, sid)) return -1;
  int have = 0;
  uint32_t at = start;
  while (at < end) {
    uint32_t lim = end;
    if (acc.on) {
      const uint32_t l = at / acc.C;
      if (!have && sid == acc.X && at == l * acc.C) {
        const uint64_t dirty = l < 64 ? (~acc.clean >> l) : 0ull;
        const uint32_t skip = dirty ? (uint32_t)__builtin_ctzll(dirty) : 64u - l;
        if (skip) {
          const uint64_t to = (uint64_t)(l + skip) * acc.C;
          at = to < end ? (uint32_t)to : end;
          continue;
        }
      }
      const uint64_t nb_end = (uint64_t)(l + 1) * acc.C;          // stop at the next chunk boundary: the test above runs there
      if (nb_end < lim) lim = (uint32_t)nb_end;
    }
    // 16 haystack bytes per load (the buffers carry 16 bytes of slack), walked from registers
    const uint4 v = *(const uint4_u1*)(hay + at);
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
    const uint32_t nb = lim - at < 16 ? lim - at : 16;
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if ((uint32_t)j < nb) {
        const uint32_t b = (wv[j >> 2] >> (8 * (j & 3))) & 0xff;
        sid = dfa_tr(d, sid, d.classes[b]);
        if (sid <= d.sp_max) {
          if (dfa_is_match(d, sid)) { have = 1; mend = at + j; }
          else if (sid == 0) return have;
          else if (dfa_is_quit(d, sid)) return -1;
        }
      }
    }
    at += nb;
  }
  if (end < hlen) {
    sid = dfa_tr(d, sid, d.classes[hay[end]]);
    if (dfa_is_match(d, sid)) { have = 1; mend = end; }
    else if (dfa_is_quit(d, sid)) return -1;
  } else {
    sid = dfa_tr(d, sid, d.alphabet_len - 1);
    if (dfa_is_match(d, sid)) { have = 1; mend = hlen; }
  }
  return have;
}
// find_rev: anchored reverse search over [start, end)
__device__ __forceinline__ int dfa_find_rev(const DfaLds& d, const uint8_t* hay, uint32_t hlen, uint32_t start, uint32_t end,
                                            uint32_t& mstart) {
  uint32_t sid;
  if (!dfa_start(d, true, end < hlen, end < hlen ? hay[end] : 0, sid)) return -1;
  int have = 0;
  for (uint32_t at = end; at-- > start;) {
    sid = dfa_tr(d, sid, d.classes[hay[at]]);
    if (sid <= d.sp_max) {
      if (dfa_is_match(d, sid)) { have = 1; mstart = at + 1; }
      else if (sid == 0) return have;
      else if (dfa_is_quit(d, sid)) return -1;
    }
  }
  if (start > 0) {
    sid = dfa_tr(d, sid, d.classes[hay[start - 1]]);
    if (dfa_is_match(d, sid)) { have = 1; mstart = start; }
    else if (dfa_is_quit(d, sid)) return -1;
  } else {
    sid = dfa_tr(d, sid, d.alphabet_len - 1);
    if (dfa_is_match(d, sid)) { have = 1; mstart = 0; }
  }
  return have;
}
__device__ __forceinline__ bool is_char_boundary(const uint8_t* hay, uint32_t hlen, uint32_t off) {
  if (off >= hlen) return off == hlen;
  return (int8_t)hay[off] >= -0x40;
}
// Automaton::try_search_fwd incl. util::empty::skip_splits_fwd
__device__ __forceinline__ int dfa_search_fwd(const DfaLds& d, const uint8_t* hay, uint32_t hlen, uint32_t start, uint32_t end,
                                              uint32_t& mend, const DfaAccel& acc = DfaAccel{0, 0, 0, false}) {
  int r = dfa_find_fwd(d, hay, hlen, start, end, mend, acc);
  if (r <= 0) return r;
  if (!(d.has_empty && d.is_utf8)) return 1;
  while (!is_char_boundary(hay, hlen, mend)) {
    start++;
    r = dfa_find_fwd(d, hay, hlen, start, end, mend, acc);
    if (r <= 0) return r;
  }
  return 1;
}
// dfa::regex::Regex::try_search
__device__ __forceinline__ int regex_search(const DfaLds& f, const DfaLds& rv, const uint8_t* hay, uint32_t hlen, uint32_t start,
                                            uint32_t end, uint32_t& ms, uint32_t& me, const DfaAccel& acc = DfaAccel{0, 0, 0, false}) {
  uint32_t e;
  int r = dfa_search_fwd(f, hay, hlen, start, end, e, acc);
  if (r <= 0) return r;
  me = e;
  if (start == e) { ms = e; return 1; }
  if (f.always_anchored) { ms = start; return 1; }
  uint32_t s;
  r = dfa_find_rev(rv, hay, hlen, start, e, s);
  if (r <= 0) return -1;     // .expect("reverse search must match if forward search does")
  ms = s;
  return 1;
}

__device__ __forceinline__ bool contains_bytes(const uint8_t* h, uint32_t hl, const uint8_t* nd, uint32_t nl) {
  if (nl == 0) return true;
  if (nl > hl) return false;
  for (uint32_t i = 0; i + nl <= hl; i++) {
    if (h[i] != nd[0]) continue;
    uint32_t j = 1;
    while (j < nl && h[i + j] == nd[j]) j++;
    if (j == nl) return true;
  }
  return false;
}
__device__ __forceinline__ bool utf8_valid(const uint8_t* s, uint32_t n) {
  uint32_t i = 0;
  while (i < n) {
    const uint32_t c = s[i];
    if (c < 0x80) { i++; continue; }
    uint32_t need, lo;
    if (c >= 0xC2 && c <= 0xDF) { need = 1; lo = 0x80; }
    else if (c >= 0xE0 && c <= 0xEF) { need = 2; lo = 0x800; }
    else if (c >= 0xF0 && c <= 0xF4) { need = 3; lo = 0x10000; }
    else return false;
    if (n - i <= need) return false;
    uint32_t cp = c & (0x3Fu >> need);
    for (uint32_t k = 1; k <= need; k++) {
      if ((s[i + k] & 0xC0) != 0x80) return false;
      cp = (cp << 6) | (s[i + k] & 0x3F);
    }
    if (cp < lo || cp > 0x10FFFF || (cp >= 0xD800 && cp <= 0xDFFF)) return false;
    i += need + 1;
  }
  return true;
}

struct DfaArgs {
  BatchDev b;                       // meta = the canonicalize pass's meta
  const RegexDev* re;               // device image of this part's regex (nullptr: registered blob was invalid)
  uint32_t part;                    // index over header parts then body parts
  uint32_t P;                       // total parts
  uint32_t is_body;
  const uint8_t* scratch_v; const uint64_t* scratch_v_off;   // the verify pass's scratch (reused preimages)
  const uint8_t* clean; const uint64_t* clean_off;           // QP-cleaned bodies
  const uint32_t* cap_off; const uint32_t* cap_str_off; const uint8_t* cap_blob;
  PartRes* out;                     // [n * P]
  uint32_t lds_tables;              // 1: both tables fit the dynamic LDS allocation
  uint32_t idle;                    // dfa_wave_kernel: the forward automaton's idle state (host-chosen; ~0: none, plain serial search)
  uint32_t decode_detail;           // re == nullptr: the ZKE_D_DFA_* section at which dense::DFA::from_bytes gives up (regex.rs:32-33)
};

// stage both automata of the part in the block's dynamic LDS (all threads of the block take part)
__device__ __forceinline__ bool dfa_stage(const DfaArgs& A, uint8_t* dlds, DfaLds& F, DfaLds& Rv) {
  const RegexDev* re = A.re;
  const bool valid = re && re->fwd.valid && re->rev.valid;
  if (valid) {
    const DfaDev* dv[2] = {&re->fwd, &re->rev};
    DfaLds* dl[2] = {&F, &Rv};
    size_t off = 0;
    for (int t = 0; t < 2; t++) {
      const DfaDev* d = dv[t];
      const size_t bytes = (size_t)d->table_len * (d->wide ? 4 : 2);
      if (A.lds_tables) {
        const uint32_t* src = (const uint32_t*)d->table;
        uint32_t* dst = (uint32_t*)(dlds + off);
        for (size_t k = threadIdx.x; k < (bytes + 3) / 4; k += blockDim.x) dst[k] = src[k];
        dl[t]->table = dlds + off;
        off += (bytes + 15) & ~(size_t)15;
      } else {
        dl[t]->table = (const void*)d->table;
      }
    }
    for (int t = 0; t < 2; t++) {
      const DfaDev* d = dv[t];
      uint8_t* cm = dlds + off; off += 256;
      uint8_t* sm = dlds + off; off += 256;
      for (uint32_t k = threadIdx.x; k < 256; k += blockDim.x) { cm[k] = d->classes[k]; sm[k] = d->start_map[k]; }
      DfaLds* l = dl[t];
      l->classes = cm; l->start_map = sm; l->wide = d->wide; l->alphabet_len = d->alphabet_len; l->sp_max = d->sp_max;
      l->quit_id = d->quit_id; l->min_match = d->min_match; l->max_match = d->max_match; l->start_kind = d->start_kind;
      l->has_empty = d->has_empty; l->is_utf8 = d->is_utf8; l->always_anchored = d->always_anchored;
      l->quitset_nonempty = d->quitset_nonempty; l->dev = d;
    }
  }
  __syncthreads();
  return valid;
}

// process_regex_parts for one (e-mail, part): find_iter, exactly one match, captures contained (core/src/regex.rs:35-46)
__device__ __forceinline__ PartRes dfa_part(const DfaArgs& A, const DfaLds& F, const DfaLds& Rv, uint32_t i, const uint8_t* hay,
                                            uint32_t hlen, const DfaAccel& acc) {
  PartRes pr{0, 0, 0, 0};
  // util::iter::Searcher: non-overlapping; an empty match abutting the previous end restarts one byte on
  uint32_t start = 0, count = 0, last_end = 0, fs = 0, fe = 0;
  bool have_last = false;
  int bad = 0;
  for (;;) {
    uint32_t ms, me;
    int r = regex_search(F, Rv, hay, hlen, start, hlen, ms, me, acc);
    if (r < 0) { bad = 1; break; }
    if (r == 0) break;
    if (ms == me && have_last && me == last_end) {
      start += 1;
      r = regex_search(F, Rv, hay, hlen, start, hlen, ms, me, acc);
      if (r < 0) { bad = 1; break; }
      if (r == 0) break;
    }
    if (count == 0) { fs = ms; fe = me; }
    if (count < 2) count++;                    // (the record's count saturates at 2)
    // no early exit at the second match: the reference collects the WHOLE iterator (regex.rs:36) before it counts, so a quit
    // state further on is still a panic (ZKE_D_RE_QUIT), not a miscount — found by tests/test_gpu_regex.py::test_mutated_automata_parity
    start = me; have_last = true; last_end = me;
  }
  pr.count = count; pr.start = fs; pr.end = fe; pr.code = 0;
  if (bad) { pr.code = ZKE_D_RE_QUIT; pr.count = 0; pr.start = 0; pr.end = 0; }
  else if (count != 1) pr.code = ZKE_D_RE_MATCH_COUNT;             // core/src/regex.rs:37
  else if (A.cap_off) {
    const uint32_t c0 = A.cap_off[(size_t)i * A.P + A.part], c1 = A.cap_off[(size_t)i * A.P + A.part + 1];
    const uint8_t* m = hay + fs; const uint32_t ml = fe - fs;
    int mvalid = -1;
    for (uint32_t c = c0; c < c1 && !pr.code; c++) {
      const uint8_t* cs = A.cap_blob + A.cap_str_off[c];
      const uint32_t cl = A.cap_str_off[c + 1] - A.cap_str_off[c];
      const uint8_t fffd[3] = {0xEF, 0xBF, 0xBD};
      if (contains_bytes(cs, cl, fffd, 3)) {
        if (mvalid < 0) mvalid = utf8_valid(m, ml) ? 1 : 0;
        if (!mvalid) { pr.code = ZKE_D_U_CAPTURE_FFFD; break; }
      }
      if (!contains_bytes(m, ml, cs, cl)) pr.code = ZKE_D_RE_CAPTURE_MISSING;   // core/src/regex.rs:43-46
    }
  }
  return pr;
}

__device__ __forceinline__ void dfa_haystack(const DfaArgs& A, uint32_t i, const EmailMeta* M, const uint8_t*& hay, uint32_t& hlen) {
  if (A.is_body) { hay = A.clean + A.clean_off[i]; hlen = M->hashed_len; }
  else { hay = (M->reuse ? A.scratch_v + A.scratch_v_off[i] : A.b.scratch + A.b.scratch_off[i]); hlen = M->preimage_len; }
}

// ---- the verdict of the regex stage, folded part by part ------------------------------------------------------------
// process_regex_parts walks the parts in order and stops at the first that fails (core/src/regex.rs:35-46; header parts, then
// body parts: circuits.rs:43-56).  The fold below is that walk with the record as its state: `decided` = a part has failed
// (or the e-mail never got this far), later parts no longer touch the record.
struct RegexFold { bool decided; };
__device__ __forceinline__ RegexFold regex_fold_begin(zke_result* R, const EmailMeta* M) {
  if (R->status != ZKE_OK) return RegexFold{true};                 // verify_email already panicked (circuits.rs:32)
  if (M->state == ST_FINAL) { R->status = M->status; R->detail = M->detail; return RegexFold{true}; }   // circuits.rs:35
  return RegexFold{false};
}
__device__ __forceinline__ void regex_fold_part(RegexFold& F, zke_result* R, uint32_t p, uint32_t n_header_parts, const PartRes& pr) {
  if (F.decided) return;
  R->regex_part = p; R->match_count = 0; R->match_start = 0; R->match_end = 0;
  if (pr.code == PART_DECODE_FAIL) { R->status = ZKE_DFA_DECODE_FAIL; R->detail = pr.count; F.decided = true; return; }   // regex.rs:32-33; detail: the blob section
  R->match_count = pr.count; R->match_start = pr.start; R->match_end = pr.end;
  if (pr.code) {
    R->status = pr.code == ZKE_D_U_CAPTURE_FFFD ? ZKE_UNSUPPORTED : (p < n_header_parts ? ZKE_HEADER_REGEX_FAIL : ZKE_BODY_REGEX_FAIL);
    R->detail = pr.code;
    F.decided = true;
  }
}

// Parts [part0, part0 + np) of a batch in ONE launch: blockIdx.y picks the part (each block stages that part's tables), so
// the parts' walks overlap instead of queueing behind each other — these launches are a few dozen to a few hundred blocks on
// a 256-CU chip, their duration is one walk's latency.  `finalize` (only with np == 1): this is the regex stage's last
// launch and it holds the last part alone — its threads fold the results of the parts before part0 (written by earlier
// launches) and then their own into the records, and the stage needs no verdict launch.
constexpr uint32_t DFA_MULTI_MAX = 8;
struct DfaMultiArgs {
  DfaArgs common;                         // part / re / is_body / lds_tables / idle are filled per part from the arrays
  uint32_t part0, np;                     // this launch's parts
  uint32_t n_header_parts;
  uint32_t finalize;
  const RegexDev* re[DFA_MULTI_MAX];
  uint32_t lds_tables[DFA_MULTI_MAX];
  uint32_t idle[DFA_MULTI_MAX];
  uint32_t detail[DFA_MULTI_MAX];
};
__device__ __forceinline__ DfaArgs dfa_part_args(const DfaMultiArgs& MA, uint32_t k) {
  DfaArgs A = MA.common;
  A.part = MA.part0 + k;
  A.re = MA.re[k]; A.lds_tables = MA.lds_tables[k]; A.idle = MA.idle[k]; A.decode_detail = MA.detail[k];
  A.is_body = A.part >= MA.n_header_parts ? 1u : 0u;
  return A;
}

// blockDim = 256; one e-mail per lane.  Dynamic LDS: fwd table | rev table | 4 x 256-byte maps.
__global__ __launch_bounds__(256) void dfa_kernel(DfaMultiArgs MA) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dlds[];
  const DfaArgs A = dfa_part_args(MA, blockIdx.y);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  DfaLds F{}, Rv{};
  const bool valid = dfa_stage(A, dlds, F, Rv);
  if (i >= A.b.n) return;
  const EmailMeta* M = A.b.meta + i;
  zke_result* R = A.b.results + i;
  RegexFold fold{false};
  if (MA.finalize) {
    fold = regex_fold_begin(R, M);
    for (uint32_t p = 0; p < MA.part0; p++) regex_fold_part(fold, R, p, MA.n_header_parts, A.out[(size_t)i * A.P + p]);
  }
  PartRes pr{PART_SKIPPED, 0, 0, 0};
  if (M->state == ST_CAND && !fold.decided) {      // (a part behind the first failing one is never looked at: regex.rs:38,45)
    if (!valid) {
      pr.code = PART_DECODE_FAIL; pr.count = A.decode_detail;
    } else {
      const uint8_t* hay; uint32_t hlen;
      dfa_haystack(A, i, M, hay, hlen);
      pr = dfa_part(A, F, Rv, i, hay, hlen, DfaAccel{0, 0, 0, false});
    }
  }
  A.out[(size_t)i * A.P + A.part] = pr;
  if (MA.finalize) regex_fold_part(fold, R, A.part, MA.n_header_parts, pr);
}

// One e-mail per WAVE (blockDim = 256: four e-mails share the staged tables).  A DFA walk is a serial chain, but
// where the chain can be cut is knowable in parallel: the 64 lanes first walk one chunk of the haystack each from
// the automaton's idle state `A.idle` and report which chunks are "clean" (DfaAccel); the exact search that follows
// — the same code as the lane-per-e-mail kernel, run wave-uniformly — then steps over runs of clean chunks and walks
// only the chunks around the matches.  4 KB bodies: 64 dependent steps in the parallel pass + a few hundred serial
// ones instead of ~4 100.
__global__ __launch_bounds__(256) void dfa_wave_kernel(DfaMultiArgs MA) {
  extern __shared__ __attribute__((aligned(16))) uint8_t dlds[];
  const DfaArgs A = dfa_part_args(MA, blockIdx.y);
  const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  DfaLds F{}, Rv{};
  const bool valid = dfa_stage(A, dlds, F, Rv);
  if (i >= A.b.n) return;
  const EmailMeta* M = A.b.meta + i;
  zke_result* R = A.b.results + i;
  RegexFold fold{false};                           // (the record is lane 0's business; every lane gets its `decided`)
  if (MA.finalize) {
    if (lane == 0) {
      fold = regex_fold_begin(R, M);
      for (uint32_t p = 0; p < MA.part0; p++) regex_fold_part(fold, R, p, MA.n_header_parts, A.out[(size_t)i * A.P + p]);
    }
    fold.decided = __builtin_amdgcn_readfirstlane((int)fold.decided) != 0;
  }
  PartRes pr{PART_SKIPPED, 0, 0, 0};
  if (M->state == ST_CAND && !fold.decided) {
    if (!valid) {
      pr.code = PART_DECODE_FAIL; pr.count = A.decode_detail;
    } else {
      const uint8_t* hay; uint32_t hlen;
      dfa_haystack(A, i, M, hay, hlen);
      DfaAccel acc{0, 0, 0, false};
      const uint32_t X = A.idle;
      if (X != 0xFFFFFFFFu && hlen >= 256) {
        // chunk size: a multiple of 16 (the walk loads 16 bytes at a time), 64 chunks cover the haystack
        const uint32_t C = ((hlen + 63) / 64 + 15) & ~15u;
        const uint32_t lo = (uint32_t)lane * C;
        bool clean = true;
        if (lo < hlen) {
          const uint32_t hi = lo + C < hlen ? lo + C : hlen;
          uint32_t sid = X;
          for (uint32_t at = lo; at < hi && clean; at += 16) {
            const uint4 v = *(const uint4_u1*)(hay + at);
            const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
            const uint32_t nb = hi - at < 16 ? hi - at : 16;
#pragma unroll
            for (int j = 0; j < 16; j++) {
              if ((uint32_t)j < nb) {
                sid = dfa_tr(F, sid, F.classes[(wv[j >> 2] >> (8 * (j & 3))) & 0xff]);
                if (sid <= F.sp_max && (dfa_is_match(F, sid) || sid == 0 || dfa_is_quit(F, sid))) clean = false;
              }
            }
          }
          clean = clean && sid == X;
        }
        acc.clean = __ballot(clean); acc.C = C; acc.X = X; acc.on = true;
      }
      pr = dfa_part(A, F, Rv, i, hay, hlen, acc);
    }
  }
  if (lane == 0) {
    A.out[(size_t)i * A.P + A.part] = pr;
    if (MA.finalize) regex_fold_part(fold, R, A.part, MA.n_header_parts, pr);
  }
}

// ---- verdict of the regex stage as a launch of its own (when the last DFA launch holds more than one part) ----------------
struct RegexFinArgs { BatchDev b; const PartRes* parts; uint32_t n_header_parts, n_body_parts; };

__global__ void regex_finalize_kernel(RegexFinArgs A) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.b.n) return;
  zke_result* R = A.b.results + i;
  RegexFold fold = regex_fold_begin(R, A.b.meta + i);
  const uint32_t P = A.n_header_parts + A.n_body_parts;
  for (uint32_t p = 0; p < P; p++) regex_fold_part(fold, R, p, A.n_header_parts, A.parts[(size_t)i * P + p]);
}

}  // namespace zke
