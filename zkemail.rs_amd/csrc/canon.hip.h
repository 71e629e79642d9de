// canon.hip.h — body canonicalisation (cfdkim canonicalize_body_{simple,relaxed}, RFC 6376
// §3.4.3 / §3.4.4, l= truncation of hash::compute_body_hash) and the per-e-mail verdict.
// Call sites in the reference: core/src/email.rs:31-33 (inside verify_email_with_key) and
// core/src/circuits.rs:34-35 (canonicalize_signed_email).
//
// One e-mail per wavefront.  Relaxed canonicalisation is a stream compaction: every lane
// owns one byte of a 64-byte chunk, the keep / insert-SP decisions are ballots, output
// offsets are popcounts of the lower lanes, and the next chunk is loaded one step ahead so
// the single dependent HBM load per step is off the critical path.  Simple canonicalisation
// moves no bytes: the SHA job points at the raw body with the trailing empty lines cut off.
#pragma once
#include "parse.hip.h"

namespace zke {

__device__ __forceinline__ uint32_t ld_coherent_u8(const uint8_t* p) {
  // bytes this wave stored earlier in the kernel: read around L1 (sc1), after the stores have drained
  return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// number of complete trailing CRLF pairs of a byte string of length len
template <class LD>
__device__ __forceinline__ uint32_t trailing_crlf_pairs(LD load, uint32_t len) {
  uint32_t matched = 0, pos = len;
  while (pos > 0) {
    const uint32_t lo = pos > 64 ? pos - 64 : 0;
    const uint32_t l = lo + lane_id();
    bool bad = false;
    if (l < pos) {
      const uint32_t c = load(l);
      bad = c != (((len - 1 - l) & 1) ? (uint32_t)'\r' : (uint32_t)'\n');
    }
    const uint64_t m = __ballot(bad);
    if (m) { matched += (pos - 1) - (lo + 63u - (uint32_t)__builtin_clzll(m)); break; }
    matched += pos - lo;
    pos = lo;
  }
  return matched / 2;
}

struct CanonArgs { BatchDev b; uint32_t mode; };

__global__ __launch_bounds__(64) void canon_body_kernel(CanonArgs A) {
  const BatchDev& B = A.b;
  const uint32_t i = blockIdx.x;
  if (i >= B.n) return;
  const int lane = lane_id();
  EmailMeta* M = B.meta + i;
  if (M->state != ST_CAND) return;
  if (A.mode == 1 && M->reuse) return;
  zke_result* R = B.results + i;
  const uint64_t r0 = B.raw_off[i];
  const uint32_t raw_len = (uint32_t)(B.raw_off[i + 1] - r0);
  const uint32_t boff = M->body_off, blen = M->body_len;
  const uint8_t* body = B.raw + r0 + boff;
  uint8_t* regB = B.scratch + B.scratch_off[i] + (((size_t)raw_len + PRE_SLACK + 15) & ~(size_t)15);
  const uint32_t flags = M->flags;
  uint32_t full = 0, src_is_raw = 0;

  if (!(flags & ZKE_F_BODY_RELAXED)) {
    if (blen == 0) {
      if (lane < 2) regB[lane] = lane ? '\n' : '\r';
      full = 2;
    } else {
      const uint32_t m = trailing_crlf_pairs([&](uint32_t l) { return (uint32_t)body[l]; }, blen);
      full = m >= 2 ? blen - 2 * (m - 1) : blen;
      src_is_raw = 1;
    }
  } else {
    uint32_t o = 0;
    uint32_t cur = (uint32_t)lane < blen ? body[lane] : OOB;
    uint32_t prev_last = OOB;                 // byte before this chunk
    for (uint32_t base = 0; base < blen; base += 64) {
      const uint32_t nl = base + 64 + lane;
      const uint32_t nxt = nl < blen ? body[nl] : OOB;          // next chunk, in flight while this one is processed
      const uint32_t c = cur;
      uint32_t cp = lane_down(c); if (lane == 0) cp = prev_last;
      uint32_t cn = lane_up(c);
      const uint32_t nfirst = __builtin_amdgcn_readfirstlane(nxt);
      if (lane == 63) cn = nfirst;
      const bool inr = base + lane < blen;
      const bool w = is_wsp(c);
      const bool k = inr && !w;
      const bool s = k && is_wsp(cp) && !(c == '\r' && cn == '\n');
      const uint64_t Km = __ballot(k), Sm = __ballot(s);
      const uint64_t below = bits_below(lane);
      const uint32_t off = o + (uint32_t)__builtin_popcountll(Km & below) + (uint32_t)__builtin_popcountll(Sm & below);
      if (s) { regB[off] = ' '; regB[off + 1] = (uint8_t)c; }
      else if (k) regB[off] = (uint8_t)c;
      o += (uint32_t)__builtin_popcountll(Km) + (uint32_t)__builtin_popcountll(Sm);
      prev_last = __builtin_amdgcn_readlane(c, 63);
      cur = nxt;
    }
    // a WSP run that ends the body is not followed by CRLF: its single SP stays
    if (blen && is_wsp((uint32_t)body[blen - 1])) { if (lane == 0) regB[o] = ' '; o++; }
    // trailing empty lines / missing final CRLF, on the bytes just written
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const uint32_t m = trailing_crlf_pairs([&](uint32_t l) { return ld_coherent_u8(regB + l); }, o);
    if (m >= 2) o -= 2 * (m - 1);
    else if (o > 0 && m == 0) { if (lane < 2) regB[o + lane] = lane ? '\n' : '\r'; o += 2; }
    full = o;
  }
  uint32_t hashed = full;
  if (flags & ZKE_F_HAS_LENGTH) {
    const uint64_t lt = ((uint64_t)M->len_tag_hi << 32) | M->len_tag_lo;
    if (lt < hashed) hashed = (uint32_t)lt;
  }
  if (lane == 0) {
    M->canon_full_len = full; M->hashed_len = hashed; M->body_src_is_raw = src_is_raw;
    if (A.mode == 0) {
      R->canon_body_len = hashed;
      ShaJob j; j.src = (uint64_t)(src_is_raw ? body : regB); j.dst = (uint64_t)R->body_hash; j.len = hashed; j.pad = 0;
      B.sha[i] = j;                             // kind 0
    }
  }
}

// ---- verdict of one signature round (thread per e-mail) -------------------------------
struct FinArgs { BatchDev b; const uint32_t* rsa_ok; uint32_t round, max_rounds; uint32_t* pending; };

__global__ void finalize_kernel(FinArgs A) {
  const BatchDev& B = A.b;
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B.n) return;
  EmailMeta* M = B.meta + i;
  zke_result* R = B.results + i;
  uint32_t status, detail;
  if (M->state == ST_PENDING) return;           // cannot happen after its own round; left for the next
  if (M->state == ST_FINAL) {
    status = M->status; detail = M->detail;
  } else {
    // cfdkim verify_email_header: bh compare (as base64 strings), b= decode, RSA verify
    uint32_t err = 0;
    {
      static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
      char b64[44];
      const uint8_t* h = R->body_hash;
      for (int t = 0; t < 10; t++) {
        const uint32_t v = ((uint32_t)h[3 * t] << 16) | ((uint32_t)h[3 * t + 1] << 8) | h[3 * t + 2];
        b64[4 * t] = T[v >> 18]; b64[4 * t + 1] = T[(v >> 12) & 63]; b64[4 * t + 2] = T[(v >> 6) & 63]; b64[4 * t + 3] = T[v & 63];
      }
      const uint32_t v = ((uint32_t)h[30] << 16) | ((uint32_t)h[31] << 8);
      b64[40] = T[v >> 18]; b64[41] = T[(v >> 12) & 63]; b64[42] = T[(v >> 6) & 63]; b64[43] = '=';
      bool same = M->bh_len == 44;
      for (int t = 0; t < 44 && same; t++) same = M->bh[t] == (uint8_t)b64[t];
      if (!same) err = ZKE_D_BODY_HASH_MISMATCH;
    }
    bool unsupported_here = false;
    if (!err && !M->sig_b64_ok) err = ZKE_D_SIG_B64;
    if (!err && M->even_modulus) { err = ZKE_D_U_EVEN_MODULUS; unsupported_here = true; }
    if (!err && !A.rsa_ok[i]) err = ZKE_D_SIG_MISMATCH;
    if (!err) {
      status = ZKE_OK; detail = 0;
      R->sig_index = M->cand_sig_index;
    } else if (M->cand_total > A.round + 1) {
      if (unsupported_here) M->unsupported = err;
      if (A.round + 1 < A.max_rounds) {
        M->state = ST_PENDING; M->cand_err = err; R->status = ZKE_DKIM_NOT_PASS; R->detail = err;
        atomicAdd(A.pending, 1u);
        return;
      }
      status = ZKE_UNSUPPORTED; detail = ZKE_D_U_TOO_MANY_SIGS;
      R->sig_index = M->last_touched_sig;
    } else {
      R->sig_index = M->last_touched_sig;
      const uint32_t uns = unsupported_here ? err : M->unsupported;
      if (uns) { status = ZKE_UNSUPPORTED; detail = uns; }
      else { status = ZKE_DKIM_NOT_PASS; detail = M->post_err ? M->post_err : err; }
    }
    M->state = ST_FINAL; M->status = status; M->detail = detail;
  }
  if (status == ZKE_OK && B.ext_null && B.ext_null[i]) status = ZKE_EXTERNAL_INPUT_NULL;   // circuits.rs:24
  R->status = status; R->detail = detail;
  if (status != ZKE_OK && status != ZKE_EXTERNAL_INPUT_NULL) {
    for (int t = 0; t < 32; t++) { R->from_domain_hash[t] = 0; R->public_key_hash[t] = 0; }
  }
}

}  // namespace zke
