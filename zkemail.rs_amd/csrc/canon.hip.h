// canon.hip.h — body canonicalisation (cfdkim canonicalize_body_{simple,relaxed}, RFC 6376
// §3.4.3 / §3.4.4, l= truncation of hash::compute_body_hash) and the per-e-mail verdict.
// Call sites in the reference: core/src/email.rs:31-33 (inside verify_email_with_key) and
// core/src/circuits.rs:34-35 (canonicalize_signed_email).
//
// One e-mail per wavefront; normally the wave is the front-end wave of that e-mail (parse.hip.h calls
// canon_body_wave when it has chosen the candidate signature).  Relaxed canonicalisation is a stream compaction:
// four consecutive bytes per lane (256 B per step), loads eight steps ahead, keep / insert-SP decisions per byte,
// output offsets from count-bit ballots, bytes compacted in LDS and stored 16 bytes per lane; the end of the body is
// settled on the bytes still in LDS.  Simple canonicalisation moves no bytes: the SHA job points at the raw body
// with the trailing empty lines cut off.
#pragma once
#include "parse.hip.h"
#include <type_traits>

namespace zke {

__device__ __forceinline__ uint32_t ld_coherent_u8(const uint8_t* p) {
  // bytes this wave stored earlier in the kernel: read around L1 (sc1), after the stores have drained
  return (uint32_t)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// number of complete trailing CRLF pairs of a byte string of length len
// (*ended: a byte that breaks the run was seen, i.e. the run does not reach the start of the range)
template <class LD>
__device__ __forceinline__ uint32_t trailing_crlf_pairs(LD load, uint32_t len, bool* ended = nullptr) {
  uint32_t matched = 0, pos = len;
  if (ended) *ended = false;
  while (pos > 0) {
    const uint32_t lo = pos > 64 ? pos - 64 : 0;
    const uint32_t l = lo + lane_id();
    bool bad = false;
    if (l < pos) {
      const uint32_t c = load(l);
      bad = c != (((len - 1 - l) & 1) ? (uint32_t)'\r' : (uint32_t)'\n');
    }
    const uint64_t m = __ballot(bad);
    if (m) { matched += (pos - 1) - (lo + 63u - (uint32_t)__builtin_clzll(m)); if (ended) *ended = true; break; }
    matched += pos - lo;
    pos = lo;
  }
  return matched / 2;
}

// Canonicalise the body of e-mail i with the calling wave.  `flags`, `boff`, `blen` and the l= value are handed
// over in registers: the wave-per-e-mail front end calls this right after it has chosen the candidate signature
// (no launch boundary, no trip through EmailMeta); regex_prep_kernel (regex.hip.h) reads them from EmailMeta for the rare
// e-mail whose canonicalize_signed_email pass cannot reuse the verify pass.
// `lds`: CANON_LDS_BYTES of 16-byte aligned LDS the wave may overwrite (the front end hands over its staging buffer).
constexpr uint32_t CANON_LDS_TRASH = 3504, CANON_LDS_BYTES = 3504 + 64;
static_assert(CANON_LDS_BYTES <= PARSE_STAGE_BYTES, "the front end lends its staging buffer to the canonicaliser");
__device__ __forceinline__ void canon_body_wave(const BatchDev& B, uint32_t i, uint32_t mode, uint32_t flags, uint32_t boff,
                                                uint32_t blen, uint64_t len_tag, uint8_t* lds, bool ignore_l, bool bucket,
                                                uint32_t hdr_len) {
  const int lane = lane_id();
  EmailMeta* M = B.meta + i;
  zke_result* R = B.results + i;
  const uint64_t r0 = B.raw_off[i];
  const uint32_t raw_len = (uint32_t)(B.raw_off[i + 1] - r0);
  const uint8_t* body = B.raw + r0 + boff;
  uint8_t* regB = B.scratch + scratch_offset(r0 - B.raw_off[0], i) + (((size_t)raw_len + PRE_SLACK + 15) & ~(size_t)15);
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
    // 4 consecutive bytes per lane, 256 B per step.  Loads run eight steps (2 KB) ahead of their use, so a step
    // never waits for HBM; output bytes are compacted into LDS and leave in 16-byte lane-contiguous stores every
    // four steps instead of eight predicated byte stores per step.
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
    typedef uint4 __attribute__((aligned(1))) uint4_store_unaligned;
    auto load4 = [&](uint32_t pos) -> uint32_t {       // bytes pos..pos+3 of the body, zero beyond the end
      if (pos + 4 <= blen) return *(const u32_unaligned*)(body + pos);
      uint32_t v = 0;
      for (uint32_t t = 0; t < 4; t++) if (pos + t < blen) v |= (uint32_t)body[pos + t] << (8 * t);
      return v;
    };
    uint32_t o = 0;                           // bytes already in regB
    constexpr int RING = 8;                   // windows in flight: 2 KB of loads ahead of their use
    uint32_t q[RING];                         // the next RING 256-byte windows, one dword per lane each
#pragma unroll
    for (int k = 0; k < RING; k++) q[k] = load4(256u * k + 4 * lane);
    uint32_t prev_last = OOB;                 // byte before this step's 256-byte window
    uint32_t fill = 0;                        // bytes compacted into LDS and not yet stored
    uint32_t last_in = OOB;                   // last byte of the body
    bool tail_done = false;
    for (uint32_t base0 = 0; base0 < blen; base0 += 256u * RING) {
      // Clean prefix?  The leading windows of the group that lie wholly in front of the body's last 65 bytes (all
      // eight, except in the body's last group) are tested with byte-parallel arithmetic, a dword per lane: no byte
      // that could be WSP (<= 0x20 with bits 1 and 2 clear: SP, TAB and seven control codes, never CR or LF) unless
      // it is SP-like (bit 5) and the byte behind it is > 0x20; no WSP at either edge.  The test errs only towards
      // "not clean".  Such windows leave relaxed canonicalisation as they came (a single SP between words stays one
      // SP) and go out as coalesced dword stores: no LDS, no per-window scalar bookkeeping — ordinary text is almost
      // all such windows.  The windows behind the prefix take the path below, which also settles the body's end.
      const uint32_t rem = blen - base0;
      uint32_t kstart = 0;                      // first window of this group that takes the window path
      // FULL: all eight windows (straight-line code, the ring is refilled); otherwise windows [0, nfull), selected
      // with scalar masks instead of branches.  true: the windows are stored and o / prev_last have moved on.
      auto clean_prefix = [&](auto full_tag, uint32_t nfull) -> bool {
        constexpr bool FULL = decltype(full_tag)::value;
        uint32_t hi[RING];                      // bit 7 of byte j: byte j of q[k] is > 0x20
#pragma unroll
        for (int k = 0; k < RING; k++) hi[k] = ((q[k] & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | q[k];
        uint32_t bad = 0, glast = 0;
#pragma unroll
        for (int k = 0; k < RING; k++) {
          const uint32_t x = q[k];
          // the dword after mine: lane + 1, lane 63 takes the next window's first (behind the group's last byte
          // nothing is known: that byte is tested apart)
          const uint32_t nfirst = k + 1 < RING ? __builtin_amdgcn_readfirstlane(hi[(k + 1) & (RING - 1)]) : 0xFFFFFFFFu;
          const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp((int)nfirst, (int)hi[k], 0x130 /*wave_shl:1*/, 0xf, 0xf, false);
          const uint32_t nh = __builtin_amdgcn_alignbit(up, hi[k], 8);                            // byte j+1 under byte j
          const uint32_t wl = __builtin_amdgcn_bitop3_b32(hi[k], x << 6, x << 5, 0x01);          // ~(a | b | c): WSP-like
          const uint32_t v = __builtin_amdgcn_bitop3_b32(wl, nh, x << 2, 0x70);                  // a & ~(b & c)
          if (FULL) bad |= v;
          else {
            const uint32_t mk = (uint32_t)k < nfull ? 0xFFFFFFFFu : 0u;                           // scalar
            bad = __builtin_amdgcn_bitop3_b32(bad, v, mk, 0xF8);                                  // a | (b & c)
            const uint32_t g = __builtin_amdgcn_readlane(x, 63) >> 24;
            glast = (uint32_t)k + 1 == nfull ? g : glast;
          }
        }
        if (FULL) glast = __builtin_amdgcn_readlane(q[RING - 1], 63) >> 24;
        if (is_wsp(glast) || __ballot((bad & 0x80808080u) != 0) != 0) return false;
        if (fill) {                                                 // bytes of an earlier, dirty group still in LDS
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          for (uint32_t b0 = 0; b0 + 16 <= fill; b0 += 1024) { const uint32_t bq = b0 + 16u * lane; if (bq + 16 <= fill) *(uint4_store_unaligned*)(regB + o + bq) = *(const uint4*)(lds + bq); }
          const uint32_t tb = (fill & ~15u) + lane;
          if (lane < 16 && tb < fill) regB[o + tb] = lds[tb];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          o += fill; fill = 0;
        }
#pragma unroll
        for (int k = 0; k < RING; k++) {
          if (FULL) {
            *(u32_unaligned*)(regB + o + 256u * k + 4 * lane) = q[k];
            q[k] = load4(base0 + 256u * RING + 256u * k + 4 * lane);
          } else if ((uint32_t)k < nfull) {
            *(u32_unaligned*)(regB + o + 256u * k + 4 * lane) = q[k];
          }
        }
        o += 256u * nfull;
        prev_last = glast;
        return true;
      };
      if (!is_wsp(prev_last)) {
        if (rem > 256u * RING) {
          if (clean_prefix(std::true_type{}, (uint32_t)RING)) continue;
        } else if (rem >= 321u) {               // the body's last group: at least 65 bytes stay for the window path
          const uint32_t nf = (rem - 65u) >> 8;
          if (clean_prefix(std::false_type{}, nf)) kstart = nf;
        }
      }
#pragma unroll
      for (int k = 0; k < RING; k++) {
        const uint32_t base = base0 + 256u * k;
        if ((uint32_t)k >= kstart && base < blen) {   // wave-uniform
          const uint32_t cur = q[k];
          q[k] = load4(base + 256u * RING + 4 * lane);                             // RING steps ahead
          const uint32_t nxtw = q[(k + 1) & (RING - 1)];                                    // the window after this one
          const uint32_t pos0 = base + 4 * lane;
          uint32_t pv = lane_down(cur >> 24); if (lane == 0) pv = prev_last;       // byte in front of my 4
          uint32_t nb = lane_up(cur & 0xff);                                        // byte after my 4
          const uint32_t nfirst = __builtin_amdgcn_readfirstlane(nxtw) & 0xff;
          if (lane == 63) nb = (base + 256 < blen) ? nfirst : OOB;
          const uint32_t lastb = __builtin_amdgcn_readlane(cur, 63) >> 24;
          if (base + 256 >= blen) {           // the window that holds the last byte of the body
            const uint32_t ix = blen - 1 - base;
            last_in = (__builtin_amdgcn_readlane(cur, ix >> 2) >> (8 * (ix & 3))) & 0xff;
          }
          // Clean window: a full 256 bytes with no TAB, no WSP in front of WSP or CR, and no WSP at either edge
          // leaves relaxed canonicalisation as it came (a single SP between words stays one SP) — copy it.
          {
            bool need = false;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const uint32_t c = (cur >> (8 * j)) & 0xff;
              const uint32_t n = (j < 3) ? ((cur >> (8 * (j + 1))) & 0xff) : nb;
              need = need || c == '\t' || (c == ' ' && (n == ' ' || n == '\t' || n == '\r'));
            }
            if (base + 256 <= blen && !is_wsp(prev_last) && !is_wsp(lastb) && __ballot(need) == 0) {
              if ((fill & 3u) == 0) *(uint32_t*)(lds + fill + 4 * lane) = cur;
              else {
#pragma unroll
                for (int j = 0; j < 4; j++) lds[fill + 4 * lane + j] = (uint8_t)(cur >> (8 * j));
              }
              fill += 256;
              prev_last = lastb;
              continue;
            }
          }
          // up to 8 output bytes of this lane, appended into a 64-bit shift register (no indexed register array)
          uint64_t out64 = 0;
          uint32_t sh = 0;                    // 8 * bytes appended
          bool pw = is_wsp(pv);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t c = (cur >> (8 * j)) & 0xff;
            const uint32_t n = (j < 3) ? ((cur >> (8 * (j + 1))) & 0xff) : nb;
            const bool inr = pos0 + j < blen;
            const bool nin = pos0 + j + 1 < blen;
            const bool w = is_wsp(c);
            const bool kp = inr && !w;
            const bool sp = kp && pw && !(c == '\r' && nin && n == '\n');
            // "SP c" (two bytes), "c" (one) or nothing
            const uint32_t piece = sp ? (0x20u | (c << 8)) : c;
            const uint32_t plen = kp ? (sp ? 16u : 8u) : 0u;
            out64 |= (uint64_t)(kp ? piece : 0u) << sh;
            sh += plen;
            pw = inr ? w : pw;
          }
          const uint32_t cnt = sh >> 3;
          // exclusive prefix of cnt (0..8) over the lanes: one ballot per bit of the count
          const uint64_t below = bits_below(lane);
          uint32_t off = 0, total = 0;
#pragma unroll
          for (int bit = 0; bit < 4; bit++) {
            const uint64_t m = __ballot((cnt >> bit) & 1);
            off += (uint32_t)__builtin_popcountll(m & below) << bit;
            total += (uint32_t)__builtin_popcountll(m) << bit;
          }
          // byte t of the lane goes to its place, or to a per-lane trash byte behind the staging area when t >= cnt
          // (an unconditional store with a selected address is cheaper than eight exec-masked ones)
#pragma unroll
          for (int t = 0; t < 8; t++) {
            const uint32_t at = (uint32_t)t < cnt ? fill + off + t : CANON_LDS_TRASH + (uint32_t)lane;
            lds[at] = (uint8_t)(out64 >> (8 * t));
          }
          fill += total;
          prev_last = lastb;
        }
      }
      // Flush when another group (at most 256 * RING + 1 bytes of output) might not fit, or at the end.  Few, large flushes:
      // on gfx9 stores share vmcnt with the loads, so every flush is a point where the prefetched loads behind it
      // have to wait for store acknowledgements.
      const bool last_group = base0 + 256u * RING >= blen;
      if (fill + 256u * RING + 4 > CANON_LDS_TRASH || last_group) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (last_group) {
          // The end of the body, settled on the bytes still in LDS (no trip through memory):
          // a WSP run that ends the body is not followed by CRLF, so its single SP stays; then trailing empty
          // lines are cut to one CRLF, or a missing final CRLF is added.
          if (is_wsp(last_in)) { if (lane == 0) lds[fill] = ' '; fill++; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          bool ended = false;
          const uint32_t m = trailing_crlf_pairs([&](uint32_t l) { return (uint32_t)lds[l]; }, fill, &ended);
          if (ended || o == 0) {               // the run of CRLFs ends inside LDS (or LDS holds the whole output)
            if (m >= 2) fill -= 2 * (m - 1);
            else if (o + fill > 0 && m == 0) { if (lane < 2) lds[fill + lane] = lane ? '\n' : '\r'; fill += 2; }
            tail_done = true;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
          }
        }
        for (uint32_t b0 = 0; b0 + 16 <= fill; b0 += 1024) { const uint32_t bq = b0 + 16u * lane; if (bq + 16 <= fill) *(uint4_store_unaligned*)(regB + o + bq) = *(const uint4*)(lds + bq); }
        {
          const uint32_t tb = (fill & ~15u) + lane;
          if (lane < 16 && tb < fill) regB[o + tb] = lds[tb];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        o += fill;
        fill = 0;
      }
    }
    // trailing empty lines / missing final CRLF when the last flush could not settle them from LDS
    if (!tail_done) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      const uint32_t m = trailing_crlf_pairs([&](uint32_t l) { return ld_coherent_u8(regB + l); }, o);
      if (m >= 2) o -= 2 * (m - 1);
      else if (o > 0 && m == 0) { if (lane < 2) regB[o + lane] = lane ? '\n' : '\r'; o += 2; }
    }
    full = o;
  }
  uint32_t hashed = full;
  // STRICTNESS SITE canon_ignores_l (mode 1 only: ignore_l): canonicalize_signed_email returns the whole canonical body
  if ((flags & ZKE_F_HAS_LENGTH) && !ignore_l) {
    if (len_tag < hashed) hashed = (uint32_t)len_tag;
  }
  if (lane == 0) {
    M->canon_full_len = full; M->hashed_len = hashed; M->body_src_is_raw = src_is_raw;
    if (mode == 0) {
      R->canon_body_len = hashed;
      ShaJob j; j.src = (uint64_t)(src_is_raw ? body : regB); j.dst = (uint64_t)R->body_hash; j.len = hashed;
      j.pad = (flags & ZKE_F_SHA1) ? 1u : 0u;
      B.sha[i] = j;                             // kind 0
    }
  }
  if (bucket && lane < 2) sha_bucket(B.order, (uint32_t)lane, B.n_pad, i, lane ? hdr_len : hashed);      // body and header preimage
}

// ---- verdict of one signature round, by the wave that ran the e-mail's RSA job -----------------------
struct FinArgs { BatchDev b; uint32_t round, max_rounds; uint32_t* pending; uint32_t debug_skip_rsa; };   // debug_skip_rsa: ablation experiments only

// The verdict of e-mail i, by ONE LANE (64 e-mails per wave at once: the loads of an e-mail's state are a dependent
// chain of a microsecond or so, which a wave walking its e-mails one after the other would pay 64 times over).
// rsa_ok: the RSA outcome (EM's shape and digest both fit); ed_ok / ed_key_bad: the Ed25519 stage's.
// Returns true when the e-mail now waits for another signature round (ST_PENDING).
__device__ __forceinline__ bool verdict_lane(const FinArgs& A, uint32_t i, bool rsa_ok, bool ed_ok, bool ed_key_bad) {
  const BatchDev& B = A.b;
  EmailMeta* M = B.meta + i;
  zke_result* R = B.results + i;
  const uint32_t state = M->state;
  uint32_t status, detail;
  if (ed_key_bad) {
    // DkimPublicKey::try_from_bytes (email.rs:28-29) fails before any signature is looked at: nothing of the
    // DKIM scan may show in the record
    status = ZKE_KEY_DECODE_FAIL; detail = ZKE_D_KEY_ED25519_POINT;
    M->state = ST_FINAL; M->status = status; M->detail = detail;
    R->sig_index = 0; R->flags = 0; R->canon_header_len = 0; R->canon_body_len = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) { ((uint32_t*)R->body_hash)[k] = 0; ((uint32_t*)R->header_hash)[k] = 0; }
  } else if (state == ST_FINAL) {
    status = M->status; detail = M->detail;
  } else {
    // cfdkim verify_email_header: bh compare (as base64 strings), b= decode, RSA verify
    uint32_t err = 0;
    {
      // base64 of the body hash (32 bytes -> 44 chars, SHA-1: 20 -> 28), three bytes / four characters at a time, against bh=
      const bool sha1 = (M->flags & ZKE_F_SHA1) != 0;
      const uint32_t hl = sha1 ? 20u : 32u, nch = sha1 ? 28u : 44u;
      const uint8_t* h = R->body_hash;
      bool bad = M->bh_len != nch;
      auto b64c = [](uint32_t six) -> uint32_t {
        return six < 26 ? 'A' + six : (six < 52 ? 'a' + (six - 26) : (six < 62 ? '0' + (six - 52) : (six == 62 ? '+' : '/')));
      };
      for (uint32_t g = 0; g < nch / 4; g++) {
        const uint32_t b0 = h[3 * g], b1 = (3 * g + 1 < hl) ? h[3 * g + 1] : 0u, b2 = (3 * g + 2 < hl) ? h[3 * g + 2] : 0u;
        const uint32_t v = (b0 << 16) | (b1 << 8) | b2;
        uint32_t c3 = b64c(v & 63);
        if (4 * g + 3 == nch - 1) c3 = '=';                  // 32 and 20 are 2 mod 3: one pad character, the last
        const uint32_t want = b64c(v >> 18) | (b64c((v >> 12) & 63) << 8) | (b64c((v >> 6) & 63) << 16) | (c3 << 24);
        const uint32_t have = (uint32_t)M->bh[4 * g] | ((uint32_t)M->bh[4 * g + 1] << 8) | ((uint32_t)M->bh[4 * g + 2] << 16) | ((uint32_t)M->bh[4 * g + 3] << 24);
        bad = bad || want != have;
      }
      if (bad) err = ZKE_D_BODY_HASH_MISMATCH;
    }
    bool unsupported_here = false;
    if (!err && !M->sig_b64_ok) err = ZKE_D_SIG_B64;
    if (!err && M->even_modulus) { err = ZKE_D_U_EVEN_MODULUS; unsupported_here = true; }
    const bool sig_ok = (M->flags & ZKE_F_ED25519) ? ed_ok : rsa_ok;
    if (!err && !sig_ok) err = ZKE_D_SIG_MISMATCH;
    if (!err) {
      status = ZKE_OK; detail = 0;
      R->sig_index = M->cand_sig_index;
    } else if (M->cand_total > A.round + 1) {
      if (A.round + 1 < A.max_rounds) {
        if (unsupported_here) M->unsupported = err;
        M->state = ST_PENDING; M->cand_err = err; R->status = ZKE_DKIM_NOT_PASS; R->detail = err;
        atomicAdd(A.pending, 1u);          // statistics only: the next round runs in this launch (verdict.hip.h)
        return true;
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
#pragma unroll
    for (int k = 0; k < 8; k++) { ((uint32_t*)R->from_domain_hash)[k] = 0; ((uint32_t*)R->public_key_hash)[k] = 0; }
  }
  return false;
}

}  // namespace zke
