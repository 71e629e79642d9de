// verdict.hip.h — the last launch of a signature round: the Ed25519 stage and the verdict of every e-mail.
//
// One wave per 16 e-mails.  First, four lanes per e-mail (ed25519.hip.h), what ed25519_email_kernel did as a launch of its
// own: the curve-point check of every 32-byte Ed25519 key (round 0; DkimPublicKey::try_from_bytes, core/src/email.rs:28-29)
// and the Ed25519 verification of a=ed25519-sha256 candidates over the SHA-256 header hash — waves without an Ed25519 e-mail
// skip it after one byte load per lane.  Then a lane per e-mail writes the verdict (verdict_lane, canon.hip.h): base64(body hash)
// against bh=, the digest bytes the RSA role left in EmailMeta against the header hash (rsa 0.9.6 pkcs1v15 verify),
// status / detail, the pending counter of the next signature round.  One launch instead of two: with many batches in
// flight every launch of a batch costs the chip's command processor several microseconds whatever it does.
#pragma once
#include "ed25519.hip.h"
#include "rsa_kernel.hip.h"

namespace zke {

struct EdVerdictArgs {
  FinArgs fin; uint32_t skip_ed;
  uint32_t* wave_count;      // the hash / modexp stage's job list is consumed: reset for the next batch
  KeyCacheEntry* cache;      // later signature rounds: per-key Montgomery constants
  uint8_t* em_out;           // parity intermediates (nullptr in production)
  uint32_t strict;           // ZKE_STRICT_* and the x= clock, for the front end of later signature rounds
  uint64_t now;
  uint32_t* wave_feedback;   // pinned host word of the slot: how long the wave-routine job list of this batch was (sizes the next
                             // batch's walkers: engine.hip, launch_hash_modexp); nullptr: nobody listens
};

// SHA-256 / SHA-1 of one message by ONE LANE (later signature rounds only: two messages per e-mail, a rare path; the
// batched kernels of sha256.hip.h are the hot one).  FIPS 180-4; the digest goes to j.dst as the batch kernels write it.
__device__ __forceinline__ void sha_lane(const ShaJob& j) {
  if (!j.dst) return;
  const uint8_t* src = (const uint8_t*)j.src;
  const uint32_t len = j.len, nblk = (len + 9 + 63) >> 6;
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  if (j.pad) { st[0] = 0x67452301; st[1] = 0xEFCDAB89; st[2] = 0x98BADCFE; st[3] = 0x10325476; st[4] = 0xC3D2E1F0; st[5] = st[6] = st[7] = 0; }
  typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
  for (uint32_t b = 0; b < nblk; b++) {
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const uint32_t o = 64 * b + 4 * k;
      uint32_t v;
      if (o + 4 <= len) v = __builtin_bswap32(*(const u32_unaligned*)(src + o));
      else {
        v = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const uint32_t q = o + t;
          const uint32_t c = q < len ? (uint32_t)src[q] : (q == len ? 0x80u : 0u);
          v |= c << (24 - 8 * t);
        }
      }
      w[k] = v;
    }
    if (b == nblk - 1) { const uint64_t bits = (uint64_t)len * 8; w[14] = (uint32_t)(bits >> 32); w[15] = (uint32_t)bits; }
    if (j.pad) sha1_compress(st, w); else sha256_compress(st, w);
  }
  uint32_t* out = (uint32_t*)j.dst;
#pragma unroll
  for (int k = 0; k < 8; k++) out[k] = (j.pad && k >= 5) ? 0u : __builtin_bswap32(st[k]);
}

// what one lane stored, the other lanes of this wave read back: past this CU's L1 (rare paths only: an agent-scope
// fence is an L2 write-back / invalidate on this chip)
__device__ __forceinline__ void wave_publish() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// The front end, the two hashes and the RSA operation of signature round `round` for the e-mails base + (bits of pend),
// by the whole wave, one e-mail after the other.  A real call: inlined, its loop-invariant addresses would be kept (and
// spilled) across the Ed25519 stage of the caller's loop.
__device__ __noinline__ void next_round(const EdVerdictArgs& A, uint32_t round, uint64_t pend_mask, uint32_t base) {
  __shared__ ParseLds L;
  const BatchDev& B = A.fin.b;
  const int lane = threadIdx.x & 63;
  for (uint64_t pend = pend_mask; pend; pend &= pend - 1) {
    const uint32_t e = base + (uint32_t)__builtin_ctzll(pend);
    const EmailMeta* M = B.meta + e;
    wave_publish();                                   // the verdict lane's EmailMeta / record stores
    ParseArgs pa{B, round, 0, 0, A.strict, A.now, nullptr, 0, nullptr, nullptr};
    parse_email<false>(pa, e, L);
    wave_publish();                                   // the front end's jobs, preimage and canonical body
    if (M->state != ST_CAND) continue;
    if (lane < 2) sha_lane(B.sha[(size_t)lane * B.n_pad + e]);      // kind 0: body, kind 1: header preimage
    wave_publish();
    if (!(M->flags & ZKE_F_ED25519))
      rsa_wave_any(B.rsa, e, nullptr, 0, nullptr, A.em_out, A.cache, B.meta, A.fin.debug_skip_rsa);     // em_ok / em_tail -> EmailMeta
  }
}

#ifndef ZKE_VERDICT_WAVES
#define ZKE_VERDICT_WAVES 2      // waves per SIMD the verdict launch is compiled for: at 1 it takes 364 registers (256 + 108 AGPRs), which
                                 // keeps the waves of other launches off its SIMD — RSA batches 26.8 M e-mails/s instead of 27.8 M,
                                 // Ed25519 batches 9.7 M instead of 9.0 M (3 / 4 waves: 7.0 / 6.4 M); round 2, and again in round 3: 168 / 128 registers with 1.5 / 1.9 KB of scratch change nothing for RSA batches
#endif
constexpr uint32_t VERDICT_EMAILS_PER_WAVE = 16;      // a DPP quad per e-mail in the Ed25519 stage

// One wave: the Ed25519 stage and the verdicts of 16 e-mails, then — for the rare e-mail whose candidate signature failed
// while another same-domain signature is untried — the next signature round of those e-mails right here: cfdkim's
// verify_email_with_key tries the candidates one after the other until one passes (behind core/src/email.rs:31-33).
// Round 0 of every e-mail runs in the batch's three launches; giving the later rounds launches of their own would charge
// every batch for them, here they cost a ballot per wave.  The loop body is the round: Ed25519 stage (a quad per e-mail),
// verdict (a lane per e-mail), and for the e-mails still pending the next round's front end, hashes and RSA by the
// whole wave, one e-mail after the other.
__global__ __launch_bounds__(64, ZKE_VERDICT_WAVES) void ed_verdict_kernel(EdVerdictArgs A) {
  const BatchDev& B = A.fin.b;
  const int lane = threadIdx.x & 63;
  const uint32_t base = blockIdx.x * VERDICT_EMAILS_PER_WAVE;
  if (blockIdx.x == 0 && lane == 0 && A.wave_count) {
    if (A.wave_feedback) __hip_atomic_store(A.wave_feedback, *A.wave_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    *A.wave_count = 0;
  }
  if (blockIdx.x == 0 && B.order)              // the hash stage has consumed the length buckets: counters back to zero for the slot's next batch
    for (uint32_t k = 0; k < 2; k++)
      for (uint32_t c = (uint32_t)lane; c < SHA_CLASSES; c += 64) B.order[sha_order_cnt(k) + c] = 0;
  FinArgs fin = A.fin;
  uint64_t active = 0;                                  // verdict lanes (= e-mails base + lane) this wave still works on
  for (uint32_t j = 0; j < VERDICT_EMAILS_PER_WAVE; j++) if (base + j < B.n) active |= 1ull << j;
  for (;;) {
    // ---- Ed25519 stage, four lanes per e-mail (quad j = e-mail base + j): the curve-point check of every 32-byte key in
    // round 0 (DkimPublicKey::try_from_bytes, core/src/email.rs:28-29), the verification of a=ed25519-sha256 candidates
    // over the SHA-256 header hash
    uint32_t ed_res = 3;                                // 3 = no Ed25519 work for this e-mail in this round
    {
      const uint32_t j = (uint32_t)lane >> 2, i = base + j;
      bool work = false, have_sig = false;
      // (quads without work run along on bytes that are always there: the start of the result records)
      const uint8_t* key = reinterpret_cast<const uint8_t*>(B.results);
      const uint8_t* msg = key;
      const uint8_t* sig = key;
      if ((active >> j) & 1) {
        const EmailMeta* M = B.meta + i;
        if (!A.skip_ed && B.key_type[i] == ZKE_KEY_ED25519 && M->key_ok == 2) {
          const bool cand = M->state == ST_CAND && (M->flags & ZKE_F_ED25519);
          if (fin.round == 0 || cand) {               // the key itself is checked in round 0
            const RsaJob* J = B.rsa + i;
            work = true;
            have_sig = cand && J->sig_len == 64;      // a b= of any other length cannot be an Ed25519 signature
            key = B.key + B.key_off[i]; msg = B.results[i].header_hash; sig = J->sig + (512 - 64);
          }
        }
      }
      if (__ballot(work)) {
        const uint32_t r = ed25519_verify_quad(key, msg, 32, sig, have_sig);
        if (work) {
          ed_res = r;
          if ((lane & 3) == 0) {
            EmailMeta* M = B.meta + i;
            if (r == 0) M->ed_key_bad = 1;
            M->ed_ok = (r == 2) ? 1u : 0u;
          }
        }
      }
    }
    // ---- verdicts, lane per e-mail (lanes 0..15); e-mail j's Ed25519 result sits in lanes 4j..4j+3
    const uint32_t mine = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(16u * ((uint32_t)lane & 15u)), (int)ed_res);
    bool again = false;
    if (lane < (int)VERDICT_EMAILS_PER_WAVE && ((active >> lane) & 1)) {
      const uint32_t i = base + (uint32_t)lane;
      const EmailMeta* M = B.meta + i;
      const bool ed_bad = (mine == 0) || (mine == 3 && M->ed_key_bad != 0);      // decided in round 0
      bool rsa_ok = false;
      if (M->state == ST_CAND && !(M->flags & ZKE_F_ED25519) && M->em_ok) {
        // rsa 0.9.6 pkcs1v15 verify, last step: EM's trailing digest (little-endian limbs) against the header hash as stored
        const uint32_t* hw = (const uint32_t*)B.results[i].header_hash;
        const uint32_t hl4 = (M->flags & ZKE_F_SHA1) ? 5u : 8u;
        rsa_ok = true;
        for (uint32_t l = 0; l < hl4; l++) rsa_ok = rsa_ok && M->em_tail[l] == __builtin_bswap32(hw[hl4 - 1 - l]);
      }
      again = verdict_lane(fin, i, rsa_ok, mine == 2, ed_bad);
    }
    active = __ballot(again);
    if (!active) return;
    // ---- the next signature round of the e-mails still undecided (verdict_lane has checked that one is allowed)
    fin.round++;
    next_round(A, fin.round, active, base);
    wave_publish();
  }
}

}  // namespace zke
