// verdict.hip.h — the last launch of a signature round: the Ed25519 stage and the verdict of every e-mail.
//
// One wave per 64 e-mails.  First, lane per e-mail, what ed25519_email_kernel did as a launch of its own: the curve-point
// check of every 32-byte Ed25519 key (round 0; DkimPublicKey::try_from_bytes, core/src/email.rs:28-29) and the Ed25519
// verification of a=ed25519-sha256 candidates over the SHA-256 header hash — waves without an Ed25519 e-mail skip it after
// one byte load per lane.  Then every lane writes its e-mail's verdict (verdict_lane, canon.hip.h): base64(body hash)
// against bh=, the digest bytes the RSA role left in EmailMeta against the header hash (rsa 0.9.6 pkcs1v15 verify),
// status / detail, the pending counter of the next signature round.  One launch instead of two: with many batches in
// flight every launch of a batch costs the chip's command processor several microseconds whatever it does.
#pragma once
#include "ed25519.hip.h"
#include "rsa_kernel.hip.h"

namespace zke {

struct EdVerdictArgs { FinArgs fin; uint32_t skip_ed; uint32_t* wave_count; };   // wave_count: the hash / modexp stage's job list is consumed — reset for the next round

__global__ __launch_bounds__(64) void ed_verdict_kernel(EdVerdictArgs A) {
  const BatchDev& B = A.fin.b;
  const int lane = threadIdx.x & 63;
  const uint32_t base = blockIdx.x * 64, i = base + (uint32_t)lane;
  if (blockIdx.x == 0 && lane == 0 && A.wave_count) *A.wave_count = 0;
  // ---- Ed25519 stage, lane per e-mail; results stay in registers (ed_ok, ed_bad) and go to EmailMeta for later rounds
  uint32_t ed_ok = 0, ed_bad = 0;
  if (i < B.n) {
    EmailMeta* M = B.meta + i;
    ed_bad = M->ed_key_bad;                          // decided in round 0
    if (!A.skip_ed && B.key_type[i] == ZKE_KEY_ED25519 && M->key_ok == 2) {
      const bool cand = M->state == ST_CAND && (M->flags & ZKE_F_ED25519);
      if (A.fin.round == 0 || cand) {                // the key itself is checked in round 0
        const RsaJob* J = B.rsa + i;
        const zke_result* R = B.results + i;
        const bool have_sig = cand && J->sig_len == 64;  // a b= of any other length cannot be an Ed25519 signature
        const uint32_t r = ed25519_verify_lane(B.key + B.key_off[i], R->header_hash, 32, J->sig + (512 - 64), have_sig);
        if (r == 0) { ed_bad = 1; M->ed_key_bad = 1; }
        ed_ok = (r == 2) ? 1u : 0u;
        M->ed_ok = ed_ok;
      }
    }
  }
  // ---- verdicts, lane per e-mail
  if (i < B.n) {
    const EmailMeta* M = B.meta + i;
    bool rsa_ok = false;
    if (M->state == ST_CAND && !(M->flags & ZKE_F_ED25519) && M->em_ok) {
      // rsa 0.9.6 pkcs1v15 verify, last step: EM's trailing digest (little-endian limbs) against the header hash as stored
      const uint32_t* hw = (const uint32_t*)B.results[i].header_hash;
      const uint32_t hl4 = (M->flags & ZKE_F_SHA1) ? 5u : 8u;
      rsa_ok = true;
      for (uint32_t l = 0; l < hl4; l++) rsa_ok = rsa_ok && M->em_tail[l] == __builtin_bswap32(hw[hl4 - 1 - l]);
    }
    verdict_lane(A.fin, i, rsa_ok, ed_ok != 0, ed_bad != 0);
  }
}

}  // namespace zke
