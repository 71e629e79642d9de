// fused.hip.h — the hash / modexp stage of a batch as ONE launch.
//
// A batch's SHA-256 launch is a few long dependency chains (a 4 KB body = 65 dependent compressions: 137 us whatever
// the batch size) that leave most of the chip idle, and the RSA modular exponentiation of the same e-mails does not
// depend on any hash — only the final EMSA compare does (call site core/src/email.rs:31-33: cfdkim computes bh, the
// header hash and the signature check one after the other; nothing orders the arithmetic inside).  One stream runs
// its kernels in order, so the two can only overlap inside one launch: workgroups [0, g_sha) are SHA-256 groups
// (sha256_pair_group: two waves per 64 messages), the rest are RSA roles — one signature per wave for keys whose
// Montgomery constants are not cached yet (it fills the cache) or that the lane-group kernels do not take, four / eight
// lanes per signature (rsa_group_wave) for everything the front end routed there.  RSA leaves EM's shape verdict and its
// digest bytes in EmailMeta; verdict_kernel joins them with the hashes.  The batch's chain becomes
// front end -> max(SHA-256, RSA) -> verdict instead of front end -> SHA-256 -> RSA.
#pragma once
#include "rsa_quad.hip.h"

namespace zke {

struct StageArgs {
  const ShaJob* sha; uint32_t n_sha;       // 4 * n_pad messages, kind-major
  const RsaJob* rsa; uint32_t n;           // one job per e-mail
  EmailMeta* meta;
  KeyCacheEntry* cache;
  uint8_t* em_out;                         // parity intermediates (nullptr in production)
  const uint32_t* wave_count;              // the front end's list of jobs for the one-signature-per-wave routine (the keys
  const uint32_t* wave_list;               // not cached yet, other exponents ...); nullptr: no list, job = 2 b + wave
  uint32_t g_sha, g_wave, g_quad, g_oct;   // workgroups per role, in this order; blockDim = 128 (two waves)
  const uint32_t* order; uint32_t n_pad;   // length buckets of the body / header-preimage hashes (BatchDev::order), or nullptr
  uint32_t debug_skip_rsa;
};

constexpr uint32_t QUAD_LDS_DWORDS = 16 * (4 * QL + 4), OCT_LDS_DWORDS = 8 * (8 * QL + 4);      // per wave

#ifndef ZKE_STAGE_WAVES
#define ZKE_STAGE_WAVES 2        // waves per SIMD the hash / modexp launch is compiled for
#endif
template <int T>
__global__ __launch_bounds__(128, ZKE_STAGE_WAVES) void hash_modexp_kernel(StageArgs A) {
  static_assert(sha256_pair_lds_bytes<T>() >= 2 * 4 * QUAD_LDS_DWORDS && sha256_pair_lds_bytes<T>() >= 2 * 4 * OCT_LDS_DWORDS,
                "the RSA roles borrow the launch's LDS");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
  uint32_t b = blockIdx.x;
  if (b < A.g_sha) {
    if (!A.order) { sha256_pair_group<T>(A.sha, A.n_sha, b, lds_raw); return; }
    // kind-major job list: groups [k gpk, (k + 1) gpk) hash kind k; bodies and header preimages by length class
    const uint32_t gpk = A.n_pad / 64, kind = b / gpk, gk = b - kind * gpk;
    const ShaOrder so{kind < 2 ? A.order + sha_order_cnt(kind) : nullptr, kind < 2 ? A.order + sha_order_key(kind, A.n_pad) : nullptr, A.n};
    sha256_pair_group<T>(A.sha + (size_t)kind * A.n_pad, A.n_pad, gk, lds_raw, &so);
    return;
  }
  b -= A.g_sha;
  const uint32_t wave = threadIdx.x >> 6;
  if (b < A.g_wave) {
    if (A.wave_list) {
      // Once a batch's keys are cached this list is empty and the role's workgroups leave at once — that is why they are
      // few and loop (a workgroup that only looks and leaves still has to be given LDS and registers on a full chip first).
      const uint32_t cnt = *A.wave_count;
      for (uint32_t t = 2 * b + wave; t < cnt; t += 2 * A.g_wave)
        rsa_wave_any(A.rsa, A.wave_list[t], nullptr, 0, nullptr, A.em_out, A.cache, A.meta, A.debug_skip_rsa);
    } else {
      const uint32_t job = 2 * b + wave;
      if (job < A.n) rsa_wave_any(A.rsa, job, nullptr, 0, nullptr, A.em_out, A.cache, A.meta, A.debug_skip_rsa);
    }
    return;
  }
  b -= A.g_wave;
  uint32_t* lds = reinterpret_cast<uint32_t*>(lds_raw);
  if (b < A.g_quad) {
    if (!A.debug_skip_rsa) rsa_group_wave<4>(A.rsa, A.n, (2 * b + wave) * 16, lds + wave * QUAD_LDS_DWORDS, nullptr, 0, nullptr, A.em_out, A.cache, A.meta);
    return;
  }
  b -= A.g_quad;
  if (!A.debug_skip_rsa) rsa_group_wave<8>(A.rsa, A.n, (2 * b + wave) * 8, lds + wave * OCT_LDS_DWORDS, nullptr, 0, nullptr, A.em_out, A.cache, A.meta);
}

}  // namespace zke
