// engine.hip — the C-ABI of include/zkemail_amd.h (ABI 0.3) over the HIP kernels in this directory.
//
// Host side: submission slots and their workspaces, kernel launches, the pinned staging of the host entry point, DFA
// registration.  No verification arithmetic runs on the host and there is no CPU fallback: without a HIP device every
// entry point returns ZKE_E_DEVICE.  Nothing runs at load time and the process environment is touched in one documented
// place (zke_process_init).  Experiment knobs exist only in -DZKE_DEV_KNOBS builds (tools/build_variant.sh).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "zkemail_amd.h"
#include "sha256.hip.h"
#include "rsa.hip.h"
#include "parse.hip.h"
#include "regex.hip.h"
#include "rsa_kernel.hip.h"
#include "rsa_quad.hip.h"
#include "fused.hip.h"
#include "ed25519.hip.h"
#include "verdict.hip.h"

using namespace zke;

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  uint64_t* generation = nullptr;      // bumped on every (re)allocation, when the owner wants to know
  int ensure(size_t need) {
    if (need <= cap) return 0;
    if (generation) ++*generation;
    if (p) (void)hipFree(p);
    p = nullptr;
    size_t want = std::max(need + need / 4, (size_t)4096);
    cap = 0;
    if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return ZKE_E_NOMEM; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline uint32_t e_k(uint32_t bits) { return (bits + 7) / 8; }
constexpr int SHA_TILE = 128;                      // bytes of a message per LDS tile (sha256.hip.h)
constexpr uint32_t SHA_PAIR_MAX_GROUPS = 512;      // launches of up to 32 768 messages use two waves per 64 messages

}  // namespace

#include "dfa_registry.hip.h"
#include "host_stage.hip.h"

// Timing marks of one batch (HIP events on the stream it ran on; zke_set_timing).
enum : int { MK_START = 0, MK_H2D, MK_FRONT, MK_HASH, MK_VERDICT, MK_PREP, MK_DFA, MK_D2H, MK_N };

// One submission slot: a stream, a private workspace and (host entry) a pinned staging image.  A batch runs in one slot
// from its first copy to its last; `slots` batches can be in flight on one engine (zke_engine_reserve).  What batches share
// lives in the engine: the per-key Montgomery constants (one cache per device) and the registered DFA tables.  Everything a
// call needs while it enqueues lives here, under `mu`: the entry points are re-entrant (include/zkemail_amd.h, "Threading").
struct Slot {
  std::mutex mu;                       // held by the call that is enqueueing into / retiring from this slot
  hipStream_t stream = nullptr;        // the slot's own stream: batches submitted with stream == NULL run here
  hipStream_t last_stream = nullptr;   // stream of the slot's previous batch (nullptr: the slot has not been used)
  hipEvent_t done = nullptr;           // recorded behind the slot's last batch; waited for when the stream changes
  hipEvent_t ev[MK_N]{};               // timing marks
  uint32_t marks = 0;                  // bit k: ev[k] was recorded for the slot's last timed batch
  zke_timings last{};                  // ... and what they said, once read
  DevBuf meta, rsa_jobs, sha_jobs, sha_order, rsa_ok, em_dbg, scratch_off, scratch, clean, meta2, scratch2, parts;
  DevBuf pending;  // device counters: e-mails that need another signature round, the wave-routine job list's length
  uint32_t* wave_feedback = nullptr;   // pinned host word the verdict launch leaves that length in: the slot's next batch sizes the
                                       // one-signature-per-wave role of its hash / modexp launch by it (0xFFFFFFFF: nothing known yet)
  // host entry: the packed input image (pinned + HBM), the records (HBM + pinned), the batch not yet delivered
  PinnedBuf h_image, h_results;
  DevBuf d_image, d_results;
  hipEvent_t host_done = nullptr;      // recorded behind the D2H of the records
  hipEvent_t h2d_done = nullptr;       // recorded behind the slot's input image on the engine's copy stream
  std::vector<CopyPool::Piece> gather; // zke_verify_emails: the 3 n pieces of the batch being packed (kept: no allocation per call)
  uint64_t host_gen = 0;               // host batches submitted to this slot
  uint64_t host_retired = 0;           // ... of which this many have been delivered to their caller's `out`
  zke_result* host_out = nullptr;      // where the pending batch's records go
  uint32_t host_n = 0;
  DevBuf* all[15] = {&meta, &rsa_jobs, &sha_jobs, &sha_order, &rsa_ok, &em_dbg, &scratch_off, &scratch, &clean, &meta2, &scratch2, &parts,
                     &pending, &d_image, &d_results};
  // hipGraph replay of a batch's kernel sequence (zke_options.replay_graphs).  The graph holds this slot's
  // workspace pointers, so it is valid only while none of them has been reallocated: `generation` counts reallocations.
  hipGraphExec_t graph_exec = nullptr;
  std::vector<uint8_t> graph_key;      // everything the captured launches depend on, byte for byte
  uint64_t generation = 0;
};

struct zke_engine {
  int device = 0;
  zke_options opt{};                // as given, defaults filled in
  uint32_t strict = 0;              // ZKE_STRICT_* from opt
  // Locks, outermost first: `big` (shared: a submission; exclusive: reserve / unregister / evict / destroy), a slot's `mu`,
  // then `reg_mu` (the registry) or `misc_mu` (the building blocks' scratch).
  std::shared_mutex big;
  std::vector<Slot*> slots;         // at least one; grows only under `big` exclusive
  std::atomic<uint32_t> ticket{0};  // round-robin cursor of the submission entry points
  std::atomic<uint32_t> last_slot{0};
  std::atomic<bool> timing{false};
  std::atomic<size_t> host_image_cap{0};   // every slot's pinned staging holds an input image of this many bytes ...
  std::atomic<uint32_t> host_n_cap{0};     // ... and this many records (grow_host_staging: all slots at once, in a quiet moment)
  hipStream_t stream = nullptr;     // = slots[0]->stream: the building-block entry points
#ifndef ZKE_COPY_STREAMS
#define ZKE_COPY_STREAMS 2
#endif
  hipStream_t copy_stream[ZKE_COPY_STREAMS]{};   // host batches' input images cross PCIe on these streams, in turn (pipeline.hip.h).  Created
                                                 // with the host entry's first batch, not with the engine: every stream takes part in HIP's
                                                 // mapping of streams onto hardware queues, and an engine that only ever sees device-resident
                                                 // batches must not pay for two it never uses (22 slots + 2: 23 M e-mails/s instead of 31 M)
  std::mutex copy_mu;
  uint32_t copy_turn = 0;
  std::mutex misc_mu;
  DevBuf misc;                      // building-block entry points
  DevBuf key_cache;                 // KeyCacheEntry[KEY_CACHE_SLOTS]: per-key Montgomery constants, kept across batches, shared by the slots
  std::shared_mutex reg_mu;
  std::vector<RegisteredDfa*> dfas; // index = id; nullptr = unregistered
  std::unordered_multimap<uint64_t, uint32_t> dfa_index;   // pair_hash -> id
  uint32_t dfa_live = 0;
  std::atomic<uint64_t> reg_clock{0};
  size_t dfa_lds_attr = 0, dfa_wave_lds_attr = 0;
  CopyPool* pool = nullptr;
  uint32_t rsa_quad_min = 256;      // lane-group RSA routines join batches of at least this many e-mails (four lanes; moduli <= 2048 bits).
                                    // Low since the modexp runs beside SHA-256 (fused.hip.h): its longer chain no longer sits behind the
                                    // hashes of a small batch; a handful of e-mails is still answered sooner by one signature per wave
  uint32_t rsa_oct_min = 128;       // ... eight lanes (moduli above 2048 bits)
#ifdef ZKE_DEV_KNOBS
  uint32_t debug_skip_rsa = 0, debug_skip_ed = 0, debug_parse_stop = 0, debug_skip_launch = 0;
#else
  static constexpr uint32_t debug_skip_rsa = 0, debug_skip_ed = 0, debug_parse_stop = 0, debug_skip_launch = 0;
#endif
};

namespace {

thread_local std::string g_err;      // zke_last_error: the calling thread's last failure

int fail(zke_engine* e, int code, const char* what, hipError_t he = hipSuccess) {
  (void)e;
  g_err = what;
  if (he != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(he); }
  return code;
}
#define HIPCHK(e, call) do { hipError_t _r = (call); if (_r != hipSuccess) return fail((e), ZKE_E_DEVICE, #call, _r); } while (0)

int launch_sha(zke_engine* e, const ShaJob* jobs, uint32_t n, hipStream_t s) {
  if (n == 0) return 0;
  // Few messages: the launch is as long as one wave's chain of compressions, so split the chain over two waves
  // (sha256_pair_kernel).  Many messages: the chip is full and the one-wave kernel does less LDS work per byte.
  const uint32_t groups = (n + 63) / 64;
  if (groups <= SHA_PAIR_MAX_GROUPS) {
    hipLaunchKernelGGL(sha256_pair_kernel<SHA_TILE>, dim3(groups), dim3(128), sha256_pair_lds_bytes<SHA_TILE>(), s, jobs, n);
  } else {
    hipLaunchKernelGGL(sha256_batch_kernel<SHA_TILE>, dim3((n + 255) / 256), dim3(256), sha256_lds_bytes<SHA_TILE>(), s, jobs, n);
  }
  HIPCHK(e, hipGetLastError());
  return 0;
}

Slot* new_slot(zke_engine* e) {
  (void)e;
  Slot* w = new Slot();
  bool ok = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&w->done, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&w->host_done, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&w->h2d_done, hipEventDisableTiming) == hipSuccess;
  for (auto& ev : w->ev) ok = ok && hipEventCreate(&ev) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&w->wave_feedback, 64, hipHostMallocMapped) == hipSuccess;
  if (ok) *w->wave_feedback = 0xFFFFFFFFu;
  if (!ok) { g_err = "slot stream / event creation"; delete w; return nullptr; }
  for (auto* b : w->all) b->generation = &w->generation;
  return w;
}
void free_slot(Slot* w) {
  if (!w) return;
  if (w->stream) (void)hipStreamSynchronize(w->stream);
  if (w->graph_exec) (void)hipGraphExecDestroy(w->graph_exec);
  for (auto* b : w->all) b->release();
  w->h_image.release(); w->h_results.release();
  for (auto& ev : w->ev) if (ev) (void)hipEventDestroy(ev);
  if (w->done) (void)hipEventDestroy(w->done);
  if (w->host_done) (void)hipEventDestroy(w->host_done);
  if (w->h2d_done) (void)hipEventDestroy(w->h2d_done);
  if (w->wave_feedback) (void)hipHostFree(w->wave_feedback);
  if (w->stream) (void)hipStreamDestroy(w->stream);
  delete w;
}

// Which lane-group RSA kernels take part in a batch of n e-mails.  Bit 0: four lanes per signature (moduli <= 2048 bits),
// bit 1: eight lanes (2049..4096 bits; any_big = the caller's hint that the batch's keys average more than an RSA-2048 key).
// The front end routes signatures by this mask and the hash / modexp launch gets the matching workgroups: the same value
// must go to both.  zke_options.rsa_lane_groups: 0 = by batch size, 1 never, 2 always.
uint32_t rsa_route_mask(const zke_engine* e, uint32_t n, bool any_big) {
  if (!e->key_cache.p || e->opt.rsa_lane_groups == 1) return 0;
  const bool always = e->opt.rsa_lane_groups == 2;
  uint32_t m = 0;
  if (always || n >= e->rsa_quad_min) m |= 1u;
  if (any_big && (always || n >= e->rsa_oct_min)) m |= 2u;
  return m;
}

int launch_stage(zke_engine* e, const StageArgs& A, hipStream_t s) {
  const uint32_t grid = A.g_sha + A.g_wave + A.g_quad + A.g_oct;
  if (!grid) return 0;
  hipLaunchKernelGGL(hash_modexp_kernel<SHA_TILE>, dim3(grid), dim3(128), sha256_pair_lds_bytes<SHA_TILE>(), s, A);
  HIPCHK(e, hipGetLastError());
  return 0;
}

// The hash / modexp stage of a batch (fused.hip.h): SHA-256 of the 4 * n_pad messages and the RSA operation of the n jobs.
// Launches of up to SHA_PAIR_MAX_GROUPS SHA-256 groups (every BASELINE-sized batch) are ONE kernel; beyond that the chip
// is full of SHA-256 waves anyway: sha256_batch_kernel first, then the RSA roles as a launch of their own.
// The one-signature-per-wave role walks the front end's job list: the e-mails whose key is not cached yet (or that the
// lane-group routines do not take).  Once a stream of batches has its keys cached the list is empty, and every workgroup of the
// role still has to be given LDS and registers on a full chip just to look and leave (57 % of the launch's waves did nothing
// else).  So the role is sized by what the slot's PREVIOUS batch needed — its verdict launch leaves the list's length in a
// pinned word, no synchronisation, one batch stale —: 8 workgroups when that was empty, up to 128 (256 waves: a batch of 1 024
// uncached keys takes four rounds of them) when every key was new.  The walkers loop, so a wrong guess costs time, never work.
#ifndef ZKE_WAVE_ROLE_MAX_GROUPS
#define ZKE_WAVE_ROLE_MAX_GROUPS 128
#endif
#ifndef ZKE_WAVE_ROLE_MIN_GROUPS
#define ZKE_WAVE_ROLE_MIN_GROUPS 8
#endif
int launch_hash_modexp(zke_engine* e, const ShaJob* sha, uint32_t n_sha, const RsaJob* rsa, uint32_t n, EmailMeta* meta,
                       uint8_t* em_out, uint32_t route_mask, const uint32_t* wave_count, const uint32_t* wave_list, const uint32_t* order,
                       uint32_t last_wave_jobs, hipStream_t s) {
  StageArgs A{};
  A.order = order; A.n_pad = n_sha / 4;
  A.sha = sha; A.n_sha = n_sha; A.rsa = rsa; A.n = n; A.meta = meta;
  A.cache = e->key_cache.as<KeyCacheEntry>();
  A.em_out = em_out;
  A.wave_count = wave_count; A.wave_list = wave_list;
  A.g_wave = (n + 1) / 2;
  if (wave_list) {
    const uint32_t want = last_wave_jobs == 0xFFFFFFFFu ? ZKE_WAVE_ROLE_MAX_GROUPS
                                                        : std::max<uint32_t>(ZKE_WAVE_ROLE_MIN_GROUPS, (std::min<uint32_t>(last_wave_jobs, n) + 1) / 2);
    A.g_wave = std::min<uint32_t>(A.g_wave, std::min<uint32_t>(want, ZKE_WAVE_ROLE_MAX_GROUPS));
  }
  A.g_quad = (route_mask & 1u) ? (n + 31) / 32 : 0;
  A.g_oct = (route_mask & 2u) ? (n + 15) / 16 : 0;
  A.debug_skip_rsa = e->debug_skip_rsa;
  const uint32_t groups = (n_sha + 63) / 64;
  if (groups <= SHA_PAIR_MAX_GROUPS) {
    A.g_sha = groups;
    return launch_stage(e, A, s);
  }
  if (int r = launch_sha(e, sha, n_sha, s)) return r;      // (beyond 512 groups: arrival order; the chip is full either way)
  A.g_sha = 0;
  return launch_stage(e, A, s);
}

// Every hipFuncSetAttribute the pipeline needs, once per engine (= per device) at creation — never lazily in the submit
// path, where a first use would land inside somebody's timed region.  The DFA kernels' LDS sizes depend on the registered
// tables: zke_dfa_register raises them.
int set_kernel_attrs(zke_engine* e) {
  HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&sha256_batch_kernel<SHA_TILE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sha256_lds_bytes<SHA_TILE>()));
  HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&sha256_pair_kernel<SHA_TILE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sha256_pair_lds_bytes<SHA_TILE>()));
  HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_modexp_kernel<SHA_TILE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sha256_pair_lds_bytes<SHA_TILE>()));
  return 0;
}

}  // namespace

#include "pipeline.hip.h"
#include "wire.hip.h"

extern "C" {

const char* zke_version(void) { return "zkemail.rs_amd 0.3 (gfx950)"; }
uint32_t zke_abi_version(void) { return 3; }

const char* zke_status_name(uint32_t status) {
  static const char* const names[] = {
      "",
      "mailparse::parse_mail(..).unwrap()  core/src/email.rs:26",
      "DkimPublicKey::try_from_bytes(..).unwrap()  core/src/email.rs:29",
      "verify_email_with_key(..).unwrap()  core/src/email.rs:33",
      "assert!(verified)  core/src/circuits.rs:13",
      ".expect(\"Value cannot be null\")  core/src/circuits.rs:24",
      "canonicalize_signed_email(..).unwrap()  core/src/circuits.rs:35",
      "dense::DFA::from_bytes(..).unwrap()  core/src/regex.rs:32-33",
      "assert!(verified) on header parts  core/src/circuits.rs:45",
      "assert!(verified) on body parts  core/src/circuits.rs:54",
      "input outside what this engine implements (ZKE_UNSUPPORTED; detail says which limit)",
  };
  return status < sizeof names / sizeof names[0] ? names[status] : "unknown status";
}

// ---- Solidity ABI encoding of the outputs (core/src/io.rs:5-53; alloy-sol-types' SolValue::abi_encode = abi.encode(value)).
// The struct is a dynamic type: 32-byte offset 0x20, then the tuple's head / tail.  A string[] is its length, one offset
// per element (relative to the start of the offsets), then each string as length + bytes padded to 32.
namespace {
struct AbiWriter {
  uint8_t* out; size_t cap, len = 0;
  void word(uint64_t v) { if (out && len + 32 <= cap) { memset(out + len, 0, 24); for (int i = 0; i < 8; i++) out[len + 24 + i] = (uint8_t)(v >> (56 - 8 * i)); } len += 32; }
  void bytes(const uint8_t* p, size_t n) {
    const size_t padded = (n + 31) & ~(size_t)31;
    if (out && len + padded <= cap) { if (n) memcpy(out + len, p, n); memset(out + len + n, 0, padded - n); }
    len += padded;
  }
};
size_t abi_string_array_size(const size_t* lens, uint32_t n) {
  size_t s = 32 + 32 * (size_t)n;
  for (uint32_t i = 0; i < n; i++) s += 32 + ((lens[i] + 31) & ~(size_t)31);
  return s;
}
void abi_string_array(AbiWriter& w, const uint8_t* const* strs, const size_t* lens, uint32_t n) {
  w.word(n);
  size_t off = 32 * (size_t)n;
  for (uint32_t i = 0; i < n; i++) { w.word(off); off += 32 + ((lens[i] + 31) & ~(size_t)31); }
  for (uint32_t i = 0; i < n; i++) { w.word(lens[i]); w.bytes(strs[i], lens[i]); }
}
}  // namespace

int zke_abi_encode(const uint8_t* from_domain_hash, const uint8_t* public_key_hash, const uint8_t* const* external_inputs,
                   const size_t* external_input_lens, uint32_t n_external_inputs, uint32_t with_matches,
                   const uint8_t* const* matches, const size_t* match_lens, uint32_t n_matches, uint8_t* out, size_t out_cap,
                   size_t* out_len) {
  if (!from_domain_hash || !public_key_hash || !out_len || (n_external_inputs && (!external_inputs || !external_input_lens)) ||
      (with_matches && n_matches && (!matches || !match_lens)) || (out_cap && !out))
    return ZKE_E_ARG;
  for (uint32_t i = 0; i < n_external_inputs; i++) if (external_input_lens[i] && !external_inputs[i]) return ZKE_E_ARG;
  for (uint32_t i = 0; with_matches && i < n_matches; i++) if (match_lens[i] && !matches[i]) return ZKE_E_ARG;
  const size_t email_tuple = 32 + 32 + 32 + abi_string_array_size(external_input_lens, n_external_inputs);
  const size_t need = with_matches ? 32 + 64 + email_tuple + abi_string_array_size(match_lens, n_matches) : 32 + email_tuple;
  *out_len = need;
  if (need > out_cap) return out_cap == 0 && !out ? 0 : ZKE_E_NOMEM;
  AbiWriter w{out, out_cap};
  w.word(0x20);
  if (with_matches) { w.word(0x40); w.word(0x40 + email_tuple); }
  w.bytes(from_domain_hash, 32);
  w.bytes(public_key_hash, 32);
  w.word(0x60);
  abi_string_array(w, external_inputs, external_input_lens, n_external_inputs);
  if (with_matches) abi_string_array(w, matches, match_lens, n_matches);
  return w.len == need ? 0 : ZKE_E_ARG;
}

int zke_shard_bounds(const uint64_t* raw_off, uint32_t n, uint32_t world, uint32_t* bounds) {
  if (!bounds || world == 0 || (n && !raw_off)) return ZKE_E_ARG;
  bounds[0] = 0;
  const uint64_t base = n ? raw_off[0] : 0, total = n ? raw_off[n] - base : 0;
  uint32_t cut = 0;
  for (uint32_t r = 1; r < world; r++) {
    // the first e-mail index whose cumulative byte count reaches total * r / world (compared without a division)
    while (cut < n && (unsigned __int128)(raw_off[cut] - base) * world < (unsigned __int128)total * r) cut++;
    bounds[r] = cut;
  }
  bounds[world] = n;
  return 0;
}

int zke_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n > 0 ? 1 : 0;
}

// HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when the runtime
// initialises.  22 submission slots + the null stream = 23 — and no more: the chip runs 24 queues of a process without
// time-slicing them, the 25th costs a factor of ten, and with the pool capped at 23 a stream the process creates on top (a
// communicator's, a framework's) shares a queue instead of adding one; DESIGN.md §5, INTEGRATION.md §5.  An explicit call, never
// a load-time side effect: the variable is left alone when the host has set it.
int zke_process_init(uint32_t hw_queues) {
  char buf[16];
  snprintf(buf, sizeof buf, "%u", hw_queues ? std::min<uint32_t>(hw_queues, 64) : 23u);
  return setenv("GPU_MAX_HW_QUEUES", buf, 0) == 0 ? 0 : ZKE_E_ARG;
}

int zke_engine_create(const zke_options* opt, zke_engine** out) {
  if (!out) return ZKE_E_ARG;
  *out = nullptr;
  (void)zke_process_init(0);          // before this process's first HIP call, if that is this one (include/zkemail_amd.h)
  if (!zke_device_available()) return fail(nullptr, ZKE_E_DEVICE, "no HIP device");
  zke_engine* e = new zke_engine();
  if (opt) e->opt = *opt; else e->opt.device = -1;
  zke_options& o = e->opt;
  if (o.slots == 0) o.slots = 1;
  if (o.slots > 64 || o.rsa_lane_groups > 2 || o.dfa_mapping > 2) { delete e; return fail(nullptr, ZKE_E_ARG, "zke_options: field out of range"); }
  o.max_sig_rounds = o.max_sig_rounds ? std::min<uint32_t>(o.max_sig_rounds, ZKE_MAX_HEADERS) : 16u;
  if (o.host_threads == 0) o.host_threads = 4;
  o.host_threads = std::min<uint32_t>(o.host_threads, 64);
  o.max_dfas = o.max_dfas ? std::max<uint32_t>(o.max_dfas, 64) : 4096u;
  e->strict = (o.enforce_expiry_x ? ZKE_STRICT_EXPIRY_X : 0u) | (o.canon_takes_verified_signature ? ZKE_STRICT_CANON_VERIFIED : 0u) |
              (o.canon_ignores_l ? ZKE_STRICT_CANON_IGNORES_L : 0u) | (o.i_must_be_subdomain ? ZKE_STRICT_I_SUBDOMAIN : 0u) |
              (o.b_removes_own_span_only ? ZKE_STRICT_B_OWN_SPAN : 0u);
  int dev = o.device;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  e->device = dev;
  if (hipSetDevice(dev) != hipSuccess) { delete e; return fail(nullptr, ZKE_E_DEVICE, "hipSetDevice"); }
  for (uint32_t k = 0; k < o.slots; k++) {
    Slot* w = new_slot(e);
    if (!w) { zke_engine_destroy(e); return ZKE_E_DEVICE; }
    e->slots.push_back(w);
  }
  e->stream = e->slots[0]->stream;
  if (!o.disable_key_cache) {
    const size_t kc_bytes = (size_t)KEY_CACHE_SLOTS * sizeof(KeyCacheEntry);
    if (e->key_cache.ensure(kc_bytes) || hipMemset(e->key_cache.p, 0, kc_bytes) != hipSuccess) {
      zke_engine_destroy(e);
      return fail(nullptr, ZKE_E_NOMEM, "key cache allocation");
    }
  }
  if (o.host_threads > 1) e->pool = new CopyPool(o.host_threads - 1);     // the caller's thread is one of them
#ifdef ZKE_DEV_KNOBS
  if (const char* rm = getenv("ZKE_RSA_QUAD_MIN")) e->rsa_quad_min = (uint32_t)atoi(rm);
  if (const char* ro = getenv("ZKE_RSA_OCT_MIN")) e->rsa_oct_min = (uint32_t)atoi(ro);
  if (getenv("ZKE_DEBUG_SKIP_RSA")) e->debug_skip_rsa = 1;              // ablation experiments: results are then meaningless
  if (getenv("ZKE_DEBUG_SKIP_ED")) e->debug_skip_ed = 1;
  if (const char* ds = getenv("ZKE_DEBUG_PARSE_STOP")) e->debug_parse_stop = (uint32_t)atoi(ds);
  if (const char* sl = getenv("ZKE_DEBUG_SKIP_LAUNCH")) e->debug_skip_launch = (uint32_t)atoi(sl);
#endif
  if (set_kernel_attrs(e)) { zke_engine_destroy(e); return ZKE_E_DEVICE; }
  *out = e;
  return 0;
}

void zke_engine_destroy(zke_engine* e) {
  if (!e) return;
  {
    std::unique_lock<std::shared_mutex> ex(e->big);
    (void)hipSetDevice(e->device);
    for (Slot* w : e->slots) { std::lock_guard<std::mutex> g(w->mu); (void)retire_host(e, *w); }     // deliver what was never waited for
    for (Slot* w : e->slots) if (w->stream) (void)hipStreamSynchronize(w->stream);
    for (Slot* w : e->slots) free_slot(w);
    e->slots.clear();
    for (auto& cs : e->copy_stream) if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); cs = nullptr; }
    e->misc.release(); e->key_cache.release();
    for (auto* d : e->dfas) if (d) { d->blob.release(); d->dev.release(); delete d; }
    e->dfas.clear();
    delete e->pool;
    e->pool = nullptr;
  }
  delete e;
}

const char* zke_last_error(const zke_engine* e) { (void)e; return g_err.c_str(); }

int zke_engine_join(zke_engine* e, void* stream) {
  if (!e) return ZKE_E_ARG;
  std::shared_lock<std::shared_mutex> sh(e->big);
  HIPCHK(e, hipSetDevice(e->device));
  hipStream_t s = (hipStream_t)stream;
  for (Slot* w : e->slots) {
    std::lock_guard<std::mutex> g(w->mu);
    if (!w->last_stream) continue;                       // never used
    if (w->last_stream == w->stream) {
      if (s == w->stream) continue;                      // stream order
      // batches on a slot's own stream leave no event behind (release_slot): record it now, behind the last one
      HIPCHK(e, hipEventRecord(w->done, w->stream));
    } else if (w->last_stream == s) {
      continue;                                          // stream order
    }
    HIPCHK(e, hipStreamWaitEvent(s, w->done, 0));
  }
  return 0;
}

int zke_engine_sync(zke_engine* e) {
  if (!e) return ZKE_E_ARG;
  std::shared_lock<std::shared_mutex> sh(e->big);
  HIPCHK(e, hipSetDevice(e->device));
  for (Slot* w : e->slots) {
    // a slot's last batch ran on the slot's own stream, or on a caller's stream with `done` recorded behind it
    hipStream_t own; hipEvent_t done; bool foreign;
    { std::lock_guard<std::mutex> g(w->mu); own = w->stream; done = w->done; foreign = w->last_stream && w->last_stream != w->stream; }
    if (foreign) HIPCHK(e, hipEventSynchronize(done));
    HIPCHK(e, hipStreamSynchronize(own));
  }
  return 0;
}

int zke_set_timing(zke_engine* e, int enabled) {
  if (!e) return ZKE_E_ARG;
  e->timing = enabled != 0;
  return 0;
}

int zke_get_slot_timings(zke_engine* e, uint32_t slot, zke_timings* t) {
  if (!e || !t) return ZKE_E_ARG;
  std::shared_lock<std::shared_mutex> sh(e->big);
  if (slot >= e->slots.size()) return fail(e, ZKE_E_ARG, "zke_get_slot_timings: no such slot");
  Slot& w = *e->slots[slot];
  std::lock_guard<std::mutex> g(w.mu);
  if (w.marks) {                            // the events are read once their stream has drained
    HIPCHK(e, hipSetDevice(e->device));
    int lastk = 0;
    for (int k = 0; k < MK_N; k++) if (w.marks & (1u << k)) lastk = k;
    HIPCHK(e, hipEventSynchronize(w.ev[lastk]));
    collect_timings(w);
  }
  *t = w.last;                              // all zeros for a slot that has not run a timed batch
  return 0;
}

int zke_get_timings(zke_engine* e, zke_timings* t) {
  if (!e) return ZKE_E_ARG;
  return zke_get_slot_timings(e, e->last_slot.load(), t);
}

// ---------------------------------------------------------------- building blocks
// (The building blocks share the engine's `misc` buffer and slot 0's stream: one at a time, under misc_mu.  The device entry
// enqueues and returns; its job list stays in `misc` until the launch has run, so a caller that overlaps two of them on
// different streams must order them itself.)
namespace {
int sha256_batch_device_locked(zke_engine* e, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n, uint8_t* digests_dev, hipStream_t s) {
  if (int r = e->misc.ensure((size_t)n * sizeof(ShaJob))) return fail(e, r, "workspace");
  hipLaunchKernelGGL(sha_jobs_from_csr_kernel, dim3((n + 255) / 256), dim3(256), 0, s, blob_dev, off_dev, n, digests_dev,
                     e->misc.as<ShaJob>());
  HIPCHK(e, hipGetLastError());
  return launch_sha(e, e->misc.as<ShaJob>(), n, s);
}
}  // namespace

int zke_sha256_batch_device(zke_engine* e, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                            uint8_t* digests_dev, void* stream) {
  if (!e) return ZKE_E_ARG;
  if (n == 0) return 0;
  std::shared_lock<std::shared_mutex> sh(e->big);
  std::lock_guard<std::mutex> g(e->misc_mu);
  HIPCHK(e, hipSetDevice(e->device));
  return sha256_batch_device_locked(e, blob_dev, off_dev, n, digests_dev, stream ? (hipStream_t)stream : e->stream);
}

int zke_sha256_batch(zke_engine* e, const uint8_t* blob, const uint64_t* off, uint32_t n, uint8_t* digests) {
  if (!e || (n && (!blob || !off || !digests))) return ZKE_E_ARG;
  if (n == 0) return 0;
  std::shared_lock<std::shared_mutex> sh(e->big);
  std::lock_guard<std::mutex> g(e->misc_mu);
  HIPCHK(e, hipSetDevice(e->device));
  const size_t total = (size_t)off[n];
  DevBuf dblob, doff, ddig;
  int r = 0;
  if ((r = dblob.ensure(total + 16)) || (r = doff.ensure((size_t)(n + 1) * 8)) || (r = ddig.ensure((size_t)n * 32))) {
    dblob.release(); doff.release(); ddig.release();
    return fail(e, r, "hipMalloc");
  }
  hipError_t he = hipSuccess;
  if (total) he = hipMemcpyAsync(dblob.p, blob, total, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(doff.p, off, (size_t)(n + 1) * 8, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) {
    r = sha256_batch_device_locked(e, dblob.as<uint8_t>(), doff.as<uint64_t>(), n, ddig.as<uint8_t>(), e->stream);
    if (r == 0) he = hipMemcpyAsync(digests, ddig.p, (size_t)n * 32, hipMemcpyDeviceToHost, e->stream);
  }
  hipError_t hs = hipStreamSynchronize(e->stream);
  dblob.release(); doff.release(); ddig.release();
  if (r) return r;
  if (he != hipSuccess) return fail(e, ZKE_E_DEVICE, "sha256 batch copy", he);
  if (hs != hipSuccess) return fail(e, ZKE_E_DEVICE, "sha256 batch sync", hs);
  return 0;
}

int zke_rsa_modexp_batch(zke_engine* e, const uint8_t* sig, const uint8_t* mod, const uint64_t* exp, uint32_t bytes,
                         uint32_t n, uint8_t* em, uint8_t* ok) {
  if (!e || bytes == 0 || bytes > ZKE_MAX_RSA_BYTES || (n && (!sig || !mod || !exp || !em || !ok))) return ZKE_E_ARG;
  if (n == 0) return 0;
  std::shared_lock<std::shared_mutex> sh(e->big);
  std::lock_guard<std::mutex> g(e->misc_mu);
  HIPCHK(e, hipSetDevice(e->device));
  std::vector<RsaJob> jobs(n);
  for (uint32_t i = 0; i < n; i++) {
    RsaJob& j = jobs[i];
    memset(&j, 0, sizeof j);
    const uint8_t* m = mod + (size_t)i * bytes;
    const uint8_t* s = sig + (size_t)i * bytes;
    memcpy(j.mod + 512 - bytes, m, bytes);
    memcpy(j.sig + 512 - bytes, s, bytes);
    uint32_t lead = 0;
    while (lead < bytes && m[lead] == 0) lead++;
    j.k = bytes - lead;
    uint32_t bits = j.k * 8;
    if (j.k) for (uint8_t t = m[lead]; !(t & 0x80); t <<= 1) bits--;
    j.bits = bits;
    j.e = exp[i];
    j.sig_len = j.k;
    j.flags = RSA_F_ACTIVE;
    ok[i] = (uint8_t)((m[bytes - 1] & 1) && bits >= 2 && memcmp(s, m, bytes) < 0);
  }
  DevBuf dj, dok, dem, dh;
  int r = 0;
  if ((r = dj.ensure(jobs.size() * sizeof(RsaJob))) || (r = dok.ensure((size_t)n * 4)) || (r = dem.ensure((size_t)n * 512)) ||
      (r = dh.ensure((size_t)n * 32))) {
    dj.release(); dok.release(); dem.release(); dh.release();
    return fail(e, r, "hipMalloc");
  }
  hipError_t he = hipMemcpyAsync(dj.p, jobs.data(), jobs.size() * sizeof(RsaJob), hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) he = hipMemsetAsync(dh.p, 0, (size_t)n * 32, e->stream);
  if (he == hipSuccess) he = hipMemsetAsync(dem.p, 0, (size_t)n * 512, e->stream);
  if (he == hipSuccess) {       // one signature per wave, no key cache, digest slots of zeros: only EM is of interest
    hipLaunchKernelGGL(rsa_verify_kernel, dim3(n), dim3(64), 0, e->stream, dj.as<RsaJob>(), n, dh.as<uint8_t>(), (size_t)32,
                       dok.as<uint32_t>(), dem.as<uint8_t>(), (KeyCacheEntry*)nullptr, (EmailMeta*)nullptr, 0u);
    he = hipGetLastError();
  }
  std::vector<uint8_t> emh((size_t)n * 512);
  if (he == hipSuccess && r == 0) he = hipMemcpyAsync(emh.data(), dem.p, emh.size(), hipMemcpyDeviceToHost, e->stream);
  hipError_t hs = hipStreamSynchronize(e->stream);
  dj.release(); dok.release(); dem.release(); dh.release();
  if (r) return r;
  if (he != hipSuccess) return fail(e, ZKE_E_DEVICE, "rsa batch", he);
  if (hs != hipSuccess) return fail(e, ZKE_E_DEVICE, "rsa batch sync", hs);
  for (uint32_t i = 0; i < n; i++) memcpy(em + (size_t)i * bytes, emh.data() + (size_t)i * 512 + 512 - bytes, bytes);
  return 0;
}

int zke_ed25519_verify_batch(zke_engine* e, const uint8_t* keys, const uint8_t* msgs, uint32_t msg_len,
                             const uint8_t* sigs, uint32_t n, uint32_t* out) {
  if (!e) return ZKE_E_ARG;
  if (n && (!keys || !msgs || !sigs || !out)) return fail(e, ZKE_E_ARG, "ed25519 batch: null pointer");
  if (msg_len == 0 || msg_len > 32) return fail(e, ZKE_E_ARG, "ed25519 batch: msg_len must be 1..32");
  if (n == 0) return 0;
  std::shared_lock<std::shared_mutex> sh(e->big);
  std::lock_guard<std::mutex> g(e->misc_mu);
  HIPCHK(e, hipSetDevice(e->device));
  DevBuf dk, dm, ds, dout;
  int r = 0;
  if ((r = dk.ensure((size_t)n * 32)) || (r = dm.ensure((size_t)n * msg_len)) || (r = ds.ensure((size_t)n * 64)) || (r = dout.ensure((size_t)n * 4))) {
    dk.release(); dm.release(); ds.release(); dout.release();
    return fail(e, r, "ed25519 batch allocation");
  }
  hipError_t he = hipMemcpyAsync(dk.p, keys, (size_t)n * 32, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(dm.p, msgs, (size_t)n * msg_len, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) he = hipMemcpyAsync(ds.p, sigs, (size_t)n * 64, hipMemcpyHostToDevice, e->stream);
  if (he == hipSuccess) {
    hipLaunchKernelGGL(ed25519_verify_kernel, dim3((n + 15) / 16), dim3(64), 0, e->stream, dk.as<uint8_t>(), dm.as<uint8_t>(), msg_len,
                       ds.as<uint8_t>(), n, dout.as<uint32_t>());
    he = hipGetLastError();
  }
  if (he == hipSuccess) he = hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream);
  hipError_t hs = hipStreamSynchronize(e->stream);
  dk.release(); dm.release(); ds.release(); dout.release();
  if (he != hipSuccess) return fail(e, ZKE_E_DEVICE, "ed25519 batch", he);
  if (hs != hipSuccess) return fail(e, ZKE_E_DEVICE, "ed25519 batch sync", hs);
  return 0;
}

}  // extern "C"
