// engine.hip — the C-ABI of include/zkemail_amd.h over the HIP kernels in this directory.
//
// Host side: workspace management, kernel launches on one stream, DFA registration.  No
// verification arithmetic runs on the host and there is no CPU fallback: without a HIP
// device every entry point returns ZKE_E_DEVICE.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
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

struct RegisteredDfa {
  bool valid = false;       // both blobs deserialise (dense::DFA::from_bytes would succeed)
  size_t lds_bytes = 0;     // repacked fwd + rev tables
  uint32_t idle = 0xFFFFFFFFu;   // forward automaton's idle state (dfa_idle_state)
  DevBuf blob;              // the repacked tables
  DevBuf dev;               // RegexDev image
  std::vector<uint8_t> fwd_copy, bwd_copy;     // the registered bytes: zke_dfa_register gives an equal pair its old id
};

inline uint32_t e_k(uint32_t bits) { return (bits + 7) / 8; }
constexpr int SHA_TILE = 128;
constexpr uint32_t SHA_PAIR_MAX_GROUPS = 512;      // launches of up to 32 768 messages use two waves per 64 messages


}  // namespace

// One submission slot: a stream and a private workspace.  A batch runs in one slot from its first kernel to its
// last; `slots` batches can be in flight on one engine (zke_engine_reserve).  What batches share lives in the engine:
// the per-key Montgomery constants (one cache per device) and the registered DFA tables.
struct Slot {
  hipStream_t stream = nullptr;        // the slot's own stream: batches submitted with stream == NULL run here
  bool owns_stream = true;
  hipStream_t last_stream = nullptr;   // stream of the slot's previous batch (nullptr: the slot has not been used)
  hipEvent_t done = nullptr;           // recorded behind the slot's last batch; waited for when the stream changes
  hipEvent_t ev[16]{};                 // per-kernel timing marks (zke_set_timing)
  int timed_marks = 0;
  bool timed_regex = false;
  DevBuf meta, rsa_jobs, sha_jobs, rsa_ok, em_dbg, scratch_off, scratch, clean, meta2, scratch2, parts;
  DevBuf pending;  // device counters: e-mails that need another signature round
  DevBuf* all[12] = {&meta, &rsa_jobs, &sha_jobs, &rsa_ok, &em_dbg, &scratch_off, &scratch, &clean, &meta2, &scratch2, &parts, &pending};
  // hipGraph replay of a batch's kernel sequence (ZKE_GRAPHS=1).  The graph holds this slot's workspace
  // pointers, so it is valid only while none of them has been reallocated: `generation` counts reallocations.
  hipGraphExec_t graph_exec = nullptr;
  std::vector<uint8_t> graph_key;      // everything the captured launches depend on, byte for byte
  uint64_t generation = 0;
};

struct zke_engine {
  int device = 0;
  hipStream_t stream = nullptr;     // = slots[0]->stream: host-mode batches and the building-block entry points
  std::string err;
  bool timing = false;
  zke_timings last{};
  hipEvent_t ev_h2d[4]{};
  std::vector<Slot*> slots;         // at least one (zke_engine_create); more after zke_engine_reserve
  uint32_t next_slot = 0;           // round-robin cursor of zke_verify_batch_device
  uint32_t last_slot = 0;           // slot of the most recent batch (zke_get_timings)
  size_t dfa_lds_attr = 0;
  std::vector<uint32_t> host_hdr_ids, host_body_ids;
  // host-mode staging of the inputs and results (zke_verify_batch)
  DevBuf in_raw, in_raw_off, in_dom, in_dom_off, in_key, in_key_off, in_ktype, in_extnull;
  DevBuf in_cap_off, in_cap_str_off, in_cap_blob;
  DevBuf results;
  DevBuf misc;   // building-block entry points
  DevBuf key_cache; // KeyCacheEntry[KEY_CACHE_SLOTS]: per-key Montgomery constants, kept across batches, shared by the slots
  std::vector<RegisteredDfa*> dfas;
  int sha_tile = SHA_TILE;
  int dfa_wave = 1;                 // regex parts: one e-mail per wave with a chunk map (ZKE_DFA_WAVE=0: one e-mail per lane)
  size_t dfa_wave_lds_attr = 0;
  uint64_t batch_key_total = 0;     // key bytes of the batch being run: > 272 per e-mail -> some modulus is above 2048 bits
  int rsa_quad = -1;                // four-lanes-per-signature RSA kernel (rsa_quad.hip.h): -1 by batch size, 0 never, 1 always (ZKE_RSA_QUAD)
  uint32_t rsa_quad_min = 256;      // -1: batches of at least this many e-mails (ZKE_RSA_QUAD_MIN).  Low since the modexp runs beside
                                    // SHA-256 (fused.hip.h): its longer chain no longer sits behind the hashes of a small batch; a
                                    // handful of e-mails is still answered sooner by the short chain of one signature per wave
  uint32_t rsa_oct_min = 128;       // ... for the eight-lane form of moduli above 2048 bits (ZKE_RSA_OCT_MIN)
  int sha_pair = -1;                // two-wave SHA-256 kernel: -1 by launch size, 0 never, 1 always (ZKE_SHA_PAIR)
  uint32_t debug_skip_rsa = 0;      // ZKE_DEBUG_SKIP_RSA: ablation experiments (results are then meaningless)
  uint32_t fuse_canon = 1;          // body canonicalisation inside the front end (ZKE_NO_FUSE_CANON=1: own launch)
  uint32_t debug_skip_ed = 0;       // ZKE_DEBUG_SKIP_ED: ablation, drops the Ed25519 stage launch (Ed25519 e-mails then fail)
  uint32_t debug_parse_stop = 0;    // ZKE_DEBUG_PARSE_STOP: timing experiments (results are then meaningless)
  uint32_t max_sig_rounds = 16;     // same-domain signatures tried per e-mail before ZKE_D_U_TOO_MANY_SIGS (options.reserved[0], up to 256)
  bool use_graphs = false;          // ZKE_GRAPHS=1: device-mode batches replay a captured hipGraph when the same descriptor comes again
  uint32_t key_cache_replicas = 1;  // ZKE_KEY_CACHE_REPLICAS (experiment): slot k uses copy k % replicas of the key cache
};

namespace {

int fail(zke_engine* e, int code, const char* what, hipError_t he = hipSuccess) {
  if (e) {
    e->err = what;
    if (he != hipSuccess) { e->err += ": "; e->err += hipGetErrorString(he); }
  }
  return code;
}
#define HIPCHK(e, call) do { hipError_t _r = (call); if (_r != hipSuccess) return fail((e), ZKE_E_DEVICE, #call, _r); } while (0)

// Kernel attributes are per device, so they belong to the engine (one engine = one device) and are set once at
// creation — never lazily in the submit path, where a first use would land inside somebody's timed region.
template <int T>
int set_sha_attrs(zke_engine* e) {
  HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&sha256_batch_kernel<T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sha256_lds_bytes<T>()));
  HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&sha256_pair_kernel<T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sha256_pair_lds_bytes<T>()));
  return 0;
}

template <int T>
int launch_sha(zke_engine* e, const ShaJob* jobs, uint32_t n, hipStream_t s) {
  if (n == 0) return 0;
  const size_t lds = sha256_lds_bytes<T>();
  // Few messages: the launch is as long as one wave's chain of compressions, so split the chain over two waves
  // (sha256_pair_kernel).  Many messages: the chip is full and the one-wave kernel does less LDS work per byte.
  const uint32_t groups = (n + 63) / 64;
  if (e->sha_pair == 1 || (e->sha_pair < 0 && groups <= SHA_PAIR_MAX_GROUPS)) {
    const size_t plds = sha256_pair_lds_bytes<T>();
    hipLaunchKernelGGL(sha256_pair_kernel<T>, dim3(groups), dim3(128), plds, s, jobs, n);
    HIPCHK(e, hipGetLastError());
    return 0;
  }
  const uint32_t grid = (n + 255) / 256;
  hipLaunchKernelGGL(sha256_batch_kernel<T>, dim3(grid), dim3(256), lds, s, jobs, n);
  HIPCHK(e, hipGetLastError());
  return 0;
}

int set_kernel_attrs(zke_engine* e);      // pipeline.hip.h

// tile size is a tuning knob (LDS per wave = 64 * (T + 16) bytes sets the occupancy); ZKE_SHA_TILE overrides for experiments
int launch_sha_any(zke_engine* e, const ShaJob* jobs, uint32_t n, hipStream_t s) {
  switch (e->sha_tile) {
    case 64: return launch_sha<64>(e, jobs, n, s);
    case 256: return launch_sha<256>(e, jobs, n, s);
    case 512: return launch_sha<512>(e, jobs, n, s);
    default: return launch_sha<SHA_TILE>(e, jobs, n, s);
  }
}
int set_sha_attrs_any(zke_engine* e) {
  switch (e->sha_tile) {
    case 64: return set_sha_attrs<64>(e);
    case 256: return set_sha_attrs<256>(e);
    case 512: return set_sha_attrs<512>(e);
    default: e->sha_tile = SHA_TILE; return set_sha_attrs<SHA_TILE>(e);
  }
}

Slot* new_slot(zke_engine* e) {
  Slot* w = new Slot();
  if (const char* xd = getenv("ZKE_X_DUMMY_STREAMS")) {        // experiment: idle streams created in front of each slot's stream (leaked)
    for (int k = 0; k < atoi(xd); k++) { hipStream_t d; (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking); }
  }
  const char* sp = getenv("ZKE_STREAM_PRIO");     // experiment: create the slot streams with an explicit priority
  if (const char* xs = getenv("ZKE_X_SHARE")) {                // experiment: slot i >= k runs on the stream of slot i - k (two workspaces per hardware queue)
    const size_t k = (size_t)atoi(xs);
    if (k && e->slots.size() >= k) { w->stream = e->slots[e->slots.size() - k]->stream; w->owns_stream = false; }
  }
  if (const char* xm = getenv("ZKE_X_CU_MASK")) {              // experiment: a slot's launches on 32 of the 256 CUs (1: bits k, k+8, ...; 2: bits 32k .. 32k+31)
    static int slot_ix = 0;
    const int k = slot_ix++ % 8, mode = atoi(xm);
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < 256; b++) if (mode == 1 ? (b % 8 == k) : (b / 32 == k)) mask[b / 32] |= 1u << (b % 32);
    if (hipExtStreamCreateWithCUMask(&w->stream, 8, mask) != hipSuccess) { e->err = "hipExtStreamCreateWithCUMask"; delete w; return nullptr; }
  }
  bool ok = (w->stream ? hipSuccess : sp ? hipStreamCreateWithPriority(&w->stream, hipStreamNonBlocking, atoi(sp)) : hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking)) == hipSuccess &&
            hipEventCreateWithFlags(&w->done, hipEventDisableTiming) == hipSuccess;
  for (auto& ev : w->ev) ok = ok && hipEventCreate(&ev) == hipSuccess;
  if (!ok) { e->err = "slot stream / event creation"; delete w; return nullptr; }
  for (auto* b : w->all) b->generation = &w->generation;
  return w;
}
void free_slot(Slot* w) {
  if (!w) return;
  if (w->stream && w->owns_stream) (void)hipStreamSynchronize(w->stream);
  if (w->graph_exec) (void)hipGraphExecDestroy(w->graph_exec);
  for (auto* b : w->all) b->release();
  for (auto& ev : w->ev) if (ev) (void)hipEventDestroy(ev);
  if (w->done) (void)hipEventDestroy(w->done);
  if (w->stream && w->owns_stream) (void)hipStreamDestroy(w->stream);
  delete w;
}

// Which lane-group RSA kernels take part in a batch of n e-mails.  Bit 0: four lanes per signature (moduli <= 2048 bits),
// bit 1: eight lanes (2049..4096 bits; any_big = the caller's hint that the batch's keys average more than an RSA-2048 key).
// The front end routes signatures by this mask and the hash / modexp launch gets the matching workgroups: the same value
// must go to both.  -1 = by batch size (ZKE_RSA_QUAD_MIN / ZKE_RSA_OCT_MIN), 0 never, 1 always (ZKE_RSA_QUAD).
uint32_t rsa_route_mask(const zke_engine* e, uint32_t n, bool any_big) {
  if (!e->key_cache.p) return 0;
  uint32_t m = 0;
  if (e->rsa_quad > 0 || (e->rsa_quad < 0 && n >= e->rsa_quad_min)) m |= 1u;
  if (any_big && (e->rsa_quad > 0 || (e->rsa_quad < 0 && n >= e->rsa_oct_min))) m |= 2u;
  return m;
}

template <int T>
int launch_stage_t(zke_engine* e, const StageArgs& A, hipStream_t s) {
  const uint32_t grid = A.g_sha + A.g_wave + A.g_quad + A.g_oct;
  if (!grid) return 0;
  hipLaunchKernelGGL(hash_modexp_kernel<T>, dim3(grid), dim3(128), sha256_pair_lds_bytes<T>(), s, A);
  HIPCHK(e, hipGetLastError());
  return 0;
}
int launch_stage_any(zke_engine* e, const StageArgs& A, hipStream_t s) {
  switch (e->sha_tile) {
    case 64: return launch_stage_t<64>(e, A, s);
    case 256: return launch_stage_t<256>(e, A, s);
    case 512: return launch_stage_t<512>(e, A, s);
    default: return launch_stage_t<SHA_TILE>(e, A, s);
  }
}

// The hash / modexp stage of a batch (fused.hip.h): SHA-256 of the 4 * n_pad messages and the RSA operation of the n jobs.
// Launches of up to SHA_PAIR_MAX_GROUPS SHA-256 groups (every BASELINE-sized batch) are ONE kernel; beyond that the chip
// is full of SHA-256 waves anyway: sha256_batch_kernel first, then the RSA roles as a launch of their own.
constexpr uint32_t WAVE_ROLE_MAX_GROUPS = 128;     // 256 waves walk the job list: a batch of 1 024 uncached keys takes four rounds of them
int launch_hash_modexp(zke_engine* e, const ShaJob* sha, uint32_t n_sha, const RsaJob* rsa, uint32_t n, EmailMeta* meta,
                       uint8_t* em_out, uint32_t route_mask, const uint32_t* wave_count, const uint32_t* wave_list, hipStream_t s) {
  StageArgs A{};
  A.sha = sha; A.n_sha = n_sha; A.rsa = rsa; A.n = n; A.meta = meta;
  A.cache = e->key_cache.as<KeyCacheEntry>();
  A.em_out = em_out;
  A.wave_count = wave_count; A.wave_list = wave_list;
  A.g_wave = wave_list ? std::min<uint32_t>((n + 1) / 2, WAVE_ROLE_MAX_GROUPS) : (n + 1) / 2;
  A.g_quad = (route_mask & 1u) ? (n + 31) / 32 : 0;
  A.g_oct = (route_mask & 2u) ? (n + 15) / 16 : 0;
  A.debug_skip_rsa = e->debug_skip_rsa;
  const uint32_t groups = (n_sha + 63) / 64;
  if (e->sha_pair == 1 || (e->sha_pair < 0 && groups <= SHA_PAIR_MAX_GROUPS)) {
    A.g_sha = groups;
    return launch_stage_any(e, A, s);
  }
  if (int r = launch_sha_any(e, sha, n_sha, s)) return r;
  A.g_sha = 0;
  return launch_stage_any(e, A, s);
}

template <int T>
int set_stage_attr(zke_engine* e) {
  HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&hash_modexp_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)sha256_pair_lds_bytes<T>()));
  return 0;
}
int set_stage_attr_any(zke_engine* e) {
  switch (e->sha_tile) {
    case 64: return set_stage_attr<64>(e);
    case 256: return set_stage_attr<256>(e);
    case 512: return set_stage_attr<512>(e);
    default: return set_stage_attr<SHA_TILE>(e);
  }
}

}  // namespace

#include "pipeline.hip.h"

// HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable when the runtime
// initialises.  A process that loads this library before it first touches HIP (every C / C++ / Rust host that links it)
// gets room for 22 submission slots + the null stream unless it has set the variable itself — and no more: the chip runs 24
// queues of a process without time-slicing them, the 25th costs a factor of ten, and with the pool capped at 23 a stream the
// process creates on top (a communicator's, a framework's) shares a queue instead of adding one; DESIGN.md §5, INTEGRATION.md §5.
__attribute__((constructor)) static void zke_default_hw_queues() { (void)setenv("GPU_MAX_HW_QUEUES", "23", 0); }

extern "C" {

const char* zke_version(void) { return "zkemail.rs_amd 0.2 (gfx950)"; }

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

int zke_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n > 0 ? 1 : 0;
}

int zke_engine_create(const zke_options* opt, zke_engine** out) {
  if (!out) return ZKE_E_ARG;
  *out = nullptr;
  if (!zke_device_available()) return ZKE_E_DEVICE;
  zke_engine* e = new zke_engine();
  int dev = opt ? opt->device : -1;
  if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
  e->device = dev;
  if (hipSetDevice(dev) != hipSuccess) { delete e; return ZKE_E_DEVICE; }
  Slot* w0 = new_slot(e);
  if (!w0) { delete e; return ZKE_E_DEVICE; }
  e->slots.push_back(w0);
  e->stream = w0->stream;
  for (auto& ev : e->ev_h2d) if (hipEventCreate(&ev) != hipSuccess) { zke_engine_destroy(e); return ZKE_E_DEVICE; }
  if (const char* kr = getenv("ZKE_KEY_CACHE_REPLICAS")) e->key_cache_replicas = std::max(1, std::min(64, atoi(kr)));
  if (!(opt && opt->reserved[3])) {      // reserved[3] != 0: no per-key cache (R^2 mod n recomputed per signature)
    const size_t kc_bytes = (size_t)e->key_cache_replicas * KEY_CACHE_SLOTS * sizeof(KeyCacheEntry);
    if (e->key_cache.ensure(kc_bytes) || hipMemset(e->key_cache.p, 0, kc_bytes) != hipSuccess) {
      zke_engine_destroy(e);
      return ZKE_E_NOMEM;
    }
  }
  if (opt && opt->reserved[0]) e->max_sig_rounds = std::min<uint32_t>(opt->reserved[0], ZKE_MAX_HEADERS);
  if (const char* g = getenv("ZKE_GRAPHS")) e->use_graphs = atoi(g) != 0;
  if (const char* st = getenv("ZKE_SHA_TILE")) e->sha_tile = atoi(st);
  if (const char* sp = getenv("ZKE_SHA_PAIR")) e->sha_pair = atoi(sp);
  if (const char* rq = getenv("ZKE_RSA_QUAD")) e->rsa_quad = atoi(rq);
  if (const char* rm = getenv("ZKE_RSA_QUAD_MIN")) e->rsa_quad_min = (uint32_t)atoi(rm);
  if (const char* ro = getenv("ZKE_RSA_OCT_MIN")) e->rsa_oct_min = (uint32_t)atoi(ro);
  if (const char* dw = getenv("ZKE_DFA_WAVE")) e->dfa_wave = atoi(dw);
  if (getenv("ZKE_DEBUG_SKIP_RSA")) e->debug_skip_rsa = 1;
  if (getenv("ZKE_DEBUG_SKIP_ED")) e->debug_skip_ed = 1;
  if (getenv("ZKE_NO_FUSE_CANON")) e->fuse_canon = 0;
  if (const char* ds = getenv("ZKE_DEBUG_PARSE_STOP")) e->debug_parse_stop = (uint32_t)atoi(ds);
  if (set_kernel_attrs(e)) { zke_engine_destroy(e); return ZKE_E_DEVICE; }
  *out = e;
  return 0;
}

void zke_engine_destroy(zke_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  for (Slot* w : e->slots) if (w->stream) (void)hipStreamSynchronize(w->stream);      // (slots may share a stream: all drained before any is freed)
  for (Slot* w : e->slots) free_slot(w);
  DevBuf* bufs[] = {&e->in_raw, &e->in_raw_off, &e->in_dom, &e->in_dom_off, &e->in_key, &e->in_key_off, &e->in_ktype,
                    &e->in_extnull, &e->in_cap_off, &e->in_cap_str_off, &e->in_cap_blob, &e->results, &e->misc, &e->key_cache};
  for (auto* b : bufs) b->release();
  for (auto* d : e->dfas) { d->blob.release(); d->dev.release(); delete d; }
  for (auto& ev : e->ev_h2d) if (ev) (void)hipEventDestroy(ev);
  delete e;
}

const char* zke_last_error(const zke_engine* e) { return e ? e->err.c_str() : "null engine"; }

int zke_engine_join(zke_engine* e, void* stream) {
  if (!e) return ZKE_E_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  hipStream_t s = (hipStream_t)stream;
  for (Slot* w : e->slots) {
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
  HIPCHK(e, hipSetDevice(e->device));
  for (Slot* w : e->slots) {
    // a slot's last batch ran on the slot's own stream, or on a caller's stream with `done` recorded behind it
    if (w->last_stream && w->last_stream != w->stream) HIPCHK(e, hipEventSynchronize(w->done));
    HIPCHK(e, hipStreamSynchronize(w->stream));
  }
  return 0;
}

int zke_set_timing(zke_engine* e, int enabled) {
  if (!e) return ZKE_E_ARG;
  e->timing = enabled != 0;
  return 0;
}

int zke_get_slot_timings(zke_engine* e, uint32_t slot, zke_timings* t) {
  if (!e || !t || slot >= e->slots.size()) return ZKE_E_ARG;
  Slot& w = *e->slots[slot];
  if (e->timing && w.timed_marks > 0) {     // device-mode batches: the events are read once their stream has drained
    HIPCHK(e, hipEventSynchronize(w.ev[w.timed_marks - 1]));
    collect_timings(e, w);
  }
  *t = e->last;
  return 0;
}

int zke_get_timings(zke_engine* e, zke_timings* t) {
  if (!e) return ZKE_E_ARG;
  return zke_get_slot_timings(e, e->last_slot, t);
}

// ---------------------------------------------------------------- building blocks
int zke_sha256_batch_device(zke_engine* e, const uint8_t* blob_dev, const uint64_t* off_dev, uint32_t n,
                            uint8_t* digests_dev, void* stream) {
  if (!e) return ZKE_E_ARG;
  if (n == 0) return 0;
  HIPCHK(e, hipSetDevice(e->device));
  hipStream_t s = stream ? (hipStream_t)stream : e->stream;
  if (int r = e->misc.ensure((size_t)n * sizeof(ShaJob))) return fail(e, r, "workspace");
  hipLaunchKernelGGL(sha_jobs_from_csr_kernel, dim3((n + 255) / 256), dim3(256), 0, s, blob_dev, off_dev, n, digests_dev,
                     e->misc.as<ShaJob>());
  HIPCHK(e, hipGetLastError());
  return launch_sha_any(e, e->misc.as<ShaJob>(), n, s);
}

int zke_sha256_batch(zke_engine* e, const uint8_t* blob, const uint64_t* off, uint32_t n, uint8_t* digests) {
  if (!e || (n && (!blob || !off || !digests))) return ZKE_E_ARG;
  if (n == 0) return 0;
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
    r = zke_sha256_batch_device(e, dblob.as<uint8_t>(), doff.as<uint64_t>(), n, ddig.as<uint8_t>(), e->stream);
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
