// pipeline.hip.h — host orchestration of one batch: workspace sizing, the kernel sequence of
// verify_email / verify_email_with_regex (core/src/circuits.rs:9-68), DFA registration.
// Included by engine.hip (single translation unit).
#pragma once

namespace {

inline uint64_t host_scratch_off(const uint64_t* raw_off, uint32_t i) { return scratch_offset(raw_off[i] - raw_off[0], i); }

struct StageTimer {
  Slot* w; hipStream_t s; int k = 0; bool on;
  StageTimer(zke_engine* e_, Slot* w_, hipStream_t s_) : w(w_), s(s_), on(e_->timing) {}
  void mark() { if (on && k < 16) (void)hipEventRecord(w->ev[k++], s); }
};

// Workspace of one slot for batches of up to n e-mails / raw_total raw bytes (P regex parts; with_regex: the buffers of
// the canonicalize_signed_email pass too).  Grows only: in steady state — or after zke_engine_reserve — this allocates nothing.
int ensure_workspace(zke_engine* e, Slot& w, uint32_t n, uint64_t raw_total, bool with_regex, uint32_t P, bool want_em) {
  const uint32_t n_pad = (n + 63) & ~63u;
  int r = 0;
  const size_t scratch_bytes = 2 * (size_t)raw_total + (size_t)(n + 1) * SCR_PER_EMAIL + 256;
  if ((r = w.meta.ensure((size_t)n * sizeof(EmailMeta))) || (r = w.rsa_jobs.ensure((size_t)n * sizeof(RsaJob))) ||
      (r = w.sha_jobs.ensure((size_t)4 * n_pad * sizeof(ShaJob))) || (r = w.rsa_ok.ensure((size_t)n * 4)) ||
      (r = w.scratch_off.ensure((size_t)(n + 1) * 16)) || (r = w.scratch.ensure(scratch_bytes)))
    return fail(e, r, "workspace allocation");
  if (!w.pending.p) {       // counters: [0] e-mails pending another signature round, [2] length of the wave-routine job list (rsa_ok)
    if ((r = w.pending.ensure(64))) return fail(e, r, "workspace allocation");
    HIPCHK(e, hipMemset(w.pending.p, 0, 64));
  }
  if (want_em && (r = w.em_dbg.ensure((size_t)n * 512))) return fail(e, r, "workspace allocation");
  if (with_regex) {
    if ((r = w.meta2.ensure((size_t)n * sizeof(EmailMeta))) || (r = w.scratch2.ensure(scratch_bytes)) ||
        (r = w.clean.ensure((size_t)raw_total + (size_t)(n + 1) * CLEAN_PER_EMAIL + 256)) ||
        (r = w.parts.ensure((size_t)n * std::max<uint32_t>(P, 1) * sizeof(PartRes))))
      return fail(e, r, "workspace allocation");
  }
  return 0;
}

// Every hipFuncSetAttribute the pipeline needs, once per engine (= per device).  The DFA kernels' LDS sizes depend on
// the registered tables: zke_dfa_register raises them.
int set_kernel_attrs(zke_engine* e) {
  if (int r = set_sha_attrs_any(e)) return r;
  if (int r = set_stage_attr_any(e)) return r;
  return 0;
}
int raise_dfa_lds_attrs(zke_engine* e, size_t lds) {
  if (lds > e->dfa_wave_lds_attr) {
    HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&dfa_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    e->dfa_wave_lds_attr = lds;
  }
  if (lds > e->dfa_lds_attr) {
    HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&dfa_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    e->dfa_lds_attr = lds;
  }
  return 0;
}

// The device pipeline.  Every pointer in `in` / out_dev is device memory.  Three launches — front end, hash / modexp
// stage, Ed25519 + verdict (which also runs the later signature rounds of the rare e-mail that needs them) — and, for
// verify_email_with_regex, the regex stage behind them.
int run_device_pipeline(zke_engine* e, Slot& w, const zke_batch* in, uint64_t raw_total, zke_result* out_dev, hipStream_t s,
                        bool want_em) {
  const uint32_t n = in->n;
  if (n == 0) return 0;
  const uint32_t n_pad = (n + 63) & ~63u;
  const uint32_t P = in->with_regex ? in->n_header_parts + in->n_body_parts : 0;
  int r = 0;
  if ((r = ensure_workspace(e, w, n, raw_total, in->with_regex != 0, P, want_em))) return r;
  uint64_t* scratch_off = w.scratch_off.as<uint64_t>();
  uint64_t* clean_off = scratch_off + (n + 1);

  StageTimer tm(e, &w, s);
  tm.mark();
  // (offsets, padding SHA jobs and the pending counter are initialised by the round-0 front-end kernel: batch_prologue)

  BatchDev B{};
  B.n = n;
  B.raw = in->raw_blob; B.raw_off = in->raw_off;
  B.dom = in->domain_blob; B.dom_off = in->domain_off;
  B.key = in->key_blob; B.key_off = in->key_off;
  B.key_type = in->key_type; B.ext_null = in->ext_null;
  B.results = out_dev;
  B.meta = w.meta.as<EmailMeta>();
  B.rsa = w.rsa_jobs.as<RsaJob>();
  B.sha = w.sha_jobs.as<ShaJob>();
  B.n_pad = n_pad;
  B.scratch = w.scratch.as<uint8_t>();
  B.scratch_off = scratch_off;
  B.clean_off = clean_off;
  B.pending = w.pending.as<uint32_t>();
  B.meta_verify = nullptr;

  const uint32_t rounds = std::max<uint32_t>(1, e->max_sig_rounds);
  // an RSA-2048 key is 270 bytes of DER: a batch whose keys average more holds some larger modulus
  const uint32_t route_mask = rsa_route_mask(e, n, e->batch_key_total > (uint64_t)n * 272);
  {
    const uint32_t round = 0;
    uint32_t* wave_count = w.pending.as<uint32_t>() + 2;
    uint32_t* wave_list = w.rsa_ok.as<uint32_t>();
    ParseArgs pa{B, round, 0, e->debug_parse_stop, e->fuse_canon, e->key_cache.as<KeyCacheEntry>(), route_mask, wave_count, wave_list};
    // experiment (ZKE_X_ANYORDER, with ZKE_X_SHARE): the front end without the queue's barrier bit — it reads nothing the slot's
    // stream has in flight (the other workspace of the pair), so it may start beside the previous batch's verdict launch
    static const bool x_anyorder = getenv("ZKE_X_ANYORDER") != nullptr;
    if (x_anyorder && s == w.stream && !e->timing)
      hipExtLaunchKernelGGL(parse_kernel, dim3((n + ZKE_PARSE_WG_WAVES - 1) / ZKE_PARSE_WG_WAVES), dim3(64 * ZKE_PARSE_WG_WAVES), PARSE_DYN_LDS, s, nullptr, nullptr, hipExtAnyOrderLaunch, pa);
    else
      hipLaunchKernelGGL(parse_kernel, dim3((n + ZKE_PARSE_WG_WAVES - 1) / ZKE_PARSE_WG_WAVES), dim3(64 * ZKE_PARSE_WG_WAVES), PARSE_DYN_LDS, s, pa);
    tm.mark();
    if (!e->fuse_canon) {      // the front end canonicalises the body itself
      CanonArgs ca{B, 0};
      hipLaunchKernelGGL(canon_body_kernel, dim3(n), dim3(64), 0, s, ca);
    }
    tm.mark();
    // hash / modexp stage: the four SHA-256 jobs and the RSA operation of every e-mail, one launch (fused.hip.h)
    static const int x_skip = getenv("ZKE_DEBUG_SKIP_LAUNCH") ? atoi(getenv("ZKE_DEBUG_SKIP_LAUNCH")) : 0;   // ablation: bit 0 stage, bit 1 verdict
    if (!(x_skip & 1) && (r = launch_hash_modexp(e, B.sha, 4 * n_pad, B.rsa, n, B.meta, want_em ? w.em_dbg.as<uint8_t>() : nullptr, route_mask,
                                wave_count, wave_list, s)))
      return r;
    tm.mark();
    // Ed25519 stage + verdicts (verdict.hip.h): bh compare, EM digest against the header hash, status / detail, pending counter
    EdVerdictArgs va{FinArgs{B, round, rounds, w.pending.as<uint32_t>(), e->debug_skip_rsa}, e->debug_skip_ed, wave_count,
                     e->key_cache.as<KeyCacheEntry>(), want_em ? w.em_dbg.as<uint8_t>() : nullptr};
    if (!(x_skip & 2)) hipLaunchKernelGGL(ed_verdict_kernel, dim3((n + VERDICT_EMAILS_PER_WAVE - 1) / VERDICT_EMAILS_PER_WAVE), dim3(64), 0, s, va);
    tm.mark(); tm.mark();      // sha_us = the hash / modexp launch, rsa_us = the Ed25519 + verdict launch (finalize_us reads 0)
  }
  HIPCHK(e, hipGetLastError());

  if (in->with_regex) {
    // canonicalize_signed_email (circuits.rs:34-35): first DKIM-Signature header, own scratch unless it is the verified one
    BatchDev B2 = B;
    B2.meta = w.meta2.as<EmailMeta>();
    B2.scratch = w.scratch2.as<uint8_t>();
    B2.meta_verify = B.meta;
    ParseArgs pa{B2, 0, 1, 0, 0, nullptr, 0, nullptr, nullptr};
    hipLaunchKernelGGL(parse_kernel, dim3((n + ZKE_PARSE_WG_WAVES - 1) / ZKE_PARSE_WG_WAVES), dim3(64 * ZKE_PARSE_WG_WAVES), PARSE_DYN_LDS, s, pa);
    CanonArgs ca{B2, 1};
    hipLaunchKernelGGL(canon_body_kernel, dim3(n), dim3(64), 0, s, ca);
    QpArgs qa{B2, B.meta, w.clean.as<uint8_t>(), clean_off, B.scratch, B.scratch_off};
    hipLaunchKernelGGL(qp_kernel, dim3(n), dim3(64), 0, s, qa);    // circuits.rs:37 runs whether or not body parts exist
    tm.mark();
    auto part_dfa = [&](uint32_t p) -> const RegisteredDfa* {
      // part ids are host-visible only in host mode; zke_verify_batch_device receives them as host arrays too
      const uint32_t id = p >= in->n_header_parts ? e->host_body_ids[p - in->n_header_parts] : e->host_hdr_ids[p];
      return id < e->dfas.size() ? e->dfas[id] : nullptr;
    };
    DfaArgs base{};
    base.b = B2; base.P = P;
    base.scratch_v = B.scratch; base.scratch_v_off = B.scratch_off;
    base.clean = w.clean.as<uint8_t>(); base.clean_off = clean_off;
    base.cap_off = in->cap_off; base.cap_str_off = in->cap_str_off; base.cap_blob = in->cap_blob;
    base.out = w.parts.as<PartRes>();
    auto part_lds = [&](const RegisteredDfa* rd, uint32_t& in_lds) -> size_t {
      in_lds = 0;
      if (rd && rd->valid && rd->lds_bytes + 1024 <= 150 * 1024) { in_lds = 1; return rd->lds_bytes + 1024; }
      return 1024;
    };
    // Which kernel: the wave-per-e-mail kernel shortens the chain (latency) but runs its serial part on one
    // lane's worth of work per wave, so it issues several times the instructions of the lane-per-e-mail kernel.
    // Body parts (KBs per e-mail) always gain; header parts (~1 KB) gain only while the batch is small enough for
    // latency to be what matters (measured: configs[2] shape, 4 096 per batch, 16.0 M e-mails/s with the lane kernel
    // against 12.7 M with the wave kernel; 1 024 per batch 11.4 M against 11.8 M).
    const uint32_t wave_from = !e->dfa_wave ? P : (n <= 1024 ? 0u : in->n_header_parts);      // parts [wave_from, P) use the wave kernel
    if (wave_from < P) {
      // one e-mail per wave; up to DFA_MULTI_MAX parts per launch (grid.y)
      for (uint32_t p0 = wave_from; p0 < P; p0 += DFA_MULTI_MAX) {
        const uint32_t np = std::min<uint32_t>(DFA_MULTI_MAX, P - p0);
        DfaMultiArgs ma{};
        ma.common = base; ma.part0 = p0; ma.n_header_parts = in->n_header_parts;
        size_t lds = 1024;
        for (uint32_t k = 0; k < np; k++) {
          const RegisteredDfa* rd = part_dfa(p0 + k);
          ma.re[k] = (rd && rd->valid) ? rd->dev.as<RegexDev>() : nullptr;
          lds = std::max(lds, part_lds(rd, ma.lds_tables[k]));
          ma.idle[k] = (rd && rd->valid) ? rd->idle : 0xFFFFFFFFu;
        }
        hipLaunchKernelGGL(dfa_wave_kernel, dim3((n + 3) / 4, np), dim3(256), lds, s, ma);
      }
    }
    {
      for (uint32_t p = 0; p < wave_from; p++) {               // one e-mail per lane, one launch per part
        const RegisteredDfa* rd = part_dfa(p);
        DfaArgs da = base;
        da.re = (rd && rd->valid) ? rd->dev.as<RegexDev>() : nullptr;
        da.part = p; da.is_body = p >= in->n_header_parts ? 1 : 0;
        const size_t lds = part_lds(rd, da.lds_tables);
        da.idle = 0xFFFFFFFFu;
        hipLaunchKernelGGL(dfa_kernel, dim3((n + 255) / 256), dim3(256), lds, s, da);
      }
    }
    RegexFinArgs rf{B2, w.parts.as<PartRes>(), in->n_header_parts, in->n_body_parts};
    hipLaunchKernelGGL(regex_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rf);
    tm.mark();
    HIPCHK(e, hipGetLastError());
  }
  tm.mark();
  w.timed_marks = tm.k;
  w.timed_regex = in->with_regex != 0;
  return 0;
}

void collect_timings(zke_engine* e, Slot& w) {
  if (!e->timing || w.timed_marks < 7) return;
  auto dt = [&](int a, int b) { float ms = 0; (void)hipEventElapsedTime(&ms, w.ev[a], w.ev[b]); return ms * 1000.f; };
  zke_timings& t = e->last;
  t.parse_us = dt(0, 1); t.canon_body_us = dt(1, 2); t.sha_us = dt(2, 3); t.rsa_us = dt(3, 4); t.finalize_us = dt(4, 5);
  if (w.timed_regex && w.timed_marks >= 9) { t.qp_us = dt(5, 6); t.dfa_us = dt(6, 7); }
  else { t.qp_us = 0; t.dfa_us = 0; }
  t.total_us = dt(0, w.timed_marks - 1);
}

// see zke_engine_reserve: 256 bytes of private memory per lane (the front end's spills are 116), never written to `sink`
__global__ void slot_warm_kernel(uint32_t* sink) {
  volatile uint32_t buf[512];
  for (int i = 0; i < 512; i++) buf[i] = (uint32_t)i * 2654435761u + threadIdx.x;
  uint32_t acc = 0;
  for (int i = 0; i < 512; i++) acc += buf[(i * 7 + threadIdx.x) & 511];
  if (acc == 0x12345678u && sink) *sink = acc;
}

// The slot's workspace is about to be overwritten by a batch on stream s: whatever ran in it before must be over.
// Same stream: stream order is enough.  Another stream: wait for the event recorded behind the previous batch.
int acquire_slot(zke_engine* e, Slot& w, hipStream_t s) {
  if (w.last_stream && w.last_stream != s) {
    // a batch on the slot's own stream leaves no event behind (release_slot): record it now, behind that batch —
    // waiting for `done` as it stood would order this batch behind nothing
    if (w.last_stream == w.stream) HIPCHK(e, hipEventRecord(w.done, w.stream));
    HIPCHK(e, hipStreamWaitEvent(s, w.done, 0));
  }
  return 0;
}
int release_slot(zke_engine* e, Slot& w, hipStream_t s) {
  // the event is needed only where the next user of the slot may sit on another stream: a caller-provided stream
  if (s != w.stream) HIPCHK(e, hipEventRecord(w.done, s));
  w.last_stream = s;
  return 0;
}

// ---- regex-automata 0.4.9 dense DFA, little-endian wire format ----
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

// dense::DFA::from_bytes restated: structure, sizes and the id validity checks.  false = would not deserialise.
bool parse_dfa_blob(const uint8_t* b, size_t n, HostDfa& h) {
  static const char LABEL[] = "rust-regex-automata-dfa-dense";
  DfaDev& d = h.d;
  size_t p = 0;
  while (p < n && p < 7 && b[p] == 0) p++;
  auto need = [&](size_t k) { return n - p >= k; };
  if (!need(32) || memcmp(b + p, LABEL, 29) || b[p + 29] != 0) return false;
  p += 32;
  if (!need(4) || rd32(b + p) != 0xFEFF) return false; p += 4;
  if (!need(4) || rd32(b + p) != 2) return false; p += 4;
  if (!need(4)) return false; p += 4;
  // Flags::from_bytes: ONE u32 bit set — bit 0 has_empty, bit 1 is_utf8, bit 2 is_always_start_anchored — as the blobs
  // regex-automata itself wrote show (tests/golden/regex_automata_*.dfa; SURVEY Appendix A.3 recalled three u32s)
  if (!need(4)) return false;
  { const uint32_t fl = rd32(b + p); d.has_empty = fl & 1u; d.is_utf8 = (fl >> 1) & 1u; d.always_anchored = (fl >> 2) & 1u; }
  p += 4;
  if (!need(8 + 256)) return false;
  d.state_len = rd32(b + p); d.stride2 = rd32(b + p + 4); p += 8;
  memcpy(d.classes, b + p, 256); p += 256;
  if (d.stride2 < 1 || d.stride2 > 9) return false;
  d.alphabet_len = (uint32_t)d.classes[255] + 2;
  if (d.alphabet_len > (1u << d.stride2)) return false;
  if (d.state_len > (1u << 26)) return false;
  const size_t tl = (size_t)d.state_len << d.stride2;
  if (!need(tl * 4)) return false;
  d.table_len = (uint32_t)tl;
  h.table.resize(tl);
  for (size_t i = 0; i < tl; i++) h.table[i] = rd32(b + p + 4 * i);
  p += tl * 4;
  const uint32_t stride = 1u << d.stride2;
  for (size_t s = 0; s < d.state_len; s++)
    for (uint32_t c = 0; c < d.alphabet_len; c++) {
      const uint32_t id = h.table[(s << d.stride2) + c];
      if (id >= tl || (id & (stride - 1))) return false;
    }
  if (!need(4 + 256 + 16)) return false;
  d.start_kind = rd32(b + p); p += 4;
  if (d.start_kind > 2) return false;
  memcpy(d.start_map, b + p, 256); p += 256;
  for (int i = 0; i < 256; i++) if (d.start_map[i] >= 6) return false;
  if (rd32(b + p) != 6) return false; p += 4;
  const uint32_t spl = rd32(b + p); p += 4;
  p += 8;
  const size_t npat = spl == 0xFFFFFFFFu ? 0 : spl;
  if (npat > (1u << 20)) return false;
  const size_t sl = 12 + 6 * npat;
  if (!need(sl * 4)) return false;
  for (size_t i = 0; i < sl; i++) {
    const uint32_t v = rd32(b + p + 4 * i);
    if (v >= tl || (v & (stride - 1))) return false;
    if (i < 12) d.starts[i] = v;
  }
  p += sl * 4;
  if (!need(4)) return false;
  const uint32_t ms_len = rd32(b + p); p += 4;
  if (ms_len > d.state_len) return false;
  if (!need((size_t)ms_len * 8 + 8)) return false;
  p += (size_t)ms_len * 8;
  p += 4;
  const uint32_t idlen = rd32(b + p); p += 4;
  if (idlen > (1u << 24) || !need((size_t)idlen * 4)) return false;
  p += (size_t)idlen * 4;
  if (!need(32)) return false;
  d.sp_max = rd32(b + p); d.quit_id = rd32(b + p + 4); d.min_match = rd32(b + p + 8); d.max_match = rd32(b + p + 12);
  const uint32_t min_accel = rd32(b + p + 16), max_accel = rd32(b + p + 20), min_start = rd32(b + p + 24), max_start = rd32(b + p + 28);
  p += 32;
  if (d.min_match > d.max_match || min_accel > max_accel || min_start > max_start) return false;
  if ((d.min_match == 0) != (d.max_match == 0)) return false;
  if (d.max_match > d.sp_max || max_accel > d.sp_max || max_start > d.sp_max) return false;
  if (tl && d.sp_max >= tl) return false;
  {
    const uint32_t nm = d.max_match ? ((d.max_match - d.min_match) >> d.stride2) + 1 : 0;
    if (nm != ms_len) return false;
  }
  if (!need(4)) return false;
  const uint32_t acc = rd32(b + p); p += 4;
  if (acc > d.state_len || !need((size_t)acc * 8)) return false;
  p += (size_t)acc * 8;
  if (!need(32)) return false;
  memcpy(d.quitset, b + p, 32);
  d.quitset_nonempty = 0;
  for (int i = 0; i < 32; i++) if (d.quitset[i]) d.quitset_nonempty = 1;
  d.wide = tl > 65536 ? 1u : 0u;
  d.valid = 1;
  return true;
}

}  // namespace

extern "C" {

int zke_dfa_register(zke_engine* e, const uint8_t* fwd, size_t fwd_len, const uint8_t* bwd, size_t bwd_len, uint32_t* out_id) {
  if (!e || !out_id || (fwd_len && !fwd) || (bwd_len && !bwd)) return ZKE_E_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  // registering the same pair again returns the id it already has (per-e-mail callers re-submit their part list)
  for (size_t k = 0; k < e->dfas.size(); k++) {
    const RegisteredDfa* d = e->dfas[k];
    if (d->fwd_copy.size() == fwd_len && d->bwd_copy.size() == bwd_len && (!fwd_len || !memcmp(d->fwd_copy.data(), fwd, fwd_len)) &&
        (!bwd_len || !memcmp(d->bwd_copy.data(), bwd, bwd_len))) {
      *out_id = (uint32_t)k;
      return 0;
    }
  }
  RegisteredDfa* rd = new RegisteredDfa();
  rd->fwd_copy.assign(fwd, fwd + fwd_len);
  rd->bwd_copy.assign(bwd, bwd + bwd_len);
  HostDfa hf, hr;
  const bool ok = parse_dfa_blob(fwd, fwd_len, hf) && parse_dfa_blob(bwd, bwd_len, hr);
  rd->valid = ok;
  if (ok) {
    auto packed = [](const HostDfa& h) { return (((size_t)h.d.table_len * (h.d.wide ? 4 : 2)) + 15) & ~(size_t)15; };
    const size_t fb = packed(hf), rb = packed(hr);
    rd->lds_bytes = fb + rb;
    rd->idle = dfa_idle_state(hf);
    int r = 0;
    if ((r = rd->blob.ensure(fb + rb + 64)) || (r = rd->dev.ensure(sizeof(RegexDev)))) { delete rd; return fail(e, r, "hipMalloc"); }
    std::vector<uint8_t> img(fb + rb + 64, 0);
    auto pack = [&](const HostDfa& h, size_t off) {
      if (h.d.wide) memcpy(img.data() + off, h.table.data(), h.table.size() * 4);
      else { uint16_t* o = reinterpret_cast<uint16_t*>(img.data() + off); for (size_t i = 0; i < h.table.size(); i++) o[i] = (uint16_t)h.table[i]; }
    };
    pack(hf, 0); pack(hr, fb);
    RegexDev rdv{};
    rdv.fwd = hf.d; rdv.rev = hr.d;
    rdv.fwd.table = (uint64_t)rd->blob.as<uint8_t>();
    rdv.rev.table = (uint64_t)(rd->blob.as<uint8_t>() + fb);
    hipError_t he = hipMemcpy(rd->blob.p, img.data(), img.size(), hipMemcpyHostToDevice);
    if (he == hipSuccess) he = hipMemcpy(rd->dev.p, &rdv, sizeof rdv, hipMemcpyHostToDevice);
    if (he != hipSuccess) { rd->blob.release(); rd->dev.release(); delete rd; return fail(e, ZKE_E_DEVICE, "dfa upload", he); }
  }
  if (ok && rd->lds_bytes + 1024 <= 150 * 1024) {      // the tables fit in LDS: the DFA kernels are launched with that much
    if (int r = raise_dfa_lds_attrs(e, rd->lds_bytes + 1024)) { rd->blob.release(); rd->dev.release(); delete rd; return r; }
  }
  e->dfas.push_back(rd);
  *out_id = (uint32_t)(e->dfas.size() - 1);
  return 0;
}

int zke_engine_reserve(zke_engine* e, uint32_t max_n, uint64_t max_raw_total, uint32_t slots, uint32_t max_regex_parts) {
  if (!e || slots == 0 || slots > 64) return ZKE_E_ARG;
  HIPCHK(e, hipSetDevice(e->device));
  while (e->slots.size() < slots) {
    Slot* w = new_slot(e);
    if (!w) return ZKE_E_DEVICE;
    e->slots.push_back(w);
  }
  if (max_n)
    for (Slot* w : e->slots)
      if (int r = ensure_workspace(e, *w, max_n, max_raw_total, max_regex_parts != 0, max_regex_parts, false)) return r;
  // A stream's hardware queue and the queue's scratch memory (the front end spills a few registers) come into being
  // with the first launch that needs them — milliseconds, and they would land in the first batch of every slot.
  // One trivial launch per slot, with more scratch per lane than any kernel of the pipeline, pays for both here.
  // Then one empty e-mail through the whole pipeline of every slot: kernel code, kernel arguments and the slot's workspace
  // pages have all been touched once before the first real batch arrives.
  if (int r = e->misc.ensure(1024)) return fail(e, r, "workspace allocation");
  HIPCHK(e, hipMemset(e->misc.p, 0, 1024));
  zke_batch wb{};
  wb.n = 1;
  uint8_t* z = e->misc.as<uint8_t>();            // 1 KB of zeros: CSR offsets {0, 0}, empty blobs, key type "rsa"
  wb.raw_blob = z + 512; wb.raw_off = reinterpret_cast<const uint64_t*>(z);
  wb.domain_blob = z + 512; wb.domain_off = reinterpret_cast<const uint64_t*>(z);
  wb.key_blob = z + 512; wb.key_off = reinterpret_cast<const uint64_t*>(z);
  wb.key_type = z + 512; wb.ext_null = nullptr;
  const bool timing = e->timing;
  e->timing = false;
  for (Slot* w : e->slots) {
    hipLaunchKernelGGL(slot_warm_kernel, dim3(1), dim3(64), 0, w->stream, (uint32_t*)nullptr);
    if (max_n)
      if (int r = run_device_pipeline(e, *w, &wb, 0, reinterpret_cast<zke_result*>(z + 768), w->stream, false)) { e->timing = timing; return r; }
  }
  e->timing = timing;
  HIPCHK(e, hipGetLastError());
  for (Slot* w : e->slots) HIPCHK(e, hipStreamSynchronize(w->stream));
  return 0;
}

int zke_verify_batch_device(zke_engine* e, const zke_batch* in, uint64_t raw_total, uint64_t domain_total, uint64_t key_total,
                            zke_result* out_dev, void* stream) {
  (void)domain_total;
  if (!e) return ZKE_E_ARG;
  if (!in || (in->n && (!out_dev || !in->raw_blob || !in->raw_off || !in->domain_off || !in->key_off || !in->key_type)))
    return fail(e, ZKE_E_ARG, "zke_verify_batch_device: null pointer");
  if (in->with_regex && ((in->n_header_parts && !in->header_part_ids) || (in->n_body_parts && !in->body_part_ids)))
    return fail(e, ZKE_E_ARG, "zke_verify_batch_device: part-id list is null");
  HIPCHK(e, hipSetDevice(e->device));
  e->batch_key_total = key_total;
  // the part-id lists are small host arrays even in device mode
  e->host_hdr_ids.assign(in->header_part_ids, in->header_part_ids + (in->with_regex ? in->n_header_parts : 0));
  e->host_body_ids.assign(in->body_part_ids, in->body_part_ids + (in->with_regex ? in->n_body_parts : 0));
  // Submission slots are taken round-robin: with S slots, S batches are in flight before a workspace is reused.
  const uint32_t slot = e->next_slot;
  e->next_slot = (slot + 1) % (uint32_t)e->slots.size();
  e->last_slot = slot;
  Slot& w = *e->slots[slot];
  hipStream_t s = stream ? (hipStream_t)stream : w.stream;
  if (int r = acquire_slot(e, w, s)) return r;
  if (!e->use_graphs || e->timing) {
    // Launched eagerly: three kernels per signature round (front end, hash / modexp stage, Ed25519 + verdict)
    if (int r = run_device_pipeline(e, w, in, raw_total, out_dev, s, false)) return r;
    return release_slot(e, w, s);
  }
  // hipGraph replay (opt-in).  A service re-submits batches that live in the same staging buffers: the second time a
  // slot sees a descriptor byte for byte — input pointers and sizes, output pointer, part ids, rounds, key-size hint —
  // its kernel sequence is captured, and replayed from then on.  The key also holds the slot's workspace generation: a
  // graph bakes in the workspace pointers, and a batch that regrew a buffer in between (DevBuf::ensure frees and
  // reallocates) would leave them dangling.  Nothing in the submit path calls hipMalloc / hipFuncSetAttribute once the
  // workspaces are reserved, so the capture contains kernel nodes only.
  std::vector<uint8_t> key(sizeof(zke_batch) + 5 * sizeof(uint64_t) + 4 * (e->host_hdr_ids.size() + e->host_body_ids.size()));
  {
    zke_batch kb = *in;
    kb.header_part_ids = nullptr; kb.body_part_ids = nullptr;          // host arrays: compared by content below
    uint8_t* p = key.data();
    memset(p, 0, key.size());
    memcpy(p, &kb.n, sizeof kb.n);                                     // field by field: the struct's padding is not copied
    size_t o = 8;
    const void* ptrs[] = {kb.raw_blob, kb.raw_off, kb.domain_blob, kb.domain_off, kb.key_blob, kb.key_off, kb.key_type, kb.ext_null,
                          kb.cap_off, kb.cap_str_off, kb.cap_blob, out_dev, s};
    for (const void* q : ptrs) { memcpy(p + o, &q, sizeof q); o += sizeof q; }
    const uint64_t nums[] = {raw_total, key_total, ((uint64_t)kb.with_regex << 32) | e->max_sig_rounds, ((uint64_t)kb.n_header_parts << 32) | kb.n_body_parts,
                             w.generation};
    (void)o;
    uint8_t* tail = p + sizeof(zke_batch);
    memcpy(tail, nums, sizeof nums);
    tail += sizeof nums;
    if (!e->host_hdr_ids.empty()) memcpy(tail, e->host_hdr_ids.data(), 4 * e->host_hdr_ids.size());
    tail += 4 * e->host_hdr_ids.size();
    if (!e->host_body_ids.empty()) memcpy(tail, e->host_body_ids.data(), 4 * e->host_body_ids.size());
  }
  static_assert(8 + 13 * sizeof(void*) <= sizeof(zke_batch), "graph key layout");
  if (w.graph_exec && key == w.graph_key) {
    HIPCHK(e, hipGraphLaunch(w.graph_exec, s));
    return release_slot(e, w, s);
  }
  if (w.graph_exec) { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; }
  if (key != w.graph_key) {               // first sighting: run eagerly (this is also what sizes the workspaces)
    if (int r = run_device_pipeline(e, w, in, raw_total, out_dev, s, false)) return r;
    // the generation may have moved: remember the key as it is now, so that an identical second call captures
    const uint64_t gen = w.generation;
    memcpy(key.data() + sizeof(zke_batch) + 4 * sizeof(uint64_t), &gen, sizeof gen);
    w.graph_key = key;
    return release_slot(e, w, s);
  }
  hipGraph_t g = nullptr;
  HIPCHK(e, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int r = run_device_pipeline(e, w, in, raw_total, out_dev, s, false);
  const hipError_t ce = hipStreamEndCapture(s, &g);
  if (r || ce != hipSuccess || !g) {
    if (g) (void)hipGraphDestroy(g);
    e->use_graphs = false;                // capture is an optimisation only: fall back to eager launches
    if (r) return r;
    if (int r2 = run_device_pipeline(e, w, in, raw_total, out_dev, s, false)) return r2;
    return release_slot(e, w, s);
  }
  hipGraphExec_t ge = nullptr;
  const hipError_t ie = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (ie != hipSuccess || !ge) {
    e->use_graphs = false;
    if (int r2 = run_device_pipeline(e, w, in, raw_total, out_dev, s, false)) return r2;
    return release_slot(e, w, s);
  }
  w.graph_exec = ge;
  HIPCHK(e, hipGraphLaunch(w.graph_exec, s));
  return release_slot(e, w, s);
}

int zke_verify_batch(zke_engine* e, const zke_batch* in, zke_result* out, zke_debug_out* dbg) {
  if (!e || !in || (in->n && (!out || !in->raw_blob || !in->raw_off || !in->domain_blob || !in->domain_off || !in->key_blob ||
                              !in->key_off || !in->key_type)))
    return ZKE_E_ARG;
  const uint32_t n = in->n;
  if (n == 0) return 0;
  if (in->with_regex && ((in->n_header_parts && !in->header_part_ids) || (in->n_body_parts && !in->body_part_ids)))
    return fail(e, ZKE_E_ARG, "zke_verify_batch: part-id list is null");
  HIPCHK(e, hipSetDevice(e->device));
  Slot& w = *e->slots[0];            // host-mode batches are synchronous: always slot 0, on its own stream
  hipStream_t s = w.stream;
  e->last_slot = 0;
  if (int ar = acquire_slot(e, w, s)) return ar;
  w.last_stream = s;
  const uint64_t raw_total = in->raw_off[n] - in->raw_off[0], dom_total = in->domain_off[n] - in->domain_off[0],
                 key_total = in->key_off[n] - in->key_off[0];
  e->batch_key_total = key_total;
  const uint32_t P = in->with_regex ? in->n_header_parts + in->n_body_parts : 0;
  const bool caps = P && in->cap_off;
  const uint32_t n_caps = caps ? in->cap_off[(size_t)n * P] : 0;
  const uint32_t cap_bytes = caps ? in->cap_str_off[n_caps] : 0;
  int r = 0;
  if ((r = e->in_raw.ensure(raw_total + 64)) || (r = e->in_raw_off.ensure((size_t)(n + 1) * 8)) ||
      (r = e->in_dom.ensure(dom_total + 64)) || (r = e->in_dom_off.ensure((size_t)(n + 1) * 8)) ||
      (r = e->in_key.ensure(key_total + 64)) || (r = e->in_key_off.ensure((size_t)(n + 1) * 8)) ||
      (r = e->in_ktype.ensure(n)) || (r = e->in_extnull.ensure(n)) || (r = e->results.ensure((size_t)n * sizeof(zke_result))))
    return fail(e, r, "input allocation");
  if (caps && ((r = e->in_cap_off.ensure(((size_t)n * P + 1) * 4)) || (r = e->in_cap_str_off.ensure(((size_t)n_caps + 1) * 4)) ||
               (r = e->in_cap_blob.ensure((size_t)cap_bytes + 64))))
    return fail(e, r, "input allocation");

  hipEvent_t h0 = e->ev_h2d[0], h1 = e->ev_h2d[1], d0 = e->ev_h2d[2], d1 = e->ev_h2d[3];
  if (e->timing) (void)hipEventRecord(h0, s);
  // rebase the CSR offsets to 0 on the way in
  std::vector<uint64_t> ro(n + 1), dofs(n + 1), ko(n + 1);
  for (uint32_t i = 0; i <= n; i++) { ro[i] = in->raw_off[i] - in->raw_off[0]; dofs[i] = in->domain_off[i] - in->domain_off[0]; ko[i] = in->key_off[i] - in->key_off[0]; }
  if (raw_total) HIPCHK(e, hipMemcpyAsync(e->in_raw.p, in->raw_blob + in->raw_off[0], raw_total, hipMemcpyHostToDevice, s));
  if (dom_total) HIPCHK(e, hipMemcpyAsync(e->in_dom.p, in->domain_blob + in->domain_off[0], dom_total, hipMemcpyHostToDevice, s));
  if (key_total) HIPCHK(e, hipMemcpyAsync(e->in_key.p, in->key_blob + in->key_off[0], key_total, hipMemcpyHostToDevice, s));
  HIPCHK(e, hipMemcpyAsync(e->in_raw_off.p, ro.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
  HIPCHK(e, hipMemcpyAsync(e->in_dom_off.p, dofs.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
  HIPCHK(e, hipMemcpyAsync(e->in_key_off.p, ko.data(), (size_t)(n + 1) * 8, hipMemcpyHostToDevice, s));
  HIPCHK(e, hipMemcpyAsync(e->in_ktype.p, in->key_type, n, hipMemcpyHostToDevice, s));
  if (in->ext_null) HIPCHK(e, hipMemcpyAsync(e->in_extnull.p, in->ext_null, n, hipMemcpyHostToDevice, s));
  if (caps) {
    HIPCHK(e, hipMemcpyAsync(e->in_cap_off.p, in->cap_off, ((size_t)n * P + 1) * 4, hipMemcpyHostToDevice, s));
    HIPCHK(e, hipMemcpyAsync(e->in_cap_str_off.p, in->cap_str_off, ((size_t)n_caps + 1) * 4, hipMemcpyHostToDevice, s));
    if (cap_bytes) HIPCHK(e, hipMemcpyAsync(e->in_cap_blob.p, in->cap_blob, cap_bytes, hipMemcpyHostToDevice, s));
  }
  // the copies above read pageable host memory that goes out of scope (ro/dofs/ko): make them complete first
  HIPCHK(e, hipStreamSynchronize(s));
  if (e->timing) (void)hipEventRecord(h1, s);

  zke_batch dv = *in;
  dv.raw_blob = e->in_raw.as<uint8_t>(); dv.raw_off = e->in_raw_off.as<uint64_t>();
  dv.domain_blob = e->in_dom.as<uint8_t>(); dv.domain_off = e->in_dom_off.as<uint64_t>();
  dv.key_blob = e->in_key.as<uint8_t>(); dv.key_off = e->in_key_off.as<uint64_t>();
  dv.key_type = e->in_ktype.as<uint8_t>();
  dv.ext_null = in->ext_null ? e->in_extnull.as<uint8_t>() : nullptr;
  dv.cap_off = caps ? e->in_cap_off.as<uint32_t>() : nullptr;
  dv.cap_str_off = caps ? e->in_cap_str_off.as<uint32_t>() : nullptr;
  dv.cap_blob = caps ? e->in_cap_blob.as<uint8_t>() : nullptr;
  e->host_hdr_ids.assign(in->header_part_ids, in->header_part_ids + (in->with_regex ? in->n_header_parts : 0));
  e->host_body_ids.assign(in->body_part_ids, in->body_part_ids + (in->with_regex ? in->n_body_parts : 0));
  const bool want_em = dbg && dbg->em;
  if ((r = run_device_pipeline(e, w, &dv, raw_total, e->results.as<zke_result>(), s, want_em))) return r;
  if (e->timing) (void)hipEventRecord(d0, s);
  HIPCHK(e, hipMemcpyAsync(out, e->results.p, (size_t)n * sizeof(zke_result), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipStreamSynchronize(s));
  if (e->timing) {
    (void)hipEventRecord(d1, s);
    (void)hipEventSynchronize(d1);
    collect_timings(e, w);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, h0, h1); e->last.h2d_us = ms * 1000.f;
    (void)hipEventElapsedTime(&ms, d0, d1); e->last.d2h_us = ms * 1000.f;
  }

  if (dbg) {   // parity intermediates: copy the scratch back and slice it on the host
    std::vector<EmailMeta> meta(n), meta2;
    HIPCHK(e, hipMemcpy(meta.data(), w.meta.p, (size_t)n * sizeof(EmailMeta), hipMemcpyDeviceToHost));
    const size_t scratch_bytes = 2 * (size_t)raw_total + (size_t)(n + 1) * SCR_PER_EMAIL + 256;
    std::vector<uint8_t> scr(scratch_bytes);
    HIPCHK(e, hipMemcpy(scr.data(), w.scratch.p, scratch_bytes, hipMemcpyDeviceToHost));
    std::vector<uint8_t> em, clean;
    if (dbg->em) { em.resize((size_t)n * 512); HIPCHK(e, hipMemcpy(em.data(), w.em_dbg.p, em.size(), hipMemcpyDeviceToHost)); }
    if (dbg->clean_body && in->with_regex) {
      meta2.resize(n);
      HIPCHK(e, hipMemcpy(meta2.data(), w.meta2.p, (size_t)n * sizeof(EmailMeta), hipMemcpyDeviceToHost));
      clean.resize((size_t)raw_total + (size_t)(n + 1) * CLEAN_PER_EMAIL + 256);
      HIPCHK(e, hipMemcpy(clean.data(), w.clean.p, clean.size(), hipMemcpyDeviceToHost));
    }
    auto put = [](uint8_t* base, size_t stride, uint32_t i, const uint8_t* src, size_t len) {
      if (!base) return;
      memset(base + (size_t)i * stride, 0, stride);
      memcpy(base + (size_t)i * stride, src, std::min(len, stride));
    };
    for (uint32_t i = 0; i < n; i++) {
      const EmailMeta& m = meta[i];
      const uint32_t raw_len = (uint32_t)(ro[i + 1] - ro[i]);
      const uint8_t* regA = scr.data() + host_scratch_off(ro.data(), i);
      const uint8_t* regB = regA + (((size_t)raw_len + PRE_SLACK + 15) & ~(size_t)15);
      const bool hashed = out[i].canon_header_len || out[i].canon_body_len || m.canon_full_len;
      put(dbg->canon_header, dbg->canon_header_stride, i, regA, hashed ? out[i].canon_header_len : 0);
      const uint8_t* body = m.body_src_is_raw ? in->raw_blob + in->raw_off[i] + m.body_off : regB;
      put(dbg->canon_body, dbg->canon_body_stride, i, body, hashed ? m.canon_full_len : 0);
      if (dbg->canon_body_full_len) dbg->canon_body_full_len[i] = hashed ? m.canon_full_len : 0;
      if (dbg->rsa_route) dbg->rsa_route[i] = m.rsa_route;
      if (dbg->em) put(dbg->em, dbg->em_stride, i, em.data() + (size_t)i * 512 + 512 - std::min<uint32_t>(512, e_k(out[i].rsa_bits)),
                       std::min<uint32_t>(512, e_k(out[i].rsa_bits)));
      if (dbg->clean_body && in->with_regex && meta2[i].state == ST_CAND)
        put(dbg->clean_body, dbg->clean_body_stride, i, clean.data() + (ro[i] + (uint64_t)i * CLEAN_PER_EMAIL), meta2[i].hashed_len);
      else if (dbg->clean_body)
        put(dbg->clean_body, dbg->clean_body_stride, i, nullptr, 0);
    }
  }
  return 0;
}

// ---- single-e-mail wrappers: a batch of one (config 1 and API-shape parity)
namespace {
struct OneEmail {
  uint64_t ro[2], dofs[2], ko[2];
  uint8_t kt, ext;
  zke_batch b{};
  OneEmail(const uint8_t* raw, size_t raw_len, const char* from_domain, size_t domain_len, const uint8_t* key, size_t key_len,
           uint32_t key_type, uint32_t external_input_null)
      : ro{0, raw_len}, dofs{0, domain_len}, ko{0, key_len}, kt((uint8_t)(key_type > ZKE_KEY_OTHER ? ZKE_KEY_OTHER : key_type)),
        ext(external_input_null ? 1 : 0) {
    static const uint8_t dummy = 0;
    b.n = 1;
    b.raw_blob = raw ? raw : &dummy; b.raw_off = ro;
    b.domain_blob = from_domain ? reinterpret_cast<const uint8_t*>(from_domain) : &dummy; b.domain_off = dofs;
    b.key_blob = key ? key : &dummy; b.key_off = ko;
    b.key_type = &kt; b.ext_null = &ext;
  }
};
}  // namespace

int zke_verify_email(zke_engine* e, const uint8_t* raw, size_t raw_len, const char* from_domain, size_t domain_len,
                     const uint8_t* key, size_t key_len, uint32_t key_type, uint32_t external_input_null, zke_result* out) {
  if (!e) return ZKE_E_ARG;
  if (!out || (raw_len && !raw) || (domain_len && !from_domain) || (key_len && !key)) return fail(e, ZKE_E_ARG, "zke_verify_email: null pointer");
  OneEmail one(raw, raw_len, from_domain, domain_len, key, key_len, key_type, external_input_null);
  return zke_verify_batch(e, &one.b, out, nullptr);
}

int zke_verify_email_with_regex(zke_engine* e, const uint8_t* raw, size_t raw_len, const char* from_domain, size_t domain_len,
                                const uint8_t* key, size_t key_len, uint32_t key_type, uint32_t external_input_null,
                                const zke_regex_part* header_parts, uint32_t n_header_parts,
                                const zke_regex_part* body_parts, uint32_t n_body_parts, zke_result* out) {
  if (!e) return ZKE_E_ARG;
  if (!out || (raw_len && !raw) || (domain_len && !from_domain) || (key_len && !key) || (n_header_parts && !header_parts) ||
      (n_body_parts && !body_parts))
    return fail(e, ZKE_E_ARG, "zke_verify_email_with_regex: null pointer");
  OneEmail one(raw, raw_len, from_domain, domain_len, key, key_len, key_type, external_input_null);
  std::vector<uint32_t> hids, bids, cap_off{0}, str_off{0};
  std::vector<uint8_t> blob;
  for (int side = 0; side < 2; side++) {
    const zke_regex_part* parts = side ? body_parts : header_parts;
    const uint32_t np = side ? n_body_parts : n_header_parts;
    for (uint32_t k = 0; k < np; k++) {
      const zke_regex_part& p = parts[k];
      if ((p.fwd_len && !p.fwd) || (p.bwd_len && !p.bwd) || (p.n_captures && (!p.captures || !p.capture_lens)))
        return fail(e, ZKE_E_ARG, "zke_verify_email_with_regex: null pointer in a part");
      uint32_t id = 0;
      if (int r = zke_dfa_register(e, p.fwd, p.fwd_len, p.bwd, p.bwd_len, &id)) return r;     // the same pair gets the same id
      (side ? bids : hids).push_back(id);
      for (uint32_t c = 0; c < p.n_captures; c++) {
        if (p.capture_lens[c] && !p.captures[c]) return fail(e, ZKE_E_ARG, "zke_verify_email_with_regex: null capture");
        blob.insert(blob.end(), p.captures[c], p.captures[c] + p.capture_lens[c]);
        str_off.push_back((uint32_t)blob.size());
      }
      cap_off.push_back((uint32_t)str_off.size() - 1);
    }
  }
  if (blob.empty()) blob.push_back(0);
  zke_batch& b = one.b;
  b.with_regex = 1;
  b.n_header_parts = n_header_parts; b.n_body_parts = n_body_parts;
  b.header_part_ids = hids.data(); b.body_part_ids = bids.data();
  b.cap_off = cap_off.data(); b.cap_str_off = str_off.data(); b.cap_blob = blob.data();
  return zke_verify_batch(e, &b, out, nullptr);
}

}  // extern "C"
