// pipeline.hip.h — host orchestration of one batch: workspace sizing, the kernel sequence of
// verify_email / verify_email_with_regex (core/src/circuits.rs:9-68), the submission entry points (device-resident and
// host-memory batches), DFA registration.  Included by engine.hip (single translation unit).
#pragma once

namespace {

inline uint64_t host_scratch_off(const uint64_t* raw_off, uint32_t i) { return scratch_offset(raw_off[i] - raw_off[0], i); }

struct StageTimer {
  Slot* w; hipStream_t s; bool on;
  StageTimer(zke_engine* e_, Slot* w_, hipStream_t s_) : w(w_), s(s_), on(e_->timing.load()) {}
  void mark(int k) { if (on && hipEventRecord(w->ev[k], s) == hipSuccess) w->marks |= 1u << k; }
};

// Length-bucket counters start a batch at zero (the verdict launch of the slot's previous batch clears them): a fresh
// allocation is cleared here.
int ensure_zeroed(DevBuf& b, size_t need) {
  const void* old = b.p;
  if (int r = b.ensure(need)) return r;
  if (b.p != old && hipMemset(b.p, 0, b.cap) != hipSuccess) return ZKE_E_DEVICE;
  return 0;
}

// Workspace of one slot for batches of up to n e-mails / raw_total raw bytes (P regex parts; with_regex: the buffers of
// the canonicalize_signed_email pass too).  Grows only: in steady state — or after zke_engine_reserve — this allocates nothing.
int ensure_workspace(zke_engine* e, Slot& w, uint32_t n, uint64_t raw_total, bool with_regex, uint32_t P, bool want_em) {
  const uint32_t n_pad = (n + 63) & ~63u;
  int r = 0;
  const size_t scratch_bytes = 2 * (size_t)raw_total + (size_t)(n + 1) * SCR_PER_EMAIL + 256;
  if ((r = w.meta.ensure((size_t)n * sizeof(EmailMeta))) || (r = w.rsa_jobs.ensure((size_t)n * sizeof(RsaJob))) ||
      (r = w.sha_jobs.ensure((size_t)4 * n_pad * sizeof(ShaJob))) || (r = w.rsa_ok.ensure((size_t)n * 4)) ||
      (r = ensure_zeroed(w.sha_order, (size_t)2 * (SHA_ORDER_KIND_WORDS + n_pad) * 4)) ||
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

// Host-entry buffers of one slot: the packed input image (pinned and in HBM) and the records (HBM and pinned).
int ensure_host_buffers(zke_engine* e, Slot& w, size_t image_bytes, uint32_t n) {
  int r = 0;
  if ((r = w.h_image.ensure(image_bytes)) || (r = w.d_image.ensure(image_bytes)) ||
      (r = w.h_results.ensure((size_t)n * sizeof(zke_result))) || (r = w.d_results.ensure((size_t)n * sizeof(zke_result))))
    return fail(e, r, "host-entry staging allocation");
  return 0;
}

// (caller holds reg_mu exclusively or is creating the engine)
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

// The device pipeline.  Every pointer in `in` / out_dev is device memory, except the part-id lists (host arrays in both
// entry points).  Three launches — front end, hash / modexp stage, Ed25519 + verdict (which also runs the later signature
// rounds of the rare e-mail that needs them) — and, for verify_email_with_regex, the regex stage behind them.
// Caller holds the slot's lock; everything the call needs is in its arguments or in the slot.
int run_device_pipeline(zke_engine* e, Slot& w, const zke_batch* in, uint64_t raw_total, uint64_t key_total, zke_result* out_dev,
                        hipStream_t s, bool want_em, uint64_t now, bool want_clean = false) {
  const uint32_t n = in->n;
  if (n == 0) return 0;
  const uint32_t n_pad = (n + 63) & ~63u;
  const uint32_t P = in->with_regex ? in->n_header_parts + in->n_body_parts : 0;
  int r = 0;
  if ((r = ensure_workspace(e, w, n, raw_total, in->with_regex != 0, P, want_em))) return r;
  uint64_t* scratch_off = w.scratch_off.as<uint64_t>();
  uint64_t* clean_off = scratch_off + (n + 1);

  StageTimer tm(e, &w, s);
  if (!(w.marks & (1u << MK_START))) tm.mark(MK_START);      // (the host entry has marked the start in front of its H2D)
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
  B.order = n < (1u << 24) ? w.sha_order.as<uint32_t>() : nullptr;      // a key holds the position within a class in 24 bits

  const uint32_t rounds = e->opt.max_sig_rounds;
  // an RSA-2048 key is 270 bytes of DER: a batch whose keys average more holds some larger modulus
  const uint32_t route_mask = rsa_route_mask(e, n, key_total > (uint64_t)n * 272);
  {
    const uint32_t round = 0;
    uint32_t* wave_count = w.pending.as<uint32_t>() + 2;
    uint32_t* wave_list = w.rsa_ok.as<uint32_t>();
    ParseArgs pa{B, round, 0, e->debug_parse_stop, e->strict, now, e->key_cache.as<KeyCacheEntry>(), route_mask, wave_count, wave_list};
    hipLaunchKernelGGL(parse_kernel, dim3((n + ZKE_PARSE_WG_WAVES - 1) / ZKE_PARSE_WG_WAVES), dim3(64 * ZKE_PARSE_WG_WAVES), PARSE_DYN_LDS, s, pa);
    tm.mark(MK_FRONT);
    // hash / modexp stage: the four SHA-256 jobs and the RSA operation of every e-mail, one launch (fused.hip.h)
    if (!(e->debug_skip_launch & 1) &&
        (r = launch_hash_modexp(e, B.sha, 4 * n_pad, B.rsa, n, B.meta, want_em ? w.em_dbg.as<uint8_t>() : nullptr, route_mask, wave_count, wave_list, B.order,
                                __atomic_load_n(w.wave_feedback, __ATOMIC_RELAXED), s)))
      return r;
    tm.mark(MK_HASH);
    // Ed25519 stage + verdicts (verdict.hip.h): bh compare, EM digest against the header hash, status / detail, pending counter
    EdVerdictArgs va{FinArgs{B, round, rounds, w.pending.as<uint32_t>(), e->debug_skip_rsa}, e->debug_skip_ed, wave_count,
                     e->key_cache.as<KeyCacheEntry>(), want_em ? w.em_dbg.as<uint8_t>() : nullptr, e->strict, now, w.wave_feedback};
    if (!(e->debug_skip_launch & 2)) hipLaunchKernelGGL(ed_verdict_kernel, dim3((n + VERDICT_EMAILS_PER_WAVE - 1) / VERDICT_EMAILS_PER_WAVE), dim3(64), 0, s, va);
    tm.mark(MK_VERDICT);
  }
  HIPCHK(e, hipGetLastError());

  if (in->with_regex) {
    // the parts of this batch, copied out of the registry (the entries themselves stay put until the engine is idle)
    PartInfo parts_small[16];
    std::vector<PartInfo> parts_big;
    PartInfo* parts = parts_small;
    if (P > 16) { parts_big.resize(P); parts = parts_big.data(); }
    {
      std::shared_lock<std::shared_mutex> rl(e->reg_mu);
      for (uint32_t p = 0; p < P; p++) {
        const uint32_t id = p >= in->n_header_parts ? in->body_part_ids[p - in->n_header_parts] : in->header_part_ids[p];
        PartInfo pi{};
        const RegisteredDfa* rd = id < e->dfas.size() ? e->dfas[id] : nullptr;
        if (rd) {
          pi.detail = rd->detail; pi.lds_bytes = rd->lds_bytes; pi.idle = rd->idle;
          pi.dev = rd->valid ? rd->dev.as<RegexDev>() : nullptr;
        }
        parts[p] = pi;
      }
    }
    // canonicalize_signed_email (circuits.rs:34-35: first DKIM-Signature header, own scratch unless it is the verified one) and
    // remove_quoted_printable_soft_breaks (circuits.rs:37), one launch (regex.hip.h, regex_prep_kernel).  The cleaned body is
    // unobservable without body parts (circuits.rs:48-56): it is then only produced when a parity buffer asks for it.
    BatchDev B2 = B;
    B2.meta = w.meta2.as<EmailMeta>();
    B2.scratch = w.scratch2.as<uint8_t>();
    B2.meta_verify = B.meta;
    B2.order = nullptr;
    PrepArgs pr{ParseArgs{B2, 0, 1, 0, e->strict, now, nullptr, 0, nullptr, nullptr},
                QpArgs{B2, B.meta, w.clean.as<uint8_t>(), clean_off, B.scratch, B.scratch_off},
                (in->n_body_parts || want_clean) ? 1u : 0u};
    hipLaunchKernelGGL(regex_prep_kernel, dim3(n), dim3(64), 0, s, pr);
    tm.mark(MK_PREP);
    DfaArgs base{};
    base.b = B2; base.P = P;
    base.scratch_v = B.scratch; base.scratch_v_off = B.scratch_off;
    base.clean = w.clean.as<uint8_t>(); base.clean_off = clean_off;
    base.cap_off = in->cap_off; base.cap_str_off = in->cap_str_off; base.cap_blob = in->cap_blob;
    base.out = w.parts.as<PartRes>();
    auto part_lds = [&](const PartInfo& pi, uint32_t& in_lds) -> size_t {
      in_lds = 0;
      if (pi.dev && pi.lds_bytes + 1024 <= 150 * 1024) { in_lds = 1; return pi.lds_bytes + 1024; }
      return 1024;
    };
    // Which kernel: the wave-per-e-mail kernel shortens the chain (latency) but runs its serial part on one
    // lane's worth of work per wave, so it issues several times the instructions of the lane-per-e-mail kernel.
    // Body parts (KBs per e-mail) always gain; header parts (~1 KB) gain only while the batch is small enough for
    // latency to be what matters (measured: configs[2] shape, 4 096 per batch, 16.0 M e-mails/s with the lane kernel
    // against 12.7 M with the wave kernel; 1 024 per batch 11.4 M against 11.8 M).  zke_options.dfa_mapping forces one.
    const uint32_t wave_from = e->opt.dfa_mapping == 1 ? P : e->opt.dfa_mapping == 2 ? 0u
                             : (n <= 1024 ? 0u : in->n_header_parts);      // parts [wave_from, P) use the wave kernel
    // Parts in order, up to DFA_MULTI_MAX per launch (grid.y): [0, wave_from) one e-mail per lane, [wave_from, P) one e-mail per
    // wave.  A last launch that holds ONE part also writes the regex verdict into the records (it folds the earlier launches'
    // parts first); otherwise the verdict is a small launch of its own behind them.  (The parts of a launch run side by side:
    // walking them one after the other inside a block, verdict folded in, was measured — configs[4] shape 10.2 M e-mails/s
    // against 11.9 M, its dfa stage 505 us alone against 387.)
    bool folded = false;
    for (uint32_t p0 = 0; p0 < P;) {
      const bool lane_kernel = p0 < wave_from;
      const uint32_t end = lane_kernel ? wave_from : P;
      const uint32_t np = std::min<uint32_t>(DFA_MULTI_MAX, end - p0);
      DfaMultiArgs ma{};
      ma.common = base; ma.part0 = p0; ma.np = np; ma.n_header_parts = in->n_header_parts;
      ma.finalize = (p0 + np >= P && np == 1) ? 1u : 0u;
      folded = ma.finalize != 0;
      size_t lds = 1024;
      for (uint32_t k = 0; k < np; k++) {
        const PartInfo& pi = parts[p0 + k];
        ma.re[k] = pi.dev;
        lds = std::max(lds, part_lds(pi, ma.lds_tables[k]));
        ma.idle[k] = (pi.dev && !lane_kernel) ? pi.idle : 0xFFFFFFFFu;
        ma.detail[k] = pi.detail;
      }
      if (lane_kernel) hipLaunchKernelGGL(dfa_kernel, dim3((n + 255) / 256, np), dim3(256), lds, s, ma);
      else hipLaunchKernelGGL(dfa_wave_kernel, dim3((n + 3) / 4, np), dim3(256), lds, s, ma);
      p0 += np;
    }
    if (!folded) {
      RegexFinArgs rf{B2, w.parts.as<PartRes>(), in->n_header_parts, in->n_body_parts};
      hipLaunchKernelGGL(regex_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, s, rf);
    }
    tm.mark(MK_DFA);
    HIPCHK(e, hipGetLastError());
  }
  return 0;
}

// Read the slot's timing marks (their events have completed) into w.last.
void collect_timings(Slot& w) {
  auto has = [&](int k) { return (w.marks >> k) & 1u; };
  auto dt = [&](int a, int b) { float ms = 0; if (hipEventElapsedTime(&ms, w.ev[a], w.ev[b]) != hipSuccess) return 0.f; return ms * 1000.f; };
  zke_timings t{};
  if (has(MK_START) && has(MK_FRONT) && has(MK_HASH) && has(MK_VERDICT)) {
    const int k0 = has(MK_H2D) ? MK_H2D : MK_START;          // where the first launch starts
    if (has(MK_H2D)) t.h2d_us = dt(MK_START, MK_H2D);
    t.front_end_us = dt(k0, MK_FRONT);
    t.hash_modexp_us = dt(MK_FRONT, MK_HASH);
    t.ed_verdict_us = dt(MK_HASH, MK_VERDICT);
    int klast = MK_VERDICT;
    if (has(MK_PREP) && has(MK_DFA)) { t.regex_prep_us = dt(MK_VERDICT, MK_PREP); t.dfa_us = dt(MK_PREP, MK_DFA); klast = MK_DFA; }
    t.total_us = dt(k0, klast);
    if (has(MK_D2H)) t.d2h_us = dt(klast, MK_D2H);
  }
  w.last = t;
  w.marks = 0;
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
  w.marks = 0;
  return 0;
}
int release_slot(zke_engine* e, Slot& w, hipStream_t s) {
  // the event is needed only where the next user of the slot may sit on another stream: a caller-provided stream
  if (s != w.stream) HIPCHK(e, hipEventRecord(w.done, s));
  w.last_stream = s;
  return 0;
}
// A batch's use of a slot, from acquire to release.  The release runs on EVERY way out — also when a launch failed after
// others were enqueued: the slot's next user (and zke_engine_join / zke_engine_sync) must be ordered behind whatever did
// reach the stream, or its workspace is overwritten under a half-launched batch.
struct SlotUse {
  zke_engine* e; Slot& w; hipStream_t s; bool armed = false;
  SlotUse(zke_engine* e_, Slot& w_, hipStream_t s_) : e(e_), w(w_), s(s_) {}
  int acquire() { const int r = acquire_slot(e, w, s); armed = (r == 0); return r; }
  int release() { armed = false; return release_slot(e, w, s); }
  ~SlotUse() { if (armed) { const std::string keep = g_err; (void)release_slot(e, w, s); g_err = keep; } }      // the first error is the one reported
};

// The slot's next submission ticket (round-robin over the slots that exist: zke_engine_reserve only appends).
Slot& next_slot(zke_engine* e, uint32_t& index) {
  index = e->ticket.fetch_add(1, std::memory_order_relaxed) % (uint32_t)e->slots.size();
  e->last_slot.store(index, std::memory_order_relaxed);
  return *e->slots[index];
}

uint64_t batch_clock(const zke_engine* e) {
  if (!(e->strict & ZKE_STRICT_EXPIRY_X)) return 0;
  return e->opt.now_unix ? e->opt.now_unix : (uint64_t)time(nullptr);
}

// Deliver a host batch that was enqueued in this slot and not waited for yet: wait for its D2H, copy the records from the
// pinned buffer to the caller's `out`.  Caller holds the slot's lock.
int retire_host(zke_engine* e, Slot& w) {
  if (w.host_retired == w.host_gen) return 0;
  w.host_retired = w.host_gen;                       // whatever happens below, the batch is no longer pending
  HIPCHK(e, hipEventSynchronize(w.host_done));
  if (w.host_out && w.host_n) memcpy(w.host_out, w.h_results.p, (size_t)w.host_n * sizeof(zke_result));
  w.host_out = nullptr;
  return 0;
}

// Sizes of a batch handed over as separate e-mails (zke_verify_emails): summed once, by the entry point
struct RefTotals { uint64_t raw = 0, dom = 0, key = 0; };

// One host-memory batch into slot w (caller holds its lock): pack -> one H2D -> the launches -> one D2H -> event.
// refs != nullptr: the e-mails come one by one (zke_email_ref); `in` then carries n only and the offsets are made here.
int submit_host(zke_engine* e, Slot& w, const zke_batch* in, zke_result* out, bool want_em, bool want_clean,
                const zke_email_ref* refs = nullptr, const RefTotals* rt = nullptr) {
  const uint32_t n = in->n;
  const uint64_t raw_total = refs ? rt->raw : in->raw_off[n] - in->raw_off[0], dom_total = refs ? rt->dom : in->domain_off[n] - in->domain_off[0],
                 key_total = refs ? rt->key : in->key_off[n] - in->key_off[0];
  const uint64_t raw_base = refs ? 0 : in->raw_off[0], dom_base = refs ? 0 : in->domain_off[0], key_base = refs ? 0 : in->key_off[0];
  const uint32_t P = in->with_regex ? in->n_header_parts + in->n_body_parts : 0;
  const bool caps = P && in->cap_off;
  const uint32_t n_caps = caps ? in->cap_off[(size_t)n * P] : 0;
  const uint32_t cap_bytes = caps ? in->cap_str_off[n_caps] : 0;
  const ImageLayout L = image_layout(n, raw_total, dom_total, key_total, caps ? (size_t)n * P + 1 : 0, caps ? (size_t)n_caps + 1 : 0, cap_bytes);
  if (int r = retire_host(e, w)) return r;           // the pinned buffers are about to be overwritten
  if (int r = ensure_host_buffers(e, w, L.total, n)) return r;      // (a no-op: the entry points have grown every slot's staging)
  uint8_t* hp = w.h_image.as<uint8_t>();
  if (refs) {
    // the CSR arrays are written where they will be read from (prefix sums over the lengths), and every e-mail's three buffers
    // go to their places in the blobs — the pool takes runs of consecutive e-mails (CopyPool::gather)
    uint64_t* ro = reinterpret_cast<uint64_t*>(hp + L.raw_off), *dofs = reinterpret_cast<uint64_t*>(hp + L.dom_off), *ko = reinterpret_cast<uint64_t*>(hp + L.key_off);
    uint8_t* kt = hp + L.key_type, *xn = hp + L.ext_null;
    w.gather.resize((size_t)3 * n);
    uint64_t r = 0, d = 0, k = 0;
    for (uint32_t i = 0; i < n; i++) {
      const zke_email_ref& m = refs[i];
      ro[i] = r; dofs[i] = d; ko[i] = k;
      kt[i] = (uint8_t)(m.key_type > ZKE_KEY_OTHER ? ZKE_KEY_OTHER : m.key_type);
      xn[i] = m.external_input_null ? 1 : 0;
      w.gather[i] = CopyPool::Piece{hp + L.raw + r, m.raw, m.raw_len};                       // three runs, each contiguous in the image
      w.gather[(size_t)n + i] = CopyPool::Piece{hp + L.dom + d, m.from_domain, m.domain_len};
      w.gather[2 * (size_t)n + i] = CopyPool::Piece{hp + L.key + k, m.key, m.key_len};
      r += m.raw_len; d += m.domain_len; k += m.key_len;
    }
    ro[n] = r; dofs[n] = d; ko[n] = k;
    if (e->pool) e->pool->gather(w.gather.data(), w.gather.size());
    else for (const auto& p : w.gather) if (p.n) stage_copy(p.dst, p.src, p.n, ZKE_GATHER_STREAM_FROM);
    if (caps) {       // the capture tables of a regex batch arrive in zke_batch's form (small)
      stage_copy(hp + L.cap_off, in->cap_off, ((size_t)n * P + 1) * 4);
      stage_copy(hp + L.cap_str_off, in->cap_str_off, ((size_t)n_caps + 1) * 4);
      if (cap_bytes) stage_copy(hp + L.cap_blob, in->cap_blob, cap_bytes);
    }
  } else {
    // the offsets are copied as they are (the kernels subtract off[0] themselves and the device pointers below are biased
    // by -off[0]): nothing is rebased, nothing is allocated, every byte is written once
    CopyPool::Piece pc[11] = {
        {hp + L.raw_off, in->raw_off, (size_t)(n + 1) * 8}, {hp + L.dom_off, in->domain_off, (size_t)(n + 1) * 8},
        {hp + L.key_off, in->key_off, (size_t)(n + 1) * 8}, {hp + L.key_type, in->key_type, n},
        {hp + L.ext_null, in->ext_null, in->ext_null ? n : 0u},
        {hp + L.cap_off, in->cap_off, caps ? ((size_t)n * P + 1) * 4 : 0}, {hp + L.cap_str_off, in->cap_str_off, caps ? ((size_t)n_caps + 1) * 4 : 0},
        {hp + L.raw, in->raw_blob + in->raw_off[0], (size_t)raw_total}, {hp + L.dom, in->domain_blob + in->domain_off[0], (size_t)dom_total},
        {hp + L.key, in->key_blob + in->key_off[0], (size_t)key_total}, {hp + L.cap_blob, in->cap_blob, caps ? (size_t)cap_bytes : 0}};
    if (e->pool) e->pool->copy(pc, 11);
    else for (const auto& p : pc) if (p.n) stage_copy(p.dst, p.src, p.n);
  }
  hipStream_t s = w.stream;
  SlotUse use(e, w, s);
  if (int r = use.acquire()) return r;
  StageTimer tm(e, &w, s);
  tm.mark(MK_START);
#ifndef ZKE_HOST_COPY_STREAM
#define ZKE_HOST_COPY_STREAM 1
#endif
  if (ZKE_HOST_COPY_STREAM && L.total >= (256u << 10)) {
    // The image crosses PCIe on one of the engine's TWO copy streams, taken in turn, and the slot's stream waits for it.  Issued
    // on the slots' own 22 streams the input copies moved 31 GB/s in aggregate — one DMA engine's rate —, on one copy stream the
    // same; on two, 48 GB/s = 84 % of the link (122 us per 1 024-e-mail batch instead of 186; three streams: 141 us, four: worse).
    // Nothing on the device has to be waited for first: the slot's previous host batch — the only earlier user of d_image — was
    // retired on the host before the image was packed.  (Small images stay on the slot's stream: the cross-stream event costs a
    // single e-mail 16 us of latency and buys nothing.)
    std::lock_guard<std::mutex> cg(e->copy_mu);
    hipStream_t cs = e->copy_stream[e->copy_turn++ % ZKE_COPY_STREAMS];
    HIPCHK(e, hipMemcpyAsync(w.d_image.p, hp, L.total, hipMemcpyHostToDevice, cs));
    HIPCHK(e, hipEventRecord(w.h2d_done, cs));
    HIPCHK(e, hipStreamWaitEvent(s, w.h2d_done, 0));
  } else {
    HIPCHK(e, hipMemcpyAsync(w.d_image.p, hp, L.total, hipMemcpyHostToDevice, s));
  }
  tm.mark(MK_H2D);
  uint8_t* dp = w.d_image.as<uint8_t>();
  zke_batch dv = *in;
  dv.raw_off = reinterpret_cast<const uint64_t*>(dp + L.raw_off);
  dv.domain_off = reinterpret_cast<const uint64_t*>(dp + L.dom_off);
  dv.key_off = reinterpret_cast<const uint64_t*>(dp + L.key_off);
  dv.raw_blob = dp + L.raw - raw_base;
  dv.domain_blob = dp + L.dom - dom_base;
  dv.key_blob = dp + L.key - key_base;
  dv.key_type = dp + L.key_type;
  dv.ext_null = (refs || in->ext_null) ? dp + L.ext_null : nullptr;
  dv.cap_off = caps ? reinterpret_cast<const uint32_t*>(dp + L.cap_off) : nullptr;
  dv.cap_str_off = caps ? reinterpret_cast<const uint32_t*>(dp + L.cap_str_off) : nullptr;
  dv.cap_blob = caps ? dp + L.cap_blob : nullptr;
  if (int r = run_device_pipeline(e, w, &dv, raw_total, key_total, w.d_results.as<zke_result>(), s, want_em, batch_clock(e), want_clean)) return r;
  HIPCHK(e, hipMemcpyAsync(w.h_results.p, w.d_results.p, (size_t)n * sizeof(zke_result), hipMemcpyDeviceToHost, s));
  tm.mark(MK_D2H);
  HIPCHK(e, hipEventRecord(w.host_done, s));
  w.host_gen++;
  w.host_out = out; w.host_n = n;
  return use.release();
}

// The host entry's staging of EVERY slot, sized for images of `image` bytes and n records.  Pinned memory that comes into being
// while other slots' copies are in flight copies at a fraction of the link's rate for the rest of its life (measured: slots that
// allocated their staging lazily, one by one under traffic, moved 9 GB/s in aggregate; the same buffers allocated together in a
// quiet moment 31 GB/s) — so growth is a stop-the-world event: no submission in progress (`big` exclusive), every pending host
// batch delivered, every stream drained, then all slots at once, with headroom so that it stays rare.
int grow_host_staging(zke_engine* e, size_t image, uint32_t n) {
  std::unique_lock<std::shared_mutex> ex(e->big);
  if (image <= e->host_image_cap.load() && n <= e->host_n_cap.load()) return 0;       // another thread grew it meanwhile
  HIPCHK(e, hipSetDevice(e->device));
  for (Slot* w : e->slots) {
    std::lock_guard<std::mutex> g(w->mu);
    if (int r = retire_host(e, *w)) return r;
    if (w->last_stream && w->last_stream != w->stream) HIPCHK(e, hipEventSynchronize(w->done));
    HIPCHK(e, hipStreamSynchronize(w->stream));
  }
  for (auto& cs : e->copy_stream)
    if (!cs) HIPCHK(e, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  const size_t want_image = std::max(image + image / 2, e->host_image_cap.load());
  const uint32_t want_n = std::max<uint32_t>(n + n / 2, e->host_n_cap.load());
  for (Slot* w : e->slots)
    if (int r = ensure_host_buffers(e, *w, want_image, want_n)) return r;
  e->host_image_cap = want_image;
  e->host_n_cap = want_n;
  return 0;
}
// what one batch needs of the staging (image_layout of its sizes)
size_t host_image_bytes(const zke_batch* in) {
  const uint32_t n = in->n;
  const uint32_t P = in->with_regex ? in->n_header_parts + in->n_body_parts : 0;
  const bool caps = P && in->cap_off;
  const uint32_t n_caps = caps ? in->cap_off[(size_t)n * P] : 0;
  return image_layout(n, in->raw_off[n] - in->raw_off[0], in->domain_off[n] - in->domain_off[0], in->key_off[n] - in->key_off[0],
                      caps ? (size_t)n * P + 1 : 0, caps ? (size_t)n_caps + 1 : 0, caps ? in->cap_str_off[n_caps] : 0).total;
}

int check_host_batch(zke_engine* e, const zke_batch* in, zke_result* out, const char* who) {
  if (!e) return ZKE_E_ARG;
  if (!in || (in->n && (!out || !in->raw_blob || !in->raw_off || !in->domain_blob || !in->domain_off || !in->key_blob ||
                        !in->key_off || !in->key_type)))
    return fail(e, ZKE_E_ARG, who);
  if (in->with_regex && ((in->n_header_parts && !in->header_part_ids) || (in->n_body_parts && !in->body_part_ids)))
    return fail(e, ZKE_E_ARG, "part-id list is null");
  // The offset arrays are in host memory here, so they are checked (three passes over n + 1 words): a length that came out
  // negative would send the staging copy, and then the kernels, outside the blobs.  (In device memory — zke_verify_batch_device —
  // they are the caller's to get right, like the pointers themselves.)
  auto rising = [](const auto* off, size_t cnt) { uint64_t bad = 0; for (size_t i = 0; i < cnt; i++) bad |= (uint64_t)(off[i + 1] < off[i]); return !bad; };
  if (in->n && !(rising(in->raw_off, in->n) && rising(in->domain_off, in->n) && rising(in->key_off, in->n)))
    return fail(e, ZKE_E_ARG, "offset array is not non-decreasing");
  if (in->n && in->raw_off[in->n] - in->raw_off[0] > (1ull << 40)) return fail(e, ZKE_E_ARG, "raw blob beyond 1 TiB");
  if (in->with_regex && in->cap_off) {
    const size_t NP = (size_t)in->n * ((size_t)in->n_header_parts + in->n_body_parts);
    if (NP && (!rising(in->cap_off, NP) || (in->cap_off[NP] && (!in->cap_str_off || !in->cap_blob || !rising(in->cap_str_off, in->cap_off[NP])))))
      return fail(e, ZKE_E_ARG, "capture offset array is not non-decreasing");
  }
  return 0;
}

// ---- the DFA registry (dfa_registry.hip.h has the blob parser and the entry type)
// (caller holds reg_mu exclusively, and the engine is idle or the entry was never handed out)
void drop_dfa(zke_engine* e, uint32_t id) {
  RegisteredDfa* d = e->dfas[id];
  auto range = e->dfa_index.equal_range(d->hash);
  for (auto it = range.first; it != range.second; ++it)
    if (it->second == id) { e->dfa_index.erase(it); break; }
  d->blob.release(); d->dev.release();
  delete d;
  e->dfas[id] = nullptr;
  e->dfa_live--;
}

// (caller holds reg_mu, shared or exclusive)
bool dfa_lookup(zke_engine* e, uint64_t h, const uint8_t* fwd, size_t fl, const uint8_t* bwd, size_t bl, uint32_t* id, bool pin) {
  auto range = e->dfa_index.equal_range(h);
  for (auto it = range.first; it != range.second; ++it) {
    RegisteredDfa* d = e->dfas[it->second];
    if (d && d->fwd_copy.size() == fl && d->bwd_copy.size() == bl && (!fl || !memcmp(d->fwd_copy.data(), fwd, fl)) &&
        (!bl || !memcmp(d->bwd_copy.data(), bwd, bl))) {
      d->last_use.store(e->reg_clock.fetch_add(1) + 1, std::memory_order_relaxed);
      if (pin) d->pins.fetch_add(1);
      *id = it->second;
      return true;
    }
  }
  return false;
}

// The registry is full: drop the least recently used pair among those zke_verify_email_with_regex registered on its own.
// The tables may be in use by batches in flight, so the engine is drained first (exclusive lock + stream syncs).
int dfa_evict_one(zke_engine* e) {
  std::unique_lock<std::shared_mutex> ex(e->big);
  HIPCHK(e, hipSetDevice(e->device));
  for (Slot* w : e->slots) {
    if (w->last_stream && w->last_stream != w->stream) HIPCHK(e, hipEventSynchronize(w->done));
    HIPCHK(e, hipStreamSynchronize(w->stream));
  }
  std::unique_lock<std::shared_mutex> rl(e->reg_mu);
  uint32_t victim = 0xFFFFFFFFu;
  uint64_t oldest = ~0ull;
  for (uint32_t k = 0; k < e->dfas.size(); k++) {
    const RegisteredDfa* d = e->dfas[k];
    if (d && d->transient && !d->pins.load() && d->last_use.load(std::memory_order_relaxed) < oldest) { oldest = d->last_use.load(std::memory_order_relaxed); victim = k; }
  }
  if (victim == 0xFFFFFFFFu) return fail(e, ZKE_E_NOMEM, "DFA registry full (zke_options.max_dfas): zke_dfa_unregister pairs no longer needed (pairs of per-e-mail calls in progress cannot be evicted)");
  drop_dfa(e, victim);
  return 0;
}

// transient: registered by a per-e-mail call on its own — evictable, and PINNED for the caller (dfa_unpin when its batch is done)
int dfa_register_impl(zke_engine* e, const uint8_t* fwd, size_t fwd_len, const uint8_t* bwd, size_t bwd_len, uint32_t* out_id, bool transient) {
  if (!e || !out_id || (fwd_len && !fwd) || (bwd_len && !bwd)) return ZKE_E_ARG;
  const uint64_t h = pair_hash(fwd, fwd_len, bwd, bwd_len);
  {
    // registering the same pair again returns the id it already has (per-e-mail callers re-submit their part list)
    std::shared_lock<std::shared_mutex> rl(e->reg_mu);
    if (dfa_lookup(e, h, fwd, fwd_len, bwd, bwd_len, out_id, transient)) return 0;
  }
  HIPCHK(e, hipSetDevice(e->device));
  RegisteredDfa* rd = new RegisteredDfa();
  auto discard = [&]() { rd->blob.release(); rd->dev.release(); delete rd; };
  rd->fwd_copy.assign(fwd, fwd + fwd_len);
  rd->bwd_copy.assign(bwd, bwd + bwd_len);
  rd->hash = h;
  rd->transient = transient;
  {
    HostDfa hf, hr;
    uint32_t det = parse_dfa_blob(fwd, fwd_len, hf);
    if (!det) { det = parse_dfa_blob(bwd, bwd_len, hr); if (det) det += ZKE_D_DFA_BWD_OFFSET; }
    rd->detail = det;
    rd->valid = det == 0;
    if (rd->valid) {
      auto packed = [](const HostDfa& x) { return (((size_t)x.d.table_len * (x.d.wide ? 4 : 2)) + 15) & ~(size_t)15; };
      const size_t fb = packed(hf), rb = packed(hr);
      rd->lds_bytes = fb + rb;
      rd->idle = dfa_idle_state(hf);
      int r = 0;
      if ((r = rd->blob.ensure(fb + rb + 64)) || (r = rd->dev.ensure(sizeof(RegexDev)))) { discard(); return fail(e, r, "hipMalloc"); }
      std::vector<uint8_t> img(fb + rb + 64, 0);
      auto pack = [&](const HostDfa& x, size_t off) {
        if (x.d.wide) memcpy(img.data() + off, x.table.data(), x.table.size() * 4);
        else { uint16_t* o = reinterpret_cast<uint16_t*>(img.data() + off); for (size_t i = 0; i < x.table.size(); i++) o[i] = (uint16_t)x.table[i]; }
      };
      pack(hf, 0); pack(hr, fb);
      RegexDev rdv{};
      rdv.fwd = hf.d; rdv.rev = hr.d;
      rdv.fwd.table = (uint64_t)rd->blob.as<uint8_t>();
      rdv.rev.table = (uint64_t)(rd->blob.as<uint8_t>() + fb);
      hipError_t he = hipMemcpy(rd->blob.p, img.data(), img.size(), hipMemcpyHostToDevice);
      if (he == hipSuccess) he = hipMemcpy(rd->dev.p, &rdv, sizeof rdv, hipMemcpyHostToDevice);
      if (he != hipSuccess) { discard(); return fail(e, ZKE_E_DEVICE, "dfa upload", he); }
    }
  }
  for (;;) {
    {
      std::unique_lock<std::shared_mutex> rl(e->reg_mu);
      if (dfa_lookup(e, h, fwd, fwd_len, bwd, bwd_len, out_id, transient)) { discard(); return 0; }      // another thread was first
      if (e->dfa_live < e->opt.max_dfas) {
        if (rd->valid && rd->lds_bytes + 1024 <= 150 * 1024)       // the tables fit in LDS: the DFA kernels are launched with that much
          if (int r = raise_dfa_lds_attrs(e, rd->lds_bytes + 1024)) { discard(); return r; }
        uint32_t id = 0;
        while (id < e->dfas.size() && e->dfas[id]) id++;
        if (id == e->dfas.size()) e->dfas.push_back(nullptr);
        rd->last_use.store(e->reg_clock.fetch_add(1) + 1, std::memory_order_relaxed);
        rd->pins.store(transient ? 1u : 0u);
        e->dfas[id] = rd;
        e->dfa_index.emplace(h, id);
        e->dfa_live++;
        *out_id = id;
        return 0;
      }
    }
    if (int r = dfa_evict_one(e)) { discard(); return r; }
  }
}

void dfa_unpin(zke_engine* e, const std::vector<uint32_t>& ids) {
  std::shared_lock<std::shared_mutex> rl(e->reg_mu);
  for (uint32_t id : ids) e->dfas[id]->pins.fetch_sub(1);          // (a pinned entry is neither evicted nor unregistered: it is there)
}

// tickets: slot index in the low 6 bits (an engine has at most 64 slots), the slot's batch count above
inline uint64_t make_ticket(uint32_t slot, uint64_t gen) { return (gen << 6) | slot; }

}  // namespace

extern "C" {

int zke_dfa_register(zke_engine* e, const uint8_t* fwd, size_t fwd_len, const uint8_t* bwd, size_t bwd_len, uint32_t* out_id) {
  return dfa_register_impl(e, fwd, fwd_len, bwd, bwd_len, out_id, false);
}

int zke_dfa_status(zke_engine* e, uint32_t id, uint32_t* detail) {
  if (!e || !detail) return ZKE_E_ARG;
  std::shared_lock<std::shared_mutex> rl(e->reg_mu);
  if (id >= e->dfas.size() || !e->dfas[id]) return fail(e, ZKE_E_DFA, "zke_dfa_status: id is not registered");
  *detail = e->dfas[id]->detail;
  return 0;
}

int zke_dfa_unregister(zke_engine* e, uint32_t id) {
  if (!e) return ZKE_E_ARG;
  std::unique_lock<std::shared_mutex> ex(e->big);         // no submission in progress ...
  HIPCHK(e, hipSetDevice(e->device));
  for (Slot* w : e->slots) {                              // ... and nothing in flight that could still read the tables
    if (w->last_stream && w->last_stream != w->stream) HIPCHK(e, hipEventSynchronize(w->done));
    HIPCHK(e, hipStreamSynchronize(w->stream));
  }
  std::unique_lock<std::shared_mutex> rl(e->reg_mu);
  if (id >= e->dfas.size() || !e->dfas[id]) return fail(e, ZKE_E_DFA, "zke_dfa_unregister: id is not registered");
  if (e->dfas[id]->pins.load()) return fail(e, ZKE_E_DFA, "zke_dfa_unregister: a zke_verify_email_with_regex call in progress uses this pair");
  drop_dfa(e, id);
  return 0;
}

int zke_engine_reserve(zke_engine* e, uint32_t max_n, uint64_t max_raw_total, uint32_t slots, uint32_t max_regex_parts) {
  if (!e || slots == 0 || slots > 64) return ZKE_E_ARG;
  std::unique_lock<std::shared_mutex> ex(e->big);
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
  const bool timing = e->timing.exchange(false);
  for (Slot* w : e->slots) {
    hipLaunchKernelGGL(slot_warm_kernel, dim3(1), dim3(64), 0, w->stream, (uint32_t*)nullptr);
    if (max_n) {
      SlotUse use(e, *w, w->stream);
      int r = use.acquire();
      if (!r) r = run_device_pipeline(e, *w, &wb, 0, 0, reinterpret_cast<zke_result*>(z + 768), w->stream, false, 0);
      if (r) { e->timing = timing; return r; }
      (void)use.release();
    }
  }
  e->timing = timing;
  HIPCHK(e, hipGetLastError());
  for (Slot* w : e->slots) HIPCHK(e, hipStreamSynchronize(w->stream));
  if (e->host_image_cap.load())          // slots created just now get the staging the others have (everything is drained: see grow_host_staging)
    for (Slot* w : e->slots)
      if (int r = ensure_host_buffers(e, *w, e->host_image_cap.load(), e->host_n_cap.load())) return r;
  return 0;
}

int zke_engine_reserve_host(zke_engine* e, uint32_t max_n, uint64_t max_input_bytes) {
  if (!e) return ZKE_E_ARG;
  // offsets, key types and 64-byte alignment on top of the blobs (image_layout)
  const size_t image = (size_t)max_input_bytes + (size_t)(max_n + 1) * 24 + 2 * (size_t)max_n + 16 * 64 + 4 * 64;
  if (image <= e->host_image_cap.load() && max_n <= e->host_n_cap.load()) return 0;
  return grow_host_staging(e, image, max_n);
}

int zke_verify_batch_device(zke_engine* e, const zke_batch* in, uint64_t raw_total, uint64_t domain_total, uint64_t key_total,
                            zke_result* out_dev, void* stream) {
  (void)domain_total;
  if (!e) return ZKE_E_ARG;
  if (!in || (in->n && (!out_dev || !in->raw_blob || !in->raw_off || !in->domain_off || !in->key_off || !in->key_type)))
    return fail(e, ZKE_E_ARG, "zke_verify_batch_device: null pointer");
  if (in->with_regex && ((in->n_header_parts && !in->header_part_ids) || (in->n_body_parts && !in->body_part_ids)))
    return fail(e, ZKE_E_ARG, "zke_verify_batch_device: part-id list is null");
  std::shared_lock<std::shared_mutex> sh(e->big);
  HIPCHK(e, hipSetDevice(e->device));
  // Submission slots are taken round-robin: with S slots, S batches are in flight before a workspace is reused.
  uint32_t slot;
  Slot& w = next_slot(e, slot);
  std::lock_guard<std::mutex> g(w.mu);
  hipStream_t s = stream ? (hipStream_t)stream : w.stream;
  const uint64_t now = batch_clock(e);
  SlotUse use(e, w, s);
  if (int r = use.acquire()) return r;
  const bool graphs = e->opt.replay_graphs && !e->timing.load() && !((e->strict & ZKE_STRICT_EXPIRY_X) && !e->opt.now_unix);
  if (!graphs) {
    // Launched eagerly: three kernels per signature round (front end, hash / modexp stage, Ed25519 + verdict)
    if (int r = run_device_pipeline(e, w, in, raw_total, key_total, out_dev, s, false, now)) return r;
    return use.release();
  }
  // hipGraph replay (opt-in).  A service re-submits batches that live in the same staging buffers: the second time a
  // slot sees a descriptor byte for byte — input pointers and sizes, output pointer, part ids, rounds, key-size hint —
  // its kernel sequence is captured, and replayed from then on.  The key also holds the slot's workspace generation: a
  // graph bakes in the workspace pointers, and a batch that regrew a buffer in between (DevBuf::ensure frees and
  // reallocates) would leave them dangling.  Nothing in the submit path calls hipMalloc / hipFuncSetAttribute once the
  // workspaces are reserved, so the capture contains kernel nodes only.
  const uint32_t nh = in->with_regex ? in->n_header_parts : 0, nb = in->with_regex ? in->n_body_parts : 0;
  std::vector<uint8_t> key(sizeof(zke_batch) + 5 * sizeof(uint64_t) + 4 * (size_t)(nh + nb));
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
    const uint64_t nums[] = {raw_total, key_total, ((uint64_t)kb.with_regex << 32) | e->opt.max_sig_rounds, ((uint64_t)kb.n_header_parts << 32) | kb.n_body_parts,
                             w.generation};
    (void)o;
    uint8_t* tail = p + sizeof(zke_batch);
    memcpy(tail, nums, sizeof nums);
    tail += sizeof nums;
    if (nh) memcpy(tail, in->header_part_ids, 4 * (size_t)nh);
    tail += 4 * (size_t)nh;
    if (nb) memcpy(tail, in->body_part_ids, 4 * (size_t)nb);
  }
  static_assert(8 + 13 * sizeof(void*) <= sizeof(zke_batch), "graph key layout");
  if (w.graph_exec && key == w.graph_key) {
    HIPCHK(e, hipGraphLaunch(w.graph_exec, s));
    return use.release();
  }
  if (w.graph_exec) { (void)hipGraphExecDestroy(w.graph_exec); w.graph_exec = nullptr; }
  if (key != w.graph_key) {               // first sighting: run eagerly (this is also what sizes the workspaces)
    if (int r = run_device_pipeline(e, w, in, raw_total, key_total, out_dev, s, false, now)) return r;
    // the generation may have moved: remember the key as it is now, so that an identical second call captures
    const uint64_t gen = w.generation;
    memcpy(key.data() + sizeof(zke_batch) + 4 * sizeof(uint64_t), &gen, sizeof gen);
    w.graph_key = key;
    return use.release();
  }
  hipGraph_t g2 = nullptr;
  HIPCHK(e, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  const int r = run_device_pipeline(e, w, in, raw_total, key_total, out_dev, s, false, now);
  const hipError_t ce = hipStreamEndCapture(s, &g2);
  hipGraphExec_t ge = nullptr;
  if (!r && ce == hipSuccess && g2 && hipGraphInstantiate(&ge, g2, nullptr, nullptr, 0) == hipSuccess && ge) {
    (void)hipGraphDestroy(g2);
    w.graph_exec = ge;
    HIPCHK(e, hipGraphLaunch(w.graph_exec, s));
    return use.release();
  }
  if (g2) (void)hipGraphDestroy(g2);
  w.graph_key.clear();                    // capture is an optimisation only: this descriptor runs eagerly
  if (r) return r;
  if (int r2 = run_device_pipeline(e, w, in, raw_total, key_total, out_dev, s, false, now)) return r2;
  return use.release();
}

int zke_verify_batch_async(zke_engine* e, const zke_batch* in, zke_result* out, uint64_t* ticket) {
  if (int r = check_host_batch(e, in, out, "zke_verify_batch_async: null pointer")) return r;
  if (!ticket) return fail(e, ZKE_E_ARG, "zke_verify_batch_async: null ticket");
  if (in->n) { const size_t img = host_image_bytes(in); if (img > e->host_image_cap.load() || in->n > e->host_n_cap.load()) if (int r = grow_host_staging(e, img, in->n)) return r; }
  std::shared_lock<std::shared_mutex> sh(e->big);
  HIPCHK(e, hipSetDevice(e->device));
  uint32_t slot;
  Slot& w = next_slot(e, slot);
  std::lock_guard<std::mutex> g(w.mu);
  if (in->n == 0) { *ticket = make_ticket(slot, w.host_retired); return 0; }      // nothing to wait for
  if (int r = submit_host(e, w, in, out, false, false)) return r;
  *ticket = make_ticket(slot, w.host_gen);
  return 0;
}

int zke_batch_wait(zke_engine* e, uint64_t ticket) {
  if (!e) return ZKE_E_ARG;
  std::shared_lock<std::shared_mutex> sh(e->big);
  const uint32_t slot = (uint32_t)(ticket & 63);
  if (slot >= e->slots.size()) return fail(e, ZKE_E_ARG, "zke_batch_wait: no such ticket");
  Slot& w = *e->slots[slot];
  std::lock_guard<std::mutex> g(w.mu);
  if ((ticket >> 6) > w.host_gen) return fail(e, ZKE_E_ARG, "zke_batch_wait: no such ticket");
  if ((ticket >> 6) <= w.host_retired) return 0;        // delivered already (waited for before, or retired by the slot's next batch)
  HIPCHK(e, hipSetDevice(e->device));
  return retire_host(e, w);
}

int zke_verify_emails_with_regex_async(zke_engine* e, const zke_email_ref* emails, uint32_t n, const zke_regex_lists* lists,
                                       zke_result* out, uint64_t* ticket) {
  if (!e) return ZKE_E_ARG;
  if (!ticket || (n && (!emails || !out))) return fail(e, ZKE_E_ARG, "zke_verify_emails_async: null pointer");
  RefTotals t;
  for (uint32_t i = 0; i < n; i++) {
    const zke_email_ref& m = emails[i];
    if ((m.raw_len && !m.raw) || (m.domain_len && !m.from_domain) || (m.key_len && !m.key)) return fail(e, ZKE_E_ARG, "zke_verify_emails_async: null buffer with a length");
    if (m.raw_len > (1ull << 40) || m.domain_len > (1ull << 32) || m.key_len > (1ull << 32)) return fail(e, ZKE_E_ARG, "zke_verify_emails_async: implausible length");
    t.raw += m.raw_len; t.dom += m.domain_len; t.key += m.key_len;
  }
  if (t.raw > (1ull << 40)) return fail(e, ZKE_E_ARG, "raw e-mails beyond 1 TiB");
  zke_batch proto{};
  proto.n = n;
  size_t cap_words = 0, cap_strs = 0, cap_bytes = 0;
  if (lists) {
    proto.with_regex = 1;
    proto.n_header_parts = lists->n_header_parts; proto.header_part_ids = lists->header_part_ids;
    proto.n_body_parts = lists->n_body_parts; proto.body_part_ids = lists->body_part_ids;
    proto.cap_off = lists->cap_off; proto.cap_str_off = lists->cap_str_off; proto.cap_blob = lists->cap_blob;
    if ((proto.n_header_parts && !proto.header_part_ids) || (proto.n_body_parts && !proto.body_part_ids)) return fail(e, ZKE_E_ARG, "part-id list is null");
    const size_t NP = (size_t)n * ((size_t)proto.n_header_parts + proto.n_body_parts);
    if (NP && proto.cap_off) {
      auto rising = [](const uint32_t* off, size_t cnt) { uint32_t bad = 0; for (size_t i = 0; i < cnt; i++) bad |= (uint32_t)(off[i + 1] < off[i]); return !bad; };
      if (!rising(proto.cap_off, NP) || (proto.cap_off[NP] && (!proto.cap_str_off || !proto.cap_blob || !rising(proto.cap_str_off, proto.cap_off[NP]))))
        return fail(e, ZKE_E_ARG, "capture offset array is not non-decreasing");
      if (proto.cap_off[NP] == 0) proto.cap_off = nullptr;            // tables without a single string: the same as no tables
      else { cap_words = NP + 1; cap_strs = (size_t)proto.cap_off[NP] + 1; cap_bytes = proto.cap_str_off[proto.cap_off[NP]]; }
    } else {
      proto.cap_off = nullptr;
    }
  }
  if (n) { const size_t img = image_layout(n, t.raw, t.dom, t.key, cap_words, cap_strs, cap_bytes).total; if (img > e->host_image_cap.load() || n > e->host_n_cap.load()) if (int r = grow_host_staging(e, img, n)) return r; }
  std::shared_lock<std::shared_mutex> sh(e->big);
  HIPCHK(e, hipSetDevice(e->device));
  uint32_t slot;
  Slot& w = next_slot(e, slot);
  std::lock_guard<std::mutex> g(w.mu);
  if (n == 0) { *ticket = make_ticket(slot, w.host_retired); return 0; }
  if (int r = submit_host(e, w, &proto, out, false, false, emails, &t)) return r;
  *ticket = make_ticket(slot, w.host_gen);
  return 0;
}

int zke_verify_emails_async(zke_engine* e, const zke_email_ref* emails, uint32_t n, zke_result* out, uint64_t* ticket) {
  return zke_verify_emails_with_regex_async(e, emails, n, nullptr, out, ticket);
}

int zke_verify_emails_with_regex(zke_engine* e, const zke_email_ref* emails, uint32_t n, const zke_regex_lists* lists, zke_result* out) {
  uint64_t ticket = 0;
  if (int r = zke_verify_emails_with_regex_async(e, emails, n, lists, out, &ticket)) return r;
  return n ? zke_batch_wait(e, ticket) : 0;
}

int zke_verify_emails(zke_engine* e, const zke_email_ref* emails, uint32_t n, zke_result* out) {
  return zke_verify_emails_with_regex(e, emails, n, nullptr, out);
}

int zke_verify_batch(zke_engine* e, const zke_batch* in, zke_result* out, zke_debug_out* dbg) {
  if (int r = check_host_batch(e, in, out, "zke_verify_batch: null pointer")) return r;
  const uint32_t n = in->n;
  if (n == 0) return 0;
  { const size_t img = host_image_bytes(in); if (img > e->host_image_cap.load() || n > e->host_n_cap.load()) if (int r = grow_host_staging(e, img, n)) return r; }
  std::shared_lock<std::shared_mutex> sh(e->big);
  HIPCHK(e, hipSetDevice(e->device));
  uint32_t slot;
  Slot& w = next_slot(e, slot);
  std::lock_guard<std::mutex> g(w.mu);
  const bool want_em = dbg && dbg->em;
  if (int r = submit_host(e, w, in, out, want_em, dbg && dbg->clean_body)) return r;
  if (int r = retire_host(e, w)) return r;
  if (!dbg) return 0;

  // parity intermediates (tests): copy the slot's scratch back and slice it on the host; the slot's lock is still held
  const uint64_t raw_total = in->raw_off[n] - in->raw_off[0];
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
    const uint64_t rel = in->raw_off[i] - in->raw_off[0];
    const uint32_t raw_len = (uint32_t)(in->raw_off[i + 1] - in->raw_off[i]);
    const uint8_t* regA = scr.data() + host_scratch_off(in->raw_off, i);
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
      put(dbg->clean_body, dbg->clean_body_stride, i, clean.data() + (rel + (uint64_t)i * CLEAN_PER_EMAIL), meta2[i].hashed_len);
    else if (dbg->clean_body)
      put(dbg->clean_body, dbg->clean_body_stride, i, nullptr, 0);
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
  // every pair this call registers or finds stays pinned until the batch has run: with the registry at its cap another
  // thread's registration evicts the least recently used transient pair and the id is handed out again
  struct Pins { zke_engine* e; std::vector<uint32_t> ids; ~Pins() { if (!ids.empty()) dfa_unpin(e, ids); } } pinned{e, {}};
  for (int side = 0; side < 2; side++) {
    const zke_regex_part* parts = side ? body_parts : header_parts;
    const uint32_t np = side ? n_body_parts : n_header_parts;
    for (uint32_t k = 0; k < np; k++) {
      const zke_regex_part& p = parts[k];
      if ((p.fwd_len && !p.fwd) || (p.bwd_len && !p.bwd) || (p.n_captures && (!p.captures || !p.capture_lens)))
        return fail(e, ZKE_E_ARG, "zke_verify_email_with_regex: null pointer in a part");
      uint32_t id = 0;
      // the same pair gets the same id: a hash lookup, not a parse; pairs registered here are the evictable ones
      if (int r = dfa_register_impl(e, p.fwd, p.fwd_len, p.bwd, p.bwd_len, &id, true)) return r;
      pinned.ids.push_back(id);
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
