// host_fuzz.cpp — the host-side parsers of the engine under AddressSanitizer / UndefinedBehaviorSanitizer, on the CPU, no GPU:
// the whole C-ABI translation unit compiled host-only (hipcc --cuda-host-only -fsanitize=address,undefined) with this main().
//   * parse_dfa_blob: the regex-automata wire format (untrusted caller bytes) — the two golden blobs cut at every length,
//     with every byte of their headers flipped, and random word substitutions;
//   * zke_wire_decode: borsh / bincode records cut at every length and with random byte flips;
//   * CopyPool: several callers at once, odd sizes and alignments, results compared with memcpy (also under -fsanitize=thread);
//   * zke_abi_encode into buffers of exactly the size it asks for, and one byte less;
//   * zke_shard_bounds, image_layout, pair_hash on edge sizes.
// tests/test_host_sanitizers.py builds and runs it.  (The same source with -fsanitize=thread instead: clean as well, run by hand —
// the second 40 s build is not worth a place in the suite.)
#include "../../zkemail.rs_amd/csrc/engine.hip"

#include <random>
#include <thread>

static std::vector<uint8_t> slurp(const char* path) {
  std::vector<uint8_t> v;
  FILE* f = fopen(path, "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
  uint8_t buf[4096]; size_t n;
  while ((n = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + n);
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: host_fuzz fwd.dfa rev.dfa record.bin [record2.bin ...]\n"); return 2; }
  std::mt19937_64 rng(12345);
  size_t cases = 0;
  // ---- dense DFA blobs
  for (int k = 1; k <= 2; k++) {
    const std::vector<uint8_t> good = slurp(argv[k]);
    { HostDfa h; if (parse_dfa_blob(good.data(), good.size(), h) != 0) { fprintf(stderr, "golden blob %d does not parse\n", k); return 1; } }
    for (size_t cut = 0; cut < good.size(); cut += (cut < 600 ? 1 : 37)) {          // every prefix is an exact-size heap buffer: a read past it trips ASan
      std::vector<uint8_t> b(good.begin(), good.begin() + cut);
      HostDfa h; if (parse_dfa_blob(b.data(), b.size(), h) == 0) { fprintf(stderr, "a proper prefix (%zu) parsed\n", cut); return 1; }
      cases++;
    }
    for (size_t at = 0; at < std::min<size_t>(good.size(), 400); at++)
      for (uint8_t x : {(uint8_t)0x01, (uint8_t)0x80, (uint8_t)0xFF}) {
        std::vector<uint8_t> b = good; b[at] ^= x;
        HostDfa h; (void)parse_dfa_blob(b.data(), b.size(), h); cases++;
      }
    for (int it = 0; it < 100000; it++) {                                            // random words anywhere (ids, counts, bounds)
      std::vector<uint8_t> b = good;
      const int m = 1 + (int)(rng() % 3);
      for (int j = 0; j < m; j++) {
        const size_t at = (rng() % (b.size() / 4)) * 4;
        const uint32_t v = (rng() & 1) ? (uint32_t)rng() : (uint32_t)(rng() % 4096);
        memcpy(b.data() + at, &v, 4);
      }
      HostDfa h;
      if (parse_dfa_blob(b.data(), b.size(), h) == 0) {
        // whatever still parses must be walkable without leaving the table: the id checks are what guarantees it
        const DfaDev& d = h.d;
        uint32_t sid = d.starts[2];
        for (int step = 0; step < 300; step++) { if (sid + d.classes[step & 255] >= h.table.size()) { fprintf(stderr, "walk leaves the table\n"); return 1; } sid = h.table[sid + d.classes[step & 255]]; }
        (void)dfa_idle_state(h);
      }
      cases++;
    }
  }
  // ---- borsh / bincode records (argv[3..]: alternately borsh EmailWithRegex, bincode EmailWithRegex)
  for (int k = 3; k < argc; k++) {
    const std::vector<uint8_t> good = slurp(argv[k]);
    const uint32_t fmt = (k - 3) & 1;
    zke_wire_doc* d = nullptr; size_t used = 0;
    if (zke_wire_decode(fmt, good.data(), good.size(), 1, &d, &used) != 0 || used != good.size()) { fprintf(stderr, "record %d does not decode\n", k); return 1; }
    zke_wire_free(d);
    for (size_t cut = 0; cut < good.size(); cut++) {
      std::vector<uint8_t> b(good.begin(), good.begin() + cut);
      if (zke_wire_decode(fmt, b.data(), b.size(), 1, &d, &used) == 0) { fprintf(stderr, "a truncated record (%zu) decoded\n", cut); return 1; }
      cases++;
    }
    for (int it = 0; it < 20000; it++) {
      std::vector<uint8_t> b = good;
      const int m = 1 + (int)(rng() % 4);
      for (int j = 0; j < m; j++) b[rng() % b.size()] = (uint8_t)rng();
      if (zke_wire_decode(fmt, b.data(), b.size(), 1, &d, &used) == 0) {
        zke_wire_email v; (void)zke_wire_view(d, &v);
        size_t sum = v.raw_len + v.domain_len + v.key_len;                           // every span lies inside the buffer
        for (uint32_t p = 0; p < v.n_header_parts; p++) { sum += v.header_parts[p].fwd_len; for (uint32_t c = 0; c < v.header_parts[p].n_captures; c++) sum += v.header_parts[p].capture_lens[c]; }
        for (uint32_t p = 0; p < v.n_body_parts; p++) sum += v.body_parts[p].bwd_len;
        if (sum > b.size()) { fprintf(stderr, "spans exceed the buffer\n"); return 1; }
        zke_wire_free(d);
      }
      cases++;
    }
  }
  // ---- CopyPool: four callers, odd sizes / alignments
  {
    CopyPool pool(3);
    std::vector<std::thread> ths;
    std::atomic<int> bad{0};
    for (int t = 0; t < 4; t++) ths.emplace_back([&, t] {
      std::mt19937_64 r(99 + t);
      for (int it = 0; it < 60; it++) {
        const size_t n = (size_t)(r() % (3u << 20)) + 1, so = r() % 61, dof = (r() % 3) * 64;
        std::vector<uint8_t> src(n + 64), dst(n + 256, 0xEE), ref(n + 256, 0xEE);
        for (size_t i = 0; i < src.size(); i += 97) src[i] = (uint8_t)r();
        CopyPool::Piece pc[2] = {{dst.data() + dof, src.data() + so, n - n / 3}, {dst.data() + dof + (n - n / 3), src.data() + so + (n - n / 3), n / 3}};
        pool.copy(pc, 2);
        memcpy(ref.data() + dof, src.data() + so, n);
        if (dst != ref) bad++;
      }
    });
    for (auto& t : ths) t.join();
    if (bad) { fprintf(stderr, "CopyPool: %d copies differ from memcpy\n", bad.load()); return 1; }
    cases += 240;
    // gather: thousands of small pieces with consecutive destinations (zke_verify_emails), from two callers at once
    std::vector<std::thread> th2;
    for (int t = 0; t < 2; t++) th2.emplace_back([&, t] {
      std::mt19937_64 r(7 + t);
      for (int it = 0; it < 12; it++) {
        const size_t np = (size_t)(r() % 3000) + 1;
        std::vector<std::vector<uint8_t>> srcs(np);
        std::vector<CopyPool::Piece> pc(np);
        size_t total = 0;
        for (auto& v : srcs) { v.resize((size_t)(r() % 5 == 0 ? r() % 20000 : r() % 600)); for (size_t i = 0; i < v.size(); i += 31) v[i] = (uint8_t)r(); total += v.size(); }
        std::vector<uint8_t> dst(total + 64, 0xEE), ref(total + 64, 0xEE);
        size_t o = r() % 64 == 0 ? 0 : r() % 48;
        for (size_t i = 0; i < np; i++) { pc[i] = CopyPool::Piece{dst.data() + o, srcs[i].empty() ? nullptr : srcs[i].data(), srcs[i].size()}; if (!srcs[i].empty()) memcpy(ref.data() + o, srcs[i].data(), srcs[i].size()); o += srcs[i].size(); if (o > total + 16) break; }
        if (o > total + 48) continue;
        dst.resize(std::max(dst.size(), o)); ref.resize(dst.size(), 0xEE);
        pool.gather(pc.data(), np);
        if (dst != ref) bad++;
      }
    });
    for (auto& t : th2) t.join();
    if (bad) { fprintf(stderr, "CopyPool::gather: %d results differ from memcpy\n", bad.load()); return 1; }
    cases += 24;
  }
  // ---- zke_abi_encode: string tables of random sizes into heap buffers of exactly the size asked for, one byte less, and none
  for (int it = 0; it < 3000; it++) {
    uint8_t h1[32], h2[32];
    for (int i = 0; i < 32; i++) { h1[i] = (uint8_t)rng(); h2[i] = (uint8_t)rng(); }
    auto table = [&](std::vector<std::vector<uint8_t>>& store, std::vector<const uint8_t*>& ptr, std::vector<size_t>& len) {
      const int n = (int)(rng() % 5);
      for (int i = 0; i < n; i++) { store.emplace_back((size_t)(rng() % 3 == 0 ? rng() % 100 : rng() % 34)); for (auto& c : store.back()) c = (uint8_t)rng(); }
      for (auto& v : store) { ptr.push_back(v.empty() ? nullptr : v.data()); len.push_back(v.size()); }
    };
    std::vector<std::vector<uint8_t>> s1, s2; std::vector<const uint8_t*> p1, p2; std::vector<size_t> l1, l2;
    table(s1, p1, l1); table(s2, p2, l2);
    const uint32_t wm = (uint32_t)(rng() & 1);
    size_t need = 0, got = 0;
    if (zke_abi_encode(h1, h2, p1.data(), l1.data(), (uint32_t)p1.size(), wm, p2.data(), l2.data(), (uint32_t)p2.size(), nullptr, 0, &need) != 0 || need % 32) { fprintf(stderr, "abi size query\n"); return 1; }
    std::vector<uint8_t> exact(need), tight(need - 1);
    if (zke_abi_encode(h1, h2, p1.data(), l1.data(), (uint32_t)p1.size(), wm, p2.data(), l2.data(), (uint32_t)p2.size(), exact.data(), exact.size(), &got) != 0 || got != need) { fprintf(stderr, "abi exact\n"); return 1; }
    if (zke_abi_encode(h1, h2, p1.data(), l1.data(), (uint32_t)p1.size(), wm, p2.data(), l2.data(), (uint32_t)p2.size(), tight.data(), tight.size(), &got) != ZKE_E_NOMEM) { fprintf(stderr, "abi tight\n"); return 1; }
    cases++;
  }
  // ---- small pure functions on edge sizes
  {
    uint32_t bounds[9];
    uint64_t off1[1] = {7};
    if (zke_shard_bounds(off1, 0, 8, bounds) != 0 || bounds[8] != 0) return 1;
    std::vector<uint64_t> off(1001); off[0] = 1ull << 62;
    for (int i = 1; i <= 1000; i++) off[i] = off[i - 1] + (rng() % 3 == 0 ? 0 : rng() % 100000);
    for (uint32_t w : {1u, 2u, 3u, 8u}) { if (zke_shard_bounds(off.data(), 1000, w, bounds) != 0 || bounds[w] != 1000) return 1; for (uint32_t r = 0; r < w; r++) if (bounds[r] > bounds[r + 1]) return 1; }
    (void)image_layout(0, 0, 0, 0, 0, 0, 0);
    (void)pair_hash(nullptr, 0, nullptr, 0);
    uint8_t x[17] = {1, 2, 3};
    if (pair_hash(x, 17, x, 3) == pair_hash(x, 16, x, 3)) return 1;
  }
  printf("host_fuzz ok: %zu cases\n", cases);
  return 0;
}
