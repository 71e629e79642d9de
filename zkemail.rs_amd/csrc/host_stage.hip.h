// host_stage.hip.h — the host side of the host-memory entry point (zke_verify_batch[_async]): pinned staging memory and the
// worker threads that fill it.
//
// The drop-in caller holds `&[Email]` in ordinary (pageable) RAM (core/src/circuits.rs:9; built at
// helpers/src/generator.rs:40-45).  A DMA engine reads pinned memory only, so every byte crosses the host's memory once
// before it crosses PCIe: pageable -> the slot's pinned image (here, by `host_threads` threads: one thread copies ~10 GB/s,
// a Gen5 x16 link moves ~55), then ONE hipMemcpyAsync of the whole image on the slot's stream.  Included by engine.hip.
#pragma once

#include <condition_variable>
#include <functional>
#include <thread>
#include <immintrin.h>

namespace {

struct PinnedBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t need) {
    if (need <= cap) return 0;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    const size_t want = std::max(need + need / 4, (size_t)4096);
    if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) { p = nullptr; return ZKE_E_NOMEM; }
    cap = want;
    return 0;
  }
  void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pageable -> pinned, streaming: non-temporal stores.  An ordinary memcpy of a piece this size reads the destination lines
// before it overwrites them (read-for-ownership) and leaves the image in the cache hierarchy, where nobody will read it — the
// DMA engine reads memory.  Streaming stores move two bytes per byte copied instead of three.  dst is 32-byte aligned (pieces
// start on 64-byte boundaries of a 4 KiB-aligned pinned buffer); the head and tail go through memcpy.
__attribute__((target("avx2"))) inline void stream_copy_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
  size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
  if (head > n) head = n;
  if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32)),
                  c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
    _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b);
    _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d);
  }
  _mm_sfence();
  if (i < n) memcpy(dst + i, src + i, n - i);
}
inline void stage_copy(void* dst, const void* src, size_t n, size_t stream_from = 64u << 10) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && n >= stream_from) stream_copy_avx2((uint8_t*)dst, (const uint8_t*)src, n);
  else memcpy(dst, src, n);
}

// A handful of threads that do nothing but memcpy.  Several callers may use the pool at once (each call waits for its own
// pieces only); with one thread configured, or for small copies, the caller's thread does the work itself.
#ifndef ZKE_GATHER_STREAM_FROM
#define ZKE_GATHER_STREAM_FROM 1024
#endif
#ifndef ZKE_GATHER_VIA_BUFFER
#define ZKE_GATHER_VIA_BUFFER 1
#endif
#ifndef ZKE_GATHER_CHUNK
#define ZKE_GATHER_CHUNK (256u << 10)
#endif
class CopyPool {
 public:
  explicit CopyPool(unsigned workers) {
    for (unsigned i = 0; i < workers; i++) threads_.emplace_back([this] { run(); });
  }
  ~CopyPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : threads_) t.join();
  }
  // the pieces of one packing job; returns when all of them are done
  struct Piece { void* dst; const void* src; size_t n; };
  void copy(const Piece* pieces, size_t count) {
    constexpr size_t CHUNK = 256 << 10;           // below this a hand-over costs more than the copy
    std::vector<Piece> work;
    for (size_t i = 0; i < count; i++) {
      const Piece& p = pieces[i];
      if (!p.n) continue;
      if (threads_.empty() || p.n < 2 * CHUNK) { work.push_back(p); continue; }
      // (one part per thread: 256 KB parts of a blob, as the gather has them, lose 5-10 % here — measured, tools/ab_scattered.sh)
      const size_t parts = std::min<size_t>(threads_.size() + 1, (p.n + CHUNK - 1) / CHUNK);
      const size_t step = ((p.n + parts - 1) / parts + 63) & ~(size_t)63;
      for (size_t o = 0; o < p.n; o += step)
        work.push_back(Piece{(uint8_t*)p.dst + o, (const uint8_t*)p.src + o, std::min(step, p.n - o)});
    }
    if (work.empty()) return;
    size_t big = 0;
    for (const Piece& p : work) big += p.n >= CHUNK;
    if (threads_.empty() || big < 2) {            // nothing worth sharing
      for (const Piece& p : work) stage_copy(p.dst, p.src, p.n);
      return;
    }
    std::vector<Task> tasks;
    tasks.reserve(work.size());
    for (const Piece& p : work) tasks.push_back(Task{&p, 1, 64u << 10, nullptr});
    run_job(tasks);
  }
  // Many small pieces whose destinations follow each other (the e-mails of a batch handed over one by one, each in its own
  // buffer: `&[Email]` as the reference holds it): consecutive pieces are grouped into tasks of ~256 KB, so a thread is handed a
  // run of e-mails, not one; pieces of a kilobyte and more go through streaming stores.
  void gather(const Piece* pieces, size_t count) {
    constexpr size_t CHUNK = 256 << 10;
    size_t total = 0;
    for (size_t i = 0; i < count; i++) total += pieces[i].n;
    if (threads_.empty() || total < 2 * CHUNK) {
      for (size_t i = 0; i < count; i++) if (pieces[i].n) stage_copy(pieces[i].dst, pieces[i].src, pieces[i].n, ZKE_GATHER_STREAM_FROM);
      return;
    }
    // A task is a run of pieces whose destinations follow each other without a gap, cut at ~256 KB: the thread that takes it
    // collects the run in a buffer of its own (the sources are scattered, the buffer stays in its cache) and writes it out as
    // ONE streaming copy — a streaming copy per 5 KB piece pays a fence and two partial lines (head, tail) per e-mail, and
    // mixes ordinary and write-combining stores on the lines where two e-mails meet.
    std::vector<Task> tasks;
    size_t first = 0, bytes = 0;
    for (size_t i = 0; i < count; i++) {
      bytes += pieces[i].n;
      const bool gap = i + 1 < count && (const uint8_t*)pieces[i + 1].dst != (const uint8_t*)pieces[i].dst + pieces[i].n;
      if (bytes >= ZKE_GATHER_CHUNK || gap || i + 1 == count) {
        tasks.push_back(Task{pieces + first, i + 1 - first, ZKE_GATHER_STREAM_FROM, nullptr, ZKE_GATHER_VIA_BUFFER != 0});
        first = i + 1; bytes = 0;
      }
    }
    run_job(tasks);
  }

 private:
  struct Job { std::mutex mu; std::condition_variable cv; size_t left = 0; };
  struct Task { const Piece* p; size_t cnt; size_t stream_from; Job* job; bool run = false; };   // run: the pieces' destinations are contiguous
  // hand the tasks to the pool, work along, return when all of them are done (`tasks` and what they point to outlive the call)
  void run_job(std::vector<Task>& tasks) {
    if (tasks.empty()) return;
    Job job;
    job.left = tasks.size();
    {
      std::lock_guard<std::mutex> g(mu_);
      for (Task& t : tasks) { t.job = &job; queue_.push_back(t); }
      pending_.fetch_add((int)tasks.size(), std::memory_order_release);
    }
    cv_.notify_all();
    // the caller works too: it takes tasks until the queue is empty, then waits for the stragglers
    for (;;) {
      Task t;
      {
        std::lock_guard<std::mutex> g(mu_);
        if (queue_.empty()) break;
        t = queue_.back(); queue_.pop_back(); pending_.fetch_sub(1, std::memory_order_relaxed);
      }
      do_task(t);
    }
    std::unique_lock<std::mutex> lk(job.mu);
    job.cv.wait(lk, [&] { return job.left == 0; });
  }
  void do_task(const Task& t) {
    if (t.run && t.cnt > 1) {
      static thread_local std::vector<uint8_t> buf;
      size_t total = 0;
      for (size_t i = 0; i < t.cnt; i++) total += t.p[i].n;
      if (buf.size() < total) buf.resize(total + (64u << 10));
      size_t o = 0;
      for (size_t i = 0; i < t.cnt; i++) if (t.p[i].n) { memcpy(buf.data() + o, t.p[i].src, t.p[i].n); o += t.p[i].n; }
      if (total) stage_copy(t.p[0].dst, buf.data(), total, t.stream_from);
    } else
    for (size_t i = 0; i < t.cnt; i++) if (t.p[i].n) stage_copy(t.p[i].dst, t.p[i].src, t.p[i].n, t.stream_from);
    std::lock_guard<std::mutex> g(t.job->mu);
    if (--t.job->left == 0) t.job->cv.notify_all();
  }
  void run() {
    for (;;) {
      Task t;
      bool have = false;
      // A streaming caller hands over the next batch's pieces a few microseconds after the last ones were done: look for them
      // for a moment before going to sleep (waking a sleeping thread costs more than a piece takes to copy).
      for (int spin = 0; spin < 4000 && !have; spin++) {
        if (pending_.load(std::memory_order_acquire)) {
          std::lock_guard<std::mutex> g(mu_);
          if (!queue_.empty()) { t = queue_.back(); queue_.pop_back(); pending_.fetch_sub(1, std::memory_order_relaxed); have = true; }
        } else {
          _mm_pause();
        }
      }
      if (!have) {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
        if (stop_ && queue_.empty()) return;
        t = queue_.back(); queue_.pop_back(); pending_.fetch_sub(1, std::memory_order_relaxed);
      }
      do_task(t);
    }
  }
  std::atomic<int> pending_{0};
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::vector<Task> queue_;
  bool stop_ = false;
};

// Layout of one batch's packed input image: the same bytes in the slot's pinned buffer and in HBM.  Offsets are into
// the image; blobs start 64-byte aligned and are followed by 64 bytes of slack (the kernels read 16 bytes at a time).
struct ImageLayout {
  size_t raw_off, dom_off, key_off, key_type, ext_null, cap_off, cap_str_off, raw, dom, key, cap_blob, total;
};
inline size_t align_up(size_t x, size_t a) { return (x + a - 1) & ~(a - 1); }
inline ImageLayout image_layout(uint32_t n, size_t raw_total, size_t dom_total, size_t key_total, size_t n_cap_off, size_t n_cap_str,
                                size_t cap_bytes) {
  ImageLayout L{};
  size_t o = 0;
  auto put = [&](size_t bytes, size_t slack) { const size_t at = o; o = align_up(o + bytes + slack, 64); return at; };
  L.raw_off = put((size_t)(n + 1) * 8, 0);
  L.dom_off = put((size_t)(n + 1) * 8, 0);
  L.key_off = put((size_t)(n + 1) * 8, 0);
  L.key_type = put(n, 0);
  L.ext_null = put(n, 0);
  L.cap_off = put(n_cap_off * 4, 0);
  L.cap_str_off = put(n_cap_str * 4, 0);
  L.raw = put(raw_total, 64);
  L.dom = put(dom_total, 64);
  L.key = put(key_total, 64);
  L.cap_blob = put(cap_bytes, 64);
  L.total = o;
  return L;
}

}  // namespace
