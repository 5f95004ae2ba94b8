// prl_publisher: the last leg of the preprocessor loop as a NATIVE worker thread - a drain's packed micro-batches leave the
// device in one copy and enter their trainers' shared-memory logs while the Python loop is already ingesting, scanning (K5)
// and packing (K6) the next chunk.
//
// Reference: `write_micro_batch_slices` inside the preprocessing loop (pipelinerl/preprocess.py:356-367, 629-648) publishes
// inline; the reader of `training_data` sees the same record sequence here.  Why native and not a Python thread: the work is
// a device wait + ~9 MB of memcpy per chunk, but a Python publisher thread has to take the GIL for every record it frames and
// every ctypes call it returns from, and the main loop is itself Python - measured, the threaded form was SLOWER than inline
// (0.039 vs 0.037 us/token, every main-loop phase 1.5-2 x longer; profiles/r05e_*).  Here the main thread only builds the
// record headers and a piece table; everything that waits or copies happens in this file without the interpreter.
//
//   submit(job)   copies the job's tables and inline bytes (record headers, sentinel batches), returns a ticket; blocks while
//                 two jobs are pending (double buffering: job k uses staging buffer k % 2)
//   worker        hipStreamWaitEvent(copy stream, the drain's "kernels done" event) -> hipMemcpyAsync device block -> page-locked
//                 staging -> for every record, in order: prl_log_appendv(log of its partition, pieces); the NEXT job's copy is
//                 issued before this job's records are gathered (its DMA hides behind the memcpys)
//   completed()   highest ticket that is in the logs (the caller releases the device block of a finished job)
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "prl_common.h"

struct prl_publisher {
  struct Job {
    uint64_t ticket = 0;
    const void* dev_block = nullptr;
    uint64_t block_bytes = 0;
    void* ready_event = nullptr;
    std::vector<prl_pub_record> recs;
    std::vector<prl_pub_piece> pieces;
    std::vector<uint8_t> inline_bytes;
  };
  int device = 0;
  std::mutex m;
  std::condition_variable cv_work, cv_done;
  std::deque<Job> queue;
  uint64_t next_ticket = 1, completed = 0;
  int pending = 0;  // submitted and not finished
  bool stop = false, failed = false;
  std::string error;
  std::thread* th = nullptr;
  hipStream_t stream = nullptr;
  void* staging[2] = {nullptr, nullptr};
  uint64_t staging_bytes[2] = {0, 0};
  std::atomic<uint64_t> busy_ns{0}, copy_ns{0};

  static uint64_t now_ns() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
  }

  bool fail(const char* what, hipError_t e) {
    std::lock_guard<std::mutex> lk(m);
    failed = true;
    error = std::string(what) + ": " + hipGetErrorString(e);
    return false;
  }

  // ---- the two halves of a job.  start_copy issues the device -> host copy of the job's block on the copy stream (asynchronous DMA);
  // finish_copy waits for it; append gathers the records into the logs.  The worker starts the NEXT job's copy before it appends the
  // current one, so a drain's DMA (~0.45 ms for 9 MB) hides behind its predecessor's memcpys.
  bool start_copy(Job& j) {
    if (!j.block_bytes) return true;
    if (!stream) {  // first job with a device block: this thread's device and its copy stream (a job of inline records needs neither)
      if (hipError_t e = hipSetDevice(device); e != hipSuccess) return fail("hipSetDevice", e);
      if (hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking); e != hipSuccess) return fail("hipStreamCreateWithFlags", e);
    }
    const int k = (int)(j.ticket & 1);
    if (staging_bytes[k] < j.block_bytes) {
      release_staging(staging[k], staging_bytes[k]);
      staging[k] = nullptr;
      staging_bytes[k] = 0;
      uint64_t want = 1ull << 20;
      while (want < j.block_bytes) want <<= 1;
      staging[k] = acquire_staging(want);
      if (!staging[k])
        if (hipError_t e = hipHostMalloc(&staging[k], want, hipHostMallocDefault); e != hipSuccess) return fail("hipHostMalloc(staging)", e);
      staging_bytes[k] = want;
    }
    if (j.ready_event)
      if (hipError_t e = hipStreamWaitEvent(stream, static_cast<hipEvent_t>(j.ready_event), 0); e != hipSuccess) return fail("hipStreamWaitEvent", e);
    if (hipError_t e = hipMemcpyAsync(staging[k], j.dev_block, j.block_bytes, hipMemcpyDeviceToHost, stream); e != hipSuccess) return fail("hipMemcpyAsync(D2H)", e);
    return true;
  }
  bool finish_copy(Job& j) {
    if (!j.block_bytes) return true;
    const uint64_t t0 = now_ns();
    if (hipError_t e = hipStreamSynchronize(stream); e != hipSuccess) return fail("hipStreamSynchronize", e);
    copy_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
    return true;
  }
  bool append(Job& j) {
    const uint8_t* host = static_cast<const uint8_t*>(staging[j.ticket & 1]);
    std::vector<prl_log_iov> iov;
    for (const prl_pub_record& r : j.recs) {
      iov.clear();
      for (uint32_t p = r.first_piece; p < r.first_piece + r.n_pieces; ++p) {
        const prl_pub_piece& pc = j.pieces[p];
        const uint8_t* src = pc.kind == PRL_PUB_FROM_BLOCK ? host + pc.src
                             : pc.kind == PRL_PUB_INLINE   ? j.inline_bytes.data() + pc.src
                                                           : reinterpret_cast<const uint8_t*>(static_cast<uintptr_t>(pc.src));
        iov.push_back(prl_log_iov{src, pc.offset, pc.nbytes});
      }
      const int rc = prl_log_appendv(static_cast<prl_log*>(r.log), iov.data(), (int32_t)iov.size(), r.nbytes);
      if (rc != PRL_OK) {
        std::lock_guard<std::mutex> lk(m);
        failed = true;
        const char* msg = prl_last_error();
        error = std::string("prl_log_appendv: ") + (msg ? msg : "?");
        return false;
      }
    }
    return true;
  }

  // page-locked staging buffers outlive a publisher: allocating one costs milliseconds (the pages are pinned and mapped into the
  // device's address space), a preprocessor that is restarted - or a benchmark that builds a loop per case - should pay that once
  static std::mutex& pool_mutex() {
    static std::mutex pm;
    return pm;
  }
  static std::vector<std::pair<void*, uint64_t>>& pool() {
    static std::vector<std::pair<void*, uint64_t>> p;
    return p;
  }
  static void* acquire_staging(uint64_t bytes) {
    std::lock_guard<std::mutex> lk(pool_mutex());
    auto& p = pool();
    for (size_t i = 0; i < p.size(); ++i)
      if (p[i].second == bytes) {
        void* got = p[i].first;
        p.erase(p.begin() + (long)i);
        return got;
      }
    return nullptr;
  }
  static void release_staging(void* ptr, uint64_t bytes) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(pool_mutex());
    if (pool().size() < 4) {
      pool().emplace_back(ptr, bytes);
    } else {
      (void)hipHostFree(ptr);
    }
  }

  bool ok_now() {
    std::lock_guard<std::mutex> lk(m);
    return !failed;
  }
  void complete(const Job& j) {
    {
      std::lock_guard<std::mutex> lk(m);
      completed = j.ticket;
      --pending;
    }
    cv_done.notify_all();
  }
  // blocking: false when the publisher is stopping and nothing is left
  bool pop(Job& j, bool wait) {
    std::unique_lock<std::mutex> lk(m);
    if (wait) cv_work.wait(lk, [&] { return stop || !queue.empty(); });
    if (queue.empty()) return false;
    j = std::move(queue.front());
    queue.pop_front();
    return true;
  }

  void run() {
    Job cur, next;
    bool have = pop(cur, true);
    bool cur_ok = have && ok_now() && start_copy(cur);
    while (have) {
      const uint64_t t0 = now_ns();
      cur_ok = cur_ok && finish_copy(cur);
      const bool have_next = pop(next, false);  // already queued: its DMA runs while this job's records are gathered
      bool next_ok = have_next && ok_now() && start_copy(next);
      if (cur_ok) append(cur);  // after a failure the remaining jobs are dropped: the error is what the caller sees
      busy_ns.fetch_add(now_ns() - t0, std::memory_order_relaxed);
      complete(cur);
      if (have_next) {
        cur = std::move(next);
        cur_ok = next_ok;
      } else {
        have = pop(cur, true);
        cur_ok = have && ok_now() && start_copy(cur);
      }
    }
    if (stream) (void)hipStreamDestroy(stream);
    for (int k = 0; k < 2; ++k) {
      release_staging(staging[k], staging_bytes[k]);
      staging[k] = nullptr;
    }
  }
};

extern "C" int prl_publisher_create(int32_t device, prl_publisher** out) {
  PRL_CHECK_ARG(out != nullptr && device >= 0, "bad argument");
  prl_publisher* p = new (std::nothrow) prl_publisher();
  if (!p) return prl::set_error(PRL_ENOMEM, "out of memory");
  p->device = device;
  try {  // std::thread reports a failed start by throwing: nothing may unwind through the C boundary
    p->th = new std::thread([p] { p->run(); });
  } catch (...) {
    p->th = nullptr;
  }
  if (!p->th) {
    delete p;
    return prl::set_error(PRL_ENOMEM, "cannot start the publisher thread");
  }
  *out = p;
  return PRL_OK;
}

extern "C" int prl_publisher_submit(prl_publisher* p, const void* dev_block, uint64_t block_bytes, void* ready_event,
                                    const prl_pub_record* recs, int32_t n_recs, const prl_pub_piece* pieces, int32_t n_pieces,
                                    const void* inline_bytes, uint64_t inline_nbytes, uint64_t* ticket) {
  PRL_CHECK_ARG(p && ticket && n_recs >= 0 && n_pieces >= 0 && (recs || n_recs == 0) && (pieces || n_pieces == 0), "bad argument");
  PRL_CHECK_ARG(block_bytes == 0 || dev_block != nullptr, "a block size without a block");
  PRL_CHECK_ARG(inline_nbytes == 0 || inline_bytes != nullptr, "inline size without bytes");
  for (int32_t r = 0; r < n_recs; ++r) {
    PRL_CHECK_ARG(recs[r].log != nullptr && (uint64_t)recs[r].first_piece + recs[r].n_pieces <= (uint64_t)n_pieces, "record %d: bad piece range", r);
    for (uint32_t q = recs[r].first_piece; q < recs[r].first_piece + recs[r].n_pieces; ++q) {
      const prl_pub_piece& pc = pieces[q];
      PRL_CHECK_ARG(pc.kind == PRL_PUB_FROM_BLOCK || pc.kind == PRL_PUB_INLINE || pc.kind == PRL_PUB_FROM_HOST, "piece %u: unknown kind %u", q, pc.kind);
      if (pc.kind == PRL_PUB_FROM_HOST) {
        PRL_CHECK_ARG(pc.src != 0 || pc.nbytes == 0, "piece %u: null host address", q);
        continue;
      }
      const uint64_t limit = pc.kind == PRL_PUB_FROM_BLOCK ? block_bytes : inline_nbytes;
      PRL_CHECK_ARG(pc.src <= limit && pc.nbytes <= limit - pc.src, "piece %u reads outside its source (%llu + %llu > %llu)", q,
                    (unsigned long long)pc.src, (unsigned long long)pc.nbytes, (unsigned long long)limit);
    }
  }
  prl_publisher::Job j;
  j.dev_block = dev_block;
  j.block_bytes = block_bytes;
  j.ready_event = ready_event;
  j.recs.assign(recs, recs + n_recs);
  j.pieces.assign(pieces, pieces + n_pieces);
  if (inline_nbytes) j.inline_bytes.assign(static_cast<const uint8_t*>(inline_bytes), static_cast<const uint8_t*>(inline_bytes) + inline_nbytes);
  {
    std::unique_lock<std::mutex> lk(p->m);
    p->cv_done.wait(lk, [&] { return p->pending < 2 || p->failed; });
    if (p->failed) return prl::set_error(PRL_EFAULT, "publisher failed: %s", p->error.c_str());
    j.ticket = p->next_ticket++;
    *ticket = j.ticket;
    ++p->pending;
    p->queue.push_back(std::move(j));
  }
  p->cv_work.notify_one();
  return PRL_OK;
}

extern "C" int prl_publisher_completed(prl_publisher* p, uint64_t* ticket) {
  PRL_CHECK_ARG(p && ticket, "bad argument");
  std::lock_guard<std::mutex> lk(p->m);
  *ticket = p->completed;
  if (p->failed) return prl::set_error(PRL_EFAULT, "publisher failed: %s", p->error.c_str());
  return PRL_OK;
}

extern "C" int prl_publisher_wait(prl_publisher* p, uint64_t ticket, int64_t timeout_ms) {
  PRL_CHECK_ARG(p != nullptr, "bad argument");
  std::unique_lock<std::mutex> lk(p->m);
  auto done = [&] { return p->completed >= ticket || p->failed; };
  if (timeout_ms < 0) {
    p->cv_done.wait(lk, done);
  } else if (!p->cv_done.wait_for(lk, std::chrono::milliseconds(timeout_ms), done)) {
    return PRL_ETIMEDOUT;
  }
  if (p->failed) return prl::set_error(PRL_EFAULT, "publisher failed: %s", p->error.c_str());
  return PRL_OK;
}

extern "C" int prl_publisher_stats(prl_publisher* p, uint64_t* busy_ns, uint64_t* copy_ns) {
  PRL_CHECK_ARG(p && busy_ns && copy_ns, "bad argument");
  *busy_ns = p->busy_ns.load(std::memory_order_relaxed);
  *copy_ns = p->copy_ns.load(std::memory_order_relaxed);
  return PRL_OK;
}

extern "C" int prl_publisher_destroy(prl_publisher* p) {
  if (!p) return PRL_OK;
  {
    std::lock_guard<std::mutex> lk(p->m);
    p->stop = true;
  }
  p->cv_work.notify_all();
  if (p->th) {
    p->th->join();
    delete p->th;
  }
  delete p;
  return PRL_OK;
}
