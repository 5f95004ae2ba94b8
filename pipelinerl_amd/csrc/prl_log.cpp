// prl_log: append-only record LOG in POSIX shared memory with per-reader cursors.
//
// The reference's stream backends are logs, not queues: a Redis stream read with XREAD from id 0
// (pipelinerl/streams.py:120-192) and a JSONL file tailed from offset 0 (:281-346).  Every reader
// sees every record from the first one, any number of readers may follow a topic (TrainerState runs in
// the actor, the preprocessor, the rollout workers and the launcher at once, pipelinerl/state.py:36-48),
// a writer never waits for a reader, and a writer that is closed and opened again appends to the same
// stream (finetune_loop.py:244 opens one per weight update).  prl_ring (a destructive MPMC queue) has
// none of these properties; this file provides them:
//
//   /<name>        control block: segment size, number of segments, first retained segment, writer
//                  mutex, commit futex word, reader cursor table
//   /<name>.<k>    segment k: header + records back to back, each [u64 nbytes][payload][pad to 8]
//
// Writers serialise on a futex mutex in the control block (its owner's thread id is recorded, a lock whose
// owner died is taken over), append into the last segment and publish with a release store of the
// segment's `committed` offset, then bump the commit futex.  A record that does not fit creates the next
// segment (sized for the record if it is larger than the default), THEN seals the old one, THEN publishes
// the new segment count: a reader that finds `sealed` can always open the successor, and a writer that
// finds the last counted segment sealed (its predecessor died in between) adopts the successor.  Readers keep (segment, offset) privately,
// start at the first retained segment, and park on the commit futex when they reach the tail - no
// polling.  Records are returned as pointers into the mapping (zero copy) valid until the next read.
//
// First touch: a fresh tmpfs page costs the writer a fault + allocation + zeroing (~3 us per 4 KiB measured: 8 ms for the
// 8.9 MB a preprocessor chunk publishes, against 1.2 ms for the copy itself).  A writer handle therefore owns a helper
// thread that populates the segment AHEAD of the append position (madvise(MADV_POPULATE_WRITE), a window of 32 MiB): the
// kernel work happens on another core, the appending thread copies into mapped pages.  `prl_log_appendv` gathers a record
// from several source ranges (header + the columns of a batch) straight into the segment - no intermediate record buffer.
//
// Retention: by default nothing is ever dropped (like a file on disk; control topics are tiny).  A log
// created with PRL_LOG_TRIM unlinks segments that every REGISTERED reader has left behind - the
// bulk topics (`training_data`, `actor`) use it, their single consumer is the trainer / preprocessor.
#include <atomic>
#include <cerrno>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <climits>
#include <csignal>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include <new>
#include <string>

#include <fcntl.h>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "prl_common.h"

namespace {

constexpr uint64_t kCtlMagic = 0x50524c4c4f474331ull;  // "PRLLOGC1"
constexpr uint64_t kSegMagic = 0x50524c4c4f475331ull;  // "PRLLOGS1"
constexpr int kMaxReaders = 64;
constexpr uint64_t kFree = ~0ull;

struct alignas(64) LogCtl {
  uint64_t magic;
  uint64_t segment_bytes;  // default payload capacity of a segment
  uint32_t flags;          // PRL_LOG_TRIM
  uint32_t _pad;
  alignas(64) std::atomic<uint32_t> lock;       // 0 free, else thread id of the writer holding it
  alignas(64) std::atomic<uint32_t> commits;    // futex word: bumped per committed record
  alignas(64) std::atomic<uint64_t> n_segments;  // segments created so far (indices 0 .. n-1)
  std::atomic<uint64_t> first_segment;           // oldest segment still linked
  std::atomic<uint64_t> n_records;
  std::atomic<uint64_t> n_bytes;
  alignas(64) std::atomic<uint64_t> reader_segment[kMaxReaders];  // kFree or the segment the reader is in
  std::atomic<uint32_t> reader_pid[kMaxReaders];
};

struct alignas(64) SegHeader {
  uint64_t magic;
  uint64_t index;
  uint64_t capacity;  // payload bytes after this header
  std::atomic<uint64_t> committed;
  std::atomic<uint32_t> sealed;
};

long futex_wait(std::atomic<uint32_t>* addr, uint32_t expected, const timespec* ts) {
  return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAIT, expected, ts, nullptr, 0);
}
long futex_wake_all(std::atomic<uint32_t>* addr) {
  return syscall(SYS_futex, reinterpret_cast<uint32_t*>(addr), FUTEX_WAKE, INT_MAX, nullptr, nullptr, 0);
}
int64_t now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}
std::string shm_name(const std::string& name) { return (!name.empty() && name[0] == '/') ? name : "/" + name; }
std::string seg_name(const std::string& base, uint64_t k) { return base + "." + std::to_string(k); }
uint64_t pad8(uint64_t n) { return (n + 7) & ~7ull; }

struct Mapping {
  uint8_t* base = nullptr;
  size_t bytes = 0;
  void reset() {
    if (base) munmap(base, bytes);
    base = nullptr;
    bytes = 0;
  }
};

// create (exclusive) or open a shm object and map it; *created tells which happened
int map_object(const std::string& name, size_t create_bytes, bool may_create, Mapping* m, bool* created) {
  *created = false;
  int fd = -1;
  if (may_create) {
    fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0) {
      *created = true;
      if (ftruncate(fd, (off_t)create_bytes) != 0) {
        const int e = errno;
        close(fd);
        shm_unlink(name.c_str());
        return prl::set_error(PRL_ENOMEM, "ftruncate(%s, %zu) failed: %s", name.c_str(), create_bytes, strerror(e));
      }
    } else if (errno != EEXIST) {
      return prl::set_error(PRL_EFAULT, "shm_open(%s) failed: %s", name.c_str(), strerror(errno));
    }
  }
  if (fd < 0) {
    fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (fd < 0) return prl::set_error(PRL_EFAULT, "shm_open(%s) failed: %s", name.c_str(), strerror(errno));
  }
  struct stat st;
  if (fstat(fd, &st) != 0 || st.st_size == 0) {  // the creator has not sized it yet
    close(fd);
    return prl::set_error(PRL_EAGAIN, "%s is being created", name.c_str());
  }
  void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return prl::set_error(PRL_ENOMEM, "mmap(%s) failed: %s", name.c_str(), strerror(errno));
  m->base = static_cast<uint8_t*>(p);
  m->bytes = (size_t)st.st_size;
  return PRL_OK;
}

}  // namespace

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23  // Linux >= 5.14
#endif

// Populates the pages of the writer's current segment ahead of the append position, on its own thread.
// (Two helpers claiming alternate slices were measured: SLOWER - 2.68 ms against 2.05 ms of publisher time per 8.9 MB chunk into one
// log, profiles/r05h_*: populating one tmpfs file from two threads contends in the kernel.)
struct Prefaulter {
  static constexpr uint64_t kWindow = 32ull << 20;  // stay this far ahead of `committed`
  static constexpr uint64_t kSlice = 2ull << 20;    // one madvise call
  std::mutex m;
  std::condition_variable cv;
  std::thread* th = nullptr;  // on the heap: a forked child abandons it (the thread does not exist there) instead of destroying it
  pid_t owner = 0;
  bool stop = false, disabled = false;
  bool busy = false;        // a madvise slice is in flight (the mutex is NOT held during the call)
  std::condition_variable idle;
  uint8_t* base = nullptr;  // payload start of the segment being written
  uint64_t capacity = 0, want = 0, done = 0, gen = 0;

  void run() {
    std::unique_lock<std::mutex> lk(m);
    for (;;) {
      cv.wait(lk, [&] { return stop || (!disabled && base && done < want); });
      if (stop) return;
      const uint64_t g = gen, from = done, n = (want - done < kSlice) ? want - done : kSlice;
      uint8_t* p = base + from;
      busy = true;
      lk.unlock();
      // page-align inwards: the first partial page was touched by the header / previous record already
      const uintptr_t a = (reinterpret_cast<uintptr_t>(p) + 4095) & ~uintptr_t(4095);
      const uintptr_t e = (reinterpret_cast<uintptr_t>(p) + n) & ~uintptr_t(4095);
      int rc = 0;
      if (e > a) rc = madvise(reinterpret_cast<void*>(a), e - a, MADV_POPULATE_WRITE);
      const int err = rc != 0 ? errno : 0;
      lk.lock();
      busy = false;
      idle.notify_all();
      // kernel without MADV_POPULATE_WRITE: appends fault their pages themselves.  Only believed while the segment is still the one
      // the call was aimed at: after a switch the old range may be unmapped or belong to someone else, and its errors mean nothing
      if (err == EINVAL && g == gen) disabled = true;
      if (g == gen) done = from + n;  // a segment change in between restarts from its own offset
    }
  }
  // called by the appending thread BEFORE it unmaps (or replaces) the segment the helper may be populating: forget the range and
  // wait for a slice in flight to return.  Without it the madvise of a stale range could land on whatever is mapped at that
  // address next - another log's tmpfs segment, a fresh anonymous arena - and allocate up to 2 MiB of pages there per slice: no
  // data is touched, but shm / RSS grow silently (round-4 advisor finding).
  void quiesce() {
    if (getpid() != owner) return;
    std::unique_lock<std::mutex> lk(m);
    base = nullptr;
    want = done = 0;
    ++gen;
    idle.wait(lk, [&] { return !busy; });
  }
  // called by the appending thread (under the log's writer lock): the segment now being written and how far it is filled
  void target(uint8_t* payload, uint64_t cap, uint64_t committed, bool new_segment) {
    if (getpid() != owner) return;  // a handle inherited through fork(): no helper here, and the mutex may have been held at the fork
    std::lock_guard<std::mutex> lk(m);
    if (new_segment || payload != base) {
      base = payload;
      capacity = cap;
      done = committed;
      ++gen;
    }
    if (done < committed) done = committed;
    const uint64_t w = committed + kWindow < cap ? committed + kWindow : cap;
    if (w > want || new_segment) want = w;
    if (want > cap) want = cap;
    if (done < want) cv.notify_one();
  }
  void start() {
    owner = getpid();
    th = new (std::nothrow) std::thread([this] { run(); });
  }
  // true: the helper is gone and the object may be deleted; false (a forked child): leave everything alone
  bool shutdown() {
    if (getpid() != owner) return false;
    if (th) {
      {
        std::lock_guard<std::mutex> lk(m);
        stop = true;
      }
      cv.notify_all();
      th->join();
      delete th;
      th = nullptr;
    }
    return true;
  }
};

struct prl_log {
  std::string name;
  Mapping ctl_map;
  LogCtl* ctl = nullptr;
  // writer side: the segment being appended to
  Mapping wseg;
  uint64_t wseg_index = kFree;
  Prefaulter* prefault = nullptr;  // created by the first append of this handle
  // reader side
  Mapping rseg;
  uint64_t rseg_index = kFree;
  uint64_t roffset = 0;
  int reader_slot = -1;
};

namespace {

SegHeader* seg_hdr(const Mapping& m) { return reinterpret_cast<SegHeader*>(m.base); }
uint8_t* seg_data(const Mapping& m) { return m.base + sizeof(SegHeader); }

int open_segment(const std::string& base, uint64_t k, Mapping* m) {
  bool created = false;
  Mapping tmp;
  if (int rc = map_object(seg_name(base, k), 0, false, &tmp, &created)) return rc;
  if (tmp.bytes < sizeof(SegHeader) || seg_hdr(tmp)->magic != kSegMagic) {
    tmp.reset();
    return prl::set_error(PRL_EAGAIN, "segment %llu of %s is not initialised yet", (unsigned long long)k, base.c_str());
  }
  m->reset();
  *m = tmp;
  return PRL_OK;
}

int create_segment(const std::string& base, uint64_t k, uint64_t capacity, Mapping* m) {
  const std::string nm = seg_name(base, k);
  shm_unlink(nm.c_str());  // a stale object of a previous incarnation
  bool created = false;
  Mapping tmp;
  if (int rc = map_object(nm, sizeof(SegHeader) + capacity, true, &tmp, &created)) return rc;
  SegHeader* h = new (tmp.base) SegHeader();
  h->index = k;
  h->capacity = capacity;
  h->committed.store(0, std::memory_order_relaxed);
  h->sealed.store(0, std::memory_order_relaxed);
  std::atomic_thread_fence(std::memory_order_release);
  h->magic = kSegMagic;
  m->reset();
  *m = tmp;
  return PRL_OK;
}

// The lock word holds the kernel thread id of the owner (unique system-wide, unlike a pid it tells two
// writers of one process apart).  A holder that died is detected through /proc/<tid>.
bool thread_alive(uint32_t tid) {
  char path[48];
  snprintf(path, sizeof path, "/proc/%u", tid);
  return access(path, F_OK) == 0;
}

void lock_ctl(LogCtl* c) {
  const uint32_t me = (uint32_t)syscall(SYS_gettid);
  for (int spins = 0;; ++spins) {
    uint32_t cur = 0;
    if (c->lock.compare_exchange_strong(cur, me, std::memory_order_acquire)) return;
    if (spins > 64) {
      if (spins % 256 == 65 && !thread_alive(cur)) {  // the holder died: take the lock over
        if (c->lock.compare_exchange_strong(cur, me, std::memory_order_acquire)) return;
        continue;
      }
      timespec ts{0, 2000000};
      futex_wait(&c->lock, cur, &ts);
    }
  }
}
void unlock_ctl(LogCtl* c) {
  c->lock.store(0, std::memory_order_release);
  futex_wake_all(&c->lock);
}

void trim(prl_log* l) {
  LogCtl* c = l->ctl;
  if (!(c->flags & PRL_LOG_TRIM)) return;
  uint64_t lo = kFree;
  int registered = 0;
  for (int i = 0; i < kMaxReaders; ++i) {
    const uint64_t s = c->reader_segment[i].load(std::memory_order_acquire);
    if (s == kFree) continue;
    const uint32_t pid = c->reader_pid[i].load(std::memory_order_relaxed);
    if (pid && kill((pid_t)pid, 0) != 0 && errno == ESRCH) {  // a reader that died without closing
      c->reader_segment[i].store(kFree, std::memory_order_release);
      continue;
    }
    ++registered;
    if (s < lo) lo = s;
  }
  if (!registered) return;  // nobody has read yet: a first reader must still find record 0
  uint64_t first = c->first_segment.load(std::memory_order_relaxed);
  const uint64_t last = c->n_segments.load(std::memory_order_relaxed) - 1;
  while (first < lo && first < last) {
    shm_unlink(seg_name(l->name, first).c_str());
    ++first;
  }
  c->first_segment.store(first, std::memory_order_release);
}

}  // namespace

extern "C" int prl_log_open(const char* name, uint64_t segment_bytes, int32_t flags, prl_log** out) {
  PRL_CHECK_ARG(name && out, "null argument");
  const bool may_create = flags & PRL_LOG_CREATE;
  PRL_CHECK_ARG(!may_create || segment_bytes >= 64, "segment_bytes must be >= 64 when creating");
  auto* l = new (std::nothrow) prl_log();
  if (!l) return prl::set_error(PRL_ENOMEM, "out of memory");
  l->name = shm_name(name);
  if (flags & PRL_LOG_TRUNCATE) prl_log_unlink(name);
  bool created = false;
  if (int rc = map_object(l->name, sizeof(LogCtl), may_create, &l->ctl_map, &created)) {
    delete l;
    return rc;
  }
  l->ctl = reinterpret_cast<LogCtl*>(l->ctl_map.base);
  if (created) {
    LogCtl* c = new (l->ctl_map.base) LogCtl();
    c->segment_bytes = segment_bytes;
    c->flags = (uint32_t)(flags & PRL_LOG_TRIM);
    c->lock.store(0);
    c->commits.store(0);
    c->n_segments.store(0);
    c->first_segment.store(0);
    c->n_records.store(0);
    c->n_bytes.store(0);
    for (int i = 0; i < kMaxReaders; ++i) {
      c->reader_segment[i].store(kFree);
      c->reader_pid[i].store(0);
    }
    Mapping seg0;
    if (int rc = create_segment(l->name, 0, segment_bytes, &seg0)) {
      l->ctl_map.reset();
      shm_unlink(l->name.c_str());
      delete l;
      return rc;
    }
    seg0.reset();
    c->n_segments.store(1, std::memory_order_release);
    std::atomic_thread_fence(std::memory_order_release);
    c->magic = kCtlMagic;
  } else if (l->ctl_map.bytes < sizeof(LogCtl) || l->ctl->magic != kCtlMagic) {
    l->ctl_map.reset();
    delete l;
    return prl::set_error(PRL_EAGAIN, "log %s is not initialised yet", name);
  }
  if (flags & PRL_LOG_READER) {  // register a cursor so that a trimming writer knows where this reader is
    const uint64_t first = l->ctl->first_segment.load(std::memory_order_acquire);
    for (int i = 0; i < kMaxReaders && l->reader_slot < 0; ++i) {
      uint64_t expect = kFree;
      if (l->ctl->reader_segment[i].compare_exchange_strong(expect, first, std::memory_order_acq_rel)) {
        l->ctl->reader_pid[i].store((uint32_t)getpid(), std::memory_order_relaxed);
        l->reader_slot = i;
      }
    }
    // more than kMaxReaders readers: the extra ones read unregistered (they cannot hold back a trim)
  }
  *out = l;
  return PRL_OK;
}

namespace {

// Reserve room for a record of `nbytes` in the last segment (rolling over when it does not fit).  Called with the writer
// lock held; on PRL_OK *dst points at the record's length word and the caller fills [dst + 8, dst + 8 + nbytes), then
// calls publish_record.
int reserve_record(prl_log* l, uint64_t nbytes, uint8_t** dst, SegHeader** hdr, uint64_t* off_out) {
  LogCtl* c = l->ctl;
  const uint64_t need = 8 + pad8(nbytes);
  int rc = PRL_OK;
  SegHeader* h = nullptr;
  uint64_t off = 0;
  bool new_segment = false;
  for (;;) {  // find (or open) the segment this record goes to
    const uint64_t last = c->n_segments.load(std::memory_order_acquire) - 1;
    if (l->wseg_index != last) {
      if (l->prefault) l->prefault->quiesce();
      rc = open_segment(l->name, last, &l->wseg);
      if (rc != PRL_OK) break;
      l->wseg_index = last;
      new_segment = true;
    }
    h = seg_hdr(l->wseg);
    if (h->sealed.load(std::memory_order_acquire)) {
      // a writer died between sealing this segment and publishing its successor (which exists: it is created
      // before the seal, and readers may already be waiting in it): adopt the successor
      Mapping next;
      rc = open_segment(l->name, last + 1, &next);
      if (rc != PRL_OK) break;
      c->n_segments.store(last + 2, std::memory_order_release);
      if (l->prefault) l->prefault->quiesce();
      l->wseg.reset();
      l->wseg = next;
      l->wseg_index = last + 1;
      new_segment = true;
      continue;
    }
    off = h->committed.load(std::memory_order_relaxed);
    if (off + need <= h->capacity) break;
    // the record does not fit.  Order: create the successor, seal this segment, publish the new count - a reader
    // that sees `sealed` finds the successor, and a writer that finds `sealed` on the last counted segment knows
    // its predecessor stopped between the last two steps
    Mapping next;
    const uint64_t cap = need > c->segment_bytes ? need : c->segment_bytes;
    rc = create_segment(l->name, last + 1, cap, &next);
    if (rc != PRL_OK) break;
    h->sealed.store(1, std::memory_order_release);
    c->n_segments.store(last + 2, std::memory_order_release);
    if (l->prefault) l->prefault->quiesce();
    l->wseg.reset();
    l->wseg = next;
    l->wseg_index = last + 1;
    new_segment = true;
    trim(l);
  }
  if (rc != PRL_OK) return rc;
  if (!l->prefault && (c->segment_bytes >= (4ull << 20))) {  // bulk topics only: control topics have tiny segments
    l->prefault = new (std::nothrow) Prefaulter();
    if (l->prefault) l->prefault->start();
    new_segment = true;
  }
  if (l->prefault) l->prefault->target(seg_data(l->wseg), h->capacity, off + need, new_segment);
  *dst = seg_data(l->wseg) + off;
  *hdr = h;
  *off_out = off;
  return PRL_OK;
}

void publish_record(prl_log* l, SegHeader* h, uint64_t off, uint64_t nbytes) {
  LogCtl* c = l->ctl;
  h->committed.store(off + 8 + pad8(nbytes), std::memory_order_release);
  c->n_records.fetch_add(1, std::memory_order_relaxed);
  c->n_bytes.fetch_add(nbytes, std::memory_order_relaxed);
  c->commits.fetch_add(1, std::memory_order_release);
  futex_wake_all(&c->commits);
}

}  // namespace

extern "C" int prl_log_append(prl_log* l, const void* data, uint64_t nbytes) {
  PRL_CHECK_ARG(l && (data || nbytes == 0), "null argument");
  LogCtl* c = l->ctl;
  lock_ctl(c);
  uint8_t* p = nullptr;
  SegHeader* h = nullptr;
  uint64_t off = 0;
  const int rc = reserve_record(l, nbytes, &p, &h, &off);
  if (rc == PRL_OK) {
    memcpy(p, &nbytes, 8);
    if (nbytes) memcpy(p + 8, data, nbytes);
    publish_record(l, h, off, nbytes);
  }
  unlock_ctl(c);
  return rc;
}

extern "C" int prl_log_appendv(prl_log* l, const prl_log_iov* iov, int32_t n_iov, uint64_t nbytes) {
  PRL_CHECK_ARG(l && (iov || n_iov == 0) && n_iov >= 0, "null argument");
  uint64_t at = 0;
  for (int i = 0; i < n_iov; ++i) {  // ascending, non-overlapping, inside the record
    PRL_CHECK_ARG(iov[i].offset >= at && iov[i].offset + iov[i].nbytes <= nbytes && (iov[i].ptr || iov[i].nbytes == 0),
                  "iov[%d] (offset %llu, %llu bytes) does not fit a record of %llu bytes after the previous piece", i,
                  (unsigned long long)iov[i].offset, (unsigned long long)iov[i].nbytes, (unsigned long long)nbytes);
    at = iov[i].offset + iov[i].nbytes;
  }
  LogCtl* c = l->ctl;
  lock_ctl(c);
  uint8_t* p = nullptr;
  SegHeader* h = nullptr;
  uint64_t off = 0;
  const int rc = reserve_record(l, nbytes, &p, &h, &off);
  if (rc == PRL_OK) {
    memcpy(p, &nbytes, 8);
    uint8_t* rec = p + 8;
    at = 0;
    for (int i = 0; i < n_iov; ++i) {
      if (iov[i].offset > at) memset(rec + at, 0, iov[i].offset - at);  // alignment gaps are zeros, whatever the page held
      if (iov[i].nbytes) memcpy(rec + iov[i].offset, iov[i].ptr, iov[i].nbytes);
      at = iov[i].offset + iov[i].nbytes;
    }
    if (nbytes > at) memset(rec + at, 0, nbytes - at);
    publish_record(l, h, off, nbytes);
  }
  unlock_ctl(c);
  return rc;
}

extern "C" int prl_log_read(prl_log* l, const void** ptr, uint64_t* nbytes, int64_t timeout_ms) {
  PRL_CHECK_ARG(l && ptr && nbytes, "null argument");
  LogCtl* c = l->ctl;
  const int64_t deadline = timeout_ms < 0 ? -1 : now_ms() + timeout_ms;
  for (;;) {
    const uint32_t seen = c->commits.load(std::memory_order_acquire);
    if (l->rseg_index == kFree) {
      const uint64_t first = c->first_segment.load(std::memory_order_acquire);
      const int rc = open_segment(l->name, first, &l->rseg);
      if (rc == PRL_OK) {
        l->rseg_index = first;
        l->roffset = 0;
        if (l->reader_slot >= 0) c->reader_segment[l->reader_slot].store(first, std::memory_order_release);
      } else if (rc != PRL_EAGAIN && c->first_segment.load(std::memory_order_acquire) == first) {
        return rc;
      } else {
        continue;  // trimmed under our feet or still being initialised: look again
      }
    }
    SegHeader* h = seg_hdr(l->rseg);
    const uint64_t committed = h->committed.load(std::memory_order_acquire);
    if (l->roffset < committed) {
      const uint8_t* p = seg_data(l->rseg) + l->roffset;
      uint64_t n;
      memcpy(&n, p, 8);
      *ptr = p + 8;
      *nbytes = n;
      l->roffset += 8 + pad8(n);
      return PRL_OK;
    }
    if (h->sealed.load(std::memory_order_acquire)) {
      // `committed` may have moved between the two loads: re-check before leaving the segment
      if (l->roffset < h->committed.load(std::memory_order_acquire)) continue;
      const uint64_t next = l->rseg_index + 1;
      if (int rc = open_segment(l->name, next, &l->rseg)) return rc;
      l->rseg_index = next;
      l->roffset = 0;
      if (l->reader_slot >= 0) c->reader_segment[l->reader_slot].store(next, std::memory_order_release);
      continue;
    }
    if (timeout_ms == 0) return prl::set_error(PRL_EAGAIN, "log tail reached");
    if (deadline >= 0) {
      const int64_t left = deadline - now_ms();
      if (left <= 0) return prl::set_error(PRL_ETIMEDOUT, "log tail reached (timeout)");
      timespec ts{(time_t)(left / 1000), (long)((left % 1000) * 1000000)};
      futex_wait(&c->commits, seen, &ts);
    } else {
      futex_wait(&c->commits, seen, nullptr);
    }
  }
}

extern "C" int prl_log_stats(prl_log* l, uint64_t* n_records, uint64_t* n_bytes, uint64_t* first_segment,
                             uint64_t* n_segments) {
  PRL_CHECK_ARG(l, "null log");
  if (n_records) *n_records = l->ctl->n_records.load(std::memory_order_acquire);
  if (n_bytes) *n_bytes = l->ctl->n_bytes.load(std::memory_order_acquire);
  if (first_segment) *first_segment = l->ctl->first_segment.load(std::memory_order_acquire);
  if (n_segments) *n_segments = l->ctl->n_segments.load(std::memory_order_acquire);
  return PRL_OK;
}

extern "C" int prl_log_close(prl_log* l) {
  if (!l) return PRL_OK;
  if (l->reader_slot >= 0) l->ctl->reader_segment[l->reader_slot].store(kFree, std::memory_order_release);
  if (l->prefault) {  // before the segment is unmapped
    if (l->prefault->shutdown()) delete l->prefault;
    l->prefault = nullptr;
  }
  l->wseg.reset();
  l->rseg.reset();
  l->ctl_map.reset();
  delete l;
  return PRL_OK;
}

extern "C" int prl_log_unlink(const char* name) {
  PRL_CHECK_ARG(name, "null name");
  const std::string base = shm_name(name);
  uint64_t first = 0, n = 0;
  {
    Mapping m;
    bool created = false;
    if (map_object(base, 0, false, &m, &created) == PRL_OK) {
      if (m.bytes >= sizeof(LogCtl) && reinterpret_cast<LogCtl*>(m.base)->magic == kCtlMagic) {
        auto* c = reinterpret_cast<LogCtl*>(m.base);
        first = c->first_segment.load();
        n = c->n_segments.load();
      }
      m.reset();
    }
  }
  for (uint64_t k = first; k < n + 1; ++k) shm_unlink(seg_name(base, k).c_str());
  shm_unlink(base.c_str());
  return PRL_OK;
}
