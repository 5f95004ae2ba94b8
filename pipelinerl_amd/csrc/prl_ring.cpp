// prl_ring: bounded multi-producer / multi-consumer record queue in POSIX shared memory.
//
// Replaces the reference's pickle-in-shared-memory queue (pipelinerl/shared_memory_array.py:
// fixed slots + two multiprocessing.Queues of slot indices) and the polling JSONL file stream
// between preprocessor and trainer (pipelinerl/streams.py:249-346, 0.1 s re-read delay).
// Records are opaque byte strings (the Python host writes a binary SoA batch record).
//
// Algorithm: Vyukov bounded MPMC queue.  Slot i carries a sequence number; a producer may
// claim position p when seq == p, publishes with seq = p + 1; a consumer may claim when
// seq == p + 1 and recycles the slot with seq = p + n_slots.  Blocking waits park on two
// futex words (one bumped per publish, one per recycle) instead of sleeping in a poll loop.
#include <atomic>
#include <cerrno>
#include <climits>
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

constexpr uint64_t kMagic = 0x50524c52494e4731ull;  // "PRLRING1"
constexpr size_t kCacheLine = 64;

struct alignas(kCacheLine) RingHeader {
  uint64_t magic;
  uint32_t n_slots;
  uint32_t _pad0;
  uint64_t slot_bytes;   // payload capacity of one slot
  uint64_t slot_stride;  // bytes between slots (header + payload, cache-line rounded)
  uint64_t slots_offset; // offset of slot 0 from the mapping base
  std::atomic<uint64_t> max_record;
  alignas(kCacheLine) std::atomic<uint64_t> head;  // next position to produce
  alignas(kCacheLine) std::atomic<uint64_t> tail;  // next position to consume
  alignas(kCacheLine) std::atomic<uint32_t> put_events;  // futex word: bumped per publish
  alignas(kCacheLine) std::atomic<uint32_t> get_events;  // futex word: bumped per recycle
};

struct alignas(kCacheLine) SlotHeader {
  std::atomic<uint64_t> seq;
  uint64_t nbytes;
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

std::string shm_name(const char* name) {
  std::string s(name);
  if (s.empty() || s[0] != '/') s = "/" + s;
  return s;
}

}  // namespace

struct prl_ring {
  RingHeader* hdr;
  uint8_t* base;
  size_t map_bytes;
  bool owner;
  std::string name;

  SlotHeader* slot(uint64_t pos) const {
    return reinterpret_cast<SlotHeader*>(base + hdr->slots_offset +
                                         (pos % hdr->n_slots) * hdr->slot_stride);
  }
  uint8_t* payload(uint64_t pos) const { return reinterpret_cast<uint8_t*>(slot(pos)) + sizeof(SlotHeader); }
};

namespace {

// Park until *word != seen or the deadline passes.  Returns false on timeout.
bool park(std::atomic<uint32_t>* word, uint32_t seen, int64_t deadline_ms) {
  if (deadline_ms < 0) {
    futex_wait(word, seen, nullptr);
    return true;
  }
  const int64_t left = deadline_ms - now_ms();
  if (left <= 0) return false;
  timespec ts{(time_t)(left / 1000), (long)((left % 1000) * 1000000)};
  futex_wait(word, seen, &ts);
  return true;
}

}  // namespace

extern "C" int prl_ring_create(const char* name, uint32_t n_slots, uint64_t slot_bytes,
                               prl_ring** out) {
  PRL_CHECK_ARG(name && out, "null argument");
  PRL_CHECK_ARG(n_slots >= 1, "n_slots must be positive");
  PRL_CHECK_ARG(slot_bytes >= 1, "slot_bytes must be positive");
  const std::string nm = shm_name(name);
  shm_unlink(nm.c_str());
  const int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return prl::set_error(PRL_EFAULT, "shm_open(%s) failed: %s", nm.c_str(), strerror(errno));
  const uint64_t stride =
      (sizeof(SlotHeader) + slot_bytes + kCacheLine - 1) / kCacheLine * kCacheLine;
  const uint64_t slots_off = (sizeof(RingHeader) + kCacheLine - 1) / kCacheLine * kCacheLine;
  const size_t bytes = slots_off + stride * n_slots;
  if (ftruncate(fd, (off_t)bytes) != 0) {
    const int e = errno;
    close(fd);
    shm_unlink(nm.c_str());
    return prl::set_error(PRL_ENOMEM, "ftruncate(%zu) failed: %s", bytes, strerror(e));
  }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    shm_unlink(nm.c_str());
    return prl::set_error(PRL_ENOMEM, "mmap(%zu) failed: %s", bytes, strerror(errno));
  }
  auto* r = new (std::nothrow) prl_ring();
  if (!r) return prl::set_error(PRL_ENOMEM, "out of memory");
  r->base = static_cast<uint8_t*>(p);
  r->hdr = new (p) RingHeader();
  r->map_bytes = bytes;
  r->owner = true;
  r->name = nm;
  r->hdr->n_slots = n_slots;
  r->hdr->slot_bytes = slot_bytes;
  r->hdr->slot_stride = stride;
  r->hdr->slots_offset = slots_off;
  r->hdr->max_record.store(0);
  r->hdr->head.store(0);
  r->hdr->tail.store(0);
  r->hdr->put_events.store(0);
  r->hdr->get_events.store(0);
  for (uint32_t i = 0; i < n_slots; ++i) {
    SlotHeader* s = new (r->slot(i)) SlotHeader();
    s->seq.store(i, std::memory_order_relaxed);
    s->nbytes = 0;
  }
  std::atomic_thread_fence(std::memory_order_release);
  r->hdr->magic = kMagic;
  *out = r;
  return PRL_OK;
}

extern "C" int prl_ring_attach(const char* name, prl_ring** out) {
  PRL_CHECK_ARG(name && out, "null argument");
  const std::string nm = shm_name(name);
  const int fd = shm_open(nm.c_str(), O_RDWR, 0600);
  if (fd < 0) return prl::set_error(PRL_EFAULT, "shm_open(%s) failed: %s", nm.c_str(), strerror(errno));
  struct stat st;
  if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(RingHeader)) {
    close(fd);
    return prl::set_error(PRL_EFAULT, "ring %s is not initialised", nm.c_str());
  }
  void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return prl::set_error(PRL_ENOMEM, "mmap failed: %s", strerror(errno));
  auto* hdr = static_cast<RingHeader*>(p);
  if (hdr->magic != kMagic) {
    munmap(p, (size_t)st.st_size);
    return prl::set_error(PRL_EFAULT, "ring %s has a bad magic", nm.c_str());
  }
  auto* r = new (std::nothrow) prl_ring();
  if (!r) return prl::set_error(PRL_ENOMEM, "out of memory");
  r->base = static_cast<uint8_t*>(p);
  r->hdr = hdr;
  r->map_bytes = (size_t)st.st_size;
  r->owner = false;
  r->name = nm;
  *out = r;
  return PRL_OK;
}

extern "C" int prl_ring_reserve(prl_ring* r, void** slot_ptr, uint64_t* ticket, int64_t timeout_ms) {
  PRL_CHECK_ARG(r && slot_ptr && ticket, "null argument");
  const int64_t deadline = timeout_ms < 0 ? -1 : now_ms() + timeout_ms;
  RingHeader* h = r->hdr;
  for (;;) {
    const uint32_t ev = h->get_events.load(std::memory_order_acquire);
    uint64_t pos = h->head.load(std::memory_order_relaxed);
    for (;;) {
      SlotHeader* s = r->slot(pos);
      const uint64_t seq = s->seq.load(std::memory_order_acquire);
      const int64_t dif = (int64_t)(seq - pos);
      if (dif == 0) {
        if (h->head.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
          *slot_ptr = r->payload(pos);
          *ticket = pos;
          return PRL_OK;
        }
      } else if (dif < 0) {
        break;  // full
      } else {
        pos = h->head.load(std::memory_order_relaxed);
      }
    }
    if (timeout_ms == 0) return prl::set_error(PRL_EAGAIN, "ring full");
    if (!park(&h->get_events, ev, deadline)) return prl::set_error(PRL_ETIMEDOUT, "ring full (timeout)");
  }
}

extern "C" int prl_ring_commit(prl_ring* r, uint64_t ticket, uint64_t nbytes) {
  PRL_CHECK_ARG(r, "null ring");
  if (nbytes > r->hdr->slot_bytes)
    return prl::set_error(PRL_EMSGSIZE, "record of %llu bytes exceeds slot size %llu",
                          (unsigned long long)nbytes, (unsigned long long)r->hdr->slot_bytes);
  SlotHeader* s = r->slot(ticket);
  s->nbytes = nbytes;
  uint64_t prev = r->hdr->max_record.load(std::memory_order_relaxed);
  while (nbytes > prev && !r->hdr->max_record.compare_exchange_weak(prev, nbytes)) {
  }
  s->seq.store(ticket + 1, std::memory_order_release);
  r->hdr->put_events.fetch_add(1, std::memory_order_release);
  futex_wake_all(&r->hdr->put_events);
  return PRL_OK;
}

extern "C" int prl_ring_acquire(prl_ring* r, const void** slot_ptr, uint64_t* nbytes,
                                uint64_t* ticket, int64_t timeout_ms) {
  PRL_CHECK_ARG(r && slot_ptr && nbytes && ticket, "null argument");
  const int64_t deadline = timeout_ms < 0 ? -1 : now_ms() + timeout_ms;
  RingHeader* h = r->hdr;
  for (;;) {
    const uint32_t ev = h->put_events.load(std::memory_order_acquire);
    uint64_t pos = h->tail.load(std::memory_order_relaxed);
    for (;;) {
      SlotHeader* s = r->slot(pos);
      const uint64_t seq = s->seq.load(std::memory_order_acquire);
      const int64_t dif = (int64_t)(seq - (pos + 1));
      if (dif == 0) {
        if (h->tail.compare_exchange_weak(pos, pos + 1, std::memory_order_relaxed)) {
          *slot_ptr = r->payload(pos);
          *nbytes = s->nbytes;
          *ticket = pos;
          return PRL_OK;
        }
      } else if (dif < 0) {
        break;  // empty (or the head-of-line producer has not committed yet)
      } else {
        pos = h->tail.load(std::memory_order_relaxed);
      }
    }
    if (timeout_ms == 0) return prl::set_error(PRL_EAGAIN, "ring empty");
    if (!park(&h->put_events, ev, deadline)) return prl::set_error(PRL_ETIMEDOUT, "ring empty (timeout)");
  }
}

extern "C" int prl_ring_release(prl_ring* r, uint64_t ticket) {
  PRL_CHECK_ARG(r, "null ring");
  r->slot(ticket)->seq.store(ticket + r->hdr->n_slots, std::memory_order_release);
  r->hdr->get_events.fetch_add(1, std::memory_order_release);
  futex_wake_all(&r->hdr->get_events);
  return PRL_OK;
}

extern "C" int prl_ring_put(prl_ring* r, const void* data, uint64_t nbytes, int64_t timeout_ms) {
  PRL_CHECK_ARG(r && (data || nbytes == 0), "null argument");
  if (nbytes > r->hdr->slot_bytes)
    return prl::set_error(PRL_EMSGSIZE, "record of %llu bytes exceeds slot size %llu",
                          (unsigned long long)nbytes, (unsigned long long)r->hdr->slot_bytes);
  void* p = nullptr;
  uint64_t ticket = 0;
  if (int rc = prl_ring_reserve(r, &p, &ticket, timeout_ms)) return rc;
  if (nbytes) memcpy(p, data, nbytes);
  return prl_ring_commit(r, ticket, nbytes);
}

extern "C" int prl_ring_get(prl_ring* r, void* buf, uint64_t cap, uint64_t* nbytes, int64_t timeout_ms) {
  PRL_CHECK_ARG(r && nbytes && (buf || cap == 0), "null argument");
  const void* p = nullptr;
  uint64_t n = 0, ticket = 0;
  if (int rc = prl_ring_acquire(r, &p, &n, &ticket, timeout_ms)) return rc;
  *nbytes = n;
  int rc = PRL_OK;
  if (n > cap) {
    rc = prl::set_error(PRL_EMSGSIZE, "buffer of %llu bytes too small for record of %llu bytes",
                        (unsigned long long)cap, (unsigned long long)n);
  } else if (n) {
    memcpy(buf, p, n);
  }
  prl_ring_release(r, ticket);
  return rc;
}

extern "C" int prl_ring_size(prl_ring* r, uint64_t* n_ready) {
  PRL_CHECK_ARG(r && n_ready, "null argument");
  const uint64_t t = r->hdr->tail.load(std::memory_order_acquire);
  const uint64_t h = r->hdr->head.load(std::memory_order_acquire);
  *n_ready = h >= t ? h - t : 0;
  return PRL_OK;
}

extern "C" int prl_ring_capacity(prl_ring* r, uint32_t* n_slots, uint64_t* slot_bytes) {
  PRL_CHECK_ARG(r, "null ring");
  if (n_slots) *n_slots = r->hdr->n_slots;
  if (slot_bytes) *slot_bytes = r->hdr->slot_bytes;
  return PRL_OK;
}

extern "C" int prl_ring_max_record_bytes(prl_ring* r, uint64_t* nbytes) {
  PRL_CHECK_ARG(r && nbytes, "null argument");
  *nbytes = r->hdr->max_record.load(std::memory_order_relaxed);
  return PRL_OK;
}

extern "C" int prl_ring_close(prl_ring* r) {
  if (!r) return PRL_OK;
  munmap(r->base, r->map_bytes);
  if (r->owner) shm_unlink(r->name.c_str());
  delete r;
  return PRL_OK;
}

extern "C" int prl_ring_detach(prl_ring* r) {
  if (!r) return PRL_OK;
  munmap(r->base, r->map_bytes);
  delete r;
  return PRL_OK;
}

extern "C" int prl_ring_unlink(const char* name) {
  PRL_CHECK_ARG(name, "null name");
  shm_unlink(shm_name(name).c_str());
  return PRL_OK;
}
