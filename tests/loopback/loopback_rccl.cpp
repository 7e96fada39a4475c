// TEST INFRASTRUCTURE, never part of libbp5.so: a host-shared-memory stand-in for the nine RCCL calls of the library, linked
// into libbp5_loopback.so only (csrc/Makefile target `loopback`).  Purpose: run N ranks of the product's multi-rank code
// (z-slab meshes, pack/unpack kernels, exchange schedules, fused dot-product corrections, per-iteration all-reduce) as N
// PROCESSES ON ONE GPU, which RCCL refuses ("Duplicate GPU detected").
//
// Semantics kept, and -- unlike the first version of this shim -- kept ASYNCHRONOUSLY, the way RCCL behaves:
//   * every call only ENQUEUES work on the caller's stream and returns; the transfer happens when the stream gets there
//     (a D2H copy into a pinned staging buffer, a host function that talks to the peer through shared memory, an H2D copy),
//     so a product bug in the ordering BETWEEN streams -- a missing hipStreamWaitEvent around a *_start / *_finish pair, a send
//     buffer re-packed while its message has not left, a ghost range read before its receive has landed -- is not hidden by a
//     blocking call and shows up as a wrong result;
//   * BP5_LOOPBACK_DELAY_US=<n>: every transfer sleeps n microseconds on its stream before it touches its buffers (the
//     communication lags behind the compute stream, as over a slow link);
//   * BP5_LOOPBACK_POISON=0 switches OFF the default poisoning: a receive first fills its destination with NaN (stream-ordered),
//     so anything that reads the destination before the message has landed computes NaN;
//   * grouped send/recv (all sends of a group are buffered before its receives wait: no deadlock between neighbours), messages
//     between a pair of ranks matched in order, all-reduce summed in rank order on every rank (the same bits everywhere).
// A failure inside a host function (timeout waiting for a peer, message size mismatch) latches an error: the destination is
// filled with NaN and every later call of the shim returns ncclSystemError.  Nothing here is timed or measured.
#include "loopback_rename.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <limits>
#include <random>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
constexpr int MAX_RANKS = 6, SLOTS = 4, MAX_REDUCE = 1024, RING = 16;
constexpr size_t SLOT_BYTES = size_t(4) << 20;
constexpr double TIMEOUT_S = 120.0;

struct Mailbox {
  std::atomic<uint64_t> written, consumed;
  size_t bytes[SLOTS];
  alignas(64) unsigned char data[SLOTS][SLOT_BYTES];
};
struct Region {
  std::atomic<int> attached, arrived, generation;
  double reduce[MAX_RANKS][MAX_REDUCE];
  Mailbox box[MAX_RANKS][MAX_RANKS]; // [from][to]
};
struct Comm {
  Region *reg;
  int rank, n;
};
struct Op {
  bool send;
  void *dev;
  size_t bytes;
  int peer;
  Comm *c;
  hipStream_t stream;
};
thread_local int group_depth = 0;
thread_local std::vector<Op> queued;
std::atomic<int> g_error{0}; // latched by host functions

int delay_us()
{
  static const int d = [] { const char *e = getenv("BP5_LOOPBACK_DELAY_US"); return e ? atoi(e) : 0; }();
  return d;
}
bool poison()
{
  static const bool p = [] { const char *e = getenv("BP5_LOOPBACK_POISON"); return !(e && e[0] == '0'); }();
  return p;
}

// pinned staging buffers: a ring; a slot is reused only after the stream work that last touched it has completed
struct Staging {
  void *host = nullptr;
  hipEvent_t done = nullptr;
  bool used = false;
};
Staging g_ring[RING];
unsigned g_next = 0;
Staging *acquire()
{
  Staging &s = g_ring[g_next++ % RING];
  if (!s.host) {
    if (hipHostMalloc(&s.host, SLOT_BYTES) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess) return nullptr;
  }
  if (s.used && hipEventSynchronize(s.done) != hipSuccess) return nullptr; // back-pressure: the host runs at most RING transfers ahead
  s.used = true;
  return &s;
}

template <class F> bool wait_for(F ready)
{
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0; !ready(); ++spin) {
    if (spin > 200) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) return false;
  }
  return true;
}
bool barrier(Comm *c)
{
  Region *r = c->reg;
  const int gen = r->generation.load();
  if (r->arrived.fetch_add(1) + 1 == c->n) {
    r->arrived.store(0);
    r->generation.fetch_add(1);
    return true;
  }
  return wait_for([&] { return r->generation.load() != gen; });
}
void fill_nan(void *host, size_t bytes)
{
  double *d = static_cast<double *>(host);
  for (size_t i = 0; i < bytes / sizeof(double); ++i) d[i] = std::numeric_limits<double>::quiet_NaN();
}

// ---- host functions (run by the HIP runtime when the stream reaches them; no HIP calls inside)
struct Xfer {
  Comm *c;
  int peer;
  void *host;
  size_t bytes;
};
void host_sleep(void *) { std::this_thread::sleep_for(std::chrono::microseconds(delay_us())); }
void host_publish(void *p) // the staged message -> the mailbox to `peer`
{
  Xfer *x = static_cast<Xfer *>(p);
  Mailbox &m = x->c->reg->box[x->c->rank][x->peer];
  if (!wait_for([&] { return m.written.load() - m.consumed.load() < SLOTS; })) g_error.store(1);
  else {
    const uint64_t k = m.written.load() % SLOTS;
    std::memcpy(m.data[k], x->host, x->bytes);
    m.bytes[k] = x->bytes;
    m.written.fetch_add(1, std::memory_order_release);
  }
  delete x;
}
void host_consume(void *p) // the next message from `peer` -> the staging buffer
{
  Xfer *x = static_cast<Xfer *>(p);
  Mailbox &m = x->c->reg->box[x->peer][x->c->rank];
  if (!wait_for([&] { return m.written.load(std::memory_order_acquire) > m.consumed.load(); })) {
    g_error.store(1);
    fill_nan(x->host, x->bytes);
  } else {
    const uint64_t k = m.consumed.load() % SLOTS;
    if (m.bytes[k] != x->bytes) {
      std::fprintf(stderr, "loopback: rank %d expected %zu bytes from rank %d, message has %zu\n", x->c->rank, x->bytes, x->peer, m.bytes[k]);
      g_error.store(2);
      fill_nan(x->host, x->bytes);
    } else
      std::memcpy(x->host, m.data[k], x->bytes);
    m.consumed.fetch_add(1, std::memory_order_release);
  }
  delete x;
}
void host_reduce(void *p) // staged contribution -> sum over the ranks, in rank order, back into the staging buffer
{
  Xfer *x = static_cast<Xfer *>(p);
  Comm *c = x->c;
  const size_t count = x->bytes / sizeof(double);
  double *mine = static_cast<double *>(x->host);
  std::memcpy(c->reg->reduce[c->rank], mine, x->bytes);
  bool ok = barrier(c);
  if (ok) {
    for (size_t i = 0; i < count; ++i) {
      double a = 0.0;
      for (int r = 0; r < c->n; ++r) a += c->reg->reduce[r][i]; // rank order: the same bits on every rank
      mine[i] = a;
    }
    ok = barrier(c); // nobody overwrites its slot before everyone has read it
  }
  if (!ok) { g_error.store(1); fill_nan(x->host, x->bytes); }
  delete x;
}

#define HIPOK(x) do { if ((x) != hipSuccess) return ncclUnhandledCudaError; } while (0)
ncclResult_t run(const Op &op)
{
  if (g_error.load()) return ncclSystemError;
  if (op.bytes > SLOT_BYTES) return ncclInvalidArgument;
  Staging *st = acquire();
  if (!st) return ncclUnhandledCudaError;
  if (op.send) {
    if (delay_us() > 0) HIPOK(hipLaunchHostFunc(op.stream, host_sleep, nullptr)); // the message leaves late: its buffer must still be intact
    HIPOK(hipMemcpyAsync(st->host, op.dev, op.bytes, hipMemcpyDeviceToHost, op.stream));
    HIPOK(hipLaunchHostFunc(op.stream, host_publish, new Xfer{op.c, op.peer, st->host, op.bytes}));
  } else {
    if (poison()) HIPOK(hipMemsetAsync(op.dev, 0xff, op.bytes, op.stream)); // NaN until the message has landed
    if (delay_us() > 0) HIPOK(hipLaunchHostFunc(op.stream, host_sleep, nullptr));
    HIPOK(hipLaunchHostFunc(op.stream, host_consume, new Xfer{op.c, op.peer, st->host, op.bytes}));
    HIPOK(hipMemcpyAsync(op.dev, st->host, op.bytes, hipMemcpyHostToDevice, op.stream));
  }
  HIPOK(hipEventRecord(st->done, op.stream));
  return ncclSuccess;
}
ncclResult_t flush()
{
  std::vector<Op> ops;
  ops.swap(queued);
  for (int pass = 0; pass < 2; ++pass) // buffered sends first, then the receives
    for (const Op &op : ops)
      if (op.send == (pass == 0)) {
        const ncclResult_t r = run(op);
        if (r != ncclSuccess) return r;
      }
  return ncclSuccess;
}
ncclResult_t enqueue(bool send, const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (t != ncclDouble || !c || peer < 0 || peer >= c->n) return ncclInvalidArgument;
  queued.push_back(Op{send, const_cast<void *>(buf), count * sizeof(double), peer, c, s});
  return group_depth ? ncclSuccess : flush();
}
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
  std::memset(id, 0, sizeof(*id));
  std::random_device rd;
  std::snprintf(id->internal, sizeof(id->internal), "/bp5lb_%d_%08x%08x", (int)getpid(), rd(), rd());
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int n, ncclUniqueId id, int rank)
{
  if (n < 1 || n > MAX_RANKS || rank < 0 || rank >= n) return ncclInvalidArgument;
  id.internal[sizeof(id.internal) - 1] = 0;
  const int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, sizeof(Region)) != 0) { close(fd); return ncclSystemError; }
  void *p = mmap(nullptr, sizeof(Region), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); // fresh object: zero pages
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  Comm *c = new Comm{static_cast<Region *>(p), rank, n};
  c->reg->attached.fetch_add(1);
  const bool ok = wait_for([&] { return c->reg->attached.load() >= n; });
  if (rank == 0) shm_unlink(id.internal); // the mappings keep the memory alive; nothing is left behind in /dev/shm
  if (!ok) { munmap(p, sizeof(Region)); delete c; return ncclSystemError; }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (c) {
    hipDeviceSynchronize(); // host functions that still name this communicator have run
    munmap(c->reg, sizeof(Region));
    delete c;
  }
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r)
{
  switch (r) {
  case ncclSuccess: return "no error";
  case ncclUnhandledCudaError: return "loopback transport: HIP call failed";
  case ncclSystemError: return "loopback transport: shared memory failure, timeout waiting for a peer or message size mismatch";
  case ncclInvalidArgument: return "loopback transport: invalid argument";
  default: return "loopback transport: error";
  }
}
ncclResult_t ncclGroupStart()
{
  ++group_depth;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd()
{
  if (group_depth <= 0) return ncclInvalidUsage;
  return --group_depth ? ncclSuccess : flush();
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
  return enqueue(true, buf, count, t, peer, comm, s);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
  return enqueue(false, buf, count, t, peer, comm, s);
}
ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s)
{
  Comm *c = reinterpret_cast<Comm *>(comm);
  if (!c || t != ncclDouble || op != ncclSum || count > MAX_REDUCE) return ncclInvalidArgument;
  if (g_error.load()) return ncclSystemError;
  Staging *st = acquire();
  if (!st) return ncclUnhandledCudaError;
  if (delay_us() > 0) HIPOK(hipLaunchHostFunc(s, host_sleep, nullptr));
  HIPOK(hipMemcpyAsync(st->host, send, count * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPOK(hipLaunchHostFunc(s, host_reduce, new Xfer{c, 0, st->host, count * sizeof(double)}));
  HIPOK(hipMemcpyAsync(recv, st->host, count * sizeof(double), hipMemcpyHostToDevice, s));
  HIPOK(hipEventRecord(st->done, s));
  return ncclSuccess;
}
}
