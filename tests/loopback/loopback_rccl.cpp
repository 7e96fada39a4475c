// TEST INFRASTRUCTURE, never part of libbp5.so: a host-shared-memory stand-in for the nine RCCL calls of the library, linked
// into libbp5_loopback.so only (csrc/Makefile target `loopback`).  Purpose: run N ranks of the product's multi-rank code
// (z-slab meshes, pack/unpack kernels, exchange schedules, fused dot-product corrections, per-iteration all-reduce) as N
// PROCESSES ON ONE GPU, which RCCL refuses ("Duplicate GPU detected").  Semantics kept: stream order (the stream is drained
// before a transfer touches a buffer and the transfer is complete when the call returns), grouped send/recv (all sends of a
// group are buffered before its receives wait: no deadlock between neighbours), messages between a pair of ranks matched in
// order, all-reduce summed in rank order on every rank.  Nothing here is timed or measured.
#include "loopback_rename.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <random>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
constexpr int MAX_RANKS = 4, SLOTS = 4, MAX_REDUCE = 1024;
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
ncclResult_t run(const Op &op)
{
  if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
  if (op.bytes > SLOT_BYTES) return ncclInvalidArgument;
  Mailbox &m = op.send ? op.c->reg->box[op.c->rank][op.peer] : op.c->reg->box[op.peer][op.c->rank];
  if (op.send) {
    if (!wait_for([&] { return m.written.load() - m.consumed.load() < SLOTS; })) return ncclSystemError;
    const uint64_t k = m.written.load() % SLOTS;
    if (hipMemcpy(m.data[k], op.dev, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    m.bytes[k] = op.bytes;
    m.written.fetch_add(1, std::memory_order_release);
  } else {
    if (!wait_for([&] { return m.written.load(std::memory_order_acquire) > m.consumed.load(); })) return ncclSystemError;
    const uint64_t k = m.consumed.load() % SLOTS;
    if (m.bytes[k] != op.bytes) {
      std::fprintf(stderr, "loopback: rank %d expected %zu bytes from rank %d, message has %zu\n", op.c->rank, op.bytes, op.peer, m.bytes[k]);
      return ncclInvalidArgument;
    }
    if (hipMemcpy(op.dev, m.data[k], op.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    m.consumed.fetch_add(1, std::memory_order_release);
  }
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
  if (c) { munmap(c->reg, sizeof(Region)); delete c; }
  return ncclSuccess;
}
const char *ncclGetErrorString(ncclResult_t r)
{
  switch (r) {
  case ncclSuccess: return "no error";
  case ncclUnhandledCudaError: return "loopback transport: HIP call failed";
  case ncclSystemError: return "loopback transport: shared memory failure or timeout waiting for a peer";
  case ncclInvalidArgument: return "loopback transport: invalid argument or message size mismatch";
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
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  double mine[MAX_REDUCE], sum[MAX_REDUCE];
  if (hipMemcpy(mine, send, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  std::memcpy(c->reg->reduce[c->rank], mine, count * sizeof(double));
  if (!barrier(c)) return ncclSystemError;
  for (size_t i = 0; i < count; ++i) {
    double a = 0.0;
    for (int r = 0; r < c->n; ++r) a += c->reg->reduce[r][i]; // rank order: the same bits on every rank
    sum[i] = a;
  }
  if (!barrier(c)) return ncclSystemError; // nobody overwrites its slot before everyone has read it
  if (hipMemcpy(recv, sum, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
}
