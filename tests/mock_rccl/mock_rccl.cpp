// mock_rccl.cpp -- TEST INFRASTRUCTURE.  A stand-in for librccl.so.1 that lets several
// processes share ONE GPU: RCCL itself refuses two ranks on the same device
// ("invalid usage"), and a gpurun box has a single MI355X.  With
// MFFT_RCCL_LIB=<this library> the product's RcclComm (csrc/comm.hip) runs unchanged --
// unique-id rendezvous, ncclCommInitRank, grouped ncclSend/ncclRecv all-to-all-v on the
// plan's streams, ncclAllReduce/ncclBroadcast for the host helpers -- and only the wire is
// replaced: messages travel through POSIX shared-memory mailboxes with host staging.
// It implements exactly the entry points comm.hip resolves with dlsym.
//
// Semantics: operations complete inside ncclGroupEnd (or immediately outside a group) after
// a hipStreamSynchronize of their stream, i.e. stronger ordering than RCCL's stream-ordered
// asynchrony, never weaker.  Progress is made on all pending sends and receives in one polling
// loop, so any matching set of grouped operations completes without deadlock.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

extern "C" {

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3,
               ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

struct Mailbox {
  std::atomic<uint64_t> full;    // bytes available in `data` (0 = empty)
  char pad[56];
};

struct Header {
  std::atomic<int> attached;
  int nranks;
  size_t slot;
};

struct mockComm {
  int nranks, rank;
  size_t slot;
  char* base;
  size_t total;
  Header* hdr;
  Mailbox* boxes;      // [src][dst]
  char* data;          // [src][dst][slot]
  std::vector<char> stage;
  Mailbox& box(int s, int d) { return boxes[(size_t)s * nranks + d]; }
  char* slotp(int s, int d) { return data + ((size_t)s * nranks + d) * slot; }
};
typedef mockComm* ncclComm_t;

struct Op {
  int kind;            // 0 send, 1 recv
  char* buf;
  size_t bytes, done;
  int peer;
  ncclComm_t comm;
  hipStream_t stream;
};
static thread_local int g_depth = 0;
static thread_local std::vector<Op> g_ops;

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclInvalidArgument: return "invalid argument";
    case ncclInvalidUsage: return "invalid usage";
    case ncclSystemError: return "unhandled system error";
    default: return "mock rccl error";
  }
}

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/mockrccl_%d_%ld_%d", (int)getpid(), (long)time(nullptr), rand());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  const char* e = getenv("MOCK_RCCL_SLOT_KB");
  const size_t slot = (e ? (size_t)atol(e) : 4096) * 1024;
  mockComm* c = new mockComm();
  c->nranks = nranks;
  c->rank = rank;
  c->slot = slot;
  const size_t nbox = (size_t)nranks * nranks;
  c->total = 4096 + nbox * sizeof(Mailbox) + nbox * slot;
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) return ncclSystemError;
  if (ftruncate(fd, (off_t)c->total) != 0) return ncclSystemError;
  c->base = static_cast<char*>(mmap(nullptr, c->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
  close(fd);
  if (c->base == MAP_FAILED) return ncclSystemError;
  c->hdr = reinterpret_cast<Header*>(c->base);
  c->boxes = reinterpret_cast<Mailbox*>(c->base + 4096);
  c->data = c->base + 4096 + nbox * sizeof(Mailbox);
  c->stage.resize(slot);
  c->hdr->attached.fetch_add(1);
  while (c->hdr->attached.load() < nranks) usleep(200);      // every rank has mapped the segment
  if (rank == 0) { usleep(20000); shm_unlink(id.internal); }
  *out = c;
  return ncclSuccess;
}

// what a communicator says about itself (csrc/comm.hip RcclComm::get_option "rccl_*"); version 9.99.0 marks the stand-in
ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->nranks;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* rank) {
  if (!c || !rank) return ncclInvalidArgument;
  *rank = c->rank;
  return ncclSuccess;
}
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int* device) {
  if (!c || !device) return ncclInvalidArgument;
  return hipGetDevice(device) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}
ncclResult_t ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = 99900;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (c) {
    munmap(c->base, c->total);
    delete c;
  }
  return ncclSuccess;
}

static ncclResult_t progress(std::vector<Op>& ops) {
  // honour stream order: everything enqueued before the group must have produced the send data
  for (Op& o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
  size_t remaining = 0;
  for (Op& o : ops) remaining += (o.bytes - o.done) + (o.bytes == 0 && o.done == 0 ? 0 : 0);
  long idle = 0;
  while (remaining) {
    bool moved = false;
    for (Op& o : ops) {
      if (o.done == o.bytes) continue;
      mockComm* c = o.comm;
      if (o.kind == 0) {
        Mailbox& b = c->box(c->rank, o.peer);
        if (b.full.load(std::memory_order_acquire) != 0) continue;
        const size_t n = std::min(c->slot, o.bytes - o.done);
        if (hipMemcpy(c->slotp(c->rank, o.peer), o.buf + o.done, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        b.full.store(n, std::memory_order_release);
        o.done += n;
        remaining -= n;
        moved = true;
      } else {
        Mailbox& b = c->box(o.peer, c->rank);
        const size_t n = b.full.load(std::memory_order_acquire);
        if (n == 0) continue;
        if (n > o.bytes - o.done) return ncclInvalidUsage;          // peer sends more than we expect
        if (hipMemcpy(o.buf + o.done, c->slotp(o.peer, c->rank), n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        b.full.store(0, std::memory_order_release);
        o.done += n;
        remaining -= n;
        moved = true;
      }
    }
    if (!moved) {
      if (++idle > 600000) return ncclSystemError;               // ~60 s without progress: a peer is gone
      usleep(100);
    } else {
      idle = 0;
    }
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return progress(ops);
}

static ncclResult_t post(int kind, void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
  if (!comm || peer < 0 || peer >= comm->nranks) return ncclInvalidArgument;
  const size_t esz = dt == ncclFloat64 ? 8 : 1;
  g_ops.push_back(Op{kind, static_cast<char*>(buf), count * esz, 0, peer, comm, s});
  if (g_depth == 0) {
    std::vector<Op> ops;
    ops.swap(g_ops);
    return progress(ops);
  }
  return ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
  return post(0, const_cast<void*>(buf), count, dt, peer, comm, s);
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t s) {
  return post(1, buf, count, dt, peer, comm, s);
}

// naive collectives on top of the mailboxes (small host-helper payloads only)
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t dt, int root, ncclComm_t comm, hipStream_t s) {
  const size_t bytes = count * (dt == ncclFloat64 ? 8 : 1);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<Op> ops;
  if (comm->rank == root) {
    for (int p = 0; p < comm->nranks; ++p)
      if (p != root) ops.push_back(Op{0, (char*)send, bytes, 0, p, comm, s});
    if (recv != send && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
  } else {
    ops.push_back(Op{1, (char*)recv, bytes, 0, root, comm, s});
  }
  return progress(ops);
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  if (dt != ncclFloat64) return ncclInvalidArgument;
  const int P = comm->nranks;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<double> mine(count), acc(count);
  if (hipMemcpy(mine.data(), send, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  acc = mine;
  double* tmp = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&tmp), count * 8 * (size_t)P) != hipSuccess) return ncclUnhandledCudaError;
  std::vector<Op> ops;
  for (int p = 0; p < P; ++p) {
    if (p == comm->rank) continue;
    ops.push_back(Op{0, (char*)send, count * 8, 0, p, comm, s});
    ops.push_back(Op{1, (char*)(tmp + (size_t)p * count), count * 8, 0, p, comm, s});
  }
  ncclResult_t r = progress(ops);
  if (r != ncclSuccess) { (void)hipFree(tmp); return r; }
  std::vector<double> other(count);
  for (int p = 0; p < P; ++p) {
    if (p == comm->rank) continue;
    if (hipMemcpy(other.data(), tmp + (size_t)p * count, count * 8, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t i = 0; i < count; ++i) acc[i] = op == ncclMax ? (other[i] > acc[i] ? other[i] : acc[i]) : acc[i] + other[i];
  }
  (void)hipFree(tmp);
  if (hipMemcpy(recv, acc.data(), count * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // extern "C"
