// comm.hip -- Self / RCCL / in-process transports (see comm.h)
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <chrono>
#include <thread>
#include <vector>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include "comm.h"
#include "mfft_internal.h"

int mfft_comm_s::work_alloc(void** p, size_t bytes) { return mfft::dev_alloc(p, bytes); }
int mfft_comm_s::work_free(void* p) { return mfft::dev_free(p); }
int mfft_comm_s::set_option(const char* key, long long) { return mfft::set_error(MFFT_ERR_INVALID, "this transport has no option '%s'", key); }
// "host_collectives": 1 when barrier / bcast_host / allreduce_host stay on the host (shared memory, condition variables) and
// never touch a device stream -- every transport but RCCL, whose host collectives are a copy + ncclAllReduce + stream sync
long long mfft_comm_s::get_option(const char* key) { return key && !strcmp(key, "host_collectives") ? 1 : -1; }

int mfft_comm_s::selftest(size_t bytes_per_peer, int timeout_ms) {
  using namespace mfft;
  const size_t n = (bytes_per_peer + 15) / 16 * 16, total = n * (size_t)size;
  void *snd = nullptr, *rcv = nullptr;
  MFFT_TRY(work_alloc(&snd, total));
  if (int rc = work_alloc(&rcv, total)) { (void)work_free(snd); return rc; }
  hipStream_t st = nullptr;
  int rc = 0;
  auto fail = [&](int code, const char* what) { rc = set_error(code, "transport self-test: %s", what); };
  std::vector<unsigned char> h(total), back(total, 0);
  for (int p = 0; p < size; ++p)                       // chunk for peer p: bytes depend on (sender, receiver, offset)
    for (size_t i = 0; i < n; ++i) h[(size_t)p * n + i] = (unsigned char)(rank * 31 + p * 7 + i * 13 + (i >> 8));
  std::vector<size_t> cnt(size, n), dsp(size);
  std::vector<int> peers(size);
  for (int p = 0; p < size; ++p) { dsp[p] = (size_t)p * n; peers[p] = p; }
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) fail(MFFT_ERR_HIP, "no stream");
  if (!rc && hipMemcpyAsync(snd, h.data(), total, hipMemcpyHostToDevice, st) != hipSuccess) fail(MFFT_ERR_HIP, "upload failed");
  if (!rc && hipMemsetAsync(rcv, 0, total, st) != hipSuccess) fail(MFFT_ERR_HIP, "memset failed");
  if (!rc) rc = alltoallv(snd, cnt.data(), dsp.data(), rcv, cnt.data(), dsp.data(), peers.data(), size, st, 0);
  if (!rc) {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t q;
    while ((q = hipStreamQuery(st)) == hipErrorNotReady) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) {
        rescue();                                      // release what can be released, then report
        fail(MFFT_ERR_INTERNAL, "the exchange did not complete in time");
        break;
      }
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (!rc && q != hipSuccess) fail(MFFT_ERR_HIP, hipGetErrorString(q));
    (void)hipGetLastError();
  }
  if (st) (void)hipStreamSynchronize(st);
  if (!rc && hipMemcpy(back.data(), rcv, total, hipMemcpyDeviceToHost) != hipSuccess) fail(MFFT_ERR_HIP, "download failed");
  if (!rc)
    for (int p = 0; p < size && !rc; ++p)              // what peer p sent to me
      for (size_t i = 0; i < n; ++i)
        if (back[(size_t)p * n + i] != (unsigned char)(p * 31 + rank * 7 + i * 13 + (i >> 8))) {
          rc = set_error(MFFT_ERR_INTERNAL, "transport self-test: rank %d received wrong data from rank %d at byte %zu", rank, p, i);
          break;
        }
  if (st) (void)hipStreamDestroy(st);
  (void)work_free(snd);
  (void)work_free(rcv);
  if (rc) abort();                                     // peers blocked in a host-side wait of this transport fail fast
  return rc;
}

namespace mfft {

// ===========================================================================
// Self
// ===========================================================================
struct SelfComm : mfft_comm_s {
  int alltoallv(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
                const size_t* rdisp, const int*, int npeers, hipStream_t s, int) override {
    if (npeers != 1) return set_error(MFFT_ERR_INVALID, "self comm: group of %d", npeers);
    if (scount[0] != rcount[0]) return set_error(MFFT_ERR_INVALID, "self comm: count mismatch");
    if (scount[0])
      MFFT_HIP(hipMemcpyAsync(static_cast<char*>(recv) + rdisp[0], static_cast<const char*>(send) + sdisp[0],
                              scount[0], hipMemcpyDeviceToDevice, s));
    return 0;
  }
  int barrier() override { return 0; }
  int bcast_host(void*, size_t, int) override { return 0; }
  int allreduce_host(double*, int, int) override { return 0; }
};

int comm_create_self(mfft_comm_s** out) {
  *out = new SelfComm();
  return 0;
}

// ===========================================================================
// RCCL (dlopen'ed so that single-GPU use never touches the library and so that
// whichever librccl.so.1 the process already holds is the one that is used)
// ===========================================================================
struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  // what the library itself says about a communicator (optional: a library without them answers -1)
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

static RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    // MFFT_RCCL_LIB overrides the library (tests use it to put a shared-memory stand-in behind
    // the same entry points, because RCCL refuses two ranks on one GPU)
    const char* names[] = {getenv("MFFT_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
    }
    if (!api.handle) return;
#define MFFT_SYM(f) api.f = reinterpret_cast<decltype(api.f)>(dlsym(api.handle, "nccl" #f))
    MFFT_SYM(GetUniqueId);
    MFFT_SYM(CommInitRank);
    MFFT_SYM(CommDestroy);
    MFFT_SYM(Send);
    MFFT_SYM(Recv);
    MFFT_SYM(GroupStart);
    MFFT_SYM(GroupEnd);
    MFFT_SYM(Broadcast);
    MFFT_SYM(AllReduce);
    MFFT_SYM(GetErrorString);
    MFFT_SYM(CommCount);
    MFFT_SYM(CommUserRank);
    MFFT_SYM(CommCuDevice);
    MFFT_SYM(GetVersion);
#undef MFFT_SYM
    ok = api.GetUniqueId && api.CommInitRank && api.CommDestroy && api.Send && api.Recv && api.GroupStart &&
         api.GroupEnd && api.Broadcast && api.AllReduce && api.GetErrorString;
  });
  return ok ? &api : nullptr;
}

#define MFFT_NCCL(api, call)                                                                        \
  do {                                                                                              \
    ncclResult_t r_ = (call);                                                                       \
    if (r_ != ncclSuccess)                                                                          \
      return set_error(MFFT_ERR_RCCL, "%s failed: %s (%s:%d)", #call, (api)->GetErrorString(r_),    \
                       __FILE__, __LINE__);                                                         \
  } while (0)

struct RcclComm : mfft_comm_s {
  RcclApi* api = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t hstream = nullptr;     // for the host-buffer helpers
  void* small_d = nullptr;           // device scratch of the small host all-reduces: allocated once (hipMalloc / hipFree per
  static constexpr size_t SMALL = 4096;   // call would synchronise the whole device under every agreement vote)
  ~RcclComm() override {
    if (comm) api->CommDestroy(comm);
    if (small_d) (void)hipFree(small_d);
    if (hstream) (void)hipStreamDestroy(hstream);
  }
  int alltoallv(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
                const size_t* rdisp, const int* peers, int npeers, hipStream_t s, int) override {
    const char* sp = static_cast<const char*>(send);
    char* rp = static_cast<char*>(recv);
    // the self chunk never enters RCCL
    for (int i = 0; i < npeers; ++i)
      if (peers[i] == rank && scount[i])
        MFFT_HIP(hipMemcpyAsync(rp + rdisp[i], sp + sdisp[i], scount[i], hipMemcpyDeviceToDevice, s));
    if (npeers > 1) {
      MFFT_NCCL(api, api->GroupStart());
      for (int i = 0; i < npeers; ++i) {
        if (peers[i] == rank) continue;
        if (scount[i]) MFFT_NCCL(api, api->Send(sp + sdisp[i], scount[i], ncclUint8, peers[i], comm, s));
        if (rcount[i]) MFFT_NCCL(api, api->Recv(rp + rdisp[i], rcount[i], ncclUint8, peers[i], comm, s));
      }
      MFFT_NCCL(api, api->GroupEnd());
    }
    return 0;
  }
  // "rccl_nranks" / "rccl_rank" / "rccl_device" / "rccl_version": what RCCL ITSELF reports for this communicator
  // (ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion) -- so that a bench line can prove which library
  // built a communicator over how many ranks (slab.py:77-81 asks MPI the same: comm.Get_size(), comm.Get_rank())
  long long get_option(const char* key) override {
    if (!key) return -1;
    if (!strcmp(key, "host_collectives")) return 0;
    int v = -1;
    if (!strcmp(key, "rccl_nranks")) return (api->CommCount && api->CommCount(comm, &v) == ncclSuccess) ? v : -1;
    if (!strcmp(key, "rccl_rank")) return (api->CommUserRank && api->CommUserRank(comm, &v) == ncclSuccess) ? v : -1;
    if (!strcmp(key, "rccl_device")) return (api->CommCuDevice && api->CommCuDevice(comm, &v) == ncclSuccess) ? v : -1;
    if (!strcmp(key, "rccl_version")) return (api->GetVersion && api->GetVersion(&v) == ncclSuccess) ? v : -1;
    return mfft_comm_s::get_option(key);
  }
  int barrier() override {
    double v = 0;
    return allreduce_host(&v, 1, 0);
  }
  int bcast_host(void* buf, size_t bytes, int root) override {
    if (bytes == 0) return 0;
    void* d = nullptr;
    MFFT_HIP(hipMalloc(&d, bytes));
    if (rank == root) MFFT_HIP(hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, hstream));
    ncclResult_t r = api->Broadcast(d, d, bytes, ncclUint8, root, comm, hstream);
    if (r != ncclSuccess) {
      (void)hipFree(d);
      return set_error(MFFT_ERR_RCCL, "ncclBroadcast failed: %s", api->GetErrorString(r));
    }
    MFFT_HIP(hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, hstream));
    MFFT_HIP(hipStreamSynchronize(hstream));
    MFFT_HIP(hipFree(d));
    return 0;
  }
  int allreduce_host(double* vals, int count, int op) override {
    if (count <= 0) return 0;
    void* d = nullptr;
    const size_t bytes = sizeof(double) * (size_t)count;
    const bool small = bytes <= SMALL;
    if (small) {
      if (!small_d) MFFT_HIP(hipMalloc(&small_d, SMALL));
      d = small_d;
    } else {
      MFFT_HIP(hipMalloc(&d, bytes));
    }
    MFFT_HIP(hipMemcpyAsync(d, vals, bytes, hipMemcpyHostToDevice, hstream));
    ncclResult_t r = api->AllReduce(d, d, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, comm, hstream);
    if (r != ncclSuccess) {
      if (!small) (void)hipFree(d);
      return set_error(MFFT_ERR_RCCL, "ncclAllReduce failed: %s", api->GetErrorString(r));
    }
    MFFT_HIP(hipMemcpyAsync(vals, d, bytes, hipMemcpyDeviceToHost, hstream));
    MFFT_HIP(hipStreamSynchronize(hstream));
    if (!small) MFFT_HIP(hipFree(d));
    return 0;
  }
};

int comm_get_unique_id(void* id128) {
  const char* tr = getenv("MFFT_TRANSPORT");
  if (tr && strcmp(tr, "ipc") == 0) return ipc_make_unique_id(id128);     // the id itself says which transport the group uses
  if (tr && *tr && strcmp(tr, "rccl") != 0) return set_error(MFFT_ERR_INVALID, "MFFT_TRANSPORT=%s: expected rccl or ipc", tr);
  RcclApi* api = rccl_api();
  if (!api) return set_error(MFFT_ERR_RCCL, "librccl.so.1 could not be loaded: %s", dlerror());
  ncclUniqueId id;
  MFFT_NCCL(api, api->GetUniqueId(&id));
  static_assert(sizeof(id) == MFFT_UNIQUE_ID_BYTES, "unique id size");
  memcpy(id128, &id, sizeof id);
  return 0;
}

int comm_create_rccl(int nranks, int rank, const void* id128, mfft_comm_s** out) {
  if (ipc_is_unique_id(id128)) return comm_create_ipc(nranks, rank, id128, out);
  RcclApi* api = rccl_api();
  if (!api) return set_error(MFFT_ERR_RCCL, "librccl.so.1 could not be loaded: %s", dlerror());
  if (nranks < 1 || rank < 0 || rank >= nranks) return set_error(MFFT_ERR_INVALID, "bad rank %d of %d", rank, nranks);
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  std::unique_ptr<RcclComm> c(new RcclComm());
  c->api = api;
  c->size = nranks;
  c->rank = rank;
  MFFT_NCCL(api, api->CommInitRank(&c->comm, nranks, id, rank));
  MFFT_HIP(hipStreamCreateWithFlags(&c->hstream, hipStreamNonBlocking));
  *out = c.release();
  return 0;
}

// ===========================================================================
// in-process group of virtual ranks (host thread per rank)
// ===========================================================================
struct LocalShared {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  long generation = 0;
  int refs = 0;
  struct Post {
    const char* send = nullptr;
    std::vector<size_t> sdisp, scount;
    hipEvent_t ready = nullptr, done = nullptr;
    int device = 0;
    void* host_ptr = nullptr;
  };
  std::vector<Post> posts;

  // host barrier of the group's threads; false = a peer never arrived (it failed or
  // deadlocked): every later barrier of the group then fails fast instead of hanging
  bool broken = false;
  bool wait_all() {
    static const long timeout_s = getenv("MFFT_LOCAL_TIMEOUT") ? atol(getenv("MFFT_LOCAL_TIMEOUT")) : 180;
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const long gen = generation;
    if (++arrived == n) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    const bool ok = cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return generation != gen || broken; });
    if (!ok || broken) {
      broken = true;
      cv.notify_all();
      return false;
    }
    return true;
  }
};

#define MFFT_LOCAL_WAIT()                                                                             \
  do {                                                                                                \
    if (!sh->wait_all())                                                                              \
      return set_error(MFFT_ERR_INTERNAL, "in-process group: a peer rank never reached the barrier "  \
                                          "(it failed, or the ranks diverged)");                      \
  } while (0)

struct LocalComm : mfft_comm_s {
  std::shared_ptr<LocalShared> sh;
  ~LocalComm() override {
    LocalShared::Post& p = sh->posts[rank];
    if (p.ready) (void)hipEventDestroy(p.ready);
    if (p.done) (void)hipEventDestroy(p.done);
    p.ready = p.done = nullptr;
  }
  int ensure_events() {
    LocalShared::Post& p = sh->posts[rank];
    if (!p.ready) {
      MFFT_HIP(hipEventCreateWithFlags(&p.ready, hipEventDisableTiming));
      MFFT_HIP(hipEventCreateWithFlags(&p.done, hipEventDisableTiming));
    }
    return 0;
  }
  int alltoallv(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
                const size_t* rdisp, const int* peers, int npeers, hipStream_t s, int) override {
    MFFT_TRY(ensure_events());
    LocalShared::Post& me = sh->posts[rank];
    int myidx = -1;
    for (int i = 0; i < npeers; ++i)
      if (peers[i] == rank) myidx = i;
    if (myidx < 0) return set_error(MFFT_ERR_INVALID, "local comm: rank %d not in its own group", rank);
    me.send = static_cast<const char*>(send);
    me.sdisp.assign(sdisp, sdisp + npeers);
    me.scount.assign(scount, scount + npeers);
    MFFT_HIP(hipEventRecord(me.ready, s));
    MFFT_LOCAL_WAIT();                                // every member has posted
    char* rp = static_cast<char*>(recv);
    for (int i = 0; i < npeers; ++i) {
      LocalShared::Post& pe = sh->posts[peers[i]];
      if (pe.scount[myidx] != rcount[i])
        return set_error(MFFT_ERR_INTERNAL, "local comm: rank %d expects %zu bytes from %d, peer sends %zu", rank,
                         rcount[i], peers[i], pe.scount[myidx]);
      if (!rcount[i]) continue;
      if (peers[i] != rank) MFFT_HIP(hipStreamWaitEvent(s, pe.ready, 0));
      MFFT_HIP(hipMemcpyAsync(rp + rdisp[i], pe.send + pe.sdisp[myidx], rcount[i], hipMemcpyDeviceToDevice, s));
    }
    MFFT_HIP(hipEventRecord(me.done, s));
    MFFT_LOCAL_WAIT();                                // every member has enqueued its pulls
    // nobody may overwrite its send buffer before all peers have pulled from it
    for (int i = 0; i < npeers; ++i)
      if (peers[i] != rank) MFFT_HIP(hipStreamWaitEvent(s, sh->posts[peers[i]].done, 0));
    MFFT_LOCAL_WAIT();                                // events may be re-recorded only after everyone waited on them
    return 0;
  }
  void abort() override {
    std::lock_guard<std::mutex> lk(sh->mu);
    sh->broken = true;
    sh->cv.notify_all();
  }
  int barrier() override {
    MFFT_LOCAL_WAIT();
    return 0;
  }
  int bcast_host(void* buf, size_t bytes, int root) override {
    sh->posts[rank].host_ptr = buf;
    MFFT_LOCAL_WAIT();
    if (rank != root && bytes) memcpy(buf, sh->posts[root].host_ptr, bytes);
    MFFT_LOCAL_WAIT();
    return 0;
  }
  int allreduce_host(double* vals, int count, int op) override {
    sh->posts[rank].host_ptr = vals;
    MFFT_LOCAL_WAIT();
    std::vector<double> acc(vals, vals + count);
    for (int r = 0; r < size; ++r) {
      if (r == rank) continue;
      const double* o = static_cast<const double*>(sh->posts[r].host_ptr);
      for (int i = 0; i < count; ++i) acc[i] = op == 1 ? (o[i] > acc[i] ? o[i] : acc[i]) : acc[i] + o[i];
    }
    MFFT_LOCAL_WAIT();
    for (int i = 0; i < count; ++i) vals[i] = acc[i];
    MFFT_LOCAL_WAIT();
    return 0;
  }
};

int comm_create_local(int nranks, const int* devices, mfft_comm_s** out) {
  if (nranks < 1) return set_error(MFFT_ERR_INVALID, "nranks must be >= 1");
  int cur = 0;
  MFFT_HIP(hipGetDevice(&cur));
  auto sh = std::make_shared<LocalShared>();
  sh->n = nranks;
  sh->posts.resize(nranks);
  for (int r = 0; r < nranks; ++r) sh->posts[r].device = devices ? devices[r] : cur;
  // peer access between distinct devices (ignore "already enabled")
  for (int a = 0; a < nranks; ++a)
    for (int b = 0; b < nranks; ++b) {
      const int da = sh->posts[a].device, db = sh->posts[b].device;
      if (da == db) continue;
      int can = 0;
      (void)hipDeviceCanAccessPeer(&can, da, db);
      if (can) {
        (void)hipSetDevice(da);
        hipError_t e = hipDeviceEnablePeerAccess(db, 0);
        if (e != hipSuccess) (void)hipGetLastError();
      }
    }
  (void)hipSetDevice(cur);
  for (int r = 0; r < nranks; ++r) {
    LocalComm* c = new LocalComm();
    c->size = nranks;
    c->rank = r;
    c->sh = sh;
    out[r] = c;
  }
  return 0;
}

}  // namespace mfft
