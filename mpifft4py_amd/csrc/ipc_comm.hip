// ipc_comm.hip -- IpcComm: one process per GPU on ONE node; every rank PULLS its chunks out of the peers' send
// buffers through IPC-mapped device memory.  Three ways to pull (MFFT_IPC_PULL, mfft_comm_set_option "ipc_pull"):
//   kernel  (default) ONE launch of ipc_pull_kernel (ipc_pull.h) reads from all peers' mappings at once, a few
//           workgroups per peer: every xGMI link of the fully connected node carries data at the same time; the launch
//           runs on the plan's communication stream, which plan.hip confines to a handful of CUs (MFFT_COMM_CUS);
//   streams one hipMemcpyAsync per peer, each on a stream of its own (fork / join events around them): the copy
//           engines of several links at once, no CU at all;
//   copy    round 2's path: the hipMemcpyAsync's one after the other on the issuing stream (one link at a time).
// The flag protocol below is the same for all three.
//
// Replaces, like RcclComm, the mpi4py collectives of the reference (comm.Alltoall slab.py:406/281, Alltoallw
// pencil.py:741-750, 1324-1333).  Why a second transport: RCCL's send/recv are kernels, and the FFT kernels keep every
// CU busy (two 1024-thread workgroups with 80 KB of LDS each), so RCCL's copy kernels queue behind them.  It also runs
// between processes that share ONE GPU, which RCCL refuses, so the process-per-GPU path is tested for real (no mock)
// on a single-GPU box.
//
// Design, each point decided by a measurement on this pool (tools/ipc_probe.hip, tests/test_gpu_multiprocess.py):
//  * Only plan work buffers are ever the SOURCE of an exchange (plan.hip), and they come from an arena this
//    communicator owns (work_alloc): segments of device memory that are exported once, when they are created, and
//    live as long as the communicator.  Exporting arbitrary buffers on demand does not survive long runs: after a
//    process has closed imported mappings (of buffers their owners had freed), hipIpcGetMemHandle of a NEW allocation
//    fails with "invalid argument" (ROCm 7.2, 4 and 8 processes).  So nothing is exported late, nothing imported is
//    closed before the communicator goes, and user arrays are never exported at all (they are only ever pulled INTO).
//  * Cross-process ordering: 32-bit sequence numbers in device memory, written into the PEER's flag array with
//    hipStreamWriteValue32 and awaited locally with hipStreamWaitValue32(>=): "ready" (sender -> receiver: my send
//    buffer holds exchange q) and "done" (receiver -> sender: I have pulled my chunk of exchange q).  Everything is
//    enqueued; the only host-side wait is for the peer's post (where its chunk lies), a few microseconds behind.
//    Interprocess HIP events would be the textbook tool, and they do work here -- for exactly 32 records: the 33rd
//    hipStreamWaitEvent on an opened interprocess event returns "invalid argument" (ROCm 7.2, ipc_probe2.hip).
//    A first version PUSHED chunks into the peers' receive buffers under the same flags: one rank in four then read
//    stale data after the wait (a remote write behind the reader's L2); a pull is an ordinary local copy command
//    of the reader's own stream, and the sender's data is released by an event record before its flag is written.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <future>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>
#include "comm.h"
#include "ipc_pull.h"
#include "relay_plan.h"
#include "mfft_internal.h"

namespace mfft {

// plain device buffers (everything that is not a plan work buffer of an IpcComm)
int dev_alloc(void** p, size_t bytes) {
  *p = nullptr;
  hipError_t e = hipMalloc(p, bytes ? bytes : 16);
  if (e != hipSuccess) return set_error(MFFT_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
  return 0;
}
int dev_free(void* p) {
  if (p) MFFT_HIP(hipFree(p));
  return 0;
}

namespace {

constexpr int IPC_MAX_RANKS = 16, IPC_MAX_CH = 2, IPC_MAX_SEG = 64, IPC_SCRATCH = 4096;
constexpr uint64_t IPC_MAGIC = 0x4D46465449504331ull;   // "MFFTIPC1"

struct IpcSegment {
  std::atomic<uint32_t> live;
  uint32_t pad;
  uint64_t base, size;
  hipIpcMemHandle_t h;
};
constexpr int IPC_RING = 4;
struct IpcPost {                    // where the receiver finds its chunk of exchange `seq` in the sender's memory
  std::atomic<uint64_t> seq;
  uint32_t seg, pad;
  uint64_t offset, bytes;
};
struct IpcPair {                    // one per ordered pair (sender, receiver) and channel
  IpcPost ring[IPC_RING];
  std::atomic<uint64_t> consumed;   // last post the receiver has read
};
// relayed sub-group exchanges (relay_plan.h): all ranks take part in lock-step, two host barriers per exchange, so a
// ring of two slots (exchange number & 1) is enough
struct IpcRPost {                   // message s -> d of relayed exchange q
  uint32_t seg, pad;
  uint64_t offset, bytes;
};
struct IpcStage {                   // where relay r has staged its stripe of the message s -> d
  uint32_t seg, pad;
  uint64_t offset;
};
struct IpcRankInfo {
  int device, pid;
  hipIpcMemHandle_t flags_h;
  IpcSegment seg[IPC_MAX_SEG];
  alignas(64) char scratch[IPC_SCRATCH];
};
struct IpcShm {
  std::atomic<uint64_t> magic;
  int nranks;
  std::atomic<uint32_t> attached, detached;
  std::atomic<uint32_t> bar_count, bar_gen, broken;
  IpcRankInfo rk[IPC_MAX_RANKS];
  IpcPair pair[IPC_MAX_RANKS /*sender*/][IPC_MAX_RANKS /*receiver*/][IPC_MAX_CH];
  IpcRPost rpost[IPC_MAX_RANKS /*s*/][IPC_MAX_RANKS /*d*/][IPC_MAX_CH][2];
  IpcStage stage[IPC_MAX_RANKS /*relay*/][IPC_MAX_RANKS /*s*/][IPC_MAX_RANKS /*d*/][IPC_MAX_CH][2];
};
// device-side flags of one rank (the READER of a flag owns it, so that polling stays local)
struct IpcFlags {
  uint32_t ready[IPC_MAX_RANKS][IPC_MAX_CH];   // ready[p]: written by sender p, "my send buffer holds exchange q"
  uint32_t done[IPC_MAX_RANKS][IPC_MAX_CH];    // done[r]:  written by receiver r, "I have pulled my chunk of exchange q"
  // relayed exchanges, all written by rank x about its relayed exchange number q:
  uint32_t rready[IPC_MAX_RANKS][IPC_MAX_CH];  //   "my send buffer holds exchange q"
  uint32_t k1done[IPC_MAX_RANKS][IPC_MAX_CH];  //   "my phase 1 is done": first hops staged in my memory, my first reads of your buffer over
  uint32_t k2done[IPC_MAX_RANKS][IPC_MAX_CH];  //   "my phase 2 is done": I read nothing of exchange q any more
};
static_assert(std::atomic<uint64_t>::is_always_lock_free && std::atomic<uint32_t>::is_always_lock_free, "shared-memory atomics");

long ipc_timeout_s() {
  static const long t = getenv("MFFT_LOCAL_TIMEOUT") ? atol(getenv("MFFT_LOCAL_TIMEOUT")) : 180;
  return t;
}

struct IpcComm : mfft_comm_s {
  IpcShm* sh = nullptr;
  std::string shm_name;
  bool creator = false;
  int device = 0;
  IpcFlags* flags = nullptr;                               // mine (device memory, exported at creation)
  bool flags_fine = false;                                 // allocated fine-grained
  std::vector<IpcFlags*> peer_flags;                       // IPC mappings of the peers' flags
  hipEvent_t release_ev[IPC_MAX_CH] = {};                  // plain events: a record releases kernel-written data
  uint32_t sseq[IPC_MAX_RANKS][IPC_MAX_CH] = {}, rseq[IPC_MAX_RANKS][IPC_MAX_CH] = {};
  std::map<std::pair<int, uint32_t>, void*> maps;          // (peer, segment) -> mapping, closed with the communicator
  // arena: first-fit over the segments, 2 MiB granules
  struct Block { uint64_t off, size; bool used; };
  struct Seg { char* base; uint64_t size; std::vector<Block> blocks; };
  std::vector<Seg> segs;
  hipStream_t last_stream[IPC_MAX_CH] = {};
  hipEvent_t last_issue[IPC_MAX_CH] = {};
  bool used_ch[IPC_MAX_CH] = {};
  // how the receiver role moves the bytes (see the head of this file)
  enum { PULL_COPY = 0, PULL_KERNEL = 1, PULL_STREAMS = 2 };
  int pull_mode = PULL_KERNEL;
  int pull_wgs = 8;                                        // kernel mode: workgroups per peer chunk
  hipStream_t pstream[IPC_MAX_RANKS] = {};                 // streams mode: one copy stream per peer (normal priority)
  // relayed sub-group exchanges (exchange_relay)
  int relay_mode = 0;                                      // 1: sub-group exchanges are relay-striped (MFFT_IPC_RELAY, "ipc_relay")
  uint32_t xseq[IPC_MAX_CH] = {};
  void* staging[IPC_MAX_CH] = {};
  size_t staging_bytes[IPC_MAX_CH] = {};
  hipEvent_t fork_ev[IPC_MAX_CH] = {}, join_ev[IPC_MAX_RANKS][IPC_MAX_CH] = {};
  hipStream_t rescue_stream = nullptr;                     // rescue(): non-blocking, never used for anything else

  IpcComm() {
    if (const char* e = getenv("MFFT_IPC_PULL")) {
      if (!strcmp(e, "copy")) pull_mode = PULL_COPY;
      else if (!strcmp(e, "streams")) pull_mode = PULL_STREAMS;
      else pull_mode = PULL_KERNEL;
    }
    if (const char* e = getenv("MFFT_IPC_PULL_WGS")) pull_wgs = std::max(1, std::min(64, atoi(e)));
    if (const char* e = getenv("MFFT_IPC_RELAY")) relay_mode = atoi(e) != 0 ? 1 : 0;
  }
  int set_option(const char* key, long long v) override {
    if (!strcmp(key, "ipc_pull")) {
      if (v < 0 || v > 2) return set_error(MFFT_ERR_INVALID, "ipc_pull: 0 copy, 1 kernel, 2 streams");
      pull_mode = (int)v;
      return 0;
    }
    if (!strcmp(key, "ipc_relay")) {        // COLLECTIVE: the relayed exchange is another protocol (host barriers, other
      if (v < 0 || v > 1) return set_error(MFFT_ERR_INVALID, "ipc_relay: 0 off, 1 on");     // flag words), so the ranks must agree
      relay_mode = (int)v;
      return agree_on_relay();
    }
    if (!strcmp(key, "ipc_pull_wgs")) {
      if (v < 1 || v > 64) return set_error(MFFT_ERR_INVALID, "ipc_pull_wgs: 1 .. 64 workgroups per peer");
      pull_wgs = (int)v;
      return 0;
    }
    return mfft_comm_s::set_option(key, v);
  }
  long long get_option(const char* key) override {
    if (!strcmp(key, "ipc_pull")) return pull_mode;
    if (!strcmp(key, "ipc_pull_wgs")) return pull_wgs;
    if (!strcmp(key, "ipc_relay")) return relay_enabled() ? 1 : 0;
    return mfft_comm_s::get_option(key);
  }

  ~IpcComm() override {
    // a broken group never satisfies the waits that are still enqueued: release mine before waiting for the device
    if (sh && sh->broken.load()) rescue();
    (void)quiesce();                          // bounded: a peer that vanished cannot hold this process for ever
    if (sh) {
      // peers may still pull from my segments or wait on my events: leave together (best effort, bounded)
      sh->detached.fetch_add(1);
      const auto t0 = std::chrono::steady_clock::now();
      while (sh->detached.load() < (uint32_t)size && !sh->broken.load() &&
             std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20))
        std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    for (auto& m : maps) (void)hipIpcCloseMemHandle(m.second);
    for (int r = 0; r < (int)peer_flags.size(); ++r)
      if (peer_flags[r] && r != rank) (void)hipIpcCloseMemHandle(peer_flags[r]);
    for (int ch = 0; ch < IPC_MAX_CH; ++ch) {
      if (release_ev[ch]) (void)hipEventDestroy(release_ev[ch]);
      if (last_issue[ch]) (void)hipEventDestroy(last_issue[ch]);
      if (fork_ev[ch]) (void)hipEventDestroy(fork_ev[ch]);
      for (int r = 0; r < IPC_MAX_RANKS; ++r)
        if (join_ev[r][ch]) (void)hipEventDestroy(join_ev[r][ch]);
    }
    for (hipStream_t ps : pstream)
      if (ps) (void)hipStreamDestroy(ps);
    if (rescue_stream) (void)hipStreamDestroy(rescue_stream);
    // (staging blocks live in the arena's segments, freed below)
    if (flags) (void)hipFree(flags);
    for (Seg& s : segs) (void)hipFree(s.base);
    if (sh) munmap(sh, sizeof(IpcShm));
    if (creator && !shm_name.empty()) (void)shm_unlink(shm_name.c_str());
  }

  template <class Pred>
  int spin(Pred ok, const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    while (!ok()) {
      if (sh->broken.load()) return set_error(MFFT_ERR_INTERNAL, "ipc transport: a peer rank failed (%s)", what);
      if (++n > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
      if ((n & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(ipc_timeout_s())) {
        sh->broken.store(1);
        return set_error(MFFT_ERR_INTERNAL, "ipc transport: timed out after %ld s waiting for %s", ipc_timeout_s(), what);
      }
    }
    return 0;
  }

  int barrier() override {
    const uint32_t gen = sh->bar_gen.load();
    if (sh->bar_count.fetch_add(1) + 1 == (uint32_t)size) {
      sh->bar_count.store(0);
      sh->bar_gen.fetch_add(1);
      return 0;
    }
    return spin([&] { return sh->bar_gen.load() != gen; }, "a host barrier");
  }
  void abort() override {
    if (sh) sh->broken.store(1);
  }
  // release every device-side wait of THIS rank (the flags are in my own memory): a hung exchange then runs to its
  // end with whatever data there is, and the communicator is marked broken
  // The fill runs on a stream of its own created hipStreamNonBlocking: a null-stream memset would queue behind every
  // BLOCKING stream of the device -- the CU-masked plan streams are such streams (hipExtStreamCreateWithCUMask has no
  // flags argument) -- i.e. behind the very wait kernel it is meant to release.
  void rescue() override {
    if (sh) sh->broken.store(1);
    if (!flags) return;
    if (!rescue_stream && hipStreamCreateWithFlags(&rescue_stream, hipStreamNonBlocking) != hipSuccess) {
      (void)hipGetLastError();
      rescue_stream = nullptr;
      return;
    }
    // 0x7FFFFFFF satisfies every ">= q" wait, cyclic comparison or not (q stays far below 2^31)
    if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(flags), 0x7FFFFFFF, sizeof(IpcFlags) / 4, rescue_stream) == hipSuccess)
      (void)hipStreamSynchronize(rescue_stream);
    else
      (void)hipGetLastError();
  }
  int bcast_host(void* buf, size_t bytes, int root) override {
    char* p = static_cast<char*>(buf);
    for (size_t off = 0; off < bytes; off += IPC_SCRATCH) {
      const size_t n = std::min<size_t>(IPC_SCRATCH, bytes - off);
      if (rank == root) memcpy(sh->rk[root].scratch, p + off, n);
      MFFT_TRY(barrier());
      if (rank != root) memcpy(p + off, sh->rk[root].scratch, n);
      MFFT_TRY(barrier());
    }
    return 0;
  }
  int allreduce_host(double* vals, int count, int op) override {
    const int per = IPC_SCRATCH / (int)sizeof(double);
    for (int off = 0; off < count; off += per) {
      const int n = std::min(per, count - off);
      memcpy(sh->rk[rank].scratch, vals + off, n * sizeof(double));
      MFFT_TRY(barrier());
      std::vector<double> acc(n);
      for (int i = 0; i < n; ++i) acc[i] = reinterpret_cast<double*>(sh->rk[0].scratch)[i];
      for (int r = 1; r < size; ++r) {
        const double* o = reinterpret_cast<const double*>(sh->rk[r].scratch);
        for (int i = 0; i < n; ++i) acc[i] = op == 1 ? (o[i] > acc[i] ? o[i] : acc[i]) : acc[i] + o[i];
      }
      MFFT_TRY(barrier());
      memcpy(vals + off, acc.data(), n * sizeof(double));
    }
    return 0;
  }

  // ---- arena of exportable work buffers -------------------------------------------------------------------------
  static constexpr uint64_t GRAN = (uint64_t)2 << 20;
  int work_alloc(void** p, size_t bytes) override {
    const uint64_t want = ((bytes ? bytes : 16) + GRAN - 1) / GRAN * GRAN;
    for (Seg& s : segs)
      for (size_t i = 0; i < s.blocks.size(); ++i) {
        Block& b = s.blocks[i];
        if (b.used || b.size < want) continue;
        if (b.size > want) {
          const Block rest{b.off + want, b.size - want, false};
          b.size = want;
          s.blocks.insert(s.blocks.begin() + i + 1, rest);
        }
        s.blocks[i].used = true;
        *p = s.base + s.blocks[i].off;
        return 0;
      }
    // a new segment: the request itself (large plans) or 64 MiB that small plans share
    if ((int)segs.size() >= IPC_MAX_SEG) return set_error(MFFT_ERR_NOMEM, "ipc transport: more than %d work segments", IPC_MAX_SEG);
    const uint64_t seg_bytes = std::max<uint64_t>(want, (uint64_t)64 << 20);
    void* base = nullptr;
    hipError_t e = hipMalloc(&base, seg_bytes);
    if (e != hipSuccess) return set_error(MFFT_ERR_NOMEM, "hipMalloc(%llu) for a work segment failed: %s", (unsigned long long)seg_bytes, hipGetErrorString(e));
    IpcSegment& pub = sh->rk[rank].seg[segs.size()];
    e = hipIpcGetMemHandle(&pub.h, base);
    if (e != hipSuccess) {
      (void)hipFree(base);
      return set_error(MFFT_ERR_HIP, "ipc transport: hipIpcGetMemHandle of a %llu-byte segment failed: %s", (unsigned long long)seg_bytes, hipGetErrorString(e));
    }
    pub.base = (uint64_t)(uintptr_t)base;
    pub.size = seg_bytes;
    pub.live.store(1, std::memory_order_release);
    Seg s{static_cast<char*>(base), seg_bytes, {}};
    s.blocks.push_back(Block{0, want, true});
    if (seg_bytes > want) s.blocks.push_back(Block{want, seg_bytes - want, false});
    segs.push_back(std::move(s));
    *p = base;
    return 0;
  }
  // Wait until nothing enqueued so far can touch a work block any more: the last exchange issued on every channel
  // (its event also covers the transform kernels queued before it on that stream) and, because a plan's kernels may run
  // on streams this object has never seen, the rest of the device -- but BOUNDED: a peer that died without marking the
  // group broken (SIGKILL) leaves a wait kernel spinning for ever, and hipDeviceSynchronize would block the host with
  // it.  After the transport's timeout the waits are released (rescue) and the group is broken.
  int quiesce() {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      MFFT_HIP(hipDeviceSynchronize());
      return 0;
    }
    // a record on the legacy null stream completes when all prior work of every BLOCKING stream has; the non-blocking
    // plan streams are covered by the channels' last_issue events
    hipError_t e = hipEventRecord(ev, nullptr);
    const auto t0 = std::chrono::steady_clock::now();
    bool released = false, stuck = false;
    auto pending = [&]() {
      if (e == hipSuccess && hipEventQuery(ev) == hipErrorNotReady) return true;
      for (int ch = 0; ch < IPC_MAX_CH; ++ch)
        if (used_ch[ch] && last_issue[ch] && hipEventQuery(last_issue[ch]) == hipErrorNotReady) return true;
      return false;
    };
    int n = 0;
    while (pending()) {
      if (++n > 200) std::this_thread::sleep_for(std::chrono::microseconds(100));
      if (!released && ((sh && sh->broken.load()) ||
                        std::chrono::steady_clock::now() - t0 > std::chrono::seconds(ipc_timeout_s()))) {
        rescue();
        released = true;
      }
      if (released && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(ipc_timeout_s() + 30)) {
        stuck = true;
        break;
      }
    }
    (void)hipGetLastError();
    (void)hipEventDestroy(ev);
    if (stuck)      // the caller must not reuse what in-flight work may still touch (work_free keeps the block)
      return set_error(MFFT_ERR_HIP, "ipc transport: work of a broken group is still pending %d s after its waits were released", (int)(ipc_timeout_s() + 30));
    if (!released) {
      MFFT_HIP(hipDeviceSynchronize());      // nothing of mine is waiting on a peer: cheap, and covers every stream
      return 0;
    }
    // Released path: the events above cover the exchanges and what was queued BEFORE them; a transform kernel queued after
    // the last exchange on a plan's non-blocking compute stream (the out-of-place x pass reads the work buffers) is covered
    // by neither.  Every wait of mine has been released, so the device drains -- wait for that, bounded all the same.
    if (!bounded_device_sync(30))
      return set_error(MFFT_ERR_HIP, "ipc transport: the device did not drain within 30 s after the waits of a broken group were released");
    return 0;
  }
  // hipDeviceSynchronize with a time limit: run by a helper thread (detached if it never returns)
  static bool bounded_device_sync(int seconds) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    auto done = std::make_shared<std::promise<bool>>();
    std::future<bool> fut = done->get_future();
    std::thread([done, dev] {
      bool ok = hipSetDevice(dev) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
      (void)hipGetLastError();
      done->set_value(ok);
    }).detach();
    return fut.wait_for(std::chrono::seconds(seconds)) == std::future_status::ready && fut.get();
  }
  int work_free(void* p) override {
    if (!p) return 0;
    // As safe as the hipFree it stands in for: the block may be handed to another plan (another stream) at once, while
    // transforms of the plan that owned it are still in flight and peers still pull from it (their "done" flags are
    // awaited in-stream only).  Once this device is idle every such wait of mine has been satisfied.  Growth is rare.
    // If the device cannot be shown idle (quiesce fails: a broken group whose work never drains) the block is NOT returned
    // to the arena -- it leaks, which nothing can reuse under in-flight kernels -- and the error goes to the caller.
    if (sh && sh->broken.load()) rescue();
    MFFT_TRY(quiesce());
    for (Seg& s : segs) {
      if (static_cast<char*>(p) < s.base || static_cast<char*>(p) >= s.base + s.size) continue;
      const uint64_t off = (uint64_t)(static_cast<char*>(p) - s.base);
      for (size_t i = 0; i < s.blocks.size(); ++i) {
        if (s.blocks[i].off != off || !s.blocks[i].used) continue;
        s.blocks[i].used = false;
        if (i + 1 < s.blocks.size() && !s.blocks[i + 1].used) {
          s.blocks[i].size += s.blocks[i + 1].size;
          s.blocks.erase(s.blocks.begin() + i + 1);
        }
        if (i > 0 && !s.blocks[i - 1].used) {
          s.blocks[i - 1].size += s.blocks[i].size;
          s.blocks.erase(s.blocks.begin() + i);
        }
        return 0;
      }
    }
    return set_error(MFFT_ERR_INVALID, "ipc transport: %p is not a work buffer of this communicator", p);
  }
  int locate(const void* p, uint32_t* seg, uint64_t* off) const {
    for (size_t i = 0; i < segs.size(); ++i)
      if (static_cast<const char*>(p) >= segs[i].base && static_cast<const char*>(p) < segs[i].base + segs[i].size) {
        *seg = (uint32_t)i;
        *off = (uint64_t)(static_cast<const char*>(p) - segs[i].base);
        return 0;
      }
    return set_error(MFFT_ERR_INVALID, "ipc transport: the send buffer %p of an exchange is not a work buffer of this communicator", p);
  }
  int remote_base(int peer, uint32_t seg, char** out) {
    if (seg >= (uint32_t)IPC_MAX_SEG || !sh->rk[peer].seg[seg].live.load(std::memory_order_acquire))
      return set_error(MFFT_ERR_INTERNAL, "ipc transport: rank %d offers data in a segment it never published", peer);
    auto key = std::make_pair(peer, seg);
    auto it = maps.find(key);
    if (it == maps.end()) {
      void* ptr = nullptr;
      MFFT_HIP(hipIpcOpenMemHandle(&ptr, sh->rk[peer].seg[seg].h, hipIpcMemLazyEnablePeerAccess));
      it = maps.emplace(key, ptr).first;
    }
    *out = static_cast<char*>(it->second);
    return 0;
  }

  int alltoallv(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
                const size_t* rdisp, const int* peers, int npeers, hipStream_t s, int channel) override {
    const int rc = exchange(send, scount, sdisp, recv, rcount, rdisp, peers, npeers, s, channel);
    // the peers are (or will be) waiting in a host barrier: let them fail fast; and whatever this rank has already
    // enqueued behind a flag that will never come must not block the device for good (plan_sync, memcpy helpers)
    if (rc != 0) rescue();
    return rc;
  }
  int peer_stream(int p, hipStream_t* out) {
    if (!pstream[p]) MFFT_HIP(hipStreamCreateWithFlags(&pstream[p], hipStreamNonBlocking));
    *out = pstream[p];
    return 0;
  }
  int exchange(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
               const size_t* rdisp, const int* peers, int npeers, hipStream_t s, int ch) {
    if (ch < 0 || ch >= IPC_MAX_CH) return set_error(MFFT_ERR_INVALID, "ipc transport: channel %d", ch);
    if (npeers > IPC_MAX_RANKS) return set_error(MFFT_ERR_INVALID, "ipc transport: group of %d", npeers);
    if (sh->broken.load()) return set_error(MFFT_ERR_INTERNAL, "ipc transport: the group is broken (a rank failed earlier)");
    int myidx = -1;
    for (int i = 0; i < npeers; ++i)
      if (peers[i] == rank) myidx = i;
    if (myidx < 0) return set_error(MFFT_ERR_INVALID, "ipc transport: rank %d not in its own group", rank);
    if (rcount[myidx] && scount[myidx] != rcount[myidx]) return set_error(MFFT_ERR_INVALID, "ipc transport: self chunk size mismatch");
    // exchanges of one channel execute in the order they are issued, also when the channel moves to another stream
    if (!last_issue[ch]) MFFT_HIP(hipEventCreateWithFlags(&last_issue[ch], hipEventDisableTiming));
    if (used_ch[ch]) MFFT_HIP(hipStreamWaitEvent(s, last_issue[ch], 0));     // (a no-op on the same stream; a new stream may reuse a dead one's handle)
    // a record of this event is a release to SYSTEM scope of what the kernels before it wrote: the readers are other devices
    if (!release_ev[ch]) MFFT_HIP(hipEventCreateWithFlags(&release_ev[ch], hipEventDisableTiming | hipEventReleaseToSystem));
    const char* sp = static_cast<const char*>(send);
    char* rp = static_cast<char*>(recv);
    // 1. sender role: say where each peer finds its chunk (host), then, on the stream, that the data is there
    bool any_send = false;
    for (int i = 0; i < npeers; ++i) any_send = any_send || (peers[i] != rank && scount[i]);
    uint32_t seg = 0;
    uint64_t soff = 0;
    if (any_send) {
      MFFT_TRY(locate(send, &seg, &soff));
      MFFT_HIP(hipEventRecord(release_ev[ch], s));
    }
    std::vector<uint32_t> qs(npeers, 0);
    // kernel mode: the flag words of all peers are written / awaited by ONE small launch each (ipc_pull.h FlagOps) instead
    // of one stream memory operation per peer (each of which is a kernel launch of its own on this stack)
    const bool batched = pull_mode == PULL_KERNEL;
    FlagOps sig, wt;
    sig.n = wt.n = 0;
    for (int i = 0; i < npeers; ++i) {
      const int p = peers[i];
      if (p == rank || !scount[i]) continue;
      const uint32_t q = ++sseq[p][ch];
      qs[i] = q;
      IpcPair& pr = sh->pair[rank][p][ch];
      if (q > (uint32_t)IPC_RING)
        MFFT_TRY(spin([&] { return pr.consumed.load(std::memory_order_acquire) + IPC_RING >= q; }, "a peer to read an earlier post"));
      IpcPost& post = pr.ring[q % IPC_RING];
      post.seg = seg; post.offset = soff + sdisp[i]; post.bytes = scount[i];
      post.seq.store(q, std::memory_order_release);
      if (batched) { sig.addr[sig.n] = &peer_flags[p]->ready[rank][ch]; sig.value[sig.n++] = q; }
      else MFFT_HIP(hipStreamWriteValue32(s, &peer_flags[p]->ready[rank][ch], q, 0));
    }
    if (batched) MFFT_HIP(launch_flags(false, sig, s));      // one launch tells every peer
    // 2. receiver role, host part: where does every chunk lie (starting with the next rank: spreads the links when the
    //    chunks are pulled one after the other).  Nothing of the receiver role is enqueued before all posts are in.
    struct Pull { int i, p; uint32_t q; const char* src; };
    std::vector<Pull> pulls;
    for (int k = 1; k < npeers; ++k) {
      const int i = (myidx + k) % npeers, p = peers[i];
      if (!rcount[i]) continue;
      const uint32_t q = ++rseq[p][ch];
      IpcPair& pr = sh->pair[p][rank][ch];
      IpcPost& post = pr.ring[q % IPC_RING];
      MFFT_TRY(spin([&] { return post.seq.load(std::memory_order_acquire) == q; }, "a peer to post its send buffer"));
      const uint32_t pseg = post.seg;
      const uint64_t off = post.offset, bytes = post.bytes;
      pr.consumed.store(q, std::memory_order_release);
      if (bytes != rcount[i])
        return set_error(MFFT_ERR_INTERNAL, "ipc transport: rank %d expects %zu bytes from %d, which sends %llu", rank, rcount[i], p,
                         (unsigned long long)bytes);
      char* rbase = nullptr;
      MFFT_TRY(remote_base(p, pseg, &rbase));
      pulls.push_back(Pull{i, p, q, rbase + off});
    }
    // 3. receiver role, device part
    if (pull_mode == PULL_KERNEL) {
      // one launch for the self chunk and every peer's chunk, behind the waits for all of them
      for (const Pull& u : pulls) { wt.addr[wt.n] = &flags->ready[u.p][ch]; wt.value[wt.n++] = u.q; }
      MFFT_HIP(launch_flags(true, wt, s));
      PullArgs pa;
      memset(&pa, 0, sizeof pa);
      if (rcount[myidx]) pa.job[pa.njobs++] = PullJob{sp + sdisp[myidx], rp + rdisp[myidx], (unsigned long long)rcount[myidx]};
      for (const Pull& u : pulls) pa.job[pa.njobs++] = PullJob{u.src, rp + rdisp[u.i], (unsigned long long)rcount[u.i]};
      // pull_wgs workgroups per chunk is sized for an xGMI link; small groups get more, so that the launch as a whole
      // (the self chunk is a local copy) never has fewer than 56 workgroups (= 7 peers x 8)
      pa.wgs = pa.njobs > 0 ? std::max(pull_wgs, (56 + pa.njobs - 1) / pa.njobs) : pull_wgs;
      MFFT_HIP(launch_pull(pa, s));
      sig.n = 0;
      for (const Pull& u : pulls) { sig.addr[sig.n] = &peer_flags[u.p]->done[rank][ch]; sig.value[sig.n++] = u.q; }
      MFFT_HIP(launch_flags(false, sig, s));
    } else if (pull_mode == PULL_STREAMS && !pulls.empty()) {
      // every peer's copy on that peer's own stream, forked from and joined back into the issuing stream.  ALL flag
      // operations stay on the issuing stream (waits before the fork, "done" writes behind the join): the per-peer
      // streams carry nothing but one copy each.  (Round 2's first form had the waits and writes on the per-peer
      // streams: 8 processes on one device were 100x slower and pipelined transforms came back wrong in 3 of 6 runs,
      // profiles/r03_ipc_pull_modes.txt; that form was kept behind an environment switch through round 3 and is gone
      // since round 4 -- the last commit that has it is 292508d.)
      if (!fork_ev[ch]) MFFT_HIP(hipEventCreateWithFlags(&fork_ev[ch], hipEventDisableTiming));
      for (const Pull& u : pulls) MFFT_HIP(hipStreamWaitValue32(s, &flags->ready[u.p][ch], u.q, hipStreamWaitValueGte, 0xFFFFFFFFu));
      MFFT_HIP(hipEventRecord(fork_ev[ch], s));
      for (const Pull& u : pulls) {
        hipStream_t ps = nullptr;
        MFFT_TRY(peer_stream(u.p, &ps));
        if (!join_ev[u.p][ch]) MFFT_HIP(hipEventCreateWithFlags(&join_ev[u.p][ch], hipEventDisableTiming));
        MFFT_HIP(hipStreamWaitEvent(ps, fork_ev[ch], 0));
        MFFT_HIP(hipMemcpyAsync(rp + rdisp[u.i], u.src, rcount[u.i], hipMemcpyDeviceToDevice, ps));
        MFFT_HIP(hipEventRecord(join_ev[u.p][ch], ps));
      }
      if (rcount[myidx]) MFFT_HIP(hipMemcpyAsync(rp + rdisp[myidx], sp + sdisp[myidx], rcount[myidx], hipMemcpyDeviceToDevice, s));
      for (const Pull& u : pulls) MFFT_HIP(hipStreamWaitEvent(s, join_ev[u.p][ch], 0));
      for (const Pull& u : pulls) MFFT_HIP(hipStreamWriteValue32(s, &peer_flags[u.p]->done[rank][ch], u.q, 0));
    } else {
      if (rcount[myidx]) MFFT_HIP(hipMemcpyAsync(rp + rdisp[myidx], sp + sdisp[myidx], rcount[myidx], hipMemcpyDeviceToDevice, s));
      for (const Pull& u : pulls) {
        MFFT_HIP(hipStreamWaitValue32(s, &flags->ready[u.p][ch], u.q, hipStreamWaitValueGte, 0xFFFFFFFFu));
        MFFT_HIP(hipMemcpyAsync(rp + rdisp[u.i], u.src, rcount[u.i], hipMemcpyDeviceToDevice, s));
        MFFT_HIP(hipStreamWriteValue32(s, &peer_flags[u.p]->done[rank][ch], u.q, 0));
      }
    }
    // 4. sender role: nobody overwrites its send buffer before all peers have pulled from it
    if (batched) {
      wt.n = 0;
      for (int i = 0; i < npeers; ++i)
        if (qs[i]) { wt.addr[wt.n] = &flags->done[peers[i]][ch]; wt.value[wt.n++] = qs[i]; }
      MFFT_HIP(launch_flags(true, wt, s));
    } else {
      for (int i = 0; i < npeers; ++i)
        if (qs[i]) MFFT_HIP(hipStreamWaitValue32(s, &flags->done[peers[i]][ch], qs[i], hipStreamWaitValueGte, 0xFFFFFFFFu));
    }
    MFFT_HIP(hipEventRecord(last_issue[ch], s));
    last_stream[ch] = s;
    used_ch[ch] = true;
    return 0;
  }

  // ---- relayed sub-group exchange ---------------------------------------------------------------------------------
  // Relaying only pays when the ranks sit on different devices (two hops over otherwise idle links); with ranks sharing
  // a device it doubles the local traffic.  Off unless asked for (MFFT_IPC_RELAY=1 or the "ipc_relay" option, the same on
  // every rank): it has moved real data between processes on one device only, and bench.py measures it as a candidate
  // of the pencil runs whenever every rank owns a device.
  bool relay_enabled() const { return relay_mode == 1; }
  // every rank calls this with its own choice; a disagreement switches relaying off everywhere and is an error everywhere
  int agree_on_relay() {
    double v[2] = {(double)relay_mode, -(double)relay_mode};
    MFFT_TRY(allreduce_host(v, 2, 1));                     // max and -min
    if (v[0] != -v[1]) {
      relay_mode = 0;
      return set_error(MFFT_ERR_INVALID, "ipc transport: the ranks disagree on ipc_relay (MFFT_IPC_RELAY / the \"ipc_relay\" option "
                                         "must be the same on every rank); relaying is switched off");
    }
    return 0;
  }
  // A relayed exchange needs every group of the partition to have the same size g with 2 <= g < P (the stripes of a
  // message are cut by the sender's group size and read by relays of other groups with theirs).  Decided from the
  // partition alone, which every rank holds, so all ranks take the same branch.
  static bool relay_fits(const int* part, int P) {
    int cnt[IPC_MAX_RANKS] = {};
    for (int r = 0; r < P; ++r) {
      if (part[r] < 0 || part[r] >= IPC_MAX_RANKS) return false;
      ++cnt[part[r]];
    }
    const int g = cnt[part[0]];
    if (g < 2 || g >= P) return false;
    for (int i = 0; i < IPC_MAX_RANKS; ++i)
      if (cnt[i] && cnt[i] != g) return false;
    return true;
  }
  int alltoallv_part(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
                     const size_t* rdisp, const int* peers, int npeers, hipStream_t s, int channel, const int* part) override {
    if (!part || !relay_enabled() || !relay_fits(part, size))
      return alltoallv(send, scount, sdisp, recv, rcount, rdisp, peers, npeers, s, channel);
    const int rc = exchange_relay(send, scount, sdisp, recv, rcount, rdisp, peers, npeers, s, channel, part);
    if (rc != 0) rescue();
    return rc;
  }
  int launch_jobs(const std::vector<PullJob>& jobs, hipStream_t s) {
    for (size_t i0 = 0; i0 < jobs.size(); i0 += PULL_MAX_JOBS) {
      PullArgs pa;
      memset(&pa, 0, sizeof pa);
      for (size_t i = i0; i < jobs.size() && i < i0 + PULL_MAX_JOBS; ++i) pa.job[pa.njobs++] = jobs[i];
      pa.wgs = std::max(pull_wgs, (56 + pa.njobs - 1) / pa.njobs);
      MFFT_HIP(launch_pull(pa, s));
    }
    return 0;
  }
  // one wave signals / awaits the flag word `which` (rready / k1done / k2done) of the listed ranks
  typedef uint32_t (IpcFlags::*FlagArray)[IPC_MAX_RANKS][IPC_MAX_CH];
  int flags_all(bool wait, FlagArray which, int ch, uint32_t q, const std::vector<int>& ranks, hipStream_t s) {
    FlagOps f;
    f.n = 0;
    for (int r : ranks) {
      if (r == rank) continue;
      // wait: word [r] of MY flags (written by r); signal: word [rank] of r's flags
      f.addr[f.n] = wait ? &(flags->*which)[r][ch] : &(peer_flags[r]->*which)[rank][ch];
      f.value[f.n] = q;
      if (++f.n == FLAG_MAX_OPS) return set_error(MFFT_ERR_INVALID, "ipc transport: too many flag operations");
    }
    MFFT_HIP(launch_flags(wait, f, s));
    return 0;
  }
  int exchange_relay(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
                     const size_t* rdisp, const int* peers, int npeers, hipStream_t s, int ch, const int* part) {
    if (ch < 0 || ch >= IPC_MAX_CH) return set_error(MFFT_ERR_INVALID, "ipc transport: channel %d", ch);
    if (sh->broken.load()) return set_error(MFFT_ERR_INTERNAL, "ipc transport: the group is broken (a rank failed earlier)");
    const int P = size, g = npeers, gid = part[rank];
    int myidx = -1, members = 0;
    for (int i = 0; i < npeers; ++i) {
      if (peers[i] == rank) myidx = i;
      if (part[peers[i]] != gid) return set_error(MFFT_ERR_INVALID, "ipc transport: peer %d is not in group %d of the partition", peers[i], gid);
    }
    for (int r = 0; r < P; ++r) members += part[r] == gid;
    if (myidx < 0 || members != g) return set_error(MFFT_ERR_INVALID, "ipc transport: the peer list and the partition disagree");
    if (rcount[myidx] && scount[myidx] != rcount[myidx]) return set_error(MFFT_ERR_INVALID, "ipc transport: self chunk size mismatch");
    if (!last_issue[ch]) MFFT_HIP(hipEventCreateWithFlags(&last_issue[ch], hipEventDisableTiming));
    if (used_ch[ch]) MFFT_HIP(hipStreamWaitEvent(s, last_issue[ch], 0));
    if (!release_ev[ch]) MFFT_HIP(hipEventCreateWithFlags(&release_ev[ch], hipEventDisableTiming | hipEventReleaseToSystem));
    const char* sp = static_cast<const char*>(send);
    char* rp = static_cast<char*>(recv);
    const uint32_t q = ++xseq[ch];
    const int slot = (int)(q & 1u);
    std::vector<int> everyone, outsiders;
    for (int r = 0; r < P; ++r) {
      if (r != rank) everyone.push_back(r);
      if (part[r] != gid) outsiders.push_back(r);
    }
    // 1. where my messages lie; on the stream: the data is there (told to EVERY rank: group peers read the direct
    //    parts, everybody else a first-hop stripe)
    uint32_t seg = 0;
    uint64_t soff = 0;
    bool any_send = false;
    for (int i = 0; i < npeers; ++i) any_send = any_send || (peers[i] != rank && scount[i]);
    if (any_send) MFFT_TRY(locate(send, &seg, &soff));
    for (int i = 0; i < npeers; ++i) {
      if (peers[i] == rank) continue;
      IpcRPost& po = sh->rpost[rank][peers[i]][ch][slot];
      po.seg = seg; po.offset = soff + sdisp[i]; po.bytes = scount[i];
    }
    MFFT_HIP(hipEventRecord(release_ev[ch], s));
    MFFT_TRY(flags_all(false, &IpcFlags::rready, ch, q, everyone, s));
    MFFT_TRY(barrier());                                   // every post of exchange q is visible
    // 2. what I pull (relay_plan.h relay_moves): as a destination -- direct halves, second hops -- and as a relay of the
    //    other groups' messages (first hops into my staging area)
    auto idx_of = [&](int r) { for (int i = 0; i < npeers; ++i) if (peers[i] == r) return i; return -1; };
    for (int i = 0; i < npeers; ++i) {                     // what my group peers send me must be what I expect
      const int p = peers[i];
      if (p == rank) continue;
      const uint64_t b = sh->rpost[p][rank][ch][slot].bytes;
      if (b != rcount[i])
        return set_error(MFFT_ERR_INTERNAL, "ipc transport: rank %d expects %zu bytes from %d, which sends %llu", rank, rcount[i], p,
                         (unsigned long long)b);
    }
    std::vector<RelayMove> moves;
    relay_moves(P, rank, part, [&](int sr, int d) -> size_t {
      if (sr == d) return sr == rank ? rcount[myidx] : 0;
      return (size_t)sh->rpost[sr][d][ch][slot].bytes;
    }, &moves);
    size_t need = 0;
    for (const RelayMove& m : moves)
      if (m.kind == 2) need += (m.bytes + 255) / 256 * 256;
    if (need > staging_bytes[ch]) {                        // (first use, or a larger exchange: rare; frees synchronise the device)
      if (staging[ch]) MFFT_TRY(work_free(staging[ch]));
      staging[ch] = nullptr;
      staging_bytes[ch] = 0;
      MFFT_TRY(work_alloc(&staging[ch], need));
      staging_bytes[ch] = need;
    }
    uint32_t stg_seg = 0;
    uint64_t stg_off = 0;
    if (need) MFFT_TRY(locate(staging[ch], &stg_seg, &stg_off));
    std::vector<PullJob> k1, k2;
    size_t fill = 0;
    for (const RelayMove& m : moves) {
      if (m.kind == 0) {
        k1.push_back(PullJob{sp + sdisp[myidx], rp + rdisp[myidx], (unsigned long long)m.bytes});
      } else if (m.kind == 1) {
        const IpcRPost po = sh->rpost[m.msg_src][rank][ch][slot];
        char* rbase = nullptr;
        MFFT_TRY(remote_base(m.msg_src, po.seg, &rbase));
        (m.phase == 1 ? k1 : k2).push_back(PullJob{rbase + po.offset + m.msg_off, rp + rdisp[idx_of(m.msg_src)] + m.msg_off,
                                                   (unsigned long long)m.bytes});
      } else if (m.kind == 2) {
        const IpcRPost po = sh->rpost[m.msg_src][m.msg_dst][ch][slot];
        char* rbase = nullptr;
        MFFT_TRY(remote_base(m.msg_src, po.seg, &rbase));
        k1.push_back(PullJob{rbase + po.offset + m.msg_off, static_cast<char*>(staging[ch]) + fill, (unsigned long long)m.bytes});
        IpcStage& st = sh->stage[rank][m.msg_src][m.msg_dst][ch][slot];
        st.seg = stg_seg; st.offset = stg_off + fill;
        fill += (m.bytes + 255) / 256 * 256;
      }
    }
    MFFT_TRY(barrier());                                   // every relay has said where it stages
    for (const RelayMove& m : moves) {
      if (m.kind != 3) continue;
      const IpcStage st = sh->stage[m.from][m.msg_src][rank][ch][slot];
      char* rbase = nullptr;
      MFFT_TRY(remote_base(m.from, st.seg, &rbase));
      k2.push_back(PullJob{rbase + st.offset, rp + rdisp[idx_of(m.msg_src)] + m.msg_off, (unsigned long long)m.bytes});
    }
    // 3. phase 1: everybody's data is there -> direct halves, self chunk, first hops -> tell everybody
    MFFT_TRY(flags_all(true, &IpcFlags::rready, ch, q, everyone, s));
    MFFT_TRY(launch_jobs(k1, s));
    MFFT_HIP(hipEventRecord(release_ev[ch], s));           // the first hops staged in MY memory are read by other devices next
    MFFT_TRY(flags_all(false, &IpcFlags::k1done, ch, q, everyone, s));
    // 4. phase 2: the relays have staged -> second halves, second hops -> tell everybody
    MFFT_TRY(flags_all(true, &IpcFlags::k1done, ch, q, outsiders, s));
    MFFT_TRY(launch_jobs(k2, s));
    MFFT_TRY(flags_all(false, &IpcFlags::k2done, ch, q, everyone, s));
    // 5. my send buffer and my staging area are free once nobody reads exchange q any more
    MFFT_TRY(flags_all(true, &IpcFlags::k1done, ch, q, everyone, s));
    MFFT_TRY(flags_all(true, &IpcFlags::k2done, ch, q, everyone, s));
    MFFT_HIP(hipEventRecord(last_issue[ch], s));
    last_stream[ch] = s;
    used_ch[ch] = true;
    return 0;
  }
};

std::string ipc_shm_name(const unsigned char* id128) {
  char nm[64] = "/mfft-";
  for (int i = 0; i < 16; ++i) snprintf(nm + 6 + 2 * i, 3, "%02x", id128[8 + i]);
  return nm;
}

}  // namespace

// a unique id of the IPC transport: magic + random bytes (names the shared-memory segment of the group)
int ipc_make_unique_id(void* id128) {
  unsigned char* b = static_cast<unsigned char*>(id128);
  memset(b, 0, MFFT_UNIQUE_ID_BYTES);
  memcpy(b, &IPC_MAGIC, 8);
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, b + 8, 16) != 16) {
    if (fd >= 0) close(fd);
    return set_error(MFFT_ERR_INTERNAL, "cannot read /dev/urandom");
  }
  close(fd);
  return 0;
}
bool ipc_is_unique_id(const void* id128) { return memcmp(id128, &IPC_MAGIC, 8) == 0; }

int comm_create_ipc(int nranks, int rank, const void* id128, mfft_comm_s** out) {
  if (nranks < 1 || nranks > IPC_MAX_RANKS || rank < 0 || rank >= nranks)
    return set_error(MFFT_ERR_INVALID, "ipc transport: rank %d of %d (at most %d ranks, one node)", rank, nranks, IPC_MAX_RANKS);
  std::unique_ptr<IpcComm> c(new IpcComm());
  c->size = nranks;
  c->rank = rank;
  MFFT_HIP(hipGetDevice(&c->device));
  c->shm_name = ipc_shm_name(static_cast<const unsigned char*>(id128));
  int fd = -1;
  if (rank == 0) {
    (void)shm_unlink(c->shm_name.c_str());
    fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return set_error(MFFT_ERR_INTERNAL, "shm_open(%s) failed: %s", c->shm_name.c_str(), strerror(errno));
    c->creator = true;
    if (ftruncate(fd, sizeof(IpcShm)) != 0) { close(fd); return set_error(MFFT_ERR_INTERNAL, "ftruncate failed: %s", strerror(errno)); }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(IpcShm)) break;
        close(fd);
        fd = -1;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(ipc_timeout_s()))
        return set_error(MFFT_ERR_INTERNAL, "ipc transport: rank 0 never created %s", c->shm_name.c_str());
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  void* m = mmap(nullptr, sizeof(IpcShm), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return set_error(MFFT_ERR_INTERNAL, "mmap of %s failed: %s", c->shm_name.c_str(), strerror(errno));
  c->sh = static_cast<IpcShm*>(m);
  IpcShm* sh = c->sh;
  // any failure from here on leaves the group unusable: say so, so that the peers' spins fail at once instead of
  // timing out and this rank's destructor does not wait for them to detach
  struct BreakOnFailure {
    IpcShm* sh; bool armed;
    ~BreakOnFailure() { if (armed) sh->broken.store(1); }
  } guard{sh, true};
  if (rank == 0) {
    sh->nranks = nranks;                                // the segment is zero-filled by ftruncate
    sh->magic.store(IPC_MAGIC, std::memory_order_release);
  } else {
    MFFT_TRY(c->spin([&] { return sh->magic.load(std::memory_order_acquire) == IPC_MAGIC; }, "rank 0 to initialise the segment"));
    if (sh->nranks != nranks) return set_error(MFFT_ERR_INVALID, "ipc transport: rank 0 says %d ranks, this rank %d", sh->nranks, nranks);
  }
  int can = 0;
  (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, c->device);
  if (!can) return set_error(MFFT_ERR_UNSUPPORTED, "ipc transport: this device has no stream memory operations");
  // my flags: zeroed, exported (the first and only export besides the work segments)
  // fine-grained device memory where the runtime has it: these words are written by OTHER devices' stream memory
  // operations and polled by this device's; ordinary (coarse-grained) memory otherwise
  {
    void* fp = nullptr;
    bool fine = hipExtMallocWithFlags(&fp, (size_t)2 << 20, hipDeviceMallocFinegrained) == hipSuccess && fp != nullptr;
    if (fine && (hipMemset(fp, 0, sizeof(IpcFlags)) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
                 hipIpcGetMemHandle(&sh->rk[rank].flags_h, fp) != hipSuccess)) {
      (void)hipGetLastError();
      (void)hipFree(fp);
      fine = false;
    }
    if (!fine) {
      (void)hipGetLastError();
      MFFT_HIP(hipMalloc(&fp, (size_t)2 << 20));
      MFFT_HIP(hipMemset(fp, 0, sizeof(IpcFlags)));
      MFFT_HIP(hipDeviceSynchronize());
      MFFT_HIP(hipIpcGetMemHandle(&sh->rk[rank].flags_h, fp));
    }
    c->flags = static_cast<IpcFlags*>(fp);
    c->flags_fine = fine;
    if (getenv("MFFT_IPC_DEBUG")) fprintf(stderr, "[mfft ipc] rank %d: flag words in %s memory\n", rank, fine ? "fine-grained" : "coarse-grained");
  }
  sh->rk[rank].device = c->device;
  sh->rk[rank].pid = (int)getpid();
  sh->attached.fetch_add(1);
  MFFT_TRY(c->spin([&] { return sh->attached.load() >= (uint32_t)nranks; }, "every rank to attach"));
  c->peer_flags.assign(nranks, nullptr);
  for (int r = 0; r < nranks; ++r) {
    if (r == rank) { c->peer_flags[r] = c->flags; continue; }
    void* p = nullptr;
    MFFT_HIP(hipIpcOpenMemHandle(&p, sh->rk[r].flags_h, hipIpcMemLazyEnablePeerAccess));
    c->peer_flags[r] = static_cast<IpcFlags*>(p);
  }
  MFFT_TRY(c->barrier());
  MFFT_TRY(c->agree_on_relay());                        // MFFT_IPC_RELAY is read per process
  if (rank == 0) {                                      // everybody holds a mapping: the name can go
    (void)shm_unlink(c->shm_name.c_str());
    c->creator = false;
  }
  guard.armed = false;
  *out = c.release();
  return 0;
}

}  // namespace mfft
