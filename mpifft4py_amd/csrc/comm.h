// comm.h -- communicator abstraction behind mfft_comm_t.
//
// Replaces the mpi4py communicator the reference receives in its constructors
// (slab.py:77-81, pencil.py:173-195).  Four transports:
//   SelfComm  : P = 1.
//   RcclComm  : one process per GPU, RCCL (loaded with dlopen at first use) over
//               xGMI; the exchange is a grouped ncclSend/ncclRecv all-to-all-v.
//   IpcComm   : one process per GPU on one node; every rank PULLS its chunks out of
//               the peers' IPC-mapped work buffers (one pull kernel over all peers, or
//               copy-engine transfers), cross-process ordering through stream memory
//               operations (ipc_comm.hip).  Selected with MFFT_TRANSPORT=ipc on rank 0
//               (the unique id names the transport).
//   LocalComm : P virtual ranks inside ONE process, each driven by its own host
//               thread, exchanging with peer-to-peer device copies.  Used for
//               single-process multi-GPU runs and to exercise the full
//               distributed algorithm on a single GPU.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <vector>

struct mfft_comm_s {
  int size = 1, rank = 0;
  virtual ~mfft_comm_s() {}
  // All-to-all-v inside a sub-group.  `peers` lists the comm ranks of the group
  // (every member passes the same list); chunk i is sent to / received from
  // peers[i].  Counts and displacements are in BYTES.  Enqueued on stream s.
  // `channel` names the issuing stream's role (0: the plan's compute stream, 1: its
  // communication stream): exchanges of one channel execute in the order they are issued.
  virtual int alltoallv(const void* send, const size_t* scount, const size_t* sdisp, void* recv,
                        const size_t* rcount, const size_t* rdisp, const int* peers, int npeers,
                        hipStream_t s, int channel = 0) = 0;
  // The same when EVERY rank of the communicator runs an exchange of its own group at this point and the groups
  // partition the ranks: part[r] = group id of rank r (the pencils' comm0 / comm1 exchanges).  A transport may then use
  // the links to ranks outside the group as well (IpcComm: two-hop relay striping, relay_plan.h); the default ignores it.
  virtual int alltoallv_part(const void* send, const size_t* scount, const size_t* sdisp, void* recv, const size_t* rcount,
                             const size_t* rdisp, const int* peers, int npeers, hipStream_t s, int channel, const int* part) {
    (void)part;
    return alltoallv(send, scount, sdisp, recv, rcount, rdisp, peers, npeers, s, channel);
  }
  virtual int barrier() = 0;
  virtual int bcast_host(void* buf, size_t bytes, int root) = 0;
  virtual int allreduce_host(double* vals, int count, int op /*0 sum, 1 max*/) = 0;
  virtual void abort() {}    // wake every rank blocked in a host-side barrier of this group
  // Plan work buffers (the only buffers an exchange ever SENDS from).  A transport that has to make them reachable
  // for its peers hands them out itself (IpcComm: an arena of IPC-exported segments); the others use hipMalloc.
  virtual int work_alloc(void** p, size_t bytes);
  virtual int work_free(void* p);
  // One small all-to-all over all ranks with a known pattern, checked on the host; waits at most timeout_ms for the
  // GPU side (a transport whose device-side waits can be released from the host does so: `rescue`).  Lets a caller
  // try a transport on a machine it has never run on without risking a hang (bench.py --transport auto).
  int selftest(size_t bytes_per_peer, int timeout_ms);
  virtual void rescue() {}
  // transport-specific knobs (mfft_comm_set_option / mfft_comm_get_option); unknown keys are an error / -1
  virtual int set_option(const char* key, long long value);
  virtual long long get_option(const char* key);
  // plans hold a reference: a communicator destroyed before its plans lives until the last of them is gone
  int plan_refs = 0;
  bool destroy_requested = false;
};

namespace mfft {
int comm_create_self(mfft_comm_s** out);
int comm_get_unique_id(void* id128);
int comm_create_rccl(int nranks, int rank, const void* id128, mfft_comm_s** out);
int comm_create_local(int nranks, const int* devices, mfft_comm_s** out);
int comm_create_ipc(int nranks, int rank, const void* id128, mfft_comm_s** out);
int ipc_make_unique_id(void* id128);
bool ipc_is_unique_id(const void* id128);
int dev_alloc(void** p, size_t bytes);    // hipMalloc / hipFree with the library's error reporting
int dev_free(void* p);
}  // namespace mfft
