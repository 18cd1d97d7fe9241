// pm_comm.cpp -- the exchange step of a position-sharded scan, behind the C ABI (include/pm_gpu.h
// "Multi-GPU exchange"): RCCL point-to-point transfers of 16-byte hit records from every rank's HBM
// into rank 0's, over xGMI.  The reference has no counterpart -- its scan is one serial pass
// (primer_match.cc:1118, pcr_match.cc:948); SURVEY.md 8(e) defines the sharded form.
//
// librccl.so is loaded on first use (dlopen), so libpm_gpu.so itself loads on a box without RCCL
// and single-rank callers never touch it.  One communicator = one rank = one process = one GPU; the
// 128-byte unique id travels from rank 0 to the others through whatever the launcher has (the host
// programs of this repo use the pipes of host/pm_ranks.cc).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/pm_gpu.h"

namespace {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl &rccl() {
  static Rccl r;
  if (r.lib || !r.err.empty()) return r;
  r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!r.lib) { r.err = std::string("cannot load librccl.so: ") + dlerror(); return r; }
  auto sym = [&](const char *name) -> void * {
    void *p = dlsym(r.lib, name);
    if (!p && r.err.empty()) r.err = std::string("librccl.so lacks ") + name;
    return p;
  };
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
  r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
  r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
  r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
  r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
  r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
  return r;
}

thread_local std::string g_comm_error;

}  // namespace

struct pm_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  void *d_recv = nullptr;                // rank 0: landing buffer of the gathered records
  size_t recv_cap = 0;
  std::string err;
};

static int cfail(pm_comm *c, int code, const std::string &msg) {
  if (c) c->err = msg; else g_comm_error = msg;
  return code;
}

static_assert(PM_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int pm_comm_unique_id(void *id_out) {
  if (!id_out) return cfail(nullptr, PM_E_INVALID, "pm_comm_unique_id: null argument");
  Rccl &r = rccl();
  if (!r.err.empty()) return cfail(nullptr, PM_E_UNSUPPORTED, r.err);
  ncclUniqueId id;
  const ncclResult_t rc = r.GetUniqueId(&id);
  if (rc != ncclSuccess) return cfail(nullptr, PM_E_HIP, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
  memcpy(id_out, &id, sizeof(id));
  return PM_OK;
}

extern "C" int pm_comm_create(int device, int rank, int world, const void *id, pm_comm **out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return cfail(nullptr, PM_E_INVALID, "pm_comm_create: bad arguments");
  Rccl &r = rccl();
  if (!r.err.empty()) return cfail(nullptr, PM_E_UNSUPPORTED, r.err);
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return cfail(nullptr, PM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  pm_comm *c = new pm_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  const ncclResult_t rc = r.CommInitRank(&c->comm, world, uid, rank);
  if (rc != ncclSuccess) { const std::string m = std::string("ncclCommInitRank: ") + r.GetErrorString(rc); delete c; return cfail(nullptr, PM_E_HIP, m); }
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { (void)r.CommDestroy(c->comm); delete c; return cfail(nullptr, PM_E_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
  *out = c;
  return PM_OK;
}

// counts[r] = records rank r contributes (every rank passes the same array: the caller's count
// exchange came first).  Every rank -- rank 0 included -- sends its n_send = counts[r] records
// straight out of HBM to rank 0; rank 0 receives them in rank order into one landing buffer and
// copies the whole list to host_out (sum of counts records).  Rank 0's own records travel as a
// send/receive to itself inside the same group (RCCL turns that into a device copy), so the exchange
// is one code path for every world size, the single-GPU one included.  d_send must not be in use by
// work still running on another stream.  A rank that fails locally aborts the communicator so that
// its peers' transfers end with an error instead of waiting for it.
extern "C" int pm_comm_gather(pm_comm *c, const void *d_send, size_t n_send, const uint64_t *counts, pm_hit *host_out) {
  if (!c || !counts || (n_send && !d_send)) return cfail(c, PM_E_INVALID, "pm_comm_gather: bad arguments");
  if (counts[c->rank] != n_send) return cfail(c, PM_E_INVALID, "pm_comm_gather: counts[rank] != n_send");
  if (!c->comm) return cfail(c, PM_E_INVALID, "pm_comm_gather: the communicator was aborted by an earlier failure");
  Rccl &r = rccl();
  auto give_up = [&](int code, const std::string &msg) {             // peers must not wait for this rank
    if (c->comm) { (void)r.CommAbort(c->comm); c->comm = nullptr; }
    return cfail(c, code, msg);
  };
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return give_up(PM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  size_t total = 0;
  for (int q = 0; q < c->world; ++q) total += (size_t)counts[q];
  if (c->rank == 0) {
    if (total && !host_out) return give_up(PM_E_INVALID, "pm_comm_gather: rank 0 needs host_out");
    if (c->recv_cap < total) {
      if (c->d_recv) (void)hipFree(c->d_recv);
      c->d_recv = nullptr;
      c->recv_cap = total + total / 4 + 1024;
      if ((e = hipMalloc(&c->d_recv, c->recv_cap * sizeof(pm_hit))) != hipSuccess) { c->recv_cap = 0; return give_up(PM_E_NOMEM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    }
  }
  ncclResult_t rc = r.GroupStart();
  if (rc != ncclSuccess) return give_up(PM_E_HIP, std::string("ncclGroupStart: ") + r.GetErrorString(rc));
  if (c->rank == 0) {
    size_t at = 0;
    for (int q = 0; q < c->world && rc == ncclSuccess; ++q) {
      if (counts[q]) rc = r.Recv(static_cast<char *>(c->d_recv) + at * sizeof(pm_hit), (size_t)counts[q] * sizeof(pm_hit), ncclUint8, q, c->comm, c->stream);
      at += (size_t)counts[q];
    }
  }
  if (n_send && rc == ncclSuccess) rc = r.Send(d_send, n_send * sizeof(pm_hit), ncclUint8, 0, c->comm, c->stream);
  const ncclResult_t rc2 = r.GroupEnd();
  if (rc != ncclSuccess || rc2 != ncclSuccess) return give_up(PM_E_HIP, std::string("RCCL send/recv: ") + r.GetErrorString(rc != ncclSuccess ? rc : rc2));
  if (c->rank == 0 && total && (e = hipMemcpyAsync(host_out, c->d_recv, total * sizeof(pm_hit), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
    return give_up(PM_E_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return give_up(PM_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  return PM_OK;
}

extern "C" void pm_comm_destroy(pm_comm *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" const char *pm_comm_last_error(const pm_comm *c) { return c ? c->err.c_str() : g_comm_error.c_str(); }
