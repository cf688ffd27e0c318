// gather_host.h -- auvp_comm_* / auvp_gather*: the multi-GPU result gather of SURVEY.md 8(b)/(e) on RCCL.
//
// Episodes are independent, so the only cross-GPU step of the path is one all-gather of result records after
// the kernels (one process per GPU, block partition of the episode index).  These entry points give a C/C++
// host that step without torch: the communicator lives in the planner handle, the collectives run on the
// handle's own HIP stream, buffers are device pointers.
//
// RCCL is bound at run time (dlopen "librccl.so.1"), not at link time: a single-GPU user never needs it, and a
// process that already holds an RCCL (PyTorch-ROCm bundles one under the same SONAME) must not get a second
// copy.  AUVP_RCCL_LIBRARY overrides the name.
//
// Included at the end of auvplan.hip (same translation unit: uses auvp_handle, fail, HIPCHK, DevBuf).
#ifndef AUVP_GATHER_HOST_H
#define AUVP_GATHER_HOST_H
#include <dlfcn.h>

namespace {

// the slice of rccl.h this file uses, declared locally so the library builds without the RCCL headers
typedef struct ncclComm* auvp_ncclComm_t;
struct auvp_ncclUniqueId { char internal[AUVP_COMM_ID_BYTES]; };
enum { AUVP_NCCL_UINT8 = 1, AUVP_NCCL_INT64 = 4 };  // ncclUint8 / ncclInt64 (rccl.h ncclDataType_t)

struct RcclApi {
  void* so = nullptr;
  int (*GetUniqueId)(auvp_ncclUniqueId*) = nullptr;
  int (*CommInitRank)(auvp_ncclComm_t*, int, auvp_ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(auvp_ncclComm_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, auvp_ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, auvp_ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err;
};

RcclApi* rccl_api() {
  static RcclApi api;
  if (api.so || !api.err.empty()) return &api;
  const char* names[] = {getenv("AUVP_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    if (!n || !*n) continue;
    api.so = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (api.so) break;
  }
  if (!api.so) { api.err = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return &api; }
  auto sym = [&](const char* s) { void* p = dlsym(api.so, s); if (!p && api.err.empty()) api.err = std::string("RCCL symbol missing: ") + s; return p; };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
  api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!api.err.empty()) { dlclose(api.so); api.so = nullptr; }
  return &api;
}

struct CommState {
  auvp_ncclComm_t comm = nullptr;
  int world = 0, rank = 0;
  DevBuf counts;  // [world] int64 byte counts of a variable-length gather
  hipEvent_t g0 = nullptr, g1 = nullptr;
  double last_ms = 0.0;
  ~CommState() {
    if (comm) { RcclApi* a = rccl_api(); if (a->so) (void)a->CommDestroy(comm); }
    if (g0) (void)hipEventDestroy(g0);
    if (g1) (void)hipEventDestroy(g1);
  }
};

CommState* comm_of(auvp_handle* h) { return static_cast<CommState*>(h->comm); }

#define RCCLCHK(h, a, call)                                                                        \
  do {                                                                                             \
    int r__ = (call);                                                                              \
    if (r__ != 0) return fail(h, AUVP_ERR_COMM, "%s: %s", #call, (a)->GetErrorString ? (a)->GetErrorString(r__) : "?"); \
  } while (0)

}  // namespace

extern "C" {

int auvp_comm_unique_id(uint8_t* id_out) {
  if (!id_out) return AUVP_ERR_ARG;
  RcclApi* a = rccl_api();
  if (!a->so) return AUVP_ERR_COMM;
  auvp_ncclUniqueId id;
  if (a->GetUniqueId(&id) != 0) return AUVP_ERR_COMM;
  memcpy(id_out, id.internal, AUVP_COMM_ID_BYTES);
  return AUVP_OK;
}

int auvp_comm_init(auvp_handle* h, int32_t world_size, int32_t rank, const uint8_t* id) {
  if (!h || !id || world_size <= 0 || rank < 0 || rank >= world_size) return h ? fail(h, AUVP_ERR_ARG, "bad communicator arguments") : AUVP_ERR_ARG;
  RcclApi* a = rccl_api();
  if (!a->so) return fail(h, AUVP_ERR_COMM, "%s", a->err.c_str());
  HIPCHK(h, hipSetDevice(h->device));
  if (h->comm) { delete comm_of(h); h->comm = nullptr; }
  CommState* c = new CommState();
  h->comm = c;
  h->comm_free = [](void* p) { delete static_cast<CommState*>(p); };
  auvp_ncclUniqueId uid;
  memcpy(uid.internal, id, AUVP_COMM_ID_BYTES);
  RCCLCHK(h, a, a->CommInitRank(&c->comm, world_size, uid, rank));
  c->world = world_size; c->rank = rank;
  HIPCHK(h, hipEventCreate(&c->g0));
  HIPCHK(h, hipEventCreate(&c->g1));
  HIPCHK(h, c->counts.reserve((size_t)world_size * sizeof(int64_t)));
  return AUVP_OK;
}

int auvp_comm_destroy(auvp_handle* h) {
  if (!h) return AUVP_ERR_ARG;
  if (h->comm) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete comm_of(h);
    h->comm = nullptr;
  }
  return AUVP_OK;
}

int auvp_gather(auvp_handle* h, const void* send_dev, size_t bytes_per_rank, void* recv_dev) {
  if (!h || !send_dev || !recv_dev) return AUVP_ERR_ARG;
  CommState* c = comm_of(h);
  if (!c || !c->comm) return fail(h, AUVP_ERR_STATE, "auvp_comm_init not called");
  RcclApi* a = rccl_api();
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(c->g0, h->stream));
  if (bytes_per_rank) RCCLCHK(h, a, a->AllGather(send_dev, recv_dev, bytes_per_rank, AUVP_NCCL_UINT8, c->comm, h->stream));
  HIPCHK(h, hipEventRecord(c->g1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, c->g0, c->g1));
  c->last_ms = ms;
  return AUVP_OK;
}

int auvp_gather_var(auvp_handle* h, const void* send_dev, int64_t send_bytes, void* recv_dev, int64_t recv_cap_bytes,
                    int64_t* counts_out) {
  if (!h || send_bytes < 0 || !counts_out || (send_bytes > 0 && !send_dev)) return AUVP_ERR_ARG;
  CommState* c = comm_of(h);
  if (!c || !c->comm) return fail(h, AUVP_ERR_STATE, "auvp_comm_init not called");
  RcclApi* a = rccl_api();
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(c->g0, h->stream));
  // phase 1: every rank's byte count
  HIPCHK(h, h->d_tmp5.reserve(sizeof(int64_t)));
  HIPCHK(h, hipMemcpyAsync(h->d_tmp5.p, &send_bytes, sizeof(int64_t), hipMemcpyHostToDevice, h->stream));
  RCCLCHK(h, a, a->AllGather(h->d_tmp5.p, c->counts.p, 1, AUVP_NCCL_INT64, c->comm, h->stream));
  HIPCHK(h, hipMemcpyAsync(counts_out, c->counts.p, (size_t)c->world * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int64_t total = 0;
  for (int r = 0; r < c->world; r++) total += counts_out[r];
  if (!recv_dev || total > recv_cap_bytes) {
    // sizes are reported (two-phase convention): the caller allocates sum(counts_out) bytes and calls again
    return recv_dev ? fail(h, AUVP_ERR_CAPACITY, "gather needs %lld bytes, buffer holds %lld", (long long)total, (long long)recv_cap_bytes)
                    : AUVP_OK;
  }
  // phase 2: rank r's block lands at the prefix offset of r -- one grouped broadcast per rank, no padding to the
  // largest block
  RCCLCHK(h, a, a->GroupStart());
  int64_t off = 0;
  for (int r = 0; r < c->world; r++) {
    if (counts_out[r] > 0) {
      void* dst = static_cast<char*>(recv_dev) + off;
      RCCLCHK(h, a, a->Broadcast(r == c->rank ? send_dev : dst, dst, (size_t)counts_out[r], AUVP_NCCL_UINT8, r, c->comm, h->stream));
    }
    off += counts_out[r];
  }
  RCCLCHK(h, a, a->GroupEnd());
  HIPCHK(h, hipEventRecord(c->g1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, c->g0, c->g1));
  c->last_ms = ms;
  return AUVP_OK;
}

double auvp_last_gather_ms(auvp_handle* h) {
  CommState* c = h ? comm_of(h) : nullptr;
  return c ? c->last_ms : -1.0;
}

}  // extern "C"
#endif
