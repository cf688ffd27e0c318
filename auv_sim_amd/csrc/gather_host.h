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

#include <mutex>

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
  int (*CommCount)(auvp_ncclComm_t, int*) = nullptr;
  int (*CommUserRank)(auvp_ncclComm_t, int*) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, auvp_ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, auvp_ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, auvp_ncclComm_t, hipStream_t) = nullptr;  // optional (gather to a root)
  int (*Recv)(void*, size_t, int, int, auvp_ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string err, bound;
};

// An RCCL that is ALREADY mapped into the process (PyTorch-ROCm bundles one as torch/lib/librccl.so) is bound first, by
// the path /proc/self/maps shows and with RTLD_NOLOAD, so a torch process keeps its one RCCL image; only a process without
// any opens one.  AUVP_RCCL_LIBRARY overrides everything.  `bound` records what was bound (auvp_comm_library).
std::string mapped_rccl_path() {
  FILE* f = fopen("/proc/self/maps", "r");
  if (!f) return "";
  char line[4096];
  std::string found;
  while (fgets(line, sizeof line, f)) {
    const char* p = strchr(line, '/');
    if (!p) continue;
    std::string path(p);
    while (!path.empty() && (path.back() == '\n' || path.back() == ' ')) path.pop_back();
    const size_t k = path.rfind('/');
    const std::string base = k == std::string::npos ? path : path.substr(k + 1);
    if (base.rfind("librccl.so", 0) == 0) { found = path; break; }
  }
  fclose(f);
  return found;
}

void rccl_api_fill(RcclApi& api) {
  const char* env = getenv("AUVP_RCCL_LIBRARY");
  std::string last_err;
  auto try_open = [&](const char* n, int flags) {
    if (api.so || !n || !*n) return;
    (void)dlerror();
    api.so = dlopen(n, flags);
    if (api.so) { api.bound = n; return; }
    const char* e = dlerror();  // glibc clears the message on the first call: read it exactly once
    if (e && !(flags & RTLD_NOLOAD)) last_err = e;
  };
  if (env && *env) {
    try_open(env, RTLD_NOW | RTLD_GLOBAL);
  } else {
    const std::string mapped = mapped_rccl_path();
    if (!mapped.empty()) try_open(mapped.c_str(), RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) try_open(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    for (const char* n : names) try_open(n, RTLD_NOW | RTLD_GLOBAL);
  }
  if (!api.so) { api.err = std::string("cannot load RCCL: ") + (last_err.empty() ? "no library found" : last_err); return; }
  auto sym = [&](const char* s) { void* p = dlsym(api.so, s); if (!p && api.err.empty()) api.err = std::string("RCCL symbol missing: ") + s; return p; };
  api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
  api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
  api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
  api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
  api.CommUserRank = reinterpret_cast<decltype(api.CommUserRank)>(sym("ncclCommUserRank"));
  api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
  api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
  api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
  api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
  api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  if (!api.err.empty()) { dlclose(api.so); api.so = nullptr; return; }
  // point-to-point (RCCL >= 2.7): only the gather to a root needs them, their absence is reported by that call
  api.Send = reinterpret_cast<decltype(api.Send)>(dlsym(api.so, "ncclSend"));
  api.Recv = reinterpret_cast<decltype(api.Recv)>(dlsym(api.so, "ncclRecv"));
  Dl_info di;
  if (dladdr(reinterpret_cast<void*>(api.AllGather), &di) && di.dli_fname) api.bound = di.dli_fname;
}

RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] { rccl_api_fill(api); });
  return &api;
}

struct CommState {
  auvp_ncclComm_t comm = nullptr;
  int world = 0, rank = 0;
  DevBuf counts;  // [world][2] int64: byte count and receive capacity of every rank (variable-length gather)
  hipEvent_t g0 = nullptr, g1 = nullptr;
  double last_ms = 0.0;
  // the gather to a root runs on a stream of its own, ordered behind the planner stream by an event: the transfer of step k
  // overlaps the kernels of step k + 1 (auvp_gather_blocks_root_async / auvp_gather_wait)
  hipStream_t gstream = nullptr;
  hipEvent_t ready = nullptr, a0 = nullptr, a1 = nullptr;
  bool pending = false;        // enqueued since the last auvp_gather_wait
  int64_t pending_bytes = 0;   // what this rank sends (or, on the root, receives) in the pending transfers
  ~CommState() {
    if (gstream) (void)hipStreamSynchronize(gstream);
    if (comm) { RcclApi* a = rccl_api(); if (a->so) (void)a->CommDestroy(comm); }
    if (g0) (void)hipEventDestroy(g0);
    if (g1) (void)hipEventDestroy(g1);
    if (ready) (void)hipEventDestroy(ready);
    if (a0) (void)hipEventDestroy(a0);
    if (a1) (void)hipEventDestroy(a1);
    if (gstream) (void)hipStreamDestroy(gstream);
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
  HIPCHK(h, hipStreamCreateWithFlags(&c->gstream, hipStreamNonBlocking));
  HIPCHK(h, hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
  HIPCHK(h, hipEventCreate(&c->a0));
  HIPCHK(h, hipEventCreate(&c->a1));
  HIPCHK(h, c->counts.reserve((size_t)world_size * 2 * sizeof(int64_t)));
  return AUVP_OK;
}

int auvp_comm_destroy(auvp_handle* h) {
  if (!h) return AUVP_ERR_ARG;
  if (h->comm) {
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (comm_of(h)->gstream) (void)hipStreamSynchronize(comm_of(h)->gstream);
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

// phase 1 of the variable-length gather: every rank's byte count AND what every rank can receive (-1: that rank passed
// no buffer), so that the decision to run phase 2 is the same on every rank -- a rank that returned early while the
// others entered the grouped broadcasts would leave them waiting
static int gather_counts_impl(auvp_handle* h, CommState* c, RcclApi* a, int64_t send_bytes, int64_t recv_cap, int64_t* counts_out,
                              bool* all_can_receive) {
  int64_t mine[2] = {send_bytes, recv_cap};
  HIPCHK(h, h->d_tmp5.reserve(2 * sizeof(int64_t)));
  HIPCHK(h, hipMemcpyAsync(h->d_tmp5.p, mine, sizeof mine, hipMemcpyHostToDevice, h->stream));
  RCCLCHK(h, a, a->AllGather(h->d_tmp5.p, c->counts.p, 2, AUVP_NCCL_INT64, c->comm, h->stream));
  std::vector<int64_t> both((size_t)c->world * 2);
  HIPCHK(h, hipMemcpyAsync(both.data(), c->counts.p, both.size() * sizeof(int64_t), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  int64_t total = 0;
  for (int r = 0; r < c->world; r++) { counts_out[r] = both[2 * (size_t)r]; total += counts_out[r]; }
  bool ok = true;
  for (int r = 0; r < c->world; r++) ok = ok && both[2 * (size_t)r + 1] >= total;
  if (all_can_receive) *all_can_receive = ok;
  return AUVP_OK;
}

// phase 2: rank r's block lands at the prefix offset of r -- one grouped broadcast per rank, no padding to the largest
// block.  The group is always closed, also when a broadcast inside it failed.
static int gather_blocks_impl(auvp_handle* h, CommState* c, RcclApi* a, const void* send_dev, void* recv_dev, const int64_t* counts) {
  int rc_group = a->GroupStart();
  if (rc_group != 0) return fail(h, AUVP_ERR_COMM, "ncclGroupStart: %s", a->GetErrorString ? a->GetErrorString(rc_group) : "?");
  int first_err = 0;
  int64_t off = 0;
  for (int r = 0; r < c->world; r++) {
    if (counts[r] > 0 && first_err == 0) {
      void* dst = static_cast<char*>(recv_dev) + off;
      first_err = a->Broadcast(r == c->rank ? send_dev : dst, dst, (size_t)counts[r], AUVP_NCCL_UINT8, r, c->comm, h->stream);
    }
    off += counts[r];
  }
  const int end_err = a->GroupEnd();
  if (first_err != 0 || end_err != 0) {
    (void)hipStreamSynchronize(h->stream);
    const int e = first_err ? first_err : end_err;
    return fail(h, AUVP_ERR_COMM, "grouped ncclBroadcast: %s", a->GetErrorString ? a->GetErrorString(e) : "?");
  }
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
  bool all_ok = false;
  int rc = gather_counts_impl(h, c, a, send_bytes, recv_dev ? recv_cap_bytes : -1, counts_out, &all_ok);
  if (rc != AUVP_OK) return rc;
  if (!all_ok) {
    // sizes are reported (two-phase convention): the caller allocates sum(counts_out) bytes and calls again.  No rank
    // enters phase 2 unless EVERY rank passed a buffer that holds the total.
    if (!recv_dev) return AUVP_OK;
    int64_t total = 0;
    for (int r = 0; r < c->world; r++) total += counts_out[r];
    return fail(h, AUVP_ERR_CAPACITY, "gather needs %lld bytes on every rank; this rank's buffer holds %lld (or another rank's is too small)",
                (long long)total, (long long)recv_cap_bytes);
  }
  rc = gather_blocks_impl(h, c, a, send_dev, recv_dev, counts_out);
  if (rc != AUVP_OK) return rc;
  HIPCHK(h, hipEventRecord(c->g1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, c->g0, c->g1));
  c->last_ms = ms;
  return AUVP_OK;
}

int auvp_gather_counts(auvp_handle* h, int64_t send_bytes, int64_t* counts_out) {
  if (!h || send_bytes < 0 || !counts_out) return AUVP_ERR_ARG;
  CommState* c = comm_of(h);
  if (!c || !c->comm) return fail(h, AUVP_ERR_STATE, "auvp_comm_init not called");
  RcclApi* a = rccl_api();
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(c->g0, h->stream));
  int rc = gather_counts_impl(h, c, a, send_bytes, -1, counts_out, nullptr);
  if (rc != AUVP_OK) return rc;
  HIPCHK(h, hipEventRecord(c->g1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, c->g0, c->g1));
  c->last_ms = ms;
  return AUVP_OK;
}

int auvp_gather_blocks(auvp_handle* h, const void* send_dev, void* recv_dev, int64_t recv_cap_bytes, const int64_t* counts) {
  if (!h || !counts || !recv_dev) return AUVP_ERR_ARG;
  CommState* c = comm_of(h);
  if (!c || !c->comm) return fail(h, AUVP_ERR_STATE, "auvp_comm_init not called");
  int64_t total = 0;
  for (int r = 0; r < c->world; r++) { if (counts[r] < 0) return fail(h, AUVP_ERR_ARG, "negative count"); total += counts[r]; }
  // counts are the same on every rank (auvp_gather_counts) and the caller sized recv_dev from them: a short buffer here is
  // a caller bug on THIS rank, reported before any collective is entered by anyone only if every rank checks the same sum
  if (total > recv_cap_bytes) return fail(h, AUVP_ERR_CAPACITY, "gather needs %lld bytes, buffer holds %lld", (long long)total, (long long)recv_cap_bytes);
  if (counts[c->rank] > 0 && !send_dev) return AUVP_ERR_ARG;
  RcclApi* a = rccl_api();
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(c->g0, h->stream));
  int rc = gather_blocks_impl(h, c, a, send_dev, recv_dev, counts);
  if (rc != AUVP_OK) return rc;
  HIPCHK(h, hipEventRecord(c->g1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, c->g0, c->g1));
  c->last_ms = ms;
  return AUVP_OK;
}

// ---- gather to ONE rank (north_star: "an RCCL gather of final paths"): ncclSend / ncclRecv in one group ----
// rank r's block lands at the prefix offset of r in the ROOT's buffer; the other ranks receive nothing.  On a fully connected
// xGMI node every peer has its own link to the root, so the seven transfers run side by side (an all-gather moves seven
// times the bytes into EVERY rank).  Enqueues on `stream`; no host synchronisation.
static int gather_root_enqueue(auvp_handle* h, CommState* c, RcclApi* a, int root, const void* send_dev, void* recv_dev,
                               const int64_t* counts, hipStream_t stream) {
  if (c->world > 1 && (!a->Send || !a->Recv)) return fail(h, AUVP_ERR_COMM, "this RCCL has no ncclSend / ncclRecv");
  const bool is_root = c->rank == root;
  int64_t my_off = 0;
  for (int r = 0; r < c->rank; r++) my_off += counts[r];
  if (is_root && counts[root] > 0)  // the root's own block: a device copy on the same stream
    HIPCHK(h, hipMemcpyAsync(static_cast<char*>(recv_dev) + my_off, send_dev, (size_t)counts[root], hipMemcpyDeviceToDevice, stream));
  if (c->world == 1) return AUVP_OK;
  int rc_group = a->GroupStart();
  if (rc_group != 0) return fail(h, AUVP_ERR_COMM, "ncclGroupStart: %s", a->GetErrorString ? a->GetErrorString(rc_group) : "?");
  int first_err = 0;
  if (is_root) {
    int64_t off = 0;
    for (int r = 0; r < c->world; r++) {
      if (r != root && counts[r] > 0 && first_err == 0)
        first_err = a->Recv(static_cast<char*>(recv_dev) + off, (size_t)counts[r], AUVP_NCCL_UINT8, r, c->comm, stream);
      off += counts[r];
    }
  } else if (counts[c->rank] > 0) {
    first_err = a->Send(send_dev, (size_t)counts[c->rank], AUVP_NCCL_UINT8, root, c->comm, stream);
  }
  const int end_err = a->GroupEnd();
  if (first_err != 0 || end_err != 0) {
    (void)hipStreamSynchronize(stream);
    const int e = first_err ? first_err : end_err;
    return fail(h, AUVP_ERR_COMM, "grouped ncclSend / ncclRecv: %s", a->GetErrorString ? a->GetErrorString(e) : "?");
  }
  return AUVP_OK;
}

static int gather_root_check(auvp_handle* h, CommState* c, int32_t root, const void* send_dev, void* recv_dev, int64_t recv_cap_bytes,
                             const int64_t* counts) {
  if (!c || !c->comm) return fail(h, AUVP_ERR_STATE, "auvp_comm_init not called");
  if (root < 0 || root >= c->world) return fail(h, AUVP_ERR_ARG, "root %d outside the communicator", (int)root);
  int64_t total = 0;
  for (int r = 0; r < c->world; r++) { if (counts[r] < 0) return fail(h, AUVP_ERR_ARG, "negative count"); total += counts[r]; }
  if (counts[c->rank] > 0 && !send_dev) return AUVP_ERR_ARG;
  // (only the root's buffer matters; a short one is the root's own caller bug, reported before it posts any receive -- the
  // senders' transfers then stay unmatched until the communicator is destroyed, as with any rank that drops out of a collective)
  if (c->rank == root && total > 0 && (!recv_dev || total > recv_cap_bytes))
    return fail(h, AUVP_ERR_CAPACITY, "the root needs %lld bytes, its buffer holds %lld", (long long)total, (long long)(recv_dev ? recv_cap_bytes : 0));
  return AUVP_OK;
}

int auvp_gather_blocks_root_async(auvp_handle* h, int32_t root, const void* send_dev, void* recv_dev, int64_t recv_cap_bytes,
                                  const int64_t* counts) {
  if (!h || !counts) return AUVP_ERR_ARG;
  CommState* c = comm_of(h);
  int rc = gather_root_check(h, c, root, send_dev, recv_dev, recv_cap_bytes, counts);
  if (rc != AUVP_OK) return rc;
  RcclApi* a = rccl_api();
  HIPCHK(h, hipSetDevice(h->device));
  // behind everything the planner stream holds so far (the kernels and copies that produced send_dev)
  HIPCHK(h, hipEventRecord(c->ready, h->stream));
  HIPCHK(h, hipStreamWaitEvent(c->gstream, c->ready, 0));
  if (!c->pending) { HIPCHK(h, hipEventRecord(c->a0, c->gstream)); c->pending_bytes = 0; }
  c->pending = true;
  rc = gather_root_enqueue(h, c, a, root, send_dev, recv_dev, counts, c->gstream);
  if (rc != AUVP_OK) return rc;
  if (c->rank == root) { for (int r = 0; r < c->world; r++) c->pending_bytes += counts[r]; }
  else c->pending_bytes += counts[c->rank];
  return AUVP_OK;
}

int auvp_gather_wait(auvp_handle* h, double* ms_out, int64_t* bytes_out) {
  if (!h) return AUVP_ERR_ARG;
  CommState* c = comm_of(h);
  if (!c || !c->comm) return fail(h, AUVP_ERR_STATE, "auvp_comm_init not called");
  if (ms_out) *ms_out = 0.0;
  if (bytes_out) *bytes_out = 0;
  if (!c->pending) return AUVP_OK;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipEventRecord(c->a1, c->gstream));
  HIPCHK(h, hipStreamSynchronize(c->gstream));
  float ms = 0.f;
  HIPCHK(h, hipEventElapsedTime(&ms, c->a0, c->a1));
  c->last_ms = ms;
  c->pending = false;
  if (ms_out) *ms_out = ms;
  if (bytes_out) *bytes_out = c->pending_bytes;
  return AUVP_OK;
}

int auvp_gather_blocks_root(auvp_handle* h, int32_t root, const void* send_dev, void* recv_dev, int64_t recv_cap_bytes,
                            const int64_t* counts) {
  int rc = auvp_gather_blocks_root_async(h, root, send_dev, recv_dev, recv_cap_bytes, counts);
  if (rc != AUVP_OK) return rc;
  return auvp_gather_wait(h, nullptr, nullptr);
}

int auvp_comm_available(void) { return rccl_api()->so ? 1 : 0; }

const char* auvp_comm_library(void) {
  RcclApi* a = rccl_api();
  return a->so ? a->bound.c_str() : a->err.c_str();
}

int auvp_comm_info(auvp_handle* h, int32_t* world_size, int32_t* rank, int32_t* rccl_ranks_seen) {
  if (!h) return AUVP_ERR_ARG;
  CommState* c = comm_of(h);
  if (!c || !c->comm) return fail(h, AUVP_ERR_STATE, "auvp_comm_init not called");
  RcclApi* a = rccl_api();
  int n = -1, r = -1;
  RCCLCHK(h, a, a->CommCount(c->comm, &n));
  RCCLCHK(h, a, a->CommUserRank(c->comm, &r));
  if (world_size) *world_size = c->world;
  if (rank) *rank = r;
  if (rccl_ranks_seen) *rccl_ranks_seen = n;
  return AUVP_OK;
}

double auvp_last_gather_ms(auvp_handle* h) {
  CommState* c = h ? comm_of(h) : nullptr;
  return c ? c->last_ms : -1.0;
}

}  // extern "C"
#endif
