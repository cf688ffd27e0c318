// mock_rccl.cpp -- TEST INFRASTRUCTURE.  A stand-in for librccl that moves bytes between PROCESSES ON ONE GPU (or anywhere) over
// Unix sockets, so that the multi-rank code of auv_sim_amd/csrc/gather_host.h -- counts, prefix offsets, root / non-root
// branches, grouped broadcasts, grouped ncclSend / ncclRecv, the gather stream and its events -- runs with world size > 1 on the
// one-GPU boxes of this pool.  RCCL itself refuses two ranks on one device; the real library is exercised at world size 1
// (tests/test_gpu_gather_rccl.py) and its entry points' signatures are checked against rccl.h (tests/test_rccl_abi.py).
//
// Loaded through AUVP_RCCL_LIBRARY (gather_host.h binds RCCL by dlopen + dlsym).  Implements exactly the entry points that
// file binds, with the semantics it relies on:
//   * every operation is ordered after the work already on `stream` and is complete when the call (or ncclGroupEnd) returns:
//     the mock synchronises the stream, stages through host memory, copies back with hipMemcpy -- a legal (if slow)
//     implementation of "enqueued on the stream";
//   * between ncclGroupStart and ncclGroupEnd operations are only recorded; ncclGroupEnd runs them: all sends on a helper
//     thread, all receives on the calling thread, so that grouped sends and receives between any ranks cannot deadlock;
//   * operations between one pair of ranks match in issue order (one FIFO socket per pair).
// Build: hipcc -shared -fPIC -o libmock_rccl.so mock_rccl.cpp -lpthread   (tests/test_gpu_gather_mock_rccl.py does it)
#include <hip/hip_runtime_api.h>
#include <errno.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <thread>
#include <vector>

namespace {
struct Comm {
  int n = 0, rank = 0;
  std::vector<int> fd;  // fd[p]: the socket to rank p (-1 for itself)
  int listen_fd = -1;
  std::string listen_path;
};
struct Op {
  int kind;  // 0 send, 1 recv, 2 broadcast, 3 allgather
  const void* src; void* dst; size_t bytes; int peer; Comm* c; hipStream_t st;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t dt_size(int dt) {
  switch (dt) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 1; }
}
bool write_all(int fd, const void* p, size_t n) {
  const char* b = static_cast<const char*>(p);
  while (n) { ssize_t k = ::write(fd, b, n); if (k <= 0) { if (errno == EINTR) continue; return false; } b += k; n -= (size_t)k; }
  return true;
}
bool read_all(int fd, void* p, size_t n) {
  char* b = static_cast<char*>(p);
  while (n) { ssize_t k = ::read(fd, b, n); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; } b += k; n -= (size_t)k; }
  return true;
}
std::string path_of(const char* id, int rank) {
  char buf[200];
  snprintf(buf, sizeof buf, "/tmp/mock_rccl_%.40s_%d", id, rank);
  return buf;
}

// run a batch of operations: device -> host for everything that leaves, sends on a helper thread, receives here, host -> device
int run_ops(std::vector<Op>& ops) {
  if (ops.empty()) return 0;
  for (auto& o : ops) if (hipStreamSynchronize(o.st) != hipSuccess) return 1;
  struct Out { int fd; std::vector<char> buf; };
  struct In { int fd; void* dst; size_t bytes; std::vector<char> buf; };
  std::vector<Out> outs;
  std::vector<In> ins;
  for (auto& o : ops) {
    Comm* c = o.c;
    auto stage = [&](const void* dev, size_t n) { std::vector<char> h(n); if (n && hipMemcpy(h.data(), dev, n, hipMemcpyDeviceToHost) != hipSuccess) h.clear(); return h; };
    if (o.kind == 0) { outs.push_back({c->fd[o.peer], stage(o.src, o.bytes)}); }
    else if (o.kind == 1) { ins.push_back({c->fd[o.peer], o.dst, o.bytes, {}}); }
    else if (o.kind == 2) {  // broadcast from o.peer
      if (c->rank == o.peer) {
        std::vector<char> h = stage(o.src, o.bytes);
        for (int p = 0; p < c->n; p++) if (p != c->rank) outs.push_back({c->fd[p], h});
        if (o.dst != o.src && o.bytes && hipMemcpy(o.dst, o.src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
      } else ins.push_back({c->fd[o.peer], o.dst, o.bytes, {}});
    } else {  // all-gather: block r of dst <- rank r's src
      std::vector<char> h = stage(o.src, o.bytes);
      for (int p = 0; p < c->n; p++) {
        char* slot = static_cast<char*>(o.dst) + (size_t)p * o.bytes;
        if (p == c->rank) { if (o.bytes && hipMemcpy(slot, o.src, o.bytes, hipMemcpyDeviceToDevice) != hipSuccess) return 1; }
        else { outs.push_back({c->fd[p], h}); ins.push_back({c->fd[p], slot, o.bytes, {}}); }
      }
    }
  }
  bool send_ok = true;
  std::thread sender([&] { for (auto& o : outs) if (!o.buf.empty() && !write_all(o.fd, o.buf.data(), o.buf.size())) send_ok = false; });
  bool recv_ok = true;
  for (auto& i : ins) { i.buf.resize(i.bytes); if (i.bytes && !read_all(i.fd, i.buf.data(), i.bytes)) recv_ok = false; }
  sender.join();
  if (!send_ok || !recv_ok) return 1;
  for (auto& i : ins) if (i.bytes && hipMemcpy(i.dst, i.buf.data(), i.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
  return 0;
}
int submit(const Op& o) {
  if (g_depth > 0) { g_ops.push_back(o); return 0; }
  std::vector<Op> one{o};
  return run_ops(one);
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } mockUniqueId;

int ncclGetUniqueId(mockUniqueId* id) {
  memset(id->internal, 0, sizeof id->internal);
  struct timespec ts; clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof id->internal, "%d_%lld_%ld", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec);
  return 0;
}

// full mesh: rank r listens; it connects to every lower rank (sending its own rank first) and accepts from every higher one
int ncclCommInitRank(Comm** out, int nranks, mockUniqueId id, int rank) {
  Comm* c = new Comm();
  c->n = nranks; c->rank = rank; c->fd.assign(nranks, -1);
  id.internal[127] = 0;
  c->listen_path = path_of(id.internal, rank);
  if (rank < nranks - 1) {
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a; memset(&a, 0, sizeof a); a.sun_family = AF_UNIX; strncpy(a.sun_path, c->listen_path.c_str(), sizeof a.sun_path - 1);
    unlink(a.sun_path);
    if (bind(c->listen_fd, (sockaddr*)&a, sizeof a) != 0 || listen(c->listen_fd, nranks) != 0) return 2;
  }
  for (int p = 0; p < rank; p++) {
    const std::string pp = path_of(id.internal, p);
    int fd = -1;
    for (int tries = 0; tries < 6000; tries++) {  // the lower rank may not be listening yet
      fd = socket(AF_UNIX, SOCK_STREAM, 0);
      sockaddr_un a; memset(&a, 0, sizeof a); a.sun_family = AF_UNIX; strncpy(a.sun_path, pp.c_str(), sizeof a.sun_path - 1);
      if (connect(fd, (sockaddr*)&a, sizeof a) == 0) break;
      close(fd); fd = -1; usleep(10000);
    }
    if (fd < 0) return 2;
    int32_t me = rank;
    if (!write_all(fd, &me, sizeof me)) return 2;
    c->fd[p] = fd;
  }
  for (int k = rank + 1; k < nranks; k++) {
    int fd = accept(c->listen_fd, nullptr, nullptr);
    int32_t who = -1;
    if (fd < 0 || !read_all(fd, &who, sizeof who) || who <= rank || who >= nranks) return 2;
    c->fd[who] = fd;
  }
  *out = c;
  return 0;
}
int ncclCommDestroy(Comm* c) {
  if (!c) return 0;
  for (int fd : c->fd) if (fd >= 0) close(fd);
  if (c->listen_fd >= 0) { close(c->listen_fd); unlink(c->listen_path.c_str()); }
  delete c;
  return 0;
}
int ncclCommCount(Comm* c, int* n) { *n = c->n; return 0; }
int ncclCommUserRank(Comm* c, int* r) { *r = c->rank; return 0; }
const char* ncclGetErrorString(int e) { return e == 0 ? "success" : (e == 2 ? "mock rccl: rendezvous failed" : "mock rccl: transfer failed"); }
int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() {
  if (g_depth <= 0) return 1;
  if (--g_depth > 0) return 0;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return run_ops(ops);
}
int ncclAllGather(const void* s, void* r, size_t count, int dt, Comm* c, hipStream_t st) { return submit({3, s, r, count * dt_size(dt), -1, c, st}); }
int ncclBroadcast(const void* s, void* r, size_t count, int dt, int root, Comm* c, hipStream_t st) { return submit({2, s, r, count * dt_size(dt), root, c, st}); }
int ncclSend(const void* s, size_t count, int dt, int peer, Comm* c, hipStream_t st) { return submit({0, s, nullptr, count * dt_size(dt), peer, c, st}); }
int ncclRecv(void* r, size_t count, int dt, int peer, Comm* c, hipStream_t st) { return submit({1, nullptr, r, count * dt_size(dt), peer, c, st}); }
}
