// Multi-GPU layer of the C ABI: a loop-closure batch is sharded over the ranks of one node (one process per GPU), and the
// ONLY exchange on the data path is one gather of the resulting se(3) poses per batch — 8 floats per alignment
// [pose(6), weightedPose, iterations] (the batch loop it shards: GlobalOptimize.cpp:480-610).
//
// Two transports behind the same entry points:
//   RCCL  (ELLC_WITH_RCCL, part of libellc_hip.so): ncclAllGather over xGMI on a stream of its own, preallocated device
//         buffers, pinned staging; up to ELLC_GATHER_DEPTH gathers may be outstanding (start / finish), so the exchange of
//         batch s overlaps the kernels of the batches behind it.
//   TCP   (always; libellc_comm.so is this file alone, built with g++): rank 0 listens on host:port, the others connect;
//         a gather is a star exchange through rank 0. Host memory only — for the CPU tests of the sharded path and for
//         hosts whose GPUs are not connected by xGMI. Same records, same order, same padding rule as the RCCL form.
#include "../../include/ellc_abi.h"
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/time.h>
#include <sys/socket.h>
#include <unistd.h>
#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#ifdef ELLC_WITH_RCCL
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#endif

#define ELLC_GATHER_DEPTH 4
#define ELLC_RECORD 8

struct ellc_comm {
  int transport = 0;   // 1 RCCL, 2 TCP
  int world = 1, rank = 0, max_total = 0, per_max = 0;
  int world_seen = 0, rank_seen = -1;   // what the TRANSPORT reports (ellc_comm_info): ncclCommCount / ncclCommUserRank; TCP: the ranks that joined rank 0
  std::string pci_bus_id;               // RCCL: hipDeviceGetPCIBusId of the communicator's device
  std::string err;
  // ring of outstanding gathers
  struct Slot {
    int total = 0, per = 0;
    std::vector<float> table;      // TCP: the gathered table
#ifdef ELLC_WITH_RCCL
    float *host_in = nullptr, *host_out = nullptr, *dev_in = nullptr, *dev_out = nullptr;
    hipEvent_t done = nullptr;
#endif
  } slot[ELLC_GATHER_DEPTH];
  int head = 0, pending = 0;       // oldest outstanding slot, number outstanding
  bool dead = false;               // a TCP exchange failed half-way: the streams are out of step, every later gather is refused
  // TCP
  int listen_fd = -1;
  std::vector<int> peer;           // rank 0: socket of every other rank (index = rank); others: [0] = socket to rank 0
#ifdef ELLC_WITH_RCCL
  int device = 0;
  ncclComm_t nccl = nullptr;
  hipStream_t stream = nullptr;
#endif
};

namespace {

ellc_status cfail(ellc_comm* c, ellc_status s, const std::string& msg) {
  if (c) c->err = msg;
  return s;
}

bool send_all(int fd, const void* p, size_t n) {
  const char* q = (const char*)p;
  while (n > 0) {
    const ssize_t k = ::send(fd, q, n, MSG_NOSIGNAL);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    q += k;
    n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void* p, size_t n) {
  char* q = (char*)p;
  while (n > 0) {
    const ssize_t k = ::recv(fd, q, n, 0);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    q += k;
    n -= (size_t)k;
  }
  return true;
}

// every blocking socket call of the communicator has a deadline: a peer that died (or a stray connection) must not hang a rank
const int kTimeoutSeconds = 60;
void set_socket_timeouts(int fd) {
  timeval tv;
  tv.tv_sec = kTimeoutSeconds;
  tv.tv_usec = 0;
  ::setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  ::setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

}  // namespace

extern "C" {

// contiguous block partition: alignment b lives on rank b / ceil(total / world)
void ellc_shard_range(int total, int world, int rank, int* lo, int* hi) {
  if (world < 1) world = 1;
  const int per = (std::max(total, 0) + world - 1) / world;
  const int l = std::min(std::max(total, 0), rank * per);
  const int h = std::min(std::max(total, 0), l + per);
  if (lo) *lo = l;
  if (hi) *hi = h;
}

const char* ellc_comm_last_error(const ellc_comm* c) { return c ? c->err.c_str() : "null communicator"; }

ellc_status ellc_comm_init_tcp(const char* host, int port, int world, int rank, int max_total, ellc_comm** out) {
  if (!out) return ELLC_ERR_BAD_ARG;
  *out = nullptr;
  if (!host || port <= 0 || port > 65535 || world < 1 || rank < 0 || rank >= world || max_total < 1) return ELLC_ERR_BAD_ARG;
  ellc_comm* c = new ellc_comm();
  c->transport = 2;
  c->world = world; c->rank = rank; c->max_total = max_total;
  c->per_max = (max_total + world - 1) / world;
  *out = c;
  c->world_seen = 1; c->rank_seen = rank;
  if (world == 1) return ELLC_OK;
  sockaddr_in addr;
  std::memset(&addr, 0, sizeof(addr));
  addr.sin_family = AF_INET;
  addr.sin_port = htons((uint16_t)port);
  if (::inet_pton(AF_INET, host, &addr.sin_addr) != 1) return cfail(c, ELLC_ERR_BAD_ARG, "ellc_comm_init_tcp: host must be a dotted IPv4 address");
  const int one = 1;
  if (rank == 0) {
    c->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (c->listen_fd < 0) return cfail(c, ELLC_ERR_HIP, "socket() failed");
    ::setsockopt(c->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (::bind(c->listen_fd, (sockaddr*)&addr, sizeof(addr)) != 0 || ::listen(c->listen_fd, world) != 0)
      return cfail(c, ELLC_ERR_HIP, std::string("bind/listen failed: ") + std::strerror(errno));
    c->peer.assign(world, -1);
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(kTimeoutSeconds);
    for (int k = 1; k < world;) {
      const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
      pollfd pf;
      pf.fd = c->listen_fd; pf.events = POLLIN; pf.revents = 0;
      const int pr = left > 0 ? ::poll(&pf, 1, (int)left) : 0;
      if (pr < 0 && errno == EINTR) continue;
      if (pr <= 0) return cfail(c, ELLC_ERR_HIP, "ellc_comm_init_tcp: not every rank connected within 60 s");
      const int fd = ::accept(c->listen_fd, nullptr, nullptr);
      if (fd < 0) return cfail(c, ELLC_ERR_HIP, "accept() failed");
      ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      set_socket_timeouts(fd);
      int r = -1;
      if (!recv_all(fd, &r, sizeof(r)) || r < 1 || r >= world || c->peer[r] != -1) {
        ::close(fd);   // a stray connection, or a rank announced twice: not one of ours — keep waiting for the real peers
        continue;
      }
      c->peer[r] = fd;
      k++;
    }
    // every peer learns how many ranks joined (its own view of the world: ellc_comm_info)
    int joined = 1;
    for (int r = 1; r < world; r++) joined += c->peer[r] >= 0 ? 1 : 0;
    c->world_seen = joined;
    for (int r = 1; r < world; r++)
      if (!send_all(c->peer[r], &joined, sizeof(joined))) return cfail(c, ELLC_ERR_HIP, "cannot acknowledge a rank");
  } else {
    int fd = -1;
    for (int attempt = 0; attempt < 600; attempt++) {   // rank 0 may not be listening yet: retry for up to a minute
      fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd < 0) return cfail(c, ELLC_ERR_HIP, "socket() failed");
      if (::connect(fd, (sockaddr*)&addr, sizeof(addr)) == 0) break;
      ::close(fd);
      fd = -1;
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (fd < 0) return cfail(c, ELLC_ERR_HIP, "cannot connect to rank 0");
    ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    set_socket_timeouts(fd);
    if (!send_all(fd, &rank, sizeof(rank))) return cfail(c, ELLC_ERR_HIP, "cannot announce the rank");
    c->peer.assign(1, fd);
    int joined = 0;
    if (!recv_all(fd, &joined, sizeof(joined))) return cfail(c, ELLC_ERR_HIP, "rank 0 did not acknowledge this rank (not every rank joined within 60 s?)");
    c->world_seen = joined;
  }
  if (c->world_seen != world) return cfail(c, ELLC_ERR_HIP, "ellc_comm_init_tcp: " + std::to_string(c->world_seen) + " ranks joined, " + std::to_string(world) + " expected");
  return ELLC_OK;
}

// What the transport itself saw: the self-check of a multi-rank run (bench.py carries it in the N > 1 line).
ellc_status ellc_comm_info(const ellc_comm* c, int* transport, int* world_seen, int* rank_seen, char* pci_bus_id, int pci_capacity) {
  if (!c) return ELLC_ERR_BAD_ARG;
  if (transport) *transport = c->transport;
  if (world_seen) *world_seen = c->world_seen;
  if (rank_seen) *rank_seen = c->rank_seen;
  if (pci_bus_id && pci_capacity > 0) {
    std::strncpy(pci_bus_id, c->pci_bus_id.c_str(), (size_t)pci_capacity - 1);
    pci_bus_id[pci_capacity - 1] = 0;
  }
  return ELLC_OK;
}

#ifdef ELLC_WITH_RCCL
ellc_status ellc_comm_unique_id(unsigned char* id128) {
  if (!id128) return ELLC_ERR_BAD_ARG;
  static_assert(sizeof(ncclUniqueId) <= 128, "unique id does not fit the ABI's 128 bytes");
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return ELLC_ERR_HIP;
  std::memset(id128, 0, 128);
  std::memcpy(id128, &id, sizeof(id));
  return ELLC_OK;
}

ellc_status ellc_comm_init_rccl(int device, const unsigned char* id128, int world, int rank, int max_total, ellc_comm** out) {
  if (!out) return ELLC_ERR_BAD_ARG;
  *out = nullptr;
  if (!id128 || world < 1 || rank < 0 || rank >= world || max_total < 1) return ELLC_ERR_BAD_ARG;
  ellc_comm* c = new ellc_comm();
  c->transport = 1;
  c->world = world; c->rank = rank; c->max_total = max_total; c->device = device;
  c->per_max = (max_total + world - 1) / world;
  *out = c;
  if (hipSetDevice(device) != hipSuccess) return cfail(c, ELLC_ERR_HIP, "cannot make the device current");
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  const ncclResult_t r = ncclCommInitRank(&c->nccl, world, id, rank);
  if (r != ncclSuccess) return cfail(c, ELLC_ERR_HIP, std::string("ncclCommInitRank: ") + ncclGetErrorString(r));
  // what RCCL itself says about the communicator (a launcher that starts fewer ranks than it claims, or two ranks on one
  // device, must not pass for an N-GPU run): kept for ellc_comm_info, and checked here
  {
    int n = 0, ur = -1, dev = -1;
    if (ncclCommCount(c->nccl, &n) != ncclSuccess || ncclCommUserRank(c->nccl, &ur) != ncclSuccess || ncclCommCuDevice(c->nccl, &dev) != ncclSuccess)
      return cfail(c, ELLC_ERR_HIP, "ncclCommCount / ncclCommUserRank / ncclCommCuDevice failed");
    c->world_seen = n; c->rank_seen = ur;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), dev) == hipSuccess) c->pci_bus_id = bus;
    if (n != world || ur != rank || dev != device)
      return cfail(c, ELLC_ERR_HIP, "RCCL reports " + std::to_string(n) + " ranks / rank " + std::to_string(ur) + " / device " + std::to_string(dev) + ", expected " +
                                        std::to_string(world) + " / " + std::to_string(rank) + " / " + std::to_string(device));
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return cfail(c, ELLC_ERR_HIP, "cannot create the gather stream");
  const size_t in_b = (size_t)c->per_max * ELLC_RECORD * sizeof(float), out_b = in_b * world;
  for (auto& s : c->slot) {
    if (hipMalloc((void**)&s.dev_in, in_b) != hipSuccess || hipMalloc((void**)&s.dev_out, out_b) != hipSuccess ||
        hipHostMalloc((void**)&s.host_in, in_b, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&s.host_out, out_b, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&s.done, hipEventDisableTiming) != hipSuccess)
      return cfail(c, ELLC_ERR_HIP, "cannot allocate the gather buffers");
  }
  return ELLC_OK;
}
#endif

ellc_status ellc_comm_destroy(ellc_comm* c) {
  if (!c) return ELLC_ERR_BAD_ARG;
  for (int fd : c->peer)
    if (fd >= 0) ::close(fd);
  if (c->listen_fd >= 0) ::close(c->listen_fd);
#ifdef ELLC_WITH_RCCL
  if (c->transport == 1) {
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto& s : c->slot) {
      if (s.dev_in) (void)hipFree(s.dev_in);
      if (s.dev_out) (void)hipFree(s.dev_out);
      if (s.host_in) (void)hipHostFree(s.host_in);
      if (s.host_out) (void)hipHostFree(s.host_out);
      if (s.done) (void)hipEventDestroy(s.done);
    }
    if (c->nccl) (void)ncclCommDestroy(c->nccl);
    if (c->stream) (void)hipStreamDestroy(c->stream);
  }
#endif
  delete c;
  return ELLC_OK;
}

// Enqueue the gather of this rank's n_local records (the rank's shard of `total`, in global order); at most
// ELLC_GATHER_DEPTH may be outstanding. Every rank calls it with the same `total`.
ellc_status ellc_gather_start(ellc_comm* c, int total, const float* local8, int n_local) {
  if (!c || total < 1 || total > c->max_total || n_local < 0 || (n_local > 0 && !local8)) return cfail(c, ELLC_ERR_BAD_ARG, "ellc_gather_start: bad argument");
  if (c->dead) return cfail(c, ELLC_ERR_HIP, "ellc_gather_start: an earlier exchange failed half-way; create a new communicator");
  if (c->pending >= ELLC_GATHER_DEPTH) return cfail(c, ELLC_ERR_NOT_READY, "ellc_gather_start: too many gathers outstanding, finish one first");
  const int per = (total + c->world - 1) / c->world;
  int lo, hi;
  ellc_shard_range(total, c->world, c->rank, &lo, &hi);
  if (n_local != hi - lo) return cfail(c, ELLC_ERR_BAD_ARG, "ellc_gather_start: n_local is not this rank's shard of total (ellc_shard_range)");
  ellc_comm::Slot& s = c->slot[(c->head + c->pending) % ELLC_GATHER_DEPTH];
  s.total = total;
  s.per = per;
  const size_t rec_b = ELLC_RECORD * sizeof(float);
  if (c->transport == 2) {
    s.table.assign((size_t)c->world * per * ELLC_RECORD, 0.0f);
    std::vector<float> mine((size_t)per * ELLC_RECORD, 0.0f);
    if (n_local) std::memcpy(mine.data(), local8, (size_t)n_local * rec_b);
    if (c->world == 1) {
      std::memcpy(s.table.data(), mine.data(), mine.size() * sizeof(float));
    } else if (c->rank == 0) {
      std::memcpy(s.table.data(), mine.data(), mine.size() * sizeof(float));
      for (int r = 1; r < c->world; r++)
        if (!recv_all(c->peer[r], s.table.data() + (size_t)r * per * ELLC_RECORD, (size_t)per * rec_b)) {
          c->dead = true;
          return cfail(c, ELLC_ERR_HIP, "gather: a peer closed the connection or timed out");
        }
      for (int r = 1; r < c->world; r++)
        if (!send_all(c->peer[r], s.table.data(), s.table.size() * sizeof(float))) {
          c->dead = true;
          return cfail(c, ELLC_ERR_HIP, "gather: cannot send the table");
        }
    } else {
      if (!send_all(c->peer[0], mine.data(), mine.size() * sizeof(float)) || !recv_all(c->peer[0], s.table.data(), s.table.size() * sizeof(float))) {
        c->dead = true;
        return cfail(c, ELLC_ERR_HIP, "gather: exchange with rank 0 failed or timed out");
      }
    }
  }
#ifdef ELLC_WITH_RCCL
  else if (c->transport == 1) {
    if (hipSetDevice(c->device) != hipSuccess) return cfail(c, ELLC_ERR_HIP, "cannot make the device current");
    std::memset(s.host_in, 0, (size_t)per * rec_b);
    if (n_local) std::memcpy(s.host_in, local8, (size_t)n_local * rec_b);
    hipError_t e = hipMemcpyAsync(s.dev_in, s.host_in, (size_t)per * rec_b, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) return cfail(c, ELLC_ERR_HIP, std::string("gather H2D: ") + hipGetErrorString(e));
    const ncclResult_t r = ncclAllGather(s.dev_in, s.dev_out, (size_t)per * ELLC_RECORD, ncclFloat, c->nccl, c->stream);
    if (r != ncclSuccess) return cfail(c, ELLC_ERR_HIP, std::string("ncclAllGather: ") + ncclGetErrorString(r));
    e = hipMemcpyAsync(s.host_out, s.dev_out, (size_t)c->world * per * rec_b, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipEventRecord(s.done, c->stream);
    if (e != hipSuccess) return cfail(c, ELLC_ERR_HIP, std::string("gather D2H: ") + hipGetErrorString(e));
  }
#endif
  else {
    return cfail(c, ELLC_ERR_BAD_ARG, "unknown transport");
  }
  c->pending++;
  return ELLC_OK;
}

// Wait for the OLDEST outstanding gather; out8 (room for out_capacity records) receives its `total` records in global order.
ellc_status ellc_gather_finish(ellc_comm* c, float* out8, int out_capacity) {
  if (!c || !out8) return cfail(c, ELLC_ERR_BAD_ARG, "ellc_gather_finish: bad argument");
  if (c->pending < 1) return cfail(c, ELLC_ERR_NOT_READY, "ellc_gather_finish: no gather outstanding");
  ellc_comm::Slot& s = c->slot[c->head];
  if (out_capacity < s.total)   // gathers of different sizes may be outstanding: never write past the caller's buffer
    return cfail(c, ELLC_ERR_CAPACITY, "ellc_gather_finish: the oldest outstanding gather holds " + std::to_string(s.total) + " records, the buffer " +
                                           std::to_string(out_capacity));
  const float* table = s.table.data();
#ifdef ELLC_WITH_RCCL
  if (c->transport == 1) {
    const hipError_t e = hipEventSynchronize(s.done);
    if (e != hipSuccess) {
      c->head = (c->head + 1) % ELLC_GATHER_DEPTH;
      c->pending--;
      return cfail(c, ELLC_ERR_HIP, std::string("gather: ") + hipGetErrorString(e));
    }
    table = s.host_out;
  }
#endif
  // rank r's block holds its shard first, padding behind it: the shards are contiguous in global order
  for (int r = 0; r < c->world; r++) {
    int lo, hi;
    ellc_shard_range(s.total, c->world, r, &lo, &hi);
    if (hi > lo) std::memcpy(out8 + (size_t)lo * ELLC_RECORD, table + (size_t)r * s.per * ELLC_RECORD, (size_t)(hi - lo) * ELLC_RECORD * sizeof(float));
  }
  c->head = (c->head + 1) % ELLC_GATHER_DEPTH;
  c->pending--;
  return ELLC_OK;
}

ellc_status ellc_gather_results(ellc_comm* c, int total, const float* local8, int n_local, float* out8) {
  const ellc_status s = ellc_gather_start(c, total, local8, n_local);
  if (s != ELLC_OK) return s;
  return ellc_gather_finish(c, out8, total);
}

}  // extern "C"
