// Single-node communicator: socket rendezvous + symmetric device buffers over HIP IPC (xGMI peers).
//
// Replaces reference src/communicator/* (communicator.cc:17-193 star rendezvous over an abstract
// unix socket or tcp, channel.cc message framing, multicast_object_manager.cc:117-220 cuMem
// export/import + NVLS multicast objects, multicast_communicator.cc:50-153 CreateTensorSync).
//
// MI355X design: xGMI is a full mesh with no multicast and no in-fabric reduction, so there is
// no "multicast object" - a symmetric allocation is one uncached (MTYPE_UC, fine-grained) device
// buffer per rank, exported with hipIpcGetMemHandle and opened by every peer; every rank ends up
// with a pointer table {peer -> VA of the peer's buffer}.  Uncached memory is what makes peer
// stores visible to a spinning consumer without kernel-boundary flushes (what RCCL uses for its
// flag/LL buffers).  The handles travel through the same rank-0 star the reference uses; a
// process-wide registry maps any address inside a local symmetric buffer to the peers' addresses,
// which is how the high-throughput all-reduce entry recovers peer pointers from the reference's
// "multicast view" arguments.
#include <arpa/inet.h>
#include <errno.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <stddef.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hpc_amd.h"
#include "hpc_common.h"

namespace {

constexpr int kMaxRanks = 64;
constexpr int kConnectTimeoutMs = 120000;

struct Comm {
  int rank = 0, world = 1, device = -1;
  int listen_fd = -1;
  std::vector<int> fds;  // rank 0: fd per peer rank (index = rank); others: fds[0] = rank 0
  std::vector<void*> local_allocs;
  std::vector<void*> opened;
};

struct Region {
  size_t bytes;
  int world, rank;
  std::vector<void*> peers;  // VA of every rank's buffer in this process
};

std::mutex g_mu;
std::map<int, Comm*> g_comms;
int g_next_id = 1;
std::map<uintptr_t, Region> g_regions;  // keyed by local base

bool send_all(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n) {
    ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    p += k;
    n -= static_cast<size_t>(k);
  }
  return true;
}
bool recv_all(int fd, void* buf, size_t n) {
  char* p = static_cast<char*>(buf);
  while (n) {
    ssize_t k = ::recv(fd, p, n, 0);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      return false;
    }
    p += k;
    n -= static_cast<size_t>(k);
  }
  return true;
}

// "unix://name" or bare "name" -> abstract-namespace unix socket; "tcp://ip:port" -> tcp
// (reference protocol.cc:10-69, listener.cc:62-85).
struct Addr {
  bool tcp = false;
  sockaddr_un un{};
  socklen_t un_len = 0;
  sockaddr_in in{};
};
bool parse_addr(const std::string& name, Addr* a) {
  std::string s = name;
  if (s.rfind("tcp://", 0) == 0) {
    s = s.substr(6);
    const size_t c = s.rfind(':');
    if (c == std::string::npos) return false;
    a->tcp = true;
    a->in.sin_family = AF_INET;
    a->in.sin_port = htons(static_cast<uint16_t>(atoi(s.substr(c + 1).c_str())));
    return inet_pton(AF_INET, s.substr(0, c).c_str(), &a->in.sin_addr) == 1;
  }
  if (s.rfind("unix://", 0) == 0) s = s.substr(7);
  s = "hpc_amd_" + s;
  if (s.size() + 1 >= sizeof(a->un.sun_path)) s.resize(sizeof(a->un.sun_path) - 2);
  a->un.sun_family = AF_UNIX;
  a->un.sun_path[0] = '\0';  // abstract namespace: no filesystem entry, vanishes with the process
  memcpy(a->un.sun_path + 1, s.data(), s.size());
  a->un_len = static_cast<socklen_t>(offsetof(sockaddr_un, sun_path) + 1 + s.size());
  return true;
}

void sleep_ms(int ms) {
  timespec ts{ms / 1000, (ms % 1000) * 1000000L};
  nanosleep(&ts, nullptr);
}

int rendezvous(Comm* c, const std::string& name) {
  if (c->world == 1) return 0;
  Addr a;
  if (!parse_addr(name, &a)) return -2;
  const int fam = a.tcp ? AF_INET : AF_UNIX;
  const sockaddr* sa = a.tcp ? reinterpret_cast<const sockaddr*>(&a.in) : reinterpret_cast<const sockaddr*>(&a.un);
  const socklen_t sl = a.tcp ? sizeof(a.in) : a.un_len;
  if (c->rank == 0) {
    c->listen_fd = ::socket(fam, SOCK_STREAM, 0);
    if (c->listen_fd < 0) return -3;
    int one = 1;
    if (a.tcp) setsockopt(c->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (::bind(c->listen_fd, sa, sl) != 0 || ::listen(c->listen_fd, kMaxRanks) != 0) return -3;
    c->fds.assign(c->world, -1);
    timeval tv{kConnectTimeoutMs / 1000, 0};
    setsockopt(c->listen_fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    for (int i = 1; i < c->world; ++i) {
      int fd = ::accept(c->listen_fd, nullptr, nullptr);
      if (fd < 0) return -3;
      int peer = -1;
      if (!recv_all(fd, &peer, sizeof(peer)) || peer <= 0 || peer >= c->world || c->fds[peer] >= 0) {
        ::close(fd);
        return -3;
      }
      c->fds[peer] = fd;
    }
  } else {
    int fd = -1;
    for (int waited = 0; waited < kConnectTimeoutMs; waited += 50) {  // reference connector.cc:31-55
      fd = ::socket(fam, SOCK_STREAM, 0);
      if (fd < 0) return -3;
      if (::connect(fd, sa, sl) == 0) break;
      ::close(fd);
      fd = -1;
      sleep_ms(50);
    }
    if (fd < 0) return -3;
    if (a.tcp) {
      int one = 1;
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    }
    if (!send_all(fd, &c->rank, sizeof(c->rank))) return -3;
    c->fds.assign(1, fd);
  }
  return 0;
}

// every rank contributes `n` bytes; `out` receives world*n bytes ordered by rank
int allgather(Comm* c, const void* in, size_t n, void* out) {
  char* o = static_cast<char*>(out);
  if (c->world == 1) {
    memcpy(o, in, n);
    return 0;
  }
  if (c->rank == 0) {
    memcpy(o, in, n);
    for (int r = 1; r < c->world; ++r)
      if (!recv_all(c->fds[r], o + r * n, n)) return -3;
    for (int r = 1; r < c->world; ++r)
      if (!send_all(c->fds[r], o, n * c->world)) return -3;
  } else {
    if (!send_all(c->fds[0], in, n)) return -3;
    if (!recv_all(c->fds[0], o, n * c->world)) return -3;
  }
  return 0;
}

Comm* find(int h) {
  auto it = g_comms.find(h);
  return it == g_comms.end() ? nullptr : it->second;
}

}  // namespace

extern "C" int hpc_comm_create(int rank, int world_size, int device_id, const char* name) {
  if (world_size < 1 || world_size > kMaxRanks || rank < 0 || rank >= world_size || !name) return -2;
  Comm* c = new Comm;
  c->rank = rank;
  c->world = world_size;
  c->device = device_id;
  const int rc = rendezvous(c, name);
  if (rc != 0) {
    for (int fd : c->fds)
      if (fd >= 0) ::close(fd);
    if (c->listen_fd >= 0) ::close(c->listen_fd);
    delete c;
    return rc;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  const int id = g_next_id++;
  g_comms[id] = c;
  return id;
}

extern "C" int hpc_comm_destroy(int handle) {
  std::lock_guard<std::mutex> lk(g_mu);
  Comm* c = find(handle);
  if (!c) return -2;
  for (void* p : c->opened) (void)hipIpcCloseMemHandle(p);
  for (void* p : c->local_allocs) {
    g_regions.erase(reinterpret_cast<uintptr_t>(p));
    (void)hipFree(p);
  }
  for (int fd : c->fds)
    if (fd >= 0) ::close(fd);
  if (c->listen_fd >= 0) ::close(c->listen_fd);
  g_comms.erase(handle);
  delete c;
  return 0;
}

extern "C" int hpc_comm_barrier(int handle) {
  Comm* c;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    c = find(handle);
  }
  if (!c) return -2;
  char tok = 1, all[kMaxRanks];
  return allgather(c, &tok, 1, all);
}

extern "C" int hpc_comm_allgather(int handle, const void* in, int64_t nbytes, void* out) {
  Comm* c;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    c = find(handle);
  }
  if (!c || !in || !out || nbytes <= 0) return -2;
  return allgather(c, in, static_cast<size_t>(nbytes), out);
}

extern "C" int hpc_comm_info(int handle, int* rank, int* world_size, int* device_id) {
  std::lock_guard<std::mutex> lk(g_mu);
  Comm* c = find(handle);
  if (!c) return -2;
  if (rank) *rank = c->rank;
  if (world_size) *world_size = c->world;
  if (device_id) *device_id = c->device;
  return 0;
}

// Allocates `nbytes` of uncached device memory on this rank, exchanges IPC handles and opens every
// peer's buffer.  ptrs_out[r] = address of rank r's buffer in THIS process (r == rank: the local
// allocation).  Collective: every rank must call it with the same nbytes.
extern "C" int hpc_comm_create_tensor_sync(int handle, int64_t nbytes, void** ptrs_out) {
  Comm* c;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    c = find(handle);
  }
  if (!c || nbytes <= 0 || !ptrs_out) return -2;
  if (c->device >= 0 && hipSetDevice(c->device) != hipSuccess) return -3;
  void* local = nullptr;
  const size_t bytes = (static_cast<size_t>(nbytes) + 4095) / 4096 * 4096;
  if (hipExtMallocWithFlags(&local, bytes, hipDeviceMallocUncached) != hipSuccess) return -3;
  (void)hipMemset(local, 0, bytes);
  (void)hipDeviceSynchronize();
  Region reg;
  reg.bytes = bytes;
  reg.world = c->world;
  reg.rank = c->rank;
  reg.peers.assign(c->world, nullptr);
  reg.peers[c->rank] = local;
  if (c->world > 1) {
    hipIpcMemHandle_t mine;
    if (hipIpcGetMemHandle(&mine, local) != hipSuccess) {
      (void)hipFree(local);
      return -3;
    }
    std::vector<hipIpcMemHandle_t> all(c->world);
    if (allgather(c, &mine, sizeof(mine), all.data()) != 0) {
      (void)hipFree(local);
      return -3;
    }
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) continue;
      void* p = nullptr;
      if (hipIpcOpenMemHandle(&p, all[r], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipFree(local);
        return -3;
      }
      reg.peers[r] = p;
      c->opened.push_back(p);
    }
    char tok = 1, toks[kMaxRanks];
    if (allgather(c, &tok, 1, toks) != 0) return -3;  // everyone has opened everyone
  }
  for (int r = 0; r < c->world; ++r) ptrs_out[r] = reg.peers[r];
  std::lock_guard<std::mutex> lk(g_mu);
  c->local_allocs.push_back(local);
  g_regions[reinterpret_cast<uintptr_t>(local)] = reg;
  return 0;
}

// Translates an address inside a LOCAL symmetric buffer into the matching address in every
// rank's buffer.  Returns world size (peer_ptrs[0..world-1], *rank_out), or -1 if `ptr` is not
// inside a registered symmetric buffer.
extern "C" int hpc_comm_lookup_peers(const void* ptr, void** peer_ptrs, int* rank_out) {
  std::lock_guard<std::mutex> lk(g_mu);
  const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
  auto it = g_regions.upper_bound(p);
  if (it == g_regions.begin()) return -1;
  --it;
  if (p >= it->first + it->second.bytes) return -1;
  const uintptr_t off = p - it->first;
  for (int r = 0; r < it->second.world; ++r)
    peer_ptrs[r] = static_cast<char*>(it->second.peers[r]) + off;
  if (rank_out) *rank_out = it->second.rank;
  return it->second.world;
}

extern "C" int64_t hpc_comm_region_bytes_left(const void* ptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
  auto it = g_regions.upper_bound(p);
  if (it == g_regions.begin()) return -1;
  --it;
  if (p >= it->first + it->second.bytes) return -1;
  return static_cast<int64_t>(it->first + it->second.bytes - p);
}
