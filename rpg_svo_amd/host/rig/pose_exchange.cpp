// pose_exchange.cpp -- see pose_exchange.h.  RCCL (rccl.h) is the only dependency besides sockets.
#include "pose_exchange.h"

#include <arpa/inet.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

namespace svo_hip {

namespace {
void sendAll(int fd, const void* p, size_t n) {
  const char* c = static_cast<const char*>(p);
  while (n) {
    const ssize_t k = ::send(fd, c, n, 0);
    if (k <= 0) throw std::runtime_error("pose exchange bootstrap: send failed");
    c += k; n -= (size_t)k;
  }
}
void recvAll(int fd, void* p, size_t n) {
  char* c = static_cast<char*>(p);
  while (n) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k <= 0) throw std::runtime_error("pose exchange bootstrap: recv failed");
    c += k; n -= (size_t)k;
  }
}
int envInt(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v && *v ? std::atoi(v) : fallback;
}
}  // namespace

namespace {
// what a rank says before it is handed the blob: who it is and which job it belongs to
struct Hello {
  uint32_t magic, rank, world, token;
};
constexpr uint32_t HELLO_MAGIC = 0x53564f52u;  // "SVOR"
struct Fd {  // closes on every path out of a scope
  int fd;
  explicit Fd(int f) : fd(f) {}
  ~Fd() { if (fd >= 0) ::close(fd); }
  Fd(const Fd&) = delete;
  Fd& operator=(const Fd&) = delete;
};
}  // namespace

void tcpBroadcast(int rank, int world, const std::string& addr, int port, void* blob, size_t bytes, int timeout_s) {
  if (world <= 1) return;
  sockaddr_in sa;
  std::memset(&sa, 0, sizeof(sa));
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  if (::inet_pton(AF_INET, addr.c_str(), &sa.sin_addr) != 1) throw std::runtime_error("pose exchange bootstrap: MASTER_ADDR must be an IPv4 address");
  // a shared secret of the job (SVO_RIG_TOKEN, the launcher sets the same value for every rank; 0 when unset)
  const uint32_t token = (uint32_t)envInt("SVO_RIG_TOKEN", 0);
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
  if (rank == 0) {
    Fd ls(::socket(AF_INET, SOCK_STREAM, 0));
    if (ls.fd < 0) throw std::runtime_error("pose exchange bootstrap: socket() failed");
    int one = 1;
    if (::setsockopt(ls.fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one)) != 0)
      throw std::runtime_error("pose exchange bootstrap: setsockopt(SO_REUSEADDR) failed");
    // bind to MASTER_ADDR itself (127.0.0.1 on a single node), not to every interface.  MASTER_ADDR need not be an
    // address of a local interface (a NAT or service address, another NIC of a multi-homed host): SVO_RIG_BIND names
    // the address to listen on then (0.0.0.0: every interface, the operator's explicit choice).  Without SVO_RIG_BIND,
    // EADDRNOTAVAIL falls back to every interface ONLY for a job that carries a secret (SVO_RIG_TOKEN != 0): the hello
    // (magic / world / token) is all that keeps a stranger from being handed the ncclUniqueId, and magic and world
    // size are no secret.
    sockaddr_in la = sa;
    if (const char* b = std::getenv("SVO_RIG_BIND")) {
      if (::inet_pton(AF_INET, b, &la.sin_addr) != 1) throw std::runtime_error("pose exchange bootstrap: SVO_RIG_BIND must be an IPv4 address");
    }
    int rc = ::bind(ls.fd, reinterpret_cast<sockaddr*>(&la), sizeof(la));
    if (rc != 0 && errno == EADDRNOTAVAIL && !std::getenv("SVO_RIG_BIND")) {
      if (token == 0)
        throw std::runtime_error("pose exchange bootstrap: MASTER_ADDR is not an address of this host; set SVO_RIG_BIND to the "
                                 "address to listen on, or SVO_RIG_TOKEN (a job secret) to listen on every interface");
      la.sin_addr.s_addr = htonl(INADDR_ANY);
      rc = ::bind(ls.fd, reinterpret_cast<sockaddr*>(&la), sizeof(la));
    }
    if (rc != 0 || ::listen(ls.fd, world) != 0) throw std::runtime_error("pose exchange bootstrap: cannot listen on MASTER_ADDR:port");
    timeval tv = {timeout_s, 0};
    // a connection has this long to introduce itself: one silent stranger must not use up the whole deadline
    timeval tv_hello = {timeout_s < 2 ? timeout_s : 2, 0};
    if (::setsockopt(ls.fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv)) != 0)
      throw std::runtime_error("pose exchange bootstrap: setsockopt(SO_RCVTIMEO) failed");
    // the blob goes to ranks 1..world-1 of THIS job, once each: a peer introduces itself first; anything else that
    // connects (wrong magic / world / token, a rank seen before) is dropped without being sent a byte
    std::vector<bool> served((size_t)world, false);
    int n_served = 0;
    while (n_served < world - 1) {
      if (std::chrono::steady_clock::now() > deadline) throw std::runtime_error("pose exchange bootstrap: a rank did not connect in time");
      Fd c(::accept(ls.fd, NULL, NULL));
      if (c.fd < 0) throw std::runtime_error("pose exchange bootstrap: a rank did not connect in time");
      ::setsockopt(c.fd, SOL_SOCKET, SO_RCVTIMEO, &tv_hello, sizeof(tv_hello));
      Hello h;
      try {
        recvAll(c.fd, &h, sizeof(h));
      } catch (const std::exception&) {
        continue;  // not one of ours
      }
      if (h.magic != HELLO_MAGIC || h.world != (uint32_t)world || h.token != token || h.rank == 0 || h.rank >= (uint32_t)world ||
          served[h.rank])
        continue;
      try {
        sendAll(c.fd, blob, bytes);
      } catch (const std::exception&) {
        continue;  // the peer went away mid-transfer: its rank is not served, it may connect again
      }
      served[h.rank] = true;
      ++n_served;
    }
  } else {
    for (;;) {
      {
        Fd c(::socket(AF_INET, SOCK_STREAM, 0));
        if (c.fd < 0) throw std::runtime_error("pose exchange bootstrap: socket() failed");
        if (::connect(c.fd, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) == 0) {
          const Hello h = {HELLO_MAGIC, (uint32_t)rank, (uint32_t)world, token};
          sendAll(c.fd, &h, sizeof(h));
          recvAll(c.fd, blob, bytes);
          return;
        }
      }
      if (std::chrono::steady_clock::now() > deadline) throw std::runtime_error("pose exchange bootstrap: rank 0 is not reachable");
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
  }
}

PoseExchange* PoseExchange::fromEnvironment() {
  const int rank = envInt("RANK", 0), world = envInt("WORLD_SIZE", 1);
  const char* addr = std::getenv("MASTER_ADDR");
  const int port = envInt("SVO_RIG_PORT", envInt("MASTER_PORT", 29600) + 17);
  return new PoseExchange(rank, world, addr && *addr ? addr : "127.0.0.1", port);
}

PoseExchange::PoseExchange(int rank, int world, const std::string& addr, int port) : rank_(rank), world_(world), comm_(NULL) {
  if (rank < 0 || world < 1 || rank >= world) throw std::runtime_error("pose exchange: bad rank / world size");
  ncclUniqueId id;
  std::memset(&id, 0, sizeof(id));
  if (rank == 0 && ncclGetUniqueId(&id) != ncclSuccess) throw std::runtime_error("pose exchange: ncclGetUniqueId failed");
  tcpBroadcast(rank, world, addr, port, &id, sizeof(id));
  ncclComm_t comm;
  if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) throw std::runtime_error("pose exchange: ncclCommInitRank failed");
  comm_ = comm;
}

PoseExchange::~PoseExchange() {
  if (comm_) ncclCommDestroy(static_cast<ncclComm_t>(comm_));
}

void PoseExchange::allGather(const double* d_local, double* d_all, int n, void* stream) {
  if (ncclAllGather(d_local, d_all, (size_t)n, ncclDouble, static_cast<ncclComm_t>(comm_), static_cast<hipStream_t>(stream)) != ncclSuccess)
    throw std::runtime_error("pose exchange: ncclAllGather failed");
}

}  // namespace svo_hip
