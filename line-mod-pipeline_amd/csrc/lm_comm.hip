// lm_comm.hip -- see lm_comm.h.
#include "lm_comm.h"

#include <arpa/inet.h>
#include <fcntl.h>
#include <poll.h>
#include <dlfcn.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <netdb.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

namespace {

typedef ncclResult_t (*fn_get_unique_id)(ncclUniqueId*);
typedef ncclResult_t (*fn_comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*fn_comm_destroy)(ncclComm_t);
typedef ncclResult_t (*fn_all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
typedef ncclResult_t (*fn_all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
typedef const char* (*fn_get_error_string)(ncclResult_t);
enum { F_UID, F_INIT, F_DESTROY, F_GATHER, F_REDUCE, F_ERRSTR };

bool send_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) {
        ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
        if (k <= 0) { if (errno == EINTR) continue; return false; }
        c += k; n -= (size_t)k;
    }
    return true;
}
bool recv_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) {
        ssize_t k = ::recv(fd, c, n, 0);
        if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; }
        c += k; n -= (size_t)k;
    }
    return true;
}

#define LM_RDV_MAGIC 0x4C4D5256   // "LMRV": first word of a peer's handshake

int ms_left(std::chrono::steady_clock::time_point deadline) {
    const auto d = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
    return d < 0 ? 0 : (int)std::min<long long>(d, 1 << 30);
}

void set_io_timeout(int fd, int ms) {
    if (ms < 1) ms = 1;
    timeval tv; tv.tv_sec = ms / 1000; tv.tv_usec = (ms % 1000) * 1000;
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

// connect() that gives up after `ms` milliseconds (non-blocking connect + poll)
bool connect_timeout(int fd, const sockaddr_in& sa, int ms) {
    const int fl = fcntl(fd, F_GETFL, 0);
    fcntl(fd, F_SETFL, fl | O_NONBLOCK);
    int rc = ::connect(fd, reinterpret_cast<const sockaddr*>(&sa), sizeof(sa));
    if (rc != 0 && errno == EINPROGRESS) {
        pollfd pf; pf.fd = fd; pf.events = POLLOUT; pf.revents = 0;
        if (::poll(&pf, 1, ms) == 1) {
            int soerr = 0; socklen_t sl = sizeof(soerr);
            if (getsockopt(fd, SOL_SOCKET, SO_ERROR, &soerr, &sl) == 0 && soerr == 0) rc = 0;
        }
    }
    fcntl(fd, F_SETFL, fl);
    return rc == 0;
}

// rank 0 -> everyone: `n` bytes over TCP.  A peer first sends {magic, rank}, then receives the payload.  One overall
// deadline on every rank; rank 0 drops connections that do not complete the handshake and keeps accepting.
bool tcp_broadcast(int rank, int world, const char* addr, int port, int timeout_s, void* buf, size_t n, std::string& err) {
    if (world <= 1) return true;
    sockaddr_in sa;
    std::memset(&sa, 0, sizeof(sa));
    sa.sin_family = AF_INET;
    sa.sin_port = htons((uint16_t)port);
    if (inet_pton(AF_INET, addr, &sa.sin_addr) != 1) {
        // not a dotted quad (e.g. MASTER_ADDR=localhost or a node name): resolve it
        addrinfo hints, *res = nullptr;
        std::memset(&hints, 0, sizeof(hints));
        hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
        if (getaddrinfo(addr, nullptr, &hints, &res) != 0 || !res) { err = std::string("bad rendezvous address: ") + addr; return false; }
        sa.sin_addr = reinterpret_cast<sockaddr_in*>(res->ai_addr)->sin_addr;
        freeaddrinfo(res);
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(timeout_s);
    if (rank == 0) {
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) { err = "socket() failed"; return false; }
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        if (::bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0 || ::listen(ls, world + 8) != 0) {
            err = std::string("rendezvous: cannot listen on ") + addr + ":" + std::to_string(port) + ": " + std::strerror(errno);
            ::close(ls);
            return false;
        }
        // the listening socket never blocks: a connection reset between poll() and accept() must send us back to poll() under the
        // one deadline, not park us in accept()
        fcntl(ls, F_SETFL, fcntl(ls, F_GETFL, 0) | O_NONBLOCK);
        std::vector<bool> seen((size_t)world, false);
        int served = 0, strays = 0;
        while (served < world - 1) {
            pollfd pf; pf.fd = ls; pf.events = POLLIN; pf.revents = 0;
            const int left = ms_left(deadline);
            const int pr = left > 0 ? ::poll(&pf, 1, left) : 0;
            if (pr < 0 && errno == EINTR) continue;
            if (pr <= 0) {
                err = "rendezvous: timed out after " + std::to_string(timeout_s) + " s with " + std::to_string(served) + " of " +
                      std::to_string(world - 1) + " peers served (" + std::to_string(strays) + " stray connections ignored)";
                ::close(ls);
                return false;
            }
            int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) continue;                        // EAGAIN / ECONNABORTED: poll again
            fcntl(fd, F_SETFL, fcntl(fd, F_GETFL, 0) & ~O_NONBLOCK);     // (an accepted socket may inherit the flag: the handshake below uses timeouts)
            // a peer that connects sends its 8 bytes at once: half a second keeps a few silent strays from eating the real peers' retry window
            set_io_timeout(fd, std::max(1, std::min(ms_left(deadline), 500)));
            int32_t hs[2] = {0, -1};
            const bool hello = recv_all(fd, hs, sizeof(hs)) && hs[0] == LM_RDV_MAGIC && hs[1] > 0 && hs[1] < world && !seen[(size_t)hs[1]];
            if (hello) {
                set_io_timeout(fd, std::max(ms_left(deadline), 1000));
                if (send_all(fd, buf, n)) { seen[(size_t)hs[1]] = true; ++served; }
                // a peer whose connection broke while the payload travelled reconnects (it retries until its deadline)
            } else {
                ++strays;
            }
            ::close(fd);
        }
        ::close(ls);
        return true;
    }
    std::string last = "no attempt made";
    for (;;) {
        int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) { err = "socket() failed"; return false; }
        if (connect_timeout(fd, sa, std::min(std::max(ms_left(deadline), 1), 2000))) {
            set_io_timeout(fd, std::max(ms_left(deadline), 1000));
            int32_t hs[2] = {LM_RDV_MAGIC, rank};
            const bool ok = send_all(fd, hs, sizeof(hs)) && recv_all(fd, buf, n);
            ::close(fd);
            if (ok) return true;
            last = "connection to rank 0 broke";
        } else {
            ::close(fd);
            last = "cannot reach rank 0";
        }
        if (ms_left(deadline) == 0) {
            err = std::string("rendezvous: ") + last + " at " + addr + ":" + std::to_string(port) + " within " + std::to_string(timeout_s) + " s";
            return false;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}

}  // namespace

bool lm_tcp_broadcast(int rank, int world, const char* addr, int port, int timeout_s, void* buf, size_t n, std::string& err) {
    return tcp_broadcast(rank, world, addr, port, timeout_s, buf, n, err);
}

bool LmComm::load(std::string& err) {
    if (dl) return true;
    const char* names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char* nme : names) { dl = dlopen(nme, RTLD_NOW | RTLD_LOCAL); if (dl) break; }
    if (!dl) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    const char* syms[6] = {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclAllReduce", "ncclGetErrorString"};
    for (int i = 0; i < 6; ++i) {
        fn[i] = dlsym(dl, syms[i]);
        if (!fn[i]) { err = std::string("librccl lacks ") + syms[i]; destroy(); return false; }
    }
    return true;
}

bool LmComm::unique_id(void* out, std::string& err) {
    static_assert(sizeof(ncclUniqueId) == LM_NCCL_ID_BYTES, "ncclUniqueId size");
    if (!load(err)) return false;
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    ncclResult_t r = reinterpret_cast<fn_get_unique_id>(fn[F_UID])(&id);
    if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + reinterpret_cast<fn_get_error_string>(fn[F_ERRSTR])(r); return false; }
    std::memcpy(out, &id, sizeof(id));
    return true;
}

bool LmComm::init_rank(int rank_, int world_, const void* idp, std::string& err) {
    if (world_ < 1 || rank_ < 0 || rank_ >= world_) { err = "bad rank / world size"; return false; }
    if (!load(err)) return false;
    if (comm) { err = "communicator already initialised"; return false; }
    ncclUniqueId id;
    std::memcpy(&id, idp, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = reinterpret_cast<fn_comm_init_rank>(fn[F_INIT])(&c, world_, id, rank_);
    if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + reinterpret_cast<fn_get_error_string>(fn[F_ERRSTR])(r); return false; }
    rank = rank_; world = world_;
    comm = c;
    return true;
}

bool LmComm::init(int rank_, int world_, const char* addr, int port, int timeout_s, std::string& err) {
    destroy();
    if (world_ < 1 || rank_ < 0 || rank_ >= world_) { err = "bad rank / world size"; return false; }
    unsigned char id[LM_NCCL_ID_BYTES] = {};
    if (!load(err)) return false;
    if (rank_ == 0 && !unique_id(id, err)) { destroy(); return false; }
    if (!tcp_broadcast(rank_, world_, addr ? addr : "127.0.0.1", port, timeout_s, id, sizeof(id), err)) { destroy(); return false; }
    if (!init_rank(rank_, world_, id, err)) { destroy(); return false; }
    return true;
}

void LmComm::destroy() {
    if (comm && fn[F_DESTROY]) reinterpret_cast<fn_comm_destroy>(fn[F_DESTROY])(static_cast<ncclComm_t>(comm));
    comm = nullptr;
    // librccl stays loaded: unloading a library that owns GPU resources and threads at exit is not worth the risk
    dl = nullptr;
}

bool LmComm::all_gather(const void* send, void* recv, size_t bytes, hipStream_t st, std::string& err) {
    if (!comm) { err = "no communicator"; return false; }
    ncclResult_t r = reinterpret_cast<fn_all_gather>(fn[F_GATHER])(send, recv, bytes, ncclUint8, static_cast<ncclComm_t>(comm), st);
    if (r != ncclSuccess) { err = std::string("ncclAllGather: ") + reinterpret_cast<fn_get_error_string>(fn[F_ERRSTR])(r); return false; }
    return true;
}

bool LmComm::all_reduce_max_f64(const void* send, void* recv, size_t n, hipStream_t st, std::string& err) {
    if (!comm) { err = "no communicator"; return false; }
    ncclResult_t r = reinterpret_cast<fn_all_reduce>(fn[F_REDUCE])(send, recv, n, ncclFloat64, ncclMax, static_cast<ncclComm_t>(comm), st);
    if (r != ncclSuccess) { err = std::string("ncclAllReduce: ") + reinterpret_cast<fn_get_error_string>(fn[F_ERRSTR])(r); return false; }
    return true;
}
