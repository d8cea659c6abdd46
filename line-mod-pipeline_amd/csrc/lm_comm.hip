// lm_comm.hip -- see lm_comm.h.
#include "lm_comm.h"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <netdb.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <unistd.h>

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

// rank 0 -> everyone: `n` bytes over TCP.  Every client first sends its rank (sanity), then receives the payload.
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
        if (::bind(ls, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) != 0 || ::listen(ls, world) != 0) {
            err = std::string("rendezvous: cannot listen on ") + addr + ":" + std::to_string(port) + ": " + std::strerror(errno);
            ::close(ls);
            return false;
        }
        timeval tv; tv.tv_sec = timeout_s; tv.tv_usec = 0;
        setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        std::vector<bool> seen((size_t)world, false);
        for (int k = 1; k < world; ++k) {
            int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) { err = "rendezvous: timed out waiting for the other ranks"; ::close(ls); return false; }
            setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
            int32_t r = -1;
            bool ok = recv_all(fd, &r, 4) && r > 0 && r < world && !seen[(size_t)r] && send_all(fd, buf, n);
            if (ok) seen[(size_t)r] = true;
            ::close(fd);
            if (!ok) { err = "rendezvous: bad handshake from a peer"; ::close(ls); return false; }
        }
        ::close(ls);
        return true;
    }
    for (;;) {
        int fd = ::socket(AF_INET, SOCK_STREAM, 0);
        if (fd < 0) { err = "socket() failed"; return false; }
        if (::connect(fd, reinterpret_cast<sockaddr*>(&sa), sizeof(sa)) == 0) {
            timeval tv; tv.tv_sec = timeout_s; tv.tv_usec = 0;
            setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
            int32_t r = rank;
            bool ok = send_all(fd, &r, 4) && recv_all(fd, buf, n);
            ::close(fd);
            if (ok) return true;
            err = "rendezvous: connection to rank 0 broke";
            return false;
        }
        ::close(fd);
        if (std::chrono::steady_clock::now() > deadline) {
            err = std::string("rendezvous: cannot reach rank 0 at ") + addr + ":" + std::to_string(port);
            return false;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
}

}  // namespace

bool lm_tcp_broadcast(int rank, int world, const char* addr, int port, int timeout_s, void* buf, size_t n, std::string& err) {
    return tcp_broadcast(rank, world, addr, port, timeout_s, buf, n, err);
}

bool LmComm::init(int rank_, int world_, const char* addr, int port, int timeout_s, std::string& err) {
    destroy();
    if (world_ < 1 || rank_ < 0 || rank_ >= world_) { err = "bad rank / world size"; return false; }
    const char* names[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char* nme : names) { dl = dlopen(nme, RTLD_NOW | RTLD_LOCAL); if (dl) break; }
    if (!dl) { err = std::string("cannot load librccl: ") + dlerror(); return false; }
    const char* syms[6] = {"ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllGather", "ncclAllReduce", "ncclGetErrorString"};
    for (int i = 0; i < 6; ++i) {
        fn[i] = dlsym(dl, syms[i]);
        if (!fn[i]) { err = std::string("librccl lacks ") + syms[i]; destroy(); return false; }
    }
    rank = rank_; world = world_;
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    if (rank == 0) {
        ncclResult_t r = reinterpret_cast<fn_get_unique_id>(fn[F_UID])(&id);
        if (r != ncclSuccess) { err = std::string("ncclGetUniqueId: ") + reinterpret_cast<fn_get_error_string>(fn[F_ERRSTR])(r); destroy(); return false; }
    }
    if (!tcp_broadcast(rank, world, addr ? addr : "127.0.0.1", port, timeout_s, &id, sizeof(id), err)) { destroy(); return false; }
    ncclComm_t c = nullptr;
    ncclResult_t r = reinterpret_cast<fn_comm_init_rank>(fn[F_INIT])(&c, world, id, rank);
    if (r != ncclSuccess) { err = std::string("ncclCommInitRank: ") + reinterpret_cast<fn_get_error_string>(fn[F_ERRSTR])(r); destroy(); return false; }
    comm = c;
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
