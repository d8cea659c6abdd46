// lm_comm.h -- RCCL communicator of one detector (the ONE exchange step of the path, the all-gather of
// the per-shard match lists).  librccl.so.1 is dlopen'ed when a communicator is created, so the single-GPU product
// carries no RCCL dependency.  The ncclUniqueIds travel from rank 0 to the other ranks over a plain TCP socket
// (addr:port; one node, so 127.0.0.1) -- no torch, no MPI.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <string>

#define LM_NCCL_ID_BYTES 128   // sizeof(ncclUniqueId)

struct LmComm {
    void* dl = nullptr;        // librccl handle
    void* comm = nullptr;      // ncclComm_t
    int rank = 0, world = 1;
    void* fn[6] = {};          // resolved entry points (see lm_comm.hip)

    // dlopen librccl and resolve the entry points
    bool load(std::string& err);
    // ncclGetUniqueId into id[LM_NCCL_ID_BYTES] (rank 0 only)
    bool unique_id(void* id, std::string& err);
    // ncclCommInitRank with an id every rank already holds.  The caller has selected the HIP device.
    bool init_rank(int rank, int world, const void* id, std::string& err);
    // load + unique_id on rank 0 + lm_tcp_broadcast + init_rank: rank 0 listens on addr:port and hands the id to the
    // world - 1 other ranks, which connect with retries, all under one deadline of timeout_s seconds.
    bool init(int rank, int world, const char* addr, int port, int timeout_s, std::string& err);
    void destroy();
    // `bytes` from every rank, rank-major, into recv (world * bytes); enqueued on st
    bool all_gather(const void* send, void* recv, size_t bytes, hipStream_t st, std::string& err);
    // element-wise max of n doubles (device buffers); enqueued on st
    bool all_reduce_max_f64(const void* send, void* recv, size_t n, hipStream_t st, std::string& err);
    ~LmComm() { destroy(); }
};

// rank 0 -> every rank: n bytes over TCP (the rendezvous of the ncclUniqueIds); host only.  ONE deadline of timeout_s
// seconds covers the whole exchange on every rank; rank 0 ignores connections that do not speak the handshake (port
// scanners, a stale process of an earlier run) and keeps accepting; the other ranks retry a refused or broken
// connection until the deadline.
bool lm_tcp_broadcast(int rank, int world, const char* addr, int port, int timeout_s, void* buf, size_t n, std::string& err);
