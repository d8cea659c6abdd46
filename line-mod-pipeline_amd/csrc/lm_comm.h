// lm_comm.h -- RCCL communicator of one detector (the ONE exchange step of the path, the all-gather of
// the per-shard match lists).  librccl.so.1 is dlopen'ed when a communicator is created, so the single-GPU product
// carries no RCCL dependency.  The ncclUniqueId travels from rank 0 to the other ranks over a plain TCP socket
// (addr:port; one node, so 127.0.0.1) -- no torch, no MPI.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <string>

struct LmComm {
    void* dl = nullptr;        // librccl handle
    void* comm = nullptr;      // ncclComm_t
    int rank = 0, world = 1;
    void* fn[6] = {};          // resolved entry points (see lm_comm.hip)

    // rank 0 listens on addr:port and hands the unique id to the world - 1 other ranks, which connect with retries
    // for up to timeout_s seconds.  The caller has selected the HIP device.  Returns false with err set.
    bool init(int rank, int world, const char* addr, int port, int timeout_s, std::string& err);
    void destroy();
    // `bytes` from every rank, rank-major, into recv (world * bytes); enqueued on st
    bool all_gather(const void* send, void* recv, size_t bytes, hipStream_t st, std::string& err);
    // element-wise max of n doubles (device buffers); enqueued on st
    bool all_reduce_max_f64(const void* send, void* recv, size_t n, hipStream_t st, std::string& err);
    ~LmComm() { destroy(); }
};

// rank 0 -> every rank: n bytes over TCP (the rendezvous LmComm::init uses for the ncclUniqueId); host only.
bool lm_tcp_broadcast(int rank, int world, const char* addr, int port, int timeout_s, void* buf, size_t n, std::string& err);
