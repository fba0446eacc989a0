// mock_rccl.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl's five entry points that libsdf_hip.so uses
// (csrc/sdf_comm.inc: ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather / ncclGetErrorString), so
// that the library's own multi-GPU step can be driven by SEVERAL PROCESSES ON ONE GPU.  (RCCL itself refuses two
// ranks on one device; the boxes this repository is built on have one.)  The ranks meet in a POSIX shared-memory
// segment named by the "unique id"; an all-gather synchronises the caller's stream, copies the rank's piece to its
// slot in host memory, waits for every rank at a barrier, and copies all pieces into the receive buffer.  The
// operation is complete when the call returns, which satisfies the stream-ordering contract of the real thing
// trivially.  Selected with SDF_RCCL_LIB (tests/test_gpu.py::test_native_exchange_between_processes_on_one_device);
// never part of the product.
//   hipcc -shared -fPIC -O2 -o mock_rccl.so tests/native/mock_rccl.cpp
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclDataType_t;

namespace {
enum { MAX_RANKS = 8 };
// bytes per rank (MOCK_RCCL_SLOT_MB, default 96; the segment is sparse: only what is written is backed by memory)
const size_t SLOT_BYTES = [] { const char *e = getenv("MOCK_RCCL_SLOT_MB"); return (size_t)(e ? atol(e) : 96) << 20; }();
struct Shared {
    std::atomic<int> arrived, generation, attached;
    char pad[52];
};
struct Comm {
    Shared *sh;
    unsigned char *slots;
    size_t map_bytes;
    int rank, nranks;
    char name[64];
    void *bounce = nullptr;
    size_t bounce_bytes = 0;
};
void barrier(Comm *c) {
    const int g = c->sh->generation.load();
    if (c->sh->arrived.fetch_add(1) == c->nranks - 1) { c->sh->arrived.store(0); c->sh->generation.fetch_add(1); }
    else while (c->sh->generation.load() == g) sched_yield();
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    static std::atomic<int> counter{0};
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/sdf_mock_rccl_%d_%d", (int)getpid(), counter.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank; c->nranks = nranks;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    c->map_bytes = sizeof(Shared) + (size_t)nranks * SLOT_BYTES;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = (Shared *)p;                                   // (a fresh segment is zero-filled: counters start at 0)
    c->slots = (unsigned char *)p + sizeof(Shared);
    c->sh->attached.fetch_add(1);
    while (c->sh->attached.load() < nranks) sched_yield();   // collective, like the real call
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return ncclSuccess;
    if (c->rank == 0) shm_unlink(c->name);
    if (c->bounce) (void)hipHostFree(c->bounce);
    munmap(c->sh, c->map_bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dtype, void *comm, hipStream_t stream) {
    Comm *c = (Comm *)comm;
    if (!c || !sendbuff || !recvbuff || dtype != 1 /* ncclUint8 */) return ncclInvalidArgument;
    if (count > SLOT_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    // (the HIP runtime never sees the shared pages: device <-> a private pinned bounce buffer <-> shared memory)
    if (count > c->bounce_bytes) {
        if (c->bounce) (void)hipHostFree(c->bounce);
        c->bounce = nullptr; c->bounce_bytes = 0;
        if (hipHostMalloc(&c->bounce, count, hipHostMallocDefault) != hipSuccess) return ncclUnhandledCudaError;
        c->bounce_bytes = count;
    }
    if (count) {
        if (hipMemcpy(c->bounce, sendbuff, count, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, c->bounce, count);
    }
    barrier(c);
    for (int r = 0; r < c->nranks && count; r++) {
        memcpy(c->bounce, c->slots + (size_t)r * SLOT_BYTES, count);
        if (hipMemcpy((unsigned char *)recvbuff + (size_t)r * count, c->bounce, count, hipMemcpyHostToDevice) != hipSuccess)
            return ncclUnhandledCudaError;
    }
    barrier(c);                                            // (the slots may be overwritten by the next collective now)
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "mock rccl: HIP error";
    case ncclSystemError: return "mock rccl: shared memory error";
    case ncclInvalidArgument: return "mock rccl: invalid argument (piece larger than the mock's slot?  MOCK_RCCL_SLOT_MB)";
    default: return "mock rccl: internal error";
    }
}

}  // extern "C"
