// TEST INFRASTRUCTURE: a shared-memory stand-in for the five RCCL entry points robo_amd/csrc/comm.hip loads
// (ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclCommDestroy, ncclGetErrorString), so that the library's
// collective entry points can be driven with world_size 2 in the GPU-less build container: "device" pointers of the
// interpreter build (tests/hipemu) are host pointers, ranks are processes on one machine, the exchange goes through a
// POSIX shared-memory segment named by the unique id.  Selected with ROBO_RCCL_LIB; never loaded by the product.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

namespace {
constexpr size_t SLOT_BYTES = 8u << 20;   // per rank and exchange
constexpr int MAX_RANKS = 8;

struct Header {
    std::atomic<int> arrived;
    std::atomic<int> generation;
};

struct Comm {
    int rank, world;
    Header* hdr;
    char* slots;
    size_t bytes;
    char name[64];
};

struct Id {
    char internal[128];
};

bool barrier(Comm* c) {
    const int gen = c->hdr->generation.load();
    if (c->hdr->arrived.fetch_add(1) + 1 == c->world) {
        c->hdr->arrived.store(0);
        c->hdr->generation.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->generation.load() == gen) {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;   // a rank is missing
    }
    return true;
}
}  // namespace

extern "C" {

int ncclGetUniqueId(Id* id) {
    static std::atomic<int> counter{0};
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/robo_fake_rccl_%d_%d", (int)getpid(), counter.fetch_add(1));
    const size_t bytes = sizeof(Header) + 64 + (size_t)MAX_RANKS * SLOT_BYTES;
    const int fd = shm_open(id->internal, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return 2;
    if (ftruncate(fd, (off_t)bytes) != 0) return 2;   // zero-filled: arrived = generation = 0
    close(fd);
    return 0;
}

int ncclCommInitRank(void** out, int nranks, Id id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return 4;
    const size_t bytes = sizeof(Header) + 64 + (size_t)MAX_RANKS * SLOT_BYTES;
    int fd = -1;
    for (int tries = 0; tries < 2000 && fd < 0; ++tries) {
        fd = shm_open(id.internal, O_RDWR, 0600);
        if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    if (fd < 0) return 2;
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    Comm* c = new Comm();
    c->rank = rank;
    c->world = nranks;
    c->hdr = reinterpret_cast<Header*>(p);
    c->slots = reinterpret_cast<char*>(p) + sizeof(Header) + 64;
    c->bytes = bytes;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    if (!barrier(c)) return 3;
    *out = c;
    return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int datatype, void* comm, void* /*stream*/) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (datatype != 8) return 4;                     // ncclFloat64 is all comm.hip sends
    const size_t total = count * 8;
    for (size_t off = 0; off < total; off += SLOT_BYTES) {
        const size_t n = total - off < SLOT_BYTES ? total - off : SLOT_BYTES;
        memcpy(c->slots + (size_t)c->rank * SLOT_BYTES, reinterpret_cast<const char*>(send) + off, n);
        if (!barrier(c)) return 3;
        for (int r = 0; r < c->world; ++r)
            memcpy(reinterpret_cast<char*>(recv) + (size_t)r * total + off, c->slots + (size_t)r * SLOT_BYTES, n);
        if (!barrier(c)) return 3;
    }
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return 0;
    const bool last = barrier(c) && c->rank == 0;
    munmap(reinterpret_cast<void*>(c->hdr), c->bytes);
    if (last) shm_unlink(c->name);
    delete c;
    return 0;
}

const char* ncclGetErrorString(int r) {
    switch (r) {
        case 0: return "success";
        case 2: return "system error (shared memory)";
        case 3: return "a rank did not arrive within 120 s";
        case 4: return "invalid argument";
        default: return "error";
    }
}

}  // extern "C"
