// librccl_stub.cpp — TEST INFRASTRUCTURE.  The nine RCCL entry points csrc/m2s_dist.cpp resolves (load_rccl), implemented for
// ranks that are separate PROCESSES sharing ONE GPU, over POSIX shared memory.
//
// Why it exists: RCCL refuses two ranks on one device, and the builder's box has one — so until a multi-GPU node runs the
// driver's scaling bench, the multi-PROCESS code path of m2s_dist.cpp (unique-id hand-over between processes, ncclCommInitRank,
// the counter all-gathers issued by the worker thread, the grouped exact-size ncclSend / ncclRecv schedule, gather to one root,
// the sample sort's exchanges, what happens when a rank dies inside a collective) had never executed.  Selected with
// M2S_RCCL_PATH=<this .so>; tests/test_gpu_dist_stub.py drives tools/mesh2splat_cli --gpus N --gather --one-device, the rank
// script tests/dist_rank.py and bench.py --gpus N through it.
//
// What it checks that real RCCL would turn into a hang or silent corruption: every rank enters the same sequence of
// all-gathers (element counts are compared across ranks), every ncclSend meets an ncclRecv of the same size (messages between a
// pair of ranks match in order) and vice versa, ranks are in range, nothing is issued on a destroyed communicator.  As in RCCL,
// point-to-point groups involve only the ranks that send or receive: there is no hidden rendezvous of the whole communicator.  Every wait is
// bounded (M2S_STUB_RCCL_TIMEOUT seconds, default 30): a rank that never arrives is an ncclSystemError on the others.
//
// Semantics: the data path is device -> shared memory -> device with blocking copies inside the call (real RCCL enqueues on the
// stream and returns; a caller that is correct with real RCCL's asynchrony and whose ranks all issue the same sequence is also
// correct here, not the other way round).  M2S_STUB_RCCL_LOG=<prefix> appends one line per call to <prefix>.<rank>;
// M2S_STUB_RCCL_DIE="<rank>:<op>" makes that rank _exit(9) when it enters its op-th all-gather (0-based), "<rank>:-<g>" (negative)
// when it flushes its g-th group (1-based) — the crash tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
constexpr int kMaxRanks = 64, kMaxMsgs = 256;
constexpr uint32_t kMagic = 0x4D325342u;

struct Msg { int32_t peer; uint32_t consumed; uint64_t bytes, off; };
struct Slot {
    uint64_t seq;            // the operation this slot describes
    uint32_t kind;           // 1 all-gather, 2 group
    uint64_t count;          // all-gather: bytes per rank
    uint32_t n_msgs;         // group: sends of this rank
    uint64_t shm_bytes;
    Msg msg[kMaxMsgs];
};
struct Ctl {
    std::atomic<uint32_t> magic, nranks, joined, dead;
    std::atomic<uint32_t> bar_count, bar_gen;
    Slot slot[kMaxRanks];
};
struct Comm {
    Ctl* ctl = nullptr;
    std::string token;
    int rank = 0, nranks = 0;
    uint64_t seq = 0;                       // collectives (all-gathers) entered
    uint64_t groups = 0;                    // point-to-point groups flushed
    uint64_t sent[kMaxRanks] = {}, recvd[kMaxRanks] = {};   // messages to / from every peer so far (they match in order)
    bool destroyed = false;
    FILE* log = nullptr;
    int die_rank = -1; long die_op = -1;
    double timeout_s = 30.0;
};
struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; hipStream_t st; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
std::atomic<uint32_t> g_id_counter{0};

size_t dtype_size(int t) {   // rccl.h ncclDataType_t
    switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: case 9: return 2; default: return 0; }
}
void logf(Comm* c, const char* fmt, ...) {
    if (!c->log) return;
    va_list ap; va_start(ap, fmt); vfprintf(c->log, fmt, ap); va_end(ap); fputc('\n', c->log); fflush(c->log);
}
std::string data_name(const Comm* c, int rank, uint64_t seq) { return "/" + c->token + "." + std::to_string(rank) + "." + std::to_string(seq); }

// all ranks of the communicator; false: somebody never arrived (the communicator is dead for everybody from then on)
bool barrier(Comm* c) {
    Ctl* k = c->ctl;
    if (k->dead.load()) return false;
    const uint32_t gen = k->bar_gen.load();
    if (k->bar_count.fetch_add(1) + 1 == (uint32_t)c->nranks) { k->bar_count.store(0); k->bar_gen.fetch_add(1); return true; }
    const auto t0 = std::chrono::steady_clock::now();
    while (k->bar_gen.load() == gen) {
        if (k->dead.load()) return false;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) { k->dead.store(1); return false; }
        usleep(50);
    }
    return !k->dead.load();
}
// shared objects are files under M2S_STUB_RCCL_DIR (default /dev/shm; a container's /dev/shm may be 64 MB — point it at /tmp then)
std::string obj_path(const std::string& name) {
    static const std::string dir = [] { const char* d = getenv("M2S_STUB_RCCL_DIR"); return std::string(d && *d ? d : "/dev/shm"); }();
    return dir + name;
}
int obj_open(const std::string& name, int flags) { return open(obj_path(name).c_str(), flags | O_CLOEXEC, 0600); }
void obj_unlink(const std::string& name) { unlink(obj_path(name).c_str()); }
void* map_shm(const std::string& name, size_t bytes, bool create) {
    const int fd = obj_open(name, create ? (O_CREAT | O_EXCL | O_RDWR) : O_RDWR);
    if (fd < 0) return nullptr;
    if (create && ftruncate(fd, (off_t)std::max<size_t>(bytes, 1)) != 0) { close(fd); obj_unlink(name); return nullptr; }
    void* p = mmap(nullptr, std::max<size_t>(bytes, 1), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}
void maybe_die_in_gather(Comm* c) {
    if (c->rank == c->die_rank && c->die_op >= 0 && (long)c->seq == c->die_op) { logf(c, "dying on purpose inside all-gather %llu", (unsigned long long)c->seq); _exit(9); }
}
void maybe_die(Comm* c) {   // in a group
    if (c->rank == c->die_rank && c->die_op < 0 && (long)c->groups + 1 == -c->die_op) { logf(c, "dying on purpose inside group %llu", (unsigned long long)c->groups); _exit(9); }
}

// One collective step.  `sends` = what this rank contributes (device pointers), `recvs` = what it expects (peer = source rank).
// kind / count are compared across ranks (all-gather); sends and receives are matched by (source, destination, order).
int run_step(Comm* c, uint32_t kind, uint64_t count, const std::vector<Op>& sends, const std::vector<Op>& recvs, hipStream_t st) {
    if (c->destroyed) return ncclInvalidUsage;
    maybe_die_in_gather(c);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;   // what the stream produced is what is sent
    if (sends.size() > (size_t)kMaxMsgs) return ncclInternalError;
    Slot& me = c->ctl->slot[c->rank];
    uint64_t total = 0;
    for (const Op& s : sends) total += s.bytes;
    const std::string mine = data_name(c, c->rank, c->seq);
    char* out = (char*)map_shm(mine, total, true);
    if (!out) return ncclSystemError;
    me.seq = c->seq; me.kind = kind; me.count = count; me.n_msgs = (uint32_t)sends.size(); me.shm_bytes = total;
    uint64_t off = 0;
    int rc = ncclSuccess;
    for (size_t i = 0; i < sends.size(); ++i) {
        me.msg[i] = Msg{ sends[i].peer, 0u, (uint64_t)sends[i].bytes, off };
        if (sends[i].bytes && hipMemcpy(out + off, sends[i].ptr, sends[i].bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
        off += sends[i].bytes;
    }
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (!barrier(c)) { munmap(out, std::max<uint64_t>(total, 1)); obj_unlink(mine); return ncclSystemError; }
    // everybody has published: same operation everywhere?
    for (int r = 0; r < c->nranks && rc == ncclSuccess; ++r) {
        const Slot& o = c->ctl->slot[r];
        if (o.seq != c->seq || o.kind != kind || (kind == 1 && o.count != count)) {
            fprintf(stderr, "[rccl stub] rank %d op %llu: rank %d is in another collective (kind %u count %llu seq %llu vs kind %u count %llu)\n", c->rank,
                    (unsigned long long)c->seq, r, o.kind, (unsigned long long)o.count, (unsigned long long)o.seq, kind, (unsigned long long)count);
            rc = ncclInvalidUsage;
        }
    }
    // receive: per source rank, its messages addressed to me, in its order
    std::vector<uint32_t> next((size_t)c->nranks, 0);
    std::vector<char*> maps((size_t)c->nranks, nullptr);
    for (const Op& r : recvs) {
        if (rc != ncclSuccess) break;
        Slot& o = c->ctl->slot[r.peer];
        uint32_t& i = next[(size_t)r.peer];
        while (i < o.n_msgs && o.msg[i].peer != c->rank) ++i;
        if (i == o.n_msgs) { fprintf(stderr, "[rccl stub] rank %d op %llu: ncclRecv from %d has no matching ncclSend\n", c->rank, (unsigned long long)c->seq, r.peer); rc = ncclInvalidUsage; break; }
        if (o.msg[i].bytes != r.bytes) { fprintf(stderr, "[rccl stub] rank %d op %llu: ncclRecv of %zu bytes from %d meets an ncclSend of %llu\n", c->rank, (unsigned long long)c->seq, r.bytes, r.peer, (unsigned long long)o.msg[i].bytes); rc = ncclInvalidArgument; break; }
        if (!maps[(size_t)r.peer]) maps[(size_t)r.peer] = r.peer == c->rank ? out : (char*)map_shm(data_name(c, r.peer, c->seq), o.shm_bytes, false);
        if (!maps[(size_t)r.peer]) { rc = ncclSystemError; break; }
        if (r.bytes && hipMemcpy(r.ptr, maps[(size_t)r.peer] + o.msg[i].off, r.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        o.msg[i].consumed = 1;
        ++i;
    }
    for (int r = 0; r < c->nranks; ++r) if (maps[(size_t)r] && r != c->rank) munmap(maps[(size_t)r], std::max<uint64_t>(c->ctl->slot[r].shm_bytes, 1));
    if (rc != ncclSuccess) c->ctl->dead.store(1);
    const bool ok = barrier(c);
    if (ok && rc == ncclSuccess)
        for (uint32_t i = 0; i < me.n_msgs; ++i)
            if (!me.msg[i].consumed) { fprintf(stderr, "[rccl stub] rank %d op %llu: ncclSend to %d was never received\n", c->rank, (unsigned long long)c->seq, me.msg[i].peer); rc = ncclInvalidUsage; }
    munmap(out, std::max<uint64_t>(total, 1));
    obj_unlink(mine);
    ++c->seq;
    return !ok && rc == ncclSuccess ? ncclSystemError : rc;
}
}  // namespace

struct StubId { uint32_t magic; char token[60]; };   // what the 128-byte ncclUniqueId carries here

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;   // rccl.h: passed BY VALUE to ncclCommInitRank

int ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    StubId* s = reinterpret_cast<StubId*>(id);
    s->magic = kMagic;
    snprintf(s->token, sizeof s->token, "m2sstub_%d_%u_%llx", (int)getpid(), g_id_counter.fetch_add(1),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    Ctl* k = (Ctl*)map_shm(std::string("/") + s->token, sizeof(Ctl), true);   // (zero-filled by ftruncate)
    if (!k) return ncclSystemError;
    k->magic.store(kMagic);
    munmap(k, sizeof(Ctl));
    return ncclSuccess;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    const StubId* s = reinterpret_cast<const StubId*>(&id);
    if (s->magic != kMagic) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->token.assign(s->token, strnlen(s->token, sizeof s->token));
    c->rank = rank; c->nranks = nranks;
    if (const char* t = getenv("M2S_STUB_RCCL_TIMEOUT")) c->timeout_s = atof(t) > 0 ? atof(t) : 30.0;
    if (const char* l = getenv("M2S_STUB_RCCL_LOG")) c->log = fopen((std::string(l) + "." + std::to_string(rank)).c_str(), "a");
    if (const char* d = getenv("M2S_STUB_RCCL_DIE")) { int r = -1; long op = -1; if (sscanf(d, "%d:%ld", &r, &op) == 2) { c->die_rank = r; c->die_op = op; } }
    c->ctl = (Ctl*)map_shm("/" + c->token, sizeof(Ctl), false);
    if (!c->ctl || c->ctl->magic.load() != kMagic) { delete c; return ncclSystemError; }
    uint32_t expect = 0;
    if (!c->ctl->nranks.compare_exchange_strong(expect, (uint32_t)nranks) && expect != (uint32_t)nranks) { delete c; return ncclInvalidArgument; }
    const uint32_t order = c->ctl->joined.fetch_add(1);
    logf(c, "ncclCommInitRank nranks %d rank %d pid %d", nranks, rank, (int)getpid());
    if (!barrier(c)) { logf(c, "bootstrap: some rank never arrived"); delete c; return ncclSystemError; }   // every rank must show up
    if (order + 1 == (uint32_t)nranks) obj_unlink(("/" + c->token));   // all mapped: the name can go (the memory stays until unmapped)
    *comm = c;
    return ncclSuccess;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c || c->destroyed) return ncclInvalidArgument;
    logf(c, "ncclCommDestroy after %llu all-gathers, %llu groups", (unsigned long long)c->seq, (unsigned long long)c->groups);
    c->destroyed = true;
    if (c->log) fclose(c->log);
    munmap(c->ctl, sizeof(Ctl));
    delete c;
    return ncclSuccess;
}

int ncclAllGather(const void* sendbuff, void* recvbuff, size_t count, int datatype, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    const size_t bytes = count * dtype_size(datatype);
    if (!c || !dtype_size(datatype) || (bytes && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
    if (g_depth) return ncclInvalidUsage;   // (m2s_dist.cpp never groups an all-gather)
    logf(c, "ncclAllGather op %llu count %zu dtype %d", (unsigned long long)c->seq, count, datatype);
    std::vector<Op> sends, recvs;
    for (int r = 0; r < c->nranks; ++r) {
        sends.push_back(Op{ true, const_cast<void*>(sendbuff), bytes, r, c, st });
        recvs.push_back(Op{ false, (char*)recvbuff + (size_t)r * bytes, bytes, r, c, st });
    }
    return run_step(c, 1, bytes, sends, recvs, st);
}

// Point-to-point messages need no rendezvous of the whole communicator (a rank with nothing to send or receive takes no part, as
// in RCCL): message k from rank s to rank d is one shared-memory object named after (s, d, k) — header {ready, consumed, bytes} +
// payload.  The sender creates and fills it; the receiver waits for it, checks the size against its ncclRecv, copies, marks it
// consumed; the sender waits for that mark and removes the object.  All waits are bounded.
struct MsgHdr { std::atomic<uint32_t> ready, consumed; uint64_t bytes; };
static std::string msg_name(const Comm* c, int src, int dst, uint64_t k) {
    return "/" + c->token + ".p" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(k);
}
static bool timed_out(Comm* c, std::chrono::steady_clock::time_point t0) {
    if (c->ctl->dead.load()) return true;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) { c->ctl->dead.store(1); return true; }
    return false;
}
static int flush_group() {
    std::vector<Op> ops;
    ops.swap(g_ops);
    if (ops.empty()) return ncclSuccess;
    Comm* c = ops[0].comm;
    if (c->destroyed) return ncclInvalidUsage;
    for (const Op& o : ops) if (o.comm != c) return ncclInvalidUsage;
    maybe_die(c);
    size_t ns = 0;
    for (const Op& o : ops) ns += o.send ? 1 : 0;
    logf(c, "group %llu: %zu sends %zu recvs", (unsigned long long)c->groups, ns, ops.size() - ns);
    ++c->groups;
    hipStream_t last = nullptr;
    bool first = true;
    for (const Op& o : ops)   // what the stream(s) produced is what is sent
        if (first || o.st != last) { if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError; last = o.st; first = false; }
    struct Out { std::string name; MsgHdr* h; size_t map_bytes; int peer; };
    std::vector<Out> outs;
    int rc = ncclSuccess;
    for (const Op& o : ops) {       // all sends first: nobody's receive waits for a send that sits behind one of my receives
        if (!o.send || rc != ncclSuccess) continue;
        Out out{ msg_name(c, c->rank, o.peer, c->sent[o.peer]++), nullptr, sizeof(MsgHdr) + o.bytes, o.peer };
        out.h = (MsgHdr*)map_shm(out.name, out.map_bytes, true);
        if (!out.h) { rc = ncclSystemError; break; }
        out.h->bytes = o.bytes;
        if (o.bytes && hipMemcpy((char*)(out.h + 1), o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) rc = ncclUnhandledCudaError;
        out.h->ready.store(1, std::memory_order_release);
        outs.push_back(out);
    }
    for (const Op& o : ops) {
        if (o.send || rc != ncclSuccess) continue;
        const std::string name = msg_name(c, o.peer, c->rank, c->recvd[o.peer]++);
        const auto t0 = std::chrono::steady_clock::now();
        MsgHdr* h = nullptr;
        size_t map_bytes = 0;
        for (;;) {
            const int fd = obj_open(name, O_RDWR);
            if (fd >= 0) {
                struct stat sb;
                if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(MsgHdr)) {
                    map_bytes = (size_t)sb.st_size;
                    void* p = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                    close(fd);
                    if (p != MAP_FAILED) { h = (MsgHdr*)p; break; }
                } else close(fd);
            }
            if (timed_out(c, t0)) break;
            usleep(50);
        }
        if (!h) { fprintf(stderr, "[rccl stub] rank %d: ncclRecv of %zu bytes from rank %d: no matching ncclSend arrived\n", c->rank, o.bytes, o.peer); rc = ncclSystemError; break; }
        while (!h->ready.load(std::memory_order_acquire)) { if (timed_out(c, t0)) { rc = ncclSystemError; break; } usleep(20); }
        if (rc == ncclSuccess && h->bytes != o.bytes) {
            fprintf(stderr, "[rccl stub] rank %d: ncclRecv of %zu bytes from rank %d meets an ncclSend of %llu bytes\n", c->rank, o.bytes, o.peer, (unsigned long long)h->bytes);
            rc = ncclInvalidArgument;
        }
        if (rc == ncclSuccess && o.bytes && hipMemcpy(o.ptr, (char*)(h + 1), o.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        h->consumed.store(rc == ncclSuccess ? 1u : 2u, std::memory_order_release);
        munmap(h, map_bytes);
    }
    if (rc != ncclSuccess) c->ctl->dead.store(1);
    for (Out& out : outs) {
        const auto t0 = std::chrono::steady_clock::now();
        while (rc == ncclSuccess && !out.h->consumed.load(std::memory_order_acquire))
            if (timed_out(c, t0)) { fprintf(stderr, "[rccl stub] rank %d: ncclSend to rank %d was never received\n", c->rank, out.peer); rc = ncclSystemError; }
            else usleep(20);
        if (rc == ncclSuccess && out.h->consumed.load() == 2u) rc = ncclInvalidArgument;
        munmap(out.h, out.map_bytes);
        obj_unlink(out.name);
    }
    return rc;
}

int ncclGroupStart() { ++g_depth; return ncclSuccess; }
int ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth) return ncclSuccess;
    return flush_group();
}

static int p2p(bool send, void* buff, size_t count, int datatype, int peer, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    const size_t bytes = count * dtype_size(datatype);
    if (!c || !dtype_size(datatype) || peer < 0 || peer >= c->nranks || (bytes && !buff)) return ncclInvalidArgument;
    logf(c, "%s group %llu peer %d bytes %zu", send ? "ncclSend" : "ncclRecv", (unsigned long long)c->groups, peer, bytes);
    g_ops.push_back(Op{ send, buff, bytes, peer, c, st });
    return g_depth ? ncclSuccess : flush_group();   // (a bare call is a group of one)
}
int ncclSend(const void* sendbuff, size_t count, int datatype, int peer, void* comm, hipStream_t st) { return p2p(true, const_cast<void*>(sendbuff), count, datatype, peer, comm, st); }
int ncclRecv(void* recvbuff, size_t count, int datatype, int peer, void* comm, hipStream_t st) { return p2p(false, recvbuff, count, datatype, peer, comm, st); }

const char* ncclGetErrorString(int r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "[rccl stub] unhandled HIP error";
        case ncclSystemError: return "[rccl stub] system error (a rank never arrived, or shared memory failed)";
        case ncclInternalError: return "[rccl stub] internal error";
        case ncclInvalidArgument: return "[rccl stub] invalid argument";
        case ncclInvalidUsage: return "[rccl stub] invalid usage (ranks disagree about the collective, or unmatched send / receive)";
        default: return "[rccl stub] unknown error";
    }
}

}  // extern "C"
