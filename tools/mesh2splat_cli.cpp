// mesh2splat — headless command-line converter.  The reference has no CLI (src/utils/argparser.hpp is
// dead code and main() ignores argv); its workflows are the GUI's LoadModel -> RunConversion -> SavePLY event
// sequence (src/renderer/guiRendererConcreteMediator.cpp:11-29,51-57,111-115) and the batch state machine
// Idle -> Loading -> Converting -> Exporting per file (:134-251).  This tool runs both through the C ABI (include/m2s.h).
//
//   mesh2splat in.glb out.ply [options]                      one file
//   mesh2splat --batch in_dir --out out_dir [options]        every *.glb of in_dir -> out_dir/<name>.ply
//
//   --density R | --quality q [--max-res 1024|2048|4096]     defaults are the GUI's: quality 0.5, max-res 1024 ->
//                                                            R = int(16 + q*(maxRes-16)) = 520 (ImGuiUI.cpp:512, main.cpp:26)
//   --std s        gaussian std, default 0.65 (main.cpp:26)   --format 0|1|2   standard 3DGS / PBR / compressed PBR
//   --cap n        0 = unlimited, default: the reference formula        --pipeline auto|multipass
//   --device d     first HIP device to use                     --timing  per-stage wall clock
//   --gpus N       one process per GPU (devices d .. d+N-1):
//                    one file : the triangle list is cut into N fragment-balanced ranges (m2s_dist_shard_ranges), every
//                               rank converts its range, the N counters are exchanged (8 bytes each: through the shared
//                               mapping the forked ranks inherit) and every rank writes ITS rows of the one output file at
//                               their final offset (m2s_export_ply_slice) — no record leaves its GPU, no RCCL needed.
//                               --gather: instead, the counters go through RCCL (m2s_dist_all_gather_counts) and the blocks
//                               are concatenated on rank 0 over xGMI (m2s_dist_gather_records), which writes the file.
//                               --threads: the N ranks are THREADS of this process instead of forked processes (the reference's
//                               shape: one executable); with --gather the exchange then runs on the in-process transport
//                               (m2s_dist_local_id: hipMemcpyPeer over the same links, no RCCL).
//                    --batch  : file k goes to GPU k mod N (independent replicas, no exchange).
// One GPU, --batch: load(k+1) | upload + convert(k) | export(k-1) overlap (a loader thread, an exporter thread, two
// contexts used alternately so that file k's records stay intact while file k+1 converts).
#include <dirent.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/m2s.h"

namespace {

struct Options {
    std::string in, out, batch_dir, out_dir;
    double quality = 0.5, std_dev = 0.65;
    long max_res = 1024, density = -1, device = 0, cap = -1, format = 0, gpus = 1;
    int pipeline = M2S_PIPELINE_AUTO;
    bool timing = false, gather = false, force_sharded = false;   // force_sharded: the --gpus code path with one rank (tests)
    bool one_device = false;                                       // tests: every rank on --device (several processes share one GPU)
    bool threads = false;                                          // ranks as threads of this process
    uint32_t R() const { return density > 0 ? (uint32_t)density : (uint32_t)(int)(16 + quality * (double)(max_res - 16)); }  // ImGuiUI.cpp:512
};

void usage() {
    std::fprintf(stderr,
                 "usage: mesh2splat in.glb out.ply [options]\n"
                 "       mesh2splat --batch in_dir --out out_dir [options]\n"
                 "options: [--density R | --quality q [--max-res M]] [--std s] [--format 0|1|2] [--device d] [--gpus N [--gather]]\n"
                 "         [--cap n (0 = unlimited, default: reference formula)] [--pipeline auto|multipass] [--timing]\n");
}

using Clock = std::chrono::steady_clock;
double ms_between(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

// ---- one file on one GPU ---------------------------------------------------------------------------------------------
int convert_one(const Options& o) {
    const auto t0 = Clock::now();
    // the HIP runtime and the context come up on a second thread while the .glb is parsed and its images decoded
    m2s_ctx* ctx = nullptr;
    m2s_status create_status = M2S_OK;
    double create_ms = 0;
    std::thread init([&] {
        const auto a = Clock::now();
        create_status = m2s_create((int)o.device, &ctx);
        if (create_status == M2S_OK) (void)m2s_prepare(ctx, M2S_PREPARE_UPLOAD | M2S_PREPARE_EXPORT | M2S_PREPARE_KERNELS);   // pinned buffers, code objects: off the critical path
        create_ms = ms_between(a, Clock::now());
    });
    m2s_host_scene* scene = nullptr;
    const m2s_status ls = m2s_load_glb(o.in.c_str(), &scene);
    const auto t1 = Clock::now();
    init.join();
    if (ls != M2S_OK) { std::fprintf(stderr, "%s\n", m2s_io_last_error()); if (ctx) m2s_destroy(ctx); return 1; }
    if (*m2s_host_scene_warnings(scene)) std::fprintf(stderr, "%s", m2s_host_scene_warnings(scene));
    if (create_status != M2S_OK) { std::fprintf(stderr, "%s\n", m2s_last_error(nullptr)); m2s_free_host_scene(scene); return 1; }
    const auto t1b = Clock::now();
    auto die = [&](const char* what) { std::fprintf(stderr, "%s: %s\n", what, m2s_last_error(ctx)); m2s_destroy(ctx); m2s_free_host_scene(scene); return 1; };
    if (m2s_set_pipeline(ctx, o.pipeline) != M2S_OK) return die("set_pipeline");
    if (m2s_set_max_gaussians(ctx, o.cap) != M2S_OK) return die("set_max_gaussians");
    (void)m2s_set_resolution_hint(ctx, o.R());   // the upload prepares for the conversion below
    if (m2s_upload_scene(ctx, m2s_host_scene_meshes(scene), m2s_host_scene_num_meshes(scene)) != M2S_OK) return die("upload");
    const auto t2 = Clock::now();
    uint64_t total = 0;
    const uint32_t R = o.R();
    if (m2s_convert(ctx, R, &total) != M2S_OK) return die("convert");
    const auto t3 = Clock::now();
    if (m2s_export_ply(ctx, o.out.c_str(), (uint32_t)o.format, (float)o.std_dev) != M2S_OK) return die("export");
    const auto t4 = Clock::now();
    std::printf("%s: %u mesh(es), %llu triangles, density %u -> %llu Gaussians (%llu stored) -> %s (format %ld)\n", o.in.c_str(),
                m2s_host_scene_num_meshes(scene), (unsigned long long)m2s_num_triangles(ctx), R, (unsigned long long)total,
                (unsigned long long)m2s_num_stored(ctx), o.out.c_str(), o.format);
    if (o.timing) {
        float up[4] = { 0, 0, 0, 0 };
        (void)m2s_last_upload_ms(ctx, up);
        std::printf("load %.2f ms (HIP runtime + context %.2f ms, on a second thread; waited %.2f ms for it) | upload %.2f ms (geometry %.2f, textures %.2f, "
                    "allocations %.2f) | convert %.3f ms (first call, incl. buffer allocation) | export %.2f ms | total %.2f ms\n",
                    ms_between(t0, t1), create_ms, ms_between(t1, t1b), ms_between(t1b, t2), up[1], up[2], up[3], ms_between(t2, t3),
                    ms_between(t3, t4), ms_between(t0, t4));
    }
    m2s_destroy(ctx);
    m2s_free_host_scene(scene);
    return 0;
}

// ---- one file on N GPUs: fork N ranks AFTER the (CPU-only) load so that they share the host scene copy-on-write -----------
struct Shared {
    std::atomic<int> id_ready;
    uint8_t id[M2S_DIST_ID_BYTES];
    std::atomic<int> failed;
    // the counters of a sharded conversion whose ranks write their own rows: N x 8 bytes through this mapping (the ranks
    // are forked children of one process on one node); RCCL is only brought up when records have to move (--gather)
    std::atomic<int> arrived;
    std::atomic<unsigned long long> counts[64];
    std::atomic<int> stage[64];   // --gather: how far each rank has got (see rendezvous below)
};

int rank_main(const Options& o, const m2s_host_scene* scene, Shared* sh, int rank, int world) {
    const m2s_mesh* meshes = m2s_host_scene_meshes(scene);
    const uint32_t n_meshes = m2s_host_scene_num_meshes(scene);
    const uint32_t R = o.R();
    const int device = (int)o.device + (o.one_device ? 0 : rank);
    auto fail = [&](const char* what, const char* msg) { std::fprintf(stderr, "[rank %d] %s: %s\n", rank, what, msg); sh->failed.store(1); return 1; };
    // --gather: RCCL calls are collectives — a rank that fails before one must not leave the others blocked inside it
    // (ncclCommInitRank, the counter all-gather, the record exchange).  Before each of them every rank reports, through the
    // shared mapping, that it got there in one piece; a rank that sees `failed` leaves instead of entering.
    auto rendezvous = [&](int stage) -> bool {
        sh->stage[rank].store(stage);
        for (;;) {
            if (sh->failed.load()) return false;
            int behind = 0;
            for (int r = 0; r < world; ++r) behind += sh->stage[r].load() < stage;
            if (!behind) return true;
            usleep(100);
        }
    };
    const auto t0 = Clock::now();
    std::vector<uint64_t> first((size_t)world), count((size_t)world);
    if (m2s_dist_shard_ranges(meshes, n_meshes, R, world, first.data(), count.data()) != M2S_OK) return fail("shard_ranges", m2s_dist_last_error(nullptr));
    m2s_ctx* ctx = nullptr;
    if (m2s_create(device, &ctx) != M2S_OK) return fail("create", m2s_last_error(nullptr));
    m2s_dist* d = nullptr;
    if (o.gather) {   // records will move between GPUs: one RCCL communicator per rank, id through the shared mapping
        if (rank == 0 && !o.threads) {   // (with --threads the parent made the in-process group's id before starting the ranks)
            if (m2s_dist_unique_id(sh->id) != M2S_OK) return fail("unique_id", m2s_dist_last_error(nullptr));
            sh->id_ready.store(1);
        } else {
            while (!sh->id_ready.load()) { if (sh->failed.load()) return 1; usleep(200); }
        }
        if (!rendezvous(1)) return 1;                 // every rank has a context and the id: nobody is left alone in ncclCommInitRank
        if (m2s_dist_create(device, sh->id, rank, world, &d) != M2S_OK) return fail("dist_create", m2s_dist_last_error(nullptr));
    }
    const auto t1 = Clock::now();
    // the merged scene may exceed the reference's cap envelope: every rank stores all of its records, the global cap is
    // applied to the row ranges below (the merged buffer keeps the first `cap` records in canonical order)
    (void)m2s_set_pipeline(ctx, o.pipeline);
    (void)m2s_set_max_gaussians(ctx, 0);
    (void)m2s_set_triangle_range(ctx, first[(size_t)rank], count[(size_t)rank]);
    (void)m2s_set_resolution_hint(ctx, R);
    if (m2s_upload_scene(ctx, meshes, n_meshes) != M2S_OK) return fail("upload", m2s_last_error(ctx));
    const auto t2 = Clock::now();
    uint64_t total = 0;
    if (m2s_convert(ctx, R, &total) != M2S_OK) return fail("convert", m2s_last_error(ctx));
    std::vector<uint64_t> counts((size_t)world), keep((size_t)world), offs((size_t)world + 1);
    if (d) {
        if (!rendezvous(2)) return 1;                 // every rank has uploaded and converted
        if (m2s_dist_all_gather_counts(d, total, counts.data(), nullptr) != M2S_OK) return fail("all_gather_counts", m2s_dist_last_error(d));
    } else {
        sh->counts[rank].store(total);
        sh->arrived.fetch_add(1);
        while (sh->arrived.load() < world) { if (sh->failed.load()) return 1; usleep(50); }
        for (int r = 0; r < world; ++r) counts[(size_t)r] = sh->counts[r].load();
    }
    uint64_t cap = 0;
    if (o.cap > 0) cap = (uint64_t)o.cap;
    else if (o.cap < 0) { const uint32_t mx = R * R * 6u * std::max<uint32_t>(1u, n_meshes); cap = std::min<uint32_t>(mx, 7000000u); }   // ConversionPass.cpp:21-24
    m2s_dist_clamp_to_cap(counts.data(), world, cap, keep.data());
    offs[0] = 0;
    for (int r = 0; r < world; ++r) offs[(size_t)r + 1] = offs[(size_t)r] + keep[(size_t)r];
    const auto t3 = Clock::now();
    uint64_t all = 0;
    for (int r = 0; r < world; ++r) all += counts[(size_t)r];
    if (!o.gather) {
        if (m2s_export_ply_slice(ctx, o.out.c_str(), (uint32_t)o.format, (float)o.std_dev, offs[(size_t)rank], keep[(size_t)rank], offs[(size_t)world]) != M2S_OK)
            return fail("export_slice", m2s_last_error(ctx));
    } else {
        // north-star variant: one RCCL exchange concatenates the blocks on rank 0, which then writes the whole file
        m2s_ctx* sink = nullptr;
        void* merged = nullptr;
        if (rank == 0) {
            if (m2s_create(device, &sink) != M2S_OK) return fail("create", m2s_last_error(nullptr));
            if (m2s_reserve_records(sink, offs[(size_t)world], &merged) != M2S_OK) return fail("reserve", m2s_last_error(sink));
        }
        if (!rendezvous(3)) return 1;                 // rank 0 has its receive buffer
        if (m2s_dist_gather_records(d, m2s_device_records(ctx), keep.data(), merged, 0, nullptr) != M2S_OK) return fail("gather_records", m2s_dist_last_error(d));
        if (m2s_dist_wait(d, nullptr) != M2S_OK) return fail("gather_records", m2s_dist_last_error(d));   // sends / receives have completed
        if (rank == 0) {
            if (m2s_set_records(sink, merged, offs[(size_t)world], R) != M2S_OK) return fail("set_records", m2s_last_error(sink));
            if (m2s_export_ply(sink, o.out.c_str(), (uint32_t)o.format, (float)o.std_dev) != M2S_OK) return fail("export", m2s_last_error(sink));
            m2s_destroy(sink);
        }
    }
    const auto t4 = Clock::now();
    if (rank == 0)
        std::printf("%s: %u mesh(es), density %u, %d GPUs -> %llu Gaussians (%llu stored) -> %s (format %ld, %s)\n", o.in.c_str(), n_meshes, R, world,
                    (unsigned long long)all, (unsigned long long)offs[(size_t)world], o.out.c_str(), o.format,
                    o.gather ? (o.threads ? "gathered on rank 0, in-process transport" : "gathered on rank 0 over RCCL") : "every rank wrote its rows");
    if (o.timing)
        std::printf("[rank %d] triangles [%llu, +%llu) -> %llu Gaussians | init (HIP%s) %.2f ms | upload %.2f ms | convert + counter exchange %.3f ms | export %.2f ms\n",
                    rank, (unsigned long long)first[(size_t)rank], (unsigned long long)count[(size_t)rank], (unsigned long long)total, o.gather ? " + RCCL" : "", ms_between(t0, t1),
                    ms_between(t1, t2), ms_between(t2, t3), ms_between(t3, t4));
    if (d) m2s_dist_destroy(d);
    m2s_destroy(ctx);
    return 0;
}

template <class F>
int fork_ranks(int world, F body) {
    std::fflush(stdout); std::fflush(stderr);
    std::vector<pid_t> pids;
    for (int r = 0; r < world; ++r) {
        const pid_t p = fork();
        if (p < 0) { std::perror("fork"); return 1; }
        if (p == 0) { const int rc = body(r); std::fflush(stdout); std::fflush(stderr); _exit(rc); }
        pids.push_back(p);
    }
    // a rank that dies (a signal, an abort inside a library) cannot tell the others: when one exits abnormally or non-zero the
    // parent gives the rest two seconds to notice `failed` by themselves and then ends them, so that nobody waits for ever
    int rc = 0;
    size_t left = pids.size();
    std::vector<char> done(pids.size(), 0);
    int grace_polls = -1;
    while (left) {
        bool progressed = false;
        for (size_t i = 0; i < pids.size(); ++i) {
            if (done[i]) continue;
            int st = 0;
            const pid_t w = waitpid(pids[i], &st, WNOHANG);
            if (w == 0) continue;
            done[i] = 1; --left; progressed = true;
            if (w < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) { rc = 1; if (grace_polls < 0) grace_polls = 2000; }
        }
        if (!left) break;
        if (grace_polls == 0) { for (size_t i = 0; i < pids.size(); ++i) if (!done[i]) kill(pids[i], SIGKILL); grace_polls = -2; }
        else if (grace_polls > 0) --grace_polls;
        if (!progressed) usleep(1000);
    }
    return rc;
}

int convert_sharded(const Options& o) {
    // NO HIP call in this process before the fork (a forked HIP runtime is unusable): the load is CPU-only
    m2s_host_scene* scene = nullptr;
    if (m2s_load_glb(o.in.c_str(), &scene) != M2S_OK) { std::fprintf(stderr, "%s\n", m2s_io_last_error()); return 1; }
    if (*m2s_host_scene_warnings(scene)) std::fprintf(stderr, "%s", m2s_host_scene_warnings(scene));
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) { std::perror("mmap"); return 1; }
    new (sh) Shared();
    sh->id_ready.store(0); sh->failed.store(0); sh->arrived.store(0);
    for (auto& st : sh->stage) st.store(0);
    std::remove(o.out.c_str());
    const int world = (int)o.gpus;
    int rc = 0;
    if (o.threads) {
        if (o.gather) {
            if (m2s_dist_local_id(world, sh->id) != M2S_OK) { std::fprintf(stderr, "local_id: %s\n", m2s_dist_last_error(nullptr)); return 1; }
            sh->id_ready.store(1);
        }
        std::vector<int> rcs((size_t)world, 1);
        std::vector<std::thread> ranks;
        for (int r = 0; r < world; ++r) ranks.emplace_back([&, r] { rcs[(size_t)r] = rank_main(o, scene, sh, r, world); });
        for (auto& t : ranks) t.join();
        for (int r = 0; r < world; ++r) rc |= rcs[(size_t)r];
    } else
        rc = fork_ranks(world, [&](int r) { return rank_main(o, scene, sh, r, world); });
    munmap(sh, sizeof(Shared));
    m2s_free_host_scene(scene);
    return rc;
}

// ---- batch -----------------------------------------------------------------------------------------------------------
template <class T>
class Channel {   // bounded blocking queue; close() ends the stream
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void push(T v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return q_.size() < cap_; }); q_.push_back(std::move(v)); cv_.notify_all(); }
    bool pop(T& v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !q_.empty() || closed_; }); if (q_.empty()) return false; v = std::move(q_.front()); q_.pop_front(); cv_.notify_all(); return true; }
    void close() { std::lock_guard<std::mutex> l(m_); closed_ = true; cv_.notify_all(); }
private:
    std::mutex m_; std::condition_variable cv_; std::deque<T> q_; size_t cap_; bool closed_ = false;
};

struct Loaded { std::string in, out; m2s_host_scene* scene = nullptr; double load_ms = 0; };
struct Converted { std::string in, out; int slot = 0; uint64_t total = 0; double load_ms = 0, upload_ms = 0, convert_ms = 0; };

int batch_on_device(const Options& o, const std::vector<std::pair<std::string, std::string>>& files, int device, int tag) {
    const auto t_begin = Clock::now();
    m2s_ctx* ctx[2] = { nullptr, nullptr };
    for (auto& c : ctx)
        if (m2s_create(device, &c) != M2S_OK) { std::fprintf(stderr, "%s\n", m2s_last_error(nullptr)); return 1; }
    for (auto c : ctx) { (void)m2s_set_pipeline(c, o.pipeline); (void)m2s_set_max_gaussians(c, o.cap); }
    Channel<Loaded> loaded(2);
    Channel<Converted> converted(1);
    std::mutex slot_m; std::condition_variable slot_cv; bool slot_busy[2] = { false, false };   // a context's records are being exported
    std::atomic<int> failures{ 0 };
    std::thread loader([&] {
        for (const auto& f : files) {
            Loaded l; l.in = f.first; l.out = f.second;
            const auto a = Clock::now();
            if (m2s_load_glb(l.in.c_str(), &l.scene) != M2S_OK) { std::fprintf(stderr, "%s: %s\n", l.in.c_str(), m2s_io_last_error()); ++failures; continue; }
            l.load_ms = ms_between(a, Clock::now());
            loaded.push(std::move(l));
        }
        loaded.close();
    });
    std::thread exporter([&] {
        Converted c;
        while (converted.pop(c)) {
            const auto a = Clock::now();
            if (m2s_export_ply(ctx[c.slot], c.out.c_str(), (uint32_t)o.format, (float)o.std_dev) != M2S_OK) {
                std::fprintf(stderr, "%s: export: %s\n", c.in.c_str(), m2s_last_error(ctx[c.slot])); ++failures;
            }
            const double ex = ms_between(a, Clock::now());
            std::printf("[gpu %d] %s -> %s: %llu Gaussians | load %.1f ms, upload %.1f ms, convert %.2f ms, export %.1f ms\n", tag, c.in.c_str(), c.out.c_str(),
                        (unsigned long long)c.total, c.load_ms, c.upload_ms, c.convert_ms, ex);
            { std::lock_guard<std::mutex> l(slot_m); slot_busy[c.slot] = false; }
            slot_cv.notify_all();
        }
    });
    Loaded l;
    int k = 0, done = 0;
    while (loaded.pop(l)) {
        const int slot = k++ & 1;
        { std::unique_lock<std::mutex> lk(slot_m); slot_cv.wait(lk, [&] { return !slot_busy[slot]; }); slot_busy[slot] = true; }
        Converted c; c.in = l.in; c.out = l.out; c.slot = slot; c.load_ms = l.load_ms;
        auto a = Clock::now();
        (void)m2s_set_resolution_hint(ctx[slot], o.R());
        bool ok = m2s_upload_scene(ctx[slot], m2s_host_scene_meshes(l.scene), m2s_host_scene_num_meshes(l.scene)) == M2S_OK;
        c.upload_ms = ms_between(a, Clock::now());
        a = Clock::now();
        ok = ok && m2s_convert(ctx[slot], o.R(), &c.total) == M2S_OK;
        c.convert_ms = ms_between(a, Clock::now());
        m2s_free_host_scene(l.scene);
        if (!ok) {
            std::fprintf(stderr, "%s: %s\n", l.in.c_str(), m2s_last_error(ctx[slot])); ++failures;
            { std::lock_guard<std::mutex> lk(slot_m); slot_busy[slot] = false; }
            slot_cv.notify_all();
            continue;
        }
        converted.push(std::move(c));
        ++done;
    }
    converted.close();
    loader.join();
    exporter.join();
    const double total_s = ms_between(t_begin, Clock::now()) * 1e-3;
    std::printf("[gpu %d] %d file(s) in %.3f s = %.2f meshes/s (load | upload + convert | export overlapped)\n", tag, done, total_s, done / std::max(total_s, 1e-9));
    for (auto c : ctx) m2s_destroy(c);
    return failures.load() ? 1 : 0;
}

int convert_batch(const Options& o) {
    std::vector<std::pair<std::string, std::string>> files;
    DIR* dir = opendir(o.batch_dir.c_str());
    if (!dir) { std::perror(o.batch_dir.c_str()); return 1; }
    while (dirent* e = readdir(dir)) {
        const std::string n = e->d_name;
        if (n.size() > 4 && n.compare(n.size() - 4, 4, ".glb") == 0)
            files.emplace_back(o.batch_dir + "/" + n, o.out_dir + "/" + n.substr(0, n.size() - 4) + ".ply");
    }
    closedir(dir);
    std::sort(files.begin(), files.end());
    if (files.empty()) { std::fprintf(stderr, "no .glb files in %s\n", o.batch_dir.c_str()); return 1; }
    if (o.gpus <= 1) return batch_on_device(o, files, (int)o.device, (int)o.device);
    // one process per GPU, file k on GPU k mod N: independent replicas (fork before any HIP call)
    const int world = (int)o.gpus;
    return fork_ranks(world, [&](int r) {
        std::vector<std::pair<std::string, std::string>> mine;
        for (size_t k = (size_t)r; k < files.size(); k += (size_t)world) mine.push_back(files[k]);
        return mine.empty() ? 0 : batch_on_device(o, mine, (int)o.device + r, (int)o.device + r);
    });
}

}  // namespace

int main(int argc, char** argv) {
    // --gpus N: one process per GPU over RCCL.  The host driver only supports dmabuf IPC; say so before any HIP runtime call of this
    // process or its forks (without it RCCL's bring-up fails with `hipIpcGetMemHandle: invalid argument`).  Never overrides the caller's value.
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    Options o;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (a == "--density") o.density = std::atol(next());
        else if (a == "--quality") o.quality = std::atof(next());
        else if (a == "--max-res") o.max_res = std::atol(next());
        else if (a == "--std") o.std_dev = std::atof(next());
        else if (a == "--format") o.format = std::atol(next());
        else if (a == "--device") o.device = std::atol(next());
        else if (a == "--cap") o.cap = std::atol(next());
        else if (a == "--gpus") o.gpus = std::atol(next());
        else if (a == "--gather") o.gather = true;
        else if (a == "--force-sharded") o.force_sharded = true;
        else if (a == "--one-device") o.one_device = true;
        else if (a == "--threads") o.threads = true;
        else if (a == "--batch") o.batch_dir = next();
        else if (a == "--out") o.out_dir = next();
        else if (a == "--pipeline") o.pipeline = std::string(next()) == "multipass" ? M2S_PIPELINE_MULTIPASS : M2S_PIPELINE_AUTO;
        else if (a == "--timing") o.timing = true;
        else if (!a.empty() && a[0] == '-') { usage(); return 2; }
        else pos.push_back(a);
    }
    if (o.gpus < 1 || o.gpus > 64) { usage(); return 2; }
    if (!o.batch_dir.empty()) {
        if (o.out_dir.empty() || !pos.empty()) { usage(); return 2; }
        return convert_batch(o);
    }
    if (pos.size() != 2) { usage(); return 2; }
    o.in = pos[0]; o.out = pos[1];
    return (o.gpus > 1 || o.force_sharded) ? convert_sharded(o) : convert_one(o);
}
