// (test infrastructure) The round-6 soak of the serve path as a C++ program for ThreadSanitizer: the library's host side (csrc/ built
// `--cuda-host-only`) on the stand-in runtime of hip_host_stub.cpp, driven from several threads at once through the C ABI --
//   * request threads (four by default), each with its own engine pair and stream-less `cv_process_image_v2` calls (the request slots of the Python class),
//   * two more request threads SHARING one engine pair (the engine's own mutexes: one forward per model at a time, one staging block),
//   * a thread running batch forwards of both models on that shared pair (what `process_images` does beside the slots),
//   * a thread that creates, loads, uses and destroys further engines the whole time (model loads, workspace growth, the block cache
//     taking and handing out blocks, graphs being buried), and calls cv_trim_memory between rounds.
// Kernels do not run; what runs is every lock, counter, table and cache the threads share (engine.h: capture_mutex, legacy_mutex,
// load_mutex, graph_mutex + the graveyard, the block cache, per-engine mutexes, thread-local error strings).  A data race in any of it
// is a ThreadSanitizer report and a non-zero exit (tests/test_engine_host_sanitizers.py; the full-size run: profiles/r06_tsan_engine_threads.txt).
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "chessvision_hip.h"

namespace {

struct Blob {
    std::vector<cv_param_t> table;
    std::vector<std::string> names;
    std::vector<std::vector<float>> data;
};

bool read_blob(const char* path, Blob& b) {                              // the format tests/test_c_abi_consumer.py writes
    FILE* f = std::fopen(path, "rb");
    if (!f) { std::perror(path); return false; }
    int32_t n = 0;
    if (std::fread(&n, 4, 1, f) != 1) return false;
    b.table.resize((size_t)n); b.names.resize((size_t)n); b.data.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
        int32_t len = 0, ndim = 0;
        if (std::fread(&len, 4, 1, f) != 1) return false;
        b.names[i].resize((size_t)len);
        if (std::fread(&b.names[i][0], 1, (size_t)len, f) != (size_t)len || std::fread(&ndim, 4, 1, f) != 1) return false;
        if (std::fread(b.table[i].shape, 8, 4, f) != 4) return false;
        size_t numel = 1;
        for (int d = 0; d < ndim; ++d) numel *= (size_t)b.table[i].shape[d];
        b.data[i].resize(numel);
        if (std::fread(b.data[i].data(), sizeof(float), numel, f) != numel) return false;
        b.table[i].ndim = ndim;
    }
    std::fclose(f);
    for (int i = 0; i < n; ++i) { b.table[i].name = b.names[i].c_str(); b.table[i].data = b.data[i].data(); }
    return true;
}

std::atomic<int> g_failures{0};
std::atomic<long> g_requests{0}, g_batches{0}, g_engines{0};

#define MUST(call)                                                                                         \
    do {                                                                                                   \
        const int rc_ = (call);                                                                            \
        if (rc_ != CV_OK) { std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cv_last_error()); ++g_failures; return; } \
    } while (0)

struct Request {                                                         // the buffers of one cv_process_image_v2 call
    std::vector<float> logits = std::vector<float>(256 * 256), probs = std::vector<float>(64 * 13);
    std::vector<uint8_t> mask = std::vector<uint8_t>(256 * 256), board = std::vector<uint8_t>(512 * 512), crops = std::vector<uint8_t>(64 * 64 * 64);
    cv_image_result_t res;
    Request() {
        std::memset(&res, 0, sizeof res);
        res.logits = logits.data(); res.mask = mask.data(); res.board = board.data(); res.probabilities = probs.data(); res.squares = crops.data();
    }
};

cv_engine_t* make_engine(int precision, const Blob& unet, const Blob& resnet) {
    cv_engine_t* e = nullptr;
    if (cv_engine_create(0, precision, &e) != CV_OK) return nullptr;
    if (cv_engine_set_chunk(e, 4, 512) != CV_OK || cv_load_unet(e, unet.table.data(), (int)unet.table.size()) != CV_OK ||
        cv_load_resnet18(e, resnet.table.data(), (int)resnet.table.size()) != CV_OK) {
        std::fprintf(stderr, "engine load: %s\n", cv_last_error());
        (void)cv_engine_destroy(e);
        return nullptr;
    }
    ++g_engines;
    return e;
}

std::atomic<int> g_side_running{0};                                     // the batch thread and the loader thread still at work

void request_loop(cv_engine_t* eng, const std::vector<uint8_t>& photo, int calls) {
    Request r;
    for (int k = 0; k < calls || (g_side_running.load() > 0 && k < 100000); ++k) {       // keep serving while the side threads work
        MUST(cv_process_image_v2(eng, eng, photo.data(), 512, 512, 0.5f, k & 1, 1, &r.res, sizeof r.res, nullptr));
        if (k % 7 == 0 && cv_process_image_v2(eng, eng, nullptr, 512, 512, 0.5f, 0, 1, &r.res, sizeof r.res, nullptr) == CV_OK) {   // an error path, with its
            std::fprintf(stderr, "null photo accepted\n"); ++g_failures; return;                                                      // thread-local message
        }
        ++g_requests;
        if (k >= calls) std::this_thread::sleep_for(std::chrono::milliseconds(2));      // past the asked-for count: keep company, leave the CPUs to the side threads
    }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc == 2 && std::strcmp(argv[1], "--race-selftest") == 0) {     // negative control: an unsynchronised counter must be REPORTED
        static int plain = 0;
        std::thread a([] { for (int i = 0; i < 100000; ++i) ++plain; }), b([] { for (int i = 0; i < 100000; ++i) ++plain; });
        a.join(); b.join();
        std::printf("selftest counter %d\n", plain);
        return 0;
    }
    if (argc < 3) { std::fprintf(stderr, "usage: engine_threads <unet.blob> <resnet.blob> [calls] [slots] [loader engines] [batch rounds]\n"); return 2; }
    const int calls = argc > 3 ? std::atoi(argv[3]) : 24, n_slots = argc > 4 ? std::atoi(argv[4]) : 4, n_loads = argc > 5 ? std::atoi(argv[5]) : 4,
              n_rounds = argc > 6 ? std::atoi(argv[6]) : 12;
    Blob unet, resnet;
    if (!read_blob(argv[1], unet) || !read_blob(argv[2], resnet)) return 2;
    std::vector<uint8_t> photo((size_t)512 * 512 * 3);
    for (size_t i = 0; i < photo.size(); ++i) photo[i] = (uint8_t)((i * 2654435761u) >> 24);

    // slots: one engine pair per request thread, loaded one after the other by the main thread (as ChessVision._build_slot does)
    std::vector<cv_engine_t*> slots;
    for (int t = 0; t < n_slots; ++t) slots.push_back(make_engine(t % 2 ? CV_PREC_F16R : CV_PREC_F16X3, unet, resnet));
    cv_engine_t* shared = make_engine(CV_PREC_F16X3, unet, resnet);
    for (cv_engine_t* e : slots) if (!e) return 1;
    if (!shared) return 1;

    g_side_running.store(2);
    std::vector<std::thread> workers;
    for (int t = 0; t < n_slots; ++t) workers.emplace_back(request_loop, slots[t], std::cref(photo), calls);
    for (int t = 0; t < 2; ++t) workers.emplace_back(request_loop, shared, std::cref(photo), calls);
    std::thread batcher([&] {                                            // batch forwards on the shared pair, sizes that regrow its workspace
        std::vector<float> x((size_t)6 * 3 * 256 * 256, 0.25f), lg((size_t)6 * 256 * 256), sq((size_t)700 * 64 * 64, 0.5f), out((size_t)700 * 13);
        struct Done { ~Done() { --g_side_running; } } done;
        for (int k = 0; k < n_rounds; ++k) {
            const int b = 1 + k % 6, s = 64 + (k * 131) % 600;
            MUST(cv_unet_forward(shared, x.data(), b, lg.data(), nullptr));
            MUST(cv_resnet18_forward(shared, sq.data(), s, out.data(), nullptr));
            size_t ws = 0;
            MUST(cv_engine_workspace_bytes(shared, &ws));
            (void)cv_engine_numeric_status(shared, nullptr);
            ++g_batches;
        }
    });
    std::thread loader([&] {                                             // engines that come and go beside all of it
        Request r;
        struct Done { ~Done() { --g_side_running; } } done;
        for (int k = 0; k < n_loads; ++k) {
            cv_engine_t* e = make_engine(k % 3 == 0 ? CV_PREC_F32 : k % 3 == 1 ? CV_PREC_F16 : CV_PREC_F16X3, unet, resnet);
            if (!e) { ++g_failures; return; }
            for (int i = 0; i < 2; ++i) MUST(cv_process_image_v2(e, e, photo.data(), 512, 512, 0.5f, 0, 1, &r.res, sizeof r.res, nullptr));
            MUST(cv_engine_destroy(e));
            if (k % 2) { size_t freed = 0; MUST(cv_trim_memory(&freed)); }
        }
    });
    for (auto& w : workers) w.join();
    batcher.join();
    loader.join();
    for (cv_engine_t* e : slots) if (cv_engine_destroy(e) != CV_OK) ++g_failures;
    if (cv_engine_destroy(shared) != CV_OK) ++g_failures;
    size_t freed = 0;
    if (cv_trim_memory(&freed) != CV_OK) ++g_failures;
    std::printf("engine threads: %ld requests from %d threads, %ld batch rounds, %ld engines loaded, %d failures\n", g_requests.load(), n_slots + 2,
                g_batches.load(), g_engines.load(), g_failures.load());
    return g_failures.load() ? 1 : 0;
}
