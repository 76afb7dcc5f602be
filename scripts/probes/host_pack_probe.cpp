// scripts/probes/host_pack_probe.cpp -- priced before building (round 6): the row-selective pass of an index that is not
// resident moves the DISTINCT looked-up rows of a unit host -> HBM with a copy KERNEL that reads the pinned file mapping
// (fetch_kernels.hip: gather_copy_kernel; 128-byte read requests on the link: 50.9 GB/s while it runs, the 184 GB pass
// 49.1 GB/s overall).  The copy engines move LINEAR memory at 57.6 GB/s (profiles/r04_h2d_probe.txt).  Would host threads
// that pack a unit's rows into a pinned staging buffer, one unit ahead of the DMA, deliver that rate?
//   hipcc -O2 --offload-arch=gfx950 scripts/probes/host_pack_probe.cpp -o scripts/probes/host_pack_probe -lpthread
//   scripts/probes/host_pack_probe [file GiB = 48] [units = 48]
// The stand-in for the file: anonymous memory in 4 KiB pages (what a page-cache mapping is), rows of 1568 bytes (C3).
// A unit: the rows of 1 / 254 of ... here: a consecutive range of rows of which 21 % are looked up (the 10k-query
// batch against the 184 GB file: 39.0 of 184 GB), ascending -- what gather_list_kernel hands over.
#include <hip/hip_runtime.h>

#include <sched.h>
#include <sys/mman.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<int> cpus_of_node(int node) {
    std::vector<int> out;
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string s;
    if (!std::getline(f, s)) return out;
    size_t p = 0;
    while (p < s.size()) {
        size_t e = s.find(',', p);
        if (e == std::string::npos) e = s.size();
        const std::string r = s.substr(p, e - p);
        const size_t d = r.find('-');
        const int a = std::atoi(r.c_str()), b = d == std::string::npos ? a : std::atoi(r.c_str() + d + 1);
        for (int c = a; c <= b; ++c) out.push_back(c);
        p = e + 1;
    }
    return out;
}

static int gpu_node() {
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, sizeof bdf, 0) != hipSuccess) return -1;
    for (char* c = bdf; *c; ++c) *c = (char)tolower(*c);
    std::ifstream f(std::string("/sys/bus/pci/devices/") + bdf + "/numa_node");
    int n = -1;
    f >> n;
    return n;
}

// a pool of T threads that run fn(job) for job in [0, njobs), jobs handed out by an atomic counter
struct Pool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv, done_cv;
    std::function<void(size_t)> fn;
    std::atomic<size_t> next{0};
    size_t njobs = 0, gen = 0, active = 0;
    bool stop = false;
    Pool(size_t T, const std::vector<int>& cpus) {
        for (size_t t = 0; t < T; ++t)
            th.emplace_back([this, t, cpus]() {
                if (!cpus.empty()) {
                    cpu_set_t set;
                    CPU_ZERO(&set);
                    for (int c : cpus) CPU_SET(c, &set);
                    (void)sched_setaffinity(0, sizeof set, &set);
                }
                size_t seen = 0;
                for (;;) {
                    {
                        std::unique_lock<std::mutex> l(m);
                        cv.wait(l, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                    }
                    for (size_t j; (j = next.fetch_add(1)) < njobs;) fn(j);
                    std::unique_lock<std::mutex> l(m);
                    if (--active == 0) done_cv.notify_all();
                }
            });
    }
    void run(size_t n, std::function<void(size_t)> f) {
        std::unique_lock<std::mutex> l(m);
        fn = std::move(f);
        njobs = n;
        next = 0;
        active = th.size();
        ++gen;
        cv.notify_all();
        done_cv.wait(l, [&] { return active == 0; });
    }
    ~Pool() {
        { std::unique_lock<std::mutex> l(m); stop = true; cv.notify_all(); }
        for (auto& t : th) t.join();
    }
};

int main(int argc, char** argv) {
    const double gib = argc > 1 ? std::atof(argv[1]) : 48.0;
    const size_t units = argc > 2 ? (size_t)std::atoi(argv[2]) : 48;
    constexpr size_t kRow = 1568;
    const size_t rows = (size_t)(gib * (1ull << 30)) / kRow;
    const size_t range = rows / units;                       // rows of the file a unit covers
    CK(hipSetDevice(0));
    const int node = gpu_node();
    std::printf("GPU on NUMA node %d; stand-in file %.1f GiB = %zu rows of %zu bytes in 4 KiB pages; %zu units of %zu rows, 21 %% of them looked up\n",
                node, gib, rows, kRow, units, range);
    uint8_t* file = (uint8_t*)mmap(nullptr, rows * kRow, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (file == MAP_FAILED) { std::perror("mmap"); return 1; }
    (void)madvise(file, rows * kRow, MADV_NOHUGEPAGE);
    {   // first touch by unbound threads: the pages of a file lie where they were read in, not next to the GPU
        Pool init(64, {});
        const size_t piece = 64ull << 20, n = (rows * kRow + piece - 1) / piece;
        init.run(n, [&](size_t j) { std::memset(file + j * piece, (int)(j * 31 + 7), std::min(piece, rows * kRow - j * piece)); });
    }
    // the looked-up rows of every unit, ascending
    std::vector<std::vector<uint32_t>> list(units);
    size_t max_rows = 0, total_rows = 0;
    {
        std::mt19937_64 rng(12345);
        for (size_t u = 0; u < units; ++u) {
            for (size_t r = 0; r < range; ++r)
                if ((rng() % 100) < 21) list[u].push_back((uint32_t)r);
            max_rows = std::max(max_rows, list[u].size());
            total_rows += list[u].size();
        }
    }
    uint8_t* stage[2];
    uint8_t* dev[2];
    hipEvent_t copied[2];
    for (int i = 0; i < 2; ++i) {
        CK(hipHostMalloc((void**)&stage[i], max_rows * kRow, hipHostMallocDefault));
        std::memset(stage[i], 1, max_rows * kRow);
        CK(hipMalloc((void**)&dev[i], max_rows * kRow));
        CK(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
    }
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    // the link alone: the staging buffers as they are, every unit copied
    {
        const double t0 = now();
        for (size_t u = 0; u < units; ++u) CK(hipMemcpyAsync(dev[u & 1], stage[u & 1], list[u].size() * kRow, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        std::printf("  DMA of the staging buffers alone (nothing packed)                   %7.2f GB/s\n", total_rows * kRow / (now() - t0) / 1e9);
    }
    const std::vector<int> near = node >= 0 ? cpus_of_node(node) : std::vector<int>();
    for (int bound = 0; bound < 2; ++bound) {
        if (bound && near.empty()) break;
        for (size_t T : {8, 16, 32, 64, 128}) {
            Pool pool(T, bound ? near : std::vector<int>());
            constexpr size_t kJob = 256;                 // rows per job
            auto pack = [&](size_t u) {
                const uint8_t* base = file + u * range * kRow;
                uint8_t* dst = stage[u & 1];
                const std::vector<uint32_t>& l = list[u];
                pool.run((l.size() + kJob - 1) / kJob, [&, base, dst](size_t j) {
                    const size_t a = j * kJob, b = std::min(l.size(), a + kJob);
                    for (size_t i = a; i < b; ++i) {
                        if (i + 4 < b) __builtin_prefetch(base + (size_t)l[i + 4] * kRow);
                        std::memcpy(dst + i * kRow, base + (size_t)l[i] * kRow, kRow);
                    }
                });
            };
            // packing alone
            double t0 = now();
            for (size_t u = 0; u < units; ++u) pack(u);
            const double pack_only = total_rows * kRow / (now() - t0) / 1e9;
            // the pipeline: pack(u) | DMA(u - 1)
            bool busy[2] = {false, false};
            t0 = now();
            for (size_t u = 0; u < units; ++u) {
                const int b = (int)(u & 1);
                if (busy[b]) CK(hipEventSynchronize(copied[b]));
                pack(u);
                CK(hipMemcpyAsync(dev[b], stage[b], list[u].size() * kRow, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(copied[b], s));
                busy[b] = true;
            }
            CK(hipStreamSynchronize(s));
            const double piped = total_rows * kRow / (now() - t0) / 1e9;
            std::printf("  %3zu threads %-28s  packing alone %7.2f GB/s   packed one unit ahead of the DMA %7.2f GB/s\n", T,
                        bound ? "on the GPU's NUMA node" : "unbound", pack_only, piped);
            std::fflush(stdout);
        }
    }
    return 0;
}
