// scripts/probes/h2d_probe.cpp -- what the PCIe link gives the out-of-core path (BASELINE configs[4]): host -> HBM
// rates of the copy forms stage.cpp can choose from, on one box, back to back.
//   hipcc -O2 --offload-arch=gfx950 scripts/probes/h2d_probe.cpp -o scripts/probes/h2d_probe && scripts/probes/h2d_probe [GiB]
// Forms: linear copy from hipHostMalloc memory; linear / 2-D (1568-byte rows -> 1664-byte pitch, the C3 page) from a
// hipHostRegister'ed file mapping (what a streamed index is); the same split over two streams; slabs of the linear
// form; and each of them with the source pages first-touched on the GPU's NUMA node or on the other one.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
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

static void pin_to(const std::vector<int>& cpus) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    sched_setaffinity(0, sizeof set, &set);
}

// fill `bytes` at p with `threads` threads bound to `cpus` (first touch decides the NUMA node of the pages)
static void touch(uint8_t* p, size_t bytes, const std::vector<int>& cpus, unsigned threads = 16) {
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < threads; ++t)
        pool.emplace_back([=, &cpus]() {
            pin_to(cpus);
            const size_t a = bytes / threads * t, e = t + 1 == threads ? bytes : bytes / threads * (t + 1);
            std::memset(p + a, (int)(t + 1), e - a);
        });
    for (auto& t : pool) t.join();
}

int main(int argc, char** argv) {
    const double gib = argc > 1 ? std::atof(argv[1]) : 4.0;
    const size_t row = 1568, pitch = 1664;
    const size_t rows = (size_t)(gib * (1ull << 30)) / row, bytes = rows * row;
    CK(hipSetDevice(0));
    char bdf[64] = {0};
    CK(hipDeviceGetPCIBusId(bdf, sizeof bdf, 0));
    std::string b = bdf;
    for (auto& c : b) c = (char)std::tolower(c);
    int gpu_node = -1;
    { std::ifstream f("/sys/bus/pci/devices/" + b + "/numa_node"); f >> gpu_node; }
    int nnodes = 0;
    while (access(("/sys/devices/system/node/node" + std::to_string(nnodes)).c_str(), F_OK) == 0) ++nnodes;
    std::printf("GPU %s on NUMA node %d of %d; %zu rows of %zu bytes = %.2f GiB\n", bdf, gpu_node, nnodes, rows, row, bytes / 1073741824.0);
    uint8_t* dev = nullptr;
    CK(hipMalloc((void**)&dev, rows * pitch));
    hipStream_t st[2];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    size_t fn_bytes = bytes;
    auto timed = [&](const char* name, auto fn) {
        fn();                                   // warm-up (maps pages for the DMA engines)
        CK(hipDeviceSynchronize());
        double best = 1e9;
        for (int r = 0; r < 3; ++r) {
            const double t0 = now();
            fn();
            CK(hipDeviceSynchronize());
            best = std::min(best, now() - t0);
        }
        std::printf("  %-78s %7.2f GB/s\n", name, fn_bytes / best / 1e9);
        std::fflush(stdout);
    };
    const int other = nnodes > 1 ? (gpu_node == 0 ? 1 : 0) : gpu_node;
    for (int pass = 0; pass < (nnodes > 1 && gpu_node >= 0 ? 2 : 1); ++pass) {
        const int node = pass == 0 ? (gpu_node >= 0 ? gpu_node : 0) : other;
        const std::vector<int> cpus = cpus_of_node(node);
        std::printf("source pages first-touched on NUMA node %d (%zu cpus)%s\n", node, cpus.size(), node == gpu_node ? " = the GPU's node" : "");
        {   // hipHostMalloc
            uint8_t* h = nullptr;
            CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault));
            touch(h, bytes, cpus);
            timed("hipHostMalloc, linear, one stream", [&]() { CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, st[0])); });
            timed("hipHostMalloc, linear, two streams (halves)", [&]() {
                CK(hipMemcpyAsync(dev, h, bytes / 2, hipMemcpyHostToDevice, st[0]));
                CK(hipMemcpyAsync(dev + bytes / 2, h + bytes / 2, bytes - bytes / 2, hipMemcpyHostToDevice, st[1]));
            });
            timed("hipHostMalloc, 2-D 1568 -> 1664, one stream", [&]() {
                CK(hipMemcpy2DAsync(dev, pitch, h, row, row, rows, hipMemcpyHostToDevice, st[0]));
            });
            CK(hipHostFree(h));
        }
        {   // a file in tmpfs, mapped and registered: what a streamed index is
            const char* dir = getenv("TMPDIR") ? getenv("TMPDIR") : "/tmp";
            const std::string path = std::string(dir) + "/h2d_probe.bin";
            const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_TRUNC, 0600);
            if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { std::perror("file"); return 1; }
            uint8_t* m = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            if (m == MAP_FAILED) { std::perror("mmap"); return 1; }
            touch(m, bytes, cpus);
            munmap(m, bytes);
            m = (uint8_t*)mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { std::perror("mmap"); return 1; }
            const double t0 = now();
            hipError_t re = hipHostRegister(m, bytes, hipHostRegisterReadOnly);
            if (re != hipSuccess) { (void)hipGetLastError(); re = hipHostRegister(m, bytes, hipHostRegisterDefault); }
            std::printf("  hipHostRegister of the mapping: %s, %.2f s\n", hipGetErrorString(re), now() - t0);
            if (re == hipSuccess) {
                timed("registered mapping, linear, one stream", [&]() { CK(hipMemcpyAsync(dev, m, bytes, hipMemcpyHostToDevice, st[0])); });
                timed("registered mapping, 2-D 1568 -> 1664, one stream (stream_chunk_in today)", [&]() {
                    CK(hipMemcpy2DAsync(dev, pitch, m, row, row, rows, hipMemcpyHostToDevice, st[0]));
                });
                timed("registered mapping, 2-D, two streams (halves)", [&]() {
                    CK(hipMemcpy2DAsync(dev, pitch, m, row, row, rows / 2, hipMemcpyHostToDevice, st[0]));
                    CK(hipMemcpy2DAsync(dev + rows / 2 * pitch, pitch, m + rows / 2 * row, row, row, rows - rows / 2, hipMemcpyHostToDevice, st[1]));
                });
                timed("registered mapping, linear, two streams (halves)", [&]() {
                    CK(hipMemcpyAsync(dev, m, bytes / 2, hipMemcpyHostToDevice, st[0]));
                    CK(hipMemcpyAsync(dev + bytes / 2, m + bytes / 2, bytes - bytes / 2, hipMemcpyHostToDevice, st[1]));
                });
                timed("registered mapping, linear, 64 MiB slabs alternating over two streams", [&]() {
                    const size_t slab = 64u << 20;
                    int k = 0;
                    for (size_t o = 0; o < bytes; o += slab, ++k)
                        CK(hipMemcpyAsync(dev + o, m + o, std::min(slab, bytes - o), hipMemcpyHostToDevice, st[k & 1]));
                });
                // column slices of a sub-index larger than a stream buffer (plan.cpp: chunk_part): all rows, `w` bytes of each
                if (pass == 0) {
                    for (size_t w : {32, 128, 288, 416, 512, 544, 576, 640, 672, 784, 896, 1024, 1536}) {
                        fn_bytes = rows * w;
                        char name[96];
                        std::snprintf(name, sizeof name, "registered mapping, 2-D column slice: %4zu of 1568 bytes per row, pitch %zu", w, (w + 127) / 128 * 128);
                        timed(name, [&]() { CK(hipMemcpy2DAsync(dev, (w + 127) / 128 * 128, m + 128, row, w, rows, hipMemcpyHostToDevice, st[0])); });
                    }
                    // the same slices with the rows split over two streams (two SDMA engines), and over four
                    hipStream_t st4[4] = {st[0], st[1], nullptr, nullptr};
                    CK(hipStreamCreateWithFlags(&st4[2], hipStreamNonBlocking));
                    CK(hipStreamCreateWithFlags(&st4[3], hipStreamNonBlocking));
                    for (int ns : {2, 4})
                        for (size_t w : {288, 544, 640, 896}) {
                            fn_bytes = rows * w;
                            const size_t dp = (w + 127) / 128 * 128;
                            char name[96];
                            std::snprintf(name, sizeof name, "registered mapping, 2-D column slice: %4zu bytes per row, rows over %d streams", w, ns);
                            timed(name, [&]() {
                                for (int k = 0; k < ns; ++k) {
                                    const size_t r0 = rows * k / ns, r1 = rows * (k + 1) / ns;
                                    CK(hipMemcpy2DAsync(dev + r0 * dp, dp, m + 128 + r0 * row, row, w, r1 - r0, hipMemcpyHostToDevice, st4[k]));
                                }
                            });
                        }
                    fn_bytes = bytes;
                }
                CK(hipHostUnregister(m));
            }
            munmap(m, bytes);
            close(fd);
            unlink(path.c_str());
        }
    }
    return 0;
}
