// cobs_amd/csrc/build.cpp -- GPU index construction: the step in front
// of the query path.  Restates on the device what the reference does per document with
// process_term / set_bit (cobs/construction/classic_index.cpp:40-130) and writes files in
// the reference's formats (cobs/file/classic_index_header.cpp:26-37,
// cobs/file/compact_index_header.cpp:20-43), so that `cobs query`, the reference's tests
// and this engine can read them.  Document parsing (FASTA, ...) stays with the caller:
// a document arrives as its sequences joined by '\n'.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cobs_gpu.h"
#include "device_types.hpp"
#include "kernels.hpp"

using namespace cobs_amd;

__attribute__((visibility("hidden"))) cobs_gpu_status cobs_gpu_set_error(cobs_gpu_status st, const char* msg);   // engine.cpp

namespace {

#define BUILD_TRY(expr)                                                             \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            const bool nodev = _e == hipErrorNoDevice || _e == hipErrorInvalidDevice; \
            std::string m = std::string(#expr) + ": " + hipGetErrorString(_e);      \
            (void)hipGetLastError();                                                \
            return cobs_gpu_set_error(nodev ? COBS_GPU_ERR_NO_DEVICE : COBS_GPU_ERR_HIP, m.c_str()); \
        }                                                                           \
    } while (0)

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
};

// calc_signature_size, cobs/util/calc_signature_size.cpp:17-33 (all arithmetic in double)
uint64_t signature_size_for(uint64_t num_elements, double num_hashes, double fpr) {
    const double ratio = -num_hashes / std::log(1.0 - std::pow(fpr, 1.0 / num_hashes));
    return (uint64_t)std::ceil((double)num_elements * ratio);
}

// number of k-grams of a document given as '\n'-separated sequences
uint64_t count_terms(const char* text, size_t len, uint32_t k) {
    uint64_t total = 0, run = 0;
    for (size_t i = 0; i <= len; ++i) {
        if (i == len || text[i] == '\n') {
            if (run >= k) total += run - k + 1;
            run = 0;
        } else {
            ++run;
        }
    }
    return total;
}

struct Params {
    uint32_t term_size = 31, canonicalize = 1, num_hashes = 1;
    double fpr = 0.3;
    uint64_t signature_size = 0, page_size = 0;
    int device = -1;
};

cobs_gpu_status read_params(const cobs_gpu_build_params* p, Params& out) {
    if (p) {
        if (p->struct_size < sizeof(cobs_gpu_build_params))
            return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "cobs_gpu_build_params.struct_size is too small");
        out.term_size = p->term_size;
        out.canonicalize = p->canonicalize;
        out.num_hashes = p->num_hashes;
        out.fpr = p->false_positive_rate;
        out.signature_size = p->signature_size;
        out.page_size = p->page_size;
        out.device = p->device;
    }
    if (out.term_size == 0 || out.num_hashes == 0 || out.num_hashes > 64 || out.canonicalize > 1)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad term_size / num_hashes / canonicalize");
    if (out.signature_size == 0 && !(out.fpr > 0.0 && out.fpr < 1.0))
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "false_positive_rate must be in (0, 1)");
    return COBS_GPU_OK;
}

cobs_gpu_status pick_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return cobs_gpu_set_error(COBS_GPU_ERR_NO_DEVICE, "no HIP device visible; libcobs_gpu has no CPU fallback");
    }
    if (device >= 0) {
        if (device >= n) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "device ordinal out of range");
        BUILD_TRY(hipSetDevice(device));
    }
    return COBS_GPU_OK;
}

// Build the bit matrix of documents [d0, d1) into `rows` x `row_size` bytes on the host.
cobs_gpu_status build_matrix(const char* const* texts, const size_t* lens, size_t d0, size_t d1,
                             const Params& pr, uint64_t sig, uint64_t row_size, std::vector<uint8_t>& out) {
    // text: documents back to back, every document followed by a separator
    std::vector<uint64_t> off(d1 - d0 + 1);
    uint64_t total = 0;
    for (size_t d = d0; d < d1; ++d) {
        off[d - d0] = total;
        total += lens[d] + 1;
    }
    off[d1 - d0] = total;
    std::vector<uint8_t> text((size_t)total);
    for (size_t d = d0; d < d1; ++d) {
        std::memcpy(text.data() + off[d - d0], texts[d], lens[d]);
        text[(size_t)(off[d - d0] + lens[d])] = '\n';
    }
    const uint64_t row_bytes = (row_size + 3) / 4 * 4;
    if (sig > (1ull << 46)) return cobs_gpu_set_error(COBS_GPU_ERR_UNSUPPORTED, "signature_size must be at most 2^46");
    DevMem d_text, d_off, d_mat;
    BUILD_TRY(hipMalloc(&d_text.p, std::max<size_t>((size_t)total, 1)));
    BUILD_TRY(hipMalloc(&d_off.p, off.size() * 8));
    BUILD_TRY(hipMalloc(&d_mat.p, (size_t)(sig * row_bytes)));
    if (total) BUILD_TRY(hipMemcpy(d_text.p, text.data(), (size_t)total, hipMemcpyHostToDevice));
    BUILD_TRY(hipMemcpy(d_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    BUILD_TRY(hipMemset(d_mat.p, 0, (size_t)(sig * row_bytes)));
    BuildArgs a;
    a.text = (const uint8_t*)d_text.p;
    a.doc_off = (const uint64_t*)d_off.p;
    a.matrix = (uint32_t*)d_mat.p;
    a.signature_size = sig;
    a.magic = ~0ull / sig;
    a.row_bytes = row_bytes;
    a.ndocs = (uint32_t)(d1 - d0);
    a.doc_bit0 = 0;
    a.term_size = pr.term_size;
    a.canonicalize = pr.canonicalize;
    a.num_hashes = pr.num_hashes;
    BUILD_TRY(launch_build(a, total, nullptr));
    out.assign((size_t)(sig * row_size), 0);
    BUILD_TRY(hipMemcpy2D(out.data(), (size_t)row_size, d_mat.p, (size_t)row_bytes, (size_t)row_size, (size_t)sig,
                          hipMemcpyDeviceToHost));
    return COBS_GPU_OK;
}

bool write_all(FILE* f, const void* p, size_t n) { return n == 0 || std::fwrite(p, 1, n, f) == n; }

template <typename T>
void put(std::string& s, T v) { s.append(reinterpret_cast<const char*>(&v), sizeof v); }

}  // namespace

extern "C" {

cobs_gpu_status cobs_gpu_build_classic(const char* const* names, const char* const* texts, const size_t* lens,
                                       size_t ndocs, const cobs_gpu_build_params* params, const char* out_path) {
    if (!names || !texts || !lens || !out_path || ndocs == 0)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    Params pr;
    cobs_gpu_status st = read_params(params, pr);
    if (st != COBS_GPU_OK) return st;
    st = pick_device(pr.device);
    if (st != COBS_GPU_OK) return st;
    uint64_t sig = pr.signature_size;
    if (sig == 0) {     // classic_construct, classic_index.cpp:571-575: sized by the largest document
        uint64_t max_terms = 0;
        for (size_t d = 0; d < ndocs; ++d) max_terms = std::max(max_terms, count_terms(texts[d], lens[d], pr.term_size));
        sig = signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
    }
    if (sig == 0) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
    const uint64_t row_size = (ndocs + 7) / 8;
    std::vector<uint8_t> matrix;
    st = build_matrix(texts, lens, 0, ndocs, pr, sig, row_size, matrix);
    if (st != COBS_GPU_OK) return st;
    std::string h = "COBS:CLASSIC_INDEX";
    put<uint32_t>(h, 1);
    put<uint32_t>(h, pr.term_size);
    put<uint8_t>(h, (uint8_t)pr.canonicalize);
    put<uint32_t>(h, (uint32_t)ndocs);
    put<uint64_t>(h, sig);
    put<uint64_t>(h, (uint64_t)pr.num_hashes);
    for (size_t d = 0; d < ndocs; ++d) { h += names[d]; h += '\n'; }
    h += "CLASSIC_INDEX";
    FILE* f = std::fopen(out_path, "wb");
    if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
    const bool ok = write_all(f, h.data(), h.size()) && write_all(f, matrix.data(), matrix.size());
    if (std::fclose(f) != 0 || !ok) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    return COBS_GPU_OK;
}

cobs_gpu_status cobs_gpu_build_compact(const char* const* names, const char* const* texts, const size_t* lens,
                                       size_t ndocs, const cobs_gpu_build_params* params, const char* out_path) {
    if (!names || !texts || !lens || !out_path || ndocs == 0)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    Params pr;
    cobs_gpu_status st = read_params(params, pr);
    if (st != COBS_GPU_OK) return st;
    st = pick_device(pr.device);
    if (st != COBS_GPU_OK) return st;
    uint64_t ps = pr.page_size;
    if (ps == 0) {      // compact_construct, compact_index.cpp:184-189
        const uint64_t v = (uint64_t)std::sqrt((double)(ndocs / 8));
        uint64_t p2 = 1;
        while (p2 < v) p2 <<= 1;
        ps = std::min<uint64_t>(std::max<uint64_t>(v == 0 ? 0 : p2, 8), 4096);
    }
    const size_t group = (size_t)(8 * ps);
    std::vector<std::pair<uint64_t, uint64_t>> plist;          // (signature_size, num_hashes)
    std::vector<std::vector<uint8_t>> mats;
    std::vector<size_t> kept;                                   // documents that made it into the file
    for (size_t g0 = 0; g0 < ndocs; g0 += group) {
        const size_t g1 = std::min(ndocs, g0 + group);
        uint64_t max_terms = 0;
        for (size_t d = g0; d < g1; ++d) max_terms = std::max(max_terms, count_terms(texts[d], lens[d], pr.term_size));
        if (max_terms == 0) continue;                           // compact_index.cpp:285-286: empty group is dropped
        const uint64_t sig = pr.signature_size ? pr.signature_size
                                               : signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
        std::vector<uint8_t> m;
        st = build_matrix(texts, lens, g0, g1, pr, sig, ps, m);  // rows padded to page_size (:116-156)
        if (st != COBS_GPU_OK) return st;
        plist.emplace_back(sig, pr.num_hashes);
        mats.push_back(std::move(m));
        for (size_t d = g0; d < g1; ++d) kept.push_back(d);
    }
    if (plist.empty()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
    std::string h = "COBS:COMPACT_INDEX";
    put<uint32_t>(h, 1);
    put<uint32_t>(h, pr.term_size);
    put<uint8_t>(h, (uint8_t)pr.canonicalize);
    put<uint32_t>(h, (uint32_t)plist.size());
    put<uint32_t>(h, (uint32_t)kept.size());
    put<uint64_t>(h, ps);
    for (auto& pe : plist) { put<uint64_t>(h, pe.first); put<uint64_t>(h, pe.second); }
    for (size_t d : kept) { h += names[d]; h += '\n'; }
    const uint64_t pad = (ps - ((h.size() + 13) % ps)) % ps;     // data starts page-aligned
    h.append((size_t)pad, '\0');
    h += "COMPACT_INDEX";
    FILE* f = std::fopen(out_path, "wb");
    if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
    bool ok = write_all(f, h.data(), h.size());
    for (auto& m : mats) ok = ok && write_all(f, m.data(), m.size());
    if (std::fclose(f) != 0 || !ok) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    return COBS_GPU_OK;
}

// The procedural index of cobs_gpu_open_synthetic as a FILE in the reference's format: the
// stand-in for `cobs classic-construct-random` (src/cobs.cpp:243-291,
// construction/classic_index.cpp:661-725) at sizes where hashing 10^10 random k-mers is not the
// point -- same header (classic_index_header.cpp:26-37 / compact_index_header.cpp:20-43), document
// names file_%06u (classic_index.cpp:668-670), bits of density ~0.3.  `cobs query`, the reference's
// tools and this engine (resident or streamed) read it.  Rows are generated on the device in
// chunks and streamed to the file: neither HBM nor host memory holds the matrix.
cobs_gpu_status cobs_gpu_write_synthetic(const cobs_gpu_synth* d, const char* out_path, int device) {
    if (!d || !d->signature_sizes || !out_path || d->num_pages == 0 || d->kind > 1)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad synthetic index description");
    if ((d->kind == 1 && d->page_size == 0) || (d->kind == 0 && d->num_pages != 1) || d->num_docs == 0 ||
        d->num_docs > 0xFFFFFFF0ull || (d->kind == 1 && d->num_docs > (uint64_t)d->num_pages * 8 * d->page_size))
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad synthetic index geometry");
    cobs_gpu_status st = pick_device(device);
    if (st != COBS_GPU_OK) return st;
    const uint64_t prb = d->kind ? d->page_size : (d->num_docs + 7) / 8;
    std::string h;
    char nm[32];
    if (d->kind == 0) {
        h = "COBS:CLASSIC_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, d->term_size);
        put<uint8_t>(h, (uint8_t)d->canonicalize);
        put<uint32_t>(h, (uint32_t)d->num_docs);
        put<uint64_t>(h, d->signature_sizes[0]);
        put<uint64_t>(h, d->num_hashes);
        for (uint64_t i = 0; i < d->num_docs; ++i) { std::snprintf(nm, sizeof nm, "file_%06u\n", (unsigned)i); h += nm; }
        h += "CLASSIC_INDEX";
    } else {
        h = "COBS:COMPACT_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, d->term_size);
        put<uint8_t>(h, (uint8_t)d->canonicalize);
        put<uint32_t>(h, d->num_pages);
        put<uint32_t>(h, (uint32_t)d->num_docs);
        put<uint64_t>(h, d->page_size);
        for (uint32_t p = 0; p < d->num_pages; ++p) { put<uint64_t>(h, d->signature_sizes[p]); put<uint64_t>(h, d->num_hashes); }
        for (uint64_t i = 0; i < d->num_docs; ++i) { std::snprintf(nm, sizeof nm, "file_%06u\n", (unsigned)i); h += nm; }
        const uint64_t pad = (d->page_size - ((h.size() + 13) % d->page_size)) % d->page_size;
        h.append((size_t)pad, '\0');
        h += "COMPACT_INDEX";
    }
    FILE* f = std::fopen(out_path, "wb");
    if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
    struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
    if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    const uint32_t pitch = (uint32_t)((prb + 7) / 8 * 8);
    const uint64_t rows_per = std::max<uint64_t>(1, (256ull << 20) / pitch);
    DevMem d_rows;
    BUILD_TRY(hipMalloc(&d_rows.p, (size_t)(rows_per * pitch)));
    struct Pinned { void* p = nullptr; ~Pinned() { if (p) (void)hipHostFree(p); } } host[2];
    hipStream_t stream = nullptr;
    BUILD_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    for (auto& hb : host) BUILD_TRY(hipHostMalloc(&hb.p, (size_t)(rows_per * prb), hipHostMallocDefault));
    int cur = 0;
    uint64_t pending = 0;       // bytes of host[cur ^ 1] still to be written
    for (uint32_t p = 0; p < d->num_pages; ++p) {
        const uint64_t sig = d->signature_sizes[p];
        const uint64_t first_doc = d->kind ? (uint64_t)p * 8 * d->page_size : 0;
        const uint64_t live = d->num_docs > first_doc ? d->num_docs - first_doc : 0;
        for (uint64_t r = 0; r < sig; r += rows_per) {
            const uint64_t n = std::min(rows_per, sig - r);
            SynthRowsArgs a;
            a.dst = (uint8_t*)d_rows.p;
            a.seed = d->seed;
            a.row0 = r;
            a.nrows = n;
            a.row_bytes = prb;
            a.live_docs = live;
            a.page = p;
            a.pitch = pitch;
            BUILD_TRY(launch_synth_rows(a, stream));
            BUILD_TRY(hipMemcpy2DAsync(host[cur].p, (size_t)prb, d_rows.p, pitch, (size_t)prb, (size_t)n,
                                       hipMemcpyDeviceToHost, stream));
            // while the device produces this chunk the host writes the previous one
            if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
                return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
            BUILD_TRY(hipStreamSynchronize(stream));
            pending = n * prb;
            cur ^= 1;
        }
    }
    if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
        return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    closer.f = nullptr;
    if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    return COBS_GPU_OK;
}

}  // extern "C"
