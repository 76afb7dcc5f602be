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
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "engine.hpp"

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
    const uint64_t* doc_terms = nullptr;
    uint64_t text_batch = 0;
};

cobs_gpu_status read_params(const cobs_gpu_build_params* p, Params& out) {
    if (p) {
        if (p->struct_size < offsetof(cobs_gpu_build_params, doc_terms))
            return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "cobs_gpu_build_params.struct_size is too small");
        if (p->struct_size >= sizeof(cobs_gpu_build_params)) out.doc_terms = p->doc_terms;
        out.term_size = p->term_size;
        out.canonicalize = p->canonicalize;
        out.num_hashes = p->num_hashes;
        out.fpr = p->false_positive_rate;
        out.signature_size = p->signature_size;
        out.page_size = p->page_size;
        out.device = p->device;
        out.text_batch = p->text_batch_bytes;
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

// Documents reach the device in batches of at most this many text bytes (the reference batches
// documents by a memory budget too: classic_index.cpp:565-659 builds one small index per batch
// and interleaves them afterwards; here every batch sets its bits straight at the documents'
// final columns of the one matrix in HBM, so there is nothing to combine).
constexpr uint64_t kTextBatchBytes = 256ull << 20;

// Set the bits of documents [d0, d1) -- columns doc_bit0 + (d - d0) -- in a zeroed device matrix of
// `sig` rows, `row_bytes` (multiple of 4) apart.
cobs_gpu_status build_into(uint32_t* d_matrix, uint64_t sig, uint64_t row_bytes, const char* const* texts,
                           const size_t* lens, size_t d0, size_t d1, uint32_t doc_bit0, const Params& pr) {
    const uint64_t text_batch = pr.text_batch ? pr.text_batch : kTextBatchBytes;
    if (sig == 0 || sig > (1ull << 46)) return cobs_gpu_set_error(COBS_GPU_ERR_UNSUPPORTED, "signature_size must be in 1..2^46");
    DevMem d_text, d_off;
    size_t text_cap = 0, off_cap = 0;
    std::vector<uint8_t> text;
    std::vector<uint64_t> off;
    for (size_t b0 = d0; b0 < d1;) {
        // documents [b0, b1): as many as fit the batch (at least one)
        size_t b1 = b0;
        uint64_t total = 0;
        while (b1 < d1 && (b1 == b0 || total + lens[b1] + 1 <= text_batch)) total += lens[b1++] + 1;
        off.assign(b1 - b0 + 1, 0);
        text.resize((size_t)total);
        uint64_t pos = 0;
        for (size_t d = b0; d < b1; ++d) {
            off[d - b0] = pos;
            std::memcpy(text.data() + pos, texts[d], lens[d]);
            text[(size_t)(pos + lens[d])] = '\n';          // every document is followed by a separator
            pos += lens[d] + 1;
        }
        off[b1 - b0] = pos;
        if (total > text_cap) {
            if (d_text.p) { (void)hipFree(d_text.p); d_text.p = nullptr; }
            BUILD_TRY(hipMalloc(&d_text.p, (size_t)total));
            text_cap = (size_t)total;
        }
        if (off.size() > off_cap) {
            if (d_off.p) { (void)hipFree(d_off.p); d_off.p = nullptr; }
            BUILD_TRY(hipMalloc(&d_off.p, off.size() * 8));
            off_cap = off.size();
        }
        if (total) BUILD_TRY(hipMemcpy(d_text.p, text.data(), (size_t)total, hipMemcpyHostToDevice));
        BUILD_TRY(hipMemcpy(d_off.p, off.data(), off.size() * 8, hipMemcpyHostToDevice));
        BuildArgs a;
        a.text = (const uint8_t*)d_text.p;
        a.doc_off = (const uint64_t*)d_off.p;
        a.matrix = d_matrix;
        a.signature_size = sig;
        a.magic = ~0ull / sig;
        a.row_bytes = row_bytes;
        a.ndocs = (uint32_t)(b1 - b0);
        a.doc_bit0 = doc_bit0 + (uint32_t)(b0 - d0);
        a.term_size = pr.term_size;
        a.canonicalize = pr.canonicalize;
        a.num_hashes = pr.num_hashes;
        BUILD_TRY(launch_build(a, total, nullptr));
        BUILD_TRY(hipStreamSynchronize(nullptr));
        b0 = b1;
    }
    return COBS_GPU_OK;
}

bool write_all(FILE* f, const void* p, size_t n);

// rows [0, rows) of a device matrix (pitch bytes apart) -> file, row_size bytes each, through two
// pinned buffers: the host writes chunk i while the device sends chunk i + 1
cobs_gpu_status stream_rows_to_file(FILE* f, const uint8_t* d_matrix, uint64_t pitch, uint64_t row_size, uint64_t rows) {
    if (rows == 0 || row_size == 0) return COBS_GPU_OK;
    const uint64_t rows_per = std::max<uint64_t>(1, (128ull << 20) / row_size);
    struct Pinned { void* p = nullptr; ~Pinned() { if (p) (void)hipHostFree(p); } } host[2];
    for (auto& hb : host) BUILD_TRY(hipHostMalloc(&hb.p, (size_t)(std::min(rows_per, rows) * row_size), hipHostMallocDefault));
    hipStream_t stream = nullptr;
    BUILD_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamDestroy(s); } } sg{stream};
    int cur = 0;
    uint64_t pending = 0;
    for (uint64_t r = 0; r < rows; r += rows_per) {
        const uint64_t n = std::min(rows_per, rows - r);
        BUILD_TRY(hipMemcpy2DAsync(host[cur].p, (size_t)row_size, d_matrix + r * pitch, (size_t)pitch, (size_t)row_size,
                                   (size_t)n, hipMemcpyDeviceToHost, stream));
        if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
            return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        BUILD_TRY(hipStreamSynchronize(stream));
        pending = n * row_size;
        cur ^= 1;
    }
    if (pending && !write_all(f, host[cur ^ 1].p, (size_t)pending))
        return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
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
        for (size_t d = 0; d < ndocs; ++d)
            max_terms = std::max(max_terms, pr.doc_terms ? pr.doc_terms[d] : count_terms(texts[d], lens[d], pr.term_size));
        sig = signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
    }
    if (sig == 0) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
    const uint64_t row_size = (ndocs + 7) / 8;
    const uint64_t row_bytes = (row_size + 3) / 4 * 4;
    DevMem d_mat;                                   // the whole matrix lives in HBM (288 GB), never in host RAM
    BUILD_TRY(hipMalloc(&d_mat.p, (size_t)(sig * row_bytes)));
    BUILD_TRY(hipMemset(d_mat.p, 0, (size_t)(sig * row_bytes)));
    st = build_into((uint32_t*)d_mat.p, sig, row_bytes, texts, lens, 0, ndocs, 0, pr);
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
    struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
    if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    st = stream_rows_to_file(f, (const uint8_t*)d_mat.p, row_bytes, row_size, sig);
    if (st != COBS_GPU_OK) return st;
    closer.f = nullptr;
    if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
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
    struct Group { size_t g0, g1; uint64_t sig; };
    std::vector<Group> groups;
    std::vector<size_t> kept;                                   // documents that made it into the file
    for (size_t g0 = 0; g0 < ndocs; g0 += group) {
        const size_t g1 = std::min(ndocs, g0 + group);
        uint64_t max_terms = 0;
        for (size_t d = g0; d < g1; ++d)
            max_terms = std::max(max_terms, pr.doc_terms ? pr.doc_terms[d] : count_terms(texts[d], lens[d], pr.term_size));
        if (max_terms == 0) continue;                           // compact_index.cpp:285-286: empty group is dropped
        const uint64_t sig = pr.signature_size ? pr.signature_size
                                               : signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
        groups.push_back(Group{g0, g1, sig});
        for (size_t d = g0; d < g1; ++d) kept.push_back(d);
    }
    if (groups.empty()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
    std::string h = "COBS:COMPACT_INDEX";
    put<uint32_t>(h, 1);
    put<uint32_t>(h, pr.term_size);
    put<uint8_t>(h, (uint8_t)pr.canonicalize);
    put<uint32_t>(h, (uint32_t)groups.size());
    put<uint32_t>(h, (uint32_t)kept.size());
    put<uint64_t>(h, ps);
    for (auto& g : groups) { put<uint64_t>(h, g.sig); put<uint64_t>(h, (uint64_t)pr.num_hashes); }
    for (size_t d : kept) { h += names[d]; h += '\n'; }
    const uint64_t pad = (ps - ((h.size() + 13) % ps)) % ps;     // data starts page-aligned
    h.append((size_t)pad, '\0');
    h += "COMPACT_INDEX";
    FILE* f = std::fopen(out_path, "wb");
    if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
    struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
    if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    // one sub-index after the other: built in HBM (rows padded to page_size, :116-156), streamed out
    const uint64_t row_bytes = (ps + 3) / 4 * 4;
    for (auto& g : groups) {
        DevMem d_mat;
        BUILD_TRY(hipMalloc(&d_mat.p, (size_t)(g.sig * row_bytes)));
        BUILD_TRY(hipMemset(d_mat.p, 0, (size_t)(g.sig * row_bytes)));
        st = build_into((uint32_t*)d_mat.p, g.sig, row_bytes, texts, lens, g.g0, g.g1, 0, pr);
        if (st != COBS_GPU_OK) return st;
        st = stream_rows_to_file(f, (const uint8_t*)d_mat.p, row_bytes, ps, g.sig);
        if (st != COBS_GPU_OK) return st;
    }
    closer.f = nullptr;
    if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
    return COBS_GPU_OK;
}

// classic_construct / compact_construct straight into a resident query index: the matrix is built
// at the engine's row pitch inside the handle's HBM blob, nothing touches a file or host memory.
cobs_gpu_status cobs_gpu_build_index(uint32_t kind, const char* const* names, const char* const* texts,
                                     const size_t* lens, size_t ndocs, const cobs_gpu_build_params* params,
                                     const cobs_gpu_options* opts, cobs_gpu_index** out) {
    if (!out) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!names || !texts || !lens || ndocs == 0 || kind > 1)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no documents");
    return guarded([&]() -> cobs_gpu_status {
        Params pr;
        cobs_gpu_status st = read_params(params, pr);
        if (st != COBS_GPU_OK) return st;
        IndexMeta meta;
        meta.kind = kind ? IndexKind::Compact : IndexKind::Classic;
        meta.term_size = pr.term_size;
        meta.canonicalize = (uint8_t)pr.canonicalize;
        meta.num_hashes = pr.num_hashes;
        struct Group { size_t g0, g1; };
        std::vector<Group> groups;
        auto terms_of = [&](size_t d) { return pr.doc_terms ? pr.doc_terms[d] : count_terms(texts[d], lens[d], pr.term_size); };
        if (kind == 0) {
            uint64_t sig = pr.signature_size, max_terms = 0;
            for (size_t d = 0; d < ndocs; ++d) max_terms = std::max(max_terms, terms_of(d));
            if (sig == 0) sig = signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr);
            if (sig == 0) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
            meta.signature_sizes.assign(1, sig);
            groups.push_back(Group{0, ndocs});
            for (size_t d = 0; d < ndocs; ++d) meta.doc_names.emplace_back(names[d]);
        } else {
            uint64_t ps = pr.page_size;
            if (ps == 0) {      // compact_construct, compact_index.cpp:184-189
                const uint64_t v = (uint64_t)std::sqrt((double)(ndocs / 8));
                uint64_t p2 = 1;
                while (p2 < v) p2 <<= 1;
                ps = std::min<uint64_t>(std::max<uint64_t>(v == 0 ? 0 : p2, 8), 4096);
            }
            meta.header_page_size = ps;
            for (size_t g0 = 0; g0 < ndocs; g0 += (size_t)(8 * ps)) {
                const size_t g1 = std::min(ndocs, g0 + (size_t)(8 * ps));
                uint64_t max_terms = 0;
                for (size_t d = g0; d < g1; ++d) max_terms = std::max(max_terms, terms_of(d));
                if (max_terms == 0) continue;
                meta.signature_sizes.push_back(pr.signature_size ? pr.signature_size
                                                                 : signature_size_for(max_terms, (double)pr.num_hashes, pr.fpr));
                groups.push_back(Group{g0, g1});
                for (size_t d = g0; d < g1; ++d) meta.doc_names.emplace_back(names[d]);
            }
            if (groups.empty()) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "documents hold no terms");
        }
        cobs_gpu_options o{};
        if (opts) std::memcpy(&o, opts, std::min<size_t>(opts->struct_size, sizeof o));
        o.struct_size = sizeof o;
        if (pr.device >= 0) o.device = pr.device;
        else if (!opts) o.device = -1;
        cobs_gpu_index* ix = nullptr;
        st = open_zeroed(std::move(meta), &o, &ix);
        if (st != COBS_GPU_OK) return st;
        std::unique_ptr<cobs_gpu_index, void (*)(cobs_gpu_index*)> guard(ix, cobs_gpu_close);
        Part& pt = ix->parts[0];
        for (Chunk& c : pt.chunks)
            for (size_t i = 0; i < c.vp.size(); ++i) {
                const Group& g = groups[c.vp[i].fp];
                st = build_into(reinterpret_cast<uint32_t*>(c.d_data + c.pages[i].base), c.pages[i].sig, c.pitch, texts, lens,
                                g.g0, g.g1, 0, pr);
                if (st != COBS_GPU_OK) return st;
            }
        *out = guard.release();
        return COBS_GPU_OK;
    });
}

// classic_combine (classic_index.cpp:195-327) on the GPU: the rows of several classic indexes with
// the same parameters are concatenated at bit granularity into one index (document names in input
// order).  Row batches of at most mem_bytes (0 = 1 GiB) travel file -> HBM -> file.
// (The reference's general-bit path ORs into an output block it never clears between batches,
// :302-312; this writes what its single-batch case writes.)
cobs_gpu_status cobs_gpu_combine_classic(const char* const* in_paths, size_t n, const char* out_path,
                                         uint64_t mem_bytes, int device) {
    if (!in_paths || n == 0 || !out_path) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "NULL argument or no inputs");
    return guarded([&]() -> cobs_gpu_status {
        std::vector<std::unique_ptr<MappedFile>> files;
        std::vector<IndexMeta> metas(n);
        std::string err;
        for (size_t i = 0; i < n; ++i) {
            files.emplace_back(new MappedFile);
            if (!in_paths[i] || !files[i]->open(in_paths[i], err)) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, err.c_str());
            if (!parse_index_header(files[i]->data(), files[i]->size(), metas[i], err) || metas[i].kind != IndexKind::Classic)
                return cobs_gpu_set_error(COBS_GPU_ERR_FORMAT, (std::string(in_paths[i]) + ": not a classic index").c_str());
            const IndexMeta &a = metas[0], &b = metas[i];
            if (a.term_size != b.term_size || a.canonicalize != b.canonicalize || a.num_hashes != b.num_hashes ||
                a.signature_sizes[0] != b.signature_sizes[0])
                return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "indexes to combine differ in term size, canonicalize, hashes or signature size");
        }
        cobs_gpu_status st = pick_device(device);
        if (st != COBS_GPU_OK) return st;
        const uint64_t sig = metas[0].signature_sizes[0];
        std::vector<uint64_t> bit_off(n + 1, 0), src_rb(n);
        uint64_t in_row_bytes = 0;
        for (size_t i = 0; i < n; ++i) {
            bit_off[i + 1] = bit_off[i] + metas[i].doc_names.size();
            src_rb[i] = metas[i].page_row_bytes();
            in_row_bytes += src_rb[i];
        }
        const uint64_t total_docs = bit_off[n];
        if (total_docs == 0 || total_docs > 0xFFFFFFF0ull) return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad document count");
        const uint64_t out_rb = (total_docs + 7) / 8;
        std::string h = "COBS:CLASSIC_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, metas[0].term_size);
        put<uint8_t>(h, metas[0].canonicalize);
        put<uint32_t>(h, (uint32_t)total_docs);
        put<uint64_t>(h, sig);
        put<uint64_t>(h, metas[0].num_hashes);
        for (size_t i = 0; i < n; ++i)
            for (const std::string& nm : metas[i].doc_names) { h += nm; h += '\n'; }
        h += "CLASSIC_INDEX";
        FILE* f = std::fopen(out_path, "wb");
        if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
        struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
        if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        // batch_size rows at a time (the reference: mem_bytes / new_row_bytes / 2, :236-238)
        const uint64_t budget = mem_bytes ? mem_bytes : 1ull << 30;
        const uint64_t batch = std::max<uint64_t>(1, std::min(sig, budget / (in_row_bytes + out_rb)));
        DevMem d_in, d_out, d_ptr, d_rb, d_off;
        BUILD_TRY(hipMalloc(&d_in.p, (size_t)(batch * in_row_bytes)));
        BUILD_TRY(hipMalloc(&d_out.p, (size_t)(batch * out_rb)));
        BUILD_TRY(hipMalloc(&d_ptr.p, n * sizeof(void*)));
        BUILD_TRY(hipMalloc(&d_rb.p, n * 8));
        BUILD_TRY(hipMalloc(&d_off.p, (n + 1) * 8));
        std::vector<const uint8_t*> ptrs(n);
        uint64_t pos = 0;
        for (size_t i = 0; i < n; ++i) { ptrs[i] = (const uint8_t*)d_in.p + pos; pos += batch * src_rb[i]; }
        BUILD_TRY(hipMemcpy(d_ptr.p, ptrs.data(), n * sizeof(void*), hipMemcpyHostToDevice));
        BUILD_TRY(hipMemcpy(d_rb.p, src_rb.data(), n * 8, hipMemcpyHostToDevice));
        BUILD_TRY(hipMemcpy(d_off.p, bit_off.data(), (n + 1) * 8, hipMemcpyHostToDevice));
        std::vector<uint8_t> host_out((size_t)(batch * out_rb));
        for (uint64_t r0 = 0; r0 < sig; r0 += batch) {
            const uint64_t rows = std::min(batch, sig - r0);
            for (size_t i = 0; i < n; ++i)
                if (src_rb[i])
                    BUILD_TRY(hipMemcpy(const_cast<uint8_t*>(ptrs[i]), files[i]->data() + metas[i].data_offset + r0 * src_rb[i],
                                        (size_t)(rows * src_rb[i]), hipMemcpyHostToDevice));
            CombineArgs a;
            a.src = (const uint8_t* const*)d_ptr.p;
            a.src_row_bytes = (const uint64_t*)d_rb.p;
            a.bit_off = (const uint64_t*)d_off.p;
            a.dst = (uint8_t*)d_out.p;
            a.dst_row_bytes = out_rb;
            a.rows = rows;
            a.nsrc = (uint32_t)n;
            BUILD_TRY(launch_combine(a, nullptr));
            BUILD_TRY(hipMemcpy(host_out.data(), d_out.p, (size_t)(rows * out_rb), hipMemcpyDeviceToHost));
            if (!write_all(f, host_out.data(), (size_t)(rows * out_rb))) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        }
        closer.f = nullptr;
        if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        return COBS_GPU_OK;
    });
}

// classic_construct_random (classic_index.cpp:661-725, `cobs classic-construct-random`,
// src/cobs.cpp:243-291): num_documents documents of document_size random 31-mers each,
// canonicalised, hashed num_hashes times into signature_size rows; names file_%06u; k = 31,
// canonicalize = 1.  The random stream is this library's own counter generator (see
// random_build_kernel), not std::mt19937: same distribution, different bits than the reference
// produces for the same seed.
cobs_gpu_status cobs_gpu_construct_random(const char* out_path, uint64_t signature_size, uint64_t num_documents,
                                          uint64_t document_size, uint64_t num_hashes, uint64_t seed, int device) {
    if (!out_path || signature_size == 0 || signature_size > (1ull << 46) || num_documents == 0 ||
        num_documents > 0xFFFFFFF0ull || num_hashes == 0 || num_hashes > 64)
        return cobs_gpu_set_error(COBS_GPU_ERR_ARG, "bad argument");
    return guarded([&]() -> cobs_gpu_status {
        cobs_gpu_status st = pick_device(device);
        if (st != COBS_GPU_OK) return st;
        const uint64_t row_size = (num_documents + 7) / 8, row_bytes = (row_size + 3) / 4 * 4;
        DevMem d_mat;
        BUILD_TRY(hipMalloc(&d_mat.p, (size_t)(signature_size * row_bytes)));
        BUILD_TRY(hipMemset(d_mat.p, 0, (size_t)(signature_size * row_bytes)));
        // launches of at most 2^31 k-mers
        const uint64_t per = document_size ? std::max<uint64_t>(1, (1ull << 31) / document_size) : num_documents;
        for (uint64_t d0 = 0; d0 < num_documents && document_size; d0 += per) {
            RandomBuildArgs a;
            a.matrix = (uint32_t*)d_mat.p;
            a.signature_size = signature_size;
            a.magic = ~0ull / signature_size;
            a.row_bytes = row_bytes;
            a.doc0 = d0;
            a.document_size = document_size;
            a.seed = seed;
            a.num_hashes = (uint32_t)num_hashes;
            BUILD_TRY(launch_random_build(a, std::min(per, num_documents - d0), nullptr));
        }
        BUILD_TRY(hipStreamSynchronize(nullptr));
        std::string h = "COBS:CLASSIC_INDEX";
        put<uint32_t>(h, 1);
        put<uint32_t>(h, 31);
        put<uint8_t>(h, 1);
        put<uint32_t>(h, (uint32_t)num_documents);
        put<uint64_t>(h, signature_size);
        put<uint64_t>(h, num_hashes);
        char nm[32];
        for (uint64_t i = 0; i < num_documents; ++i) { std::snprintf(nm, sizeof nm, "file_%06u\n", (unsigned)i); h += nm; }
        h += "CLASSIC_INDEX";
        FILE* f = std::fopen(out_path, "wb");
        if (!f) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, (std::string("could not create ") + out_path).c_str());
        struct Closer { FILE* f; ~Closer() { if (f) std::fclose(f); } } closer{f};
        if (!write_all(f, h.data(), h.size())) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        st = stream_rows_to_file(f, (const uint8_t*)d_mat.p, row_bytes, row_size, signature_size);
        if (st != COBS_GPU_OK) return st;
        closer.f = nullptr;
        if (std::fclose(f) != 0) return cobs_gpu_set_error(COBS_GPU_ERR_OPEN, "short write");
        return COBS_GPU_OK;
    });
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
