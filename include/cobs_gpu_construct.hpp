// include/cobs_gpu_construct.hpp -- C++17 host-side mirror of the reference's construction API,
// implemented over the C ABI of cobs_gpu.h (header only).  The documents are listed and parsed by
// libcobs_gpu.so, their terms hashed and the signature bits set on the GPU; the index files are
// byte-for-byte what the reference writes.
//
// Reference interface being mirrored (same names, argument meaning and defaults):
//   enum class cobs::FileType, cobs::StringToFileType            (cobs/document_list.hpp:35-56)
//   struct cobs::DocumentEntry { path_, type_, name_, size_, subdoc_index_, term_size_, term_count_;
//                                num_terms(k); process_terms(k, callback) }     (:62-151)
//   class  cobs::DocumentList { DocumentList(); DocumentList(root, filter); add; add_recursive;
//                               size; operator[]; sort_by_path; sort_by_size }   (:154-430)
//   struct cobs::ClassicIndexParameters, cobs::CompactIndexParameters
//                                       (cobs/construction/classic_index.hpp:29-53, compact_index.hpp:24-45)
//   void cobs::classic_construct(const DocumentList&, out_file, tmp_path, params)   (classic_index.hpp:62-64)
//   void cobs::compact_construct(DocumentList, index_file, tmp_path, params)        (compact_index.hpp:52-54)
//   void cobs::classic_construct_random(out_file, signature_size, num_documents,
//                                       document_size, num_hashes, seed)            (classic_index.hpp:83-86)
//   void cobs::compact_combine_into_compact(in_dir..., out_file, page_size)          (compact_index.hpp:56-60)
// Where the reference terminates the process (die / exit) these throw cobs_gpu::Error.  There is no
// temporary directory (the matrix is built in HBM): tmp_path, mem_bytes, num_threads and
// keep_temporary are accepted and unused.
#pragma once

#include <cstdint>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "cobs_gpu_construct.h"
#include "cobs_gpu_search.hpp"

namespace cobs_gpu {

enum class FileType : uint32_t {
    Any = COBS_GPU_FILETYPE_ANY, Text = COBS_GPU_FILETYPE_TEXT, Cortex = COBS_GPU_FILETYPE_CORTEX,
    KMerBuffer = COBS_GPU_FILETYPE_KMER_BUFFER, Fasta = COBS_GPU_FILETYPE_FASTA, Fastq = COBS_GPU_FILETYPE_FASTQ,
    FastaMulti = COBS_GPU_FILETYPE_FASTA_MULTI, FastqMulti = COBS_GPU_FILETYPE_FASTQ_MULTI, List = COBS_GPU_FILETYPE_LIST,
};

namespace detail {
inline void check(cobs_gpu_status st) {
    if (st != COBS_GPU_OK) throw Error(st, cobs_gpu_last_error());
}
}  // namespace detail

inline FileType StringToFileType(const std::string& s) {
    uint32_t ft = 0;
    detail::check(cobs_gpu_filetype_from_string(s.c_str(), &ft));
    return (FileType)ft;
}

struct DocumentEntry {
    std::string path_;
    FileType type_ = FileType::Any;
    std::string name_;
    size_t size_ = 0;
    size_t subdoc_index_ = 0;
    size_t term_size_ = 0;
    size_t term_count_ = 0;

    //! calculate number of terms in file
    size_t num_terms(size_t k) const {
        uint64_t n = 0;
        detail::check(cobs_gpu_doclist_num_terms(list_, index_, (uint32_t)k, &n));
        return (size_t)n;
    }
    //! process terms: callback(const char* term) for every term (term_size characters, not terminated)
    template <typename Callback>
    void process_terms(unsigned term_size, Callback callback) const {
        uint64_t n = 0;
        detail::check(cobs_gpu_doclist_terms(list_, index_, term_size, nullptr, 0, &n));
        std::vector<char> buf((size_t)n * term_size + 1);
        detail::check(cobs_gpu_doclist_terms(list_, index_, term_size, buf.data(), (size_t)n * term_size, &n));
        for (uint64_t i = 0; i < n; ++i) callback(buf.data() + i * term_size);
    }

private:
    friend class DocumentList;
    const cobs_gpu_doclist* list_ = nullptr;     // valid while the DocumentList it came from is unchanged
    size_t index_ = 0;
};

class DocumentList {
public:
    DocumentList() { detail::check(cobs_gpu_doclist_create(&dl_)); }
    explicit DocumentList(const std::string& root, FileType filter = FileType::Any) : DocumentList() {
        add_recursive(root, filter);
    }
    ~DocumentList() { cobs_gpu_doclist_free(dl_); }
    DocumentList(const DocumentList&) = delete;
    DocumentList& operator=(const DocumentList&) = delete;
    DocumentList(DocumentList&& o) noexcept : dl_(o.dl_) { o.dl_ = nullptr; }

    void add(const std::string& path) { detail::check(cobs_gpu_doclist_add(dl_, path.c_str())); }
    void add_recursive(const std::string& root, FileType filter = FileType::Any) {
        detail::check(cobs_gpu_doclist_add_recursive(dl_, root.c_str(), (uint32_t)filter));
    }
    //! an in-memory document: its sequences joined by '\n' (no reference counterpart)
    void add_memory(const std::string& name, const std::string& text) {
        detail::check(cobs_gpu_doclist_add_memory(dl_, name.c_str(), text.data(), text.size()));
    }
    size_t size() const { return cobs_gpu_doclist_size(dl_); }
    DocumentEntry operator[](size_t i) const {
        cobs_gpu_doc_entry e;
        detail::check(cobs_gpu_doclist_entry(dl_, i, &e));
        DocumentEntry d;
        d.path_ = e.path;
        d.type_ = (FileType)e.type;
        d.name_ = e.name;
        d.size_ = (size_t)e.size;
        d.subdoc_index_ = (size_t)e.subdoc_index;
        d.term_size_ = (size_t)e.term_size;
        d.term_count_ = (size_t)e.term_count;
        d.list_ = dl_;
        d.index_ = i;
        return d;
    }
    void sort_by_path() { detail::check(cobs_gpu_doclist_sort(dl_, COBS_GPU_SORT_BY_PATH)); }
    void sort_by_size() { detail::check(cobs_gpu_doclist_sort(dl_, COBS_GPU_SORT_BY_SIZE)); }
    const cobs_gpu_doclist* handle() const { return dl_; }

private:
    cobs_gpu_doclist* dl_ = nullptr;
};

struct ClassicIndexParameters {
    unsigned term_size = 31;
    uint8_t canonicalize = 1;
    unsigned num_hashes = 1;
    double false_positive_rate = 0.3;
    uint64_t signature_size = 0;
    uint64_t mem_bytes = 0;          // unused: there are no temporary batches
    size_t num_threads = 0;          // unused
    bool clobber = false;
    bool continue_ = false;
    bool keep_temporary = false;     // unused
    int device = -1;                 // which GPU builds the index (not in the reference)
};

struct CompactIndexParameters {
    unsigned term_size = 31;
    uint8_t canonicalize = 1;
    unsigned num_hashes = 1;
    double false_positive_rate = 0.3;
    uint64_t page_size = 0;
    uint64_t mem_bytes = 0;
    size_t num_threads = 0;
    bool clobber = false;
    bool continue_ = false;
    bool keep_temporary = false;
    int device = -1;
};

namespace detail {
inline bool ends_with(const std::string& s, const char* suffix) {
    const std::string x(suffix);
    return s.size() >= x.size() && s.compare(s.size() - x.size(), x.size(), x) == 0;
}
inline void check_output(const std::string& out_file, const char* ext, bool clobber, bool cont) {
    // classic_index.cpp:596-612, compact_index.cpp:176-210
    if (!ends_with(out_file, ext)) throw Error(COBS_GPU_ERR_ARG, std::string("Error: COBS index file must end with ") + ext);
    struct stat st;
    if (::stat(out_file.c_str(), &st) == 0 && !clobber && !cont)
        throw Error(COBS_GPU_ERR_ARG, "Output file exists, will not overwrite without --clobber");
}
template <typename P>
cobs_gpu_build_params params_of(const P& p, uint64_t signature_size, uint64_t page_size) {
    cobs_gpu_build_params b{};
    b.struct_size = sizeof b;
    b.term_size = p.term_size;
    b.canonicalize = p.canonicalize;
    b.num_hashes = p.num_hashes;
    b.false_positive_rate = p.false_positive_rate;
    b.signature_size = signature_size;
    b.page_size = page_size;
    b.device = p.device;
    return b;
}
}  // namespace detail

inline void classic_construct(const DocumentList& filelist, const std::string& out_file, const std::string& /*tmp_path*/,
                              ClassicIndexParameters params) {
    detail::check_output(out_file, ".cobs_classic", params.clobber, params.continue_);
    const cobs_gpu_build_params b = detail::params_of(params, params.signature_size, 0);
    detail::check(cobs_gpu_build_classic_list(filelist.handle(), &b, out_file.c_str()));
}

inline void compact_construct(const DocumentList& doc_list, const std::string& index_file, const std::string& /*tmp_path*/,
                              CompactIndexParameters params) {
    detail::check_output(index_file, ".cobs_compact", params.clobber, params.continue_);
    const cobs_gpu_build_params b = detail::params_of(params, 0, params.page_size);
    detail::check(cobs_gpu_build_compact_list(doc_list.handle(), &b, index_file.c_str()));
}

//! the same construction straight into a query object: build -> query without an index file
inline ClassicSearch construct_search(const DocumentList& doc_list, const CompactIndexParameters& params, bool compact) {
    const cobs_gpu_build_params b = detail::params_of(params, 0, params.page_size);
    cobs_gpu_options o{};
    o.struct_size = sizeof o;
    o.device = params.device;
    cobs_gpu_index* ix = nullptr;
    detail::check(cobs_gpu_build_index_list(compact ? 1u : 0u, doc_list.handle(), &b, &o, &ix));
    return ClassicSearch(ix);
}

inline void classic_construct_random(const std::string& out_file, uint64_t signature_size, uint64_t num_documents,
                                     size_t document_size, uint64_t num_hashes, size_t seed, int device = -1) {
    detail::check(cobs_gpu_construct_random(out_file.c_str(), signature_size, num_documents, document_size, num_hashes, seed, device));
}

//! classic_combine over explicit files (the reference walks a directory: classic_index.hpp:76-78)
inline void classic_combine(const std::vector<std::string>& in_files, const std::string& out_file, uint64_t mem_bytes = 0,
                            int device = -1) {
    std::vector<const char*> cp;
    for (const auto& f : in_files) cp.push_back(f.c_str());
    detail::check(cobs_gpu_combine_classic(cp.data(), cp.size(), out_file.c_str(), mem_bytes, device));
}

inline void compact_combine_into_compact(const std::vector<std::string>& in_files, const std::string& out_file, uint64_t page_size) {
    std::vector<const char*> cp;
    for (const auto& f : in_files) cp.push_back(f.c_str());
    detail::check(cobs_gpu_combine_compact(cp.data(), cp.size(), out_file.c_str(), page_size));
}

}  // namespace cobs_gpu
