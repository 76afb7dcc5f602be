// cobs_amd/csrc/index_file.cpp -- header parsing + file mapping (host only).
#include "index_file.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>

namespace cobs_amd {

namespace {

// Bounds-checked forward reader over the mapped header bytes.
class Cursor {
public:
    Cursor(const uint8_t* p, size_t n) : p_(p), n_(n) {}
    bool ok() const { return ok_; }
    size_t pos() const { return pos_; }
    size_t left() const { return n_ - pos_; }
    void skip(size_t n) {
        if (!ok_ || n > n_ - pos_) { ok_ = false; return; }
        pos_ += n;
    }
    template <typename T>
    T pod() {
        T v{};
        if (!ok_ || sizeof(T) > n_ - pos_) { ok_ = false; return v; }
        std::memcpy(&v, p_ + pos_, sizeof(T));
        pos_ += sizeof(T);
        return v;
    }
    bool word(const char* w) {
        const size_t n = std::strlen(w);
        if (!ok_ || n > n_ - pos_ || std::memcmp(p_ + pos_, w, n) != 0) { ok_ = false; return false; }
        pos_ += n;
        return true;
    }
    // one '\n'-terminated line (the reference writes names with std::endl and
    // reads them back with std::getline)
    std::string line() {
        std::string out;
        if (!ok_) return out;
        const void* nl = std::memchr(p_ + pos_, '\n', n_ - pos_);
        if (!nl) { ok_ = false; return out; }
        const size_t end = (size_t)((const uint8_t*)nl - p_);
        out.assign((const char*)p_ + pos_, end - pos_);
        pos_ = end + 1;
        return out;
    }

private:
    const uint8_t* p_;
    size_t n_;
    size_t pos_ = 0;
    bool ok_ = true;
};

constexpr const char* kMagic = "COBS:";
constexpr const char* kClassicWord = "CLASSIC_INDEX";
constexpr const char* kCompactWord = "COMPACT_INDEX";
constexpr uint32_t kVersion = 1;

bool try_classic(const uint8_t* data, size_t len, IndexMeta& m) {
    Cursor c(data, len);
    if (!c.word(kMagic) || !c.word(kClassicWord)) return false;
    if (c.pod<uint32_t>() != kVersion || !c.ok()) return false;
    m = IndexMeta{};
    m.kind = IndexKind::Classic;
    m.term_size = c.pod<uint32_t>();
    m.canonicalize = c.pod<uint8_t>();
    const uint32_t ndocs = c.pod<uint32_t>();
    const uint64_t sig = c.pod<uint64_t>();
    m.num_hashes = c.pod<uint64_t>();
    if (!c.ok()) return false;
    if (ndocs > c.left()) return false;            // every name takes at least its '\n'
    m.signature_sizes.assign(1, sig);
    m.doc_names.reserve(ndocs);
    for (uint32_t i = 0; i < ndocs && c.ok(); ++i) m.doc_names.push_back(c.line());
    if (!c.ok() || !c.word(kClassicWord)) return false;
    m.data_offset = c.pos();
    return true;
}

bool try_compact(const uint8_t* data, size_t len, IndexMeta& m) {
    Cursor c(data, len);
    if (!c.word(kMagic) || !c.word(kCompactWord)) return false;
    if (c.pod<uint32_t>() != kVersion || !c.ok()) return false;
    m = IndexMeta{};
    m.kind = IndexKind::Compact;
    m.term_size = c.pod<uint32_t>();
    m.canonicalize = c.pod<uint8_t>();
    const uint32_t nparams = c.pod<uint32_t>();
    const uint32_t ndocs = c.pod<uint32_t>();
    m.header_page_size = c.pod<uint64_t>();
    if (!c.ok() || nparams == 0 || m.header_page_size == 0) return false;
    if (nparams > c.left() / 16 || ndocs > c.left()) return false;     // 16 bytes per sub-index, >= 1 per name
    if (m.header_page_size > (1ull << 28)) return false;                // also keeps the padding arithmetic small
    m.signature_sizes.reserve(nparams);
    for (uint32_t p = 0; p < nparams && c.ok(); ++p) {
        const uint64_t sig = c.pod<uint64_t>();
        const uint64_t nh = c.pod<uint64_t>();
        if (p == 0) m.num_hashes = nh;
        else if (nh != m.num_hashes) return false;   // all sub-indexes share H (compact_index/search_file.cpp:23-27)
        m.signature_sizes.push_back(sig);
    }
    m.doc_names.reserve(ndocs);
    for (uint32_t i = 0; i < ndocs && c.ok(); ++i) m.doc_names.push_back(c.line());
    if (!c.ok()) return false;
    // zero padding so that the matrix starts at a multiple of page_size
    const uint64_t ps = m.header_page_size;
    const uint64_t pad = (ps - ((c.pos() + std::strlen(kCompactWord)) % ps)) % ps;
    c.skip(pad);
    if (!c.word(kCompactWord)) return false;
    m.data_offset = c.pos();
    return true;
}

}  // namespace

bool parse_index_header(const uint8_t* data, size_t len, IndexMeta& meta, std::string& err) {
    if (try_classic(data, len, meta) || try_compact(data, len, meta)) {
        // matrix bytes with checked arithmetic: a crafted signature_size must not wrap to a small number
        uint64_t need = meta.data_offset;
        const uint64_t prb = meta.page_row_bytes();
        for (uint64_t sig : meta.signature_sizes) {
            uint64_t bytes = 0;
            if (__builtin_mul_overflow(prb, sig, &bytes) || __builtin_add_overflow(need, bytes, &need)) {
                err = "index header describes a matrix larger than 2^64 bytes";
                return false;
            }
        }
        if (need > len) {
            err = "index file is shorter than its header promises";
            return false;
        }
        return true;
    }
    err = "not a COBS classic or compact index";
    return false;
}

MappedFile::~MappedFile() {
    if (data_) ::munmap(const_cast<uint8_t*>(data_), size_);
    if (fd_ >= 0) ::close(fd_);
}

bool MappedFile::open(const std::string& path, std::string& err) {
    fd_ = ::open(path.c_str(), O_RDONLY);
    if (fd_ < 0) { err = "could not open index file " + path + ": " + std::strerror(errno); return false; }
    struct stat st;
    if (::fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) { err = "not a regular file: " + path; return false; }
    if (st.st_size == 0) { err = "empty file: " + path; return false; }
    void* p = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (p == MAP_FAILED) { err = "mmap failed for " + path + ": " + std::strerror(errno); return false; }
    data_ = (const uint8_t*)p;
    size_ = (size_t)st.st_size;
    return true;
}

}  // namespace cobs_amd
