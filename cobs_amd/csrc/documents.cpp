// cobs_amd/csrc/documents.cpp -- document lists and document readers in front of the GPU
// construction (the data formats on the input side of the path).
//
// What the reference's readers emit through process_terms(k, callback) is restated here as TERM
// TEXT: bytes whose k-grams are exactly the callback's terms, in the callback's order, which is
// what build_kernel hashes.  Every reader follows its reference counterpart's observable
// behaviour, buffer-edge effects included, so that an index built from the same files holds the
// same bits:
//   Text        cobs/text_file.hpp:26-72        (64 KiB buffer, overlap copied from offset wb)
//   Cortex      cobs/cortex_file.hpp:29-158     (McCortex v6, one colour; k-mer decode of kmer.hpp)
//   KMerBuffer  cobs/kmer_buffer.hpp:49-73, cobs/file/kmer_buffer_header.cpp:20-37
//   Fasta       cobs/fasta_file.hpp:53-183      (index pass + term pass with the stale `pos`)
//   Fastq       cobs/fastq_file.hpp:53-198
//   FastaMulti  cobs/fasta_multifile.hpp:38-63, 134-180
//   lists       cobs/document_list.hpp:154-411
// The .cobs_cache side files of the reference are never written or read.
#include "documents.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <thread>
#include <tuple>

namespace fs = std::filesystem;

__attribute__((visibility("hidden"))) cobs_gpu_status cobs_gpu_set_error(cobs_gpu_status st, const char* msg);   // engine.cpp

namespace cobs_amd {

namespace {

cobs_gpu_status err(cobs_gpu_status st, const std::string& m) { return cobs_gpu_set_error(st, m.c_str()); }

bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && std::memcmp(s.data() + s.size() - n, suffix, n) == 0;
}

// a file's bytes while it is parsed
struct View {
    const char* p = nullptr;
    size_t n = 0;
    const char* data() const { return p ? p : ""; }        // never null: memchr / memcpy of 0 bytes stay defined
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    char operator[](size_t i) const { return p[i]; }
};

// The bytes of a document file: a read-only mapping of the page cache (parsed in place: between
// the page cache and the staging buffer a character is copied once), or -- .gz, when `gunzip`: the
// reference inflates FASTA / FASTQ paths ending in .gz only, fasta_file.hpp:92-103,
// fastq_file.hpp:93-104 -- inflated into the caller's scratch string.
struct FileBytes {
    void* map = nullptr;
    size_t len = 0;
    View v;
    FileBytes() = default;
    FileBytes(const FileBytes&) = delete;
    FileBytes& operator=(const FileBytes&) = delete;
    ~FileBytes() { if (map) ::munmap(map, len); }
};

cobs_gpu_status read_file(const std::string& path, bool gunzip, std::string& scratch, FileBytes& fb) {
    if (gunzip && ends_with(path, ".gz")) {
        scratch.clear();
        gzFile g = gzopen(path.c_str(), "rb");
        if (!g) return err(COBS_GPU_ERR_OPEN, "could not open document " + path);
        gzbuffer(g, 1u << 20);
        char buf[1 << 16];
        for (;;) {
            const int n = gzread(g, buf, sizeof buf);
            if (n < 0) { gzclose(g); return err(COBS_GPU_ERR_FORMAT, "corrupt gzip stream in " + path); }
            if (n == 0) break;
            scratch.append(buf, (size_t)n);
        }
        gzclose(g);
        fb.v = View{scratch.data(), scratch.size()};
        return COBS_GPU_OK;
    }
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return err(COBS_GPU_ERR_OPEN, "could not open document " + path);
    struct stat st;
    if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        return err(COBS_GPU_ERR_OPEN, "could not open document " + path);
    }
    if (st.st_size > 0) {
        void* m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            return err(COBS_GPU_ERR_OPEN, "could not map document " + path);
        }
        (void)::madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
        fb.map = m;
        fb.len = (size_t)st.st_size;
        fb.v = View{(const char*)m, (size_t)st.st_size};
    }
    ::close(fd);
    return COBS_GPU_OK;
}

// std::getline over a buffer: calls fn(ptr, len) per line; a trailing '\n' does not start another
// line, a last line without '\n' is a line.  fn returns false to stop.  -> offset after the last
// line consumed (what tellg() reports)
template <typename F>
size_t for_lines(const View& data, size_t pos, F fn) {
    while (pos < data.size()) {
        const void* nl = std::memchr(data.data() + pos, '\n', data.size() - pos);
        const size_t end = nl ? (size_t)((const char*)nl - data.data()) : data.size();
        const size_t next = nl ? end + 1 : end;
        if (!fn(data.data() + pos, end - pos, next)) return next;
        pos = next;
    }
    return pos;
}

// cobs::base_name (cobs/util/file.hpp:69-76): file name cut at its first '.'
std::string base_name(const std::string& path) {
    std::string r = fs::path(path).filename().string();
    const size_t dot = r.find('.');
    return dot == std::string::npos ? r : r.substr(0, dot);
}

std::string pad_index(uint64_t i, int width = 6) {      // cobs/util/misc.hpp:57-60
    char b[32];
    std::snprintf(b, sizeof b, "%0*u", width, (unsigned)i);
    return b;
}

// ---- Text -----------------------------------------------------------------------------------
// text_file.hpp:44-66: a 64 KiB buffer is refilled behind the last term_size - 1 characters --
// which are copied from offset wb - (k - 1), i.e. measured from the bytes READ, not from the
// buffer's fill pos + wb; from the second refill on the carried characters are therefore not the
// last ones.  Each buffer state is one raw stretch of term text.
cobs_gpu_status text_terms(const View& data, uint32_t k, TermSink& out, std::vector<TermSeg>& segs) {
    constexpr size_t kBuf = 64 * 1024;
    if (k == 0 || k > kBuf) return err(COBS_GPU_ERR_UNSUPPORTED, "term size does not fit the text reader's buffer");
    std::vector<char> buffer(kBuf);
    size_t pos = 0, off = 0;
    bool eof = false;
    while (!eof) {
        const size_t want = kBuf - pos;
        const size_t wb = std::min(want, data.size() - off);
        std::memcpy(buffer.data() + pos, data.data() + off, wb);
        off += wb;
        eof = wb < want;                            // istream::read sets eofbit when it comes up short
        if (pos + wb >= k) {
            segs.push_back(TermSeg{(uint64_t)out.size, (uint64_t)(pos + wb), true});
            out.put(buffer.data(), pos + wb);
        }
        if (wb + 1 < k) break;
        std::memmove(buffer.data(), buffer.data() + wb - (k - 1), k - 1);
        pos = k - 1;
    }
    return COBS_GPU_OK;
}

// ---- k-mers packed four bases to a byte (kmer.hpp:69-99, kmer.cpp:148-213) ---------------------
// The string of a packed k-mer reads its bytes from the last to the first, each byte its four
// bases from the high bit pair down (00 A, 01 C, 10 G, 11 T); a k-mer whose length is not a
// multiple of four leaves out the first 4 - k % 4 bases of the first byte read.
void unpack_kmer(const uint8_t* packed, uint32_t kmer_size, TermSink& out) {
    const uint32_t nbytes = (kmer_size + 3) / 4;
    char buf[4];
    for (uint32_t i = 0; i < nbytes; ++i) {
        const uint8_t b = packed[nbytes - 1 - i];
        const uint32_t first = (i == 0 && kmer_size % 4 != 0) ? 4 - kmer_size % 4 : 0;
        for (uint32_t j = 0; j < 4; ++j) buf[j] = "ACGT"[(b >> (6 - 2 * j)) & 3];
        out.put(buf + first, 4 - first);
    }
}

// ---- Cortex ----------------------------------------------------------------------------------
struct CortexHeader {
    uint32_t version = 0, kmer_size = 0, words = 0, colors = 0;
    std::string name;
    size_t data_begin = 0;
};

cobs_gpu_status cortex_header(const View& d, const std::string& path, CortexHeader& h) {
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= d.size(); };
    auto magic = [&]() {
        if (!need(6) || std::memcmp(d.data() + p, "CORTEX", 6) != 0) return false;
        p += 6;
        return true;
    };
    auto u32 = [&](uint32_t& v) {
        if (!need(4)) return false;
        std::memcpy(&v, d.data() + p, 4);
        p += 4;
        return true;
    };
    const std::string bad = "CortexFile: magic number not found @ " + path;
    if (!magic()) return err(COBS_GPU_ERR_FORMAT, bad);
    if (!u32(h.version)) return err(COBS_GPU_ERR_FORMAT, "corrupted .ctx file");
    if (h.version != 6) return err(COBS_GPU_ERR_FORMAT, "Invalid .ctx file version (" + std::to_string(h.version) + ")");
    if (!u32(h.kmer_size) || !u32(h.words) || !u32(h.colors)) return err(COBS_GPU_ERR_FORMAT, "corrupted .ctx file");
    if (h.colors != 1)
        return err(COBS_GPU_ERR_FORMAT, "Invalid number of colors (" + std::to_string(h.colors) + "), must be 1");
    p += 12 * (size_t)h.colors;                     // mean read length u32 + total length u64 per colour
    for (uint32_t c = 0; c < h.colors; ++c) {
        uint32_t n = 0;
        if (!u32(n) || !need(n)) return err(COBS_GPU_ERR_FORMAT, "corrupted .ctx file");
        h.name.assign(d.data() + p, n);
        p += n;
    }
    p += 16 * (size_t)h.colors;                     // error rates
    for (uint32_t c = 0; c < h.colors; ++c) {
        p += 12;                                    // cleaning flags and thresholds
        uint32_t n = 0;
        if (!u32(n)) return err(COBS_GPU_ERR_FORMAT, "corrupted .ctx file");
        p += n;                                     // graph name
    }
    if (!magic()) return err(COBS_GPU_ERR_FORMAT, bad);
    // (64-bit arithmetic: the fields are whatever the file says)
    if (h.kmer_size == 0 || h.words == 0 || (uint64_t)8 * h.words < ((uint64_t)h.kmer_size + 3) / 4 || p > d.size())
        return err(COBS_GPU_ERR_FORMAT, "corrupted .ctx file");
    h.data_begin = p;
    return COBS_GPU_OK;
}

uint64_t cortex_num_kmers(const CortexHeader& h, size_t file_size) {
    return (file_size - h.data_begin) / (8ull * h.words + 5ull * h.colors);
}

// one record = the packed k-mer (8 * words bytes) + 5 bytes of colour data; the k-mer string is a
// sequence of its own (cortex_file.hpp:118-152)
cobs_gpu_status cortex_terms(const View& d, const std::string& path, uint32_t k, TermSink& out,
                             std::vector<TermSeg>& segs) {
    CortexHeader h;
    cobs_gpu_status st = cortex_header(d, path, h);
    if (st != COBS_GPU_OK) return st;
    const uint64_t n = cortex_num_kmers(h, d.size());
    const size_t rec = 8 * (size_t)h.words + 5 * (size_t)h.colors;
    const uint64_t begin = out.size;
    if (k <= h.kmer_size) {
        for (uint64_t r = 0; r < n; ++r) {
            unpack_kmer((const uint8_t*)d.data() + h.data_begin + r * rec, h.kmer_size, out);
            out.put('\n');
        }
    }
    segs.push_back(TermSeg{begin, (uint64_t)out.size - begin, false});
    return COBS_GPU_OK;
}

// ---- KMerBuffer (.cobs_doc) -------------------------------------------------------------------
struct KMerBufferHeader {
    uint32_t kmer_size = 0;
    std::string name;
    size_t data_begin = 0;
};

cobs_gpu_status kmer_buffer_header(const View& d, const std::string& path, KMerBufferHeader& h) {
    static const char kBegin[] = "COBS:DOCUMENT";
    size_t p = 0;
    if (d.size() < 13 + 8 || std::memcmp(d.data(), kBegin, 13) != 0)
        return err(COBS_GPU_ERR_FORMAT, "invalid file type: " + path);
    p = 13;
    uint32_t version = 0;
    std::memcpy(&version, d.data() + p, 4);
    p += 4;
    if (version != 1) return err(COBS_GPU_ERR_FORMAT, "invalid file version: " + path);
    std::memcpy(&h.kmer_size, d.data() + p, 4);
    p += 4;
    const void* z = std::memchr(d.data() + p, 0, d.size() - p);
    if (!z) return err(COBS_GPU_ERR_FORMAT, "invalid file type: " + path);
    h.name.assign(d.data() + p, (const char*)z - (d.data() + p));
    p += h.name.size() + 1;
    if (d.size() < p + 8 || std::memcmp(d.data() + p, "DOCUMENT", 8) != 0)
        return err(COBS_GPU_ERR_FORMAT, "invalid file type: " + path);
    h.data_begin = p + 8;
    if (h.kmer_size == 0 || h.kmer_size > (1u << 20)) return err(COBS_GPU_ERR_FORMAT, "invalid file type: " + path);
    return COBS_GPU_OK;
}

cobs_gpu_status kmer_buffer_terms(const View& d, const std::string& path, uint32_t k, TermSink& out,
                                  std::vector<TermSeg>& segs) {
    KMerBufferHeader h;
    cobs_gpu_status st = kmer_buffer_header(d, path, h);
    if (st != COBS_GPU_OK) return st;
    // document_list.hpp:116-129: only 31-mers, read through KMerBuffer<31>
    if (k != 31 || h.kmer_size != 31) return err(COBS_GPU_ERR_UNSUPPORTED, ".cobs_doc documents hold 31-mers only: " + path);
    const size_t rec = 8;
    const uint64_t n = (d.size() - h.data_begin) / rec;
    const uint64_t begin = out.size;
    for (uint64_t r = 0; r < n; ++r) {
        unpack_kmer((const uint8_t*)d.data() + h.data_begin + r * rec, 31, out);
        out.put('\n');
    }
    segs.push_back(TermSeg{begin, (uint64_t)out.size - begin, false});
    return COBS_GPU_OK;
}

// ---- FASTA ------------------------------------------------------------------------------------
bool is_comment(char c) { return c == '>' || c == ';'; }

// compute_index (fasta_file.hpp:53-90): size and the histogram of sequence lengths, a sequence
// being the lines between comment / empty lines
cobs_gpu_status fasta_index(const View& d, const std::string& path, DocEntry& e) {
    e.size = 0;
    e.run_hist.clear();
    // the first getline running into the end of the file (no '\n' at all) leaves an empty index (:62-63)
    if (std::memchr(d.data(), '\n', d.size()) == nullptr) return COBS_GPU_OK;
    uint64_t run = 0;
    bool first = true, bad = false;
    for_lines(d, 0, [&](const char* ln, size_t n, size_t) {
        if (first) {
            first = false;
            if (n == 0 || !is_comment(ln[0])) { bad = true; return false; }
            e.size += n + 1;
            return true;
        }
        e.size += n + 1;
        if (n == 0 || is_comment(ln[0])) {
            if (run) ++e.run_hist[run];
            run = 0;
        } else {
            run += n;
        }
        return true;
    });
    if (bad) return err(COBS_GPU_ERR_FORMAT, "FastaFile: file does not start with > or ; - " + path);
    if (run) ++e.run_hist[run];
    return COBS_GPU_OK;
}

// process_terms (fasta_file.hpp:155-182).  The reference appends every line to ONE string and,
// after emitting its k-grams, keeps the last k - 1 characters; a line is taken as a comment by
// the character at index `pos` of that string -- pos is 0 while the string is shorter than k (the
// test then looks at the first character of the sequence so far, and a comment line is swallowed
// into it) and keeps its old value k - 1 after the string was cleared (a line of exactly k - 1
// characters then counts as empty; a shorter one is read past its end, taken as sequence here).
// Emitted: each sequence so far as one line of term text.
void fasta_terms(const View& d, uint32_t k, TermSink& out) {
    size_t run_begin = out.size;     // the sequence being collected lies at out.data[run_begin, out.size)
    size_t held = 0, pos = 0;        // held = how many of its characters the reference still has
    auto flush = [&]() {
        if (out.size > run_begin) out.put('\n');
        run_begin = out.size;
        held = 0;
    };
    for_lines(d, 0, [&](const char* ln, size_t n, size_t) {
        if (out.overflow) return false;      // the sink is full (the file grew since it was listed): `held` would outrun out.size
        const size_t size = held + n;
        bool comment;
        if (size == pos) comment = true;
        else if (pos < size) comment = is_comment(pos < held ? out.data[out.size - held + pos] : ln[pos - held]);
        else comment = false;
        if (comment) { flush(); return true; }
        out.put(ln, n);
        if (size > k - 1) { held = k - 1; pos = k - 1; }
        else { held = size; pos = 0; }
        return true;
    });
    flush();
}

// ---- FASTQ ------------------------------------------------------------------------------------
// compute_index / process_terms (fastq_file.hpp:53-86, 163-182): records of four lines, the second
// is the read
cobs_gpu_status fastq_scan(const View& d, const std::string& path, DocEntry* index, TermSink* text) {
    uint64_t line_num = 0, size = 0;
    std::string bad;
    for_lines(d, 0, [&](const char* ln, size_t n, size_t) {
        size += n + 1;
        switch (line_num % 4) {
        case 0:
            if (n == 0 || ln[0] != '@') bad = " does not start with @ - ";
            break;
        case 1:
            if (index) ++index->run_hist[n];
            if (text) { text->put(ln, n); text->put('\n'); }
            break;
        case 2:
            if (n == 0 || ln[0] != '+') bad = " does not start with + - ";
            break;
        default: break;
        }
        if (!bad.empty()) return false;
        ++line_num;
        return true;
    });
    if (!bad.empty()) return err(COBS_GPU_ERR_FORMAT, "FastqFile: line " + std::to_string(line_num) + bad + path);
    if (index) index->size = size;
    return COBS_GPU_OK;
}

// ---- multi-FASTA ------------------------------------------------------------------------------
struct Subdoc {
    uint64_t pos_begin, size;
};

// compute_index (fasta_multifile.hpp:134-180): a sub-document per '>' line, running to the next
// line that starts with '>' or ';'; its size is the sum of its line lengths
cobs_gpu_status mfasta_index(const View& d, const std::string& path, std::vector<Subdoc>& out) {
    if (d.empty() || !is_comment(d[0]))
        return err(COBS_GPU_ERR_FORMAT, "FastaMultifile: file does not start with > or ; - " + path);
    bool in_doc = false;
    Subdoc cur{0, 0};
    for_lines(d, 0, [&](const char* ln, size_t n, size_t next) {
        const char c0 = n ? ln[0] : '\0';
        if (in_doc && !is_comment(c0)) { cur.size += n; return true; }
        if (in_doc) { out.push_back(cur); in_doc = false; }
        // a header that is the file's last line WITHOUT a newline leaves the stream at its end and the
        // reference's loop (`while (is.good())`) stops before taking it
        const bool has_newline = ln + n < d.data() + d.size();
        if (c0 == '>' && has_newline) { in_doc = true; cur = Subdoc{(uint64_t)next, 0}; }
        return true;                    // ';' comments, empty and stray lines between documents are skipped
    });
    if (in_doc) out.push_back(cur);
    return COBS_GPU_OK;
}

// FastaSubfile::process_terms (fasta_multifile.hpp:38-63): lines are appended to one string whose
// k-grams are emitted, then `data.erase(0, data.size() - k + 1)` keeps the last k - 1 characters --
// computed in size_t, so a string shorter than k - 1 is erased completely.
void mfasta_terms(const View& d, uint64_t pos_begin, uint32_t k, TermSink& out) {
    size_t run_begin = out.size;     // the sequence so far lies at out.data[run_begin, out.size)
    size_t held = 0;
    auto end_run = [&]() {
        if (out.size - run_begin >= k) out.put('\n');
        else out.size = run_begin;                  // no term in it: leave nothing behind
        run_begin = out.size;
        held = 0;
    };
    for_lines(d, (size_t)pos_begin, [&](const char* ln, size_t n, size_t) {
        if (n && is_comment(ln[0])) return false;
        if (held + n == 0) return true;
        out.put(ln, n);
        const size_t size = held + n;
        if (size + 1 < k) end_run();        // everything is dropped: the sequence so far ends here
        else held = k - 1;
        return true;
    });
    end_run();
}

}  // namespace

// ------------------------------------------------------------------------------------------------

FileType identify_filetype(const std::string& p) {
    if (ends_with(p, ".txt")) return FileType::Text;
    if (ends_with(p, ".ctx") || ends_with(p, ".cortex")) return FileType::Cortex;
    if (ends_with(p, ".cobs_doc")) return FileType::KMerBuffer;
    for (const char* e : {".fa", ".fasta", ".fna", ".ffn", ".faa", ".frn"})
        if (ends_with(p, e) || ends_with(p, (std::string(e) + ".gz").c_str())) return FileType::Fasta;
    for (const char* e : {".fq", ".fastq"})
        if (ends_with(p, e) || ends_with(p, (std::string(e) + ".gz").c_str())) return FileType::Fastq;
    if (ends_with(p, ".mfasta")) return FileType::FastaMulti;
    if (ends_with(p, ".mfastq")) return FileType::FastqMulti;
    if (ends_with(p, ".list")) return FileType::List;
    return FileType::Any;
}

bool parse_filetype(const std::string& in, FileType& out) {       // cobs/document_list.cpp:15-32
    std::string s = in;
    for (char& c : s) c = (char)std::tolower((unsigned char)c);
    if (s == "any" || s == "*") out = FileType::Any;
    else if (s == "text" || s == "txt") out = FileType::Text;
    else if (s == "cortex" || s == "ctx") out = FileType::Cortex;
    else if (s == "cobs" || s == "cobs_doc") out = FileType::KMerBuffer;
    else if (s == "fasta") out = FileType::Fasta;
    else if (s == "fastq") out = FileType::Fastq;
    else if (s == "list") out = FileType::List;
    else return false;
    return true;
}

cobs_gpu_status load_entries(const std::string& path, std::vector<DocEntry>& out) {
    const FileType ft = identify_filetype(path);
    DocEntry e;
    e.path = path;
    e.type = ft;
    std::error_code ec;
    std::string scratch;
    FileBytes fb;
    const View& d = fb.v;
    cobs_gpu_status st;
    switch (ft) {
    case FileType::Text: {
        e.name = base_name(path);
        e.size = (uint64_t)fs::file_size(path, ec);
        if (ec) return err(COBS_GPU_ERR_OPEN, "could not open document " + path);
        out.push_back(std::move(e));
        return COBS_GPU_OK;
    }
    case FileType::Cortex: {
        if ((st = read_file(path, false, scratch, fb)) != COBS_GPU_OK) return st;
        CortexHeader h;
        if ((st = cortex_header(d, path, h)) != COBS_GPU_OK) return st;
        e.name = h.name;
        e.size = d.size();
        e.term_size = h.kmer_size;
        e.term_count = cortex_num_kmers(h, d.size());
        out.push_back(std::move(e));
        return COBS_GPU_OK;
    }
    case FileType::KMerBuffer: {
        if ((st = read_file(path, false, scratch, fb)) != COBS_GPU_OK) return st;
        KMerBufferHeader h;
        if ((st = kmer_buffer_header(d, path, h)) != COBS_GPU_OK) return st;
        e.name = h.name;
        e.size = d.size();
        e.term_size = h.kmer_size;
        e.term_count = (d.size() - h.data_begin) / ((h.kmer_size + 3) / 4);
        out.push_back(std::move(e));
        return COBS_GPU_OK;
    }
    case FileType::Fasta: {
        if ((st = read_file(path, true, scratch, fb)) != COBS_GPU_OK) return st;
        if ((st = fasta_index(d, path, e)) != COBS_GPU_OK) return st;
        e.data_bytes = d.size();
        e.name = base_name(path);
        out.push_back(std::move(e));
        return COBS_GPU_OK;
    }
    case FileType::Fastq: {
        if ((st = read_file(path, true, scratch, fb)) != COBS_GPU_OK) return st;
        if ((st = fastq_scan(d, path, &e, nullptr)) != COBS_GPU_OK) return st;
        e.data_bytes = d.size();
        e.name = base_name(path);
        out.push_back(std::move(e));
        return COBS_GPU_OK;
    }
    case FileType::FastaMulti: {
        if ((st = read_file(path, false, scratch, fb)) != COBS_GPU_OK) return st;
        std::vector<Subdoc> subs;
        if ((st = mfasta_index(d, path, subs)) != COBS_GPU_OK) return st;
        for (size_t i = 0; i < subs.size(); ++i) {
            DocEntry s = e;
            s.name = base_name(path) + '_' + pad_index(i);
            s.size = subs[i].size;
            s.subdoc_index = i;
            s.pos_begin = subs[i].pos_begin;
            out.push_back(std::move(s));
        }
        return COBS_GPU_OK;
    }
    default:
        return err(COBS_GPU_ERR_FORMAT, "DocumentList: unknown document file to add: " + path);
    }
}

static bool accept(const std::string& path, FileType filter) {   // document_list.hpp:165-196
    const FileType ft = identify_filetype(path);
    if (filter == FileType::Any)
        return ft == FileType::Text || ft == FileType::Cortex || ft == FileType::KMerBuffer || ft == FileType::Fasta ||
               ft == FileType::Fastq || ft == FileType::FastaMulti || ft == FileType::FastqMulti;
    return ft == filter;
}

void sort_entries(std::vector<DocEntry>& list, uint32_t by) {
    if (by == COBS_GPU_SORT_BY_SIZE)         // sort_by_size, document_list.hpp:424-430
        std::stable_sort(list.begin(), list.end(), [](const DocEntry& a, const DocEntry& b) {
            return std::tie(a.size, a.path) < std::tie(b.size, b.path);
        });
    else                                     // DocumentEntry::operator<, document_list.hpp:78-82
        std::stable_sort(list.begin(), list.end(), [](const DocEntry& a, const DocEntry& b) {
            return std::tie(a.path, a.subdoc_index) < std::tie(b.path, b.subdoc_index);
        });
}

cobs_gpu_status add_recursive(const std::string& root, FileType filter, std::vector<DocEntry>& list) {
    std::vector<std::string> paths;
    std::error_code ec;
    if (fs::is_directory(root, ec)) {
        for (fs::recursive_directory_iterator it(root, ec), end; !ec && it != end; it.increment(ec))
            if (!it->is_directory() && accept(it->path().string(), filter)) paths.push_back(it->path().string());
        if (ec) return err(COBS_GPU_ERR_OPEN, "could not scan directory " + root);
    } else if (ends_with(root, ".list") || filter == FileType::List) {
        std::string scratch;
        FileBytes fb;
        if (read_file(root, false, scratch, fb) != COBS_GPU_OK) return err(COBS_GPU_ERR_OPEN, "DocumentList: could not open .list file: " + root);
        const View& d = fb.v;
        const fs::path parent = fs::path(root).parent_path();
        for_lines(d, 0, [&](const char* ln, size_t n, size_t) {
            if (n == 0 || ln[0] == '#') return true;
            fs::path file(std::string(ln, n));
            if (!file.is_absolute()) file = parent / file;
            paths.push_back(file.string());
            return true;
        });
    } else if (fs::is_regular_file(root, ec)) {
        paths.push_back(root);
    }
    std::sort(paths.begin(), paths.end());
    const bool single = paths.size() == 1 && paths[0] == root;
    // the index pass reads every file once; host threads share the files (the reference runs
    // this loop in a parallel_for too, :387-403)
    struct Loaded {
        std::vector<DocEntry> entries;
        cobs_gpu_status status = COBS_GPU_OK;
        std::string error;
    };
    std::vector<Loaded> loaded(paths.size());
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < paths.size();) {
            loaded[i].status = load_entries(paths[i], loaded[i].entries);
            if (loaded[i].status != COBS_GPU_OK) loaded[i].error = cobs_gpu_last_error();
        }
    };
    const size_t nthreads = std::min<size_t>({paths.size(), std::max(1u, std::thread::hardware_concurrency()), 64});
    if (nthreads <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (size_t t = 0; t < nthreads; ++t) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
    for (Loaded& l : loaded) {
        if (l.status != COBS_GPU_OK && single) return err(l.status, l.error);
        // a file that cannot be read is reported and left out, the scan goes on (:393-402)
        if (l.status != COBS_GPU_OK) {
            std::fprintf(stderr, "EXCEPTION: %s\n", l.error.c_str());
            continue;
        }
        for (DocEntry& e : l.entries) list.push_back(std::move(e));
    }
    sort_entries(list, COBS_GPU_SORT_BY_PATH);
    return COBS_GPU_OK;
}

// k-grams of '\n'-separated sequences
static uint64_t count_terms(const std::string& t, uint32_t k) {
    uint64_t total = 0, run = 0;
    for (size_t i = 0; i <= t.size(); ++i) {
        if (i == t.size() || t[i] == '\n') {
            if (run >= k) total += run - k + 1;
            run = 0;
        } else {
            ++run;
        }
    }
    return total;
}

uint64_t num_terms(const DocEntry& e, uint32_t k) {          // document_list.hpp:85-112
    switch (e.type) {
    case FileType::Text:
    case FileType::FastaMulti:
        return e.size < k ? 0 : e.size - k + 1;
    case FileType::Cortex:
    case FileType::KMerBuffer:
        return e.term_size >= k ? e.term_count * (e.term_size - k + 1) : 0;
    case FileType::Fasta:
    case FileType::Fastq: {
        uint64_t total = 0;
        for (const auto& p : e.run_hist) total += p.second * (p.first < k ? 0 : p.first - k + 1);
        return total;
    }
    case FileType::Memory:
        return count_terms(e.text, k);
    default:
        return 0;
    }
}

uint64_t term_text_bound(const DocEntry& e, uint32_t k) {
    switch (e.type) {
    case FileType::Text:        // one stretch per refill of the reader's buffer, each carrying k - 1 characters over
        return e.size + (uint64_t)(k - 1) * (e.size / (64 * 1024 - std::min<uint64_t>(k - 1, 64 * 1024 - 1)) + 2);
    case FileType::Cortex:
    case FileType::KMerBuffer:
        return e.term_count * (e.term_size + 1);
    case FileType::Memory:
        return e.text.size() + 1;
    case FileType::Fasta:
    case FileType::Fastq:       // line formats: every sequence costs its characters plus one separator
        return e.data_bytes + 1;
    default:
        return e.size + 1;
    }
}

cobs_gpu_status load_terms(const DocEntry& e, uint32_t k, TermSink& out, std::vector<TermSeg>& segs, std::string& scratch) {
    if (k == 0) return err(COBS_GPU_ERR_ARG, "term size 0");
    // a text file is read through a 64 KiB buffer that carries k - 1 characters over: with k near the buffer size
    // the term text (and its bound) grows to ~ size x k
    if (e.type == FileType::Text && k > 32 * 1024) return err(COBS_GPU_ERR_UNSUPPORTED, "term size above 32768 for text documents");
    FileBytes fb;
    const View& d = fb.v;
    cobs_gpu_status st = COBS_GPU_OK;
    const uint64_t begin = out.size;
    bool one_stretch = true;
    switch (e.type) {
    case FileType::Memory:
        out.put(e.text.data(), e.text.size());
        out.put('\n');
        break;
    case FileType::Text:
        if ((st = read_file(e.path, false, scratch, fb)) != COBS_GPU_OK) return st;
        st = text_terms(d, k, out, segs);
        one_stretch = false;
        break;
    case FileType::Cortex:
        if ((st = read_file(e.path, false, scratch, fb)) != COBS_GPU_OK) return st;
        st = cortex_terms(d, e.path, k, out, segs);
        one_stretch = false;
        break;
    case FileType::KMerBuffer:
        if ((st = read_file(e.path, false, scratch, fb)) != COBS_GPU_OK) return st;
        st = kmer_buffer_terms(d, e.path, k, out, segs);
        one_stretch = false;
        break;
    case FileType::Fasta:
        if ((st = read_file(e.path, true, scratch, fb)) != COBS_GPU_OK) return st;
        fasta_terms(d, k, out);
        break;
    case FileType::Fastq:
        if ((st = read_file(e.path, true, scratch, fb)) != COBS_GPU_OK) return st;
        if ((st = fastq_scan(d, e.path, nullptr, &out)) != COBS_GPU_OK) return st;
        break;
    case FileType::FastaMulti:
        if ((st = read_file(e.path, false, scratch, fb)) != COBS_GPU_OK) return st;
        mfasta_terms(d, e.pos_begin, k, out);
        break;
    default:
        return err(COBS_GPU_ERR_FORMAT, "DocumentEntry: unknown file type");
    }
    if (st != COBS_GPU_OK) return st;
    // the space was sized from what the list recorded about the file (term_text_bound)
    if (out.overflow) return err(COBS_GPU_ERR_FORMAT, "document changed since it was listed: " + e.path);
    if (one_stretch) segs.push_back(TermSeg{begin, (uint64_t)out.size - begin, false});
    return COBS_GPU_OK;
}

}  // namespace cobs_amd
