// cobs_amd/csrc/cobs_gpu_tools.cpp -- the construction-side sub-tools of the reference's `cobs`
// program on top of libcobs_gpu.so, with their flags and output (reference src/cobs.cpp):
//
//   cobs_gpu_query doc-list PATH / doc-dump PATH (host only)                       (:75-161)
//   cobs_gpu_query print-parameters [-h H] [-f FPR] [-n N] / print-kmers QUERY [-k K] (host only)   (:532-599)
//   cobs_gpu_query classic-construct INPUT OUT.cobs_classic [flags]               (:163-244)
//   cobs_gpu_query compact-construct INPUT OUT.cobs_compact [flags] [-p PAGE]     (:294-380)
//   cobs_gpu_query classic-combine IN_DIR OUT.cobs_classic                        (:1044-1060)
//   cobs_gpu_query compact-construct-combine IN_DIR OUT.cobs_compact [-p PAGE]   (:383-408)
//
// Flags of the two constructors: --file-type, -h/--num-hashes, -f/--false-positive-rate,
// -k/--term-size, --no-canonicalize, -C/--clobber, --continue; -m/--memory, -T/--threads,
// --keep-temporary, --tmp-path are accepted and ignored (there are no temporary files: the matrix
// is built in HBM), -d/--device selects the GPU.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <string>
#include <vector>

#include "../../include/cobs_gpu_construct.hpp"

namespace {

int fail() {
    std::fprintf(stderr, "EXCEPTION: %s\n", cobs_gpu_last_error());
    return 1;
}

struct Args {
    std::vector<std::string> positional;
    std::string file_type = "any";
    cobs_gpu_build_params p{};
    bool no_canonicalize = false, clobber = false, cont = false;
    bool parse(int argc, char** argv, bool compact) {
        p.struct_size = sizeof p;
        p.term_size = 31;
        p.canonicalize = 1;
        p.num_hashes = 1;
        p.false_positive_rate = 0.3;
        p.device = -1;
        for (int i = 0; i < argc; ++i) {
            const std::string a = argv[i];
            auto need = [&]() -> const char* {
                if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(1); }
                return argv[++i];
            };
            if (a == "--file-type") file_type = need();
            else if (a == "-h" || a == "--num-hashes") p.num_hashes = (uint32_t)std::strtoul(need(), nullptr, 10);
            else if (a == "-f" || a == "--false-positive-rate") p.false_positive_rate = std::atof(need());
            else if (a == "-k" || a == "--term-size") p.term_size = (uint32_t)std::strtoul(need(), nullptr, 10);
            else if (compact && (a == "-p" || a == "--page-size")) p.page_size = std::strtoull(need(), nullptr, 10);
            else if (a == "--no-canonicalize") no_canonicalize = true;
            else if (a == "-C" || a == "--clobber") clobber = true;
            else if (a == "--continue") cont = true;
            else if (a == "-d" || a == "--device") p.device = std::atoi(need());
            else if (a == "-m" || a == "--memory" || a == "-T" || a == "--threads" || a == "--tmp-path") (void)need();
            else if (a == "--keep-temporary") {}
            else if (!a.empty() && a[0] == '-' && a.size() > 1) { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); return false; }
            else positional.push_back(a);
        }
        p.canonicalize = no_canonicalize ? 0 : 1;
        return true;
    }
};

bool ends_with(const std::string& s, const char* suffix) {
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// what `cobs classic-construct` / `compact-construct` print before they build (reference src/cobs.cpp:41-73
// print_document_list, called at :235 and :373): one line per document, then the k-mer statistics
void print_document_list(const cobs_gpu::DocumentList& filelist, size_t term_size, std::ostream& os = std::cout) {
    size_t min_kmers = size_t(-1), max_kmers = 0, total_kmers = 0;
    const size_t n = filelist.size();
    os << "--- document list (" << n << " entries) ---" << std::endl;
    for (size_t i = 0; i < n; ++i) {
        const cobs_gpu::DocumentEntry d = filelist[i];
        const size_t num_terms = d.num_terms(term_size);
        std::error_code ec;
        const auto bytes = std::filesystem::file_size(d.path_, ec);
        os << "document[" << i << "] size " << (ec ? (std::uintmax_t)d.size_ : bytes) << " " << term_size << "-mers " << num_terms
           << " : " << d.path_ << " : " << d.name_ << std::endl;
        min_kmers = std::min(min_kmers, num_terms);
        max_kmers = std::max(max_kmers, num_terms);
        total_kmers += num_terms;
    }
    os << "--- end of document list (" << n << " entries) ---" << std::endl;
    os << "documents: " << n << std::endl;
    if (n != 0) {
        os << "minimum " << term_size << "-mers: " << min_kmers << std::endl;
        os << "maximum " << term_size << "-mers: " << max_kmers << std::endl;
        os << "average " << term_size << "-mers: " << static_cast<size_t>(static_cast<double>(total_kmers) / n) << std::endl;
        os << "total " << term_size << "-mers: " << total_kmers << std::endl;
    }
}

int construct(int argc, char** argv, bool compact) {
    Args a;
    if (!a.parse(argc, argv, compact) || a.positional.size() != 2) {
        std::fprintf(stderr, "usage: cobs_gpu_query %s INPUT OUT_FILE [--file-type T] [-h HASHES] [-f FPR] [-k K]%s "
                             "[--no-canonicalize] [-C] [-d DEVICE]\n",
                     compact ? "compact-construct" : "classic-construct", compact ? " [-p PAGE_SIZE]" : "");
        return 1;
    }
    // the C++17 mirror of the reference's construction API (include/cobs_gpu_construct.hpp),
    // statement for statement what src/cobs.cpp:235-241 / :373-377 do
    cobs_gpu::DocumentList filelist(a.positional[0], cobs_gpu::StringToFileType(a.file_type));
    print_document_list(filelist, a.p.term_size);
    if (compact) {
        cobs_gpu::CompactIndexParameters p;
        p.term_size = a.p.term_size; p.canonicalize = (uint8_t)a.p.canonicalize; p.num_hashes = a.p.num_hashes;
        p.false_positive_rate = a.p.false_positive_rate; p.page_size = a.p.page_size;
        p.clobber = a.clobber; p.continue_ = a.cont; p.device = a.p.device;
        cobs_gpu::compact_construct(filelist, a.positional[1], "", p);
    } else {
        cobs_gpu::ClassicIndexParameters p;
        p.term_size = a.p.term_size; p.canonicalize = (uint8_t)a.p.canonicalize; p.num_hashes = a.p.num_hashes;
        p.false_positive_rate = a.p.false_positive_rate;
        p.clobber = a.clobber; p.continue_ = a.cont; p.device = a.p.device;
        cobs_gpu::classic_construct(filelist, a.positional[1], "", p);
    }
    return 0;
}

// canonicalize_kmer of the reference (cobs/util/query.cpp:143-199) for `doc-dump`: the first strict difference between
// the forward base and the complement of the mirrored base among the first k/2 positions decides (forward smaller or
// no difference: the k-mer as it is; else its reverse complement); a character that is not A, C, G or T maps to 0 and
// makes the k-mer invalid.  -> false if the k-mer holds such a character
bool canonicalize_kmer(const char* in, char* out, size_t k) {
    auto fwd = [](char c) -> char { return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? c : 0; };
    auto rev = [](char c) -> char { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : 0; };
    bool good = true, reverse = false;
    for (size_t i = 0; i < k / 2; ++i) {
        const char f = fwd(in[i]), r = rev(in[k - 1 - i]);
        if (f != r) { reverse = r < f; break; }
    }
    for (size_t i = 0; i < k; ++i) {
        const char c = reverse ? rev(in[k - 1 - i]) : fwd(in[i]);
        good = good && c != 0;
        out[i] = c;
    }
    return good;
}

// `cobs doc-list PATH [--file-type T] [-k K]` and `cobs doc-dump PATH [-k K] [--no-canonicalize] [--file-type T]`
// (reference src/cobs.cpp:75-161): host-only views of a document list
int doc_tool(int argc, char** argv, bool dump) {
    std::string path, file_type = "any";
    unsigned term_size = 31;
    bool no_canonicalize = false;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--file-type" && i + 1 < argc) file_type = argv[++i];
        else if ((a == "-k" || a == "--term-size") && i + 1 < argc) term_size = (unsigned)std::strtoul(argv[++i], nullptr, 10);
        else if (dump && a == "--no-canonicalize") no_canonicalize = true;
        else if (!a.empty() && a[0] == '-' && a.size() > 1) { std::fprintf(stderr, "unknown flag %s\n", a.c_str()); return 1; }
        else path = a;
    }
    if (path.empty() || term_size == 0) {
        std::fprintf(stderr, "usage: cobs_gpu_query %s PATH [-k K] [--file-type T]%s\n", dump ? "doc-dump" : "doc-list",
                     dump ? " [--no-canonicalize]" : "");
        return 1;
    }
    cobs_gpu::DocumentList filelist(path, cobs_gpu::StringToFileType(file_type));
    if (!dump) {
        print_document_list(filelist, term_size);
        return 0;
    }
    std::cerr << "Found " << filelist.size() << " documents." << std::endl;
    std::vector<char> kmer(term_size);
    for (size_t i = 0; i < filelist.size(); ++i) {
        const cobs_gpu::DocumentEntry d = filelist[i];
        std::cerr << "document[" << i << "] : " << d.path_ << " : " << d.name_ << std::endl;
        d.process_terms(term_size, [&](const char* t) {
            if (no_canonicalize) { std::cout.write(t, term_size) << '\n'; return; }
            if (!canonicalize_kmer(t, kmer.data(), term_size)) std::cout << "Invalid DNA base pair: " << std::string(t, term_size) << std::endl;
            else std::cout.write(kmer.data(), term_size) << '\n';
        });
        std::cout.flush();
        std::cerr << "document[" << i << "] : " << d.num_terms(term_size) << " terms." << std::endl;
    }
    return 0;
}

// `cobs print-parameters [-h H] [-f FPR] [-n N]` (reference src/cobs.cpp:532-568): the signature size ratio
// -h / ln(1 - fpr^(1/h)) (cobs/util/calc_signature_size.cpp:17-24), or with -n the signature size
// ceil(n * ratio) (:26-34) in bits and bytes.  -n takes the reference's byte-size spellings (4096, 4K, 4Ki, 2Mi ...).
int print_parameters(int argc, char** argv) {
    unsigned num_hashes = 1;
    double fpr = 0.3;
    uint64_t num_elements = 0;
    auto size_arg = [](const char* t, uint64_t& out) {
        char* end = nullptr;
        const double v = std::strtod(t, &end);
        if (end == t || v < 0) return false;
        double unit = 1;
        if (*end != 0) {
            const char* units = "KMGTPE";
            const char c = *end >= 'a' && *end <= 'z' ? (char)(*end - 32) : *end;
            const char* u = std::strchr(units, c);
            if (u != nullptr) {
                const bool iec = end[1] == 'i';
                unit = std::pow(iec ? 1024.0 : 1000.0, (double)(u - units + 1));
                end += iec ? 2 : 1;
            }
            if (*end == 'B' || *end == 'b') ++end;
            if (*end != 0) return false;
        }
        out = (uint64_t)(v * unit);
        return true;
    };
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if ((a == "-h" || a == "--num-hashes") && i + 1 < argc) num_hashes = (unsigned)std::strtoul(argv[++i], nullptr, 10);
        else if ((a == "-f" || a == "--false-positive-rate") && i + 1 < argc) fpr = std::strtod(argv[++i], nullptr);
        else if ((a == "-n" || a == "--num-elements") && i + 1 < argc && size_arg(argv[i + 1], num_elements)) ++i;
        else { std::fprintf(stderr, "usage: cobs_gpu_query print-parameters [-h NUM_HASHES] [-f FALSE_POSITIVE_RATE] [-n NUM_ELEMENTS]\n"); return 1; }
    }
    const double ratio = -(double)num_hashes / std::log(1 - std::pow(fpr, 1 / (double)num_hashes));
    if (!(ratio > 0)) {          // (the reference dies here: die_unless(result > 0))
        std::fprintf(stderr, "EXCEPTION: no signature size for %u hashes at false positive rate %g\n", num_hashes, fpr);
        return 1;
    }
    if (num_elements == 0) {
        std::cout << ratio << '\n';
        return 0;
    }
    const uint64_t signature_size = (uint64_t)std::ceil((double)num_elements * ratio);
    // tlx::format_iec_units (tlx is an un-vendored submodule of the reference; its published form): the number over
    // the largest power of 1024 below it, three decimals, a space and "", "Ki", "Mi" ...
    double scaled = (double)(signature_size / 8);
    unsigned scale = 0;
    static const char* endings[] = {"", "Ki", "Mi", "Gi", "Ti", "Pi", "Ei"};
    while (scaled >= 1024.0) { scaled /= 1024.0; ++scale; }
    char iec[64];
    std::snprintf(iec, sizeof iec, "%.3f %s", scaled, endings[scale]);
    std::cout << "signature_size = " << signature_size << '\n';
    std::cout << "signature_bytes = " << signature_size / 8 << " = " << iec << '\n';
    return 0;
}

// `cobs print-kmers QUERY [-k K]` (reference src/cobs.cpp:570-599): the canonical form of the query's k-mers, one per
// line, "Invalid DNA base pair: ..." for one with a character outside ACGT.  Like the reference it walks
// i < |query| - k: the query's LAST k-mer is not printed (and a query shorter than k prints nothing here; there the
// unsigned difference wraps).
int print_kmers(int argc, char** argv) {
    std::string query;
    unsigned k = 31;
    bool have = false;
    for (int i = 0; i < argc; ++i) {
        const std::string a = argv[i];
        if ((a == "-k" || a == "--kmer-size") && i + 1 < argc) k = (unsigned)std::strtoul(argv[++i], nullptr, 10);
        else if (!have && (a.empty() || a[0] != '-')) { query = a; have = true; }
        else { have = false; break; }
    }
    if (!have || k == 0) {
        std::fprintf(stderr, "usage: cobs_gpu_query print-kmers QUERY [-k KMER_SIZE]\n");
        return 1;
    }
    std::vector<char> kmer(k);
    for (size_t i = 0; i + k < query.size(); ++i) {
        if (!canonicalize_kmer(query.data() + i, kmer.data(), k))
            std::cout << "Invalid DNA base pair: " << std::string(query.data() + i, k) << std::endl;
        else
            std::cout << std::string(kmer.data(), k) << '\n';
    }
    return 0;
}

}  // namespace

// -> -1 if argv[1] is not one of the sub-tools, else the exit code
static int tools(int argc, char** argv);

int cobs_gpu_tools_main(int argc, char** argv) {
    try {
        return tools(argc, argv);
    } catch (const cobs_gpu::Error& e) {
        // the reference prints "EXCEPTION: ..." and returns -1 (src/cobs.cpp:1070-1076) or exits
        std::fprintf(stderr, "%s%s\n", e.status == COBS_GPU_ERR_ARG && std::strncmp(e.what(), "COBS_GPU", 8) != 0 ? "" : "EXCEPTION: ", e.what());
        return 1;
    }
}

static int tools(int argc, char** argv) {
    if (argc < 2) return -1;
    const std::string tool = argv[1];
    if (tool == "doc-list") return doc_tool(argc - 2, argv + 2, false);
    if (tool == "doc-dump") return doc_tool(argc - 2, argv + 2, true);
    if (tool == "print-parameters") return print_parameters(argc - 2, argv + 2);
    if (tool == "print-kmers") return print_kmers(argc - 2, argv + 2);
    if (tool == "classic-construct") return construct(argc - 2, argv + 2, false);
    if (tool == "compact-construct") return construct(argc - 2, argv + 2, true);
    if (tool == "classic-combine" || tool == "compact-construct-combine") {
        // `cobs classic-combine IN_DIR OUT_FILE`: the .cobs_classic files of a directory, in path order
        std::vector<std::string> pos;
        int device = -1;
        uint64_t mem = 0, page_size = 8192;               // default of compact-construct-combine (src/cobs.cpp:392-395)
        const bool compact = tool == "compact-construct-combine";
        for (int i = 2; i < argc; ++i) {
            const std::string a = argv[i];
            if ((a == "-d" || a == "--device") && i + 1 < argc) device = std::atoi(argv[++i]);
            else if (compact && (a == "-p" || a == "--page-size") && i + 1 < argc) page_size = std::strtoull(argv[++i], nullptr, 10);
            else if ((a == "-m" || a == "--memory") && i + 1 < argc) mem = std::strtoull(argv[++i], nullptr, 10);
            else if ((a == "-T" || a == "--threads") && i + 1 < argc) ++i;
            else if (a == "--keep-temporary") {}
            else pos.push_back(a);
        }
        if (pos.size() != 2) {
            std::fprintf(stderr, "usage: cobs_gpu_query classic-combine IN_DIR OUT_FILE [-m BYTES] [-d DEVICE]\n"
                                 "       cobs_gpu_query compact-construct-combine IN_DIR OUT_FILE [-p PAGE_SIZE]\n");
            return 1;
        }
        std::vector<std::string> files;
        std::error_code ec;
        for (std::filesystem::recursive_directory_iterator it(pos[0], ec), end; !ec && it != end; it.increment(ec))
            if (!it->is_directory() && ends_with(it->path().string(), ".cobs_classic")) files.push_back(it->path().string());
        std::sort(files.begin(), files.end());
        if (files.empty()) { std::fprintf(stderr, "no .cobs_classic files in %s\n", pos[0].c_str()); return 1; }
        std::vector<const char*> cp;
        for (auto& f : files) cp.push_back(f.c_str());
        if (compact) return cobs_gpu_combine_compact(cp.data(), cp.size(), pos[1].c_str(), page_size) == COBS_GPU_OK ? 0 : fail();
        return cobs_gpu_combine_classic(cp.data(), cp.size(), pos[1].c_str(), mem, device) == COBS_GPU_OK ? 0 : fail();
    }
    return -1;
}
