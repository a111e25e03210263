// pq_encode — the command line of the reference's extraction/descriptor_PQ.py (:274-283, :286-349, :352-377) for the part of it
// that belongs to the matcher's data path: turning templates with fp32 texture descriptors into gallery templates with PQ codes,
// with the nearest-codeword search done on an MI355X through the C ABI (afis_encode_rolled_dat).
//
//   pq_encode --fprint_type rolled --input_dir <dir>/ --output_dir <dir>/ [-c <codebook.dat>] [-d <device>]
//   pq_encode --fprint_type latent --input_dir <dir>/ --output_dir <dir>/      (re-writes latent templates unchanged, :297-309)
//   pq_encode --fprint_type latent --input_file <f.dat> --output_dir <dir>/    (:352-364)
//
// As in the reference: the directory arguments are string-prefixes (they must end with '/'); the output name is the input's
// basename up to its first '.', plus ".dat"; rolled inputs are processed in the order of the integer formed by the digits of
// their path; a rolled input without a texture template produces the 2-byte file the reference writes (:338-342); the codebook
// comes from -c or from CodebookPath in ../afis.config; "PQ: <file>" is printed per input; `--fprint_type rolled --input_file`
// prints the reference's "not available" message.
// Deliberate differences: inputs are in the matcher's own latent .dat layout (fp32 texture descriptors, descriptor_PQ.py:80-175)
// instead of the extraction-internal TF_C layout (template_2.py:730-840, out of scope); every texture template of a file is
// encoded (the reference encodes only the first and writes the others' floats as bytes, :343-349).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <string>
#include <vector>

#include "../../include/afis_matcher.h"
#include "template_io.h"

namespace fs = std::filesystem;
using namespace afis;

static std::string arg_of(int argc, char** argv, const char* name)
{
    for (int i = 1; i + 1 < argc; ++i) if (!strcmp(argv[i], name)) return argv[i + 1];
    return "";
}

static std::string config_value(const std::string& key)               // flat JSON: "key": "value"
{
    std::ifstream f((fs::current_path().parent_path() / "afis.config").string());
    std::string s((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const size_t k = s.find("\"" + key + "\"");
    if (k == std::string::npos) return "";
    const size_t c = s.find(':', k), a = s.find('"', c), b = s.find('"', a + 1);
    return (c == std::string::npos || a == std::string::npos || b == std::string::npos) ? "" : s.substr(a + 1, b - a - 1);
}

static std::vector<fs::path> glob_dat(const std::string& prefix)      // glob.glob(input_dir + '*.dat')
{
    std::vector<fs::path> out;
    const fs::path dir = fs::path(prefix).parent_path();
    const std::string stem_prefix = fs::path(prefix).filename().string();
    std::error_code ec;
    for (const auto& e : fs::directory_iterator(dir.empty() ? fs::path(".") : dir, ec)) {
        const std::string name = e.path().filename().string();
        if (e.is_regular_file() && e.path().extension() == ".dat" && name.compare(0, stem_prefix.size(), stem_prefix) == 0) out.push_back(e.path());
    }
    return out;
}

static std::string out_name(const std::string& output_dir, const fs::path& in)
{
    const std::string base = in.filename().string();
    return output_dir + base.substr(0, base.find('.')) + ".dat";
}

static bool write_file(const std::string& path, const void* p, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write((const char*)p, (std::streamsize)n);
    return (bool)f;
}

int main(int argc, char** argv)
{
    const std::string type = arg_of(argc, argv, "--fprint_type").empty() ? "latent" : arg_of(argc, argv, "--fprint_type");
    const std::string input_dir = arg_of(argc, argv, "--input_dir"), input_file = arg_of(argc, argv, "--input_file");
    const std::string output_dir = arg_of(argc, argv, "--output_dir");
    std::string lower = type; std::transform(lower.begin(), lower.end(), lower.begin(), ::tolower);
    const bool is_latent = lower == "latent";
    if (output_dir.empty() || (input_dir.empty() && input_file.empty())) { std::cout << "Missing args." << std::endl; return 0; }
    std::error_code ec; fs::create_directories(fs::path(output_dir), ec);

    if (is_latent) {                                                    // :297-309 / :352-364: read and re-write
        std::vector<fs::path> files;
        if (!input_dir.empty()) { files = glob_dat(input_dir); std::sort(files.begin(), files.end()); }
        else files.push_back(input_file);
        std::vector<uint8_t> b;
        for (const fs::path& f : files) {
            std::cout << "PQ: " << f.string() << std::endl;
            HostTemplate t;
            if (!read_file(f.string(), b)) { std::cerr << "pq_encode: cannot read " << f << std::endl; return 2; }
            parse_latent_dat(b.data(), b.size(), t);
            const std::vector<uint8_t> w = write_latent_dat(t);
            if (!write_file(out_name(output_dir, f), w.data(), w.size())) { std::cerr << "pq_encode: cannot write to " << output_dir << std::endl; return 2; }
        }
        return 0;
    }
    if (input_dir.empty()) {                                            // :365-366
        std::cout << "Single template PQ is not available for rolled prints. Please specify an input directory instead." << std::endl;
        return 0;
    }

    std::string codebook_path = arg_of(argc, argv, "-c");
    if (codebook_path.empty()) codebook_path = config_value("CodebookPath");
    std::vector<uint8_t> cb;
    if (codebook_path.empty() || !read_file(codebook_path, cb) || cb.empty()) { std::cerr << "pq_encode: no codebook (-c or CodebookPath in ../afis.config)" << std::endl; return 2; }
    const std::string dev = arg_of(argc, argv, "-d");
    afis_ctx* ctx = nullptr;
    if (int rc = afis_create_from_codebook(&ctx, cb.data(), cb.size(), dev.empty() ? 0 : atoi(dev.c_str())); rc != AFIS_OK) {
        std::cerr << "pq_encode: afis_create failed (" << rc << "): " << afis_last_error(nullptr) << std::endl;
        return 2;
    }
    std::vector<fs::path> files = glob_dat(input_dir);
    auto digits = [](const fs::path& p) {                              // int(''.join(filter(str.isdigit, filename))), :331
        unsigned long long v = 0; bool any = false;
        for (char c : p.string()) if (c >= '0' && c <= '9') { v = v * 10 + (unsigned)(c - '0'); any = true; }
        return any ? v : 0ull;
    };
    std::stable_sort(files.begin(), files.end(), [&](const fs::path& a, const fs::path& b) { return digits(a) < digits(b); });
    std::vector<uint8_t> b, out;
    int ret = 0;
    for (const fs::path& f : files) {
        std::cout << "PQ: " << f.string() << std::endl;
        if (!read_file(f.string(), b)) { std::cerr << "pq_encode: cannot read " << f << std::endl; ret = 2; break; }
        HostTemplate probe;
        const int prc = parse_latent_dat(b.data(), b.size(), probe);
        if (prc != 0 || probe.tex.empty()) {                             // :338-342
            const uint16_t zero = 0;
            write_file(out_name(output_dir, f), &zero, 2);
            continue;
        }
        size_t need = 0; int load_rc = 0;
        if (afis_encode_rolled_dat(ctx, b.data(), b.size(), nullptr, 0, &need, &load_rc) != AFIS_OK) { std::cerr << "pq_encode: " << afis_last_error(ctx) << std::endl; ret = 2; break; }
        out.resize(need);
        if (afis_encode_rolled_dat(ctx, b.data(), b.size(), out.data(), out.size(), &need, &load_rc) != AFIS_OK) { std::cerr << "pq_encode: " << afis_last_error(ctx) << std::endl; ret = 2; break; }
        if (!write_file(out_name(output_dir, f), out.data(), need)) { std::cerr << "pq_encode: cannot write to " << output_dir << std::endl; ret = 2; break; }
    }
    afis_destroy(ctx);
    return ret;
}
