// match — drop-in for the reference's `./match` (matching/main.cpp:35-87 + the drivers Matcher::One2List_matching /
// Matcher::List2List_matching, matching/matcher.cpp:96-337), with the per-pair scoring done on an MI355X through the C ABI of
// include/afis_matcher.h.
//
//   ./match -l <latent.dat> | -ldir <latent dir>  -g <gallery dir>  -s <score dir>/  -c <codebook.dat>
//
// Same behaviour as the reference where it is defined:
//   * ../afis.config (relative to the PARENT of the current directory, main.cpp:41-44) supplies CodebookPath, ScorePath,
//     GalleryTemplateDirectory, LatentTemplateDirectory when a flag is absent; the flags win.  Unlike the reference, a missing
//     config file is only an error when a value is actually needed from it.
//   * -s is string-concatenated with the latent's stem (matcher.cpp:157,199,222): it must end with '/'.
//   * -l   writes <stem>.csv: header `filename,score`, then the top 24 as `<rank>"<path>",<score>` (default float formatting).
//   * -ldir writes one <stem>.csv per latent with one `"<path>",<score %.3f>` line per gallery file, in directory order.
//   * the same stdout lines (gallery size, template counts, rank table, total duration).
// Differences, all deliberate: the gallery is parsed once and kept in HBM instead of being re-read for every pair
// (matcher.cpp:173/:278); rank ties are broken by ascending gallery index (the reference's std::sort leaves them unspecified,
// matcher.cpp:306-309); the correspondence CSVs of the top 24, which the reference writes to the hard-coded
// /LatentAFIS/scores/corr<latent>_<rolled>_<i>.csv (matcher.cpp:325-327, :405, :497-505), go to <score dir>/corr<latent>_<rolled>_<i>.csv
// (or to the prefix given with -corr).
// Additions: -g may name a packed gallery container (one file, include/afis_matcher.h: afis_gallery_load) instead of a directory;
// -pack <file> writes the gallery given by -g as such a container (alone: pack and exit).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/afis_matcher.h"
#include "template_io.h"

namespace fs = std::filesystem;
using namespace afis;

namespace {

// argparser.h:5-24 — flat token search
struct ArgParser {
    std::vector<std::string> tokens;
    ArgParser(int argc, char** argv) { for (int i = 1; i < argc; ++i) tokens.push_back(argv[i]); }
    const std::string& getCmdOption(const std::string& option) const
    {
        static const std::string empty;
        auto it = std::find(tokens.begin(), tokens.end(), option);
        if (it != tokens.end() && ++it != tokens.end()) return *it;
        return empty;
    }
    bool cmdOptionExists(const std::string& option) const { return std::find(tokens.begin(), tokens.end(), option) != tokens.end(); }
};

// afis.config is a flat JSON object of string values (afis.config:1-17); this reads exactly that.
std::map<std::string, std::string> read_flat_json(const std::string& path)
{
    std::map<std::string, std::string> kv;
    std::ifstream in(path);
    if (!in) return kv;
    std::stringstream ss; ss << in.rdbuf();
    const std::string s = ss.str();
    size_t i = 0;
    auto read_string = [&](std::string& out) -> bool {
        while (i < s.size() && s[i] != '"') ++i;
        if (i >= s.size()) return false;
        ++i; out.clear();
        while (i < s.size() && s[i] != '"') { if (s[i] == '\\' && i + 1 < s.size()) ++i; out.push_back(s[i++]); }
        ++i;
        return true;
    };
    std::string k, v;
    while (read_string(k)) {
        while (i < s.size() && s[i] != ':' ) ++i;
        if (i >= s.size()) break;
        ++i;
        while (i < s.size() && isspace((unsigned char)s[i])) ++i;
        if (i < s.size() && s[i] == '"') { if (!read_string(v)) break; kv[k] = v; }
    }
    return kv;
}

std::vector<fs::path> list_dat(const std::string& dir)
{
    std::vector<fs::path> files;
    for (fs::directory_iterator it(dir), end; it != end; ++it)
        if (it->path().extension() == ".dat") files.push_back(it->path());     // directory order, unsorted (matcher.cpp:103-130)
    return files;
}

struct Latent {
    HostTemplate t;
    std::vector<afis_minutiae_view> mv;
    std::vector<afis_texture_view> tv;
    afis_template_view view;
    void load(const fs::path& p)
    {
        std::vector<uint8_t> b; read_file(p.string(), b);
        (void)parse_latent_dat(b.data(), b.size(), t);                           // return code ignored, as matcher.cpp:150/:259
        mv.clear(); tv.clear();
        for (const HostMinutiae& m : t.minu) mv.push_back({m.n(), m.x.data(), m.y.data(), m.ori.data(), m.des_len, m.des.data()});
        for (const HostTexture& x : t.tex) tv.push_back({x.n(), x.x.data(), x.y.data(), x.ori.data(), x.des_len, x.des.data(), nullptr});
        view = {(int)mv.size(), mv.data(), (int)tv.size(), tv.data()};
    }
};

#define CHECK(ctx, call) do { int rc_ = (call); if (rc_ != AFIS_OK) { std::cerr << "match: " #call " failed (" << rc_ << "): " << afis_last_error(ctx) << std::endl; return 2; } } while (0)

// -g is either the reference's directory of rolled .dat files or ONE packed gallery container (afis_gallery_save); the container
// carries the paths the templates came from, so the score files read the same either way.
std::vector<fs::path> list_gallery(const std::string& g)
{
    if (!fs::is_regular_file(fs::path(g))) return list_dat(g);
    std::vector<fs::path> files;
    size_t need = 0;
    if (afis_gallery_file_names(g.c_str(), 0, -1, nullptr, 0, &need) != AFIS_OK) { std::cerr << "match: " << afis_last_error(nullptr) << std::endl; return files; }
    std::vector<char> buf(need + 1);
    if (afis_gallery_file_names(g.c_str(), 0, -1, buf.data(), need, &need) != AFIS_OK) return files;
    for (size_t at = 0; at < need; at += strlen(&buf[at]) + 1) files.emplace_back(std::string(&buf[at]));
    return files;
}

int load_gallery(afis_ctx* ctx, const std::string& g, const std::vector<fs::path>& files, const std::string& pack_to)
{
    if (fs::is_regular_file(fs::path(g))) {
        CHECK(ctx, afis_gallery_load(ctx, g.c_str(), 0, -1));
    } else {
        std::vector<uint8_t> b;
        for (const fs::path& f : files) {
            read_file(f.string(), b);
            int load_rc = 0;
            CHECK(ctx, afis_gallery_add_dat(ctx, b.data(), b.size(), &load_rc));
        }
    }
    if (!pack_to.empty()) {
        std::vector<std::string> names; std::vector<const char*> np;
        for (const fs::path& f : files) names.push_back(f.string());
        for (const std::string& n : names) np.push_back(n.c_str());
        CHECK(ctx, afis_gallery_save(ctx, pack_to.c_str(), np.data()));
    }
    CHECK(ctx, afis_gallery_commit(ctx, 0));
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    ArgParser args(argc, argv);
    if (args.cmdOptionExists("-h") || args.cmdOptionExists("--help")) {
        std::cout << "usage: match -l <latent.dat> | -ldir <latent dir>  -g <gallery dir>  -s <score dir>/  -c <codebook.dat>  [-d <device>] [-corr <prefix>] [-pack <gallery container to write>]\n       -g may name a packed gallery container instead of a directory\n";
        return 0;
    }
    const auto config = read_flat_json((fs::current_path().parent_path() / "afis.config").string());
    auto from_config = [&](const char* key, std::string& out) -> bool {
        auto it = config.find(key);
        if (it == config.end()) { std::cerr << "match: no value for " << key << " (flag missing and ../afis.config has none)" << std::endl; return false; }
        out = it->second; return true;
    };

    std::string codebook_path, score_path, gallery_path;
    if (args.cmdOptionExists("-c")) codebook_path = args.getCmdOption("-c");
    else if (!from_config("CodebookPath", codebook_path)) return 2;
    if (args.cmdOptionExists("-s")) score_path = args.getCmdOption("-s");
    else { std::cout << "Missing argument for score directory. Using default from afis.config" << std::endl; if (!from_config("ScorePath", score_path)) return 2; }
    std::error_code ec; fs::create_directory(fs::path(score_path), ec);
    if (args.cmdOptionExists("-g")) gallery_path = args.getCmdOption("-g");
    else { std::cout << "Missing argument for gallery directory. Using default from afis.config" << std::endl; if (!from_config("GalleryTemplateDirectory", gallery_path)) return 2; }
    const int device = args.cmdOptionExists("-d") ? atoi(args.getCmdOption("-d").c_str()) : 0;

    std::vector<uint8_t> cb;
    if (!read_file(codebook_path, cb) || cb.empty()) { std::cout << "codebook is empty!" << std::endl; return 2; }
    afis_ctx* ctx = nullptr;
    if (int rc = afis_create_from_codebook(&ctx, cb.data(), cb.size(), device); rc != AFIS_OK) {
        std::cerr << "match: afis_create failed (" << rc << "): " << afis_last_error(nullptr) << std::endl;
        return 2;
    }

    using clk = std::chrono::high_resolution_clock;
    int ret = 0;
    const std::string pack_to = args.cmdOptionExists("-pack") ? args.getCmdOption("-pack") : "";
    if (!pack_to.empty() && !args.cmdOptionExists("-l") && !args.cmdOptionExists("-ldir")) {          // pack only
        std::vector<fs::path> rolled = list_gallery(gallery_path);
        if (rolled.empty()) { std::cout << "No rolled templates found in directory: " << gallery_path << std::endl; afis_destroy(ctx); return -1; }
        ret = load_gallery(ctx, gallery_path, rolled, pack_to);
        if (ret == 0) std::cout << "Packed " << rolled.size() << " templates into " << pack_to << std::endl;
        afis_destroy(ctx);
        return ret;
    }
    if (args.cmdOptionExists("-l")) {
        // ---- One2List_matching, matcher.cpp:216-337 ----
        const fs::path latent_file(args.getCmdOption("-l"));
        const std::string score_file = score_path + latent_file.stem().string() + ".csv";
        std::vector<fs::path> rolled = list_gallery(gallery_path);
        if (rolled.empty()) { std::cout << "No rolled templates found in directory: " << gallery_path << std::endl; afis_destroy(ctx); return -1; }
        const auto t0 = clk::now();
        std::cout << "Latent Query: " << latent_file << std::endl;
        std::cout << "Gallery size: " << rolled.size() << std::endl;
        if ((ret = load_gallery(ctx, gallery_path, rolled, pack_to)) != 0) { afis_destroy(ctx); return ret; }
        Latent L; L.load(latent_file);
        if (L.view.n_minu <= 0 && L.view.n_tex <= 0) { std::ofstream out(score_file); out << 0 << std::endl; }       // :260-268
        const int k = (int)std::min<size_t>(24, rolled.size());
        std::vector<int64_t> idx(k); std::vector<float> sc(k); int32_t status = 0;
        CHECK(ctx, afis_search(ctx, &L.view, 1, nullptr, nullptr, &status, k, idx.data(), sc.data()));
        if (status == AFIS_QUERY_LATENT_EMPTY) { std::cout << "Matching failed: latent template is empty. Exiting." << std::endl; afis_destroy(ctx); return 1; }
        std::ofstream out(score_file);
        out << "filename,score" << std::endl;
        std::cout << "Match Results" << std::endl << "----------------" << std::endl << "Rank     Filename      Score" << std::endl;
        // correspondence files for the top 24 (matcher.cpp:311-328): one "lx,ly,rx,ry" line per surviving correspondence
        std::vector<int32_t> counts((size_t)k * 3); std::vector<int16_t> xy((size_t)k * 3 * 120 * 4);
        CHECK(ctx, afis_correspondences(ctx, &L.view, idx.data(), k, counts.data(), xy.data()));
        const std::string corr_prefix = args.cmdOptionExists("-corr") ? args.getCmdOption("-corr") : score_path + "corr";
        for (int j = 0; j < k; ++j) {
            out << std::to_string(j + 1) << rolled[idx[j]] << "," << sc[j] << std::endl;
            for (int i = 0; i < 3; ++i) {
                const int n = counts[(size_t)j * 3 + i];
                if (n < 0) continue;
                std::ofstream cf(corr_prefix + latent_file.stem().string() + "_" + rolled[idx[j]].stem().string() + "_" + std::to_string(i) + ".csv");
                const int16_t* p = &xy[((size_t)j * 3 + i) * 120 * 4];
                for (int t = 0; t < n; ++t) cf << p[t * 4] << "," << p[t * 4 + 1] << "," << p[t * 4 + 2] << "," << p[t * 4 + 3] << std::endl;
            }
            std::cout << std::to_string(j + 1) << "        " << rolled[idx[j]].filename() << "       " << sc[j] << std::endl;
        }
        std::cout << "Total matching duration (ms): " << std::chrono::duration<double, std::milli>(clk::now() - t0).count() << std::endl;
    } else {
        // ---- List2List_matching, matcher.cpp:96-214 ----
        std::string latent_dir;
        if (args.cmdOptionExists("-ldir")) latent_dir = args.getCmdOption("-ldir");
        else {
            std::cout << "Missing argument for latent template or directory. Assuming batch matching, using default directory from afis.config" << std::endl;
            if (!from_config("LatentTemplateDirectory", latent_dir)) { afis_destroy(ctx); return 2; }
        }
        std::vector<fs::path> latents = list_dat(latent_dir);
        for (const fs::path& p : latents) std::cout << "latent template file" << p << std::endl;
        if (latents.empty()) { std::cout << "No latent templates found in directory: " << latent_dir << std::endl; afis_destroy(ctx); return -1; }
        std::vector<fs::path> rolled = list_gallery(gallery_path);
        for (const fs::path& p : rolled) std::cout << "rolled template file" << p << std::endl;
        if (rolled.empty()) { std::cout << "No rolled templates found in directory: " << gallery_path << std::endl; afis_destroy(ctx); return -1; }
        std::cout << "Gallery size: " << rolled.size() << std::endl;
        const auto t0 = clk::now();
        if ((ret = load_gallery(ctx, gallery_path, rolled, pack_to)) != 0) { afis_destroy(ctx); return ret; }
        const size_t G = rolled.size();
        const size_t batch = 16;
        for (size_t i0 = 0; i0 < latents.size(); i0 += batch) {
            const size_t nb = std::min(batch, latents.size() - i0);
            std::vector<Latent> Ls(nb); std::vector<afis_template_view> views(nb);
            for (size_t i = 0; i < nb; ++i) { Ls[i].load(latents[i0 + i]); views[i] = Ls[i].view; }
            std::vector<float> scores(nb * G); std::vector<int32_t> status(nb);
            CHECK(ctx, afis_search(ctx, views.data(), (int)nb, scores.data(), nullptr, status.data(), 0, nullptr, nullptr));
            for (size_t i = 0; i < nb; ++i) {
                const fs::path& lf = latents[i0 + i];
                std::cout << lf << std::endl;
                std::cout << "Latent minutiae templates: " << Ls[i].view.n_minu << std::endl;
                std::cout << "Latent texture templates: " << Ls[i].view.n_tex << std::endl;
                const std::string csv = score_path + lf.stem().string() + ".csv";
                if (Ls[i].view.n_minu <= 0 && Ls[i].view.n_tex <= 0) {                    // :153-163
                    std::cout << "No minutiae or texture templates found" << std::endl;
                    std::ofstream out(csv); out << 0 << std::endl;
                    continue;
                }
                if (status[i] == AFIS_QUERY_LATENT_EMPTY) { std::cout << "Matching failed: latent template is empty. Skipping." << std::endl; continue; }   // :191-194
                std::ofstream out(csv);
                for (size_t j = 0; j < G; ++j) out << rolled[j] << "," << std::setprecision(3) << std::fixed << scores[i * G + j] << std::endl;   // :201-204
            }
        }
        std::cout << "Total matching duration (ms): " << std::chrono::duration<double, std::milli>(clk::now() - t0).count() << std::endl;
    }
    afis_destroy(ctx);
    return ret;
}
