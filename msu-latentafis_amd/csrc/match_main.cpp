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
// matcher.cpp:306-309; with -tie 1|2 the list is std::sort's own, as the reference binary's — with several ranks on the gathered score column); the correspondence CSVs of the top 24, which the reference writes to the hard-coded
// /LatentAFIS/scores/corr<latent>_<rolled>_<i>.csv (matcher.cpp:325-327, :405, :497-505), go to <score dir>/corr<latent>_<rolled>_<i>.csv
// (or to the prefix given with -corr).
// Multi-GPU: started once per GPU with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment
// (e.g. `python -m torch.distributed.run --no-python --nproc-per-node 8 ./match ...`), every rank loads one contiguous shard of
// the gallery (balanced by texture points when -g is a container, by template count for a directory), scores all latents against
// it, and ONE RCCL all-gather per batch brings the per-shard top-24 lists (-l) or score columns (-ldir) together; rank 0 merges
// (score descending, index ascending) and writes exactly the files a single process writes (rank_exchange.h).
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
#include <thread>
#include <atomic>
#include <vector>

#include "../../include/afis_matcher.h"
#include "cli_util.h"
#include "rank_exchange.h"
#include "template_io.h"

namespace fs = std::filesystem;
using namespace afis;

namespace {

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

// Wall time of the job's stages, printed to stderr when AFIS_MATCH_TIMING is set (the reference prints only the total, matcher.cpp:209, :333):
// scan = directory / container listing, load = reading + parsing the rolled templates, commit = SoA packing + upload to HBM,
// latents = reading + parsing the latent files, search = afis_search (+ afis_correspondences), exchange = the multi-rank exchange step, write = the CSV files.
struct StageClock {
    double scan = 0, load = 0, commit = 0, latents = 0, search = 0, exchange = 0, write = 0;
    double device = 0, dev_bound = 0, dev_minu = 0;     // of `search`: the device's own clock over the searches' launch groups (afis_timing.total_ms), its bound pass and its minutiae stage
    void add_device(const afis_ctx* ctx)
    {
        if (!getenv("AFIS_MATCH_TIMING")) return;
        afis_timing t; memset(&t, 0, sizeof(t));
        if (afis_get_timing2(ctx, &t, sizeof(t)) == AFIS_OK) {
            device += t.total_ms; dev_bound += t.adc_bound_ms; dev_minu += t.minu_ms;
            if (getenv("AFIS_MATCH_TIMING")[0] == '2')
                fprintf(stderr, "match: search call on the device's clock (ms): total %.1f  bound %.1f  refine %.1f  texture_lists %.1f  candidates %.1f  minutiae_lists %.1f  (minutiae stage %.1f)\n",
                        t.total_ms, t.adc_bound_ms, t.adc_refine_ms, t.tex_tail_ms, t.cands_ms, t.minu_graph_ms, t.minu_ms);
        }
    }
    static double now() { return std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now().time_since_epoch()).count(); }
    void report(int rank, double total) const
    {
        if (!getenv("AFIS_MATCH_TIMING")) return;
        fprintf(stderr, "match[rank %d] timing (ms): scan %.1f  load %.1f  commit %.1f  latents %.1f  search %.1f  exchange %.1f  write %.1f  total %.1f\n",
                rank, scan, load, commit, latents, search, exchange, write, total);
        fprintf(stderr, "match[rank %d] of search, on the device's clock (ms): groups %.1f  bound_pass %.1f  minutiae_stage %.1f\n", rank, device, dev_bound, dev_minu);
    }
};
StageClock g_clock;

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

// loads templates [lo, hi) of the gallery and commits them with global indices
int load_gallery(afis_ctx* ctx, const std::string& g, const std::vector<fs::path>& files, const std::string& pack_to, int64_t lo = 0, int64_t hi = -1)
{
    if (hi < 0) hi = (int64_t)files.size();
    const double t_load = StageClock::now();
    if (fs::is_regular_file(fs::path(g))) {
        CHECK(ctx, afis_gallery_load(ctx, g.c_str(), lo, hi - lo));
    } else {
        // The reference re-reads every rolled file for every pair (matcher.cpp:173, :278); here each file is read and parsed ONCE, in slices of 8192 files:
        // the reads are spread over the host's threads, the parsing is afis_gallery_add_dat_batch's (also threaded), the order is the listing's.
        // While one slice is parsed and appended, the next one is read.
        const int64_t slice = 8192;
        const unsigned n_thr = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        typedef std::vector<std::vector<uint8_t>> Bufs;
        Bufs cur, nxt;
        auto read_slice = [&](int64_t s0, Bufs& bufs) {
            const int64_t n = std::min(slice, hi - s0);
            bufs.assign((size_t)n, {});
            std::vector<std::thread> th;
            for (unsigned t = 0; t < n_thr; ++t)
                th.emplace_back([&, t]() { for (int64_t i = t; i < n; i += n_thr) read_file(files[(size_t)(s0 + i)].string(), bufs[(size_t)i]); });
            for (std::thread& x : th) x.join();
        };
        std::vector<const void*> ptrs; std::vector<size_t> lens; std::vector<int> rcs;
        if (lo < hi) read_slice(lo, cur);
        for (int64_t s0 = lo; s0 < hi; s0 += slice) {
            const int64_t n = std::min(slice, hi - s0);
            struct Joined { std::thread t; ~Joined() { if (t.joinable()) t.join(); } } ahead;     // joined on every way out of the iteration
            if (s0 + slice < hi) ahead.t = std::thread(read_slice, s0 + slice, std::ref(nxt));
            ptrs.resize((size_t)n); lens.resize((size_t)n); rcs.assign((size_t)n, 0);
            for (int64_t i = 0; i < n; ++i) { ptrs[(size_t)i] = cur[(size_t)i].data(); lens[(size_t)i] = cur[(size_t)i].size(); }
            CHECK(ctx, afis_gallery_add_dat_batch(ctx, ptrs.data(), lens.data(), n, rcs.data()));
            if (s0 == lo) CHECK(ctx, afis_gallery_reserve(ctx, hi - lo));          // the first slice says how large a template is: room for the rest, once
            for (int64_t i = 0; i < n; ++i)
                if (rcs[(size_t)i] == 8) fprintf(stderr, "warning: %s: descriptor length outside 1..192, template discarded (scores -1)\n", files[(size_t)(s0 + i)].string().c_str());
            if (ahead.t.joinable()) ahead.t.join();
            cur.swap(nxt);
        }
    }
    if (!pack_to.empty()) {
        std::vector<std::string> names; std::vector<const char*> np;
        for (const fs::path& f : files) names.push_back(f.string());
        for (const std::string& n : names) np.push_back(n.c_str());
        CHECK(ctx, afis_gallery_save(ctx, pack_to.c_str(), np.data()));
    }
    const double t_commit = StageClock::now();
    g_clock.load += t_commit - t_load;
    CHECK(ctx, afis_gallery_commit(ctx, lo));
    g_clock.commit += StageClock::now() - t_commit;
    return 0;
}

// the ranks of one job: identical gallery listing on every rank (checked), shard bounds, the exchange
struct Job {
    RankWorld w;
    bool multi = false;
    std::vector<std::pair<int64_t, int64_t>> bounds{{0, 0}};
    int64_t lo = 0, hi = 0, g_max = 0;
    bool root() const { return w.rank == 0; }
    // Agreement point: every rank reports its local status; all of them get the first failure (in rank order) and leave together.
    // Without it a rank that fails (say, a corrupt rolled .dat in ITS shard) exits while the others wait in the next collective.
    int agree(int code)
    {
        if (!multi) return code;
        std::string e;
        const int r = world_agree(w, code, e);
        if (r == -1000) { std::cerr << "match[rank " << w.rank << "]: " << e << std::endl; return 2; }
        if (r != 0 && code == 0) std::cerr << "match[rank " << w.rank << "]: stopping, another rank failed (" << r << ")" << std::endl;
        return r;
    }
    int owner(int64_t idx) const { for (int r = 0; r < (int)bounds.size(); ++r) if (idx >= bounds[(size_t)r].first && idx < bounds[(size_t)r].second) return r; return -1; }
};

#define JOBCHK(call) do { std::string e_; if (!(call)) { std::cerr << "match[rank " << job.w.rank << "]: " << e_ << std::endl; return 2; } } while (0)

int plan_shards(Job& job, const std::string& gallery_path, const std::vector<fs::path>& rolled)
{
    const int64_t G = (int64_t)rolled.size();
    std::vector<int32_t> weights;
    if (job.multi && fs::is_regular_file(fs::path(gallery_path))) {          // balance by texture points, the cost driver
        weights.resize((size_t)G);
        int64_t g2 = 0;
        if (afis_gallery_file_info(gallery_path.c_str(), &g2, nullptr, nullptr, weights.data()) != AFIS_OK || g2 != G) { std::cerr << "match: " << afis_last_error(nullptr) << std::endl; return 2; }
    }
    job.bounds = shard_bounds(G, weights, job.w.world);
    job.lo = job.bounds[(size_t)job.w.rank].first; job.hi = job.bounds[(size_t)job.w.rank].second;
    job.g_max = 0;
    for (const auto& b : job.bounds) job.g_max = std::max(job.g_max, b.second - b.first);
    return 0;
}

// every rank must see the same gallery in the same order
int check_same_gallery(Job& job, const std::vector<fs::path>& rolled)
{
    if (!job.multi) return 0;
    uint64_t h[2] = {(uint64_t)rolled.size(), 1469598103934665603ull};
    for (const fs::path& p : rolled) for (char c : p.string()) { h[1] ^= (unsigned char)c; h[1] *= 1099511628211ull; }
    std::vector<uint64_t> all((size_t)job.w.world * 2);
    JOBCHK(world_all_gather(job.w, h, all.data(), sizeof(h), e_));
    for (int r = 0; r < job.w.world; ++r)
        if (all[(size_t)r * 2] != h[0] || all[(size_t)r * 2 + 1] != h[1]) { std::cerr << "match[rank " << job.w.rank << "]: rank " << r << " lists a different gallery" << std::endl; return 2; }
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    ArgParser args(argc, argv);
    if (args.cmdOptionExists("-h") || args.cmdOptionExists("--help")) {
        std::cout << "usage: match -l <latent.dat> | -ldir <latent dir>  -g <gallery dir>  -s <score dir>/  -c <codebook.dat>  [-d <device>] [-corr <prefix>] [-pack <gallery container to write>] [-tie <0|1|2>]\n       -tie: the order of equal sort keys — 0 ascending index (default), 1 candidate norms as libstdc++'s std::sort leaves them, 2 the scores of the greedy selections as well (the reference binary's scores)\n       -g may name a packed gallery container instead of a directory\n";
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
    Job job;
    world_from_env(job.w);
    job.multi = job.w.world > 1 || getenv("AFIS_FORCE_EXCHANGE") != nullptr;   // AFIS_FORCE_EXCHANGE: run the exchange path (RCCL communicator of one rank) in a single-process job
    if (!job.root()) std::cout.rdbuf(nullptr);                                  // rank 0 speaks for the job; errors still go to stderr
    const int device = args.cmdOptionExists("-d") ? atoi(args.getCmdOption("-d").c_str()) : (job.w.world > 1 ? job.w.local_rank : 0);

    std::vector<uint8_t> cb;
    if (!read_file(codebook_path, cb) || cb.empty()) { std::cout << "codebook is empty!" << std::endl; return 2; }
    afis_ctx* ctx = nullptr;
    if (int rc = afis_create_from_codebook(&ctx, cb.data(), cb.size(), device); rc != AFIS_OK) {
        std::cerr << "match: afis_create failed (" << rc << "): " << afis_last_error(nullptr) << std::endl;
        return 2;
    }
    const int tie_level = args.cmdOptionExists("-tie") ? atoi(args.getCmdOption("-tie").c_str()) : 0;
    if (args.cmdOptionExists("-tie")) {                                          // option ref_tie_order (include/afis_matcher.h): 2 = equal keys in the order the reference binary's std::sort leaves them
        if (int rc = afis_set_option(ctx, "ref_tie_order", tie_level); rc != AFIS_OK) {
            std::cerr << "match: -tie: " << afis_last_error(ctx) << std::endl;
            afis_destroy(ctx);
            return 2;
        }
    }

    if (job.multi) {
        std::string err;
        if (!world_init(job.w, device, err)) { std::cerr << "match[rank " << job.w.rank << "]: " << err << std::endl; afis_destroy(ctx); return 2; }
    }
    auto finish = [&](int code) { if (job.multi) world_finalize(job.w); afis_destroy(ctx); return code; };
    auto api = [&](int rc, const char* what) -> int {                            // C-ABI return code -> process status (reported, not yet acted on)
        if (rc == AFIS_OK) return 0;
        std::cerr << "match[rank " << job.w.rank << "]: " << what << " failed (" << rc << "): " << afis_last_error(ctx) << std::endl;
        return 2;
    };
    auto xchg = [&](const void* send, void* recv, size_t bytes) -> bool {        // the exchange step; a failure here ends the job on this rank
        std::string e;
        if (world_all_gather(job.w, send, recv, bytes, e)) return true;
        std::cerr << "match[rank " << job.w.rank << "]: " << e << std::endl;
        return false;
    };
    using clk = std::chrono::high_resolution_clock;
    int ret = 0;
    const std::string pack_to = args.cmdOptionExists("-pack") ? args.getCmdOption("-pack") : "";
    if (!pack_to.empty() && job.w.world > 1) { std::cerr << "match: -pack is a single-process operation" << std::endl; return finish(2); }
    if (!pack_to.empty() && !args.cmdOptionExists("-l") && !args.cmdOptionExists("-ldir")) {          // pack only
        std::vector<fs::path> rolled = list_gallery(gallery_path);
        if (rolled.empty()) { std::cout << "No rolled templates found in directory: " << gallery_path << std::endl; return finish(-1); }
        ret = load_gallery(ctx, gallery_path, rolled, pack_to);
        if (ret == 0) std::cout << "Packed " << rolled.size() << " templates into " << pack_to << std::endl;
        return finish(ret);
    }
    if (args.cmdOptionExists("-l")) {
        // ---- One2List_matching, matcher.cpp:216-337 ----
        const fs::path latent_file(args.getCmdOption("-l"));
        const std::string score_file = score_path + latent_file.stem().string() + ".csv";
        const double t_scan = StageClock::now();
        std::vector<fs::path> rolled = list_gallery(gallery_path);
        g_clock.scan += StageClock::now() - t_scan;
        if (rolled.empty()) { std::cout << "No rolled templates found in directory: " << gallery_path << std::endl; return finish(-1); }
        const auto t0 = clk::now();
        std::cout << "Latent Query: " << latent_file << std::endl;
        std::cout << "Gallery size: " << rolled.size() << std::endl;
        if ((ret = job.agree(plan_shards(job, gallery_path, rolled))) != 0) return finish(ret);
        if ((ret = check_same_gallery(job, rolled)) != 0) return finish(ret);
        if ((ret = job.agree(load_gallery(ctx, gallery_path, rolled, pack_to, job.lo, job.hi))) != 0) return finish(ret);
        double t_s = StageClock::now();
        Latent L; L.load(latent_file);
        g_clock.latents += StageClock::now() - t_s;
        if (job.root() && L.view.n_minu <= 0 && L.view.n_tex <= 0) { std::ofstream out(score_file); out << 0 << std::endl; }       // :260-268
        const int k = (int)std::min<size_t>(24, rolled.size());
        constexpr int kk = 24;                                                   // fixed-size per-rank block of the exchange
        std::vector<int64_t> idx(kk); std::vector<float> sc(kk); int32_t status = 0;
        t_s = StageClock::now();
        // -tie >= 1: the rank list as the reference binary makes it — libstdc++'s std::sort of the gallery indices on the non-strict score comparator (matcher.cpp:306-309), run
        // (afis_rank_list) on the score column itself: equal scores (the zero scores of a small gallery's tail) come out in ITS order, not by ascending index.  Several ranks:
        // the shards' score columns are gathered (the exchange of -ldir, one latent) and every rank sorts the same whole column.
        const bool ref_rank_order = tie_level >= 1;
        const size_t G = rolled.size(), Gl = (size_t)(job.hi - job.lo), Gm = (size_t)job.g_max;
        std::vector<float> column(ref_rank_order ? (job.multi ? std::max<size_t>(Gl, 1) : G) : 0);
        if ((ret = job.agree(api(afis_search(ctx, &L.view, 1, ref_rank_order ? column.data() : nullptr, nullptr, &status, kk, idx.data(), sc.data()), "afis_search"))) != 0) return finish(ret);   // padded with -1 beyond the shard
        if (ref_rank_order && status != AFIS_QUERY_LATENT_EMPTY) {
            if (job.multi) {
                std::vector<float> block(std::max<size_t>(Gm, 1), -1.0f), all((size_t)job.w.world * block.size()), whole(G);
                memcpy(block.data(), column.data(), Gl * sizeof(float));
                if (!xchg(block.data(), all.data(), block.size() * sizeof(float))) return finish(2);
                for (int r = 0; r < job.w.world; ++r) {
                    const size_t lo = (size_t)job.bounds[(size_t)r].first, n = (size_t)(job.bounds[(size_t)r].second - job.bounds[(size_t)r].first);
                    memcpy(&whole[lo], &all[(size_t)r * block.size()], n * sizeof(float));
                }
                column.swap(whole);
            }
            if ((ret = api(afis_rank_list(column.data(), (int64_t)column.size(), 1, k, idx.data(), sc.data()), "afis_rank_list")) != 0) return finish(ret);
        }
        g_clock.search += StageClock::now() - t_s;
        if (status == AFIS_QUERY_LATENT_EMPTY) { std::cout << "Matching failed: latent template is empty. Exiting." << std::endl; return finish(1); }
        if (job.multi && !ref_rank_order) {                                      // the exchange step: per-shard top-24 -> merged top-24
            std::vector<int64_t> all_i((size_t)job.w.world * kk); std::vector<float> all_s((size_t)job.w.world * kk);
            if (!xchg(idx.data(), all_i.data(), kk * sizeof(int64_t)) || !xchg(sc.data(), all_s.data(), kk * sizeof(float))) return finish(2);
            merge_topk(all_i, all_s, job.w.world, kk, kk, idx, sc);
        }
        // correspondence files for the top 24 (matcher.cpp:311-328): one "lx,ly,rx,ry" line per surviving correspondence;
        // every rank exports the pairs of its own shard
        constexpr size_t kXY = 3 * 120 * 4;
        std::vector<int32_t> counts((size_t)kk * 3, -1); std::vector<int16_t> xy((size_t)kk * kXY, 0);
        {
            std::vector<int64_t> mine; std::vector<int> pos;
            for (int j = 0; j < k; ++j) if (idx[j] >= job.lo && idx[j] < job.hi) { mine.push_back(idx[j]); pos.push_back(j); }
            std::vector<int32_t> c((size_t)mine.size() * 3 + 1); std::vector<int16_t> v((size_t)mine.size() * kXY + 1);
            t_s = StageClock::now();
            if ((ret = job.agree(api(afis_correspondences(ctx, &L.view, mine.data(), (int)mine.size(), c.data(), v.data()), "afis_correspondences"))) != 0) return finish(ret);
            g_clock.search += StageClock::now() - t_s;
            for (size_t a = 0; a < mine.size(); ++a) {
                memcpy(&counts[(size_t)pos[a] * 3], &c[a * 3], 3 * sizeof(int32_t));
                memcpy(&xy[(size_t)pos[a] * kXY], &v[a * kXY], kXY * sizeof(int16_t));
            }
        }
        if (job.multi) {
            std::vector<int32_t> all_c((size_t)job.w.world * counts.size()); std::vector<int16_t> all_v((size_t)job.w.world * xy.size());
            if (!xchg(counts.data(), all_c.data(), counts.size() * sizeof(int32_t)) || !xchg(xy.data(), all_v.data(), xy.size() * sizeof(int16_t))) return finish(2);
            for (int j = 0; j < k; ++j) {
                const int r = job.owner(idx[j]);
                if (r < 0) continue;
                memcpy(&counts[(size_t)j * 3], &all_c[(size_t)r * counts.size() + (size_t)j * 3], 3 * sizeof(int32_t));
                memcpy(&xy[(size_t)j * kXY], &all_v[(size_t)r * xy.size() + (size_t)j * kXY], kXY * sizeof(int16_t));
            }
        }
        if (!job.root()) return finish(0);
        t_s = StageClock::now();
        std::ofstream out(score_file);
        out << "filename,score" << std::endl;
        std::cout << "Match Results" << std::endl << "----------------" << std::endl << "Rank     Filename      Score" << std::endl;
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
        g_clock.write += StageClock::now() - t_s;
        const double total_l = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        std::cout << "Total matching duration (ms): " << total_l << std::endl;
        g_clock.report(job.w.rank, total_l + g_clock.scan);
    } else {
        // ---- List2List_matching, matcher.cpp:96-214 ----
        std::string latent_dir;
        if (args.cmdOptionExists("-ldir")) latent_dir = args.getCmdOption("-ldir");
        else {
            std::cout << "Missing argument for latent template or directory. Assuming batch matching, using default directory from afis.config" << std::endl;
            if (!from_config("LatentTemplateDirectory", latent_dir)) return finish(2);
        }
        const double t_scan = StageClock::now();
        std::vector<fs::path> latents = list_dat(latent_dir);
        for (const fs::path& p : latents) std::cout << "latent template file" << p << std::endl;
        if (latents.empty()) { std::cout << "No latent templates found in directory: " << latent_dir << std::endl; return finish(-1); }
        std::vector<fs::path> rolled = list_gallery(gallery_path);
        {   // one "rolled template file<path>" line per gallery file, as the reference prints them (matcher.cpp:127): a million lines go out in one piece
            std::string lines;
            for (const fs::path& p : rolled) { std::ostringstream q; q << p; lines += "rolled template file"; lines += q.str(); lines += '\n'; }
            std::cout << lines << std::flush;
        }
        g_clock.scan += StageClock::now() - t_scan;
        if (rolled.empty()) { std::cout << "No rolled templates found in directory: " << gallery_path << std::endl; return finish(-1); }
        std::cout << "Gallery size: " << rolled.size() << std::endl;
        const auto t0 = clk::now();
        if ((ret = job.agree(plan_shards(job, gallery_path, rolled))) != 0) return finish(ret);
        if ((ret = check_same_gallery(job, rolled)) != 0) return finish(ret);
        if ((ret = job.agree(load_gallery(ctx, gallery_path, rolled, pack_to, job.lo, job.hi))) != 0) return finish(ret);
        const size_t G = rolled.size(), Gl = (size_t)(job.hi - job.lo), Gm = (size_t)job.g_max;
        const size_t batch = 16;
        std::vector<std::string> quoted;                                         // the gallery paths as `operator<<(ostream&, path)` prints them (quoted, escaped) + ","
        struct Joined { std::thread t; ~Joined() { if (t.joinable()) t.join(); } } writer_;   // formats and writes the previous batch's score files (declared after what it reads)
        std::thread& writer = writer_.t;
        for (size_t i0 = 0; i0 < latents.size(); i0 += batch) {
            const size_t nb = std::min(batch, latents.size() - i0);
            double t_s = StageClock::now();
            std::vector<Latent> Ls(nb); std::vector<afis_template_view> views(nb);
            for (size_t i = 0; i < nb; ++i) { Ls[i].load(latents[i0 + i]); views[i] = Ls[i].view; }
            g_clock.latents += StageClock::now() - t_s;
            std::vector<float> scores(nb * G); std::vector<int32_t> status(nb);
            t_s = StageClock::now();
            if (!job.multi) {
                if ((ret = api(afis_search(ctx, views.data(), (int)nb, scores.data(), nullptr, status.data(), 0, nullptr, nullptr), "afis_search")) != 0) return finish(ret);
                g_clock.search += StageClock::now() - t_s; g_clock.add_device(ctx);
            } else {                                                             // the exchange step: score columns of every shard
                std::vector<float> part(nb * std::max<size_t>(Gl, 1)), block(nb * std::max<size_t>(Gm, 1), -1.0f), all((size_t)job.w.world * block.size());
                if ((ret = job.agree(api(afis_search(ctx, views.data(), (int)nb, part.data(), nullptr, status.data(), 0, nullptr, nullptr), "afis_search"))) != 0) return finish(ret);
                g_clock.search += StageClock::now() - t_s; g_clock.add_device(ctx);
                t_s = StageClock::now();
                for (size_t i = 0; i < nb; ++i) memcpy(&block[i * Gm], &part[i * Gl], Gl * sizeof(float));
                if (!xchg(block.data(), all.data(), block.size() * sizeof(float))) return finish(2);
                for (int r = 0; r < job.w.world; ++r) {
                    const size_t lo = (size_t)job.bounds[(size_t)r].first, n = (size_t)(job.bounds[(size_t)r].second - job.bounds[(size_t)r].first);
                    for (size_t i = 0; i < nb; ++i) memcpy(&scores[i * G + lo], &all[(size_t)r * block.size() + i * Gm], n * sizeof(float));
                }
                g_clock.exchange += StageClock::now() - t_s;
                if (!job.root()) continue;
            }
            t_s = StageClock::now();
            if (quoted.empty()) { quoted.reserve(G); for (size_t j = 0; j < G; ++j) { std::ostringstream q; q << rolled[j]; quoted.push_back(q.str() + ","); } }
            std::vector<std::pair<std::string, size_t>> files;                   // (csv path, row of `scores`) of this batch's score files
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
                files.emplace_back(csv, i);
            }
            // :201-204: one `"<path>",<score %.3f>` line per gallery file.  The same bytes as `out << path << "," << setprecision(3) << fixed << score << endl`,
            // formatted into one buffer per file and written once (endl flushes every line: 10^5 - 10^6 write calls per latent).  The files of a batch are
            // independent: they are formatted and written by a few threads while the next batch is searched; the previous batch's writer is joined first.
            if (writer.joinable()) writer.join();
            writer = std::thread([files = std::move(files), scores = std::move(scores), &quoted, G]() {
                std::atomic<size_t> next{0};
                auto work = [&]() {
                    char num[64];
                    for (size_t f = next.fetch_add(1); f < files.size(); f = next.fetch_add(1)) {
                        const float* row = &scores[files[f].second * G];
                        std::string buf;
                        buf.reserve(G * (quoted.empty() ? 16 : quoted[0].size() + 12));
                        for (size_t j = 0; j < G; ++j) { buf += quoted[j]; const int n = snprintf(num, sizeof(num), "%.3f\n", (double)row[j]); buf.append(num, (size_t)n); }
                        std::ofstream out(files[f].first, std::ios::binary);
                        out.write(buf.data(), (std::streamsize)buf.size());
                    }
                };
                const size_t n_thr = std::min<size_t>(files.size(), std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
                std::vector<std::thread> th;
                for (size_t t = 1; t < n_thr; ++t) th.emplace_back(work);
                work();
                for (std::thread& x : th) x.join();
            });
            g_clock.write += StageClock::now() - t_s;
        }
        const double t_w = StageClock::now();
        if (writer.joinable()) writer.join();
        g_clock.write += StageClock::now() - t_w;
        const double total_d = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        std::cout << "Total matching duration (ms): " << total_d << std::endl;
        g_clock.report(job.w.rank, total_d + g_clock.scan);
    }
    return finish(ret);
}
