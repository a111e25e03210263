#include "template_io.h"

#include <algorithm>
#include <thread>
#include <atomic>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <type_traits>

#include <cstdio>
#include <cstring>

namespace afis {
namespace {

constexpr int kMaxMinutiae = 2000;   // matcher.cpp:788
constexpr int kMaxDesLength = 192;   // matcher.cpp:789
constexpr int kBadDesLength = 8;     // parser return code with no reference counterpart: a descriptor length outside 1..192
constexpr int kMaxBlkSize = 100;     // matcher.cpp:790

// std::ifstream semantics: a read past the end delivers what is left and every later read delivers nothing.
struct Cursor {
    const uint8_t* p; size_t len; size_t pos = 0; bool fail = false;
    void read(void* dst, size_t n)
    {
        if (fail) return;
        const size_t avail = len - pos;
        if (n > avail) { memcpy(dst, p + pos, avail); pos = len; fail = true; return; }
        memcpy(dst, p + pos, n); pos += n;
    }
    template <class T> T get() { T v{}; read(&v, sizeof(T)); return v; }
};

template <class Tex>
int parse_common(Cursor& c, HostTemplate& out, bool rolled)
{
    int16_t header[12]; c.read(header, sizeof(header));
    out.h = c.get<int16_t>(); out.w = c.get<int16_t>();
    int blkH = c.get<int16_t>(), blkW = c.get<int16_t>();
    int n_minu_tpl = c.get<uint8_t>();
    if (blkH > 50) blkH = 50;
    if (blkW > 50) blkW = 50;
    out.blkH = blkH; out.blkW = blkW;
    if (c.fail) n_minu_tpl = 0;
    for (int i = 0; i < n_minu_tpl; ++i) {
        const int n = c.get<int16_t>();
        if (c.fail) break;
        if (n <= 0) continue;                                  // dropped: later template indices shift (:835-836)
        if (n > kMaxMinutiae) return 2;
        if (blkH > kMaxBlkSize || blkW > kMaxBlkSize) return 4;
        HostMinutiae m;
        m.x.resize(n); m.y.resize(n); m.ori.resize(n);
        c.read(m.x.data(), 2 * (size_t)n); c.read(m.y.data(), 2 * (size_t)n); c.read(m.ori.data(), 4 * (size_t)n);
        const int dl = c.get<int16_t>();
        if (c.fail) break;
        if (dl <= 0 || dl > kMaxDesLength) return kBadDesLength;   // the reference has no check here (it overruns a stack buffer): stop, keep what was parsed
        m.des_len = dl; m.des.assign((size_t)n * dl, 0.f);
        c.read(m.des.data(), 4 * (size_t)n * dl);
        out.minu.push_back(std::move(m));
    }
    int n_tex_tpl = c.get<uint8_t>();
    if (c.fail) n_tex_tpl = 0;
    for (int i = 0; i < n_tex_tpl; ++i) {
        const int n = c.get<int16_t>();
        if (c.fail) break;
        if (n <= 0) continue;
        if (n > kMaxMinutiae) return -1;
        HostTexture t;
        t.x.resize(n); t.y.resize(n); t.ori.resize(n);
        c.read(t.x.data(), 2 * (size_t)n); c.read(t.y.data(), 2 * (size_t)n); c.read(t.ori.data(), 4 * (size_t)n);
        const int dl = c.get<int16_t>();
        if (c.fail) break;
        if (dl <= 0 || dl > kMaxDesLength) return kBadDesLength;   // as above: never parse texture templates from a misaligned cursor
        t.des_len = dl;
        if (rolled) {
            // the reference reads n*des_len floats here (a 4x over-read that runs into EOF, :975) and keeps the
            // first n*des_len bytes as PQ codes (include.h:401-406)
            std::vector<uint8_t> raw((size_t)n * dl * 4, 0);
            c.read(raw.data(), raw.size());
            t.codes.assign(raw.begin(), raw.begin() + (size_t)n * dl);
        } else {
            t.des.assign((size_t)n * dl, 0.f);
            c.read(t.des.data(), 4 * (size_t)n * dl);
        }
        out.tex.push_back(std::move(t));
    }
    return 0;
}

void put(std::vector<uint8_t>& o, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; o.insert(o.end(), b, b + n); }
template <class T> void put(std::vector<uint8_t>& o, T v) { put(o, &v, sizeof(T)); }

void write_head(std::vector<uint8_t>& o, const HostTemplate& t)
{
    uint16_t header[12] = {1};                                 // [0] = template version
    put(o, header, sizeof(header));
    if (t.minu.empty()) { uint16_t z[4] = {0, 0, 0, 0}; put(o, z, sizeof(z)); return; }
    put<uint16_t>(o, (uint16_t)t.h); put<uint16_t>(o, (uint16_t)t.w);
    put<uint16_t>(o, (uint16_t)(t.blkH > 50 ? 50 : t.blkH)); put<uint16_t>(o, (uint16_t)(t.blkW > 50 ? 50 : t.blkW));
    put<uint8_t>(o, (uint8_t)t.minu.size());
    for (const HostMinutiae& m : t.minu) {
        const int n = m.n() > kMaxMinutiae ? kMaxMinutiae : m.n();
        put<uint16_t>(o, (uint16_t)n);
        if (n <= 0) continue;
        put(o, m.x.data(), 2 * (size_t)n); put(o, m.y.data(), 2 * (size_t)n); put(o, m.ori.data(), 4 * (size_t)n);
        put<uint16_t>(o, (uint16_t)m.des_len);
        put(o, m.des.data(), 4 * (size_t)n * m.des_len);
    }
}

}  // namespace

int parse_latent_dat(const void* bytes, size_t len, HostTemplate& out)
{
    out = HostTemplate();
    if (len == 0) return 1;                                    // :798-801
    Cursor c{(const uint8_t*)bytes, len};
    return parse_common<HostTexture>(c, out, false);
}

int parse_rolled_dat(const void* bytes, size_t len, HostTemplate& out)
{
    out = HostTemplate();
    if (len <= 10) return 1;                                   // :899-902
    Cursor c{(const uint8_t*)bytes, len};
    return parse_common<HostTexture>(c, out, true);
}

bool parse_codebook(const void* bytes, size_t len, HostCodebook& out)
{
    Cursor c{(const uint8_t*)bytes, len};
    out.M = c.get<int16_t>(); out.K = c.get<int16_t>(); out.dsub = c.get<int16_t>();
    const long n = (long)out.M * out.K * out.dsub;
    if (c.fail || n <= 0) return false;
    out.words.assign(n, 0.f);
    c.read(out.words.data(), 4 * (size_t)n);
    return !c.fail;
}

std::vector<uint8_t> write_latent_dat(const HostTemplate& t)
{
    std::vector<uint8_t> o;
    write_head(o, t);
    if (t.minu.empty()) return o;
    put<uint8_t>(o, (uint8_t)t.tex.size());
    for (const HostTexture& x : t.tex) {
        const int n = x.n() > kMaxMinutiae ? kMaxMinutiae : x.n();
        put<uint16_t>(o, (uint16_t)n);
        if (n <= 0) continue;
        put(o, x.x.data(), 2 * (size_t)n); put(o, x.y.data(), 2 * (size_t)n); put(o, x.ori.data(), 4 * (size_t)n);
        put<uint16_t>(o, (uint16_t)x.des_len);
        put(o, x.des.data(), 4 * (size_t)n * x.des_len);
    }
    return o;
}

std::vector<uint8_t> write_rolled_dat(const HostTemplate& t)
{
    std::vector<uint8_t> o;
    write_head(o, t);
    if (t.minu.empty()) return o;
    put<uint8_t>(o, (uint8_t)t.tex.size());
    for (const HostTexture& x : t.tex) {
        const int n = x.n() > kMaxMinutiae ? kMaxMinutiae : x.n();
        put<uint16_t>(o, (uint16_t)n);
        if (n <= 0) continue;
        put(o, x.x.data(), 2 * (size_t)n); put(o, x.y.data(), 2 * (size_t)n); put(o, x.ori.data(), 4 * (size_t)n);
        put<uint16_t>(o, (uint16_t)x.des_len);
        put(o, x.codes.data(), (size_t)n * x.des_len);
    }
    return o;
}

bool read_file(const std::string& path, std::vector<uint8_t>& out)
{
    out.clear();
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n > 0) { out.resize((size_t)n); if (fread(out.data(), 1, (size_t)n, f) != (size_t)n) { fclose(f); out.clear(); return false; } }
    fclose(f);
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// packed gallery container (layout in template_io.h)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr char kGalMagic[8] = {'A', 'F', 'I', 'S', 'G', 'A', 'L', '1'};
constexpr int kGalSections = 13;
struct GalHeader {
    char magic[8];
    uint32_t version, des_len, code_len, reserved;
    int64_t G, n_minu, n_tex, names_bytes;
    uint64_t off[kGalSections];
};
static_assert(sizeof(GalHeader) == 56 + 8 * kGalSections, "container header layout");

struct Mapped {                       // read-only mmap of a whole file
    const uint8_t* p = nullptr; size_t len = 0; int fd = -1;
    bool open_file(const std::string& path, std::string& err)
    {
        fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) { err = "cannot open " + path; return false; }
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < (off_t)sizeof(GalHeader)) { err = path + ": not a gallery container (too small)"; return false; }
        len = (size_t)st.st_size;
        void* m = mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { err = "mmap failed for " + path; return false; }
        p = (const uint8_t*)m;
        return true;
    }
    ~Mapped() { if (p) munmap((void*)p, len); if (fd >= 0) ::close(fd); }
};

bool check_header(const Mapped& m, const std::string& path, GalHeader& h, size_t sizes[kGalSections], std::string& err)
{
    memcpy(&h, m.p, sizeof(h));
    if (memcmp(h.magic, kGalMagic, 8) != 0 || h.version != 1) { err = path + ": not an AFISGAL1 container"; return false; }
    if (h.des_len != 96 || h.code_len != 16 || h.G < 0 || h.n_minu < 0 || h.n_tex < 0 || h.names_bytes < 0) { err = path + ": unsupported container geometry"; return false; }
    const size_t G = (size_t)h.G, NM = (size_t)h.n_minu, NT = (size_t)h.n_tex;
    const size_t want[kGalSections] = {(G + 1) * 8, (G + 1) * 8, G, NM * 2, NM * 2, NM * 4, NM * 96 * 4, NT * 2, NT * 2, NT * 4, NT * 16, (G + 1) * 8, (size_t)h.names_bytes};
    for (int i = 0; i < kGalSections; ++i) {
        sizes[i] = want[i];
        if (h.off[i] % 64 != 0 || h.off[i] > m.len || want[i] > m.len - h.off[i]) { err = path + ": truncated or corrupt container"; return false; }
    }
    return true;
}
}  // namespace

bool gallery_append_template(HostGallery& g, const HostTemplate& t)
{
    // the staged arrays are fixed-width (96 floats / 16 code bytes per point): refuse anything else instead of reading past a vector
    if (!t.minu.empty() && (t.minu[0].des_len != 96 || t.minu[0].des.size() != (size_t)t.minu[0].n() * 96)) return false;
    if (!t.tex.empty() && (t.tex[0].des_len != 16 || t.tex[0].codes.size() < (size_t)t.tex[0].n() * 16)) return false;
    if (!t.minu.empty()) {
        const HostMinutiae& m = t.minu[0];
        g.mx.insert(g.mx.end(), m.x.begin(), m.x.end()); g.my.insert(g.my.end(), m.y.begin(), m.y.end());
        g.mori.insert(g.mori.end(), m.ori.begin(), m.ori.end()); g.mdes.insert(g.mdes.end(), m.des.begin(), m.des.end());
    }
    g.minu_off.push_back((int64_t)g.mx.size());
    if (!t.tex.empty()) {
        const HostTexture& x = t.tex[0];
        const int n = x.n() < 1000 ? x.n() : 1000;                           // matcher.cpp:546-547
        g.tx.insert(g.tx.end(), x.x.begin(), x.x.begin() + n); g.ty.insert(g.ty.end(), x.y.begin(), x.y.begin() + n);
        g.tori.insert(g.tori.end(), x.ori.begin(), x.ori.begin() + n); g.tcodes.insert(g.tcodes.end(), x.codes.begin(), x.codes.begin() + (size_t)n * 16);
    }
    g.tex_off.push_back((int64_t)g.tx.size());
    g.empty.push_back(t.minu.empty() && t.tex.empty() ? 1 : 0);
    return true;
}

bool write_gallery_container(const std::string& path, const HostGallery& g, const std::vector<std::string>& names, std::string& err)
{
    const size_t G = (size_t)g.size(), NM = g.mx.size(), NT = g.tx.size();
    if (!names.empty() && names.size() != G) { err = "write_gallery_container: names must be empty or one per template"; return false; }
    std::vector<int64_t> name_off(G + 1, 0);
    std::string blob;
    for (size_t i = 0; i < G; ++i) { if (!names.empty()) blob += names[i]; name_off[i + 1] = (int64_t)blob.size(); }
    GalHeader h = {};
    memcpy(h.magic, kGalMagic, 8);
    h.version = 1; h.des_len = 96; h.code_len = 16;
    h.G = (int64_t)G; h.n_minu = (int64_t)NM; h.n_tex = (int64_t)NT; h.names_bytes = (int64_t)blob.size();
    const void* ptr[kGalSections] = {g.minu_off.data(), g.tex_off.data(), g.empty.data(), g.mx.data(), g.my.data(), g.mori.data(), g.mdes.data(),
                                     g.tx.data(), g.ty.data(), g.tori.data(), g.tcodes.data(), name_off.data(), blob.data()};
    const size_t sz[kGalSections] = {(G + 1) * 8, (G + 1) * 8, G, NM * 2, NM * 2, NM * 4, NM * 96 * 4, NT * 2, NT * 2, NT * 4, NT * 16, (G + 1) * 8, blob.size()};
    if (g.minu_off.size() != G + 1 || g.tex_off.size() != G + 1 || g.mdes.size() != NM * 96 || g.tcodes.size() != NT * 16) { err = "write_gallery_container: inconsistent gallery"; return false; }
    uint64_t pos = (sizeof(GalHeader) + 63) / 64 * 64;
    for (int i = 0; i < kGalSections; ++i) { h.off[i] = pos; pos = (pos + sz[i] + 63) / 64 * 64; }
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) { err = "cannot create " + path; return false; }
    static const char zeros[64] = {0};
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1;
    uint64_t at = sizeof(h);
    for (int i = 0; i < kGalSections && ok; ++i) {
        if (h.off[i] > at) { ok = fwrite(zeros, 1, (size_t)(h.off[i] - at), f) == (size_t)(h.off[i] - at); at = h.off[i]; }
        if (sz[i] && ok) ok = fwrite(ptr[i], 1, sz[i], f) == sz[i];
        at += sz[i];
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) err = "write failed for " + path;
    return ok;
}

bool gallery_container_info(const std::string& path, GalleryFileInfo& info, std::string& err)
{
    Mapped m; GalHeader h; size_t sizes[kGalSections];
    if (!m.open_file(path, err) || !check_header(m, path, h, sizes, err)) return false;
    info.G = h.G; info.n_minu = h.n_minu; info.n_tex = h.n_tex;
    return true;
}

GalleryMapping::~GalleryMapping() { if (base_) munmap(base_, len_); if (fd_ >= 0) ::close(fd_); }

std::unique_ptr<GalleryMapping> map_gallery_container(const std::string& path, std::string& err)
{
    Mapped m; GalHeader h; size_t sizes[kGalSections];
    if (!m.open_file(path, err) || !check_header(m, path, h, sizes, err)) return nullptr;
    const int64_t* mo = (const int64_t*)(m.p + h.off[0]); const int64_t* to = (const int64_t*)(m.p + h.off[1]);
    if (mo[0] != 0 || to[0] != 0) { err = path + ": corrupt offsets"; return nullptr; }
    for (int64_t i = 0; i < h.G; ++i)
        if (mo[i + 1] < mo[i] || to[i + 1] < to[i] || mo[i + 1] > h.n_minu || to[i + 1] > h.n_tex) { err = path + ": corrupt offsets"; return nullptr; }
    std::unique_ptr<GalleryMapping> g(new GalleryMapping);
    g->G = h.G; g->n_minu = h.n_minu; g->n_tex = h.n_tex; g->path = path;
    g->minu_off = mo; g->tex_off = to; g->empty = m.p + h.off[2];
    g->mx = (const int16_t*)(m.p + h.off[3]); g->my = (const int16_t*)(m.p + h.off[4]); g->mori = (const float*)(m.p + h.off[5]); g->mdes = (const float*)(m.p + h.off[6]);
    g->tx = (const int16_t*)(m.p + h.off[7]); g->ty = (const int16_t*)(m.p + h.off[8]); g->tori = (const float*)(m.p + h.off[9]); g->tcodes = m.p + h.off[10];
    g->base_ = (void*)m.p; g->len_ = m.len; g->fd_ = m.fd;
    m.p = nullptr; m.fd = -1;                                              // the mapping now belongs to *g
    return g;
}

// The template range [first, first + count) of a container that is ALREADY mapped and validated (afis_gallery_load) copied into host arrays: the same bytes
// afis_gallery_commit would have uploaded, no second open of the path (a file replaced in between would be read without the per-template checks the load did).
// The mapping is private to the file as it was opened; a file truncated since then would fault on access, so the size is checked again first.
bool copy_from_mapping(const GalleryMapping& g, int64_t first, int64_t count, HostGallery& out, std::string& err)
{
    if (first < 0 || count < 0 || first + count > g.G) { err = g.path + ": template range outside the container"; return false; }
    struct stat st;
    if (g.fd_ >= 0 && (fstat(g.fd_, &st) != 0 || (size_t)st.st_size < g.len_)) { err = g.path + ": the container was truncated after it was loaded"; return false; }
    const int64_t m0 = g.minu_off[first], m1 = g.minu_off[first + count], t0 = g.tex_off[first], t1 = g.tex_off[first + count];
    struct Job { uint8_t* dst; const uint8_t* src; size_t bytes; };
    std::vector<Job> jobs;
    auto app = [&](auto& vec, const auto* src_arr, int64_t a, int64_t b, size_t per) {
        typedef typename std::remove_reference<decltype(vec)>::type V;
        typedef typename V::value_type T;
        const size_t old = vec.size(), n = (size_t)(b - a) * per;
        vec.resize(old + n);
        const uint8_t* src = (const uint8_t*)(src_arr + (size_t)a * per);
        uint8_t* dst = (uint8_t*)(vec.data() + old);
        const size_t chunk = (size_t)8 << 20;
        for (size_t o = 0; o < n * sizeof(T); o += chunk) jobs.push_back({dst + o, src + o, std::min(chunk, n * sizeof(T) - o)});
    };
    app(out.mx, g.mx, m0, m1, 1); app(out.my, g.my, m0, m1, 1); app(out.mori, g.mori, m0, m1, 1); app(out.mdes, g.mdes, m0, m1, 96);
    app(out.tx, g.tx, t0, t1, 1); app(out.ty, g.ty, t0, t1, 1); app(out.tori, g.tori, t0, t1, 1); app(out.tcodes, g.tcodes, t0, t1, 16);
    {
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t j = next.fetch_add(1); j < jobs.size(); j = next.fetch_add(1)) memcpy(jobs[j].dst, jobs[j].src, jobs[j].bytes); };
        const size_t n_thr = std::min<size_t>(std::max<size_t>(jobs.size(), 1), std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
        std::vector<std::thread> th;
        for (size_t t = 1; t < n_thr; ++t) th.emplace_back(work);
        work();
        for (std::thread& x : th) x.join();
    }
    const int64_t mb = out.minu_off.back() - m0, tb = out.tex_off.back() - t0;
    for (int64_t i = first; i < first + count; ++i) { out.minu_off.push_back(g.minu_off[i + 1] + mb); out.tex_off.push_back(g.tex_off[i + 1] + tb); out.empty.push_back(g.empty[i]); }
    return true;
}

bool read_gallery_container(const std::string& path, int64_t first, int64_t count, HostGallery& out, std::vector<std::string>* names,
                            std::vector<int32_t>* tex_counts, std::string& err, bool load_data)
{
    Mapped m; GalHeader h; size_t sizes[kGalSections];
    if (!m.open_file(path, err) || !check_header(m, path, h, sizes, err)) return false;
    if (count < 0) count = h.G - first;
    if (first < 0 || count < 0 || first + count > h.G) { err = path + ": template range outside the container"; return false; }
    const int64_t* mo = (const int64_t*)(m.p + h.off[0]); const int64_t* to = (const int64_t*)(m.p + h.off[1]);
    for (int64_t i = 0; i < h.G; ++i)
        if (mo[i + 1] < mo[i] || to[i + 1] < to[i] || mo[i + 1] > h.n_minu || to[i + 1] > h.n_tex || mo[0] != 0 || to[0] != 0) { err = path + ": corrupt offsets"; return false; }
    if (tex_counts) { tex_counts->resize((size_t)h.G); for (int64_t i = 0; i < h.G; ++i) (*tex_counts)[(size_t)i] = (int32_t)(to[i + 1] - to[i]); }
    const int64_t m0 = mo[first], m1 = mo[first + count], t0 = to[first], t1 = to[first + count];
    if (load_data) {                                                       // listing the names / counts of a 5 GB container must not copy its arrays
    // The arrays are grown without a zero-fill (BulkVec) and filled by a few threads: a shard of 100 000 templates is 5 GB, and one thread
    // faulting in and copying that much is a second per container (the page cache delivers several times that to parallel readers).
    struct Job { uint8_t* dst; const uint8_t* src; size_t bytes; };
    std::vector<Job> jobs;
    auto app = [&](auto& vec, int sec, int64_t a, int64_t b, size_t per) {
        typedef typename std::remove_reference<decltype(vec)>::type V;
        typedef typename V::value_type T;
        const size_t old = vec.size(), n = (size_t)(b - a) * per;
        vec.resize(old + n);
        const uint8_t* src = m.p + h.off[sec] + (size_t)a * per * sizeof(T);
        uint8_t* dst = (uint8_t*)(vec.data() + old);
        const size_t chunk = (size_t)8 << 20;
        for (size_t o = 0; o < n * sizeof(T); o += chunk) jobs.push_back({dst + o, src + o, std::min(chunk, n * sizeof(T) - o)});
    };
    app(out.mx, 3, m0, m1, 1); app(out.my, 4, m0, m1, 1); app(out.mori, 5, m0, m1, 1); app(out.mdes, 6, m0, m1, 96);
    app(out.tx, 7, t0, t1, 1); app(out.ty, 8, t0, t1, 1); app(out.tori, 9, t0, t1, 1); app(out.tcodes, 10, t0, t1, 16);
    {
        std::atomic<size_t> next{0};
        auto work = [&]() { for (size_t j = next.fetch_add(1); j < jobs.size(); j = next.fetch_add(1)) memcpy(jobs[j].dst, jobs[j].src, jobs[j].bytes); };
        const size_t n_thr = std::min<size_t>(std::max<size_t>(jobs.size(), 1), std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
        std::vector<std::thread> th;
        for (size_t t = 1; t < n_thr; ++t) th.emplace_back(work);
        work();
        for (std::thread& x : th) x.join();
    }
    const int64_t mb = out.minu_off.back() - m0, tb = out.tex_off.back() - t0;
    const uint8_t* emp = m.p + h.off[2];
    for (int64_t i = first; i < first + count; ++i) { out.minu_off.push_back(mo[i + 1] + mb); out.tex_off.push_back(to[i + 1] + tb); out.empty.push_back(emp[i]); }
    }
    if (names) {
        const int64_t* no = (const int64_t*)(m.p + h.off[11]); const char* blob = (const char*)(m.p + h.off[12]);
        for (int64_t i = first; i < first + count; ++i) {
            if (no[i + 1] < no[i] || no[i + 1] > h.names_bytes) { err = path + ": corrupt name table"; return false; }
            names->emplace_back(blob + no[i], (size_t)(no[i + 1] - no[i]));
        }
    }
    return true;
}

}  // namespace afis
