#include "template_io.h"

#include <cstdio>
#include <cstring>

namespace afis {
namespace {

constexpr int kMaxMinutiae = 2000;   // matcher.cpp:788
constexpr int kMaxDesLength = 192;   // matcher.cpp:789
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
        if (c.fail || dl <= 0 || dl > kMaxDesLength) break;
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
        if (c.fail || dl <= 0 || dl > kMaxDesLength) break;
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

}  // namespace afis
