// tio_check — host-only self-test hook for template_io (no GPU, no HIP): parses a latent/rolled .dat or a codebook and prints a
// canonical summary (return code, template counts, per-template sizes and FNV-1a hashes of the payloads); `roundtrip` re-writes the
// parsed template and reports whether the bytes are identical.  Used by tests/test_host.py to cross-check the C++ parser/writer
// against the Python mirror and the oracle's parser.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "template_io.h"

using namespace afis;

static unsigned long long fnv(const void* p, size_t n, unsigned long long h = 1469598103934665603ull)
{
    const unsigned char* b = (const unsigned char*)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: tio_check latent|rolled|codebook|roundtrip-latent|roundtrip-rolled <file>\n"
                                    "       tio_check gallery-pack <out container> <rolled .dat>...\n"
                                    "       tio_check gallery-dump <container> [first count]\n"
                                    "       tio_check gallery-map <container> [first count]\n"); return 2; }
    const std::string mode = argv[1];
    if (mode == "gallery-pack") {                        // the host half of `match -pack`: rolled .dat files -> one container
        HostGallery g; std::vector<std::string> names; std::vector<uint8_t> fb;
        for (int i = 3; i < argc; ++i) {
            HostTemplate t;
            if (!read_file(argv[i], fb)) { printf("rc=io\n"); return 1; }
            if (parse_rolled_dat(fb.data(), fb.size(), t) < 0) { t.minu.clear(); t.tex.clear(); }
            if (!gallery_append_template(g, t)) { fprintf(stderr, "%s: descriptor width is not 96 / 16\n", argv[i]); return 3; }
            names.push_back(argv[i]);
        }
        std::string err;
        if (!write_gallery_container(argv[2], g, names, err)) { printf("error=%s\n", err.c_str()); return 1; }
        printf("ok G=%lld\n", (long long)g.size());
        return 0;
    }
    if (mode == "gallery-dump") {
        const long long first = argc > 4 ? atoll(argv[3]) : 0, count = argc > 4 ? atoll(argv[4]) : -1;
        HostGallery g; std::vector<std::string> names; std::vector<int32_t> tc; std::string err; GalleryFileInfo info;
        if (!gallery_container_info(argv[2], info, err) || !read_gallery_container(argv[2], first, count, g, &names, &tc, err)) { printf("error=%s\n", err.c_str()); return 1; }
        printf("G=%lld n_minu=%lld n_tex=%lld range=%lld\n", (long long)info.G, (long long)info.n_minu, (long long)info.n_tex, (long long)g.size());
        unsigned long long h = fnv(g.minu_off.data(), g.minu_off.size() * 8); h = fnv(g.tex_off.data(), g.tex_off.size() * 8, h); h = fnv(g.empty.data(), g.empty.size(), h);
        printf("offsets hash=%016llx\n", h);
        h = fnv(g.mx.data(), g.mx.size() * 2); h = fnv(g.my.data(), g.my.size() * 2, h); h = fnv(g.mori.data(), g.mori.size() * 4, h); h = fnv(g.mdes.data(), g.mdes.size() * 4, h);
        printf("minutiae hash=%016llx\n", h);
        h = fnv(g.tx.data(), g.tx.size() * 2); h = fnv(g.ty.data(), g.ty.size() * 2, h); h = fnv(g.tori.data(), g.tori.size() * 4, h); h = fnv(g.tcodes.data(), g.tcodes.size(), h);
        printf("texture hash=%016llx\n", h);
        h = 1469598103934665603ull; for (const std::string& n : names) h = fnv(n.c_str(), n.size() + 1, h);
        printf("names hash=%016llx\n", h);
        h = fnv(tc.data(), tc.size() * 4);
        printf("tex_counts hash=%016llx\n", h);
        return 0;
    }
    if (mode == "gallery-map") {                         // the same hashes as gallery-dump, read IN PLACE through map_gallery_container (what afis_gallery_load keeps and afis_gallery_commit uploads from)
        long long first = argc > 4 ? atoll(argv[3]) : 0, count = argc > 4 ? atoll(argv[4]) : -1;
        std::string err;
        std::unique_ptr<GalleryMapping> gm = map_gallery_container(argv[2], err);
        if (!gm) { printf("error=%s\n", err.c_str()); return 1; }
        if (count < 0) count = gm->G - first;
        if (first < 0 || count < 0 || first + count > gm->G) { printf("error=range outside the container\n"); return 1; }
        const int64_t m0 = gm->minu_off[first], m1 = gm->minu_off[first + count], t0 = gm->tex_off[first], t1 = gm->tex_off[first + count];
        std::vector<int64_t> mo, to;
        for (long long i = first; i <= first + count; ++i) { mo.push_back(gm->minu_off[i] - m0); to.push_back(gm->tex_off[i] - t0); }
        printf("G=%lld n_minu=%lld n_tex=%lld range=%lld\n", (long long)gm->G, (long long)gm->n_minu, (long long)gm->n_tex, count);
        unsigned long long h = fnv(mo.data(), mo.size() * 8); h = fnv(to.data(), to.size() * 8, h); h = fnv(gm->empty + first, (size_t)count, h);
        printf("offsets hash=%016llx\n", h);
        const size_t nm = (size_t)(m1 - m0), nt = (size_t)(t1 - t0);
        h = fnv(gm->mx + m0, nm * 2); h = fnv(gm->my + m0, nm * 2, h); h = fnv(gm->mori + m0, nm * 4, h); h = fnv(gm->mdes + (size_t)m0 * 96, nm * 96 * 4, h);
        printf("minutiae hash=%016llx\n", h);
        h = fnv(gm->tx + t0, nt * 2); h = fnv(gm->ty + t0, nt * 2, h); h = fnv(gm->tori + t0, nt * 4, h); h = fnv(gm->tcodes + (size_t)t0 * 16, nt * 16, h);
        printf("texture hash=%016llx\n", h);
        return 0;
    }
    std::vector<uint8_t> b;
    if (!read_file(argv[2], b)) { printf("rc=io\n"); return 1; }
    if (mode == "codebook") {
        HostCodebook cb;
        const bool ok = parse_codebook(b.data(), b.size(), cb);
        printf("ok=%d M=%d K=%d dsub=%d hash=%016llx\n", ok ? 1 : 0, cb.M, cb.K, cb.dsub, fnv(cb.words.data(), cb.words.size() * 4));
        return 0;
    }
    const bool rolled = mode.find("rolled") != std::string::npos;
    HostTemplate t;
    const int rc = rolled ? parse_rolled_dat(b.data(), b.size(), t) : parse_latent_dat(b.data(), b.size(), t);
    if (mode.rfind("roundtrip", 0) == 0) {
        const std::vector<uint8_t> w = rolled ? write_rolled_dat(t) : write_latent_dat(t);
        printf("rc=%d identical=%d in=%zu out=%zu\n", rc, (w == b) ? 1 : 0, b.size(), w.size());
        return 0;
    }
    printf("rc=%d n_minu=%zu n_tex=%zu h=%d w=%d blkH=%d blkW=%d\n", rc, t.minu.size(), t.tex.size(), t.h, t.w, t.blkH, t.blkW);
    for (const HostMinutiae& m : t.minu) {
        unsigned long long h = fnv(m.x.data(), m.x.size() * 2); h = fnv(m.y.data(), m.y.size() * 2, h); h = fnv(m.ori.data(), m.ori.size() * 4, h);
        h = fnv(m.des.data(), m.des.size() * 4, h);
        printf("minu n=%d des_len=%d hash=%016llx\n", m.n(), m.des_len, h);
    }
    for (const HostTexture& x : t.tex) {
        unsigned long long h = fnv(x.x.data(), x.x.size() * 2); h = fnv(x.y.data(), x.y.size() * 2, h); h = fnv(x.ori.data(), x.ori.size() * 4, h);
        h = fnv(x.des.data(), x.des.size() * 4, h); h = fnv(x.codes.data(), x.codes.size(), h);
        printf("tex n=%d des_len=%d hash=%016llx\n", x.n(), x.des_len, h);
    }
    return 0;
}
