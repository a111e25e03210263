// afis_gallery.cpp — the gallery side of the C ABI (include/afis_matcher.h): staging of rolled templates (views, .dat bytes, packed arrays, the AFISGAL1 container),
// the PQ encoder entry points, and afis_gallery_commit: SoA packing, upload through pinned buffers, the device-side derived streams.
// Replaces the per-pair load_FP_template(rolled) of matching/matcher.cpp:173 / :278: parse once, keep the shard resident in HBM.
#include "afis_ctx.h"
#include <sys/stat.h>

using namespace afis;

namespace afis {

// the staged gallery as host arrays: a container that afis_gallery_load only mapped is copied into ctx->hg now
int materialise(afis_ctx* ctx)
{
    if (!ctx->pend) return AFIS_OK;
    std::string err;
    HostGallery add;
    if (!copy_from_mapping(*ctx->pend, ctx->pend_first, ctx->pend_count, add, err)) return fail(ctx, AFIS_EFORMAT, "gallery container: " + err);   // from the mapping afis_gallery_load validated: no second open of the path
    ctx->hg = std::move(add);
    ctx->pend.reset(); ctx->pend_first = ctx->pend_count = 0;
    return AFIS_OK;
}

// t->codes == NULL: `encoded` holds the PQ codes the device made from t->des (afis_gallery_add)
void append_entry(HostGallery& hg, const afis_minutiae_view* m, const afis_texture_view* t, const uint8_t* encoded = nullptr)
{
    if (m && m->n > 0) {
        hg.mx.insert(hg.mx.end(), m->x, m->x + m->n); hg.my.insert(hg.my.end(), m->y, m->y + m->n);
        hg.mori.insert(hg.mori.end(), m->ori, m->ori + m->n);
        hg.mdes.insert(hg.mdes.end(), m->des, m->des + (size_t)m->n * kDes);
    }
    hg.minu_off.push_back((int64_t)hg.mx.size());
    if (t && t->n > 0) {
        const int n = std::min(t->n, kTexMax);                              // matcher.cpp:546-547
        hg.tx.insert(hg.tx.end(), t->x, t->x + n); hg.ty.insert(hg.ty.end(), t->y, t->y + n);
        hg.tori.insert(hg.tori.end(), t->ori, t->ori + n);
        const uint8_t* codes = t->codes ? t->codes : encoded;
        hg.tcodes.insert(hg.tcodes.end(), codes, codes + (size_t)n * kM);
    }
    hg.tex_off.push_back((int64_t)hg.tx.size());
    hg.empty.push_back((!(m && m->n > 0) && !(t && t->n > 0)) ? 1 : 0);
}

int check_rolled(afis_ctx* ctx, const afis_template_view& t)
{
    if (t.n_minu < 0 || t.n_tex < 0 || (t.n_minu > 0 && !t.minu) || (t.n_tex > 0 && !t.tex)) return fail(ctx, AFIS_EINVAL, "rolled template: bad view");
    if (t.n_minu > 0) {
        const afis_minutiae_view& m = t.minu[0];
        if (m.n <= 0 || m.n > 2000 || !m.x || !m.y || !m.ori || !m.des) return fail(ctx, AFIS_EINVAL, "rolled minutiae template: bad view (n must be 1..2000)");
        if (m.des_len != kDes) return fail(ctx, AFIS_EINVAL, "rolled minutiae template: des_len must be 96");
    }
    if (t.n_tex > 0) {
        const afis_texture_view& x = t.tex[0];
        if (x.n <= 0 || x.n > 2000 || !x.x || !x.y || !x.ori || (!x.codes && !x.des)) return fail(ctx, AFIS_EINVAL, "rolled texture template: bad view (n must be 1..2000, codes or des required)");
        if (x.codes ? x.des_len != kM : x.des_len != kDes)
            return fail(ctx, AFIS_EINVAL, "rolled texture template: des_len must be 16 with PQ codes, 96 with fp32 descriptors (encoded on the device)");
    }
    return AFIS_OK;
}

void views_of(const HostTemplate& t, std::vector<afis_minutiae_view>& mv, std::vector<afis_texture_view>& tv, afis_template_view& out)
{
    mv.clear(); tv.clear();
    for (const HostMinutiae& m : t.minu) mv.push_back({m.n(), m.x.data(), m.y.data(), m.ori.data(), m.des_len, m.des.data()});
    for (const HostTexture& x : t.tex) tv.push_back({x.n(), x.x.data(), x.y.data(), x.ori.data(), x.des_len, x.des.empty() ? nullptr : x.des.data(), x.codes.empty() ? nullptr : x.codes.data()});
    out.n_minu = (int)mv.size(); out.minu = mv.data(); out.n_tex = (int)tv.size(); out.tex = tv.data();
}

void free_gallery_dev(afis_ctx* c)
{
    c->g_minu_off.release(); c->g_minu_xy.release(); c->g_minu_ori.release(); c->g_minu_des.release(); c->g_minu_frag.release(); c->g_minu_tile_off.release();
    c->g_tex_off.release(); c->g_tex_xy.release(); c->g_tex_ori.release(); c->g_tex_codes.release(); c->g_tex_codes_cf.release(); c->g_tex_cf_blk.release(); c->g_tex_codes_q.release(); c->g_tex_q_blk.release(); c->g_tex_t32_blk.release(); c->g_empty.release(); c->g_task_ctr.release();
    c->g_codes_p.release(); c->g_nrm_p.release(); c->g_tile_meta.release(); c->mf_gal_built = false; c->codes_cf_built = false; c->codes_q_built = false;
}

}  // namespace afis

extern "C" {

int afis_gallery_add(afis_ctx* ctx, const afis_template_view* t, int n)
{
    if (!ctx || (n > 0 && !t)) return fail(ctx, AFIS_EINVAL, "afis_gallery_add: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    for (int i = 0; i < n; ++i) { int rc = check_rolled(ctx, t[i]); if (rc) return rc; }
    std::vector<uint8_t> enc;
    for (int i = 0; i < n; ++i) {
        const afis_texture_view* x = t[i].n_tex > 0 ? &t[i].tex[0] : nullptr;
        if (x && !x->codes) {                                               // fp32 descriptors: PQ-encode on the device
            enc.resize((size_t)x->n * kM);
            int rc = afis_pq_encode(ctx, x->des, x->n, enc.data());
            if (rc != AFIS_OK) return rc;
        }
        append_entry(ctx->hg, t[i].n_minu > 0 ? &t[i].minu[0] : nullptr, x, enc.data());
    }
    return AFIS_OK;
}

// PQ encoder: TrainedPQEncoder.encode_multi (extraction/descriptor_PQ.py:19-27) on the device, in slices that fit a fixed
// staging buffer.
int afis_pq_encode(afis_ctx* ctx, const float* des, int64_t n, uint8_t* codes)
{
    if (!ctx || n < 0 || (n > 0 && (!des || !codes))) return fail(ctx, AFIS_EINVAL, "afis_pq_encode: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t slice = 1 << 20;                                          // 1 Mi points = 384 MiB of descriptors per launch
    DevBuf d_des, d_codes;
    int rc = AFIS_OK;
    for (int64_t i0 = 0; i0 < n && rc == AFIS_OK; i0 += slice) {
        const int64_t m = std::min(slice, n - i0);
        if (d_des.ensure((size_t)m * kDes * 4) != hipSuccess || d_codes.ensure((size_t)m * kM) != hipSuccess ||
            hipMemcpyAsync(d_des.p, des + i0 * kDes, (size_t)m * kDes * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            launch_pq_encode(d_des.as<float>(), m, ctx->codewords.as<float>(), d_codes.as<uint8_t>(), ctx->stream) != hipSuccess ||
            hipMemcpyAsync(codes + i0 * kM, d_codes.p, (size_t)m * kM, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess)
            rc = fail(ctx, AFIS_EDEVICE, std::string("afis_pq_encode: ") + hipGetErrorString(hipGetLastError()));
    }
    d_des.release(); d_codes.release();
    return rc;
}

// The rolled branch of descriptor_PQ.py::encode_PQ (:332-349): a template whose texture descriptors are fp32 (the latent
// on-disk layout, descriptor_PQ.py:80-175) is rewritten in the rolled layout (:178-272) with every texture template's
// descriptors replaced by their PQ codes.
int afis_encode_rolled_dat(afis_ctx* ctx, const void* bytes, size_t len, void* out, size_t out_cap, size_t* out_len, int* load_rc)
{
    if (!ctx || !out_len || (len > 0 && !bytes)) return fail(ctx, AFIS_EINVAL, "afis_encode_rolled_dat: bad argument");
    HostTemplate t;
    const int rc = parse_latent_dat(bytes, len, t);
    if (load_rc) *load_rc = rc;
    if (rc < 0) { t.minu.clear(); t.tex.clear(); }
    for (HostTexture& x : t.tex) {
        if (x.des_len != kDes) return fail(ctx, AFIS_EINVAL, "afis_encode_rolled_dat: texture descriptors must be 96-d fp32");
        x.codes.resize((size_t)x.n() * kM);
        const int e = afis_pq_encode(ctx, x.des.data(), x.n(), x.codes.data());
        if (e != AFIS_OK) return e;
        x.des.clear(); x.des_len = kM;
    }
    const std::vector<uint8_t> w = write_rolled_dat(t);
    *out_len = w.size();
    if (!out || out_cap < w.size()) return out ? fail(ctx, AFIS_EINVAL, "afis_encode_rolled_dat: output buffer too small") : AFIS_OK;
    memcpy(out, w.data(), w.size());
    return AFIS_OK;
}

int afis_gallery_add_dat(afis_ctx* ctx, const void* bytes, size_t len, int* load_rc)
{
    if (!ctx) return AFIS_EINVAL;
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add_dat: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    HostTemplate t;
    int rc = parse_rolled_dat(bytes, len, t);
    // matcher.cpp:173-177: a negative code discards the template.  Code 8 (a descriptor length outside 1..192, where the reference overruns a
    // stack buffer) is this parser's own: the cursor is misaligned from there on, so the partial template is discarded too (score -1).
    if (rc < 0 || rc == 8) { t.minu.clear(); t.tex.clear(); }
    if (load_rc) *load_rc = rc;
    std::vector<afis_minutiae_view> mv; std::vector<afis_texture_view> tv; afis_template_view v;
    views_of(t, mv, tv, v);
    int ok = check_rolled(ctx, v);
    if (ok != AFIS_OK) return ok;
    append_entry(ctx->hg, v.n_minu > 0 ? &v.minu[0] : nullptr, v.n_tex > 0 ? &v.tex[0] : nullptr);
    return AFIS_OK;
}

// n rolled .dat files at once: parsed on the host's threads (a 100k-file gallery is 5 GB of parsing: 3.7 s on one thread), appended in order.
int afis_gallery_add_dat_batch(afis_ctx* ctx, const void* const* bytes, const size_t* lens, int64_t n, int* load_rc)
{
    if (!ctx || n < 0 || (n > 0 && (!bytes || !lens))) return fail(ctx, AFIS_EINVAL, "afis_gallery_add_dat_batch: bad argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add_dat_batch: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    std::vector<HostTemplate> ts((size_t)n);
    std::vector<int> rcs((size_t)n, 0);
    parallel_for(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            int rc = parse_rolled_dat(bytes[i], lens[i], ts[(size_t)i]);
            if (rc < 0 || rc == 8) { ts[(size_t)i].minu.clear(); ts[(size_t)i].tex.clear(); }     // as afis_gallery_add_dat
            rcs[(size_t)i] = rc;
        }
    });
    std::vector<afis_minutiae_view> mv; std::vector<afis_texture_view> tv; afis_template_view v;
    for (int64_t i = 0; i < n; ++i) {                                      // validate everything before anything is appended
        views_of(ts[(size_t)i], mv, tv, v);
        int ok = check_rolled(ctx, v);
        if (ok != AFIS_OK) return ok;
        if (v.n_tex > 0 && !v.tex[0].codes) return fail(ctx, AFIS_EFORMAT, "afis_gallery_add_dat_batch: rolled texture template without PQ codes");
    }
    // append_entry for all of them at once: the slots follow from the counts, the staged arrays grow once (without a zero-fill) and the templates
    // are copied to their slots by the host's threads (appending one by one was a serial pass over 50 KB per template)
    HostGallery& hg = ctx->hg;
    std::vector<int64_t> mo((size_t)n + 1), to((size_t)n + 1);
    mo[0] = (int64_t)hg.mx.size(); to[0] = (int64_t)hg.tx.size();
    for (int64_t i = 0; i < n; ++i) {
        const HostTemplate& t = ts[(size_t)i];
        mo[(size_t)i + 1] = mo[(size_t)i] + (t.minu.empty() ? 0 : t.minu[0].n());
        to[(size_t)i + 1] = to[(size_t)i] + (t.tex.empty() ? 0 : std::min(t.tex[0].n(), kTexMax));          // matcher.cpp:546-547
    }
    const size_t M = (size_t)mo[(size_t)n], X = (size_t)to[(size_t)n];
    hg.mx.resize(M); hg.my.resize(M); hg.mori.resize(M); hg.mdes.resize(M * kDes);
    hg.tx.resize(X); hg.ty.resize(X); hg.tori.resize(X); hg.tcodes.resize(X * kM);
    parallel_for(n, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const HostTemplate& t = ts[(size_t)i];
            const size_t a = (size_t)mo[(size_t)i], nm = (size_t)(mo[(size_t)i + 1] - mo[(size_t)i]);
            if (nm) {
                const HostMinutiae& m = t.minu[0];
                memcpy(&hg.mx[a], m.x.data(), nm * 2); memcpy(&hg.my[a], m.y.data(), nm * 2); memcpy(&hg.mori[a], m.ori.data(), nm * 4);
                memcpy(&hg.mdes[a * kDes], m.des.data(), nm * kDes * 4);
            }
            const size_t b = (size_t)to[(size_t)i], nt = (size_t)(to[(size_t)i + 1] - to[(size_t)i]);
            if (nt) {
                const HostTexture& x = t.tex[0];
                memcpy(&hg.tx[b], x.x.data(), nt * 2); memcpy(&hg.ty[b], x.y.data(), nt * 2); memcpy(&hg.tori[b], x.ori.data(), nt * 4);
                memcpy(&hg.tcodes[b * kM], x.codes.data(), nt * kM);
            }
        }
    });
    for (int64_t i = 0; i < n; ++i) {
        hg.minu_off.push_back(mo[(size_t)i + 1]); hg.tex_off.push_back(to[(size_t)i + 1]);
        hg.empty.push_back(mo[(size_t)i + 1] == mo[(size_t)i] && to[(size_t)i + 1] == to[(size_t)i] ? 1 : 0);
        if (load_rc) load_rc[i] = rcs[(size_t)i];
    }
    return AFIS_OK;
}

int afis_gallery_reserve(afis_ctx* ctx, int64_t n_templates)
{
    if (!ctx || n_templates < 0) return fail(ctx, AFIS_EINVAL, "afis_gallery_reserve: bad argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_reserve: gallery already committed");
    if (ctx->pend) return AFIS_OK;                                          // a mapped container is not staged in host arrays at all
    HostGallery& hg = ctx->hg;
    const double have = (double)hg.size();
    if ((double)n_templates <= have) return AFIS_OK;
    const double scale = have > 0 ? (double)n_templates / have * 1.02 : 0;  // 2 % headroom over the running average
    const size_t nm = have > 0 ? (size_t)((double)hg.mx.size() * scale) : (size_t)n_templates * 80;
    const size_t nt = have > 0 ? (size_t)((double)hg.tx.size() * scale) : (size_t)n_templates * 800;
    try {
        hg.mx.reserve(nm); hg.my.reserve(nm); hg.mori.reserve(nm); hg.mdes.reserve(nm * kDes);
        hg.tx.reserve(nt); hg.ty.reserve(nt); hg.tori.reserve(nt); hg.tcodes.reserve(nt * kM);
        hg.minu_off.reserve((size_t)n_templates + 1); hg.tex_off.reserve((size_t)n_templates + 1); hg.empty.reserve((size_t)n_templates);
    } catch (const std::bad_alloc&) { return fail(ctx, AFIS_EINVAL, "afis_gallery_reserve: out of host memory"); }
    return AFIS_OK;
}

int afis_gallery_add_packed(afis_ctx* ctx, int64_t n, const int64_t* minu_off, const int16_t* minu_x, const int16_t* minu_y,
                            const float* minu_ori, const float* minu_des, const int64_t* tex_off, const int16_t* tex_x,
                            const int16_t* tex_y, const float* tex_ori, const uint8_t* tex_codes)
{
    if (!ctx || n < 0 || !minu_off || !tex_off) return fail(ctx, AFIS_EINVAL, "afis_gallery_add_packed: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_add_packed: gallery already committed");
    if (int rc_ = materialise(ctx)) return rc_;
    for (int64_t i = 0; i < n; ++i) {
        const int64_t nm = minu_off[i + 1] - minu_off[i], nt = tex_off[i + 1] - tex_off[i];
        if (nm < 0 || nm > 2000 || nt < 0 || nt > 2000) return fail(ctx, AFIS_EINVAL, "afis_gallery_add_packed: template point count must be 0..2000");
    }
    HostGallery& hg = ctx->hg;
    const int64_t m0 = minu_off[0], m1 = minu_off[n], t0 = tex_off[0];
    hg.mx.insert(hg.mx.end(), minu_x + m0, minu_x + m1); hg.my.insert(hg.my.end(), minu_y + m0, minu_y + m1);
    hg.mori.insert(hg.mori.end(), minu_ori + m0, minu_ori + m1);
    hg.mdes.insert(hg.mdes.end(), minu_des + m0 * kDes, minu_des + m1 * kDes);
    const int64_t mbase = hg.minu_off.back() - m0;
    for (int64_t i = 0; i < n; ++i) {
        hg.minu_off.push_back(minu_off[i + 1] + mbase);
        const int64_t a = tex_off[i], nt = std::min<int64_t>(tex_off[i + 1] - a, kTexMax);
        hg.tx.insert(hg.tx.end(), tex_x + a, tex_x + a + nt); hg.ty.insert(hg.ty.end(), tex_y + a, tex_y + a + nt);
        hg.tori.insert(hg.tori.end(), tex_ori + a, tex_ori + a + nt);
        hg.tcodes.insert(hg.tcodes.end(), tex_codes + a * kM, tex_codes + (a + nt) * kM);
        hg.tex_off.push_back((int64_t)hg.tx.size());
        hg.empty.push_back((minu_off[i + 1] == minu_off[i] && nt == 0) ? 1 : 0);
    }
    (void)t0;
    return AFIS_OK;
}

// ---- packed gallery container (layout in template_io.h) ---------------------------------------------------
int afis_gallery_save(afis_ctx* ctx, const char* path, const char* const* names)
{
    if (!ctx || !path) return fail(ctx, AFIS_EINVAL, "afis_gallery_save: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_save: the host staging copy is released at commit; save before afis_gallery_commit");
    if (int rc_ = materialise(ctx)) return rc_;
    std::vector<std::string> nm;
    if (names) for (int64_t i = 0; i < ctx->hg.size(); ++i) nm.emplace_back(names[i] ? names[i] : "");
    std::string err;
    if (!write_gallery_container(path, ctx->hg, nm, err)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_save: " + err);
    return AFIS_OK;
}

int afis_gallery_load(afis_ctx* ctx, const char* path, int64_t first, int64_t count)
{
    if (!ctx || !path) return fail(ctx, AFIS_EINVAL, "afis_gallery_load: null argument");
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_load: gallery already committed");
    std::string err;
    if (ctx->hg.size() == 0 && !ctx->pend) {                               // the usual case (one container, or one shard of it): map it, validate it, read it at the commit
        std::unique_ptr<GalleryMapping> gm = map_gallery_container(path, err);
        if (!gm) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: " + err);
        if (count < 0) count = gm->G - first;
        if (first < 0 || count < 0 || first + count > gm->G) return fail(ctx, AFIS_EFORMAT, std::string("afis_gallery_load: ") + path + ": template range outside the container");
        for (int64_t i = first; i < first + count; ++i) {
            const int64_t nm = gm->minu_off[i + 1] - gm->minu_off[i], nt = gm->tex_off[i + 1] - gm->tex_off[i];
            if (nm > 2000 || nt > kTexMax || (gm->empty[i] != 0) != (nm == 0 && nt == 0)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: template counts out of range");
        }
        ctx->pend = std::move(gm); ctx->pend_first = first; ctx->pend_count = count;
        return AFIS_OK;
    }
    if (int rc_ = materialise(ctx)) return rc_;
    HostGallery add;                                                       // parsed aside so a bad file leaves the staged gallery untouched
    if (!read_gallery_container(path, first, count, add, nullptr, nullptr, err)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: " + err);
    const int64_t n = add.size();
    for (int64_t i = 0; i < n; ++i) {
        const int64_t nm = add.minu_off[i + 1] - add.minu_off[i], nt = add.tex_off[i + 1] - add.tex_off[i];
        if (nm > 2000 || nt > kTexMax || (add.empty[i] != 0) != (nm == 0 && nt == 0)) return fail(ctx, AFIS_EFORMAT, "afis_gallery_load: template counts out of range");
    }
    HostGallery& hg = ctx->hg;
    if (hg.size() == 0) { hg = std::move(add); return AFIS_OK; }
    const int64_t mb = hg.minu_off.back(), tb = hg.tex_off.back();
    hg.mx.insert(hg.mx.end(), add.mx.begin(), add.mx.end()); hg.my.insert(hg.my.end(), add.my.begin(), add.my.end());
    hg.mori.insert(hg.mori.end(), add.mori.begin(), add.mori.end()); hg.mdes.insert(hg.mdes.end(), add.mdes.begin(), add.mdes.end());
    hg.tx.insert(hg.tx.end(), add.tx.begin(), add.tx.end()); hg.ty.insert(hg.ty.end(), add.ty.begin(), add.ty.end());
    hg.tori.insert(hg.tori.end(), add.tori.begin(), add.tori.end()); hg.tcodes.insert(hg.tcodes.end(), add.tcodes.begin(), add.tcodes.end());
    for (int64_t i = 0; i < n; ++i) { hg.minu_off.push_back(mb + add.minu_off[i + 1]); hg.tex_off.push_back(tb + add.tex_off[i + 1]); hg.empty.push_back(add.empty[i]); }
    return AFIS_OK;
}

int afis_gallery_file_info(const char* path, int64_t* G, int64_t* n_minutiae, int64_t* n_tex_points, int32_t* tex_counts)
{
    if (!path) return fail(nullptr, AFIS_EINVAL, "afis_gallery_file_info: null argument");
    std::string err;
    GalleryFileInfo info;
    if (!gallery_container_info(path, info, err)) return fail(nullptr, AFIS_EFORMAT, "afis_gallery_file_info: " + err);
    if (G) *G = info.G;
    if (n_minutiae) *n_minutiae = info.n_minu;
    if (n_tex_points) *n_tex_points = info.n_tex;
    if (tex_counts) {
        HostGallery none; std::vector<int32_t> tc;
        if (!read_gallery_container(path, 0, 0, none, nullptr, &tc, err, false)) return fail(nullptr, AFIS_EFORMAT, "afis_gallery_file_info: " + err);
        memcpy(tex_counts, tc.data(), tc.size() * sizeof(int32_t));
    }
    return AFIS_OK;
}

int afis_gallery_file_names(const char* path, int64_t first, int64_t count, char* buf, size_t cap, size_t* need)
{
    if (!path || !need) return fail(nullptr, AFIS_EINVAL, "afis_gallery_file_names: null argument");
    std::string err;
    HostGallery none; std::vector<std::string> names;
    if (!read_gallery_container(path, first, count, none, &names, nullptr, err, false)) return fail(nullptr, AFIS_EFORMAT, "afis_gallery_file_names: " + err);
    size_t total = 0;
    for (const std::string& n : names) total += n.size() + 1;
    *need = total;
    if (!buf) return AFIS_OK;
    if (cap < total) return fail(nullptr, AFIS_EINVAL, "afis_gallery_file_names: buffer too small");
    char* w = buf;
    for (const std::string& n : names) { memcpy(w, n.c_str(), n.size() + 1); w += n.size() + 1; }
    return AFIS_OK;
}

// The arrays of a shard are 50 KB per template (5 GB per 100 000): a pageable hipMemcpy moves them at 8-11 GB/s through the runtime's one staging thread.
// Here they go through two pinned 64 MB buffers: the host's threads fill one (from the staged arrays or straight from a mapped container: that is where
// the page cache is read) while the DMA engine empties the other.
struct PinnedPipe {
    static constexpr size_t kCap = (size_t)64 << 20;
    void* buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false}; int k = 0;
    hipError_t init()
    {
        for (int i = 0; i < 2; ++i) {
            hipError_t e = hipHostMalloc(&buf[i], kCap, hipHostMallocDefault); if (e != hipSuccess) return e;
            e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming); if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    ~PinnedPipe() { for (int i = 0; i < 2; ++i) { if (ev[i]) (void)hipEventDestroy(ev[i]); if (buf[i]) (void)hipHostFree(buf[i]); } }
};

static hipError_t upload_bulk(PinnedPipe& pp, DevBuf& b, const void* src, size_t bytes, hipStream_t s)
{
    hipError_t e = b.ensure(std::max<size_t>(bytes, 16));
    if (e != hipSuccess || bytes == 0) return e;
    if (bytes < ((size_t)4 << 20)) return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, s);
    for (size_t off = 0; off < bytes; off += PinnedPipe::kCap) {
        const size_t n = std::min(PinnedPipe::kCap, bytes - off);
        const int slot = pp.k & 1;
        if (pp.used[slot]) { e = hipEventSynchronize(pp.ev[slot]); if (e != hipSuccess) return e; }
        const uint8_t* from = (const uint8_t*)src + off; uint8_t* to = (uint8_t*)pp.buf[slot];
        parallel_for((int64_t)((n + 4095) / 4096), [&](int64_t lo, int64_t hi) { const size_t a = (size_t)lo * 4096, z = std::min(n, (size_t)hi * 4096); memcpy(to + a, from + a, z - a); });
        e = hipMemcpyAsync((uint8_t*)b.p + off, pp.buf[slot], n, hipMemcpyHostToDevice, s); if (e != hipSuccess) return e;
        e = hipEventRecord(pp.ev[slot], s); if (e != hipSuccess) return e;
        pp.used[slot] = true; ++pp.k;
    }
    return hipSuccess;
}

static int commit_shard(afis_ctx* ctx, int64_t index_base);

// A failed commit leaves the context as it was before the call: not committed, no half-uploaded shard on the device, the staged templates (host arrays or the mapped
// container) still in place, so that the caller may retry or destroy.
int afis_gallery_commit(afis_ctx* ctx, int64_t index_base)
{
    if (!ctx) return AFIS_EINVAL;
    if (ctx->committed) return fail(ctx, AFIS_ESTATE, "afis_gallery_commit: already committed");
    const int rc = commit_shard(ctx, index_base);
    if (rc != AFIS_OK) {
        ctx->committed = false;
        (void)hipStreamSynchronize(ctx->stream);
        free_gallery_dev(ctx);
        ctx->gal = GalleryDev();
    }
    return rc;
}

static int commit_shard(afis_ctx* ctx, int64_t index_base)
{
    // The staged shard as plain arrays: ctx->hg, or the mapped container's range (offsets rebased to the shard's first point).
    HostGallery& hg = ctx->hg;
    const GalleryMapping* gm = ctx->pend.get();
    const int64_t G = gm ? ctx->pend_count : hg.size();
    const int64_t* src_mo = gm ? gm->minu_off + ctx->pend_first : hg.minu_off.data();
    const int64_t* src_to = gm ? gm->tex_off + ctx->pend_first : hg.tex_off.data();
    const int64_t m0 = src_mo[0], t0 = src_to[0];
    const size_t NM = (size_t)(src_mo[G] - m0), NT = (size_t)(src_to[G] - t0);
    const int16_t* s_mx = gm ? gm->mx + m0 : hg.mx.data(); const int16_t* s_my = gm ? gm->my + m0 : hg.my.data();
    const float* s_mori = gm ? gm->mori + m0 : hg.mori.data(); const float* s_mdes = gm ? gm->mdes + (size_t)m0 * kDes : hg.mdes.data();
    const int16_t* s_tx = gm ? gm->tx + t0 : hg.tx.data(); const int16_t* s_ty = gm ? gm->ty + t0 : hg.ty.data();
    const float* s_tori = gm ? gm->tori + t0 : hg.tori.data(); const uint8_t* s_tcodes = gm ? gm->tcodes + (size_t)t0 * kM : hg.tcodes.data();
    const uint8_t* s_empty = gm ? gm->empty + ctx->pend_first : hg.empty.data();
    if (G > 0x7fffffff / 8 || NM > 0x7fffffffull || NT > 0x7fffffffull)
        return fail(ctx, AFIS_EINVAL, "afis_gallery_commit: shard too large for 32-bit point offsets; split the gallery into more shards");
    if (gm && gm->fd_ >= 0) {                                               // the arrays are read straight from the mapping, by several threads: a file truncated since afis_gallery_load would fault.  This check catches a truncation that
                                                                            // happened BEFORE the commit; one that happens while the copy below runs still faults (SIGBUS): a container must not be modified while a context has it loaded (include/afis_matcher.h says so)
        struct stat st;
        if (fstat(gm->fd_, &st) != 0 || (size_t)st.st_size < gm->len_) return fail(ctx, AFIS_EFORMAT, "gallery container: " + gm->path + " was truncated after it was loaded");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool clock_it = getenv("AFIS_COMMIT_TIMING") != nullptr;           // development aid: where the commit's time goes, on stderr
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_prev = now();
    auto lap = [&](const char* what) { if (clock_it) { (void)hipStreamSynchronize(ctx->stream); const double t = now(); fprintf(stderr, "commit: %-28s %8.1f ms\n", what, t - t_prev); t_prev = t; } };
    PinnedPipe pp;
    HIPCHK(ctx, pp.init());
    lap("pinned buffers");
    std::vector<int32_t> mo(G + 1), to(G + 1);
    int max_nR = 0;
    for (int64_t i = 0; i <= G; ++i) { mo[i] = (int32_t)(src_mo[i] - m0); to[i] = (int32_t)(src_to[i] - t0); }
    for (int64_t i = 0; i < G; ++i) max_nR = std::max(max_nR, mo[i + 1] - mo[i]);
    HIPCHK(ctx, upload_bulk(pp, ctx->g_minu_des, s_mdes, NM * kDes * sizeof(float), ctx->stream));     // the big one first: the fragment kernel below runs while the rest is uploaded
    lap("minutiae descriptors");
    HIPCHK(ctx, upload(ctx->g_minu_off, mo, ctx->stream));
    std::vector<int32_t> toff((size_t)G + 1, 0);
    {   // the descriptors as MFMA operand fragments: laid out on the device from the descriptors just uploaded (round 3 transposed them on the host and uploaded another 34 KB per template)
        for (int64_t t = 0; t < G; ++t) toff[(size_t)t + 1] = toff[(size_t)t] + (mo[t + 1] - mo[t] + 15) / 16;
        HIPCHK(ctx, upload(ctx->g_minu_tile_off, toff, ctx->stream));
        HIPCHK(ctx, ctx->g_minu_frag.ensure(std::max<size_t>((size_t)toff[(size_t)G] * 6 * 64 * 16, 16)));
        HIPCHK(ctx, launch_fragment_tiles(ctx->g_minu_des.as<float>(), ctx->g_minu_off.as<int32_t>(), ctx->g_minu_tile_off.as<int32_t>(), (int)G, ctx->g_minu_frag.p, ctx->stream));
    }
    lap("fragment tiles");
    std::vector<short2> mxy(NM), txy(NT);
    parallel_for((int64_t)NM, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) mxy[(size_t)i] = make_short2(s_mx[i], s_my[i]); });
    parallel_for((int64_t)NT, [&](int64_t lo, int64_t hi) { for (int64_t i = lo; i < hi; ++i) txy[(size_t)i] = make_short2(s_tx[i], s_ty[i]); });
    lap("xy packing");
    HIPCHK(ctx, upload_bulk(pp, ctx->g_minu_xy, mxy.data(), NM * sizeof(short2), ctx->stream));
    HIPCHK(ctx, upload_bulk(pp, ctx->g_minu_ori, s_mori, NM * sizeof(float), ctx->stream));
    HIPCHK(ctx, upload(ctx->g_tex_off, to, ctx->stream));
    HIPCHK(ctx, upload_bulk(pp, ctx->g_tex_xy, txy.data(), NT * sizeof(short2), ctx->stream));
    HIPCHK(ctx, upload_bulk(pp, ctx->g_tex_ori, s_tori, NT * sizeof(float), ctx->stream));
    lap("small arrays");
    HIPCHK(ctx, upload_bulk(pp, ctx->g_tex_codes, s_tcodes, NT * kM, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    lap("texture codes");
    {   // block offsets of the direct conflict-free kernel's code stream (variants 6 / 7): (blocks + 1) x 64 entries per template.  The stream
        // itself — a full copy of the PQ codes — is laid out on the device at the first use of those variants (k_codes_cf); the default path
        // never builds it.
        std::vector<int32_t> cfb(G + 1);
        int64_t nblk = 0;
        for (int64_t t = 0; t < G; ++t) { cfb[t] = (int32_t)nblk; const int64_t n = (int64_t)(to[t + 1] - to[t]); nblk += n > 0 ? (n + 63) / 64 + 1 : 0; }
        cfb[G] = (int32_t)nblk;
        if (nblk > 0x7fffffff / 64) return fail(ctx, AFIS_EINVAL, "afis_gallery_commit: shard too large for the ADC code stream; split the gallery into more shards");
        ctx->cf_blocks = nblk; ctx->codes_cf_built = false;
        HIPCHK(ctx, upload(ctx->g_tex_cf_blk, cfb, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    {   // block offsets of the quantised path's code stream (ceil(n/64) blocks per template); the stream itself is made on first use
        std::vector<int32_t> qb(G + 1);
        int64_t nb = 0;
        for (int64_t t = 0; t < G; ++t) { qb[t] = (int32_t)nb; nb += ((int64_t)(to[t + 1] - to[t]) + 63) / 64; }
        qb[G] = (int32_t)nb;
        ctx->q_blocks = nb;
        HIPCHK(ctx, upload(ctx->g_tex_q_blk, qb, ctx->stream));
    }
    {   // tile offsets of the matrix-core bound pass's stream (ceil(n/32) tiles of 32 points per template); the stream itself is made on first use
        std::vector<int32_t> tb(G + 1);
        int64_t nt = 0;
        for (int64_t t = 0; t < G; ++t) { tb[t] = (int32_t)nt; nt += ((int64_t)(to[t + 1] - to[t]) + 31) / 32; }
        tb[G] = (int32_t)nt;
        if (nt > 0x7fffffff / 32) return fail(ctx, AFIS_EINVAL, "afis_gallery_commit: shard too large for the bound pass's code stream; split the gallery into more shards");
        ctx->t32_tiles = nt;
        HIPCHK(ctx, upload(ctx->g_tex_t32_blk, tb, ctx->stream));
    }
    { DevBuf& eb = ctx->g_empty; HIPCHK(ctx, eb.ensure(std::max<size_t>((size_t)G, 16))); if (G) HIPCHK(ctx, hipMemcpyAsync(eb.p, s_empty, (size_t)G, hipMemcpyHostToDevice, ctx->stream)); }
    HIPCHK(ctx, ctx->g_task_ctr.ensure(64));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    GalleryDev& g = ctx->gal;
    g.G = (int32_t)G;
    g.minu_off = ctx->g_minu_off.as<int32_t>(); g.minu_xy = ctx->g_minu_xy.as<short2>(); g.minu_ori = ctx->g_minu_ori.as<float>();
    g.minu_des = ctx->g_minu_des.as<float>(); g.minu_frag = ctx->g_minu_frag.as<float4>(); g.minu_tile_off = ctx->g_minu_tile_off.as<int32_t>(); g.tex_off = ctx->g_tex_off.as<int32_t>(); g.tex_xy = ctx->g_tex_xy.as<short2>();
    g.tex_ori = ctx->g_tex_ori.as<float>(); g.tex_codes = ctx->g_tex_codes.as<uint4>(); g.tex_codes_cf = nullptr; g.tex_cf_blk = ctx->g_tex_cf_blk.as<int32_t>(); g.empty = ctx->g_empty.as<uint8_t>();
    g.task_ctr = ctx->g_task_ctr.as<int32_t>();
    ctx->max_nR = max_nR;
    ctx->total_tex_points = (int64_t)NT; ctx->total_minutiae = (int64_t)NM;
    ctx->index_base = index_base;
    ctx->committed = true;
    if (ctx->adc_variant == 9 && G > 0) {                                    // the default path's derived streams belong to the resident gallery: built here, not by the first search
        int rcg = ensure_mf_gallery(ctx, ctx->stream);
        if (rcg != AFIS_OK) return rcg;
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        lap("bound pass's code stream");
    }
    // the host staging copy is no longer needed
    std::vector<uint8_t> e(s_empty, s_empty + G);
    if (hg.mdes.capacity() > ((size_t)16 << 20)) {                           // a large staging copy is released by a thread of its own
        // The pages go back in 32 MB pieces (madvise takes the address-space lock shared and briefly); one munmap of 3 GB holds it exclusively for
        // a third of a second, and every allocation the caller makes next — the commit's own clean-up, the first search — would wait for it.
        HostGallery* old = new HostGallery(std::move(ctx->hg));
        ctx->staging_reaper = std::thread([old]() {
            std::vector<std::pair<uintptr_t, size_t>> pieces;
            auto drop = [&](void* p, size_t bytes) {
                const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, z = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
                for (uintptr_t q = a; q < z; q += (uintptr_t)32 << 20) pieces.emplace_back(q, (size_t)std::min<uintptr_t>((uintptr_t)32 << 20, z - q));
            };
            drop(old->mdes.data(), old->mdes.capacity() * sizeof(float)); drop(old->tcodes.data(), old->tcodes.capacity());
            drop(old->mori.data(), old->mori.capacity() * 4); drop(old->tori.data(), old->tori.capacity() * 4);
            std::atomic<size_t> next{0};
            auto work = [&]() { for (size_t i = next.fetch_add(1); i < pieces.size(); i = next.fetch_add(1)) (void)madvise((void*)pieces[i].first, pieces[i].second, MADV_DONTNEED); };
            std::thread helpers[3];                                          // four threads return 5 GB in a quarter of the time one takes
            for (std::thread& h : helpers) h = std::thread(work);
            work();
            for (std::thread& h : helpers) h.join();
            delete old;
        });
    }
    lap("offset tables");
    ctx->hg = HostGallery(); ctx->hg.empty = std::move(e);
    ctx->pend.reset(); ctx->pend_first = ctx->pend_count = 0;
    lap("staging released");
    return AFIS_OK;
}

int64_t afis_gallery_size(const afis_ctx* ctx) { return !ctx ? 0 : ctx->pend ? ctx->pend_count : (int64_t)ctx->hg.empty.size(); }

}  // extern "C"
