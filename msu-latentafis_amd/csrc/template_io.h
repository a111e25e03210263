// template_io.h — host-side data model and the reference's on-disk formats (C++17, no third-party code).
//   codebook .dat : matching/matcher.cpp:74-93
//   latent   .dat : reader matching/matcher.cpp:785-884, writer extraction/descriptor_PQ.py:80-175
//   rolled   .dat : reader matching/matcher.cpp:886-983, writer extraction/descriptor_PQ.py:178-272
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace afis {

// Allocator whose `construct()` default-initialises: resize() of a trivially constructible element type leaves the new elements
// unwritten instead of zero-filling them (the staged gallery is 5 GB per 100 000 templates; every byte is overwritten by the copy
// that follows the resize, and the zero-fill would be a second, single-threaded pass over it).
template <class T> struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { typedef DefaultInit<U> other; };
    DefaultInit() = default;
    template <class U> DefaultInit(const DefaultInit<U>&) {}
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class... A> void construct(U* p, A&&... a) { ::new ((void*)p) U(std::forward<A>(a)...); }
};
template <class T> using BulkVec = std::vector<T, DefaultInit<T>>;

struct HostMinutiae {            // MinutiaeTemplate, matching/include.h:203-252
    std::vector<int16_t> x, y;   // pixels
    std::vector<float> ori;
    int des_len = 0;
    std::vector<float> des;      // [n][des_len]
    int n() const { return (int)x.size(); }
};
struct HostTexture {             // LatentTextureTemplate / RolledTextureTemplatePQ, include.h:298-485
    std::vector<int16_t> x, y;   // block units
    std::vector<float> ori;
    int des_len = 0;
    std::vector<float> des;      // latent: [n][96]
    std::vector<uint8_t> codes;  // rolled: [n][16]
    int n() const { return (int)x.size(); }
};
struct HostTemplate {            // LatentFPTemplate / RolledFPTemplate, include.h:519-558 (zero-minutiae templates already dropped)
    std::vector<HostMinutiae> minu;
    std::vector<HostTexture> tex;
    int h = 0, w = 0, blkH = 0, blkW = 0;
};
struct HostCodebook {            // matcher.cpp:70-93
    int M = 0, K = 0, dsub = 0;
    std::vector<float> words;    // [M][K][dsub]
};

// The staged gallery shard in host memory (what afis_gallery_add* accumulate and afis_gallery_commit uploads): only rolled
// minutiae template 0 and texture template 0 of every file (the ones the matcher reads, matcher.cpp:406,413), texture counts
// clamped to 1000 (matcher.cpp:546-547), points of all templates concatenated with CSR offsets.
struct HostGallery {
    std::vector<int64_t> minu_off{0}, tex_off{0};
    BulkVec<int16_t> mx, my, tx, ty;
    BulkVec<float> mori, mdes, tori;
    BulkVec<uint8_t> tcodes;
    std::vector<uint8_t> empty;  // 1 = the file had neither template (status 2, score -1)
    int64_t size() const { return (int64_t)empty.size(); }
};

// Packed gallery container: ONE file instead of 100k-1M tiny .dat files, the SoA arrays of HostGallery as they
// are, every section 64-byte aligned so the file can be mmap-ed and a contiguous template range [first, first+count) — a
// shard — read without touching the rest.  Little-endian.
//   0   char[8]  "AFISGAL1"      8  u32 version (1), u32 des_len (96), u32 code_len (16), u32 0
//   24  i64 G, i64 n_minutiae, i64 n_texture_points, i64 names_bytes
//   56  u64 offset[13]: minu_off i64[G+1] | tex_off i64[G+1] | empty u8[G] | minu_x i16[] | minu_y i16[] | minu_ori f32[] |
//       minu_des f32[][96] | tex_x i16[] | tex_y i16[] | tex_ori f32[] | tex_codes u8[][16] | name_off i64[G+1] | names char[]
// names[i] = the path of the .dat file template i came from (what the score files print).
// appends one parsed rolled template the way afis_gallery_add_dat does: minutiae template 0, texture template 0 clamped to 1000
bool gallery_append_template(HostGallery& g, const HostTemplate& t);   // false: descriptor / code width is not 96 / 16
struct GalleryFileInfo { int64_t G = 0, n_minu = 0, n_tex = 0; };
bool write_gallery_container(const std::string& path, const HostGallery& g, const std::vector<std::string>& names, std::string& err);
bool gallery_container_info(const std::string& path, GalleryFileInfo& info, std::string& err);
// appends templates [first, first+count) (count < 0: to the end) to `out`; names / tex_counts (texture points of EVERY template in
// the file, for balanced sharding) are optional
// A validated container read IN PLACE (mmap): what afis_gallery_load keeps until the commit, which uploads the shard's arrays straight from the
// mapping instead of copying them into a HostGallery first.  Pointers are to the whole file's arrays; offsets are the file's (not rebased).
struct GalleryMapping {
    int64_t G = 0, n_minu = 0, n_tex = 0;
    const int64_t* minu_off = nullptr; const int64_t* tex_off = nullptr; const uint8_t* empty = nullptr;
    const int16_t *mx = nullptr, *my = nullptr, *tx = nullptr, *ty = nullptr;
    const float *mori = nullptr, *mdes = nullptr, *tori = nullptr;
    const uint8_t* tcodes = nullptr;
    std::string path;
    GalleryMapping() = default;
    GalleryMapping(const GalleryMapping&) = delete;
    GalleryMapping& operator=(const GalleryMapping&) = delete;
    ~GalleryMapping();
    void* base_ = nullptr; size_t len_ = 0; int fd_ = -1;
};
std::unique_ptr<GalleryMapping> map_gallery_container(const std::string& path, std::string& err);   // header, section sizes and offsets checked; null on error
bool copy_from_mapping(const GalleryMapping& g, int64_t first, int64_t count, HostGallery& out, std::string& err);   // a range of an already validated mapping -> host arrays
bool read_gallery_container(const std::string& path, int64_t first, int64_t count, HostGallery& out, std::vector<std::string>* names,
                            std::vector<int32_t>* tex_counts, std::string& err, bool load_data = true);   // load_data false: names / counts only, the arrays are not copied

// Return codes of the two parsers are the reference's: 0 ok, 1 empty file (latent: size <= 0, rolled: size <= 10),
// 2 too many minutiae in a minutiae template, 4 ridge-flow block too large, -1 too many points in a texture template.
// On a non-zero code `out` holds whatever had been parsed before the error, exactly as the reference leaves it.
int parse_latent_dat(const void* bytes, size_t len, HostTemplate& out);
int parse_rolled_dat(const void* bytes, size_t len, HostTemplate& out);
bool parse_codebook(const void* bytes, size_t len, HostCodebook& out);

std::vector<uint8_t> write_latent_dat(const HostTemplate& t);
std::vector<uint8_t> write_rolled_dat(const HostTemplate& t);

bool read_file(const std::string& path, std::vector<uint8_t>& out);

}  // namespace afis
