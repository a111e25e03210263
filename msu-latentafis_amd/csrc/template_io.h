// template_io.h — host-side data model and the reference's on-disk formats (C++17, no third-party code).
//   codebook .dat : matching/matcher.cpp:74-93
//   latent   .dat : reader matching/matcher.cpp:785-884, writer extraction/descriptor_PQ.py:80-175
//   rolled   .dat : reader matching/matcher.cpp:886-983, writer extraction/descriptor_PQ.py:178-272
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace afis {

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
