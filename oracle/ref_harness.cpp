// ref_harness.cpp — thin C-ABI driver around the REFERENCE's own header, compiled from where it lies
// (-I /root/reference/matching; nothing from the reference is copied into this repo).
//
// TEST INFRASTRUCTURE ONLY (see oracle/afis_oracle.cpp header).  Output: oracle/_ref/libafis_ref.so.
//
// Only matching/include.h is buildable in this image: it is self-contained.  matching/matcher.cpp needs
// Eigen and Boost.Filesystem, which are absent from the reference tree and from the image, so it is
// treated as unbuildable (no stand-in headers are written for it).  What this pins:
//   * LatentTextureTemplate::compute_dist_to_codewords (include.h:327-359)  -> S4, the per-query PQ LUT
//   * RolledTextureTemplatePQ(n,x,y,ori,des_len,des)  (include.h:401-406)   -> PQ code extraction from the
//     float-typed read buffer, and the short->int point conversion (include.h:171-193)
//   * MinutiaeTemplate(...) (include.h:215-237)                                -> descriptor/point copy
//   * ArgParser (matching/argparser.h, also self-contained once <string> and <algorithm> are in scope, as main.cpp has them)
//                                                                              -> the `match` command line's token rules
//   * json.hpp (the JSON library vendored in matching/, used by main.cpp:41-44 to read ../afis.config)
//                                                                              -> the config-file fallback of the CLIs
#include "include.h"
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include "argparser.h"
#include <fstream>
#include "json.hpp"

extern "C" {

// S4 through the reference class.  des: [n][des_len]; codewords: [M][K][dsub]; out: [n][M][K].
void ref_build_lut(int n, const short* x, const short* y, const float* ori, int des_len, const float* des,
                   float* codewords, int M, int dsub, int K, float* out)
{
    LatentTextureTemplate t(n, x, y, ori, des_len, des);
    t.compute_dist_to_codewords(codewords, M, dsub, K);
    memcpy(out, t.m_dist_codewords, sizeof(float) * (size_t)n * M * K);
    delete[] t.m_dist_codewords;   // the reference class never frees it
    t.m_dist_codewords = NULL;
}

// Rolled texture template construction exactly as matcher.cpp:977 calls it: `des` is the float-typed
// buffer the file bytes were read into.  Returns codes [n][des_len] and the converted points.
void ref_rolled_texture(int n, const short* x, const short* y, const float* ori, int des_len, const float* des,
                        uint8_t* codes_out, int* x_out, int* y_out, float* ori_out)
{
    RolledTextureTemplatePQ t(n, x, y, ori, des_len, des);
    memcpy(codes_out, t.m_desPQ, (size_t)n * des_len);
    for (int i = 0; i < n; ++i) { x_out[i] = t.m_minutiae[i].x; y_out[i] = t.m_minutiae[i].y; ori_out[i] = t.m_minutiae[i].ori; }
}

// Minutiae template construction as matcher.cpp:854 / :955 call it.
void ref_minutiae_template(int n, const short* x, const short* y, const float* ori, int des_len, const float* des,
                           float* des_out, int* x_out, int* y_out, float* ori_out)
{
    float oimg[1] = {0};
    MinutiaeTemplate t(n, x, y, ori, des_len, des, 1, 1, oimg);
    memcpy(des_out, t.m_des, sizeof(float) * (size_t)n * des_len);
    for (int i = 0; i < n; ++i) { x_out[i] = t.m_minutiae[i].x; y_out[i] = t.m_minutiae[i].y; ori_out[i] = t.m_minutiae[i].ori; }
}

double ref_pi(void) { return PI; }

// main.cpp:41-44 + :50/:60/:71: `ifstream i(path); json config; i >> config; string v = config[key];`
// returns 1 and the value, 0 when the key is absent, -1 when the file does not parse or the value is not a string
int ref_config_get(const char* path, const char* key, char* out, int cap)
{
    out[0] = 0;
    try {
        std::ifstream i(path);
        nlohmann::json config;
        i >> config;
        if (!config.contains(key)) return 0;
        const std::string v = config[key];
        strncpy(out, v.c_str(), (size_t)cap - 1); out[cap - 1] = 0;
        return 1;
    } catch (...) { return -1; }
}

// matching/argparser.h through the reference class: returns cmdOptionExists(opt); out = getCmdOption(opt)
int ref_arg(int argc, char** argv, const char* opt, char* out, int cap)
{
    ArgParser a(argc, argv);
    const std::string& v = a.getCmdOption(opt);
    strncpy(out, v.c_str(), (size_t)cap - 1); out[cap - 1] = 0;
    return a.cmdOptionExists(opt) ? 1 : 0;
}

}  // extern "C"
