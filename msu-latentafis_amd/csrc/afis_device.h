// afis_device.h — device-side data layout and kernel launchers shared by the HIP translation units.
// gfx950 (MI355X / CDNA4) only; wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Experiment knobs (environment variables that re-create an earlier round's schedule, or skip work for a timing-only run) exist in libafis_hip_test.so only
// (-DAFIS_EXPERIMENTAL_KERNELS): the product library does not read them, and their names do not occur in it (tests/test_host.py compares `strings libafis_hip.so`
// with the table of operational variables in INTEGRATION.md section E).
#ifdef AFIS_EXPERIMENTAL_KERNELS
#define AFIS_EXPERIMENT_ENV(name) (getenv(name))
#else
#define AFIS_EXPERIMENT_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace afis {

constexpr int kM = 16;            // PQ sub-quantizers         (codebook header, matcher.cpp:74)
constexpr int kK = 256;           // codewords per sub-quantizer
constexpr int kDsub = 6;          // dims per sub-quantizer
constexpr int kDes = 96;          // descriptor length
constexpr int kTexMax = 1000;     // MaxNRolledMinu / MaxNLatentMinu, matcher.h:31-32
constexpr int kTopTex = 200;      // Matcher::N, matcher.cpp:33
constexpr int kTopMinu = 120;     // topN, matcher.cpp:479
constexpr int kDistN = 50;        // dist_N, matcher.cpp:45
constexpr int kTileRows = 8;      // latent texture rows whose LUT one workgroup keeps in LDS (8*16 KB = 128 KB)
constexpr int kTileFloats = kTileRows * kM * kK;   // 32768 floats per LUT tile

// Rolled gallery shard, SoA in HBM.  Points of all templates are concatenated; *_off are CSR offsets.
struct GalleryDev {
    int32_t G = 0;
    const int32_t* minu_off = nullptr;   // [G+1]
    const short2*  minu_xy = nullptr;    // [NM]   pixel coords
    const float*   minu_ori = nullptr;   // [NM]
    const float*   minu_des = nullptr;   // [NM][96]
    const float4*  minu_frag = nullptr;  // the same descriptors as MFMA operand fragments: per template ceil(n/16) tiles of 16 descriptors,
                                         // tile = [v < 6][lane < 64] float4 with lane l, component c = des[16*tile + (l&15)][4*(4v + c) + (l>>4)]
                                         // (zero rows pad the last tile): a wave fetches a fragment with six fully coalesced 1 KB loads
    const int32_t* minu_tile_off = nullptr;  // [G+1] first tile of each template in minu_frag
    const int32_t* tex_off = nullptr;    // [G+1]  (counts already clamped to 1000)
    const short2*  tex_xy = nullptr;     // [NT]   block coords
    const float*   tex_ori = nullptr;    // [NT]
    const uint4*   tex_codes = nullptr;  // [NT]   16 PQ code bytes per point, byte m = sub-quantizer m
    const uint4*   tex_codes_cf = nullptr;  // the same bytes as a per-template stream of (blocks + 1) x 64 lane entries, permuted per
                                            // lane class and half-period shifted for the conflict-free ADC kernel (adc.hip)
    const int32_t* tex_cf_blk = nullptr;    // [G+1] offset of each template's stream in tex_codes_cf, in 64-entry blocks
    const uint8_t* empty = nullptr;      // [G] 1 = rolled template has neither minutiae nor texture (score -1)
    int32_t* task_ctr = nullptr;         // [4] next-task counters of the graph kernels (texture, minutiae), the refine kernel and the candidate kernel: zeroed by their launchers
};

// A group of latents resident on the device (selected minutiae templates 26, 2, 11 + texture template 0).
struct QueryDev {
    int32_t nq = 0;
    const int32_t* lm_off = nullptr;     // [nq*3+1] minutiae points of (query, selected template s); empty range = absent
    const short2*  lm_xy = nullptr;
    const float*   lm_ori = nullptr;
    const float*   lm_des = nullptr;     // [NLM][96]
    const float4*  lm_frag = nullptr;    // latent descriptors as MFMA operand fragments (layout as GalleryDev::minu_frag)
    const int32_t* lm_tile_off = nullptr;   // [nq*3+1]
    const int32_t* lt_off = nullptr;     // [nq+1] texture rows (clamped to 1000)
    const short2*  lt_xy = nullptr;
    const float*   lt_ori = nullptr;
    const float*   lt_des = nullptr;     // [NLT][96]
    const int32_t* tile_off = nullptr;   // [nq+1] LUT tiles (8 rows each) per query
    const int32_t* tile16_off = nullptr; // [nq+1] 16-row tiles of adc_variant 8's quantised table
    int32_t n_tiles16 = 0;
    const int32_t* tex_slot = nullptr;   // [nq] index of the texture score in the reference's score vector (= #latent minutiae templates), -1 = no texture
    const int32_t* status = nullptr;     // [nq] AFIS_QUERY_*
    int32_t n_tiles = 0;
    int32_t lt_pad = 0;                  // row stride of the rowmax buffers (max rows over the group, multiple of 8)
};

// ---- launchers (each enqueues on `stream`, returns hipGetLastError()) ------------------------------------------
// S4: LUT tiles.  variant selects the in-tile layout the ADC kernel of the same variant reads.
#ifdef __HIPCC__
// Correctly rounded fp32 square root for x == 0 or 1 <= x < 2^64 (squared distances): the hardware v_sqrt_f32 is only accurate to
// 1 ulp (and HIP's __fsqrt_rn IS that instruction, not a correctly rounded sqrt), so the result is settled between its two
// neighbours with two exact residuals — the core of LLVM's own correctly rounded sqrtf expansion, without the denormal scaling
// and the inf/nan handling this argument range does not need.  Checked against the definition of round-to-nearest for every
// float in the range by tools/ubench/sqrt_exact.hip.
__device__ __forceinline__ float sqrt_rn_pos(float x)
{
    float r = __builtin_amdgcn_sqrtf(x);
    const float dn = __int_as_float(__float_as_int(r) - 1), up = __int_as_float(__float_as_int(r) + 1);
    const float e_dn = __builtin_fmaf(-dn, r, x), e_up = __builtin_fmaf(-up, r, x);
    r = e_dn <= 0.0f ? dn : r;
    r = e_up > 0.0f ? up : r;
    return r;
}
#endif

#ifdef __HIPCC__
// S4, one table entry: lut[i][m][k] = sum_{d<6} (des[i][6m+d] - cw[m][k][d])^2, d ascending, difference, product and sum rounded separately
// (include.h:327-359; every translation unit is built with -ffp-contract=off).  Shared by the table builders (adc.hip), the PQ encoder and the
// exact recomputation of adc_mfma.hip: one definition, one set of bits.
__device__ __forceinline__ float lut_entry(const float* __restrict__ des6, const float* __restrict__ cw6)
{
    float dist = 0.0f;
#pragma unroll
    for (int d = 0; d < kDsub; ++d) {
        float t = des6[d] - cw6[d];
        float t2 = t * t;
        dist += t2;
    }
    return dist;
}
#endif

hipError_t launch_lut_build(const QueryDev& q, const float* codewords, float* lut_tiles, int variant, hipStream_t stream);
// S5+S6: ADC similarity + per-row (max, first argmax) for queries [q0, q0+nq) against gallery templates.
hipError_t launch_adc_rowmax(const QueryDev& q, const GalleryDev& g, const float* lut_tiles, int chunk, int variant,
                             float* rm_val, int32_t* rm_arg, hipStream_t stream);
// adc_variant 8 (adc.hip): 16-bit fixed-point LUT tiles of 16 rows + per-row (offset, step, margin); the lane-ordered code stream it reads
// (ceil(n/64) blocks per template, first block q_blk[t]); S5+S6 with integer sums as a bound pass.
hipError_t launch_lutq_build(const QueryDev& q, int n_rows_total, const float* codewords, float* row_min, float* row_rng, void* tiles, void* rowc, hipStream_t stream);
hipError_t launch_codes_q(const GalleryDev& g, const int32_t* q_blk, void* out, hipStream_t stream);
// the direct conflict-free kernel's lane-ordered code stream (variants 6 / 7), laid out on the device at first use (g.tex_cf_blk = its block offsets)
hipError_t launch_codes_cf(const GalleryDev& g, void* out, hipStream_t stream);
// lut32 != NULL (adc_variant 8): the quantised pass only bounds the candidates, which are then evaluated exactly from the fp32 table in the
// reference layout [row][16][256] (launch_lut_reference_layout over all latent texture rows of the group): exact results, bit for bit.
hipError_t launch_adc_rowmax_q(const QueryDev& q, const GalleryDev& g, const void* codes_q, const int32_t* q_blk, const void* lutq_tiles, const void* rowc,
                               const float* lut32, int chunk, int share, float* rm_val, int32_t* rm_arg, hipStream_t stream);
// adc_variant 9 (adc_mfma.hip): fp16 matrix-core bound pass + exact recomputation.  launch_mf_codebook: fp16 codebook (16-byte entries) and
// |cw|^2 table, once per context.  launch_mf_tiles: tile-aligned (32 points) codes / point terms / tile directory of the gallery (first use).
// launch_mf_rows: per latent row of a query group the fp16 B fragments and (c, Es, Tg, force).  launch_adc_mfma: the bound pass ->
// rec[template * R_pad + row].  launch_tex_refine: bounds -> the rows that can reach the top 200 -> exact (max, first arg-max).  Two output forms: compact (rm_n != NULL: what a
// search uses) — per pair the evaluated rows side by side in row order, value in rm_cv[pair * lt_pad + slot], (row | point << 16) in rm_arg[pair * lt_pad + slot], their count in rm_n[pair];
// rm_val is not touched — or dense (rm_n == NULL): value and point at rm_val / rm_arg[pair * lt_pad + row], -inf for the other rows (all_rows != 0: every row exactly; the parity taps use it).
hipError_t launch_mf_codebook(const float* codewords, void* cw16, float* cwn, hipStream_t stream);
hipError_t launch_mf_tiles(const GalleryDev& g, const int32_t* t32_blk, const float* cwn, void* codes_p, float* nrm_p, void* tile_meta, hipStream_t stream);
hipError_t launch_mf_rows(const float* lt_des, int n_rows, int n_rb, const float* codewords, const float* cwn, void* bfrag, void* rowk, hipStream_t stream);
hipError_t launch_adc_mfma(const GalleryDev& g, const void* codes_p, const float* nrm_p, const void* tile_meta, const int32_t* tile0, const void* cw16,
                           const void* bfrag, const void* rowk, int n_rows, int n_rb, int R_pad, int chunk, int blocks_per_wave, void* rec,
                           unsigned long long* diag /* NULL or a diagnostics row: clock samples */, hipStream_t stream);
hipError_t launch_tex_refine(const QueryDev& q, const GalleryDev& g, const float* codewords, const void* rec, const void* rowk, int R_pad, int all_rows,
                             float* rm_val, int32_t* rm_arg, unsigned long long* stats, float* rm_cv, int32_t* rm_n, hipStream_t stream);
// one correspondence of a minutiae-template list (S3 output), 8 bytes
struct MinuCand { float sim; short li, ri; };

// ---- shape classes of the fast candidate kernel k_minu_cands_rt<S> (minu.hip), S = 1, 2, 4: workgroups of 256 S threads.  A task (nL latent x nR rolled
// minutiae) fits class S when its similarity matrix fits the class's LDS array with row stride rt_row_stride() and the selection's 32 keys per thread
// cover it: nL <= rt_max_rows(S, nR).  It is done by the smallest class that takes it; tasks no class takes go to the any-shape kernel.
// (matcher.cpp:788-790 lets a template carry 2000 minutiae; extraction_rolled.py:105-108 caps nothing for rolled prints.)
#ifdef __HIPCC__
#define AFIS_HOST_DEVICE __host__ __device__
#else
#define AFIS_HOST_DEVICE
#endif
AFIS_HOST_DEVICE constexpr int rt_class_max_rolled(int S) { return S == 1 ? 128 : S == 2 ? 256 : 512; }
AFIS_HOST_DEVICE constexpr int rt_class_max_latent(int S) { return S == 1 ? 64 : 256; }
// floats of the similarity matrix in LDS: 32 similarities per thread (+ the odd stride's padding column) for S = 1, 2; the large class takes what a CU's 160 KB leave
// beside the other arrays (38 912 floats = 152 KB) and walks the rows beyond its threads' 32 register keys a second time (k_minu_cands_rt: "key blocks")
AFIS_HOST_DEVICE constexpr int rt_class_simi_floats(int S) { return S == 1 ? 8192 : S == 2 ? 8192 * 2 + 256 : 38912; }
AFIS_HOST_DEVICE constexpr int rt_class_keys_per_thread(int S) { return S == 4 ? 64 : 32; }
AFIS_HOST_DEVICE inline int rt_row_stride(int S, int nR) { return S == 1 ? (((nR & 1) || nR == 128) ? nR : nR + 1) : (nR | 1); }
AFIS_HOST_DEVICE inline int rt_max_rows(int S, int nR)            // the largest latent minutiae count class S takes against nR rolled minutiae (0: none); non-decreasing in S
{
    if (nR <= 0 || nR > rt_class_max_rolled(S)) return 0;
    int m = rt_class_keys_per_thread(S) * ((256 * S) / nR);                  // thread = (column, row phase): 32 keys in registers (the large class: and as many again recomputed)
    const int cap = rt_class_simi_floats(S) / rt_row_stride(S, nR);
    m = m < cap ? m : cap;
    return m < rt_class_max_latent(S) ? m : rt_class_max_latent(S);
}
// ---- launch groups (host): the latents of a search are cut into contiguous runs that are launched together (afis_search.cpp::afis_queries_upload).
// launch_group_latents: latents per launch at most when the option is 0 — about five million (latent, rolled) pairs, between 10 and 128 latents.
// launch_group_cuts: the group ends (exclusive) for n = rows_prefix.size() - 1 latents, at most `per` latents each.  by_row_groups (adc_variant 9): the bound pass works in row
// groups of 768 latent texture rows and a launch pays for its last one in full, so the cuts are placed where the total number of row groups is smallest (dynamic programme
// over the cut positions; ties: fewer launches, then the earliest last cut); otherwise runs of `per`.  rows_prefix[i] = latent texture rows of latents 0 .. i-1.  Results never depend on the cuts.
inline long long launch_group_latents(long long G) { if (G < 1) G = 1; const long long w = (5000000 + G / 2) / G; return w < 10 ? 10 : w > 128 ? 128 : w; }
inline void launch_group_cuts(const long long* rows_prefix, int n, int per, bool by_row_groups, int* cuts /* [n] at most */, int* n_cuts)
{
    *n_cuts = 0;
    if (n <= 0) return;
    if (per < 1) per = 1;
    if (!by_row_groups || n == 1) { for (int i = per; i < n; i += per) cuts[(*n_cuts)++] = i; cuts[(*n_cuts)++] = n; return; }
    const long long kInf = 1ll << 60, rg_rows = 768;
    long long* best = new long long[(size_t)n + 1]; int* from = new int[(size_t)n + 1]; int* cnt = new int[(size_t)n + 1];
    best[0] = 0; from[0] = 0; cnt[0] = 0;
    for (int i = 1; i <= n; ++i) {
        best[i] = kInf; from[i] = 0; cnt[i] = 0;
        for (int j = (i - per > 0 ? i - per : 0); j < i; ++j) {
            const long long c = best[j] + (rows_prefix[i] - rows_prefix[j] + rg_rows - 1) / rg_rows;
            if (c < best[i] || (c == best[i] && cnt[j] + 1 < cnt[i])) { best[i] = c; from[i] = j; cnt[i] = cnt[j] + 1; }
        }
    }
    int m = 0;
    for (int i = n; i > 0; i = from[i]) cuts[m++] = i;
    for (int a = 0, b = m - 1; a < b; ++a, --b) { const int t = cuts[a]; cuts[a] = cuts[b]; cuts[b] = t; }
    *n_cuts = m;
    delete[] best; delete[] from; delete[] cnt;
}
// ints of the candidate stage's control buffer: [fallback count | n_tasks fallback task ids | 8 control words | 3 G work-list entries]
inline size_t minu_fb_ints(size_t n_tasks, size_t G) { return 1 + n_tasks + 8 + 3 * G; }
// floats of global scratch one workgroup of k_minu_cands needs: simi[n] | keys[n] | rowsum[2048] | colsum[2048], n = the largest latent x rolled minutiae count of the launch
// (rounded to 64); with option s3_tie_order three more arrays of n words (the index array std::sort permutes and the two pointers' stops: minu.hip)
inline size_t minu_scratch_floats(int max_nL, int max_nR, int s3_tie_order)
{
    const size_t n = ((size_t)(max_nL > 1 ? max_nL : 1) * (size_t)(max_nR > 1 ? max_nR : 1) + 63) / 64 * 64;
    return (s3_tie_order ? 5 : 2) * n + 4096;
}
// One row of 16 unsigned 64-bit diagnostics per launch group, zeroed at the start of a search and read back with its results (afis_timing):
enum { kDiagFallback = 0,       // candidate tasks handed to the any-shape kernel
       kDiagSmall = 1,          // tasks done by k_minu_cands_rt<1>, <2>, <4> (kDiagSmall + class)
       kDiagCandsClk = 4, kDiagCandsWall = 5,      // sampled workgroups of the candidate kernel: shader cycles and 100 MHz ticks they lived
       kDiagBoundClk = 6, kDiagBoundWall = 7,      // the same for the bound pass
       kDiagWords = 16 };
// S7+S8b+S9: texture lists, one wave per (query, gallery template) -> parts[(q*G+g)*4+3]; rm_n != NULL: the compact form above (rm_val unused), otherwise the dense one (rm_cv unused).
// tap_stage (this launcher and launch_graph_minutiae): bits 0-7 = the stage a parity tap stops after (2 = the whole scorer); bit 8 = option ref_tie_order 2 (the <1> instantiation of the
// kernel: equal selectable scores of S8 / S9 in std::sort's order, graph.hip::sort_scores)
hipError_t launch_graph_texture(const QueryDev& q, const GalleryDev& g, const float* table_dist,
                                const float* rm_val, const int32_t* rm_arg, const float* rm_cv, const int32_t* rm_n, float* parts, MinuCand* tap_out, int32_t* tap_n, int tap_stage, hipStream_t stream);
// S1-S3 for the three selected latent minutiae templates: correspondence lists in rank order, cands[task][120], cand_n[task]
// (task = (q*3+s)*G + g).  Pairs of up to 256 latent x 512 rolled minutiae and 38 912 similarities go through the rolled-template-stationary MFMA kernel in one of its
// three shape classes (above); what it cannot take (other shapes, degenerate key distributions) it appends to `fallback` (count, task ids), which the
// generic kernel then works off.  force_generic: a flags word — bit 0: the generic kernel does every task; bit 1: option s3_tie_order (equal norms in std::sort's order where the generic kernel holds the matrix in LDS).  max_nL / max_nR: the longest latent list of the launch and the
// longest rolled minutiae template of the gallery (which classes can have work at all).
hipError_t launch_minu_cands(const QueryDev& q, const GalleryDev& g, float* scratch, size_t scratch_floats_per_wg, int n_wg,
                             int force_generic, MinuCand* cands, int32_t* cand_n, int32_t* fallback /* minu_fb_ints() ints */, int max_nL, int max_nR,
                             unsigned long long* diag /* NULL or a diagnostics row */, hipStream_t stream);
// S8a+S9 on those lists, one wave per list -> parts[(q*G+g)*4+{0,1,2}]
hipError_t launch_graph_minutiae(const QueryDev& q, const GalleryDev& g, const MinuCand* cands, const int32_t* cand_n,
                                 float* parts, short4* corr_out, int32_t* corr_n, MinuCand* tap_out, int32_t* tap_n, int tap_stage, hipStream_t stream, bool join = false);
// S10: fusion -> scores[q*G+g]
hipError_t launch_fuse(const QueryDev& q, const GalleryDev& g, const float* parts, float* scores, hipStream_t stream);

// S11: per-query rank list over the shard's scores [n_q][G] on the device (score descending, index ascending); out_idx[n_q][k] carries
// index_base + local index (-1 / -inf beyond G)
hipError_t launch_topk(const float* scores, int n_q, int G, int k, long long index_base, long long* out_idx, float* out_score, hipStream_t stream);

// optional in-kernel phase timers (build with PHASE_TIMING=1); zeros otherwise
hipError_t launch_pq_encode(const float* des, long long n, const float* codewords, uint8_t* codes, hipStream_t stream);
// descriptors -> MFMA operand fragment tiles on the device (pq_encode.hip); off / tile_off: [n_templates + 1] device arrays
hipError_t launch_fragment_tiles(const float* des, const int32_t* off, const int32_t* tile_off, int n_templates, void* frag, hipStream_t stream);
hipError_t read_phase_cycles(unsigned long long* out32, bool reset);
hipError_t read_graph_phase_cycles(unsigned long long* out16, bool reset);

// debug tap: the angle stage's atan2 on the integer grid [-R, R]^2, out[(dy + R) * (2R + 1) + (dx + R)] (device pointer)
hipError_t launch_debug_atan2_grid(int R, float* out, hipStream_t stream);
hipError_t launch_debug_graph_arith(unsigned long long* cnt8, hipStream_t stream);
// debug tap: LUT in the reference layout [n][16][256]
hipError_t launch_lut_reference_layout(const float* des, int n, const float* codewords, float* out, hipStream_t stream);

}  // namespace afis
