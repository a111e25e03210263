// adc_direct.hip — the DIRECT exact ADC kernels (adc_variant 0-3, 6, 7) and their tile LUT: S4 (per-query PQ look-up table) and S5+S6 (ADC similarity + per-row
// max / first arg-max) evaluated entry by entry, with no bound pass.  EXPERIMENTAL / reference kernels: rounds 1-2 shipped them; since round 3 the product path is
// adc_variant 9 (adc_mfma.hip + adc_refine.hip) with adc_variant 8 (adc.hip, the LDS-table design of north_star) as the selectable alternative.  These kernels are built
// only into libafis_hip_test.so (-DAFIS_EXPERIMENTAL_KERNELS), where the parity tests use them as a second, independent witness of the row maxima.
//
//
// Reference: LatentTextureTemplate::compute_dist_to_codewords (matching/include.h:327-359) and
// Matcher::One2One_texture_matching method 1 + row arg-max (matching/matcher.cpp:563-595, :723-735).
//
// S5 is fp32 add/sub only; with the reference's 4-accumulator order kept and FMA contraction off
// (-ffp-contract=off) the results are bit-identical to the CPU.
//
// Work decomposition (MI355X): one workgroup keeps the LUT of kTileRows = 8 latent texture rows in LDS
// (8 x 16 KB = 128 KB of the CU's 160 KB) and streams a chunk of gallery templates through it; every wave of
// the workgroup owns whole gallery templates (lane <-> rolled texture point), so the per-row (max, argmax)
// reduction is intra-wave only.  Blocks that share a gallery chunk are consecutive on one XCD (block b runs on
// XCD b % 8) so the chunk's PQ codes are fetched from HBM once per XCD and re-read from that XCD's L2.
#include "afis_device.h"
#include "adc_common.h"
#include <type_traits>

namespace afis {

// ---------------------------------------------------------------------------------------------------------------
// S4.  lut[i][m][k] = sum_{d<6} (des[i][6m+d] - cw[m][k][d])^2, d ascending, product and sum rounded separately.
// Tile layouts (float index inside the 32768-float tile of rows r = 0..7, rq = r/4, r4 = r%4):
//   variant 0 : ((rq*16 + m)*256 + k)*4 + r4
//   variant 1 : (((mg*256 + k)*4 + c)*2 + rq)*4 + r4   with m = 4*mg + c   (chain-major: bank slot depends on (c,rq))
//   variant 4 : ((k*2 + (mg&1))*16 + c*4 + rq*2 + (mg>>1))*4 + r4   — the 16-byte bank slot (float4 index mod 16) is
//               (chain c, row-quad rq, mg>>1); the entry of m = 0 holds lut - 6 (see k_adc_rowmax_cf)
//   variant 6 : ((mg&1)*4096 + k*16 + c*4 + rq*2 + (mg>>1))*4 + r4   — same bank slots; the byte address is
//               (mg&1) << 16 | code << 8 | slot << 4, i.e. one v_perm_b32 of the code word (see k_adc_rowmax_cf)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lut_build(QueryDev q, const float* __restrict__ codewords, float* __restrict__ lut_tiles, int variant)
{
    const int tile = blockIdx.x >> 4;                       // 16 blocks of 256 threads per tile = 4096 (m,k) entries
    const int mk = ((blockIdx.x & 15) << 8) | threadIdx.x;
    const int m = mk >> 8, k = mk & 255;
    int qi = 0;
    while (qi + 1 < q.nq && tile >= q.tile_off[qi + 1]) ++qi;
    const int row0 = (tile - q.tile_off[qi]) * kTileRows;
    const int base = q.lt_off[qi], n = q.lt_off[qi + 1] - base;
    float cw6[kDsub];
#pragma unroll
    for (int d = 0; d < kDsub; ++d) cw6[d] = codewords[(m * kK + k) * kDsub + d];
    float v[kTileRows];
#pragma unroll
    for (int r = 0; r < kTileRows; ++r) {
        int row = row0 + r; if (row >= n) row = n - 1;       // padding rows duplicate the last row; never read back
        const float* d6 = q.lt_des + (size_t)(base + row) * kDes + m * kDsub;
        float des6[kDsub];
#pragma unroll
        for (int d = 0; d < kDsub; ++d) des6[d] = d6[d];
        v[r] = lut_entry(des6, cw6);
    }
    float4* t4 = reinterpret_cast<float4*>(lut_tiles + (size_t)tile * kTileFloats);
    if (variant >= 4) {
        const int mg = m >> 2, c = m & 3;
        const float six = m == 0 ? 6.0f : 0.0f;                 // -(l0 - 6) == 6 - l0 exactly
        const int e = (mg & 1) * 4096 + k * 16 + c * 4 + (mg >> 1);
        t4[e + 0] = make_float4(v[0] - six, v[1] - six, v[2] - six, v[3] - six);
        t4[e + 2] = make_float4(v[4] - six, v[5] - six, v[6] - six, v[7] - six);
    } else if ((variant & 1) == 0) {
        t4[(0 * 16 + m) * 256 + k] = make_float4(v[0], v[1], v[2], v[3]);
        t4[(1 * 16 + m) * 256 + k] = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        const int mg = m >> 2, c = m & 3;
        t4[((mg * 256 + k) * 4 + c) * 2 + 0] = make_float4(v[0], v[1], v[2], v[3]);
        t4[((mg * 256 + k) * 4 + c) * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
}

hipError_t launch_lut_build(const QueryDev& q, const float* codewords, float* lut_tiles, int variant, hipStream_t stream)
{
    if (q.n_tiles <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_lut_build, dim3(q.n_tiles * 16), dim3(256), 0, stream, q, codewords, lut_tiles, variant);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// S5 + S6
// ---------------------------------------------------------------------------------------------------------------
// workgroup size is a template parameter: 512 threads = 2 waves per SIMD (<= 256 VGPRs), 1024 = 4 waves per SIMD (<= 128 VGPRs)

template <int VARIANT, int kAdcThreads>
__global__ __launch_bounds__(kAdcThreads) void k_adc_rowmax(QueryDev q, GalleryDev g, const float* __restrict__ lut_tiles,
                                                            int chunk, int n_chunks, float* __restrict__ rm_val, int32_t* __restrict__ rm_arg)
{
    __shared__ float4 s_lut[kTileFloats / 4];                 // 128 KB

    // XCD-aware mapping: blocks with the same (b % 8) run on one XCD; walk all LUT tiles of one gallery chunk
    // back-to-back there so the chunk's codes stay in that XCD's L2.
    const int b = blockIdx.x, xcd = b & 7, seq = b >> 3;
    const int tile = seq % q.n_tiles;
    const int chunk_id = (seq / q.n_tiles) * 8 + xcd;
    if (chunk_id >= n_chunks) return;

    int qi = 0;
    while (qi + 1 < q.nq && tile >= q.tile_off[qi + 1]) ++qi;
    const int row0 = (tile - q.tile_off[qi]) * kTileRows;
    const int n_lt = q.lt_off[qi + 1] - q.lt_off[qi];

    {   // stage the LUT tile: 128 KB, 16-byte coalesced
        const float4* src = reinterpret_cast<const float4*>(lut_tiles + (size_t)tile * kTileFloats);
        for (int i = threadIdx.x; i < kTileFloats / 4; i += kAdcThreads) s_lut[i] = src[i];
    }
    __syncthreads();

    constexpr int kAdcWaves = kAdcThreads / 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g_lo = chunk_id * chunk;
    const int g_hi = min(g.G, g_lo + chunk);

    // variant 1: per-lane phase.  Lane phase (pc, pr) rotates which (chain, row-quad) the lane touches at each
    // unrolled step, so the 16 lanes a ds_read_b128 services together spread over 8 distinct bank-slot classes.
    const int pc = (lane >> 1) & 3, pr = lane & 1;
    int sh[4], off8[8];
    float init[4];
    if (VARIANT == 1) {
        const int perm[4] = {0, 2, 1, 3};                      // physical slot order (d1,d3,d2,d4): the final
#pragma unroll                                                 // (P0+P2)+(P1+P3) is then rotation-invariant
        for (int c = 0; c < 4; ++c) {
            const int chain = perm[(c + pc) & 3];
            sh[c] = 8 * chain;
            init[c] = chain == 0 ? 6.0f : 0.0f;
#pragma unroll
            for (int rq = 0; rq < 2; ++rq) off8[c * 2 + rq] = chain * 2 + (rq ^ pr);
        }
    }

    for (int gi = g_lo + wave; gi < g_hi; gi += kAdcWaves) {
        const int p0 = g.tex_off[gi], n_pts = g.tex_off[gi + 1] - p0;
        if (n_pts <= 0) continue;
        float best[kTileRows];
        int bidx[kTileRows];
#pragma unroll
        for (int r = 0; r < kTileRows; ++r) { best[r] = -INFINITY; bidx[r] = 0x7fffffff; }

        for (int p = lane; p < n_pts; p += 64) {
            const uint4 cw = g.tex_codes[p0 + p];
            const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
            float acc[4][kTileRows];
            if (VARIANT == 0) {
#pragma unroll
                for (int r = 0; r < kTileRows; ++r) { acc[0][r] = 6.0f; acc[1][r] = 0.0f; acc[2][r] = 0.0f; acc[3][r] = 0.0f; }  // matcher.cpp:571-574
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {              // chain c sees m = c, c+4, c+8, c+12 in this order (matcher.cpp:577-591)
                        const int m = mg * 4 + c;
                        const uint32_t code = (w[mg] >> (8 * c)) & 255u;
                        const float4 a = s_lut[(0 * 16 + m) * 256 + code];
                        const float4 bb = s_lut[(1 * 16 + m) * 256 + code];
                        acc[c][0] -= a.x; acc[c][1] -= a.y; acc[c][2] -= a.z; acc[c][3] -= a.w;
                        acc[c][4] -= bb.x; acc[c][5] -= bb.y; acc[c][6] -= bb.z; acc[c][7] -= bb.w;
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < kTileRows; ++r) acc[c][r] = init[c];
#pragma unroll
                for (int mg = 0; mg < 4; ++mg) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t code = __builtin_amdgcn_ubfe(w[mg], sh[c], 8);
                        const int e = mg * 2048 + (int)code * 8;
                        const float4 a = s_lut[e + off8[c * 2 + 0]];
                        const float4 bb = s_lut[e + off8[c * 2 + 1]];
                        acc[c][0] -= a.x; acc[c][1] -= a.y; acc[c][2] -= a.z; acc[c][3] -= a.w;
                        acc[c][4] -= bb.x; acc[c][5] -= bb.y; acc[c][6] -= bb.z; acc[c][7] -= bb.w;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < kTileRows; ++r) {
                float s;
                if (VARIANT == 0) s = (acc[0][r] + acc[1][r]) + (acc[2][r] + acc[3][r]);                         // matcher.cpp:592
                else s = (acc[0][r] + acc[2][r]) + (acc[1][r] + acc[3][r]);   // physical order (d1,d3,d2,d4) rotated: same two pair sums
                if (s > best[r]) { best[r] = s; bidx[r] = p; }
            }
        }
        if (VARIANT == 1 && pr) {                              // this lane kept rows 4..7 in slots 0..3
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float tv = best[r]; best[r] = best[r + 4]; best[r + 4] = tv;
                int ti = bidx[r]; bidx[r] = bidx[r + 4]; bidx[r + 4] = ti;
            }
        }
        float outv = 0.f; int outi = 0;
#pragma unroll
        for (int r = 0; r < kTileRows; ++r) {
            wave_argmax(best[r], bidx[r]);
            if (lane == r) { outv = best[r]; outi = bidx[r]; }
        }
        if (lane < kTileRows && row0 + lane < n_lt) {
            const size_t o = ((size_t)qi * g.G + gi) * q.lt_pad + row0 + lane;
            rm_val[o] = outv;
            rm_arg[o] = outi == 0x7fffffff ? 0 : outi;     // no similarity ever exceeded -inf (an inf / NaN / overflowing latent row): the first point, as std::max_element
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Conflict-free variants (6 = 512 threads, 7 = 1024 threads).  A ds_read_b128 is serviced 16 lanes at a time, each lane on
// one of 16 bank slots of 16 bytes.  Here the 16 lanes of a group are 16 different CLASSES (a = lane & 15): chain rotation
// pc (4) x row-quad swap pr (2) x a half-period shift pm (2).  At every read instruction lane class (pc, pr, pm) touches chain
// perm[(c+pc)&3], row quad r^pr and sub-quantizer group mg = (j + 2*pm) & 3, and the LUT tile stores (chain, row-quad, mg>>1)
// in the slot index, so the 16 lanes of a group always hit 16 distinct slots whatever their PQ codes are: zero bank conflicts
// by construction (SQ_LDS_BANK_CONFLICT = 0 in profiles/).
//   * lanes with pm = 1 run half a period late: steps j = 0,1 finish the PREVIOUS block's point (mg = 2,3), steps j = 2,3 start
//     the current one (mg = 0,1).  GalleryDev::tex_codes_cf is laid out for exactly this: entry (block k, lane l) holds, for a
//     late lane, the code words of mg 2,3 of point (k-1)*64+l and of mg 0,1 of point k*64+l, so every lane just reads its entry
//     (one extra drain block per template, zero padded; no bounds checks, no selects).
//   * a chain is restarted with x = fma(x, keep, -v), keep = 0 for the lanes that start a point at this step and 1 for the
//     rest; fma(x, 1, -v) == x - v and fma(x, 0, -v) == 0 - v, and the tile holds l0 - 6 for m = 0, so the first chain
//     starts at 6 - l0 exactly as matcher.cpp:571-580.  Each chain still sees m = c, c+4, c+8, c+12 in this order.
//   * the finished sums (d1+d2)+(d3+d4) of the late lanes (after j = 1) and of the on-time lanes (after j = 3) land in the same
//     registers, so ONE first-maximum update per block serves all 64 lanes.
// The loop is VALU-issue bound (packed adds cost 5.0 cycles per SIMD, 4-byte encodings 2.8, VOP3 encodings 4.3 — tools/ubench):
//   * LUT byte address = (mg&1) << 16 | code << 8 | slot << 4 is assembled by ONE v_perm_b32 from the packed code word and a
//     per-lane constant (byte 0 = slot << 4, byte 3 = 1); the other row quad is a0 ^ 32.
//   * the running first maximum is updated under the EXEC mask of the lanes that have just finished a point, so the compare
//     writes VCC and the two selects use the short encoding.
//   * the per-template reduction over the 64 lanes is TRANSPOSED: at the first three butterfly stages a lane keeps half of its
//     rows and hands the other half to its partner (8 -> 4 -> 2 -> 1 rows), so 11 exchanges replace 48; stages within a row of
//     16 lanes are DPP moves (quad_perm, row_ror), only the last two cross 16-lane rows through ds_bpermute.  Stage one needs
//     no selects because partner lanes (pr = lane & 1) already hold their row quads in swapped slots.
//   The reduction is over the total order (value descending, point index ascending), so its result does not depend on the tree.
// ---------------------------------------------------------------------------------------------------------------
// if (s > best) { best = s; idx = p; } as compare-to-VCC + two short-encoded selects (the compiler's choice, compares into SGPR
// pairs + VOP3 selects, costs 4.3 + 2 x 4.3 issue cycles per row instead of 3 x 2.8)
__device__ __forceinline__ void first_max_update(float& best, int& idx, float s, int p)
{
    asm("v_cmp_gt_f32 vcc, %2, %0\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_cndmask_b32 %1, %1, %3, vcc"
        : "+v"(best), "+v"(idx) : "v"(s), "v"(p) : "vcc");
}

template <int kAdcThreads>
__global__ __launch_bounds__(kAdcThreads) void k_adc_rowmax_cf(QueryDev q, GalleryDev g, const float* __restrict__ lut_tiles,
                                                                int chunk, int n_chunks, float* __restrict__ rm_val, int32_t* __restrict__ rm_arg)
{
    __shared__ float4 s_lut[kTileFloats / 4];                 // 128 KB
    __shared__ int s_next;                                    // next unclaimed gallery template of the chunk
    const int b = blockIdx.x, xcd = b & 7, seq = b >> 3;
    const int tile = seq % q.n_tiles;
    const int chunk_id = (seq / q.n_tiles) * 8 + xcd;
    if (chunk_id >= n_chunks) return;
    int qi = 0;
    while (qi + 1 < q.nq && tile >= q.tile_off[qi + 1]) ++qi;
    const int row0 = (tile - q.tile_off[qi]) * kTileRows;
    const int n_lt = q.lt_off[qi + 1] - q.lt_off[qi];
    {
        const float4* src = reinterpret_cast<const float4*>(lut_tiles + (size_t)tile * kTileFloats);
        for (int i = threadIdx.x; i < kTileFloats / 4; i += kAdcThreads) s_lut[i] = src[i];
        if (threadIdx.x == 0) s_next = 0;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int g_lo = chunk_id * chunk;
    const int g_hi = min(g.G, g_lo + chunk);

    const int a = lane & 15, pr = a & 1, pc = (a >> 1) & 3, pm = a >> 3;
    const bool late = pm != 0;
    uint32_t so[4][2];                                         // byte 0: slot << 4 of (physical chain slot c, half hi), byte 3: 1
    {
        const int perm[4] = {0, 2, 1, 3};                      // physical order (d1,d3,d2,d4): (P0+P2)+(P1+P3) is rotation-invariant
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) so[c][hi] = 0x01000000u | (uint32_t)((perm[(c + pc) & 3] * 4 + pr * 2 + (hi ^ pm)) * 16);
    }
    const float keep0 = late ? 1.0f : 0.0f;                    // step j = 0 restarts the chains of the on-time lanes
    const float keep2 = late ? 0.0f : 1.0f;                    // step j = 2 restarts the chains of the late lanes
    const char* lut_b = reinterpret_cast<const char*>(s_lut);
    const int lane_pt = lane - 64 * pm;

    // templates differ in size (600..1000 points): waves claim them one at a time, so no wave idles at the end of the chunk.
    // The claim and the offsets of the NEXT template are fetched while the current one is processed, and its first code entry
    // during the current one's drain block, so a wave never waits for global memory between templates.
    auto claim = [&]() -> int {
        int c = 0;
        if (lane == 0) c = atomicAdd(&s_next, 1);
        return g_lo + __builtin_amdgcn_readfirstlane(c);
    };
    auto stream_of = [&](int gidx, int& n, int& cf_blk) {
        n = 0; cf_blk = 0;
        if (gidx < g_hi) { n = g.tex_off[gidx + 1] - g.tex_off[gidx]; cf_blk = g.tex_cf_blk[gidx]; }
    };
    int gi = claim(), n_pts, cf_blk;
    stream_of(gi, n_pts, cf_blk);
    uint4 cw_next = make_uint4(0, 0, 0, 0);
    if (n_pts > 0) cw_next = g.tex_codes_cf[(size_t)cf_blk * 64 + lane];
    while (gi < g_hi) {
        const int gi_cur = gi, n_blocks = (n_pts + 63) >> 6, n_cur = n_pts;
        const uint4* cfp = g.tex_codes_cf + ((size_t)cf_blk * 64 + lane);
        gi = claim();
        stream_of(gi, n_pts, cf_blk);                          // the next template
        const uint4* cfp_next = g.tex_codes_cf + ((size_t)cf_blk * 64 + lane);
        if (n_cur <= 0) {                                      // empty texture template: nothing to write (the scorer never reads it)
            if (n_pts > 0) cw_next = cfp_next[0];
            continue;
        }
        float best[kTileRows]; int bidx[kTileRows];
#pragma unroll
        for (int r = 0; r < kTileRows; ++r) { best[r] = -INFINITY; bidx[r] = 0x7fffffff; }
        v2f P[4][2][2];                                        // [physical chain][row-quad slot][row pair]: explicit packed fp32
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) P[c][r][i] = v2f{0.0f, 0.0f};
        v2f S[2][2];                                           // finished sums (d1+d2)+(d3+d4), matcher.cpp:592
        auto sums = [&]() {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 2; ++i) S[r][i] = (P[0][r][i] + P[2][r][i]) + (P[1][r][i] + P[3][r][i]);
        };

        // one block = 64 points x 8 rows.  kSteps = 4 for a full block; the drain block after the last one runs only the two
        // steps the late lanes still need (kSteps = 2).
        auto block = [&](int blk, auto steps_tag, const uint4* prefetch, bool do_prefetch) {
            constexpr int kSteps = decltype(steps_tag)::value;
            const uint4 cw = cw_next;
            if (do_prefetch) cw_next = *prefetch;              // the next block's codes, or the next template's first entry
            const uint32_t w[4] = {cw.x, cw.y, cw.z, cw.w};
            constexpr int kGroup = kAdcThreads >= 1024 ? 1 : (kSteps < 4 ? kSteps : 4);
#pragma unroll
            for (int jg = 0; jg < kSteps; jg += kGroup) {
                float4 v[kGroup][4][2];
#pragma unroll
                for (int jj = 0; jj < kGroup; ++jj)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int j = jg + jj;
                        // bytes of the address: [slot << 4 (so byte 0)] [code (w byte c)] [(j&1) ? 1 (so byte 3) : 0] [0]
                        const uint32_t sel = 0x0c000004u | (uint32_t)(c << 8) | ((j & 1) ? 0x00070000u : 0x000c0000u);
                        const uint32_t a0 = __builtin_amdgcn_perm(so[c][j >> 1], w[j], sel);
                        const uint32_t a1 = a0 ^ 32u;                                        // the other row quad: slot bit 1
                        v[jj][c][0] = *reinterpret_cast<const float4*>(lut_b + a0);
                        v[jj][c][1] = *reinterpret_cast<const float4*>(lut_b + a1);
                    }
#pragma unroll
                for (int jj = 0; jj < kGroup; ++jj) {
                    const int j = jg + jj;
                    const v2f keep = j == 0 ? v2f{keep0, keep0} : v2f{keep2, keep2};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const float4 x = v[jj][c][r];
                            const v2f lo{x.x, x.y}, hi{x.z, x.w};
                            if (j == 0 || j == 2) {            // steps that restart the chains of one half of the lanes
                                P[c][r][0] = __builtin_elementwise_fma(P[c][r][0], keep, -lo);
                                P[c][r][1] = __builtin_elementwise_fma(P[c][r][1], keep, -hi);
                            } else {
                                P[c][r][0] -= lo; P[c][r][1] -= hi;
                            }
                        }
                    if (j == 1) sums();                          // the late lanes have just finished the previous block's point
                    if (j == kSteps - 1) {
                        if (kSteps == 4 && !late) sums();        // the on-time lanes have finished this block's point
                        const int p = blk * 64 + lane_pt;        // the point each lane has finished: (blk - pm) * 64 + lane
                        if ((unsigned)p < (unsigned)n_cur) {     // (in the drain block this is false for every on-time lane)
#pragma unroll
                            for (int r = 0; r < 2; ++r)
#pragma unroll
                                for (int i = 0; i < 2; ++i) {
                                    first_max_update(best[r * 4 + 2 * i], bidx[r * 4 + 2 * i], S[r][i].x, p);
                                    first_max_update(best[r * 4 + 2 * i + 1], bidx[r * 4 + 2 * i + 1], S[r][i].y, p);
                                }
                        }
                    }
                }
            }
        };
        for (int blk = 0; blk < n_blocks; ++blk) block(blk, std::integral_constant<int, 4>{}, cfp + (size_t)(blk + 1) * 64, true);
        block(n_blocks, std::integral_constant<int, 2>{}, cfp_next, n_pts > 0);

        // ---- transposed first-maximum reduction over the wave ------------------------------------------------------
        // physical slot k of a lane holds row (pr ^ (k >> 2)) * 4 + (k & 3)
        constexpr int kXor1 = 0xB1, kXor2 = 0x4E, kRor4 = 0x124, kRor8 = 0x128;   // quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8
#pragma unroll
        for (int k = 0; k < 4; ++k)                            // lane ^ 1 keeps the other row quad in ITS slots 0..3
            argmax_merge(best[k], bidx[k], dpp_f<kXor1>(best[k + 4]), dpp_i<kXor1>(bidx[k + 4]));
        const bool b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
        float wv[2]; int wi[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {                          // lane ^ 2: keep slots {0,1} (b1 = 0) or {2,3} (b1 = 1)
            const float sv = b1 ? best[k] : best[k + 2]; const int si = b1 ? bidx[k] : bidx[k + 2];
            wv[k] = b1 ? best[k + 2] : best[k]; wi[k] = b1 ? bidx[k + 2] : bidx[k];
            argmax_merge(wv[k], wi[k], dpp_f<kXor2>(sv), dpp_i<kXor2>(si));
        }
        float rv; int ri;
        {                                                      // lanes +-4 in the row of 16 have the other b2: keep w[b2]
            const float sv = b2 ? wv[0] : wv[1]; const int si = b2 ? wi[0] : wi[1];
            rv = b2 ? wv[1] : wv[0]; ri = b2 ? wi[1] : wi[0];
            argmax_merge(rv, ri, dpp_f<kRor4>(sv), dpp_i<kRor4>(si));
        }
        argmax_merge(rv, ri, dpp_f<kRor8>(rv), dpp_i<kRor8>(ri));          // lane ^ 8: same row, the other two quads
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            const float ov = __shfl_xor(rv, off); const int oi = __shfl_xor(ri, off);
            argmax_merge(rv, ri, ov, oi);
        }
        const int row = row0 + (lane & 1) * 4 + (lane & 2) + ((lane >> 2) & 1);
        if (lane < kTileRows && row < n_lt) {
            const size_t o = ((size_t)qi * g.G + gi_cur) * q.lt_pad + row;
            rm_val[o] = rv;
            rm_arg[o] = ri == 0x7fffffff ? 0 : ri;        // no similarity ever exceeded -inf (an inf / NaN / overflowing latent row): the first point, as std::max_element
        }
    }
}

hipError_t launch_adc_rowmax(const QueryDev& q, const GalleryDev& g, const float* lut_tiles, int chunk, int variant,
                             float* rm_val, int32_t* rm_arg, hipStream_t stream)
{
    if (q.n_tiles <= 0 || g.G <= 0) return hipSuccess;
    const int n_chunks = (g.G + chunk - 1) / chunk;
    const long long blocks = (long long)((n_chunks + 7) / 8) * 8 * q.n_tiles;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const int threads = (variant == 2 || variant == 3) ? 1024 : 512;   // variants 2,3 = variants 0,1 with 1024-thread workgroups
    switch (variant) {
    case 0: hipLaunchKernelGGL((k_adc_rowmax<0, 512>), dim3((unsigned)blocks), dim3(threads), 0, stream, q, g, lut_tiles, chunk, n_chunks, rm_val, rm_arg); break;
    case 1: hipLaunchKernelGGL((k_adc_rowmax<1, 512>), dim3((unsigned)blocks), dim3(threads), 0, stream, q, g, lut_tiles, chunk, n_chunks, rm_val, rm_arg); break;
    case 2: hipLaunchKernelGGL((k_adc_rowmax<0, 1024>), dim3((unsigned)blocks), dim3(threads), 0, stream, q, g, lut_tiles, chunk, n_chunks, rm_val, rm_arg); break;
    case 6: hipLaunchKernelGGL((k_adc_rowmax_cf<512>), dim3((unsigned)blocks), dim3(512), 0, stream, q, g, lut_tiles, chunk, n_chunks, rm_val, rm_arg); break;
    case 7: hipLaunchKernelGGL((k_adc_rowmax_cf<1024>), dim3((unsigned)blocks), dim3(1024), 0, stream, q, g, lut_tiles, chunk, n_chunks, rm_val, rm_arg); break;
    default: hipLaunchKernelGGL((k_adc_rowmax<1, 1024>), dim3((unsigned)blocks), dim3(threads), 0, stream, q, g, lut_tiles, chunk, n_chunks, rm_val, rm_arg); break;
    }
    return hipGetLastError();
}

// Lane-ordered code stream of the direct conflict-free kernel (k_adc_rowmax_cf, variants 6 / 7), laid out on first use: template t owns
// (blocks + 1) x 64 entries of 16 bytes starting at block cf_blk[t]; entry (block k, lane l) belongs to lane class a = l & 15
// (pc = (a >> 1) & 3, pm = a >> 3).  Dword d carries sub-quantizer group mg = (d + 2 pm) & 3, byte c of it chain perm[(c + pc) & 3]; lanes
// with pm = 1 run half a period late, so their dwords 0, 1 (mg 2, 3) come from point (k - 1) * 64 + l and their dwords 2, 3 (mg 0, 1) from point
// k * 64 + l.  Entries without a point are zero.  grid = G, block = 64.
__global__ __launch_bounds__(64) void k_codes_cf(GalleryDev g, uint4* __restrict__ out)
{
    const int t = blockIdx.x, l = threadIdx.x;
    const int p0 = g.tex_off[t], n = g.tex_off[t + 1] - p0;
    if (n <= 0) return;
    const int blocks = (n + 63) >> 6;
    const int a = l & 15, pc = (a >> 1) & 3, pm = a >> 3;
    const int perm[4] = {0, 2, 1, 3};
    for (int k = 0; k <= blocks; ++k) {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int mg = (d + 2 * pm) & 3;
            const int pt = (pm && d < 2 ? k - 1 : k) * 64 + l;
            if (pt < 0 || pt >= n) continue;
            const uint4 c = g.tex_codes[p0 + pt];
            const uint32_t src = mg == 0 ? c.x : mg == 1 ? c.y : mg == 2 ? c.z : c.w;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) w[d] |= ((src >> (8 * perm[(cc + pc) & 3])) & 255u) << (8 * cc);
        }
        out[((size_t)g.tex_cf_blk[t] + k) * 64 + l] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

hipError_t launch_codes_cf(const GalleryDev& g, void* out, hipStream_t stream)
{
    if (g.G <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_codes_cf, dim3(g.G), dim3(64), 0, stream, g, (uint4*)out);
    return hipGetLastError();
}

}  // namespace afis
