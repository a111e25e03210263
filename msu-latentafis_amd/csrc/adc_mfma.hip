// adc_mfma.hip — S5 + S6 (+ the part of S7 that decides which rows matter), adc_variant 9:
//   a matrix-core BOUND pass over every (latent texture row, rolled texture point) cell, then the reference's own table arithmetic
//   (include.h:327-359, matcher.cpp:563-595, :723-735) on the few cells that can decide a result.  Results are the reference's bits.
//
// Why a contraction is hiding in the table look-ups.  lut[i][m][c] = |a_im - cw_mc|^2 (a = latent descriptor, cw = codeword), so
//   sim(i, j) = 6 - sum_m lut[i][m][code_jm] = (6 - |a_i|^2) + 2 (a_i . b_j - |b_j|^2 / 2),   b_j = the codewords of point j side by side.
// The bracket is a 96-long dot product per cell: 671 x 800 x 96 MACs per pair, which the fp16 matrix cores do in a sixth of the time
// the LDS table look-ups of variant 8 take.  The price is precision: fp16 operands put the computed G(i, j) = a_i~ . b_j~ - |b_j|^2 / 2
// within E_i of the real number — good enough to BOUND (E_i is computed per latent row from the row's own rounding residuals and the
// codebook's, no norm assumption) but not to score.  So:
//   1. k_adc_mfma: per (row, rolled template) the location of the largest G and of everything within T_i = 2 E_i + ... of it — normally
//      one point.  Lane = latent row (the MFMA's N index), registers = rolled points (M index): the maximum over a template's points is
//      lane-local, tracked over two partitions of the lane's values (by accumulator slot, by tile) with v_max3 / v_med3 at 1.25 VALU per
//      value; the intersection of the best slot and the best tile IS the point, and the runners-up of both partitions bound every other
//      candidate (a point within T of the best lies in a slot AND a tile whose maxima are within T of it).
//   2. k_tex_refine: per (latent, rolled) pair the rows that can still reach the top 200 (S7) by their bounds c_i + 2 G +- Es_i, and for
//      those rows the exact similarity of each candidate point — the table entries RECOMPUTED from the fp32 descriptor and the fp32
//      codebook in LDS with lut_entry(), summed in the reference's four chains.  Every other row gets -inf (it cannot be among the 200).
// The rolled points' fp16 reconstructions never exist in memory: a workgroup decodes 128 points at a time from their 16 code bytes through
// the fp16 codebook in LDS (64 KB) straight into MFMA operand layout (another 48 KB of LDS, double buffered); HBM carries 20 bytes per point.
#include "afis_device.h"
#include <algorithm>
#ifndef AFIS_MF_ABLATE
#define AFIS_MF_ABLATE 0
#endif
#ifndef AFIS_MF_PREFETCH
#define AFIS_MF_PREFETCH 1
#endif

namespace afis {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr float kMfNeg = -1.0e30f;          // "no point": the accumulator start of padding points

__device__ __forceinline__ uint32_t f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(uint32_t x) { return __uint_as_float(x); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }   // v_max3_f32
__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// ---------------------------------------------------------------------------------------------------------------------------------
// The bound pass.  grid = row groups (768 latent rows) x gallery chunks, block = 768 = 12 waves, three per SIMD (<= 168 VGPRs).
//   LDS: fp16 codebook (64 KB) + two stages of 6 tiles: per tile 12 operand groups x 32 points x 16 B (group gidx = halves 8 gidx .. 8 gidx + 7
//   of the point's 96, i.e. exactly what lane (point, k half) of MFMA step gidx / 2 wants: reads and writes are conflict free), 32 point terms
//   and the tile's directory entry (template, tile index | last: it travels with the tile — a scalar load in the tile loop would share the operand reads' lgkmcnt).
//   A template owns ceil(n / 32) tiles of the stream (16 padding points per template on average; the pair-of-tiles alignment this replaced had 32: -2 %).
//   Every thread decodes one (point, 4 sub-quantizers) item per stage: 4 codebook reads (12 bytes used of 16), 3 operand-group writes — three
//   groups are exactly four codewords' 24 halves, so no repacking arithmetic at all.
//   Wave w keeps the B fragments of row blocks 2w, 2w + 1 in 48 registers for its whole life and runs every tile through both.
//   D[point][row]: lane = (row = lane & 31, h = lane >> 5), register r = point (r & 3) + 8 (r >> 2) + 4 h of the tile; the C operand of the first
//   MFMA of a tile is the points' term n_j, so a finished accumulator IS G.
// Tracking: a lane sees 16 values per tile and row block.  Two partitions of all the values it sees over a template:  G: 8 slots, slot k =
//   registers k and k + 8 of every tile (running maxima m[k]);  H: 2 groups per tile, registers 0..7 and 8..15 (top three group maxima
//   tb >= ts >= tu, low 6 bits = group id = 2 * tile + register half).  A slot and a group meet in exactly one value.  24 VALU per 16 values.
// Records: rec[template * R_pad + row] = (G of the row's best point, descriptor); the two lane halves h of a row are merged before the store:
//   the PRIMARY half (the one holding the best value): bits 0-5 / 6-11 best / second group (group = 2 * tile + register half), 12-14 / 15-17 best / second
//   slot (slot k = registers k, k + 8), 18 second group within T, 19 second slot within T; 21 = which half is primary;
//   candidates = {groups} x {slots}; register r = slot + 8 * (group & 1), point = 32 * (group >> 1) + (r & 3) + 8 (r >> 2) + 4 h;
//   22 = the OTHER half's best lies within T of the row's: its best cell (bits 23-28 group, 29-31 slot) is a candidate as well;
//   20 "many": a third group or slot within T in the primary half, runners-up within T in the other half too, or a forced row: every point is evaluated.
// What was tried on this kernel and left out (all within 3 % of this form): two waves per SIMD with the tracking of tile
// i - 1 interleaved between the MFMAs of tile i (two accumulator sets); the same with a three-stage LDS ring and operand reads one tile ahead;
// the two waves of a SIMD half a period apart (one in its MFMA burst while the other tracks); the decode spread over the tile steps;
// s_setprio 3 around the MFMA burst; one pipelined stream per wave after tools/ubench/mfma_stream_model.hip with the decode between the MFMAs and the
// stage barrier in the middle of a step (3 % faster, 256 registers with spills: not kept); an s_sleep of 0 / 200 / 400 cycles by wave class after each
// stage barrier, so that the three waves of a SIMD start their stages staggered (no change); fixed wave priorities 3 / 2 / 1 for the three waves of a SIMD (no change).
// ---------------------------------------------------------------------------------------------------------------------------------
// NB = row blocks (of 32 latent rows) per wave.  A workgroup always covers 24 row blocks = 768 rows: 24 / NB waves.  Every thread decodes one
// (point, 4 sub-quantizers) item per stage, so a stage is threads / 128 tiles.
//   NB = 2: 12 waves, three per SIMD (<= 168 registers), stages of 6 tiles.  Every wave reads every operand tile from LDS itself: 10 KB per tile and wave (6 KB of
//           A fragments + 4 KB of point terms) for 12 MFMAs — 12 waves x 10 KB = 960 LDS-cycles per tile round against 1152 matrix-pipe cycles per SIMD: the LDS return path
//           is all but co-critical (with the tracking stubbed out AND the decode skipped the kernel still takes 0.83 of its time: profiles/r04_bound_pass_ablation.json).
//   NB = 3:  8 waves, two per SIMD (<= 256 registers), stages of 4 tiles: the same 10 KB feed 18 MFMAs (the point terms are shared by the three row blocks), a third less
//           LDS traffic per MFMA.
constexpr int kM12RowBlocks = 24;
template <int TILES> struct __align__(16) M12Stage {
    uint4 a[TILES][12][32];
    float nrm[TILES][32];
    int2 meta[TILES];
};

// The records are written once and read ~100 ms later by the recomputation kernel: 9 GB per launch that no cache can hold.  AFIS_MF_NT_STORE (experiment) marks the stores
// non-temporal so that they do not displace what the kernels running beside the pass keep in the L2 (the candidate kernel's latent fragments).
// Measured (tools/lib_ab.py, 20 latents x 100 000, default schedule): 426.6 against 427.3 ms per group, candidates 173.8 against 174.1 — nothing; the candidate kernel beside
// the pass takes 2.05 x its time alone on the chip, i.e. what half the CUs cost.  Not shipped.
#ifndef AFIS_MF_NT_STORE
#define AFIS_MF_NT_STORE 0
#endif
__device__ __forceinline__ void store_rec(uint2* p, uint32_t a, uint32_t b)
{
#if AFIS_MF_NT_STORE
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 v; v.x = a; v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
#else
    *p = make_uint2(a, b);
#endif
}

template <int NB>
__global__ __launch_bounds__(64 * (kM12RowBlocks / NB)) void k_adc_mfma(GalleryDev g, const uint4* __restrict__ codes_p, const float* __restrict__ nrm_p,
                                                            const int2* __restrict__ tile_meta, const int32_t* __restrict__ tile0, const uint4* __restrict__ cw16,
                                                            const uint4* __restrict__ bfrag, const float4* __restrict__ rowk, int n_rows, int n_rb, int R_pad,
                                                            int n_rg, int chunk, uint2* __restrict__ rec, unsigned long long* __restrict__ diag, int xcd_map)
{
    constexpr int kWaves = kM12RowBlocks / NB, kThreads = 64 * kWaves, kStageTiles = kThreads / 128;
    static_assert(kWaves * NB == kM12RowBlocks && kStageTiles * 128 == kThreads, "row blocks per wave must divide 24, and the threads must decode whole tiles");
    __shared__ uint4 s_cw[kM * kK];                                     // 64 KB
    __shared__ M12Stage<kStageTiles> s_st[2];                           // 2 x 6.3 KB per tile of the stage
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2: the row groups of ONE gallery chunk go to ONE XCD, one after the other, so that a chunk's codes
    // cross the fabric once per round of row groups instead of once per XCD (the launcher rounds the chunk count up to a multiple of 8; chunks beyond the shard return at once).
    const int slot = xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int rg = slot % n_rg, chunk_id = xcd_map ? (slot / n_rg) * 8 + (int)(blockIdx.x & 7) : slot / n_rg;
    const int t_lo = chunk_id * chunk, t_hi = min(g.G, t_lo + chunk);
    if (t_lo >= t_hi) return;
    const int tile_lo = tile0[t_lo], tile_hi = tile0[t_hi];                // tiles of 32 rolled points; a template owns ceil(n/32) of them
    const int n_tiles = tile_hi - tile_lo;
    if (n_tiles <= 0) return;
    const int n_stages = (n_tiles + kStageTiles - 1) / kStageTiles;
    // The clock the chip holds under THIS kernel (afis_timing.bound_clock_ghz): one lane of every 64th workgroup reads the shader-cycle counter (s_memtime) and the
    // constant 100 MHz counter (s_memrealtime) when its workgroup starts and when it ends; the sums of the differences go to the launch group's diagnostics row.
    const bool sampler = diag != nullptr && (blockIdx.x & 63) == 0 && wave == 0;      // wave-uniform: the two counters stay in scalar registers; lane 0 adds them up at the end
    unsigned long long clk0 = 0, wall0 = 0;
    if (sampler) { clk0 = __builtin_readcyclecounter(); wall0 = wall_clock64(); }
    for (int i = tid; i < kM * kK; i += kThreads) s_cw[i] = cw16[i];

    const int h = lane >> 5, col = lane & 31;
    const int rb0 = rg * kM12RowBlocks + wave * NB;
    const bool wave_ok = rb0 < n_rb;
    half8 bf[NB][6];
    float Tg[NB]; bool force[NB];
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int rb = rb0 + blk;
#pragma unroll
        for (int kk = 0; kk < 6; ++kk) {
            const uint4 v = rb < n_rb ? bfrag[((size_t)rb * 6 + kk) * 64 + lane] : make_uint4(0, 0, 0, 0);
            bf[blk][kk] = __builtin_bit_cast(half8, v);
        }
        const int row = rb * 32 + col;
        const float4 rk = row < n_rows ? rowk[row] : make_float4(0.f, 0.f, 0.f, 0.f);
        Tg[blk] = rk.z; force[blk] = rk.w != 0.0f;
    }
    const int pp = tid & 31, pQ = (tid >> 5) & 3, pj = tid >> 7;       // point in tile, sub-quantizer quad, tile of the stage
    const bool meta_thread = pQ == 1 && pp == 0;
    struct Pf { uint32_t code; float nrm; int2 meta; };
    auto fetch = [&](int s, Pf& f) {
        const int tile = tile_lo + kStageTiles * s + pj;
        f.code = 0u; f.nrm = kMfNeg; f.meta = make_int2(0, 0);
        if (tile < tile_hi) {
            const size_t e = (size_t)tile * 32 + pp;
            f.code = reinterpret_cast<const uint32_t*>(codes_p)[e * 4 + pQ];
            if (pQ == 0) f.nrm = nrm_p[e];
            if (meta_thread) f.meta = tile_meta[tile];
        }
    };
    auto decode = [&](int buf, const Pf& f) {
        uint4 w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = s_cw[(4 * pQ + i) * kK + ((f.code >> (8 * i)) & 255u)];
        M12Stage<kStageTiles>& st = s_st[buf];
        st.a[pj][3 * pQ + 0][pp] = make_uint4(w[0].x, w[0].y, w[0].z, w[1].x);
        st.a[pj][3 * pQ + 1][pp] = make_uint4(w[1].y, w[1].z, w[2].x, w[2].y);
        st.a[pj][3 * pQ + 2][pp] = make_uint4(w[2].z, w[3].x, w[3].y, w[3].z);
        if (pQ == 0) {                                                   // the address is rebuilt from the thread index here: hoisted out of the stage loop it was two more live registers than the kernel has (168 at three waves per SIMD) and went to scratch
            int t = tid; asm volatile("" : "+v"(t));
            st.nrm[t >> 7][t & 31] = f.nrm;
        }
        if (meta_thread) st.meta[pj] = f.meta;
    };
    Pf pf_cur, pf_nxt;
    fetch(0, pf_cur);
    __syncthreads();
    decode(0, pf_cur);
    fetch(1, pf_cur);
    __syncthreads();

    float m[NB][8], tb[NB], ts[NB], tu[NB];
    auto reset = [&]() {
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
            for (int k = 0; k < 8; ++k) m[blk][k] = kMfNeg;
            tb[blk] = ts[blk] = tu[blk] = kMfNeg;
        }
    };
    reset();
    auto track = [&](int blk, const floatx16& X, uint32_t gid) {
        float lo = max3f(X[0], X[1], X[2]), hi = max3f(X[8], X[9], X[10]);
        lo = max3f(lo, X[3], X[4]); hi = max3f(hi, X[11], X[12]);
        lo = max3f(lo, X[5], X[6]); hi = max3f(hi, X[13], X[14]);
        lo = fmaxf(lo, X[7]); hi = fmaxf(hi, X[15]);
        const float el = u2f((f2u(lo) & ~63u) | gid), eh = u2f((f2u(hi) & ~63u) | (gid + 1u));
        tu[blk] = med3f(ts[blk], el, tu[blk]); ts[blk] = med3f(tb[blk], ts[blk], el); tb[blk] = fmaxf(tb[blk], el);
        tu[blk] = med3f(ts[blk], eh, tu[blk]); ts[blk] = med3f(tb[blk], ts[blk], eh); tb[blk] = fmaxf(tb[blk], eh);
#pragma unroll
        for (int k = 0; k < 8; ++k) m[blk][k] = max3f(m[blk][k], X[k], X[k + 8]);
    };
    // Per (template, row) ONE record: the two lane halves of a row (lanes col and col + 32: the points 8q + 0..3 and 8q + 4..7 of every tile) are merged before
    // the store.  v_permlane32_swap of (block 2p's, block 2p + 1's) registers hands lanes 0-31 both halves of block 2p's rows and lanes 32-63 both halves of
    // block 2p + 1's, so one merge serves 64 rows and the wave stores 512 contiguous bytes per template and block pair (round 3 stored 16 B per row and
    // template, 17.9 GB per launch, and the recomputation kernel read both halves of every row).  An unpaired last block (NB odd) is swapped with itself: both
    // lane halves then hold its rows, the lower one stores.
    typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
    auto finish_template = [&](int tmpl) {
        uint32_t val[NB], dsc[NB];
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            float b3 = kMfNeg, s3 = kMfNeg, u3 = kMfNeg;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float e = u2f((f2u(m[blk][k]) & ~7u) | (uint32_t)k);
                u3 = med3f(s3, e, u3); s3 = med3f(b3, s3, e); b3 = fmaxf(b3, e);
            }
            const float thr = fminf(tb[blk], b3) - Tg[blk];
            const bool many = (tu[blk] >= thr) | (u3 >= thr) | force[blk];
            val[blk] = f2u(b3);
            dsc[blk] = (f2u(tb[blk]) & 63u) | ((f2u(ts[blk]) & 63u) << 6) | ((f2u(b3) & 7u) << 12) | ((f2u(s3) & 7u) << 15) |
                       ((ts[blk] >= thr ? 1u : 0u) << 18) | ((s3 >= thr ? 1u : 0u) << 19) | ((many ? 1u : 0u) << 20);
        }
#pragma unroll
        for (int p0 = 0; p0 < NB; p0 += 2) {
            constexpr bool kDummy = false; (void)kDummy;
            const bool paired = p0 + 1 < NB;                             // compile-time after unrolling
            const int p1 = paired ? p0 + 1 : p0;
            const uint2v rv = __builtin_amdgcn_permlane32_swap(val[p0], val[p1], false, false);   // .x: half 0's record of row (block p0 + h, col), .y: half 1's
            const uint2v rd = __builtin_amdgcn_permlane32_swap(dsc[p0], dsc[p1], false, false);
            const float TgM = paired ? (h ? Tg[p1] : Tg[p0]) : Tg[p0];
            const float v0 = u2f(rv.x), v1 = u2f(rv.y);
            const bool sw = v1 > v0;                                     // the half that holds the row's best value is the primary one (ties: half 0)
            const float V = fmaxf(v0, v1), vo = fminf(v0, v1);
            const uint32_t dp = sw ? rd.y : rd.x, dn = sw ? rd.x : rd.y;
            const bool in_o = vo >= V - TgM;                             // the other half's best is within reach of the row maximum: its best cell is a candidate too,
            const bool o_more = (dn & (7u << 18)) != 0u;                 // and if it has runners-up of its own the row is evaluated over every point
            const uint32_t cell = (dn & 63u) | (((dn >> 12) & 7u) << 6);
            const uint32_t D = (dp & 0x1fffffu) | ((sw ? 1u : 0u) << 21) | ((in_o ? 1u : 0u) << 22) | (cell << 23) | (((in_o & o_more) ? 1u : 0u) << 20);
            // padding rows of a partial row block store too (their records are never read: R_pad covers them); only a row block beyond the last is skipped
            const int rbm = rb0 + p0 + (paired ? h : 0);
            if (rbm < n_rb && (paired || h == 0)) store_rec(&rec[(size_t)tmpl * R_pad + (size_t)rbm * 32 + col], f2u(V), D);
        }
        reset();
    };

    for (int s = 0; s < n_stages; ++s) {
        fetch(s + 2, pf_nxt);
        const M12Stage<kStageTiles>& st = s_st[s & 1];
        if (wave_ok) {
            // The operand groups of tile j + 1 are read from LDS while tile j's results are still on their way out of the matrix pipe and being tracked (round 5): a wave's chain used to be
            // read operands -> 12 MFMAs -> track -> read operands ..., each link waiting for the one before (with the operands read once per stage, a timing-only ablation, the pass took
            // 24 % less).  The reads are issued right behind the last MFMA of tile j into the SAME registers — an MFMA has read its A operand long before the LDS answers — so the
            // prefetch costs no register; only a stage's first tile still waits for its operands.  (The two forms of rounds 3-4 that broke the chain kept two operand AND two accumulator
            // sets: 236 registers, two waves per SIMD, slower.)
            half8 af[6];
#if AFIS_MF_PREFETCH
#pragma unroll
            for (int kk = 0; kk < 6; ++kk) af[kk] = __builtin_bit_cast(half8, st.a[0][2 * kk + h][col]);
#endif
#pragma unroll
            for (int j = 0; j < kStageTiles; ++j) {
#if AFIS_MF_ABLATE == 5 || AFIS_MF_ABLATE == 7          // timing experiments only: the operands are read from LDS once per stage, not per tile
                const int jr = 0;
#else
                const int jr = j;
#endif
                floatx16 nrm;
#if AFIS_MF_ABLATE == 10                                 // timing experiment only (wrong results): the point terms are not read — what delivering the C operand for free could save at most
#pragma unroll
                for (int r = 0; r < 16; ++r) nrm[r] = 0.0f;
#else
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 v = *reinterpret_cast<const float4*>(&st.nrm[jr][8 * q4 + 4 * h]);
                    nrm[4 * q4] = v.x; nrm[4 * q4 + 1] = v.y; nrm[4 * q4 + 2] = v.z; nrm[4 * q4 + 3] = v.w;
                }
#endif
                const int2 mv = st.meta[j];
#if !AFIS_MF_PREFETCH
#pragma unroll
                for (int kk = 0; kk < 6; ++kk) af[kk] = __builtin_bit_cast(half8, st.a[jr][2 * kk + h][col]);
#endif
                floatx16 X[NB];
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) X[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[0], bf[blk][0], nrm, 0, 0, 0);
#pragma unroll
                for (int kk = 1; kk < 6; ++kk) {
#pragma unroll
                    for (int blk = 0; blk < NB; ++blk) X[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[kk], bf[blk][kk], X[blk], 0, 0, 0);
                }
#if AFIS_MF_PREFETCH
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < kStageTiles) {
#pragma unroll
                    for (int kk = 0; kk < 6; ++kk) af[kk] = __builtin_bit_cast(half8, st.a[j + 1][2 * kk + h][col]);
                }
                __builtin_amdgcn_sched_barrier(0);
#endif
                const int my = __builtin_amdgcn_readfirstlane(mv.y);
                const uint32_t gid = (uint32_t)(2 * (my & 255));
#if AFIS_MF_ABLATE == 1 || (AFIS_MF_ABLATE >= 4 && AFIS_MF_ABLATE <= 7)          // timing experiments only (wrong results): no tracking
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) m[blk][0] = max3f(m[blk][0], X[blk][0], X[blk][15]);
                (void)gid;
#elif AFIS_MF_ABLATE == 11                               // timing experiment only (wrong results): the cheapest conceivable packed tracking — 8 conversions to fp16 pairs + 8 v_pk_max_f16 per 16 values, no group tracking
#pragma unroll
                for (int blk = 0; blk < NB; ++blk)
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                        const half2v pk = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(X[blk][k], X[blk][k + 8]));
                        const half2v cur = __builtin_bit_cast(half2v, m[blk][k]);
                        m[blk][k] = __builtin_bit_cast(float, __builtin_elementwise_max(cur, pk));
                    }
                (void)gid;
#else
#pragma unroll
                for (int blk = 0; blk < NB; ++blk) track(blk, X[blk], gid);
#endif
                if (my & 256) finish_template(__builtin_amdgcn_readfirstlane(mv.x));
            }
        }
#if AFIS_MF_ABLATE == 3 || (AFIS_MF_ABLATE >= 4 && AFIS_MF_ABLATE <= 7)           // timing experiments only: the stages after the first two are not decoded (stale operands)
        if (s + 1 < n_stages && s < 1) decode((s + 1) & 1, pf_cur);
#else
        if (s + 1 < n_stages) decode((s + 1) & 1, pf_cur);
#endif
        pf_cur = pf_nxt;
#if AFIS_MF_ABLATE == 6 || AFIS_MF_ABLATE == 7           // timing experiments only: no stage barrier
        if (s < 1) __syncthreads();
#else
        __syncthreads();
#endif
    }
    if (sampler && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) { atomicAdd(&diag[kDiagBoundClk], (unsigned long long)__builtin_readcyclecounter() - clk0); atomicAdd(&diag[kDiagBoundWall], (unsigned long long)wall_clock64() - wall0); }
}

// ---- launcher ----------------------------------------------------------------------------------------------------------------
hipError_t launch_adc_mfma(const GalleryDev& g, const void* codes_p, const float* nrm_p, const void* tile_meta, const int32_t* tile0, const void* cw16,
                           const void* bfrag, const void* rowk, int n_rows, int n_rb, int R_pad, int chunk, int blocks_per_wave, void* rec, unsigned long long* diag, hipStream_t stream)
{
    if (n_rb <= 0 || g.G <= 0) return hipSuccess;
    const int n_chunks = (g.G + chunk - 1) / chunk;
    const int n_rg = (n_rb + kM12RowBlocks - 1) / kM12RowBlocks;
    static const bool xcd_map = AFIS_EXPERIMENT_ENV("AFIS_MF_NO_XCD_MAP") == nullptr;       // experiment knob: round-robin chunks as in rounds 3-4
    const long long blocks = (long long)n_rg * (xcd_map ? (n_chunks + 7) / 8 * 8 : n_chunks);
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#ifdef AFIS_EXPERIMENTAL_KERNELS                                          // three row blocks per wave (8 waves, 224 registers): -1 % alone on the chip, +2 % in the default schedule; test library only
    if (blocks_per_wave == 3) {
        hipLaunchKernelGGL(k_adc_mfma<3>, dim3((unsigned)blocks), dim3(64 * (kM12RowBlocks / 3)), 0, stream, g, (const uint4*)codes_p, nrm_p, (const int2*)tile_meta, tile0,
                           (const uint4*)cw16, (const uint4*)bfrag, (const float4*)rowk, n_rows, n_rb, R_pad, n_rg, chunk, (uint2*)rec, diag, xcd_map ? 1 : 0);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(k_adc_mfma<2>, dim3((unsigned)blocks), dim3(64 * (kM12RowBlocks / 2)), 0, stream, g, (const uint4*)codes_p, nrm_p, (const int2*)tile_meta, tile0,
                           (const uint4*)cw16, (const uint4*)bfrag, (const float4*)rowk, n_rows, n_rb, R_pad, n_rg, chunk, (uint2*)rec, diag, xcd_map ? 1 : 0);
    return hipGetLastError();
}

}  // namespace afis
