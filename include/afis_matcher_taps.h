/* afis_matcher_taps.h — parity-test taps of the MI355X matcher: stage intermediates the tests compare with oracle/ bit for bit.
 * TEST INFRASTRUCTURE: exported by libafis_hip_test.so only (csrc/Makefile: afis_api.cpp with -DAFIS_PARITY_TAPS); the product
 * library libafis_hip.so (include/afis_matcher.h) does not contain them. */
#ifndef AFIS_MATCHER_TAPS_H
#define AFIS_MATCHER_TAPS_H

#include "afis_matcher.h"

#ifdef __cplusplus
extern "C" {
#endif

/* S4: the per-query LUT of queries[0].tex[0], out = [n][16][256] in the reference's m_dist_codewords layout. */
int afis_debug_lut(afis_ctx* ctx, const afis_template_view* query, float* out, int32_t* n_rows);
/* S5+S6: row maxima / first arg-max of latent texture 0 vs gallery template g (g is shard-local). */
int afis_debug_texture_rowmax(afis_ctx* ctx, const afis_template_view* query, int64_t g,
                              float* val, int32_t* arg, int32_t* n_rows);

/* S3 / S7 / S8 / S9: the correspondence list of (query, gallery template g) inside one scorer, as (sim, latent index, rolled
 * index) triples in list order.  which: 0 = texture scorer, 1..3 = minutiae scorer of selected latent template 27 / 3 / 12;
 * stage: 0 = the candidates (top 120 / top 200), 1 = after the distance filter, 2 = after the angle filter.  Capacity 200.
 * *n = -1 when the reference does not run that scorer for the pair. */
int afis_debug_stage_list(afis_ctx* ctx, const afis_template_view* query, int64_t g, int which, int stage,
                          float* sim, int32_t* li, int32_t* ri, int32_t* n);

/* S9: the angle stage's atan2 (matching/matcher.cpp:1516, :1524) on every integer coordinate difference of the grid
 * [-R, R]^2: out[(dy + R) * (2R + 1) + (dx + R)] = line angle atan2(dy, dx) as the device evaluates it.  R <= 4096. */
int afis_debug_atan2_grid(afis_ctx* ctx, int R, float* out);

/* S8: the distance stage's packed arithmetic (csrc/graph_arith.h: a one-transcendental correctly rounded square root of integers and
 * a square-root-free "H != 0" test with a guard band) against the plain evaluation of matching/matcher.cpp:1246-1272, :1372-1393 that
 * it replaces, on the device.  out8[0] = integers n in [0, 2*2047^2] whose root differs; out8[1..3] = texture pairs checked (all of
 * [0, 4802]^2), pairs inside the guard band, wrong decisions; out8[4..6] = the same for minutiae pairs near the 30 px threshold
 * (4e8 of them).  out8[0], [3], [6] must be 0. */
int afis_debug_graph_arith(afis_ctx* ctx, unsigned long long* out8);
/* adc_variant 9 with afis_set_option("mf_stats", 1): counters of the selection / recomputation kernel since the last reset:
 * out8[0] pairs, [1] latent rows, [2] rows evaluated exactly, [3] candidate cells evaluated, [4] rows evaluated over every point,
 * [5] rows whose exact maximum lay outside the bounds the selection used (a self-check: must be 0). */
int afis_debug_refine_stats(afis_ctx* ctx, unsigned long long* out8, int reset);

/* In-kernel phase timers (only when the library is built with PHASE_TIMING=1; all zeros otherwise): 32 cycle counters
 * accumulated since the last reset.  Development aid. */
int afis_debug_phase_cycles(afis_ctx* ctx, unsigned long long* out32, int reset);

#ifdef __cplusplus
}
#endif
#endif /* AFIS_MATCHER_TAPS_H */
