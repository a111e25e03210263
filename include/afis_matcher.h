/* afis_matcher.h — C ABI of the MI355X-native latent-vs-gallery matcher (libafis_hip.so).
 *
 * This is the drop-in boundary for the hot path of prip-lab/MSU-LatentAFIS `matching/`: the body of the
 * reference's OpenMP gallery loop (matching/matcher.cpp:168-190 == :273-295), i.e. everything reached from
 * PQ::Matcher::One2One_matching_selected_templates (matcher.h:43) for every (latent, rolled) pair, plus the
 * score fusion at matcher.cpp:188/:293.  The reference has no plugin/FFI interface of its own; a maintainer
 * replaces the loop body with one afis_search() call (see INTEGRATION.md for the exact patch).
 *
 * Conventions: plain C types only, no exceptions cross the boundary, every function returns 0 on success or a
 * negative AFIS_E* code (afis_last_error() gives the text).  A context is bound to ONE HIP device and is used
 * by one host thread at a time (PQ::Matcher is not re-entrant either).  All pointers are HOST pointers; the
 * library copies what it needs, the caller keeps ownership.  There is no CPU fallback: without a usable
 * gfx950 device afis_create() fails.
 */
#ifndef AFIS_MATCHER_H
#define AFIS_MATCHER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFIS_OK            0
#define AFIS_EINVAL       -1   /* bad argument / unsupported shape                     */
#define AFIS_EDEVICE      -2   /* HIP error, no device, out of device memory            */
#define AFIS_ESTATE       -3   /* call order (search before commit, add after commit)   */
#define AFIS_EFORMAT      -4   /* malformed template / codebook bytes                   */

/* Per-query status, mirrors One2One_matching_selected_templates' return (matcher.cpp:383-391). */
#define AFIS_QUERY_OK            0
#define AFIS_QUERY_LATENT_EMPTY  1   /* whole query skipped, its scores are left at -1 (matcher.cpp:191-194) */

typedef struct afis_ctx afis_ctx;   /* opaque; owns all device memory */

/* One minutiae template (reference: MinutiaeTemplate, matching/include.h:203-252).  x,y in pixels. */
typedef struct afis_minutiae_view {
    int32_t        n;        /* number of minutiae (> 0; n <= 0 templates are dropped, matcher.cpp:835-836) */
    const int16_t* x;        /* [n] */
    const int16_t* y;        /* [n] */
    const float*   ori;      /* [n] radians */
    int32_t        des_len;  /* descriptor length, must be 96 on this path */
    const float*   des;      /* [n][des_len] row-major fp32 */
} afis_minutiae_view;

/* One texture (virtual-minutiae) template.  x,y in BLOCK units ((px-24)/16).
 * Latent (LatentTextureTemplate, include.h:298-364): des = fp32 [n][96], codes = NULL.
 * Rolled (RolledTextureTemplatePQ, include.h:366-485): codes = u8 [n][16] PQ codes, des = NULL. */
typedef struct afis_texture_view {
    int32_t        n;
    const int16_t* x;
    const int16_t* y;
    const float*   ori;
    int32_t        des_len;  /* 96 (latent) or 16 (rolled) */
    const float*   des;
    const uint8_t* codes;
} afis_texture_view;

/* A whole fingerprint template (LatentFPTemplate / RolledFPTemplate, include.h:519-558), already stripped of
 * zero-minutiae templates exactly as Matcher::load_FP_template does (indices are post-strip indices). */
typedef struct afis_template_view {
    int32_t                   n_minu;
    const afis_minutiae_view* minu;
    int32_t                   n_tex;
    const afis_texture_view*  tex;
} afis_template_view;

/* Per-stage device time of the last afis_search call, milliseconds, from HIP events on the context's stream. */
typedef struct afis_timing {
    float   lut_ms;        /* S4  per-query LUT build                                  */
    float   adc_ms;        /* S5+S6 texture ADC similarity + row arg-max (dominant)     */
    float   tex_tail_ms;   /* S7+S8b+S9 on the texture correspondences                  */
    float   minu_ms;       /* S1-S3+S8a+S9 for the three selected minutiae templates    */
    float   fuse_ms;       /* S10 fusion                                                */
    float   topk_ms;       /* S11 per-query rank lists on the device (k <= 64)           */
    float   total_ms;      /* sum of the stages above                                   */
    int32_t adc_launches;  /* number of ADC kernel launches in the call                 */
    int64_t adc_lookups;   /* LUT look-ups performed by those launches                  */
    int64_t pairs;         /* (query, gallery template) pairs scored                    */
    float   adc_bound_ms;  /* part of adc_ms: the bound pass over every cell (adc_variant 8: the whole kernel; 9: k_adc_mfma) */
    float   adc_refine_ms; /* part of adc_ms: adc_variant 9's selection + exact recomputation kernel; 0 otherwise             */
    float   cands_ms;      /* part of minu_ms: S1-S3 (descriptor GEMM, normalisation, top-120 candidates)                      */
    float   minu_graph_ms; /* part of minu_ms: S8a + S9 on the minutiae correspondence lists                                   */
    int32_t launch_groups; /* launch groups the queries were cut into                                                          */
    int32_t overlapped_groups; /* of those: groups that ran the overlapped schedule (option bound_cus); the others ran their kernels back to back */
    /* round 5 (afis_get_timing2 with the caller's struct size): where the minutiae candidate tasks went, and the clocks the device held */
    int64_t minu_tasks;          /* (latent minutiae list, rolled template) candidate tasks of the call with minutiae on both sides   */
    int64_t minu_fallback_tasks; /* of those: tasks the shape-class kernels handed to the any-shape kernel (k_minu_cands)             */
    int64_t minu_tasks_small;    /* tasks done by the small  shape class of the fast kernel (k_minu_cands_rt<1>: 256 threads, <= 64 x 128) */
    int64_t minu_tasks_medium;   /* ... by the medium class (k_minu_cands_rt<2>: 512 threads, <= 16 384 similarities)                     */
    int64_t minu_tasks_large;    /* ... by the large  class (k_minu_cands_rt<4>: 1024 threads, <= 38 912 similarities incl. the stride padding, <= 256 latent x 512 rolled) */
    float   bound_clock_ghz;     /* shader clock under the bound pass: s_memtime ticks / s_memrealtime (100 MHz) over workgroup lifetimes, mean of the sampled workgroups; 0 = not sampled */
    float   cands_clock_ghz;     /* the same under the candidate kernel (a kernel that is not power-limited, for comparison)          */
} afis_timing;

/* Replaces PQ::Matcher::Matcher(code_file) (matcher.cpp:31-94).  codewords = [M][K][dsub] fp32 exactly as stored
 * in the codebook .dat after its 3 x int16 header.  Only M=16, K=256, dsub=6 is supported. */
int afis_create(afis_ctx** out, const float* codewords, int M, int K, int dsub, int device_id);
/* Same, from the bytes of a codebook .dat file (3 x int16 header + floats). */
int afis_create_from_codebook(afis_ctx** out, const void* codebook_bytes, size_t len, int device_id);
void afis_destroy(afis_ctx* ctx);
const char* afis_last_error(const afis_ctx* ctx);   /* ctx may be NULL: last afis_create failure */

/* Identity of HIP device `device_id` as the runtime reports it: marketing name + architecture, PCI bus id ("0000:c1:00.0", hipDeviceGetPCIBusId), UUID as 32 hex digits
 * (hipDeviceGetUuid), compute units.  Any output may be NULL.  One process per GPU: a multi-GPU job reports these per rank so that it can be told
 * from ranks sharing one device. */
int afis_device_info(int device_id, char* name, size_t name_cap, char* pci_bus_id, size_t pci_cap, char* uuid_hex /* >= 33 bytes */, size_t uuid_cap, int* n_cus);

/* Gallery build: replaces the per-pair load_FP_template(rolled) at matcher.cpp:173/:278 — parse once, keep the
 * gallery resident in HBM.  Only minutiae template 0 and texture template 0 of a rolled template are ever used
 * by the reference (matcher.cpp:406,:413).  Templates keep insertion order; index = position. */
int afis_gallery_add(afis_ctx* ctx, const afis_template_view* templates, int n);
/* Parse one rolled .dat (layout of Matcher::load_FP_template(string, RolledFPTemplate&), matcher.cpp:886-983).
 * *load_rc receives the reference's return code (0 ok, 1 empty file, 2, 4, -1); an entry is ALWAYS appended so
 * indices stay aligned with the caller's file list (empty entry => score -1, matcher.cpp:184-187). */
int afis_gallery_add_dat(afis_ctx* ctx, const void* bytes, size_t len, int* load_rc);
/* n rolled .dat files in one call: parsed on the host's threads, appended in order; load_rc (optional) receives n reader codes.
 * Nothing is appended when any file is rejected (AFIS_EINVAL, as afis_gallery_add_dat). */
int afis_gallery_add_dat_batch(afis_ctx* ctx, const void* const* bytes, const size_t* lens, int64_t n, int* load_rc);
/* Hint: the staged gallery will grow to about n_templates templates.  The host arrays reserve address space for them pro rata (from what is
 * staged so far; 80 minutiae and 800 texture points per template if nothing is), so that a gallery added in many slices is not re-copied every
 * time an array outgrows its allocation (a 100 000-file directory: 5 GB staged, 6 GB of re-copying without the hint).  Never required. */
int afis_gallery_reserve(afis_ctx* ctx, int64_t n_templates);
/* Bulk add of n templates with exactly one minutiae and one texture template each, as concatenated arrays with
 * CSR offsets (off[n+1], in points).  A zero-length range means "template absent". */
int afis_gallery_add_packed(afis_ctx* ctx, int64_t n,
                            const int64_t* minu_off, const int16_t* minu_x, const int16_t* minu_y,
                            const float* minu_ori, const float* minu_des /* [sum][96] */,
                            const int64_t* tex_off, const int16_t* tex_x, const int16_t* tex_y,
                            const float* tex_ori, const uint8_t* tex_codes /* [sum][16] */);
/* SoA-pack and upload.  index_base is added to every index afis_search reports (gallery sharding: each rank
 * commits its contiguous shard with the shard's global offset). */
int afis_gallery_commit(afis_ctx* ctx, int64_t index_base);
int64_t afis_gallery_size(const afis_ctx* ctx);

/* The hot path.  Replaces the body of the OpenMP loop of One2List_matching / List2List_matching
 * (matcher.cpp:168-190, :273-295) for n_q latents at once.
 *   scores      [n_q][G] or NULL : final fused score per gallery template, -1 where the rolled template is empty
 *                                  (or for every entry of a latent-empty query)
 *   parts       [n_q][G][4] or NULL : s0, s1, s2 (latent minutiae templates 26, 2, 11) and the texture score
 *   status      [n_q] or NULL    : AFIS_QUERY_*
 *   k, topk_idx [n_q][k], topk_score [n_q][k] : rank list, score descending, ties by ascending index
 *                                  (the reference's tie order is unspecified, matcher.cpp:306-309 — a caller that wants the
 *                                  binary's order of EQUAL scores passes the score column to afis_rank_list(..., ref_order 1), as match -l -tie does); padded with
 *                                  idx -1 when k > G.  k = 0 skips it. */
int afis_search(afis_ctx* ctx, const afis_template_view* queries, int n_q,
                float* scores, float* parts, int32_t* status,
                int k, int64_t* topk_idx, float* topk_score);
/* Same with the queries given as latent .dat bytes (Matcher::load_FP_template(string, LatentFPTemplate&),
 * matcher.cpp:785-884). */
int afis_search_dat(afis_ctx* ctx, const void* const* latent_bytes, const size_t* lens, int n_q,
                    float* scores, float* parts, int32_t* status,
                    int k, int64_t* topk_idx, float* topk_score);

/* Queries resident in HBM before the timed region (bench): upload once, search many times.
 * afis_search_resident (and afis_search on top of it) runs the query groups back to back on the context's stream and syncs once.
 * For k <= 64 the rank lists are made on the device (per-query top-k kernel over the shard's scores, score descending / index
 * ascending) and only n_q x k x 12 bytes return to the host; the [n_q][G] score matrix crosses PCIe only when `scores` is given
 * (-ldir mode) and the per-part scores only when `parts` is.  k > 64 sorts on the host. */
typedef struct afis_queries afis_queries;
int afis_queries_upload(afis_ctx* ctx, const afis_template_view* queries, int n_q, afis_queries** out);
int afis_search_resident(afis_ctx* ctx, afis_queries* q,
                         float* scores, float* parts, int32_t* status,
                         int k, int64_t* topk_idx, float* topk_score);
void afis_queries_free(afis_ctx* ctx, afis_queries* q);

/* Packed gallery container (no reference counterpart: the reference re-parses every rolled .dat for every pair,
 * matching/matcher.cpp:173,:278).  One mmap-able file holding the staged gallery's SoA arrays (layout: csrc/template_io.h), so a
 * 100k-1M template gallery is loaded — whole, or one contiguous shard per GPU — without touching 100k-1M small files.
 * afis_gallery_save   writes the templates staged so far (before afis_gallery_commit); names[i] (optional) = the path template
 *                     i was read from, kept for the score files.
 * afis_gallery_load   appends templates [first, first+count) of the file (count < 0: to the end) to the staged gallery.  Into an EMPTY staging
 *                     area (the usual case: one container, or one shard of it per rank) the file is validated and kept mapped, and
 *                     afis_gallery_commit uploads the range straight from the mapping (no host copy of its 50 KB per template): the
 *                     file must not be truncated or rewritten between the two calls (a truncation found at commit time is AFIS_EFORMAT; one that happens WHILE the commit copies from the mapping
 *                     faults in the host process, as for any mapped file).  Any other staging call in between first copies
 *                     the range into host memory, as every load into a non-empty staging area does.
 * afis_gallery_file_info  template / point totals, and (optional) the texture point count of every template, the quantity shards
 *                     are balanced by.
 * afis_gallery_file_names  the names of a range as consecutive NUL-terminated strings; buf == NULL only reports *need. */
int afis_gallery_save(afis_ctx* ctx, const char* path, const char* const* names);
int afis_gallery_load(afis_ctx* ctx, const char* path, int64_t first, int64_t count);
int afis_gallery_file_info(const char* path, int64_t* G, int64_t* n_minutiae, int64_t* n_tex_points, int32_t* tex_counts /*[G] or NULL*/);
int afis_gallery_file_names(const char* path, int64_t first, int64_t count, char* buf, size_t cap, size_t* need);

/* Matcher::One2One_matching_all_templates (matching/matcher.cpp:339-374) for one latent against the whole resident gallery:
 * scores[g][i] for i < n_minu = latent minutiae template i vs rolled minutiae template 0, scores[g][n_minu + t] = latent texture
 * template t vs rolled texture template 0; zero where the reference leaves the zero-filled vector untouched.
 *   scores        [G][query->n_minu + query->n_tex]
 *   rolled_status [G] or NULL : 0, or 2 = rolled template empty (the reference returns 2 before scoring, :350-353)
 *   query_status  NULL or out : AFIS_QUERY_LATENT_EMPTY when the latent has no template at all (:345-348) */
int afis_match_all_templates(afis_ctx* ctx, const afis_template_view* query, float* scores, int32_t* rolled_status, int32_t* query_status);

/* The rank list of One2List_matching, matcher.cpp:306-309, from a score column (host memory, no device work): idx[0 .. k) / sc[0 .. k) = the k best of scores[0 .. n), score descending.
 * ref_order 0: equal scores by ascending index (what afis_search's own top-k delivers); 1: the reference's statement itself — std::sort of the indices 0 .. n-1 on the non-strict
 * comparator scores[a] > scores[b], with this library's libstdc++ — so that equal scores (the zero scores at the tail of a small gallery's list) come out in the order the reference
 * binary leaves them.  k > n: the rest is padded with idx -1, sc 0.  sc may be NULL. */
int afis_rank_list(const float* scores, int64_t n, int ref_order, int k, int64_t* idx, float* sc);

/* PQ encoder — replaces TrainedPQEncoder.encode_multi (extraction/descriptor_PQ.py:19-27, scipy.cluster.vq.vq per
 * sub-space): codes[i][m] = index of the codeword of sub-quantizer m nearest (squared L2, fp32, first minimum) to
 * des[i][6m .. 6m+5].  des: [n][96] fp32, codes: [n][16] u8, host pointers.  afis_gallery_add calls it for rolled texture
 * views that carry fp32 descriptors (codes == NULL, des_len == 96). */
int afis_pq_encode(afis_ctx* ctx, const float* des, int64_t n, uint8_t* codes);

/* The rolled branch of descriptor_PQ.py::encode_PQ (:332-349) for one file: `bytes` is a template with fp32 texture
 * descriptors in the latent on-disk layout (descriptor_PQ.py:80-175); the result is the same template in the rolled layout
 * (:178-272), texture descriptors replaced by PQ codes.  out == NULL only reports *out_len.  load_rc: the reader's code. */
int afis_encode_rolled_dat(afis_ctx* ctx, const void* bytes, size_t len, void* out, size_t out_cap, size_t* out_len, int* load_rc);

/* Correspondence export — replaces One2One_matching_selected_templates(..., save_corr = true, corr_file) as called for the
 * top-24 of One2List_matching (matching/matcher.cpp:321-327, :376-417, :497-505).  For one latent and each of the n listed
 * gallery templates (indices as reported by afis_search, i.e. including index_base) it re-runs the three minutiae scorers and
 * returns the correspondences that survive both graph filters, in the reference's order (corr3):
 *   counts[i*3 + s]                 number of survivors for selected latent template s (0 -> 27th, 1 -> 3rd, 2 -> 12th)
 *   xy[((i*3 + s)*120 + t)*4 + 0..3] latent x, latent y, rolled x, rolled y of survivor t  (one line of <corr_file>_<s>.csv)
 * counts is -1 where the reference does not run that scorer and writes no file (latent empty, rolled empty or without a
 * minutiae template, latent without the selected template), and 0 where it writes an empty file. */
int afis_correspondences(afis_ctx* ctx, const afis_template_view* query, const int64_t* gallery_idx, int n,
                         int32_t* counts /*[n][3]*/, int16_t* xy /*[n][3][120][4]*/);

/* afis_get_timing2 copies min(struct_size, sizeof(afis_timing)) bytes: pass sizeof(afis_timing) of the header the caller was compiled
 * against, so that a library with a longer struct never writes past the caller's.  afis_get_timing (kept for callers of the round-2
 * header) fills only the fields up to `pairs` (48 bytes, the struct of that header); the fields after it need afis_get_timing2. */
int afis_get_timing(const afis_ctx* ctx, afis_timing* out);
int afis_get_timing2(const afis_ctx* ctx, afis_timing* out, size_t struct_size);
/* Tunables (INTEGRATION.md section E has the table; timings in profiles/r04_tables.md).  "adc_variant" — every variant gives bit-identical results:
 * 9 [default] = an fp16 matrix-core pass over all (latent row, rolled point) cells bounds every row maximum and pins its candidate points; the rows that can
 * reach a pair's top 200 then get the exact fp32 value of their candidates, the table entries recomputed in the reference's arithmetic and order
 * (adc_mfma.hip, adc_refine.hip); 8 = a 16-bit fixed-point LDS-table pass bounds the candidates, which are then evaluated exactly from an fp32 table in HBM/L2
 * (the north_star's LDS-LUT design; 1.6 x the time of 9).  libafis_hip.so (the product) accepts 9 and 8 only; the direct exact kernels of rounds 1-2 — 7 = conflict-free lane classes, 1024-thread
 * workgroups (2.9 x); 6 = the same with 512; 0 = plain LDS gather, 1 = chain/row-quad rotated lanes, 2/3 = 0/1 with 1024-thread workgroups — are reference kernels built into libafis_hip_test.so only.
 * "mf_blocks" (form of variant 9's bound pass, bit-identical: 2 = two row blocks per wave, the only value of the product library; libafis_hip_test.so also takes 3), "bound_cus" (below), "query_batch" (latents per launch group),
 * "chunk" (gallery templates per workgroup), "minu_generic" (force the generic minutiae candidate kernel), "s3_tie_order" (0 [default]: candidate norms that tie — in practice the zero norms that fill a list with fewer
 * than 120 positive similarities — are taken in ascending element order; 1: in the order libstdc++'s std::sort leaves them, i.e. what the reference binary's matcher.cpp:473-476 delivers:
 * such lists — and the rare list in which two positive norms tie — then go through the any-shape candidate kernel, one wave of which runs the sort; about +2 % of a search on structured templates), "ref_tie_order" (0 [default], 1 = "s3_tie_order" 1, 2 = in addition the greedy selections of S8 and S9 — matcher.cpp:1301 / :1423 / :1590 — walk
 * equal SCORES in std::sort's order: that is where mated pairs differ, whose dozens of surviving correspondences tie exactly at S9; with 2 the scores are the reference binary's on all but three of
 * 710 000 synthetic pairs [the texture top-200 sort of S7 stays in index order]; no measurable cost beyond level 1; the match CLI: -tie <n>), "search_timeout_s" (every host wait of a search is bounded: after this many seconds
 * without the device finishing, afis_search returns AFIS_EDEVICE instead of blocking; default 600, AFIS_SEARCH_TIMEOUT_S; <= 0 = unbounded; "search_timeout_ms" sets the same bound in
 * milliseconds; afis_get_option reads "search_timeout_s" rounded UP to whole seconds and "search_timeout_ms" exactly.  After such a timeout the device may still be working on the call: the caller's output
 * buffers must stay valid until afis_destroy, or until a later call on the context succeeds; afis_queries_free then only parks the handle (its device buffers are released by the next call that finds the device idle),
 * and afis_destroy — which has to wait for the device — may block where the search did), "rowmax_budget_mb" (device memory of a launch group's per-pair
 * buffers; default 60 % of the free memory), "mf_stats" (adc_variant 9: collect the counters the parity tap afis_debug_refine_stats reads).  ("lut_dtype" accepts only 32: the
 * opt-in 16-bit tolerance path of rounds 1-2 did not meet its stated tolerance and was removed; every remaining path is bit-exact.)
 * Returns AFIS_EINVAL for unknown names. */
int afis_set_option(afis_ctx* ctx, const char* name, int64_t value);
/* The value an option has now (0 = automatic where the table says so).  Read-only: "minu_fast_max_latent" (256), "minu_fast_max_rolled" (512), "minu_fast_max_cells"
 * (38 912, counted with the odd row stride's padding column): the largest minutiae counts / latent x rolled similarities a candidate task may have to run in the matrix-core kernel's shape classes; larger tasks (the
 * reference's reader allows 2000 minutiae per template, matcher.cpp:788-790) go to the any-shape kernel — same results, slower (afis_timing.minu_fallback_tasks counts them).  "bound_cus": 128 by default — the bound pass runs on a stream confined to half of the chip's CUs
 * (hipExtStreamCreateWithCUMask) with the minutiae stage beside it on the other half: the pass is power-limited, half the CUs deliver 0.64 of its throughput;
 * 0 = one stream, kernels back to back; 32 ... 224 in steps of 32; the environment variable AFIS_BOUND_CUS sets the initial value. */
int afis_get_option(const afis_ctx* ctx, const char* name, int64_t* value);

/* The parity-test taps (stage intermediates: afis_debug_*) are NOT part of this library: they are declared in
 * include/afis_matcher_taps.h and exported only by libafis_hip_test.so (the same objects with afis_api.cpp built -DAFIS_PARITY_TAPS),
 * which tests/ load.  libafis_hip.so exports exactly the functions declared above. */

#ifdef __cplusplus
}
#endif
#endif /* AFIS_MATCHER_H */
